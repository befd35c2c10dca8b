// pcie_probe.hip -- how fast do frame-sized buffers cross PCIe, alone and in both directions at once, by hipMemcpyAsync (SDMA)
// and by a copy kernel that reads / writes pinned host memory directly?  Decides how the host-pointer pipeline
// (rr_pipeline_submit) moves its batches.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 pcie_probe.hip -o /tmp/pcie_probe && /tmp/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_spin(float* out, int iters) {      // something for the CUs to do meanwhile
  float a = threadIdx.x;
  for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) out[0] = a;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t frame = 1242 * 375 * 3 + 1242 * 375 * 4 + 7200 * 112;   // what one KITTI frame uploads (u8 image + f32 depth + drops)
  const int F = 64, reps = 6;
  const size_t bytes = frame * F / 16 * 16;
  void *h_up, *h_down, *d_up, *d_down;
  CK(hipHostMalloc(&h_up, bytes, hipHostMallocDefault));
  CK(hipHostMalloc(&h_down, bytes, hipHostMallocDefault));
  CK(hipMalloc(&d_up, bytes));
  CK(hipMalloc(&d_down, bytes));
  memset(h_up, 1, bytes);
  memset(h_down, 0, bytes);
  CK(hipMemset(d_down, 2, bytes));
  hipStream_t s_up, s_down, s_k;
  CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
  float* d_spin;
  CK(hipMalloc(&d_spin, 4));
  auto timeit = [&](const char* name, auto fn, double moved) {
    fn();
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < reps; r++) fn();
    CK(hipDeviceSynchronize());
    const double dt = (now() - t0) / reps;
    printf("%-64s %7.2f ms  %6.1f GB/s (sum of directions)\n", name, dt * 1e3, moved / dt / 1e9);
  };
  for (int pieces : {1, 64 * 3}) {       // one big copy, or frame-sized pieces
    const size_t pb = bytes / pieces / 16 * 16;
    printf("-- %d piece(s) of %.2f MB\n", pieces, pb / 1e6);
    auto up_sdma = [&] { for (int p = 0; p < pieces; p++) CK(hipMemcpyAsync((char*)d_up + p * pb, (char*)h_up + p * pb, pb, hipMemcpyHostToDevice, s_up)); };
    auto down_sdma = [&] { for (int p = 0; p < pieces; p++) CK(hipMemcpyAsync((char*)h_down + p * pb, (char*)d_down + p * pb, pb, hipMemcpyDeviceToHost, s_down)); };
    timeit("H2D hipMemcpyAsync", up_sdma, (double)pb * pieces);
    timeit("D2H hipMemcpyAsync", down_sdma, (double)pb * pieces);
    timeit("H2D + D2H hipMemcpyAsync, two streams", [&] { up_sdma(); down_sdma(); }, 2.0 * pb * pieces);
    for (int blocks : {32, 128, 512}) {
      char nm[128];
      auto up_k = [&] { hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, s_up, (const uint4*)h_up, (uint4*)d_up, pb * pieces / 16); };
      auto down_k = [&] { hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, s_down, (const uint4*)d_down, (uint4*)h_down, pb * pieces / 16); };
      snprintf(nm, sizeof nm, "H2D copy kernel reading pinned host memory, %d blocks", blocks);
      timeit(nm, up_k, (double)pb * pieces);
      snprintf(nm, sizeof nm, "D2H copy kernel writing pinned host memory, %d blocks", blocks);
      timeit(nm, down_k, (double)pb * pieces);
      snprintf(nm, sizeof nm, "H2D kernel (%d blocks) + D2H hipMemcpyAsync", blocks);
      timeit(nm, [&] { up_k(); down_sdma(); }, 2.0 * pb * pieces);
      snprintf(nm, sizeof nm, "H2D hipMemcpyAsync + D2H kernel (%d blocks)", blocks);
      timeit(nm, [&] { up_sdma(); down_k(); }, 2.0 * pb * pieces);
      snprintf(nm, sizeof nm, "H2D kernel + D2H kernel (%d blocks each)", blocks);
      timeit(nm, [&] { up_k(); down_k(); }, 2.0 * pb * pieces);
    }
    // with the CUs busy (a compute kernel filling the chip on a third stream)
    timeit("H2D + D2H hipMemcpyAsync under a chip-filling kernel", [&] { hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s_k, d_spin, 400000); up_sdma(); down_sdma(); }, 2.0 * pb * pieces);
    timeit("H2D kernel(128) + D2H hipMemcpyAsync under a chip-filling kernel", [&] { hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s_k, d_spin, 400000);
      hipLaunchKernelGGL(k_copy16, dim3(128), dim3(256), 0, s_up, (const uint4*)h_up, (uint4*)d_up, pb * pieces / 16); down_sdma(); }, 2.0 * pb * pieces);
    timeit("(the chip-filling kernel alone)", [&] { hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s_k, d_spin, 400000); }, 0.0);
  }
  // correctness of the kernel paths
  CK(hipMemcpy(h_down, d_down, 64, hipMemcpyDeviceToHost));
  printf("check: d_down[0] = %d (2), h_up[0] = %d (1)\n", ((unsigned char*)h_down)[0], ((unsigned char*)h_up)[0]);
  return 0;
}
