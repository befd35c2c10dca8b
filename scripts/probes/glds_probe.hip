// What gfx950's LDS-DMA loads do for lanes that are switched off in EXEC, and whether the 16-byte form needs aligned
// addresses (scripts/probes: hipcc --offload-arch=gfx950 glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline void glds_dword(const void* src, uint32_t lds_byte) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)lds_byte)) : "memory");
}
__device__ inline void glds_dwordx4(const void* src, uint32_t lds_byte) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)lds_byte)) : "memory");
}
// mode 0: dword, lanes [lo, hi) active;  mode 1: dwordx4, lanes [lo, hi) active;  src_shift: byte offset added to the source
__global__ void probe(const uint32_t* src, uint32_t* out, int mode, int lo, int hi, int src_shift) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[512];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) buf[i] = 0xAAAA0000u + i;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)buf;
  const char* s = reinterpret_cast<const char*>(src) + src_shift;
  if (lane >= lo && lane < hi) {
    if (mode == 0) glds_dword(s + lane * 4, base + 256);
    else glds_dwordx4(s + lane * 16, base + 256);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = buf[i];
}
int main() {
  std::vector<uint32_t> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = 0x10000u + i;
  uint32_t *d_src, *d_out;
  hipMalloc(&d_src, 4096); hipMalloc(&d_out, 2048);
  hipMemcpy(d_src, h.data(), 4096, hipMemcpyHostToDevice);
  struct { int mode, lo, hi, shift; } cases[] = {{0, 0, 64, 0}, {0, 8, 40, 0}, {1, 0, 64, 0}, {1, 8, 40, 0}, {1, 0, 64, 8}, {1, 0, 64, 4}, {0, 0, 64, 2}};
  for (auto c : cases) {
    probe<<<1, 64>>>(d_src, d_out, c.mode, c.lo, c.hi, c.shift);
    std::vector<uint32_t> o(512);
    hipMemcpy(o.data(), d_out, 2048, hipMemcpyDeviceToHost);
    int changed = 0, first = -1, last = -1, wrong = 0, zeros = 0;
    const int per = c.mode ? 4 : 1;
    for (int i = 0; i < 512; i++) {
      if (o[i] != 0xAAAA0000u + i) {
        changed++;
        if (first < 0) first = i;
        last = i;
        if (o[i] == 0) zeros++;
        const int k = i - 64;                                   // dword index behind the destination base (256 bytes in)
        const uint32_t want = 0x10000u + k + c.shift / 4;
        if (c.shift % 4 == 0 && o[i] != want) wrong++;
      }
    }
    printf("mode %s lanes [%d,%d) src+%d: %d dwords changed, first %d last %d (expected %d..%d), %d not the expected value, %d zeros; dword[64+%d]=%08x\n",
           c.mode ? "x4" : "x1", c.lo, c.hi, c.shift, changed, first, last, 64 + c.lo * per, 64 + c.hi * per - 1, wrong, zeros, c.lo * per, o[64 + c.lo * per]);
  }
  return 0;
}
