// Issue cost of the instructions the tile kernels are made of, on gfx950: cycles per wave-instruction of a long run of
// INDEPENDENT instructions of one kind (8 chains), with 1 and with 4 waves per SIMD (s_memtime around the run).  And two LDS
// questions: does ds_read_b32 take a 2-byte-aligned address, and what does a 64-lane ds_read_b64 gather over a 2 KB table cost.
//   hipcc --offload-arch=gfx950 -O2 valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64
#define RUN8(ASM)                                                                                         \
  for (int it = 0; it < REP; it++)                                                                        \
    asm volatile(ASM(0, 8) ASM(1, 9) ASM(2, 10) ASM(3, 11) ASM(4, 12) ASM(5, 13) ASM(6, 14) ASM(7, 15) ASM(0, 8) ASM(1, 9) ASM(2, 10) ASM(3, 11) ASM(4, 12) ASM(5, 13) ASM(6, 14) ASM(7, 15) \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),                \
                   "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])                 \
                 : "v"(c), "v"(ci));

#define A_ADD64(k, j) "v_add_f64 %" #k ", %" #k ", %16\n\t"
#define A_MUL64(k, j) "v_mul_f64 %" #k ", %" #k ", %16\n\t"
#define A_FMA64(k, j) "v_fma_f64 %" #k ", %" #k ", %16, %16\n\t"
#define A_LDEXP(k, j) "v_ldexp_f64 %" #k ", %" #k ", %17\n\t"
#define A_CVTU(k, j) "v_cvt_f64_u32 %" #k ", %" #j "\n\t"
#define A_CVTF(k, j) "v_cvt_f64_f32 %" #k ", %" #j "\n\t"
#define A_MULLO(k, j) "v_mul_lo_u32 %" #j ", %" #j ", %17\n\t"
#define A_MUL24(k, j) "v_mul_u32_u24 %" #j ", %" #j ", %17\n\t"
#define A_MAD24(k, j) "v_mad_u32_u24 %" #j ", %" #j ", %17, %17\n\t"
#define A_BFE(k, j) "v_bfe_u32 %" #j ", %" #j ", 3, 9\n\t"
#define A_MED3(k, j) "v_med3_i32 %" #j ", %" #j ", 0, %17\n\t"
#define A_ADD32(k, j) "v_add_u32 %" #j ", %" #j ", %17\n\t"
#define A_LSHLADD(k, j) "v_lshl_add_u32 %" #j ", %" #j ", 3, %17\n\t"
#define A_CVT32(k, j) "v_cvt_f32_u32 %" #j ", %" #j "\n\t"
#define A_RNDNE(k, j) "v_rndne_f64 %" #k ", %" #k "\n\t"
#define A_CVTI(k, j) "v_cvt_i32_f64 %" #j ", %" #k "\n\t"

__global__ void probe(int which, unsigned long long* out, double* sink) {
  double d[8];
  uint32_t u[8];
  for (int k = 0; k < 8; k++) { d[k] = 1.0 + threadIdx.x * 1e-3 + k; u[k] = threadIdx.x * 7 + k; }
  const double c = 1.0000001;
  const uint32_t ci = 3;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  switch (which) {
    case 0: RUN8(A_ADD64) break;
    case 1: RUN8(A_MUL64) break;
    case 2: RUN8(A_FMA64) break;
    case 3: RUN8(A_LDEXP) break;
    case 4: RUN8(A_CVTU) break;
    case 5: RUN8(A_CVTF) break;
    case 6: RUN8(A_MULLO) break;
    case 7: RUN8(A_MUL24) break;
    case 8: RUN8(A_MAD24) break;
    case 9: RUN8(A_BFE) break;
    case 10: RUN8(A_MED3) break;
    case 11: RUN8(A_ADD32) break;
    case 12: RUN8(A_LSHLADD) break;
    case 13: RUN8(A_CVT32) break;
    case 14: RUN8(A_RNDNE) break;
    case 15: RUN8(A_CVTI) break;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < 8; k++) s += d[k] + u[k];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(&out[which], t1 - t0);
}

// LDS: mode 0 ds_read_u16 x2 at 2-byte aligned addresses; 1 ds_read_b32 at the same (2 mod 4) addresses; 2 ds_read_b64 gather from a
// 256-entry table with pseudo-random indices; 3 the same with one index for all lanes (broadcast)
__global__ void lds_probe(int mode, unsigned long long* out, uint32_t* res) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = (uint8_t)(i * 7 + 1);
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)buf;
  uint32_t addr = base + 2 + (threadIdx.x & 63) * 76;                // a pair-texture row per lane
  uint32_t idx = ((threadIdx.x * 2654435761u) >> 20) & 255u;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; it++) {
    uint32_t a = 0, b = 0;
    if (mode == 0) {
      asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:2\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b) : "v"(addr) : "memory");
      acc += a | (b << 16);
    } else if (mode == 1) {
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
      acc += a;
    } else {
      const uint32_t ad = base + (mode == 2 ? idx : 5u) * 8;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&a) : "v"(ad) : "memory");
      uint64_t v;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
      acc += (uint32_t)v;
      idx = (idx * 5 + 1 + (uint32_t)v) & 255u;
    }
    addr += (acc & 1) ? 4 : 4;
    if (addr > base + 7000) addr -= 4096;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  res[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) atomicMax(&out[32 + mode], t1 - t0);
}

int main() {
  unsigned long long* d_out;
  double* d_sink;
  uint32_t* d_res;
  (void)hipMalloc(&d_out, 64 * 8); (void)hipMalloc(&d_sink, 8 * 1024 * 256); (void)hipMalloc(&d_res, 4 * 1024 * 256);
  const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_ldexp_f64", "v_cvt_f64_u32", "v_cvt_f64_f32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24",
                         "v_bfe_u32", "v_med3_i32", "v_add_u32", "v_lshl_add_u32", "v_cvt_f32_u32", "v_rndne_f64", "v_cvt_i32_f64"};
  for (int waves : {1, 4}) {          // waves per SIMD: one workgroup of 256 or 1024 threads on one CU
    hipMemset(d_out, 0, 64 * 8);
    for (int w = 0; w < 16; w++) probe<<<1, 256 * waves>>>(w, d_out, d_sink);
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, d_out, 64 * 8, hipMemcpyDeviceToHost);
    printf("%d wave(s) per SIMD: cycles per wave-instruction (a SIMD's issue time = this / waves)\n", waves);
    for (int w = 0; w < 16; w++) printf("  %-16s %6.2f   per SIMD %5.2f\n", names[w], (double)h[w] / (REP * 16.0), (double)h[w] / (REP * 16.0) / waves);
  }
  // unaligned ds_read_b32: same values as two ds_read_u16?
  hipMemset(d_out, 0, 64 * 8);
  std::vector<uint32_t> r0(64), r1(64);
  lds_probe<<<1, 64>>>(0, d_out, d_res); hipMemcpy(r0.data(), d_res, 256, hipMemcpyDeviceToHost);
  lds_probe<<<1, 64>>>(1, d_out, d_res); hipMemcpy(r1.data(), d_res, 256, hipMemcpyDeviceToHost);
  int same = 0;
  for (int i = 0; i < 64; i++) same += r0[i] == r1[i];
  printf("ds_read_b32 at 2-byte aligned addresses: %d of 64 lanes equal the two-u16 result\n", same);
  for (int waves : {1, 4, 16}) {
    hipMemset(d_out, 0, 64 * 8);
    for (int m = 0; m < 4; m++) lds_probe<<<1, 64 * waves>>>(m, d_out, d_res);
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, d_out, 64 * 8, hipMemcpyDeviceToHost);
    printf("%2d wave(s) per CU, dependent LDS round trips, cycles per trip: 2 x u16 %.1f   b32 unaligned %.1f   b64 gather (x2 per trip) %.1f   b64 broadcast (x2) %.1f\n", waves,
           h[32] / 256.0, h[33] / 256.0, h[34] / 256.0, h[35] / 256.0);
  }
  return 0;
}
