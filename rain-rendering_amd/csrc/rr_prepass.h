// rr_prepass.h -- gfx950 kernels of the two per-frame pre-passes that feed the hot path
// (SURVEY 8f "next" #1, #2):
//   fog-like rain attenuation     reference common/add_attenuation.py:26-95
//   environment-map estimation    reference common/bad_weather.py:742-853 + generator.py:407-408
// Both are HBM-bound image passes (one thread per pixel, coalesced rows); they exist so that
// rainy_bg and env_xyY are born in HBM instead of crossing PCIe as float64 planes.
// Arithmetic follows the numpy restatement in oracle/prepass.py op for op: the Gaussian taps
// are summed in scipy.ndimage.correlate1d's symmetric order, borders are BORDER_REFLECT_101.
// The per-pixel bodies are `__host__ __device__` (like rr_device.h) so that tests/hostemu can run
// the very same arithmetic on the CPU; the __global__ wrappers only compute indices.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RRP_HD __host__ __device__ inline
#else
#define RRP_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace rrpre {

constexpr int KMAX = 33;            // max Gaussian taps

struct Kernels {                    // passed by value to the kernels
  int fog_k, env_k;
  double fog_w[KMAX];
  double env_w[KMAX];
};

// element types of a frame's arrays (PreFrame.types).  The arithmetic is float64 whatever the types: a float32 output is
// the float64 result rounded once, a uint8 image is bytes / 255.0 (generator.py:352) formed where it is read.
enum { PRE_BG_F32 = 1, PRE_BG_U8 = 2, PRE_RAINY_F32 = 4, PRE_ENV_F32 = 8, PRE_DEPTH_U16 = 16 };

struct PreFrame {
  const void* bg;                   // H*W*3: float64, float32 (PRE_BG_F32) or uint8 (PRE_BG_U8)
  const void* depth;                // H*W float32 or float64 metres, or (PRE_DEPTH_U16, depth_f64 == 0) the uint16 samples of the depth
                                    // file: metres = sample / 256 in float32 (generator.py:366: exact, a power of two)
  void* rainy;                      // H*W*3: float64 or float32 (PRE_RAINY_F32)
  void* env_xyY;                    // H*We*3: float64 or float32 (PRE_ENV_F32) (may be null)
  uint8_t* env_u8;                  // H*We*3 BGR (may be null)
  uint8_t* r8;                      // H*W*3 (rainy * 255).astype(uint8) of the fog pass' float64 result: what the map's gather
                                    // reads (bad_weather.py:764); null in the map-only mode (the gather then reads `bg`)
  double beta_ext, beta_hg, irr_num, irr_den;
  int32_t depth_f64, types;
};

RRP_HD double load_unit(const void* a, int types, int64_t i) {          // one element of `bg`
  if (types & PRE_BG_U8) return (double)((const uint8_t*)a)[i] / 255.0;
  if (types & PRE_BG_F32) return (double)((const float*)a)[i];
  return ((const double*)a)[i];
}
RRP_HD void store_rainy(const void* a, int types, int64_t i, double v) {
  if (types & PRE_RAINY_F32) ((float*)const_cast<void*>(a))[i] = (float)v;
  else ((double*)const_cast<void*>(a))[i] = v;
}
RRP_HD uint8_t unit_to_byte(double v) { return (uint8_t)((uint32_t)(int)(v * 255.0) & 255u); }   // (image * 255).astype(uint8)

struct EnvGeom {
  int H, W, cw, lw, We;
  const int32_t* src;               // [H*cw] source pixel index (r*W+c) or -1
  const int32_t* top_row;           // [cw] row the unfilled pixels of the top half copy
  const int32_t* bot_row;           // [cw] same, bottom half
  const uint8_t* need_h;            // [H*We] 1 where the horizontal blur's value is read by some unfilled cell's vertical blur
                                    // (build_env_need; a function of the geometry and the tap count); null = every cell
};

struct PreScratch {                 // [frame][...]
  double* fext;                     // [H*W]       (these three: tap counts other than 25 only -- the three-kernel form)
  double* tmpF;                     // [H*W]
  double* tmpL;                     // [3][H*W] (planar: the vertical pass reads whole rows of one plane)
  uint8_t* r8;                      // [H*W*3] PreFrame.r8 of the frames
  double* part;                     // [64*3]
  double* mean;                     // [3]
  uint32_t* epack;                  // [H*We] b | g<<8 | r<<16 | mask<<24
  double* etmp;                     // [H*We*3]
};

constexpr int FOG_BLOCKS = 64;

RRP_HD double one_minus(double fe, int f64) { return f64 ? 1.0 - fe : (double)(1.0f - (float)fe); }
RRP_HD double clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }

RRP_HD int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

#if defined(__HIPCC__)
// irradiance = (4 N^2 image) / (t G pi); per-channel mean           add_attenuation.py:51-58
__global__ void __launch_bounds__(256) k_fog_sum(const PreFrame* fr, int H, int W, PreScratch sc) {
  const int f = blockIdx.y;
  const PreFrame F = fr[f];
  const int64_t px = (int64_t)H * W;
  double s0 = 0, s1 = 0, s2 = 0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < px; p += (int64_t)FOG_BLOCKS * 256) {
    s0 += (F.irr_num * load_unit(F.bg, F.types, p * 3 + 0)) / F.irr_den;
    s1 += (F.irr_num * load_unit(F.bg, F.types, p * 3 + 1)) / F.irr_den;
    s2 += (F.irr_num * load_unit(F.bg, F.types, p * 3 + 2)) / F.irr_den;
  }
  __shared__ double red[3][256];
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  red[2][threadIdx.x] = s2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int c = 0; c < 3; c++) red[c][threadIdx.x] += red[c][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 3) sc.part[((int64_t)f * FOG_BLOCKS + blockIdx.x) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void k_fog_mean(int H, int W, PreScratch sc) {
  const int f = blockIdx.x, c = threadIdx.x;
  if (c >= 3) return;
  double s = 0;
  for (int b = 0; b < FOG_BLOCKS; b++) s += sc.part[((int64_t)f * FOG_BLOCKS + b) * 3 + c];
  sc.mean[f * 3 + c] = s / (double)((int64_t)H * W);
}

#endif

// extinction map f_ext = exp(-beta_ext * depth_km)                      add_attenuation.py:45-49
RRP_HD double fog_ext_val(const PreFrame& F, double d64, float d32) {
  if (F.depth_f64) return exp(-F.beta_ext * (d64 / 1000.0));
  const float t = d32 / 1000.0f;    // numpy keeps float32: float32 / int, weak python scalar * float32
  return (double)expf((float)(-F.beta_ext) * t);
}
RRP_HD float depth_f32_at(const PreFrame& F, int64_t p) {     // (depth_f64 == 0)
  return (F.types & PRE_DEPTH_U16) ? (float)((const uint16_t*)F.depth)[p] / 256.0f : ((const float*)F.depth)[p];
}
RRP_HD void fog_ext_px(const PreFrame& F, int f, int H, int W, const PreScratch& sc, int64_t p) {
  const int64_t px = (int64_t)H * W;
  sc.fext[f * px + p] = F.depth_f64 ? fog_ext_val(F, ((const double*)F.depth)[p], 0.0f) : fog_ext_val(F, 0.0, depth_f32_at(F, p));
}


// The four source planes of the fog blur at one pixel: f_ext and l_in = clip(beta_hg * mean * (1 - f_ext))
// per channel (:66-73).  Evaluated once per SOURCE pixel, then shared by the taps that read it.
RRP_HD void fog_src(double fe, int f64, const double k3[3], double out[4]) {
  const double om = one_minus(fe, f64);
  out[0] = fe;
  for (int c = 0; c < 3; c++) out[1 + c] = clip01(k3[c] * om);
}

// horizontal pass (:79-80) at staged position i: S[p][i + j], j = -half..half, are plane p's taps
// (LDS on the device, plain arrays in tests/hostemu).  scipy.ndimage.correlate1d's symmetric order.
RRP_HD void fog_h_taps(const double* S, int pitch, int i, const Kernels& kn, int f64, double out[4]) {
  const int half = kn.fog_k / 2;
  for (int p = 0; p < 4; p++) {
    const double* v = S + p * pitch + i;
    double a = v[0] * kn.fog_w[half];
    for (int j = -half; j < 0; j++) a += (v[j] + v[-j]) * kn.fog_w[half + j];
    out[p] = a;
  }
  if (!f64) out[0] = (double)(float)out[0];          // the float32 depth path keeps f_ext in float32
}

constexpr int FOG_SEG = 256;        // output columns per staged row segment

// stages plane values of columns x0-half .. x0+n+half-1 of row y (reflect-101) into S[4][pitch]
RRP_HD void fog_stage_px(const PreFrame& F, int f, int H, int W, const Kernels& kn, const PreScratch& sc, int y, int x0, int i,
                         double* S, int pitch) {
  const int half = kn.fog_k / 2;
  const double fe = sc.fext[(int64_t)f * H * W + (int64_t)y * W + reflect101(x0 - half + i, W)];
  double k3[3], o[4];
  for (int c = 0; c < 3; c++) k3[c] = F.beta_hg * sc.mean[f * 3 + c];
  fog_src(fe, F.depth_f64, k3, o);
  for (int p = 0; p < 4; p++) S[p * pitch + i] = o[p];
}
RRP_HD void fog_h_store(int f, int H, int W, const PreScratch& sc, int y, int x, const double o[4]) {
  const int64_t px = (int64_t)H * W, q = (int64_t)y * W + x;
  sc.tmpF[f * px + q] = o[0];
  for (int c = 0; c < 3; c++) sc.tmpL[(f * 3 + c) * px + q] = o[1 + c];
}

// vertical pass + rainy = clip(image * f_ext + l_in, 0, 1)  (:82-86,93), any tap count: one output per call
RRP_HD void fog_v_px(const PreFrame& F, int f, int H, int W, const Kernels& kn, const PreScratch& sc, int y, int x) {
  const int64_t px = (int64_t)H * W, base = f * px;
  const int half = kn.fog_k / 2;
  const int64_t p0 = base + (int64_t)y * W + x;
  double aF = sc.tmpF[p0] * kn.fog_w[half], aL[3];
  for (int c = 0; c < 3; c++) aL[c] = sc.tmpL[(p0 - base) + (f * 3 + c) * px] * kn.fog_w[half];
  for (int j = -half; j < 0; j++) {
    const int64_t pa = base + (int64_t)reflect101(y + j, H) * W + x, pb = base + (int64_t)reflect101(y - j, H) * W + x;
    const double w = kn.fog_w[half + j];
    aF += (sc.tmpF[pa] + sc.tmpF[pb]) * w;
    for (int c = 0; c < 3; c++) aL[c] += (sc.tmpL[(pa - base) + (f * 3 + c) * px] + sc.tmpL[(pb - base) + (f * 3 + c) * px]) * w;
  }
  if (!F.depth_f64) aF = (double)(float)aF;
  const int64_t q = ((int64_t)y * W + x) * 3;
  for (int c = 0; c < 3; c++) {
    const double v = clip01(load_unit(F.bg, F.types, q + c) * aF + aL[c]);
    store_rainy(F.rainy, F.types, q + c, v);
    if (F.r8) F.r8[q + c] = unit_to_byte(v);
  }
}

// --- the 25-tap fog layer in ONE kernel (round 4) --------------------------------------------------------------------
// The three-kernel form above writes f_ext, then four planes of horizontal sums, then reads those planes back 32 rows per
// 8 output rows: 150 B of HBM traffic per pixel for a result of 24 B.  FogTile keeps all of it in LDS: a workgroup owns
// TC columns and walks down a segment of rows RB rows at a time;
//   stage   RB source rows x (TC + 2 HALF) columns: f_ext from the depth (fog_ext_val) and the three l_in planes -> S
//   hpass   the horizontal sums of those RB rows -> a ring of RB + 2 HALF rows of horizontal sums
//   vpass   the vertical sums of the RB output rows whose window is now complete, the blend with the image, the stores.
// Rows and columns beyond the frame are staged from their BORDER_REFLECT_101 sources, so that the passes never index
// outside [0, H) x [0, W); every sum folds its taps in the order of fog_h_taps / fog_v_px above, so that the float64
// result is THE SAME BITS as the three-kernel form's (tests/test_prepass_hostemu.py runs both on the host).
// Thread roles (256 threads; the __global__ wrapper and tests/hostemu call the same functions with tid = 0..255):
//   stage   item i = tid, tid + 256 < RB * PITCH: source row i / PITCH, staged column i % PITCH
//   hpass   row tid >> 5, planes 2 * ((tid >> 4) & 1) + {0, 1}, columns 2 * (tid & 15) + {0, 1}: a window of 26 staged
//           values gives both columns' sums (13 taps each side shared)
//   vpass   wave w = tid >> 6: columns 16 * (w & 1) + (lane & 15), rows 4 * (w >> 1) + {0..3}, plane lane >> 4: a window of
//           28 ring rows gives the four rows' sums; planes 1..3 need plane 0's sum of their pixel: lane & 15 of the same wave
//           (a wave shuffle on the device), then store 48 consecutive channel values per row.
template <int HALF>
struct FogTile {
  static constexpr int TC = 32, RB = 8, PITCH = TC + 2 * HALF, RING = RB + 2 * HALF, NB = RING / RB, VR = 4;
  static_assert((2 * HALF) % RB == 0 && RB * TC == 256 && PITCH % 2 == 0, "tile shape");
  typedef double d2 __attribute__((vector_size(16)));

  // (source pixel index of stage item i of h-block k, or -1) -- the depth value is loaded by the caller (the device
  // prefetches the next block's while this block's sums are folded)
  RRP_HD static int64_t stage_src(int H, int W, int x0, int hs, int k, int i) {
    if (i >= RB * PITCH) return -1;
    const int rr = i / PITCH, ci = i - rr * PITCH;
    return (int64_t)reflect101(hs + k * RB + rr, H) * W + reflect101(x0 - HALF + ci, W);
  }
  RRP_HD static void stage_put(const PreFrame& F, const double k3[3], int i, double d64, float d32, double* S) {
    const int rr = i / PITCH, ci = i - rr * PITCH;
    double o[4];
    fog_src(fog_ext_val(F, d64, d32), F.depth_f64, k3, o);
    for (int p = 0; p < 4; p++) S[(rr * 4 + p) * PITCH + ci] = o[p];
  }
  RRP_HD static void hpass(const Kernels& kn, int f64, int k, int tid, const double* S, double* ring) {
    const int rr = tid >> 5, pp = (tid >> 4) & 1, cp = tid & 15;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int p = 2 * pp + u;
      const d2* src = reinterpret_cast<const d2*>(S + (rr * 4 + p) * PITCH + 2 * cp);
      double v[2 * HALF + 2];
#pragma unroll
      for (int j = 0; j <= HALF; j++) {
        const d2 t = src[j];
        v[2 * j] = t[0];
        v[2 * j + 1] = t[1];
      }
      double a0 = v[HALF] * kn.fog_w[HALF], a1 = v[HALF + 1] * kn.fog_w[HALF];
#pragma unroll
      for (int j = -HALF; j < 0; j++) {
        a0 += (v[HALF + j] + v[HALF - j]) * kn.fog_w[HALF + j];
        a1 += (v[HALF + 1 + j] + v[HALF + 1 - j]) * kn.fog_w[HALF + j];
      }
      if (p == 0 && !f64) {                          // the float32 depth path keeps f_ext in float32
        a0 = (double)(float)a0;
        a1 = (double)(float)a1;
      }
      d2 o;
      o[0] = a0;
      o[1] = a1;
      *reinterpret_cast<d2*>(ring + (((k & (NB - 1)) * RB + rr) * 4 + p) * TC + 2 * cp) = o;
    }
  }
  // the VR vertical sums of v-block m (whose window is h-blocks m .. m + NB - 1) for thread tid's column and plane
  RRP_HD static void vtaps(const Kernels& kn, int f64, int m, int tid, const double* ring, double a[VR]) {
    const int w = tid >> 6, lane = tid & 63, col = 16 * (w & 1) + (lane & 15), p = lane >> 4, r0 = VR * (w >> 1);
    double v[VR + 2 * HALF];
#pragma unroll
    for (int j = 0; j < VR + 2 * HALF; j++) {
      const int q = r0 + j;                          // row of the 32-row window
      v[j] = ring[((((m + (q / RB)) & (NB - 1)) * RB + (q % RB)) * 4 + p) * TC + col];
    }
#pragma unroll
    for (int r = 0; r < VR; r++) {
      double s = v[HALF + r] * kn.fog_w[HALF];
#pragma unroll
      for (int j = -HALF; j < 0; j++) s += (v[HALF + r + j] + v[HALF + r - j]) * kn.fog_w[HALF + j];
      a[r] = (p == 0 && !f64) ? (double)(float)s : s;
    }
  }
  // rainy = clip(image * f_ext + l_in, 0, 1) (:93) of the thread's channel; aF = plane 0's sums of the same pixels
  RRP_HD static void vstore(const PreFrame& F, int H, int W, int x0, int ys, int ye, int m, int tid, const double a[VR], const double aF[VR]) {
    const int w = tid >> 6, lane = tid & 63, x = x0 + 16 * (w & 1) + (lane & 15), p = lane >> 4, y0 = ys + m * RB + VR * (w >> 1);
    if (p == 0 || x >= W) return;
    double img[VR];
#pragma unroll
    for (int r = 0; r < VR; r++) {
      const int y = y0 + r < ye ? y0 + r : ye - 1;
      img[r] = load_unit(F.bg, F.types, ((int64_t)y * W + x) * 3 + (p - 1));
    }
#pragma unroll
    for (int r = 0; r < VR; r++) {
      if (y0 + r >= ye) break;
      const int64_t q = ((int64_t)(y0 + r) * W + x) * 3 + (p - 1);
      const double v = clip01(img[r] * aF[r] + a[r]);
      store_rainy(F.rainy, F.types, q, v);
      if (F.r8) F.r8[q] = unit_to_byte(v);
    }
  }
};

// --- environment map ------------------------------------------------------------------------
RRP_HD uint32_t bg8_px(const PreFrame& F, int32_t p) {      // (background*255).astype(uint8)
  const int64_t q = (int64_t)p * 3;
  if (F.r8) return (uint32_t)F.r8[q] | ((uint32_t)F.r8[q + 1] << 8) | ((uint32_t)F.r8[q + 2] << 16);
  const uint32_t b = unit_to_byte(load_unit(F.bg, F.types, q + 0));
  const uint32_t g = unit_to_byte(load_unit(F.bg, F.types, q + 1));
  const uint32_t r = unit_to_byte(load_unit(F.bg, F.types, q + 2));
  return b | (g << 8) | (r << 16);
}

// the cylinder column a map column shows: centre strip, mirrored left and right sides
RRP_HD int env_col(int cw, int x) {
  const int lw = cw / 2, We = cw + 2 * lw, wr = cw - cw / 2;
  if (x >= We - wr) return cw - 1 - (x - (We - wr));          // right side (written last, :806-811)
  if (x < lw) return lw - 1 - x;                               // left side (:798-804)
  return x - lw;                                               // centre (:792-796)
}

// cylindrical un-projection, column fills, mirrored sides                    bad_weather.py:742-813
RRP_HD void env_build_px(const PreFrame& F, int f, const EnvGeom& g, const PreScratch& sc, int r, int x) {
  const int c = env_col(g.cw, x);
  const int32_t s = g.src[(int64_t)r * g.cw + c];
  const int half = g.H / 2;
  uint32_t v = 0;
  if (s >= 0) {
    v = bg8_px(F, s) | 0xff000000u;
  } else {
    int rr = -1;
    if (r < half) rr = g.top_row[c];
    else if (r >= g.H - half) rr = g.bot_row[c];
    if (rr >= 0) {
      const int32_t s2 = g.src[(int64_t)rr * g.cw + c];
      if (s2 >= 0) v = bg8_px(F, s2);
    }
  }
  sc.epack[((int64_t)f * g.H + r) * g.We + x] = v;
}

// (only the cells without a source pixel are blurred (:815-817) -- a tenth of the map -- and which they are is a property of
// the geometry: the horizontal sums nobody reads are not made)
RRP_HD void env_h_px(int f, const EnvGeom& g, const Kernels& kn, const PreScratch& sc, int r, int x) {
  if (g.need_h && !g.need_h[(int64_t)r * g.We + x]) return;
  const uint32_t* row = sc.epack + ((int64_t)f * g.H + r) * g.We;
  const int half = kn.env_k / 2;
  const uint32_t v0 = row[x];
  double a[3];
  for (int c = 0; c < 3; c++) a[c] = (double)((v0 >> (8 * c)) & 255u) * kn.env_w[half];
  for (int j = -half; j < 0; j++) {
    const uint32_t va = row[reflect101(x + j, g.We)], vb = row[reflect101(x - j, g.We)];
    const double w = kn.env_w[half + j];
    for (int c = 0; c < 3; c++) a[c] += ((double)((va >> (8 * c)) & 255u) + (double)((vb >> (8 * c)) & 255u)) * w;
  }
  const int64_t plane = (int64_t)g.H * g.We;                    // planar [channel][row][column]: the vertical pass reads columns
  double* o = sc.etmp + (int64_t)f * 3 * plane + (int64_t)r * g.We + x;
  o[0] = a[0];
  o[plane] = a[1];
  o[2 * plane] = a[2];
}

// vertical pass where the map is unfilled (:815-817), /255, RGB -> xyY (my_utils.py:55-68), nan -> 0
// the three bytes of map cell (r, x), as doubles 0..255
RRP_HD void env_v_bgr(int f, const EnvGeom& g, const Kernels& kn, const PreScratch& sc, int r, int x, double bgr[3]) {
  const int64_t plane = (int64_t)g.H * g.We, q = (int64_t)r * g.We + x;
  const uint32_t v0 = sc.epack[(int64_t)f * plane + q];
  if ((v0 >> 24) == 0) {
    const int half = kn.env_k / 2;
    const double* t = sc.etmp + (int64_t)f * 3 * plane;
    double a[3];
    for (int c = 0; c < 3; c++) a[c] = t[c * plane + q] * kn.env_w[half];
    for (int j = -half; j < 0; j++) {
      const int64_t qa = (int64_t)reflect101(r + j, g.H) * g.We + x, qb = (int64_t)reflect101(r - j, g.H) * g.We + x;
      const double w = kn.env_w[half + j];
      for (int c = 0; c < 3; c++) a[c] += (t[c * plane + qa] + t[c * plane + qb]) * w;
    }
    for (int c = 0; c < 3; c++) {
      double v = rint(a[c]);
      bgr[c] = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    }
  } else {
    for (int c = 0; c < 3; c++) bgr[c] = (double)((v0 >> (8 * c)) & 255u);
  }
}
// RGB / 255 -> xyY (my_utils.py:55-68), nan -> 0
RRP_HD void env_xyY_of(double R, double G, double B, double out[3]) {
  const double X = (R * 0.49000 + G * 0.17697 + B * 0.00000) / 0.17697;
  const double Y = (R * 0.31000 + G * 0.81240 + B * 0.01000) / 0.17697;
  const double Z = (R * 0.20000 + G * 0.01063 + B * 0.99000) / 0.17697;
  const double s = X + Y + Z;
  double xx = X / s, yy = Y / s;
  if (xx != xx) xx = 0.0;
  if (yy != yy) yy = 0.0;
  out[0] = xx;
  out[1] = yy;
  out[2] = Y;
}
RRP_HD void env_v_out(const PreFrame& F, const EnvGeom& g, int r, int x, const double bgr[3]) {
  const int64_t q = (int64_t)r * g.We + x;
  if (F.env_u8) {
    F.env_u8[q * 3 + 0] = (uint8_t)bgr[0];
    F.env_u8[q * 3 + 1] = (uint8_t)bgr[1];
    F.env_u8[q * 3 + 2] = (uint8_t)bgr[2];
  }
  if (F.env_xyY) {
    double o3[3];
    env_xyY_of(bgr[2] / 255.0, bgr[1] / 255.0, bgr[0] / 255.0, o3);
    if (F.types & PRE_ENV_F32) {
      float* o = (float*)F.env_xyY + q * 3;
      o[0] = (float)o3[0];
      o[1] = (float)o3[1];
      o[2] = (float)o3[2];
    } else {
      double* o = (double*)F.env_xyY + q * 3;
      o[0] = o3[0];
      o[1] = o3[1];
      o[2] = o3[2];
    }
  }
}
RRP_HD void env_v_px(const PreFrame& F, int f, const EnvGeom& g, const Kernels& kn, const PreScratch& sc, int r, int x) {
  double bgr[3];
  env_v_bgr(f, g, kn, sc, r, x, bgr);
  env_v_out(F, g, r, x, bgr);
}

// Host: the cells of the horizontal blur that some unfilled cell's vertical blur (half taps up and down, reflected rows) reads.
inline void build_env_need(int H, int cw, int half, const int32_t* src, uint8_t* need) {
  const int We = cw + 2 * (cw / 2);
  for (int64_t i = 0; i < (int64_t)H * We; i++) need[i] = 0;
  for (int r = 0; r < H; r++)
    for (int x = 0; x < We; x++)
      if (src[(int64_t)r * cw + env_col(cw, x)] < 0)
        for (int j = -half; j <= half; j++) need[(int64_t)reflect101(r + j, H) * We + x] = 1;
}

// Host side of rr_set_envmap_geometry: cell -> source-pixel table and the column-fill rows of
// fill_matrices (bad_weather.py:821-853): the first filled row of each column seen from the top of the
// top half / from the bottom over rows [H/2, H); argmax of an all-false column is its first row.
// src: H*cw, top/bot: cw.  Returns false when a cell or source pixel lies outside the frame.
inline bool build_env_tables(int H, int W, int cw, int n_uniq, const int32_t* uniq, const int32_t* first, int32_t* src,
                             int32_t* top, int32_t* bot) {
  for (int64_t i = 0; i < (int64_t)H * cw; i++) src[i] = -1;
  for (int i = 0; i < n_uniq; i++) {
    if (uniq[i] < 0 || uniq[i] >= H * cw || first[i] < 0 || first[i] >= H * W) return false;
    src[uniq[i]] = first[i];
  }
  const int half = H / 2;
  for (int c = 0; c < cw; c++) {
    int t = 0;
    for (int r = 0; r < half; r++)
      if (src[(int64_t)r * cw + c] >= 0) {
        t = r;
        break;
      }
    top[c] = t;
    int b = H - 1;
    for (int r = H - 1; r >= half; r--)
      if (src[(int64_t)r * cw + c] >= 0) {
        b = r;
        break;
      }
    bot[c] = b;
  }
  return true;
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(256) k_fog_ext(const PreFrame* fr, int H, int W, PreScratch sc) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < (int64_t)H * W) fog_ext_px(fr[blockIdx.y], blockIdx.y, H, W, sc, p);
}
// one workgroup per (row, segment of FOG_SEG columns): the segment plus its halos is staged in LDS as
// four planes, then every thread folds its 2*half+1 taps from LDS
__global__ void __launch_bounds__(256) k_fog_h(const PreFrame* fr, int H, int W, Kernels kn, PreScratch sc) {
  constexpr int PITCH = FOG_SEG + KMAX - 1;
  __shared__ double S[4 * PITCH];
  const int f = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * FOG_SEG, half = kn.fog_k / 2;
  const PreFrame F = fr[f];
  const int n = min(FOG_SEG, W - x0) + 2 * half;
  for (int i = threadIdx.x; i < n; i += 256) fog_stage_px(F, f, H, W, kn, sc, y, x0, i, S, PITCH);
  __syncthreads();
  const int x = x0 + threadIdx.x;
  if (x < W) {
    double o[4];
    fog_h_taps(S, PITCH, half + threadIdx.x, kn, F.depth_f64, o);
    fog_h_store(f, H, W, sc, y, x, o);
  }
}
__global__ void __launch_bounds__(256) k_fog_v(const PreFrame* fr, int H, int W, Kernels kn, PreScratch sc) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < W) fog_v_px(fr[blockIdx.z], blockIdx.z, H, W, kn, sc, blockIdx.y, x);
}
// the fog layer of one column strip and row segment (FogTile above); grid (strips, segments, frames)
template <int HALF>
__global__ void __launch_bounds__(256, 3) k_fog_tile(const PreFrame* fr, int H, int W, int seg_rows, Kernels kn, PreScratch sc) {
  using T = FogTile<HALF>;
  __shared__ double S[T::RB * 4 * T::PITCH] __attribute__((aligned(16)));
  __shared__ double ring[T::RING * 4 * T::TC] __attribute__((aligned(16)));
  const int f = blockIdx.z, tid = threadIdx.x, x0 = blockIdx.x * T::TC, ys = blockIdx.y * seg_rows;
  const PreFrame F = fr[f];
  const int ye = min(H, ys + seg_rows), hs = ys - HALF;
  const int n_iter = (ye - ys + T::RB - 1) / T::RB + T::NB - 1;
  double k3[3];
  for (int c = 0; c < 3; c++) k3[c] = F.beta_hg * sc.mean[f * 3 + c];
  // the depth values of the NEXT block's stage items are in flight while this block's sums are folded
  double d64[2] = {0.0, 0.0};
  float d32[2] = {0.0f, 0.0f};
  auto fetch = [&](int k) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int64_t p = T::stage_src(H, W, x0, hs, k, tid + 256 * u);
      if (p < 0) continue;
      if (F.depth_f64) d64[u] = ((const double*)F.depth)[p];
      else d32[u] = depth_f32_at(F, p);
    }
  };
  fetch(0);
  for (int k = 0; k < n_iter; k++) {
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (tid + 256 * u < T::RB * T::PITCH) T::stage_put(F, k3, tid + 256 * u, d64[u], d32[u], S);
    if (k + 1 < n_iter) fetch(k + 1);
    __syncthreads();
    T::hpass(kn, F.depth_f64, k, tid, S, ring);
    __syncthreads();                                   // (also: the next stage may overwrite S, the next hpass the oldest ring rows)
    const int m = k - (T::NB - 1);
    if (m >= 0) {
      double a[T::VR], aF[T::VR];
      T::vtaps(kn, F.depth_f64, m, tid, ring, a);
#pragma unroll
      for (int r = 0; r < T::VR; r++) aF[r] = __shfl(a[r], (int)(threadIdx.x & 15));
      T::vstore(F, H, W, x0, ys, ye, m, tid, a, aF);
    }
  }
}
__global__ void __launch_bounds__(256) k_env_build(const PreFrame* fr, EnvGeom g, PreScratch sc) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < g.We) env_build_px(fr[blockIdx.z], blockIdx.z, g, sc, blockIdx.y, x);
}
__global__ void __launch_bounds__(256) k_env_h(EnvGeom g, Kernels kn, PreScratch sc) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < g.We) env_h_px(blockIdx.z, g, kn, sc, blockIdx.y, x);
}
// 256 cells of a map row per workgroup.  The cell bytes are whole numbers 0..255, so value / 255.0 comes from a table of the 256
// quotients (one IEEE division per thread instead of three per cell: the same bits); the three values of a cell go through
// LDS so that the workgroup stores its 768 values as consecutive elements.
// (r04, measured and dropped: a thread per column and 8 rows with the column's 22 horizontal sums in registers -- 0.70 ms per
// 64 KITTI frames against 0.47: a tenth of the cells is blurred at all, and the row-strided stores cost more than the shared
// window saves; and 4 rows per workgroup with one table of quotients -- 0.52 against 0.47.)
__global__ void __launch_bounds__(256) k_env_v(const PreFrame* fr, EnvGeom g, Kernels kn, PreScratch sc) {
  __shared__ double s_unit[256];
  __shared__ double s_val[768];
  __shared__ uint8_t s_byte[768];
  const int f = blockIdx.z, r = blockIdx.y, x0 = blockIdx.x * 256, tid = threadIdx.x, x = x0 + tid;
  const PreFrame F = fr[f];
  s_unit[tid] = (double)tid / 255.0;
  double bgr[3] = {0.0, 0.0, 0.0};
  if (x < g.We) env_v_bgr(f, g, kn, sc, r, x, bgr);
  __syncthreads();
  for (int c = 0; c < 3; c++) s_byte[tid * 3 + c] = (uint8_t)bgr[c];
  if (F.env_xyY) {
    double o3[3];
    env_xyY_of(s_unit[(int)bgr[2] & 255], s_unit[(int)bgr[1] & 255], s_unit[(int)bgr[0] & 255], o3);
    for (int c = 0; c < 3; c++) s_val[tid * 3 + c] = o3[c];
  }
  __syncthreads();
  const int n = 3 * min(256, g.We - x0);
  const int64_t base = ((int64_t)r * g.We + x0) * 3;
  for (int i = tid; i < n; i += 256) {
    if (F.env_u8) F.env_u8[base + i] = s_byte[i];
    if (F.env_xyY) {
      if (F.types & PRE_ENV_F32) ((float*)F.env_xyY)[base + i] = (float)s_val[i];
      else ((double*)F.env_xyY)[base + i] = s_val[i];
    }
  }
}
#endif

}  // namespace rrpre
