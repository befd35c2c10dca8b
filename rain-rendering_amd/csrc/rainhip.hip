// rainhip.hip -- gfx950 kernels and the C ABI (include/rainhip.h) of the rain-streak
// rendering + compositing hot path.  Written for MI355X only: wave64, 256 CUs, no
// portability layer.  See DESIGN.md for the kernel chain and the HBM layout.
//
// The chain of one call (grid.y = frame; DESIGN.md section 5 has the measurements):
//   second stream (RR_OPT_COLOUR_STREAM), the colour branch -- three numbers per drop:
//     k_fov_dda        a thread per drop, no LDS: FOV polygon in float with error bounds, row spans by two cursors under OpenCV's
//                      fill rule (16-byte edge records)  (k_fov_spans<NCH, false>: edge-parallel, float64 / caller-made polygons)
//     k_fov_spans<NCH, true>   the drops float cannot decide and the wrapping polygons, from the frame's list, in float64
//     k_fov_sums32 / k_fov_sums   workgroup (frame, band of map rows, chunk of drops): prefix rows in LDS, P[xr+1] - P[xl]
//     (general path, maps beyond the fast path's limits or RR_OPT_FOV_FILL_RULE: k_fov_poly_general, k_env_prefix,
//      k_env_consts, k_fov_sums_general)
//   caller's stream:
//     k_plan           one thread per drop: geometry, homography / rotation, CoC, footprint -> DropPlan, raw-tile key, list record
//     k_scan           per-frame exclusive scan of tile sizes -> arena offsets (+ the frame's zero line)
//     k_dedup          drops with equal raw-tile keys share one tile (batch-wide election)
//     k_lists, k_rows_scatter, k_rows_shares   work lists from the list records; the batch's tiles bucketed by texture, cut into shares
//     k_tile_rows      raw alpha tiles of the whole batch (round 6): a wave per tile, the bucket's texture resident in LDS -- row
//                      walks for rotate + flip + INTER_AREA, a lane per pixel for the Big drops' bicubic warpPerspective
//     k_tile_generic / k_tile_big / k_tile   what k_tile_rows does not take: rare modes; Big tiles of more than 8192 pixels, a
//                      thread per pixel; integer-ratio and oversized rotate tiles, a workgroup per tile
//     k_blur_weights, k_blur_small, k_blur_fused_dma (-DRR_EXPERIMENTS: k_blur_fused), k_blur_big_weights, k_blur<0|1>
//                      separable defocus blur of the effective tile: a wave per small tile; LDS sub-tiles staged by
//                      LDS-DMA for radii 5..48; zero-skipping taps for larger ones
//     -- join --
//     k_colour         band partials -> colour constants, compositor records, bbox, status
//     k_bin_rows (k_bin), [k_pad_visits], k_composite32 (k_composite)
//                      ordered per-tile drop lists (ballot compaction, no atomics); float64 mask in drop order, float
//                      (or float64) colours, tile sums
//     k_means, k_finalize16 (k_finalize), [k_png_image, k_png_mask, k_pngz_blocks, k_pngz_pack]
//                      mean-contrast shift, clip, truncating u8 quantisation; optional PNG scanlines / zlib streams
//   elsewhere: k_particles / k_particle_draws (drop tables born on the device), k_png_unfilter (input files' scanlines),
//   k_pad_textures, k_pair_textures, k_copy_pieces / k_copy_small (batched copies, descriptors)
// rr_prepass.h holds the fog / environment-map pre-pass kernels, rr_host.cpp the host-only helpers.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "rainhip.h"
#include "rr_device.h"
#include "rr_prepass.h"
#include "rr_deflate.h"
#include "rr_pngrows.h"
#include "rr_particles.h"

using namespace rr;

namespace {

constexpr int TILE = 16;            // screen tile edge (256 threads = 16x16 pixels)
constexpr int CTILE = 64;           // coarse binning tile (4x4 screen tiles)
constexpr int MAX_R = 416;          // >= 4*RR_MAX_SHIFT/10 + 1
constexpr int POLY_STRIDE = RR_MAX_FOV + 4;
constexpr int COL_PARTS = 8;        // row bands of the environment map; partials are added in band order (fixed: results
                                    // do not depend on the batch size)

struct FrameDesc {
  const void* bg;                  // float64, float32 or uint8 (in_types: RR_IN_BG_*)
  const void* rainy_bg;            // float64, float32 or uint8 (RR_IN_RAINY_*; the BG flags when rainy_bg == bg)
  const void* env;                 // float64 or float32 (RR_IN_ENV_F32)
  const void* omega;               // as env
  const rr_drop* drops;
  uint8_t* rgb;
  double* comp_out;                // H*W*3 composite before the mean shift (user buffer or ctx scratch); float[] when comp_f32
  double* mask_f64;
  int32_t* mask_i32;
  int32_t* status;
  uint8_t* png_image;              // optional: PNG scanlines (Sub filter) of the RGBA rainy image, H * (1 + 4 W) bytes
  uint8_t* png_mask;               // optional: same for the colour-mapped rain mask
  const void* depth;               // optional (RR_OPT_DEPTH_OCCLUSION): scene depth in metres, H*W float32 / float64
  const rr_ext_tile* ext;          // optional: caller-made tiles / FOV polygons per drop (device pointers inside)
  double* colour_out;              // optional: n_drops * 3 colour constants (rr_frame_out.drop_colour)
  const int32_t* n_drops_dev;      // optional: the drop count lives on the device (k_patch_counts)
  int32_t comp_f32, in_types;      // comp_out holds floats (the float-colour compositor wrote it); element types of the inputs
  int32_t depth_f64;
  int32_t n_drops;
  int32_t strategy;
  double opacity;
};

// Pointers that arrive inside FrameDesc are loaded from memory, so the compiler has to treat them as generic
// ("flat": every access also counts against the LDS counter).  They are all device-global: say so.
template <class T>
using global_ptr = T __attribute__((address_space(1)))*;
template <class T>
__device__ inline global_ptr<T> as_global(T* p) { return (global_ptr<T>)p; }
// Data a kernel only reads and an EARLIER kernel wrote: the constant address space lets a wave-uniform access use the
// scalar unit (s_load into SGPRs) even after barriers, which otherwise count as possible writers.
template <class T>
using const_ptr = const T __attribute__((address_space(4)))*;
template <class T>
__device__ inline const_ptr<T> as_constant(const T* p) { return (const_ptr<T>)p; }
__device__ inline rr_drop load_drop(const rr_drop* p) {       // 14 global 8-byte loads
  static_assert(sizeof(rr_drop) % 8 == 0, "rr_drop layout");
  rr_drop d;
  const global_ptr<const uint64_t> src = as_global(reinterpret_cast<const uint64_t*>(p));
  uint64_t* dst = reinterpret_cast<uint64_t*>(&d);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(rr_drop) / 8); k++) dst[k] = src[k];
  return d;
}

// three channels of pixel `pix` of an image input as float64: kind 0 float64, 1 float32, 2 uint8 (value / 255.0, the
// reference's bg = cv2.imread(...) / 255.0, generator.py:352)
__device__ inline void load_px3(const void* base, int kind, int64_t pix, double out[3]) {
  if (kind == 0) {
    const global_ptr<const double> s = as_global(static_cast<const double*>(base)) + pix * 3;
    out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
  } else if (kind == 1) {
    const global_ptr<const float> s = as_global(static_cast<const float*>(base)) + pix * 3;
    out[0] = (double)s[0]; out[1] = (double)s[1]; out[2] = (double)s[2];
  } else {
    const global_ptr<const uint8_t> s = as_global(static_cast<const uint8_t*>(base)) + pix * 3;
    out[0] = (double)s[0] / 255.0; out[1] = (double)s[1] / 255.0; out[2] = (double)s[2] / 255.0;
  }
}
// The same in two steps -- the loads now, the conversion when the values are needed -- for two pixels of one image: a kernel
// that asks for all its pixels first and converts afterwards waits for one memory round trip instead of one per load_px3.
struct RawPx {
  uint32_t a, b, c, d, e, f;         // float64: three (lo, hi) pairs; float32: a b c; uint8: a b c
};
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ inline void load_px3_raw2(const void* base, int kind, int64_t pa, int64_t pb, RawPx& qa, RawPx& qb) {
  if (kind == 0) {
    const global_ptr<const u32x2_t> s = as_global(static_cast<const u32x2_t*>(base));
    const u32x2_t a0 = s[pa * 3], a1 = s[pa * 3 + 1], a2 = s[pa * 3 + 2], b0 = s[pb * 3], b1 = s[pb * 3 + 1], b2 = s[pb * 3 + 2];
    qa = RawPx{a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
    qb = RawPx{b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
  } else if (kind == 1) {
    const global_ptr<const uint32_t> s = as_global(static_cast<const uint32_t*>(base));
    const uint32_t a0 = s[pa * 3], a1 = s[pa * 3 + 1], a2 = s[pa * 3 + 2], b0 = s[pb * 3], b1 = s[pb * 3 + 1], b2 = s[pb * 3 + 2];
    qa = RawPx{a0, a1, a2, 0u, 0u, 0u};
    qb = RawPx{b0, b1, b2, 0u, 0u, 0u};
  } else {
    const global_ptr<const uint8_t> s = as_global(static_cast<const uint8_t*>(base));
    const uint32_t a0 = s[pa * 3], a1 = s[pa * 3 + 1], a2 = s[pa * 3 + 2], b0 = s[pb * 3], b1 = s[pb * 3 + 1], b2 = s[pb * 3 + 2];
    qa = RawPx{a0, a1, a2, 0u, 0u, 0u};
    qb = RawPx{b0, b1, b2, 0u, 0u, 0u};
  }
}
__device__ inline void decode_px3(const RawPx& q, int kind, double out[3]) {
  if (kind == 0) {
    out[0] = __hiloint2double((int)q.b, (int)q.a);
    out[1] = __hiloint2double((int)q.d, (int)q.c);
    out[2] = __hiloint2double((int)q.f, (int)q.e);
  } else if (kind == 1) {
    out[0] = (double)__uint_as_float(q.a);
    out[1] = (double)__uint_as_float(q.b);
    out[2] = (double)__uint_as_float(q.c);
  } else {
    out[0] = (double)q.a / 255.0;
    out[1] = (double)q.b / 255.0;
    out[2] = (double)q.c / 255.0;
  }
}
__device__ inline int bg_kind(const FrameDesc& fr) { return (fr.in_types & RR_IN_BG_U8) ? 2 : ((fr.in_types & RR_IN_BG_F32) ? 1 : 0); }
__device__ inline int rainy_kind(const FrameDesc& fr) {
  if (fr.rainy_bg == fr.bg) return bg_kind(fr);
  return (fr.in_types & RR_IN_RAINY_U8) ? 2 : ((fr.in_types & RR_IN_RAINY_F32) ? 1 : 0);
}
// float loads that may straddle an 8-byte boundary (rows of an odd-width map start on odd elements)
typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

__device__ inline void wave_lds_sync() {
  // wave-private LDS hand-off: LDS operations of one wave execute in issue order; this only
  // stops the compiler from moving accesses across the hand-off point.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Phase clocks (builds with -DRR_PHASES only: scripts/phase_timing.sh): shader cycles a wave spends between the PH()
// marks of a kernel, summed over all waves -- where the latency-bound tile / blur kernels spend their time.  The product
// build carries none of this.
#ifdef RR_PHASES
__device__ unsigned long long g_phase[8][8];
#define PH_DECL unsigned long long ph_t0_ = __builtin_readcyclecounter(), ph_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define PH(k)                                                   \
  {                                                             \
    const unsigned long long ph_n_ = __builtin_readcyclecounter(); \
    ph_acc_[k] += ph_n_ - ph_t0_;                               \
    ph_t0_ = ph_n_;                                             \
  }
#define PH_FLUSH(kid)                                           \
  if ((threadIdx.x & 63) == 0) {                                \
    for (int ph_q_ = 0; ph_q_ < 8; ph_q_++)                     \
      if (ph_acc_[ph_q_]) atomicAdd(&g_phase[kid][ph_q_], ph_acc_[ph_q_]); \
  }
#define PH_PARAMS , unsigned long long& ph_t0_, unsigned long long (&ph_acc_)[8]
#define PH_PASS , ph_t0_, ph_acc_
#define PH_COUNT(k) ph_acc_[k] += 1;
#else
#define PH_PARAMS
#define PH_PASS
#define PH_COUNT(k)
#define PH_DECL
#define PH_WAITVM
#define PH(k)
#define PH_FLUSH(kid)
#endif

struct Scratch {                    // per-batch device scratch, all indexed [frame][...]
  DropPlan* plan;
  CompRec* comp;
  CompRec32* comp32;
  int32_t* poly;                    // [frame][drop][2][POLY_STRIDE]
  int32_t* npts;
  int64_t* sizes;
  double* prefix;                   // [frame][He][We+1][4]: general path only (maps beyond HE_MAX / FOV_WE_MAX)
  double* fband;                    // [frame][COL_PARTS][2] = sum w, sum Y*w of every row band
  double* arena;                    // [frame][arena_cap]
  double* partial;                  // [frame][ntiles][4] per screen tile: sum(composite), sum(bg), min(mask), max(mask)
  double* means;                    // [frame][4] = mean(composite), mean(bg), min(mask), max(mask)
  int64_t* arena_need;              // [frame]
  int32_t* overflow;                // the batch's own arena-overflow flag (ctx: [1 + RR_PIPE_SLOTS], one per pipeline slot + the device-pointer calls)
  unsigned long long* need_max;     // [1] largest per-frame arena need of every batch since the arena was last (re)sized
  int32_t* list_rot;                // [frames * drops] batch-global indices of the drops k_tile renders, a piece per frame (r06)
  int32_t* rot_total;               // [1] their number (behind rows_next: the same memset)
  int32_t* list_gen;                // [frame][drops]  drops taken by k_tile_generic
  int32_t* list_slow;               // [frame][drops]  blurred drops the fused kernel cannot take
  int4* blur_items;                 // [frame][8*drops] (drop, first sub-tile, #sub-tiles, -)
  int32_t* list_small;              // [frame][drops]  blurred drops handled one wave each (k_blur_small)
  double* colpart;                  // [frame][COL_PARTS][5][drops] FOV partial sums per envmap row band
  double* wtab;                     // [frame][drops][2][BR_MAX+1] normalised Gaussian half tables of the blurred drops (k_blur_weights)
  double* wtab_big;                 // [frame][SLOW_CAP][2][MAX_R+1] the same for the first SLOW_CAP large-radius drops of a frame (k_blur_big_weights)
  const uint8_t* tex_pad;           // the textures with their 2-texel zero border, as k_tile stages them (k_pad_textures); NULL: staged byte by byte
  const int64_t* tex_poff;          // [texture] offset of its padded copy (a multiple of 16)
  uint4* fov_erec;                  // [frame][n_fov][drops] k_fov_dda's edge records (rr_device.h dda_edge_record: 16 bytes)
  uint32_t* fov_pix;                // [frame][n_fov][drops] k_fov_dda's vertex pixels, x | y << 16 (a wave stores 256 contiguous bytes per vertex)
  int32_t* fov_list;                // [frame][drops] drops k_fov_dda leaves to k_fov_spans (wrapping polygons, float64 decisions)
  int32_t* fov_list_n;              // [frame] their number
  uint8_t* blended;                 // [frame][drop] 1: the drop is composited (k_colour)
  int32_t* pad_first;               // RR_OPT_WILD_PIXELS only (else null), [frame][H*W]: lowest index of a composited drop whose padded
  int32_t* eff_first;               //   rectangle covers the pixel OUTSIDE / INSIDE the tile the compositor blends (k_pad_visits)
  uint32_t* spans;                  // [frame][Hp][Dp] FOV row spans xl | (xr+1) << 16, 0 = empty row; Hp = He rounded up to 4, Dp = drops + 1
                                    // rounded up to 8; column `drops` of every row stays all zeros: what a drop without a polygon
                                    // reads.  (r05: a row at a time -- k_fov_sums32 fetches the NEXT row's spans while it works on
                                    // this one; the [row quad][drop][4] layout of r04 cost it a memory round trip per four rows.)
  int4* bbox;                       // [frame][drops] footprint (x0,y0,x1,y1), empty when not composited
  uint16_t* clist;                  // [frame][coarse tiles][drops] ordered drop indices per 64x64 coarse tile
  int32_t* ccount;                  // [frame][coarse tiles]
  int32_t* counts;                  // [frame][8] = #rot, #gen, #blur items, #slow, #small, -, -, #duplicate raw tiles
  int32_t* list_big;                // [frame][drops] Big drops (bicubic warp) rendered by k_tile_big, one thread per pixel
  int32_t* big_off;                 // [frame][drops+1] exclusive prefix of their tile sizes, in pixels
  int32_t* bigs_list;               // [frame][drops] the frame's Big drops (k_plan) for k_plan_big: the 8 x 8 solve of their homography
  int32_t* bigs_n;                  // [frame] their number (zeroed per call)
  uint4* tkey;                      // [frame][drops][2] raw_tile_key (rr_device.h): what k_dedup compares (r06: 32 bytes instead of the plan's ~200)
  uint4* lrec;                      // [frame][drops] ListRec: the drop's work-list classes, made once by k_plan (r06; k_lists read the plans twice)
  int32_t* canon;                   // [frame][drops] batch-global index of the drop whose raw tile this drop uses
  int32_t* htab;                    // [2*frames*drops] open-addressing table of k_dedup (0 = empty, else index+1)
  int32_t blur_bx, blur_by;         // LDS capacities (doubles) of the fused blur's two staging tiles (RR_OPT_BLUR_WORKGROUPS)
  // k_tile_rows (r06): the rotate + INTER_AREA tiles of the WHOLE batch in one list, bucketed by texture
  int32_t* rows_list;               // [frame][drops] frame-local indices of the drops k_tile_rows renders (k_lists)
  int32_t* rows_n;                  // [frame][2] their number: rotate + INTER_AREA tiles, Big tiles
  int32_t* rows_hist;               // [RW_TEX_MAX] tiles per texture, batch-wide (zeroed per call)
  unsigned long long* rows_cost;    // [RW_TEX_MAX] their estimated cost (behind rows_hist: one memset)
  int32_t* rows_next;               // [1] the next share of the list (behind rows_cost: the same memset)
  int32_t* rows_bounds;             // [RW_SHARE_MAX + 1] first tile of every share (k_rows_shares)
  int32_t* rows_fbase;              // [frame][RW_TEX_MAX] first slot of the frame inside its textures' buckets
  int32_t* rows_sorted;             // [frames * drops] batch-global drop indices, bucket after bucket (k_rows_scatter)
  const uint8_t* tex_pair;          // pair textures (k_pair_textures) and their offsets (multiples of 16)
  const int64_t* tex_qoff;
  int32_t colpart_f32;              // the band partials lie in colpart as floats (k_fov_sums32's own width: half the bytes for it and k_colour)
  int32_t rows_on, n_tex;           // RR_OPT_TILE_ROWS and a database of at most RW_TEX_MAX textures
  int32_t big_on, n_buckets;        // Big (bicubic) tiles ride in the same list, buckets n_tex .. 2 n_tex - 1 (2 n_tex <= RW_TEX_MAX)
};

// ---------------------------------------------------------------------------
// environment map prefix sums
// ---------------------------------------------------------------------------
// One wave per environment-map row, 64 consecutive texels per step: coalesced loads
// (24 B + 8 B per lane), a Hillis-Steele scan across the wave with shuffles, the running
// carry in a register, one coalesced 32-byte store per lane.
__global__ __launch_bounds__(256) void k_env_prefix(const FrameDesc* frames, Dims dm, double* prefix) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), f = blockIdx.y;
  if (row >= dm.He) return;
  const FrameDesc& fr = frames[f];
  const int We = dm.We;
  const bool e32 = (fr.in_types & RR_IN_ENV_F32) != 0;
  const global_ptr<const double> env = as_global(static_cast<const double*>(fr.env)) + (int64_t)row * We * 3;
  const global_ptr<const double> om = as_global(static_cast<const double*>(fr.omega)) + (int64_t)row * We;
  const global_ptr<const float> envf = as_global(static_cast<const float*>(fr.env)) + (int64_t)row * We * 3;
  const global_ptr<const float> omf = as_global(static_cast<const float*>(fr.omega)) + (int64_t)row * We;
  double* P = prefix + ((int64_t)f * dm.He + row) * (int64_t)(We + 1) * 4;
  if (lane == 0) { P[0] = 0.0; P[1] = 0.0; P[2] = 0.0; P[3] = 0.0; }
  double carry[4] = {0, 0, 0, 0};
  for (int c0 = 0; c0 < We; c0 += 64) {
    const int c = c0 + lane;
    double v[4] = {0, 0, 0, 0};
    if (c < We) {
      const double w = e32 ? (double)omf[c] : om[c];
      v[0] = (e32 ? (double)envf[c * 3 + 0] : env[c * 3 + 0]) * w;
      v[1] = (e32 ? (double)envf[c * 3 + 1] : env[c * 3 + 1]) * w;
      v[2] = (e32 ? (double)envf[c * 3 + 2] : env[c * 3 + 2]) * w;
      v[3] = w;
    }
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double u = __shfl_up(v[k], ofs);
        if (lane >= ofs) v[k] += u;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] += carry[k];
    if (c < We) {
      double* o = P + (int64_t)(c + 1) * 4;
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) carry[k] = __shfl(v[k], 63);
  }
}

__global__ __launch_bounds__(256) void k_env_consts(Dims dm, const double* prefix, double* fband) {
  const int f = blockIdx.x, t = threadIdx.x;
  double sY = 0, sW = 0;
  for (int r = t; r < dm.He; r += 256) {
    const double* last = prefix + (((int64_t)f * dm.He + r) * (int64_t)(dm.We + 1) + dm.We) * 4;
    sY += last[2];
    sW += last[3];
  }
  __shared__ double a[256], b[256];
  a[t] = sY;
  b[t] = sW;
  __syncthreads();
  for (int ofs = 128; ofs > 0; ofs >>= 1) {
    if (t < ofs) {
      a[t] += a[t + ofs];
      b[t] += b[t + ofs];
    }
    __syncthreads();
  }
  if (t < COL_PARTS) {                 // same layout as k_fov_sums' row-band totals: band 0 carries everything
    fband[(f * COL_PARTS + t) * 2 + 0] = t == 0 ? b[0] : 0.0;
    fband[(f * COL_PARTS + t) * 2 + 1] = t == 0 ? a[0] : 0.0;
  }
}

// blur work layout: rr_device.h (blur_layout, blur_is_small -- host-compiled too, so that the CPU tier can sweep them)
// blurred drops neither k_blur_small nor k_blur_fused can take (radius > BR_MAX): two global passes
// through one extra padded scratch tile
__device__ inline bool blur_is_slow(const DropPlan& p, const Scratch& sc) {
  return p.r1 > 0 && !blur_is_small(p) && !blur_layout(p, sc.blur_bx, sc.blur_by).fused;
}

// ---------------------------------------------------------------------------
// per-drop plan
// ---------------------------------------------------------------------------
// One thread per drop.  The 312-byte plans leave through LDS: a thread storing its own record would touch 78 different
// lines with 4-byte pieces (the L2 then writes partial lines back: 5x the payload); staged 32 records at a time per
// wave, the stores are whole, consecutive lines.
constexpr int PLAN_DW = (int)(sizeof(DropPlan) / 4);
// A drop's work-list classes, 16 bytes (r06): k_lists walks the frame's drops twice (count, then fill) and used to derive
// these from the 312-byte plan both times.
//   x: tile class (bits 0-3) | blur class (bits 4-6) | texture << 8      y: tile cost (LC_ROWS, LC_BIG_LDS) / pixels (LC_BIG)
//   z: fused blur layout wo | ho << 16                                   w: its number of sub-tiles
enum { LC_NONE = 0, LC_BIG_LDS = 1, LC_BIG = 2, LC_ROWS = 3, LC_ROT_INT = 4, LC_ROT = 5, LC_GEN = 6 };
enum { LB_NONE = 0, LB_SMALL = 1, LB_FUSED = 2, LB_SLOW = 3 };
__device__ inline uint4 make_list_rec(const DropPlan& p, const int32_t* tex_h, const int32_t* tex_w, const Scratch& sc);
__global__ __launch_bounds__(128) void k_plan(const FrameDesc* frames, Dims dm, rr_camera cam, const int32_t* tex_h,
                                              const int32_t* tex_w, int max_drops, int use_npts, Scratch sc) {
  static_assert(sizeof(DropPlan) % 4 == 0, "DropPlan layout");
  const int f = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const FrameDesc& fr = frames[f];
  __shared__ uint32_t s_plan[2][32 * PLAN_DW];
  const bool act = i < fr.n_drops;
  const int64_t gi = (int64_t)f * max_drops + (act ? i : 0);
  DropPlan p;
  if (act) {
    rr_drop d = load_drop(fr.drops + i);
    int64_t size = 0;
    ExtGeom xg{false, 0, 0, 0, 0};
    if (fr.ext) {
      const rr_ext_tile& e = fr.ext[i];
      if (e.alpha) xg = ExtGeom{true, e.tw, e.th, e.min_x, e.min_y};
    }
    plan_drop<true>(d, cam, dm, tex_h, tex_w, fr.opacity, fr.strategy, p, size, xg);      // (a Big drop's homography: k_plan_big)
    // the FOV polygon is evaluated for every drop: in the reference its failure is raised before the circle of confusion
    // is looked at (bad_weather.py:363-373 vs :416) -- k_colour gives such a drop its status and keeps it out of the blend.
    // use_npts (the colour branch ran BEFORE this kernel, on the same stream): a drop without a polygon gets no tile either;
    // with the colour branch on its own stream (RR_OPT_COLOUR_STREAM) the polygon is not known yet and the rare tile is
    // rendered for nothing.
    const int npts = use_npts ? sc.npts[gi] : 1;  // 0: failed; -1: 'white' strategy (never used)
    if (p.status != RR_DROP_OK || npts == 0) size = 0;
    if (size > 0 && blur_is_slow(p, sc)) size += ((int64_t)p.ew * p.eh + 15) & ~15LL;     // dense scratch tile of the two-pass blur
    sc.sizes[gi] = size;
    uint32_t key[8];
    raw_tile_key(d, p, key);
    if (p.kind == KIND_EXT) key[0] = 0xffffffffu;
    sc.tkey[2 * gi] = make_uint4(key[0], key[1], key[2], key[3]);
    sc.tkey[2 * gi + 1] = make_uint4(key[4], key[5], key[6], key[7]);
    sc.lrec[gi] = make_list_rec(p, tex_h, tex_w, sc);
  }
  {                                                          // the wave's Big drops whose tile will be rendered -> the frame's list
    const bool big = act && p.status == RR_DROP_OK && p.kind == KIND_BIG && sc.sizes[gi] > 0;
    const unsigned long long m = __ballot(big);
    if (m != 0ull) {
      int base = 0;
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&sc.bigs_n[f], __popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1);
      if (big) sc.bigs_list[(int64_t)f * max_drops + base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    }
  }
  const int wave_i0 = blockIdx.x * blockDim.x + wave * 64;         // first drop of this wave
  uint32_t* stage = s_plan[wave];
  uint32_t* out = reinterpret_cast<uint32_t*>(sc.plan + (int64_t)f * max_drops + wave_i0);
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if (act && (lane >> 5) == half) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&p);
#pragma unroll
      for (int k = 0; k < PLAN_DW; k++) stage[(lane & 31) * PLAN_DW + k] = src[k];
    }
    wave_lds_sync();
    const int first = wave_i0 + half * 32;
    const int nrec = imax(imin(32, fr.n_drops - first), 0);
    for (int k = lane; k < nrec * PLAN_DW; k += 64) out[half * 32 * PLAN_DW + k] = stage[k];
    wave_lds_sync();
  }
}

// The Big drops' inverse homographies (rr_device.h plan_big_homography: the 8 x 8 solve in registers -- 170 of them and some
// scratch), for the frame's list of Big drops only: inside k_plan it held EVERY wave of that latency-bound kernel to two per
// SIMD for the sixth of the drops that need it.
__global__ __launch_bounds__(128) void k_plan_big(const FrameDesc* frames, const int32_t* tex_h, const int32_t* tex_w, int max_drops, Scratch sc) {
  const int f = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= sc.bigs_n[f]) return;
  const int i = sc.bigs_list[(int64_t)f * max_drops + j];
  const rr_drop d = load_drop(frames[f].drops + i);
  double mi[9];
  plan_big_homography(d, tex_w[d.tex_index], tex_h[d.tex_index], mi);
  double* o = sc.plan[(int64_t)f * max_drops + i].mi;
#pragma unroll
  for (int k = 0; k < 9; k++) o[k] = mi[k];
}

// one block per frame: exclusive scan of arena sizes
__global__ __launch_bounds__(1024) void k_scan(const FrameDesc* frames, int max_drops, int64_t arena_cap, Scratch sc) {
  const int f = blockIdx.x, t = threadIdx.x;
  const int n = frames[f].n_drops;
  const int chunk = (n + 1023) / 1024;
  const int i0 = t * chunk, i1 = min(i0 + chunk, n);
  const int64_t base = (int64_t)f * max_drops;
  int64_t s = 0;
  for (int i = i0; i < i1; i++) s += sc.sizes[base + i];
  __shared__ int64_t sh[1024];
  sh[t] = s;
  __syncthreads();
  for (int ofs = 1; ofs < 1024; ofs <<= 1) {
    int64_t v = (t >= ofs) ? sh[t - ofs] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  // the first line (16 doubles) of a frame's arena stays all zeros: where the float compositor's lanes outside a footprint
  // read their "sample" from
  int64_t run = ((t == 0) ? 0 : sh[t - 1]) + 16;
  const int64_t frame_base = (int64_t)f * arena_cap;
  if (t < 16 && arena_cap >= 16) sc.arena[frame_base + t] = 0.0;
  for (int i = i0; i < i1; i++) {
    DropPlan& p = sc.plan[base + i];
    int64_t sz = sc.sizes[base + i];
    if (run + sz > arena_cap) {
      // does not fit: never touch the arena for this drop; host regrows and re-runs
      sc.sizes[base + i] = 0;
    } else {
      p.a0_off = frame_base + run;
      p.a1_off = frame_base + run + (((int64_t)p.tw * p.th + 15) & ~15LL);
    }
    run += sz;
  }
  if (t == 1023) {
    const int64_t need = sh[1023] + 16;
    sc.arena_need[f] = need;
    if (need > arena_cap) {
      atomicExch(sc.overflow, 1);
      atomicMax(sc.need_max, (unsigned long long)need);           // sticky until the host regrows the arena
    }
  }
}

// ---------------------------------------------------------------------------
// raw-tile de-duplication
// ---------------------------------------------------------------------------
// The raw alpha tile of a drop (before the defocus blur) is a pure function of the texture, the
// flip, the tile size and the rotation / homography -- nothing of the drop's position, depth or
// colour enters.  Streak end points are integer pixels and there are 50 textures, so within a
// batch most drops share their raw tile with another drop (typically 5 of 6 at 16 frames per
// batch).  One thread per drop: drops with bit-identical tile parameters elect one of them
// through an open-addressing table (atomicCAS); the others take over its arena offset and drop
// out of the tile work lists.  Which drop wins the election is irrelevant: every candidate
// would write the same bits.
// r06: what is compared is the 32-byte raw_tile_key (rr_device.h) k_plan leaves per drop -- the inputs the tile is a pure
// function of -- not the ~200 bytes of plan fields derived from them (the kernel read every 312-byte plan through LDS and
// a second one per probe: 2.2 GB per 512 frames).
__device__ inline uint32_t raw_tile_hash(const uint4& a, const uint4& b) {
  uint32_t h = 2166136261u;
  auto mix = [&](uint32_t v) { h = (h ^ v) * 16777619u; };
  mix(a.x); mix(a.y); mix(a.z); mix(a.w); mix(b.x); mix(b.y); mix(b.z); mix(b.w);
  h ^= h >> 15;
  return h;
}
__global__ __launch_bounds__(256) void k_dedup(const FrameDesc* frames, int max_drops, int n_frames, int enable, Scratch sc) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + (int)threadIdx.x;
  const int n = frames[f].n_drops;
  if (i >= n) return;
  const int gi = f * max_drops + i;
  const uint4 ka = sc.tkey[2 * (int64_t)gi], kb = sc.tkey[2 * (int64_t)gi + 1];
  int canon = gi;
  if (enable && ka.x != 0xffffffffu && sc.sizes[gi] != 0) {
    const uint32_t cap = 2u * (uint32_t)n_frames * (uint32_t)max_drops;
    uint32_t h = raw_tile_hash(ka, kb) % cap;
    for (;;) {
      const int prev = atomicCAS(&sc.htab[h], 0, gi + 1);
      if (prev == 0) break;                               // this drop renders the tile
      const uint4 pa = sc.tkey[2 * (int64_t)(prev - 1)], pb = sc.tkey[2 * (int64_t)(prev - 1) + 1];
      if (pa.x == ka.x && pa.y == ka.y && pa.z == ka.z && pa.w == ka.w && pb.x == kb.x && pb.y == kb.y && pb.z == kb.z && pb.w == kb.w) {
        canon = prev - 1;
        break;
      }
      h = h + 1 == cap ? 0 : h + 1;
    }
    if (canon != gi) sc.plan[gi].a0_off = sc.plan[canon].a0_off;    // an elected drop never changes its own offset
  }
  sc.canon[gi] = canon;
  // the frame's duplicate counter (a diagnostic): one atomic per wave -- one per drop is a chain of thousands of
  // serialised read-modify-writes of the same address, which was most of this kernel's time
  const unsigned long long dup = __ballot(canon != gi);
  if (dup != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)dup) - 1) atomicAdd(&sc.counts[f * 8 + 7], __popcll(dup));
}

// ---------------------------------------------------------------------------
// colour: FOV polygon -> row spans -> sums over the environment map
// ---------------------------------------------------------------------------
// The reference reduces the whole environment map under a polygon mask for every drop
// (bad_weather.py:383-409).  Here the mask of a drop is its per-row span [xl, xr] (k_fov_spans) and the
// masked sums are differences of row prefix sums (k_fov_sums).  The prefix sums of a row only ever live
// in LDS: every environment-map row is read from HBM once per batch.
constexpr int HE_MAX = 1024;        // tallest map of the fast path: 16 chunks of 64 rows in registers
constexpr int FOV_WE_MAX = 4096;    // widest map of the fast path: (We + 1) * 32 B of LDS, <= 4 columns per thread
constexpr int FOV_GROUPS = 3;       // drops per wave in k_fov_spans (n_fov = 20: 60 of 64 lanes busy)

// fov_rowspan (rr_device.h) with the rounded edge intersection evaluated in double: exact,
// because |2*num+den| < 2^26 and a non-integer quotient is at least 1/(2*den) away from an integer.
__device__ inline bool fov_rowspan_fast(const int32_t* px, const int32_t* py, int n, int y, int We, int& xl, int& xr) {
  int lo = 1 << 30, hi = -(1 << 30);
  for (int i = 0; i < n; i++) {
    const int j = (i + 1 == n) ? 0 : i + 1;
    const int x0 = px[i], y0 = py[i], x1 = px[j], y1 = py[j];
    const int ylo = imin(y0, y1), yhi = imax(y0, y1);
    if (y < ylo || y > yhi) continue;
    if (y0 == y1) {
      lo = imin(lo, imin(x0, x1));
      hi = imax(hi, imax(x0, x1));
    } else {
      const bool sw = y1 < y0;
      const int xa = sw ? x1 : x0, yA = sw ? y1 : y0, xb = sw ? x0 : x1, yB = sw ? y0 : y1;
      const int den = yB - yA;
      const int nn = 2 * (xb - xa) * (y - yA) + den;
      const int xv = xa + (int)floor((double)nn / (double)(2 * den));
      lo = imin(lo, xv);
      hi = imax(hi, xv);
    }
  }
  xl = imax(lo, 0);
  xr = imin(hi, We - 1);
  return xl <= xr;
}

// FOV polygon and its row spans.  One wave handles FOV_GROUPS drops:
//   1. every (drop, vertex) pair has a lane: spin direction, sphere intersection, lat-long pixel (fov_vertex); the
//      wrap test of the reference (bad_weather.py:669-695) is a ballot inside the 20-lane group, the four border
//      vertices of a wrapping polygon are inserted by the lane in front of the gap.  The polygon goes to
//      wave-private LDS, never to global memory.
//   2. per drop, edges two at a time (one per half wave), lanes along the rows an edge covers: the x of edge e at row y is
//      xa + floor((2*dx*(y-ya) + den) / (2*den)) (fov_rowspan), evaluated with a float reciprocal and an exact
//      integer fix-up (|2*dx*dy| < 2^23 is checked by the host), folded into the row's [min, max] with LDS
//      ds_min / ds_max (order-free, no return value).  The spans leave as one u32 per row, xl | (xr + 1) << 16
//      (0 = empty), in the [row quad][drop] layout k_fov_sums reads coalesced.
template <int NCH, bool from_list>
__global__ __launch_bounds__(256) void k_fov_spans(const FrameDesc* frames, Dims dm, rr_camera cam, int max_drops, int Hp, int Dp, int use32, int cv_rule, Scratch sc) {
  const int f = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const FrameDesc& fr = frames[f];
  const int N = cam.n_fov, G = imin(64 / N, FOV_GROUPS);
  __shared__ double s_phi[2][RR_MAX_FOV];
  __shared__ float s_phi32[2][RR_MAX_FOV];
  __shared__ int s_px[4][FOV_GROUPS][POLY_STRIDE], s_py[4][FOV_GROUPS][POLY_STRIDE];
  __shared__ int s_xl[4][NCH * 64], s_xr[4][NCH * 64];         // per wave: the row spans of the drop being converted
  __shared__ int4 s_edge[4][3 * POLY_STRIDE];                  // per wave: the edges of that drop
  if (threadIdx.x < RR_MAX_FOV) {
    s_phi[0][threadIdx.x] = cam.phi_cos[threadIdx.x];
    s_phi[1][threadIdx.x] = cam.phi_sin[threadIdx.x];
    s_phi32[0][threadIdx.x] = (float)cam.phi_cos[threadIdx.x];
    s_phi32[1][threadIdx.x] = (float)cam.phi_sin[threadIdx.x];
  }
  __syncthreads();
  // from_list: the wave's drops are entries of the frame's list of drops k_fov_dda left to this kernel, not neighbours
  const int n_eff = from_list ? sc.fov_list_n[f] : fr.n_drops;
  for (int wslot = blockIdx.x * 4 + wave; wslot * G < n_eff; wslot += gridDim.x * 4) {      // (a loop in list mode only: the grid covers every drop otherwise)
  const int s0 = wslot * G;                                    // first drop slot of this wave
  const int g = lane / N, k = lane - g * N;                    // (drop slot, vertex)
  const bool act = g < G && s0 + g < n_eff;
  const int dg = act ? (from_list ? sc.fov_list[(int64_t)f * max_drops + s0 + g] : s0 + g) : 0;       // the drop of this lane's group
  const int64_t gi = (int64_t)f * max_drops + dg;
  if (fr.strategy == 1) {                                      // 'white': the FOV is computed by the reference but never used
    if (act && k == 0) sc.npts[gi] = -1;
    return;
  }
  const bool ext = act && fr.ext && fr.ext[dg].alpha != nullptr;      // the caller's polygon (rr_ext_tile)
  const int base = g * N, nxt_lane = base + (k + 1 == N ? 0 : k + 1);
  const unsigned long long gmask = act ? (((N >= 64 ? 0ull : (1ull << N)) - 1ull) << base) : 0ull;
  // the vertex of this lane: its pixel on the map (truncated like pyclipper's integer cast), the wrap test of the side
  // that starts here, whether its coordinates are finite; ok: fov_setup's verdict for the drop
  int ipx = 0, ipy = 0;
  bool cnd = false, vbad = true, ok = false;
  bool need64 = act && !ext && !use32;
  if (use32) {
    // float vertices (rr_device.h, "the same polygon in float32"): every predicate that decides the drop's status or the
    // wrap structure carries an error bound; a drop that comes close is evaluated again in float64 below
    float azf = 0.f, erf = 0.f, pxf = 0.f, pyf = 0.f;
    int uns = 0;
    if (act && !ext) {
      FovSetup32 F;
      const rr_drop d = load_drop(fr.drops + dg);
      fov_setup32(d, (float)cam.fov_cos, (float)cam.fov_sin, F, uns);
      fov_vertex32(F, (float)cam.radius, s_phi32[0][k], s_phi32[1][k], dm.He, dm.We, azf, erf, pxf, pyf, uns);
    }
    const float az_n = __shfl(azf, act ? nxt_lane : lane), er_n = __shfl(erf, act ? nxt_lane : lane);
    bool c32 = false;
    if (act && !ext) c32 = fov_wrap_cnd32(azf, az_n, erf, er_n, uns);
    unsigned gor = 0;                                          // the reason bits of all vertices of the drop
#pragma unroll
    for (int b = 0; b < 8; b++)
      if (__ballot((uns >> b) & 1) & gmask) gor |= 1u << b;
    const bool certain_fail = (gor & 128u) && !(gor & 3u);     // a vertex without intersection: [] like the reference
    const unsigned long long bt32 = __ballot(act && !ext && c32) & gmask, bf32 = __ballot(act && !ext && !c32) & gmask;
    const bool wrap32 = __popcll(bt32) == 1 || __popcll(bf32) == 1;
    const int r = imin(imax((int)pyf, 0), dm.He - 1);
    const int r0 = __shfl(r, act ? base : lane);
    const bool spread = (__ballot(iabs(r - r0) >= 2) & gmask) != 0ull;       // some vertex two rows away from the first one
    // (a sliver: whether Clipper takes the path -- poly_all_collinear below -- is float64's to say)
    const int tx = (int)pxf, ty = (int)pyf;
    const int x0v = __shfl(tx, act ? base : lane), y0v = __shfl(ty, act ? base : lane);
    const int xav = __shfl(tx, act ? base + N / 2 : lane), yav = __shfl(ty, act ? base + N / 2 : lane);
    const bool sure_nc = (__ballot(poly_surely_not_collinear_step(xav - x0v, yav - y0v, tx - x0v, ty - y0v)) & gmask) != 0ull;
    need64 = act && !ext && !certain_fail && ((gor & ~128u) != 0u || !bt32 || !bf32 || (!wrap32 && !spread) || !sure_nc);
    if (act && !ext && !need64) {
      ipx = (int)pxf;
      ipy = (int)pyf;
      cnd = c32;
      vbad = certain_fail;
      ok = true;
    }
  }
  if (__ballot(need64) != 0ull) {                              // (wave-uniform) float64: every drop, or the ones float cannot decide
    double az = 0.0, ptx = 0.0, pty = 0.0;
    bool ok64 = false;
    if (need64) {
      FovSetup F;
      const rr_drop d = load_drop(fr.drops + dg);
      ok64 = fov_setup(d, cam, F);
      fov_vertex(F, cam, s_phi[0][k], s_phi[1][k], dm.He, dm.We, az, ptx, pty);
    }
    const double az_next = __shfl(az, need64 ? nxt_lane : lane);
    if (need64) {
      cnd = fov_wrap_cnd(az, az_next);
      vbad = !fov_coord_ok(ptx) || !fov_coord_ok(pty);
      ok = ok64;
      ipx = vbad ? 0 : (int32_t)ptx;
      ipy = vbad ? 0 : (int32_t)pty;
    }
  }
  const int ipy_next = __shfl(ipy, act ? nxt_lane : lane);
  const unsigned long long bt = __ballot(act && cnd) & gmask, bf = __ballot(act && !cnd) & gmask;
  const unsigned long long bad = __ballot(act && vbad) & gmask;
  int m = 0;
  if (act && ok && bt && bf && !bad) {
    const int count_true = __popcll(bt), count_false = __popcll(bf);
    const bool top = count_true == 1, wrap = top || count_false == 1;
    const int pp = (top ? __ffsll((long long)bt) : __ffsll((long long)bf)) - 1 - base;
    m = wrap ? N + 4 : N;
    int* qx = s_px[wave][g];
    int* qy = s_py[wave][g];
    const int idx = (wrap && k > pp) ? k + 4 : k;
    qx[idx] = ipx;
    qy[idx] = ipy;
    if (wrap && k == pp) {                                     // the border vertices between pp and pp + 1
      const int cols = dm.We, rows = dm.He;
      if (top) {
        qx[pp + 1] = cols; qy[pp + 1] = ipy;
        qx[pp + 2] = cols; qy[pp + 2] = 0;
        qx[pp + 3] = 0;    qy[pp + 3] = 0;
        qx[pp + 4] = 0;    qy[pp + 4] = ipy_next;
      } else {
        qx[pp + 1] = 0;    qy[pp + 1] = ipy;
        qx[pp + 2] = 0;    qy[pp + 2] = rows;
        qx[pp + 3] = cols; qy[pp + 3] = rows;
        qx[pp + 4] = cols; qy[pp + 4] = ipy_next;
      }
    }
  }
  if (ext) {                                                   // lane k of the group copies vertex k, k + N, ...
    const rr_ext_tile e = fr.ext[dg];
    const int np = imin(imax(e.n_poly, 0), POLY_STRIDE);
    bool fine = true;
    for (int v = k; v < np; v += N) {
      const double vx = e.poly_xy[2 * v], vy = e.poly_xy[2 * v + 1];
      fine = fine && fov_coord_ok(vx) && fov_coord_ok(vy);
      s_px[wave][g][v] = (int32_t)vx;
      s_py[wave][g][v] = (int32_t)vy;
    }
    const unsigned long long bad_e = __ballot(!fine) & gmask;
    m = bad_e ? 0 : np;
  }
  wave_lds_sync();
  {
    // Clipper's AddPath rejects a path whose vertices are all collinear (rr_device.h poly_all_collinear): lane k of the
    // group tests the vertices k, k + N, ... against the line through vertex 0 and any vertex that differs from it
    const int* qx = s_px[wave][imin(g, FOV_GROUPS - 1)];
    const int* qy = s_py[wave][imin(g, FOV_GROUPS - 1)];
    const int x0v = m > 0 ? qx[0] : 0, y0v = m > 0 ? qy[0] : 0;
    int vdif = -1;
    for (int v = k; v < m; v += N)
      if (vdif < 0 && (qx[v] != x0v || qy[v] != y0v)) vdif = v;
    const unsigned long long bd = __ballot(vdif >= 0) & gmask;
    const int va = __shfl(vdif, bd ? __ffsll((long long)bd) - 1 : lane);
    bool nz = false;
    if (bd) {
      const int64_t ax = (int64_t)qx[va] - x0v, ay = (int64_t)qy[va] - y0v;
      for (int v = k; v < m; v += N) nz = nz || (ax * ((int64_t)qy[v] - y0v) - ay * ((int64_t)qx[v] - x0v) != 0);
    }
    if ((__ballot(nz) & gmask) == 0ull) m = 0;                 // (also a group without a polygon)
  }
  if (act && k == 0) sc.npts[gi] = m;
  const int He = dm.He;
  int* xl = s_xl[wave];
  int* xr = s_xr[wave];
  uint32_t packed[FOV_GROUPS][NCH];                            // the finished spans of the wave's drops, lane = row
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    xl[c * 64 + lane] = 1 << 30;
    xr[c * 64 + lane] = -(1 << 30);
#pragma unroll
    for (int k = 0; k < FOV_GROUPS; k++) packed[k][c] = 0;
  }
  for (int gg = 0; gg < G; gg++) {
    const int mg = __builtin_amdgcn_readfirstlane(__shfl(m, gg * N));
    if (mg <= 0) continue;                                     // no polygon: k_fov_sums never reads this drop's spans
    // RR_OPT_FOV_FILL_RULE 1: OpenCV's rule (rr_device.h fov_rowspan_cv) for the closed, monotone N-gon with every vertex on the
    // map (fov_fill_rule_cv_applies); everything else -- the wrapping polygons above all -- keeps the span rule
    const bool cvg = cv_rule && fov_fill_rule_cv_applies(s_px[wave][gg], s_py[wave][gg], mg, N, He, dm.We);     // (wave-uniform)
    // lane e describes edge e and leaves the description in LDS: (ylo, first row, last row, xa), (dx, den, 1/(2 den)),
    // (walker step, hh, hr: edge_cv_consts)
    int cnt = 0;                                               // rows of the map the edge touches (<= 0: none)
    if (lane < mg) {
      const int j = (lane + 1 == mg) ? 0 : lane + 1;
      const int x0 = s_px[wave][gg][lane], y0 = s_py[wave][gg][lane], x1 = s_px[wave][gg][j], y1 = s_py[wave][gg][j];
      const bool swp = y1 < y0;
      const int ylo = swp ? y1 : y0, yhi = swp ? y0 : y1;
      const int xa = swp ? x1 : x0, dx = (swp ? x0 : x1) - xa;
      const int den = yhi - ylo;
      const float inv = den > 0 ? 1.0f / (float)(2 * den) : 0.f;
      const int ra = imax(ylo, 0), rb = imin(yhi, He - 1);
      cnt = rb - ra + 1;
      int d16 = 0, q_hh = 0, q_hr = 0;
      if (cvg && den > 0) edge_cv_consts(dx, den, d16, q_hh, q_hr);
      s_edge[wave][3 * lane] = make_int4(ylo, ra, rb, xa);
      s_edge[wave][3 * lane + 1] = make_int4(dx, den, __float_as_int(inv), 0);
      s_edge[wave][3 * lane + 2] = make_int4(d16, q_hh, q_hr, 0);
    }
    wave_lds_sync();                                           // edges published; span tables initialised (start / previous read-out)
    // Two edges per step, one per half wave (an edge covers ~30 rows: a whole wave per edge would idle half its
    // lanes), 32 rows of each per inner step; every lane fetches its edge's description with 16-byte reads.
    for (int e = 0; e < mg; e += 2) {
      const int eb = imin(e + 1, mg - 1);                      // (an odd last edge is done by both halves: min / max are idempotent)
      const int nmax = imax(__builtin_amdgcn_readlane(cnt, e), __builtin_amdgcn_readlane(cnt, eb));
      if (nmax <= 0) continue;
      const int me = lane < 32 ? e : eb;
      const int4 A = s_edge[wave][3 * me], B = s_edge[wave][3 * me + 1], C = s_edge[wave][3 * me + 2];
      const int ylo = A.x, ra = A.y, rbv = A.z, xa = A.w, dx = B.x, den = B.y;
      const float inv = __int_as_float(B.z);
      const bool hz = den == 0;                                // horizontal edge: both end points on its one row
      const int hz_l = imin(xa, xa + dx), hz_h = imax(xa, xa + dx);
      const int dx2 = 2 * dx, dn = 2 * den;
      for (int c = 0; c < nmax; c += 32) {
        const int y = ra + c + (lane & 31);
        if (y <= rbv) {                                        // a row is touched by one lane
          const int t = y - ylo;
          const int n2 = __mul24(dx2, t);                      // exact: |dx2 * t| < 2^23 (host check)
          int q = (int)floorf((float)n2 * inv);                // floor(n2 / dn) up to +-1 ...
          int rem = n2 - __mul24(q, dn);
          if (rem < 0) { q -= 1; rem += dn; }
          else if (rem >= dn) { q += 1; rem -= dn; }           // ... made exact
          int l = xa + q + (rem >= den ? 1 : 0), h = l;        // the span rule: floor((n2 + den) / dn)
          if (cvg && !hz) edge_row_cv(xa, xa + dx, den, dx, C.x, C.y, C.z, t, q, rem, l, h);
          atomicMin(&xl[y], hz ? hz_l : l);
          atomicMax(&xr[y], hz ? hz_h : h);
        }
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const int y = c * 64 + lane;
      const int a = imax(xl[y], 0), b = imin(xr[y], dm.We - 1);
      xl[y] = 1 << 30;                                         // ready for the wave's next drop
      xr[y] = -(1 << 30);
      uint32_t v = 0;
      if (y < He && a <= b) v = (uint32_t)a | ((uint32_t)(b + 1) << 16);
#pragma unroll
      for (int k = 0; k < FOV_GROUPS; k++)
        if (k == gg) packed[k][c] = v;
    }
  }
  // spans[frame][row][drop]
  bool have[FOV_GROUPS];                                       // (cross-lane reads stay outside the divergent stores)
  int dk[FOV_GROUPS];                                          // the drop of group k
#pragma unroll
  for (int k = 0; k < FOV_GROUPS; k++) {
    have[k] = k < G && __builtin_amdgcn_readfirstlane(__shfl(m, k < G ? k * N : 0)) > 0;
    dk[k] = __builtin_amdgcn_readfirstlane(__shfl(dg, k < G ? k * N : 0));
  }
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int y = c * 64 + lane;
    if (y < Hp) {
      uint32_t* out = sc.spans + ((int64_t)f * Hp + y) * Dp;
#pragma unroll
      for (int k = 0; k < FOV_GROUPS; k++)
        if (have[k]) out[dk[k]] = packed[k][c];
    }
  }
  if (!from_list) break;
  wave_lds_sync();                                             // (the wave's LDS tables are free for its next drops)
  }
}

// ---------------------------------------------------------------------------
// FOV polygon and row spans, one THREAD per drop (r04; float colour branch)
// ---------------------------------------------------------------------------
// k_fov_spans spreads a drop over a third of a wave: its edge walk (LDS min / max per row, a read-out pass) costs ~850
// issued instructions per drop, most of them with few lanes busy -- the kernel saturates the vector AND the scalar unit.
// Here a lane owns a drop from beginning to end:
//   1. the N vertices in float (fov_vertex32), the wrap test against the previous vertex, the predicates' error bounds --
//      fov_polygon_auto lane by lane; the vertex pixels wait in global memory ([vertex][drop]; rounds 4-5: in wave-private
//      LDS);
//   2. a drop whose polygon float can decide and that does not wrap has N vertices on a closed curve that every map row
//      crosses at most twice (a circle around the drop's direction that contains no pole): two cursors walk down from its
//      top vertex, one along each side, and every row's span is the min / max over the edges that touch the row -- the
//      very candidates fov_rowspan / k_fov_spans fold, evaluated with the same exact integer division -- so the spans
//      are identical.  A row's span leaves as a dword per lane, 256 bytes per wave, contiguous (the [row][drop] layout
//      k_fov_sums32 reads).  Every lane busy;
//   3. everything else -- a wrapping polygon (24 vertices, not monotone), a predicate within its error bound (float64
//      decides), a vertex sequence that is not monotone after all -- goes to the frame's list for k_fov_spans (from_list),
//      a fraction of a percent of the drops.
// r06, RR_OPT_FOV_FILL_RULE 1 (the default): the spans are what cv2.fillConvexPoly sets (rr_device.h fov_rowspan_cv: the
// outline's Bresenham pixels + the 16.16 edge walkers) for the polygons OpenCV's rule applies to (every vertex on the map),
// the span rule's otherwise.  Both by incremental cursors (rr_device.h DdaCursors): every edge's divisions are done once,
// before the walk, into a 16-byte record per edge and lane (global memory: step, outline constants, the edge's pixels on its
// first row, its lower end); a row costs adds and compares, and a cursor fetches its next record an edge ahead.  No LDS.
#ifndef RR_DDA_WAVES
#define RR_DDA_WAVES 1          // waves per workgroup: 1 (r06: no LDS, no barrier -- single waves fit into whatever a CU has free: 5.57 -> 5.31 ms alone, step 27.9 -> 27.6)
#endif
constexpr int DDA_WAVES = RR_DDA_WAVES;
__global__ __launch_bounds__(64 * DDA_WAVES) void k_fov_dda(const FrameDesc* frames, Dims dm, rr_camera cam, int max_drops, int Hp, int Dp, int cv_rule, Scratch sc) {
  const int f = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const FrameDesc& fr = frames[f];
  const int N = cam.n_fov;
  // NO LDS at all (r06): the tile / blur kernels of the other stream fill a CU's 160 KB, and a workgroup that needs a single
  // byte of it waits for one of theirs to leave.  The spin terms come from the kernel arguments (scalar loads, k is
  // wave-uniform), the vertex pixels wait in global memory (pixv below).
  const int i = (blockIdx.x * DDA_WAVES + wave) * 64 + lane;   // this lane's drop
  const bool act = i < fr.n_drops;
  const int64_t gi = (int64_t)f * max_drops + (act ? i : 0);
  if (fr.strategy == 1) {                                      // 'white': the FOV is computed by the reference but never used
    if (act) sc.npts[gi] = -1;
    return;
  }
  // vertex k's pixel, x | y << 16, in global memory ([frame][vertex][drop]: 256 contiguous bytes per wave and vertex, read
  // back from L2 when the edge records are made)
  uint4* const erec = sc.fov_erec + (int64_t)f * N * max_drops + (act ? i : 0);
  uint32_t* const gpix = sc.fov_pix + (int64_t)f * N * max_drops + (act ? i : 0);
  auto pixv = [&](int kk) -> uint32_t& { return gpix[(int64_t)kk * max_drops]; };
  uint32_t top_xy = 0;
  // ---- 1. vertices ----
  int uns = 0;                                                 // reason bits (rr_device.h): float64 has to decide
  int count_true = 0, count_false = 0;
  int ktop = 0, ytop = 1 << 30, ybot = -(1 << 30), r_first = 0;
  bool spread = false, on_map = true;
  int turns = 0, dir = 0, dir_first = 0;                       // sign changes of the vertices' row sequence (a closed monotone curve: 2)
  if (act) {
    FovSetup32 F;
    const rr_drop d = load_drop(fr.drops + i);
    fov_setup32(d, (float)cam.fov_cos, (float)cam.fov_sin, F, uns);
    float az_prev = 0.f, er_prev = 0.f, az0 = 0.f, er0 = 0.f;
    int y_prev = 0, y0v = 0;
    for (int k = 0; k < N; k++) {
      float az, er, pxf, pyf;
      fov_vertex32(F, (float)cam.radius, (float)cam.phi_cos[k], (float)cam.phi_sin[k], dm.He, dm.We, az, er, pxf, pyf, uns);
      const int ix = (int)pxf, iy = (int)pyf;
      const uint32_t vxy = (uint32_t)(ix & 0xffff) | ((uint32_t)(iy & 0xffff) << 16);
      pixv(k) = vxy;
      on_map = on_map && ix >= 0 && ix < dm.We && iy >= 0 && iy < dm.He;      // (fov_fill_rule_cv_applies)
      const int r = imin(imax(iy, 0), dm.He - 1);
      if (k == 0) { az0 = az; er0 = er; r_first = r; y0v = iy; }
      else {
        const bool c = fov_wrap_cnd32(az_prev, az, er_prev, er, uns);      // side k-1 -> k
        count_true += c ? 1 : 0;
        count_false += c ? 0 : 1;
        spread = spread || iabs(r - r_first) >= 2;
        const int sg = iy > y_prev ? 1 : (iy < y_prev ? -1 : 0);
        if (sg != 0) {
          if (dir == 0) dir_first = sg;
          else if (sg != dir) turns++;
          dir = sg;
        }
      }
      if (iy < ytop) { ytop = iy; ktop = k; top_xy = vxy; }
      ybot = imax(ybot, iy);
      az_prev = az; er_prev = er; y_prev = iy;
    }
    {                                                          // the closing side N-1 -> 0
      const bool c = fov_wrap_cnd32(az_prev, az0, er_prev, er0, uns);
      count_true += c ? 1 : 0;
      count_false += c ? 0 : 1;
      const int sg = y0v > y_prev ? 1 : (y0v < y_prev ? -1 : 0);
      if (sg != 0) {
        if (dir != 0 && sg != dir) turns++;
        dir = sg;
      }
      if (dir != 0 && dir_first != 0 && dir != dir_first) turns++;        // around the closing point
    }
  }
  if (act) {                                                   // a sliver: whether Clipper takes the path (poly_all_collinear) is float64's to say
    const uint32_t v0 = pixv(0), va = pixv(N / 2);
    const int x0v = (int)(v0 & 0xffffu), y0v = (int)(v0 >> 16), ax = (int)(va & 0xffffu) - x0v, ay = (int)(va >> 16) - y0v;
    bool sure_nc = false;
    for (int k = 1; k < N; k++) {
      const uint32_t v = pixv(k);
      sure_nc = sure_nc || poly_surely_not_collinear_step(ax, ay, (int)(v & 0xffffu) - x0v, (int)(v >> 16) - y0v);
    }
    if (!sure_nc) uns |= 256;
  }
  // ---- classification (fov_polygon_auto) ----
  const bool certain_fail = (uns & 128) && !(uns & 3);         // a vertex without intersection: [] like the reference
  const bool wrap = count_true == 1 || count_false == 1;
  const bool undecided = (uns & ~128) != 0 || count_true == 0 || count_false == 0 || (!wrap && !spread);
  const bool monotone = turns <= 2;
  const bool mine = act && !certain_fail && !undecided && !wrap && monotone;     // a sure, closed, monotone N-gon: spans below
  if (act && certain_fail) sc.npts[gi] = 0;
  if (act && !certain_fail && !mine) {                         // float64 / the general edge walk: k_fov_spans, from the list
    const int pos = atomicAdd(&sc.fov_list_n[f], 1);
    sc.fov_list[(int64_t)f * max_drops + pos] = i;
  }
  if (mine) sc.npts[gi] = N;
  if (__ballot(mine) == 0ull) return;
  // ---- 2. spans of the sure drops: two cursors down from the top vertex (rr_device.h DdaCursors) ----
  // The records of the edges {k, k + 1} (upper end first) go to global memory, [frame][edge][drop], 16 bytes each: the
  // walker's step, the packed outline constants, the edge's pixels on its first row and its lower end.  A wave stores 1 KB
  // per edge in one piece, a cursor loads its next record an edge (~30 rows) ahead.  (In LDS they left room for 10 waves per
  // CU instead of 24: 7.8 ms against 5.5 -- r06 A/B log.)  Round 6, second form: with the first row's pixels and the lower
  // end in the record a lane that takes an edge evaluates neither DdaCursors' pixels() nor a vertex fetch -- the path the whole
  // wave walks through on the 89 % of the rows on which one of its 64 cursors meets a vertex.
  const bool cvr = cv_rule && on_map;                          // (a vertex off the map: the span rule, like the oracle)
  if (mine) {
    const uint32_t vfirst = pixv(0);
    uint32_t va = vfirst;
    for (int k = 0; k < N; k++) {
      const uint32_t vb = k + 1 == N ? vfirst : pixv(k + 1);
      const int x0 = (int)(va & 0xffffu), y0 = (int)(va >> 16), x1 = (int)(vb & 0xffffu), y1 = (int)(vb >> 16);
      const bool swp = y1 < y0;
      uint32_t r[4];
      dda_edge_record(swp ? x1 : x0, swp ? y1 : y0, swp ? x0 : x1, swp ? y0 : y1, cvr, r);
      erec[(int64_t)k * max_drops] = make_uint4(r[0], r[1], r[2], r[3]);
      va = vb;
    }
  }
  auto rec = [&](int kk, uint32_t r[4]) {
    const uint4 v = erec[(int64_t)kk * max_drops];
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  };
  DdaCursors<decltype(rec)> cur;
  if (mine) cur.init(rec, N, ktop, top_xy);
  uint32_t* out = sc.spans + (int64_t)f * Hp * Dp + i;
  for (int y = 0; y < Hp; y++) {
    int lo = 1 << 30, hi = -(1 << 30);
    if (mine && y >= ytop && y <= ybot) cur.row(rec, y, lo, hi);
    const int a = imax(lo, 0), b = imin(hi, dm.We - 1);
    if (mine) out[(int64_t)y * Dp] = (y < dm.He && a <= b) ? ((uint32_t)a | ((uint32_t)(b + 1) << 16)) : 0u;
  }
}

// ---- wave64 scans on DPP (cross-lane moves inside the VALU; __shfl_* would go through the LDS crossbar) ----
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_move_f64(double v) {        // lanes without a source (bound_ctrl) or masked rows read 0.0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double row16_incl_scan_f64(double v) {  // inclusive scan inside every row of 16 lanes
  v += dpp_move_f64<0x111, 0xf>(v);                      // row_shr:1
  v += dpp_move_f64<0x112, 0xf>(v);                      // row_shr:2
  v += dpp_move_f64<0x114, 0xf>(v);                      // row_shr:4
  v += dpp_move_f64<0x118, 0xf>(v);                      // row_shr:8
  return v;
}
__device__ inline double wave_incl_scan_f64(double v) {
  v = row16_incl_scan_f64(v);
  v += dpp_move_f64<0x142, 0xa>(v);                      // row_bcast:15 into rows 1 and 3
  v += dpp_move_f64<0x143, 0xc>(v);                      // row_bcast:31 into rows 2 and 3
  return v;
}
// the same scans for min / max: a lane without a source (and a row the mask leaves out) keeps its own value -- min(v, v) = v
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_keep_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_min_scan_f64(double v) {     // lane 63: the wave's minimum
  v = dmin(v, dpp_keep_f64<0x111, 0xf>(v));
  v = dmin(v, dpp_keep_f64<0x112, 0xf>(v));
  v = dmin(v, dpp_keep_f64<0x114, 0xf>(v));
  v = dmin(v, dpp_keep_f64<0x118, 0xf>(v));
  v = dmin(v, dpp_keep_f64<0x142, 0xa>(v));
  v = dmin(v, dpp_keep_f64<0x143, 0xc>(v));
  return v;
}
__device__ inline double wave_max_scan_f64(double v) {     // lane 63: the wave's maximum
  v = dmax(v, dpp_keep_f64<0x111, 0xf>(v));
  v = dmax(v, dpp_keep_f64<0x112, 0xf>(v));
  v = dmax(v, dpp_keep_f64<0x114, 0xf>(v));
  v = dmax(v, dpp_keep_f64<0x118, 0xf>(v));
  v = dmax(v, dpp_keep_f64<0x142, 0xa>(v));
  v = dmax(v, dpp_keep_f64<0x143, 0xc>(v));
  return v;
}
__device__ inline double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// Sums of (x*w, y*w, Y*w, w) under every drop's spans.  Workgroup (frame, row band, component half, chunk of NT*DPT
// drops); half 0 accumulates (x*w, y*w), half 1 (Y*w, w).  Per map row of the band:
//   1. the row (xyY + solid angle per texel) is read from HBM -- once per batch: the workgroups of a (frame, band) run
//      on the same XCD in lock step and the later ones find it in L2 -- and the inclusive prefix sums of the half's two
//      components are built in LDS.  Wave w owns the columns [w*Cw, (w+1)*Cw) in passes of 128 consecutive columns,
//      TWO per lane: the lane adds its pair, one DPP scan per pass runs over the pair sums (carry between passes, wave
//      totals through LDS), and the pair's first column gets (inclusive sum - second value);
//   2. every thread takes its DPT drops' look-ups P[xr + 1] - P[xl] from LDS (an empty span is (0, 0): an exact
//      zero, no branch) and keeps the running sums in registers.
// The scan is the expensive part (it does not depend on the drops): splitting the four components over two workgroups
// that each carry twice the drops halves it, two columns per lane halve it again.  The next row's global loads are
// issued before the look-ups of the current one.  Band partials go to colpart[frame][band][5][drop]; the half-1
// workgroup of chunk 0 also leaves the band's row totals (sum w, sum Y*w) for the frame constants.  LDS: P as one array
// of double2: 16-byte entries spread a wave's random look-ups over all 64 banks.
constexpr int FOV_EMAX = 2;
template <int DPT, int EMAX>
__global__ __launch_bounds__(1024) void k_fov_sums(const FrameDesc* frames, Dims dm, int max_drops, int Hp, int Dp, int rpb, int nchunk, Scratch sc) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  const int We = dm.We;
  double2* s_P = reinterpret_cast<double2*>(s_dyn);      // [We + 1] inclusive prefix of the half's two components; entry 0 = zeros
  double* s_wt = reinterpret_cast<double*>(s_P + (We + 1));     // [16][2] wave totals of the row being scanned
  const int f = blockIdx.y, band = blockIdx.x % COL_PARTS, part = blockIdx.x / COL_PARTS;
  const int half = part & 1, chunk = part >> 1;
  const int NT = blockDim.x, t = threadIdx.x, lane = t & 63, nw = NT >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const FrameDesc& fr = frames[f];
  const int n = fr.n_drops;
  // the chunks of a frame take equal shares of its drops: workgroups that stream the same map rows then advance
  // at the same pace, and the one behind finds the rows in its XCD's L2
  const int per = (n + nchunk - 1) / nchunk;             // <= NT * DPT
  const int d0 = chunk * per, d1 = imin(n, d0 + per);
  if (d0 >= n) return;
  const int y0 = band * rpb, y1 = imin(dm.He, y0 + rpb);
  const int Cw = (((We + nw - 1) / nw) + 1) & ~1;        // columns per wave (even), taken in passes of 128 (<= EMAX passes)
  const int cw0 = wave * Cw;
  // spans of frame f: [Hp][Dp]; column max_drops of every row is all zeros (what a
  // drop without a polygon reads)
  const uint32_t* spf = sc.spans + (int64_t)f * Hp * Dp;
  uint32_t sp[DPT];
#pragma unroll
  for (int d = 0; d < DPT; d++) {
    const int i = d0 + d * NT + t;
    sp[d] = (uint32_t)((i < d1 && sc.npts[(int64_t)f * max_drops + i] > 0) ? i : max_drops);
  }
  if (t == 0) s_P[0] = make_double2(0.0, 0.0);
  double S[DPT][2];
  uint32_t any = 0;                                      // bit d: drop d had a non-empty span in this band
#pragma unroll
  for (int d = 0; d < DPT; d++) S[d][0] = S[d][1] = 0.0;
  double tot0 = 0.0, tot1 = 0.0;                         // row totals, kept by the thread that owns the last column
  double pa[EMAX][2], pb[EMAX][2];                       // the lane's column pair (first / second column) x two components
  const bool e32 = (fr.in_types & RR_IN_ENV_F32) != 0;
  auto load_row = [&](int y) {
    const global_ptr<const double> env = as_global(static_cast<const double*>(fr.env)) + (int64_t)y * We * 3;
    const global_ptr<const double> om = as_global(static_cast<const double*>(fr.omega)) + (int64_t)y * We;
    const global_ptr<const float> envf = as_global(static_cast<const float*>(fr.env)) + (int64_t)y * We * 3;
    const global_ptr<const float> omf = as_global(static_cast<const float*>(fr.omega)) + (int64_t)y * We;
    auto E = [&](int i) { return e32 ? (double)envf[i] : env[i]; };
    auto O = [&](int i) { return e32 ? (double)omf[i] : om[i]; };
#pragma unroll
    for (int e = 0; e < EMAX; e++) {
      const int cl = e * 128 + 2 * lane, c = cw0 + cl;
      pa[e][0] = pa[e][1] = pb[e][0] = pb[e][1] = 0.0;
      if (cl < Cw && c < We) {
        const double w = O(c);
        pa[e][0] = (half ? E(c * 3 + 2) : E(c * 3 + 0)) * w;
        pa[e][1] = half ? w : E(c * 3 + 1) * w;
        if (c + 1 < We) {
          const double w1 = O(c + 1);
          pb[e][0] = (half ? E(c * 3 + 5) : E(c * 3 + 3)) * w1;
          pb[e][1] = half ? w1 : E(c * 3 + 4) * w1;
        }
      }
    }
  };
  if (y0 < y1) load_row(y0);
  for (int yq = y0; yq < y1; yq += 4) {                  // y0 is a multiple of four: one 16-byte span load per drop and quad
    uint4 q[DPT];
#pragma unroll
    for (int d = 0; d < DPT; d++)                          // (rows yq .. yq + 3 < Hp: the spans are padded to a multiple of four rows)
      q[d] = make_uint4(spf[(int64_t)yq * Dp + sp[d]], spf[(int64_t)(yq + 1) * Dp + sp[d]], spf[(int64_t)(yq + 2) * Dp + sp[d]],
                        spf[(int64_t)(yq + 3) * Dp + sp[d]]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int y = yq + j;
      if (y >= y1) break;
      // ---- 1. prefix sums of row y into LDS ----
      double carry[2] = {0.0, 0.0};
      double ps[EMAX][2];                                // inclusive prefix through the lane's second column
#pragma unroll
      for (int e = 0; e < EMAX; e++) {
        if (e * 128 < Cw) {
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const double v = wave_incl_scan_f64(pa[e][k] + pb[e][k]) + carry[k];
            ps[e][k] = v;
            carry[k] = readlane_f64(v, 63);
          }
        }
      }
      if (lane == 0) { s_wt[wave * 2 + 0] = carry[0]; s_wt[wave * 2 + 1] = carry[1]; }
      __syncthreads();                                   // wave totals visible; the previous row's look-ups are done
      double basev[2];                                   // totals of the waves in front: 16-lane scan of the wave totals
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const double wt = row16_incl_scan_f64(lane < nw ? s_wt[lane * 2 + k] : 0.0);
        const double u = readlane_f64(wt, wave > 0 ? wave - 1 : 0);
        basev[k] = wave > 0 ? u : 0.0;
      }
#pragma unroll
      for (int e = 0; e < EMAX; e++) {
        const int cl = e * 128 + 2 * lane, c = cw0 + cl;
        if (cl < Cw && c < We) {
          const double o0 = basev[0] + ps[e][0], o1 = basev[1] + ps[e][1];       // through column c + 1
          s_P[c + 1] = make_double2(o0 - pb[e][0], o1 - pb[e][1]);               // through column c
          if (c + 1 < We) s_P[c + 2] = make_double2(o0, o1);
          if (c == We - 1 || c + 1 == We - 1) { tot0 += o0; tot1 += o1; }        // (an absent second column added 0)
        }
      }
      if (y + 1 < y1) load_row(y + 1);                   // in flight under the look-ups
      __syncthreads();                                   // row prefix complete
      // ---- 2. look-ups ----
#pragma unroll
      for (int d = 0; d < DPT; d++) {
        const uint32_t v = j == 0 ? q[d].x : (j == 1 ? q[d].y : (j == 2 ? q[d].z : q[d].w));
        any |= (v != 0u ? 1u : 0u) << d;
        const double2 h = s_P[v >> 16], l = s_P[v & 0xffffu];
        S[d][0] += h.x - l.x;
        S[d][1] += h.y - l.y;
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DPT; d++) {
    const int i = d0 + d * NT + t;
    if (i < d1) {
      double* o = sc.colpart + ((int64_t)(f * COL_PARTS + band) * 5) * max_drops + i;
      o[(int64_t)max_drops * (2 * half)] = S[d][0];
      o[(int64_t)max_drops * (2 * half + 1)] = S[d][1];
      if (half == 0) o[(int64_t)max_drops * 4] = ((any >> d) & 1u) ? 1.0 : 0.0;
    }
  }
  {                                                      // the thread that owns the last column
    const int cl = (We - 1 - cw0) & ~1;
    if (half == 1 && chunk == 0 && We - 1 >= cw0 && cl < Cw && ((cl >> 1) & 63) == lane) {
      sc.fband[(f * COL_PARTS + band) * 2 + 0] = tot1;   // sum w
      sc.fband[(f * COL_PARTS + band) * 2 + 1] = tot0;   // sum Y*w
    }
  }
}

// ---- the same sums in float32 (RR_OPT_FOV_F32; default off until measured and checked on the GPU) ----
// The sums only feed the drop's colour constants, and those only scale rainy_image, whose contract is +-1 LSB (the mask
// never sees them).  A float prefix row loses ~6e-5 absolute on prefixes of ~1e3, i.e. ~1e-6 of a span's sum: three orders
// below an LSB.  One workgroup then carries ALL FOUR components (16-byte LDS entries as before, so half the LDS bytes per
// look-up and half the workgroups, barriers and row scans of the float64 kernel), the scans move one dword per DPP step.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_move_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ inline float row16_incl_scan_f32(float v) {
  v += dpp_move_f32<0x111, 0xf>(v);
  v += dpp_move_f32<0x112, 0xf>(v);
  v += dpp_move_f32<0x114, 0xf>(v);
  v += dpp_move_f32<0x118, 0xf>(v);
  return v;
}
__device__ inline float wave_incl_scan_f32(float v) {
  v = row16_incl_scan_f32(v);
  v += dpp_move_f32<0x142, 0xa>(v);
  v += dpp_move_f32<0x143, 0xc>(v);
  return v;
}
__device__ inline uint32_t wave_incl_scan_u32(uint32_t v) {              // (zeros shift in: bound_ctrl)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
  return v;
}
__device__ inline float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// r05: the kernel's time per map row was 4.75 us for 1 us of vector work: with ONE workgroup per CU nothing covers the
// latency of the next row's loads (issued a lookup phase ahead) nor that of the span pieces (four rows at a time, used at
// once).  Measured with the loads replaced by arithmetic: 3.8 ms -> 2.7 (no row loads) / 3.1 (no span loads) / 1.8 (neither).
// Now both travel a whole row ahead: a row's texels stay in registers AS LOADED and the scan forms its products from them
// first thing -- the same registers then receive the next row at once (r04 asked for it after the prefix row had been
// written, a lookup phase before it was needed); the spans come a dword per drop and row ([row][drop] layout), the next
// row's requested at the top of the current one: 16 registers in place of the quads' 32, and the kernel no longer spills.
// Sums, scan and their order are unchanged: same bits.
template <int DPT, int EMAX, bool E32>
__global__ __launch_bounds__(1024) void k_fov_sums32(const FrameDesc* frames, Dims dm, int max_drops, int Hp, int Dp, int rpb, int nchunk, Scratch sc) {
  extern __shared__ __attribute__((aligned(16))) float s_dyn32[];
  const int We = dm.We;
  float4* s_P = reinterpret_cast<float4*>(s_dyn32);      // [We + 1] inclusive prefix of (x*w, y*w, Y*w, w); entry 0 = zeros
  float* s_wt = reinterpret_cast<float*>(s_P + (We + 1));  // [16][4] wave totals of the row being scanned
  const int f = blockIdx.y, band = blockIdx.x % COL_PARTS, chunk = blockIdx.x / COL_PARTS;
  const int NT = blockDim.x, t = threadIdx.x, lane = t & 63, nw = NT >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const FrameDesc& fr = frames[f];
  const int n = fr.n_drops;
  const int per = (n + nchunk - 1) / nchunk;
  const int d0 = chunk * per, d1 = imin(n, d0 + per);
  if (d0 >= n) return;
  const int y0 = band * rpb, y1 = imin(dm.He, y0 + rpb);
  const int Cw = (((We + nw - 1) / nw) + 1) & ~1;
  const int cw0 = wave * Cw;
  uint32_t sp[DPT];
#pragma unroll
  for (int d = 0; d < DPT; d++) {
    const int i = d0 + d * NT + t;
    sp[d] = 4u * (uint32_t)((i < d1 && sc.npts[(int64_t)f * max_drops + i] > 0) ? i : max_drops);      // byte offset inside a span row
  }
  if (t == 0) s_P[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  float S[DPT][4];
  uint32_t any = 0;
#pragma unroll
  for (int d = 0; d < DPT; d++) S[d][0] = S[d][1] = S[d][2] = S[d][3] = 0.f;
  double totY = 0.0, totw = 0.0;                         // row totals (Y*w, w), kept by the thread that owns the last column
  // A map row in registers: the lane's two texels per pass AS LOADED (x y Y x' | y' Y' | w w') -- the products with the solid
  // angle are formed where the scan takes them (formed at the load, they made the load synchronous: the multiplications
  // waited for the data a few instructions after the request).  The two array pointers of the frame are read ONCE, into
  // scalar registers: left to the compiler they were fetched from the frame descriptor again for every row, with a
  // vmcnt(0) that drained every load in flight.
  struct RowRegs {
    float4_u a[EMAX];
    float2_u d[EMAX], w[EMAX];
  };
  // Loads of the row loop go through buffer descriptors (base in scalar registers, a 32-bit lane offset, a scalar row offset):
  // no 64-bit address per load and lane -- with plain pointers the addresses alone took 30 registers and the kernel spilled.
  // Every one of them is UNCONDITIONAL and the same for every lane (clamped offsets; what a lane must not use is discarded
  // where the scan forms its products): the compiler then counts the loads in flight exactly.  With a load inside a branch it
  // waited with vmcnt(0) -- for the row it had just requested.
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  auto descriptor = [](const void* p) {                    // raw buffer, no stride, 2 GB range (gfx9 word 3: 32-bit data format)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs_env = descriptor(fr.env), rs_om = descriptor(fr.omega);
  const __amdgpu_buffer_rsrc_t rs_sp = descriptor(sc.spans + (int64_t)f * Hp * Dp);
  auto load_row = [&](int y, RowRegs& R) {
#pragma unroll
    for (int e = 0; e < EMAX; e++) {
      const int c = cw0 + e * 128 + 2 * lane;
      const int cb = imax(imin(c, We - 2), 0);             // the pair (cb, cb + 1) exists; the lane of the map's last column reads (We - 2, We - 1)
      if (E32) {                                           // float map: 12 + 4 bytes per texel
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs_env, cb * 12, y * We * 12, 0);          // x y Y of texel cb, x of texel cb + 1
        const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs_env, cb * 12 + 16, y * We * 12, 0);      // y Y of texel cb + 1
        const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs_om, cb * 4, y * We * 4, 0);
        R.a[e] = float4_u{__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
        R.d[e] = float2_u{__uint_as_float(d.x), __uint_as_float(d.y)};
        R.w[e] = float2_u{__uint_as_float(w.x), __uint_as_float(w.y)};
      } else {                                             // float64 map: rounded to float as it arrives (the products are float either way)
        auto f64at = [&](const __amdgpu_buffer_rsrc_t& rs, int voff, int soff) {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
          return (float)__hiloint2double((int)v.y, (int)v.x);
        };
        const int so = y * We * 24, vo = cb * 24;
        R.a[e] = float4_u{f64at(rs_env, vo, so), f64at(rs_env, vo + 8, so), f64at(rs_env, vo + 16, so), f64at(rs_env, vo + 24, so)};
        R.d[e] = float2_u{f64at(rs_env, vo + 32, so), f64at(rs_env, vo + 40, so)};
        R.w[e] = float2_u{f64at(rs_om, cb * 8, y * We * 8), f64at(rs_om, cb * 8 + 8, y * We * 8)};
      }
    }
  };
  // one map row: R holds it; `cur` are this row's spans, `nxt` receives the next row's
  auto row = [&](int y, RowRegs& R, uint32_t (&cur)[DPT], uint32_t (&nxt)[DPT]) {
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    float ps[EMAX][4], pa[EMAX][4], pb[EMAX][4];           // the lane's two columns per pass: (x*w, y*w, Y*w, w)
#pragma unroll
    for (int e = 0; e < EMAX; e++) {
      const int cl = e * 128 + 2 * lane, c = cw0 + cl;
      const bool v0 = cl < Cw && c < We, v1 = cl < Cw && c + 1 < We, last = v0 && !v1;       // last: the map's last column (the SECOND texel of what was loaded)
      const float wa = v0 ? (last ? R.w[e].y : R.w[e].x) : 0.f, wb = v1 ? R.w[e].y : 0.f;
      pa[e][0] = (last ? R.a[e].w : R.a[e].x) * wa; pa[e][1] = (last ? R.d[e].x : R.a[e].y) * wa; pa[e][2] = (last ? R.d[e].y : R.a[e].z) * wa; pa[e][3] = wa;
      pb[e][0] = R.a[e].w * wb; pb[e][1] = R.d[e].x * wb; pb[e][2] = R.d[e].y * wb; pb[e][3] = wb;
    }
    load_row(imin(y + 1, y1 - 1), R);                      // R is free: the next row, a whole row ahead (the last row asks for itself again)
#pragma unroll
    for (int d = 0; d < DPT; d++) nxt[d] = __builtin_amdgcn_raw_buffer_load_b32(rs_sp, (int)sp[d], imin(y + 1, Hp - 1) * Dp * 4, 0);
#pragma unroll
    for (int e = 0; e < EMAX; e++) {
      if (e * 128 < Cw) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float v = wave_incl_scan_f32(pa[e][k] + pb[e][k]) + carry[k];
          ps[e][k] = v;
          carry[k] = readlane_f32(v, 63);
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) s_wt[wave * 4 + k] = carry[k];
    }
    __syncthreads();
    float basev[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float wt = row16_incl_scan_f32(lane < nw ? s_wt[lane * 4 + k] : 0.f);
      const float u = readlane_f32(wt, wave > 0 ? wave - 1 : 0);
      basev[k] = wave > 0 ? u : 0.f;
    }
#pragma unroll
    for (int e = 0; e < EMAX; e++) {
      const int cl = e * 128 + 2 * lane, c = cw0 + cl;
      if (cl < Cw && c < We) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = basev[k] + ps[e][k];                 // through column c + 1
        s_P[c + 1] = make_float4(o[0] - pb[e][0], o[1] - pb[e][1], o[2] - pb[e][2], o[3] - pb[e][3]);
        if (c + 1 < We) s_P[c + 2] = make_float4(o[0], o[1], o[2], o[3]);
        if (c == We - 1 || c + 1 == We - 1) { totY += (double)o[2]; totw += (double)o[3]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DPT; d++) {
      const uint32_t v = cur[d];
      any |= (v != 0u ? 1u : 0u) << d;
      const float4 h = s_P[v >> 16], l = s_P[v & 0xffffu];
      S[d][0] += h.x - l.x;
      S[d][1] += h.y - l.y;
      S[d][2] += h.z - l.z;
      S[d][3] += h.w - l.w;
    }
#pragma unroll
    for (int d = 0; d < DPT; d++) cur[d] = nxt[d];
  };
  RowRegs RA;
  uint32_t qa[DPT], qb[DPT];
  if (y0 < y1) {
    load_row(y0, RA);
#pragma unroll
    for (int d = 0; d < DPT; d++) qa[d] = __builtin_amdgcn_raw_buffer_load_b32(rs_sp, (int)sp[d], y0 * Dp * 4, 0);
  }
  for (int y = y0; y < y1; y++) row(y, RA, qa, qb);
#pragma unroll
  for (int d = 0; d < DPT; d++) {
    const int i = d0 + d * NT + t;
    if (i < d1) {
      float* o = reinterpret_cast<float*>(sc.colpart) + ((int64_t)(f * COL_PARTS + band) * 5) * max_drops + i;      // (sc.colpart_f32)
#pragma unroll
      for (int k = 0; k < 4; k++) o[(int64_t)max_drops * k] = S[d][k];
      o[(int64_t)max_drops * 4] = ((any >> d) & 1u) ? 1.0f : 0.0f;
    }
  }
  {
    const int cl = (We - 1 - cw0) & ~1;
    if (chunk == 0 && We - 1 >= cw0 && cl < Cw && ((cl >> 1) & 63) == lane) {
      sc.fband[(f * COL_PARTS + band) * 2 + 0] = totw;   // sum w
      sc.fband[(f * COL_PARTS + band) * 2 + 1] = totY;   // sum Y*w
    }
  }
}

// ---- general path (maps taller than HE_MAX / wider than FOV_WE_MAX): polygon per thread, prefix table in HBM ----
__global__ __launch_bounds__(128) void k_fov_poly_general(const FrameDesc* frames, Dims dm, rr_camera cam, int max_drops, Scratch sc) {
  const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const FrameDesc& fr = frames[f];
  if (i >= fr.n_drops) return;
  const int64_t gi = (int64_t)f * max_drops + i;
  int32_t* px = sc.poly + gi * 2 * POLY_STRIDE;
  const rr_drop d = load_drop(fr.drops + i);
  int npts = fov_polygon(d, cam, dm.He, dm.We, px, px + POLY_STRIDE);
  if (fr.strategy == 1) npts = -1;
  sc.npts[gi] = npts;
}

// one wave per drop: per-row edge scan x prefix table (gathers from HBM / L2)
__global__ __launch_bounds__(256) void k_fov_sums_general(const FrameDesc* frames, Dims dm, int max_drops, Scratch sc, int cv_rule, int n_fov) {
  const int f = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= frames[f].n_drops) return;
  const int64_t gi = (int64_t)f * max_drops + i;
  const int n = sc.npts[gi];
  double S[4] = {0, 0, 0, 0};
  int any = 0;
  if (n > 0) {
    const int32_t* px = sc.poly + gi * 2 * POLY_STRIDE;
    const int32_t* py = px + POLY_STRIDE;
    int ymin = lane < n ? py[lane] : 0x7fffffff, ymax = lane < n ? py[lane] : -0x7fffffff;
    for (int ofs = 32; ofs > 0; ofs >>= 1) {
      ymin = min(ymin, __shfl_xor(ymin, ofs));
      ymax = max(ymax, __shfl_xor(ymax, ofs));
    }
    const int ya = max(ymin, 0), yb = min(ymax, dm.He - 1);
    const double* P = sc.prefix + (int64_t)f * dm.He * (int64_t)(dm.We + 1) * 4;
    // RR_OPT_FOV_FILL_RULE 1: OpenCV's own fill rule for the polygons it is defined for (rr_device.h fov_rowspan_cv)
    const bool cv = cv_rule && fov_fill_rule_cv_applies(px, py, n, n_fov, dm.He, dm.We);
    for (int y = ya + lane; y <= yb; y += 64) {
      int xl_, xr_;
      if (cv ? fov_rowspan_cv(px, py, n, y, dm.We, xl_, xr_) : fov_rowspan_fast(px, py, n, y, dm.We, xl_, xr_)) {
        any = 1;
        const double* row = P + (int64_t)y * (dm.We + 1) * 4;
        const double* hi = row + (int64_t)(xr_ + 1) * 4;
        const double* lo = row + (int64_t)xl_ * 4;
        for (int k = 0; k < 4; k++) S[k] += hi[k] - lo[k];             // P[row][0] == 0
      }
    }
    for (int ofs = 32; ofs > 0; ofs >>= 1) {
      for (int k = 0; k < 4; k++) S[k] += __shfl_xor(S[k], ofs);
      any |= __shfl_xor(any, ofs);
    }
  }
  if (lane == 0) {                                       // band 0 carries everything, the other bands are zero
    for (int b = 0; b < COL_PARTS; b++) {
      double* o = sc.colpart + ((int64_t)(f * COL_PARTS + b) * 5) * max_drops + i;
      for (int k = 0; k < 4; k++) o[(int64_t)max_drops * k] = b == 0 ? S[k] : 0.0;
      o[(int64_t)max_drops * 4] = b == 0 ? (double)any : 0.0;
    }
  }
}

// Colour, pass 2: one thread per drop adds the band partials in band order and writes the
// compositor record.
__global__ __launch_bounds__(256) void k_colour(const FrameDesc* frames, Dims dm, int max_drops, double cam_exposure, Scratch sc) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const FrameDesc& fr = frames[f];
  if (i >= fr.n_drops) return;
  const int64_t gi = (int64_t)f * max_drops + i;
  const DropPlan& p = sc.plan[gi];
  CompRec rec;
  rec.x0 = rec.y0 = rec.x1 = rec.y1 = 0;
  rec.ox = rec.oy = rec.pitch = rec.pad = 0;
  rec.off = 0;
  rec.tau_one = rec.g = 0;
  rec.K[0] = rec.K[1] = rec.K[2] = 0;
  rec.zdist = fr.depth ? fabs(as_global((const double*)fr.drops[i].wps)[2]) : 0.0;
  int status = p.status;
  bool blended = false;
  const int n = sc.npts[gi];
  if (n == 0) status = RR_DROP_FOV_FAIL;
  if (n < 0) {                         // rendering_strategy 'white': gray tile, no colour
    if (status == RR_DROP_OK && sc.sizes[gi] > 0) {
      blended = true;
      rec.K[0] = rec.K[1] = rec.K[2] = 1.0;
      rec.x0 = p.vis_x0;
      rec.y0 = p.vis_y0;
      rec.x1 = p.vis_x0 + p.vis_w;
      rec.y1 = p.vis_y0 + p.vis_h;
      rec.ox = -p.vis_x0;
      rec.oy = -p.vis_y0;
      rec.pitch = p.tw;
      rec.off = p.a0_off;
      rec.tau_one = p.tau_one;
      rec.g = p.g;
    }
  }
  if (n > 0) {
    double S[4] = {0, 0, 0, 0};
    bool any = false;
    double sumW = 0.0, sumY = 0.0;                       // whole-map sums (bad_weather.py:403-404), band order
    for (int b = 0; b < COL_PARTS; b++) {
      const int64_t at = ((int64_t)(f * COL_PARTS + b) * 5) * max_drops + i;
      if (sc.colpart_f32) {                                // (wave-uniform) float partials: the same values, read as they were summed
        const float* part = reinterpret_cast<const float*>(sc.colpart) + at;
        for (int k = 0; k < 4; k++) S[k] += (double)part[(int64_t)max_drops * k];
        any = any || part[(int64_t)max_drops * 4] != 0.0f;
      } else {
        const double* part = sc.colpart + at;
        for (int k = 0; k < 4; k++) S[k] += part[(int64_t)max_drops * k];
        any = any || part[(int64_t)max_drops * 4] != 0.0;
      }
      sumW += sc.fband[(f * COL_PARTS + b) * 2 + 0];
      sumY += sc.fband[(f * COL_PARTS + b) * 2 + 1];
    }
    if (!any) status = RR_DROP_EMPTY_FOV;
    if (status == RR_DROP_OK && sc.sizes[gi] > 0) {
      blended = true;
      colour_from_sums(S, sumW, sumY / sumW, rec.K);
      if (p.r1 > 0) {                  // finished effective tile written by the blur kernels
        const int fx0 = p.vis_x0 - p.crop_x + (p.shift - p.r2), fy0 = p.vis_y0 - p.crop_y + (p.shift - p.r1);   // its frame position
        rec.x0 = imax(p.vis_x0, fx0);
        rec.y0 = imax(p.vis_y0, fy0);
        rec.x1 = imin(p.vis_x0 + p.vis_w, fx0 + p.ew);
        rec.y1 = imin(p.vis_y0 + p.vis_h, fy0 + p.eh);
        if (rec.x1 <= rec.x0 || rec.y1 <= rec.y0) { rec.x0 = rec.y0 = rec.x1 = rec.y1 = 0; }
        rec.ox = -(fx0 - p.epad);
        rec.oy = -fy0;
        rec.pitch = p.epitch;
        rec.off = p.a1_off;
      } else {                         // no blur: the pad is exact zeros (a no-op in the blend); read the raw tile
        const int rx0 = p.vis_x0 - p.crop_x + p.shift, ry0 = p.vis_y0 - p.crop_y + p.shift;   // frame position of raw (0,0)
        rec.x0 = imax(p.vis_x0, rx0);
        rec.y0 = imax(p.vis_y0, ry0);
        rec.x1 = imin(p.vis_x0 + p.vis_w, rx0 + p.tw);
        rec.y1 = imin(p.vis_y0 + p.vis_h, ry0 + p.th);
        if (rec.x1 <= rec.x0 || rec.y1 <= rec.y0) { rec.x0 = rec.y0 = rec.x1 = rec.y1 = 0; }
        rec.ox = -rx0;
        rec.oy = -ry0;
        rec.pitch = p.tw;
        rec.off = p.a0_off;
      }
      rec.tau_one = p.tau_one;
      rec.g = p.g;
    }
  }
  // pad = 1: the compositor takes the literal blend_pixel for this drop (its short form needs every factor finite and
  // the quotient A * tau / exposure in the normal range; a caller-made tile may hold anything)
  {
    const double BIG = 1.0e50;
    const bool tame = fabs(rec.tau_one) < BIG && (rec.tau_one == 0.0 || rec.tau_one > 1.0e-200) && fabs(rec.g) < BIG &&
                      fabs(rec.K[0]) < BIG && fabs(rec.K[1]) < BIG && fabs(rec.K[2]) < BIG && p.kind != KIND_EXT;
    rec.pad = tame ? 0 : 1;
  }
  sc.comp[gi] = rec;
  {
    CompRec32 r32;
    r32.xx = (uint32_t)(rec.x0 & 0xffff) | ((uint32_t)(rec.x1 & 0xffff) << 16);
    r32.yy = (uint32_t)(rec.y0 & 0xffff) | ((uint32_t)(rec.y1 & 0xffff) << 16);
    r32.base = rec.off + ((int64_t)rec.oy * rec.pitch + rec.ox);
    r32.pitch_slow = (uint32_t)rec.pitch | ((uint32_t)(rec.pad ? 1 : 0) << 31);
    r32.te = (float)(rec.tau_one / cam_exposure);
    r32.kg[0] = (float)(rec.K[0] * rec.g); r32.kg[1] = (float)(rec.K[1] * rec.g); r32.kg[2] = (float)(rec.K[2] * rec.g);
    r32.zdist = rec.zdist;
    r32.spare0 = 0;
    r32.spare[0] = r32.spare[1] = r32.spare[2] = r32.spare[3] = 0;
    sc.comp32[gi] = r32;
  }
  sc.bbox[gi] = make_int4(rec.x0, rec.y0, rec.x1, rec.y1);
  sc.blended[gi] = blended ? 1 : 0;
  if (fr.colour_out) {
    const global_ptr<double> ko = as_global(fr.colour_out) + (int64_t)i * 3;
    ko[0] = rec.K[0]; ko[1] = rec.K[1]; ko[2] = rec.K[2];
  }
  if (fr.status) as_global(fr.status)[i] = status;
}

// ---------------------------------------------------------------------------
// tile synthesis
// ---------------------------------------------------------------------------
// One 256-thread block per drop.
//   * the texture (u8, with a 2-texel zero border) and the 256-entry v/255.0 table live in LDS;
//   * cv2.resize(INTER_AREA) of the rotated canvas -- ~nW*nH bilinear samples for a handful of
//     output pixels -- is evaluated in three LDS-staged steps per chunk of canvas rows:
//       1a  lanes run ALONG canvas rows and sample only the column interval that can touch
//           the texture (samples outside are exactly 0 and adding them is a no-op);
//       1b  one lane per (canvas row, destination column) folds its samples left to right
//           with the resizeArea_ weights -> s_buf;
//       2   one lane per output pixel folds s_buf top to bottom.
//     Every fold runs in the order resizeArea_ uses, so the tile is bit-identical to
//     raw_tile_pixel (rr_device.h, the one-thread-per-pixel definition) / the oracle.
constexpr int TEX_LDS = 11776;      // padded texels: (h+4)*(w+4) <= TEX_LDS (a multiple of 16: load_tex_copy)
constexpr int NW_MAX = 384;
constexpr int TW_MAX = 64;
constexpr int BUF_MAX = 512;
constexpr int CAN_W = 480;        // doubles of sample staging per wave
constexpr int ROWS_S = 32;        // canvas rows whose intervals a wave works out at a time (general path)
constexpr int ROWS_W = 16;        // canvas rows a wave stages at most

// Columns rx of canvas row `ry` (un-flipped row number) whose bilinear footprint can touch
// the texture, conservatively (+-1 texel, +-1 column): [xa, xa+n).  Samples outside are
// exactly 0.0.  The interval length is bounded by tile_pitch() for every row.
struct RowGeom {                    // per-drop constants of row_interval
  double inv[2];                    // 1 / (m0*1024), 1 / (m3*1024); 0 when the axis does not depend on rx
  double U[2];
};
__device__ inline RowGeom row_geom(const DropPlan& p, int sh, int sw) {
  RowGeom g;
  const double A[2] = {p.ma[0] * 1024.0, p.ma[3] * 1024.0};
  for (int k = 0; k < 2; k++) g.inv[k] = fabs(A[k]) < 1e-6 ? 0.0 : 1.0 / A[k];
  g.U[0] = (double)(sw + 1) * 1024.0;
  g.U[1] = (double)(sh + 1) * 1024.0;
  return g;
}
// X0/Y0 are the row's fixed-point terms (rot_X0 / rot_Y0)
__device__ inline void row_interval(const DropPlan& p, const RowGeom& g, int X0, int Y0, int& xa, int& n) {
  double lo = 0.0, hi = (double)(p.nW - 1);
  const double C[2] = {(double)X0, (double)Y0};
  const double L = -2048.0;
  bool empty = false;
  for (int k = 0; k < 2; k++) {
    if (g.inv[k] == 0.0) {
      if (C[k] < L - 1024.0 || C[k] > g.U[k] + 1024.0) empty = true;
    } else {
      double t0 = (L - C[k]) * g.inv[k], t1 = (g.U[k] - C[k]) * g.inv[k];
      if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
      lo = fmax(lo, floor(t0) - 2.0);
      hi = fmin(hi, ceil(t1) + 2.0);
    }
  }
  if (empty || lo > hi) { xa = 0; n = 0; return; }
  xa = (int)lo;
  n = (int)hi - xa + 1;
}
// upper bound of row_interval's n over all rows
__device__ inline int tile_pitch(const DropPlan& p, int sh, int sw) {
  double w = (double)p.nW;
  const double A[2] = {fabs(p.ma[0]) * 1024.0, fabs(p.ma[3]) * 1024.0};
  const double span[2] = {(double)(sw + 1) * 1024.0 + 2048.0, (double)(sh + 1) * 1024.0 + 2048.0};
  for (int k = 0; k < 2; k++)
    if (A[k] >= 1e-6) w = fmin(w, span[k] / A[k] + 8.0);
  return (int)w;
}

// which drops take the LDS-staged rotate+area-resize path (everything else: k_tile_generic)
// fixed-point coordinates stay inside the int32 / short range the LDS sampler assumes
__device__ inline bool tile_coords_safe(const DropPlan& p) {
  const double cmax = (fabs(p.ma[1]) * p.nH + fabs(p.ma[2]) + fabs(p.ma[0]) * p.nW + fabs(p.ma[4]) * p.nH + fabs(p.ma[5]) +
                       fabs(p.ma[3]) * p.nW) * 1024.0 + 64.0;
  return cmax < 3.0e7;
}
__device__ inline bool tile_is_fast(const DropPlan& p, int sh, int sw) {
  if (!((sh + 4) * (sw + 4) <= TEX_LDS && p.kind == KIND_ROT && p.nW <= NW_MAX && p.tw <= TW_MAX && tile_coords_safe(p))) return false;
  if (p.rs_mode == RS_AREA_FAST) return true;                      // integer ratios: per-wave sequential chains
  const int rows_per_dy = (int)ceil(p.scale_y) + 4;          // (floor(dy1 * s) + 1) - (floor(dy0 * s) - 1) + 1 <= floor(k * s) + 4 rows per group
  return p.rs_mode == RS_AREA && rows_per_dy <= BUF_MAX;             // wide tiles are folded in column chunks
}

// ---- k_tile_rows (round 6): a WAVE per rotate + INTER_AREA tile, lanes = canvas rows (rr_device.h "row walks") ----
constexpr int RW_WAVES = 16;          // waves of a workgroup (one workgroup per CU: the LDS holds one texture for all of them)
constexpr int RW_NW = 324;            // canvas columns of a tile (sh = 320, sw = 32: nW <= 321)
constexpr int RW_BUF = 344;           // doubles of cell sums per wave
constexpr int RW_PAIR_BYTES = 24640;  // pair texture in LDS: (320 + 3) * 38 * 2 rounded up to 16
constexpr int RW_TEX_MAX = 1024;      // textures of a database the batch-wide list is bucketed by
constexpr int RW_SHARE_MAX = 4096;    // (>= compute units * RW_SHARES)
struct RowsWave {                     // wave-private LDS of k_tile_rows
  ColEnt col[RW_NW];
  double buf[RW_BUF];
  uint8_t cell[RW_NW + 12];
  uint16_t cfirst[64], clast[64];     // first / last canvas column of every destination cell
};
static_assert(sizeof(RowsWave) == 8528 && sizeof(RowsWave) % 16 == 0, "RowsWave layout (col[] is read 16 bytes at a time)");
// estimated cost of a tile of this texture (its samples ~ the padded texture's area, plus the per-tile set-up), for the
// split of the sorted list among the workgroups
// estimated cost of a tile in walk iterations (passes x columns a row can touch, + the per-pass and per-tile set-up)
__device__ inline int rows_tile_cost(const DropPlan& p, int sh, int sw) {
  int R, NS;
  rows_pass_shape(p.tw, RW_BUF, R, NS);
  const int passes = (p.nH + R - 1) / R;
  return passes * ((imin(tile_pitch(p, sh, sw), p.nW) / NS + 3) / 2 + 6) + 24;
}
__device__ inline bool tile_is_rows(const DropPlan& p, int sh, int sw) {
  return p.kind == KIND_ROT && p.rs_mode == RS_AREA && p.scale_x >= 2.0 && p.nW <= RW_NW && p.tw <= 64 && p.tw >= 1 && p.th >= 1 && p.th <= 64 &&
         pair_bytes(sh, sw) <= RW_PAIR_BYTES && tile_coords_safe(p);
}
// Big (bicubic warp) tiles k_tile_rows takes: the padded texture (2-texel zero border, k_pad_textures) in the LDS region of the
// pair texture.  A tile is ONE wave's work there, 64 pixels a pass: tiles of more than BIG_ROWS_MAX_PX pixels stay with
// k_tile_big, whose threads take a pixel each across the whole device (nuScenes at f/1.8 has Big tiles of 10^5 pixels: as a
// single wave's 1900 passes one of them was 0.75 ms, the whole kernel's time -- r06 A/B log).
constexpr int BIG_ROWS_MAX_PX = 8192;
__device__ inline bool tile_is_big_lds(const DropPlan& p, int sh, int sw) {
  return p.kind == KIND_BIG && (int64_t)(sh + 4) * (sw + 4) + 48 <= RW_PAIR_BYTES && p.tw >= 1 && p.th >= 1 && (int64_t)p.tw * p.th <= BIG_ROWS_MAX_PX && p.bw0 >= 1;
}
__global__ __launch_bounds__(256) void k_pair_textures(const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w,
                                                       const int64_t* tex_off, const int64_t* tex_qoff, uint8_t* pairs) {
  const int i = blockIdx.x, sh = tex_h[i], sw = tex_w[i], P = pair_pitch(sw);
  const uint8_t* g = texels + tex_off[i];
  uint16_t* dst = reinterpret_cast<uint16_t*>(pairs + tex_qoff[i]);
  for (int k = threadIdx.x; k < (sh + 3) * P; k += 256) {
    const int y = k / P - 2, x = k - (y + 2) * P - 2;
    const bool xin = x >= 0 && x < sw;
    const uint32_t lo = (xin && y >= 0 && y < sh) ? g[y * sw + x] : 0u, hi = (xin && y + 1 >= 0 && y + 1 < sh) ? g[(y + 1) * sw + x] : 0u;
    dst[k] = (uint16_t)(lo | (hi << 8));
  }
}

// The textures once more, each with its 2-texel zero border and pitch w + 4 -- byte for byte what load_tex_padded builds in
// LDS -- made when the database is set: a tile then stages its texture with 16-byte copies (three per thread for a 32x320
// texture) instead of placing 10 K bytes one by one (four index computations and four one-byte LDS writes per dword).
__global__ __launch_bounds__(256) void k_pad_textures(const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w,
                                                      const int64_t* tex_off, const int64_t* tex_poff, uint8_t* pad) {
  const int i = blockIdx.x, sh = tex_h[i], sw = tex_w[i], P = sw + 4;
  const uint8_t* g = texels + tex_off[i];
  uint8_t* dst = pad + tex_poff[i];
  for (int k = threadIdx.x; k < (sh + 4) * P; k += 256) {
    const int y = k / P - 2, x = k - (y + 2) * P - 2;
    dst[k] = (y >= 0 && y < sh && x >= 0 && x < sw) ? g[y * sw + x] : (uint8_t)0;
  }
}
// (the copy may run up to 15 bytes past (sh + 4) * (sw + 4): zeros of the padded copy into spare bytes of s_tex, whose
//  size is a multiple of 16)
__device__ inline void load_tex_copy(uint8_t* s_tex, const uint8_t* gpad, int sh, int sw) {
  const int n16 = ((sh + 4) * (sw + 4) + 15) >> 4;
  const uint4* g = reinterpret_cast<const uint4*>(gpad);
  uint4* d = reinterpret_cast<uint4*>(s_tex);
  for (int k = threadIdx.x; k < n16; k += 256) d[k] = g[k];
}

// texture -> LDS with a 2-texel zero border (pitch sw+4).  Border texels are zeroed directly,
// the interior is copied with independent dword loads (textures are 16-byte aligned by
// pack_streak_db; unaligned bases fall back to byte loads).  Caller syncs afterwards.
__device__ inline void load_tex_padded(uint8_t* s_tex, const uint8_t* gtex, int sh, int sw) {
  const int t = threadIdx.x, P = sw + 4;
  for (int k = t; k < 4 * P; k += 256) {               // two rows above, two below
    const int r = k / P, x = k - r * P;
    s_tex[(r < 2 ? r : sh + r) * P + x] = 0;
  }
  for (int k = t; k < 4 * sh; k += 256) {               // two columns left, two right
    const int y = k >> 2, c = k & 3;
    s_tex[(y + 2) * P + (c < 2 ? c : sw + c)] = 0;
  }
  const int nbytes = sh * sw;
  const float inv_sw = 1.0f / (float)sw;
  if ((reinterpret_cast<uintptr_t>(gtex) & 3u) == 0) {
    const uint32_t* g4 = reinterpret_cast<const uint32_t*>(gtex);
    const int nd = nbytes >> 2;
    for (int k = t; k < nd; k += 256) {
      const uint32_t v = g4[k];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int idx = 4 * k + j;
        const int y = (int)(((float)idx + 0.5f) * inv_sw), x = idx - y * sw;
        s_tex[(y + 2) * P + (x + 2)] = (uint8_t)(v >> (8 * j));
      }
    }
    for (int idx = (nd << 2) + t; idx < nbytes; idx += 256) {
      const int y = (int)(((float)idx + 0.5f) * inv_sw), x = idx - y * sw;
      s_tex[(y + 2) * P + (x + 2)] = gtex[idx];
    }
  } else {
    for (int idx = t; idx < nbytes; idx += 256) {
      const int y = (int)(((float)idx + 0.5f) * inv_sw), x = idx - y * sw;
      s_tex[(y + 2) * P + (x + 2)] = gtex[idx];
    }
  }
}

// fixed-point bilinear sample of the padded LDS texture (same arithmetic as rot_sample; valid
// when tile_coords_safe(p) holds)
__device__ inline double lds_rot_sample(const uint8_t* s_tex, const double* s_lut, int P, int sh, int sw, int X0, int Y0,
                                        int2 d) {
  const int X = (X0 + d.x) >> 5, Y = (Y0 + d.y) >> 5;
  int sx = X >> 5, sy = Y >> 5;
  const int fx = X & 31, fy = Y & 31;
  sx = imin(imax(sx, -2), sw);
  sy = imin(imax(sy, -2), sh);
  const uint8_t* q = s_tex + (sy + 2) * P + (sx + 2);
  const double v00 = s_lut[q[0]], v01 = s_lut[q[1]], v10 = s_lut[q[P]], v11 = s_lut[q[P + 1]];
  const double ax_ = (double)(32 - fx), bx_ = (double)fx, ay_ = (double)(32 - fy), by_ = (double)fy;
  // integer-valued weights; the common factor 2^-10 is applied once (exact)
  const double sm = ((v00 * (ay_ * ax_) + v01 * (ay_ * bx_)) + v10 * (by_ * ax_)) + v11 * (by_ * bx_);
  return sm * (1.0 / 1024.0);
}

// two samples at once: all LDS loads of both samples are issued before either is consumed
__device__ inline void lds_rot_sample2(const uint8_t* s_tex, const double* s_lut, int P, int sh, int sw, int XA, int YA, int2 dA,
                                       int XB, int YB, int2 dB, double& outA, double& outB) {
  const int Xa = (XA + dA.x) >> 5, Ya = (YA + dA.y) >> 5, Xb = (XB + dB.x) >> 5, Yb = (YB + dB.y) >> 5;
  const int sxa = imin(imax(Xa >> 5, -2), sw), sya = imin(imax(Ya >> 5, -2), sh);
  const int sxb = imin(imax(Xb >> 5, -2), sw), syb = imin(imax(Yb >> 5, -2), sh);
  const uint8_t* qa = s_tex + (sya + 2) * P + (sxa + 2);
  const uint8_t* qb = s_tex + (syb + 2) * P + (sxb + 2);
  const int a0 = qa[0], a1 = qa[1], a2 = qa[P], a3 = qa[P + 1];
  const int b0 = qb[0], b1 = qb[1], b2 = qb[P], b3 = qb[P + 1];
  const double va0 = s_lut[a0], va1 = s_lut[a1], va2 = s_lut[a2], va3 = s_lut[a3];
  const double vb0 = s_lut[b0], vb1 = s_lut[b1], vb2 = s_lut[b2], vb3 = s_lut[b3];
  const int fxa = Xa & 31, fya = Ya & 31, fxb = Xb & 31, fyb = Yb & 31;
  const double axa = (double)(32 - fxa), bxa = (double)fxa, aya = (double)(32 - fya), bya = (double)fya;
  const double axb = (double)(32 - fxb), bxb = (double)fxb, ayb = (double)(32 - fyb), byb = (double)fyb;
  const double sa = ((va0 * (aya * axa) + va1 * (aya * bxa)) + va2 * (bya * axa)) + va3 * (bya * bxa);
  const double sb = ((vb0 * (ayb * axb) + vb1 * (ayb * bxb)) + vb2 * (byb * axb)) + vb3 * (byb * bxb);
  outA = sa * (1.0 / 1024.0);
  outB = sb * (1.0 / 1024.0);
}

constexpr int GEN_SLICES = 16;
// Big drops (bicubic warp) and the rare resize modes: one thread per output pixel, texels in LDS.
__global__ __launch_bounds__(256) void k_tile_generic(const FrameDesc* frames, int max_drops, const uint8_t* texels,
                                                      const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off,
                                                      const float* ctab, Scratch sc) {
  const int f = blockIdx.y, t = threadIdx.x;
  __shared__ double s_lut[256];
  __shared__ __attribute__((aligned(16))) uint8_t s_tex[TEX_LDS];
  __shared__ int2 s_adbd[NW_MAX];
  const int n_items = sc.counts[f * 8 + 1];
  if ((int)blockIdx.x >= n_items * GEN_SLICES) return;                    // (nearly every frame: nothing here)
  s_lut[t] = (double)t / 255.0;
  // the generic list is a handful of heavy tiles (every pixel a long sequential sum): each is split
  // over GEN_SLICES workgroups by pixel index, or one of them would be the tail of the whole batch
  for (int work = blockIdx.x; work < n_items * GEN_SLICES; work += gridDim.x) {      // grid-stride over (item, slice)
  const int item = work / GEN_SLICES, slice = work - item * GEN_SLICES;
  const int li = sc.list_gen[(int64_t)f * max_drops + item];
  const int64_t gi = (int64_t)f * max_drops + li;
  const DropPlan& p = sc.plan[gi];
  if (p.kind == KIND_EXT) {                               // the caller's tile: copy (clip like generator.py:132,170)
    const double* src = frames[f].ext[li].alpha;
    double* A0e = sc.arena + p.a0_off;
    for (int idx = t + 256 * slice; idx < p.tw * p.th; idx += 256 * GEN_SLICES) A0e[idx] = clip01(src[idx]);
    continue;
  }
  const int sh = tex_h[p.tex], sw = tex_w[p.tex];
  __syncthreads();
  const uint8_t* gtex = texels + tex_off[p.tex];
  const int P = sw + 4;
  // Staging the texture in LDS costs one pass over all its texels; a small Big tile (bicubic: 16 taps
  // per output pixel) touches fewer texels than that, and the 50 textures (~350 KB) live in L2 anyway
  const bool tex_fits = (sh + 4) * P <= TEX_LDS && !(p.kind == KIND_BIG && p.tw * p.th * 12 < sh * sw);
  if (tex_fits) {
    if (sc.tex_pad) load_tex_copy(s_tex, sc.tex_pad + sc.tex_poff[p.tex], sh, sw);
    else load_tex_padded(s_tex, gtex, sh, sw);
  }
  double* A0 = sc.arena + p.a0_off;      // raw tile, pitch tw
  // integer-ratio INTER_AREA (ResizeAreaFast): the per-pixel chain is sequential by definition;
  // keep it short with the LDS fixed-point sampler
  const bool area_fast = tex_fits && p.kind == KIND_ROT && p.rs_mode == RS_AREA_FAST && p.nW <= NW_MAX && tile_coords_safe(p);
  if (area_fast)
    for (int rx = t; rx < p.nW; rx += 256) s_adbd[rx] = make_int2((int)rot_adelta(p, rx), (int)rot_bdelta(p, rx));
  __syncthreads();
  const int n = p.tw * p.th;
  if (area_fast) {
    const int area = p.isx * p.isy, n4 = area & ~3;
    const float scale = 1.0f / (float)area;
    for (int idx = t + 256 * slice; idx < n; idx += 256 * GEN_SLICES) {
      const int dy = idx / p.tw, dx = idx - dy * p.tw;
      double sum = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0;
      int k = 0;
      for (int ky = 0; ky < p.isy; ky++) {
        const int c = dy * p.isy + ky;
        const int ry = p.flip ? (p.nH - 1 - c) : c;
        const int X0 = (int)rot_X0(p, ry), Y0 = (int)rot_Y0(p, ry);
        for (int kx = 0; kx < p.isx; kx++, k++) {
          const double v = lds_rot_sample(s_tex, s_lut, P, sh, sw, X0, Y0, s_adbd[dx * p.isx + kx]);
          if (k >= n4) sum = sum + v;
          else {
            const int m = k & 3;
            if (m == 0) q0 = v;
            else if (m == 1) q1 = v;
            else if (m == 2) q2 = v;
            else sum = sum + (((q0 + q1) + q2) + v);
          }
        }
      }
      A0[dy * p.tw + dx] = clip01(sum * (double)scale);
    }
  } else if (tex_fits) {
    TexLutPad tx{s_tex, s_lut, sh, sw};
    for (int idx = t + 256 * slice; idx < n; idx += 256 * GEN_SLICES) {
      int y = idx / p.tw, x = idx - y * p.tw;
      A0[idx] = raw_tile_pixel(p, tx, ctab, x, y);
    }
  } else {
    TexLut tx{gtex, s_lut, sh, sw};
    for (int idx = t + 256 * slice; idx < n; idx += 256 * GEN_SLICES) {
      int y = idx / p.tw, x = idx - y * p.tw;
      A0[idx] = raw_tile_pixel(p, tx, ctab, x, y);
    }
  }
  }
}


// TexLut whose at4 -- the four neighbours of one texture row the bicubic interior case reads -- is ONE (unaligned) 4-byte
// global load instead of four 1-byte loads: a Big pixel then issues 4 loads, not 16 (the texture bytes come from L2).
struct TexLutWide {
  const uint8_t* t;
  const double* lut;
  int h, w;
  __device__ double at(int64_t y, int64_t x) const { return lut[t[y * w + x]]; }
  __device__ void at4(int64_t y, int64_t x, double v[4]) const {
    uint32_t u;
    __builtin_memcpy(&u, as_global(t) + (y * w + x), 4);
    v[0] = lut[u & 0xffu]; v[1] = lut[(u >> 8) & 0xffu]; v[2] = lut[(u >> 16) & 0xffu]; v[3] = lut[u >> 24];
  }
  __device__ double tap(int64_t y, int64_t x) const {
    if (y < 0 || y >= h || x < 0 || x >= w) return 0.0;
    return at(y, x);
  }
};

// Big drops (cv2.warpPerspective, INTER_CUBIC): ONE THREAD PER OUTPUT PIXEL over the concatenation of
// all Big tiles of the frame (k_lists' pixel prefix).  A Big tile has ~400 pixels, so a workgroup per
// drop was all per-item latency; flattened, every lane has a pixel and the grid is full.  Texels come
// from global memory (the streak DB is L2 resident), the v/255 table and the cubic table from LDS.
__global__ __launch_bounds__(256) void k_tile_big(const FrameDesc* frames, int max_drops, const uint8_t* texels,
                                                  const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off,
                                                  const float* ctab, Scratch sc) {
  const int f = blockIdx.y, t = threadIdx.x;
  __shared__ double s_lut[256];
  __shared__ float s_ctab[128];
  const int n_big = sc.counts[f * 8 + 5], total = sc.counts[f * 8 + 6];
  if (total == 0) return;                           // (r06: with k_tile_rows taking the Big tiles whose texture fits its LDS, nearly every frame)
  s_lut[t] = (double)t / 255.0;
  if (t < 128) s_ctab[t] = ctab[t];
  const int32_t* lbig = sc.list_big + (int64_t)f * max_drops;
  const int32_t* boff = sc.big_off + (int64_t)f * max_drops + f;
  const int lane = t & 63;
  __syncthreads();                                  // tables staged; no barrier below
  PH_DECL
  for (int pb = blockIdx.x * 256; pb < total; pb += gridDim.x * 256) {
    // The item of a wave's first pixel: the last j with boff[j] <= pix0, by a 64-ary search over the pixel prefix -- every
    // lane probes one position, a ballot narrows [lo, hi) to one step: two rounds of independent loads for up to 4096 items
    // (r04: one thread's binary search between two block barriers was a third of this kernel's wave time).
    const int pix0 = pb + (t & ~63);
    if (pix0 >= total) continue;                    // (wave-uniform)
    int lo = 0, hi = n_big;
    while (hi - lo > 1) {
      const int step = (hi - lo + 63) >> 6, idx = lo + lane * step;
      const bool le = idx < hi && boff[idx] <= pix0;
      const int cnt = __popcll(__ballot(le));       // >= 1: boff[lo] <= pix0 holds throughout
      lo += (cnt - 1) * step;
      hi = imin(hi, lo + step);
    }
    PH(0)                                           // search
    const int pix = pb + t;
    if (pix >= total) continue;
    int j = lo;
    while (boff[j + 1] <= pix) j++;
    const int64_t gi = (int64_t)f * max_drops + lbig[j];
    const DropPlan& p = sc.plan[gi];
    const int local = pix - boff[j];
    const int y = local / p.tw, x = local - y * p.tw;     // (r04: a reciprocal-multiply with a fix-up in place of this division measured 9 % SLOWER)
    TexLutWide tx{texels + tex_off[p.tex], s_lut, tex_h[p.tex], tex_w[p.tex]};
    PH(1)                                           // plan fields
    sc.arena[p.a0_off + local] = warp_big_pixel(p, tx, s_ctab, x, y);
    PH(2)                                           // the pixel
  }
  PH_FLUSH(1)
}

__global__ __launch_bounds__(256) void k_tile(const FrameDesc* frames, int max_drops, const uint8_t* texels,
                                              const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off,
                                              Scratch sc) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __shared__ DropPlan sp;
  __shared__ double s_lut[256];
  __shared__ __attribute__((aligned(16))) uint8_t s_tex[TEX_LDS];
  __shared__ int2 s_adbd[NW_MAX];
  __shared__ AreaSpan s_ax[TW_MAX];
  __shared__ double s_buf[BUF_MAX];
  __shared__ double s_can[4][CAN_W];
  __shared__ int4 s_row[4][ROWS_S];
  const int n_items = *sc.rot_total;                                       // (r06: ONE list for the batch -- with k_tile_rows in front a frame has a handful of these tiles or none)
  if ((int)blockIdx.x >= n_items) return;
  s_lut[t] = (double)t / 255.0;
  PH_DECL
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {      // grid-stride over the rot-fast list
  const int64_t gi = sc.list_rot[item];
  __syncthreads();
  PH(7)
  {
    const int32_t* src = reinterpret_cast<const int32_t*>(&sc.plan[gi]);
    int32_t* dst = reinterpret_cast<int32_t*>(&sp);
    for (int k = t; k < (int)(sizeof(DropPlan) / 4); k += 256) dst[k] = src[k];
  }
  __syncthreads();
  const DropPlan& p = sp;
  const int sh = tex_h[p.tex], sw = tex_w[p.tex];
  const uint8_t* gtex = texels + tex_off[p.tex];
  const int P = sw + 4;
  if (sc.tex_pad) load_tex_copy(s_tex, sc.tex_pad + sc.tex_poff[p.tex], sh, sw);
  else load_tex_padded(s_tex, gtex, sh, sw);
  double* A0 = sc.arena + p.a0_off;
  const int tw = p.tw, th = p.th;
  const double sy_scale = p.scale_y;
  for (int rx = t; rx < p.nW; rx += 256) s_adbd[rx] = make_int2((int)rot_adelta(p, rx), (int)rot_bdelta(p, rx));
  for (int dx = t; dx < tw; dx += 256) s_ax[dx] = area_span(p.nW, p.scale_x, dx);
  __syncthreads();
  PH(0)                                             // plan + texture + per-column tables staged
  const RowGeom geom = row_geom(p, sh, sw);
  const int pitch = imin(imax(tile_pitch(p, sh, sw), 1), CAN_W);
  const int Rw = imax(imin(ROWS_W, CAN_W / pitch), 1);       // canvas rows a wave stages at a time
  const float inv_pitch = 1.0f / (float)pitch;
  double* can = s_can[wave];
  int4* rowp = s_row[wave];
  if (p.rs_mode == RS_AREA_FAST) {
    // cv2's ResizeAreaFast: every output pixel is ONE sequential chain over its isx*isy source
    // samples (k = ky*isx + kx, groups of four pre-added).  The chains of a destination row are
    // split over up to four waves; a wave stages only the canvas columns of its own pixels and
    // lane l carries the chain of its l-th pixel.
    const int isx = p.isx, isy = p.isy, area = isx * isy, n4 = area & ~3;
    const float scale = 1.0f / (float)area;
    const int nsplit = th >= 3 ? 1 : (th == 2 ? 2 : 4);        // waves per destination row
    const int rows_par = 4 / nsplit;
    for (int dyb = 0; dyb < th; dyb += rows_par) {
      const int dy = dyb + wave / nsplit, part = wave % nsplit;
      const int dx0 = (part * tw) / nsplit, dx1 = ((part + 1) * tw) / nsplit;
      if (dy >= th || dx1 <= dx0) continue;
      const int colA = dx0 * isx, colB = dx1 * isx;
      const int pitch2 = imax(imin(pitch, colB - colA), 1);
      const int Rw2 = imax(imin(ROWS_W, CAN_W / pitch2), 1);
      const float inv_pitch2 = 1.0f / (float)pitch2;
      double sum = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0;
      int k = 0;
      for (int c0 = dy * isy; c0 < (dy + 1) * isy; c0 += Rw2) {
        const int nr = imin(Rw2, (dy + 1) * isy - c0);
        if (lane < nr) {
          const int c = c0 + lane;
          const int ry = p.flip ? (p.nH - 1 - c) : c;
          const int X0 = (int)rot_X0(p, ry), Y0 = (int)rot_Y0(p, ry);
          int xa, n;
          row_interval(p, geom, X0, Y0, xa, n);
          const int xa2 = imax(xa, colA), xe2 = imin(xa + n, colB);
          rowp[lane] = make_int4(X0, Y0, xa2, imax(imin(xe2 - xa2, pitch2), 0));
        }
        wave_lds_sync();
        const int nidx = nr * pitch2;
        for (int idx = lane; idx < nidx; idx += 128) {
          const int ia = idx, ib = imin(idx + 64, nidx - 1);
          const int ra = (int)(((float)ia + 0.5f) * inv_pitch2), xa = ia - ra * pitch2;
          const int rb = (int)(((float)ib + 0.5f) * inv_pitch2), xb = ib - rb * pitch2;
          const int4 rwa = rowp[ra], rwb = rowp[rb];
          const bool oka = xa < rwa.w, okb = (idx + 64 < nidx) && xb < rwb.w;
          const int2 da = s_adbd[rwa.z + (oka ? xa : 0)], db = s_adbd[rwb.z + (okb ? xb : 0)];
          double va, vb;
          lds_rot_sample2(s_tex, s_lut, P, sh, sw, rwa.x, rwa.y, da, rwb.x, rwb.y, db, va, vb);
          if (oka) can[ia] = va;
          if (okb) can[ib] = vb;
        }
        wave_lds_sync();
        const bool mine = lane < dx1 - dx0;
        const int colL = (dx0 + lane) * isx;               // this lane's pixel: canvas columns colL .. colL + isx - 1 of every row
        for (int r = 0; r < nr; r++) {
          const int4 rw = rowp[r];
          const double* row = can + r * pitch2 - rw.z;
          const int vlo = mine ? rw.z : (1 << 30), vhi = rw.z + rw.w;      // staged (possibly non-zero) columns; everything else is an exact zero
          auto at = [&](int col) { return (col >= vlo && col < vhi) ? row[col] : 0.0; };
          // k (wave-uniform) counts the chain's samples across rows: a group of four that lies inside one row is taken at once
          // (r06: one sample per iteration with a four-way branch on k & 3 made a 107 x 107 block -- a 3-pixel-wide tile of a
          // 321-column canvas -- 0.3 ms of one lane's dependent iterations: the whole kernel's time)
          int kx = 0;
          while (kx < isx) {
            if ((k & 3) == 0 && kx + 3 < isx && k + 3 < n4) {
              const double v0 = at(colL + kx), v1 = at(colL + kx + 1), v2 = at(colL + kx + 2), v3 = at(colL + kx + 3);
              sum = sum + (((v0 + v1) + v2) + v3);
              kx += 4;
              k += 4;
              continue;
            }
            const double v = at(colL + kx);
            if (k >= n4) sum = sum + v;
            else {
              const int m = k & 3;
              if (m == 0) q0 = v;
              else if (m == 1) q1 = v;
              else if (m == 2) q2 = v;
              else sum = sum + (((q0 + q1) + q2) + v);
            }
            kx++;
            k++;
          }
        }
        wave_lds_sync();
      }
      if (lane < dx1 - dx0) A0[dy * tw + dx0 + lane] = clip01(sum * (double)scale);
    }
    continue;
  }
  // very wide tiles: the row sums of one destination row must fit s_buf, so the destination columns
  // are taken in chunks of twc (a handful of drops per frame; their canvas rows are sampled once per chunk)
  const int twc_max = imax(imin(tw, BUF_MAX / ((int)ceil(sy_scale) + 4)), 1);
  for (int dxa = 0; dxa < tw; dxa += twc_max) {
  const int twc = imin(twc_max, tw - dxa);
  const float inv_twc = 1.0f / (float)twc;
  int k_dy = (int)(((double)(BUF_MAX / twc) - 4.0) / sy_scale);      // (r06: was - 3.0; a group could be one row longer than s_buf)
  if (k_dy < 1) k_dy = 1;
  for (int dy0 = 0; dy0 < th; dy0 += k_dy) {
    const int dy1 = imin(dy0 + k_dy, th);
    const int lo = imax((int)floor((double)dy0 * sy_scale) - 1, 0);
    const int hi = imin((int)floor((double)dy1 * sy_scale) + 1, p.nH - 1);
    // ---- canvas rows lo..hi: every wave takes an EQUAL contiguous share, Rw rows at a time, no block barrier needed
    //      (r04: with chunks of Rw rows dealt round-robin the waves finished a group up to a chunk apart -- 7 chunks for
    //      4 waves -- and the phase clocks showed 35 % of the kernel's wave time spent at the barrier below) ----
    const int share = (hi - lo + 4) >> 2;
    const int w_lo = lo + wave * share, w_hi = imin(hi, w_lo + share - 1);
    // the rows' sampling intervals, up to ROWS_S rows at a time (one lane each: in chunks of Rw ~ 12 rows three quarters of
    // this float64 arithmetic ran on a fifth of the lanes -- 13 % of the kernel's wave time)
    for (int q0 = w_lo; q0 <= w_hi; q0 += ROWS_S) {
    const int nq = imin(ROWS_S, w_hi - q0 + 1);
    if (lane < nq) {
      const int c = q0 + lane;
      const int ry = p.flip ? (p.nH - 1 - c) : c;
      const int X0 = (int)rot_X0(p, ry), Y0 = (int)rot_Y0(p, ry);
      int xa, n;
      row_interval(p, geom, X0, Y0, xa, n);
      s_row[wave][lane] = make_int4(X0, Y0, xa, imin(n, pitch));
    }
    wave_lds_sync();
    for (int r0 = q0; r0 < q0 + nq; r0 += Rw) {
      const int nr = imin(Rw, q0 + nq - r0);
      const int4* rowp = s_row[wave] + (r0 - q0);
      PH(1)                                         // row intervals
      // ---- 1a: bilinear samples of the rotated texture, lanes flattened over (row, column) ----
      const int nidx = nr * pitch;
      for (int idx = lane; idx < nidx; idx += 128) {
        const int ia = idx, ib = imin(idx + 64, nidx - 1);
        const int ra = (int)(((float)ia + 0.5f) * inv_pitch), xa = ia - ra * pitch;
        const int rb = (int)(((float)ib + 0.5f) * inv_pitch), xb = ib - rb * pitch;
        const int4 rwa = rowp[ra], rwb = rowp[rb];
        const bool oka = xa < rwa.w, okb = (idx + 64 < nidx) && xb < rwb.w;
        const int2 da = s_adbd[rwa.z + (oka ? xa : 0)], db = s_adbd[rwb.z + (okb ? xb : 0)];
        double va, vb;
        lds_rot_sample2(s_tex, s_lut, P, sh, sw, rwa.x, rwa.y, da, rwb.x, rwb.y, db, va, vb);
        if (oka) can[ia] = va;
        if (okb) can[ib] = vb;
      }
      wave_lds_sync();
      PH(2)                                         // samples
      // ---- 1b: horizontal folds, one lane per (row, destination column) ----
      const int items = nr * twc;
      for (int it = lane; it < items; it += 64) {
        const int r = (int)(((float)it + 0.5f) * inv_twc), dxl = it - r * twc, dx = dxa + dxl;
        const AreaSpan ax = s_ax[dx];
        const int4 rw = rowp[r];
        const int xlo = rw.z, xhi = rw.z + rw.w - 1;          // staged (possibly non-zero) columns
        const double* row = can + r * pitch - rw.z;
        double b = 0.0;
        // resizeArea_ order: left partial cell, full cells, right partial cell; columns outside
        // [xlo, xhi] hold exact zeros and are skipped
        if (ax.has_l && ax.s1 - 1 >= xlo && ax.s1 - 1 <= xhi) b = b + row[ax.s1 - 1] * (double)ax.a_l;
        {
          const int m0 = imax(ax.s1, xlo), m1 = imin(ax.s2 - 1, xhi);
          const double am = (double)ax.a_m;
          int sx = m0;
          for (; sx + 3 <= m1; sx += 4) {
            const double v0 = row[sx], v1 = row[sx + 1], v2 = row[sx + 2], v3 = row[sx + 3];
            b = b + v0 * am;
            b = b + v1 * am;
            b = b + v2 * am;
            b = b + v3 * am;
          }
          for (; sx <= m1; sx++) b = b + row[sx] * am;
        }
        if (ax.has_r && ax.s2 >= xlo && ax.s2 <= xhi) b = b + row[ax.s2] * (double)ax.a_r;
        s_buf[(r0 - lo + r) * twc + dxl] = b;
      }
      wave_lds_sync();
      PH(3)                                         // horizontal folds
    }
    }
    __syncthreads();
    PH(4)                                           // waiting for the other waves
    // ---- 2: vertical folds ----
    const int npx = (dy1 - dy0) * twc;
    for (int it = t; it < npx; it += 256) {
      const int r = it / twc, dxl = it - r * twc, dx = dxa + dxl;
      const int dy = dy0 + r;
      const AreaSpan ay = area_span(p.nH, sy_scale, dy);
      double acc = 0.0;
      bool first = true;
      if (ay.has_l) {
        acc = (double)ay.a_l * s_buf[(ay.s1 - 1 - lo) * twc + dxl];
        first = false;
      }
      for (int sy = ay.s1; sy < ay.s2; sy++) {
        double v = (double)ay.a_m * s_buf[(sy - lo) * twc + dxl];
        acc = first ? v : acc + v;
        first = false;
      }
      if (ay.has_r) {
        double v = (double)ay.a_r * s_buf[(ay.s2 - lo) * twc + dxl];
        acc = first ? v : acc + v;
      }
      A0[dy * tw + dx] = clip01(acc);
    }
    __syncthreads();
    PH(5)                                           // vertical folds + store
  }
  }
  }
  PH_FLUSH(0)
}

// ---------------------------------------------------------------------------
// k_tile_rows (round 6): rotate_bound -> flip -> resize(INTER_AREA) (generator.py:163-170) by row walks
// ---------------------------------------------------------------------------
// k_tile gave a workgroup to every tile and staged plan + texture for each (a fifth of its wave time), passed every sample
// through LDS twice (sample -> horizontal fold -> vertical fold) and met at block barriers.  Here:
//   * the batch's tiles are one list bucketed by texture (k_lists' histogram, k_rows_scatter); a workgroup of 16 waves --
//     one per CU, the LDS holds ONE texture for all of them -- takes a contiguous share of it (split by estimated cost), so
//     a texture is staged a couple of times per workgroup instead of once per tile;
//   * a WAVE renders a tile: no block barrier inside a tile.  Waves pull the share's tiles one by one (an LDS counter); the
//     workgroup only meets when a wave pulls a tile of another texture than the resident one (the list is sorted, so every
//     wave soon does): barrier, stage the texture of the lowest pending tile, barrier;
//   * a LANE owns a canvas row and walks its columns left to right (rr_device.h): row terms X0 / Y0 and the interval of
//     columns that can touch the texture live in the lane's registers, the per-column terms (adelta, bdelta, the fold
//     weights) in a 16-byte table entry, the 2 x 2 texels of a sample are two adjacent elements of the "pair texture"
//     (texel | texel below << 8), the horizontal fold runs in a register.  Two columns per iteration: both columns' table
//     entries, then both texel pairs, then all eight table values are requested before any is used;
//   * cell sums -> LDS (one double per row and destination column) -> the vertical fold, a lane per output pixel.
// Every sum folds in resizeArea_'s order: the tile is bit-identical to raw_tile_pixel (the CPU tier runs the same column table
// and walk rule against it: tests/test_tile_rows_host.py; test_raw_tile_dedup_is_invisible, test_known_answers and the oracle
// tests on the GPU).
// The batch-wide list cut into n_shares pieces of equal estimated cost: bounds[q] = first tile of share q (k_tile_rows'
// workgroups take shares off a counter).  Buckets are weighted by the summed cost of their tiles (k_lists), a bucket's
// tiles count at their average.  One workgroup.
__global__ __launch_bounds__(1024) void k_rows_shares(int n_shares, Scratch sc) {
  __shared__ int64_t cum_cost[RW_TEX_MAX + 1];
  __shared__ int32_t cum_cnt[RW_TEX_MAX + 1];
  const int t = threadIdx.x, lane = t & 63, n_tex = sc.n_buckets;      // (buckets: textures, twice over when Big tiles ride along)
  if (t < 64) {
    const int per = (n_tex + 63) >> 6;
    int64_t c_cost = 0;
    int c_cnt = 0;
    for (int k = 0; k < per; k++) {
      const int tt = lane * per + k;
      if (tt < n_tex) {
        c_cnt += sc.rows_hist[tt];
        c_cost += (int64_t)sc.rows_cost[tt];
      }
    }
    int64_t i_cost = c_cost;
    int i_cnt = c_cnt;
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
      const int64_t vc = (int64_t)__shfl_up((long long)i_cost, ofs);
      const int vn = __shfl_up(i_cnt, ofs);
      if (lane >= ofs) { i_cost += vc; i_cnt += vn; }
    }
    int64_t e_cost = i_cost - c_cost;
    int e_cnt = i_cnt - c_cnt;
    for (int k = 0; k < per; k++) {
      const int tt = lane * per + k;
      if (tt < n_tex) {
        cum_cost[tt] = e_cost;
        cum_cnt[tt] = e_cnt;
        e_cnt += sc.rows_hist[tt];
        e_cost += (int64_t)sc.rows_cost[tt];
      }
    }
    if (lane == 63) { cum_cost[n_tex] = i_cost; cum_cnt[n_tex] = i_cnt; }
  }
  __syncthreads();
  const int64_t total = cum_cost[n_tex];
  for (int w = t; w <= n_shares; w += 1024) {
    int bound = cum_cnt[n_tex];
    if (w < n_shares && total > 0) {
      const int64_t target = total / (int64_t)n_shares * w + (total % (int64_t)n_shares) * w / (int64_t)n_shares;
      int lo = 0, hi = n_tex;                    // the last bucket whose first tile starts at or before `target`
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cum_cost[mid] <= target) lo = mid; else hi = mid;
      }
      const int cn = cum_cnt[lo + 1] - cum_cnt[lo];
      const int64_t bucket = cum_cost[lo + 1] - cum_cost[lo];
      const int64_t into = bucket > 0 ? (target - cum_cost[lo]) * cn / bucket : 0;
      bound = cum_cnt[lo] + (into < (int64_t)cn ? (int)into : cn);
    }
    sc.rows_bounds[w] = bound;
  }
}

struct RowsShared {                   // the ONE shared variable of k_tile_rows
  double lut[256];                    // v / 255.0; first, i.e. at LDS address 0 (checked at run time): the sampler's byte * 8 IS the address
  uint8_t pair[RW_PAIR_BYTES];        // the resident texture
  RowsWave w[RW_WAVES];
  int pend[RW_WAVES], ptex[RW_WAVES];
  int next, first, end, cur;
};
typedef const double __attribute__((address_space(3)))* lds_cdouble;
// (byte B of v) << 3 in one instruction
template <int B>
__device__ inline uint32_t byte_x8(uint32_t v, uint32_t three) {
  uint32_t r;
  if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(v));
  else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(v));
  else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(v));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(v));
  return r;
}
__device__ inline int med3i(int x, int lo, int hi) {      // clamp(x, lo, hi), lo <= hi
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
}
// Bilinear sample of warpAffine (rot_sample's arithmetic) WITHOUT its division by 1024 (the column table's weights carry
// it).  Xb, Yb: the 10-bit fixed-point source coordinates + 2048 (two texels of zero border, so that the clamped
// coordinates index the pair texture directly).  The two halves: addresses + texel fetch, then table look-ups + blend.
struct RowsSample {
  uint32_t ta, tb;                    // pair elements (sy, sx), (sy, sx + 1): texel | texel below << 8
};
__device__ inline RowsSample rows_fetch(const uint8_t* s_pair, int Xb, int Yb, int P2, int sh2, int sw2) {
  const int sx = med3i(Xb >> 10, 0, sw2), sy = med3i(Yb >> 10, 0, sh2);
  const uint16_t* q = reinterpret_cast<const uint16_t*>(s_pair + (sy * P2 + (sx << 1)));
  return RowsSample{q[0], q[1]};
}
__device__ inline double rows_blend(const RowsSample& t, int Xb, int Yb, uint32_t three) {
  const double v00 = *(lds_cdouble)(uintptr_t)byte_x8<0>(t.ta, three), v10 = *(lds_cdouble)(uintptr_t)byte_x8<1>(t.ta, three);
  const double v01 = *(lds_cdouble)(uintptr_t)byte_x8<0>(t.tb, three), v11 = *(lds_cdouble)(uintptr_t)byte_x8<1>(t.tb, three);
  const double bx_ = (double)((Xb >> 5) & 31), by_ = (double)((Yb >> 5) & 31);
  const double ax_ = 32.0 - bx_, ay_ = 32.0 - by_;
  return ((v00 * (ay_ * ax_) + v01 * (ay_ * bx_)) + v10 * (by_ * ax_)) + v11 * (by_ * bx_);
}

// one tile, by one wave
__device__ inline void rows_tile(const DropPlan& p, int sh, int sw, RowsShared& S, RowsWave& W, double* arena, uint32_t three PH_PARAMS) {
  const int lane = threadIdx.x & 63;
  const int tw = p.tw, th = p.th, nW = p.nW, nH = p.nH;
  const int P2 = pair_pitch(sw) * 2, sh2 = sh + 2, sw2 = sw + 2;
  const uint8_t* s_pair = S.pair;
  double* A0 = arena + p.a0_off;
  // ---- column table ----
  for (int x = lane; x < nW; x += 64) {
    W.col[x] = ColEnt{(int32_t)rot_adelta(p, x), (int32_t)rot_bdelta(p, x), 0u, 0u};
    W.cell[x] = 0;
  }
  wave_lds_sync();
  if (lane < tw) coltab_cell_pass1(p, lane, W.col, W.cell, W.cfirst, W.clast);
  wave_lds_sync();
  if (lane < tw) coltab_cell_pass2(p, lane, W.col, W.cell);
  // the vertical axis: lane dy keeps area_span(dy) -- a pass' folds fetch theirs with four lane shuffles instead of
  // evaluating it (three float64 divisions) per pass
  uint32_t vq0, vq1, vq2, vq3;
  {
    const AreaSpan ay = area_span(nH, p.scale_y, imin(lane, th - 1));
    vq0 = (uint32_t)ay.s1 | ((uint32_t)ay.s2 << 15) | ((uint32_t)(ay.has_l ? 1 : 0) << 30) | ((uint32_t)(ay.has_r ? 1 : 0) << 31);
    vq1 = f32_bits(ay.a_l);
    vq2 = f32_bits(ay.a_m);
    vq3 = f32_bits(ay.a_r);
  }
  wave_lds_sync();
  PH(1)                                               // column table
  const RowGeom geom = row_geom(p, sh, sw);
  int R, NS;
  rows_pass_shape(tw, RW_BUF, R, NS);
  const int seg = (int)(((float)lane + 0.5f) / (float)R), r = lane - seg * R;     // this lane's segment and row of the pass
  const int colA = W.cfirst[0], colB = W.clast[tw - 1];
  const float inv_tw = 1.0f / (float)tw;
  double* const carry = W.buf + R * tw;               // accumulators of the destination row that straddles a pass end
  for (int R0 = 0; R0 < nH; R0 += R) {
    const int R1 = imin(R0 + R, nH) - 1, nrows = R1 - R0 + 1;
    for (int k = lane; k < nrows * tw; k += 64) W.buf[k] = 0.0;
    // ---- the pass: lane = (canvas row R0 + r, segment `seg` of the cells the row touches) ----
    int X0 = 0, Y0 = 0, xq = 0, left = 0, dA = 0, dB = 0;
    if (r < nrows && seg < NS) {
      const int c = R0 + r;
      const int ry = p.flip ? (nH - 1 - c) : c;
      X0 = (int)rot_X0(p, ry);
      Y0 = (int)rot_Y0(p, ry);
      int xa, n;
      row_interval(p, geom, X0, Y0, xa, n);
      const int xs0 = imax(xa, colA), xe0 = imin(xa + n - 1, colB);
      if (xs0 <= xe0) {
        const int dlo = W.cell[xs0], dhi = imin(W.cell[xe0] + 1, tw - 1);   // (+ 1: the row's last column may also start the next cell)
        rows_segment(dlo, dhi, NS, seg, dA, dB);
        if (dA <= dhi) {
          xq = imax(xs0, (int)W.cfirst[dA]);
          left = imax(imin(xe0, (int)W.clast[dB - 1]) - xq + 1, 0);
        }
      }
      X0 += 2048;
      Y0 += 2048;
    }
    wave_lds_sync();
    double* out = W.buf + r * tw;
    double* const lane_end = out + dB;
    const bool any_col = left > 0;
    double b = 0.0;
    if (any_col) {
      const int d0 = W.cell[xq];
      if (d0 < dA) {                                 // the segment's first column still ends the cell before it (the lane to the
        const ColEnt e = W.col[xq];                  //  left folds that): only its left-partial role is ours
        const int X = X0 + e.ad, Y = Y0 + e.bd;
        b = rows_blend(rows_fetch(s_pair, X, Y, P2, sh2, sw2), X, Y, three) * (double)bits_f32(e.w2);
        out += dA;
        xq++;
        left--;
      } else {
        out += d0;
      }
    }
    PH(2)                                             // pass set-up: clear, row terms, intervals
    // Two columns per iteration.  The table entries are read whether or not the lane still has columns (a lane that is
    // done reads on into whatever follows -- always inside the workgroup's LDS, always a finite sample -- and folds it
    // with weight 0); each entry as ONE 16-byte read (left to itself the compiler reads 12 bytes and fetches w2 inside the
    // flush branch: a second LDS round trip on the critical path of nearly every iteration).
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const volatile u32x4_t __attribute__((address_space(3)))* cp = (const volatile u32x4_t __attribute__((address_space(3)))*)(W.col + xq);
    while (__ballot(left > 0) != 0ull) {
      const u32x4_t qa = cp[0], qb = cp[1];
      const ColEnt ea{(int32_t)qa.x, (int32_t)qa.y, qa.z, qa.w}, eb{(int32_t)qb.x, (int32_t)qb.y, qb.z, qb.w};
      const int XA = X0 + ea.ad, YA = Y0 + ea.bd, XB = X0 + eb.ad, YB = Y0 + eb.bd;
      const RowsSample ta = rows_fetch(s_pair, XA, YA, P2, sh2, sw2), tb = rows_fetch(s_pair, XB, YB, P2, sh2, sw2);
      const double sa = rows_blend(ta, XA, YA, three), sb = rows_blend(tb, XB, YB, three);
      const uint32_t wa = left > 0 ? ea.w1 : 0u, wb = left > 1 ? eb.w1 : 0u;
      b = b + sa * (double)bits_f32(wa & 0x7fffffffu);
      if ((int32_t)wa < 0) {
        *out++ = b;
        b = sa * (double)bits_f32(ea.w2);
      }
      b = b + sb * (double)bits_f32(wb & 0x7fffffffu);
      if ((int32_t)wb < 0) {
        *out++ = b;
        b = sb * (double)bits_f32(eb.w2);
      }
      cp += 2;
      left -= 2;
      PH_COUNT(6)
    }
    if (any_col && out < lane_end) *out = b;          // the columns ended inside a cell
    PH(3)                                             // the walk
    wave_lds_sync();
    // ---- vertical folds of the destination rows that read rows R0 .. R1: a lane per (destination row, column) ----
    const int dyG = imax((int)floor((double)R0 * p.inv_sy) - 1, 0);            // the first candidate; rows before it ended before R0
    const int dyE = imin((int)floor((double)(R1 + 1) * p.inv_sy) + 2, th);     // one past the last candidate
    const int n_out = (dyE - dyG) * tw;
    for (int it0 = 0; it0 < n_out; it0 += 64) {
      const int it = it0 + lane;
      const bool valid = it < n_out;
      const int dq = (int)(((float)it + 0.5f) * inv_tw), dx = it - dq * tw, dy = valid ? dyG + dq : 0;
      AreaSpan ay;
      {
        const uint32_t q0 = (uint32_t)__shfl((int)vq0, dy);
        ay.s1 = (int16_t)(q0 & 0x7fffu);
        ay.s2 = (int16_t)((q0 >> 15) & 0x7fffu);
        ay.has_l = (int16_t)((q0 >> 30) & 1u);
        ay.has_r = (int16_t)(q0 >> 31);
        ay.a_l = bits_f32((uint32_t)__shfl((int)vq1, dy));
        ay.a_m = bits_f32((uint32_t)__shfl((int)vq2, dy));
        ay.a_r = bits_f32((uint32_t)__shfl((int)vq3, dy));
      }
      int fr, lr;
      vfold_rows(ay, fr, lr);
      if (!valid || lr < R0 || fr > R1) continue;
      double acc = 0.0;
      bool first = true;
      if (fr < R0) {                                 // begun in an earlier pass
        acc = carry[dx];
        first = false;
      }
      vfold_part(ay, R0, R1, acc, first, [&](int row) { return W.buf[(row - R0) * tw + dx]; });
      if (lr <= R1) A0[dy * tw + dx] = clip01(acc);
      else carry[dx] = acc;
    }
    wave_lds_sync();
    PH(4)                                             // vertical folds + store
  }
}

// ---- Big drops in the same kernel: cv2.warpPerspective(INTER_CUBIC) (warp_big_pixel), a wave per tile, a lane per pixel ----
// The bucket's texture is resident with its 2-texel zero border (k_pad_textures' copy, pitch sw + 4) behind 16 zero bytes
// at S.pair: a row's four taps are the 4 bytes at the column clamped to [-4, sw] and the row clamped to [-2, sh + 1] --
// whatever falls outside the texture is a zero of the border (or of the neighbouring row's border: the rows are
// contiguous), which is what the reference's BORDER_CONSTANT taps are.  They are fetched as the two ALIGNED dwords around
// them and a v_alignbit: an unaligned ds_read_b32 is legal on gfx950 and three times slower (measured: the Big tiles of
// 512 frames 3.0 ms with it, 1.6 ms with the aligned pair; the table look-ups are 0.7 ms of that, the texel reads 0.3).  The 16 products are those of
// warp_big_pixel; they are summed row by row for interior windows and tap by tap for windows over the border, as there.
// The plan is wave-uniform (scalar registers): no search for a pixel's tile, no per-lane plan loads, no divisions by tw
// beyond a float multiply with an exact fix-up (k_tile_big: one thread per pixel of the frame's concatenated tiles).
constexpr int BIG_LDS_LEAD = 16;      // zero bytes in front of the padded texture
#ifndef RR_BIG_NP
#define RR_BIG_NP 1
#endif
#ifndef RR_BIG_COST
#define RR_BIG_COST 2          // a pass of 64 Big pixels in walk iterations of the rotate tiles (the split of the list into shares)
#endif
__device__ inline int div_by_f(int n, int d, float inv_d) {      // n / d for 0 <= n < 2^24, d >= 1
  int q = (int)((float)n * inv_d);
  const int r = n - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}
__device__ inline void big_row_products(uint32_t u, float cyi, const float cx[4], uint32_t three, double pr[4]) {
  const double v0 = *(lds_cdouble)(uintptr_t)byte_x8<0>(u, three), v1 = *(lds_cdouble)(uintptr_t)byte_x8<1>(u, three);
  const double v2 = *(lds_cdouble)(uintptr_t)byte_x8<2>(u, three), v3 = *(lds_cdouble)(uintptr_t)byte_x8<3>(u, three);
  const float w0 = cyi * cx[0], w1 = cyi * cx[1], w2 = cyi * cx[2], w3 = cyi * cx[3];
  pr[0] = v0 * (double)w0;
  pr[1] = v1 * (double)w1;
  pr[2] = v2 * (double)w2;
  pr[3] = v3 * (double)w3;
}
template <int NP>                    // pixels per lane and pass: their dependent chains (division, LDS round trips, 16-term sums) interleave
__device__ inline void big_tile(const DropPlan& p, int sh, int sw, const uint8_t* s_pad, const float* s_ctab, double* arena, uint32_t three PH_PARAMS) {
  const int lane = threadIdx.x & 63;
  const int tw = p.tw, n = tw * p.th, bw0 = p.bw0, P = sw + 4;
  const float inv_bw = 1.0f / (float)bw0;
  const uint8_t* tex00 = s_pad + BIG_LDS_LEAD + 2 * P + 2;      // texel (0, 0)
  double* A0 = arena + p.a0_off;
  const int width1 = imax(sw - 3, 0), height1 = imax(sh - 3, 0);
  // pixel it = it0 + 64 k + lane of the tile, row-major: (y, x) of the first one by a division, the following ones (64 apart)
  // incrementally.  Lanes past the tile's end run on (rows >= th: any coordinates clamp into the resident texture) and store nothing.
  const int dq = 64 / tw, dr = 64 - dq * tw;
  int y = div_by_f(lane, tw, 1.0f / (float)tw), x = lane - y * tw;
  for (int it0 = 0; it0 < n; it0 += 64 * NP) {
    BigCoord c[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int bx = tw <= bw0 ? 0 : div_by_f(x, bw0, inv_bw) * bw0;
      c[k] = warp_big_coord(p, bx, x, y);
      x += dr;
      y += dq;
      if (x >= tw) { x -= tw; y += 1; }
    }
    uint32_t u[NP][4];
    float cx[NP][4], cy[NP][4];
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const uint8_t* col = tex00 + med3i(c[k].sx, -4, sw);
#pragma unroll
      for (int i = 0; i < 4; i++) {                   // the two aligned dwords around the window, shifted into place
        const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)(col + med3i(c[k].sy + i, -2, sh + 1) * P);
        const uint32_t* q = (const uint32_t*)(__attribute__((address_space(3))) uint32_t*)(uintptr_t)(a & ~3u);
        const uint32_t lo = q[0], hi = q[1];
        u[k][i] = __builtin_amdgcn_alignbit(hi, lo, (a & 3u) * 8u);
      }
      const float4 a = *reinterpret_cast<const float4*>(s_ctab + c[k].fx * 4), b = *reinterpret_cast<const float4*>(s_ctab + c[k].fy * 4);
      cx[k][0] = a.x; cx[k][1] = a.y; cx[k][2] = a.z; cx[k][3] = a.w;
      cy[k][0] = b.x; cy[k][1] = b.y; cy[k][2] = b.z; cy[k][3] = b.w;
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
      double pr[4][4];
#pragma unroll
      for (int i = 0; i < 4; i++) big_row_products(u[k][i], cy[k][i], cx[k], three, pr[i]);
      const bool interior = c[k].sx >= 0 && c[k].sx < width1 && c[k].sy >= 0 && c[k].sy < height1;
      double sum = 0.0;
      if (interior) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          double r = (pr[i][0] + pr[i][1]) + pr[i][2];
          r = r + pr[i][3];
          sum = (i == 0) ? r : sum + r;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) sum = sum + pr[i][j];
      }
      const int it = it0 + 64 * k + lane;
      if (it < n) A0[it] = clip01(sum);
    }
  }
  PH(7)                                               // a Big tile
}

__global__ __launch_bounds__(1024) void k_tile_rows(int n_shares, const int32_t* tex_h, const int32_t* tex_w, const float* ctab, Scratch sc) {
  __shared__ __attribute__((aligned(16))) RowsShared S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)S.lut != 0u) __builtin_trap();   // (see RowsShared)
  S.lut[t & 255] = (double)(t & 255) / 255.0;
  RowsWave& W = S.w[wave];
  uint32_t three = 3;
  asm volatile("" : "+v"(three));                // (a vector register: the SDWA shift takes no literal)
  constexpr int DONE = 0x7fffffff;
  PH_DECL
  // The list is cut into n_shares shares (a few per workgroup) of equal estimated cost; a workgroup takes the next share off a
  // device-wide counter when its waves have run out of tiles (a static cut left the slowest workgroup 40 % behind the mean).
  if (t == 0) S.cur = -1;
  for (;;) {
    __syncthreads();                             // (later rounds: every wave is done with its share)
    if (t == 0) {
      const int share = (int)atomicAdd(sc.rows_next, 1);
      const bool any = share < n_shares;
      S.next = any ? sc.rows_bounds[share] : 0;
      S.end = any ? sc.rows_bounds[share + 1] : 0;
      S.first = any ? 1 : 0;
    }
    __syncthreads();
    if (!S.first) break;                         // no share left
    const int end = S.end;
    int pending = -1, ptex = -1, gi = 0;         // pending: index of the pulled tile in the list; DONE: the share is used up
    bool ctab_ok = false;
    for (;;) {
      if (pending < 0) {
        int i = 0;
        if (lane == 0) i = atomicAdd(&S.next, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i < end) {
          pending = i;
          gi = __builtin_amdgcn_readfirstlane(sc.rows_sorted[i]);
          ptex = __builtin_amdgcn_readfirstlane(as_constant(&sc.plan[gi])->tex);      // the tile's bucket: its texture, + n_tex for a Big tile
          if (__builtin_amdgcn_readfirstlane(as_constant(&sc.plan[gi])->kind) == KIND_BIG) ptex += sc.n_tex;
        } else {
          pending = DONE;
        }
      }
      if (pending != DONE && ptex == S.cur) {
        DropPlan p;
        {
          const const_ptr<uint32_t> src = as_constant(reinterpret_cast<const uint32_t*>(&sc.plan[gi]));
          uint32_t* dst = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
          for (int k = 0; k < (int)(sizeof(DropPlan) / 4); k++) dst[k] = src[k];
        }
        PH(0)                                     // pull + plan
        if (ptex < sc.n_tex) {
          rows_tile(p, tex_h[ptex], tex_w[ptex], S, W, sc.arena, three PH_PASS);
          ctab_ok = false;
        } else {
          float* s_ctab = reinterpret_cast<float*>(W.buf);      // the bicubic weights, wave-private (W is otherwise idle in a Big tile)
          if (!ctab_ok) {
            wave_lds_sync();
            s_ctab[lane] = ctab[lane];
            s_ctab[lane + 64] = ctab[lane + 64];
            wave_lds_sync();
            ctab_ok = true;
          }
          big_tile<RR_BIG_NP>(p, tex_h[ptex - sc.n_tex], tex_w[ptex - sc.n_tex], S.pair, s_ctab, sc.arena, three PH_PASS);
        }
        pending = -1;
        continue;
      }
      // another texture (or nothing left): meet the other waves
      if (lane == 0) { S.pend[wave] = pending; S.ptex[wave] = ptex; }
      __syncthreads();                            // nobody reads the resident texture any more
      int m = DONE, mtex = -1;
      for (int k = 0; k < RW_WAVES; k++)
        if (S.pend[k] < m) { m = S.pend[k]; mtex = S.ptex[k]; }
      if (m == DONE) break;                       // every wave is out of tiles
      if (mtex < sc.n_tex) {
        const int64_t nb = pair_bytes(tex_h[mtex], tex_w[mtex]);
        const uint4* g = reinterpret_cast<const uint4*>(sc.tex_pair + sc.tex_qoff[mtex]);
        uint4* d = reinterpret_cast<uint4*>(S.pair);
        for (int k = t; k < (int)(nb >> 4); k += 1024) d[k] = g[k];
      } else {                                    // a Big bucket: 16 zero bytes, then the texture with its zero border (+ 2 bytes: a window's
        const int bt = mtex - sc.n_tex;           //  last row may be read two bytes past it -- zeros of the padded copy's slack or of the next copy's border)
        const int n16 = ((tex_h[bt] + 4) * (tex_w[bt] + 4) + 2 + 15) >> 4;
        const uint4* g = reinterpret_cast<const uint4*>(sc.tex_pad + sc.tex_poff[bt]);
        uint4* d = reinterpret_cast<uint4*>(S.pair);
        if (t == 0) d[0] = uint4{0u, 0u, 0u, 0u};
        for (int k = t; k < n16; k += 1024) d[k + 1] = g[k];
      }
      if (t == 0) S.cur = mtex;
      __syncthreads();
      PH(5)                                       // texture switch (waiting for the other waves + staging)
    }
  }
  PH_FLUSH(5)
}

// ---------------------------------------------------------------------------
// work lists: which kernel takes which drop, blur work split into items of sub-tiles
// ---------------------------------------------------------------------------
constexpr int BLUR_ITEMS_PER_DROP = 8;

__device__ inline int blur_subtiles(const DropPlan& p, const BlurLayout& L) {
  return ((p.ew + L.wo - 1) / L.wo) * ((p.eh + L.ho - 1) / L.ho);
}

__device__ inline uint4 make_list_rec(const DropPlan& p, const int32_t* tex_h, const int32_t* tex_w, const Scratch& sc) {
  if (p.status != RR_DROP_OK) return make_uint4(0u, 0u, 0u, 0u);
  const int sh = tex_h[p.tex], sw = tex_w[p.tex];
  uint32_t cls, cost = 0u, bcls = LB_NONE, bwh = 0u, bns = 0u;
  if (p.kind == KIND_BIG) {
    if (sc.big_on && tile_is_big_lds(p, sh, sw)) { cls = LC_BIG_LDS; cost = (uint32_t)(RR_BIG_COST * ((p.tw * p.th + 63) / 64) + 4); }
    else { cls = LC_BIG; cost = (uint32_t)(p.tw * p.th); }
  } else if (sc.rows_on && p.kind != KIND_EXT && tile_is_rows(p, sh, sw)) { cls = LC_ROWS; cost = (uint32_t)rows_tile_cost(p, sh, sw); }
  else if (p.kind != KIND_EXT && tile_is_fast(p, sh, sw)) cls = p.rs_mode == RS_AREA_FAST ? LC_ROT_INT : LC_ROT;
  else cls = LC_GEN;
  if (p.r1 > 0) {
    if (blur_is_small(p)) bcls = LB_SMALL;
    else {
      const BlurLayout L = blur_layout(p, sc.blur_bx, sc.blur_by);
      if (L.fused) { bcls = LB_FUSED; bwh = (uint32_t)L.wo | ((uint32_t)L.ho << 16); bns = (uint32_t)blur_subtiles(p, L); }
      else bcls = LB_SLOW;
    }
  }
  return make_uint4(cls | (bcls << 4) | ((uint32_t)p.tex << 8), cost, bwh, bns);
}

__global__ __launch_bounds__(1024) void k_lists(const FrameDesc* frames, int max_drops, const int32_t* tex_h, const int32_t* tex_w,
                                                Scratch sc) {
  const int f = blockIdx.x, t = threadIdx.x;
  const int n = frames[f].n_drops;
  const int chunk = (n + 1023) / 1024;
  const int i0 = t * chunk, i1 = min(i0 + chunk, n);
  const int64_t base = (int64_t)f * max_drops;
  __shared__ int s_hist[RW_TEX_MAX], s_cost[RW_TEX_MAX];
  if (sc.rows_on) {
    for (int k = t; k < sc.n_buckets; k += 1024) s_hist[k] = s_cost[k] = 0;
    __syncthreads();
  }
  // #rot-fast (general), #generic, #fused blur items, #slow blur, #small blur, #rot-fast (integer ratio), #big, big pixels, #row-walk tiles
  constexpr int NC = 10;             // (+ Big tiles that ride in the row-walk list)
  int c[NC] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = i0; i < i1; i++) {
    const uint4 r = sc.lrec[base + i];
    const uint32_t cls = r.x & 15u, bcls = (r.x >> 4) & 7u;
    if (cls == LC_NONE || sc.sizes[base + i] == 0) continue;
    const int tex = (int)(r.x >> 8);
    if (sc.canon[base + i] == (int)(base + i)) {         // duplicates of another drop's raw tile render nothing
      if (cls == LC_BIG_LDS) {
        c[9]++;
        atomicAdd(&s_hist[sc.n_tex + tex], 1);
        atomicAdd(&s_cost[sc.n_tex + tex], (int)r.y);
      } else if (cls == LC_BIG) { c[6]++; c[7] += (int)r.y; }
      else if (cls == LC_ROWS) {
        c[8]++;
        atomicAdd(&s_hist[tex], 1);
        atomicAdd(&s_cost[tex], (int)r.y);
      }
      else if (cls == LC_ROT_INT) c[5]++;
      else if (cls == LC_ROT) c[0]++;
      else c[1]++;
    }
    if (bcls == LB_SMALL) c[4]++;
    else if (bcls == LB_FUSED) c[2] += imin((int)r.w, BLUR_ITEMS_PER_DROP);
    else if (bcls == LB_SLOW) c[3]++;
  }
  __shared__ int sh[1024][NC];
  for (int k = 0; k < NC; k++) sh[t][k] = c[k];
  __syncthreads();
  for (int ofs = 1; ofs < 1024; ofs <<= 1) {
    int v[NC] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (t >= ofs)
      for (int k = 0; k < NC; k++) v[k] = sh[t - ofs][k];
    __syncthreads();
    for (int k = 0; k < NC; k++) sh[t][k] += v[k];
    __syncthreads();
  }
  int o[NC];
  for (int k = 0; k < NC; k++) o[k] = (t == 0) ? 0 : sh[t - 1][k];
  const int n_int = sh[1023][5];     // integer-ratio drops go to the FRONT of the rot list (longest blocks first)
  const int n_rot_rows = sh[1023][8];
  o[0] += n_int;
  __shared__ int s_rot_base;
  if (t == 0) s_rot_base = atomicAdd(sc.rot_total, sh[1023][0] + sh[1023][5]);      // the frame's piece of the batch-wide list (any order of the frames)
  __syncthreads();
  int32_t* lrot = sc.list_rot + s_rot_base;
  int32_t* lgen = sc.list_gen + base;
  int32_t* lslow = sc.list_slow + base;
  int32_t* lsmall = sc.list_small + base;
  int32_t* lbig = sc.list_big + base;
  int32_t* lrows = sc.rows_list + base;
  int32_t* boff = sc.big_off + base + f;                 // n_big + 1 entries per frame
  int4* items = sc.blur_items + base * BLUR_ITEMS_PER_DROP;
  for (int i = i0; i < i1; i++) {
    const uint4 r = sc.lrec[base + i];
    const uint32_t cls = r.x & 15u, bcls = (r.x >> 4) & 7u;
    if (cls == LC_NONE || sc.sizes[base + i] == 0) continue;
    if (sc.canon[base + i] == (int)(base + i)) {
      if (cls == LC_BIG_LDS) lrows[n_rot_rows + o[9]++] = i;      // (behind the frame's rotate tiles)
      else if (cls == LC_BIG) {
        boff[o[6]] = o[7];
        lbig[o[6]++] = i;
        o[7] += (int)r.y;
      } else if (cls == LC_ROWS) lrows[o[8]++] = i;
      else if (cls == LC_ROT_INT) lrot[o[5]++] = (int32_t)(base + i);
      else if (cls == LC_ROT) lrot[o[0]++] = (int32_t)(base + i);
      else lgen[o[1]++] = i;
    }
    if (bcls == LB_SMALL) lsmall[o[4]++] = i;
    else if (bcls == LB_FUSED) {
      const int ns = (int)r.w;
      const int ni = imin(ns, BLUR_ITEMS_PER_DROP);
      const int per = (ns + ni - 1) / ni;
      for (int k = 0; k < ni; k++) {
        const int st0 = k * per;
        items[o[2]++] = make_int4(i, st0, imax(imin(per, ns - st0), 0), (int)r.z);   // layout travels with the item
      }
    } else if (bcls == LB_SLOW) lslow[o[3]++] = i;
  }
  if (t == 1023) {
    for (int k = 0; k < 5; k++) sc.counts[f * 8 + k] = sh[1023][k];
    sc.counts[f * 8 + 0] = sh[1023][0] + sh[1023][5];
    sc.counts[f * 8 + 5] = sh[1023][6];
    sc.counts[f * 8 + 6] = sh[1023][7];
    boff[sh[1023][6]] = sh[1023][7];
    sc.rows_n[2 * f] = sh[1023][8];
    sc.rows_n[2 * f + 1] = sh[1023][9];
  }
  if (sc.rows_on) {                  // the frame's place inside the batch-wide buckets (any order of the frames will do)
    for (int k = t; k < sc.n_buckets; k += 1024) {
      const int cnt = s_hist[k];
      sc.rows_fbase[(int64_t)f * RW_TEX_MAX + k] = cnt ? atomicAdd(&sc.rows_hist[k], cnt) : 0;
      if (cnt) atomicAdd(&sc.rows_cost[k], (unsigned long long)s_cost[k]);
    }
  }
}

// The frame's row-walk tiles into the batch-wide list: bucket start (prefix of the batch's histogram) + the frame's slot
// inside the bucket + a rank among the frame's tiles of that texture.  One workgroup per frame.
__global__ __launch_bounds__(1024) void k_rows_scatter(int max_drops, Scratch sc) {
  const int f = blockIdx.x, t = threadIdx.x;
  __shared__ int s_start[RW_TEX_MAX], s_rank[RW_TEX_MAX];
  const int nf = sc.rows_n[2 * f] + sc.rows_n[2 * f + 1];
  if (nf == 0) return;
  const int own = t < sc.n_buckets ? sc.rows_hist[t] : 0;
  s_start[t] = own;
  __syncthreads();
  for (int ofs = 1; ofs < 1024; ofs <<= 1) {
    const int v = t >= ofs ? s_start[t - ofs] : 0;
    __syncthreads();
    s_start[t] += v;
    __syncthreads();
  }
  const int excl = s_start[t] - own;
  __syncthreads();
  s_start[t] = excl;
  s_rank[t] = t < sc.n_buckets ? sc.rows_fbase[(int64_t)f * RW_TEX_MAX + t] : 0;
  __syncthreads();
  const int64_t base = (int64_t)f * max_drops;
  for (int j = t; j < nf; j += 1024) {
    const int i = sc.rows_list[base + j];
    const DropPlan& p = sc.plan[base + i];
    const int tex = p.tex + (p.kind == KIND_BIG ? sc.n_tex : 0);      // the bucket
    const int pos = s_start[tex] + atomicAdd(&s_rank[tex], 1);
    sc.rows_sorted[pos] = (int32_t)(base + i);
  }
}

// gaussian half-table: hw[k] = w[k], k = 0..r (centre at r), sequential normalisation
__device__ void gauss_half_table(double sigma, int r, double* hw /*LDS, r+1*/) {
  const int t = threadIdx.x;
  for (int k = t; k <= r; k += blockDim.x) hw[k] = gauss_phi(sigma, r - k);   // hw[k] = phi(|k - r|)
  __syncthreads();
  __shared__ double s_tot;
  if (t == 0) {
    double tot = 0.0;
    for (int x = -r; x <= r; x++) tot = tot + hw[r - (x < 0 ? -x : x)];
    s_tot = tot;
  }
  __syncthreads();
  const double tot = s_tot;
  for (int k = t; k <= r; k += blockDim.x) hw[k] = hw[k] / tot;
  __syncthreads();
}


// Normalised Gaussian half tables of every blurred drop: hw[k] = w(|k - r|), k = 0..r, with the oracle's left-to-right
// normalisation sum.  16 lanes per (drop, axis): the r + 1 exponentials (the expensive part: exp from + - * / only) are
// spread over the lanes, one lane adds them up in the oracle's order (x = -r..r), every lane divides its own entries.
// A table is 8*(r+1) bytes for the blur kernels to load.
constexpr int BW_LANES = 4;         // lanes per (drop, axis): most radii are below 8 -- with 16 lanes per table (r04) a third of them had an exponential to evaluate
__global__ __launch_bounds__(256) void k_blur_weights(const FrameDesc* frames, int max_drops, Scratch sc) {
  constexpr int NG = 256 / BW_LANES;
  __shared__ double s_hw[NG][BR_MAX + 2];
  const int f = blockIdx.y, grp = threadIdx.x / BW_LANES, l = threadIdx.x % BW_LANES;
  const int idx = blockIdx.x * NG + grp;
  const int i = idx >> 1, axis = idx & 1;
  int r = 0;
  double sigma = 0.0;
  const int64_t gi = (int64_t)f * max_drops + (i < frames[f].n_drops ? i : 0);
  if (i < frames[f].n_drops) {
    const DropPlan& p = sc.plan[gi];
    if (p.status == RR_DROP_OK && sc.sizes[gi] != 0 && p.r1 > 0 && p.r1 <= BR_MAX) {
      r = axis ? p.r2 : p.r1;
      sigma = axis ? p.sig2 : p.sig1;
    }
  }
  double* tab = s_hw[grp];
  for (int k = l; k <= r && r > 0; k += BW_LANES) tab[k] = gauss_phi(sigma, r - k);       // hw[k] = phi(|k - r|)
  wave_lds_sync();                                       // (a group never spans two waves)
  if (l == 0 && r > 0) {
    double tot = 0.0;
    for (int x = -r; x <= r; x++) tot = tot + tab[r - (x < 0 ? -x : x)];
    tab[BR_MAX + 1] = tot;
  }
  wave_lds_sync();
  if (r > 0) {
    const double tot = tab[BR_MAX + 1];
    double* hw = sc.wtab + (gi * 2 + axis) * (BR_MAX + 1);
    for (int k = l; k <= r; k += BW_LANES) hw[k] = tab[k] / tot;
  }
}

// Four consecutive outputs (stride `st` doubles apart) of the symmetric correlate1d of radius r:
//   acc_k = c[k]*w[r];  for ii = -r..-1:  acc_k = acc_k + (c[k+ii] + c[k-ii]) * w[ii+r]
// As the tap distance shrinks the upper/lower operand windows of the four outputs slide by one
// element, so each tap needs two new LDS values instead of eight; the loop is unrolled by four so
// that the sliding is pure register renaming (no moves).  c0 = centre of output 0.
// hw(k) = w(|k - r|), k = 0..r: an LDS table (fused kernel) or the lanes of a register (wave-per-drop kernel).
template <class W>
__device__ inline void blur4(const double* c0, int st, W hw, int r, double& acc0, double& acc1, double& acc2, double& acc3) {
  // element k * st of the window: a 24-bit multiply (|k| <= 2 r + 4, st < 4096).  A full 32-bit one compiles to v_mad_u64_u32 with a
  // 64-bit addend, for whose unused upper half the register allocator may pick a register that a prefetch load is still writing:
  // the wait-count pass then guards the multiplication with vmcnt(0) -- k_blur_small waited for its NEXT drop's loads in the
  // middle of the current drop's row pass (r05).
  auto at = [&](int k) { return c0[__mul24(k, st)]; };
  const double wc = hw(r);
  acc0 = c0[0] * wc; acc1 = at(1) * wc; acc2 = at(2) * wc; acc3 = at(3) * wc;
  double a0 = at(-r), a1 = at(1 - r), a2 = at(2 - r), a3 = at(3 - r);
  double b0 = at(r), b1 = at(1 + r), b2 = at(2 + r), b3 = at(3 + r);
  int ii = -r;
  for (; ii + 3 < 0; ii += 4) {
    const double w0 = hw(ii + r), w1 = hw(ii + 1 + r), w2 = hw(ii + 2 + r), w3 = hw(ii + 3 + r);
    const double na0 = at(4 + ii), na1 = at(5 + ii), na2 = at(6 + ii), na3 = at(7 + ii);
    const double nb0 = at(-ii - 1), nb1 = at(-ii - 2), nb2 = at(-ii - 3), nb3 = at(-ii - 4);
    acc0 = acc0 + (a0 + b0) * w0; acc1 = acc1 + (a1 + b1) * w0; acc2 = acc2 + (a2 + b2) * w0; acc3 = acc3 + (a3 + b3) * w0;
    acc0 = acc0 + (a1 + nb0) * w1; acc1 = acc1 + (a2 + b0) * w1; acc2 = acc2 + (a3 + b1) * w1; acc3 = acc3 + (na0 + b2) * w1;
    acc0 = acc0 + (a2 + nb1) * w2; acc1 = acc1 + (a3 + nb0) * w2; acc2 = acc2 + (na0 + b0) * w2; acc3 = acc3 + (na1 + b1) * w2;
    acc0 = acc0 + (a3 + nb2) * w3; acc1 = acc1 + (na0 + nb1) * w3; acc2 = acc2 + (na1 + nb0) * w3; acc3 = acc3 + (na2 + b0) * w3;
    a0 = na0; a1 = na1; a2 = na2; a3 = na3;
    b3 = nb0; b2 = nb1; b1 = nb2; b0 = nb3;
  }
  for (; ii < 0; ii++) {
    const double w = hw(ii + r);
    const double na = at(4 + ii);                               // next upper element of output 3
    const double nb = at(-ii - 1);                              // next lower element of output 0
    acc0 = acc0 + (a0 + b0) * w;
    acc1 = acc1 + (a1 + b1) * w;
    acc2 = acc2 + (a2 + b2) * w;
    acc3 = acc3 + (a3 + b3) * w;
    a0 = a1; a1 = a2; a2 = a3; a3 = na;
    b3 = b2; b2 = b1; b1 = b0; b0 = nb;
  }
}

// The same outputs one or two per lane, each with its own chain (no shared window): for the tiles of k_blur_small whose
// four-output blocks leave most of the wave idle (a 4 x 14 tile of radius 2 is 16 blocks of the row pass: a quarter of the
// lanes).  The arithmetic of an output is blur4's, operand for operand.
template <int NO, class W>
__device__ inline void blur_n(const double* c0, int st, W hw, int r, double (&acc)[NO]) {
  const double wc = hw(r);
#pragma unroll
  for (int j = 0; j < NO; j++) acc[j] = c0[__mul24(j, st)] * wc;
  for (int ii = -r; ii < 0; ii++) {
    const double w = hw(ii + r);
#pragma unroll
    for (int j = 0; j < NO; j++) acc[j] = acc[j] + (c0[__mul24(j + ii, st)] + c0[__mul24(j - ii, st)]) * w;
  }
}

#ifdef RR_EXPERIMENTS                // (r04's register-staged form: the LDS-DMA kernel below replaced it in r05; kept for A/B builds, RR_OPT_BLUR_DMA 0)
// ---------------------------------------------------------------------------
// fused defocus blur: both axes of the separable filter through LDS
// ---------------------------------------------------------------------------
// WPE = waves per SIMD the register allocation is held to (= workgroups per CU; the LDS capacities in sc.blur_bx / blur_by
// are chosen to match, see above).  LDS (dynamic): hw1 | hw2 | X[blur_bx] | Y[blur_by].
template <int WPE>
__global__ __launch_bounds__(256, WPE) void k_blur_fused(const FrameDesc* frames, int max_drops, Scratch sc) {
  const int f = blockIdx.y, t = threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* hw1 = s_dyn;
  double* hw2 = s_dyn + (BR_MAX + 1);
  double* X = s_dyn + 2 * (BR_MAX + 1);                                 // 98 doubles in front: X and Y stay 16-byte aligned
  double* Y = X + sc.blur_bx;
  const int n_items = sc.counts[f * 8 + 2];
  if ((int)blockIdx.x >= n_items) return;
  // r04: the item record and the plan of the NEXT item travel ahead of the current one's arithmetic, by vector loads (lane l
  // reads dword l of the record; the fields come out by v_readlane): the load phase of an item was a chain of three
  // dependent round trips (item -> plan -> raw tile), 48 % of the kernel's wave time, and only the last one is left.
  // (Scalar loads cannot do this: they share the LDS operations' counter and return out of order.)
  const int lane = t & 63, G = gridDim.x;
  const global_ptr<const uint32_t> items_w = as_global(reinterpret_cast<const uint32_t*>(sc.blur_items + (int64_t)f * max_drops * BLUR_ITEMS_PER_DROP));
  const DropPlan* plans = sc.plan + (int64_t)f * max_drops;
  auto load_item = [&](int item) { return items_w[(int64_t)item * 4 + (lane & 3)]; };
  auto load_plan = [&](int i) { return as_global(reinterpret_cast<const uint32_t*>(plans + i))[lane]; };
  struct PlanView {                   // the plan fields this kernel uses, wave-uniform
    int r1, r2, ew, eh, tw, th, epitch, epad;
    long long a0_off, a1_off;
  };
  auto unpack = [&](uint32_t pv) {
    auto F = [&](size_t byte_off) { return (int)__builtin_amdgcn_readlane((int)pv, (int)(byte_off / 4)); };
    PlanView o;
    o.r1 = F(offsetof(DropPlan, r1)); o.r2 = F(offsetof(DropPlan, r2)); o.ew = F(offsetof(DropPlan, ew)); o.eh = F(offsetof(DropPlan, eh));
    o.tw = F(offsetof(DropPlan, tw)); o.th = F(offsetof(DropPlan, th)); o.epitch = F(offsetof(DropPlan, epitch)); o.epad = F(offsetof(DropPlan, epad));
    o.a0_off = (long long)(((unsigned long long)(unsigned)F(offsetof(DropPlan, a0_off) + 4) << 32) | (unsigned)F(offsetof(DropPlan, a0_off)));
    o.a1_off = (long long)(((unsigned long long)(unsigned)F(offsetof(DropPlan, a1_off) + 4) << 32) | (unsigned)F(offsetof(DropPlan, a1_off)));
    return o;
  };
  uint32_t iv_cur = load_item(blockIdx.x);
  uint32_t pv_cur = load_plan((int)__builtin_amdgcn_readlane((int)iv_cur, 0));
  uint32_t iv_nxt = ((int)blockIdx.x + G < n_items) ? load_item(blockIdx.x + G) : 0u;
  int cur = -1;
  PH_DECL
  for (int it = blockIdx.x; it < n_items; it += G) {                    // grid-stride over (drop, sub-tile range) items
  const int4 item = make_int4((int)__builtin_amdgcn_readlane((int)iv_cur, 0), (int)__builtin_amdgcn_readlane((int)iv_cur, 1),
                              (int)__builtin_amdgcn_readlane((int)iv_cur, 2), (int)__builtin_amdgcn_readlane((int)iv_cur, 3));
  const int64_t gi = (int64_t)f * max_drops + item.x;
  const PlanView p = unpack(pv_cur);
  {                                   // the next item's plan (its record arrived an item ago) and the record after that
    const uint32_t pv_n = (it + G < n_items) ? load_plan((int)__builtin_amdgcn_readlane((int)iv_nxt, 0)) : 0u;
    const uint32_t iv_n = (it + 2 * G < n_items) ? load_item(it + 2 * G) : 0u;
    iv_cur = iv_nxt;
    pv_cur = pv_n;
    iv_nxt = iv_n;
  }
  BlurLayout L{1, item.w & 0xffff, item.w >> 16, 0};               // computed once, by k_lists
  const int r1 = p.r1, r2 = p.r2, pw = p.ew, ph = p.eh;            // the tile being produced is the EFFECTIVE tile
  // the weight tables depend on the drop only; they are fetched by waves 0 and 1 while the first
  // batch of tile loads is in flight (every item ends with a barrier, so the old tables are free)
  bool need_tables = cur != item.x;
  const double* src = sc.arena + p.a0_off;          // raw tile (tw x th); the pad is implicit zeros
  double* dst = sc.arena + p.a1_off;                // finished effective tile (ew x eh); raw sits at (r2, r1) inside it
  const int tw = p.tw, th = p.th;
  const int ntx = (pw + L.wo - 1) / L.wo;
  for (int st = item.y; st < item.y + item.z; st++) {
    const int sty = st / ntx, stx = st - sty * ntx;
    const int y0 = sty * L.ho, x0 = stx * L.wo;
    {
      const int ho = imin(L.ho, ph - y0);
      const int wo = imin(L.wo, pw - x0);
      const int hop = (ho + 3) & ~3;                      // rows/columns padded to the 4-output blocks
      const int wi = wo + 2 * r2, hi = hop + 2 * r1;      // LDS input tile with zero halos (+ slack rows)
      const int yp = blur_y_pitch(wo, r2);                // odd pitch of the row-pass result
      // Only the columns under the raw tile carry data through the row pass (it filters along y: a column
      // without raw pixels stays exactly zero).  X holds those wd columns, the row pass runs over them; Y is
      // cleared as a whole first (wide stores, no index arithmetic) and the row pass writes its columns over it.
      const int xa = imax(0, 2 * r2 - x0), xb = imin(wi, tw + 2 * r2 - x0);   // data columns of the haloed sub-tile
      const int wd = imax(xb - xa, 0);                    // 0: the sub-tile lies beside the raw tile (all zeros)
      const int wdd = imax(wd, 1);
      const float inv_wd = 1.0f / (float)wdd;
      // data columns of the haloed tile -> LDS; eight independent global loads in flight per thread.  By the
      // choice of xa / xb every column is inside the raw tile; only the row needs a test.  (row, column) of the
      // thread's first element by one float division, the following ones (256 apart) incrementally.
      const int nx = wd * hi;
      const int dq = (int)((256.0f + 0.5f) * inv_wd), dr = 256 - dq * wdd;
      const int xs = x0 - 2 * r2 + xa, ys = y0 - 2 * r1;
      int yy = (int)(((float)t + 0.5f) * inv_wd), xc = t - yy * wdd;
      for (int base = t; base < nx || need_tables; base += 2048) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int y = ys + yy;
          v[k] = (base + 256 * k < nx && (unsigned)y < (unsigned)th) ? src[y * tw + (xs + xc)] : 0.0;
          xc += dr;
          yy += dq;
          if (xc >= wdd) { xc -= wdd; yy += 1; }
        }
        if (need_tables) {
          const double* wt = sc.wtab + gi * 2 * (BR_MAX + 1);                // built by k_blur_weights
          if (t <= r1) hw1[t] = wt[t];
          if (t >= 64 && t - 64 <= r2 && r2 > 0) hw2[t - 64] = wt[(BR_MAX + 1) + t - 64];
          need_tables = false;           // (the barrier after the tile load publishes the tables)
          cur = item.x;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (base + 256 * k < nx) X[base + 256 * k] = v[k];
      }
      {
        const int nz2 = (yp * hop + 1) >> 1;              // Y := 0, two doubles per store (capacity is even: no overrun)
        double2* Y2 = reinterpret_cast<double2*>(Y);
        for (int i = t; i < nz2; i += 256) Y2[i] = make_double2(0.0, 0.0);
      }
      PH(0)                                         // plan, tables, raw sub-tile -> LDS (issue)
      __syncthreads();
      PH(1)                                         // barrier: the loads land
      // axis 0 (rows, sigma = c): symmetric correlate1d.  A thread owns data column xc and FOUR consecutive
      // rows; as the tap distance shrinks the upper/lower operand windows slide by one row, so each
      // step needs two new LDS values instead of eight (register rotation).  Lanes run along x.
      {
        const int nrb = hop >> 2;
        const int nv = nrb * wd;
        for (int idx = t; idx < nv; idx += 256) {
          const int rb = (int)(((float)idx + 0.5f) * inv_wd), xq = idx - rb * wd;
          const double* c0 = X + (4 * rb + r1) * wd + xq;                     // centre of the first of the four rows
          double acc0, acc1, acc2, acc3;
          blur4(c0, wd, [&](int k) { return hw1[k]; }, r1, acc0, acc1, acc2, acc3);
          double* o = Y + 4 * rb * yp + xa + xq;                              // Y has hop rows: the slack rows are never read
          o[0] = acc0;
          o[yp] = acc1;
          o[2 * yp] = acc2;
          o[3 * yp] = acc3;
        }
      }
      PH(2)                                         // row pass
      __syncthreads();
      PH(3)
      // axis 1 (columns, sigma = c/2): a thread owns row y and four consecutive columns; lanes run
      // down the rows (odd pitch -> distinct banks).
      {
        const int ncb = (wo + 3) >> 2;
        const int nh = ncb * ho;
        const float inv_ho = 1.0f / (float)ho;
        for (int idx = t; idx < nh; idx += 256) {
          const int cb = (int)(((float)idx + 0.5f) * inv_ho), yq = idx - cb * ho;
          const double* c0 = Y + yq * yp + 4 * cb + r2;                       // centre of the first of the four columns
          double acc0, acc1, acc2, acc3;
          if (r2 > 0) {
            blur4(c0, 1, [&](int k) { return hw2[k]; }, r2, acc0, acc1, acc2, acc3);
          } else {
            acc0 = c0[0]; acc1 = c0[1]; acc2 = c0[2]; acc3 = c0[3];
          }
          const int xo = 4 * cb;
          double* o = dst + (int64_t)(y0 + yq) * p.epitch + p.epad + (x0 + xo);
          if (xo + 3 < wo) {
            o[0] = acc0; o[1] = acc1; o[2] = acc2; o[3] = acc3;
          } else {
            o[0] = acc0;
            if (xo + 1 < wo) o[1] = acc1;
            if (xo + 2 < wo) o[2] = acc2;
          }
        }
      }
      PH(4)                                         // column pass + store
      __syncthreads();
      PH(5)
    }
  }
  }
  PH_FLUSH(3)
}

#endif

// ---------------------------------------------------------------------------
// fused defocus blur, staged by LDS-DMA a sub-tile ahead (r05; the default)
// ---------------------------------------------------------------------------
// k_blur_fused spends 47 % of its wave time in the load phase of a sub-tile (a round trip to memory through registers,
// nothing else to do meanwhile) and its software-pipelined form died of register pressure (r04: 168 VGPRs).  gfx950's
// LDS-DMA loads (global_load_lds_dword: every lane names a global address, the 64 dwords land side by side in LDS, no
// register in between) make the pipeline free:
//   barrier A | row pass (X, hw1 -> Y) | barrier B | ISSUE the next sub-tile's loads into X | column pass (Y, hw2 -> HBM)
//   | barrier C | clear Y | wait for the loads, zero the halo rows | barrier A ...
// X is dead after the row pass, so the next sub-tile's raw data (same item or the next item of the workgroup) travels
// while the column pass, its stores and the clearing of Y run.  The weight tables of a new drop travel the same way into
// the other of two table slots (the column pass still reads the current drop's hw2).
//   * an element of X is two dwords: lanes 2j, 2j + 1 of a wave carry element j of its 32-element piece; a round of the four
//     waves moves 128 consecutive elements of X; (row, column) of a lane's element by the incremental division of the
//     r04 loader.  Rows of the haloed tile outside the raw tile are not loaded (lanes masked) and zeroed after the wait;
//   * the loads are issued by inline assembly (M0 = LDS address of the wave's piece): the compiler neither counts them nor
//     orders LDS reads behind them -- a counted wait of its own can only wait longer; the one wait that matters is the
//     explicit vmcnt(0) in front of barrier A;
//   * item records and plans travel two items ahead (vector loads, fields by v_readlane), as the plan of the next item is
//     needed a sub-tile early.
// Arithmetic, fold order and the LDS images of X, Y and the tables are k_blur_fused's: the tiles are the same bits
// (tests/test_gpu_properties.py, RR_OPT_BLUR_DMA 0 / 1).
__device__ inline void glds_dword(const void* src, uint32_t lds_byte) {     // LDS[lds_byte + 4 * lane] <- *(const uint32_t*)src, asynchronously
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)lds_byte))      // (wave-uniform by construction; the compiler wants the proof)
               : "memory");
}
__device__ inline void glds_dwordx4(const void* src, uint32_t lds_byte) {   // LDS[lds_byte + 16 * lane .. + 16) <- 16 bytes at src (both 16-byte aligned)
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)lds_byte))
               : "memory");
}
template <int WPE>
__global__ __launch_bounds__(256, WPE) void k_blur_fused_dma(const FrameDesc* frames, int max_drops, Scratch sc) {
  const int f = blockIdx.y, t = threadIdx.x, lane = t & 63, G = gridDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  constexpr int TAB = 2 * (BR_MAX + 1);                                 // hw1 | hw2 of one drop
  double* tabs = s_dyn;                                                 // two generations
  double* X = s_dyn + 2 * TAB;                                          // 196 doubles in front: X and Y stay 16-byte aligned
  double* Y = X + sc.blur_bx;
  const uint32_t lds_tabs = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)tabs;
  const uint32_t lds_x = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)X;
  const int n_items = sc.counts[f * 8 + 2];
  if ((int)blockIdx.x >= n_items) return;
  const global_ptr<const uint32_t> items_w = as_global(reinterpret_cast<const uint32_t*>(sc.blur_items + (int64_t)f * max_drops * BLUR_ITEMS_PER_DROP));
  const DropPlan* plans = sc.plan + (int64_t)f * max_drops;
  auto load_item = [&](int item) { return items_w[(int64_t)item * 4 + (lane & 3)]; };
  auto load_plan = [&](int i) { return as_global(reinterpret_cast<const uint32_t*>(plans + i))[lane]; };
  struct PlanView {                   // the plan fields this kernel uses, wave-uniform
    int r1, r2, ew, eh, tw, th, epitch, epad;
    long long a0_off, a1_off;
  };
  auto unpack = [&](uint32_t pv) {
    auto F = [&](size_t byte_off) { return (int)__builtin_amdgcn_readlane((int)pv, (int)(byte_off / 4)); };
    PlanView o;
    o.r1 = F(offsetof(DropPlan, r1)); o.r2 = F(offsetof(DropPlan, r2)); o.ew = F(offsetof(DropPlan, ew)); o.eh = F(offsetof(DropPlan, eh));
    o.tw = F(offsetof(DropPlan, tw)); o.th = F(offsetof(DropPlan, th)); o.epitch = F(offsetof(DropPlan, epitch)); o.epad = F(offsetof(DropPlan, epad));
    o.a0_off = (long long)(((unsigned long long)(unsigned)F(offsetof(DropPlan, a0_off) + 4) << 32) | (unsigned)F(offsetof(DropPlan, a0_off)));
    o.a1_off = (long long)(((unsigned long long)(unsigned)F(offsetof(DropPlan, a1_off) + 4) << 32) | (unsigned)F(offsetof(DropPlan, a1_off)));
    return o;
  };
  auto item_of = [&](uint32_t iv) {
    return make_int4((int)__builtin_amdgcn_readlane((int)iv, 0), (int)__builtin_amdgcn_readlane((int)iv, 1),
                     (int)__builtin_amdgcn_readlane((int)iv, 2), (int)__builtin_amdgcn_readlane((int)iv, 3));
  };
  struct Geo {                        // one sub-tile (wave-uniform), k_blur_fused's quantities
    int y0, x0, ho, wo, hop, hi, yp, xa, wd, nx, xs, ys, e_lo, e_hi;
  };
  auto geo = [&](const PlanView& p, int Lwo, int Lho, int st) {
    Geo g;
    const int ntx = (p.ew + Lwo - 1) / Lwo;
    const int sty = st / ntx, stx = st - sty * ntx;
    g.y0 = sty * Lho; g.x0 = stx * Lwo;
    g.ho = imin(Lho, p.eh - g.y0);
    g.wo = imin(Lwo, p.ew - g.x0);
    g.hop = (g.ho + 3) & ~3;
    const int wi = g.wo + 2 * p.r2;
    g.hi = g.hop + 2 * p.r1;
    g.yp = blur_y_pitch(g.wo, p.r2);
    g.xa = imax(0, 2 * p.r2 - g.x0);
    const int xb = imin(wi, p.tw + 2 * p.r2 - g.x0);
    g.wd = imax(xb - g.xa, 0);
    g.nx = g.wd * g.hi;
    g.xs = g.x0 - 2 * p.r2 + g.xa; g.ys = g.y0 - 2 * p.r1;
    // rows [ra, rb) of the haloed tile lie inside the raw tile: elements [e_lo, e_hi) of X are loaded, the rest is zero
    const int ra = imin(imax(-g.ys, 0), g.hi), rb = imax(imin(p.th - g.ys, g.hi), ra);
    g.e_lo = ra * g.wd; g.e_hi = rb * g.wd;
    return g;
  };
  // the loads of one sub-tile: raw data rows into X, and (new drop) its two weight tables into table slot tb
  auto stage = [&](const PlanView& p, const Geo& g, int drop, bool tables, int tb) {
    if (g.e_hi > g.e_lo && g.wd == p.tw) {
      // A sub-tile as wide as the raw tile (nearly all of them: blur_layout prefers full-width bands) is ONE contiguous run
      // of the raw tile: element e of X is element ys * tw + e of the tile.  16 bytes per lane -- a quarter of the load
      // instructions: the dword form spent more wave time ISSUING its loads (64 addresses per 256 bytes) than the column
      // pass takes (profiles/r05_phase_clocks_blur_dma_dword.txt).  The LDS side of a 16-byte piece is aligned (pairs start
      // on even elements), the global side need not be (scripts/probes/glds_probe.hip: any 4-byte alignment loads
      // correctly, lanes switched off in EXEC write nothing).  An odd first / last loaded element travels alone (two
      // dwords): a pair must not reach into the halo, whose zeros are STORED after the wait -- nothing orders a plain LDS
      // store behind a DMA write to the same address (a first version let the pair bring the neighbour's element along and
      // zeroed it afterwards: wrong tiles now and then).
      const char* src = reinterpret_cast<const char*>(sc.arena + p.a0_off + (int64_t)g.ys * p.tw);
      const int a = (g.e_lo + 1) & ~1, b = g.e_hi & ~1;
      for (int e0 = (a & ~511); e0 < b; e0 += 512) {
        const int e = e0 + wave * 128 + 2 * lane;
        if (e >= a && e < b) glds_dwordx4(src + (int64_t)e * 8, lds_x + (uint32_t)(e0 + wave * 128) * 8u);
      }
      if (wave == 0 && (g.e_lo & 1) && lane < 2) glds_dword(src + (int64_t)g.e_lo * 8 + lane * 4, lds_x + (uint32_t)g.e_lo * 8u);
      if (wave == 1 && (g.e_hi & 1) && g.e_hi - 1 >= g.e_lo && lane < 2) glds_dword(src + (int64_t)(g.e_hi - 1) * 8 + lane * 4, lds_x + (uint32_t)(g.e_hi - 1) * 8u);
    } else if (g.e_hi > g.e_lo) {
      const int wdd = imax(g.wd, 1);
      const float inv_wd = 1.0f / (float)wdd;
      const int dq = (int)((128.0f + 0.5f) * inv_wd), dr = 128 - dq * wdd;
      const char* src = reinterpret_cast<const char*>(sc.arena + p.a0_off) + (lane & 1) * 4;
      const int c0 = g.e_lo >> 7, c1 = (g.e_hi + 127) >> 7;               // pieces of 128 elements that hold loaded rows
      int e = c0 * 128 + wave * 32 + (lane >> 1);
      int yy = (int)(((float)e + 0.5f) * inv_wd), xc = e - yy * wdd;
      for (int c = c0; c < c1; c++) {
        if (e >= g.e_lo && e < g.e_hi)
          glds_dword(src + (int64_t)((g.ys + yy) * p.tw + (g.xs + xc)) * 8, lds_x + (uint32_t)(c * 128 + wave * 32) * 8u);
        e += 128;
        xc += dr;
        yy += dq;
        if (xc >= wdd) { xc -= wdd; yy += 1; }
      }
    }
    if (tables) {                     // 2 * 49 doubles = 196 dwords, as they lie in wtab (k_blur_weights): hw1 | hw2
      const char* wt = reinterpret_cast<const char*>(sc.wtab + ((int64_t)f * max_drops + drop) * TAB);     // (784 bytes per drop: 16-byte aligned)
      if (wave == 3 && lane < TAB / 2) glds_dwordx4(wt + lane * 16, lds_tabs + (uint32_t)(tb * TAB) * 8u);
    }
  };
  // item records and plans, two items ahead
  uint32_t iv0 = load_item(blockIdx.x);
  uint32_t pv0 = load_plan((int)__builtin_amdgcn_readlane((int)iv0, 0));
  uint32_t iv1 = load_item(imin((int)blockIdx.x + G, n_items - 1));
  uint32_t pv1 = load_plan((int)__builtin_amdgcn_readlane((int)iv1, 0));
  uint32_t iv2 = load_item(imin((int)blockIdx.x + 2 * G, n_items - 1));
  int it = blockIdx.x;
  int4 item = item_of(iv0);
  PlanView p = unpack(pv0);
  int st = item.y, tb = 0;
  Geo g = geo(p, item.w & 0xffff, item.w >> 16, st);
  stage(p, g, item.x, true, tb);
  PH_DECL
  for (;;) {
    const int r1 = p.r1, r2 = p.r2;
    const double* hw1 = tabs + tb * TAB;
    const double* hw2 = hw1 + (BR_MAX + 1);
    {
      const int nz2 = (g.yp * g.hop + 1) >> 1;            // Y := 0, two doubles per store (capacity is even: no overrun)
      double2* Y2 = reinterpret_cast<double2*>(Y);
      for (int i = t; i < nz2; i += 256) Y2[i] = make_double2(0.0, 0.0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this sub-tile's loads (issued a sub-tile ago) have landed
    // (here, where everything in flight has just been waited for: the compiler guards the reads of the prefetched record /
    //  plan with vmcnt(0), which after barrier B would wait for the column pass' stores -- 9 % of the wave time)
    // ---- the next sub-tile: of this item, or the first one of the workgroup's next item (its loads go out after barrier B) ----
    bool has_next = true, new_item = false;
    int4 nitem = item;
    PlanView np = p;
    int nst = st + 1, ntb = tb;
    if (nst >= item.y + item.z) {
      if (it + G < n_items) {
        new_item = true;
        nitem = item_of(iv1);
        np = unpack(pv1);
        nst = nitem.y;
        if (nitem.x != item.x) ntb = tb ^ 1;
      } else {
        has_next = false;
      }
    }
    {
      // keep the records / plans two items ahead.  Unconditional loads (the same record again while the item lasts): a
      // load whose result had to be merged with the old value at the end of a branch was waited for on the spot.  BEFORE the
      // sub-tile's loads go out: the compiler waits for the previous round of these two with vmcnt(0), which would drain them.
      iv1 = new_item ? iv2 : iv1;
      pv1 = load_plan((int)__builtin_amdgcn_readlane((int)iv1, 0));
      iv2 = load_item(imin(new_item ? it + 3 * G : it + 2 * G, n_items - 1));              // (past the list: the last record again, never used)
    }
    for (int i = t; i < g.e_lo; i += 256) X[i] = 0.0;     // halo rows above / below the raw tile
    for (int i = g.e_hi + t; i < g.nx; i += 256) X[i] = 0.0;
    PH(0)
    __syncthreads();                                      // A
    PH(1)
    {                                                     // axis 0 (rows): as k_blur_fused
      const int wd = g.wd, nrb = g.hop >> 2, nv = nrb * wd;
      const float inv_wd = 1.0f / (float)imax(wd, 1);
      for (int idx = t; idx < nv; idx += 256) {
        const int rb = (int)(((float)idx + 0.5f) * inv_wd), xq = idx - rb * wd;
        const double* c0 = X + (4 * rb + r1) * wd + xq;
        double acc0, acc1, acc2, acc3;
        blur4(c0, wd, [&](int k) { return hw1[k]; }, r1, acc0, acc1, acc2, acc3);
        double* o = Y + 4 * rb * g.yp + g.xa + xq;
        o[0] = acc0;
        o[g.yp] = acc1;
        o[2 * g.yp] = acc2;
        o[3 * g.yp] = acc3;
      }
    }
    PH(2)
    __syncthreads();                                      // B: X is free
    PH(3)
    Geo ng = g;
    if (has_next) {
      ng = geo(np, nitem.w & 0xffff, nitem.w >> 16, nst);
      stage(np, ng, nitem.x, ntb != tb, ntb);
    }
    PH(6)                                                 // the next sub-tile's loads issued
    {                                                     // axis 1 (columns) -> global: as k_blur_fused
      const int ncb = (g.wo + 3) >> 2, nh = ncb * g.ho;
      const float inv_ho = 1.0f / (float)g.ho;
      double* dst = sc.arena + p.a1_off;
      for (int idx = t; idx < nh; idx += 256) {
        const int cb = (int)(((float)idx + 0.5f) * inv_ho), yq = idx - cb * g.ho;
        const double* c0 = Y + yq * g.yp + 4 * cb + r2;
        double acc0, acc1, acc2, acc3;
        if (r2 > 0) {
          blur4(c0, 1, [&](int k) { return hw2[k]; }, r2, acc0, acc1, acc2, acc3);
        } else {
          acc0 = c0[0]; acc1 = c0[1]; acc2 = c0[2]; acc3 = c0[3];
        }
        const int xo = 4 * cb;
        double* o = dst + (int64_t)(g.y0 + yq) * p.epitch + p.epad + (g.x0 + xo);
        if (xo + 3 < g.wo) {
          o[0] = acc0; o[1] = acc1; o[2] = acc2; o[3] = acc3;
        } else {
          o[0] = acc0;
          if (xo + 1 < g.wo) o[1] = acc1;
          if (xo + 2 < g.wo) o[2] = acc2;
        }
      }
    }
    PH(4)
    if (!has_next) break;
    __syncthreads();                                      // C: Y and the old table slot are free
    PH(5)
    if (new_item) it += G;
    item = nitem; p = np; st = nst; tb = ntb; g = ng;
  }
  PH_FLUSH(3)
}

// Small blurred tiles (most of them): one WAVE per drop, wave-private LDS, no block barrier.  Same scheme as the fused
// kernel -- data columns of the haloed tile in X, row pass into Y (odd pitch), column pass to global memory, four
// outputs per lane with the rotating register window of blur4 -- with the filter weights in registers: lane l holds
// w(distance l) of each axis and the fold loop takes them with v_readlane.
// r04: the phase clocks showed 63 % of this kernel's wave time in the load phase (a chain of dependent global loads per
// drop: list entry -> plan -> raw tile / weights, ~2 us each time).  It is now a software pipeline over the wave's drops:
//   * a drop's plan fields travel as wave-uniform values, fetched TWO drops ahead -- by VECTOR loads (lane l reads dword l
//     of the record, v_readlane hands out the fields): scalar loads share the LDS operations' counter and return out of
//     order, so a scalar prefetch in flight turned every LDS wait of the row pass into a wait for memory (measured: the
//     row pass went from 11 % to 54 % of the wave time);
//   * its raw tile and weights are loaded ONE drop ahead, into registers, while the current drop is being filtered.  X
//     holds the tw columns under the raw tile with 2*r1 zero rows above: its data rows ARE the raw tile's dense layout, so
//     the load is a linear copy -- four 16-byte loads per lane cover the 512 doubles X can hold -- written to LDS when
//     the previous drop's column pass has finished; only the halo rows are cleared.
typedef double double2_v __attribute__((ext_vector_type(2)));     // (a native vector: loads through address-space pointers)
struct SmallItem {                  // what k_blur_small needs of a drop, all wave-uniform
  int li, r1, r2, tw, th, pw, ph, epitch, epad;
  long long a0, a1;
};
__global__ __launch_bounds__(256, 4) void k_blur_small(const FrameDesc* frames, int max_drops, Scratch sc) {
  const int f = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) double Xs[4][BS_X], Ys[4][BS_Y];
  double* X = Xs[wave];
  double* Y = Ys[wave];
  const int n_items = sc.counts[f * 8 + 4];
  const global_ptr<const int32_t> list = as_global(sc.list_small + (int64_t)f * max_drops);
  const DropPlan* plans = sc.plan + (int64_t)f * max_drops;
  const int stride = gridDim.x * 4;
  int it = blockIdx.x * 4 + wave;
  if (it >= n_items) return;
  // dword `lane` of a plan (its first 256 bytes hold every field used here); unpack: the fields as wave-uniform values
  auto load_plan = [&](int li) { return as_global(reinterpret_cast<const uint32_t*>(plans + li))[lane]; };
  auto unpack = [&](uint32_t pv, int li) {
    auto F = [&](size_t byte_off) { return (int)__builtin_amdgcn_readlane((int)pv, (int)(byte_off / 4)); };
    static_assert(offsetof(DropPlan, a1_off) + 8 <= 256, "DropPlan layout");
    SmallItem o;
    o.li = li;
    o.r1 = F(offsetof(DropPlan, r1)); o.r2 = F(offsetof(DropPlan, r2)); o.tw = F(offsetof(DropPlan, tw)); o.th = F(offsetof(DropPlan, th));
    o.pw = F(offsetof(DropPlan, ew)); o.ph = F(offsetof(DropPlan, eh)); o.epitch = F(offsetof(DropPlan, epitch)); o.epad = F(offsetof(DropPlan, epad));
    o.a0 = (long long)(((unsigned long long)(unsigned)F(offsetof(DropPlan, a0_off) + 4) << 32) | (unsigned)F(offsetof(DropPlan, a0_off)));
    o.a1 = (long long)(((unsigned long long)(unsigned)F(offsetof(DropPlan, a1_off) + 4) << 32) | (unsigned)F(offsetof(DropPlan, a1_off)));
    return o;
  };
  double2_v R[4];                     // the raw tile of the drop AFTER the current one, lane-linear
  double nw1 = 0.0, nw2 = 0.0;        // its weights: lane l holds w(distance l) of each axis
  auto issue = [&](const SmallItem& d) {
    const global_ptr<const double2_v> raw2 = as_global(reinterpret_cast<const double2_v*>(sc.arena + d.a0));   // tiles start on 128-byte lines
    const int n = d.tw * d.th;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k2 = lane + 64 * j;
      R[j] = double2_v{0.0, 0.0};
      if (2 * k2 < n) R[j] = raw2[k2];               // (an odd tile's last pair reads one double of the arena's own padding)
    }
    const global_ptr<const double> wt = as_global(sc.wtab + ((int64_t)f * max_drops + d.li) * 2 * (BR_MAX + 1));   // hw[k] = w(|k - r|), k_blur_weights
    nw1 = (lane <= d.r1) ? wt[d.r1 - lane] : 0.0;
    nw2 = (d.r2 > 0 && lane <= d.r2) ? wt[(BR_MAX + 1) + d.r2 - lane] : 0.0;
  };
  PH_DECL
  // prologue: drop `it` entirely, the plan of the next one and the list entry of the one after that in flight
  const int li0 = __builtin_amdgcn_readfirstlane(list[it]);
  SmallItem cur = unpack(load_plan(li0), li0);
  issue(cur);
  bool has_next = it + stride < n_items;
  int li1 = has_next ? __builtin_amdgcn_readfirstlane(list[it + stride]) : li0;
  uint32_t pv_next = load_plan(li1);
  int32_t li2_v = (it + 2 * stride < n_items) ? list[it + 2 * stride] : 0;
  for (;;) {
    const int r1 = cur.r1, r2 = cur.r2, pw = cur.pw, ph = cur.ph, tw = cur.tw, th = cur.th;   // effective tile pw x ph
    const int php = (ph + 3) & ~3, hi = php + 2 * r1, yp = blur_y_pitch(pw, r2);
    double* tile = sc.arena + cur.a1;               // finished effective tile
    // ---- stage: X = [2*r1 zero rows | raw tile | zero rows], tw columns; Y cleared as a whole (the row pass writes its
    //      tw columns at column offset 2*r2) ----
    {
      const int n = tw * th, top = 2 * r1 * tw;     // (top is even: the data start on a 16-byte boundary of X)
      double2* X2 = reinterpret_cast<double2*>(X);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k2 = lane + 64 * j;
        if (2 * k2 < n) X2[(top >> 1) + k2] = make_double2(R[j].x, 2 * k2 + 1 < n ? R[j].y : 0.0);
      }
      for (int i = lane; i < (top >> 1); i += 64) X2[i] = make_double2(0.0, 0.0);
      const int e0 = (top + n + 1) >> 1, e1 = (tw * hi + 1) >> 1;       // pairs behind the data (X holds tw * hi <= BS_X doubles, BS_X even)
      for (int i = e0 + lane; i < e1; i += 64) X2[i] = make_double2(0.0, 0.0);
      const int nz2 = (yp * php + 1) >> 1;          // (yp * php is even and <= BS_Y)
      double2* Y2 = reinterpret_cast<double2*>(Y);
      for (int i = lane; i < nz2; i += 64) Y2[i] = make_double2(0.0, 0.0);
    }
    const double w1 = nw1, w2 = nw2;
    wave_lds_sync();
    PH(0)                                           // waiting for the prefetched tile + staging it
    // ---- the next drop's loads go out now and land under this drop's arithmetic (its plan arrived with this drop's
    //      tile: vector loads return in order) ----
    SmallItem nxt = cur;
    if (has_next) {
      nxt = unpack(pv_next, li1);
      issue(nxt);
      if (it + 2 * stride < n_items) {
        li1 = __builtin_amdgcn_readfirstlane(li2_v);
        pv_next = load_plan(li1);
        if (it + 3 * stride < n_items) li2_v = list[it + 3 * stride];
      }
    }
    const float inv_tw = 1.0f / (float)tw;
    // axis 0 (rows): a lane owns data column x and four consecutive rows
    const int nv4 = (php >> 2) * tw;                 // four-row blocks of the row pass
    if (nv4 <= 16) {                                 // (r06) a quarter of the wave or less: an output per lane, php * tw <= 64 lanes
      const int idx = lane;
      if (idx < 4 * nv4) {
        const int row = (int)(((float)idx + 0.5f) * inv_tw), x = idx - row * tw;
        double a[1];
        blur_n<1>(X + (__mul24(row + r1, tw) + x), tw, [&](int k) { return readlane_f64(w1, r1 - k); }, r1, a);
        Y[__mul24(row, yp) + 2 * r2 + x] = a[0];
      }
    } else if (nv4 <= 32) {                          // half the wave: two rows per lane
      const int idx = lane;
      if (idx < 2 * nv4) {
        const int rb = (int)(((float)idx + 0.5f) * inv_tw), x = idx - rb * tw;
        double a[2];
        blur_n<2>(X + (__mul24(2 * rb + r1, tw) + x), tw, [&](int k) { return readlane_f64(w1, r1 - k); }, r1, a);
        double* o = Y + (__mul24(2 * rb, yp) + 2 * r2 + x);
        o[0] = a[0];
        o[yp] = a[1];
      }
    } else {
      const int nv = nv4;
      for (int idx = lane; idx < nv; idx += 64) {
        const int rb = (int)(((float)idx + 0.5f) * inv_tw), x = idx - rb * tw;
        const double* c0 = X + (__mul24(4 * rb + r1, tw) + x);       // (24-bit multiply-add: see the note at the prefetch below)
        double a0, a1, a2, a3;
        blur4(c0, tw, [&](int k) { return readlane_f64(w1, r1 - k); }, r1, a0, a1, a2, a3);
        double* o = Y + (__mul24(4 * rb, yp) + 2 * r2 + x);          // Y has php rows: the slack rows are never read
        o[0] = a0;
        o[yp] = a1;
        o[2 * yp] = a2;
        o[3 * yp] = a3;
      }
    }
    wave_lds_sync();
    PH(1)                                           // row pass
    // axis 1 (columns) -> global: a lane owns row y and four consecutive columns; lanes run down the rows
    const int ncb = (pw + 3) >> 2, nh = ncb * ph;
    const float inv_ph = 1.0f / (float)ph;
    if (nh <= 16 && r2 > 0) {                        // (r06) a column per lane: pw * ph <= 64 lanes
      const int idx = lane;
      if (idx < pw * ph) {
        const int x = (int)(((float)idx + 0.5f) * inv_ph), yq = idx - x * ph;
        double a[1];
        blur_n<1>(Y + (__mul24(yq, yp) + x + r2), 1, [&](int k) { return readlane_f64(w2, r2 - k); }, r2, a);
        tile[(int64_t)yq * cur.epitch + cur.epad + x] = a[0];
      }
    } else if (nh <= 32 && r2 > 0) {                 // two columns per lane
      const int idx = lane, nc2 = (pw + 1) >> 1;
      if (idx < nc2 * ph) {
        const int cb = (int)(((float)idx + 0.5f) * inv_ph), yq = idx - cb * ph, xo = 2 * cb;
        double a[2];
        blur_n<2>(Y + (__mul24(yq, yp) + xo + r2), 1, [&](int k) { return readlane_f64(w2, r2 - k); }, r2, a);
        double* o = tile + (int64_t)yq * cur.epitch + cur.epad + xo;
        o[0] = a[0];
        if (xo + 1 < pw) o[1] = a[1];
      }
    } else {
      for (int idx = lane; idx < nh; idx += 64) {
        const int cb = (int)(((float)idx + 0.5f) * inv_ph), yq = idx - cb * ph;
        const double* c0 = Y + (__mul24(yq, yp) + 4 * cb + r2);
        double a0, a1, a2, a3;
        if (r2 > 0) {
          blur4(c0, 1, [&](int k) { return readlane_f64(w2, r2 - k); }, r2, a0, a1, a2, a3);
        } else {
          a0 = c0[0]; a1 = c0[1]; a2 = c0[2]; a3 = c0[3];
        }
        const int xo = 4 * cb;
        double* o = tile + (int64_t)yq * cur.epitch + cur.epad + xo;
        if (xo + 3 < pw) {
          o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
        } else {
          o[0] = a0;
          if (xo + 1 < pw) o[1] = a1;
          if (xo + 2 < pw) o[2] = a2;
        }
      }
    }
    wave_lds_sync();
    PH(2)                                           // column pass + store
    if (!has_next) break;
    it += stride;
    cur = nxt;
    has_next = it + stride < n_items;
  }
  PH_FLUSH(2)
}

// ---- large defocus radii (r > BR_MAX: drops a few centimetres from a fast lens) and tiles no LDS layout takes ----
// The raw tile is small, the blurred one large: an output sample of the row pass has at most 2 * th taps whose operands
// are not both outside the raw tile (exact zeros; acc + (0 + 0) * w == acc), the column pass 2 * tw.  Taps are visited in
// the reference's order (ii ascending), only the all-zero ones are left out, so the sums keep their bits.
//   pass 0 (rows)  work unit = (drop, raw column): the column (th values) in LDS, one thread per output row
//                  -> tmp[eh][tw] (dense, in the scratch tile k_plan reserved behind the finished tile)
//   pass 1 (cols)  work unit = (drop, group of BIG_ROWS output rows): their tmp rows in LDS, one thread per output sample
constexpr int BIG_GROUPS = 64;      // work units per drop and pass (a unit strides over the columns / row groups)
constexpr int BIG_ROWS = 16;
constexpr int BIG_COL_LDS = 2048;   // longest raw column staged in LDS (longer: read from global memory)

// f(ii) for ii ascending over ([a0, a1] u [b0, b1]) n [-r, -1]
template <class F>
__device__ inline void visit_taps(int a0, int a1, int b0, int b1, int r, F f) {
  a0 = imax(a0, -r); a1 = imin(a1, -1);
  b0 = imax(b0, -r); b1 = imin(b1, -1);
  if (a0 > a1) { a0 = b0; a1 = b1; b0 = 0; b1 = -1; }                  // A empty: B alone
  if (b0 <= b1 && b0 < a0) { int t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; }
  if (b0 <= b1 && b0 <= a1 + 1) { a1 = imax(a1, b1); b0 = 0; b1 = -1; }  // overlapping / adjacent: one run
  for (int ii = a0; ii <= a1; ii++) f(ii);
  for (int ii = b0; ii <= b1; ii++) f(ii);
}

// Both Gaussian half tables of a large-radius drop, once per drop (a table is up to 417 exponentials and a sequential
// normalisation sum: rebuilt by each of a drop's 128 work units it was 40 % of the large-radius blur)
constexpr int SLOW_CAP = 256;       // large-radius drops per frame whose tables are kept (the rest build their own)
__global__ __launch_bounds__(256) void k_blur_big_weights(const FrameDesc* frames, int max_drops, Scratch sc) {
  const int f = blockIdx.y;
  __shared__ double hw[MAX_R + 1];
  const int n_items = imin(sc.counts[f * 8 + 3], SLOW_CAP);
  for (int w = blockIdx.x; w < 2 * n_items; w += gridDim.x) {
    const int it = w >> 1, axis = w & 1;
    const DropPlan& p = sc.plan[(int64_t)f * max_drops + sc.list_slow[(int64_t)f * max_drops + it]];
    const int r = axis ? p.r2 : p.r1;
    __syncthreads();
    if (r > 0) {
      gauss_half_table(axis ? p.sig2 : p.sig1, r, hw);
      double* o = sc.wtab_big + (((int64_t)f * SLOW_CAP + it) * 2 + axis) * (MAX_R + 1);
      for (int k = threadIdx.x; k <= r; k += 256) o[k] = hw[k];
    }
  }
}

template <int AXIS>
__global__ __launch_bounds__(256) void k_blur(const FrameDesc* frames, int max_drops, Scratch sc) {
  const int f = blockIdx.y, t = threadIdx.x;
  __shared__ double hw[MAX_R + 1];
  __shared__ double s_in[BIG_COL_LDS];
  const int n_items = sc.counts[f * 8 + 3];
  int cur = -1;
  for (int w = blockIdx.x; w < n_items * BIG_GROUPS; w += gridDim.x) {
    const int it = w / BIG_GROUPS, g = w - it * BIG_GROUPS;
    const int64_t gi = (int64_t)f * max_drops + sc.list_slow[(int64_t)f * max_drops + it];
    const DropPlan& p = sc.plan[gi];
    const int r = AXIS == 0 ? p.r1 : p.r2;
    const int tw = p.tw, th = p.th, ew = p.ew, eh = p.eh, r1 = p.r1, r2 = p.r2;
    if (AXIS == 0 ? g >= tw : g * BIG_ROWS >= eh) continue;            // nothing for this unit
    const double* raw = sc.arena + p.a0_off;
    double* fin = sc.arena + p.a1_off;               // pitch p.epitch, first column p.epad
    double* tmp = fin + (((int64_t)p.epitch * eh + 15) & ~15LL);   // scratch reserved by k_plan for slow drops: eh x tw
    __syncthreads();
    if (cur != it) {
      if (r > 0) {
        if (it < SLOW_CAP) {                               // made once per drop by k_blur_big_weights
          const double* wt = sc.wtab_big + (((int64_t)f * SLOW_CAP + it) * 2 + AXIS) * (MAX_R + 1);
          for (int k = t; k <= r; k += 256) hw[k] = wt[k];
          __syncthreads();
        } else {
          gauss_half_table(AXIS == 0 ? p.sig1 : p.sig2, r, hw);
        }
      }
      cur = it;
    }
    if (AXIS == 0) {
      for (int rx = g; rx < tw; rx += BIG_GROUPS) {
        const bool lds = th <= BIG_COL_LDS;
        __syncthreads();
        if (lds)
          for (int k = t; k < th; k += 256) s_in[k] = raw[(int64_t)k * tw + rx];
        __syncthreads();
        auto R = [&](int yy) -> double {               // row yy of the effective tile: raw row yy - r1
          const int ry = yy - r1;
          return (ry >= 0 && ry < th) ? (lds ? s_in[ry] : raw[(int64_t)ry * tw + rx]) : 0.0;
        };
        for (int y = t; y < eh; y += 256) {
          double acc = R(y) * hw[r];
          visit_taps(r1 - y, r1 - y + th - 1, y - r1 - th + 1, y - r1, r, [&](int ii) { acc = acc + (R(y + ii) + R(y - ii)) * hw[ii + r]; });
          tmp[(int64_t)y * tw + rx] = acc;
        }
      }
    } else {
      for (int y0 = g * BIG_ROWS; y0 < eh; y0 += BIG_GROUPS * BIG_ROWS) {
        const int nr = imin(BIG_ROWS, eh - y0);
        const bool lds = nr * tw <= BIG_COL_LDS;
        __syncthreads();
        if (lds)
          for (int k = t; k < nr * tw; k += 256) s_in[k] = tmp[(int64_t)y0 * tw + k];
        __syncthreads();
        for (int idx = t; idx < nr * ew; idx += 256) {
          const int yy = idx / ew, x = idx - yy * ew;
          auto T = [&](int xx) -> double {             // column xx of the effective tile: raw column xx - r2
            const int rx = xx - r2;
            return (rx >= 0 && rx < tw) ? (lds ? s_in[yy * tw + rx] : tmp[(int64_t)(y0 + yy) * tw + rx]) : 0.0;
          };
          double acc;
          if (r2 > 0) {
            acc = T(x) * hw[r];
            visit_taps(r2 - x, r2 - x + tw - 1, x - r2 - tw + 1, x - r2, r, [&](int ii) { acc = acc + (T(x + ii) + T(x - ii)) * hw[ii + r]; });
          } else {
            acc = T(x);
          }
          fin[(int64_t)(y0 + yy) * p.epitch + p.epad + x] = acc;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// compositor: one block per 16x16 screen tile, drops applied in reference order
// ---------------------------------------------------------------------------
// Coarse binning: one block per 64x64 coarse tile scans every drop's footprint once and keeps,
// IN DROP ORDER (ballot + prefix popcount, no atomics, no sort), the ones that touch it.
__global__ __launch_bounds__(256) void k_bin(const FrameDesc* frames, Dims dm, int max_drops, int ctiles_x, int nct, Scratch sc) {
  const int f = blockIdx.y, ct = blockIdx.x;
  const int cty = ct / ctiles_x, ctx_ = ct - cty * ctiles_x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = frames[f].n_drops;
  const int x0 = ctx_ * CTILE, y0 = cty * CTILE, x1 = min(x0 + CTILE, dm.W), y1 = min(y0 + CTILE, dm.H);
  const int4* bbox = sc.bbox + (int64_t)f * max_drops;
  uint16_t* out = sc.clist + ((int64_t)f * nct + ct) * max_drops;
  __shared__ int s_cnt[4];
  int total = 0;
  for (int base = 0; base < n; base += 256) {
    const int i = base + t;
    bool hit = false;
    if (i < n) {
      const int4 bb = bbox[i];
      hit = bb.x < x1 && bb.z > x0 && bb.y < y1 && bb.w > y0;
    }
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = total;
    for (int w = 0; w < 4; w++) {
      const int cw = s_cnt[w];
      if (w < wave) off += cw;
      total += cw;
    }
    if (hit) out[off + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    __syncthreads();
  }
  if (t == 0) sc.ccount[(int64_t)f * nct + ct] = total;
}

// r05: the same lists, a workgroup per ROW of coarse tiles.  k_bin tests every footprint against every coarse tile (120 tiles
// x 7200 drops per KITTI frame, two block barriers per 256 tests).  Here the drops are taken in segments of BIN_SEG: the
// workgroup first keeps, in drop order, the ones that reach its 64-pixel row of the frame (a quarter of them) with their x
// range in LDS; then every WAVE takes tiles of the row by itself and walks that short list with ballots -- no block barrier,
// a third of the tests.  Same lists, same order (tests/test_gpu_properties.py compares every output with RR_OPT_BIN_ROWS 0).
constexpr int BIN_SEG = 8192;
__global__ __launch_bounds__(256) void k_bin_rows(const FrameDesc* frames, Dims dm, int max_drops, int ctiles_x, int nct, Scratch sc) {
  const int f = blockIdx.y, cty = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = frames[f].n_drops;
  const int y0 = cty * CTILE, y1 = min(y0 + CTILE, dm.H);
  const int4* bbox = sc.bbox + (int64_t)f * max_drops;
  __shared__ uint16_t s_idx[BIN_SEG], s_x0[BIN_SEG], s_x1[BIN_SEG];
  __shared__ int s_cnt[4];
  __shared__ int s_total[64];                            // entries written so far, per tile of the row (ctiles_x <= 64: frames up to 4096 wide)
  for (int k = t; k < ctiles_x; k += 256) s_total[k] = 0;
  for (int seg = 0; seg < n; seg += BIN_SEG) {
    const int seg_n = min(BIN_SEG, n - seg);
    int m = 0;                                           // drops of the segment that reach this row
    for (int base = 0; base < seg_n; base += 256) {
      const int i = seg + base + t;
      bool hit = false;
      int4 bb = make_int4(0, 0, 0, 0);
      if (base + t < seg_n) {
        bb = bbox[i];
        hit = bb.x < bb.z && bb.y < y1 && bb.w > y0;      // (an empty footprint reaches nothing)
      }
      const unsigned long long bal = __ballot(hit);
      if (lane == 0) s_cnt[wave] = __popcll(bal);
      __syncthreads();
      int off = m;
      for (int w = 0; w < 4; w++) {
        const int cw = s_cnt[w];
        if (w < wave) off += cw;
        m += cw;
      }
      if (hit) {
        const int o = off + __popcll(bal & ((1ull << lane) - 1ull));
        s_idx[o] = (uint16_t)i;
        s_x0[o] = (uint16_t)imax(bb.x, 0);
        s_x1[o] = (uint16_t)imin(bb.z, 65535);
      }
      __syncthreads();
    }
    // every wave: its tiles of the row against the row's list
    for (int ctx_ = wave; ctx_ < ctiles_x; ctx_ += 4) {
      const int x0 = ctx_ * CTILE, x1 = min(x0 + CTILE, dm.W);
      uint16_t* out = sc.clist + ((int64_t)f * nct + (cty * ctiles_x + ctx_)) * max_drops;
      int total = s_total[ctx_];
      for (int base = 0; base < m; base += 64) {
        const int k = base + lane;
        const bool hit = k < m && (int)s_x0[k] < x1 && (int)s_x1[k] > x0;
        const unsigned long long bal = __ballot(hit);
        if (hit) out[total + __popcll(bal & ((1ull << lane) - 1ull))] = s_idx[k];
        total += __popcll(bal);
      }
      if (lane == 0) s_total[ctx_] = total;              // (a tile belongs to one wave: no one else reads it before the end)
    }
    __syncthreads();                                     // the list is free for the next segment
  }
  __syncthreads();
  for (int k = t; k < ctiles_x; k += 256) sc.ccount[(int64_t)f * nct + cty * ctiles_x + k] = s_total[k];
}

// RR_OPT_WILD_PIXELS: the reference blends a drop over its whole PADDED rectangle (bad_weather.py:429-446); outside the tile the
// compositor reads, the drop image is exact zeros and the blend reduces to np.clip(pixel, 0, 1) -- a no-op for the values in
// [0, 1] that rainy_bg holds by contract, which is why the compositor never visits the pad.  For a caller whose rainy_bg holds
// anything, that clip is the one thing the pad does, and only where it comes BEFORE the pixel's first real blend (every blend
// ends with a clip, and a clip of a clipped value changes nothing).  One wave per drop marks, per pixel, the lowest index of a
// composited drop that covers it with its pad (pad_first) / with its tile (eff_first); the compositors clip a pixel's input
// value first where pad_first < eff_first.
__global__ __launch_bounds__(256) void k_pad_visits(const FrameDesc* frames, Dims dm, int max_drops, Scratch sc) {
  const int f = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= frames[f].n_drops) return;
  const int64_t gi = (int64_t)f * max_drops + i;
  if (!sc.blended[gi]) return;
  const DropPlan& p = sc.plan[gi];
  const int4 e = sc.bbox[gi];
  const int w = p.vis_w, n = p.vis_w * p.vis_h;
  const int64_t base = (int64_t)f * dm.H * dm.W;
  for (int k = lane; k < n; k += 64) {
    const int y = p.vis_y0 + k / w, x = p.vis_x0 + k % w;
    if (x < 0 || y < 0 || x >= dm.W || y >= dm.H) continue;
    const bool in_tile = x >= e.x && x < e.z && y >= e.y && y < e.w;
    atomicMin((in_tile ? sc.eff_first : sc.pad_first) + base + (int64_t)y * dm.W + x, i);
  }
}
__device__ inline double clip_unit_np(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }   // np.clip(v, 0, 1): a NaN stays

__global__ __launch_bounds__(256) void k_composite(const FrameDesc* frames, Dims dm, rr_camera cam, int max_drops, int tiles_x,
                                                   int tiles_y, int ctiles_x, int nct, Scratch sc) {
  const int f = blockIdx.y;
  // Workgroups go to the 8 XCDs round robin (blockIdx.x % 8), each with its own L2.  XCD k takes the k-th eighth of
  // the frame's screen tiles (row-major), so the workgroups that share an L2 cover one compact band of the image: a
  // drop's alpha tile, split over neighbouring screen tiles, is fetched from HBM once, not once per screen tile.
  const int ntiles = tiles_x * tiles_y, per_xcd = (ntiles + 7) / 8;
  const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
  if (tile >= ntiles) return;
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const FrameDesc& fr = frames[f];
  const int tx0 = txi * TILE, ty0 = tyi * TILE, tx1 = min(tx0 + TILE, dm.W), ty1 = min(ty0 + TILE, dm.H);
  // a wave owns one 8x8 quarter of the 16x16 tile (drops are narrow and tall: a square wave footprint has the most
  // lanes inside a drop's rectangle) and walks its OWN ordered list: only the drops that reach its quarter
  static_assert(TILE == 16, "k_composite: four 8x8 quarters");
  const int px = tx0 + 8 * (wave & 1) + (lane & 7), py = ty0 + 8 * (wave >> 1) + (lane >> 3);
  const bool live = px < dm.W && py < dm.H;
  const int64_t pix = (int64_t)py * dm.W + px;
  double c[3] = {0, 0, 0}, m = 0.0;
  double sum_b = 0.0;                // this pixel's share of sum(bg) for the mean shift (generator.py:462)
  if (live) {
    load_px3(fr.rainy_bg, rainy_kind(fr), pix, c);
    if (fr.bg == fr.rainy_bg) {      // no fog pre-pass: one read serves both
      sum_b = (c[0] + c[1]) + c[2];
    } else {
      double b[3];
      load_px3(fr.bg, bg_kind(fr), pix, b);
      sum_b = (b[0] + b[1]) + b[2];
    }
  }
  if (sc.pad_first && live) {        // RR_OPT_WILD_PIXELS: some drop's all-zero pad reaches this pixel before any tile does
    const int64_t q = (int64_t)f * dm.H * dm.W + pix;
    if (sc.pad_first[q] < sc.eff_first[q])
      for (int k = 0; k < 3; k++) c[k] = clip_unit_np(c[k]);
  }
  // depth-occlusion option (default off; not part of the reference's output): a drop farther than the scene at a
  // pixel is hidden there
  double scene = 1.0e300;
  if (fr.depth && live)
    scene = fr.depth_f64 ? as_global((const double*)fr.depth)[pix] : (double)as_global((const float*)fr.depth)[pix];
  // the short form of the blend (see step) needs finite pixel values and an exposure whose reciprocal divides exactly
  const double ex = cam.exposure_s, ex_rcp = 1.0 / ex;
  const bool tame_px = !live || (fabs(c[0]) < 1.0e50 && fabs(c[1]) < 1.0e50 && fabs(c[2]) < 1.0e50);
  const bool slow_wave = __ballot(!tame_px) != 0ull || (__double_as_longlong(ex) & 0xFFFFFFFFFFFFFLL) == 0xFFFFFFFFFFFFFLL ||
                         !(ex > 1.0e-100 && ex < 1.0e100);
  const CompRec* comp = sc.comp + (int64_t)f * max_drops;
  const int4* bbox = sc.bbox + (int64_t)f * max_drops;
  const double* arena = sc.arena;
  const int ct = (ty0 / CTILE) * ctiles_x + (tx0 / CTILE);
  const uint16_t* clist = sc.clist + ((int64_t)f * nct + ct) * max_drops;
  const int n = sc.ccount[(int64_t)f * nct + ct];
  __shared__ int s_list[4][256];                                // per quarter: drop indices in compositing order
  __shared__ int s_cnt[4][4];                                   // [wave][quarter] hits among the wave's 64 candidates
  const int xm = tx0 + 8, ym = ty0 + 8;
  for (int base = 0; base < n; base += 256) {
    const int k = base + t;
    unsigned hitm = 0;                                          // bit q: the drop's footprint reaches quarter q
    int i = 0;
    if (k < n) {
      i = clist[k];
      const int4 bb = bbox[i];
      if (bb.x < tx1 && bb.z > tx0 && bb.y < ty1 && bb.w > ty0) {
        const unsigned xl = bb.x < xm, xr = bb.z > xm, yt = bb.y < ym, yb = bb.w > ym;
        hitm = (xl & yt) | ((xr & yt) << 1) | ((xl & yb) << 2) | ((xr & yb) << 3);
      }
    }
    unsigned long long bal[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bal[q] = __ballot((hitm >> q) & 1u);
    if (lane < 4) s_cnt[wave][lane] = __popcll(lane == 0 ? bal[0] : (lane == 1 ? bal[1] : (lane == 2 ? bal[2] : bal[3])));
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int off = 0, tq = 0;
      for (int w = 0; w < 4; w++) {
        const int cw = s_cnt[w][q];
        if (w < wave) off += cw;
        tq += cw;
      }
      if ((hitm >> q) & 1u) s_list[q][off + __popcll(bal[q] & ((1ull << lane) - 1ull))] = i;
      if (q == wave) total = tq;
    }
    __syncthreads();
    const int* lst = s_list[wave];
    total = __builtin_amdgcn_readfirstlane(total);              // the same in every lane: says so to the compiler (uniform loops)
    // Software pipeline over the ordered entries, three stages deep: while entry e is blended, the alpha sample of
    // entry e+1 is in flight and so is the RECORD of entry e+2.  A record is wave-uniform: it is fetched whole with
    // wide scalar loads into SGPRs (fetch; the constant address space keeps them on the scalar unit) one iteration
    // before its first use, and the footprint test is branch-free.  (Measured and rejected: records staged through
    // LDS with four samples in flight per lane -- 20 % slower.)
    struct RecS {
      int x0, y0, x1, y1, ox, oy, pitch, slow;
      long long off;
      double tau, g, k0, k1, k2, z;
    };
    auto fetch = [&](int idx) {
      const const_ptr<CompRec> r = as_constant(comp) + __builtin_amdgcn_readfirstlane(idx);
      RecS o{r->x0, r->y0, r->x1, r->y1, r->ox, r->oy, r->pitch, r->pad, (long long)r->off, r->tau_one, r->g, r->K[0], r->K[1], r->K[2], r->zdist};
      return o;
    };
    auto inside = [&](const RecS& r) { return live & (px >= r.x0) & (px < r.x1) & (py >= r.y0) & (py < r.y1); };
    // (every lane loads: one outside the footprint reads the tile's first sample -- the same line for all of them -- so
    // that the number of loads in flight is known and the wait before a blend covers the OLDER load only)
    auto sample = [&](const RecS& r, bool in) {
      const int64_t o = in ? (int64_t)(py + r.oy) * r.pitch + (px + r.ox) : 0;
      return arena[r.off + o];
    };
    // One step: entry e (record rcur, alpha Acur) is blended while entry e+1 (record rnext, fetched a step ago) is
    // sampled and the record of entry e+2 is fetched.  The loop below is unrolled three times over three register
    // sets, so nothing is moved between steps: a load is waited for where its value is first used, one step later.
    auto step = [&](int e, const RecS& rcur, double Acur, bool incur, const RecS& rnext, double& Anext, bool& innext, RecS& rfetch,
                    int& i_nn) {
      innext = inside(rnext) & (e + 1 < total);                  // (past the end: entry 0 again, never blended)
      Anext = sample(rnext, innext);
      rfetch = fetch(i_nn);
      i_nn = lst[e + 3 < total ? e + 3 : 0];                     // list index of entry e + 3, read one step ahead
      if (incur && !(rcur.z > scene)) {
        if (slow_wave | rcur.slow) {                             // (wave-uniform) the literal form
          const double K[3] = {rcur.k0, rcur.k1, rcur.k2};
          blend_pixel(Acur, rcur.tau, cam.exposure_s, rcur.g, K, c, m);
        } else {
          // Same result bits as blend_pixel (bad_weather.py:443-446,450) from half the instructions, valid because every
          // factor is finite here (k_colour's flag, the pixel test above) so no NaN can arise and clip == clamp:
          //   t = (A * tau) / exposure: with y = RN(1 / d), q0 = a * y, the value fma(fma(-q0, d, a), y, q0) is the
          //   correctly rounded a / d (Markstein; d's significand not all ones -- part of slow_wave -- and a, a / d
          //   in the normal range or zero: tau is 0 or > 1e-200, an alpha sample is 0 or far above 1e-40).
          const double a = Acur * rcur.tau;
          const double q0 = a * ex_rcp;
          const double u = 1.0 - __builtin_fma(__builtin_fma(-q0, ex, a), ex_rcp, q0);
          const double v0 = u * c[0] + (Acur * rcur.k0) * rcur.g, v1 = u * c[1] + (Acur * rcur.k1) * rcur.g,
                       v2 = u * c[2] + (Acur * rcur.k2) * rcur.g;
          c[0] = __builtin_fmin(__builtin_fmax(v0, 0.0), 1.0);
          c[1] = __builtin_fmin(__builtin_fmax(v1, 0.0), 1.0);
          c[2] = __builtin_fmin(__builtin_fmax(v2, 0.0), 1.0);
          m = m + Acur;
        }
      }
    };
    if (total > 0) {
      RecS R0 = fetch(lst[0]), R1 = fetch(lst[total > 1 ? 1 : 0]), R2 = R0;
      int i_nn = lst[total > 2 ? 2 : 0];
      bool in0 = inside(R0), in1 = false, in2 = false;
      double A0 = sample(R0, in0), A1 = 0.0, A2 = 0.0;
      for (int e = 0; e < total; e += 3) {
        step(e, R0, A0, in0, R1, A1, in1, R2, i_nn);
        if (e + 1 >= total) break;
        step(e + 1, R1, A1, in1, R2, A2, in2, R0, i_nn);
        if (e + 2 >= total) break;
        step(e + 2, R2, A2, in2, R0, A0, in0, R1, i_nn);
      }
    }
    __syncthreads();
  }
  double sum_c = 0.0;
  if (live) {
    const global_ptr<double> o = as_global(fr.comp_out) + pix * 3;
    o[0] = c[0];
    o[1] = c[1];
    o[2] = c[2];
    if (fr.mask_f64) as_global(fr.mask_f64)[pix] = m;
    if (fr.mask_i32) as_global(fr.mask_i32)[pix] = (int32_t)floor(m * 255.0);
    sum_c = (c[0] + c[1]) + c[2];
  }
  __shared__ double ra[256], rb[256], rlo[256], rhi[256];
  ra[t] = sum_c;
  rb[t] = sum_b;
  rlo[t] = live ? m : 1.0e300;
  rhi[t] = live ? m : -1.0e300;
  __syncthreads();
  for (int ofs = 128; ofs > 0; ofs >>= 1) {
    if (t < ofs) {
      ra[t] += ra[t + ofs];
      rb[t] += rb[t + ofs];
      rlo[t] = dmin(rlo[t], rlo[t + ofs]);
      rhi[t] = dmax(rhi[t], rhi[t + ofs]);
    }
    __syncthreads();
  }
  if (t == 0) {
    double* part = sc.partial + ((int64_t)f * tiles_x * tiles_y + tile) * 4;
    part[0] = ra[0];
    part[1] = rb[0];
    part[2] = rlo[0];
    part[3] = rhi[0];
  }
}

// ---------------------------------------------------------------------------
// compositor with float colours (the default when the caller does not ask for the float64 composite)
// ---------------------------------------------------------------------------
// What must be bit-exact is the mask: it stays a float64 sum of float64 alpha samples in drop order.  The three colour
// channels only have to land within 1 LSB of a uint8 (BASELINE.json), and hundreds of float blends err by ~1e-5: they are
// carried in float -- full-rate VALU, packed two pixels at a time -- with the drop's factors pre-multiplied in its
// 64-byte record (one scalar 16-dword load).  A workgroup takes a 16 x 32 screen tile; a wave owns an 8 x 16 part and its
// own ordered drop list, a lane TWO pixels of one column, eight rows apart (drops are narrow and tall), so the scalar work
// per list entry is shared by 128 pixels.  A pixel outside an entry's footprint blends with alpha 0: (1 - 0) c + 0 = c, and
// the clamp leaves c alone as long as c is in [0, 1] -- true for the fog pre-pass' output (it ends with np.clip) and
// after every blend; a wave that finds anything else (NaN, out of range) among its pixels, and an entry whose factors are
// not tame, take the literal float64 blend_pixel per pixel instead.  The composite leaves as floats (12 B per pixel).
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int TILE32_H = 32;
// clamp(a * b + c, 0, 1) on both halves in one packed instruction (the compiler keeps the clamp as two extra instructions)
__device__ inline float2_t pk_fma_clamp(float2_t a, float2_t b, float2_t c) {
  float2_t d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int WPE, bool BATCH>      // WPE: waves per SIMD the register allocation is held to (RR_OPT_COMPOSITE_WAVES); BATCH: RR_OPT_COMPOSITE_BATCH
__global__ __launch_bounds__(256, WPE) void k_composite32(const FrameDesc* frames, Dims dm, rr_camera cam, int max_drops, int tiles_x,
                                                     int tiles_y, int ctiles_x, int nct, int64_t arena_cap, Scratch sc) {
  const int f = blockIdx.y;
  const int ntiles = tiles_x * tiles_y, per_xcd = (ntiles + 7) / 8;
  const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);      // XCD k: the k-th eighth of the tiles (shared L2)
  if (tile >= ntiles) return;
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  PH_DECL
  const FrameDesc& fr = frames[f];
  const int tx0 = txi * TILE, ty0 = tyi * TILE32_H, tx1 = min(tx0 + TILE, dm.W), ty1 = min(ty0 + TILE32_H, dm.H);
  const int px = tx0 + 8 * (wave & 1) + (lane & 7);
  const int py0 = ty0 + 16 * (wave >> 1) + (lane >> 3), py1 = py0 + 8;
  const bool live0 = px < dm.W && py0 < dm.H, live1 = px < dm.W && py1 < dm.H;
  const int64_t pix0 = (int64_t)py0 * dm.W + px, pix1 = (int64_t)py1 * dm.W + px;
  float2_t c0 = {0.f, 0.f}, c1 = {0.f, 0.f}, c2 = {0.f, 0.f};       // channel k of (pixel 0, pixel 1)
  double m0 = 0.0, m1 = 0.0;
  double sum_b = 0.0;
  bool ok_px = true;                 // every colour value of the lane's live pixels is in [0, 1]
  // Everything the wave needs before its first blend is REQUESTED first and looked at afterwards (r05: the phase clocks gave
  // a quarter of the kernel's wave time to this prologue when every load_px3 was waited for in turn): the first entry of the
  // tile's list, the pixels of both images as raw words (a pixel outside the frame reads the tile's first pixel: no lane
  // branches around a load), the scene depth, then the box of that first entry.
  const int ct = (ty0 / CTILE) * ctiles_x + (tx0 / CTILE);
  const uint16_t* clist = sc.clist + ((int64_t)f * nct + ct) * max_drops;
  const int4* bbox = sc.bbox + (int64_t)f * max_drops;
  const int i_first = imin((int)as_global(clist)[imin(t, max_drops - 1)], max_drops - 1);      // (past the list's end: any valid index)
  const int64_t pq = (int64_t)ty0 * dm.W + tx0, pq0 = live0 ? pix0 : pq, pq1 = live1 ? pix1 : pq;
  const int rk = rainy_kind(fr), bk = bg_kind(fr);
  const bool same_bg = fr.bg == fr.rainy_bg;
  RawPx qr0 = {0u, 0u, 0u, 0u, 0u, 0u}, qr1 = qr0, qb0 = qr0, qb1 = qr0;
  load_px3_raw2(fr.rainy_bg, rk, pq0, pq1, qr0, qr1);
  const int4 bb_first = bbox[i_first];
  if (!same_bg) load_px3_raw2(fr.bg, bk, pq0, pq1, qb0, qb1);
  double scene0 = 1.0e300, scene1 = 1.0e300;
  if (fr.depth) {
    if (fr.depth_f64) {
      scene0 = as_global((const double*)fr.depth)[pq0];
      scene1 = as_global((const double*)fr.depth)[pq1];
    } else {
      const float d0 = as_global((const float*)fr.depth)[pq0], d1 = as_global((const float*)fr.depth)[pq1];
      scene0 = (double)d0;
      scene1 = (double)d1;
    }
    scene0 = live0 ? scene0 : 1.0e300;
    scene1 = live1 ? scene1 : 1.0e300;
  }
  {
    double in0[3], in1[3];
    decode_px3(qr0, rk, in0);
    decode_px3(qr1, rk, in1);
    if (!live0) in0[0] = in0[1] = in0[2] = 0.0;
    if (!live1) in1[0] = in1[1] = in1[2] = 0.0;
    if (same_bg) {
      sum_b = ((in0[0] + in0[1]) + in0[2]) + ((in1[0] + in1[1]) + in1[2]);
    } else {
      double b[3];
      if (live0) {
        decode_px3(qb0, bk, b);
        sum_b = (b[0] + b[1]) + b[2];
      }
      if (live1) {
        decode_px3(qb1, bk, b);
        sum_b += (b[0] + b[1]) + b[2];
      }
    }
    if (sc.pad_first) {              // RR_OPT_WILD_PIXELS (see k_pad_visits)
      const int64_t q = (int64_t)f * dm.H * dm.W;
      if (live0 && sc.pad_first[q + pix0] < sc.eff_first[q + pix0])
        for (int k = 0; k < 3; k++) in0[k] = clip_unit_np(in0[k]);
      if (live1 && sc.pad_first[q + pix1] < sc.eff_first[q + pix1])
        for (int k = 0; k < 3; k++) in1[k] = clip_unit_np(in1[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) ok_px = ok_px && in0[k] >= 0.0 && in0[k] <= 1.0 && in1[k] >= 0.0 && in1[k] <= 1.0;
    c0 = float2_t{(float)in0[0], (float)in1[0]};
    c1 = float2_t{(float)in0[1], (float)in1[1]};
    c2 = float2_t{(float)in0[2], (float)in1[2]};
  }
  const bool has_depth = fr.depth != nullptr;
  const bool slow_wave = __ballot(!ok_px) != 0ull;
  const CompRec32* comp = sc.comp32 + (int64_t)f * max_drops;
  const CompRec* comp64 = sc.comp + (int64_t)f * max_drops;
  const double* arena = sc.arena;
  const int n = sc.ccount[(int64_t)f * nct + ct];
  __shared__ int s_list[4][256];
  __shared__ int s_cnt[4][4];
  const int xm = tx0 + 8, ym = ty0 + 16;
  const float2_t one2 = {1.f, 1.f};
  const int64_t zero_at = (int64_t)f * arena_cap;                  // the frame's all-zero arena line (k_scan)
  PH_WAITVM
  PH(0)
  for (int base = 0; base < n; base += 256) {
    const int k = base + t;
    unsigned hitm = 0;
    int i = 0;
    if (k < n) {
      i = base == 0 ? i_first : (int)clist[k];
      const int4 bb = base == 0 ? bb_first : bbox[i];
      if (bb.x < tx1 && bb.z > tx0 && bb.y < ty1 && bb.w > ty0) {
        const unsigned xl = bb.x < xm, xr = bb.z > xm, yt = bb.y < ym, yb = bb.w > ym;
        hitm = (xl & yt) | ((xr & yt) << 1) | ((xl & yb) << 2) | ((xr & yb) << 3);
      }
    }
    unsigned long long bal[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bal[q] = __ballot((hitm >> q) & 1u);
    if (lane < 4) s_cnt[wave][lane] = __popcll(lane == 0 ? bal[0] : (lane == 1 ? bal[1] : (lane == 2 ? bal[2] : bal[3])));
    PH(1)
    __syncthreads();
    PH(2)
    int total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int off = 0, tq = 0;
      for (int w = 0; w < 4; w++) {
        const int cw = s_cnt[w][q];
        if (w < wave) off += cw;
        tq += cw;
      }
      if ((hitm >> q) & 1u) s_list[q][off + __popcll(bal[q] & ((1ull << lane) - 1ull))] = i;
      if (q == wave) total = tq;
    }
    PH(1)
    __syncthreads();
    PH(2)
    const int* lst = s_list[wave];
    total = __builtin_amdgcn_readfirstlane(total);
    if constexpr (BATCH) {
    // r05: the records of up to 64 list entries at a time live in twelve VECTOR registers, lane j holding entry j's (three
    // 16-byte loads per lane, all 64 records in flight at once), and an entry's fields reach the scalar registers by
    // v_readlane when they are needed.  The entry loop then holds no scalar load and no LDS read: nothing in it waits on
    // lgkmcnt (the scalar record fetch of the r04 loop, issued a step ahead, had to be complete at the top of the next
    // step -- scalar loads return out of order, any wait on them is a wait for all), one record set is live instead of
    // three, and the alpha samples run TWO entries ahead of the blend instead of one.
    for (int sub = 0; sub < total; sub += 64) {
      const int nb = imin(64, total - sub);
      const int my = lst[sub + imin(lane, nb - 1)];               // this lane's entry: its drop
      const auto rp = as_global(reinterpret_cast<const u32x4_t*>(comp + my));
      const u32x4_t ra = rp[0], rb = rp[1], rc = rp[2];
      PH_WAITVM
      PH(3)
      auto rl = [](uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); };
      // the two samples of entry j (a pixel outside the footprint, and every pixel of an entry past the batch, reads the
      // frame's zero line: the sample IS 0.0)
      auto issue = [&](int j, double& An0, double& An1, int& inb) {
        const int jj = imin(j, nb - 1);
        const uint32_t xx = rl(ra.x, jj), yy = rl(ra.y, jj);
        const int64_t bs = (int64_t)(((uint64_t)rl(ra.w, jj) << 32) | (uint64_t)rl(ra.z, jj));
        const int pitch = (int)(rl(rb.x, jj) & 0x7fffffffu);
        const int x0 = (int)(xx & 0xffffu), x1 = (int)(xx >> 16), y0 = (int)(yy & 0xffffu), y1 = (int)(yy >> 16);
        const bool inx = (px >= x0) & (px < x1) & (j < nb);
        const bool in0 = inx & live0 & (py0 >= y0) & (py0 < y1), in1 = inx & live1 & (py1 >= y0) & (py1 < y1);
        const int64_t o0 = in0 ? bs + ((int64_t)py0 * pitch + px) : zero_at;
        const int64_t o1 = in1 ? bs + ((int64_t)py1 * pitch + px) : zero_at;
        An0 = arena[o0];
        An1 = arena[o1];
        inb = (in0 ? 1 : 0) | (in1 ? 2 : 0);
      };
      auto blend = [&](int j, double Acur0, double Acur1, int inb) {
        const float te = __uint_as_float(rl(rb.y, j)), k0 = __uint_as_float(rl(rb.z, j)), k1 = __uint_as_float(rl(rb.w, j)),
                    k2 = __uint_as_float(rl(rc.x, j));
        const bool slow = (rl(rb.x, j) >> 31) != 0u;
        bool v0 = (inb & 1) != 0, v1 = (inb & 2) != 0;
        double A0 = Acur0, A1 = Acur1;                            // (0.0 outside the footprint)
        if (has_depth) {                                          // depth-occlusion option: hidden where the drop is behind the scene
          const double z = __hiloint2double((int)rl(rc.w, j), (int)rl(rc.z, j));
          v0 = v0 && !(z > scene0);
          v1 = v1 && !(z > scene1);
          A0 = v0 ? Acur0 : 0.0;
          A1 = v1 ? Acur1 : 0.0;
        }
        if (slow_wave | slow) {                                   // (wave-uniform) literal float64 blend, pixel by pixel
          const const_ptr<CompRec> r64 = as_constant(comp64) + __builtin_amdgcn_readlane(my, j);
          const double K[3] = {r64->K[0], r64->K[1], r64->K[2]};
          const double tau = r64->tau_one, g = r64->g;
          if (v0) {
            double c[3] = {(double)c0.x, (double)c1.x, (double)c2.x};
            blend_pixel(Acur0, tau, cam.exposure_s, g, K, c, m0);
            c0.x = (float)c[0]; c1.x = (float)c[1]; c2.x = (float)c[2];
          }
          if (v1) {
            double c[3] = {(double)c0.y, (double)c1.y, (double)c2.y};
            blend_pixel(Acur1, tau, cam.exposure_s, g, K, c, m1);
            c0.y = (float)c[0]; c1.y = (float)c[1]; c2.y = (float)c[2];
          }
        } else {
          const float2_t Af = {(float)A0, (float)A1};
          const float2_t u = __builtin_elementwise_fma(Af, float2_t{-te, -te}, one2);        // 1 - A te
          c0 = pk_fma_clamp(u, c0, Af * k0);                      // clamp((1 - A te) c + A kg, 0, 1)
          c1 = pk_fma_clamp(u, c1, Af * k1);
          c2 = pk_fma_clamp(u, c2, Af * k2);
          m0 = m0 + A0;                                           // (+ 0.0 outside the footprint: the mask keeps its bits)
          m1 = m1 + A1;
        }
      };
      double A00, A01, A10, A11, A20, A21;
      int b0, b1, b2;
      issue(0, A00, A01, b0);
      issue(1, A10, A11, b1);
      // Whole triples in a loop without an early exit (a `break` between the steps becomes, after the compiler has
      // structurised the loop, an edge from the middle of the body back to its header -- never taken, but the wait-count
      // pass then opens EVERY iteration with vmcnt(0): the r04 loop had that wait), the last one or two entries behind it.
      // Every step issues exactly two loads: the counts are static.
      int e = 0;
      for (; e + 3 <= nb; e += 3) {
        issue(e + 2, A20, A21, b2);
        blend(e, A00, A01, b0);
        issue(e + 3, A00, A01, b0);
        blend(e + 1, A10, A11, b1);
        issue(e + 4, A10, A11, b1);
        blend(e + 2, A20, A21, b2);
      }
      if (e < nb) blend(e, A00, A01, b0);
      if (e + 1 < nb) blend(e + 1, A10, A11, b1);
      // the (zero-line) samples requested past the batch's end are waited for here: a load left in flight would make the
      // compiler open the next batch's loop with vmcnt(0) on every iteration (it cannot place the stray load in the queue)
      asm volatile("" ::"v"(A00), "v"(A01), "v"(A10), "v"(A11));
      PH(4)
    }
    } else {
    struct RecS {
      uint32_t xx, yy;
      int ox, oy, pitch, slow;
      long long off;
      float te, k0, k1, k2;
      double z;
    };
    auto fetch = [&](int idx) {
      const const_ptr<CompRec32> r = as_constant(comp) + __builtin_amdgcn_readfirstlane(idx);
      const uint32_t ps = r->pitch_slow;
      RecS o{r->xx, r->yy, 0, 0, (int)(ps & 0x7fffffffu), (int)(ps >> 31), (long long)r->base, r->te, r->kg[0], r->kg[1], r->kg[2], r->zdist};
      return o;
    };
    // Three stages, as in k_composite: entry e is blended while the samples of entry e + 1 are in flight and the record of
    // entry e + 2 is fetched.  Every lane loads two samples per entry (a pixel outside the footprint reads the tile's first
    // sample -- one line for all of them -- and its value is replaced by 0), so the number of loads in flight is known.
    auto step = [&](int e, const RecS& rcur, double Acur0, double Acur1, int icur, const RecS& rnext, double& An0, double& An1, int& inext,
                    RecS& rfetch, int& i_nn, int& i_cur_next) {
      {
        const int x0 = (int)(rnext.xx & 0xffffu), x1 = (int)(rnext.xx >> 16), y0 = (int)(rnext.yy & 0xffffu), y1 = (int)(rnext.yy >> 16);
        const bool inx = (px >= x0) & (px < x1) & (e + 1 < total);
        const bool in0 = inx & live0 & (py0 >= y0) & (py0 < y1), in1 = inx & live1 & (py1 >= y0) & (py1 < y1);
        // outside the footprint: the frame's zero line (k_scan) -- the sample IS 0.0, nothing to select afterwards
        const int64_t o0 = in0 ? rnext.off + ((int64_t)(py0 + rnext.oy) * rnext.pitch + (px + rnext.ox)) : zero_at;
        const int64_t o1 = in1 ? rnext.off + ((int64_t)(py1 + rnext.oy) * rnext.pitch + (px + rnext.ox)) : zero_at;
        An0 = arena[o0];
        An1 = arena[o1];
        inext = (in0 ? 1 : 0) | (in1 ? 2 : 0);
      }
      rfetch = fetch(i_nn);
      i_cur_next = i_nn;
      i_nn = lst[e + 3 < total ? e + 3 : 0];
      bool v0 = (icur & 1) != 0, v1 = (icur & 2) != 0;
      if (has_depth) {                                            // depth-occlusion option: hidden where the drop is behind the scene
        v0 = v0 && !(rcur.z > scene0);
        v1 = v1 && !(rcur.z > scene1);
      }
      double A0 = Acur0, A1 = Acur1;                              // (0.0 outside the footprint)
      if (has_depth) {
        A0 = v0 ? Acur0 : 0.0;
        A1 = v1 ? Acur1 : 0.0;
      }
      if (slow_wave | rcur.slow) {                                // (wave-uniform) literal float64 blend, pixel by pixel
        const const_ptr<CompRec> r64 = as_constant(comp64) + __builtin_amdgcn_readfirstlane(icur >> 2);
        const double K[3] = {r64->K[0], r64->K[1], r64->K[2]};
        const double tau = r64->tau_one, g = r64->g;
        if (v0) {
          double c[3] = {(double)c0.x, (double)c1.x, (double)c2.x};
          blend_pixel(Acur0, tau, cam.exposure_s, g, K, c, m0);
          c0.x = (float)c[0]; c1.x = (float)c[1]; c2.x = (float)c[2];
        }
        if (v1) {
          double c[3] = {(double)c0.y, (double)c1.y, (double)c2.y};
          blend_pixel(Acur1, tau, cam.exposure_s, g, K, c, m1);
          c0.y = (float)c[0]; c1.y = (float)c[1]; c2.y = (float)c[2];
        }
      } else {
        const float2_t Af = {(float)A0, (float)A1};
        const float2_t u = __builtin_elementwise_fma(Af, float2_t{-rcur.te, -rcur.te}, one2);      // 1 - A te
        c0 = pk_fma_clamp(u, c0, Af * rcur.k0);                   // clamp((1 - A te) c + A kg, 0, 1)
        c1 = pk_fma_clamp(u, c1, Af * rcur.k1);
        c2 = pk_fma_clamp(u, c2, Af * rcur.k2);
        m0 = m0 + A0;                                             // (+ 0.0 outside the footprint: the mask keeps its bits)
        m1 = m1 + A1;
      }
    };
    if (total > 0) {
      // icur packs the two "inside" bits with the drop index (for the rare float64 path): (index << 2) | bits
      int idx0 = lst[0], idx1 = lst[total > 1 ? 1 : 0];
      RecS R0 = fetch(idx0), R1 = fetch(idx1), R2 = R0;
      int i_nn = lst[total > 2 ? 2 : 0];
      int in0 = 0, in1 = 0, in2 = 0, id0 = idx0, id1 = idx1, id2 = 0;
      double A00, A01, A10 = 0.0, A11 = 0.0, A20 = 0.0, A21 = 0.0;
      {
        const int x0 = (int)(R0.xx & 0xffffu), x1 = (int)(R0.xx >> 16), y0 = (int)(R0.yy & 0xffffu), y1 = (int)(R0.yy >> 16);
        const bool inx = (px >= x0) & (px < x1);
        const bool a = inx & live0 & (py0 >= y0) & (py0 < y1), b = inx & live1 & (py1 >= y0) & (py1 < y1);
        A00 = arena[a ? R0.off + ((int64_t)(py0 + R0.oy) * R0.pitch + (px + R0.ox)) : zero_at];
        A01 = arena[b ? R0.off + ((int64_t)(py1 + R0.oy) * R0.pitch + (px + R0.ox)) : zero_at];
        in0 = (a ? 1 : 0) | (b ? 2 : 0);
      }
      for (int e = 0; e < total; e += 3) {
        step(e, R0, A00, A01, in0 | (id0 << 2), R1, A10, A11, in1, R2, i_nn, id2);
        if (e + 1 >= total) break;
        step(e + 1, R1, A10, A11, in1 | (id1 << 2), R2, A20, A21, in2, R0, i_nn, id0);
        if (e + 2 >= total) break;
        step(e + 2, R2, A20, A21, in2 | (id2 << 2), R0, A00, A01, in0, R1, i_nn, id1);
      }
    }
    }
    PH(4)
    __syncthreads();
    PH(5)
  }
  // The composite before the mean shift goes to the library's scratch (this compositor only runs when no caller wants it):
  // as three floats per pixel, or (r05, RR_OPT_COMPOSITE_U16, default) as three 16-bit codes in an 8-byte word -- one store per pixel, two thirds of the bytes here and in
  // k_finalize.  A value in [0, 1] becomes rint(v * 65534) (error <= 2^-17, an LSB of the uint8 image is 2^-8); anything
  // else -- only possible where no drop was blended, every blend ends with a clamp, or where the input was NaN -- becomes
  // the code 65535: k_finalize then takes the pixel's channel from rainy_bg itself, which is what this lane holds.
  auto code16 = [](float v) -> uint32_t { return (v >= 0.f && v <= 1.f) ? (uint32_t)(v * 65534.0f + 0.5f) : 65535u; };
  // (r06) The tile's sums FIRST, the pixel stores last: __syncthreads() carries a full memory fence, and with the stores in
  // front of it every wave sat at the barrier until its six stores per lane had been acknowledged -- the phase clocks gave
  // "stores + tile reduction" 38 % of the kernel's wave time.  After the barrier nothing waits for the stores any more.
  double sum_c = 0.0;
  if (live0) sum_c = ((double)c0.x + (double)c1.x) + (double)c2.x;
  if (live1) sum_c += ((double)c0.y + (double)c1.y) + (double)c2.y;
  // the tile's four numbers: inside a wave by DPP (no LDS, no barrier), across the four waves through 16 doubles of LDS
  // and ONE barrier (r04: a 256-entry LDS tree with nine)
  const double wa = readlane_f64(wave_incl_scan_f64(sum_c), 63), wb = readlane_f64(wave_incl_scan_f64(sum_b), 63);
  const double wlo = readlane_f64(wave_min_scan_f64(dmin(live0 ? m0 : 1.0e300, live1 ? m1 : 1.0e300)), 63);
  const double whi = readlane_f64(wave_max_scan_f64(dmax(live0 ? m0 : -1.0e300, live1 ? m1 : -1.0e300)), 63);
  __shared__ double s_red[4][4];
  if (lane == 0) {
    s_red[wave][0] = wa;
    s_red[wave][1] = wb;
    s_red[wave][2] = wlo;
    s_red[wave][3] = whi;
  }
  __syncthreads();
  double ra[1], rb[1], rlo[1], rhi[1];
  if (t == 0) {
    ra[0] = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
    rb[0] = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
    rlo[0] = dmin(dmin(s_red[0][2], s_red[1][2]), dmin(s_red[2][2], s_red[3][2]));
    rhi[0] = dmax(dmax(s_red[0][3], s_red[1][3]), dmax(s_red[2][3], s_red[3][3]));
  }
  if (t == 0) {
    double* part = sc.partial + ((int64_t)f * tiles_x * tiles_y + tile) * 4;
    part[0] = ra[0];
    part[1] = rb[0];
    part[2] = rlo[0];
    part[3] = rhi[0];
  }
  const bool c16 = fr.comp_f32 == 2;
  if (live0) {
    if (c16) {                                                    // four 16-bit words per pixel (the fourth is 0): one 8-byte store
      as_global(reinterpret_cast<u32x2_t*>(fr.comp_out))[pix0] = u32x2_t{code16(c0.x) | (code16(c1.x) << 16), code16(c2.x)};
    } else {
      const global_ptr<float> o = as_global(reinterpret_cast<float*>(fr.comp_out)) + pix0 * 3;
      o[0] = c0.x; o[1] = c1.x; o[2] = c2.x;
    }
    if (fr.mask_f64) as_global(fr.mask_f64)[pix0] = m0;
    if (fr.mask_i32) as_global(fr.mask_i32)[pix0] = (int32_t)floor(m0 * 255.0);
  }
  if (live1) {
    if (c16) {
      as_global(reinterpret_cast<u32x2_t*>(fr.comp_out))[pix1] = u32x2_t{code16(c0.y) | (code16(c1.y) << 16), code16(c2.y)};
    } else {
      const global_ptr<float> o = as_global(reinterpret_cast<float*>(fr.comp_out)) + pix1 * 3;
      o[0] = c0.y; o[1] = c1.y; o[2] = c2.y;
    }
    if (fr.mask_f64) as_global(fr.mask_f64)[pix1] = m1;
    if (fr.mask_i32) as_global(fr.mask_i32)[pix1] = (int32_t)floor(m1 * 255.0);
  }
  PH(6)
  PH_FLUSH(4)
}

__global__ __launch_bounds__(256) void k_means(Dims dm, int ntiles, Scratch sc) {
  const int f = blockIdx.x, t = threadIdx.x;
  double a = 0, b = 0, lo = 1.0e300, hi = -1.0e300;
  const double* part = sc.partial + (int64_t)f * ntiles * 4;
  for (int i = t; i < ntiles; i += 256) {
    a += part[i * 4];
    b += part[i * 4 + 1];
    lo = dmin(lo, part[i * 4 + 2]);
    hi = dmax(hi, part[i * 4 + 3]);
  }
  __shared__ double ra[256], rb[256], rlo[256], rhi[256];
  ra[t] = a;
  rb[t] = b;
  rlo[t] = lo;
  rhi[t] = hi;
  __syncthreads();
  for (int ofs = 128; ofs > 0; ofs >>= 1) {
    if (t < ofs) {
      ra[t] += ra[t + ofs];
      rb[t] += rb[t + ofs];
      rlo[t] = dmin(rlo[t], rlo[t + ofs]);
      rhi[t] = dmax(rhi[t], rhi[t + ofs]);
    }
    __syncthreads();
  }
  if (t == 0) {
    const double cnt = (double)dm.H * (double)dm.W * 3.0;
    sc.means[f * 4 + 0] = ra[0] / cnt;
    sc.means[f * 4 + 1] = rb[0] / cnt;
    sc.means[f * 4 + 2] = rlo[0];                  // min / max of rainy_mask: the normalisation of its colour-mapped PNG
    sc.means[f * 4 + 3] = rhi[0];
  }
}

// generator.py:461-466 + matplotlib's float->uint8 truncation
// The same for the 16-bit composite codes of k_composite32 (RR_OPT_COMPOSITE_U16), FOUR pixels per thread: 24 bytes in by
// three 8-byte loads, 12 bytes out by one 12-byte store (a thread per pixel moved its 6 + 3 bytes with three 2-byte loads
// and three byte stores and was slower than the float form it replaces, which reads twice the bytes).  The last pixels of a
// frame (H * W is not a multiple of four) and a frame whose uint8 image does not start on a 4-byte boundary take scalar
// accesses.  Code 65535 = the pixel's own rainy_bg value (see k_composite32).
__global__ __launch_bounds__(256) void k_finalize16(const FrameDesc* frames, Dims dm, Scratch sc) {
  const int f = blockIdx.y;
  const int64_t npix = (int64_t)dm.H * dm.W;
  const int64_t pix0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (pix0 >= npix) return;
  const FrameDesc& fr = frames[f];
  const double diff = sc.means[f * 4 + 0] - sc.means[f * 4 + 1];
  const global_ptr<const u32x2_t> s = as_global(reinterpret_cast<const u32x2_t*>(fr.comp_out)) + pix0;      // a coded pixel: c0 | c1 << 16, c2
  const global_ptr<uint8_t> o = as_global(fr.rgb) + pix0 * 3;
  const int cnt = (int)(npix - pix0 < 4 ? npix - pix0 : 4);
  uint32_t q[12];
  {
    u32x2_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = s[k < cnt ? k : 0];
#pragma unroll
    for (int k = 0; k < 4; k++) { q[3 * k] = w[k].x & 0xffffu; q[3 * k + 1] = w[k].x >> 16; q[3 * k + 2] = w[k].y & 0xffffu; }
  }
  uint32_t out[3] = {0u, 0u, 0u};                  // 12 bytes: R G B of four pixels
#pragma unroll
  for (int px = 0; px < 4; px++) {
    double c[3];
    bool special = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      c[k] = (double)q[px * 3 + k] * (1.0 / 65534.0);
      special = special || q[px * 3 + k] == 65535u;
    }
    if (special && px < cnt) {
      double in[3];
      load_px3(fr.rainy_bg, rainy_kind(fr), pix0 + px, in);
#pragma unroll
      for (int k = 0; k < 3; k++)
        if (q[px * 3 + k] == 65535u) c[k] = (double)(float)in[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double v = clip01(c[2 - k] - diff);      // BGR -> RGB
      const uint32_t byte = (uint32_t)(uint8_t)(int)(v * 255.0);
      const int bi = px * 3 + k;
      out[bi >> 2] |= byte << (8 * (bi & 3));
    }
  }
  if (cnt == 4 && (reinterpret_cast<uintptr_t>(fr.rgb) & 3u) == 0) {
    const global_ptr<uint32_t> o4 = reinterpret_cast<global_ptr<uint32_t>>(o);
    o4[0] = out[0]; o4[1] = out[1]; o4[2] = out[2];
  } else {
    for (int k = 0; k < 3 * cnt; k++) o[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
  }
}

__global__ __launch_bounds__(256) void k_finalize(const FrameDesc* frames, Dims dm, Scratch sc) {
  const int f = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (int64_t)dm.H * dm.W) return;
  const FrameDesc& fr = frames[f];
  const double diff = sc.means[f * 4 + 0] - sc.means[f * 4 + 1];
  const global_ptr<uint8_t> o = as_global(fr.rgb) + pix * 3;
  double c[3];
  if (fr.comp_f32 == 2) {                          // 16-bit codes (k_composite32): 65535 = the pixel's own rainy_bg value
    const u32x2_t w = as_global(reinterpret_cast<const u32x2_t*>(fr.comp_out))[pix];
    const uint32_t q0 = w.x & 0xffffu, q1 = w.x >> 16, q2 = w.y & 0xffffu;
    c[0] = (double)q0 * (1.0 / 65534.0); c[1] = (double)q1 * (1.0 / 65534.0); c[2] = (double)q2 * (1.0 / 65534.0);
    if (q0 == 65535u || q1 == 65535u || q2 == 65535u) {
      double in[3];
      load_px3(fr.rainy_bg, rainy_kind(fr), pix, in);
      if (q0 == 65535u) c[0] = (double)(float)in[0];
      if (q1 == 65535u) c[1] = (double)(float)in[1];
      if (q2 == 65535u) c[2] = (double)(float)in[2];
    }
  } else if (fr.comp_f32) {
    const global_ptr<const float> s = as_global(reinterpret_cast<const float*>(fr.comp_out)) + pix * 3;
    c[0] = (double)s[0]; c[1] = (double)s[1]; c[2] = (double)s[2];
  } else {
    const global_ptr<const double> s = as_global((const double*)fr.comp_out) + pix * 3;
    c[0] = s[0]; c[1] = s[1]; c[2] = s[2];
  }
  for (int k = 0; k < 3; k++) {
    double v = clip01(c[2 - k] - diff);      // BGR -> RGB
    o[k] = (uint8_t)(int)(v * 255.0);
  }
}

// ---------------------------------------------------------------------------
// the two PNG files of a frame, ready for deflate  
// ---------------------------------------------------------------------------
// plt.imsave writes RGBA PNGs (generator.py:466-467).  A PNG scanline is one filter byte + the filtered pixels; with
// filter type 1 (Sub) every byte is stored minus the byte one pixel to the left.  These kernels leave the scanlines of
// both files in HBM, so the host only runs zlib over them and frames the chunks.
__global__ __launch_bounds__(256) void k_png_image(const FrameDesc* frames, Dims dm) {
  const int f = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const FrameDesc& fr = frames[f];
  if (!fr.png_image || pix >= (int64_t)dm.H * dm.W) return;
  const int y = (int)(pix / dm.W), x = (int)(pix - (int64_t)y * dm.W);
  const global_ptr<const uint8_t> s = as_global((const uint8_t*)fr.rgb) + pix * 3;
  const global_ptr<uint8_t> o = as_global(fr.png_image) + (int64_t)y * (1 + 4 * dm.W) + 1 + 4 * x;
  if (x == 0) {
    o[-1] = 1;                                        // filter type: Sub
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255;
  } else {
    o[0] = (uint8_t)(s[0] - s[-3]); o[1] = (uint8_t)(s[1] - s[-2]); o[2] = (uint8_t)(s[2] - s[-1]);
    o[3] = 0;                                         // alpha 255 - 255
  }
}

// plt.imsave(path, rainy_mask) (generator.py:467): Normalize(min, max), then the colour map's 256-entry byte table at
// int(norm * 256) clipped to 255 (matplotlib Colormap.__call__, bytes=True).  lut: [256][4] RGBA.
__device__ inline int mask_lut_index(double a, double lo, double hi) {
  if (!(hi > lo)) return 0;
  const double v = ((a - lo) / (hi - lo)) * 256.0;
  const long long i = (long long)v;
  return i < 0 ? 0 : (i > 255 ? 255 : (int)i);
}
__global__ __launch_bounds__(256) void k_png_mask(const FrameDesc* frames, Dims dm, const uint8_t* lut, Scratch sc) {
  const int f = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const FrameDesc& fr = frames[f];
  __shared__ uint32_t s_lut[256];
  s_lut[threadIdx.x] = reinterpret_cast<const uint32_t*>(lut)[threadIdx.x];
  __syncthreads();
  if (!fr.png_mask || pix >= (int64_t)dm.H * dm.W) return;
  const int y = (int)(pix / dm.W), x = (int)(pix - (int64_t)y * dm.W);
  const double lo = sc.means[f * 4 + 2], hi = sc.means[f * 4 + 3];
  const global_ptr<const double> m = as_global((const double*)fr.mask_f64) + pix;
  const uint32_t c = s_lut[mask_lut_index(m[0], lo, hi)];
  const uint32_t l = x > 0 ? s_lut[mask_lut_index(m[-1], lo, hi)] : 0u;
  const global_ptr<uint8_t> o = as_global(fr.png_mask) + (int64_t)y * (1 + 4 * dm.W) + 1 + 4 * x;
  if (x == 0) o[-1] = 1;
#pragma unroll
  for (int k = 0; k < 4; k++) o[k] = (uint8_t)(((c >> (8 * k)) & 0xffu) - ((l >> (8 * k)) & 0xffu));
}

// The scanline filters of an INPUT file reversed (rr_pngrows.h): one wave per file, 64 consecutive rows at a time, lane l on
// row r0 + l one pixel behind lane l - 1.  At step s lane l reconstructs pixel x = s - l: its left neighbour is its own last
// output, its upper neighbour what lane l - 1 produced a step ago (a wave shuffle; for lane 0 the last row of the previous 64,
// kept in LDS by lane 63), its upper-left neighbour the upper one of its previous step.  The filtered bytes of the next step
// are loaded a step ahead.  BPP 3: R G B bytes -> the B G R uint8 image; BPP 2: big-endian samples -> uint16.
template <int BPP>
__global__ __launch_bounds__(64) void k_png_unfilter(const uint8_t* rows_base, int64_t rows_stride, uint8_t* out_base, int64_t out_stride, int H, int W) {
  extern __shared__ uint8_t unf_prev[];                 // BPP * W: the row above the wave's first one
  const int lane = threadIdx.x;
  const uint8_t* rows = rows_base + (int64_t)blockIdx.x * rows_stride;
  uint8_t* out = out_base + (int64_t)blockIdx.x * out_stride;
  const int64_t RB = 1 + (int64_t)BPP * W;
  for (int r0 = 0; r0 < H; r0 += 64) {
    const int r = r0 + lane;
    const bool valid = r < H;
    const uint8_t* row = rows + (int64_t)(valid ? r : 0) * RB;
    const int ft = valid ? (int)row[0] : 0;
    int left[BPP], upl[BPP], cur[BPP], nxt[BPP];
#pragma unroll
    for (int c = 0; c < BPP; c++) left[c] = upl[c] = cur[c] = nxt[c] = 0;
    if (valid && lane == 0) {
#pragma unroll
      for (int c = 0; c < BPP; c++) nxt[c] = row[1 + c];
    }
    for (int s = 0; s < W + 63; s++) {
      const int x = s - lane;
      const bool active = valid && x >= 0 && x < W;
      int f[BPP], up[BPP];
#pragma unroll
      for (int c = 0; c < BPP; c++) {
        f[c] = nxt[c];
        up[c] = __shfl_up(cur[c], 1);                     // lane l - 1 at pixel x, one step ago
      }
      if (lane == 0 && active) {
#pragma unroll
        for (int c = 0; c < BPP; c++) up[c] = r0 > 0 ? (int)unf_prev[x * BPP + c] : 0;
      }
      {                                                   // the filtered bytes of the next step (pixel x + 1)
        const int xn = x + 1;
        if (valid && xn >= 0 && xn < W) {
#pragma unroll
          for (int c = 0; c < BPP; c++) nxt[c] = row[1 + (int64_t)xn * BPP + c];
        }
      }
      if (active) {
#pragma unroll
        for (int c = 0; c < BPP; c++) {
          const int o = rrrows::png_unfilter_byte(ft, f[c], left[c], up[c], upl[c]);
          upl[c] = up[c];
          left[c] = o;
          cur[c] = o;
        }
        if (BPP == 3) {
          uint8_t* o = out + ((int64_t)r * W + x) * 3;
          o[0] = (uint8_t)cur[2];
          o[1] = (uint8_t)cur[1];
          o[2] = (uint8_t)cur[0];
        } else {
          reinterpret_cast<uint16_t*>(out)[(int64_t)r * W + x] = (uint16_t)((cur[0] << 8) | cur[1]);
        }
        if (lane == 63) {
#pragma unroll
          for (int c = 0; c < BPP; c++) unf_prev[x * BPP + c] = (uint8_t)cur[c];
        }
      }
    }
  }
}

// RR_OPT_PNG_DEFLATE: the scanlines of both files become the zlib streams of their IDAT chunks on the device (rr_deflate.h).
// k_pngz_blocks: one workgroup of 512 threads per 32 KB block of a file's scanlines (grid: blocks x files; file = 2 * frame +
// {image, mask}); its working set (the block, its compressed form, the code tables: 76 KB) is dynamic LDS.  The compressed block goes to the
// block's slot in the scratch, its size and Adler-32 sums to its record.
__global__ __launch_bounds__(512) void k_pngz_blocks(const FrameDesc* frames, int64_t n_bytes, int nb, uint8_t* slots, rrz::BlockMeta* meta) {
  using namespace rrz;
  extern __shared__ __attribute__((aligned(16))) uint8_t pngz_lds[];
  BlockState& S = *reinterpret_cast<BlockState*>(pngz_lds);
  const int file = blockIdx.y, k = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const FrameDesc& fr = frames[file >> 1];
  const uint8_t* rows = (file & 1) ? fr.png_mask : fr.png_image;
  if (!rows) return;
  const int64_t at = (int64_t)k * BLOCK;
  const int len = (int)(n_bytes - at < BLOCK ? n_bytes - at : BLOCK), last = k == nb - 1;
  const uint8_t* src = rows + at;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {       // (the library's own staging: 16-byte loads, all of a thread's in flight)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const global_ptr<const u32x4> w = as_global(reinterpret_cast<const u32x4*>(src));
    u32x4* d = reinterpret_cast<u32x4*>(S.in);
    const int nq = len >> 4;
    u32x4 v[BLOCK / 16 / NT];
#pragma unroll
    for (int u = 0; u < BLOCK / 16 / NT; u++) v[u] = tid + u * NT < nq ? w[tid + u * NT] : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < BLOCK / 16 / NT; u++)
      if (tid + u * NT < nq) d[tid + u * NT] = v[u];
    for (int i = (len & ~15) + tid; i < len; i += NT) S.in[i] = src[i];
  } else {
    for (int i = tid; i < len; i += NT) S.in[i] = src[i];
  }
  p0_init(S, tid, len, last);
  __syncthreads();
  // The lane's token of a chunk of the wave's quarter: lane_token from the ballot of the sequence starts.  A lane's byte and its
  // predecessor are two independent LDS reads, and the NEXT chunk's pair is in flight while this chunk's token is worked out.
  const int w0 = wave * WAVE_BYTES;
  const int n_chunks = len > w0 ? (min(len - w0, WAVE_BYTES) + CHUNK - 1) / CHUNK : 0;
  auto fetch = [&](int c, int& b, int& prev) {
    const int idx = w0 + c * CHUNK + lane;
    b = (int)S.in[idx < len ? idx : 0];
    prev = (int)S.in[lane > 0 && idx <= len ? idx - 1 : 0];
  };
  auto token_of = [&](int c, int b, int prev, bool& valid) {
    const int nv = min(CHUNK, len - (w0 + c * CHUNK));
    valid = lane < nv;
    const unsigned long long start = __ballot(valid && (lane == 0 || b != prev));
    return lane_token(lane, nv, start, b);
  };
  {                                                      // (two chunks per iteration: two independent chains for the scheduler)
    uint32_t s1 = 0, s2 = 0;
    int b0 = 0, p0 = 0, b1 = 0, p1 = 0, c0n = 0, p0n = 0, c1n = 0, p1n = 0;
    if (n_chunks > 0) fetch(0, b0, p0);
    if (n_chunks > 1) fetch(1, b1, p1);
    for (int c = 0; c < n_chunks; c += 2) {
      if (c + 2 < n_chunks) fetch(c + 2, c0n, p0n);
      if (c + 3 < n_chunks) fetch(c + 3, c1n, p1n);
      bool v0, v1 = false;
      const Tok t0 = token_of(c, b0, p0, v0);
      Tok t1{0, 0};
      if (c + 1 < n_chunks) t1 = token_of(c + 1, b1, p1, v1);
      p1_lane(S, wave, t0, b0, len - (w0 + c * CHUNK + lane), v0, s1, s2);
      p1_lane(S, wave, t1, b1, len - (w0 + (c + 1) * CHUNK + lane), v1, s1, s2);
      b0 = c0n; p0 = p0n; b1 = c1n; p1 = p1n;
    }
    s1 = wave_incl_scan_u32(s1);                         // the wave's sums: lane 63 of the inclusive scans
    s2 = wave_incl_scan_u32(s2 % 65521u);
    if (lane == 63) {
      S.ad1[wave] = s1;
      S.ad2[wave] = s2;
    }
  }
  __syncthreads();
  p1_sum(S, tid);
  __syncthreads();
  p2_rank(S, tid);
  __syncthreads();
  p3_tree(S, tid);
  __syncthreads();
  p4_depth(S, tid);
  __syncthreads();
  p5_limit(S, tid);
  __syncthreads();
  p6_assign(S, tid);
  __syncthreads();
  p7_codes(S, tid);
  p8_header(S, tid);                                     // (reads the lengths only)
  p9_wave_bits(S, tid);
  __syncthreads();
  p10_decide(S, tid);
  __syncthreads();
  p11_clear(S, tid);
  __syncthreads();
  p12_ends(S, tid);
  if (!S.stored) {
    uint32_t base = wave_base(S, wave);
    int b0 = 0, p0 = 0, b1 = 0, p1 = 0, c0n = 0, p0n = 0, c1n = 0, p1n = 0;
    if (n_chunks > 0) fetch(0, b0, p0);
    if (n_chunks > 1) fetch(1, b1, p1);
    for (int c = 0; c < n_chunks; c += 2) {
      if (c + 2 < n_chunks) fetch(c + 2, c0n, p0n);
      if (c + 3 < n_chunks) fetch(c + 3, c1n, p1n);
      bool v0, v1 = false;
      const Tok t0 = token_of(c, b0, p0, v0);
      Tok t1{0, 0};
      if (c + 1 < n_chunks) t1 = token_of(c + 1, b1, p1, v1);
      uint32_t n0, n1;
      const uint32_t code0 = token_code(S, t0, n0), code1 = token_code(S, t1, n1);
      const uint32_t i0 = wave_incl_scan_u32(n0), i1 = wave_incl_scan_u32(n1);
      const uint32_t base1 = base + (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
      if (n0) or_bits(S.out, base + i0 - n0, code0, n0);
      if (n1) or_bits(S.out, base1 + i1 - n1, code1, n1);
      base = base1 + (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
      b0 = c0n; p0 = p0n; b1 = c1n; p1 = p1n;
    }
  }
  p12b_stored_bytes(S, tid);
  p13_meta(S, tid, meta + (int64_t)file * nb + k);
  __syncthreads();
  uint32_t* dst = reinterpret_cast<uint32_t*>(slots + ((int64_t)file * nb + k) * SLOT_BYTES);
  const int words = (int)((S.bytes + 3) >> 2);
  for (int i = tid; i < words; i += NT) dst[i] = S.out[i];
}
// k_pngz_pack: the blocks of a file behind each other in the file's scanline buffer, in front of them the 16-byte header and
// the zlib header, behind them the Adler-32 -- if all of it fits the buffer; else the scanlines stay (every workgroup of the
// file reaches the same verdict from the same records).
__global__ __launch_bounds__(256) void k_pngz_pack(const FrameDesc* frames, int64_t n_bytes, int nb, const uint8_t* slots, const rrz::BlockMeta* meta) {
  using namespace rrz;
  const int file = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
  const FrameDesc& fr = frames[file >> 1];
  uint8_t* rows = (file & 1) ? fr.png_mask : fr.png_image;
  if (!rows) return;
  const BlockMeta* fm = meta + (int64_t)file * nb;
  __shared__ int64_t s_total, s_off;
  if (tid == 0) {
    s_total = stream_bytes(fm, nb);
    s_off = block_offset(fm, k);
  }
  __syncthreads();
  if (PNGZ_HEADER + s_total > n_bytes) return;
  const uint8_t* src = slots + ((int64_t)file * nb + k) * SLOT_BYTES;
  uint8_t* dst = rows + s_off;
  const int bytes = (int)fm[k].bytes;
  for (int i = tid; i < bytes; i += 256) dst[i] = src[i];
  if (k == 0 && tid == 0) pack_ends(rows, fm, nb);
}


// ---------------------------------------------------------------------------
// drop tables born on the device: particle generator + packer (BASELINE configs[4])
// ---------------------------------------------------------------------------
// k_particles: one workgroup per frame walks the frame's particles 512 at a time.  Thread t makes particle base + t
// (three Philox blocks: any lane can make any particle), applies the loader's derived fields and the frame filter
// (rr_particles.h), and the survivors are written IN PARTICLE ORDER (the reference composites in file order): ballot +
// prefix popcount inside a wave, the wave totals through LDS, a running base per frame.  The record goes out through LDS
// so that the stores are whole lines; tex_index temporarily holds the first texture of the drop's block of ten.
constexpr int DROP_DW = (int)(sizeof(rr_drop) / 4);
__global__ __launch_bounds__(512) void k_particles(const rr_sim_frame* sims, int H, int W, const double* dgrid, const double* cdf_tabs,
                                                    int n_grid, const double* ratio_db, rr_drop* out, int cap, int32_t* n_out) {
  const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __shared__ rr_sim_frame s_sf;
  __shared__ int s_cnt[8];
  __shared__ uint32_t s_stage[8][64 * DROP_DW];            // per wave: 64 records (57 KB)
  if (t < (int)(sizeof(rr_sim_frame) / 4)) reinterpret_cast<uint32_t*>(&s_sf)[t] = reinterpret_cast<const uint32_t*>(sims + f)[t];
  __syncthreads();
  const rr_sim_frame sf = s_sf;
  const double* cdf = cdf_tabs + (int64_t)sf.table * n_grid;
  double rdb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) rdb[k] = ratio_db[k];
  rr_drop* fout = out + (int64_t)f * cap;
  int base_out = 0;
  for (int base = 0; base < sf.n_particles; base += 512) {
    const int i = base + t;
    bool keep = false;
    rr_drop d;
    if (i < sf.n_particles) {
      rrsim::Particle p;
      rrsim::make_particle(sf, dgrid, cdf, n_grid, (uint32_t)i, p);
      double ratio;
      keep = rrsim::derive_drop(p, sf.render_scale, W, H, d, ratio);
      d.tex_index = 10 * rrsim::texture_bucket(ratio, rdb);
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = base_out, tot = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const int c = s_cnt[w];
      if (w < wave) off += c;
      tot += c;
    }
    const int nw = __popcll(bal);                            // records of this wave: consecutive output slots from `off`
    if (keep) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&d);
      uint32_t* dst = s_stage[wave] + __popcll(bal & ((1ull << lane) - 1ull)) * DROP_DW;
#pragma unroll
      for (int k = 0; k < DROP_DW; k++) dst[k] = src[k];
    }
    wave_lds_sync();
    const int room = imax(imin(nw, cap - off), 0);          // what does not fit is not stored (the count still says so)
    uint32_t* o = reinterpret_cast<uint32_t*>(fout + off);
    for (int k = lane; k < room * DROP_DW; k += 64) o[k] = s_stage[wave][k];
    base_out += tot;
    __syncthreads();
  }
  if (t == 0) n_out[f] = base_out;
}

// k_particle_draws: the renderer's per-drop random draws of one frame (np.random.seed(draw_seed); per drop one
// randint(lo, lo + 10), per non-Big drop one normal(0, 0): bad_weather.py:252-264, generator.py:136) from numpy's legacy
// MT19937 stream, bit for bit what rr_host_frame_draws makes on the host.  The stream is sequential by nature (how many
// words a drop consumes depends on the words): ONE WAVE per frame.  The 624-word state lives in wave-private LDS and is
// regenerated by all lanes (the recurrence allows 227 independent updates at a time); the consumer runs wave-uniform:
// lane l holds tempered word l of the current group of 64, v_readlane hands the next one to the scalar side, and the
// texture indices of 64 drops are gathered into a register and stored together.  With a standard deviation of 0 the
// normal deviate itself is never used: only its rejection loop (how many words it eats) and the cached-second-value
// toggle are followed.
__device__ inline uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
__global__ __launch_bounds__(64) void k_particle_draws(const rr_sim_frame* sims, rr_drop* out, int cap, const int32_t* n_out) {
  const int f = blockIdx.x, lane = threadIdx.x;
  __shared__ uint32_t key[624];
  const int n = imin(n_out[f], cap);
  rr_drop* drops = out + (int64_t)f * cap;
  {                                                          // init_genrand (numpy _legacy_seeding): sequential by definition
    uint32_t seed = sims[f].draw_seed;
    if (lane == 0)
      for (int pos = 0; pos < 624; pos++) {
        key[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
      }
  }
  wave_lds_sync();
  auto regenerate = [&]() {                                   // mt19937_gen: three dependent sweeps + the last word
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    auto sweep = [&](int a, int b, int ofs) {
      for (int k0 = a; k0 < b; k0 += 64) {
        const int kk = k0 + lane;
        uint32_t v = 0;
        if (kk < b) {
          const uint32_t y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
          v = key[kk + ofs] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        wave_lds_sync();                                     // key[kk + 1] of the neighbouring lane is read before it is overwritten
        if (kk < b) key[kk] = v;
        wave_lds_sync();
      }
    };
    sweep(0, 227, 397);
    sweep(227, 454, -227);
    sweep(454, 623, -227);
    if (lane == 0) {
      const uint32_t y = (key[623] & UPPER) | (key[0] & LOWER);
      key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    wave_lds_sync();
  };
  int pos = 624;                                              // next state word (624: regenerate first)
  uint32_t cur = 0;                                           // lane l: tempered word pos0 + l of the current group
  int avail = 0, taken = 0;                                   // words in the group / already handed out
  auto next_u32 = [&]() -> uint32_t {                         // wave-uniform
    if (taken == avail) {
      if (pos == 624) {
        regenerate();
        pos = 0;
      }
      avail = imin(64, 624 - pos);
      cur = lane < avail ? mt_temper(key[pos + lane]) : 0u;
      pos += avail;
      taken = 0;
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)cur, taken++);
  };
  auto next_double = [&]() -> double {
    const int32_t a = (int32_t)(next_u32() >> 5), b = (int32_t)(next_u32() >> 6);
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
  };
  bool has_gauss = false;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    int lo = 0, big = 1;
    if (i < n) {
      const global_ptr<const int32_t> r = as_global(reinterpret_cast<const int32_t*>(drops + i));
      big = r[6] == 0;                                        // rr_drop.type
      lo = r[7];                                              // rr_drop.tex_index: first texture of the block of ten
    }
    int mine = 0;
    const int cnt = imin(64, n - base);
    for (int k = 0; k < cnt; k++) {
      const int lo_k = __builtin_amdgcn_readlane(lo, k), big_k = __builtin_amdgcn_readlane(big, k);
      uint32_t v;                                             // randint(lo, lo + 10): masked rejection, rng = 9, mask = 15
      do {
        v = next_u32() & 15u;
      } while (v > 9u);
      if (lane == k) mine = lo_k + (int)v;
      if (!big_k) {                                           // legacy gauss (polar Box-Muller, second deviate cached)
        if (has_gauss) {
          has_gauss = false;
        } else {
          double r2;
          do {
            const double x1 = 2.0 * next_double() - 1.0;
            const double x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
          } while (r2 >= 1.0 || r2 == 0.0);
          has_gauss = true;
        }
      }
    }
    if (i < n) as_global(reinterpret_cast<int32_t*>(drops + i))[7] = mine;
  }
}

// ---------------------------------------------------------------------------
// batched copies between pinned host memory and the device, by kernel
// ---------------------------------------------------------------------------
// A batch crosses PCIe as hundreds of frame-sized pieces (every frame has its own host arrays).  Issued as one
// hipMemcpyAsync each, the pieces of the two directions queue behind each other -- measured on MI355X: 192 pieces of
// 1.4 MB move at 40 GB/s in one direction and at the same 40 GB/s IN TOTAL when both directions run (scripts/probes/
// pcie_probe.hip).  One kernel per direction that walks a list of pieces and reads / writes the pinned host memory
// directly moves 56 GB/s one way and 90 GB/s both ways, next to the compute kernels.
struct CopyPiece {
  const void* src;
  void* dst;
  uint64_t bytes;
};
struct HostCopy {                    // one copy between host and device memory (host-side bookkeeping)
  void* dst;
  const void* src;
  size_t bytes;
};
__global__ __launch_bounds__(256) void k_copy_pieces(const CopyPiece* pieces, int n) {
  const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
  for (int j = 0; j < n; j++) {
    const CopyPiece p = pieces[j];
    const uintptr_t a = reinterpret_cast<uintptr_t>(p.src) | reinterpret_cast<uintptr_t>(p.dst);
    if ((a & 15) == 0) {
      const size_t n16 = p.bytes >> 4;
      const uint4* s16 = static_cast<const uint4*>(p.src);
      uint4* d16 = static_cast<uint4*>(p.dst);
      for (size_t i = gtid; i < n16; i += gsz) d16[i] = s16[i];
      const size_t done = n16 << 4;
      if (gtid < p.bytes - done) static_cast<uint8_t*>(p.dst)[done + gtid] = static_cast<const uint8_t*>(p.src)[done + gtid];
    } else {                                                   // small odd pieces (counts, short status arrays)
      for (size_t i = gtid; i < p.bytes; i += gsz) static_cast<uint8_t*>(p.dst)[i] = static_cast<const uint8_t*>(p.src)[i];
    }
  }
}

// A batch's descriptor records (FrameDesc / PreFrame / rr_sim_frame) go from their pinned host buffer to the device by
// KERNEL, not by hipMemcpyAsync: a DMA request in the compute stream waits in the DMA queue until the kernels in front of
// it have run, and every copy queued after it -- the previous batch's download, the next batch's upload, on other
// streams -- waits with it (measured: the download of batch k started when batch k+1 reached its descriptor copy).
__global__ __launch_bounds__(256) void k_copy_small(const uint32_t* src_pinned_host, uint32_t* dst, int n_words) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_words; i += gridDim.x * 256) dst[i] = src_pinned_host[i];
}

// drop counts that only exist on the device (rr_frame_in.n_drops_dev): patched into the frame descriptors before the
// first kernel of the chain reads them; n_drops of the descriptor is the capacity
__global__ void k_set_i32(int32_t* p, int32_t v) { *p = v; }

__global__ void k_patch_counts(FrameDesc* frames, int n) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n || !frames[f].n_drops_dev) return;
  const int c = *frames[f].n_drops_dev;
  frames[f].n_drops = imax(imin(c, frames[f].n_drops), 0);
}

}  // namespace

// ===========================================================================
// host side
// ===========================================================================
struct ProfEntry {
  const char* name;
  hipEvent_t a, b;
};

struct rr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // streak DB
  uint8_t* d_tex = nullptr;
  bool own_tex = false;
  int32_t* d_tex_h = nullptr;
  int32_t* d_tex_w = nullptr;
  int64_t* d_tex_off = nullptr;
  uint8_t* d_tex_pad = nullptr;      // padded copies (k_pad_textures) + their offsets
  int64_t* d_tex_poff = nullptr;
  bool padded_tex = true;            // RR_OPT_PADDED_TEXTURES
  uint8_t* d_tex_pair = nullptr;     // pair textures (k_pair_textures) + their offsets: what k_tile_rows stages
  int64_t* d_tex_qoff = nullptr;
  int tile_rows = 2;                 // RR_OPT_TILE_ROWS: rotate + INTER_AREA tiles by row walks, a wave per tile (k_tile_rows); 2: the Big (bicubic) tiles as well
  int n_cu = 256;                    // compute units of the device (persistent kernels size their grid by it)
  int erec_nfov = 0;                 // n_fov the edge-record scratch was sized for
  int rows_shares = 2;               // RR_OPT_ROWS_SHARES: shares of the tile list per workgroup of k_tile_rows
  bool png_deflate = false;          // RR_OPT_PNG_DEFLATE: the PNG outputs hold zlib streams (rr_deflate.h)
  uint8_t* d_pngz_slots = nullptr;
  rrz::BlockMeta* d_pngz_meta = nullptr;
  size_t pngz_cap = 0;               // blocks
  bool pngz_attr = false;
  bool wild_pixels = false;          // RR_OPT_WILD_PIXELS: rainy_bg may hold values outside [0, 1] (k_pad_visits)
  int32_t *d_pad_first = nullptr, *d_eff_first = nullptr;
  size_t pad_cap = 0;                // elements of each
  bool pipe_f32 = true;              // RR_OPT_PIPELINE_F32: float32 hand-over from the pre-pass to the hot path inside rr_pipeline_*
  bool dda_attr = false;
  bool bin_rows = true;              // RR_OPT_BIN_ROWS
  int fill_rule = 1;                 // RR_OPT_FOV_FILL_RULE: 1 (default since r06) OpenCV's fillConvexPoly; 0 the span rule of rounds 1-5
  bool composite_u16 = true;         // RR_OPT_COMPOSITE_U16
  bool blur_dma = true;              // RR_OPT_BLUR_DMA
  int fov_dda = 1;                   // RR_OPT_FOV_DDA: a thread per drop for the polygons of the float colour branch (1: k_fov_dda, 2: k_fov_walk)
  int comp_waves = 0;                // RR_OPT_COMPOSITE_WAVES: waves per SIMD the float compositor's registers are held to (0: the kernel's own choice)
  int colour_stream = 1;             // RR_OPT_COLOUR_STREAM: 0 one stream; 1 the FOV chain on a second stream
  hipStream_t s_col = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool comp_batch = true;            // RR_OPT_COMPOSITE_BATCH: list entries' records 64 at a time in vector registers, samples two entries ahead
  int n_tex = 0;
  std::vector<int32_t> h_tex_h, h_tex_w;   // host copies of the database's metadata (rr_bcast_streak_db hands them to the other contexts)
  std::vector<int64_t> h_tex_off;
  int64_t tex_bytes = 0;             // size of the texel buffer
  float* d_ctab = nullptr;
  // particle generator (rr_set_particle_tables / rr_generate_drops_device)
  double *d_dgrid = nullptr, *d_cdf = nullptr, *d_ratio_db = nullptr;
  int n_grid = 0, n_tables = 0, n_ratio = 0;
  rr_sim_frame* d_sims = nullptr;
  int cap_sims = 0;
  rr_drop* d_gen_drops = nullptr;    // staging of rr_generate_drops (host-pointer variant)
  int32_t* d_gen_counts = nullptr;
  size_t cap_gen_drops = 0, cap_gen_counts = 0;
  uint8_t* d_lut = nullptr;         // [256][4] RGBA colour map of the rain-mask PNG (rr_set_colormap)
  bool have_lut = false;
  rr_camera cam;
  bool have_cam = false, have_db = false;
  // scratch
  Scratch sc{};
  FrameDesc* d_frames = nullptr;
  int cap_frames = 0, cap_drops = 0;
  Dims cap_dims{0, 0, 0, 0};
  int64_t arena_cap = 0;            // doubles per frame
  double* d_comp_out = nullptr;     // [frame][H*W*3] when the caller passes rainy_bg_out == NULL
  // Descriptor staging: a batch's FrameDesc / PreFrame records go to the device with an asynchronous copy from PINNED
  // memory (a pageable source makes hipMemcpyAsync wait for the stream -- i.e. for the previous batch's kernels -- inside
  // the submit call, which serialises the pipeline slots).  A ring of buffers, each guarded by an event recorded behind
  // its copy, lets several batches be queued before the first copy has executed.
  struct DescRing {
    static constexpr int N = 2 * RR_PIPE_SLOTS + 2;
    void* host[N] = {};
    size_t cap[N] = {};
    hipEvent_t ev[N] = {};
    bool used[N] = {};
    int next = 0;
  } ring_frames, ring_pre, ring_sims;
  // last launch (for rr_synchronize bookkeeping)
  int last_n = 0;
  // host-pointer entry points: device staging per pipeline slot (slot 0 serves the synchronous calls)
  struct Staging {
    double *bg = nullptr, *rainy = nullptr, *env = nullptr, *omega = nullptr, *comp = nullptr, *mask = nullptr;
    rr_drop* drops = nullptr;
    uint8_t* rgb = nullptr;
    int32_t *mask_i = nullptr, *status = nullptr;
    double* colour = nullptr;        // rr_frame_out.drop_colour
    int32_t* ndrops = nullptr;       // drop counts of generated drop tables (rr_frame_in.sim)
    double* depth = nullptr;         // pre-pass input (float32 or float64 per frame slot of 8 bytes/pixel)
    uint8_t* bg8 = nullptr;          // pre-pass input given as bytes (rr_prepass_in.bg_u8)
    uint8_t* env_u8 = nullptr;
    uint8_t *png_i = nullptr, *png_m = nullptr;   // PNG scanlines of the image / of the colour-mapped mask
    int frames = 0, drops_cap = 0;
    Dims dims{0, 0, 0, 0};
  };
  struct Slot {
    Staging st;
    hipEvent_t ev_up = nullptr, ev_comp = nullptr, ev_down = nullptr;
    std::vector<HostCopy> down;                       // the batch's device-to-host copies, issued by rr_pipeline_wait
    CopyPiece *d_up = nullptr, *d_down = nullptr;     // piece lists of the copy kernels
    CopyPiece *h_up = nullptr, *h_down = nullptr;     // (pinned)
    size_t cap_up = 0, cap_down = 0;
    int64_t* h_flags = nullptr;      // pinned: [0] arena-overflow flag as seen after this batch
    bool busy = false, rendered = false;
    std::vector<void*> ext_blobs;     // device copies of caller-made tiles (rr_ext_tile), freed when the slot is reused
  } slots[RR_PIPE_SLOTS];
  hipStream_t s_up = nullptr, s_down = nullptr;
  // pre-pass (fog + environment map)
  rrpre::Kernels pk{};
  bool have_pk = false, have_eg = false;
  std::vector<int32_t> eg_src_host;  // the geometry's cell -> source pixel table (host copy: rrpre::build_env_need)
  uint8_t* d_need_h = nullptr;
  int need_half = -1;
  rrpre::EnvGeom eg{};
  int32_t *d_esrc = nullptr, *d_etop = nullptr, *d_ebot = nullptr;
  rrpre::PreScratch psc{};
  rrpre::PreFrame* d_pre = nullptr;
  int pre_frames = 0, pre_H = 0, pre_W = 0, pre_We = 0;
  bool pre_planes = false;           // the float64 planes of the three-kernel fog layer are allocated (tap counts other than 25)
  // profiling
  bool prof = false;
  // options (rr_set_option): none of them changes a result bit
  bool dedup = true;                 // RR_OPT_DEDUP: share bit-identical raw tiles inside a batch (k_dedup)
  int fov_threads = 0, fov_dpt = 0;  // RR_OPT_FOV_THREADS / RR_OPT_FOV_DROPS_PER_THREAD: 0 = chosen by the library
  int blur_wg = 0;                   // RR_OPT_BLUR_WORKGROUPS: workgroups per CU of the fused blur (3, 4 or 5); 0: the library's choice (4)
  bool general_fov = false;          // RR_OPT_GENERAL_FOV: force the general colour path (prefix table in HBM)
  int fov_f32 = 2;                   // RR_OPT_FOV_F32: float32 vertices / prefix rows / sums in the colour branch (image within 1 LSB): 0 never,
                                     // 1 always, 2 (default) whenever the compositor blends float colours (no float64 composite asked for)
  bool depth_occlusion = false;      // RR_OPT_DEPTH_OCCLUSION: hide drops behind the scene depth (changes the output; default off)
  bool copy_kernels = false;         // RR_OPT_COPY_KERNELS: batched copy kernels for pinned host buffers (default: hipMemcpyAsync;
                                     // measured slower than the DMA engines once the pieces are merged)
  double* d_omega = nullptr;         // resident solid-angle map (rr_set_solid_angles): frames may pass omega == NULL
  float* d_omega32 = nullptr;        // the same as floats (frames whose map is float: RR_IN_ENV_F32)
  int omega_He = 0, omega_We = 0;
  std::vector<std::pair<const char*, size_t>> host_allocs;   // rr_host_alloc blocks: pieces inside one block may be merged across padding
  std::string host_err;              // rr_host_last_error (under host_mu)
  std::mutex host_mu;                // guards host_allocs: rr_host_alloc / rr_host_free may be called from another thread than the one that
                                     // submits batches (the driver page-locks the slots of later batches while the first one is decoded)
  bool composite_f64 = false;        // RR_OPT_COMPOSITE_F64: float64 colours in the compositor even when nobody asks for the composite
  int scratch_hp = 0;                // span pitch the scratch was sized for
  bool scratch_general = false;      // prefix table / polygons of the general colour path allocated
  std::vector<ProfEntry> prof_pending;
  std::vector<rr_kernel_stat> prof_stats;
  std::vector<hipEvent_t> ev_pool;
};

namespace {

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
      return RR_E_HIP;                                                                            \
    }                                                                                             \
  } while (0)
#define HIPCHK_CTX(c_, call)                                                                      \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (c_)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return RR_E_HIP;                                                                            \
    }                                                                                             \
  } while (0)

template <typename T>
int dev_alloc(rr_ctx* ctx, T*& p, size_t count) {
  if (p) {
    hipFree(p);
    p = nullptr;
  }
  if (count == 0) count = 1;
  HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
  return RR_OK;
}

hipEvent_t get_event(rr_ctx* ctx) {
  if (!ctx->ev_pool.empty()) {
    hipEvent_t e = ctx->ev_pool.back();
    ctx->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

// next buffer of a descriptor ring (>= bytes, pinned); waits for the copy that last read it
int ring_acquire(rr_ctx* ctx, rr_ctx::DescRing& r, size_t bytes, int& idx, void*& host) {
  idx = r.next;
  r.next = (r.next + 1) % rr_ctx::DescRing::N;
  if (!r.ev[idx]) HIPCHK(hipEventCreateWithFlags(&r.ev[idx], hipEventDisableTiming));
  if (r.used[idx]) HIPCHK(hipEventSynchronize(r.ev[idx]));
  r.used[idx] = false;
  if (bytes > r.cap[idx]) {
    if (r.host[idx]) HIPCHK(hipHostFree(r.host[idx]));
    r.host[idx] = nullptr;
    r.cap[idx] = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    HIPCHK(hipHostMalloc(&r.host[idx], want, hipHostMallocDefault));
    r.cap[idx] = want;
  }
  host = r.host[idx];
  return RR_OK;
}
int ring_commit(rr_ctx* ctx, rr_ctx::DescRing& r, int idx, hipStream_t s) {
  HIPCHK(hipEventRecord(r.ev[idx], s));
  r.used[idx] = true;
  return RR_OK;
}

struct ProfScope {
  rr_ctx* ctx;
  hipStream_t s;
  ProfEntry pe;
  bool on;
  ProfScope(rr_ctx* c, hipStream_t st, const char* name) : ctx(c), s(st), on(c->prof) {
    if (on) {
      pe.name = name;
      pe.a = get_event(c);
      pe.b = get_event(c);
      hipEventRecord(pe.a, s);
    }
  }
  ~ProfScope() {
    if (on) {
      hipEventRecord(pe.b, s);
      ctx->prof_pending.push_back(pe);
    }
  }
};

void prof_collect(rr_ctx* ctx) {
  for (auto& pe : ctx->prof_pending) {
    float ms = 0.f;
    hipEventSynchronize(pe.b);
    hipEventElapsedTime(&ms, pe.a, pe.b);
    bool found = false;
    for (auto& s : ctx->prof_stats)
      if (!strcmp(s.name, pe.name)) {
        s.launches++;
        s.total_ms += ms;
        found = true;
        break;
      }
    if (!found) {
      rr_kernel_stat s;
      memset(&s, 0, sizeof(s));
      strncpy(s.name, pe.name, sizeof(s.name) - 1);
      s.launches = 1;
      s.total_ms = ms;
      ctx->prof_stats.push_back(s);
    }
    ctx->ev_pool.push_back(pe.a);
    ctx->ev_pool.push_back(pe.b);
  }
  ctx->prof_pending.clear();
}

// the fast colour path needs the span state of a drop in registers and a map row in LDS, and its exact span
// arithmetic needs |2*dx*dy| < 2^23 (k_fov_spans)
bool fov_fast_path(const rr_ctx* ctx, const Dims& dm) {
  return !ctx->general_fov && dm.He <= HE_MAX && dm.We <= FOV_WE_MAX && (int64_t)dm.We * dm.He < (1 << 22) && ctx->cam.n_fov * 1 <= 64 &&
         ctx->cam.n_fov >= 3;
}

int ensure_scratch(rr_ctx* ctx, int n, int max_drops, const Dims& dm, bool need_comp_out) {
  const bool grow_frames = n > ctx->cap_frames;
  const bool grow_drops = max_drops > ctx->cap_drops;
  const bool dims_change = dm.H != ctx->cap_dims.H || dm.W != ctx->cap_dims.W || dm.He != ctx->cap_dims.He || dm.We != ctx->cap_dims.We;
  const bool general = !fov_fast_path(ctx, dm);
  if (grow_frames || grow_drops || dims_change || (need_comp_out && !ctx->d_comp_out) || general != ctx->scratch_general || ctx->cam.n_fov > ctx->erec_nfov) {
    HIPCHK(hipDeviceSynchronize());
    const int F = grow_frames ? n : ctx->cap_frames;
    const int D = grow_drops ? max_drops : ctx->cap_drops;
    const size_t fd = (size_t)F * (size_t)(D > 0 ? D : 1);
    const int Hp = (dm.He + 3) & ~3;
    int rc;
    if ((rc = dev_alloc(ctx, ctx->sc.plan, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.comp, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.comp32, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.npts, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.sizes, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.list_rot, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.rows_list, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.rows_sorted, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.rows_n, (size_t)F * 2))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.rows_hist, (size_t)RW_TEX_MAX * 3 + 16))) return rc;      // + RW_TEX_MAX 8-byte cost sums + the share counter
    ctx->sc.rows_cost = reinterpret_cast<unsigned long long*>(ctx->sc.rows_hist + RW_TEX_MAX);
    ctx->sc.rows_next = ctx->sc.rows_hist + 3 * RW_TEX_MAX;
    ctx->sc.rot_total = ctx->sc.rows_next + 1;
    if ((rc = dev_alloc(ctx, ctx->sc.rows_bounds, (size_t)RW_SHARE_MAX + 1))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.rows_fbase, (size_t)F * RW_TEX_MAX))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.fov_list, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.fov_erec, general ? 1 : fd * (size_t)(ctx->cam.n_fov > 0 ? ctx->cam.n_fov : RR_MAX_FOV)))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.fov_pix, general ? 1 : fd * (size_t)(ctx->cam.n_fov > 0 ? ctx->cam.n_fov : RR_MAX_FOV)))) return rc;
    ctx->erec_nfov = ctx->cam.n_fov;
    if ((rc = dev_alloc(ctx, ctx->sc.fov_list_n, (size_t)F))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.list_gen, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.list_slow, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.blur_items, fd * 8))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.list_small, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.colpart, fd * COL_PARTS * 5))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.wtab, fd * 2 * (BR_MAX + 1)))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.wtab_big, (size_t)F * SLOW_CAP * 2 * (MAX_R + 1)))) return rc;
    if (general) {                     // polygons and the prefix table only exist on the general colour path
      if ((rc = dev_alloc(ctx, ctx->sc.poly, fd * 2 * POLY_STRIDE))) return rc;
      if ((rc = dev_alloc(ctx, ctx->sc.prefix, (size_t)F * dm.He * (size_t)(dm.We + 1) * 4))) return rc;
      if ((rc = dev_alloc(ctx, ctx->sc.spans, 1))) return rc;
    } else {
      if ((rc = dev_alloc(ctx, ctx->sc.poly, 1))) return rc;
      if ((rc = dev_alloc(ctx, ctx->sc.prefix, 1))) return rc;
      const size_t per_frame = (size_t)(((D > 0 ? D : 1) + 1 + 7) & ~7) * (size_t)Hp;
      if ((rc = dev_alloc(ctx, ctx->sc.spans, (size_t)F * per_frame))) return rc;
      HIPCHK(hipMemset(ctx->sc.spans, 0, sizeof(uint32_t) * (size_t)F * per_frame));      // incl. every frame's zero row
    }
    ctx->scratch_general = general;
    ctx->scratch_hp = Hp;
    if ((rc = dev_alloc(ctx, ctx->sc.bbox, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.blended, fd))) return rc;
    {
      const size_t nct = (size_t)((dm.W + CTILE - 1) / CTILE) * ((dm.H + CTILE - 1) / CTILE);
      if ((rc = dev_alloc(ctx, ctx->sc.clist, fd * nct))) return rc;
      if ((rc = dev_alloc(ctx, ctx->sc.ccount, (size_t)F * nct))) return rc;
    }
    if ((rc = dev_alloc(ctx, ctx->sc.counts, (size_t)F * 8))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.canon, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.bigs_list, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.bigs_n, (size_t)F))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.tkey, fd * 2))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.lrec, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.list_big, fd))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.big_off, fd + F))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.htab, fd * 2))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.fband, (size_t)F * COL_PARTS * 2))) return rc;
    const int ntiles = ((dm.W + TILE - 1) / TILE) * ((dm.H + TILE - 1) / TILE);
    if ((rc = dev_alloc(ctx, ctx->sc.partial, (size_t)F * ntiles * 4))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.means, (size_t)F * 4))) return rc;
    if ((rc = dev_alloc(ctx, ctx->sc.arena_need, (size_t)F))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_frames, (size_t)F))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_comp_out, (size_t)F * dm.H * dm.W * 3))) return rc;
    // arena: keep per-frame capacity, reallocate for the new frame count
    if (ctx->arena_cap == 0) ctx->arena_cap = (int64_t)(D > 0 ? D : 1) * 1024 + (1 << 20);      // a multiple of 16 doubles
    if ((rc = dev_alloc(ctx, ctx->sc.arena, (size_t)F * (size_t)ctx->arena_cap))) return rc;
    ctx->cap_frames = F;
    ctx->cap_drops = D;
    ctx->cap_dims = dm;
  }
  return RR_OK;
}

// After an overflow: a larger arena, sized by the largest need any batch reported since the last (re)size (need_max is
// sticky on the device, so the batch that overflowed is covered even when later batches were queued behind it); the
// capacity never shrinks.
// Every batch has its OWN overflow flag (per pipeline slot; flag 0 for the device-pointer calls): with several batches in
// flight a shared flag cannot say WHICH of them ran against the short arena.  A batch that finds its flag set calls this:
// the arena only grows if the sticky maximum says it is still too small (an earlier wait may have grown it already for
// both), and the caller clears its own flag.
int grow_arena(rr_ctx* ctx) {
  HIPCHK(hipDeviceSynchronize());
  unsigned long long need = 0;
  HIPCHK(hipMemcpy(&need, ctx->sc.need_max, sizeof(need), hipMemcpyDeviceToHost));
  if ((int64_t)need > ctx->arena_cap) {
    const int64_t cap = ((int64_t)need + (int64_t)need / 4 + (1 << 16) + 15) & ~15LL;
    ctx->arena_cap = cap;
    int rc = dev_alloc(ctx, ctx->sc.arena, (size_t)ctx->cap_frames * (size_t)cap);
    if (rc) return rc;
  }
  HIPCHK(hipMemset(ctx->sc.need_max, 0, sizeof(unsigned long long)));
  return RR_OK;
}

int enqueue(rr_ctx* ctx, int n, const rr_frame_in* in, const rr_frame_out* out, hipStream_t s, int ovf_idx = 0) {
  if (!ctx->have_cam || !ctx->have_db) {
    ctx->err = "streak DB and camera must be set before rendering";
    return RR_E_STATE;
  }
  if (n <= 0 || !in || !out) {
    ctx->err = "bad frame batch";
    return RR_E_ARG;
  }
  Dims dm{in[0].H, in[0].W, in[0].He, in[0].We};
  int max_drops = 0;
  bool need_comp = false, want_png = false, any_f64_comp = false;
  for (int f = 0; f < n; f++) {
    if (in[f].H != dm.H || in[f].W != dm.W || in[f].He != dm.He || in[f].We != dm.We) {
      ctx->err = "all frames of a batch must share H,W,He,We";
      return RR_E_ARG;
    }
    if (in[f].strategy != 0 && in[f].strategy != 1) {
      ctx->err = "rendering strategy must be 0 (default) or 1 ('white'); 'naive_db' is broken in the reference (bad_weather.py:355)";
      return RR_E_ARG;
    }
    {
      const int t = in[f].in_types;
      if ((t & ~(RR_IN_BG_F32 | RR_IN_BG_U8 | RR_IN_ENV_F32 | RR_IN_RAINY_F32 | RR_IN_RAINY_U8)) || ((t & RR_IN_BG_F32) && (t & RR_IN_BG_U8)) ||
          ((t & RR_IN_RAINY_F32) && (t & RR_IN_RAINY_U8))) {
        ctx->err = "rr_frame_in.in_types: unknown bits, or two element types for one array";
        return RR_E_ARG;
      }
    }
    if (!in[f].omega && !(ctx->d_omega && ctx->omega_He == dm.He && ctx->omega_We == dm.We)) {
      ctx->err = "omega == NULL needs rr_set_solid_angles for this map size";
      return RR_E_STATE;
    }
    if (in[f].n_drops < 0 || in[f].n_drops > 65536 || !in[f].bg || !in[f].rainy_bg || !in[f].env_xyY ||
        (in[f].n_drops > 0 && !in[f].drops) || !out[f].rainy_rgb) {
      ctx->err = "null frame pointer or n_drops outside [0, 2^16] (generator.py:425)";
      return RR_E_ARG;
    }
    if (out[f].mask_png && (!out[f].mask_f64 || !ctx->have_lut)) {
      ctx->err = "mask_png needs mask_f64 and rr_set_colormap";
      return RR_E_ARG;
    }
    if (in[f].n_drops > max_drops) max_drops = in[f].n_drops;
    if (!out[f].rainy_bg_out) need_comp = true;
    else any_f64_comp = true;
    want_png = want_png || out[f].rainy_png || out[f].mask_png;
  }
  const bool use32 = !any_f64_comp && !ctx->composite_f64;
  const bool fov32 = ctx->fov_f32 == 1 || (ctx->fov_f32 == 2 && use32);      // float colour branch (RR_OPT_FOV_F32)
  if (dm.H <= 0 || dm.W <= 0 || dm.He <= 0 || dm.We <= 0) {
    ctx->err = "bad frame size";
    return RR_E_ARG;
  }
  int rc = ensure_scratch(ctx, n, max_drops, dm, need_comp);
  if (rc) return rc;
  const int D = ctx->cap_drops > 0 ? ctx->cap_drops : 1;
  int ring_idx;
  void* ring_host;
  bool any_dev_count = false;
  if ((rc = ring_acquire(ctx, ctx->ring_frames, sizeof(FrameDesc) * (size_t)n, ring_idx, ring_host))) return rc;
  FrameDesc* h_frames = static_cast<FrameDesc*>(ring_host);
  for (int f = 0; f < n; f++) {
    FrameDesc& fd = h_frames[f];
    fd.bg = in[f].bg;
    fd.rainy_bg = in[f].rainy_bg;
    fd.env = in[f].env_xyY;
    fd.omega = in[f].omega ? (const void*)in[f].omega : ((in[f].in_types & RR_IN_ENV_F32) ? (const void*)ctx->d_omega32 : (const void*)ctx->d_omega);
    fd.drops = in[f].drops;
    fd.rgb = out[f].rainy_rgb;
    fd.comp_out = out[f].rainy_bg_out ? out[f].rainy_bg_out : ctx->d_comp_out + (size_t)f * dm.H * dm.W * 3;
    fd.mask_f64 = out[f].mask_f64;
    fd.mask_i32 = out[f].mask_i32;
    fd.status = out[f].drop_status;
    fd.png_image = out[f].rainy_png;
    fd.png_mask = out[f].mask_png;
    if (ctx->depth_occlusion && in[f].depth && in[f].depth_f64 != 0 && in[f].depth_f64 != 1) {
      ctx->err = "RR_OPT_DEPTH_OCCLUSION needs a float32 or float64 depth map";
      return RR_E_ARG;
    }
    fd.depth = ctx->depth_occlusion ? in[f].depth : nullptr;
    fd.depth_f64 = in[f].depth_f64 == 1 ? 1 : 0;
    fd.ext = in[f].ext;
    fd.colour_out = out[f].drop_colour;
    fd.n_drops_dev = in[f].n_drops_dev;
    fd.comp_f32 = use32 ? ((ctx->composite_u16 && !ctx->wild_pixels) ? 2 : 1) : 0;
    fd.in_types = in[f].in_types;
    any_dev_count = any_dev_count || in[f].n_drops_dev;
    fd.n_drops = in[f].n_drops;
    fd.strategy = in[f].strategy;
    fd.opacity = in[f].opacity_attenuation;
  }
  static_assert(sizeof(FrameDesc) % 4 == 0 && sizeof(rrpre::PreFrame) % 4 == 0 && sizeof(rr_sim_frame) % 4 == 0, "descriptor sizes");
  hipLaunchKernelGGL(k_copy_small, dim3(16), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(h_frames), reinterpret_cast<uint32_t*>(ctx->d_frames),
                     (int)(sizeof(FrameDesc) * (size_t)n / 4));
  if ((rc = ring_commit(ctx, ctx->ring_frames, ring_idx, s))) return rc;
  if (any_dev_count) hipLaunchKernelGGL(k_patch_counts, dim3((n + 63) / 64), dim3(64), 0, s, ctx->d_frames, n);
  const int tiles_x = (dm.W + TILE - 1) / TILE, tiles_y = (dm.H + TILE - 1) / TILE;
  const int ntiles = tiles_x * tiles_y;
  Scratch sc = ctx->sc;
  sc.overflow = ctx->sc.overflow + ovf_idx;
  sc.tex_pad = ctx->padded_tex ? ctx->d_tex_pad : nullptr;
  sc.tex_poff = ctx->d_tex_poff;
  sc.tex_pair = ctx->d_tex_pair;
  sc.tex_qoff = ctx->d_tex_qoff;
  sc.n_tex = ctx->n_tex;
  sc.rows_on = (ctx->tile_rows && ctx->d_tex_pair && ctx->n_tex <= RW_TEX_MAX) ? 1 : 0;
  sc.big_on = (sc.rows_on && ctx->tile_rows >= 2 && ctx->padded_tex && ctx->d_tex_pad && 2 * ctx->n_tex <= RW_TEX_MAX) ? 1 : 0;
  sc.n_buckets = sc.big_on ? 2 * ctx->n_tex : ctx->n_tex;
  // k_tile_rows: one workgroup of 16 waves per CU (the LDS holds one texture + 16 wave-private tables); fewer for small batches
  const int rows_wgs = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(ctx->n_cu, RW_SHARE_MAX / ctx->rows_shares), ((int64_t)n * max_drops + 31) / 32));
  const int blur_wg = ctx->blur_wg ? ctx->blur_wg : 4;      // workgroups per CU the fused blur is sized for
  sc.blur_bx = blur_wg == 3 ? 3072 : (blur_wg == 5 ? 2304 : 2816);
  sc.blur_by = blur_wg == 5 ? 1600 : 2048;
  // One in-order stream: FOV spans -> plan -> scan -> dedup -> lists -> FOV sums -> colour -> tiles -> blur ->
  // composite -> finalise.
  // Work-list kernels take their items grid-stride, the list lengths only exist on the device: with many frames per call
  // a per-frame grid sized for the worst case is mostly workgroups that find nothing to do (hundreds of thousands of
  // them per launch) -- the per-frame grid shrinks as the batch grows, keeping >= 16 K workgroups in flight overall.
  auto grid_cap = [&](int single_frame) { return imax(64, imin(single_frame, 16384 / n)); };
  if (max_drops > 0) {
    const bool fast = fov_fast_path(ctx, dm);
    const bool cv_rule = ctx->fill_rule == 1;              // RR_OPT_FOV_FILL_RULE: OpenCV's fillConvexPoly where it applies
    // r05 (RR_OPT_COLOUR_STREAM): two chains that only meet in k_colour can run on two streams of the library.
    //   the FOV chain   k_fov_dda -> k_fov_spans (the list) -> k_fov_sums32: three numbers per drop; integer issue, then
    //                   one 1024-thread workgroup per CU waiting on loads
    //   bookkeeping     k_plan -> k_scan -> k_dedup -> k_lists (-> k_colour): chains of dependent loads, small workgroups
    // 1 (default): the FOV chain on the second stream beside plan .. tiles .. blur, k_colour behind the blur: 31.3 ms per 512
    //    frames against 32.2 on one stream.  k_fov_sums32's workgroups (16 waves and a map row of LDS) only get onto a CU when
    //    the tile / blur kernels, which fill the LDS, leave one: 19 ms between its events instead of 2.6, and the caller's
    //    stream waits a millisecond for it at the end (profiles/r05_two_stream_timeline.txt); a high-priority stream
    //    changes nothing.
    // 2: bookkeeping (and k_colour) on the second stream, the FOV chain in front of the tile kernels on the caller's, so
    //    that the tile / blur kernels have the device to themselves: 31.6 ms -- k_fov_dda (4.5 ms beside k_plan instead of 3.4)
    //    and k_fov_sums32 (3.8 beside k_dedup / k_lists instead of 2.6) lose what the tail gains (r05_ab_w.txt).
    // 0: one in-order stream (r04).
    hipStream_t fs = s;
    const hipStream_t bs = s;                               // (r05's mode 2 had plan .. lists + k_colour on the second stream: slower, removed in r06)
    if (ctx->colour_stream) {
      // every handle on its own: one that could not be made is tried again by the next call instead of being used as null
      if (!ctx->s_col) HIPCHK(hipStreamCreateWithFlags(&ctx->s_col, hipStreamNonBlocking));
      if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      if (!ctx->ev_join) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
      HIPCHK(hipEventRecord(ctx->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(ctx->s_col, ctx->ev_fork, 0));
      fs = ctx->s_col;
    }
    // (whatever way this block is left -- an error return included -- the caller's stream waits for the second one: a
    // synchronisation of `s` then covers everything that was enqueued here)
    struct Join {
      rr_ctx* c;
      hipStream_t s, cs;
      bool open;
      hipError_t now() {
        if (!open) return hipSuccess;
        open = false;
        hipError_t e = hipEventRecord(c->ev_join, cs);
        return e != hipSuccess ? e : hipStreamWaitEvent(s, c->ev_join, 0);
      }
      ~Join() { (void)now(); }
    } join{ctx, s, ctx->s_col, fs != s};
    const int Hp = ctx->scratch_hp, Dp = (D + 1 + 7) & ~7;
    if (fast) {
      ProfScope ps(ctx, fs, "k_fov_spans");
      const int G = imin(64 / ctx->cam.n_fov, FOV_GROUPS);
      // float colour branch: a thread per drop (k_fov_dda) for the polygons float decides and that do not wrap; the rest
      // (a fraction of a percent) through the frame's list to k_fov_spans in float64.  Caller-made polygons (rr_ext_tile)
      // and the float64 colour branch: k_fov_spans for every drop.
      bool any_ext = false;
      for (int f = 0; f < n; f++) any_ext = any_ext || in[f].ext != nullptr;
      const bool dda = fov32 && ctx->fov_dda != 0 && !any_ext;
      const int v32 = fov32 ? 1 : 0;
      const dim3 grid((max_drops + 4 * G - 1) / (4 * G), n);
      if (dda) {
        HIPCHK(hipMemsetAsync(sc.fov_list_n, 0, sizeof(int32_t) * (size_t)n, fs));
        {
          hipLaunchKernelGGL(k_fov_dda, dim3((max_drops + 64 * DDA_WAVES - 1) / (64 * DDA_WAVES), n), dim3(64 * DDA_WAVES), 0, fs, ctx->d_frames, dm, ctx->cam, D, Hp, Dp,
                             cv_rule ? 1 : 0, sc);
        }
        const dim3 lgrid(imin((int)grid.x, 8), n);           // the list is short: a few workgroups per frame walk it, in float64
        if (dm.He <= 384)
          hipLaunchKernelGGL((k_fov_spans<6, true>), lgrid, dim3(256), 0, fs, ctx->d_frames, dm, ctx->cam, D, Hp, Dp, 0, cv_rule ? 1 : 0, sc);
        else if (dm.He <= 512)
          hipLaunchKernelGGL((k_fov_spans<8, true>), lgrid, dim3(256), 0, fs, ctx->d_frames, dm, ctx->cam, D, Hp, Dp, 0, cv_rule ? 1 : 0, sc);
        else
          hipLaunchKernelGGL((k_fov_spans<16, true>), lgrid, dim3(256), 0, fs, ctx->d_frames, dm, ctx->cam, D, Hp, Dp, 0, cv_rule ? 1 : 0, sc);
      } else if (dm.He <= 384)
        hipLaunchKernelGGL((k_fov_spans<6, false>), grid, dim3(256), 0, fs, ctx->d_frames, dm, ctx->cam, D, Hp, Dp, v32, cv_rule ? 1 : 0, sc);
      else if (dm.He <= 512)
        hipLaunchKernelGGL((k_fov_spans<8, false>), grid, dim3(256), 0, fs, ctx->d_frames, dm, ctx->cam, D, Hp, Dp, v32, cv_rule ? 1 : 0, sc);
      else
        hipLaunchKernelGGL((k_fov_spans<16, false>), grid, dim3(256), 0, fs, ctx->d_frames, dm, ctx->cam, D, Hp, Dp, v32, cv_rule ? 1 : 0, sc);
    } else {
      ProfScope ps(ctx, fs, "k_fov_poly");
      hipLaunchKernelGGL(k_fov_poly_general, dim3((max_drops + 127) / 128, n), dim3(128), 0, fs, ctx->d_frames, dm, ctx->cam, D, sc);
    }
    sc.colpart_f32 = (fast && fov32) ? 1 : 0;
    if (fast) {
      ProfScope ps(ctx, fs, "k_fov_sums");
      // a chunk of NT*DPT drops re-scans the band's rows, so DPT grows with the drop count (register budget: 4
      // doubles of running sums + one 16-byte span piece per drop)
      const size_t row_bytes = ((size_t)(dm.We + 1) * 2 + 16 * 2) * sizeof(double);      // P + wave totals
      // a wave owns ceil(We / waves) columns (rounded up to even), in passes of 128
      auto passes = [&](int nt) { return ((((dm.We + nt / 64 - 1) / (nt / 64)) + 1) / 2 * 2 + 127) / 128; };
      int NT = ctx->fov_threads ? ctx->fov_threads : 1024;
      if (passes(NT) > FOV_EMAX) NT = 1024;
      const bool e1 = passes(NT) <= 1;
      const int DPT = ctx->fov_dpt ? ctx->fov_dpt : (max_drops <= NT ? 1 : (max_drops <= 2 * NT ? 2 : (max_drops <= 4 * NT ? 4 : 8)));
      const int rpb = ((dm.He + COL_PARTS - 1) / COL_PARTS + 3) & ~3;
      const int nchunk = (max_drops + NT * DPT - 1) / (NT * DPT);
      const dim3 grid(COL_PARTS * 2 * nchunk, n);         // x = band + COL_PARTS * (half + 2 * chunk): a band's workgroups share an XCD
      auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(NT), row_bytes, fs, ctx->d_frames, dm, D, Hp, Dp, rpb, nchunk, sc);
        return hipSuccess;
      };
      auto launch32 = [&](auto kern) -> hipError_t {         // all four components in one workgroup: half the grid
        const size_t bytes = ((size_t)(dm.We + 1) * 4 + 16 * 4) * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(COL_PARTS * nchunk, n), dim3(NT), bytes, fs, ctx->d_frames, dm, D, Hp, Dp, rpb, nchunk, sc);
        return hipSuccess;
      };
      // (the element type of the map is a template parameter: a run-time choice inside the row loader cost the 8-drops-per-
      //  thread variant 250 bytes of scratch per lane; a batch is all float or takes the float64 loader)
      bool all32 = true;
      for (int f = 0; f < n; f++) all32 = all32 && (in[f].in_types & RR_IN_ENV_F32);
      for (int f = 0; f < n && !all32; f++)
        if (in[f].in_types & RR_IN_ENV_F32) {
          ctx->err = "RR_IN_ENV_F32 must be set for every frame of a batch or for none";
          return RR_E_ARG;
        }
      hipError_t e = fov32
                         ? (all32 ? (e1 ? (DPT == 1 ? launch32(k_fov_sums32<1, 1, true>) : DPT == 2 ? launch32(k_fov_sums32<2, 1, true>) : DPT == 4 ? launch32(k_fov_sums32<4, 1, true>) : launch32(k_fov_sums32<8, 1, true>))
                                        : (DPT == 1 ? launch32(k_fov_sums32<1, 2, true>) : DPT == 2 ? launch32(k_fov_sums32<2, 2, true>) : DPT == 4 ? launch32(k_fov_sums32<4, 2, true>) : launch32(k_fov_sums32<8, 2, true>)))
                                  : (e1 ? (DPT == 1 ? launch32(k_fov_sums32<1, 1, false>) : DPT == 2 ? launch32(k_fov_sums32<2, 1, false>) : DPT == 4 ? launch32(k_fov_sums32<4, 1, false>) : launch32(k_fov_sums32<8, 1, false>))
                                        : (DPT == 1 ? launch32(k_fov_sums32<1, 2, false>) : DPT == 2 ? launch32(k_fov_sums32<2, 2, false>) : DPT == 4 ? launch32(k_fov_sums32<4, 2, false>) : launch32(k_fov_sums32<8, 2, false>))))
                     : e1 ? (DPT == 1 ? launch(k_fov_sums<1, 1>) : DPT == 2 ? launch(k_fov_sums<2, 1>) : DPT == 4 ? launch(k_fov_sums<4, 1>) : launch(k_fov_sums<8, 1>))
                        : (DPT == 1 ? launch(k_fov_sums<1, 2>) : DPT == 2 ? launch(k_fov_sums<2, 2>) : DPT == 4 ? launch(k_fov_sums<4, 2>) : launch(k_fov_sums<8, 2>));
      if (e != hipSuccess) {
        ctx->err = std::string("k_fov_sums: ") + hipGetErrorString(e);
        return RR_E_HIP;
      }
    } else {
      {
        ProfScope ps(ctx, fs, "k_env_prefix");
        hipLaunchKernelGGL(k_env_prefix, dim3((dm.He + 3) / 4, n), dim3(256), 0, fs, ctx->d_frames, dm, sc.prefix);
        hipLaunchKernelGGL(k_env_consts, dim3(n), dim3(256), 0, fs, dm, sc.prefix, sc.fband);
      }
      {
        ProfScope ps(ctx, fs, "k_fov_sums_general");
        hipLaunchKernelGGL(k_fov_sums_general, dim3((max_drops + 3) / 4, n), dim3(256), 0, fs, ctx->d_frames, dm, D, sc, ctx->fill_rule, ctx->cam.n_fov);
      }
    }
    {
      ProfScope ps(ctx, bs, "k_plan");
      HIPCHK(hipMemsetAsync(sc.bigs_n, 0, sizeof(int32_t) * (size_t)n, bs));
      hipLaunchKernelGGL(k_plan, dim3((max_drops + 127) / 128, n), dim3(128), 0, bs, ctx->d_frames, dm, ctx->cam, ctx->d_tex_h,
                         ctx->d_tex_w, D, fs == bs ? 1 : 0, sc);
      hipLaunchKernelGGL(k_plan_big, dim3((max_drops + 127) / 128, n), dim3(128), 0, bs, ctx->d_frames, ctx->d_tex_h, ctx->d_tex_w, D, sc);
    }
    {
      ProfScope ps(ctx, bs, "k_scan");
      hipLaunchKernelGGL(k_scan, dim3(n), dim3(1024), 0, bs, ctx->d_frames, D, ctx->arena_cap, sc);
    }
    {
      ProfScope ps(ctx, bs, "k_dedup");
      HIPCHK(hipMemsetAsync(sc.htab, 0, sizeof(int32_t) * 2 * (size_t)n * D, bs));
      HIPCHK(hipMemsetAsync(sc.counts, 0, sizeof(int32_t) * 8 * (size_t)n, bs));
      HIPCHK(hipMemsetAsync(sc.rows_hist, 0, sizeof(int32_t) * (RW_TEX_MAX * 3 + 16), bs));      // histogram, cost sums, share counter, rot_total
      hipLaunchKernelGGL(k_dedup, dim3((max_drops + 255) / 256, n), dim3(256), 0, bs, ctx->d_frames, D, n, ctx->dedup ? 1 : 0, sc);
    }
    {
      ProfScope ps(ctx, bs, "k_lists");
      hipLaunchKernelGGL(k_lists, dim3(n), dim3(1024), 0, bs, ctx->d_frames, D, ctx->d_tex_h, ctx->d_tex_w, sc);
      if (sc.rows_on) {
        hipLaunchKernelGGL(k_rows_scatter, dim3(n), dim3(1024), 0, bs, D, sc);
        hipLaunchKernelGGL(k_rows_shares, dim3(1), dim3(1024), 0, bs, rows_wgs * ctx->rows_shares, sc);
      }
    }
    {
      ProfScope ps(ctx, s, "k_tile_generic");
      // (rare modes: a few workgroups per frame take them grid-stride; r06: 64 per frame were 32 K workgroups per 512 frames, 0.2 ms of
      //  launches that found nothing to do)
      hipLaunchKernelGGL(k_tile_generic, dim3(imin((max_drops + 3) / 4, imax(2, imin(64, 2048 / n))), n), dim3(256), 0, s, ctx->d_frames, D, ctx->d_tex,
                         ctx->d_tex_h, ctx->d_tex_w, ctx->d_tex_off, ctx->d_ctab, sc);
    }
    {
      ProfScope ps(ctx, s, "k_tile_big");
      hipLaunchKernelGGL(k_tile_big, dim3(sc.big_on ? imax(16, imin(1024, 16384 / n)) : 1024, n), dim3(256), 0, s, ctx->d_frames, D, ctx->d_tex, ctx->d_tex_h, ctx->d_tex_w,
                         ctx->d_tex_off, ctx->d_ctab, sc);
    }
    if (sc.rows_on) {
      ProfScope ps(ctx, s, "k_tile_rows");
      // one workgroup of 16 waves per CU (the LDS holds one texture + 16 wave-private tables); fewer for small batches:
      // a workgroup's share of the list should be worth staging a texture for
      hipLaunchKernelGGL(k_tile_rows, dim3(rows_wgs), dim3(64 * RW_WAVES), 0, s, rows_wgs * ctx->rows_shares, ctx->d_tex_h, ctx->d_tex_w, ctx->d_ctab, sc);
    }
    {
      ProfScope ps(ctx, s, "k_tile");
      // after de-duplication a frame keeps a fraction of its tiles: a capped grid (items are taken
      // grid-stride) avoids dispatching tens of thousands of empty workgroups
      // one batch-wide list, taken grid-stride (the length only exists on the device): with k_tile_rows in front only the
      // integer-ratio and out-of-range tiles are left -- 64 workgroups per FRAME were 0.8 ms of empty launches per 512 frames
      const int64_t cap = sc.rows_on ? 2048 : 16384;
      hipLaunchKernelGGL(k_tile, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)n * max_drops, cap))), dim3(256), 0, s, ctx->d_frames, D, ctx->d_tex,
                         ctx->d_tex_h, ctx->d_tex_w, ctx->d_tex_off, sc);
    }
    {
      ProfScope ps(ctx, s, "k_blur_weights");
      hipLaunchKernelGGL(k_blur_weights, dim3((2 * max_drops + 256 / BW_LANES - 1) / (256 / BW_LANES), n), dim3(256), 0, s, ctx->d_frames, D, sc);
    }
    {
      ProfScope ps(ctx, s, "k_blur_small");
      hipLaunchKernelGGL(k_blur_small, dim3(imin((max_drops + 3) / 4, grid_cap(2048)), n), dim3(256), 0, s, ctx->d_frames, D, sc);
    }
    {
      ProfScope ps(ctx, s, "k_blur_fused");
      const size_t lds = sizeof(double) * (size_t)(2 * (BR_MAX + 1) + sc.blur_bx + sc.blur_by);
      const dim3 grid(imin((max_drops + 1) / 2, grid_cap(4096)), n);
      if (ctx->blur_dma) {                                 // r05: staged a sub-tile ahead by LDS-DMA; + a second generation of weight tables
        const size_t lds2 = lds + sizeof(double) * 2 * (BR_MAX + 1);
        if (blur_wg == 3) hipLaunchKernelGGL(k_blur_fused_dma<3>, grid, dim3(256), lds2, s, ctx->d_frames, D, sc);
        else if (blur_wg == 5) hipLaunchKernelGGL(k_blur_fused_dma<5>, grid, dim3(256), lds2, s, ctx->d_frames, D, sc);
        else hipLaunchKernelGGL(k_blur_fused_dma<4>, grid, dim3(256), lds2, s, ctx->d_frames, D, sc);
      }
#ifdef RR_EXPERIMENTS
      else if (blur_wg == 3) hipLaunchKernelGGL(k_blur_fused<3>, grid, dim3(256), lds, s, ctx->d_frames, D, sc);
      else if (blur_wg == 5) hipLaunchKernelGGL(k_blur_fused<5>, grid, dim3(256), lds, s, ctx->d_frames, D, sc);
      else hipLaunchKernelGGL(k_blur_fused<4>, grid, dim3(256), lds, s, ctx->d_frames, D, sc);
#endif
    }
    {
      ProfScope ps(ctx, s, "k_blur_big_weights");
      hipLaunchKernelGGL(k_blur_big_weights, dim3(64, n), dim3(256), 0, s, ctx->d_frames, D, sc);
    }
    {
      ProfScope ps(ctx, s, "k_blur_rows");
      hipLaunchKernelGGL(k_blur<0>, dim3(256, n), dim3(256), 0, s, ctx->d_frames, D, sc);
    }
    {
      ProfScope ps(ctx, s, "k_blur_cols");
      hipLaunchKernelGGL(k_blur<1>, dim3(256, n), dim3(256), 0, s, ctx->d_frames, D, sc);
    }
    HIPCHK(join.now());                                    // the caller's stream waits for the second one: k_bin needs k_colour's records
    {                                                      // behind the blur (mode 1: and behind the join)
      ProfScope ps(ctx, s, "k_colour");
      hipLaunchKernelGGL(k_colour, dim3((max_drops + 255) / 256, n), dim3(256), 0, s, ctx->d_frames, dm, D, ctx->cam.exposure_s, sc);
    }
  }
  const int ctiles_x = (dm.W + CTILE - 1) / CTILE, nct = ctiles_x * ((dm.H + CTILE - 1) / CTILE);
  {
    ProfScope ps(ctx, s, "k_bin");
    if (ctx->bin_rows && ctiles_x <= 64)
      hipLaunchKernelGGL(k_bin_rows, dim3(nct / ctiles_x, n), dim3(256), 0, s, ctx->d_frames, dm, D, ctiles_x, nct, sc);
    else
      hipLaunchKernelGGL(k_bin, dim3(nct, n), dim3(256), 0, s, ctx->d_frames, dm, D, ctiles_x, nct, sc);
  }
  sc.pad_first = sc.eff_first = nullptr;
  bool wild = ctx->wild_pixels;
  for (int f = 0; f < n; f++) wild = wild && !in[f].ext;
  if (wild) {
    const size_t need = (size_t)n * dm.H * dm.W;
    if (need > ctx->pad_cap) {
      HIPCHK(hipDeviceSynchronize());
      int rc;
      if ((rc = dev_alloc(ctx, ctx->d_pad_first, need))) return rc;
      if ((rc = dev_alloc(ctx, ctx->d_eff_first, need))) return rc;
      ctx->pad_cap = need;
    }
    sc.pad_first = ctx->d_pad_first;
    sc.eff_first = ctx->d_eff_first;
    HIPCHK(hipMemsetAsync(sc.pad_first, 0x7f, sizeof(int32_t) * need, s));
    HIPCHK(hipMemsetAsync(sc.eff_first, 0x7f, sizeof(int32_t) * need, s));
    ProfScope ps(ctx, s, "k_pad_visits");
    hipLaunchKernelGGL(k_pad_visits, dim3((D + 3) / 4, n), dim3(256), 0, s, ctx->d_frames, dm, D, sc);
  }
  // float colours unless a caller wants the float64 composite (or RR_OPT_COMPOSITE_F64): see k_composite32
  int ntiles_c = ntiles;
  if (use32) {
    const int tiles_y32 = (dm.H + TILE32_H - 1) / TILE32_H;
    ntiles_c = tiles_x * tiles_y32;
    ProfScope ps(ctx, s, "k_composite");
    const dim3 grid(((ntiles_c + 7) / 8) * 8, n);
#define RR_COMP32(W, B) hipLaunchKernelGGL((k_composite32<W, B>), grid, dim3(256), 0, s, ctx->d_frames, dm, ctx->cam, D, tiles_x, tiles_y32, ctiles_x, nct, ctx->arena_cap, sc)
    const int cw = ctx->comp_waves ? ctx->comp_waves : (ctx->comp_batch ? 5 : 6);
    if (ctx->comp_batch) {
      if (cw == 8) RR_COMP32(8, true); else if (cw == 7) RR_COMP32(7, true); else if (cw == 6) RR_COMP32(6, true); else if (cw == 5) RR_COMP32(5, true); else RR_COMP32(4, true);
    } else {
      if (cw == 8) RR_COMP32(8, false); else if (cw == 7) RR_COMP32(7, false); else RR_COMP32(6, false);
    }
#undef RR_COMP32
  } else {
    ProfScope ps(ctx, s, "k_composite");
    hipLaunchKernelGGL(k_composite, dim3(((ntiles + 7) / 8) * 8, n), dim3(256), 0, s, ctx->d_frames, dm, ctx->cam, D, tiles_x, tiles_y, ctiles_x, nct,
                       sc);
  }
  {
    ProfScope ps(ctx, s, "k_means");
    hipLaunchKernelGGL(k_means, dim3(n), dim3(256), 0, s, dm, ntiles_c, sc);
  }
  {
    ProfScope ps(ctx, s, "k_finalize");
    if (use32 && ctx->composite_u16 && !ctx->wild_pixels)       // (every frame's comp_f32 is 2: the coded composite, four pixels per thread)
      hipLaunchKernelGGL(k_finalize16, dim3((unsigned)(((int64_t)dm.H * dm.W + 1023) / 1024), n), dim3(256), 0, s, ctx->d_frames, dm, sc);
    else
      hipLaunchKernelGGL(k_finalize, dim3((unsigned)(((int64_t)dm.H * dm.W + 255) / 256), n), dim3(256), 0, s, ctx->d_frames, dm, sc);
  }
  if (want_png) {
    ProfScope ps(ctx, s, "k_png_rows");
    const dim3 grid((unsigned)(((int64_t)dm.H * dm.W + 255) / 256), n);
    hipLaunchKernelGGL(k_png_image, grid, dim3(256), 0, s, ctx->d_frames, dm);
    if (ctx->have_lut) hipLaunchKernelGGL(k_png_mask, grid, dim3(256), 0, s, ctx->d_frames, dm, ctx->d_lut, sc);
  }
  if (want_png && ctx->png_deflate) {
    const int64_t n_bytes = (int64_t)dm.H * (1 + 4 * (int64_t)dm.W);
    const int nb = (int)rrz::blocks_of(n_bytes);
    const size_t need = (size_t)2 * n * nb;
    if (need > ctx->pngz_cap) {
      HIPCHK(hipDeviceSynchronize());
      int rc;
      if ((rc = dev_alloc(ctx, ctx->d_pngz_slots, need * rrz::SLOT_BYTES))) return rc;
      if ((rc = dev_alloc(ctx, ctx->d_pngz_meta, need))) return rc;
      ctx->pngz_cap = need;
    }
    if (!ctx->pngz_attr) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pngz_blocks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(rrz::BlockState)));
      ctx->pngz_attr = true;
    }
    ProfScope ps(ctx, s, "k_pngz");
    hipLaunchKernelGGL(k_pngz_blocks, dim3(nb, 2 * n), dim3(rrz::NT), sizeof(rrz::BlockState), s, ctx->d_frames, n_bytes, nb, ctx->d_pngz_slots, ctx->d_pngz_meta);
    hipLaunchKernelGGL(k_pngz_pack, dim3(nb, 2 * n), dim3(256), 0, s, ctx->d_frames, n_bytes, nb, ctx->d_pngz_slots, ctx->d_pngz_meta);
  }
  HIPCHK(hipGetLastError());
  ctx->last_n = n;
  return RR_OK;
}

int enqueue_prepass(rr_ctx* ctx, int n, const rr_prepass_in* in, const rr_prepass_out* out, hipStream_t s) {
  if (!ctx->have_pk) {
    ctx->err = "rr_set_prepass_kernels must be called before the pre-pass";
    return RR_E_STATE;
  }
  if (n <= 0 || !in || !out) {
    ctx->err = "bad pre-pass batch";
    return RR_E_ARG;
  }
  const int H = in[0].H, W = in[0].W;
  if (H <= 0 || W <= 0) {
    ctx->err = "bad frame size";
    return RR_E_ARG;
  }
  bool want_env = false;
  const bool env_only = in[0].mode == RR_PRE_ENV_ONLY;
  for (int f = 0; f < n; f++) {
    if (in[f].H != H || in[f].W != W) {
      ctx->err = "all frames of a batch must share H,W";
      return RR_E_ARG;
    }
    if ((in[f].mode != 0 && in[f].mode != RR_PRE_ENV_ONLY) || (in[f].mode == RR_PRE_ENV_ONLY) != env_only) {
      ctx->err = "pre-pass: mode must be 0 or RR_PRE_ENV_ONLY, the same for every frame of a batch";
      return RR_E_ARG;
    }
    if (env_only ? (!in[f].bg || !(out[f].env_xyY || out[f].env_bgr_u8))
                 : (!in[f].bg || !in[f].depth || !out[f].rainy_bg || !(in[f].irr_den != 0.0))) {
      ctx->err = "pre-pass: null pointer or zero irradiance denominator (bg_u8 is for the host entry points only)";
      return RR_E_ARG;
    }
    if (in[f].depth_f64 < 0 || in[f].depth_f64 > RR_DEPTH_U16) {
      ctx->err = "pre-pass: depth_f64 is 0 (float32), 1 (float64) or RR_DEPTH_U16";
      return RR_E_ARG;
    }
    if ((in[f].in_types & ~(RR_IN_BG_F32 | RR_IN_BG_U8)) || in[f].in_types == (RR_IN_BG_F32 | RR_IN_BG_U8) ||
        (out[f].out_types & ~(RR_OUT_RAINY_F32 | RR_OUT_ENV_F32)) || out[f].out_types != out[0].out_types) {
      ctx->err = "pre-pass: in_types is RR_IN_BG_F32 or RR_IN_BG_U8 (or 0), out_types RR_OUT_* bits, the same for every frame";
      return RR_E_ARG;
    }
    if (out[f].env_xyY || out[f].env_bgr_u8) want_env = true;
  }
  if (want_env && (!ctx->have_eg || ctx->eg.H != H || ctx->eg.W != W)) {
    ctx->err = "rr_set_envmap_geometry must be called for this frame size before an environment map is requested";
    return RR_E_STATE;
  }
  const int We = want_env ? ctx->eg.We : 0;
  const bool tiled = ctx->pk.fog_k == 25;        // the reference's 25 taps: the one-kernel fog layer (FogTile), no float64 planes in HBM
  const bool planes = !env_only && !tiled;
  if (n > ctx->pre_frames || H != ctx->pre_H || W != ctx->pre_W || We > ctx->pre_We || (planes && !ctx->pre_planes)) {
    HIPCHK(hipDeviceSynchronize());
    const int F = n > ctx->pre_frames ? n : ctx->pre_frames;
    const int we = We > ctx->pre_We ? We : ctx->pre_We;
    const size_t px = (size_t)H * W, ex = (size_t)H * (we > 0 ? we : 1);
    const bool pl = planes || ctx->pre_planes;
    int rc;
    if ((rc = dev_alloc(ctx, ctx->psc.fext, pl ? F * px : 1))) return rc;
    if ((rc = dev_alloc(ctx, ctx->psc.tmpF, pl ? F * px : 1))) return rc;
    if ((rc = dev_alloc(ctx, ctx->psc.tmpL, pl ? F * px * 3 : 1))) return rc;
    if ((rc = dev_alloc(ctx, ctx->psc.r8, F * px * 3))) return rc;
    if ((rc = dev_alloc(ctx, ctx->psc.part, (size_t)F * rrpre::FOG_BLOCKS * 3))) return rc;
    if ((rc = dev_alloc(ctx, ctx->psc.mean, (size_t)F * 3))) return rc;
    if ((rc = dev_alloc(ctx, ctx->psc.epack, F * ex))) return rc;
    if ((rc = dev_alloc(ctx, ctx->psc.etmp, F * ex * 3))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_pre, (size_t)F))) return rc;
    ctx->pre_frames = F;
    ctx->pre_H = H;
    ctx->pre_W = W;
    ctx->pre_We = we;
    ctx->pre_planes = pl;
  }
  int ring_idx, rrc;
  void* ring_host;
  if ((rrc = ring_acquire(ctx, ctx->ring_pre, sizeof(rrpre::PreFrame) * (size_t)n, ring_idx, ring_host))) return rrc;
  rrpre::PreFrame* h_pre = static_cast<rrpre::PreFrame*>(ring_host);
  for (int f = 0; f < n; f++) {
    rrpre::PreFrame& p = h_pre[f];
    p.bg = in[f].bg;
    p.depth = in[f].depth;
    p.rainy = env_only ? nullptr : out[f].rainy_bg;     // (map-only: the map kernels read the caller's image, `bg`)
    p.env_xyY = out[f].env_xyY;
    p.env_u8 = out[f].env_bgr_u8;
    p.r8 = (!env_only && want_env) ? ctx->psc.r8 + (size_t)f * H * W * 3 : nullptr;
    p.beta_ext = in[f].beta_ext;
    p.beta_hg = in[f].beta_hg;
    p.irr_num = in[f].irr_num;
    p.irr_den = in[f].irr_den;
    p.depth_f64 = in[f].depth_f64 == 1 ? 1 : 0;            // (uint16 samples become float32 metres: the float32 arithmetic)
    p.types = (in[f].depth_f64 == RR_DEPTH_U16 ? rrpre::PRE_DEPTH_U16 : 0) |
              ((in[f].in_types & RR_IN_BG_F32) ? rrpre::PRE_BG_F32 : 0) | ((in[f].in_types & RR_IN_BG_U8) ? rrpre::PRE_BG_U8 : 0) |
              ((out[f].out_types & RR_OUT_RAINY_F32) ? rrpre::PRE_RAINY_F32 : 0) | ((out[f].out_types & RR_OUT_ENV_F32) ? rrpre::PRE_ENV_F32 : 0);
  }
  hipLaunchKernelGGL(k_copy_small, dim3(16), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(h_pre), reinterpret_cast<uint32_t*>(ctx->d_pre),
                     (int)(sizeof(rrpre::PreFrame) * (size_t)n / 4));
  if ((rrc = ring_commit(ctx, ctx->ring_pre, ring_idx, s))) return rrc;
  const rrpre::PreScratch sc = ctx->psc;
  const unsigned px_blocks = (unsigned)(((int64_t)H * W + 255) / 256);
  if (!env_only) {
    ProfScope ps(ctx, s, "k_fog_stats");
    hipLaunchKernelGGL(rrpre::k_fog_sum, dim3(rrpre::FOG_BLOCKS, n), dim3(256), 0, s, ctx->d_pre, H, W, sc);
    hipLaunchKernelGGL(rrpre::k_fog_mean, dim3(n), dim3(64), 0, s, H, W, sc);
    if (!tiled) hipLaunchKernelGGL(rrpre::k_fog_ext, dim3(px_blocks, n), dim3(256), 0, s, ctx->d_pre, H, W, sc);
  }
  if (!env_only && tiled) {
    // column strips x row segments x frames; a segment costs 2 * 12 rows of horizontal sums it shares with its neighbours,
    // so segments are only cut while the grid is short of a few workgroups per slot (3 per CU fit: 46 KB of LDS each)
    using T = rrpre::FogTile<12>;
    const int strips = (W + T::TC - 1) / T::TC;
    int segs = 1;
    while ((int64_t)strips * n * segs < 3072 && H / (segs + 1) >= 64) segs++;
    const int seg_rows = (((H + segs - 1) / segs) + T::RB - 1) / T::RB * T::RB;
    ProfScope ps(ctx, s, "k_fog_tile");
    hipLaunchKernelGGL(rrpre::k_fog_tile<12>, dim3(strips, (H + seg_rows - 1) / seg_rows, n), dim3(256), 0, s, ctx->d_pre, H, W, seg_rows, ctx->pk, sc);
  }
  if (!env_only && !tiled) {
    ProfScope ps(ctx, s, "k_fog_h");
    hipLaunchKernelGGL(rrpre::k_fog_h, dim3((W + rrpre::FOG_SEG - 1) / rrpre::FOG_SEG, H, n), dim3(256), 0, s, ctx->d_pre, H, W, ctx->pk, sc);
  }
  if (!env_only && !tiled) {
    ProfScope ps(ctx, s, "k_fog_v");
    hipLaunchKernelGGL(rrpre::k_fog_v, dim3((W + 255) / 256, H, n), dim3(256), 0, s, ctx->d_pre, H, W, ctx->pk, sc);
  }
  if (want_env) {
    if (ctx->need_half != ctx->pk.env_k / 2) {           // once per geometry and tap count
      std::vector<uint8_t> need((size_t)H * ctx->eg.We);
      rrpre::build_env_need(H, ctx->eg.cw, ctx->pk.env_k / 2, ctx->eg_src_host.data(), need.data());
      HIPCHK(hipStreamSynchronize(s));
      int rc = dev_alloc(ctx, ctx->d_need_h, need.size());
      if (rc) return rc;
      HIPCHK(hipMemcpy(ctx->d_need_h, need.data(), need.size(), hipMemcpyHostToDevice));
      ctx->eg.need_h = ctx->d_need_h;
      ctx->need_half = ctx->pk.env_k / 2;
    }
    const rrpre::EnvGeom g = ctx->eg;
    const dim3 grid((g.We + 255) / 256, H, n);
    {
      ProfScope ps(ctx, s, "k_env_build");
      hipLaunchKernelGGL(rrpre::k_env_build, grid, dim3(256), 0, s, ctx->d_pre, g, sc);
    }
    {
      ProfScope ps(ctx, s, "k_env_h");
      hipLaunchKernelGGL(rrpre::k_env_h, grid, dim3(256), 0, s, g, ctx->pk, sc);
    }
    {
      ProfScope ps(ctx, s, "k_env_v");
      hipLaunchKernelGGL(rrpre::k_env_v, grid, dim3(256), 0, s, ctx->d_pre, g, ctx->pk, sc);
    }
  }
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// particle generator + packer of n frames (device output); see include/rainhip.h
int enqueue_particles(rr_ctx* ctx, int n, const rr_sim_frame* sims, int H, int W, rr_drop* drops_out, int cap, int32_t* n_out, hipStream_t s) {
  if (n <= 0 || !sims || !drops_out || !n_out || cap <= 0 || H <= 0 || W <= 0) {
    ctx->err = "rr_generate_drops_device: bad argument";
    return RR_E_ARG;
  }
  if (!ctx->have_db || !ctx->d_dgrid) {
    ctx->err = "particle generator: the streak database and the diameter tables (rr_set_particle_tables) must be set first";
    return RR_E_STATE;
  }
  if (ctx->n_ratio < 4) {
    ctx->err = "particle generator: the streak database has fewer than four distinct width / height ratios (take_drop_texture, bad_weather.py:250-265)";
    return RR_E_STATE;
  }
  for (int f = 0; f < n; f++) {
    const rr_sim_frame& sf = sims[f];
    if (sf.n_particles < 0 || sf.sensor_w <= 0 || sf.sensor_h <= 0 || sf.render_scale <= 0 || sf.table < 0 || sf.table >= ctx->n_tables ||
        !(sf.fpx > 0) || !(sf.min_px > 0) || !(sf.z_far > 0) || !(sf.exposure_s > 0) || !(sf.margin >= 0) ||
        sf.sensor_w / sf.render_scale != W || sf.sensor_h / sf.render_scale != H) {
      ctx->err = "rr_sim_frame: bad settings (sizes must be positive, sensor / render_scale must be the rendered frame, table in range)";
      return RR_E_ARG;
    }
  }
  int rc;
  if (n > ctx->cap_sims) {
    HIPCHK(hipDeviceSynchronize());
    if ((rc = dev_alloc(ctx, ctx->d_sims, (size_t)n))) return rc;
    ctx->cap_sims = n;
  }
  int ring_idx;
  void* host;
  if ((rc = ring_acquire(ctx, ctx->ring_sims, sizeof(rr_sim_frame) * (size_t)n, ring_idx, host))) return rc;
  memcpy(host, sims, sizeof(rr_sim_frame) * (size_t)n);
  hipLaunchKernelGGL(k_copy_small, dim3(16), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(host), reinterpret_cast<uint32_t*>(ctx->d_sims),
                     (int)(sizeof(rr_sim_frame) * (size_t)n / 4));
  if ((rc = ring_commit(ctx, ctx->ring_sims, ring_idx, s))) return rc;
  {
    ProfScope ps(ctx, s, "k_particles");
    hipLaunchKernelGGL(k_particles, dim3(n), dim3(512), 0, s, ctx->d_sims, H, W, ctx->d_dgrid, ctx->d_cdf, ctx->n_grid, ctx->d_ratio_db,
                       drops_out, cap, n_out);
  }
  {
    ProfScope ps(ctx, s, "k_particle_draws");
    hipLaunchKernelGGL(k_particle_draws, dim3(n), dim3(64), 0, s, ctx->d_sims, drops_out, cap, n_out);
  }
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// returns RR_OK, or RR_E_ARENA after growing the arena (caller re-enqueues)
int check_overflow(rr_ctx* ctx, hipStream_t s) {
  int32_t ovf = 0;
  HIPCHK(hipMemcpyAsync(&ovf, ctx->sc.overflow, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (!ovf) return RR_OK;
  int rc = grow_arena(ctx);
  if (rc) return rc;
  HIPCHK(hipMemset(ctx->sc.overflow, 0, sizeof(int32_t)));      // flag 0: the device-pointer calls since the last rr_synchronize
  ctx->err = "tile arena overflow: arena regrown, re-enqueue the batch";
  return RR_E_ARENA;
}

}  // namespace

extern "C" {

int rr_version(void) { return RR_VERSION; }
int rr_sizeof_drop(void) { return (int)sizeof(rr_drop); }
int rr_sizeof_camera(void) { return (int)sizeof(rr_camera); }
int rr_sizeof_frame_in(void) { return (int)sizeof(rr_frame_in); }
int rr_sizeof_frame_out(void) { return (int)sizeof(rr_frame_out); }
int rr_sizeof_prepass_in(void) { return (int)sizeof(rr_prepass_in); }
int rr_sizeof_prepass_out(void) { return (int)sizeof(rr_prepass_out); }
int rr_sizeof_prepass_kernels(void) { return (int)sizeof(rr_prepass_kernels); }

int rr_create(rr_ctx** out, int device) {
  if (!out) return RR_E_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return RR_E_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RR_E_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx9", 4) != 0) return RR_E_NO_DEVICE;
  // every kernel here is written for wave64: ballots are 64 bits wide, scans run over 64 lanes, k_png_unfilter and the
  // wave-per-item kernels hand data between the lanes of ONE wave without a barrier (a 64-thread block must be one wave)
  if (prop.warpSize != 64) return RR_E_NO_DEVICE;
  rr_ctx* ctx = new rr_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return RR_E_HIP;
  }
  float tab[128];
  build_cubic_tab(tab);
  if (hipMalloc((void**)&ctx->d_ctab, sizeof(tab)) != hipSuccess ||
      hipMemcpy(ctx->d_ctab, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess ||
      // arena-overflow flags (one per pipeline slot + one for the device-pointer calls) and the sticky largest need
      hipMalloc((void**)&ctx->sc.overflow, sizeof(int32_t) * (1 + RR_PIPE_SLOTS)) != hipSuccess ||
      hipMemset(ctx->sc.overflow, 0, sizeof(int32_t) * (1 + RR_PIPE_SLOTS)) != hipSuccess ||
      hipMalloc((void**)&ctx->sc.need_max, sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(ctx->sc.need_max, 0, sizeof(unsigned long long)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    rr_destroy(ctx);
    return RR_E_HIP;
  }
  *out = ctx;
  return RR_OK;
}

int rr_destroy(rr_ctx* ctx) {
  if (!ctx) return RR_E_ARG;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  if (ctx->own_tex) hipFree(ctx->d_tex);
  hipFree(ctx->d_tex_h);
  hipFree(ctx->d_tex_w);
  hipFree(ctx->d_tex_off);
  hipFree(ctx->d_tex_pad);
  hipFree(ctx->d_tex_poff);
  hipFree(ctx->d_ctab);
  hipFree(ctx->d_lut);
  hipFree(ctx->sc.plan);
  hipFree(ctx->sc.comp);
  hipFree(ctx->sc.comp32);
  hipFree(ctx->sc.poly);
  hipFree(ctx->sc.npts);
  hipFree(ctx->sc.sizes);
  hipFree(ctx->sc.list_rot);
  hipFree(ctx->sc.fov_erec);
  hipFree(ctx->sc.fov_pix);
  hipFree(ctx->sc.rows_list);
  hipFree(ctx->sc.rows_sorted);
  hipFree(ctx->sc.rows_n);
  hipFree(ctx->sc.rows_hist);
  hipFree(ctx->sc.rows_fbase);
  hipFree(ctx->sc.rows_bounds);
  hipFree(ctx->d_tex_pair);
  hipFree(ctx->d_tex_qoff);
  hipFree(ctx->sc.fov_list);
  hipFree(ctx->sc.fov_list_n);
  hipFree(ctx->sc.list_gen);
  hipFree(ctx->sc.list_slow);
  hipFree(ctx->sc.blur_items);
  hipFree(ctx->sc.counts);
  hipFree(ctx->sc.canon);
  hipFree(ctx->sc.bigs_list);
  hipFree(ctx->sc.bigs_n);
  hipFree(ctx->sc.tkey);
  hipFree(ctx->sc.lrec);
  hipFree(ctx->sc.list_big);
  hipFree(ctx->sc.big_off);
  hipFree(ctx->sc.htab);
  hipFree(ctx->sc.list_small);
  hipFree(ctx->sc.colpart);
  hipFree(ctx->sc.wtab);
  hipFree(ctx->sc.wtab_big);
  hipFree(ctx->sc.spans);
  hipFree(ctx->sc.bbox);
  hipFree(ctx->sc.blended);
  hipFree(ctx->d_pad_first);
  hipFree(ctx->d_pngz_slots);
  hipFree(ctx->d_pngz_meta);
  hipFree(ctx->d_eff_first);
  hipFree(ctx->sc.clist);
  hipFree(ctx->sc.ccount);
  hipFree(ctx->sc.prefix);
  hipFree(ctx->sc.fband);
  hipFree(ctx->sc.arena);
  hipFree(ctx->sc.partial);
  hipFree(ctx->sc.means);
  hipFree(ctx->sc.arena_need);
  hipFree(ctx->sc.overflow);
  hipFree(ctx->sc.need_max);
  hipFree(ctx->d_omega);
  hipFree(ctx->d_omega32);
  hipFree(ctx->d_dgrid);
  hipFree(ctx->d_cdf);
  hipFree(ctx->d_ratio_db);
  hipFree(ctx->d_sims);
  hipFree(ctx->d_gen_drops);
  hipFree(ctx->d_gen_counts);
  for (auto* r : {&ctx->ring_frames, &ctx->ring_pre, &ctx->ring_sims})
    for (int k = 0; k < rr_ctx::DescRing::N; k++) {
      if (r->host[k]) hipHostFree(r->host[k]);
      if (r->ev[k]) hipEventDestroy(r->ev[k]);
    }
  hipFree(ctx->d_frames);
  hipFree(ctx->d_comp_out);
  for (auto& sl : ctx->slots) {
    hipFree(sl.st.bg);
    hipFree(sl.st.rainy);
    hipFree(sl.st.env);
    hipFree(sl.st.omega);
    hipFree(sl.st.comp);
    hipFree(sl.st.mask);
    hipFree(sl.st.drops);
    hipFree(sl.st.rgb);
    hipFree(sl.st.mask_i);
    hipFree(sl.st.status);
    hipFree(sl.st.colour);
    hipFree(sl.st.ndrops);
    hipFree(sl.st.depth);
    hipFree(sl.st.env_u8);
    hipFree(sl.st.bg8);
    hipFree(sl.st.png_i);
    hipFree(sl.st.png_m);
    if (sl.ev_up) hipEventDestroy(sl.ev_up);
    if (sl.ev_comp) hipEventDestroy(sl.ev_comp);
    if (sl.ev_down) hipEventDestroy(sl.ev_down);
    if (sl.h_flags) hipHostFree(sl.h_flags);
    hipFree(sl.d_up);
    hipFree(sl.d_down);
    if (sl.h_up) hipHostFree(sl.h_up);
    if (sl.h_down) hipHostFree(sl.h_down);
    for (void* b : sl.ext_blobs) hipFree(b);
  }
  if (ctx->s_col) hipStreamDestroy(ctx->s_col);
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (ctx->s_up) hipStreamDestroy(ctx->s_up);
  if (ctx->s_down) hipStreamDestroy(ctx->s_down);
  hipFree(ctx->d_esrc);
  hipFree(ctx->d_etop);
  hipFree(ctx->d_ebot);
  hipFree(ctx->d_pre);
  hipFree(ctx->psc.fext);
  hipFree(ctx->psc.tmpF);
  hipFree(ctx->psc.tmpL);
  hipFree(ctx->psc.r8);
  hipFree(ctx->d_need_h);
  hipFree(ctx->psc.part);
  hipFree(ctx->psc.mean);
  hipFree(ctx->psc.epack);
  hipFree(ctx->psc.etmp);
  for (auto& pe : ctx->prof_pending) {
    hipEventDestroy(pe.a);
    hipEventDestroy(pe.b);
  }
  for (auto e : ctx->ev_pool) hipEventDestroy(e);
  for (auto& blk : ctx->host_allocs) hipHostFree(const_cast<char*>(blk.first));      // rr_host_alloc blocks the caller kept
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return RR_OK;
}

const char* rr_last_error(rr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int set_db_meta(rr_ctx* ctx, const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off, int32_t n_tex) {
  int rc;
  if ((rc = dev_alloc(ctx, ctx->d_tex_h, (size_t)n_tex))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_tex_w, (size_t)n_tex))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_tex_off, (size_t)n_tex))) return rc;
  HIPCHK(hipMemcpy(ctx->d_tex_h, tex_h, sizeof(int32_t) * n_tex, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_tex_w, tex_w, sizeof(int32_t) * n_tex, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_tex_off, tex_off, sizeof(int64_t) * n_tex, hipMemcpyHostToDevice));
  ctx->n_tex = n_tex;
  ctx->h_tex_h.assign(tex_h, tex_h + n_tex);
  ctx->h_tex_w.assign(tex_w, tex_w + n_tex);
  ctx->h_tex_off.assign(tex_off, tex_off + n_tex);
  ctx->have_db = true;
  {                                     // the padded copies k_tile / k_tile_generic stage from
    std::vector<int64_t> poff((size_t)n_tex);
    int64_t total = 0;
    for (int i = 0; i < n_tex; i++) {
      poff[(size_t)i] = total;
      total += (((int64_t)tex_h[i] + 4) * ((int64_t)tex_w[i] + 4) + 15) & ~15LL;
    }
    if ((rc = dev_alloc(ctx, ctx->d_tex_poff, (size_t)n_tex))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_tex_pad, (size_t)total + 16))) return rc;
    HIPCHK(hipMemcpy(ctx->d_tex_poff, poff.data(), sizeof(int64_t) * (size_t)n_tex, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ctx->d_tex_pad, 0, (size_t)total + 16));
    HIPCHK(hipDeviceSynchronize());    // (the copies above and the fill may still be in flight on the null stream; ours does not wait for it)
    hipLaunchKernelGGL(k_pad_textures, dim3(n_tex), dim3(256), 0, ctx->stream, ctx->d_tex, ctx->d_tex_h, ctx->d_tex_w, ctx->d_tex_off,
                       ctx->d_tex_poff, ctx->d_tex_pad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  {                                     // the pair textures k_tile_rows stages (texel | texel below << 8, rr_device.h)
    std::vector<int64_t> qoff((size_t)n_tex);
    int64_t total = 0;
    for (int i = 0; i < n_tex; i++) {
      qoff[(size_t)i] = total;
      total += pair_bytes(tex_h[i], tex_w[i]);
    }
    if ((rc = dev_alloc(ctx, ctx->d_tex_qoff, (size_t)n_tex))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_tex_pair, (size_t)total + 16))) return rc;
    HIPCHK(hipMemcpy(ctx->d_tex_qoff, qoff.data(), sizeof(int64_t) * (size_t)n_tex, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ctx->d_tex_pair, 0, (size_t)total + 16));
    HIPCHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_pair_textures, dim3(n_tex), dim3(256), 0, ctx->stream, ctx->d_tex, ctx->d_tex_h, ctx->d_tex_w, ctx->d_tex_off,
                       ctx->d_tex_qoff, ctx->d_tex_pair);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  {                                     // DBManager.ratio = np.unique(w / h) (bad_weather.py:134,144): take_drop_texture's thresholds
    std::vector<double> r;
    for (int i = 0; i < n_tex; i++) r.push_back((double)tex_w[i] / (double)tex_h[i]);
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
    ctx->n_ratio = (int)r.size();
    if ((rc = dev_alloc(ctx, ctx->d_ratio_db, r.size() < 4 ? 4 : r.size()))) return rc;
    HIPCHK(hipMemcpy(ctx->d_ratio_db, r.data(), sizeof(double) * r.size(), hipMemcpyHostToDevice));
  }
  return RR_OK;
}

int rr_set_streak_db(rr_ctx* ctx, const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off,
                     int32_t n_tex) {
  if (!ctx) return RR_E_ARG;
  if (!texels || !tex_h || !tex_w || !tex_off || n_tex <= 0) {
    ctx->err = "rr_set_streak_db: bad argument";
    return RR_E_ARG;
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  int64_t total = 0;
  for (int i = 0; i < n_tex; i++) {
    if (tex_h[i] <= 0 || tex_w[i] <= 0 || tex_off[i] < 0) {
      ctx->err = "rr_set_streak_db: bad texture shape";
      return RR_E_ARG;
    }
    int64_t end = tex_off[i] + (int64_t)tex_h[i] * tex_w[i];
    if (end > total) total = end;
  }
  if (ctx->own_tex && ctx->d_tex) hipFree(ctx->d_tex);
  ctx->d_tex = nullptr;
  HIPCHK(hipMalloc((void**)&ctx->d_tex, (size_t)total));
  ctx->own_tex = true;
  HIPCHK(hipMemcpy(ctx->d_tex, texels, (size_t)total, hipMemcpyHostToDevice));
  ctx->tex_bytes = total;
  return set_db_meta(ctx, tex_h, tex_w, tex_off, n_tex);
}

int rr_set_streak_db_device(rr_ctx* ctx, const uint8_t* texels_dev, int64_t n_bytes, const int32_t* tex_h, const int32_t* tex_w,
                            const int64_t* tex_off, int32_t n_tex) {
  if (!ctx) return RR_E_ARG;
  if (!texels_dev || !tex_h || !tex_w || !tex_off || n_tex <= 0 || n_bytes <= 0) {
    ctx->err = "rr_set_streak_db_device: bad argument";
    return RR_E_ARG;
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  for (int i = 0; i < n_tex; i++) {
    if (tex_h[i] <= 0 || tex_w[i] <= 0 || tex_off[i] < 0 || tex_off[i] + (int64_t)tex_h[i] * tex_w[i] > n_bytes) {
      ctx->err = "rr_set_streak_db_device: texture outside the buffer";
      return RR_E_ARG;
    }
  }
  if (ctx->own_tex && ctx->d_tex) hipFree(ctx->d_tex);
  ctx->d_tex = nullptr;
  // private copy: the caller's (broadcast) buffer may be released afterwards
  HIPCHK(hipMalloc((void**)&ctx->d_tex, (size_t)n_bytes));
  ctx->own_tex = true;
  HIPCHK(hipMemcpy(ctx->d_tex, texels_dev, (size_t)n_bytes, hipMemcpyDeviceToDevice));
  ctx->tex_bytes = n_bytes;
  return set_db_meta(ctx, tex_h, tex_w, tex_off, n_tex);
}

// SURVEY 8b's collective for a host that drives several GPUs from ONE process: the streak database of ctxs[0] reaches the
// other contexts without going back through the host.  Contexts on the root's device take a device-to-device copy; contexts on
// other devices an ncclBroadcast in one group call (RCCL, librccl.so.1 loaded on first use: the library has no link-time
// dependency on it -- this repo's own drivers run one PROCESS per GPU and broadcast through torch.distributed, sharding.py).
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
  }
};
RcclApi g_rccl;
}  // namespace

// ncclBroadcast of nb bytes from src (on devs[0]) into dst[r] on devs[r], r = 0 .. (rank 0 receives into dst[0], which may
// be src itself), one group call, synchronous
static int rccl_bcast(const std::vector<int>& devs, const void* src, const std::vector<void*>& dst, const std::vector<hipStream_t>& streams,
                      size_t nb, std::string& err) {
  if (!g_rccl.load()) {
    err = "rr_bcast_streak_db: contexts on several devices need RCCL (librccl.so.1 could not be loaded)";
    return RR_E_STATE;
  }
  std::vector<void*> comms(devs.size(), nullptr);
  int rc = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
  if (rc == 0) {
    rc = g_rccl.GroupStart();
    for (size_t r = 0; r < devs.size() && rc == 0; r++) {
      hipSetDevice(devs[r]);
      rc = g_rccl.Broadcast(src, dst[r], nb, /*ncclUint8*/ 1, 0, comms[r], streams[r]);
    }
    const int rc2 = g_rccl.GroupEnd();
    if (rc == 0) rc = rc2;
    for (size_t r = 0; r < devs.size(); r++) {
      hipSetDevice(devs[r]);
      if (hipStreamSynchronize(streams[r]) != hipSuccess && rc == 0) rc = -1;
    }
  }
  for (void* cm : comms)
    if (cm) g_rccl.CommDestroy(cm);
  if (rc != 0) {
    err = std::string("rr_bcast_streak_db: RCCL: ") + (rc > 0 && g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "stream error");
    return RR_E_HIP;
  }
  return RR_OK;
}

// What a box with ONE GPU can check of the multi-device leg: the context's database broadcast by RCCL in a communicator of
// one rank (its own device) into a scratch buffer, compared byte for byte.  RR_OK, or RR_E_STATE without RCCL / a database.
int rr_bcast_selftest(rr_ctx* ctx) {
  if (!ctx) return RR_E_ARG;
  if (!ctx->have_db || !ctx->d_tex || ctx->tex_bytes <= 0) {
    ctx->err = "rr_bcast_selftest: no streak database";
    return RR_E_STATE;
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  uint8_t* tmp = nullptr;
  HIPCHK(hipMalloc((void**)&tmp, (size_t)ctx->tex_bytes));
  HIPCHK(hipMemset(tmp, 0xa5, (size_t)ctx->tex_bytes));
  int rc = rccl_bcast({ctx->device}, ctx->d_tex, {tmp}, {ctx->stream}, (size_t)ctx->tex_bytes, ctx->err);
  if (rc == RR_OK) {
    std::vector<uint8_t> a((size_t)ctx->tex_bytes), b((size_t)ctx->tex_bytes);
    if (hipMemcpy(a.data(), ctx->d_tex, a.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(b.data(), tmp, b.size(), hipMemcpyDeviceToHost) != hipSuccess) {
      ctx->err = "rr_bcast_selftest: read-back failed";
      rc = RR_E_HIP;
    } else if (a != b) {
      ctx->err = "rr_bcast_selftest: the broadcast buffer differs from the database";
      rc = RR_E_HIP;
    }
  }
  hipFree(tmp);
  return rc;
}

int rr_bcast_streak_db(rr_ctx** ctxs, int32_t n_ctx) {
  if (!ctxs || n_ctx <= 0 || !ctxs[0]) return RR_E_ARG;
  rr_ctx* root = ctxs[0];
  if (!root->have_db || !root->d_tex || root->tex_bytes <= 0) {
    root->err = "rr_bcast_streak_db: ctxs[0] has no streak database (rr_set_streak_db first)";
    return RR_E_STATE;
  }
  for (int i = 1; i < n_ctx; i++)
    if (!ctxs[i] || ctxs[i] == root) {
      root->err = "rr_bcast_streak_db: null or repeated context";
      return RR_E_ARG;
    }
  const int64_t nb = root->tex_bytes;
  const int32_t n_tex = root->n_tex;
  HIPCHK_CTX(root, hipSetDevice(root->device));
  HIPCHK_CTX(root, hipDeviceSynchronize());
  // receiving buffers (a context's previous database goes)
  std::vector<rr_ctx*> remote;                               // contexts on other devices: one rank each, the root is rank 0
  for (int i = 1; i < n_ctx; i++) {
    rr_ctx* c = ctxs[i];
    HIPCHK_CTX(c, hipSetDevice(c->device));
    HIPCHK_CTX(c, hipDeviceSynchronize());
    if (c->own_tex && c->d_tex) hipFree(c->d_tex);
    c->d_tex = nullptr;
    HIPCHK_CTX(c, hipMalloc((void**)&c->d_tex, (size_t)nb));
    c->own_tex = true;
    c->tex_bytes = nb;
    if (c->device == root->device) HIPCHK_CTX(c, hipMemcpy(c->d_tex, root->d_tex, (size_t)nb, hipMemcpyDeviceToDevice));
    else remote.push_back(c);
  }
  if (!remote.empty()) {
    // one communicator rank per DEVICE (RCCL refuses two ranks on one): the first context of a device receives, its
    // siblings copy from it afterwards
    std::vector<int> devs{root->device};
    std::vector<rr_ctx*> first{root};
    for (rr_ctx* c : remote) {
      bool seen = false;
      for (int d : devs) seen = seen || d == c->device;
      if (!seen) { devs.push_back(c->device); first.push_back(c); }
    }
    std::vector<void*> dst;
    std::vector<hipStream_t> streams;
    for (rr_ctx* c : first) { dst.push_back(c->d_tex); streams.push_back(c->stream); }
    const int rc = rccl_bcast(devs, root->d_tex, dst, streams, (size_t)nb, root->err);
    if (rc != RR_OK) return rc;
    for (rr_ctx* c : remote) {                               // siblings of a receiving context on the same device
      bool is_first = false;
      rr_ctx* src = nullptr;
      for (size_t r = 0; r < devs.size(); r++)
        if (devs[r] == c->device) { src = first[r]; is_first = first[r] == c; }
      if (!is_first) {
        HIPCHK_CTX(c, hipSetDevice(c->device));
        HIPCHK_CTX(c, hipMemcpy(c->d_tex, src->d_tex, (size_t)nb, hipMemcpyDeviceToDevice));
      }
    }
  }
  // metadata + the derived textures (zero-bordered copies, pair textures) on every receiving context
  const std::vector<int32_t> th = root->h_tex_h, tw = root->h_tex_w;
  const std::vector<int64_t> to = root->h_tex_off;
  for (int i = 1; i < n_ctx; i++) {
    rr_ctx* c = ctxs[i];
    HIPCHK_CTX(c, hipSetDevice(c->device));
    const int rc = set_db_meta(c, th.data(), tw.data(), to.data(), n_tex);
    if (rc != RR_OK) return rc;
  }
  hipSetDevice(root->device);
  return RR_OK;
}

int rr_set_camera(rr_ctx* ctx, const rr_camera* cam) {
  if (!ctx) return RR_E_ARG;
  if (!cam || cam->n_fov < 3 || cam->n_fov > RR_MAX_FOV || !(cam->exposure_s > 0) || !(cam->tau_zero > 0)) {
    ctx->err = "rr_set_camera: bad argument";
    return RR_E_ARG;
  }
  ctx->cam = *cam;
  ctx->have_cam = true;
  return RR_OK;
}

int rr_render_frames_device(rr_ctx* ctx, int32_t n, const rr_frame_in* in, const rr_frame_out* out, void* stream) {
  if (!ctx) return RR_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return enqueue(ctx, n, in, out, s);
}

int rr_synchronize(rr_ctx* ctx) {
  if (!ctx) return RR_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  if (ctx->last_n > 0 && ctx->sc.overflow) {
    int rc = check_overflow(ctx, ctx->stream);
    if (rc) return rc;
  }
  return RR_OK;
}

int rr_set_particle_tables(rr_ctx* ctx, int32_t n_tables, int32_t n_grid, const double* d_grid, const double* cdf) {
  if (!ctx) return RR_E_ARG;
  if (n_tables <= 0 || n_grid < 2 || !d_grid || !cdf) {
    ctx->err = "rr_set_particle_tables: bad argument";
    return RR_E_ARG;
  }
  for (int t = 0; t < n_tables; t++) {
    const double* c = cdf + (size_t)t * n_grid;
    bool ok = c[0] == 0.0 && c[n_grid - 1] == 1.0;
    for (int k = 1; k < n_grid && ok; k++) ok = c[k] >= c[k - 1] && d_grid[k] > d_grid[k - 1];
    // the cell a number u in (0, 1) falls into must have a positive width: the first and the last cell are the ones u can
    // reach with cdf[j] <= u < cdf[j + 1] for every u only if no run of equal values touches ... any run is skipped by
    // the search (it returns the LAST j with cdf[j] <= u), so equal neighbours are harmless except at the very end
    ok = ok && c[n_grid - 2] < 1.0;
    if (!ok) {
      ctx->err = "rr_set_particle_tables: d_grid must ascend, every cdf must run from 0 to 1 without decreasing (and reach 1 only at its last entry)";
      return RR_E_ARG;
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  int rc;
  if ((rc = dev_alloc(ctx, ctx->d_dgrid, (size_t)n_grid))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_cdf, (size_t)n_tables * n_grid))) return rc;
  HIPCHK(hipMemcpy(ctx->d_dgrid, d_grid, sizeof(double) * n_grid, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_cdf, cdf, sizeof(double) * (size_t)n_tables * n_grid, hipMemcpyHostToDevice));
  ctx->n_grid = n_grid;
  ctx->n_tables = n_tables;
  return RR_OK;
}

int rr_generate_drops_device(rr_ctx* ctx, int32_t n, const rr_sim_frame* frames, int32_t H, int32_t W, rr_drop* drops_out, int32_t cap,
                             int32_t* n_out, void* stream) {
  if (!ctx) return RR_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  return enqueue_particles(ctx, n, frames, H, W, drops_out, cap, n_out, stream ? (hipStream_t)stream : ctx->stream);
}

int rr_generate_drops(rr_ctx* ctx, int32_t n, const rr_sim_frame* frames, int32_t H, int32_t W, rr_drop* drops_out, int32_t cap, int32_t* n_out) {
  if (!ctx) return RR_E_ARG;
  if (n <= 0 || cap <= 0 || !drops_out || !n_out) {
    ctx->err = "rr_generate_drops: bad argument";
    return RR_E_ARG;
  }
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  const size_t nd = (size_t)n * (size_t)cap;
  if (nd > ctx->cap_gen_drops) {
    HIPCHK(hipDeviceSynchronize());
    if ((rc = dev_alloc(ctx, ctx->d_gen_drops, nd))) return rc;
    ctx->cap_gen_drops = nd;
  }
  if ((size_t)n > ctx->cap_gen_counts) {
    HIPCHK(hipDeviceSynchronize());
    if ((rc = dev_alloc(ctx, ctx->d_gen_counts, (size_t)n))) return rc;
    ctx->cap_gen_counts = (size_t)n;
  }
  if ((rc = enqueue_particles(ctx, n, frames, H, W, ctx->d_gen_drops, cap, ctx->d_gen_counts, ctx->stream))) return rc;
  HIPCHK(hipMemcpyAsync(n_out, ctx->d_gen_counts, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(drops_out, ctx->d_gen_drops, sizeof(rr_drop) * nd, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RR_OK;
}

int rr_sizeof_sim_frame(void) { return (int)sizeof(rr_sim_frame); }

}  // extern "C"

// ---------------------------------------------------------------------------
// host-pointer entry points: slots of device staging, three streams
// ---------------------------------------------------------------------------
// A batch given by HOST pointers goes through one of RR_PIPE_SLOTS staging slots:
//   upload   (stream s_up)     caller's buffers -> slot staging (pinned buffers from rr_host_alloc move at PCIe rate)
//   compute  (ctx->stream)     [bytes -> unit interval] -> [pre-pass] -> [hot path], device scratch shared by all slots
//   download (stream s_down)   slot staging -> caller's buffers, plus the arena-overflow flag
// chained by events, so that the upload of batch k+1 and the download of batch k-1 run under the kernels of
// batch k.  rr_render_frames / rr_prepass_frames / rr_pipeline_frames are submit + wait on slot 0.
namespace {

struct CopyList {                      // merges copies whose source AND destination continue the previous one
  using C = HostCopy;
  std::vector<C> v;
  const std::vector<std::pair<const char*, size_t>>* blocks = nullptr;   // rr_host_alloc blocks of the context
  bool host_is_src = true;
  // both host addresses inside ONE rr_host_alloc block: the bytes between them are the caller's own padding
  bool same_block(const void* a, const void* b) const {
    if (!blocks) return false;
    for (const auto& blk : *blocks) {
      const char *lo = blk.first, *hi = blk.first + blk.second;
      if ((const char*)a >= lo && (const char*)a < hi) return (const char*)b >= lo && (const char*)b < hi;
    }
    return false;
  }
  void add(void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    // (a frame adds one piece per array: the piece this one continues is a few entries back)
    for (size_t back = 1; back <= v.size() && back <= 12; back++) {
      C& p = v[v.size() - back];
      const ptrdiff_t gd = (char*)dst - ((char*)p.dst + p.bytes), gs = (const char*)src - ((const char*)p.src + p.bytes);
      // contiguous on both sides -- or separated on both sides by the same few bytes of alignment padding inside one
      // page-locked block of the caller and (always) inside the library's own staging: one copy, padding included
      if (gd == gs && gd >= 0 && (gd == 0 || (gd < 16 && same_block(host_is_src ? p.src : p.dst, host_is_src ? src : (const void*)dst)))) {
        p.bytes += (size_t)gd + bytes;
        return;
      }
    }
    v.push_back(C{dst, src, bytes});
  }
};

// Is this host address device-accessible (page-locked: rr_host_alloc / hipHostMalloc / hipHostRegister)?
bool host_is_pinned(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();                              // pageable memory: "invalid value", not an error of ours
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// The copies of one direction of one slot.  Pieces whose host side is pinned and whose ends are 16-byte aligned (or that
// are tiny) go into ONE launch of k_copy_pieces; the rest -- pageable or oddly aligned buffers -- one hipMemcpyAsync each.
int issue(rr_ctx* ctx, const CopyList& cl, hipMemcpyKind kind, hipStream_t s, CopyPiece*& d_list, CopyPiece*& h_list, size_t& cap) {
  std::vector<CopyPiece> ker;
  const bool up = kind == hipMemcpyHostToDevice;
  const void* last_base = nullptr;
  bool last_pinned = false;
  for (const auto& c : cl.v) {
    const void* host = up ? c.src : c.dst;
    bool use_kernel = false;
    if (ctx->copy_kernels) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(c.src) | reinterpret_cast<uintptr_t>(c.dst);
      if ((a & 15) == 0 || c.bytes <= 4096) {
        // (consecutive pieces usually come from one allocation: ask the runtime once per 2 MB neighbourhood)
        const void* base = reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(host) >> 21);
        if (base != last_base) {
          last_pinned = host_is_pinned(host);
          last_base = base;
        }
        use_kernel = last_pinned && host_is_pinned(static_cast<const char*>(host) + c.bytes - 1) ;
      }
    }
    if (use_kernel) ker.push_back(CopyPiece{c.src, c.dst, (uint64_t)c.bytes});
    else HIPCHK(hipMemcpyAsync(c.dst, c.src, c.bytes, kind, s));
  }
  if (!ker.empty()) {
    if (ker.size() > cap) {
      HIPCHK(hipStreamSynchronize(s));
      if (d_list) HIPCHK(hipFree(d_list));
      if (h_list) HIPCHK(hipHostFree(h_list));
      d_list = h_list = nullptr;
      cap = 0;
      const size_t want = ker.size() * 2 + 64;
      HIPCHK(hipMalloc((void**)&d_list, want * sizeof(CopyPiece)));
      HIPCHK(hipHostMalloc((void**)&h_list, want * sizeof(CopyPiece), hipHostMallocDefault));
      cap = want;
    }
    // (the slot is not in flight: its previous lists have been consumed)
    memcpy(h_list, ker.data(), ker.size() * sizeof(CopyPiece));
    HIPCHK(hipMemcpyAsync(d_list, h_list, ker.size() * sizeof(CopyPiece), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_copy_pieces, dim3(128), dim3(256), 0, s, d_list, (int)ker.size());
  }
  return RR_OK;
}

// Per-frame strides of the staging arrays, padded so that every frame's piece starts on a 16-byte boundary (the copy
// kernels move 16 bytes per lane): bytes for the uint8 arrays, elements for the others.
struct Strides {
  size_t px3b, ex3b, pngb;      // uint8: image, environment map, PNG scanlines
  size_t pxd, px3d, exd, ex3d;  // float64 (also the float32 / float64 depth slot): H*W, H*W*3, He*We, He*We*3 elements, even
  size_t pxi;                   // int32: H*W elements, a multiple of 4
};
Strides strides_of(const Dims& dm) {
  const size_t px = (size_t)dm.H * dm.W, ex = (size_t)dm.He * dm.We;
  Strides t;
  t.px3b = (px * 3 + 15) & ~(size_t)15;
  t.ex3b = (ex * 3 + 15) & ~(size_t)15;
  t.pngb = ((size_t)dm.H * (1 + 4 * (size_t)dm.W) + 15) & ~(size_t)15;
  t.pxd = (px + 1) & ~(size_t)1;
  t.px3d = (px * 3 + 1) & ~(size_t)1;
  t.exd = (ex + 1) & ~(size_t)1;
  t.ex3d = (ex * 3 + 1) & ~(size_t)1;
  t.pxi = (px + 3) & ~(size_t)3;
  return t;
}

int slot_reserve(rr_ctx* ctx, rr_ctx::Staging& st, int n, int max_drops, const Dims& dm) {
  max_drops = (max_drops + 3) & ~3;                      // int32 per-drop arrays: every frame's piece 16-byte aligned
  if (n > st.frames || max_drops > st.drops_cap || dm.H != st.dims.H || dm.W != st.dims.W || dm.He != st.dims.He || dm.We != st.dims.We) {
    HIPCHK(hipDeviceSynchronize());
    const Strides t = strides_of(dm);
    int F = n > st.frames ? n : st.frames, D = max_drops > st.drops_cap ? max_drops : st.drops_cap, rc;
    if ((rc = dev_alloc(ctx, st.bg, F * t.px3d))) return rc;
    if ((rc = dev_alloc(ctx, st.rainy, F * t.px3d))) return rc;
    if ((rc = dev_alloc(ctx, st.env, F * t.ex3d))) return rc;
    if ((rc = dev_alloc(ctx, st.omega, F * t.exd))) return rc;
    if ((rc = dev_alloc(ctx, st.comp, F * t.px3d))) return rc;
    if ((rc = dev_alloc(ctx, st.mask, F * t.pxd))) return rc;
    if ((rc = dev_alloc(ctx, st.drops, (size_t)F * D))) return rc;
    if ((rc = dev_alloc(ctx, st.rgb, F * t.px3b))) return rc;
    if ((rc = dev_alloc(ctx, st.mask_i, F * t.pxi))) return rc;
    if ((rc = dev_alloc(ctx, st.status, (size_t)F * D))) return rc;
    if ((rc = dev_alloc(ctx, st.colour, (size_t)F * D * 3))) return rc;
    if ((rc = dev_alloc(ctx, st.ndrops, (size_t)((F + 3) & ~3)))) return rc;
    if ((rc = dev_alloc(ctx, st.depth, F * t.pxd))) return rc;
    if ((rc = dev_alloc(ctx, st.env_u8, F * t.ex3b))) return rc;
    if ((rc = dev_alloc(ctx, st.bg8, F * t.px3b))) return rc;
    if ((rc = dev_alloc(ctx, st.png_i, F * t.pngb))) return rc;
    if ((rc = dev_alloc(ctx, st.png_m, F * t.pngb))) return rc;
    st.frames = F;
    st.drops_cap = D;
    st.dims = dm;
  }
  return RR_OK;
}

int slot_init(rr_ctx* ctx, rr_ctx::Slot& sl) {
  if (sl.ev_up) return RR_OK;
  if (!ctx->s_up) {
    HIPCHK(hipStreamCreateWithFlags(&ctx->s_up, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->s_down, hipStreamNonBlocking));
  }
  HIPCHK(hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&sl.ev_comp, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&sl.ev_down, hipEventDisableTiming));
  HIPCHK(hipHostMalloc((void**)&sl.h_flags, sizeof(int64_t) * 2, hipHostMallocDefault));
  return RR_OK;
}

// Validate the WHOLE batch before anything is copied or mutated (sizes, pointers, per-frame dimensions).
int validate_host_batch(rr_ctx* ctx, int n, const rr_prepass_in* pre, const rr_frame_in* in, const rr_frame_out* out,
                        const rr_prepass_out* pre_out, Dims& dm, int& max_drops) {
  if (n <= 0 || (!pre && !in) || (in && !out) || (pre && !in && !pre_out)) {
    ctx->err = "bad frame batch";
    return RR_E_ARG;
  }
  if (in) dm = Dims{in[0].H, in[0].W, in[0].He, in[0].We};
  else {
    bool want_env = false;
    for (int f = 0; f < n; f++) want_env = want_env || pre_out[f].env_xyY || pre_out[f].env_bgr_u8;
    dm = Dims{pre[0].H, pre[0].W, pre[0].H, (want_env && ctx->have_eg) ? ctx->eg.We : 1};
  }
  if (dm.H <= 0 || dm.W <= 0 || dm.He <= 0 || dm.We <= 0) {
    ctx->err = "bad frame size";
    return RR_E_ARG;
  }
  if (pre && in)
    for (int f = 0; f < n; f++)
      if (in[f].ext) {
        ctx->err = "rr_ext_tile is taken by rr_render_frames / rr_render_frames_device only";
        return RR_E_ARG;
      }
  if (pre && in && (!ctx->have_eg || dm.He != dm.H || dm.We != ctx->eg.We)) {
    ctx->err = "pipeline: He/We must be H / rr_envmap_width() of the geometry set for this frame size";
    return RR_E_ARG;
  }
  max_drops = 1;
  for (int f = 0; f < n; f++) {
    if (pre) {
      if (pre[f].H != dm.H || pre[f].W != dm.W) {
        ctx->err = "all frames of a batch must share H,W";
        return RR_E_ARG;
      }
      const bool env_only = pre[f].mode == RR_PRE_ENV_ONLY;
      if (env_only && in) {
        ctx->err = "RR_PRE_ENV_ONLY is for rr_prepass_frames only";
        return RR_E_ARG;
      }
      if ((!pre[f].bg && !pre[f].bg_u8) || (!env_only && (!pre[f].depth || !(pre[f].irr_den != 0.0)))) {
        ctx->err = "null pre-pass pointer or zero irradiance denominator";
        return RR_E_ARG;
      }
      const int bgk = pre[f].in_types & (RR_IN_BG_F32 | RR_IN_BG_U8 | RR_IN_BG_PNG_ROWS);
      if ((pre[f].in_types & ~(RR_IN_BG_F32 | RR_IN_BG_U8 | RR_IN_BG_PNG_ROWS)) || (bgk & (bgk - 1)) ||
          (pre_out && ((pre_out[f].out_types & ~(RR_OUT_RAINY_F32 | RR_OUT_ENV_F32)) || pre_out[f].out_types != pre_out[0].out_types))) {
        ctx->err = "pre-pass: in_types is ONE of RR_IN_BG_F32 / RR_IN_BG_U8 / RR_IN_BG_PNG_ROWS (or 0), out_types RR_OUT_* bits, the same for every frame";
        return RR_E_ARG;
      }
      // files handed over as filtered scanlines (rr_io_read_frames_rows): for every frame of the batch or for none (one launch
      // of the un-filter kernel per kind of file), and not in the map-only mode (its image is any float image)
      if (((pre[f].in_types ^ pre[0].in_types) & RR_IN_BG_PNG_ROWS) || ((pre[f].depth_f64 == RR_DEPTH_PNG_ROWS) != (pre[0].depth_f64 == RR_DEPTH_PNG_ROWS)) ||
          (env_only && ((pre[f].in_types & RR_IN_BG_PNG_ROWS) || pre[f].depth_f64 == RR_DEPTH_PNG_ROWS)) ||
          ((pre[f].in_types & RR_IN_BG_PNG_ROWS) && pre[f].bg_u8) || pre[f].depth_f64 < 0 || pre[f].depth_f64 > RR_DEPTH_PNG_ROWS) {
        ctx->err = "pre-pass: PNG scanlines (RR_IN_BG_PNG_ROWS / RR_DEPTH_PNG_ROWS) for every frame of a batch or for none, not with bg_u8 or RR_PRE_ENV_ONLY";
        return RR_E_ARG;
      }
      if (!in && !env_only && !pre_out[f].rainy_bg) {
        ctx->err = "pre-pass: null rainy_bg output";
        return RR_E_ARG;
      }
    }
    if (in) {
      if (in[f].H != dm.H || in[f].W != dm.W || in[f].He != dm.He || in[f].We != dm.We) {
        ctx->err = "all frames of a batch must share H,W,He,We";
        return RR_E_ARG;
      }
      if (in[f].n_drops_dev) {
        ctx->err = "n_drops_dev is a device pointer: rr_render_frames_device only";
        return RR_E_ARG;
      }
      if ((in[f].sim != nullptr) != (in[0].sim != nullptr) || (in[f].sim && in[f].ext)) {
        ctx->err = "generated drop tables (rr_frame_in.sim): for every frame of a batch or for none, and without rr_ext_tile";
        return RR_E_ARG;
      }
      const int cap_f = in[f].sim ? (in[f].n_drops > 0 ? in[f].n_drops : in[f].sim->n_particles) : in[f].n_drops;
      if (cap_f < 0 || cap_f > 65536) {
        ctx->err = "n_drops outside [0, 2^16] (generator.py:425)";
        return RR_E_ARG;
      }
      if (in[f].strategy != 0 && in[f].strategy != 1) {
        ctx->err = "rendering strategy must be 0 (default) or 1 ('white'); 'naive_db' is broken in the reference (bad_weather.py:355)";
        return RR_E_ARG;
      }
      if (!in[f].omega && !(ctx->d_omega && ctx->omega_He == dm.He && ctx->omega_We == dm.We)) {
        ctx->err = "omega == NULL needs rr_set_solid_angles for this map size";
        return RR_E_STATE;
      }
      if ((!pre && (!in[f].bg || !in[f].rainy_bg || !in[f].env_xyY)) || (!in[f].sim && in[f].n_drops > 0 && !in[f].drops) ||
          (!out[f].rainy_rgb && !out[f].rainy_png)) {
        ctx->err = "null frame pointer (an image output is needed: rainy_rgb or rainy_png)";
        return RR_E_ARG;
      }
      if (out[f].mask_png && !ctx->have_lut) {
        ctx->err = "mask_png needs rr_set_colormap";
        return RR_E_STATE;
      }
      if (cap_f > max_drops) max_drops = cap_f;
    }
  }
  if (in && (!ctx->have_cam || !ctx->have_db)) {
    ctx->err = "streak DB and camera must be set before rendering";
    return RR_E_STATE;
  }
  if (pre && !ctx->have_pk) {
    ctx->err = "rr_set_prepass_kernels must be called before the pre-pass";
    return RR_E_STATE;
  }
  return RR_OK;
}

int host_submit(rr_ctx* ctx, int slot, int32_t n, const rr_prepass_in* pre, const rr_frame_in* in, const rr_frame_out* out,
                const rr_prepass_out* pre_out) {
  if (slot < 0 || slot >= RR_PIPE_SLOTS) {
    ctx->err = "bad pipeline slot";
    return RR_E_ARG;
  }
  rr_ctx::Slot& sl = ctx->slots[slot];
  if (sl.busy) {
    ctx->err = "pipeline slot still in flight: call rr_pipeline_wait first";
    return RR_E_STATE;
  }
  HIPCHK(hipSetDevice(ctx->device));
  Dims dm{0, 0, 0, 0};
  int max_drops = 1, rc;
  if ((rc = validate_host_batch(ctx, n, pre, in, out, pre_out, dm, max_drops))) return rc;
  if ((rc = slot_init(ctx, sl))) return rc;
  for (void* b : sl.ext_blobs) hipFree(b);             // the slot's previous batch is complete (not busy)
  sl.ext_blobs.clear();
  auto& st = sl.st;
  // Staging strides that follow the caller's layout where that lets a whole array travel as one copy: drop tables laid
  // out at a constant distance (frames back to back in one block) get that distance on the device too
  int drop_stride = 0;
  if (in && n > 1 && in[0].drops && !in[0].sim) {
    const ptrdiff_t hs = (const char*)in[1].drops - (const char*)in[0].drops;
    bool regular = hs > 0 && hs % (ptrdiff_t)sizeof(rr_drop) == 0 && hs / (ptrdiff_t)sizeof(rr_drop) >= max_drops &&
                   hs / (ptrdiff_t)sizeof(rr_drop) <= 65536 + 16 && (hs / (ptrdiff_t)sizeof(rr_drop)) % 4 == 0;
    for (int f = 2; regular && f < n; f++) regular = (const char*)in[f].drops - (const char*)in[f - 1].drops == hs;
    if (regular) drop_stride = (int)(hs / (ptrdiff_t)sizeof(rr_drop));
  }
  if ((rc = slot_reserve(ctx, st, n, drop_stride ? drop_stride : max_drops, dm))) return rc;
  if (!drop_stride) drop_stride = st.drops_cap;
  bool all_f32_depth = pre != nullptr || in != nullptr;     // (no float64 map in the batch: uint16 samples fit the float32 slots too)
  for (int f = 0; f < n; f++) all_f32_depth = all_f32_depth && (pre ? pre[f].depth_f64 : in[f].depth_f64) != 1;      // (RR_DEPTH_PNG_ROWS becomes uint16 samples)
  auto depth_el = [](int kind) { return kind == 1 ? (size_t)8 : (kind == RR_DEPTH_U16 ? (size_t)2 : (size_t)4); };
  const size_t px = (size_t)dm.H * dm.W, ex = (size_t)dm.He * dm.We, png_bytes = (size_t)dm.H * (1 + 4 * (size_t)dm.W);
  const Strides T = strides_of(dm);
  // the depth slot: float32 maps of a batch lie 4 * H * W bytes (rounded to 16) apart, float64 ones (or a mix) 8 * H * W
  const size_t depth_stride = all_f32_depth ? (((size_t)dm.H * dm.W * 4 + 15) & ~(size_t)15) : T.pxd * 8;
  auto depth_at = [&](int f) { return reinterpret_cast<double*>(reinterpret_cast<char*>(st.depth) + (size_t)f * depth_stride); };
  hipStream_t s = ctx->stream;
  std::vector<rr_frame_in> din(in ? n : 0);
  std::vector<rr_frame_out> dout(in ? n : 0);
  std::vector<rr_prepass_in> pin(pre ? n : 0);
  std::vector<rr_prepass_out> pout(pre ? n : 0);
  CopyList up, down;
  std::vector<std::pair<const char*, size_t>> host_blocks;     // (a snapshot: another thread may be page-locking further blocks)
  {
    std::lock_guard<std::mutex> lk(ctx->host_mu);
    host_blocks = ctx->host_allocs;
  }
  up.blocks = down.blocks = &host_blocks;
  up.host_is_src = true;
  down.host_is_src = false;
  std::vector<rr_sim_frame> sims;
  // Width of the two arrays the pre-pass hands on (the staging is sized for float64 either way).  Where pre_out downloads
  // one, its out_types decide; an array that never leaves the device is float32 in the pipeline (RR_OPT_PIPELINE_F32; the
  // map only with the resident solid angles, which exist in both widths) and float64 in a pre-pass-only call.
  int pre_types = 0;
  if (pre) {
    const int asked = pre_out ? pre_out[0].out_types : 0;
    bool down_rainy = false, down_env = false, resident_omega = in != nullptr;
    for (int f = 0; f < n; f++) {
      down_rainy = down_rainy || (pre_out && pre_out[f].rainy_bg);
      down_env = down_env || (pre_out && pre_out[f].env_xyY);
      resident_omega = resident_omega && !in[f].omega;
    }
    const bool narrow = in && ctx->pipe_f32;
    if (down_rainy ? (asked & RR_OUT_RAINY_F32) != 0 : narrow) pre_types |= RR_OUT_RAINY_F32;
    if (down_env ? (asked & RR_OUT_ENV_F32) != 0 : (narrow && resident_omega)) pre_types |= RR_OUT_ENV_F32;
    if (in && (pre_types & RR_OUT_ENV_F32) && !resident_omega) {
      ctx->err = "pipeline: a float32 xyY map (RR_OUT_ENV_F32) needs the resident solid angles (omega == NULL in every frame)";
      return RR_E_ARG;
    }
  }
  const size_t rainy_el = (pre_types & RR_OUT_RAINY_F32) ? 4 : 8, penv_el = (pre_types & RR_OUT_ENV_F32) ? 4 : 8;
  // ---- upload ----
  for (int f = 0; pre && f < n; f++) {
    pin[f] = pre[f];
    // the image in the caller's element type: bytes stay bytes (bg = bytes / 255.0 is formed where a kernel reads it)
    const void* src = pre[f].bg_u8 ? (const void*)pre[f].bg_u8 : pre[f].bg;
    const bool rows_i = (pre[f].in_types & RR_IN_BG_PNG_ROWS) != 0, rows_d = pre[f].depth_f64 == RR_DEPTH_PNG_ROWS;
    pin[f].in_types = (pre[f].bg_u8 || rows_i) ? RR_IN_BG_U8 : pre[f].in_types;
    const size_t bg_el = (pin[f].in_types & RR_IN_BG_U8) ? 1 : ((pin[f].in_types & RR_IN_BG_F32) ? 4 : 8);
    pin[f].bg = bg_el == 1 ? (const void*)(st.bg8 + f * T.px3b) : (const void*)(st.bg + f * T.px3d);
    pin[f].bg_u8 = nullptr;
    pin[f].depth = depth_at(f);
    // a file's filtered scanlines (k_png_unfilter makes the image bytes / the depth samples of them before the pre-pass) wait in
    // the frame's fog-layer / composite slots: free until the pre-pass / the compositor write them
    if (rows_i) up.add(st.rainy + f * T.px3d, src, (size_t)dm.H * (1 + 3 * (size_t)dm.W));
    else up.add(const_cast<void*>(pin[f].bg), src, px * 3 * bg_el);
    const bool env_only = pre[f].mode == RR_PRE_ENV_ONLY;
    if (rows_d) {
      up.add(st.comp + f * T.px3d, pre[f].depth, (size_t)dm.H * (1 + 2 * (size_t)dm.W));
      pin[f].depth_f64 = RR_DEPTH_U16;
    } else if (!env_only) {
      up.add((void*)pin[f].depth, pre[f].depth, px * depth_el(pre[f].depth_f64));
    }
    pout[f].out_types = pre_types;
    pout[f].reserved = 0;
    pout[f].rainy_bg = st.rainy + f * T.px3d;
    const bool env = in || pre_out[f].env_xyY || pre_out[f].env_bgr_u8;
    pout[f].env_xyY = env ? st.env + f * T.ex3d : nullptr;
    pout[f].env_bgr_u8 = (pre_out && pre_out[f].env_bgr_u8) ? st.env_u8 + f * T.ex3b : nullptr;
  }
  for (int f = 0; in && f < n; f++) {
    din[f] = in[f];
    din[f].bg = pre ? pin[f].bg : (const void*)(st.bg + f * T.px3d);
    din[f].rainy_bg = st.rainy + f * T.px3d;
    din[f].env_xyY = st.env + f * T.ex3d;
    // the solid-angle map depends on the map size only: NULL = the resident one (rr_set_solid_angles); frames that pass
    // the same host array share one upload
    const bool same_omega = !in[f].omega || (f > 0 && in[f].omega == in[0].omega);
    din[f].omega = !in[f].omega ? nullptr : (same_omega ? din[0].omega : st.omega + f * T.exd);
    din[f].drops = st.drops + (size_t)f * drop_stride;
    if (pre) {                        // the pre-pass' depth buffer doubles as the occlusion depth
      if (pre[f].depth_f64 >= RR_DEPTH_U16 && ctx->depth_occlusion) {
        ctx->err = "RR_OPT_DEPTH_OCCLUSION needs a float depth map (not RR_DEPTH_U16 / RR_DEPTH_PNG_ROWS)";
        return RR_E_ARG;
      }
      din[f].depth = pre[f].depth_f64 >= RR_DEPTH_U16 ? nullptr : depth_at(f);
      din[f].depth_f64 = pre[f].depth_f64 == 1 ? 1 : 0;
    } else if (in[f].depth && ctx->depth_occlusion) {
      up.add((void*)depth_at(f), in[f].depth, px * (in[f].depth_f64 ? 8 : 4));
      din[f].depth = depth_at(f);
    } else {
      din[f].depth = nullptr;
    }
    // element sizes of the caller's arrays (rr_frame_in.in_types; the staging is sized for float64)
    // (behind the pre-pass: the image as the caller gave it to the pre-pass, the fog layer and the map at the hand-over width)
    const int ty = !pre ? in[f].in_types
                        : (pin[f].in_types | ((pre_types & RR_OUT_RAINY_F32) ? RR_IN_RAINY_F32 : 0) | ((pre_types & RR_OUT_ENV_F32) ? RR_IN_ENV_F32 : 0));
    din[f].in_types = ty;
    const size_t bg_el = (ty & RR_IN_BG_U8) ? 1 : ((ty & RR_IN_BG_F32) ? 4 : 8);
    const bool shared_bg = in[f].rainy_bg == in[f].bg;         // one array for both: one upload, and the kernels see one pointer
    const size_t rainy_in_el = shared_bg ? bg_el : ((ty & RR_IN_RAINY_U8) ? 1 : ((ty & RR_IN_RAINY_F32) ? 4 : 8));
    const size_t env_el = (ty & RR_IN_ENV_F32) ? 4 : 8;
    if (!pre) {
      up.add((void*)din[f].bg, in[f].bg, px * 3 * bg_el);
      if (shared_bg) din[f].rainy_bg = din[f].bg;
      else up.add((void*)din[f].rainy_bg, in[f].rainy_bg, px * 3 * rainy_in_el);
      up.add((void*)din[f].env_xyY, in[f].env_xyY, ex * 3 * env_el);
    }
    if (!same_omega) up.add((void*)din[f].omega, in[f].omega, ex * env_el);
    din[f].sim = nullptr;
    if (in[f].sim) {                  // the drop table is generated on the device (below): nothing to upload
      sims.push_back(*in[f].sim);
      din[f].n_drops = in[f].n_drops > 0 ? in[f].n_drops : in[f].sim->n_particles;
      din[f].n_drops_dev = st.ndrops + f;
    } else {
      up.add((void*)din[f].drops, in[f].drops, sizeof(rr_drop) * (size_t)in[f].n_drops);
    }
    dout[f].rainy_rgb = st.rgb + f * T.px3b;
    dout[f].rainy_bg_out = out[f].rainy_bg_out ? st.comp + f * T.px3d : nullptr;
    dout[f].mask_f64 = st.mask + f * T.pxd;
    dout[f].mask_i32 = out[f].mask_i32 ? st.mask_i + f * T.pxi : nullptr;
    dout[f].drop_status = st.status + (size_t)f * drop_stride;
    dout[f].drop_colour = out[f].drop_colour ? st.colour + (size_t)f * drop_stride * 3 : nullptr;
    din[f].ext = nullptr;
    if (in[f].ext && in[f].n_drops > 0) {
      // caller-made tiles (the single-drop seam; not a throughput path): one device blob per frame =
      // [rr_ext_tile x n_drops | alphas | polygons], pointers rewritten to the device copies
      const int nd = in[f].n_drops;
      size_t bytes = sizeof(rr_ext_tile) * (size_t)nd;
      for (int k = 0; k < nd; k++) {
        const rr_ext_tile& e = in[f].ext[k];
        if (!e.alpha) continue;
        if (e.tw <= 0 || e.th <= 0 || e.n_poly < 0 || e.n_poly > POLY_STRIDE || (e.n_poly > 0 && !e.poly_xy)) {
          ctx->err = "bad rr_ext_tile";
          return RR_E_ARG;
        }
        bytes += sizeof(double) * ((size_t)e.tw * e.th + 2 * (size_t)e.n_poly);
      }
      char* blob = nullptr;
      HIPCHK(hipMalloc((void**)&blob, bytes));
      sl.ext_blobs.push_back(blob);
      std::vector<rr_ext_tile> dev(in[f].ext, in[f].ext + nd);
      size_t off = sizeof(rr_ext_tile) * (size_t)nd;
      for (int k = 0; k < nd; k++) {
        rr_ext_tile& e = dev[k];
        if (!e.alpha) continue;
        const size_t na = sizeof(double) * (size_t)e.tw * e.th, np = sizeof(double) * 2 * (size_t)e.n_poly;
        HIPCHK(hipMemcpy(blob + off, e.alpha, na, hipMemcpyHostToDevice));
        e.alpha = reinterpret_cast<const double*>(blob + off);
        off += na;
        if (np) HIPCHK(hipMemcpy(blob + off, e.poly_xy, np, hipMemcpyHostToDevice));
        e.poly_xy = reinterpret_cast<const double*>(blob + off);
        off += np;
      }
      HIPCHK(hipMemcpy(blob, dev.data(), sizeof(rr_ext_tile) * (size_t)nd, hipMemcpyHostToDevice));
      din[f].ext = reinterpret_cast<const rr_ext_tile*>(blob);
    }
    dout[f].rainy_png = out[f].rainy_png ? st.png_i + (size_t)f * T.pngb : nullptr;
    dout[f].mask_png = out[f].mask_png ? st.png_m + (size_t)f * T.pngb : nullptr;
  }
  // from here on copies that read the caller's buffers are queued: a failing call drains them before it returns (the
  // caller is free to release its buffers after an error)
  auto drained = [&](int code) {
    (void)hipStreamSynchronize(ctx->s_up);
    (void)hipStreamSynchronize(s);
    return code;
  };
  if ((rc = issue(ctx, up, hipMemcpyHostToDevice, ctx->s_up, sl.d_up, sl.h_up, sl.cap_up))) return drained(rc);
  if (hipEventRecord(sl.ev_up, ctx->s_up) != hipSuccess || hipStreamWaitEvent(s, sl.ev_up, 0) != hipSuccess) {
    ctx->err = "pipeline: hipEventRecord / hipStreamWaitEvent failed";
    return drained(RR_E_HIP);
  }
  // ---- compute ----
  if (pre && ((pre[0].in_types & RR_IN_BG_PNG_ROWS) || pre[0].depth_f64 == RR_DEPTH_PNG_ROWS)) {
    ProfScope ps(ctx, s, "k_png_unfilter");
    if (pre[0].in_types & RR_IN_BG_PNG_ROWS)
      hipLaunchKernelGGL(k_png_unfilter<3>, dim3(n), dim3(64), 3 * (size_t)dm.W, s, reinterpret_cast<const uint8_t*>(st.rainy), (int64_t)(T.px3d * 8),
                         st.bg8, (int64_t)T.px3b, dm.H, dm.W);
    if (pre[0].depth_f64 == RR_DEPTH_PNG_ROWS)
      hipLaunchKernelGGL(k_png_unfilter<2>, dim3(n), dim3(64), 2 * (size_t)dm.W, s, reinterpret_cast<const uint8_t*>(st.comp), (int64_t)(T.px3d * 8),
                         reinterpret_cast<uint8_t*>(st.depth), (int64_t)depth_stride, dm.H, dm.W);
  }
  if (pre && (rc = enqueue_prepass(ctx, n, pin.data(), pout.data(), s))) return drained(rc);
  if (!sims.empty() && (rc = enqueue_particles(ctx, n, sims.data(), dm.H, dm.W, st.drops, drop_stride, st.ndrops, s))) return drained(rc);
  if (in) {                           // the slot's own overflow flag, cleared in stream order before the batch's k_scan may set it
    hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, s, ctx->sc.overflow + 1 + slot, 0);
    if ((rc = enqueue(ctx, n, din.data(), dout.data(), s, 1 + slot))) return drained(rc);
  }
  if (hipEventRecord(sl.ev_comp, s) != hipSuccess) {
    ctx->err = "pipeline: hipEventRecord failed";
    return drained(RR_E_HIP);
  }
  // ---- download: only the LIST is made here.  The copies are issued by rr_pipeline_wait once the kernels have finished:
  // a device-to-host copy queued now would sit in the DMA queue waiting for its kernels -- and hold up the NEXT batch's
  // upload queued behind it (measured: the GPU then idles for the length of an upload between two batches).
  for (int f = 0; in && f < n; f++) {
    if (out[f].rainy_rgb) down.add(out[f].rainy_rgb, dout[f].rainy_rgb, px * 3);
    if (out[f].rainy_bg_out) down.add(out[f].rainy_bg_out, dout[f].rainy_bg_out, px * 3 * sizeof(double));
    if (out[f].mask_f64) down.add(out[f].mask_f64, dout[f].mask_f64, px * sizeof(double));
    if (out[f].mask_i32) down.add(out[f].mask_i32, dout[f].mask_i32, px * sizeof(int32_t));
    if (out[f].drop_status) down.add(out[f].drop_status, dout[f].drop_status, sizeof(int32_t) * (size_t)din[f].n_drops);
    if (out[f].drop_colour) down.add(out[f].drop_colour, dout[f].drop_colour, sizeof(double) * 3 * (size_t)din[f].n_drops);
    if (out[f].n_drops_out && in[f].sim) down.add(out[f].n_drops_out, st.ndrops + f, sizeof(int32_t));
    if (out[f].rainy_png) down.add(out[f].rainy_png, dout[f].rainy_png, png_bytes);
    if (out[f].mask_png) down.add(out[f].mask_png, dout[f].mask_png, png_bytes);
  }
  for (int f = 0; pre && pre_out && f < n; f++) {
    if (pre_out[f].rainy_bg && pre[f].mode != RR_PRE_ENV_ONLY) down.add(pre_out[f].rainy_bg, pout[f].rainy_bg, px * 3 * rainy_el);
    if (pre_out[f].env_xyY) down.add(pre_out[f].env_xyY, pout[f].env_xyY, ex * 3 * penv_el);
    if (pre_out[f].env_bgr_u8) down.add(pre_out[f].env_bgr_u8, pout[f].env_bgr_u8, ex * 3);
  }
  sl.down = std::move(down.v);
  sl.h_flags[0] = 0;
  sl.busy = true;
  sl.rendered = in != nullptr;
  return RR_OK;
}

// RR_OK, or RR_E_ARENA after growing the tile arena (the batch's outputs are invalid: submit it again)
int host_wait(rr_ctx* ctx, int slot) {
  if (slot < 0 || slot >= RR_PIPE_SLOTS) {
    ctx->err = "bad pipeline slot";
    return RR_E_ARG;
  }
  rr_ctx::Slot& sl = ctx->slots[slot];
  if (!sl.busy) return RR_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipEventSynchronize(sl.ev_comp));             // the batch's kernels are done (later batches keep the GPU busy meanwhile)
  {
    CopyList down;
    down.v = std::move(sl.down);
    sl.down.clear();
    int rc = issue(ctx, down, hipMemcpyDeviceToHost, ctx->s_down, sl.d_down, sl.h_down, sl.cap_down);
    if (rc) {                                            // (nothing may keep writing the caller's buffers after an error)
      (void)hipStreamSynchronize(ctx->s_down);
      sl.busy = false;
      return rc;
    }
    if (sl.rendered) HIPCHK(hipMemcpyAsync(&sl.h_flags[0], ctx->sc.overflow + 1 + slot, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->s_down));
    HIPCHK(hipStreamSynchronize(ctx->s_down));
  }
  sl.busy = false;
  if (sl.rendered && (int32_t)sl.h_flags[0] != 0) {
    // THIS batch ran against a short arena (its own flag, read behind its own kernels).  Other batches in flight report
    // through their own flags; the arena grows once for all of them (the sticky maximum covers every batch whose k_scan
    // has run: grow_arena waits for the device first).
    int rc = grow_arena(ctx);
    if (rc) return rc;
    ctx->err = "tile arena overflow: arena regrown, submit the batch again";
    return RR_E_ARENA;
  }
  return RR_OK;
}

int run_host(rr_ctx* ctx, int32_t n, const rr_prepass_in* pre, const rr_frame_in* in, const rr_frame_out* out,
             const rr_prepass_out* pre_out) {
  for (int attempt = 0;; attempt++) {
    int rc = host_submit(ctx, 0, n, pre, in, out, pre_out);
    if (rc) return rc;
    rc = host_wait(ctx, 0);
    if (rc != RR_E_ARENA || attempt == 2) return rc;
  }
}

}  // namespace

extern "C" {

int rr_render_frames(rr_ctx* ctx, int32_t n, const rr_frame_in* in, const rr_frame_out* out) {
  if (!ctx) return RR_E_ARG;
  if (!in || !out) {
    ctx->err = "bad frame batch";
    return RR_E_ARG;
  }
  return run_host(ctx, n, nullptr, in, out, nullptr);
}

int rr_prepass_frames(rr_ctx* ctx, int32_t n, const rr_prepass_in* in, const rr_prepass_out* out) {
  if (!ctx) return RR_E_ARG;
  if (!in || !out) {
    ctx->err = "bad pre-pass batch";
    return RR_E_ARG;
  }
  return run_host(ctx, n, in, nullptr, nullptr, out);
}

int rr_pipeline_frames(rr_ctx* ctx, int32_t n, const rr_prepass_in* pre, const rr_frame_in* in, const rr_frame_out* out,
                       const rr_prepass_out* pre_out) {
  if (!ctx) return RR_E_ARG;
  if (!pre || !in || !out) {
    ctx->err = "bad pipeline batch";
    return RR_E_ARG;
  }
  return run_host(ctx, n, pre, in, out, pre_out);
}

int rr_pipeline_submit(rr_ctx* ctx, int32_t slot, int32_t n, const rr_prepass_in* pre, const rr_frame_in* in, const rr_frame_out* out,
                       const rr_prepass_out* pre_out) {
  if (!ctx) return RR_E_ARG;
  if (!in || !out) {
    ctx->err = "bad pipeline batch";
    return RR_E_ARG;
  }
  return host_submit(ctx, slot, n, pre, in, out, pre_out);
}

int rr_pipeline_wait(rr_ctx* ctx, int32_t slot) {
  if (!ctx) return RR_E_ARG;
  return host_wait(ctx, slot);
}

int rr_host_alloc(rr_ctx* ctx, void** out, int64_t bytes) {
  if (!ctx) return RR_E_ARG;
  if (!out || bytes <= 0) {
    ctx->err = "rr_host_alloc: bad argument";
    return RR_E_ARG;
  }
  *out = nullptr;
  {
    hipError_t e = hipSetDevice(ctx->device);                  // (no ctx->err from here on: this may be a second thread)
    if (e == hipSuccess) e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      std::lock_guard<std::mutex> lk(ctx->host_mu);            // (the message under the allocator's lock, not in ctx->err: see above)
      ctx->host_err = std::string("rr_host_alloc: hipHostMalloc of ") + std::to_string((long long)bytes) + " bytes: " + hipGetErrorString(e);
      return RR_E_HIP;
    }
  }
  std::lock_guard<std::mutex> lk(ctx->host_mu);
  ctx->host_err.clear();
  ctx->host_allocs.emplace_back((const char*)*out, (size_t)bytes);
  return RR_OK;
}

const char* rr_host_last_error(rr_ctx* ctx) {
  if (!ctx) return "null context";
  std::lock_guard<std::mutex> lk(ctx->host_mu);
  return ctx->host_err.c_str();
}

int rr_host_free(rr_ctx* ctx, void* p) {
  if (!ctx) return RR_E_ARG;
  if (p) {
    {
      std::lock_guard<std::mutex> lk(ctx->host_mu);
      for (size_t k = 0; k < ctx->host_allocs.size(); k++)
        if (ctx->host_allocs[k].first == (const char*)p) {
          ctx->host_allocs.erase(ctx->host_allocs.begin() + (ptrdiff_t)k);
          break;
        }
    }
    HIPCHK(hipHostFree(p));
  }
  return RR_OK;
}

int rr_set_solid_angles(rr_ctx* ctx, int32_t He, int32_t We, const double* omega) {
  if (!ctx) return RR_E_ARG;
  if (He <= 0 || We <= 0 || !omega) {
    ctx->err = "rr_set_solid_angles: bad argument";
    return RR_E_ARG;
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  int rc = dev_alloc(ctx, ctx->d_omega, (size_t)He * We);
  if (rc) return rc;
  HIPCHK(hipMemcpy(ctx->d_omega, omega, sizeof(double) * (size_t)He * We, hipMemcpyHostToDevice));
  {
    std::vector<float> of((size_t)He * We);
    for (size_t i = 0; i < of.size(); i++) of[i] = (float)omega[i];
    if ((rc = dev_alloc(ctx, ctx->d_omega32, of.size()))) return rc;
    HIPCHK(hipMemcpy(ctx->d_omega32, of.data(), sizeof(float) * of.size(), hipMemcpyHostToDevice));
  }
  ctx->omega_He = He;
  ctx->omega_We = We;
  return RR_OK;
}

int rr_prepass_frames_device(rr_ctx* ctx, int32_t n, const rr_prepass_in* in, const rr_prepass_out* out, void* stream) {
  if (!ctx) return RR_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  return enqueue_prepass(ctx, n, in, out, stream ? (hipStream_t)stream : ctx->stream);
}

int rr_set_prepass_kernels(rr_ctx* ctx, const rr_prepass_kernels* k) {
  if (!ctx) return RR_E_ARG;
  if (!k || k->fog_ksize < 1 || k->env_ksize < 1 || k->fog_ksize > RR_MAX_TAPS || k->env_ksize > RR_MAX_TAPS ||
      !(k->fog_ksize & 1) || !(k->env_ksize & 1)) {
    ctx->err = "rr_set_prepass_kernels: tap counts must be odd and <= RR_MAX_TAPS";
    return RR_E_ARG;
  }
  static_assert(RR_MAX_TAPS == rrpre::KMAX, "tap capacity");
  // scipy.ndimage.correlate1d (the oracle's convolution) folds symmetric kernels; anything else is refused --
  // BEFORE the context is touched
  for (int i = 0; i < k->fog_ksize / 2; i++)
    if (k->fog_w[i] != k->fog_w[k->fog_ksize - 1 - i]) {
      ctx->err = "rr_set_prepass_kernels: fog kernel is not symmetric";
      return RR_E_ARG;
    }
  for (int i = 0; i < k->env_ksize / 2; i++)
    if (k->env_w[i] != k->env_w[k->env_ksize - 1 - i]) {
      ctx->err = "rr_set_prepass_kernels: envmap kernel is not symmetric";
      return RR_E_ARG;
    }
  ctx->pk.fog_k = k->fog_ksize;
  ctx->pk.env_k = k->env_ksize;
  for (int i = 0; i < RR_MAX_TAPS; i++) {
    ctx->pk.fog_w[i] = i < k->fog_ksize ? k->fog_w[i] : 0.0;
    ctx->pk.env_w[i] = i < k->env_ksize ? k->env_w[i] : 0.0;
  }
  ctx->have_pk = true;
  return RR_OK;
}

int rr_set_envmap_geometry(rr_ctx* ctx, int32_t H, int32_t W, int32_t cw, int32_t n_uniq, const int32_t* uniq, const int32_t* first) {
  if (!ctx) return RR_E_ARG;
  if (H <= 0 || W <= 0 || cw <= 0 || n_uniq < 0 || (n_uniq > 0 && (!uniq || !first)) || (int64_t)H * cw > INT32_MAX) {
    ctx->err = "rr_set_envmap_geometry: bad argument";
    return RR_E_ARG;
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  std::vector<int32_t> src((size_t)H * cw), top(cw), bot(cw);
  if (!rrpre::build_env_tables(H, W, cw, n_uniq, uniq, first, src.data(), top.data(), bot.data())) {
    ctx->err = "rr_set_envmap_geometry: cell or source pixel outside the frame";
    return RR_E_ARG;
  }
  int rc;
  if ((rc = dev_alloc(ctx, ctx->d_esrc, src.size()))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_etop, top.size()))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_ebot, bot.size()))) return rc;
  HIPCHK(hipMemcpy(ctx->d_esrc, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_etop, top.data(), top.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_ebot, bot.data(), bot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  ctx->eg.H = H;
  ctx->eg.W = W;
  ctx->eg.cw = cw;
  ctx->eg.lw = cw / 2;
  ctx->eg.We = cw + 2 * (cw / 2);
  ctx->eg.src = ctx->d_esrc;
  ctx->eg.top_row = ctx->d_etop;
  ctx->eg.bot_row = ctx->d_ebot;
  ctx->eg.need_h = nullptr;
  ctx->eg_src_host = std::move(src);
  ctx->need_half = -1;                 // the map of needed horizontal sums is made with the first batch (it depends on the tap count)
  ctx->have_eg = true;
  return RR_OK;
}

int rr_envmap_width(rr_ctx* ctx) {
  if (!ctx) return RR_E_ARG;
  if (!ctx->have_eg) {
    ctx->err = "no environment-map geometry set";
    return RR_E_STATE;
  }
  return ctx->eg.We;
}

int rr_set_colormap(rr_ctx* ctx, const uint8_t* lut_rgba) {
  if (!ctx) return RR_E_ARG;
  if (!lut_rgba) {
    ctx->err = "rr_set_colormap: null table";
    return RR_E_ARG;
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->d_lut) HIPCHK(hipMalloc((void**)&ctx->d_lut, 1024));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(ctx->d_lut, lut_rgba, 1024, hipMemcpyHostToDevice));
  ctx->have_lut = true;
  return RR_OK;
}

int rr_set_option(rr_ctx* ctx, int32_t option, int32_t value) {
  if (!ctx) return RR_E_ARG;
  switch (option) {
    case RR_OPT_DEDUP: ctx->dedup = value != 0; return RR_OK;
    case RR_OPT_GENERAL_FOV: ctx->general_fov = value != 0; return RR_OK;
    case RR_OPT_DEPTH_OCCLUSION: ctx->depth_occlusion = value != 0; return RR_OK;
    case RR_OPT_COMPOSITE_F64: ctx->composite_f64 = value != 0; return RR_OK;
    case RR_OPT_COPY_KERNELS: ctx->copy_kernels = value != 0; return RR_OK;
    case RR_OPT_PADDED_TEXTURES: ctx->padded_tex = value != 0; return RR_OK;
    case RR_OPT_TILE_ROWS: ctx->tile_rows = value < 0 ? 0 : (value > 2 ? 2 : value); return RR_OK;
    case RR_OPT_ROWS_SHARES: ctx->rows_shares = value < 1 ? 1 : (value > 8 ? 8 : value); return RR_OK;
    case RR_OPT_FOV_FILL_RULE: ctx->fill_rule = value == 1 ? 1 : 0; return RR_OK;
    case RR_OPT_BIN_ROWS: ctx->bin_rows = value != 0; return RR_OK;
    case RR_OPT_COMPOSITE_U16: ctx->composite_u16 = value != 0; return RR_OK;
    case RR_OPT_BLUR_DMA:
#ifdef RR_EXPERIMENTS
      ctx->blur_dma = value != 0;
      return RR_OK;
#else
      if (value == 0) break;                                  // the register-staged kernel is only in -DRR_EXPERIMENTS builds
      return RR_OK;
#endif
    case RR_OPT_FOV_DDA: ctx->fov_dda = value != 0 ? 1 : 0; return RR_OK;       // (2 was k_fov_walk, r05: measured, no faster, removed in r06)
    case RR_OPT_PIPELINE_F32: ctx->pipe_f32 = value != 0; return RR_OK;
    case RR_OPT_WILD_PIXELS: ctx->wild_pixels = value != 0; return RR_OK;
    case RR_OPT_PNG_DEFLATE: ctx->png_deflate = value != 0; return RR_OK;
    case RR_OPT_COMPOSITE_WAVES:
      if (value != 0 && (value < 4 || value > 8)) break;
      ctx->comp_waves = value;
      return RR_OK;
    case RR_OPT_COMPOSITE_BATCH: ctx->comp_batch = value != 0; return RR_OK;
    case RR_OPT_COLOUR_STREAM:
      if (value < 0 || value > 2) break;
      ctx->colour_stream = value ? 1 : 0;                   // (2: r05's other split, measured slower and removed in r06 -- runs as 1)
      return RR_OK;
    case RR_OPT_FOV_F32:
      if (value < 0 || value > 2) break;
      ctx->fov_f32 = value;
      return RR_OK;
    case RR_OPT_FOV_THREADS:
      if (value != 0 && value != 512 && value != 1024) break;
      ctx->fov_threads = value;
      return RR_OK;
    case RR_OPT_BLUR_WORKGROUPS:
      if (value != 0 && value != 3 && value != 4 && value != 5) break;
      ctx->blur_wg = value;
      return RR_OK;
    case RR_OPT_FOV_DROPS_PER_THREAD:
      if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) break;
      ctx->fov_dpt = value;
      return RR_OK;
    default: break;
  }
  ctx->err = "rr_set_option: unknown option or value";
  return RR_E_ARG;
}

int rr_batch_counts(rr_ctx* ctx, int32_t frame, int32_t out[8]) {
  if (!ctx || !out || frame < 0 || frame >= ctx->last_n || !ctx->sc.counts) return RR_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, ctx->sc.counts + (size_t)frame * 8, sizeof(int32_t) * 8, hipMemcpyDeviceToHost));
  int32_t rows = 0;                    // the row-walk kernel's share of the rotate + resize tiles (k_lists writes it with or without the option)
  int32_t rows2[2] = {0, 0};
  HIPCHK(hipMemcpy(rows2, ctx->sc.rows_n + 2 * (size_t)frame, sizeof(int32_t) * 2, hipMemcpyDeviceToHost));
  rows = rows2[0];
  out[0] += rows;
  out[5] += rows2[1];                  // Big tiles rendered by k_tile_rows
  return RR_OK;
}

int rr_profile_enable(rr_ctx* ctx, int32_t on) {
  if (!ctx) return RR_E_ARG;
  ctx->prof = on != 0;
  return RR_OK;
}

int rr_profile_reset(rr_ctx* ctx) {
  if (!ctx) return RR_E_ARG;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  prof_collect(ctx);
  ctx->prof_stats.clear();
  return RR_OK;
}

int rr_profile_read(rr_ctx* ctx, rr_kernel_stat* out, int32_t cap) {
  if (!ctx || !out || cap < 0) return RR_E_ARG;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  prof_collect(ctx);
  int n = (int)ctx->prof_stats.size();
  if (n > cap) n = cap;
  for (int i = 0; i < n; i++) out[i] = ctx->prof_stats[i];
  return n;
}

}  // extern "C"

#ifdef RR_PHASES
// (phase-clock builds only; not part of include/rainhip.h) out[64] = g_phase, then cleared when `reset`
extern "C" int rr_debug_phases(rr_ctx* ctx, unsigned long long* out, int reset) {
  if (!ctx || !out) return RR_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 64));
  if (reset) {
    unsigned long long z[64] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
  }
  return RR_OK;
}
#endif
