// rr_device.h -- per-drop and per-pixel arithmetic of the rain-streak hot path.
//
// Every function here is `__host__ __device__`: the HIP kernels in rainhip.hip call
// them on gfx950, and tests/hostemu compiles the very same functions with g++ so the
// arithmetic can be unit-tested against the numpy oracle without a GPU (the product
// never runs the host instantiation).
//
// Arithmetic contract (DESIGN.md "bit-exactness"): IEEE double with the evaluation
// order spelled out below, no FMA contraction (-ffp-contract=off and the pragma
// below), no device transcendental on any value that reaches an alpha tile.  Device
// libm (sqrt/atan2) is used ONLY for the field-of-view polygon, which feeds the
// per-drop colour constant (tolerance +-1 LSB on rainy_image).
#pragma once
#include <math.h>
#include <stdint.h>

#include "rainhip.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RR_HD __host__ __device__ inline
#else
#define RR_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace rr {

enum { KIND_BIG = 0, KIND_ROT = 1, KIND_EXT = 2 };   // EXT: the tile and the FOV polygon come from the caller (rr_ext_tile)
enum { RS_AREA = 0, RS_AREA_FAST = 1, RS_LINEAR = 2 };

struct Dims {
  int32_t H, W, He, We;
};

// Everything a kernel needs to know about one drop.  Written by plan_drop (one thread
// per drop), arena offsets filled in by the per-frame scan.
struct DropPlan {
  int32_t status;            // RR_DROP_*
  int32_t kind;              // KIND_BIG | KIND_ROT
  int32_t tex;               // texture index
  int32_t flip;              // cv2.flip(drop, 0) (generator.py:165)
  int32_t tw, th;            // raw tile (before the defocus pad)
  int32_t shift;             // int(10*c) (bad_weather.py:293)
  int32_t pw, ph;            // padded tile
  int32_t r1, r2;            // gaussian radii along axis 0 (rows) / axis 1 (cols)
  int32_t vis_x0, vis_y0, vis_w, vis_h;   // footprint inside the frame
  int32_t crop_x, crop_y;    // padded-tile coordinates of the footprint origin
  int32_t ew;                // effective blurred tile width  tw + 2*r2 (== tw when not blurred)
  int32_t bw0;               // warpPerspective block width
  int32_t nW, nH;            // rotate_bound canvas
  int32_t rs_mode;           // RS_*
  int32_t isx, isy;          // integer scales (RS_AREA_FAST)
  int32_t eh;                // effective blurred tile height th + 2*r1
  int32_t epitch, epad;      // storage of the finished effective tile: row pitch and the column of its first pixel
  int64_t a0_off, a1_off;    // arena offsets in doubles: raw tile (tw x th) / finished effective tile (ew x eh, blurred drops only)
  double sig1, sig2;         // c, c/2 (bad_weather.py:291)
  double tau_one, g;         // exposure*length_opacity, tau_one/tau_zero (bad_weather.py:425-427,443)
  double mi[9];              // inverse homography (Big)
  double ma[6];              // inverse rotate_bound affine (non-Big)
  double scale_x, scale_y, inv_sx, inv_sy;   // cv2.resize scales
};

// What the compositor reads per (screen tile, drop): 88 bytes.
struct CompRec {
  int32_t x0, y0, x1, y1;    // footprint, exclusive upper bounds; empty if x1<=x0
  int32_t ox, oy;            // tile_x = px + ox, tile_y = py + oy
  int32_t pitch, pad;
  int64_t off;               // arena offset of the finished tile
  double tau_one, g;
  double K[3];               // BGR colour constants
  double zdist;              // distance of the drop from the camera, |world z| (depth-occlusion option only)
};

// The same record for the float-colour compositor (k_composite32): 64 bytes, of which the kernel reads the first 48 as three
// 16-byte words.  The tile address is folded: sample (px, py) of the frame is arena[base + py * pitch + px]
// (base = off + oy * pitch + ox of CompRec); the colour factors are pre-multiplied and rounded to float:
//   te = tau_one / exposure, kg[c] = K[c] * g   (blend: c' = clamp((1 - A te) c + A kg))
struct CompRec32 {
  uint32_t xx;               // x0 | x1 << 16  (footprint, exclusive upper bounds; empty if x1 <= x0)
  uint32_t yy;               // y0 | y1 << 16
  int64_t base;              // arena index of the sample under frame pixel (0, 0)
  uint32_t pitch_slow;       // pitch | slow << 31; slow: a factor is not tame (or the tile is the caller's): the literal double blend
  float te, kg[3];
  int32_t spare0;
  double zdist;
  int32_t spare[4];
};
static_assert(sizeof(CompRec32) == 64, "CompRec32 is read as 16-byte words");

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
RR_HD int64_t cv_round(double v) {   // saturate_cast<int>(double): rint + saturate
  if (!(v == v)) return -2147483648LL;
  if (v < -2147483648.0) return -2147483648LL;
  if (v > 2147483647.0) return 2147483647LL;
  return (int64_t)rint(v);
}
RR_HD int64_t sat_short(int64_t v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
RR_HD int32_t imin(int32_t a, int32_t b) { return b < a ? b : a; }
RR_HD int32_t imax(int32_t a, int32_t b) { return b > a ? b : a; }
RR_HD double dmin(double a, double b) { return b < a ? b : a; }   // Python / std::min semantics
RR_HD double dmax(double a, double b) { return b > a ? b : a; }
RR_HD double clip01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }   // np.clip keeps NaN
RR_HD int32_t iabs(int32_t a) { return a < 0 ? -a : a; }

// exp(x), -700 < x <= 0, from + - * / only (oracle/render.py det_exp).
RR_HD double det_exp(double x) {
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double INV_LN2 = 1.44269504088896338700e+00;
  double k = rint(x * INV_LN2);
  double r = (x - k * LN2_HI) - k * LN2_LO;
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 1.0 / 2.0;
  p = p * r + 1.0;
  p = p * r + 1.0;
  return ldexp(p, (int)k);
}

// un-normalised gaussian weight phi(i) for tap distance i (scipy _gaussian_kernel1d)
RR_HD double gauss_phi(double sigma, int i) {
  double sigma2 = sigma * sigma;
  return det_exp(-0.5 / sigma2 * (double)(i * i));
}

// interpolateCubic(i/32) in float, A = -0.75 (OpenCV imgwarp.cpp)
inline void build_cubic_tab(float* tab /*32*4*/) {
  const float A = -0.75f;
  const float scale = 1.0f / 32.0f;
  for (int i = 0; i < 32; i++) {
    volatile float x = (float)i * scale;
    volatile float xp = x + 1.0f;
    volatile float t0 = A * xp;
    t0 = t0 - 5.0f * A;
    t0 = t0 * xp;
    t0 = t0 + 8.0f * A;
    t0 = t0 * xp;
    t0 = t0 - 4.0f * A;
    volatile float t1 = (A + 2.0f) * x;
    t1 = t1 - (A + 3.0f);
    t1 = t1 * x;
    t1 = t1 * x;
    t1 = t1 + 1.0f;
    volatile float xm = 1.0f - x;
    volatile float t2 = (A + 2.0f) * xm;
    t2 = t2 - (A + 3.0f);
    t2 = t2 * xm;
    t2 = t2 * xm;
    t2 = t2 + 1.0f;
    volatile float t3 = 1.0f - t0;
    t3 = t3 - t1;
    t3 = t3 - t2;
    tab[i * 4 + 0] = t0;
    tab[i * 4 + 1] = t1;
    tab[i * 4 + 2] = t2;
    tab[i * 4 + 3] = t3;
  }
}

// Texture accessors: streaks_light[idx] / 255.0 (bad_weather.py:252) with BORDER_CONSTANT 0.
// TexGlobal divides on the fly; TexLut reads the same quotients from a 256-entry table
// (identical bits), which the tile kernel keeps in LDS next to the texels.
struct TexGlobal {
  const uint8_t* t;
  int h, w;
  RR_HD double at(int64_t y, int64_t x) const { return (double)t[y * w + x] / 255.0; }
  RR_HD void at4(int64_t y, int64_t x, double v[4]) const { v[0] = at(y, x); v[1] = at(y, x + 1); v[2] = at(y, x + 2); v[3] = at(y, x + 3); }
  RR_HD double tap(int64_t y, int64_t x) const {
    if (y < 0 || y >= h || x < 0 || x >= w) return 0.0;
    return at(y, x);
  }
};
struct TexLut {
  const uint8_t* t;
  const double* lut;
  int h, w;
  RR_HD double at(int64_t y, int64_t x) const { return lut[t[y * w + x]]; }
  RR_HD void at4(int64_t y, int64_t x, double v[4]) const { v[0] = at(y, x); v[1] = at(y, x + 1); v[2] = at(y, x + 2); v[3] = at(y, x + 3); }
  RR_HD double tap(int64_t y, int64_t x) const {
    if (y < 0 || y >= h || x < 0 || x >= w) return 0.0;
    return at(y, x);
  }
};

// LDS copy with a 2-texel zero border (pitch w+4): lets the hot sampling loop clamp
// coordinates to [-2, w] x [-2, h] instead of testing every tap.
struct TexLutPad {
  const uint8_t* t;
  const double* lut;
  int h, w;
  RR_HD double at(int64_t y, int64_t x) const { return lut[t[(y + 2) * (w + 4) + (x + 2)]]; }
  RR_HD void at4(int64_t y, int64_t x, double v[4]) const { v[0] = at(y, x); v[1] = at(y, x + 1); v[2] = at(y, x + 2); v[3] = at(y, x + 3); }
  RR_HD double tap(int64_t y, int64_t x) const {
    if (y < 0 || y >= h || x < 0 || x >= w) return 0.0;
    return at(y, x);
  }
};

// ---------------------------------------------------------------------------
// Big drops: cv2.warpPerspective(INTER_CUBIC)  (generator.py:126-132)
// ---------------------------------------------------------------------------
// source coordinates of output pixel (x, y): top-left tap (sx, sy) of the 4x4 bicubic window and the 1/32 fractions.
// bx: the first column of the pixel's block, (x / bw0) * bw0 (WarpPerspectiveInvoker walks blocks of bw0 columns and
// forms X0, Y0, W0 at a block's first column)
struct BigCoord {
  int sx, sy, fx, fy;
};
RR_HD BigCoord warp_big_coord(const DropPlan& p, int bx, int x, int y) {
  const double* Mi = p.mi;
  double x1 = (double)(x - bx);
  double bxf = (double)bx, yf = (double)y;
  double X0 = Mi[0] * bxf + Mi[1] * yf + Mi[2];
  double Y0 = Mi[3] * bxf + Mi[4] * yf + Mi[5];
  double W0 = Mi[6] * bxf + Mi[7] * yf + Mi[8];
  double W = W0 + Mi[6] * x1;
  W = (W != 0.0) ? 32.0 / W : 0.0;
  double fX = dmax(-2147483648.0, dmin(2147483647.0, (X0 + Mi[0] * x1) * W));
  double fY = dmax(-2147483648.0, dmin(2147483647.0, (Y0 + Mi[3] * x1) * W));
  // (fX, fY are inside [-2^31, 2^31 - 1] and never NaN -- dmin / dmax return their first argument for one -- so cv_round's
  //  saturate_cast<int> is the rounding alone and fits an int: no 64-bit conversion on the device)
  const int X = (int)rint(fX), Y = (int)rint(fY);
  const int qx = X >> 5, qy = Y >> 5;
  return BigCoord{(qx < -32768 ? -32768 : (qx > 32767 ? 32767 : qx)) - 1, (qy < -32768 ? -32768 : (qy > 32767 ? 32767 : qy)) - 1, X & 31, Y & 31};
}
template <class Tex>
RR_HD double warp_big_pixel(const DropPlan& p, const Tex& tx, const float* ctab, int x, int y) {
  const int sh = tx.h, sw = tx.w;
  const BigCoord c = warp_big_coord(p, (x / p.bw0) * p.bw0, x, y);
  const int64_t sx = c.sx, sy = c.sy;
  const int fx = c.fx, fy = c.fy;
  const float* cx = ctab + fx * 4;
  const float* cy = ctab + fy * 4;
  int width1 = imax(sw - 3, 0), height1 = imax(sh - 3, 0);
  bool interior = sx >= 0 && sx < width1 && sy >= 0 && sy < height1;
  double sum = 0.0;
  if (interior) {
    for (int i = 0; i < 4; i++) {
      float w0 = cy[i] * cx[0], w1 = cy[i] * cx[1], w2 = cy[i] * cx[2], w3 = cy[i] * cx[3];
      double v[4];
      tx.at4(sy + i, sx, v);                                   // four neighbours of one texture row
      double r = (v[0] * (double)w0 + v[1] * (double)w1) + v[2] * (double)w2;
      r = r + v[3] * (double)w3;
      sum = (i == 0) ? r : sum + r;
    }
  } else {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        float w = cy[i] * cx[j];
        sum = sum + tx.tap(sy + i, sx + j) * (double)w;
      }
  }
  return clip01(sum);
}

// ---------------------------------------------------------------------------
// Medium/Small drops: imutils.rotate_bound -> cv2.flip -> cv2.resize(INTER_AREA)
// (generator.py:163-170)
// ---------------------------------------------------------------------------
// warpAffine(INTER_LINEAR) sample given the fixed-point row/column terms
// (X0, Y0 include round_delta; adelta/bdelta are the per-column increments)
template <class Tex>
RR_HD double rot_sample(const Tex& tx, int64_t X0, int64_t Y0, int64_t adelta, int64_t bdelta) {
  int64_t X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  int64_t sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
  int fx = (int)(X & 31), fy = (int)(Y & 31);
  double w00 = (double)((32 - fy) * (32 - fx)) / 1024.0;
  double w01 = (double)((32 - fy) * fx) / 1024.0;
  double w10 = (double)(fy * (32 - fx)) / 1024.0;
  double w11 = (double)(fy * fx) / 1024.0;
  double v00, v01, v10, v11;
  if (sx >= 0 && sx + 1 < tx.w && sy >= 0 && sy + 1 < tx.h) {
    v00 = tx.at(sy, sx); v01 = tx.at(sy, sx + 1); v10 = tx.at(sy + 1, sx); v11 = tx.at(sy + 1, sx + 1);
  } else {
    v00 = tx.tap(sy, sx); v01 = tx.tap(sy, sx + 1); v10 = tx.tap(sy + 1, sx); v11 = tx.tap(sy + 1, sx + 1);
  }
  return ((v00 * w00 + v01 * w01) + v10 * w10) + v11 * w11;
}
RR_HD int64_t rot_adelta(const DropPlan& p, int rx) { return cv_round(p.ma[0] * (double)rx * 1024.0); }
RR_HD int64_t rot_bdelta(const DropPlan& p, int rx) { return cv_round(p.ma[3] * (double)rx * 1024.0); }
RR_HD int64_t rot_X0(const DropPlan& p, int ry) { return cv_round((p.ma[1] * (double)ry + p.ma[2]) * 1024.0) + 16; }
RR_HD int64_t rot_Y0(const DropPlan& p, int ry) { return cv_round((p.ma[4] * (double)ry + p.ma[5]) * 1024.0) + 16; }

// one pixel of warpAffine(INTER_LINEAR) of the texture into the nW x nH canvas
template <class Tex>
RR_HD double rot_pixel(const DropPlan& p, const Tex& tx, int ry, int rx) {
  return rot_sample(tx, rot_X0(p, ry), rot_Y0(p, ry), rot_adelta(p, rx), rot_bdelta(p, rx));
}

// the image cv2.resize reads: rotated canvas, vertically flipped if p.flip
template <class Tex>
RR_HD double canvas_pixel(const DropPlan& p, const Tex& tx, int cy, int cx) {
  return rot_pixel(p, tx, p.flip ? (p.nH - 1 - cy) : cy, cx);
}

// computeResizeAreaTab for one destination index
struct AreaSpan {
  int16_t s1, s2;            // full cells s1 .. s2-1
  int16_t has_l, has_r;      // partial cells at s1-1 and s2
  float a_l, a_m, a_r;
};
RR_HD AreaSpan area_span(int ssize, double scale, int d) {
  AreaSpan a;
  double fsx1 = (double)d * scale;
  double fsx2 = fsx1 + scale;
  double cell = dmin(scale, (double)ssize - fsx1);
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = imin(sx2, ssize - 1);
  sx1 = imin(sx1, sx2);
  a.s1 = (int16_t)sx1;
  a.s2 = (int16_t)sx2;
  a.has_l = ((double)sx1 - fsx1 > 1e-3) ? 1 : 0;
  a.a_l = (float)(((double)sx1 - fsx1) / cell);
  a.a_m = (float)(1.0 / cell);
  a.has_r = (fsx2 - (double)sx2 > 1e-3) ? 1 : 0;
  a.a_r = (float)(dmin(dmin(fsx2 - (double)sx2, 1.0), cell) / cell);
  return a;
}

// horizontal pass of resizeArea_ for one source row and one destination column
template <class Tex>
RR_HD double area_hsum(const DropPlan& p, const Tex& tx, const AreaSpan& ax, int sy) {
  double b = 0.0;
  if (ax.has_l) b = b + canvas_pixel(p, tx, sy, ax.s1 - 1) * (double)ax.a_l;
  for (int sx = ax.s1; sx < ax.s2; sx++) b = b + canvas_pixel(p, tx, sy, sx) * (double)ax.a_m;
  if (ax.has_r) b = b + canvas_pixel(p, tx, sy, ax.s2) * (double)ax.a_r;
  return b;
}

// the area_mode branch of cv::resize's INTER_LINEAR coefficient set-up
RR_HD void lin_coord(int ssize, double scale, double inv_scale, int d, bool is_x, int& s, float& f, bool& tail) {
  s = (int)floor((double)d * scale);
  f = (float)((double)(d + 1) - (double)(s + 1) * inv_scale);
  f = (f <= 0.0f) ? 0.0f : f - floorf(f);
  tail = false;
  if (is_x) {
    if (s < 0) { f = 0.0f; s = 0; }
    if (s + 1 >= ssize) {
      tail = true;
      if (s >= ssize - 1) { f = 0.0f; s = ssize - 1; }
    }
  }
}

template <class Tex>
RR_HD double lin_hrow(const DropPlan& p, const Tex& tx, int row, int s, float f, bool tail) {
  if (tail) return canvas_pixel(p, tx, row, s) * 1.0;
  float a0 = 1.0f - f;
  return canvas_pixel(p, tx, row, s) * (double)a0 + canvas_pixel(p, tx, row, s + 1) * (double)f;
}

template <class Tex>
RR_HD double resize_pixel(const DropPlan& p, const Tex& tx, int dx, int dy) {
  double v;
  if (p.rs_mode == RS_AREA_FAST) {
    int area = p.isx * p.isy;
    float scale = 1.0f / (float)area;
    int by = dy * p.isy, bx = dx * p.isx;
    double s = 0.0;
    int k = 0;
    for (; k <= area - 4; k += 4) {
      double q[4];
      for (int t = 0; t < 4; t++) {
        int kk = k + t;
        q[t] = canvas_pixel(p, tx, by + kk / p.isx, bx + kk % p.isx);
      }
      s = s + (((q[0] + q[1]) + q[2]) + q[3]);
    }
    for (; k < area; k++) s = s + canvas_pixel(p, tx, by + k / p.isx, bx + k % p.isx);
    v = s * (double)scale;
  } else if (p.rs_mode == RS_AREA) {
    AreaSpan ax = area_span(p.nW, p.scale_x, dx);
    AreaSpan ay = area_span(p.nH, p.scale_y, dy);
    double acc = 0.0;
    bool first = true;
    if (ay.has_l) {
      acc = (double)ay.a_l * area_hsum(p, tx, ax, ay.s1 - 1);
      first = false;
    }
    for (int sy = ay.s1; sy < ay.s2; sy++) {
      double t = (double)ay.a_m * area_hsum(p, tx, ax, sy);
      acc = first ? t : acc + t;
      first = false;
    }
    if (ay.has_r) {
      double t = (double)ay.a_r * area_hsum(p, tx, ax, ay.s2);
      acc = first ? t : acc + t;
    }
    v = acc;
  } else {
    int sx, sy;
    float fx, fy;
    bool tlx, tly;
    lin_coord(p.nW, p.scale_x, p.inv_sx, dx, true, sx, fx, tlx);
    lin_coord(p.nH, p.scale_y, p.inv_sy, dy, false, sy, fy, tly);
    int r0 = imin(imax(sy, 0), p.nH - 1), r1 = imin(imax(sy + 1, 0), p.nH - 1);
    float b0 = 1.0f - fy;
    v = lin_hrow(p, tx, r0, sx, fx, tlx) * (double)b0 + lin_hrow(p, tx, r1, sx, fx, tlx) * (double)fy;
  }
  return clip01(v);
}

// raw (un-blurred) alpha of tile pixel (x, y) in raw-tile coordinates
template <class Tex>
RR_HD double raw_tile_pixel(const DropPlan& p, const Tex& tx, const float* ctab, int x, int y) {
  return p.kind == KIND_BIG ? warp_big_pixel(p, tx, ctab, x, y) : resize_pixel(p, tx, x, y);
}

// ---------------------------------------------------------------------------
// The same tile by ROW WALKS (k_tile_rows, round 6): one lane owns one canvas row and walks the columns that can touch
// the texture from left to right.  The horizontal folds of resizeArea_ then run in a register -- a column's sample is
// added to the sum of the destination cell it belongs to, with that cell's weight, in exactly area_hsum's order -- and
// only the finished cell sums (one double per row and destination column) pass through LDS for the vertical fold.
//
// A column table says, per canvas column, what a walk does with the sample s it took there:
//     b = b + s * |w1|;  if (sign bit of w1) { cell sum is complete: store b, next cell; b = s * w2; }
//   w1   the column's weight in the cell it is a full or right-partial column of (a_m / a_r); a_l when the column is
//        ONLY the left partial of a cell (no cell owns it: 0.0 + s * a_l == s * a_l, the value area_hsum starts from);
//        0 when no cell reads the column.  Sign bit: the column is the last one its cell reads.
//   w2   a_l of the NEXT cell when the column is also that cell's left partial, else 0 (then b restarts at +0.0).
// Requires scale_x >= 2 (every cell then has a full or right-partial column to carry the sign bit).
// s is the bilinear sample BEFORE its division by 1024 and the table holds weight * 2^-10 (exact in float): the factor
// is a power of two, so s * (w * 2^-10) and (s * 2^-10) * w round the same exact product to the same double.
// ---------------------------------------------------------------------------
struct alignas(16) ColEnt {
  int32_t ad, bd;            // rot_adelta / rot_bdelta of the column
  uint32_t w1, w2;           // float bits, see above
};
constexpr float COL_W_SCALE = 1.0f / 1024.0f;
RR_HD uint32_t f32_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
RR_HD float bits_f32(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

// pass 1, one call per destination column d (after every entry was set to {ad, bd, 0, 0} and cell[] to 0): the columns
// cell d owns.  `cell[x]` = the destination column a walk that STARTS at x is in.
RR_HD void coltab_cell_pass1(const DropPlan& p, int d, ColEnt* col, uint8_t* cell, uint16_t* cfirst, uint16_t* clast) {
  const AreaSpan A = area_span(p.nW, p.scale_x, d);
  cfirst[d] = (uint16_t)((A.has_l && A.s1 >= 1) ? A.s1 - 1 : A.s1);       // first / last canvas column the cell reads
  clast[d] = (uint16_t)(A.has_r ? A.s2 : A.s2 - 1);
  const uint32_t am = f32_bits(A.a_m * COL_W_SCALE);
  for (int x = A.s1; x < A.s2; x++) { col[x].w1 = am; cell[x] = (uint8_t)d; }
  int R = A.s2 - 1;
  if (A.has_r) { col[A.s2].w1 = f32_bits(A.a_r * COL_W_SCALE); cell[A.s2] = (uint8_t)d; R = A.s2; }
  if (R >= 0) col[R].w1 |= 0x80000000u;
}
// pass 2 (after pass 1 of EVERY cell): the left partial column of cell d -- shared with the end of cell d - 1, or nobody's
RR_HD void coltab_cell_pass2(const DropPlan& p, int d, ColEnt* col, uint8_t* cell) {
  const AreaSpan A = area_span(p.nW, p.scale_x, d);
  if (!A.has_l || A.s1 < 1) return;
  const int c = A.s1 - 1;
  if (col[c].w1 & 0x80000000u) col[c].w2 = f32_bits(A.a_l * COL_W_SCALE);
  else { col[c].w1 = f32_bits(A.a_l * COL_W_SCALE); cell[c] = (uint8_t)d; }
}
// k_tile_rows takes the canvas rows of a tile in PASSES of R consecutive rows, every row cut into S segments of whole
// destination cells: R * S <= 64 lanes, a lane per (row, segment).  The cell sums of a pass (R x tw doubles) and the
// accumulators of the one destination row whose rows straddle the end of the pass (tw doubles) share `buf` doubles.
RR_HD void rows_pass_shape(int tw, int buf, int& R, int& S) {
  const int r_max = imax(imin(64, buf / tw - 1), 1);
  S = (64 + r_max - 1) / r_max;
  R = 64 / S;
}
// the cells [dA, dB) lane segment j of S takes of a row that touches cells dlo .. dhi
RR_HD void rows_segment(int dlo, int dhi, int S, int j, int& dA, int& dB) {
  const int cps = (int)(((float)(dhi - dlo + S) + 0.5f) / (float)S);     // ceil((dhi - dlo + 1) / S): small integers, exact in float
  dA = dlo + j * cps;
  dB = imin(dA + cps, dhi + 1);
}
// canvas rows destination row dy reads: left partial, full rows, right partial (area_span of the vertical axis)
RR_HD void vfold_rows(const AreaSpan& a, int& first_row, int& last_row) {
  first_row = a.has_l ? a.s1 - 1 : a.s1;
  last_row = a.has_r ? a.s2 : a.s2 - 1;
}
// The part of a destination pixel's vertical fold that lies in canvas rows [R0, R1], in resize_pixel's order: continues
// `acc` (first: no term yet).  get(row) = the horizontal cell sum of that canvas row.
template <class Get>
RR_HD void vfold_part(const AreaSpan& ay, int R0, int R1, double& acc, bool& first, Get&& get) {
  if (ay.has_l && ay.s1 - 1 >= R0 && ay.s1 - 1 <= R1) {
    acc = (double)ay.a_l * get(ay.s1 - 1);
    first = false;
  }
  const int m1 = imin(ay.s2 - 1, R1);
  for (int sy = imax(ay.s1, R0); sy <= m1; sy++) {
    const double v = (double)ay.a_m * get(sy);
    acc = first ? v : acc + v;
    first = false;
  }
  if (ay.has_r && ay.s2 >= R0 && ay.s2 <= R1) {
    const double v = (double)ay.a_r * get(ay.s2);
    acc = first ? v : acc + v;
    first = false;
  }
}
// pair texture: u16 element (y, x) = texel (y, x) | texel (y + 1, x) << 8 for y = -2 .. sh, x = -2 .. pitch - 3 (zeros
// outside the texture): the 2 x 2 neighbourhood of a bilinear sample is two adjacent elements.  pitch / 2 is odd, so
// lanes on consecutive texture rows read different LDS banks.
RR_HD int pair_pitch(int sw) { int q = sw + 4; while ((q & 3) != 2) q++; return q; }
RR_HD int64_t pair_bytes(int sh, int sw) { return ((int64_t)(sh + 3) * pair_pitch(sw) * 2 + 15) & ~15LL; }

// ---------------------------------------------------------------------------
// field-of-view polygon   (bad_weather.py:596-704)
// ---------------------------------------------------------------------------
RR_HD void rotmat(const double a[3], double c, double s, double R[9]) {   // bad_weather.py:532-538
  double omc = 1.0 - c;
  R[0] = c + omc * (a[0] * a[0]);
  R[1] = s * (-a[2]) + omc * (a[0] * a[1]);
  R[2] = s * (a[1]) + omc * (a[0] * a[2]);
  R[3] = s * (a[2]) + omc * (a[1] * a[0]);
  R[4] = c + omc * (a[1] * a[1]);
  R[5] = s * (-a[0]) + omc * (a[1] * a[2]);
  R[6] = s * (-a[1]) + omc * (a[2] * a[0]);
  R[7] = s * (a[0]) + omc * (a[2] * a[1]);
  R[8] = c + omc * (a[2] * a[2]);
}
RR_HD void vecmat(const double v[3], const double R[9], double o[3]) {     // np.dot(v, R)
  o[0] = v[0] * R[0] + v[1] * R[3] + v[2] * R[6];
  o[1] = v[0] * R[1] + v[1] * R[4] + v[2] * R[7];
  o[2] = v[0] * R[2] + v[1] * R[5] + v[2] * R[8];
}
RR_HD double py_mod(double a, double b) {   // numpy float remainder, b > 0
  double m = fmod(a, b);
  if (m != 0.0 && m < 0.0) m += b;
  return m;
}

// The polygon is evaluated in three steps so that the HIP kernel can give every (drop, vertex) pair its own
// lane (k_fov_spans) while tests/hostemu runs the same arithmetic serially:
//   fov_setup   per drop: mid-point, view direction, tilted direction v                (bad_weather.py:596-628)
//   fov_vertex  per vertex k: spin, sphere intersection, lat-long pixel                 (bad_weather.py:630-664)
//   fov_polygon the serial composition + wrap handling                                  (bad_weather.py:669-704)
struct FovSetup {
  double pos[3], n[3], v[3];
};
RR_HD bool fov_setup(const rr_drop& d, const rr_camera& cam, FovSetup& F) {
  double* pos = F.pos;
  double* n = F.n;
  pos[0] = (d.wps[0] + d.wpe[0]) / 2.0;
  pos[1] = (d.wps[2] + d.wpe[2]) / 2.0;
  pos[2] = (d.wps[1] + d.wpe[1]) / 2.0;
  double nrm = sqrt(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
  n[0] = pos[0] / nrm; n[1] = pos[1] / nrm; n[2] = pos[2] / nrm;
  double a = n[0], b = n[1], c = n[2];
  double dd = pos[0] * n[0] + pos[1] * n[1] + pos[2] * n[2];
  if (b == 0.0) b = 0.001;
  double ppx = pos[1], ppz = 0.0;
  double ppy = (-a * ppx + dd - c * ppz) / b;
  double uu[3] = {pos[0] - ppx, pos[1] - ppy, pos[2] - ppz};
  double un = sqrt(uu[0] * uu[0] + uu[1] * uu[1] + uu[2] * uu[2]);
  uu[0] /= un; uu[1] /= un; uu[2] /= un;
  if (!(uu[0] == uu[0]) || !(uu[1] == uu[1]) || !(uu[2] == uu[2])) return false;
  double rv[3] = {uu[1] * n[2] - uu[2] * n[1], uu[2] * n[0] - uu[0] * n[2], uu[0] * n[1] - uu[1] * n[0]};
  double R[9];
  rotmat(rv, cam.fov_cos, cam.fov_sin, R);
  vecmat(n, R, F.v);
  return true;
}
// vertex k: azimuth (for the wrap test) and the float pixel position on the lat-long map
RR_HD void fov_vertex(const FovSetup& F, const rr_camera& cam, double phi_cos, double phi_sin, int He, int We, double& azimuth,
                      double& ptx, double& pty) {
  const double PI = 3.141592653589793;
  const double TWO_PI = 2.0 * PI;
  const double* pos = F.pos;
  double M[9], dir[3];
  rotmat(F.n, phi_cos, phi_sin, M);
  vecmat(F.v, M, dir);
  double qa = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
  double qb = 2 * dir[0] * pos[0] + 2 * dir[1] * pos[1] + 2 * dir[2] * pos[2];
  double qc = pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2] - cam.radius * cam.radius;
  double disc = qb * qb - 4 * qa * qc;
  double t1 = (-qb + sqrt(disc)) / (2 * qa);
  double P[3] = {pos[0] + t1 * dir[0], pos[1] + t1 * dir[1], pos[2] + t1 * dir[2]};
  double el = atan2(P[2], sqrt(P[0] * P[0] + P[1] * P[1]));
  double az = atan2(P[1], P[0]);
  if (az < 0) az += TWO_PI;
  if (el < 0) el += TWO_PI;
  if (az > TWO_PI) az -= TWO_PI;
  if (el > TWO_PI) el -= TWO_PI;
  azimuth = py_mod((TWO_PI - az) - PI / 2.0, TWO_PI);
  double u = azimuth / TWO_PI;
  double elevation = py_mod(el + PI / 2.0, TWO_PI);
  double vv = 1.0 - elevation / PI;
  ptx = u * (double)We;
  pty = vv * (double)He;
}
// wrap test of one polygon side (bad_weather.py:669-672): np.isclose(df, 0) or df < 0
RR_HD bool fov_wrap_cnd(double az_k, double az_next) {
  double df = az_next - az_k;
  bool close = fabs(df) <= 1e-8;       // np.isclose(df, 0): false for NaN
  return close || (df < 0);
}
RR_HD bool fov_coord_ok(double v) { return fabs(v) < 1e15; }   // NaN/inf -> Clipper range error

// returns the number of vertices (20 or 24), or 0 where the reference's `except:` fires;
// vertices are truncated toward zero like pyclipper's integer cast.
// Clipper (pyclipper.Pyclipper.AddPath, bad_weather.py:368) strips duplicate and collinear vertices of a closed path and
// rejects it when fewer than three are left: a truncated polygon whose vertices all lie on one line (or one point) raises
// ClipperException, i.e. the drop is skipped (oracle/cvlike.py polygon_all_collinear).
RR_HD bool poly_all_collinear(const int32_t* px, const int32_t* py, int n) {
  int a = -1;
  for (int k = 1; k < n && a < 0; k++)
    if (px[k] != px[0] || py[k] != py[0]) a = k;
  if (a < 0) return true;
  const int64_t ax = (int64_t)px[a] - px[0], ay = (int64_t)py[a] - py[0];
  for (int k = 1; k < n; k++)
    if (ax * ((int64_t)py[k] - py[0]) - ay * ((int64_t)px[k] - px[0]) != 0) return false;
  return true;
}
// The float polygon's side of it: true when the vertices are not collinear whichever way each of them moved by a texel (a
// float vertex is within 1e-3 texels of the float64 one: its truncation differs by one texel at most).  Cross product of
// (P_a - P_0) and (P_k - P_0), a = n / 2: a move of the three points by one texel changes it by less than the bound.
RR_HD bool poly_surely_not_collinear_step(int ax, int ay, int bx, int by) {
  const int64_t cr = (int64_t)ax * by - (int64_t)ay * bx;
  const int64_t bound = 2 * ((int64_t)iabs(ax) + iabs(ay) + iabs(bx) + iabs(by) + 4);
  return (cr < 0 ? -cr : cr) > bound;
}
RR_HD bool poly_surely_not_collinear(const int32_t* px, const int32_t* py, int n) {
  const int a = n / 2;
  for (int k = 1; k < n; k++)
    if (poly_surely_not_collinear_step(px[a] - px[0], py[a] - py[0], px[k] - px[0], py[k] - py[0])) return true;
  return false;
}

RR_HD int fov_polygon(const rr_drop& d, const rr_camera& cam, int He, int We, int32_t* px, int32_t* py) {
  int N = cam.n_fov;
  FovSetup F;
  if (!fov_setup(d, cam, F)) return 0;
  double ptx[RR_MAX_FOV], pty[RR_MAX_FOV], azs[RR_MAX_FOV + 1];
  for (int k = 0; k < N; k++) fov_vertex(F, cam, cam.phi_cos[k], cam.phi_sin[k], He, We, azs[k], ptx[k], pty[k]);
  azs[N] = azs[0];
  int count_true = 0, count_false = 0, pos_true = -1, pos_false = -1;
  for (int k = 0; k < N; k++) {
    bool cnd = fov_wrap_cnd(azs[k], azs[k + 1]);
    if (cnd) { count_true++; if (pos_true < 0) pos_true = k; }
    else { count_false++; if (pos_false < 0) pos_false = k; }
  }
  if (pos_true < 0 || pos_false < 0) return 0;
  double fx[RR_MAX_FOV + 4], fy[RR_MAX_FOV + 4];
  int m = 0;
  double rows = (double)He, cols = (double)We;
  if (count_true == 1 || count_false == 1) {
    bool top = (count_true == 1);
    int pp = top ? pos_true : pos_false;
    for (int k = 0; k <= pp; k++) { fx[m] = ptx[k]; fy[m] = pty[k]; m++; }
    int nxt = (pp + 1) % N;
    if (top) {
      fx[m] = cols; fy[m] = pty[pp]; m++;
      fx[m] = cols; fy[m] = 0; m++;
      fx[m] = 0; fy[m] = 0; m++;
      fx[m] = 0; fy[m] = pty[nxt]; m++;
    } else {
      fx[m] = 0; fy[m] = pty[pp]; m++;
      fx[m] = 0; fy[m] = rows; m++;
      fx[m] = cols; fy[m] = rows; m++;
      fx[m] = cols; fy[m] = pty[nxt]; m++;
    }
    for (int k = pp + 1; k < N; k++) { fx[m] = ptx[k]; fy[m] = pty[k]; m++; }
  } else {
    for (int k = 0; k < N; k++) { fx[m] = ptx[k]; fy[m] = pty[k]; m++; }
  }
  for (int k = 0; k < m; k++) {
    if (!fov_coord_ok(fx[k]) || !fov_coord_ok(fy[k])) return 0;
    px[k] = (int32_t)fx[k];
    py[k] = (int32_t)fy[k];
  }
  if (poly_all_collinear(px, py, m)) return 0;               // Clipper rejects the path: skipped like a polygon that could not be made
  return m;
}

// ---------------------------------------------------------------------------
// the same polygon in float32 (colour branch; k_fov_spans' default)
// ---------------------------------------------------------------------------
// The polygon only feeds the drop's colour constants (rainy_image: +-1 LSB); what it must NOT change is a drop's status
// (the mask is bit-exact) or the shape of the polygon (20 vertices, or 24 with a wrap inserted in front of one of them).
// So the vertices are evaluated in float, and every predicate that decides a status or the wrap structure is checked
// against an error bound: a drop any of whose predicates is closer to its threshold than that bound is `unsure` and is
// evaluated again by the float64 functions above -- its polygon is then the reference's, bit for bit.  A float vertex
// lands within ~1e-3 texels of the float64 one; where that crosses an integer the truncated pixel moves by one texel
// (a relative change of the span sums of ~1e-4 in a few rows: three orders below an LSB of the image).
struct FovSetup32 {
  float pos[3], n[3], v[3];
};
RR_HD void rotmat32(const float a[3], float c, float s, float R[9]) {
  float omc = 1.0f - c;
  R[0] = c + omc * (a[0] * a[0]);
  R[1] = s * (-a[2]) + omc * (a[0] * a[1]);
  R[2] = s * (a[1]) + omc * (a[0] * a[2]);
  R[3] = s * (a[2]) + omc * (a[1] * a[0]);
  R[4] = c + omc * (a[1] * a[1]);
  R[5] = s * (-a[0]) + omc * (a[1] * a[2]);
  R[6] = s * (-a[1]) + omc * (a[2] * a[0]);
  R[7] = s * (a[0]) + omc * (a[2] * a[1]);
  R[8] = c + omc * (a[2] * a[2]);
}
RR_HD void vecmat32(const float v[3], const float R[9], float o[3]) {
  o[0] = v[0] * R[0] + v[1] * R[3] + v[2] * R[6];
  o[1] = v[0] * R[1] + v[1] * R[4] + v[2] * R[7];
  o[2] = v[0] * R[2] + v[1] * R[5] + v[2] * R[8];
}
// returns false where fov_setup would (never when `unsure` is clear); unsure: a float64 evaluation must decide
RR_HD bool fov_setup32(const rr_drop& d, float fov_cos, float fov_sin, FovSetup32& F, int& unsure) {
  float* pos = F.pos;
  float* n = F.n;
  pos[0] = (float)((d.wps[0] + d.wpe[0]) / 2.0);
  pos[1] = (float)((d.wps[2] + d.wpe[2]) / 2.0);
  pos[2] = (float)((d.wps[1] + d.wpe[1]) / 2.0);
  float nrm = sqrtf(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
  n[0] = pos[0] / nrm; n[1] = pos[1] / nrm; n[2] = pos[2] / nrm;
  float a = n[0], b = n[1], c = n[2];
  float dd = pos[0] * n[0] + pos[1] * n[1] + pos[2] * n[2];
  // b == 0 -> 0.001 in the reference: a drop whose mid-point has (nearly) no depth; the in-plane direction below is
  // ill-conditioned for small |b| anyway
  if (!(fabsf(b) > 1e-2f) || !(nrm > 1e-6f) || !(nrm < 1e6f)) unsure |= 1;
  float ppx = pos[1], ppz = 0.0f;
  float ppy = (-a * ppx + dd - c * ppz) / b;
  float uu[3] = {pos[0] - ppx, pos[1] - ppy, pos[2] - ppz};
  float un = sqrtf(uu[0] * uu[0] + uu[1] * uu[1] + uu[2] * uu[2]);
  if (!(un > 1e-3f * nrm)) unsure |= 1;                    // (also NaN)
  uu[0] /= un; uu[1] /= un; uu[2] /= un;
  float rv[3] = {uu[1] * n[2] - uu[2] * n[1], uu[2] * n[0] - uu[0] * n[2], uu[0] * n[1] - uu[1] * n[0]};
  float R[9];
  rotmat32(rv, fov_cos, fov_sin, R);
  vecmat32(n, R, F.v);
  return true;
}
// vertex k in float: azimuth, a bound of its error (grows towards the poles of the map), float pixel position
RR_HD void fov_vertex32(const FovSetup32& F, float radius, float phi_cos, float phi_sin, int He, int We, float& azimuth, float& az_err,
                        float& ptx, float& pty, int& unsure) {
  const float PI = 3.14159265358979f, TWO_PI = 6.28318530717959f;
  const float* pos = F.pos;
  float M[9], dir[3];
  rotmat32(F.n, phi_cos, phi_sin, M);
  vecmat32(F.v, M, dir);
  float qa = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
  float qb = 2.0f * (dir[0] * pos[0] + dir[1] * pos[1] + dir[2] * pos[2]);
  float p2 = pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2];
  float qc = p2 - radius * radius;
  float disc = qb * qb - 4.0f * qa * qc;
  // the sign of the discriminant decides a status (no intersection -> NaN -> no polygon): clearly negative is a certain
  // failure (bit 128: the drop is farther out than the sphere), close to zero is for float64 to decide
  {
    const float scale = 1e-3f * (qb * qb + 4.0f * qa * fabsf(qc));
    if (disc < -scale) unsure |= 128;
    else if (!(disc > scale)) unsure |= 2;
  }
  float t1 = (-qb + sqrtf(fmaxf(disc, 0.0f))) / (2.0f * qa);
  float P[3] = {pos[0] + t1 * dir[0], pos[1] + t1 * dir[1], pos[2] + t1 * dir[2]};
  float rho = sqrtf(P[0] * P[0] + P[1] * P[1]);
  float el = atan2f(P[2], rho);
  float az = atan2f(P[1], P[0]);
  if (az < 0.0f) az += TWO_PI;
  float x = (TWO_PI - az) - 0.5f * PI;                          // in (-pi/2, 3 pi/2]: one conditional add is the modulus
  if (x < 0.0f) x += TWO_PI;
  azimuth = x;
  // |P| = radius up to rounding; the azimuth of a point at horizontal distance rho is known to ~(absolute error of P) / rho
  az_err = 1e-6f * (radius + sqrtf(p2)) / fmaxf(rho, 1e-30f) + 2e-6f;   // (measured: <= 0.3 of this, scripts/fov_f32_polygons.py)
  if (!(az_err < 0.05f)) unsure |= 4;                        // on top of a pole (also NaN)
  if (azimuth < 4.0f * az_err || azimuth > TWO_PI - 4.0f * az_err) unsure |= 8;   // which side of the seam of the map?
  float elevation = el + 0.5f * PI;                            // [0, pi]
  ptx = (azimuth / TWO_PI) * (float)We;
  pty = (1.0f - elevation / PI) * (float)He;
}
// wrap test of one polygon side; unsure when the difference is within the two vertices' error of either threshold
RR_HD bool fov_wrap_cnd32(float az_k, float az_next, float err_k, float err_next, int& unsure) {
  float df = az_next - az_k;
  if (!(fabsf(df) > 4.0f * (err_k + err_next) + 1e-5f)) unsure |= 16;
  return df < 0.0f;
}
// serial composition (tests/hostemu): the float polygon when every predicate is clear of its threshold, else the float64
// one; *used32 says which.  Same vertex order and wrap insertion as fov_polygon.
RR_HD int fov_polygon_auto(const rr_drop& d, const rr_camera& cam, int He, int We, int32_t* px, int32_t* py, int* used32) {
  const int N = cam.n_fov;
  int unsure = 0;
  FovSetup32 F;
  fov_setup32(d, (float)cam.fov_cos, (float)cam.fov_sin, F, unsure);
  float ptx[RR_MAX_FOV], pty[RR_MAX_FOV], azs[RR_MAX_FOV + 1], ers[RR_MAX_FOV + 1];
  for (int k = 0; k < N; k++)
    fov_vertex32(F, (float)cam.radius, (float)cam.phi_cos[k], (float)cam.phi_sin[k], He, We, azs[k], ers[k], ptx[k], pty[k], unsure);
  azs[N] = azs[0];
  ers[N] = ers[0];
  int count_true = 0, count_false = 0, pos_true = -1, pos_false = -1;
  for (int k = 0; k < N; k++) {
    bool cnd = fov_wrap_cnd32(azs[k], azs[k + 1], ers[k], ers[k + 1], unsure);
    if (cnd) { count_true++; if (pos_true < 0) pos_true = k; }
    else { count_false++; if (pos_false < 0) pos_false = k; }
  }
  int m = 0;
  if ((unsure & 128) && !(unsure & (1 | 2))) {                 // a vertex certainly has no intersection: [] like the reference
    if (used32) *used32 = 1;
    return 0;
  }
  if (!unsure && pos_true >= 0 && pos_false >= 0) {
    const bool wrap = count_true == 1 || count_false == 1, top = count_true == 1;
    const int pp = top ? pos_true : pos_false;
    const int rows = He, cols = We;
    for (int k = 0; k < N; k++) {
      px[m] = (int32_t)ptx[k]; py[m] = (int32_t)pty[k]; m++;
      if (wrap && k == pp) {
        const int nxt = (pp + 1) % N;
        if (top) {
          px[m] = cols; py[m] = (int32_t)pty[pp]; m++;
          px[m] = cols; py[m] = 0; m++;
          px[m] = 0; py[m] = 0; m++;
          px[m] = 0; py[m] = (int32_t)pty[nxt]; m++;
        } else {
          px[m] = 0; py[m] = (int32_t)pty[pp]; m++;
          px[m] = 0; py[m] = rows; m++;
          px[m] = cols; py[m] = rows; m++;
          px[m] = cols; py[m] = (int32_t)pty[nxt]; m++;
        }
      }
    }
    // a polygon that covers (almost) no row of the map could flip RR_DROP_EMPTY_FOV with a one-texel move: float64 decides
    // (the kernel's rule: no vertex two rows away from the first one; a wrapping polygon spans the map)
    if (!wrap) {
      const int r0 = imin(imax((int)pty[0], 0), He - 1);
      bool spread = false;
      for (int k = 1; k < N; k++) spread = spread || iabs(imin(imax((int)pty[k], 0), He - 1) - r0) >= 2;
      if (!spread) unsure |= 32;
    }
    if (!poly_surely_not_collinear(px, py, m)) unsure |= 256;  // a sliver: whether Clipper takes the path is float64's to say
  } else {
    unsure |= 64;                                              // (or already unsure) all sides alike: the reference returns [] -- float64 says so
  }
  if (used32) *used32 = unsure ? -unsure : 1;            // (tests: <= 0 tells why the float64 polygon was taken)
  if (unsure) return fov_polygon(d, cam, He, We, px, py);
  return m;
}

// FOV row span at pixel row y (oracle/cvlike.py fov_rowspans). false if the row is empty.
RR_HD bool fov_rowspan(const int32_t* px, const int32_t* py, int n, int y, int We, int& xl, int& xr) {
  int lo = 1 << 30, hi = -(1 << 30);
  for (int i = 0; i < n; i++) {
    int j = (i + 1 == n) ? 0 : i + 1;
    int x0 = px[i], y0 = py[i], x1 = px[j], y1 = py[j];
    int ylo = imin(y0, y1), yhi = imax(y0, y1);
    if (y < ylo || y > yhi) continue;
    if (y0 == y1) {
      lo = imin(lo, imin(x0, x1));
      hi = imax(hi, imax(x0, x1));
    } else {
      int xa, yA, xb, yB;
      if (y1 < y0) { xa = x1; yA = y1; xb = x0; yB = y0; } else { xa = x0; yA = y0; xb = x1; yB = y1; }
      int64_t den = yB - yA;
      int64_t num = (int64_t)(xb - xa) * (int64_t)(y - yA);
      // floor((2*num + den) / (2*den)), exact: |values| < 2^40
      int64_t nn = 2 * num + den, dn = 2 * den;
      int64_t q = nn / dn;
      if ((nn % dn != 0) && ((nn < 0) != (dn < 0))) q -= 1;
      int xv = xa + (int)q;
      lo = imin(lo, xv);
      hi = imax(hi, xv);
    }
  }
  xl = imax(lo, 0);
  xr = imin(hi, We - 1);
  return xl <= xr;
}

// ---- OpenCV's own rule (r05): what cv2.fillConvexPoly(mask, s, 1) sets in row y ----
// The reference fills the polygon with OpenCV 3.2's FillConvexPoly (bad_weather.py:388): the OUTLINE drawn edge by edge
// with Line() -- an 8-connected Bresenham line from the edge's left end point -- and the interior by two edge walkers in
// 16.16 fixed point, x += dx per row with dx = ((xb - xa) * 2^17 + den) / (2 den) (C division), a row filled from
// (min + 2^15) >> 16 to (max + 2^15) >> 16; the walkers take an edge on the rows ya <= y < yb.  oracle/cvlike.py
// cv_fill_convex_poly restates it literally (pixel by pixel); this is its closed form per edge and row, t = y - ya:
//   N = 2 dx t = Q * 2 den + R   (0 <= R < 2 den)
//   steep edge (den > |dx|): one pixel per row, xa + ceil((2 |dx| i - den) / (2 den)) from the left end -- for either
//     direction that is  xa + Q + [R >= den + 1]   (the span rule above has [R >= den]: ties round the other way);
//   shallow edge: a run of pixels per row,  [xa + Q - h + [R >= hr],  xa + Q + h + [R + hr >= 2 den]]  with
//     |dx| = h * 2 den + hr, clamped to the edge's own x range (the runs of the first and the last row stop at the vertex);
//   walker: (xa * 2^16 + t * dx16 + 2^15) >> 16 on the rows t < den.
// A row's span is min / max over the edges that touch it.  Applies to the closed N-gon that compute_fov_plane_points
// returns when it does not wrap (rows monotone down one side and up the other, every vertex on the map: no clipLine, no
// dependence on where Clipper starts its output); a wrapping (N + 4)-gon is not y-monotone, FillConvexPoly's result for it
// depends on the vertex Clipper happens to list first, which cannot be reconstructed: those keep the span rule.
// tests/test_fill_rules.py: closed form == the literal restatement on every polygon of the test scenes, in any rotation
// and orientation of the vertex list.
RR_HD bool fov_fill_rule_cv_applies(const int32_t* px, const int32_t* py, int n, int n_fov, int He, int We);
RR_HD bool fov_rowspan_cv(const int32_t* px, const int32_t* py, int n, int y, int We, int& xl, int& xr) {
  int lo = 1 << 30, hi = -(1 << 30);
  for (int i = 0; i < n; i++) {
    const int j = (i + 1 == n) ? 0 : i + 1;
    const int x0 = px[i], y0 = py[i], x1 = px[j], y1 = py[j];
    if (y < imin(y0, y1) || y > imax(y0, y1)) continue;
    if (y0 == y1) {                                          // Line() along the row
      lo = imin(lo, imin(x0, x1));
      hi = imax(hi, imax(x0, x1));
      continue;
    }
    const bool swp = y1 < y0;
    const int xa = swp ? x1 : x0, ya = swp ? y1 : y0, xb = swp ? x0 : x1, yb = swp ? y0 : y1;
    const int den = yb - ya, dn = 2 * den, dx = xb - xa, dxa = iabs(dx), t = y - ya;
    const int64_t N2 = (int64_t)2 * dx * t;
    int64_t Q = N2 / dn;
    if (N2 % dn != 0 && N2 < 0) Q -= 1;                      // floor
    const int R = (int)(N2 - Q * dn);
    int l, h_;
    if (den > dxa) {
      l = h_ = xa + (int)Q + (R >= den + 1 ? 1 : 0);
    } else {
      const int hh = dxa / dn, hr = dxa - hh * dn;
      l = xa + (int)Q - hh + (R >= hr ? 1 : 0);
      h_ = xa + (int)Q + hh + (R + hr >= dn ? 1 : 0);
      l = imax(l, imin(xa, xb));
      h_ = imin(h_, imax(xa, xb));
    }
    lo = imin(lo, l);
    hi = imax(hi, h_);
    if (t < den) {                                           // the edge walker's pixel
      const int64_t num = ((int64_t)dx << 17) + den;
      const int64_t dx16 = num / dn;                         // C division: toward zero
      const int s = (int)((((int64_t)xa << 16) + (int64_t)t * dx16 + 32768) >> 16);
      lo = imin(lo, s);
      hi = imax(hi, s);
    }
  }
  xl = imax(lo, 0);
  xr = imin(hi, We - 1);
  return xl <= xr;
}

// The per-edge constants of OpenCV's rule (upper end (xa, ya), lower end (xb, yb), den = yb - ya > 0, dx = xb - xa):
//   dx16  the walker's step,  hh / hr  quotient / remainder of |dx| by 2 den (the outline's pixels per row).
RR_HD void edge_cv_consts(int dx, int den, int& dx16, int& hh, int& hr) {
  const int dn = 2 * den, dxa = iabs(dx);
  const int num = dx * 131072 + den;                         // |dx| <= 4095 (the fast colour path's maps): below 2^30
  dx16 = num / dn;                                           // C division: toward zero
  hh = dxa / dn;
  hr = dxa - hh * dn;
}
// One edge's part of row t = y - ya (0 <= t <= den) under OpenCV's rule: [l, h].  Q, R: floor quotient and remainder of
// 2 dx t by 2 den.  (fov_rowspan_cv evaluates the same per edge; the span rule's single pixel is xa + Q + [R >= den].)
RR_HD void edge_row_cv(int xa, int xb, int den, int dx, int dx16, int hh, int hr, int t, int Q, int R, int& l, int& h) {
  const int dn = 2 * den;
  if (den > iabs(dx)) {
    l = h = xa + Q + (R >= den + 1 ? 1 : 0);
  } else {
    l = xa + Q - hh + (R >= hr ? 1 : 0);
    h = xa + Q + hh + (R + hr >= dn ? 1 : 0);
    l = imax(l, imin(xa, xb));
    h = imin(h, imax(xa, xb));
  }
  if (t < den) {                                             // the edge walker's pixel: |t * dx16| <= |dx| * 2^16 < 2^29
    const int s = xa + ((t * dx16 + 32768) >> 16);
    l = imin(l, s);
    h = imax(h, s);
  }
}

// Row spans of a closed polygon whose vertex rows go down one side and up the other (every row crosses it at most twice:
// a circle on the sphere that contains no pole), by two cursors walking down from its top vertex, one along each side
// (k_fov_dda: a thread per drop).  row(y) returns the min / max over the edges that touch row y of their pixels on that row
// -- under the span rule (fov_rowspan) or OpenCV's (fov_rowspan_cv), per polygon -- for y from the top vertex' row to the
// bottom one, IN ASCENDING ORDER (every call advances the cursors).
//
// Round 6: both rules by INCREMENTAL stepping.  With Q(t), R(t) the floor quotient and remainder of 2 dx t by 2 den (t = y - ya):
//   span rule      x = xa + Q + [R >= den]
//   OpenCV, steep  x = xa + Q + [R >= den + 1]
//   OpenCV, shallow  [xa + Q - hh + [R >= hr],  xa + Q + hh + [R >= 2 den - hr]]  clamped to the edge's own x range
// i.e. one form  [xa + Q - hh + [R >= tl],  xa + Q + hh + [R >= th]]  with per-edge constants, and a row down the edge is
// Q += qs + carry, R += rs - carry * 2 den  (qs, rs: quotient and remainder of 2 dx by 2 den); OpenCV's edge walker adds the
// pixel xa + (w >> 16), w = 2^15 + t * dx16.  An edge's divisions are done ONCE, before the walk (all lanes busy), into an
// 8-byte record per edge and lane; a cursor fetches the next vertex and record an edge ahead, so that taking an edge costs no
// wait for the LDS.
// Limits: den <= 1023, |dx| <= 4095 (maps of the fast colour path: 1024 x 4096).
RR_HD int mul24i(int a, int b) {                            // a * b for |a|, |b| < 2^23 (the 24-bit multiplier on the device)
#if defined(__HIP_DEVICE_COMPILE__)
  return __mul24(a, b);
#else
  return a * b;
#endif
}
// Record of the undirected edge between an upper end (xu, yu) and a lower end (xl, yl), den = yl - yu, dx = xl - xu, made
// before the walk (all lanes busy: this is where the divisions are), 16 bytes (round 6; 8 before):
//   r[0] = dx16 = ((dx << 17) + den) / (2 den), C division: OpenCV's walker step -- and qs = r[0] >> 16 (the fraction of
//          2 dx / 2 den is 0 or at least 1 / 2046: the half a unit dx16 is rounded by never reaches the next integer)
//   r[1] = hh | tl << 11 | same << 22 | walker << 23     (same: th = tl; else th = 2 den - tl)
//   r[2] = the edge's pixels on its FIRST row (the upper end's), lo | hi << 16 -- DdaCursors::pixels at t = 0, evaluated here
//          once instead of by every lane of a wave whenever one of its 64 cursors takes an edge
//   r[3] = the lower end, x | y << 16: a cursor needs no vertex table while it walks
// A horizontal edge {k, k + 1} (xu = vertex k, xl = vertex k + 1 by the callers' convention): r[1] = 0, r[2] = both end
// points' span, and r[0] = vertex k's x -- the end of the edge for the cursor that walks the vertices downwards in index.
struct DdaSide {                                             // one cursor: its current edge (xa, ya) .. (xb, yb)
  int xa, xb, yb, kv;
  int mn, mx;                                                // min / max of xa, xb: the edge's own x range
  int Q, R, W;                                               // at the row the cursor stands on
  int dn, hh, tl, th, qs, rs, d16, wk;                       // the edge's constants (dn == 0: horizontal; wk: OpenCV's walker)
  uint32_t n_r0, n_r1, n_r2, n_r3;                           // the NEXT edge's record, fetched an edge ahead
};
// the edge's constants from (its ends and) the first two words of its record; the cursor at t = 0
RR_HD void dda_take_edge(DdaSide& S, int xa, int ya, int xb, int yb, uint32_t w0, uint32_t w1) {
  S.xa = xa;
  S.xb = xb;
  S.yb = yb;
  S.mn = imin(xa, xb);
  S.mx = imax(xa, xb);
  const int den = yb - ya, dx = xb - xa;
  S.dn = 2 * den;
  S.Q = 0;
  S.R = 0;
  S.W = 32768;
  S.d16 = (int)w0;
  S.wk = (int)((w1 >> 23) & 1u);
  S.hh = (int)(w1 & 0x7ffu);
  S.tl = (int)((w1 >> 11) & 0x7ffu);
  S.th = ((w1 >> 22) & 1u) ? S.tl : S.dn - S.tl;
  S.qs = S.d16 >> 16;
  S.rs = 2 * dx - mul24i(S.qs, S.dn);
}
// the pixels [x0, x1] of the cursor's edge on the row it stands on (y; y == yb: its last row)
RR_HD void dda_pixels(const DdaSide& S, int y, int& x0, int& x1) {
  const int base = S.xa + S.Q;
  x0 = imax(base - S.hh + (S.R >= S.tl ? 1 : 0), S.mn);     // (the clamps: no-ops for the one-pixel forms)
  x1 = imin(base + S.hh + (S.R >= S.th ? 1 : 0), S.mx);
  if (S.wk && y < S.yb) {                                    // OpenCV's walker: rows ya <= y < yb
    const int sx = S.xa + (S.W >> 16);
    x0 = imin(x0, sx);
    x1 = imax(x1, sx);
  }
  if (S.dn == 0) {                                           // horizontal: both end points
    x0 = S.mn;
    x1 = S.mx;
  }
}
RR_HD void dda_edge_record(int xu, int yu, int xl, int yl, bool cv, uint32_t r[4]) {
  const int den = yl - yu, dx = xl - xu;
  r[0] = r[1] = 0u;
  if (den > 0) {                                             // (horizontal: the cursor takes both end points)
    int dx16, hh, hr;
    edge_cv_consts(dx, den, dx16, hh, hr);
    r[0] = (uint32_t)dx16;
    if (!cv) r[1] = 0u | ((uint32_t)den << 11) | (1u << 22);
    else if (den > iabs(dx)) r[1] = 0u | ((uint32_t)(den + 1) << 11) | (1u << 22) | (1u << 23);
    else r[1] = (uint32_t)hh | ((uint32_t)hr << 11) | (1u << 23);
  }
  DdaSide S;
  dda_take_edge(S, xu, yu, xl, yl, r[0], r[1]);
  int x0, x1;
  dda_pixels(S, yu, x0, x1);
  r[2] = (uint32_t)(x0 & 0xffff) | ((uint32_t)(x1 & 0xffff) << 16);
  r[3] = (uint32_t)(xl & 0xffff) | ((uint32_t)(yl & 0xffff) << 16);
  if (den <= 0) r[0] = (uint32_t)(xu & 0xffff);
}
// E: rec(k, r) = the record of edge {k, k + 1} (upper end first).
template <class E>
struct DdaCursors {
  DdaSide s0, s1;                                            // cursor 0 walks the vertices upwards in index from the top vertex, cursor 1 downwards
  int used, N;
  template <int C>
  RR_HD static int next_k(int k, int n) { return C == 0 ? (k + 1 == n ? 0 : k + 1) : (k == 0 ? n - 1 : k - 1); }
  // the record of the edge a cursor at vertex kv takes next
  template <int C>
  RR_HD void fetch(DdaSide& S, const E& rec) const {
    uint32_t r[4];
    rec(C == 0 ? S.kv : next_k<C>(S.kv, N), r);
    S.n_r0 = r[0]; S.n_r1 = r[1]; S.n_r2 = r[2]; S.n_r3 = r[3];
  }
  // the cursor moves on to the edge that was fetched ahead (from (xb, yb), which becomes (xa, ya)) and fetches the one after
  // it; STEP: the cursor stands on the new edge's first row, whose pixels come from the record -- it is left on the second
  template <int C, bool STEP>
  RR_HD void advance_edge(DdaSide& S, const E& rec, int& lo, int& hi) const {
    const int xa = S.xb, ya = S.yb;
    uint32_t w0 = S.n_r0;
    const uint32_t w1 = S.n_r1, w2 = S.n_r2, w3 = S.n_r3;
    S.kv = next_k<C>(S.kv, N);
    fetch<C>(S, rec);
    int xe = (int)(w3 & 0xffffu);
    const int ye = (int)(w3 >> 16);
    if (ye == ya) {                                          // horizontal: which end the cursor arrives at depends on its direction
      if (C == 1) xe = (int)w0;
      w0 = 0u;
    }
    dda_take_edge(S, xa, ya, xe, ye, w0, w1);
    if (STEP) {
      lo = imin(lo, (int)(w2 & 0xffffu));
      hi = imax(hi, (int)(w2 >> 16));
      if (S.dn > 0) {                                        // t = 1 (a horizontal edge: the next vertex follows on this row)
        const int carry = S.rs >= S.dn ? 1 : 0;
        S.R = S.rs - (carry ? S.dn : 0);
        S.Q = S.qs + carry;
        S.W = 32768 + S.d16;
      }
    }
  }
  template <int C>
  RR_HD void init_side(DdaSide& S, const E& rec, int ktop, uint32_t top_xy) const {
    S.xb = (int)(top_xy & 0xffffu);                          // "the end of the edge before": the top vertex
    S.yb = (int)(top_xy >> 16);
    S.kv = ktop;
    fetch<C>(S, rec);
    int lo = 0, hi = 0;
    advance_edge<C, false>(S, rec, lo, hi);
  }
  RR_HD void init(const E& rec, int n, int ktop, uint32_t top_xy) {      // top_xy: the top vertex, x | y << 16
    N = n;
    used = 2;                                                // edges taken so far (both cursors together; N in all)
    init_side<0>(s0, rec, ktop, top_xy);
    init_side<1>(s1, rec, ktop, top_xy);
  }
  RR_HD static void step(DdaSide& S) {                       // a row down the current edge
    const int r = S.R + S.rs;
    const int carry = r >= S.dn ? 1 : 0;
    S.R = r - (carry ? S.dn : 0);
    S.Q += S.qs + carry;
    S.W += S.d16;
  }
  template <int C>
  RR_HD void side_row(DdaSide& S, const E& rec, int y, int& lo, int& hi) {
    {
      int x0, x1;
      dda_pixels(S, y, x0, x1);
      lo = imin(lo, x0);
      hi = imax(hi, x1);
    }
    step(S);                                                 // (unused once the cursor switches below)
    while (y == S.yb && used < N) {                          // a vertex row: the edges that start here touch it too
      used++;
      advance_edge<C, true>(S, rec, lo, hi);
    }
  }
  RR_HD void row(const E& rec, int y, int& lo, int& hi) {
    lo = 1 << 30;
    hi = -(1 << 30);
    side_row<0>(s0, rec, y, lo, hi);
    side_row<1>(s1, rec, y, lo, hi);
  }
};

// number of times the vertices' row sequence changes direction around the loop (a closed monotone curve: 2; all on one row: 0)
RR_HD int poly_row_turns(const int32_t* py, int n) {
  int turns = 0, dir = 0, dir_first = 0;
  for (int k = 1; k <= n; k++) {
    const int a = py[k - 1], b = py[k == n ? 0 : k];
    const int sg = b > a ? 1 : (b < a ? -1 : 0);
    if (sg != 0) {
      if (dir == 0) dir_first = sg;
      else if (sg != dir) turns++;
      dir = sg;
    }
  }
  if (dir != 0 && dir_first != 0 && dir != dir_first) turns++;
  return turns;
}

RR_HD bool fov_fill_rule_cv_applies(const int32_t* px, const int32_t* py, int n, int n_fov, int He, int We) {
  if (n != n_fov || n < 3) return false;                     // (a wrapping polygon has n_fov + 4 vertices)
  for (int k = 0; k < n; k++)
    if (px[k] < 0 || px[k] >= We || py[k] < 0 || py[k] >= He) return false;
  return poly_row_turns(py, n) <= 2;
}

// colour constants from the FOV sums (bad_weather.py:397-412, my_utils.py:55-85)
// S = {sum x*w, sum y*w, sum Y*w, sum w} over the FOV mask.
RR_HD void colour_from_sums(const double S[4], double sum_omega, double ambient, double Kbgr[3]) {
  double x = S[0] / S[3], y = S[1] / S[3];
  double avg_fov_lum = S[2] / sum_omega;
  double drop_Y = 0.94 * avg_fov_lum + 0.06 * ambient;
  // a gray pixel v has Y = v*(0.31+0.8124+0.01)/0.17697 (row-vector times matrix)
  double Yg = ((0.31 + 0.8124) + 0.01) / 0.17697 * drop_Y;
  double X = (Yg * x) / y;
  double Z = (Yg * (1 - x - y)) / y;
  double r = (X * 0.41847 + Yg * -0.091169) + Z * 0.0009209;
  double g = (X * -0.15866 + Yg * 0.25243) + Z * -0.0025498;
  double b = (X * -0.082835 + Yg * 0.015708) + Z * 0.1786;
  Kbgr[0] = b;
  Kbgr[1] = g;
  Kbgr[2] = r;
}

// ---------------------------------------------------------------------------
// per-drop plan  (generator.py:119-174, bad_weather.py:286-329,416-427)
// ---------------------------------------------------------------------------
// Gaussian elimination with partial pivoting, the operations of oracle/cvlike.py solve8 in their order.  Every index is a
// compile-time constant once the loops are unrolled (the pivot row is swapped in by comparing each candidate's number
// with the pivot's, not by indexing with it): on the GPU the 8 x 8 system then lives in registers instead of scratch
// memory -- k_plan's dynamically indexed private arrays were 5 GB of scratch traffic per launch.
RR_HD void solve8(double A[8][8], double b[8], double x[8]) {   // oracle/cvlike.py solve8
#pragma unroll
  for (int col = 0; col < 8; col++) {
    int piv = col;
    double best = fabs(A[col][col]);
#pragma unroll
    for (int r = col + 1; r < 8; r++)
      if (fabs(A[r][col]) > best) { best = fabs(A[r][col]); piv = r; }
#pragma unroll
    for (int r = col + 1; r < 8; r++)
      if (r == piv) {
#pragma unroll
        for (int c = 0; c < 8; c++) { double t = A[col][c]; A[col][c] = A[r][c]; A[r][c] = t; }
        double t = b[col]; b[col] = b[r]; b[r] = t;
      }
#pragma unroll
    for (int r = col + 1; r < 8; r++) {
      double f = A[r][col] / A[col][col];
#pragma unroll
      for (int c = col + 1; c < 8; c++) A[r][c] = A[r][c] - f * A[col][c];
      b[r] = b[r] - f * b[col];
    }
  }
#pragma unroll
  for (int r = 7; r >= 0; r--) {
    double s = b[r];
#pragma unroll
    for (int c = r + 1; c < 8; c++) s = s - A[r][c] * x[c];
    x[r] = s / A[r][r];
  }
}

RR_HD void invert3(const double m[9], double t[9]) {             // cv::invert 3x3
  double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (d == 0.0 || !(fabs(d) < 1.7e308)) {
    for (int i = 0; i < 9; i++) t[i] = 0.0;
    return;
  }
  d = 1.0 / d;
  t[0] = (m[4] * m[8] - m[5] * m[7]) * d;
  t[1] = (m[2] * m[7] - m[1] * m[8]) * d;
  t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
  t[3] = (m[5] * m[6] - m[3] * m[8]) * d;
  t[4] = (m[0] * m[8] - m[2] * m[6]) * d;
  t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
  t[6] = (m[3] * m[7] - m[4] * m[6]) * d;
  t[7] = (m[1] * m[6] - m[0] * m[7]) * d;
  t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
}

// Fills everything in `p` except the arena offsets.  `size_out` = doubles of arena needed.
RR_HD int py_slice_index(int i, int n) {   // CPython slice normalisation of one bound, step 1
  if (i < 0) { i += n; if (i < 0) i = 0; }
  else if (i > n) i = n;
  return i;
}

// What a drop's RAW tile (before the defocus blur) is a function of, 32 bytes: everything plan_drop derives for the tile
// kernels -- the inverse homography / rotation, canvas, scales, block width -- follows from these inputs and the texture's
// size, so two drops of a batch with equal keys get bit-identical tiles and share one (k_dedup compares keys instead of
// the ~200 bytes of derived plan fields).  key[0] = kind | flip << 4 | tex << 8 (0xffffffff: never shared -- a skipped
// drop, a caller-made tile, endpoint widths outside int32), key[1] = tw | th << 16,
//   Big drops (generator.py:126-132):    key[2..7] = x0 - minx, y0 - miny, x1 - minx, y1 - miny, floor(iw1), floor(iw2)
//   Medium / Small (generator.py:133-171): key[2..5] = the bits of cos, sin of the streak's angle
RR_HD void raw_tile_key(const rr_drop& d, const DropPlan& p, uint32_t key[8]) {
  for (int k = 0; k < 8; k++) key[k] = 0u;
  key[0] = 0xffffffffu;
  if (p.status != RR_DROP_OK || (p.kind != KIND_BIG && p.kind != KIND_ROT)) return;
  if (p.tw < 0 || p.tw > 65535 || p.th < 0 || p.th > 65535) return;
  if (p.kind == KIND_BIG) {
    const double d0 = floor(d.iw1), d1 = floor(d.iw2);
    if (!(d0 > -2.0e9 && d0 < 2.0e9 && d1 > -2.0e9 && d1 < 2.0e9)) return;
    const int minx = imax(imin(d.x0, d.x1), 0), miny = imax(imin(d.y0, d.y1), 0);
    key[2] = (uint32_t)(d.x0 - minx);
    key[3] = (uint32_t)(d.y0 - miny);
    key[4] = (uint32_t)(d.x1 - minx);
    key[5] = (uint32_t)(d.y1 - miny);
    key[6] = (uint32_t)(int32_t)d0;
    key[7] = (uint32_t)(int32_t)d1;
  } else {
    uint64_t a, b;
    const double al = d.rot_cos, be = d.rot_sin;
    __builtin_memcpy(&a, &al, 8);
    __builtin_memcpy(&b, &be, 8);
    key[2] = (uint32_t)a; key[3] = (uint32_t)(a >> 32);
    key[4] = (uint32_t)b; key[5] = (uint32_t)(b >> 32);
  }
  key[0] = (uint32_t)p.kind | ((uint32_t)(p.flip & 1) << 4) | ((uint32_t)p.tex << 8);
  key[1] = (uint32_t)p.tw | ((uint32_t)p.th << 16);
}

// The inverse homography of a Big drop (generator.py:126-132: warping_points -> getPerspectiveTransform -> warpPerspective's
// inversion), from the drop's end points and the texture's size alone.  The 8 x 8 solve is the register-hungry part of the
// per-drop plan: on the device k_plan leaves it to k_plan_big (round 6; DEFER_BIG below), which runs it for the Big drops only.
RR_HD void plan_big_homography(const rr_drop& d, int sw, int sh, double mi[9]) {
  const double d0 = floor(d.iw1), d1 = floor(d.iw2);
  const int minx = imax(imin(d.x0, d.x1), 0), miny = imax(imin(d.y0, d.y1), 0);
  float src[4][2] = {{0.f, 0.f}, {(float)sw, 0.f}, {(float)sw, (float)sh}, {0.f, (float)sh}};
  float dst[4][2];
  dst[0][0] = (float)(d.x0 - minx);                        dst[0][1] = (float)(d.y0 - miny);
  dst[1][0] = (float)((double)(d.x0 - minx) + d0);         dst[1][1] = (float)(d.y0 - miny);
  dst[2][0] = (float)(((double)(d.x1 - minx) + d1) + 0.001); dst[2][1] = (float)(d.y1 - miny);
  dst[3][0] = (float)((double)(d.x1 - minx) + 0.001);      dst[3][1] = (float)(d.y1 - miny);
  double A[8][8], bb[8], xx[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) A[i][j] = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double sx = src[i][0], sy = src[i][1], ddx = dst[i][0], ddy = dst[i][1];
    A[i][0] = A[i + 4][3] = sx;
    A[i][1] = A[i + 4][4] = sy;
    A[i][2] = A[i + 4][5] = 1.0;
    A[i][6] = -sx * ddx;
    A[i][7] = -sy * ddx;
    A[i + 4][6] = -sx * ddy;
    A[i + 4][7] = -sy * ddy;
    bb[i] = ddx;
    bb[i + 4] = ddy;
  }
  solve8(A, bb, xx);
  double M[9] = {xx[0], xx[1], xx[2], xx[3], xx[4], xx[5], xx[6], xx[7], 1.0};
  invert3(M, mi);
}

// The caller's tile, by value: (has_ext, tw, th, min_x, min_y).  (A pointer to a local rr_ext_tile kept that struct in scratch
// memory on the device.)
struct ExtGeom {
  bool has;
  int tw, th, min_x, min_y;
};
template <bool DEFER_BIG = false>     // true: a Big drop's mi[] is left zero (the caller runs plan_big_homography later)
RR_HD void plan_drop(const rr_drop& d, const rr_camera& cam, const Dims& dm, const int32_t* tex_h, const int32_t* tex_w,
                     double opacity_attenuation, int strategy, DropPlan& p, int64_t& size_out, const ExtGeom xg);
template <bool DEFER_BIG = false>
RR_HD void plan_drop(const rr_drop& d, const rr_camera& cam, const Dims& dm, const int32_t* tex_h, const int32_t* tex_w,
                     double opacity_attenuation, int strategy, DropPlan& p, int64_t& size_out, const rr_ext_tile* ext = nullptr) {
  const bool has = ext && ext->alpha;
  plan_drop<DEFER_BIG>(d, cam, dm, tex_h, tex_w, opacity_attenuation, strategy, p, size_out,
                       ExtGeom{has, has ? ext->tw : 0, has ? ext->th : 0, has ? ext->min_x : 0, has ? ext->min_y : 0});
}
template <bool DEFER_BIG>
RR_HD void plan_drop(const rr_drop& d, const rr_camera& cam, const Dims& dm, const int32_t* tex_h, const int32_t* tex_w,
                     double opacity_attenuation, int strategy, DropPlan& p, int64_t& size_out, const ExtGeom xg) {
  size_out = 0;
  p.status = RR_DROP_OK;
  p.tex = d.tex_index;
  p.flip = 0;
  p.bw0 = 1;
  p.nW = p.nH = 0;
  p.rs_mode = RS_AREA;
  p.isx = p.isy = 1;
  p.eh = 0;
  p.epitch = p.epad = 0;
  p.a0_off = p.a1_off = 0;
  p.ew = 0;
  p.scale_x = p.scale_y = p.inv_sx = p.inv_sy = 1.0;
  for (int i = 0; i < 9; i++) p.mi[i] = 0.0;
  for (int i = 0; i < 6; i++) p.ma[i] = 0.0;
  p.vis_x0 = p.vis_y0 = p.vis_w = p.vis_h = p.crop_x = p.crop_y = 0;
  p.tw = p.th = p.pw = p.ph = p.shift = p.r1 = p.r2 = 0;
  p.sig1 = p.sig2 = 0.0;
  const int W = dm.W, H = dm.H;
  const int sh = tex_h[d.tex_index], sw = tex_w[d.tex_index];

  int minCx, minCy;
  if (xg.has) {
    // the caller's tile (RainRenderer.add_drop_to_image's `drop` and `drop_minC`, bad_weather.py:336-338)
    p.kind = KIND_EXT;
    p.tw = imax(xg.tw, 1);
    p.th = imax(xg.th, 1);
    minCx = xg.min_x;
    minCy = xg.min_y;
  } else if (d.type == 0) {
    p.kind = KIND_BIG;
    double d0 = floor(d.iw1), d1 = floor(d.iw2);
    int minx = imax(imin(d.x0, d.x1), 0), miny = imax(imin(d.y0, d.y1), 0);
    double maxx = dmin(dmax((double)d.x0 + d0, (double)d.x1 + d1), (double)W);
    int maxy = imin(imax(d.y0, d.y1), H);
    int s0 = (int)(maxx - (double)minx), s1 = maxy - miny;
    p.tw = imax(s0, 1);
    p.th = imax(s1, 1);
    if (!DEFER_BIG) plan_big_homography(d, sw, sh, p.mi);
    int bh0 = imin(16, p.th);
    int bw0 = imin(1024 / bh0, p.tw);
    p.bw0 = bw0;
    minCx = minx;
    minCy = miny;
  } else {
    p.kind = KIND_ROT;
    p.flip = (d.x1 > W / 2) ? 1 : 0;
    p.th = imax(iabs(d.y1 - d.y0), 2);
    p.tw = imax(iabs(d.x1 - d.x0), d.max_width + 2);
    // imutils.rotate_bound geometry
    double cX = (double)sw / 2.0, cY = (double)sh / 2.0;
    double cxf = (double)(float)cX, cyf = (double)(float)cY;
    double al = d.rot_cos, be = d.rot_sin;
    double M[6] = {al, be, (1 - al) * cxf - be * cyf, -be, al, be * cxf + (1 - al) * cyf};
    double cs = fabs(M[0]), sn = fabs(M[1]);
    int nW = (int)((double)sh * sn + (double)sw * cs);
    int nH = (int)((double)sh * cs + (double)sw * sn);
    M[2] += ((double)nW / 2.0) - cX;
    M[5] += ((double)nH / 2.0) - cY;
    p.nW = nW;
    p.nH = nH;
    // cv::warpAffine's in-place inversion
    double D = M[0] * M[4] - M[1] * M[3];
    D = (D != 0.0) ? 1.0 / D : 0.0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11;
    M[1] = M[1] * (-D);
    M[3] = M[3] * (-D);
    M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1;
    M[5] = b2;
    for (int i = 0; i < 6; i++) p.ma[i] = M[i];
    if (nW <= 0 || nH <= 0) {
      p.status = RR_DROP_FOV_FAIL;   // cv2.resize of an empty image raises in the reference
      return;
    }
    p.inv_sx = (double)p.tw / (double)nW;
    p.inv_sy = (double)p.th / (double)nH;
    p.scale_x = 1.0 / p.inv_sx;
    p.scale_y = 1.0 / p.inv_sy;
    int isx = (int)cv_round(p.scale_x), isy = (int)cv_round(p.scale_y);
    const double EPS = 2.220446049250313e-16;
    bool fast = fabs(p.scale_x - (double)isx) < EPS && fabs(p.scale_y - (double)isy) < EPS;
    if (p.scale_x >= 1.0 && p.scale_y >= 1.0) {
      if (fast) { p.rs_mode = RS_AREA_FAST; p.isx = isx; p.isy = isy; }
      else p.rs_mode = RS_AREA;
    } else {
      p.rs_mode = RS_LINEAR;
    }
    minCx = d.x0;
    minCy = d.y0;
  }

  if (strategy == 1) {
    // rendering_strategy 'white' (bad_weather.py:349-353): no colour, no defocus, no clamp of the
    // tile origin -- the region is a plain numpy slice, negative starts wrap like Python's.
    p.pw = p.tw;
    p.ph = p.th;
    const int sx = py_slice_index(minCx, W), ex = py_slice_index(minCx + p.tw, W);
    const int sy = py_slice_index(minCy, H), ey = py_slice_index(minCy + p.th, H);
    p.vis_x0 = sx;
    p.vis_y0 = sy;
    p.vis_w = imax(ex - sx, 0);
    p.vis_h = imax(ey - sy, 0);
    p.tau_one = cam.exposure_s * 1.0;
    p.g = p.tau_one / cam.tau_zero;
    p.ew = p.tw;
    p.eh = p.th;
    size_out = (p.vis_w > 0 && p.vis_h > 0) ? (((int64_t)p.tw * p.th + 15) & ~15LL) : 0;
    return;
  }
  // circle of confusion (bad_weather.py:286-298,464-469)
  double o = fabs(d.wps[2]);
  double cc = ((o - cam.focus_plane) * cam.focal_sq) / (o * (cam.focus_plane - cam.focal_m) * cam.f_number);
  cc = fabs(cc / cam.sensor_px);
  if (!(cc < 1.7e308)) { p.status = RR_DROP_BAD_COC; return; }
  if (10.0 * cc >= (double)(RR_MAX_SHIFT + 1)) { p.status = RR_DROP_TOO_BIG; return; }
  p.shift = (int)(10.0 * cc);
  p.sig1 = cc;
  p.sig2 = cc / 2.0;
  p.r1 = (p.sig1 > 1e-15) ? (int)(4.0 * p.sig1 + 0.5) : 0;
  p.r2 = (p.sig2 > 1e-15) ? (int)(4.0 * p.sig2 + 0.5) : 0;
  p.pw = p.tw + 2 * p.shift;
  p.ph = p.th + 2 * p.shift;

  // placement and crop (bad_weather.py:418-422,429-441)
  int tmpx = minCx - p.shift, tmpy = minCy - p.shift;
  int mcx = imin(imax(tmpx, 0), W), mcy = imin(imax(tmpy, 0), H);
  int dx = mcx - tmpx, dy = mcy - tmpy;
  int cw, ch;
  if (dx < 0) { cw = imax(p.pw + dx, 0); p.crop_x = 0; } else { cw = imax(p.pw - dx, 0); p.crop_x = dx; }
  if (dy < 0) { ch = imax(p.ph + dy, 0); p.crop_y = 0; } else { ch = imax(p.ph - dy, 0); p.crop_y = dy; }
  p.vis_x0 = mcx;
  p.vis_y0 = mcy;
  p.vis_w = imax(imin(mcx + cw, W) - mcx, 0);
  p.vis_h = imax(imin(mcy + ch, H) - mcy, 0);

  // blend scalars (bad_weather.py:376,425-427)
  double d_avg = (d.iw1 + d.iw2) / 2.0;
  double length_opacity = opacity_attenuation * d_avg / ((double)d.length + d_avg);
  p.tau_one = cam.exposure_s * length_opacity;
  p.g = p.tau_one / cam.tau_zero;

  // The reference pads the tile by shift = int(10c) >= radius, but the truncated Gaussian only
  // reaches r1 rows / r2 columns beyond the raw tile: everything further out is EXACTLY zero, and
  // a zero alpha is a no-op in the blend and in the mask.  Only this effective tile is ever
  // materialised:  origin (shift - r2, shift - r1) inside the padded tile.
  p.ew = p.tw + 2 * p.r2;
  p.eh = p.th + 2 * p.r1;
  // The finished tile is stored densely (pitch = its width): most tiles are narrower than a 128-byte line, so
  // consecutive rows share lines and a padded, grid-aligned pitch would only add traffic (measured: +30 % compositor time).
  p.epad = 0;
  p.epitch = p.ew;
  // arena need (multiples of 16 doubles: every tile starts on a 128-byte line): the raw tile, plus the effective tile
  // when the drop is defocus-blurred
  size_out = (p.vis_w > 0 && p.vis_h > 0) ? (((int64_t)p.tw * p.th + 15) & ~15LL) + (p.r1 > 0 ? (((int64_t)p.epitch * p.eh + 15) & ~15LL) : 0) : 0;
}

// one output sample of the symmetric correlate1d (scipy ni_filters.c), zero extension.
// src: padded tile (pitch pw), hw: half-table hw[k] = w[k], k = 0..r (w[r] is the centre).
RR_HD double blur_axis0(const double* src, int pw, int ph, int x, int y, const double* hw, int r) {
  double acc = src[(int64_t)y * pw + x] * hw[r];
  for (int ii = -r; ii < 0; ii++) {
    int ya = y + ii, yb = y - ii;
    double va = (ya >= 0) ? src[(int64_t)ya * pw + x] : 0.0;
    double vb = (yb < ph) ? src[(int64_t)yb * pw + x] : 0.0;
    acc = acc + (va + vb) * hw[ii + r];
  }
  return acc;
}
RR_HD double blur_axis1(const double* src, int pw, int ph, int x, int y, const double* hw, int r) {
  (void)ph;
  const double* row = src + (int64_t)y * pw;
  double acc = row[x] * hw[r];
  for (int ii = -r; ii < 0; ii++) {
    int xa = x + ii, xb = x - ii;
    double va = (xa >= 0) ? row[xa] : 0.0;
    double vb = (xb < pw) ? row[xb] : 0.0;
    acc = acc + (va + vb) * hw[ii + r];
  }
  return acc;
}

// alpha-composite one drop sample into one pixel (bad_weather.py:443-446,450)
RR_HD void blend_pixel(double A, double tau_one, double exposure, double g, const double K[3], double bgr[3], double& mask) {
  double t = (A * tau_one) / exposure;
  double u = 1.0 - t;
  for (int c = 0; c < 3; c++) {
    double v = u * bgr[c] + (A * K[c]) * g;
    bgr[c] = clip01(v);
  }
  mask = mask + A;
}

// ---------------------------------------------------------------------------
// defocus blur: which kernel takes a drop and how its tile is cut into LDS-sized sub-tiles
// (tests/test_blur_layout.py sweeps these on the host: every sub-tile must fit the capacities)
// ---------------------------------------------------------------------------
// blur work layout (needed by k_colour to know where the finished tile will live)
// LDS capacities of the fused blur (doubles): input sub-tile incl. halo / result of the row (axis 0) pass.  They set how
// many workgroups share a CU (160 KB of LDS; the kernel is compiled for the matching register budget):
//   3 per CU: 3072 + 2048     4 per CU: 2816 + 2048 (default)     5 per CU: 2304 + 1600
// Smaller tiles mean more sub-tiles per drop (more halo re-loaded), more workgroups mean better latency hiding.
constexpr int BR_MAX = 48;          // largest axis-0 radius the fused kernel takes

struct BlurLayout {
  int fused;        // 0: two global passes (k_blur<0>, k_blur<1>)
  int wo, ho;       // output sub-tile
  int single;       // one sub-tile covers the padded tile: result written in place (A0)
};
// LDS footprints of an output sub-tile wo x ho: each thread filters four consecutive outputs
// along the filter axis with a rotating register window, so rows/columns are padded to
// multiples of four (zero / don't-care slack); the column-pass tile has an odd pitch so that
// lanes running down a column hit distinct LDS banks.
// X holds only the columns under the raw tile (tw wide): the others stay exactly zero through the row pass.
RR_HD int blur_x_doubles(int wo, int ho, int r1, int r2, int tw) { return imin(wo + 2 * r2, tw) * (((ho + 3) & ~3) + 2 * r1); }
RR_HD int blur_y_pitch(int wo, int r2) { return (((wo + 3) & ~3) + 2 * r2) | 1; }
RR_HD int blur_y_doubles(int wo, int ho, int r2) { return blur_y_pitch(wo, r2) * ((ho + 3) & ~3); }

RR_HD BlurLayout blur_layout(const DropPlan& p, int BX_MAX, int BY_MAX) {
  BlurLayout b{0, 0, 0, 0};
  if (p.r1 <= 0 || p.r1 > BR_MAX) return b;
  // Whole tile if it fits ...
  if (blur_x_doubles(p.ew, p.eh, p.r1, p.r2, p.tw) <= BX_MAX && blur_y_doubles(p.ew, p.eh, p.r2) <= BY_MAX) {
    b.fused = 1; b.wo = p.ew; b.ho = p.eh; b.single = 1;
    return b;
  }
  // ... else full-width bands of rows (no column is filtered twice by the row pass; only the 2*r1 halo rows are
  // loaded again per band), as tall as the two capacities allow ...
  {
    const int hy = (BY_MAX / blur_y_pitch(p.ew, p.r2)) & ~3;
    const int hx = ((BX_MAX / imin(p.ew + 2 * p.r2, p.tw)) - 2 * p.r1) & ~3;
    const int hb = imin(hy, hx);
    if (hb >= imax(8, p.r1)) {
      b.fused = 1; b.wo = p.ew; b.ho = imin(hb, p.eh);
      return b;
    }
  }
  // ... else (wide tiles with large radii) the output sub-tile that roughly maximises wo*ho under the two capacities --
  // halo-aware aspect: (wo+2*r2)*(ho+2*r1) <= BX_MAX.
  const double rr2 = (double)imax(p.r2, 1), rr1 = (double)p.r1;
  int wi = (int)sqrt((double)BX_MAX * rr2 / rr1);            // ideal haloed width
  wi = imax(imin(wi, p.ew + 2 * p.r2), 2 * p.r2 + 4);
  int hi = BX_MAX / wi;
  hi = imin(hi, ((p.eh + 3) & ~3) + 2 * p.r1);
  wi = imin(BX_MAX / hi, p.ew + 2 * p.r2);                   // give unused height back to the width
  int wo = wi - 2 * p.r2, ho = (hi - 2 * p.r1) & ~3;         // ho a multiple of four: no slack rows wasted
  if (ho > p.eh) ho = p.eh;
  for (int it = 0; it < 64 && wo >= 1 && ho >= 1; it++) {
    if (blur_x_doubles(wo, ho, p.r1, p.r2, p.tw) <= BX_MAX && blur_y_doubles(wo, ho, p.r2) <= BY_MAX) {
      b.fused = 1; b.wo = wo; b.ho = ho; b.single = (wo >= p.ew && ho >= p.eh) ? 1 : 0;
      return b;
    }
    if (blur_x_doubles(wo, ho, p.r1, p.r2, p.tw) > BX_MAX) { if (wo > 4) wo -= 1; else ho -= 4; }
    else ho -= 4;
  }
  return b;                                                  // halo alone exceeds the LDS: two-pass fallback
}
// small blurred tiles are filtered by one wave each, in place (k_blur_small)
constexpr int BS_X = 512;           // doubles per wave: data columns of the haloed input tile (tw x (eh + 2 r1))
constexpr int BS_Y = 768;           // doubles per wave: after the row pass (halo columns kept)

RR_HD bool blur_is_small(const DropPlan& p) {      // (rows padded to the four-output blocks of blur4)
  const int php = (p.eh + 3) & ~3;
  return p.r1 > 0 && p.r1 <= 31 && p.tw * (php + 2 * p.r1) <= BS_X && blur_y_pitch(p.ew, p.r2) * php <= BS_Y;
}


}  // namespace rr
