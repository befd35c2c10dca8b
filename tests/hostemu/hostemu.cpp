// tests/hostemu/hostemu.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the __host__ __device__ arithmetic of rain-rendering_amd/csrc/rr_device.h with
// g++ and drives it with plain loops that mirror the kernel chain of rainhip.hip, so the
// CPU-only test tier can check the kernel arithmetic against the numpy oracle bit for bit
// without a GPU.  Nothing in the product loads this library; it exists to find arithmetic
// bugs before spending GPU minutes.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rr_device.h"
#include "rr_prepass.h"
#include "rr_deflate.h"
#include "rr_pngrows.h"
#include "rr_particles.h"

using namespace rr;

extern "C" {

int emu_sizeof_plan(void) { return (int)sizeof(DropPlan); }

// plan + polygon for n drops.  poly: n*2*36 int32, npts: n, sizes: n
int emu_plan(const rr_drop* drops, int n, const rr_camera* cam, int H, int W, int He, int We, const int32_t* tex_h,
             const int32_t* tex_w, double opacity, DropPlan* plans, int32_t* poly, int32_t* npts, int64_t* sizes) {
  const int strategy = 0;
  Dims dm{H, W, He, We};
  for (int i = 0; i < n; i++) {
    int64_t size = 0;
    plan_drop(drops[i], *cam, dm, tex_h, tex_w, opacity, strategy, plans[i], size);
    int32_t* px = poly + (int64_t)i * 2 * 36;
    npts[i] = fov_polygon(drops[i], *cam, He, We, px, px + 36);
    if (plans[i].status != RR_DROP_OK || npts[i] == 0) size = 0;
    sizes[i] = size;
  }
  return 0;
}

// raw_tile_key (what k_dedup compares) of n planned drops: keys n*8 uint32
int emu_raw_tile_keys(const rr_drop* drops, int n, const DropPlan* plans, uint32_t* keys) {
  for (int i = 0; i < n; i++) raw_tile_key(drops[i], plans[i], keys + (int64_t)i * 8);
  return 0;
}

// the colour branch's default polygon: float32 vertices unless a status / wrap predicate is within its error bound of the
// threshold (then the float64 polygon).  poly: n*2*36 int32, npts: n, used32: n (1 = the float polygon was kept)
int emu_fov_auto(const rr_drop* drops, int n, const rr_camera* cam, int He, int We, int32_t* poly, int32_t* npts, int32_t* used32) {
  for (int i = 0; i < n; i++) {
    int32_t* px = poly + (int64_t)i * 2 * 36;
    int u = 0;
    npts[i] = fov_polygon_auto(drops[i], *cam, He, We, px, px + 36, &u);
    used32[i] = u;
  }
  return 0;
}

// the float vertices' error model: out[i] = max over the vertices of drop i of |azimuth32 - azimuth64| / az_err (the bound
// fov_vertex32 reports), -1 for drops either evaluation rejects.  What the unsure margins (8 x the bound) rest on.
int emu_fov_error_ratio(const rr_drop* drops, int n, const rr_camera* cam, int He, int We, double* out, double* max_px) {
  for (int i = 0; i < n; i++) {
    out[i] = -1.0;
    max_px[i] = 0.0;
    FovSetup F;
    FovSetup32 G;
    int unsure = 0;
    if (!fov_setup(drops[i], *cam, F)) continue;
    fov_setup32(drops[i], (float)cam->fov_cos, (float)cam->fov_sin, G, unsure);
    double worst = 0.0;
    bool okd = true;
    for (int k = 0; k < cam->n_fov; k++) {
      double az, ptx, pty;
      float az32, err, px32, py32;
      fov_vertex(F, *cam, cam->phi_cos[k], cam->phi_sin[k], He, We, az, ptx, pty);
      fov_vertex32(G, (float)cam->radius, (float)cam->phi_cos[k], (float)cam->phi_sin[k], He, We, az32, err, px32, py32, unsure);
      if (!(fabs(ptx) < 1e15) || !(fabs(pty) < 1e15)) { okd = false; break; }
      double d = fabs((double)az32 - az);
      if (d > 3.14159265358979) d = 6.283185307179586 - d;      // the two sides of the seam are neighbours
      worst = fmax(worst, d / (double)err);
      double dx = fabs((double)px32 - ptx);
      if (dx > 0.5 * We) dx = We - dx;
      max_px[i] = fmax(max_px[i], fmax(dx, fabs((double)py32 - pty)));
    }
    if (okd && !(unsure & 7)) out[i] = worst;
  }
  return 0;
}

// RR_OPT_FOV_FILL_RULE of the library: 1 (default since r06) OpenCV's rule where it applies (fov_rowspan_cv), 0 the span rule (fov_rowspan)
static int g_fill_rule = 1;
void emu_set_fill_rule(int rule) { g_fill_rule = rule; }
// spans of polygon (px, py)[n] under either rule: xl[y], xr[y] for y in [0, He), xl > xr = empty; returns 1 when OpenCV's rule applies
int emu_rowspans(const int32_t* px, const int32_t* py, int n, int n_fov, int He, int We, int rule, int32_t* xl, int32_t* xr) {
  const bool cv = rule == 1 && fov_fill_rule_cv_applies(px, py, n, n_fov, He, We);
  for (int y = 0; y < He; y++) {
    int a, b;
    const bool any = cv ? fov_rowspan_cv(px, py, n, y, We, a, b) : fov_rowspan(px, py, n, y, We, a, b);
    xl[y] = any ? a : 1;
    xr[y] = any ? b : 0;
  }
  return cv ? 1 : 0;
}

// k_fov_dda's row spans (rr_device.h DdaCursors) of polygon (px, py)[n] against the rule itself (rule 0: fov_rowspan, min / max
// over every edge that touches the row; rule 1: fov_rowspan_cv, OpenCV's FillConvexPoly in closed form): returns the number of
// rows of [0, He) on which they differ, -1 when the polygon's rows are not monotone (the kernel hands those to the
// edge-parallel kernel) or, for rule 1, when OpenCV's rule does not apply to it.
int emu_dda_check(const int32_t* px, const int32_t* py, int n, int He, int We, int rule) {
  if (poly_row_turns(py, n) > 2) return -1;
  if (rule == 1 && !fov_fill_rule_cv_applies(px, py, n, n, He, We)) return -1;
  int ktop = 0, ytop = py[0], ybot = py[0];
  for (int k = 1; k < n; k++) {
    if (py[k] < ytop) { ytop = py[k]; ktop = k; }
    if (py[k] > ybot) ybot = py[k];
  }
  std::vector<uint32_t> recs(4 * (size_t)n);
  for (int k = 0; k < n; k++) {                   // the record of edge {k, k + 1}, upper end first (k_fov_dda makes them before its row loop)
    const int j = k + 1 == n ? 0 : k + 1;
    const bool swp = py[j] < py[k];
    dda_edge_record(swp ? px[j] : px[k], swp ? py[j] : py[k], swp ? px[k] : px[j], swp ? py[k] : py[j], rule == 1, &recs[4 * (size_t)k]);
  }
  auto rec = [&](int k, uint32_t r[4]) { for (int q = 0; q < 4; q++) r[q] = recs[4 * (size_t)k + q]; };
  DdaCursors<decltype(rec)> cur;
  cur.init(rec, n, ktop, (uint32_t)px[ktop] | ((uint32_t)py[ktop] << 16));
  int bad = 0;
  for (int y = 0; y < He; y++) {
    int lo = 1 << 30, hi = -(1 << 30);
    if (y >= ytop && y <= ybot) cur.row(rec, y, lo, hi);
    const int a = imax(lo, 0), b = imin(hi, We - 1);
    int xl, xr;
    const bool any = rule == 1 ? fov_rowspan_cv(px, py, n, y, We, xl, xr) : fov_rowspan(px, py, n, y, We, xl, xr);
    if (any != (a <= b) || (any && (xl != a || xr != b))) bad++;
  }
  return bad;
}

// finished (padded, blurred) alpha tile of one drop into out[ph*pw]
int emu_tile(const DropPlan* p, const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off,
             double* out) {
  float ctab[128];
  build_cubic_tab(ctab);
  TexGlobal tx{texels + tex_off[p->tex], tex_h[p->tex], tex_w[p->tex]};
  const int n = p->pw * p->ph;
  std::vector<double> a0(n), a1(n);
  for (int idx = 0; idx < n; idx++) {
    int y = idx / p->pw, x = idx - y * p->pw;
    int rx = x - p->shift, ry = y - p->shift;
    double v = 0.0;
    if (rx >= 0 && rx < p->tw && ry >= 0 && ry < p->th) v = raw_tile_pixel(*p, tx, ctab, rx, ry);
    a0[idx] = v;
  }
  auto half = [](double sigma, int r, std::vector<double>& hw) {
    hw.resize(r + 1);
    for (int k = 0; k <= r; k++) hw[k] = gauss_phi(sigma, r - k);
    double tot = 0.0;
    for (int x = -r; x <= r; x++) tot = tot + hw[r - (x < 0 ? -x : x)];
    for (int k = 0; k <= r; k++) hw[k] = hw[k] / tot;
  };
  std::vector<double> hw;
  double* cur = a0.data();
  if (p->r1 > 0) {
    half(p->sig1, p->r1, hw);
    for (int idx = 0; idx < n; idx++) {
      int y = idx / p->pw, x = idx - y * p->pw;
      a1[idx] = blur_axis0(a0.data(), p->pw, p->ph, x, y, hw.data(), p->r1);
    }
    cur = a1.data();
    if (p->r2 > 0) {
      half(p->sig2, p->r2, hw);
      for (int idx = 0; idx < n; idx++) {
        int y = idx / p->pw, x = idx - y * p->pw;
        a0[idx] = blur_axis1(a1.data(), p->pw, p->ph, x, y, hw.data(), p->r2);
      }
      cur = a0.data();
    }
  }
  memcpy(out, cur, sizeof(double) * n);
  return 0;
}

// ---- the row-walk form of the rotate + INTER_AREA tile (k_tile_rows, rainhip.hip), lane by lane ----
// Same column table (coltab_cell_pass1 / 2), same walk rule, same passes and row segments (rows_pass_shape / rows_segment
// with `buf` doubles of cell sums; the vertical folds continue across passes through the carry) and the same 2 x 2 fetch
// from a pair texture as the kernel; the lanes of a wave run one after the other.
// out[th*tw] = the raw tile; also out_def[th*tw] = raw_tile_pixel (the definition).  Returns the number of pixels that
// differ in any bit, or -1 when the plan is not one k_tile_rows takes.
int emu_tile_rows(const DropPlan* pp, const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off, int buf,
                  double* out, double* out_def) {
  const DropPlan& p = *pp;
  const int sh = tex_h[p.tex], sw = tex_w[p.tex];
  if (!(p.kind == KIND_ROT && p.rs_mode == RS_AREA && p.scale_x >= 2.0 && p.tw >= 1 && p.tw <= 64 && p.th >= 1)) return -1;
  float ctab[128];
  build_cubic_tab(ctab);
  TexGlobal tx{texels + tex_off[p.tex], sh, sw};
  const uint8_t* g = texels + tex_off[p.tex];
  // pair texture (k_pair_textures)
  const int P = pair_pitch(sw);
  std::vector<uint16_t> pair((size_t)(sh + 3) * P);
  for (int k = 0; k < (sh + 3) * P; k++) {
    const int y = k / P - 2, x = k - (y + 2) * P - 2;
    const bool xin = x >= 0 && x < sw;
    const uint32_t lo = (xin && y >= 0 && y < sh) ? g[y * sw + x] : 0u, hi = (xin && y + 1 >= 0 && y + 1 < sh) ? g[(y + 1) * sw + x] : 0u;
    pair[k] = (uint16_t)(lo | (hi << 8));
  }
  double lut[256];
  for (int k = 0; k < 256; k++) lut[k] = (double)k / 255.0;
  auto sample = [&](int X, int Y) {
    const int sx = imin(imax(X >> 10, -2), sw), sy = imin(imax(Y >> 10, -2), sh);
    const int e = (sy + 2) * P + (sx + 2);
    const uint32_t u = (uint32_t)pair[e] | ((uint32_t)pair[e + 1] << 16);
    const double v00 = lut[u & 0xffu], v10 = lut[(u >> 8) & 0xffu], v01 = lut[(u >> 16) & 0xffu], v11 = lut[u >> 24];
    const int fx = (X >> 5) & 31, fy = (Y >> 5) & 31;
    const double ax_ = (double)(32 - fx), bx_ = (double)fx, ay_ = (double)(32 - fy), by_ = (double)fy;
    return ((v00 * (ay_ * ax_) + v01 * (ay_ * bx_)) + v10 * (by_ * ax_)) + v11 * (by_ * bx_);   // (the table's weights carry the / 1024)
  };
  const int tw = p.tw, th = p.th, nW = p.nW, nH = p.nH;
  std::vector<ColEnt> col((size_t)nW);
  std::vector<uint8_t> cell((size_t)nW, 0);
  for (int x = 0; x < nW; x++) col[x] = ColEnt{(int32_t)rot_adelta(p, x), (int32_t)rot_bdelta(p, x), 0u, 0u};
  uint16_t cfirst[64], clast[64];
  for (int d = 0; d < tw; d++) coltab_cell_pass1(p, d, col.data(), cell.data(), cfirst, clast);
  for (int d = 0; d < tw; d++) coltab_cell_pass2(p, d, col.data(), cell.data());
  std::vector<double> B((size_t)buf);
  const double sy_scale = p.scale_y;
  int R, NS;
  rows_pass_shape(tw, buf, R, NS);
  if ((R + 1) * tw > buf || R * NS > 64) return -2;
  const int colA = cfirst[0], colB = clast[tw - 1];
  double* const carry = B.data() + R * tw;
  for (int R0 = 0; R0 < nH; R0 += R) {
    const int R1 = imin(R0 + R, nH) - 1, nrows = R1 - R0 + 1;
    for (int k = 0; k < nrows * tw; k++) B[k] = 0.0;
    for (int lane = 0; lane < 64; lane++) {
      const int seg = (int)(((float)lane + 0.5f) / (float)R), r = lane - seg * R;
      if (!(r < nrows && seg < NS)) continue;
      const int c = R0 + r, ry = p.flip ? (nH - 1 - c) : c;
      const int X0 = (int)rot_X0(p, ry), Y0 = (int)rot_Y0(p, ry);
      // the interval of columns whose samples can be non-zero, found by brute force with a margin (the kernel's row_interval
      // is a conservative bound of the same thing; any superset gives the same sums)
      int xa = nW, xe = -1;
      for (int x = 0; x < nW; x++)
        if (sample(X0 + col[x].ad, Y0 + col[x].bd) != 0.0) { xa = imin(xa, x); xe = imax(xe, x); }
      xa = imax(xa - (c % 3), 0);                     // ragged margins: a walk may start and end anywhere outside the non-zero run
      xe = imin(xe + (c % 2), nW - 1);
      const int xs0 = imax(xa, colA), xe0 = imin(xe, colB);
      if (xs0 > xe0) continue;
      const int dlo = cell[xs0], dhi = imin(cell[xe0] + 1, tw - 1);
      int dA, dB;
      rows_segment(dlo, dhi, NS, seg, dA, dB);
      if (dA > dhi) continue;
      int xq = imax(xs0, (int)cfirst[dA]), left = imax(imin(xe0, (int)clast[dB - 1]) - xq + 1, 0);
      double* o = B.data() + r * tw;
      double* const lane_end = o + dB;
      if (left <= 0) continue;
      double b = 0.0;
      const int d0 = cell[xq];
      if (d0 < dA) {
        const ColEnt e = col[xq];
        b = sample(X0 + e.ad, Y0 + e.bd) * (double)bits_f32(e.w2);
        o += dA;
        xq++;
        left--;
      } else {
        o += d0;
      }
      for (; left > 0; left--, xq++) {
        const ColEnt e = col[xq];
        const double sv = sample(X0 + e.ad, Y0 + e.bd);
        b = b + sv * (double)bits_f32(e.w1 & 0x7fffffffu);
        if ((int32_t)e.w1 < 0) {
          if (o >= lane_end) return -3;
          if (*o != 0.0) return -5;                   // two lanes wrote one cell
          *o++ = b;
          b = sv * (double)bits_f32(e.w2);
        }
      }
      if (o < lane_end) {
        if (*o != 0.0) return -5;
        *o = b;
      }
    }
    const int dyG = imax((int)floor((double)R0 * p.inv_sy) - 1, 0);
    const int dyE = imin((int)floor((double)(R1 + 1) * p.inv_sy) + 2, th);
    // (the candidates must cover every destination row that reads a row of the pass)
    for (int dy = 0; dy < th; dy++) {
      const AreaSpan ay = area_span(nH, sy_scale, dy);
      int fr, lr;
      vfold_rows(ay, fr, lr);
      if (!(lr < R0 || fr > R1) && (dy < dyG || dy >= dyE)) return -4;
    }
    std::vector<double> carry_in(carry, carry + tw);       // lanes read the carry before any lane writes it
    const float inv_tw = 1.0f / (float)tw;
    for (int it = 0; it < (dyE - dyG) * tw; it++) {
      const int dq = (int)(((float)it + 0.5f) * inv_tw), dx = it - dq * tw, dy = dyG + dq;
      if (dx < 0 || dx >= tw) return -6;
      const AreaSpan ay = area_span(nH, sy_scale, dy);
      int fr, lr;
      vfold_rows(ay, fr, lr);
      if (lr < R0 || fr > R1) continue;
      double acc = 0.0;
      bool first = true;
      if (fr < R0) { acc = carry_in[dx]; first = false; }
      vfold_part(ay, R0, R1, acc, first, [&](int row) { return B[(row - R0) * tw + dx]; });
      if (lr <= R1) out[dy * tw + dx] = clip01(acc);
      else carry[dx] = acc;
    }
  }
  int bad = 0;
  for (int y = 0; y < th; y++)
    for (int x = 0; x < tw; x++) {
      out_def[y * tw + x] = raw_tile_pixel(p, tx, ctab, x, y);
      if (memcmp(&out_def[y * tw + x], &out[y * tw + x], 8) != 0) bad++;
    }
  return bad;
}

// whole frame, mirroring the kernel chain.  depth (may be NULL): the depth-occlusion OPTION (RR_OPT_DEPTH_OCCLUSION;
// definition: oracle/render.py _visible) -- H*W scene depth in metres, float32 or float64; a drop whose |world z| is
// greater than the scene depth at a pixel is neither blended nor added to the mask there.
int emu_render_frame_depth(int H, int W, int He, int We, const double* bg, const double* rainy_bg, const double* env,
                           const double* omega, const rr_drop* drops, int n, const rr_camera* cam, double opacity,
                           const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off,
                           uint8_t* rgb, double* comp_out, double* mask, int32_t* mask_i32, int32_t* status, double* Kout, int strategy,
                           const void* depth, int depth_f64) {
  Dims dm{H, W, He, We};
  // prefix table + frame constants
  std::vector<double> P((size_t)He * (We + 1) * 4, 0.0);
  double sumW = 0, sumY = 0;
  for (int r = 0; r < He; r++) {
    double run[4] = {0, 0, 0, 0};
    double* row = P.data() + (size_t)r * (We + 1) * 4;
    for (int c = 0; c < We; c++) {
      double w = omega[(size_t)r * We + c];
      const double* e = env + ((size_t)r * We + c) * 3;
      run[0] += e[0] * w; run[1] += e[1] * w; run[2] += e[2] * w; run[3] += w;
      for (int k = 0; k < 4; k++) row[(size_t)(c + 1) * 4 + k] = run[k];
    }
    sumY += run[2];
    sumW += run[3];
  }
  const double ambient = sumY / sumW;
  std::vector<DropPlan> plans(n);
  std::vector<CompRec> recs(n);
  std::vector<std::vector<double>> tiles(n);
  for (int i = 0; i < n; i++) {
    int64_t size = 0;
    DropPlan& p = plans[i];
    plan_drop(drops[i], *cam, dm, tex_h, tex_w, opacity, strategy, p, size);
    int32_t px[36], py[36];
    int np_ = fov_polygon(drops[i], *cam, He, We, px, py);
    if (strategy == 1) np_ = -1;            // 'white': the FOV is never consulted
    if (p.status != RR_DROP_OK || np_ == 0) size = 0;
    CompRec& rec = recs[i];
    memset(&rec, 0, sizeof(rec));
    rec.zdist = fabs(drops[i].wps[2]);
    int st = p.status;
    if (strategy == 1) {                    // 'white': no colour, no FOV dependency
      if (size > 0) {
        rec.K[0] = rec.K[1] = rec.K[2] = 1.0;
        rec.x0 = p.vis_x0; rec.y0 = p.vis_y0; rec.x1 = p.vis_x0 + p.vis_w; rec.y1 = p.vis_y0 + p.vis_h;
        rec.ox = -p.vis_x0; rec.oy = -p.vis_y0;
        rec.pitch = p.pw; rec.tau_one = p.tau_one; rec.g = p.g;
        tiles[i].resize((size_t)p.pw * p.ph);
        emu_tile(&p, texels, tex_h, tex_w, tex_off, tiles[i].data());
      }
    } else if (np_ == 0) st = RR_DROP_FOV_FAIL;
    if (strategy != 1 && np_ > 0) {
      int ymin = py[0], ymax = py[0];
      for (int k = 1; k < np_; k++) { ymin = imin(ymin, py[k]); ymax = imax(ymax, py[k]); }
      int ya = imax(ymin, 0), yb = imin(ymax, He - 1);
      double S[4] = {0, 0, 0, 0};
      bool any = false;
      const bool cv_rule = g_fill_rule == 1 && fov_fill_rule_cv_applies(px, py, np_, cam->n_fov, He, We);
      for (int y = ya; y <= yb; y++) {
        int xl, xr;
        if (cv_rule ? fov_rowspan_cv(px, py, np_, y, We, xl, xr) : fov_rowspan(px, py, np_, y, We, xl, xr)) {
          any = true;
          const double* row = P.data() + (size_t)y * (We + 1) * 4;
          for (int k = 0; k < 4; k++) S[k] += row[(size_t)(xr + 1) * 4 + k] - row[(size_t)xl * 4 + k];
        }
      }
      if (!any) st = RR_DROP_EMPTY_FOV;
      if (st == RR_DROP_OK && size > 0) {
        colour_from_sums(S, sumW, ambient, rec.K);
        rec.x0 = p.vis_x0; rec.y0 = p.vis_y0; rec.x1 = p.vis_x0 + p.vis_w; rec.y1 = p.vis_y0 + p.vis_h;
        rec.ox = p.crop_x - p.vis_x0; rec.oy = p.crop_y - p.vis_y0;
        rec.pitch = p.pw; rec.tau_one = p.tau_one; rec.g = p.g;
        tiles[i].resize((size_t)p.pw * p.ph);
        emu_tile(&p, texels, tex_h, tex_w, tex_off, tiles[i].data());
      }
    }
    if (status) status[i] = st;
    if (Kout) { Kout[i * 3] = rec.K[0]; Kout[i * 3 + 1] = rec.K[1]; Kout[i * 3 + 2] = rec.K[2]; }
  }
  // compositor, drop order per pixel
  for (size_t k = 0; k < (size_t)H * W * 3; k++) comp_out[k] = rainy_bg[k];
  for (size_t k = 0; k < (size_t)H * W; k++) mask[k] = 0.0;
  for (int i = 0; i < n; i++) {
    const CompRec& r = recs[i];
    for (int y = r.y0; y < r.y1; y++)
      for (int x = r.x0; x < r.x1; x++) {
        double A = tiles[i][(size_t)(y + r.oy) * r.pitch + (x + r.ox)];
        size_t pix = (size_t)y * W + x;
        if (depth) {
          const double scene = depth_f64 ? ((const double*)depth)[pix] : (double)((const float*)depth)[pix];
          if (r.zdist > scene) continue;
        }
        blend_pixel(A, r.tau_one, cam->exposure_s, r.g, r.K, comp_out + pix * 3, mask[pix]);
      }
  }
  double sc = 0, sb = 0;
  for (size_t k = 0; k < (size_t)H * W * 3; k++) { sc += comp_out[k]; sb += bg[k]; }
  const double cnt = (double)H * W * 3.0;
  const double diff = sc / cnt - sb / cnt;
  for (size_t pix = 0; pix < (size_t)H * W; pix++) {
    mask_i32[pix] = (int32_t)floor(mask[pix] * 255.0);
    for (int k = 0; k < 3; k++) {
      double v = clip01(comp_out[pix * 3 + (2 - k)] - diff);
      rgb[pix * 3 + k] = (uint8_t)(int)(v * 255.0);
    }
  }
  return 0;
}

int emu_render_frame(int H, int W, int He, int We, const double* bg, const double* rainy_bg, const double* env,
                     const double* omega, const rr_drop* drops, int n, const rr_camera* cam, double opacity,
                     const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w, const int64_t* tex_off,
                     uint8_t* rgb, double* comp_out, double* mask, int32_t* mask_i32, int32_t* status, double* Kout, int strategy) {
  return emu_render_frame_depth(H, W, He, We, bg, rainy_bg, env, omega, drops, n, cam, opacity, texels, tex_h, tex_w, tex_off, rgb,
                                comp_out, mask, mask_i32, status, Kout, strategy, nullptr, 0);
}

// Pre-pass of one frame (fog + environment map) with the per-pixel bodies of rr_prepass.h.
// rainy: H*W*3, env_xyY: H*We*3, env_u8: H*We*3 (We = cw + 2*(cw/2)); returns We or < 0.
// types: rrpre::PRE_* bits (element types of bg / rainy / env_xyY).  tiled: 0 = the three-kernel fog layer (k_fog_ext,
// k_fog_h, k_fog_v), 1 = FogTile<12> the way k_fog_tile drives it (column strips x segments of seg_rows rows, 256 thread
// roles per step, "barriers" = the ends of the loops over tid).
int emu_prepass(int H, int W, const void* bg, const void* depth, int depth_f64, double beta_ext, double beta_hg, double irr_num,
                double irr_den, int fog_k, const double* fog_w, int env_k, const double* env_w, int cw, int n_uniq,
                const int32_t* uniq, const int32_t* first, void* rainy, void* env_xyY, uint8_t* env_u8, int types, int tiled,
                int seg_rows) {
  using namespace rrpre;
  Kernels kn{};
  kn.fog_k = fog_k;
  kn.env_k = env_k;
  for (int i = 0; i < fog_k; i++) kn.fog_w[i] = fog_w[i];
  for (int i = 0; i < env_k; i++) kn.env_w[i] = env_w[i];
  std::vector<int32_t> src((size_t)H * cw), top(cw), bot(cw);
  if (!build_env_tables(H, W, cw, n_uniq, uniq, first, src.data(), top.data(), bot.data())) return -1;
  std::vector<uint8_t> need((size_t)H * (cw + 2 * (cw / 2)));
  build_env_need(H, cw, env_k / 2, src.data(), need.data());
  EnvGeom g{H, W, cw, cw / 2, cw + 2 * (cw / 2), src.data(), top.data(), bot.data(), tiled ? need.data() : nullptr};
  const size_t px = (size_t)H * W, ex = (size_t)H * g.We;
  std::vector<double> fext(px), tmpF(px), tmpL(px * 3), mean(3), etmp(ex * 3, -1.0e300);     // (a sum read without having been made would show)
  std::vector<uint32_t> epack(ex);
  std::vector<uint8_t> r8(px * 3);
  PreScratch sc{fext.data(), tmpF.data(), tmpL.data(), r8.data(), nullptr, mean.data(), epack.data(), etmp.data()};
  PreFrame F{bg, depth, rainy, env_xyY, env_u8, r8.data(), beta_ext, beta_hg, irr_num, irr_den, depth_f64, types};
  for (int c = 0; c < 3; c++) {         // k_fog_sum / k_fog_mean (sum order differs from the device's tree: ~1e-16)
    double s = 0;
    for (size_t p = 0; p < px; p++) s += (irr_num * load_unit(bg, types, (int64_t)(p * 3 + c))) / irr_den;
    mean[c] = s / (double)px;
  }
  if (tiled) {
    if (fog_k != 25 || seg_rows <= 0 || seg_rows % 8) return -2;
    using T = FogTile<12>;
    std::vector<double> S(T::RB * 4 * T::PITCH), ring(T::RING * 4 * T::TC);
    double k3[3];
    for (int c = 0; c < 3; c++) k3[c] = beta_hg * mean[c];
    for (int ys = 0; ys < H; ys += seg_rows)
      for (int x0 = 0; x0 < W; x0 += T::TC) {
        const int ye = ys + seg_rows < H ? ys + seg_rows : H, hs = ys - 12;
        const int n_iter = (ye - ys + T::RB - 1) / T::RB + T::NB - 1;
        for (int k = 0; k < n_iter; k++) {
          for (int i = 0; i < T::RB * T::PITCH; i++) {
            const int64_t p = T::stage_src(H, W, x0, hs, k, i);
            T::stage_put(F, k3, i, depth_f64 ? ((const double*)depth)[p] : 0.0, depth_f64 ? 0.0f : depth_f32_at(F, p), S.data());
          }
          for (int tid = 0; tid < 256; tid++) T::hpass(kn, depth_f64, k, tid, S.data(), ring.data());
          const int m = k - (T::NB - 1);
          if (m < 0) continue;
          double a[256][T::VR];
          for (int tid = 0; tid < 256; tid++) T::vtaps(kn, depth_f64, m, tid, ring.data(), a[tid]);
          for (int tid = 0; tid < 256; tid++) T::vstore(F, H, W, x0, ys, ye, m, tid, a[tid], a[(tid & ~63) | (tid & 15)]);
        }
      }
  } else {
    for (size_t p = 0; p < px; p++) fog_ext_px(F, 0, H, W, sc, (int64_t)p);
    {                                     // k_fog_h: staged row segments (LDS on the device)
      const int half = fog_k / 2, pitch = FOG_SEG + KMAX - 1;
      std::vector<double> S(4 * pitch);
      for (int y = 0; y < H; y++)
        for (int x0 = 0; x0 < W; x0 += FOG_SEG) {
          const int nseg = (W - x0 < FOG_SEG ? W - x0 : FOG_SEG);
          for (int i = 0; i < nseg + 2 * half; i++) fog_stage_px(F, 0, H, W, kn, sc, y, x0, i, S.data(), pitch);
          for (int t = 0; t < nseg; t++) {
            double o[4];
            fog_h_taps(S.data(), pitch, half + t, kn, depth_f64, o);
            fog_h_store(0, H, W, sc, y, x0 + t, o);
          }
        }
    }
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) fog_v_px(F, 0, H, W, kn, sc, y, x);
  }
  for (int r = 0; r < H; r++)
    for (int x = 0; x < g.We; x++) env_build_px(F, 0, g, sc, r, x);
  for (int r = 0; r < H; r++)
    for (int x = 0; x < g.We; x++) env_h_px(0, g, kn, sc, r, x);
  for (int r = 0; r < H; r++)
    for (int x = 0; x < g.We; x++) env_v_px(F, 0, g, kn, sc, r, x);
  return g.We;
}

// The blur work split of one drop (rr_device.h: blur_is_small, blur_layout).  out = {small, fused, wo, ho}; returns the
// number of sub-tiles whose LDS footprint -- computed the way k_blur_fused / k_blur_small index their tiles --
// exceeds a capacity (must be 0).
int emu_blur_layout(int ew, int eh, int r1, int r2, int tw, int th, int bx, int by, int32_t* out) {
  DropPlan p{};
  p.ew = ew; p.eh = eh; p.r1 = r1; p.r2 = r2; p.tw = tw; p.th = th;
  const bool small = blur_is_small(p);
  const BlurLayout L = blur_layout(p, bx, by);
  out[0] = small; out[1] = L.fused; out[2] = L.wo; out[3] = L.ho;
  int bad = 0;
  if (small) {                                     // k_blur_small: X = tw x (php + 2 r1), Y = pitch x php
    const int php = (eh + 3) & ~3;
    if (tw * (php + 2 * r1) > BS_X || blur_y_pitch(ew, r2) * php > BS_Y || r1 > 63 || r2 > 63) bad++;
  } else if (L.fused) {                            // k_blur_fused: per sub-tile X = wd x (hop + 2 r1), Y = pitch x hop
    if (L.wo < 1 || L.ho < 1 || L.wo > 0xffff || L.ho > 0x7fff) return 1 << 30;
    const int ntx = (ew + L.wo - 1) / L.wo, nty = (eh + L.ho - 1) / L.ho;
    for (int sty = 0; sty < nty; sty++)
      for (int stx = 0; stx < ntx; stx++) {
        const int y0 = sty * L.ho, x0 = stx * L.wo;
        const int ho = imin(L.ho, eh - y0), wo = imin(L.wo, ew - x0), hop = (ho + 3) & ~3;
        const int wi = wo + 2 * r2, hi = hop + 2 * r1, yp = blur_y_pitch(wo, r2);
        const int xa = imax(0, 2 * r2 - x0), xb = imin(wi, tw + 2 * r2 - x0), wd = imax(xb - xa, 0);
        if (wd * hi > bx || yp * hop > by || ((yp * hop) & 1) || xa + wd > wi) bad++;
        // the column pass reads columns up to 4 * (ceil(wo / 4) - 1) + 2 r2 + 3 of a row of pitch yp
        if (4 * (((wo + 3) >> 2) - 1) + 2 * r2 + 3 >= yp) bad++;
      }
  }
  return bad;
}

// The compositor's short blend divides by the launch's exposure through its reciprocal: q0 = a * y, q = fma(fma(-q0, d, a), y, q0)
// with y = 1 / d (k_composite).  Returns how many of n pseudo-random numerators (products alpha * tau like the kernel's,
// and raw bit patterns over 400 binades) give a result that differs from a / d -- Markstein's theorem says none.
int64_t emu_reciprocal_division_mismatches(double d, int64_t n, uint64_t seed) {
  const double y = 1.0 / d;
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  int64_t bad = 0;
  for (int64_t i = 0; i < n; i++) {
    const uint64_t r = rnd();
    double a;
    if ((i & 3) == 0) {
      const uint64_t e = 1023 - 200 + (r >> 52) % 400, bits = (e << 52) | (r & 0xFFFFFFFFFFFFFull);
      memcpy(&a, &bits, 8);
    } else {
      const double A = (double)(r >> 11) * (1.0 / 9007199254740992.0);
      const double tau = (double)(rnd() >> 11) * (50.0 / 9007199254740992.0);
      a = A * tau;
    }
    const double q0 = a * y;
    const double q = fma(fma(-q0, d, a), y, q0);
    if (q != a / d) bad++;
  }
  return bad;
}


// One frame of the particle generator + packer (rr_particles.h under plain loops: what k_particles / k_particle_draws do
// on the device), incl. the renderer's per-drop draws from numpy's legacy MT19937 stream (restated here; the product's
// host version is rr_host.cpp, the device's k_particle_draws).  Also returns the raw particles (n_particles records of
// 15 doubles: wp1 wp2 wd ip1 ip2 iw1 iw2) when `particles` is not NULL.
int emu_generate_drops(const rr_sim_frame* sf, int H, int W, const double* dgrid, const double* cdf_tabs, int n_grid,
                       const double* ratio_db, rr_drop* out, int cap, int32_t* n_out, double* particles) {
  const double* cdf = cdf_tabs + (size_t)sf->table * n_grid;
  int n = 0;
  for (int i = 0; i < sf->n_particles; i++) {
    rrsim::Particle p;
    rrsim::make_particle(*sf, dgrid, cdf, n_grid, (uint32_t)i, p);
    if (particles) {
      double* q = particles + (size_t)i * 15;
      for (int k = 0; k < 3; k++) { q[k] = p.wp1[k]; q[3 + k] = p.wp2[k]; }
      q[6] = p.wd; q[7] = p.ip1[0]; q[8] = p.ip1[1]; q[9] = p.ip2[0]; q[10] = p.ip2[1]; q[11] = p.iw1; q[12] = p.iw2;
    }
    rr_drop d;
    double ratio;
    if (!rrsim::derive_drop(p, sf->render_scale, W, H, d, ratio)) continue;
    d.tex_index = 10 * rrsim::texture_bucket(ratio, ratio_db);
    if (n < cap) out[n] = d;
    n++;
  }
  *n_out = n;
  // MT19937, numpy's legacy seeding; randint(lo, lo + 10) by masked rejection; legacy gauss' consumption
  uint32_t key[624];
  uint32_t seed = sf->draw_seed;
  for (int pos = 0; pos < 624; pos++) { key[pos] = seed; seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u; }
  int pos = 624;
  auto u32 = [&]() -> uint32_t {
    if (pos == 624) {
      for (int kk = 0; kk < 624; kk++) {
        const uint32_t y = (key[kk] & 0x80000000u) | (key[(kk + 1) % 624] & 0x7fffffffu);
        key[kk] = key[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      pos = 0;
    }
    uint32_t y = key[pos++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
  };
  auto dbl = [&]() -> double {
    const int32_t a = (int32_t)(u32() >> 5), b = (int32_t)(u32() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  };
  bool has_gauss = false;
  const int m = n < cap ? n : cap;
  for (int k = 0; k < m; k++) {
    uint32_t v;
    do { v = u32() & 15u; } while (v > 9u);
    out[k].tex_index += (int32_t)v;
    if (out[k].type != 0) {
      if (has_gauss) has_gauss = false;
      else {
        double r2;
        do {
          const double x1 = 2.0 * dbl() - 1.0, x2 = 2.0 * dbl() - 1.0;
          r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        has_gauss = true;
      }
    }
  }
  return 0;
}


// The device's PNG entropy coder (rr_deflate.h: k_pngz_blocks + k_pngz_pack) on the host, thread roles in loops.  rows: n bytes of
// filtered scanlines; dst: n bytes.  Returns the length of the zlib stream that now lies behind the 16-byte header in dst, or 0
// if header + stream do not fit n bytes (dst is then a copy of rows, as the device leaves the scanlines in place).
int64_t emu_pngz(const uint8_t* rows, int64_t n, uint8_t* dst) {
  using namespace rrz;
  const int nb = (int)blocks_of(n);
  std::vector<BlockMeta> meta(nb);
  std::vector<uint8_t> slots((size_t)nb * SLOT_BYTES);
  std::vector<BlockState> st(1);
  BlockState& S = st[0];
  for (int k = 0; k < nb; k++) {
    const int len = (int)(n - (int64_t)k * BLOCK < BLOCK ? n - (int64_t)k * BLOCK : BLOCK), last = k == nb - 1;
    memcpy(S.in, rows + (int64_t)k * BLOCK, len);
#define ALL(call) for (int tid = 0; tid < NT; tid++) call
    // the lanes' tokens of chunk c of wave `wave` (k_pngz_blocks: a shuffle for the predecessor, a ballot for the starts)
    auto chunk = [&](int wave, int c, Tok* tok, int* bs, int* idx, bool* valid) {
      const int w0 = wave * WAVE_BYTES, nv = len - (w0 + c * CHUNK) < CHUNK ? len - (w0 + c * CHUNK) : CHUNK;
      uint64_t start = 0;
      for (int lane = 0; lane < 64; lane++) {
        idx[lane] = w0 + c * CHUNK + lane;
        valid[lane] = lane < nv;
        bs[lane] = valid[lane] ? S.in[idx[lane]] : 0;
        if (valid[lane] && (lane == 0 || bs[lane] != bs[lane - 1])) start |= 1ull << lane;
      }
      for (int lane = 0; lane < 64; lane++) tok[lane] = lane_token(lane, nv, start, bs[lane]);
    };
    auto chunks_of = [&](int wave) {
      const int w0 = wave * WAVE_BYTES;
      return len > w0 ? ((len - w0 < WAVE_BYTES ? len - w0 : WAVE_BYTES) + CHUNK - 1) / CHUNK : 0;
    };
    Tok tok[64];
    int bs[64], idx[64];
    bool valid[64];
    ALL(p0_init(S, tid, len, last));
    for (int wave = 0; wave < NW; wave++) {
      uint32_t s1[64] = {0}, s2[64] = {0};
      for (int c = 0; c < chunks_of(wave); c++) {
        chunk(wave, c, tok, bs, idx, valid);
        for (int lane = 0; lane < 64; lane++) p1_lane(S, wave, tok[lane], bs[lane], len - idx[lane], valid[lane], s1[lane], s2[lane]);
      }
      uint32_t a1 = 0, a2 = 0;                            // (the wave's sums: a DPP reduction on the device)
      for (int lane = 0; lane < 64; lane++) {
        a1 += s1[lane];
        a2 += s2[lane] % 65521u;
      }
      S.ad1[wave] = a1;
      S.ad2[wave] = a2;
    }
    ALL(p1_sum(S, tid));
    ALL(p2_rank(S, tid));
    ALL(p3_tree(S, tid));
    ALL(p4_depth(S, tid));
    ALL(p5_limit(S, tid));
    ALL(p6_assign(S, tid));
    ALL(p7_codes(S, tid));
    ALL(p8_header(S, tid));
    ALL(p9_wave_bits(S, tid));
    ALL(p10_decide(S, tid));
    ALL(p11_clear(S, tid));
    ALL(p12_ends(S, tid));
    if (!S.stored)
      for (int wave = 0; wave < NW; wave++) {
        uint32_t base = wave_base(S, wave);
        for (int c = 0; c < chunks_of(wave); c++) {
          chunk(wave, c, tok, bs, idx, valid);
          for (int lane = 0; lane < 64; lane++) {
            uint32_t nbits;
            const uint32_t code = token_code(S, tok[lane], nbits);
            if (nbits) or_bits(S.out, base, code, nbits);
            base += nbits;
          }
        }
      }
    ALL(p12b_stored_bytes(S, tid));
    ALL(p13_meta(S, tid, &meta[k]));
#undef ALL
    memcpy(slots.data() + (size_t)k * SLOT_BYTES, S.out, S.bytes);
  }
  memcpy(dst, rows, (size_t)n);
  const int64_t total = stream_bytes(meta.data(), nb);
  if (PNGZ_HEADER + total > n) return 0;
  for (int k = 0; k < nb; k++) memcpy(dst + block_offset(meta.data(), k), slots.data() + (size_t)k * SLOT_BYTES, meta[k].bytes);
  pack_ends(dst, meta.data(), nb);
  return total;
}


// The scanline filters of an input file reversed with the rule k_png_unfilter applies (rr_pngrows.h), pixel by pixel in raster
// order.  rows: H rows of 1 + bpp * W bytes; out: bpp 3 -> H*W*3 bytes B G R, bpp 2 -> H*W uint16.  Returns 0, or -1 for a bad
// filter type.
int emu_png_unfilter(const uint8_t* rows, int H, int W, int bpp, uint8_t* out) {
  const size_t rb = 1 + (size_t)bpp * W;
  std::vector<uint8_t> img((size_t)H * W * bpp);
  for (int y = 0; y < H; y++) {
    const int ft = rows[rb * y];
    if (ft > 4) return -1;
    for (int x = 0; x < W; x++)
      for (int c = 0; c < bpp; c++) {
        const size_t i = ((size_t)y * W + x) * bpp + c;
        const int left = x > 0 ? img[i - bpp] : 0, up = y > 0 ? img[i - (size_t)W * bpp] : 0, upl = (x > 0 && y > 0) ? img[i - (size_t)W * bpp - bpp] : 0;
        img[i] = (uint8_t)rrrows::png_unfilter_byte(ft, rows[rb * y + 1 + (size_t)x * bpp + c], left, up, upl);
      }
  }
  for (size_t px = 0; px < (size_t)H * W; px++) {
    if (bpp == 3) {
      out[px * 3] = img[px * 3 + 2];
      out[px * 3 + 1] = img[px * 3 + 1];
      out[px * 3 + 2] = img[px * 3];
    } else {
      reinterpret_cast<uint16_t*>(out)[px] = (uint16_t)((img[px * 2] << 8) | img[px * 2 + 1]);
    }
  }
  return 0;
}

}  // extern "C"
