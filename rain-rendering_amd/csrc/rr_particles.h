// rr_particles.h -- the rain-particle generator and the drop-table packer, per particle (SURVEY 8f "next" #4;
// BASELINE.json configs[4]: "in-kernel particle simulation (no XML)").
//
// `__host__ __device__` like rr_device.h: k_particles (rainhip.hip) runs these functions on gfx950, tests/hostemu
// compiles them with g++, and rain-rendering_amd/tools/particles.py states the same arithmetic in numpy (its bit-exact
// host statement; the model itself is documented there).  IEEE double, the evaluation order spelled out, no FMA
// contraction, + - * / sqrt rint only (sqrt is correctly rounded on gfx950 and in numpy); the one transcendental -- exp
// in the terminal velocity -- is rr::det_exp.
//
//   reference side: tools/simulation.py drives a closed-source simulator with the settings of common/db.py:41-70;
//   its XML (bad_weather.py:192-211) passes through the loader's derived fields (bad_weather.py:208-241) and the frame
//   filter of Generator.run (generator.py:413-420).  The last two are followed to the letter below.
#pragma once
#include "rr_device.h"

namespace rrsim {

using rr::det_exp;

// ---- Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> four 32-bit words ----
RR_HD void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += W0;
    k1 += W1;
  }
}
// a 32-bit word as a number strictly inside (0, 1): (w + 1/2) / 2^32, exact in double
RR_HD double unit32(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

struct Particle {                 // what one <streak .../> element of the simulator's XML carries (bad_weather.py:192-211)
  double wp1[3], wp2[3];          // world position at the start / end of the exposure, camera at the origin looking along -z
  double wd;                      // diameter, metres
  double ip1[2], ip2[2];          // sensor position, pixels, origin bottom-left
  double iw1, iw2;                // image width, pixels
};

// terminal velocity of a drop of diameter d_mm, m/s (Atlas, Srivastava & Sekhon 1973)
RR_HD double terminal_velocity(double d_mm) { return 9.65 - 10.3 * det_exp(-0.6 * d_mm); }

// inverse-CDF sample of the diameter table: largest j with cdf[j] <= u (j <= n - 2), linear inside the cell
RR_HD double sample_diameter(const double* dgrid, const double* cdf, int n, double u) {
  int lo = 0, hi = n - 1;                       // invariant: cdf[lo] <= u (cdf[0] = 0 < u), answer in [lo, hi)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid; else hi = mid;
  }
  const double slope = (dgrid[lo + 1] - dgrid[lo]) / (cdf[lo + 1] - cdf[lo]);
  return dgrid[lo] + (u - cdf[lo]) * slope;
}

// particle i of frame sf: three Philox blocks, counter = (i, frame, block, 0)
RR_HD void make_particle(const rr_sim_frame& sf, const double* dgrid, const double* cdf, int n_grid, uint32_t i, Particle& p) {
  uint32_t a[4] = {i, sf.frame, 0u, 0u}, b[4] = {i, sf.frame, 1u, 0u}, c[4] = {i, sf.frame, 2u, 0u};
  philox4x32_10(a, sf.key0, sf.key1);
  philox4x32_10(b, sf.key0, sf.key1);
  philox4x32_10(c, sf.key0, sf.key1);
  const double W = (double)sf.sensor_w, H = (double)sf.sensor_h;
  // diameter (mm) from the table; only drops that can show at least min_px wide are simulated: depth <= z_max(D)
  const double D = sample_diameter(dgrid, cdf, n_grid, unit32(a[0]));
  const double wd = D * 1e-3;
  const double z_max = rr::dmin((wd * sf.fpx) / sf.min_px, sf.z_far);
  // uniform in the frustum's volume: the depth's density is 3 z^2 / z_max^3, the law of the largest of three uniforms
  const double u1 = unit32(a[1]), u2 = unit32(a[2]), u3 = unit32(a[3]);
  const double depth = rr::dmax(z_max * rr::dmax(rr::dmax(u1, u2), u3), 0.05);
  // position on the (slightly enlarged) sensor, from the bottom-left corner
  const double lo_x = -sf.margin * W, hi_x = (1.0 + sf.margin) * W, lo_y = -sf.margin * H, hi_y = (1.0 + sf.margin) * H;
  const double px = lo_x + (hi_x - lo_x) * unit32(b[0]);
  const double py = lo_y + (hi_y - lo_y) * unit32(b[1]);
  const double X = ((px - W / 2.0) * depth) / sf.fpx;
  const double Y = ((py - H / 2.0) * depth) / sf.fpx;
  const double Z = -depth;
  // horizontal wind: a centred sum of four uniforms scaled to unit variance (bell-shaped, bounded), times wind_sigma
  const double s4 = ((unit32(c[0]) + unit32(c[1])) + (unit32(c[2]) + unit32(c[3]))) - 2.0;
  const double wind = (s4 * 1.7320508075688772) * sf.wind_sigma;
  const double t = sf.exposure_s;
  const double X2 = X + wind * t;
  const double Y2 = Y - terminal_velocity(D) * t;
  const double Z2 = Z + sf.speed_mps * t;
  const double depth2 = rr::dmax(-Z2, 0.05);
  p.wp1[0] = X; p.wp1[1] = Y; p.wp1[2] = Z;
  p.wp2[0] = X2; p.wp2[1] = Y2; p.wp2[2] = Z2;
  p.wd = wd;
  p.ip1[0] = px; p.ip1[1] = py;
  p.ip2[0] = W / 2.0 + (sf.fpx * X2) / depth2;
  p.ip2[1] = H / 2.0 + (sf.fpx * Y2) / depth2;
  p.iw1 = (wd * sf.fpx) / depth;
  p.iw2 = (wd * sf.fpx) / depth2;
}

// ceil(sqrt(n)) of a non-negative integer, exactly (np.ceil(np.sqrt(.)) of the loader gives the same: a non-integer root
// is further from an integer than the rounding error of a correctly rounded sqrt)
RR_HD int64_t ceil_sqrt(int64_t n) {
  int64_t s = (int64_t)sqrt((double)n);
  while (s * s < n) s++;
  while (s > 0 && (s - 1) * (s - 1) >= n) s--;
  return s;
}

// The loader's derived fields (DBManager.load_streaks_from_xml, bad_weather.py:208-241) and the frame filter
// (Generator.run, generator.py:413-420) for one particle; W x H is the RENDERED frame (sensor / render_scale).  Fills every
// field of `d` except tex_index; `ratio` is Streak.ratio (take_drop_texture's key).  Returns whether the streak is kept.
RR_HD bool derive_drop(const Particle& p, int render_scale, int W, int H, rr_drop& d, double& ratio) {
  const double rs = (double)render_scale;
  double sx = p.ip1[0] / rs, sy = p.ip1[1] / rs, ex = p.ip2[0] / rs, ey = p.ip2[1] / rs;
  const double w1 = p.iw1 / rs, w2 = p.iw2 / rs;
  sy = (double)H - sy;                                    // bad_weather.py:221-222
  ey = (double)H - ey;
  const double d0 = fabs(sx - ex), d1 = fabs(sy - ey);
  const double mwf = rr::dmax(w1, w2);                    // max(iw1, iw2): Python's max keeps the first on ties / NaN
  const int64_t max_width = (int64_t)mwf;                 // int(): truncation
  const double nrm = sqrt(d0 * d0 + d1 * d1);
  const double dir2y = -(d1 / nrm);
  const double cos_theta = (d0 / nrm) * 0.0 + dir2y * -1.0;
  const double actual_length = d1 / cos_theta;
  ratio = (double)max_width / actual_length;
  const int64_t x0 = (int64_t)rint(sx), y0 = (int64_t)rint(sy), x1 = (int64_t)rint(ex), y1 = (int64_t)rint(ey);   // round half to even
  const int64_t ddx = x0 - x1, ddy = y0 - y1;
  const int64_t length = ceil_sqrt(ddx * ddx + ddy * ddy);
  const int type = max_width >= 4 ? 0 : (max_width > 1 ? 1 : 2);
  d.x0 = (int32_t)x0; d.y0 = (int32_t)y0; d.x1 = (int32_t)x1; d.y1 = (int32_t)y1;
  d.max_width = (int32_t)max_width;
  d.length = (int32_t)length;
  d.type = type;
  d.tex_index = 0;
  d.iw1 = w1;
  d.iw2 = w2;
  d.wps[0] = p.wp1[0]; d.wps[1] = p.wp1[1]; d.wps[2] = p.wp1[2] * -1.0;      // bad_weather.py:223-224
  d.wpe[0] = p.wp2[0]; d.wpe[1] = p.wp2[1]; d.wpe[2] = p.wp2[2] * -1.0;
  // rotation of the streak texture (non-Big): cos / sin of -(theta) with theta = acos(-dy / n) (generator.py:138-145,163)
  // evaluated exactly: cos = -dy / n, sin = -|dx| / n, n = |start - end| over the integer positions
  if (type == 0) {
    d.rot_cos = 1.0;
    d.rot_sin = 0.0;
  } else {
    const double fx = (double)ddx, fy = (double)ddy;
    const double n1 = sqrt(fx * fx + fy * fy);
    d.rot_cos = (fx / n1) * 0.0 + (fy / n1) * -1.0;
    d.rot_sin = -(fabs(fx) / n1);
  }
  const int64_t m = H > W ? H : W;
  const bool in_s = 0 <= x0 && x0 < W && 0 <= y0 && y0 < H, in_e = 0 <= x1 && x1 < W && 0 <= y1 && y1 < H;
  return max_width >= 1 && length >= 1 &&                                     // the loader's own test (bad_weather.py:238)
         max_width < m && length < m && (in_s || in_e);                        // generator.py:413-420
}

// the block of ten textures take_drop_texture draws from (bad_weather.py:250-265): NaN falls through to the last one
RR_HD int texture_bucket(double ratio, const double* ratio_db) {
  int b = 4;
  for (int k = 3; k >= 0; k--)
    if (ratio < ratio_db[k]) b = k;
  return b;
}

}  // namespace rrsim
