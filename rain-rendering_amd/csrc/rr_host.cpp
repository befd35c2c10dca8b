// rr_host.cpp -- host-only entry points of librainhip.so (no device code).
//
// rr_host_drop_draws: the per-frame random draws of the reference's drop loop, reproduced
// bit for bit from numpy's legacy RandomState so that the driver can prepare frames on
// worker threads (the reference draws from the process-global generator inside its Python
// per-drop loop):
//   np.random.seed(frame_index)                              common/generator.py:318
//   per drop: np.random.randint(10*b, 10*b+10)               common/bad_weather.py:252-264
//             np.random.normal(0, noise_std)  (non-Big only) common/generator.py:136
// Algorithms (numpy/random: _mt19937.pyx _legacy_seeding, mt19937.c, distributions.c
// buffered_bounded_masked_uint32, legacy-distributions.c legacy_gauss): MT19937 seeded with
// Knuth's LCG; randint by masked rejection on one 32-bit word per attempt; normal by the polar
// Box-Muller with the second deviate cached.  Checked against numpy itself in
// tests/test_host_logic.py.
#include <cmath>
#include <cstdint>

#include "rainhip.h"
#include "rr_parallel.h"

namespace {

struct MT {
  uint32_t key[624];
  int pos;
  int has_gauss;
  double gauss;
};

void mt_seed(MT& s, uint32_t seed) {
  for (int pos = 0; pos < 624; pos++) {
    s.key[pos] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
  }
  s.pos = 624;
  s.has_gauss = 0;
  s.gauss = 0.0;
}

void mt_gen(MT& s) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  uint32_t y;
  int kk;
  for (kk = 0; kk < 624 - 397; kk++) {
    y = (s.key[kk] & UPPER) | (s.key[kk + 1] & LOWER);
    s.key[kk] = s.key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
  }
  for (; kk < 623; kk++) {
    y = (s.key[kk] & UPPER) | (s.key[kk + 1] & LOWER);
    s.key[kk] = s.key[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
  }
  y = (s.key[623] & UPPER) | (s.key[0] & LOWER);
  s.key[623] = s.key[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
  s.pos = 0;
}

inline uint32_t mt_u32(MT& s) {
  if (s.pos == 624) mt_gen(s);
  uint32_t y = s.key[s.pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

inline double mt_double(MT& s) {
  const int32_t a = (int32_t)(mt_u32(s) >> 5), b = (int32_t)(mt_u32(s) >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

// RandomState.randint(low, high) for a range that fits 32 bits (masked rejection)
inline int64_t mt_randint(MT& s, int64_t low, int64_t high) {
  const uint32_t rng = (uint32_t)(high - 1 - low);
  if (rng == 0) return low;
  if (rng == 0xffffffffu) return low + (int64_t)mt_u32(s);
  uint32_t mask = rng;
  mask |= mask >> 1;
  mask |= mask >> 2;
  mask |= mask >> 4;
  mask |= mask >> 8;
  mask |= mask >> 16;
  uint32_t v;
  while ((v = (mt_u32(s) & mask)) > rng) {
  }
  return low + (int64_t)v;
}

inline double mt_gauss(MT& s) {
  if (s.has_gauss) {
    const double t = s.gauss;
    s.gauss = 0.0;
    s.has_gauss = 0;
    return t;
  }
  double f, x1, x2, r2;
  do {
    x1 = 2.0 * mt_double(s) - 1.0;
    x2 = 2.0 * mt_double(s) - 1.0;
    r2 = x1 * x1 + x2 * x2;
  } while (r2 >= 1.0 || r2 == 0.0);
  f = std::sqrt(-2.0 * std::log(r2) / r2);
  s.gauss = f * x1;
  s.has_gauss = 1;
  return f * x2;
}

}  // namespace

extern "C" int rr_host_drop_draws(uint32_t seed, int32_t n, const int32_t* tex_lo, const uint8_t* is_big, double noise_std,
                                  int32_t* tex_index, double* noise) {
  if (n < 0 || (n > 0 && (!tex_lo || !is_big || !tex_index || !noise))) return RR_E_ARG;
  MT s;
  mt_seed(s, seed);
  for (int k = 0; k < n; k++) {
    tex_index[k] = (int32_t)mt_randint(s, tex_lo[k], (int64_t)tex_lo[k] + 10);
    noise[k] = is_big[k] ? 0.0 : 0.0 + noise_std * mt_gauss(s);
  }
  return RR_OK;
}

// ---------------------------------------------------------------------------------------------------
// rr_host_frame_draws / rr_host_assemble_drops: one frame's drop table off the Python interpreter.
//   step 1  the frame filter of Generator.run (common/generator.py:413-420), the texture bucket of
//           take_drop_texture (common/bad_weather.py:250-265) and the frame's random draws (above);
//   (the caller's numpy evaluates the rotation terms: acos / cos / sin stay with numpy so that their bits do)
//   step 2  rr_drop records from the table columns.
// Both run without the GIL when called through ctypes: the driver's I/O threads scale.
// ---------------------------------------------------------------------------------------------------
extern "C" int64_t rr_host_frame_draws(const rr_streak_table* t, int32_t W, int32_t H, const double* ratio_db, int32_t n_ratio,
                                       uint32_t seed, double noise_std, int64_t* keep, int32_t* tex_index, double* noise) {
  if (!t || t->n < 0 || !ratio_db || n_ratio < 4 || !keep || !tex_index || !noise) return RR_E_ARG;
  const int64_t m = H > W ? H : W;
  MT s;
  mt_seed(s, seed);
  int64_t nk = 0;
  for (int64_t i = 0; i < t->n; i++) {
    const int64_t sx = t->ips[2 * i], sy = t->ips[2 * i + 1], ex = t->ipe[2 * i], ey = t->ipe[2 * i + 1];
    const bool in_s = 0 <= sx && sx < W && 0 <= sy && sy < H, in_e = 0 <= ex && ex < W && 0 <= ey && ey < H;
    if (!(1 <= t->max_width[i] && t->max_width[i] < m && 1 <= t->length[i] && t->length[i] < m && (in_s || in_e))) continue;
    const double r = t->ratio[i];
    int b = 4;                                               // NaN falls through to the last block, like the reference's else
    for (int k = 3; k >= 0; k--)
      if (r < ratio_db[k]) b = k;
    keep[nk] = i;
    tex_index[nk] = (int32_t)mt_randint(s, 10 * b, 10 * b + 10);
    noise[nk] = t->type[i] == 0 ? 0.0 : 0.0 + noise_std * mt_gauss(s);
    nk++;
  }
  return nk;
}

extern "C" int rr_host_assemble_drops(const rr_streak_table* t, int64_t n_keep, const int64_t* keep, const int32_t* tex_index,
                                      const double* rot_cos, const double* rot_sin, rr_drop* out) {
  if (!t || n_keep < 0 || (n_keep > 0 && (!keep || !tex_index || !rot_cos || !rot_sin || !out))) return RR_E_ARG;
  for (int64_t k = 0; k < n_keep; k++) {
    const int64_t i = keep[k];
    if (i < 0 || i >= t->n) return RR_E_ARG;
    rr_drop& d = out[k];
    d.x0 = (int32_t)t->ips[2 * i];
    d.y0 = (int32_t)t->ips[2 * i + 1];
    d.x1 = (int32_t)t->ipe[2 * i];
    d.y1 = (int32_t)t->ipe[2 * i + 1];
    d.max_width = (int32_t)t->max_width[i];
    d.length = (int32_t)t->length[i];
    d.type = t->type[i];
    d.tex_index = tex_index[k];
    d.iw1 = t->iw1[i];
    d.iw2 = t->iw2[i];
    for (int c = 0; c < 3; c++) {
      d.wps[c] = t->wps[3 * i + c];
      d.wpe[c] = t->wpe[3 * i + c];
    }
    const bool big = t->type[i] == 0;
    d.rot_cos = big ? 1.0 : rot_cos[k];
    d.rot_sin = big ? 0.0 : rot_sin[k];
  }
  return RR_OK;
}

// One frame without angular noise in ONE call: filter + draws + records.  Without noise the rotation terms of a streak
// depend on its table entry alone, so the caller evaluates them once per simulated frame (numpy: rot_cos / rot_sin per
// table ENTRY) and every rendered frame that uses the table gathers them.  Returns the number of kept streaks; records
// beyond `cap` are not written.
static int64_t pack_frame_quiet(const rr_streak_table* t, int32_t W, int32_t H, const double* ratio_db, int32_t n_ratio, uint32_t seed,
                                const double* rot_cos, const double* rot_sin, rr_drop* out, int64_t cap) {
  if (!t || t->n < 0 || !ratio_db || n_ratio < 4 || !rot_cos || !rot_sin || (cap > 0 && !out) || cap < 0) return RR_E_ARG;
  const int64_t m = H > W ? H : W;
  MT s;
  mt_seed(s, seed);
  int64_t nk = 0;
  for (int64_t i = 0; i < t->n; i++) {
    const int64_t sx = t->ips[2 * i], sy = t->ips[2 * i + 1], ex = t->ipe[2 * i], ey = t->ipe[2 * i + 1];
    const bool in_s = 0 <= sx && sx < W && 0 <= sy && sy < H, in_e = 0 <= ex && ex < W && 0 <= ey && ey < H;
    if (!(1 <= t->max_width[i] && t->max_width[i] < m && 1 <= t->length[i] && t->length[i] < m && (in_s || in_e))) continue;
    const double r = t->ratio[i];
    int b = 4;
    for (int k = 3; k >= 0; k--)
      if (r < ratio_db[k]) b = k;
    const int32_t tex = (int32_t)mt_randint(s, 10 * b, 10 * b + 10);
    const bool big = t->type[i] == 0;
    if (!big) (void)mt_gauss(s);                       // np.random.normal(0, 0) still draws (generator.py:136)
    if (nk < cap) {
      rr_drop& d = out[nk];
      d.x0 = (int32_t)sx;
      d.y0 = (int32_t)sy;
      d.x1 = (int32_t)ex;
      d.y1 = (int32_t)ey;
      d.max_width = (int32_t)t->max_width[i];
      d.length = (int32_t)t->length[i];
      d.type = t->type[i];
      d.tex_index = tex;
      d.iw1 = t->iw1[i];
      d.iw2 = t->iw2[i];
      for (int c = 0; c < 3; c++) {
        d.wps[c] = t->wps[3 * i + c];
        d.wpe[c] = t->wpe[3 * i + c];
      }
      d.rot_cos = big ? 1.0 : rot_cos[i];
      d.rot_sin = big ? 0.0 : rot_sin[i];
    }
    nk++;
  }
  return nk;
}

extern "C" int rr_host_pack_frames(int32_t n, const rr_streak_table* const* tables, const double* const* rot_cos, const double* const* rot_sin,
                                   int32_t W, int32_t H, const double* ratio_db, int32_t n_ratio, const uint32_t* seeds, rr_drop* out,
                                   int64_t out_stride, int64_t cap, int32_t threads, int64_t* counts) {
  if (n < 0 || !counts || (n > 0 && (!tables || !rot_cos || !rot_sin || !seeds)) || out_stride < cap) return RR_E_ARG;
  rrpar::parallel_for(n, threads, [&](int k) {
    counts[k] = tables[k] ? pack_frame_quiet(tables[k], W, H, ratio_db, n_ratio, seeds[k], rot_cos[k], rot_sin[k],
                                             out ? out + (size_t)k * (size_t)out_stride : nullptr, cap)
                          : (int64_t)RR_E_ARG;
  });
  return RR_OK;
}

extern "C" int rr_sizeof_streak_table(void) { return (int)sizeof(rr_streak_table); }

// ---------------------------------------------------------------------------------------------------
// rr_host_parse_particles: the particles XML of the rain simulator (schema read by the reference's
// DBManager.load_streaks_from_xml, common/bad_weather.py:192-211) -> flat records.
//   root element: any name; its child elements are frames (attributes id, t, d, rs; tag names ignored);
//   their child elements are drops (attributes pid, wp1, wp2 = "(x;y;z)", wd1, wd2, ip1, ip2 = "(x;y)", iw1, iw2).
// Numbers are converted with strtod / strtoll: the same correctly rounded doubles Python's float() gives.
// Only the raw attribute values are returned; everything derived (render scale, y flip, z sign, widths,
// ratios, rounding, the pid dictionary) stays in the loader that calls this.
// ---------------------------------------------------------------------------------------------------
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Attr {
  const char* name;
  size_t nlen;
  const char* val;
  size_t vlen;
};

bool attr_is(const Attr& a, const char* n) { return a.nlen == strlen(n) && memcmp(a.name, n, a.nlen) == 0; }

// The file buffer is NUL-terminated, so strtoll / strtod run in place and stop at the field's delimiter.
// whole-token integer like Python's int(str): optional surrounding blanks, optional sign, decimal digits
bool parse_i64(const char* s, size_t n, int64_t& out) {
  errno = 0;
  char* end = nullptr;
  long long v = strtoll(s, &end, 10);
  if (end == s || errno || end > s + n) return false;
  while (end < s + n && (*end == ' ' || *end == '\t' || *end == '\n' || *end == '\r')) end++;
  if (end != s + n) return false;
  out = (int64_t)v;
  return true;
}

bool parse_f64(const char* s, size_t n, double& out) {
  char* end = nullptr;
  out = strtod(s, &end);
  if (end == s || end > s + n) return false;
  while (end < s + n && (*end == ' ' || *end == '\t' || *end == '\n' || *end == '\r')) end++;
  return end == s + n;
}

// "(a;b;c)"[1:-1].split(';') -> k doubles (bad_weather.py:202-207)
bool parse_vec(const char* s, size_t n, double* out, int k) {
  if (n < 2) return false;
  const char* p = s + 1;
  const char* e = s + n - 1;
  for (int i = 0; i < k; i++) {
    const char* q = (const char*)memchr(p, ';', (size_t)(e - p));
    const char* stop = q ? q : e;
    if ((i < k - 1) != (q != nullptr)) return false;     // exactly k fields
    if (!parse_f64(p, (size_t)(stop - p), out[i])) return false;
    p = stop + 1;
  }
  return true;
}

}  // namespace

extern "C" int rr_host_parse_particles(const char* path, rr_particle_frame* frames, int64_t cap_frames, rr_particle* drops,
                                       int64_t cap_drops, int64_t* n_frames, int64_t* n_drops) {
  if (!path || !n_frames || !n_drops || cap_frames < 0 || cap_drops < 0 || (cap_frames > 0 && !frames) || (cap_drops > 0 && !drops))
    return RR_E_ARG;
  *n_frames = *n_drops = 0;
  FILE* fh = fopen(path, "rb");
  if (!fh) return RR_E_ARG;
  std::vector<char> buf;
  {
    fseek(fh, 0, SEEK_END);
    long sz = ftell(fh);
    fseek(fh, 0, SEEK_SET);
    if (sz < 0) { fclose(fh); return RR_E_ARG; }
    buf.resize((size_t)sz + 1);
    size_t got = fread(buf.data(), 1, (size_t)sz, fh);
    fclose(fh);
    buf[got] = 0;
    buf.resize(got + 1);
  }
  const char* p = buf.data();
  const char* end = p + buf.size() - 1;
  int depth = 0;
  int64_t nf = 0, nd = 0;
  std::vector<Attr> attrs;
  while (p < end) {
    const char* lt = (const char*)memchr(p, '<', (size_t)(end - p));
    if (!lt) break;
    p = lt + 1;
    if (p >= end) return RR_E_PARSE;
    if (*p == '?') {                                         // <? ... ?>
      const char* q = strstr(p, "?>");
      if (!q) return RR_E_PARSE;
      p = q + 2;
      continue;
    }
    if (*p == '!') {
      if (end - p >= 3 && p[1] == '-' && p[2] == '-') {      // <!-- ... -->
        const char* q = strstr(p, "-->");
        if (!q) return RR_E_PARSE;
        p = q + 3;
        continue;
      }
      return RR_E_UNSUPPORTED;                               // DOCTYPE / CDATA: leave it to a full XML parser
    }
    if (*p == '/') {                                         // </name>
      const char* q = (const char*)memchr(p, '>', (size_t)(end - p));
      if (!q || depth <= 0) return RR_E_PARSE;
      depth--;
      p = q + 1;
      continue;
    }
    // element: name, attributes, '>' or '/>'
    while (p < end && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r' && *p != '>' && *p != '/') p++;
    attrs.clear();
    bool self_close = false;
    for (;;) {
      while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
      if (p >= end) return RR_E_PARSE;
      if (*p == '>') { p++; break; }
      if (*p == '/') {
        if (p + 1 >= end || p[1] != '>') return RR_E_PARSE;
        self_close = true;
        p += 2;
        break;
      }
      Attr a;
      a.name = p;
      while (p < end && *p != '=' && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r' && *p != '>') p++;
      a.nlen = (size_t)(p - a.name);
      while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
      if (p >= end || *p != '=') return RR_E_PARSE;
      p++;
      while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
      if (p >= end || (*p != '"' && *p != '\'')) return RR_E_PARSE;
      const char quote = *p++;
      a.val = p;
      const char* q = (const char*)memchr(p, quote, (size_t)(end - p));
      if (!q) return RR_E_PARSE;
      a.vlen = (size_t)(q - p);
      if (memchr(a.val, '&', a.vlen)) return RR_E_UNSUPPORTED;      // entity references: full XML parser
      p = q + 1;
      attrs.push_back(a);
    }
    const int level = depth;                                  // 0: root, 1: frame, 2: drop, deeper: ignored
    if (!self_close) depth++;
    if (level == 1) {
      rr_particle_frame fr;
      memset(&fr, 0, sizeof(fr));
      int have = 0;
      for (const Attr& a : attrs) {
        int64_t* dst = attr_is(a, "id") ? &fr.id : attr_is(a, "t") ? &fr.t : attr_is(a, "d") ? &fr.d : attr_is(a, "rs") ? &fr.rs : nullptr;
        if (!dst) continue;
        if (!parse_i64(a.val, a.vlen, *dst)) return RR_E_PARSE;
        have |= attr_is(a, "id") ? 1 : attr_is(a, "t") ? 2 : attr_is(a, "d") ? 4 : 8;
      }
      if (have != 15) return RR_E_PARSE;                      // a missing attribute is a KeyError in the reference
      fr.first_drop = nd;
      fr.n_drops = 0;
      if (nf < cap_frames) frames[nf] = fr;
      nf++;
    } else if (level == 2) {
      rr_particle d;
      memset(&d, 0, sizeof(d));
      int have = 0;
      for (const Attr& a : attrs) {
        bool ok = true;
        if (attr_is(a, "pid")) { ok = parse_i64(a.val, a.vlen, d.pid); have |= 1; }
        else if (attr_is(a, "wp1")) { ok = parse_vec(a.val, a.vlen, d.wp1, 3); have |= 2; }
        else if (attr_is(a, "wp2")) { ok = parse_vec(a.val, a.vlen, d.wp2, 3); have |= 4; }
        else if (attr_is(a, "wd1")) { ok = parse_f64(a.val, a.vlen, d.wd1); have |= 8; }
        else if (attr_is(a, "wd2")) { ok = parse_f64(a.val, a.vlen, d.wd2); have |= 16; }
        else if (attr_is(a, "ip1")) { ok = parse_vec(a.val, a.vlen, d.ip1, 2); have |= 32; }
        else if (attr_is(a, "ip2")) { ok = parse_vec(a.val, a.vlen, d.ip2, 2); have |= 64; }
        else if (attr_is(a, "iw1")) { ok = parse_f64(a.val, a.vlen, d.iw1); have |= 128; }
        else if (attr_is(a, "iw2")) { ok = parse_f64(a.val, a.vlen, d.iw2); have |= 256; }
        if (!ok) return RR_E_PARSE;
      }
      if (have != 511) return RR_E_PARSE;
      if (nd < cap_drops) drops[nd] = d;
      if (nf >= 1 && nf <= cap_frames) frames[nf - 1].n_drops++;
      nd++;
    }
  }
  if (depth != 0) return RR_E_PARSE;
  *n_frames = nf;
  *n_drops = nd;
  return RR_OK;
}

extern "C" int rr_sizeof_particle(void) { return (int)sizeof(rr_particle); }
extern "C" int rr_sizeof_particle_frame(void) { return (int)sizeof(rr_particle_frame); }
