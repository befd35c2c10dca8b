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
