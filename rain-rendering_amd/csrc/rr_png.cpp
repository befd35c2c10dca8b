// rr_png.cpp -- host-only PNG codec of librainhip.so for the driver's I/O threads.
//
// The reference reads frames with cv2.imread and writes them with plt.imsave (common/generator.py:352,360,466-467).
// A Python decoder / encoder holds the interpreter lock for part of every file, which serialises the driver's I/O
// threads at a few hundred frames per second; these entry points are called through ctypes (no lock held):
//   rr_png_info / rr_png_read_bgr8 / rr_png_read_gray16   what cv2.imread(path) / cv2.imread(path, IMREAD_UNCHANGED)
//                                                          return for the PNGs of the datasets (8-bit colour, 16-bit depth)
//   rr_png_write_scanlines                                 an RGBA PNG from the Sub-filtered scanlines the library's
//                                                          rr_frame_out.rainy_png / mask_png deliver (zlib deflate + framing)
// Non-interlaced files with colour types gray / RGB / palette / RGBA are decoded; anything else returns
// RR_E_UNSUPPORTED and the caller uses its general-purpose decoder.  zlib does the (de)compression.
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "rainhip.h"
#include "rr_parallel.h"

namespace {

// Working memory of the codec, reused from file to file and from call to call.  A frame's files need buffers of megabytes
// (the file, its IDAT stream, the scanlines, the pixels, the deflate output); allocated per file they are mmap'ed and
// unmapped every time, and with a dozen threads doing that at once the process spends more time in the kernel (address-space
// lock, page faults, TLB shoot-downs) than in the codec (measured: 7 ms of system time per frame beside 6 ms of work).
// Buffers only grow; a lease takes a set from the pool (or makes one) and gives it back.
template <class T>
class Pool {
 public:
  class Lease {
   public:
    explicit Lease(Pool& p) : pool_(p), obj_(p.take()) {}
    ~Lease() { pool_.give(obj_); }
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
    T& operator*() { return *obj_; }
    T* operator->() { return obj_; }

   private:
    Pool& pool_;
    T* obj_;
  };

 private:
  T* take() {
    {
      std::lock_guard<std::mutex> g(m_);
      if (!free_.empty()) {
        T* o = free_.back();
        free_.pop_back();
        return o;
      }
    }
    return new T();
  }
  void give(T* o) {
    {
      std::lock_guard<std::mutex> g(m_);
      if (free_.size() < 64) {                          // (more than any sane number of I/O threads: the rest is freed)
        free_.push_back(o);
        return;
      }
    }
    delete o;
  }
  std::mutex m_;
  std::vector<T*> free_;
};
// grow-only view of a vector: at least n elements, old contents kept, nothing re-initialised once it is large enough
template <class V>
inline void at_least(V& v, size_t n) {
  if (v.size() < n) v.resize(n + n / 8);
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Png {
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  void reset() {                                      // (capacity kept)
    w = h = 0;
    depth = ctype = interlace = 0;
    idat.clear();
    plte.clear();
  }
};

// the file's bytes in buf[0 .. size) (buf itself may be longer: it is reused)
int read_file(const char* path, std::vector<uint8_t>& buf, size_t& size) {
  size = 0;
  FILE* fh = fopen(path, "rb");
  if (!fh) return RR_E_ARG;
  fseek(fh, 0, SEEK_END);
  long sz = ftell(fh);
  fseek(fh, 0, SEEK_SET);
  if (sz < 0) { fclose(fh); return RR_E_ARG; }
  at_least(buf, (size_t)sz);
  size_t got = fread(buf.data(), 1, (size_t)sz, fh);
  fclose(fh);
  size = (size_t)sz;
  return got == (size_t)sz ? RR_OK : RR_E_PARSE;
}

int parse_chunks(const std::vector<uint8_t>& fbuf, size_t fsize, Png& p, bool want_data) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  struct Span {                                         // (the first fsize bytes of the reused buffer)
    const uint8_t* d;
    size_t n;
    size_t size() const { return n; }
    const uint8_t* data() const { return d; }
    const uint8_t& operator[](size_t i) const { return d[i]; }
  } f{fbuf.data(), fsize};
  p.reset();
  if (f.size() < 8 + 25 || memcmp(f.data(), sig, 8) != 0) return RR_E_PARSE;
  size_t pos = 8;
  bool have_hdr = false;
  while (pos + 12 <= f.size()) {
    const uint32_t len = be32(&f[pos]);
    const uint8_t* tag = &f[pos + 4];
    if (pos + 12 + (size_t)len > f.size()) return RR_E_PARSE;
    const uint8_t* data = &f[pos + 8];
    if (!memcmp(tag, "IHDR", 4)) {
      if (len != 13) return RR_E_PARSE;
      p.w = be32(data);
      p.h = be32(data + 4);
      p.depth = data[8];
      p.ctype = data[9];
      p.interlace = data[12];
      have_hdr = true;
      if (!want_data) return RR_OK;
    } else if (!memcmp(tag, "PLTE", 4)) {
      p.plte.assign(data, data + len);
    } else if (!memcmp(tag, "IDAT", 4)) {
      p.idat.insert(p.idat.end(), data, data + len);
    } else if (!memcmp(tag, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  return have_hdr ? RR_OK : RR_E_PARSE;
}

// ---------------------------------------------------------------------------------------------------------------------
// The two checksums of the format at memory speed.  A frame's files carry ~6 MB through Adler-32 (every zlib stream read
// or written) and ~1.7 MB through CRC-32 (the IDAT chunks written); zlib's own loops move 3 and 1 GB/s on the driver's
// hosts -- together 10 % of an I/O thread's time per frame.  Chosen once, by the CPU's feature flags; zlib's functions
// otherwise.  (tests/test_host_logic.py compares both with zlib on every length and alignment.)
// ---------------------------------------------------------------------------------------------------------------------
__attribute__((target("ssse3"))) uint32_t adler32_ssse3(uint32_t adler, const uint8_t* p, size_t n) {
  uint64_t a = adler & 0xffffu, b = adler >> 16;
  const __m128i weights = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
  const __m128i ones = _mm_set1_epi16(1), zero = _mm_setzero_si128();
  while (n >= 16) {
    // b grows by a for every byte: over a stretch of chunks, 16 * (a at each chunk's start) + the position-weighted bytes.
    // 5552 bytes keep every 32-bit lane below 2^32 (zlib's NMAX)
    const size_t blk = (n < 5552 ? n : 5552) & ~(size_t)15;
    const uint64_t chunks = blk / 16;
    __m128i va = zero, vprev = zero, vb = zero;
    for (size_t k = 0; k < blk; k += 16) {
      const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + k));
      vprev = _mm_add_epi32(vprev, va);
      va = _mm_add_epi32(va, _mm_sad_epu8(d, zero));
      vb = _mm_add_epi32(vb, _mm_madd_epi16(_mm_maddubs_epi16(d, weights), ones));
    }
    auto hsum = [](__m128i v) {
      alignas(16) uint32_t t[4];
      _mm_store_si128(reinterpret_cast<__m128i*>(t), v);
      return (uint64_t)t[0] + t[1] + t[2] + t[3];
    };
    b = (b + 16 * (a * chunks + hsum(vprev)) + hsum(vb)) % 65521u;
    a = (a + hsum(va)) % 65521u;
    p += blk;
    n -= blk;
  }
  for (; n; n--) {
    a += *p++;
    b += a;
  }
  return (uint32_t)(((b % 65521u) << 16) | (a % 65521u));
}

// CRC-32 (the reflected 0xEDB88320 polynomial) by carry-less multiplication: four 128-bit lanes folded 64 bytes at a time,
// then 128 -> 64 -> 32 bits with a Barrett reduction (Gopal et al., "Fast CRC computation for generic polynomials using
// PCLMULQDQ", Intel 2009; the constants are x^k mod P for the fold distances).  `crc` and the result are the register's
// inner value (the caller applies the format's initial / final inversion); n >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_fold_pclmul(uint32_t crc, const uint8_t* buf, size_t n) {
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
  auto ld = [](const void* q) { return _mm_loadu_si128(reinterpret_cast<const __m128i*>(q)); };
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = ld(buf);
  x2 = ld(buf + 16);
  x3 = ld(buf + 32);
  x4 = ld(buf + 48);
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k1k2));
  buf += 64;
  n -= 64;
  while (n >= 64) {
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
    x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
    x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = ld(buf);
    y6 = ld(buf + 16);
    y7 = ld(buf + 32);
    y8 = ld(buf + 48);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    buf += 64;
    n -= 64;
  }
  x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k3k4));                // four lanes into one
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (n >= 16) {
    x2 = ld(buf);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    buf += 16;
    n -= 16;
  }
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                                     // 128 -> 64 bits
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(k5k0));
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(poly));                 // Barrett: 64 -> 32 bits
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

struct CpuFeatures {
  bool ssse3, pclmul;
  CpuFeatures() {
    __builtin_cpu_init();
    ssse3 = __builtin_cpu_supports("ssse3");
    pclmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  }
};
const CpuFeatures& cpu() {
  static const CpuFeatures f;
  return f;
}

// zlib's adler32(1, p, n) continued from `adler`
uint32_t fast_adler32(uint32_t adler, const uint8_t* p, size_t n) {
  if (cpu().ssse3) return adler32_ssse3(adler, p, n);
  while (n > 0) {                                    // (uInt lengths)
    const size_t part = n < (1u << 30) ? n : (1u << 30);
    adler = (uint32_t)adler32(adler, p, (uInt)part);
    p += part;
    n -= part;
  }
  return adler;
}
// zlib's crc32(crc, p, n)
uint32_t fast_crc32(uint32_t crc, const uint8_t* p, size_t n) {
  if (cpu().pclmul && n >= 64) {
    const size_t body = n & ~(size_t)15;
    crc = ~crc32_fold_pclmul(~crc, p, body);
    p += body;
    n -= body;
  }
  while (n > 0) {
    const size_t part = n < (1u << 30) ? n : (1u << 30);
    crc = (uint32_t)crc32(crc, p, (uInt)part);
    p += part;
    n -= part;
  }
  return crc;
}

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0; }

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast inflate of a complete zlib stream into a buffer of known size (the reader always knows it: h * (stride + 1)).
// zlib's inflate moves ~190 MB/s here; the image of a frame (1.4 MB of scanlines) costs 7-8 ms of the I/O thread that
// reads it.  This decoder keeps 64 bits of input in a register, resolves a symbol with one look-up in a 12-bit (literal /
// length; two literals per entry where both codes fit) or 8-bit (distance) table (longer codes: one more look-up in a
// sub-table) and copies matches eight bytes at a time.  It accepts every valid stream (stored, fixed and dynamic blocks); anything it cannot vouch for -- a malformed
// header or code, output or input that does not end where it must, a wrong Adler-32 -- makes it return false, and the
// caller hands the stream to zlib.  `in` must be readable for 16 bytes past n (the caller pads), `out` for 16 past out_len.
// ---------------------------------------------------------------------------------------------------------------------
namespace inflate_fast {

constexpr int LBITS = 12, DBITS = 8;
constexpr int LSIZE = (1 << LBITS) + 1400, DSIZE = (1 << DBITS) + 700;
enum { K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_LIT2 = 5 };
inline uint32_t mk(int kind, int value, int ebits, int nbits) {
  return ((uint32_t)value << 16) | ((uint32_t)kind << 13) | ((uint32_t)ebits << 8) | (uint32_t)nbits;
}
const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// decode table of a canonical code given by its lengths.  is_dist selects the symbol meaning.  false: over-subscribed
// code, symbol out of range, table too small (the caller falls back to zlib)
bool build(const uint8_t* len, int nsym, bool is_dist, int tbits, uint32_t* tab, int tsize) {
  int count[16] = {0};
  for (int s = 0; s < nsym; s++) count[len[s]]++;
  count[0] = 0;
  int64_t left = 1;
  for (int l = 1; l <= 15; l++) {
    left = (left << 1) - count[l];
    if (left < 0) return false;                         // over-subscribed
  }
  int next[16] = {0}, code = 0;
  for (int l = 1; l <= 15; l++) {
    code = (code + count[l - 1]) << 1;
    next[l] = code;
  }
  const int psize = 1 << tbits;
  for (int i = 0; i < psize; i++) tab[i] = mk(K_BAD, 0, 0, 1);
  auto entry = [&](int s, int nbits) -> uint32_t {
    if (is_dist) return s < 30 ? mk(K_BASE, DIST_BASE[s], DIST_EXTRA[s], nbits) : mk(K_BAD, 0, 0, nbits);
    if (s < 256) return mk(K_LIT, s, 0, nbits);
    if (s == 256) return mk(K_EOB, 0, 0, nbits);
    return s < 286 ? mk(K_BASE, LEN_BASE[s - 257], LEN_EXTRA[s - 257], nbits) : mk(K_BAD, 0, 0, nbits);
  };
  auto reverse = [](int c, int l) {
    int r = 0;
    for (int k = 0; k < l; k++) r |= ((c >> k) & 1) << (l - 1 - k);
    return r;
  };
  // pass 1: short codes into the primary table; longest code of every long prefix
  uint8_t sub_bits[1 << LBITS];                         // (tbits <= LBITS)
  memset(sub_bits, 0, sizeof(sub_bits));
  std::vector<int> rev(nsym, 0);
  for (int s = 0; s < nsym; s++) {
    const int l = len[s];
    if (!l) continue;
    const int r = reverse(next[l]++, l);
    rev[s] = r;
    if (l <= tbits) {
      const uint32_t e = entry(s, l);
      for (int j = r; j < psize; j += 1 << l) tab[j] = e;
    } else {
      const int pre = r & (psize - 1);
      if (l - tbits > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - tbits);
    }
  }
  // pass 2: sub-tables
  int used = psize;
  for (int pre = 0; pre < psize; pre++) {
    if (!sub_bits[pre]) continue;
    const int n = 1 << sub_bits[pre];
    if (used + n > tsize) return false;
    tab[pre] = mk(K_SUB, used, sub_bits[pre], tbits);
    for (int j = 0; j < n; j++) tab[used + j] = mk(K_BAD, 0, 0, 1);
    used += n;
  }
  for (int s = 0; s < nsym; s++) {
    const int l = len[s];
    if (l <= tbits) continue;
    const int pre = rev[s] & (psize - 1);
    const uint32_t link = tab[pre];
    const int start = (int)(link >> 16), sb = (int)((link >> 8) & 31);
    const uint32_t e = entry(s, l - tbits);
    for (int j = rev[s] >> tbits; j < (1 << sb); j += 1 << (l - tbits)) tab[start + j] = e;
  }
  return true;
}

// Two literals per look-up: where a primary entry is a literal of l1 bits and the remaining tbits - l1 index bits hold a
// complete second literal code, the entry becomes the PAIR (value = first | second << 8, length = both).  Table-driven
// decoding is one dependent chain per symbol (mask, load, shift: ~7 cycles); image rows are nearly all literals of 4-8
// bits, so most look-ups of such a stream then yield two bytes.  One pass over the primary table per block.
void pair_literals(uint32_t* tab, int tbits) {
  const int psize = 1 << tbits;
  uint32_t single[1 << LBITS];
  memcpy(single, tab, sizeof(uint32_t) * (size_t)psize);
  for (int idx = 0; idx < psize; idx++) {
    const uint32_t e1 = single[idx];
    if (((e1 >> 13) & 7) != K_LIT) continue;
    const int l1 = (int)(e1 & 255), rest = tbits - l1;
    if (rest < 1) continue;
    const uint32_t e2 = single[idx >> l1];            // (the unknown bits above `rest` read as zeros: irrelevant for a code that fits)
    if (((e2 >> 13) & 7) != K_LIT || (int)(e2 & 255) > rest) continue;
    tab[idx] = mk(K_LIT2, (int)((e1 >> 16) | ((e2 >> 16) << 8)), 0, l1 + (int)(e2 & 255));
  }
}

struct Tables {
  uint32_t lt[LSIZE], dt[DSIZE];
};

const Tables* fixed_tables() {                          // the fixed code of RFC 1951 3.2.6, built once (thread-safe static)
  struct Holder {
    Tables t;
    bool ok;
    Holder() {
      uint8_t l[288], d[32];
      for (int i = 0; i < 144; i++) l[i] = 8;
      for (int i = 144; i < 256; i++) l[i] = 9;
      for (int i = 256; i < 280; i++) l[i] = 7;
      for (int i = 280; i < 288; i++) l[i] = 8;
      for (int i = 0; i < 32; i++) d[i] = 5;
      ok = build(l, 288, false, LBITS, t.lt, LSIZE) && build(d, 32, true, DBITS, t.dt, DSIZE);
      if (ok) pair_literals(t.lt, LBITS);
    }
  };
  static const Holder h;
  return h.ok ? &h.t : nullptr;
}

bool inflate(const uint8_t* in, size_t n, uint8_t* out, size_t out_len, Tables* dyn_space = nullptr) {
  if (n < 6) return false;
  if ((in[0] & 0x0f) != 8 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return false;    // deflate, no preset dictionary
  const uint8_t* ip = in + 2;
  const uint8_t* const iend = in + n - 4;               // the Adler-32 trailer starts here
  uint8_t* op = out;
  uint8_t* const oend = out + out_len;
  uint64_t bb = 0;
  int bc = 0;
  // at least 56 valid bits after a refill (reads up to 8 bytes at ip: padded by the caller); bits past iend are zeros
  auto refill = [&]() {
    uint64_t w;
    memcpy(&w, ip, 8);
    bb |= w << bc;
    const int adv = (63 - bc) >> 3;
    ip += adv;
    bc += adv << 3;
  };
  std::unique_ptr<Tables> own;                          // (26 KB of dynamic-block tables: the caller's when it has some)
  if (!dyn_space) own.reset(dyn_space = new Tables);
  Tables* const dyn = dyn_space;
  for (;;) {
    if (ip > iend + 8) return false;                    // ran far past the input
    refill();
    const int last = (int)(bb & 1), type = (int)((bb >> 1) & 3);
    bb >>= 3;
    bc -= 3;
    if (type == 0) {                                    // stored: byte-align, LEN, NLEN, bytes
      const int drop = bc & 7;
      bb >>= drop;
      bc -= drop;
      // give the whole bytes still in the bit buffer back to the input
      ip -= bc >> 3;
      bb = 0;
      bc = 0;
      if (ip + 4 > iend) return false;
      const uint32_t len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
      if ((len ^ nlen) != 0xffffu) return false;
      ip += 4;
      if (ip + len > iend || op + len > oend) return false;
      memcpy(op, ip, len);
      ip += len;
      op += len;
    } else if (type == 1 || type == 2) {
      const Tables* T;
      if (type == 1) {
        T = fixed_tables();
        if (!T) return false;
      } else {
        const int hlit = (int)(bb & 31) + 257, hdist = (int)((bb >> 5) & 31) + 1, hclen = (int)((bb >> 10) & 15) + 4;
        bb >>= 14;
        bc -= 14;
        if (hlit > 286 || hdist > 30) return false;
        static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        refill();
        for (int i = 0; i < hclen; i++) {
          if (bc < 3) refill();
          cl[order[i]] = (uint8_t)(bb & 7);
          bb >>= 3;
          bc -= 3;
        }
        uint32_t ct[1 << 7];
        if (!build(cl, 19, false, 7, ct, 1 << 7)) return false;      // (symbols 0..18 decode as "literals")
        uint8_t lens[286 + 30 + 138];
        int k = 0;
        while (k < hlit + hdist) {
          if (ip > iend + 8) return false;
          refill();
          const uint32_t e = ct[bb & 127];
          if (((e >> 13) & 7) != K_LIT) return false;
          const int nb = (int)(e & 255), sym = (int)(e >> 16);
          bb >>= nb;
          bc -= nb;
          if (sym < 16) {
            lens[k++] = (uint8_t)sym;
          } else {
            int rep, val = 0;
            if (sym == 16) {
              if (k == 0) return false;
              val = lens[k - 1];
              rep = 3 + (int)(bb & 3);
              bb >>= 2; bc -= 2;
            } else if (sym == 17) {
              rep = 3 + (int)(bb & 7);
              bb >>= 3; bc -= 3;
            } else {
              rep = 11 + (int)(bb & 127);
              bb >>= 7; bc -= 7;
            }
            if (k + rep > hlit + hdist) return false;
            while (rep--) lens[k++] = (uint8_t)val;
          }
        }
        if (lens[256] == 0) return false;               // no end-of-block code
        if (!build(lens, hlit, false, LBITS, dyn->lt, LSIZE) || !build(lens + hlit, hdist, true, DBITS, dyn->dt, DSIZE)) return false;
        pair_literals(dyn->lt, LBITS);
        T = dyn;
      }
      const uint32_t* lt = T->lt;
      const uint32_t* dt = T->dt;
      for (;;) {
        if (ip > iend + 8) return false;
        refill();                                       // >= 56 bits: a literal/length (<= 15 + 5) and a distance (<= 15 + 13) fit
        uint32_t e = lt[bb & ((1u << LBITS) - 1)];
        int kind = (int)((e >> 13) & 7);
        if (kind == K_SUB) {
          bb >>= LBITS;
          bc -= LBITS;
          e = lt[(e >> 16) + (bb & ((1u << ((e >> 8) & 31)) - 1))];
          kind = (int)((e >> 13) & 7);
        }
        bb >>= (e & 255);
        bc -= (int)(e & 255);
        if (kind == K_LIT || kind == K_LIT2) {
          // one or two literals -- then up to three more look-ups in the primary table from the bits already loaded
          // (>= 56 - 15 left after the first, <= LBITS = 12 per look-up)
          if (kind == K_LIT) {
            if (op >= oend) return false;
            *op++ = (uint8_t)(e >> 16);
          } else {
            if (op + 2 > oend) return false;
            const uint16_t two = (uint16_t)(e >> 16);
            memcpy(op, &two, 2);                          // (little-endian host: first literal first)
            op += 2;
          }
          for (int more = 0; more < 3; more++) {
            const uint32_t e2 = lt[bb & ((1u << LBITS) - 1)];
            const int k2 = (int)((e2 >> 13) & 7);
            if (k2 == K_LIT) {
              if (op >= oend) break;
              *op++ = (uint8_t)(e2 >> 16);
            } else if (k2 == K_LIT2) {
              if (op + 2 > oend) break;
              const uint16_t two = (uint16_t)(e2 >> 16);
              memcpy(op, &two, 2);
              op += 2;
            } else {
              break;
            }
            bb >>= (e2 & 255);
            bc -= (int)(e2 & 255);
          }
          continue;
        }
        if (kind == K_EOB) break;
        if (kind != K_BASE) return false;
        const int eb = (int)((e >> 8) & 31);
        const size_t length = (e >> 16) + (size_t)(bb & ((1u << eb) - 1));
        bb >>= eb;
        bc -= eb;
        uint32_t d = dt[bb & ((1u << DBITS) - 1)];
        if (((d >> 13) & 7) == K_SUB) {
          bb >>= DBITS;
          bc -= DBITS;
          d = dt[(d >> 16) + (bb & ((1u << ((d >> 8) & 31)) - 1))];
        }
        if (((d >> 13) & 7) != K_BASE) return false;
        bb >>= (d & 255);
        bc -= (int)(d & 255);
        if (bc < 13) refill();
        const int deb = (int)((d >> 8) & 31);
        const size_t dist = (d >> 16) + (size_t)(bb & ((1u << deb) - 1));
        bb >>= deb;
        bc -= deb;
        if (dist > (size_t)(op - out) || length > (size_t)(oend - op)) return false;
        const uint8_t* src = op - dist;
        uint8_t* dst = op;
        op += length;
        if (dist >= 8) {                                // (may write up to 7 bytes past op: the caller's 16 spare bytes)
          for (size_t k = 0; k < length; k += 8) memcpy(dst + k, src + k, 8);
        } else if (dist == 1) {
          memset(dst, src[0], length);
        } else {
          for (size_t k = 0; k < length; k++) dst[k] = src[k];
        }
      }
    } else {
      return false;
    }
    if (last) break;
  }
  if (op != oend) return false;
  // the trailer: give back the unread whole bytes, then four bytes of Adler-32 exactly at the end of the input
  ip -= bc >> 3;
  if (ip != iend) return false;
  const uint32_t want = ((uint32_t)iend[0] << 24) | ((uint32_t)iend[1] << 16) | ((uint32_t)iend[2] << 8) | iend[3];
  return fast_adler32(1u, out, out_len) == want;
}

}  // namespace inflate_fast

// Paeth rows after their first pixel (BPP = bytes per pixel: 1, 2, 3, 4, 6 or 8).  The channels of a pixel sit in the
// 16-bit lanes of one SSE2 register: the three distances, the choice (a on ties, then b: min + compare, no branches --
// on image data the three-way choice is unpredictable) and the sum are a dozen vector operations per PIXEL instead of per
// byte, one dependent chain through `a`.  Loads and stores move 4 (BPP <= 4) or 8 bytes: up to 3 / 2 bytes beyond the
// pixel are read (the rows' own following bytes) and written (overwritten by the next pixel; the caller's buffers have 16
// spare bytes behind the last row).
template <int BPP>
void paeth_row(uint8_t* cur, const uint8_t* src, const uint8_t* up, size_t stride) {
  const __m128i zero = _mm_setzero_si128();
  auto load = [&](const uint8_t* q) {
    if (BPP <= 4) {
      uint32_t w;
      memcpy(&w, q, 4);
      return _mm_unpacklo_epi8(_mm_cvtsi32_si128((int)w), zero);
    }
    return _mm_unpacklo_epi8(_mm_loadl_epi64(reinterpret_cast<const __m128i*>(q)), zero);
  };
  __m128i a = load(cur), c = load(up);               // the first pixel: left / upper-left neighbours of the second
  size_t i = BPP;
  for (; i + BPP <= stride; i += BPP) {
    const __m128i bb = load(up + i), d = load(src + i);
    const __m128i pa_s = _mm_sub_epi16(bb, c), pb_s = _mm_sub_epi16(a, c), pc_s = _mm_add_epi16(pa_s, pb_s);
    const __m128i pa = _mm_max_epi16(pa_s, _mm_sub_epi16(zero, pa_s));
    const __m128i pb = _mm_max_epi16(pb_s, _mm_sub_epi16(zero, pb_s));
    const __m128i pc = _mm_max_epi16(pc_s, _mm_sub_epi16(zero, pc_s));
    const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
    const __m128i is_a = _mm_cmpeq_epi16(smallest, pa), is_b = _mm_cmpeq_epi16(smallest, pb);
    // pred = is_a ? a : (is_b ? b : c)
    const __m128i bc = _mm_or_si128(_mm_and_si128(is_b, bb), _mm_andnot_si128(is_b, c));
    const __m128i pred = _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, bc));
    a = _mm_and_si128(_mm_add_epi16(pred, d), _mm_set1_epi16(0xff));
    c = bb;
    const __m128i packed = _mm_packus_epi16(a, a);
    if (BPP <= 4) {
      const uint32_t w = (uint32_t)_mm_cvtsi128_si32(packed);
      memcpy(cur + i, &w, 4);
    } else {
      _mm_storel_epi64(reinterpret_cast<__m128i*>(cur + i), packed);
    }
  }
}

// TWO consecutive Paeth rows at once: a row's pixel needs the pixel above it, so row y + 1 can follow row y one pixel
// behind -- two independent dependency chains in flight instead of one (the single-row loop is bound by the latency of
// its chain: subtract, absolute values, minimum, compare, select, add).  cur1 = the row after cur0 (its "up" row);
// both rows' first pixels are done by the caller.  Stores move whole words like paeth_row: the last store of row 0 runs
// into the first bytes of row 1, which are therefore written again at the end.
template <int BPP>
void paeth_rows2(uint8_t* cur0, const uint8_t* src0, const uint8_t* up0, uint8_t* cur1, const uint8_t* src1, size_t stride) {
  const __m128i zero = _mm_setzero_si128(), ff = _mm_set1_epi16(0xff);
  auto load = [&](const uint8_t* q) {
    if (BPP <= 4) {
      uint32_t w;
      memcpy(&w, q, 4);
      return _mm_unpacklo_epi8(_mm_cvtsi32_si128((int)w), zero);
    }
    return _mm_unpacklo_epi8(_mm_loadl_epi64(reinterpret_cast<const __m128i*>(q)), zero);
  };
  auto store = [&](uint8_t* q, __m128i v) {
    const __m128i packed = _mm_packus_epi16(v, v);
    if (BPP <= 4) {
      const uint32_t w = (uint32_t)_mm_cvtsi128_si32(packed);
      memcpy(q, &w, 4);
    } else {
      _mm_storel_epi64(reinterpret_cast<__m128i*>(q), packed);
    }
  };
  auto predict = [&](__m128i a, __m128i b, __m128i c) {
    const __m128i pa_s = _mm_sub_epi16(b, c), pb_s = _mm_sub_epi16(a, c), pc_s = _mm_add_epi16(pa_s, pb_s);
    const __m128i pa = _mm_max_epi16(pa_s, _mm_sub_epi16(zero, pa_s));
    const __m128i pb = _mm_max_epi16(pb_s, _mm_sub_epi16(zero, pb_s));
    const __m128i pc = _mm_max_epi16(pc_s, _mm_sub_epi16(zero, pc_s));
    const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
    const __m128i is_a = _mm_cmpeq_epi16(smallest, pa), is_b = _mm_cmpeq_epi16(smallest, pb);
    const __m128i bc = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
    return _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, bc));
  };
  uint8_t lead1[8];
  memcpy(lead1, cur1, BPP);
  __m128i a0 = load(cur0), c0 = load(up0);            // row 0: left and upper-left of its pixel 1
  __m128i a1 = load(cur1), c1 = a0;                    // row 1: left of its pixel 1; upper-left = row 0's pixel 0
  size_t i = BPP;                                      // byte offset of row 0's pixel; row 1 works on i - BPP
  for (; i + BPP <= stride; i += BPP) {
    const __m128i b0 = load(up0 + i), d0 = load(src0 + i);
    const __m128i n0 = _mm_and_si128(_mm_add_epi16(predict(a0, b0, c0), d0), ff);
    if (i >= 2 * BPP) {                                // row 1's pixel i / BPP - 1: above it row 0's previous pixel (a0)
      const __m128i d1 = load(src1 + i - BPP);
      const __m128i n1 = _mm_and_si128(_mm_add_epi16(predict(a1, a0, c1), d1), ff);
      store(cur1 + i - BPP, n1);
      a1 = n1;
      c1 = a0;
    }
    store(cur0 + i, n0);
    c0 = b0;
    a0 = n0;
  }
  if (i >= 2 * BPP) {                                  // row 1's last pixel
    const __m128i d1 = load(src1 + i - BPP);
    store(cur1 + i - BPP, _mm_and_si128(_mm_add_epi16(predict(a1, a0, c1), d1), ff));
  }
  memcpy(cur1, lead1, BPP);
}

struct DecodeScratch {
  std::vector<uint8_t> file, zin, raw, img, zero, bgr;
  std::vector<uint16_t> d16;
  std::vector<int32_t> x0, x1, y0, y1;              // resize tables (rr_io_read_frames_scaled)
  std::vector<double> wx, wy;
  Png png;
  inflate_fast::Tables tables;
};
Pool<DecodeScratch> g_decode_pool;

// inflate: sc.raw = h rows of 1 + `stride` bytes, the filter type of a row in front of its filtered bytes
int inflate_rows(const Png& p, DecodeScratch& sc, size_t& stride) {
  const int ch = channels_of(p.ctype);
  if (!ch || p.interlace || (p.depth != 8 && p.depth != 16) || (p.ctype == 3 && p.depth != 8)) return RR_E_UNSUPPORTED;
  const size_t bpp = (size_t)ch * p.depth / 8;
  stride = (size_t)p.w * bpp;
  const size_t raw_len = (stride + 1) * p.h;
  std::vector<uint8_t>& raw = sc.raw;
  at_least(raw, raw_len + 16);                          // (spare bytes: the fast decoder copies matches in 8-byte pieces)
  {
    // (it also loads 8 bytes at a time; between two of its bound checks a malformed stream can pull the read position up
    //  to ~30 bytes past the end -- block header + code-length codes, or the refills of one match -- hence 64 spare bytes)
    std::vector<uint8_t>& zin = sc.zin;
    at_least(zin, p.idat.size() + 64);
    memcpy(zin.data(), p.idat.data(), p.idat.size());
    memset(zin.data() + p.idat.size(), 0, 64);
    if (!inflate_fast::inflate(zin.data(), p.idat.size(), raw.data(), raw_len, &sc.tables)) {      // anything unusual: zlib decides
      uLongf out_len = (uLongf)raw_len;
      if (uncompress(raw.data(), &out_len, p.idat.data(), (uLong)p.idat.size()) != Z_OK || out_len != raw_len) return RR_E_PARSE;
    }
  }
  return RR_OK;
}

// inflate + reverse the scanline filters: sc.img = h rows of `stride` bytes
int decode(const Png& p, DecodeScratch& sc, size_t& stride) {
  int rc = inflate_rows(p, sc, stride);
  if (rc) return rc;
  const int ch = channels_of(p.ctype);
  const size_t bpp = (size_t)ch * p.depth / 8;
  std::vector<uint8_t>& raw = sc.raw;
  std::vector<uint8_t>& img = sc.img;
  at_least(img, stride * p.h + 16);                    // (16 spare bytes: the Paeth rows store whole words)
  if (sc.zero.size() < stride + 16) sc.zero.assign(stride + 16 + stride / 8, 0);      // the row above the first one, never written
  const std::vector<uint8_t>& zero = sc.zero;
  for (uint32_t y = 0; y < p.h; y++) {
    const uint8_t* src = &raw[(stride + 1) * y];
    const int ft = src[0];
    src++;
    uint8_t* cur = &img[stride * y];
    const uint8_t* up = y ? &img[stride * (y - 1)] : zero.data();
    const size_t lead = bpp < stride ? bpp : stride;  // the first pixel has no left neighbour
    switch (ft) {
      case 0: memcpy(cur, src, stride); break;
      case 1:
        memcpy(cur, src, lead);
        for (size_t i = lead; i < stride; i++) cur[i] = (uint8_t)(src[i] + cur[i - bpp]);
        break;
      case 2:
        for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(src[i] + up[i]);
        break;
      case 3:
        for (size_t i = 0; i < lead; i++) cur[i] = (uint8_t)(src[i] + (up[i] >> 1));
        for (size_t i = lead; i < stride; i++) cur[i] = (uint8_t)(src[i] + ((cur[i - bpp] + up[i]) >> 1));
        break;
      case 4:
        for (size_t i = 0; i < lead; i++) cur[i] = (uint8_t)(src[i] + up[i]);          // paeth(0, b, 0) = b
        // the next row is a Paeth row too: both at once (one-byte pixels excepted: a word-wise store at the end of row 0
        // would run over more of row 1 than its first pixel)
        if (y + 1 < p.h && raw[(stride + 1) * (y + 1)] == 4 && stride >= 2 * bpp && bpp >= 2) {
          const uint8_t* src1 = &raw[(stride + 1) * (y + 1)] + 1;
          uint8_t* cur1 = &img[stride * (y + 1)];
          for (size_t i = 0; i < lead; i++) cur1[i] = (uint8_t)(src1[i] + cur[i]);
          switch (bpp) {
            case 2: paeth_rows2<2>(cur, src, up, cur1, src1, stride); break;
            case 3: paeth_rows2<3>(cur, src, up, cur1, src1, stride); break;
            case 4: paeth_rows2<4>(cur, src, up, cur1, src1, stride); break;
            case 6: paeth_rows2<6>(cur, src, up, cur1, src1, stride); break;
            default: paeth_rows2<8>(cur, src, up, cur1, src1, stride); break;
          }
          y++;
          break;
        }
        switch (bpp) {                                // left / upper-left neighbours stay in registers, one chain per channel
          case 1: paeth_row<1>(cur, src, up, stride); break;
          case 2: paeth_row<2>(cur, src, up, stride); break;
          case 3: paeth_row<3>(cur, src, up, stride); break;
          case 4: paeth_row<4>(cur, src, up, stride); break;
          case 6: paeth_row<6>(cur, src, up, stride); break;
          default: paeth_row<8>(cur, src, up, stride); break;
        }
        break;
      default: return RR_E_PARSE;
    }
  }
  return RR_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Fast deflate for filtered scanlines (strategy 3).  zlib spends ~10 ns per input byte even with Z_RLE / Z_HUFFMAN_ONLY
// (per-symbol tallying, bit-at-a-time block emission); at a few hundred frames per second on a handful of cores that is the
// driver's bottleneck.  Sub-filtered image rows need no string matching: small residuals (a skewed byte histogram) and
// runs of one value (flat regions, the colour-mapped mask).  So: per block of 128 KB, one pass turns the input into
// literals and distance-1 matches (a run of the previous byte, like Z_RLE), a histogram gives a dynamic Huffman code
// (RFC 1951 section 3.2.7; lengths limited to 15 bits), and the symbols are emitted through a 64-bit bit buffer.  The result
// is an ordinary zlib stream (any inflate reads it); about the size of Z_RLE's, several times faster.
// ---------------------------------------------------------------------------------------------------------------------
struct BitWriter {                                    // LSB-first bit buffer over a caller-sized byte array
  uint8_t* p;
  uint64_t acc = 0;
  int n = 0;
  explicit BitWriter(uint8_t* dst) : p(dst) {}
  inline void put(uint32_t bits, int len) {           // len <= 32; at most 7 bits are pending before the call
    acc |= (uint64_t)bits << n;
    n += len;
    memcpy(p, &acc, 8);                               // (little-endian host) the whole bytes of acc
    p += n >> 3;
    acc >>= n & ~7;
    n &= 7;
  }
  inline void add(uint32_t bits, int len) {           // no store: the caller flushes before acc could overflow
    acc |= (uint64_t)bits << n;
    n += len;
  }
  inline void flush() {
    memcpy(p, &acc, 8);
    p += n >> 3;
    acc >>= n & ~7;
    n &= 7;
  }
  void finish() {                                     // pad to a byte boundary
    if (n > 0) *p++ = (uint8_t)acc;
    n = 0;
    acc = 0;
  }
};

// code lengths of a minimum-redundancy prefix code, limited to max_len bits (count-based fix-up of over-long codes)
void huffman_lengths(const uint32_t* freq, int nsym, int max_len, uint8_t* len) {
  struct Node { uint64_t w; int a, b; };
  std::vector<int> used;
  for (int i = 0; i < nsym; i++) {
    len[i] = 0;
    if (freq[i]) used.push_back(i);
  }
  if (used.empty()) return;
  if (used.size() == 1) { len[used[0]] = 1; return; }
  // two-queue construction over the symbols sorted by weight
  std::sort(used.begin(), used.end(), [&](int x, int y) { return freq[x] != freq[y] ? freq[x] < freq[y] : x < y; });
  const int m = (int)used.size();
  std::vector<Node> nodes((size_t)2 * m);
  for (int i = 0; i < m; i++) nodes[i] = Node{freq[used[i]], -1, -1};
  int leaf = 0, inner = m, made = m;
  auto take = [&]() {
    if (leaf < m && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return leaf++;
    return inner++;
  };
  while (made < 2 * m - 1) {
    const int x = take(), y = take();
    nodes[made++] = Node{nodes[x].w + nodes[y].w, x, y};
  }
  std::vector<int> depth((size_t)2 * m, 0);
  for (int i = 2 * m - 2; i >= m; i--) {              // parents come after their children: walk down from the root
    depth[nodes[i].a] = depth[i] + 1;
    depth[nodes[i].b] = depth[i] + 1;
  }
  std::vector<int> count((size_t)max_len + 2, 0);
  for (int i = 0; i < m; i++) count[depth[i] > max_len ? max_len : depth[i]]++;
  uint64_t total = 0;
  for (int l = 1; l <= max_len; l++) total += (uint64_t)count[l] << (max_len - l);
  while (total > (1ull << max_len)) {                 // Kraft sum too large after the clamp: lengthen the cheapest codes
    count[max_len]--;
    for (int l = max_len - 1; l >= 1; l--)
      if (count[l]) {
        count[l]--;
        count[l + 1] += 2;
        break;
      }
    total--;
  }
  int at = m - 1;                                     // `used` is sorted by rising weight: the rarest symbols get the longest codes
  int l = max_len;
  std::vector<int> c2(count);
  for (int i = 0; i < m; i++) {
    while (l > 0 && c2[l] == 0) l--;
    len[used[i]] = (uint8_t)l;
    c2[l]--;
  }
  (void)at;
}

// canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first bit buffer
void huffman_codes(const uint8_t* len, int nsym, uint16_t* code) {
  int bl_count[16] = {0};
  for (int i = 0; i < nsym; i++) bl_count[len[i]]++;
  bl_count[0] = 0;
  int next[16] = {0}, c = 0;
  for (int b = 1; b <= 15; b++) {
    c = (c + bl_count[b - 1]) << 1;
    next[b] = c;
  }
  for (int i = 0; i < nsym; i++) {
    const int L = len[i];
    code[i] = 0;
    if (!L) continue;
    uint32_t v = (uint32_t)next[L]++, r = 0;
    for (int k = 0; k < L; k++) r |= ((v >> k) & 1u) << (L - 1 - k);
    code[i] = (uint16_t)r;
  }
}

struct LenCode { uint16_t sym; uint8_t ebits; uint16_t eval; };
struct LenTable {                                     // match length 3..258 -> (symbol, extra bits, extra value)
  LenCode tab[259];
  LenTable() {
    static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const int extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int L = 0; L < 3; L++) tab[L] = LenCode{0, 0, 0};
    for (int L = 3; L <= 258; L++) {
      int k = 28;
      while (base[k] > L) k--;
      tab[L] = LenCode{(uint16_t)(257 + k), (uint8_t)extra[k], (uint16_t)(L - base[k])};
    }
  }
};
const LenCode* length_table() {
  static const LenTable t;                            // (initialised once, thread-safe: the driver encodes on many threads)
  return t.tab;
}

// the zlib stream (header, deflate blocks, adler32) of n bytes
struct ByteBuf {                                      // uninitialised storage (a std::vector would zero megabytes per file)
  std::unique_ptr<uint8_t[]> mem;
  size_t cap = 0, len = 0;
  void reserve_raw(size_t c) {                        // (grow-only: the buffer is reused from file to file)
    if (c > cap) {
      mem.reset(new uint8_t[c + c / 8]);
      cap = c + c / 8;
    }
    len = 0;
  }
  uint8_t* data() { return mem.get(); }
};
inline uint64_t load64(const uint8_t* p) {
  uint64_t w;
  memcpy(&w, p, 8);
  return w;
}
struct Run {                                          // a run of the byte before it: in[pos .. pos + len) == in[pos - 1]
  uint32_t pos, len;                                  // (pos relative to the block; 3 <= len <= 258)
};
struct EncodeScratch {
  ByteBuf z;
  std::vector<Run> runs;
};
Pool<EncodeScratch> g_encode_pool;

void fast_deflate(const uint8_t* in, size_t n, ByteBuf& out, std::vector<Run>& runs) {
  const LenCode* LT = length_table();
  const size_t BLOCK = 128 * 1024;
  // worst case: every byte a 15-bit literal is impossible for a Huffman code of the block's own histogram (< 9 bits per
  // byte on average); 2 n + slack is a safe roof, trimmed at the end
  out.reserve_raw(2 * n + (n / BLOCK + 2) * 512 + 64);
  out.data()[0] = 0x78;
  out.data()[1] = 0x01;
  BitWriter bw(out.data() + 2);
  at_least(runs, BLOCK / 3 + 2);
  size_t pos = 0;
  if (n == 0) {                                       // one empty stored block
    bw.put(1, 1); bw.put(0, 2); bw.finish();
    bw.put(0xffff0000u, 32);
  }
  const uint64_t ONES = 0x0101010101010101ull, HIGH = 0x8080808080808080ull;
  while (pos < n) {
    const size_t end = pos + BLOCK < n ? pos + BLOCK : n;
    const bool last = end == n;
    // ---- pass 1: the block's runs (the previous byte repeated >= 3 times, like Z_RLE) and the histogram of what is left.
    // 64 bytes at a time: byte-wise compares of the chunk with itself one byte earlier give a 64-bit mask of the bytes
    // that repeat their predecessor; three set bits in a row start a run.  Chunks without one (most chunks of filtered
    // image rows) are 64 literals.  Eight histograms: neighbours often hold the same value (every fourth byte of an opaque
    // RGBA row is the same alpha residual), and a counter incremented twice in a row waits for its own store.
    uint32_t fl[286] = {0}, fd[30] = {0};
    uint32_t h[8][256];
    memset(h, 0, sizeof(h));
    size_t nr = 0;
    size_t i = pos;
    if (i == 0) {                                     // the very first byte has no predecessor
      h[0][in[0]]++;
      i = 1;
    }
    auto take_run = [&](size_t at) {                  // in[at - 1] == in[at] == in[at + 1] == in[at + 2], at + 2 < end: its length
      const uint8_t b = in[at];
      size_t r = 3;
      const size_t lim = end - at < 258 ? end - at : 258;
      const uint64_t bb = ONES * b;
      bool open = true;
      while (r + 8 <= lim) {
        const uint64_t v = load64(in + at + r) ^ bb;
        if (v) {
          r += (size_t)(__builtin_ctzll(v) >> 3);
          open = false;
          break;
        }
        r += 8;
      }
      while (open && r < lim && in[at + r] == b) r++;
      runs[nr++] = Run{(uint32_t)(at - pos), (uint32_t)r};
      fl[LT[r].sym]++;
      fd[0]++;
      return r;
    };
    while (i + 66 <= end) {
      uint64_t eq = 0;                                // bit k: in[i + k] == in[i + k - 1]
      for (int j = 0; j < 4; j++) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(in + i + 16 * j));
        const __m128i p = _mm_loadu_si128(reinterpret_cast<const __m128i*>(in + i + 16 * j - 1));
        eq |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(a, p)) << (16 * j);
      }
      const uint64_t e64 = in[i + 64] == in[i + 63], e65 = in[i + 65] == in[i + 64];
      uint64_t r3 = eq & ((eq >> 1) | (e64 << 63)) & ((eq >> 2) | (e64 << 62) | (e65 << 63));
      size_t cur = 0;
      while (r3) {
        const size_t k = (size_t)__builtin_ctzll(r3);
        for (size_t j = cur; j < k; j++) h[j & 7][in[i + j]]++;
        cur = k + take_run(i + k);
        if (cur >= 64) break;
        r3 &= ~0ull << cur;
      }
      if (cur >= 64) {
        i += cur;
        continue;
      }
      const uint8_t* c = in + i;
      for (size_t j = cur; j < 64; j++) h[j & 7][c[j]]++;
      i += 64;
    }
    while (i < end) {
      const uint8_t b = in[i];
      if (b == in[i - 1] && i + 2 < end && in[i + 1] == b && in[i + 2] == b) i += take_run(i);
      else h[i++ & 7][b]++;
    }
    for (int k = 0; k < 256; k++) fl[k] = (h[0][k] + h[1][k]) + (h[2][k] + h[3][k]) + ((h[4][k] + h[5][k]) + (h[6][k] + h[7][k]));
    fl[256] = 1;                                      // end of block
    if (!fd[0]) fd[0] = 1;                            // a distance code must exist
    uint8_t ll[286], dl[30];
    uint16_t lc[286], dc[30];
    huffman_lengths(fl, 286, 12, ll);                 // 12 bits: four literals (48 bits) + 7 pending fit the 64-bit buffer
    huffman_lengths(fd, 30, 15, dl);
    huffman_codes(ll, 286, lc);
    huffman_codes(dl, 30, dc);
    int nlit = 286;
    while (nlit > 257 && ll[nlit - 1] == 0) nlit--;
    const int ndist = 1;
    // the code lengths themselves, Huffman coded without the repeat symbols (a few hundred bits per 128 KB block)
    uint32_t fc[19] = {0};
    for (int k = 0; k < nlit; k++) fc[ll[k]]++;
    for (int k = 0; k < ndist; k++) fc[dl[k]]++;
    uint8_t cl[19];
    uint16_t cc[19];
    huffman_lengths(fc, 19, 7, cl);
    huffman_codes(cl, 19, cc);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int ncl = 19;
    while (ncl > 4 && cl[order[ncl - 1]] == 0) ncl--;
    bw.put(last ? 1 : 0, 1);
    bw.put(2, 2);
    bw.put((uint32_t)(nlit - 257), 5);
    bw.put((uint32_t)(ndist - 1), 5);
    bw.put((uint32_t)(ncl - 4), 4);
    for (int k = 0; k < ncl; k++) bw.put(cl[order[k]], 3);
    for (int k = 0; k < nlit; k++) bw.put(cc[ll[k]], cl[ll[k]]);
    for (int k = 0; k < ndist; k++) bw.put(cc[dl[k]], cl[dl[k]]);
    // ---- pass 2: the literals between two runs straight from the input, four per store (code and length in one table
    // entry); a run: length symbol + extra bits + the distance code, pre-merged
    uint32_t lit[256];
    for (int k = 0; k < 256; k++) lit[k] = (uint32_t)lc[k] | ((uint32_t)ll[k] << 16);
    const uint8_t* q = in + pos;
    const uint8_t* const qe = in + end;
    for (size_t r = 0; r <= nr; r++) {
      const uint8_t* const stop = r < nr ? in + pos + runs[r].pos : qe;
      while (q + 4 <= stop) {
        const uint32_t a = lit[q[0]], b = lit[q[1]], c = lit[q[2]], d = lit[q[3]];
        bw.add(a & 0xffffu, (int)(a >> 16));
        bw.add(b & 0xffffu, (int)(b >> 16));
        bw.add(c & 0xffffu, (int)(c >> 16));
        bw.add(d & 0xffffu, (int)(d >> 16));
        bw.flush();
        q += 4;
      }
      while (q < stop) {
        const uint32_t a = lit[*q++];
        bw.put(a & 0xffffu, (int)(a >> 16));
      }
      if (r < nr) {
        const LenCode& L = LT[runs[r].len];
        uint32_t bits = lc[L.sym];
        int len = ll[L.sym];
        bits |= (uint32_t)L.eval << len;
        len += L.ebits;
        bits |= (uint32_t)dc[0] << len;               // distance 1: code 0, no extra bits
        len += dl[0];
        bw.put(bits, len);                            // <= 12 + 5 + 1 bits
        q += runs[r].len;
      }
    }
    bw.put(lc[256], ll[256]);
    pos = end;
  }
  bw.finish();
  const uint32_t ad = fast_adler32(1u, in, n);
  uint8_t* q = bw.p;
  q[0] = (uint8_t)(ad >> 24); q[1] = (uint8_t)(ad >> 16); q[2] = (uint8_t)(ad >> 8); q[3] = (uint8_t)ad;
  out.len = (size_t)(q + 4 - out.data());
}

}  // namespace

static int rr_png_info_impl(const char* path, int32_t* w, int32_t* h, int32_t* channels, int32_t* bit_depth) {
  if (!path || !w || !h || !channels || !bit_depth) return RR_E_ARG;
  Pool<DecodeScratch>::Lease sc(g_decode_pool);
  size_t fsize = 0;
  int rc = RR_OK;
  {                                                   // the signature and the IHDR chunk: the first 33 bytes of the file
    FILE* fh = fopen(path, "rb");
    if (!fh) return RR_E_ARG;
    at_least(sc->file, 64);
    fsize = fread(sc->file.data(), 1, 33, fh);
    fclose(fh);
  }
  Png& p = sc->png;
  if ((rc = parse_chunks(sc->file, fsize, p, false))) return rc;
  *w = (int32_t)p.w;
  *h = (int32_t)p.h;
  *channels = p.ctype == 3 ? 3 : channels_of(p.ctype);
  *bit_depth = p.depth;
  if (p.interlace || !channels_of(p.ctype)) return RR_E_UNSUPPORTED;
  return RR_OK;
}

// cv2.imread(path): 8 bits per channel, three channels, B G R
static int rr_png_read_bgr8_impl(const char* path, uint8_t* out, int32_t H, int32_t W) {
  if (!path || !out) return RR_E_ARG;
  Pool<DecodeScratch>::Lease sc(g_decode_pool);
  size_t fsize = 0;
  int rc = read_file(path, sc->file, fsize);
  if (rc) return rc;
  Png& p = sc->png;
  if ((rc = parse_chunks(sc->file, fsize, p, true))) return rc;
  if ((int32_t)p.w != W || (int32_t)p.h != H) return RR_E_ARG;
  if (p.depth != 8) return RR_E_UNSUPPORTED;          // cv2 scales 16-bit colour down: left to the general decoder
  size_t stride = 0;
  if ((rc = decode(p, *sc, stride))) return rc;
  const std::vector<uint8_t>& img = sc->img;
  const int ch = channels_of(p.ctype);
  if (p.ctype == 2 || p.ctype == 6) {                 // RGB / RGBA: the datasets' case, without per-pixel decisions
    for (int y = 0; y < H; y++) {
      const uint8_t* s = &img[stride * y];
      uint8_t* o = out + (size_t)y * W * 3;
      for (int x = 0; x < W; x++, s += ch, o += 3) {
        o[0] = s[2];
        o[1] = s[1];
        o[2] = s[0];
      }
    }
    return RR_OK;
  }
  for (int y = 0; y < H; y++) {
    const uint8_t* s = &img[stride * y];
    uint8_t* o = out + (size_t)y * W * 3;
    for (int x = 0; x < W; x++) {
      uint8_t r, g, b;
      if (p.ctype == 0 || p.ctype == 4) { r = g = b = s[x * ch]; }
      else if (p.ctype == 3) {
        const size_t k = (size_t)s[x] * 3;
        if (k + 2 >= p.plte.size()) return RR_E_PARSE;
        r = p.plte[k]; g = p.plte[k + 1]; b = p.plte[k + 2];
      } else { r = s[x * ch]; g = s[x * ch + 1]; b = s[x * ch + 2]; }
      o[x * 3] = b; o[x * 3 + 1] = g; o[x * 3 + 2] = r;
    }
  }
  return RR_OK;
}

// cv2.imread(path, cv2.IMREAD_UNCHANGED) of a 16-bit single-channel PNG (depth maps: metres * 256, generator.py:365)
static int rr_png_read_gray16_impl(const char* path, uint16_t* out, int32_t H, int32_t W) {
  if (!path || !out) return RR_E_ARG;
  Pool<DecodeScratch>::Lease sc(g_decode_pool);
  size_t fsize = 0;
  int rc = read_file(path, sc->file, fsize);
  if (rc) return rc;
  Png& p = sc->png;
  if ((rc = parse_chunks(sc->file, fsize, p, true))) return rc;
  if ((int32_t)p.w != W || (int32_t)p.h != H) return RR_E_ARG;
  if (p.ctype != 0 || p.depth != 16) return RR_E_UNSUPPORTED;
  size_t stride = 0;
  if ((rc = decode(p, *sc, stride))) return rc;
  const std::vector<uint8_t>& img = sc->img;
  for (size_t i = 0; i < (size_t)H * W; i++) out[i] = (uint16_t)((img[2 * i] << 8) | img[2 * i + 1]);   // PNG is big-endian
  return RR_OK;
}

// RGBA PNG from its filtered scanlines: H rows of 1 + 4*W bytes (filter byte + filtered pixels).
// strategy: 0 zlib's default (LZ77 + Huffman), 1 Z_RLE (run lengths + Huffman: on filtered image data as small as
// level 1 of the default strategy or smaller, at half the time), 2 Z_HUFFMAN_ONLY, 3 the library's own run-length +
// dynamic-Huffman encoder (fast_deflate above; `level` is ignored).
static int rr_png_write_scanlines_impl(const char* path, const uint8_t* rows, int32_t W, int32_t H, int32_t level, int32_t strategy) {
  if (!path || !rows || W <= 0 || H <= 0 || level < 0 || level > 9 || strategy < 0 || strategy > 3) return RR_E_ARG;
  const uLong n = (uLong)H * (1 + 4 * (uLong)W);
  std::vector<uint8_t> z;
  Pool<EncodeScratch>::Lease es(g_encode_pool);
  ByteBuf& zf = es->z;
  const uint8_t* zdata = nullptr;
  uLongf clen = 0;
  if (memcmp(rows, "RRZ1", 4) == 0) {               // entropy-coded on the device (RR_OPT_PNG_DEFLATE): the IDAT payload as it is
    // (a scanline buffer starts with a filter type 0..4).  The stream is written as it stands, so what can be checked
    // cheaply is: the header's two reserved words are zero, the length fits the buffer, the payload opens with the zlib
    // header the device writes (CMF 0x78, FLG 0x01: a valid pair, no preset dictionary) and is long enough to hold one
    // block and the Adler-32 -- a stale or half-written buffer (a download after a failed batch) fails one of them.
    if (n < 16) return RR_E_PARSE;
    uint32_t hd[4];
    memcpy(hd, rows, 16);
    const uint32_t L = hd[1];
    if (hd[2] != 0 || hd[3] != 0 || (uLong)L + 16 > n || L < 2 + 5 + 4) return RR_E_PARSE;
    if (rows[16] != 0x78 || rows[17] != 0x01) return RR_E_PARSE;
    zdata = rows + 16;
    clen = L;
  } else if (strategy == 3) {
    fast_deflate(rows, (size_t)n, zf, es->runs);
    zdata = zf.data();
    clen = (uLongf)zf.len;
  } else {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15, 9, strategy == 1 ? Z_RLE : strategy == 2 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY) != Z_OK)
      return RR_E_PARSE;
    z.resize(deflateBound(&zs, n));
    zs.next_in = const_cast<Bytef*>(rows);
    zs.avail_in = (uInt)n;
    zs.next_out = z.data();
    zs.avail_out = (uInt)z.size();
    const int zrc = deflate(&zs, Z_FINISH);
    clen = zs.total_out;
    deflateEnd(&zs);
    if (zrc != Z_STREAM_END) return RR_E_PARSE;
    zdata = z.data();
  }
  FILE* fh = fopen(path, "wb");
  if (!fh) return RR_E_ARG;
  auto put32 = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
  auto chunk = [&](const char* tag, const uint8_t* data, uint32_t len) {
    uint8_t hd[8], tl[4];
    put32(hd, len);
    memcpy(hd + 4, tag, 4);
    uint32_t c = (uint32_t)crc32(0L, (const Bytef*)tag, 4);
    if (len) c = fast_crc32(c, data, len);
    put32(tl, (uint32_t)c);
    return fwrite(hd, 1, 8, fh) == 8 && (len == 0 || fwrite(data, 1, len, fh) == len) && fwrite(tl, 1, 4, fh) == 4;
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  uint8_t ihdr[13];
  put32(ihdr, (uint32_t)W);
  put32(ihdr + 4, (uint32_t)H);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  bool ok = fwrite(sig, 1, 8, fh) == 8 && chunk("IHDR", ihdr, 13) && chunk("IDAT", zdata, (uint32_t)clen) && chunk("IEND", nullptr, 0);
  ok = (fclose(fh) == 0) && ok;
  return ok ? RR_OK : RR_E_ARG;
}

// The zlib stream strategy 3 writes, for n arbitrary bytes: out must hold rr_deflate_bound(n) bytes; returns the stream's
// length (tests inflate it with zlib and compare; scripts time it against zlib's strategies).
extern "C" int64_t rr_deflate_bound(int64_t n) { return n + n / 4 + 4096; }
static int64_t rr_deflate_fast_impl(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap) {
  if (!in || !out || n < 0) return RR_E_ARG;
  Pool<EncodeScratch>::Lease es(g_encode_pool);
  ByteBuf& z = es->z;
  fast_deflate(in, (size_t)n, z, es->runs);
  if ((int64_t)z.len > cap) return RR_E_ARG;
  memcpy(out, z.data(), z.len);
  return (int64_t)z.len;
}

// The reader's own inflate on a complete zlib stream whose decoded size is known (tests compare it with zlib on streams of
// every kind; returns 1 when it vouches for the result, 0 when the caller should use zlib).
static int rr_inflate_fast_impl(const uint8_t* in, int64_t n, uint8_t* out, int64_t out_len) {
  if (!in || !out || n < 0 || out_len < 0) return RR_E_ARG;
  std::vector<uint8_t> zin((size_t)n + 64, 0), buf((size_t)out_len + 16);
  memcpy(zin.data(), in, (size_t)n);
  const bool ok = inflate_fast::inflate(zin.data(), (size_t)n, buf.data(), (size_t)out_len);
  if (ok) memcpy(out, buf.data(), (size_t)out_len);
  return ok ? 1 : 0;
}

// No exception crosses the C ABI: an allocation failure (absurd sizes in a damaged file) or any other C++ exception inside
// the codec comes back as RR_E_PARSE.
extern "C" int rr_png_info(const char* path, int32_t* w, int32_t* h, int32_t* channels, int32_t* bit_depth) {
  try {
    return rr_png_info_impl(path, w, h, channels, bit_depth);
  } catch (...) {
    return RR_E_PARSE;
  }
}
extern "C" int rr_png_read_bgr8(const char* path, uint8_t* out, int32_t H, int32_t W) {
  try {
    return rr_png_read_bgr8_impl(path, out, H, W);
  } catch (...) {
    return RR_E_PARSE;
  }
}
extern "C" int rr_png_read_gray16(const char* path, uint16_t* out, int32_t H, int32_t W) {
  try {
    return rr_png_read_gray16_impl(path, out, H, W);
  } catch (...) {
    return RR_E_PARSE;
  }
}
extern "C" int rr_png_write_scanlines(const char* path, const uint8_t* rows, int32_t W, int32_t H, int32_t level, int32_t strategy) {
  try {
    return rr_png_write_scanlines_impl(path, rows, W, H, level, strategy);
  } catch (...) {
    return RR_E_PARSE;
  }
}
// ---------------------------------------------------------------------------------------------------------------------
// Batch forms for the driver: the frames of one pipeline batch decoded straight into the (page-locked) input block of
// their slot, and the two files of every frame written from the slot's scanline blocks, on worker threads inside the
// library.  One call per batch from the driver: with one Python call per frame and file, a few dozen interpreter threads
// spent most of their time handing the interpreter lock to each other (scripts/driver_host_only.py).
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int rr_io_read_frames(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                                 uint8_t* bg_u8, int64_t bg_stride, float* depth_f32, int64_t depth_stride, int32_t threads,
                                 int32_t* status) {
  if (n < 0 || H <= 0 || W <= 0 || !status || (n > 0 && (!image_paths || !bg_u8)) || (depth_paths && !depth_f32) ||
      bg_stride < (int64_t)H * W * 3 || (depth_paths && depth_stride < (int64_t)H * W * 4))
    return RR_E_ARG;
  rrpar::parallel_for(n, threads, [&](int k) {
    int rc;
    try {
      rc = image_paths[k] ? rr_png_read_bgr8_impl(image_paths[k], bg_u8 + (size_t)k * (size_t)bg_stride, H, W) : RR_E_ARG;
      if (rc == RR_OK && depth_paths) {
        if (!depth_paths[k]) {
          rc = RR_E_ARG;
        } else {
          // depth = cv2.imread(f, IMREAD_UNCHANGED).astype(np.float32) / 256.   (generator.py:360-365; exact: a power of two)
          Pool<DecodeScratch>::Lease sc(g_decode_pool);      // (only its d16 buffer: the reader below leases its own set)
          std::vector<uint16_t>& d16 = sc->d16;
          at_least(d16, (size_t)H * W);
          rc = rr_png_read_gray16_impl(depth_paths[k], d16.data(), H, W);
          if (rc == RR_OK) {
            float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(depth_f32) + (size_t)k * (size_t)depth_stride);
            for (size_t i = 0; i < (size_t)H * W; i++) o[i] = (float)d16[i] / 256.0f;
          }
        }
      }
    } catch (...) {
      rc = RR_E_PARSE;
    }
    status[k] = rc;
  });
  return RR_OK;
}

// cv2.resize(img, (dw, dh)) for float images as the driver states it (common/imgops.resize_linear: INTER_LINEAR, half-pixel
// centres, edge clamp, float64): source index and weight of every destination row / column ...
extern "C" int rr_io_read_frames_u16(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                                     uint8_t* bg_u8, int64_t bg_stride, uint16_t* depth_u16, int64_t depth_stride, int32_t threads,
                                     int32_t* status) {
  if (n < 0 || H <= 0 || W <= 0 || !status || (n > 0 && (!image_paths || !bg_u8)) || (depth_paths && !depth_u16) ||
      bg_stride < (int64_t)H * W * 3 || (depth_paths && depth_stride < (int64_t)H * W * 2))
    return RR_E_ARG;
  rrpar::parallel_for(n, threads, [&](int k) {
    int rc;
    try {
      rc = image_paths[k] ? rr_png_read_bgr8_impl(image_paths[k], bg_u8 + (size_t)k * (size_t)bg_stride, H, W) : RR_E_ARG;
      if (rc == RR_OK && depth_paths)     // the samples as cv2.imread(f, IMREAD_UNCHANGED) returns them; / 256 happens on the device
        rc = depth_paths[k] ? rr_png_read_gray16_impl(depth_paths[k], reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(depth_u16) + (size_t)k * (size_t)depth_stride), H, W)
                            : RR_E_ARG;
    } catch (...) {
      rc = RR_E_PARSE;
    }
    status[k] = rc;
  });
  return RR_OK;
}

// One file as filtered scanlines for the device (rr_io_read_frames_rows): H rows of 1 + bpp * W bytes.  A file of exactly the
// expected kind (8-bit RGB for bpp 3, 16-bit gray for bpp 2; not interlaced) hands over what its IDAT stream inflates to,
// filter types checked; any other file the readers above accept is decoded here and laid out as rows of filter type 0 in
// PNG sample order (R G B / big-endian), so that the device sees one format.
static int read_rows_impl(const char* path, uint8_t* rows, int32_t H, int32_t W, int bpp) {
  if (!path || !rows) return RR_E_ARG;
  const size_t rb = 1 + (size_t)bpp * W;
  {
    Pool<DecodeScratch>::Lease sc(g_decode_pool);
    size_t fsize = 0;
    int rc = read_file(path, sc->file, fsize);
    if (rc) return rc;
    Png& p = sc->png;
    if ((rc = parse_chunks(sc->file, fsize, p, true))) return rc;
    if ((int32_t)p.w != W || (int32_t)p.h != H) return RR_E_ARG;
    const bool native = !p.interlace && ((bpp == 3 && p.ctype == 2 && p.depth == 8) || (bpp == 2 && p.ctype == 0 && p.depth == 16));
    if (native) {
      size_t stride = 0;
      if ((rc = inflate_rows(p, *sc, stride))) return rc;
      const uint8_t* raw = sc->raw.data();
      for (int y = 0; y < H; y++)
        if (raw[rb * (size_t)y] > 4) return RR_E_PARSE;
      memcpy(rows, raw, rb * (size_t)H);
      return RR_OK;
    }
  }
  if (bpp == 3) {
    std::vector<uint8_t> bgr((size_t)H * W * 3);
    int rc = rr_png_read_bgr8_impl(path, bgr.data(), H, W);
    if (rc) return rc;
    for (int y = 0; y < H; y++) {
      uint8_t* o = rows + rb * (size_t)y;
      const uint8_t* s = &bgr[(size_t)y * W * 3];
      *o++ = 0;
      for (int x = 0; x < W; x++, s += 3, o += 3) { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; }
    }
    return RR_OK;
  }
  std::vector<uint16_t> d16((size_t)H * W);
  int rc = rr_png_read_gray16_impl(path, d16.data(), H, W);
  if (rc) return rc;
  for (int y = 0; y < H; y++) {
    uint8_t* o = rows + rb * (size_t)y;
    *o++ = 0;
    for (int x = 0; x < W; x++, o += 2) { const uint16_t v = d16[(size_t)y * W + x]; o[0] = (uint8_t)(v >> 8); o[1] = (uint8_t)v; }
  }
  return RR_OK;
}

extern "C" int rr_io_read_frames_rows(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                                      uint8_t* image_rows, int64_t image_stride, uint8_t* depth_rows, int64_t depth_stride, int32_t threads,
                                      int32_t* status) {
  if (n < 0 || H <= 0 || W <= 0 || !status || (n > 0 && (!image_paths || !image_rows)) || (depth_paths && !depth_rows) ||
      image_stride < (int64_t)H * (1 + 3 * (int64_t)W) || (depth_paths && depth_stride < (int64_t)H * (1 + 2 * (int64_t)W)))
    return RR_E_ARG;
  rrpar::parallel_for(n, threads, [&](int k) {
    int rc;
    try {
      rc = image_paths[k] ? read_rows_impl(image_paths[k], image_rows + (size_t)k * (size_t)image_stride, H, W, 3) : RR_E_ARG;
      if (rc == RR_OK && depth_paths) rc = depth_paths[k] ? read_rows_impl(depth_paths[k], depth_rows + (size_t)k * (size_t)depth_stride, H, W, 2) : RR_E_ARG;
    } catch (...) {
      rc = RR_E_PARSE;
    }
    status[k] = rc;
  });
  return RR_OK;
}

static void linear_coords(int d, int s, std::vector<int32_t>& i0, std::vector<int32_t>& i1, std::vector<double>& w) {
  at_least(i0, (size_t)d);
  at_least(i1, (size_t)d);
  at_least(w, (size_t)d);
  const double scale = (double)s / (double)d;
  for (int k = 0; k < d; k++) {
    const double f = ((double)k + 0.5) * scale - 0.5;
    const double fl = std::floor(f);
    const int64_t a = (int64_t)fl;
    w[(size_t)k] = a < 0 ? 0.0 : f - fl;
    const int64_t b = a + 1;
    i0[(size_t)k] = (int32_t)(a < 0 ? 0 : (a > s - 1 ? s - 1 : a));
    i1[(size_t)k] = (int32_t)(b < 0 ? 0 : (b > s - 1 ? s - 1 : b));
  }
}
// ... and one output sample: (a00 * (1 - wx) + a01 * wx) * (1 - wy) + (a10 * (1 - wx) + a11 * wx) * wy, the products and sums
// in numpy's order
static inline double bilinear(double a00, double a01, double a10, double a11, double wx, double wy) {
  const double ux = 1 - wx, uy = 1 - wy;
  const double top = a00 * ux + a01 * wx, bot = a10 * ux + a11 * wx;
  return top * uy + bot * wy;
}
struct UnitLut {                                       // byte / 255.0 (generator.py:352), as numpy divides
  double v[256];
  UnitLut() {
    for (int i = 0; i < 256; i++) v[i] = (double)i / 255.0;
  }
};

// The frame loader of Generator.run for a render scale other than 1 (generator.py:352-381; the Cityscapes plug-in's
// default): image / 255 resized to (W, H) = file size // render_scale as float64, the depth map (metres) resized to
// (its size * depth_scale) // render_scale when that differs from its own size.  Depth and image must end up H x W (the
// reference would crop the image otherwise: RR_E_ARG, the caller's general loader handles such a frame).
extern "C" int rr_io_read_frames_scaled(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                                        int32_t render_scale, int32_t depth_scale, double* bg_f64, int64_t bg_stride, float* depth_f32,
                                        int64_t depth_stride, int32_t threads, int32_t* status) {
  if (n < 0 || H <= 0 || W <= 0 || render_scale < 1 || depth_scale < 1 || !status || (n > 0 && (!image_paths || !bg_f64)) ||
      (depth_paths && !depth_f32) || bg_stride < (int64_t)H * W * 24 || (depth_paths && depth_stride < (int64_t)H * W * 4))
    return RR_E_ARG;
  static const UnitLut lut;
  rrpar::parallel_for(n, threads, [&](int k) {
    int rc = RR_OK;
    try {
      Pool<DecodeScratch>::Lease sc(g_decode_pool);     // (the readers below lease their own sets: this one holds the temporaries)
      int32_t sw = 0, sh = 0, ch = 0, bits = 0;
      if (!image_paths[k] || (rc = rr_png_info_impl(image_paths[k], &sw, &sh, &ch, &bits)) != RR_OK) {
        status[k] = rc ? rc : RR_E_ARG;
        return;
      }
      if (sw / render_scale != W || sh / render_scale != H) {
        status[k] = RR_E_ARG;
        return;
      }
      std::vector<uint8_t>& bgr = sc->bgr;
      at_least(bgr, (size_t)sh * sw * 3);
      if ((rc = rr_png_read_bgr8_impl(image_paths[k], bgr.data(), sh, sw)) != RR_OK) {
        status[k] = rc;
        return;
      }
      double* out = reinterpret_cast<double*>(reinterpret_cast<char*>(bg_f64) + (size_t)k * (size_t)bg_stride);
      if (sh == H && sw == W) {
        for (size_t i = 0; i < (size_t)H * W * 3; i++) out[i] = lut.v[bgr[i]];
      } else {
        linear_coords(H, sh, sc->y0, sc->y1, sc->wy);
        linear_coords(W, sw, sc->x0, sc->x1, sc->wx);
        for (int y = 0; y < H; y++) {
          const uint8_t* r0 = bgr.data() + (size_t)sc->y0[(size_t)y] * sw * 3;
          const uint8_t* r1 = bgr.data() + (size_t)sc->y1[(size_t)y] * sw * 3;
          const double wy = sc->wy[(size_t)y];
          double* o = out + (size_t)y * W * 3;
          for (int x = 0; x < W; x++) {
            const size_t xa = (size_t)sc->x0[(size_t)x] * 3, xb = (size_t)sc->x1[(size_t)x] * 3;
            const double wx = sc->wx[(size_t)x];
            for (int c = 0; c < 3; c++)
              o[(size_t)x * 3 + c] = bilinear(lut.v[r0[xa + c]], lut.v[r0[xb + c]], lut.v[r1[xa + c]], lut.v[r1[xb + c]], wx, wy);
          }
        }
      }
      if (depth_paths) {
        int32_t dw = 0, dh = 0;
        if (!depth_paths[k] || (rc = rr_png_info_impl(depth_paths[k], &dw, &dh, &ch, &bits)) != RR_OK) {
          status[k] = rc ? rc : RR_E_ARG;
          return;
        }
        const int64_t th = ((int64_t)dh * depth_scale) / render_scale, tw = ((int64_t)dw * depth_scale) / render_scale;
        if (th != H || tw != W) {
          status[k] = RR_E_ARG;
          return;
        }
        std::vector<uint16_t>& d16 = sc->d16;
        at_least(d16, (size_t)dh * dw);
        if ((rc = rr_png_read_gray16_impl(depth_paths[k], d16.data(), dh, dw)) != RR_OK) {
          status[k] = rc;
          return;
        }
        float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(depth_f32) + (size_t)k * (size_t)depth_stride);
        if (dh == H && dw == W) {
          for (size_t i = 0; i < (size_t)H * W; i++) o[i] = (float)d16[i] / 256.0f;
        } else {                                       // cv2.resize keeps float32: float64 inside resize_linear, rounded once
          linear_coords(H, dh, sc->y0, sc->y1, sc->wy);
          linear_coords(W, dw, sc->x0, sc->x1, sc->wx);
          auto m = [&](size_t idx) { return (double)((float)d16[idx] / 256.0f); };
          for (int y = 0; y < H; y++) {
            const size_t r0 = (size_t)sc->y0[(size_t)y] * dw, r1 = (size_t)sc->y1[(size_t)y] * dw;
            const double wy = sc->wy[(size_t)y];
            for (int x = 0; x < W; x++) {
              const size_t xa = (size_t)sc->x0[(size_t)x], xb = (size_t)sc->x1[(size_t)x];
              o[(size_t)y * W + x] = (float)bilinear(m(r0 + xa), m(r0 + xb), m(r1 + xa), m(r1 + xb), sc->wx[(size_t)x], wy);
            }
          }
        }
      }
    } catch (...) {
      rc = RR_E_PARSE;
    }
    status[k] = rc;
  });
  return RR_OK;
}

extern "C" int rr_io_write_frames(int32_t n, const char* const* image_paths, const char* const* mask_paths, const uint8_t* rows_image,
                                  const uint8_t* rows_mask, int64_t rows_stride, int32_t W, int32_t H, int32_t threads, int32_t* status) {
  if (n < 0 || H <= 0 || W <= 0 || !status || (image_paths && !rows_image) || (mask_paths && !rows_mask) ||
      rows_stride < (int64_t)H * (1 + 4 * (int64_t)W))
    return RR_E_ARG;
  // two jobs per frame (the files are independent): 2n items keep every thread busy to the end of a batch
  std::vector<int32_t> rc2;
  try {
    rc2.assign((size_t)2 * (size_t)(n > 0 ? n : 0), RR_OK);
  } catch (...) {
    return RR_E_PARSE;
  }
  rrpar::parallel_for(2 * n, threads, [&](int j) {
    const int k = j >> 1;
    const bool mask = j & 1;
    const char* path = mask ? (mask_paths ? mask_paths[k] : nullptr) : (image_paths ? image_paths[k] : nullptr);
    if (!path) return;
    const uint8_t* rows = (mask ? rows_mask : rows_image) + (size_t)k * (size_t)rows_stride;
    int rc;
    try {
      rc = rr_png_write_scanlines_impl(path, rows, W, H, 1, 3);
    } catch (...) {
      rc = RR_E_PARSE;
    }
    rc2[(size_t)j] = rc;
  });
  for (int k = 0; k < n; k++) status[k] = rc2[2 * (size_t)k] ? rc2[2 * (size_t)k] : rc2[2 * (size_t)k + 1];
  return RR_OK;
}

// the writer's / reader's checksums (tests compare them with zlib's)
extern "C" uint32_t rr_adler32(uint32_t adler, const uint8_t* p, int64_t n) { return (p && n > 0) ? fast_adler32(adler, p, (size_t)n) : adler; }
extern "C" uint32_t rr_crc32(uint32_t crc, const uint8_t* p, int64_t n) { return (p && n > 0) ? fast_crc32(crc, p, (size_t)n) : crc; }
extern "C" int64_t rr_deflate_fast(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap) {
  try {
    return rr_deflate_fast_impl(in, n, out, cap);
  } catch (...) {
    return RR_E_PARSE;
  }
}
extern "C" int rr_inflate_fast(const uint8_t* in, int64_t n, uint8_t* out, int64_t out_len) {
  try {
    return rr_inflate_fast_impl(in, n, out, out_len);
  } catch (...) {
    return RR_E_PARSE;
  }
}
