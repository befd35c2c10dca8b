// rr_deflate.h -- entropy coding of the two output PNG files on the device (SURVEY 8f "next" #3, the writer's side;
// the reference saves both with plt.imsave, generator.py:466-467).
// k_png_image / k_png_mask leave the filtered scanlines of a file in HBM: H rows of 1 + 4 W bytes.  This header turns them
// into the zlib stream of the file's IDAT chunk, so that the host only frames chunks and computes one CRC:
//   * the scanlines are cut into blocks of 32 KB, one workgroup of 256 threads each; a thread owns a span of 128 bytes
//   * tokens of a span: its first byte is a literal; a byte that repeats its predecessor at least three times in a row starts
//     a run (length 3 .. the rest of the span, distance 1), anything else is a literal -- the rule of the host encoder
//     (rr_png.cpp fast_deflate, like zlib's Z_RLE), except that a run ends with its span, so that spans are independent
//   * one dynamic-Huffman deflate block per 32 KB: histogram (LDS atomics), symbols ranked by (count, symbol) in parallel,
//     the two-queue tree construction by one thread (the only serial step: <= 286 iterations), depths, the count-based fix-up
//     to 15 bits, canonical codes; the code lengths themselves Huffman-coded without repeat symbols
//   * bit positions by a workgroup scan of the spans' bit counts, every span's codes OR-ed into the block's buffer in LDS
//   * a block that is not the file's last ends with an empty stored block (the "sync flush" of zlib / pigz), so every block
//     starts on a byte boundary and the blocks of a file are simply laid behind each other (k_pngz_pack); a block whose
//     dynamic form would be larger than its bytes is a stored block
//   * Adler-32: per block (sum of bytes, position-weighted sum), combined by the pack kernel.
// Result in the file's scanline buffer (capacity H * (1 + 4 W), unchanged): a 16-byte header {PNGZ_MAGIC, length of the zlib
// stream, 0, 0} followed by the stream -- if that fits (always, unless the image is incompressible: the first byte of a
// scanline buffer is a PNG filter type 0..4, never the magic's 'R').  Otherwise the scanlines are left as they are and the
// host compresses them as before.  rr_png_write_scanlines / rr_io_write_frames (rr_png.cpp) take either.
// The functions are `__host__ __device__` in "thread role" form (tid = 0..255, barriers between the phases) like FogTile in
// rr_prepass.h: tests/hostemu runs the same code on the CPU and zlib inflates what it produces (tests/test_deflate_hostemu.py).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RRZ_HD __host__ __device__ inline
#else
#define RRZ_HD inline
#endif

namespace rrz {

constexpr int SPAN = 128, NT = 256, BLOCK = SPAN * NT;
constexpr int NSYM = 288;            // literals 0..255, end of block 256, lengths 257..285 (286, 287 never used)
constexpr int MAXL = 15;
constexpr int OUT_WORDS = BLOCK / 4 + 16;
constexpr uint32_t PNGZ_MAGIC = 0x315a5252u;          // the bytes 'R' 'R' 'Z' '1'
constexpr int PNGZ_HEADER = 16;
constexpr int SLOT_BYTES = BLOCK + 64;                // one block's compressed form in the scratch (a stored block: BLOCK + 5)

#if defined(__HIP_DEVICE_COMPILE__)
#define RRZ_OR(p, v) atomicOr((p), (v))
#define RRZ_ADD(p, v) atomicAdd((p), (v))
#define RRZ_MAX(p, v) atomicMax((p), (v))
#else
#define RRZ_OR(p, v) (*(p) |= (v))
#define RRZ_ADD(p, v) (*(p) += (v))
#define RRZ_MAX(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#endif

struct BlockMeta {                   // per block, in HBM: what the pack kernel needs
  uint32_t bytes;                    // length of the block's compressed form
  uint32_t len;                      // uncompressed bytes
  uint32_t s1, s2;                   // sum b[i] and sum (len - i) * b[i], both mod 65521
};

struct BlockState {                  // one workgroup's working set: LDS on the device
  uint8_t in[BLOCK];
  uint32_t out[OUT_WORDS];
  uint32_t freq[NSYM];
  uint32_t w[2 * NSYM];              // tree: weights of the leaves (in sorted order) and of the inner nodes
  uint32_t scan[2][NT];              // workgroup scan (double-buffered Hillis-Steele)
  uint32_t cnt[MAXL + 1];            // symbols per code length
  uint32_t ad1[NT], ad2[NT];
  uint32_t own[NT];                  // bits of each span's tokens
  uint16_t sorted[NSYM];             // used symbols by rising (count, symbol)
  uint16_t parent[2 * NSYM];
  uint16_t code[NSYM];               // canonical code, bit-reversed (deflate packs codes most significant bit first)
  uint16_t cc[19];
  uint8_t len[NSYM];
  uint8_t cl[19];
  int32_t n, last, m, nlit, ncl, stored;
  uint32_t hdr_fixed, hdr_var, data_bits, bytes;
};

// length symbol of a run of r bytes (3 <= r <= 257): symbol, number of extra bits, their value (RFC 1951 3.2.5)
RRZ_HD void length_code(int r, int& sym, int& ebits, int& eval) {
  const int x = r - 3;
  if (x < 8) {
    sym = 257 + x;
    ebits = 0;
    eval = 0;
    return;
  }
  int lg = 0;
  while ((x >> (lg + 1)) != 0) lg++;                    // floor(log2 x), 3..7
  ebits = lg - 2;
  sym = 265 + 4 * (ebits - 1) + ((x >> ebits) - 4);
  eval = x & ((1 << ebits) - 1);
}

// the tokens of a span: f(value, is_run) -- a literal byte, or a run of `value` bytes at distance 1
template <class F>
RRZ_HD void for_tokens(const uint8_t* s, int n, F&& f) {
  if (n <= 0) return;
  f((int)s[0], 0);
  int i = 1;
  while (i < n) {
    const uint8_t b = s[i];
    if (b == s[i - 1] && i + 2 < n && s[i + 1] == b && s[i + 2] == b) {
      int r = 3;
      while (i + r < n && s[i + r] == b) r++;           // (a span has 128 bytes: r <= 127 < 258)
      f(r, 1);
      i += r;
    } else {
      f((int)b, 0);
      i++;
    }
  }
}

RRZ_HD uint32_t bit_reverse(uint32_t v, int n) {
  uint32_t r = 0;
  for (int k = 0; k < n; k++) r |= ((v >> k) & 1u) << (n - 1 - k);
  return r;
}

// LSB-first bit writer of one thread over the block's buffer: the words at both ends of its bit range may be shared with the
// neighbouring ranges (OR), the words in between are its own
struct BitOut {
  uint32_t* out;
  uint64_t acc;
  int nacc;
  uint32_t wpos;
  bool first;
  RRZ_HD BitOut(uint32_t* o, uint32_t bitpos) : out(o), acc(0), nacc((int)(bitpos & 31u)), wpos(bitpos >> 5), first(true) {}
  RRZ_HD void put(uint32_t bits, int n) {               // n <= 32
    acc |= (uint64_t)bits << nacc;
    nacc += n;
    if (nacc >= 32) {
      if (first) RRZ_OR(out + wpos, (uint32_t)acc);
      else out[wpos] = (uint32_t)acc;
      first = false;
      wpos++;
      acc >>= 32;
      nacc -= 32;
    }
  }
  RRZ_HD void finish() {
    if (nacc > 0) RRZ_OR(out + wpos, (uint32_t)acc);
  }
};

// ---- the phases of one block (every function: one thread role; a barrier after each) --------------------------------------
// P0: the block's bytes into S.in (the caller copies them: coalesced on the device), then: clear, Adler partials
RRZ_HD void p0_init(BlockState& S, int tid, int n, int last) {
  for (int k = tid; k < OUT_WORDS; k += NT) S.out[k] = 0;
  for (int k = tid; k < NSYM; k += NT) {
    S.freq[k] = 0;
    S.len[k] = 0;
    S.code[k] = 0;
  }
  if (tid <= MAXL) S.cnt[tid] = 0;
  if (tid == 0) {
    S.n = n;
    S.last = last;
    S.m = 0;
    S.nlit = 257;
    S.stored = 0;
  }
  const int a = tid * SPAN, b = a + SPAN < n ? a + SPAN : n;
  uint32_t s1 = 0, s2 = 0;
  for (int i = a; i < b; i++) {
    s1 += S.in[i];
    s2 += (uint32_t)(n - i) * S.in[i];                  // <= 128 * 32768 * 255 < 2^31
  }
  S.ad1[tid] = s1;
  S.ad2[tid] = s2 % 65521u;
}
// P1: histogram of the tokens
RRZ_HD void p1_hist(BlockState& S, int tid) {
  const int a = tid * SPAN, b = a + SPAN < S.n ? a + SPAN : S.n;
  for_tokens(S.in + a, b - a, [&](int v, int run) {
    if (run) {
      int sym, eb, ev;
      length_code(v, sym, eb, ev);
      RRZ_ADD(&S.freq[sym], 1u);
    } else {
      RRZ_ADD(&S.freq[v], 1u);
    }
  });
  if (tid == 0) RRZ_ADD(&S.freq[256], 1u);              // end of block
}
// P2: rank of every used symbol among the used ones by (count, symbol)
RRZ_HD void p2_rank(BlockState& S, int tid) {
  for (int s = tid; s < NSYM; s += NT) {
    const uint32_t f = S.freq[s];
    if (!f) continue;
    int r = 0;
    for (int j = 0; j < NSYM; j++) {
      const uint32_t g = S.freq[j];
      r += (g != 0 && (g < f || (g == f && j < s))) ? 1 : 0;
    }
    S.sorted[r] = (uint16_t)s;
    RRZ_ADD(&S.m, 1);
  }
}
// P3 (thread 0): two-queue Huffman construction over the sorted leaves; node i < m is leaf sorted[i], the root is 2m - 2
RRZ_HD void p3_tree(BlockState& S, int tid) {
  if (tid != 0) return;
  const int m = S.m;                                      // >= 2: a block has at least one byte and the end-of-block symbol
  for (int i = 0; i < m; i++) S.w[i] = S.freq[S.sorted[i]];
  int leaf = 0, inner = m, made = m;
  while (made < 2 * m - 1) {
    int pick[2];
    for (int k = 0; k < 2; k++) {
      if (leaf < m && (inner >= made || S.w[leaf] <= S.w[inner])) pick[k] = leaf++;
      else pick[k] = inner++;
    }
    S.w[made] = S.w[pick[0]] + S.w[pick[1]];
    S.parent[pick[0]] = (uint16_t)made;
    S.parent[pick[1]] = (uint16_t)made;
    made++;
  }
}
// P4: depth of every leaf (hops to the root), counted per length with the clamp to MAXL
RRZ_HD void p4_depth(BlockState& S, int tid) {
  const int m = S.m, root = 2 * m - 2;
  for (int i = tid; i < m; i += NT) {
    int d = 0, v = i;
    while (v != root) {
      v = S.parent[v];
      d++;
    }
    RRZ_ADD(&S.cnt[d > MAXL ? MAXL : d], 1u);
  }
}
// count-based fix-up of a Kraft sum that the clamp made too large: lengthen the cheapest codes (as rr_png.cpp huffman_lengths)
RRZ_HD void kraft_fix(uint32_t* cnt, int max_len) {
  uint64_t total = 0;
  for (int l = 1; l <= max_len; l++) total += (uint64_t)cnt[l] << (max_len - l);
  while (total > (1ull << max_len)) {
    cnt[max_len]--;
    for (int l = max_len - 1; l >= 1; l--)
      if (cnt[l]) {
        cnt[l]--;
        cnt[l + 1] += 2;
        break;
      }
    total--;
  }
}
RRZ_HD void p5_limit(BlockState& S, int tid) {
  if (tid == 0) kraft_fix(S.cnt, MAXL);
}
// P6: lengths by sorted position: the rarest symbols get the longest codes
RRZ_HD void p6_assign(BlockState& S, int tid) {
  for (int i = tid; i < S.m; i += NT) {
    int l = MAXL;
    uint32_t acc = S.cnt[l];
    while ((uint32_t)i >= acc) {
      l--;
      acc += S.cnt[l];
    }
    const int s = S.sorted[i];
    S.len[s] = (uint8_t)l;
    if (s >= 257) RRZ_MAX(&S.nlit, s + 1);
  }
}
// P7: canonical codes (RFC 1951 3.2.2): first code of a length from the counts, then the symbol's place among its length
RRZ_HD void p7_codes(BlockState& S, int tid) {
  for (int s = tid; s < NSYM; s += NT) {
    const int L = S.len[s];
    if (!L) continue;
    uint32_t c = 0;
    for (int b = 1; b <= L; b++) c = (c + (b > 1 ? S.cnt[b - 1] : 0u)) << 1;
    uint32_t idx = 0;
    for (int j = 0; j < s; j++) idx += S.len[j] == L ? 1u : 0u;
    S.code[s] = (uint16_t)bit_reverse(c + idx, L);
  }
}
// serial Huffman lengths of a small alphabet (the 19 code-length symbols, limit 7)
RRZ_HD void small_lengths(const uint32_t* freq, int nsym, int max_len, uint8_t* len) {
  int used[19], m = 0;
  for (int i = 0; i < nsym; i++) {
    len[i] = 0;
    if (freq[i]) used[m++] = i;
  }
  if (m == 0) return;
  if (m == 1) {
    len[used[0]] = 1;
    return;
  }
  for (int i = 1; i < m; i++) {                          // insertion sort by (count, symbol)
    const int u = used[i];
    int j = i - 1;
    while (j >= 0 && (freq[used[j]] > freq[u] || (freq[used[j]] == freq[u] && used[j] > u))) {
      used[j + 1] = used[j];
      j--;
    }
    used[j + 1] = u;
  }
  uint32_t w[38];
  int parent[38];
  for (int i = 0; i < m; i++) w[i] = freq[used[i]];
  int leaf = 0, inner = m, made = m;
  while (made < 2 * m - 1) {
    int pick[2];
    for (int k = 0; k < 2; k++) {
      if (leaf < m && (inner >= made || w[leaf] <= w[inner])) pick[k] = leaf++;
      else pick[k] = inner++;
    }
    w[made] = w[pick[0]] + w[pick[1]];
    parent[pick[0]] = parent[pick[1]] = made;
    made++;
  }
  uint32_t cnt[16] = {0};
  for (int i = 0; i < m; i++) {
    int d = 0, v = i;
    while (v != 2 * m - 2) {
      v = parent[v];
      d++;
    }
    cnt[d > max_len ? max_len : d]++;
  }
  kraft_fix(cnt, max_len);
  int l = max_len;
  for (int i = 0; i < m; i++) {
    while (l > 0 && cnt[l] == 0) l--;
    len[used[i]] = (uint8_t)l;
    cnt[l]--;
  }
}
RRZ_HD void small_codes(const uint8_t* len, int nsym, uint16_t* code) {
  int bl[16] = {0};
  for (int i = 0; i < nsym; i++) bl[len[i]]++;
  bl[0] = 0;
  int next[16] = {0}, c = 0;
  for (int b = 1; b <= 15; b++) {
    c = (c + bl[b - 1]) << 1;
    next[b] = c;
  }
  for (int i = 0; i < nsym; i++) code[i] = len[i] ? (uint16_t)bit_reverse((uint32_t)next[len[i]]++, len[i]) : 0;
}
RRZ_HD int cl_order(int k) {
  const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  return order[k];
}
// P8 (thread 0): the code of the code lengths; the fixed part of the block header goes out
RRZ_HD void p8_header(BlockState& S, int tid) {
  if (tid != 0) return;
  uint32_t fc[19];
  for (int k = 0; k < 19; k++) fc[k] = k <= MAXL ? S.cnt[k] : 0u;
  fc[0] = (uint32_t)(S.nlit - S.m);                       // the unused symbols below nlit
  fc[1] += 1;                                             // the one distance code (distance 1), length 1
  small_lengths(fc, 19, 7, S.cl);
  small_codes(S.cl, 19, S.cc);
  int ncl = 19;
  while (ncl > 4 && S.cl[cl_order(ncl - 1)] == 0) ncl--;
  S.ncl = ncl;
  BitOut bo(S.out, 0);
  bo.put(S.last ? 1u : 0u, 1);
  bo.put(2u, 2);
  bo.put((uint32_t)(S.nlit - 257), 5);
  bo.put(0u, 5);                                          // HDIST: one distance code
  bo.put((uint32_t)(ncl - 4), 4);
  for (int k = 0; k < ncl; k++) bo.put(S.cl[cl_order(k)], 3);
  bo.finish();
  S.hdr_fixed = 17u + 3u * (uint32_t)ncl;
}
// bit count of entry k of the code-length sequence (the nlit literal/length code lengths, then the distance code's)
RRZ_HD uint32_t hdr_entry_bits(const BlockState& S, int k) { return k < S.nlit ? S.cl[S.len[k]] : (k == S.nlit ? S.cl[1] : 0u); }
// P9a: scan input = bits of the two entries 2 tid, 2 tid + 1
RRZ_HD void p9_hdr_bits(BlockState& S, int tid) { S.scan[0][tid] = hdr_entry_bits(S, 2 * tid) + hdr_entry_bits(S, 2 * tid + 1); }
// workgroup scan, step s = 0..7 (inclusive Hillis-Steele from buffer s & 1 to the other); the result is in scan[0]
RRZ_HD void scan_step(BlockState& S, int tid, int s) {
  const uint32_t* a = S.scan[s & 1];
  uint32_t* b = S.scan[(s & 1) ^ 1];
  const int d = 1 << s;
  b[tid] = a[tid] + (tid >= d ? a[tid - d] : 0u);
}
// P9b: the entries' codes at their positions (scan[0] = inclusive sums)
RRZ_HD void p9_hdr_emit(BlockState& S, int tid) {
  const uint32_t incl = S.scan[0][tid];
  const uint32_t b0 = hdr_entry_bits(S, 2 * tid), b1 = hdr_entry_bits(S, 2 * tid + 1);
  if (b0 + b1) {
    BitOut bo(S.out, S.hdr_fixed + incl - b0 - b1);
    if (b0) bo.put(2 * tid < S.nlit ? S.cc[S.len[2 * tid]] : S.cc[1], (int)b0);
    if (b1) bo.put(2 * tid + 1 < S.nlit ? S.cc[S.len[2 * tid + 1]] : S.cc[1], (int)b1);
    bo.finish();
  }
  if (tid == NT - 1) S.hdr_var = incl;
}
// P10a: bits of a span's tokens
RRZ_HD void p10_span_bits(BlockState& S, int tid) {
  const int a = tid * SPAN, b = a + SPAN < S.n ? a + SPAN : S.n;
  uint32_t bits = 0;
  for_tokens(S.in + a, b - a, [&](int v, int run) {
    if (run) {
      int sym, eb, ev;
      length_code(v, sym, eb, ev);
      bits += (uint32_t)S.len[sym] + (uint32_t)eb + 1u;  // + the distance code: one bit
    } else {
      bits += S.len[v];
    }
  });
  S.scan[0][tid] = bits;
  S.own[tid] = bits;
}
// P10b (after the scan): size of the dynamic form; a block that does not shrink is stored
RRZ_HD void p10_decide(BlockState& S, int tid) {
  if (tid != 0) return;
  S.data_bits = S.scan[0][NT - 1];
  const uint64_t bits = (uint64_t)S.hdr_fixed + S.hdr_var + S.data_bits + S.len[256];
  // (not the last block: three header bits of the empty stored block, padding, then LEN and NLEN)
  uint32_t bytes = S.last ? (uint32_t)((bits + 7) >> 3) : (uint32_t)((bits + 3 + 7) >> 3) + 4u;
  if (bytes >= (uint32_t)S.n + 5u) {
    S.stored = 1;
    bytes = (uint32_t)S.n + 5u;
  }
  S.bytes = bytes;
}
// P11 (stored blocks only): the header bits written so far are dropped
RRZ_HD void p11_clear(BlockState& S, int tid) {
  if (!S.stored) return;
  for (int k = tid; k < OUT_WORDS; k += NT) S.out[k] = 0;
}
// P12: the block's data
RRZ_HD void p12_emit(BlockState& S, int tid) {
  if (S.stored) {
    uint8_t* ob = reinterpret_cast<uint8_t*>(S.out);
    if (tid == 0) {
      ob[0] = S.last ? 1 : 0;                             // BFINAL, BTYPE 00, padding
      ob[1] = (uint8_t)(S.n & 255);
      ob[2] = (uint8_t)(S.n >> 8);
      ob[3] = (uint8_t)(~S.n & 255);
      ob[4] = (uint8_t)((~S.n >> 8) & 255);
    }
    return;                                               // (the bytes follow in p12b: other words than the header's)
  }
  const int a = tid * SPAN, b = a + SPAN < S.n ? a + SPAN : S.n;
  const uint32_t incl = S.scan[0][tid], own = S.own[tid];
  const uint32_t base = S.hdr_fixed + S.hdr_var;
  if (own) {
    BitOut bo(S.out, base + incl - own);
    for_tokens(S.in + a, b - a, [&](int v, int run) {
      if (run) {
        int sym, eb, ev;
        length_code(v, sym, eb, ev);
        bo.put(S.code[sym], S.len[sym]);
        if (eb) bo.put((uint32_t)ev, eb);
        bo.put(0u, 1);                                    // distance 1: the only distance code, one bit
      } else {
        bo.put(S.code[v], S.len[v]);
      }
    });
    bo.finish();
  }
  if (tid == 0) {                                         // end of block, then -- unless the file ends here -- the empty stored block
    const uint32_t endpos = base + S.data_bits;
    BitOut bo(S.out, endpos);
    bo.put(S.code[256], S.len[256]);
    bo.finish();
    if (!S.last) {
      const uint32_t after = endpos + S.len[256] + 3;     // three zero bits: BFINAL 0, BTYPE 00
      const uint32_t at = (after + 7) >> 3;               // LEN 0000, NLEN ffff on the next byte boundary
      RRZ_OR(&S.out[(at + 2) >> 2], 0xffu << (8 * ((at + 2) & 3)));
      RRZ_OR(&S.out[(at + 3) >> 2], 0xffu << (8 * ((at + 3) & 3)));
    }
  }
}
// P12b (stored blocks): the raw bytes behind the five header bytes
RRZ_HD void p12b_stored_bytes(BlockState& S, int tid) {
  if (!S.stored) return;
  uint8_t* ob = reinterpret_cast<uint8_t*>(S.out) + 5;
  const int a = tid * SPAN, b = a + SPAN < S.n ? a + SPAN : S.n;
  for (int i = a; i < b; i++) ob[i] = S.in[i];
}
// P13 (thread 0): the block's record
RRZ_HD void p13_meta(BlockState& S, int tid, BlockMeta* meta) {
  if (tid != 0) return;
  uint64_t s1 = 0, s2 = 0;
  for (int k = 0; k < NT; k++) {
    s1 += S.ad1[k];
    s2 += S.ad2[k];
  }
  meta->bytes = S.bytes;
  meta->len = (uint32_t)S.n;
  meta->s1 = (uint32_t)(s1 % 65521u);
  meta->s2 = (uint32_t)(s2 % 65521u);
}

// ---- the file level -------------------------------------------------------------------------------------------------------
RRZ_HD int64_t blocks_of(int64_t n) { return (n + BLOCK - 1) / BLOCK; }
// length of the zlib stream of a file: 2 header bytes, the blocks, Adler-32
RRZ_HD int64_t stream_bytes(const BlockMeta* meta, int nb) {
  int64_t t = 2 + 4;
  for (int k = 0; k < nb; k++) t += meta[k].bytes;
  return t;
}
RRZ_HD uint32_t adler_of(const BlockMeta* meta, int nb) {
  uint64_t A = 1, B = 0;
  for (int k = 0; k < nb; k++) {
    B = (B + (uint64_t)meta[k].len * A + meta[k].s2) % 65521u;
    A = (A + meta[k].s1) % 65521u;
  }
  return (uint32_t)((B << 16) | A);
}
// the 16-byte header, the zlib header and the trailer (one thread per file); the blocks are copied by pack_block
RRZ_HD void pack_ends(uint8_t* dst, const BlockMeta* meta, int nb) {
  const int64_t total = stream_bytes(meta, nb);
  const uint32_t hd[4] = {PNGZ_MAGIC, (uint32_t)total, 0u, 0u};
  memcpy(dst, hd, 16);
  dst[PNGZ_HEADER] = 0x78;
  dst[PNGZ_HEADER + 1] = 0x01;
  const uint32_t ad = adler_of(meta, nb);
  uint8_t* q = dst + PNGZ_HEADER + total - 4;
  q[0] = (uint8_t)(ad >> 24);
  q[1] = (uint8_t)(ad >> 16);
  q[2] = (uint8_t)(ad >> 8);
  q[3] = (uint8_t)ad;
}
RRZ_HD int64_t block_offset(const BlockMeta* meta, int k) {
  int64_t o = PNGZ_HEADER + 2;
  for (int j = 0; j < k; j++) o += meta[j].bytes;
  return o;
}

}  // namespace rrz
