// rr_deflate.h -- entropy coding of the two output PNG files on the device (SURVEY 8f "next" #3, the writer's side;
// the reference saves both with plt.imsave, generator.py:466-467).
// k_png_image / k_png_mask leave the filtered scanlines of a file in HBM: H rows of 1 + 4 W bytes.  This header turns them
// into the zlib stream of the file's IDAT chunk, so that the host only frames chunks and computes one CRC:
//   * the scanlines are cut into blocks of 32 KB, one workgroup of 512 threads each; a wave owns an eighth of the block and
//     walks it in chunks of 64 bytes, one lane per byte
//   * tokens of a chunk: a byte that differs from its predecessor (or opens the chunk) starts a sequence of equal bytes; a
//     sequence of four or more is its first byte as a literal and one run (length = the rest, distance 1), a shorter one is
//     literals -- what the host encoder's rule (rr_png.cpp fast_deflate, like zlib's Z_RLE) gives inside 64 bytes.  Which
//     token a lane emits follows from the ballot of the sequence starts by bit operations: no lane waits for another
//   * one dynamic-Huffman deflate block per 32 KB: histogram (LDS atomics), symbols ranked by (count, symbol) in parallel,
//     the two-queue tree construction by one thread (the only serial step: <= 286 iterations), depths, the count-based fix-up
//     to 15 bits, canonical codes; the code lengths themselves go out with a fixed 4-bit code (no repeat symbols: 144 bytes of
//     header per block, and nothing serial to build)
//   * bit positions: a wave's first bit from the per-wave histograms (sum of count x code length), inside the wave a running
//     base plus the wave's prefix sum of the chunk's code lengths; every lane ORs its code into the block's buffer in LDS
//   * a block that is not the file's last ends with an empty stored block (the "sync flush" of zlib / pigz), so every block
//     starts on a byte boundary and the blocks of a file are simply laid behind each other (k_pngz_pack); a block whose
//     dynamic form would be larger than its bytes is a stored block
//   * Adler-32: per block (sum of bytes, position-weighted sum), combined by the pack kernel.
// Result in the file's scanline buffer (capacity H * (1 + 4 W), unchanged): a 16-byte header {PNGZ_MAGIC, length of the zlib
// stream, 0, 0} followed by the stream -- if that fits (always, unless the image is incompressible: the first byte of a
// scanline buffer is a PNG filter type 0..4, never the magic's 'R').  Otherwise the scanlines are left as they are and the
// host compresses them as before.  rr_png_write_scanlines / rr_io_write_frames (rr_png.cpp) take either.
// The functions are `__host__ __device__`: the phases in "thread role" form (tid = 0..511, barriers between them) like FogTile
// in rr_prepass.h, the two token passes per lane (lane_token / token_code from a chunk's ballot; the ballot, the prefix sum and
// the running base are three lines of wave intrinsics in k_pngz_blocks and three loops in tests/hostemu): the same code runs
// on the CPU and zlib inflates what it produces (tests/test_deflate_hostemu.py).
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

constexpr int NT = 512, NW = NT / 64, BLOCK = 32768, WAVE_BYTES = BLOCK / NW, CHUNK = 64;
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
  uint32_t freq4[NW][NSYM / 2];      // per wave: what the wave's eighth of the block holds (its bit count follows from it);
                                     // two 16-bit counters per word (a wave sees 4096 bytes)
  uint32_t freq[NSYM];
  uint32_t w[2 * NSYM];              // tree: weights of the leaves (in sorted order) and of the inner nodes
  uint32_t cnt[MAXL + 1];            // symbols per code length
  uint32_t ad1[NW], ad2[NW];         // Adler partials of the waves' eighths
  uint32_t wave_bits[NW];
  uint16_t sorted[NSYM];             // used symbols by rising (count, symbol)
  uint16_t parent[2 * NSYM];
  uint32_t cl32[NSYM];               // canonical code, bit-reversed (deflate packs codes most significant bit first) | length << 16
  uint8_t len[NSYM];
  int32_t n, last, m, nlit, stored;
  uint32_t hdr_bits, data_bits, bytes;
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
#if defined(__HIP_DEVICE_COMPILE__)
  const int lg = 31 - __builtin_clz((unsigned)x);       // floor(log2 x), 3..7
#else
  int lg = 0;
  while ((x >> (lg + 1)) != 0) lg++;
#endif
  ebits = lg - 2;
  sym = 265 + 4 * (ebits - 1) + ((x >> ebits) - 4);
  eval = x & ((1 << ebits) - 1);
}

// The token lane `lane` of a chunk emits.  nv: bytes in the chunk (lanes >= nv hold nothing); start: bit i set where byte i
// opens a sequence of equal bytes (bit 0 always); b: the lane's byte.  kind 0: nothing (the byte lies inside a run), 1: the
// literal b, 2: a run of v bytes at distance 1.
struct Tok {
  int kind, v;
};
RRZ_HD int ctz64(uint64_t x) {
  int n = 0;
  while (!(x & 1ull)) {
    x >>= 1;
    n++;
  }
  return n;
}
RRZ_HD int fls64(uint64_t x) {                           // index of the highest set bit (x != 0)
  int n = 0;
  while (x >>= 1) n++;
  return n;
}
RRZ_HD Tok lane_token(int lane, int nv, uint64_t start, int b) {
  if (lane >= nv) return Tok{0, 0};
  const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
  const int j0 = 63 - __builtin_clzll(start & upto);
  const uint64_t above = lane == 63 ? 0ull : (start >> (lane + 1));
  const int next = above ? lane + 1 + __builtin_ctzll(above) : 64;
#else
  const int j0 = fls64(start & upto);
  const uint64_t above = lane == 63 ? 0ull : (start >> (lane + 1));
  const int next = above ? lane + 1 + ctz64(above) : 64;
#endif
  const int L = (next < nv ? next : nv) - j0;            // length of the lane's sequence
  if (lane == j0 || L < 4) return Tok{1, b};
  return lane == j0 + 1 ? Tok{2, L - 1} : Tok{0, 0};
}
// a token under the block's code: its bits (at most 15 + 5 + 1, the distance bit of a run -- 0 -- on top) and their number
RRZ_HD uint32_t token_code(const BlockState& S, Tok t, uint32_t& nbits) {
  if (t.kind == 1) {
    const uint32_t e = S.cl32[t.v];
    nbits = e >> 16;
    return e & 0xffffu;
  }
  if (t.kind != 2) {
    nbits = 0;
    return 0;
  }
  int sym, eb, ev;
  length_code(t.v, sym, eb, ev);
  const uint32_t e = S.cl32[sym];
  nbits = (e >> 16) + (uint32_t)eb + 1u;
  return (e & 0xffffu) | ((uint32_t)ev << (e >> 16));
}
RRZ_HD int token_symbol(Tok t) {
  if (t.kind == 1) return t.v;
  int sym, eb, ev;
  length_code(t.v, sym, eb, ev);
  return sym;
}
// a code of n bits at bit position pos of the block's buffer (other lanes write the bits around it)
RRZ_HD void or_bits(uint32_t* out, uint32_t pos, uint32_t bits, uint32_t n) {
  if (!n) return;
  const uint32_t w = pos >> 5, sh = pos & 31u;
  RRZ_OR(out + w, bits << sh);
  if (sh + n > 32u) RRZ_OR(out + w + 1, bits >> (32u - sh));
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
// P0: the block's bytes are in S.in (the caller copies them); clear the rest
RRZ_HD void p0_init(BlockState& S, int tid, int n, int last) {
  for (int k = tid; k < OUT_WORDS; k += NT) S.out[k] = 0;
  for (int k = tid; k < NSYM; k += NT) {
    if (k < NSYM / 2)
      for (int w = 0; w < NW; w++) S.freq4[w][k] = 0;
    S.len[k] = 0;
    S.cl32[k] = 0;
  }
  if (tid <= MAXL) S.cnt[tid] = 0;
  if (tid < NW) S.wave_bits[tid] = 0;
  if (tid == 0) {
    S.n = n;
    S.last = last;
    S.m = 0;
    S.nlit = 257;
    S.stored = 0;
  }
}
// P1 (wave form, see k_pngz_blocks / emu_pngz): one chunk of a wave's quarter -- the lane's token into the wave's histogram, its
// byte into the lane's Adler partials.
RRZ_HD void p1_lane(BlockState& S, int wave, Tok t, int b, int left, bool valid, uint32_t& s1, uint32_t& s2) {
  if (t.kind) {
    const int sym = token_symbol(t);
    RRZ_ADD(&S.freq4[wave][sym >> 1], 1u << (16 * (sym & 1)));
  }
  if (valid) {                                           // left: the block's length minus the byte's position in it
    s1 += (uint32_t)b;
    s2 += (uint32_t)left * (uint32_t)b;                  // per lane: 64 chunks x 32768 x 255 < 2^32            // per lane: 128 chunks x 32768 x 255 < 2^32
  }
}
RRZ_HD uint32_t wave_count(const BlockState& S, int w, int sym) { return (S.freq4[w][sym >> 1] >> (16 * (sym & 1))) & 0xffffu; }
// P1b: the block's histogram
RRZ_HD void p1_sum(BlockState& S, int tid) {
  for (int k = tid; k < NSYM; k += NT) {
    uint32_t f = k == 256 ? 1u : 0u;                     // end of block
    for (int w = 0; w < NW; w++) f += wave_count(S, w, k);
    S.freq[k] = f;
  }
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
  // (the heads of the two queues stay in registers: one load per node taken, not two per comparison)
  int leaf = 0, inner = m, made = m;
  uint32_t wl = S.w[0], wi = 0;
  while (made < 2 * m - 1) {
    uint32_t sum = 0;
    for (int k = 0; k < 2; k++) {
      if (leaf < m && (inner >= made || wl <= wi)) {
        sum += wl;
        S.parent[leaf++] = (uint16_t)made;
        if (leaf < m) wl = S.w[leaf];
      } else {
        sum += wi;
        S.parent[inner++] = (uint16_t)made;
        if (inner < made) wi = S.w[inner];
      }
    }
    S.w[made] = sum;
    if (inner == made) wi = sum;                          // the new node is the head of the inner queue
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
    S.cl32[s] = bit_reverse(c + idx, L) | ((uint32_t)L << 16);
  }
}
// P8: the block header.  Fixed part by thread 0: BFINAL, BTYPE 10, HLIT, HDIST 0 (one distance code), HCLEN 15 and the 19
// code-length-code lengths in the order of RFC 1951 3.2.7: symbols 16, 17, 18 unused, every length 0..15 a 4-bit code (the
// canonical code of sixteen 4-bit symbols: the length value itself, bits reversed).  Then the nlit literal/length code lengths
// and the distance code's length 1, four bits each: entry k at bit 74 + 4 k, two entries per thread, no scan needed.
constexpr uint32_t HDR_FIXED = 17 + 3 * 19;
RRZ_HD void p8_header(BlockState& S, int tid) {
  if (tid == 0) {
    BitOut bo(S.out, 0);
    bo.put(S.last ? 1u : 0u, 1);
    bo.put(2u, 2);
    bo.put((uint32_t)(S.nlit - 257), 5);
    bo.put(0u, 5);
    bo.put(15u, 4);
    for (int k = 0; k < 19; k++) bo.put(k < 3 ? 0u : 4u, 3);
    bo.finish();
    S.hdr_bits = HDR_FIXED + 4u * (uint32_t)(S.nlit + 1);
  }
  const int k0 = 2 * tid;
  if (k0 <= S.nlit) {
    BitOut bo(S.out, HDR_FIXED + 4u * (uint32_t)k0);
    bo.put(bit_reverse(k0 < S.nlit ? S.len[k0] : 1u, 4), 4);
    if (k0 + 1 <= S.nlit) bo.put(bit_reverse(k0 + 1 < S.nlit ? S.len[k0 + 1] : 1u, 4), 4);
    bo.finish();
  }
}
// P9: a wave's bit count from its histogram: sum of count x (code length + extra bits + the distance bit of a run)
RRZ_HD void p9_wave_bits(BlockState& S, int tid) {
  for (int k = tid; k < NSYM; k += NT) {
    if (!S.len[k] || k == 256) continue;
    uint32_t cost = S.len[k];
    if (k > 256) {
      const int j = k - 257;                              // extra bits of length symbol 257 + j (RFC 1951 3.2.5)
      cost += (j < 8 ? 0u : (uint32_t)((j - 4) >> 2)) + 1u;
    }
    for (int w = 0; w < NW; w++)
      if (wave_count(S, w, k)) RRZ_ADD(&S.wave_bits[w], wave_count(S, w, k) * cost);
  }
}
// P10 (thread 0): size of the dynamic form; a block that does not shrink is stored
RRZ_HD void p10_decide(BlockState& S, int tid) {
  if (tid != 0) return;
  S.data_bits = 0;
  for (int w = 0; w < NW; w++) S.data_bits += S.wave_bits[w];
  const uint64_t bits = (uint64_t)S.hdr_bits + S.data_bits + S.len[256];
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
// first data bit of a wave's quarter
RRZ_HD uint32_t wave_base(const BlockState& S, int wave) {
  uint32_t b = S.hdr_bits;
  for (int w = 0; w < wave; w++) b += S.wave_bits[w];
  return b;
}
// P12 (thread 0 of a dynamic block): end of block, then -- unless the file ends here -- the empty stored block; (of a stored
// block): its five header bytes.  The tokens themselves go out in wave form (or_bits at base + prefix sums).
RRZ_HD void p12_ends(BlockState& S, int tid) {
  if (tid != 0) return;
  if (S.stored) {
    uint8_t* ob = reinterpret_cast<uint8_t*>(S.out);
    ob[0] = S.last ? 1 : 0;                               // BFINAL, BTYPE 00, padding
    ob[1] = (uint8_t)(S.n & 255);
    ob[2] = (uint8_t)(S.n >> 8);
    ob[3] = (uint8_t)(~S.n & 255);
    ob[4] = (uint8_t)((~S.n >> 8) & 255);
    return;
  }
  const uint32_t endpos = S.hdr_bits + S.data_bits;
  or_bits(S.out, endpos, S.cl32[256] & 0xffffu, S.len[256]);
  if (!S.last) {
    const uint32_t after = endpos + S.len[256] + 3;       // three zero bits: BFINAL 0, BTYPE 00
    const uint32_t at = (after + 7) >> 3;                 // LEN 0000, NLEN ffff on the next byte boundary
    RRZ_OR(&S.out[(at + 2) >> 2], 0xffu << (8 * ((at + 2) & 3)));
    RRZ_OR(&S.out[(at + 3) >> 2], 0xffu << (8 * ((at + 3) & 3)));
  }
}
// P12b (stored blocks): the raw bytes behind the five header bytes
RRZ_HD void p12b_stored_bytes(BlockState& S, int tid) {
  if (!S.stored) return;
  uint8_t* ob = reinterpret_cast<uint8_t*>(S.out) + 5;
  for (int i = tid; i < S.n; i += NT) ob[i] = S.in[i];
}
// P13 (thread 0): the block's record
RRZ_HD void p13_meta(BlockState& S, int tid, BlockMeta* meta) {
  if (tid != 0) return;
  uint64_t s1 = 0, s2 = 0;
  for (int k = 0; k < NW; k++) {
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
