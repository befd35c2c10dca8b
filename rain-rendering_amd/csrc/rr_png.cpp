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
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rainhip.h"

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Png {
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
};

int read_file(const char* path, std::vector<uint8_t>& buf) {
  FILE* fh = fopen(path, "rb");
  if (!fh) return RR_E_ARG;
  fseek(fh, 0, SEEK_END);
  long sz = ftell(fh);
  fseek(fh, 0, SEEK_SET);
  if (sz < 0) { fclose(fh); return RR_E_ARG; }
  buf.resize((size_t)sz);
  size_t got = fread(buf.data(), 1, (size_t)sz, fh);
  fclose(fh);
  return got == (size_t)sz ? RR_OK : RR_E_PARSE;
}

int parse_chunks(const std::vector<uint8_t>& f, Png& p, bool want_data) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
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

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0; }

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// inflate + reverse the scanline filters: `img` = h rows of `stride` bytes
int decode(const Png& p, std::vector<uint8_t>& img, size_t& stride) {
  const int ch = channels_of(p.ctype);
  if (!ch || p.interlace || (p.depth != 8 && p.depth != 16) || (p.ctype == 3 && p.depth != 8)) return RR_E_UNSUPPORTED;
  const size_t bpp = (size_t)ch * p.depth / 8;
  stride = (size_t)p.w * bpp;
  std::vector<uint8_t> raw((stride + 1) * p.h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, p.idat.data(), (uLong)p.idat.size()) != Z_OK || out_len != raw.size()) return RR_E_PARSE;
  img.resize(stride * p.h);
  for (uint32_t y = 0; y < p.h; y++) {
    const uint8_t* src = &raw[(stride + 1) * y];
    const int ft = src[0];
    src++;
    uint8_t* cur = &img[stride * y];
    const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
    switch (ft) {
      case 0: memcpy(cur, src, stride); break;
      case 1:
        for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(src[i] + (i >= bpp ? cur[i - bpp] : 0));
        break;
      case 2:
        for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(src[i] + (up ? up[i] : 0));
        break;
      case 3:
        for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(src[i] + (((i >= bpp ? cur[i - bpp] : 0) + (up ? up[i] : 0)) >> 1));
        break;
      case 4:
        for (size_t i = 0; i < stride; i++)
          cur[i] = (uint8_t)(src[i] + paeth(i >= bpp ? cur[i - bpp] : 0, up ? up[i] : 0, (up && i >= bpp) ? up[i - bpp] : 0));
        break;
      default: return RR_E_PARSE;
    }
  }
  return RR_OK;
}

}  // namespace

extern "C" int rr_png_info(const char* path, int32_t* w, int32_t* h, int32_t* channels, int32_t* bit_depth) {
  if (!path || !w || !h || !channels || !bit_depth) return RR_E_ARG;
  std::vector<uint8_t> f;
  int rc = read_file(path, f);
  if (rc) return rc;
  Png p;
  if ((rc = parse_chunks(f, p, false))) return rc;
  *w = (int32_t)p.w;
  *h = (int32_t)p.h;
  *channels = p.ctype == 3 ? 3 : channels_of(p.ctype);
  *bit_depth = p.depth;
  if (p.interlace || !channels_of(p.ctype)) return RR_E_UNSUPPORTED;
  return RR_OK;
}

// cv2.imread(path): 8 bits per channel, three channels, B G R
extern "C" int rr_png_read_bgr8(const char* path, uint8_t* out, int32_t H, int32_t W) {
  if (!path || !out) return RR_E_ARG;
  std::vector<uint8_t> f;
  int rc = read_file(path, f);
  if (rc) return rc;
  Png p;
  if ((rc = parse_chunks(f, p, true))) return rc;
  if ((int32_t)p.w != W || (int32_t)p.h != H) return RR_E_ARG;
  if (p.depth != 8) return RR_E_UNSUPPORTED;          // cv2 scales 16-bit colour down: left to the general decoder
  std::vector<uint8_t> img;
  size_t stride = 0;
  if ((rc = decode(p, img, stride))) return rc;
  const int ch = channels_of(p.ctype);
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
extern "C" int rr_png_read_gray16(const char* path, uint16_t* out, int32_t H, int32_t W) {
  if (!path || !out) return RR_E_ARG;
  std::vector<uint8_t> f;
  int rc = read_file(path, f);
  if (rc) return rc;
  Png p;
  if ((rc = parse_chunks(f, p, true))) return rc;
  if ((int32_t)p.w != W || (int32_t)p.h != H) return RR_E_ARG;
  if (p.ctype != 0 || p.depth != 16) return RR_E_UNSUPPORTED;
  std::vector<uint8_t> img;
  size_t stride = 0;
  if ((rc = decode(p, img, stride))) return rc;
  for (size_t i = 0; i < (size_t)H * W; i++) out[i] = (uint16_t)((img[2 * i] << 8) | img[2 * i + 1]);   // PNG is big-endian
  return RR_OK;
}

// RGBA PNG from its filtered scanlines: H rows of 1 + 4*W bytes (filter byte + filtered pixels).
// strategy: 0 zlib's default (LZ77 + Huffman), 1 Z_RLE (run lengths + Huffman: on filtered image data as small as
// level 1 of the default strategy or smaller, at half the time), 2 Z_HUFFMAN_ONLY.
extern "C" int rr_png_write_scanlines(const char* path, const uint8_t* rows, int32_t W, int32_t H, int32_t level, int32_t strategy) {
  if (!path || !rows || W <= 0 || H <= 0 || level < 0 || level > 9 || strategy < 0 || strategy > 2) return RR_E_ARG;
  const uLong n = (uLong)H * (1 + 4 * (uLong)W);
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, level, Z_DEFLATED, 15, 9, strategy == 1 ? Z_RLE : strategy == 2 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY) != Z_OK)
    return RR_E_PARSE;
  std::vector<uint8_t> z(deflateBound(&zs, n));
  zs.next_in = const_cast<Bytef*>(rows);
  zs.avail_in = (uInt)n;
  zs.next_out = z.data();
  zs.avail_out = (uInt)z.size();
  const int zrc = deflate(&zs, Z_FINISH);
  const uLongf clen = zs.total_out;
  deflateEnd(&zs);
  if (zrc != Z_STREAM_END) return RR_E_PARSE;
  FILE* fh = fopen(path, "wb");
  if (!fh) return RR_E_ARG;
  auto put32 = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
  auto chunk = [&](const char* tag, const uint8_t* data, uint32_t len) {
    uint8_t hd[8], tl[4];
    put32(hd, len);
    memcpy(hd + 4, tag, 4);
    uLong c = crc32(0L, (const Bytef*)tag, 4);
    if (len) c = crc32(c, data, len);
    put32(tl, (uint32_t)c);
    return fwrite(hd, 1, 8, fh) == 8 && (len == 0 || fwrite(data, 1, len, fh) == len) && fwrite(tl, 1, 4, fh) == 4;
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  uint8_t ihdr[13];
  put32(ihdr, (uint32_t)W);
  put32(ihdr + 4, (uint32_t)H);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  bool ok = fwrite(sig, 1, 8, fh) == 8 && chunk("IHDR", ihdr, 13) && chunk("IDAT", z.data(), (uint32_t)clen) && chunk("IEND", nullptr, 0);
  ok = (fclose(fh) == 0) && ok;
  return ok ? RR_OK : RR_E_ARG;
}
