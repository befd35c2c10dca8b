// Test infrastructure: the PNG codec of the library (rain-rendering_amd/csrc/rr_png.cpp, compiled INTO this program) under
// AddressSanitizer + UndefinedBehaviorSanitizer.  tests/test_codec_sanitized.py builds and runs it.
//   readers   every PNG given on the command line, corrupted in four ways a few hundred times each (bit flips, truncation,
//             overwritten stretches, header bytes): any answer is fine, a memory error is not;
//   inflate   zlib streams of several kinds, intact and damaged: an intact stream must be vouched for with zlib's bytes,
//             nothing may be written more than 7 bytes past the output (the documented slack);
//   deflate   inputs of every length around the block and vector sizes in exact-size heap buffers: zlib must inflate the
//             stream back to the input.
#include "../../rain-rendering_amd/csrc/rr_png.cpp"

#include <random>

static int fuzz_readers(int argc, char** argv, int rounds) {
  std::mt19937 g(7);
  int answered = 0;
  for (int fi = 1; fi < argc; fi++) {
    std::vector<uint8_t> f;
    size_t fsz = 0;
    if (read_file(argv[fi], f, fsz)) return 1;
    Png p0;
    parse_chunks(f, fsz, p0, false);
    const int H = (int)p0.h, W = (int)p0.w;
    std::vector<uint8_t> out((size_t)H * W * 3);
    std::vector<uint16_t> out16((size_t)H * W);
    std::vector<uint8_t> rows3((size_t)H * (1 + 3 * (size_t)W)), rows2((size_t)H * (1 + 2 * (size_t)W));   // exact sizes: an overrun is seen
    const std::string tmp = std::string(argv[fi]) + ".fuzz";
    for (int it = 0; it < rounds; it++) {
      std::vector<uint8_t> m(f.begin(), f.begin() + (long)fsz);
      switch (it % 4) {
        case 0:
          for (int k = 0; k < 1 + (int)(g() % 8); k++) m[g() % m.size()] ^= (uint8_t)(1u << (g() % 8));
          break;
        case 1: m.resize(g() % m.size()); break;
        case 2: {
          const size_t a = g() % m.size(), n = g() % 2000;
          for (size_t k = a; k < m.size() && k < a + n; k++) m[k] = (uint8_t)g();
          break;
        }
        default: {
          const size_t a = 8 + g() % 40;
          if (a < m.size()) m[a] = (uint8_t)g();
        }
      }
      FILE* fh = fopen(tmp.c_str(), "wb");
      fwrite(m.data(), 1, m.size(), fh);
      fclose(fh);
      int32_t w, h, c, d;
      rr_png_read_bgr8(tmp.c_str(), out.data(), H, W);
      rr_png_read_gray16(tmp.c_str(), out16.data(), H, W);
      {                                   // the inflate-only reader (rr_io_read_frames_rows): the file as image and as depth map
        const char* path = tmp.c_str();
        int32_t st = 0;
        rr_io_read_frames_rows(1, &path, nullptr, H, W, rows3.data(), (int64_t)rows3.size(), nullptr, 0, 1, &st);
        if (st == RR_OK)
          for (int y = 0; y < H; y++)
            if (rows3[(size_t)y * (1 + 3 * (size_t)W)] > 4) return 2;          // only valid filter types may be handed to the device
        rr_io_read_frames_rows(1, &path, &path, H, W, rows3.data(), (int64_t)rows3.size(), rows2.data(), (int64_t)rows2.size(), 1, &st);
      }
      rr_png_info(tmp.c_str(), &w, &h, &c, &d);
      answered++;
    }
    remove(tmp.c_str());
  }
  printf("readers: %d damaged files answered\n", answered);
  return 0;
}

static int fuzz_inflate(int rounds) {
  std::mt19937 g(11);
  int vouched = 0, declined = 0, wrong = 0;
  for (int it = 0; it < rounds; it++) {
    const size_t n = 1 + g() % (it % 50 == 0 ? 300000 : 6000);
    std::vector<uint8_t> b(n);
    for (size_t i = 0; i < n; i++) {
      switch (it % 5) {
        case 0: b[i] = (uint8_t)g(); break;
        case 1: b[i] = (uint8_t)(g() % 4); break;
        case 2: b[i] = (uint8_t)((i / 37) & 255); break;
        case 3: b[i] = (uint8_t)(g() % 3 == 0 ? g() % 16 : 0); break;
        default: b[i] = (uint8_t)("the quick brown fox "[i % 20] + (g() % 50 == 0));
      }
    }
    uLongf cl = compressBound((uLong)n);
    std::vector<uint8_t> z(cl + 64, 0);
    compress2(z.data(), &cl, b.data(), (uLong)n, (int)(g() % 10));
    const bool damage = it % 2;
    size_t zl = cl;
    if (damage) {
      const int k = (int)(g() % 3);
      if (k == 0) z[g() % cl] ^= (uint8_t)(1u << (g() % 8));
      else if (k == 1) zl = g() % cl;
      else for (int q = 0; q < 5; q++) z[g() % cl] = (uint8_t)g();
    }
    std::vector<uint8_t> zin(zl + 64, 0), out(n + 16, 0xAA);
    memcpy(zin.data(), z.data(), zl);
    const bool ok = inflate_fast::inflate(zin.data(), zl, out.data(), n);
    if (ok) {
      vouched++;
      if (!damage && memcmp(out.data(), b.data(), n)) wrong++;
    } else {
      declined++;
      if (!damage) wrong++;
    }
    for (int q = 0; q < 8; q++)
      if (out[n + 8 + q] != 0xAA) wrong++;
  }
  printf("inflate: vouched %d declined %d wrong %d\n", vouched, declined, wrong);
  return wrong != 0;
}

static int fuzz_deflate(int rounds) {
  std::mt19937 g(5);
  int wrong = 0;
  ByteBuf z;
  std::vector<Run> runs;
  for (int it = 0; it < rounds; it++) {
    size_t n = it < 300 ? (size_t)it : 1 + g() % (it % 40 == 0 ? 400000 : 5000);
    if (it % 97 == 0) n = 131072 * (1 + g() % 3) + (g() % 5) - 2;
    uint8_t* exact = new uint8_t[n ? n : 1];
    uint8_t cur = 0;
    for (size_t i = 0; i < n; i++) {
      switch (it % 6) {
        case 0: exact[i] = (uint8_t)g(); break;
        case 1: if (g() % 20 == 0) cur = (uint8_t)g(); exact[i] = cur; break;
        case 2: exact[i] = (uint8_t)(g() % 2); break;
        case 3: exact[i] = 0; break;
        case 4: if (g() % 300 == 0) cur = (uint8_t)g(); exact[i] = cur; break;
        default: exact[i] = (uint8_t)((i % 4 == 3) ? 0 : g() % 7);
      }
    }
    fast_deflate(exact, n, z, runs);
    std::vector<uint8_t> back(n + 1);
    uLongf bl = (uLongf)(n + 1);
    if (uncompress(back.data(), &bl, z.data(), (uLong)z.len) != Z_OK || bl != n || memcmp(back.data(), exact, n)) wrong++;
    delete[] exact;
  }
  printf("deflate: wrong %d\n", wrong);
  return wrong != 0;
}

// rows that claim to be a stream the device made (RR_OPT_PNG_DEFLATE: 'RRZ1', length, two zero words, stream).  The writer takes
// the stream as the IDAT payload as it is, so it must refuse what a stale or half-written buffer looks like: a length that does
// not fit the buffer or cannot hold a block and the Adler-32, reserved words that are not zero, a payload that does not open
// with the zlib header the device writes (0x78 0x01).
static int fuzz_device_payload(const char* dir) {
  const int W = 7, H = 5;
  const size_t n = (size_t)H * (1 + 4 * W);
  const std::string path = std::string(dir) + "/payload.png";
  int bad = 0;
  for (int variant = 0; variant < 4; variant++)                  // 0: well-formed header, 1: no zlib header, 2 / 3: a reserved word set
    for (uint32_t L : {0u, 5u, 6u, 10u, 11u, (uint32_t)(n - 16), (uint32_t)(n - 15), (uint32_t)n, 0x7fffffffu, 0xffffffffu}) {
      uint8_t* rows = new uint8_t[n];
      memset(rows, 0, n);
      memcpy(rows, "RRZ1", 4);
      memcpy(rows + 4, &L, 4);
      if (variant != 1) { rows[16] = 0x78; rows[17] = 0x01; }
      if (variant >= 2) rows[8 + 4 * (variant - 2)] = 1;
      const int rc = rr_png_write_scanlines(path.c_str(), rows, W, H, 1, 3);
      const bool fits = variant == 0 && L >= 11 && (size_t)L + 16 <= n;
      if ((rc == RR_OK) != fits) bad++;
      delete[] rows;
    }
  remove(path.c_str());
  printf("device payloads: wrong %d\n", bad);
  return bad != 0;
}

int main(int argc, char** argv) {
  int rc = fuzz_readers(argc, argv, 160);
  if (argc > 1) {
    std::string dir(argv[1]);
    dir = dir.substr(0, dir.find_last_of('/'));
    rc |= fuzz_device_payload(dir.c_str());
  }
  rc |= fuzz_inflate(1200);
  rc |= fuzz_deflate(900);
  return rc;
}
