// Test infrastructure: the host build of the device's byte-level code (tests/hostemu/hostemu.cpp: csrc/rr_deflate.h -- the
// entropy coder of the output PNG files -- and csrc/rr_pngrows.h -- the scanline filters of the input files) under
// AddressSanitizer + UndefinedBehaviorSanitizer, on inputs of every length around the chunk / wave / block sizes in exact-size
// heap buffers.  zlib must inflate every stream back to its input.  tests/test_codec_sanitized.py builds and runs it.
#include "../hostemu/hostemu.cpp"

#include <zlib.h>

#include <cstdio>
#include <random>

int main() {
  std::mt19937 g(3);
  int wrong = 0, coded = 0, kept = 0;
  for (int it = 0; it < 420; it++) {
    size_t n = it < 200 ? (size_t)it + 1 : 1 + g() % (it % 30 == 0 ? 140000 : 9000);
    if (it % 41 == 0) n = 32768 * (1 + g() % 3) + (g() % 5) - 2;
    if (it % 43 == 0) n = 4096 * (1 + g() % 9) + (g() % 3) - 1;
    uint8_t* in = new uint8_t[n];
    uint8_t* out = new uint8_t[n];
    uint8_t cur = 0;
    for (size_t i = 0; i < n; i++) {
      switch (it % 6) {
        case 0: in[i] = (uint8_t)g(); break;
        case 1: if (g() % 20 == 0) cur = (uint8_t)g(); in[i] = cur; break;
        case 2: in[i] = (uint8_t)(g() % 2); break;
        case 3: in[i] = 0; break;
        case 4: if (g() % 300 == 0) cur = (uint8_t)g(); in[i] = cur; break;
        default: in[i] = (uint8_t)((i % 4 == 3) ? 0 : g() % 7);
      }
    }
    const int64_t total = emu_pngz(in, (int64_t)n, out);
    if (total == 0) {
      kept++;
      if (memcmp(in, out, n)) wrong++;
    } else {
      coded++;
      uint32_t L;
      memcpy(&L, out + 4, 4);
      std::vector<uint8_t> back(n + 1);
      uLongf bl = (uLongf)(n + 1);
      if (memcmp(out, "RRZ1", 4) || (int64_t)L != total || 16 + (size_t)total > n ||
          uncompress(back.data(), &bl, out + 16, (uLong)total) != Z_OK || bl != n || memcmp(back.data(), in, n))
        wrong++;
    }
    delete[] in;
    delete[] out;
  }
  printf("device deflate: coded %d kept %d wrong %d\n", coded, kept, wrong);
  // scanline filters: random filter types and bytes, every width around a lane / wave count
  int bad = 0;
  for (int it = 0; it < 120; it++) {
    const int H = 1 + (int)(g() % 140), W = 1 + (int)(g() % 90), bpp = it % 2 ? 3 : 2;
    const size_t rb = 1 + (size_t)bpp * W;
    uint8_t* rows = new uint8_t[rb * H];
    for (size_t i = 0; i < rb * H; i++) rows[i] = (uint8_t)g();
    for (int y = 0; y < H; y++) rows[rb * y] = (uint8_t)(g() % 5);
    uint8_t* out = new uint8_t[(size_t)H * W * (bpp == 3 ? 3 : 2)];
    if (emu_png_unfilter(rows, H, W, bpp, out) != 0) bad++;
    rows[rb * (g() % H)] = 5 + (uint8_t)(g() % 200);           // a filter type that does not exist
    if (emu_png_unfilter(rows, H, W, bpp, out) != -1) bad++;
    delete[] rows;
    delete[] out;
  }
  printf("unfilter: wrong %d\n", bad);
  return (wrong != 0) | (bad != 0);
}
