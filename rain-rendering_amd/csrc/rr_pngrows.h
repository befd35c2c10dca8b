// rr_pngrows.h -- the INPUT files' scanline filters reversed on the device (SURVEY 8f "next" #3, the reader's side;
// the reference reads both files of a frame with cv2.imread, generator.py:352,360).
// The host inflates a file's IDAT stream (sequential by nature) and hands over the filtered scanlines as they come out:
// H rows of 1 + BPP * W bytes, a filter-type byte in front of every row (PNG specification, section 9: None, Sub, Up,
// Average, Paeth).  Reversing the filters is a recurrence over (left, up, upper-left) neighbours -- sequential along a row
// for Sub / Average / Paeth, from row to row for Up / Average / Paeth -- that the host spent 40 % of its decode time on.
// k_png_unfilter (rainhip.hip) runs it as a wavefront: a wave takes 64 consecutive rows, lane l row r0 + l, skewed by one
// pixel per lane, so that at step s lane l is at pixel s - l and its upper neighbours are what lane l - 1 produced one and
// two steps earlier (a wave shuffle, a register).  The per-byte rule below is shared with tests/hostemu.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RRR_HD __host__ __device__ inline
#else
#define RRR_HD inline
#endif

namespace rrrows {

RRR_HD int png_paeth(int a, int b, int c) {             // PNG specification 9.4: a = left, b = up, c = upper left
  const int p = a + b - c;
  const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
// the byte a filtered byte f stands for, given the bytes already reconstructed around it (0 outside the image)
RRR_HD int png_unfilter_byte(int ft, int f, int left, int up, int upleft) {
  int pred = 0;
  if (ft == 1) pred = left;
  else if (ft == 2) pred = up;
  else if (ft == 3) pred = (left + up) >> 1;
  else if (ft == 4) pred = png_paeth(left, up, upleft);
  return (f + pred) & 255;
}

}  // namespace rrrows
