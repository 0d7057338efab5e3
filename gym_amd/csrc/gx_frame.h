// Shared pieces of the 210x160x3 frame rasterisers (Pong, Breakout, SpaceInvaders).
//
// A frame row is 160 pixels = 40 groups of four pixels; a lane produces one
// group = 12 bytes and writes it with one store, so a wave writes 768
// contiguous bytes and a workgroup walks one env's 100 800-byte frame front
// to back.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace gx {

constexpr int kScreenW = 160, kScreenH = 210, kFrameBytes = kScreenW * kScreenH * 3;
constexpr int kGroupsPerRow = kScreenW / 4, kGroupsPerFrame = kScreenH * kGroupsPerRow;

struct U3 {
  uint32_t x, y, z;
};

// four 24-bit colours (r | g << 8 | b << 16) -> twelve bytes r0 g0 b0 r1 ...
__device__ __forceinline__ U3 pack_rgb4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  U3 v;
  v.x = c0 | (c1 << 24);
  v.y = (c1 >> 8) | (c2 << 16);
  v.z = (c2 >> 16) | (c3 << 8);
  return v;
}

// bit i set when pixel 4g+i lies in [x0, x0+w)
__device__ __forceinline__ uint32_t span_mask(int x0, int w, int g) {
  int lo = x0 - 4 * g, hi = lo + w;
  lo = lo < 0 ? 0 : lo;
  hi = hi > 4 ? 4 : hi;
  return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}

// Row `row` (0..4) of the 3x5 digit glyphs, bit dx = column dx.  The table of
// Envs.Pong.cs:228-237 and Envs.Breakout.cs:321-330 (identical), 3 bits per row.
__device__ __forceinline__ uint32_t glyph_row(uint32_t digit, int row) {
  uint32_t g;
  switch (digit) {
    case 0: g = 7 | 5 << 3 | 5 << 6 | 5 << 9 | 7 << 12; break;
    case 1: g = 2 | 3 << 3 | 2 << 6 | 2 << 9 | 7 << 12; break;
    case 2: g = 7 | 4 << 3 | 7 << 6 | 1 << 9 | 7 << 12; break;
    case 3: g = 7 | 4 << 3 | 7 << 6 | 4 << 9 | 7 << 12; break;
    case 4: g = 5 | 5 << 3 | 7 << 6 | 4 << 9 | 4 << 12; break;
    case 5: g = 7 | 1 << 3 | 7 << 6 | 4 << 9 | 7 << 12; break;
    case 6: g = 7 | 1 << 3 | 7 << 6 | 5 << 9 | 7 << 12; break;
    case 7: g = 7 | 4 << 3 | 4 << 6 | 4 << 9 | 4 << 12; break;
    case 8: g = 7 | 5 << 3 | 7 << 6 | 5 << 9 | 7 << 12; break;
    default: g = 7 | 5 << 3 | 7 << 6 | 4 << 9 | 7 << 12; break;
  }
  return (g >> (3 * row)) & 7u;
}

// ORs the glyph rows `row` of the decimal string of `score` into a 160-bit
// row mask (5 words in LDS): character i of the string sits at x0 + pitch*i,
// a '-' occupies a cell and draws nothing (the glyph switch's default arm).
__device__ __forceinline__ void score_row_mask(uint32_t* mask, int score, int row, int x0, int pitch, bool centred) {
  uint32_t mag = score < 0 ? 0u - (uint32_t)score : (uint32_t)score;
  int ndig = 1;
  for (uint32_t t = mag; t >= 10; t /= 10) ndig++;
  const int len = ndig + (score < 0 ? 1 : 0);
  const int startX = centred ? x0 - (len * 6) / 2 : x0;  // Envs.Pong.cs:214
  for (int i = len - 1; i >= len - ndig; i--) {
    const uint32_t g = glyph_row(mag % 10, row);
    mag /= 10;
#pragma unroll
    for (int dx = 0; dx < 3; dx++) {
      const int px = startX + pitch * i + dx;
      if (((g >> dx) & 1u) && px >= 0 && px < kScreenW) atomicOr(&mask[px >> 5], 1u << (px & 31));
    }
  }
}

}  // namespace gx
