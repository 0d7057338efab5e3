// Shared pieces of the 210x160x3 frame rasterisers (Pong, Breakout, SpaceInvaders).
//
// A frame row is 160 pixels = 40 groups of four pixels; a lane produces one
// group = 12 bytes and writes it with one store, so a wave writes 768
// contiguous bytes and a workgroup walks one env's 100 800-byte frame front
// to back.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>

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

// ---- 16-byte chunks -------------------------------------------------------
// A frame is 6300 chunks of 16 bytes, a row 30 of them; chunk j of a row holds row bytes [16 j, 16 j + 16),
// i.e. bytes of the six pixels p0 .. p0 + 5 with p0 = floor(16 j / 3), starting `phase` = 16 j mod 3 = j mod 3
// bytes into pixel p0.
struct Bytes16 {
  uint32_t x, y, z, w;
};
// the sixteen bytes of chunk j of a row in which exactly the pixels [x0, x0 + w) are white
constexpr Bytes16 chunk_bytes(int x0, int w, int j) {
  uint32_t d[4] = {0, 0, 0, 0};
  for (int i = 0; i < 16; i++) {
    const int pixel = (16 * j + i) / 3;
    if (pixel >= x0 && pixel < x0 + w) d[i >> 2] |= 0xFFu << (8 * (i & 3));
  }
  return Bytes16{d[0], d[1], d[2], d[3]};
}

__device__ __forceinline__ uint32_t span_bits6(int x0, int w, int p0) {
  int lo = x0 - p0, hi = lo + w;
  lo = lo < 0 ? 0 : lo;
  hi = hi > 6 ? 6 : hi;
  return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}

// six pixel bits -> the sixteen bytes that start `phase` bytes into the first pixel (0xFF per byte of a set pixel)
__device__ __forceinline__ uint4 expand_white6(uint32_t m, int phase) {
  // all-ones where the pixel's bit is set
  const uint32_t b0 = 0u - (m & 1u), b1 = 0u - ((m >> 1) & 1u), b2 = 0u - ((m >> 2) & 1u);
  const uint32_t b3 = 0u - ((m >> 3) & 1u), b4 = 0u - ((m >> 4) & 1u), b5 = 0u - ((m >> 5) & 1u);
  // pixel q covers bits [24 q, 24 q + 24) of a 144-bit string
  const uint32_t w0 = (b0 & 0x00FFFFFFu) | (b1 & 0xFF000000u);
  const uint32_t w1 = (b1 & 0x0000FFFFu) | (b2 & 0xFFFF0000u);
  const uint32_t w2 = (b2 & 0x000000FFu) | (b3 & 0xFFFFFF00u);
  const uint32_t w3 = (b4 & 0x00FFFFFFu) | (b5 & 0xFF000000u);
  const uint32_t w4 = b5 & 0x0000FFFFu;
  const uint32_t sh = 8u * (uint32_t)phase;  // 0, 8 or 16
  return make_uint4(__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh),
                    __builtin_amdgcn_alignbit(w3, w2, sh), __builtin_amdgcn_alignbit(w4, w3, sh));
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


// Launch geometry of the three rasters.  A frame is written by `split` workgroups (its parts), each a run of
// whole steps of THREADS x 16 bytes laid on the frame buffer's own THREADS x 16-byte grid (a frame starts 16-byte
// but not cache-line aligned, 100 800 = 787.5 x 128: the first and the last step of a frame are partly masked, and
// no cache line is written by two workgroups).
//
// The hardware deals the workgroups of a grid to the eight XCDs in turn, in the order x fastest, then y, then z:
// with grid = (8, split, frames per XCD) blockIdx.x IS the XCD a workgroup runs on, and XCD x writes the x-th
// eighth of the batch's frames in order -- one forward-moving stretch of the frame buffer per XCD (per L2)
// rather than every eighth piece of one stretch shared with the other seven.  Measured, the write rate then varies
// smoothly with the launch shape; with one linear grid in launch order it jumped by 30 % between neighbouring
// shapes.  The three indices also spare every wave the divisions by `split` a linear index needs (about sixty
// scalar instructions: short workgroups are bound by the scalar unit, which all waves of a CU share).
struct RasterLaunch {
  uint64_t env_first;    // this launch covers envs [env_first, env_first + env_count)
  uint32_t env_count, envs_per_xcd;
  int per_part[2];       // steps per part of a frame of kSteps / kSteps + 1 steps
};
template <int THREADS>
__device__ __forceinline__ bool raster_part(const RasterLaunch& g, uint64_t& env, int& c_first, int& c_end) {
  constexpr int kChunks = kFrameBytes / 16, kSteps = (kChunks + THREADS - 1) / THREADS;
  const uint32_t local = blockIdx.x * g.envs_per_xcd + blockIdx.z;
  if (local >= g.env_count) return false;
  env = g.env_first + local;
  const int head = (int)((env * (uint64_t)kChunks) & (THREADS - 1));
  const int steps = (head + kChunks + THREADS - 1) / THREADS;  // kSteps or one more
  const int per_part = steps == kSteps ? g.per_part[0] : g.per_part[1];
  const int part = (int)blockIdx.y;
  c_first = part * per_part * THREADS - head;
  c_end = min(kChunks, (part + 1) * per_part * THREADS - head);
  return c_first < c_end;
}
// launch(grid, geometry) once per slice of the batch a grid can hold (z <= 65 535: 524 280 frames)
template <class F>
hipError_t launch_raster_slices(uint64_t num_envs, int threads, int split, F&& launch) {
  const int steps = (kFrameBytes / 16 + threads - 1) / threads;
  // GYMX_RASTER_SLICE=<frames>: a smaller slice, so that tests reach the several-launch path with small batches
  static const uint64_t kSlice = [] {
    const char* e = std::getenv("GYMX_RASTER_SLICE");
    const long long v = e ? std::atoll(e) : 0;
    return v >= 1 && v <= 8ll * 65535ll ? (uint64_t)v : 8ull * 65535ull;
  }();
  for (uint64_t first = 0; first < num_envs; first += kSlice) {
    const uint64_t count = num_envs - first < kSlice ? num_envs - first : kSlice;
    RasterLaunch g{first, (uint32_t)count, (uint32_t)((count + 7) / 8), {(steps + split - 1) / split, (steps + split) / split}};
    launch(dim3(8u, (unsigned)split, g.envs_per_xcd), g);
  }
  return hipGetLastError();
}

// GYMX_FRAME_RASTER (Pong: GYMX_PONG_RASTER) = <threads>,<workgroups per frame>: the tuning aid of the three
// raster launchers
struct RasterShape { int threads, split; };
inline RasterShape raster_shape_from_env(const char* var, RasterShape dflt, int max_threads) {
  const char* s = std::getenv(var);
  int t = 0, sp = 0;
  if (s && std::sscanf(s, "%d,%d", &t, &sp) == 2 && t >= 64 && t <= max_threads && (t & (t - 1)) == 0 && sp >= 1 && sp <= 128)
    return RasterShape{t, sp};
  return dflt;
}

}  // namespace gx
