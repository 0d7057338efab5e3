// Breakout: game state stepped one env per lane in binary64 (the 6x18 brick
// field is 108 bits in four state words), and the 210x160x3 frame of every
// env rasterised by a separate bandwidth-bound kernel.
//
// The reference clears a 100 800-byte array and paints up to 108 brick
// rectangles, the paddle, the ball, the score digits and the life markers in
// that order with per-pixel bounds checks on every step
// (Envs.Breakout.cs:229-304).  Everything painted after the bricks is white,
// so the final colour of a pixel is: white if any white object covers it,
// else the colour of the brick row covering it, else black -- a pure function
// of the pixel position and the state, evaluated per 16-byte chunk (six pixels)
// as two bit masks, white and brick.
#pragma clang fp contract(off)

#include <cstdio>
#include <cstdlib>

#include "gx_device.h"
#include "gx_frame.h"
#include "gx_frame_step.h"
#include "gx_launch.h"

namespace gx {

namespace bk {
constexpr double kPaddleSpeed = 5.0, kBallSpeed = 3.0;
constexpr int kPaddleH = 4, kPaddleW = 16, kBallSize = 2;
constexpr int kBrickW = 8, kBrickH = 6, kBrickRows = 6, kBrickCols = 18, kBrickStartY = 60;
constexpr uint32_t kLastWordMask = (1u << (kBrickRows * kBrickCols - 96)) - 1u;  // 108 bits: 12 in word 3

struct State {
  double ballX, ballY, ballVelX, ballVelY, paddleX;
  int32_t score, lives, steps;
  bool stuck;
  uint32_t bricks[4];
  // the NextDouble() value the caller supplied for this env's next release of the ball (gx_vec_set_rng_draws);
  // negative: none, the generator is used
  double force0 = -1.0;
};

__device__ __forceinline__ double dmax(double a, double b) { return a >= b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a <= b ? a : b; }

// Envs.Breakout.cs:62-87
__device__ __forceinline__ void reset(State& s) {
  s.bricks[0] = s.bricks[1] = s.bricks[2] = 0xFFFFFFFFu;
  s.bricks[3] = kLastWordMask;
  s.paddleX = kScreenW / 2.0 - kPaddleW / 2.0;
  s.ballX = kScreenW / 2.0;
  s.ballY = kScreenH - 30;
  s.ballVelX = 0;
  s.ballVelY = 0;
  s.stuck = true;
  s.score = 0;
  s.lives = 5;
  s.steps = 0;
}

// Envs.Breakout.cs:89-199
__device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
  s.steps++;
  int rew = 0;
  if (action == 1) {
    if (s.stuck) {
      s.stuck = false;
      // the env's only random draw: in-step stream, draw 0, unless the caller supplied the value
      double u;
      if (s.force0 >= 0.0) {
        u = s.force0;
        s.force0 = -1.0;
      } else {
        u = philox_block(key, env, 1, 0).uniform(0);
      }
      s.ballVelX = (u - 0.5) * 2.0 * kBallSpeed;
      s.ballVelY = -kBallSpeed;
    }
  } else if (action == 2) {
    s.paddleX = dmin(kScreenW - kPaddleW, s.paddleX + kPaddleSpeed);
    if (s.stuck) s.ballX = s.paddleX + kPaddleW / 2.0;
  } else if (action == 3) {
    s.paddleX = dmax(0, s.paddleX - kPaddleSpeed);
    if (s.stuck) s.ballX = s.paddleX + kPaddleW / 2.0;
  }
  if (!s.stuck) {
    s.ballX += s.ballVelX;
    s.ballY += s.ballVelY;
    if (s.ballX <= 0 || s.ballX >= kScreenW - kBallSize) {
      s.ballVelX = -s.ballVelX;
      s.ballX = dmax(0, dmin(kScreenW - kBallSize, s.ballX));
    }
    if (s.ballY <= 0) {
      s.ballVelY = -s.ballVelY;
      s.ballY = 0;
    }
    if (s.ballY + kBallSize >= kScreenH - 20 && s.ballY + kBallSize <= kScreenH - 16 && s.ballX + kBallSize >= s.paddleX &&
        s.ballX <= s.paddleX + kPaddleW) {
      s.ballVelY = -fabs(s.ballVelY);
      const double relativeIntersectX = (s.ballX + kBallSize / 2.0) - (s.paddleX + kPaddleW / 2.0);
      const double normalizedIntersectX = relativeIntersectX / (kPaddleW / 2.0);
      s.ballVelX = normalizedIntersectX * kBallSpeed * 0.75;
    }
    // (int) truncates toward zero: a ball with 54 < y < 60 lands in brick row 0
    const int brickRow = (int)((s.ballY - kBrickStartY) / kBrickH);
    const int brickCol = (int)(s.ballX / kBrickW);
    if (brickRow >= 0 && brickRow < kBrickRows && brickCol >= 0 && brickCol < kBrickCols) {
      const int bit = brickRow * kBrickCols + brickCol;
      const uint32_t m = 1u << (bit & 31);
      bool hit = false;
#pragma unroll
      for (int w = 0; w < 4; w++)
        if ((bit >> 5) == w && (s.bricks[w] & m)) {
          s.bricks[w] &= ~m;
          hit = true;
        }
      if (hit) {
        s.ballVelY = -s.ballVelY;
        rew = (kBrickRows - brickRow) + 1;
        s.score += rew;
      }
    }
    if (s.ballY > kScreenH) {
      s.lives--;
      if (s.lives > 0) {
        s.ballX = s.paddleX + kPaddleW / 2.0;
        s.ballY = kScreenH - 30;
        s.ballVelX = 0;
        s.ballVelY = 0;
        s.stuck = true;
      }
    }
  }
  const bool allBricksGone = (s.bricks[0] | s.bricks[1] | s.bricks[2] | s.bricks[3]) == 0;
  if (allBricksGone) rew += 100;
  done = s.lives <= 0 || allBricksGone || s.steps >= 10000;
  reward = (float)rew;
}

// state image: reals ballX, ballY, ballVelX, ballVelY, paddleX; ints score, lives,
// ballStuck, four brick words, steps
__device__ __forceinline__ void load(const StepArgs& a, uint64_t i, State& s) {
  const double* r = static_cast<const double*>(a.state_real);
  s.ballX = r[i]; s.ballY = r[a.stride + i]; s.ballVelX = r[2 * a.stride + i]; s.ballVelY = r[3 * a.stride + i];
  s.paddleX = r[4 * a.stride + i];
  const int32_t* n = a.state_int;
  s.score = n[i]; s.lives = n[a.stride + i]; s.stuck = n[2 * a.stride + i] != 0;
#pragma unroll
  for (int w = 0; w < 4; w++) s.bricks[w] = (uint32_t)n[(3 + w) * a.stride + i];
  s.bricks[3] &= kLastWordMask;
  s.steps = n[7 * a.stride + i];
}
__device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) {
  double* r = static_cast<double*>(a.state_real);
  r[i] = s.ballX; r[a.stride + i] = s.ballY; r[2 * a.stride + i] = s.ballVelX; r[3 * a.stride + i] = s.ballVelY;
  r[4 * a.stride + i] = s.paddleX;
  int32_t* n = a.state_int;
  n[i] = s.score; n[a.stride + i] = s.lives; n[2 * a.stride + i] = s.stuck ? 1 : 0;
#pragma unroll
  for (int w = 0; w < 4; w++) n[(3 + w) * a.stride + i] = (int32_t)s.bricks[w];
  n[7 * a.stride + i] = s.steps;
}
}  // namespace bk

struct BreakoutGame {
  using State = bk::State;
  static constexpr int kActions = 4;  // Discrete(4), Envs.Breakout.cs:44
  static __device__ __forceinline__ void load(const StepArgs& a, uint64_t i, State& s) { bk::load(a, i, s); }
  static __device__ __forceinline__ void prime(const StepArgs& a, uint64_t i, State& s) {
    if (a.forced_draws) s.force0 = a.forced_draws[i];
  }
  static __device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) { bk::store(a, i, s); }
  static __device__ __forceinline__ void reset(State& s, const RngKey&, uint64_t) { bk::reset(s); }
  static __device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
    bk::step(s, action, key, env, reward, done);
  }
  static __device__ __forceinline__ int32_t steps(const State& s) { return s.steps; }
  // 0 (int)paddleX, 1-2 the ball's corner, 3 lives, 4 score, 5-8 the 108 brick bits
  static constexpr int kRecordDwords = 16;
  static __device__ __forceinline__ void pack(const State& s, uint32_t* record) {
    // (int) casts truncate toward zero, Envs.Breakout.cs:263-274
    const int px = (int)s.paddleX;
    const int bx = (int)(s.stuck ? s.paddleX + bk::kPaddleW / 2.0 - bk::kBallSize / 2.0 : s.ballX);
    const int by = s.stuck ? kScreenH - 30 : (int)s.ballY;
    int4* out = reinterpret_cast<int4*>(record);
    out[0] = make_int4(px, bx, by, s.lives);
    out[1] = make_int4(s.score, (int)s.bricks[0], (int)s.bricks[1], (int)s.bricks[2]);
    out[2] = make_int4((int)s.bricks[3], 0, 0, 0);
  }
};

// The frame in 16-byte chunks (gx_frame.h), `split` consecutive workgroups of THREADS lanes per env.
// Three rows in four hold nothing (everything but the score rows 10..14, the life markers 25..27, the brick
// rows 60..95, the paddle rows 190..193 and the ball's two rows) and are stored as zeros after one row test.
// Elsewhere a lane derives two six-pixel masks for its chunk -- white objects, and brick pixels of the row's
// colour -- and spreads them to bytes: white is 0xFF in every channel, so it may simply be OR-ed over the
// brick colour it hides (the reference paints bricks first and everything white after them,
// Envs.Breakout.cs:229-285).
template <int THREADS>
__global__ void __launch_bounds__(THREADS) breakout_raster_kernel(StepArgs a, uint8_t* __restrict__ frames, RasterLaunch g) {
  __shared__ uint32_t digit_rows[5][6];  // rows 10..14 as 160-bit masks, plus a zero word for the 64-bit window
  constexpr int kChunksPerRow = kScreenW * 3 / 16;
  uint64_t env;
  int c_first, c_end;
  if (!raster_part<THREADS>(g, env, c_first, c_end)) return;
  const uint32_t* rec = a.render + env * BreakoutGame::kRecordDwords;
  const int px = (int)rec[0], bx = (int)rec[1], by = (int)rec[2], lives = (int)rec[3], score = (int)rec[4];
  // the 108 brick bits as two 64-bit halves (no indexed array: it would be placed in LDS)
  const uint64_t bricks_lo = (uint64_t)rec[5] | ((uint64_t)rec[6] << 32);
  const uint64_t bricks_hi = (uint64_t)rec[7] | ((uint64_t)rec[8] << 32);
  // the score rows 10..14 are chunks 300..449: only a workgroup that covers some of them builds their masks
  if (c_first < 15 * kChunksPerRow && c_end > 10 * kChunksPerRow) {
    if (threadIdx.x < 30) digit_rows[threadIdx.x / 6][threadIdx.x % 6] = 0;
    __syncthreads();
    // Envs.Breakout.cs:306-314: character i of the score at x = 10 + 6 i, rows 10..14
    if (threadIdx.x < 5) score_row_mask(digit_rows[threadIdx.x], score, threadIdx.x, 10, 6, false);
    __syncthreads();
  }

  constexpr int kBrickRowsPx = bk::kBrickRows * bk::kBrickH, kPaddleY = kScreenH - 20;
  uint4* frame = reinterpret_cast<uint4*>(frames + env * (uint64_t)kFrameBytes);
  for (int c = c_first + (int)threadIdx.x; c < c_end; c += THREADS) {
    if (c < 0) continue;
    const int y = c / kChunksPerRow, j = c - y * kChunksPerRow;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    const bool brick_row = (uint32_t)(y - bk::kBrickStartY) < (uint32_t)kBrickRowsPx;
    if (brick_row || (uint32_t)(y - 10) < 5u || (uint32_t)(y - 25) < 3u || (uint32_t)(y - kPaddleY) < (uint32_t)bk::kPaddleH ||
        (uint32_t)(y - by) < (uint32_t)bk::kBallSize) {
      const int p0 = (16 * j) / 3, ph = j % 3;
      uint32_t white = 0;
      if ((uint32_t)(y - kPaddleY) < (uint32_t)bk::kPaddleH) white |= span_bits6(px, bk::kPaddleW, p0);
      if ((uint32_t)(y - by) < (uint32_t)bk::kBallSize) white |= span_bits6(bx, bk::kBallSize, p0);
      if ((uint32_t)(y - 10) < 5u) {
        const uint32_t* d = digit_rows[y - 10];
        const uint64_t win = (uint64_t)d[p0 >> 5] | ((uint64_t)d[(p0 >> 5) + 1] << 32);
        white |= (uint32_t)(win >> (p0 & 31)) & 63u;
      }
      if ((uint32_t)(y - 25) < 3u) {
        // life marker i covers x in [10 + 8 i, 16 + 8 i), Envs.Breakout.cs:281-284
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const int x = p0 + q - 10;
          if (x >= 0 && (x & 7) < 6 && (x >> 3) < lives) white |= 1u << q;
        }
      }
      v = expand_white6(white, ph);
      if (brick_row) {
        const int row = (y - bk::kBrickStartY) / bk::kBrickH, dy = (y - bk::kBrickStartY) - row * bk::kBrickH;
        // the window's pixels lie in the 8-pixel cells cell_a (the first n_a of them) and cell_a + 1; a brick is
        // drawn 7 pixels wide and 5 high, the last column and the last row of its cell stay black
        const int cell_a = p0 >> 3, n_a = min(6, 8 - (p0 & 7));
        const int bit_a = row * bk::kBrickCols + cell_a, bit_b = bit_a + 1;
        const bool has_a = cell_a < bk::kBrickCols && (((bit_a < 64 ? bricks_lo : bricks_hi) >> (bit_a & 63)) & 1ull);
        const bool has_b = cell_a + 1 < bk::kBrickCols && (((bit_b < 64 ? bricks_lo : bricks_hi) >> (bit_b & 63)) & 1ull);
        const uint32_t first = (1u << n_a) - 1u;
        uint32_t brick = (has_a ? first : 0u) | (has_b ? 63u & ~first : 0u);
        brick &= ~(1u << (7 - (p0 & 7)));  // the pixel with x mod 8 == 7 (bit 7 - p0 mod 8; beyond the window when >= 6)
        if (dy < bk::kBrickH - 1 && brick) {
          // Envs.Breakout.cs:248-253, r | g << 8 | b << 16
          const uint32_t rgb = row == 0 ? 0x0000FFu : row == 1 ? 0x00A5FFu : row == 2 ? 0x00FFFFu : row == 3 ? 0x00FF00u : row == 4 ? 0xFF0000u : 0x800080u;
          // the colour repeated over the chunk: a 96-bit pattern of three bytes, shifted by the phase
          const uint32_t w0 = rgb | (rgb << 24), w1 = (rgb >> 8) | (rgb << 16), w2 = (rgb >> 16) | (rgb << 8);
          const uint32_t sh = 8u * (uint32_t)ph;
          const uint32_t c0 = __builtin_amdgcn_alignbit(w1, w0, sh), c1 = __builtin_amdgcn_alignbit(w2, w1, sh);
          const uint32_t c2 = __builtin_amdgcn_alignbit(w0, w2, sh);
          const uint4 e = expand_white6(brick, ph);
          v.x |= e.x & c0; v.y |= e.y & c1; v.z |= e.z & c2; v.w |= e.w & c0;
        }
      }
    }
    frame[c] = v;
  }
}

hipError_t launch_breakout_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) { return launch_frame_step<BreakoutGame>(a, policy, grid, stream); }
hipError_t launch_breakout_reset(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_reset<BreakoutGame>(a, grid, stream); }
static_assert(BreakoutGame::kRecordDwords * 4 == kBreakoutRecordBytes, "the render records are allocated by this size");
hipError_t launch_breakout_pack(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_pack<BreakoutGame>(a, grid, stream); }
hipError_t launch_breakout_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream) {
  if (a.num_envs == 0) return hipSuccess;
  // 30 two-wave workgroups per frame, two 2 KiB steps each (see the Pong raster).  Measured at 65 536 envs (step +
  // raster, ms): 128 x 30 0.957, x 28 0.965, x 26 0.967, x 25 0.993; 256 x 13 0.964, x 14 0.982, x 12 1.02;
  // 64 x 34 1.00; three steps per wave 1.04-1.05; one 256-thread workgroup per frame 1.15
  // (tools/raster_shape_scan.sh, tools/raster_shape_ab.sh).
  static const RasterShape shape = raster_shape_from_env("GYMX_FRAME_RASTER", RasterShape{128, 30}, 256);
  return launch_raster_slices(a.num_envs, shape.threads, shape.split, [&](dim3 grid, const RasterLaunch& g) {
    if (shape.threads == 64) hipLaunchKernelGGL(breakout_raster_kernel<64>, grid, dim3(64), 0, stream, a, frames, g);
    else if (shape.threads == 128) hipLaunchKernelGGL(breakout_raster_kernel<128>, grid, dim3(128), 0, stream, a, frames, g);
    else hipLaunchKernelGGL(breakout_raster_kernel<256>, grid, dim3(256), 0, stream, a, frames, g);
  });
}
hipError_t launch_breakout_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) { return launch_frame_sample<BreakoutGame>(a, out, grid, stream); }

}  // namespace gx
