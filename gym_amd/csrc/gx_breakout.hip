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
// of the pixel position and the state, evaluated four pixels per lane.
#pragma clang fp contract(off)

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
      // the env's only random draw: in-step stream, draw 0
      s.ballVelX = (philox_block(key, env, 1, 0).uniform(0) - 0.5) * 2.0 * kBallSpeed;
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
  static __device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) { bk::store(a, i, s); }
  static __device__ __forceinline__ void reset(State& s, const RngKey&, uint64_t) { bk::reset(s); }
  static __device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
    bk::step(s, action, key, env, reward, done);
  }
  static __device__ __forceinline__ int32_t steps(const State& s) { return s.steps; }
};

// One workgroup per env, one four-pixel group per lane and iteration.
__global__ void __launch_bounds__(kBlock) breakout_raster_kernel(StepArgs a, uint8_t* __restrict__ frames) {
  __shared__ uint32_t digit_rows[5][5];  // rows 10..14 of the frame as 160-bit masks
  const uint64_t env = blockIdx.x;
  const double* r = static_cast<const double*>(a.state_real);
  const int32_t* n = a.state_int;
  const double paddleX = r[4 * a.stride + env];
  const bool stuck = n[2 * a.stride + env] != 0;
  // (int) casts truncate toward zero, Envs.Breakout.cs:263-274
  const int px = (int)paddleX;
  const int bx = stuck ? (int)(paddleX + bk::kPaddleW / 2.0 - bk::kBallSize / 2.0) : (int)r[env];
  const int by = stuck ? kScreenH - 30 : (int)r[a.stride + env];
  const int lives = n[a.stride + env];
  uint32_t bricks[4];
#pragma unroll
  for (int w = 0; w < 4; w++) bricks[w] = (uint32_t)n[(3 + w) * a.stride + env];
  if (threadIdx.x < 25) digit_rows[threadIdx.x / 5][threadIdx.x % 5] = 0;
  __syncthreads();
  // Envs.Breakout.cs:306-314: character i of the score at x = 10 + 6 i, rows 10..14
  if (threadIdx.x < 5) score_row_mask(digit_rows[threadIdx.x], n[env], threadIdx.x, 10, 6, false);
  __syncthreads();

  uint8_t* frame = frames + env * (uint64_t)kFrameBytes;
  for (int c = threadIdx.x; c < kGroupsPerFrame; c += kBlock) {
    const int y = c / kGroupsPerRow, g = c - y * kGroupsPerRow;
    uint32_t white = 0;
    if (y >= kScreenH - 20 && y < kScreenH - 20 + bk::kPaddleH) white |= span_mask(px, bk::kPaddleW, g);
    if (y == by || y == by + 1) white |= span_mask(bx, bk::kBallSize, g);
    if (y >= 10 && y < 15) white |= (digit_rows[y - 10][g >> 3] >> (4 * (g & 7))) & 0xFu;
    if (y >= 25 && y < 28) {
      // life marker i covers x in [10 + 8 i, 16 + 8 i), Envs.Breakout.cs:281-284
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int x = 4 * g + i - 10;
        if (x >= 0 && (x & 7) < 6 && (x >> 3) < lives) white |= 1u << i;
      }
    }
    uint32_t brick_rgb = 0, brick_px = 0;
    if (y >= bk::kBrickStartY && y < bk::kBrickStartY + bk::kBrickRows * bk::kBrickH && g < 2 * bk::kBrickCols) {
      const int row = (y - bk::kBrickStartY) / bk::kBrickH, dy = (y - bk::kBrickStartY) - row * bk::kBrickH;
      const int bit = row * bk::kBrickCols + (g >> 1);
      uint32_t word = bricks[0];
#pragma unroll
      for (int w = 1; w < 4; w++) word = (bit >> 5) == w ? bricks[w] : word;
      if (dy < bk::kBrickH - 1 && ((word >> (bit & 31)) & 1u)) {
        // a brick is drawn 7 pixels wide: the last column of its 8-pixel cell stays black
        brick_px = (g & 1) ? 0x7u : 0xFu;
        // Envs.Breakout.cs:248-253, r | g << 8 | b << 16
        brick_rgb = row == 0 ? 0x0000FFu : row == 1 ? 0x00A5FFu : row == 2 ? 0x00FFFFu : row == 3 ? 0x00FF00u : row == 4 ? 0xFF0000u : 0x800080u;
      }
    }
    uint32_t col[4];
#pragma unroll
    for (int i = 0; i < 4; i++) col[i] = ((white >> i) & 1u) ? 0xFFFFFFu : (((brick_px >> i) & 1u) ? brick_rgb : 0u);
    *reinterpret_cast<U3*>(frame + (uint64_t)c * 12) = pack_rgb4(col[0], col[1], col[2], col[3]);
  }
}

hipError_t launch_breakout_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) { return launch_frame_step<BreakoutGame>(a, policy, grid, stream); }
hipError_t launch_breakout_reset(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_reset<BreakoutGame>(a, grid, stream); }
hipError_t launch_breakout_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream) {
  if (a.num_envs == 0) return hipSuccess;
  hipLaunchKernelGGL(breakout_raster_kernel, dim3((unsigned)a.num_envs), dim3(kBlock), 0, stream, a, frames);
  return hipGetLastError();
}
hipError_t launch_breakout_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) { return launch_frame_sample<BreakoutGame>(a, out, grid, stream); }

}  // namespace gx
