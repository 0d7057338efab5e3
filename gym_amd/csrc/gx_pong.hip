// Pong: game state stepped one env per lane in binary64, and the 210x160x3
// frame of every env rasterised by a separate bandwidth-bound kernel.
//
// The reference clears a 100 800-byte array and draws ~30 rectangles with
// per-pixel bounds checks on every step (Envs.Pong.cs:167-209).  Every drawn
// pixel is (255,255,255), so draw order is irrelevant and the frame is a
// binary mask: each lane derives the six pixels that touch its sixteen bytes of
// a row from the object positions and writes them as one 16-byte store; a frame
// is written by 17 consecutive two-wave workgroups.
#pragma clang fp contract(off)

#include <cstdio>
#include <cstdlib>

#include "gx_device.h"
#include "gx_frame.h"
#include "gx_frame_step.h"
#include "gx_launch.h"

namespace gx {

constexpr double kPaddleSpeed = 4.0, kBallSpeed = 3.0;
constexpr int kPaddleH = 15, kPaddleW = 2, kBallSize = 2;

struct PongState {
  double ballX, ballY, ballVelX, ballVelY, leftPaddleY, rightPaddleY;
  int32_t leftScore, rightScore, steps;
  // the two NextDouble() values the caller supplied for this env's next ball reset (gx_vec_set_rng_draws);
  // force0 < 0: none, the generator is used
  double force0 = -1.0, force1 = 0.0;
};

// Envs.Pong.cs:159-165 (stream 0 from Reset, stream 1 from inside Step)
__device__ __forceinline__ void pong_reset_ball(PongState& p, const RngKey& key, uint64_t env, uint32_t stream) {
  double u0, u1;
  if (p.force0 >= 0.0) {  // supplied draws serve one ball reset
    u0 = p.force0;
    u1 = p.force1;
    p.force0 = -1.0;
  } else {
    Draws d = philox_block(key, env, stream, 0);
    u0 = d.uniform(0);
    u1 = d.uniform(1);
  }
  p.ballX = kScreenW / 2.0;
  p.ballY = kScreenH / 2.0;
  p.ballVelX = u0 > 0.5 ? kBallSpeed : -kBallSpeed;
  p.ballVelY = (u1 - 0.5) * 2.0 * kBallSpeed;
}

// Envs.Pong.cs:54-70
__device__ __forceinline__ void pong_reset(PongState& p, const RngKey& key, uint64_t env) {
  pong_reset_ball(p, key, env, 0);
  p.leftPaddleY = kScreenH / 2.0 - kPaddleH / 2.0;
  p.rightPaddleY = kScreenH / 2.0 - kPaddleH / 2.0;
  p.leftScore = 0;
  p.rightScore = 0;
  p.steps = 0;
}

// Math.Max / Math.Min on doubles (no NaN can occur here)
__device__ __forceinline__ double dmax(double a, double b) { return a >= b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a <= b ? a : b; }

// Envs.Pong.cs:72-145
__device__ __forceinline__ void pong_step(PongState& p, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
  p.steps++;
  float rew = 0.0f;
  if (action == 2) p.rightPaddleY = dmax(0, p.rightPaddleY - kPaddleSpeed);
  else if (action == 3) p.rightPaddleY = dmin(kScreenH - kPaddleH, p.rightPaddleY + kPaddleSpeed);

  const double ballCenterY = p.ballY + kBallSize / 2.0;
  const double leftPaddleCenterY = p.leftPaddleY + kPaddleH / 2.0;
  if (ballCenterY < leftPaddleCenterY - 2) p.leftPaddleY = dmax(0, p.leftPaddleY - kPaddleSpeed * 0.8);
  else if (ballCenterY > leftPaddleCenterY + 2) p.leftPaddleY = dmin(kScreenH - kPaddleH, p.leftPaddleY + kPaddleSpeed * 0.8);

  p.ballX += p.ballVelX;
  p.ballY += p.ballVelY;

  if (p.ballY <= 0 || p.ballY >= kScreenH - kBallSize) {
    p.ballVelY = -p.ballVelY;
    p.ballY = dmax(0, dmin(kScreenH - kBallSize, p.ballY));
  }
  if (p.ballX <= kPaddleW && p.ballY + kBallSize >= p.leftPaddleY && p.ballY <= p.leftPaddleY + kPaddleH) {
    p.ballVelX = fabs(p.ballVelX);
    const double rel = (p.leftPaddleY + kPaddleH / 2.0) - (p.ballY + kBallSize / 2.0);
    p.ballVelY = -rel * 0.1;
  }
  if (p.ballX + kBallSize >= kScreenW - kPaddleW && p.ballY + kBallSize >= p.rightPaddleY &&
      p.ballY <= p.rightPaddleY + kPaddleH) {
    p.ballVelX = -fabs(p.ballVelX);
    const double rel = (p.rightPaddleY + kPaddleH / 2.0) - (p.ballY + kBallSize / 2.0);
    p.ballVelY = -rel * 0.1;
  }
  if (p.ballX < 0) {
    p.rightScore++;
    rew = 1.0f;
    pong_reset_ball(p, key, env, 1);
  } else if (p.ballX > kScreenW) {
    p.leftScore++;
    rew = -1.0f;
    pong_reset_ball(p, key, env, 1);
  }
  done = p.leftScore >= 21 || p.rightScore >= 21 || p.steps >= 10000;
  reward = rew;
}

__device__ __forceinline__ void pong_load(const StepArgs& a, uint64_t i, PongState& p) {
  const double* r = static_cast<const double*>(a.state_real);
  p.ballX = r[i]; p.ballY = r[a.stride + i]; p.ballVelX = r[2 * a.stride + i]; p.ballVelY = r[3 * a.stride + i];
  p.leftPaddleY = r[4 * a.stride + i]; p.rightPaddleY = r[5 * a.stride + i];
  p.leftScore = a.state_int[i]; p.rightScore = a.state_int[a.stride + i]; p.steps = a.state_int[2 * a.stride + i];
}
__device__ __forceinline__ void pong_store(const StepArgs& a, uint64_t i, const PongState& p) {
  double* r = static_cast<double*>(a.state_real);
  r[i] = p.ballX; r[a.stride + i] = p.ballY; r[2 * a.stride + i] = p.ballVelX; r[3 * a.stride + i] = p.ballVelY;
  r[4 * a.stride + i] = p.leftPaddleY; r[5 * a.stride + i] = p.rightPaddleY;
  a.state_int[i] = p.leftScore; a.state_int[a.stride + i] = p.rightScore; a.state_int[2 * a.stride + i] = p.steps;
}

struct PongGame {
  using State = PongState;
  static constexpr int kActions = 4;  // Discrete(4), Envs.Pong.cs:36
  static __device__ __forceinline__ void load(const StepArgs& a, uint64_t i, State& s) { pong_load(a, i, s); }
  static __device__ __forceinline__ void prime(const StepArgs& a, uint64_t i, State& s) {
    if (a.forced_draws) {
      s.force0 = a.forced_draws[i];
      s.force1 = a.forced_draws[a.stride + i];
    }
  }
  static __device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) { pong_store(a, i, s); }
  static __device__ __forceinline__ void reset(State& s, const RngKey& key, uint64_t env) { pong_reset(s, key, env); }
  static __device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
    pong_step(s, action, key, env, reward, done);
  }
  static __device__ __forceinline__ int32_t steps(const State& s) { return s.steps; }
  // the ball's corner, the paddles' tops ((int) casts truncate toward zero, Envs.Pong.cs:173-179), the two scores
  static constexpr int kRecordDwords = 8;
  static __device__ __forceinline__ void pack(const State& s, uint32_t* record) {
    int4* out = reinterpret_cast<int4*>(record);
    out[0] = make_int4((int)s.ballX, (int)s.ballY, (int)s.leftPaddleY, (int)s.rightPaddleY);
    out[1] = make_int4(s.leftScore, s.rightScore, 0, 0);
  }
};

// One 16-byte store per lane.  A frame is 6300 chunks of 16 bytes, a row 30 of them;
// chunk j of a row holds row bytes [16 j, 16 j + 16), i.e. bytes of the six pixels p0 .. p0 + 5 with
// p0 = floor(16 j / 3), starting `phase` = 16 j mod 3 = j mod 3 bytes into pixel p0.  A lane derives the six
// pixel bits from the object positions and spreads them to sixteen bytes with a few bit operations.
// grid = (envs, split): workgroup (env, part) writes its share of the env's frame in steps of THREADS x 16
// bytes.  A frame starts 16-byte but not 4 KiB aligned (100 800 = 24 x 4096 + 2496), so the steps are laid
// on the frame buffer's own THREADS x 16-byte grid -- the first and the last step of a frame partly masked --
// and every wave's 1 KiB store lies within one aligned block.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) pong_raster16_kernel(StepArgs a, uint8_t* __restrict__ frames, RasterLaunch g) {
  __shared__ uint32_t digit_rows[5][6];  // rows 20..24 as 160-bit masks, plus a zero word for the 64-bit window
  constexpr int kChunksPerRow = kScreenW * 3 / 16;
  uint64_t env;
  int c_first, c_end;
  if (!raster_part<THREADS>(g, env, c_first, c_end)) return;

  const uint32_t* rec = a.render + env * PongGame::kRecordDwords;
  const int bx = (int)rec[0], by = (int)rec[1], ly = (int)rec[2], ry = (int)rec[3];
  // the score rows 20..24 are chunks 600..749: only a workgroup that covers some of them builds their masks
  if (c_first < 25 * kChunksPerRow && c_end > 20 * kChunksPerRow) {
    if (threadIdx.x < 30) digit_rows[threadIdx.x / 6][threadIdx.x % 6] = 0;
    __syncthreads();
    if (threadIdx.x < 10) {
      // Envs.Pong.cs:211-251: decimal string centred on centerX, digit i at startX + 8 i, rows 20..24
      const int side = threadIdx.x / 5, row = threadIdx.x % 5;
      const int score = (int)rec[4 + side];
      score_row_mask(digit_rows[row], score, row, side ? 3 * kScreenW / 4 : kScreenW / 4, 8, true);
    }
    __syncthreads();
  }

  uint4* frame = reinterpret_cast<uint4*>(frames + env * (uint64_t)kFrameBytes);
  // The paddles and the centre line sit in fixed chunks of a row (0, 29, and 14 / 15), so their sixteen bytes
  // are compile-time constants selected by the chunk index and one row test; only the ball's two rows and the
  // five score rows go through the general pixel-window path, which a wave skips when none of its lanes is
  // in such a row.  (40 vector instructions per 16-byte store; the general path for every chunk took 88.)
  constexpr int kRightChunk = kChunksPerRow - 1, kDashChunkA = (3 * (kScreenW / 2 - 1)) / 16, kDashChunkB = (3 * (kScreenW / 2)) / 16;
  constexpr Bytes16 kLeft = chunk_bytes(0, kPaddleW, 0), kRight = chunk_bytes(kScreenW - kPaddleW, kPaddleW, kRightChunk);
  constexpr Bytes16 kDashA = chunk_bytes(kScreenW / 2 - 1, 1, kDashChunkA), kDashB = chunk_bytes(kScreenW / 2, 1, kDashChunkB);
  static_assert(kDashChunkA == 14 && kDashChunkB == 15 && kLeft.z == 0 && kLeft.w == 0 && kRight.x == 0 && kRight.y == 0 &&
                kDashA.x == 0 && kDashA.y == 0 && kDashA.z == 0 && kDashB.y == 0 && kDashB.z == 0 && kDashB.w == 0, "chunk layout");
  for (int c = c_first + (int)threadIdx.x; c < c_end; c += THREADS) {
    if (c < 0) continue;
    const int y = c / kChunksPerRow, j = c - y * kChunksPerRow;
    const bool left = j == 0 && (uint32_t)(y - ly) < (uint32_t)kPaddleH;            // left paddle, x in {0,1}
    const bool right = j == kRightChunk && (uint32_t)(y - ry) < (uint32_t)kPaddleH;  // right paddle, x in {158,159}
    const bool dash = (y & 7) < 4;                                                    // centre dashes, x in {79,80}
    uint4 v;
    v.x = left ? kLeft.x : (dash && j == kDashChunkB ? kDashB.x : 0u);
    v.y = left ? kLeft.y : 0u;
    v.z = right ? kRight.z : 0u;
    v.w = right ? kRight.w : (dash && j == kDashChunkA ? kDashA.w : 0u);
    if ((uint32_t)(y - by) < (uint32_t)kBallSize || (uint32_t)(y - 20) < 5u) {
      const int p0 = (16 * j) / 3, ph = j % 3;
      uint32_t m = 0;
      if ((uint32_t)(y - by) < (uint32_t)kBallSize) m |= span_bits6(bx, kBallSize, p0);  // ball, x in {bx, bx+1}
      if ((uint32_t)(y - 20) < 5u) {
        const uint32_t* d = digit_rows[y - 20];
        const uint64_t win = (uint64_t)d[p0 >> 5] | ((uint64_t)d[(p0 >> 5) + 1] << 32);
        m |= (uint32_t)(win >> (p0 & 31)) & 63u;
      }
      const uint4 e = expand_white6(m, ph);
      v.x |= e.x; v.y |= e.y; v.z |= e.z; v.w |= e.w;
    }
    frame[c] = v;
  }
}

hipError_t launch_pong_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) { return launch_frame_step<PongGame>(a, policy, grid, stream); }
hipError_t launch_pong_reset(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_reset<PongGame>(a, grid, stream); }
static_assert(PongGame::kRecordDwords * 4 == kPongRecordBytes, "the render records are allocated by this size");
hipError_t launch_pong_pack(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_pack<PongGame>(a, grid, stream); }
hipError_t launch_pong_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream) {
  if (a.num_envs == 0) return hipSuccess;
  // 28 two-wave workgroups per frame, two 2 KiB steps each (a frame is 50 or 51 such steps: the last two or three
  // workgroups of a frame find nothing left and retire).  Measured at 65 536 envs (step + raster, ms): 128 x 28
  // 0.932, x 30 0.946, x 26 0.949, x 25 1.00; 256 x 14 0.938, x 13 0.950, x 12 1.01; 512 x 7 0.955; 64 x 34 0.97;
  // three steps per wave 1.01-1.04, one 256-thread workgroup walking a whole frame 1.2.  One-wave workgroups of two
  // steps would need more than 2.4 million workgroups a millisecond, which is the rate this grid is dealt at
  // (64 x 50 1.39, x 66 1.83, x 99 2.75).
  static const RasterShape shape = raster_shape_from_env("GYMX_PONG_RASTER", RasterShape{128, 28}, 1024);
  return launch_raster_slices(a.num_envs, shape.threads, shape.split, [&](dim3 grid, const RasterLaunch& g) {
    if (shape.threads == 64) hipLaunchKernelGGL(pong_raster16_kernel<64>, grid, dim3(64), 0, stream, a, frames, g);
    else if (shape.threads == 128) hipLaunchKernelGGL(pong_raster16_kernel<128>, grid, dim3(128), 0, stream, a, frames, g);
    else if (shape.threads == 512) hipLaunchKernelGGL(pong_raster16_kernel<512>, grid, dim3(512), 0, stream, a, frames, g);
    else if (shape.threads == 1024) hipLaunchKernelGGL(pong_raster16_kernel<1024>, grid, dim3(1024), 0, stream, a, frames, g);
    else hipLaunchKernelGGL(pong_raster16_kernel<256>, grid, dim3(256), 0, stream, a, frames, g);
  });
}
hipError_t launch_pong_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) { return launch_frame_sample<PongGame>(a, out, grid, stream); }

}  // namespace gx
