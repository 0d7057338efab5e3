// Pong: game state stepped one env per lane in binary64, and the 210x160x3
// frame of every env rasterised by a separate bandwidth-bound kernel.
//
// The reference clears a 100 800-byte array and draws ~30 rectangles with
// per-pixel bounds checks on every step (Envs.Pong.cs:167-209).  Every drawn
// pixel is (255,255,255), so draw order is irrelevant and the frame is a
// binary mask: each lane derives four consecutive pixels of a row from the
// object positions and writes them as one 12-byte store, a workgroup writing
// one env's frame front to back with fully coalesced stores.
#pragma clang fp contract(off)

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
};

// Envs.Pong.cs:159-165 (stream 0 from Reset, stream 1 from inside Step)
__device__ __forceinline__ void pong_reset_ball(PongState& p, const RngKey& key, uint64_t env, uint32_t stream) {
  Draws d = philox_block(key, env, stream, 0);
  p.ballX = kScreenW / 2.0;
  p.ballY = kScreenH / 2.0;
  p.ballVelX = d.uniform(0) > 0.5 ? kBallSpeed : -kBallSpeed;
  p.ballVelY = (d.uniform(1) - 0.5) * 2.0 * kBallSpeed;
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
  static __device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) { pong_store(a, i, s); }
  static __device__ __forceinline__ void reset(State& s, const RngKey& key, uint64_t env) { pong_reset(s, key, env); }
  static __device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
    pong_step(s, action, key, env, reward, done);
  }
  static __device__ __forceinline__ int32_t steps(const State& s) { return s.steps; }
};

// One workgroup per env.  A frame row is 160 pixels = 40 groups of four
// pixels = 40 lanes x 12 bytes; the frame is 8400 such groups.
__global__ void __launch_bounds__(kBlock) pong_raster_kernel(StepArgs a, uint8_t* __restrict__ frames) {
  __shared__ uint32_t digit_rows[5][5];  // rows 20..24 of the frame as 160-bit masks
  const uint64_t env = blockIdx.x;
  const double* r = static_cast<const double*>(a.state_real);
  // (int) casts truncate toward zero, Envs.Pong.cs:173-179
  const int bx = (int)r[env], by = (int)r[a.stride + env];
  const int ly = (int)r[4 * a.stride + env], ry = (int)r[5 * a.stride + env];
  if (threadIdx.x < 25) digit_rows[threadIdx.x / 5][threadIdx.x % 5] = 0;
  __syncthreads();
  if (threadIdx.x < 10) {
    // Envs.Pong.cs:211-251: decimal string centred on centerX, digit i at startX + 8 i, rows 20..24
    const int side = threadIdx.x / 5, row = threadIdx.x % 5;
    const int score = a.state_int[(uint64_t)side * a.stride + env];
    score_row_mask(digit_rows[row], score, row, side ? 3 * kScreenW / 4 : kScreenW / 4, 8, true);
  }
  __syncthreads();

  uint8_t* frame = frames + env * (uint64_t)kFrameBytes;
  for (int c = threadIdx.x; c < kScreenH * 40; c += kBlock) {
    const int y = c / 40, g = c - y * 40;
    uint32_t nib = 0;
    if (g == 0 && y >= ly && y < ly + kPaddleH) nib |= 0x3u;           // left paddle, x in {0,1}
    if (g == 39 && y >= ry && y < ry + kPaddleH) nib |= 0xCu;          // right paddle, x in {158,159}
    if ((y & 7) < 4) nib |= g == 19 ? 0x8u : (g == 20 ? 0x1u : 0u);    // centre dashes, x in {79,80}
    if (y == by || y == by + 1) {                                      // ball, x in {bx, bx+1}
      const int rel = bx - 4 * g;
      if (rel >= -1 && rel <= 3) nib |= ((0x3u << (rel + 1)) >> 1) & 0xFu;
    }
    if (y >= 20 && y < 25) nib |= (digit_rows[y - 20][g >> 3] >> (4 * (g & 7))) & 0xFu;
    // four pixels -> twelve bytes, three bytes of 0xFF per set pixel
    const uint32_t white = 0x00FFFFFFu;
    *reinterpret_cast<U3*>(frame + (uint64_t)c * 12) =
        pack_rgb4((0u - (nib & 1u)) & white, (0u - ((nib >> 1) & 1u)) & white, (0u - ((nib >> 2) & 1u)) & white,
                  (0u - ((nib >> 3) & 1u)) & white);
  }
}

hipError_t launch_pong_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) { return launch_frame_step<PongGame>(a, policy, grid, stream); }
hipError_t launch_pong_reset(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_reset<PongGame>(a, grid, stream); }
hipError_t launch_pong_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream) {
  if (a.num_envs == 0) return hipSuccess;
  hipLaunchKernelGGL(pong_raster_kernel, dim3((unsigned)a.num_envs), dim3(kBlock), 0, stream, a, frames);
  return hipGetLastError();
}
hipError_t launch_pong_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) { return launch_frame_sample<PongGame>(a, out, grid, stream); }

}  // namespace gx
