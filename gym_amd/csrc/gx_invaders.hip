// SpaceInvaders: game state stepped one env per lane in binary64, and the
// 210x160x3 frame of every env painted by a workgroup into a four-bit-per-
// pixel palette image in LDS (16.8 KB, so eight workgroups fit a CU), in the
// reference's own drawing order, then expanded to RGB with coalesced 12-byte
// stores.
//
// The reference keeps its bullets in List<(double,double)>
// (Envs.SpaceInvaders.cs:35-36).  Player bullets are capped at three by the
// source (:195).  Alien bullets are fired at most once every 60 steps (:136)
// and leave the screen within 40 (4.2 pixels per step from y >= 46), so a game
// never holds more than one; the state image has room for two and a third is
// dropped.  Both lists keep the source's order: RemoveAt moves later elements
// down, Add appends.
#pragma clang fp contract(off)

#include "gx_device.h"
#include "gx_frame.h"
#include "gx_frame_step.h"
#include "gx_launch.h"

namespace gx {

namespace si {
constexpr double kPlayerSpeed = 3.0, kBulletSpeed = 6.0, kAlienSpeed = 0.5, kAlienDropSpeed = 8.0;
constexpr int kPlayerW = 8, kPlayerH = 6, kAlienW = 6, kAlienH = 6;  // bullets are 2 x 4 (eight pixels, see the raster)
constexpr int kRows = 5, kCols = 11, kMaxPB = 3, kMaxAB = 2;
constexpr uint64_t kAllAliens = (1ull << (kRows * kCols)) - 1ull;

struct State {
  double playerX, formX, formY;
  double pbx[kMaxPB], pby[kMaxPB], abx[kMaxAB], aby[kMaxAB];
  int32_t score, lives, frame, npb, nab, steps;
  bool moveRight;
  uint64_t aliens;  // bit row * 11 + col
};

__device__ __forceinline__ double dmax(double a, double b) { return a >= b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a <= b ? a : b; }

// Envs.SpaceInvaders.cs:69-96
__device__ __forceinline__ void reset(State& s) {
  s.aliens = kAllAliens;
  s.playerX = kScreenW / 2.0 - kPlayerW / 2.0;
  s.formX = 10;
  s.formY = 40;
  s.moveRight = true;
  s.npb = 0;
  s.nab = 0;
#pragma unroll
  for (int i = 0; i < kMaxPB; i++) s.pbx[i] = s.pby[i] = 0;
#pragma unroll
  for (int i = 0; i < kMaxAB; i++) s.abx[i] = s.aby[i] = 0;
  s.score = 0;
  s.lives = 3;
  s.steps = 0;
  s.frame = 0;
}

// List.RemoveAt(i) on a register-resident list of N slots: later elements move down, the
// vacated slot reads 0.  Written without dynamic indexing.
template <int N>
__device__ __forceinline__ void remove_at(double (&x)[N], double (&y)[N], int32_t& n, int i) {
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (k >= i && k < n) {
      const int nx = k + 1 < N ? k + 1 : k;
      const bool has_next = k + 1 < n && k + 1 < N;
      x[k] = has_next ? x[nx] : 0.0;
      y[k] = has_next ? y[nx] : 0.0;
    }
  }
  n--;
}

// Envs.SpaceInvaders.cs:192-199
__device__ __forceinline__ void fire_player(State& s) {
  if (s.npb < kMaxPB) {
    const double x = s.playerX + kPlayerW / 2.0, y = kScreenH - 40;
#pragma unroll
    for (int k = 0; k < kMaxPB; k++)
      if (k == s.npb) { s.pbx[k] = x; s.pby[k] = y; }
    s.npb++;
  }
}

// Envs.SpaceInvaders.cs:98-190
__device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
  s.steps++;
  s.frame++;
  int rew = 0;
  if (action == 2 || action == 4) s.playerX = dmin(kScreenW - kPlayerW, s.playerX + kPlayerSpeed);
  if (action == 3 || action == 5) s.playerX = dmax(0, s.playerX - kPlayerSpeed);
  if (action == 1 || action == 4 || action == 5) fire_player(s);
  // UpdateBullets, :226-257 (last element first)
#pragma unroll
  for (int i = kMaxPB - 1; i >= 0; i--)
    if (i < s.npb) {
      s.pby[i] -= kBulletSpeed;
      if (s.pby[i] < 0) remove_at(s.pbx, s.pby, s.npb, i);
    }
#pragma unroll
  for (int i = kMaxAB - 1; i >= 0; i--)
    if (i < s.nab) {
      s.aby[i] += kBulletSpeed * 0.7;
      if (s.aby[i] > kScreenH) remove_at(s.abx, s.aby, s.nab, i);
    }
  // UpdateAliens every 10 frames, :259-279
  if (s.frame % 10 == 0) {
    if (s.moveRight) {
      s.formX += kAlienSpeed;
      if (s.formX + kCols * 12 > kScreenW - 10) {
        s.moveRight = false;
        s.formY += kAlienDropSpeed;
      }
    } else {
      s.formX -= kAlienSpeed;
      if (s.formX < 10) {
        s.moveRight = true;
        s.formY += kAlienDropSpeed;
      }
    }
  }
  // alien fire, :136-139 and :201-224: draw 0 decides, draw 1 picks among the columns that
  // still hold an alien (their lowest one shoots), in column order
  if (s.frame % 60 == 0) {
    const Draws d = philox_block(key, env, 1, 0);
    if (d.uniform(0) < 0.3) {
      int ns = 0;
      for (int col = 0; col < kCols; col++) {
        bool any = false;
        for (int row = 0; row < kRows; row++) any = any || ((s.aliens >> (row * kCols + col)) & 1ull);
        ns += any ? 1 : 0;
      }
      if (ns > 0) {
        const int pick = d.below(1, ns);
        int seen = 0, srow = 0, scol = 0;
        for (int col = 0; col < kCols; col++) {
          int low = -1;
          for (int row = 0; row < kRows; row++)
            if ((s.aliens >> (row * kCols + col)) & 1ull) low = row;
          if (low >= 0) {
            if (seen == pick) { srow = low; scol = col; }
            seen++;
          }
        }
        if (s.nab < kMaxAB) {
          const double x = s.formX + scol * 12 + kAlienW / 2.0, y = s.formY + srow * 10 + kAlienH;
#pragma unroll
          for (int k = 0; k < kMaxAB; k++)
            if (k == s.nab) { s.abx[k] = x; s.aby[k] = y; }
          s.nab++;
        }
      }
    }
  }
  // CheckCollisions, :281-339.  A bullet lies in at most one alien box (boxes are 6 wide on a
  // 12 pitch and 6 high on a 10 pitch), the scan order of the source is kept anyway.
#pragma unroll
  for (int i = kMaxPB - 1; i >= 0; i--)
    if (i < s.npb) {
      const double bx = s.pbx[i], by = s.pby[i];
      bool hit = false;
      for (int row = 0; row < kRows && !hit; row++)
        for (int col = 0; col < kCols && !hit; col++) {
          const int bit = row * kCols + col;
          if ((s.aliens >> bit) & 1ull) {
            const double alienX = s.formX + col * 12, alienY = s.formY + row * 10;
            if (bx >= alienX && bx <= alienX + kAlienW && by >= alienY && by <= alienY + kAlienH) {
              s.aliens &= ~(1ull << bit);
              const int points = row == 0 ? 30 : (row <= 2 ? 20 : 10);
              s.score += points;
              rew += points;
              hit = true;
            }
          }
        }
      if (hit) remove_at(s.pbx, s.pby, s.npb, i);
    }
#pragma unroll
  for (int i = kMaxAB - 1; i >= 0; i--)
    if (i < s.nab) {
      const double bx = s.abx[i], by = s.aby[i];
      if (bx >= s.playerX && bx <= s.playerX + kPlayerW && by >= kScreenH - 40 && by <= kScreenH - 40 + kPlayerH) {
        remove_at(s.abx, s.aby, s.nab, i);
        s.lives--;
        rew -= 50;
      }
    }
  if (s.formY + (kRows * 8) > kScreenH - 40) s.lives = 0;
  if (s.aliens == 0) {
    rew += 100;
    s.aliens = kAllAliens;
    s.formY = 40;
    s.formX = 10;
  }
  done = s.lives <= 0 || s.steps >= 10000;
  reward = (float)rew;
}

// state image: reals playerX, formX, formY, 3 player bullets (x, y), 2 alien bullets (x, y);
// ints score, lives, frameCounter, moveRight, player bullet count, alien bullet count, alien
// bits 0..31, alien bits 32..54, steps
__device__ __forceinline__ void load(const StepArgs& a, uint64_t i, State& s) {
  const double* r = static_cast<const double*>(a.state_real);
  s.playerX = r[i]; s.formX = r[a.stride + i]; s.formY = r[2 * a.stride + i];
#pragma unroll
  for (int k = 0; k < kMaxPB; k++) { s.pbx[k] = r[(3 + 2 * k) * a.stride + i]; s.pby[k] = r[(4 + 2 * k) * a.stride + i]; }
#pragma unroll
  for (int k = 0; k < kMaxAB; k++) { s.abx[k] = r[(9 + 2 * k) * a.stride + i]; s.aby[k] = r[(10 + 2 * k) * a.stride + i]; }
  const int32_t* n = a.state_int;
  s.score = n[i]; s.lives = n[a.stride + i]; s.frame = n[2 * a.stride + i]; s.moveRight = n[3 * a.stride + i] != 0;
  const int npb = n[4 * a.stride + i], nab = n[5 * a.stride + i];
  s.npb = npb < 0 ? 0 : (npb > kMaxPB ? kMaxPB : npb);
  s.nab = nab < 0 ? 0 : (nab > kMaxAB ? kMaxAB : nab);
  s.aliens = ((uint64_t)(uint32_t)n[6 * a.stride + i] | ((uint64_t)(uint32_t)n[7 * a.stride + i] << 32)) & kAllAliens;
  s.steps = n[8 * a.stride + i];
}
__device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) {
  double* r = static_cast<double*>(a.state_real);
  r[i] = s.playerX; r[a.stride + i] = s.formX; r[2 * a.stride + i] = s.formY;
#pragma unroll
  for (int k = 0; k < kMaxPB; k++) { r[(3 + 2 * k) * a.stride + i] = s.pbx[k]; r[(4 + 2 * k) * a.stride + i] = s.pby[k]; }
#pragma unroll
  for (int k = 0; k < kMaxAB; k++) { r[(9 + 2 * k) * a.stride + i] = s.abx[k]; r[(10 + 2 * k) * a.stride + i] = s.aby[k]; }
  int32_t* n = a.state_int;
  n[i] = s.score; n[a.stride + i] = s.lives; n[2 * a.stride + i] = s.frame; n[3 * a.stride + i] = s.moveRight ? 1 : 0;
  n[4 * a.stride + i] = s.npb; n[5 * a.stride + i] = s.nab;
  n[6 * a.stride + i] = (int32_t)(uint32_t)s.aliens; n[7 * a.stride + i] = (int32_t)(uint32_t)(s.aliens >> 32);
  n[8 * a.stride + i] = s.steps;
}
}  // namespace si

struct InvadersGame {
  using State = si::State;
  static constexpr int kActions = 6;  // Discrete(6), Envs.SpaceInvaders.cs:47
  static __device__ __forceinline__ void load(const StepArgs& a, uint64_t i, State& s) { si::load(a, i, s); }
  static __device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) { si::store(a, i, s); }
  static __device__ __forceinline__ void reset(State& s, const RngKey&, uint64_t) { si::reset(s); }
  static __device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
    si::step(s, action, key, env, reward, done);
  }
  static __device__ __forceinline__ int32_t steps(const State& s) { return s.steps; }
};

// palette of the frame: index -> r | g << 8 | b << 16
enum : uint32_t { kBlack = 0, kMagenta, kCyan, kYellow, kGreen, kRed, kWhite };
__device__ __forceinline__ uint32_t palette(uint32_t idx) {
  // Envs.SpaceInvaders.cs:374-378 alien rows, :393 yellow, :399 red, :455 and :409 green, :521 white
  return idx == kMagenta ? 0xFF00FFu : idx == kCyan ? 0xFFFF00u : idx == kYellow ? 0x00FFFFu : idx == kGreen ? 0x00FF00u :
         idx == kRed ? 0x0000FFu : idx == kWhite ? 0xFFFFFFu : 0u;
}

// One pixel of the palette image (eight 4-bit pixels per word) with the reference's
// DrawRectangle bounds check (Envs.SpaceInvaders.cs:461-478).  Clear-then-set with two LDS
// atomics: lanes painting other pixels of the same word do not interfere, and lanes painting
// the same pixel within a layer write the same colour, so the last clear is always followed
// by a set of that colour.
__device__ __forceinline__ void put(uint32_t* img, int px, int py, uint32_t c) {
  if (px >= 0 && px < kScreenW && py >= 0 && py < kScreenH) {
    const int idx = py * kScreenW + px, sh = (idx & 7) * 4;
    atomicAnd(&img[idx >> 3], ~(0xFu << sh));
    atomicOr(&img[idx >> 3], c << sh);
  }
}

// One workgroup per env.  Layers are painted in the source's order with a barrier between
// layers of different colour; within a layer every pixel receives one colour.
__global__ void __launch_bounds__(kBlock) invaders_raster_kernel(StepArgs a, uint8_t* __restrict__ frames) {
  constexpr int kImgWords = kScreenW * kScreenH / 8;
  __shared__ uint32_t img[kImgWords];
  const uint64_t env = blockIdx.x;
  const int tid = threadIdx.x;
  const double* r = static_cast<const double*>(a.state_real);
  const int32_t* n = a.state_int;
  for (int w = tid; w < kImgWords; w += kBlock) img[w] = 0;
  __syncthreads();

  // aliens: 55 x 36 sprite cells, Envs.SpaceInvaders.cs:361-385, 413-435
  {
    const double formX = r[a.stride + env], formY = r[2 * a.stride + env];
    const uint64_t aliens = (uint64_t)(uint32_t)n[6 * a.stride + env] | ((uint64_t)(uint32_t)n[7 * a.stride + env] << 32);
    // sprite rows, bit dx = column dx: .X..X. / ..XX.. / .XXXX. / XX..XX / XXXXXX / .X..X.
    const uint32_t sprite = 0x12u | (0x0Cu << 6) | (0x1Eu << 12) | (0x33u << 18) | (0x3Fu << 24);
    const uint32_t last_row = 0x12u;
    for (int c = tid; c < si::kRows * si::kCols * 36; c += kBlock) {
      const int cell = c / 36, p = c - cell * 36, dy = p / 6, dx = p - dy * 6;
      const int row = cell / si::kCols, col = cell - row * si::kCols;
      const uint32_t bits = dy < 5 ? (sprite >> (6 * dy)) & 0x3Fu : last_row;
      if (((aliens >> cell) & 1ull) && ((bits >> dx) & 1u)) {
        const int x = (int)(formX + col * 12), y = (int)(formY + row * 10);  // (int) truncates toward zero
        put(img, x + dx, y + dy, (uint32_t)(kMagenta + row));  // magenta, cyan, yellow, green, red
      }
    }
  }
  __syncthreads();
  // player cannon, :388, 437-459: rows ...XX... / ..XXXX.. / ..XXXX.. / .XXXXXX. / XXXXXXXX x2
  const uint64_t cannon = 0x18ull | (0x3Cull << 8) | (0x3Cull << 16) | (0x7Eull << 24) | (0xFFull << 32) | (0xFFull << 40);
  if (tid < 48) {
    const int dy = tid >> 3, dx = tid & 7;
    if ((cannon >> (8 * dy + dx)) & 1ull) put(img, (int)r[env] + dx, kScreenH - 40 + dy, kGreen);
  }
  __syncthreads();
  // player bullets (yellow) then alien bullets (red), :391-400
  {
    const int npb = n[4 * a.stride + env], nab = n[5 * a.stride + env];
    if (tid < 8 * si::kMaxPB) {
      const int b = tid >> 3, p = tid & 7;
      if (b < npb) put(img, (int)r[(3 + 2 * b) * a.stride + env] + (p & 1), (int)r[(4 + 2 * b) * a.stride + env] + (p >> 1), kYellow);
    }
    __syncthreads();
    if (tid < 8 * si::kMaxAB) {
      const int b = tid >> 3, p = tid & 7;
      if (b < nab) put(img, (int)r[(9 + 2 * b) * a.stride + env] + (p & 1), (int)r[(10 + 2 * b) * a.stride + env] + (p >> 1), kRed);
    }
  }
  __syncthreads();
  // score (white), :403, 480-525: character i at x = 10 + 6 i, rows 10..14
  {
    __shared__ uint32_t digit_rows[5][5];
    if (tid < 25) digit_rows[tid / 5][tid % 5] = 0;
    __syncthreads();
    if (tid < 5) score_row_mask(digit_rows[tid], n[env], tid, 10, 6, false);
    __syncthreads();
    for (int c = tid; c < 5 * kScreenW; c += kBlock) {
      const int dy = c / kScreenW, x = c - dy * kScreenW;
      if ((digit_rows[dy][x >> 5] >> (x & 31)) & 1u) put(img, x, 10 + dy, kWhite);
    }
  }
  __syncthreads();
  // lives (green cannons), :404, 489-495: cannon i at (120 - 10 i, 10); only those that can
  // touch the screen are visited
  {
    const int lives = n[a.stride + env];
    const int shown = lives < 0 ? 0 : (lives > 13 ? 13 : lives);  // x + 7 < 0 from i = 13 on
    for (int c = tid; c < shown * 48; c += kBlock) {
      const int i = c / 48, p = c - i * 48, dy = p >> 3, dx = p & 7;
      if ((cannon >> (8 * dy + dx)) & 1ull) put(img, (kScreenW - 40) - i * 10 + dx, 10 + dy, kGreen);
    }
  }
  __syncthreads();
  // ground line, :407-410
  for (int c = tid; c < 2 * kScreenW / 8; c += kBlock) img[(kScreenH - 20) * kScreenW / 8 + c] = 0x11111111u * kGreen;
  __syncthreads();

  uint8_t* frame = frames + env * (uint64_t)kFrameBytes;
  for (int c = tid; c < kGroupsPerFrame; c += kBlock) {
    const uint32_t w = img[c >> 1] >> ((c & 1) * 16);
    *reinterpret_cast<U3*>(frame + (uint64_t)c * 12) =
        pack_rgb4(palette(w & 0xFu), palette((w >> 4) & 0xFu), palette((w >> 8) & 0xFu), palette((w >> 12) & 0xFu));
  }
}

hipError_t launch_invaders_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) { return launch_frame_step<InvadersGame>(a, policy, grid, stream); }
hipError_t launch_invaders_reset(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_reset<InvadersGame>(a, grid, stream); }
hipError_t launch_invaders_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream) {
  if (a.num_envs == 0) return hipSuccess;
  hipLaunchKernelGGL(invaders_raster_kernel, dim3((unsigned)a.num_envs), dim3(kBlock), 0, stream, a, frames);
  return hipGetLastError();
}
hipError_t launch_invaders_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) { return launch_frame_sample<InvadersGame>(a, out, grid, stream); }

}  // namespace gx
