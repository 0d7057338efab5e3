// SpaceInvaders: game state stepped one env per lane in binary64, and the
// 210x160x3 frame of every env written by a workgroup, each lane deciding the
// sixteen bytes of a chunk (six pixels) at a time from the game state: every
// layer present in the row gives a 6-bit mask of the pixels it paints, applied
// in the reference's own drawing order.
//
// The reference keeps its bullets in List<(double,double)>
// (Envs.SpaceInvaders.cs:35-36).  Player bullets are capped at three by the
// source (:195).  Alien bullets are fired at most once every 60 steps (:136)
// and leave the screen within 40 (4.2 pixels per step from y >= 46), so a game
// never holds more than one; the state image has room for two and a third is
// dropped.  Both lists keep the source's order: RemoveAt moves later elements
// down, Add appends.
#pragma clang fp contract(off)

#include <cstdio>
#include <cstdlib>

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

// One workgroup per env, one 16-byte chunk (six pixels' bytes) per lane and iteration; nothing is staged: a
// lane decides what its chunk holds from the game state.  The layers are applied in the
// source's drawing order (Envs.SpaceInvaders.cs:355-411: aliens, cannon, player bullets, alien
// bullets, score, lives, ground line), a later layer overwriting an earlier one, which is what
// painting into a frame does; a pixel off the screen is never visited, which is the reference's
// DrawRectangle bounds check (:461-478).
template <int THREADS>
__global__ void __launch_bounds__(THREADS) invaders_raster_kernel(StepArgs a, uint8_t* __restrict__ frames, int split) {
  __shared__ uint32_t digit_rows[5][6];  // rows 10..14 of the frame as 160-bit masks (the score), plus a zero word for the 64-bit window
  __shared__ int32_t alien_x[si::kCols], alien_y[si::kRows];
  __shared__ uint8_t row_layers[kScreenH];  // which layers can touch a frame row (most rows: none)
  // the parts of a frame are consecutive workgroups, their steps laid on the frame buffer's own THREADS x 16-byte
  // grid (see the Breakout raster)
  const uint64_t env = blockIdx.x / (uint32_t)split;
  const int part = (int)(blockIdx.x % (uint32_t)split);
  const int tid = threadIdx.x;
  constexpr int kChunksPerRow = kScreenW * 3 / 16, kChunks = kFrameBytes / 16;
  const int head = (int)((env * (uint64_t)kChunks) & (THREADS - 1));
  const int steps = (head + kChunks + THREADS - 1) / THREADS;
  const int per_part = (steps + split - 1) / split;
  const int c_first = part * per_part * THREADS - head, c_end = min(kChunks, (part + 1) * per_part * THREADS - head);
  if (c_first >= c_end) return;
  const double* r = static_cast<const double*>(a.state_real);
  const int32_t* n = a.state_int;
  // every load is issued before the first barrier: one round trip to memory per workgroup
  const double formX = r[a.stride + env], formY = r[2 * a.stride + env];
  const int32_t score = n[env];
  const uint64_t aliens = (uint64_t)(uint32_t)n[6 * a.stride + env] | ((uint64_t)(uint32_t)n[7 * a.stride + env] << 32);
  const int player_x = (int)r[env];
  const int lives = n[a.stride + env];
  const int shown = lives < 0 ? 0 : (lives > 13 ? 13 : lives);  // cannon 13 would start at x = -10
  int npb = n[4 * a.stride + env], nab = n[5 * a.stride + env];
  npb = npb > si::kMaxPB ? si::kMaxPB : npb;
  nab = nab > si::kMaxAB ? si::kMaxAB : nab;
  int pbx[si::kMaxPB], pby[si::kMaxPB], abx[si::kMaxAB], aby[si::kMaxAB];
#pragma unroll
  for (int b = 0; b < si::kMaxPB; b++) { pbx[b] = (int)r[(3 + 2 * b) * a.stride + env]; pby[b] = (int)r[(4 + 2 * b) * a.stride + env]; }
#pragma unroll
  for (int b = 0; b < si::kMaxAB; b++) { abx[b] = (int)r[(9 + 2 * b) * a.stride + env]; aby[b] = (int)r[(10 + 2 * b) * a.stride + env]; }
  // column / row origins exactly as the source computes them, (int)(formation + 12 col) and
  // (int)(formation + 10 row): the cast truncates toward zero, so consecutive origins are 12 (10)
  // apart give or take one
  if (tid < si::kCols) alien_x[tid] = (int)(formX + tid * 12);
  if (tid < si::kRows) alien_y[tid] = (int)(formY + tid * 10);
  if (tid >= 32 && tid < 62) digit_rows[(tid - 32) / 6][(tid - 32) % 6] = 0;
  __syncthreads();
  // score, :403, 480-525: character i at x = 10 + 6 i, rows 10..14
  if (tid < 5) score_row_mask(digit_rows[tid], score, tid, 10, 6, false);
  const int x0 = alien_x[0], y0 = alien_y[0];
  // alien sprite rows, bit dx = column dx: .X..X. / ..XX.. / .XXXX. / XX..XX / XXXXXX / .X..X. (:413-435)
  const uint64_t sprite = 0x12ull | (0x0Cull << 6) | (0x1Eull << 12) | (0x33ull << 18) | (0x3Full << 24) | (0x12ull << 30);
  // cannon rows ...XX... / ..XXXX.. / ..XXXX.. / .XXXXXX. / XXXXXXXX x2 (:437-459)
  const uint64_t cannon = 0x18ull | (0x3Cull << 8) | (0x3Cull << 16) | (0x7Eull << 24) | (0xFFull << 32) | (0xFFull << 40);

  enum : uint32_t { kLayerAliens = 1, kLayerCannon = 2, kLayerPlayerBullets = 4, kLayerAlienBullets = 8, kLayerHud = 16, kLayerGround = 32 };
  // only the rows this workgroup writes
  for (int y = max(c_first, 0) / kChunksPerRow + tid; y <= (c_end - 1) / kChunksPerRow; y += THREADS) {
    uint32_t f = 0;
    // a row of the formation that still holds an alien and whose sprites cover y
#pragma unroll
    for (int row = 0; row < si::kRows; row++)
      if ((unsigned)(y - alien_y[row]) < 6u && ((aliens >> (row * si::kCols)) & ((1ull << si::kCols) - 1ull)) != 0ull) f |= kLayerAliens;
    if ((unsigned)(y - (kScreenH - 40)) < 6u) f |= kLayerCannon;
#pragma unroll
    for (int b = 0; b < si::kMaxPB; b++)
      if (b < npb && (unsigned)(y - pby[b]) < 4u) f |= kLayerPlayerBullets;
#pragma unroll
    for (int b = 0; b < si::kMaxAB; b++)
      if (b < nab && (unsigned)(y - aby[b]) < 4u) f |= kLayerAlienBullets;
    if ((unsigned)(y - 10) < 6u) f |= kLayerHud;
    if ((unsigned)(y - (kScreenH - 20)) < 2u) f |= kLayerGround;
    row_layers[y] = (uint8_t)f;
  }
  __syncthreads();

  // One 16-byte chunk per lane and iteration (gx_frame.h: the bytes of the six pixels p0 .. p0 + 5 of a row).
  // Per layer a lane builds the 6-bit mask of the window's pixels the layer paints, spreads it to byte
  // masks and lets the layer's colour replace what earlier layers left there.
  uint4* frame = reinterpret_cast<uint4*>(frames + env * (uint64_t)kFrameBytes);
  for (int c = c_first + tid; c < c_end; c += THREADS) {
    if (c < 0) continue;
    const int y = c / kChunksPerRow, j = c - y * kChunksPerRow;
    const uint32_t layers = row_layers[y];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (layers != 0u) {
      const int p0 = (16 * j) / 3, ph = j % 3;
      const uint32_t sh8 = 8u * (uint32_t)ph;
      // mask: bit q set = pixel p0 + q takes the colour rgb (r | g << 8 | b << 16)
      auto paint = [&](uint32_t mask, uint32_t rgb) {
        if (mask == 0u) return;
        const uint4 e = expand_white6(mask, ph);
        const uint32_t w0 = rgb | (rgb << 24), w1 = (rgb >> 8) | (rgb << 16), w2 = (rgb >> 16) | (rgb << 8);
        const uint32_t c0 = __builtin_amdgcn_alignbit(w1, w0, sh8), c1 = __builtin_amdgcn_alignbit(w2, w1, sh8);
        const uint32_t c2 = __builtin_amdgcn_alignbit(w0, w2, sh8);
        v.x = (v.x & ~e.x) | (e.x & c0); v.y = (v.y & ~e.y) | (e.y & c1);
        v.z = (v.z & ~e.z) | (e.z & c2); v.w = (v.w & ~e.w) | (e.w & c0);
      };
      // `bits`, a sprite row whose bit 0 is the pixel at x = left, as seen by the window
      auto window = [&](uint32_t bits, int left) -> uint32_t {
        const int sh = left - p0;
        return sh >= 6 || sh <= -32 ? 0u : (sh >= 0 ? bits << sh : bits >> -sh) & 63u;
      };
      // aliens, :361-385.  Row origins are 9..11 apart and a sprite is 6 tall, so the row whose sprite can cover
      // y is floor((y - y0 + 1) / 10); columns are 11..13 apart and 6 wide, so a sprite that meets the window
      // covers its first or its last pixel, and the column of a pixel x is floor((x - x0 + 1) / 12).
      if (layers & kLayerAliens) {
        int row = ((y - y0 + 1) * 205) >> 11;  // / 10, exact below 1024
        row = row > si::kRows - 1 ? si::kRows - 1 : (row < 0 ? 0 : row);
        const int dy = y - alien_y[row];
        if ((unsigned)dy < 6u) {
          const uint32_t bits = (uint32_t)(sprite >> (6 * dy)) & 63u;
          uint32_t m = 0;
#pragma unroll
          for (int end = 0; end < 2; end++) {
            const int d = p0 + 5 * end - x0 + 1;
            if (d >= 0 && d < 512) {
              const int col = (d * 171) >> 11;  // / 12, exact below 512
              if (col < si::kCols && ((aliens >> (row * si::kCols + col)) & 1ull)) m |= window(bits, alien_x[col]);
            }
          }
          paint(m, palette(kMagenta + (uint32_t)row));  // magenta, cyan, yellow, green, red
        }
      }
      // player cannon at ((int)playerX, 170), :388
      if (layers & kLayerCannon) paint(window((uint32_t)(cannon >> (8 * (y - (kScreenH - 40)))) & 0xFFu, player_x), palette(kGreen));
      // player bullets (yellow) then alien bullets (red), 2 x 4 pixels each, :391-400
      if (layers & kLayerPlayerBullets) {
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < si::kMaxPB; b++)
          if (b < npb && (unsigned)(y - pby[b]) < 4u) m |= span_bits6(pbx[b], 2, p0);
        paint(m, palette(kYellow));
      }
      if (layers & kLayerAlienBullets) {
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < si::kMaxAB; b++)
          if (b < nab && (unsigned)(y - aby[b]) < 4u) m |= span_bits6(abx[b], 2, p0);
        paint(m, palette(kRed));
      }
      if (layers & kLayerHud) {
        const int dy = y - 10;
        // score (white), :403
        if (dy < 5) {
          const uint32_t* dr = digit_rows[dy];
          const uint64_t win = (uint64_t)dr[p0 >> 5] | ((uint64_t)dr[(p0 >> 5) + 1] << 32);
          paint((uint32_t)(win >> (p0 & 31)) & 63u, palette(kWhite));
        }
        // lives (green cannons), :404, 489-495: cannon k covers x in [120 - 10 k, 127 - 10 k]; 8 wide and 10
        // apart, so one that meets the window covers its first or its last pixel
        const uint32_t bits = (uint32_t)(cannon >> (8 * dy)) & 0xFFu;
        uint32_t m = 0;
#pragma unroll
        for (int end = 0; end < 2; end++) {
          const int u = (kScreenW - 40) + 7 - (p0 + 5 * end);
          if (u >= 0) {
            const int k = (u * 205) >> 11;  // / 10
            if (k < shown) m |= window(bits, (kScreenW - 40) - 10 * k);
          }
        }
        paint(m, palette(kGreen));
      }
      // ground line, :407-410
      if (layers & kLayerGround) paint(63u, palette(kGreen));
    }
    frame[c] = v;
  }
}

hipError_t launch_invaders_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) { return launch_frame_step<InvadersGame>(a, policy, grid, stream); }
hipError_t launch_invaders_reset(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_reset<InvadersGame>(a, grid, stream); }
hipError_t launch_invaders_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream) {
  if (a.num_envs == 0) return hipSuccess;
  // Three consecutive one-wave workgroups per frame, 33 steps each: 1.19 ms at 65 536 envs (step + raster) against
  // 1.22 for one 256-thread workgroup per frame, 1.20 for 64 x 4, 1.22 for 128 x 2; five or more workgroups per
  // frame lose to the prologue every workgroup repeats (twenty state loads, three LDS tables, two barriers):
  // 64 x 5 1.54, 64 x 9 1.96.
  // Tuning aid: GYMX_FRAME_RASTER=<threads>,<workgroups per frame>, threads 64, 128 or 256.
  static int threads = 64, split = 3;
  static const bool parsed = [] {
    const char* s = std::getenv("GYMX_FRAME_RASTER");
    int t = 0, sp = 0;
    if (s && std::sscanf(s, "%d,%d", &t, &sp) == 2 && (t == 64 || t == 128 || t == 256) && sp >= 1 && sp <= 128) { threads = t; split = sp; }
    return true;
  }();
  (void)parsed;
  const dim3 grid((unsigned)(a.num_envs * (uint64_t)split));
  if (threads == 64) hipLaunchKernelGGL(invaders_raster_kernel<64>, grid, dim3(64), 0, stream, a, frames, split);
  else if (threads == 128) hipLaunchKernelGGL(invaders_raster_kernel<128>, grid, dim3(128), 0, stream, a, frames, split);
  else hipLaunchKernelGGL(invaders_raster_kernel<256>, grid, dim3(256), 0, stream, a, frames, split);
  return hipGetLastError();
}
hipError_t launch_invaders_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) { return launch_frame_sample<InvadersGame>(a, out, grid, stream); }

}  // namespace gx
