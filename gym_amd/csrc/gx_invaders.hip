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
  static __device__ __forceinline__ void prime(const StepArgs&, uint64_t, State&) {}
  static __device__ __forceinline__ void store(const StepArgs& a, uint64_t i, const State& s) { si::store(a, i, s); }
  static __device__ __forceinline__ void reset(State& s, const RngKey&, uint64_t) { si::reset(s); }
  static __device__ __forceinline__ void step(State& s, int action, const RngKey& key, uint64_t env, float& reward, bool& done) {
    si::step(s, action, key, env, reward, done);
  }
  static __device__ __forceinline__ int32_t steps(const State& s) { return s.steps; }
  // 0 (int)playerX, 1 score, 2 life cannons shown, 3 player bullets | alien bullets << 8 (both clamped to the
  // slots), 4-5 alien bits, 6-7 / 8-9 the formation's origin (doubles: column and row origins are casts of sums
  // with it), 10-12 / 13-15 player bullet x / y, 16-17 / 18-19 alien bullet x / y, 20-27 the rows
  // that are not blank; padded to a cache line
  static constexpr int kRecordDwords = 32, kRowMaskAt = 20;
  static __device__ __forceinline__ void pack(const State& s, uint32_t* record) {
    const int npb = s.npb > si::kMaxPB ? si::kMaxPB : (s.npb < 0 ? 0 : s.npb);
    const int nab = s.nab > si::kMaxAB ? si::kMaxAB : (s.nab < 0 ? 0 : s.nab);
    int4* out = reinterpret_cast<int4*>(record);
    // cannon 13 of the lives would start at x = -10
    out[0] = make_int4((int)s.playerX, s.score, s.lives < 0 ? 0 : (s.lives > 13 ? 13 : s.lives), npb | (nab << 8));
    out[1] = make_int4((int32_t)(uint32_t)s.aliens, (int32_t)(uint32_t)(s.aliens >> 32), __double2loint(s.formX), __double2hiint(s.formX));
    out[2] = make_int4(__double2loint(s.formY), __double2hiint(s.formY), (int)s.pbx[0], (int)s.pbx[1]);
    out[3] = make_int4((int)s.pbx[2], (int)s.pby[0], (int)s.pby[1], (int)s.pby[2]);
    out[4] = make_int4((int)s.abx[0], (int)s.abx[1], (int)s.aby[0], (int)s.aby[1]);
    // dwords 20-27: bit y set = something can touch frame row y (most rows are blank), the eighth dword zero
    uint32_t m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto add_rows = [&](int first, int count) {  // count <= 6; any first (a written state may hold anything)
      first = first < -64 ? -64 : (first > 512 ? 512 : first);
#pragma unroll
      for (int d = 0; d < 7; d++) {
        int lo = first - 32 * d, hi = lo + count;
        lo = lo < 0 ? 0 : lo;
        hi = hi > 32 ? 32 : hi;
        if (lo < hi) m[d] |= ((1u << (hi - lo)) - 1u) << lo;
      }
    };
#pragma unroll
    for (int q = 0; q < si::kRows; q++)
      if (((s.aliens >> (q * si::kCols)) & ((1ull << si::kCols) - 1ull)) != 0ull) add_rows((int)(s.formY + q * 10), 6);
    add_rows(kScreenH - 40, 6);  // the cannon
#pragma unroll
    for (int b = 0; b < si::kMaxPB; b++)
      if (b < npb) add_rows((int)s.pby[b], 4);
#pragma unroll
    for (int b = 0; b < si::kMaxAB; b++)
      if (b < nab) add_rows((int)s.aby[b], 4);
    add_rows(10, 6);              // score and lives
    add_rows(kScreenH - 20, 2);  // ground line
    out[5] = make_int4((int)m[0], (int)m[1], (int)m[2], (int)m[3]);
    out[6] = make_int4((int)m[4], (int)m[5], (int)m[6], 0);
  }
};
static_assert(si::kMaxPB == 3 && si::kMaxAB == 2, "the record's bullet slots");

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
__global__ void __launch_bounds__(THREADS) invaders_raster_kernel(StepArgs a, uint8_t* __restrict__ frames, RasterLaunch g) {
  __shared__ uint32_t digit_rows[5][6];  // rows 10..14 of the frame as 160-bit masks (the score), plus a zero word for the 64-bit window
  constexpr int kChunksPerRow = kScreenW * 3 / 16;
  const int tid = threadIdx.x;
  uint64_t env;
  int c_first, c_end;
  if (!raster_part<THREADS>(g, env, c_first, c_end)) return;
  // the env's record: one cache line, read with scalar loads (the address is the same for every lane)
  const uint32_t* rec = a.render + env * InvadersGame::kRecordDwords;
  const int player_x = (int)rec[0];
  const int32_t score = (int32_t)rec[1];
  const int shown = (int)rec[2];
  const int npb = (int)(rec[3] & 0xFFu), nab = (int)(rec[3] >> 8);
  const uint64_t aliens = (uint64_t)rec[4] | ((uint64_t)rec[5] << 32);
  const double formX = __hiloint2double((int)rec[7], (int)rec[6]), formY = __hiloint2double((int)rec[9], (int)rec[8]);
  const int pbx[si::kMaxPB] = {(int)rec[10], (int)rec[11], (int)rec[12]}, pby[si::kMaxPB] = {(int)rec[13], (int)rec[14], (int)rec[15]};
  const int abx[si::kMaxAB] = {(int)rec[16], (int)rec[17]}, aby[si::kMaxAB] = {(int)rec[18], (int)rec[19]};
  // column / row origins exactly as the source computes them, (int)(formation + 12 col) and
  // (int)(formation + 10 row): the cast truncates toward zero, so consecutive origins are 12 (10)
  // apart give or take one
  auto alien_x = [&](int col) { return (int)(formX + col * 12); };
  auto alien_y = [&](int row) { return (int)(formY + row * 10); };
  const int x0 = (int)formX, y0 = (int)formY;
  // 64 bits of the blank-row mask from the 32-row group of this workgroup's first row on; a workgroup whose rows
  // reach beyond them takes every row for painted
  const int y_lo = max(c_first, 0) / kChunksPerRow, y_hi = (c_end - 1) / kChunksPerRow, mask_base = y_lo & ~31;
  const uint32_t* mask_words = rec + InvadersGame::kRowMaskAt + (y_lo >> 5);
  const uint64_t painted_rows = y_hi - mask_base < 64 ? (uint64_t)mask_words[0] | ((uint64_t)mask_words[1] << 32) : ~0ull;
  // the score rows 10..14 are chunks 300..449: only a workgroup that covers some of them builds their masks
  // (:403, 480-525: character i at x = 10 + 6 i)
  if (c_first < 15 * kChunksPerRow && c_end > 10 * kChunksPerRow) {
    if (tid < 30) digit_rows[tid / 6][tid % 6] = 0;
    __syncthreads();
    if (tid < 5) score_row_mask(digit_rows[tid], score, tid, 10, 6, false);
    __syncthreads();
  }
  // alien sprite rows, bit dx = column dx: .X..X. / ..XX.. / .XXXX. / XX..XX / XXXXXX / .X..X. (:413-435)
  const uint64_t sprite = 0x12ull | (0x0Cull << 6) | (0x1Eull << 12) | (0x33ull << 18) | (0x3Full << 24) | (0x12ull << 30);
  // cannon rows ...XX... / ..XXXX.. / ..XXXX.. / .XXXXXX. / XXXXXXXX x2 (:437-459)
  const uint64_t cannon = 0x18ull | (0x3Cull << 8) | (0x3Cull << 16) | (0x7Eull << 24) | (0xFFull << 32) | (0xFFull << 40);

  // One 16-byte chunk per lane and iteration (gx_frame.h: the bytes of the six pixels p0 .. p0 + 5 of a row).
  // Per layer a lane builds the 6-bit mask of the window's pixels the layer paints, spreads it to byte
  // masks and lets the layer's colour replace what earlier layers left there.
  uint4* frame = reinterpret_cast<uint4*>(frames + env * (uint64_t)kFrameBytes);
  for (int c = c_first + tid; c < c_end; c += THREADS) {
    if (c < 0) continue;
    const int y = c / kChunksPerRow, j = c - y * kChunksPerRow;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((painted_rows >> ((y - mask_base) & 63)) & 1ull) {
      // Which layers touch this row.  Row origins of the formation are 9..11 apart and a sprite is 6 tall, so the one
      // row of aliens whose sprites can cover y is floor((y - y0 + 1) / 10).
      int row = ((y - y0 + 1) * 205) >> 11;  // / 10, exact below 1024
      row = row > si::kRows - 1 ? si::kRows - 1 : (row < 0 ? 0 : row);
      const int dy_alien = y - alien_y(row);
      const bool in_aliens = (unsigned)dy_alien < 6u && ((aliens >> (row * si::kCols)) & ((1ull << si::kCols) - 1ull)) != 0ull;
      const bool in_cannon = (unsigned)(y - (kScreenH - 40)) < 6u;
      bool in_pb = false, in_ab = false;
#pragma unroll
      for (int b = 0; b < si::kMaxPB; b++) in_pb = in_pb || (b < npb && (unsigned)(y - pby[b]) < 4u);
#pragma unroll
      for (int b = 0; b < si::kMaxAB; b++) in_ab = in_ab || (b < nab && (unsigned)(y - aby[b]) < 4u);
      const bool in_hud = (unsigned)(y - 10) < 6u, in_ground = (unsigned)(y - (kScreenH - 20)) < 2u;
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
      // aliens, :361-385.  Columns are 11..13 apart and 6 wide, so a sprite that meets the window covers its
      // first or its last pixel, and the column of a pixel x is floor((x - x0 + 1) / 12).
      if (in_aliens) {
        {
          const uint32_t bits = (uint32_t)(sprite >> (6 * dy_alien)) & 63u;
          uint32_t m = 0;
#pragma unroll
          for (int end = 0; end < 2; end++) {
            const int d = p0 + 5 * end - x0 + 1;
            if (d >= 0 && d < 512) {
              const int col = (d * 171) >> 11;  // / 12, exact below 512
              if (col < si::kCols && ((aliens >> (row * si::kCols + col)) & 1ull)) m |= window(bits, alien_x(col));
            }
          }
          paint(m, palette(kMagenta + (uint32_t)row));  // magenta, cyan, yellow, green, red
        }
      }
      // player cannon at ((int)playerX, 170), :388
      if (in_cannon) paint(window((uint32_t)(cannon >> (8 * (y - (kScreenH - 40)))) & 0xFFu, player_x), palette(kGreen));
      // player bullets (yellow) then alien bullets (red), 2 x 4 pixels each, :391-400
      if (in_pb) {
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < si::kMaxPB; b++)
          if (b < npb && (unsigned)(y - pby[b]) < 4u) m |= span_bits6(pbx[b], 2, p0);
        paint(m, palette(kYellow));
      }
      if (in_ab) {
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < si::kMaxAB; b++)
          if (b < nab && (unsigned)(y - aby[b]) < 4u) m |= span_bits6(abx[b], 2, p0);
        paint(m, palette(kRed));
      }
      if (in_hud) {
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
      if (in_ground) paint(63u, palette(kGreen));
    }
    frame[c] = v;
  }
}

hipError_t launch_invaders_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) { return launch_frame_step<InvadersGame>(a, policy, grid, stream); }
hipError_t launch_invaders_reset(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_reset<InvadersGame>(a, grid, stream); }
static_assert(InvadersGame::kRecordDwords * 4 == kInvadersRecordBytes, "the render records are allocated by this size");
hipError_t launch_invaders_pack(const StepArgs& a, dim3 grid, hipStream_t stream) { return launch_frame_pack<InvadersGame>(a, grid, stream); }
hipError_t launch_invaders_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream) {
  if (a.num_envs == 0) return hipSuccess;
  // Nine four-wave workgroups per frame, three 4 KiB steps each.  Measured at 65 536 envs (step + raster, ms):
  // 256 x 9 1.059, x 11 1.08, x 7 1.09, x 6 1.12, x 10 1.18, x 12 1.24; 128 x 18 1.062, x 17 1.065, x 16 1.075,
  // x 20 1.11; 64 x 34 1.074; two steps per wave 1.23-1.46: this raster spends about 50 vector and as many scalar
  // instructions per 16-byte store, twice what the other two do, and the shorter a wave the more its set-up weighs.
  static const RasterShape shape = raster_shape_from_env("GYMX_FRAME_RASTER", RasterShape{256, 9}, 256);
  return launch_raster_slices(a.num_envs, shape.threads, shape.split, [&](dim3 grid, const RasterLaunch& g) {
    if (shape.threads == 64) hipLaunchKernelGGL(invaders_raster_kernel<64>, grid, dim3(64), 0, stream, a, frames, g);
    else if (shape.threads == 128) hipLaunchKernelGGL(invaders_raster_kernel<128>, grid, dim3(128), 0, stream, a, frames, g);
    else hipLaunchKernelGGL(invaders_raster_kernel<256>, grid, dim3(256), 0, stream, a, frames, g);
  });
}
hipError_t launch_invaders_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) { return launch_frame_sample<InvadersGame>(a, out, grid, stream); }

}  // namespace gx
