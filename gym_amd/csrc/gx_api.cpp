// C ABI of libgymx.so (include/gymx.h): contexts, batches, host and zero-copy
// step paths, state get/set, statistics and the RCCL statistics all-reduce.
//
// There is no CPU implementation behind this file: every entry point that
// computes launches a gfx950 kernel, and fails with GX_ERR_HIP when no device
// is present.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/gymx.h"
#include "gx_device.h"
#include "gx_launch.h"

using namespace gx;

// ------------------------------------------------------------------ errors ---
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(e_ == hipErrorOutOfMemory ? GX_ERR_OOM : GX_ERR_HIP, "%s failed: %s", #expr, \
                  hipGetErrorString(e_));                                                     \
  } while (0)

// ----------------------------------------------------------------- handles ---
struct gx_ctx {
  int device = 0;
  int num_cus = 0;
  hipStream_t compute = nullptr;
  hipStream_t copy = nullptr;
  hipEvent_t ev_in = nullptr;
  hipEvent_t ev_out = nullptr;
  char* scratch = nullptr;  // this device's copy of g_rows_scratch (gx_rows_step)
  std::atomic<int> live_handles{0};  // batches and communicators created on this context and not yet destroyed (distinct handles may be driven from distinct threads)
};

struct gx_vec {
  gx_ctx* ctx = nullptr;
  gx_desc desc{};
  gx_info info{};
  std::string map;
  uint64_t n = 0, stride = 0, tick = 0;
  int envs_per_lane = 4;
  int block = 256;  // threads per workgroup of this batch's step kernels
  bool is_classic = false, is_tabular = false, is_frame = false, is_acrobot64 = false;  // is_frame: Pong, Breakout, SpaceInvaders
  bool is_classic64 = false;  // GX_MODE_FP64 on one of the five classic-control envs
  // Acrobot in EXACT mode: two more real rows (the angles) behind the six of the state image, see AcrobotAngles
  bool acrobot_angles = false;
  int classic_id = 0;  // what the classic-control launchers take: the env id, or kEnvAcrobotAngles
  bool is_blackjack = false;  // int-state env like the tabular ones (is_tabular is set too), with kernels of its own
  bool ep_return_f64 = false;
  // device memory
  void* d_real = nullptr;
  int32_t* d_int = nullptr;
  void* d_reward = nullptr;
  uint8_t* d_done = nullptr;
  uint8_t* d_trunc = nullptr;
  void* d_ep_return = nullptr;
  int32_t* d_ep_length = nullptr;
  StatSlot* d_slots = nullptr;  // [num_slots] (one per wave of the step grid) + 1 result slot
  int num_slots = 0;
  float* d_obs = nullptr;
  uint8_t* d_frames = nullptr;
  uint32_t* d_render = nullptr;  // frame envs: one packed record per env between the step and the raster
  uint32_t* d_table = nullptr;
  std::vector<void*> row_allocs;  // what hipMalloc returned for the row arrays below that start at an offset into their allocation
  size_t row_skew = 0;            // bytes: the k-th row array of a batch starts k * row_skew into its allocation
  int row_lane = 0;
  double* d_forced = nullptr;   // Pong [2][stride], Breakout [1][stride]: caller-supplied uniforms (gx_vec_set_rng_draws), allocated on first use
  bool forced_pending = false;  // the next step or reset launch hands d_forced to its kernel, once
  uint32_t* d_error = nullptr;
  void* d_actions = nullptr;  // staging for the host path
  uint32_t* h_error = nullptr;  // pinned
  StatSlot* h_stat = nullptr;   // pinned
  // Host path of small batches: pinned, device-mapped staging.  The step kernel reads the
  // actions from it, a gather kernel writes the requested results into it, gx_vec_wait copies
  // them to the caller's buffers.
  char* h_stage = nullptr;  // host address
  char* d_stage = nullptr;  // the same memory as the device addresses it
  size_t stage_reward = 0, stage_done = 0, stage_trunc = 0, stage_obs = 0, stage_obs_pitch = 0;  // actions sit at offset 0
  bool stage_pending = false;  // a staged step whose results gx_vec_wait still has to hand over
  void* pend_obs = nullptr;
  void* pend_reward = nullptr;
  uint8_t* pend_done = nullptr;
  uint8_t* pend_trunc = nullptr;
  bool frames_valid = false;
  // set by set_state: the state may violate the invariants the BOUNDED kernels rely on
  // until one step with auto-reset has run
  bool state_dirty = false;
  // A caller's own stream is never kept: it may be destroyed the moment the caller has synchronised it.  A step
  // launched on one whose outcome a later call of the library must wait for (the invalid-action flag; a launch that
  // reads caller-supplied draws) is followed on that stream by a record of this event, which the library owns.
  hipEvent_t ev_foreign = nullptr;
  bool foreign_pending = false;  // ev_foreign was recorded after the most recent such launch
  uint32_t sweep_parity = 0;  // flips every step launch: alternate launches sweep in opposite directions
  // the open-episode return is ret_per_step x the step counter and its row is not maintained
  // (fused auto-reset + statistics on CartPole, MountainCar, Acrobot); ends at the first set_state
  bool derive_returns = false;
  int ret_per_step = 0;
  // tabular
  int table_states = 0, table_actions = 0, table_outcomes = 1, start_state = 0, step_limit = 0;
  int table_lut = 8;  // reward values behind the packed entries
  std::vector<gx_transition> user_table;  // GX_ENV_TABLE: the caller's entries until they are compiled (gx_vec_create)
  size_t table_lds_bytes = 0;
  dim3 grid;
  unsigned grid_single = 0;  // workgroups of a single step where that differs from `grid` (large classic-control batches); 0: the same
};

struct gx_comm {
  gx_ctx* ctx = nullptr;
  void* lib = nullptr;
  void* comm = nullptr;
  double* d_buf = nullptr;
  double* h_buf = nullptr;
  int rank = 0, world = 1;
};

static size_t dtype_size(int dt) { return dt == GX_F64 ? 8 : (dt == GX_U8 ? 1 : 4); }

// ------------------------------------------------------------------ library ---
extern "C" uint32_t gx_version(void) { return GX_VERSION; }
extern "C" const char* gx_last_error(void) { return g_err; }

extern "C" int gx_device_count(int* count) {
  if (!count) return fail(GX_ERR_INVALID, "gx_device_count: null argument");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(GX_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *count = n;
  return GX_OK;
}

// ------------------------------------------------- gx_rows_step's scratch ---
// gx_rows_step allocates nothing: the per-wave statistics records of its step kernel, the
// reduced record and the invalid-action flag live in this array, of which every GPU the library
// runs on holds its own copy at its own address until the process ends.  Host code never uses
// the array's name as an address: gx_ctx_create asks the runtime for this device's copy once
// and keeps the pointer in the context.
constexpr uint64_t kRowsMaxEnvs = GX_ROWS_MAX_ENVS;
// One record per wave of the largest grid gx_vec_create gives a batch of GX_ROWS_MAX_ENVS envs: 2^24 envs / 4 per
// lane / 64 lanes = 65 536 one-wave workgroups' worth of work, run as 4 4/9 rounds of the 8 192 workgroups a 256-CU
// part holds at once = 36 408 workgroups; 40 960 leaves room for parts with other CU counts (the grid is clamped to it).
constexpr int kRowsMaxSlots = 40960;
constexpr size_t kRowsResultOffset = (size_t)kRowsMaxSlots * sizeof(StatSlot);
constexpr size_t kRowsErrorOffset = kRowsResultOffset + sizeof(StatSlot);
constexpr size_t kRowsScratchBytes = kRowsErrorOffset + 256;  // 2.5 MiB + 320 bytes
static __device__ __align__(256) unsigned char g_rows_scratch[kRowsScratchBytes];

// ------------------------------------------------------------------ context ---
extern "C" int gx_ctx_create(int device, gx_ctx** out) {
  if (!out) return fail(GX_ERR_INVALID, "gx_ctx_create: null out pointer");
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(GX_ERR_INVALID, "gx_ctx_create: device %d of %d", device, n);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(GX_ERR_UNSUPPORTED, "gx_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
  gx_ctx* c = new (std::nothrow) gx_ctx();
  if (!c) return fail(GX_ERR_OOM, "gx_ctx_create: host allocation failed");
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  hipError_t e = hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
  if (e != hipSuccess) {
    gx_ctx_destroy(c);
    return fail(GX_ERR_HIP, "gx_ctx_create: stream/event creation failed: %s", hipGetErrorString(e));
  }
  // this device is current: the address returned is the one of its copy of the array
  void* scratch = nullptr;
  e = hipGetSymbolAddress(&scratch, HIP_SYMBOL(g_rows_scratch));
  if (e != hipSuccess || !scratch) {
    gx_ctx_destroy(c);
    return fail(GX_ERR_HIP, "gx_ctx_create: hipGetSymbolAddress(g_rows_scratch) failed: %s", hipGetErrorString(e));
  }
  c->scratch = static_cast<char*>(scratch);
  *out = c;
  return GX_OK;
}

extern "C" int gx_ctx_destroy(gx_ctx* c) {
  if (!c) return GX_OK;
  // a batch's destructor synchronises this context's streams: the context must outlive it
  if (c->live_handles > 0)
    return fail(GX_ERR_INVALID, "gx_ctx_destroy: %d batch(es) / communicator(s) of this context are still alive; destroy them first", c->live_handles.load());
  (void)hipSetDevice(c->device);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->compute) (void)hipStreamDestroy(c->compute);
  if (c->copy) (void)hipStreamDestroy(c->copy);
  delete c;
  return GX_OK;
}

extern "C" int gx_ctx_synchronize(gx_ctx* c) {
  if (!c) return fail(GX_ERR_INVALID, "gx_ctx_synchronize: null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->compute));
  HIP_TRY(hipStreamSynchronize(c->copy));
  return GX_OK;
}

extern "C" int gx_ctx_stream(gx_ctx* c, void** stream) {
  if (!c || !stream) return fail(GX_ERR_INVALID, "gx_ctx_stream: null argument");
  *stream = c->compute;
  return GX_OK;
}

// ------------------------------------------------- tabular table compilation ---
// Packs an env into entry[(s * (A + 1) + col) * O + o] followed by 8 rewards.
namespace {

constexpr uint32_t kTerminalBit = 1u << 24;
const char* const kFrozenLake4x4 = "SFFF/FHFH/FFFH/HFFG";  // Envs.FrozenLake.cs:25-30
const char* const kFrozenLake8x8 = "SFFFFFFF/FFFFFFFF/FFFHFFFF/FFFFFHFF/FFFHFFFF/FHHFFFHF/FHFFHFHF/FFFHFFFG";

struct Grid {
  int nrow = 0, ncol = 0, goal = -1;
  std::vector<uint8_t> hole;
};

// Envs.FrozenLake.cs:31-42
int parse_map(const std::string& rows, Grid& g) {
  std::vector<std::string> lines;
  std::string cur;
  for (char ch : rows) {
    if (ch == '/' || ch == '\n') {
      if (!cur.empty()) lines.push_back(cur);
      cur.clear();
    } else {
      cur.push_back(ch);
    }
  }
  if (!cur.empty()) lines.push_back(cur);
  if (lines.empty()) return -1;
  g.nrow = (int)lines.size();
  g.ncol = (int)lines[0].size();
  if ((long long)g.nrow * g.ncol > 4096) return -1;
  g.hole.assign((size_t)g.nrow * g.ncol, 0);
  g.goal = -1;
  for (int r = 0; r < g.nrow; r++) {
    if ((int)lines[r].size() != g.ncol) return -1;
    for (int c = 0; c < g.ncol; c++) {
      if (lines[r][c] == 'G') g.goal = r * g.ncol + c;
      if (lines[r][c] == 'H') g.hole[(size_t)r * g.ncol + c] = 1;
    }
  }
  return 0;
}

// Envs.FrozenLake.cs:54-63 / Envs.CliffWalking.cs:38-47
int grid_move(const Grid& g, int s, int a) {
  int row = s / g.ncol, col = s % g.ncol;
  switch (a) {
    case 0: col = col - 1 > 0 ? col - 1 : 0; break;
    case 1: row = row + 1 < g.nrow - 1 ? row + 1 : g.nrow - 1; break;
    case 2: col = col + 1 < g.ncol - 1 ? col + 1 : g.ncol - 1; break;
    case 3: row = row - 1 > 0 ? row - 1 : 0; break;
    default: break;
  }
  return row * g.ncol + col;
}

// upstream Taxi-v3 transition (extension; the reference's Taxi is a stub)
void taxi_full(int s, int a, int& ns, int& ridx, bool& term) {
  static const int loc_r[4] = {0, 0, 4, 4}, loc_c[4] = {0, 4, 0, 3};
  static const uint8_t wall_e[5][5] = {
      {0, 1, 0, 0, 1}, {0, 1, 0, 0, 1}, {0, 0, 0, 0, 1}, {1, 0, 1, 0, 1}, {1, 0, 1, 0, 1}};
  int dest = s % 4, pass = (s / 4) % 5, col = (s / 20) % 5, row = s / 100;
  ridx = 0;  // -1
  term = false;
  switch (a) {
    case 0: row = row + 1 < 4 ? row + 1 : 4; break;
    case 1: row = row - 1 > 0 ? row - 1 : 0; break;
    case 2: if (!wall_e[row][col]) col += 1; break;
    case 3: if (col > 0 && !wall_e[row][col - 1]) col -= 1; break;
    case 4:
      if (pass < 4 && row == loc_r[pass] && col == loc_c[pass]) pass = 4;
      else ridx = 1;  // -10
      break;
    case 5: {
      int at = -1;
      for (int k = 0; k < 4; k++)
        if (row == loc_r[k] && col == loc_c[k]) at = k;
      if (pass == 4 && at == dest) { pass = dest; ridx = 2; term = true; }  // +20
      else if (pass == 4 && at >= 0) pass = at;
      else ridx = 1;
      break;
    }
    default: break;
  }
  ns = ((row * 5 + col) * 5 + pass) * 4 + dest;
}

// GX_ENV_TABLE: the caller's P[s][a][o] -> (next state, reward, done), checked and packed like the built-in envs'
// tables.  Column A, the built-in envs' "any other action" case, repeats action 0: an action outside the space is an
// error here (the host path rejects it, the zero-copy path raises the batch's error flag).
int build_user_table(gx_vec* v, std::vector<uint32_t>& tab, std::vector<float>& lut) {
  const gx_desc& d = v->desc;
  const int S = d.n_states, A = d.n_actions, O = d.n_outcomes == 0 ? 1 : d.n_outcomes;
  if (v->user_table.empty()) return fail(GX_ERR_INVALID, "gx_vec_create: GX_ENV_TABLE needs gx_desc.table");
  if (S < 1 || S > 65536) return fail(GX_ERR_INVALID, "gx_vec_create: n_states %d outside 1..65536", S);
  if (A < 1 || A > 255) return fail(GX_ERR_INVALID, "gx_vec_create: n_actions %d outside 1..255", A);
  if (O != 1 && O != 3) return fail(GX_ERR_UNSUPPORTED, "gx_vec_create: n_outcomes %d (1, or 3 equally likely outcomes)", O);
  if (v->user_table.size() != (size_t)S * A * O)
    return fail(GX_ERR_INVALID, "gx_vec_create: table_len %zu, expected n_states * n_actions * n_outcomes = %zu", v->user_table.size(), (size_t)S * A * O);
  if (d.start_state < -1 || d.start_state >= S) return fail(GX_ERR_INVALID, "gx_vec_create: start_state %d outside -1..%d", d.start_state, S - 1);
  if (d.step_limit < 0) return fail(GX_ERR_INVALID, "gx_vec_create: negative step_limit");
  const bool stats = (d.flags & GX_RECORD_STATS) != 0;
  lut.clear();
  tab.assign((size_t)S * (A + 1) * O, 0);
  for (int s = 0; s < S; s++)
    for (int col = 0; col <= A; col++)
      for (int o = 0; o < O; o++) {
        const int a = col < A ? col : 0;
        const size_t at = ((size_t)s * A + a) * O + o;
        const gx_transition& t = v->user_table[at];
        if (t.next_state < 0 || t.next_state >= S)
          return fail(GX_ERR_INVALID, "gx_vec_create: table entry %zu (state %d, action %d, outcome %d) leads to state %d outside 0..%d", at, s, a, o, t.next_state, S - 1);
        if (!std::isfinite(t.reward)) return fail(GX_ERR_INVALID, "gx_vec_create: table entry %zu has a reward that is not finite", at);
        if (stats && (t.reward != std::nearbyint(t.reward) || std::fabs(t.reward) > 1048576.0f))
          return fail(GX_ERR_UNSUPPORTED, "gx_vec_create: table entry %zu pays %g; with GX_RECORD_STATS rewards must be integers of magnitude at most 2^20", at, (double)t.reward);
        size_t ridx = 0;
        while (ridx < lut.size() && std::memcmp(&lut[ridx], &t.reward, sizeof(float)) != 0) ridx++;  // by bits: -0.0f and 0.0f stay apart
        if (ridx == lut.size()) {
          if (lut.size() == 256) return fail(GX_ERR_UNSUPPORTED, "gx_vec_create: the table pays more than 256 distinct rewards");
          lut.push_back(t.reward);
        }
        tab[((size_t)s * (A + 1) + col) * O + o] = (uint32_t)t.next_state | ((uint32_t)ridx << 16) | (t.done ? kTerminalBit : 0u);
      }
  lut.resize(std::max<size_t>(8, (lut.size() + 3) & ~(size_t)3), 0.0f);
  v->table_states = S; v->table_actions = A; v->table_outcomes = O;
  v->start_state = d.start_state;
  v->step_limit = d.step_limit > 0 ? d.step_limit : std::numeric_limits<int32_t>::max();
  return GX_OK;
}

int build_table(gx_vec* v, std::vector<uint32_t>& tab) {
  const int env = v->desc.env_id;
  float lut[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<float> user_lut;
  if (env == GX_ENV_TABLE) {
    int rc = build_user_table(v, tab, user_lut);
    if (rc != GX_OK) return rc;
  } else
  if (env == GX_ENV_FROZENLAKE || env == GX_ENV_CLIFFWALKING) {
    Grid g;
    const bool cliff = env == GX_ENV_CLIFFWALKING;
    if (cliff) {
      // Envs.CliffWalking.cs:15-27
      g.nrow = 4; g.ncol = 12; g.goal = 47;
      g.hole.assign(48, 0);
      for (int c = 1; c < 11; c++) g.hole[36 + c] = 1;
      v->start_state = 36;
      lut[0] = -1.0f; lut[1] = -100.0f; lut[2] = 0.0f;  // Envs.CliffWalking.cs:50-51
    } else {
      std::string rows = v->map.empty() ? kFrozenLake4x4 : (v->map == "8x8" ? kFrozenLake8x8 : v->map);
      if (parse_map(rows, g) != 0) return fail(GX_ERR_INVALID, "gx_vec_create: malformed FrozenLake map");
      v->start_state = 0;
      lut[0] = 0.0f; lut[1] = 1.0f;  // Envs.FrozenLake.cs:66
    }
    const int S = g.nrow * g.ncol, A = 4;
    const int O = (!cliff && (v->desc.flags & GX_SLIPPERY)) ? 3 : 1;
    v->table_states = S; v->table_actions = A; v->table_outcomes = O;
    v->step_limit = 100;  // Envs.FrozenLake.cs:65, Envs.CliffWalking.cs:49
    v->info.obs_dim = 1;
    tab.assign((size_t)S * (A + 1) * O + 8, 0);
    for (int s = 0; s < S; s++)
      for (int col = 0; col <= A; col++)
        for (int o = 0; o < O; o++) {
          // slippery: the executed action is a-1, a or a+1 (mod 4)
          int a = col < A ? (O == 3 ? ((col + 3 + o) & 3) : col) : -1;
          int ns = grid_move(g, s, a);
          const bool goal = ns == g.goal, hole = g.hole[(size_t)ns] != 0;
          uint32_t ridx = cliff ? (goal ? 2u : (hole ? 1u : 0u)) : (goal ? 1u : 0u);
          tab[((size_t)s * (A + 1) + col) * O + o] = (uint32_t)ns | (ridx << 16) | ((goal || hole) ? kTerminalBit : 0u);
        }
  } else if (env == GX_ENV_TAXI) {
    const int S = 500, A = 6;
    const bool full = (v->desc.flags & GX_TAXI_FULL) != 0;
    v->table_states = S; v->table_actions = A; v->table_outcomes = 1;
    v->step_limit = 200;  // Envs.Taxi.cs:29
    v->start_state = full ? -2 : -1;
    lut[0] = -1.0f; lut[1] = -10.0f; lut[2] = 20.0f;  // Envs.Taxi.cs:30 uses -1 only
    tab.assign((size_t)S * (A + 1) + 8, 0);
    for (int s = 0; s < S; s++)
      for (int col = 0; col <= A; col++) {
        int ns = s, ridx = 0;
        bool term = false;
        if (full) taxi_full(s, col < A ? col : -1, ns, ridx, term);
        tab[(size_t)s * (A + 1) + col] = (uint32_t)ns | ((uint32_t)ridx << 16) | (term ? kTerminalBit : 0u);
      }
  } else {
    return GX_OK;
  }
  const float* rewards = lut;
  if (env == GX_ENV_TABLE) {
    const size_t entries = tab.size();
    tab.resize(entries + user_lut.size());
    std::memcpy(&tab[entries], user_lut.data(), user_lut.size() * sizeof(float));
    v->table_lut = (int)user_lut.size();
    rewards = user_lut.data();
  } else {
    std::memcpy(&tab[tab.size() - 8], lut, sizeof lut);
  }
  // Does every non-terminal transition pay the same integer reward?  Then, with auto-reset on,
  // the open-episode return is that reward times the step counter (see derive_returns).
  {
    bool constant = true, seen = false;
    float c = 0.0f;
    for (size_t i = 0; i + (size_t)v->table_lut < tab.size(); i++) {
      if (tab[i] & kTerminalBit) continue;
      const float r = rewards[(tab[i] >> 16) & 0xFFu];
      if (!seen) { c = r; seen = true; }
      else if (r != c) constant = false;
    }
    const uint32_t both = GX_AUTORESET | GX_RECORD_STATS;
    if (seen && constant && c == (float)(int)c && (v->desc.flags & both) == both) {
      v->derive_returns = true;
      v->ret_per_step = (int)c;
    }
  }
  return GX_OK;
}

void fill_info(gx_vec* v) {
  gx_info& in = v->info;
  const int env = v->desc.env_id;
  in.num_envs = v->n;
  in.env_id = env;
  in.real_dtype = GX_F32;
  in.obs_dtype = GX_F32;
  in.action_dtype = GX_I32;
  in.reward_dtype = GX_F32;
  in.obs_is_state = 1;
  in.action_low = in.action_high = 0.0f;
  switch (env) {
    case GX_ENV_CARTPOLE: in.n_real = 4; in.n_int = 1; in.obs_dim = 4; in.action_n = 2; break;   // Envs.CartPole.cs:12-16
    case GX_ENV_MOUNTAINCAR: in.n_real = 2; in.n_int = 1; in.obs_dim = 2; in.action_n = 3; break; // Envs.MountainCar.cs:12-16
    case GX_ENV_MOUNTAINCAR_CONT:                                                                 // Envs.MountainCarContinuous.cs:12-13
      in.n_real = 2; in.n_int = 1; in.obs_dim = 2; in.action_n = 0; in.action_dtype = GX_F32;
      in.action_low = -1.0f; in.action_high = 1.0f; in.reward_dtype = GX_F64;
      break;
    case GX_ENV_PENDULUM:                                                                         // Envs.Pendulum.cs:12-13
      in.n_real = 3; in.n_int = 1; in.obs_dim = 3; in.action_n = 0; in.action_dtype = GX_F32;
      in.action_low = -2.0f; in.action_high = 2.0f; in.reward_dtype = GX_F64;
      break;
    case GX_ENV_ACROBOT:                                                                          // Envs.Acrobot.cs:12-16
      in.obs_dim = 6; in.action_n = 3; in.n_int = 1;
      if (v->is_acrobot64) { in.n_real = 4; in.real_dtype = GX_F64; in.obs_is_state = 0; }
      else in.n_real = 6;
      break;
    case GX_ENV_FROZENLAKE: in.n_real = 0; in.n_int = 2; in.obs_dim = 1; in.obs_dtype = GX_I32; in.action_n = 4; break;
    case GX_ENV_TAXI: in.n_real = 0; in.n_int = 2; in.obs_dim = 1; in.obs_dtype = GX_I32; in.action_n = 6; break;
    case GX_ENV_CLIFFWALKING: in.n_real = 0; in.n_int = 2; in.obs_dim = 1; in.obs_dtype = GX_I32; in.action_n = 4; break;
    case GX_ENV_DUMMY: in.n_real = 0; in.n_int = 2; in.obs_dim = 1; in.obs_dtype = GX_I32; in.action_n = 3; break;
    case GX_ENV_PONG:                                                                             // Envs.Pong.cs:36-41
      in.n_real = 6; in.real_dtype = GX_F64; in.n_int = 3; in.obs_dim = 210 * 160 * 3; in.obs_dtype = GX_U8;
      in.action_n = 4; in.obs_is_state = 0;
      break;
    case GX_ENV_BREAKOUT:                                                                         // Envs.Breakout.cs:44-48
      in.n_real = 5; in.real_dtype = GX_F64; in.n_int = 8; in.obs_dim = 210 * 160 * 3; in.obs_dtype = GX_U8;
      in.action_n = 4; in.obs_is_state = 0;
      break;
    case GX_ENV_SPACEINVADERS:                                                                    // Envs.SpaceInvaders.cs:47-51
      in.n_real = 13; in.real_dtype = GX_F64; in.n_int = 9; in.obs_dim = 210 * 160 * 3; in.obs_dtype = GX_U8;
      in.action_n = 6; in.obs_is_state = 0;
      break;
    case GX_ENV_BLACKJACK: in.n_real = 0; in.n_int = 3; in.obs_dim = 1; in.obs_dtype = GX_I32; in.action_n = 2; break;  // Envs.Blackjack.cs:12-13
    case GX_ENV_TABLE: in.n_real = 0; in.n_int = 2; in.obs_dim = 1; in.obs_dtype = GX_I32; in.action_n = v->desc.n_actions; break;
  }
  if (v->is_classic64) in.real_dtype = in.obs_dtype = GX_F64;  // the observation is the binary64 state image
  in.state_bytes = v->n * ((uint64_t)in.n_real * dtype_size(in.real_dtype) + (uint64_t)in.n_int * 4);
  in.obs_bytes = v->n * (uint64_t)in.obs_dim * dtype_size(in.obs_dtype);
}

// the frame envs: game-state kernel, then the raster kernel over every env's frame
hipError_t frame_step(const gx_vec* v, const StepArgs& a, int policy, hipStream_t stream) {
  switch (v->desc.env_id) {
    case GX_ENV_PONG: return launch_pong_step(a, policy, v->grid, stream);
    case GX_ENV_BREAKOUT: return launch_breakout_step(a, policy, v->grid, stream);
    default: return launch_invaders_step(a, policy, v->grid, stream);
  }
}
hipError_t frame_reset(const gx_vec* v, const StepArgs& a, hipStream_t stream) {
  switch (v->desc.env_id) {
    case GX_ENV_PONG: return launch_pong_reset(a, v->grid, stream);
    case GX_ENV_BREAKOUT: return launch_breakout_reset(a, v->grid, stream);
    default: return launch_invaders_reset(a, v->grid, stream);
  }
}
// the render records again, after the states were written from outside
hipError_t frame_pack(const gx_vec* v, const StepArgs& a, hipStream_t stream) {
  switch (v->desc.env_id) {
    case GX_ENV_PONG: return launch_pong_pack(a, v->grid, stream);
    case GX_ENV_BREAKOUT: return launch_breakout_pack(a, v->grid, stream);
    default: return launch_invaders_pack(a, v->grid, stream);
  }
}
hipError_t frame_raster(const gx_vec* v, const StepArgs& a, hipStream_t stream) {
  switch (v->desc.env_id) {
    case GX_ENV_PONG: return launch_pong_raster(a, v->d_frames, stream);
    case GX_ENV_BREAKOUT: return launch_breakout_raster(a, v->d_frames, stream);
    default: return launch_invaders_raster(a, v->d_frames, stream);
  }
}

StepArgs make_args(const gx_vec* v, const void* d_actions, int k_steps, uint64_t action_stride) {
  StepArgs a{};
  a.state_real = v->d_real;
  a.state_int = v->d_int;
  a.actions = d_actions;
  a.reward = v->d_reward;
  a.done = v->d_done;
  a.truncated = v->d_trunc;
  a.ep_return = v->d_ep_return;
  a.ep_length = v->d_ep_length;
  a.stat_slots = v->d_slots;
  a.obs = v->d_obs;
  a.table = v->d_table;
  a.render = v->d_render;
  a.error_flag = v->d_error;
  a.num_envs = v->n;
  a.stride = v->stride;
  a.action_stride = action_stride;
  a.env_index_base = v->desc.env_index_base;
  a.tick = v->tick;
  a.seed_lo = (uint32_t)v->desc.seed;
  a.seed_hi = (uint32_t)(v->desc.seed >> 32);
  a.flags = v->desc.flags;
  a.time_limit = v->desc.time_limit;
  a.k_steps = k_steps;
  a.reverse = v->sweep_parity;
  a.table_states = v->table_states;
  a.table_actions = v->table_actions;
  a.table_outcomes = v->table_outcomes;
  a.table_lut = v->table_lut;
  a.strict_actions = v->desc.env_id == GX_ENV_TABLE ? 1 : 0;
  a.start_state = v->start_state;
  a.step_limit = v->step_limit;
  a.derive_returns = v->derive_returns ? 1 : 0;
  a.ret_per_step = v->ret_per_step;
  a.forced_draws = v->forced_pending ? v->d_forced : nullptr;
  return a;
}

// Only Acrobot has actions the reference throws on (Envs.Acrobot.cs:45,51); the host path
// validates them before launch, so the device flag matters for the zero-copy path alone.
// A caller-supplied table has no column for an action outside its space and treats one the same way.
bool has_error_flag(const gx_vec* v) { return v->desc.env_id == GX_ENV_ACROBOT || v->desc.env_id == GX_ENV_TABLE; }

// waits for the most recent launch on a caller's stream that the library has to order itself after
int wait_foreign(gx_vec* v) {
  if (v->foreign_pending) {
    HIP_TRY(hipEventSynchronize(v->ev_foreign));
    v->foreign_pending = false;
  }
  return GX_OK;
}

// per-step output rows of a rollout (gx_vec_rollout_out)
struct StepOut {
  void* reward = nullptr;
  uint8_t* done = nullptr;
  uint8_t* truncated = nullptr;
  bool skip_last = false;
};

int launch_step(gx_vec* v, const void* d_actions, int k_steps, uint64_t action_stride, int policy, hipStream_t stream,
                const StepOut* out = nullptr) {
  StepArgs a = make_args(v, d_actions, k_steps, action_stride);
  if (out) {
    a.out_reward = out->reward;
    a.out_done = out->done;
    a.out_truncated = out->truncated;
    a.skip_last = out->skip_last ? 1 : 0;
  }
  hipError_t e;
  const dim3 grid = (k_steps == 1 && v->grid_single) ? dim3(v->grid_single) : v->grid;
  const uint32_t both = GX_AUTORESET | GX_RECORD_STATS;
  const bool fused = (v->desc.flags & both) == both && v->desc.time_limit == 0;
  const bool bounded = (v->desc.flags & GX_AUTORESET) && !v->state_dirty;
  if (v->is_acrobot64) e = launch_acrobot64_step(a, policy, fused, grid, stream);
  else if (v->is_classic64) e = launch_classic64_step(v->desc.env_id, a, policy, fused, bounded, grid, stream);
  else if (v->is_classic) e = launch_classic_step(v->classic_id, a, v->desc.mode, policy, fused, bounded, grid, stream);
  else if (v->is_blackjack) e = launch_blackjack_step(a, policy, grid, stream);
  else if (v->is_tabular) e = launch_tabular_step(v->desc.env_id, a, policy, grid, v->block, v->table_lds_bytes, stream);
  else {
    e = frame_step(v, a, policy, stream);
    if (e == hipSuccess) e = frame_raster(v, a, stream);
    v->frames_valid = true;
  }
  if (e != hipSuccess) return fail(GX_ERR_HIP, "step kernel launch failed: %s", hipGetErrorString(e));
  v->tick += (uint64_t)k_steps;
  v->forced_pending = false;
  v->state_dirty = false;
  v->sweep_parity ^= 1u;
  if (stream != v->ctx->compute && (a.forced_draws || (has_error_flag(v) && policy == GX_POLICY_DEVICE_ACTIONS))) {
    HIP_TRY(hipEventRecord(v->ev_foreign, stream));
    v->foreign_pending = true;
  }
  return GX_OK;
}

int launch_sample(gx_vec* v, uint64_t tick, void* d_out, hipStream_t stream) {
  StepArgs a = make_args(v, nullptr, 1, 0);
  a.tick = tick;
  uint64_t blocks = (v->n + kBlock - 1) / kBlock;
  dim3 grid((unsigned)std::min<uint64_t>(blocks ? blocks : 1, (uint64_t)v->ctx->num_cus * 8));
  hipError_t e;
  if (v->is_acrobot64) e = launch_acrobot64_sample(a, d_out, grid, stream);
  else if (v->is_classic64) e = launch_classic64_sample(v->desc.env_id, a, d_out, grid, stream);
  else if (v->is_classic) e = launch_classic_sample(v->classic_id, a, d_out, grid, stream);
  else if (v->is_blackjack) e = launch_blackjack_sample(a, d_out, grid, stream);
  else if (v->is_tabular) e = launch_tabular_sample(v->desc.env_id, a, d_out, grid, stream);
  else if (v->desc.env_id == GX_ENV_PONG) e = launch_pong_sample(a, d_out, grid, stream);
  else if (v->desc.env_id == GX_ENV_BREAKOUT) e = launch_breakout_sample(a, d_out, grid, stream);
  else e = launch_invaders_sample(a, d_out, grid, stream);
  if (e != hipSuccess) return fail(GX_ERR_HIP, "sample kernel launch failed: %s", hipGetErrorString(e));
  return GX_OK;
}

// copy `rows` SoA rows of `n` elements between a packed host image and the
// strided device image
int copy_rows(void* dst, const void* src, size_t elem, int rows, uint64_t n, uint64_t dst_stride, uint64_t src_stride,
              hipMemcpyKind kind, hipStream_t stream) {
  if (rows == 0 || n == 0) return GX_OK;
  HIP_TRY(hipMemcpy2DAsync(dst, dst_stride * elem, src, src_stride * elem, n * elem, (size_t)rows, kind, stream));
  return GX_OK;
}

}  // namespace

// -------------------------------------------------------------------- batch ---
extern "C" int gx_vec_destroy(gx_vec* v) {
  if (!v) return GX_OK;
  if (v->ctx) {
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->compute);
    (void)hipStreamSynchronize(v->ctx->copy);
  }
  for (void* p : v->row_allocs) (void)hipFree(p);  // d_real, d_int, d_reward, d_done, d_trunc, d_actions, d_ep_return, d_ep_length
  void* dev[] = {v->d_slots, v->d_obs, v->d_frames, v->d_render, v->d_table, v->d_error, v->d_forced};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (v->h_error) (void)hipHostFree(v->h_error);
  if (v->h_stage) (void)hipHostFree(v->h_stage);
  if (v->h_stat) (void)hipHostFree(v->h_stat);
  if (v->ev_foreign) (void)hipEventDestroy(v->ev_foreign);
  if (v->ctx) v->ctx->live_handles--;
  delete v;
  return GX_OK;
}

extern "C" int gx_vec_create(gx_ctx* ctx, const gx_desc* desc, gx_vec** out) {
  if (!ctx || !desc || !out) return fail(GX_ERR_INVALID, "gx_vec_create: null argument");
  *out = nullptr;
  static_assert(offsetof(gx_desc, table) == GX_DESC_SIZE_V1, "the fields before gx_desc.table are the first version of the struct");
  if (desc->struct_size != sizeof(gx_desc) && desc->struct_size != GX_DESC_SIZE_V1)
    return fail(GX_ERR_INVALID, "gx_vec_create: gx_desc.struct_size is %u, expected %zu (or %u, without the GX_ENV_TABLE fields)", desc->struct_size, sizeof(gx_desc), GX_DESC_SIZE_V1);
  if (desc->env_id < 0 || desc->env_id >= GX_ENV_COUNT) return fail(GX_ERR_INVALID, "gx_vec_create: unknown env id %d", desc->env_id);
  if (desc->env_id == GX_ENV_TABLE && desc->struct_size != sizeof(gx_desc))
    return fail(GX_ERR_INVALID, "gx_vec_create: GX_ENV_TABLE needs the whole gx_desc (struct_size %zu)", sizeof(gx_desc));
  if (desc->env_id == GX_ENV_TABLE && (!desc->table || desc->table_len == 0 || desc->table_len > (1ull << 26)))
    return fail(GX_ERR_INVALID, "gx_vec_create: GX_ENV_TABLE needs gx_desc.table with 1..2^26 entries");
  if (desc->num_envs == 0) return fail(GX_ERR_INVALID, "gx_vec_create: num_envs must be positive");
  if (desc->num_envs > (1ull << 31)) return fail(GX_ERR_UNSUPPORTED, "gx_vec_create: more than 2^31 envs per batch");
  if (desc->mode != GX_MODE_EXACT && desc->mode != GX_MODE_FAST32 && desc->mode != GX_MODE_FP64) return fail(GX_ERR_INVALID, "gx_vec_create: unknown mode %d", desc->mode);
  if (desc->time_limit < 0) return fail(GX_ERR_INVALID, "gx_vec_create: negative time_limit");
  const uint32_t known = GX_AUTORESET | GX_RECORD_STATS | GX_SLIPPERY | GX_TAXI_FULL | GX_ACROBOT_RK4 | GX_BLACKJACK_FULL;
  if (desc->flags & ~known) return fail(GX_ERR_INVALID, "gx_vec_create: unknown flag bits 0x%x", desc->flags & ~known);
  HIP_TRY(hipSetDevice(ctx->device));

  gx_vec* v = new (std::nothrow) gx_vec();
  if (!v) return fail(GX_ERR_OOM, "gx_vec_create: host allocation failed");
  v->ctx = ctx;
  ctx->live_handles++;
  std::memcpy(&v->desc, desc, desc->struct_size);  // a first-version struct leaves the table fields zero
  v->desc.struct_size = sizeof(gx_desc);
  if (desc->map_rows) v->map = desc->map_rows;
  v->desc.map_rows = nullptr;
  if (desc->env_id == GX_ENV_TABLE) v->user_table.assign(desc->table, desc->table + desc->table_len);
  v->desc.table = nullptr;  // the caller's array is not kept
  v->n = desc->num_envs;
  const int env = desc->env_id;
  v->is_acrobot64 = env == GX_ENV_ACROBOT && (desc->flags & GX_ACROBOT_RK4);
  v->is_classic = env <= GX_ENV_ACROBOT;
  v->is_classic64 = v->is_classic && !v->is_acrobot64 && desc->mode == GX_MODE_FP64;
  v->acrobot_angles = env == GX_ENV_ACROBOT && !v->is_acrobot64 && desc->mode == GX_MODE_EXACT;
  if (const char* s = std::getenv("GYMX_ACROBOT_ANGLES")) {  // tuning aid: 0 = the six-row form
    if (std::atoi(s) == 0) v->acrobot_angles = false;
  }
  v->classic_id = v->acrobot_angles ? kEnvAcrobotAngles : env;
  v->is_blackjack = env == GX_ENV_BLACKJACK;
  v->is_tabular = env == GX_ENV_FROZENLAKE || env == GX_ENV_TAXI || env == GX_ENV_CLIFFWALKING || env == GX_ENV_DUMMY || v->is_blackjack || env == GX_ENV_TABLE;
  v->is_frame = env == GX_ENV_PONG || env == GX_ENV_BREAKOUT || env == GX_ENV_SPACEINVADERS;
  v->envs_per_lane = (v->is_acrobot64 || v->is_classic64) ? 2 : (v->is_frame ? 1 : 4);
  v->stride = (v->n + 3) & ~3ull;  // rows stay 16-byte aligned for every element width
  // Classic control: rows of a large batch that would lie an exact multiple of 64 KiB apart (16 Mi envs: 64 MiB)
  // are moved off that power of two by 256 elements (measured at 16 Mi envs: CartPole 0.144 -> 0.140 ms, MountainCar
  // 0.0884 -> 0.0863, the others 0.8-1.7 % faster; any pad from 256 to 768 elements does, 64 are slower than none).
  // The table-driven envs gain nothing from it (-1.2 to +0.8 %) and keep their rows as they are.
  // GYMX_ROW_PAD (elements) overrides for every env: a tuning aid.
  if (const char* s = std::getenv("GYMX_ROW_PAD")) v->stride += (uint64_t)std::atoll(s) & ~3ull;
  else if (v->is_classic && v->stride >= (1u << 16) && (v->stride & 0x3FFFu) == 0) {
    v->stride += 256;
    v->row_skew = 1024;  // and the separately allocated row arrays continue the 1 KiB staircase (see rows_alloc below)
  } else if (v->is_tabular && !v->is_blackjack && v->stride >= (1u << 16) && (v->stride & 0x3FFFu) == 0) {
    // the table-driven envs keep their stride (see above) but start their row arrays 4 KiB apart: full Taxi-v3, which
    // has a return row more than the others, steps 64 Mi envs in 0.343 instead of 0.357 ms; the others -0.1 to -0.3 %
    v->row_skew = 4096;
  }
  v->ep_return_f64 = env == GX_ENV_MOUNTAINCAR_CONT || env == GX_ENV_PENDULUM;
  fill_info(v);

#define VEC_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      gx_vec_destroy(v);                                                                \
      return fail(e_ == hipErrorOutOfMemory ? GX_ERR_OOM : GX_ERR_HIP, "%s failed: %s", \
                  #expr, hipGetErrorString(e_));                                        \
    }                                                                                   \
  } while (0)

  std::vector<uint32_t> table;
  if (v->is_tabular && env != GX_ENV_DUMMY && !v->is_blackjack) {
    int rc = build_table(v, table);
    if (rc != GX_OK) { gx_vec_destroy(v); return rc; }
    v->user_table.clear();
    v->user_table.shrink_to_fit();
    table.resize((table.size() + 3) & ~(size_t)3, 0);  // the kernels stage it in 16-byte pieces
    VEC_TRY(hipMalloc((void**)&v->d_table, table.size() * 4));
    VEC_TRY(hipMemcpy(v->d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    // stage in LDS when it fits in 64 KiB, else the kernel reads the table through L2
    v->table_lds_bytes = table.size() * 4 <= 65536 ? table.size() * 4 : 0;
  }

  const gx_info& in = v->info;
  const size_t rsz = dtype_size(in.real_dtype);
  const size_t real_rows = (size_t)in.n_real + (v->acrobot_angles ? 2 : 0);
  // The row arrays of a batch are separate allocations, which the allocator aligns alike: env group q of every one
  // of them would sit at the same offset of a large power of two.  Where the rows of one array are padded apart
  // (above) each array also starts a different number of row_skew bytes into its allocation, continuing the count
  // of the rows before it: worth another 0.2-1 % (CartPole 0.1397 -> 0.1393 ms, MountainCar 0.0862 -> 0.0854,
  // Pendulum 0.1755 -> 0.1737).  GYMX_ROW_SKEW (bytes) overrides: a tuning aid.
  if (const char* s = std::getenv("GYMX_ROW_SKEW")) v->row_skew = (size_t)std::atoll(s) & ~(size_t)255;
  auto rows_alloc = [v](void** out, size_t bytes, int rows) -> hipError_t {
    const size_t off = (size_t)v->row_lane * v->row_skew;
    void* raw = nullptr;
    hipError_t e = hipMalloc(&raw, bytes + off);
    if (e != hipSuccess) return e;
    v->row_allocs.push_back(raw);
    *out = static_cast<char*>(raw) + off;
    v->row_lane += rows;
    return hipSuccess;
  };
  if (in.n_real) VEC_TRY(rows_alloc(&v->d_real, real_rows * v->stride * rsz, (int)real_rows));
  VEC_TRY(rows_alloc((void**)&v->d_int, (size_t)in.n_int * v->stride * 4, in.n_int));
  VEC_TRY(rows_alloc(&v->d_reward, v->stride * dtype_size(in.reward_dtype), 1));
  VEC_TRY(rows_alloc((void**)&v->d_done, v->stride, 1));
  if (desc->time_limit > 0) VEC_TRY(rows_alloc((void**)&v->d_trunc, v->stride, 1));
  VEC_TRY(rows_alloc(&v->d_actions, v->stride * 4, 1));
  VEC_TRY(hipEventCreateWithFlags(&v->ev_foreign, hipEventDisableTiming));
  VEC_TRY(hipMalloc((void**)&v->d_error, 256));
  VEC_TRY(hipMemset(v->d_error, 0, 256));
  VEC_TRY(hipHostMalloc((void**)&v->h_error, 64, hipHostMallocDefault));
  VEC_TRY(hipHostMalloc((void**)&v->h_stat, sizeof(StatSlot), hipHostMallocDefault));
  *v->h_error = 0;
  if (in.n_real) VEC_TRY(hipMemset(v->d_real, 0, real_rows * v->stride * rsz));
  VEC_TRY(hipMemset(v->d_int, 0, (size_t)in.n_int * v->stride * 4));
  VEC_TRY(hipMemset(v->d_reward, 0, v->stride * dtype_size(in.reward_dtype)));
  VEC_TRY(hipMemset(v->d_done, 0, v->stride));
  if (v->d_trunc) VEC_TRY(hipMemset(v->d_trunc, 0, v->stride));
  VEC_TRY(hipMemset(v->d_actions, 0, v->stride * 4));
  if (v->is_acrobot64) VEC_TRY(hipMalloc((void**)&v->d_obs, (size_t)6 * v->stride * 4));
  if (!v->is_frame) {
    // staging for the host path while a step's traffic is small enough that latency, not PCIe
    // bandwidth, sets its cost
    size_t limit = kStageLimitBytes;
    if (const char* s = std::getenv("GYMX_STAGE_LIMIT")) limit = (size_t)std::atoll(s);  // tuning aid; 0 disables
    auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t esz = dtype_size(in.obs_dtype);
    v->stage_reward = up16(v->stride * 4);
    v->stage_done = v->stage_reward + up16(v->n * dtype_size(in.reward_dtype));
    v->stage_trunc = v->stage_done + up16(v->n);
    v->stage_obs = v->stage_trunc + up16(v->n);
    v->stage_obs_pitch = up16(v->n * esz);
    const size_t total = v->stage_obs + (size_t)in.obs_dim * v->stage_obs_pitch;
    if (total <= limit) {
      VEC_TRY(hipHostMalloc((void**)&v->h_stage, total, hipHostMallocMapped));
      VEC_TRY(hipHostGetDevicePointer((void**)&v->d_stage, v->h_stage, 0));
      std::memset(v->h_stage, 0, total);
    }
  }
  if (v->is_frame) VEC_TRY(hipMalloc((void**)&v->d_frames, v->n * (size_t)in.obs_dim));
  if (env == GX_ENV_SPACEINVADERS) VEC_TRY(hipMalloc((void**)&v->d_render, v->n * kInvadersRecordBytes));
  if (env == GX_ENV_BREAKOUT) VEC_TRY(hipMalloc((void**)&v->d_render, v->n * kBreakoutRecordBytes));
  if (env == GX_ENV_PONG) VEC_TRY(hipMalloc((void**)&v->d_render, v->n * kPongRecordBytes));

  // launch geometry: workgroups grid-stride over lanes of `envs_per_lane` envs, two trips
  // each.  Measured on CartPole, MountainCar, Pendulum and FrozenLake at 4 Mi to 64 Mi envs:
  // half as many workgroups as lane groups beats both one trip each and a fixed 8 x CUs grid
  // by 4-10 %.  Small batches keep at least four workgroups per CU.
  {
    uint64_t groups = (v->n + v->envs_per_lane - 1) / v->envs_per_lane;
    v->block = kBlock;
    if (v->is_classic) v->block = kClassicBlock;
    if (v->is_tabular && !v->is_blackjack) {
      v->block = table.size() * 4 <= kSmallTableBytes ? kTabularBlockSmallTable : kTabularBlock;
      if (const char* s = std::getenv("GYMX_TABULAR_BLOCK")) {  // tuning aid: 64, 128 or 256
        const int b = std::atoi(s);
        if (b == 64 || b == 128 || b == 256) v->block = b;
      }
    }
    uint64_t blocks = (groups + v->block - 1) / v->block;
    uint64_t cap = std::max<uint64_t>((blocks + 1) / 2, (uint64_t)ctx->num_cus * 4);
    // a large table is staged in LDS once per workgroup: four trips amortise that better
    // (Taxi at 64 Mi envs: 0.322 -> 0.311 ms, full Taxi-v3 0.364 -> 0.358 ms)
    if (v->is_tabular && table.size() * 4 > kSmallTableBytes) cap = std::max<uint64_t>((blocks + 3) / 4, (uint64_t)ctx->num_cus * 4);
    // Classic control, large batches: a grid of r + 4/9 times the workgroups the chip holds at once (r whole, 32
    // one-wave workgroups per CU) instead of exactly half the workgroups -- not a whole number of rounds, and a second
    // trip that starts a little past the middle of the batch.  Measured on CartPole with device actions: 8 / 12 / 16 /
    // 20 / 32 / 64 Mi envs -0.3 / -1.1 / -4.0 / -0.6 / +0.3 / -1.3 % (16 Mi: 0.139 -> 0.1335 ms; the minimum is broad,
    // 4.4 to 5 rounds); the other classic-control envs at 16 Mi envs gain 0-2.4 %.
    // The fused rollouts, which are arithmetic from end to end, keep the even split (their trips then take the same time).
    const uint64_t resident = (uint64_t)ctx->num_cus * 32;
    uint64_t cap_single = 0;
    if (v->is_classic && cap >= 2 * resident && cap < blocks) cap_single = (cap / resident) * resident + resident * 4 / 9;
    if (const char* s = std::getenv("GYMX_GRID_BLOCKS")) {
      long long b = std::atoll(s);
      if (b > 0) { cap = (uint64_t)b; cap_single = 0; }
    }
    v->grid = dim3((unsigned)std::max<uint64_t>(1, std::min(blocks, cap)));
    if (cap_single) v->grid_single = (unsigned)std::min(blocks, cap_single);
  }
  if (v->is_classic) {
    const uint32_t both = GX_AUTORESET | GX_RECORD_STATS;
    const bool constant_reward = env == GX_ENV_CARTPOLE || env == GX_ENV_MOUNTAINCAR || env == GX_ENV_ACROBOT;
    v->derive_returns = constant_reward && (desc->flags & both) == both && desc->time_limit == 0;
    v->ret_per_step = env == GX_ENV_CARTPOLE ? 1 : -1;  // Envs.CartPole.cs:72, Envs.MountainCar.cs:45, Envs.Acrobot.cs:88
  }
  if (desc->flags & GX_RECORD_STATS) {
    // one slot per wave of the step grid, and never fewer than one kBlock workgroup has waves
    v->num_slots = std::max((int)std::max(v->grid.x, v->grid_single) * (v->block / 64), kBlock / 64);
    VEC_TRY(hipMalloc((void**)&v->d_slots, ((size_t)v->num_slots + 1) * sizeof(StatSlot)));
    VEC_TRY(launch_stats_init(v->d_slots, v->num_slots + 1, ctx->compute));
    const size_t esz = v->ep_return_f64 ? 8 : 4;
    VEC_TRY(rows_alloc(&v->d_ep_return, v->stride * esz, 1));
    VEC_TRY(hipMemset(v->d_ep_return, 0, v->stride * esz));
    if (!(desc->flags & GX_AUTORESET)) {
      // without auto-reset the env's own step counter keeps running past done
      // while RecordEpisodeStatistics restarts its length at 0
      VEC_TRY(rows_alloc((void**)&v->d_ep_length, v->stride * 4, 1));
      VEC_TRY(hipMemset(v->d_ep_length, 0, v->stride * 4));
    }
  }
  if (v->is_frame) {
    // the Pong, Breakout and SpaceInvaders constructors call Reset() (Envs.Pong.cs:47-52,
    // Envs.Breakout.cs:54-60, Envs.SpaceInvaders.cs:59-67); tick stays 0
    StepArgs a = make_args(v, nullptr, 1, 0);
    VEC_TRY(frame_reset(v, a, ctx->compute));
    VEC_TRY(frame_raster(v, a, ctx->compute));
    v->frames_valid = true;
  } else if (v->is_tabular && env != GX_ENV_DUMMY && env != GX_ENV_TAXI) {
    // field initialisers: FrozenLake `_state` = 0, CliffWalking `_state` = 0 until Reset
  }
  if (v->is_acrobot64) {
    StepArgs a = make_args(v, nullptr, 1, 0);
    VEC_TRY(launch_acrobot64_reset(a, 1, v->grid, ctx->compute));
  }
  VEC_TRY(hipStreamSynchronize(ctx->compute));
#undef VEC_TRY
  *out = v;
  return GX_OK;
}

extern "C" int gx_vec_info(const gx_vec* v, gx_info* out) {
  if (!v || !out) return fail(GX_ERR_INVALID, "gx_vec_info: null argument");
  *out = v->info;
  return GX_OK;
}

extern "C" int gx_vec_tick(const gx_vec* v, uint64_t* tick) {
  if (!v || !tick) return fail(GX_ERR_INVALID, "gx_vec_tick: null argument");
  *tick = v->tick;
  return GX_OK;
}

extern "C" int gx_vec_reset(gx_vec* v) {
  if (!v) return fail(GX_ERR_INVALID, "gx_vec_reset: null batch");
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  StepArgs a = make_args(v, nullptr, 1, 0);
  if (v->desc.flags & GX_RECORD_STATS) {
    // EnvWrapper.cs:34-38; with auto-reset the episode length is the step counter
    const int32_t* len_row = v->d_ep_length ? v->d_ep_length : v->d_int + (size_t)(v->info.n_int - 1) * v->stride;
    uint64_t blocks = (v->n + kBlock - 1) / kBlock;
    dim3 grid((unsigned)std::min<uint64_t>(blocks, (uint64_t)v->num_slots / (kBlock / 64)));  // the flush kernel runs kBlock threads
    HIP_TRY(launch_stats_flush_open(a, v->ep_return_f64, len_row, grid, c->compute));
  }
  hipError_t e;
  if (v->is_acrobot64) e = launch_acrobot64_reset(a, 0, v->grid, c->compute);
  else if (v->is_classic64) e = launch_classic64_reset(v->desc.env_id, a, v->grid, c->compute);
  else if (v->is_classic) e = launch_classic_reset(v->classic_id, a, v->grid, c->compute);
  else if (v->is_blackjack) e = launch_blackjack_reset(a, v->grid, c->compute);
  else if (v->is_tabular) e = launch_tabular_reset(v->desc.env_id, a, v->grid, c->compute);
  else {
    e = frame_reset(v, a, c->compute);
    if (e == hipSuccess) e = frame_raster(v, a, c->compute);
  }
  if (e != hipSuccess) return fail(GX_ERR_HIP, "reset kernel launch failed: %s", hipGetErrorString(e));
  v->forced_pending = false;
  v->tick += 1;
  HIP_TRY(hipStreamSynchronize(c->compute));
  return GX_OK;
}

extern "C" int gx_vec_get_state(gx_vec* v, void* host, size_t bytes) {
  if (!v || !host) return fail(GX_ERR_INVALID, "gx_vec_get_state: null argument");
  if (bytes != v->info.state_bytes) return fail(GX_ERR_INVALID, "gx_vec_get_state: buffer is %zu bytes, state image is %llu", bytes, (unsigned long long)v->info.state_bytes);
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->compute));
  const size_t rsz = dtype_size(v->info.real_dtype);
  char* h = static_cast<char*>(host);
  int rc = copy_rows(h, v->d_real, rsz, v->info.n_real, v->n, v->n, v->stride, hipMemcpyDeviceToHost, c->copy);
  if (rc) return rc;
  rc = copy_rows(h + (size_t)v->info.n_real * v->n * rsz, v->d_int, 4, v->info.n_int, v->n, v->n, v->stride, hipMemcpyDeviceToHost, c->copy);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->copy));
  return GX_OK;
}

extern "C" int gx_vec_set_state(gx_vec* v, const void* host, size_t bytes) {
  if (!v || !host) return fail(GX_ERR_INVALID, "gx_vec_set_state: null argument");
  if (bytes != v->info.state_bytes) return fail(GX_ERR_INVALID, "gx_vec_set_state: buffer is %zu bytes, state image is %llu", bytes, (unsigned long long)v->info.state_bytes);
  if (v->is_acrobot64) {
    // This env keeps its angles wrapped to [-pi, pi] and evaluates sin/cos with a reduction that
    // is only valid for moderate arguments: an image must hold finite values and angles within
    // 1024 rad (the first step wraps them).
    const double* h = static_cast<const double*>(host);
    for (uint64_t i = 0; i < 4 * v->n; i++) {
      const bool angle = i < 2 * v->n;
      if (!std::isfinite(h[i]) || (angle && std::fabs(h[i]) > 1024.0))
        return fail(GX_ERR_INVALID, "gx_vec_set_state: component %llu of env %llu is %g; the RK4 Acrobot takes finite values and angles within 1024 rad",
                    (unsigned long long)(i / v->n), (unsigned long long)(i % v->n), h[i]);
    }
  }
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->compute));
  if (v->derive_returns) {
    // the caller may set any step counter: from here on the return row is kept explicitly,
    // starting from what the current counters imply
    StepArgs a = make_args(v, nullptr, 1, 0);
    uint64_t blocks = (v->n + kBlock - 1) / kBlock;
    dim3 grid((unsigned)std::min<uint64_t>(blocks, (uint64_t)c->num_cus * 8));
    HIP_TRY(launch_stats_materialize_returns(a, v->d_int + (size_t)(v->info.n_int - 1) * v->stride, grid, c->compute));
    HIP_TRY(hipStreamSynchronize(c->compute));
    v->derive_returns = false;
  }
  const size_t rsz = dtype_size(v->info.real_dtype);
  const char* h = static_cast<const char*>(host);
  int rc = copy_rows(v->d_real, h, rsz, v->info.n_real, v->n, v->stride, v->n, hipMemcpyHostToDevice, c->copy);
  if (rc) return rc;
  rc = copy_rows(v->d_int, h + (size_t)v->info.n_real * v->n * rsz, 4, v->info.n_int, v->n, v->stride, v->n, hipMemcpyHostToDevice, c->copy);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->copy));
  v->state_dirty = true;
  StepArgs a = make_args(v, nullptr, 1, 0);
  if (v->is_frame) {
    HIP_TRY(frame_pack(v, a, c->compute));
    HIP_TRY(frame_raster(v, a, c->compute));
  }
  if (v->is_acrobot64) HIP_TRY(launch_acrobot64_reset(a, 1, v->grid, c->compute));
  if (v->acrobot_angles) HIP_TRY(launch_classic_derive(v->classic_id, a, v->grid, c->compute));
  HIP_TRY(hipStreamSynchronize(c->compute));
  return GX_OK;
}

extern "C" int gx_vec_set_rng_draws(gx_vec* v, const double* u, size_t count) {
  if (!v) return fail(GX_ERR_INVALID, "gx_vec_set_rng_draws: null batch");
  if (v->desc.env_id != GX_ENV_PONG && v->desc.env_id != GX_ENV_BREAKOUT)
    return fail(GX_ERR_UNSUPPORTED, "gx_vec_set_rng_draws: env id %d; Pong (ball reset) and Breakout (ball release) take their draws from the caller -- replay the other envs with gx_vec_set_state", v->desc.env_id);
  const size_t per_env = v->desc.env_id == GX_ENV_PONG ? 2 : 1;
  if (!u || count == 0) {  // withdraw draws that were set and not used yet
    v->forced_pending = false;
    return GX_OK;
  }
  if (count != per_env * (size_t)v->n)
    return fail(GX_ERR_INVALID, "gx_vec_set_rng_draws: %zu values, expected %zu x %llu ([%zu][num_envs])", count, per_env, (unsigned long long)v->n, per_env);
  for (size_t i = 0; i < count; i++)
    if (!(u[i] >= 0.0 && u[i] < 1.0))  // Random.NextDouble() is in [0, 1); NaN fails the test too
      return fail(GX_ERR_INVALID, "gx_vec_set_rng_draws: value %zu is %g, outside [0, 1)", i, u[i]);
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  if (!v->d_forced) {
    hipError_t e = hipMalloc((void**)&v->d_forced, (size_t)2 * v->stride * sizeof(double));
    if (e != hipSuccess) {
      v->d_forced = nullptr;
      return fail(e == hipErrorOutOfMemory ? GX_ERR_OOM : GX_ERR_HIP, "gx_vec_set_rng_draws: hipMalloc failed: %s", hipGetErrorString(e));
    }
  }
  HIP_TRY(hipStreamSynchronize(c->compute));  // a launch that still reads the previous draws
  int rc = wait_foreign(v);
  if (rc) return rc;
  rc = copy_rows(v->d_forced, u, sizeof(double), (int)per_env, v->n, v->stride, v->n, hipMemcpyHostToDevice, c->copy);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->copy));
  v->forced_pending = true;
  return GX_OK;
}

// The observation rows as the device holds them: (first row, bytes between rows).
static const char* obs_rows(const gx_vec* v, size_t* pitch) {
  const size_t esz = dtype_size(v->info.obs_dtype);
  *pitch = v->stride * esz;
  if (v->is_tabular) return reinterpret_cast<const char*>(v->d_int);
  return v->is_acrobot64 ? reinterpret_cast<const char*>(v->d_obs) : static_cast<const char*>(v->d_real);
}

// Staged host path: hands the results of the pending step to the caller's buffers.
static int finish_staged(gx_vec* v) {
  HIP_TRY(hipStreamSynchronize(v->ctx->compute));
  const gx_info& in = v->info;
  if (v->pend_reward) std::memcpy(v->pend_reward, v->h_stage + v->stage_reward, v->n * dtype_size(in.reward_dtype));
  if (v->pend_done) std::memcpy(v->pend_done, v->h_stage + v->stage_done, v->n);
  if (v->pend_trunc) std::memcpy(v->pend_trunc, v->h_stage + v->stage_trunc, v->n);
  if (v->pend_obs) {
    const size_t row = v->n * dtype_size(in.obs_dtype);
    for (int c = 0; c < in.obs_dim; c++)
      std::memcpy(static_cast<char*>(v->pend_obs) + (size_t)c * row, v->h_stage + v->stage_obs + (size_t)c * v->stage_obs_pitch, row);
  }
  v->stage_pending = false;
  return GX_OK;
}

// enqueue the device->host copy of the observation on `stream`
static int copy_obs(gx_vec* v, void* host, hipStream_t stream) {
  const gx_info& in = v->info;
  if (v->is_frame) {
    HIP_TRY(hipMemcpyAsync(host, v->d_frames, in.obs_bytes, hipMemcpyDeviceToHost, stream));
    return GX_OK;
  }
  if (v->is_tabular) {
    HIP_TRY(hipMemcpyAsync(host, v->d_int, v->n * 4, hipMemcpyDeviceToHost, stream));
    return GX_OK;
  }
  const void* src = v->is_acrobot64 ? (const void*)v->d_obs : (const void*)v->d_real;
  return copy_rows(host, src, dtype_size(in.obs_dtype), in.obs_dim, v->n, v->n, v->stride, hipMemcpyDeviceToHost, stream);
}

extern "C" int gx_vec_get_obs(gx_vec* v, void* host, size_t bytes) {
  if (!v || !host) return fail(GX_ERR_INVALID, "gx_vec_get_obs: null argument");
  if (bytes != v->info.obs_bytes) return fail(GX_ERR_INVALID, "gx_vec_get_obs: buffer is %zu bytes, observation is %llu", bytes, (unsigned long long)v->info.obs_bytes);
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->compute));
  int rc = copy_obs(v, host, c->copy);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->copy));
  return GX_OK;
}

extern "C" int gx_vec_step_async(gx_vec* v, const void* h_actions, void* h_obs, void* h_reward, uint8_t* h_done,
                                 uint8_t* h_truncated) {
  if (!v || !h_actions) return fail(GX_ERR_INVALID, "gx_vec_step_async: null argument");
  if (h_truncated && !v->d_trunc) return fail(GX_ERR_INVALID, "gx_vec_step_async: truncated requested but the batch has no time limit");
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  if (v->desc.env_id == GX_ENV_ACROBOT) {
    // torque[action] throws IndexOutOfRangeException (Envs.Acrobot.cs:45,51);
    // the whole call is rejected before any env is stepped
    const int32_t* act = static_cast<const int32_t*>(h_actions);
    uint32_t bad = 0;  // branch-free scan (vectorises); the offender is looked up only if there is one
    for (uint64_t i = 0; i < v->n; i++) bad |= (uint32_t)((uint32_t)act[i] > 2u);
    if (bad)
      for (uint64_t i = 0; i < v->n; i++)
        if ((uint32_t)act[i] > 2u) return fail(GX_ERR_INVALID, "Acrobot: action %d of env %llu is outside {0,1,2}", act[i], (unsigned long long)i);
  }
  if (v->desc.env_id == GX_ENV_TABLE) {
    // an action outside a caller-supplied table's space has no transition: rejected like Acrobot's
    const int32_t* act = static_cast<const int32_t*>(h_actions);
    const uint32_t na = (uint32_t)v->table_actions;
    uint32_t bad = 0;
    for (uint64_t i = 0; i < v->n; i++) bad |= (uint32_t)((uint32_t)act[i] >= na);
    if (bad)
      for (uint64_t i = 0; i < v->n; i++)
        if ((uint32_t)act[i] >= na) return fail(GX_ERR_INVALID, "table env: action %d of env %llu is outside Discrete(%u)", act[i], (unsigned long long)i, na);
  }
  if (v->h_stage) {
    if (v->stage_pending) {  // a step that was never waited for: its results go to its buffers first
      int rc = finish_staged(v);
      if (rc) return rc;
    }
    std::memcpy(v->h_stage, h_actions, v->n * 4);
    int rc = launch_step(v, v->d_stage, 1, v->stride, GX_POLICY_DEVICE_ACTIONS, c->compute);
    if (rc) return rc;
    GatherArgs g{};
    auto add = [&g](const void* src, size_t offset, size_t bytes, char* base) { g.seg[g.count++] = GatherSegment{src, base + offset, bytes}; };
    if (h_reward) add(v->d_reward, v->stage_reward, v->n * dtype_size(v->info.reward_dtype), v->d_stage);
    if (h_done) add(v->d_done, v->stage_done, v->n, v->d_stage);
    if (h_truncated) add(v->d_trunc, v->stage_trunc, v->n, v->d_stage);
    if (h_obs) {
      size_t pitch = 0;
      const char* rows = obs_rows(v, &pitch);
      for (int c = 0; c < v->info.obs_dim; c++)
        add(rows + (size_t)c * pitch, v->stage_obs + (size_t)c * v->stage_obs_pitch, v->n * dtype_size(v->info.obs_dtype), v->d_stage);
    }
    if (g.count) {
      const uint64_t blocks = (v->n * 8 / 16 + kBlock - 1) / kBlock;  // the widest row: 8 bytes per env
      HIP_TRY(launch_gather_segments(g, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(blocks, 256))), c->compute));
    }
    v->pend_obs = h_obs; v->pend_reward = h_reward; v->pend_done = h_done; v->pend_trunc = h_truncated;
    v->stage_pending = true;
    return GX_OK;
  }
  HIP_TRY(hipMemcpyAsync(v->d_actions, h_actions, v->n * 4, hipMemcpyHostToDevice, c->copy));
  HIP_TRY(hipEventRecord(c->ev_in, c->copy));
  HIP_TRY(hipStreamWaitEvent(c->compute, c->ev_in, 0));
  int rc = launch_step(v, v->d_actions, 1, v->stride, GX_POLICY_DEVICE_ACTIONS, c->compute);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(c->ev_out, c->compute));
  HIP_TRY(hipStreamWaitEvent(c->copy, c->ev_out, 0));
  if (h_obs) {
    rc = copy_obs(v, h_obs, c->copy);
    if (rc) return rc;
  }
  if (h_reward) HIP_TRY(hipMemcpyAsync(h_reward, v->d_reward, v->n * dtype_size(v->info.reward_dtype), hipMemcpyDeviceToHost, c->copy));
  if (h_done) HIP_TRY(hipMemcpyAsync(h_done, v->d_done, v->n, hipMemcpyDeviceToHost, c->copy));
  if (h_truncated) HIP_TRY(hipMemcpyAsync(h_truncated, v->d_trunc, v->n, hipMemcpyDeviceToHost, c->copy));
  return GX_OK;
}

extern "C" int gx_vec_wait(gx_vec* v) {
  if (!v) return fail(GX_ERR_INVALID, "gx_vec_wait: null batch");
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  if (v->stage_pending) {
    int rc = finish_staged(v);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(c->compute));
  // steps enqueued on a caller's stream may still be running: the flag is read (and cleared) after them
  if (int rc = wait_foreign(v)) return rc;
  if (has_error_flag(v)) HIP_TRY(hipMemcpyAsync(v->h_error, v->d_error, 4, hipMemcpyDeviceToHost, c->copy));
  HIP_TRY(hipStreamSynchronize(c->copy));
  if (has_error_flag(v) && *v->h_error) {
    *v->h_error = 0;
    HIP_TRY(hipMemsetAsync(v->d_error, 0, 4, c->copy));
    HIP_TRY(hipStreamSynchronize(c->copy));
    return fail(GX_ERR_INVALID, "a step since the last wait received an action outside the action space where the reference throws");
  }
  return GX_OK;
}

extern "C" int gx_vec_step_device(gx_vec* v, const void* d_actions, void* stream) {
  if (!v || !d_actions) return fail(GX_ERR_INVALID, "gx_vec_step_device: null argument");
  if ((uintptr_t)d_actions & 15) return fail(GX_ERR_INVALID, "gx_vec_step_device: d_actions must be 16-byte aligned");
  HIP_TRY(hipSetDevice(v->ctx->device));
  return launch_step(v, d_actions, 1, v->stride, GX_POLICY_DEVICE_ACTIONS, stream ? (hipStream_t)stream : v->ctx->compute);
}

extern "C" int gx_vec_step_random(gx_vec* v, void* stream) {
  if (!v) return fail(GX_ERR_INVALID, "gx_vec_step_random: null batch");
  HIP_TRY(hipSetDevice(v->ctx->device));
  return launch_step(v, nullptr, 1, 0, GX_POLICY_RANDOM, stream ? (hipStream_t)stream : v->ctx->compute);
}

extern "C" int gx_vec_rollout(gx_vec* v, int k_steps, int policy, const void* d_actions, void* stream) {
  if (!v) return fail(GX_ERR_INVALID, "gx_vec_rollout: null batch");
  if (k_steps <= 0) return fail(GX_ERR_INVALID, "gx_vec_rollout: k_steps must be positive");
  if (policy != GX_POLICY_RANDOM && policy != GX_POLICY_DEVICE_ACTIONS) return fail(GX_ERR_INVALID, "gx_vec_rollout: unknown policy %d", policy);
  if (policy == GX_POLICY_DEVICE_ACTIONS) {
    if (!d_actions) return fail(GX_ERR_INVALID, "gx_vec_rollout: DEVICE_ACTIONS needs d_actions[k_steps][stride]");
    if ((uintptr_t)d_actions & 15) return fail(GX_ERR_INVALID, "gx_vec_rollout: d_actions must be 16-byte aligned");
  }
  HIP_TRY(hipSetDevice(v->ctx->device));
  return launch_step(v, d_actions, k_steps, v->stride, policy, stream ? (hipStream_t)stream : v->ctx->compute);
}

extern "C" int gx_vec_rollout_out(gx_vec* v, int k_steps, int policy, const void* d_actions, int out, void* d_reward_rows,
                                  uint8_t* d_done_rows, uint8_t* d_truncated_rows, void* stream) {
  if (!v) return fail(GX_ERR_INVALID, "gx_vec_rollout_out: null batch");
  if (k_steps <= 0) return fail(GX_ERR_INVALID, "gx_vec_rollout_out: k_steps must be positive");
  if (policy != GX_POLICY_RANDOM && policy != GX_POLICY_DEVICE_ACTIONS) return fail(GX_ERR_INVALID, "gx_vec_rollout_out: unknown policy %d", policy);
  if (policy == GX_POLICY_DEVICE_ACTIONS) {
    if (!d_actions) return fail(GX_ERR_INVALID, "gx_vec_rollout_out: DEVICE_ACTIONS needs d_actions[k_steps][stride]");
    if ((uintptr_t)d_actions & 15) return fail(GX_ERR_INVALID, "gx_vec_rollout_out: d_actions must be 16-byte aligned");
  }
  StepOut o;
  switch (out) {
    case GX_OUT_LAST: break;
    case GX_OUT_NONE: o.skip_last = true; break;
    case GX_OUT_ALL:
      if (!d_reward_rows || !d_done_rows) return fail(GX_ERR_INVALID, "gx_vec_rollout_out: GX_OUT_ALL needs reward and done rows [k_steps][stride]");
      if (d_truncated_rows && !v->d_trunc) return fail(GX_ERR_INVALID, "gx_vec_rollout_out: truncated rows requested but the batch has no time limit");
      if (((uintptr_t)d_reward_rows | (uintptr_t)d_done_rows | (uintptr_t)d_truncated_rows) & 15)
        return fail(GX_ERR_INVALID, "gx_vec_rollout_out: output rows must be 16-byte aligned");
      o.reward = d_reward_rows; o.done = d_done_rows; o.truncated = d_truncated_rows;
      break;
    default: return fail(GX_ERR_INVALID, "gx_vec_rollout_out: unknown output mode %d", out);
  }
  HIP_TRY(hipSetDevice(v->ctx->device));
  return launch_step(v, d_actions, k_steps, v->stride, policy, stream ? (hipStream_t)stream : v->ctx->compute, &o);
}

extern "C" int gx_vec_sample_actions(gx_vec* v, uint64_t tick, void* d_actions, void* stream) {
  if (!v || !d_actions) return fail(GX_ERR_INVALID, "gx_vec_sample_actions: null argument");
  HIP_TRY(hipSetDevice(v->ctx->device));
  return launch_sample(v, tick, d_actions, stream ? (hipStream_t)stream : v->ctx->compute);
}

extern "C" int gx_vec_device_view(gx_vec* v, gx_view* out) {
  if (!v || !out) return fail(GX_ERR_INVALID, "gx_vec_device_view: null argument");
  out->num_envs = v->n;
  out->stride = v->stride;
  out->state_real = v->d_real;
  out->state_int = v->d_int;
  out->obs = v->is_frame ? (void*)v->d_frames : (v->is_acrobot64 ? (void*)v->d_obs : (v->is_tabular ? (void*)v->d_int : v->d_real));
  out->reward = v->d_reward;
  out->done = v->d_done;
  out->truncated = v->d_trunc;
  return GX_OK;
}

static int reduce_stats(gx_vec* v, hipStream_t stream) {
  if (!(v->desc.flags & GX_RECORD_STATS)) return fail(GX_ERR_INVALID, "the batch was created without GX_RECORD_STATS");
  HIP_TRY(launch_stats_reduce(v->d_slots, v->num_slots, v->d_slots + v->num_slots, stream));
  return GX_OK;
}

extern "C" int gx_vec_stats(gx_vec* v, gx_stats* out) {
  if (!v || !out) return fail(GX_ERR_INVALID, "gx_vec_stats: null argument");
  gx_ctx* c = v->ctx;
  HIP_TRY(hipSetDevice(c->device));
  int rc = reduce_stats(v, c->compute);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(v->h_stat, v->d_slots + v->num_slots, sizeof(StatSlot), hipMemcpyDeviceToHost, c->compute));
  HIP_TRY(hipStreamSynchronize(c->compute));
  out->episodes = v->h_stat->episodes;
  out->sum_return = v->h_stat->sum_return;
  out->sum_length = v->h_stat->sum_length;
  out->sum_return_sq = v->h_stat->sum_return_sq;
  out->min_return = v->h_stat->min_return;
  out->max_return = v->h_stat->max_return;
  return GX_OK;
}

// ------------------------------------------------- the no-allocation entry ---
extern "C" int gx_rows_step(gx_ctx* c, const gx_rows* r, int k_steps, void* stream_) {
  if (!c || !r) return fail(GX_ERR_INVALID, "gx_rows_step: null argument");
  if (r->struct_size != sizeof(gx_rows)) return fail(GX_ERR_INVALID, "gx_rows_step: gx_rows.struct_size is %u, expected %zu", r->struct_size, sizeof(gx_rows));
  if (r->env_id < GX_ENV_CARTPOLE || r->env_id > GX_ENV_ACROBOT) return fail(GX_ERR_UNSUPPORTED, "gx_rows_step: env id %d is not one of the five classic-control envs", r->env_id);
  if (r->mode != GX_MODE_EXACT && r->mode != GX_MODE_FAST32) return fail(GX_ERR_UNSUPPORTED, "gx_rows_step: mode %d (binary32 modes only)", r->mode);
  if (r->num_envs == 0) return fail(GX_ERR_INVALID, "gx_rows_step: num_envs must be positive");
  if (r->num_envs > kRowsMaxEnvs)
    return fail(GX_ERR_INVALID, "gx_rows_step: %llu envs exceed the %llu the library's fixed scratch array serves", (unsigned long long)r->num_envs, (unsigned long long)kRowsMaxEnvs);
  if ((r->stride & 3) || r->stride < r->num_envs) return fail(GX_ERR_INVALID, "gx_rows_step: stride %llu must be a multiple of 4 and >= num_envs", (unsigned long long)r->stride);
  if (k_steps <= 0) return fail(GX_ERR_INVALID, "gx_rows_step: k_steps must be positive");
  const bool f64_return = r->env_id == GX_ENV_MOUNTAINCAR_CONT || r->env_id == GX_ENV_PENDULUM;
  if (!r->state_real || !r->steps || !r->reward || !r->done || !r->stats) return fail(GX_ERR_INVALID, "gx_rows_step: null row pointer");
  if (f64_return && !r->ep_return) return fail(GX_ERR_INVALID, "gx_rows_step: this env needs the ep_return row");
  // the kernel that tolerates unbounded angles keeps the return row explicitly (Acrobot re-derives its
  // angles with atan2 every step and has no such kernel)
  if (!r->ep_return && !r->state_bounded && r->env_id != GX_ENV_ACROBOT)
    return fail(GX_ERR_INVALID, "gx_rows_step: state_bounded = 0 needs an int32 ep_return row for this env");
  const void* ptrs[] = {r->state_real, r->steps, r->ep_return, r->actions, r->reward, r->done, r->stats};
  for (const void* p : ptrs)
    if ((uintptr_t)p & 15) return fail(GX_ERR_INVALID, "gx_rows_step: row pointers must be 16-byte aligned");
  if (!c->scratch) return fail(GX_ERR_HIP, "gx_rows_step: the context has no scratch address");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : c->compute;

  // the geometry gx_vec_create gives the same batch, within the slots the scratch array holds
  const uint64_t groups = (r->num_envs + 3) / 4;
  const uint64_t blocks = (groups + kClassicBlock - 1) / kClassicBlock;
  uint64_t cap = std::max<uint64_t>((blocks + 1) / 2, (uint64_t)c->num_cus * 4);
  const uint64_t resident = (uint64_t)c->num_cus * 32;
  if (k_steps == 1 && cap >= 2 * resident && cap < blocks) cap = (cap / resident) * resident + resident * 4 / 9;  // as gx_vec_create
  // the clamp is on statistics records, one per wave, not on workgroups
  const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(blocks, cap), (uint64_t)kRowsMaxSlots / (kClassicBlock / 64))));
  const int num_slots = std::max((int)grid.x * (kClassicBlock / 64), kBlock / 64);
  static_assert(kBlock / 64 <= kRowsMaxSlots, "the smallest grid's records fit the scratch array");
  if (num_slots > kRowsMaxSlots) return fail(GX_ERR_INVALID, "gx_rows_step: %d statistics records exceed the scratch array's %d", num_slots, kRowsMaxSlots);

  StatSlot* slots = reinterpret_cast<StatSlot*>(c->scratch);
  StatSlot* result = reinterpret_cast<StatSlot*>(c->scratch + kRowsResultOffset);
  uint32_t* err = reinterpret_cast<uint32_t*>(c->scratch + kRowsErrorOffset);
  // the array keeps the last call's contents: the records this call adds into start empty
  HIP_TRY(launch_stats_init(slots, num_slots, stream));
  HIP_TRY(hipMemsetAsync(err, 0, 4, stream));

  StepArgs a{};
  a.state_real = r->state_real;
  a.state_int = r->steps;
  a.actions = r->actions;
  a.reward = r->reward;
  a.done = r->done;
  a.ep_return = r->ep_return;
  a.stat_slots = slots;
  a.error_flag = err;
  a.num_envs = r->num_envs;
  a.stride = r->stride;
  a.action_stride = r->stride;
  a.env_index_base = r->env_index_base;
  a.tick = r->tick;
  a.seed_lo = (uint32_t)r->seed;
  a.seed_hi = (uint32_t)(r->seed >> 32);
  a.flags = GX_AUTORESET | GX_RECORD_STATS;
  a.k_steps = k_steps;
  a.reverse = (uint32_t)(r->tick & 1u);
  a.derive_returns = r->ep_return ? 0 : 1;
  a.ret_per_step = r->env_id == GX_ENV_CARTPOLE ? 1 : -1;
  const int policy = r->actions ? GX_POLICY_DEVICE_ACTIONS : GX_POLICY_RANDOM;
  HIP_TRY(launch_classic_step(r->env_id, a, r->mode, policy, /*fused=*/true, /*bounded=*/r->state_bounded != 0, grid, stream));
  HIP_TRY(launch_stats_reduce(slots, num_slots, result, stream));
  // StatSlot begins with the six fields of gx_stats in the same order
  static_assert(sizeof(gx_stats) == 48 && sizeof(gx_stats) <= sizeof(StatSlot), "gx_stats is the head of StatSlot");
  HIP_TRY(hipMemcpyAsync(r->stats, result, sizeof(gx_stats), hipMemcpyDeviceToDevice, stream));
  if (r->error) HIP_TRY(hipMemcpyAsync(r->error, err, 4, hipMemcpyDeviceToDevice, stream));
  return GX_OK;
}

// --------------------------------------------------------------------- RCCL ---
// librccl is opened on first use so that single-GPU users never load it.
namespace {
typedef int (*fn_get_unique_id)(void*);
struct UniqueId {  // ncclUniqueId, passed by value
  char internal[128];
};
typedef int (*fn_comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef const char* (*fn_error_string)(int);
constexpr int kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2;

void* open_rccl() {
  void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  return lib;
}
}  // namespace

extern "C" int gx_comm_unique_id(void* id128) {
  if (!id128) return fail(GX_ERR_INVALID, "gx_comm_unique_id: null argument");
  void* lib = open_rccl();
  if (!lib) return fail(GX_ERR_RCCL, "cannot load librccl.so: %s", dlerror());
  auto get = (fn_get_unique_id)dlsym(lib, "ncclGetUniqueId");
  if (!get) return fail(GX_ERR_RCCL, "librccl.so has no ncclGetUniqueId");
  int rc = get(id128);
  if (rc != 0) return fail(GX_ERR_RCCL, "ncclGetUniqueId failed with %d", rc);
  return GX_OK;
}

extern "C" int gx_comm_destroy(gx_comm* m) {
  if (!m) return GX_OK;
  if (m->ctx) (void)hipSetDevice(m->ctx->device);
  if (m->comm && m->lib) {
    auto destroy = (fn_comm_destroy)dlsym(m->lib, "ncclCommDestroy");
    if (destroy) (void)destroy(m->comm);
  }
  if (m->d_buf) (void)hipFree(m->d_buf);
  if (m->h_buf) (void)hipHostFree(m->h_buf);
  if (m->ctx) m->ctx->live_handles--;
  delete m;
  return GX_OK;
}

extern "C" int gx_comm_init(gx_ctx* ctx, const void* id128, int rank, int world, gx_comm** out) {
  if (!ctx || !id128 || !out) return fail(GX_ERR_INVALID, "gx_comm_init: null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(GX_ERR_INVALID, "gx_comm_init: rank %d of %d", rank, world);
  HIP_TRY(hipSetDevice(ctx->device));
  gx_comm* m = new (std::nothrow) gx_comm();
  if (!m) return fail(GX_ERR_OOM, "gx_comm_init: host allocation failed");
  m->ctx = ctx;
  ctx->live_handles++;
  m->rank = rank;
  m->world = world;
  m->lib = open_rccl();
  if (!m->lib) {
    gx_comm_destroy(m);
    return fail(GX_ERR_RCCL, "cannot load librccl.so: %s", dlerror());
  }
  auto init = (fn_comm_init_rank_t)dlsym(m->lib, "ncclCommInitRank");
  if (!init) {
    gx_comm_destroy(m);
    return fail(GX_ERR_RCCL, "librccl.so has no ncclCommInitRank");
  }
  UniqueId id;
  std::memcpy(id.internal, id128, sizeof id.internal);
  int rc = init(&m->comm, world, id, rank);
  if (rc != 0) {
    m->comm = nullptr;
    gx_comm_destroy(m);
    return fail(GX_ERR_RCCL, "ncclCommInitRank failed with %d", rc);
  }
  hipError_t e = hipMalloc((void**)&m->d_buf, 8 * sizeof(double));
  if (e == hipSuccess) e = hipHostMalloc((void**)&m->h_buf, 8 * sizeof(double), hipHostMallocDefault);
  if (e != hipSuccess) {
    gx_comm_destroy(m);
    return fail(GX_ERR_HIP, "gx_comm_init: buffer allocation failed: %s", hipGetErrorString(e));
  }
  *out = m;
  return GX_OK;
}

extern "C" int gx_stats_allreduce(gx_comm* m, gx_vec* v, gx_stats* out) {
  if (!m || !v || !out) return fail(GX_ERR_INVALID, "gx_stats_allreduce: null argument");
  gx_stats local;
  int rc = gx_vec_stats(v, &local);
  if (rc) return rc;
  gx_ctx* c = m->ctx;
  // counts are exact in double below 2^53; min is reduced as max of the negation
  m->h_buf[0] = (double)local.episodes;
  m->h_buf[1] = (double)local.sum_length;
  m->h_buf[2] = local.sum_return;
  m->h_buf[3] = local.sum_return_sq;
  m->h_buf[4] = -local.min_return;
  m->h_buf[5] = local.max_return;
  HIP_TRY(hipMemcpyAsync(m->d_buf, m->h_buf, 6 * sizeof(double), hipMemcpyHostToDevice, c->compute));
  auto all_reduce = (fn_all_reduce)dlsym(m->lib, "ncclAllReduce");
  auto group_start = (fn_group)dlsym(m->lib, "ncclGroupStart");
  auto group_end = (fn_group)dlsym(m->lib, "ncclGroupEnd");
  if (!all_reduce || !group_start || !group_end) return fail(GX_ERR_RCCL, "librccl.so lacks ncclAllReduce/ncclGroupStart/ncclGroupEnd");
  int r0 = group_start();
  int r1 = all_reduce(m->d_buf, m->d_buf, 4, kNcclFloat64, kNcclSum, m->comm, c->compute);
  int r2 = all_reduce(m->d_buf + 4, m->d_buf + 4, 2, kNcclFloat64, kNcclMax, m->comm, c->compute);
  int r3 = group_end();
  if (r0 || r1 || r2 || r3) return fail(GX_ERR_RCCL, "ncclAllReduce failed (%d %d %d %d)", r0, r1, r2, r3);
  HIP_TRY(hipMemcpyAsync(m->h_buf, m->d_buf, 6 * sizeof(double), hipMemcpyDeviceToHost, c->compute));
  HIP_TRY(hipStreamSynchronize(c->compute));
  out->episodes = (int64_t)m->h_buf[0];
  out->sum_length = (int64_t)m->h_buf[1];
  out->sum_return = m->h_buf[2];
  out->sum_return_sq = m->h_buf[3];
  out->min_return = -m->h_buf[4];
  out->max_return = m->h_buf[5];
  return GX_OK;
}
