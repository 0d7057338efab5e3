// Host-callable launchers implemented by the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "gx_device.h"

namespace gx {

// classic control (gx_classic.hip)
// fused: auto-reset + RecordEpisodeStatistics on and no TimeLimit; bounded: the env's angle
// invariants hold for the current state (see gx_classic_kernel.h)
// env_id: a GX_ENV_* of the five, or kEnvAcrobotAngles -- Acrobot in EXACT mode on rows the library owns, which
// has two more real rows (the angles) behind the six of the state image
constexpr int kEnvAcrobotAngles = 1000;
hipError_t launch_classic_step(int env_id, const StepArgs& a, int mode, int policy, bool fused, bool bounded, dim3 grid,
                               hipStream_t stream);
hipError_t launch_classic_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream);
hipError_t launch_classic_sample(int env_id, const StepArgs& a, void* out, dim3 grid, hipStream_t stream);
// the derived rows from the rows of the state image, after the image was written from outside (kEnvAcrobotAngles only)
hipError_t launch_classic_derive(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream);

// Acrobot's part of the above (gx_acrobot.hip); env_id GX_ENV_ACROBOT or kEnvAcrobotAngles
hipError_t launch_acrobot_step(int env_id, const StepArgs& a, int mode, int policy, bool fused, bool bounded, dim3 grid,
                               hipStream_t stream);
hipError_t launch_acrobot_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream);
hipError_t launch_acrobot_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream);
hipError_t launch_acrobot_derive(const StepArgs& a, dim3 grid, hipStream_t stream);

// the same five envs in binary64, GX_MODE_FP64 (gx_classic64.hip)
hipError_t launch_classic64_step(int env_id, const StepArgs& a, int policy, bool fused, bool bounded, dim3 grid,
                                 hipStream_t stream);
hipError_t launch_classic64_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream);
hipError_t launch_classic64_sample(int env_id, const StepArgs& a, void* out, dim3 grid, hipStream_t stream);

// Acrobot, RK4 in binary64 (gx_acrobot64.hip)
hipError_t launch_acrobot64_step(const StepArgs& a, int policy, bool fused, dim3 grid, hipStream_t stream);
hipError_t launch_acrobot64_reset(const StepArgs& a, int only_observe, dim3 grid, hipStream_t stream);
hipError_t launch_acrobot64_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream);

// table-driven envs and DummyEnv (gx_tabular.hip)
hipError_t launch_tabular_step(int env_id, const StepArgs& a, int policy, dim3 grid, int block, size_t lds_bytes, hipStream_t stream);
hipError_t launch_tabular_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream);
hipError_t launch_tabular_sample(int env_id, const StepArgs& a, void* out, dim3 grid, hipStream_t stream);

// Blackjack (gx_blackjack.hip); always launched with kBlock threads per workgroup
hipError_t launch_blackjack_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream);
hipError_t launch_blackjack_reset(const StepArgs& a, dim3 grid, hipStream_t stream);
hipError_t launch_blackjack_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream);

// Pong (gx_pong.hip)
hipError_t launch_pong_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream);
hipError_t launch_pong_reset(const StepArgs& a, dim3 grid, hipStream_t stream);
// a.render: the records the step / reset / pack kernels of the game wrote
hipError_t launch_pong_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream);
hipError_t launch_pong_pack(const StepArgs& a, dim3 grid, hipStream_t stream);
constexpr size_t kPongRecordBytes = 32;
hipError_t launch_pong_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream);

// Breakout (gx_breakout.hip)
hipError_t launch_breakout_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream);
hipError_t launch_breakout_reset(const StepArgs& a, dim3 grid, hipStream_t stream);
hipError_t launch_breakout_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream);
hipError_t launch_breakout_pack(const StepArgs& a, dim3 grid, hipStream_t stream);
constexpr size_t kBreakoutRecordBytes = 64;
hipError_t launch_breakout_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream);

// SpaceInvaders (gx_invaders.hip)
hipError_t launch_invaders_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream);
hipError_t launch_invaders_reset(const StepArgs& a, dim3 grid, hipStream_t stream);
hipError_t launch_invaders_raster(const StepArgs& a, uint8_t* frames, hipStream_t stream);
hipError_t launch_invaders_pack(const StepArgs& a, dim3 grid, hipStream_t stream);
constexpr size_t kInvadersRecordBytes = 128;
hipError_t launch_invaders_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream);

// host path of small batches (gx_hostpath.hip)
struct GatherSegment {
  const void* src;  // device row
  void* dst;        // device address of the pinned staging buffer
  uint64_t bytes;
};
constexpr int kMaxGatherSegments = 12;
struct GatherArgs {
  GatherSegment seg[kMaxGatherSegments];
  int count;
};
hipError_t launch_gather_segments(const GatherArgs& g, dim3 grid, hipStream_t stream);

// statistics (gx_stats.hip)
hipError_t launch_stats_reduce(const StatSlot* slots, int num_slots, StatSlot* out, hipStream_t stream);
hipError_t launch_stats_init(StatSlot* slots, int num_slots, hipStream_t stream);
// EnvWrapper.cs:34-38: Reset() with an open episode records it first
hipError_t launch_stats_materialize_returns(const StepArgs& a, const int32_t* length_row, dim3 grid, hipStream_t stream);
hipError_t launch_stats_flush_open(const StepArgs& a, int ep_return_is_f64, const int32_t* length_row, dim3 grid,
                                   hipStream_t stream);

}  // namespace gx
