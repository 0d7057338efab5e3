// Episode-statistics kernels: slot initialisation, the final reduction over
// the per-wave slots, and the Reset()-time flush of open episodes.
#include "gx_device.h"
#include "gx_launch.h"

namespace gx {

__global__ void __launch_bounds__(kBlock) stats_init_kernel(StatSlot* slots, int n) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    StatSlot s;
    s.episodes = 0; s.sum_return = 0.0; s.sum_length = 0; s.sum_return_sq = 0.0;
    s.min_return = __builtin_huge_val(); s.max_return = -__builtin_huge_val();
    s.pad[0] = s.pad[1] = 0;
    slots[i] = s;
  }
}

// One workgroup; slot 0 of `scratch` is the commit target.
__global__ void __launch_bounds__(kBlock) stats_reduce_kernel(const StatSlot* slots, int n, StatSlot* out) {
  __shared__ StatSlot part[kBlock];
  StatSlot t;
  t.episodes = 0; t.sum_return = 0.0; t.sum_length = 0; t.sum_return_sq = 0.0;
  t.min_return = __builtin_huge_val(); t.max_return = -__builtin_huge_val();
  t.pad[0] = t.pad[1] = 0;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const StatSlot o = slots[i];
    t.episodes += o.episodes; t.sum_return += o.sum_return; t.sum_length += o.sum_length;
    t.sum_return_sq += o.sum_return_sq;
    t.min_return = o.min_return < t.min_return ? o.min_return : t.min_return;
    t.max_return = o.max_return > t.max_return ? o.max_return : t.max_return;
  }
  part[threadIdx.x] = t;
  __syncthreads();
  // fixed-order tree so the floating-point sums are reproducible run to run
  for (int w = kBlock / 2; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
      StatSlot& x = part[threadIdx.x];
      const StatSlot& o = part[threadIdx.x + w];
      x.episodes += o.episodes; x.sum_return += o.sum_return; x.sum_length += o.sum_length;
      x.sum_return_sq += o.sum_return_sq;
      x.min_return = o.min_return < x.min_return ? o.min_return : x.min_return;
      x.max_return = o.max_return > x.max_return ? o.max_return : x.max_return;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = part[0];
}

// RecordEpisodeStatistics.Reset(), EnvWrapper.cs:32-42: an episode that is
// open (length > 0) is recorded before the counters are zeroed.
template <class EpRet>
__global__ void __launch_bounds__(kBlock) stats_flush_open_kernel(StepArgs a, const int32_t* length_row) {
  StatAcc acc;
  EpRet* ep_ret = static_cast<EpRet*>(a.ep_return);
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock) {
    const int len = length_row[i];
    if (len > 0) acc.add(a.derive_returns ? (double)(a.ret_per_step * len) : (double)ep_ret[i], len);
    ep_ret[i] = 0;
    if (a.ep_length) a.ep_length[i] = 0;
  }
  block_commit_stats(acc, a.stat_slots);
}

// Writes the derived returns (ret_per_step x step counter) into the ep_return row, for a batch
// that stops deriving them (its state is about to be replaced from outside).
__global__ void __launch_bounds__(kBlock) stats_materialize_returns_kernel(StepArgs a, const int32_t* length_row) {
  int32_t* ep_ret = static_cast<int32_t*>(a.ep_return);
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock)
    ep_ret[i] = a.ret_per_step * length_row[i];
}

hipError_t launch_stats_materialize_returns(const StepArgs& a, const int32_t* length_row, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(stats_materialize_returns_kernel, grid, dim3(kBlock), 0, stream, a, length_row);
  return hipGetLastError();
}

hipError_t launch_stats_init(StatSlot* slots, int num_slots, hipStream_t stream) {
  hipLaunchKernelGGL(stats_init_kernel, dim3((num_slots + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, slots, num_slots);
  return hipGetLastError();
}

hipError_t launch_stats_reduce(const StatSlot* slots, int num_slots, StatSlot* out, hipStream_t stream) {
  hipLaunchKernelGGL(stats_reduce_kernel, dim3(1), dim3(kBlock), 0, stream, slots, num_slots, out);
  return hipGetLastError();
}

hipError_t launch_stats_flush_open(const StepArgs& a, int ep_return_is_f64, const int32_t* length_row, dim3 grid,
                                   hipStream_t stream) {
  if (ep_return_is_f64)
    hipLaunchKernelGGL(stats_flush_open_kernel<double>, grid, dim3(kBlock), 0, stream, a, length_row);
  else
    hipLaunchKernelGGL(stats_flush_open_kernel<int32_t>, grid, dim3(kBlock), 0, stream, a, length_row);
  return hipGetLastError();
}

}  // namespace gx
