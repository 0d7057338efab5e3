// Host path of small batches: the results of one step (reward, done, truncated and the
// observation rows) are gathered by one kernel into a pinned, device-mapped staging buffer, so
// a step costs two kernel launches and one stream synchronisation instead of an upload and up
// to four downloads on a second stream with two events between them.
#include "gx_device.h"
#include "gx_launch.h"

namespace gx {

// Copies every segment: 16-byte vectors while both ends allow it, then the tail bytes.
// Sources and destinations are 16-byte aligned (device rows; staging offsets).
__global__ void __launch_bounds__(kBlock) gather_segments_kernel(GatherArgs g) {
  for (int s = 0; s < g.count; s++) {
    const char* src = static_cast<const char*>(g.seg[s].src);
    char* dst = static_cast<char*>(g.seg[s].dst);
    const uint64_t bytes = g.seg[s].bytes;
    const uint64_t vecs = bytes / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < vecs; i += (uint64_t)gridDim.x * kBlock)
      reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    const uint64_t tail = bytes - vecs * 16;
    if (blockIdx.x == 0 && threadIdx.x < tail) dst[vecs * 16 + threadIdx.x] = src[vecs * 16 + threadIdx.x];
  }
}

hipError_t launch_gather_segments(const GatherArgs& g, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(gather_segments_kernel, grid, dim3(kBlock), 0, stream, g);
  return hipGetLastError();
}

}  // namespace gx
