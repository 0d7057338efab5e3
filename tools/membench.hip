// Micro-benchmark: R read streams -> W write streams of 16-byte lanes, the access
// shape of the step kernels.  Explores grid mapping and cache policy.
// build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { f4* r[8]; f4* w[8]; };

template <int R, int W, int MODE, int NT>
__global__ void __launch_bounds__(256) k(Ptrs p, size_t nvec, size_t chunk) {
  size_t start, end, stride;
  if (MODE == 0) { start = (size_t)blockIdx.x * 256 + threadIdx.x; end = nvec; stride = (size_t)gridDim.x * 256; }
  else { start = (size_t)blockIdx.x * chunk + threadIdx.x; end = min(nvec, (size_t)(blockIdx.x + 1) * chunk); stride = 256; }
  for (size_t i = start; i < end; i += stride) {
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < R; s++) {
      f4 v = NT ? __builtin_nontemporal_load(&p.r[s][i]) : p.r[s][i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
#pragma unroll
    for (int s = 0; s < W; s++) {
      f4 o = {acc.x + s, acc.y, acc.z, acc.w};
      if (NT) __builtin_nontemporal_store(o, &p.w[s][i]); else p.w[s][i] = o;
    }
  }
}

template <int R, int W, int MODE, int NT>
double run(Ptrs p, size_t nvec, int blocks, int iters) {
  size_t chunk = ((nvec + blocks - 1) / blocks + 255) / 256 * 256;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<R, W, MODE, NT>), dim3(blocks), dim3(256), 0, 0, p, nvec, chunk);
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<R, W, MODE, NT>), dim3(blocks), dim3(256), 0, 0, p, nvec, chunk);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return (double)(R + W) * nvec * 16 * iters / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  size_t nvec = (size_t)(1 << 24) / 4;  // 16 Mi floats per stream = 4 Mi float4
  Ptrs p;
  // in-place style: write streams alias the read streams where possible (state rows are updated in place)
  for (int s = 0; s < 8; s++) { CHECK(hipMalloc(&p.r[s], nvec * 16)); CHECK(hipMemset(p.r[s], 0, nvec * 16)); }
  int inplace = argc > 1 ? atoi(argv[1]) : 1;
  for (int s = 0; s < 8; s++) { if (inplace) p.w[s] = p.r[s]; else { CHECK(hipMalloc(&p.w[s], nvec * 16)); } }
  int iters = 20;
  for (int blocks : {1024, 2048, 4096, 16384}) {
    printf("blocks %5d | R1W1 gs %.0f nt %.0f | R4W4 gs %.0f ch %.0f nt %.0f ntch %.0f | R7W7 gs %.0f ch %.0f nt %.0f ntch %.0f | R7W8 gs %.0f nt %.0f\n", blocks,
           run<1, 1, 0, 0>(p, nvec, blocks, iters), run<1, 1, 0, 1>(p, nvec, blocks, iters),
           run<4, 4, 0, 0>(p, nvec, blocks, iters), run<4, 4, 1, 0>(p, nvec, blocks, iters), run<4, 4, 0, 1>(p, nvec, blocks, iters), run<4, 4, 1, 1>(p, nvec, blocks, iters),
           run<7, 7, 0, 0>(p, nvec, blocks, iters), run<7, 7, 1, 0>(p, nvec, blocks, iters), run<7, 7, 0, 1>(p, nvec, blocks, iters), run<7, 7, 1, 1>(p, nvec, blocks, iters),
           run<7, 8, 0, 0>(p, nvec, blocks, iters), run<7, 8, 0, 1>(p, nvec, blocks, iters));
    fflush(stdout);
  }
  return 0;
}
