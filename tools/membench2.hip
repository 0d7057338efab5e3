// HBM streaming ceilings at footprints beyond the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { f4* r[8]; f4* w[8]; };

template <int R, int W, int U, int NT>
__global__ void __launch_bounds__(256) k(Ptrs p, size_t nvec, float* sink) {
  f4 acc = {0, 0, 0, 0};
  // each thread handles U vectors 256 apart per iteration (U KiB per wave per stream in flight)
  for (size_t i = ((size_t)blockIdx.x * 256) * U + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256 * U) {
    f4 loc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; u++) {
      size_t j = i + (size_t)u * 256;
      if (j < nvec) {
#pragma unroll
        for (int s = 0; s < R; s++) { f4 v = NT ? __builtin_nontemporal_load(&p.r[s][j]) : p.r[s][j]; loc += v; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      size_t j = i + (size_t)u * 256;
      if (j < nvec) {
#pragma unroll
        for (int s = 0; s < W; s++) { f4 o = loc + (float)s; if (NT) __builtin_nontemporal_store(o, &p.w[s][j]); else p.w[s][j] = o; }
      }
    }
    acc += loc;
  }
  if (W == 0 && acc.x == 12345.678f) sink[0] = acc.y;
}

template <int R, int W, int U, int NT>
double run(Ptrs p, size_t nvec, int blocks, float* sink) {
  int iters = 5;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<R, W, U, NT>), dim3(blocks), dim3(256), 0, 0, p, nvec, sink);
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<R, W, U, NT>), dim3(blocks), dim3(256), 0, 0, p, nvec, sink);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return (double)(R + W) * nvec * 16 * iters / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 26;   // log2 floats per stream (26 -> 256 MiB per stream)
  size_t nvec = ((size_t)1 << lg) / 4;
  Ptrs p; float* sink;
  CHECK(hipMalloc(&sink, 64));
  for (int s = 0; s < 8; s++) { CHECK(hipMalloc(&p.r[s], nvec * 16)); CHECK(hipMemset(p.r[s], 0, nvec * 16)); CHECK(hipMalloc(&p.w[s], nvec * 16)); CHECK(hipMemset(p.w[s], 0, nvec * 16)); }
  printf("floats/stream 2^%d = %.0f MiB per stream\n", lg, nvec * 16 / 1048576.0);
  for (int blocks : {2048, 8192}) {
    printf("blocks %d\n", blocks);
    printf("  R1W0 %.0f  R2W0 %.0f R8W0 %.0f | nt R8W0 %.0f\n", run<1,0,1,0>(p,nvec,blocks,sink), run<2,0,1,0>(p,nvec,blocks,sink), run<8,0,1,0>(p,nvec,blocks,sink), run<8,0,1,1>(p,nvec,blocks,sink));
    printf("  R0W1 %.0f  R0W2 %.0f R0W8 %.0f | nt R0W8 %.0f\n", run<0,1,1,0>(p,nvec,blocks,sink), run<0,2,1,0>(p,nvec,blocks,sink), run<0,8,1,0>(p,nvec,blocks,sink), run<0,8,1,1>(p,nvec,blocks,sink));
    printf("  R1W1 %.0f nt %.0f | R2W2 %.0f nt %.0f | R4W4 %.0f nt %.0f | R7W7 %.0f nt %.0f | R8W8 %.0f nt %.0f\n",
      run<1,1,1,0>(p,nvec,blocks,sink), run<1,1,1,1>(p,nvec,blocks,sink), run<2,2,1,0>(p,nvec,blocks,sink), run<2,2,1,1>(p,nvec,blocks,sink),
      run<4,4,1,0>(p,nvec,blocks,sink), run<4,4,1,1>(p,nvec,blocks,sink), run<7,7,1,0>(p,nvec,blocks,sink), run<7,7,1,1>(p,nvec,blocks,sink), run<8,8,1,0>(p,nvec,blocks,sink), run<8,8,1,1>(p,nvec,blocks,sink));
    printf("  U=2: R1W1 %.0f R4W4 %.0f R7W7 %.0f nt %.0f | U=4: R1W1 %.0f R4W4 %.0f nt %.0f\n",
      run<1,1,2,0>(p,nvec,blocks,sink), run<4,4,2,0>(p,nvec,blocks,sink), run<7,7,2,0>(p,nvec,blocks,sink), run<7,7,2,1>(p,nvec,blocks,sink),
      run<1,1,4,0>(p,nvec,blocks,sink), run<4,4,4,0>(p,nvec,blocks,sink), run<4,4,4,1>(p,nvec,blocks,sink));
    fflush(stdout);
  }
  return 0;
}
