// CartPole-shaped in-place update (6 read+write rows, 1 read-only, 1 write-only + a byte row),
// 16 Mi envs, launched back to back: forward-only sweeps vs alternating sweep direction, so the
// tail of one sweep is still in the 256 MiB Infinity Cache when the next one starts there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { f4* rw[6]; f4* ro; f4* wo; unsigned* flags; };

template <int NT_OUT>
__global__ void __launch_bounds__(256) k(Ptrs p, size_t nvec, int reverse) {
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < nvec; i0 += (size_t)gridDim.x * 256) {
    size_t i = reverse ? nvec - 1 - i0 : i0;
    f4 v[6];
#pragma unroll
    for (int s = 0; s < 6; s++) v[s] = p.rw[s][i];
    f4 a = p.ro[i];
#pragma unroll
    for (int s = 0; s < 6; s++) p.rw[s][i] = v[s] + a;
    f4 o = v[0] + v[5];
    if (NT_OUT) { __builtin_nontemporal_store(o, &p.wo[i]); __builtin_nontemporal_store((unsigned)(o.x > 0), &p.flags[i]); }
    else { p.wo[i] = o; p.flags[i] = o.x > 0; }
  }
}

static size_t g_lds = 0;  // dynamic LDS per workgroup: caps resident workgroups per CU (occupancy experiment)

template <int NT_OUT>
double run(Ptrs p, size_t nvec, int blocks, int alternate) {
  int iters = 40;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 4; i++) hipLaunchKernelGGL((k<NT_OUT>), dim3(blocks), dim3(256), g_lds, 0, p, nvec, alternate ? (i & 1) : 0);
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<NT_OUT>), dim3(blocks), dim3(256), g_lds, 0, p, nvec, alternate ? (i & 1) : 0);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double bytes = (double)nvec * (6 * 32 + 16 + 16 + 4);
  return bytes * iters / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 24;
  g_lds = argc > 2 ? (size_t)atoi(argv[2]) : 0;
  if (g_lds > 65536) {
    CHECK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
    CHECK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
  }
  printf("dynamic LDS %zu B per workgroup -> at most %d workgroups (x4 waves) per CU\n", g_lds, g_lds ? (int)(163840 / g_lds) : 8);
  size_t nvec = ((size_t)1 << lg) / 4;
  Ptrs p;
  for (int s = 0; s < 6; s++) { CHECK(hipMalloc(&p.rw[s], nvec * 16)); CHECK(hipMemset(p.rw[s], 0, nvec * 16)); }
  CHECK(hipMalloc(&p.ro, nvec * 16)); CHECK(hipMemset(p.ro, 0, nvec * 16));
  CHECK(hipMalloc(&p.wo, nvec * 16)); CHECK(hipMalloc(&p.flags, nvec * 4));
  printf("2^%d envs: footprint %.0f MiB\n", lg, nvec * (8.0 * 16 + 4) / 1048576.0);
  for (int blocks : {1024, 2048, 4096}) {
    printf("blocks %d: forward %.0f GB/s  alternating %.0f GB/s | nt outputs: forward %.0f alternating %.0f\n", blocks,
           run<0>(p, nvec, blocks, 0), run<0>(p, nvec, blocks, 1), run<1>(p, nvec, blocks, 0), run<1>(p, nvec, blocks, 1));
    fflush(stdout);
  }
  return 0;
}
