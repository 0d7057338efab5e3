// CartPole-shaped in-place update (5 read+write rows, 1 read-only, 1 write-only + a byte row),
// 16 Mi envs, launched back to back.  Compares cache policies for a footprint (464 MiB) that
// exceeds the 256 MiB Infinity Cache:
//   alt     : alternating sweep direction, write-only rows non-temporal (what the engine does)
//   split R : the first R of the batch with ordinary loads/stores, the rest non-temporal both ways,
//             so that a fixed part could stay cache-resident while the rest streams past it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { f4* rw[5]; f4* ro; f4* wo; unsigned* flags; };

template <int SPLIT>
__global__ void __launch_bounds__(64) k(Ptrs p, size_t nvec, int reverse, size_t resident) {
  for (size_t i0 = (size_t)blockIdx.x * 64 + threadIdx.x; i0 < nvec; i0 += (size_t)gridDim.x * 64) {
    size_t i = reverse ? nvec - 1 - i0 : i0;
    f4 v[5], a;
    const bool stream = SPLIT && i >= resident;  // uniform per wave when resident is a multiple of 64
    if (stream) {
#pragma unroll
      for (int s = 0; s < 5; s++) v[s] = __builtin_nontemporal_load(&p.rw[s][i]);
      a = __builtin_nontemporal_load(&p.ro[i]);
#pragma unroll
      for (int s = 0; s < 5; s++) __builtin_nontemporal_store(v[s] + a, &p.rw[s][i]);
    } else {
#pragma unroll
      for (int s = 0; s < 5; s++) v[s] = p.rw[s][i];
      a = p.ro[i];
#pragma unroll
      for (int s = 0; s < 5; s++) p.rw[s][i] = v[s] + a;
    }
    f4 o = v[0] + v[4];
    __builtin_nontemporal_store(o, &p.wo[i]);
    __builtin_nontemporal_store((unsigned)(o.x > 0), &p.flags[i]);
  }
}

template <int SPLIT>
double run(Ptrs p, size_t nvec, int alternate, size_t resident) {
  const int iters = 40;
  const int blocks = (int)((nvec / 64 + 1) / 2);
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 4; i++) hipLaunchKernelGGL((k<SPLIT>), dim3(blocks), dim3(64), 0, 0, p, nvec, alternate ? (i & 1) : 0, resident);
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<SPLIT>), dim3(blocks), dim3(64), 0, 0, p, nvec, alternate ? (i & 1) : 0, resident);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double bytes = (double)nvec * (5 * 32 + 16 + 16 + 4);
  return bytes * iters / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 24;
  size_t nvec = ((size_t)1 << lg) / 4;
  Ptrs p;
  for (int s = 0; s < 5; s++) { CHECK(hipMalloc(&p.rw[s], nvec * 16)); CHECK(hipMemset(p.rw[s], 0, nvec * 16)); }
  CHECK(hipMalloc(&p.ro, nvec * 16)); CHECK(hipMemset(p.ro, 0, nvec * 16));
  CHECK(hipMalloc(&p.wo, nvec * 16)); CHECK(hipMalloc(&p.flags, nvec * 4));
  printf("2^%d envs: footprint %.0f MiB\n", lg, nvec * (7.0 * 16 + 4) / 1048576.0);
  printf("forward %.0f GB/s, alternating %.0f GB/s\n", run<0>(p, nvec, 0, 0), run<0>(p, nvec, 1, 0));
  for (int pct : {0, 20, 30, 40, 50, 60, 100}) {
    size_t resident = (nvec * pct / 100) & ~(size_t)63;
    printf("split %3d%% resident: forward %.0f GB/s, alternating %.0f GB/s\n", pct, run<1>(p, nvec, 0, resident), run<1>(p, nvec, 1, resident));
    fflush(stdout);
  }
  return 0;
}
