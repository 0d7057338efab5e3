// As membench4's alternating sweep (5 read+write rows, 1 read-only, 1 write-only + a byte row,
// 64-thread workgroups, two trips each), with a stand-in for the step arithmetic between the
// loads and the stores (WORK dependent multiply-adds per lane per trip), and optionally the
// second trip's loads issued before the first trip's arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { f4* rw[5]; f4* ro; f4* wo; unsigned* flags; };

struct Regs { f4 v[5]; f4 a; };
__device__ __forceinline__ void load(const Ptrs& p, size_t i, Regs& r) {
#pragma unroll
  for (int s = 0; s < 5; s++) r.v[s] = p.rw[s][i];
  r.a = p.ro[i];
}
template <int WORK>
__device__ __forceinline__ void work_store(const Ptrs& p, size_t i, Regs& r) {
  f4 x = r.v[0], y = r.v[1];
#pragma unroll 4
  for (int w = 0; w < WORK / 8; w++) { x = x * y + r.a; y = y * x + r.a; }  // 8 scalar fmas per iteration
  r.v[0] = x; r.v[1] = y;
#pragma unroll
  for (int s = 0; s < 5; s++) p.rw[s][i] = r.v[s] + r.a;
  f4 o = r.v[0] + r.v[4];
  __builtin_nontemporal_store(o, &p.wo[i]);
  __builtin_nontemporal_store((unsigned)(o.x > 0), &p.flags[i]);
}

template <int WORK, int PREFETCH>
__global__ void __launch_bounds__(64) k(Ptrs p, size_t nvec, int reverse) {
  const size_t stride = (size_t)gridDim.x * 64;
  size_t i0 = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i0 >= nvec) return;
  if (PREFETCH) {
    Regs cur, nxt;
    load(p, reverse ? nvec - 1 - i0 : i0, cur);
    for (;;) {
      const size_t n0 = i0 + stride;
      const bool more = n0 < nvec;
      if (more) load(p, reverse ? nvec - 1 - n0 : n0, nxt);
      work_store<WORK>(p, reverse ? nvec - 1 - i0 : i0, cur);
      if (!more) break;
      cur = nxt;
      i0 = n0;
    }
  } else {
    for (; i0 < nvec; i0 += stride) {
      Regs cur;
      const size_t i = reverse ? nvec - 1 - i0 : i0;
      load(p, i, cur);
      work_store<WORK>(p, i, cur);
    }
  }
}

static size_t g_lds = 0;  // dynamic LDS per workgroup: caps the resident waves per CU

template <int WORK, int PREFETCH>
double run(Ptrs p, size_t nvec) {
  const int iters = 40;
  const int blocks = (int)((nvec / 64 + 1) / 2);
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 4; i++) hipLaunchKernelGGL((k<WORK, PREFETCH>), dim3(blocks), dim3(64), g_lds, 0, p, nvec, i & 1);
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<WORK, PREFETCH>), dim3(blocks), dim3(64), g_lds, 0, p, nvec, i & 1);
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
  for (size_t lds : {(size_t)0, (size_t)6800, (size_t)10200, (size_t)13600}) {
    g_lds = lds;
    printf("LDS %5zu B per workgroup (at most %d waves per CU):\n", lds, lds ? (int)(163840 / lds) : 32);
    printf("  work    0: plain %.0f GB/s, prefetch %.0f GB/s\n", run<0, 0>(p, nvec), run<0, 1>(p, nvec));
    printf("  work 1024: plain %.0f GB/s, prefetch %.0f GB/s\n", run<1024, 0>(p, nvec), run<1024, 1>(p, nvec));
    printf("  work 2048: plain %.0f GB/s, prefetch %.0f GB/s\n", run<2048, 0>(p, nvec), run<2048, 1>(p, nvec));
    printf("  work 3072: plain %.0f GB/s, prefetch %.0f GB/s\n", run<3072, 0>(p, nvec), run<3072, 1>(p, nvec));
    fflush(stdout);
  }
  return 0;
}
