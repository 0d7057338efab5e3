// How many cycles a wave needs between dependent v_fma_f64, and how many resident waves per SIMD hide it:
// CHAINS independent Horner chains per lane, WAVES workgroup waves per SIMD (one workgroup of 4*WAVES waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CHAINS>
__global__ void __launch_bounds__(1024) chain_kernel(double* out, double z, int iters) {
  double r[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) r[c] = (double)threadIdx.x * 1e-9 + c;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) r[c] = __builtin_fma(z, r[c], 0.333);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += r[c];
  if (s == 12345.678) out[threadIdx.x] = s;
}

template <int CHAINS>
static void run(int waves_per_simd, double* d) {
  const int total_fma = 1 << 16;                 // per lane
  const int iters = total_fma / (16 * CHAINS);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  dim3 grid(256), block(64 * 4 * waves_per_simd);
  hipLaunchKernelGGL(chain_kernel<CHAINS>, grid, block, 0, 0, d, 0.999, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(chain_kernel<CHAINS>, grid, block, 0, 0, d, 0.999, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  // wave-instructions per SIMD = waves_per_simd * total_fma; at 4 cycles each and 2.4 GHz:
  const double ideal_ms = (double)waves_per_simd * total_fma * 4 / 2.4e9 * 1e3;
  printf("chains %d waves/SIMD %d: %.3f ms (4 cycles per instruction at 2.4 GHz: %.3f ms) -> %.2f cycles per wave-instruction issued per SIMD\n",
         CHAINS, waves_per_simd, ms, ideal_ms, ms / ideal_ms * 4);
}

int main() {
  double* d; CK(hipMalloc(&d, 8192));
  for (int w = 1; w <= 4; w++) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
  return 0;
}
