// How many cycles a wave needs between dependent v_fma_f64 (and v_fma_f32, for comparison), and how many resident waves per SIMD hide it:
// CHAINS independent Horner chains per lane, WAVES workgroup waves per SIMD (one workgroup of 4*WAVES waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T, int CHAINS>
__global__ void __launch_bounds__(1024) chain_kernel(T* out, T z, int iters) {
  T r[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) r[c] = (T)threadIdx.x * (T)1e-9 + (T)c;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) r[c] = sizeof(T) == 8 ? (T)__builtin_fma((double)z, (double)r[c], 0.333) : (T)__builtin_fmaf((float)z, (float)r[c], 0.333f);
    }
  }
  T s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += r[c];
  if (s == (T)12345.678) out[threadIdx.x] = s;
}

template <class T, int CHAINS>
static void run(int waves_per_simd, T* d) {
  const int total_fma = 1 << 16;                 // per lane
  const int iters = total_fma / (16 * CHAINS);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  dim3 grid(256), block(64 * 4 * waves_per_simd);
  hipLaunchKernelGGL((chain_kernel<T, CHAINS>), grid, block, 0, 0, d, (T)0.999, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((chain_kernel<T, CHAINS>), grid, block, 0, 0, d, (T)0.999, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  // wave-instructions per SIMD = waves_per_simd * total_fma; at 4 cycles each and 2.4 GHz:
  const double ideal_ms = (double)waves_per_simd * total_fma * 4 / 2.4e9 * 1e3;
  printf("%s chains %d waves/SIMD %d: %.3f ms (4 cycles per instruction at 2.4 GHz: %.3f ms) -> %.2f cycles per wave-instruction issued per SIMD\n",
         sizeof(T) == 8 ? "f64" : "f32", CHAINS, waves_per_simd, ms, ideal_ms, ms / ideal_ms * 4);
}

int main() {
  double* d; CK(hipMalloc(&d, 8192));
  for (int w = 1; w <= 4; w++) { run<double, 1>(w, d); run<double, 2>(w, d); run<double, 4>(w, d); }
  run<double, 8>(4, d);
  float* f = reinterpret_cast<float*>(d);
  for (int w = 1; w <= 4; w *= 2) { run<float, 1>(w, f); run<float, 4>(w, f); }
  run<float, 8>(4, f);
  return 0;
}
