// How accurate is v_rcp_f64, and how many Newton steps does math::rcp64 need?  Prints the largest error, in units
// in the last place of the correctly rounded quotient, of the raw estimate and of one and two refinement steps,
// over 2^26 arguments spread over [0.5, 8).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>

__global__ void probe(double* worst) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // a low-discrepancy walk over four binades
  const double x = 0.5 * exp2(4.0 * (double)((i * 0x9E3779B97F4A7C15ull) >> 11) * 0x1.0p-53);
  const double exact = 1.0 / x;  // IEEE divide
  const double ulp = exp2((double)(ilogb(exact) - 52));
  double r0 = __builtin_amdgcn_rcp(x);
  double r1 = fma(fma(-x, r0, 1.0), r0, r0);
  double r2 = fma(fma(-x, r1, 1.0), r1, r1);
  double e[3] = {fabs(r0 - exact) / ulp, fabs(r1 - exact) / ulp, fabs(r2 - exact) / ulp};
  for (int k = 0; k < 3; k++) {
    // doubles that are non-negative order like their bit patterns
    atomicMax((unsigned long long*)&worst[k], (unsigned long long)__double_as_longlong(e[k]));
  }
}

int main() {
  double* d;
  hipMalloc(&d, 3 * sizeof(double));
  hipMemset(d, 0, 3 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1 << 18), dim3(256), 0, 0, d);
  double h[3];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("v_rcp_f64 worst error over 2^26 arguments in [0.5, 8): raw %.4g ulp, one Newton step %.4g ulp, two steps %.4g ulp\n", h[0], h[1], h[2]);
  return 0;
}
