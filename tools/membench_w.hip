// Pure-write patterns over a buffer of 65 536 x 100 800 bytes (the frames of a 65 536-env batch).
//   hipcc --offload-arch=gfx950 -O3 tools/membench_w.hip -o tools/membench_w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kFrame = 100800, kEnvs = 65536;
struct U3 { uint32_t x, y, z; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A: one workgroup per frame, 12-byte stores (the raster kernels' pattern)
__global__ void __launch_bounds__(256) w12_per_frame(uint8_t* p, uint32_t v) {
  uint8_t* f = p + (uint64_t)blockIdx.x * kFrame;
  for (int c = threadIdx.x; c < kFrame / 12; c += 256) *reinterpret_cast<U3*>(f + (uint64_t)c * 12) = U3{v, v, v};
}
// B: one workgroup per frame, 16-byte stores
__global__ void __launch_bounds__(256) w16_per_frame(uint8_t* p, uint32_t v) {
  u32x4* f = reinterpret_cast<u32x4*>(p + (uint64_t)blockIdx.x * kFrame);
  for (int c = threadIdx.x; c < kFrame / 16; c += 256) f[c] = u32x4{v, v, v, v};
}
// C: grid-stride over the whole buffer, 16-byte stores
__global__ void __launch_bounds__(256) w16_grid(uint8_t* p, uint32_t v, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < n16; c += (uint64_t)gridDim.x * 256) f[c] = u32x4{v, v, v, v};
}
// D: one workgroup per frame, each lane writes 4 consecutive 16-byte vectors (64 bytes per lane)
__global__ void __launch_bounds__(256) w16x4_per_frame(uint8_t* p, uint32_t v) {
  u32x4* f = reinterpret_cast<u32x4*>(p + (uint64_t)blockIdx.x * kFrame);
  for (int c = threadIdx.x * 4; c < kFrame / 16; c += 256 * 4) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (c + k < kFrame / 16) f[c + k] = u32x4{v, v, v, v};
  }
}
// E: as A with 4 frames per workgroup
__global__ void __launch_bounds__(256) w12_four_frames(uint8_t* p, uint32_t v) {
  uint8_t* f = p + (uint64_t)blockIdx.x * 4 * kFrame;
  for (int c = threadIdx.x; c < 4 * kFrame / 12; c += 256) *reinterpret_cast<U3*>(f + (uint64_t)c * 12) = U3{v, v, v};
}
// F: grid-stride over the whole buffer, 12-byte stores
__global__ void __launch_bounds__(256) w12_grid(uint8_t* p, uint32_t v, uint64_t n12) {
  for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < n12; c += (uint64_t)gridDim.x * 256)
    *reinterpret_cast<U3*>(p + c * 12) = U3{v, v, v};
}
// G: one workgroup per frame, 12-byte stores, 512 threads
__global__ void __launch_bounds__(512) w12_per_frame_512(uint8_t* p, uint32_t v) {
  uint8_t* f = p + (uint64_t)blockIdx.x * kFrame;
  for (int c = threadIdx.x; c < kFrame / 12; c += 512) *reinterpret_cast<U3*>(f + (uint64_t)c * 12) = U3{v, v, v};
}

// H: no loop: a workgroup writes U x 4 KiB with U unrolled 16-byte stores per lane (the shape of a
// library fill kernel)
template <int U>
__global__ void __launch_bounds__(256) w16_chunk(uint8_t* p, uint32_t v, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  const uint64_t base = (uint64_t)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
  for (int k = 0; k < U; k++)
    if (base + (uint64_t)k * 256 < n16) f[base + (uint64_t)k * 256] = u32x4{v, v, v, v};
}
// J: one workgroup per frame, 12-byte stores, four per lane per trip
__global__ void __launch_bounds__(256) w12_per_frame_x4(uint8_t* p, uint32_t v) {
  uint8_t* f = p + (uint64_t)blockIdx.x * kFrame;
  for (int c = threadIdx.x; c < kFrame / 12; c += 1024) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (c + k * 256 < kFrame / 12) *reinterpret_cast<U3*>(f + (uint64_t)(c + k * 256) * 12) = U3{v, v, v};
  }
}
// K: 12-byte stores, no loop: a workgroup writes 256 x U groups
template <int U>
__global__ void __launch_bounds__(256) w12_chunk(uint8_t* p, uint32_t v, uint64_t n12) {
  const uint64_t base = (uint64_t)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
  for (int k = 0; k < U; k++)
    if (base + (uint64_t)k * 256 < n12) *reinterpret_cast<U3*>(p + (base + (uint64_t)k * 256) * 12) = U3{v, v, v};
}

// L: no loop, every wave writes U consecutive KiB (lane-contiguous within each store)
template <int U>
__global__ void __launch_bounds__(256) w16_wave_chunk(uint8_t* p, uint32_t v, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t base = wave * 64 * U + (threadIdx.x & 63);
#pragma unroll
  for (int k = 0; k < U; k++)
    if (base + (uint64_t)k * 64 < n16) f[base + (uint64_t)k * 64] = u32x4{v, v, v, v};
}
// N: H x1 shifted by `shift` bytes (alignment sensitivity)
__global__ void __launch_bounds__(256) w16_chunk_shift(uint8_t* p, uint32_t v, uint64_t n16, uint64_t shift) {
  u32x4* f = reinterpret_cast<u32x4*>(p + shift);
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i + 1024 < n16) f[i] = u32x4{v, v, v, v};
}
// Q: H x1 with the state of the chunk's frame read first (nine cached loads per lane, as a raster would)
__global__ void __launch_bounds__(256) w16_chunk_state(uint8_t* p, const double* st, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const uint64_t env = i / 6300;
  double a = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) a += st[(uint64_t)k * kEnvs + env];
  const uint32_t v = (uint32_t)(int)a;
  f[i] = u32x4{v, v, v, v};
}
// R: one workgroup of 64 lanes per 1 KiB, no loop
__global__ void __launch_bounds__(64) w16_chunk_64(uint8_t* p, uint32_t v, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i < n16) f[i] = u32x4{v, v, v, v};
}

// Q2: as Q, but the state comes through scalar loads: a wave's 64 chunks touch at most two frames
template <int THREADS>
__global__ void __launch_bounds__(THREADS) w16_chunk_state_scalar(uint8_t* p, const double* st, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  const uint64_t i = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
  const uint64_t i0 = i & ~63ull;
  const uint32_t e0 = __builtin_amdgcn_readfirstlane((uint32_t)(i0 / 6300));
  const uint32_t e1 = __builtin_amdgcn_readfirstlane((uint32_t)((i0 + 63) / 6300));
  double a0 = 0, a1 = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { a0 += st[(uint64_t)k * kEnvs + e0]; a1 += st[(uint64_t)k * kEnvs + (e1 < kEnvs ? e1 : e0)]; }
  if (i >= n16) return;
  const uint32_t v = (uint32_t)(int)((uint32_t)(i / 6300) == e0 ? a0 : a1);
  f[i] = u32x4{v, v, v, v};
}
// S: no loop, one store per lane, bigger workgroups
template <int THREADS>
__global__ void __launch_bounds__(THREADS) w16_chunk_big(uint8_t* p, uint32_t v, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  const uint64_t i = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
  if (i < n16) f[i] = u32x4{v, v, v, v};
}

// T: the shape a raster could take: 4 KiB-aligned chunk per workgroup, one 16-byte store per lane,
// the (at most two) frames of a wave described by 64-byte records fetched with scalar loads, and
// some dependent integer work per lane
struct Rec { int32_t v[16]; };
template <int WORK>
__global__ void __launch_bounds__(256) w16_chunk_records(uint8_t* p, const Rec* recs, uint64_t n16) {
  u32x4* f = reinterpret_cast<u32x4*>(p);
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t i0 = i & ~63ull;
  const uint32_t e0 = __builtin_amdgcn_readfirstlane((uint32_t)(i0 / 6300));
  uint32_t e1 = __builtin_amdgcn_readfirstlane((uint32_t)((i0 + 63) / 6300));
  e1 = e1 < kEnvs ? e1 : e0;
  const Rec r0 = recs[e0], r1 = recs[e1];
  if (i >= n16) return;
  const uint32_t env = (uint32_t)(i / 6300), c = (uint32_t)(i - (uint64_t)env * 6300);
  const bool first = env == e0;
  uint32_t acc = c;
#pragma unroll
  for (int k = 0; k < 16; k++) acc = acc * 33u + (uint32_t)(first ? r0.v[k] : r1.v[k]);
#pragma unroll
  for (int k = 0; k < WORK; k++) acc = (acc ^ (acc >> 7)) * 0x9E3779B1u + k;
  f[i] = u32x4{acc, acc + 1, acc + 2, acc + 3};
}

// U: per-frame workgroups with every frame starting on a 4 KiB boundary (pitch 102 400)
constexpr int kPitch = 102400;
__global__ void __launch_bounds__(256) w16_per_frame_pitched(uint8_t* p, uint32_t v) {
  u32x4* f = reinterpret_cast<u32x4*>(p + (uint64_t)blockIdx.x * kPitch);
  for (int c = threadIdx.x; c < kFrame / 16; c += 256) f[c] = u32x4{v, v, v, v};
}
__global__ void __launch_bounds__(256) w12_per_frame_pitched(uint8_t* p, uint32_t v) {
  uint8_t* f = p + (uint64_t)blockIdx.x * kPitch;
  for (int c = threadIdx.x; c < kFrame / 12; c += 256) *reinterpret_cast<U3*>(f + (uint64_t)c * 12) = U3{v, v, v};
}

template <class F>
static void run(const char* name, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  std::printf("%-22s %.3f ms  %.0f GB/s\n", name, ms, (double)kFrame * kEnvs / (ms * 1e-3) / 1e9);
}

int main() {
  uint8_t* p;
  const uint64_t n = (uint64_t)kFrame * kEnvs;
  if (hipMalloc(&p, n) != hipSuccess) return 1;
  run("hipMemsetAsync", [&] { hipMemsetAsync(p, 0, n, 0); });
  run("A w12 per frame", [&] { hipLaunchKernelGGL(w12_per_frame, dim3(kEnvs), dim3(256), 0, 0, p, 1u); });
  run("B w16 per frame", [&] { hipLaunchKernelGGL(w16_per_frame, dim3(kEnvs), dim3(256), 0, 0, p, 2u); });
  run("C w16 grid 2048", [&] { hipLaunchKernelGGL(w16_grid, dim3(2048), dim3(256), 0, 0, p, 3u, n / 16); });
  run("C w16 grid 16384", [&] { hipLaunchKernelGGL(w16_grid, dim3(16384), dim3(256), 0, 0, p, 3u, n / 16); });
  run("D w16x4 per frame", [&] { hipLaunchKernelGGL(w16x4_per_frame, dim3(kEnvs), dim3(256), 0, 0, p, 4u); });
  run("E w12 four frames", [&] { hipLaunchKernelGGL(w12_four_frames, dim3(kEnvs / 4), dim3(256), 0, 0, p, 5u); });
  run("F w12 grid 2048", [&] { hipLaunchKernelGGL(w12_grid, dim3(2048), dim3(256), 0, 0, p, 6u, n / 12); });
  run("F w12 grid 16384", [&] { hipLaunchKernelGGL(w12_grid, dim3(16384), dim3(256), 0, 0, p, 6u, n / 12); });
  run("G w12 per frame 512", [&] { hipLaunchKernelGGL(w12_per_frame_512, dim3(kEnvs), dim3(512), 0, 0, p, 7u); });
  run("H w16 chunk x4", [&] { hipLaunchKernelGGL(w16_chunk<4>, dim3((unsigned)((n / 16 + 1023) / 1024)), dim3(256), 0, 0, p, 8u, n / 16); });
  run("H w16 chunk x8", [&] { hipLaunchKernelGGL(w16_chunk<8>, dim3((unsigned)((n / 16 + 2047) / 2048)), dim3(256), 0, 0, p, 8u, n / 16); });
  run("H w16 chunk x1", [&] { hipLaunchKernelGGL(w16_chunk<1>, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, 8u, n / 16); });
  run("J w12 per frame x4", [&] { hipLaunchKernelGGL(w12_per_frame_x4, dim3(kEnvs), dim3(256), 0, 0, p, 9u); });
  run("K w12 chunk x4", [&] { hipLaunchKernelGGL(w12_chunk<4>, dim3((unsigned)((n / 12 + 1023) / 1024)), dim3(256), 0, 0, p, 10u, n / 12); });
  run("K w12 chunk x1", [&] { hipLaunchKernelGGL(w12_chunk<1>, dim3((unsigned)((n / 12 + 255) / 256)), dim3(256), 0, 0, p, 10u, n / 12); });
  run("L w16 wave chunk x4", [&] { hipLaunchKernelGGL(w16_wave_chunk<4>, dim3((unsigned)((n / 16 + 1023) / 1024)), dim3(256), 0, 0, p, 8u, n / 16); });
  run("L w16 wave chunk x2", [&] { hipLaunchKernelGGL(w16_wave_chunk<2>, dim3((unsigned)((n / 16 + 511) / 512)), dim3(256), 0, 0, p, 8u, n / 16); });
  run("N shift 1600", [&] { hipLaunchKernelGGL(w16_chunk_shift, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, 8u, n / 16, 1600); });
  run("N shift 0", [&] { hipLaunchKernelGGL(w16_chunk_shift, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, 8u, n / 16, 0); });
  double* st;
  hipMalloc(&st, 9 * kEnvs * sizeof(double));
  hipMemset(st, 0, 9 * kEnvs * sizeof(double));
  run("Q w16 chunk + state", [&] { hipLaunchKernelGGL(w16_chunk_state, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, st, n / 16); });
  run("R w16 chunk 64 lanes", [&] { hipLaunchKernelGGL(w16_chunk_64, dim3((unsigned)((n / 16 + 63) / 64)), dim3(64), 0, 0, p, 8u, n / 16); });
  run("Q2 scalar state 256", [&] { hipLaunchKernelGGL(w16_chunk_state_scalar<256>, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, st, n / 16); });
  run("Q2 scalar state 1024", [&] { hipLaunchKernelGGL(w16_chunk_state_scalar<1024>, dim3((unsigned)((n / 16 + 1023) / 1024)), dim3(1024), 0, 0, p, st, n / 16); });
  run("S 512 threads", [&] { hipLaunchKernelGGL(w16_chunk_big<512>, dim3((unsigned)((n / 16 + 511) / 512)), dim3(512), 0, 0, p, 8u, n / 16); });
  run("S 1024 threads", [&] { hipLaunchKernelGGL(w16_chunk_big<1024>, dim3((unsigned)((n / 16 + 1023) / 1024)), dim3(1024), 0, 0, p, 8u, n / 16); });
  Rec* recs;
  hipMalloc(&recs, kEnvs * sizeof(Rec));
  hipMemset(recs, 0, kEnvs * sizeof(Rec));
  run("T records, work 0", [&] { hipLaunchKernelGGL(w16_chunk_records<0>, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, recs, n / 16); });
  run("T records, work 10", [&] { hipLaunchKernelGGL(w16_chunk_records<10>, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, recs, n / 16); });
  run("T records, work 25", [&] { hipLaunchKernelGGL(w16_chunk_records<25>, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, p, recs, n / 16); });
  hipFree(recs);
  uint8_t* pp;
  if (hipMalloc(&pp, (uint64_t)kPitch * kEnvs) == hipSuccess) {
    run("U w16 per frame pitched", [&] { hipLaunchKernelGGL(w16_per_frame_pitched, dim3(kEnvs), dim3(256), 0, 0, pp, 1u); });
    run("U w12 per frame pitched", [&] { hipLaunchKernelGGL(w12_per_frame_pitched, dim3(kEnvs), dim3(256), 0, 0, pp, 1u); });
    hipFree(pp);
  }
  hipFree(st);
  hipFree(p);
  return 0;
}
