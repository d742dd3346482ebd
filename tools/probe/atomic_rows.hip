// Atomic-footprint probe for gfx950 (round 6): how much does ONE global_atomic_add_f64 instruction per loop iteration cost an
// issue-bound loop when it commits to 1 row (today's K7: 12 lanes, one 128-byte row) or to 4 rows at once (a K7 whose wave
// holds 4x4 pixels x 4 list entries: 40 lanes, four rows)?  Filler = F8 groups of 8 independent v_fma_f32 per iteration.
// 8 waves per SIMD (2048 workgroups of 256 threads on 256 CUs); prints ns per iteration per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/probe/atomic_rows.hip -o /tmp/atomic_rows && /tmp/atomic_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define V8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum Pat { NONE = 0, ROW1_F64_12, ROW4_F64_40, ROW4_F64_40_WG, ROW4_F32_40, ROW1_F64_10_HALF, ROW2_F64_20, NPAT };
static const char* kNames[NPAT] = {
    "no atomic", "1 row, 12 lanes f64 (today)", "4 rows, 40 lanes f64, rows per wave", "4 rows, 40 lanes f64, rows per workgroup",
    "4 rows, 40 lanes f32 (64-B rows)", "1 row, 10 lanes f64, every 2nd iteration", "2 rows, 20 lanes f64"};

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int PAT, int F8>
__global__ void __launch_bounds__(256) k_probe(float* out, double* rows64, float* rows32, uint32_t P, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 1e-4f;
  const int lane = threadIdx.x & 63;
  const uint32_t wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t key = (PAT == ROW4_F64_40_WG) ? blockIdx.x : wave_id;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < F8; ++f)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : V8 : "v"(m), "v"(c));
    if constexpr (PAT == ROW1_F64_12) {
      const uint32_t row = hash32(key * 7919u + (uint32_t)it) % P;
      if (lane < 12) unsafeAtomicAdd(rows64 + 16 * (size_t)row + lane, (double)a0);
    } else if constexpr (PAT == ROW4_F64_40 || PAT == ROW4_F64_40_WG) {
      const uint32_t row = hash32(key * 7919u + (uint32_t)it * 4u + (uint32_t)(lane >> 4)) % P;
      if ((lane & 15) < 10) unsafeAtomicAdd(rows64 + 16 * (size_t)row + (lane & 15), (double)a0);
    } else if constexpr (PAT == ROW4_F32_40) {
      const uint32_t row = hash32(key * 7919u + (uint32_t)it * 4u + (uint32_t)(lane >> 4)) % P;
      if ((lane & 15) < 10) unsafeAtomicAdd(rows32 + 16 * (size_t)row + (lane & 15), a0);
    } else if constexpr (PAT == ROW1_F64_10_HALF) {
      const uint32_t row = hash32(key * 7919u + (uint32_t)it) % P;
      if ((it & 1) && lane < 10) unsafeAtomicAdd(rows64 + 16 * (size_t)row + lane, (double)a0);
    } else if constexpr (PAT == ROW2_F64_20) {
      const uint32_t row = hash32(key * 7919u + (uint32_t)it * 2u + (uint32_t)(lane >> 4)) % P;
      if (lane < 32 && (lane & 15) < 10) unsafeAtomicAdd(rows64 + 16 * (size_t)row + (lane & 15), (double)a0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int PAT, int F8>
static void run(float* out, double* r64, float* r32, uint32_t P) {
  const int iters = 1500, groups = 2048;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<PAT, F8>), dim3(groups), dim3(256), 0, 0, out, r64, r32, P, 100);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<PAT, F8>), dim3(groups), dim3(256), 0, 0, out, r64, r32, P, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // 8 waves per SIMD, each `iters` iterations: ns per iteration of one wave's turn on its SIMD = ms / (iters * 8)
  printf("F=%3d fma  %-44s %8.1f ns per iteration per SIMD (%.3f ms)\n", F8 * 8, kNames[PAT], best * 1e6f / (iters * 8.f), best);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int F8>
static void run_all(float* out, double* r64, float* r32, uint32_t P) {
  run<NONE, F8>(out, r64, r32, P);
  run<ROW1_F64_12, F8>(out, r64, r32, P);
  run<ROW2_F64_20, F8>(out, r64, r32, P);
  run<ROW4_F64_40, F8>(out, r64, r32, P);
  run<ROW4_F64_40_WG, F8>(out, r64, r32, P);
  run<ROW4_F32_40, F8>(out, r64, r32, P);
  run<ROW1_F64_10_HALF, F8>(out, r64, r32, P);
}

int main() {
  const uint32_t P = 500000;
  float* out; double* r64; float* r32;
  hipMalloc(&out, 2048 * 256 * 4);
  hipMalloc(&r64, (size_t)P * 16 * 8);
  hipMalloc(&r32, (size_t)P * 16 * 4);
  hipMemset(r64, 0, (size_t)P * 16 * 8);
  hipMemset(r32, 0, (size_t)P * 16 * 4);
  run_all<10>(out, r64, r32, P);
  run_all<14>(out, r64, r32, P);
  run_all<18>(out, r64, r32, P);
  run_all<24>(out, r64, r32, P);
  return 0;
}
