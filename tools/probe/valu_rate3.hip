// Third issue-rate probe for gfx950 (see valu_rate.hip): VOPC / VOP2 forms that go through VCC, their cost inside a
// stream of FMAs, s_nop, scalar loads of uniform records, single-lane / grouped LDS reads, float atomics.
// 8 waves per SIMD; wall ns per counted instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define V8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum Kind { CMP_E32 = 0, CMP_E64, MIX_CND32, MIX_CND64, MIX_CMP32, FMA8, SNOP0, SNOP1, SLOAD8, SLOAD4, LDS128_1LANE,
            LDS128_4GRP, LDS128_ALL, READFIRST, ATOMIC10, ADD3, LSHLADD, MINI32, MULDPP, FMA_SGPR, MIN_F32, MAX3, NKINDS };
static const char* kNames[NKINDS] = {
    "v_cmp_ge_f32_e32 vcc", "v_cmp_ge_f32_e64 sgpr", "7 v_fma + 1 v_cndmask_e32 vcc (per 8)", "7 v_fma + 1 v_cndmask_e64 sgpr (per 8)",
    "7 v_fma + 1 v_cmp_e32 vcc (per 8)", "8 v_fma (per 8)", "s_nop 0", "s_nop 1", "s_load_dwordx8 (L2-resident records)",
    "s_load_dwordx4 x3 (one 48-B record)", "ds_read_b128 exec = 1 lane", "ds_read_b128 4 distinct addresses", "ds_read_b128 64 distinct (conflict-free)",
    "v_readfirstlane_b32", "global_atomic_add_f32 (10 lanes, 48-B row per wave)", "v_add3_u32", "v_lshl_add_u32", "v_min_i32",
    "v_mul_f32_dpp quad_perm", "v_fma_f32 with an SGPR operand", "v_min_f32", "v_max3_f32"};

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, const float* __restrict__ recs, float* atom, int iters) {
  __shared__ float4 lds[512];
  lds[threadIdx.x] = make_float4(1, 2, 3, 4);
  lds[threadIdx.x + 256] = make_float4(1, 2, 3, 4);
  __syncthreads();
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 1e-4f;
  const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)(blockIdx.x & 1);
  const int lane = threadIdx.x & 63;
  const int addr1 = (blockIdx.x & 15) * 16;
  const int addr4 = ((lane & 3) * 64 + (blockIdx.x & 15)) * 16;
  const int addr64 = threadIdx.x * 16;
  float4 q = make_float4(0, 0, 0, 0);
  float sacc = 0.f;
  asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  const uint32_t wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == CMP_E32) {
      REP16(asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1\n v_cmp_ge_f32_e32 vcc, %1, %2\n v_cmp_ge_f32_e32 vcc, %2, %3\n v_cmp_ge_f32_e32 vcc, %3, %4\n v_cmp_ge_f32_e32 vcc, %4, %5\n v_cmp_ge_f32_e32 vcc, %5, %6\n v_cmp_ge_f32_e32 vcc, %6, %7\n v_cmp_ge_f32_e32 vcc, %7, %0\n" : V8 : : "vcc");)
    } else if constexpr (KIND == CMP_E64) {
      REP16(asm volatile("v_cmp_ge_f32_e64 s[20:21], %0, %1\n v_cmp_ge_f32_e64 s[22:23], %1, %2\n v_cmp_ge_f32_e64 s[20:21], %2, %3\n v_cmp_ge_f32_e64 s[22:23], %3, %4\n v_cmp_ge_f32_e64 s[20:21], %4, %5\n v_cmp_ge_f32_e64 s[22:23], %5, %6\n v_cmp_ge_f32_e64 s[20:21], %6, %7\n v_cmp_ge_f32_e64 s[22:23], %7, %0\n" : V8 : : "s20", "s21", "s22", "s23");)
    } else if constexpr (KIND == MIX_CND32) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_cndmask_b32_e32 %7, %7, %6, vcc\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n" : V8 : "v"(m), "v"(c) : "vcc");)
    } else if constexpr (KIND == MIX_CND64) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_cndmask_b32_e64 %7, %7, %6, %10\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n" : V8 : "v"(m), "v"(c), "s"(mask));)
    } else if constexpr (KIND == MIX_CMP32) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_cmp_ge_f32_e32 vcc, %7, %6\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n" : V8 : "v"(m), "v"(c) : "vcc");)
    } else if constexpr (KIND == FMA8 || KIND == FMA_SGPR) {
      if constexpr (KIND == FMA8) {
        REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : V8 : "v"(m), "v"(c));)
      } else {
        REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : V8 : "s"(m), "v"(c));)
      }
    } else if constexpr (KIND == SNOP0) {
      REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");)
    } else if constexpr (KIND == SNOP1) {
      REP16(asm volatile("s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n");)
    } else if constexpr (KIND == SLOAD8) {
      // 8 loads of 32 B from pseudo-random 48-byte records of a 24 MB table (uniform per wave), waited for together
      const float* base = recs;
      uint32_t r = (wave_id * 2654435761u + (uint32_t)it * 40503u) & 0x7FFFFu;
      REP16({
        const uint32_t o0 = __builtin_amdgcn_readfirstlane((r & 0x7FFFFu) * 48u); r = r * 1664525u + 1013904223u;
        const uint32_t o1 = __builtin_amdgcn_readfirstlane((r & 0x7FFFFu) * 48u); r = r * 1664525u + 1013904223u;
        const uint32_t o2 = __builtin_amdgcn_readfirstlane((r & 0x7FFFFu) * 48u); r = r * 1664525u + 1013904223u;
        const uint32_t o3 = __builtin_amdgcn_readfirstlane((r & 0x7FFFFu) * 48u); r = r * 1664525u + 1013904223u;
        float s0, s1, s2, s3;
        asm volatile("s_load_dwordx8 s[20:27], %4, %5\n s_load_dwordx8 s[28:35], %4, %6\n s_load_dwordx8 s[36:43], %4, %7\n s_load_dwordx8 s[44:51], %4, %8\n"
                     "s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, s20\n v_mov_b32 %1, s28\n v_mov_b32 %2, s36\n v_mov_b32 %3, s44\n"
                     : "=v"(s0), "=v"(s1), "=v"(s2), "=v"(s3) : "s"(base), "s"(o0), "s"(o1), "s"(o2), "s"(o3)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35",
                       "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "memory");
        sacc += s0 + s1 + s2 + s3;
      })
    } else if constexpr (KIND == SLOAD4) {
      const float* base = recs;
      uint32_t r = (wave_id * 2654435761u + (uint32_t)it * 40503u) & 0x7FFFFu;
      REP16({
        const uint32_t o0 = __builtin_amdgcn_readfirstlane((r & 0x7FFFFu) * 48u); r = r * 1664525u + 1013904223u;
        const uint32_t o1 = __builtin_amdgcn_readfirstlane((r & 0x7FFFFu) * 48u); r = r * 1664525u + 1013904223u;
        float s0, s1;
        asm volatile("s_load_dwordx4 s[20:23], %2, %3\n s_load_dwordx4 s[24:27], %2, %3 offset:16\n s_load_dwordx4 s[28:31], %2, %3 offset:32\n"
                     "s_load_dwordx4 s[32:35], %2, %4\n s_load_dwordx4 s[36:39], %2, %4 offset:16\n s_load_dwordx4 s[40:43], %2, %4 offset:32\n"
                     "s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, s20\n v_mov_b32 %1, s32\n"
                     : "=v"(s0), "=v"(s1) : "s"(base), "s"(o0), "s"(o1)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35",
                       "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "memory");
        sacc += s0 + s1;
      })
    } else if constexpr (KIND == LDS128_1LANE) {
      REP16(asm volatile("s_mov_b64 exec, 1\n ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:256\n ds_read_b128 %0, %1 offset:512\n"
                         "ds_read_b128 %0, %1 offset:1024\n ds_read_b128 %0, %1 offset:2048\n ds_read_b128 %0, %1 offset:3072\n ds_read_b128 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, -1\n" : "=v"(q) : "v"(addr1) : "memory");)
    } else if constexpr (KIND == LDS128_4GRP) {
      REP16(asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:256\n ds_read_b128 %0, %1 offset:512\n"
                         "ds_read_b128 %0, %1 offset:1024\n ds_read_b128 %0, %1 offset:2048\n ds_read_b128 %0, %1 offset:3072\n ds_read_b128 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)\n" : "=v"(q) : "v"(addr4) : "memory");)
    } else if constexpr (KIND == LDS128_ALL) {
      REP16(asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n"
                         "ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n s_waitcnt lgkmcnt(0)\n" : "=v"(q) : "v"(addr64) : "memory");)
    } else if constexpr (KIND == READFIRST) {
      REP16(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s22, %6\n v_readfirstlane_b32 s23, %7\n" : V8 : : "s20", "s21", "s22", "s23");)
    } else if constexpr (KIND == ATOMIC10) {
      // what K7 does per (splat, block): lanes 0..9 add to a 48-byte row picked pseudo-randomly in a 24 MB table
      uint32_t r = (wave_id * 2654435761u + (uint32_t)it * 40503u);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        r = r * 1664525u + 1013904223u;
        const uint32_t row = __builtin_amdgcn_readfirstlane(r >> 13) & 0x7FFFFu;
        if (lane < 10) unsafeAtomicAdd(atom + 12 * (size_t)row + lane, a0);
      }
    } else if constexpr (KIND == ADD3) {
      REP16(asm volatile("v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8\n" : V8 : "v"(m));)
    } else if constexpr (KIND == LSHLADD) {
      REP16(asm volatile("v_lshl_add_u32 %0, %0, 4, %8\n v_lshl_add_u32 %1, %1, 4, %8\n v_lshl_add_u32 %2, %2, 4, %8\n v_lshl_add_u32 %3, %3, 4, %8\n v_lshl_add_u32 %4, %4, 4, %8\n v_lshl_add_u32 %5, %5, 4, %8\n v_lshl_add_u32 %6, %6, 4, %8\n v_lshl_add_u32 %7, %7, 4, %8\n" : V8 : "v"(m));)
    } else if constexpr (KIND == MINI32) {
      REP16(asm volatile("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %8\n v_min_i32 %5, %5, %8\n v_min_i32 %6, %6, %8\n v_min_i32 %7, %7, %8\n" : V8 : "v"(m));)
    } else if constexpr (KIND == MIN_F32) {
      REP16(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n" : V8 : "v"(m));)
    } else if constexpr (KIND == MAX3) {
      REP16(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n" : V8 : "v"(m), "v"(c));)
    } else if constexpr (KIND == MULDPP) {
      REP16(asm volatile("v_mul_f32_dpp %0, %1, %1 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %1, %2, %2 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mul_f32_dpp %2, %3, %3 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %3, %4, %4 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mul_f32_dpp %4, %5, %5 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %5, %6, %6 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mul_f32_dpp %6, %7, %7 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %7, %0, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" : V8);)
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q.x + q.y + q.z + q.w + sacc;
  if (s == 123.456f) out[0] = s;
}

template <int KIND>
static void run(float* out, const float* recs, float* atom, int cus, double insts_per_rep = 8) {
  const int iters = (KIND == SLOAD8 || KIND == SLOAD4 || KIND == ATOMIC10) ? 10 : 100;
  const double insts_per_wave = (double)iters * 16 * insts_per_rep;
  const int wps = 8, blocks = cus * wps;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, recs, atom, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, recs, atom, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %7.3f ns per counted instruction per SIMD  (kernel %.1f us)\n", kNames[KIND], (double)ms * 1e6 / (insts_per_wave * wps), ms * 1e3);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs; 8 waves per SIMD\n", p.gcnArchName, p.multiProcessorCount);
  float *out, *recs, *atom;
  (void)hipMalloc(&out, 64);
  (void)hipMalloc(&recs, (size_t)(1 << 19) * 48 + 256);
  (void)hipMalloc(&atom, (size_t)(1 << 19) * 48 + 256);
  (void)hipMemset(recs, 0, (size_t)(1 << 19) * 48 + 256);
  (void)hipMemset(atom, 0, (size_t)(1 << 19) * 48 + 256);
  const int cus = p.multiProcessorCount;
  run<FMA8>(out, recs, atom, cus); run<FMA_SGPR>(out, recs, atom, cus);
  run<CMP_E32>(out, recs, atom, cus); run<CMP_E64>(out, recs, atom, cus);
  run<MIX_CND32>(out, recs, atom, cus); run<MIX_CND64>(out, recs, atom, cus); run<MIX_CMP32>(out, recs, atom, cus);
  run<SNOP0>(out, recs, atom, cus); run<SNOP1>(out, recs, atom, cus);
  run<SLOAD8>(out, recs, atom, cus, 4); run<SLOAD4>(out, recs, atom, cus, 2);
  run<LDS128_1LANE>(out, recs, atom, cus); run<LDS128_4GRP>(out, recs, atom, cus); run<LDS128_ALL>(out, recs, atom, cus);
  run<READFIRST>(out, recs, atom, cus); run<ATOMIC10>(out, recs, atom, cus, 1);
  run<ADD3>(out, recs, atom, cus); run<LSHLADD>(out, recs, atom, cus); run<MINI32>(out, recs, atom, cus);
  run<MULDPP>(out, recs, atom, cus); run<MIN_F32>(out, recs, atom, cus); run<MAX3>(out, recs, atom, cus);
  return 0;
}
