// VALU issue-rate probe for gfx950: how many shader cycles one SIMD spends per wave64 instruction, for the
// instruction kinds the compositing kernels (render.hip) are made of, at 1 / 2 / 4 / 8 waves per SIMD.
// Answers the design question "does v_pk_fma_f32 (two pixels per lane) buy anything on this part?".
//   build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip ; run on the GPU box: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

enum Kind { FMA = 0, PKFMA, MUL, PKMUL, PKADD, EXP, RCP, LDEXP, RNDNE, CVT, MED3, CNDMASK, CMP, DPPADD, DPPMOV,
            PERMSWAP, MINU, FMA_DEP, PKFMA_DEP, NKINDS };
static const char* kNames[NKINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                     "v_exp_f32", "v_rcp_f32", "v_ldexp_f32", "v_rndne_f32", "v_cvt_i32_f32",
                                     "v_med3_f32", "v_cndmask_b32", "v_cmp_le_f32(sgpr)", "v_add_f32 dpp quad_perm",
                                     "v_mov_b32 dpp row_ror", "v_permlane32_swap", "v_min_u32", "v_fma_f32 dependent",
                                     "v_pk_fma_f32 dependent"};

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, uint64_t* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float m = 0.999f, c = 1e-4f;
  const f2 pm = {m, m}, pc = {c, c};
  int i0 = threadIdx.x;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == FMA) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if constexpr (KIND == FMA_DEP) {
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));)
    } else if constexpr (KIND == PKFMA) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
    } else if constexpr (KIND == PKFMA_DEP) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(p0) : "v"(pm), "v"(pc));)
    } else if constexpr (KIND == PKMUL) {
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));)
    } else if constexpr (KIND == PKADD) {
      REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));)
    } else if constexpr (KIND == DPPADD) {
      REP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == DPPMOV) {
      REP16(asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == PERMSWAP) {
      REP16(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == CMP) {
      REP16(asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n v_cmp_le_f32 s[22:23], %1, %2\n v_cmp_le_f32 s[24:25], %2, %3\n v_cmp_le_f32 s[26:27], %3, %4\n"
                         "v_cmp_le_f32 s[20:21], %4, %5\n v_cmp_le_f32 s[22:23], %5, %6\n v_cmp_le_f32 s[24:25], %6, %7\n v_cmp_le_f32 s[26:27], %7, %0\n"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if constexpr (KIND == CNDMASK) {
      REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
    } else if constexpr (KIND == LDEXP) {
      REP16(asm volatile("v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n"
                         "v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0 & 1));)
    } else if constexpr (KIND == MED3) {
      REP16(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                         "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else {
#define ONEOP(OP)                                                                                                       \
  REP16(asm volatile(OP " %0, %0\n " OP " %1, %1\n " OP " %2, %2\n " OP " %3, %3\n " OP " %4, %4\n " OP " %5, %5\n " OP   \
                        " %6, %6\n " OP " %7, %7\n"                                                                     \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
      if constexpr (KIND == EXP) { ONEOP("v_exp_f32") }
      else if constexpr (KIND == RCP) { ONEOP("v_rcp_f32") }
      else if constexpr (KIND == RNDNE) { ONEOP("v_rndne_f32") }
      else if constexpr (KIND == CVT) { ONEOP("v_cvt_i32_f32") }
      else if constexpr (KIND == MUL) {
        REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                           "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
      } else if constexpr (KIND == MINU) {
        REP16(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n"
                           "v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
            p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(float* out, uint64_t* cyc, int cus) {
  const int iters = 200;
  const double insts_per_wave = (double)iters * 16 * 8;
  printf("%-26s", kNames[KIND]);
  for (int wps : {1, 2, 4, 8}) {
    // blocks of 256 threads = one wave per SIMD; wps blocks per CU
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);   // warm
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= h.size();
    // cycle counter ticks per instruction per SIMD: a wave's elapsed ticks / its instructions / waves sharing the SIMD
    const double ticks_per_inst_simd = avg / insts_per_wave / wps;
    const double ns_per_inst_simd = (double)ms * 1e6 / (insts_per_wave * wps);
    printf("  w/SIMD=%d: %6.2f tick %6.3f ns", wps, ticks_per_inst_simd, ns_per_inst_simd);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  printf("columns: per wave64 instruction and SIMD -- cycle-counter ticks (s_memtime, 100 MHz?) and wall ns (whole kernel)\n");
  float* out; uint64_t* cyc;
  hipMalloc(&out, 64);
  hipMalloc(&cyc, 8 * 4 * 8 * 1024);
  const int cus = p.multiProcessorCount;
  run<FMA>(out, cyc, cus); run<PKFMA>(out, cyc, cus); run<MUL>(out, cyc, cus); run<PKMUL>(out, cyc, cus);
  run<PKADD>(out, cyc, cus); run<EXP>(out, cyc, cus); run<RCP>(out, cyc, cus); run<LDEXP>(out, cyc, cus);
  run<RNDNE>(out, cyc, cus); run<CVT>(out, cyc, cus); run<MED3>(out, cyc, cus); run<CNDMASK>(out, cyc, cus);
  run<CMP>(out, cyc, cus); run<DPPADD>(out, cyc, cus); run<DPPMOV>(out, cyc, cus); run<PERMSWAP>(out, cyc, cus);
  run<MINU>(out, cyc, cus); run<FMA_DEP>(out, cyc, cus); run<PKFMA_DEP>(out, cyc, cus);
  return 0;
}
