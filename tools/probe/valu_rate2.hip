// Second VALU issue-rate probe for gfx950 (see valu_rate.hip): selects, adds, integer ops, exec-masked moves, DPP with
// bank masks, LDS reads, and whether transcendental / permlane ops overlap with plain FMAs from other waves.
// All at 8 waves per SIMD (256 CUs x 8 blocks of 256 threads); prints wall ns per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define V8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum Kind { ADD = 0, SUB, FMAC, MAXF, MOV, AND, LSHL, ADDU, CVTFU, CND_VCC, CND_VCC_W, CND_SGPR, CND_E64_VCC, MOV_EXEC,
            DPP_BANK, DPP_BCTRL, MIX_FMA_EXP, MIX_FMA_SWAP, MIX_FMA_ADD, LDS_B128, LDS_B64, LDS_B32, SALU_MIX, MBCNT,
            READLANE, MULLEG, FMAMK, SWAP16, NKINDS };
static const char* kNames[NKINDS] = {
    "v_add_f32", "v_sub_f32", "v_fmac_f32", "v_max_f32", "v_mov_b32", "v_and_b32", "v_lshlrev_b32", "v_add_u32",
    "v_cvt_f32_u32", "v_cndmask vcc (vcc never written)", "v_cndmask vcc (vcc written once)", "v_cndmask_e64 sgpr pair",
    "v_cndmask_e64 vcc", "s_mov exec + v_mov + s_mov exec", "v_add_f32_dpp bank_mask:0x3", "v_add_f32_dpp bound_ctrl",
    "4 v_fma + 4 v_exp (per inst)", "4 v_fma + 4 permlane32_swap", "4 v_fma + 4 v_add_f32",
    "ds_read_b128 (broadcast addr)", "ds_read_b64 (broadcast addr)", "ds_read_b32 (broadcast addr)",
    "8 v_fma + 8 s_and_b64 (per v_fma)", "v_mbcnt_lo", "v_readlane_b32", "v_mul_legacy_f32", "v_fmamk_f32",
    "v_permlane16_swap"};

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters) {
  __shared__ float4 lds[512];
  lds[threadIdx.x] = make_float4(1, 2, 3, 4);
  lds[threadIdx.x + 256] = make_float4(1, 2, 3, 4);
  __syncthreads();
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 1e-4f;
  const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)(blockIdx.x & 1);
  const int addr = (blockIdx.x & 15) * 16;
  float4 q = make_float4(0, 0, 0, 0);
  float2 q2 = make_float2(0, 0);
  float q1 = 0;
  if constexpr (KIND == CND_VCC_W || KIND == CND_E64_VCC) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  for (int it = 0; it < iters; ++it) {
#define OP2(OP) REP16(asm volatile(OP " %0, %0, %8\n " OP " %1, %1, %8\n " OP " %2, %2, %8\n " OP " %3, %3, %8\n " OP " %4, %4, %8\n " OP " %5, %5, %8\n " OP " %6, %6, %8\n " OP " %7, %7, %8\n" : V8 : "v"(m));)
    if constexpr (KIND == ADD) { OP2("v_add_f32") }
    else if constexpr (KIND == SUB) { OP2("v_sub_f32") }
    else if constexpr (KIND == FMAC) { REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n" : V8 : "v"(m), "v"(c));) }
    else if constexpr (KIND == MAXF) { OP2("v_max_f32") }
    else if constexpr (KIND == MULLEG) { OP2("v_mul_legacy_f32") }
    else if constexpr (KIND == FMAMK) { REP16(asm volatile("v_fmamk_f32 %0, %0, 0x3aaddd0a, %8\n v_fmamk_f32 %1, %1, 0x3aaddd0a, %8\n v_fmamk_f32 %2, %2, 0x3aaddd0a, %8\n v_fmamk_f32 %3, %3, 0x3aaddd0a, %8\n v_fmamk_f32 %4, %4, 0x3aaddd0a, %8\n v_fmamk_f32 %5, %5, 0x3aaddd0a, %8\n v_fmamk_f32 %6, %6, 0x3aaddd0a, %8\n v_fmamk_f32 %7, %7, 0x3aaddd0a, %8\n" : V8 : "v"(m));) }
    else if constexpr (KIND == MOV) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" : V8);) }
    else if constexpr (KIND == AND) { OP2("v_and_b32") }
    else if constexpr (KIND == LSHL) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n" : V8);) }
    else if constexpr (KIND == ADDU) { OP2("v_add_u32") }
    else if constexpr (KIND == CVTFU) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n" : V8);) }
    else if constexpr (KIND == CND_VCC || KIND == CND_VCC_W) {
      REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n" : V8 : : "vcc");)
    } else if constexpr (KIND == CND_E64_VCC) {
      REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n"
                         "v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %5, %5, %6, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc\n v_cndmask_b32_e64 %7, %7, %0, vcc\n" : V8 : : "vcc");)
    } else if constexpr (KIND == CND_SGPR) {
      REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %8\n v_cndmask_b32_e64 %1, %1, %2, %8\n v_cndmask_b32_e64 %2, %2, %3, %8\n v_cndmask_b32_e64 %3, %3, %4, %8\n"
                         "v_cndmask_b32_e64 %4, %4, %5, %8\n v_cndmask_b32_e64 %5, %5, %6, %8\n v_cndmask_b32_e64 %6, %6, %7, %8\n v_cndmask_b32_e64 %7, %7, %0, %8\n" : V8 : "s"(mask));)
    } else if constexpr (KIND == MOV_EXEC) {
      // a select done as: exec <- mask; v_mov; exec <- all   (counts as ONE select per triple)
      REP16(asm volatile("s_mov_b64 exec, %8\n v_mov_b32 %0, %1\n s_mov_b64 exec, -1\n s_mov_b64 exec, %8\n v_mov_b32 %1, %2\n s_mov_b64 exec, -1\n"
                         "s_mov_b64 exec, %8\n v_mov_b32 %2, %3\n s_mov_b64 exec, -1\n s_mov_b64 exec, %8\n v_mov_b32 %3, %4\n s_mov_b64 exec, -1\n"
                         "s_mov_b64 exec, %8\n v_mov_b32 %4, %5\n s_mov_b64 exec, -1\n s_mov_b64 exec, %8\n v_mov_b32 %5, %6\n s_mov_b64 exec, -1\n"
                         "s_mov_b64 exec, %8\n v_mov_b32 %6, %7\n s_mov_b64 exec, -1\n s_mov_b64 exec, %8\n v_mov_b32 %7, %0\n s_mov_b64 exec, -1\n" : V8 : "s"(mask));)
    } else if constexpr (KIND == DPP_BANK) {
      REP16(asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %1, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                         "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %3, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                         "v_add_f32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %5, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                         "v_add_f32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %7, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n" : V8);)
    } else if constexpr (KIND == DPP_BCTRL) {
      REP16(asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" : V8);)
    } else if constexpr (KIND == MIX_FMA_EXP) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %1, %1, %8, %9\n v_exp_f32 %5, %5\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %6, %6\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %7, %7\n" : V8 : "v"(m), "v"(c));)
    } else if constexpr (KIND == MIX_FMA_SWAP) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_permlane32_swap_b32 %4, %5\n v_fma_f32 %1, %1, %8, %9\n v_permlane32_swap_b32 %6, %7\n v_fma_f32 %2, %2, %8, %9\n v_permlane32_swap_b32 %5, %6\n v_fma_f32 %3, %3, %8, %9\n v_permlane32_swap_b32 %7, %4\n" : V8 : "v"(m), "v"(c));)
    } else if constexpr (KIND == MIX_FMA_ADD) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_add_f32 %4, %4, %8\n v_fma_f32 %1, %1, %8, %9\n v_add_f32 %5, %5, %8\n v_fma_f32 %2, %2, %8, %9\n v_add_f32 %6, %6, %8\n v_fma_f32 %3, %3, %8, %9\n v_add_f32 %7, %7, %8\n" : V8 : "v"(m), "v"(c));)
    } else if constexpr (KIND == LDS_B128) {
      REP16(asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:256\n ds_read_b128 %0, %1 offset:512\n"
                         "ds_read_b128 %0, %1 offset:1024\n ds_read_b128 %0, %1 offset:2048\n ds_read_b128 %0, %1 offset:3072\n ds_read_b128 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)\n" : "=v"(q) : "v"(addr) : "memory");)
    } else if constexpr (KIND == LDS_B64) {
      REP16(asm volatile("ds_read_b64 %0, %1\n ds_read_b64 %0, %1 offset:4096\n ds_read_b64 %0, %1 offset:256\n ds_read_b64 %0, %1 offset:512\n"
                         "ds_read_b64 %0, %1 offset:1024\n ds_read_b64 %0, %1 offset:2048\n ds_read_b64 %0, %1 offset:3072\n ds_read_b64 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)\n" : "=v"(q2) : "v"(addr) : "memory");)
    } else if constexpr (KIND == LDS_B32) {
      REP16(asm volatile("ds_read_b32 %0, %1\n ds_read_b32 %0, %1 offset:4096\n ds_read_b32 %0, %1 offset:256\n ds_read_b32 %0, %1 offset:512\n"
                         "ds_read_b32 %0, %1 offset:1024\n ds_read_b32 %0, %1 offset:2048\n ds_read_b32 %0, %1 offset:3072\n ds_read_b32 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)\n" : "=v"(q1) : "v"(addr) : "memory");)
    } else if constexpr (KIND == SALU_MIX) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n s_and_b64 s[20:21], s[22:23], %10\n v_fma_f32 %1, %1, %8, %9\n s_or_b64 s[22:23], s[20:21], %10\n v_fma_f32 %2, %2, %8, %9\n s_and_b64 s[20:21], s[22:23], %10\n v_fma_f32 %3, %3, %8, %9\n s_or_b64 s[22:23], s[20:21], %10\n"
                         "v_fma_f32 %4, %4, %8, %9\n s_and_b64 s[20:21], s[22:23], %10\n v_fma_f32 %5, %5, %8, %9\n s_or_b64 s[22:23], s[20:21], %10\n v_fma_f32 %6, %6, %8, %9\n s_and_b64 s[20:21], s[22:23], %10\n v_fma_f32 %7, %7, %8, %9\n s_or_b64 s[22:23], s[20:21], %10\n"
                         : V8 : "v"(m), "v"(c), "s"(mask) : "s20", "s21", "s22", "s23", "scc");)
    } else if constexpr (KIND == MBCNT) {
      REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_lo_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_lo_u32_b32 %3, -1, %3\n v_mbcnt_lo_u32_b32 %4, -1, %4\n v_mbcnt_lo_u32_b32 %5, -1, %5\n v_mbcnt_lo_u32_b32 %6, -1, %6\n v_mbcnt_lo_u32_b32 %7, -1, %7\n" : V8);)
    } else if constexpr (KIND == READLANE) {
      REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 3\n" : V8 : : "s20", "s21", "s22", "s23");)
    } else if constexpr (KIND == SWAP16) {
      REP16(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                         "v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %4\n v_permlane16_swap_b32 %5, %6\n v_permlane16_swap_b32 %7, %0\n" : V8);)
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q.x + q.y + q.z + q.w + q2.x + q2.y + q1;
  if (s == 123.456f) out[0] = s;
}

template <int KIND>
static void run(float* out, int cus, double insts_per_rep = 8) {
  const int iters = 100;
  const double insts_per_wave = (double)iters * 16 * insts_per_rep;
  const int wps = 8, blocks = cus * wps;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %7.3f ns per instruction per SIMD  (kernel %.1f us)\n", kNames[KIND], (double)ms * 1e6 / (insts_per_wave * wps), ms * 1e3);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs; 8 waves per SIMD\n", p.gcnArchName, p.multiProcessorCount);
  float* out;
  (void)hipMalloc(&out, 64);
  const int cus = p.multiProcessorCount;
  run<ADD>(out, cus); run<SUB>(out, cus); run<FMAC>(out, cus); run<FMAMK>(out, cus); run<MULLEG>(out, cus); run<MAXF>(out, cus);
  run<MOV>(out, cus); run<AND>(out, cus); run<LSHL>(out, cus); run<ADDU>(out, cus); run<CVTFU>(out, cus);
  run<CND_VCC>(out, cus); run<CND_VCC_W>(out, cus); run<CND_E64_VCC>(out, cus); run<CND_SGPR>(out, cus);
  run<MOV_EXEC>(out, cus); run<DPP_BANK>(out, cus); run<DPP_BCTRL>(out, cus);
  run<MIX_FMA_EXP>(out, cus); run<MIX_FMA_SWAP>(out, cus); run<MIX_FMA_ADD>(out, cus);
  run<LDS_B128>(out, cus); run<LDS_B64>(out, cus); run<LDS_B32>(out, cus); run<SALU_MIX>(out, cus);
  run<MBCNT>(out, cus); run<READLANE>(out, cus); run<SWAP16>(out, cus);
  return 0;
}
