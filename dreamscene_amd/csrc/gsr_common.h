// gsr_common.h -- shared device helpers for the gfx950 (CDNA4, wave64) rasterizer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsrast.h"

// entries per checkpoint / backward work item (GsrBinning.seg_len; 0 = 256)
static inline uint32_t gsr_seg_len(const GsrBinning& b) { return (b.seg_len == 64u || b.seg_len == 128u) ? b.seg_len : 256u; }

#define GSR_WAVE 64
#define GSR_NEAR_Z 0.2f
#define GSR_ALPHA_MIN (1.0f / 255.0f)
#define GSR_ALPHA_MAX 0.99f
#define GSR_T_MIN 0.0001f
#define GSR_LOWPASS 0.3f

// Thread-local last HIP error (api.hip owns the definition).
extern thread_local int gsr_tls_hip_error;

#define GSR_HIP(call)                          \
  do {                                         \
    hipError_t _e = (call);                    \
    if (_e != hipSuccess) {                    \
      gsr_tls_hip_error = (int)_e;             \
      return GSR_EHIP;                         \
    }                                          \
  } while (0)

// Every entry point runs on the device that owns the caller's buffers, whatever the calling thread's current device is
// (hipSetDevice from the pointer attributes per call); the previous device is restored on return.
struct GsrDeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit GsrDeviceGuard(const void* device_ptr) {
    hipPointerAttribute_t a;
    if (!device_ptr || hipPointerGetAttributes(&a, device_ptr) != hipSuccess) {
      (void)hipGetLastError();   // not a pointer the runtime knows: leave the current device alone
      return;
    }
    if (hipGetDevice(&prev) != hipSuccess || a.device == prev) return;
    changed = hipSetDevice(a.device) == hipSuccess;
  }
  ~GsrDeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
  GsrDeviceGuard(const GsrDeviceGuard&) = delete;
  GsrDeviceGuard& operator=(const GsrDeviceGuard&) = delete;
};

struct GsrProfile {
  static constexpr int kMax = 4096;
  hipEvent_t ev[kMax][2];
  int stage[kMax];
  int n = 0;
  int created = 0;
  uint32_t mask = 0xFFFFFFFFu;
  uint32_t every = 1;                       // record one of every `every` occurrences of a stage ...
  uint32_t tick[GSR_STAGE_COUNT] = {};      // ... counted per stage
};

// RAII-free stage bracket: records start/stop events on `stream` if profiling is on.
struct GsrStageTimer {
  GsrProfile* p;
  hipStream_t s;
  int slot;
  GsrStageTimer(GsrProfile* prof, hipStream_t stream, int stage) : p(prof), s(stream), slot(-1) {
    if (!p || p->n >= GsrProfile::kMax || !((p->mask >> stage) & 1u)) return;
    if (p->every > 1 && (p->tick[stage]++ % p->every) != 0) return;
    slot = p->n++;
    if (slot >= p->created) {
      (void)hipEventCreate(&p->ev[slot][0]);
      (void)hipEventCreate(&p->ev[slot][1]);
      p->created = slot + 1;
    }
    p->stage[slot] = stage;
    (void)hipEventRecord(p->ev[slot][0], s);
  }
  void stop() {
    if (slot >= 0) (void)hipEventRecord(p->ev[slot][1], s);
    slot = -1;
  }
  ~GsrStageTimer() { stop(); }
};

// ---- clearing device memory with a KERNEL, never hipMemsetAsync: a captured step (hipGraph) whose first node is a memset
// node was observed to start before the work enqueued ahead of the graph launch had finished (ROCm 7.2: the backward
// graph ran into the tail of the forward's compositing kernel; tests/test_graph.py). Plain kernel nodes keep stream order.
namespace {
__global__ void __launch_bounds__(256) k_gsr_zero16(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
// `rows` blocks of `words` 32-bit words, `pitch` bytes apart (any 4-byte alignment)
__global__ void __launch_bounds__(256) k_gsr_zero_words(uint32_t* __restrict__ p, size_t pitch, uint32_t words) {
  uint32_t* row = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(p) + (size_t)blockIdx.y * pitch);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) row[i] = 0u;
}
}  // namespace
// bytes: a multiple of 4. 16-byte aligned regions of a multiple of 16 bytes go through 16-byte stores.
static inline hipError_t gsr_zero_async(void* p, size_t bytes, hipStream_t stream, size_t pitch = 0, uint32_t rows = 1) {
  if (bytes == 0 || rows == 0) return hipSuccess;
  if (rows == 1 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (bytes & 15u) == 0) {
    const size_t n16 = bytes / 16;
    const uint32_t grid = (uint32_t)((n16 + 255) / 256 < 8192 ? (n16 + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_gsr_zero16, dim3(grid), dim3(256), 0, stream, reinterpret_cast<uint4*>(p), n16);
  } else {
    const uint32_t words = (uint32_t)(bytes / 4);
    const uint32_t gx = (words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024;
    hipLaunchKernelGGL(k_gsr_zero_words, dim3(gx, rows), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(p), pitch, words);
  }
  return hipGetLastError();
}

// float -> int32: truncating, saturating, NaN -> 0 (SEMANTICS.md; identical to the C oracle's f2i_sat).
__device__ __forceinline__ int32_t gsr_f2i_sat(float x) {
  if (!(x == x)) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)x;
}

// The same conversion as ONE instruction: v_cvt_i32_f32 truncates, saturates and maps NaN to 0 by definition (a C++
// cast is undefined out of range, so the instruction is named explicitly).
__device__ __forceinline__ int32_t gsr_f2i_sat_fast(float x) {
  int32_t i;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(x));
  return i;
}

// ---- wave64 cross-lane reductions on DPP (no LDS traffic) ---------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = true>
__device__ __forceinline__ float gsr_dpp(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}

// Sum over the 64 lanes; the total is valid in lane 63.
__device__ __forceinline__ float gsr_wave_sum_to_lane63(float v) {
  v += gsr_dpp<0xB1>(v);                     // quad_perm [1,0,3,2]
  v += gsr_dpp<0x4E>(v);                     // quad_perm [2,3,0,1]
  v += gsr_dpp<0x141>(v);                    // row_half_mirror
  v += gsr_dpp<0x140>(v);                    // row_mirror  -> every lane holds its row's sum
  v += gsr_dpp<0x142, 0xA, 0xF, false>(v);   // row_bcast:15 into rows 1,3
  v += gsr_dpp<0x143, 0xC, 0xF, false>(v);   // row_bcast:31 into rows 2,3
  return v;
}

__device__ __forceinline__ uint32_t gsr_wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
    v = v > t ? v : t;
  }
  return v;
}

__device__ __forceinline__ int gsr_lane() { return (int)__lane_id(); }

// ---- spherical harmonics (utils/sh_utils.py:25-102 in the reference) --------------------------------------
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
#define GSR_SH_C2_0 1.0925484305920792f
#define GSR_SH_C2_1 (-1.0925484305920792f)
#define GSR_SH_C2_2 0.31539156525252005f
#define GSR_SH_C2_3 (-1.0925484305920792f)
#define GSR_SH_C2_4 0.5462742152960396f
#define GSR_SH_C3_0 (-0.5900435899266435f)
#define GSR_SH_C3_1 2.890611442640554f
#define GSR_SH_C3_2 (-0.4570457994644658f)
#define GSR_SH_C3_3 0.3731763325901154f
#define GSR_SH_C3_4 (-0.4570457994644658f)
#define GSR_SH_C3_5 1.445305721320277f
#define GSR_SH_C3_6 (-0.5900435899266435f)
