// preprocess.hip -- K1 (per-Gaussian EWA projection + SH colour) and K8 (its backward), gfx950.
//
// Replaces the preprocess stage of the rasterizer DreamScene imports (scene_gaussian.py:11-12); the math
// follows SEMANTICS.md / SURVEY.md Appendix A.1, A.3 and the Python statements the reference does hold:
// cov3D gs_renderer.py:124-172, SH utils/sh_utils.py:25-102, projection utils/graphics_utils.py:29-36.
//
// Both kernels are HBM-streaming (44 + 12K bytes in per Gaussian for K1; 276 in / 248 out for K8 at K=16).
// One thread per Gaussian. K1: every lane pulls its own 12K-byte SH row with 16-byte loads straight into registers (the
// SH stride is a template parameter; the rows of a wave are contiguous, so L1/TA serve the pieces of a line to successive
// loads -- measured faster than the LDS transpose of round 1, which cost 50 KB of LDS per block). K8: the SH rows are
// read the same way; dL/dSH is written back coalesced through an LDS transpose (direct row stores measured 25 % slower).
//
// This translation unit is built with -ffp-contract=off: every fp32 operator that feeds an integer artefact
// (depth bits, radius, tile rectangle) rounds exactly once, in the order written -- the same order as
// oracle/gsr_oracle.c -- which is what makes radii / tile counts / sort keys bit-exact against the oracle.
#include "gsr_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct ViewConst {
  float V[16];
  float PV[16];
  float cam[3];
};

__device__ __forceinline__ void load_view(const GsrView& v, ViewConst& c) {
#pragma unroll
  for (int i = 0; i < 16; ++i) { c.V[i] = v.viewmatrix[i]; c.PV[i] = v.projmatrix[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) c.cam[i] = v.campos[i];
}

// The constants of view vv inside a loop over the views of a batched launch: read through the CONSTANT address space, i.e.
// with scalar loads (s_load, counted by lgkmcnt). As ordinary global loads (the address is uniform but the compiler cannot
// prove that the kernel's own stores leave it alone) they are vector-memory operations, and on gfx9 those return IN ORDER
// with the vector-memory stores: the first load of view vv + 1 waited for the write acknowledgement of everything view vv
// had just stored -- one HBM write latency per view and wave in K1 and K8.
typedef const __attribute__((address_space(4))) float gsr_cfloat;
__device__ __forceinline__ gsr_cfloat* gsr_const(const float* p) { return (gsr_cfloat*)(uintptr_t)p; }
__device__ __forceinline__ void load_view_const(const float* viewmatrix, const float* projmatrix, const float* campos, ViewConst& c) {
  gsr_cfloat* V = gsr_const(viewmatrix);
  gsr_cfloat* PV = gsr_const(projmatrix);
  gsr_cfloat* cam = gsr_const(campos);
#pragma unroll
  for (int i = 0; i < 16; ++i) { c.V[i] = V[i]; c.PV[i] = PV[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) c.cam[i] = cam[i];
}

// ---- multi-model ("scene") input: GsrScene flattened for the kernels (passed by value in the kernel arguments).
// A workgroup never straddles two models: model m owns the workgroups [fblk[m], fblk[m+1]) and its Gaussians keep
// their place first[m] + row in the concatenated index space every other kernel works in.
struct SceneTab {
  int32_t n;
  int32_t first[GSR_MAX_MODELS + 1];
  int32_t fblk[GSR_MAX_MODELS + 1];
  const float* xyz[GSR_MAX_MODELS];
  const float* scaling[GSR_MAX_MODELS];
  const float* rotation[GSR_MAX_MODELS];
  const float* opacity[GSR_MAX_MODELS];
  const float* dc[GSR_MAX_MODELS];
  const float* rest[GSR_MAX_MODELS];
  const float* scale_noise;
  const float* sh_noise;
  float* scales_out;
  float* rotations_out;
  float* opacities_out;
};
struct SceneGradTab {
  const float* dL_dscales_out;
  float* xyz[GSR_MAX_MODELS];
  float* scaling[GSR_MAX_MODELS];
  float* rotation[GSR_MAX_MODELS];
  float* opacity[GSR_MAX_MODELS];
  float* dc[GSR_MAX_MODELS];
  float* rest[GSR_MAX_MODELS];
};
struct NoScene {};

// Which rows this thread / wave works on. Without a scene: row == concatenated index.
struct Rows {
  int m;               // model (0 without a scene)
  int64_t i;           // concatenated Gaussian index
  int64_t row;         // row inside the model's tensors
  int64_t wave_row;    // row of the wave's first lane
  int64_t wave_i;      // concatenated index of the wave's first lane
  int n_valid;         // rows of this wave that exist
  bool ok;             // this lane's row exists
};
template <bool SCENE, typename TAB>
__device__ __forceinline__ Rows resolve_rows(const TAB& sc, int P) {
  const int tid = threadIdx.x, wave = tid >> 6;
  Rows r;
  if constexpr (SCENE) {
    int m = 0;
    for (int k = 1; k < sc.n; ++k) m += ((int)blockIdx.x >= sc.fblk[k]) ? 1 : 0;
    const int64_t cnt = (int64_t)sc.first[m + 1] - sc.first[m];
    const int64_t b0 = ((int64_t)blockIdx.x - sc.fblk[m]) * 256;
    r.m = m;
    r.row = b0 + tid;
    r.wave_row = b0 + wave * 64;
    r.i = sc.first[m] + r.row;
    r.wave_i = sc.first[m] + r.wave_row;
    r.n_valid = (int)min((int64_t)64, max((int64_t)0, cnt - r.wave_row));
    r.ok = r.row < cnt;
  } else {
    r.m = 0;
    r.i = r.row = (int64_t)blockIdx.x * 256 + tid;
    r.wave_i = r.wave_row = (int64_t)blockIdx.x * 256 + wave * 64;
    r.n_valid = (int)min((int64_t)64, max((int64_t)0, (int64_t)P - r.wave_row));
    r.ok = r.i < P;
  }
  return r;
}

// The activations of GaussianModel (gs_renderer.py:464-488) and scene_render's augmentations (scene_gaussian.py:844-852)
constexpr float kSqrtPoint2 = 0.44721359549995793f;   // 0.2 ** 0.5
struct ActScale { float act, pre, out; };              // exp(raw); after the noise; after the clamp
__device__ __forceinline__ ActScale act_scale(float raw, bool noisy, float n) {
  ActScale a;
  a.act = expf(raw);
  a.pre = noisy ? a.act + n * ((kSqrtPoint2 * a.act) / 4.0f) : a.act;
  a.out = noisy ? fmaxf(a.pre, 0.0f) : a.act;
  return a;
}
__device__ __forceinline__ float act_quat_norm(const float4 q) {
  return fmaxf(sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w), 1e-12f);
}
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void quat_to_R(const float4 q, float R[9]) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  R[0] = 1.0f - 2.0f * (y * y + z * z);
  R[1] = 2.0f * (x * y - r * z);
  R[2] = 2.0f * (x * z + r * y);
  R[3] = 2.0f * (x * y + r * z);
  R[4] = 1.0f - 2.0f * (x * x + z * z);
  R[5] = 2.0f * (y * z - r * x);
  R[6] = 2.0f * (x * z - r * y);
  R[7] = 2.0f * (y * z + r * x);
  R[8] = 1.0f - 2.0f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from(const float s0, const float s1, const float s2, const float R[9],
                                           float c6[6]) {
  float L[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    L[3 * i + 0] = R[3 * i + 0] * s0;
    L[3 * i + 1] = R[3 * i + 1] * s1;
    L[3 * i + 2] = R[3 * i + 2] * s2;
  }
#define GSR_SIG(i, j) ((L[3 * i] * L[3 * j] + L[3 * i + 1] * L[3 * j + 1]) + L[3 * i + 2] * L[3 * j + 2])
  c6[0] = GSR_SIG(0, 0); c6[1] = GSR_SIG(0, 1); c6[2] = GSR_SIG(0, 2);
  c6[3] = GSR_SIG(1, 1); c6[4] = GSR_SIG(1, 2); c6[5] = GSR_SIG(2, 2);
#undef GSR_SIG
}

// The EWA chain shared by K1 and K8 (identical operator order => identical values in both).
struct Ewa {
  float tx, ty, tz, txc, tyc, J00, J02, J11, J12;
  float M0[3], M1[3], U0[3], U1[3];
  float ca, cb, cc, det;
  bool clx, cly;
};

__device__ __forceinline__ void ewa_forward(const ViewConst& vc, float px, float py, float pz, const float c6[6],
                                            float fx, float fy, float limx, float limy, Ewa& e) {
  const float* V = vc.V;
  e.tx = ((V[0] * px + V[4] * py) + V[8] * pz) + V[12];
  e.ty = ((V[1] * px + V[5] * py) + V[9] * pz) + V[13];
  e.tz = ((V[2] * px + V[6] * py) + V[10] * pz) + V[14];
  const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
  const float txz = e.tx / e.tz, tyz = e.ty / e.tz;
  e.clx = (txz < -limx) || (txz > limx);
  e.cly = (tyz < -limy) || (tyz > limy);
  e.txc = fminf(limx, fmaxf(-limx, txz)) * e.tz;
  e.tyc = fminf(limy, fmaxf(-limy, tyz)) * e.tz;
  e.J00 = fx / e.tz;
  e.J02 = -(fx * e.txc) / (e.tz * e.tz);
  e.J11 = fy / e.tz;
  e.J12 = -(fy * e.tyc) / (e.tz * e.tz);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    e.M0[r] = e.J00 * V[4 * r + 0] + e.J02 * V[4 * r + 2];
    e.M1[r] = e.J11 * V[4 * r + 1] + e.J12 * V[4 * r + 2];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    e.U0[j] = (e.M0[0] * S[j] + e.M0[1] * S[3 + j]) + e.M0[2] * S[6 + j];
    e.U1[j] = (e.M1[0] * S[j] + e.M1[1] * S[3 + j]) + e.M1[2] * S[6 + j];
  }
  e.ca = ((e.U0[0] * e.M0[0] + e.U0[1] * e.M0[1]) + e.U0[2] * e.M0[2]) + GSR_LOWPASS;
  e.cb = (e.U0[0] * e.M1[0] + e.U0[1] * e.M1[1]) + e.U0[2] * e.M1[2];
  e.cc = ((e.U1[0] * e.M1[0] + e.U1[1] * e.M1[1]) + e.U1[2] * e.M1[2]) + GSR_LOWPASS;
  e.det = e.ca * e.cc - e.cb * e.cb;
}

__device__ __forceinline__ void sh_basis(int D, float x, float y, float z, float b[16]) {
  b[0] = GSR_SH_C0;
  if (D > 0) {
    b[1] = -GSR_SH_C1 * y; b[2] = GSR_SH_C1 * z; b[3] = -GSR_SH_C1 * x;
    if (D > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = GSR_SH_C2_0 * xy; b[5] = GSR_SH_C2_1 * yz; b[6] = GSR_SH_C2_2 * (2.0f * zz - xx - yy);
      b[7] = GSR_SH_C2_3 * xz; b[8] = GSR_SH_C2_4 * (xx - yy);
      if (D > 2) {
        b[9] = GSR_SH_C3_0 * y * (3.0f * xx - yy);
        b[10] = GSR_SH_C3_1 * xy * z;
        b[11] = GSR_SH_C3_2 * y * (4.0f * zz - xx - yy);
        b[12] = GSR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        b[13] = GSR_SH_C3_4 * x * (4.0f * zz - xx - yy);
        b[14] = GSR_SH_C3_5 * z * (xx - yy);
        b[15] = GSR_SH_C3_6 * x * (xx - 3.0f * yy);
      }
    }
  }
}

// colour_c = sum_k b_k sh[k][c] accumulated in ascending k (the oracle's order), written so that every index into the
// register array b[] is a compile-time constant (a runtime-indexed register array costs s_set_gpr_idx round trips).
#define GSR_SH_BAND(K0, K1)                                   \
  _Pragma("unroll") for (int k = K0; k <= K1; ++k) {          \
    acc[0] = acc[0] + b[k] * sh[3 * k];                       \
    acc[1] = acc[1] + b[k] * sh[3 * k + 1];                   \
    acc[2] = acc[2] + b[k] * sh[3 * k + 2];                   \
  }
__device__ __forceinline__ void sh_colour(int D, const float* sh, const float b[16], float acc[3]) {
  acc[0] = b[0] * sh[0]; acc[1] = b[0] * sh[1]; acc[2] = b[0] * sh[2];
  if (D > 0) {
    GSR_SH_BAND(1, 3)
    if (D > 1) {
      GSR_SH_BAND(4, 8)
      if (D > 2) { GSR_SH_BAND(9, 15) }
    }
  }
}
// same, for a register row of exactly KN coefficients (bands beyond KN cannot be active: D is validated against K)
template <int KN>
__device__ __forceinline__ void sh_colour_n(int D, const float* sh, const float b[16], float acc[3]) {
  acc[0] = b[0] * sh[0]; acc[1] = b[0] * sh[1]; acc[2] = b[0] * sh[2];
  if constexpr (KN >= 4) {
    if (D > 0) {
      GSR_SH_BAND(1, 3)
      if constexpr (KN >= 9) {
        if (D > 1) {
          GSR_SH_BAND(4, 8)
          if constexpr (KN >= 16) {
            if (D > 2) { GSR_SH_BAND(9, 15) }
          }
        }
      }
    }
  }
}
#undef GSR_SH_BAND

// Row stride (in floats) of one Gaussian's SH block inside the LDS transpose buffer: odd => the 64 lanes of a
// wave reading "their" row element k hit 64 different banks pairs (ds_read_b32, 32-lane groups).
__host__ __device__ __forceinline__ int sh_lds_stride(int K) { return (3 * K) | 1; }

// Coalesced global -> LDS load of the wave's SH block. `vis` = ballot of lanes whose Gaussian needs its row.
template <int KT>
__device__ __forceinline__ void stage_sh_in(const float* __restrict__ shs, int64_t wave_first, int n_valid, int K,
                                            unsigned long long vis, float* lds_wave) {
  const int F = KT > 0 ? 3 * KT : 3 * K;          // compile-time for the common strides: / and % become mul-shift
  const int stride = F | 1;
  const int total = n_valid * F;                                   // floats in the wave's block
  const float* src = shs + wave_first * (int64_t)F;
  const int lane = gsr_lane();
  for (int q = lane * 4; q < total; q += 64 * 4) {
    const int g0 = q / F, g1 = (q + 3) / F;
    const bool need = ((vis >> g0) & 1ull) || ((g1 < 64) && ((vis >> g1) & 1ull));
    if (!need) continue;
    float4 v;
    if (q + 3 < total) {
      v = *reinterpret_cast<const float4*>(src + q);
    } else {
      v.x = src[q];
      v.y = (q + 1 < total) ? src[q + 1] : 0.f;
      v.z = (q + 2 < total) ? src[q + 2] : 0.f;
      v.w = 0.f;
    }
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = q + k;
      if (f < total) {
        const int g = f / F, o = f - g * F;
        lds_wave[g * stride + o] = e[k];
      }
    }
  }
}

// Scene input: rows of F floats (features_dc: 3, features_rest: 3K-3) land at float offset o0 of the lanes' LDS rows.
__device__ __forceinline__ void stage_rows_in(const float* __restrict__ src, int F, int o0, int lds_stride, int n_valid,
                                              unsigned long long vis, float* lds_wave) {
  const int total = n_valid * F;
  const int lane = gsr_lane();
  for (int f = lane; f < total; f += 64) {
    const int g = f / F, o = f - g * F;
    if ((vis >> g) & 1ull) lds_wave[g * lds_stride + o0 + o] = src[f];
  }
}
// 16-byte vector with dword alignment: rows of 3K-3 floats start on 4-byte boundaries; gfx950 global loads / stores of
// dwordx4 only need dword alignment
typedef float gsr_f4u __attribute__((ext_vector_type(4), aligned(4)));

// One lane's row of F floats straight from global memory (dst may be registers or the lane's LDS row).
template <int F>
__device__ __forceinline__ void load_row(const float* __restrict__ src, float* dst) {
#pragma unroll
  for (int q = 0; q + 3 < F; q += 4) {
    const gsr_f4u t = *reinterpret_cast<const gsr_f4u*>(src + q);
    dst[q] = t.x; dst[q + 1] = t.y; dst[q + 2] = t.z; dst[q + 3] = t.w;
  }
#pragma unroll
  for (int q = F & ~3; q < F; ++q) dst[q] = src[q];
}

// LDS rows (float offset o0, F floats each) -> the wave's contiguous [n_valid, F] block in global memory, 16 bytes
// per lane and step. F == 0: runtime row length Fr. Rows of culled Gaussians hold zeros (the caller cleared them), so
// accumulating adds nothing there; 16-byte pieces that only cover culled rows are skipped when accumulating.
template <int F>
__device__ __forceinline__ void stage_rows_out(float* __restrict__ dst, int Fr, int o0, int lds_stride, int n_valid,
                                               const float* lds_wave, bool accumulate, unsigned long long vis) {
  const int FF = F > 0 ? F : Fr;
  const int total = n_valid * FF;
  const int lane = gsr_lane();
  for (int q = lane * 4; q < total; q += 64 * 4) {
    const int g0 = q / FF, g1 = min(n_valid - 1, (q + 3) / FF);
    if (accumulate && !(((vis >> g0) | (vis >> g1)) & 1ull)) continue;
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = q + k;
      const int g = f / FF, o = f - g * FF;
      e[k] = (f < total) ? lds_wave[g * lds_stride + o0 + o] : 0.f;
    }
    if (q + 3 < total) {
      float4 o = make_float4(e[0], e[1], e[2], e[3]);
      if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(dst + q);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      *reinterpret_cast<float4*>(dst + q) = o;
    } else {
      for (int k = 0; k < 4; ++k)
        if (q + k < total) dst[q + k] = accumulate ? dst[q + k] + e[k] : e[k];
    }
  }
}

// LDS -> global coalesced store of the wave's [n_valid, 3K] block.
template <int KT>
__device__ __forceinline__ void stage_sh_out(float* __restrict__ dst_base, int64_t wave_first, int n_valid, int K,
                                             const float* lds_wave, bool accumulate) {
  const int F = KT > 0 ? 3 * KT : 3 * K;
  const int stride = F | 1;
  const int total = n_valid * F;
  float* dst = dst_base + wave_first * (int64_t)F;
  const int lane = gsr_lane();
  for (int q = lane * 4; q < total; q += 64 * 4) {
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = q + k;
      const int g = f / F, o = f - g * F;
      e[k] = (f < total) ? lds_wave[g * stride + o] : 0.f;
    }
    if (q + 3 < total) {
      float4 o = make_float4(e[0], e[1], e[2], e[3]);
      if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(dst + q);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      *reinterpret_cast<float4*>(dst + q) = o;
    } else {
      for (int k = 0; k < 4; ++k)
        if (q + k < total) dst[q + k] = accumulate ? dst[q + k] + e[k] : e[k];
    }
  }
}

// ---- the per-view projection of one Gaussian (K1), shared by the single-view and the multi-view kernel
struct Proj {
  bool vis;
  int32_t radius;
  uint32_t ntiles, rect;   // rect: x0 | y0 << 8 | (w-1) << 16 | (h-1) << 24 (grids up to 256 x 256 tiles)
  float q0x, q0y, ca, cb, cc, depth;
};
// near-plane cull (view depth > 0.2) and the NDC position with the reference's 1/(w + 1e-7) (graphics_utils.py:29-36)
__device__ __forceinline__ bool proj_in_front(const ViewConst& vc, float px, float py, float pz, float& ndcx, float& ndcy) {
  const float tzq = ((vc.V[2] * px + vc.V[6] * py) + vc.V[10] * pz) + vc.V[14];
  if (!(tzq > GSR_NEAR_Z)) return false;
  const float* PV = vc.PV;
  const float hx = ((PV[0] * px + PV[4] * py) + PV[8] * pz) + PV[12];
  const float hy = ((PV[1] * px + PV[5] * py) + PV[9] * pz) + PV[13];
  const float hw = ((PV[3] * px + PV[7] * py) + PV[11] * pz) + PV[15];
  const float pw = 1.0f / (hw + 0.0000001f);
  ndcx = hx * pw; ndcy = hy * pw;
  return true;
}
__device__ __forceinline__ void proj_footprint(const ViewConst& vc, float px, float py, float pz, const float c6[6],
                                               float fx, float fy, float limx, float limy, int W, int H, int gx, int gy,
                                               float ndcx, float ndcy, Proj& o) {
  o.vis = false; o.radius = 0; o.ntiles = 0; o.rect = 0;
  Ewa e;
  ewa_forward(vc, px, py, pz, c6, fx, fy, limx, limy, e);
  if ((fabsf(e.det) > 0.0f) && (fabsf(e.det) < INFINITY)) {
    const float inv = 1.0f / e.det;
    const float mid = 0.5f * (e.ca + e.cc);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - e.det));
    const int32_t radius = gsr_f2i_sat(ceilf(3.0f * sqrtf(lam)));
    const float pxl = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float pyl = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float rf = (float)radius;
    const int32_t x0 = min(gx, max(0, gsr_f2i_sat((pxl - rf) * 0.0625f)));
    const int32_t y0 = min(gy, max(0, gsr_f2i_sat((pyl - rf) * 0.0625f)));
    const int32_t x1 = min(gx, max(0, gsr_f2i_sat(((pxl + rf) + 15.0f) * 0.0625f)));
    const int32_t y1 = min(gy, max(0, gsr_f2i_sat(((pyl + rf) + 15.0f) * 0.0625f)));
    o.ntiles = (uint32_t)((x1 - x0) * (y1 - y0));
    o.rect = (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)(x1 - x0 - 1) << 16) | ((uint32_t)(y1 - y0 - 1) << 24);
    if (o.ntiles != 0) {
      o.vis = true;
      o.radius = radius;
      o.q0x = pxl; o.q0y = pyl;
      o.ca = e.cc * inv; o.cb = -e.cb * inv; o.cc = e.ca * inv;
      o.depth = e.tz;
    }
  }
}
// Level of the conic form below which a splat can pass the alpha >= 1/255 gate: sigma exp(-q/2) >= 1/255 <=>
// q = d^T Conic d <= 2 ln(255 sigma) =: tau (slightly inflated). Used only to skip (pixel block, splat) pairs that
// cannot contribute (render.hip); negative = the splat contributes nowhere. Not part of any parity artefact.
__device__ __forceinline__ float splat_tau(float opac) {
  return (opac * 255.0f > 1.0f) ? 2.0f * logf(opac * 255.0f) * 1.0001f + 0.001f : -1.f;
}

// tanfov and the active SH degree travel by value in GsrView -- unless GsrView.dynamic names a device f32[4]
// (tanfovx, tanfovy, sh_degree, reserved): then the kernels take them from there WHEN THEY RUN, which is what lets a
// captured graph (hipGraph) of a step be replayed with the next step's cameras (dreamscene_amd/graph.py).
struct ViewDyn {
  float tanfovx, tanfovy;
  int sh_degree;
};
__device__ __forceinline__ ViewDyn view_dyn(const float* __restrict__ dyn, float tfx, float tfy, int D) {
  ViewDyn d;
  d.tanfovx = tfx; d.tanfovy = tfy; d.sh_degree = D;
  if (dyn) {
    gsr_cfloat* c = gsr_const(dyn);     // (scalar loads: see load_view_const)
    d.tanfovx = c[0]; d.tanfovy = c[1]; d.sh_degree = (int)c[2];
  }
  return d;
}

// --------------------------------------------------------------------------------------------------------- K1
// The loads of a Gaussian are issued as early as their addresses are known instead of behind the test that makes them
// necessary (EARLY below, and k_preprocess_views): rotation / scales / opacity (32 B) with the position, the SH row as soon as
// the Gaussian is in front of a camera -- two memory round trips per wave instead of three, the second one under the footprint
// arithmetic. Measured (round 4, one call, same box): k_preprocess_views<16> 65.8 -> 64.7 us at C3, k_preprocess<16> 30.9 ->
// 29.9 us per view: the kernel is not bound by its round trips (VALU 50 % busy, 4 TB/s of mixed read / write traffic).
template <int KT, bool SCENE = false, typename TAB = NoScene>
__global__ void __launch_bounds__(256)
k_preprocess(const GsrView v, const GsrGaussians g, const TAB sc, float* __restrict__ splat,
             int32_t* __restrict__ radii, uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ depth_keys,
             uint32_t* __restrict__ rects, uint32_t* __restrict__ sort_state, const uint32_t sort_state_words) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // the depth sort that follows starts from a zeroed state (digit histograms, tickets, look-back words; radix_sort.h): cleared
  // here, a few words per workgroup, instead of by a launch of its own in front of the sort
  for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < sort_state_words; w += gridDim.x * 256u) sort_state[w] = 0u;
  const int P = v.P, W = v.image_width, H = v.image_height, K = KT > 0 ? KT : v.sh_stride;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const Rows rw = resolve_rows<SCENE>(sc, P);
  const int64_t i = rw.i, row = rw.row, wave_first = rw.wave_row;
  const int n_valid = rw.n_valid;
  const float *p_xyz = g.means3D, *p_scale = g.scales, *p_rot = g.rotations, *p_opac = g.opacities;
  if constexpr (SCENE) {
    p_xyz = sc.xyz[rw.m]; p_scale = sc.scaling[rw.m]; p_rot = sc.rotation[rw.m]; p_opac = sc.opacity[rw.m];
  }
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const ViewDyn vd = view_dyn(v.dynamic, v.tanfovx, v.tanfovy, v.sh_degree);
  const float fx = (float)W / (2.0f * vd.tanfovx), fy = (float)H / (2.0f * vd.tanfovy);
  const float limx = 1.3f * vd.tanfovx, limy = 1.3f * vd.tanfovy;

  ViewConst vc;
  load_view(v, vc);

  bool vis = false;
  float px = 0, py = 0, pz = 0;
  float q0x = 0, q0y = 0, ca_ = 0, cb_ = 0, cc_ = 0, depth = 0, opac = 0, tau_ = -1.f;
  int32_t radius = 0;
  uint32_t ntiles = 0;
  uint32_t rect = 0;   // x0 | y0 << 8 | (w-1) << 16 | (h-1) << 24 of the tile rectangle (grids up to 256 x 256 tiles)
  constexpr bool EARLY = !SCENE;
  constexpr int FH = (EARLY && KT > 0) ? 3 * KT : 1;
  float shr_h[FH];
  if (rw.ok) {
    px = p_xyz[3 * row]; py = p_xyz[3 * row + 1]; pz = p_xyz[3 * row + 2];
    float sa_e[3] = {0.f, 0.f, 0.f}, opac_e = 0.f;
    float4 q_e = make_float4(0.f, 0.f, 0.f, 0.f);
    float c6_e[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (EARLY) {
      if (!g.cov3D_precomp) {
        sa_e[0] = p_scale[3 * row]; sa_e[1] = p_scale[3 * row + 1]; sa_e[2] = p_scale[3 * row + 2];
        q_e = *reinterpret_cast<const float4*>(p_rot + 4 * row);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6_e[k] = g.cov3D_precomp[6 * i + k];
      }
      opac_e = p_opac[row];
    }
    if constexpr (SCENE) {
      // activated values the caller gets back (scene_render returns the augmented scales, scene_gaussian.py:892)
      if (sc.scales_out) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float n = sc.scale_noise ? sc.scale_noise[3 * i + k] : 0.f;
          sc.scales_out[3 * i + k] = act_scale(p_scale[3 * row + k], sc.scale_noise != nullptr, n).out;
        }
      }
      if (sc.rotations_out) {
        const float4 q = *reinterpret_cast<const float4*>(p_rot + 4 * row);
        const float nrm = act_quat_norm(q);
        *reinterpret_cast<float4*>(sc.rotations_out + 4 * i) = make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);
      }
      if (sc.opacities_out) sc.opacities_out[i] = act_sigmoid(p_opac[row]);
    }
    float ndcx, ndcy;
    if (proj_in_front(vc, px, py, pz, ndcx, ndcy)) {
      if constexpr (EARLY && KT > 0) {
        if (g.shs) load_row<FH>(g.shs + (size_t)i * FH, shr_h);     // in flight during the footprint arithmetic
      }
      float c6[6];
      if (g.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = EARLY ? c6_e[k] : g.cov3D_precomp[6 * i + k];
      } else {
        const float mod = v.scale_modifier;
        float sa[3];
        float4 q;
        if constexpr (EARLY) {
          sa[0] = sa_e[0]; sa[1] = sa_e[1]; sa[2] = sa_e[2]; q = q_e;
        } else {
          sa[0] = p_scale[3 * row]; sa[1] = p_scale[3 * row + 1]; sa[2] = p_scale[3 * row + 2];
          q = *reinterpret_cast<const float4*>(p_rot + 4 * row);
        }
        if constexpr (SCENE) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float n = sc.scale_noise ? sc.scale_noise[3 * i + k] : 0.f;
            sa[k] = act_scale(sa[k], sc.scale_noise != nullptr, n).out;
          }
          const float nrm = act_quat_norm(q);
          q = make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);
        }
        const float s0 = mod * sa[0], s1 = mod * sa[1], s2 = mod * sa[2];
        float R[9];
        quat_to_R(q, R);
        cov3d_from(s0, s1, s2, R, c6);
      }
      Proj pr;
      proj_footprint(vc, px, py, pz, c6, fx, fy, limx, limy, W, H, gx, gy, ndcx, ndcy, pr);
      radius = pr.radius; ntiles = pr.ntiles; rect = pr.rect;
      if (pr.vis) {
        vis = true;
        q0x = pr.q0x; q0y = pr.q0y; ca_ = pr.ca; cb_ = pr.cb; cc_ = pr.cc; depth = pr.depth;
        if constexpr (EARLY) opac = opac_e;
        else opac = SCENE ? act_sigmoid(p_opac[row]) : p_opac[row];
        tau_ = splat_tau(opac);
      }
    }
  }

  // ---- colour
  float rgb[3] = {0.f, 0.f, 0.f};
  if constexpr (SCENE && KT > 0) {
    // raw leaves, compile-time K: each lane pulls its features_dc / features_rest rows straight into registers
    if (vis) {
      constexpr int F = 3 * KT;
      float shr[F];
      load_row<3>(sc.dc[rw.m] + row * 3, shr);
      if constexpr (KT > 1) load_row<F - 3>(sc.rest[rw.m] + row * (F - 3), shr + 3);
      if (sc.sh_noise) {
        float nz[F];
        load_row<F>(sc.sh_noise + (size_t)i * F, nz);
#pragma unroll
        for (int k = 0; k < F; ++k) shr[k] = shr[k] + nz[k] * (kSqrtPoint2 * shr[k]);
      }
      float dx = px - vc.cam[0], dy = py - vc.cam[1], dz = pz - vc.cam[2];
      const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float b[16];
      sh_basis(vd.sh_degree, dx, dy, dz, b);
      float acc[3];
      sh_colour_n<KT>(vd.sh_degree, shr, b, acc);
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = fmaxf(acc[c] + 0.5f, 0.0f);
    }
  } else if (!SCENE && g.shs && KT > 0) {
    // compile-time SH stride: every lane pulls its own 12*KT-byte row straight into registers (measured faster
    // than the coalesced-load + LDS-transpose path K8 uses for its read-modify-write of the same block: the rows
    // of a wave are contiguous, so L1/TA serve the 16-byte pieces of one line to successive loads)
    if (vis) {
      constexpr int F = 3 * (KT > 0 ? KT : 1);
      float shr[F];
      if constexpr (EARLY) {
#pragma unroll
        for (int q = 0; q < F; ++q) shr[q] = shr_h[q];
      } else {
        const float* rp = g.shs + (size_t)i * F;
        if constexpr (F % 4 == 0) {
          const float4* r = reinterpret_cast<const float4*>(rp);
#pragma unroll
          for (int q = 0; q < F / 4; ++q) { const float4 t = r[q]; shr[4*q] = t.x; shr[4*q+1] = t.y; shr[4*q+2] = t.z; shr[4*q+3] = t.w; }
        } else {
#pragma unroll
          for (int q = 0; q < F; ++q) shr[q] = rp[q];
        }
      }
      float dx = px - vc.cam[0], dy = py - vc.cam[1], dz = pz - vc.cam[2];
      const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float b[16];
      sh_basis(vd.sh_degree, dx, dy, dz, b);
      float acc[3];
      sh_colour_n<(KT > 0 ? KT : 1)>(vd.sh_degree, shr, b, acc);
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = fmaxf(acc[c] + 0.5f, 0.0f);
    }
  } else if (SCENE || g.shs) {
    const unsigned long long vmask = __ballot(vis);
    float* lw = lds + wave * (64 * sh_lds_stride(K));
    if constexpr (SCENE) {
      // features_dc / features_rest rows of the wave, side by side in the lanes' LDS rows; then the SH noise
      if (vmask) {
        stage_rows_in(sc.dc[rw.m] + wave_first * 3, 3, 0, sh_lds_stride(K), n_valid, vmask, lw);
        if (K > 1) stage_rows_in(sc.rest[rw.m] + wave_first * (3 * K - 3), 3 * K - 3, 3, sh_lds_stride(K), n_valid, vmask, lw);
      }
      if (sc.sh_noise && vis) {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float* shw = lw + lane * sh_lds_stride(K);
        const float* nz = sc.sh_noise + (size_t)i * (3 * K);
        for (int k = 0; k < 3 * K; ++k) shw[k] = shw[k] + nz[k] * (kSqrtPoint2 * shw[k]);
      }
    } else if (vmask) stage_sh_in<KT>(g.shs, wave_first, n_valid, K, vmask, lw);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (vis) {
      float dx = px - vc.cam[0], dy = py - vc.cam[1], dz = pz - vc.cam[2];
      const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float b[16];
      sh_basis(vd.sh_degree, dx, dy, dz, b);
      const float* sh = lw + lane * sh_lds_stride(K);
      float acc[3];
      sh_colour(vd.sh_degree, sh, b, acc);
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = fmaxf(acc[c] + 0.5f, 0.0f);
    }
  } else if (vis) {
    rgb[0] = g.colors_precomp[3 * i]; rgb[1] = g.colors_precomp[3 * i + 1]; rgb[2] = g.colors_precomp[3 * i + 2];
  }

  if (rw.ok) {
    radii[i] = radius;
    tiles_touched[i] = ntiles;
    depth_keys[i] = vis ? __float_as_uint(depth) : 0xFFFFFFFFu;   // culled Gaussians are dropped by the depth sort
    rects[i] = rect;
    if (vis) {
      float4* o = reinterpret_cast<float4*>(splat + 12 * i);
      o[0] = make_float4(q0x, q0y, ca_, cb_);
      o[1] = make_float4(cc_, opac, depth, rgb[0]);
      o[2] = make_float4(rgb[1], rgb[2], tau_, 0.f);
    }
  }

}


// ------------------------------------------------------------------------------------------- K1 over several views
// The parameter rows (44 + 12K bytes per Gaussian) are read once for all views of a step; per view only the 48-byte
// splat record, radius, tile count, depth key and tile rectangle are written. cov3D is view independent.
struct K1Views {
  int32_t nv;
  const float* viewmatrix[GSR_MAX_BATCH_VIEWS];
  const float* projmatrix[GSR_MAX_BATCH_VIEWS];
  const float* campos[GSR_MAX_BATCH_VIEWS];
  float tanfovx[GSR_MAX_BATCH_VIEWS];
  float tanfovy[GSR_MAX_BATCH_VIEWS];
  int32_t sh_degree[GSR_MAX_BATCH_VIEWS];
  const float* dyn[GSR_MAX_BATCH_VIEWS];    // GsrView.dynamic of every view (NULL: the by-value entries above)
  int32_t per_view_scales;
  const float* scales[GSR_MAX_BATCH_VIEWS];
  // scene input (raw leaves): per-view noise samples and the per-view activated scales handed back to the caller
  const float* scale_noise[GSR_MAX_BATCH_VIEWS];
  const float* sh_noise[GSR_MAX_BATCH_VIEWS];
  float* scales_out[GSR_MAX_BATCH_VIEWS];
  float* splat[GSR_MAX_BATCH_VIEWS];
  int32_t* radii[GSR_MAX_BATCH_VIEWS];
  uint32_t* tiles_touched[GSR_MAX_BATCH_VIEWS];
  uint32_t* depth_keys[GSR_MAX_BATCH_VIEWS];
  uint32_t* rects[GSR_MAX_BATCH_VIEWS];
  uint32_t* sort_state[GSR_MAX_BATCH_VIEWS];   // state of the depth sort that follows, cleared here (see k_preprocess)
  uint32_t sort_state_words;
};

__device__ __forceinline__ void k1_clear_sort_state(const K1Views& vb) {
  for (int vv = 0; vv < vb.nv; ++vv)
    for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < vb.sort_state_words; w += gridDim.x * 256u) vb.sort_state[vv][w] = 0u;
}

template <int KT>
// (4 waves per SIMD: 128 VGPRs instead of 131 at K = 16, no spills -- the kernel is latency-bound on its division chains.
//  Round 6 probe, one call: the SAME kernel without its colour -- no SH row loaded or held, ~60 VGPRs -- takes 11.7 us per view at 4
//  AND at 6 waves per SIMD, 10.0 at 8, against 16.1-17.3 with the colour: the geometry alone is 40-47 us per 4-view step at 3.2 TB/s
//  of its own traffic, whatever the occupancy; the colour adds ~20 us for 96 MB of SH rows, i.e. it already runs at a stream's
//  rate. A geometry pass + a separate colour pass would cost 47 + >= 22 us: the split round 4's review asked for does not pay.)
#ifndef GSR_K1_WAVES
#define GSR_K1_WAVES 4
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GSR_K1_WAVES, GSR_K1_WAVES)))
k_preprocess_views(const GsrView v, const GsrGaussians g, const K1Views vb) {
  constexpr int F = 3 * KT;
  const int P = v.P, W = v.image_width, H = v.image_height;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  k1_clear_sort_state(vb);
  if (i >= P) return;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const float px = g.means3D[3 * i], py = g.means3D[3 * i + 1], pz = g.means3D[3 * i + 2];
  // ONE memory round trip in front of the arithmetic instead of three (position -> scales / rotation -> SH row, each
  // issued only after the test that needs the one before): rotation, scales and opacity (32 B) are fetched with the
  // position whatever the tests will say; the SH row (12K B) as soon as the position shows the Gaussian in front of ANY of the
  // views (3 FMAs per view) -- it is in flight while cov3D and the first footprint are computed.
  const float4 q_l = *reinterpret_cast<const float4*>(g.rotations + 4 * i);
  const float opac = g.opacities[i];
  float sl[3] = {0.f, 0.f, 0.f};
  if (!vb.per_view_scales) { const float* sc = vb.scales[0]; sl[0] = sc[3 * i]; sl[1] = sc[3 * i + 1]; sl[2] = sc[3 * i + 2]; }
  bool front_any = false;
  for (int vv = 0; vv < vb.nv; ++vv) {
    gsr_cfloat* V = gsr_const(vb.viewmatrix[vv]);
    front_any |= (((V[2] * px + V[6] * py) + V[10] * pz) + V[14]) > GSR_NEAR_Z;
  }
  if (!front_any) {        // behind every camera: the culled record of every view, and out (no join in front of the loop below:
    for (int vv = 0; vv < vb.nv; ++vv) {   // the wait counts of the loads stay exact)
      vb.radii[vv][i] = 0; vb.tiles_touched[vv][i] = 0u; vb.depth_keys[vv][i] = 0xFFFFFFFFu; vb.rects[vv][i] = 0u;
    }
    return;
  }
  float c6[6], shr[F];
  load_row<F>(g.shs + (size_t)i * F, shr);
  if (!vb.per_view_scales) {
    const float mod = v.scale_modifier;
    float R[9];
    quat_to_R(q_l, R);
    cov3d_from(mod * sl[0], mod * sl[1], mod * sl[2], R, c6);
  }
  const float tau = splat_tau(opac);
  // View 0 is peeled: the wait for the SH row stands behind its footprint arithmetic and in front of its stores, so the
  // loop over the other views has no load of the prologue pending -- a wait INSIDE the loop would also wait for the
  // previous view's stores (on gfx9 loads and stores share vmcnt and complete in order).
  auto one_view = [&](const int vv, auto first) {
    ViewConst vc;
    load_view_const(vb.viewmatrix[vv], vb.projmatrix[vv], vb.campos[vv], vc);
    const ViewDyn vd = view_dyn(vb.dyn[vv], vb.tanfovx[vv], vb.tanfovy[vv], vb.sh_degree[vv]);
    const float tfx = vd.tanfovx, tfy = vd.tanfovy;
    const float fx = (float)W / (2.0f * tfx), fy = (float)H / (2.0f * tfy);
    Proj pr;
    pr.vis = false; pr.radius = 0; pr.ntiles = 0; pr.rect = 0;
    float ndcx, ndcy;
    if (proj_in_front(vc, px, py, pz, ndcx, ndcy)) {
      if (vb.per_view_scales) {
        const float mod = v.scale_modifier;
        const float* sc = vb.scales[vv];
        const float s0 = mod * sc[3 * i], s1 = mod * sc[3 * i + 1], s2 = mod * sc[3 * i + 2];
        float R[9];
        quat_to_R(q_l, R);
        cov3d_from(s0, s1, s2, R, c6);
      }
      proj_footprint(vc, px, py, pz, c6, fx, fy, 1.3f * tfx, 1.3f * tfy, W, H, gx, gy, ndcx, ndcy, pr);
    }
    if constexpr (decltype(first)::value) {
#pragma unroll
      for (int k = 0; k < F; ++k) asm volatile("" : "+v"(shr[k]));
    }
    float rgb[3] = {0.f, 0.f, 0.f};
    if (pr.vis) {
      float dx = px - vc.cam[0], dy = py - vc.cam[1], dz = pz - vc.cam[2];
      const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float b[16];
      sh_basis(vd.sh_degree, dx, dy, dz, b);
      float acc[3];
      sh_colour_n<KT>(vd.sh_degree, shr, b, acc);
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = fmaxf(acc[c] + 0.5f, 0.0f);
    }
    vb.radii[vv][i] = pr.radius;
    vb.tiles_touched[vv][i] = pr.ntiles;
    vb.depth_keys[vv][i] = pr.vis ? __float_as_uint(pr.depth) : 0xFFFFFFFFu;
    vb.rects[vv][i] = pr.rect;
    if (pr.vis) {
      float4* o = reinterpret_cast<float4*>(vb.splat[vv] + 12 * i);
      o[0] = make_float4(pr.q0x, pr.q0y, pr.ca, pr.cb);
      o[1] = make_float4(pr.cc, opac, pr.depth, rgb[0]);
      o[2] = make_float4(rgb[1], rgb[2], tau, 0.f);
    }
  };
  one_view(0, std::true_type{});
  for (int vv = 1; vv < vb.nv; ++vv) one_view(vv, std::false_type{});
}


// K1 over several views of a SCENE (raw leaves of several models, activations fused; see k_preprocess<K, true>): the
// raw rows are read once, exp / normalize / sigmoid are applied once, the per-view scale noise (and SH noise) per view.
template <int KT>
__global__ void __launch_bounds__(256)
k_preprocess_views_scene(const GsrView v, const SceneTab sc, const K1Views vb) {
  constexpr int F = 3 * KT;
  const int W = v.image_width, H = v.image_height;
  k1_clear_sort_state(vb);
  const Rows rw = resolve_rows<true>(sc, v.P);
  if (!rw.ok) return;
  const int64_t i = rw.i, row = rw.row;
  const int m = rw.m;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const float* xyz = sc.xyz[m];
  const float px = xyz[3 * row], py = xyz[3 * row + 1], pz = xyz[3 * row + 2];
  float aact[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) aact[k] = expf(sc.scaling[m][3 * row + k]);
  bool have_R = false, have_cov = false, have_sh = false;
  float R[9], c6[6], shr[F];
  float opac = 0.f, tau = -1.f;
  for (int vv = 0; vv < vb.nv; ++vv) {
    const float* sn = vb.scale_noise[vv];
    float sa[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      sa[k] = sn ? fmaxf(aact[k] + sn[3 * i + k] * ((kSqrtPoint2 * aact[k]) / 4.0f), 0.0f) : aact[k];
    if (vb.scales_out[vv]) {
      float* so = vb.scales_out[vv];
      so[3 * i] = sa[0]; so[3 * i + 1] = sa[1]; so[3 * i + 2] = sa[2];
    }
    ViewConst vc;
    load_view_const(vb.viewmatrix[vv], vb.projmatrix[vv], vb.campos[vv], vc);
    const ViewDyn vd = view_dyn(vb.dyn[vv], vb.tanfovx[vv], vb.tanfovy[vv], vb.sh_degree[vv]);
    const float tfx = vd.tanfovx, tfy = vd.tanfovy;
    const float fx = (float)W / (2.0f * tfx), fy = (float)H / (2.0f * tfy);
    Proj pr;
    pr.vis = false; pr.radius = 0; pr.ntiles = 0; pr.rect = 0;
    float ndcx, ndcy;
    if (proj_in_front(vc, px, py, pz, ndcx, ndcy)) {
      if (!have_R) {
        have_R = true;
        float4 q = *reinterpret_cast<const float4*>(sc.rotation[m] + 4 * row);
        const float nrm = act_quat_norm(q);
        q = make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);
        quat_to_R(q, R);
      }
      if (!have_cov || sn) {
        have_cov = true;
        const float mod = v.scale_modifier;
        cov3d_from(mod * sa[0], mod * sa[1], mod * sa[2], R, c6);
      }
      proj_footprint(vc, px, py, pz, c6, fx, fy, 1.3f * tfx, 1.3f * tfy, W, H, gx, gy, ndcx, ndcy, pr);
    }
    float rgb[3] = {0.f, 0.f, 0.f};
    if (pr.vis) {
      if (!have_sh) {
        have_sh = true;
        load_row<3>(sc.dc[m] + row * 3, shr);
        if constexpr (KT > 1) load_row<F - 3>(sc.rest[m] + row * (F - 3), shr + 3);
        opac = act_sigmoid(sc.opacity[m][row]);
        tau = splat_tau(opac);
      }
      float dx = px - vc.cam[0], dy = py - vc.cam[1], dz = pz - vc.cam[2];
      const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float b[16];
      sh_basis(vd.sh_degree, dx, dy, dz, b);
      float acc[3];
      if (vb.sh_noise[vv]) {
        float shv[F];
        const float* nz = vb.sh_noise[vv] + (size_t)i * F;
#pragma unroll
        for (int k = 0; k < F; ++k) shv[k] = shr[k] + nz[k] * (kSqrtPoint2 * shr[k]);
        sh_colour_n<KT>(vd.sh_degree, shv, b, acc);
      } else {
        sh_colour_n<KT>(vd.sh_degree, shr, b, acc);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = fmaxf(acc[c] + 0.5f, 0.0f);
    }
    vb.radii[vv][i] = pr.radius;
    vb.tiles_touched[vv][i] = pr.ntiles;
    vb.depth_keys[vv][i] = pr.vis ? __float_as_uint(pr.depth) : 0xFFFFFFFFu;
    vb.rects[vv][i] = pr.rect;
    if (pr.vis) {
      float4* o = reinterpret_cast<float4*>(vb.splat[vv] + 12 * i);
      o[0] = make_float4(pr.q0x, pr.q0y, pr.ca, pr.cb);
      o[1] = make_float4(pr.cc, opac, pr.depth, rgb[0]);
      o[2] = make_float4(rgb[1], rgb[2], tau, 0.f);
    }
  }
}

// d colour / d (unit view direction), contracted with s_k = <sh_k, dL/dcolour>: the derivative of the SH basis
__device__ __forceinline__ void sh_ddir(int D, float x, float y, float z, const float s[16], float& ddx, float& ddy,
                                        float& ddz) {
  ddx = 0.f; ddy = 0.f; ddz = 0.f;
  if (D > 0) {
    ddy += -GSR_SH_C1 * s[1]; ddz += GSR_SH_C1 * s[2]; ddx += -GSR_SH_C1 * s[3];
    if (D > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      ddx += GSR_SH_C2_0 * y * s[4] + GSR_SH_C2_2 * (-2.0f * x) * s[6] + GSR_SH_C2_3 * z * s[7] + GSR_SH_C2_4 * 2.0f * x * s[8];
      ddy += GSR_SH_C2_0 * x * s[4] + GSR_SH_C2_1 * z * s[5] + GSR_SH_C2_2 * (-2.0f * y) * s[6] + GSR_SH_C2_4 * (-2.0f * y) * s[8];
      ddz += GSR_SH_C2_1 * y * s[5] + GSR_SH_C2_2 * 4.0f * z * s[6] + GSR_SH_C2_3 * x * s[7];
      if (D > 2) {
        ddx += GSR_SH_C3_0 * 6.0f * xy * s[9] + GSR_SH_C3_1 * yz * s[10] + GSR_SH_C3_2 * (-2.0f * xy) * s[11] +
               GSR_SH_C3_3 * (-6.0f * xz) * s[12] + GSR_SH_C3_4 * (4.0f * zz - 3.0f * xx - yy) * s[13] +
               GSR_SH_C3_5 * 2.0f * xz * s[14] + GSR_SH_C3_6 * 3.0f * (xx - yy) * s[15];
        ddy += GSR_SH_C3_0 * 3.0f * (xx - yy) * s[9] + GSR_SH_C3_1 * xz * s[10] +
               GSR_SH_C3_2 * (4.0f * zz - xx - 3.0f * yy) * s[11] + GSR_SH_C3_3 * (-6.0f * yz) * s[12] +
               GSR_SH_C3_4 * (-2.0f * xy) * s[13] + GSR_SH_C3_5 * (-2.0f * yz) * s[14] + GSR_SH_C3_6 * (-6.0f * xy) * s[15];
        ddz += GSR_SH_C3_1 * xy * s[10] + GSR_SH_C3_2 * 8.0f * yz * s[11] +
               GSR_SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy) * s[12] + GSR_SH_C3_4 * 8.0f * xz * s[13] +
               GSR_SH_C3_5 * (xx - yy) * s[14];
      }
    }
  }
}

// Steps (2a)-(5) of K8 for one view: K7's moments -> dL/d(ndc xy), dL/dconic -> cov2D -> dL/dSigma (dS, assigned) and
// the position terms (added to dp); camera gradients (assigned) when want_cam.
__device__ __forceinline__ void geom_backward(const ViewConst& vc, const Ewa& e, float fx, float fy, int W, int H,
                                              float px, float py, float pz, float S1, float S2, float S3, float S4,
                                              float S5, float gdep, bool want_cam, float& gndx, float& gndy,
                                              float dS[9], float dp[3], float dview[12], float dproj[12]) {
  const float* V = vc.V;
  const float* PV = vc.PV;
  // (2) K7's sums -> dL/d(ndc xy) and dL/dcov2D. With d = centre - pixel, (u, v) = -conic d and q = dL/dG G per pixel:
  // S1 = sum q u, S2 = sum q v are dL/d(pixel centre); dL/dSigma = 1/2 sum q (conic d)(conic d)^T, i.e. S3 = sum q u^2,
  // S4 = sum q u v, S5 = sum q v^2 are the covariance gradient up to the factors below -- K7 forms them per pixel
  // (render.hip). The lineage goes through dL/dconic and divides by det^2 + 1e-7 instead of det^2: the factor
  // det^2 / (det^2 + 1e-7) keeps that regulariser (SEMANTICS.md section 5).
  const float ca = e.ca, cb = e.cb, cc = e.cc;
  gndx = S1 * (0.5f * (float)W);
  gndy = S2 * (0.5f * (float)H);
  const float det2 = e.det * e.det;
  const float reg = det2 * (1.0f / (det2 + 0.0000001f));
  const float dca = (0.5f * S3) * reg, dcb = S4 * reg, dcc = (0.5f * S5) * reg;
  (void)ca; (void)cb; (void)cc;
  // (3) cov2D = M Sigma M^T
  const float h = 0.5f * dcb;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dS[3 * r + c] = (e.M0[r] * (dca * e.M0[c] + h * e.M1[c])) + (e.M1[r] * (h * e.M0[c] + dcc * e.M1[c]));
  float dM0[3], dM1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    dM0[j] = 2.0f * (dca * e.U0[j] + h * e.U1[j]);
    dM1[j] = 2.0f * (h * e.U0[j] + dcc * e.U1[j]);
  }
  const float dJ00 = (dM0[0] * V[0] + dM0[1] * V[4]) + dM0[2] * V[8];
  const float dJ02 = (dM0[0] * V[2] + dM0[1] * V[6]) + dM0[2] * V[10];
  const float dJ11 = (dM1[0] * V[1] + dM1[1] * V[5]) + dM1[2] * V[9];
  const float dJ12 = (dM1[0] * V[2] + dM1[1] * V[6]) + dM1[2] * V[10];
  const float tzi = 1.0f / e.tz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
  float dt[3];
  dt[0] = e.clx ? 0.0f : (-fx * tz2 * dJ02);
  dt[1] = e.cly ? 0.0f : (-fy * tz2 * dJ12);
  dt[2] = ((-fx * tz2 * dJ00 - fy * tz2 * dJ11) + (2.0f * fx * e.txc) * tz3 * dJ02) + (2.0f * fy * e.tyc) * tz3 * dJ12;
  dt[2] += gdep;                                                               // (5) depth
#pragma unroll
  for (int r = 0; r < 3; ++r) dp[r] += (V[4 * r] * dt[0] + V[4 * r + 1] * dt[1]) + V[4 * r + 2] * dt[2];
  // (4) ndc -> p
  const float hx = ((PV[0] * px + PV[4] * py) + PV[8] * pz) + PV[12];
  const float hy = ((PV[1] * px + PV[5] * py) + PV[9] * pz) + PV[13];
  const float hw = ((PV[3] * px + PV[7] * py) + PV[11] * pz) + PV[15];
  const float pw = 1.0f / (hw + 0.0000001f);
  const float dh[3] = {pw * gndx, pw * gndy, -(pw * pw) * (hx * gndx + hy * gndy)};
#pragma unroll
  for (int r = 0; r < 3; ++r) dp[r] += (PV[4 * r] * dh[0] + PV[4 * r + 1] * dh[1]) + PV[4 * r + 3] * dh[2];

  if (want_cam) {
    const float p4[4] = {px, py, pz, 1.0f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dview[3 * r + c] = p4[r] * dt[c];
      dproj[3 * r + 0] = p4[r] * dh[0];
      dproj[3 * r + 1] = p4[r] * dh[1];
      dproj[3 * r + 2] = p4[r] * dh[2];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      dview[3 * r + 0] += e.J00 * dM0[r];
      dview[3 * r + 1] += e.J11 * dM1[r];
      dview[3 * r + 2] += e.J02 * dM0[r] + e.J12 * dM1[r];
    }
  }
}

// Step (6) of K8: dL/dSigma -> dL/dscale (of the scales as given) and dL/dquaternion (of the quaternion as given).
__device__ __forceinline__ void sigma_backward(const float dS[9], const float R[9], const float s3[3], float mod,
                                               const float4 q, float dscale[3], float drot[4]) {
  float L[9], dL[9], dR[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2) L[3 * a + b2] = R[3 * a + b2] * s3[b2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc += (dS[3 * a + k] + dS[3 * k + a]) * L[3 * k + b2];
      dL[3 * a + b2] = acc;
    }
#pragma unroll
  for (int b2 = 0; b2 < 3; ++b2) {
    const float ds = (dL[b2] * R[b2] + dL[3 + b2] * R[3 + b2]) + dL[6 + b2] * R[6 + b2];
    dscale[b2] = mod * ds;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2) dR[3 * a + b2] = dL[3 * a + b2] * s3[b2];
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  drot[0] = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
  drot[1] = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
  drot[2] = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
  drot[3] = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}
// ---- K7's per-Gaussian sums: a row of 16 doubles, 12 used (render.hip, render_bwd_body) -- wave results added across
// waves in double, rounded to fp32 HERE, once. partial_rows() hands them out as the chain rule below was written:
// a = (S1, S2, S3, S4), b = (S5, dL/dopacity, r, g), c = (b, depth, b', depth').
struct PartialRaw { float4 w[6]; };     // as loaded: 12 doubles, still raw bits (conversions wait for the loads: do them late)
__device__ __forceinline__ PartialRaw partial_load(const float* __restrict__ partials, int64_t i) {
  const float4* pp = reinterpret_cast<const float4*>(partials + GSR_PARTIAL_WORDS * i);
  PartialRaw r;
#pragma unroll
  for (int k = 0; k < 6; ++k) r.w[k] = pp[k];
  return r;
}
__device__ __forceinline__ PartialRaw partial_none() {
  PartialRaw r;
#pragma unroll
  for (int k = 0; k < 6; ++k) r.w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  return r;
}
__device__ __forceinline__ float f64_words(float lo, float hi) {
  return (float)__hiloint2double(__float_as_int(hi), __float_as_int(lo));
}
__device__ __forceinline__ void partial_rows(const PartialRaw& r, float4& a, float4& b, float4& c) {
  a = make_float4(f64_words(r.w[0].x, r.w[0].y), f64_words(r.w[0].z, r.w[0].w), f64_words(r.w[1].x, r.w[1].y), f64_words(r.w[1].z, r.w[1].w));
  b = make_float4(f64_words(r.w[2].x, r.w[2].y), f64_words(r.w[2].z, r.w[2].w), f64_words(r.w[3].x, r.w[3].y), f64_words(r.w[3].z, r.w[3].w));
  c = make_float4(f64_words(r.w[4].x, r.w[4].y), f64_words(r.w[4].z, r.w[4].w), f64_words(r.w[5].x, r.w[5].y), f64_words(r.w[5].z, r.w[5].w));
}
__device__ __forceinline__ void partial_zero(float* __restrict__ partials, int64_t i) {
  float4* pp = reinterpret_cast<float4*>(partials + GSR_PARTIAL_WORDS * i);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 6; ++k) pp[k] = z;
}

// --------------------------------------------------------------------------------------------------------- K8
// partials [P,16 doubles] from K7 (see above): S1 = sum q u, S2 = sum q v, S3 = sum q u^2, S4 = sum q u v, S5 = sum q v^2
// [(u, v) = -conic d], dL/dopacity, dL/dr, dL/dg, dL/db, dL/ddepth; q = dL/dG * G
template <int KT, bool SCENE = false, typename TAB = NoScene, typename GTAB = NoScene>
__global__ void __launch_bounds__(256)
k_preprocess_bwd(const GsrView v, const GsrGaussians g, const TAB sc, const GTAB sg,
                 const int32_t* __restrict__ radii, const float* __restrict__ partials, const GsrGrads out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float cam_red[4][32];
  const ViewDyn vd = view_dyn(v.dynamic, v.tanfovx, v.tanfovy, v.sh_degree);
  const int P = v.P, W = v.image_width, H = v.image_height, K = KT > 0 ? KT : v.sh_stride, D = vd.sh_degree;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const Rows rw = resolve_rows<SCENE>(sc, P);
  const int64_t i = rw.i, row = rw.row, wave_first = rw.wave_row;
  const int n_valid = rw.n_valid;
  const float *p_xyz = g.means3D, *p_scale = g.scales, *p_rot = g.rotations;
  if constexpr (SCENE) { p_xyz = sc.xyz[rw.m]; p_scale = sc.scaling[rw.m]; p_rot = sc.rotation[rw.m]; }
  const float fx = (float)W / (2.0f * vd.tanfovx), fy = (float)H / (2.0f * vd.tanfovy);
  const float limx = 1.3f * vd.tanfovx, limy = 1.3f * vd.tanfovy;
  const float mod = v.scale_modifier;
  const bool want_cam = (out.dL_dview != nullptr) || (out.dL_dproj != nullptr) || (out.dL_dcampos != nullptr);

  ViewConst vc;
  load_view(v, vc);

  const bool vis = rw.ok && (radii[i] > 0);
  float px = 0, py = 0, pz = 0;
  float4 pa = make_float4(0, 0, 0, 0), pb = pa, pc = pa;
  if (vis) {
    px = p_xyz[3 * row]; py = p_xyz[3 * row + 1]; pz = p_xyz[3 * row + 2];
    partial_rows(partial_load(partials, i), pa, pb, pc);
    pc.x += pc.z; pc.y += pc.w;      // (K7 commits the last two sums from the two halves of a wave: render.hip, reduce10)
  }
  const float S1 = pa.x, S2 = pa.y, S3 = pa.z, S4 = pa.w, S5 = pb.x, gop = pb.y;
  float gndx = 0.f, gndy = 0.f;
  const float grgb[3] = {pb.z, pb.w, pc.x};
  const float gdep = pc.y;

  float dp[3] = {0.f, 0.f, 0.f};
  float dview[12];   // rows 0..3 x cols 0..2 of dL/dviewmatrix
  float dproj[12];   // rows 0..3 x cols {0,1,3} of dL/dprojmatrix
  float dcam[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 12; ++k) { dview[k] = 0.f; dproj[k] = 0.f; }

  // ---- (1) colour -> SH coefficients, view direction
  if (SCENE || g.shs) {
    const unsigned long long vmask = __ballot(vis);
    const int stride = sh_lds_stride(K);
    float* lw = lds + wave * (64 * stride);
    float* sh = lw + lane * stride;
    if constexpr (SCENE) {
      if constexpr (KT > 0) {
        if (vis) {   // each lane parks its own rows in its LDS row
          load_row<3>(sc.dc[rw.m] + row * 3, sh);
          if constexpr (KT > 1) load_row<3 * KT - 3>(sc.rest[rw.m] + row * (3 * KT - 3), sh + 3);
        }
      } else {
        if (vmask) {
          stage_rows_in(sc.dc[rw.m] + wave_first * 3, 3, 0, stride, n_valid, vmask, lw);
          if (K > 1) stage_rows_in(sc.rest[rw.m] + wave_first * (3 * K - 3), 3 * K - 3, 3, stride, n_valid, vmask, lw);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      if (sc.sh_noise && vis) {   // the augmented coefficients K1 saw
        const float* nz = sc.sh_noise + (size_t)i * (3 * K);
        for (int k = 0; k < 3 * K; ++k) sh[k] = sh[k] + nz[k] * (kSqrtPoint2 * sh[k]);
      }
    } else if constexpr (KT > 0 && (3 * KT) % 4 == 0) {
      // compile-time stride: each lane pulls its own row (as K1 does) and parks it in its LDS row; LDS is still
      // needed for the coalesced write-back of dL/dSH
      if (vis) {
        const float4* r = reinterpret_cast<const float4*>(g.shs + (size_t)i * (3 * KT));
#pragma unroll
        for (int q = 0; q < (3 * KT) / 4; ++q) {
          const float4 t = r[q];
          sh[4 * q] = t.x; sh[4 * q + 1] = t.y; sh[4 * q + 2] = t.z; sh[4 * q + 3] = t.w;
        }
      }
    } else {
      if (vmask) stage_sh_in<KT>(g.shs, wave_first, n_valid, K, vmask, lw);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (vis) {
      const float vx = px - vc.cam[0], vy = py - vc.cam[1], vz = pz - vc.cam[2];
      const float len = sqrtf((vx * vx + vy * vy) + vz * vz);
      const float x = vx / len, y = vy / len, z = vz / len;
      float b[16];
      sh_basis(D, x, y, z, b);
      const int nb = (D + 1) * (D + 1);
      float gch[3];
      {
        float acc[3];
        sh_colour(D, sh, b, acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) gch[c] = (acc[c] + 0.5f < 0.0f) ? 0.0f : grgb[c];   // K1's clamp decision (same operator order)
      }
      float s[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) s[k] = 0.f;
      // s_k = <sh_k, g>, and the row is overwritten in place by dL/dsh_k = b_k g (constant register indices per band)
#define GSR_SH_BWD_BAND(K0, K1)                                                                        \
  _Pragma("unroll") for (int k = K0; k <= K1; ++k) {                                                   \
    s[k] = (sh[3 * k] * gch[0] + sh[3 * k + 1] * gch[1]) + sh[3 * k + 2] * gch[2];                     \
    sh[3 * k] = b[k] * gch[0]; sh[3 * k + 1] = b[k] * gch[1]; sh[3 * k + 2] = b[k] * gch[2];           \
  }
      GSR_SH_BWD_BAND(0, 0)
      if (D > 0) {
        GSR_SH_BWD_BAND(1, 3)
        if (D > 1) {
          GSR_SH_BWD_BAND(4, 8)
          if (D > 2) { GSR_SH_BWD_BAND(9, 15) }
        }
      }
#undef GSR_SH_BWD_BAND
      for (int k = 3 * nb; k < 3 * K; ++k) sh[k] = 0.f;
      if constexpr (SCENE) {
        if (sc.sh_noise) {   // d(sh + n c sh)/dsh = 1 + c n
          const float* nz = sc.sh_noise + (size_t)i * (3 * K);
          for (int k = 0; k < 3 * nb; ++k) sh[k] = sh[k] * (1.0f + kSqrtPoint2 * nz[k]);
        }
      }
      float ddx, ddy, ddz;
      sh_ddir(D, x, y, z, s, ddx, ddy, ddz);
      const float dot = (x * ddx + y * ddy) + z * ddz;
      const float dvx = (ddx - x * dot) / len, dvy = (ddy - y * dot) / len, dvz = (ddz - z * dot) / len;
      dp[0] += dvx; dp[1] += dvy; dp[2] += dvz;
      dcam[0] = -dvx; dcam[1] = -dvy; dcam[2] = -dvz;
    } else if (lane < n_valid) {
      for (int k = 0; k < 3 * K; ++k) sh[k] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // accumulating: a wave without a visible Gaussian adds nothing to its rows
    if constexpr (SCENE) {
      if (!(out.accumulate && vmask == 0ull)) {
        if (sg.dc[rw.m])
          stage_rows_out<3>(sg.dc[rw.m] + wave_first * 3, 3, 0, stride, n_valid, lw, out.accumulate != 0, vmask);
        if (K > 1 && sg.rest[rw.m])
          stage_rows_out<(KT > 1 ? 3 * KT - 3 : 0)>(sg.rest[rw.m] + wave_first * (3 * K - 3), 3 * K - 3, 3, stride, n_valid,
                                                   lw, out.accumulate != 0, vmask);
      }
    } else if (out.dL_dshs && !(out.accumulate && vmask == 0ull))
      stage_sh_out<KT>(out.dL_dshs, wave_first, n_valid, K, lw, out.accumulate != 0);
  }

  float dscale[3] = {0.f, 0.f, 0.f};
  float drot[4] = {0.f, 0.f, 0.f, 0.f};
  float dc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float dsc_draw[3] = {1.f, 1.f, 1.f};   // scene: d(returned scale)/d(raw scaling)
  if (vis) {
    float c6[6];
    float R[9];
    float s3[3] = {0.f, 0.f, 0.f};
    float4 q = make_float4(1, 0, 0, 0);
    [[maybe_unused]] float4 qraw = q;
    [[maybe_unused]] float qnorm = 1.0f;
    if (g.cov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; ++k) c6[k] = g.cov3D_precomp[6 * i + k];
    } else {
      float sa[3] = {p_scale[3 * row], p_scale[3 * row + 1], p_scale[3 * row + 2]};
      q = *reinterpret_cast<const float4*>(p_rot + 4 * row);
      if constexpr (SCENE) {
        qraw = q;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float n = sc.scale_noise ? sc.scale_noise[3 * i + k] : 0.f;
          const ActScale a = act_scale(sa[k], sc.scale_noise != nullptr, n);
          sa[k] = a.out;
          // d out / d raw = exp(raw) * (1 + n sqrt(0.2)/4) where the clamp passes (torch.clamp: pre >= 0)
          dsc_draw[k] = sc.scale_noise ? (a.pre >= 0.0f ? a.act * (1.0f + n * (kSqrtPoint2 / 4.0f)) : 0.0f) : a.act;
        }
        qnorm = act_quat_norm(q);
        q = make_float4(q.x / qnorm, q.y / qnorm, q.z / qnorm, q.w / qnorm);
      }
      s3[0] = mod * sa[0]; s3[1] = mod * sa[1]; s3[2] = mod * sa[2];
      quat_to_R(q, R);
      cov3d_from(s3[0], s3[1], s3[2], R, c6);
    }
    Ewa e;
    ewa_forward(vc, px, py, pz, c6, fx, fy, limx, limy, e);

    float dS[9];
    geom_backward(vc, e, fx, fy, W, H, px, py, pz, S1, S2, S3, S4, S5, gdep, want_cam, gndx, gndy, dS, dp, dview, dproj);

    // (6) Sigma -> its parameters
    if (g.cov3D_precomp) {
      dc6[0] = dS[0]; dc6[1] = dS[1] + dS[3]; dc6[2] = dS[2] + dS[6];
      dc6[3] = dS[4]; dc6[4] = dS[5] + dS[7]; dc6[5] = dS[8];
    } else {
      sigma_backward(dS, R, s3, mod, q, dscale, drot);
      if constexpr (SCENE) {
        // through exp (+ noise, clamp) and through q = raw / |raw|:  d raw = (dq - q <q, dq>) / |raw|
#pragma unroll
        for (int k = 0; k < 3; ++k) dscale[k] = dscale[k] * dsc_draw[k];
        const float qd = ((q.x * drot[0] + q.y * drot[1]) + q.z * drot[2]) + q.w * drot[3];
        drot[0] = (drot[0] - q.x * qd) / qnorm; drot[1] = (drot[1] - q.y * qd) / qnorm;
        drot[2] = (drot[2] - q.z * qd) / qnorm; drot[3] = (drot[3] - q.w * qd) / qnorm;
      }
    }
  }

  if (vis && out.stat_denom) {   // densification statistics of this view (gs_renderer.py:1061-1065)
    out.stat_xyz_gradient_accum[i] += sqrtf(gndx * gndx + gndy * gndy);
    out.stat_denom[i] += 1.0f;
    out.stat_max_radii2D[i] = fmaxf(out.stat_max_radii2D[i], (float)radii[i]);
  }
  if constexpr (SCENE) {
    if (rw.ok) {
      out.dL_dmeans2D[3 * i] = gndx; out.dL_dmeans2D[3 * i + 1] = gndy; out.dL_dmeans2D[3 * i + 2] = 0.f;
      // gradient arriving through the RETURNED scales (the trainers' loss_scale, object_trainer.py:378-379): it
      // reaches every Gaussian, visible or not
      const bool has_gs = sg.dL_dscales_out != nullptr;
      if (has_gs) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (!vis) {
            const float n = sc.scale_noise ? sc.scale_noise[3 * i + k] : 0.f;
            const ActScale a = act_scale(p_scale[3 * row + k], sc.scale_noise != nullptr, n);
            dsc_draw[k] = sc.scale_noise ? (a.pre >= 0.0f ? a.act * (1.0f + n * (kSqrtPoint2 / 4.0f)) : 0.0f) : a.act;
          }
          dscale[k] += sg.dL_dscales_out[3 * i + k] * dsc_draw[k];
        }
      }
      if (out.accumulate && !vis) {
        float* o = sg.scaling[rw.m];
        if (has_gs && o) {
#pragma unroll
          for (int k = 0; k < 3; ++k) o[3 * row + k] += dscale[k];
        }
      } else {
        const bool acc = out.accumulate != 0;
        float gop_raw = 0.f;
        if (vis) {
          const float sg_ = act_sigmoid(sc.opacity[rw.m][row]);
          gop_raw = gop * (sg_ * (1.0f - sg_));
        }
        float* o;
        if ((o = sg.xyz[rw.m])) {
#pragma unroll
          for (int k = 0; k < 3; ++k) o[3 * row + k] = acc ? o[3 * row + k] + dp[k] : dp[k];
        }
        if ((o = sg.scaling[rw.m])) {
#pragma unroll
          for (int k = 0; k < 3; ++k) o[3 * row + k] = acc ? o[3 * row + k] + dscale[k] : dscale[k];
        }
        if ((o = sg.rotation[rw.m])) {
          float4 t = make_float4(drot[0], drot[1], drot[2], drot[3]);
          if (acc) {
            const float4 old = *reinterpret_cast<const float4*>(o + 4 * row);
            t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
          }
          *reinterpret_cast<float4*>(o + 4 * row) = t;
        }
        if ((o = sg.opacity[rw.m])) o[row] = acc ? o[row] + gop_raw : gop_raw;
      }
    }
  } else if (i < P && out.accumulate && !vis) {
    // accumulating: a culled Gaussian adds nothing; only the per-view means2D gradient is (re)written
    out.dL_dmeans2D[3 * i] = 0.f; out.dL_dmeans2D[3 * i + 1] = 0.f; out.dL_dmeans2D[3 * i + 2] = 0.f;
  } else if (i < P) {
    float gop_o = gop;
    if (out.accumulate) {
      // sum over views on the device (the reference accumulates C_batch_size views per optimizer step,
      // training/object_trainer.py:302-382); means2D is per view and is never accumulated
      dp[0] += out.dL_dmeans3D[3 * i]; dp[1] += out.dL_dmeans3D[3 * i + 1]; dp[2] += out.dL_dmeans3D[3 * i + 2];
      gop_o += out.dL_dopacities[i];
      if (out.dL_dscales) { dscale[0] += out.dL_dscales[3 * i]; dscale[1] += out.dL_dscales[3 * i + 1]; dscale[2] += out.dL_dscales[3 * i + 2]; }
      if (out.dL_drotations) {
        const float4 o = *reinterpret_cast<const float4*>(out.dL_drotations + 4 * i);
        drot[0] += o.x; drot[1] += o.y; drot[2] += o.z; drot[3] += o.w;
      }
      if (out.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dc6[k] += out.dL_dcov3D[6 * i + k];
      }
    }
    out.dL_dmeans3D[3 * i] = dp[0]; out.dL_dmeans3D[3 * i + 1] = dp[1]; out.dL_dmeans3D[3 * i + 2] = dp[2];
    out.dL_dmeans2D[3 * i] = gndx; out.dL_dmeans2D[3 * i + 1] = gndy; out.dL_dmeans2D[3 * i + 2] = 0.f;
    out.dL_dopacities[i] = gop_o;
    if (out.dL_dcolors) {
      float c0 = grgb[0], c1 = grgb[1], c2 = grgb[2];
      if (out.accumulate) { c0 += out.dL_dcolors[3 * i]; c1 += out.dL_dcolors[3 * i + 1]; c2 += out.dL_dcolors[3 * i + 2]; }
      out.dL_dcolors[3 * i] = c0; out.dL_dcolors[3 * i + 1] = c1; out.dL_dcolors[3 * i + 2] = c2;
    }
    if (out.dL_dscales) { out.dL_dscales[3 * i] = dscale[0]; out.dL_dscales[3 * i + 1] = dscale[1]; out.dL_dscales[3 * i + 2] = dscale[2]; }
    if (out.dL_drotations) *reinterpret_cast<float4*>(out.dL_drotations + 4 * i) = make_float4(drot[0], drot[1], drot[2], drot[3]);
    if (out.dL_dcov3D) {
#pragma unroll
      for (int k = 0; k < 6; ++k) out.dL_dcov3D[6 * i + k] = dc6[k];
    }
  }

  // ---- camera gradients: block reduction, one atomic per value per block
  if (want_cam) {
    float vals[27];
#pragma unroll
    for (int k = 0; k < 12; ++k) { vals[k] = dview[k]; vals[12 + k] = dproj[k]; }
    vals[24] = dcam[0]; vals[25] = dcam[1]; vals[26] = dcam[2];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float s = gsr_wave_sum_to_lane63(vals[k]);
      if (lane == 63) cam_red[wave][k] = s;
    }
    __syncthreads();
    if (tid < 27) {
      const float s = (cam_red[0][tid] + cam_red[1][tid]) + (cam_red[2][tid] + cam_red[3][tid]);
      if (tid < 12) {
        if (out.dL_dview) unsafeAtomicAdd(out.dL_dview + 4 * (tid / 3) + (tid % 3), s);
      } else if (tid < 24) {
        const int k = tid - 12, r = k / 3, c = k % 3;
        if (out.dL_dproj) unsafeAtomicAdd(out.dL_dproj + 4 * r + (c == 2 ? 3 : c), s);
      } else {
        if (out.dL_dcampos) unsafeAtomicAdd(out.dL_dcampos + (tid - 24), s);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------- K8 over several views
// The views of one optimizer step share their Gaussians (object_trainer.py:302-382): one pass reads every
// parameter row once, loops over the views' cameras / K7 partials, sums the gradients in registers and writes them
// once -- 4 views: ~185 B per Gaussian and view instead of ~700 (the SH rows dominate both the reads and the writes).
// dL/dSigma is summed over the views before the (view-independent) step to scales / quaternion.
struct K8Views {
  int32_t nv;
  const float* viewmatrix[GSR_MAX_BATCH_VIEWS];
  const float* projmatrix[GSR_MAX_BATCH_VIEWS];
  const float* campos[GSR_MAX_BATCH_VIEWS];
  float tanfovx[GSR_MAX_BATCH_VIEWS];
  float tanfovy[GSR_MAX_BATCH_VIEWS];
  int32_t sh_degree[GSR_MAX_BATCH_VIEWS];   // the active degree may differ per view (scene_render's sh_deg_aug)
  const float* dyn[GSR_MAX_BATCH_VIEWS];    // GsrView.dynamic of every view (NULL: the by-value entries above)
  const int32_t* radii[GSR_MAX_BATCH_VIEWS];
  float* partials[GSR_MAX_BATCH_VIEWS];
  // GsrGrads.reach of every view (all set or all NULL) and its contract: restore = zero the consumed sums and marks again
  unsigned long long* reach[GSR_MAX_BATCH_VIEWS];
  int32_t restore;
  float* dL_dmeans2D[GSR_MAX_BATCH_VIEWS];
  // per-view scales (the trainers add fresh noise to the activated scales of every view, scene_gaussian.py:1004-1008):
  // then every view has its own scales tensor and its own scale gradient; the other parameters are shared
  int32_t per_view_scales;
  const float* scales[GSR_MAX_BATCH_VIEWS];
  float* dL_dscales[GSR_MAX_BATCH_VIEWS];
  // scene input (raw leaves): per-view noise samples, per-view gradient arriving through the returned scales
  const float* scale_noise[GSR_MAX_BATCH_VIEWS];
  const float* sh_noise[GSR_MAX_BATCH_VIEWS];
  const float* dL_dscales_out[GSR_MAX_BATCH_VIEWS];
  // bit k set: view k's densification statistics count (the reference's trainers use the LAST view of a step only,
  // object_trainer.py:386-390; a caller sets the stat_* pointers on the GsrGrads entries of the views that count)
  uint32_t stat_mask;
};

// ------------------------------------------------------------------------------- K8, sparse over the Gaussians
// K7 reaches few Gaussians: behind the first opaque layers nothing receives a gradient (C3: 4-5 % of the Gaussians
// per view, 16 % in the union of four views; 2 M Gaussians: 8 %). A (Gaussian, view) pair whose ten K7 sums are all
// zero contributes exactly zero to every output of K8 -- every term of the chain rule is a product with one of them --
// so the ~1 200 instructions per pair, and the 236 B parameter row, are only worth touching for the Gaussians some
// view reached. Those are scattered (nearly every wave holds one), so the REACHED form of the kernel below compacts
// them inside the workgroup, which owns kK8Block = 1 024 consecutive Gaussians:
//   A. every thread classifies four Gaussians (radii + K7's marks, GsrGrads.reach, of every view; without marks: the K7
//      sums themselves) -> "reached by some view?". The workgroup clears the gradient rows (and the per-view rows) of its
//      1 024 Gaussians with coalesced stores (unless accumulating); a Gaussian nothing reached gets its visibility
//      statistics here and is done;
//   B. the reached ones are listed in LDS (ballot / popcount prefix, ascending) and the four waves walk the list in rounds
//      of 256 (wave slots rotated by the block index so that partial rounds do not pile up on one SIMD), running the dense
//      chain rule on them -- same arithmetic, same summation order over the views, bit-identical rows -- and writing each
//      row over the cleared one.
// Where a workgroup spends its time (tools/k8_stamps.py: realtime stamps inside the kernel; C3, 4 views, 489 workgroups of
// 1 024 Gaussians, 161 of them reached on average, all resident at once; us since the workgroup's entry, mean):
//   classified 14.5 | parameters + SH row in 27.6 | rows cleared 47.9 | views done 54.0 58.4 62.5 66.6 | rows stored 69.2
// and the launch takes 93 (the workgroups with two rounds). What the numbers say:
//  * the chain rule is ~1 400 instructions per view and lane and runs at ~7 cycles per instruction -- one wave per SIMD,
//    every instruction dependent on the last: 4.3 us per view whatever the occupancy of the lanes;
//  * every memory round trip on the critical path costs 3-10 us while all workgroups are in the same phase, so requests
//    are issued in batches (all slices and views of the classification at once; parameters, SH row and the first view's
//    sums at once; the next view's sums before the current view's arithmetic) and the per-view constants come through
//    scalar loads (load_view_const): on gfx9 a vector load issued behind a vector store waits for the store's
//    acknowledgement, which is how the per-view stores used to serialise the views;
//  * the 142 MB of zeros are 20 us of HBM writes during which nothing else progresses (all workgroups clear at the same
//    time). Next lead: clear from the idle wave(s) only, skipping the reached rows, so that it overlaps the chain rule;
//  * 230 -> 260 registers (the prefetch) halved the occupancy and cost 88 -> 121 us: amdgpu_waves_per_eu(2) pins it.
// Round 2 gave a workgroup 256 Gaussians and ONE wave of it the ~40 reached ones (three shifts of workgroups, 101 us);
// C3 in the opacity-0.1 initial state (67 % reached: three or four rounds per workgroup) is slower in this form than
// with the old dense fallback (151 vs 124 us per launch) -- 3 % of that step.
// Forms that were measured and dropped: (i) chunks of (Gaussian, view) pairs dealt to the waves with the results
// summed by ds_add_f32 into a 256-row LDS tile (0.5 ns per pair: slower than dense once 5 % are active); (ii) a separate
// classify launch appending to global lists kept in spare words of the K7 sums, then the dense kernel over the lists
// (one list: 31 000 same-address atomics at 2 M Gaussians; one list per 8 192 Gaussians: the live workgroups of the
// second launch all land on three of the eight XCDs); (iii) lane j of every wave holding the j-th reached Gaussian,
// wave w running view w, results added to an LDS tile view after view between barriers.
#ifndef GSR_K8_BLOCK
#define GSR_K8_BLOCK 1024
#endif
constexpr int kK8Block = GSR_K8_BLOCK;       // Gaussians per workgroup of the REACHED form (a multiple of 256)

// one lane's row of F floats -> global memory (dword-aligned 16-byte pieces)
template <int F>
__device__ __forceinline__ void store_row(float* __restrict__ dst, const float* src, bool accumulate) {
#pragma unroll
  for (int q = 0; q + 3 < F; q += 4) {
    gsr_f4u t;
    t.x = src[q]; t.y = src[q + 1]; t.z = src[q + 2]; t.w = src[q + 3];
    if (accumulate) {
      const gsr_f4u o = *reinterpret_cast<const gsr_f4u*>(dst + q);
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    *reinterpret_cast<gsr_f4u*>(dst + q) = t;
  }
#pragma unroll
  for (int q = F & ~3; q < F; ++q) dst[q] = accumulate ? dst[q] + src[q] : src[q];
}

// zeros over n floats starting at dst (n < 2^31), the 256 threads of the block together; VEC: dst is 16-byte aligned
template <bool VEC>
__device__ __forceinline__ void block_zero(float* __restrict__ dst, int n) {
  const int tid = threadIdx.x;
  if constexpr (VEC) {
    const int n4 = n >> 2;
    for (int q = tid; q < n4; q += 256) reinterpret_cast<float4*>(dst)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = (n4 << 2) + tid; q < n; q += 256) dst[q] = 0.f;
  } else {
    for (int q = tid; q < n; q += 256) dst[q] = 0.f;
  }
}

// zeros over the rows of F floats [0, n) at dst (n <= 64) whose bit in `skip` is clear, one wave; rows stay whole: a
// 16-byte store never straddles into a skipped row (F % 4 == 0 and dst 16-byte aligned, or 4-byte stores)
template <int F>
__device__ __forceinline__ void wave_zero_rows(float* __restrict__ dst, int n, unsigned long long skip, int lane) {
  if ((~skip & (n >= 64 ? ~0ull : ((1ull << n) - 1ull))) == 0ull) return;       // (uniform: nothing to clear in this chunk)
  if constexpr (F % 4 == 0) {
    constexpr int Q = F / 4;
    float4* d = reinterpret_cast<float4*>(dst);
    for (int q = lane; q < n * Q; q += 64)
      if (!((skip >> (q / Q)) & 1ull)) d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int f = lane; f < n * F; f += 64)
      if (!((skip >> (f / F)) & 1ull)) dst[f] = 0.f;
  }
}

template <int KT, bool PVS, bool REACHED = false, int KPER = 4>   // PVS: per-view scales; REACHED: the sparse form described above;
// KPER: 256-Gaussian slices per workgroup of the REACHED form (4: 1 024 Gaussians, for grids that fill the chip once; 1: small P)
// (two waves per SIMD = two workgroups per CU: one resident wave of workgroups at 500 k Gaussians. Without the attribute the
//  allocator took 260 registers, one workgroup per CU, and the kernel ran its workgroups in two shifts: 88 -> 121 us)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
k_preprocess_bwd_views(const GsrView v, const GsrGaussians g, const K8Views vb, const GsrGrads out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int kPer = REACHED ? KPER : 1;                  // Gaussians classified per thread
  __shared__ uint32_t wcnt[REACHED ? 4 * kPer : 1];
  __shared__ uint16_t reached_list[REACHED ? 256 * KPER : 1];
  __shared__ unsigned long long lmask[REACHED ? 4 * kPer : 1];   // reached Gaussians of the 64-row chunk (slice, wave)
  __shared__ unsigned long long vmask[REACHED ? GSR_MAX_BATCH_VIEWS : 1][REACHED ? 4 * kPer : 1];   // ... per view (K7's marks)
  __shared__ uint32_t zero_next;                                  // next chunk nobody has cleared yet
  // GsrGrads.zero_outside: rows of chunk c that MAY be non-zero on entry (the previous writer's reached_mask; all ones: unknown)
  __shared__ unsigned long long omask[REACHED ? 4 * kPer : 1];
  constexpr int F = 3 * KT;
  const int W = v.image_width, H = v.image_height;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t P = v.P;
  // REACHED: the workgroup's 4 kPer chunks of 64 consecutive Gaussians lie gridDim.x * 64 apart -- how many Gaussians the
  // views reached varies along the index (C3: 161 of 1 024 on average, 566 in the densest contiguous block, 455 with four
  // slices of 256, and the kernel ends with the workgroup that has the most rounds); chunks from sixteen places average it.
  // Chunk c = 4 t + wave belongs to the lanes of `wave` in slice t of phase A.
  const auto chunk_base = [&](int c) { return ((int64_t)c * gridDim.x + blockIdx.x) * 64; };
  const int64_t first = REACHED ? chunk_base(0) : (int64_t)blockIdx.x * 256;
  int64_t i = first + tid;
  bool ok = i < P;
  constexpr bool sparse = REACHED;
  int cnt = 0;           // REACHED: entries of reached_list
#ifdef GSR_K8_STAMPS
  unsigned long long ts[12];
  int nts = 0;
#define GSR_K8_STAMP() do { if (nts < 12) ts[nts++] = wall_clock64(); } while (0)
#else
#define GSR_K8_STAMP() do { } while (0)
#endif
  GSR_K8_STAMP();   // 0: entry
  // phase A's verdicts (slice t of a thread = Gaussian chunk_base(4 t + wave) + lane)
  bool reached[kPer];
  unsigned long long rmask[kPer];
  if constexpr (REACHED) {
    // ---- A: which of the workgroup's Gaussians did some view reach?
    // The loads of all slices of a view are issued together (every one of them is a memory latency if it is consumed
    // where it is issued: 2 x 4 x 4 dependent round trips were 17 of this kernel's 95 us).
#pragma unroll
    for (int t = 0; t < kPer; ++t) { reached[t] = false; rmask[t] = 0ull; }
    if (vb.reach[0]) {
      // K7 marked the Gaussians it committed sums for (GsrGrads.reach: one bit per Gaussian and view): one 8-byte word per
      // (view, chunk) -- thread 4 kPer vv + c fetches it -- instead of the 40-byte sums of every visible Gaussian
      static_assert(GSR_MAX_BATCH_VIEWS * 4 * kPer <= 256, "one thread per (view, chunk) word");
      if (tid < vb.nv * 4 * kPer) {
        const int vv = tid / (4 * kPer), c = tid % (4 * kPer);
        const int64_t r0 = chunk_base(c);
        vmask[vv][c] = r0 < P ? vb.reach[vv][r0 >> 6] : 0ull;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < kPer; ++t) {
        for (int vv = 0; vv < vb.nv; ++vv) rmask[t] |= vmask[vv][4 * t + wave];
        reached[t] = (rmask[t] >> lane) & 1ull;
      }
    } else {
      // without marks: the sums themselves (the loads of all slices of four views are issued together)
      for (int v0 = 0; v0 < vb.nv; v0 += 4) {
        int32_t r[4][kPer];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int t = 0; t < kPer; ++t) {
            const int64_t it = chunk_base(4 * t + wave) + lane;
            r[u][t] = ((v0 + u < vb.nv) && (it < P)) ? vb.radii[v0 + u][it] : 0;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int t = 0; t < kPer; ++t) {
            if (r[u][t] > 0) {
              float4 pa, pb, pc;
              partial_rows(partial_load(vb.partials[v0 + u], chunk_base(4 * t + wave) + lane), pa, pb, pc);
              reached[t] = reached[t] || (pa.x != 0.f) || (pa.y != 0.f) || (pa.z != 0.f) || (pa.w != 0.f) || (pb.x != 0.f) ||
                           (pb.y != 0.f) || (pb.z != 0.f) || (pb.w != 0.f) || (pc.x != 0.f) || (pc.y != 0.f) ||
                           (pc.z != 0.f) || (pc.w != 0.f);
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < kPer; ++t) rmask[t] = __ballot(reached[t]);
    }
#pragma unroll
    for (int t = 0; t < kPer; ++t)
      if (lane == 0) { wcnt[t * 4 + wave] = (uint32_t)__popcll(rmask[t]); lmask[t * 4 + wave] = rmask[t]; }
    if (tid < 4 * kPer) {
      // (the old word of chunk c is read here, by the workgroup that owns the chunk, before wave_exit stores the new one)
      const int64_t r0 = chunk_base(tid);
      omask[tid] = (out.zero_outside && !out.accumulate && out.reached_mask && r0 < P)
                       ? reinterpret_cast<const unsigned long long*>(out.reached_mask)[r0 >> 6] : ~0ull;
    }
    if (tid == 0) zero_next = 0u;
    GSR_K8_STAMP();   // 1: classified
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kPer; ++t) {
      uint32_t off = 0;
      for (int q = 0; q < t * 4 + wave; ++q) off += wcnt[q];
      if (reached[t]) reached_list[off + (uint32_t)__popcll(rmask[t] & ((1ull << lane) - 1ull))] = (uint16_t)(t * 256 + tid);
    }
    for (int q = 0; q < 4 * kPer; ++q) cnt += (int)wcnt[q];
    __syncthreads();
    // (everything this phase STORES comes after the first loads of phase B, in the epilogue of round 0: vector-memory
    //  operations return in order on gfx9, so a load issued behind a store waits for the store's write acknowledgement)
  }
  const float mod = v.scale_modifier;
  constexpr int stride = F | 1;
  float* lw = lds + wave * (64 * stride);
  float* sh = lw + lane * stride;
  // ---- B (REACHED): rounds of 256 list entries, wave slot rotated by the block index; otherwise one pass, thread = Gaussian
  for (int round = 0;; round += 256) {
  bool active = true;    // REACHED: this wave has entries in this round (uniform over the wave)
  int entry = 0;         // REACHED: the lane's list entry
  if constexpr (REACHED) {
    const int chunk = round + ((wave + 4 - (int)(blockIdx.x & 3u)) & 3) * 64;
    active = chunk < cnt;
    ok = active && (chunk + lane < cnt);
    entry = ok ? (int)reached_list[chunk + lane] : 0;
    i = chunk_base(entry >> 6) + (entry & 63);      // (entry = 256 t + tid = 64 (4 t + wave) + lane)
  }
  const int64_t wave_first = (int64_t)blockIdx.x * 256 + wave * 64;      // (the dense form's coalesced write-back)
  const int n_valid = (int)min((int64_t)64, max((int64_t)0, P - wave_first));

  bool any = ok;         // REACHED: a listed Gaussian is visible in the view that reached it
  if constexpr (!REACHED) {
    any = false;
    for (int vv = 0; vv < vb.nv; ++vv) any = any || (ok && vb.radii[vv][i] > 0);
  }
  const unsigned long long amask = __ballot(any);

  // one view ahead: radius, K7's mark and (REACHED: unconditionally -- an unmarked row is zero or ignored) the K7 sums of
  // the next view are requested before the current view's arithmetic and before its stores
  int32_t r_nx = 0;
  uint32_t mk_nx = 1u;
  PartialRaw p_nx = partial_none();   // (raw words: the f64 -> f32 conversions wait for the loads, so they happen where the row is consumed)
  const auto prefetch_view = [&](int vv) {
    r_nx = ok ? vb.radii[vv][i] : 0;
    if (!vb.reach[0]) mk_nx = 1u;
    else if constexpr (REACHED) mk_nx = (uint32_t)((vmask[vv][entry >> 6] >> (entry & 63)) & 1ull);
    else mk_nx = ok ? (uint32_t)((vb.reach[vv][i >> 6] >> (i & 63)) & 1ull) : 0u;
    if constexpr (REACHED) {
      // (only the views that MARKED the Gaussian: a listed Gaussian is reached by 1.3 of the 4 views of a step on average, and
      //  a row is 96 bytes since the sums are doubles -- round 3 loaded the 48-byte rows of all views unconditionally. The mark
      //  comes from the LDS copy of K7's bit words: no memory round trip in front of the load)
      if (ok && mk_nx) p_nx = partial_load(vb.partials[vv], i);
      else p_nx = partial_none();
    }
  };
  // every request of the round goes out before anything is consumed: one memory latency, not four
  float px = 0, py = 0, pz = 0;
  float R[9], c6[6], s3[3] = {0.f, 0.f, 0.f};
  float4 q = make_float4(1, 0, 0, 0);
  if (any) {
    px = g.means3D[3 * i]; py = g.means3D[3 * i + 1]; pz = g.means3D[3 * i + 2];
    q = *reinterpret_cast<const float4*>(g.rotations + 4 * i);
    if constexpr (!PVS) { s3[0] = g.scales[3 * i]; s3[1] = g.scales[3 * i + 1]; s3[2] = g.scales[3 * i + 2]; }
  }
  prefetch_view(0);
  if (any) {
    load_row<F>(g.shs + (size_t)i * F, sh);      // the lane's own SH row, kept (read only) in its LDS row
    quat_to_R(q, R);
    if constexpr (!PVS) {
      s3[0] = mod * s3[0]; s3[1] = mod * s3[1]; s3[2] = mod * s3[2];
      cov3d_from(s3[0], s3[1], s3[2], R, c6);
    }
  }
  GSR_K8_STAMP();   // 2: parameters + SH row in, view 0 requested
  // ---- what a wave does when it has no (more) list entries: the words of the exchange mask and the visibility statistics
  // of its lanes' Gaussians, then zeros over the rows nothing reached -- 64-row chunks claimed from a counter, so the
  // waves that are idle from the start clear while the others run the chain rule, and nothing the chain rule loads
  // queues behind those stores (the reached rows are written by the chain rule alone: the two never touch the same row)
  const auto wave_exit = [&]() {
    if (vb.restore && vb.reach[0] && tid < vb.nv * 4 * kPer) {      // GsrGrads.scratch_clean: K7's marks back to zero
      const int64_t r0 = chunk_base(tid % (4 * kPer));
      if (r0 < P) vb.reach[tid / (4 * kPer)][r0 >> 6] = 0ull;
    }
#pragma unroll
    for (int t = 0; t < kPer; ++t) {
      const int64_t base = chunk_base(4 * t + wave), it = base + lane;
      const unsigned long long rm = lmask[t * 4 + wave];      // (from LDS: nothing of phase A stays in registers)
      if (lane == 0 && out.reached_mask && base < P) {
        // the rows a gradient exchange has to move (GsrGrads.reached_mask): one word per wave and slice, owned by this wave
        unsigned long long* w = reinterpret_cast<unsigned long long*>(out.reached_mask) + (base >> 6);
        if (out.accumulate) *w |= rm; else *w = rm;
      }
      if (!((rm >> lane) & 1ull) && it < P && out.stat_denom) {   // visibility statistics (the reached ones: in the chain rule)
        float n = 0.f, rmax = 0.f;
        for (int vv = 0; vv < vb.nv; ++vv)
          if ((vb.stat_mask >> vv) & 1u) {
            const int32_t r = vb.radii[vv][it];
            if (r > 0) { n += 1.0f; rmax = fmaxf(rmax, (float)r); }
          }
        if (n > 0.f) {
          out.stat_denom[it] += n;
          out.stat_max_radii2D[it] = fmaxf(out.stat_max_radii2D[it], rmax);
        }
      }
    }
    for (;;) {
      uint32_t c = 0;
      if (lane == 0) c = atomicAdd(&zero_next, 1u);
      c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
      if (c >= (uint32_t)(4 * kPer)) break;
      const int64_t r0 = chunk_base((int)c);
      const int n = (int)max((int64_t)0, min((int64_t)64, P - r0));
      if (n == 0) continue;
      // rows to clear: the ones nothing reached now -- of those, with GsrGrads.zero_outside, only the ones the previous writer
      // of these buffers reached (everything else is zero already)
      const unsigned long long known0 = out.accumulate ? 0ull : ~omask[c];
      const unsigned long long skip = lmask[c] | ((out.zero_outside & 1) ? known0 : 0ull);
      const unsigned long long skip_pv = lmask[c] | ((out.zero_outside & 2) ? known0 : 0ull);
      if (!out.accumulate) {
        wave_zero_rows<F>(out.dL_dshs + r0 * F, n, skip, lane);
        wave_zero_rows<3>(out.dL_dmeans3D + r0 * 3, n, skip, lane);
        wave_zero_rows<1>(out.dL_dopacities + r0, n, skip, lane);
        if constexpr (!PVS) wave_zero_rows<3>(out.dL_dscales + r0 * 3, n, skip, lane);
        wave_zero_rows<4>(out.dL_drotations + r0 * 4, n, skip, lane);
      }
      for (int vv = 0; vv < vb.nv; ++vv) {      // the per-view rows: what the chain rule would produce from zeros
#ifdef GSR_K8_STAMPS
        if (c == 0 && vv == 0) continue;        // (the stamps are left there)
#endif
        wave_zero_rows<3>(vb.dL_dmeans2D[vv] + r0 * 3, n, skip_pv, lane);
        if constexpr (PVS) wave_zero_rows<3>(vb.dL_dscales[vv] + r0 * 3, n, skip_pv, lane);
      }
    }
  };
  GSR_K8_STAMP();   // 3
  if constexpr (REACHED) {
    if (!active) { wave_exit(); break; }
  }
  float drot[4] = {0.f, 0.f, 0.f, 0.f};

  float dsh[F];
#pragma unroll
  for (int k = 0; k < F; ++k) dsh[k] = 0.f;
  float dp[3] = {0.f, 0.f, 0.f}, dS[9], gop = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) dS[k] = 0.f;

  for (int vv = 0; vv < vb.nv; ++vv) {
    const int32_t rad_v = r_nx;
    const bool take = mk_nx != 0u;                      // (an unmarked Gaussian's sums are zero: not used)
    const PartialRaw p_cur = p_nx;
    float4 pa, pb, pc;
    if (vv + 1 < vb.nv) prefetch_view(vv + 1);
    if constexpr (!REACHED) {     // (the wave owns the mark word of its 64 Gaussians; the REACHED form: wave_exit)
      if (vb.restore && vb.reach[0] && ok && (i & 63) == 0) vb.reach[vv][i >> 6] = 0ull;
    }
    const bool vis = ok && (rad_v > 0);
    // A view that SAW the Gaussian but composited nothing of it (no mark: its ten K7 sums are zero) contributes exact zeros
    // to every output: its chain rule -- ~1 400 dependent instructions -- is skipped (round 4; a listed Gaussian is marked by
    // 1.3 of the 4 views of a step on average at C3, and the kernel is bound by that dependent chain). What does not depend
    // on the sums still happens below: the visibility statistics, the zero rows of dL/dmeans2D and of per-view scales.
    const bool work = vis && take;
    float gndx = 0.f, gndy = 0.f;
    if (work) {
      ViewConst vc;
      load_view_const(vb.viewmatrix[vv], vb.projmatrix[vv], vb.campos[vv], vc);
      const ViewDyn vd = view_dyn(vb.dyn[vv], vb.tanfovx[vv], vb.tanfovy[vv], vb.sh_degree[vv]);
      const float tfx = vd.tanfovx, tfy = vd.tanfovy;
      const int D = vd.sh_degree;
      const float fx = (float)W / (2.0f * tfx), fy = (float)H / (2.0f * tfy);
      const float limx = 1.3f * tfx, limy = 1.3f * tfy;
      if constexpr (!REACHED) {
        partial_rows(partial_load(vb.partials[vv], i), pa, pb, pc);
      } else {
        partial_rows(p_cur, pa, pb, pc);
      }
      pc.x += pc.z; pc.y += pc.w;      // (K7 commits the last two sums from the two halves of a wave: render.hip, reduce10)
      if (vb.restore) partial_zero(vb.partials[vv], i);     // GsrGrads.scratch_clean: leave the scratch as it was found
      gop += pb.y;
      const float grgb[3] = {pb.z, pb.w, pc.x};
      // (1) colour -> SH coefficients, view direction
      {
        const float vx = px - vc.cam[0], vy = py - vc.cam[1], vz = pz - vc.cam[2];
        const float len = sqrtf((vx * vx + vy * vy) + vz * vz);
        const float x = vx / len, y = vy / len, z = vz / len;
        float b[16];
        sh_basis(D, x, y, z, b);
        float acc[3];
        sh_colour_n<KT>(D, sh, b, acc);
        float gch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gch[c] = (acc[c] + 0.5f < 0.0f) ? 0.0f : grgb[c];   // K1's clamp decision
        float s[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] = 0.f;
#define GSR_SH_ACC_BAND(K0, K1)                                                                      \
  _Pragma("unroll") for (int k = K0; k <= K1; ++k) {                                                 \
    s[k] = (sh[3 * k] * gch[0] + sh[3 * k + 1] * gch[1]) + sh[3 * k + 2] * gch[2];                   \
    dsh[3 * k] += b[k] * gch[0]; dsh[3 * k + 1] += b[k] * gch[1]; dsh[3 * k + 2] += b[k] * gch[2];   \
  }
        GSR_SH_ACC_BAND(0, 0)
        if constexpr (KT >= 4) {
          if (D > 0) {
            GSR_SH_ACC_BAND(1, 3)
            if constexpr (KT >= 9) {
              if (D > 1) {
                GSR_SH_ACC_BAND(4, 8)
                if constexpr (KT >= 16) {
                  if (D > 2) { GSR_SH_ACC_BAND(9, 15) }
                }
              }
            }
          }
        }
#undef GSR_SH_ACC_BAND
        float ddx, ddy, ddz;
        sh_ddir(D, x, y, z, s, ddx, ddy, ddz);
        const float dot = (x * ddx + y * ddy) + z * ddz;
        dp[0] += (ddx - x * dot) / len; dp[1] += (ddy - y * dot) / len; dp[2] += (ddz - z * dot) / len;
      }
      // (2)-(5) geometry of this view
      if constexpr (PVS) {
        const float* sc = vb.scales[vv];
        s3[0] = mod * sc[3 * i]; s3[1] = mod * sc[3 * i + 1]; s3[2] = mod * sc[3 * i + 2];
        cov3d_from(s3[0], s3[1], s3[2], R, c6);
      }
      Ewa e;
      ewa_forward(vc, px, py, pz, c6, fx, fy, limx, limy, e);
      float dSv[9], dview[12], dproj[12];
      geom_backward(vc, e, fx, fy, W, H, px, py, pz, pa.x, pa.y, pa.z, pa.w, pb.x, pc.y, false, gndx, gndy, dSv, dp, dview,
                    dproj);
      if constexpr (PVS) {      // this view's own scales: its own scale gradient; the quaternion's is summed
        float ds_v[3], dr_v[4];
        sigma_backward(dSv, R, s3, mod, q, ds_v, dr_v);
        float* o = vb.dL_dscales[vv];
        o[3 * i] = ds_v[0]; o[3 * i + 1] = ds_v[1]; o[3 * i + 2] = ds_v[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) drot[k] += dr_v[k];
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) dS[k] += dSv[k];
      }
    }
    if (vis && out.stat_denom && ((vb.stat_mask >> vv) & 1u)) {
      out.stat_xyz_gradient_accum[i] += sqrtf(gndx * gndx + gndy * gndy);
      out.stat_denom[i] += 1.0f;
      out.stat_max_radii2D[i] = fmaxf(out.stat_max_radii2D[i], (float)rad_v);
    }
    if (ok) {
      float* m2 = vb.dL_dmeans2D[vv];
      m2[3 * i] = gndx; m2[3 * i + 1] = gndy; m2[3 * i + 2] = 0.f;
      if (PVS && !work) {
        float* o = vb.dL_dscales[vv];
        o[3 * i] = 0.f; o[3 * i + 1] = 0.f; o[3 * i + 2] = 0.f;
      }
    }
    GSR_K8_STAMP();   // 4..: a view done
  }

  float dscale[3] = {0.f, 0.f, 0.f};
  if (any && !PVS) sigma_backward(dS, R, s3, mod, q, dscale, drot);

  if (sparse) {   // the lane's own row over the cleared one
    if (ok && out.dL_dshs && !(out.accumulate && !any)) store_row<F>(out.dL_dshs + (size_t)i * F, dsh, out.accumulate != 0);
  } else {
    // gradient rows -> LDS (zeros for Gaussians no view saw) -> coalesced write-back
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < n_valid) {
#pragma unroll
      for (int k = 0; k < F; ++k) sh[k] = dsh[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (out.dL_dshs && !(out.accumulate && amask == 0ull))
      stage_sh_out<KT>(out.dL_dshs, wave_first, n_valid, KT, lw, out.accumulate != 0);
  }

  if (ok && !(out.accumulate && !any)) {
    if (out.accumulate) {
      dp[0] += out.dL_dmeans3D[3 * i]; dp[1] += out.dL_dmeans3D[3 * i + 1]; dp[2] += out.dL_dmeans3D[3 * i + 2];
      gop += out.dL_dopacities[i];
      if constexpr (!PVS) {
        dscale[0] += out.dL_dscales[3 * i]; dscale[1] += out.dL_dscales[3 * i + 1]; dscale[2] += out.dL_dscales[3 * i + 2];
      }
      const float4 o = *reinterpret_cast<const float4*>(out.dL_drotations + 4 * i);
      drot[0] += o.x; drot[1] += o.y; drot[2] += o.z; drot[3] += o.w;
    }
    out.dL_dmeans3D[3 * i] = dp[0]; out.dL_dmeans3D[3 * i + 1] = dp[1]; out.dL_dmeans3D[3 * i + 2] = dp[2];
    out.dL_dopacities[i] = gop;
    if constexpr (!PVS) {
      out.dL_dscales[3 * i] = dscale[0]; out.dL_dscales[3 * i + 1] = dscale[1]; out.dL_dscales[3 * i + 2] = dscale[2];
    }
    *reinterpret_cast<float4*>(out.dL_drotations + 4 * i) = make_float4(drot[0], drot[1], drot[2], drot[3]);
  }
#ifdef GSR_K8_STAMPS
  GSR_K8_STAMP();     // rows stored (issued)
  if constexpr (REACHED) {
    // timing experiment only (results are destroyed): wave slot 0 of every workgroup leaves its stamps, as 10 ns ticks since
    // entry, in the first floats of view 0's dL_dmeans2D rows of this workgroup
    if (round == 0 && lane == 0 && ((wave + 4 - (int)(blockIdx.x & 3u)) & 3) == 0) {
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long tend = wall_clock64();
      float* o = vb.dL_dmeans2D[0] + first * 3;
      for (int k = 0; k < nts; ++k) o[k] = (float)(ts[k] - ts[0]);
      o[nts] = (float)(tend - ts[0]);
      o[14] = (float)cnt; o[15] = (float)nts;
    }
  }
#endif
  if constexpr (!REACHED) break;
  else if (round + 256 + ((wave + 4 - (int)(blockIdx.x & 3u)) & 3) * 64 >= cnt) { wave_exit(); break; }
  }   // rounds
}



// K8 over several views of a SCENE: the raw rows are read once, every view has its own (possibly noisy) scales, the
// gradients of the raw leaves are summed over the views in registers and written once per model tensor.
template <int KT>
__global__ void __launch_bounds__(256)
k_preprocess_bwd_views_scene(const GsrView v, const SceneTab sc, const SceneGradTab sg, const K8Views vb,
                             const GsrGrads out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int F = 3 * KT;
  const int W = v.image_width, H = v.image_height;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const Rows rw = resolve_rows<true>(sc, v.P);
  const int64_t i = rw.i, row = rw.row, wave_first = rw.wave_row;
  const int m = rw.m, n_valid = rw.n_valid;
  const bool ok = rw.ok;
  const float mod = v.scale_modifier;
  constexpr int stride = F | 1;
  float* lw = lds + wave * (64 * stride);
  float* sh = lw + lane * stride;

  bool any = false, has_gs = false;
  for (int vv = 0; vv < vb.nv; ++vv) {
    any = any || (ok && vb.radii[vv][i] > 0);
    has_gs = has_gs || (vb.dL_dscales_out[vv] != nullptr);
  }
  const unsigned long long amask = __ballot(any);

  float px = 0, py = 0, pz = 0, R[9], aact[3] = {0.f, 0.f, 0.f}, qnorm = 1.0f;
  float4 q = make_float4(1, 0, 0, 0);
  if (ok && (any || has_gs)) {
#pragma unroll
    for (int k = 0; k < 3; ++k) aact[k] = expf(sc.scaling[m][3 * row + k]);
  }
  if (any) {
    const float* xyz = sc.xyz[m];
    px = xyz[3 * row]; py = xyz[3 * row + 1]; pz = xyz[3 * row + 2];
    q = *reinterpret_cast<const float4*>(sc.rotation[m] + 4 * row);
    qnorm = act_quat_norm(q);
    q = make_float4(q.x / qnorm, q.y / qnorm, q.z / qnorm, q.w / qnorm);
    quat_to_R(q, R);
    load_row<3>(sc.dc[m] + row * 3, sh);                     // raw SH row, kept (read only) in the lane's LDS row
    if constexpr (KT > 1) load_row<F - 3>(sc.rest[m] + row * (F - 3), sh + 3);
  }

  float dsh[F];
#pragma unroll
  for (int k = 0; k < F; ++k) dsh[k] = 0.f;
  float dp[3] = {0.f, 0.f, 0.f}, gop = 0.f, drot[4] = {0.f, 0.f, 0.f, 0.f}, dsraw[3] = {0.f, 0.f, 0.f};

  for (int vv = 0; vv < vb.nv; ++vv) {
    const bool vis = ok && (vb.radii[vv][i] > 0);
    float gndx = 0.f, gndy = 0.f;
    // this view's scales and their derivative w.r.t. the raw (log) scaling
    float sa[3] = {0.f, 0.f, 0.f}, dsc[3] = {0.f, 0.f, 0.f};
    if (ok && (vis || vb.dL_dscales_out[vv])) {
      const float* sn = vb.scale_noise[vv];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float n = sn ? sn[3 * i + k] : 0.f;
        const float pre = sn ? aact[k] + n * ((kSqrtPoint2 * aact[k]) / 4.0f) : aact[k];
        sa[k] = sn ? fmaxf(pre, 0.0f) : aact[k];
        dsc[k] = sn ? (pre >= 0.0f ? aact[k] * (1.0f + n * (kSqrtPoint2 / 4.0f)) : 0.0f) : aact[k];
      }
      if (vb.dL_dscales_out[vv]) {   // loss on the returned scales: reaches every Gaussian
        const float* gs = vb.dL_dscales_out[vv];
#pragma unroll
        for (int k = 0; k < 3; ++k) dsraw[k] += gs[3 * i + k] * dsc[k];
      }
    }
    if (vis) {
      ViewConst vc;
      load_view_const(vb.viewmatrix[vv], vb.projmatrix[vv], vb.campos[vv], vc);
      const ViewDyn vd = view_dyn(vb.dyn[vv], vb.tanfovx[vv], vb.tanfovy[vv], vb.sh_degree[vv]);
      const float tfx = vd.tanfovx, tfy = vd.tanfovy;
      const int D = vd.sh_degree;
      const float fx = (float)W / (2.0f * tfx), fy = (float)H / (2.0f * tfy);
      float4 pa, pb, pc;
      partial_rows(partial_load(vb.partials[vv], i), pa, pb, pc);
      pc.x += pc.z; pc.y += pc.w;      // (see k_preprocess_bwd)
      gop += pb.y;
      const float grgb[3] = {pb.z, pb.w, pc.x};
      // (1) colour -> SH coefficients (through this view's noise), view direction
      {
        const float vx = px - vc.cam[0], vy = py - vc.cam[1], vz = pz - vc.cam[2];
        const float len = sqrtf((vx * vx + vy * vy) + vz * vz);
        const float x = vx / len, y = vy / len, z = vz / len;
        float b[16];
        sh_basis(D, x, y, z, b);
        const float* nz = vb.sh_noise[vv] ? vb.sh_noise[vv] + (size_t)i * F : nullptr;
        float shv[F];
#pragma unroll
        for (int k = 0; k < F; ++k) shv[k] = nz ? sh[k] + nz[k] * (kSqrtPoint2 * sh[k]) : sh[k];
        float acc[3];
        sh_colour_n<KT>(D, shv, b, acc);
        float gch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gch[c] = (acc[c] + 0.5f < 0.0f) ? 0.0f : grgb[c];
        float s[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] = 0.f;
        const int nb = (D + 1) * (D + 1);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k < nb) {
            s[k] = (shv[3 * k] * gch[0] + shv[3 * k + 1] * gch[1]) + shv[3 * k + 2] * gch[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float gk = b[k] * gch[c];
              dsh[3 * k + c] += nz ? gk * (1.0f + kSqrtPoint2 * nz[3 * k + c]) : gk;
            }
          }
        }
        float ddx, ddy, ddz;
        sh_ddir(D, x, y, z, s, ddx, ddy, ddz);
        const float dot = (x * ddx + y * ddy) + z * ddz;
        dp[0] += (ddx - x * dot) / len; dp[1] += (ddy - y * dot) / len; dp[2] += (ddz - z * dot) / len;
      }
      // (2)-(6) geometry of this view with this view's scales
      const float s3[3] = {mod * sa[0], mod * sa[1], mod * sa[2]};
      float c6[6];
      cov3d_from(s3[0], s3[1], s3[2], R, c6);
      Ewa e;
      ewa_forward(vc, px, py, pz, c6, fx, fy, 1.3f * tfx, 1.3f * tfy, e);
      float dSv[9], dview[12], dproj[12];
      geom_backward(vc, e, fx, fy, W, H, px, py, pz, pa.x, pa.y, pa.z, pa.w, pb.x, pc.y, false, gndx, gndy, dSv, dp, dview,
                    dproj);
      float ds_v[3], dr_v[4];
      sigma_backward(dSv, R, s3, mod, q, ds_v, dr_v);
#pragma unroll
      for (int k = 0; k < 3; ++k) dsraw[k] += ds_v[k] * dsc[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) drot[k] += dr_v[k];
      if (out.stat_denom && ((vb.stat_mask >> vv) & 1u)) {
        out.stat_xyz_gradient_accum[i] += sqrtf(gndx * gndx + gndy * gndy);
        out.stat_denom[i] += 1.0f;
        out.stat_max_radii2D[i] = fmaxf(out.stat_max_radii2D[i], (float)vb.radii[vv][i]);
      }
    }
    if (ok) {
      float* m2 = vb.dL_dmeans2D[vv];
      m2[3 * i] = gndx; m2[3 * i + 1] = gndy; m2[3 * i + 2] = 0.f;
    }
  }

  // gradient rows -> LDS (zeros for Gaussians no view saw) -> coalesced write-back into features_dc / features_rest
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < n_valid) {
#pragma unroll
    for (int k = 0; k < F; ++k) sh[k] = dsh[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool acc = out.accumulate != 0;
  if (!(acc && amask == 0ull)) {
    if (sg.dc[m]) stage_rows_out<3>(sg.dc[m] + wave_first * 3, 3, 0, stride, n_valid, lw, acc, amask);
    if constexpr (KT > 1) {
      if (sg.rest[m]) stage_rows_out<F - 3>(sg.rest[m] + wave_first * (F - 3), F - 3, 3, stride, n_valid, lw, acc, amask);
    }
  }

  if (ok) {
    if (acc && !any) {
      float* o = sg.scaling[m];
      if (has_gs && o) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o[3 * row + k] += dsraw[k];
      }
    } else {
      // through q = raw / |raw| and sigmoid
      const float qd = ((q.x * drot[0] + q.y * drot[1]) + q.z * drot[2]) + q.w * drot[3];
      const float dr[4] = {(drot[0] - q.x * qd) / qnorm, (drot[1] - q.y * qd) / qnorm, (drot[2] - q.z * qd) / qnorm,
                           (drot[3] - q.w * qd) / qnorm};
      float gop_raw = 0.f;
      if (any) {
        const float sgm = act_sigmoid(sc.opacity[m][row]);
        gop_raw = gop * (sgm * (1.0f - sgm));
      }
      float* o;
      if ((o = sg.xyz[m])) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o[3 * row + k] = acc ? o[3 * row + k] + dp[k] : dp[k];
      }
      if ((o = sg.scaling[m])) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o[3 * row + k] = acc ? o[3 * row + k] + dsraw[k] : dsraw[k];
      }
      if ((o = sg.rotation[m])) {
        float4 t = make_float4(dr[0], dr[1], dr[2], dr[3]);
        if (acc) {
          const float4 old = *reinterpret_cast<const float4*>(o + 4 * row);
          t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
        }
        *reinterpret_cast<float4*>(o + 4 * row) = t;
      }
      if ((o = sg.opacity[m])) o[row] = acc ? o[row] + gop_raw : gop_raw;
    }
  }
}

}  // namespace

uint32_t* gsr_depth_keys(const GsrGeom& geom, int32_t P);   // binning.hip: first key buffer of the depth sort
uint32_t* gsr_depth_sort_state(const GsrGeom& geom, int32_t P, uint32_t* words);   // binning.hip: state K1 clears for the sort
uint32_t* gsr_tile_rects(const GsrGeom& geom, int32_t P);   // binning.hip: packed tile rectangles, one per Gaussian

size_t gsr_preprocess_lds_bytes(int K) { return (size_t)4 * 64 * sh_lds_stride(K) * sizeof(float); }

// GsrScene (host) -> the by-value kernel tables
static uint32_t scene_tables(const GsrScene& sc, const GsrSceneGrads* sgr, SceneTab& t, SceneGradTab& gt) {
  t = SceneTab{};
  gt = SceneGradTab{};
  t.n = sc.n_models;
  gt.dL_dscales_out = sgr ? sgr->dL_dscales_out : nullptr;
  int32_t first = 0, blk = 0;
  for (int m = 0; m < sc.n_models; ++m) {
    const GsrModel& md = sc.models[m];
    t.first[m] = first; t.fblk[m] = blk;
    first += md.count; blk += (md.count + 255) / 256;
    t.xyz[m] = md.xyz; t.scaling[m] = md.scaling; t.rotation[m] = md.rotation; t.opacity[m] = md.opacity;
    t.dc[m] = md.features_dc; t.rest[m] = md.features_rest;
    if (sgr) {
      const GsrModelGrads& mg = sgr->models[m];
      gt.xyz[m] = mg.xyz; gt.scaling[m] = mg.scaling; gt.rotation[m] = mg.rotation; gt.opacity[m] = mg.opacity;
      gt.dc[m] = mg.features_dc; gt.rest[m] = mg.features_rest;
    }
  }
  for (int m = sc.n_models; m <= GSR_MAX_MODELS; ++m) { t.first[m] = first; t.fblk[m] = blk; }
  t.scale_noise = sc.scale_noise; t.sh_noise = sc.sh_noise;
  t.scales_out = sc.scales_out; t.rotations_out = sc.rotations_out; t.opacities_out = sc.opacities_out;
  return (uint32_t)blk;
}

int gsr_launch_preprocess(const GsrView& v, const GsrGaussians& g, GsrGeom& geom, hipStream_t stream) {
  if (g.scene) {
    SceneTab t; SceneGradTab gt;
    const uint32_t nbs = scene_tables(*g.scene, nullptr, t, gt);
    const size_t lds = gsr_preprocess_lds_bytes(v.sh_stride);
    uint32_t ss_words = 0;
    uint32_t* ss = gsr_depth_sort_state(geom, v.P, &ss_words);
#define GSR_LAUNCH_K1S(KT)                                                                                         \
  hipLaunchKernelGGL((k_preprocess<KT, true, SceneTab>), dim3(nbs), dim3(256), (KT) > 0 ? 0 : lds, stream, v, g, t, \
                     geom.splat, geom.radii, geom.tiles_touched, gsr_depth_keys(geom, v.P), gsr_tile_rects(geom, v.P), \
                     ss, ss_words)
    switch (v.sh_stride) {
      case 16: GSR_LAUNCH_K1S(16); break;
      case 9: GSR_LAUNCH_K1S(9); break;
      case 4: GSR_LAUNCH_K1S(4); break;
      case 1: GSR_LAUNCH_K1S(1); break;
      default: GSR_LAUNCH_K1S(0); break;
    }
#undef GSR_LAUNCH_K1S
    GSR_HIP(hipGetLastError());
    return GSR_OK;
  }
  const uint32_t nb = gsr_num_blocks(v.P);
  const size_t lds = g.shs ? gsr_preprocess_lds_bytes(v.sh_stride) : 0;
  // compile-time strides read their rows directly (no LDS); only the generic stride stages through LDS
  uint32_t ss_words = 0;
  uint32_t* ss = gsr_depth_sort_state(geom, v.P, &ss_words);
#define GSR_LAUNCH_K1(KT)                                                                                         \
  hipLaunchKernelGGL(k_preprocess<KT>, dim3(nb), dim3(256), (KT) > 0 ? 0 : lds, stream, v, g, NoScene{}, geom.splat, \
                     geom.radii, geom.tiles_touched, gsr_depth_keys(geom, v.P), gsr_tile_rects(geom, v.P), ss, ss_words)
  switch (g.shs ? v.sh_stride : 0) {
    case 16: GSR_LAUNCH_K1(16); break;
    case 9: GSR_LAUNCH_K1(9); break;
    case 4: GSR_LAUNCH_K1(4); break;
    case 1: GSR_LAUNCH_K1(1); break;
    default: GSR_LAUNCH_K1(0); break;
  }
#undef GSR_LAUNCH_K1
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

// GSR_K8_SPARSE=0 keeps the dense kernels (one chain rule per visible Gaussian and view) for comparison runs.
static bool gsr_k8_sparse() {
  static const bool on = [] {
    const char* e = getenv("GSR_K8_SPARSE");
    return !(e && e[0] == '0');
  }();
  return on;
}

bool gsr_preprocess_bwd_views_supported(const GsrView& v, const GsrGaussians& g, const GsrGrads& out);
int gsr_launch_preprocess_bwd_views(int n_views, const GsrView* views, const GsrGaussians* gs, const GsrGeom* geoms,
                                    const GsrGrads* outs, hipStream_t stream);

namespace {
__global__ void __launch_bounds__(256) k_mask_all(unsigned long long* __restrict__ m, int64_t P) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nw = (P + 63) >> 6;
  if (w >= nw) return;
  const int64_t rest = P - (w << 6);
  m[w] = rest >= 64 ? ~0ull : ((1ull << rest) - 1ull);
}
}  // namespace
// a form of K8 that does not classify the reached Gaussians: every row may be non-zero
static void reached_mask_all(const GsrGrads& out, int32_t P, hipStream_t stream) {
  if (!out.reached_mask || P <= 0) return;
  const int64_t nw = ((int64_t)P + 63) >> 6;
  hipLaunchKernelGGL(k_mask_all, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<unsigned long long*>(out.reached_mask), (int64_t)P);
}

// Does the form of K8 the launchers below pick honour GsrGrads.scratch_clean (zero the sums and marks it consumed)? The
// views template does; the scene forms and the single-view kernel for colours / precomputed covariances / camera
// gradients do not -- gsr_backward* then clears the scratch after them.
bool gsr_k8_form_restores(bool views_entry, const GsrView& v, const GsrGaussians& g, const GsrGrads& out) {
  if (!out.reach || !out.scratch_clean || g.scene) return false;
  if (views_entry) return true;
  return gsr_k8_sparse() && v.sh_stride >= 9 && !out.dL_dcolors && !out.dL_dcov3D && gsr_preprocess_bwd_views_supported(v, g, out);
}

int gsr_launch_preprocess_bwd(const GsrView& v, const GsrGaussians& g, const GsrGeom& geom, const GsrGrads& out,
                              hipStream_t stream) {
  // the trainers' case (SH rows, scales + rotations, no camera gradients): the sparse kernel with one view
  if (!g.scene && gsr_k8_sparse() && v.sh_stride >= 9 && !out.dL_dcolors && !out.dL_dcov3D && gsr_preprocess_bwd_views_supported(v, g, out))
    return gsr_launch_preprocess_bwd_views(1, &v, &g, &geom, &out, stream);
  reached_mask_all(out, v.P, stream);
  if (g.scene) {
    SceneTab t; SceneGradTab gt;
    const uint32_t nbs = scene_tables(*g.scene, out.scene, t, gt);
    const size_t lds = gsr_preprocess_lds_bytes(v.sh_stride);
#define GSR_LAUNCH_K8S(KT)                                                                                         \
  hipLaunchKernelGGL((k_preprocess_bwd<KT, true, SceneTab, SceneGradTab>), dim3(nbs), dim3(256), lds, stream, v, g, t, \
                     gt, geom.radii, out.partials, out)
    switch (v.sh_stride) {
      case 16: GSR_LAUNCH_K8S(16); break;
      case 9: GSR_LAUNCH_K8S(9); break;
      case 4: GSR_LAUNCH_K8S(4); break;
      case 1: GSR_LAUNCH_K8S(1); break;
      default: GSR_LAUNCH_K8S(0); break;
    }
#undef GSR_LAUNCH_K8S
    GSR_HIP(hipGetLastError());
    return GSR_OK;
  }
  const uint32_t nb = gsr_num_blocks(v.P);
  const size_t lds = g.shs ? gsr_preprocess_lds_bytes(v.sh_stride) : 0;
#define GSR_LAUNCH_K8(KT)                                                                                          \
  hipLaunchKernelGGL(k_preprocess_bwd<KT>, dim3(nb), dim3(256), lds, stream, v, g, NoScene{}, NoScene{}, geom.radii, \
                     out.partials, out)
  switch (g.shs ? v.sh_stride : 0) {
    case 16: GSR_LAUNCH_K8(16); break;
    case 9: GSR_LAUNCH_K8(9); break;
    case 4: GSR_LAUNCH_K8(4); break;
    case 1: GSR_LAUNCH_K8(1); break;
    default: GSR_LAUNCH_K8(0); break;
  }
#undef GSR_LAUNCH_K8
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

// K8 for n_views views of the same Gaussians in one pass. Supported: shs with K in {1, 4, 9, 16}, (scales, rotations),
// no camera gradients, no scene table (the caller falls back to one gsr_launch_preprocess_bwd per view otherwise).
bool gsr_preprocess_bwd_views_supported(const GsrView& v, const GsrGaussians& g, const GsrGrads& out) {
  const int K = v.sh_stride;
  if (g.scene)
    return out.scene && (K == 1 || K == 4 || K == 9 || K == 16) && !out.dL_dview && !out.dL_dproj && !out.dL_dcampos;
  return g.shs && !g.scene && g.scales && g.rotations && !g.cov3D_precomp && !g.colors_precomp &&
         (K == 1 || K == 4 || K == 9 || K == 16) && !out.dL_dview && !out.dL_dproj && !out.dL_dcampos &&
         out.dL_dshs && out.dL_dscales && out.dL_drotations && out.dL_dmeans3D && out.dL_dopacities;
}

int gsr_launch_preprocess_bwd_views(int n_views, const GsrView* views, const GsrGaussians* gs, const GsrGeom* geoms,
                                    const GsrGrads* outs, hipStream_t stream) {
  const GsrGaussians& g = gs[0];
  K8Views vb = K8Views{};
  vb.nv = n_views;
  for (int k = 0; k < n_views; ++k) {
    if (g.scene) {
      vb.scale_noise[k] = gs[k].scene->scale_noise; vb.sh_noise[k] = gs[k].scene->sh_noise;
      vb.dL_dscales_out[k] = outs[k].scene ? outs[k].scene->dL_dscales_out : nullptr;
    }
    vb.scales[k] = gs[k].scales; vb.dL_dscales[k] = outs[k].dL_dscales;
    if (gs[k].scales != g.scales) vb.per_view_scales = 1;
    vb.viewmatrix[k] = views[k].viewmatrix; vb.projmatrix[k] = views[k].projmatrix; vb.campos[k] = views[k].campos;
    vb.tanfovx[k] = views[k].tanfovx; vb.tanfovy[k] = views[k].tanfovy; vb.sh_degree[k] = views[k].sh_degree;
    vb.dyn[k] = views[k].dynamic;
    vb.radii[k] = geoms[k].radii; vb.partials[k] = outs[k].partials; vb.dL_dmeans2D[k] = outs[k].dL_dmeans2D;
    vb.reach[k] = g.scene ? nullptr : reinterpret_cast<unsigned long long*>(outs[k].reach);
    if ((outs[k].reach != nullptr) != (outs[0].reach != nullptr)) return GSR_EINVAL;
  }
  vb.restore = (!g.scene && outs[0].reach && outs[0].scratch_clean) ? 1 : 0;
  // densification statistics: the views whose GsrGrads entry names the statistics tensors (all the same ones)
  GsrGrads out0 = outs[0];
  out0.stat_max_radii2D = nullptr; out0.stat_xyz_gradient_accum = nullptr; out0.stat_denom = nullptr;
  for (int k = 0; k < n_views; ++k) {
    if (!outs[k].stat_denom) continue;
    if (out0.stat_denom && (outs[k].stat_denom != out0.stat_denom || outs[k].stat_max_radii2D != out0.stat_max_radii2D ||
                            outs[k].stat_xyz_gradient_accum != out0.stat_xyz_gradient_accum))
      return GSR_EINVAL;
    out0.stat_max_radii2D = outs[k].stat_max_radii2D; out0.stat_xyz_gradient_accum = outs[k].stat_xyz_gradient_accum;
    out0.stat_denom = outs[k].stat_denom;
    vb.stat_mask |= 1u << k;
  }
  const GsrView& v = views[0];
  const size_t lds = gsr_preprocess_lds_bytes(v.sh_stride);
  if (g.scene || !(gsr_k8_sparse() && v.sh_stride >= 9)) reached_mask_all(out0, v.P, stream);
  if (g.scene) {
    SceneTab t; SceneGradTab gt;
    const uint32_t nbs = scene_tables(*g.scene, outs[0].scene, t, gt);
#define GSR_LAUNCH_K8VS(KT) \
  hipLaunchKernelGGL(k_preprocess_bwd_views_scene<KT>, dim3(nbs), dim3(256), lds, stream, v, t, gt, vb, out0)
    switch (v.sh_stride) {
      case 16: GSR_LAUNCH_K8VS(16); break;
      case 9: GSR_LAUNCH_K8VS(9); break;
      case 4: GSR_LAUNCH_K8VS(4); break;
      case 1: GSR_LAUNCH_K8VS(1); break;
      default: return GSR_EINVAL;
    }
#undef GSR_LAUNCH_K8VS
    GSR_HIP(hipGetLastError());
    return GSR_OK;
  }
  const uint32_t nb = gsr_num_blocks(v.P);
  // (rows of 12 floats or fewer, K <= 4: skipping them saves less than the classification costs -- the 2 M indoor scene
  //  at K = 4 measured 36 us per view sparse, 33 dense -- so those keep the dense kernel)
  if (gsr_k8_sparse() && v.sh_stride >= 9) {
    // 1 024 Gaussians per workgroup when that gives every CU a workgroup, 256 otherwise (the form holds two workgroups per
    // CU: 512 slots. 100 k Gaussians: 98 workgroups of 1 024 took 51 us against 26 for 391 of 256 in one shift; from
    // ~260 k on the small workgroups need two shifts and the large ones win)
    const bool big = (int64_t)v.P >= (int64_t)256 * kK8Block;
    const int64_t per_wg = big ? kK8Block : 256;
    const uint32_t nbr = (uint32_t)(((int64_t)v.P + per_wg - 1) / per_wg);
#define GSR_LAUNCH_K8SP1(KT, PVS_)                                                                                         \
  if (big) hipLaunchKernelGGL((k_preprocess_bwd_views<KT, PVS_, true, kK8Block / 256>), dim3(nbr), dim3(256), lds, stream, v, g, vb, out0); \
  else hipLaunchKernelGGL((k_preprocess_bwd_views<KT, PVS_, true, 1>), dim3(nbr), dim3(256), lds, stream, v, g, vb, out0)
#define GSR_LAUNCH_K8SP(KT)                                    \
  if (vb.per_view_scales) { GSR_LAUNCH_K8SP1(KT, true); }      \
  else { GSR_LAUNCH_K8SP1(KT, false); }
    switch (v.sh_stride) {
      case 16: GSR_LAUNCH_K8SP(16); break;
      case 9: GSR_LAUNCH_K8SP(9); break;
      default: return GSR_EINVAL;
    }
#undef GSR_LAUNCH_K8SP1
#undef GSR_LAUNCH_K8SP
    GSR_HIP(hipGetLastError());
    return GSR_OK;
  }
#define GSR_LAUNCH_K8V(KT)                                                                                       \
  if (vb.per_view_scales)                                                                                        \
    hipLaunchKernelGGL((k_preprocess_bwd_views<KT, true>), dim3(nb), dim3(256), lds, stream, v, g, vb, out0);  \
  else                                                                                                        \
    hipLaunchKernelGGL((k_preprocess_bwd_views<KT, false>), dim3(nb), dim3(256), lds, stream, v, g, vb, out0)
  switch (v.sh_stride) {
    case 16: GSR_LAUNCH_K8V(16); break;
    case 9: GSR_LAUNCH_K8V(9); break;
    case 4: GSR_LAUNCH_K8V(4); break;
    case 1: GSR_LAUNCH_K8V(1); break;
    default: return GSR_EINVAL;
  }
#undef GSR_LAUNCH_K8V
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

// K1 for n_views views of the same Gaussians in one pass (shs with K in {1,4,9,16}, scales + rotations, no scene table).
bool gsr_preprocess_views_supported(const GsrView& v, const GsrGaussians& g) {
  const int K = v.sh_stride;
  if (g.scene) return K == 1 || K == 4 || K == 9 || K == 16;
  return g.shs && !g.scene && g.scales && g.rotations && !g.cov3D_precomp && !g.colors_precomp &&
         (K == 1 || K == 4 || K == 9 || K == 16);
}

int gsr_launch_preprocess_views(int n_views, const GsrView* views, const GsrGaussians* gs, GsrGeom* geoms,
                                hipStream_t stream) {
  const GsrGaussians& g = gs[0];
  K1Views vb = K1Views{};
  vb.nv = n_views;
  for (int k = 0; k < n_views; ++k) {
    if (g.scene) {
      vb.scale_noise[k] = gs[k].scene->scale_noise; vb.sh_noise[k] = gs[k].scene->sh_noise;
      vb.scales_out[k] = gs[k].scene->scales_out;
    }
    vb.scales[k] = gs[k].scales;
    if (gs[k].scales != g.scales) vb.per_view_scales = 1;
    vb.viewmatrix[k] = views[k].viewmatrix; vb.projmatrix[k] = views[k].projmatrix; vb.campos[k] = views[k].campos;
    vb.tanfovx[k] = views[k].tanfovx; vb.tanfovy[k] = views[k].tanfovy; vb.sh_degree[k] = views[k].sh_degree;
    vb.dyn[k] = views[k].dynamic;
    vb.splat[k] = geoms[k].splat; vb.radii[k] = geoms[k].radii; vb.tiles_touched[k] = geoms[k].tiles_touched;
    vb.depth_keys[k] = gsr_depth_keys(geoms[k], views[k].P); vb.rects[k] = gsr_tile_rects(geoms[k], views[k].P);
    vb.sort_state[k] = gsr_depth_sort_state(geoms[k], views[k].P, &vb.sort_state_words);
  }
  const GsrView& v = views[0];
  if (g.scene) {
    SceneTab t; SceneGradTab gt;
    const uint32_t nbs = scene_tables(*g.scene, nullptr, t, gt);
#define GSR_LAUNCH_K1VS(KT) hipLaunchKernelGGL(k_preprocess_views_scene<KT>, dim3(nbs), dim3(256), 0, stream, v, t, vb)
    switch (v.sh_stride) {
      case 16: GSR_LAUNCH_K1VS(16); break;
      case 9: GSR_LAUNCH_K1VS(9); break;
      case 4: GSR_LAUNCH_K1VS(4); break;
      case 1: GSR_LAUNCH_K1VS(1); break;
      default: return GSR_EINVAL;
    }
#undef GSR_LAUNCH_K1VS
    GSR_HIP(hipGetLastError());
    return GSR_OK;
  }
  const uint32_t nb = gsr_num_blocks(v.P);
#define GSR_LAUNCH_K1V(KT) hipLaunchKernelGGL(k_preprocess_views<KT>, dim3(nb), dim3(256), 0, stream, v, g, vb)
  switch (v.sh_stride) {
    case 16: GSR_LAUNCH_K1V(16); break;
    case 9: GSR_LAUNCH_K1V(9); break;
    case 4: GSR_LAUNCH_K1V(4); break;
    case 1: GSR_LAUNCH_K1V(1); break;
    default: return GSR_EINVAL;
  }
#undef GSR_LAUNCH_K1V
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}
