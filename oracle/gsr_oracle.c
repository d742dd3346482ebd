/* ORACLE (test infrastructure, NOT product code) -- scalar C restatement, forward AND backward, of the
 * differentiable 3D Gaussian splatting rasterizer DreamScene imports as `diff_gaussian_rasterization`
 * (/root/reference scene_gaussian.py:11-12; call sites :586-646, :737-870, :951-1023).
 *
 * PARITY UNPINNED for the rasterizer arithmetic: its source (un-vendored, un-pinned clone
 * DreamScene-Project/comp-diff-gaussian-rasterization, README.md:47-51) is absent from /root/reference and
 * the reference has no tests / golden vectors. This file restates the published
 * algorithm of that rasterizer family (SEMANTICS.md in this repo) and is cross-checked
 * against oracle/torch_oracle.py (independent vectorised forward, autograd backward) and against the pieces
 * the reference states in Python: cov3D (gs_renderer.py:124-172), SH basis (utils/sh_utils.py:25-102),
 * projection / cameras (utils/graphics_utils.py:29-81, utils/cam_utils.py:196-210) -- tests/golden/.
 *
 * Every fp32 expression below is evaluated exactly as written (one IEEE rounding per operator, no FMA
 * contraction: build with -ffp-contract=off, no -ffast-math). The HIP kernels mirror the same operator
 * order for everything that feeds an integer artefact (depth bits, radii, tile rects, N, sort order), which
 * is what makes those artefacts bit-exact between this oracle and the GPU (SEMANTICS.md "op order").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Built twice (oracle/Makefile): libgsr_oracle.so -- scalar, one thread, deterministic: THE checker; and
 * libgsr_oracle_omp.so (-DORC_OMP -fopenmp) -- the same arithmetic per (pixel, splat) with the outer loops spread over
 * the host's cores, used ONLY by bench.py to time the CPU path on all cores (cpu_baseline.cores = threads used). In
 * the OpenMP build the per-Gaussian sums of the backward are formed per tile first (different association), so its
 * gradients equal the scalar build's to fp32 summation order, not bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef ORC_OMP
#include <omp.h>
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(dynamic, 16)")
int orc_threads(void) { return omp_get_max_threads(); }
#else
#define ORC_PARALLEL_FOR
int orc_threads(void) { return 1; }
#endif



#define BLOCK 16
#define NEAR_Z 0.2f
#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f
#define T_MIN 0.0001f
#define LOWPASS 0.3f

typedef struct {
  int32_t P, M, D, H, W;           /* M = SH coefficients stored per Gaussian (K), D = active degree      */
  float tanfovx, tanfovy, scale_modifier;
  float bg[3];
  float view[16];                  /* world_view_transform as stored by RCamera (cam_utils.py:196-197)    */
  float proj[16];                  /* full_proj_transform (cam_utils.py:205-209)                          */
  float campos[3];                 /* camera_center (cam_utils.py:210)                                    */
  int32_t prefiltered;
  int32_t score_mode;              /* 0: += opacity per contributing (pixel,splat); 1: += alpha*T         */
} OrcView;

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

/* float -> int32, truncating, saturating, NaN -> 0 (what the device conversion does; x86 would give INT_MIN) */
static int32_t f2i_sat(float x) {
  if (!(x == x)) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)x;
}
static int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }

/* exp() of the compositing loops -- THE definition (SEMANTICS.md section 4): 2^(x log2 e) with a round-to-nearest
 * split t = n + f, f in [-1/2, 1/2], a degree-5 polynomial in Horner form and an exact scaling by 2^n. IEEE
 * operations only (one rounding per *, -, fmaf; rintf = round-half-even; ldexpf exact), so render.hip evaluates the
 * very same expression tree (v_mul, v_rndne, v_sub, 5 x v_fma, v_cvt_i32, v_ldexp) and the two agree BIT FOR BIT --
 * which is what keeps the hard gates (alpha >= 1/255, T >= 1e-4) on the same side in both. Accuracy: <= 1.7e-7
 * relative from the polynomial + the rounding of t (<= 6e-8 |t|), the class of the lineage's __expf. exp(0) = 1. */
#define EXP_L2E 1.44269502162933349609375f /* float(log2 e) */
#define EXP_C1 0.6931470036506653f
#define EXP_C2 0.24022242426872253f
#define EXP_C3 0.05550733581185341f
#define EXP_C4 0.009671512991189957f
#define EXP_C5 0.001326472731307149f
static float orc_exp(float x) {
  const float t = x * EXP_L2E;
  const float n = rintf(t);
  const float f = t - n;
  float p = EXP_C5;
  p = fmaf(p, f, EXP_C4);
  p = fmaf(p, f, EXP_C3);
  p = fmaf(p, f, EXP_C2);
  p = fmaf(p, f, EXP_C1);
  p = fmaf(p, f, 1.0f);
  return ldexpf(p, f2i_sat(n));
}
/* The quadratic form of the compositing loops, with its rounding points fixed (SEMANTICS.md section 4):
 * power = dx (hA dx + nB dy) + (hC dy) dy with hA = -A/2, nB = -B, hC = -C/2 (exact), two fused multiply-adds. */
static float orc_power(float A, float B, float C, float dx, float dy) {
  const float hA = -0.5f * A, nB = -B, hC = -0.5f * C;
  return fmaf(dx, fmaf(hA, dx, nB * dy), (hC * dy) * dy);
}
static int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

static void quat_to_R(const float* q, float R[9]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
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

/* Sigma = (R diag(s)) (R diag(s))^T as [xx,xy,xz,yy,yz,zz]   (gs_renderer.py:79-88,149-172) */
static void cov3d(const float* scale, float mod, const float* q, float c6[6]) {
  float R[9], L[9];
  quat_to_R(q, R);
  const float s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
  for (int i = 0; i < 3; ++i) {
    L[3 * i + 0] = R[3 * i + 0] * s0;
    L[3 * i + 1] = R[3 * i + 1] * s1;
    L[3 * i + 2] = R[3 * i + 2] * s2;
  }
#define SIG(i, j) ((L[3 * i] * L[3 * j] + L[3 * i + 1] * L[3 * j + 1]) + L[3 * i + 2] * L[3 * j + 2])
  c6[0] = SIG(0, 0); c6[1] = SIG(0, 1); c6[2] = SIG(0, 2);
  c6[3] = SIG(1, 1); c6[4] = SIG(1, 2); c6[5] = SIG(2, 2);
#undef SIG
}

/* SH basis values for unit direction (x,y,z); utils/sh_utils.py:56-102 */
static void sh_basis(int D, float x, float y, float z, float* b) {
  b[0] = SH_C0;
  if (D > 0) {
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (D > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
      b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
      if (D > 2) {
        b[9] = SH_C3[0] * y * (3.0f * xx - yy);
        b[10] = SH_C3[1] * xy * z;
        b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
        b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
        b[14] = SH_C3[5] * z * (xx - yy);
        b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
      }
    }
  }
}

/* --------------------------------------------------------------------------------------------------- K1 */
/* Per-Gaussian projection. Outputs are zero for culled Gaussians (radii = 0, tiles_touched = 0).
 * rect = (x0,y0,x1,y1) half-open tile rectangle. clamped[3] = colour channel was clamped at 0. */
void orc_preprocess(const OrcView* v, const float* means3D, const float* scales, const float* rots,
                    const float* cov3D_precomp, const float* opacities, const float* shs,
                    const float* colors_precomp, float* depth, float* xy, float* conic_opacity, float* rgb,
                    int32_t* radii, int32_t* rect, uint32_t* tiles_touched, uint8_t* clamped, float* cov3D_out) {
  const int P = v->P, W = v->W, H = v->H;
  const float* V = v->view;
  const float* PV = v->proj;
  const int gx = (W + BLOCK - 1) / BLOCK, gy = (H + BLOCK - 1) / BLOCK;
  const float fx = (float)W / (2.0f * v->tanfovx), fy = (float)H / (2.0f * v->tanfovy);
  const float limx = 1.3f * v->tanfovx, limy = 1.3f * v->tanfovy;
  ORC_PARALLEL_FOR
  for (int i = 0; i < P; ++i) {
    radii[i] = 0; tiles_touched[i] = 0;
    depth[i] = 0; xy[2 * i] = xy[2 * i + 1] = 0;
    for (int k = 0; k < 4; ++k) { conic_opacity[4 * i + k] = 0; rect[4 * i + k] = 0; }
    for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const float tx = ((V[0] * px + V[4] * py) + V[8] * pz) + V[12];
    const float ty = ((V[1] * px + V[5] * py) + V[9] * pz) + V[13];
    const float tz = ((V[2] * px + V[6] * py) + V[10] * pz) + V[14];
    if (!(tz > NEAR_Z)) continue;
    const float hx = ((PV[0] * px + PV[4] * py) + PV[8] * pz) + PV[12];
    const float hy = ((PV[1] * px + PV[5] * py) + PV[9] * pz) + PV[13];
    const float hw = ((PV[3] * px + PV[7] * py) + PV[11] * pz) + PV[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float ndcx = hx * pw, ndcy = hy * pw;

    float c6[6];
    if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, sizeof c6);
    else cov3d(scales + 3 * i, v->scale_modifier, rots + 4 * i, c6);
    if (cov3D_out) memcpy(cov3D_out + 6 * i, c6, sizeof c6);
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};

    const float txz = tx / tz, tyz = ty / tz;
    const float txc = fminf(limx, fmaxf(-limx, txz)) * tz;
    const float tyc = fminf(limy, fmaxf(-limy, tyz)) * tz;
    const float J00 = fx / tz, J02 = -(fx * txc) / (tz * tz);
    const float J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
    float M0[3], M1[3];                       /* M = J * A, A[c][r] = V[4r+c] (world->view rotation)  */
    for (int r = 0; r < 3; ++r) {
      M0[r] = J00 * V[4 * r + 0] + J02 * V[4 * r + 2];
      M1[r] = J11 * V[4 * r + 1] + J12 * V[4 * r + 2];
    }
    float U0[3], U1[3];                       /* U = M * Sigma */
    for (int j = 0; j < 3; ++j) {
      U0[j] = (M0[0] * S[j] + M0[1] * S[3 + j]) + M0[2] * S[6 + j];
      U1[j] = (M1[0] * S[j] + M1[1] * S[3 + j]) + M1[2] * S[6 + j];
    }
    const float ca = ((U0[0] * M0[0] + U0[1] * M0[1]) + U0[2] * M0[2]) + LOWPASS;
    const float cb = (U0[0] * M1[0] + U0[1] * M1[1]) + U0[2] * M1[2];
    const float cc = ((U1[0] * M1[0] + U1[1] * M1[1]) + U1[2] * M1[2]) + LOWPASS;
    const float det = ca * cc - cb * cb;
    if (!(fabsf(det) > 0.0f) || !(fabsf(det) < INFINITY)) continue;
    const float inv = 1.0f / det;
    const float mid = 0.5f * (ca + cc);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const int32_t radius = f2i_sat(ceilf(3.0f * sqrtf(lam)));
    const float pxl = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float pyl = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float rf = (float)radius;
    const int32_t x0 = imin(gx, imax(0, f2i_sat((pxl - rf) * 0.0625f)));
    const int32_t y0 = imin(gy, imax(0, f2i_sat((pyl - rf) * 0.0625f)));
    const int32_t x1 = imin(gx, imax(0, f2i_sat(((pxl + rf) + 15.0f) * 0.0625f)));
    const int32_t y1 = imin(gy, imax(0, f2i_sat(((pyl + rf) + 15.0f) * 0.0625f)));
    if ((x1 - x0) * (y1 - y0) == 0) continue;

    if (colors_precomp) {
      for (int k = 0; k < 3; ++k) rgb[3 * i + k] = colors_precomp[3 * i + k];
    } else {
      float dx = px - v->campos[0], dy = py - v->campos[1], dz = pz - v->campos[2];
      const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float b[16];
      sh_basis(v->D, dx, dy, dz, b);
      const int nb = (v->D + 1) * (v->D + 1);
      const float* sh = shs + (size_t)i * v->M * 3;
      for (int c = 0; c < 3; ++c) {
        float acc = b[0] * sh[c];
        for (int k = 1; k < nb; ++k) acc = acc + b[k] * sh[3 * k + c];
        acc = acc + 0.5f;
        clamped[3 * i + c] = acc < 0.0f;
        rgb[3 * i + c] = fmaxf(acc, 0.0f);
      }
    }
    depth[i] = tz;
    radii[i] = radius;
    xy[2 * i] = pxl; xy[2 * i + 1] = pyl;
    conic_opacity[4 * i + 0] = cc * inv;
    conic_opacity[4 * i + 1] = -cb * inv;
    conic_opacity[4 * i + 2] = ca * inv;
    conic_opacity[4 * i + 3] = opacities[i];
    rect[4 * i + 0] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
    tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
  }
}

/* ------------------------------------------------------------------------------------------------ K2-K5 */
/* Returns N. If keys/vals are NULL only counts. ranges[tile] = (start,end), (0,0) for empty tiles. */
uint64_t orc_bin_sort(int32_t P, int32_t H, int32_t W, const int32_t* rect, const float* depth,
                      const uint32_t* tiles_touched, uint64_t* keys, uint32_t* vals, uint32_t* ranges) {
  const int gx = (W + BLOCK - 1) / BLOCK, gy = (H + BLOCK - 1) / BLOCK;
  uint64_t N = 0;
  for (int i = 0; i < P; ++i) N += tiles_touched[i];
  if (!keys) return N;
  uint64_t off = 0;
  for (int i = 0; i < P; ++i) {
    if (!tiles_touched[i]) continue;
    uint32_t dbits;
    memcpy(&dbits, depth + i, 4);
    for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
      for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) {
        keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
        vals[off] = (uint32_t)i;
        ++off;
      }
  }
#ifdef ORC_OMP
  /* Same order as the stable 64-bit sort below, spread over the cores: stable counting sort by tile id (emission order
   * = Gaussian index order inside a tile), then every tile's segment stably LSD-sorted by the 32 depth bits. */
  {
    const size_t nt = (size_t)gx * gy;
    uint64_t* k2 = (uint64_t*)malloc((N ? N : 1) * sizeof(uint64_t));
    uint32_t* v2 = (uint32_t*)malloc((N ? N : 1) * sizeof(uint32_t));
    size_t* start = (size_t*)calloc(nt + 1, sizeof(size_t));
    for (uint64_t j = 0; j < N; ++j) start[(keys[j] >> 32) + 1]++;
    for (size_t t = 0; t < nt; ++t) start[t + 1] += start[t];
    size_t* cur = (size_t*)malloc(nt * sizeof(size_t));
    memcpy(cur, start, nt * sizeof(size_t));
    for (uint64_t j = 0; j < N; ++j) {
      const size_t d = cur[keys[j] >> 32]++;
      k2[d] = keys[j]; v2[d] = vals[j];
    }
    memset(ranges, 0, nt * 2 * sizeof(uint32_t));
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t t = 0; t < nt; ++t) {
      const size_t a = start[t], n = start[t + 1] - start[t];
      if (!n) continue;
      uint64_t *ka = k2 + a, *kb = keys + a;
      uint32_t *va = v2 + a, *vb = vals + a;
      for (int pass = 0; pass < 4; ++pass) {
        size_t cnt[257];
        memset(cnt, 0, sizeof cnt);
        const int sh = 8 * pass;
        for (size_t j = 0; j < n; ++j) cnt[((ka[j] >> sh) & 255) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (size_t j = 0; j < n; ++j) {
          const size_t d = cnt[(ka[j] >> sh) & 255]++;
          kb[d] = ka[j]; vb[d] = va[j];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
      }
      /* 4 passes: back in k2/v2 -> copy to the output arrays */
      memcpy(keys + a, k2 + a, n * sizeof(uint64_t));
      memcpy(vals + a, v2 + a, n * sizeof(uint32_t));
      ranges[2 * t] = (uint32_t)a; ranges[2 * t + 1] = (uint32_t)(a + n);
    }
    free(k2); free(v2); free(start); free(cur);
    return N;
  }
#endif
  /* stable LSD radix sort on all 64 key bits */
  uint64_t* k2 = (uint64_t*)malloc((N ? N : 1) * sizeof(uint64_t));
  uint32_t* v2 = (uint32_t*)malloc((N ? N : 1) * sizeof(uint32_t));
  uint64_t *ka = keys, *kb = k2;
  uint32_t *va = vals, *vb = v2;
  for (int pass = 0; pass < 8; ++pass) {
    size_t cnt[257];
    memset(cnt, 0, sizeof cnt);
    const int sh = 8 * pass;
    for (uint64_t j = 0; j < N; ++j) cnt[((ka[j] >> sh) & 255) + 1]++;
    for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
    for (uint64_t j = 0; j < N; ++j) {
      const size_t d = cnt[(ka[j] >> sh) & 255]++;
      kb[d] = ka[j]; vb[d] = va[j];
    }
    uint64_t* tk = ka; ka = kb; kb = tk;
    uint32_t* tv = va; va = vb; vb = tv;
  }
  /* 8 passes: result is back in keys/vals */
  free(k2); free(v2);
  memset(ranges, 0, (size_t)gx * gy * 2 * sizeof(uint32_t));
  for (uint64_t j = 0; j < N; ++j) {
    const uint32_t t = (uint32_t)(keys[j] >> 32);
    if (j == 0 || (uint32_t)(keys[j - 1] >> 32) != t) ranges[2 * t] = (uint32_t)j;
    if (j == N - 1 || (uint32_t)(keys[j + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(j + 1);
  }
  return N;
}

/* --------------------------------------------------------------------------------------------------- K6 */
/* accumulator type of the per-Gaussian SUMS over pixels (importance score here, the gradient sums in K7): see bwd_pixel */
#ifdef ORC_OMP
typedef float acc_t;
#else
typedef double acc_t;
#endif
static void fwd_pixel(const OrcView* v, int px, int py, const uint32_t* ranges, const uint32_t* point_list,
                      const float* xy, const float* conic_opacity, const float* rgb, const float* depth,
                      float* out_image, float* out_depth_alpha, float* final_T, uint32_t* n_contrib,
                      acc_t* important_score) {
  const int W = v->W, H = v->H;
  const int gx = (W + BLOCK - 1) / BLOCK;
  const int tile = (py / BLOCK) * gx + (px / BLOCK);
  const uint32_t a0 = ranges[2 * tile], a1 = ranges[2 * tile + 1];
  const float pxf = (float)px, pyf = (float)py;
  float T = 1.0f, C[3] = {0, 0, 0}, Dp = 0.0f, Wt = 0.0f;
  uint32_t contributor = 0, last = 0;
  for (uint32_t j = a0; j < a1; ++j) {
    ++contributor;
    const uint32_t g = point_list[j];
    const float dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
    const float* co = conic_opacity + 4 * g;
    const float power = orc_power(co[0], co[1], co[2], dx, dy);
    if (power > 0.0f) continue;
    const float alpha = fminf(ALPHA_MAX, co[3] * orc_exp(power));
    if (alpha < ALPHA_MIN) continue;
    const float test_T = T * (1.0f - alpha);
    if (test_T < T_MIN) break;
    const float w = alpha * T;
    for (int c = 0; c < 3; ++c) C[c] += rgb[3 * g + c] * w;
    Dp += depth[g] * w;
    Wt += w;
    if (important_score) {
      const float sc = (v->score_mode == 0) ? co[3] : w;
#ifdef ORC_OMP
#pragma omp atomic
#endif
      important_score[g] += sc;
    }
    T = test_T;
    last = contributor;
  }
  const size_t pix = (size_t)py * W + px;
  final_T[pix] = T;
  n_contrib[pix] = last;
  for (int c = 0; c < 3; ++c) out_image[(size_t)c * H * W + pix] = C[c] + T * v->bg[c];
  out_depth_alpha[pix] = Dp;
  out_depth_alpha[(size_t)H * W + pix] = Wt;
}

void orc_render_fwd(const OrcView* v, const uint32_t* ranges, const uint32_t* point_list, const float* xy,
                    const float* conic_opacity, const float* rgb, const float* depth, float* out_image,
                    float* out_depth_alpha, float* final_T, uint32_t* n_contrib, float* important_score) {
  const int W = v->W, H = v->H;
#ifdef ORC_OMP
  const int gx = (W + BLOCK - 1) / BLOCK, gy = (H + BLOCK - 1) / BLOCK;
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < gx * gy; ++tile) {
    const int ty = tile / gx, tx = tile - ty * gx;
    for (int py = ty * BLOCK; py < (ty + 1) * BLOCK && py < H; ++py)
      for (int px = tx * BLOCK; px < (tx + 1) * BLOCK && px < W; ++px)
        fwd_pixel(v, px, py, ranges, point_list, xy, conic_opacity, rgb, depth, out_image, out_depth_alpha, final_T,
                  n_contrib, important_score);
  }
#else
  /* scalar build: the score terms (fp32, fixed operator order) are summed in double, like the gradient sums of K7 */
  acc_t* sc = important_score ? (acc_t*)calloc((size_t)v->P + 1, sizeof(acc_t)) : NULL;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px)
      fwd_pixel(v, px, py, ranges, point_list, xy, conic_opacity, rgb, depth, out_image, out_depth_alpha, final_T,
                n_contrib, sc);
  if (sc) {
    for (size_t i = 0; i < (size_t)v->P; ++i) important_score[i] += (float)sc[i];
    free(sc);
  }
#endif
}

/* --------------------------------------------------------------------------------------------------- K7 */
/* Accumulates (+=) into dL_dxy_ndc[P,2] (already scaled to d/d(ndc): x0.5W, x0.5H), dL_dconic[P,3]
 * (the covariance-gradient sums sum q u^2, sum q u v, sum q v^2 -- see the loop body), dL_dopacity[P], dL_drgb[P,3], dL_ddepth[P]. Caller zeroes them.
 * bwd_pixel: one pixel's reverse traversal; the sums go to row `g` of the five arrays (scalar build) or, when `local`
 * is set, to row k = position in the tile's list (OpenMP build: per-tile sums first, merged afterwards). */
/* The per-(pixel, splat) terms are fp32 with a fixed operator order; their SUMS over the pixels are accumulated in double by
 * the scalar build (the checker): the rasterizer lineage adds these terms with atomics in no particular order, so the value
 * an implementation is held to is the exact sum of the fp32 terms, not one particular fp32 summation order (with random
 * upstream gradients the terms of a large splat cancel to ~1/sqrt(#pixels) of their magnitude, and a sequential fp32 sum over
 * 10^4 pixels is itself 1e-5 .. 1e-4 off). The OpenMP build (used for timing only) keeps float sums. */
static void bwd_pixel(const OrcView* v, int px, int py, const uint32_t* ranges, const uint32_t* point_list,
                      const float* xy, const float* conic_opacity, const float* rgb, const float* depth,
                      const float* final_T, const uint32_t* n_contrib, const float* dL_dimage,
                      const float* dL_ddepth_alpha, acc_t* dL_dxy_ndc, acc_t* dL_dconic, acc_t* dL_dopacity,
                      acc_t* dL_drgb, acc_t* dL_ddepth, int local) {
  const int W = v->W, H = v->H;
  const int gx = (W + BLOCK - 1) / BLOCK;
  const float sx = 0.5f * (float)W, sy = 0.5f * (float)H;
  const int tile = (py / BLOCK) * gx + (px / BLOCK);
  const uint32_t a0 = ranges[2 * tile];
  const size_t pix = (size_t)py * W + px;
  const float pxf = (float)px, pyf = (float)py;
  /* The GATES are the forward's: power, G = exp, alpha in fp32 with the fixed operator order (orc_power / orc_exp), so the
   * backward walks exactly the contributors the forward composited. Everything downstream of the gates -- the transmittance
   * multiplied back up, the colour / depth / alpha composited behind, dL/dalpha, the per-pixel terms of the sums -- is
   * evaluated in rt_t: double in the scalar build (the checker holds an implementation to the exact gradient of the
   * function the fp32 forward computed: the recurrences run over 100+ layers and (value - behind) cancels, so an fp32
   * evaluation of THIS restatement is itself 1e-5 .. 2e-4 off on pixels with a large upstream gradient -- e.g. the pixel
   * the reference's disp normalisation pins its (max - min) on --, measured against float64: tests/test_boundary_fixture.py),
   * float in the OpenMP timing build. */
#ifdef ORC_OMP
  typedef float rt_t;
#else
  typedef double rt_t;
#endif
  const float Tf = final_T[pix];
  rt_t T = Tf;
  const float gC[3] = {dL_dimage[pix], dL_dimage[(size_t)H * W + pix], dL_dimage[(size_t)2 * H * W + pix]};
  const float gD = dL_ddepth_alpha[pix], gA = dL_ddepth_alpha[(size_t)H * W + pix];
  const rt_t bg_dot = ((rt_t)v->bg[0] * gC[0] + (rt_t)v->bg[1] * gC[1]) + (rt_t)v->bg[2] * gC[2];
  rt_t last_alpha = 0, last_c[3] = {0, 0, 0}, rec_c[3] = {0, 0, 0}, last_z = 0, rec_z = 0, rec_a = 0;
  for (uint32_t k = n_contrib[pix]; k-- > 0;) {
    const uint32_t g = point_list[a0 + k];
    const size_t r = local ? (size_t)k : (size_t)g;
    const float dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
    const float* co = conic_opacity + 4 * g;
    const float power = orc_power(co[0], co[1], co[2], dx, dy);
    if (power > 0.0f) continue;
    const float G = orc_exp(power);
    const float alpha = fminf(ALPHA_MAX, co[3] * G);
    if (alpha < ALPHA_MIN) continue;
    T = T / ((rt_t)1 - alpha);
    const rt_t w = alpha * T;
    rt_t dL_dalpha = 0;
    for (int c = 0; c < 3; ++c) {
      rec_c[c] = last_alpha * last_c[c] + ((rt_t)1 - last_alpha) * rec_c[c];
      last_c[c] = rgb[3 * g + c];
      dL_dalpha += (rgb[3 * g + c] - rec_c[c]) * gC[c];
      dL_drgb[3 * r + c] += w * gC[c];
    }
    rec_z = last_alpha * last_z + ((rt_t)1 - last_alpha) * rec_z;
    last_z = depth[g];
    dL_dalpha += (depth[g] - rec_z) * gD;
    dL_ddepth[r] += w * gD;
    rec_a = last_alpha + ((rt_t)1 - last_alpha) * rec_a;
    dL_dalpha += ((rt_t)1 - rec_a) * gA;
    dL_dalpha *= T;
    last_alpha = alpha;
    dL_dalpha += (-(rt_t)Tf / ((rt_t)1 - alpha)) * bg_dot;
    /* With (u, v) = -Sigma^-1 d = -(A dx + B dy, C dy + B dx) and q = dL/dG G:  dG/dd = G (u, v), so q (u, v) is this
     * pixel's share of dL/d(pixel centre); and dL/dSigma = 1/2 sum q (Sigma^-1 d)(Sigma^-1 d)^T, so q (u^2, u v, v^2) is its
     * share of the gradient of the 2-D covariance ITSELF. Both are formed here, per pixel; the `dL_dconic` rows hold
     * (sum q u^2, sum q u v, sum q v^2) and orc_preprocess_bwd only scales them (SEMANTICS.md section 5). The lineage
     * sums dL/dconic = -1/2 q (dx^2, 2 dx dy, dy^2) and converts afterwards, which is the same in exact arithmetic and
     * loses cond(Sigma)^2 digits in fp32 (needle-shaped splats: 1e-1 against float64 autograd). */
    const rt_t q = ((rt_t)co[3] * dL_dalpha) * G;
    const rt_t u = -((rt_t)co[0] * dx + (rt_t)co[1] * dy), w2 = -((rt_t)co[2] * dy + (rt_t)co[1] * dx);
    const rt_t m1 = q * u, m2 = q * w2;
    dL_dxy_ndc[2 * r] += m1 * sx;
    dL_dxy_ndc[2 * r + 1] += m2 * sy;
    dL_dconic[3 * r] += m1 * u;
    dL_dconic[3 * r + 1] += m1 * w2;
    dL_dconic[3 * r + 2] += m2 * w2;
    dL_dopacity[r] += G * dL_dalpha;
  }
}

void orc_render_bwd(const OrcView* v, const uint32_t* ranges, const uint32_t* point_list, const float* xy,
                    const float* conic_opacity, const float* rgb, const float* depth, const float* final_T,
                    const uint32_t* n_contrib, const float* dL_dimage, const float* dL_ddepth_alpha,
                    float* dL_dxy_ndc, float* dL_dconic, float* dL_dopacity, float* dL_drgb, float* dL_ddepth) {
  const int W = v->W, H = v->H;
#ifdef ORC_OMP
  const int gx = (W + BLOCK - 1) / BLOCK, gy = (H + BLOCK - 1) / BLOCK;
#pragma omp parallel
  {
    float* loc = NULL;
    size_t loc_cap = 0;
#pragma omp for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
      const uint32_t a0 = ranges[2 * tile], a1 = ranges[2 * tile + 1];
      const size_t n = a1 - a0;
      if (!n) continue;
      if (n > loc_cap) { free(loc); loc_cap = n + n / 2; loc = (float*)malloc(loc_cap * 10 * sizeof(float)); }
      memset(loc, 0, n * 10 * sizeof(float));
      float *lxy = loc, *lcon = loc + 2 * n, *lop = loc + 5 * n, *lrgb = loc + 6 * n, *ldep = loc + 9 * n;
      const int ty = tile / gx, tx = tile - ty * gx;
      for (int py = ty * BLOCK; py < (ty + 1) * BLOCK && py < H; ++py)
        for (int px = tx * BLOCK; px < (tx + 1) * BLOCK && px < W; ++px)
          bwd_pixel(v, px, py, ranges, point_list, xy, conic_opacity, rgb, depth, final_T, n_contrib, dL_dimage,
                    dL_ddepth_alpha, lxy, lcon, lop, lrgb, ldep, 1);
      for (size_t k = 0; k < n; ++k) {     /* the tile's sums -> the Gaussians' rows */
        const size_t g = point_list[a0 + k];
        const float add[10] = {lxy[2 * k], lxy[2 * k + 1], lcon[3 * k], lcon[3 * k + 1], lcon[3 * k + 2], lop[k],
                               lrgb[3 * k], lrgb[3 * k + 1], lrgb[3 * k + 2], ldep[k]};
        float* dst[10] = {dL_dxy_ndc + 2 * g, dL_dxy_ndc + 2 * g + 1, dL_dconic + 3 * g, dL_dconic + 3 * g + 1,
                          dL_dconic + 3 * g + 2, dL_dopacity + g, dL_drgb + 3 * g, dL_drgb + 3 * g + 1,
                          dL_drgb + 3 * g + 2, dL_ddepth + g};
        for (int c = 0; c < 10; ++c) {
          if (add[c] == 0.0f) continue;
#pragma omp atomic
          *dst[c] += add[c];
        }
      }
    }
    free(loc);
  }
#else
  const size_t P = (size_t)v->P;
  acc_t* acc = (acc_t*)calloc(P * 10 + 1, sizeof(acc_t));
  acc_t *axy = acc, *acon = acc + 2 * P, *aop = acc + 5 * P, *argb = acc + 6 * P, *adep = acc + 9 * P;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px)
      bwd_pixel(v, px, py, ranges, point_list, xy, conic_opacity, rgb, depth, final_T, n_contrib, dL_dimage,
                dL_ddepth_alpha, axy, acon, aop, argb, adep, 0);
  for (size_t i = 0; i < 2 * P; ++i) dL_dxy_ndc[i] += (float)axy[i];
  for (size_t i = 0; i < 3 * P; ++i) dL_dconic[i] += (float)acon[i];
  for (size_t i = 0; i < P; ++i) dL_dopacity[i] += (float)aop[i];
  for (size_t i = 0; i < 3 * P; ++i) dL_drgb[i] += (float)argb[i];
  for (size_t i = 0; i < P; ++i) dL_ddepth[i] += (float)adep[i];
  free(acc);
#endif
}

/* --------------------------------------------------------------------------------------------------- K8 */
/* Chain rule to the inputs. Outputs are overwritten (zero for culled Gaussians, radii == 0).
 * dL_dmeans2D[P,3] = (dL_dxy_ndc, 0). dL_dview/dL_dproj[16], dL_dcampos[3] accumulate (+=) if non-NULL. */
void orc_preprocess_bwd(const OrcView* v, const float* means3D, const float* scales, const float* rots,
                        const float* cov3D_precomp, const float* shs, const int32_t* radii,
                        const uint8_t* clamped, const float* dL_dxy_ndc, const float* dL_dconic,
                        const float* dL_drgb, const float* dL_ddepth, float* dL_dmeans3D, float* dL_dmeans2D,
                        float* dL_dscales, float* dL_drots, float* dL_dcov3D, float* dL_dshs, float* dL_dview,
                        float* dL_dproj, float* dL_dcampos) {
  const int P = v->P, W = v->W, H = v->H, M = v->M, D = v->D;
  const float* V = v->view;
  const float* PV = v->proj;
  const float fx = (float)W / (2.0f * v->tanfovx), fy = (float)H / (2.0f * v->tanfovy);
  const float limx = 1.3f * v->tanfovx, limy = 1.3f * v->tanfovy;
  const float mod = v->scale_modifier;
#ifdef ORC_OMP
#pragma omp parallel for schedule(dynamic, 64) if (!dL_dview && !dL_dproj && !dL_dcampos)
#endif
  for (int i = 0; i < P; ++i) {
    for (int k = 0; k < 3; ++k) { dL_dmeans3D[3 * i + k] = 0; dL_dmeans2D[3 * i + k] = 0; }
    if (dL_dscales) for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = 0;
    if (dL_drots) for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0;
    if (dL_dcov3D) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0;
    if (dL_dshs) for (int k = 0; k < 3 * M; ++k) dL_dshs[(size_t)i * M * 3 + k] = 0;
    if (!(radii[i] > 0)) continue;
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    float dp[3] = {0, 0, 0};

    /* (1) colour -> SH coefficients and view direction */
    if (shs) {
      float vx = px - v->campos[0], vy = py - v->campos[1], vz = pz - v->campos[2];
      const float len = sqrtf((vx * vx + vy * vy) + vz * vz);
      const float x = vx / len, y = vy / len, z = vz / len;
      float b[16];
      sh_basis(D, x, y, z, b);
      const int nb = (D + 1) * (D + 1);
      const float* sh = shs + (size_t)i * M * 3;
      float g[3];
      for (int c = 0; c < 3; ++c) g[c] = clamped[3 * i + c] ? 0.0f : dL_drgb[3 * i + c];
      for (int k = 0; k < nb; ++k)
        for (int c = 0; c < 3; ++c) dL_dshs[(size_t)i * M * 3 + 3 * k + c] = b[k] * g[c];
      /* s_k = sum_c sh[k][c] * g[c];  dL/ddir = sum_k dbasis_k/ddir * s_k */
      float s[16];
      for (int k = 0; k < nb; ++k) s[k] = (sh[3 * k] * g[0] + sh[3 * k + 1] * g[1]) + sh[3 * k + 2] * g[2];
      float ddx = 0, ddy = 0, ddz = 0;
      if (D > 0) {
        ddy += -SH_C1 * s[1]; ddz += SH_C1 * s[2]; ddx += -SH_C1 * s[3];
        if (D > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          ddx += SH_C2[0] * y * s[4] + SH_C2[2] * (-2.0f * x) * s[6] + SH_C2[3] * z * s[7] + SH_C2[4] * 2.0f * x * s[8];
          ddy += SH_C2[0] * x * s[4] + SH_C2[1] * z * s[5] + SH_C2[2] * (-2.0f * y) * s[6] + SH_C2[4] * (-2.0f * y) * s[8];
          ddz += SH_C2[1] * y * s[5] + SH_C2[2] * 4.0f * z * s[6] + SH_C2[3] * x * s[7];
          if (D > 2) {
            ddx += SH_C3[0] * 6.0f * xy * s[9] + SH_C3[1] * yz * s[10] + SH_C3[2] * (-2.0f * xy) * s[11] +
                   SH_C3[3] * (-6.0f * xz) * s[12] + SH_C3[4] * (4.0f * zz - 3.0f * xx - yy) * s[13] +
                   SH_C3[5] * 2.0f * xz * s[14] + SH_C3[6] * 3.0f * (xx - yy) * s[15];
            ddy += SH_C3[0] * 3.0f * (xx - yy) * s[9] + SH_C3[1] * xz * s[10] +
                   SH_C3[2] * (4.0f * zz - xx - 3.0f * yy) * s[11] + SH_C3[3] * (-6.0f * yz) * s[12] +
                   SH_C3[4] * (-2.0f * xy) * s[13] + SH_C3[5] * (-2.0f * yz) * s[14] + SH_C3[6] * (-6.0f * xy) * s[15];
            ddz += SH_C3[1] * xy * s[10] + SH_C3[2] * 8.0f * yz * s[11] + SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy) * s[12] +
                   SH_C3[4] * 8.0f * xz * s[13] + SH_C3[5] * (xx - yy) * s[14];
          }
        }
      }
      /* through dir = v/|v| */
      const float dot = (x * ddx + y * ddy) + z * ddz;
      const float dvx = (ddx - x * dot) / len, dvy = (ddy - y * dot) / len, dvz = (ddz - z * dot) / len;
      dp[0] += dvx; dp[1] += dvy; dp[2] += dvz;
      if (dL_dcampos) { dL_dcampos[0] -= dvx; dL_dcampos[1] -= dvy; dL_dcampos[2] -= dvz; }
    }

    /* recompute the forward chain */
    const float tx = ((V[0] * px + V[4] * py) + V[8] * pz) + V[12];
    const float ty = ((V[1] * px + V[5] * py) + V[9] * pz) + V[13];
    const float tz = ((V[2] * px + V[6] * py) + V[10] * pz) + V[14];
    float c6[6];
    if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, sizeof c6);
    else cov3d(scales + 3 * i, mod, rots + 4 * i, c6);
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    const float txz = tx / tz, tyz = ty / tz;
    const int clx = (txz < -limx) || (txz > limx), cly = (tyz < -limy) || (tyz > limy);
    const float txc = fminf(limx, fmaxf(-limx, txz)) * tz;
    const float tyc = fminf(limy, fmaxf(-limy, tyz)) * tz;
    const float J00 = fx / tz, J02 = -(fx * txc) / (tz * tz);
    const float J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
    float M0[3], M1[3], U0[3], U1[3];
    for (int r = 0; r < 3; ++r) {
      M0[r] = J00 * V[4 * r + 0] + J02 * V[4 * r + 2];
      M1[r] = J11 * V[4 * r + 1] + J12 * V[4 * r + 2];
    }
    for (int j = 0; j < 3; ++j) {
      U0[j] = (M0[0] * S[j] + M0[1] * S[3 + j]) + M0[2] * S[6 + j];
      U1[j] = (M1[0] * S[j] + M1[1] * S[3 + j]) + M1[2] * S[6 + j];
    }
    const float ca = ((U0[0] * M0[0] + U0[1] * M0[1]) + U0[2] * M0[2]) + LOWPASS;
    const float cb = (U0[0] * M1[0] + U0[1] * M1[1]) + U0[2] * M1[2];
    const float cc = ((U1[0] * M1[0] + U1[1] * M1[1]) + U1[2] * M1[2]) + LOWPASS;
    const float det = ca * cc - cb * cb;

    /* (2) the per-pixel sums (bwd_pixel) -> dL/dcov2D = 1/2 sum q (Sigma^-1 d)(Sigma^-1 d)^T; the lineage reaches the same
     * quantity through dL/dconic with the denominator det^2 + 1e-7 instead of det^2: the factor below keeps that */
    const float det2 = det * det;
    const float reg = det2 * (1.0f / (det2 + 0.0000001f));
    const float dca = (0.5f * dL_dconic[3 * i]) * reg, dcb = dL_dconic[3 * i + 1] * reg, dcc = (0.5f * dL_dconic[3 * i + 2]) * reg;

    /* (3) cov2D = M Sigma M^T: dSigma = M^T G2 M (G2 symmetric with off-diagonal dcb/2), dM = 2 G2 M Sigma */
    const float h = 0.5f * dcb;
    float dS[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        dS[3 * r + c] = (M0[r] * (dca * M0[c] + h * M1[c])) + (M1[r] * (h * M0[c] + dcc * M1[c]));
    float dM0[3], dM1[3];
    for (int j = 0; j < 3; ++j) {
      dM0[j] = 2.0f * (dca * U0[j] + h * U1[j]);
      dM1[j] = 2.0f * (h * U0[j] + dcc * U1[j]);
    }
    /* M = J A: dJ = dM A^T ; A[c][r] = V[4r+c] */
    const float dJ00 = (dM0[0] * V[0] + dM0[1] * V[4]) + dM0[2] * V[8];
    const float dJ02 = (dM0[0] * V[2] + dM0[1] * V[6]) + dM0[2] * V[10];
    const float dJ11 = (dM1[0] * V[1] + dM1[1] * V[5]) + dM1[2] * V[9];
    const float dJ12 = (dM1[0] * V[2] + dM1[1] * V[6]) + dM1[2] * V[10];
    const float tzi = 1.0f / tz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
    float dt[3];
    dt[0] = clx ? 0.0f : (-fx * tz2 * dJ02);
    dt[1] = cly ? 0.0f : (-fy * tz2 * dJ12);
    dt[2] = ((-fx * tz2 * dJ00 - fy * tz2 * dJ11) + (2.0f * fx * txc) * tz3 * dJ02) + (2.0f * fy * tyc) * tz3 * dJ12;
    /* (5) depth */
    dt[2] += dL_ddepth[i];
    for (int r = 0; r < 3; ++r) dp[r] += (V[4 * r] * dt[0] + V[4 * r + 1] * dt[1]) + V[4 * r + 2] * dt[2];

    /* (4) ndc xy -> p through proj with 1/(w+1e-7) */
    const float hx = ((PV[0] * px + PV[4] * py) + PV[8] * pz) + PV[12];
    const float hy = ((PV[1] * px + PV[5] * py) + PV[9] * pz) + PV[13];
    const float hw = ((PV[3] * px + PV[7] * py) + PV[11] * pz) + PV[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float gx_ = dL_dxy_ndc[2 * i], gy_ = dL_dxy_ndc[2 * i + 1];
    const float dh[3] = {pw * gx_, pw * gy_, -(pw * pw) * (hx * gx_ + hy * gy_)};
    for (int r = 0; r < 3; ++r) dp[r] += (PV[4 * r] * dh[0] + PV[4 * r + 1] * dh[1]) + PV[4 * r + 3] * dh[2];
    dL_dmeans2D[3 * i] = gx_; dL_dmeans2D[3 * i + 1] = gy_;
    for (int r = 0; r < 3; ++r) dL_dmeans3D[3 * i + r] = dp[r];

    if (dL_dview) {
      const float p4[4] = {px, py, pz, 1.0f};
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c) dL_dview[4 * r + c] += p4[r] * dt[c];
      /* A path: dA[c][r] = sum_i J[i][c] dM[i][r]; J rows: (J00,0,J02), (0,J11,J12) */
      for (int r = 0; r < 3; ++r) {
        dL_dview[4 * r + 0] += J00 * dM0[r];
        dL_dview[4 * r + 1] += J11 * dM1[r];
        dL_dview[4 * r + 2] += J02 * dM0[r] + J12 * dM1[r];
      }
    }
    if (dL_dproj) {
      const float p4[4] = {px, py, pz, 1.0f};
      for (int r = 0; r < 4; ++r) {
        dL_dproj[4 * r + 0] += p4[r] * dh[0];
        dL_dproj[4 * r + 1] += p4[r] * dh[1];
        dL_dproj[4 * r + 3] += p4[r] * dh[2];
      }
    }

    /* (6) Sigma -> 6-vector, or -> scale / quaternion */
    if (cov3D_precomp) {
      if (dL_dcov3D) {
        dL_dcov3D[6 * i + 0] = dS[0];
        dL_dcov3D[6 * i + 1] = dS[1] + dS[3];
        dL_dcov3D[6 * i + 2] = dS[2] + dS[6];
        dL_dcov3D[6 * i + 3] = dS[4];
        dL_dcov3D[6 * i + 4] = dS[5] + dS[7];
        dL_dcov3D[6 * i + 5] = dS[8];
      }
    } else {
      float R[9], L[9], dL[9], dR[9];
      const float* q = rots + 4 * i;
      quat_to_R(q, R);
      const float s[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
      for (int a = 0; a < 3; ++a)
        for (int b2 = 0; b2 < 3; ++b2) L[3 * a + b2] = R[3 * a + b2] * s[b2];
      /* dL = (dS + dS^T) L */
      for (int a = 0; a < 3; ++a)
        for (int b2 = 0; b2 < 3; ++b2) {
          float acc = 0;
          for (int k = 0; k < 3; ++k) acc += (dS[3 * a + k] + dS[3 * k + a]) * L[3 * k + b2];
          dL[3 * a + b2] = acc;
        }
      for (int b2 = 0; b2 < 3; ++b2) {
        const float ds = (dL[b2] * R[b2] + dL[3 + b2] * R[3 + b2]) + dL[6 + b2] * R[6 + b2];
        dL_dscales[3 * i + b2] = mod * ds;
      }
      for (int a = 0; a < 3; ++a)
        for (int b2 = 0; b2 < 3; ++b2) dR[3 * a + b2] = dL[3 * a + b2] * s[b2];
      const float r = q[0], x = q[1], y = q[2], z = q[3];
      dL_drots[4 * i + 0] = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      dL_drots[4 * i + 1] = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
      dL_drots[4 * i + 2] = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
      dL_drots[4 * i + 3] = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
  }
}
