// render.hip -- K6 front-to-back alpha compositing and K7 its reverse-order backward, gfx950 (wave64).
//
// One 256-thread workgroup per 16x16 pixel tile; each of the 4 waves owns an 8x8 pixel block so that the
// wave-level early-outs (whole wave finished / no lane of the wave touched by this splat) follow the screen
// footprint of a splat as tightly as a 64-wide wave allows. Splat records (3 x float4, written by K1) are
// gathered by list index into LDS 256 at a time and then read back as wave-uniform broadcasts.
// Semantics: SURVEY.md Appendix A.2 / A.3, SEMANTICS.md; outputs as consumed at scene_gaussian.py:1012-1032.
#include "gsr_common.h"

#ifndef GSR_FAST_EXP
#define GSR_FAST_EXP 1
#endif

namespace {

constexpr int kBatch = 256;

__device__ __forceinline__ float gsr_exp(float x) {
#if GSR_FAST_EXP
  return __expf(x);
#else
  return expf(x);
#endif
}

struct TilePix {
  int px, py;
  bool inside;
};

__device__ __forceinline__ TilePix tile_pixel(int tile, int gx, int W, int H) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ty = tile / gx, tx = tile - ty * gx;
  TilePix p;
  p.px = tx * GSR_TILE + (wave & 1) * 8 + (lane & 7);
  p.py = ty * GSR_TILE + (wave >> 1) * 8 + (lane >> 3);
  p.inside = (p.px < W) && (p.py < H);
  return p;
}

// --------------------------------------------------------------------------------------------------------- K6
template <bool SCORE>
__global__ void __launch_bounds__(256)
k_render_fwd(const int W, const int H, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
             const float4* __restrict__ splat, const float* __restrict__ bg, float* __restrict__ out_color,
             float* __restrict__ out_da, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             float* __restrict__ score, const int score_mode) {
  __shared__ float4 s0[kBatch], s1[kBatch], s2[kBatch];
  __shared__ uint32_t sid[kBatch];
  const int gx = (W + GSR_TILE - 1) / GSR_TILE;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const TilePix p = tile_pixel(tile, gx, W, H);
  const float pxf = (float)p.px, pyf = (float)p.py;
  const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];

  bool done = !p.inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
  uint32_t last = 0;

  for (uint32_t base = r0; base < r1; base += kBatch) {
    if (__syncthreads_count(done) == 256) break;
    const uint32_t idx = base + tid;
    if (idx < r1) {
      const uint32_t id = point_list[idx];
      const float4* r = splat + 3 * (size_t)id;
      s0[tid] = r[0]; s1[tid] = r[1]; s2[tid] = r[2];
      if (SCORE) sid[tid] = id;
    }
    __syncthreads();
    const int n = (int)min((uint32_t)kBatch, r1 - base);
    for (int j = 0; j < n; ++j) {
      if (__ballot(!done) == 0ull) break;
      const float4 a = s0[j];
      const float4 b = s1[j];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      const float alpha = fminf(GSR_ALPHA_MAX, b.y * gsr_exp(power));
      bool hit = !done && (power <= 0.0f) && (alpha >= GSR_ALPHA_MIN);
      const float test_T = T * (1.0f - alpha);
      if (hit && test_T < GSR_T_MIN) { done = true; hit = false; }
      if (SCORE) {
        // per-wave reduction before the global atomic: 1 atomic per (wave, splat) instead of up to 64
        const unsigned long long hm = __ballot(hit);
        if (hm) {
          float sc;
          if (score_mode == 0) {
            sc = b.y * (float)__popcll(hm);
          } else {
            sc = gsr_wave_sum_to_lane63(hit ? alpha * T : 0.f);
            sc = __shfl(sc, 63, 64);
          }
          if (lane == 0) unsafeAtomicAdd(score + sid[j], sc);
        }
      }
      if (hit) {
        const float4 c = s2[j];
        const float w = alpha * T;
        C0 += b.w * w; C1 += c.x * w; C2 += c.y * w;
        Dp += b.z * w;
        Wt += w;
        T = test_T;
        last = (base - r0) + (uint32_t)j + 1u;
      }
    }
  }
  if (p.inside) {
    const size_t pix = (size_t)p.py * W + p.px, HW = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_color[pix] = C0 + T * bg[0];
    out_color[HW + pix] = C1 + T * bg[1];
    out_color[2 * HW + pix] = C2 + T * bg[2];
    out_da[pix] = Dp;
    out_da[HW + pix] = Wt;
  }
}

// --------------------------------------------------------------------------------------------------------- K7
// Accumulates into partials [P,12]:
//   (dL/dndc_x, dL/dndc_y, dL/dconic_a, dL/dconic_b, dL/dconic_c, dL/dopacity, dL/dr, dL/dg, dL/db, dL/ddepth, -, -)
__global__ void __launch_bounds__(256)
k_render_bwd(const int W, const int H, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
             const float4* __restrict__ splat, const float* __restrict__ bg, const float* __restrict__ final_T,
             const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
             const float* __restrict__ dL_dda, float* __restrict__ partials) {
  __shared__ float4 s0[kBatch], s1[kBatch], s2[kBatch];
  __shared__ uint32_t sid[kBatch];
  __shared__ uint32_t wmax[4];
  const int gx = (W + GSR_TILE - 1) / GSR_TILE;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const TilePix p = tile_pixel(tile, gx, W, H);
  const float pxf = (float)p.px, pyf = (float)p.py;
  const uint32_t r0 = ranges[2 * tile];
  const size_t pix = (size_t)p.py * W + p.px, HW = (size_t)H * W;

  const float Tf = p.inside ? final_T[pix] : 0.f;
  const uint32_t last = p.inside ? n_contrib[pix] : 0u;
  float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
  if (p.inside) {
    gC0 = dL_dcolor[pix]; gC1 = dL_dcolor[HW + pix]; gC2 = dL_dcolor[2 * HW + pix];
    gD = dL_dda[pix]; gA = dL_dda[HW + pix];
  }
  const float bg_dot = (bg[0] * gC0 + bg[1] * gC1) + bg[2] * gC2;
  const float sx = 0.5f * (float)W, sy = 0.5f * (float)H;

  const uint32_t wm = gsr_wave_max_u32(last);
  if (lane == 0) wmax[wave] = wm;
  __syncthreads();
  const uint32_t tile_max = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));

  float T = Tf;
  float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, rc0 = 0.f, rc1 = 0.f, rc2 = 0.f;
  float last_z = 0.f, rec_z = 0.f, rec_a = 0.f;

  for (uint32_t hi = tile_max; hi > 0; hi = (hi > kBatch) ? hi - kBatch : 0u) {
    const int n = (int)min((uint32_t)kBatch, hi);
    __syncthreads();
    if (tid < n) {
      const uint32_t id = point_list[r0 + (hi - 1u - (uint32_t)tid)];
      const float4* r = splat + 3 * (size_t)id;
      s0[tid] = r[0]; s1[tid] = r[1]; s2[tid] = r[2];
      sid[tid] = id;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
      const uint32_t pos = hi - 1u - (uint32_t)j;      // 0-based list position
      const bool live = pos < last;
      if (__ballot(live) == 0ull) continue;
      const float4 a = s0[j];
      const float4 b = s1[j];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      const float G = gsr_exp(power);
      const float alpha = fminf(GSR_ALPHA_MAX, b.y * G);
      const bool hit = live && (power <= 0.0f) && (alpha >= GSR_ALPHA_MIN);
      if (__ballot(hit) == 0ull) continue;
      float v[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) v[k] = 0.f;
      if (hit) {
        const float4 c = s2[j];
        T = T / (1.0f - alpha);
        const float w = alpha * T;
        float dL_dalpha;
        rc0 = last_alpha * lc0 + (1.0f - last_alpha) * rc0; lc0 = b.w;
        rc1 = last_alpha * lc1 + (1.0f - last_alpha) * rc1; lc1 = c.x;
        rc2 = last_alpha * lc2 + (1.0f - last_alpha) * rc2; lc2 = c.y;
        dL_dalpha = (b.w - rc0) * gC0 + (c.x - rc1) * gC1 + (c.y - rc2) * gC2;
        rec_z = last_alpha * last_z + (1.0f - last_alpha) * rec_z; last_z = b.z;
        dL_dalpha += (b.z - rec_z) * gD;
        rec_a = last_alpha + (1.0f - last_alpha) * rec_a;
        dL_dalpha += (1.0f - rec_a) * gA;
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-Tf / (1.0f - alpha)) * bg_dot;
        const float dL_dG = b.y * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        v[0] = dL_dG * (-gdx * a.z - gdy * a.w);
        v[1] = dL_dG * (-gdy * b.x - gdx * a.w);
        v[2] = -0.5f * gdx * dx * dL_dG;
        v[3] = -gdx * dy * dL_dG;
        v[4] = -0.5f * gdy * dy * dL_dG;
        v[5] = G * dL_dalpha;
        v[6] = w * gC0; v[7] = w * gC1; v[8] = w * gC2;
        v[9] = w * gD;
      }
#pragma unroll
      for (int k = 0; k < 10; ++k) v[k] = gsr_wave_sum_to_lane63(v[k]);
      if (lane == 63) {
        float* dst = partials + 12 * (size_t)sid[j];
        unsafeAtomicAdd(dst + 0, v[0] * sx);
        unsafeAtomicAdd(dst + 1, v[1] * sy);
#pragma unroll
        for (int k = 2; k < 10; ++k) unsafeAtomicAdd(dst + k, v[k]);
      }
    }
  }
}

}  // namespace

int gsr_launch_render_fwd(const GsrView& v, const GsrGeom& geom, const GsrBinning& b, GsrImages& img,
                          hipStream_t stream) {
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  const float4* splat = reinterpret_cast<const float4*>(geom.splat);
  if (img.important_score) {
    hipLaunchKernelGGL(k_render_fwd<true>, dim3(tiles), dim3(256), 0, stream, v.image_width, v.image_height,
                       b.ranges, b.point_list, splat, v.bg, img.color, img.depth_alpha, img.final_T, img.n_contrib,
                       img.important_score, v.score_mode);
  } else {
    hipLaunchKernelGGL(k_render_fwd<false>, dim3(tiles), dim3(256), 0, stream, v.image_width, v.image_height,
                       b.ranges, b.point_list, splat, v.bg, img.color, img.depth_alpha, img.final_T, img.n_contrib,
                       (float*)nullptr, 0);
  }
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

int gsr_launch_render_bwd(const GsrView& v, const GsrGeom& geom, const GsrBinning& b, const GsrImages& img,
                          const GsrImageGrads& ig, GsrGrads& out, hipStream_t stream) {
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  hipLaunchKernelGGL(k_render_bwd, dim3(tiles), dim3(256), 0, stream, v.image_width, v.image_height, b.ranges,
                     b.point_list, reinterpret_cast<const float4*>(geom.splat), v.bg, img.final_T, img.n_contrib,
                     ig.dL_dcolor, ig.dL_ddepth_alpha, out.partials);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}
