// render.hip -- K6 front-to-back alpha compositing and K7 its reverse-order backward, gfx950 (wave64).
//
// Work items, not tiles: K6 takes one 8x8 pixel quarter of a 16x16 tile per 256-thread workgroup (four lanes share a
// pixel and take four consecutive candidates of the list per step; a whole-tile variant with one pixel per lane serves
// scenes of thousands of shallow tiles), K7 one (tile, 256-entry segment) with one 8x8 block per wave, restarted from
// the per-pixel checkpoints K6 leaves at every 256-entry boundary.
//  * Splat records (3 x float4, written by K1) are gathered by list index 256 at a time into an LDS stage (double-
//    buffered in K6: the gather of batch b+1 is issued before batch b is consumed, ONE workgroup barrier per batch).
//  * At staging time every thread tests "its" splat EXACTLY against the four pixel blocks of the workgroup (minimum of
//    the conic form over the block rectangle vs the splat's tau: block_reach). K6's waves compact the survivors of their
//    block into a byte list in LDS; K7's waves walk a 64-bit ballot with scalar code. Splats that cannot touch a wave's
//    pixels cost it no vector instructions; the per-pixel gates (power > 0, alpha < 1/255, T < 1e-4) are evaluated
//    unchanged on the survivors, as 64-bit lane masks on the scalar unit.
//  * K7 reduces the 10 per-splat gradient sums over the wave's 64 pixels with a transposed butterfly (24 VALU ops,
//    reduce10) that leaves them in 10 different lanes: ONE global_atomic_add_f32 instruction commits a splat.
//  * Load balance: tiles differ in cost by orders of magnitude (empty / silhouette / deep). The work lists are ordered
//    heaviest-first on the device (k_work_order_fwd / _bwd) and workgroup b simply takes item b: the hardware
//    dispatcher hands workgroups out in index order as slots free up, i.e. longest-processing-time-first scheduling.
//    Several views share a launch through a 1-D grid over (item, view) -- see the kernel wrappers below.
//  * Both kernels are bound by VALU issue (SQ counters: > 80 % of the issue slots): the code below is shaped by
//    instruction count (scalar lane masks, v_med3, v_rcp, integer min on float bits, no packed fp32 -- the file is built
//    with -fno-slp-vectorize, see build.py and DESIGN.md).
// Semantics: SURVEY.md Appendix A.2 / A.3, SEMANTICS.md; outputs as consumed at scene_gaussian.py:1012-1032.
#include "gsr_common.h"
#include <cstdlib>

// exp() and the quadratic form of the compositing loops are DEFINED operation by operation (SEMANTICS.md section 4) and
// evaluated identically here and in oracle/gsr_oracle.c (orc_exp, orc_power): the hard gates (power > 0,
// alpha < 1/255, T < 1e-4) are discontinuities, and any rounding difference in front of them puts about one
// (pixel, splat) pair per 10^6 pixels on the other side. With IEEE operations only (no v_exp_f32, whose bits are the
// hardware's) both sides produce the same bits, so n_contrib and final_T are bit-exact against the oracle.
//   exp:   t = x * float(log2 e); n = rint(t); f = t - n; p = Horner degree 5 in f (fma); result = ldexp(p, int(n))
//          (10 full-rate VALU operations; the former compensated v_exp_f32 version took 6 incl. one transcendental)
//   power: dx * (hA dx + nB dy) + (hC dy) dy with hA = -A/2, nB = -B, hC = -C/2 formed (exactly) when a splat is
//          staged: 3 multiplies + 2 fma per evaluation instead of 5 + 2.
namespace {

constexpr int kBatch = 256;

// One IEEE rounding per operation, never contracted into an FMA (HIP's __fmul_rn / __fsub_rn are plain operators and
// WOULD be contracted under the default -ffp-contract=fast; the pragma removes the `contract` flag from the
// instructions generated inside these functions, and inlining keeps instruction flags).
__device__ __forceinline__ float gsr_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float gsr_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float gsr_exp(float x) {
  const float t = gsr_mul(x, 1.44269502162933349609375f);
  const float n = __builtin_rintf(t);
  const float f = gsr_sub(t, n);
  float p = 0.001326472731307149f;
  p = __fmaf_rn(p, f, 0.009671512991189957f);
  p = __fmaf_rn(p, f, 0.05550733581185341f);
  p = __fmaf_rn(p, f, 0.24022242426872253f);
  p = __fmaf_rn(p, f, 0.6931470036506653f);
  p = __fmaf_rn(p, f, 1.0f);
  return __builtin_amdgcn_ldexpf(p, gsr_f2i_sat_fast(n));
}
// power from the staged (hA, nB, hC) = (-A/2, -B, -C/2)
__device__ __forceinline__ float gsr_power(float hA, float nB, float hC, float dx, float dy) {
  return __fmaf_rn(dx, __fmaf_rn(hA, dx, gsr_mul(nB, dy)), gsr_mul(gsr_mul(hC, dy), dy));
}

struct TilePix {
  int px, py, bx, by;   // pixel, and origin of the wave's 8x8 block
  bool inside;
};

__device__ __forceinline__ TilePix tile_pixel(int tile, int gx, int W, int H) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ty = tile / gx, tx = tile - ty * gx;
  TilePix p;
  p.bx = tx * GSR_TILE + (wave & 1) * 8;
  p.by = ty * GSR_TILE + (wave >> 1) * 8;
  p.px = p.bx + (lane & 7);
  p.py = p.by + (lane >> 3);
  p.inside = (p.px < W) && (p.py < H);
  return p;
}

// Can the splat pass the alpha gate anywhere on the W1 x W1 pixel block with origin (bx, by)?  Exact up to the
// inflation of tau: minimum of the convex form q(d) = A dx^2 + 2 B dx dy + C dy^2 over the block's rectangle (in
// d = centre - pixel coordinates) compared with tau. A convex function whose free minimum (d = 0) lies outside the
// rectangle takes its minimum on a face the centre sees from outside -- at most one vertical and one horizontal
// face. ex / ey = the coordinate of the rectangle nearest to 0 on each axis (the facing face, or 0 when the centre
// is inside that axis range, which only adds interior points); on each of the two lines q is a parabola with a
// clamped vertex. Centre inside the rectangle: ex = ey = 0 and both candidates are 0.
template <int W1>
__device__ __forceinline__ bool block_reach(float A, float C, float B2, float nBiA, float nBiC, float tau, float cx,
                                            float cy, float bx, float by) {
  const float dx1 = cx - bx, dx0 = dx1 - (float)(W1 - 1), dy1 = cy - by, dy0 = dy1 - (float)(W1 - 1);
  const float ex = __builtin_amdgcn_fmed3f(dx0, 0.f, dx1), ey = __builtin_amdgcn_fmed3f(dy0, 0.f, dy1);
  const float y = __builtin_amdgcn_fmed3f(dy0, nBiC * ex, dy1);
  const float q1 = A * ex * ex + (B2 * ex + C * y) * y;
  const float x = __builtin_amdgcn_fmed3f(dx0, nBiA * ey, dx1);
  const float q2 = C * ey * ey + (B2 * ey + A * x) * x;
  return fminf(q1, q2) <= tau;
}

// 4-bit mask over the 2x2 grid of W1 x W1 blocks with origin (x0, y0): bit (wave index) set = the splat can reach it.
// W1 = 8: the four 8x8 blocks of a 16x16 tile (K7); W1 = 4: the four 4x4 blocks of an 8x8 quarter (K6).
template <int W1>
__device__ __forceinline__ uint32_t block_mask_t(const float4 q0, const float4 q1, const float4 q2, int x0i, int y0i) {
  const float tau = q2.z;
  if (!(tau >= 0.f)) return 0u;
  const float A = q0.z, B = q0.w, C = q1.x;
  // vertex of q along a vertical line dx = e: dy = -B e / C; along a horizontal line dy = e: dx = -B e / A
  const float nBiA = -B * __builtin_amdgcn_rcpf(A), nBiC = -B * __builtin_amdgcn_rcpf(C), B2 = 2.f * B;
  const float x0 = (float)x0i, y0 = (float)y0i;
  uint32_t m = 0;
  m |= (uint32_t)block_reach<W1>(A, C, B2, nBiA, nBiC, tau, q0.x, q0.y, x0, y0);
  m |= (uint32_t)block_reach<W1>(A, C, B2, nBiA, nBiC, tau, q0.x, q0.y, x0 + (float)W1, y0) << 1;
  m |= (uint32_t)block_reach<W1>(A, C, B2, nBiA, nBiC, tau, q0.x, q0.y, x0, y0 + (float)W1) << 2;
  m |= (uint32_t)block_reach<W1>(A, C, B2, nBiA, nBiC, tau, q0.x, q0.y, x0 + (float)W1, y0 + (float)W1) << 3;
  return m;
}

template <int CTRL>
__device__ __forceinline__ int gsr_dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Block-wide exclusive scan helper for a single 1024-thread workgroup walking an array in chunks of 1024.
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t x, uint32_t* wave_tot, uint32_t* carry_s, bool update) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += wave_tot[w];
  const uint32_t carry = *carry_s;
  const uint32_t excl = carry + woff + inc - x;
  __syncthreads();
  if (update && tid == 1023) *carry_s = carry + woff + inc;
  __syncthreads();
  return excl;
}

// Forward work list: tile ids ordered heaviest-first (length classes of floor(log2(len)) with GSR_ORDER_FRAC_BITS more bits,
// descending; the order inside a class is arbitrary); zeroes tile_depth. Single workgroup.
// Several views at once: workgroup blockIdx.x builds the list of view blockIdx.x (pointer tables in the kernel arguments).
struct WorkFwdViews {
  const uint32_t* ranges[GSR_MAX_BATCH_VIEWS];
  uint32_t* tile_depth[GSR_MAX_BATCH_VIEWS];
  uint32_t* work[GSR_MAX_BATCH_VIEWS];
  uint32_t* stats_host[GSR_MAX_BATCH_VIEWS];
};
struct WorkBwdViews {
  const uint32_t* tile_depth[GSR_MAX_BATCH_VIEWS];
  uint32_t* items[GSR_MAX_BATCH_VIEWS];
  uint32_t items_cap[GSR_MAX_BATCH_VIEWS];
  uint32_t seg_len[GSR_MAX_BATCH_VIEWS];
};

// Length classes per octave of the forward work list = 2^GSR_ORDER_FRAC_BITS. Round 6, one call, two interleaved runs each: 1 / 4 /
// 8 classes per octave: K6 44.6 / 44.3 / 44.2 us per view at C3 (value 5 059 / 5 088 / 4 994), 108.4 / 107.6 / 107.8 in the
// opacity-0.1 state -- the heaviest-first order inside an octave is worth half a percent, finer than 4 nothing (gpurun_out/r6f).
#ifndef GSR_ORDER_FRAC_BITS
#define GSR_ORDER_FRAC_BITS 2
#endif
constexpr int kOrderFrac = GSR_ORDER_FRAC_BITS;
constexpr int kOrderClasses = 2 + (32 << kOrderFrac);
__device__ __forceinline__ uint32_t order_class(uint32_t len) {
  if (len == 0u) return 0u;
  const int msb = 31 - __clz(len);
  if constexpr (kOrderFrac == 0) return (uint32_t)msb + 1u;
  const uint32_t frac = (msb >= kOrderFrac ? (len >> (msb - kOrderFrac)) : (len << (kOrderFrac - msb))) & ((1u << kOrderFrac) - 1u);
  return 1u + ((uint32_t)msb << kOrderFrac) + frac;
}
__global__ void __launch_bounds__(1024)
k_work_order_fwd(const uint32_t n_tiles, const WorkFwdViews wv) {
  const uint32_t* __restrict__ ranges = wv.ranges[blockIdx.x];
  uint32_t* __restrict__ tile_depth = wv.tile_depth[blockIdx.x];
  uint32_t* __restrict__ work = wv.work[blockIdx.x];
  uint32_t* __restrict__ stats_host = wv.stats_host[blockIdx.x];
  __shared__ uint32_t cnt[kOrderClasses], cur[kOrderClasses];
  const int tid = threadIdx.x;
  if (tid < kOrderClasses) cnt[tid] = 0;
  __syncthreads();
  for (uint32_t t = tid; t < n_tiles; t += 1024) {
    const uint32_t len = ranges[2 * t + 1] - ranges[2 * t];
    atomicAdd(&cnt[order_class(len)], 1u);
    tile_depth[t] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (int b = kOrderClasses - 1; b >= 0; --b) { cur[b] = run; run += cnt[b]; }
    work[n_tiles] = n_tiles - cnt[0];          // number of non-empty tiles (a statistic for the host's mode choice)
    if (stats_host) *stats_host = n_tiles - cnt[0];   // page-locked host word, written straight from the kernel
  }
  __syncthreads();
  for (uint32_t t = tid; t < n_tiles; t += 1024) {
    const uint32_t len = ranges[2 * t + 1] - ranges[2 * t];
    work[atomicAdd(&cur[order_class(len)], 1u)] = t;
  }
}

// Backward work list: one item (tile, segment) per started kb-entry segment of [0, tile_depth[tile]) -- kb = GsrBinning.seg_len,
// the distance of the forward's checkpoints. Order = longest processing time first for the in-order hardware dispatch: full
// segments by what is left of the tile's depth behind their start (r = d - kb s: the more is left, the more pixels are still
// alive; segment 0 of a deep tile is the heaviest item there is, the last full segment of any tile the lightest), 16
// buckets of 256 entries; then the partial tails, longest first, 16 buckets. (Round 4, one call: against "all full segments in
// tile order, then the tails" K7 -1 % ... -2 % in every configuration and this kernel 8.8 -> 6.7 us.)
// items[0] = number of items, items[2 + 2 i] = tile, items[3 + 2 i] = segment. Single workgroup per view.
__global__ void __launch_bounds__(1024)
k_work_order_bwd(const uint32_t n_tiles, const WorkBwdViews wv) {
  const uint32_t* __restrict__ tile_depth = wv.tile_depth[blockIdx.x];
  uint32_t* __restrict__ items = wv.items[blockIdx.x];
  const uint32_t items_cap = wv.items_cap[blockIdx.x];
  const uint32_t kb = wv.seg_len[blockIdx.x];
  __shared__ uint32_t cnt[32], cur[32];
  const int tid = threadIdx.x;
  if (tid < 32) cnt[tid] = 0;
  __syncthreads();
  for (uint32_t t = tid; t < n_tiles; t += 1024) {
    const uint32_t d = tile_depth[t];
    if (d == 0) continue;
    const uint32_t full = d / kb, tail = d % kb;
    for (uint32_t sgi = 0; sgi < full; ++sgi) atomicAdd(&cnt[15u - min(15u, (d - kb * sgi - 1u) >> 8)], 1u);
    if (tail) atomicAdd(&cnt[16u + 15u - ((tail - 1u) * 16u) / kb], 1u);
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (int b = 0; b < 32; ++b) { cur[b] = run; run += cnt[b]; }
    items[0] = min(run, items_cap);
  }
  __syncthreads();
  for (uint32_t t = tid; t < n_tiles; t += 1024) {
    const uint32_t d = tile_depth[t];
    if (d == 0) continue;
    const uint32_t full = d / kb, tail = d % kb;
    for (uint32_t sgi = 0; sgi < full; ++sgi) {
      const uint32_t i = atomicAdd(&cur[15u - min(15u, (d - kb * sgi - 1u) >> 8)], 1u);
      if (i < items_cap) { items[2 + 2 * i] = t; items[3 + 2 * i] = sgi; }
    }
    if (tail) {
      const uint32_t i = atomicAdd(&cur[16u + 15u - ((tail - 1u) * 16u) / kb], 1u);
      if (i < items_cap) { items[2 + 2 * i] = t; items[3 + 2 * i] = full; }
    }
  }
}

// Double-buffered staging of 256 list entries. The reach mask of an entry rides in the unused fourth component of its
// third splat row (s2[..].w): 24 KB instead of 29 KB per workgroup = one more workgroup per CU; the Gaussian ids are only
// staged by the score variant.
template <bool SCORE, int PAD = 0>
struct Stage {
  float4 s0[2][kBatch + PAD], s1[2][kBatch + PAD], s2[2][kBatch + PAD];
  uint32_t sid[SCORE ? 2 : 1][SCORE ? kBatch : 1];
  // score variant, pixel counts (score_mode 0 / 2): per staged entry, the pixels of the workgroup's 8x8 quarter that composited
  // it -- added with LDS integer atomics by the four waves while they walk the batch and flushed to the global counters ONCE per
  // batch by the thread that staged the entry (round 5: one global atomic per (wave, step, slot) made k_render_fwd<true> 2.4x
  // the plain kernel: 127 vs 53 us for one view of C3, profiles/r05_score_kernel_stats.txt)
  uint32_t cnt[SCORE ? 2 : 1][SCORE ? kBatch : 1];
};
__device__ __forceinline__ uint32_t stage_mask(const float4& s2row) { return __float_as_uint(s2row.w); }

// --------------------------------------------------------------------------------------------------------- K6
// Forward compositing, "list-parallel lanes": FOUR lanes share a pixel and take four consecutive candidates of
// the list per step:
//   * work item = one 8x8 pixel quarter of a 16x16 tile, handled by a 256-thread workgroup; wave = 4x4 pixels;
//     lane = 16 (pixel row) + 4 slot + pixel column;
//   * each lane evaluates alpha of "its" candidate; the transmittance in front of it is T * (exclusive product
//     of the earlier slots' (1-alpha)) -- three bank-masked DPP multiplies (row_shr:4), no LDS; the T < 1e-4 stop is one
//     comparison per lane; the new T is the minimum over the slots of the survivors' T(1-alpha) (row_ror:4, :8);
//   * colour / depth / alpha partial sums stay per lane and are folded over the slots once, at the end.
// 4x more (and 4x finer) work items, tighter 4x4 culling; same gates in the same list order on the same bits (the
// transmittance is multiplied up in list order inside the quad; only the colour / depth / alpha SUMS associate
// differently from a sequential loop, at the 1e-7 level).
// KB = distance of the checkpoints in list entries (GsrBinning.seg_len: 256, 128 or 64). The batches stay 256 entries long;
// for KB < 256 a wave's candidate list is padded to a multiple of four at every KB boundary inside the batch, so that a
// step never straddles one, and the state is written out when the loop reaches that point.
template <bool SCORE, int KB>
__device__ __forceinline__ void
render_fwd_body(const uint32_t item, const int W, const int H, const uint32_t* __restrict__ work, float* __restrict__ ckpt,
             const uint32_t* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const float4* __restrict__ splat, const float* __restrict__ bg,
             float* __restrict__ out_color, float* __restrict__ out_da, float* __restrict__ final_T,
             uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_depth, float* __restrict__ score,
             const int score_mode) {
  // Row kBatch of every staged array is a candidate no pixel takes (opacity 0 -> alpha 0 < 1/255): the per-wave candidate
  // lists are padded with it to a multiple of four, so the compositing loop has no partial step.
  __shared__ Stage<SCORE, 1> st;
  __shared__ uint16_t cand[4][kBatch + 16];  // per wave: the batch's candidates for its 4x4 block, in list order (+ padding)
  if (threadIdx.x < 2) {
    st.s0[threadIdx.x][kBatch] = make_float4(0.f, 0.f, 0.f, 0.f);
    st.s1[threadIdx.x][kBatch] = make_float4(0.f, 0.f, 0.f, 0.f);
    st.s2[threadIdx.x][kBatch] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (SCORE) { st.cnt[0][threadIdx.x] = 0u; st.cnt[1][threadIdx.x] = 0u; }
  // pixel counts of the batch staged in buffer b -> the global counters (thread t flushes the entry it staged, then clears it)
  auto flush_counts = [&](const int b) {
    if constexpr (SCORE) {
      const uint32_t c = st.cnt[b][threadIdx.x];
      if (c) {
        atomicAdd(reinterpret_cast<uint32_t*>(score) + st.sid[b][threadIdx.x], c);
        st.cnt[b][threadIdx.x] = 0u;
      }
    }
  };
  const int gx = (W + GSR_TILE - 1) / GSR_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // lane = 16 (pixel row) + 4 slot + (pixel column): the slots of a pixel are the four BANKS of a DPP row, so a step of the
  // transmittance scan is one bank-masked v_mul_f32_dpp row_shr:4 (lanes of the other banks keep their value) instead of a
  // quad_perm multiply plus a select
  const int slot = (lane >> 2) & 3, pcol = lane & 3, prow = lane >> 4;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  {
    // Workgroup b takes item b of the heaviest-first work list: the hardware dispatcher hands workgroups out in
    // index order as CU slots free up, i.e. it performs longest-processing-time-first scheduling for us.
    const uint32_t tile = work[item >> 2];
    const int quarter = (int)(item & 3u);
    const int ty = (int)tile / gx, tx = (int)tile - ty * gx;
    const int q_x0 = tx * GSR_TILE + (quarter & 1) * 8, q_y0 = ty * GSR_TILE + (quarter >> 1) * 8;
    const int px = q_x0 + (wave & 1) * 4 + pcol, py = q_y0 + (wave >> 1) * 4 + prow;
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    if (r0 == r1) {
      // empty tile (most of the image around an object): background only. The workgroup of quarter 0 writes the whole
      // 16x16 tile, one pixel per thread; the other three leave (a quarter's full prologue / fold / store path costs
      // ~100 instructions per wave, and four fifths of C3's workgroups are of this kind).
      if (quarter == 0) {
        const int ex = tx * GSR_TILE + (tid & 15), ey = ty * GSR_TILE + (tid >> 4);
        if (ex < W && ey < H) {
          const size_t pix = (size_t)ey * W + ex, HW = (size_t)H * W;
          final_T[pix] = 1.0f;
          n_contrib[pix] = 0u;
          out_color[pix] = bg0; out_color[HW + pix] = bg1; out_color[2 * HW + pix] = bg2;
          out_da[pix] = 0.f; out_da[HW + pix] = 0.f;
        }
      }
      return;
    }

    // pixels that are finished, as a 64-bit lane mask of the wave: all the gate logic below runs on the scalar unit
    unsigned long long donem = __builtin_amdgcn_ballot_w64(!inside);
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
    uint32_t last = 0;

    uint32_t nid = 0;
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
    if (r0 + tid < r1) {
      nid = point_list[r0 + tid];
      const float4* r = splat + 3 * (size_t)nid;
      n0 = r[0]; n1 = r[1]; n2 = r[2];
    }
    int buf = 0;
    for (uint32_t base = r0; base < r1; base += kBatch, buf ^= 1) {
      const int n = (int)min((uint32_t)kBatch, r1 - base);
      // staged with the conic as (hA, nB, hC) = (-A/2, -B, -C/2): what gsr_power takes (exact scalings)
      st.s0[buf][tid] = make_float4(n0.x, n0.y, -0.5f * n0.z, -n0.w);
      st.s1[buf][tid] = make_float4(-0.5f * n1.x, n1.y, n1.z, n1.w);
      st.s2[buf][tid] = make_float4(n2.x, n2.y, n2.z,
                                    __uint_as_float((tid < n) ? block_mask_t<4>(n0, n1, n2, q_x0, q_y0) : 0u));
      if constexpr (SCORE) st.sid[buf][tid] = nid;      // (the previous batch's ids and counts live in buffer buf ^ 1)
      const int n_done = __syncthreads_count(__builtin_amdgcn_inverse_ballot_w64(donem));
      // every wave is past the compositing of the previous batch (buffer buf ^ 1): its counts are complete
      if (score_mode != 1 && base != r0) flush_counts(buf ^ 1);
      if (n_done == 256) break;
      // checkpoint of the per-pixel prefix state at list position pos (r0 + a multiple of KB): lets the backward start a
      // traversal there (k_render_bwd splits deep tiles into independent segments of KB entries)
      auto checkpoint = [&](const uint32_t pos) {
        if (ckpt == nullptr) return;     // forward only (GsrImages.ckpt NULL): nobody will start a backward traversal here
        float f0 = C0, f1 = C1, f2 = C2, f3 = Dp, f4 = Wt;       // folds over the slots (row_ror:4, :8): all lanes take part
        f0 += gsr_dpp<0x124>(f0); f0 += gsr_dpp<0x128>(f0);
        f1 += gsr_dpp<0x124>(f1); f1 += gsr_dpp<0x128>(f1);
        f2 += gsr_dpp<0x124>(f2); f2 += gsr_dpp<0x128>(f2);
        f3 += gsr_dpp<0x124>(f3); f3 += gsr_dpp<0x128>(f3);
        f4 += gsr_dpp<0x124>(f4); f4 += gsr_dpp<0x128>(f4);
        if (slot == 0 && inside) {
          // slot = absolute list position / KB: boundaries of one list are KB apart and the first boundary of a tile lies
          // >= KB entries after the end of the previous tile's list, so slots never collide
          float* ck = ckpt + (size_t)(pos / KB) * (6 * 256) + ((py - ty * GSR_TILE) * GSR_TILE + (px - tx * GSR_TILE));
          ck[0] = T; ck[256] = f0; ck[512] = f1; ck[768] = f2; ck[1024] = f3; ck[1280] = f4;
        }
      };
      if (base != r0) checkpoint(base);
      {
        const uint32_t idx = base + kBatch + tid;
        if (idx < r1) {
          nid = point_list[idx];
          const float4* r = splat + 3 * (size_t)nid;
          n0 = r[0]; n1 = r[1]; n2 = r[2];
        }
      }
      // this wave's candidates of the batch (entries whose reach mask has the wave's 4x4 block), compacted in list
      // order into a byte list of its own: the compositing loop then reads "its" candidate with one LDS load instead
      // of peeling four bits off a 64-bit scalar mask per step, and only the last step of a batch can be partial
      int cnt = 0;
      int cut_at[3] = {-1, -1, -1};        // step index of the KB boundaries inside the batch (KB < 256)
#pragma unroll
      for (int k = 0; k < kBatch / 64; ++k) {
        if (k * 64 >= n) break;
        const bool m = (stage_mask(st.s2[buf][k * 64 + lane]) >> wave) & 1u;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(m);
        if (m) cand[wave][cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint8_t)(k * 64 + lane);
        cnt += (int)__popcll(bal);
        if constexpr (KB < kBatch) {
          if (((k + 1) * 64) % KB == 0 && k + 1 < kBatch / 64 && (k + 1) * 64 < n) {
            const int pad = (-cnt) & 3;
            if (lane < pad) cand[wave][cnt + lane] = (uint16_t)kBatch;
            cnt += pad;
            cut_at[(k + 1) * 64 / KB - 1] = cnt;
          }
        }
      }
      if (lane < 3) cand[wave][cnt + lane] = (uint16_t)kBatch;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      {
        for (int i = 0;; i += 4) {
          if (donem == ~0ull) break;       // (no pixel of the wave composites any further: nobody reads its checkpoints)
          if constexpr (KB < kBatch) {
#pragma unroll
            for (int c = 0; c < kBatch / KB - 1; ++c)
              if (i == cut_at[c]) checkpoint(base + (uint32_t)((c + 1) * KB));
          }
          if (i >= cnt) break;
          const int j = (int)cand[wave][i + slot];
          const float4 a = st.s0[buf][j];
          const float4 b = st.s1[buf][j];
          const float2 c = *reinterpret_cast<const float2*>(&st.s2[buf][j]);
          const float dx = a.x - pxf, dy = a.y - pyf;
          const float power = gsr_power(a.z, a.w, b.x, dx, dy);
          const float alpha = fminf(GSR_ALPHA_MAX, gsr_mul(b.y, gsr_exp(power)));
          const unsigned long long gm = ~donem & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                        __builtin_amdgcn_ballot_w64(alpha >= GSR_ALPHA_MIN);
          const bool g = __builtin_amdgcn_inverse_ballot_w64(gm);
          // transmittance after this slot, multiplied up in LIST ORDER -- ((T f0) f1) f2 ... with f = 1 - alpha of a
          // gated slot, 1 otherwise -- so that it carries the bits of the sequential recurrence T <- T (1 - alpha) (the
          // T < 1e-4 stop is a hard gate: SEMANTICS.md section 4). Three dependent quad steps: after step k slot k is final.
          const float fgate = g ? gsr_sub(1.0f, alpha) : 1.0f;
          float test_T = gsr_mul(T, fgate);
          // (each step: the lanes of slots >= k take test_T of slot - 1 times their own factor; s_nop: a DPP operand needs two
          //  wait states after its producer and the hazard recognizer does not look inside an asm block)
          asm("s_nop 1\n\t"
              "v_mul_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
              "s_nop 1\n\t"
              "v_mul_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xc\n\t"
              "s_nop 1\n\t"
              "v_mul_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0x8"
              : "+v"(test_T) : "v"(fgate));
          // alpha times the transmittance in FRONT of the slot: T for slot 0, test_T of slot - 1 for the others
          float w_all = alpha * T;
          asm("s_nop 1\n\t"
              "v_mul_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xe"
              : "+v"(w_all) : "v"(test_T), "v"(alpha));
          // The first slot (in list order) whose own contribution would drop T below the threshold stops the pixel, it
          // and everything behind it. A pixel still in play holds T >= 1e-4 and the slots' test_T only decrease along the
          // list (factors <= 1, one rounding each), so "a gated slot <= mine fell below the threshold" IS "my test_T is
          // below it": no scan over the quad, and the quad's stop flag is slot 3's comparison.
          const unsigned long long stopm = __builtin_amdgcn_ballot_w64(test_T < GSR_T_MIN);
          const bool hit = __builtin_amdgcn_inverse_ballot_w64(gm & ~stopm);
          const float w = hit ? w_all : 0.0f;
          C0 = fmaf(b.w, w, C0); C1 = fmaf(c.x, w, C1); C2 = fmaf(c.y, w, C2);
          Dp = fmaf(b.z, w, Dp);
          Wt += w;
          last = hit ? ((base - r0) + (uint32_t)j + 1u) : last;
          if constexpr (SCORE) {
            // the pixels of the wave that composite slot s's splat: the 16 lanes holding this slot
            const unsigned long long hm = __ballot(hit) & (0x000F000F000F000Full << (4 * slot));
            if (score_mode != 1) {
              // weight = opacity per contributing (pixel, splat): the kernel counts the pixels -- integer atomics, exact and
              // independent of the order -- and k_score_finalize multiplies by the opacity once (mode 0) or the caller does
              // (mode 2: raw counts, summed over many views first). A float sum of thousands of EQUAL increments rounds the
              // same way every time (measured 6e-5 relative on the sum over 48 views).
              // (LDS integer atomic: the four waves of the quarter meet in one counter per staged entry; the padding row
              //  kBatch never hits: alpha 0)
              if (hm != 0ull && lane == 4 * slot) atomicAdd(&st.cnt[buf][j & (kBatch - 1)], (uint32_t)__popcll(hm));
            } else {
              float ws = w;                               // sum over the lanes sharing the slot: xor 1, 2, 16, 32
              ws += gsr_dpp<0xB1>(ws);                    // quad_perm [1,0,3,2]
              ws += gsr_dpp<0x4E>(ws);                    // quad_perm [2,3,0,1]
              ws += __shfl_xor(ws, 16, 64);
              ws += __shfl_xor(ws, 32, 64);
              if (hm != 0ull && lane == 4 * slot) unsafeAtomicAdd(score + st.sid[buf][j & (kBatch - 1)], ws);
            }
          }
          // T after the quad: the survivors' T(1-alpha) only decrease along the list -> quad minimum (T >= 1e-4 > 0:
          // the order of positive floats is the order of their bit patterns, and v_min_u32 takes a DPP operand)
          uint32_t tn = __float_as_uint(hit ? test_T : T);
          tn = min(tn, (uint32_t)gsr_dpp_i<0x124>((int)tn));   // row_ror:4
          tn = min(tn, (uint32_t)gsr_dpp_i<0x128>((int)tn));   // row_ror:8
          T = __uint_as_float(tn);
          // slot 3's flag -> all four lanes of its quad. On the SCALAR unit: its instructions issue beside the vector
          // instructions of the other waves (removing 22 of them from this loop changed nothing: A/B in one gpurun call,
          // 44.0 vs 45.0 us per view -- the loop is bound by its ~50 vector instructions), so mask arithmetic belongs there
          unsigned long long quad_stop = (stopm >> 12) & 0x000F000F000F000Full;
          quad_stop |= quad_stop << 4;
          quad_stop |= quad_stop << 8;
          donem |= quad_stop;
        }
      }
    }
    if constexpr (SCORE) {
      // the counts of the last batch that was composited (buffer buf ^ 1 after the loop's own flip; all zero when the loop left
      // through the "everything finished" exit, whose flush ran already)
      __syncthreads();
      if (score_mode != 1) flush_counts(buf ^ 1);
    }
    // fold the four slots of each pixel
    C0 += gsr_dpp<0x124>(C0); C0 += gsr_dpp<0x128>(C0);
    C1 += gsr_dpp<0x124>(C1); C1 += gsr_dpp<0x128>(C1);
    C2 += gsr_dpp<0x124>(C2); C2 += gsr_dpp<0x128>(C2);
    Dp += gsr_dpp<0x124>(Dp); Dp += gsr_dpp<0x128>(Dp);
    Wt += gsr_dpp<0x124>(Wt); Wt += gsr_dpp<0x128>(Wt);
    {
      int l = (int)last;
      l = max(l, gsr_dpp_i<0x124>(l));
      l = max(l, gsr_dpp_i<0x128>(l));
      last = (uint32_t)l;
    }
    if (inside && slot == 0) {
      const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
      final_T[pix] = T;
      n_contrib[pix] = last;
      out_color[pix] = C0 + T * bg0;
      out_color[HW + pix] = C1 + T * bg1;
      out_color[2 * HW + pix] = C2 + T * bg2;
      out_da[pix] = Dp;
      out_da[HW + pix] = Wt;
    }
    // deepest contributor of the tile: the backward's cost key and its starting depth
    const uint32_t wm = gsr_wave_max_u32(last);
    if (lane == 0 && wm) atomicMax(tile_depth + tile, wm);
  }
}

// K6, whole-tile variant: work item = one 16x16 tile, 4 waves = four 8x8 blocks, ONE pixel per lane. It spends the
// fewest instructions per (pixel, splat) evaluation (about half of the list-parallel kernel) and is the right
// choice when thousands of similar, shallow tiles saturate the machine (camera inside a room: every tile active,
// ~250 entries each); with few, deep tiles its long per-pixel chains make the tail (1 M Gaussians at 512^2:
// 377 us vs 124 us). The host picks the variant per call from the previous view's statistics
// (GsrBinning.fwd_mode); both produce the same images up to the association of the transmittance product.
template <bool SCORE>
__device__ __forceinline__ void
render_fwd_tile_body(const uint32_t item, const int W, const int H, const uint32_t* __restrict__ work, float* __restrict__ ckpt,
                  const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                  const float4* __restrict__ splat, const float* __restrict__ bg, float* __restrict__ out_color,
                  float* __restrict__ out_da, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                  uint32_t* __restrict__ tile_depth, float* __restrict__ score, const int score_mode) {
  __shared__ Stage<SCORE> st;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE;
  const int tile = (int)work[item];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const TilePix p = tile_pixel(tile, gx, W, H);
  const int tile_x0 = p.bx - (wave & 1) * 8, tile_y0 = p.by - (wave >> 1) * 8;
  const float pxf = (float)p.px, pyf = (float)p.py;
  const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
  unsigned long long donem = __builtin_amdgcn_ballot_w64(!p.inside);   // finished pixels, as a lane mask
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
  uint32_t last = 0;
  uint32_t nid = 0;
  float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
  if (r0 + tid < r1) {
    nid = point_list[r0 + tid];
    const float4* r = splat + 3 * (size_t)nid;
    n0 = r[0]; n1 = r[1]; n2 = r[2];
  }
  int buf = 0;
  for (uint32_t base = r0; base < r1; base += kBatch, buf ^= 1) {
    const int n = (int)min((uint32_t)kBatch, r1 - base);
    st.s0[buf][tid] = make_float4(n0.x, n0.y, -0.5f * n0.z, -n0.w);      // conic staged as (hA, nB, hC)
    st.s1[buf][tid] = make_float4(-0.5f * n1.x, n1.y, n1.z, n1.w);
    st.s2[buf][tid] = make_float4(n2.x, n2.y, n2.z,
                                  __uint_as_float((tid < n) ? block_mask_t<8>(n0, n1, n2, tile_x0, tile_y0) : 0u));
    if constexpr (SCORE) st.sid[buf][tid] = nid;
    if (__syncthreads_count(__builtin_amdgcn_inverse_ballot_w64(donem)) == 256) break;
    if (base != r0 && p.inside && ckpt != nullptr) {
      float* ck = ckpt + (size_t)(base / kBatch) * (6 * 256) + ((p.py - tile_y0) * GSR_TILE + (p.px - tile_x0));
      ck[0] = T; ck[256] = C0; ck[512] = C1; ck[768] = C2; ck[1024] = Dp; ck[1280] = Wt;
    }
    {
      const uint32_t idx = base + kBatch + tid;
      if (idx < r1) {
        nid = point_list[idx];
        const float4* r = splat + 3 * (size_t)nid;
        n0 = r[0]; n1 = r[1]; n2 = r[2];
      }
    }
    for (int k = 0; k < kBatch / 64; ++k) {
      if (k * 64 >= n) break;
      unsigned long long bits = __ballot((stage_mask(st.s2[buf][k * 64 + lane]) >> wave) & 1u);
      while (bits) {
        if (donem == ~0ull) break;
        const int j = k * 64 + __builtin_ctzll(bits);
        bits &= bits - 1ull;
        const float4 a = st.s0[buf][j];
        const float4 b = st.s1[buf][j];
        const float4 c = st.s2[buf][j];
        const float dx = a.x - pxf, dy = a.y - pyf;
        const float power = gsr_power(a.z, a.w, b.x, dx, dy);
        const float alpha = fminf(GSR_ALPHA_MAX, gsr_mul(b.y, gsr_exp(power)));
        const float test_T = gsr_mul(T, gsr_sub(1.0f, alpha));
        const unsigned long long gm = ~donem & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                      __builtin_amdgcn_ballot_w64(alpha >= GSR_ALPHA_MIN);
        const unsigned long long stopm = gm & __builtin_amdgcn_ballot_w64(test_T < GSR_T_MIN);
        donem |= stopm;
        const bool hit = __builtin_amdgcn_inverse_ballot_w64(gm & ~stopm);
        const float w = hit ? alpha * T : 0.0f;
        if constexpr (SCORE) {
          const unsigned long long hm = __ballot(hit);       // one atomic per (wave, splat), not per pixel
          if (hm) {
            if (score_mode != 1) {      // pixel counts, integer atomics (see render_fwd_body)
              if (lane == 0) atomicAdd(reinterpret_cast<uint32_t*>(score) + st.sid[buf][j], (uint32_t)__popcll(hm));
            } else {
              float sc = gsr_wave_sum_to_lane63(w);
              sc = __shfl(sc, 63, 64);
              if (lane == 0) unsafeAtomicAdd(score + st.sid[buf][j], sc);
            }
          }
        }
        C0 = fmaf(b.w, w, C0); C1 = fmaf(c.x, w, C1); C2 = fmaf(c.y, w, C2);
        Dp = fmaf(b.z, w, Dp);
        Wt += w;
        T = hit ? test_T : T;
        last = hit ? ((base - r0) + (uint32_t)j + 1u) : last;
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
  const uint32_t wm = gsr_wave_max_u32(last);
  if (lane == 0 && wm) atomicMax(tile_depth + tile, wm);
}

// --------------------------------------------------------------------------------------------------------- K7
// Transposed butterfly over the 64 lanes for 10 values (see file header): at every step a lane keeps one register of a
// pair and hands the other one to its partner, so the number of live registers halves while the sums grow:
//   10 -> 5 over lane bit 3 (row_ror:8 = lane ^ 8), -> 3 over bit 2 (row_half_mirror = lane ^ 7), -> 2 over bit 5
//   (permlane32 swap), -> 1 over bit 4 (permlane16 swap), then quad_perm xor 2 and xor 1 complete the sums.
// The steps with the most pairs use the cheapest primitive (issue costs): bits 3 and 2 are also BANKS of a
// DPP row (lanes 4i..4i+3), so "select, then add the partner's other register" is two DPP adds with complementary bank
// masks (a bank-masked DPP write leaves the other lanes' destination alone: 2 x 1.85 ns per pair); a permlane swap +
// add is 4.7 ns per pair and is left for the two steps with 2 and 1 pairs. Round 3 had the order bit 5, 4, 3, 2
// (5 + 3 swaps): 50.5 ns of issue per splat, this order 47.4.
// On return lane L (b_k = bit k of L) holds, in rows 0 and 2 (b4 = 0), the wave total of component b3 + 2 b2 + 4 b5; in row 1
// the total over lanes 0-31 of component 8 + b3 and in row 3 the total over lanes 32-63 of the same (word 10 + b3).
__device__ __forceinline__ float add_swap32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_swap16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float reduce10(const float v[10], int lane) {
  // (s_nop: the inputs come straight from VALU instructions and a DPP operand needs two wait states after its producer;
  //  the hazard recognizer does not look inside an asm block)
  float p0, p1, p2, p3, p4, q0, q1, q2;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %1, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %4, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %5, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %6, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %6, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %7, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf"
      : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(q0), "=&v"(q1), "=&v"(q2)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]));
  // q0: components b3 + 2 b2, q1: 4 + b3 + 2 b2, q2: 8 + b3 (in all lanes: its pair partner is the pad)
  const float r0 = add_swap32(q0, q1);      // lanes 0-31: q0 (components b3 + 2 b2), lanes 32-63: q1 (4 + ...)
  // q2 has no partner register: it skips the bit-5 step (a swap with a zero register, an add) and is committed from BOTH
  // halves of the wave into DIFFERENT words of the 12-float row -- row 1 adds its half's sums of components 8, 9 to words
  // 8, 9, row 3 to words 10, 11, K8 adds the two. (Into the same words it would be two lanes of one atomic instruction on
  // one address, and those serialise: 55 -> 102 us per view.)
  float R = add_swap16(r0, q2);             // rows 0, 2: r0 (complete), rows 1, 3: q2 summed over rows {0,1} / {2,3}
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "+v"(R));
  (void)lane;
  return R;
}

// Accumulates into partials [P,16 doubles], with q = dL/dG * G per (pixel, splat), d = centre - pixel and (u, v) = -Sigma^-1 d =
// (-(A dx + B dy), -(C dy + B dx)) (Sigma^-1 = the conic):
//   (sum q u, sum q v, sum q u^2, sum q u v, sum q v^2, dL/dopacity, dL/dr, dL/dg, dL/db, dL/ddepth, dL/db', dL/ddepth', -, -, -, -)
// (the last two pairs: the sums over the lower / upper half of a wave, added by K8 -- see reduce10)
// Inside a wave the sums are reduced in fp32 in a FIXED order (reduce10); ACROSS waves they are added by atomics in whatever
// order the workgroups arrive. In fp32 that order showed: the covariance chain amplifies sum q u^2 / q u v / q v^2 by
// cond(Sigma)^2 on needle-shaped splats and dL/dopacity collects the 7e4-weighted extremal pixels of the reference's disp
// normalisation (scene_gaussian.py:1025-1032) -- the worst dL/drotations entry of the needle case moved between 2.7e-6 and
// 1.5e-5 from run to run, dL/dopacity of the boundary records between 1e-5 and 1e-4 (round 3). Now every wave result is
// added in DOUBLE (global_atomic_add_f64: 29 spare mantissa bits over the fp32 addends -- the sum of a splat's wave results is
// exact, hence the same in every order, as long as the addends' exponents span less than 2^29; beyond that span -- the 7e4-
// weighted extremal pixels of the disp normalisation next to a near-zero contribution -- an addend can move the double by
// 2^-53 of the sum, visible in fp32 only on a rounding tie) and K8 rounds the total to fp32 once: the backward is
// order-independent within that span (tests: same bits over eight runs, <= 4 one-ulp ties per tensor allowed).
// One atomic instruction per (splat, block) as before -- 12 lanes, one 128-byte row. (Measured first: doubles for the four
// sensitive sums only, in a second atomic instruction next to the f32 one: K7 217 -> 238 us -- two atomic instructions per
// iteration run into the atomic issue limit of ~80 ns per instruction and SIMD, DESIGN.md "Issue costs".)
// dG/dd = G (u, v), so the first two sums are dL/d(pixel centre), and dL/dSigma = 1/2 sum q (Sigma^-1 d)(Sigma^-1 d)^T, so
// the other three are the gradient of the 2-D covariance itself (K8 only scales them) -- both formed PER PIXEL, as the
// scalar oracle does (gsr_oracle.c, orc_pixel_bwd; SEMANTICS.md section 5). Rounds 1-2 summed the raw moments of d
// (sum q dx, ..., sum q dy^2) and contracted them with the conic afterwards: exact in exact arithmetic, but the
// contraction cancels digits on needle-shaped splats -- A dx and B dy are of opposite sign and equal size along the
// needle -- and the lineage's conic -> covariance step amplifies the rounding of the sums by cond(Sigma)^2 (measured
// against float64 autograd: 1e-1 on dL/dscales of 1000 : 1 splats, which the reference's scale noise + clamp(.., 0)
// produces, scene_gaussian.py:1005-1008).
//
// Work item = (tile, segment): the <= 256 list entries [256 s, min(256 (s+1), tile_depth)) of one tile, for all of
// its 256 pixels, traversed back to front. The reverse traversal of a pixel is a serial recurrence over its whole
// depth (up to thousands of splats), and the deepest tiles used to set the kernel time; segments make the items
// uniform and independent. A pixel whose last contributor lies beyond the segment starts from the forward's
// checkpoint at the segment end (prefix transmittance T_c and prefix sums C_c, D_c, W_c): the colour / depth /
// alpha composited BEHIND that point, normalised to start there, is (X_final - X_c) / T_c, which is exactly the
// `rec` state the sequential traversal would carry at that position. Other pixels start from their final state.
// Round 3, measured and not kept (both bit-identical in their results; A/B in one gpurun call, C3, 4-view launch):
//  * two pixels per lane (a wave owns a 16x8 half tile, 2 waves per item, the reduction and the atomic shared by the two
//    8x8 blocks; NOT v_pk_* arithmetic: v_pk_fma_f32 issues at half the rate of v_fma_f32 on gfx950, tools/probe): 279 us
//    against 238 -- what the shared reduction saves (37 of ~95 instructions per block) the three-way control flow and
//    its register copies give back, and half as many waves hide less latency;
//  * the four waves of a workgroup adding their sums of a splat into an LDS row (ds_add_f32, ten lanes) and the workgroup
//    committing every staged entry once at the end (2.5x fewer global lane-atomics, 16x fewer atomic instructions):
//    444 us against 238 -- float atomics on LDS are an order of magnitude slower than the global ones they replace;
//  * (and the control: the same kernel with the global atomic compiled out runs 236 us against 238 -- the atomics are
//    free, the loop is bound by instruction issue);
//  * the launch zero-filling the 142 MB of gradient buffers K8's sparse form otherwise clears itself ("K7 is VALU-bound,
//    the stores are free"): K8 100 -> 86 us, K7 236 -> 274 us. The stores are not free: K7's atomics share the path.
template <int KB>
__device__ __forceinline__ void
render_bwd_body(const uint32_t item, const int W, const int H, const uint32_t* __restrict__ items, const uint32_t* __restrict__ tile_depth,
             const float* __restrict__ ckpt,
             const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
             const float4* __restrict__ splat, const float* __restrict__ bg, const float* __restrict__ color,
             const float* __restrict__ depth_alpha, const float* __restrict__ final_T,
             const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
             const float* __restrict__ dL_dda, float* __restrict__ partials, unsigned long long* __restrict__ reach) {
  __shared__ float4 s0[KB], s1[KB], s2[KB];
  __shared__ uint32_t sid[KB], smask[KB];
  __shared__ unsigned long long hitw[KB / 64];    // staged entries some wave committed sums for (-> GsrGrads.reach)
  const int gx = (W + GSR_TILE - 1) / GSR_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (item >= items[0]) return;
 {
  const int tile = (int)items[2 + 2 * item];
  const uint32_t seg = items[3 + 2 * item];
  const uint32_t depth = tile_depth[tile];
  const uint32_t lo = seg * KB, hi = min(lo + (uint32_t)KB, depth);
  const int n = (int)(hi - lo);

  const TilePix p = tile_pixel(tile, gx, W, H);
  const int tile_x0 = p.bx - (wave & 1) * 8, tile_y0 = p.by - (wave >> 1) * 8;
  const float pxf = (float)p.px, pyf = (float)p.py;
  const uint32_t r0 = ranges[2 * tile];
  const size_t pix = (size_t)p.py * W + p.px, HW = (size_t)H * W;

  // Depths are staged RELATIVE to the depth of the segment's last entry (uniform: two scalar loads). The loop carries
  // R' = sum_k w_k (s_k - c) over the splats k composited behind, with s = <(colour, depth, 1), upstream gradient> and the
  // per-pixel constant c = zref g_depth + g_alpha, next to A = sum_k w_k (the alpha composited behind): s - R = (s' - R') +
  // c (1 - A). With the reference's disp normalisation |g_depth|, |g_alpha| reach 7e4 on the extremal pixels
  // (scene_gaussian.py:1025-1032): s and R are then ~4e5 each and, in front of an opaque object (A -> 1), cancel to
  // g_depth (z - z_behind) ~ 1e3 -- in the plain form that difference carried eps x 4e5 of rounding (dL/dopacity of the
  // boundary records: 2e-5 .. 9e-5 of max|ref|, deterministic since the sums are added in double); in this form the large
  // part c (1 - A) vanishes exactly where the cancellation happens and s' - R' is a difference of terms ~ g_depth x 0.1.
  const float zref = reinterpret_cast<const float*>(splat + 3 * (size_t)point_list[r0 + (hi - 1u)] + 1)[2];
  // stage the segment, last entry first (one gather per thread)
  {
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = make_float4(0, 0, -1.f, -1.f);
    uint32_t nid = 0;
    if (tid < n) {
      nid = point_list[r0 + (hi - 1u - (uint32_t)tid)];
      const float4* r = splat + 3 * (size_t)nid;
      n0 = r[0]; n1 = r[1]; n2 = r[2];
    }
    if (tid < KB) {                                                        // (256 threads, KB <= 256 staged rows)
      s0[tid] = make_float4(n0.x, n0.y, -0.5f * n0.z, -n0.w);            // conic staged as (hA, nB, hC)
      s1[tid] = make_float4(-0.5f * n1.x, n1.y, n1.z - zref, n1.w);
      s2[tid] = n2;
      sid[tid] = nid;
      smask[tid] = (tid < n) ? block_mask_t<8>(n0, n1, n2, tile_x0, tile_y0) : 0u;
    }
    if (tid < KB / 64) hitw[tid] = 0ull;
  }

  const float Tf = p.inside ? final_T[pix] : 0.f;
  const uint32_t last = p.inside ? n_contrib[pix] : 0u;
  float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
  if (p.inside) {
    gC0 = dL_dcolor[pix]; gC1 = dL_dcolor[HW + pix]; gC2 = dL_dcolor[2 * HW + pix];
    gD = dL_dda[pix]; gA = dL_dda[HW + pix];
  }
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float bg_dot = (bg0 * gC0 + bg1 * gC1) + bg2 * gC2;

  float T = Tf;
  float R = 0.f, A = 0.f;
  if (last > hi) {
    // this pixel keeps compositing beyond the segment: start from the forward's checkpoint at position hi
    const float* ck = ckpt + (size_t)((r0 + hi) / KB) * (6 * 256) + ((p.py - tile_y0) * GSR_TILE + (p.px - tile_x0));
    const float Tc = ck[0];
    const float inv = 1.0f / Tc;
    T = Tc;
    const float rc0 = ((color[pix] - Tf * bg0) - ck[256]) * inv;
    const float rc1 = ((color[HW + pix] - Tf * bg1) - ck[512]) * inv;
    const float rc2 = ((color[2 * HW + pix] - Tf * bg2) - ck[768]) * inv;
    const float rec_z = (depth_alpha[pix] - ck[1024]) * inv;
    const float rec_a = (depth_alpha[HW + pix] - ck[1280]) * inv;
    R = rc0 * gC0 + rc1 * gC1 + rc2 * gC2 + (rec_z - zref * rec_a) * gD;     // = R - c A  (the g_alpha terms cancel exactly)
    A = rec_a;
  }
  const float cshift = zref * gD + gA;

  // one lane per quad commits with the single atomic of a splat's 10 sums; its component: bit 3 -> 1, bit 2 -> 2, bit 5 -> 4, bit 4 -> 8 (see reduce10)
  const int b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
  const int comp = b4 ? 8 + 2 * b5 + b3 : (b3 | (b2 << 1) | (b5 << 2));      // double of the 16-double row (see reduce10)
  const bool commit = ((lane & 3) == 0) && !(b4 && b2);                      // 8 + 4 lanes, twelve different doubles
  __syncthreads();

  for (int k = 0; k < KB / 64; ++k) {
    if (k * 64 >= n) break;
    unsigned long long bits = __ballot((smask[k * 64 + lane] >> wave) & 1u);
    unsigned long long hitk = 0ull;       // (scalar unit: free next to the vector instructions of the other waves)
    while (bits) {
      const int jb = __builtin_ctzll(bits);
      const int j = k * 64 + jb;
      bits &= bits - 1ull;
      const uint32_t pos = hi - 1u - (uint32_t)j;      // 0-based list position
      const unsigned long long livem = __builtin_amdgcn_ballot_w64(pos < last);
      if (livem == 0ull) continue;
      const float4 a = s0[j];
      const float4 b = s1[j];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = gsr_power(a.z, a.w, b.x, dx, dy);
      const float G = gsr_exp(power);
      const float alpha = fminf(GSR_ALPHA_MAX, gsr_mul(b.y, G));
      // the gates as 64-bit lane masks on the scalar unit
      const unsigned long long hitm = livem & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                      __builtin_amdgcn_ballot_w64(alpha >= GSR_ALPHA_MIN);
      if (hitm == 0ull) continue;
      hitk |= 1ull << jb;
      const bool hit = __builtin_amdgcn_inverse_ballot_w64(hitm);
      // per-lane factors of the 10 sums; lanes without a hit contribute zeros (only these three are cleared)
      float qv = 0.f, wv = 0.f, gdl = 0.f;
      if (hit) {
        const float4 c = s2[j];
        // v_rcp_f32 (1 ulp): T is only reconstructed for the gradient weights here, no gate depends on it. (A Newton step
        // on the reciprocal was tried against the reference-derived boundary records, where dL/dopacity sits at 2e-5 ..
        // 9e-5: no change -- that error is the fp32 noise of the 7e4 upstream spike, tests/test_boundary_fixture.py.)
        const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
        T = T * inv;
        const float w = alpha * T;
        // R = <(colour, depth, alpha) composited behind this splat, normalised to start here; upstream gradient>: the
        // recurrence of the behind-state is linear, so its dot product with the pixel's upstream gradient can be
        // carried instead of its five components (dL/dalpha only ever needs that dot product)
        const float sdot = b.w * gC0 + c.x * gC1 + c.y * gC2 + b.z * gD;       // s - c  (b.z is staged relative to zref)
        const float t1 = 1.0f - A;
        float dL_dalpha = ((sdot - R) + cshift * t1) * T;
        dL_dalpha -= (Tf * inv) * bg_dot;
        R = alpha * sdot + (1.0f - alpha) * R;
        A = __fmaf_rn(alpha, t1, A);
        // raw moments of q = dL/dG * G over the pixels; K8 turns them into dL/dmean2D and dL/dconic
        qv = (b.y * dL_dalpha) * G;
        gdl = G * dL_dalpha;
        wv = w;
      }
      float v[10];
      {
        // (u, v) = -Sigma^-1 d:  u = -(A dx + B dy) = fma(2 hA, dx, nB dy),  v = -(C dy + B dx) = fma(2 hC, dy, nB dx)
        // (hA = -A/2, nB = -B, hC = -C/2: the doublings are exact; the same expression tree as orc_pixel_bwd)
        const float u = __fmaf_rn(a.z + a.z, dx, gsr_mul(a.w, dy)), w2 = __fmaf_rn(b.x + b.x, dy, gsr_mul(a.w, dx));
        const float m1 = qv * u, m2 = qv * w2;
        v[0] = m1; v[1] = m2;
        v[2] = m1 * u; v[3] = m1 * w2; v[4] = m2 * w2;
        v[5] = gdl;
        v[6] = wv * gC0; v[7] = wv * gC1; v[8] = wv * gC2;
        v[9] = wv * gD;
      }
      const float sred = reduce10(v, lane);
      if (commit) unsafeAtomicAdd(reinterpret_cast<double*>(partials) + (GSR_PARTIAL_WORDS / 2) * (size_t)sid[j] + comp, (double)sred);
    }
    if (reach && hitk && lane == 0) atomicOr(&hitw[k], hitk);
  }
  if (reach) {
    // the Gaussians this item committed sums for: one bit each, set by the thread that staged the entry (outside the loop:
    // a handful of 64-bit atomics per item)
    __syncthreads();
    if (tid < n && ((hitw[tid >> 6] >> (tid & 63)) & 1ull)) atomicOr(reach + (sid[tid] >> 6), 1ull << (sid[tid] & 63u));
  }
 }
}


}  // namespace

// fixed grid for strided kernels: workgroups per CU x CU count (cached per device)
static int persistent_groups(int per_cu) {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cus[dev] == 0) {
    hipDeviceProp_t p;
    cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return cus[dev] * per_cu;
}

// ---- kernels: 1-D grids over (work item, view). per_view = 0: the VIEW runs fastest -- workgroup b takes item
// b / n_views of view b % n_views, so the dispatcher (index order) hands out the heaviest items of all views first and
// the empty tail of every view's list last (skewed lists, C3: K7 64 -> 58 us per view). per_view = items per view: the
// views run one after the other -- when thousands of tiles weigh about the same (camera inside a room) the order does not
// matter for balance and keeping one view's splat records in the L2s at a time does (indoor: 8 % faster this way).
// Pointer tables in the kernel arguments; one view = tables of one
struct FwdViews {
  const uint32_t* work[GSR_MAX_BATCH_VIEWS];
  float* ckpt[GSR_MAX_BATCH_VIEWS];
  const uint32_t* ranges[GSR_MAX_BATCH_VIEWS];
  const uint32_t* point_list[GSR_MAX_BATCH_VIEWS];
  const float4* splat[GSR_MAX_BATCH_VIEWS];
  const float* bg[GSR_MAX_BATCH_VIEWS];
  float* out_color[GSR_MAX_BATCH_VIEWS];
  float* out_da[GSR_MAX_BATCH_VIEWS];
  float* final_T[GSR_MAX_BATCH_VIEWS];
  uint32_t* n_contrib[GSR_MAX_BATCH_VIEWS];
  uint32_t* tile_depth[GSR_MAX_BATCH_VIEWS];
  float* score[GSR_MAX_BATCH_VIEWS];
};
struct BwdViews {
  const uint32_t* items[GSR_MAX_BATCH_VIEWS];
  const uint32_t* tile_depth[GSR_MAX_BATCH_VIEWS];
  const float* ckpt[GSR_MAX_BATCH_VIEWS];
  const uint32_t* ranges[GSR_MAX_BATCH_VIEWS];
  const uint32_t* point_list[GSR_MAX_BATCH_VIEWS];
  const float4* splat[GSR_MAX_BATCH_VIEWS];
  const float* bg[GSR_MAX_BATCH_VIEWS];
  const float* color[GSR_MAX_BATCH_VIEWS];
  const float* depth_alpha[GSR_MAX_BATCH_VIEWS];
  const float* final_T[GSR_MAX_BATCH_VIEWS];
  const uint32_t* n_contrib[GSR_MAX_BATCH_VIEWS];
  const float* dL_dcolor[GSR_MAX_BATCH_VIEWS];
  const float* dL_dda[GSR_MAX_BATCH_VIEWS];
  float* partials[GSR_MAX_BATCH_VIEWS];
  unsigned long long* reach[GSR_MAX_BATCH_VIEWS];
};

template <bool SCORE, int KB = kBatch>
__global__ void __launch_bounds__(256)
k_render_fwd(const int W, const int H, const FwdViews fv, const int score_mode, const uint32_t n_views,
             const uint32_t per_view) {
  const uint32_t item = per_view ? blockIdx.x % per_view : blockIdx.x / n_views;
  const int y = (int)(per_view ? blockIdx.x / per_view : blockIdx.x - item * n_views);
  render_fwd_body<SCORE, KB>(item, W, H, fv.work[y], fv.ckpt[y], fv.ranges[y], fv.point_list[y], fv.splat[y], fv.bg[y],
                         fv.out_color[y], fv.out_da[y], fv.final_T[y], fv.n_contrib[y], fv.tile_depth[y], fv.score[y],
                         score_mode);
}
template <bool SCORE>
__global__ void __launch_bounds__(256)
k_render_fwd_tile(const int W, const int H, const FwdViews fv, const int score_mode, const uint32_t n_views,
                  const uint32_t per_view) {
  const uint32_t item = per_view ? blockIdx.x % per_view : blockIdx.x / n_views;
  const int y = (int)(per_view ? blockIdx.x / per_view : blockIdx.x - item * n_views);
  render_fwd_tile_body<SCORE>(item, W, H, fv.work[y], fv.ckpt[y], fv.ranges[y], fv.point_list[y], fv.splat[y], fv.bg[y],
                              fv.out_color[y], fv.out_da[y], fv.final_T[y], fv.n_contrib[y], fv.tile_depth[y],
                              fv.score[y], score_mode);
}
template <int KB = kBatch>
__global__ void __launch_bounds__(256)
k_render_bwd(const int W, const int H, const BwdViews bv, const uint32_t n_views, const uint32_t per_view) {
  const uint32_t item = per_view ? blockIdx.x % per_view : blockIdx.x / n_views;
  const int y = (int)(per_view ? blockIdx.x / per_view : blockIdx.x - item * n_views);
  render_bwd_body<KB>(item, W, H, bv.items[y], bv.tile_depth[y], bv.ckpt[y], bv.ranges[y], bv.point_list[y], bv.splat[y], bv.bg[y],
                  bv.color[y], bv.depth_alpha[y], bv.final_T[y], bv.n_contrib[y], bv.dL_dcolor[y], bv.dL_dda[y],
                  bv.partials[y], bv.reach[y]);
}

// The stage timers (GSR_STAGE_RENDER_FWD / _BWD) bracket the compositing kernel alone (not the work-list kernel), so
// that bench.py's roofline entry and the rocprofv3 average of that kernel measure the same thing.
// Work lists of n views (same image size) in one launch.
int gsr_launch_work_order_fwd(int n, const GsrView* views, const GsrBinning* bs, const GsrImages* imgs, hipStream_t stream) {
  const uint32_t tiles = gsr_num_tiles(views[0].image_height, views[0].image_width);
  WorkFwdViews wv = WorkFwdViews{};
  for (int k = 0; k < n; ++k) {
    wv.ranges[k] = bs[k].ranges; wv.tile_depth[k] = imgs[k].tile_depth; wv.work[k] = bs[k].tile_work;
    wv.stats_host[k] = bs[k].stats_host;
  }
  hipLaunchKernelGGL(k_work_order_fwd, dim3((uint32_t)n), dim3(1024), 0, stream, tiles, wv);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}
int gsr_launch_work_order_bwd(int n, const GsrView* views, const GsrBinning* bs, const GsrImages* imgs, hipStream_t stream) {
  const uint32_t tiles = gsr_num_tiles(views[0].image_height, views[0].image_width);
  WorkBwdViews wv = WorkBwdViews{};
  for (int k = 0; k < n; ++k) {
    wv.tile_depth[k] = imgs[k].tile_depth; wv.items[k] = bs[k].tile_work + tiles; wv.items_cap[k] = bs[k].bwd_items_cap;
    wv.seg_len[k] = gsr_seg_len(bs[k]);
  }
  hipLaunchKernelGGL(k_work_order_bwd, dim3((uint32_t)n), dim3(1024), 0, stream, tiles, wv);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

// score_mode 0: pixel counts (u32, left by K6 in the score buffer) -> opacity x count, in place
struct ScoreViews {
  float* score[GSR_MAX_BATCH_VIEWS];
  const float* splat[GSR_MAX_BATCH_VIEWS];
  const int32_t* radii[GSR_MAX_BATCH_VIEWS];
};
namespace {
__global__ void __launch_bounds__(256) k_score_finalize(const ScoreViews sv, const int32_t P) {
  const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
  if (i >= P) return;
  float* sc = sv.score[blockIdx.y];
  const uint32_t c = reinterpret_cast<const uint32_t*>(sc)[i];
  // (rows of culled Gaussians are never written: their count is 0 and their opacity is not read)
  sc[i] = (c && sv.radii[blockIdx.y][i] > 0) ? sv.splat[blockIdx.y][12 * (size_t)i + 5] * (float)c : 0.0f;
}
}  // namespace

// K6 of n views in one launch (their work lists must have been built; same image size, forward variant and score
// output for all of them -- the caller checks).
int gsr_launch_render_fwd_views(int n, const GsrView* views, const GsrGeom* geoms, const GsrBinning* bs, GsrImages* imgs,
                                hipStream_t stream, GsrProfile* prof) {
  const GsrView& v = views[0];
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  FwdViews fv = FwdViews{};
  for (int k = 0; k < n; ++k) {
    fv.work[k] = bs[k].tile_work; fv.ckpt[k] = imgs[k].ckpt; fv.ranges[k] = bs[k].ranges;
    fv.point_list[k] = bs[k].point_list; fv.splat[k] = reinterpret_cast<const float4*>(geoms[k].splat);
    fv.bg[k] = views[k].bg; fv.out_color[k] = imgs[k].color; fv.out_da[k] = imgs[k].depth_alpha;
    fv.final_T[k] = imgs[k].final_T; fv.n_contrib[k] = imgs[k].n_contrib; fv.tile_depth[k] = imgs[k].tile_depth;
    fv.score[k] = imgs[k].important_score;
  }
  const bool score = imgs[0].important_score != nullptr;
  const uint32_t ny = (uint32_t)n;
  GsrStageTimer timer(prof, stream, GSR_STAGE_RENDER_FWD);
  if (bs[0].fwd_mode == 1) {
    if (score) hipLaunchKernelGGL(k_render_fwd_tile<true>, dim3(tiles * ny), dim3(256), 0, stream, v.image_width, v.image_height, fv, v.score_mode, ny, tiles);
    else hipLaunchKernelGGL(k_render_fwd_tile<false>, dim3(tiles * ny), dim3(256), 0, stream, v.image_width, v.image_height, fv, 0, ny, tiles);
  } else {
#define GSR_LAUNCH_K6(SC, KB) hipLaunchKernelGGL((k_render_fwd<SC, KB>), dim3(tiles * 4 * ny), dim3(256), 0, stream, v.image_width, v.image_height, fv, (SC) ? v.score_mode : 0, ny, 0u)
    const uint32_t kb = gsr_seg_len(bs[0]);
    if (score) { if (kb == 64) GSR_LAUNCH_K6(true, 64); else if (kb == 128) GSR_LAUNCH_K6(true, 128); else GSR_LAUNCH_K6(true, 256); }
    else { if (kb == 64) GSR_LAUNCH_K6(false, 64); else if (kb == 128) GSR_LAUNCH_K6(false, 128); else GSR_LAUNCH_K6(false, 256); }
#undef GSR_LAUNCH_K6
  }
  GSR_HIP(hipGetLastError());
  timer.stop();
  if (score && v.score_mode == 0 && v.P > 0) {
    ScoreViews sv = ScoreViews{};
    for (int k = 0; k < n; ++k) { sv.score[k] = imgs[k].important_score; sv.splat[k] = geoms[k].splat; sv.radii[k] = geoms[k].radii; }
    hipLaunchKernelGGL(k_score_finalize, dim3(((uint32_t)v.P + 255u) / 256u, ny), dim3(256), 0, stream, sv, v.P);
    GSR_HIP(hipGetLastError());
  }
  return GSR_OK;
}
int gsr_launch_render_fwd(const GsrView& v, const GsrGeom& geom, const GsrBinning& b, GsrImages& img,
                          hipStream_t stream, GsrProfile* prof) {
  return gsr_launch_render_fwd_views(1, &v, &geom, &b, &img, stream, prof);
}

// K7 of n views in one launch (work lists built; same image size and the same seg_len: the caller checks).
int gsr_launch_render_bwd_views(int n, const GsrView* views, const GsrGeom* geoms, const GsrBinning* bs,
                                const GsrImages* imgs, const GsrImageGrads* igs, GsrGrads* outs, hipStream_t stream,
                                GsrProfile* prof) {
  const GsrView& v = views[0];
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  BwdViews bv = BwdViews{};
  uint32_t items_cap = 0;
  for (int k = 0; k < n; ++k) {
    bv.items[k] = bs[k].tile_work + tiles; bv.tile_depth[k] = imgs[k].tile_depth; bv.ckpt[k] = imgs[k].ckpt;
    bv.ranges[k] = bs[k].ranges; bv.point_list[k] = bs[k].point_list;
    bv.splat[k] = reinterpret_cast<const float4*>(geoms[k].splat); bv.bg[k] = views[k].bg; bv.color[k] = imgs[k].color;
    bv.depth_alpha[k] = imgs[k].depth_alpha; bv.final_T[k] = imgs[k].final_T; bv.n_contrib[k] = imgs[k].n_contrib;
    bv.dL_dcolor[k] = igs[k].dL_dcolor; bv.dL_dda[k] = igs[k].dL_ddepth_alpha; bv.partials[k] = outs[k].partials;
    bv.reach[k] = reinterpret_cast<unsigned long long*>(outs[k].reach);
    items_cap = bs[k].bwd_items_cap > items_cap ? bs[k].bwd_items_cap : items_cap;
  }
  GsrStageTimer timer(prof, stream, GSR_STAGE_RENDER_BWD);
  // (per_view: the same regime switch as the forward variant)
#define GSR_LAUNCH_K7(KB) hipLaunchKernelGGL(k_render_bwd<KB>, dim3(items_cap * (uint32_t)n), dim3(256), 0, stream, v.image_width, v.image_height, bv, (uint32_t)n, bs[0].fwd_mode == 1 ? items_cap : 0u)
  const uint32_t kb = gsr_seg_len(bs[0]);
  if (kb == 64) GSR_LAUNCH_K7(64); else if (kb == 128) GSR_LAUNCH_K7(128); else GSR_LAUNCH_K7(256);
#undef GSR_LAUNCH_K7
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}
int gsr_launch_render_bwd(const GsrView& v, const GsrGeom& geom, const GsrBinning& b, const GsrImages& img,
                          const GsrImageGrads& ig, GsrGrads& out, hipStream_t stream, GsrProfile* prof) {
  return gsr_launch_render_bwd_views(1, &v, &geom, &b, &img, &ig, &out, stream, prof);
}
