// binning.hip -- tile binning: depth order, (tile, depth)-ordered pair lists, tile ranges. gfx950, integer work.
//
// Required result (SEMANTICS.md): the list the rasterizer lineage obtains by emitting,
// per visible Gaussian and per overlapped 16x16 tile, key = tile_id << 32 | fp32 bits of view depth, value =
// Gaussian index, in Gaussian-index-major order, and STABLE-sorting the N pairs by the 64-bit key.
// That order is (tile, depth bits, Gaussian index). It is produced here without ever sorting 64-bit keys over N:
//   1. stable LSD radix sort of the P Gaussians by depth bits (culled ones keyed 0xFFFFFFFF go last)
//        -> order (depth bits, Gaussian index);
//   2. pairs are emitted in THAT order with key = tile id only;
//   3. stable LSD radix sort of the N pairs by tile id (2 passes of 8 bits up to 65536 tiles)
//        -> order (tile, depth bits, Gaussian index)  == the reference order, bit for bit.
// Traffic: 4 passes over P x 8 B + 2 passes over N x 8 B instead of 6 passes over N x 12 B.
// The sorted value list and the tile ranges are bit-exact against oracle/gsr_oracle.c (orc_bin_sort); the
// 64-bit keys can be reconstructed on request (GsrBinning.keys_sorted) for the parity tests.
//
// "Capacity mode": the pair count N is data dependent. Every N-sized kernel takes the true count from device
// memory and clamps it to the capacity of the caller's buffers, so the whole forward can be enqueued without a
// host round trip; the host checks N against the capacity afterwards (gsrast.h, gsr_forward_render).
#include "gsr_common.h"
#include "radix_sort.h"

namespace {

// ------------------------------------------------------------------------------------- depth-ordered counts
// Per-256 sums of tiles_touched taken in depth order (feeds the scan that yields N and the emission offsets).
__global__ void __launch_bounds__(256)
k_sorted_block_sums(const uint64_t* __restrict__ n_vis, const uint32_t* __restrict__ sorted_idx,
                    const uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wave_tiles[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s = (int64_t)blockIdx.x * 256 + tid;
  uint32_t c = (s < (int64_t)*n_vis) ? tiles_touched[sorted_idx[s]] : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
  if (lane == 0) wave_tiles[wave] = c;
  __syncthreads();
  if (tid == 0) block_sums[blockIdx.x] = (wave_tiles[0] + wave_tiles[1]) + (wave_tiles[2] + wave_tiles[3]);
}

// In-place exclusive scan of the per-256 sums; offsets[nb] = N (low 32 bits), *n_pairs = N (64-bit).
__global__ void __launch_bounds__(1024) k_scan_blocks(uint32_t* __restrict__ sums, uint32_t nb,
                                                      uint64_t* __restrict__ n_pairs) {
  __shared__ uint64_t wave_tot[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t idx = base + tid;
    const uint64_t x = idx < nb ? (uint64_t)sums[idx] : 0ull;
    uint64_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wave_tot[w];
    const uint64_t carry = carry_s;
    const uint64_t excl = carry + woff + inc - x;
    if (idx < nb) sums[idx] = (uint32_t)excl;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) {
    sums[nb] = (uint32_t)carry_s;
    *n_pairs = carry_s;
  }
}

// ------------------------------------------------------------------------------------------- pair emission
// Thread s handles the s-th Gaussian in depth order; pairs beyond `cap` are dropped (capacity mode).
__global__ void __launch_bounds__(256)
k_emit_pairs(const uint64_t* __restrict__ n_vis, const int W, const int H, const float* __restrict__ splat,
             const int32_t* __restrict__ radii,
             const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ sorted_idx,
             const uint32_t* __restrict__ block_offsets, const uint64_t cap, uint32_t* __restrict__ keys,
             uint32_t* __restrict__ vals, uint32_t* __restrict__ ranges, const uint32_t n_range_words) {
  __shared__ uint32_t wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s = (int64_t)blockIdx.x * 256 + tid;
  // tile ranges start out as (0,0): cleared here (grid-stride) instead of by a separate fill launch
  for (uint32_t w = (uint32_t)s; w < n_range_words; w += gridDim.x * 256u) ranges[w] = 0u;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const bool in = s < (int64_t)*n_vis;          // sorted_idx holds the visible Gaussians only, in depth order
  const uint32_t i = in ? sorted_idx[s] : 0u;
  const uint32_t cnt = in ? tiles_touched[i] : 0u;
  uint32_t inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t off = block_offsets[blockIdx.x] + inc - cnt;
  for (int w = 0; w < wave; ++w) off += wave_tot[w];

  int x0 = 0, y0 = 0, x1 = 0;
  if (cnt) {
    const float4 q0 = *reinterpret_cast<const float4*>(splat + 12 * (size_t)i);
    const float rf = (float)radii[i];
    x0 = min(gx, max(0, gsr_f2i_sat((q0.x - rf) * 0.0625f)));
    y0 = min(gy, max(0, gsr_f2i_sat((q0.y - rf) * 0.0625f)));
    x1 = min(gx, max(0, gsr_f2i_sat(((q0.x + rf) + 15.0f) * 0.0625f)));
  }
  const int rw = x1 - x0;
  constexpr uint32_t kCoop = 32;
  if (cnt && cnt <= kCoop) {   // small footprints: the owning lane writes its own pairs
    for (uint32_t k = 0; k < cnt; ++k) {
      const int ty = y0 + (int)(k / (uint32_t)rw), tx = x0 + (int)(k % (uint32_t)rw);
      if ((uint64_t)off + k < cap) {
        keys[off + k] = (uint32_t)(ty * gx + tx);
        vals[off + k] = i;
      }
    }
  }
  // large footprints: the whole wave writes one Gaussian's pairs together (coalesced, no long serial tail)
  unsigned long long big = __ballot(cnt > kCoop);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t c = (uint32_t)__shfl((int)cnt, src, 64);
    const uint32_t o = (uint32_t)__shfl((int)off, src, 64);
    const int sx0 = __shfl(x0, src, 64), sy0 = __shfl(y0, src, 64), srw = __shfl(rw, src, 64);
    const uint32_t sid = (uint32_t)__shfl((int)i, src, 64);
    for (uint32_t k = lane; k < c; k += 64) {
      const int ty = sy0 + (int)(k / (uint32_t)srw), tx = sx0 + (int)(k % (uint32_t)srw);
      if ((uint64_t)o + k < cap) {
        keys[o + k] = (uint32_t)(ty * gx + tx);
        vals[o + k] = sid;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- ranges
__global__ void __launch_bounds__(256)
k_tile_ranges(const uint32_t* __restrict__ tile_keys, const uint64_t* __restrict__ n_dev, uint64_t cap,
              uint32_t* __restrict__ ranges) {
  const uint64_t n = eff_count(n_dev, cap);
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t t = tile_keys[j];
  if (j == 0 || tile_keys[j - 1] != t) ranges[2 * t] = (uint32_t)j;
  if (j == n - 1 || tile_keys[j + 1] != t) ranges[2 * t + 1] = (uint32_t)(j + 1);
}

// debug / parity: the 64-bit keys of the reference formulation, rebuilt from the sorted lists
__global__ void __launch_bounds__(256)
k_rebuild_keys(const uint32_t* __restrict__ tile_keys, const uint32_t* __restrict__ point_list,
               const float* __restrict__ splat, const uint64_t* __restrict__ n_dev, uint64_t cap,
               uint64_t* __restrict__ keys64) {
  const uint64_t n = eff_count(n_dev, cap);
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t dbits = __float_as_uint(splat[12 * (size_t)point_list[j] + 6]);
  keys64[j] = ((uint64_t)tile_keys[j] << 32) | dbits;
}

}  // namespace

extern "C" uint32_t gsr_num_tiles(int32_t H, int32_t W) {
  return (uint32_t)(((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE));
}
extern "C" uint32_t gsr_num_blocks(int32_t P) { return (uint32_t)((P + 255) / 256); }

// Scratch of the projection stage (depth sort): keys x2, values x2 (one of them becomes sorted_idx), histograms.
extern "C" size_t gsr_project_scratch_bytes(int32_t P) {
  const uint64_t m = P > 0 ? (uint64_t)P : 1;
  return 4 * align256(m * 4) + align256((size_t)kRadix * sort_blocks(m, kItemsSmall) * 4) + align256(kRadix * 4) + 1024;
}

// Scratch of the binning stage: tile keys x2, one value ping buffer, histograms.
extern "C" size_t gsr_sort_scratch_bytes(uint64_t n, uint32_t n_tiles) {
  (void)n_tiles;
  const uint64_t m = n ? n : 1;
  return 3 * align256(m * 4) + align256((size_t)kRadix * sort_blocks(m, kItemsLarge) * 4) + align256(kRadix * 4) + 1024;
}

struct ProjectScratch {
  uint32_t *k0, *k1, *v0, *v1, *hist, *totals;
};
static ProjectScratch carve_project(void* scratch, int32_t P) {
  const uint64_t m = P > 0 ? (uint64_t)P : 1;
  char* b = (char*)scratch;
  ProjectScratch s;
  s.k0 = (uint32_t*)b; b += align256(m * 4);
  s.k1 = (uint32_t*)b; b += align256(m * 4);
  s.v0 = (uint32_t*)b; b += align256(m * 4);
  s.v1 = (uint32_t*)b; b += align256(m * 4);
  s.hist = (uint32_t*)b; b += align256((size_t)kRadix * sort_blocks(m, kItemsSmall) * 4);
  s.totals = (uint32_t*)b;
  return s;
}

uint32_t* gsr_depth_keys(const GsrGeom& geom, int32_t P) { return carve_project(geom.scratch, P).k0; }

// After K1 (which wrote the depth keys into scratch.k0): depth sort, depth-ordered block sums, scan -> N.
int gsr_launch_depth_order(GsrGeom& geom, int32_t P, uint64_t* n_pairs_dev, uint32_t depth_skip_mask, hipStream_t stream,
                           GsrProfile* prof) {
  if (geom.scratch_bytes < gsr_project_scratch_bytes(P) || !geom.scratch) return GSR_ESCRATCH;
  ProjectScratch s = carve_project(geom.scratch, P);
  uint64_t* n_vis_dev = n_pairs_dev + 1;     // number of visible Gaussians, next to the pair count (block_offsets tail)
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SORT);
    (void)depth_skip_mask;
    const int where = radix_sort_u32<kItemsSmall>(s.k0, s.v0, s.k1, s.v1, nullptr, (uint64_t)P, 32, true, n_vis_dev,
                                                   s.hist, s.totals, stream);
    geom.sorted_idx = where ? s.v1 : s.v0;
    GSR_HIP(hipGetLastError());
  }
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SCAN);
    const uint32_t nb = gsr_num_blocks(P);
    hipLaunchKernelGGL(k_sorted_block_sums, dim3(nb), dim3(256), 0, stream, n_vis_dev, geom.sorted_idx, geom.tiles_touched,
                       geom.block_offsets);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, geom.block_offsets, nb, n_pairs_dev);
    GSR_HIP(hipGetLastError());
  }
  return GSR_OK;
}

// Emits, tile-sorts and ranges. `cap` = pairs the buffers hold; n_dev (may be NULL = exactly cap pairs) is the
// true count on the device. On return binning.point_list holds the sorted values.
int gsr_launch_binning(const GsrView& v, const GsrGeom& geom, uint64_t cap, const uint64_t* n_dev,
                       const uint64_t* n_dev_vis, GsrBinning& b, hipStream_t stream, GsrProfile* prof) {
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  if (cap == 0 || v.P == 0) {
    GSR_HIP(hipMemsetAsync(b.ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t), stream));
    return GSR_OK;
  }
  if (b.scratch_bytes < gsr_sort_scratch_bytes(cap, tiles) || !b.scratch) return GSR_ESCRATCH;
  if (!geom.sorted_idx) return GSR_EINVAL;
  char* base = (char*)b.scratch;
  uint32_t* keys_a = (uint32_t*)base; base += align256(cap * 4);
  uint32_t* keys_b = (uint32_t*)base; base += align256(cap * 4);
  uint32_t* vals_t = (uint32_t*)base; base += align256(cap * 4);
  uint32_t* hist = (uint32_t*)base; base += align256((size_t)kRadix * sort_blocks(cap, kItemsLarge) * 4);
  uint32_t* totals = (uint32_t*)base;

  int tile_bits = 0;
  while ((1u << tile_bits) < tiles) ++tile_bits;
  if (tile_bits == 0) tile_bits = 1;
  const int passes = (tile_bits + kRadixBits - 1) / kRadixBits;
  // choose the first value buffer so that the last pass lands in point_list
  uint32_t* va = (passes % 2 == 0) ? b.point_list : vals_t;
  uint32_t* vb = (passes % 2 == 0) ? vals_t : b.point_list;
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_DUPLICATE);
    hipLaunchKernelGGL(k_emit_pairs, dim3(gsr_num_blocks(v.P)), dim3(256), 0, stream, n_dev_vis, v.image_width,
                       v.image_height, geom.splat, geom.radii, geom.tiles_touched, geom.sorted_idx, geom.block_offsets,
                       cap, keys_a, va, b.ranges, tiles * 2);
    GSR_HIP(hipGetLastError());
  }
  uint32_t* sorted_keys;
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SORT);
    const int where = radix_sort_u32<kItemsLarge>(keys_a, va, keys_b, vb, n_dev, cap, tile_bits, false, 0u, hist, totals,
                                                   stream);
    sorted_keys = where ? keys_b : keys_a;
    GSR_HIP(hipGetLastError());
  }
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_RANGES);
    const uint32_t nb = (uint32_t)((cap + 255) / 256);
    hipLaunchKernelGGL(k_tile_ranges, dim3(nb), dim3(256), 0, stream, sorted_keys, n_dev, cap, b.ranges);
    if (b.keys_sorted)
      hipLaunchKernelGGL(k_rebuild_keys, dim3(nb), dim3(256), 0, stream, sorted_keys, b.point_list, geom.splat, n_dev,
                         cap, b.keys_sorted);
    GSR_HIP(hipGetLastError());
  }
  return GSR_OK;
}
