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

namespace {

constexpr int kSortThreads = 256;
constexpr int kItemsLarge = 16;   // keys per thread for the N-sized tile sort (4096 keys / workgroup)
constexpr int kItemsSmall = 4;    // ... for the P-sized depth sort: 4x more workgroups, 4x shorter rank chains
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

__device__ __forceinline__ uint64_t eff_count(const uint64_t* n_dev, uint64_t cap) {
  if (!n_dev) return cap;
  const uint64_t n = *n_dev;
  return n < cap ? n : cap;
}

// ---------------------------------------------------------------------------------------------- radix sort
// One LSD pass = histogram -> per-digit exclusive scan over workgroups -> stable scatter.
// Element order inside a workgroup: e = blk*4096 + wave*1024 + item*64 + lane.
template <int ITEMS>
__device__ __forceinline__ uint64_t sort_index(uint32_t blk, int wave, int item, int lane) {
  return (uint64_t)blk * (kSortThreads * ITEMS) + (uint64_t)(wave * (64 * ITEMS) + item * 64 + lane);
}

// lanes of the wave holding the same 8-bit digit as this lane (among `valid` lanes)
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < kRadixBits; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

template <int ITEMS>
__global__ void __launch_bounds__(kSortThreads)
k_radix_hist(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ n_dev, uint64_t cap, int shift,
             uint32_t nblk, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[kRadix];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t n = eff_count(n_dev, cap);
  h[tid] = 0;
  __syncthreads();
  if ((uint64_t)blockIdx.x * (kSortThreads * ITEMS) < n) {
#pragma unroll 4
    for (int it = 0; it < ITEMS; ++it) {
      const uint64_t e = sort_index<ITEMS>(blockIdx.x, wave, it, lane);
      const bool valid = e < n;
      const uint32_t d = valid ? ((keys[e] >> shift) & (kRadix - 1)) : 0u;
      const unsigned long long m = match_digit(d, valid);
      // one LDS atomic per distinct digit per wave (digits of tile ids / exponents are heavily clustered)
      if (valid && lane == __ffsll((long long)m) - 1) atomicAdd(&h[d], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  hist[(uint64_t)tid * nblk + blockIdx.x] = h[tid];
}

// Workgroup d scans row d of hist[256][nblk] in place (exclusive) and writes the row total to totals[d].
__global__ void __launch_bounds__(256) k_radix_scan(uint32_t* __restrict__ hist, uint32_t nblk,
                                                    uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* row = hist + (uint64_t)blockIdx.x * nblk;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  constexpr uint32_t kPer = 4;
  for (uint32_t base = 0; base < nblk; base += 256 * kPer) {
    uint32_t x[kPer];
    uint32_t s = 0;
    const uint32_t first = base + tid * kPer;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      x[k] = (first + k < nblk) ? row[first + k] : 0u;
      s += x[k];
    }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wave_tot[w];
    const uint32_t carry = carry_s;
    uint32_t run = carry + woff + inc - s;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      if (first + k < nblk) row[first + k] = run;
      run += x[k];
    }
    __syncthreads();
    if (tid == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) totals[blockIdx.x] = carry_s;
}

// IOTA: values are the element indices themselves (first pass of the depth sort), vals_in unused.
template <bool IOTA, int ITEMS>
__global__ void __launch_bounds__(kSortThreads)
k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const uint64_t* __restrict__ n_dev,
                uint64_t cap, int shift, uint32_t nblk, const uint32_t* __restrict__ hist,
                const uint32_t* __restrict__ totals) {
  __shared__ uint32_t wh[4][kRadix];   // running per-wave digit counters, then per-wave global bases
  __shared__ uint32_t dbase[kRadix];   // exclusive scan of the 256 digit totals
  __shared__ uint32_t wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t n = eff_count(n_dev, cap);
  if ((uint64_t)blockIdx.x * (kSortThreads * ITEMS) >= n) return;
#pragma unroll
  for (int w = 0; w < 4; ++w) wh[w][tid] = 0;
  {
    const uint32_t x = totals[tid];
    uint32_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    dbase[tid] = woff + inc - x;
  }
  __syncthreads();
  volatile uint32_t* mywh = wh[wave];
  uint32_t key[ITEMS];
  uint32_t val[ITEMS];
  uint32_t rank[ITEMS];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const uint64_t e = sort_index<ITEMS>(blockIdx.x, wave, it, lane);
    const bool valid = e < n;
    key[it] = valid ? keys_in[e] : 0xFFFFFFFFu;
    val[it] = IOTA ? (uint32_t)e : (valid ? vals_in[e] : 0u);
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const uint64_t e = sort_index<ITEMS>(blockIdx.x, wave, it, lane);
    const bool valid = e < n;
    const uint32_t d = (key[it] >> shift) & (kRadix - 1);
    const unsigned long long m = match_digit(d, valid);
    const int leader = __ffsll((long long)m) - 1;
    uint32_t old = 0;
    if (valid && lane == leader) {
      old = mywh[d];
      mywh[d] = old + (uint32_t)__popcll(m);
    }
    old = (uint32_t)__shfl((int)old, valid ? leader : lane, 64);
    rank[it] = old + (uint32_t)__popcll(m & lt);
  }
  __syncthreads();
  {
    uint32_t run = dbase[tid] + hist[(uint64_t)tid * nblk + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = wh[w][tid];
      wh[w][tid] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const uint64_t e = sort_index<ITEMS>(blockIdx.x, wave, it, lane);
    if (e < n) {
      const uint32_t d = (key[it] >> shift) & (kRadix - 1);
      const uint32_t pos = wh[wave][d] + rank[it];
      keys_out[pos] = key[it];
      vals_out[pos] = val[it];
    }
  }
}

// ------------------------------------------------------------------------------------- depth-ordered counts
// Per-256 sums of tiles_touched taken in depth order (feeds the scan that yields N and the emission offsets).
__global__ void __launch_bounds__(256)
k_sorted_block_sums(const int P, const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ tiles_touched,
                    uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wave_tiles[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s = (int64_t)blockIdx.x * 256 + tid;
  uint32_t c = (s < P) ? tiles_touched[sorted_idx[s]] : 0u;
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
k_emit_pairs(const int P, const int W, const int H, const float* __restrict__ splat, const int32_t* __restrict__ radii,
             const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ sorted_idx,
             const uint32_t* __restrict__ block_offsets, const uint64_t cap, uint32_t* __restrict__ keys,
             uint32_t* __restrict__ vals, uint32_t* __restrict__ ranges, const uint32_t n_range_words) {
  __shared__ uint32_t wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s = (int64_t)blockIdx.x * 256 + tid;
  // tile ranges start out as (0,0): cleared here (grid-stride) instead of by a separate fill launch
  for (uint32_t w = (uint32_t)s; w < n_range_words; w += gridDim.x * 256u) ranges[w] = 0u;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const uint32_t i = (s < P) ? sorted_idx[s] : 0u;
  const uint32_t cnt = (s < P) ? tiles_touched[i] : 0u;
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

__host__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

uint32_t sort_blocks(uint64_t n, int items) {
  const uint64_t t = (uint64_t)kSortThreads * items;
  return (uint32_t)((n + t - 1) / t);
}

// One full LSD sort of (u32 key, u32 value) over the key bits [0, bits). Buffers ping-pong between (k0,v0) and
// (k1,v1); returns 0 if the result is in (k0,v0), 1 if in (k1,v1). iota: values of the first executed pass are
// the element indices. skip_mask: bit p set = digit p is identical in all keys, the pass is the identity: skipped.
template <int ITEMS>
int radix_sort_u32(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, const uint64_t* n_dev, uint64_t cap,
                   int bits, bool iota, uint32_t skip_mask, uint32_t* hist, uint32_t* totals, hipStream_t stream) {
  const uint32_t nblk = sort_blocks(cap, ITEMS);
  const int passes = (bits + kRadixBits - 1) / kRadixBits;
  uint32_t *ka = k0, *va = v0, *kb = k1, *vb = v1;
  int flips = 0;
  bool first = true;
  for (int p = 0; p < passes; ++p) {
    if ((skip_mask >> p) & 1u) continue;
    const int shift = p * kRadixBits;
    hipLaunchKernelGGL(k_radix_hist<ITEMS>, dim3(nblk), dim3(kSortThreads), 0, stream, ka, n_dev, cap, shift, nblk, hist);
    hipLaunchKernelGGL(k_radix_scan, dim3(kRadix), dim3(256), 0, stream, hist, nblk, totals);
    if (iota && first)
      hipLaunchKernelGGL((k_radix_scatter<true, ITEMS>), dim3(nblk), dim3(kSortThreads), 0, stream, ka, va, kb, vb,
                         n_dev, cap, shift, nblk, hist, totals);
    else
      hipLaunchKernelGGL((k_radix_scatter<false, ITEMS>), dim3(nblk), dim3(kSortThreads), 0, stream, ka, va, kb, vb,
                         n_dev, cap, shift, nblk, hist, totals);
    first = false;
    uint32_t* t = ka; ka = kb; kb = t;
    t = va; va = vb; vb = t;
    ++flips;
  }
  return flips & 1;
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
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SORT);
    const int where = radix_sort_u32<kItemsSmall>(s.k0, s.v0, s.k1, s.v1, nullptr, (uint64_t)P, 32, true, depth_skip_mask,
                                                   s.hist, s.totals, stream);
    geom.sorted_idx = where ? s.v1 : s.v0;
    GSR_HIP(hipGetLastError());
  }
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SCAN);
    const uint32_t nb = gsr_num_blocks(P);
    hipLaunchKernelGGL(k_sorted_block_sums, dim3(nb), dim3(256), 0, stream, P, geom.sorted_idx, geom.tiles_touched,
                       geom.block_offsets);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, geom.block_offsets, nb, n_pairs_dev);
    GSR_HIP(hipGetLastError());
  }
  return GSR_OK;
}

// Emits, tile-sorts and ranges. `cap` = pairs the buffers hold; n_dev (may be NULL = exactly cap pairs) is the
// true count on the device. On return binning.point_list holds the sorted values.
int gsr_launch_binning(const GsrView& v, const GsrGeom& geom, uint64_t cap, const uint64_t* n_dev, GsrBinning& b,
                       hipStream_t stream, GsrProfile* prof) {
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
    hipLaunchKernelGGL(k_emit_pairs, dim3(gsr_num_blocks(v.P)), dim3(256), 0, stream, v.P, v.image_width,
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
