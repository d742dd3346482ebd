// binning.hip -- K2 scan, K3 (tile|depth) key emission, K4 stable LSD radix sort, K5 tile ranges. gfx950.
//
// Integer / byte work, HBM-bound. Semantics (SEMANTICS.md): every visible Gaussian
// emits one (key,value) per overlapped 16x16 tile, key = tile_id << 32 | fp32 bits of view depth, value = Gaussian
// index, in Gaussian-index-major / row-major-tile order; a STABLE sort by key gives each tile a front-to-back
// list with ties resolved by emission order. The sorted value list and the per-tile ranges are bit-exact
// against oracle/gsr_oracle.c (orc_bin_sort).
#include "gsr_common.h"

namespace {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 16;                       // keys per thread
constexpr int kSortTile = kSortThreads * kSortItems; // 4096 keys per workgroup
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

// ----------------------------------------------------------------------------------------------------- K2
// In-place exclusive scan of the per-256-Gaussian tile counts; offsets[nb] = N (low 32 bits), *n_pairs = N.
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

// ----------------------------------------------------------------------------------------------------- K3
__global__ void __launch_bounds__(256)
k_duplicate(const int P, const int W, const int H, const float* __restrict__ splat,
            const int32_t* __restrict__ radii, const uint32_t* __restrict__ tiles_touched,
            const uint32_t* __restrict__ block_offsets, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  __shared__ uint32_t wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const uint32_t cnt = (i < P) ? tiles_touched[i] : 0u;
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
  uint32_t dbits = 0;
  if (cnt) {
    const float4 q0 = *reinterpret_cast<const float4*>(splat + 12 * i);
    const float4 q1 = *reinterpret_cast<const float4*>(splat + 12 * i + 4);
    const float rf = (float)radii[i];
    x0 = min(gx, max(0, gsr_f2i_sat((q0.x - rf) * 0.0625f)));
    y0 = min(gy, max(0, gsr_f2i_sat((q0.y - rf) * 0.0625f)));
    x1 = min(gx, max(0, gsr_f2i_sat(((q0.x + rf) + 15.0f) * 0.0625f)));
    dbits = __float_as_uint(q1.z);
  }
  const int rw = x1 - x0;
  // small footprints: the owning lane writes its own pairs
  constexpr uint32_t kCoop = 32;
  if (cnt && cnt <= kCoop) {
    for (uint32_t k = 0; k < cnt; ++k) {
      const int ty = y0 + (int)(k / (uint32_t)rw), tx = x0 + (int)(k % (uint32_t)rw);
      keys[off + k] = ((uint64_t)(uint32_t)(ty * gx + tx) << 32) | dbits;
      vals[off + k] = (uint32_t)i;
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
    const uint32_t sd = (uint32_t)__shfl((int)dbits, src, 64);
    const uint32_t sid = (uint32_t)(blockIdx.x * 256 + wave * 64 + src);
    for (uint32_t k = lane; k < c; k += 64) {
      const int ty = sy0 + (int)(k / (uint32_t)srw), tx = sx0 + (int)(k % (uint32_t)srw);
      keys[o + k] = ((uint64_t)(uint32_t)(ty * gx + tx) << 32) | sd;
      vals[o + k] = sid;
    }
  }
}

// ----------------------------------------------------------------------------------------------------- K4
// One LSD pass = histogram -> exclusive scan (digit-major over workgroups) -> stable scatter.
// Element order inside a workgroup: e = blk*4096 + wave*1024 + item*64 + lane.
__device__ __forceinline__ uint64_t sort_index(uint32_t blk, int wave, int item, int lane) {
  return (uint64_t)blk * kSortTile + (uint64_t)(wave * (64 * kSortItems) + item * 64 + lane);
}

__global__ void __launch_bounds__(kSortThreads)
k_radix_hist(const uint64_t* __restrict__ keys, uint64_t n, int shift, uint32_t nblk, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[kRadix];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  h[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const uint64_t e = sort_index(blockIdx.x, wave, it, lane);
    if (e < n) atomicAdd(&h[(uint32_t)(keys[e] >> shift) & (kRadix - 1)], 1u);
  }
  __syncthreads();
  hist[(uint64_t)tid * nblk + blockIdx.x] = h[tid];
}

// Per-digit exclusive scan: workgroup d scans row d of hist[256][nblk] in place and writes the row total to
// totals[d]. (The 256 totals are scanned by every scatter workgroup on the fly.)
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

__global__ void __launch_bounds__(kSortThreads)
k_radix_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint64_t n, int shift, uint32_t nblk,
                const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals) {
  __shared__ uint32_t wh[4][kRadix];   // running per-wave digit counters, then per-wave global bases
  __shared__ uint32_t dbase[kRadix];   // exclusive scan of the 256 digit totals
  __shared__ uint32_t wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  uint64_t key[kSortItems];
  uint32_t val[kSortItems];
  uint32_t rank[kSortItems];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const uint64_t e = sort_index(blockIdx.x, wave, it, lane);
    const bool valid = e < n;
    key[it] = valid ? keys_in[e] : ~0ull;
    val[it] = valid ? vals_in[e] : 0u;
  }
#pragma unroll
  for (int it = 0; it < kSortItems; ++it) {
    const uint64_t e = sort_index(blockIdx.x, wave, it, lane);
    const bool valid = e < n;
    const uint32_t d = (uint32_t)(key[it] >> shift) & (kRadix - 1);
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kRadixBits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    // m = valid lanes of this wave holding the same digit (for invalid lanes m is unused)
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
  for (int it = 0; it < kSortItems; ++it) {
    const uint64_t e = sort_index(blockIdx.x, wave, it, lane);
    if (e < n) {
      const uint32_t d = (uint32_t)(key[it] >> shift) & (kRadix - 1);
      const uint32_t pos = wh[wave][d] + rank[it];
      keys_out[pos] = key[it];
      vals_out[pos] = val[it];
    }
  }
}

// ----------------------------------------------------------------------------------------------------- K5
__global__ void __launch_bounds__(256)
k_tile_ranges(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ ranges) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t t = (uint32_t)(keys[j] >> 32);
  if (j == 0 || (uint32_t)(keys[j - 1] >> 32) != t) ranges[2 * t] = (uint32_t)j;
  if (j == n - 1 || (uint32_t)(keys[j + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(j + 1);
}

__host__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" uint32_t gsr_num_tiles(int32_t H, int32_t W) {
  return (uint32_t)(((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE));
}
extern "C" uint32_t gsr_num_blocks(int32_t P) { return (uint32_t)((P + 255) / 256); }

static uint32_t sort_blocks(uint64_t n) { return (uint32_t)((n + kSortTile - 1) / kSortTile); }

extern "C" size_t gsr_sort_scratch_bytes(uint64_t n, uint32_t n_tiles) {
  (void)n_tiles;
  const uint64_t m = n ? n : 1;
  return 2 * align256(m * 8) + align256(m * 4) + align256((size_t)kRadix * sort_blocks(m) * 4) + align256(kRadix * 4) + 1024;
}

int gsr_sort_key_bits(uint32_t n_tiles) {
  int tb = 0;
  while ((1u << tb) < n_tiles) ++tb;
  return 32 + tb;
}

int gsr_launch_scan(GsrGeom& geom, int32_t P, uint64_t* n_pairs_dev, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, geom.block_offsets, gsr_num_blocks(P), n_pairs_dev);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

// Emits, sorts and ranges. On return binning.point_list holds the sorted values.
int gsr_launch_binning(const GsrView& v, const GsrGeom& geom, uint64_t n, GsrBinning& b, hipStream_t stream,
                       GsrProfile* prof) {
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  GSR_HIP(hipMemsetAsync(b.ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t), stream));
  if (n == 0) return GSR_OK;
  if (b.scratch_bytes < gsr_sort_scratch_bytes(n, tiles) || !b.scratch) return GSR_ESCRATCH;
  const uint32_t nblk_ = sort_blocks(n);
  char* base = (char*)b.scratch;
  uint64_t* keys_a = (uint64_t*)base; base += align256(n * 8);
  uint64_t* keys_b = (uint64_t*)base; base += align256(n * 8);
  uint32_t* vals_t = (uint32_t*)base; base += align256(n * 4);
  uint32_t* hist = (uint32_t*)base; base += align256((size_t)kRadix * nblk_ * 4);
  uint32_t* totals = (uint32_t*)base;

  const int bits = gsr_sort_key_bits(tiles);
  const int passes = (bits + kRadixBits - 1) / kRadixBits;
  // choose the first value buffer so that the last pass lands in point_list
  uint32_t* va = (passes % 2 == 0) ? b.point_list : vals_t;
  uint32_t* vb = (passes % 2 == 0) ? vals_t : b.point_list;
  uint64_t *ka = keys_a, *kb = keys_b;

  {
    GsrStageTimer t(prof, stream, GSR_STAGE_DUPLICATE);
    hipLaunchKernelGGL(k_duplicate, dim3(gsr_num_blocks(v.P)), dim3(256), 0, stream, v.P, v.image_width,
                       v.image_height, geom.splat, geom.radii, geom.tiles_touched, geom.block_offsets, ka, va);
    GSR_HIP(hipGetLastError());
  }
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SORT);
    const uint32_t nblk = sort_blocks(n);
    for (int p = 0; p < passes; ++p) {
      const int shift = p * kRadixBits;
      hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(kSortThreads), 0, stream, ka, n, shift, nblk, hist);
      hipLaunchKernelGGL(k_radix_scan, dim3(kRadix), dim3(256), 0, stream, hist, nblk, totals);
      hipLaunchKernelGGL(k_radix_scatter, dim3(nblk), dim3(kSortThreads), 0, stream, ka, va, kb, vb, n, shift, nblk, hist,
                         totals);
      uint64_t* tk = ka; ka = kb; kb = tk;
      uint32_t* tv = va; va = vb; vb = tv;
    }
    GSR_HIP(hipGetLastError());
  }
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_RANGES);
    hipLaunchKernelGGL(k_tile_ranges, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, ka, n, b.ranges);
    GSR_HIP(hipGetLastError());
    if (b.keys_sorted) GSR_HIP(hipMemcpyAsync(b.keys_sorted, ka, n * 8, hipMemcpyDeviceToDevice, stream));
  }
  return GSR_OK;
}
