// radix_sort.h -- stable LSD radix sort of (u32 key, u32 value) pairs on gfx950, shared by the tile binning
// (binning.hip) and the k-nearest-neighbour grid (knn.hip). See binning.hip for the design notes.
#pragma once
#include "gsr_common.h"

namespace {

constexpr int kSortThreads = 256;
constexpr int kItemsLarge = 16;   // keys per thread for the N-sized tile sort (4096 keys / workgroup)
constexpr int kItemsSmall = 4;    // ... for the P-sized depth sort: 4x more workgroups, 4x shorter rank chains
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

// Batched launches (several views at once): blockIdx.y selects the view, whose buffers sit `bstride` bytes further.
template <typename T>
__device__ __forceinline__ T* batch_ptr(T* p, size_t bstride) {
  return p ? reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (size_t)blockIdx.y * bstride) : p;
}

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

// DROP: elements whose key is the sentinel 0xFFFFFFFF are not counted (and not written by the scatter): the first
// pass of the depth sort compacts the culled Gaussians away for free, later passes run on the survivors only.
template <int ITEMS, bool DROP>
__global__ void __launch_bounds__(kSortThreads)
k_radix_hist(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ n_dev, uint64_t cap, int shift,
             uint32_t nblk, uint32_t* __restrict__ hist, size_t bstride) {
  keys = batch_ptr(keys, bstride); n_dev = batch_ptr(n_dev, bstride); hist = batch_ptr(hist, bstride);
  __shared__ uint32_t h[kRadix];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t n = eff_count(n_dev, cap);
  h[tid] = 0;
  __syncthreads();
  if ((uint64_t)blockIdx.x * (kSortThreads * ITEMS) < n) {
#pragma unroll 4
    for (int it = 0; it < ITEMS; ++it) {
      const uint64_t e = sort_index<ITEMS>(blockIdx.x, wave, it, lane);
      bool valid = e < n;
      const uint32_t kk = valid ? keys[e] : 0u;
      if (DROP) valid = valid && (kk != 0xFFFFFFFFu);
      const uint32_t d = (kk >> shift) & (kRadix - 1);
      const unsigned long long m = match_digit(d, valid);
      // one LDS atomic per distinct digit per wave (digits of tile ids / exponents are heavily clustered)
      if (valid && lane == __ffsll((long long)m) - 1) atomicAdd(&h[d], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  hist[(uint64_t)tid * nblk + blockIdx.x] = h[tid];
}

// Workgroup d scans row d of hist[256][nblk] in place (exclusive) and writes the row total to totals[d].
// n_items (device, may be NULL): only the first ceil(*n_items / per_block) entries of a row are in use.
__global__ void __launch_bounds__(256) k_radix_scan(uint32_t* __restrict__ hist, uint32_t nblk_stride,
                                                    uint32_t* __restrict__ totals,
                                                    const uint64_t* __restrict__ n_items, uint32_t per_block,
                                                    size_t bstride) {
  hist = batch_ptr(hist, bstride); totals = batch_ptr(totals, bstride); n_items = batch_ptr(n_items, bstride);
  uint32_t nblk = nblk_stride;
  if (n_items) {
    const uint64_t used = (*n_items + per_block - 1) / per_block;
    if (used < nblk) nblk = (uint32_t)used;
  }
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* row = hist + (uint64_t)blockIdx.x * nblk_stride;
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
template <bool IOTA, int ITEMS, bool DROP>
__global__ void __launch_bounds__(kSortThreads)
k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const uint64_t* __restrict__ n_dev,
                uint64_t cap, int shift, uint32_t nblk, const uint32_t* __restrict__ hist,
                const uint32_t* __restrict__ totals, uint64_t* __restrict__ n_out, size_t bstride) {
  keys_in = batch_ptr(keys_in, bstride); vals_in = batch_ptr(vals_in, bstride);
  keys_out = batch_ptr(keys_out, bstride); vals_out = batch_ptr(vals_out, bstride);
  n_dev = batch_ptr(n_dev, bstride); hist = batch_ptr(hist, bstride); totals = batch_ptr(totals, bstride);
  n_out = batch_ptr(n_out, bstride);
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
    if (DROP && n_out && blockIdx.x == 0 && tid == kSortThreads - 1) *n_out = (uint64_t)(woff + inc);   // survivors
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
  const auto is_valid = [&](int it) {
    const uint64_t e = sort_index<ITEMS>(blockIdx.x, wave, it, lane);
    return (e < n) && (!DROP || key[it] != 0xFFFFFFFFu);
  };
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const bool valid = is_valid(it);
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
    if (is_valid(it)) {
      const uint32_t d = (key[it] >> shift) & (kRadix - 1);
      const uint32_t pos = wh[wave][d] + rank[it];
      keys_out[pos] = key[it];
      vals_out[pos] = val[it];
    }
  }
}

__host__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

uint32_t sort_blocks(uint64_t n, int items) {
  const uint64_t t = (uint64_t)kSortThreads * items;
  return (uint32_t)((n + t - 1) / t);
}

// One full LSD sort of (u32 key, u32 value) over the key bits [0, bits). Buffers ping-pong between (k0,v0) and
// (k1,v1); returns 0 if the result is in (k0,v0), 1 if in (k1,v1). iota: values of the first pass are the element
// indices. n_compact (device, may be NULL; only with iota): the first pass drops the elements keyed 0xFFFFFFFF and
// stores the number of survivors there; the remaining passes (and the caller) work on that many elements.
template <int ITEMS>
int radix_sort_u32(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, const uint64_t* n_dev, uint64_t cap,
                   int bits, bool iota, uint64_t* n_compact, uint32_t* hist, uint32_t* totals, hipStream_t stream,
                   int batch = 1, size_t bstride = 0) {
  const uint32_t nblk = sort_blocks(cap, ITEMS);
  const int passes = (bits + kRadixBits - 1) / kRadixBits;
  const dim3 grid(nblk, (uint32_t)batch), grid_scan(kRadix, (uint32_t)batch);
  uint32_t *ka = k0, *va = v0, *kb = k1, *vb = v1;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * kRadixBits;
    if (iota && p == 0 && n_compact) {
      hipLaunchKernelGGL((k_radix_hist<ITEMS, true>), grid, dim3(kSortThreads), 0, stream, ka, n_dev, cap, shift, nblk,
                         hist, bstride);
      hipLaunchKernelGGL(k_radix_scan, grid_scan, dim3(256), 0, stream, hist, nblk, totals, (const uint64_t*)nullptr, 1u,
                         bstride);
      hipLaunchKernelGGL((k_radix_scatter<true, ITEMS, true>), grid, dim3(kSortThreads), 0, stream, ka, va, kb, vb, n_dev,
                         cap, shift, nblk, hist, totals, n_compact, bstride);
      n_dev = n_compact;
    } else {
      hipLaunchKernelGGL((k_radix_hist<ITEMS, false>), grid, dim3(kSortThreads), 0, stream, ka, n_dev, cap, shift, nblk,
                         hist, bstride);
      hipLaunchKernelGGL(k_radix_scan, grid_scan, dim3(256), 0, stream, hist, nblk, totals, (const uint64_t*)nullptr, 1u,
                         bstride);
      if (iota && p == 0)
        hipLaunchKernelGGL((k_radix_scatter<true, ITEMS, false>), grid, dim3(kSortThreads), 0, stream, ka, va, kb, vb,
                           n_dev, cap, shift, nblk, hist, totals, (uint64_t*)nullptr, bstride);
      else
        hipLaunchKernelGGL((k_radix_scatter<false, ITEMS, false>), grid, dim3(kSortThreads), 0, stream, ka, va, kb, vb,
                           n_dev, cap, shift, nblk, hist, totals, (uint64_t*)nullptr, bstride);
    }
    uint32_t* t = ka; ka = kb; kb = t;
    t = va; va = vb; vb = t;
  }
  return passes & 1;
}

}  // namespace
