// depth_sort_distribution.h (round-4 experiment, NOT part of libgsrast; was csrc/depth_sort.h) -- order of the visible Gaussians of a view by (depth bits, Gaussian index), gfx950.
//
// Round 4. The LSD radix sort of the depth keys (radix_sort.h: one-sweep passes with decoupled look-back) moves 2 MB of
// keys per view and took 79 us for one view / 101 us for the four views of a step: four DEPENDENT passes, each waiting
// for look-back words to cross the XCDs (DESIGN.md "Binning"). Depth keys of one view are fp32 bit patterns inside one
// narrow interval [lo, hi] (an object 5.35 away: 2^21 consecutive integers), i.e. they are nearly uniformly distributed
// numbers -- a distribution sort needs ONE global partition step, everything after it is local to a workgroup:
//
//   k_ds_range    per-workgroup min / max of the visible keys (plain stores: no atomics, nothing to initialise), and
//                 clears the histograms;
//   k_ds_hist     fine histogram over NB = 2^lognb equal-width bins of [lo, hi] (bin = (key - lo) >> shift; every
//                 workgroup derives lo / shift from the min / max words itself) + coarse sums per 256 bins;
//   k_ds_plan     workgroup g: exclusive prefix of the coarse sums in front of group g (every workgroup scans the <= 1024
//                 coarse words itself: no chain), scan of its 256 fine bins -> the START POSITION of every bin, in place;
//                 and the bucket table: bucket j = the bins whose start lies in [j T, (j+1) T), T = 2048 -- whole bins, so
//                 a bucket is at most T - 1 + (largest bin) keys;
//   k_ds_scatter  key -> position = atomicAdd(start[bin], 1): (key, index) pairs grouped by bin, arrival order inside a bin;
//   k_ds_sort     one workgroup per bucket: LSD radix sort of its <= 4096 pairs INSIDE LDS on the bits of key - min(key of
//                 the bucket) that actually vary (two 8-bit passes at 500 k Gaussians), ties (equal depth bits) ordered by
//                 Gaussian index by counting inside the run of equal keys; writes sorted_idx.
// Five launches without any cross-workgroup wait; the result is the unique order by (key, index), i.e. exactly what the
// stable LSD sort of index-ordered keys produced (the lists stay bit-exact against the oracle).
// Buckets beyond the LDS capacity (one bin of the 2^lognb holds more than T keys: thousands of Gaussians within 2^-lognb of
// the depth range, e.g. the "all depths equal" test) are sorted by their workgroup alone, out of global memory, with a
// stable 7-pass LSD radix over (index, key) bytes -- slow (one workgroup), correct, and never taken by a real scene.
#pragma once
#include "radix_sort.h"

namespace {

constexpr uint32_t kDsT = 2048;        // bucket granularity (positions)
constexpr uint32_t kDsCap = 4096;      // pairs a workgroup sorts inside LDS
constexpr int kDsItems = 16;           // kDsCap / 256
constexpr uint32_t kDsChunk = 2048;    // keys per workgroup of the range / hist / scatter kernels
constexpr int kDsReplicas = 8;         // coarse sums kept per (workgroup index mod 8) ~ per XCD: 8x fewer same-line atomics

__host__ __device__ inline int ds_lognb(int64_t P) {
  int bits = 0;
  while (bits < 31 && (1ll << bits) <= P) ++bits;      // bits = floor(log2 P) + 1
  const int l = bits - 1;
  return l < 12 ? 12 : (l > 18 ? 18 : l);
}
__host__ inline uint32_t ds_chunks(int64_t P) { return (uint32_t)(((P > 0 ? P : 1) + kDsChunk - 1) / kDsChunk); }
__host__ inline uint32_t ds_buckets(int64_t P) { return (uint32_t)(((P > 0 ? P : 1) + kDsT - 1) / kDsT) + 1; }

// u32 words: [wgmm 2 per chunk][coarse kDsReplicas x NC][fine NB][bstart buckets + 1]
struct DsState {
  uint32_t *wgmm, *coarse, *fine, *bstart;
};
__host__ __device__ inline size_t ds_state_words(int64_t P) {
  const size_t nb = (size_t)1 << ds_lognb(P);
  const size_t chunks = (size_t)(((P > 0 ? P : 1) + kDsChunk - 1) / kDsChunk);
  const size_t buckets = (size_t)(((P > 0 ? P : 1) + kDsT - 1) / kDsT) + 1;
  return 2 * chunks + (size_t)kDsReplicas * (nb / 256) + nb + buckets + 1 + 64;
}
__host__ __device__ inline DsState ds_carve(uint32_t* base, int64_t P) {
  const size_t nb = (size_t)1 << ds_lognb(P);
  const size_t chunks = (size_t)(((P > 0 ? P : 1) + kDsChunk - 1) / kDsChunk);
  DsState s;
  s.wgmm = base;
  s.coarse = s.wgmm + ((2 * chunks + 15) & ~(size_t)15);
  s.fine = s.coarse + (size_t)kDsReplicas * (nb / 256);
  s.bstart = s.fine + nb;
  return s;
}

__device__ __forceinline__ uint32_t ds_wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
    v = v < t ? v : t;
  }
  return v;
}

// lo and shift of the view from the per-chunk min / max words (every workgroup for itself: 2 x chunks words, L2-resident)
struct DsRange {
  uint32_t lo, shift;
  bool any;
};
__device__ __forceinline__ DsRange ds_range(const uint32_t* __restrict__ wgmm, uint32_t chunks, int lognb, uint32_t* sh /*[10]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  for (uint32_t c = tid; c < chunks; c += 256) {
    const uint2 w = reinterpret_cast<const uint2*>(wgmm)[c];
    mn = w.x < mn ? w.x : mn;
    mx = w.y > mx ? w.y : mx;
  }
  mn = ds_wave_min(mn);
  mx = gsr_wave_max_u32(mx);
  if (lane == 0) { sh[wave] = mn; sh[4 + wave] = mx; }
  __syncthreads();
  mn = min(min(sh[0], sh[1]), min(sh[2], sh[3]));
  mx = max(max(sh[4], sh[5]), max(sh[6], sh[7]));
  __syncthreads();
  DsRange r;
  r.any = mn <= mx;
  r.lo = mn;
  const uint32_t range = r.any ? mx - mn : 0u;
  const int bits = range ? 32 - __clz(range) : 0;
  r.shift = bits > lognb ? (uint32_t)(bits - lognb) : 0u;
  return r;
}

// ---- per-chunk min / max of the visible keys; clears the histograms (grid: chunks x views)
__global__ void __launch_bounds__(256)
k_ds_range(const uint32_t* __restrict__ keys, const int64_t P, const uint32_t chunks, uint32_t* __restrict__ state,
           const uint32_t clear_words, size_t bstride) {
  keys = batch_ptr(keys, bstride); state = batch_ptr(state, bstride);
  const DsState s = ds_carve(state, P);
  __shared__ uint32_t sh[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this workgroup's share of the words to clear: coarse | fine (contiguous)
  {
    const uint32_t per = (clear_words + chunks - 1) / chunks;
    const uint32_t c0 = blockIdx.x * per, c1 = min(clear_words, c0 + per);
    for (uint32_t w = c0 + tid; w < c1; w += 256) s.coarse[w] = 0u;
  }
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  const int64_t base = (int64_t)blockIdx.x * kDsChunk;
#pragma unroll
  for (int it = 0; it < (int)(kDsChunk / 256); ++it) {
    const int64_t e = base + it * 256 + tid;
    const uint32_t k = e < P ? keys[e] : 0xFFFFFFFFu;
    if (k != 0xFFFFFFFFu) {
      mn = k < mn ? k : mn;
      mx = k > mx ? k : mx;
    }
  }
  mn = ds_wave_min(mn);
  mx = gsr_wave_max_u32(mx);
  if (lane == 0) { sh[wave] = mn; sh[4 + wave] = mx; }
  __syncthreads();
  if (tid == 0)
    reinterpret_cast<uint2*>(s.wgmm)[blockIdx.x] =
        make_uint2(min(min(sh[0], sh[1]), min(sh[2], sh[3])), max(max(sh[4], sh[5]), max(sh[6], sh[7])));
}

// ---- fine histogram + coarse sums (grid: chunks x views)
__global__ void __launch_bounds__(256)
k_ds_hist(const uint32_t* __restrict__ keys, const int64_t P, const uint32_t chunks, const int lognb,
          uint32_t* __restrict__ state, size_t bstride) {
  keys = batch_ptr(keys, bstride); state = batch_ptr(state, bstride);
  const DsState s = ds_carve(state, P);
  __shared__ uint32_t sh[8];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kDsChunk;
  uint32_t k[kDsChunk / 256];
#pragma unroll
  for (int it = 0; it < (int)(kDsChunk / 256); ++it) {          // (the keys are requested before the range words)
    const int64_t e = base + it * 256 + tid;
    k[it] = e < P ? keys[e] : 0xFFFFFFFFu;
  }
  const DsRange r = ds_range(s.wgmm, chunks, lognb, sh);
  uint32_t* coarse = s.coarse + (size_t)(blockIdx.x & (kDsReplicas - 1)) * ((size_t)1 << (lognb - 8));
#pragma unroll
  for (int it = 0; it < (int)(kDsChunk / 256); ++it) {
    if (k[it] == 0xFFFFFFFFu) continue;
    const uint32_t bin = (k[it] - r.lo) >> r.shift;
    atomicAdd(s.fine + bin, 1u);
    atomicAdd(coarse + (bin >> 8), 1u);
  }
}

// ---- bin starts + bucket table (grid: NB / 256 x views)
__global__ void __launch_bounds__(256)
k_ds_plan(const int64_t P, const int lognb, uint32_t* __restrict__ state, uint64_t* __restrict__ n_vis, size_t bstride) {
  state = batch_ptr(state, bstride); n_vis = batch_ptr(n_vis, bstride);
  const DsState s = ds_carve(state, P);
  __shared__ uint32_t wsum[4], wtot[4], wall[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nc = 1u << (lognb - 8), g = blockIdx.x;
  // coarse sums: in front of this group, and all of them
  uint32_t before = 0, all = 0;
  for (uint32_t c = tid; c < nc; c += 256) {
    uint32_t v = 0;
#pragma unroll
    for (int rp = 0; rp < kDsReplicas; ++rp) v += s.coarse[(size_t)rp * nc + c];
    all += v;
    if (c < g) before += v;
  }
  const uint32_t f = s.fine[(size_t)g * 256 + tid];
  uint32_t inc = f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
    if (lane >= o) inc += t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += (uint32_t)__shfl_xor((int)before, o, 64);
    all += (uint32_t)__shfl_xor((int)all, o, 64);
  }
  if (lane == 63) wtot[wave] = inc;
  if (lane == 0) { wsum[wave] = before; wall[wave] = all; }
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += wtot[w];
  const uint32_t pre = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  const uint32_t n = (wall[0] + wall[1]) + (wall[2] + wall[3]);
  const uint32_t start = pre + woff + inc - f;
  s.fine[(size_t)g * 256 + tid] = start;       // k_ds_scatter counts up from here
  // bucket j starts at the first bin start >= j T: the (non-empty) bin covering position j T - 1 ends there
  if (f) {
    const uint32_t nx = start + f;
    for (uint32_t j = start / kDsT + 1; j <= nx / kDsT; ++j) s.bstart[j] = nx;
  }
  if (g == 0 && tid == 0) {
    s.bstart[0] = 0u;
    s.bstart[(n + kDsT - 1) / kDsT] = n;       // (n a multiple of T: the last bin wrote the same value)
    *n_vis = (uint64_t)n;
  }
}

// ---- pairs grouped by bin (grid: chunks x views)
__global__ void __launch_bounds__(256)
k_ds_scatter(const uint32_t* __restrict__ keys, const int64_t P, const uint32_t chunks, const int lognb,
             uint32_t* __restrict__ state, uint32_t* __restrict__ tmpk, uint32_t* __restrict__ tmpv, size_t bstride) {
  keys = batch_ptr(keys, bstride); state = batch_ptr(state, bstride);
  tmpk = batch_ptr(tmpk, bstride); tmpv = batch_ptr(tmpv, bstride);
  const DsState s = ds_carve(state, P);
  __shared__ uint32_t sh[8];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kDsChunk;
  uint32_t k[kDsChunk / 256];
#pragma unroll
  for (int it = 0; it < (int)(kDsChunk / 256); ++it) {
    const int64_t e = base + it * 256 + tid;
    k[it] = e < P ? keys[e] : 0xFFFFFFFFu;
  }
  const DsRange r = ds_range(s.wgmm, chunks, lognb, sh);
  uint32_t pos[kDsChunk / 256];
#pragma unroll
  for (int it = 0; it < (int)(kDsChunk / 256); ++it)           // all returning atomics in flight together
    pos[it] = (k[it] != 0xFFFFFFFFu) ? atomicAdd(s.fine + ((k[it] - r.lo) >> r.shift), 1u) : 0u;
#pragma unroll
  for (int it = 0; it < (int)(kDsChunk / 256); ++it) {
    if (k[it] == 0xFFFFFFFFu) continue;
    tmpk[pos[it]] = k[it];
    tmpv[pos[it]] = (uint32_t)(base + it * 256 + tid);
  }
}

// ---- one bucket per workgroup (grid: buckets x views)
// F2: a bucket beyond the LDS capacity, sorted by this workgroup alone with a stable LSD radix sort out of global memory:
// 3 passes over the bytes of the Gaussian index (< 2^24), then 4 over the bytes of the key -> order (key, index). Ping-pong
// between (tmpk, tmpv) and (altk, altv) on the bucket's own range [a, a + m): an odd number of passes ends in alt = output.
__device__ __noinline__ void ds_sort_global(uint32_t* __restrict__ srck, uint32_t* __restrict__ srcv,
                                            uint32_t* __restrict__ dstk, uint32_t* __restrict__ dstv, const uint32_t a,
                                            const uint32_t m, uint32_t (*wh)[kRadix], uint32_t* base, uint32_t* wtot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int pass = 0; pass < 7; ++pass) {
    const bool by_val = pass < 3;
    const int shift = 8 * (by_val ? pass : pass - 3);
    base[tid] = 0;
    __syncthreads();
    for (uint32_t q = tid; q < m; q += 256) {
      const uint32_t x = by_val ? srcv[a + q] : srck[a + q];
      atomicAdd(&base[(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    {
      const uint32_t x = base[tid];
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
      base[tid] = woff + inc - x;
    }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < m; c0 += 1024) {
#pragma unroll
      for (int w = 0; w < 4; ++w) wh[w][tid] = 0;
      __syncthreads();
      uint32_t kk[4], vv[4], rank[4];
      bool ok[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t e = c0 + (uint32_t)(wave * 256 + it * 64 + lane);
        ok[it] = e < m;
        kk[it] = ok[it] ? srck[a + e] : 0u;
        vv[it] = ok[it] ? srcv[a + e] : 0u;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t d = ((by_val ? vv[it] : kk[it]) >> shift) & 255u;
        const unsigned long long mm = match_digit(d, ok[it]);
        const int leader = __ffsll((long long)mm) - 1;
        uint32_t old = 0;
        if (ok[it] && lane == leader) {
          old = wh[wave][d];
          wh[wave][d] = old + (uint32_t)__popcll(mm);
        }
        old = (uint32_t)__shfl((int)old, ok[it] ? leader : lane, 64);
        rank[it] = old + (uint32_t)__popcll(mm & lt);
        GSR_LDS_ORDER();
      }
      __syncthreads();
      {
        uint32_t run = base[tid];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t c = wh[w][tid];
          wh[w][tid] = run;
          run += c;
        }
        base[tid] = run;
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (!ok[it]) continue;
        const uint32_t d = ((by_val ? vv[it] : kk[it]) >> shift) & 255u;
        const uint32_t p = a + wh[wave][d] + rank[it];
        dstk[p] = kk[it];
        dstv[p] = vv[it];
      }
      __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    uint32_t* t = srck; srck = dstk; dstk = t;
    t = srcv; srcv = dstv; dstv = t;
  }
}

__global__ void __launch_bounds__(256)
k_ds_sort(const int64_t P, uint32_t* __restrict__ state, const uint64_t* __restrict__ n_vis, uint32_t* __restrict__ tmpk,
          uint32_t* __restrict__ tmpv, uint32_t* __restrict__ altk, uint32_t* __restrict__ out_idx, size_t bstride) {
  state = batch_ptr(state, bstride); n_vis = batch_ptr(n_vis, bstride); tmpk = batch_ptr(tmpk, bstride);
  tmpv = batch_ptr(tmpv, bstride); altk = batch_ptr(altk, bstride); out_idx = batch_ptr(out_idx, bstride);
  const DsState s = ds_carve(state, P);
  __shared__ uint32_t skey[kDsCap], sval[kDsCap];
  __shared__ uint32_t wh[4][kRadix];
  __shared__ uint32_t sbase[kRadix];
  __shared__ uint32_t wtot[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = (uint32_t)*n_vis, j = blockIdx.x;
  if ((uint64_t)j * kDsT >= n) return;
  const uint32_t a = s.bstart[j], e = s.bstart[j + 1];
  if (e <= a) return;
  const uint32_t m = e - a;
  if (m > kDsCap) {
    ds_sort_global(tmpk, tmpv, altk, out_idx, a, m, wh, sbase, wtot);
    return;
  }
  // element q of the bucket sits with thread (wave, lane) as item it: q = wave * chunk + it * 64 + lane -- wave-major, so
  // that "earlier element" = (earlier wave, or same wave and earlier item, or same item and lower lane): what the stable
  // ranking below assumes; chunk = the waves' equal share, a multiple of 64
  const int nit = (int)((m + 255u) / 256u);
  const uint32_t chunk = (uint32_t)nit * 64u;
  uint32_t key[kDsItems], val[kDsItems], rank[kDsItems];
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
  for (int it = 0; it < kDsItems; ++it) {
    key[it] = 0xFFFFFFFFu; val[it] = 0u;
    if (it < nit) {
      const uint32_t q = (uint32_t)wave * chunk + (uint32_t)(it * 64 + lane);
      if (q < m) {
        key[it] = tmpk[a + q];
        val[it] = tmpv[a + q];
        mn = key[it] < mn ? key[it] : mn;
        mx = key[it] > mx ? key[it] : mx;
      }
    }
  }
  mn = ds_wave_min(mn);
  mx = gsr_wave_max_u32(mx);
  if (lane == 0) { wtot[wave] = mn; wtot[4 + wave] = mx; }
  __syncthreads();
  const uint32_t kmin = min(min(wtot[0], wtot[1]), min(wtot[2], wtot[3]));
  const uint32_t kmax = max(max(wtot[4], wtot[5]), max(wtot[6], wtot[7]));
  const uint32_t range = kmax - kmin;
  const int npass = range ? (32 - __clz(range) + 7) / 8 : 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int pass = 0; pass < npass; ++pass) {
    const int shift = 8 * pass;
#pragma unroll
    for (int w = 0; w < 4; ++w) wh[w][tid] = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
      if (it < nit) {
        const uint32_t q = (uint32_t)wave * chunk + (uint32_t)(it * 64 + lane);
        const bool valid = q < m;
        const uint32_t d = ((key[it] - kmin) >> shift) & 255u;
        const unsigned long long mm = match_digit(d, valid);
        const int leader = __ffsll((long long)mm) - 1;
        uint32_t old = 0;
        if (valid && lane == leader) {
          old = wh[wave][d];
          wh[wave][d] = old + (uint32_t)__popcll(mm);
        }
        old = (uint32_t)__shfl((int)old, valid ? leader : lane, 64);
        rank[it] = old + (uint32_t)__popcll(mm & lt);
        GSR_LDS_ORDER();
      }
    }
    __syncthreads();
    // thread d: digit d's count over the waves -> the waves' offsets inside its run; runs laid out by an exclusive scan
    {
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t c = wh[w][tid];
        wh[w][tid] = run;
        run += c;
      }
      uint32_t inc = run;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += t;
      }
      if (lane == 63) wtot[wave] = inc;
      __syncthreads();
      uint32_t woff = 0;
      for (int w = 0; w < wave; ++w) woff += wtot[w];
      const uint32_t lstart = woff + inc - run;
#pragma unroll
      for (int w = 0; w < 4; ++w) wh[w][tid] += lstart;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
      if (it < nit) {
        const uint32_t q = (uint32_t)wave * chunk + (uint32_t)(it * 64 + lane);
        if (q < m) {
          const uint32_t lp = wh[wave][((key[it] - kmin) >> shift) & 255u] + rank[it];
          skey[lp] = key[it];
          sval[lp] = val[it];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
      if (it < nit) {
        const uint32_t q = (uint32_t)wave * chunk + (uint32_t)(it * 64 + lane);
        if (q < m) { key[it] = skey[q]; val[it] = sval[q]; }
      }
    }
    __syncthreads();
  }
  if (npass == 0) {          // all keys equal: the pairs still sit in registers only
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
      if (it < nit) {
        const uint32_t q = (uint32_t)wave * chunk + (uint32_t)(it * 64 + lane);
        if (q < m) { skey[q] = key[it]; sval[q] = val[it]; }
      }
    }
    __syncthreads();
  }
  // ties: inside a run of equal keys the pairs are in arrival order of the scatter's atomics; the reference order is by
  // Gaussian index. position = run start + number of smaller indices in the run (runs are 1 long except for equal depths)
#pragma unroll
  for (int it = 0; it < kDsItems; ++it) {
    if (it < nit) {
      const uint32_t q = (uint32_t)wave * chunk + (uint32_t)(it * 64 + lane);
      if (q < m) {
        const uint32_t k = key[it], v = val[it];
        uint32_t first = q, smaller = 0;
        while (first > 0 && skey[first - 1] == k) {
          --first;
          smaller += sval[first] < v;
        }
        for (uint32_t r2 = q + 1; r2 < m && skey[r2] == k; ++r2) smaller += sval[r2] < v;
        out_idx[a + first + smaller] = v;
      }
    }
  }
}

// The whole depth order of `batch` views (projection scratch buffers bstride bytes apart): keys (0xFFFFFFFF = culled) ->
// sorted_idx (visible Gaussians in (key, index) order) and the visible count. tmpk / tmpv / altk: three P-word buffers.
int depth_sort_launch(const uint32_t* keys, uint32_t* sorted_idx, uint32_t* tmpk, uint32_t* tmpv, uint32_t* altk,
                      uint32_t* state, uint64_t* n_vis, int64_t P, hipStream_t stream, int batch, size_t bstride) {
  const int lognb = ds_lognb(P);
  const uint32_t chunks = ds_chunks(P), nb = 1u << lognb, nc = nb >> 8;
  const uint32_t ny = (uint32_t)batch;
  hipLaunchKernelGGL(k_ds_range, dim3(chunks, ny), dim3(256), 0, stream, keys, P, chunks, state,
                     (uint32_t)(kDsReplicas * nc + nb), bstride);
  hipLaunchKernelGGL(k_ds_hist, dim3(chunks, ny), dim3(256), 0, stream, keys, P, chunks, lognb, state, bstride);
  hipLaunchKernelGGL(k_ds_plan, dim3(nc, ny), dim3(256), 0, stream, P, lognb, state, n_vis, bstride);
  hipLaunchKernelGGL(k_ds_scatter, dim3(chunks, ny), dim3(256), 0, stream, keys, P, chunks, lognb, state, tmpk, tmpv, bstride);
  hipLaunchKernelGGL(k_ds_sort, dim3(ds_buckets(P), ny), dim3(256), 0, stream, P, state, (const uint64_t*)n_vis, tmpk, tmpv,
                     altk, sorted_idx, bstride);
  return 0;
}

}  // namespace
