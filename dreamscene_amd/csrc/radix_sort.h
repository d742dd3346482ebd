// radix_sort.h -- stable LSD radix sort of (u32 key, u32 value) pairs on gfx950, shared by the tile binning
// (binning.hip) and the k-nearest-neighbour grid (knn.hip). See binning.hip for the design notes.
#pragma once
#include "gsr_common.h"
#include <type_traits>

namespace {

constexpr int kSortThreads = 256;
constexpr int kItemsLarge = 16;   // keys per thread for the N-sized tile sort (4096 keys / workgroup)
constexpr int kItemsSmall = 4;    // ... for the P-sized depth sort: 4x more workgroups, 4x shorter rank chains
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

// Batched launches (several views at once): blockIdx.y selects the view, whose buffers sit `bstride` bytes further.
// (byte arithmetic on the pointer itself, not through an integer: the compiler then still knows the address is GLOBAL and
//  emits global_load / global_store -- a round trip through uintptr_t gave flat_* accesses, which count against both
//  memory counters and serialise with the LDS traffic of the same wave)
template <typename T>
__device__ __forceinline__ T* batch_ptr(T* p, size_t bstride) {
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
  return p ? reinterpret_cast<T*>(reinterpret_cast<B*>(p) + (size_t)blockIdx.y * bstride) : p;
}
// Wave-private LDS counters updated by a leader lane and read by the wave's next step: LDS operations of one wave execute
// in program order, so ordinary accesses separated by a compiler barrier are enough (`volatile` made them flat_* sc0 sc1)
#define GSR_LDS_ORDER() asm volatile("" ::: "memory")

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
// THREADS per workgroup: a row is walked in chunks of 4 THREADS entries, three barriers each -- the column path's rows are
// 7 813 entries long at 500 k Gaussians (one per run of 64), 8 chunks with 256 threads, 2 with 1 024 (round 5: 8.8 us per
// 4-view launch with 256; the kernel is a serial chain of chunks, nothing else).
template <int THREADS>
__device__ __forceinline__ void radix_scan_body(uint32_t* __restrict__ hist, uint32_t nblk_stride, uint32_t* __restrict__ totals,
                                                const uint64_t* __restrict__ n_items, uint32_t per_block, size_t bstride) {
  hist = batch_ptr(hist, bstride); totals = batch_ptr(totals, bstride); n_items = batch_ptr(n_items, bstride);
  uint32_t nblk = nblk_stride;
  if (n_items) {
    const uint64_t used = (*n_items + per_block - 1) / per_block;
    if (used < nblk) nblk = (uint32_t)used;
  }
  constexpr int NW = THREADS / 64;
  __shared__ uint32_t wave_tot[NW];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* row = hist + (uint64_t)blockIdx.x * nblk_stride;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  constexpr uint32_t kPer = 4;
  for (uint32_t base = 0; base < nblk; base += THREADS * kPer) {
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
#pragma unroll
    for (int w = 0; w < NW; ++w) woff += (w < wave) ? wave_tot[w] : 0u;
    const uint32_t carry = carry_s;
    uint32_t run = carry + woff + inc - s;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      if (first + k < nblk) row[first + k] = run;
      run += x[k];
    }
    __syncthreads();
    if (tid == THREADS - 1) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) totals[blockIdx.x] = carry_s;
}
__global__ void __launch_bounds__(256) k_radix_scan(uint32_t* __restrict__ hist, uint32_t nblk_stride,
                                                    uint32_t* __restrict__ totals,
                                                    const uint64_t* __restrict__ n_items, uint32_t per_block,
                                                    size_t bstride) {
  radix_scan_body<256>(hist, nblk_stride, totals, n_items, per_block, bstride);
}
__global__ void __launch_bounds__(1024) k_radix_scan_wide(uint32_t* __restrict__ hist, uint32_t nblk_stride,
                                                          uint32_t* __restrict__ totals,
                                                          const uint64_t* __restrict__ n_items, uint32_t per_block,
                                                          size_t bstride) {
  radix_scan_body<1024>(hist, nblk_stride, totals, n_items, per_block, bstride);
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
  uint32_t* mywh = wh[wave];
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
    GSR_LDS_ORDER();
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

// ================================================================================= one-sweep passes (round 3)
// The three-kernel pass above (histogram -> 256-row scan -> scatter) costs 12 dependent launches for a 32-bit key on a
// few MB: launch-latency bound (157 us per 4-view step for 8 MB of keys). The one-sweep form needs
// 2 + passes launches:
//   k_os_hist   reads the keys ONCE and leaves the global digit histograms of ALL passes (a digit histogram does not
//               depend on the order of the keys, so it is valid for every later pass; culled keys are not counted);
//   k_os_pass   one launch per pass: a workgroup takes the next tile (a ticket: tiles start in ticket order, so every
//               predecessor of a tile is running or done -- nothing below depends on the dispatch order or on where a
//               workgroup runs), ranks its keys, publishes its per-digit counts and obtains the number of equal digits
//               in all earlier tiles by DECOUPLED LOOK-BACK over the predecessors' published words; then it sorts the
//               tile by digit in LDS and writes runs of consecutive positions (coalesced; the old scatter stored
//               element by element).
// Look-back words: one u32 per (tile, digit) = tag << 28 | value, tag = 2 pass + 1 (this tile's count: "aggregate") or
// 2 pass + 2 (count of this and all earlier tiles: "inclusive"), 0 = nothing yet. The table is zeroed ONCE per sort (tags
// tell the passes apart). Written and polled with relaxed agent-scope atomics ONLY -- a single aligned word is its own
// flag, so no fence is needed (an agent-scope release fence writes the XCD's L2 back: measured 3.5x on K8 in round 2).
// Values need 28 bits: sorts of 2^28 elements or more take the three-kernel passes.
// Where a pass spends its time (tools/probe/sort_phases.hip, 4 views x 500 k depth keys, 2048-key tiles: 245 tiles per view,
// all resident and starting together; us, mean per tile): ticket + digit-base scan 2.7, load + rank 5.0, publish + look-back
// 10 (tile 1: 0.5, tile 15: 5, the last quarter of the list: 12.6), LDS sort 1.7, write 1.4; launch span 28. The look-back is
// the wait for the predecessors' words to become VISIBLE across XCDs plus the walk: a window of 32 or 64 instead of 16 is
// slower (144 / 197 vs 131 us per sort: the walk is bandwidth, ~1 KB per predecessor and tile), and a two-level scheme
// (group rows of 16 tiles: 31 words per digit instead of up to 245) needs one more publish -> visible hop and takes the same
// 10 us. What helps is fewer tiles: 4096-key tiles halve the chain (look-back 6 us, pass 21 us, sort 132 -> 112 us).
#ifndef GSR_OS_ITEMS_SMALL
#define GSR_OS_ITEMS_SMALL 16
#endif
constexpr int kOsItemsSmall = GSR_OS_ITEMS_SMALL;     // keys per thread, P-sized sorts (4096-key tiles: see the phase timings below)
constexpr int kOsMaxPasses = 4;
#ifndef GSR_OS_HIST_TILE
#define GSR_OS_HIST_TILE 2048
#endif
constexpr uint32_t kOsHistTile = GSR_OS_HIST_TILE;  // keys per workgroup of k_os_hist
constexpr uint32_t kOsTableOff = kOsMaxPasses * kRadix + 16;   // u32 words: [passes][256] histograms | 4 tickets (+ pad) | table
// A pass whose digit is the same for every key is the identity; when it is the LAST pass and the caller allows it, nothing
// is copied: this word of the state is set instead and the result stays in the buffers the pass would have read (the top
// byte of the depths of one object: 8 us of every depth sort)
constexpr uint32_t kOsSkipFlag = kOsMaxPasses * kRadix + 8;
// Words [kOsEarlyN, kOsEarlyN + 1] (one aligned u64): the sum of a per-key WEIGHT over the keys that survive the first
// pass, accumulated by k_os_hist when the caller hands it a weight array -- for the depth sort the packed tile rectangle
// of every Gaussian (w x h = its number of (tile, Gaussian) pairs), i.e. the pair count N of the view. The first tile of
// the first pass copies it to a caller-given (page-locked host) word: N reaches the host when the depth sort has barely
// started (K1 + one histogram launch) instead of after the whole sort + the column counts (round 5: the per-view
// interface blocked ~100 us per call on that word).
constexpr uint32_t kOsEarlyN = kOsMaxPasses * kRadix + 10;
constexpr uint64_t kOsMaxN = 1ull << 28;

typedef __attribute__((address_space(1))) uint32_t gsr_gu32;
__device__ __forceinline__ gsr_gu32* gsr_global(uint32_t* p) { return (gsr_gu32*)(uintptr_t)p; }

__host__ __device__ inline uint32_t os_tiles(uint64_t n, int items) {
  const uint64_t t = (uint64_t)kSortThreads * items;
  return (uint32_t)((n + t - 1) / t);
}
// u32 words of the one-sweep state of a sort of up to n keys
__host__ inline size_t os_state_words(uint64_t n, int items) { return (size_t)kOsTableOff + (size_t)kRadix * os_tiles(n, items); }

template <bool DROP>
__global__ void __launch_bounds__(kSortThreads)
k_os_hist(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ n_dev, uint64_t cap, int passes,
          uint32_t* __restrict__ os, size_t bstride, const uint32_t* __restrict__ wrect) {
  keys = batch_ptr(keys, bstride); n_dev = batch_ptr(n_dev, bstride); os = batch_ptr(os, bstride);
  wrect = batch_ptr(wrect, bstride);
  __shared__ uint32_t h[kOsMaxPasses][kRadix];
  __shared__ uint32_t wsum[kSortThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const uint64_t n = eff_count(n_dev, cap);
  const uint64_t base = (uint64_t)blockIdx.x * kOsHistTile;
  if (base >= n) return;
#pragma unroll
  for (int p = 0; p < kOsMaxPasses; ++p) h[p][tid] = 0;
  __syncthreads();
  constexpr int kPer = kOsHistTile / kSortThreads;
  uint32_t kk[kPer];
  uint32_t weight = 0;          // (this thread's share of the weight sum, see kOsEarlyN)
#pragma unroll
  for (int it = 0; it < kPer; ++it) {
    const uint64_t e = base + (uint64_t)(it * kSortThreads + tid);
    kk[it] = e < n ? keys[e] : 0xFFFFFFFFu;
    if (wrect && e < n && (!DROP || kk[it] != 0xFFFFFFFFu)) {
      const uint32_t r = wrect[e];                             // packed x0 | y0 << 8 | (w - 1) << 16 | (h - 1) << 24
      weight += (((r >> 16) & 255u) + 1u) * ((r >> 24) + 1u);
    }
  }
  if (wrect) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) weight += (uint32_t)__shfl_xor((int)weight, o, 64);
    if (lane == 0) wsum[tid >> 6] = weight;
  }
#pragma unroll
  for (int it = 0; it < kPer; ++it) {
    const uint64_t e = base + (uint64_t)(it * kSortThreads + tid);
    const bool valid = (e < n) && (!DROP || kk[it] != 0xFFFFFFFFu);
    const unsigned long long vm = __ballot(valid);
    if (vm == 0ull) continue;
    const uint32_t k = kk[it];
    // low digits: effectively random -> one LDS atomic per key and digit
    if (valid) {
      atomicAdd(&h[0][k & (kRadix - 1)], 1u);
      if (passes > 1) atomicAdd(&h[1][(k >> kRadixBits) & (kRadix - 1)], 1u);
    }
    if (passes > 2) {
      // high digits (depth exponent / high mantissa bits, high tile or cell bits): the same for the whole wave most of
      // the time -> one atomic per wave and digit instead of 64 on one address
      const int first = __ffsll((long long)vm) - 1;
      const uint32_t hi = k >> (2 * kRadixBits), hi0 = (uint32_t)__shfl((int)hi, first, 64);
      if (__ballot(valid && hi != hi0) == 0ull) {
        if (lane == first) {
          atomicAdd(&h[2][hi0 & (kRadix - 1)], (uint32_t)__popcll(vm));
          if (passes > 3) atomicAdd(&h[3][hi0 >> kRadixBits], (uint32_t)__popcll(vm));
        }
      } else if (valid) {
        atomicAdd(&h[2][hi & (kRadix - 1)], 1u);
        if (passes > 3) atomicAdd(&h[3][hi >> kRadixBits], 1u);
      }
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    const uint32_t c = h[p][tid];
    if (c) atomicAdd(&os[p * kRadix + tid], c);
  }
  if (wrect && tid == 0) {
    const uint64_t w = ((uint64_t)wsum[0] + wsum[1]) + ((uint64_t)wsum[2] + wsum[3]);
    if (w) atomicAdd(reinterpret_cast<unsigned long long*>(os + kOsEarlyN), (unsigned long long)w);
  }
}

// Look-back of one digit (column `col` of the table) from tile `tile` - 1 downwards: sum of the aggregates met, up to and
// including the first inclusive word. The words of the next kOsWindow predecessors are requested together (independent
// loads in flight) and consumed in order, so a walk over k published words costs k / kOsWindow round trips, not k.
#ifndef GSR_OS_WINDOW
#define GSR_OS_WINDOW 16
#endif
constexpr int kOsWindow = GSR_OS_WINDOW;
__device__ __forceinline__ uint32_t os_look_back(gsr_gu32* table, uint32_t tile, int col, uint32_t tagA, uint32_t tagI) {
  uint32_t excl = 0;
  int p = (int)tile - 1;
  for (;;) {
    uint32_t w[kOsWindow];
#pragma unroll
    for (int j = 0; j < kOsWindow; ++j)
      w[j] = (p - j >= 0) ? __hip_atomic_load(table + (size_t)(p - j) * kRadix + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : 0u;
    bool done = false, stalled = false;
#pragma unroll
    for (int j = 0; j < kOsWindow; ++j) {
      if (done || stalled) continue;
      const uint32_t tag = w[j] & 0xF0000000u;
      if (tag == tagI) { excl += w[j] & 0x0FFFFFFFu; done = true; }
      else if (tag == tagA) { excl += w[j] & 0x0FFFFFFFu; --p; }
      else stalled = true;                  // not published yet: poll again from here (tile 0 publishes inclusive words)
    }
    if (done) return excl;
    if (stalled) __builtin_amdgcn_s_sleep(1);
  }
}

// Phase stamps for tools/probe/sort_phases.hip (compiled out of the library)
#ifdef GSR_OS_TRACE
__device__ unsigned long long* g_os_trace = nullptr;      // [pass][view][block][8] realtime stamps (100 MHz)
#define GSR_OS_STAMP(k) do { if (threadIdx.x == 0 && g_os_trace) \
    g_os_trace[(((size_t)pass * 4 + blockIdx.y) * 4096 + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define GSR_OS_STAMP(k) do { } while (0)
#endif

template <bool IOTA, int ITEMS, bool DROP>
__global__ void __launch_bounds__(kSortThreads)
k_os_pass(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
          uint32_t* __restrict__ vals_out, const uint64_t* __restrict__ n_dev, uint64_t cap, int pass,
          uint32_t* __restrict__ os, uint64_t* __restrict__ n_out, size_t bstride, const int may_skip,
          uint64_t* __restrict__ early_out) {
  keys_in = batch_ptr(keys_in, bstride); vals_in = batch_ptr(vals_in, bstride);
  keys_out = batch_ptr(keys_out, bstride); vals_out = batch_ptr(vals_out, bstride);
  n_dev = batch_ptr(n_dev, bstride); os = batch_ptr(os, bstride); n_out = batch_ptr(n_out, bstride);
  constexpr uint32_t T = kSortThreads * ITEMS;
  __shared__ uint32_t wh[4][kRadix];   // running per-wave digit counters, then the waves' offsets inside the tile's digit run
  __shared__ uint32_t gbase[kRadix];   // global position of local position 0 of digit d's run (may wrap: used modulo 2^32)
  __shared__ uint32_t skey[T], sval[T];
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t s_tile, s_ntile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int shift = pass * kRadixBits;
  const uint64_t n = eff_count(n_dev, cap);
  if constexpr (!DROP) {
    // Every key carries the same digit (the top byte of the depths of one object, the high byte of 10-bit tile ids, ...):
    // the pass is the identity permutation -- a coalesced copy: no ticket, no ranking, no look-back (all workgroups of the
    // pass take this branch, nobody waits for anybody).
    const uint32_t x = os[pass * kRadix + tid];
    if (__syncthreads_or((uint64_t)x == n && n > 0)) {
      if (may_skip) {
        // the LAST pass of a sort whose consumers take the result from wherever it is (kOsSkipFlag): nothing moves
        if (blockIdx.x == 0 && tid == 0) os[kOsSkipFlag] = 1u;
        return;
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const uint64_t e = sort_index<ITEMS>(blockIdx.x, wave, it, lane);
        if (e < n) {
          keys_out[e] = keys_in[e];
          vals_out[e] = IOTA ? (uint32_t)e : vals_in[e];
        }
      }
      return;
    }
  }
  GSR_OS_STAMP(0);
  if (tid == 0) s_tile = atomicAdd(&os[kOsMaxPasses * kRadix + pass], 1u);      // the ticket: tiles START in this order
#pragma unroll
  for (int w = 0; w < 4; ++w) wh[w][tid] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  // the weight sum k_os_hist left (kOsEarlyN) -> the caller's word of this view (page-locked host memory: the pair count of
  // the view, long before the sort is over). Ticket 0 exists in every launch (n > 0 or not).
  if (early_out && tile == 0 && tid == 0)
    __hip_atomic_store(early_out + blockIdx.y, *reinterpret_cast<const uint64_t*>(os + kOsEarlyN), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  if ((uint64_t)tile * T >= n) return;     // (every later ticket is beyond n as well: nobody waits for this tile)
  // exclusive scan of the pass's global digit histogram: where digit d starts in the output
  uint32_t dbase;
  {
    const uint32_t x = os[pass * kRadix + tid];
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
    dbase = woff + inc - x;
    if (DROP && n_out && tile == 0 && tid == kSortThreads - 1) *n_out = (uint64_t)(woff + inc);   // survivors
  }
  GSR_OS_STAMP(1);
#ifdef GSR_OS_TRACE
  if (threadIdx.x == 0 && g_os_trace) g_os_trace[(((size_t)pass * 4 + blockIdx.y) * 4096 + blockIdx.x) * 8 + 7] = tile;
#endif
  uint32_t* mywh = wh[wave];
  uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const uint64_t e = sort_index<ITEMS>(tile, wave, it, lane);
    const bool valid = e < n;
    key[it] = valid ? keys_in[e] : 0xFFFFFFFFu;
    val[it] = IOTA ? (uint32_t)e : (valid ? vals_in[e] : 0u);
  }
  const auto is_valid = [&](int it) {
    const uint64_t e = sort_index<ITEMS>(tile, wave, it, lane);
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
    GSR_LDS_ORDER();
  }
  __syncthreads();
  GSR_OS_STAMP(2);
  // thread d: this tile's count of digit d, the waves' offsets inside the run, the run's start inside the tile
  uint32_t cnt;
  {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = wh[w][tid];
      wh[w][tid] = run;
      run += c;
    }
    cnt = run;
  }
  // publish the count, then look back (thread d for digit d)
  gsr_gu32* table = gsr_global(os + kOsTableOff);
  const uint32_t tagA = (uint32_t)(2 * pass + 1) << 28, tagI = (uint32_t)(2 * pass + 2) << 28;
  uint32_t excl = 0;
  if (tile == 0) {
    __hip_atomic_store(table + tid, tagI | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __hip_atomic_store(table + (size_t)tile * kRadix + tid, tagA | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    excl = os_look_back(table, tile, tid, tagA, tagI);
    __hip_atomic_store(table + (size_t)tile * kRadix + tid, tagI | (excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  GSR_OS_STAMP(3);
  // start of digit d's run inside the tile: exclusive scan of cnt over the digits
  uint32_t lstart;
  {
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
      if (lane >= o) inc += t;
    }
    __syncthreads();                       // (wtot is free again: everybody is past the first scan)
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    lstart = woff + inc - cnt;
    if (tid == kSortThreads - 1) s_ntile = woff + inc;
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) wh[w][tid] += lstart;        // wave w's first local position of digit d
  gbase[tid] = dbase + excl - lstart;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (is_valid(it)) {
      const uint32_t d = (key[it] >> shift) & (kRadix - 1);
      const uint32_t lp = wh[wave][d] + rank[it];
      skey[lp] = key[it];
      sval[lp] = val[it];
    }
  }
  __syncthreads();
  GSR_OS_STAMP(4);
  const uint32_t ntile = s_ntile;
  for (uint32_t i = tid; i < ntile; i += kSortThreads) {
    const uint32_t k = skey[i];
    const uint32_t pos = gbase[(k >> shift) & (kRadix - 1)] + i;
    keys_out[pos] = k;
    vals_out[pos] = sval[i];
  }
  GSR_OS_STAMP(5);
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
// bytes of the `hist` scratch region a sort of up to n keys needs (three-kernel passes or one-sweep state, whichever is larger)
__host__ inline size_t sort_hist_bytes(uint64_t n, int items_legacy, int items_os) {
  const size_t legacy = (size_t)kRadix * sort_blocks(n, items_legacy) * 4;
  const size_t os = os_state_words(n, items_os) * 4;
  return align256(legacy > os ? legacy : os);
}

template <int ITEMS>
int radix_sort_u32_legacy(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, const uint64_t* n_dev, uint64_t cap,
                          int bits, bool iota, uint64_t* n_compact, uint32_t* hist, uint32_t* totals, hipStream_t stream,
                          int batch = 1, size_t bstride = 0);

// One full LSD sort of (u32 key, u32 value) over the key bits [0, bits). Buffers ping-pong between (k0,v0) and
// (k1,v1); returns 0 if the result is in (k0,v0), 1 if in (k1,v1). iota: values of the first pass are the element
// indices. n_compact (device, may be NULL; only with iota): the first pass drops the elements keyed 0xFFFFFFFF and
// stores the number of survivors there; the remaining passes (and the caller) work on that many elements.
// hist: sort_hist_bytes(cap, ITEMS, OS_ITEMS) bytes; totals: kRadix words (three-kernel passes only).
// 2 + passes launches (one-sweep, see above); sorts of 2^28 keys or more take the three-kernel passes.
template <int ITEMS, int OS_ITEMS>
int radix_sort_u32(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, const uint64_t* n_dev, uint64_t cap,
                   int bits, bool iota, uint64_t* n_compact, uint32_t* hist, uint32_t* totals, hipStream_t stream,
                   int batch = 1, size_t bstride = 0, bool state_cleared = false, bool last_pass_may_skip = false,
                   const uint32_t* early_rects = nullptr, uint64_t* early_out = nullptr) {
  const int passes = (bits + kRadixBits - 1) / kRadixBits;
  if (cap >= kOsMaxN || passes > kOsMaxPasses)
    return radix_sort_u32_legacy<ITEMS>(k0, v0, k1, v1, n_dev, cap, bits, iota, n_compact, hist, totals, stream, batch, bstride);
  const uint32_t ntile = os_tiles(cap, OS_ITEMS), nhist = (uint32_t)((cap + kOsHistTile - 1) / kOsHistTile);
  const dim3 grid(ntile, (uint32_t)batch), grid_h(nhist ? nhist : 1, (uint32_t)batch);
  // state_cleared: the caller's previous kernel left the os_state_words(cap, OS_ITEMS) words at `hist` zero (K1 does, for the
  // depth sort: one launch less in front of a latency-bound chain)
  if (!state_cleared &&
      gsr_zero_async(hist, os_state_words(cap, OS_ITEMS) * 4, stream, bstride, (uint32_t)batch) != hipSuccess) return passes & 1;
  const bool drop = iota && n_compact;
  // early_rects / early_out (depth sort of the column path only): see kOsEarlyN
  if (!drop) { early_rects = nullptr; early_out = nullptr; }
  if (!early_out) early_rects = nullptr;
  if (drop)
    hipLaunchKernelGGL((k_os_hist<true>), grid_h, dim3(kSortThreads), 0, stream, k0, n_dev, cap, passes, hist, bstride,
                       early_rects);
  else
    hipLaunchKernelGGL((k_os_hist<false>), grid_h, dim3(kSortThreads), 0, stream, k0, n_dev, cap, passes, hist, bstride,
                       (const uint32_t*)nullptr);
  uint32_t *ka = k0, *va = v0, *kb = k1, *vb = v1;
  for (int p = 0; p < passes; ++p) {
    if (p == 0 && drop) {
      hipLaunchKernelGGL((k_os_pass<true, OS_ITEMS, true>), grid, dim3(kSortThreads), 0, stream, ka, va, kb, vb, n_dev, cap, p,
                         hist, n_compact, bstride, 0, early_rects ? early_out : (uint64_t*)nullptr);
      n_dev = n_compact;
    } else if (p == 0 && iota) {
      hipLaunchKernelGGL((k_os_pass<true, OS_ITEMS, false>), grid, dim3(kSortThreads), 0, stream, ka, va, kb, vb, n_dev, cap, p,
                         hist, (uint64_t*)nullptr, bstride, 0, (uint64_t*)nullptr);
    } else {
      hipLaunchKernelGGL((k_os_pass<false, OS_ITEMS, false>), grid, dim3(kSortThreads), 0, stream, ka, va, kb, vb, n_dev, cap, p,
                         hist, (uint64_t*)nullptr, bstride, (last_pass_may_skip && p == passes - 1 && p > 0) ? 1 : 0,
                         (uint64_t*)nullptr);
    }
    uint32_t* t = ka; ka = kb; kb = t;
    t = va; va = vb; vb = t;
  }
  return passes & 1;
}

template <int ITEMS>
int radix_sort_u32_legacy(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, const uint64_t* n_dev, uint64_t cap,
                   int bits, bool iota, uint64_t* n_compact, uint32_t* hist, uint32_t* totals, hipStream_t stream,
                   int batch, size_t bstride) {
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
