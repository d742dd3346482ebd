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
//
// Column path (grids of at most 256 x 256 tiles, i.e. images up to 4096 x 4096): step 2 and the first pass of
// step 3 are ONE kernel. tile id = ty * gx + tx, so an LSD sort by tile id is "stable by tx, then stable by ty".
// A Gaussian's pairs with a given tx are its tiles ty = y0..y1-1, contiguous in emission order, so the position of
// the strip (Gaussian s, column tx) in the tx-sorted list is  colstart[tx] + sum over Gaussians s' < s (depth
// order) covering tx of h(s')  -- a per-column prefix sum over the depth-ordered rectangles. k_col_hist / k_radix_scan
// / k_col_plan compute it per run of 64 Gaussians (and N falls out), k_emit_cols writes every pair straight to its
// tx-sorted position (one wave per run; the run's pairs are the wave's elements), and the remaining pass (by ty, 1-byte
// keys, workgroups aligned to column starts) also yields the tile ranges: the first workgroup of column tx knows,
// for every ty, the final position of the first pair of tile (ty, tx).
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


// =================================================================================== column path (see header)
constexpr int kColRun = 64;        // Gaussians (in depth order) per emission wave: short runs = many stores in flight
constexpr int kPass2Items = 16;    // pairs per thread in the ty pass (4096 per workgroup)
constexpr uint32_t kPass2Block = kSortThreads * kPass2Items;

__device__ __forceinline__ unsigned long long match_digit_n(uint32_t d, bool valid, int nbits) {
  unsigned long long m = __ballot(valid);
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

__device__ __forceinline__ uint32_t rect_x0(uint32_t r) { return r & 255u; }
__device__ __forceinline__ uint32_t rect_y0(uint32_t r) { return (r >> 8) & 255u; }
__device__ __forceinline__ uint32_t rect_w(uint32_t r) { return ((r >> 16) & 255u) + 1u; }
__device__ __forceinline__ uint32_t rect_h(uint32_t r) { return (r >> 24) + 1u; }

// Workgroup b of a 1-D grid runs on XCD b % 8 (each XCD has its own L2). Neighbouring blocks -- runs of one column here,
// runs of 64 depth-ordered Gaussians in k_emit_cols -- write neighbouring list positions, i.e. they share the cache lines at
// their seams; on different XCDs each L2 holds its part of such a line and writes it back partially (a read-modify-write
// at the memory: k_emit_cols moved 177 MB for 49 MB of pairs). So logical block ids are dealt in CONTIGUOUS ranges per XCD:
// raw index b -> (b % 8) * ceil(n / 8) + b / 8 for the n blocks in use (the grid is rounded up to a multiple of 8).
__device__ __forceinline__ uint32_t xcd_chunked(uint32_t b, uint32_t n) {
  const uint32_t per = (n + 7u) / 8u, j = b >> 3, id = (b & 7u) * per + j;
  return (j < per && id < n) ? id : 0xFFFFFFFFu;
}

// Per run of 64 depth-ordered Gaussians (one wave): pairs per tile column -> hist1[tx][run]; also the rectangles in
// depth order.
__global__ void __launch_bounds__(256)
k_col_hist(const uint64_t* __restrict__ n_vis, const uint32_t* __restrict__ sorted_idx,
           const uint32_t* __restrict__ sorted_idx_alt, const uint32_t* __restrict__ os_state,
           const uint32_t* __restrict__ rects, uint32_t* __restrict__ rect_sorted, const int gx, const uint32_t nrun,
           uint32_t* __restrict__ hist1, size_t bstride) {
  n_vis = batch_ptr(n_vis, bstride); rects = batch_ptr(rects, bstride);
  // the depth order is in `sorted_idx`, or -- when the last pass of the sort was the identity and moved nothing
  // (radix_sort.h, kOsSkipFlag) -- still in the buffer that pass would have read
  sorted_idx = batch_ptr(batch_ptr(os_state, bstride)[kOsSkipFlag] ? sorted_idx_alt : sorted_idx, bstride);
  rect_sorted = batch_ptr(rect_sorted, bstride); hist1 = batch_ptr(hist1, bstride);
  __shared__ uint32_t bins[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int w = 0; w < 4; ++w) bins[w][tid] = 0;
  __syncthreads();
  uint32_t* mybins = bins[wave];
  // (blocks dealt to the XCDs in contiguous ranges, see xcd_chunked: the four runs of a block are 16 bytes of every column's
  //  row of hist1, neighbouring blocks share those lines)
  const int64_t nv = (int64_t)*n_vis;
  const uint32_t blk = xcd_chunked(blockIdx.x, (uint32_t)((nv + 255) / 256));
  if (blk == 0xFFFFFFFFu) return;
  const int64_t s = (int64_t)blk * 256 + tid;
  const bool in = s < nv;
  uint32_t r = 0, w = 0, h = 0, x0 = 0;
  if (in) {
    r = rects[sorted_idx[s]];
    rect_sorted[s] = r;
    x0 = rect_x0(r); w = rect_w(r); h = rect_h(r);
  }
  constexpr uint32_t kCoop = 8;
  if (in && w <= kCoop)
    for (uint32_t c = 0; c < w; ++c) atomicAdd(&mybins[x0 + c], h);
  unsigned long long big = __ballot(in && w > kCoop);
  while (big) {   // wide footprints: the wave adds one Gaussian's columns together
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t sx0 = (uint32_t)__shfl((int)x0, src, 64), sw = (uint32_t)__shfl((int)w, src, 64);
    const uint32_t sh = (uint32_t)__shfl((int)h, src, 64);
    for (uint32_t c = lane; c < sw; c += 64) atomicAdd(&mybins[sx0 + c], sh);
  }
  __syncthreads();
  if (tid < gx) {
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const uint32_t run = blk * 4 + w2;
      if (run < nrun && (int64_t)run * kColRun < nv) hist1[(uint64_t)tid * nrun + run] = bins[w2][tid];
    }
  }
}

// colstart[0..gx] = exclusive scan of the column totals (saturating u32), *n_pairs = N (64-bit).
__global__ void __launch_bounds__(256)
k_col_plan(const uint32_t* __restrict__ totals1, const int gx, uint32_t* __restrict__ colstart,
           uint64_t* __restrict__ n_pairs, size_t bstride, uint64_t* __restrict__ n_pairs_all) {
  totals1 = batch_ptr(totals1, bstride); colstart = batch_ptr(colstart, bstride); n_pairs = batch_ptr(n_pairs, bstride);
  __shared__ uint64_t wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t x = tid < gx ? (uint64_t)totals1[tid] : 0ull;
  uint64_t inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  uint64_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += wtot[w];
  const uint64_t excl = woff + inc - x;
  if (tid < gx) colstart[tid] = excl > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)excl;
  if (tid == 255) {
    const uint64_t n = woff + inc;
    colstart[gx] = n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n;
    *n_pairs = n;
    if (n_pairs_all) n_pairs_all[blockIdx.y] = n;   // batched: the counts of all views side by side (one copy to the host)
  }
}

// One wave per run of 64 depth-ordered Gaussians. Lanes enumerate the run's PAIRS (Gaussian-major, then row-major
// inside the rectangle) 64 at a time; a pair's position in the "sorted by tx" list is the running counter of its
// column (start: colstart + scanned hist1) plus its rank among the round's pairs of the same column -- the stable
// scatter of a radix pass, fused with the generation of its input. One word per pair: ty << 24 | Gaussian index.
// Workgroups of kEmitWaves independent waves (one run each, wave-private LDS, no workgroup barrier). Round 6, one call, 1 / 4 / 8
// waves per workgroup: 12.7 / 12.7 / 12.8 us per view at C3 (the kernel is the serial chain of each wave -- ~22 k cycles for ~900
// vector instructions, 26 waves resident per CU by SQ_WAVE_CYCLES -- not the dispatch of its 31 k workgroups), 9.5 / 8.2 / 8.9 us
// per view on the 2 M indoor scene: four.
constexpr int kEmitWaves = 4;
__global__ void __launch_bounds__(64 * kEmitWaves)
k_emit_cols(const uint64_t* __restrict__ n_vis, const int gx, const int nbits_x, const uint32_t* __restrict__ rect_sorted,
            const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ sorted_idx_alt,
            const uint32_t* __restrict__ os_state, const uint32_t* __restrict__ hist1, const uint32_t nrun,
            const uint32_t* __restrict__ colstart, const uint32_t cap, uint32_t* __restrict__ vals, size_t ps, size_t ss) {
  // several views per launch (blockIdx.y): projection scratch buffers ps bytes apart, sort scratch buffers ss bytes apart
  n_vis = batch_ptr(n_vis, ps); rect_sorted = batch_ptr(rect_sorted, ps);
  sorted_idx = batch_ptr(batch_ptr(os_state, ps)[kOsSkipFlag] ? sorted_idx_alt : sorted_idx, ps);      // (see k_col_hist)
  hist1 = batch_ptr(hist1, ps); colstart = batch_ptr(colstart, ps); vals = batch_ptr(vals, ss);
  __shared__ uint32_t col_run_s[kEmitWaves][256];
  __shared__ uint32_t pbase_s[kEmitWaves][65];
  __shared__ uint32_t rs_s[kEmitWaves][64], ids_s[kEmitWaves][64], mg_s[kEmitWaves][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* col_run = col_run_s[wave];
  uint32_t* pbase = pbase_s[wave];
  uint32_t *rs = rs_s[wave], *ids = ids_s[wave], *mg = mg_s[wave];
  const int64_t nv = (int64_t)*n_vis;
  const uint32_t n_runs = (uint32_t)((nv + kColRun - 1) / kColRun);
  // (groups of kEmitWaves consecutive runs dealt to the XCDs in contiguous ranges: see col_blocks)
  const uint32_t grp = xcd_chunked(blockIdx.x, (n_runs + kEmitWaves - 1) / kEmitWaves);
  if (grp == 0xFFFFFFFFu) return;
  const uint32_t run_id = grp * kEmitWaves + (uint32_t)wave;
  if (run_id >= n_runs) return;
  const int64_t s0 = (int64_t)run_id * kColRun;
  for (int tx = lane; tx < gx; tx += 64) col_run[tx] = colstart[tx] + hist1[(uint64_t)tx * nrun + run_id];
  const bool in = s0 + lane < nv;
  const uint32_t r = in ? rect_sorted[s0 + lane] : 0u;
  const uint32_t w = rect_w(r);
  const uint32_t cnt = in ? w * rect_h(r) : 0u;
  uint32_t inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
    if (lane >= o) inc += t;
  }
  const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64);
  pbase[lane] = inc - cnt;
  if (lane == 0) pbase[64] = total;
  rs[lane] = r;
  ids[lane] = in ? sorted_idx[s0 + lane] : 0u;
  mg[lane] = w > 1 ? 0xFFFFFFFFu / w + 1u : 0u;   // floor(q / w) == umulhi(q, mg) for q < 2^16, w <= 256
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the LDS above is the wave's own)
  __builtin_amdgcn_wave_barrier();
  uint32_t* run = col_run;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t q0 = 0; q0 < total; q0 += 64) {
    const uint32_t q = q0 + (uint32_t)lane;
    const bool valid = q < total;
    int i = 0;   // the Gaussian pair q belongs to: largest i with pbase[i] <= q
#pragma unroll
    for (int st = 32; st > 0; st >>= 1)
      if (pbase[i + st] <= q) i += st;
    const uint32_t rr = rs[i], gid = ids[i], m_ = mg[i];
    const uint32_t rem = valid ? q - pbase[i] : 0u;
    const uint32_t ww = rect_w(rr);
    const uint32_t k = m_ ? __umulhi(rem, m_) : rem;
    const uint32_t c = m_ ? rem - k * ww : 0u;
    const uint32_t tx = rect_x0(rr) + c, ty = rect_y0(rr) + k;
    const unsigned long long m = match_digit_n(tx, valid, nbits_x);
    const int leader = __ffsll((long long)m) - 1;
    uint32_t old = 0;
    if (valid && lane == leader) {
      old = run[tx];
      run[tx] = old + (uint32_t)__popcll(m);
    }
    old = (uint32_t)__shfl((int)old, valid ? leader : lane, 64);
    const uint32_t p = old + (uint32_t)__popcll(m & lt);
    if (valid && p < cap) vals[p] = gid | (ty << 24);
    GSR_LDS_ORDER();
  }
}

// ---- second pass: stable by ty over column-aligned workgroups
struct ColBlocks {
  uint32_t blk;              // this workgroup's logical block (0xFFFFFFFF: none)
  uint32_t col, base, end;   // its column and element range [base, end)
  uint32_t total;            // logical blocks in use
};

// Every workgroup derives the block table from colstart: column tx owns max(1, ceil(cnt/4096)) workgroups.
__device__ __forceinline__ ColBlocks col_blocks(const uint32_t* __restrict__ colstart, int gx, uint32_t cap,
                                                uint32_t raw_b, uint32_t* sh_fb /*[257]*/, uint32_t* sh_tmp /*[8]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t cs = 0, ce = 0;
  if (tid < gx) {
    cs = min(colstart[tid], cap);
    ce = min(colstart[tid + 1], cap);
    if (ce < cs) ce = cs;
  }
  const uint32_t nb = tid < gx ? max(1u, (ce - cs + kPass2Block - 1) / kPass2Block) : 0u;
  uint32_t inc = nb;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) sh_tmp[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += sh_tmp[w];
  const uint32_t fb = woff + inc - nb;
  sh_fb[tid] = fb;
  if (tid == 255) sh_fb[256] = fb + nb;
  __syncthreads();
  ColBlocks cb;
  cb.total = sh_fb[256];
  cb.blk = xcd_chunked(raw_b, cb.total);
  if (nb && cb.blk >= fb && cb.blk < fb + nb) {       // (0xFFFFFFFF matches nobody)
    sh_tmp[4] = (uint32_t)tid;
    sh_tmp[5] = cs + (cb.blk - fb) * kPass2Block;
    sh_tmp[6] = ce;
  }
  __syncthreads();
  const bool any = cb.blk != 0xFFFFFFFFu;
  cb.col = any ? sh_tmp[4] : 0xFFFFFFFFu;
  cb.base = any ? sh_tmp[5] : 0u;
  cb.end = any ? min(sh_tmp[6], cb.base + kPass2Block) : 0u;
  return cb;
}

__global__ void __launch_bounds__(kSortThreads)
k_row_hist(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ colstart, const int gx, const uint32_t cap,
           const int nbits, const uint32_t nblk, uint32_t* __restrict__ hist, size_t ps, size_t ss) {
  keys = batch_ptr(keys, ss); colstart = batch_ptr(colstart, ps); hist = batch_ptr(hist, ss);
  __shared__ uint32_t h[kRadix];
  __shared__ uint32_t sh_fb[257];
  __shared__ uint32_t sh_tmp[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  h[tid] = 0;
  const ColBlocks cb = col_blocks(colstart, gx, cap, blockIdx.x, sh_fb, sh_tmp);
  // (the scan runs over all nblk columns of the table: the ones past the blocks in use are zeroed by their raw index)
  if (blockIdx.x >= cb.total && blockIdx.x < nblk) hist[(uint64_t)tid * nblk + blockIdx.x] = 0u;
  if (cb.blk == 0xFFFFFFFFu) return;
  if (cb.base < cb.end) {
    uint32_t kv[kPass2Items];
#pragma unroll
    for (int it = 0; it < kPass2Items; ++it) {   // all loads in flight before the first ballot
      const uint32_t e = cb.base + (uint32_t)(wave * (64 * kPass2Items) + it * 64 + lane);
      kv[it] = e < cb.end ? keys[e] : 0xFFFFFFFFu;
    }
    // (one LDS atomic per element: the rows of 64 consecutive pairs of a column are a few short runs of consecutive ty
    //  of unrelated Gaussians -- little to aggregate, and the ballot rounds of a wave-aggregated count cost more)
#pragma unroll
    for (int it = 0; it < kPass2Items; ++it)
      if (kv[it] != 0xFFFFFFFFu) atomicAdd(&h[kv[it] >> 24], 1u);
  }
  __syncthreads();
  hist[(uint64_t)tid * nblk + cb.blk] = h[tid];
}

// per-view outputs of the ty pass (they live in separately allocated per-view state buffers)
struct RowOut {
  uint32_t* point_list[GSR_MAX_BATCH_VIEWS];
  uint32_t* ranges[GSR_MAX_BATCH_VIEWS];
};

// Stable scatter by ty. The first workgroup of every column also writes the ranges of the column's tiles.
__global__ void __launch_bounds__(kSortThreads)
k_row_scatter(const uint32_t* __restrict__ vals_in, const RowOut ro, const uint32_t* __restrict__ colstart, const int gx,
              const int gy, const uint32_t cap, const int nbits, const uint32_t nblk, const uint32_t* __restrict__ hist,
              const uint32_t* __restrict__ totals, size_t ps, size_t ss) {
  vals_in = batch_ptr(vals_in, ss); colstart = batch_ptr(colstart, ps); hist = batch_ptr(hist, ss);
  totals = batch_ptr(totals, ss);
  uint32_t* __restrict__ vals_out = ro.point_list[blockIdx.y];
  uint32_t* __restrict__ ranges = ro.ranges[blockIdx.y];
  __shared__ uint32_t wh[4][kRadix];
  __shared__ uint32_t dbase[kRadix];
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t sh_fb[257];
  __shared__ uint32_t sh_tmp[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ColBlocks cb = col_blocks(colstart, gx, cap, blockIdx.x, sh_fb, sh_tmp);
  if (cb.col == 0xFFFFFFFFu) return;
  const uint32_t blk = cb.blk;
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
  const uint32_t my_hist = hist[(uint64_t)tid * nblk + blk];
  if (blk == sh_fb[cb.col] && tid < gy) {   // tile (ty = tid, tx = col): [first pair, first pair of tx + 1)
    const uint32_t a = dbase[tid] + my_hist;
    const uint32_t b = dbase[tid] + hist[(uint64_t)tid * nblk + sh_fb[cb.col + 1]];
    uint2 rg = make_uint2(a, b);
    if (a == b) rg = make_uint2(0u, 0u);
    reinterpret_cast<uint2*>(ranges)[(uint32_t)tid * (uint32_t)gx + cb.col] = rg;
  }
  if (cb.base >= cb.end) return;
  uint32_t* mywh = wh[wave];
  uint32_t val[kPass2Items];   // ty << 24 | Gaussian index; 0xFFFFFFFF = no element (index 0xFFFFFF never occurs)
  uint32_t rank[kPass2Items];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < kPass2Items; ++it) {
    const uint32_t e = cb.base + (uint32_t)(wave * (64 * kPass2Items) + it * 64 + lane);
    val[it] = e < cb.end ? vals_in[e] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int it = 0; it < kPass2Items; ++it) {
    const bool valid = val[it] != 0xFFFFFFFFu;
    const uint32_t d = val[it] >> 24;
    const unsigned long long m = match_digit_n(d, valid, nbits);
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
    uint32_t run = dbase[tid] + my_hist;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = wh[w][tid];
      wh[w][tid] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kPass2Items; ++it) {
    if (val[it] != 0xFFFFFFFFu) vals_out[wh[wave][val[it] >> 24] + rank[it]] = val[it] & 0xFFFFFFu;
  }
}

// debug / parity: 64-bit keys of the reference formulation rebuilt from the ranges (one workgroup per tile)
__global__ void __launch_bounds__(256)
k_rebuild_keys_ranges(const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const float* __restrict__ splat, uint64_t* __restrict__ keys64) {
  const uint32_t t = blockIdx.x;
  const uint32_t a = ranges[2 * t], b = ranges[2 * t + 1];
  for (uint32_t j = a + threadIdx.x; j < b; j += 256) {
    const uint32_t dbits = __float_as_uint(splat[12 * (size_t)point_list[j] + 6]);
    keys64[j] = ((uint64_t)t << 32) | dbits;
  }
}

}  // namespace

extern "C" uint32_t gsr_num_tiles(int32_t H, int32_t W) {
  return (uint32_t)(((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE));
}
extern "C" uint32_t gsr_num_blocks(int32_t P) { return (uint32_t)((P + 255) / 256); }

// The column path packs (ty, Gaussian index) into one word: grids up to 256 x 256 tiles, fewer than 2^24 Gaussians.
static bool use_columns(int32_t H, int32_t W, int32_t P) {
  return (W + GSR_TILE - 1) / GSR_TILE <= 256 && (H + GSR_TILE - 1) / GSR_TILE <= 256 && P < (1 << 24);
}
bool gsr_uses_columns(const GsrView& v) { return use_columns(v.image_height, v.image_width, v.P); }
static uint32_t col_runs(int32_t P) { return (uint32_t)(((P > 0 ? P : 1) + kColRun - 1) / kColRun); }

// Scratch of the projection stage: depth-sort keys x2 / values x2 (one of them becomes sorted_idx), histograms,
// packed tile rectangles and the per-run column histogram of the column path.
extern "C" size_t gsr_project_scratch_bytes(int32_t P) {
  const uint64_t m = P > 0 ? (uint64_t)P : 1;
  return 5 * align256(m * 4) + sort_hist_bytes(m, kItemsSmall, kOsItemsSmall) + align256(kRadix * 4) +
         align256((size_t)256 * col_runs((int32_t)m) * 4) + align256(256 * 4) + align256(264 * 4) + 256 + 1024;
}

// Scratch of the binning stage: tile keys x2, one value ping buffer, histograms (the column path needs less).
extern "C" size_t gsr_sort_scratch_bytes(uint64_t n, uint32_t n_tiles) {
  (void)n_tiles;
  const uint64_t m = n ? n : 1;
  return 3 * align256(m * 4) + sort_hist_bytes(m, kItemsLarge, kItemsLarge) + align256((size_t)kRadix * 260 * 4) +
         align256(kRadix * 4) + 1024;
}

struct ProjectScratch {
  uint32_t *k0, *k1, *v0, *v1, *hist, *totals, *rects, *hist1, *totals1, *colstart;
  uint64_t* counts;   // [0] = N (pairs), [1] = visible Gaussians
};
static ProjectScratch carve_project(void* scratch, int32_t P) {
  const uint64_t m = P > 0 ? (uint64_t)P : 1;
  char* b = (char*)scratch;
  ProjectScratch s;
  s.k0 = (uint32_t*)b; b += align256(m * 4);
  s.k1 = (uint32_t*)b; b += align256(m * 4);
  s.v0 = (uint32_t*)b; b += align256(m * 4);
  s.v1 = (uint32_t*)b; b += align256(m * 4);
  s.hist = (uint32_t*)b; b += sort_hist_bytes(m, kItemsSmall, kOsItemsSmall);
  s.totals = (uint32_t*)b; b += align256(kRadix * 4);
  s.rects = (uint32_t*)b; b += align256(m * 4);
  s.hist1 = (uint32_t*)b; b += align256((size_t)256 * col_runs((int32_t)m) * 4);
  s.totals1 = (uint32_t*)b; b += align256(256 * 4);
  s.colstart = (uint32_t*)b; b += align256(264 * 4);
  s.counts = (uint64_t*)b;
  return s;
}

uint32_t* gsr_depth_keys(const GsrGeom& geom, int32_t P) { return carve_project(geom.scratch, P).k0; }
uint32_t* gsr_tile_rects(const GsrGeom& geom, int32_t P) { return carve_project(geom.scratch, P).rects; }
// The words K1 clears for the depth sort that follows it (the one-sweep state: digit histograms, tickets, look-back words):
// K1 and the sort are always enqueued as a pair (gsr_forward_project*), so the sort launches no clear of its own -- one
// launch less in front of a latency-bound chain. 0 words when the sort takes the three-kernel passes.
// (Round 4 also folded k_col_plan into the tail of the column scan and k_work_order_fwd into the tail of k_row_scatter --
//  "the workgroup that draws the last ticket does the next kernel's work", release / acquire on the ticket: lists
//  bit-identical, but an agent-scope release writes the XCD's whole L2 back: k_row_scatter 46 -> 86 us per 4-view launch for
//  the 8 us launch it saved, the fused scan + plan 17.8 us against 8.8 + 4.9. Not kept: gpurun_out/r4e, profiles/HISTORY.md.)
uint32_t* gsr_depth_sort_state(const GsrGeom& geom, int32_t P, uint32_t* words) {
  const uint64_t m = P > 0 ? (uint64_t)P : 1;
  *words = m < kOsMaxN ? (uint32_t)os_state_words(m, kOsItemsSmall) : 0u;
  return carve_project(geom.scratch, P).hist;
}
// device words [0] = N (pairs), [1] = number of visible Gaussians; they live in the projection scratch
uint64_t* gsr_pair_counts(const GsrGeom& geom, int32_t P) { return carve_project(geom.scratch, P).counts; }

// After K1 (which wrote the depth keys into scratch.k0): depth sort, depth-ordered counts, scan -> N.
// batch > 1: `batch` views whose projection scratch buffers are `bstride` bytes apart (geom = the first view's) go
// through every launch together (blockIdx.y = view); n_pairs_all (device, may be NULL) receives the N of all views.
// n_pairs_all may be page-locked HOST memory (device-visible): on the column path the first pass of the depth sort stores the
// counts there itself (radix_sort.h, kOsEarlyN).
// early: store N into n_pairs_all from the first pass of the depth sort (radix_sort.h, kOsEarlyN; column path only) instead of
// from k_col_plan. It costs the histogram kernel one more word per key (k_os_hist 17.5 -> 21.9 us per 4-view launch at C3) and
// buys a caller that blocks on the word ~80 us per view: the single-view entry point asks for it, the batched one (whose callers
// are GPU-bound and wait for all views anyway) does not.
int gsr_launch_depth_order(GsrGeom& geom, const GsrView& v, hipStream_t stream, GsrProfile* prof, int batch,
                           size_t bstride, uint64_t* n_pairs_all, bool early) {
  const int32_t P = v.P;
  if (geom.scratch_bytes < gsr_project_scratch_bytes(P) || !geom.scratch) return GSR_ESCRATCH;
  ProjectScratch s = carve_project(geom.scratch, P);
  uint64_t* n_pairs_dev = s.counts;
  uint64_t* n_vis_dev = s.counts + 1;
  const bool columns = use_columns(v.image_height, v.image_width, v.P);
  if (batch > 1 && !columns) return GSR_EINVAL;
  int where;
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SORT);
    // (round 4 measured a distribution sort in its place -- equal-width bins over [min, max] of the depth bits, global
    //  histogram + returning atomics, buckets of whole bins sorted inside LDS: tools/probe/depth_sort_distribution.h, lists
    //  bit-identical -- at 108 us per view against 45 for this one: random-address global atomics run at ~25 G/s on this
    //  part (2 M keys x 3 atomics = 290 us per 4-view step), and two LDS-local radix passes over 2 M pairs cost what two
    //  one-sweep passes cost: profiles/r04_distribution_sort_kernel_stats.txt)
    // (column path: an identity last pass -- the top byte of the depths of one object -- is not copied; k_col_hist / k_emit_cols
    //  take the order from the buffer the flag word names)
    where = radix_sort_u32<kItemsSmall, kOsItemsSmall>(s.k0, s.v0, s.k1, s.v1, nullptr, (uint64_t)P, 32, true, n_vis_dev, s.hist,
                                                       s.totals, stream, batch, bstride, /*state_cleared=*/true,
                                                       /*last_pass_may_skip=*/columns,
                                                       /*early N (kOsEarlyN): the column path's packed rectangles*/
                                                       (columns && early) ? s.rects : nullptr,
                                                       (columns && early) ? n_pairs_all : nullptr);
    geom.sorted_idx = where ? s.v1 : s.v0;
    GSR_HIP(hipGetLastError());
  }
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_SCAN);
    const uint32_t nb = gsr_num_blocks(P);
    if (columns) {
      const int gx = (v.image_width + GSR_TILE - 1) / GSR_TILE;
      uint32_t* rect_sorted = where ? s.k0 : s.k1;   // the key buffer the sort result is NOT in
      const uint32_t nrun = col_runs(P);
      const uint32_t nby = (uint32_t)batch;
      hipLaunchKernelGGL(k_col_hist, dim3((nb + 7u) / 8u * 8u, nby), dim3(256), 0, stream, n_vis_dev, geom.sorted_idx,
                         (const uint32_t*)(where ? s.v0 : s.v1), (const uint32_t*)s.hist, s.rects, rect_sorted, gx, nrun, s.hist1,
                         bstride);
      // (rows of one entry per run of 64 Gaussians: the wide scan walks them in a quarter of the chunks)
      if (nrun > 1024u)
        hipLaunchKernelGGL(k_radix_scan_wide, dim3(gx, nby), dim3(1024), 0, stream, s.hist1, nrun, s.totals1,
                           (const uint64_t*)n_vis_dev, (uint32_t)kColRun, bstride);
      else
        hipLaunchKernelGGL(k_radix_scan, dim3(gx, nby), dim3(256), 0, stream, s.hist1, nrun, s.totals1,
                           (const uint64_t*)n_vis_dev, (uint32_t)kColRun, bstride);
      // (early: the host's copy of N was stored by the first pass of the depth sort, above. k_col_plan must then NOT store it
      //  again -- a caller that polled the early word has moved on, and a late second store could land in the word after the
      //  caller re-armed it for its next call on this stream. ONE store per call, here or there.)
      hipLaunchKernelGGL(k_col_plan, dim3(1, nby), dim3(256), 0, stream, s.totals1, gx, s.colstart, n_pairs_dev, bstride,
                         early ? (uint64_t*)nullptr : n_pairs_all);
    } else {
      hipLaunchKernelGGL(k_sorted_block_sums, dim3(nb), dim3(256), 0, stream, n_vis_dev, geom.sorted_idx,
                         geom.tiles_touched, geom.block_offsets);
      hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, geom.block_offsets, nb, n_pairs_dev);
    }
    GSR_HIP(hipGetLastError());
  }
  return GSR_OK;
}

// Column path of gsr_launch_binning for n views whose projection scratch buffers (ps) and sort scratch buffers (ss) are
// equally spaced (n == 1: any buffers). geoms / bs point at the first view's structs.
static int launch_binning_columns(int n, const GsrView& v, const GsrGeom* geoms, uint64_t cap64, GsrBinning* bs, size_t ps,
                                  size_t ss, hipStream_t stream, GsrProfile* prof) {
  const GsrGeom& geom = geoms[0];
  const int gx = (v.image_width + GSR_TILE - 1) / GSR_TILE, gy = (v.image_height + GSR_TILE - 1) / GSR_TILE;
  const uint32_t cap = cap64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap64;
  ProjectScratch s = carve_project(geom.scratch, v.P);
  const uint32_t* rect_sorted = (geom.sorted_idx == s.v0) ? s.k1 : s.k0;
  const uint64_t* n_dev_vis = s.counts + 1;
  const uint32_t nrun = col_runs(v.P);
  const uint32_t nblk = (uint32_t)((cap64 + kPass2Block - 1) / kPass2Block) + (uint32_t)gx + 2;
  char* base = (char*)bs[0].scratch;
  uint32_t* vals1 = (uint32_t*)base; base += align256(cap64 * 4);
  uint32_t* hist = (uint32_t*)base; base += align256((size_t)kRadix * nblk * 4);
  uint32_t* totals = (uint32_t*)base;
  RowOut ro = RowOut{};
  for (int k = 0; k < n; ++k) { ro.point_list[k] = bs[k].point_list; ro.ranges[k] = bs[k].ranges; }
  const uint32_t ny = (uint32_t)n;
  int nbits = 1;
  while ((1 << nbits) < gy) ++nbits;
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_DUPLICATE);
    int nbits_x = 1;
    while ((1 << nbits_x) < gx) ++nbits_x;
    const uint32_t ngrp = (nrun + kEmitWaves - 1) / kEmitWaves;
    hipLaunchKernelGGL(k_emit_cols, dim3((ngrp + 7u) / 8u * 8u, ny), dim3(64 * kEmitWaves), 0, stream, n_dev_vis, gx, nbits_x, rect_sorted,
                       geom.sorted_idx, (const uint32_t*)((geom.sorted_idx == s.v0) ? s.v1 : s.v0), (const uint32_t*)s.hist,
                       s.hist1, nrun, s.colstart, cap, vals1, ps, ss);
    GSR_HIP(hipGetLastError());
  }
  {
    // (round 3 also built this pass as ONE launch with decoupled look-back over the column-aligned blocks, row starts from
    //  per-row totals left by k_col_hist: bit-identical lists, 87 us per 4-view launch against 84 for these three -- the
    //  816 blocks of a view are all resident at once, so the look-back chain, not the data, sets the time; not kept)
    GsrStageTimer t(prof, stream, GSR_STAGE_SORT);
    hipLaunchKernelGGL(k_row_hist, dim3((nblk + 7u) / 8u * 8u, ny), dim3(kSortThreads), 0, stream, vals1, s.colstart, gx, cap, nbits, nblk,
                       hist, ps, ss);
    if (nblk > 1024u)
      hipLaunchKernelGGL(k_radix_scan_wide, dim3(kRadix, ny), dim3(1024), 0, stream, hist, nblk, totals, (const uint64_t*)nullptr,
                         1u, ss);
    else
      hipLaunchKernelGGL(k_radix_scan, dim3(kRadix, ny), dim3(256), 0, stream, hist, nblk, totals, (const uint64_t*)nullptr,
                         1u, ss);
    hipLaunchKernelGGL(k_row_scatter, dim3((nblk + 7u) / 8u * 8u, ny), dim3(kSortThreads), 0, stream, vals1, ro, s.colstart, gx, gy, cap,
                       nbits, nblk, hist, totals, ps, ss);
    GSR_HIP(hipGetLastError());
  }
  for (int k = 0; k < n; ++k) {
    if (!bs[k].keys_sorted) continue;
    GsrStageTimer t(prof, stream, GSR_STAGE_RANGES);
    hipLaunchKernelGGL(k_rebuild_keys_ranges, dim3(gx * gy), dim3(256), 0, stream, bs[k].ranges, bs[k].point_list,
                       geoms[k].splat, bs[k].keys_sorted);
    GSR_HIP(hipGetLastError());
  }
  return GSR_OK;
}

// Binning of n views in shared launches; returns GSR_EINVAL (nothing enqueued) if the views' buffers do not allow it
// (column path only, equally spaced projection / sort scratch buffers, enough sort scratch): the caller then loops.
int gsr_launch_binning_batch(int n, const GsrView* views, const GsrGeom* geoms, uint64_t cap, GsrBinning* bs,
                             hipStream_t stream, GsrProfile* prof) {
  const GsrView& v = views[0];
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  if (n < 2 || cap == 0 || v.P == 0 || !use_columns(v.image_height, v.image_width, v.P)) return GSR_EINVAL;
  const size_t ps = (size_t)((char*)geoms[1].scratch - (char*)geoms[0].scratch);
  const size_t ss = (size_t)((char*)bs[1].scratch - (char*)bs[0].scratch);
  const size_t need = gsr_sort_scratch_bytes(cap, tiles);
  for (int k = 0; k < n; ++k) {
    if (!geoms[k].sorted_idx || !bs[k].scratch || bs[k].scratch_bytes < need) return GSR_EINVAL;
    if ((char*)geoms[k].scratch != (char*)geoms[0].scratch + (size_t)k * ps) return GSR_EINVAL;
    if ((char*)bs[k].scratch != (char*)bs[0].scratch + (size_t)k * ss) return GSR_EINVAL;
    if ((char*)geoms[k].sorted_idx != (char*)geoms[0].sorted_idx + (size_t)k * ps) return GSR_EINVAL;
  }
  if (ss < need || (ss & 255u) || (ps & 255u)) return GSR_EINVAL;
  return launch_binning_columns(n, v, geoms, cap, bs, ps, ss, stream, prof);
}

// Emits, tile-sorts and ranges. `cap` = pairs the buffers hold; n_dev (may be NULL = exactly cap pairs) is the
// true count on the device. On return binning.point_list holds the sorted values.
int gsr_launch_binning(const GsrView& v, const GsrGeom& geom, uint64_t cap, const uint64_t* n_dev,
                       const uint64_t* n_dev_vis, GsrBinning& b, hipStream_t stream, GsrProfile* prof) {
  const uint32_t tiles = gsr_num_tiles(v.image_height, v.image_width);
  if (cap == 0 || v.P == 0) {
    GSR_HIP(gsr_zero_async(b.ranges, (size_t)tiles * 2 * sizeof(uint32_t), stream));
    return GSR_OK;
  }
  if (b.scratch_bytes < gsr_sort_scratch_bytes(cap, tiles) || !b.scratch) return GSR_ESCRATCH;
  if (!geom.sorted_idx) return GSR_EINVAL;
  if (use_columns(v.image_height, v.image_width, v.P)) {
    (void)n_dev_vis;
    return launch_binning_columns(1, v, &geom, cap, &b, 0, 0, stream, prof);
  }
  char* base = (char*)b.scratch;
  uint32_t* keys_a = (uint32_t*)base; base += align256(cap * 4);
  uint32_t* keys_b = (uint32_t*)base; base += align256(cap * 4);
  uint32_t* vals_t = (uint32_t*)base; base += align256(cap * 4);
  uint32_t* hist = (uint32_t*)base; base += sort_hist_bytes(cap, kItemsLarge, kItemsLarge);
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
    const int where = radix_sort_u32<kItemsLarge, kItemsLarge>(keys_a, va, keys_b, vb, n_dev, cap, tile_bits, false, nullptr, hist,
                                                   totals, stream);
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
