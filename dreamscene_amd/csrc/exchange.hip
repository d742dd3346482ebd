// exchange.hip -- device side of the sparse gradient-exchange formats (include/gsrast.h, GsrRowSet).
//
// Views are sharded one per GPU and the only exchange of a step is the sum of the per-rank parameter gradients
// (training/object_trainer.py:302-382 sums the C_batch_size views of a step in one process). Behind the opaque front layers of
// an object nothing receives a gradient: 16 % of a rank's rows are non-zero at C3 (K8 leaves them as a bitmap,
// GsrGrads.reached_mask), so the row formats of multiview.GradExchange move (index, row) messages instead of the whole arena.
// Their device side was torch index arithmetic over five tensors: 0.5 - 0.7 ms per step at C3 for 18 MB of rows
// (profiles/r05_exchange_device_c3.json) -- as much as the step itself. These three kernels are that device side at memory speed:
//   gsr_rows_pack    bitmap -> ascending row indices + the rows, gathered from the planar regions into one [n, F] message
//   gsr_rows_unpack  message -> regions: add (rank-order accumulation) or store (disjoint owners), optionally marking a bitmap
// A "row set" is any table whose logical row i is the concatenation of slices of up to 8 strided regions: the planar gradient
// arena (means3D | scales | rotations | opacities | the ACTIVE SH columns of shs) as well as a plain row-major buffer.
#include "gsr_common.h"

namespace {

struct Regions {
  float* ptr[GSR_ROWSET_MAX_REGIONS];
  int32_t width[GSR_ROWSET_MAX_REGIONS], stride[GSR_ROWSET_MAX_REGIONS], first[GSR_ROWSET_MAX_REGIONS + 1];
  int32_t n, F, rows;
};

__host__ inline int make_regions(const GsrRowSet* rs, Regions& r) {
  if (!rs || rs->n_regions < 1 || rs->n_regions > GSR_ROWSET_MAX_REGIONS || rs->rows < 0) return GSR_EINVAL;
  r.n = rs->n_regions;
  r.rows = rs->rows;
  int f = 0;
  for (int k = 0; k < r.n; ++k) {
    const GsrRowRegion& g = rs->regions[k];
    if (!g.ptr || g.width < 1 || g.stride < g.width) return GSR_EINVAL;
    if ((int64_t)rs->rows * (int64_t)g.stride > (int64_t)1 << 40) return GSR_EINVAL;   // (element offsets stay far inside int64)
    r.ptr[k] = g.ptr; r.width[k] = g.width; r.stride[k] = g.stride; r.first[k] = f;
    f += g.width;
  }
  r.first[r.n] = f;
  r.F = f;
  return f <= 1024 ? GSR_OK : GSR_EINVAL;
}

// address of element f of logical row i. (A chain of selects over compile-time indices: a run-time index into the by-value
// table would send it through scratch memory -- the first build did, and packed 18 MB in 85 us.)
__device__ __forceinline__ float* row_elem(const Regions& r, int64_t i, int f) {
  float* p = r.ptr[0];
  int st = r.stride[0], f0 = 0;
#pragma unroll
  for (int j = 1; j < GSR_ROWSET_MAX_REGIONS; ++j) {
    const bool in = (j < r.n) && (f >= r.first[j]);
    p = in ? r.ptr[j] : p;
    st = in ? r.stride[j] : st;
    f0 = in ? r.first[j] : f0;
  }
  return p + i * (int64_t)st + (f - f0);
}

// per 64-row word: number of set bits in front of it (exclusive scan), total -> *count. One workgroup.
__global__ void __launch_bounds__(1024)
k_rows_offsets(const unsigned long long* __restrict__ mask, const int32_t n_words, const int32_t rows, uint32_t* __restrict__ offs,
               uint32_t* __restrict__ count) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_words; base += 1024) {
    const int w = base + tid;
    unsigned long long m = w < n_words ? mask[w] : 0ull;
    if (w == n_words - 1 && (rows & 63)) m &= (1ull << (rows & 63)) - 1ull;      // bits beyond the last row do not count
    const uint32_t x = (uint32_t)__popcll(m);
    uint32_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) woff += (k < wave) ? wave_tot[k] : 0u;
    const uint32_t carry = carry_s;
    if (w < n_words) offs[w] = carry + woff + inc - x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) *count = carry_s;
}

// One wave per word of 64 rows: the set rows' indices in ascending order, then every set row copied into the message by
// the whole wave (lane f copies element f: a row of <= 64 floats per step; wider rows in chunks of 64).
__global__ void __launch_bounds__(256)
k_rows_pack(const Regions r, const unsigned long long* __restrict__ mask, const int32_t n_words, const int32_t rows,
            const uint32_t* __restrict__ offs, uint32_t* __restrict__ idx, float* __restrict__ out, const uint32_t cap) {
  const int lane = threadIdx.x & 63;
  const int w = (int)(blockIdx.x * 4u + (threadIdx.x >> 6));
  if (w >= n_words) return;
  unsigned long long m = mask[w];
  if (w == n_words - 1 && (rows & 63)) m &= (1ull << (rows & 63)) - 1ull;
  if (m == 0ull) return;
  const uint32_t o0 = offs[w];
  __shared__ uint8_t bitpos[4][64];         // per wave: the set rows of its word, in order
  uint8_t* mybits = bitpos[threadIdx.x >> 6];
  if ((m >> lane) & 1ull) {
    const uint32_t k = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    mybits[k] = (uint8_t)lane;
    if (o0 + k < cap) idx[o0 + k] = (uint32_t)(w * 64 + lane);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // the word's nset x F elements dealt to the 64 lanes: every load instruction has all lanes busy on independent addresses
  const int nset = (int)__popcll(m), total = nset * r.F;
  for (int e = lane; e < total; e += 64) {
    const int k = e / r.F, f = e - k * r.F;
    const uint32_t pos = o0 + (uint32_t)k;
    if (pos < cap) out[(size_t)pos * r.F + f] = *row_elem(r, (int64_t)w * 64 + mybits[k], f);
  }
}

// message -> regions. MODE 0: add, 1: store. One thread per (message row, element).
// PRECONDITION (both modes): the indices of ONE message are distinct -- the add is a plain read-modify-write, two entries on one
// row would lose an update. Every message of this library is (gsr_rows_pack emits each set bit once, ascending); a caller
// assembling its own must not pad with repeated indices (pad with an index outside [row_base, row_base + rows): dropped).
template <int MODE>
__global__ void __launch_bounds__(256)
k_rows_unpack(const Regions r, const uint32_t* __restrict__ idx, const float* __restrict__ in, const uint32_t n,
              const int64_t row_base, unsigned long long* __restrict__ touched) {
  const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint32_t j = (uint32_t)(t / (uint32_t)r.F);
  if (j >= n) return;
  const int f = (int)(t - (uint64_t)j * (uint32_t)r.F);
  const int64_t i = (int64_t)idx[j] - row_base;
  if (i < 0 || i >= (int64_t)r.rows) return;      // (an index outside the set is dropped, never written through)
  float* p = row_elem(r, i, f);
  const float v = in[(size_t)j * r.F + f];
  if (MODE == 0) *p += v; else *p = v;
  if (touched && f == 0) atomicOr(touched + (i >> 6), 1ull << (i & 63));
}

// ------------------------------------------------------------------------------------------------ self-describing row messages
// The (index, row) messages above need the row COUNT on the host before anything can be sent (it sizes the wire buffers): one
// blocking read per pack, a second one for the ranks' counts -- ten dependent host-driven steps, 0.34 ms at C3 for kernels that take
// 66 us (profiles/r05_exchange_device_c3.json). A row MESSAGE has a fixed, speculated capacity and describes itself:
//   [ header 256 B: count, cap, rows, F | bitmap u64[n_words] | offs u32[n_words] (rows in front of each word) | rows f32[cap][F] ]
// so that ONE launch packs it (no count leaves the device), ONE fixed-size all-gather moves the W messages, and ONE launch applies
// them: for every row any rank sent, the sum of the ranks' contributions in RANK ORDER -- ((g_0 + g_1) + g_2) + ... over the ranks
// that hold the row -- is STORED into the set (bit-identical on every rank; rows nobody sent stay as they are: zero). No index
// list travels (bitmap + offs locate a row: 12 B per 64 rows instead of 4 B per row), nothing is zero-filled, and no read-modify-
// write touches the arena. A message that does not fit its capacity says so in its header: the apply kernel then changes nothing
// and reports the largest count (the caller repeats the step with more room -- the arena still holds its own gradients).
constexpr uint32_t kMsgHdrBytes = 256;
constexpr int kMsgWordsPerBlock = 16;       // k_msg_pack: 4 waves x 4 words of 64 rows

struct MsgLayout { size_t bitmap, offs, rows, total; };
__host__ __device__ inline MsgLayout msg_layout(int32_t n_rows, int32_t F, uint32_t cap) {
  const size_t words = ((size_t)(n_rows > 0 ? n_rows : 1) + 63) / 64;
  MsgLayout l;
  l.bitmap = kMsgHdrBytes;
  l.offs = l.bitmap + ((words * 8 + 255) & ~(size_t)255);
  l.rows = l.offs + ((words * 4 + 255) & ~(size_t)255);
  l.total = l.rows + (((size_t)cap * (size_t)F * 4 + 255) & ~(size_t)255);
  return l;
}

__device__ __forceinline__ unsigned long long msg_word(const unsigned long long* mask, int w, int n_words, int rows) {
  unsigned long long m = mask[w];
  if (w == n_words - 1 && (rows & 63)) m &= (1ull << (rows & 63)) - 1ull;      // bits beyond the last row do not count
  return m;
}

// Workgroup b packs words [16 b, 16 b + 16): it counts the set bits in FRONT of its range itself (a strided pass over at most
// n_words words of the L2-resident bitmap: 62 KB at 500 k rows -- no scan launch, no look-back chain), then every wave copies the
// rows of its four words. The last workgroup knows the total and writes the header.
__global__ void __launch_bounds__(256)
k_msg_pack(const Regions r, const unsigned long long* __restrict__ mask, const int32_t n_words, unsigned char* __restrict__ msg,
           const MsgLayout lay, const uint32_t cap) {
  __shared__ uint32_t red[4];
  __shared__ uint32_t wpre[kMsgWordsPerBlock + 1];
  __shared__ uint8_t bitpos[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int first = (int)blockIdx.x * kMsgWordsPerBlock;
  uint32_t c = 0;
  for (int w = tid; w < first; w += 256) c += (uint32_t)__popcll(mask[w]);     // (the masked last word is never in front of a range)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
  if (lane == 0) red[wave] = c;
  if (tid < kMsgWordsPerBlock) {
    const int w = first + tid;
    wpre[tid + 1] = w < n_words ? (uint32_t)__popcll(msg_word(mask, w, n_words, r.rows)) : 0u;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = (red[0] + red[1]) + (red[2] + red[3]);
    wpre[0] = run;
    for (int k = 1; k <= kMsgWordsPerBlock; ++k) { const uint32_t x = wpre[k]; wpre[k] = run; run += x; }
    // (now wpre[j + 1] = rows in front of word first + j, and `run` = rows up to the end of this range)
    if (blockIdx.x == gridDim.x - 1) {
      uint32_t* hdr = reinterpret_cast<uint32_t*>(msg);
      hdr[0] = run; hdr[1] = cap; hdr[2] = (uint32_t)r.rows; hdr[3] = (uint32_t)r.F;
    }
  }
  __syncthreads();
  unsigned long long* bm = reinterpret_cast<unsigned long long*>(msg + lay.bitmap);
  uint32_t* offs = reinterpret_cast<uint32_t*>(msg + lay.offs);
  float* out = reinterpret_cast<float*>(msg + lay.rows);
  uint8_t* mybits = bitpos[wave];
  for (int k = 0; k < 4; ++k) {
    const int j = wave * 4 + k, w = first + j;
    if (w >= n_words) break;
    const unsigned long long m = msg_word(mask, w, n_words, r.rows);
    const uint32_t o0 = wpre[j + 1];          // after the scan above wpre[j + 1] = rows in front of word first + j
    if (lane == 0) { bm[w] = m; offs[w] = o0; }
    if (m == 0ull) continue;
    if ((m >> lane) & 1ull) mybits[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int nset = (int)__popcll(m), total = nset * r.F;
    for (int e = lane; e < total; e += 64) {
      const int q = e / r.F, f = e - q * r.F;
      const uint32_t pos = o0 + (uint32_t)q;
      if (pos < cap) out[(size_t)pos * r.F + f] = *row_elem(r, (int64_t)w * 64 + mybits[q], f);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// One wave per word of 64 rows: union of the W bitmaps; every row of the union = the ranks' rows added in rank order, stored.
// *status (one u64, may be page-locked host memory) = largest count << 32 | 1 (applied) or 2 (some message overflowed its
// capacity, or does not match this call's shape: NOTHING applied), stored by the first workgroup before any row is touched.
template <int MAXW>
__global__ void __launch_bounds__(256)
k_msg_apply(const Regions r, const unsigned char* __restrict__ msgs, const size_t msg_stride, const int W, const int32_t n_words,
            const MsgLayout lay, const uint32_t cap, unsigned long long* __restrict__ status,
            unsigned long long* __restrict__ touched) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
  uint32_t worst = 0;
  bool bad = false;
#pragma unroll
  for (int q = 0; q < MAXW; ++q) {
    if (q < W) {
      const uint32_t* hdr = reinterpret_cast<const uint32_t*>(msgs + (size_t)q * msg_stride);
      const uint32_t n = hdr[0];
      worst = n > worst ? n : worst;
      bad = bad || n > cap || hdr[1] != cap || hdr[2] != (uint32_t)r.rows || hdr[3] != (uint32_t)r.F;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && status)
    __hip_atomic_store(status, ((unsigned long long)worst << 32) | (bad ? 2ull : 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (bad || w >= n_words) return;
  unsigned long long m[MAXW], U = 0ull;
  uint32_t o[MAXW];
#pragma unroll
  for (int q = 0; q < MAXW; ++q) {
    m[q] = 0ull; o[q] = 0u;
    if (q < W) {
      const unsigned char* base = msgs + (size_t)q * msg_stride;
      m[q] = reinterpret_cast<const unsigned long long*>(base + lay.bitmap)[w];
      o[q] = reinterpret_cast<const uint32_t*>(base + lay.offs)[w];
      U |= m[q];
    }
  }
  if (touched && lane == 0) touched[w] = U;
  while (U) {
    const int b = __builtin_ctzll(U);
    U &= U - 1ull;
    const unsigned long long below = (1ull << b) - 1ull;
    for (int f = lane; f < r.F; f += 64) {
      float v[MAXW];
#pragma unroll
      for (int q = 0; q < MAXW; ++q) {          // all the loads of the row first (independent), then the adds in rank order
        v[q] = 0.f;
        if (q < W && ((m[q] >> b) & 1ull)) {
          const float* rows = reinterpret_cast<const float*>(msgs + (size_t)q * msg_stride + lay.rows);
          v[q] = rows[(size_t)(o[q] + (uint32_t)__popcll(m[q] & below)) * r.F + f];
        }
      }
      float acc = 0.f;
      bool any = false;
#pragma unroll
      for (int q = 0; q < MAXW; ++q) {
        if (q < W && ((m[q] >> b) & 1ull)) { acc = any ? __fadd_rn(acc, v[q]) : v[q]; any = true; }
      }
      *row_elem(r, (int64_t)w * 64 + b, f) = acc;
    }
  }
}

}  // namespace

// out[i] = ((s_0[i] + s_1[i]) + s_2[i]) + ... : the local sum of the `direct` exchange, slices in rank order (the association every
// rank uses for its slice, so the replicas stay bit-identical), one pass over the W slices instead of W - 1 read-modify-writes.
template <typename V>
__global__ void __launch_bounds__(256) k_sum_slices(const V* in, int W, size_t stride, size_t n, V* out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    V a = in[i];
    for (int w = 1; w < W; ++w) {
      const V b = in[(size_t)w * stride + i];
      if constexpr (sizeof(V) == 16) {
        a.x = __fadd_rn(a.x, b.x); a.y = __fadd_rn(a.y, b.y); a.z = __fadd_rn(a.z, b.z); a.w = __fadd_rn(a.w, b.w);
      } else {
        a = __fadd_rn(a, b);
      }
    }
    out[i] = a;
  }
}

extern "C" {

int gsr_sum_slices(const float* slices, int32_t n_slices, uint64_t slice_floats, uint64_t stride_floats, float* out,
                   void* stream_) {
  if (!slices || !out || n_slices < 1 || stride_floats < slice_floats) return GSR_EINVAL;
  if (slice_floats == 0) return GSR_OK;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(out);
  const bool vec = ((reinterpret_cast<uintptr_t>(slices) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0 &&
                   (slice_floats & 3u) == 0 && (stride_floats & 3u) == 0;
  const uint64_t n = vec ? slice_floats / 4 : slice_floats;
  const uint64_t blocks = (n + 255) / 256;
  const dim3 grid((uint32_t)(blocks < 16384 ? blocks : 16384));
  if (vec)
    hipLaunchKernelGGL(k_sum_slices<float4>, grid, dim3(256), 0, stream, reinterpret_cast<const float4*>(slices), n_slices,
                       (size_t)(stride_floats / 4), (size_t)n, reinterpret_cast<float4*>(out));
  else
    hipLaunchKernelGGL(k_sum_slices<float>, grid, dim3(256), 0, stream, slices, n_slices, (size_t)stride_floats, (size_t)n, out);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

size_t gsr_rows_scratch_bytes(int32_t rows) {
  const size_t words = ((size_t)(rows > 0 ? rows : 1) + 63) / 64;
  return ((words * 4 + 255) & ~(size_t)255) + 256;
}

int gsr_rows_pack(const GsrRowSet* rs, const uint64_t* mask, uint32_t* idx, float* rows_out, uint32_t cap, uint32_t* count,
                  void* scratch, size_t scratch_bytes, void* stream_) {
  Regions r;
  const int rc = make_regions(rs, r);
  if (rc) return rc;
  if (!mask || !idx || !rows_out || !count || !scratch || (reinterpret_cast<uintptr_t>(mask) & 7u)) return GSR_EINVAL;
  if (scratch_bytes < gsr_rows_scratch_bytes(rs->rows)) return GSR_ESCRATCH;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(idx);
  const int32_t n_words = (rs->rows + 63) / 64;
  uint32_t* offs = reinterpret_cast<uint32_t*>(scratch);
  if (n_words == 0) return gsr_zero_async(count, sizeof(uint32_t), stream) == hipSuccess ? GSR_OK : GSR_EHIP;
  hipLaunchKernelGGL(k_rows_offsets, dim3(1), dim3(1024), 0, stream, reinterpret_cast<const unsigned long long*>(mask), n_words,
                     rs->rows, offs, count);
  hipLaunchKernelGGL(k_rows_pack, dim3((uint32_t)(n_words + 3) / 4u), dim3(256), 0, stream, r,
                     reinterpret_cast<const unsigned long long*>(mask), n_words, rs->rows, (const uint32_t*)offs, idx, rows_out, cap);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

int gsr_rows_unpack(const GsrRowSet* rs, const uint32_t* idx, const float* rows_in, uint32_t n, int64_t row_base, int32_t mode,
                    uint64_t* touched, void* stream_) {
  Regions r;
  const int rc = make_regions(rs, r);
  if (rc) return rc;
  if (mode != 0 && mode != 1) return GSR_EINVAL;
  if (n == 0) return GSR_OK;
  if (!idx || !rows_in || (touched && (reinterpret_cast<uintptr_t>(touched) & 7u))) return GSR_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(idx);
  const uint64_t total = (uint64_t)n * (uint32_t)r.F;
  if ((total + 255) / 256 > 0x7FFFFFFFull) return GSR_EINVAL;     // (one thread per element: n * F up to 2^39)
  const dim3 grid((uint32_t)((total + 255) / 256));
  if (mode == 0)
    hipLaunchKernelGGL(k_rows_unpack<0>, grid, dim3(256), 0, stream, r, idx, rows_in, n, row_base,
                       reinterpret_cast<unsigned long long*>(touched));
  else
    hipLaunchKernelGGL(k_rows_unpack<1>, grid, dim3(256), 0, stream, r, idx, rows_in, n, row_base,
                       reinterpret_cast<unsigned long long*>(touched));
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

size_t gsr_rowmsg_bytes(int32_t rows, int32_t row_floats, uint32_t cap) {
  return msg_layout(rows, row_floats > 0 ? row_floats : 1, cap).total;
}

int gsr_rowmsg_pack(const GsrRowSet* rs, const uint64_t* mask, void* msg, uint32_t cap, void* stream_) {
  Regions r;
  const int rc = make_regions(rs, r);
  if (rc) return rc;
  if (!mask || !msg || (reinterpret_cast<uintptr_t>(mask) & 7u) || (reinterpret_cast<uintptr_t>(msg) & 255u)) return GSR_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(msg);
  const int32_t n_words = (rs->rows + 63) / 64;
  const MsgLayout lay = msg_layout(rs->rows, r.F, cap);
  const uint32_t blocks = (uint32_t)((n_words + kMsgWordsPerBlock - 1) / kMsgWordsPerBlock);
  hipLaunchKernelGGL(k_msg_pack, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, r, reinterpret_cast<const unsigned long long*>(mask),
                     n_words, reinterpret_cast<unsigned char*>(msg), lay, cap);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

int gsr_rowmsg_apply(const GsrRowSet* rs, const void* msgs, uint64_t msg_stride, int32_t n_msgs, uint32_t cap, uint64_t* status,
                     uint64_t* touched, void* stream_) {
  Regions r;
  const int rc = make_regions(rs, r);
  if (rc) return rc;
  if (!msgs || n_msgs < 1 || n_msgs > 16 || (reinterpret_cast<uintptr_t>(msgs) & 255u) || (msg_stride & 255u)) return GSR_EINVAL;
  if ((touched && (reinterpret_cast<uintptr_t>(touched) & 7u)) || (status && (reinterpret_cast<uintptr_t>(status) & 7u))) return GSR_EINVAL;
  const MsgLayout lay = msg_layout(rs->rows, r.F, cap);
  if (msg_stride < lay.total) return GSR_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(msgs);
  const int32_t n_words = (rs->rows + 63) / 64;
  const dim3 grid((uint32_t)((n_words + 3) / 4 > 0 ? (n_words + 3) / 4 : 1));
  const unsigned char* m = reinterpret_cast<const unsigned char*>(msgs);
  if (n_msgs <= 8)
    hipLaunchKernelGGL(k_msg_apply<8>, grid, dim3(256), 0, stream, r, m, (size_t)msg_stride, n_msgs, n_words, lay, cap,
                       reinterpret_cast<unsigned long long*>(status), reinterpret_cast<unsigned long long*>(touched));
  else
    hipLaunchKernelGGL(k_msg_apply<16>, grid, dim3(256), 0, stream, r, m, (size_t)msg_stride, n_msgs, n_words, lay, cap,
                       reinterpret_cast<unsigned long long*>(status), reinterpret_cast<unsigned long long*>(touched));
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

}  // extern "C"
