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
constexpr int kMsgWordsPerBlock = 4;        // k_msg_pack: one word of 64 rows per wave

struct MsgLayout { size_t bitmap, offs, rows, total; };
__host__ __device__ inline MsgLayout msg_layout(int32_t n_rows, int32_t F, uint32_t cap) {
  const size_t words = ((size_t)(n_rows > 0 ? n_rows : 1) + 63) / 64;
  MsgLayout l;
  l.bitmap = kMsgHdrBytes;
  l.offs = l.bitmap + ((words * 8 + 255) & ~(size_t)255);
  l.rows = l.offs + ((words * 4 + 255) & ~(size_t)255);
  l.total = l.rows + ((((size_t)cap * (size_t)F * 4 > 16 ? (size_t)cap * (size_t)F * 4 : (size_t)16) + 255) & ~(size_t)255);
  return l;
}

__device__ __forceinline__ unsigned long long msg_word(const unsigned long long* mask, int w, int n_words, int rows) {
  unsigned long long m = mask[w];
  if (w == n_words - 1 && (rows & 63)) m &= (1ull << (rows & 63)) - 1ull;      // bits beyond the last row do not count
  return m;
}

// Lane -> (row slot, chunk) of a wave-wide copy of rows of F floats. A CHUNK is VW consecutive floats of a message row: VW = 4 for
// F >= 4 -- one dwordx4 per lane on the message side (message rows are F floats apart, 4-byte aligned: gfx950 takes unaligned
// dwordx4) --, VW = 1 for narrower rows. C = ceil(F / VW) chunks per row; the last chunk of a row whose F is no multiple of VW
// starts at F - VW, i.e. it overlaps its neighbour (the doubly covered floats are computed identically by both lanes; the arena
// side skips them). C <= 64: 64 / C rows per instruction, lane = slot * C + chunk -- the chunk a lane handles never changes, so the
// regions its floats live in (base pointer, stride) are selected ONCE per wave (the second build re-selected them per element from
// the by-value region table and spent its time on the scalar reloads of that table, 55 / 106 us for pack / apply at C3);
// C > 64: one row per instruction, chunks in groups of 64.
// Round 6, fourth -> fifth build: with one FLOAT per lane the apply kernel issued W dword loads per row of the union whatever the
// messages held -- 85 us at C3 (148 MB of rows) and 81 us at C4 (75 MB): bound by the number of memory instructions, not bytes.
template <int VW> struct __attribute__((aligned(4))) Vec { float v[VW]; };
struct ChunkMap {
  int C;           // chunks per row
  int slots;       // rows per wave instruction
  int slot, c;     // this lane's row slot and chunk (group 0)
  bool active;
};
template <int VW>
__device__ __forceinline__ ChunkMap chunk_map(int F, int lane, uint32_t c_magic) {
  ChunkMap cm;
  cm.C = (F + VW - 1) / VW;
  if (cm.C <= 64) {
    cm.slots = 64 / cm.C;
    cm.slot = c_magic ? (int)__umulhi((uint32_t)lane, c_magic) : lane;
    cm.c = lane - cm.slot * cm.C;
    cm.active = cm.slot < cm.slots;
  } else {
    cm.slots = 1; cm.slot = 0; cm.c = lane; cm.active = true;
  }
  return cm;
}
// first float of chunk c
template <int VW>
__device__ __forceinline__ int chunk_start(int F, int c) { return min(c * VW, F - VW); }
struct ElemRef { float* base; int stride; };       // element f of row i = base[i * stride]
__device__ __forceinline__ ElemRef elem_ref(const Regions& r, int f) {
  float* p = r.ptr[0];
  int st = r.stride[0], f0 = 0;
#pragma unroll
  for (int j = 1; j < GSR_ROWSET_MAX_REGIONS; ++j) {
    const bool in = (j < r.n) && (f >= r.first[j]);
    p = in ? r.ptr[j] : p;
    st = in ? r.stride[j] : st;
    f0 = in ? r.first[j] : f0;
  }
  ElemRef e;
  e.base = p + (f - f0);
  e.stride = st;
  return e;
}

// SLICES (the sparse reduce-scatter): a row set of `rows` rows cut into slices of slice_rows rows (a multiple of 64), slice y =
// rows [y slice_rows, min(rows, (y + 1) slice_rows)); message y describes slice y with slice-local row numbers. slice_rows = 0:
// one message for the whole set.
__host__ __device__ inline int32_t slice_len(int32_t rows, int32_t slice_rows, int y) {
  if (slice_rows <= 0) return rows;
  const int64_t left = (int64_t)rows - (int64_t)y * slice_rows;
  return left <= 0 ? 0 : (left < slice_rows ? (int32_t)left : slice_rows);
}

// Workgroup b packs words [4 b, 4 b + 4) of slice blockIdx.y, one word of 64 rows per wave: it counts the set bits in FRONT of its
// range itself (a strided pass over at most n_words words of the L2-resident bitmap: 62 KB at 500 k rows -- no scan launch, no
// look-back chain), then every wave copies the rows of its word, four row-instructions in flight. The last workgroup writes the header.
template <int VW>
__global__ void __launch_bounds__(256)
k_msg_pack(const Regions r, const unsigned long long* __restrict__ mask, unsigned char* __restrict__ msg, const size_t msg_stride,
           const int32_t slice_rows, const MsgLayout lay, const uint32_t cap, const uint32_t c_magic) {
  __shared__ uint32_t red[4];
  __shared__ uint8_t bitpos[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y = (int)blockIdx.y;
  const int64_t row0 = (int64_t)y * (slice_rows > 0 ? slice_rows : 0);
  const int32_t rows_here = slice_len(r.rows, slice_rows, y);
  const int32_t n_words = (rows_here + 63) / 64;
  mask += row0 >> 6;
  msg += (size_t)y * msg_stride;
  const int first = min((int)blockIdx.x * kMsgWordsPerBlock, n_words);
  uint32_t c = 0;
  for (int w = tid; w < first; w += 256) c += (uint32_t)__popcll(mask[w]);     // (the masked last word is never in front of a range)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
  if (lane == 0) red[wave] = c;
  __syncthreads();
  uint32_t o0 = (red[0] + red[1]) + (red[2] + red[3]);
  uint32_t range_total = 0;
  unsigned long long m = 0ull;
#pragma unroll
  for (int j = 0; j < kMsgWordsPerBlock; ++j) {
    const int wj = first + j;
    const unsigned long long mj = wj < n_words ? msg_word(mask, wj, n_words, rows_here) : 0ull;
    const uint32_t pj = (uint32_t)__popcll(mj);
    if (j < wave) o0 += pj;
    if (j == wave) m = mj;
    range_total += pj;
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    uint32_t* hdr = reinterpret_cast<uint32_t*>(msg);
    hdr[0] = (red[0] + red[1]) + (red[2] + red[3]) + range_total; hdr[1] = cap; hdr[2] = (uint32_t)rows_here; hdr[3] = (uint32_t)r.F;
    hdr[4] = 0u;
  }
  const int w = first + wave;
  if ((int)blockIdx.x * kMsgWordsPerBlock + wave >= n_words) return;
  if (lane == 0) {
    reinterpret_cast<unsigned long long*>(msg + lay.bitmap)[w] = m;
    reinterpret_cast<uint32_t*>(msg + lay.offs)[w] = o0;
  }
  if (m == 0ull) return;
  float* out = reinterpret_cast<float*>(msg + lay.rows);
  uint8_t* mybits = bitpos[wave];
  if ((m >> lane) & 1ull) mybits[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int nset = (int)__popcll(m), F = r.F;
  const ChunkMap cm = chunk_map<VW>(F, lane, c_magic);
  for (int c0 = 0; c0 < cm.C; c0 += 64) {            // (one trip unless the row has more than 64 chunks)
    const int ch = cm.c + c0;
    const bool lane_on = cm.active && ch < cm.C;
    const int fs = chunk_start<VW>(F, lane_on ? ch : 0);
    ElemRef src[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) src[j] = elem_ref(r, fs + j);
    for (int k0 = 0; k0 < nset; k0 += 4 * cm.slots) {
      Vec<VW> v[4];
      bool on[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * cm.slots + cm.slot;
        on[u] = lane_on && k < nset && o0 + (uint32_t)k < cap;
        // (branch-free: a lane without a row re-reads the word's first set row -- see msg_merge_word)
        const int64_t row = row0 + (int64_t)w * 64 + mybits[k < nset ? k : 0];
#pragma unroll
        for (int j = 0; j < VW; ++j) v[u].v[j] = src[j].base[row * (int64_t)src[j].stride];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * cm.slots + cm.slot;
        if (on[u]) *reinterpret_cast<Vec<VW>*>(out + (size_t)(o0 + (uint32_t)k) * F + fs) = v[u];
      }
    }
  }
}

// The status word an apply kernel leaves: bits 1:0 = 1 (applied) / 2 (nothing applied), bits 32:2 = the largest count among the
// messages (clamped to 2^31 - 1: a poisoned owner message says 0xFFFFFFFF), bits 63:33 = the largest count their senders RECEIVED
// (header word 4: owners' messages of the sparse reduce-scatter; 0 otherwise).
__device__ __forceinline__ unsigned long long msg_status(bool bad, uint32_t worst, uint32_t worst_in) {
  const unsigned long long a = worst > 0x7FFFFFFFu ? 0x7FFFFFFFull : (unsigned long long)worst;
  const unsigned long long b = worst_in > 0x7FFFFFFFu ? 0x7FFFFFFFull : (unsigned long long)worst_in;
  return (bad ? 2ull : 1ull) | (a << 2) | (b << 33);
}

// Header check of n messages (uniform): largest count, and whether every message fits `cap` and has this call's shape.
// rows_of(q) = the number of rows message q must describe.
template <int MAXW, typename RowsOf>
__device__ __forceinline__ bool msg_headers_bad(const unsigned char* __restrict__ msgs, size_t msg_stride, int W, uint32_t cap, int F,
                                                RowsOf rows_of, uint32_t& worst, uint32_t& worst_in) {
  bool bad = false;
  worst = 0;
  worst_in = 0;
#pragma unroll
  for (int q = 0; q < MAXW; ++q) {
    if (q < W) {
      const uint32_t* hdr = reinterpret_cast<const uint32_t*>(msgs + (size_t)q * msg_stride);
      const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[0]);
      worst = n > worst ? n : worst;
      const uint32_t ni = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[4]);      // (an owner's message: the largest count it received)
      worst_in = ni > worst_in ? ni : worst_in;
      bad = bad || n > cap || hdr[1] != cap || hdr[2] != (uint32_t)rows_of(q) || hdr[3] != (uint32_t)F;
    }
  }
  return bad;
}

// The merge of word w of W messages (all describing the same rows): for the k-th row of the union (bit b of the word), the
// messages' rows added in RANK ORDER -> put(k, b, f, sum). The bitmap words and row offsets of the messages are wave-uniform
// (scalar registers); lane b holds, for row b of the word, its position in every message (offset + set bits below b) and the mask
// of the messages that hold it; a lane working on row b fetches both with ds_bpermute. kTrip row-instructions per trip of the row
// loop: their loads (up to MAXW each) are all issued before the first add -- a word holds ~13 rows of the union at C3; with
// dwordx4 lanes a row-instruction covers 4 rows of 59 floats: two trips per word. (Second build, one float per lane, two
// row-instructions per trip: 106 us at C3; the same kernel without its row loads 59, without its stores 82 -- a chain of round
// trips, not bytes: gpurun_out/r6h. Fifth build, dwordx4: 65 us.)
// Returns the union word (0: nothing to do).
template <int MAXW, int VW, typename Put>
__device__ __forceinline__ unsigned long long
msg_merge_word(const unsigned char* __restrict__ msgs, const size_t msg_stride, const int W, const int w, const MsgLayout lay,
               const int F, const uint32_t c_magic, uint8_t* ubits, Put put) {
  // kTrip x MAXW x VW registers of loads in flight. (dwordx4 lanes, MAXW = 8: one row-instruction per trip instead of two frees 40
  // registers -- 4 waves per SIMD instead of 3 -- and changes nothing: apply 67 against 65 us at C3, gpurun_out/r6ad.)
  constexpr int kTrip = MAXW == 1 ? 4 : (VW > 1 ? (MAXW <= 8 ? 2 : 1) : (MAXW <= 8 ? 4 : 2));
  const int lane = threadIdx.x & 63;
  unsigned long long U = 0ull;
  uint32_t posl[MAXW];         // lane b: position of row b of the word in message q (meaningful where the message holds the row)
  uint32_t hasl = 0;           // lane b: bit q set = message q holds row b
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < MAXW; ++q) {
    posl[q] = 0u;
    if (q < W) {
      const unsigned char* base = msgs + (size_t)q * msg_stride;
      const unsigned long long mv = reinterpret_cast<const unsigned long long*>(base + lay.bitmap)[w];
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mv);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mv >> 32));
      const unsigned long long mq = ((unsigned long long)hi << 32) | lo;
      const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(base + lay.offs)[w]);
      posl[q] = o + (uint32_t)__popcll(mq & below);
      hasl |= (uint32_t)((mq >> lane) & 1ull) << q;
      U |= mq;
    }
  }
  if (U == 0ull) return 0ull;
  if ((U >> lane) & 1ull) ubits[__popcll(U & below)] = (uint8_t)lane;      // k -> bit index of the k-th row of the union
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int nrow = (int)__popcll(U);
  const ChunkMap cm = chunk_map<VW>(F, lane, c_magic);
  const float* rowsq[MAXW];
#pragma unroll
  for (int q = 0; q < MAXW; ++q)
    rowsq[q] = reinterpret_cast<const float*>(msgs + (size_t)(q < W ? q : 0) * msg_stride + lay.rows);
  for (int c0 = 0; c0 < cm.C; c0 += 64) {            // (one trip unless the row has more than 64 chunks)
    const int ch = cm.c + c0;
    const bool lane_on = cm.active && ch < cm.C;
    const int fs = chunk_start<VW>(F, lane_on ? ch : 0);
    for (int k0 = 0; k0 < nrow; k0 += kTrip * cm.slots) {
      Vec<VW> v[kTrip][MAXW];
      uint32_t has[kTrip];
      int b[kTrip], kk[kTrip];
#pragma unroll
      for (int t = 0; t < kTrip; ++t) {
        kk[t] = k0 + t * cm.slots + cm.slot;
        const bool on = lane_on && kk[t] < nrow;
        b[t] = ubits[kk[t] < nrow ? kk[t] : 0];
        has[t] = (uint32_t)__shfl((int)hasl, b[t], 64);
        if (!on) has[t] = 0u;
        // BRANCH-FREE loads: a lane whose message does not hold the row (or a message slot >= W) reads the first chunk of that
        // message's rows and discards it. With the loads under `if (has >> q & 1)` every one of the up to kTrip x MAXW loads of a
        // trip sat in its own control-flow region and was waited for before the next was issued: ~13 us per trip, 109 us for the
        // kernel at C3 whatever the number of rows (gpurun_out/r6k) -- serialised round trips, not bytes.
#pragma unroll
        for (int q = 0; q < MAXW; ++q) {
          const uint32_t pos = (uint32_t)__shfl((int)posl[q], b[t], 64);
          const bool h = (has[t] >> q) & 1u;
          const size_t at = h ? (size_t)pos * F + fs : 0;
          v[t][q] = *reinterpret_cast<const Vec<VW>*>(rowsq[q] + at);
        }
      }
#pragma unroll
      for (int t = 0; t < kTrip; ++t) {
        if (has[t] == 0u) continue;
        Vec<VW> acc;
#pragma unroll
        for (int j = 0; j < VW; ++j) acc.v[j] = 0.f;
        bool any = false;
#pragma unroll
        for (int q = 0; q < MAXW; ++q)
          if (q < W && ((has[t] >> q) & 1u)) {
#pragma unroll
            for (int j = 0; j < VW; ++j) acc.v[j] = any ? __fadd_rn(acc.v[j], v[t][q].v[j]) : v[t][q].v[j];
            any = true;
          }
        put(kk[t], b[t], c0 == 0, ch, fs, acc);
      }
    }
  }
  return U;
}

// Messages -> row set. One wave per word of 64 rows.
//   slice_rows = 0: the W messages all describe the whole set; every row any of them holds receives the ranks' contributions
//                   added in rank order, STORED (rows nobody holds are left as they are).
//   slice_rows > 0: message y describes slice y (blockIdx.y): its rows are stored into the set's rows [y slice_rows, ...) -- the
//                   last step of the sparse reduce-scatter (the owners' reduced slices, disjoint).
// *status (one u64, may be page-locked host memory; msg_status above): applied, or NOTHING applied because some message overflowed
// its capacity / does not match this call's shape; stored by the first workgroup before any row is touched.
// MAXW: the most messages the call may carry (headers); MERGE: the most messages that describe one word (MAXW, or 1 in slice mode --
// with MERGE = MAXW there the W - 1 message slots nobody fills still cost a discarded load each: 43 -> 31 us at C3, gpurun_out/r6ad).
template <int MAXW, int MERGE, int VW>
__global__ void __launch_bounds__(256)
k_msg_apply(const Regions r, const unsigned char* __restrict__ msgs, const size_t msg_stride, const int W, const int32_t slice_rows,
            const MsgLayout lay, const uint32_t cap, unsigned long long* __restrict__ status,
            unsigned long long* __restrict__ touched, const uint32_t c_magic) {
  __shared__ uint8_t ubits_s[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
  const int y = (int)blockIdx.y;
  uint32_t worst, worst_in;
  const int32_t total_rows = r.rows;
  const bool bad = msg_headers_bad<MAXW>(msgs, msg_stride, W, cap, r.F,
                                          [=](int q) { return slice_len(total_rows, slice_rows, q); }, worst, worst_in);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && status)
    __hip_atomic_store(status, msg_status(bad, worst, worst_in), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int32_t rows_here = slice_len(r.rows, slice_rows, y);
  if (bad || w >= (rows_here + 63) / 64) return;
  const int64_t row0 = (int64_t)y * (slice_rows > 0 ? slice_rows : 0);
  // (the chunk a lane stores never changes: the regions of its floats are selected once -- chunk_map / elem_ref)
  const ChunkMap cm = chunk_map<VW>(r.F, lane, c_magic);
  const int fs0 = chunk_start<VW>(r.F, (cm.active && cm.c < cm.C) ? cm.c : 0);
  ElemRef d0[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) d0[j] = elem_ref(r, fs0 + j);
  const Regions& rr = r;
  auto put = [&](int k, int b, bool first_group, int ch, int fs, const Vec<VW>& v) {
    (void)k;
    const int64_t row = row0 + (int64_t)w * 64 + b;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      if (fs + j < ch * VW) continue;              // (the floats an overlapping last chunk shares with its neighbour: the neighbour stores them)
      const ElemRef d = first_group ? d0[j] : elem_ref(rr, fs + j);            // (later groups only for rows of more than 64 chunks)
      d.base[row * (int64_t)d.stride] = v.v[j];
    }
  };
  const unsigned char* data = slice_rows > 0 ? msgs + (size_t)y * msg_stride : msgs;
  const unsigned long long U = msg_merge_word<MERGE, VW>(data, msg_stride, slice_rows > 0 ? 1 : W, w, lay, r.F, c_magic, ubits_s[wave], put);
  if (touched && lane == 0) touched[(row0 >> 6) + w] = U;        // (the word of the union, zero included)
}

// Messages -> message: the OWNER side of the sparse reduce-scatter. The W messages describe the same slice (one from every rank);
// the output describes the union of their rows, every row = the ranks' contributions added in rank order. The output's count may
// exceed cap_out (recorded in its header, the rows beyond are not written); if an INPUT does not fit / match, the output's count is
// 0xFFFFFFFF: whoever applies it applies nothing (gsr_rowmsg_apply_slices), i.e. the arena of no rank is touched.
template <int MAXW, int VW>
__global__ void __launch_bounds__(256)
k_msg_reduce(const unsigned char* __restrict__ msgs, const size_t msg_stride, const int W, const int32_t rows, const int F,
             const MsgLayout lay_in, const uint32_t cap_in, unsigned char* __restrict__ out, const MsgLayout lay_out,
             const uint32_t cap_out, const uint32_t c_magic) {
  __shared__ uint8_t ubits_s[4][64];
  __shared__ uint32_t red[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int32_t n_words = (rows + 63) / 64;
  uint32_t worst, worst_in;
  const bool bad = msg_headers_bad<MAXW>(msgs, msg_stride, W, cap_in, F, [=](int) { return rows; }, worst, worst_in);
  uint32_t* hdr = reinterpret_cast<uint32_t*>(out);
  if (bad) {
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
      hdr[0] = 0xFFFFFFFFu; hdr[1] = cap_out; hdr[2] = (uint32_t)rows; hdr[3] = (uint32_t)F; hdr[4] = worst;
    }
    return;
  }
  // rows of the union in front of this workgroup's words
  const int first = min((int)blockIdx.x * kMsgWordsPerBlock, n_words);
  auto union_word = [&](int wq) {
    unsigned long long u = 0ull;
#pragma unroll
    for (int q = 0; q < MAXW; ++q)
      if (q < W) u |= reinterpret_cast<const unsigned long long*>(msgs + (size_t)q * msg_stride + lay_in.bitmap)[wq];
    return u;
  };
  uint32_t c = 0;
  for (int wq = tid; wq < first; wq += 256) c += (uint32_t)__popcll(union_word(wq));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
  if (lane == 0) red[wave] = c;
  __syncthreads();
  uint32_t o0 = (red[0] + red[1]) + (red[2] + red[3]);
  uint32_t range_total = 0;
#pragma unroll
  for (int j = 0; j < kMsgWordsPerBlock; ++j) {
    const int wj = first + j;
    const uint32_t pj = wj < n_words ? (uint32_t)__popcll(union_word(wj)) : 0u;
    if (j < wave) o0 += pj;
    range_total += pj;
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    hdr[0] = (red[0] + red[1]) + (red[2] + red[3]) + range_total; hdr[1] = cap_out; hdr[2] = (uint32_t)rows; hdr[3] = (uint32_t)F;
    hdr[4] = worst;                    // the largest count this owner received: the capacity policy of the first phase needs it
  }
  const int w = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kMsgWordsPerBlock + wave);
  if (w >= n_words) return;
  float* orow = reinterpret_cast<float*>(out + lay_out.rows);
  auto put = [&](int k, int b, bool first_group, int ch, int fs, const Vec<VW>& v) {
    (void)b; (void)first_group; (void)ch;
    const uint32_t pos = o0 + (uint32_t)k;
    // (an overlapping last chunk re-writes floats of its neighbour with the same values: same loads, same order of adds)
    if (pos < cap_out) *reinterpret_cast<Vec<VW>*>(orow + (size_t)pos * F + fs) = v;
  };
  const unsigned long long U = msg_merge_word<MAXW, VW>(msgs, msg_stride, W, w, lay_in, F, c_magic, ubits_s[wave], put);
  if (lane == 0) {
    reinterpret_cast<unsigned long long*>(out + lay_out.bitmap)[w] = U;
    reinterpret_cast<uint32_t*>(out + lay_out.offs)[w] = o0;
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

static uint32_t f_magic_of(int F) { return (uint32_t)(0xFFFFFFFFu / (uint32_t)F + 1u); }   // floor(x / F) = umulhi(x, magic); 0 for F = 1
// floats per lane on the message side (chunk_map) and the magic of the chunks per row
static int msg_vw(int F) { return F >= 4 ? 4 : 1; }
static uint32_t c_magic_of(int F) { return f_magic_of((F + msg_vw(F) - 1) / msg_vw(F)); }

static int rowmsg_pack(const GsrRowSet* rs, const uint64_t* mask, void* msgs, uint64_t msg_stride, int32_t n_slices,
                       int32_t slice_rows, uint32_t cap, void* stream_) {
  Regions r;
  const int rc = make_regions(rs, r);
  if (rc) return rc;
  if (!mask || !msgs || (reinterpret_cast<uintptr_t>(mask) & 7u) || (reinterpret_cast<uintptr_t>(msgs) & 255u)) return GSR_EINVAL;
  if (slice_rows < 0 || (slice_rows & 63) || n_slices < 1 || (slice_rows == 0 && n_slices != 1) || (msg_stride & 255u)) return GSR_EINVAL;
  if (slice_rows > 0 && (int64_t)n_slices * slice_rows < rs->rows) return GSR_EINVAL;
  const int32_t per = slice_rows > 0 ? slice_rows : rs->rows;
  const MsgLayout lay = msg_layout(per, r.F, cap);
  if (n_slices > 1 && msg_stride < lay.total) return GSR_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(msgs);
  const int32_t n_words = (per + 63) / 64;
  const uint32_t blocks = (uint32_t)((n_words + kMsgWordsPerBlock - 1) / kMsgWordsPerBlock);
  const dim3 grid(blocks ? blocks : 1u, (uint32_t)n_slices);
  if (msg_vw(r.F) == 4)
    hipLaunchKernelGGL(k_msg_pack<4>, grid, dim3(256), 0, stream, r, reinterpret_cast<const unsigned long long*>(mask),
                       reinterpret_cast<unsigned char*>(msgs), (size_t)msg_stride, slice_rows, lay, cap, c_magic_of(r.F));
  else
    hipLaunchKernelGGL(k_msg_pack<1>, grid, dim3(256), 0, stream, r, reinterpret_cast<const unsigned long long*>(mask),
                       reinterpret_cast<unsigned char*>(msgs), (size_t)msg_stride, slice_rows, lay, cap, c_magic_of(r.F));
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

int gsr_rowmsg_pack(const GsrRowSet* rs, const uint64_t* mask, void* msg, uint32_t cap, void* stream_) {
  return rowmsg_pack(rs, mask, msg, 0, 1, 0, cap, stream_);
}

int gsr_rowmsg_pack_slices(const GsrRowSet* rs, const uint64_t* mask, void* msgs, uint64_t msg_stride, int32_t n_slices,
                           int32_t slice_rows, uint32_t cap, void* stream_) {
  if (slice_rows <= 0) return GSR_EINVAL;
  return rowmsg_pack(rs, mask, msgs, msg_stride, n_slices, slice_rows, cap, stream_);
}

static int rowmsg_apply(const GsrRowSet* rs, const void* msgs, uint64_t msg_stride, int32_t n_msgs, int32_t slice_rows, uint32_t cap,
                        uint64_t* status, uint64_t* touched, void* stream_) {
  Regions r;
  const int rc = make_regions(rs, r);
  if (rc) return rc;
  if (!msgs || n_msgs < 1 || n_msgs > 16 || (reinterpret_cast<uintptr_t>(msgs) & 255u) || (msg_stride & 255u)) return GSR_EINVAL;
  if ((touched && (reinterpret_cast<uintptr_t>(touched) & 7u)) || (status && (reinterpret_cast<uintptr_t>(status) & 7u))) return GSR_EINVAL;
  if (slice_rows < 0 || (slice_rows & 63) || (slice_rows > 0 && (int64_t)n_msgs * slice_rows < rs->rows)) return GSR_EINVAL;
  const int32_t per = slice_rows > 0 ? slice_rows : rs->rows;
  const MsgLayout lay = msg_layout(per, r.F, cap);
  if (msg_stride < lay.total) return GSR_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(msgs);
  const int32_t n_words = (per + 63) / 64;
  const dim3 grid((uint32_t)((n_words + 3) / 4 > 0 ? (n_words + 3) / 4 : 1), slice_rows > 0 ? (uint32_t)n_msgs : 1u);
  const unsigned char* m = reinterpret_cast<const unsigned char*>(msgs);
  unsigned long long* st = reinterpret_cast<unsigned long long*>(status);
  unsigned long long* tc = reinterpret_cast<unsigned long long*>(touched);
  const uint32_t cmg = c_magic_of(r.F);
  const bool wide = n_msgs > 8, vec = msg_vw(r.F) == 4;
#define GSR_APPLY(MAXW, MERGE, VW) \
  hipLaunchKernelGGL((k_msg_apply<MAXW, MERGE, VW>), grid, dim3(256), 0, stream, r, m, (size_t)msg_stride, n_msgs, slice_rows, lay, cap, st, tc, cmg)
  if (slice_rows > 0) {
    if (vec) GSR_APPLY(16, 1, 4); else GSR_APPLY(16, 1, 1);
  } else if (!wide) {
    if (vec) GSR_APPLY(8, 8, 4); else GSR_APPLY(8, 8, 1);
  } else {
    if (vec) GSR_APPLY(16, 16, 4); else GSR_APPLY(16, 16, 1);
  }
#undef GSR_APPLY
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

int gsr_rowmsg_apply(const GsrRowSet* rs, const void* msgs, uint64_t msg_stride, int32_t n_msgs, uint32_t cap, uint64_t* status,
                     uint64_t* touched, void* stream_) {
  return rowmsg_apply(rs, msgs, msg_stride, n_msgs, 0, cap, status, touched, stream_);
}

int gsr_rowmsg_apply_slices(const GsrRowSet* rs, const void* msgs, uint64_t msg_stride, int32_t n_slices, int32_t slice_rows,
                            uint32_t cap, uint64_t* status, uint64_t* touched, void* stream_) {
  if (slice_rows <= 0) return GSR_EINVAL;
  return rowmsg_apply(rs, msgs, msg_stride, n_slices, slice_rows, cap, status, touched, stream_);
}

int gsr_rowmsg_reduce(int32_t rows, int32_t layout_rows, int32_t row_floats, const void* msgs, uint64_t msg_stride, int32_t n_msgs,
                      uint32_t cap_in, void* msg_out, uint32_t cap_out, void* stream_) {
  if (rows < 0 || row_floats < 1 || row_floats > 1024 || !msgs || !msg_out || n_msgs < 1 || n_msgs > 16) return GSR_EINVAL;
  if ((reinterpret_cast<uintptr_t>(msgs) & 255u) || (reinterpret_cast<uintptr_t>(msg_out) & 255u) || (msg_stride & 255u)) return GSR_EINVAL;
  if (layout_rows <= 0) layout_rows = rows;
  if (layout_rows < rows) return GSR_EINVAL;
  const MsgLayout lin = msg_layout(layout_rows, row_floats, cap_in), lout = msg_layout(layout_rows, row_floats, cap_out);
  if (msg_stride < lin.total) return GSR_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(msgs);
  const int32_t n_words = (rows + 63) / 64;
  const dim3 grid((uint32_t)((n_words + kMsgWordsPerBlock - 1) / kMsgWordsPerBlock > 0 ? (n_words + kMsgWordsPerBlock - 1) / kMsgWordsPerBlock : 1));
  const unsigned char* m = reinterpret_cast<const unsigned char*>(msgs);
  unsigned char* mo = reinterpret_cast<unsigned char*>(msg_out);
  const uint32_t cmg = c_magic_of(row_floats);
  const bool wide = n_msgs > 8, vec = msg_vw(row_floats) == 4;
#define GSR_REDUCE(MAXW, VW) \
  hipLaunchKernelGGL((k_msg_reduce<MAXW, VW>), grid, dim3(256), 0, stream, m, (size_t)msg_stride, n_msgs, rows, row_floats, lin, \
                     cap_in, mo, lout, cap_out, cmg)
  if (!wide && vec) GSR_REDUCE(8, 4); else if (!wide) GSR_REDUCE(8, 1); else if (vec) GSR_REDUCE(16, 4); else GSR_REDUCE(16, 1);
#undef GSR_REDUCE
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

}  // extern "C"
