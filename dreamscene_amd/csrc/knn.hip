// knn.hip -- mean squared distance to the 3 nearest neighbours of every point: the MI355X-native replacement of
// `simple_knn._C.distCUDA2` (un-vendored CUDA, README.md:48,51), which the reference calls once per model
// initialisation to seed the Gaussian scales (gs_renderer.py:9, 590-594). SURVEY.md section 8(f) rank 1.
//
// Exact 3-NN on a uniform grid hash, all on the device, no host round trip:
//   k_knn_bbox      bounding box (ordered-int atomics)
//   k_knn_cells     cell id of every point; the grid has R^3 cells with R = clamp(cbrt(N/2), 1, 256) per axis
//   radix sort      (cell id, point index) with the binning's stable LSD sort
//   k_knn_ranges    [start,end) of every non-empty cell in the sorted order
//   k_knn_query     one thread per point (in cell order: neighbours share cache lines): grows a cube of cells
//                   shell by shell until the 3rd best squared distance is <= (s * cell)^2, which no point
//                   outside the visited cube can beat  -> exact
// Output: (d1^2 + d2^2 + d3^2) / 3 per point, in the caller's point order. Fewer than 4 points: the missing
// neighbours count as FLT_MAX (as an exhaustive search with FLT_MAX-initialised bests does).
#include <float.h>

#include "gsr_common.h"
#include "radix_sort.h"

namespace {

constexpr int kMaxR = 256;

struct KnnGrid {
  float lo[3];
  float inv_cell;   // cells per unit length
  float cell;
  int R;
};

__device__ __forceinline__ int f2ord(float f) {          // float -> int with the same ordering
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// bbox[0..2] = min (ordered ints), bbox[3..5] = max; initialised to +/- "infinity" by the caller
__global__ void __launch_bounds__(256) k_knn_bbox(const float* __restrict__ pts, int n, int* __restrict__ bbox) {
  int mn[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int o = f2ord(pts[3 * i + a]);
      mn[a] = min(mn[a], o);
      mx[a] = max(mx[a], o);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      atomicMin(bbox + a, mn[a]);
      atomicMax(bbox + 3 + a, mx[a]);
    }
  }
}

__device__ __forceinline__ KnnGrid make_grid(const int* bbox, int n) {
  KnnGrid g;
  float ext = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = ord2f(bbox[a]);
    ext = fmaxf(ext, ord2f(bbox[3 + a]) - g.lo[a]);
  }
  int R = (int)cbrtf(0.5f * (float)n);
  R = max(1, min(kMaxR, R));
  if (!(ext > 0.f)) { ext = 1.f; R = 1; }
  g.R = R;
  g.cell = ext / (float)R;
  g.inv_cell = (float)R / ext;
  return g;
}

__device__ __forceinline__ int cell_coord(const KnnGrid& g, float v, int a) {
  const int c = (int)((v - g.lo[a]) * g.inv_cell);
  return max(0, min(g.R - 1, c));
}

__global__ void __launch_bounds__(256)
k_knn_cells(const float* __restrict__ pts, int n, const int* __restrict__ bbox, uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KnnGrid g = make_grid(bbox, n);
  const int cx = cell_coord(g, pts[3 * i], 0), cy = cell_coord(g, pts[3 * i + 1], 1), cz = cell_coord(g, pts[3 * i + 2], 2);
  keys[i] = (uint32_t)((cz * g.R + cy) * g.R + cx);
}

__global__ void __launch_bounds__(256)
k_knn_ranges(const uint32_t* __restrict__ keys, int n, uint32_t* __restrict__ ranges) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t c = keys[j];
  if (j == 0 || keys[j - 1] != c) ranges[2 * c] = (uint32_t)j;
  if (j == n - 1 || keys[j + 1] != c) ranges[2 * c + 1] = (uint32_t)(j + 1);
}

__device__ __forceinline__ void push3(float d, float& b0, float& b1, float& b2) {
  if (d < b2) {
    if (d < b1) {
      b2 = b1;
      if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
    } else {
      b2 = d;
    }
  }
}

__global__ void __launch_bounds__(256)
k_knn_query(const float* __restrict__ pts, int n, const int* __restrict__ bbox, const uint32_t* __restrict__ sorted_idx,
            const uint32_t* __restrict__ ranges, float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const KnnGrid g = make_grid(bbox, n);
  const uint32_t me = sorted_idx[j];
  const float px = pts[3 * me], py = pts[3 * me + 1], pz = pts[3 * me + 2];
  const int cx = cell_coord(g, px, 0), cy = cell_coord(g, py, 1), cz = cell_coord(g, pz, 2);
  float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
  const int R = g.R;
  for (int s = 0; s < R; ++s) {
    // all cells at Chebyshev distance exactly s
    const int z0 = max(0, cz - s), z1 = min(R - 1, cz + s);
    const int y0 = max(0, cy - s), y1 = min(R - 1, cy + s);
    const int x0 = max(0, cx - s), x1 = min(R - 1, cx + s);
    for (int z = z0; z <= z1; ++z) {
      const bool zf = (z == cz - s) || (z == cz + s);
      for (int y = y0; y <= y1; ++y) {
        const bool yf = zf || (y == cy - s) || (y == cy + s);
        // on a face of the cube every x of the row belongs to the shell, otherwise only its two ends
        const int step = yf ? 1 : max(1, 2 * s);
        for (int x = yf ? x0 : cx - s; x <= x1; x += step) {
          if (x < x0) continue;
          const uint32_t c = (uint32_t)((z * R + y) * R + x);
          const uint32_t a = ranges[2 * c], e = ranges[2 * c + 1];
          for (uint32_t k = a; k < e; ++k) {
            const uint32_t o = sorted_idx[k];
            if (o == me) continue;
            const float dx = pts[3 * o] - px, dy = pts[3 * o + 1] - py, dz = pts[3 * o + 2] - pz;
            push3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
          }
        }
      }
    }
    const float reach = (float)s * g.cell;      // every unvisited point is farther than this
    if (b2 <= reach * reach) break;
  }
  out[me] = (b0 + b1 + b2) / 3.0f;
}

// bbox[0..2] = +INT_MAX (running minima), bbox[3..5] = INT_MIN (running maxima)
__global__ void k_knn_init(int* __restrict__ bbox) {
  if (threadIdx.x < 6) bbox[threadIdx.x] = threadIdx.x < 3 ? 0x7FFFFFFF : (int)0x80000000;
}

}  // namespace

extern "C" size_t gsr_knn_scratch_bytes(int32_t n) {
  const uint64_t m = n > 0 ? (uint64_t)n : 1;
  const size_t cells = (size_t)kMaxR * kMaxR * kMaxR;
  const uint64_t r = (uint64_t)cbrt(0.5 * (double)m) + 1;
  const size_t rc = (size_t)std::min<uint64_t>(r * r * r, cells);
  return 4 * align256(m * 4) + sort_hist_bytes(m, kItemsSmall, kOsItemsSmall) + align256(256 * 4) +
         align256(rc * 8) + 256 + 1024;
}

// points [n,3] fp32 (device) -> out [n] fp32 (device); scratch: gsr_knn_scratch_bytes(n) bytes, 256-byte aligned.
// Replaces simple_knn._C.distCUDA2 (gs_renderer.py:590-593).
extern "C" int gsr_knn_mean_dist2(const float* points, int32_t n, float* out, void* scratch, size_t scratch_bytes,
                                  void* stream_) {
  if (n < 0 || (n > 0 && (!points || !out))) return GSR_EINVAL;
  if (n == 0) return GSR_OK;
  if (!scratch || scratch_bytes < gsr_knn_scratch_bytes(n)) return GSR_ESCRATCH;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(points);
  const uint64_t m = (uint64_t)n;
  char* b = (char*)scratch;
  uint32_t* k0 = (uint32_t*)b; b += align256(m * 4);
  uint32_t* k1 = (uint32_t*)b; b += align256(m * 4);
  uint32_t* v0 = (uint32_t*)b; b += align256(m * 4);
  uint32_t* v1 = (uint32_t*)b; b += align256(m * 4);
  uint32_t* hist = (uint32_t*)b; b += sort_hist_bytes(m, kItemsSmall, kOsItemsSmall);
  uint32_t* totals = (uint32_t*)b; b += align256(256 * 4);
  int* bbox = (int*)b; b += 256;
  uint32_t* ranges = (uint32_t*)b;
  int R = (int)cbrt(0.5 * (double)m);
  R = std::max(1, std::min(kMaxR, R));
  // the device recomputes R the same way from n (cbrtf): keep one cell of slack per axis for rounding
  const size_t cells = (size_t)std::min<uint64_t>((uint64_t)(R + 1) * (R + 1) * (R + 1), (uint64_t)kMaxR * kMaxR * kMaxR);
  hipLaunchKernelGGL(k_knn_init, dim3(1), dim3(64), 0, stream, bbox);   // (min, max) sentinels: no host->device copy
  GSR_HIP(gsr_zero_async(ranges, cells * 8, stream));
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_knn_bbox, dim3(std::min(nb, 1024)), dim3(256), 0, stream, points, n, bbox);
  hipLaunchKernelGGL(k_knn_cells, dim3(nb), dim3(256), 0, stream, points, n, bbox, k0);
  const int where = radix_sort_u32<kItemsSmall, kOsItemsSmall>(k0, v0, k1, v1, nullptr, m, 24, true, nullptr, hist, totals, stream);
  const uint32_t* sk = where ? k1 : k0;
  const uint32_t* sv = where ? v1 : v0;
  hipLaunchKernelGGL(k_knn_ranges, dim3(nb), dim3(256), 0, stream, sk, n, ranges);
  hipLaunchKernelGGL(k_knn_query, dim3(nb), dim3(256), 0, stream, points, n, bbox, sv, ranges, out);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}
