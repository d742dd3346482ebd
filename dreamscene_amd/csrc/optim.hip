// optim.hip -- SURVEY.md section 8(f) rank 3: the optimizer half of the post-raster epilogue, gfx950.
//
// The reference steps torch.optim.Adam(l, lr=0.0, eps=1e-15) over 6-7 parameter groups per GaussianModel
// (gs_renderer.py:615-653): per step and group a chain of elementwise kernels over every parameter. Here all groups
// go through ONE launch: a table of (param, grad, exp_avg, exp_avg_sq, numel, lr) in the kernel arguments, one
// pass over memory (16 bytes read + 12 written per element, + 4 when the gradient is cleared in the same pass).
// The arithmetic is torch's single-tensor Adam (torch/optim/adam.py, no amsgrad / weight decay / maximize), one
// rounding per operator (this file is built with -ffp-contract=off):
//   m <- m + (g - m) (1 - beta1);  v <- v beta2 + (1 - beta2) g g
//   p <- p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include "gsr_common.h"

namespace {

struct AdamTab {
  int32_t n;
  int32_t fblk[GSR_MAX_ADAM_GROUPS + 1];
  float* param[GSR_MAX_ADAM_GROUPS];
  float* grad[GSR_MAX_ADAM_GROUPS];
  float* m[GSR_MAX_ADAM_GROUPS];
  float* v[GSR_MAX_ADAM_GROUPS];
  int64_t numel[GSR_MAX_ADAM_GROUPS];
  float step_size[GSR_MAX_ADAM_GROUPS];   // lr / (1 - beta1^t)
};

constexpr int kAdamPerBlock = 256 * 4;

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float w1, float beta2,
                                         float w2, float bc2_sqrt, float eps) {
  m = m + (g - m) * w1;
  v = v * beta2 + (w2 * g) * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256)
k_adam(const AdamTab t, const float w1, const float beta2, const float w2, const float bc2_sqrt, const float eps,
       const int zero_grad) {
  int gi = 0;
  for (int k = 1; k < t.n; ++k) gi += ((int)blockIdx.x >= t.fblk[k]) ? 1 : 0;
  const int64_t e0 = ((int64_t)blockIdx.x - t.fblk[gi]) * kAdamPerBlock + (int64_t)threadIdx.x * 4;
  const int64_t n = t.numel[gi];
  if (e0 >= n) return;
  float* __restrict__ P = t.param[gi];
  float* __restrict__ G = t.grad[gi];
  float* __restrict__ M = t.m[gi];
  float* __restrict__ V = t.v[gi];
  const float ss = t.step_size[gi];
  if (e0 + 3 < n) {
    float4 p = *reinterpret_cast<const float4*>(P + e0);
    const float4 g = *reinterpret_cast<const float4*>(G + e0);
    float4 m = *reinterpret_cast<const float4*>(M + e0);
    float4 v = *reinterpret_cast<const float4*>(V + e0);
    adam_one(p.x, g.x, m.x, v.x, ss, w1, beta2, w2, bc2_sqrt, eps);
    adam_one(p.y, g.y, m.y, v.y, ss, w1, beta2, w2, bc2_sqrt, eps);
    adam_one(p.z, g.z, m.z, v.z, ss, w1, beta2, w2, bc2_sqrt, eps);
    adam_one(p.w, g.w, m.w, v.w, ss, w1, beta2, w2, bc2_sqrt, eps);
    *reinterpret_cast<float4*>(P + e0) = p;
    *reinterpret_cast<float4*>(M + e0) = m;
    *reinterpret_cast<float4*>(V + e0) = v;
    if (zero_grad) *reinterpret_cast<float4*>(G + e0) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int64_t e = e0; e < n; ++e) {
      float p = P[e], m = M[e], v = V[e];
      adam_one(p, G[e], m, v, ss, w1, beta2, w2, bc2_sqrt, eps);
      P[e] = p; M[e] = m; V[e] = v;
      if (zero_grad) G[e] = 0.f;
    }
  }
}

}  // namespace

extern "C" int gsr_adam_step(const GsrAdamGroup* groups, int32_t n_groups, int32_t step, double beta1, double beta2,
                             double eps, int32_t zero_grad, void* stream_) {
  if (!groups || n_groups < 0 || n_groups > GSR_MAX_ADAM_GROUPS || step < 1) return GSR_EINVAL;
  if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return GSR_EINVAL;
  AdamTab t = AdamTab{};
  // the hyper-parameters are Python floats (doubles) in torch: derived constants are formed in double and rounded once
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  int32_t blk = 0, n = 0;
  for (int i = 0; i < n_groups; ++i) {
    const GsrAdamGroup& g = groups[i];
    if (g.numel < 0) return GSR_EINVAL;
    if (g.numel == 0) continue;
    if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq) return GSR_EINVAL;
    if (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15u) return GSR_EINVAL;
    t.fblk[n] = blk;
    t.param[n] = g.param; t.grad[n] = g.grad; t.m[n] = g.exp_avg; t.v[n] = g.exp_avg_sq;
    t.numel[n] = g.numel;
    t.step_size[n] = (float)((double)g.lr / bc1);
    blk += (int32_t)((g.numel + kAdamPerBlock - 1) / kAdamPerBlock);
    ++n;
  }
  t.n = n;
  for (int i = n; i <= GSR_MAX_ADAM_GROUPS; ++i) t.fblk[i] = blk;
  if (blk == 0) return GSR_OK;
  GsrDeviceGuard dev(t.param[0]);
  hipLaunchKernelGGL(k_adam, dim3((uint32_t)blk), dim3(256), 0, (hipStream_t)stream_, t, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (int)zero_grad);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}
