/* A caller of libgsrast.so written in plain C against include/gsrast.h only -- no Python, no torch: what a binding in
 * another host language does. `caller nogpu` exercises the entry points that need no device
 * (version, error strings, sizes, argument validation); `caller gpu` allocates with the HIP runtime's C API, renders
 * 256 Gaussians forward + backward on the current device and checks the results are sane. Built by tests/test_abi.py
 * with gcc (not hipcc): the header is C. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsrast.h"

#ifdef WITH_HIP
#include <hip/hip_runtime_api.h>
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define GSRCHECK(x) do { int rc_ = (x); if (rc_ != GSR_OK) { fprintf(stderr, "%s: %s (%d)\n", #x, gsr_strerror(rc_), rc_); return 3; } } while (0)

static void* dalloc(size_t n) { void* p = NULL; if (hipMalloc(&p, n ? n : 1) != hipSuccess) return NULL; hipMemset(p, 0, n ? n : 1); return p; }
static void* upload(const void* h, size_t n) { void* p = dalloc(n); if (p) hipMemcpy(p, h, n, hipMemcpyHostToDevice); return p; }

static int run_gpu(void) {
  enum { P = 256, K = 4, H = 64, W = 80 };
  const float tanfov = 0.5f, zn = 0.01f, zf = 100.0f;
  float means[P * 3], scales[P * 3], rots[P * 4], opac[P], shs[P * K * 3];
  unsigned s = 12345u;
  for (int i = 0; i < P; ++i) {
    float r[3];
    for (int k = 0; k < 3; ++k) { s = s * 1664525u + 1013904223u; r[k] = (float)(s >> 8) / 16777216.0f; }
    means[3 * i] = (r[0] - 0.5f) * 1.6f; means[3 * i + 1] = (r[1] - 0.5f) * 1.2f; means[3 * i + 2] = 2.0f + 2.0f * r[2];
    for (int k = 0; k < 3; ++k) scales[3 * i + k] = 0.05f + 0.05f * r[k];
    rots[4 * i] = 1.f; rots[4 * i + 1] = rots[4 * i + 2] = rots[4 * i + 3] = 0.f;
    opac[i] = 0.3f + 0.6f * r[0];
    for (int k = 0; k < K * 3; ++k) shs[i * K * 3 + k] = (k < 3) ? (r[k] - 0.5f) * 3.0f : 0.05f * (r[k % 3] - 0.5f);
  }
  float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};     /* camera at the origin looking down +z */
  float proj[16] = {0};                                                   /* view . projection^T (row-vector form) */
  proj[0] = 1.0f / tanfov; proj[5] = 1.0f / tanfov; proj[10] = zf / (zf - zn); proj[11] = 1.0f; proj[14] = -(zf * zn) / (zf - zn);
  float bg[3] = {0.f, 0.f, 0.f}, campos[3] = {0.f, 0.f, 0.f};

  GsrView v; memset(&v, 0, sizeof v);
  v.P = P; v.sh_stride = K; v.sh_degree = 1; v.image_height = H; v.image_width = W;
  v.tanfovx = tanfov * (float)W / (float)H; v.tanfovy = tanfov; v.scale_modifier = 1.0f;
  proj[0] = 1.0f / v.tanfovx;
  v.bg = (const float*)upload(bg, sizeof bg); v.viewmatrix = (const float*)upload(view, sizeof view);
  v.projmatrix = (const float*)upload(proj, sizeof proj); v.campos = (const float*)upload(campos, sizeof campos);
  GsrGaussians g; memset(&g, 0, sizeof g);
  g.means3D = (const float*)upload(means, sizeof means); g.opacities = (const float*)upload(opac, sizeof opac);
  g.shs = (const float*)upload(shs, sizeof shs); g.scales = (const float*)upload(scales, sizeof scales);
  g.rotations = (const float*)upload(rots, sizeof rots);

  const uint32_t tiles = gsr_num_tiles(H, W);
  GsrGeom geom; memset(&geom, 0, sizeof geom);
  geom.splat = (float*)dalloc((size_t)P * 48); geom.radii = (int32_t*)dalloc(P * 4);
  geom.tiles_touched = (uint32_t*)dalloc(P * 4); geom.block_offsets = (uint32_t*)dalloc((gsr_num_blocks(P) + 8) * 4);
  geom.scratch_bytes = gsr_project_scratch_bytes(P); geom.scratch = dalloc(geom.scratch_bytes);
  uint64_t n_pairs = 0;
  GSRCHECK(gsr_forward_project(&v, &g, &geom, &n_pairs, NULL, NULL));      /* NULL stream = the default stream */
  if (n_pairs == 0 || n_pairs > (uint64_t)P * tiles) { fprintf(stderr, "implausible pair count %llu\n", (unsigned long long)n_pairs); return 4; }

  GsrBinning b; memset(&b, 0, sizeof b);
  b.point_list = (uint32_t*)dalloc(n_pairs * 4); b.ranges = (uint32_t*)dalloc((size_t)tiles * 8);
  b.bwd_items_cap = (uint32_t)(n_pairs / 256 + tiles);
  b.tile_work = (uint32_t*)dalloc(((size_t)tiles + 2 + 2 * (size_t)b.bwd_items_cap) * 4 + 64);
  b.scratch_bytes = gsr_sort_scratch_bytes(n_pairs, tiles); b.scratch = dalloc(b.scratch_bytes);
  GsrImages im; memset(&im, 0, sizeof im);
  im.color = (float*)dalloc((size_t)3 * H * W * 4); im.depth_alpha = (float*)dalloc((size_t)2 * H * W * 4);
  im.final_T = (float*)dalloc((size_t)H * W * 4); im.n_contrib = (uint32_t*)dalloc((size_t)H * W * 4);
  im.tile_depth = (uint32_t*)dalloc((size_t)tiles * 4); im.ckpt = (float*)dalloc((n_pairs / 256 + 1) * 6 * 256 * 4);
  GSRCHECK(gsr_forward_render(&v, &geom, n_pairs, &b, &im, NULL, NULL));

  static float img[3 * H * W], da[2 * H * W];
  HIPCHECK(hipMemcpy(img, im.color, sizeof img, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(da, im.depth_alpha, sizeof da, hipMemcpyDeviceToHost));
  double sum = 0, asum = 0;
  for (int i = 0; i < 3 * H * W; ++i) { if (!isfinite(img[i]) || img[i] < 0.f) { fprintf(stderr, "bad pixel\n"); return 5; } sum += img[i]; }
  for (int i = 0; i < H * W; ++i) { const float a = da[H * W + i]; if (!(a >= 0.f && a <= 1.0001f)) { fprintf(stderr, "alpha out of range\n"); return 5; } asum += a; }
  if (!(asum > 10.0) || !(sum > 1.0)) { fprintf(stderr, "nothing was rendered (sum %g, alpha %g)\n", sum, asum); return 5; }

  /* backward from dL/dimage = 1, dL/d(depth_alpha) = 0: dL/dopacity must be positive somewhere, everything finite */
  static float gimg[3 * H * W], gda[2 * H * W];
  for (int i = 0; i < 3 * H * W; ++i) gimg[i] = 1.0f;
  GsrImageGrads ig; ig.dL_dcolor = (const float*)upload(gimg, sizeof gimg); ig.dL_ddepth_alpha = (const float*)upload(gda, sizeof gda);
  GsrGrads gr; memset(&gr, 0, sizeof gr);
  gr.dL_dmeans3D = (float*)dalloc(P * 12); gr.dL_dmeans2D = (float*)dalloc(P * 12); gr.dL_dopacities = (float*)dalloc(P * 4);
  gr.dL_dshs = (float*)dalloc((size_t)P * K * 12); gr.dL_dscales = (float*)dalloc(P * 12); gr.dL_drotations = (float*)dalloc(P * 16);
  gr.partials = (float*)dalloc((size_t)P * GSR_PARTIAL_WORDS * 4);
  GSRCHECK(gsr_backward(&v, &g, &geom, &b, &im, &ig, &gr, NULL, NULL));
  HIPCHECK(hipDeviceSynchronize());
  static float gop[P], gm[P * 3];
  HIPCHECK(hipMemcpy(gop, gr.dL_dopacities, sizeof gop, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(gm, gr.dL_dmeans3D, sizeof gm, hipMemcpyDeviceToHost));
  int pos = 0;
  for (int i = 0; i < P; ++i) { if (!isfinite(gop[i]) || !isfinite(gm[3 * i])) { fprintf(stderr, "non-finite gradient\n"); return 6; } pos += gop[i] > 0.f; }
  if (pos < P / 4) { fprintf(stderr, "dL/dopacity positive for only %d Gaussians\n", pos); return 6; }
  /* a second device thread-of-control state: errors are reported, not thrown */
  GsrView bad = v; bad.sh_degree = 7;
  if (gsr_forward_project(&bad, &g, &geom, &n_pairs, NULL, NULL) != GSR_EINVAL) return 7;
  printf("C_CALLER_GPU_OK pairs=%llu image_sum=%.3f alpha_sum=%.1f positive_dL_dopacity=%d\n", (unsigned long long)n_pairs, sum, asum, pos);
  return 0;
}
#endif

static int run_nogpu(void) {
  if (gsr_version() != GSR_VERSION) return 1;
  if (!strstr(gsr_strerror(GSR_EINVAL), "invalid argument")) return 1;
  if (!strstr(gsr_strerror(GSR_ESCRATCH), "scratch")) return 1;
  if (gsr_num_tiles(1024, 1024) != 4096u || gsr_num_tiles(17, 33) != 2u * 3u) return 1;
  if (gsr_num_blocks(257) != 2u) return 1;
  if (gsr_project_scratch_bytes(1000) == 0 || gsr_sort_scratch_bytes(1000, 16) == 0 || gsr_knn_scratch_bytes(1000) == 0) return 1;
  /* argument validation happens before any device work */
  GsrView v; memset(&v, 0, sizeof v);
  GsrGaussians g; memset(&g, 0, sizeof g);
  GsrGeom geom; memset(&geom, 0, sizeof geom);
  uint64_t n = 7;
  if (gsr_forward_project(NULL, &g, &geom, &n, NULL, NULL) != GSR_EINVAL) return 1;
  v.P = 4; v.image_height = 16; v.image_width = 16; v.tanfovx = v.tanfovy = 1.0f;      /* NULL camera tensors */
  if (gsr_forward_project(&v, &g, &geom, &n, NULL, NULL) != GSR_EINVAL) return 1;
  if (gsr_backward(&v, &g, &geom, NULL, NULL, NULL, NULL, NULL, NULL) != GSR_EINVAL) return 1;
  if (gsr_adam_step(NULL, 1, 1, 0.9, 0.999, 1e-15, 0, NULL) != GSR_EINVAL) return 1;
  if (gsr_knn_mean_dist2(NULL, -1, NULL, NULL, 0, NULL) != GSR_EINVAL) return 1;
  printf("C_CALLER_NOGPU_OK version=%d\n", gsr_version());
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "gpu")) {
#ifdef WITH_HIP
    return run_gpu();
#else
    fprintf(stderr, "built without WITH_HIP\n"); return 9;
#endif
  }
  return run_nogpu();
}
