// api.hip -- the extern "C" boundary declared in include/gsrast.h.
#include "gsr_common.h"

thread_local int gsr_tls_hip_error = 0;

// stage launchers (preprocess.hip, binning.hip, render.hip)
int gsr_launch_preprocess(const GsrView&, const GsrGaussians&, GsrGeom&, hipStream_t);
int gsr_launch_preprocess_bwd(const GsrView&, const GsrGaussians&, const GsrGeom&, const GsrGrads&, hipStream_t);
bool gsr_preprocess_views_supported(const GsrView&, const GsrGaussians&);
int gsr_launch_preprocess_views(int n_views, const GsrView* views, const GsrGaussians* gs, GsrGeom* geoms, hipStream_t);
bool gsr_preprocess_bwd_views_supported(const GsrView&, const GsrGaussians&, const GsrGrads&);
int gsr_launch_preprocess_bwd_views(int n_views, const GsrView* views, const GsrGaussians* gs, const GsrGeom* geoms,
                                    const GsrGrads* outs, hipStream_t);
int gsr_launch_depth_order(GsrGeom&, const GsrView&, hipStream_t, GsrProfile*, int batch, size_t bstride,
                           uint64_t* n_pairs_all, bool early);
uint64_t* gsr_pair_counts(const GsrGeom&, int32_t P);
bool gsr_uses_columns(const GsrView&);
int gsr_launch_binning(const GsrView&, const GsrGeom&, uint64_t cap, const uint64_t* n_dev, const uint64_t* n_dev_vis,
                       GsrBinning&, hipStream_t, GsrProfile*);
int gsr_launch_binning_batch(int n, const GsrView* views, const GsrGeom* geoms, uint64_t cap, GsrBinning* bs, hipStream_t,
                             GsrProfile*);
int gsr_launch_render_fwd(const GsrView&, const GsrGeom&, const GsrBinning&, GsrImages&, hipStream_t, GsrProfile*);
int gsr_launch_render_fwd_views(int n, const GsrView* views, const GsrGeom* geoms, const GsrBinning* bs, GsrImages* imgs,
                                hipStream_t, GsrProfile*);
int gsr_launch_render_bwd_views(int n, const GsrView* views, const GsrGeom* geoms, const GsrBinning* bs,
                                const GsrImages* imgs, const GsrImageGrads* igs, GsrGrads* outs, hipStream_t, GsrProfile*);
int gsr_launch_work_order_fwd(int n, const GsrView* views, const GsrBinning* bs, const GsrImages* imgs, hipStream_t);
int gsr_launch_work_order_bwd(int n, const GsrView* views, const GsrBinning* bs, const GsrImages* imgs, hipStream_t);
bool gsr_k8_form_restores(bool views_entry, const GsrView& v, const GsrGaussians& g, const GsrGrads& out);
int gsr_launch_render_bwd(const GsrView&, const GsrGeom&, const GsrBinning&, const GsrImages&, const GsrImageGrads&,
                          GsrGrads&, hipStream_t, GsrProfile*);

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_view(const GsrView* v) {
  if (!v) return GSR_EINVAL;
  if (v->P < 0 || v->image_height <= 0 || v->image_width <= 0) return GSR_EINVAL;
  if (v->sh_degree < 0 || v->sh_degree > 3) return GSR_EINVAL;
  if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->campos) return GSR_EINVAL;
  if (!(v->tanfovx > 0.f) || !(v->tanfovy > 0.f)) return GSR_EINVAL;
  return GSR_OK;
}

int check_scene(const GsrView* v, const GsrGaussians* g) {
  const GsrScene* sc = g->scene;
  if (g->means3D || g->opacities || g->shs || g->colors_precomp || g->scales || g->rotations || g->cov3D_precomp)
    return GSR_EINVAL;
  if (sc->n_models < 1 || sc->n_models > GSR_MAX_MODELS) return GSR_EINVAL;
  const int K = v->sh_stride, nb = (v->sh_degree + 1) * (v->sh_degree + 1);
  if (K < nb || K < 1 || K > 64) return GSR_EINVAL;
  int64_t total = 0;
  for (int m = 0; m < sc->n_models; ++m) {
    const GsrModel& md = sc->models[m];
    if (md.count < 0) return GSR_EINVAL;
    total += md.count;
    if (md.count == 0) continue;
    if (!md.xyz || !md.scaling || !md.rotation || !md.opacity || !md.features_dc) return GSR_EINVAL;
    if (K > 1 && !md.features_rest) return GSR_EINVAL;
    if (!aligned16(md.rotation)) return GSR_EINVAL;
  }
  if (total != (int64_t)v->P) return GSR_EINVAL;
  if (sc->rotations_out && !aligned16(sc->rotations_out)) return GSR_EINVAL;
  return GSR_OK;
}

int check_gaussians(const GsrView* v, const GsrGaussians* g) {
  if (!g) return GSR_EINVAL;
  if (v->P == 0) return GSR_OK;
  if (g->scene) return check_scene(v, g);
  if (!g->means3D || !g->opacities) return GSR_EINVAL;
  if ((g->shs != nullptr) == (g->colors_precomp != nullptr)) return GSR_EINVAL;
  const bool sr = g->scales != nullptr && g->rotations != nullptr;
  const bool any_sr = g->scales != nullptr || g->rotations != nullptr;
  if (g->cov3D_precomp ? any_sr : !sr) return GSR_EINVAL;
  if (g->shs) {
    const int nb = (v->sh_degree + 1) * (v->sh_degree + 1);
    if (v->sh_stride < nb || v->sh_stride > 64) return GSR_EINVAL;
    if (!aligned16(g->shs)) return GSR_EINVAL;
  }
  if (g->rotations && !aligned16(g->rotations)) return GSR_EINVAL;
  return GSR_OK;
}

// The views of a batch share their Gaussians; only `scales` may be a different tensor per view.
bool same_except_scales(const GsrGaussians& a, const GsrGaussians& b) {
  if (a.scene || b.scene) {   // scene input: the same models for every view; noise samples / returned scales may differ
    if (!a.scene || !b.scene || a.scene->n_models != b.scene->n_models) return false;
    for (int m = 0; m < a.scene->n_models; ++m) {
      const GsrModel &x = a.scene->models[m], &y = b.scene->models[m];
      if (x.count != y.count || x.xyz != y.xyz || x.scaling != y.scaling || x.rotation != y.rotation ||
          x.opacity != y.opacity || x.features_dc != y.features_dc || x.features_rest != y.features_rest)
        return false;
    }
    return true;
  }
  return a.means3D == b.means3D && a.opacities == b.opacities && a.shs == b.shs && a.colors_precomp == b.colors_precomp &&
         a.rotations == b.rotations && a.cov3D_precomp == b.cov3D_precomp &&
         (a.scales != nullptr) == (b.scales != nullptr);
}

}  // namespace

extern "C" {

int gsr_version(void) { return GSR_VERSION; }

int gsr_last_hip_error(void) { return gsr_tls_hip_error; }

const char* gsr_strerror(int code) {
  switch (code) {
    case GSR_OK: return "ok";
    case GSR_EINVAL: return "invalid argument (shape, null pointer, alignment or unsupported SH degree)";
    case GSR_ECAPACITY: return "tile-pair count does not fit 32-bit list positions";
    case GSR_EHIP: return hipGetErrorString((hipError_t)gsr_tls_hip_error);
    case GSR_ESCRATCH: return "sort scratch buffer too small";
    default: return "unknown gsrast error";
  }
}

GsrProfile* gsr_profile_create(void) { return new GsrProfile(); }

void gsr_profile_destroy(GsrProfile* p) {
  if (!p) return;
  for (int i = 0; i < p->created; ++i) {
    (void)hipEventDestroy(p->ev[i][0]);
    (void)hipEventDestroy(p->ev[i][1]);
  }
  delete p;
}

void gsr_profile_set_stage_mask(GsrProfile* p, uint32_t mask) {
  if (p) p->mask = mask;
}

void gsr_profile_set_sampling(GsrProfile* p, uint32_t every) {
  if (!p) return;
  p->every = every ? every : 1;
  for (int i = 0; i < GSR_STAGE_COUNT; ++i) p->tick[i] = 0;
}

int gsr_profile_collect(GsrProfile* p, double* ms, int64_t* counts) {
  if (!p || !ms || !counts) return GSR_EINVAL;
  for (int i = 0; i < p->n; ++i) {
    GSR_HIP(hipEventSynchronize(p->ev[i][1]));
    float t = 0.f;
    GSR_HIP(hipEventElapsedTime(&t, p->ev[i][0], p->ev[i][1]));
    ms[p->stage[i]] += (double)t;
    counts[p->stage[i]] += 1;
  }
  p->n = 0;
  return GSR_OK;
}

namespace {
struct PackViews {
  const float* bg[GSR_MAX_BATCH_VIEWS];
  const float* viewmatrix[GSR_MAX_BATCH_VIEWS];
  const float* projmatrix[GSR_MAX_BATCH_VIEWS];
  const float* campos[GSR_MAX_BATCH_VIEWS];
  float tanfovx[GSR_MAX_BATCH_VIEWS], tanfovy[GSR_MAX_BATCH_VIEWS], sh_degree[GSR_MAX_BATCH_VIEWS];
};
// one workgroup per view, one thread per float of the packed row
__global__ void __launch_bounds__(64) k_pack_views(const PackViews pv, float* __restrict__ packed) {
  const int k = blockIdx.x, t = threadIdx.x;
  float x = 0.f;
  if (t < 3) x = pv.bg[k][t];
  else if (t >= 4 && t < 20) x = pv.viewmatrix[k][t - 4];
  else if (t >= 20 && t < 36) x = pv.projmatrix[k][t - 20];
  else if (t >= 36 && t < 39) x = pv.campos[k][t - 36];
  else if (t == 40) x = pv.tanfovx[k];
  else if (t == 41) x = pv.tanfovy[k];
  else if (t == 42) x = pv.sh_degree[k];
  if (t < GSR_PACKED_VIEW_FLOATS) packed[k * GSR_PACKED_VIEW_FLOATS + t] = x;
}
}  // namespace

int gsr_pack_views(int32_t n_views, const GsrView* views, float* packed, void* stream_) {
  if (n_views < 1 || n_views > GSR_MAX_BATCH_VIEWS || !views || !packed || !aligned16(packed)) return GSR_EINVAL;
  PackViews pv = PackViews{};
  for (int k = 0; k < n_views; ++k) {
    const int rc = check_view(&views[k]);
    if (rc) return rc;
    // the same bound check_gaussians applies at capture time: a replayed view must not ask for more SH bands than stored
    if (views[k].sh_stride > 0 && (views[k].sh_degree + 1) * (views[k].sh_degree + 1) > views[k].sh_stride) return GSR_EINVAL;
    pv.bg[k] = views[k].bg; pv.viewmatrix[k] = views[k].viewmatrix; pv.projmatrix[k] = views[k].projmatrix;
    pv.campos[k] = views[k].campos;
    pv.tanfovx[k] = views[k].tanfovx; pv.tanfovy[k] = views[k].tanfovy; pv.sh_degree[k] = (float)views[k].sh_degree;
  }
  GsrDeviceGuard dev(packed);
  hipLaunchKernelGGL(k_pack_views, dim3((uint32_t)n_views), dim3(64), 0, (hipStream_t)stream_, pv, packed);
  GSR_HIP(hipGetLastError());
  return GSR_OK;
}

static uint64_t* n_pairs_device(const GsrGeom* geom, int32_t P) { return gsr_pair_counts(*geom, P); }

static int forward_project(const GsrView* v, const GsrGaussians* g, GsrGeom* geom, uint64_t* n_pairs_host,
                           void* stream_, GsrProfile* prof, bool sync) {
  int rc = check_view(v);
  if (rc) return rc;
  rc = check_gaussians(v, g);
  if (rc) return rc;
  if (!geom || !n_pairs_host) return GSR_EINVAL;
  *n_pairs_host = 0;
  if (v->P == 0) return GSR_OK;
  if (!geom->splat || !geom->radii || !geom->tiles_touched || !geom->block_offsets || !geom->scratch) return GSR_EINVAL;
  if (!aligned16(geom->splat)) return GSR_EINVAL;
  if (geom->scratch_bytes < gsr_project_scratch_bytes(v->P)) return GSR_ESCRATCH;
  // async: the word holds GSR_N_PENDING until a kernel has stored the count (gsrast.h: the caller may poll it)
  if (!sync) *n_pairs_host = GSR_N_PENDING;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(geom->splat);
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_PREPROCESS);
    rc = gsr_launch_preprocess(*v, *g, *geom, stream);
    if (rc) return rc;
  }
  uint64_t* n_dev = n_pairs_device(geom, v->P);
  // async + column path: n_pairs_host is page-locked and k_col_plan stores N there itself (no copy operation)
  const bool direct = !sync && gsr_uses_columns(*v);
  rc = gsr_launch_depth_order(*geom, *v, stream, prof, 1, 0, direct ? n_pairs_host : nullptr, /*early=*/true);
  if (rc) return rc;
  if (!direct) GSR_HIP(hipMemcpyAsync(n_pairs_host, n_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  if (sync) {
    GSR_HIP(hipStreamSynchronize(stream));
    if (*n_pairs_host >= (1ull << 32)) return GSR_ECAPACITY;
  }
  return GSR_OK;
}

int gsr_forward_project_batch(int32_t n_views, const GsrView* views, const GsrGaussians* gs, GsrGeom* geoms,
                              uint64_t* n_pairs_pinned, void* stream_, GsrProfile* prof) {
  if (n_views < 1 || n_views > GSR_MAX_BATCH_VIEWS || !views || !gs || !geoms || !n_pairs_pinned) return GSR_EINVAL;
  const GsrGaussians* g = &gs[0];
  const GsrView& v0 = views[0];
  size_t bstride = 0;
  for (int k = 0; k < n_views; ++k) {
    int rc = check_view(&views[k]);
    if (rc) return rc;
    rc = check_gaussians(&views[k], &gs[k]);
    if (rc) return rc;
    if (views[k].P != v0.P || views[k].image_height != v0.image_height || views[k].image_width != v0.image_width ||
        views[k].sh_stride != v0.sh_stride || !same_except_scales(gs[k], gs[0]))
      return GSR_EINVAL;
    const GsrGeom& ge = geoms[k];
    n_pairs_pinned[k] = 0;
    if (v0.P == 0) continue;
    if (!ge.splat || !ge.radii || !ge.tiles_touched || !ge.block_offsets || !ge.scratch || !aligned16(ge.splat))
      return GSR_EINVAL;
    if (ge.scratch_bytes < gsr_project_scratch_bytes(v0.P)) return GSR_ESCRATCH;
    if (k == 1) bstride = (size_t)((char*)ge.scratch - (char*)geoms[0].scratch);
    if (k >= 1 && ((char*)ge.scratch != (char*)geoms[0].scratch + (size_t)k * bstride || bstride < geoms[0].scratch_bytes ||
                   (bstride & 255u)))
      return GSR_EINVAL;   // the views' projection scratch buffers must be equally spaced
  }
  if (v0.P == 0) return GSR_OK;
  for (int k = 0; k < n_views; ++k) n_pairs_pinned[k] = GSR_N_PENDING;   // until a kernel has stored view k's count
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(geoms[0].splat);
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_PREPROCESS);
    bool same_view_consts = true;
    for (int k = 1; k < n_views; ++k)
      same_view_consts = same_view_consts && views[k].scale_modifier == v0.scale_modifier;
    if (n_views > 1 && same_view_consts && gsr_preprocess_views_supported(v0, *g)) {
      const int rc = gsr_launch_preprocess_views(n_views, views, gs, geoms, stream);
      if (rc) return rc;
    } else {
      for (int k = 0; k < n_views; ++k) {
        const int rc = gsr_launch_preprocess(views[k], gs[k], geoms[k], stream);
        if (rc) return rc;
      }
    }
  }
  // k_col_plan stores the views' counts straight into the caller's page-locked array (no copy operation)
  int rc = gsr_launch_depth_order(geoms[0], v0, stream, prof, n_views, bstride, n_pairs_pinned, /*early=*/false);
  if (rc) return rc;
  for (int k = 1; k < n_views; ++k)
    geoms[k].sorted_idx = reinterpret_cast<uint32_t*>((char*)geoms[0].sorted_idx + (size_t)k * bstride);
  return GSR_OK;
}

int gsr_forward_project(const GsrView* v, const GsrGaussians* g, GsrGeom* geom, uint64_t* n_pairs_host, void* stream_,
                        GsrProfile* prof) {
  return forward_project(v, g, geom, n_pairs_host, stream_, prof, true);
}

int gsr_forward_project_async(const GsrView* v, const GsrGaussians* g, GsrGeom* geom, uint64_t* n_pairs_pinned,
                              void* stream_, GsrProfile* prof) {
  return forward_project(v, g, geom, n_pairs_pinned, stream_, prof, false);
}

static int check_render(const GsrView* v, const GsrGeom* geom, uint64_t n_pairs, const GsrBinning* b,
                        const GsrImages* img) {
  int rc = check_view(v);
  if (rc) return rc;
  if (!geom || !b || !img) return GSR_EINVAL;
  if (!b->ranges || !img->color || !img->depth_alpha || !img->final_T || !img->n_contrib) return GSR_EINVAL;
  if (!b->tile_work || !img->tile_depth) return GSR_EINVAL;      // (img->ckpt may be NULL: forward only, gsrast.h)
  if (b->seg_len != 0u && b->seg_len != 64u && b->seg_len != 128u && b->seg_len != 256u) return GSR_EINVAL;
  if (b->fwd_mode == 1 && gsr_seg_len(*b) != 256u) return GSR_EINVAL;     // the whole-tile forward checkpoints every 256 entries
  if ((uint64_t)b->bwd_items_cap < n_pairs / gsr_seg_len(*b) + gsr_num_tiles(v->image_height, v->image_width)) return GSR_EINVAL;
  if (n_pairs && (!b->point_list || !geom->splat)) return GSR_EINVAL;
  if (v->P > 0 && (!geom->scratch || geom->scratch_bytes < gsr_project_scratch_bytes(v->P))) return GSR_ESCRATCH;
  if (n_pairs >= (1ull << 32)) return GSR_ECAPACITY;
  return GSR_OK;
}

static int render_binning(const GsrView* v, const GsrGeom* geom, uint64_t n_pairs, GsrBinning* b, hipStream_t stream,
                          GsrProfile* prof) {
  if (v->P == 0) n_pairs = 0;
  const uint64_t* n_dev = (b->count_on_device && v->P > 0) ? n_pairs_device(geom, v->P) : nullptr;
  const uint64_t* n_vis = v->P > 0 ? n_pairs_device(geom, v->P) + 1 : nullptr;
  return gsr_launch_binning(*v, *geom, n_pairs, n_dev, n_vis, *b, stream, prof);
}

int gsr_forward_render(const GsrView* v, const GsrGeom* geom, uint64_t n_pairs, GsrBinning* b, GsrImages* img,
                       void* stream_, GsrProfile* prof) {
  int rc = check_render(v, geom, n_pairs, b, img);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(img->color);
  rc = render_binning(v, geom, n_pairs, b, stream, prof);
  if (rc) return rc;
  rc = gsr_launch_work_order_fwd(1, v, b, img, stream);
  if (rc) return rc;
  return gsr_launch_render_fwd(*v, *geom, *b, *img, stream, prof);
}

int gsr_forward_render_batch(int32_t n_views, const GsrView* views, const GsrGeom* geoms, uint64_t n_pairs,
                             GsrBinning* bs, GsrImages* imgs, void* stream_, GsrProfile* prof) {
  if (n_views < 1 || n_views > GSR_MAX_BATCH_VIEWS || !views || !geoms || !bs || !imgs) return GSR_EINVAL;
  for (int k = 0; k < n_views; ++k) {
    const int rc = check_render(&views[k], &geoms[k], n_pairs, &bs[k], &imgs[k]);
    if (rc) return rc;
    if (views[k].image_height != views[0].image_height || views[k].image_width != views[0].image_width) return GSR_EINVAL;
  }
  // score_mode 0 leaves u32 pixel COUNTS in important_score and converts them in place once per view (k_score_finalize): two
  // views adding into one buffer would have the second conversion reinterpret the first view's floats as counts. Sums over
  // views take score_mode 2 (raw counts, converted by the caller) or 1 (float atomics).
  for (int k = 0; k < n_views; ++k)
    for (int j = 0; j < k; ++j)
      if (imgs[k].important_score && imgs[k].important_score == imgs[j].important_score && views[k].score_mode == 0)
        return GSR_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(imgs[0].color);
  // capacity mode + equally spaced scratch buffers: emission and the ty pass of all views share their launches
  bool all_dev = true;
  for (int k = 0; k < n_views; ++k) all_dev = all_dev && bs[k].count_on_device && views[k].P == views[0].P;
  // (the batched launcher answers GSR_EINVAL before launching anything when the layout does not qualify)
  int brc = all_dev ? gsr_launch_binning_batch(n_views, views, geoms, n_pairs, bs, stream, prof) : GSR_EINVAL;
  if (brc != GSR_OK && brc != GSR_EINVAL) return brc;
  if (brc == GSR_EINVAL) {
    for (int k = 0; k < n_views; ++k) {
      const int rc = render_binning(&views[k], &geoms[k], n_pairs, &bs[k], stream, prof);
      if (rc) return rc;
    }
  }
  int rc = gsr_launch_work_order_fwd(n_views, views, bs, imgs, stream);   // the work lists of all views in one launch
  if (rc) return rc;
  // K6 of all views in one launch when they use the same variant / outputs
  bool uniform = true;
  for (int k = 1; k < n_views; ++k)
    uniform = uniform && bs[k].fwd_mode == bs[0].fwd_mode && gsr_seg_len(bs[k]) == gsr_seg_len(bs[0]) &&
              (imgs[k].important_score != nullptr) == (imgs[0].important_score != nullptr) &&
              views[k].score_mode == views[0].score_mode;
  if (uniform) return gsr_launch_render_fwd_views(n_views, views, geoms, bs, imgs, stream, prof);
  for (int k = 0; k < n_views; ++k) {
    rc = gsr_launch_render_fwd(views[k], geoms[k], bs[k], imgs[k], stream, prof);
    if (rc) return rc;
  }
  return GSR_OK;
}

static int check_backward(const GsrView* v, const GsrGaussians* g, const GsrGeom* geom, const GsrBinning* b,
                          const GsrImages* img, const GsrImageGrads* ig, const GsrGrads* out) {
  int rc = check_view(v);
  if (rc) return rc;
  rc = check_gaussians(v, g);
  if (rc) return rc;
  if (!geom || !b || !img || !ig || !out) return GSR_EINVAL;
  if (v->P == 0) return GSR_OK;
  if (!ig->dL_dcolor || !ig->dL_ddepth_alpha || !img->final_T || !img->n_contrib || !b->ranges) return GSR_EINVAL;
  if (!b->tile_work || !img->tile_depth || !img->ckpt || !img->color || !img->depth_alpha) return GSR_EINVAL;
  if (!out->partials || !aligned16(out->partials)) return GSR_EINVAL;
  if (out->scratch_clean && !out->reach) return GSR_EINVAL;
  if (out->zero_outside & ~3) return GSR_EINVAL;
  if (out->zero_outside && (!out->reached_mask || out->accumulate)) return GSR_EINVAL;   // (the mask is what says which rows to clear)
  if (out->reach && (reinterpret_cast<uintptr_t>(out->reach) & 7u)) return GSR_EINVAL;
  {
    const int n_stat = (out->stat_max_radii2D != nullptr) + (out->stat_xyz_gradient_accum != nullptr) +
                       (out->stat_denom != nullptr);
    if (n_stat != 0 && n_stat != 3) return GSR_EINVAL;
  }
  if (g->scene) {
    if (!out->scene || !out->dL_dmeans2D) return GSR_EINVAL;
    if (out->dL_dmeans3D || out->dL_dopacities || out->dL_dshs || out->dL_dcolors || out->dL_dscales ||
        out->dL_drotations || out->dL_dcov3D)
      return GSR_EINVAL;
    for (int m = 0; m < g->scene->n_models; ++m)
      if (out->scene->models[m].rotation && !aligned16(out->scene->models[m].rotation)) return GSR_EINVAL;
  } else if (!out->dL_dmeans3D || !out->dL_dmeans2D || !out->dL_dopacities) return GSR_EINVAL;
  if (out->dL_dshs && (!g->shs || !aligned16(out->dL_dshs))) return GSR_EINVAL;
  if (out->dL_dcolors && !g->colors_precomp) return GSR_EINVAL;
  if ((out->dL_dscales || out->dL_drotations) && g->cov3D_precomp) return GSR_EINVAL;
  if (out->dL_dcov3D && !g->cov3D_precomp) return GSR_EINVAL;
  if (out->dL_drotations && !aligned16(out->dL_drotations)) return GSR_EINVAL;
  return GSR_OK;
}

// The scratch of one view (GsrGrads.partials + .reach) -> zeros
static int clear_scratch(const GsrView* v, const GsrGrads* out, hipStream_t stream, bool partials = true) {
  if (partials) GSR_HIP(gsr_zero_async(out->partials, (size_t)v->P * GSR_PARTIAL_WORDS * sizeof(float), stream));
  if (out->reach) GSR_HIP(gsr_zero_async(out->reach, ((size_t)v->P + 63) / 64 * 8, stream));
  return GSR_OK;
}

static int backward_render(const GsrView* v, const GsrGeom* geom, const GsrBinning* b, const GsrImages* img,
                           const GsrImageGrads* ig, GsrGrads* out, hipStream_t stream, GsrProfile* prof,
                           bool clear_partials = true) {
  if (!out->scratch_clean) {
    const int rc = clear_scratch(v, out, stream, clear_partials);
    if (rc) return rc;
  }
  return gsr_launch_render_bwd(*v, *geom, *b, *img, *ig, *out, stream, prof);
}

int gsr_backward(const GsrView* v, const GsrGaussians* g, const GsrGeom* geom, const GsrBinning* b,
                 const GsrImages* img, const GsrImageGrads* ig, GsrGrads* out, void* stream_, GsrProfile* prof) {
  int rc = check_backward(v, g, geom, b, img, ig, out);
  if (rc) return rc;
  if (v->P == 0) return GSR_OK;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(out->partials);
  rc = gsr_launch_work_order_bwd(1, v, b, img, stream);
  if (rc) return rc;
  rc = backward_render(v, geom, b, img, ig, out, stream, prof);
  if (rc) return rc;
  {
    GsrStageTimer t(prof, stream, GSR_STAGE_PREPROCESS_BWD);
    rc = gsr_launch_preprocess_bwd(*v, *g, *geom, *out, stream);
    if (rc) return rc;
  }
  if (out->scratch_clean && !gsr_k8_form_restores(false, *v, *g, *out)) return clear_scratch(v, out, stream);
  return GSR_OK;
}

int gsr_backward_views(int32_t n_views, const GsrView* views, const GsrGaussians* gs, const GsrGeom* geoms,
                       const GsrBinning* bs, const GsrImages* imgs, const GsrImageGrads* igs, GsrGrads* outs,
                       void* stream_, GsrProfile* prof) {
  if (n_views < 1 || n_views > GSR_MAX_BATCH_VIEWS || !views || !gs || !geoms || !bs || !imgs || !igs || !outs)
    return GSR_EINVAL;
  const GsrGaussians* g = &gs[0];
  bool per_view_scales = false;
  for (int k = 0; k < n_views; ++k) {
    const int rc = check_backward(&views[k], &gs[k], &geoms[k], &bs[k], &imgs[k], &igs[k], &outs[k]);
    if (rc) return rc;
    if (!same_except_scales(gs[k], gs[0])) return GSR_EINVAL;
    per_view_scales = per_view_scales || gs[k].scales != gs[0].scales;
    if (views[k].P != views[0].P || views[k].image_height != views[0].image_height ||
        views[k].image_width != views[0].image_width || views[k].sh_stride != views[0].sh_stride ||
        views[k].scale_modifier != views[0].scale_modifier)
      return GSR_EINVAL;
  }
  if (views[0].P == 0) return GSR_OK;
  hipStream_t stream = (hipStream_t)stream_;
  GsrDeviceGuard dev(outs[0].partials);
  if (per_view_scales)       // every view then needs its own scale gradient buffer
    for (int k = 0; k < n_views; ++k)
      for (int j = 0; j < k; ++j)
        if (!outs[k].dL_dscales || outs[k].dL_dscales == outs[j].dL_dscales) return GSR_EINVAL;
  const bool fused = n_views > 1 && gsr_preprocess_bwd_views_supported(views[0], *g, outs[0]);
  if (per_view_scales && !fused) return GSR_EINVAL;   // per-view scales are only supported by the fused pass
  // the views' partials usually are the rows of one [n_views, P, 32] tensor: one clear instead of n_views
  const size_t pbytes = (size_t)views[0].P * GSR_PARTIAL_WORDS * sizeof(float);
  bool contiguous = true;
  for (int k = 1; k < n_views; ++k)
    contiguous = contiguous && ((char*)outs[k].partials == (char*)outs[0].partials + (size_t)k * pbytes);
  // GsrGrads.scratch_clean: the caller hands over zeroed scratch and gets it back zeroed -- nothing to clear (all views or none)
  const bool clean = outs[0].scratch_clean != 0;
  for (int k = 1; k < n_views; ++k)
    if ((outs[k].scratch_clean != 0) != clean || (outs[k].reach != nullptr) != (outs[0].reach != nullptr)) return GSR_EINVAL;
  if (clean) contiguous = false;
  if (contiguous) GSR_HIP(gsr_zero_async(outs[0].partials, pbytes * (size_t)n_views, stream));
  {
    const int rc = gsr_launch_work_order_bwd(n_views, views, bs, imgs, stream);   // all views' work lists in one launch
    if (rc) return rc;
  }
  if (fused) {
    // K7 of all views in one launch
    if (!clean)
      for (int k = 0; k < n_views; ++k) {
        const int rc = clear_scratch(&views[k], &outs[k], stream, !contiguous);
        if (rc) return rc;
      }
    bool one_launch = true;      // (views rendered with different item lengths / forward variants: one K7 launch per view)
    for (int k = 1; k < n_views; ++k)
      one_launch = one_launch && gsr_seg_len(bs[k]) == gsr_seg_len(bs[0]) && bs[k].fwd_mode == bs[0].fwd_mode;
    for (int k = 0; k < (one_launch ? 1 : n_views); ++k) {
      const int rc = gsr_launch_render_bwd_views(one_launch ? n_views : 1, &views[k], &geoms[k], &bs[k], &imgs[k], &igs[k], &outs[k],
                                                 stream, prof);
      if (rc) return rc;
    }
  } else {
    for (int k = 0; k < n_views; ++k) {
      const int rc = backward_render(&views[k], &geoms[k], &bs[k], &imgs[k], &igs[k], &outs[k], stream, prof, !contiguous);
      if (rc) return rc;
      // unsupported combination: K8 view by view, views 1.. added to what view 0 wrote
      GsrGrads o = outs[k];
      const GsrGrads& o0 = outs[0];
      o.dL_dmeans3D = o0.dL_dmeans3D; o.dL_dopacities = o0.dL_dopacities; o.dL_dshs = o0.dL_dshs;
      o.dL_dcolors = o0.dL_dcolors; o.dL_drotations = o0.dL_drotations;
      if (!per_view_scales) o.dL_dscales = o0.dL_dscales;
      o.dL_dcov3D = o0.dL_dcov3D;      // (o.scene stays view k's own table: same model tensors, its own dL_dscales_out)
      o.accumulate = (k > 0) ? 1 : o0.accumulate;
      GsrStageTimer t(prof, stream, GSR_STAGE_PREPROCESS_BWD);
      const int rc2 = gsr_launch_preprocess_bwd(views[k], gs[k], geoms[k], o, stream);
      if (rc2) return rc2;
      if (clean && !gsr_k8_form_restores(false, views[k], gs[k], o)) {
        const int rc3 = clear_scratch(&views[k], &outs[k], stream);
        if (rc3) return rc3;
      }
    }
  }
  if (fused) {
    {
      GsrStageTimer t(prof, stream, GSR_STAGE_PREPROCESS_BWD);
      const int rc = gsr_launch_preprocess_bwd_views(n_views, views, gs, geoms, outs, stream);
      if (rc) return rc;
    }
    if (clean && !gsr_k8_form_restores(true, views[0], *g, outs[0]))
      for (int k = 0; k < n_views; ++k) {
        const int rc = clear_scratch(&views[k], &outs[k], stream);
        if (rc) return rc;
      }
  }
  return GSR_OK;
}

}  // extern "C"
