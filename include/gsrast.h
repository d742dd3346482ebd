/* gsrast.h -- C ABI of libgsrast.so, the MI355X-native differentiable 3D Gaussian splatting rasterizer.
 *
 * This is the boundary a binding of the reference would target. The reference (DreamScene) reaches its
 * rasterizer through the Python package `diff_gaussian_rasterization` (scene_gaussian.py:11-12), whose
 * native half is `diff_gaussian_rasterization._C` (un-vendored CUDA, README.md:47-51). The entry points below
 * replace that native half one for one:
 *
 *   gsr_forward_project + gsr_forward_render   <->  _C.rasterize_gaussians            (the forward the
 *        autograd.Function behind GaussianRasterizer.forward calls; call sites scene_gaussian.py:637-646,
 *        861-870, 1012-1021; outputs image / radii / depth_alpha [/ important_score], :637, :1012)
 *   gsr_backward                               <->  _C.rasterize_gaussians_backward   (invoked by
 *        loss.backward(), training/object_trainer.py:382, training/scene_trainer.py:881)
 *   GsrView                                    <->  GaussianRasterizationSettings     (12 fields,
 *        scene_gaussian.py:951-964)
 *
 * Rules of the boundary:
 *   - plain C: POD structs of raw DEVICE pointers and sizes, no torch / C++ types;
 *   - the caller owns every buffer (inputs, outputs, state saved for backward, scratch); the library
 *     allocates nothing that outlives a call and keeps no mutable global state: re-entrant, any thread;
 *   - all work is enqueued on the caller's stream (a hipStream_t passed as void*); the only host
 *     synchronisation is inside gsr_forward_project (it returns the data-dependent pair count N);
 *   - every entry point returns 0 or a negative GSR_E* code, never throws; gsr_strerror() explains;
 *   - fp32 contiguous tensors only; layouts are those of the reference call sites, quoted per field.
 */
#ifndef GSRAST_H
#define GSRAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 1
#define GSR_TILE 16 /* 16x16 pixel tiles */

enum {
  GSR_OK = 0,
  GSR_EINVAL = -1,    /* bad shape / null pointer / unsupported SH degree            */
  GSR_ECAPACITY = -2, /* pair count N does not fit the 32-bit list positions         */
  GSR_EHIP = -3,      /* a HIP call failed; gsr_last_hip_error() has the hipError_t  */
  GSR_ESCRATCH = -4   /* scratch buffer too small for this (P, N)                    */
};

/* Per-view configuration == GaussianRasterizationSettings (scene_gaussian.py:951-964). The four tensors
 * stay on the device exactly as RCamera holds them (utils/cam_utils.py:196-210): row-vector convention,
 * i.e. viewmatrix is the TRANSPOSED world-to-camera matrix, projmatrix = viewmatrix @ projection^T. */
typedef struct GsrView {
  int32_t P;            /* number of Gaussians                                                        */
  int32_t sh_stride;    /* K: SH coefficients stored per Gaussian = (max_sh_degree+1)^2; 0 if no SHs   */
  int32_t sh_degree;    /* D: active degree 0..3 (raster_settings.sh_degree)                           */
  int32_t image_height, image_width;
  float tanfovx, tanfovy, scale_modifier;
  int32_t prefiltered;  /* accepted for API parity; no effect (frustum test always runs)               */
  int32_t score_mode;   /* important_score weight: 0 = opacity per contributing (pixel,splat), 1 = alpha*T, 2 = RAW COUNTS:
                           the number of contributing pixels as uint32 in the (zero-initialised) score buffer, no opacity
                           factor -- for callers that add many views into one buffer and weight once at the end (mode 0
                           counts the same way internally, integer atomics, and multiplies by the opacity itself) */
  const float* bg;         /* device f32[3]  */
  const float* viewmatrix; /* device f32[16] */
  const float* projmatrix; /* device f32[16] */
  const float* campos;     /* device f32[3]  */
  const float* dynamic;    /* NULL, or device f32[4] = (tanfovx, tanfovy, sh_degree, reserved): when set, the kernels take
                              these three from HERE at the time they run instead of the by-value fields above -- with bg /
                              viewmatrix / projmatrix / campos already being device tensors, every per-camera quantity then
                              lives in device memory and a captured graph of a step (hipGraph) can be replayed with new
                              cameras by rewriting that memory (gsr_pack_views). The by-value fields must still be valid
                              (they are what the host-side checks see) */
} GsrView;

/* ---- gather-free multi-model input with the activations fused into K1 / K8 -------
 * `scene_render` (scene_gaussian.py:673-893) concatenates the leaf tensors of every visible GaussianModel with
 * torch.cat on every call (:753-843), after running their activations (gs_renderer.py:464-488: exp, normalize,
 * sigmoid, cat(features_dc, features_rest)) and, when training, the noise augmentations (:844-852). A GsrScene hands
 * the RAW leaf tensors of up to GSR_MAX_MODELS models to K1 / K8 instead: Gaussian i of the concatenated index space
 * is row i - first(m) of model m, nothing is copied, and culled Gaussians never touch their SH rows. */
#define GSR_MAX_MODELS 16
typedef struct GsrModel {      /* leaf tensors of one GaussianModel (gs_renderer.py:185-203), all on the device */
  int32_t count;               /* Gaussians in this model                                                   */
  int32_t reserved_;
  const float* xyz;            /* [count,3]      _xyz                                                      */
  const float* scaling;        /* [count,3]      _scaling  (log scale;   activation exp,       :465-466)    */
  const float* rotation;       /* [count,4]      _rotation (raw w,x,y,z; activation normalize, :469-470)    */
  const float* opacity;        /* [count,1]      _opacity  (logit;       activation sigmoid,   :487-488)    */
  const float* features_dc;    /* [count,1,3]    _features_dc                                               */
  const float* features_rest;  /* [count,K-1,3]  _features_rest (NULL if K == 1)                            */
} GsrModel;
typedef struct GsrScene {
  int32_t n_models;            /* 1..GSR_MAX_MODELS; GsrView.P must equal the sum of the counts, sh_stride = K */
  int32_t reserved_;
  GsrModel models[GSR_MAX_MODELS];
  const float* scale_noise;    /* NULL or [P,3]   N(0,1) samples: scales <- max(0, s + n * (sqrt(0.2) * s / 4)) (:849-852) */
  const float* sh_noise;       /* NULL or [P,K,3] N(0,1) samples: shs <- shs + n * (sqrt(0.2) * shs)           (:844-847) */
  float* scales_out;           /* NULL or [P,3]: the activated (and augmented) scales scene_render returns (:892) */
  float* rotations_out;        /* NULL or [P,4]: normalised quaternions (parity tests)                        */
  float* opacities_out;        /* NULL or [P]:   sigmoid(opacity)        (parity tests)                        */
} GsrScene;
/* Gradients w.r.t. the RAW leaf tensors of each model (same shapes). NULL entries are skipped. */
typedef struct GsrModelGrads {
  float *xyz, *scaling, *rotation, *opacity, *features_dc, *features_rest;
} GsrModelGrads;
typedef struct GsrSceneGrads {
  GsrModelGrads models[GSR_MAX_MODELS];
  const float* dL_dscales_out; /* NULL or [P,3]: gradient arriving through GsrScene.scales_out (the trainers put a loss
                                  on the returned scales, object_trainer.py:378-379); chained into `scaling` */
} GsrSceneGrads;

/* Inputs of GaussianRasterizer.forward (scene_gaussian.py:1012-1021). Exactly one of shs / colors_precomp and
 * exactly one of (scales, rotations) / cov3D_precomp is non-NULL -- or `scene` is set and all of them are NULL. */
typedef struct GsrGaussians {
  const float* means3D;        /* [P,3]                                            */
  const float* opacities;      /* [P] (the reference passes [P,1])                  */
  const float* shs;            /* [P,K,3] coefficient-major (gs_renderer.py:480-484) */
  const float* colors_precomp; /* [P,3]                                            */
  const float* scales;         /* [P,3] post-activation                            */
  const float* rotations;      /* [P,4] (w,x,y,z), already normalised              */
  const float* cov3D_precomp;  /* [P,6] [xx,xy,xz,yy,yz,zz] (gs_renderer.py:79-88)  */
  const GsrScene* scene;       /* host pointer or NULL: raw multi-model input (see GsrScene) */
} GsrGaussians;

/* Projected per-Gaussian state: written by gsr_forward_project, read by render and backward (save it).
 * splat holds 12 floats per Gaussian as three float4 rows q0,q1,q2 (row-major [P][12]):
 *   q0 = (x_pix, y_pix, conic_a, conic_b)   q1 = (conic_c, opacity, view_depth, r)   q2 = (g, b, tau, 0)
 * (tau = 2 ln(255 opacity), slightly inflated: the level of the conic form inside which alpha can reach 1/255;
 *  < 0 if nowhere; used only to skip work)
 * Rows of culled Gaussians (radii == 0) are left unwritten. */
typedef struct GsrGeom {
  float* splat;            /* [P,12], 16-byte aligned */
  int32_t* radii;          /* [P]  screen radius in pixels, 0 = culled (output `radii`)            */
  uint32_t* tiles_touched; /* [P]  number of 16x16 tiles overlapped                               */
  uint32_t* block_offsets; /* [gsr_num_blocks(P)+8], 8-byte aligned; grids of more than 256 x 256 tiles only: exclusive
                              scan of the tile counts of each run of 256 Gaussians IN DEPTH ORDER, [nb] = N (low 32 bits) */
  void* scratch;           /* gsr_project_scratch_bytes(P) bytes, 256-byte aligned: depth-sort buffers, column counts
                              and the device-side counts (N, visible Gaussians). Must stay alive until
                              gsr_forward_render has been enqueued; not needed for backward                       */
  size_t scratch_bytes;
  uint32_t* sorted_idx;    /* OUT (set by gsr_forward_project*), OPAQUE: one of the depth sort's two value buffers inside
                              `scratch`, handed from gsr_forward_project* to gsr_forward_render*. It holds the Gaussian
                              indices in (depth bits, index) order ONLY when the sort's last pass moved keys; a last pass
                              that is the identity (the top byte of one object's depths) is skipped, the order then lives
                              in the OTHER buffer and a flag word in `scratch` says so -- the library's consumers read the
                              flag, a caller must not read this pointer (the depth order is observable through
                              GsrBinning.point_list, whose entries are in (tile, depth bits, index) order)           */
} GsrGeom;

/* Tile binning. point_list / ranges are saved for backward; the rest is scratch for the forward only. */
typedef struct GsrBinning {
  uint32_t* point_list; /* [N] Gaussian index per (tile, depth)-sorted pair                         */
  uint32_t* ranges;     /* [tiles,2] (start,end) into point_list; (0,0) for an empty tile            */
  uint32_t* tile_work;  /* [tiles + 2 + 2*bwd_items_cap] u32: forward work list (tile ids, heaviest first) followed by
                           the number of non-empty tiles, then the backward's (tile, segment) item list; written by
                           the forward and by the backward: keep it with the saved state                          */
  uint32_t bwd_items_cap; /* capacity of the backward item list: >= n_pairs / seg_len + tiles                     */
  uint32_t seg_len;       /* list entries per checkpoint of the forward = per work item of the backward: 256 (also: 0), 128
                             or 64. A backward item is a serial recurrence over its entries, so a launch cannot end before its
                             longest item has (256 entries: 65-90 us on MI355X); a launch with little total work (one view, a
                             small scene) finishes sooner with shorter items, a large one pays for the extra checkpoints and
                             prologues. Same value for the forward and the backward of a view; views of one batched
                             call that differ in it are composited by one launch each instead of one for all; fwd_mode 1
                             requires 256. Sizes: bwd_items_cap (above), GsrImages.ckpt */
  uint64_t* keys_sorted;/* [N] optional: receives the sorted 64-bit keys (tile<<32 | depth bits); may be NULL */
  void* scratch;        /* gsr_sort_scratch_bytes(N, tiles) bytes, 256-byte aligned                   */
  size_t scratch_bytes;
  int32_t count_on_device; /* 0: n_pairs passed to gsr_forward_render is the exact N (after the synchronous
                              gsr_forward_project). 1: capacity mode -- n_pairs is the CAPACITY of point_list /
                              scratch; the kernels take the true N from the device (GsrGeom.block_offsets tail)
                              and clamp it to the capacity; the caller compares N with the capacity afterwards
                              and re-runs gsr_forward_render with larger buffers if it did not fit            */
  int32_t fwd_mode;        /* forward compositing variant: 0 = four 8x8-quarter work items per tile, 4 lanes per pixel
                              (few / deep tiles); 1 = one work item per tile, one pixel per lane (thousands of
                              shallow tiles). Same semantics; a performance choice                 */
  uint32_t* stats_host;    /* optional page-locked (device-visible, e.g. hipHostMalloc / torch pinned) host word: a
                              kernel stores the number of non-empty tiles of this view there, the statistic a caller
                              can base the next call's fwd_mode on                                                */
} GsrBinning;

/* Per-pixel outputs (scene_gaussian.py:1012,1023) and the per-pixel state backward needs. */
typedef struct GsrImages {
  float* color;           /* [3,H,W]                                                         */
  float* depth_alpha;     /* [2,H,W]: plane 0 = sum z_i a_i T_i, plane 1 = sum a_i T_i        */
  float* final_T;         /* [H,W] transmittance after the last contributor                  */
  uint32_t* n_contrib;    /* [H,W] 1-based list position of the last contributor             */
  uint32_t* tile_depth;   /* [tiles] max of n_contrib over the tile (written by forward, read by backward) */
  float* ckpt;            /* [n_pairs/seg_len + 1][6][256] per-pixel prefix state (T, C rgb, depth, alpha) at the
                             seg_len-entry boundaries of the tile lists (slot = absolute list position / seg_len;
                             GsrBinning.seg_len), written by the forward as far as it composites, read by the backward.
                             NULL = FORWARD ONLY: the forward writes no checkpoints (inference renders, importance
                             scores: ~5 % of K6 and the largest region of the saved state); gsr_backward* then refuses
                             the view with GSR_EINVAL */
  float* important_score; /* [P], or NULL (score_flag False). MUST BE ALL ZERO ON ENTRY: K6 adds to it. score_mode 1: float
                             sums (several views may add into one buffer); score_mode 2: u32 pixel counts (bit patterns; several
                             views may add into one buffer, the caller converts); score_mode 0: u32 counts converted IN PLACE
                             to opacity x count at the end of the call -- one buffer per view (views of one batched call that
                             share a buffer are refused with GSR_EINVAL: the second conversion would read the first one's
                             floats as counts)                                                                             */
} GsrImages;

typedef struct GsrImageGrads {
  const float* dL_dcolor;       /* [3,H,W] */
  const float* dL_ddepth_alpha; /* [2,H,W] */
} GsrImageGrads;

/* Gradients w.r.t. the forward inputs. NULL = not wanted (must be NULL where the input was NULL).
 * Every non-NULL output is fully overwritten (zeros for culled Gaussians) unless `accumulate`. dL_dview / dL_dproj /
 * dL_dcampos treat viewmatrix, projmatrix and campos as three independent inputs. */
typedef struct GsrGrads {
  float* dL_dmeans3D;   /* [P,3]                                                                     */
  float* dL_dmeans2D;   /* [P,3] (d/d ndc_x, d/d ndc_y, 0): what lands in viewspace_points.grad     */
  float* dL_dopacities; /* [P]                                                                       */
  float* dL_dshs;       /* [P,K,3]                                                                   */
  float* dL_dcolors;    /* [P,3]                                                                     */
  float* dL_dscales;    /* [P,3]                                                                     */
  float* dL_drotations; /* [P,4]                                                                     */
  float* dL_dcov3D;     /* [P,6]                                                                     */
  float* dL_dview;      /* [16] or NULL; zero-initialised by the caller                              */
  float* dL_dproj;      /* [16] or NULL; zero-initialised by the caller                              */
  float* dL_dcampos;    /* [3]  or NULL; zero-initialised by the caller                              */
  float* partials;      /* scratch [P,32] 32-bit words (128 B / Gaussian, 16-byte aligned) = 16 f64 per Gaussian, 12 used:
                           sum q u, sum q v, sum q u^2, sum q u v, sum q v^2, dL/dopacity, dL/dr, dL/dg, dL/db, dL/ddepth
                           (the last two as two half-wave sums each: doubles 8-11). K7 reduces a splat's sums over the 64
                           pixels of a wave in fp32 in a fixed order and adds the wave's result ACROSS waves in double
                           (one global_atomic_add_f64 instruction, 12 lanes, one 128-byte line). A double holds the sum of
                           fp32 addends exactly while their exponents span less than 2^29: within that span the order in
                           which K7's workgroups arrive does not show in the gradients (bit-reproducible backward); an
                           addend more than 2^29 below the running sum can move the double by one unit of its last place,
                           which reaches the fp32 result only on a rounding tie (about one value in 10^9)            */
  int32_t accumulate;   /* 0: overwrite the parameter gradients; 1: ADD this view's gradients to what the buffers
                           hold (device-side sum over the views of one optimizer step). dL_dmeans2D is per view
                           and always overwritten; dL_dview/proj/campos always accumulate                     */
  int32_t reserved_;
  /* SURVEY.md section 8(f) rank 3, densification statistics fused into K8 (all three NULL or all three set, [P] fp32):
   * what the trainers do after backward with the view's radii / visibility_filter / viewspace_points.grad
   * (object_trainer.py:386-390, gs_renderer.py:1061-1065), for the visible Gaussians (radii > 0) of THIS view:
   *   max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += ||dL_dmeans2D[:2]||; denom += 1              */
  float* stat_max_radii2D;
  float* stat_xyz_gradient_accum;
  float* stat_denom;
  const GsrSceneGrads* scene; /* host pointer; required iff GsrGaussians.scene was given: the parameter gradients go
                                 to the models' raw leaves and dL_dmeans3D/scales/rotations/opacities/shs must be NULL */
  uint64_t* reached_mask;     /* NULL or device u64[(P + 63) / 64]: bit i % 64 of word i / 64 is set when Gaussian i MAY have
                                 received a non-zero parameter gradient from this call -- the rows a multi-GPU gradient
                                 exchange has to send (everything behind the opaque front layers of
                                 an object is exactly zero). K8 classifies the Gaussians K7 reached anyway (its sparse form);
                                 forms of K8 that do not classify set every bit. accumulate = 0: the words are overwritten;
                                 accumulate = 1: OR-ed into (the union over the views added to the gradient buffers) */
  uint64_t* reach;            /* NULL or device scratch u64[(P + 63) / 64] next to `partials`: K7 sets bit i % 64 of word
                                 i / 64 for the Gaussians it composited into this view, so that K8 finds them without
                                 reading the [P,12] sums of every visible Gaussian (96 MB of the 4-view step at 500 k
                                 Gaussians, of which 15 MB are non-zero): one word per 64 Gaussians and view */
  int32_t scratch_clean;      /* 0: `partials` (and `reach`) hold anything on entry -- the library clears them first -- and
                                 anything on return. 1 (needs `reach`): the caller keeps both buffers between calls and
                                 guarantees they are ALL ZERO on entry; the library leaves them all zero on return (K8 zeroes
                                 the rows and marks it consumed), so no clear is launched at all. A call that returns an
                                 error leaves them undefined.                                                          */
  int32_t zero_outside;       /* What the caller KNOWS about the outputs as they are on entry (accumulate = 0 only; needs reached_mask,
                                 which must then hold the mask the previous writer of these buffers left):
                                 bit 0: every row of the summed-gradient outputs (dL_dmeans3D / dL_dopacities / dL_dshs / dL_dscales
                                        / dL_drotations) OUTSIDE reached_mask is zero -- i.e. the buffers hold what the previous
                                        gsr_backward* call with these same buffers and this same reached_mask wrote, untouched;
                                 bit 1: the same for the per-view outputs of every view of the call (dL_dmeans2D, per-view dL_dscales).
                                 The sparse form of K8 then writes only the rows this call reaches and zeros over the rows the mask
                                 names but this call does not reach, instead of zeros over everything nothing reached (84 % of
                                 the rows at 500 k Gaussians @1024^2: 118 + 24 MB per 4-view step). Results identical. The other
                                 forms of K8 ignore it (they write every row and set every bit of reached_mask). 0: nothing known. */
} GsrGrads;

/* Optional per-stage timing with HIP events on the caller's stream (bench.py uses it for `roofline`). */
enum {
  GSR_STAGE_PREPROCESS = 0, GSR_STAGE_SCAN, GSR_STAGE_DUPLICATE, GSR_STAGE_SORT, GSR_STAGE_RANGES,
  GSR_STAGE_RENDER_FWD, GSR_STAGE_RENDER_BWD, GSR_STAGE_PREPROCESS_BWD, GSR_STAGE_COUNT
};
typedef struct GsrProfile GsrProfile;
GsrProfile* gsr_profile_create(void);
void gsr_profile_destroy(GsrProfile*);
/* Record only the stages whose bit (1u << GSR_STAGE_*) is set; default all. */
void gsr_profile_set_stage_mask(GsrProfile*, uint32_t mask);
/* Record only one of every `every` occurrences of each stage (bounds the cost of the two event records per occurrence
 * inside a timed region); counts[] of gsr_profile_collect reports how many were recorded. */
void gsr_profile_set_sampling(GsrProfile*, uint32_t every);
/* Synchronises the recorded events and ADDS each stage's elapsed ms into ms[GSR_STAGE_COUNT] and the
 * number of recordings into counts[]; then clears the recordings. */
int gsr_profile_collect(GsrProfile*, double* ms, int64_t* counts);

int gsr_version(void);
const char* gsr_strerror(int code);
int gsr_last_hip_error(void); /* thread-local hipError_t of the last GSR_EHIP on this thread */

size_t gsr_project_scratch_bytes(int32_t P);
size_t gsr_sort_scratch_bytes(uint64_t n_pairs, uint32_t n_tiles);
uint32_t gsr_num_tiles(int32_t image_height, int32_t image_width);
uint32_t gsr_num_blocks(int32_t P); /* entries of GsrGeom.block_offsets minus one */

/* K1 projection (cull, cov3D, EWA cov2D, conic, radius, tile rect, SH colour), stable depth sort of the P
 * Gaussians, scan of the depth-ordered tile counts.
 * Synchronises `stream` once and stores the pair count N in *n_pairs_host (ordinary host memory). */
int gsr_forward_project(const GsrView*, const GsrGaussians*, GsrGeom*, uint64_t* n_pairs_host, void* stream,
                        GsrProfile* prof);
/* Same work, no host synchronisation: N lands in *n_pairs_pinned (must be page-locked, device-visible host memory:
 * hipHostMalloc / torch pinned; a kernel stores it there) and is valid once the caller has waited for the work enqueued
 * so far. Used with capacity mode
 * (GsrBinning.count_on_device) so that a whole forward is enqueued without draining the GPU.
 * The word holds GSR_N_PENDING from the call until a kernel has stored N (P == 0: 0 at once). On grids of at most 256 x 256
 * tiles with P < 2^24 the count is stored EARLY -- by the first workgroup of the depth sort's first pass, i.e. after K1 and
 * one histogram launch, long before the sort and the column counts are over (stored once per call: a caller that has
 * read it may re-arm the word for its next call at once): a caller may poll the word (relaxed loads of page-locked memory)
 * instead of waiting for the stream, and fall back to the stream when it still reads GSR_N_PENDING after the enqueued work
 * has finished. */
#define GSR_N_PENDING UINT64_MAX
int gsr_forward_project_async(const GsrView*, const GsrGaussians*, GsrGeom*, uint64_t* n_pairs_pinned, void* stream,
                              GsrProfile* prof);

/* The same for n_views views of the SAME Gaussians (the C_batch_size views of one optimizer step,
 * object_trainer.py:302-382): the launch-latency-bound depth sorts and column counts of all views go through each
 * launch together. Requirements: equal P / image size / sh_stride, image at most 4096 x 4096, and the views'
 * GsrGeom.scratch buffers equally spaced (geoms[k].scratch == geoms[0].scratch + k * stride, stride a multiple of 256).
 * gaussians[k] are the inputs of view k: identical pointers for every view except `scales`, which may be a different
 * tensor per view (the trainers add fresh noise to the activated scales of every view, scene_gaussian.py:1004-1008);
 * with a `scene`, every view has its own GsrScene holding the SAME models and its own noise samples / scales_out.
 * K1 runs once over all views (parameter rows read once) for shs or scene input with K in {1,4,9,16}.
 * No host synchronisation: n_pairs_pinned[n_views] (page-locked, device-visible) receives the counts straight from a
 * kernel and is valid once the work enqueued so far has finished (GSR_N_PENDING until then; pollable as described at
 * gsr_forward_project_async, but stored at the END of the projection here -- the early store costs the batched launches
 * more than their GPU-bound callers gain).
 * The views then continue with gsr_forward_render_batch (or each with its own gsr_forward_render). */
#define GSR_MAX_BATCH_VIEWS 16
#define GSR_PARTIAL_WORDS 32   /* 32-bit words per Gaussian of GsrGrads.partials (16 doubles) */
int gsr_forward_project_batch(int32_t n_views, const GsrView* views, const GsrGaussians* gaussians /* [n_views] */,
                              GsrGeom* geoms, uint64_t* n_pairs_pinned, void* stream, GsrProfile* prof);

/* Camera block of n_views views in one launch: packed[k] = (bg[3], 0, viewmatrix[16], projmatrix[16], campos[3], 0,
 * tanfovx, tanfovy, sh_degree, 0) = GSR_PACKED_VIEW_FLOATS floats (every tensor 16-byte aligned), gathered from the
 * views' device tensors and by-value fields. A caller that replays a captured step points the GsrView tensors of the
 * CAPTURED views into such a block (bg = row, viewmatrix = row + 4, projmatrix = row + 20, campos = row + 36,
 * dynamic = row + 40) and refreshes it with this call from the settings of the step at hand. packed must be 16-byte
 * aligned; no host synchronisation. */
#define GSR_PACKED_VIEW_FLOATS 44
int gsr_pack_views(int32_t n_views, const GsrView* views, float* packed, void* stream);

/* K3 pair emission in depth order, K4 stable tile sort, K5 tile ranges, K6 front-to-back compositing. */
int gsr_forward_render(const GsrView*, const GsrGeom*, uint64_t n_pairs, GsrBinning*, GsrImages*, void* stream,
                       GsrProfile* prof);

/* The same for n_views views projected together (gsr_forward_project_batch), all with buffers for n_pairs pairs:
 * binning per view, the heaviest-first work lists of all views in one launch, then K6 per view. */
int gsr_forward_render_batch(int32_t n_views, const GsrView* views, const GsrGeom* geoms, uint64_t n_pairs,
                             GsrBinning* binnings, GsrImages* images, void* stream, GsrProfile* prof);

/* K7 reverse traversal of every pixel's blend list + K8 chain rule to the inputs. */
int gsr_backward(const GsrView*, const GsrGaussians*, const GsrGeom*, const GsrBinning*, const GsrImages*,
                 const GsrImageGrads*, GsrGrads*, void* stream, GsrProfile* prof);

/* Backward of n_views views of the SAME Gaussians (forwarded together or not): K7 per view, then ONE K8 pass that reads
 * every parameter row once, loops over the views and writes the summed parameter gradients once (supported for shs with
 * K in {1,4,9,16} + scales/rotations without camera gradients; other combinations run K8 view by view with the same
 * result). All arrays have n_views entries. Per view: outs[k].partials and outs[k].dL_dmeans2D (and dL_dview / dL_dproj /
 * dL_dcampos, stat_*); the parameter gradient pointers and `accumulate` are taken from outs[0] (give every
 * entry the same ones): the SUM over the views is written (accumulate = 0) or added (accumulate = 1) there. The
 * densification statistics are updated for the views whose entry sets the stat_* pointers (all such entries must name
 * the same three tensors) -- the reference's trainers count the LAST view of a step only (object_trainer.py:386-390) --
 * once per such view that saw the Gaussian. With per-view `scales` (see gsr_forward_project_batch)
 * every outs[k].dL_dscales is its own [P,3] buffer and receives view k's scale gradient (never accumulated). With a
 * scene every outs[k].scene names the SAME model gradient tensors (summed over the views) and its own dL_dscales_out. */
int gsr_backward_views(int32_t n_views, const GsrView* views, const GsrGaussians* gaussians /* [n_views] */,
                       const GsrGeom* geoms,
                       const GsrBinning* binnings, const GsrImages* images, const GsrImageGrads* image_grads,
                       GsrGrads* outs, void* stream, GsrProfile* prof);

/* ---- replacement of `simple_knn._C.distCUDA2` (gs_renderer.py:9, 590-593) --------
 * out[i] = mean of the squared distances from points[i] to its 3 nearest other points (exact; FLT_MAX stands in
 * for neighbours that do not exist when n < 4). points [n,3] fp32, out [n] fp32, both on the device; scratch:
 * gsr_knn_scratch_bytes(n) bytes, 256-byte aligned. Enqueued on `stream`, no host synchronisation. */
size_t gsr_knn_scratch_bytes(int32_t n);
int gsr_knn_mean_dist2(const float* points, int32_t n, float* out, void* scratch, size_t scratch_bytes, void* stream);

/* ---- the optimizer of the post-raster epilogue ----------------------------------
 * One launch of torch.optim.Adam's update (no amsgrad / weight decay) over all parameter groups of a model
 * (gs_renderer.py:615-653: Adam(l, lr=0.0, eps=1e-15), six groups with their own lr). All pointers are device fp32,
 * 16-byte aligned; `step` is the 1-based step count (bias correction); zero_grad != 0 clears the gradients in the same
 * pass. The densification statistics of that epilogue are fused into K8 (GsrGrads.stat_*). */
#define GSR_MAX_ADAM_GROUPS 32
typedef struct GsrAdamGroup {
  float* param;        /* [numel] updated in place        */
  float* grad;         /* [numel] (cleared if zero_grad)   */
  float* exp_avg;      /* [numel] first moment             */
  float* exp_avg_sq;   /* [numel] second moment            */
  int64_t numel;
  float lr;
  float reserved_;
} GsrAdamGroup;
int gsr_adam_step(const GsrAdamGroup* groups, int32_t n_groups, int32_t step, double beta1, double beta2, double eps,
                  int32_t zero_grad, void* stream);

/* ---- device side of the sparse gradient-exchange formats -------------------------------------
 * One view per GPU, the only exchange of a step is the sum of the ranks' parameter gradients (the reference sums the
 * C_batch_size views of a step in one process, training/object_trainer.py:302-382). K8 leaves the rows that can be non-zero
 * as a bitmap (GsrGrads.reached_mask: 16 % of the rows at 500 k Gaussians @1024^2); the row formats of the exchange move
 * (row index, row) messages. A ROW SET is a table whose logical row i is the concatenation of slices of up to
 * GSR_ROWSET_MAX_REGIONS strided regions -- element f of row i, in region k, lives at regions[k].ptr[i * stride + (f - first_k)]:
 * the planar gradient arena (means3D [P,3] | scales [P,3] | rotations [P,4] | opacities [P,1] | the active SH columns of shs
 * [P,K,3]: width 3 (D+1)^2, stride 3 K) as well as a plain row-major buffer (one region). All device pointers, fp32. */
#define GSR_ROWSET_MAX_REGIONS 8
typedef struct GsrRowRegion {
  float* ptr;
  int32_t width;    /* floats of a logical row that live in this region          */
  int32_t stride;   /* floats between consecutive rows of this region (>= width) */
} GsrRowRegion;
typedef struct GsrRowSet {
  int32_t rows;       /* logical rows */
  int32_t n_regions;  /* 1..GSR_ROWSET_MAX_REGIONS */
  GsrRowRegion regions[GSR_ROWSET_MAX_REGIONS];
} GsrRowSet;
size_t gsr_rows_scratch_bytes(int32_t rows);
/* The rows whose bit is set in mask (u64[(rows + 63) / 64], bit i % 64 of word i / 64; bits beyond `rows` ignored) as a message:
 * idx[n] = their indices in ascending order, rows_out[n][F] = the rows (F = sum of the region widths), *count = n (device u32;
 * the caller reads it to size its wire buffers). At most `cap` rows are written; n may exceed cap (then re-run with more room).
 * scratch: gsr_rows_scratch_bytes(rows) bytes, 256-byte aligned. Two launches, no host synchronisation. */
int gsr_rows_pack(const GsrRowSet* set, const uint64_t* mask, uint32_t* idx, float* rows_out, uint32_t cap, uint32_t* count,
                  void* scratch, size_t scratch_bytes, void* stream);
/* The reverse: row idx[j] - row_base of the set receives rows_in[j] (mode 1: stored, mode 0: ADDED -- one message per launch
 * and the indices of a message distinct, so messages applied in rank order give every rank the same sums); touched (NULL or
 * u64[(rows + 63) / 64]): the bits of the rows written are OR-ed in (the owner side of a sparse reduce-scatter). Entries whose
 * idx[j] - row_base falls outside [0, set->rows) are skipped. */
int gsr_rows_unpack(const GsrRowSet* set, const uint32_t* idx, const float* rows_in, uint32_t n, int64_t row_base, int32_t mode,
                    uint64_t* touched, void* stream);
/* The local sum of the `direct` exchange (multiview.GradExchange: one all-to-all, THIS, one all-gather): out[i] =
 * ((s_0[i] + s_1[i]) + s_2[i]) + ... over n_slices slices of slice_floats floats, slice w at slices + w * stride_floats -- rank
 * order, the association every rank applies to the slice it owns. out may be slice 0 (in place). One launch. */
int gsr_sum_slices(const float* slices, int32_t n_slices, uint64_t slice_floats, uint64_t stride_floats, float* out, void* stream);
/* Self-describing row messages: the host-read-free forms of the sparse exchanges (multiview.GradExchange `rows` / `sparse_rs`; they
 * stand in for the count reads gsr_rows_pack needs -- the sequential accumulation all of it implements: training/object_trainer.py:
 * 302-382). A message of `rows` rows of `row_floats` floats with a fixed capacity of `cap` rows is gsr_rowmsg_bytes() bytes (a multiple
 * of 256): [header u32[64]: count, cap, rows, F, largest count its sender received | bitmap u64[(rows+63)/64] | rows in front of each
 * 64-row word u32[...] | rows f32[cap][F]]. Messages are 256-byte aligned; message q of an array sits at msgs + q * msg_stride.
 *   gsr_rowmsg_pack          ONE launch: the rows whose bit is set in `mask` (as gsr_rows_pack) -> msg. A count above cap is recorded
 *                            in the header (the rows beyond cap are not written).
 *   gsr_rowmsg_pack_slices   ONE launch: the set cut into n_slices slices of slice_rows rows (a multiple of 64; n_slices * slice_rows
 *                            >= rows): message y = slice y with slice-local row numbers -- what a rank sends to owner y.
 *   gsr_rowmsg_reduce        ONE launch, messages -> message: n_msgs (<= 16) messages describing the SAME rows (one per rank, rank
 *                            order) -> the message of their union, every row = ((g_0 + g_1) + ...) over the messages that hold it. If
 *                            an input does not fit cap_in (or has another shape) the output's count is 0xFFFFFFFF. layout_rows (0 =
 *                            rows): the row count the messages' layout was sized for (slice_rows of a slice message; >= rows).
 *   gsr_rowmsg_apply         ONE launch over n_msgs messages of the whole set (rank order): every row ANY message holds receives the
 *                            rank-ordered sum over the messages that hold it, STORED (rows nobody holds are left as they are); touched
 *                            (NULL or u64[(rows+63)/64]) receives the union bitmap (every word of it, if the messages are applied).
 *   gsr_rowmsg_apply_slices  ONE launch: message y (an owner's reduced slice) stored into rows [y * slice_rows, ...) of the set;
 *                            touched as above (the bitmaps of the slices side by side: slice_rows is a multiple of 64).
 * Both apply forms write NOTHING if any header's count exceeds cap (or its cap / rows / F differ from the call's). *status (NULL, or
 * one u64, device or page-locked host memory), stored when the kernel STARTS: bits 1:0 = 1 applied / 2 nothing applied, bits 32:2 the
 * largest count among the messages, bits 63:33 the largest count their senders received (owners' messages; 0 otherwise). */
size_t gsr_rowmsg_bytes(int32_t rows, int32_t row_floats, uint32_t cap);
int gsr_rowmsg_pack(const GsrRowSet* set, const uint64_t* mask, void* msg, uint32_t cap, void* stream);
int gsr_rowmsg_pack_slices(const GsrRowSet* set, const uint64_t* mask, void* msgs, uint64_t msg_stride, int32_t n_slices,
                           int32_t slice_rows, uint32_t cap, void* stream);
int gsr_rowmsg_reduce(int32_t rows, int32_t layout_rows, int32_t row_floats, const void* msgs, uint64_t msg_stride, int32_t n_msgs,
                      uint32_t cap_in, void* msg_out, uint32_t cap_out, void* stream);
int gsr_rowmsg_apply(const GsrRowSet* set, const void* msgs, uint64_t msg_stride, int32_t n_msgs, uint32_t cap, uint64_t* status,
                     uint64_t* touched, void* stream);
int gsr_rowmsg_apply_slices(const GsrRowSet* set, const void* msgs, uint64_t msg_stride, int32_t n_slices, int32_t slice_rows,
                            uint32_t cap, uint64_t* status, uint64_t* touched, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSRAST_H */
