/* d4gs.h -- C ABI of libd4gs.so: MI355X-native (gfx950) 4D-Gaussian exposure rasterizer.
 *
 * Drop-in boundary for ONE path of ZcsrenlongZ/Deblur4DGS: everything `SceneModel.render`
 * (reference flow3d/scene_model.py:162-487) does on the device for one blurry frame --
 *   per-Gaussian activations            flow3d/params.py:39-43,70-84
 *   motion-basis deformation            flow3d/params.py:142-180, flow3d/transforms.py:41-53
 *   pose compose + camera delta         flow3d/scene_model.py:67-120,352-353
 *   gsplat.rendering.rasterization      flow3d/scene_model.py:360-373  (gsplat==1.1.1, packed=False)
 *   exposure blend                      flow3d/scene_model.py:386-397
 * forward and backward.  The reference reaches that code through Python (torch autograd + the gsplat
 * CUDA extension); there is no C interface in the reference to copy, so the entry points below are what a
 * ctypes / pybind binding of this path binds (see INTEGRATION.md for the stub).
 *
 * Conventions
 *  - plain C, no torch types; every pointer is a DEVICE pointer unless marked [host].
 *  - the library never allocates, frees or synchronises: outputs and scratch are caller-provided,
 *    every call only enqueues work on `stream` (a hipStream_t passed as void*).
 *  - return 0 on success, negative D4GS_E* on error; message via d4gs_last_error() (thread-local).
 *  - all floating point is fp32; matrices are row-major; quaternions are wxyz.
 *  - instance index i = s*N + g (sub-sample s, Gaussian g); the first G Gaussians are dynamic.
 */
#ifndef D4GS_H
#define D4GS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libd4gs.so is built with -fvisibility=hidden: the entry points below are its whole dynamic symbol table. */
#define D4GS_API __attribute__((visibility("default")))

#define D4GS_VERSION 305
#define D4GS_TILE 16
#define D4GS_GEOM_STRIDE 8 /* floats per instance record: x, y, opacity, depth, conic a, b, c, pad */

enum {
  D4GS_OK = 0,
  D4GS_EINVAL = -1,     /* bad argument / unsupported shape */
  D4GS_ELAUNCH = -2,    /* hipLaunch / runtime error (message holds hipGetErrorString) */
  D4GS_ECAPACITY = -3   /* caller-provided buffer too small */
};

enum { /* D4gsDims.flags */
  D4GS_RAW_PARAMS = 1,  /* quats/scales/opacities/motion_coefs are raw leaves: apply normalize/exp/sigmoid/softmax
                           (params.py:39-43).  When clear, scales/opacities are used as given (gsplat seam). */
  D4GS_RAW_COLORS = 2,  /* the first `n_sigmoid` colour channels are raw: apply sigmoid (params.py:40) */
  D4GS_LAZY_SORT = 8,   /* occluded / large-footprint scenes (most of every tile list lies behind the tile's last contributor):
                           every list is split into a NEAR part (the nearest depth buckets, about D4gsIsect.near_target keys)
                           and the rest; near parts are emitted, sorted and composited first, the far part of a list is
                           emitted and sorted - and the tile composited again over its whole list - only if the tile did not
                           saturate within the near part (d4gs_raster_fwd does that between its two passes).  Same lists where they matter, same image and gradients bit for bit;
                           needs D4gsProjOut.lazy_ws. */
  D4GS_EXACT_TILES = 16, /* (v305) on top of D4GS_EXACT_CULL: inside the tight rectangle, bin a splat only into the tiles its alpha >= 1/255
                           ellipse actually reaches (the corner tiles of a 2 x 2 ... 8 x 8 rectangle usually are not reached: 15 - 40 % shorter
                           lists at 720p).  The per-tile test runs in d4gs_project_fwd, the pairs of a wave's 64 instances spread over its
                           lanes, and leaves a 64-bit tile mask per instance (D4gsProjOut.tile_masks, required with the flag).  Same images
                           and gradients bit for bit; tiles_touched / the lists get shorter.  Pays from ~3 tiles per instance on.
                           The flag must be THE SAME in the dims passed to d4gs_project_fwd, d4gs_bin_sort, d4gs_raster_fwd and
                           d4gs_raster_bwd of one render (the counts and offsets are built from the masks; the binning walks them):
                           with the flag and no tile_masks those calls return D4GS_EINVAL. */
  D4GS_EXACT_CULL = 4   /* bin a splat only into tiles that hold a pixel with alpha >= 1/255 (tight ellipse
                           sigma <= ln(255*opacity), intersected with gsplat's 3-sigma tile rectangle).  Pixels in
                           the dropped tiles would fail gsplat's alpha test anyway, so images and gradients are
                           unchanged; only the intersection lists (tiles_touched / flatten ids) get shorter. */
};

enum { D4GS_DEPTH_NONE = 0, D4GS_DEPTH_ED = 1, D4GS_DEPTH_D = 2 }; /* render_mode RGB / RGB+ED / RGB+D */
/* D4gsRasterGrads.row_mode.  DENSE: every row of isect_grad is written (rows behind a tile's last contributor zero-filled);
 * SPARSE: only replayed rows are written and flagged in isect_live (large-footprint / occluded scenes).  Same gradients, bitwise. */
enum { D4GS_ROWS_AUTO = 0, D4GS_ROWS_DENSE = 1, D4GS_ROWS_SPARSE = 2 };

typedef struct D4gsDims {
  int32_t N;          /* Gaussians */
  int32_t G;          /* dynamic Gaussians (deformed by the motion bases); 0 = static scene */
  int32_t K;          /* motion bases */
  int32_t T;          /* frames per basis */
  int32_t S;          /* exposure sub-samples in this call */
  int32_t D;          /* colour channels supplied by the caller (without the depth channel) */
  int32_t width, height;
  int32_t depth_mode; /* D4GS_DEPTH_* ; channels rendered = D + (depth_mode != 0) */
  int32_t flags;      /* D4GS_RAW_* */
  int32_t n_sigmoid;  /* with D4GS_RAW_COLORS: channels [0,n_sigmoid) get sigmoid */
  float near_plane, far_plane, eps2d, radius_clip; /* gsplat defaults 0.01, 1e10, 0.3, 0 */
} D4gsDims;

/* leaf inputs of the deform+project stage */
typedef struct D4gsProjIn {
  const float *means;        /* [N,3] */
  const float *quats;        /* [N,4] wxyz */
  const float *scales;       /* [N,3] */
  const float *opacities;    /* [N]   */
  const float *colors;       /* [N,D] row-major, or NULL if every channel is generated (mask) */
  const float *motion_coefs; /* [G,K] or NULL */
  const float *rots;         /* [K,T,6] or NULL */
  const float *transls;      /* [K,T,3] or NULL */
  const float *times;        /* [S] exposure times (frame units); ignored when G == 0 */
  const float *RTs;          /* [S,3,4] camera deltas, or NULL = identity */
  const float *viewmat;      /* [4,4] world->camera */
  const float *Kmat;         /* [3,3] intrinsics */
} D4gsProjIn;

/* per-instance outputs + binning counters produced by d4gs_project_fwd */
typedef struct D4gsProjOut {
  float *means2d;          /* [S,N,2] pixel units */
  float *depths;           /* [S,N]   */
  float *conics;           /* [S,N,3] */
  int32_t *radii;          /* [S,N]  >0 <=> visible */
  float *opac_act;         /* [N]    activated opacity */
  float *ctab;             /* [N,DP] activated colour table, DP = 4*ceil(D/4); 16-byte aligned (rows are stored / read as 16-byte words:
                              d4gs_project_fwd / d4gs_project_bwd return D4GS_EINVAL otherwise) */
  float *geom;             /* [S*N,8] packed raster record */
  int32_t *tile_rects;     /* [S*N,2] packed tile rectangle: x0 | x1<<16 , y0 | y1<<16 (min incl., max excl.) */
  int32_t *tiles_touched;  /* [S*N] */
  int32_t *isect_offsets;  /* [S*N] exclusive scan of tiles_touched (emission index base of every instance); complete after
                              d4gs_bin_sort: for small tile grids the binning stage finishes the scan itself */
  int32_t *lazy_ws;        /* [D4gsSizes.lazy_ws] scratch of D4GS_LAZY_SORT (per (tile, depth bucket) counts, near counts, pivots,
                              slot cursors, tile flags, depth range); NULL without the flag.  (Until v302 this slot was the
                              unused `tile_ranks`.) */
  int32_t *tile_counts;    /* [2*S*tiles]: [0,T) splats per tile, [T,2T) per-tile slot cursors - zeroed by
                              d4gs_project_fwd and consumed by d4gs_bin_sort, which therefore runs once per projection
                              (a launch refused by the capacity check does not touch them); T = S*tiles */
  int32_t *tile_offsets;   /* [S*tiles+1] exclusive scan of tile_counts */
  int64_t *n_isect;        /* [4] {total intersections, longest tile list, sampled entries, sampled live entries}: [0], [1]
                              are read back by the host to size the next stage and to pick the sort size classes.  [2], [3]
                              (zeroed by d4gs_project_fwd, accumulated by d4gs_raster_fwd over a fixed sample of <= 128 tiles -
                              device-scope atomics queue up memory-side, one per tile would cost the composite 5 - 30 %): list
                              entries of the sampled tiles, and how many of them lie at or in front of their tile's last
                              contributor, i.e. the rows the backward will replay.  [3] / [2] estimates the live fraction a
                              caller can choose D4gsRasterGrads.row_mode from (deblur4dgs_amd/engine.py does, one render late) */
  int32_t *scan_ws;        /* [d4gs_scan_ws_elems(S*N)] scratch */
  float *blend_bases;      /* [S,K,16] or NULL (v305, appended): the time-blended motion bases of the S sub-samples (row k: transl 3,
                              6-D rotation 6, 7 pad floats), written by d4gs_project_fwd when G > 0 and read by d4gs_project_bwd with
                              scalar loads.  NULL: d4gs_project_bwd builds the table itself (one small launch more). */
  uint64_t *tile_masks;    /* [S*N] (v305, appended; needed with D4GS_EXACT_TILES only): bit (ty - y0) * 8 + (tx - x0) = tile (tx, ty) of the
                              instance's packed rectangle is binned; 0 = no mask, the whole rectangle is (rectangles wider or taller
                              than 8 tiles, single tiles, flag off) */
} D4gsProjOut;

typedef struct D4gsIsect {
  int64_t n_isect;         /* [host] CAPACITY of the four lists below (elements), >= D4gsProjOut.n_isect[0].  The host
                            * may size them from a guess: every kernel of d4gs_bin_sort / d4gs_raster_fwd compares the
                            * device-side count with this value and returns at once (outputs untouched) when it does
                            * not fit, so the caller reads D4gsProjOut.n_isect AFTER launching and re-launches with
                            * exact sizes in that case - no host sync sits between the projection and the rasterizer.
                            * d4gs_raster_bwd needs the exact count. */
  int64_t max_tile_count;  /* [host] upper bound of D4gsProjOut.n_isect[1], checked on the device like n_isect
                            * (<= 0: unknown, launch every sort class) */
  int64_t near_target;     /* [host] D4GS_LAZY_SORT: keys per tile list to aim the near part at (<= 0: 1024).  About twice the
                            * entries a tile is expected to consume before it saturates. */
  uint64_t *keys;          /* [n_isect] scratch: (depth bits << 32 | emission index) per tile slot */
  int32_t *gid_of_emit;    /* [n_isect] Gaussian id of each emission index */
  int32_t *sorted_gid;     /* [n_isect] per-tile depth-sorted Gaussian ids (flatten_ids) */
  int32_t *sorted_emit;    /* [n_isect] emission index of each sorted slot */
} D4gsIsect;

typedef struct D4gsRaster {
  const float *background; /* [D] or NULL (depth channel's background is 0) */
  float *render_colors;    /* [S,H,W,D+depth] */
  float *render_alphas;    /* [S,H,W] */
  int32_t *last_ids;       /* [S,H,W] index (into sorted_gid) of the last composited splat */
  float *final_T;          /* [S,H,W] transmittance left behind the last splat.  The backward starts from this instead
                              of 1 - render_alphas (as gsplat does): for nearly opaque pixels (T ~ 1e-4) the fp32
                              subtraction loses ~3 digits, which shows up as a ~5e-4 relative bias of the gradients. */
  float *seg_state;        /* [D4gsSizes.seg_state] scratch or NULL.  Few-tile launches (S * tiles <= 1280: one or two exposure
                              sub-samples of a 288x512 frame, i.e. a rank of BASELINE config 4) cannot fill 256 CUs with one
                              workgroup per tile; given this buffer, d4gs_raster_fwd also stores every pixel's transmittance and
                              accumulated channels at up to 7 depth-segment boundaries of its tile list (+ the final ones) and
                              d4gs_raster_bwd replays the segments in parallel workgroups, each starting from the stored state.
                              Same image bit for bit; gradients equal to the unsegmented replay up to fp32 rounding of the
                              hand-off (deterministic).  NULL, or a configuration whose seg_state size is 0: one workgroup per
                              tile as before.  The SAME value must be passed to the forward and its backward. */
} D4gsRaster;

/* gradients w.r.t. the raster stage's per-instance inputs */
typedef struct D4gsRasterGrads {
  const float *v_render_colors; /* [S,H,W,D+depth] */
  const float *v_render_alphas; /* [S,H,W] or NULL */
  float *isect_grad;            /* [n_isect, 6+D+depth] scratch, 16-byte aligned: the row of every intersection that was
                                   replayed (at or before its tile's last contributor); the other rows are NOT written */
  uint8_t *isect_live;          /* [n_isect rounded up to a multiple of 4] scratch, 4-byte aligned: 1 = the row above was
                                   written by this call, 0 = skip it */
  float *v_means2d;             /* [S,N,2]  (= means2d.grad contract, trainer.py:975) */
  float *v_conics;              /* [S,N,3] */
  float *v_depths;              /* [S,N]   (zeros when depth_mode == 0) */
  float *v_opac_act;            /* [N]     summed over S */
  float *v_ctab;                /* [N,DP]  summed over S; 16-byte aligned (d4gs_project_bwd reads it as 16-byte words) */
  /* SURVEY 8f-1, fused: when stats_grad_norm_acc != NULL the gather epilogue also does the accumulation loop of
   * Trainer._prepare_control_step (flow3d/trainer.py:967-989) for this render - per visible instance (radii > 0), in
   * sub-sample order: grad_norm_acc[g] += |v_means2d * (W/2, H/2) * stats_batch_size * S|, vis_count[g] += 1, and, only
   * if stats_update_max_radii (the reference's index_put result is discarded, so upstream never updates it),
   * max_radii[g] = max(max_radii[g], radius / max(W, H)).  Same arithmetic and order as d4gs_control_stats. */
  float *stats_grad_norm_acc;   /* [N] or NULL */
  int64_t *stats_vis_count;     /* [N] */
  float *stats_max_radii;       /* [N] */
  int32_t stats_batch_size;
  int32_t stats_update_max_radii;
  int32_t row_mode;             /* D4GS_ROWS_*: AUTO picks dense / sparse rows from the list capacity per instance */
} D4gsRasterGrads;

/* leaf gradients produced by d4gs_project_bwd (all overwritten, not accumulated) */
typedef struct D4gsLeafGrads {
  float *v_means, *v_quats, *v_scales, *v_opacities, *v_colors; /* [N,3] [N,4] [N,3] [N] [N,D] */
  float *v_motion_coefs; /* [G,K] or NULL */
  float *v_rots;         /* [K,T,6] or NULL */
  float *v_transls;      /* [K,T,3] or NULL */
  float *v_times;        /* [S] or NULL */
  float *v_RTs;          /* [S,3,4] or NULL */
  float *v_viewmat;      /* [4,4] or NULL (test-time pose optimisation, validator.py:437-452) */
  float *partials;       /* [d4gs_bwd_partials_elems(dims)] scratch for the deterministic 2-level reduction */
} D4gsLeafGrads;

D4GS_API int d4gs_version(void);
/* optional per-kernel HIP-event timing (bench.py's roofline object): enable, run, then collect
 * "kernel_name launches total_ms" lines.  on = 1 times every kernel (two stream events per launch: ~0.12 ms per
 * cfg2 frame), on = 2 only the rasterization kernels (k_raster*), on = 3 only the composite backward (k_raster_bwd*: what bench.py's
 * roofline object needs from its timed region), 0 = off (default; no
 * events are created). */
D4GS_API void d4gs_profile_enable(int on);
D4GS_API int d4gs_profile_collect(char *buf /* [host] */, size_t cap);
D4GS_API const char *d4gs_last_error(void);
/* The two MEASURED device ceilings bench.py quotes its roofline fractions against: a device-to-device stream copy over
 * `scratch` (first half -> second half, best of 8) and an FMA issue loop at 8 waves per SIMD.  out[0] = copy GB/s (read + write),
 * out[1] = fp32 TFLOP/s with v_pk_fma_f32, out[2] = fp32 TFLOP/s with v_fma_f32 (what the composite kernels issue), out[3] = bytes
 * copied per launch.  Diagnostic: creates its own HIP events and WAITS for them - not for timed regions or stream captures. */
D4GS_API int d4gs_measure_peaks(void *scratch /* device, >= 64 MiB */, size_t scratch_bytes, double *out /* [host] [4] */, void *stream);
/* D4gsProjOut.n_isect -> PINNED (device-addressable: hipHostMalloc / torch pin_memory) host memory, by a one-wave kernel on `stream`
 * that stores the four counts there (a kernel node when the stream is being captured - how a step replayed from a HIP graph keeps
 * reporting its list sizes: engine.GraphWatch; a device-to-host copy node would hold up the kernels behind it).  Read the buffer after
 * an event recorded behind this call has completed.  (Pageable memory is accepted too: it gets an ordinary device-to-host copy.) */
D4GS_API int d4gs_copy_counts(const int64_t *n_isect /* device [4] */, int64_t *host_pinned /* [4] */, void *stream);
D4GS_API size_t d4gs_scan_ws_elems(int64_t n_instances);
D4GS_API size_t d4gs_bwd_partials_elems(const D4gsDims *dims);

/* Element counts of every caller-allocated buffer for one configuration (the "workspace query" of SURVEY 8b): the
 * per-instance / per-tile buffers of D4gsProjOut, the image buffers of D4gsRaster, and - per intersection, to be
 * multiplied by the list capacity the caller chooses (see D4gsIsect.n_isect) - the rows of D4gsIsect and of
 * D4gsRasterGrads.isect_grad.  Element types are the ones of the struct fields. */
typedef struct {
  int64_t means2d, depths, conics, radii, opac_act, ctab, geom, tile_rects, tiles_touched, isect_offsets;
  int64_t tile_counts, tile_offsets, n_isect, scan_ws;          /* D4gsProjOut */
  int64_t render_colors, render_alphas, last_ids, final_T;       /* D4gsRaster (seg_state: last field of this struct) */
  int64_t isect_grad_row;                                        /* floats per intersection in isect_grad (isect_live: 1 byte) */
  int64_t bwd_partials;                                          /* D4gsLeafGrads.partials */
  int64_t seg_state;                                             /* D4gsRaster.seg_state; 0 = this configuration does not use depth segments */
  int64_t lazy_ws;                                               /* D4gsProjOut.lazy_ws (int32 elements; needed with D4GS_LAZY_SORT only) */
  int32_t tiles_x, tiles_y, channels;                            /* tile grid; D + depth channel */
  int64_t blend_bases;                                           /* D4gsProjOut.blend_bases (v305, appended; 0 when G == 0) */
  int64_t tile_masks;                                            /* D4gsProjOut.tile_masks (uint64 elements; needed with D4GS_EXACT_TILES) */
} D4gsSizes;
D4GS_API int d4gs_query_sizes(const D4gsDims *dims, D4gsSizes *sizes);

/* a1-a6 + projection + tile counting + scans.  Replaces params.py:39-43,142-180, transforms.py:41-53,
 * scene_model.py:67-120,352-353 and gsplat fully_fused_projection_fwd + isect_tiles pass 1 for all S. */
D4GS_API int d4gs_project_fwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsProjOut *out, void *stream);

/* isect_tiles pass 2 + per-tile depth sort (replaces gsplat isect_tiles / radix sort / isect_offset_encode). */
D4GS_API int d4gs_bin_sort(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, void *stream);

/* rasterize_to_pixels_fwd for all S sub-samples (+ expected-depth normalisation when depth_mode == ED). */
D4GS_API int d4gs_raster_fwd(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, const D4gsRaster *r,
                    void *stream);

/* rasterize_to_pixels_bwd + per-instance gather of the per-intersection gradients (deterministic, no float atomics). */
D4GS_API int d4gs_raster_bwd(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, const D4gsRaster *r,
                    const D4gsRasterGrads *g, void *stream);

/* fully_fused_projection_bwd + deformation / activation adjoints for all S, reduced to leaf gradients. */
D4GS_API int d4gs_project_bwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsProjOut *proj,
                     const float *v_means2d, const float *v_conics, const float *v_depths, const float *v_opac_act,
                     const float *v_ctab, const D4gsLeafGrads *grads, void *stream);

/* a11 track channels (scene_model.py:258-289): camera-space position of every Gaussian at S target times,
 * points[s,g] = RTs[s] * deform(g, times[s]) (then viewmat, normally identity).  Uses dims N/G/K/T/S and
 * in->means, motion_coefs, rots, transls, times (= target_ts), RTs (= target_w2cs[:, :3, :], may be NULL), viewmat,
 * Kmat; every other input may be NULL.  The backward fills v_means, v_motion_coefs, v_rots, v_transls, v_times,
 * v_RTs, v_viewmat of D4gsLeafGrads (the other leaf pointers are not touched). */
D4GS_API int d4gs_points_fwd(const D4gsDims *dims, const D4gsProjIn *in, float *points /* [S,N,3] */, void *stream);
D4GS_API int d4gs_points_bwd(const D4gsDims *dims, const D4gsProjIn *in, const float *v_points /* [S,N,3] */,
                    const D4gsLeafGrads *grads, void *stream);

/* a4/a5 pose API of the S2 seam (flow3d/scene_model.py:58-120; called by the reference's Trainer at
 * flow3d/trainer.py:303,478,485,701,818 and its Renderer at flow3d/renderer.py:37): for the S = B times in in->times
 *   transforms[g,b] = [cont_6d_to_rmat(r6) | transl] of the time-blended, coefficient-weighted bases   (G dynamic rows)
 *   means[g,b]      = R means_g + transl                          (static rows g >= G: means_g)
 *   quats[g,b]      = normalize( wxyz( rotmat_to_unitquat(R)_xyzw (x) xyzw(normalize(quats_g)) ) )   (static: normalize)
 * Uses dims N/G/K/T/S and in->means, quats (only when `quats` is requested), motion_coefs, rots, transls, times; in->RTs
 * [S,3,4] and in->viewmat (both optional, NULL = identity) are applied to the MEANS only - that is d4gs_points_fwd, the
 * a11 track channels.  Any of the three outputs may be NULL.  g_major selects the layout: 0 = time-major [S,N,...]
 * ([S,G,3,4] for transforms), 1 = the reference's Gaussian-major (G,B,...) tensors.
 * d4gs_poses_bwd takes the three gradients in the same struct (NULL = zero) and fills v_means, v_quats (if quats'
 * gradient was given and v_quats != NULL), v_motion_coefs, v_rots, v_transls, v_times, v_RTs of D4gsLeafGrads. */
typedef struct D4gsPoses {
  float *means;       /* [S,N,3] | [N,S,3] */
  float *quats;       /* [S,N,4] | [N,S,4]  wxyz, unit */
  float *transforms;  /* [S,G,3,4] | [G,S,3,4] */
  int32_t g_major;
} D4gsPoses;
D4GS_API int d4gs_poses_fwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsPoses *out, void *stream);
D4GS_API int d4gs_poses_bwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsPoses *v_out, const D4gsLeafGrads *grads,
                   void *stream);

/* Densification statistics, Trainer._prepare_control_step (flow3d/trainer.py:953-990) for one render of
 * S sub-samples: for every visible instance (radii > 0)
 *   grad_norm_acc[g] += || xys_grad[s,g] * (W/2, H/2) * batch_size * S ||,  vis_count[g] += 1,
 *   max_radii[g] = max(max_radii[g], radii[s,g] / max(W, H))  -- ONLY when update_max_radii != 0: the reference
 *   computes this maximum and then drops it (out-of-place `index_put`, trainer.py:987-989), so its `max_radii`
 *   stays at its initial value; 0 reproduces that.   Running stats are updated in place. */
D4GS_API int d4gs_control_stats(int32_t S, int32_t N, const float *xys_grad /* [S,N,2] */, const int32_t *radii /* [S,N] */,
                       int32_t width, int32_t height, int32_t batch_size, float *grad_norm_acc /* [N] */,
                       int64_t *vis_count /* [N] */, float *max_radii /* [N] */, int32_t update_max_radii,
                       void *stream);

/* SURVEY 8f-1, control steps: the row surgery of GaussianParams.densify_params / cull_params (flow3d/params.py:86-118)
 * and of dup_in_optim / remove_from_optim (flow3d/trainer.py:1199-1236) as stream compaction.
 * d4gs_control_plan: flags [N] (uint8) -> src_map (capacity 3N for a densify plan, N for a cull plan) in the reference's
 * row order - rows with split == 0, then rows with dup != 0, then the rows with split != 0 TWICE (x[split].repeat(2));
 * with dup == NULL the first flag array is a cull mask and the plan keeps the rows whose flag is 0.
 * counts [4] (device) = {n_keep, n_dup, n_split, n_out}.  d4gs_gather_rows: out[i,:] = in[src_map[i],:] for one tensor
 * of `row_floats` 32-bit words per row; rows >= zero_from are zero-filled (new Adam moments: zero_from = n_keep), rows
 * >= add_from get `add` added (split halves of `scales`: add_from = n_keep + n_dup, add = -log 1.6); pass INT64_MAX
 * to disable either. */
D4GS_API int d4gs_control_plan(int32_t N, const uint8_t *split_or_cull, const uint8_t *dup, int32_t *src_map, int32_t *counts,
                      void *stream);
D4GS_API int d4gs_gather_rows(const int32_t *src_map, int64_t n_out, int32_t row_floats, const float *in, float *out,
                     int64_t zero_from, int64_t add_from, float add, void *stream);

/* a12 camera path: the generator of `RTs [S,3,4]` / `times [S]` that feed d4gs_project_fwd.
 * Replaces the eager chain of MoveModel.forward_start_end_mid (flow3d/models/move_model.py:138-166):
 *   pypose se3.Exp(delta0/1) -> linear_interpolation (flow3d/models/utils/spline_utils.py:371-408) -> SE3.Log ->
 *   se3_to_SE3 (spline_utils.py:197-215, with the [tau,phi]-read-as-[w,u] convention of move_model.py:146-147), and
 *   the exposure-time lerp of move_model.py:118-135,151-158 (half-width clamp(relu(time_params[index]),0.1,0.9) on
 *   interior frames 0 < index < n_time_params-1, else 0; pass time_params = NULL for stage "first").
 * delta0/delta1 [6] are the two MLP head outputs.  jac [S,12,12] = d RTs[s,i] / d (delta0|delta1)[j] (NULL to skip),
 * dtimes [S] = d times / d time_params[index], deltaT [2] = {|half-width|, its derivative}.  The backward is the
 * mat-vec v_delta = v_RTs . jac plus the time_params row (v_* inputs may be NULL = zero). */
D4GS_API int d4gs_camera_path_fwd(const float *delta0, const float *delta1, int32_t S, const float *time_params,
                         int32_t n_time_params, int32_t index, float t, float *RTs /* [S,3,4] */,
                         float *jac /* [S,12,12] */, float *times /* [S] */, float *dtimes /* [S] */,
                         float *deltaT /* [2] */, void *stream);
D4GS_API int d4gs_camera_path_bwd(const float *jac, const float *dtimes, const float *deltaT, const float *v_RTs,
                         const float *v_times, const float *v_deltaT, int32_t S, int32_t index,
                         int32_t n_time_params, float *v_delta0 /* [6] */, float *v_delta1 /* [6] */,
                         float *v_time_params /* [n_time_params] */, void *stream);
/* MoveModel.preprocessPose + positional embedding (move_model.py:12-63,104-110; spline_utils.py:177-195):
 * R [3,3] with row stride r_stride (4 for the top-left block of a [4,4] w2c), T [3] with element stride t_stride
 * -> enc [66] = [x, sin(x f), cos(x f)]_{f=1,2,4,8,16}, x = SE3_to_se3([R|T]). */
D4GS_API int d4gs_pose_encode(const float *R, int32_t r_stride, const float *T, int32_t t_stride, float *enc /* [66] */,
                     void *stream);
/* Its input gradient (test-time pose refinement differentiates the render w.r.t. w2c, which also feeds the
 * MoveModel: flow3d/validator.py:442-448 -> flow3d/scene_model.py:249-256): v_enc [66] -> v_R [3,3] dense, v_T [3]. */
D4GS_API int d4gs_pose_encode_bwd(const float *R, int32_t r_stride, const float *T, int32_t t_stride, const float *v_enc,
                         float *v_R /* [9] */, float *v_T /* [3] */, void *stream);

/* The whole MoveModel (move_model.py:66-166) for one pose, one call each way: d4gs_pose_encode -> the 9-layer MLP
 * (66 -> 64 x4 LeakyReLU(0.01) -> 64, heads 64 -> 64 -> 6; ~30 GEMV launches per render in eager PyTorch) ->
 * d4gs_camera_path_fwd.  Layer order in w[] / b[]: RT_main.0, .2, .4, .6, .8, RT_head0.0, .2, RT_head1.0, .2;
 * weights are [out,in] row-major as nn.Linear stores them. */
typedef struct {
  const float *w[9];
  const float *b[9];
  const float *time_params; /* [n_time_params] or NULL */
  int32_t n_time_params;
} D4gsMoveModelParams;
typedef struct {
  float *enc;    /* [66]  scratch */
  float *acts;   /* [514] saved layer inputs (backward) */
  float *delta;  /* [12]  the two head outputs */
  float *RTs;    /* [S,3,4] */
  float *jac;    /* [S,12,12] */
  float *times;  /* [S] */
  float *dtimes; /* [S] */
  float *deltaT; /* [2] */
} D4gsMoveModelOut;
typedef struct {
  float *v_w[9];
  float *v_b[9];
  float *v_time_params; /* [n_time_params] */
  float *v_delta;       /* [12] scratch */
  float *v_enc;         /* [66] gradient of the pose encoding (input of d4gs_pose_encode_bwd), or NULL to skip */
} D4gsMoveModelGrads;
D4GS_API int d4gs_move_model_fwd(const float *R, int32_t r_stride, const float *T, int32_t t_stride, const D4gsMoveModelParams *p,
                        int32_t S, int32_t index, float t, int32_t stage_first, const D4gsMoveModelOut *out, void *stream);
D4GS_API int d4gs_move_model_bwd(const D4gsMoveModelParams *p, const D4gsMoveModelOut *out, const float *v_RTs,
                        const float *v_times, const float *v_deltaT, int32_t S, int32_t index,
                        const D4gsMoveModelGrads *grads, void *stream);

/* the photometric loss term, fused (flow3d/trainer.py:388-392,575-586):
 *   loss = w_l1 * mean|pred*m - gt*m| + w_ssim * (1 - SSIM(pred*m, gt*m)),
 * SSIM = pytorch_msssim.SSIM(data_range=1, size_average=True, channel=3) [11-tap sigma-1.5 window, no padding].
 * pred, gt [B,H,W,3] channel-last; mask [B,H,W] or NULL.  Forward: loss[3] = {loss, l1, ssim} (device), plus what the
 * backward needs: maps [B,H-10,W-10,3,3], partials [d4gs_photometric_blocks(B,H,W), 2] scratch.  Backward:
 * v_pred [B,H,W,3] = dL/dpred * v_loss[0] (v_loss is a device scalar). */
D4GS_API int64_t d4gs_photometric_blocks(int32_t B, int32_t H, int32_t W);
D4GS_API int d4gs_photometric_fwd(const float *pred, const float *gt, const float *mask, int32_t B, int32_t H, int32_t W, int32_t C,
                         float w_l1, float w_ssim, float *maps, float *partials, float *loss, void *stream);
D4GS_API int d4gs_photometric_bwd(const float *pred, const float *gt, const float *mask, const float *maps, const float *v_loss,
                         int32_t B, int32_t H, int32_t W, int32_t C, float w_l1, float w_ssim, float *v_pred,
                         void *stream);

/* a9 exposure blend (scene_model.py:386-397): out = mean_S; policy[c] 1 -> max over {raw_0..raw_{S-2}, mean},
 * 2 -> min over the same set (the reference's in-place quirk); acc = mean_S alphas. */
D4GS_API int d4gs_blend_fwd(int32_t S, int64_t n_pixels, int32_t C, const int32_t *policy /* [host] [C] */,
                   const float *renders /* [S,P,C] */, const float *alphas /* [S,P] */, float *out /* [P,C] */,
                   float *acc /* [P] */, void *stream);
D4GS_API int d4gs_blend_bwd(int32_t S, int64_t n_pixels, int32_t C, const int32_t *policy /* [host] */, const float *renders,
                   const float *out, const float *v_out, const float *v_acc, float *v_renders /* [S,P,C] */,
                   float *v_alphas /* [S,P] */, void *stream);

/* One call each way: d4gs_forward = d4gs_project_fwd + d4gs_bin_sort + d4gs_raster_fwd (+ d4gs_blend_fwd when
 * io->blended != NULL), d4gs_backward = (d4gs_blend_bwd +) d4gs_raster_bwd + d4gs_project_bwd.  Same kernels, same order:
 * bit-identical to the staged calls.  Every buffer that only lives between / inside the two calls sits in ONE caller-provided
 * workspace (256-byte aligned, d4gs_frame_workspace_bytes(dims, isect_capacity) bytes; the forward's part of it must reach the
 * backward untouched); what the caller reads is in D4gsFrameIO.  `isect_capacity` / `max_tile_hint` are D4gsIsect.n_isect /
 * .max_tile_count: a guess the kernels check on the device - read io->n_isect afterwards and call again if it did not fit.
 * The backward takes any of v_blended, v_acc, v_renders, v_alphas when io->blended was given (losses on the blurry frame and
 * on the per-sub-sample images, flow3d/trainer.py:575-618), else v_renders [+ v_alphas]; a caller that needs `means2d` as an
 * autograd intermediate or renders more than 16 colour channels uses the staged entry points. */
typedef struct D4gsFrameIO {
  float *blended;          /* [H,W,D+depth] or NULL: no blend */
  float *acc;              /* [H,W] (with blended) */
  float *renders;          /* [S,H,W,D+depth] */
  float *alphas;           /* [S,H,W] */
  float *means2d;          /* [S,N,2] */
  int32_t *radii;          /* [S,N] */
  int64_t *n_isect;        /* [4] device: {intersections, longest tile list, sampled entries, sampled live entries} (D4gsProjOut.n_isect) */
  const float *background; /* [D] or NULL */
  const int32_t *policy;   /* [host] [D+depth] blend policy per channel (0 mean, 1 max, 2 min) or NULL = all mean */
  int64_t near_target;     /* [host] D4GS_LAZY_SORT: D4gsIsect.near_target of the frame (<= 0: 1024) */
  int64_t *counts_pinned;  /* [PINNED host memory, device-addressable] [4] or NULL (v305, appended): d4gs_forward also leaves the four
                              counts of n_isect there - stored by its LAST kernel (the blend, when io->blended is given: no extra
                              launch; otherwise one one-wave kernel), i.e. what d4gs_copy_counts would do in a launch of its own.
                              Read it after an event recorded behind d4gs_forward has completed. */
} D4gsFrameIO;
typedef struct D4gsFrameGrads {
  const float *v_blended, *v_acc;     /* [H,W,D+depth], [H,W] or NULL */
  const float *v_renders, *v_alphas;  /* [S,H,W,D+depth], [S,H,W] or NULL */
  float *v_means2d;                   /* [S,N,2] out: the means2d.grad contract (trainer.py:975) */
  float *stats_grad_norm_acc;         /* fused densification statistics, as in D4gsRasterGrads; NULL = off */
  int64_t *stats_vis_count;
  float *stats_max_radii;
  int32_t stats_batch_size, stats_update_max_radii, row_mode;
} D4gsFrameGrads;
D4GS_API size_t d4gs_frame_workspace_bytes(const D4gsDims *dims, int64_t isect_capacity);
/* The prefix of that workspace d4gs_forward alone touches (everything but the backward's scratch: image-gradient stack, per-
 * intersection gradient rows, per-instance gradients, block partials - roughly half): a forward that will never be followed by
 * d4gs_backward (validation, viewer) may pass a workspace of this size. */
D4GS_API size_t d4gs_frame_workspace_bytes_fwd(const D4gsDims *dims, int64_t isect_capacity);
D4GS_API int d4gs_forward(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io, void *workspace, size_t ws_bytes,
                 int64_t isect_capacity, int64_t max_tile_hint, void *stream);
D4GS_API int d4gs_backward(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io, const D4gsFrameGrads *g,
                  const D4gsLeafGrads *leaf /* .partials is ignored: it lives in the workspace */, void *workspace,
                  size_t ws_bytes, int64_t isect_capacity, int64_t max_tile_hint, void *stream);

/* CPU twins of the two calls above (BASELINE config 1, the CPU-runnable plumbing case): the same arithmetic rules
 * and the same structs with every pointer a HOST pointer; no stream, no workspace (scratch is allocated and freed inside; the
 * backward re-runs the forward), io->n_isect [4] host.  Scalar fp32, one thread - a separate entry point for hosts without a
 * GPU, never a fallback: d4gs_forward and the Python seams still refuse CPU tensors.  csrc/cpu_twin.hip. */
D4GS_API int d4gs_forward_cpu(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io);
D4GS_API int d4gs_backward_cpu(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io, const D4gsFrameGrads *g,
                      const D4gsLeafGrads *leaf);

/* Exposure sharding (one process per GPU): the same blend when this process holds only the sub-samples
 * s_first + j * s_stride, j < S_local, of the S_total.  Collectives stay with the caller (RCCL through torch.distributed):
 *   forward : partial_fwd -> all-reduce SUM of part [P,C+1] (colours + alpha) and MAX of cand [P,npol] (the max / min
 *             policy channels in ascending channel order, min packed as -x; -inf where this rank has no candidate) -> finish_fwd;
 *   backward: winner (lowest own s <= S_total-2 whose raw value equals the blended one, else S_total) -> all-reduce MIN ->
 *             bwd.  The mean channels need no collective backward (the loss is evaluated on every rank).
 * Equals d4gs_blend_fwd/bwd on the full stack up to the order of the S-term sums. */
typedef struct D4gsShardBlend {
  int32_t S_total, S_local, s_first, s_stride, C;
  int64_t n_pixels;
  const int32_t *policy; /* [host] [C] */
} D4gsShardBlend;
D4GS_API int d4gs_blend_shard_partial_fwd(const D4gsShardBlend *b, const float *renders /* [S_local,P,C] */,
                                 const float *alphas /* [S_local,P] */, float *part /* [P,C+1] */, float *cand /* [P,npol] */,
                                 void *stream);
D4GS_API int d4gs_blend_shard_finish_fwd(const D4gsShardBlend *b, const float *part, const float *cand, float *out /* [P,C] */,
                                float *acc /* [P] */, void *stream);
D4GS_API int d4gs_blend_shard_winner(const D4gsShardBlend *b, const float *renders, const float *out, int32_t *win /* [P,npol] */,
                            void *stream);
D4GS_API int d4gs_blend_shard_bwd(const D4gsShardBlend *b, const float *v_out, const float *v_acc /* or NULL */, const int32_t *win,
                         float *v_renders /* [S_local,P,C] */, float *v_alphas /* [S_local,P] */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* D4GS_H */
