// capi.hip -- the extern "C" surface of libd4gs.so (declared in include/d4gs.h) and error plumbing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdint.h>

#include "common.h"

int d4gs_project_fwd_impl(const D4gsDims *, const D4gsProjIn *, const D4gsProjOut *, hipStream_t);
int d4gs_bin_sort_impl(const D4gsDims *, const D4gsProjOut *, const D4gsIsect *, hipStream_t);
int d4gs_raster_fwd_impl(const D4gsDims *, const D4gsProjOut *, const D4gsIsect *, const D4gsRaster *, hipStream_t);
int d4gs_raster_bwd_impl(const D4gsDims *, const D4gsProjOut *, const D4gsIsect *, const D4gsRaster *,
                         const D4gsRasterGrads *, const BlendAdj *, hipStream_t);
int d4gs_project_bwd_impl(const D4gsDims *, const D4gsProjIn *, const D4gsProjOut *, const float *, const float *,
                          const float *, const float *, const float *, const D4gsLeafGrads *, hipStream_t);
int d4gs_poses_fwd_impl(const D4gsDims *, const D4gsProjIn *, const D4gsPoses *, hipStream_t);
int d4gs_poses_bwd_impl(const D4gsDims *, const D4gsProjIn *, const D4gsPoses *, const D4gsLeafGrads *, hipStream_t);
int d4gs_control_stats_impl(int32_t, int32_t, const float *, const int32_t *, int32_t, int32_t, int32_t, float *,
                            int64_t *, float *, int32_t, hipStream_t);
int d4gs_control_plan_impl(int32_t, const uint8_t *, const uint8_t *, int32_t *, int32_t *, hipStream_t);
int d4gs_gather_rows_impl(const int32_t *, int64_t, int32_t, const float *, float *, int64_t, int64_t, float, hipStream_t);
int d4gs_blend_fwd_impl(int32_t, int64_t, int32_t, const int32_t *, const float *, const float *, float *, float *, int8_t *,
                        hipStream_t, const int64_t *n_isect = nullptr, int64_t *counts_pinned = nullptr);
int d4gs_blend_bwd_impl(int32_t, int64_t, int32_t, const int32_t *, const float *, const float *, const float *,
                        const float *, float *, float *, hipStream_t);

int d4gs_blend_shard_impl(int, const D4gsShardBlend *, const void *, const void *, const void *, void *, void *, hipStream_t);

int d4gs_pose_encode_impl(const float *, int32_t, const float *, int32_t, float *, hipStream_t);
int d4gs_camera_path_fwd_impl(const float *, const float *, int32_t, const float *, int32_t, float, int32_t, float *,
                              float *, float *, float *, float *, hipStream_t);
int d4gs_camera_path_bwd_impl(const float *, const float *, const float *, const float *, const float *,
                              const float *, int32_t, int32_t, int32_t, float *, float *, float *, hipStream_t);

int d4gs_move_model_fwd_impl(const float *, int32_t, const float *, int32_t, const float *const *, const float *const *,
                             int32_t, const float *, int32_t, float, int32_t, float *, float *, float *, float *, float *,
                             float *, float *, float *, hipStream_t);
int d4gs_move_model_bwd_impl(const float *, const float *, const float *, const float *, const float *const *,
                             const float *const *, const float *, const float *, const float *, int32_t, int32_t, int32_t,
                             float *, float *const *, float *const *, float *, float *, hipStream_t);
int d4gs_pose_encode_bwd_impl(const float *, int32_t, const float *, int32_t, const float *, float *, float *, hipStream_t);

int d4gs_photometric_fwd_impl(const float *, const float *, const float *, int32_t, int32_t, int32_t, float, float, float *,
                              float *, float *, hipStream_t);
int d4gs_photometric_bwd_impl(const float *, const float *, const float *, const float *, const float *, int32_t, int32_t,
                              int32_t, float, float, float *, hipStream_t);

static thread_local char g_err[512] = "";

// ---- per-kernel event profiler -------------------------------------------------------------------------
#include <mutex>
#include <string>
#include <vector>
namespace {
struct ProfRec {
  const char *name;
  hipEvent_t a, b;
};
int g_prof_on = 0;  // 0 off, 1 every kernel, 2 only the rasterization kernels (k_raster*), 3 only the composite backward (k_raster_bwd*)
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;
}  // namespace

ProfScope::ProfScope(const char *name, hipStream_t s) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  if (g_prof_on == 2 && strncmp(name, "k_raster", 8) != 0) return;
  if (g_prof_on == 3 && strncmp(name, "k_raster_bwd", 12) != 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.name = name;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[slot].b, stream);
}

void d4gs_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int d4gs_check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    d4gs_set_error("%s: %s", what, hipGetErrorString(e));
    return D4GS_ELAUNCH;
  }
  return D4GS_OK;
}

static int check_dims(const D4gsDims *d) {
  if (!d) {
    d4gs_set_error("dims is NULL");
    return D4GS_EINVAL;
  }
  if (d->N < 0 || d->S <= 0 || d->width <= 0 || d->height <= 0 || d->D <= 0) {
    d4gs_set_error("bad dims N=%d S=%d W=%d H=%d D=%d", d->N, d->S, d->width, d->height, d->D);
    return D4GS_EINVAL;
  }
  if (d->G < 0 || d->G > d->N || (d->G > 0 && (d->K <= 0 || d->K > D4GS_MAX_K || d->T <= 0))) {
    d4gs_set_error("bad motion dims G=%d K=%d (max %d) T=%d", d->G, d->K, D4GS_MAX_K, d->T);
    return D4GS_EINVAL;
  }
  return D4GS_OK;
}

extern "C" {

int d4gs_version(void) { return D4GS_VERSION; }

// The four counts travel to PINNED host memory by a one-wave kernel that stores them there (pinned memory is device-addressable),
// not by a device-to-host copy: a copy node is a trip through the DMA engine that the kernels behind it in the stream wait for -
// ~30 us per replay of a captured step (bench.py --graph 1.420 against 1.387 ms for the same graph without the copy), and a bubble
// between the forward and the backward of every eager deferred-check step.
__global__ void k_copy_counts(const int64_t *__restrict__ n_isect, int64_t *__restrict__ host_pinned) {
  if (threadIdx.x < 4) __hip_atomic_store(host_pinned + threadIdx.x, n_isect[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // extern "C"
int d4gs_copy_counts_impl(const int64_t *n_isect, int64_t *host_pinned, hipStream_t stream) {
  if (!n_isect || !host_pinned) {
    d4gs_set_error("d4gs_copy_counts: NULL argument");
    return D4GS_EINVAL;
  }
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, host_pinned) != hipSuccess || at.type != hipMemoryTypeHost) {
    (void)hipGetLastError();  // pageable memory: the device cannot address it - an ordinary (for pageable memory: synchronous) copy,
    // which has no place in a stream capture (it would wait for a stream that is only being recorded)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      d4gs_set_error("d4gs_copy_counts: the stream is being captured and the destination is not pinned host memory");
      return D4GS_EINVAL;
    }
    (void)hipGetLastError();
    hipError_t e = hipMemcpyAsync(host_pinned, n_isect, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) {
      d4gs_set_error("d4gs_copy_counts: %s", hipGetErrorString(e));
      return D4GS_ELAUNCH;
    }
    return D4GS_OK;
  }
  D4GS_LAUNCH("k_copy_counts", k_copy_counts, dim3(1), dim3(64), 0, stream, n_isect, host_pinned);
  return d4gs_check_launch("k_copy_counts");
}
extern "C" {
int d4gs_copy_counts(const int64_t *n_isect, int64_t *host_pinned, void *stream) {
  return d4gs_copy_counts_impl(n_isect, host_pinned, (hipStream_t)stream);
}

int d4gs_query_sizes(const D4gsDims *d, D4gsSizes *z) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!z) {
    d4gs_set_error("query_sizes: sizes is NULL");
    return D4GS_EINVAL;
  }
  const int64_t N = d->N, S = d->S, SN = S * N, W = d->width, H = d->height;
  const int64_t tw = (W + D4GS_TILE - 1) / D4GS_TILE, th = (H + D4GS_TILE - 1) / D4GS_TILE;
  const int64_t nch = d->D + (d->depth_mode != D4GS_DEPTH_NONE ? 1 : 0), DP = (d->D + 3) & ~3;
  z->means2d = SN * 2, z->depths = SN, z->conics = SN * 3, z->radii = SN, z->opac_act = N, z->ctab = N * DP;
  z->geom = SN * D4GS_GEOM_STRIDE, z->tile_rects = SN * 2, z->tiles_touched = SN, z->isect_offsets = SN;
  z->tile_counts = 2 * S * tw * th, z->tile_offsets = S * tw * th + 1, z->n_isect = 4;
  z->scan_ws = (int64_t)d4gs_scan_ws_elems(SN);
  z->render_colors = S * H * W * nch, z->render_alphas = S * H * W, z->last_ids = S * H * W, z->final_T = S * H * W;
  z->isect_grad_row = 6 + nch;
  z->bwd_partials = (int64_t)d4gs_bwd_partials_elems(d);
  z->seg_state = d4gs_seg_state_elems(d);
  z->lazy_ws = d4gs_lazy_ws_elems((int)S, (int)(tw * th));
  z->tiles_x = (int32_t)tw, z->tiles_y = (int32_t)th, z->channels = (int32_t)nch;
  z->blend_bases = d->G > 0 ? S * (int64_t)d->K * 16 : 0;
  z->tile_masks = SN;
  return D4GS_OK;
}

void d4gs_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on;
}

/* Waits for the recorded events, writes "name count total_ms\n" per kernel into buf, clears the records. */
int d4gs_profile_collect(char *buf, size_t cap) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  struct Agg { std::string name; int n; double ms; };
  std::vector<Agg> agg;
  for (auto &r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      bool found = false;
      for (auto &x : agg)
        if (x.name == r.name) { x.n++, x.ms += ms; found = true; break; }
      if (!found) agg.push_back({r.name, 1, ms});
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  size_t off = 0;
  for (auto &x : agg) {
    int w = snprintf(buf + off, off < cap ? cap - off : 0, "%s %d %.6f\n", x.name.c_str(), x.n, x.ms);
    if (w < 0 || off + (size_t)w >= cap) break;
    off += (size_t)w;
  }
  if (cap) buf[off < cap ? off : cap - 1] = 0;
  return (int)agg.size();
}
const char *d4gs_last_error(void) { return g_err; }

int d4gs_project_fwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsProjOut *out, void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!in || !out || !in->means || !in->quats || !in->scales || !in->opacities || !in->colors || !in->viewmat ||
      !in->Kmat) {
    d4gs_set_error("d4gs_project_fwd: NULL required input");
    return D4GS_EINVAL;
  }
  if (dims->G > 0 && (!in->motion_coefs || !in->rots || !in->transls || !in->times)) {
    d4gs_set_error("d4gs_project_fwd: G>0 needs motion_coefs/rots/transls/times");
    return D4GS_EINVAL;
  }
  if (dims->N == 0) {
    d4gs_set_error("d4gs_project_fwd: N == 0");
    return D4GS_EINVAL;
  }
  if (((uintptr_t)out->ctab & 15) != 0) {  // the colour-table rows are stored (and read back) 16 bytes at a time
    d4gs_set_error("d4gs_project_fwd: D4gsProjOut.ctab must be 16-byte aligned");
    return D4GS_EINVAL;
  }
  return d4gs_project_fwd_impl(dims, in, out, (hipStream_t)stream);
}

static int check_binned(const char *who, const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect) {
  if (!proj || !isect || !proj->geom || !proj->ctab || !proj->depths || !proj->tile_rects || !proj->tiles_touched ||
      !proj->isect_offsets || !proj->tile_counts || !proj->tile_offsets || !proj->n_isect) {
    d4gs_set_error("%s: NULL projection buffer", who);
    return D4GS_EINVAL;
  }
  if (isect->n_isect > 0 && (!isect->keys || !isect->gid_of_emit || !isect->sorted_gid || !isect->sorted_emit)) {
    d4gs_set_error("%s: NULL intersection list", who);
    return D4GS_EINVAL;
  }
  // tiles_touched / isect_offsets / tile_counts were built from popcount(mask) by d4gs_project_fwd: walking whole rectangles
  // against them (no masks) would run past every range they describe
  if ((dims->flags & D4GS_EXACT_TILES) && !proj->tile_masks) {
    d4gs_set_error("%s: D4GS_EXACT_TILES needs the D4gsProjOut.tile_masks d4gs_project_fwd wrote under the same flag", who);
    return D4GS_EINVAL;
  }
  return D4GS_OK;
}

int d4gs_bin_sort(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if ((rc = check_binned("d4gs_bin_sort", dims, proj, isect))) return rc;
  return d4gs_bin_sort_impl(dims, proj, isect, (hipStream_t)stream);
}

int d4gs_raster_fwd(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, const D4gsRaster *r,
                    void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if ((rc = check_binned("d4gs_raster_fwd", dims, proj, isect))) return rc;
  if (!r || !r->render_colors || !r->render_alphas || !r->last_ids || !r->final_T) {
    d4gs_set_error("d4gs_raster_fwd: NULL output buffer");
    return D4GS_EINVAL;
  }
  return d4gs_raster_fwd_impl(dims, proj, isect, r, (hipStream_t)stream);
}

int d4gs_raster_bwd(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, const D4gsRaster *r,
                    const D4gsRasterGrads *g, void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if ((rc = check_binned("d4gs_raster_bwd", dims, proj, isect))) return rc;
  if (!r || !r->render_colors || !r->render_alphas || !r->last_ids || !r->final_T || !g || !g->v_render_colors ||
      !g->isect_grad || !g->isect_live || !g->v_means2d || !g->v_conics || !g->v_depths || !g->v_opac_act || !g->v_ctab) {
    d4gs_set_error("d4gs_raster_bwd: NULL forward state or gradient buffer");
    return D4GS_EINVAL;
  }
  if (((uintptr_t)g->isect_grad & 15) != 0 || ((uintptr_t)g->isect_live & 3) != 0) {  // k_gather streams the rows as
    d4gs_set_error("d4gs_raster_bwd: isect_grad must be 16-byte aligned, isect_live 4-byte aligned");  // 16-byte words,
    return D4GS_EINVAL;                                                                           // the flags as 4-byte words
  }
  if (g->stats_grad_norm_acc && (!g->stats_vis_count || !g->stats_max_radii || !proj->radii || g->stats_batch_size <= 0)) {
    d4gs_set_error("d4gs_raster_bwd: fused statistics need vis_count, max_radii, radii and a positive batch size");
    return D4GS_EINVAL;
  }
  return d4gs_raster_bwd_impl(dims, proj, isect, r, g, nullptr, (hipStream_t)stream);
}

int d4gs_project_bwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsProjOut *proj, const float *v_means2d,
                     const float *v_conics, const float *v_depths, const float *v_opac_act, const float *v_ctab,
                     const D4gsLeafGrads *grads, void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!in || !proj || !grads || !in->means || !in->quats || !in->scales || !in->opacities || !in->colors || !in->viewmat ||
      !in->Kmat || !proj->radii || !proj->conics || !proj->ctab || !proj->opac_act || !v_means2d || !v_conics ||
      !v_depths || !v_opac_act || !v_ctab || !grads->v_means || !grads->v_quats || !grads->v_scales ||
      !grads->v_opacities || !grads->v_colors || !grads->partials) {
    d4gs_set_error("d4gs_project_bwd: NULL required buffer");
    return D4GS_EINVAL;
  }
  if (dims->G > 0 && (!in->motion_coefs || !in->rots || !in->transls || !in->times || !grads->v_motion_coefs ||
                      !grads->v_rots || !grads->v_transls)) {
    d4gs_set_error("d4gs_project_bwd: G>0 needs motion_coefs/rots/transls/times and their gradient buffers");
    return D4GS_EINVAL;
  }
  if ((((uintptr_t)v_ctab | (uintptr_t)proj->ctab) & 15) != 0) {  // k_project_bwd reads the colour rows and their gradients as float4
    d4gs_set_error("d4gs_project_bwd: v_ctab and D4gsProjOut.ctab must be 16-byte aligned");
    return D4GS_EINVAL;
  }
  return d4gs_project_bwd_impl(dims, in, proj, v_means2d, v_conics, v_depths, v_opac_act, v_ctab, grads,
                               (hipStream_t)stream);
}

static int check_poses(const char *who, const D4gsDims *d, const D4gsProjIn *in, int need_quats) {
  if (!d || !in || d->N <= 0 || d->S <= 0 || d->G < 0 || d->G > d->N || !in->means || (in->viewmat && !in->Kmat) ||
      (need_quats && !in->quats) ||
      (d->G > 0 && (d->K <= 0 || d->K > D4GS_MAX_K || d->T <= 0 || !in->motion_coefs || !in->rots || !in->transls ||
                    !in->times))) {
    d4gs_set_error("%s: bad dims or NULL required input", who);
    return D4GS_EINVAL;
  }
  return D4GS_OK;
}

int d4gs_poses_fwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsPoses *out, void *stream) {
  if (!out || (!out->means && !out->quats && !out->transforms)) {
    d4gs_set_error("d4gs_poses_fwd: no output requested");
    return D4GS_EINVAL;
  }
  int rc = check_poses("d4gs_poses_fwd", dims, in, out->quats != NULL);
  if (rc) return rc;
  return d4gs_poses_fwd_impl(dims, in, out, (hipStream_t)stream);
}

int d4gs_poses_bwd(const D4gsDims *dims, const D4gsProjIn *in, const D4gsPoses *v_out, const D4gsLeafGrads *grads,
                   void *stream) {
  if (!v_out) {
    d4gs_set_error("d4gs_poses_bwd: NULL gradient struct");
    return D4GS_EINVAL;
  }
  int rc = check_poses("d4gs_poses_bwd", dims, in, v_out->quats != NULL);
  if (rc) return rc;
  if (!grads || !grads->v_means || !grads->partials || (v_out->quats && !grads->v_quats) ||
      (dims->G > 0 && (!grads->v_motion_coefs || !grads->v_rots || !grads->v_transls))) {
    d4gs_set_error("d4gs_poses_bwd: NULL gradient buffer");
    return D4GS_EINVAL;
  }
  return d4gs_poses_bwd_impl(dims, in, v_out, grads, (hipStream_t)stream);
}

/* a11 track channels: the means half of the pose API in the target cameras, time-major */
int d4gs_points_fwd(const D4gsDims *dims, const D4gsProjIn *in, float *points, void *stream) {
  if (!points) {
    d4gs_set_error("d4gs_points_fwd: NULL output");
    return D4GS_EINVAL;
  }
  D4gsPoses o = {points, NULL, NULL, 0};
  return d4gs_poses_fwd(dims, in, &o, stream);
}

int d4gs_points_bwd(const D4gsDims *dims, const D4gsProjIn *in, const float *v_points, const D4gsLeafGrads *grads,
                    void *stream) {
  if (!v_points) {
    d4gs_set_error("d4gs_points_bwd: NULL gradient");
    return D4GS_EINVAL;
  }
  D4gsPoses v = {(float *)v_points, NULL, NULL, 0};
  return d4gs_poses_bwd(dims, in, &v, grads, stream);
}

int d4gs_control_stats(int32_t S, int32_t N, const float *xys_grad, const int32_t *radii, int32_t width, int32_t height,
                       int32_t batch_size, float *grad_norm_acc, int64_t *vis_count, float *max_radii,
                       int32_t update_max_radii, void *stream) {
  return d4gs_control_stats_impl(S, N, xys_grad, radii, width, height, batch_size, grad_norm_acc, vis_count, max_radii,
                                 update_max_radii, (hipStream_t)stream);
}

int d4gs_control_plan(int32_t N, const uint8_t *split_or_cull, const uint8_t *dup, int32_t *src_map, int32_t *counts,
                      void *stream) {
  if (N <= 0 || !split_or_cull || !src_map || !counts) {
    d4gs_set_error("d4gs_control_plan: bad argument (N=%d)", N);
    return D4GS_EINVAL;
  }
  return d4gs_control_plan_impl(N, split_or_cull, dup, src_map, counts, (hipStream_t)stream);
}

int d4gs_gather_rows(const int32_t *src_map, int64_t n_out, int32_t row_floats, const float *in, float *out,
                     int64_t zero_from, int64_t add_from, float add, void *stream) {
  if (n_out < 0 || row_floats <= 0 || (n_out > 0 && (!src_map || !in || !out))) {
    d4gs_set_error("d4gs_gather_rows: bad argument (n_out=%lld row_floats=%d)", (long long)n_out, row_floats);
    return D4GS_EINVAL;
  }
  return d4gs_gather_rows_impl(src_map, n_out, row_floats, in, out, zero_from, add_from, add, (hipStream_t)stream);
}

int d4gs_blend_fwd(int32_t S, int64_t n_pixels, int32_t C, const int32_t *policy, const float *renders,
                   const float *alphas, float *out, float *acc, void *stream) {
  return d4gs_blend_fwd_impl(S, n_pixels, C, policy, renders, alphas, out, acc, nullptr, (hipStream_t)stream);
}

int d4gs_blend_bwd(int32_t S, int64_t n_pixels, int32_t C, const int32_t *policy, const float *renders,
                   const float *out, const float *v_out, const float *v_acc, float *v_renders, float *v_alphas,
                   void *stream) {
  return d4gs_blend_bwd_impl(S, n_pixels, C, policy, renders, out, v_out, v_acc, v_renders, v_alphas,
                             (hipStream_t)stream);
}

int d4gs_blend_shard_partial_fwd(const D4gsShardBlend *b, const float *renders, const float *alphas, float *part, float *cand,
                                 void *stream) {
  if (!part || (b && b->S_local > 0 && (!renders || !alphas))) {
    d4gs_set_error("blend_shard_partial_fwd: NULL buffer");
    return D4GS_EINVAL;
  }
  return d4gs_blend_shard_impl(0, b, renders, alphas, nullptr, part, cand, (hipStream_t)stream);
}

int d4gs_blend_shard_finish_fwd(const D4gsShardBlend *b, const float *part, const float *cand, float *out, float *acc,
                                void *stream) {
  if (!part || !out || !acc) {
    d4gs_set_error("blend_shard_finish_fwd: NULL buffer");
    return D4GS_EINVAL;
  }
  return d4gs_blend_shard_impl(1, b, part, cand, nullptr, out, acc, (hipStream_t)stream);
}

int d4gs_blend_shard_winner(const D4gsShardBlend *b, const float *renders, const float *out, int32_t *win, void *stream) {
  if (!out || !win) {
    d4gs_set_error("blend_shard_winner: NULL buffer");
    return D4GS_EINVAL;
  }
  return d4gs_blend_shard_impl(2, b, renders, out, nullptr, win, nullptr, (hipStream_t)stream);
}

int d4gs_blend_shard_bwd(const D4gsShardBlend *b, const float *v_out, const float *v_acc, const int32_t *win,
                         float *v_renders, float *v_alphas, void *stream) {
  if (!v_out || (b && b->S_local > 0 && (!v_renders || !v_alphas))) {
    d4gs_set_error("blend_shard_bwd: NULL buffer");
    return D4GS_EINVAL;
  }
  return d4gs_blend_shard_impl(3, b, v_out, v_acc, win, v_renders, v_alphas, (hipStream_t)stream);
}

int d4gs_pose_encode(const float *R, int32_t r_stride, const float *T, int32_t t_stride, float *enc, void *stream) {
  if (!R || !T || !enc || r_stride < 3 || t_stride < 1) {
    d4gs_set_error("pose_encode: NULL buffer or bad stride (r_stride=%d t_stride=%d)", r_stride, t_stride);
    return D4GS_EINVAL;
  }
  return d4gs_pose_encode_impl(R, r_stride, T, t_stride, enc, (hipStream_t)stream);
}

int d4gs_camera_path_fwd(const float *delta0, const float *delta1, int32_t S, const float *time_params,
                         int32_t n_time_params, int32_t index, float t, float *RTs, float *jac, float *times,
                         float *dtimes, float *deltaT, void *stream) {
  if (S <= 0 || S > 4096) {
    d4gs_set_error("camera_path: bad S=%d", S);
    return D4GS_EINVAL;
  }
  if (!delta0 || !delta1 || !RTs || !times || !dtimes || !deltaT) {
    d4gs_set_error("camera_path: NULL buffer");
    return D4GS_EINVAL;
  }
  /* move_model.py:118-131: only interior frames of the per-frame table move the exposure window */
  int moving = time_params != NULL && index > 0 && index < n_time_params - 1;
  return d4gs_camera_path_fwd_impl(delta0, delta1, S, time_params, index, t, moving, RTs, jac, times, dtimes, deltaT,
                                   (hipStream_t)stream);
}

int d4gs_camera_path_bwd(const float *jac, const float *dtimes, const float *deltaT, const float *v_RTs,
                         const float *v_times, const float *v_deltaT, int32_t S, int32_t index,
                         int32_t n_time_params, float *v_delta0, float *v_delta1, float *v_time_params,
                         void *stream) {
  if (S <= 0 || n_time_params < 0 || n_time_params > 52 || !jac || !dtimes || !deltaT || !v_delta0 || !v_delta1 ||
      (n_time_params > 0 && !v_time_params)) {
    d4gs_set_error("camera_path_bwd: bad arguments (S=%d n_time_params=%d)", S, n_time_params);
    return D4GS_EINVAL;
  }
  return d4gs_camera_path_bwd_impl(jac, dtimes, deltaT, v_RTs, v_times, v_deltaT, S, index, n_time_params, v_delta0,
                                   v_delta1, v_time_params, (hipStream_t)stream);
}

int d4gs_move_model_fwd(const float *R, int32_t r_stride, const float *T, int32_t t_stride, const D4gsMoveModelParams *p,
                        int32_t S, int32_t index, float t, int32_t stage_first, const D4gsMoveModelOut *o, void *stream) {
  if (!R || !T || !p || !o || r_stride < 3 || t_stride < 1 || S <= 0 || S > 4096) {
    d4gs_set_error("move_model_fwd: bad arguments (S=%d r_stride=%d t_stride=%d)", S, r_stride, t_stride);
    return D4GS_EINVAL;
  }
  for (int l = 0; l < 9; l++)
    if (!p->w[l] || !p->b[l]) {
      d4gs_set_error("move_model_fwd: layer %d has a NULL weight or bias", l);
      return D4GS_EINVAL;
    }
  if (!o->enc || !o->acts || !o->delta || !o->RTs || !o->times || !o->dtimes || !o->deltaT) {
    d4gs_set_error("move_model_fwd: NULL output buffer");
    return D4GS_EINVAL;
  }
  const int moving = !stage_first && p->time_params && index > 0 && index < p->n_time_params - 1;
  return d4gs_move_model_fwd_impl(R, r_stride, T, t_stride, p->w, p->b, S, p->time_params, index, t, moving, o->enc,
                                  o->acts, o->delta, o->RTs, o->jac, o->times, o->dtimes, o->deltaT, (hipStream_t)stream);
}

int d4gs_move_model_bwd(const D4gsMoveModelParams *p, const D4gsMoveModelOut *o, const float *v_RTs, const float *v_times,
                        const float *v_deltaT, int32_t S, int32_t index, const D4gsMoveModelGrads *g, void *stream) {
  if (!p || !o || !g || S <= 0 || p->n_time_params < 0 || p->n_time_params > 52 || !o->jac || !o->acts || !g->v_delta ||
      (p->n_time_params > 0 && !g->v_time_params)) {
    d4gs_set_error("move_model_bwd: bad arguments");
    return D4GS_EINVAL;
  }
  for (int l = 0; l < 9; l++)
    if (!g->v_w[l] || !g->v_b[l]) {
      d4gs_set_error("move_model_bwd: layer %d has a NULL gradient buffer", l);
      return D4GS_EINVAL;
    }
  return d4gs_move_model_bwd_impl(o->jac, o->dtimes, o->deltaT, o->acts, p->w, p->b, v_RTs, v_times, v_deltaT, S, index,
                                  p->n_time_params, g->v_delta, g->v_w, g->v_b, g->v_time_params, g->v_enc,
                                  (hipStream_t)stream);
}

int d4gs_pose_encode_bwd(const float *R, int32_t r_stride, const float *T, int32_t t_stride, const float *v_enc,
                         float *v_R, float *v_T, void *stream) {
  if (!R || !T || !v_enc || !v_R || !v_T || r_stride < 3 || t_stride < 1) {
    d4gs_set_error("pose_encode_bwd: NULL buffer or bad stride (r_stride=%d t_stride=%d)", r_stride, t_stride);
    return D4GS_EINVAL;
  }
  return d4gs_pose_encode_bwd_impl(R, r_stride, T, t_stride, v_enc, v_R, v_T, (hipStream_t)stream);
}

int d4gs_photometric_fwd(const float *pred, const float *gt, const float *mask, int32_t B, int32_t H, int32_t W, int32_t C,
                         float w_l1, float w_ssim, float *maps, float *partials, float *loss, void *stream) {
  if (C != 3 || B <= 0 || H < 11 || W < 11) {
    d4gs_set_error("photometric: needs C == 3 and H, W >= 11 (B=%d H=%d W=%d C=%d)", B, H, W, C);
    return D4GS_EINVAL;
  }
  if (!pred || !gt || !maps || !partials || !loss) {
    d4gs_set_error("photometric_fwd: NULL buffer");
    return D4GS_EINVAL;
  }
  return d4gs_photometric_fwd_impl(pred, gt, mask, B, H, W, w_l1, w_ssim, maps, partials, loss, (hipStream_t)stream);
}

int d4gs_photometric_bwd(const float *pred, const float *gt, const float *mask, const float *maps, const float *v_loss,
                         int32_t B, int32_t H, int32_t W, int32_t C, float w_l1, float w_ssim, float *v_pred,
                         void *stream) {
  if (C != 3 || B <= 0 || H < 11 || W < 11 || !pred || !gt || !maps || !v_loss || !v_pred) {
    d4gs_set_error("photometric_bwd: bad arguments");
    return D4GS_EINVAL;
  }
  return d4gs_photometric_bwd_impl(pred, gt, mask, maps, v_loss, B, H, W, w_l1, w_ssim, v_pred, (hipStream_t)stream);
}

}  // extern "C"
