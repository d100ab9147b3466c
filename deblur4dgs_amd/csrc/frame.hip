// frame.hip -- one-call entry points of the path: d4gs_forward = deform + project + bin + sort + composite (+ the
// exposure blend) of all S sub-samples, d4gs_backward = its adjoint down to the leaf gradients; every scratch buffer of
// both lives in ONE caller-provided workspace (d4gs_frame_workspace_bytes).  Host code only: the stages are the same
// kernels the staged entry points launch (d4gs_project_fwd ... d4gs_project_bwd), in the same order, so results are
// bit-identical to the staged calls; what this saves is the caller's per-stage allocation / binding work (one ctypes
// call and one workspace tensor instead of five calls and ~30 tensors per direction).
#include "common.h"

int d4gs_project_fwd_impl(const D4gsDims *, const D4gsProjIn *, const D4gsProjOut *, hipStream_t);
int d4gs_bin_sort_impl(const D4gsDims *, const D4gsProjOut *, const D4gsIsect *, hipStream_t);
int d4gs_raster_fwd_impl(const D4gsDims *, const D4gsProjOut *, const D4gsIsect *, const D4gsRaster *, hipStream_t);
int d4gs_raster_bwd_impl(const D4gsDims *, const D4gsProjOut *, const D4gsIsect *, const D4gsRaster *,
                         const D4gsRasterGrads *, const BlendAdj *, hipStream_t);
int d4gs_project_bwd_impl(const D4gsDims *, const D4gsProjIn *, const D4gsProjOut *, const float *, const float *,
                          const float *, const float *, const float *, const D4gsLeafGrads *, hipStream_t);
int d4gs_blend_fwd_impl(int32_t, int64_t, int32_t, const int32_t *, const float *, const float *, float *, float *, int8_t *,
                        hipStream_t, const int64_t *n_isect = nullptr, int64_t *counts_pinned = nullptr);
int d4gs_copy_counts_impl(const int64_t *n_isect, int64_t *host_pinned, hipStream_t stream);
int d4gs_blend_bwd_add_impl(int32_t, int64_t, int32_t, const int32_t *, const float *, const float *, const float *,
                            const float *, float *, float *, const float *, const float *, hipStream_t, const int8_t *win = nullptr);

namespace {

struct Carve {
  char *base;
  size_t off;
  template <typename T>
  T *take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct FrameBufs {
  D4gsProjOut proj;
  D4gsIsect isect;
  D4gsRaster raster;
  float *v_renders, *v_alphas, *isect_grad, *v_conics, *v_depths, *v_opac_act, *v_ctab, *partials;
  uint8_t *isect_live;
  int8_t *blend_win;  // [H*W][channels] who takes the gradient of a max / min channel (k_blend_fwd -> the composite backward's prologue)
  size_t bytes;      // the whole workspace
  size_t fwd_bytes;  // its prefix d4gs_forward uses (the backward scratch follows)
};

FrameBufs carve(const D4gsDims *d, int64_t cap, void *ws) {
  FrameBufs b{};
  Carve c{reinterpret_cast<char *>(ws), 0};
  D4gsSizes z;
  (void)d4gs_query_sizes(d, &z);
  const size_t m = (size_t)(cap > 0 ? cap : 1);
  b.proj.depths = c.take<float>(z.depths), b.proj.conics = c.take<float>(z.conics);
  b.proj.opac_act = c.take<float>(z.opac_act), b.proj.ctab = c.take<float>(z.ctab), b.proj.geom = c.take<float>(z.geom);
  b.proj.tile_rects = c.take<int32_t>(z.tile_rects), b.proj.tiles_touched = c.take<int32_t>(z.tiles_touched);
  b.proj.isect_offsets = c.take<int32_t>(z.isect_offsets), b.proj.tile_counts = c.take<int32_t>(z.tile_counts);
  b.proj.tile_offsets = c.take<int32_t>(z.tile_offsets), b.proj.scan_ws = c.take<int32_t>(z.scan_ws);
  b.proj.lazy_ws = (d->flags & D4GS_LAZY_SORT) ? c.take<int32_t>(z.lazy_ws) : nullptr;
  b.proj.blend_bases = z.blend_bases > 0 ? c.take<float>(z.blend_bases) : nullptr;
  b.proj.tile_masks = (d->flags & D4GS_EXACT_TILES) ? c.take<uint64_t>(z.tile_masks) : nullptr;
  b.isect.keys = c.take<uint64_t>(m), b.isect.gid_of_emit = c.take<int32_t>(m);
  b.isect.sorted_gid = c.take<int32_t>(m), b.isect.sorted_emit = c.take<int32_t>(m);
  b.raster.last_ids = c.take<int32_t>(z.last_ids), b.raster.final_T = c.take<float>(z.final_T);
  b.raster.seg_state = z.seg_state > 0 ? c.take<float>(z.seg_state) : nullptr;  // few-tile launches: depth-segment boundary states
  b.blend_win = c.take<int8_t>((size_t)(z.render_colors / (d->S > 0 ? d->S : 1)));
  b.fwd_bytes = (c.off + 255) & ~(size_t)255;
  // backward scratch
  b.v_renders = c.take<float>(z.render_colors), b.v_alphas = c.take<float>(z.render_alphas);
  b.isect_grad = c.take<float>(m * (size_t)z.isect_grad_row), b.isect_live = c.take<uint8_t>((m + 3) & ~(size_t)3);
  b.v_conics = c.take<float>(z.conics), b.v_depths = c.take<float>(z.depths);
  b.v_opac_act = c.take<float>(z.opac_act), b.v_ctab = c.take<float>(z.ctab), b.partials = c.take<float>(z.bwd_partials);
  b.bytes = (c.off + 255) & ~(size_t)255;
  return b;
}

int check_frame(const char *who, const D4gsDims *d, const D4gsProjIn *in, const D4gsFrameIO *io, void *ws, size_t ws_bytes,
                int64_t cap, bool forward_only = false) {
  D4gsSizes z;
  int rc = d4gs_query_sizes(d, &z);  // validates dims
  if (rc) return rc;
  if (!in || !io || !ws || !io->renders || !io->alphas || !io->means2d || !io->radii || !io->n_isect ||
      (io->blended && !io->acc) || !in->means || !in->quats || !in->scales || !in->opacities || !in->colors || !in->viewmat ||
      !in->Kmat || (d->G > 0 && (!in->motion_coefs || !in->rots || !in->transls || !in->times)) || d->N == 0) {
    d4gs_set_error("%s: NULL required buffer", who);
    return D4GS_EINVAL;
  }
  if (((uintptr_t)ws & 255) != 0) {
    d4gs_set_error("%s: the workspace must be 256-byte aligned", who);
    return D4GS_EINVAL;
  }
  const FrameBufs need = carve(d, cap, nullptr);
  if (ws_bytes < (forward_only ? need.fwd_bytes : need.bytes)) {
    d4gs_set_error("%s: workspace of %zu bytes, need %zu (%s)", who, ws_bytes, forward_only ? need.fwd_bytes : need.bytes,
                   forward_only ? "d4gs_frame_workspace_bytes_fwd; d4gs_backward needs d4gs_frame_workspace_bytes" : "d4gs_frame_workspace_bytes");
    return D4GS_ECAPACITY;
  }
  return D4GS_OK;
}

void bind_io(FrameBufs &b, const D4gsFrameIO *io, int64_t cap, int64_t max_hint) {
  b.proj.means2d = io->means2d, b.proj.radii = io->radii, b.proj.n_isect = io->n_isect;
  b.isect.n_isect = cap > 0 ? cap : 1, b.isect.max_tile_count = max_hint, b.isect.near_target = io->near_target;
  b.raster.background = io->background, b.raster.render_colors = io->renders, b.raster.render_alphas = io->alphas;
}

}  // namespace

extern "C" {

size_t d4gs_frame_workspace_bytes(const D4gsDims *dims, int64_t isect_capacity) {
  D4gsSizes z;
  if (d4gs_query_sizes(dims, &z)) return 0;
  return carve(dims, isect_capacity, nullptr).bytes;
}

size_t d4gs_frame_workspace_bytes_fwd(const D4gsDims *dims, int64_t isect_capacity) {
  D4gsSizes z;
  if (d4gs_query_sizes(dims, &z)) return 0;
  return carve(dims, isect_capacity, nullptr).fwd_bytes;
}

int d4gs_forward(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io, void *ws, size_t ws_bytes,
                 int64_t isect_capacity, int64_t max_tile_hint, void *stream_) {
  int rc = check_frame("d4gs_forward", dims, in, io, ws, ws_bytes, isect_capacity, /*forward_only=*/true);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (io->counts_pinned) {  // a kernel stores to it: it must be device-addressable host memory (hipHostMalloc / torch pin_memory)
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, io->counts_pinned) != hipSuccess || at.type != hipMemoryTypeHost) {
      (void)hipGetLastError();
      d4gs_set_error("d4gs_forward: io->counts_pinned is not pinned host memory");
      return D4GS_EINVAL;
    }
  }
  FrameBufs b = carve(dims, isect_capacity, ws);
  bind_io(b, io, isect_capacity, max_tile_hint);
  if ((rc = d4gs_project_fwd_impl(dims, in, &b.proj, stream))) return rc;
  if (isect_capacity < 0)  // COUNT ONLY: io->n_isect (and means2d / radii) are what the caller wanted
    return io->counts_pinned ? d4gs_copy_counts_impl(io->n_isect, io->counts_pinned, stream) : D4GS_OK;
  if ((rc = d4gs_bin_sort_impl(dims, &b.proj, &b.isect, stream))) return rc;
  if ((rc = d4gs_raster_fwd_impl(dims, &b.proj, &b.isect, &b.raster, stream))) return rc;
  if (io->blended) {
    const int nch = dims->D + (dims->depth_mode != D4GS_DEPTH_NONE ? 1 : 0);
    rc = d4gs_blend_fwd_impl(dims->S, (int64_t)dims->width * dims->height, nch, io->policy, io->renders, io->alphas, io->blended,
                             io->acc, b.blend_win, stream,  // (the winner map: read by the folded adjoint or by k_blend_bwd)
                             io->n_isect, io->counts_pinned);
  } else if (io->counts_pinned) {
    rc = d4gs_copy_counts_impl(io->n_isect, io->counts_pinned, stream);
  }
  return rc;
}

int d4gs_backward(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io, const D4gsFrameGrads *g,
                  const D4gsLeafGrads *leaf, void *ws, size_t ws_bytes, int64_t isect_capacity, int64_t max_tile_hint,
                  void *stream_) {
  int rc = check_frame("d4gs_backward", dims, in, io, ws, ws_bytes, isect_capacity);
  if (rc) return rc;
  if (!g || !leaf || !g->v_means2d || !leaf->v_means || !leaf->v_quats || !leaf->v_scales || !leaf->v_opacities ||
      !leaf->v_colors || (dims->G > 0 && (!leaf->v_motion_coefs || !leaf->v_rots || !leaf->v_transls)) ||
      (io->blended ? (!g->v_blended && !g->v_acc && !g->v_renders && !g->v_alphas) : !g->v_renders)) {
    d4gs_set_error("d4gs_backward: NULL gradient buffer (a blended frame takes any of v_blended, v_acc, v_renders, v_alphas; an "
                   "unblended one v_renders [+ v_alphas])");
    return D4GS_EINVAL;
  }
  hipStream_t stream = (hipStream_t)stream_;
  FrameBufs b = carve(dims, isect_capacity, ws);
  bind_io(b, io, isect_capacity, max_tile_hint);
  const float *v_renders = g->v_renders, *v_alphas = g->v_alphas;
  // The blend's adjoint rides in the composite backward's prologue (common.h BlendAdj) unless the caller also holds gradients on the
  // sub-sample images themselves: no k_blend_bwd launch, no [S,H,W,channels] gradient stack written and read back.
  // D4GS_FUSE_BLEND_BWD=0: the separate kernel (A/B; same bits).
  static const bool fuse_env = []() { const char *e = getenv("D4GS_FUSE_BLEND_BWD"); return !(e && e[0] == '0'); }();
  const int nch_all = dims->D + (dims->depth_mode != D4GS_DEPTH_NONE ? 1 : 0);
  // (narrow renders only: the 17-channel composite backward sits at its register budget - with the blend prologue it ran 1 018 -> 1 086 us
  // on the reference's training shape, more than the 42 us k_blend_bwd costs there; profiles/r04t_ab_blend_bwd.txt)
  const bool fuse_blend = fuse_env && io->blended && !g->v_renders && !g->v_alphas && dims->D <= 5;
  BlendAdj ba{};
  if (fuse_blend) {
    ba.v_blended = g->v_blended, ba.v_acc = g->v_acc, ba.win = b.blend_win;
    for (int c = 0; c < nch_all; c++)
      if (io->policy && io->policy[c] != 0) ba.non_mean |= (uint64_t)1 << c;
    v_renders = nullptr, v_alphas = nullptr;
  } else if (io->blended) {
    const int nch = dims->D + (dims->depth_mode != D4GS_DEPTH_NONE ? 1 : 0);
    // (gradients the caller holds on the sub-sample images themselves are summed in by the same kernel)
    if ((rc = d4gs_blend_bwd_add_impl(dims->S, (int64_t)dims->width * dims->height, nch, io->policy, io->renders, io->blended,
                                      g->v_blended, g->v_acc, b.v_renders, b.v_alphas, g->v_renders, g->v_alphas, stream,
                                      b.blend_win)))
      return rc;
    v_renders = b.v_renders, v_alphas = b.v_alphas;
  }
  D4gsRasterGrads rg{};
  rg.v_render_colors = v_renders, rg.v_render_alphas = v_alphas, rg.isect_grad = b.isect_grad, rg.isect_live = b.isect_live;
  rg.v_means2d = g->v_means2d, rg.v_conics = b.v_conics, rg.v_depths = b.v_depths, rg.v_opac_act = b.v_opac_act;
  rg.v_ctab = b.v_ctab;
  rg.stats_grad_norm_acc = g->stats_grad_norm_acc, rg.stats_vis_count = g->stats_vis_count;
  rg.stats_max_radii = g->stats_max_radii, rg.stats_batch_size = g->stats_batch_size;
  rg.stats_update_max_radii = g->stats_update_max_radii, rg.row_mode = g->row_mode;
  if (rg.stats_grad_norm_acc && (!rg.stats_vis_count || !rg.stats_max_radii || rg.stats_batch_size <= 0)) {
    d4gs_set_error("d4gs_backward: fused statistics need vis_count, max_radii and a positive batch size");
    return D4GS_EINVAL;
  }
  if ((rc = d4gs_raster_bwd_impl(dims, &b.proj, &b.isect, &b.raster, &rg, fuse_blend ? &ba : nullptr, stream))) return rc;
  D4gsLeafGrads lg = *leaf;
  lg.partials = b.partials;
  return d4gs_project_bwd_impl(dims, in, &b.proj, g->v_means2d, b.v_conics, b.v_depths, b.v_opac_act, b.v_ctab, &lg, stream);
}

}  // extern "C"
