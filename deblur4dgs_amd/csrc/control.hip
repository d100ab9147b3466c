// control.hip -- densification statistics, the direct consumer of the raster
// backward's `means2d.grad` and of `radii`.
//
// Replaces the per-sub-sample loop of Trainer._prepare_control_step (reference flow3d/trainer.py:953-990):
// S x { where / clone / 2 scalings / norm / index_add_ x2 / index_select / maximum / index_put } torch launches
// become one streaming kernel: one lane per Gaussian walks its S sub-samples in order (same summation order as the
// reference's sequential index_add_), 12 B read per instance, 16 B read-modify-write per Gaussian.  HBM-bound.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) k_control_stats(int S, int N, const float *xys_grad, const int32_t *radii,
                                                       float sx, float sy, float max_wh, float *grad_norm_acc,
                                                       int64_t *vis_count, const float *max_radii, float *max_radii_out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  float acc = grad_norm_acc[g], mr = max_radii[g];
  int64_t vc = vis_count[g];
  for (int s = 0; s < S; s++) {
    const size_t i = (size_t)s * N + g;
    const int r = radii[i];
    if (r > 0) {
      const float2 v = *reinterpret_cast<const float2 *>(xys_grad + i * 2);
      const float gx = v.x * sx, gy = v.y * sy;
      acc += sqrtf(gx * gx + gy * gy);
      vc += 1;
      mr = fmaxf(mr, (float)r / max_wh);
    }
  }
  grad_norm_acc[g] = acc;
  vis_count[g] = vc;
  if (max_radii_out) max_radii_out[g] = mr;
}

// ---------------------------------------------------------------------------------------------------------------
// Row surgery of a control step (GaussianParams.densify_params / cull_params, flow3d/params.py:86-118, and the Adam
// state surgery of dup_in_optim / remove_from_optim, flow3d/trainer.py:1199-1236) as stream compaction:
//   k_control_plan   one pass over the decision flags -> `src[i]` = source row of output row i, in the reference's
//                    row order: kept rows (not split), then duplicated rows, then every split row twice (cull: kept
//                    rows only).  counts = {n_keep, n_dup, n_split, n_out}.
//   k_gather_rows    out[i] = in[src[i]] for ONE tensor of row width w (parameters, exp_avg, exp_avg_sq, running
//                    statistics); rows >= zero_from are zero-filled (fresh Adam moments), rows >= add_from get +add
//                    (the two halves of a split: log-scale - log 1.6).
// A control step runs every 100 training steps on a few MB: one single-block plan (N / 1024 chunked block scans) and
// one streaming launch per tensor instead of ~15 boolean-index / cat launches per tensor with a host sync each.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_scan_1024(int v, int *total, int *lds /* [17] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < 16; w++) {
    const int x = lds[w];
    if (w < wave) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(1024) k_control_plan(int N, const uint8_t *__restrict__ split,
                                                        const uint8_t *__restrict__ dup, int32_t *__restrict__ src,
                                                        int32_t *__restrict__ counts) {
  __shared__ int lds[17];
  __shared__ int tot[3];
  // category 0: kept (not split / not culled), 1: duplicated, 2: split
  auto flag = [&](int c, int g) -> int {
    if (g >= N) return 0;
    return c == 0 ? !split[g] : c == 1 ? (dup ? dup[g] != 0 : 0) : (dup ? split[g] != 0 : 0);
  };
  for (int c = 0; c < 3; c++) {  // totals first: the offsets of categories 1 and 2 depend on them
    int carry = 0;
    for (int base = 0; base < N; base += 1024) {
      int t;
      block_scan_1024(flag(c, base + threadIdx.x), &t, lds);
      carry += t;
    }
    if (threadIdx.x == 0) tot[c] = carry;
  }
  __syncthreads();
  const int n_keep = tot[0], n_dup = tot[1], n_split = tot[2];
  for (int c = 0; c < 3; c++) {
    const int off = c == 0 ? 0 : c == 1 ? n_keep : n_keep + n_dup;
    int carry = 0;
    for (int base = 0; base < N; base += 1024) {
      const int g = base + threadIdx.x, f = flag(c, g);
      int t;
      const int ex = block_scan_1024(f, &t, lds);
      if (f) {
        src[off + carry + ex] = g;
        if (c == 2) src[off + n_split + carry + ex] = g;  // `x[should_split].repeat(2)`: the block of split rows twice
      }
      carry += t;
    }
  }
  if (threadIdx.x == 0) counts[0] = n_keep, counts[1] = n_dup, counts[2] = n_split, counts[3] = n_keep + n_dup + 2 * n_split;
}

__global__ void __launch_bounds__(256) k_gather_rows(const int32_t *__restrict__ src, int64_t n_out, int w,
                                                     const float *__restrict__ in, float *__restrict__ out,
                                                     int64_t zero_from, int64_t add_from, float add) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_out * w) return;
  const int64_t i = idx / w;
  const int c = (int)(idx - i * w);
  float v = i >= zero_from ? 0.f : in[(int64_t)src[i] * w + c];
  if (i >= add_from) v += add;
  out[idx] = v;
}

}  // namespace

int d4gs_control_plan_impl(int32_t N, const uint8_t *split_or_cull, const uint8_t *dup, int32_t *src_map, int32_t *counts,
                           hipStream_t stream) {
  D4GS_LAUNCH("k_control_plan", k_control_plan, dim3(1), dim3(1024), 0, stream, N, split_or_cull, dup, src_map, counts);
  return d4gs_check_launch("k_control_plan");
}

int d4gs_gather_rows_impl(const int32_t *src_map, int64_t n_out, int32_t row_floats, const float *in, float *out,
                          int64_t zero_from, int64_t add_from, float add, hipStream_t stream) {
  const int64_t n = n_out * row_floats;
  if (n == 0) return D4GS_OK;
  D4GS_LAUNCH("k_gather_rows", k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src_map, n_out,
              row_floats, in, out, zero_from, add_from, add);
  return d4gs_check_launch("k_gather_rows");
}

int d4gs_control_stats_impl(int32_t S, int32_t N, const float *xys_grad, const int32_t *radii, int32_t width,
                            int32_t height, int32_t batch_size, float *grad_norm_acc, int64_t *vis_count,
                            float *max_radii, int32_t update_max_radii, hipStream_t stream) {
  if (S <= 0 || N <= 0 || !xys_grad || !radii || !grad_norm_acc || !vis_count || !max_radii) {
    d4gs_set_error("d4gs_control_stats: bad argument");
    return D4GS_EINVAL;
  }
  // xys_grad[..., 0] *= W / 2 * batch_size * S ; [..., 1] *= H / 2 * batch_size * S   (trainer.py:976-977)
  const float sx = (float)width / 2.0f * (float)batch_size * (float)S;
  const float sy = (float)height / 2.0f * (float)batch_size * (float)S;
  const float inv = (float)(width > height ? width : height);
  D4GS_LAUNCH("k_control_stats", k_control_stats, dim3((N + 255) / 256), dim3(256), 0, stream, S, N, xys_grad, radii, sx,
              sy, inv, grad_norm_acc, vis_count, (const float *)max_radii, update_max_radii ? max_radii : (float *)nullptr);
  return d4gs_check_launch("k_control_stats");
}
