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

}  // namespace

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
