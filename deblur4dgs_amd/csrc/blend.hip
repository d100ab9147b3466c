// blend.hip -- exposure blend of the S sub-sample renders into the blurry frame, forward and backward.
//
// Replaces flow3d/scene_model.py:386-397 (three torch.stack of the full S-stack + mean / max / min):
//   out[c] = mean_s raw_s[c]                      for every channel,
//   policy[c] == 1: out[c] = max{raw_0..raw_{S-2}, mean}   (reference writes the mean in place into the last
//   policy[c] == 2: out[c] = min{raw_0..raw_{S-2}, mean}    sub-sample BEFORE taking max/min - reproduced)
//   acc = mean_s alpha_s.
// One lane per (pixel, channel); streams S values once (HBM-bound).  Ties resolve to the lowest s, then the mean,
// like torch.max/min(dim=0) on the reference's stack order.
#include "common.h"

namespace {

struct Policy {
  int8_t p[64];
};

__global__ void __launch_bounds__(256) k_blend_fwd(int S, int64_t P, int C, const Policy policy, const float *renders,
                                                   const float *alphas, float *out, float *acc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t PC = P * C;
  if (i < PC) {
    const int c = (int)(i % C);
    float sum = 0.f;
    for (int s = 0; s < S; s++) sum += renders[s * PC + i];
    const float mean = (S == 1) ? renders[i] : sum / (float)S;
    float v = mean;
    const int pol = policy.p[c];
    if (pol == 1) {
      for (int s = 0; s + 1 < S; s++) v = fmaxf(v, renders[s * PC + i]);
    } else if (pol == 2) {
      for (int s = 0; s + 1 < S; s++) v = fminf(v, renders[s * PC + i]);
    }
    out[i] = v;
  }
  if (i < P) {
    float sum = 0.f;
    for (int s = 0; s < S; s++) sum += alphas[s * P + i];
    acc[i] = (S == 1) ? alphas[i] : sum / (float)S;
  }
}

__global__ void __launch_bounds__(256) k_blend_bwd(int S, int64_t P, int C, const Policy policy, const float *renders,
                                                   const float *out, const float *v_out, const float *v_acc,
                                                   float *v_renders, float *v_alphas) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t PC = P * C;
  if (i < PC) {
    const int c = (int)(i % C);
    const int pol = policy.p[c];
    const float g = v_out[i];
    const float inv = 1.f / (float)S;
    int winner = -1;  // -1: the mean receives the gradient
    if (pol != 0 && S > 1) {
      const float o = out[i];
      for (int s = 0; s + 1 < S; s++)
        if (renders[s * PC + i] == o) {
          winner = s;
          break;
        }
    }
    for (int s = 0; s < S; s++) v_renders[s * PC + i] = (winner < 0) ? g * inv : (s == winner ? g : 0.f);
  }
  if (i < P) {
    const float g = v_acc ? v_acc[i] / (float)S : 0.f;
    for (int s = 0; s < S; s++) v_alphas[s * P + i] = g;
  }
}

}  // namespace

int d4gs_blend_fwd_impl(int32_t S, int64_t P, int32_t C, const int32_t *policy, const float *renders,
                        const float *alphas, float *out, float *acc, hipStream_t stream) {
  const int64_t n = P * C;
  Policy pol;
  if (C > 64 || C <= 0) {
    d4gs_set_error("blend: C=%d out of range (1..64)", C);
    return D4GS_EINVAL;
  }
  for (int c = 0; c < 64; c++) pol.p[c] = (c < C && policy) ? (int8_t)policy[c] : 0;
  D4GS_LAUNCH("k_blend_fwd", k_blend_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, P, C, pol, renders,
                     alphas, out, acc);
  return d4gs_check_launch("k_blend_fwd");
}

int d4gs_blend_bwd_impl(int32_t S, int64_t P, int32_t C, const int32_t *policy, const float *renders, const float *out,
                        const float *v_out, const float *v_acc, float *v_renders, float *v_alphas, hipStream_t stream) {
  const int64_t n = P * C;
  Policy pol;
  if (C > 64 || C <= 0) {
    d4gs_set_error("blend: C=%d out of range (1..64)", C);
    return D4GS_EINVAL;
  }
  for (int c = 0; c < 64; c++) pol.p[c] = (c < C && policy) ? (int8_t)policy[c] : 0;
  D4GS_LAUNCH("k_blend_bwd", k_blend_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, P, C, pol, renders,
                     out, v_out, v_acc, v_renders, v_alphas);
  return d4gs_check_launch("k_blend_bwd");
}
