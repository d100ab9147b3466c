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

// `win` (optional, [P][C] bytes): for the max / min channels, the sub-sample k_blend_bwd's first-match rule would hand the gradient to
// (-1: the mean) - the one-call backward reads it instead of re-deriving it from the S - 1 renders (frame.hip, BlendAdj)
// sum of x[s * stride] over s < S in ascending s, the loads of eight sub-samples in flight at a time.  (Round 5: written as a plain
// `for` the compiler kept ONE load in flight per lane - load, s_waitcnt vmcnt(0), add, branch - so a lane paid S dependent memory
// round trips: k_blend_fwd ran at 2.0 - 2.4 TB/s; same additions in the same order, bit for bit.)
__device__ __forceinline__ float d4gs_sum_strided(const float *__restrict__ x, int S, int64_t stride) {
  float sum = 0.f;
  int s = 0;
  for (; s + 8 <= S; s += 8) {
    float r[8];
#pragma unroll
    for (int u = 0; u < 8; u++) r[u] = x[(int64_t)(s + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u++) sum += r[u];
  }
  if (s + 4 <= S) {
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; u++) r[u] = x[(int64_t)(s + u) * stride];
#pragma unroll
    for (int u = 0; u < 4; u++) sum += r[u];
    s += 4;
  }
  for (; s < S; s++) sum += x[(int64_t)s * stride];
  return sum;
}

template <bool WIN>
__global__ void __launch_bounds__(256) k_blend_fwd(int S, int64_t P, int C, const Policy policy, const float *renders,
                                                   const float *alphas, float *out, float *acc, int8_t *win,
                                                   const int64_t *__restrict__ n_isect, int64_t *__restrict__ counts_pinned) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // D4gsFrameIO.counts_pinned: the frame's last kernel also reports the list sizes to pinned host memory (every earlier kernel
  // of the stream - the composite that accumulates n_isect[2..3] included - has completed): k_copy_counts' store, minus its launch
  if (counts_pinned && i < 4) __hip_atomic_store(counts_pinned + i, n_isect[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int64_t PC = P * C;
  if (i < PC) {
    const int c = (int)(i % C);
    const float sum = d4gs_sum_strided(renders + i, S, PC);
    const float mean = (S == 1) ? renders[i] : sum / (float)S;
    float v = mean;
    const int pol = policy.p[c];
    // (best, w): the extreme raw value and the FIRST sub-sample that attains it - if the blended value equals it, that is the
    // sub-sample k_blend_bwd's first-match rule hands the gradient to
    float best = 0.f;
    int w = -1;
    if (pol != 0) {  // (eight raw values in flight at a time, then the same comparisons in the same order)
      for (int s0 = 0; s0 + 1 < S; s0 += 8) {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = (s0 + u + 1 < S) ? renders[(int64_t)(s0 + u) * PC + i] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (s0 + u + 1 >= S) break;
          if (pol == 1) {
            v = fmaxf(v, r[u]);
            if (WIN && (w < 0 || r[u] > best)) best = r[u], w = s0 + u;
          } else {
            v = fminf(v, r[u]);
            if (WIN && (w < 0 || r[u] < best)) best = r[u], w = s0 + u;
          }
        }
      }
    }
    out[i] = v;
    if (WIN && pol != 0) win[i] = (int8_t)((w >= 0 && best == v) ? w : -1);
  }
  if (i < P) {
    const float sum = d4gs_sum_strided(alphas + i, S, P);
    acc[i] = (S == 1) ? alphas[i] : sum / (float)S;
  }
}

__global__ void __launch_bounds__(256) k_blend_bwd(int S, int64_t P, int C, const Policy policy, const float *renders,
                                                   const float *out, const float *v_out, const float *v_acc,
                                                   float *v_renders, float *v_alphas, const float *add_r, const float *add_a,
                                                   const int8_t *__restrict__ win /* k_blend_fwd's winner map, or NULL: searched here */) {
  // add_r / add_a (optional): gradients the caller holds on the per-sub-sample images themselves (losses on `exposure_imgs`,
  // flow3d/trainer.py:599-618) - summed in here, so the composite backward sees one gradient per sub-sample
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t PC = P * C;
  if (i < PC) {
    const int c = (int)(i % C);
    const int pol = policy.p[c];
    const float g = v_out ? v_out[i] : 0.f;
    const float inv = 1.f / (float)S;
    int winner = -1;  // -1: the mean receives the gradient
    if (pol != 0 && S > 1) {
      if (win) {
        winner = win[i];  // (the one-call path: the forward left it - no reads of the S - 1 renders here)
      } else {
        const float o = out[i];
        for (int s0 = 0; s0 + 1 < S && winner < 0; s0 += 8) {  // first match, eight candidates in flight at a time
          float r[8];
#pragma unroll
          for (int u = 0; u < 8; u++) r[u] = (s0 + u + 1 < S) ? renders[(int64_t)(s0 + u) * PC + i] : 0.f;
#pragma unroll
          for (int u = 7; u >= 0; u--)
            if (s0 + u + 1 < S && r[u] == o) winner = s0 + u;  // (descending u: the lowest matching index stays)
        }
      }
    }
    if (add_r) {
      for (int s = 0; s < S; s++) {
        const float v = (winner < 0) ? g * inv : (s == winner ? g : 0.f);
        v_renders[s * PC + i] = v + add_r[s * PC + i];
      }
    } else {
      for (int s = 0; s < S; s++) v_renders[s * PC + i] = (winner < 0) ? g * inv : (s == winner ? g : 0.f);
    }
  }
  if (i < P) {
    const float g = v_acc ? v_acc[i] / (float)S : 0.f;
    for (int s = 0; s < S; s++) v_alphas[s * P + i] = add_a ? g + add_a[s * P + i] : g;
  }
}


// ---- exposure-sharded blend (one process per GPU, rank r holds the sub-samples s_first + j * s_stride) ---------------
// forward : k_shard_part_fwd -> [SUM all-reduce of part, MAX all-reduce of cand] -> k_shard_fin_fwd
// backward: k_shard_win -> [MIN all-reduce of win] -> k_shard_bwd
// Same result as k_blend_fwd/bwd on the full stack up to the order of the S-term sum (rank-local partial sums first).
struct ShardArgs {
  int S, Sl, s_first, s_stride, C, npol;
  int64_t P;
  Policy policy;
  int8_t pol_ch[64];  // the npol policy channels
};

__global__ void __launch_bounds__(256) k_shard_part_fwd(const ShardArgs a, const float *renders, const float *alphas,
                                                        float *part /* [P][C+1] */, float *cand /* [P][npol] */) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C1 = a.C + 1;
  if (i < a.P * C1) {
    const int64_t px = i / C1;
    const int c = (int)(i - px * C1);
    float sum = 0.f;
    if (c < a.C) {
      for (int j = 0; j < a.Sl; j++) sum += renders[((int64_t)j * a.P + px) * a.C + c];
    } else {
      for (int j = 0; j < a.Sl; j++) sum += alphas[(int64_t)j * a.P + px];
    }
    part[i] = sum;
  }
  if (i < a.P * a.npol) {
    const int64_t px = i / a.npol;
    const int c = a.pol_ch[i - px * a.npol];
    const float sign = a.policy.p[c] == 1 ? 1.f : -1.f;
    float best = -INFINITY;  // candidates: raw_s of the owned s <= S - 2 (the reference's last slot holds the mean)
    for (int j = 0; j < a.Sl; j++)
      if (a.s_first + j * a.s_stride <= a.S - 2) best = fmaxf(best, sign * renders[((int64_t)j * a.P + px) * a.C + c]);
    cand[i] = best;
  }
}

__global__ void __launch_bounds__(256) k_shard_fin_fwd(const ShardArgs a, const float *part, const float *cand, float *out,
                                                       float *acc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C1 = a.C + 1;
  if (i >= a.P * C1) return;
  const int64_t px = i / C1;
  const int c = (int)(i - px * C1);
  const float mean = a.S > 1 ? part[i] / (float)a.S : part[i];
  if (c == a.C) {
    acc[px] = mean;
    return;
  }
  float v = mean;
  const int pol = a.policy.p[c];
  if (pol != 0 && a.S > 1) {
    int slot = 0;
    while (a.pol_ch[slot] != c) slot++;
    const float sign = pol == 1 ? 1.f : -1.f;
    v = sign * fmaxf(cand[px * a.npol + slot], sign * mean);
  }
  out[px * a.C + c] = v;
}

// lowest owned s <= S - 2 whose raw value equals the blended value (S = none here: another rank's, or the mean)
__global__ void __launch_bounds__(256) k_shard_win(const ShardArgs a, const float *renders, const float *out, int32_t *win) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.P * a.npol) return;
  const int64_t px = i / a.npol;
  const int c = a.pol_ch[i - px * a.npol];
  const float o = out[px * a.C + c];
  int w = a.S;
  for (int j = a.Sl - 1; j >= 0; j--) {
    const int s = a.s_first + j * a.s_stride;
    if (s <= a.S - 2 && renders[((int64_t)j * a.P + px) * a.C + c] == o) w = s;
  }
  win[i] = w;
}

__global__ void __launch_bounds__(256) k_shard_bwd(const ShardArgs a, const float *v_out, const float *v_acc, const int32_t *win,
                                                   float *v_renders, float *v_alphas) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C1 = a.C + 1;
  if (i >= a.P * C1) return;
  const int64_t px = i / C1;
  const int c = (int)(i - px * C1);
  const float inv = a.S > 1 ? 1.f / (float)a.S : 1.f;
  if (c == a.C) {
    const float g = v_acc ? (a.S > 1 ? v_acc[px] / (float)a.S : v_acc[px]) : 0.f;  // (a division, as in k_blend_bwd)
    for (int j = 0; j < a.Sl; j++) v_alphas[(int64_t)j * a.P + px] = g;
    return;
  }
  const float g = v_out[px * a.C + c];
  int w = -1;  // -1: the mean receives the gradient
  if (a.policy.p[c] != 0 && a.S > 1) {
    int slot = 0;
    while (a.pol_ch[slot] != c) slot++;
    const int ww = win[px * a.npol + slot];
    if (ww < a.S) w = ww;
  }
  for (int j = 0; j < a.Sl; j++) {
    const int s = a.s_first + j * a.s_stride;
    v_renders[((int64_t)j * a.P + px) * a.C + c] = w < 0 ? g * inv : (s == w ? g : 0.f);
  }
}

}  // namespace

int d4gs_blend_fwd_impl(int32_t S, int64_t P, int32_t C, const int32_t *policy, const float *renders,
                        const float *alphas, float *out, float *acc, int8_t *win, hipStream_t stream,
                        const int64_t *n_isect, int64_t *counts_pinned) {
  const int64_t n = P * C;
  if (!n_isect) counts_pinned = nullptr;
  Policy pol;
  if (C > 64 || C <= 0) {
    d4gs_set_error("blend: C=%d out of range (1..64)", C);
    return D4GS_EINVAL;
  }
  for (int c = 0; c < 64; c++) pol.p[c] = (c < C && policy) ? (int8_t)policy[c] : 0;
  if (win)
    D4GS_LAUNCH("k_blend_fwd", k_blend_fwd<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, P, C, pol, renders, alphas, out, acc, win, n_isect, counts_pinned);
  else
    D4GS_LAUNCH("k_blend_fwd", k_blend_fwd<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, P, C, pol, renders, alphas, out, acc, win, n_isect, counts_pinned);
  return d4gs_check_launch("k_blend_fwd");
}

int d4gs_blend_bwd_add_impl(int32_t S, int64_t P, int32_t C, const int32_t *policy, const float *renders, const float *out,
                            const float *v_out, const float *v_acc, float *v_renders, float *v_alphas, const float *add_r,
                            const float *add_a, hipStream_t stream, const int8_t *win = nullptr);
int d4gs_blend_bwd_impl(int32_t S, int64_t P, int32_t C, const int32_t *policy, const float *renders, const float *out,
                        const float *v_out, const float *v_acc, float *v_renders, float *v_alphas, hipStream_t stream) {
  return d4gs_blend_bwd_add_impl(S, P, C, policy, renders, out, v_out, v_acc, v_renders, v_alphas, nullptr, nullptr, stream);
}

int d4gs_blend_bwd_add_impl(int32_t S, int64_t P, int32_t C, const int32_t *policy, const float *renders, const float *out,
                            const float *v_out, const float *v_acc, float *v_renders, float *v_alphas, const float *add_r,
                            const float *add_a, hipStream_t stream, const int8_t *win) {
  const int64_t n = P * C;
  Policy pol;
  if (C > 64 || C <= 0) {
    d4gs_set_error("blend: C=%d out of range (1..64)", C);
    return D4GS_EINVAL;
  }
  for (int c = 0; c < 64; c++) pol.p[c] = (c < C && policy) ? (int8_t)policy[c] : 0;
  D4GS_LAUNCH("k_blend_bwd", k_blend_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, P, C, pol, renders,
                     out, v_out, v_acc, v_renders, v_alphas, add_r, add_a, win);
  return d4gs_check_launch("k_blend_bwd");
}

static int shard_args(const D4gsShardBlend *b, ShardArgs &a) {
  if (!b || b->C <= 0 || b->C > 64 || b->S_total <= 0 || b->S_local < 0 || b->n_pixels <= 0 || b->s_stride <= 0) {
    d4gs_set_error("blend_shard: bad descriptor");
    return D4GS_EINVAL;
  }
  a.S = b->S_total, a.Sl = b->S_local, a.s_first = b->s_first, a.s_stride = b->s_stride, a.C = b->C, a.P = b->n_pixels;
  a.npol = 0;
  for (int c = 0; c < 64; c++) {
    a.policy.p[c] = (c < b->C && b->policy) ? (int8_t)b->policy[c] : 0;
    a.pol_ch[c] = 0;
  }
  for (int c = 0; c < b->C; c++)
    if (a.policy.p[c] != 0) a.pol_ch[a.npol++] = (int8_t)c;
  return D4GS_OK;
}

int d4gs_blend_shard_impl(int what, const D4gsShardBlend *b, const void *p0, const void *p1, const void *p2, void *o0, void *o1,
                          hipStream_t stream) {
  ShardArgs a;
  int rc = shard_args(b, a);
  if (rc) return rc;
  const int64_t n1 = a.P * (a.C + 1), np = a.P * (a.npol > 0 ? a.npol : 1);
  const unsigned g1 = (unsigned)((n1 + 255) / 256), gp = (unsigned)((np + 255) / 256);
  switch (what) {
    case 0:
      D4GS_LAUNCH("k_shard_part_fwd", k_shard_part_fwd, dim3(g1 > gp ? g1 : gp), dim3(256), 0, stream, a, (const float *)p0,
                  (const float *)p1, (float *)o0, (float *)o1);
      break;
    case 1:
      D4GS_LAUNCH("k_shard_fin_fwd", k_shard_fin_fwd, dim3(g1), dim3(256), 0, stream, a, (const float *)p0, (const float *)p1,
                  (float *)o0, (float *)o1);
      break;
    case 2:
      if (a.npol > 0)
        D4GS_LAUNCH("k_shard_win", k_shard_win, dim3(gp), dim3(256), 0, stream, a, (const float *)p0, (const float *)p1, (int32_t *)o0);
      break;
    default:
      D4GS_LAUNCH("k_shard_bwd", k_shard_bwd, dim3(g1), dim3(256), 0, stream, a, (const float *)p0, (const float *)p1,
                  (const int32_t *)p2, (float *)o0, (float *)o1);
  }
  return d4gs_check_launch("k_shard_blend");
}
