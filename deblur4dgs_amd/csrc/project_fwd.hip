// project_fwd.hip -- fused activations + motion-basis deformation + camera delta + perspective projection
// + tile counting for ALL exposure sub-samples of one blurry frame, and the two exclusive scans that
// lay out the intersection lists.
//
// Replaces, per render call (reference flow3d/scene_model.py:323-373, S iterations of ~60 torch launches):
//   params.py:39-43 (activations), params.py:142-180 (compute_transforms), transforms.py:41-53,
//   scene_model.py:89-102 (pose compose; done on rotation MATRICES),
//   scene_model.py:352-353 (camera delta), gsplat fully_fused_projection_fwd, gsplat isect_tiles pass 1.
//
// Work decomposition: one lane per Gaussian g, looping over the S sub-samples, so leaf parameters are read
// from HBM once and reused S times (coalesced SoA-by-tensor loads: consecutive lanes read consecutive rows).
// Per-block LDS: the S*K*9 time-blended bases (broadcast reads) and the block's softmaxed coefficients.
#include <stdlib.h>

#include "common.h"

namespace {

struct FwdArgs {
  D4gsDims d;
  D4gsProjIn in;
  D4gsProjOut out;
  int tw, th;
  int count_apart;  // 1: k_count_tiles does the tile counting / ranking (LDS-aggregated), 0: this kernel's global atomics
  int use_table;    // 1: out.blend_bases was filled by k_bases_table in front of this kernel - the blended bases are read from it with
                    //    scalar loads (no LDS slab); 0: every block blends its own LDS slab, block 0 also publishes the table for the backward
};

// time-blended bases: Bs[s][k][0:3] = transl, [3:9] = 6-D rotation   (params.py:152-177; w uses the clamped floor)
__device__ __forceinline__ void preblend_bases(const FwdArgs &a, float *Bs) {
#pragma clang fp contract(off)  // two products, one sum: torch's lerp and k_bases_table's, bit for bit
  const int K = a.d.K, T = a.d.T;
  for (int idx = threadIdx.x; idx < a.d.S * K * 9; idx += blockDim.x) {
    int s = idx / (K * 9), r = idx - s * K * 9, k = r / 9, j = r - k * 9;
    float t = a.in.times[s];
    float ff = fminf(fmaxf(floorf(t), 0.f), (float)(T - 1));
    float cf = fminf(fmaxf(ceilf(t), 0.f), (float)(T - 1));
    float w = t - ff;
    int f = (int)ff, c = (int)cf;
    float vf, vc;
    if (j < 3) {
      vf = a.in.transls[(k * T + f) * 3 + j];
      vc = a.in.transls[(k * T + c) * 3 + j];
    } else {
      vf = a.in.rots[(k * T + f) * 6 + j - 3];
      vc = a.in.rots[(k * T + c) * 6 + j - 3];
    }
    Bs[idx] = (1.f - w) * vf + w * vc;
  }
}

template <bool XT /* D4GS_EXACT_TILES: the wave-cooperative per-tile ellipse test (its own instantiation: 88 instead of 64 VGPRs) */,
          bool TAB /* FwdArgs.use_table: blended bases from the global table (scalar loads) instead of the block's LDS slab */>
__global__ void __launch_bounds__(D4GS_PROJ_BLOCK)
#ifdef XT_WAVES  // (A/B) the exact-tiles instantiations at XT_WAVES waves per SIMD (6: 80 VGPRs + a 16-byte scratch frame; default: 83-85, 5 waves)
__attribute__((amdgpu_waves_per_eu(XT ? XT_WAVES : 1, XT ? XT_WAVES : 10)))
#endif
k_project_fwd(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const D4gsDims &d = a.d;
  const int N = d.N, G = d.G, K = d.K, S = d.S;
  float *Bs = smem;                                  // [S][K][9]
  float *cf = smem + ((S * K * 9 + 3) & ~3);         // [K][BLOCK]
  const int tid = threadIdx.x;
  const int g = blockIdx.x * D4GS_PROJ_BLOCK + tid;
  const Cam cam = load_cam(a.in.viewmat, a.in.Kmat, d.width, d.height);
  const bool dyn_block = (blockIdx.x * D4GS_PROJ_BLOCK) < G;
  // Round 6: gridDim.y sub-sample GROUPS.  A lane loops over S sub-samples, so the launch has only N / 64 waves - cfg2 4 688, the reference's
  // training shape 2 188 for the chip's 8 192 wave slots (2.2 / 1.1 resident waves per SIMD: latency nobody hides).  Group y takes the
  // iterations [y, y + 1) x ceil(S / gridDim.y) of the same loop; everything per Gaussian is recomputed per group and STORED by group 0 only.
  const bool grp0 = blockIdx.y == 0;
  if (a.count_apart && a.out.tile_counts && grp0) {  // k_count_tiles (next in the stream) accumulates into a zeroed histogram: no memset node
    const int n2 = 2 * S * a.tw * a.th;
    for (int z = g; z < n2; z += gridDim.x * D4GS_PROJ_BLOCK) a.out.tile_counts[z] = 0;
  }
  if ((d.flags & D4GS_LAZY_SORT) && a.out.lazy_ws && grp0) {  // the lazy-sort counters start from zero as well
    const int64_t nl = d4gs_lazy_ws_elems(S, a.tw * a.th);
    for (int64_t z = g; z < nl; z += (int64_t)gridDim.x * D4GS_PROJ_BLOCK) a.out.lazy_ws[z] = 0;
  }
  typedef const __attribute__((address_space(4))) float *cfloat_p;
  if (dyn_block && !TAB) preblend_bases(a, Bs);
  if (blockIdx.x == 0 && grp0 && dyn_block && !TAB && a.out.blend_bases) {  // the table k_project_bwd reads with scalar loads (include/d4gs.h)
    __syncthreads();
    for (int idx = tid; idx < S * K * 16; idx += D4GS_PROJ_BLOCK) {
      const int j = idx & 15, sk = idx >> 4;
      a.out.blend_bases[idx] = j < 9 ? Bs[sk * 9 + j] : 0.f;
    }
  }

  const bool active = g < N;
  const bool raw = d.flags & D4GS_RAW_PARAMS;
  float mu[3] = {0, 0, 0}, Rq[9], sc[3] = {1, 1, 1}, opac = 0.f;
  if (active) {
    mu[0] = a.in.means[g * 3];
    mu[1] = a.in.means[g * 3 + 1];
    mu[2] = a.in.means[g * 3 + 2];
    const float4 q = *reinterpret_cast<const float4 *>(a.in.quats + (size_t)g * 4);
    float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    quat_to_rotmat(q.x * inv, q.y * inv, q.z * inv, q.w * inv, Rq);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float v = a.in.scales[g * 3 + j];
      sc[j] = raw ? expf(v) : v;
    }
    opac = a.in.opacities[g];
    if (raw) opac = 1.f / (1.f + expf(-opac));
    if (grp0) a.out.opac_act[g] = opac;
    const int D = grp0 ? d.D : 0, DP = (D + 3) & ~3;  // (the colour table row: group 0)
    // the colour table row: the channels that may need the sigmoid one by one (as before: the exp expansion stays single), the rest
    // 16 bytes at a time - a dword per channel and lane is DP stores of 64 scattered 4-byte pieces per wave: with 16 channels
    // k_project_fwd took 110 us against 64 with 3 (round 5)
    const int nsig4 = (d.flags & D4GS_RAW_COLORS) ? min((d.n_sigmoid + 3) & ~3, DP) : 0;
    for (int ch = 0; ch < nsig4; ch++) {
      float v = 0.f;
      if (ch < D) {
        v = a.in.colors[(size_t)g * D + ch];
        if (ch < d.n_sigmoid) v = 1.f / (1.f + expf(-v));
      }
      a.out.ctab[(size_t)g * DP + ch] = v;
    }
    const bool vec_in = (D & 3) == 0 && (((uintptr_t)a.in.colors) & 15) == 0;
    for (int ch = nsig4; ch < DP; ch += 4) {
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vec_in) {
        v4 = *reinterpret_cast<const float4 *>(a.in.colors + (size_t)g * D + ch);
      } else {
        if (ch < D) v4.x = a.in.colors[(size_t)g * D + ch];
        if (ch + 1 < D) v4.y = a.in.colors[(size_t)g * D + ch + 1];
        if (ch + 2 < D) v4.z = a.in.colors[(size_t)g * D + ch + 2];
        if (ch + 3 < D) v4.w = a.in.colors[(size_t)g * D + ch + 3];
      }
      *reinterpret_cast<float4 *>(a.out.ctab + (size_t)g * DP + ch) = v4;
    }
  }
  if (active) {
    if (g < G) {  // softmax(motion_coefs)  params.py:43
      const float *mc = a.in.motion_coefs + (size_t)g * K;
      float m = -INFINITY;
      for (int k = 0; k < K; k++) m = fmaxf(m, mc[k]);
      float sum = 0.f;
      for (int k = 0; k < K; k++) {
        float e = expf(mc[k] - m);
        cf[k * D4GS_PROJ_BLOCK + tid] = e;
        sum += e;
      }
      float is = 1.f / sum;
      for (int k = 0; k < K; k++) cf[k * D4GS_PROJ_BLOCK + tid] *= is;
    }
  }
  if (dyn_block) __syncthreads();

  // each wave walks the sub-samples from a different start, so resident waves hit all S*tiles counters at once
  const int rot = __builtin_amdgcn_readfirstlane((blockIdx.x * (D4GS_PROJ_BLOCK / 64) + (tid >> 6)) % S);
  constexpr bool exact_tiles = XT;
  __shared__ int xt_pre[XT ? D4GS_PROJ_BLOCK : 1];        // exclusive prefix of the wave's candidate (instance, tile) pairs
  __shared__ float4 xt_rec[XT ? D4GS_PROJ_BLOCK : 1][2];  // per instance: centre, conic, tau, packed rectangle origin / width
  __shared__ uint32_t xt_mask[XT ? D4GS_PROJ_BLOCK : 1][2];
  const int per_grp = (S + (int)gridDim.y - 1) / (int)gridDim.y;
  const int it_end = min(S, ((int)blockIdx.y + 1) * per_grp);
  for (int it = (int)blockIdx.y * per_grp; it < it_end; it++) {
    if constexpr (!XT) {  // (the plain instantiation keeps its round-4 control flow - and its 64 VGPRs: idle lanes leave at once)
      if (!active) continue;
    }
    const int s = (it + rot) % S;
    // what the lane's instance leaves behind for the (wave-cooperative) exact tile test and the stores after it
    int cnt = 0, radius_out = 0;
    int2 rect = make_int2(0, 0);
    float m2x = 0.f, m2y = 0.f, dep = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
    if (active) {
    float mw[3], Rm[9];
    if (g < G) {
      float v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (TAB) {  // many bases: rows of the global table through the scalar cache (see d4gs_project_fwd_impl)
        for (int k = 0; k < K; k += 2) {  // two rows per iteration (half as many scalar-load waits), as in k_project_bwd
          const bool two = k + 1 < K;
          const float c0 = cf[k * D4GS_PROJ_BLOCK + tid], c1 = two ? cf[(k + 1) * D4GS_PROJ_BLOCK + tid] : 0.f;
          cfloat_p B0 = (cfloat_p)(uintptr_t)a.out.blend_bases + ((size_t)s * K + k) * 16;
          cfloat_p B1 = (cfloat_p)(uintptr_t)a.out.blend_bases + ((size_t)s * K + (two ? k + 1 : k)) * 16;
          float r0[9], r1[9];
#pragma unroll
          for (int j = 0; j < 9; j++) r0[j] = B0[j], r1[j] = B1[j];
#pragma unroll
          for (int j = 0; j < 9; j++) v9[j] += c0 * r0[j];
#pragma unroll
          for (int j = 0; j < 9; j++) v9[j] += c1 * r1[j];
        }
      } else {
        const float *B = Bs + s * K * 9;
        for (int k = 0; k < K; k++) {
          float c = cf[k * D4GS_PROJ_BLOCK + tid];
#pragma unroll
          for (int j = 0; j < 9; j++) v9[j] += c * B[k * 9 + j];
        }
      }
      GS6 gs;
      gram_schmidt(v9 + 3, gs);
      float Rd[9] = {gs.x[0], gs.y[0], gs.z[0], gs.x[1], gs.y[1], gs.z[1], gs.x[2], gs.y[2], gs.z[2]};
#pragma unroll
      for (int i = 0; i < 3; i++) mw[i] = Rd[i * 3] * mu[0] + Rd[i * 3 + 1] * mu[1] + Rd[i * 3 + 2] * mu[2] + v9[i];
      mat3_mul(Rd, Rq, Rm);
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) mw[i] = mu[i];
#pragma unroll
      for (int i = 0; i < 9; i++) Rm[i] = Rq[i];
    }
    if (a.in.RTs) {  // camera delta: means only (scene_model.py:352-353)
      const float *RT = a.in.RTs + s * 12;
      float t0 = RT[0] * mw[0] + RT[1] * mw[1] + RT[2] * mw[2] + RT[3];
      float t1 = RT[4] * mw[0] + RT[5] * mw[1] + RT[6] * mw[2] + RT[7];
      float t2 = RT[8] * mw[0] + RT[9] * mw[1] + RT[10] * mw[2] + RT[11];
      mw[0] = t0, mw[1] = t1, mw[2] = t2;
    }
    ProjOut p;
    project_instance(cam, mw, Rm, sc, d, p);
    radius_out = p.radius;
    if (p.radius > 0) {
      float idet = 1.f / p.det;
      ca = p.c * idet, cb = -p.b * idet, cc = p.a * idet;
      m2x = p.mx, m2y = p.my, dep = p.pc[2];
      g0 = make_float4(m2x, m2y, opac, dep);
      g1 = make_float4(ca, cb, cc, 0.f);
      int x0, y0, x1, y1;
      tile_rect(m2x, m2y, p.radius, a.tw, a.th, x0, y0, x1, y1);
      if (d.flags & D4GS_EXACT_CULL) tight_rect(m2x, m2y, opac, p.a, p.c, x0, y0, x1, y1);
      rect = make_int2(x0 | (x1 << 16), y0 | (y1 << 16));
      cnt = (x1 - x0) * (y1 - y0);
    }
    if constexpr (XT) {  // everything the mask cannot change is stored now: only (cnt, rect) stay live across the cooperative test
      const size_t i = (size_t)s * N + g;
      a.out.radii[i] = radius_out;
      *reinterpret_cast<float2 *>(a.out.means2d + i * 2) = make_float2(m2x, m2y);
      a.out.depths[i] = dep;
      a.out.conics[i * 3] = ca;
      a.out.conics[i * 3 + 1] = cb;
      a.out.conics[i * 3 + 2] = cc;
      float4 *gp = reinterpret_cast<float4 *>(a.out.geom + i * D4GS_GEOM_STRIDE);
      gp[0] = g0;
      gp[1] = g1;
    }
    }  // active
    // ---- D4GS_EXACT_TILES: which tiles of the tight rectangle does the alpha >= 1/255 ellipse reach?  The (instance, tile) pairs of
    // the wave's 64 instances are spread over its lanes (a lane-private loop over the rectangle diverges: rectangles of 1 ... 64 tiles
    // in one wave - the round-4 attempt inside k_count_tiles / k_emit paid 300 us for it on cfg5); rectangles of 2 ... 64 tiles, at
    // most 8 wide and 8 tall, get a mask; the rest keep their whole rectangle (mask 0).
    uint64_t mask = 0;
    if constexpr (exact_tiles) {
      const int x0 = rect.x & 0xffff, x1 = rect.x >> 16, y0 = rect.y & 0xffff, y1 = rect.y >> 16;
      const int w = x1 - x0, h = y1 - y0;
      // (a 1 x n rectangle cannot lose a tile: the ellipse touches both ends of its bounding box inside the single row / column)
      const bool cand = w >= 2 && h >= 2 && w <= 8 && h <= 8;
      const int np = cand ? cnt : 0;
      const int lane = tid & 63, wbase = tid & ~63;
      int inc = np;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      const int total = __shfl(inc, 63);
      if (total > 0) {  // (wave-uniform)
        xt_pre[tid] = inc - np;
        // conic of the BLURRED covariance = the geom record's; tau as tight_rect computes it
        const float tau = __logf(255.f * opac) * 1.01f + 0.02f;
        xt_rec[tid][0] = make_float4(m2x, m2y, ca, cb);
        xt_rec[tid][1] = make_float4(cc, tau, __int_as_float(x0 | (y0 << 16)), __int_as_float(w));
        xt_mask[tid][0] = 0u, xt_mask[tid][1] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (int pp = lane; pp < total; pp += 64) {
          int lo = 0;  // the last instance whose prefix is <= pp (instances without pairs share their successor's prefix)
#pragma unroll
          for (int step = 32; step > 0; step >>= 1)
            if (xt_pre[wbase + lo + step] <= pp) lo += step;
          const int q = pp - xt_pre[wbase + lo];
          const float4 r0 = xt_rec[wbase + lo][0], r1 = xt_rec[wbase + lo][1];
          const int ww = __float_as_int(r1.w), xy = __float_as_int(r1.z);
          const int ry = (int)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)ww));  // q / ww for q < 64, ww <= 8
          const int rx = q - ry * ww;
          const int tx = (xy & 0xffff) + rx, ty = (xy >> 16) + ry;
          const float X0 = (float)(tx * D4GS_TILE) + 0.5f, Y0 = (float)(ty * D4GS_TILE) + 0.5f;
          const float X1 = fminf(X0 + (float)(D4GS_TILE - 1), (float)d.width - 0.5f), Y1 = fminf(Y0 + (float)(D4GS_TILE - 1), (float)d.height - 0.5f);
          if (d4gs_ellipse_hits_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, X0, X1, Y0, Y1)) {
            const int bit = ry * 8 + rx;
            atomicOr(&xt_mask[wbase + lo][bit >> 5], 1u << (bit & 31));
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (cand) {
          mask = (uint64_t)xt_mask[tid][0] | ((uint64_t)xt_mask[tid][1] << 32);
          cnt = __popcll(mask);
          if (cnt == 0) rect = make_int2(0, 0);  // (the ellipse reaches no pixel centre of any tile: nothing to bin)
        }
        __builtin_amdgcn_wave_barrier();  // the arrays are rewritten by the next pass
      }
    }
    if (!active) continue;
    const size_t i = (size_t)s * N + g;
    if (!a.count_apart && cnt > 0) {  // tile grids too big for the LDS histogram: plain global atomics
      int *tc = a.out.tile_counts + (size_t)s * a.tw * a.th;
      const int x0 = rect.x & 0xffff, x1 = rect.x >> 16, y0 = rect.y & 0xffff, y1 = rect.y >> 16;
      if (mask) {
        for (uint64_t m = mask; m; m &= m - 1) {
          const int b = __ffsll((long long)m) - 1;
          atomicAdd(tc + (y0 + (b >> 3)) * a.tw + x0 + (b & 7), 1);
        }
      } else {
        for (int ty = y0; ty < y1; ty++)
          for (int tx = x0; tx < x1; tx++) atomicAdd(tc + ty * a.tw + tx, 1);
      }
    }
    a.out.tiles_touched[i] = cnt;
    *reinterpret_cast<int2 *>(a.out.tile_rects + i * 2) = rect;
    if constexpr (XT) {
      a.out.tile_masks[i] = mask;
    } else {
      a.out.radii[i] = radius_out;
      *reinterpret_cast<float2 *>(a.out.means2d + i * 2) = make_float2(m2x, m2y);
      a.out.depths[i] = dep;
      a.out.conics[i * 3] = ca;
      a.out.conics[i * 3 + 1] = cb;
      a.out.conics[i * 3 + 2] = cc;
      float4 *gp = reinterpret_cast<float4 *>(a.out.geom + i * D4GS_GEOM_STRIDE);
      gp[0] = g0;
      gp[1] = g1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// a4/a5 + a11: SceneModel.compute_transforms / compute_poses_fg / compute_poses_all (flow3d/scene_model.py:58-120) and the
// track-channel positions (scene_model.py:258-289) for B = S times.  A block owns 64 Gaussians, its 4 waves are 4 time
// slots (time b = slot, slot + 4, ...): the time is wave-uniform, its blended bases one small LDS slab per wave - any B.
// ---------------------------------------------------------------------------------------------------
constexpr int POSE_GPB = 64, POSE_SLOTS = 4;
struct PosesArgs {
  D4gsDims d;
  D4gsProjIn in;
  D4gsPoses out;
};
__global__ void __launch_bounds__(POSE_GPB * POSE_SLOTS) k_poses_fwd(const PosesArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const D4gsDims &d = a.d;
  const int N = d.N, G = d.G, S = d.S;
  const int K = G > 0 ? d.K : 0, KP = K | 1, nk = K * 9, nk4 = (nk + 3) & ~3;
  float *cf = smem;                    // [64][KP]
  float *bsl = cf + POSE_GPB * KP;     // [4][nk4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int slot = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x * POSE_GPB + lane;
  const bool dyn_block = (blockIdx.x * POSE_GPB) < G;
  const bool active = g < N, isdyn = active && g < G;
  const bool has_cam = a.in.viewmat != nullptr;
  Cam cam;
  if (has_cam) cam = load_cam(a.in.viewmat, a.in.Kmat, d.width, d.height);
  float mu[3] = {0, 0, 0}, qh[4] = {1, 0, 0, 0};
  if (active) mu[0] = a.in.means[g * 3], mu[1] = a.in.means[g * 3 + 1], mu[2] = a.in.means[g * 3 + 2];
  if (active && a.out.quats) {
    const float4 q = *reinterpret_cast<const float4 *>(a.in.quats + (size_t)g * 4);
    const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    qh[0] = q.x * inv, qh[1] = q.y * inv, qh[2] = q.z * inv, qh[3] = q.w * inv;
  }
  if (dyn_block && slot == 0) {
    for (int k = 0; k < K; k++) cf[lane * KP + k] = 0.f;
    if (isdyn && !(d.flags & D4GS_RAW_PARAMS)) {  // activated coefficients (MotionBases.compute_transforms' own signature)
      for (int k = 0; k < K; k++) cf[lane * KP + k] = a.in.motion_coefs[(size_t)g * K + k];
    } else if (isdyn) {  // softmax(motion_coefs)  params.py:43
      const float *mc = a.in.motion_coefs + (size_t)g * K;
      float m = -INFINITY;
      for (int k = 0; k < K; k++) m = fmaxf(m, mc[k]);
      float sum = 0.f;
      for (int k = 0; k < K; k++) {
        float e = expf(mc[k] - m);
        cf[lane * KP + k] = e;
        sum += e;
      }
      float is = 1.f / sum;
      for (int k = 0; k < K; k++) cf[lane * KP + k] *= is;
    }
  }
  __syncthreads();
  float *B = bsl + slot * nk4;
  for (int s = slot; s < S; s += POSE_SLOTS) {
    if (dyn_block) {
      const int T = d.T;
      const float t = a.in.times[s];
      const float ff = fminf(fmaxf(floorf(t), 0.f), (float)(T - 1));
      const float cfl = fminf(fmaxf(ceilf(t), 0.f), (float)(T - 1));
      const float w = t - ff;
      const int f = (int)ff, c = (int)cfl;
      for (int idx = lane; idx < nk; idx += 64) {
        const int k = idx / 9, j = idx - k * 9;
        float vf, vc;
        if (j < 3) {
          vf = a.in.transls[(k * T + f) * 3 + j];
          vc = a.in.transls[(k * T + c) * 3 + j];
        } else {
          vf = a.in.rots[(k * T + f) * 6 + j - 3];
          vc = a.in.rots[(k * T + c) * 6 + j - 3];
        }
        B[idx] = (1.f - w) * vf + w * vc;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (active) {
      const size_t i = a.out.g_major ? (size_t)g * S + s : (size_t)s * N + g;
      float mw[3] = {mu[0], mu[1], mu[2]}, qo[4] = {qh[0], qh[1], qh[2], qh[3]};
      if (isdyn) {
        float v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < K; k++) {
          const float c = cf[lane * KP + k];
#pragma unroll
          for (int j = 0; j < 9; j++) v9[j] += c * B[k * 9 + j];
        }
        GS6 gs;
        gram_schmidt(v9 + 3, gs);
        const float Rd[9] = {gs.x[0], gs.y[0], gs.z[0], gs.x[1], gs.y[1], gs.z[1], gs.x[2], gs.y[2], gs.z[2]};
#pragma unroll
        for (int r = 0; r < 3; r++) mw[r] = Rd[r * 3] * mu[0] + Rd[r * 3 + 1] * mu[1] + Rd[r * 3 + 2] * mu[2] + v9[r];
        if (a.out.transforms) {
          float *tp = a.out.transforms + (a.out.g_major ? (size_t)g * S + s : (size_t)s * G + g) * 12;
#pragma unroll
          for (int r = 0; r < 3; r++)
            *reinterpret_cast<float4 *>(tp + r * 4) = make_float4(Rd[r * 3], Rd[r * 3 + 1], Rd[r * 3 + 2], v9[r]);
        }
        if (a.out.quats) {
          PoseQ pq;
          pose_quat(Rd, qh, pq);
#pragma unroll
          for (int c = 0; c < 4; c++) qo[c] = pq.out[c];
        }
      }
      if (a.out.quats) *reinterpret_cast<float4 *>(a.out.quats + i * 4) = make_float4(qo[0], qo[1], qo[2], qo[3]);
      if (a.out.means) {
        if (a.in.RTs) {  // target camera / camera delta (means only)
          const float *RT = a.in.RTs + s * 12;
          const float t0 = RT[0] * mw[0] + RT[1] * mw[1] + RT[2] * mw[2] + RT[3];
          const float t1 = RT[4] * mw[0] + RT[5] * mw[1] + RT[6] * mw[2] + RT[7];
          const float t2 = RT[8] * mw[0] + RT[9] * mw[1] + RT[10] * mw[2] + RT[11];
          mw[0] = t0, mw[1] = t1, mw[2] = t2;
        }
        if (has_cam) {
          const float t0 = cam.R[0] * mw[0] + cam.R[1] * mw[1] + cam.R[2] * mw[2] + cam.t[0];
          const float t1 = cam.R[3] * mw[0] + cam.R[4] * mw[1] + cam.R[5] * mw[2] + cam.t[1];
          const float t2 = cam.R[6] * mw[0] + cam.R[7] * mw[1] + cam.R[8] * mw[2] + cam.t[2];
          mw[0] = t0, mw[1] = t1, mw[2] = t2;
        }
        a.out.means[i * 3] = mw[0], a.out.means[i * 3 + 1] = mw[1], a.out.means[i * 3 + 2] = mw[2];
      }
    }
    __builtin_amdgcn_wave_barrier();  // the slab is rewritten in the next pass
  }
}

// ---------------------------------------------------------------------------------------------------
// Tile counting, LDS-aggregated.  Device-scope atomics execute memory-side on this part (~27 G/s on spread
// addresses): one per (splat, tile) cost k_project_fwd 0.15 of its 0.23 ms.  Here a 1024-lane block takes 4096
// instances of ONE sub-sample, counts their tiles into an LDS histogram of that sub-sample's tile grid and adds
// every non-empty bin to the global counter with one atomic: ~13x fewer global atomics.  (k_emit later repeats the
// same histogram to hand out the slots, see binning.hip.)
// ---------------------------------------------------------------------------------------------------
constexpr int COUNT_THREADS = 1024;

// D4GS_LAZY_SORT: the depth range of every sub-sample's binned instances, the scale of d4gs_depth_bucket - from a SAMPLE (every
// 16th instance): the bucket map clamps outside the range, so any range gives a monotone, consistent partition; an exact one
// costs 36 - 116 us in same-address atomics and reads for nothing
__global__ void __launch_bounds__(256) k_depth_range(const float *__restrict__ depths, const int *__restrict__ tiles_touched, int N,
                                                     uint32_t *__restrict__ zr) {
  const int s = blockIdx.y;
  uint32_t lo = 0u, hi = 0u;  // lo holds ~bits: the maximum of ~bits is the minimum of bits
  for (int g = (blockIdx.x * 256 + threadIdx.x) * 16; g < N; g += gridDim.x * 256 * 16) {
    const size_t i = (size_t)s * N + g;
    if (tiles_touched[i] > 0) {
      const uint32_t b = __float_as_uint(depths[i]);
      lo = max(lo, ~b), hi = max(hi, b);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lo = max(lo, (uint32_t)__shfl_xor((int)lo, o)), hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
  __shared__ uint32_t slo[4], shi[4];
  if ((threadIdx.x & 63) == 0) slo[threadIdx.x >> 6] = lo, shi[threadIdx.x >> 6] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = max(max(slo[0], slo[1]), max(slo[2], slo[3])), hi = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
    if (hi) atomicMax(zr + 2 * s, lo), atomicMax(zr + 2 * s + 1, hi);
  }
}

// D4GS_LAZY_SORT: per tile, the near part = the nearest depth buckets up to (at least) `target` keys
__global__ void __launch_bounds__(256) k_lazy_pivot(LazyWs w, int n_tiles, int target) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tiles) return;
  int cum = 0, b = 0;
  for (; b < w.nb; b++) {
    cum += w.hist[(size_t)t * w.nb + b];
    if (cum >= target) break;
  }
  w.near[t] = cum;
  w.pivot[t] = b < w.nb ? b : w.nb - 1;
}

template <int COUNT_PER_THREAD, bool LAZY>
__global__ void __launch_bounds__(COUNT_THREADS) k_count_tiles(const int *__restrict__ tile_rects,
                                                              const int *__restrict__ tiles_touched, int N, int S, int tw,
                                                              int th, int *__restrict__ tile_counts, int *__restrict__ chunk_sums,
                                                              const float *__restrict__ depths, LazyWs lazy,
                                                              const uint64_t *__restrict__ tile_masks /* D4GS_EXACT_TILES or NULL */) {
  extern __shared__ int hist[];  // [tiles] (+ LAZY: [tiles][nb] per depth bucket)
  __shared__ int wsum[COUNT_THREADS / 64];
  const int tiles = tw * th, tid = threadIdx.x;
  const int s = blockIdx.x % S, chunk = blockIdx.x / S;  // neighbouring blocks work on different sub-samples' counters
  for (int z = tid; z < (LAZY ? tiles * (1 + lazy.nb) : tiles); z += COUNT_THREADS) hist[z] = 0;
  __syncthreads();
  int *hist2 = hist + tiles;
  const uint32_t zlo = LAZY ? ~lazy.zr[2 * s] : 0u, zhi = LAZY ? lazy.zr[2 * s + 1] : 0u;
  int touched = 0;  // this lane's share of the chunk's intersection count (fused scan: see d4gs_fused_scan)
  if constexpr (LAZY) {
    // (the lazy instantiation - cfg5: a 150 KB histogram, ONE resident workgroup per CU - keeps the instance-by-instance walk: with all
    // loads up front its sixteen waves load together and then hammer the LDS together, 199 -> 226 us)
  #pragma unroll
    for (int q = 0; q < COUNT_PER_THREAD; q++) {
      const int g = (chunk * COUNT_PER_THREAD + q) * COUNT_THREADS + tid;
      if (g >= N) continue;
      const size_t i = (size_t)s * N + g;
      const int tt = tiles_touched[i];
      touched += tt;
      if (tt == 0) continue;
      const int2 rc = *reinterpret_cast<const int2 *>(tile_rects + i * 2);
      const int x0 = rc.x & 0xffff, x1 = rc.x >> 16, y0 = rc.y & 0xffff, y1 = rc.y >> 16;
      const int bk = LAZY ? d4gs_depth_bucket(depths[i], zlo, zhi, lazy.nb) : 0;
      const uint64_t mask = tile_masks ? tile_masks[i] : 0;
      if (mask) {  // D4GS_EXACT_TILES: the tiles the ellipse reaches, bit (ty - y0) * 8 + (tx - x0)
        for (uint64_t m = mask; m; m &= m - 1) {
          const int b = __ffsll((long long)m) - 1, t = (y0 + (b >> 3)) * tw + x0 + (b & 7);
          atomicAdd(&hist[t], 1);
          if (LAZY) atomicAdd(&hist2[t * lazy.nb + bk], 1);
        }
        continue;
      }
      for (int ty = y0; ty < y1; ty++)
        for (int tx = x0; tx < x1; tx++) {
          atomicAdd(&hist[ty * tw + tx], 1);
          if (LAZY) atomicAdd(&hist2[(ty * tw + tx) * lazy.nb + bk], 1);
        }
    }
  } else {
    // (all of the lane's loads first, waited for once - k_emit's note in binning.hip: instance by instance this was 2 x COUNT_PER_THREAD
    // dependent round trips to memory)
    int tts[COUNT_PER_THREAD], rxs[COUNT_PER_THREAD], rys[COUNT_PER_THREAD];
    float deps[COUNT_PER_THREAD];
    uint64_t msks[COUNT_PER_THREAD];
  #pragma unroll
    for (int q = 0; q < COUNT_PER_THREAD; q++) {
      const int g = (chunk * COUNT_PER_THREAD + q) * COUNT_THREADS + tid;
      tts[q] = 0, rxs[q] = 0, rys[q] = 0, deps[q] = 0.f, msks[q] = 0;
      if (g < N) {
        const size_t i = (size_t)s * N + g;
        tts[q] = tiles_touched[i];
        const int2 rc = *reinterpret_cast<const int2 *>(tile_rects + i * 2);
        rxs[q] = rc.x, rys[q] = rc.y;
        if (LAZY) deps[q] = depths[i];
        if (tile_masks) msks[q] = tile_masks[i];
      }
    }
  #pragma unroll
    for (int q = 0; q < COUNT_PER_THREAD; q++) {
      uint32_t ml = (uint32_t)msks[q], mh = (uint32_t)(msks[q] >> 32);
      asm volatile("" : "+v"(tts[q]), "+v"(rxs[q]), "+v"(rys[q]), "+v"(deps[q]), "+v"(ml), "+v"(mh));
      msks[q] = ((uint64_t)mh << 32) | ml;
    }
  #pragma unroll
    for (int q = 0; q < COUNT_PER_THREAD; q++) {
      const int tt = tts[q];
      touched += tt;
      if (tt == 0) continue;
      const int x0 = rxs[q] & 0xffff, x1 = rxs[q] >> 16, y0 = rys[q] & 0xffff, y1 = rys[q] >> 16;
      const int bk = LAZY ? d4gs_depth_bucket(deps[q], zlo, zhi, lazy.nb) : 0;
      const uint64_t mask = msks[q];
      if (mask) {  // D4GS_EXACT_TILES: the tiles the ellipse reaches, bit (ty - y0) * 8 + (tx - x0)
        for (uint64_t m = mask; m; m &= m - 1) {
          const int b = __ffsll((long long)m) - 1, t = (y0 + (b >> 3)) * tw + x0 + (b & 7);
          atomicAdd(&hist[t], 1);
          if (LAZY) atomicAdd(&hist2[t * lazy.nb + bk], 1);
        }
        continue;
      }
      for (int ty = y0; ty < y1; ty++)
        for (int tx = x0; tx < x1; tx++) {
          atomicAdd(&hist[ty * tw + tx], 1);
          if (LAZY) atomicAdd(&hist2[(ty * tw + tx) * lazy.nb + bk], 1);
        }
    }
  }
  if (chunk_sums) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) touched += __shfl_xor(touched, o);
    if ((tid & 63) == 0) wsum[tid >> 6] = touched;
  }
  __syncthreads();
  for (int z = tid; z < tiles; z += COUNT_THREADS) {
    const int c = hist[z];
    if (c > 0) atomicAdd(tile_counts + (size_t)s * tiles + z, c);
  }
  if (LAZY)
    for (int z = tid; z < tiles * lazy.nb; z += COUNT_THREADS) {
      const int c = hist2[z];
      if (c > 0) atomicAdd(lazy.hist + (size_t)s * tiles * lazy.nb + z, c);
    }
  if (chunk_sums && tid == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < COUNT_THREADS / 64; w++) t += wsum[w];
    chunk_sums[(size_t)s * gridDim.x / S + chunk] = t;  // (sub-sample, chunk) order = the flat instance order
  }
}

// ---------------------------------------------------------------------------------------------------
// exclusive scan (int32) : block sums -> one block scans the sums -> apply.  2048 items per block.
// ---------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int block_exclusive_scan(int v, int *total, int *lds /* [SCAN_THREADS/64 + 1] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
    int x = lds[w];
    if (w < wave) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_sums(const int *in, int64_t n, int *sums) {
  __shared__ int lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++)
    if (base + j < n) s += in[base + j];
  int tot;
  block_exclusive_scan(s, &tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of `n` ints in place (looping with a carry); optionally writes the total to
// out[n] (int32) and to total64.
struct ScanJob {
  const int *in, *in2;
  int *out;
  int64_t n;
  int write_last;
  int64_t *total64, *max64;
};
// gridDim.x = 2: block 0 runs job A, block 1 job B - the two single-block scans of the binning chain (block sums of the
// instance scan; tile counts -> tile offsets) do not depend on each other, so they share one launch.
__global__ void __launch_bounds__(1024) k_scan_single(const ScanJob ja, const ScanJob jb) {
  const ScanJob &job = blockIdx.x ? jb : ja;
  const int *in = job.in, *in2 = job.in2;
  int *out = job.out;
  const int64_t n = job.n;
  const int write_last = job.write_last;
  int64_t *total64 = job.total64, *max64 = job.max64;
  __shared__ int lds[20];
  __shared__ int smax;
  if (threadIdx.x == 0) smax = 0;
  __syncthreads();
  int64_t carry = 0;
  int vmax = 0;
  // eight consecutive items per thread and round.  (This kernel's time - 12 / 41 / 84 us on cfg2 / cfg3 / cfg5 - is NOT its rounds: one
  // item per thread, 57 rounds on cfg5, measured the same.  It grows with the tile count because its first loads wait for k_count_tiles'
  // device-scope atomics on the same counters to drain: the backlog of the kernel in front is charged here.)
  constexpr int SI = 8;
  for (int64_t base = 0; base < n; base += (int64_t)blockDim.x * SI) {
    const int64_t i0 = base + (int64_t)threadIdx.x * SI;
    int v[SI], sum = 0;
#pragma unroll
    for (int j = 0; j < SI; j++) {
      v[j] = i0 + j < n ? in[i0 + j] + (in2 ? in2[i0 + j] : 0) : 0;
      vmax = max(vmax, v[j]);
      sum += v[j];
    }
    int tot;
    int64_t run = carry + block_exclusive_scan(sum, &tot, lds);
#pragma unroll
    for (int j = 0; j < SI; j++) {
      if (i0 + j < n) out[i0 + j] = (int)run;
      run += v[j];
    }
    carry += tot;
  }
  if (max64) {
    atomicMax(&smax, vmax);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (write_last) out[n] = (int)carry;
    if (total64) total64[0] = carry, total64[2] = 0, total64[3] = 0;  // ([2], [3]: the forward composite's live-row sample)
    if (max64) *max64 = smax;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(const int *in, const int *sums_ex, int64_t n, int *out) {
  __shared__ int lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    s += v[j];
  }
  int tot;
  int ex = block_exclusive_scan(s, &tot, lds) + sums_ex[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

}  // namespace

extern "C" size_t d4gs_scan_ws_elems(int64_t n_instances) {  // (block sums of the two-pass scan / chunk sums of the fused one)
  return (size_t)((n_instances + COUNT_THREADS - 1) / COUNT_THREADS) + 64;
}

// D4GS_LAZY_SORT in effect?  (needs the LDS-aggregated counting pass and the scratch buffer; D4GS_LAZY=0 forces it off)
bool d4gs_lazy_on(const D4gsDims *d, const D4gsProjOut *out) {
  if (!(d->flags & D4GS_LAZY_SORT) || !out->lazy_ws || d->N <= 0) return false;
  const char *env = getenv("D4GS_LAZY");
  if (env && env[0] == '0') return false;
  const size_t tiles = (size_t)((d->width + D4GS_TILE - 1) / D4GS_TILE) * ((d->height + D4GS_TILE - 1) / D4GS_TILE);
  return sizeof(int) * tiles * (1 + d4gs_lazy_buckets((int)tiles)) <= 150 * 1024 && sizeof(int) * tiles <= 64 * 1024 &&
         getenv("D4GS_COUNT_IN_PROJECT") == nullptr;
}
int d4gs_lazy_pivot_launch(const D4gsDims *d, const D4gsProjOut *out, int64_t near_target, hipStream_t stream) {
  const int tiles = ((d->width + D4GS_TILE - 1) / D4GS_TILE) * ((d->height + D4GS_TILE - 1) / D4GS_TILE), nt = d->S * tiles;
  const int target = near_target > 0 ? (int)(near_target > (1 << 20) ? (1 << 20) : near_target) : 1024;
  D4GS_LAUNCH("k_lazy_pivot", k_lazy_pivot, dim3((nt + 255) / 256), dim3(256), 0, stream, d4gs_lazy_carve(out->lazy_ws, d->S, tiles), nt, target);
  return d4gs_check_launch("k_lazy_pivot");
}

// (256 = the CU count of the one device this library is built for - gfx950 / MI355X; the choice must not depend on a device query:
// d4gs_query_sizes answers on GPU-less hosts too, and k_count_tiles / k_emit only need to agree with each other)
int d4gs_chunk_per_thread(const D4gsDims *d) {
  const int64_t blocks4 = (((int64_t)d->N + 4 * COUNT_THREADS - 1) / (4 * COUNT_THREADS)) * d->S;
  // (round 5, measured and dropped: 8 / 16 instances per lane on 720p grids - 4x fewer per-bin device-scope atomics - made k_emit
  // SLOWER, cfg5 720 -> 1 139 us, cfg3 193 -> 245: its time is not those atomics; profiles/r05_ab_chunk16.txt)
  if (blocks4 < 256) return 1;
  // Round 6: the 1024-lane blocks of k_count_tiles / k_emit fit two per CU = 512 at a time.  cfg2's 592 blocks (74 chunks x 8 sub-samples)
  // are one full round and one of 80 blocks that costs as much again; five or six instances per lane make the same launch ONE round
  // (472 blocks).  Only when it saves a round - beyond two rounds the longer dependent chain per lane costs more than the tail.
  static const int pt_env = getenv("D4GS_CHUNK_PT") ? atoi(getenv("D4GS_CHUNK_PT")) : 0;  // A/B hook: 2 ... 6
  if (pt_env >= 2 && pt_env <= 6) return pt_env;
  const int64_t slots = 512;
  auto blocks_at = [&](int pt) { return (((int64_t)d->N + (int64_t)pt * COUNT_THREADS - 1) / ((int64_t)pt * COUNT_THREADS)) * d->S; };
  if (d->flags & D4GS_LAZY_SORT) return 4;  // (the lazy instantiations walk more per pair: 5 measured slower)
  // the FEWEST instances per lane that still make the launch one round: shorter dependent chains per lane, more blocks in flight
  // (the reference's training shape: 4 -> 3 per lane, 385 -> 506 blocks)
  for (int pt = 2; pt <= 6; pt++)
    if (blocks_at(pt) <= slots) return pt;
  return 4;
}

// Fused scan of the per-instance intersection counts (small tile grids, i.e. every BASELINE config): k_count_tiles also
// leaves the sum of each (sub-sample, 4096-instance chunk), one single-block scan turns the sums into chunk bases, and
// k_emit - whose blocks are the same chunks - scans inside its chunk and writes isect_offsets itself: no k_scan_sums /
// k_scan_apply launches.  Same flat order, same offsets.  Both d4gs_project_fwd and d4gs_bin_sort ask this predicate.
int d4gs_fused_scan_chunks(const D4gsDims *d) {
  static const bool force_in_kernel = getenv("D4GS_COUNT_IN_PROJECT") != nullptr;
  static const bool off = getenv("D4GS_NO_FUSED_SCAN") != nullptr;  // A/B hook
  const int tw = (d->width + D4GS_TILE - 1) / D4GS_TILE, th = (d->height + D4GS_TILE - 1) / D4GS_TILE;
  if (off || force_in_kernel || d->N <= 0 || sizeof(int) * (size_t)tw * th > 64 * 1024) return 0;
  const int per_block = COUNT_THREADS * d4gs_chunk_per_thread(d);
  const int64_t nchunks = (d->N + per_block - 1) / per_block;
  if ((size_t)(nchunks * d->S) + 1 > d4gs_scan_ws_elems((int64_t)d->S * d->N)) return 0;  // tiny N, many sub-samples
  return (int)nchunks;
}

static size_t fwd_lds_bytes(const D4gsDims *dims) {
  if (dims->G <= 0) return 0;
  return sizeof(float) * (((size_t)dims->S * dims->K * 9 + 3) & ~(size_t)3) + sizeof(float) * (size_t)dims->K * D4GS_PROJ_BLOCK;
}

int d4gs_poses_fwd_impl(const D4gsDims *dims, const D4gsProjIn *in, const D4gsPoses *out, hipStream_t stream) {
  PosesArgs a;
  a.d = *dims;
  a.in = *in;
  a.out = *out;
  const int K = dims->G > 0 ? dims->K : 0;
  const size_t lds = sizeof(float) * ((size_t)POSE_GPB * (K | 1) + POSE_SLOTS * (((size_t)K * 9 + 3) & ~(size_t)3));
  const int blocks = (dims->N + POSE_GPB - 1) / POSE_GPB;
  D4GS_LAUNCH("k_poses_fwd", k_poses_fwd, dim3(blocks), dim3(POSE_GPB * POSE_SLOTS), lds, stream, a);
  return d4gs_check_launch("k_poses_fwd");
}

int d4gs_project_fwd_impl(const D4gsDims *dims, const D4gsProjIn *in, const D4gsProjOut *out, hipStream_t stream) {
  FwdArgs a;
  a.d = *dims;
  a.in = *in;
  a.out = *out;
  a.tw = (dims->width + D4GS_TILE - 1) / D4GS_TILE;
  a.th = (dims->height + D4GS_TILE - 1) / D4GS_TILE;
  const int64_t n_inst = (int64_t)dims->S * dims->N;
  const int64_t n_tiles = (int64_t)dims->S * a.tw * a.th;
  size_t lds = 0;
  if (dims->G > 0) lds = sizeof(float) * (((size_t)dims->S * dims->K * 9 + 3) & ~(size_t)3) +
                         sizeof(float) * (size_t)dims->K * D4GS_PROJ_BLOCK;
  if (lds > 64 * 1024) {
    d4gs_set_error("S*K too large for the LDS-resident bases (S=%d K=%d)", dims->S, dims->K);
    return D4GS_EINVAL;
  }
  if ((dims->flags & D4GS_EXACT_TILES) && (!out->tile_masks || !(dims->flags & D4GS_EXACT_CULL))) {
    d4gs_set_error("D4GS_EXACT_TILES needs D4gsProjOut.tile_masks and D4GS_EXACT_CULL (the masks refine the tight rectangle)");
    return D4GS_EINVAL;
  }
  const int blocks = (dims->N + D4GS_PROJ_BLOCK - 1) / D4GS_PROJ_BLOCK;
  const size_t hist_bytes = sizeof(int) * (size_t)a.tw * a.th;
  static const bool force_in_kernel = getenv("D4GS_COUNT_IN_PROJECT") != nullptr;  // test hook for the fallback
  a.count_apart = hist_bytes <= 64 * 1024 && dims->N > 0 && !force_in_kernel;  // bigger tile grids: in-kernel atomics
  const int fused_chunks = a.count_apart ? d4gs_fused_scan_chunks(dims) : 0;
  if (!a.count_apart) {  // (count_apart: k_project_fwd zeroes the histogram itself, k_count_tiles fills it afterwards)
    hipError_t e = hipMemsetAsync(out->tile_counts, 0, sizeof(int32_t) * 2 * n_tiles, stream);
    if (e != hipSuccess) {
      d4gs_set_error("hipMemsetAsync(tile_counts): %s", hipGetErrorString(e));
      return D4GS_ELAUNCH;
    }
  }
  // Many motion bases (K >= 10: cfg5's 12, the reference's 20): the S x K x 9 blended values are built ONCE by a small launch in front
  // and every lane's 9 K multiply-adds per sub-sample take them as scalar operands instead of LDS broadcast reads - refdefault 73.4 ->
  // 61.2 us, cfg5 382 -> 357; at K = 6 the saving (1.5 us) is less than the launch (profiles/r05_ab_bases_table.txt).
  // D4GS_FWD_TABLE=0 / 1 forces it off / on (A/B).
  static const int tab_env = []() { const char *e = getenv("D4GS_FWD_TABLE"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  a.use_table = dims->G > 0 && out->blend_bases && (tab_env < 0 ? dims->K >= 10 : tab_env == 1);
  if (a.use_table) {
    int rc0 = d4gs_bases_table_launch(dims, in, out->blend_bases, stream);
    if (rc0) return rc0;
  }
  {
    const bool xt = dims->flags & D4GS_EXACT_TILES;
    const void *fn = xt ? (a.use_table ? (const void *)k_project_fwd<true, true> : (const void *)k_project_fwd<true, false>)
                        : (a.use_table ? (const void *)k_project_fwd<false, true> : (const void *)k_project_fwd<false, false>);
    // sub-sample groups (the kernel's header comment): as many as keep the launch within ONE round of the chip's 2 048 four-wave slots
    static const int sg_env = getenv("D4GS_PROJ_SG") ? atoi(getenv("D4GS_PROJ_SG")) : 0;  // A/B hook
    // (measured, profiles/r06_ab_proj_groups.txt: the training shape - 547 blocks - 57.7 -> 44.1 us with 3 groups; cfg2 - 1 172 blocks - 65.3 -> 62.3
    // with 2, 71 with 4; cfg5 - 3 907 blocks, two rounds already - slower with 2)
    int sg = sg_env > 0 ? sg_env : 2048 / (blocks > 0 ? blocks : 1);
    if (sg_env <= 0 && sg < 2 && blocks < 1536) sg = 2;
    sg = sg < 1 ? 1 : (sg > dims->S ? dims->S : sg);
    ProfScope _ps("k_project_fwd", stream);
    void *kargs[] = {(void *)&a};
    (void)hipLaunchKernel(fn, dim3(blocks, sg), dim3(D4GS_PROJ_BLOCK), kargs, lds, stream);
  }
  int rc = d4gs_check_launch("k_project_fwd");
  if (rc) return rc;
  if (a.count_apart) {
    const int pt = d4gs_chunk_per_thread(dims), per_block = COUNT_THREADS * pt;
    const int cblocks = ((dims->N + per_block - 1) / per_block) * dims->S;
    const bool lazy = d4gs_lazy_on(dims, out);
    LazyWs lw{};
    size_t cbytes = hist_bytes;
    if (lazy) {  // depth range of every sub-sample first: it scales the (tile, depth bucket) histogram
      lw = d4gs_lazy_carve(out->lazy_ws, dims->S, a.tw * a.th);
      cbytes += hist_bytes * lw.nb;
      const int rblocks = 8;
      D4GS_LAUNCH("k_depth_range", k_depth_range, dim3(rblocks, dims->S), dim3(256), 0, stream, (const float *)out->depths,
                  (const int *)out->tiles_touched, dims->N, lw.zr);
    }
#define D4GS_COUNT(PT_, LZ_)                                                                                                  \
  do {                                                                                                                        \
    if (cbytes > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_count_tiles<PT_, LZ_>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); /* (+ the kernel's static LDS <= 160 KB) */ \
    D4GS_LAUNCH("k_count_tiles", (k_count_tiles<PT_, LZ_>), dim3(cblocks), dim3(COUNT_THREADS), cbytes, stream,              \
                (const int *)out->tile_rects, (const int *)out->tiles_touched, dims->N, dims->S, a.tw, a.th, out->tile_counts, \
                fused_chunks ? out->scan_ws : (int *)nullptr, (const float *)out->depths, lw,                                 \
                (const uint64_t *)((dims->flags & D4GS_EXACT_TILES) ? out->tile_masks : nullptr));                            \
  } while (0)
    if (pt == 1 && lazy) D4GS_COUNT(1, true);
    else if (pt == 1) D4GS_COUNT(1, false);
    else if (pt == 2 && lazy) D4GS_COUNT(2, true);
    else if (pt == 2) D4GS_COUNT(2, false);
    else if (pt == 3 && lazy) D4GS_COUNT(3, true);
    else if (pt == 3) D4GS_COUNT(3, false);
    else if (pt == 5 && lazy) D4GS_COUNT(5, true);
    else if (pt == 5) D4GS_COUNT(5, false);
    else if (pt == 6 && lazy) D4GS_COUNT(6, true);
    else if (pt == 6) D4GS_COUNT(6, false);
    else if (lazy) D4GS_COUNT(4, true);
    else D4GS_COUNT(4, false);
#undef D4GS_COUNT
    // (the near / far pivot of every tile is chosen in d4gs_bin_sort, where the caller's D4gsIsect.near_target is known)
    rc = d4gs_check_launch("k_count_tiles");
    if (rc) return rc;
  }
  const int sblocks = fused_chunks ? fused_chunks * dims->S : (int)((n_inst + SCAN_TILE - 1) / SCAN_TILE);
  if (!fused_chunks)
    D4GS_LAUNCH("k_scan_sums", k_scan_sums, dim3(sblocks), dim3(SCAN_THREADS), 0, stream, out->tiles_touched, n_inst,
                       out->scan_ws);
  const ScanJob jsums{out->scan_ws, nullptr, out->scan_ws, (int64_t)sblocks, 0, out->n_isect, nullptr};
  const ScanJob jtiles{out->tile_counts, out->tile_counts + n_tiles, out->tile_offsets, (int64_t)n_tiles, 1, nullptr, out->n_isect + 1};
  D4GS_LAUNCH("k_scan_single", k_scan_single, dim3(2), dim3(1024), 0, stream, jsums, jtiles);
  if (!fused_chunks)  // (fused: k_emit writes isect_offsets)
    D4GS_LAUNCH("k_scan_apply", k_scan_apply, dim3(sblocks), dim3(SCAN_THREADS), 0, stream, out->tiles_touched, out->scan_ws,
                       n_inst, out->isect_offsets);
  return d4gs_check_launch("scan");
}
