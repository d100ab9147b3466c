// raster_bwd.hip -- adjoint of the per-tile composite for all S sub-samples, plus the per-instance gather.
//
// Replaces gsplat rasterize_to_pixels_bwd<CDIM> (back-to-front replay, warp-reduce + atomicAdd into
// v_means2d / v_conics / v_colors / v_opacities) and the autograd of the RGB+ED division.
// Reference: the backward of flow3d/scene_model.py:360-373, driven by flow3d/trainer.py:231.
//
// CDNA4 mapping (k_raster_bwd_q, "variant B", the only composite backward in libd4gs.so; details at raster_bwd_q_body)
//   * one 256-lane workgroup per 16x16 tile (few-tile launches: per (tile, depth segment)), its 4 waves own the 4 8x8 quadrants,
//     1 pixel / lane; 64-splat batches are staged back to front once per tile; a wave replays a staged splat only if its tight
//     alpha >= 1/255 box touches the quadrant and it lies at or before the quadrant's last contributor (ballot);
//   * each replayed splat's row (6 moments + channels) is reduced over the wave's 64 lanes with v_permlane32/16_swap + DPP adds
//     into the wave's own LDS slab; the 4 slabs are added in fixed order at write-out.  60 VGPRs / 15.4 KB LDS for D <= 5 ->
//     8 waves per SIMD.  (Variant A - one wave per tile, 2x2 pixels per lane - and variant C - MFMA reductions - live in
//     variants/raster_bwd_variants.inc, tests' A/B library only.)
//   * NO float atomics: the wave's reduced gradient row (x, y, conic a/b/c, opacity, colours, depth) is written to
//     a per-INTERSECTION buffer at the splat's emission index.  Rows of one Gaussian instance are contiguous there,
//     so k_gather sums them with plain loads in a fixed order -> bitwise reproducible gradients, and no
//     fabric-level atomic traffic (device-scope float atomics are serialised memory-side on MI355X's 8 XCDs).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct RasterBwdArgs {
  int N, S, width, height, tw, th;
  int ed;
  const float *geom;
  const float *ctab;
  const float *background;
  const int32_t *tile_offsets;
  const int32_t *sorted_gid;
  const int32_t *sorted_emit;
  const float *out;     // forward render_colors (needed to undo the ED division)
  const float *alphas;
  const int32_t *last_ids;
  const float *final_T;
  const float *v_out;
  const float *v_alphas;  // may be null
  float *isect_grad;
  uint8_t *live;         // [cap] 1 = row written.  Sparse mode: variant B marks the rows it replays (buffer zeroed first)
  int sparse;            // 0: every row is written (rows behind a tile's last contributor are zero-filled), no flags
  const int64_t *n_dev;  // device {total, longest list}; the lists were sized for (cap, max_hint) - see binning.hip
  int64_t cap, max_hint;
  const float *seg_state;  // SEG instantiations: the forward's per-pixel state at the depth-segment boundaries (common.h)
  int fuse_blend;          // 1: v_out / v_alphas are not given - the pixel's gradients come from the blended frame's (`blend`)
  BlendAdj blend;
#ifdef D4GS_TRACE  // A/B builds only (scripts/trace_wgs.py): per-workgroup {start, end} wall clock, hardware id, list entries
  unsigned long long *trace;
#endif
};
#ifdef D4GS_TRACE
#define D4GS_TRACE_BEGIN const unsigned long long _t0 = wall_clock64();
#define D4GS_TRACE_END(n_)                                                                                     \
  if (a.trace && threadIdx.x == 0) {                                                                           \
    unsigned long long *tr = a.trace + (size_t)blockIdx.x * 4;                                                 \
    tr[0] = _t0, tr[1] = wall_clock64(), tr[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32), tr[3] = (unsigned long long)(n_); \
  }
#else
#define D4GS_TRACE_BEGIN
#define D4GS_TRACE_END(n_)
#endif

// The forward stage skips its work when the device-side intersection count exceeds what the caller sized the lists for
// (optimistic / deferred sizing, engine.py).  The backward must not touch those lists either: tile_offsets are computed from
// the TRUE counts and would index past the end of sorted_gid / isect_grad.
__device__ __forceinline__ bool lists_overflowed(const int64_t *n_dev, int64_t cap, int64_t max_hint) {
  return n_dev && (n_dev[0] > cap || (max_hint > 0 && n_dev[1] > max_hint));
}


__device__ __forceinline__ int xcd_remap_b(int b, int n_blocks) {
  const int per = (n_blocks + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}


// ---------------------------------------------------------------------------------------------------------------
// Variant B (default): one 256-lane workgroup per tile = 4 waves, each wave owns an 8x8 QUADRANT (1 pixel / lane),
// batches of 64 splats staged back-to-front once per tile together with their tight alpha >= 1/255 boxes.  Every
// wave ballots which staged splats (a) touch its quadrant and (b) lie at or before the quadrant's last contributor,
// and replays only those.  Each wave reduces its row over its 64 lanes (permlane swaps) into its own LDS slab; the
// four slabs are added in fixed order when the batch is written out - still no atomics, still deterministic.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// amdgpu_waves_per_eu(8, 8): the kernel's time follows ~ 415 us + 2000 us / (waves per SIMD) on cfg2 (measured by padding the
// workgroup's LDS: 2 -> 1413, 3 -> 1034, 4 -> 863, 7 -> 700 us); asked for 8 the compiler fits D <= 5 in 52-64 VGPRs without
// scratch (68 before: 7 waves): 700 -> 687 us.  The wide instantiations (D = 8, 16) keep their registers - the hint cannot be met.
// SEG (round 4, few-tile launches): one workgroup per (tile, depth segment) instead of per tile.  A segment [lo, hi_ex) of the
// list is replayed exactly like a whole list, except that a pixel whose last contributor lies BEHIND the segment does not start
// from (final T, empty suffix) but from the state the sequential replay would have when it arrives at hi_ex - 1: the
// transmittance the forward stored at that boundary and the suffix dot product <v_out, C_final - C_boundary> (what `bsum` has
// accumulated by then).  Rows are still written once, by the segment that owns them: no atomics, run-to-run deterministic;
// against the unsegmented replay the gradients differ by the fp32 rounding of that hand-off.
template <int D, bool DEPTH, bool SEG = false>
__device__ __forceinline__ void raster_bwd_q_body(const RasterBwdArgs &a) {
#pragma clang fp contract(off)
  constexpr int NCH = D + (DEPTH ? 1 : 0);
  constexpr int DP = (D + 3) & ~3;
  constexpr int DV = DP / 4;
  constexpr int R = 6 + NCH;
  // >= 16 channels (the reference's 17-channel training renders): 4 waves per SIMD instead of 3 (round 4).  The kernel's time goes
  // with 1 / occupancy and its workgroup needed 50.4 KB of LDS and 140 VGPRs; an 8-hit `fac` tile (the MFMA's other 8 columns idle:
  // twice the matrix instructions per hit), the staged boxes as 4 x f16 and an even slab stride bring it to 40.6 KB / 113 VGPRs
  // without scratch: refdefault's backward 1 066 -> 1 023 us, cfg2 with 17 channels 1 315 -> 1 258 (D4GS_BWD16_4W=0: the old shape).
#ifndef D4GS_BWD16_4W
#define D4GS_BWD16_4W 1
#endif
  constexpr bool W4 = D4GS_BWD16_4W && D >= 16;
  constexpr int RP = W4 ? (R + 1) : ((R + 1) | 1);
  constexpr int NB = 64;  // splats per batch
  // With >= 16 colour channels the colour gradients V_c[j] = sum_p vo[c][p] * fac[p][j] of channels 0..15 leave the
  // VALU: they are a [16 ch x 64 px] x [64 px x 16 hits] product, A = this quadrant's image gradient (fixed for the
  // whole kernel, 16 registers), B = the hits' `fac` parked in LDS; one v_mfma_f32_16x16x4_f32 per hit replaces 16
  // multiplies + 16 wave reductions.  The remaining rows (6 moments + channels >= 16, i.e. depth) stay on the VALU.
  constexpr int MC = D >= 16 ? 16 : 0;        // channels reduced on the matrix pipe
  constexpr int RV = R - MC;                  // rows reduced on the VALU
  constexpr int CB = MC ? RV + 1 : 6;         // slab column of channel 0 (VALU rows, their pad slot, then colours)
  constexpr int MH = W4 ? 8 : 16;             // hits parked per flush (the MFMA's 16 columns: the upper 8 idle when MH = 8)
  constexpr int FS = MH + 1;                  // fac tile [64 px][MH hits], row stride (bank-conflict padding)
  static_assert(!MC || CB + MC <= RP, "slab row too short");
  constexpr float LN2 = 0.6931471805599453f;
  __shared__ float sfac[MC ? 4 * 64 * FS : 1];
  __shared__ int shit[MC ? 4 * MH : 1];
  __shared__ float4 sg0[NB];
  __shared__ float4 sg1[NB];
  __shared__ typename std::conditional<W4, uint2, float4>::type sbox[NB];  // W4: 4 x f16, tile-local (the forward's pack_box)
  __shared__ float4 scol[NB * DV];
  __shared__ __attribute__((aligned(16))) float sgrad[4 * NB * RP];
  __shared__ int shi[4];

  D4GS_TRACE_BEGIN
  if (lists_overflowed(a.n_dev, a.cap, a.max_hint)) return;
  const int n_tiles_s = a.tw * a.th;
  const int n_tiles = a.S * n_tiles_s;
  // SEG: block b runs on XCD b % 8; inside an XCD the order is SEGMENT-major over the XCD's tiles (all first segments, then
  // all second ones, ...): a tile's segments share its XCD's L2, and the slots past a list's end - workgroups that exit at
  // once - come last.  (Tile-major order interleaves working and empty workgroups with a period the dispatcher's round robin
  // over shader engines / CUs aliases with: measured, half of the CUs received only empty workgroups.)
  const int per_xcd = (n_tiles + 7) >> 3, bi = blockIdx.x >> 3;
  const int seg = SEG ? bi / per_xcd : 0;
  const int t = SEG ? (blockIdx.x & 7) * per_xcd + (bi - seg * per_xcd) : xcd_remap_b(blockIdx.x, n_tiles);
  if (t >= n_tiles || seg >= D4GS_SEG_MAX) return;
  int start = a.tile_offsets[t], end = a.tile_offsets[t + 1];
  if constexpr (SEG) {  // from here on [start, end) is this workgroup's segment of the list
    const int sl = d4gs_seg_len(end - start);
    start += seg * sl;
    end = min(start + sl, end);
  }
  if (end <= start) return;
  const int s = t / n_tiles_s, tl = t - s * n_tiles_s;
  const int ty = tl / a.tw, tx = tl - ty * a.tw;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int qx0 = tx * D4GS_TILE + (wv & 1) * 8, qy0 = ty * D4GS_TILE + (wv >> 1) * 8;
  const int x = qx0 + (lane & 7), y = qy0 + (lane >> 3);
  const bool inside = x < a.width && y < a.height;
  const float pxf = (float)x + 0.5f, pyf = (float)y + 0.5f;
  // (W4: boxes and quadrant bounds relative to the tile origin, where 4 x f16 resolve them)
  const float tx0f = W4 ? (float)(tx * D4GS_TILE) : 0.f, ty0f = W4 ? (float)(ty * D4GS_TILE) : 0.f;
  const float qlx = (float)qx0 - tx0f + 0.5f, qhx = (float)qx0 - tx0f + 7.5f, qly = (float)qy0 - ty0f + 0.5f, qhy = (float)qy0 - ty0f + 7.5f;

  float T = 1.f, va = 0.f, vo[NCH], bsum = 0.f;
  int last = -1;
#pragma unroll
  for (int c = 0; c < NCH; c++) vo[c] = 0.f;
  if (inside) {
    const size_t pix = ((size_t)s * a.height + y) * a.width + x;
    last = a.last_ids[pix];
    const float al = a.alphas[pix];
    const float Tfin = a.final_T[pix];
    float v_al;
    if (a.fuse_blend) {  // (workgroup-uniform) the exposure blend's adjoint, k_blend_bwd's arithmetic
      const size_t pb = (size_t)y * a.width + x;
      const float inv = 1.f / (float)a.S;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const float g = a.blend.v_blended ? a.blend.v_blended[pb * NCH + c] : 0.f;
        float v = g * inv;
        if (((a.blend.non_mean >> c) & 1) && a.S > 1) {
          const int winner = a.blend.win[pb * NCH + c];  // -1: the mean receives the gradient
          if (winner >= 0) v = s == winner ? g : 0.f;
        }
        vo[c] = v;
      }
      v_al = a.blend.v_acc ? a.blend.v_acc[pb] / (float)a.S : 0.f;
    } else {
      const float *vp = a.v_out + pix * NCH;
#pragma unroll
      for (int c = 0; c < NCH; c++) vo[c] = vp[c];
      v_al = a.v_alphas ? a.v_alphas[pix] : 0.f;
    }
    if (DEPTH && a.ed) {
      const float den = fmaxf(al, 1e-10f);
      const float vd = vo[D];
      if (al >= 1e-10f) v_al -= vd * a.out[pix * NCH + D] / den;
      vo[D] = vd / den;
    }
    float bgdot = 0.f;
    if (a.background) {
#pragma unroll
      for (int c = 0; c < D; c++) bgdot += a.background[c] * vo[c];
    }
    va = Tfin * (v_al - bgdot);
    T = Tfin;
    if constexpr (SEG) {
      if (last >= end) {  // contributors behind this segment: take over at its back boundary (slot seg + 1; slot 0 = final state)
        const float *sf = a.seg_state + (size_t)t * D4GS_SEG_MAX * (1 + NCH) * 256 + ((y - ty * D4GS_TILE) * D4GS_TILE + (x - tx * D4GS_TILE));
        const float *sb = sf + (size_t)(seg + 1) * (1 + NCH) * 256;
        T = sb[0];
#pragma unroll
        for (int c = 0; c < NCH; c++) bsum = __builtin_fmaf(vo[c], sf[(1 + c) * 256] - sb[(1 + c) * 256], bsum);
        last = end - 1;
      }
    }
  }
  // last contributor of this quadrant / of the tile
  int whi = last < start ? start - 1 : last;
  // ... and of each of the quadrant's four 4x4 blocks (round 5): a staged splat is replayed only if its box reaches a block that still
  // has a contributor at or behind it.  The replays this prunes lit no lane (the pixels inside the box were all saturated in front
  // of the splat, the others fail the alpha test), so the rows are the same bit for bit.
#ifndef D4GS_BWD_BLOCK_LAST
#define D4GS_BWD_BLOCK_LAST 1
#endif
  int wq = whi;  // after xor 1, 2 (x within the block) and 8, 16 (y within the block): the lane's own block
#pragma unroll
  for (int o = 1; o <= 16; o = o == 2 ? 8 : o << 1) wq = max(wq, __shfl_xor(wq, o));
  const int wq0 = __builtin_amdgcn_readlane(wq, 0), wq1 = __builtin_amdgcn_readlane(wq, 4), wq2 = __builtin_amdgcn_readlane(wq, 32),
            wq3 = __builtin_amdgcn_readlane(wq, 36);  // blocks (x, y) = (0, 0), (1, 0), (0, 1), (1, 1)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) whi = max(whi, __shfl_xor(whi, o));
  whi = min(whi, end - 1);
  if (lane == 0) shi[wv] = whi;
  __syncthreads();
  const int hi = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
  const size_t inst_base = (size_t)s * a.N;
  float *myslab = sgrad + wv * NB * RP;
  const int wvs = __builtin_amdgcn_readfirstlane(wv);  // wave-uniform: keeps the slab row address scalar
  // A fragments (lane l holds A[row = l & 15][k = l >> 4]): channel l & 15 at quadrant pixel 4 kk + (l >> 4)
  float afrag[MC ? 16 : 1];
  float *myfac = sfac + (MC ? wv * 64 * FS : 0);
  int *myhit = shit + (MC ? wv * MH : 0);
  int nh = 0;
  if constexpr (MC > 0) {
    static_assert(256 * MC <= 4 * NB * RP, "exchange buffer");
#pragma unroll
    for (int c = 0; c < MC; c++) sgrad[tid * MC + c] = vo[c];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; kk++) afrag[kk] = sgrad[(wv * 64 + 4 * kk + (lane >> 4)) * MC + (lane & 15)];
  }
  // nhits parked hits -> colour rows of this wave's slab (C/D layout: col = l & 15 (hit), row = (l >> 4) * 4 + reg)
  auto flush = [&](int nhits) {
    if constexpr (MC > 0) {
      __builtin_amdgcn_wave_barrier();
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // even / odd k-steps, added in fixed order
      // MH = 8: columns 8..15 of the B operand repeat columns 0..7 (their products are never stored) - plain loads, all sixteen in
      // flight, instead of sixteen exec-masked ones each waited for in front of its MFMA
      const float *bp = myfac + (lane >> 4) * FS + (lane & (MH - 1));
      float bv[16];
#pragma unroll
      for (int kk = 0; kk < 16; kk++) bv[kk] = bp[4 * kk * FS];
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[kk], bv[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[kk + 1], bv[kk + 1], acc1, 0, 0, 0);
      }
      acc0 += acc1;
      const int n = lane & 15;
      if (n < nhits) {
        float *dst = myslab + myhit[n] * RP + CB + (lane >> 4) * 4;
        dst[0] = acc0[0], dst[1] = acc0[1], dst[2] = acc0[2], dst[3] = acc0[3];
      }
      __builtin_amdgcn_wave_barrier();
    }
  };
  // Rows behind the tile's last contributor are never replayed.  Dense mode zero-fills them here (no whole-buffer
  // memset); sparse mode leaves them unwritten - their `live` byte stays 0 and k_gather skips them: in occluded /
  // large-footprint scenes that is most rows and zero-filling them costs more than the replay itself.
  if (!a.sparse)
    for (int idx = hi + 1 + tid; idx < end; idx += 256) {
      float *dst = a.isect_grad + (size_t)a.sorted_emit[idx] * R;
#pragma unroll
      for (int r = 0; r < R; r++) dst[r] = 0.f;
    }

  // va - (the suffix sum of fac * d over the contributors behind the current one), kept as ONE running value: a multiply-add per hit
  // instead of a subtraction and a multiply-add
  float vab = va - bsum;
  for (int bh = hi; bh >= start; bh -= NB) {
    __syncthreads();
    int emit = -1;
    if (tid < NB) {
      const int idx = bh - tid;
      if (idx >= start) {
        const int gid = a.sorted_gid[idx];
        emit = a.sorted_emit[idx];
        const float4 *gp = reinterpret_cast<const float4 *>(a.geom + (inst_base + gid) * D4GS_GEOM_STRIDE);
        const float4 q0 = gp[0], q1 = gp[1];
        sg0[tid] = q0;
        sg1[tid] = stage_conic(q1.x, q1.y, q1.z, __builtin_amdgcn_rcpf(q0.z));
        const float tau = __logf(255.f * q0.z) * 1.01f + 0.02f;
        const float det = q1.x * q1.z - q1.y * q1.y;
        const float idet = 1.f / det;
        float ex = -1.f, ey = -1.f;
        if (tau > 0.f && det > 0.f) {
          ex = sqrtf(2.f * tau * q1.z * idet) + 1e-3f;
          ey = sqrtf(2.f * tau * q1.x * idet) + 1e-3f;
        }
        if constexpr (W4)
          sbox[tid] = ex < 0.f ? d4gs_pack_box(1e30f, -1e30f, 1e30f, -1e30f)
                               : d4gs_pack_box(q0.x - ex - tx0f, q0.x + ex - tx0f, q0.y - ey - ty0f, q0.y + ey - ty0f);
        else
          sbox[tid] = ex < 0.f ? make_float4(1e30f, -1e30f, 1e30f, -1e30f)
                               : make_float4(q0.x - ex, q0.x + ex, q0.y - ey, q0.y + ey);
        const float4 *cp = reinterpret_cast<const float4 *>(a.ctab + (size_t)gid * DP);
#pragma unroll
        for (int v = 0; v < DV; v++) {
          float4 c4 = cp[v];
          // the depth rides in the colour record's padding slot (D = 3: r, g, b, depth): one LDS read per replay instead of two
          if constexpr (DEPTH && (D & 3) != 0) {
            if (v == D / 4) (&c4.x)[D & 3] = q0.w;
          }
          scol[tid * DV + v] = c4;
        }
      }
    }
#ifndef D4GS_BWD_ZERO128
#define D4GS_BWD_ZERO128 1
#endif
    if constexpr (D4GS_BWD_ZERO128) {  // 16 bytes per store (4 * NB * RP floats = 64 RP float4)
      for (int z = tid; z < NB * RP; z += 256) reinterpret_cast<float4 *>(sgrad)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int z = tid; z < 4 * NB * RP; z += 256) sgrad[z] = 0.f;
    }
    __syncthreads();
    const int nb = min(NB, bh - start + 1);
#pragma unroll
    for (int k = 0; k < (NB + 63) / 64; k++) {
      const int jj = k * 64 + lane;
      bool hit = false;
      if (jj < nb && bh - jj <= whi) {
        float4 bx;
        if constexpr (W4) bx = d4gs_unpack_box(sbox[jj]);
        else bx = sbox[jj];
        if constexpr (D4GS_BWD_BLOCK_LAST) {
          const int cur = bh - jj;
          const bool X0 = (bx.x <= qlx + 3.f) && (bx.y >= qlx), X1 = (bx.x <= qhx) && (bx.y >= qlx + 4.f);
          const bool Y0 = (bx.z <= qly + 3.f) && (bx.w >= qly), Y1 = (bx.z <= qhy) && (bx.w >= qly + 4.f);
          hit = (X0 && Y0 && cur <= wq0) || (X1 && Y0 && cur <= wq1) || (X0 && Y1 && cur <= wq2) || (X1 && Y1 && cur <= wq3);
        } else {
          hit = (bx.x <= qhx) && (bx.y >= qlx) && (bx.z <= qhy) && (bx.w >= qly);
        }
      }
      unsigned long long m = __ballot(hit);
      while (m) {
        const int j = k * 64 + (__ffsll((long long)m) - 1);
        m &= m - 1;
        const int cur = bh - j;
#ifndef D4GS_BWD_PIN_ADDR
#define D4GS_BWD_PIN_ADDR 1
#endif
        // narrow kernels: the staged records' byte offset in ONE pinned VGPR for all three reads (the compiler re-materialised it from
        // the scalar for the read behind the early-out): 627 -> 619 us on cfg2.  (The 17-channel kernel got SLOWER with it, and with g0
        // as one 16-byte read instead of 12 + 4: 888 -> 926 us on refdefault - it keeps the plain indexing.)
        int jb = j * 16;
        if constexpr (D4GS_BWD_PIN_ADDR && MC == 0) asm volatile("" : "+v"(jb));
        const float4 g0 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sg0) + jb);
        const float4 g1 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sg1) + jb);
        const float dx = g0.x - pxf, dy = g0.y - pyf;
        const float sig2 = splat_sigma2(g1, dx, dy);
        const float ov = g0.z * __builtin_amdgcn_exp2f(-sig2);
        const float alpha = fminf(0.999f, ov);
        const bool valid = (cur <= last) && (sig2 >= 0.f) && (alpha >= (1.f / 255.f));
        if (!__any(valid)) continue;
        const float am = valid ? alpha : 0.f;
        const float ra = __builtin_amdgcn_rcpf(1.f - am);
        T *= ra;
        const float fac = am * T;
        float row[RV];
        float d = 0.f;
#pragma unroll
        for (int v = 0; v < DV; v++) {
          const float4 c4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(scol) + jb * DV + 16 * v);
          constexpr int DC = D + ((DEPTH && (D & 3) != 0) ? 1 : 0);  // channels the record holds (vo[D] = the depth channel)
          if (v * 4 < DC) d = __builtin_fmaf(vo[v * 4], c4.x, d);
          if (v * 4 + 1 < DC) d = __builtin_fmaf(vo[v * 4 + 1], c4.y, d);
          if (v * 4 + 2 < DC) d = __builtin_fmaf(vo[v * 4 + 2], c4.z, d);
          if (v * 4 + 3 < DC) d = __builtin_fmaf(vo[v * 4 + 3], c4.w, d);
        }
        if (DEPTH && (D & 3) == 0) d = __builtin_fmaf(vo[D], g0.w, d);
#pragma unroll
        for (int c = MC; c < NCH; c++) row[6 + c - MC] = fac * vo[c];
        if constexpr (MC > 0) {
          myfac[lane * FS + nh] = fac;
          if (lane == 0) myhit[nh] = j;
        }
        const float v_alpha = __builtin_fmaf(T, d, ra * vab);
        vab = __builtin_fmaf(-fac, d, vab);
        const bool ok = valid && (ov <= 0.999f);
        const float vs = ok ? -ov * v_alpha : 0.f;
        // raw pixel sums only; the per-splat linear map (conic, ln2, 1/2, 1/opacity) is applied once at write-out
        const float vsx = vs * dx, vsy = vs * dy;
        row[0] = vsx;
        row[1] = vsy;
        row[2] = vsx * dx;
        row[3] = vsx * dy;
        row[4] = vsy * dy;
        row[5] = vs;
        wave_sum_store(row, sgrad, (wvs * NB + j) * RP, lane);
        if constexpr (MC > 0) {
          if (++nh == MH) {
            flush(MH);
            nh = 0;
          }
        }
      }
    }
    if constexpr (MC > 0) {
      if (nh) flush(nh);
      nh = 0;
    }
    __syncthreads();
    if (emit >= 0) {
      float sum[R];  // back in the row order of isect_grad: 6 moments, then channels 0..NCH-1
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int q = !MC || r < 6 ? r : (r - 6 < MC ? CB + r - 6 : r - MC);
        sum[r] = (sgrad[tid * RP + q] + sgrad[(NB + tid) * RP + q]) + (sgrad[(2 * NB + tid) * RP + q] + sgrad[(3 * NB + tid) * RP + q]);
        if constexpr (d4gs_wave_sum_split_last<RV>()) {  // (wave_sum_store: the last VALU row arrives as two partial sums, slots RV - 1 and RV)
          if (q == RV - 1)
            sum[r] += (sgrad[tid * RP + RV] + sgrad[(NB + tid) * RP + RV]) + (sgrad[(2 * NB + tid) * RP + RV] + sgrad[(3 * NB + tid) * RP + RV]);
        }
      }
      const float4 g1 = sg1[tid];  // conic * log2(e), 1 / opacity
      float *dst = a.isect_grad + (size_t)emit * R;
      const float o0 = (2.f * g1.x * sum[0] + g1.y * sum[1]) * LN2;  // dL/dx = a Sum(vs dx) + b Sum(vs dy); g1 = (a/2, b, c/2) log2e
      const float o1 = (g1.y * sum[0] + 2.f * g1.z * sum[1]) * LN2;
      if constexpr (R % 2 == 0 && R <= 12) {  // even rows start on 8-byte boundaries: half as many stores (narrow rows only: registers)
        float2 *d2 = reinterpret_cast<float2 *>(dst);
        d2[0] = make_float2(o0, o1);
        d2[1] = make_float2(0.5f * sum[2], sum[3]);
        d2[2] = make_float2(0.5f * sum[4], -sum[5] * g1.w);
#pragma unroll
        for (int c = 0; c < NCH; c += 2) d2[3 + c / 2] = make_float2(sum[6 + c], sum[7 + c]);
      } else {
        dst[0] = o0;
        dst[1] = o1;
        dst[2] = 0.5f * sum[2];
        dst[3] = sum[3];
        dst[4] = 0.5f * sum[4];
        dst[5] = -sum[5] * g1.w;
#pragma unroll
        for (int c = 0; c < NCH; c++) dst[6 + c] = sum[6 + c];
      }
      if (a.sparse) a.live[emit] = 1;
    }
  }
  D4GS_TRACE_END(end - start)
}
// Two entry points over one body: the narrow instantiations (D <= 5) are asked for 8 waves per SIMD (the hint changes the
// scheduler's register budget); the wide ones keep the compiler's default - the hint cannot be met there and only perturbs them.
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_raster_bwd_q8(const RasterBwdArgs a) {
  raster_bwd_q_body<D, DEPTH>(a);
}
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256)
#if defined(D4GS_BWD16_4W) && D4GS_BWD16_4W
__attribute__((amdgpu_waves_per_eu(D >= 16 ? 4 : 1, D >= 16 ? 4 : 10)))
#endif
k_raster_bwd_q(const RasterBwdArgs a) {
  raster_bwd_q_body<D, DEPTH>(a);
}
// the same two over (tile, depth segment) workgroups
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_raster_bwd_qs8(const RasterBwdArgs a) {
  raster_bwd_q_body<D, DEPTH, true>(a);
}
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256) k_raster_bwd_qs(const RasterBwdArgs a) {
  raster_bwd_q_body<D, DEPTH, true>(a);
}

#ifdef D4GS_VARIANTS
#include "variants/raster_bwd_variants.inc"
#endif


// one lane per Gaussian, looping over sub-samples: sums the contiguous per-intersection rows of each instance
struct GatherArgs {
  int N, S, D, DP, depth;
  const int32_t *tiles_touched;
  const int32_t *isect_offsets;
  const float *isect_grad;
  const int64_t *n_dev;  // see RasterBwdArgs: an overflowed render has no valid rows - its gradients are zeros
  int64_t cap, max_hint;
  int64_t rows;  // rows of isect_grad the caller allocated (>= 1)
  const uint8_t *live;  // [rows] 1 = the row was written by the composite backward (sparse mode)
  int sparse;
  float *v_means2d, *v_conics, *v_depths, *v_opac_act, *v_ctab;
  // fused densification statistics (trainer.py:967-989); stats_acc == nullptr: off
  const int32_t *radii;
  float *stats_acc;
  int64_t *stats_vis;
  float *stats_mr;
  float sx, sy, max_wh;
  int update_mr;
};

// The rows of the 64 instances a wave owns (same sub-sample, consecutive Gaussians) form ONE contiguous span of
// isect_grad: the wave streams it into LDS with coalesced loads and every lane then sums its own rows from there, in
// the same k order as a direct read (bit-identical).  Spans longer than the LDS budget (wide splats) take several chunks.
// Round 4: a block owns 64 Gaussians and its 4 waves are 4 sub-sample SLOTS (wave w gathers sub-samples w, w + 4, ...), like
// k_project_bwd: the kernel used to loop over the S sub-samples inside ONE wave per 64 Gaussians - N / 64 waves in all (4 688 on
// cfg2 / cfg3, 2 188 on the reference's training shape), each a chain of S dependent flag / row fetches; 2 waves per SIMD resident
// on average (profiles/r04t_pmc_gather.txt).  The per-Gaussian sums over the sub-samples (opacity, colours, densification statistics)
// keep their order: after every round of 4 sub-samples the slots hand their contributions to wave 0 through LDS and wave 0 adds
// them in ascending s - bit for bit what the single wave computed.
// rows of LDS per wave: wide rows (D >= 8: 56 - 92 bytes) take half the rows, i.e. about the same bytes - the training shape's gather
// 91 -> 77 us with twice the blocks per CU; narrow rows lose with smaller stages (cfg2 50 -> 53 -> 61 us at 128 / 96;
// profiles/r04t_ab_gather_rows.txt)
#ifndef D4GS_GATHER_ROWS_SPARSE
#define D4GS_GATHER_ROWS_SPARSE 192
#endif
constexpr int gather_rows(int D, bool sparse) { return D >= 8 ? 96 : sparse ? D4GS_GATHER_ROWS_SPARSE : 192; }
constexpr int GATHER_SC = 1024;                         // rows per super-chunk of the cooperative sparse path (16 flags per lane)
template <int D, bool DEPTH, bool SPARSE, int SLOTS /* waves per block: 4, or S when the call has fewer sub-samples */>
__global__ void __launch_bounds__(SLOTS * 64) k_gather(const GatherArgs a) {
  constexpr int NCH = D + (DEPTH ? 1 : 0);
  constexpr int DP = (D + 3) & ~3;
  constexpr int R = 6 + NCH;
  constexpr int GATHER_ROWS = gather_rows(D, SPARSE);
  __shared__ __attribute__((aligned(16))) float stage[SLOTS * GATHER_ROWS * R];
  __shared__ uint32_t slive[SPARSE ? SLOTS * 64 : 1];  // flags of a chunk as 4-byte words (<= GATHER_ROWS + 3 bytes)
  // cooperative sparse path (super-chunks of SC rows): the flags, the exclusive live-row count in front of every 16-row group
  // and the compacted list of live rows
  __shared__ __attribute__((aligned(16))) uint32_t sflag[SPARSE ? SLOTS * (GATHER_SC / 4) : 1];
  __shared__ uint16_t sgrp[SPARSE ? SLOTS * (GATHER_SC / 16 + 1) : 1];
  __shared__ uint16_t slist[SPARSE ? SLOTS * GATHER_SC : 1];
  static_assert(GATHER_ROWS + 6 <= 256, "one flag word per lane");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = blockIdx.x * 64 + lane;
  const bool in = g < a.N;
  float *mine = stage + wv * GATHER_ROWS * R;
  uint32_t *mylive32 = slive + (SPARSE ? wv * 64 : 0);
  const uint8_t *mylive = reinterpret_cast<const uint8_t *>(mylive32);
  float vo = 0.f, vc[D];
#pragma unroll
  for (int c = 0; c < D; c++) vc[c] = 0.f;
  // densification statistics of this Gaussian, accumulated over the sub-samples in order (k_control_stats' arithmetic)
  const bool overflow = lists_overflowed(a.n_dev, a.cap, a.max_hint);  // grid-uniform
  const bool stats = a.stats_acc != nullptr && !overflow;
  float st_acc = 0.f, st_mr = 0.f;
  int64_t st_vis = 0;
  if (stats && in) st_acc = a.stats_acc[g], st_vis = a.stats_vis[g], st_mr = a.stats_mr[g];
  // the (count, offset) pair of the next sub-sample is fetched while the current one is streamed and summed
  const size_t gi = in ? g : a.N - 1;
  // (an overflowed render never wrote isect_offsets - k_emit returned early: do not read it, stream nothing)
  int cnt_n = 0, off_n = 0;
  if (wv < a.S) cnt_n = (in && !overflow) ? a.tiles_touched[(size_t)wv * a.N + gi] : 0, off_n = overflow ? 0 : a.isect_offsets[(size_t)wv * a.N + gi];
  static_assert(GATHER_ROWS * R >= 64 * (3 + D), "the hand-off record of a slot lives in its stage slice");
  for (int s0 = 0; s0 < a.S; s0 += SLOTS) {
    const int s = s0 + wv;  // (wave-uniform)
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.f;
    if (s < a.S) {
    const size_t i = (size_t)s * a.N + gi;
    const int cnt = cnt_n, off = off_n;
    if (s + SLOTS < a.S) {
      cnt_n = (in && !overflow) ? a.tiles_touched[i + (size_t)SLOTS * a.N] : 0;
      off_n = overflow ? 0 : a.isect_offsets[i + (size_t)SLOTS * a.N];
    }
    // span of the wave: [first lane's offset, last lane's offset + count)
    const int base = __builtin_amdgcn_readfirstlane(off);
    int endl = overflow ? 0 : off + cnt;  // overflowed lists: the offsets index past the buffer - nothing is streamed
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) endl = max(endl, __shfl_xor(endl, o));
    // the span is streamed in chunks of GATHER_ROWS rows; a lane's rows [off, off + cnt) are contiguous, so it adds
    // the part of them that lies in the current chunk - always in ascending k, whatever the chunking.  The copy moves
    // 16-byte words: the span start is rounded down to a row whose byte offset is a multiple of 16 (every APER-th row;
    // GATHER_ROWS keeps the later chunks aligned) and the last word may carry up to 3 floats of the next row.
    constexpr int APER = (R % 4 == 0) ? 1 : (R % 2 == 0) ? 2 : 4;
    static_assert(GATHER_ROWS % APER == 0 && (GATHER_ROWS * R) % 4 == 0, "chunks must start on 16-byte words");
    int cb0 = base & ~(APER - 1), skip_to = 0;
    if constexpr (SPARSE) {
      // Occluded / large-footprint scenes: most rows of the span were never written.  A super-chunk of SC rows is handled
      // by the whole wave: (1) every lane fetches 16 flag bytes, (2) a wave scan compacts the LIVE rows' indices into a list
      // (row order = instance-major, ascending k), (3) the wave fetches those rows 64 at a time - one row per lane, all
      // loads in flight together - into the LDS stage, (4) every lane adds ITS rows from the stage: their positions are the
      // live-row counts in front of its first / behind its last row (group prefix + popcount of the group's flags).  Same
      // k order as the streaming path -> the same bits.  From the first super-chunk that is more than 1 / 2 live on (`L * 2 >
      // se - sb` below) the rest of the span takes the streaming path.
      uint32_t *myflag = sflag + wv * (GATHER_SC / 4);
      uint16_t *mygrp = sgrp + wv * (GATHER_SC / 16 + 1);
      uint16_t *mylist = slist + wv * GATHER_SC;
      const int rows4 = (int)((a.rows + 3) & ~(int64_t)3);  // the flag buffer is padded to whole words
      for (int sb = base & ~15; sb < endl;) {
        const int se = min(sb + GATHER_SC, endl);
        // (1) 16 flags of rows sb + 16 lane .. + 15 (words past the span / the buffer read as 0)
        uint32_t f[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const int r0 = sb + 16 * lane + 4 * w;
          f[w] = (r0 < se && r0 < rows4) ? reinterpret_cast<const uint32_t *>(a.live)[r0 >> 2] & 0x01010101u : 0u;
        }
        // rows in front of the span start (sb rounds down) or behind its end (a word may straddle it) belong to other
        // waves: drop their flags
        if (sb < base || (se & 3)) {  // wave-uniform
#pragma unroll
          for (int w = 0; w < 4; w++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const int r = sb + 16 * lane + 4 * w + b;
              if (r < base || r >= se) f[w] &= ~(1u << (8 * b));
            }
        }
        const int cnt16 = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);
        int incl = cnt16;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o);
          if (lane >= o) incl += t;
        }
        const int L = __shfl(incl, 63);  // live rows of the super-chunk (wave-uniform)
        if (L == 0) {
          sb = skip_to = se;
          continue;
        }
        if (L * 2 > se - sb) break;  // at least half of the rows are live: stream the rest of the span
        const int excl = incl - cnt16;
        *reinterpret_cast<uint4 *>(myflag + 4 * lane) = make_uint4(f[0], f[1], f[2], f[3]);
        mygrp[lane] = (uint16_t)excl;
        if (lane == 63) mygrp[64] = (uint16_t)L;
        // (2) the list
        int pos = excl;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
          for (int b = 0; b < 4; b++)
            if (f[w] & (1u << (8 * b))) mylist[pos++] = (uint16_t)(16 * lane + 4 * w + b);
        __builtin_amdgcn_wave_barrier();
        // (3) + (4) in rounds of at most GATHER_ROWS live rows (the stage): one live row per lane and iteration, then
        // every lane adds the rows of the round that are its own - their list positions are the live-row counts in front of
        // its first / behind its last row (group prefix + popcount of the group's flags); rounds ascend -> ascending k
        const int k0 = max(off, sb), k1 = min(off + cnt, se);
        int p0 = 0, p1 = 0;
        if (k0 < k1) {
          auto before = [&](int k) -> int {  // live rows of the super-chunk in front of row k (sb <= k <= se)
            const int q = k - sb, gq = q >> 4, w = (q >> 2) & 3, b = q & 3;
            if (gq >= 64) return (int)mygrp[64];
            int n = mygrp[gq];
            const uint32_t *fw = myflag + 4 * gq;
            for (int i = 0; i < w; i++) n += (int)__popc(fw[i]);
            return n + (int)__popc(fw[w] & ((1u << (8 * b)) - 1u));
          };
          p0 = before(k0), p1 = before(k1);
        }
        for (int rb = 0; rb < L; rb += GATHER_ROWS) {
          const int re = min(rb + GATHER_ROWS, L);
          for (int e = rb + lane; e < re; e += 64) {
            const float *row = a.isect_grad + (size_t)(sb + mylist[e]) * R;
            float *dst = mine + (e - rb) * R;
            if constexpr (R % 2 == 0) {
#pragma unroll
              for (int r = 0; r < R; r += 2) *reinterpret_cast<float2 *>(dst + r) = *reinterpret_cast<const float2 *>(row + r);
            } else {
#pragma unroll
              for (int r = 0; r < R; r++) dst[r] = row[r];
            }
          }
          __builtin_amdgcn_wave_barrier();
          const int q0 = max(p0, rb), q1 = min(p1, re);
          const float *row = mine + (q0 - rb) * R;
          for (int pp = q0; pp < q1; pp++, row += R) {
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] += row[r];
          }
          __builtin_amdgcn_wave_barrier();
        }
        sb = skip_to = se;
      }
      cb0 = max(cb0, skip_to & ~(APER - 1));  // rows below skip_to are done: the loop below starts at their (aligned) end
    }
    for (int cb = cb0; cb < endl; cb += GATHER_ROWS) {
      const int ce = min(cb + GATHER_ROWS, endl);
      // the chunk's flag bytes, fetched as words BEFORE the rows so that their latency hides behind the row stream
      const int fb = cb & ~3;
      uint32_t fl = 0u;
      if constexpr (SPARSE)
        if (lane < ((ce - fb + 3) >> 2)) fl = reinterpret_cast<const uint32_t *>(a.live)[(fb >> 2) + lane];
      if constexpr (SPARSE) {
        // Mostly dead chunk (occluded scene): streaming it would move mainly stale bytes.  Every lane then reads just its
        // own flagged rows straight from memory - uncoalesced, but a fraction of the traffic.  Same k order, same sums.
        int nlive = __popc(fl & 0x01010101u);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nlive += __shfl_xor(nlive, o);
        if (nlive * 8 < ce - cb) {  // wave-uniform
          mylive32[lane] = fl;
          __builtin_amdgcn_wave_barrier();
          const int k0 = max(max(off, cb), skip_to), k1 = min(off + cnt, ce);
          for (int k = k0; k < k1; k++) {
            if (!mylive[k - fb]) continue;
            const float *row = a.isect_grad + (size_t)k * R;
            if constexpr (R % 2 == 0) {
#pragma unroll
              for (int r = 0; r < R; r += 2) {
                const float2 v = *reinterpret_cast<const float2 *>(row + r);
                acc[r] += v.x, acc[r + 1] += v.y;
              }
            } else {
#pragma unroll
              for (int r = 0; r < R; r++) acc[r] += row[r];
            }
          }
          __builtin_amdgcn_wave_barrier();
          continue;
        }
      }
      const float4 *src = reinterpret_cast<const float4 *>(a.isect_grad + (size_t)cb * R);
      float4 *dst4 = reinterpret_cast<float4 *>(mine);
      // (the buffer holds a.rows rows: the last word of the last row must not be read past its end)
      const int nf4 = (int)min((int64_t)((ce - cb) * R + 3) / 4, ((a.rows - cb) * (int64_t)R) / 4);
      const int tail0 = nf4 * 4, tail1 = (ce - cb) * R;  // floats the word copy could not cover (end of the buffer)
#pragma unroll 4
      for (int f = lane; f < nf4; f += 64) dst4[f] = src[f];
      if (tail0 + lane < tail1) mine[tail0 + lane] = a.isect_grad[(size_t)cb * R + tail0 + lane];
      if constexpr (SPARSE) mylive32[lane] = fl;
      __builtin_amdgcn_wave_barrier();
      const int k0 = max(max(off, cb), skip_to), k1 = min(off + cnt, ce);
      const float *row = mine + (k0 - cb) * R;
      for (int k = k0; k < k1; k++, row += R) {
        if constexpr (SPARSE)
          if (!mylive[k - fb]) continue;  // never written (behind its tile's last contributor): the bytes are stale
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] += row[r];
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (in) {
      *reinterpret_cast<float2 *>(a.v_means2d + i * 2) = make_float2(acc[0], acc[1]);
      a.v_conics[i * 3] = acc[2];
      a.v_conics[i * 3 + 1] = acc[3];
      a.v_conics[i * 3 + 2] = acc[4];
      a.v_depths[i] = DEPTH ? acc[6 + (DEPTH ? D : 0)] : 0.f;
    }
    }  // s < S
    // hand-off: (v_opacity, v_colour[D], v_x, v_y) of this slot's sub-sample, element-major in the slot's (now idle) stage slice
    mine[lane] = acc[5];
#pragma unroll
    for (int c = 0; c < D; c++) mine[(1 + c) * 64 + lane] = acc[6 + c];
    mine[(1 + D) * 64 + lane] = acc[0], mine[(2 + D) * 64 + lane] = acc[1];
    __syncthreads();
    if (wv == 0) {
      for (int w = 0; w < SLOTS && s0 + w < a.S; w++) {  // ascending s: the order of the single-wave loop
        const float *rec = stage + w * GATHER_ROWS * R;
        if (stats && in) {
          const int r = a.radii[(size_t)(s0 + w) * a.N + g];
          if (r > 0) {
            const float gx = rec[(1 + D) * 64 + lane] * a.sx, gy = rec[(2 + D) * 64 + lane] * a.sy;
            st_acc += sqrtf(gx * gx + gy * gy);
            st_vis += 1;
            st_mr = fmaxf(st_mr, (float)r / a.max_wh);
          }
        }
        vo += rec[lane];
#pragma unroll
        for (int c = 0; c < D; c++) vc[c] += rec[(1 + c) * 64 + lane];
      }
    }
    __syncthreads();  // (the slices are stages again in the next round)
  }
  if (!in || wv != 0) return;
  if (stats) {
    a.stats_acc[g] = st_acc;
    a.stats_vis[g] = st_vis;
    if (a.update_mr) a.stats_mr[g] = st_mr;
  }
  a.v_opac_act[g] = vo;
#pragma unroll
  for (int c = 0; c < DP; c++) a.v_ctab[(size_t)g * DP + c] = c < D ? vc[c < D ? c : 0] : 0.f;
}

// Row mode when the caller leaves it to the library (D4gsRasterGrads.row_mode == D4GS_ROWS_AUTO).  deblur4dgs_amd/engine.py does
// NOT: since round 3 it passes DENSE / SPARSE from the live-row fraction the forward composite sampled on the previous render of
// the shape (engine.row_mode_for; D4gsProjOut.n_isect[2..3]).  This fallback only knows the rows per (sub-sample, Gaussian)
// instance the lists were sized for - large footprints go with dead rows: measured 1.8 rows per instance -> dense is 3 %
// faster; 7.4 -> sparse is 13 % faster.
static bool choose_sparse(int row_mode, int64_t n_isect, int64_t n_inst, bool lazy) {
  if (lazy) return true;  // D4GS_LAZY_SORT leaves the far part of an unflagged list as emitted: its dead rows have no sorted_emit
  if (row_mode == D4GS_ROWS_DENSE) return false;
  if (row_mode == D4GS_ROWS_SPARSE) return true;
  return n_isect >= 6 * (n_inst > 0 ? n_inst : 1);
}

template <int D, bool DEPTH>
int launch_bwd(RasterBwdArgs &a, GatherArgs &ga, int64_t n_isect, int row_mode, bool lazy, hipStream_t stream) {
  const int n_tiles = a.S * a.tw * a.th;
  const int blocks = ((n_tiles + 7) / 8) * 8;
  bool launched = false;
  a.sparse = ga.sparse = choose_sparse(row_mode, n_isect, (int64_t)a.S * a.N, lazy) ? 1 : 0;
#ifdef D4GS_VARIANTS  // the A/B build only (tests/libd4gs_variants.so): environment-selected reference variants, dense rows
  static const bool wave_per_tile = getenv("D4GS_BWD_WAVE_PER_TILE") != nullptr;  // variant A
  static const bool use_mfma = getenv("D4GS_BWD_MFMA") != nullptr;                // variant C
  if (wave_per_tile || use_mfma) {
    constexpr int R = 6 + D + (DEPTH ? 1 : 0);
    a.sparse = ga.sparse = 0;
    if (wave_per_tile) {  // variant A relies on a zeroed buffer
      hipError_t e = hipMemsetAsync(a.isect_grad, 0, sizeof(float) * (size_t)R * (size_t)(n_isect > 0 ? n_isect : 1), stream);
      if (e != hipSuccess) {
        d4gs_set_error("hipMemsetAsync(isect_grad): %s", hipGetErrorString(e));
        return D4GS_ELAUNCH;
      }
    }
    if (n_isect > 0) {
      if (wave_per_tile) D4GS_LAUNCH("k_raster_bwd", (k_raster_bwd<D, DEPTH>), dim3(blocks), dim3(64), 0, stream, a);
      else D4GS_LAUNCH("k_raster_bwd_m", (k_raster_bwd_m<D, DEPTH>), dim3(blocks), dim3(256), 0, stream, a);
    }
    launched = true;
  }
#endif
  if (!launched) {
    if (a.sparse) {  // the composite marks the rows it writes
      hipError_t e = hipMemsetAsync(a.live, 0, (((size_t)(n_isect > 0 ? n_isect : 1)) + 3) & ~(size_t)3, stream);
      if (e != hipSuccess) {
        d4gs_set_error("hipMemsetAsync(isect_live): %s", hipGetErrorString(e));
        return D4GS_ELAUNCH;
      }
    }
    if (n_isect > 0 && a.seg_state) {
      const int sblocks = ((n_tiles + 7) / 8) * 8 * D4GS_SEG_MAX;
      if constexpr (D <= 5) D4GS_LAUNCH("k_raster_bwd_q", (k_raster_bwd_qs8<D, DEPTH>), dim3(sblocks), dim3(256), 0, stream, a);
      else D4GS_LAUNCH("k_raster_bwd_q", (k_raster_bwd_qs<D, DEPTH>), dim3(sblocks), dim3(256), 0, stream, a);
    } else if (n_isect > 0) {
      static const int q_env = getenv("D4GS_BWD_Q") ? atoi(getenv("D4GS_BWD_Q")) : 0;  // A/B hook: workgroups per CU
      if constexpr (D <= 5) {
        const int pad = d4gs_lds_pad_for_wgs_per_cu((const void *)k_raster_bwd_q8<D, DEPTH>, q_env);
        D4GS_LAUNCH("k_raster_bwd_q", (k_raster_bwd_q8<D, DEPTH>), dim3(blocks), dim3(256), pad, stream, a);
      } else {
        const int pad = d4gs_lds_pad_for_wgs_per_cu((const void *)k_raster_bwd_q<D, DEPTH>, q_env);
        D4GS_LAUNCH("k_raster_bwd_q", (k_raster_bwd_q<D, DEPTH>), dim3(blocks), dim3(256), pad, stream, a);
      }
    }
  }
  int rc = d4gs_check_launch("k_raster_bwd");
  if (rc) return rc;
  // 4 sub-sample slots per block; a call with fewer sub-samples (a rank's share of an exposure-sharded frame) would idle the rest
  const int slots = ga.S >= 3 ? 4 : ga.S == 2 ? 2 : 1;
  const dim3 ggrid((ga.N + 63) / 64), gblock(slots * 64);
#define D4GS_GATHER(SP_, SL_) D4GS_LAUNCH("k_gather", (k_gather<D, DEPTH, SP_, SL_>), ggrid, gblock, 0, stream, ga)
  if (ga.sparse) {
    if (slots == 4) D4GS_GATHER(true, 4);
    else if (slots == 2) D4GS_GATHER(true, 2);
    else D4GS_GATHER(true, 1);
  } else {
    if (slots == 4) D4GS_GATHER(false, 4);
    else if (slots == 2) D4GS_GATHER(false, 2);
    else D4GS_GATHER(false, 1);
  }
#undef D4GS_GATHER
  return d4gs_check_launch("k_gather");
}

}  // namespace

int d4gs_raster_bwd_impl(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, const D4gsRaster *r,
                         const D4gsRasterGrads *g, const BlendAdj *blend, hipStream_t stream) {
  RasterBwdArgs a;
  a.fuse_blend = blend != nullptr;
  a.blend = blend ? *blend : BlendAdj{};
  a.N = dims->N, a.S = dims->S, a.width = dims->width, a.height = dims->height;
  a.tw = (dims->width + D4GS_TILE - 1) / D4GS_TILE;
  a.th = (dims->height + D4GS_TILE - 1) / D4GS_TILE;
  a.ed = dims->depth_mode == D4GS_DEPTH_ED;
  a.geom = proj->geom, a.ctab = proj->ctab, a.background = r->background;
  a.tile_offsets = proj->tile_offsets, a.sorted_gid = isect->sorted_gid, a.sorted_emit = isect->sorted_emit;
  a.out = r->render_colors, a.alphas = r->render_alphas, a.last_ids = r->last_ids, a.final_T = r->final_T;
  a.v_out = g->v_render_colors, a.v_alphas = g->v_render_alphas, a.isect_grad = g->isect_grad, a.live = g->isect_live;
  a.n_dev = proj->n_isect, a.cap = isect->n_isect, a.max_hint = isect->max_tile_count;
  a.seg_state = d4gs_seg_on(dims, isect, r) ? r->seg_state : nullptr;
#ifdef D4GS_TRACE
  a.trace = getenv("D4GS_TRACE_PTR") ? (unsigned long long *)strtoull(getenv("D4GS_TRACE_PTR"), nullptr, 0) : nullptr;
#endif
  GatherArgs ga;
  ga.n_dev = proj->n_isect, ga.cap = isect->n_isect, ga.max_hint = isect->max_tile_count;
  ga.rows = isect->n_isect > 0 ? isect->n_isect : 1;
  ga.live = g->isect_live;
  ga.N = dims->N, ga.S = dims->S, ga.D = dims->D, ga.DP = (dims->D + 3) & ~3;
  ga.depth = dims->depth_mode != D4GS_DEPTH_NONE;
  ga.tiles_touched = proj->tiles_touched, ga.isect_offsets = proj->isect_offsets, ga.isect_grad = g->isect_grad;
  ga.v_means2d = g->v_means2d, ga.v_conics = g->v_conics, ga.v_depths = g->v_depths, ga.v_opac_act = g->v_opac_act;
  ga.v_ctab = g->v_ctab;
  ga.radii = proj->radii, ga.stats_acc = g->stats_grad_norm_acc, ga.stats_vis = g->stats_vis_count;
  ga.stats_mr = g->stats_max_radii, ga.update_mr = g->stats_update_max_radii;
  // xys_grad[..., 0] *= W / 2 * batch_size * S ; [..., 1] *= H / 2 * batch_size * S   (trainer.py:976-977)
  ga.sx = (float)dims->width / 2.0f * (float)g->stats_batch_size * (float)dims->S;
  ga.sy = (float)dims->height / 2.0f * (float)g->stats_batch_size * (float)dims->S;
  ga.max_wh = (float)(dims->width > dims->height ? dims->width : dims->height);
  const bool dep = dims->depth_mode != D4GS_DEPTH_NONE;
  const bool lazy = d4gs_lazy_on(dims, proj);
#define D4GS_CASE(DD)                                                                             \
  case DD:                                                                                        \
    return dep ? launch_bwd<DD, true>(a, ga, isect->n_isect, g->row_mode, lazy, stream) : launch_bwd<DD, false>(a, ga, isect->n_isect, g->row_mode, lazy, stream);
  switch (dims->D) {
    D4GS_CASE(1)
    D4GS_CASE(2)
    D4GS_CASE(3)
    D4GS_CASE(4)
    D4GS_CASE(5)
    D4GS_CASE(8)
    D4GS_CASE(16)
    default:
      d4gs_set_error("unsupported colour channel count D=%d (instantiated: 1,2,3,4,5,8,16; render wider colour "
                     "vectors in chunks over the same projection / tile lists)", dims->D);
      return D4GS_EINVAL;
  }
#undef D4GS_CASE
}

#ifdef D4GS_VARIANTS
// Test hook of the A/B build only (tests/libd4gs_variants.so, tests/test_gpu_wave_sum.py): ONE wave sums R per-lane values over its 64
// lanes with wave_sum_store - every folding / ladder pattern the kernels instantiate (and the ones they do not), in the slot order the
// callers rely on.
namespace {
template <int R>
__global__ void __launch_bounds__(64) k_test_wave_sum(const float *in, float *out) {
  __shared__ float slab[32];
  const int lane = threadIdx.x;
  float v[R];
#pragma unroll
  for (int r = 0; r < R; r++) v[r] = in[lane * R + r];
  if (lane < 32) slab[lane] = 0.f;
  __syncthreads();
  wave_sum_store(v, slab, 0, lane);
  __syncthreads();
  if (lane < R) {
    float s = slab[lane];
    if (d4gs_wave_sum_split_last<R>() && lane == R - 1) s += slab[R];
    out[lane] = s;
  }
}
}  // namespace
extern "C" D4GS_API int d4gs_test_wave_sum(int R, const float *in, float *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
#define D4GS_WS(RR) \
  case RR:          \
    hipLaunchKernelGGL(k_test_wave_sum<RR>, dim3(1), dim3(64), 0, st, in, out); \
    break;
  switch (R) {
    D4GS_WS(1) D4GS_WS(2) D4GS_WS(3) D4GS_WS(4) D4GS_WS(5) D4GS_WS(6) D4GS_WS(7) D4GS_WS(8) D4GS_WS(9) D4GS_WS(10) D4GS_WS(11) D4GS_WS(12)
    D4GS_WS(13) D4GS_WS(14) D4GS_WS(15) D4GS_WS(16) D4GS_WS(17) D4GS_WS(21) D4GS_WS(22) D4GS_WS(23)
    default:
      return D4GS_EINVAL;
  }
#undef D4GS_WS
  return hipGetLastError() == hipSuccess ? 0 : D4GS_ELAUNCH;
}
#endif
