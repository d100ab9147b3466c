// raster_fwd.hip -- per-tile front-to-back alpha compositing for all S exposure sub-samples.
//
// Replaces gsplat rasterize_to_pixels_fwd<CDIM> (+ the Python-side expected-depth division of
// rasterization(render_mode="RGB+ED")); reference call site flow3d/scene_model.py:360-373.
//
// CDNA4 mapping (k_raster_fwd_r, "variant D", the only forward in libd4gs.so): ONE 256-lane workgroup per 16x16 tile, its 4
// waves own the 4 8x8 QUADRANTS, and each 16-lane ROW of a wave is an independent 4x4-pixel rasterizer (1 pixel / lane):
//   * batches of 256 (128 for wide colour records) splats are staged once per tile in LDS - geometry, conic, colours and the
//     splat's tight alpha >= 1/255 box (4 x f16, tile-local) - by the workgroup's lanes: one coalesced id load + one 32-B
//     geom gather + the colour row per lane;
//   * per batch every wave compacts one list of staged indices PER ROW (ballot + mbcnt over the boxes that touch the row's 4x4
//     block, order preserved) and its four rows walk their own lists in lock-step: one iteration composites up to four
//     different splats, the loop body is branch-free, saturation is checked once per 16 iterations;
//   * 54 VGPRs / 18.7 KB LDS for D <= 4 -> 8 waves per SIMD.
// The earlier mappings (A: one wave per tile with 2x2 pixels per lane; B: quadrant waves without row lists) live in
// variants/raster_fwd_variants.inc and are compiled only into the tests' A/B library.
// Block -> tile mapping is XCD-aware: consecutive logical tiles (same sub-sample, neighbouring tiles, shared
// splats) land on the same XCD so the gathered records hit that XCD's L2.
#include <stdlib.h>

#include "common.h"

namespace {

struct RasterFwdArgs {
  int N, S, width, height, tw, th;
  int ed;  // divide the depth channel by max(alpha, 1e-10)
  const float *geom;
  const float *ctab;
  const float *background;  // [D] or null
  const int32_t *tile_offsets;
  const int32_t *sorted_gid;
  float *out;      // [S,H,W,NCH]
  float *alphas;   // [S,H,W]
  int32_t *last_ids;
  float *final_T;
  int64_t *n_dev;  // device {total, longest list, sampled entries, sampled live entries}: [0], [1] vs the capacity the lists were
                   // sized for (see binning.hip); [2], [3]: this kernel's live-row sample
  int64_t cap, max_hint;
  float *seg_state;  // SEG instantiations: [tiles][D4GS_SEG_MAX][1 + NCH][256] boundary states (common.h "depth segments")
  // D4GS_LAZY_SORT: pass 1 composites the (sorted) near part of every list and flags the tiles that did not saturate within it
  // (their outputs are not written); pass 2, after the far parts of the flagged lists have been sorted, composites those tiles
  // over their whole lists.  pass 0: one pass over whole lists.
  const int32_t *lazy_near;
  int32_t *lazy_flag;
  int lazy_pass;
#ifdef D4GS_TRACE  // A/B builds only (scripts/trace_wgs.py --fwd): per-workgroup wall clock {start, end}, hardware id, list entries,
  unsigned long long *trace;  // and the time spent in the staging / list-building / compositing phases of its batches
#endif
};
#ifdef D4GS_TRACE
#define D4GS_TCLK(v_) const unsigned long long v_ = wall_clock64();
#define D4GS_TADD(acc_, a_, b_) acc_ += (b_) - (a_);
#else
#define D4GS_TCLK(v_)
#define D4GS_TADD(acc_, a_, b_)
#endif

__device__ __forceinline__ int xcd_remap(int b, int n_blocks) {
  const int per = (n_blocks + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

#ifdef D4GS_VARIANTS
#include "variants/raster_fwd_variants.inc"
#endif

// ---------------------------------------------------------------------------------------------------------------
// Variant D: variant B's tile / quadrant mapping, but the four 16-lane ROWS of a quadrant-wave are four independent
// 4x4-pixel rasterizers.  On the headline scene 68 % of the (quadrant, splat) pairs variant B replays light at most
// 8 of the 64 lanes (scripts/pair_stats.py): splats are small.  Here every wave compacts, per staged batch, one list
// of staged indices per row (the splats whose tight alpha >= 1/255 box touches that row's 4x4 block; ballot + mbcnt,
// order preserved) and the rows walk their own lists in lock-step: iteration i composites up to four different splats.
// Per-pixel arithmetic and order are unchanged -> bit-identical to variants A and B.
// ---------------------------------------------------------------------------------------------------------------
// A staged splat's tight alpha >= 1/255 box, relative to the tile origin, as 4 x f16 (8 bytes instead of 16: the batch
// arrays fit 20 KB and 8 workgroups share a CU).  Only the range around the tile matters, so the bounds are clamped to
// [-32, 48] (f16 spacing <= 1/32 there) and widened by 0.04 px before the round-to-nearest conversion: the packed box
// always contains the exact one.  It is a filter - the per-pixel alpha test decides - so the image does not change.
template <bool PACKED> struct BoxT;
template <> struct BoxT<true> {
  typedef uint2 type;
  static __device__ __forceinline__ uint2 pack(float x0, float x1, float y0, float y1) { return d4gs_pack_box(x0, x1, y0, y1); }
  static __device__ __forceinline__ float4 unpack(uint2 p) { return d4gs_unpack_box(p); }
};
template <> struct BoxT<false> {
  typedef float4 type;
  static __device__ __forceinline__ float4 pack(float x0, float x1, float y0, float y1) { return make_float4(x0, x1, y0, y1); }
  static __device__ __forceinline__ float4 unpack(float4 p) { return p; }
};

// 16 x byte B of a dword in one instruction (SDWA byte select on the shift's operand; the compiler emits v_bfe_u32 + v_lshlrev_b32)
template <int B>
__device__ __forceinline__ int d4gs_byte_x16(uint32_t w) {
  int r;
  if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(4u), "v"(w));
  if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(4u), "v"(w));
  if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(4u), "v"(w));
  if constexpr (B == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(4u), "v"(w));
  return r;
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int D, bool DEPTH, bool SEG = false>
__device__ __forceinline__ void raster_fwd_r_body(const RasterFwdArgs &a) {
#pragma clang fp contract(off)
  constexpr int NCH = D + (DEPTH ? 1 : 0);
  constexpr int DP = (D + 3) & ~3;
  constexpr int DV = DP / 4;
  constexpr int FB = (DV > 2 || (SEG && D4GS_SEG_UNIT < 256)) ? 128 : 256;  // splats per batch (indices fit one byte); wide colour records: smaller
                                           // batches keep 4+ workgroups per CU (measured 17-ch: 0.45 -> 0.42 ms)
  // 16 colour channels (the reference's 17-channel training renders, round 6): the per-step update acc[c] += colour[c] * vis of
  // channels 0..15 leaves the VALU.  It is an outer product per QUAD of lanes - the four lanes of a quad sit in one 16-lane row,
  // i.e. composite the SAME splat - which is what v_mfma_f32_4x4x1_16b_f32 computes: lane l supplies A[i = l & 3] and B[j = l & 3]
  // of block l >> 2 and receives D[i][l & 3] += A[i] * B[l & 3] in register i (scripts/microbench/mfma4x4.hip).  With A = colour
  // channel 4 (l & 3) + m of the row's splat and B = the lane's own visibility, four of them (m = 0..3; 8 cycles each on the matrix
  // pipe) replace sixteen v_fma_f32 (64+ cycles of the VALU pipe that bounds this kernel), bit for bit: one fused multiply-add
  // per element either way.  Lane i of a quad reads float4 i of the staged record (channels 4 i .. 4 i + 3) - ONE 16-byte LDS read
  // instead of four (the 64-byte record read by every lane was 2/3 of the step's LDS return traffic) - and MFMA m takes its
  // component m: accm[m][i] = channel 4 i + m.
#ifndef D4GS_FWD_MFMA
#define D4GS_FWD_MFMA 1
#endif
  constexpr bool MF = D4GS_FWD_MFMA && D == 16;
  // One-record colours, unsegmented lists (round 5): slot 0 of every batch is a NULL record (opacity 0, empty box) and a batch stages
  // FB - 1 splats into slots 1 .. FB - 1.  The rows' lists start out as zeros, so the steps the lock-step loop runs past a row's own
  // count composite the null record - no per-step "is this row still active" compare.  (Segment boundaries are multiples of the
  // batch size: the segmented variants keep FB splats per batch and the compare.)
#ifndef D4GS_FWD_NULL0
#define D4GS_FWD_NULL0 1
#endif
#ifndef D4GS_FWD_WIDE4
#define D4GS_FWD_WIDE4 1  // the matrix-pipe kernel also takes the four-steps-per-read loop and the null record (round 6)
#endif
  constexpr bool LIKE_NARROW = DV == 1 || (MF && D4GS_FWD_WIDE4);
  constexpr bool NULL0 = D4GS_FWD_NULL0 && LIKE_NARROW && !SEG;
  constexpr int FBE = NULL0 ? FB - 1 : FB;  // splats per batch
  __shared__ float4 sg0[FB];
  __shared__ float4 sg1[FB];
  __shared__ typename BoxT<(DV <= 1)>::type sbox[FB];  // tight box in tile-local pixels
  __shared__ float4 scol[FB * DV];
  __shared__ __attribute__((aligned(16))) unsigned char slist[4 * 4 * FB];  // [wave][row][position]

  D4GS_TCLK(_t0)
#ifdef D4GS_TRACE
  unsigned long long _ts = 0, _tl = 0, _tc = 0, _ti = 0;
  const unsigned long long _c0 = __builtin_amdgcn_s_memtime();  // shader clock (wall_clock64 = the constant 100 MHz one)
#endif
  if (a.n_dev && (a.n_dev[0] > a.cap || (a.max_hint > 0 && a.n_dev[1] > a.max_hint))) return;
  const int n_tiles_s = a.tw * a.th;
  const int n_tiles = a.S * n_tiles_s;
  const int t = xcd_remap(blockIdx.x, n_tiles);
  if (t >= n_tiles) return;
  const int s = t / n_tiles_s, tl = t - s * n_tiles_s;
  const int ty = tl / a.tw, tx = tl - ty * a.tw;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = lane >> 4, kq = lane & 15;
  const int qx0 = tx * D4GS_TILE + (wv & 1) * 8, qy0 = ty * D4GS_TILE + (wv >> 1) * 8;
  const int x = qx0 + (row & 1) * 4 + (kq & 3), y = qy0 + (row >> 1) * 4 + (kq >> 2);
  const bool inside = x < a.width && y < a.height;
  const float pxf = (float)x + 0.5f, pyf = (float)y + 0.5f;
  // pixel-centre extents of the two block columns / block rows of this quadrant
  // (packed boxes are relative to the tile origin; the unpacked ones stay in image coordinates)
  constexpr bool PB = DV <= 1;  // boxes packed as 4 x f16 where that buys the 8th workgroup per CU (see pack_box)
  const float tx0f = PB ? (float)(tx * D4GS_TILE) : 0.f, ty0f = PB ? (float)(ty * D4GS_TILE) : 0.f;
  const float qlx = (float)qx0 - tx0f, qly = (float)qy0 - ty0f;
  const float xl0 = qlx + 0.5f, xh0 = qlx + 3.5f, xl1 = qlx + 4.5f, xh1 = qlx + 7.5f;
  const float yl0 = qly + 0.5f, yh0 = qly + 3.5f, yl1 = qly + 4.5f, yh1 = qly + 7.5f;
  unsigned char *wlist = slist + wv * 4 * FB;
  const unsigned char *mylist = wlist + row * FB;

  float T = 1.f, acc[MF ? (NCH - 16 > 0 ? NCH - 16 : 1) : NCH];  // MF: channels >= 16 only (the depth channel)
  f32x4_t accm[MF ? 4 : 1];                                         // MF: accm[m][i] = channel 4 i + m
  int last = 0;
  bool done = !inside, has_last = false;
#pragma unroll
  for (int c = 0; c < (MF ? NCH - 16 : NCH); c++) acc[c] = 0.f;
#pragma unroll
  for (int m = 0; m < (MF ? 4 : 1); m++) accm[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto chan = [&](int c) -> float {  // accumulated channel c (compile-time c: the callers' loops are unrolled)
    if constexpr (MF) return c < 16 ? accm[c & 3][c >> 2] : acc[c - 16];
    else return acc[c];
  };

  const int start = a.tile_offsets[t];
  int end = a.tile_offsets[t + 1];
  const int end_full = end;
  bool has_far = false;  // workgroup-uniform
  // lazy_flag[t]: 0 - the near part was enough (so far), 1 - pass 1 found the tile unsaturated: its far part is to be emitted and
  // sorted, 2 - that has been done (by an earlier d4gs_raster_fwd over the same binning: the channel chunks of a wide render share the
  // lists, and the far keys must be emitted exactly once).  A tile flagged by an earlier call goes straight to pass 2.
  if (a.lazy_pass == 1) {
    const int nn = a.lazy_near[t];
    has_far = nn < end - start;
    if (has_far && a.lazy_flag[t]) return;
    end = min(end, start + nn);
  } else if (a.lazy_pass == 2 && !a.lazy_flag[t]) return;
  const size_t inst_base = (size_t)s * a.N;
  // SEG: the pixel's state (T, accumulated channels) is stored at every depth-segment boundary of the list and at its end, in
  // tile-local pixel order, for the segmented backward (raster_bwd.hip).  Slot 0 = final state, slot k = after k * seglen entries.
  const int seglen = SEG ? d4gs_seg_len(end_full - start) : 0;  // (of the WHOLE list, as the backward computes it - a lazy first pass
  //                                                                  composites a truncated one)
  float *seg_px = SEG ? a.seg_state + (size_t)t * D4GS_SEG_MAX * (1 + NCH) * 256 + ((y - ty * D4GS_TILE) * D4GS_TILE + (x - tx * D4GS_TILE)) : nullptr;
  auto seg_store = [&](int k) {
    if (inside) {
      float *p = seg_px + (size_t)k * (1 + NCH) * 256;
      p[0] = T;
#pragma unroll
      for (int c = 0; c < NCH; c++) p[(1 + c) * 256] = chan(c);
    }
  };
  for (int b = start; b < end; b += FBE) {
    if (__syncthreads_and(done)) break;  // also orders the previous batch's LDS reads before restaging
    D4GS_TCLK(_ta)
    const int idx = b + tid - (NULL0 ? 1 : 0);
    if (NULL0 && tid == 0) {
      sg0[0] = make_float4(0.f, 0.f, 0.f, 0.f);  // opacity 0: alpha = 0 < 1/255 at every pixel
      sg1[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      sbox[0] = BoxT<PB>::pack(1e30f, -1e30f, 1e30f, -1e30f);
#pragma unroll
      for (int v = 0; v < DV; v++) scol[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (tid < FB && idx < end) {
      const int gid = a.sorted_gid[idx];
      const float4 *gp = reinterpret_cast<const float4 *>(a.geom + (inst_base + gid) * D4GS_GEOM_STRIDE);
      const float4 q0 = gp[0], q1 = gp[1];
      sg0[tid] = q0;
      sg1[tid] = stage_conic(q1.x, q1.y, q1.z, 0.f);
      const float tau = __logf(255.f * q0.z) * 1.01f + 0.02f;
      const float det = q1.x * q1.z - q1.y * q1.y;
      const float idet = 1.f / det;
      float ex = -1.f, ey = -1.f;
      if (tau > 0.f && det > 0.f) {
        ex = sqrtf(2.f * tau * q1.z * idet) + 1e-3f;
        ey = sqrtf(2.f * tau * q1.x * idet) + 1e-3f;
      }
      sbox[tid] = ex < 0.f ? BoxT<PB>::pack(1e30f, -1e30f, 1e30f, -1e30f) : BoxT<PB>::pack(q0.x - ex - tx0f, q0.x + ex - tx0f, q0.y - ey - ty0f, q0.y + ey - ty0f);
      const float4 *cp = reinterpret_cast<const float4 *>(a.ctab + (size_t)gid * DP);
#pragma unroll
      for (int v = 0; v < DV; v++) scol[tid * DV + v] = cp[v];
    }
    __syncthreads();
    D4GS_TCLK(_tb)
    D4GS_TADD(_ts, _ta, _tb)
    const int nb = min(FBE, end - b) + (NULL0 ? 1 : 0);  // slots in use
    // ---- per-row lists of this wave's quadrant ----
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;  // wave-uniform list lengths
    // The wave's four lists start out as zeros (one 16-byte store per lane and batch): the composite loop reads list bytes past a
    // row's count (no divergent branch in it) and takes them as they are - staged splat 0 exists in every batch, a stale index could
    // point at a never-staged record (NaN * 0) - instead of replacing them lane by lane in every step.
    constexpr bool NARROW = LIKE_NARROW;  // one colour record (or the matrix-pipe accumulators): the four-steps-per-read loop below; other wide records keep the one-byte loop (regs)
    if constexpr (NARROW) {
      if (lane * 16 < 4 * FB) reinterpret_cast<uint4 *>(wlist)[lane] = make_uint4(0u, 0u, 0u, 0u);
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < FB / 64; k++) {
      const int jj = k * 64 + lane;
      bool h0 = false, h1 = false, h2 = false, h3 = false;
      if (jj < nb) {
        const float4 bx = BoxT<PB>::unpack(sbox[jj]);
        const bool X0 = (bx.x <= xh0) && (bx.y >= xl0), X1 = (bx.x <= xh1) && (bx.y >= xl1);
        const bool Y0 = (bx.z <= yh0) && (bx.w >= yl0), Y1 = (bx.z <= yh1) && (bx.w >= yl1);
        h0 = X0 && Y0, h1 = X1 && Y0, h2 = X0 && Y1, h3 = X1 && Y1;
      }
      const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
      const unsigned char jb = (unsigned char)jj;
      if (h0) wlist[0 * FB + c0 + __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u))] = jb;
      if (h1) wlist[1 * FB + c1 + __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u))] = jb;
      if (h2) wlist[2 * FB + c2 + __builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0u))] = jb;
      if (h3) wlist[3 * FB + c3 + __builtin_amdgcn_mbcnt_hi((unsigned)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m3, 0u))] = jb;
      c0 += __popcll(m0), c1 += __popcll(m1), c2 += __popcll(m2), c3 += __popcll(m3);
    }
    __builtin_amdgcn_wave_barrier();
    D4GS_TCLK(_tc0)
    D4GS_TADD(_tl, _tb, _tc0)
    const int cnt = row == 0 ? c0 : row == 1 ? c1 : row == 2 ? c2 : c3;
    const int imax = max(max(c0, c1), max(c2, c3));
    int last16 = -1;  // 16 x the slot of the batch's last contributor to this pixel
    unsigned long long donem = __builtin_amdgcn_ballot_w64(done);  // the saturated pixels of this wave as a lane mask (all 64 lanes are active here)
    // One composite step of this lane's row: staged splat `jr` (ignored past the row's count).
    auto step = [&](const unsigned long long actm, const int j16 /* 16 x the staged splat's slot: the byte offset of its records */) {
      const float4 g0 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sg0) + j16);
      const float4 g1 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sg1) + j16);
      const float dx = g0.x - pxf, dy = g0.y - pyf;
      const float sigma = splat_sigma2(g1, dx, dy);  // sigma * log2(e)
      const float alpha = fminf(0.999f, g0.z * __builtin_amdgcn_exp2f(-sigma));
      const float nT = T * (1.f - alpha);  // (as one FMA, T - alpha T: measured 1 % of this kernel, not worth leaving the reference's rounding)
      // The lane masks are handled as what they are - 64-bit scalars: ONE compare against the saturation threshold feeds both the
      // composite (as it is) and `done` (its complement, a scalar and-not).  Written with bools the compiler issues that compare twice
      // (v_cmp_ge + v_cmp_nge / v_cmp_lt + v_cmp_nlt).
      // (one ballot per compare: the ballot of a conjunction is lowered through a VGPR - v_cndmask + v_cmp_ne)
      const unsigned long long vm = (NULL0 ? ~0ull : actm) & __builtin_amdgcn_ballot_w64(sigma >= 0.f) & __builtin_amdgcn_ballot_w64(alpha >= (1.f / 255.f)) & ~donem;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(nT > 1e-4f);  // (hip's __ballot goes through an int: v_cndmask + v_cmp)
      donem |= vm & ~mm;
      const bool valid = __builtin_amdgcn_inverse_ballot_w64(vm & mm);
      const float vis = valid ? alpha * T : 0.f;
      if constexpr (MF) {
        const float4 ct = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(scol) + j16 * DV + 16 * (lane & 3));
        accm[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.x, vis, accm[0], 0, 0, 0);
        accm[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.y, vis, accm[1], 0, 0, 0);
        accm[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.z, vis, accm[2], 0, 0, 0);
        accm[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.w, vis, accm[3], 0, 0, 0);
        if (DEPTH) acc[0] = __builtin_fmaf(g0.w, vis, acc[0]);
      } else {
#pragma unroll
        for (int v = 0; v < DV; v++) {
          const float4 c4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(scol) + j16 * DV + 16 * v);
          if (v * 4 < D) acc[v * 4] = __builtin_fmaf(c4.x, vis, acc[v * 4]);
          if (v * 4 + 1 < D) acc[v * 4 + 1] = __builtin_fmaf(c4.y, vis, acc[v * 4 + 1]);
          if (v * 4 + 2 < D) acc[v * 4 + 2] = __builtin_fmaf(c4.z, vis, acc[v * 4 + 2]);
          if (v * 4 + 3 < D) acc[v * 4 + 3] = __builtin_fmaf(c4.w, vis, acc[v * 4 + 3]);
        }
        if (DEPTH) acc[D] = __builtin_fmaf(g0.w, vis, acc[D]);
      }
      T = valid ? nT : T;
      last16 = valid ? j16 : last16;
    };
#ifndef D4GS_FWD_LIST4
#define D4GS_FWD_LIST4 1
#endif
    if constexpr (D4GS_FWD_LIST4 && NARROW) {
      // Four list positions per LDS read (the rows' lists start on 4-byte boundaries, FB is a multiple of 4): one address add and one read
      // per four steps; up to three steps past the longest list run with no row active.  Measured against the one-byte loop (cfg2,
      // profiles/r05_ab_fwd_list4.txt): 267 -> 238 us; with wide records the four unrolled steps cost the occupancy (17 channels:
      // 80 -> 126 VGPRs), so those keep the loop below.
      for (int i0 = 0; i0 < imax; i0 += 16) {  // every 16 iterations: is the whole wave saturated?
        const int i1 = min(i0 + 16, imax);
        for (int i = i0; i < i1; i += 4) {
          const uint32_t w4 = *reinterpret_cast<const uint32_t *>(mylist + i);
          step(NULL0 ? 0ull : __builtin_amdgcn_ballot_w64(i < cnt), d4gs_byte_x16<0>(w4));
          step(NULL0 ? 0ull : __builtin_amdgcn_ballot_w64(i + 1 < cnt), d4gs_byte_x16<1>(w4));
          step(NULL0 ? 0ull : __builtin_amdgcn_ballot_w64(i + 2 < cnt), d4gs_byte_x16<2>(w4));
          step(NULL0 ? 0ull : __builtin_amdgcn_ballot_w64(i + 3 < cnt), d4gs_byte_x16<3>(w4));
        }
        if (donem == ~0ull) break;
      }
      done = __builtin_amdgcn_inverse_ballot_w64(donem);
    } else {
      for (int i0 = 0; i0 < imax; i0 += 16) {
        const int i1 = min(i0 + 16, imax);
        for (int i = i0; i < i1; i++) {
          const bool act = i < cnt;
          // the byte is read unconditionally (no divergent branch in the loop; i < FB stays inside the row's list) and
          // replaced by staged splat 0 past the row's count: a stale index could point at a never-staged record (NaN * 0)
          const int jr = (int)mylist[i];
          const int j = act ? jr : 0;
          const float4 g0 = sg0[j], g1 = sg1[j];
          const float dx = g0.x - pxf, dy = g0.y - pyf;
          const float sigma = splat_sigma2(g1, dx, dy);  // sigma * log2(e)
          const float alpha = fminf(0.999f, g0.z * __builtin_amdgcn_exp2f(-sigma));
          bool valid = act && !done && (sigma >= 0.f) && (alpha >= (1.f / 255.f));
          const float nT = T * (1.f - alpha);
          const bool sat = nT <= 1e-4f;
          done = done || (valid && sat);
          valid = valid && !sat;
          const float vis = valid ? alpha * T : 0.f;
          if constexpr (MF) {
            const float4 ct = scol[j * 4 + (lane & 3)];  // channels 4 i .. 4 i + 3, i = this lane's place in its quad
            accm[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.x, vis, accm[0], 0, 0, 0);
            accm[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.y, vis, accm[1], 0, 0, 0);
            accm[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.z, vis, accm[2], 0, 0, 0);
            accm[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(ct.w, vis, accm[3], 0, 0, 0);
            if (DEPTH) acc[0] = __builtin_fmaf(g0.w, vis, acc[0]);
          } else {
#pragma unroll
            for (int v = 0; v < DV; v++) {
              const float4 c4 = scol[j * DV + v];
              if (v * 4 < D) acc[v * 4] = __builtin_fmaf(c4.x, vis, acc[v * 4]);
              if (v * 4 + 1 < D) acc[v * 4 + 1] = __builtin_fmaf(c4.y, vis, acc[v * 4 + 1]);
              if (v * 4 + 2 < D) acc[v * 4 + 2] = __builtin_fmaf(c4.z, vis, acc[v * 4 + 2]);
              if (v * 4 + 3 < D) acc[v * 4 + 3] = __builtin_fmaf(c4.w, vis, acc[v * 4 + 3]);
            }
            if (DEPTH) acc[D] = __builtin_fmaf(g0.w, vis, acc[D]);
          }
          T = valid ? nT : T;
          last16 = valid ? 16 * j : last16;
        }
        if (__all(done)) break;
      }
    }
    last = last16 >= 0 ? b + (last16 >> 4) - (NULL0 ? 1 : 0) : last;  // list index of the batch's last contributor, formed once per batch
    has_last = has_last || last16 >= 0;
    D4GS_TCLK(_tc1)
    D4GS_TADD(_tc, _tc0, _tc1)
#ifdef D4GS_TRACE
    _ti += (unsigned long long)imax;
#endif
    if constexpr (SEG) {
      const int e = b + FB - start;  // entries composited so far (a boundary is never the end of the list: slot 0 holds that)
      if (e < end - start && e % seglen == 0) seg_store(e / seglen);
    }
  }
  if (has_far) {  // lazy pass 1: did every pixel of the tile saturate inside the near part?  If not, the tile is done again in pass 2
    if (!__syncthreads_and(done)) {
      if (tid == 0) a.lazy_flag[t] = 1;
      return;
    }
  }
  if (a.lazy_pass == 2 && tid == 0) a.lazy_flag[t] = 2;  // (workgroup t is the only reader and writer of flag[t] in this launch)
  if constexpr (SEG) seg_store(0);

  // live-row sample (include/d4gs.h, D4gsProjOut.n_isect[2..3]): every `stride`-th tile adds its list length and the entries
  // up to its last contributor - at most 128 tiles, so the device-scope atomics never queue up
  const int lstride = (n_tiles + 127) >> 7;
  if (a.n_dev && t % lstride == 0) {  // workgroup-uniform
    __shared__ int slive_hi[4];
    int hi = (inside && last >= start && has_last) ? last : start - 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hi = max(hi, __shfl_xor(hi, o));
    if (lane == 0) slive_hi[wv] = hi;
    __syncthreads();
    if (tid == 0) {
      const int h = max(max(slive_hi[0], slive_hi[1]), max(slive_hi[2], slive_hi[3]));
      atomicAdd(reinterpret_cast<unsigned long long *>(a.n_dev + 2), (unsigned long long)(end_full - start));
      atomicAdd(reinterpret_cast<unsigned long long *>(a.n_dev + 3), (unsigned long long)(h - start + 1));
    }
  }
  if (inside) {
    const size_t pix = ((size_t)s * a.height + y) * a.width + x;
    const float al = 1.f - T;
    a.alphas[pix] = al;
    a.final_T[pix] = T;
    a.last_ids[pix] = last;
    float *o = a.out + pix * NCH;
#pragma unroll
    for (int c = 0; c < D; c++) o[c] = chan(c) + (a.background ? T * a.background[c] : 0.f);
    if (DEPTH) o[D] = a.ed ? chan(D) / fmaxf(al, 1e-10f) : chan(D);
  }
#ifdef D4GS_TRACE
  if (a.trace && tid == 0) {
    unsigned long long *tr = a.trace + (size_t)blockIdx.x * 10;
    tr[8] = __builtin_amdgcn_s_memtime() - _c0;
    tr[0] = _t0, tr[1] = wall_clock64(), tr[3] = (unsigned long long)(end - start), tr[4] = _ts, tr[5] = _tl, tr[6] = _tc, tr[7] = _ti;
    tr[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
  }
#endif
}
// Two entry points over one body: the narrow instantiations (D <= 4) are asked for 8 waves per SIMD (the hint changes the
// scheduler's register budget); the wide ones keep the compiler's default - the hint cannot be met there and only perturbs them.
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_raster_fwd_r8(const RasterFwdArgs a) {
  raster_fwd_r_body<D, DEPTH>(a);
}
// (D = 16, the matrix-pipe accumulators with the four-steps-per-read loop: asked for 6 waves per SIMD the kernel fits 76 VGPRs without
// scratch; left alone the compiler takes 96 + 16 -> 4 waves)
#ifndef D4GS_FWD_WIDE_WAVES
#define D4GS_FWD_WIDE_WAVES 6
#endif
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256)
#if D4GS_FWD_WIDE_WAVES > 0
__attribute__((amdgpu_waves_per_eu(D == 16 ? D4GS_FWD_WIDE_WAVES : 1, D == 16 ? D4GS_FWD_WIDE_WAVES : 10)))
#endif
k_raster_fwd_r(const RasterFwdArgs a) {
  raster_fwd_r_body<D, DEPTH>(a);
}
// the same two, also storing the segment-boundary states (few-tile launches; the kernels above stay as they are)
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_raster_fwd_rs8(const RasterFwdArgs a) {
  raster_fwd_r_body<D, DEPTH, true>(a);
}
template <int D, bool DEPTH>
__global__ void __launch_bounds__(256)
#if D4GS_FWD_WIDE_WAVES > 0
__attribute__((amdgpu_waves_per_eu(D == 16 ? D4GS_FWD_WIDE_WAVES : 1, D == 16 ? D4GS_FWD_WIDE_WAVES : 10)))
#endif
k_raster_fwd_rs(const RasterFwdArgs a) {
  raster_fwd_r_body<D, DEPTH, true>(a);
}

template <int D, bool DEPTH>
int launch_fwd(const RasterFwdArgs &a, hipStream_t stream) {
  const int n_tiles = a.S * a.tw * a.th;
  const int blocks = ((n_tiles + 7) / 8) * 8;
#ifdef D4GS_VARIANTS  // the A/B build only (tests/libd4gs_variants.so): environment-selected reference variants
  static const bool wave_per_tile = getenv("D4GS_FWD_WAVE_PER_TILE") != nullptr;  // variant A
  static const bool quads = getenv("D4GS_FWD_QUADS") != nullptr;                  // variant B
  if (wave_per_tile) {
    D4GS_LAUNCH("k_raster_fwd", (k_raster_fwd<D, DEPTH>), dim3(blocks), dim3(64), 0, stream, a);
    return d4gs_check_launch("k_raster_fwd");
  }
  if (quads) {
    D4GS_LAUNCH("k_raster_fwd_q", (k_raster_fwd_q<D, DEPTH>), dim3(blocks), dim3(256), 0, stream, a);
    return d4gs_check_launch("k_raster_fwd_q");
  }
#endif
  if (a.seg_state) {
    if constexpr (D <= 4) D4GS_LAUNCH("k_raster_fwd_r", (k_raster_fwd_rs8<D, DEPTH>), dim3(blocks), dim3(256), 0, stream, a);
    else D4GS_LAUNCH("k_raster_fwd_r", (k_raster_fwd_rs<D, DEPTH>), dim3(blocks), dim3(256), 0, stream, a);
    return d4gs_check_launch("k_raster_fwd_r");
  }
  static const int q_env = getenv("D4GS_FWD_Q") ? atoi(getenv("D4GS_FWD_Q")) : 0;  // A/B hook: workgroups per CU
  if constexpr (D <= 4) {
    const int pad = d4gs_lds_pad_for_wgs_per_cu((const void *)k_raster_fwd_r8<D, DEPTH>, q_env);
    D4GS_LAUNCH("k_raster_fwd_r", (k_raster_fwd_r8<D, DEPTH>), dim3(blocks), dim3(256), pad, stream, a);
  } else {
    const int pad = d4gs_lds_pad_for_wgs_per_cu((const void *)k_raster_fwd_r<D, DEPTH>, q_env);
    D4GS_LAUNCH("k_raster_fwd_r", (k_raster_fwd_r<D, DEPTH>), dim3(blocks), dim3(256), pad, stream, a);
  }
  return d4gs_check_launch("k_raster_fwd_r");
}

}  // namespace

// ---- depth segments: when, and how much state (common.h) ----
int64_t d4gs_seg_state_elems(const D4gsDims *d) {
  const int64_t tw = (d->width + D4GS_TILE - 1) / D4GS_TILE, th = (d->height + D4GS_TILE - 1) / D4GS_TILE;
  const int64_t n_tiles = (int64_t)d->S * tw * th;
  if (n_tiles > (d->D <= 4 ? D4GS_SEG_TILES_MAX_NARROW : D4GS_SEG_TILES_MAX)) return 0;
  const int64_t nch = d->D + (d->depth_mode != D4GS_DEPTH_NONE ? 1 : 0);
  return n_tiles * D4GS_SEG_MAX * (1 + nch) * 256;
}
bool d4gs_seg_on(const D4gsDims *d, const D4gsIsect *isect, const D4gsRaster *r) {
  const char *env = getenv("D4GS_SEG");  // (read per call: tests toggle it) "0": never, "1": whenever the buffer is there (tests); default: see below
  if (!r->seg_state || d4gs_seg_state_elems(d) == 0 || (env && env[0] == '0')) return false;
#ifdef D4GS_VARIANTS  // the environment-selected reference variants neither write nor read the boundary states
  if (getenv("D4GS_FWD_WAVE_PER_TILE") || getenv("D4GS_FWD_QUADS") || getenv("D4GS_BWD_WAVE_PER_TILE") || getenv("D4GS_BWD_MFMA"))
    return false;
#endif
  if (env && env[0] == '1') return true;
  // worth it only when some list spans more than one segment; the longest list is known from the previous render of the
  // shape (D4gsIsect.max_tile_count; <= 0 = unknown).  Forward and backward of one render see the same D4gsIsect.
  return isect->max_tile_count > D4GS_SEG_UNIT;
}

int d4gs_raster_fwd_impl(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, const D4gsRaster *r,
                         hipStream_t stream) {
  RasterFwdArgs a;
  a.N = dims->N, a.S = dims->S, a.width = dims->width, a.height = dims->height;
  a.tw = (dims->width + D4GS_TILE - 1) / D4GS_TILE;
  a.th = (dims->height + D4GS_TILE - 1) / D4GS_TILE;
  a.ed = dims->depth_mode == D4GS_DEPTH_ED;
  a.geom = proj->geom, a.ctab = proj->ctab, a.background = r->background;
  a.tile_offsets = proj->tile_offsets, a.sorted_gid = isect->sorted_gid;
  a.n_dev = proj->n_isect, a.cap = isect->n_isect, a.max_hint = isect->max_tile_count;
  a.out = r->render_colors, a.alphas = r->render_alphas, a.last_ids = r->last_ids, a.final_T = r->final_T;
  a.seg_state = d4gs_seg_on(dims, isect, r) ? r->seg_state : nullptr;
  a.lazy_near = nullptr, a.lazy_flag = nullptr, a.lazy_pass = 0;
  const bool lazy = d4gs_lazy_on(dims, proj) && isect->n_isect > 0;
  if (lazy) {
    const LazyWs lw = d4gs_lazy_carve(proj->lazy_ws, dims->S, a.tw * a.th);
    a.lazy_near = lw.near, a.lazy_flag = lw.flag, a.lazy_pass = 1;
  }
#ifdef D4GS_TRACE
  a.trace = getenv("D4GS_TRACE_FWD_PTR") ? (unsigned long long *)strtoull(getenv("D4GS_TRACE_FWD_PTR"), nullptr, 0) : nullptr;
#endif
  const bool dep = dims->depth_mode != D4GS_DEPTH_NONE;
  auto launch = [&](const RasterFwdArgs &aa) -> int {
#define D4GS_CASE(DD)                                                   \
  case DD:                                                              \
    return dep ? launch_fwd<DD, true>(aa, stream) : launch_fwd<DD, false>(aa, stream);
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
  };
  int rc = launch(a);
  if (rc || !lazy) return rc;
  if ((rc = d4gs_lazy_far_sort(dims, proj, isect, stream))) return rc;
  a.lazy_pass = 2;
  return launch(a);
}
