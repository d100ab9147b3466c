// project_bwd.hip -- adjoint of project_fwd.hip: projection, camera delta, motion-basis deformation and the
// activations, for ALL S sub-samples, reduced to leaf gradients; and (MODE_POSES) the adjoint of d4gs_poses_fwd.
//
// Replaces gsplat fully_fused_projection_bwd and the torch autograd of flow3d/scene_model.py:67-120,352-353,
// flow3d/params.py:39-43,142-180, flow3d/transforms.py:41-53 (reference backward: flow3d/trainer.py:231).
//
// Work decomposition (round 3): a 256-lane block owns 64 Gaussians; its 4 waves are 4 sub-sample SLOTS.  Lane l of
// wave w evaluates the adjoint chain of instance (Gaussian g0 + l, sub-sample s) for s = w, w + 4, w + 8, ...  So
//   * the sub-sample is WAVE-UNIFORM: its time-blended bases are one small LDS slab per wave (any S; round 2 kept all
//     S*K*9 of them resident), its camera delta is scalar loads, per-instance loads are fully coalesced;
//   * a shared gradient of sub-sample s (bases, camera delta) is the reduction of ONE wave: the wave ladder writes the
//     block partial directly - no cross-wave sum, no block barrier inside the sub-sample loop;
//   * per-Gaussian leaf gradients are accumulated over a slot's sub-samples in registers (fixed order), the 4 slots are
//     then added through LDS in fixed order ((0+1)+(2+3)) and written once - no atomics, bit-reproducible.
// Small S (round 4; a rank of an exposure-sharded frame renders S / P sub-samples): with fewer than 4 sub-samples the slots
// beyond S would idle (S = 1: three of the four waves).  The block then owns 64 * NSG Gaussians, NSG = 4 / SL SUB-GROUPS of SL
// slots each (SL = 1 for S = 1, 2 for S = 2, 4 otherwise): wave w is slot w % SL of sub-group w / SL, everything above holds per
// sub-group, and a sub-group leaves its own shared-gradient partial vector.  S >= 3 is the round-3 mapping, bit for bit.
// (Round 2: one lane per Gaussian looping over S with 27 accumulators at 253 VGPRs, 2 waves / SIMD, 4 688 waves for
// 2 048 wave slots on cfg2.)  Block partials of the shared gradients are summed over blocks by k_reduce_partials in a
// fixed order and scattered to rots / transls / times by k_finish.
// (wave_sum_store's packed ladder - common.h - is for the composite backward; here its four s_nop-separated steps sit on the critical
// path of a 4-wave kernel: measured 106.7 -> 108.7 us on cfg2, so this file keeps the three-register ladder)
#ifndef D4GS_PACKED_LADDER
#define D4GS_PACKED_LADDER 0
#endif
#include <type_traits>
#include "common.h"

namespace {

constexpr int BLK = 256;
constexpr int GPB = 64;   // Gaussians per block = lanes of a wave
constexpr int SLOTS = 4;  // sub-sample slots = waves of a block
// per-lane accumulators kept in LDS (read-modify-write by their owner lane only): v_mu 3, v_q 4, v_sc 3, v_view 12 (+1: odd
// stride).  The K > 8 variant is LDS-bound in occupancy (coefficient gradients [256][K]), so it keeps v_view in registers.
constexpr int nacc_of(bool mfma) { return mfma ? 11 : 23; }
enum { MODE_RENDER = 0, MODE_POSES = 1 };

struct BwdArgs {
  D4gsDims d;
  D4gsProjIn in;
  const int32_t *radii;
  const float *conics;
  const float *ctab;
  const float *opac_act;
  const float *v_means2d, *v_conics, *v_depths, *v_opac_act, *v_ctab;
  D4gsLeafGrads g;
  int n_shared;  // S*(9K+12)+12
  // MODE_POSES (adjoint of d4gs_poses_fwd / d4gs_points_fwd); any of the three may be NULL = zero gradient
  const float *v_points;      // [S,N,3] | [N,S,3]
  const float *v_quats_out;   // [S,N,4] | [N,S,4]
  const float *v_transforms;  // [S,G,3,4] | [G,S,3,4]
  int g_major;                // 1: the Gaussian-major layouts (the reference's (G,B,...) tensors)
  int persist;                // 1: a fixed grid of blocks walks the 64-Gaussian groups and keeps its shared-gradient sums in LDS
  const float *btab;          // [S][K][16] time-blended bases of every sub-sample (D4gsProjOut.blend_bases, written by k_project_fwd; or
                              // k_bases_table here, for callers without it), read with scalar loads
};
// slots per sub-group for S sub-samples (the poses adjoint keeps the 4-slot mapping: it is not on the per-frame path)
template <int MODE>
int slots_of(int S) { return (MODE != MODE_RENDER || S >= 3) ? 4 : S == 2 ? 2 : 1; }

#ifndef PB_MFMA_WAVES
#define PB_MFMA_WAVES 3
#endif
#ifndef PB_NO_SB
#define SB() __builtin_amdgcn_sched_barrier(0)
#else
#define SB()
#endif
// accumulate into the lane's private LDS row: plain read-modify-write (ds_add_f32, 22 per pass, measured 2.6x slower for the
// whole kernel: 306 vs 117 us on cfg2 - LDS float atomics run at a fraction of the plain DS rate on gfx950)
#define ACC(slot_, val_) ac[slot_] += (val_)
#define VIEW(i_, val_)                     \
  do {                                     \
    if constexpr (MFMA) v_view_r[i_] += (val_); \
    else ac[10 + (i_)] += (val_);           \
  } while (0)

// v_q += (d R(q) / d q)^T vR for a unit quaternion q = (w, x, y, z)
__device__ __forceinline__ void rotmat_adj_to_quat(const float *q, const float *vR, float *v_q) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  v_q[0] += 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
  v_q[1] += 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[3] + vR[1]) + z * (vR[6] + vR[2]) + w * (vR[7] - vR[5]));
  v_q[2] += 2.f * (x * (vR[3] + vR[1]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[7] + vR[5]) + w * (vR[2] - vR[6]));
  v_q[3] += 2.f * (x * (vR[6] + vR[2]) + y * (vR[7] + vR[5]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
}

// The blended-bases table (round 5).  The time-blended bases of the S sub-samples are the same numbers for every block, yet every wave
// used to re-blend its sub-sample's [K][9] slab per pass and read it back from LDS twice (54 + 54 broadcast reads at K = 6: two thirds
// of the kernel's LDS instructions, the LDS pipe ~56 % busy).  With the table in global memory the reads are SCALAR loads (wave-uniform
// address, constant address space: s_load_dwordx8 + dword per basis) and the values are SGPR operands of the multiply-adds: no LDS
// traffic, no transient VGPRs (128 VGPRs + a 24-byte scratch frame -> 122, none).  Row stride 16 floats.  Measured (A/B build, same
// run): cfg2 119.5 -> 112.5 us, cfg3 118.5 -> 112.6, refdefault (K = 20) 128 -> 118, cfg5 (K = 12) 967 -> 736 (profiles/r05_ab_bases_table.txt).
// contract(off): the same two products and one sum as k_project_fwd's preblend_bases (and the reference's lerp), bit for bit - the table a
// caller without `blend_bases` gets here must be the one the forward rendered with (tests/test_gpu_exposure.py).
__global__ void __launch_bounds__(256) k_bases_table(const D4gsDims d, const float *times, const float *rots, const float *transls, float *btab) {
#pragma clang fp contract(off)
  const int s = blockIdx.x, K = d.K, T = d.T;
  const float t = times[s];
  const float ff = fminf(fmaxf(floorf(t), 0.f), (float)(T - 1));
  const float cf_ = fminf(fmaxf(ceilf(t), 0.f), (float)(T - 1));
  const float w = t - ff;
  const int f = (int)ff, c = (int)cf_;
  for (int idx = threadIdx.x; idx < K * 16; idx += 256) {
    const int k = idx >> 4, j = idx & 15;
    float v = 0.f;
    if (j < 3) v = (1.f - w) * transls[(k * T + f) * 3 + j] + w * transls[(k * T + c) * 3 + j];
    else if (j < 9) v = (1.f - w) * rots[(k * T + f) * 6 + j - 3] + w * rots[(k * T + c) * 6 + j - 3];
    btab[((size_t)s * K + k) * 16 + j] = v;
  }
}
// A kernel argument read afresh from the kernarg segment where it is used (s_load at the use site).  k_project_bwd sits at the SGPR limit:
// its ~370-byte argument block was loaded at entry and ~130 of those values were parked in VGPR lanes (v_writelane) and restored with
// v_readlane, 16-register tuples at a time - ~170 restores in the per-group epilogue alone.  The epilogue's pointers are read with this.
#ifndef PB_KARG_RELOAD
#define PB_KARG_RELOAD 1
#endif
template <typename T>
__device__ __forceinline__ T pb_karg_at(size_t off) {
  typedef const volatile __attribute__((address_space(4))) T *kp;
  const __attribute__((address_space(4))) char *base = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
  return *(kp)(base + off);
}
#define PB_KARG(f_) (PB_KARG_RELOAD ? pb_karg_at<std::remove_cv_t<std::remove_reference_t<decltype(a.f_)>>>((size_t)((const char *)&(a.f_) - (const char *)&a)) : a.f_)
typedef const __attribute__((address_space(4))) float *cfloat_p;
#define PB_BROW(k_) ((cfloat_p)(uintptr_t)a.btab + ((size_t)s * K + (k_)) * 16)
template <int MODE, bool MFMA /* K > 8: basis-gradient column sums on the matrix pipe */, int SL /* slots per sub-group */>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(MFMA ? PB_MFMA_WAVES : 4))) k_project_bwd(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const D4gsDims &d = a.d;
  const int N = d.N, G = d.G, S = d.S;
  const bool dyn = G > 0;
  const int K = dyn ? d.K : 0;
  const int KP = K | 1;
  const int nk = K * 9, nop = nk + 13;
  const bool shared = dyn || a.in.RTs;
  const int no = nk + 12;
  constexpr int NSG = SLOTS / SL;  // sub-groups per block (SL = 4: one, the round-3 mapping)
  float *cf = smem;                   // [NSG][GPB][KP] softmaxed coefficients of the block's Gaussians
  float *vcf = cf + NSG * GPB * KP;   // [BLK][KP]    their gradients, per slot, summed over the slot's sub-samples
  float *red = vcf + BLK * KP;        // [SLOTS][nop] per-wave reduction slab (+ dump slot)
  constexpr int NACC = nacc_of(MFMA);
  float *accs = red + SLOTS * nop;    // [BLK][NACC]  per-lane leaf accumulators (cross-slot sum at the end)
  float *svec = accs + BLK * NACC;    // [BLK][9]     K > 8 only: (v_transl 3, v_r6 6) of every lane, MFMA B operand
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = wv / SL, slot = wv - sub * SL;  // this wave: sub-sample slot `slot` of sub-group `sub`
  const bool raw = d.flags & D4GS_RAW_PARAMS;
  const bool has_cam = MODE == MODE_RENDER || a.in.viewmat != nullptr;
  Cam cam;
  if (has_cam) cam = load_cam(a.in.viewmat, a.in.Kmat, d.width, d.height);
  // Persistent blocks: block b walks the groups b, b + gridDim.x, ... and adds every group's shared-gradient sums into
  // wacc (each (sub-sample, element) is owned by one wave, the groups come in a fixed order -> deterministic); ONE partial
  // vector per block leaves the kernel instead of one per group (cfg5: 120 MB of partials written and read back).
  float *wacc = svec + (MFMA ? BLK * 9 : 0);  // [NSG][n_shared], persist only
  if (a.persist) {
    for (int o = tid; o < NSG * a.n_shared; o += BLK) wacc[o] = 0.f;
  }
  const int ngroups = (N + NSG * GPB - 1) / (NSG * GPB);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
  const int g = (grp * NSG + sub) * GPB + lane;
  const bool dyn_block = ((grp * NSG + sub) * GPB) < G;  // (wave-uniform: the property of this wave's 64 Gaussians)
  float *cfl = cf + (sub * GPB + lane) * KP;             // this lane's coefficient row
  const bool active = g < N;
  const bool isdyn = active && g < G;

  float mu[3] = {0, 0, 0}, sc[3] = {1, 1, 1}, qh[4] = {1, 0, 0, 0}, inv_qn = 1.f;
  if (active) {
    const float *const i_means = PB_KARG(in.means);
    mu[0] = i_means[g * 3], mu[1] = i_means[g * 3 + 1], mu[2] = i_means[g * 3 + 2];
  }
  const bool need_q = MODE == MODE_RENDER || a.v_quats_out != nullptr;
  if (active && need_q) {
    const float4 q = *reinterpret_cast<const float4 *>(PB_KARG(in.quats) + (size_t)g * 4);
    inv_qn = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    qh[0] = q.x * inv_qn, qh[1] = q.y * inv_qn, qh[2] = q.z * inv_qn, qh[3] = q.w * inv_qn;
  }
  if (MODE == MODE_RENDER && active) {
    const float *const i_scales = PB_KARG(in.scales);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float v = i_scales[g * 3 + j];
      sc[j] = raw ? expf(v) : v;
    }
  }
  if (dyn_block) {
    for (int k = 0; k < K; k++) vcf[tid * KP + k] = 0.f;
    if (slot == 0) {  // softmax(motion_coefs) params.py:43, once per Gaussian
      const float *const i_coefs = PB_KARG(in.motion_coefs);
      for (int k = 0; k < K; k++) cfl[k] = 0.f;
      if (MODE == MODE_POSES && isdyn && !raw) {  // activated coefficients, used as given
        for (int k = 0; k < K; k++) cfl[k] = i_coefs[(size_t)g * K + k];
      } else if (isdyn) {
        const float *mc = i_coefs + (size_t)g * K;
        float m = -INFINITY;
        for (int k = 0; k < K; k++) m = fmaxf(m, mc[k]);
        float sum = 0.f;
        for (int k = 0; k < K; k++) {
          float e = expf(mc[k] - m);
          cfl[k] = e;
          sum += e;
        }
        float is = 1.f / sum;
        for (int k = 0; k < K; k++) cfl[k] *= is;
      }
    }
  }
  __syncthreads();

  // accumulators over the slot's sub-samples live in LDS, one private row per lane (22 registers less across the chain;
  // a row is only ever touched by its own lane, in program order -> deterministic)
  float *ac = accs + tid * NACC;  // [0:3] v_mu, [3:7] v_q, [7:10] sc * v_sc, [10:22] dL/dRcw (9, row-major) then dL/dt (3)
#pragma unroll
  for (int c = 0; c < NACC - 1; c++) ac[c] = 0.f;
  float v_view_r[MFMA ? 12 : 1];
#pragma unroll
  for (int c = 0; c < (MFMA ? 12 : 1); c++) v_view_r[c] = 0.f;
  float *part = a.persist ? wacc + sub * a.n_shared : a.g.partials + ((size_t)grp * NSG + sub) * a.n_shared;
  float *mine = red + wv * nop;

  // Time-blended bases of sub-sample s (params.py:152-177; w uses the clamped floor): rows of the global table a.btab, read with
  // SCALAR loads where they are used (PB_BROW) - rounds 3-4 re-blended a [K][9] LDS slab per wave and pass and read it back twice.

  for (int s = slot; s < S; s += SL) {  // wave-uniform
    float vec[21];  // v9 adjoint (transl 3 + r6 6) + camera-delta adjoint 12
#pragma unroll
    for (int r = 0; r < 21; r++) vec[r] = 0.f;
    const unsigned i = (MODE == MODE_POSES && a.g_major) ? (unsigned)(active ? g : 0) * S + s : (unsigned)s * N + (active ? g : 0);  // < 2^28
    // every per-instance input of the pass is requested here, up front: the deformation forward (phase 1) needs none of
    // them and runs while they are in flight
    int radius = 0;
    float in_con[3] = {0, 0, 0}, in_vcon[3] = {0, 0, 0}, in_vm2[2] = {0, 0}, in_vdep = 0.f;
    if (MODE == MODE_RENDER && active) {
      radius = PB_KARG(radii)[i];
      const float *const i_con = PB_KARG(conics), *const i_vcon = PB_KARG(v_conics), *const i_vm2 = PB_KARG(v_means2d);
#pragma unroll
      for (int c = 0; c < 3; c++) in_con[c] = i_con[i * 3 + c], in_vcon[c] = i_vcon[i * 3 + c];
      in_vm2[0] = i_vm2[i * 2], in_vm2[1] = i_vm2[i * 2 + 1];
      in_vdep = PB_KARG(v_depths)[i];
    }
    if (active) {
      // The chain is written phase by phase with scheduling fences (SB) between the phases: left alone, the scheduler
      // interleaves the ~1500 independent multiply-adds of the whole chain for ILP and needs > 256 VGPRs (round 2:
      // 253 VGPRs, 2 waves / SIMD); fenced, the live set at any point is what the mathematics needs.
      // ---- phase 1: deformation forward ----
      float mw0[3], Rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, bb[3] = {0, 0, 0};
      float gs_inv_na = 1.f, gs_inv_nb = 1.f, gs_d = 0.f;
      if (isdyn) {
        float v9[9];
#pragma unroll
        for (int j = 0; j < 9; j++) v9[j] = 0.f;
        // (two table rows per iteration: half as many waits for the scalar loads - 114 -> 107.5 us on cfg2; an odd K's last
        // iteration reads its row twice with the second coefficient 0)
        for (int k = 0; k < K; k += 2) {
          const bool two = k + 1 < K;
          const float c0 = cfl[k], c1 = two ? cfl[k + 1] : 0.f;
          cfloat_p B0 = PB_BROW(k), B1 = PB_BROW(two ? k + 1 : k);
          float r0[9], r1[9];
#pragma unroll
          for (int j = 0; j < 9; j++) r0[j] = B0[j], r1[j] = B1[j];
#pragma unroll
          for (int j = 0; j < 9; j++) v9[j] += c0 * r0[j];
#pragma unroll
          for (int j = 0; j < 9; j++) v9[j] += c1 * r1[j];
        }
        GS6 gs;
        gram_schmidt(v9 + 3, gs);
        Rd[0] = gs.x[0], Rd[1] = gs.y[0], Rd[2] = gs.z[0];
        Rd[3] = gs.x[1], Rd[4] = gs.y[1], Rd[5] = gs.z[1];
        Rd[6] = gs.x[2], Rd[7] = gs.y[2], Rd[8] = gs.z[2];
        gs_inv_na = gs.inv_na, gs_inv_nb = gs.inv_nb, gs_d = gs.d;
        bb[0] = v9[6], bb[1] = v9[7], bb[2] = v9[8];
#pragma unroll
        for (int r = 0; r < 3; r++) mw0[r] = Rd[r * 3] * mu[0] + Rd[r * 3 + 1] * mu[1] + Rd[r * 3 + 2] * mu[2] + v9[r];
      } else {
#pragma unroll
        for (int r = 0; r < 3; r++) mw0[r] = mu[r];
      }
      SB();
      if (MODE == MODE_POSES || radius > 0) {
      float v_mw[3] = {0, 0, 0}, vRm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (MODE == MODE_POSES) {
        float v_pc[3] = {0, 0, 0};
        if (a.v_points) {
#pragma unroll
          for (int r = 0; r < 3; r++) v_pc[r] = a.v_points[i * 3 + r];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) v_mw[c] = has_cam ? cam.R[c] * v_pc[0] + cam.R[3 + c] * v_pc[1] + cam.R[6 + c] * v_pc[2] : v_pc[c];
      } else {
        // ---- phase 2: camera forward (gsplat fully_fused_projection; the visibility decision is
        // the saved radius, so no culling tests here) ----
        float Rm[9];
        {
          float Rq[9];
          quat_to_rotmat(qh[0], qh[1], qh[2], qh[3], Rq);
          if (isdyn) mat3_mul(Rd, Rq, Rm);
          else {
#pragma unroll
            for (int r = 0; r < 9; r++) Rm[r] = Rq[r];
          }
        }
        float mw[3] = {mw0[0], mw0[1], mw0[2]};
        if (a.in.RTs) {
          const float *RT = a.in.RTs + s * 12;
#pragma unroll
          for (int r = 0; r < 3; r++) mw[r] = RT[r * 4] * mw0[0] + RT[r * 4 + 1] * mw0[1] + RT[r * 4 + 2] * mw0[2] + RT[r * 4 + 3];
        }
        float pc[3];
#pragma unroll
        for (int r = 0; r < 3; r++) pc[r] = cam.R[r * 3] * mw[0] + cam.R[r * 3 + 1] * mw[1] + cam.R[r * 3 + 2] * mw[2] + cam.t[r];
        float M[9];  // Rcw Rm diag(sc): the camera-space "sqrt" of the covariance
        {
          float Wm[9];
          mat3_mul(cam.R, Rm, Wm);
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) M[r * 3 + c] = Wm[r * 3 + c] * sc[c];
        }
        SB();
        const float rz = __builtin_amdgcn_rcpf(pc[2]), rz2 = rz * rz, rz3 = rz2 * rz;
        const float xr = pc[0] * rz, yr = pc[1] * rz;
        const bool in_x = (xr <= cam.limx) && (xr >= -cam.limx), in_y = (yr <= cam.limy) && (yr >= -cam.limy);
        const float tx = pc[2] * fminf(cam.limx, fmaxf(-cam.limx, xr)), ty = pc[2] * fminf(cam.limy, fmaxf(-cam.limy, yr));
        const float J00 = cam.fx * rz, J11 = cam.fy * rz, J02 = -cam.fx * tx * rz2, J12 = -cam.fy * ty * rz2;
        // ---- phase 3: adjoint of conic = inverse(cov2d_blur), cov2d = J covc J^T ----
        float w00, w01, w11;
        {
          const float A = in_con[0], Bc = in_con[1], C = in_con[2];
          const float vA = in_vcon[0], vB = 0.5f * in_vcon[1], vC = in_vcon[2];
          const float t00 = A * vA + Bc * vB, t01 = A * vB + Bc * vC, t10 = Bc * vA + C * vB, t11 = Bc * vB + C * vC;
          w00 = -(t00 * A + t01 * Bc), w01 = -(t00 * Bc + t01 * C), w11 = -(t10 * Bc + t11 * C);
        }
        float vJ00, vJ02, vJ11, vJ12;
        {
          const float cxx = M[0] * M[0] + M[1] * M[1] + M[2] * M[2], cxy = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
          const float cxz = M[0] * M[6] + M[1] * M[7] + M[2] * M[8], cyy = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
          const float cyz = M[3] * M[6] + M[4] * M[7] + M[5] * M[8], czz = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
          const float js00 = J00 * cxx + J02 * cxz, js01 = J00 * cxy + J02 * cyz, js02 = J00 * cxz + J02 * czz;  // J covc (2x3)
          const float js10 = J11 * cxy + J12 * cxz, js11 = J11 * cyy + J12 * cyz, js12 = J11 * cyz + J12 * czz;
          vJ00 = 2.f * (w00 * js00 + w01 * js10), vJ02 = 2.f * (w00 * js02 + w01 * js12);
          vJ11 = 2.f * (w01 * js01 + w11 * js11), vJ12 = 2.f * (w01 * js02 + w11 * js12);
        }
        SB();
        float v_pc[3];
        {
          const float vm0 = in_vm2[0], vm1 = in_vm2[1];
          v_pc[0] = cam.fx * rz * vm0;
          v_pc[1] = cam.fy * rz * vm1;
          v_pc[2] = -(cam.fx * pc[0] * vm0 + cam.fy * pc[1] * vm1) * rz2 + in_vdep;
          if (in_x) v_pc[0] += -cam.fx * rz2 * vJ02; else v_pc[2] += -cam.fx * rz3 * vJ02 * tx;
          if (in_y) v_pc[1] += -cam.fy * rz2 * vJ12; else v_pc[2] += -cam.fy * rz3 * vJ12 * ty;
          v_pc[2] += -cam.fx * rz2 * vJ00 - cam.fy * rz2 * vJ11 + 2.f * cam.fx * tx * rz3 * vJ02 + 2.f * cam.fy * ty * rz3 * vJ12;
        }
        // ---- phase 4: covc = M M^T: v_M = (v_covc + v_covc^T) M, v_covc = J^T w J (symmetric, 6 distinct entries) ----
        float vW[9];
        {
          const float a0 = w00 * J00, a1 = w01 * J11, a2 = w00 * J02 + w01 * J12;  // row 0 of (w J)
          const float b0 = w01 * J00, b1 = w11 * J11, b2 = w01 * J02 + w11 * J12;  // row 1 of (w J)
          const float s00 = 2.f * J00 * a0, s01 = J00 * a1 + J11 * b0, s02 = J00 * a2 + (J02 * a0 + J12 * b0);
          const float s11 = 2.f * J11 * b1, s12 = J11 * b2 + (J02 * a1 + J12 * b1), s22 = 2.f * (J02 * a2 + J12 * b2);
          float vsc[3] = {0, 0, 0};
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float m0 = M[c], m1 = M[3 + c], m2 = M[6 + c];
            const float vm0 = s00 * m0 + s01 * m1 + s02 * m2, vm1 = s01 * m0 + s11 * m1 + s12 * m2, vm2 = s02 * m0 + s12 * m1 + s22 * m2;
            vsc[c] = m0 * vm0 + m1 * vm1 + m2 * vm2;  // = sc[c] * dL/dsc[c]  (M = Wm diag(sc))
            vW[c] = vm0 * sc[c], vW[3 + c] = vm1 * sc[c], vW[6 + c] = vm2 * sc[c];
          }
#pragma unroll
          for (int c = 0; c < 3; c++) ACC(7 + c, vsc[c]);
        }
        SB();
        // ---- phase 5: view-matrix gradient (dL/dRcw = vW Rm^T + v_pc mw^T, dL/dt = v_pc), instance rotation ----
        {
          float tmp[9];
          mat3_mul_bt(vW, Rm, tmp);
#pragma unroll
          for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int c = 0; c < 3; c++) VIEW(r * 3 + c, tmp[r * 3 + c] + v_pc[r] * mw[c]);
            VIEW(9 + r, v_pc[r]);
          }
        }
        mat3_mul_at(cam.R, vW, vRm);  // Rcw^T vW
#pragma unroll
        for (int c = 0; c < 3; c++) v_mw[c] = cam.R[c] * v_pc[0] + cam.R[3 + c] * v_pc[1] + cam.R[6 + c] * v_pc[2];
        SB();
      }
      // ---- camera delta ----
      float v_mw0[3] = {v_mw[0], v_mw[1], v_mw[2]};
      if (a.in.RTs) {
        const float *RT = a.in.RTs + s * 12;
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
          for (int c = 0; c < 3; c++) vec[9 + r * 4 + c] = v_mw[r] * mw0[c];
          vec[9 + r * 4 + 3] = v_mw[r];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) v_mw0[c] = RT[c] * v_mw[0] + RT[4 + c] * v_mw[1] + RT[8 + c] * v_mw[2];
      }
      // ---- deformation ----
      if (isdyn) {
#pragma unroll
        for (int c = 0; c < 3; c++) ACC(c, Rd[c] * v_mw0[0] + Rd[3 + c] * v_mw0[1] + Rd[6 + c] * v_mw0[2]);
        float vRd[9];
        if (MODE == MODE_RENDER) {
          float Rq[9];
          quat_to_rotmat(qh[0], qh[1], qh[2], qh[3], Rq);
          mat3_mul_bt(vRm, Rq, vRd);  // vRm Rq^T
          float tmp[9];
          mat3_mul_at(Rd, vRm, tmp);  // Rd^T vRm = dL/dRq
          {
            float vq[4] = {0, 0, 0, 0};
            rotmat_adj_to_quat(qh, tmp, vq);
#pragma unroll
            for (int c = 0; c < 4; c++) ACC(3 + c, vq[c]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 9; r++) vRd[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) vRd[r * 3 + c] += v_mw0[r] * mu[c];
        if (MODE == MODE_POSES) {
          if (a.v_transforms) {  // transforms [.,3,4] = [Rd | transl]
            const float *vt = a.v_transforms + (a.g_major ? (unsigned)g * S + s : (unsigned)s * G + g) * 12u;
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
              for (int c = 0; c < 3; c++) vRd[r * 3 + c] += vt[r * 4 + c];
              v_mw0[r] += vt[r * 4 + 3];
            }
          }
          if (a.v_quats_out) {  // quats_out = normalize(rotmat_to_unitquat(Rd) (x) qh)  scene_model.py:94-102
            PoseQ pq;
            pose_quat(Rd, qh, pq);
            float vo[4];
#pragma unroll
            for (int c = 0; c < 4; c++) vo[c] = a.v_quats_out[i * 4 + c];
            float vq[4] = {0, 0, 0, 0};
            pose_quat_adj(pq, qh, vo, vRd, vq);
#pragma unroll
            for (int c = 0; c < 4; c++) ACC(3 + c, vq[c]);
          }
        }
        SB();
        // Gram-Schmidt adjoint (columns x, y, z of Rd; transforms.py:41-53)
        const float gx[3] = {Rd[0], Rd[3], Rd[6]}, gy[3] = {Rd[1], Rd[4], Rd[7]};
        float vx[3] = {vRd[0], vRd[3], vRd[6]}, vy[3] = {vRd[1], vRd[4], vRd[7]}, vz[3] = {vRd[2], vRd[5], vRd[8]};
        // z = x cross y
        vx[0] += gy[1] * vz[2] - gy[2] * vz[1];
        vx[1] += gy[2] * vz[0] - gy[0] * vz[2];
        vx[2] += gy[0] * vz[1] - gy[1] * vz[0];
        vy[0] += vz[1] * gx[2] - vz[2] * gx[1];
        vy[1] += vz[2] * gx[0] - vz[0] * gx[2];
        vy[2] += vz[0] * gx[1] - vz[1] * gx[0];
        // y = bp * inv_nb
        float vbp[3];
        {
          const bool clamped = gs_inv_nb >= 1e12f;
          const float dty = clamped ? 0.f : (vy[0] * gy[0] + vy[1] * gy[1] + vy[2] * gy[2]);
#pragma unroll
          for (int c = 0; c < 3; c++) vbp[c] = (vy[c] - dty * gy[c]) * gs_inv_nb;
        }
        // bp = b - (b.x) x
        const float vbx = vbp[0] * gx[0] + vbp[1] * gx[1] + vbp[2] * gx[2];
        float vb[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          vb[c] = vbp[c] - vbx * gx[c];
          vx[c] += -gs_d * vbp[c] - vbx * bb[c];
        }
        // x = a * inv_na
        float va_[3];
        {
          const bool clamped = gs_inv_na >= 1e12f;
          const float dtx = clamped ? 0.f : (vx[0] * gx[0] + vx[1] * gx[1] + vx[2] * gx[2]);
#pragma unroll
          for (int c = 0; c < 3; c++) va_[c] = (vx[c] - dtx * gx[c]) * gs_inv_na;
        }
        vec[0] = v_mw0[0], vec[1] = v_mw0[1], vec[2] = v_mw0[2];
        vec[3] = va_[0], vec[4] = va_[1], vec[5] = va_[2];
        vec[6] = vb[0], vec[7] = vb[1], vec[8] = vb[2];
        SB();
        // coefficient gradients: v_c[k] += B_s[k] . vec[0:9]
        for (int k = 0; k < K; k += 2) {
          const bool two = k + 1 < K;
          cfloat_p B0 = PB_BROW(k), B1 = PB_BROW(two ? k + 1 : k);
          float r0[9], r1[9];
#pragma unroll
          for (int j = 0; j < 9; j++) r0[j] = B0[j], r1[j] = B1[j];
          float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
          for (int j = 0; j < 9; j++) acc0 += r0[j] * vec[j];
#pragma unroll
          for (int j = 0; j < 9; j++) acc1 += r1[j] * vec[j];
          vcf[tid * KP + k] += acc0;
          if (two) vcf[tid * KP + k + 1] += acc1;
        }

      } else {
#pragma unroll
        for (int c = 0; c < 3; c++) ACC(c, v_mw0[c]);
        if (MODE == MODE_RENDER) {
          float vq[4] = {0, 0, 0, 0};
          rotmat_adj_to_quat(qh, vRm, vq);
#pragma unroll
          for (int c = 0; c < 4; c++) ACC(3 + c, vq[c]);
        }
        else if (a.v_quats_out) {  // static Gaussian: quats_out = normalize(q)
#pragma unroll
          for (int c = 0; c < 4; c++) ACC(3 + c, a.v_quats_out[i * 4 + c]);
        }
      }
      }  // visible
    }
    SB();
    // ---- shared gradients of sub-sample s: 9K column sums weighted by cf + 12 plain, over THIS wave's 64 Gaussians ----
    if (shared) {
      if (dyn) {
        if (dyn_block && !MFMA) {  // few bases: K x (9 multiplies + a 9-value wave reduction) on the VALU
          for (int k = 0; k < K; k++) {
            const float c = cfl[k];
            float p[9];
#pragma unroll
            for (int jj = 0; jj < 9; jj++) p[jj] = c * vec[jj];
            wave_sum_store(p, mine, k * 9, lane);
          }
        } else if (dyn_block) {
          // v_Bs[k][j] = sum over the wave's 64 Gaussians of coef[g][k] * vec[g][j]: a [K x 64] x [64 x 9] product on the
          // matrix pipe: 16 v_mfma_f32_16x16x4_f32 per 16 bases (A = coefficients, B = vec, both read from LDS in the
          // operand layout); exact f32, fixed order.
          typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int jj = 0; jj < 9; jj++) svec[tid * 9 + jj] = vec[jj];
          __builtin_amdgcn_wave_barrier();  // a wave only reads the rows of its own 64 lanes
          const int ln = lane & 15, lk = lane >> 4;
          for (int mt = 0; mt < K; mt += 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const bool am = mt + ln < K;
#pragma unroll 4
            for (int kk = 0; kk < 16; kk++) {
              const int src = 4 * kk + lk;  // the Gaussian (lane of this wave) this operand element belongs to
              const float av = am ? cf[(sub * GPB + src) * KP + mt + ln] : 0.f;
              const float bv = ln < 9 ? svec[(wv * 64 + src) * 9 + ln] : 0.f;
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
            if (ln < 9) {
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const int m = mt + lk * 4 + r;  // D layout: row = (lane >> 4) * 4 + r, column = lane & 15
                if (m < K) mine[m * 9 + ln] = acc[r];
              }
            }
          }
        } else {
          for (int o = lane; o < nk; o += 64) mine[o] = 0.f;
        }
      }
      {
        float q[12];
#pragma unroll
        for (int r = 0; r < 12; r++) q[r] = vec[9 + r];
        wave_sum_store(q, mine, nk, lane);
      }
      __builtin_amdgcn_wave_barrier();
      if (a.persist) {
        for (int o = lane; o < no; o += 64) part[s * no + o] += mine[o];
      } else {
        for (int o = lane; o < no; o += 64) part[s * no + o] = mine[o];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- viewmat partials (12 plain column sums): per-wave ladder + fixed-order sum of the 4 slots ----
  {
    float v_view[12];
#pragma unroll
    for (int c = 0; c < 12; c++) {
      if constexpr (MFMA) v_view[c] = v_view_r[c];
      else v_view[c] = ac[10 + c];
    }
    wave_sum_store(v_view, mine, 0, lane);
  }
  __syncthreads();
  if (tid < 12) {  // (wave 0 = sub-group 0: the block's view-matrix sum rides in ITS partial vector, zeros in the others')
    const float v = (red[tid] + red[nop + tid]) + (red[2 * nop + tid] + red[3 * nop + tid]);
    if (a.persist) part[S * no + tid] += v;
    else {
      part[S * no + tid] = v;
      for (int u = 1; u < NSG; u++) part[(size_t)u * a.n_shared + S * no + tid] = 0.f;
    }
  }

  if (active) {
  // sum of the sub-group's SL slots' accumulators of Gaussian `lane`, fixed order
  const int row0 = sub * SL * 64 + lane;  // the row of slot 0 in the [BLK][.] per-lane arrays
  auto slots4 = [&](int c) {
    const float *p = accs + row0 * NACC + c;
    if (SL == 4) return (p[0] + p[64 * NACC]) + (p[128 * NACC] + p[192 * NACC]);
    if (SL == 2) return p[0] + p[64 * NACC];
    return p[0];
  };
  // ---- per-Gaussian leaves: the sub-group's SL waves split the four tensor groups (group p goes to slot p % SL) ----
  if (slot == 0) {
    float *const o_means = PB_KARG(g.v_means);
    o_means[g * 3] = slots4(0), o_means[g * 3 + 1] = slots4(1), o_means[g * 3 + 2] = slots4(2);
    if (MODE == MODE_RENDER) {
      float *const o_scales = PB_KARG(g.v_scales);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const float v = slots4(7 + j);  // accumulated as sc * dL/dsc
        o_scales[g * 3 + j] = raw ? v : v / sc[j];
      }
    }
  }
  if (slot == 1 % SL) {
    float *const o_quats = PB_KARG(g.v_quats);
    if (need_q && o_quats) {  // adjoint of q / max(|q|, eps)
      const float w = qh[0], x = qh[1], y = qh[2], z = qh[3];
      const float vq0 = slots4(3), vq1 = slots4(4), vq2 = slots4(5), vq3 = slots4(6);
      const float dot = vq0 * w + vq1 * x + vq2 * y + vq3 * z;
      float4 o4 = make_float4((vq0 - dot * w) * inv_qn, (vq1 - dot * x) * inv_qn, (vq2 - dot * y) * inv_qn,
                              (vq3 - dot * z) * inv_qn);
      *reinterpret_cast<float4 *>(o_quats + (size_t)g * 4) = o4;
    }
  }
  if (slot == 2 % SL) {
    if (isdyn) {  // softmax adjoint
      const bool act = MODE == MODE_POSES && !raw;
      float dot = 0.f;
      float *vr = vcf + row0 * KP;
      for (int k = 0; k < K; k++) {
        const float v = SL == 4 ? (vr[k] + vr[64 * KP + k]) + (vr[128 * KP + k] + vr[192 * KP + k]) : SL == 2 ? vr[k] + vr[64 * KP + k] : vr[k];
        vr[k] = v;  // (only this lane touches its slot-0 row from here on)
        dot += cfl[k] * v;
      }
      float *const o_coefs = PB_KARG(g.v_motion_coefs);
      for (int k = 0; k < K; k++)
        o_coefs[(size_t)g * K + k] = act ? vr[k] : cfl[k] * (vr[k] - dot);
    }
  }
  if (slot == 3 % SL && MODE == MODE_RENDER) {
    const float o = PB_KARG(opac_act)[g];
    const float vo = PB_KARG(v_opac_act)[g];
    PB_KARG(g.v_opacities)[g] = raw ? vo * o * (1.f - o) : vo;
    const float *const i_vctab = PB_KARG(v_ctab);
    const float *const i_ctab = PB_KARG(ctab);
    float *const o_colors = PB_KARG(g.v_colors);
    const int D = d.D, DP = (D + 3) & ~3;
    const bool vec_out = (D & 3) == 0 && (((uintptr_t)o_colors) & 15) == 0;
    for (int ch = 0; ch < DP; ch += 4) {  // 16 bytes at a time (the table rows are 16-byte aligned), as in k_project_fwd
      float4 v4 = *reinterpret_cast<const float4 *>(i_vctab + (size_t)g * DP + ch);
      if ((d.flags & D4GS_RAW_COLORS) && ch < d.n_sigmoid) {
        const float4 c4 = *reinterpret_cast<const float4 *>(i_ctab + (size_t)g * DP + ch);
        const float *cp = &c4.x;
        float *vp = &v4.x;
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (ch + u < d.n_sigmoid) vp[u] *= cp[u] * (1.f - cp[u]);
      }
      if (vec_out) {
        *reinterpret_cast<float4 *>(o_colors + (size_t)g * D + ch) = v4;
      } else {
        const float *vp = &v4.x;
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (ch + u < D) o_colors[(size_t)g * D + ch + u] = vp[u];
      }
    }
  }
  }  // active
  __syncthreads();  // cf / vcf / accs / red are rewritten by the next group
  }  // groups
  if (a.persist) {  // ONE partial vector per block: the sub-groups' sums added in fixed order
    __syncthreads();
    float *dst = a.g.partials + (size_t)blockIdx.x * a.n_shared;
    for (int o = tid; o < a.n_shared; o += BLK) {
      float v = wacc[o];
#pragma unroll
      for (int u = 1; u < NSG; u++) v += wacc[u * a.n_shared + o];
      dst[o] = v;
    }
  }
}

// Sum the per-block partial vectors in a fixed order (deterministic), two coalesced passes: thread o of chunk c adds its
// output's values of the chunk's blocks one after the other (neighbouring lanes read neighbouring floats), then one more
// launch of the same kernel adds the RCH chunk sums.  (Round 2 had a wave stride over the blocks of one output: 4-byte
// reads n floats apart - 30 us once the 64-Gaussian blocks made 4x as many partials.)
constexpr int RCH = 64;
// rows [b0, b1) of `x` (row length n), column o, summed into four interleaved accumulators ((a0 + a1) + (a2 + a3) at the end): the one
// order both reduction passes use.  Sixteen rows are LOADED before they are added (round 5: the `#pragma unroll 4` loop still waited
// for each group of four - a lane of the one-block k_finish paid 16 dependent round trips for its 64 chunk sums); same additions, same order.
__device__ __forceinline__ float d4gs_sum_rows(const float *__restrict__ x, int b0, int b1, int n, int o) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = b0;
  for (; b + 15 < b1; b += 16) {
    float r[16];
#pragma unroll
    for (int u = 0; u < 16; u++) r[u] = x[(size_t)(b + u) * n + o];
#pragma unroll
    for (int u = 0; u < 16; u += 4) a0 += r[u], a1 += r[u + 1], a2 += r[u + 2], a3 += r[u + 3];
  }
  for (; b + 3 < b1; b += 4) {
    const float r0 = x[(size_t)b * n + o], r1 = x[(size_t)(b + 1) * n + o], r2 = x[(size_t)(b + 2) * n + o], r3 = x[(size_t)(b + 3) * n + o];
    a0 += r0, a1 += r1, a2 += r2, a3 += r3;
  }
  for (; b < b1; b++) a0 += x[(size_t)b * n + o];
  return (a0 + a1) + (a2 + a3);
}
__global__ void __launch_bounds__(256) k_reduce_partials(const float *partials, int n_blocks, int n, float *out) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const int per = (n_blocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(b0 + per, n_blocks);
  out[(size_t)blockIdx.y * n + o] = d4gs_sum_rows(partials, b0, b1, n, o);
}

// scatter the reduced shared gradients: v_Bs -> rots / transls / times ; camera deltas ; viewmat.
// One thread per LEAF element (basis k, frame, component): it walks the sub-samples in order and adds the (1 - w) share
// of those whose floor frame it is, then the w share of those whose ceil frame it is - independent loads, one store, no
// read-modify-write chains through memory (the first version's (k, j) owners did S dependent global updates: 9 us).
// `red_in`: n_in > 0: [n_in][n_shared] chunk sums of k_reduce_partials' first pass - every block first adds them (the second
// pass's arithmetic and order, so the totals are the same bits) into LDS and works from there: one launch less per frame;
// n_in == 0: the [n_shared] totals themselves, in global memory (shared vectors beyond the LDS budget).
__global__ void __launch_bounds__(1024) k_finish(const BwdArgs a, const float *red_in, int n_in) {
  extern __shared__ __attribute__((aligned(16))) float sred[];
  const float *red = red_in;
  if (n_in > 0) {
    const int n = a.n_shared;
    for (int o = threadIdx.x; o < n; o += blockDim.x) sred[o] = d4gs_sum_rows(red_in, 0, n_in, n, o);
    __syncthreads();
    red = sred;
  }
  const D4gsDims &d = a.d;
  const int K = d.K, T = d.T, S = d.S;
  const bool dyn = d.G > 0;
  const int nk = dyn ? K * 9 : 0;
  const int stride = nk + 12;
  int fin = dyn ? K * T * 9 : 0;
  fin = max(max(fin, S * 12), 16);
  // (grid-stride: the mid-size launch is ONE block of 1 024 lanes that has just summed the chunk sums into LDS)
  for (int gtid = blockIdx.x * blockDim.x + threadIdx.x; gtid < fin; gtid += gridDim.x * blockDim.x) {
  if (dyn) {
    if (gtid < K * T * 9) {
      const int k = gtid / (T * 9), r = gtid - k * T * 9, fr = r / 9, j = r - fr * 9;
      float acc = 0.f;
      for (int s = 0; s < S; s++) {
        const float t = a.in.times[s];
        const float ff = fminf(fmaxf(floorf(t), 0.f), (float)(T - 1));
        const float cfl = fminf(fmaxf(ceilf(t), 0.f), (float)(T - 1));
        const float w = t - ff;
        const float v = red[s * stride + k * 9 + j];
        if ((int)ff == fr) acc += (1.f - w) * v;
        if ((int)cfl == fr) acc += w * v;
      }
      if (j < 3) a.g.v_transls[(k * T + fr) * 3 + j] = acc;
      else a.g.v_rots[(k * T + fr) * 6 + j - 3] = acc;
    }
    // v_times[s] = dL/dw = sum_kj (base_c - base_f) * v_Bs
    if (gtid < S) {
      const int s = gtid;
      const float t = a.in.times[s];
      const float ff = fminf(fmaxf(floorf(t), 0.f), (float)(T - 1));
      const float cfl = fminf(fmaxf(ceilf(t), 0.f), (float)(T - 1));
      const int f = (int)ff, c = (int)cfl;
      float acc = 0.f;
      for (int k = 0; k < K; k++)
        for (int j = 0; j < 9; j++) {
          float bf, bc;
          if (j < 3) {
            bf = a.in.transls[(k * T + f) * 3 + j], bc = a.in.transls[(k * T + c) * 3 + j];
          } else {
            bf = a.in.rots[(k * T + f) * 6 + j - 3], bc = a.in.rots[(k * T + c) * 6 + j - 3];
          }
          acc += (bc - bf) * red[s * stride + k * 9 + j];
        }
      if (a.g.v_times) a.g.v_times[s] = acc;
    }
  }
  if (a.g.v_RTs && gtid < S * 12) {
    const int s = gtid / 12, r = gtid - s * 12;
    a.g.v_RTs[gtid] = red[s * stride + nk + r];
  }
  if (a.g.v_viewmat && gtid < 16) {
    const int r = gtid / 4, c = gtid % 4;
    float v = 0.f;
    if (r < 3) v = c < 3 ? red[S * stride + r * 3 + c] : red[S * stride + 9 + r];
    a.g.v_viewmat[gtid] = v;
  }
  }
}

int n_shared_of(const D4gsDims *d) { return d->S * ((d->G > 0 ? d->K * 9 : 0) + 12) + 12; }

}  // namespace

int d4gs_bases_table_launch(const D4gsDims *dims, const D4gsProjIn *in, float *btab, hipStream_t stream) {
  D4GS_LAUNCH("k_bases_table", k_bases_table, dim3(dims->S), dim3(256), 0, stream, *dims, in->times, in->rots, in->transls, btab);
  return d4gs_check_launch("k_bases_table");
}

constexpr int PERSIST_MAX_SHARED = 8192;  // floats of LDS a block may spend on its shared-gradient sums

// The persistent grid is exactly what the device keeps resident (CUs x blocks per CU for this kernel's registers and
// LDS): no second, partially filled round of blocks.  Cached per (device, kernel, LDS size); the summation order of the
// shared gradients follows the grid size, so results are bit-reproducible on a given device model, not across models.
// -> 0 when there is no device to ask (the size query of a GPU-less host).
static int resident_blocks(const void *fn, size_t lds) {
  struct Entry { int dev; const void *fn; size_t lds; int blocks; };
  thread_local Entry cache[8];
  thread_local int n_cache = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  for (int i = 0; i < n_cache; i++)
    if (cache[i].dev == dev && cache[i].fn == fn && cache[i].lds == lds) return cache[i].blocks;
  int cus = 0, per = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, BLK, lds) != hipSuccess || per < 1 || cus < 1) {
    (void)hipGetLastError();
    return 0;
  }
  if (n_cache < 8) cache[n_cache++] = Entry{dev, fn, lds, cus * per};
  return cus * per;
}

template <int MODE>
static const void *kernel_of(bool mfma, int sl) {
  if constexpr (MODE == MODE_RENDER) {
    if (sl == 1) return mfma ? (const void *)k_project_bwd<MODE, true, 1> : (const void *)k_project_bwd<MODE, false, 1>;
    if (sl == 2) return mfma ? (const void *)k_project_bwd<MODE, true, 2> : (const void *)k_project_bwd<MODE, false, 2>;
  }
  return mfma ? (const void *)k_project_bwd<MODE, true, 4> : (const void *)k_project_bwd<MODE, false, 4>;
}

// launch shape of k_project_bwd for one configuration: grid, LDS bytes, persistent or one group per block
struct BwdPlan {
  int blocks, persist, sl, nparts;  // nparts: shared-gradient partial vectors the launch leaves (blocks x sub-groups)
  size_t lds;
  const void *fn;
};
template <int MODE>
static BwdPlan plan_bwd(const D4gsDims *dims) {
  BwdPlan p;
  const int K = dims->G > 0 ? dims->K : 0;
  const int KP = K | 1;
  const size_t nk = (size_t)K * 9;
  const int n_shared = n_shared_of(dims);
  p.sl = slots_of<MODE>(dims->S);
  p.fn = kernel_of<MODE>(K > 8, p.sl);
  const int nsg = SLOTS / p.sl;
  auto lds_of = [&](bool persist) {
    return sizeof(float) * ((size_t)nsg * GPB * KP + (size_t)BLK * KP + SLOTS * (nk + 13) +
                            (size_t)BLK * nacc_of(K > 8) + (K > 8 ? (size_t)BLK * 9 : 0) + (persist ? (size_t)nsg * n_shared : 0));
  };
  p.blocks = (dims->N + nsg * GPB - 1) / (nsg * GPB);
  p.persist = 0;
  p.lds = lds_of(false);
  if (n_shared <= PERSIST_MAX_SHARED) {
    // Persistent blocks are worth it when every block walks many groups, or when all groups cost the same: a block's
    // groups are b, b + grid, ..., so a scene with few groups per block AND two kinds of groups (dynamic ones cost ~2.5x a
    // static one; the reference's 40 k + 100 k training shape: 2.8 groups per block) is balanced better by the hardware's
    // own block scheduler - measured 113 us one group per block against 146 us persistent (cfg5, 20 groups per block:
    // 1149 against 1012 us).
    const int res = resident_blocks(p.fn, lds_of(true));
    const bool uniform = dims->G == 0 || dims->G == dims->N;
    if (res > 0 && p.blocks >= (uniform ? 2 : 8) * res) p.blocks = res, p.persist = 1, p.lds = lds_of(true);
    // sub-group mapping (S < 3): the shared-gradient sums always go through LDS, so that a block leaves ONE partial vector (the
    // fixed-order sum of its sub-groups') whether it walks one group or many - 4x fewer vectors for k_reduce_partials
    else if (nsg > 1) p.persist = 1, p.lds = lds_of(true);
  }
  p.nparts = p.persist ? p.blocks : p.blocks * nsg;
  return p;
}

extern "C" size_t d4gs_bwd_partials_elems(const D4gsDims *d) {
  // (the render and the poses instantiation have different register counts: take the larger grid of the two)
  const int b0 = plan_bwd<MODE_RENDER>(d).nparts, b1 = plan_bwd<MODE_POSES>(d).nparts;
  const size_t extra = (size_t)d->S * (d->G > 0 ? d->K : 0) * 16;  // room for the blended-bases table of a caller without one
  return ((size_t)(b0 > b1 ? b0 : b1) + RCH + 1) * (size_t)n_shared_of(d) + extra;
}

template <int MODE>
static int launch_project_bwd(BwdArgs &a, const D4gsDims *dims, const D4gsLeafGrads *grads, hipStream_t stream,
                              const float *blend_bases = nullptr) {
  a.n_shared = n_shared_of(dims);
  const BwdPlan pl = plan_bwd<MODE>(dims);
  a.persist = pl.persist;
  const size_t lds = pl.lds;
  const int blocks = pl.blocks;
  if (lds > 160 * 1024) {
    d4gs_set_error("project_bwd: LDS budget exceeded (K=%d)", dims->K);
    return D4GS_EINVAL;
  }
  if (lds > 64 * 1024)  // per call: the attribute belongs to the (function, device) pair and a process may drive several GPUs
    (void)hipFuncSetAttribute(pl.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const char *name = MODE == MODE_RENDER ? "k_project_bwd" : "k_project_bwd[poses]";
  a.btab = blend_bases;  // D4gsProjOut.blend_bases, left by k_project_fwd
  if (dims->G > 0 && !blend_bases) {  // a caller without it (and the poses adjoint): built here, behind the partial vectors
    float *btab = grads->partials + ((size_t)pl.nparts + RCH + 1) * a.n_shared;
    a.btab = btab;
    int rc0 = d4gs_bases_table_launch(dims, &a.in, btab, stream);
    if (rc0) return rc0;
  }
  {
    ProfScope _ps(name, stream);
    void *kargs[] = {(void *)&a};
    (void)hipLaunchKernel(pl.fn, dim3(blocks), dim3(BLK), kargs, lds, stream);
  }
  int rc = d4gs_check_launch(name);
  if (rc) return rc;
  float *red2 = grads->partials + (size_t)pl.nparts * a.n_shared;  // [RCH][n_shared] chunk sums, then [n_shared] totals
  float *red = red2 + (size_t)RCH * a.n_shared;
  D4GS_LAUNCH("k_reduce_partials", k_reduce_partials, dim3((a.n_shared + 255) / 256, RCH), dim3(256), 0, stream,
              (const float *)grads->partials, pl.nparts, a.n_shared, red2);
  int fin = dims->G > 0 ? dims->K * dims->T * 9 : 0;
  if (fin < dims->S * 12) fin = dims->S * 12;
  if (fin < 16) fin = 16;
  const size_t fin_lds = sizeof(float) * (size_t)a.n_shared;
  // second reduction pass inside k_finish - for SHORT shared vectors only (one or two sub-samples: a rank of a sharded frame):
  // every block of k_finish redoes it (RCH x n_shared loads), which costs more than the launch it saves from ~256 floats on
  // (measured: cfg2 S = 8, 540 floats: 13 -> 19 us; refdefault, 2 124 floats: 13 -> 60 us; S = 1, 78 floats: 24 -> 16 us)
  if (a.n_shared <= 160) {
    D4GS_LAUNCH("k_finish", k_finish, dim3((fin + 255) / 256), dim3(256), fin_lds, stream, a, (const float *)red2, RCH);
    return d4gs_check_launch("k_finish");
  }
  // Mid-size shared vectors (round 5; cfg2: 540 floats, 1 296 leaf elements): ONE block of 1 024 lanes does the second pass once - RCH x
  // n_shared coalesced loads, the same additions in the same order, so the totals are the same bits - and then walks the leaf elements
  // itself: one launch instead of two (k_reduce_partials' second pass + a six-block k_finish).  D4GS_FINISH_ONE=0: the two launches (A/B).
  static const bool one_env = []() { const char *e = getenv("D4GS_FINISH_ONE"); return !(e && e[0] == '0'); }();
  if (one_env && a.n_shared <= 1024 && fin <= 8192) {
    D4GS_LAUNCH("k_finish", k_finish, dim3(1), dim3(1024), fin_lds, stream, a, (const float *)red2, RCH);
    return d4gs_check_launch("k_finish");
  }
  D4GS_LAUNCH("k_reduce_partials", k_reduce_partials, dim3((a.n_shared + 255) / 256, 1), dim3(256), 0, stream,
              (const float *)red2, RCH, a.n_shared, red);
  D4GS_LAUNCH("k_finish", k_finish, dim3((fin + 255) / 256), dim3(256), 0, stream, a, (const float *)red, 0);
  return d4gs_check_launch("k_finish");
}

int d4gs_poses_bwd_impl(const D4gsDims *dims, const D4gsProjIn *in, const D4gsPoses *v_out, const D4gsLeafGrads *grads,
                        hipStream_t stream) {
  BwdArgs a{};
  a.d = *dims;
  a.in = *in;
  a.g = *grads;
  a.v_points = v_out->means, a.v_quats_out = v_out->quats, a.v_transforms = v_out->transforms, a.g_major = v_out->g_major;
  return launch_project_bwd<MODE_POSES>(a, dims, grads, stream);
}

int d4gs_project_bwd_impl(const D4gsDims *dims, const D4gsProjIn *in, const D4gsProjOut *proj, const float *v_means2d,
                          const float *v_conics, const float *v_depths, const float *v_opac_act, const float *v_ctab,
                          const D4gsLeafGrads *grads, hipStream_t stream) {
  BwdArgs a{};
  a.d = *dims;
  a.in = *in;
  a.radii = proj->radii, a.conics = proj->conics, a.ctab = proj->ctab, a.opac_act = proj->opac_act;
  a.v_means2d = v_means2d, a.v_conics = v_conics, a.v_depths = v_depths, a.v_opac_act = v_opac_act, a.v_ctab = v_ctab;
  a.g = *grads;
  return launch_project_bwd<MODE_RENDER>(a, dims, grads, stream, proj->blend_bases);
}
