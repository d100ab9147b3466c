// project_bwd.hip -- adjoint of project_fwd.hip: projection, camera delta, motion-basis deformation and the
// activations, for ALL S sub-samples, reduced to leaf gradients.
//
// Replaces gsplat fully_fused_projection_bwd and the torch autograd of flow3d/scene_model.py:67-120,352-353,
// flow3d/params.py:39-43,142-180, flow3d/transforms.py:41-53 (reference backward: flow3d/trainer.py:231).
//
// One lane per Gaussian, looping over the sub-samples: per-Gaussian leaf gradients (means, quats, scales,
// opacity, colours, motion coefficients) are accumulated over S in registers / LDS in a FIXED order and written
// once - no atomics.  The small SHARED gradients (time-blended bases S*K*9, camera deltas S*12, viewmat 12) are
// reduced per block through LDS column sums, written as per-block partials, summed over blocks by a second kernel
// in a fixed order, and finally scattered to rots / transls / times by k_finish.  Every sum is deterministic.
#include "common.h"

namespace {

constexpr int BLK = D4GS_PROJ_BLOCK;
constexpr int NV = 21;  // per-thread per-s vector: v9 (transl 3 + r6 6) + camera-delta adjoint 12; odd stride

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
  const float *v_points;  // non-null: "points only" mode - adjoint of d4gs_points_fwd ([S,N,3] camera-space means)
};

__device__ __forceinline__ void preblend_bases_b(const BwdArgs &a, float *Bs) {
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

__global__ void __launch_bounds__(BLK) k_project_bwd(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const D4gsDims &d = a.d;
  const int N = d.N, G = d.G, K = d.K, S = d.S;
  const int KP = K | 1;
  const int nBs = (S * K * 9 + 3) & ~3;
  float *Bs = smem;                  // [S][K][9]
  float *cf = Bs + nBs;              // [BLK][KP]   softmaxed coefficients
  float *vcf = cf + BLK * KP;        // [BLK][KP]   their gradients, summed over s
  float *psum = vcf + BLK * KP;       // [2][4][9K+13]  per-wave segment sums (+ dump slot), double-buffered over s
  float *svec = psum + 8 * (9 * K + 13);  // [BLK][9]  this sub-sample's (v_transl 3, v_r6 6) of every lane: MFMA B operand
  const int tid = threadIdx.x;
  const int g = blockIdx.x * BLK + tid;
  const Cam cam = load_cam(a.in.viewmat, a.in.Kmat, d.width, d.height);
  const bool dyn = G > 0;
  const bool dyn_block = (blockIdx.x * BLK) < G;
  if (dyn_block) preblend_bases_b(a, Bs);
  const bool active = g < N;
  const bool isdyn = active && g < G;
  const bool raw = d.flags & D4GS_RAW_PARAMS;
  const bool pts = a.v_points != nullptr;

  float mu[3] = {0, 0, 0}, Rq[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, sc[3] = {1, 1, 1}, qh[4] = {1, 0, 0, 0}, inv_qn = 1.f;
  if (active) mu[0] = a.in.means[g * 3], mu[1] = a.in.means[g * 3 + 1], mu[2] = a.in.means[g * 3 + 2];
  if (active && !pts) {
    const float4 q = *reinterpret_cast<const float4 *>(a.in.quats + (size_t)g * 4);
    inv_qn = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    qh[0] = q.x * inv_qn, qh[1] = q.y * inv_qn, qh[2] = q.z * inv_qn, qh[3] = q.w * inv_qn;
    quat_to_rotmat(qh[0], qh[1], qh[2], qh[3], Rq);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float v = a.in.scales[g * 3 + j];
      sc[j] = raw ? expf(v) : v;
    }
  }
  if (dyn_block) {
    for (int k = 0; k < K; k++) cf[tid * KP + k] = 0.f, vcf[tid * KP + k] = 0.f;
    if (isdyn) {
      const float *mc = a.in.motion_coefs + (size_t)g * K;
      float m = -INFINITY;
      for (int k = 0; k < K; k++) m = fmaxf(m, mc[k]);
      float sum = 0.f;
      for (int k = 0; k < K; k++) {
        float e = expf(mc[k] - m);
        cf[tid * KP + k] = e;
        sum += e;
      }
      float is = 1.f / sum;
      for (int k = 0; k < K; k++) cf[tid * KP + k] *= is;
    }
  }
  __syncthreads();

  float v_mu[3] = {0, 0, 0}, v_Rq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, v_sc[3] = {0, 0, 0};
  float v_view[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // dL/dRcw (9, row-major) then dL/dt (3)
  float *part = a.g.partials + (size_t)blockIdx.x * a.n_shared;

  for (int s = 0; s < S; s++) {
    float vec[NV];
#pragma unroll
    for (int r = 0; r < NV; r++) vec[r] = 0.f;
    const size_t i = (size_t)s * N + (active ? g : 0);
    if (active && (pts || a.radii[i] > 0)) {
      // ---- recompute the forward ----
      float mw0[3], Rm[9], Rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, v9[9];
      GS6 gs;
      if (isdyn) {
#pragma unroll
        for (int j = 0; j < 9; j++) v9[j] = 0.f;
        const float *B = Bs + s * K * 9;
        for (int k = 0; k < K; k++) {
          float c = cf[tid * KP + k];
#pragma unroll
          for (int j = 0; j < 9; j++) v9[j] += c * B[k * 9 + j];
        }
        gram_schmidt(v9 + 3, gs);
        Rd[0] = gs.x[0], Rd[1] = gs.y[0], Rd[2] = gs.z[0];
        Rd[3] = gs.x[1], Rd[4] = gs.y[1], Rd[5] = gs.z[1];
        Rd[6] = gs.x[2], Rd[7] = gs.y[2], Rd[8] = gs.z[2];
#pragma unroll
        for (int r = 0; r < 3; r++) mw0[r] = Rd[r * 3] * mu[0] + Rd[r * 3 + 1] * mu[1] + Rd[r * 3 + 2] * mu[2] + v9[r];
        mat3_mul(Rd, Rq, Rm);
      } else {
#pragma unroll
        for (int r = 0; r < 3; r++) mw0[r] = mu[r];
#pragma unroll
        for (int r = 0; r < 9; r++) Rm[r] = Rq[r];
      }
      float mw[3] = {mw0[0], mw0[1], mw0[2]};
      float RT[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      if (a.in.RTs) {
#pragma unroll
        for (int r = 0; r < 12; r++) RT[r] = a.in.RTs[s * 12 + r];
#pragma unroll
        for (int r = 0; r < 3; r++) mw[r] = RT[r * 4] * mw0[0] + RT[r * 4 + 1] * mw0[1] + RT[r * 4 + 2] * mw0[2] + RT[r * 4 + 3];
      }
      float v_pc[3], vRm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (pts) {
#pragma unroll
        for (int r = 0; r < 3; r++) v_pc[r] = a.v_points[i * 3 + r];
      } else {
        ProjOut p;
        D4gsDims dd = d;
        dd.near_plane = -INFINITY, dd.far_plane = INFINITY;  // the visibility decision is the saved radius
        project_instance(cam, mw, Rm, sc, dd, p);
        // ---- adjoint of conic = inverse(cov2d_blur) ----
        const float A = a.conics[i * 3], Bc = a.conics[i * 3 + 1], C = a.conics[i * 3 + 2];
        const float vA = a.v_conics[i * 3], vB = 0.5f * a.v_conics[i * 3 + 1], vC = a.v_conics[i * 3 + 2];
        const float t00 = A * vA + Bc * vB, t01 = A * vB + Bc * vC, t10 = Bc * vA + C * vB, t11 = Bc * vB + C * vC;
        const float w00 = -(t00 * A + t01 * Bc), w01 = -(t00 * Bc + t01 * C), w11 = -(t10 * Bc + t11 * C);
        // ---- cov2d = J covc J^T ----
        const float rz = p.rz, rz2 = rz * rz, rz3 = rz2 * rz;
        const float J00 = cam.fx * rz, J11 = cam.fy * rz, J02 = p.J02, J12 = p.J12;
        const float cxx = p.covc[0], cxy = p.covc[1], cxz = p.covc[2], cyy = p.covc[3], cyz = p.covc[4], czz = p.covc[5];
        // JS = J covc (2x3)
        const float js00 = J00 * cxx + J02 * cxz, js01 = J00 * cxy + J02 * cyz, js02 = J00 * cxz + J02 * czz;
        const float js10 = J11 * cxy + J12 * cxz, js11 = J11 * cyy + J12 * cyz, js12 = J11 * cyz + J12 * czz;
        // v_J = 2 * w * JS  (only the structurally non-zero entries)
        const float vJ00 = 2.f * (w00 * js00 + w01 * js10);
        const float vJ02 = 2.f * (w00 * js02 + w01 * js12);
        const float vJ11 = 2.f * (w01 * js01 + w11 * js11);
        const float vJ12 = 2.f * (w01 * js02 + w11 * js12);
        // v_covc = J^T w J (symmetric 3x3)
        const float a0 = w00 * J00, a1 = w01 * J11, a2 = w00 * J02 + w01 * J12;  // row 0 of (w J)
        const float b0 = w01 * J00, b1 = w11 * J11, b2 = w01 * J02 + w11 * J12;  // row 1 of (w J)
        float vS[9];
        vS[0] = J00 * a0, vS[1] = J00 * a1, vS[2] = J00 * a2;
        vS[3] = J11 * b0, vS[4] = J11 * b1, vS[5] = J11 * b2;
        vS[6] = J02 * a0 + J12 * b0, vS[7] = J02 * a1 + J12 * b1, vS[8] = J02 * a2 + J12 * b2;
        // ---- camera-space mean ----
        const float vm0 = a.v_means2d[i * 2], vm1 = a.v_means2d[i * 2 + 1];
        const float x = p.pc[0], y = p.pc[1], z = p.pc[2];
        const float tx = -J02 / (cam.fx * rz2), ty = -J12 / (cam.fy * rz2);
        v_pc[0] = cam.fx * rz * vm0;
        v_pc[1] = cam.fy * rz * vm1;
        v_pc[2] = -(cam.fx * x * vm0 + cam.fy * y * vm1) * rz2 + a.v_depths[i];
        if (p.in_x) v_pc[0] += -cam.fx * rz2 * vJ02; else v_pc[2] += -cam.fx * rz3 * vJ02 * tx;
        if (p.in_y) v_pc[1] += -cam.fy * rz2 * vJ12; else v_pc[2] += -cam.fy * rz3 * vJ12 * ty;
        v_pc[2] += -cam.fx * rz2 * vJ00 - cam.fy * rz2 * vJ11 + 2.f * cam.fx * tx * rz3 * vJ02 + 2.f * cam.fy * ty * rz3 * vJ12;
        (void)z;
        // ---- covc = M M^T, M = (Rcw Rm) diag(sc) ----
        float vM[9];
        {
          float sym[9];
  #pragma unroll
          for (int r = 0; r < 3; r++)
  #pragma unroll
            for (int c = 0; c < 3; c++) sym[r * 3 + c] = vS[r * 3 + c] + vS[c * 3 + r];
          mat3_mul(sym, p.M, vM);
        }
        float Wm[9], vW[9];
        mat3_mul(cam.R, Rm, Wm);
  #pragma unroll
        for (int r = 0; r < 3; r++)
  #pragma unroll
          for (int c = 0; c < 3; c++) vW[r * 3 + c] = vM[r * 3 + c] * sc[c];
  #pragma unroll
        for (int c = 0; c < 3; c++) v_sc[c] += Wm[c] * vM[c] + Wm[3 + c] * vM[3 + c] + Wm[6 + c] * vM[6 + c];
        mat3_mul_at(cam.R, vW, vRm);  // Rcw^T vW
        {
          float tmp[9];
          mat3_mul_bt(vW, Rm, tmp);    // vW Rm^T  -> dL/dRcw
  #pragma unroll
          for (int r = 0; r < 9; r++) v_view[r] += tmp[r];
  #pragma unroll
          for (int r = 0; r < 3; r++) {
  #pragma unroll
            for (int c = 0; c < 3; c++) v_view[r * 3 + c] += v_pc[r] * mw[c];
            v_view[9 + r] += v_pc[r];
          }
        }
      }
      float v_mw[3];
#pragma unroll
      for (int c = 0; c < 3; c++) v_mw[c] = cam.R[c] * v_pc[0] + cam.R[3 + c] * v_pc[1] + cam.R[6 + c] * v_pc[2];
      // ---- camera delta ----
      float v_mw0[3] = {v_mw[0], v_mw[1], v_mw[2]};
      if (a.in.RTs) {
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
        for (int c = 0; c < 3; c++) v_mu[c] += Rd[c] * v_mw0[0] + Rd[3 + c] * v_mw0[1] + Rd[6 + c] * v_mw0[2];
        float vRd[9];
        mat3_mul_bt(vRm, Rq, vRd);  // vRm Rq^T
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) vRd[r * 3 + c] += v_mw0[r] * mu[c];
        {
          float tmp[9];
          mat3_mul_at(Rd, vRm, tmp);  // Rd^T vRm
#pragma unroll
          for (int r = 0; r < 9; r++) v_Rq[r] += tmp[r];
        }
        // Gram-Schmidt adjoint (columns x, y, z of Rd)
        float vx[3] = {vRd[0], vRd[3], vRd[6]}, vy[3] = {vRd[1], vRd[4], vRd[7]}, vz[3] = {vRd[2], vRd[5], vRd[8]};
        // z = x cross y
        vx[0] += gs.y[1] * vz[2] - gs.y[2] * vz[1];
        vx[1] += gs.y[2] * vz[0] - gs.y[0] * vz[2];
        vx[2] += gs.y[0] * vz[1] - gs.y[1] * vz[0];
        vy[0] += vz[1] * gs.x[2] - vz[2] * gs.x[1];
        vy[1] += vz[2] * gs.x[0] - vz[0] * gs.x[2];
        vy[2] += vz[0] * gs.x[1] - vz[1] * gs.x[0];
        // y = bp * inv_nb
        float vbp[3];
        {
          const bool clamped = gs.inv_nb >= 1e12f;
          const float dty = clamped ? 0.f : (vy[0] * gs.y[0] + vy[1] * gs.y[1] + vy[2] * gs.y[2]);
#pragma unroll
          for (int c = 0; c < 3; c++) vbp[c] = (vy[c] - dty * gs.y[c]) * gs.inv_nb;
        }
        // bp = b - (b.x) x
        const float vbx = vbp[0] * gs.x[0] + vbp[1] * gs.x[1] + vbp[2] * gs.x[2];
        const float *bb = v9 + 6;
        float vb[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          vb[c] = vbp[c] - vbx * gs.x[c];
          vx[c] += -gs.d * vbp[c] - vbx * bb[c];
        }
        // x = a * inv_na
        float va_[3];
        {
          const bool clamped = gs.inv_na >= 1e12f;
          const float dtx = clamped ? 0.f : (vx[0] * gs.x[0] + vx[1] * gs.x[1] + vx[2] * gs.x[2]);
#pragma unroll
          for (int c = 0; c < 3; c++) va_[c] = (vx[c] - dtx * gs.x[c]) * gs.inv_na;
        }
        vec[0] = v_mw0[0], vec[1] = v_mw0[1], vec[2] = v_mw0[2];
        vec[3] = va_[0], vec[4] = va_[1], vec[5] = va_[2];
        vec[6] = vb[0], vec[7] = vb[1], vec[8] = vb[2];
        // coefficient gradients: v_c[k] += B_s[k] . vec[0:9]
        const float *B = Bs + s * K * 9;
        for (int k = 0; k < K; k++) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 9; j++) acc += B[k * 9 + j] * vec[j];
          vcf[tid * KP + k] += acc;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; c++) v_mu[c] += v_mw0[c];
#pragma unroll
        for (int r = 0; r < 9; r++) v_Rq[r] += vRm[r];
      }
    }
    // ---- block column sums for sub-sample s: 9K weighted by cf, 12 plain ----
    // Every wave reduces its 64 lanes with the permlane-swap ladder (common.h) and leaves its totals in its own
    // psum segment; the 4 segments are then added in fixed order -> deterministic.  (The first version walked the 256
    // per-thread vectors through LDS: 2 LDS reads per term, 25 % of the kernel.)
    if (dyn || a.in.RTs) {
      const int nk = dyn ? K * 9 : 0, no = nk + 12, nop = no + 1;  // +1: wave_sum_store's dump slot
      const int lane = tid & 63, seg = tid >> 6;
      // double-buffered over s: the buffer written now was last read two sub-samples ago, before the barrier of s - 1,
      // so ONE barrier per sub-sample (writes -> reads) is enough
      float *pb = psum + (s & 1) * 4 * nop;
      float *mine = pb + seg * nop;
      if (dyn) {
        if (dyn_block && K <= 8) {  // few bases: K x (9 multiplies + a 9-value wave reduction) on the VALU
          for (int k = 0; k < K; k++) {
            const float c = cf[tid * KP + k];
            float p[9];
#pragma unroll
            for (int jj = 0; jj < 9; jj++) p[jj] = c * vec[jj];
            wave_sum_store(p, mine + k * 9, lane);
          }
        } else if (dyn_block) {
          // v_Bs[k][j] = sum over the wave's 64 Gaussians of coef[g][k] * vec[g][j]: a [K x 64] x [64 x 9] product.  On the
          // matrix pipe: 16 v_mfma_f32_16x16x4_f32 per 16 bases (A = coefficients, B = vec, both read from LDS in the
          // operand layout); exact f32, fixed order.  Measured (MI355X): K = 6 167 vs 158 us on the VALU (10 of the 16
          // rows idle, a dependent MFMA chain at 2 waves per SIMD), K = 12 1011 vs 1095, K = 20 111 vs 119 - hence K > 8.
          // (Four independent accumulator chains need 19 more registers: 1 wave per SIMD, 246 us.)
          typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int jj = 0; jj < 9; jj++) svec[tid * 9 + jj] = vec[jj];
          __builtin_amdgcn_wave_barrier();  // a wave only reads the rows of its own 64 lanes
          const int wbase = tid & ~63, ln = lane & 15, lk = lane >> 4;
          for (int mt = 0; mt < K; mt += 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const bool am = mt + ln < K;
#pragma unroll 4
            for (int kk = 0; kk < 16; kk++) {
              const int src = wbase + 4 * kk + lk;  // the Gaussian (lane of this wave) this operand element belongs to
              const float av = am ? cf[src * KP + mt + ln] : 0.f;
              const float bv = ln < 9 ? svec[src * 9 + ln] : 0.f;
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
          __builtin_amdgcn_wave_barrier();
        } else {
          for (int o = lane; o < nk; o += 64) mine[o] = 0.f;
        }
      }
      {
        float q[12];
#pragma unroll
        for (int r = 0; r < 12; r++) q[r] = vec[9 + r];
        wave_sum_store(q, mine + nk, lane);
      }
      __syncthreads();
      for (int o = tid; o < no; o += BLK)
        part[s * no + o] = (pb[o] + pb[nop + o]) + (pb[2 * nop + o] + pb[3 * nop + o]);
    }
  }

  // ---- viewmat partials (12 plain column sums): the same per-wave ladder + fixed-order sum of the 4 segments ----
  {
    const int nk = dyn ? K * 9 : 0;
    __syncthreads();  // the last sub-sample's psum has been consumed
    wave_sum_store(v_view, psum + (tid >> 6) * 13, tid & 63);
    __syncthreads();
    if (tid < 12) part[S * (nk + 12) + tid] = (psum[tid] + psum[13 + tid]) + (psum[26 + tid] + psum[39 + tid]);
  }

  if (!active) return;
  // ---- per-Gaussian leaves ----
  a.g.v_means[g * 3] = v_mu[0], a.g.v_means[g * 3 + 1] = v_mu[1], a.g.v_means[g * 3 + 2] = v_mu[2];
  if (!pts) {
    const float w = qh[0], x = qh[1], y = qh[2], z = qh[3];
    const float *vR = v_Rq;
    float vq[4];
    vq[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
    vq[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[3] + vR[1]) + z * (vR[6] + vR[2]) + w * (vR[7] - vR[5]));
    vq[2] = 2.f * (x * (vR[3] + vR[1]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[7] + vR[5]) + w * (vR[2] - vR[6]));
    vq[3] = 2.f * (x * (vR[6] + vR[2]) + y * (vR[7] + vR[5]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    const float dot = vq[0] * w + vq[1] * x + vq[2] * y + vq[3] * z;
    float4 o4 = make_float4((vq[0] - dot * w) * inv_qn, (vq[1] - dot * x) * inv_qn, (vq[2] - dot * y) * inv_qn,
                            (vq[3] - dot * z) * inv_qn);
    *reinterpret_cast<float4 *>(a.g.v_quats + (size_t)g * 4) = o4;
  }
  if (!pts) {
#pragma unroll
    for (int j = 0; j < 3; j++) a.g.v_scales[g * 3 + j] = raw ? v_sc[j] * sc[j] : v_sc[j];
    const float o = a.opac_act[g];
    const float vo = a.v_opac_act[g];
    a.g.v_opacities[g] = raw ? vo * o * (1.f - o) : vo;
    const int D = d.D, DP = (D + 3) & ~3;
    for (int ch = 0; ch < D; ch++) {
      float v = a.v_ctab[(size_t)g * DP + ch];
      if ((d.flags & D4GS_RAW_COLORS) && ch < d.n_sigmoid) {
        const float c = a.ctab[(size_t)g * DP + ch];
        v *= c * (1.f - c);
      }
      a.g.v_colors[(size_t)g * D + ch] = v;
    }
  }
  if (isdyn) {  // softmax adjoint
    float dot = 0.f;
    for (int k = 0; k < K; k++) dot += cf[tid * KP + k] * vcf[tid * KP + k];
    for (int k = 0; k < K; k++)
      a.g.v_motion_coefs[(size_t)g * K + k] = cf[tid * KP + k] * (vcf[tid * KP + k] - dot);
  }
}

// sum the per-block partial vectors (deterministic): one workgroup per 4 outputs, 64 lanes stride the blocks,
// fixed-shape tree over the 64 partial sums.
__global__ void __launch_bounds__(256) k_reduce_partials(const float *partials, int n_blocks, int n, float *out) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (o < n)
    for (int b = lane; b < n_blocks; b += 64) acc += partials[(size_t)b * n + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (o < n && lane == 0) out[o] = acc;
}

// scatter the reduced shared gradients: v_Bs -> rots / transls / times ; camera deltas ; viewmat.
// One thread per LEAF element (basis k, frame, component): it walks the sub-samples in order and adds the (1 - w) share
// of those whose floor frame it is, then the w share of those whose ceil frame it is - independent loads, one store, no
// read-modify-write chains through memory (the first version's (k, j) owners did S dependent global updates: 9 us).
__global__ void __launch_bounds__(256) k_finish(const BwdArgs a, const float *red) {
  const D4gsDims &d = a.d;
  const int K = d.K, T = d.T, S = d.S;
  const bool dyn = d.G > 0;
  const int nk = dyn ? K * 9 : 0;
  const int stride = nk + 12;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
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

int n_shared_of(const D4gsDims *d) { return d->S * ((d->G > 0 ? d->K * 9 : 0) + 12) + 12; }

}  // namespace

extern "C" size_t d4gs_bwd_partials_elems(const D4gsDims *d) {
  const size_t blocks = ((size_t)d->N + BLK - 1) / BLK;
  return (blocks + 1) * (size_t)n_shared_of(d);
}

static int launch_project_bwd(BwdArgs &a, const D4gsDims *dims, const D4gsLeafGrads *grads, hipStream_t stream);

int d4gs_points_bwd_impl(const D4gsDims *dims, const D4gsProjIn *in, const float *v_points, const D4gsLeafGrads *grads,
                         hipStream_t stream) {
  BwdArgs a{};
  a.d = *dims;
  a.in = *in;
  a.g = *grads;
  a.v_points = v_points;
  return launch_project_bwd(a, dims, grads, stream);
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
  a.v_points = nullptr;
  return launch_project_bwd(a, dims, grads, stream);
}

static int launch_project_bwd(BwdArgs &a, const D4gsDims *dims, const D4gsLeafGrads *grads, hipStream_t stream) {
  a.n_shared = n_shared_of(dims);
  const int K = dims->G > 0 ? dims->K : 0;
  const int KP = K | 1;
  size_t lds = sizeof(float) * ((((size_t)dims->S * K * 9 + 3) & ~(size_t)3) + 2 * (size_t)BLK * KP +
                                8 * ((size_t)K * 9 + 13) + (size_t)BLK * 9);
  if (lds > 160 * 1024) {
    d4gs_set_error("project_bwd: LDS budget exceeded (S=%d K=%d)", dims->S, dims->K);
    return D4GS_EINVAL;
  }
  // per call: the attribute belongs to the (function, device) pair and a process may drive several GPUs
  (void)hipFuncSetAttribute((const void *)k_project_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int blocks = (dims->N + BLK - 1) / BLK;
  D4GS_LAUNCH("k_project_bwd", k_project_bwd, dim3(blocks), dim3(BLK), lds, stream, a);
  int rc = d4gs_check_launch("k_project_bwd");
  if (rc) return rc;
  float *red = grads->partials + (size_t)blocks * a.n_shared;
  D4GS_LAUNCH("k_reduce_partials", k_reduce_partials, dim3((a.n_shared + 3) / 4), dim3(256), 0, stream, grads->partials, blocks,
                     a.n_shared, red);
  int fin = dims->G > 0 ? dims->K * dims->T * 9 : 0;
  if (fin < dims->S * 12) fin = dims->S * 12;
  if (fin < 16) fin = 16;
  D4GS_LAUNCH("k_finish", k_finish, dim3((fin + 255) / 256), dim3(256), 0, stream, a, (const float *)red);
  return d4gs_check_launch("k_finish");
}
