// common.h -- device helpers shared by the d4gs HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/d4gs.h"

#define D4GS_WAVE 64
#define D4GS_MAX_K 32         // motion bases held in LDS per block
#define D4GS_PROJ_BLOCK 256


// host-side error plumbing (capi.cpp)
void d4gs_set_error(const char *fmt, ...);
int d4gs_check_launch(const char *what);


// 3x3 row-major helpers -----------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const float *A, const float *B, float *C) {  // C = A B
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_mul_bt(const float *A, const float *B, float *C) {  // C = A B^T
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      C[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
__device__ __forceinline__ void mat3_mul_at(const float *A, const float *B, float *C) {  // C = A^T B
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// unit quaternion (w,x,y,z) -> rotation matrix
__device__ __forceinline__ void quat_to_rotmat(float w, float x, float y, float z, float *R) {
  R[0] = 1.f - 2.f * (y * y + z * z);
  R[1] = 2.f * (x * y - w * z);
  R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z);
  R[4] = 1.f - 2.f * (x * x + z * z);
  R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y);
  R[7] = 2.f * (y * z + w * x);
  R[8] = 1.f - 2.f * (x * x + y * y);
}

// Gram-Schmidt of a 6-D rotation (transforms.py:41-53): columns x, y, z.  Keeps the pieces the adjoint needs.
struct GS6 {
  float x[3], y[3], z[3];
  float inv_na, inv_nb;  // 1/max(|a|,eps), 1/max(|b'|,eps)
  float d;               // b . x
};
__device__ __forceinline__ void gram_schmidt(const float *r6, GS6 &o) {
  float na = sqrtf(r6[0] * r6[0] + r6[1] * r6[1] + r6[2] * r6[2]);
  o.inv_na = 1.f / fmaxf(na, 1e-12f);
#pragma unroll
  for (int i = 0; i < 3; i++) o.x[i] = r6[i] * o.inv_na;
  o.d = r6[3] * o.x[0] + r6[4] * o.x[1] + r6[5] * o.x[2];
  float bp[3];
#pragma unroll
  for (int i = 0; i < 3; i++) bp[i] = r6[3 + i] - o.d * o.x[i];
  float nb = sqrtf(bp[0] * bp[0] + bp[1] * bp[1] + bp[2] * bp[2]);
  o.inv_nb = 1.f / fmaxf(nb, 1e-12f);
#pragma unroll
  for (int i = 0; i < 3; i++) o.y[i] = bp[i] * o.inv_nb;
  o.z[0] = o.x[1] * o.y[2] - o.x[2] * o.y[1];
  o.z[1] = o.x[2] * o.y[0] - o.x[0] * o.y[2];
  o.z[2] = o.x[0] * o.y[1] - o.x[1] * o.y[0];
}

// ---- pose compose on quaternions (flow3d/scene_model.py:94-102; roma 1.5.0 conventions, XYZW storage there) --------
// rotmat_to_unitquat: 4-way branch on argmax(R00, R11, R22, trace) (first maximum wins), candidate c (x, y, z, w),
// then c / |c|.  I = 0..2: the diagonal branches, I = 3: the trace branch.
__device__ __forceinline__ int rq_choice(const float *R) {
  const float tr = R[0] + R[4] + R[8];
  int ch = 0;
  float best = R[0];
  if (R[4] > best) best = R[4], ch = 1;
  if (R[8] > best) best = R[8], ch = 2;
  if (tr > best) ch = 3;
  return ch;
}
template <int I>
__device__ __forceinline__ void rq_cand(const float *R, float *c) {
  const float tr = R[0] + R[4] + R[8];
  if constexpr (I == 3) {
    c[0] = R[7] - R[5], c[1] = R[2] - R[6], c[2] = R[3] - R[1], c[3] = 1.f + tr;
  } else {
    constexpr int J = (I + 1) % 3, Kk = (I + 2) % 3;
    c[I] = 1.f - tr + 2.f * R[I * 3 + I];
    c[J] = R[J * 3 + I] + R[I * 3 + J];
    c[Kk] = R[Kk * 3 + I] + R[I * 3 + Kk];
    c[3] = R[Kk * 3 + J] - R[J * 3 + Kk];
  }
}
template <int I>
__device__ __forceinline__ void rq_cand_adj(const float *vc, float *vR) {  // vR += (d c / d R)^T vc
  if constexpr (I == 3) {
    vR[7] += vc[0], vR[5] -= vc[0], vR[2] += vc[1], vR[6] -= vc[1], vR[3] += vc[2], vR[1] -= vc[2];
    vR[0] += vc[3], vR[4] += vc[3], vR[8] += vc[3];
  } else {
    constexpr int J = (I + 1) % 3, Kk = (I + 2) % 3;
    vR[I * 3 + I] += vc[I], vR[J * 3 + J] -= vc[I], vR[Kk * 3 + Kk] -= vc[I];
    vR[J * 3 + I] += vc[J], vR[I * 3 + J] += vc[J];
    vR[Kk * 3 + I] += vc[Kk], vR[I * 3 + Kk] += vc[Kk];
    vR[Kk * 3 + J] += vc[3], vR[J * 3 + Kk] -= vc[3];
  }
}
__device__ __forceinline__ void rq_cand_dyn(int ch, const float *R, float *c) {
  if (ch == 0) rq_cand<0>(R, c);
  else if (ch == 1) rq_cand<1>(R, c);
  else if (ch == 2) rq_cand<2>(R, c);
  else rq_cand<3>(R, c);
}
__device__ __forceinline__ void rq_cand_adj_dyn(int ch, const float *vc, float *vR) {
  if (ch == 0) rq_cand_adj<0>(vc, vR);
  else if (ch == 1) rq_cand_adj<1>(vc, vR);
  else if (ch == 2) rq_cand_adj<2>(vc, vR);
  else rq_cand_adj<3>(vc, vR);
}
// Hamilton product r = p (x) q, all (w, x, y, z)
__device__ __forceinline__ void quat_mul(const float *p, const float *q, float *r) {
  r[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
  r[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
  r[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
  r[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}
__device__ __forceinline__ void quat_mul_adj(const float *p, const float *q, const float *vr, float *vp, float *vq) {
  vp[0] = vr[0] * q[0] + vr[1] * q[1] + vr[2] * q[2] + vr[3] * q[3];
  vp[1] = -vr[0] * q[1] + vr[1] * q[0] - vr[2] * q[3] + vr[3] * q[2];
  vp[2] = -vr[0] * q[2] + vr[1] * q[3] + vr[2] * q[0] - vr[3] * q[1];
  vp[3] = -vr[0] * q[3] - vr[1] * q[2] + vr[2] * q[1] + vr[3] * q[0];
  vq[0] = vr[0] * p[0] + vr[1] * p[1] + vr[2] * p[2] + vr[3] * p[3];
  vq[1] = -vr[0] * p[1] + vr[1] * p[0] + vr[2] * p[3] - vr[3] * p[2];
  vq[2] = -vr[0] * p[2] - vr[1] * p[3] + vr[2] * p[0] + vr[3] * p[1];
  vq[3] = -vr[0] * p[3] + vr[1] * p[2] - vr[2] * p[1] + vr[3] * p[0];
}
// compose: q_out = normalize( rotmat_to_unitquat(Rd) (x) qh ), everything (w, x, y, z); qh is the unit Gaussian quaternion.
// Keeps what the adjoint needs.
struct PoseQ {
  int ch;
  float p[4];      // unit quaternion of Rd (w, x, y, z)
  float inv_nc;    // 1 / |candidate|
  float out[4];    // the result
  float inv_nr;    // 1 / max(|product|, 1e-12)
};
__device__ __forceinline__ void pose_quat(const float *Rd, const float *qh, PoseQ &o) {
  float c[4];
  o.ch = rq_choice(Rd);
  rq_cand_dyn(o.ch, Rd, c);
  o.inv_nc = 1.f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
  o.p[0] = c[3] * o.inv_nc, o.p[1] = c[0] * o.inv_nc, o.p[2] = c[1] * o.inv_nc, o.p[3] = c[2] * o.inv_nc;
  float r[4];
  quat_mul(o.p, qh, r);
  o.inv_nr = 1.f / fmaxf(sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]), 1e-12f);
#pragma unroll
  for (int i = 0; i < 4; i++) o.out[i] = r[i] * o.inv_nr;
}
// adjoint: v_out (w,x,y,z) -> vRd += ..., v_qh += ...
__device__ __forceinline__ void pose_quat_adj(const PoseQ &o, const float *qh, const float *v_out, float *vRd, float *v_qh) {
  const float dt = v_out[0] * o.out[0] + v_out[1] * o.out[1] + v_out[2] * o.out[2] + v_out[3] * o.out[3];
  float vr[4], vp[4], vq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) vr[i] = (v_out[i] - dt * o.out[i]) * o.inv_nr;
  quat_mul_adj(o.p, qh, vr, vp, vq);
#pragma unroll
  for (int i = 0; i < 4; i++) v_qh[i] += vq[i];
  const float dp = vp[0] * o.p[0] + vp[1] * o.p[1] + vp[2] * o.p[2] + vp[3] * o.p[3];
  float vc[4];  // candidate order (x, y, z, w)
  vc[0] = (vp[1] - dp * o.p[1]) * o.inv_nc, vc[1] = (vp[2] - dp * o.p[2]) * o.inv_nc;
  vc[2] = (vp[3] - dp * o.p[3]) * o.inv_nc, vc[3] = (vp[0] - dp * o.p[0]) * o.inv_nc;
  rq_cand_adj_dyn(o.ch, vc, vRd);
}

// gsplat isect_tiles rectangle: tile_min inclusive, tile_max exclusive (float32 arithmetic as upstream)
__device__ __forceinline__ void tile_rect(float mx, float my, int radius, int tw, int th, int &x0, int &y0, int &x1,
                                          int &y1) {
  const float inv = 1.0f / D4GS_TILE;
  float tx = mx * inv, ty = my * inv, tr = (float)radius * inv;
  x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tw);
  y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)th);
  x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tw);
  y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)th);
}

// D4GS_EXACT_CULL: shrink the gsplat tile rectangle to the tiles that contain a pixel centre inside the ellipse
// sigma <= tau, tau = ln(255*opacity) (alpha = opacity*exp(-sigma) >= 1/255).  Half-extents of that ellipse are
// sqrt(2 tau Sigma_xx), sqrt(2 tau Sigma_yy) with Sigma the blurred 2-D covariance.  tau carries a 1% + 0.02 margin
// and the extents 1e-3 px so fp32 rounding of sigma / exp in the rasterizer can never admit a pixel outside it.
__device__ __forceinline__ void tight_rect(float mx, float my, float opac, float cov_xx, float cov_yy, int &x0, int &y0,
                                           int &x1, int &y1) {
  const float tau = __logf(255.f * opac) * 1.01f + 0.02f;
  if (!(tau > 0.f)) {
    x1 = x0, y1 = y0;
    return;
  }
  const float ex = sqrtf(2.f * tau * cov_xx) + 1e-3f, ey = sqrtf(2.f * tau * cov_yy) + 1e-3f;
  const float inv = 1.0f / D4GS_TILE;
  // pixel j has centre j + 0.5; pixels with |mx - (j+0.5)| <= ex
  const float jx0 = ceilf(mx - ex - 0.5f), jx1 = floorf(mx + ex - 0.5f);
  const float jy0 = ceilf(my - ey - 0.5f), jy1 = floorf(my + ey - 0.5f);
  const int tx0 = (int)fmaxf(floorf(jx0 * inv), (float)x0), tx1 = (int)fminf(floorf(jx1 * inv) + 1.f, (float)x1);
  const int ty0 = (int)fmaxf(floorf(jy0 * inv), (float)y0), ty1 = (int)fminf(floorf(jy1 * inv) + 1.f, (float)y1);
  x0 = tx0, x1 = max(tx1, tx0), y0 = ty0, y1 = max(ty1, ty0);
  if (x1 == x0 || y1 == y0) x1 = x0, y1 = y0;
}

// D4GS_EXACT_TILES: does the ellipse sigma(p) = (a u^2 + c v^2) / 2 + b u v <= tau, (u, v) = p - centre, reach the rectangle of pixel
// centres [X0, X1] x [Y0, Y1]?  sigma is convex: if the centre is outside the rectangle the minimum over it lies on an edge facing the
// centre, where sigma is a 1-D parabola - minimise it, clamp to the edge, evaluate.  tau is tight_rect's (1 % + 0.02 margin): the
// continuous rectangle is a superset of its pixel centres, so no tile holding a pixel with alpha >= 1/255 is ever dropped.
__device__ __forceinline__ bool d4gs_ellipse_hits_rect(float mx, float my, float a, float b, float c, float tau, float X0, float X1,
                                                       float Y0, float Y1) {
  const float U0 = X0 - mx, U1 = X1 - mx, V0 = Y0 - my, V1 = Y1 - my;
  const float uc = fminf(fmaxf(0.f, U0), U1), vc = fminf(fmaxf(0.f, V0), V1);  // the rectangle's point nearest to the centre, per axis
  if (uc == 0.f && vc == 0.f) return true;
  float best = 3.0e38f;
  if (uc != 0.f) {  // the vertical edge u = uc: v* = -b uc / c
    const float v = fminf(fmaxf(-b * uc * __builtin_amdgcn_rcpf(c), V0), V1);
    best = 0.5f * (a * uc * uc + c * v * v) + b * uc * v;
  }
  if (vc != 0.f) {  // the horizontal edge v = vc: u* = -b vc / a
    const float u = fminf(fmaxf(-b * vc * __builtin_amdgcn_rcpf(a), U0), U1);
    best = fminf(best, 0.5f * (a * u * u + c * vc * vc) + b * u * vc);
  }
  return best <= tau * 1.001f + 1e-4f;  // (the clamped 1-D minimiser is computed with an approximate reciprocal: a hair of slack)
}

// camera constants, uniform across the grid (passed by value as a kernel argument -> SGPRs)
struct Cam {
  float R[9];  // world->camera rotation
  float t[3];
  float fx, fy, cx, cy;
  float limx, limy;  // 1.3 * tan(fov/2)
};

// viewmat [4,4] / K [3,3] live in device memory (viewmat may carry a gradient); addresses are grid-uniform so
// these become scalar loads.
__device__ __forceinline__ Cam load_cam(const float *V, const float *Km, int width, int height) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) c.R[i * 3 + j] = V[i * 4 + j];
    c.t[i] = V[i * 4 + 3];
  }
  c.fx = Km[0], c.fy = Km[4], c.cx = Km[2], c.cy = Km[5];
  c.limx = 1.3f * (0.5f * (float)width / c.fx);
  c.limy = 1.3f * (0.5f * (float)height / c.fy);
  return c;
}

// Perspective projection of one instance (gsplat fully_fused_projection).
// Rm = instance rotation (world), sc = scales, mw = world mean AFTER the camera delta.
struct ProjOut {
  float pc[3];      // camera-space mean
  float M[9];       // Rcw * Rm * diag(sc)   (camera-space "sqrt" of the covariance)
  float covc[6];    // camera-space covariance (xx,xy,xz,yy,yz,zz)
  float J02, J12;   // -fx*tx/z^2, -fy*ty/z^2
  float rz;
  float a, b, c;    // blurred 2-D covariance
  float det;
  float mx, my;
  int radius;       // 0 = culled
  bool in_x, in_y;  // inside the 1.3*tan(fov) clamp
};

__device__ __forceinline__ void project_instance(const Cam &cam, const float *mw, const float *Rm, const float *sc,
                                                 const D4gsDims &d, ProjOut &o) {
  o.radius = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) o.pc[i] = cam.R[i * 3] * mw[0] + cam.R[i * 3 + 1] * mw[1] + cam.R[i * 3 + 2] * mw[2] + cam.t[i];
  float z = o.pc[2];
  if (!(z >= d.near_plane && z <= d.far_plane)) return;
  float W_[9];
  mat3_mul(cam.R, Rm, W_);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) o.M[i * 3 + j] = W_[i * 3 + j] * sc[j];
  const float *M = o.M;
  o.covc[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
  o.covc[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  o.covc[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
  o.covc[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  o.covc[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
  o.covc[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
  float rz = 1.f / z, rz2 = rz * rz;
  o.rz = rz;
  float xr = o.pc[0] * rz, yr = o.pc[1] * rz;
  o.in_x = (xr <= cam.limx) && (xr >= -cam.limx);
  o.in_y = (yr <= cam.limy) && (yr >= -cam.limy);
  float tx = z * fminf(cam.limx, fmaxf(-cam.limx, xr));
  float ty = z * fminf(cam.limy, fmaxf(-cam.limy, yr));
  float J00 = cam.fx * rz, J11 = cam.fy * rz;
  o.J02 = -cam.fx * tx * rz2;
  o.J12 = -cam.fy * ty * rz2;
  // cov2d = J covc J^T,  J = [[J00,0,J02],[0,J11,J12]]
  const float cxx = o.covc[0], cxy = o.covc[1], cxz = o.covc[2], cyy = o.covc[3], cyz = o.covc[4], czz = o.covc[5];
  float c00 = J00 * J00 * cxx + 2.f * J00 * o.J02 * cxz + o.J02 * o.J02 * czz;
  float c01 = J00 * J11 * cxy + J00 * o.J12 * cxz + o.J02 * J11 * cyz + o.J02 * o.J12 * czz;
  float c11 = J11 * J11 * cyy + 2.f * J11 * o.J12 * cyz + o.J12 * o.J12 * czz;
  o.a = c00 + d.eps2d;
  o.b = c01;
  o.c = c11 + d.eps2d;
  o.det = o.a * o.c - o.b * o.b;
  if (!(o.det > 0.f)) return;
  float mid = 0.5f * (o.a + o.c);
  float v1 = mid + sqrtf(fmaxf(0.01f, mid * mid - o.det));
  float radius = ceilf(3.f * sqrtf(v1));
  if (radius <= d.radius_clip) return;
  o.mx = cam.fx * o.pc[0] * rz + cam.cx;
  o.my = cam.fy * o.pc[1] * rz + cam.cy;
  if (o.mx + radius <= 0.f || o.mx - radius >= (float)d.width || o.my + radius <= 0.f ||
      o.my - radius >= (float)d.height)
    return;
  o.radius = (int)radius;
}

// ---- depth segments of the tile lists (round 4) ---------------------------------------------------------------------
// A rank of an exposure-sharded frame composites S / P sub-samples: at S = 1 and 288x512 that is 576 tiles = 2 304 waves for 8 192
// wave slots, and the composite kernels' time goes with 1 / occupancy.  The BACKWARD therefore splits every tile list into up to
// D4GS_SEG_MAX depth segments that separate workgroups replay in parallel: the forward leaves, per pixel, the transmittance and
// the accumulated channels at every segment boundary (+ the final ones), from which a segment's workgroup starts exactly where
// the sequential replay would be (T at the boundary; the suffix dot product <v_out, C_final - C_boundary>).
// Segment length of a list of `len` entries: a multiple of 256 (= the forward's batch or twice it), at most D4GS_SEG_MAX segments.
#ifndef D4GS_SEG_MAX  // (overridable for A/B builds: scripts/ab_run.sh)
#define D4GS_SEG_MAX 8
#endif
#ifndef D4GS_SEG_UNIT
#define D4GS_SEG_UNIT 256
#endif
#ifndef D4GS_SEG_TILES_MAX
#define D4GS_SEG_TILES_MAX 1280  // S * tiles above this: the plain kernels already fill the machine (>= 5 waves per SIMD)
#endif
// Round 6: renders of <= 4 colour channels keep their segments up to 2.5 rounds of the chip's 2048 workgroup slots.  The workgroup
// trace of cfg2 (4608 tiles; profiles/r06_trace_bwd_cfg2.txt) shows full residency until the last workgroup is dispatched and then a
// linear drain that lasts one workgroup LIFETIME (p50 233 us of a 620-us kernel) at ~77 % of the full rate; a quarter-length unit drains
// in a quarter of the time.  Measured, plain vs segments (profiles/r06_ab_seg_threshold.txt): 3456 tiles 491 -> 437 us, 4608 619 -> 580
// (forward +4: it writes the boundary states), 6912 880 -> 873 (+6), 9216 and up slower - the states cost in proportion to the tiles,
// the drain does not.  The 17-channel kernels lose at every size above the rank shares (18 floats of state per pixel and boundary).
#ifndef D4GS_SEG_TILES_MAX_NARROW
#define D4GS_SEG_TILES_MAX_NARROW 5120
#endif
__host__ __device__ __forceinline__ int d4gs_seg_len(int len) {
  const int per = (len + D4GS_SEG_UNIT * D4GS_SEG_MAX - 1) / (D4GS_SEG_UNIT * D4GS_SEG_MAX);
  return D4GS_SEG_UNIT * (per > 1 ? per : 1);
}
// floats of D4gsRaster.seg_state for one configuration (0: segments are not used for it)
int64_t d4gs_seg_state_elems(const D4gsDims *d);
// do the composite kernels of this launch use segments?  (the forward writes the boundary states, the backward replays by segment)
bool d4gs_seg_on(const D4gsDims *d, const D4gsIsect *isect, const D4gsRaster *r);

// ---- D4GS_LAZY_SORT: near / far partition of the tile lists (include/d4gs.h) ----------------------------------------
// lazy_ws (int32): hist [T][NB] (tile, depth bucket) counts | near [T] keys in the near part | pivot [T] last near bucket |
// cur [T][2] near / far slot cursors | flag [T] 1 = the tile did not saturate within its near part | zr [S][2] depth range of the
// sub-sample as float bits (zr[0] = ~min bits, zr[1] = max bits: both grow under atomicMax from a zeroed buffer).  T = S * tiles.
struct LazyWs {
  int32_t *hist, *near, *pivot, *cur, *flag;
  uint32_t *zr;
  int nb;
};
__host__ __device__ __forceinline__ int d4gs_lazy_buckets(int tiles_per_subsample) { return tiles_per_subsample <= 2048 ? 8 : 4; }
__host__ __device__ __forceinline__ int64_t d4gs_lazy_ws_elems(int S, int tiles_per_subsample) {
  const int64_t T = (int64_t)S * tiles_per_subsample;
  return T * (d4gs_lazy_buckets(tiles_per_subsample) + 5) + 2 * (int64_t)S + 8;
}
__host__ __device__ __forceinline__ LazyWs d4gs_lazy_carve(int32_t *ws, int S, int tiles_per_subsample) {
  LazyWs w;
  const int64_t T = (int64_t)S * tiles_per_subsample;
  w.nb = d4gs_lazy_buckets(tiles_per_subsample);
  w.hist = ws, w.near = w.hist + T * w.nb, w.pivot = w.near + T, w.cur = w.pivot + T, w.flag = w.cur + 2 * T;
  w.zr = reinterpret_cast<uint32_t *>(w.flag + T);
  return w;
}
// depth bucket of an instance: piecewise linear in the float's bit pattern (monotone in the depth, no transcendental, the same
// integer arithmetic wherever it is evaluated) between the sub-sample's smallest and largest visible depth
__device__ __forceinline__ int d4gs_depth_bucket(float depth, uint32_t zmin_bits, uint32_t zmax_bits, int nb) {
  const uint32_t b = __float_as_uint(depth);
  const uint64_t span = (uint64_t)(zmax_bits - zmin_bits) + 1u;
  const uint64_t off = b > zmin_bits ? (uint64_t)(b - zmin_bits) : 0u;
  const int k = (int)((off * (uint64_t)nb) / span);
  return k < nb ? k : nb - 1;
}

// The exposure blend's adjoint (blend.hip k_blend_bwd) folded into the composite backward's prologue by the one-call path (frame.hip):
// a pixel of sub-sample s takes v_blended / S on the mean channels, the whole of it on a max / min channel iff s is the first
// sub-sample (of the first S - 1) whose value equals the blended one (k_blend_fwd leaves that index), and v_acc / S on alpha - k_blend_bwd's expressions, bit for bit.
struct BlendAdj {
  const float *v_blended;  // [H,W,channels] or NULL
  const float *v_acc;      // [H,W] or NULL
  const int8_t *win;       // [H,W,channels] max / min channels: the sub-sample the gradient goes to, -1 = the mean (k_blend_fwd)
  uint64_t non_mean;       // bit c: channel c is a max / min channel
};
bool d4gs_lazy_on(const D4gsDims *d, const D4gsProjOut *out);
int d4gs_lazy_pivot_launch(const D4gsDims *d, const D4gsProjOut *out, int64_t near_target, hipStream_t stream);
int d4gs_lazy_far_sort(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, hipStream_t stream);
// [S][K][16] time-blended bases (project_bwd.hip: k_bases_table), for callers of d4gs_project_bwd without D4gsProjOut.blend_bases
int d4gs_bases_table_launch(const D4gsDims *dims, const D4gsProjIn *in, float *btab, hipStream_t stream);
// Instances per lane of the 1024-lane blocks of k_count_tiles / k_emit (their chunks must agree): 4 - or 1 when 4 would leave
// fewer blocks than CUs (one or two sub-samples of a few 100 k Gaussians: S = 1, N = 300 k gave 74 blocks for 256 CUs).
int d4gs_chunk_per_thread(const D4gsDims *d);
// > 0: the fused scan is in effect for this configuration and this is its chunk count per sub-sample (project_fwd.hip)
int d4gs_fused_scan_chunks(const D4gsDims *d);

// Optional per-kernel HIP-event profiler (off by default; used by bench.py for the roofline object).
struct ProfScope {
  int slot;
  hipStream_t stream;
  ProfScope(const char *name, hipStream_t s);
  ~ProfScope();
};

// sigma * log2(e) of a splat at a pixel offset (dx, dy) in 5 operations.  g1 = (a/2, b, c/2) * log2(e) as staged by
// the raster kernels (stage_conic).  Used verbatim by every forward variant AND the backward's alpha recomputation:
// explicit FMAs with contraction off, so every kernel (and every unrolled copy of a loop body) rounds identically -
// the forward/backward valid-pixel decisions and the "variants bit-identical", "exact cull on == off" guarantees
// depend on it.
__device__ __forceinline__ float splat_sigma2(const float4 g1, float dx, float dy) {
#pragma clang fp contract(off)
  const float inner = __builtin_fmaf(g1.x, dx, g1.y * dy);  // a/2 dx + b dy
  return __builtin_fmaf(dx, inner, (g1.z * dy) * dy);       // dx (a/2 dx + b dy) + c/2 dy^2
}
// LDS record of a conic for splat_sigma2; w is free for the caller
__device__ __forceinline__ float4 stage_conic(float a, float b, float c, float w) {
  constexpr float LOG2E_ = 1.4426950408889634f;
  return make_float4(a * (0.5f * LOG2E_), b * LOG2E_, c * (0.5f * LOG2E_), w);
}

// ---- fused DPP adds ------------------------------------------------------------------------------------------
// hipcc lowers `x + update_dpp(x)` to v_mov_b32_dpp + v_add_f32 (2 issue slots per step).  The asm blocks below
// issue the fused v_add_f32_dpp instead.  Inside one block consecutive steps on the same register are >= 3
// instructions apart (DPP read-after-VALU-write needs 2 wait states); the short blocks pad with s_nop.  The
// leading s_nop covers a compiler VALU write right before the block.
#define D4GS_C1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define D4GS_C2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define D4GS_C3 "row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define D4GS_C4 "row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define D4GS_C5 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define D4GS_C6 "row_bcast:31 row_mask:0xc bank_mask:0xf"
#define D4GS_DA(i, c) "v_add_f32_dpp %" #i ", %" #i ", %" #i " " c "\n\t"
#define D4GS_S2(c) D4GS_DA(0, c) D4GS_DA(1, c) "s_nop 0\n\t"
#define D4GS_S1(c) D4GS_DA(0, c) "s_nop 1\n\t"

// ---- wave reduction with gfx950 permlane swaps ---------------------------------------------------------------
// v_permlane32_swap / v_permlane16_swap fold TWO registers per instruction, so R per-lane values are reduced in
// ~2.6 R instructions instead of 6 R:   4 values -> 3 swaps + 3 adds + 4 row steps on ONE register whose 16-lane
// rows then hold (v0, v2, v1, v3);  2 values -> 1 swap + 1 add + 4 row steps + row_bcast:15 (rows 1 / 3 hold
// v0 / v1);  1 value -> the 6-step DPP ladder (row 3 holds it).  Lanes 0/16/32/48 then scatter the totals.
__device__ __forceinline__ float swap32_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
#define D4GS_R3(c) D4GS_DA(0, c) D4GS_DA(1, c) D4GS_DA(2, c)
__device__ __forceinline__ void rowsum3(float &a, float &b, float &c) {
  asm volatile("s_nop 1\n\t" D4GS_R3(D4GS_C1) D4GS_R3(D4GS_C2) D4GS_R3(D4GS_C3) D4GS_R3(D4GS_C4)
               : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void rowsum2(float &a, float &b) {
  asm volatile("s_nop 1\n\t" D4GS_S2(D4GS_C1) D4GS_S2(D4GS_C2) D4GS_S2(D4GS_C3) D4GS_S2(D4GS_C4) : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void rowsum1(float &a) {
  asm volatile("s_nop 1\n\t" D4GS_S1(D4GS_C1) D4GS_S1(D4GS_C2) D4GS_S1(D4GS_C3) D4GS_S1(D4GS_C4) : "+v"(a));
}
__device__ __forceinline__ void bcast15(float &a) {
  asm volatile("s_nop 1\n\t" D4GS_DA(0, D4GS_C5) : "+v"(a));
}
__device__ __forceinline__ void bcast15_31(float &a) {
  asm volatile("s_nop 1\n\t" D4GS_DA(0, D4GS_C5) "s_nop 1\n\t" D4GS_DA(0, D4GS_C6) : "+v"(a));
}
template <int NR>
__device__ __forceinline__ void rowsum_all(float (&z)[NR]) {
  constexpr int n3 = NR / 3, r = NR % 3;
#pragma unroll
  for (int i = 0; i < n3; i++) rowsum3(z[3 * i], z[3 * i + 1], z[3 * i + 2]);
  if constexpr (r == 2) rowsum2(z[NR - 2], z[NR - 1]);
  if constexpr (r == 1) rowsum1(z[NR - 1]);
}
// Sum v[0..R) over the wave and store the R totals to arr[at .. at + R) (LDS); arr[at + R] must be a writable pad
// slot.  `arr` is the __shared__ array itself and `at` an int offset so that the address math stays 32-bit.
// Three folded registers (R = 9 / 10: the composite backward's rows, k_project_bwd's per-basis sums) share ONE 16-lane ladder
// (round 5): after the first quad step the odd lanes of z0 take z1's pair sums, after the second lane 3 of every quad takes z2's
// quad sum, and the remaining two steps rotate by whole quads (row_ror:4 / :8 keep lane & 3) - 10 DPP / select instructions
// instead of 13, one LDS store instead of three.  Slot lane & 3 of every row then holds: 0 -> z0, 1 -> z1, 3 -> z2 (whose rows the
// row_bcast steps then combine in lanes 12..15).
#ifndef D4GS_PACKED_LADDER
#define D4GS_PACKED_LADDER 1
#endif
#define D4GS_C7 "row_ror:4 row_mask:0xf bank_mask:0xf"
#define D4GS_C8 "row_ror:8 row_mask:0xf bank_mask:0xf"
#define D4GS_C5S "row_bcast:15 row_mask:0xa bank_mask:0x8"
#define D4GS_C6S "row_bcast:31 row_mask:0xc bank_mask:0x8"
template <bool SINGLE /* z2 is one value (rows 0..3 -> row 3) instead of two (rows 0,1 -> 1; 2,3 -> 3) */>
__device__ __forceinline__ float rowsum3_packed(float a, float b, float c) {
  const uint64_t odd = 0xaaaaaaaaaaaaaaaaull, l3 = 0x8888888888888888ull;
  if constexpr (SINGLE)
    asm volatile("s_nop 1\n\t" D4GS_R3(D4GS_C1) "v_cndmask_b32_e64 %0, %0, %1, %3\n\ts_nop 0\n\t" D4GS_DA(2, D4GS_C2) D4GS_DA(0, D4GS_C2)
                 "v_cndmask_b32_e64 %0, %0, %2, %4\n\ts_nop 1\n\t" D4GS_DA(0, D4GS_C7) "s_nop 1\n\t" D4GS_DA(0, D4GS_C8)
                 "s_nop 1\n\t" D4GS_DA(0, D4GS_C5S) "s_nop 1\n\t" D4GS_DA(0, D4GS_C6S)
                 : "+v"(a), "+v"(b), "+v"(c)
                 : "s"(odd), "s"(l3));
  else
    asm volatile("s_nop 1\n\t" D4GS_R3(D4GS_C1) "v_cndmask_b32_e64 %0, %0, %1, %3\n\ts_nop 0\n\t" D4GS_DA(2, D4GS_C2) D4GS_DA(0, D4GS_C2)
                 "v_cndmask_b32_e64 %0, %0, %2, %4\n\ts_nop 1\n\t" D4GS_DA(0, D4GS_C7) "s_nop 1\n\t" D4GS_DA(0, D4GS_C8)
                 "s_nop 1\n\t" D4GS_DA(0, D4GS_C5S)
                 : "+v"(a), "+v"(b), "+v"(c)
                 : "s"(odd), "s"(l3));
  return a;
}
// Two folded registers (R = 6 / 7: the VALU rows of the 17-channel composite backward): after the first quad step the odd lanes take
// the second register - 6 (7) instructions instead of 9 (15).  PAIR: the second register holds two values (rows 0,1 / 2,3), combined
// by a row_bcast step in lanes 12..15; otherwise it already holds one value per row.
template <bool PAIR>
__device__ __forceinline__ float rowsum2_packed(float a, float b) {
  const uint64_t odd = 0xaaaaaaaaaaaaaaaaull;
  if constexpr (PAIR)
    asm volatile("s_nop 1\n\t" D4GS_DA(0, D4GS_C1) D4GS_DA(1, D4GS_C1) "v_cndmask_b32_e64 %0, %0, %1, %2\n\ts_nop 1\n\t" D4GS_DA(0, D4GS_C2)
                 "s_nop 1\n\t" D4GS_DA(0, D4GS_C7) "s_nop 1\n\t" D4GS_DA(0, D4GS_C8) "s_nop 1\n\t" D4GS_DA(0, D4GS_C5S)
                 : "+v"(a), "+v"(b)
                 : "s"(odd));
  else
    asm volatile("s_nop 1\n\t" D4GS_DA(0, D4GS_C1) D4GS_DA(1, D4GS_C1) "v_cndmask_b32_e64 %0, %0, %1, %2\n\ts_nop 1\n\t" D4GS_DA(0, D4GS_C2)
                 "s_nop 1\n\t" D4GS_DA(0, D4GS_C7) "s_nop 1\n\t" D4GS_DA(0, D4GS_C8)
                 : "+v"(a), "+v"(b)
                 : "s"(odd));
  return a;
}
// Slab slots written for R values: R, except R = 7 through the packed ladder, whose last value arrives as two partial sums in slots 6
// and 7 (the reader adds them).
template <int R>
constexpr bool d4gs_wave_sum_split_last() { return D4GS_PACKED_LADDER && R == 7; }
template <int R>
__device__ __forceinline__ void wave_sum_store(float (&v)[R], float *arr, int at, int lane) {
  constexpr int n4 = R / 4, rem = R % 4, n2 = rem / 2, n1 = rem % 2, NR = n4 + n2 + n1;
  if constexpr (D4GS_PACKED_LADDER && n4 == 1 && n2 == 1) {  // R = 6, 7
    const float z0 = swap16_add(swap32_add(v[0], v[1]), swap32_add(v[2], v[3]));  // rows (v0, v2, v1, v3)
    const float t = swap32_add(v[4], v[5]);                                        // lanes 0..31 v4, 32..63 v5
    const int r = lane >> 4, k = lane & 15;
    const int o4 = ((r & 1) << 1) | (r >> 1);
    int ats = at;
    asm volatile("" : "+s"(ats));
    if constexpr (n1) {
      // the seventh value rides unfolded: rows (v4, v6 lower half, v5, v6 upper half) -> slots 4, 6, 5, 7
      const float q = rowsum2_packed<false>(z0, swap16_add(t, v[6]));
      if (k < 2) arr[ats + (k == 0 ? o4 : (r == 0 ? 4 : r == 1 ? 6 : r == 2 ? 5 : 7))] = q;
    } else {
      const float q = rowsum2_packed<true>(z0, t);
      if (k == 0 || (k == 15 && (r & 1))) arr[ats + (k == 0 ? o4 : 4 + (r >> 1))] = q;
    }
    return;
  }
  float z[NR];
#pragma unroll
  for (int g = 0; g < n4; g++)
    z[g] = swap16_add(swap32_add(v[4 * g], v[4 * g + 1]), swap32_add(v[4 * g + 2], v[4 * g + 3]));
  if constexpr (n2) z[n4] = swap32_add(v[4 * n4], v[4 * n4 + 1]);
  if constexpr (n1) z[NR - 1] = v[R - 1];
  if constexpr (D4GS_PACKED_LADDER && n4 == 2 && n2 + n1 == 1) {
    const float q = rowsum3_packed<n1 == 1>(z[0], z[1], z[2]);
    const int r = lane >> 4, k = lane & 15;
    const int o4 = ((r & 1) << 1) | (r >> 1);  // rows hold (v0, v2, v1, v3)
    // (z2's totals are taken from lane 15 of the row: bank_mask 0x8 = lanes 12..15 of a row are the ones the row_bcast steps write)
    const bool w = k < 2 || (k == 15 && (n1 ? r == 3 : (r & 1)));
    const int off = k < 2 ? 4 * k + o4 : 8 + (n1 ? 0 : (r >> 1));
    int ats = at;  // (wave-uniform at every call site; pinned to an SGPR so that the row offset is not folded into a 64-bit VALU mad)
    asm volatile("" : "+s"(ats));
    if (w) arr[ats + off] = q;
    return;
  }
  rowsum_all(z);
  if constexpr (n2) bcast15(z[n4]);
  if constexpr (n1) bcast15_31(z[NR - 1]);
  if ((lane & 15) == 0) {
    const int r = lane >> 4;
    const int o4 = ((r & 1) << 1) | (r >> 1);  // rows hold (v0, v2, v1, v3)
#pragma unroll
    for (int g = 0; g < n4; g++) arr[at + 4 * g + o4] = z[g];
    if constexpr (n2) arr[at + ((r & 1) ? 4 * n4 + (r >> 1) : R)] = z[n4];
    if constexpr (n1) arr[at + (r == 3 ? R - 1 : R)] = z[NR - 1];
  }
}
template <int R>
__device__ __forceinline__ void wave_sum_store(float (&v)[R], float *dst, int lane) {
  wave_sum_store(v, dst, 0, lane);
}

// ---- a staged splat's tight alpha >= 1/255 box as 4 x f16 in tile-local pixels (raster kernels) ----
__device__ __forceinline__ uint2 d4gs_pack_box(float x0, float x1, float y0, float y1) {
  const float lo = -32.f, hi = 48.f;
  x0 = fminf(fmaxf(x0 - 0.04f, lo), hi), x1 = fminf(fmaxf(x1 + 0.04f, lo), hi);
  y0 = fminf(fmaxf(y0 - 0.04f, lo), hi), y1 = fminf(fmaxf(y1 + 0.04f, lo), hi);
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 a = {(_Float16)x0, (_Float16)x1}, b = {(_Float16)y0, (_Float16)y1};
  return make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
}
__device__ __forceinline__ float4 d4gs_unpack_box(uint2 p) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 a = __builtin_bit_cast(h2, p.x), b = __builtin_bit_cast(h2, p.y);
  return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}


// Dynamic-LDS padding that caps a kernel's residency at `q` workgroups per CU (160 KB of LDS per CU; allocation granule taken as
// 1280 B): the composites are issue-bound at 8 workgroups (= 8 waves per SIMD), and a launch whose workgroup count is not a multiple
// of 256 x 8 ends in a partial round that runs at a fraction of the chip's issue rate (cfg2: 4608 workgroups = 2.25 rounds of 2048;
// the last 512 take 0.43 of a full round's time).  Capping at q = 6 makes the same launch exactly 3 full rounds of 1536.
inline int d4gs_lds_pad_for_wgs_per_cu(const void *kernel, int q) {
  if (q <= 0 || q >= 8) return 0;
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, kernel) != hipSuccess) return 0;
  const int lds_cu = 160 * 1024, granule = 1280;
  const int need = ((lds_cu / (q + 1) + 1 + granule - 1) / granule) * granule;  // > LDS / (q + 1): q + 1 workgroups no longer fit
  if ((long)need * q > lds_cu) return 0;
  const int pad = need - (int)at.sharedSizeBytes;
  return pad > 0 ? pad : 0;
}

#define D4GS_LAUNCH(name, kernel, grid, block, lds, stream, ...)            \
  do {                                                                        \
    ProfScope _ps(name, stream);                                              \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);        \
  } while (0)
