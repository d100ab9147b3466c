// cpu_twin.hip -- d4gs_forward_cpu / d4gs_backward_cpu: the CPU twins of the one-call entry points ("CPU twins
// (*_cpu, host pointers, no stream) for config 1"; BASELINE.json configs[0] is the reference's CPU-runnable plumbing case).
//
// HOST code only (this file holds no kernel).  It is a separate entry point, NOT a fallback: nothing in the package routes a
// device render here, d4gs_forward / the Python seams still refuse CPU tensors.  Same contract as d4gs_forward / d4gs_backward
// (include/d4gs.h) with every pointer a HOST pointer: flow3d/params.py:39-43,142-180 (activations, motion bases),
// flow3d/transforms.py:41-53 (cont_6d_to_rmat), flow3d/scene_model.py:67-120,352-353 (pose compose, camera delta),
// gsplat 1.1.1 rasterization(packed=False) at flow3d/scene_model.py:360-373, the blend of :386-397.
//
// Written for clarity, scalar fp32, single thread:
//   * the per-instance chain (leaf parameters -> 2-D mean, conic, depth) is ONE function template over its scalar type: `float`
//     in the forward, `Var` - a node of a small reverse-mode tape - in the backward, so there is exactly one statement of the
//     math and its adjoint cannot drift from it;
//   * tile lists, depth sort and the per-pixel composite / its back-to-front replay follow gsplat's rules literally;
//   * the backward re-runs the forward (no workspace crosses the two calls).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/d4gs.h"

void d4gs_set_error(const char *fmt, ...);

namespace {

// ---- reverse-mode tape -----------------------------------------------------------------------------------------------
struct Tape {
  std::vector<int> pa, pb;
  std::vector<float> da, db;
  std::vector<double> adj;
  int push(int a, float d_a, int b, float d_b) {
    pa.push_back(a), pb.push_back(b), da.push_back(d_a), db.push_back(d_b);
    return (int)pa.size() - 1;
  }
  void clear() { pa.clear(), pb.clear(), da.clear(), db.clear(); }
  void sweep() {  // adj[] holds the output seeds; on return adj[i] = d L / d node i
    for (int i = (int)pa.size() - 1; i >= 0; i--) {
      const double v = adj[i];
      if (v == 0.0) continue;
      if (pa[i] >= 0) adj[pa[i]] += (double)da[i] * v;
      if (pb[i] >= 0) adj[pb[i]] += (double)db[i] * v;
    }
  }
};
thread_local Tape *g_tape = nullptr;

struct Var {
  float v;
  int id;  // -1: constant
  Var() : v(0.f), id(-1) {}
  Var(float x) : v(x), id(-1) {}
  Var(float x, int i) : v(x), id(i) {}
};
inline Var input(float x) { return Var(x, g_tape->push(-1, 0.f, -1, 0.f)); }
inline Var node(float v, const Var &a, float d_a, const Var &b, float d_b) {
  if (a.id < 0 && b.id < 0) return Var(v);
  return Var(v, g_tape->push(a.id, d_a, b.id, d_b));
}
inline Var operator+(const Var &a, const Var &b) { return node(a.v + b.v, a, 1.f, b, 1.f); }
inline Var operator-(const Var &a, const Var &b) { return node(a.v - b.v, a, 1.f, b, -1.f); }
inline Var operator*(const Var &a, const Var &b) { return node(a.v * b.v, a, b.v, b, a.v); }
inline Var operator/(const Var &a, const Var &b) { return node(a.v / b.v, a, 1.f / b.v, b, -a.v / (b.v * b.v)); }
inline Var operator-(const Var &a) { return node(-a.v, a, -1.f, Var(), 0.f); }
inline Var sqrt_(const Var &a) {
  const float r = sqrtf(a.v);
  return node(r, a, r > 0.f ? 0.5f / r : 0.f, Var(), 0.f);
}
inline Var exp_(const Var &a) {
  const float e = expf(a.v);
  return node(e, a, e, Var(), 0.f);
}
inline Var clamp_lo(const Var &a, float lo) { return a.v >= lo ? a : Var(lo); }                    // max(a, lo): torch's subgradient
inline Var clamp_(const Var &a, float lo, float hi) { return a.v < lo ? Var(lo) : (a.v > hi ? Var(hi) : a); }
inline float sqrt_(float a) { return sqrtf(a); }
inline float exp_(float a) { return expf(a); }
inline float clamp_lo(float a, float lo) { return a >= lo ? a : lo; }
inline float clamp_(float a, float lo, float hi) { return a < lo ? lo : (a > hi ? hi : a); }
inline float val(float a) { return a; }
inline float val(const Var &a) { return a.v; }

// ---- the per-instance chain ----------------------------------------------------------------------------------------
struct Camera {
  float fx, fy, cx, cy, limx, limy;
};
template <class T>
struct InstIn {
  T mu[3], q[4], sc[3];   // raw leaves (scales / quats raw or activated per the flags)
  const T *coef;          // [K] raw motion coefficients (softmax inside, as d4gs_project_fwd); dynamic Gaussians only
  const T *Bf, *Bc;       // [K][9] bases at floor(t) / ceil(t): (transl 3, rot6 6) per basis
  T w;                    // t - floor(t) (clamped frames, params.py:152-173)
  const T *RT;            // [12] camera delta or nullptr
  const T *V;             // [12] world -> camera (rows of R | t)
  bool dynamic;
  int K;
};
template <class T>
struct InstOut {
  T mx, my, ca, cb, cc, depth;  // 2-D mean, conic, camera-space depth
  float cov_a, cov_c;           // blurred 2-D covariance diagonal (tight rectangle)
  int radius;                   // 0 = culled
};

template <class T>
void instance_chain(const D4gsDims &d, const Camera &cam, const InstIn<T> &in, InstOut<T> &o) {
  const bool raw = d.flags & D4GS_RAW_PARAMS;
  o.radius = 0;
  o.cov_a = o.cov_c = 0.f;
  // normalize(quats) (F.normalize: eps 1e-12 on the norm) -> rotation matrix
  T qn = sqrt_(in.q[0] * in.q[0] + in.q[1] * in.q[1] + in.q[2] * in.q[2] + in.q[3] * in.q[3]);
  T iq = T(1.f) / clamp_lo(qn, 1e-12f);
  T w = in.q[0] * iq, x = in.q[1] * iq, y = in.q[2] * iq, z = in.q[3] * iq;
  T Rq[9] = {T(1.f) - T(2.f) * (y * y + z * z), T(2.f) * (x * y - w * z), T(2.f) * (x * z + w * y),
             T(2.f) * (x * y + w * z), T(1.f) - T(2.f) * (x * x + z * z), T(2.f) * (y * z - w * x),
             T(2.f) * (x * z - w * y), T(2.f) * (y * z + w * x), T(1.f) - T(2.f) * (x * x + y * y)};
  T sc[3];
  for (int j = 0; j < 3; j++) sc[j] = raw ? exp_(in.sc[j]) : in.sc[j];
  T mw[3] = {in.mu[0], in.mu[1], in.mu[2]}, Rm[9];
  for (int i = 0; i < 9; i++) Rm[i] = Rq[i];
  if (in.dynamic) {
    // softmax(motion_coefs) (params.py:43), bases lerped in time, blended by the coefficients (params.py:142-180)
    std::vector<T> c(in.K);
    {
      float m = -INFINITY;
      for (int k = 0; k < in.K; k++) m = fmaxf(m, val(in.coef[k]));
      T sum = T(0.f);
      for (int k = 0; k < in.K; k++) c[k] = exp_(in.coef[k] - T(m)), sum = sum + c[k];
      for (int k = 0; k < in.K; k++) c[k] = c[k] / sum;
    }
    T v9[9];
    for (int j = 0; j < 9; j++) v9[j] = T(0.f);
    for (int k = 0; k < in.K; k++)
      for (int j = 0; j < 9; j++) v9[j] = v9[j] + c[k] * ((T(1.f) - in.w) * in.Bf[k * 9 + j] + in.w * in.Bc[k * 9 + j]);
    // cont_6d_to_rmat (transforms.py:41-53): Gram-Schmidt, columns x, y, z
    const T *r6 = v9 + 3;
    T ina = T(1.f) / clamp_lo(sqrt_(r6[0] * r6[0] + r6[1] * r6[1] + r6[2] * r6[2]), 1e-12f);
    T gx[3] = {r6[0] * ina, r6[1] * ina, r6[2] * ina};
    T dd = r6[3] * gx[0] + r6[4] * gx[1] + r6[5] * gx[2];
    T bp[3] = {r6[3] - dd * gx[0], r6[4] - dd * gx[1], r6[5] - dd * gx[2]};
    T inb = T(1.f) / clamp_lo(sqrt_(bp[0] * bp[0] + bp[1] * bp[1] + bp[2] * bp[2]), 1e-12f);
    T gy[3] = {bp[0] * inb, bp[1] * inb, bp[2] * inb};
    T gz[3] = {gx[1] * gy[2] - gx[2] * gy[1], gx[2] * gy[0] - gx[0] * gy[2], gx[0] * gy[1] - gx[1] * gy[0]};
    T Rd[9] = {gx[0], gy[0], gz[0], gx[1], gy[1], gz[1], gx[2], gy[2], gz[2]};
    for (int i = 0; i < 3; i++) mw[i] = Rd[i * 3] * in.mu[0] + Rd[i * 3 + 1] * in.mu[1] + Rd[i * 3 + 2] * in.mu[2] + v9[i];
    // pose compose (scene_model.py:94-102) as a product of rotation matrices: identical on SO(3)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rm[i * 3 + j] = Rd[i * 3] * Rq[j] + Rd[i * 3 + 1] * Rq[3 + j] + Rd[i * 3 + 2] * Rq[6 + j];
  }
  if (in.RT) {  // camera delta: means only (scene_model.py:352-353)
    T t0 = in.RT[0] * mw[0] + in.RT[1] * mw[1] + in.RT[2] * mw[2] + in.RT[3];
    T t1 = in.RT[4] * mw[0] + in.RT[5] * mw[1] + in.RT[6] * mw[2] + in.RT[7];
    T t2 = in.RT[8] * mw[0] + in.RT[9] * mw[1] + in.RT[10] * mw[2] + in.RT[11];
    mw[0] = t0, mw[1] = t1, mw[2] = t2;
  }
  // gsplat fully_fused_projection
  const T *V = in.V;
  T pc[3];
  for (int i = 0; i < 3; i++) pc[i] = V[i * 4] * mw[0] + V[i * 4 + 1] * mw[1] + V[i * 4 + 2] * mw[2] + V[i * 4 + 3];
  const float zf = val(pc[2]);
  if (!(zf >= d.near_plane && zf <= d.far_plane)) return;
  T M[9];  // Rcw Rm diag(sc)
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[i * 3 + j] = (V[i * 4] * Rm[j] + V[i * 4 + 1] * Rm[3 + j] + V[i * 4 + 2] * Rm[6 + j]) * sc[j];
  T cxx = M[0] * M[0] + M[1] * M[1] + M[2] * M[2], cxy = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  T cxz = M[0] * M[6] + M[1] * M[7] + M[2] * M[8], cyy = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  T cyz = M[3] * M[6] + M[4] * M[7] + M[5] * M[8], czz = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
  T rz = T(1.f) / pc[2], rz2 = rz * rz;
  T tx = pc[2] * clamp_(pc[0] * rz, -cam.limx, cam.limx), ty = pc[2] * clamp_(pc[1] * rz, -cam.limy, cam.limy);
  T J00 = T(cam.fx) * rz, J11 = T(cam.fy) * rz, J02 = -(T(cam.fx) * tx * rz2), J12 = -(T(cam.fy) * ty * rz2);
  T c00 = J00 * J00 * cxx + T(2.f) * J00 * J02 * cxz + J02 * J02 * czz;
  T c01 = J00 * J11 * cxy + J00 * J12 * cxz + J02 * J11 * cyz + J02 * J12 * czz;
  T c11 = J11 * J11 * cyy + T(2.f) * J11 * J12 * cyz + J12 * J12 * czz;
  T a = c00 + T(d.eps2d), b = c01, c = c11 + T(d.eps2d);
  T det = a * c - b * b;
  if (!(val(det) > 0.f)) return;
  const float mid = 0.5f * (val(a) + val(c));
  const float lam = mid + sqrtf(fmaxf(0.01f, mid * mid - val(det)));
  const float radius = ceilf(3.f * sqrtf(lam));
  if (radius <= d.radius_clip) return;
  o.mx = T(cam.fx) * pc[0] * rz + T(cam.cx);
  o.my = T(cam.fy) * pc[1] * rz + T(cam.cy);
  const float mx = val(o.mx), my = val(o.my);
  if (mx + radius <= 0.f || mx - radius >= (float)d.width || my + radius <= 0.f || my - radius >= (float)d.height) return;
  T idet = T(1.f) / det;
  o.ca = c * idet, o.cb = -(b * idet), o.cc = a * idet;
  o.depth = pc[2];
  o.cov_a = val(a), o.cov_c = val(c);
  o.radius = (int)radius;
}

// ---- everything one render holds ------------------------------------------------------------------------------------
struct Frame {
  D4gsDims d;
  Camera cam;
  int tw, th, NCH;
  std::vector<float> opac, ctab;                    // activated [N], [N,D]
  std::vector<float> mx, my, ca, cb, cc, depth;     // [S*N]
  std::vector<int> radius;
  std::vector<int> tile_start;                      // [S*tiles + 1]
  std::vector<int> list;                            // instance index (s*N + g) per sorted intersection
  std::vector<int> last;                            // [S*H*W] list position of the last contributor, -1 none
  std::vector<float> finalT;
};

inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

void tile_rect(const Frame &f, float mx, float my, int radius, float opac, float cov_a, float cov_c, int &x0, int &y0, int &x1,
               int &y1) {
  const float inv = 1.0f / D4GS_TILE;
  const float tx = mx * inv, ty = my * inv, tr = (float)radius * inv;
  x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)f.tw), y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)f.th);
  x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)f.tw), y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)f.th);
  if (!(f.d.flags & D4GS_EXACT_CULL)) return;
  // tiles holding a pixel centre inside sigma <= ln(255 opacity) (+ margins): pixels elsewhere fail alpha >= 1/255 anyway
  const float tau = logf(255.f * opac) * 1.01f + 0.02f;
  if (!(tau > 0.f)) {
    x1 = x0, y1 = y0;
    return;
  }
  const float ex = sqrtf(2.f * tau * cov_a) + 1e-3f, ey = sqrtf(2.f * tau * cov_c) + 1e-3f;
  const float jx0 = ceilf(mx - ex - 0.5f), jx1 = floorf(mx + ex - 0.5f), jy0 = ceilf(my - ey - 0.5f), jy1 = floorf(my + ey - 0.5f);
  const int a0 = (int)fmaxf(floorf(jx0 * inv), (float)x0), a1 = (int)fminf(floorf(jx1 * inv) + 1.f, (float)x1);
  const int b0 = (int)fmaxf(floorf(jy0 * inv), (float)y0), b1 = (int)fminf(floorf(jy1 * inv) + 1.f, (float)y1);
  x0 = a0, x1 = std::max(a1, a0), y0 = b0, y1 = std::max(b1, b0);
  if (x1 == x0 || y1 == y0) x1 = x0, y1 = y0;
}

// time slot of a sub-sample: clamped floor / ceil frames and the lerp weight (params.py:152-173)
void time_slot(const D4gsDims &d, float t, int &fl, int &ce, float &w) {
  const float ff = fminf(fmaxf(floorf(t), 0.f), (float)(d.T - 1)), cf = fminf(fmaxf(ceilf(t), 0.f), (float)(d.T - 1));
  fl = (int)ff, ce = (int)cf, w = t - ff;
}
void gather_bases(const D4gsDims &d, const D4gsProjIn &in, int frame, float *B /* [K][9] */) {
  for (int k = 0; k < d.K; k++) {
    for (int j = 0; j < 3; j++) B[k * 9 + j] = in.transls[(k * d.T + frame) * 3 + j];
    for (int j = 0; j < 6; j++) B[k * 9 + 3 + j] = in.rots[(k * d.T + frame) * 6 + j];
  }
}

int check_args(const D4gsDims *d, const D4gsProjIn *in, const D4gsFrameIO *io, const char *who) {
  if (!d || !in || !io || !in->means || !in->quats || !in->scales || !in->opacities || !in->viewmat || !in->Kmat || !io->renders ||
      !io->alphas || !io->means2d || !io->radii || (d && d->D > 0 && !in->colors)) {
    d4gs_set_error("%s: NULL required argument", who);
    return D4GS_EINVAL;
  }
  if (d->N < 0 || d->S <= 0 || d->width <= 0 || d->height <= 0 || d->D < 0 || d->G < 0 || d->G > d->N ||
      (d->G > 0 && (d->K <= 0 || d->T <= 0 || !in->motion_coefs || !in->rots || !in->transls || !in->times))) {
    d4gs_set_error("%s: bad dims N=%d G=%d K=%d T=%d S=%d W=%d H=%d D=%d", who, d->N, d->G, d->K, d->T, d->S, d->width, d->height, d->D);
    return D4GS_EINVAL;
  }
  if (io->blended && !io->acc) {
    d4gs_set_error("%s: io->blended needs io->acc", who);
    return D4GS_EINVAL;
  }
  return D4GS_OK;
}

void forward_impl(const D4gsDims &d, const D4gsProjIn &in, const D4gsFrameIO &io, Frame &f) {
  const int N = d.N, S = d.S, W = d.width, H = d.height, D = d.D;
  f.d = d;
  f.tw = (W + D4GS_TILE - 1) / D4GS_TILE, f.th = (H + D4GS_TILE - 1) / D4GS_TILE;
  f.NCH = D + (d.depth_mode != D4GS_DEPTH_NONE ? 1 : 0);
  f.cam.fx = in.Kmat[0], f.cam.fy = in.Kmat[4], f.cam.cx = in.Kmat[2], f.cam.cy = in.Kmat[5];
  f.cam.limx = 1.3f * (0.5f * (float)W / f.cam.fx), f.cam.limy = 1.3f * (0.5f * (float)H / f.cam.fy);
  const bool raw = d.flags & D4GS_RAW_PARAMS;
  f.opac.resize(N), f.ctab.assign((size_t)N * std::max(D, 1), 0.f);
  for (int g = 0; g < N; g++) {
    f.opac[g] = raw ? sigmoidf(in.opacities[g]) : in.opacities[g];
    for (int c = 0; c < D; c++) {
      float v = in.colors[(size_t)g * D + c];
      if ((d.flags & D4GS_RAW_COLORS) && c < d.n_sigmoid) v = sigmoidf(v);
      f.ctab[(size_t)g * D + c] = v;
    }
  }
  const size_t SN = (size_t)S * N;
  f.mx.assign(SN, 0.f), f.my.assign(SN, 0.f), f.ca.assign(SN, 0.f), f.cb.assign(SN, 0.f), f.cc.assign(SN, 0.f), f.depth.assign(SN, 0.f);
  f.radius.assign(SN, 0);
  const int n_tiles = f.tw * f.th;
  std::vector<int> count((size_t)S * n_tiles + 1, 0);
  std::vector<int> rect(SN * 4, 0);
  std::vector<float> Bf(std::max(d.K, 1) * 9), Bc(std::max(d.K, 1) * 9);
  float V12[12];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) V12[i * 4 + j] = in.viewmat[i * 4 + j];
  for (int s = 0; s < S; s++) {
    float w = 0.f;
    if (d.G > 0) {
      int fl, ce;
      time_slot(d, in.times[s], fl, ce, w);
      gather_bases(d, in, fl, Bf.data()), gather_bases(d, in, ce, Bc.data());
    }
    for (int g = 0; g < N; g++) {
      InstIn<float> ii;
      for (int j = 0; j < 3; j++) ii.mu[j] = in.means[g * 3 + j], ii.sc[j] = in.scales[g * 3 + j];
      for (int j = 0; j < 4; j++) ii.q[j] = in.quats[(size_t)g * 4 + j];
      ii.dynamic = g < d.G, ii.K = d.K;
      ii.coef = ii.dynamic ? in.motion_coefs + (size_t)g * d.K : nullptr;
      ii.Bf = Bf.data(), ii.Bc = Bc.data(), ii.w = w;
      ii.RT = in.RTs ? in.RTs + s * 12 : nullptr;
      ii.V = V12;
      InstOut<float> o;
      instance_chain<float>(d, f.cam, ii, o);
      const size_t i = (size_t)s * N + g;
      f.radius[i] = o.radius;
      if (o.radius > 0) {
        f.mx[i] = o.mx, f.my[i] = o.my, f.ca[i] = o.ca, f.cb[i] = o.cb, f.cc[i] = o.cc, f.depth[i] = o.depth;
        int x0, y0, x1, y1;
        tile_rect(f, o.mx, o.my, o.radius, f.opac[g], o.cov_a, o.cov_c, x0, y0, x1, y1);
        rect[i * 4] = x0, rect[i * 4 + 1] = y0, rect[i * 4 + 2] = x1, rect[i * 4 + 3] = y1;
        for (int ty = y0; ty < y1; ty++)
          for (int tx = x0; tx < x1; tx++) count[(size_t)s * n_tiles + ty * f.tw + tx]++;
      }
      io.radii[i] = o.radius;
      io.means2d[i * 2] = f.mx[i], io.means2d[i * 2 + 1] = f.my[i];
    }
  }
  // tile lists: counting sort by tile, then by (depth, instance) inside every tile (gsplat: tile | depth-bits keys, stable)
  f.tile_start.assign((size_t)S * n_tiles + 1, 0);
  for (size_t t = 0; t < (size_t)S * n_tiles; t++) f.tile_start[t + 1] = f.tile_start[t] + count[t];
  const int n_isect = f.tile_start[(size_t)S * n_tiles];
  f.list.resize(n_isect);
  std::vector<int> cur(f.tile_start.begin(), f.tile_start.end() - 1);
  for (size_t i = 0; i < SN; i++) {
    if (f.radius[i] <= 0) continue;
    const int s = (int)(i / N);
    for (int ty = rect[i * 4 + 1]; ty < rect[i * 4 + 3]; ty++)
      for (int tx = rect[i * 4]; tx < rect[i * 4 + 2]; tx++) f.list[cur[(size_t)s * n_tiles + ty * f.tw + tx]++] = (int)i;
  }
  int longest = 0;
  for (size_t t = 0; t < (size_t)S * n_tiles; t++) {
    std::stable_sort(f.list.begin() + f.tile_start[t], f.list.begin() + f.tile_start[t + 1],
                     [&](int a, int b) { return f.depth[a] < f.depth[b]; });  // (entries arrive in instance order: ties keep it)
    longest = std::max(longest, f.tile_start[t + 1] - f.tile_start[t]);
  }
  // composite, front to back
  const int NCH = f.NCH;
  f.last.assign((size_t)S * H * W, -1), f.finalT.assign((size_t)S * H * W, 1.f);
  long long live = 0, sampled = 0;  // the same tile sample as the device path: every ceil(tiles / 128)-th tile
  const int lstride = (S * n_tiles + 127) >> 7;
  std::vector<float> acc(std::max(NCH, 1));
  for (int s = 0; s < S; s++)
    for (int tyi = 0; tyi < f.th; tyi++)
      for (int txi = 0; txi < f.tw; txi++) {
        const int t = s * n_tiles + tyi * f.tw + txi, b0 = f.tile_start[t], b1 = f.tile_start[t + 1];
        int tile_hi = b0 - 1;
        for (int y = tyi * D4GS_TILE; y < std::min((tyi + 1) * D4GS_TILE, H); y++)
          for (int x = txi * D4GS_TILE; x < std::min((txi + 1) * D4GS_TILE, W); x++) {
            const float px = (float)x + 0.5f, py = (float)y + 0.5f;
            float T = 1.f;
            int last = -1;
            std::fill(acc.begin(), acc.end(), 0.f);
            for (int k = b0; k < b1; k++) {
              const int i = f.list[k], g = i % N;
              const float dx = f.mx[i] - px, dy = f.my[i] - py;
              const float sigma = 0.5f * (f.ca[i] * dx * dx + f.cc[i] * dy * dy) + f.cb[i] * dx * dy;
              const float alpha = fminf(0.999f, f.opac[g] * expf(-sigma));
              if (sigma < 0.f || alpha < 1.f / 255.f) continue;
              const float nT = T * (1.f - alpha);
              if (nT <= 1e-4f) break;
              const float vis = alpha * T;
              for (int c = 0; c < D; c++) acc[c] += f.ctab[(size_t)g * D + c] * vis;
              if (NCH > D) acc[D] += f.depth[i] * vis;
              T = nT, last = k;
            }
            const size_t pix = ((size_t)s * H + y) * W + x;
            f.last[pix] = last, f.finalT[pix] = T;
            tile_hi = std::max(tile_hi, last);
            const float al = 1.f - T;
            io.alphas[pix] = al;
            float *o = io.renders + pix * NCH;
            for (int c = 0; c < D; c++) o[c] = acc[c] + (io.background ? T * io.background[c] : 0.f);
            if (NCH > D) o[D] = d.depth_mode == D4GS_DEPTH_ED ? acc[D] / fmaxf(al, 1e-10f) : acc[D];
          }
        if (t % lstride == 0) sampled += b1 - b0, live += tile_hi - b0 + 1;
      }
  if (io.n_isect) io.n_isect[0] = n_isect, io.n_isect[1] = longest, io.n_isect[2] = sampled, io.n_isect[3] = live;  // D4gsProjOut.n_isect
  // blend (scene_model.py:386-397 incl. the in-place max / min quirk: candidates raw_0 .. raw_{S-2} and the mean)
  if (io.blended) {
    const size_t P = (size_t)H * W, PC = P * NCH;
    for (size_t i = 0; i < PC; i++) {
      float sum = 0.f;
      for (int s = 0; s < S; s++) sum += io.renders[s * PC + i];
      float v = S == 1 ? io.renders[i] : sum / (float)S;
      const int pol = io.policy ? io.policy[i % NCH] : 0;
      for (int s = 0; s + 1 < S && pol; s++) v = pol == 1 ? fmaxf(v, io.renders[s * PC + i]) : fminf(v, io.renders[s * PC + i]);
      io.blended[i] = v;
    }
    for (size_t i = 0; i < P; i++) {
      float sum = 0.f;
      for (int s = 0; s < S; s++) sum += io.alphas[s * P + i];
      io.acc[i] = S == 1 ? io.alphas[i] : sum / (float)S;
    }
  }
}

}  // namespace

extern "C" int d4gs_forward_cpu(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io) {
  int rc = check_args(dims, in, io, "d4gs_forward_cpu");
  if (rc) return rc;
  Frame f;
  forward_impl(*dims, *in, *io, f);
  return D4GS_OK;
}

extern "C" int d4gs_backward_cpu(const D4gsDims *dims, const D4gsProjIn *in, const D4gsFrameIO *io, const D4gsFrameGrads *g,
                                 const D4gsLeafGrads *leaf) {
  int rc = check_args(dims, in, io, "d4gs_backward_cpu");
  if (rc) return rc;
  if (!g || !leaf || !leaf->v_means || !leaf->v_quats || !leaf->v_scales || !leaf->v_opacities || (dims->D > 0 && !leaf->v_colors) ||
      (io->blended ? !(g->v_blended || g->v_acc || g->v_renders || g->v_alphas) : !g->v_renders)) {
    d4gs_set_error("d4gs_backward_cpu: NULL gradient argument");
    return D4GS_EINVAL;
  }
  const D4gsDims &d = *dims;
  Frame f;
  forward_impl(d, *in, *io, f);  // (rewrites io's outputs with the same values)
  const int N = d.N, S = d.S, W = d.width, H = d.height, D = d.D, NCH = f.NCH, n_tiles = f.tw * f.th;
  const size_t P = (size_t)H * W, PC = P * NCH, SN = (size_t)S * N;
  // ---- blend backward: one gradient per sub-sample image ----
  std::vector<float> vR((size_t)S * PC, 0.f), vA((size_t)S * P, 0.f);
  if (io->blended) {
    for (size_t i = 0; i < PC; i++) {
      const int pol = io->policy ? io->policy[i % NCH] : 0;
      const float gg = g->v_blended ? g->v_blended[i] : 0.f;
      int winner = -1;
      if (pol != 0 && S > 1)
        for (int s = 0; s + 1 < S; s++)
          if (io->renders[s * PC + i] == io->blended[i]) {
            winner = s;
            break;
          }
      for (int s = 0; s < S; s++) vR[s * PC + i] = winner < 0 ? gg / (float)S : (s == winner ? gg : 0.f);
    }
    if (g->v_acc)
      for (size_t i = 0; i < P; i++)
        for (int s = 0; s < S; s++) vA[s * P + i] = g->v_acc[i] / (float)S;
  }
  if (g->v_renders)
    for (size_t i = 0; i < (size_t)S * PC; i++) vR[i] += g->v_renders[i];
  if (g->v_alphas)
    for (size_t i = 0; i < (size_t)S * P; i++) vA[i] += g->v_alphas[i];
  // ---- composite backward: back-to-front replay per pixel (gsplat rasterize_to_pixels_bwd) ----
  std::vector<double> v_mx(SN, 0.0), v_my(SN, 0.0), v_ca(SN, 0.0), v_cb(SN, 0.0), v_cc(SN, 0.0), v_dep(SN, 0.0);
  std::vector<double> v_op(N, 0.0), v_col((size_t)N * std::max(D, 1), 0.0);
  std::vector<float> vo(std::max(NCH, 1)), buf(std::max(NCH, 1));
  for (int s = 0; s < S; s++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const size_t pix = ((size_t)s * H + y) * W + x;
        const int t = s * n_tiles + (y / D4GS_TILE) * f.tw + x / D4GS_TILE, b0 = f.tile_start[t];
        const float Tfin = f.finalT[pix], al = io->alphas[pix];
        for (int c = 0; c < NCH; c++) vo[c] = vR[pix * NCH + c];
        float v_al = vA[pix];
        if (NCH > D && d.depth_mode == D4GS_DEPTH_ED) {  // o = acc / max(alpha, 1e-10)
          const float den = fmaxf(al, 1e-10f), vd = vo[D];
          if (al >= 1e-10f) v_al -= vd * io->renders[pix * NCH + D] / den;
          vo[D] = vd / den;
        }
        float bgdot = 0.f;
        if (io->background)
          for (int c = 0; c < D; c++) bgdot += io->background[c] * vo[c];
        // alpha_pixel = 1 - T_final and the background term T_final * bg: d T_final / d alpha_k = -T_final / (1 - alpha_k)
        const float vT = v_al - bgdot;
        float T = Tfin;
        std::fill(buf.begin(), buf.end(), 0.f);
        const float px = (float)x + 0.5f, py = (float)y + 0.5f;
        for (int k = f.last[pix]; k >= b0; k--) {
          const int i = f.list[k], gg = i % N;
          const float dx = f.mx[i] - px, dy = f.my[i] - py;
          const float sigma = 0.5f * (f.ca[i] * dx * dx + f.cc[i] * dy * dy) + f.cb[i] * dx * dy;
          const float ov = f.opac[gg] * expf(-sigma), alpha = fminf(0.999f, ov);
          if (sigma < 0.f || alpha < 1.f / 255.f) continue;
          const float ra = 1.f / (1.f - alpha);
          T *= ra;  // transmittance in front of this splat
          const float fac = alpha * T;
          float v_alpha = 0.f;
          for (int c = 0; c < NCH; c++) {
            const float col = c < D ? f.ctab[(size_t)gg * D + c] : f.depth[i];
            v_alpha += (col * T - buf[c] * ra) * vo[c];
            if (c < D) v_col[(size_t)gg * D + c] += (double)(fac * vo[c]);
            else v_dep[i] += (double)(fac * vo[c]);
            buf[c] += col * fac;
          }
          v_alpha += Tfin * ra * vT;
          if (ov <= 0.999f) {  // alpha = min(0.999, o e^-sigma): no gradient through the clamp
            const float vs = -ov * v_alpha;  // d / d sigma
            v_mx[i] += (double)(vs * (f.ca[i] * dx + f.cb[i] * dy));
            v_my[i] += (double)(vs * (f.cc[i] * dy + f.cb[i] * dx));
            v_ca[i] += (double)(0.5f * vs * dx * dx), v_cb[i] += (double)(vs * dx * dy), v_cc[i] += (double)(0.5f * vs * dy * dy);
            v_op[gg] += (double)(expf(-sigma) * v_alpha);
          }
        }
      }
  // the means2d.grad contract (trainer.py:975) + fused densification statistics (trainer.py:967-989)
  if (g->v_means2d)
    for (size_t i = 0; i < SN; i++) g->v_means2d[i * 2] = (float)v_mx[i], g->v_means2d[i * 2 + 1] = (float)v_my[i];
  if (g->stats_grad_norm_acc) {
    const float sx = (float)W / 2.f * (float)g->stats_batch_size * (float)S, sy = (float)H / 2.f * (float)g->stats_batch_size * (float)S;
    const float max_wh = (float)std::max(W, H);
    for (int s = 0; s < S; s++)
      for (int gi = 0; gi < N; gi++) {
        const size_t i = (size_t)s * N + gi;
        if (f.radius[i] <= 0) continue;
        const float gx = (float)v_mx[i] * sx, gy = (float)v_my[i] * sy;
        g->stats_grad_norm_acc[gi] += sqrtf(gx * gx + gy * gy);
        g->stats_vis_count[gi] += 1;
        if (g->stats_update_max_radii) g->stats_max_radii[gi] = fmaxf(g->stats_max_radii[gi], (float)f.radius[i] / max_wh);
      }
  }
  // ---- leaf gradients: activations by hand, the per-instance chain through the tape ----
  const bool raw = d.flags & D4GS_RAW_PARAMS;
  const int K = d.G > 0 ? d.K : 0;
  std::vector<double> a_means((size_t)N * 3, 0.0), a_quats((size_t)N * 4, 0.0), a_scales((size_t)N * 3, 0.0);
  std::vector<double> a_coef((size_t)d.G * std::max(K, 1), 0.0), a_rots, a_transls, a_times(S, 0.0), a_RT((size_t)S * 12, 0.0), a_V(12, 0.0);
  if (K) a_rots.assign((size_t)K * d.T * 6, 0.0), a_transls.assign((size_t)K * d.T * 3, 0.0);
  Tape tape;
  g_tape = &tape;
  std::vector<float> Bf(std::max(K, 1) * 9), Bc(std::max(K, 1) * 9);
  for (int s = 0; s < S; s++) {
    float w = 0.f;
    int fl = 0, ce = 0;
    if (K) {
      time_slot(d, in->times[s], fl, ce, w);
      gather_bases(d, *in, fl, Bf.data()), gather_bases(d, *in, ce, Bc.data());
    }
    for (int gi = 0; gi < N; gi++) {
      const size_t i = (size_t)s * N + gi;
      if (f.radius[i] <= 0) continue;
      if (v_mx[i] == 0.0 && v_my[i] == 0.0 && v_ca[i] == 0.0 && v_cb[i] == 0.0 && v_cc[i] == 0.0 && v_dep[i] == 0.0) continue;
      tape.clear();
      InstIn<Var> ii;
      for (int j = 0; j < 3; j++) ii.mu[j] = input(in->means[gi * 3 + j]);
      for (int j = 0; j < 4; j++) ii.q[j] = input(in->quats[(size_t)gi * 4 + j]);
      for (int j = 0; j < 3; j++) ii.sc[j] = input(in->scales[gi * 3 + j]);
      ii.dynamic = gi < d.G, ii.K = K;
      std::vector<Var> coef, vBf, vBc, vRT, vV(12);
      Var vw;
      if (ii.dynamic) {
        coef.resize(K), vBf.resize(K * 9), vBc.resize(K * 9);
        for (int k = 0; k < K; k++) coef[k] = input(in->motion_coefs[(size_t)gi * K + k]);
        for (int j = 0; j < K * 9; j++) vBf[j] = input(Bf[j]), vBc[j] = input(Bc[j]);
        vw = input(w);
        ii.coef = coef.data(), ii.Bf = vBf.data(), ii.Bc = vBc.data(), ii.w = vw;
      }
      if (in->RTs) {
        vRT.resize(12);
        for (int j = 0; j < 12; j++) vRT[j] = input(in->RTs[s * 12 + j]);
        ii.RT = vRT.data();
      } else {
        ii.RT = nullptr;
      }
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) vV[r * 4 + c] = input(in->viewmat[r * 4 + c]);
      ii.V = vV.data();
      InstOut<Var> o;
      instance_chain<Var>(d, f.cam, ii, o);
      if (o.radius <= 0) continue;  // (cannot happen: same arithmetic as the forward)
      tape.adj.assign(tape.pa.size(), 0.0);
      auto seed = [&](const Var &v, double a) {
        if (v.id >= 0) tape.adj[v.id] += a;
      };
      seed(o.mx, v_mx[i]), seed(o.my, v_my[i]), seed(o.ca, v_ca[i]), seed(o.cb, v_cb[i]), seed(o.cc, v_cc[i]), seed(o.depth, v_dep[i]);
      tape.sweep();
      auto ad = [&](const Var &v) { return v.id >= 0 ? tape.adj[v.id] : 0.0; };
      for (int j = 0; j < 3; j++) a_means[(size_t)gi * 3 + j] += ad(ii.mu[j]), a_scales[(size_t)gi * 3 + j] += ad(ii.sc[j]);
      for (int j = 0; j < 4; j++) a_quats[(size_t)gi * 4 + j] += ad(ii.q[j]);
      if (ii.dynamic) {
        for (int k = 0; k < K; k++) {
          a_coef[(size_t)gi * K + k] += ad(coef[k]);
          for (int j = 0; j < 3; j++)
            a_transls[((size_t)k * d.T + fl) * 3 + j] += ad(vBf[k * 9 + j]), a_transls[((size_t)k * d.T + ce) * 3 + j] += ad(vBc[k * 9 + j]);
          for (int j = 0; j < 6; j++)
            a_rots[((size_t)k * d.T + fl) * 6 + j] += ad(vBf[k * 9 + 3 + j]), a_rots[((size_t)k * d.T + ce) * 6 + j] += ad(vBc[k * 9 + 3 + j]);
        }
        a_times[s] += ad(vw);  // d w / d t = 1 (floor / ceil are piecewise constant)
      }
      if (in->RTs)
        for (int j = 0; j < 12; j++) a_RT[(size_t)s * 12 + j] += ad(vRT[j]);
      for (int j = 0; j < 12; j++) a_V[j] += ad(vV[j]);
    }
  }
  g_tape = nullptr;
  for (size_t j = 0; j < (size_t)N * 3; j++) leaf->v_means[j] = (float)a_means[j], leaf->v_scales[j] = (float)a_scales[j];
  for (size_t j = 0; j < (size_t)N * 4; j++) leaf->v_quats[j] = (float)a_quats[j];
  for (int gi = 0; gi < N; gi++) {
    const float o = f.opac[gi];
    leaf->v_opacities[gi] = (float)(raw ? v_op[gi] * (double)(o * (1.f - o)) : v_op[gi]);
    for (int c = 0; c < D; c++) {
      double v = v_col[(size_t)gi * D + c];
      if ((d.flags & D4GS_RAW_COLORS) && c < d.n_sigmoid) {
        const float cv = f.ctab[(size_t)gi * D + c];
        v *= (double)(cv * (1.f - cv));
      }
      leaf->v_colors[(size_t)gi * D + c] = (float)v;
    }
  }
  if (leaf->v_motion_coefs)
    for (size_t j = 0; j < (size_t)d.G * K; j++) leaf->v_motion_coefs[j] = (float)a_coef[j];
  if (leaf->v_rots)
    for (size_t j = 0; j < a_rots.size(); j++) leaf->v_rots[j] = (float)a_rots[j];
  if (leaf->v_transls)
    for (size_t j = 0; j < a_transls.size(); j++) leaf->v_transls[j] = (float)a_transls[j];
  if (leaf->v_times)
    for (int s = 0; s < S; s++) leaf->v_times[s] = (float)a_times[s];
  if (leaf->v_RTs)
    for (size_t j = 0; j < (size_t)S * 12; j++) leaf->v_RTs[j] = (float)a_RT[j];
  if (leaf->v_viewmat) {
    for (int j = 0; j < 16; j++) leaf->v_viewmat[j] = j < 12 ? (float)a_V[j] : 0.f;
  }
  return D4GS_OK;
}
