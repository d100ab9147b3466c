// camera.hip -- a12: the camera-delta / exposure-time generator that feeds the rasterizer `RTs [S,3,4]`, `times [S]`.
//
// Reference: MoveModel.forward_start_end_mid (flow3d/models/move_model.py:138-166) ->
//   pypose se3.Exp of the two MLP head outputs, linear_interpolation (spline_utils.py:371-408: lerp t, slerp q),
//   SE3.Log, se3_to_SE3 (spline_utils.py:197-215, reading the [tau,phi] log as [w,u] -- move_model.py:146-147),
//   and the exposure-time lerp with the relu/clamp'ed per-frame half-width (move_model.py:118-135,151-158);
//   MoveModel.preprocessPose + positional embedding (move_model.py:12-63,104-110; spline_utils.py:177-195).
//
// In eager PyTorch that chain is ~700 tiny kernels forward and ~1500 backward per render (three renders per
// training step); it is latency, not work: 12 differentiable inputs, 12*S outputs.  Here it is ONE launch: thread
// (s, j) evaluates sub-sample s in dual-number arithmetic carrying the tangent d/d delta[j], so the forward writes
// the poses AND the full Jacobian [S,12,12]; the backward is a 12S x 12 mat-vec in a second tiny kernel.
#include "common.h"

namespace {

// ---- first-order dual numbers -----------------------------------------------------------------------------
struct Dual {
  float v, d;
};
__device__ inline Dual mk(float v, float d = 0.f) { return Dual{v, d}; }
__device__ inline Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ inline Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ inline Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ inline Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ inline Dual operator/(Dual a, Dual b) {
  float q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__device__ inline Dual operator+(Dual a, float b) { return {a.v + b, a.d}; }
__device__ inline Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ inline Dual operator-(Dual a, float b) { return {a.v - b, a.d}; }
__device__ inline Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ inline Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ inline Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ inline Dual operator/(Dual a, float b) { return {a.v / b, a.d / b}; }
__device__ inline Dual operator/(float a, Dual b) { return mk(a) / b; }
__device__ inline Dual Sqrt(Dual a) {
  float s = sqrtf(a.v);
  return {s, a.d / (2.f * s)};
}
__device__ inline Dual Sin(Dual a) { return {sinf(a.v), cosf(a.v) * a.d}; }
__device__ inline Dual Cos(Dual a) { return {cosf(a.v), -sinf(a.v) * a.d}; }
__device__ inline Dual Atan(Dual a) { return {atanf(a.v), a.d / (1.f + a.v * a.v)}; }
__device__ inline Dual Acos(Dual a) { return {acosf(a.v), -a.d / sqrtf(1.f - a.v * a.v)}; }
__device__ inline float Acos(float a) { return acosf(a); }
// torch.clamp: the gradient passes on the closed interval
__device__ inline Dual Clamp(Dual a, float lo, float hi) { return {fminf(fmaxf(a.v, lo), hi), (a.v >= lo && a.v <= hi) ? a.d : 0.f}; }
__device__ inline float Clamp(float a, float lo, float hi) { return fminf(fmaxf(a, lo), hi); }
// `theta % math.pi` for theta = acos(.) >= 0: slope 1
__device__ inline Dual Fmod(Dual a, float m) { return {fmodf(a.v, m), a.d}; }
__device__ inline float Fmod(float a, float m) { return fmodf(a, m); }

// plain floats through the same templates (pose encoding carries no tangent)
__device__ inline float mkT(float v, float *) { return v; }
__device__ inline Dual mkT(float v, Dual *) { return mk(v); }
__device__ inline float Sqrt(float a) { return sqrtf(a); }
__device__ inline float val(float a) { return a; }
__device__ inline float val(Dual a) { return a.v; }
__device__ inline float zero_tangent(float a) { return a; }
__device__ inline Dual zero_tangent(Dual a) { return mk(a.v); }

template <typename T>
struct V3 {
  T x, y, z;
};
template <typename T>
struct M3 {
  T m[3][3];
};

template <typename T>
__device__ inline T dot(V3<T> a, V3<T> b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <typename T>
__device__ inline V3<T> cross(V3<T> a, V3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T>
__device__ inline M3<T> skew(V3<T> w) {
  T O = mkT(0.f, (T *)nullptr);
  M3<T> K;
  K.m[0][0] = O, K.m[0][1] = -w.z, K.m[0][2] = w.y;
  K.m[1][0] = w.z, K.m[1][1] = O, K.m[1][2] = -w.x;
  K.m[2][0] = -w.y, K.m[2][1] = w.x, K.m[2][2] = O;
  return K;
}
template <typename T>
__device__ inline M3<T> matmul(const M3<T> &a, const M3<T> &b) {
  M3<T> c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return c;
}
template <typename T>
__device__ inline V3<T> matvec(const M3<T> &a, V3<T> v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
// I*ci + a*K + b*K2
template <typename T>
__device__ inline M3<T> eye_plus(float ci, T a, const M3<T> &K, T b, const M3<T> &K2) {
  M3<T> r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = (i == j ? ci : 0.f) + a * K.m[i][j] + b * K2.m[i][j];
  return r;
}

// L2 norm with torch's sub-gradient 0 at the origin (the zero-initialised heads sit exactly there)
template <typename T>
__device__ inline T norm3(V3<T> w) {
  T n2 = w.x * w.x + w.y * w.y + w.z * w.z;
  if (val(n2) == 0.f) return mkT(0.f, (T *)nullptr);
  return Sqrt(n2);
}

// spline_utils.py:20-54: sum_{i<=10} (-1)^i x^(2i) / denom_i, the reference's term order
template <typename T>
__device__ inline T taylor(T x, int kind) {
  T x2 = x * x, p = mkT(1.f, (T *)nullptr), ans = mkT(0.f, (T *)nullptr);
  double denom = 1.0;
  for (int i = 0; i <= 10; ++i) {
    double step = kind == 0 ? (i > 0 ? (2.0 * i) * (2.0 * i + 1) : 1.0)
                            : (kind == 1 ? (2.0 * i + 1) * (2.0 * i + 2) : (2.0 * i + 2) * (2.0 * i + 3));
    denom *= step;
    if (i > 0) p = p * x2;
    ans = ans + ((i & 1) ? -1.f : 1.f) * p / (float)denom;
  }
  return ans;
}

// pypose-style guarded coefficient: theta^2 > eps ? big(theta) : small(theta^2)
#define D4GS_GUARDED(t2, th, BIG, SMALL) ((t2).v > 1e-12f ? ([&](Dual th) { return BIG; })(Sqrt(t2)) : ([&](Dual x) { return SMALL; })(t2))

struct Q4 {
  V3<Dual> v;
  Dual w;
};

__device__ inline Q4 so3_exp(V3<Dual> phi) {
  Dual t2 = dot(phi, phi);
  Dual im = D4GS_GUARDED(t2, th, Sin(0.5f * th) / th, 0.5f - x / 48.f + x * x / 3840.f);
  Dual re = D4GS_GUARDED(t2, th, Cos(0.5f * th), 1.f - x / 8.f + x * x / 384.f);
  return {{phi.x * im, phi.y * im, phi.z * im}, re};
}
__device__ inline V3<Dual> so3_log(Q4 q) {
  Dual n2 = dot(q.v, q.v);
  Dual w = q.w;
  Dual f = D4GS_GUARDED(n2, n, 2.f * Atan(n / w) / n, 2.f / w - 2.f * x / (3.f * w * w * w));
  return {f * q.v.x, f * q.v.y, f * q.v.z};
}
__device__ inline Q4 so3_mul(Q4 p, Q4 q) {
  V3<Dual> c = cross(p.v, q.v);
  return {{p.w * q.v.x + q.w * p.v.x + c.x, p.w * q.v.y + q.w * p.v.y + c.y, p.w * q.v.z + q.w * p.v.z + c.z},
          p.w * q.w - dot(p.v, q.v)};
}
__device__ inline Q4 so3_inv(Q4 q) { return {{-q.v.x, -q.v.y, -q.v.z}, q.w}; }

__device__ inline M3<Dual> left_jacobian(V3<Dual> phi) {
  Dual t2 = dot(phi, phi);
  M3<Dual> K = skew(phi), K2 = matmul(K, K);
  Dual c1 = D4GS_GUARDED(t2, th, (1.f - Cos(th)) / (th * th), 0.5f - x / 24.f);
  Dual c2 = D4GS_GUARDED(t2, th, (th - Sin(th)) / (th * th * th), 1.f / 6.f - x / 120.f);
  return eye_plus(1.f, c1, K, c2, K2);
}
__device__ inline M3<Dual> left_jacobian_inv(V3<Dual> phi) {
  Dual t2 = dot(phi, phi);
  M3<Dual> K = skew(phi), K2 = matmul(K, K);
  Dual c2 = D4GS_GUARDED(t2, th, (1.f - th * Cos(0.5f * th) / (2.f * Sin(0.5f * th))) / (th * th), 1.f / 12.f + x / 720.f);
  return eye_plus(1.f, mk(-0.5f), K, c2, K2);
}

struct SE3 {
  V3<Dual> t;
  Q4 q;
};
__device__ inline SE3 se3_exp(const Dual *xi) {  // [tau, phi]
  V3<Dual> tau = {xi[0], xi[1], xi[2]}, phi = {xi[3], xi[4], xi[5]};
  return {matvec(left_jacobian(phi), tau), so3_exp(phi)};
}

// ---- forward: thread (s, j) -------------------------------------------------------------------------------
__global__ void k_camera_path(const float *__restrict__ delta0, const float *__restrict__ delta1, int S,
                              const float *__restrict__ time_params, int index, float t, int moving,
                              float *__restrict__ RTs, float *__restrict__ jac, float *__restrict__ times,
                              float *__restrict__ dtimes, float *__restrict__ deltaT) {
  int tid = threadIdx.x + blockIdx.x * blockDim.x;
  int s = tid / 12, j = tid % 12;
  if (s >= S) return;
  Dual a[6], b[6];
  for (int k = 0; k < 6; ++k) a[k] = mk(delta0[k], j == k ? 1.f : 0.f), b[k] = mk(delta1[k], j == k + 6 ? 1.f : 0.f);
  SE3 X0 = se3_exp(a), X1 = se3_exp(b);
  // torch.linspace(0, 1, S): start + step*i below the midpoint, end - step*(S-1-i) above it
  float stepu = S > 1 ? 1.f / (float)(S - 1) : 0.f;
  float u = s < S / 2 ? stepu * (float)s : 1.f - stepu * (float)(S - 1 - s);
  if (S == 1) u = 0.f;
  V3<Dual> tt = {(1.f - u) * X0.t.x + u * X1.t.x, (1.f - u) * X0.t.y + u * X1.t.y, (1.f - u) * X0.t.z + u * X1.t.z};
  V3<Dual> r = so3_log(so3_mul(so3_inv(X0.q), X1.q));
  Q4 q = so3_mul(X0.q, so3_exp({u * r.x, u * r.y, u * r.z}));
  // SE3.Log -> [tau, phi]; se3_to_SE3 reads it as [w, u]
  V3<Dual> phi = so3_log(q);
  V3<Dual> w = matvec(left_jacobian_inv(phi), tt);
  M3<Dual> wx = skew(w), wx2 = matmul(wx, wx);
  Dual theta = norm3(w);
  Dual A = taylor(theta, 0), B = taylor(theta, 1), Cc = taylor(theta, 2);
  M3<Dual> R = eye_plus(1.f, A, wx, B, wx2), V = eye_plus(1.f, B, wx, Cc, wx2);
  V3<Dual> Vu = matvec(V, phi);
  Dual out[12] = {R.m[0][0], R.m[0][1], R.m[0][2], Vu.x, R.m[1][0], R.m[1][1], R.m[1][2], Vu.y,
                  R.m[2][0], R.m[2][1], R.m[2][2], Vu.z};
  for (int i = 0; i < 12; ++i) {
    if (jac) jac[((size_t)s * 12 + i) * 12 + j] = out[i].d;
    if (j == 0) RTs[s * 12 + i] = out[i].v;
  }
  if (j == 0) {
    // move_model.py:118-135,151-158: half-width d = clamp(relu(p), 0.1, 0.9) on interior frames of stage 2, else 0
    float d = 0.f, gate = 0.f;
    if (moving) {
      float p = time_params[index];
      float rl = p > 0.f ? p : 0.f;
      d = fminf(fmaxf(rl, 0.1f), 0.9f);
      gate = (p > 0.f && rl >= 0.1f && rl <= 0.9f) ? 1.f : 0.f;
    }
    float wgt = (float)s / (float)(S - 1);  // arange(S) / (S-1); NaN for S == 1 exactly as the reference
    float t0 = d * -1.0f + t, t1 = d * 1.0f + t;
    times[s] = t0 * (1.0f - wgt) + t1 * wgt;
    dtimes[s] = gate * (wgt - (1.0f - wgt));
    if (s == 0) deltaT[0] = fabsf(d * 1.0f), deltaT[1] = gate;  // [value, d value / d p]
  }
}

// ---- backward: v_delta[j] = sum_{s,i} v_RTs[s,i] * jac[s,i,j];  v_time_params[index] ---------------------
__global__ void k_camera_path_bwd(const float *__restrict__ jac, const float *__restrict__ dtimes,
                                  const float *__restrict__ deltaT, const float *__restrict__ v_RTs,
                                  const float *__restrict__ v_times, const float *__restrict__ v_deltaT, int S,
                                  int index, int n_time_params, float *__restrict__ v_delta0,
                                  float *__restrict__ v_delta1, float *__restrict__ v_time_params) {
  int j = threadIdx.x;
  if (j < 12) {
    float acc = 0.f;
    if (v_RTs)
      for (int k = 0; k < S * 12; ++k) acc += v_RTs[k] * jac[(size_t)k * 12 + j];
    (j < 6 ? v_delta0[j] : v_delta1[j - 6]) = acc;
  } else if (j < 12 + n_time_params) {
    int p = j - 12;
    float acc = 0.f;
    if (p == index) {
      if (v_times)
        for (int s = 0; s < S; ++s) acc += v_times[s] * dtimes[s];
      if (v_deltaT) acc += v_deltaT[0] * deltaT[1];
    }
    v_time_params[p] = acc;
  }
}

// ---- pose encoding: SE3_to_se3 + positional embedding -----------------------------------------------------
// MoveModel.preprocessPose (spline_utils.py:177-195) on plain floats (forward) or dual numbers (input gradient)
template <typename T>
__device__ inline void pose_to_se3(const T *R, const T *Tv, T *x) {
  // SO3_to_so3 (spline_utils.py:177-184)
  T trace = R[0] + R[4] + R[8];
  T c = Clamp((trace - 1.f) / 2.f, -1.f + 1e-7f, 1.f - 1e-7f);
  T theta = Fmod(Acos(c), 3.14159265358979323846f);
  T A = taylor(theta, 0);
  T f = 1.f / (2.f * A + 1e-8f);
  V3<T> w = {f * (R[7] - R[5]), f * (R[2] - R[6]), f * (R[3] - R[1])};
  // SE3_to_se3 (spline_utils.py:187-195)
  M3<T> wx = skew(w), wx2 = matmul(wx, wx);
  T th = norm3(w);
  T A2 = taylor(th, 0), B2 = taylor(th, 1);
  T cc = (1.f - A2 / (2.f * B2)) / (th * th + 1e-8f);
  M3<T> invV = eye_plus(1.f, mkT(-0.5f, (T *)nullptr), wx, cc, wx2);
  V3<T> tv = {Tv[0], Tv[1], Tv[2]};
  V3<T> u = matvec(invV, tv);
  x[0] = w.x, x[1] = w.y, x[2] = w.z, x[3] = u.x, x[4] = u.y, x[5] = u.z;
}

__global__ void k_pose_encode(const float *__restrict__ Rp, int r_stride, const float *__restrict__ Tp, int t_stride,
                              float *__restrict__ enc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float R[9], T[3], x[6];
  for (int i = 0; i < 3; ++i) {
    T[i] = Tp[i * t_stride];
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = Rp[i * r_stride + j];
  }
  pose_to_se3(R, T, x);
  // move_model.py:12-63: [x, sin(x f), cos(x f)] for f = 1, 2, 4, 8, 16
  for (int k = 0; k < 6; ++k) enc[k] = x[k];
  float fr = 1.f;
  for (int l = 0; l < 5; ++l, fr *= 2.f)
    for (int k = 0; k < 6; ++k) {
      enc[6 + 12 * l + k] = sinf(x[k] * fr);
      enc[6 + 12 * l + 6 + k] = cosf(x[k] * fr);
    }
}

// Input gradient of the pose encoding (test-time pose refinement differentiates the render w.r.t. w2c, which also
// feeds this module: flow3d/validator.py:442-448 -> scene_model.py:249-256).  Thread j < 12 re-evaluates
// SE3_to_se3 in dual numbers with the tangent of input j (R row-major 0..8, T 9..11) and contracts it with the
// embedding's adjoint: v_in[j] = sum_k (v_enc[k] + sum_l f_l (cos(x_k f_l) v_sin - sin(x_k f_l) v_cos)) dx_k/d in_j.
__global__ void k_pose_encode_bwd(const float *__restrict__ Rp, int r_stride, const float *__restrict__ Tp, int t_stride,
                                  const float *__restrict__ v_enc, float *__restrict__ v_R, float *__restrict__ v_T) {
  const int j = threadIdx.x;
  if (j >= 12 || blockIdx.x != 0) return;
  Dual R[9], T[3], x[6];
  for (int i = 0; i < 3; ++i) {
    T[i] = mk(Tp[i * t_stride], j == 9 + i ? 1.f : 0.f);
    for (int c = 0; c < 3; ++c) R[i * 3 + c] = mk(Rp[i * r_stride + c], j == i * 3 + c ? 1.f : 0.f);
  }
  pose_to_se3(R, T, x);
  float acc = 0.f;
  for (int k = 0; k < 6; ++k) {
    float vx = v_enc[k], fr = 1.f;
    for (int l = 0; l < 5; ++l, fr *= 2.f)
      vx += fr * (cosf(x[k].v * fr) * v_enc[6 + 12 * l + k] - sinf(x[k].v * fr) * v_enc[6 + 12 * l + 6 + k]);
    acc += vx * x[k].d;
  }
  if (j < 9) v_R[j] = acc;
  else v_T[j - 9] = acc;
}

// ---- the MoveModel MLP (move_model.py:66-110): 66 -> 64 x4 (LeakyReLU 0.01) -> 64, two heads 64 -> 64 -> 6 ---------
// 30 k parameters, batch 1: in eager PyTorch 9 GEMV launches forward and ~30 backward per render.  One block here.
// Layer order everywhere: main.0, main.2, main.4, main.6, main.8, head0.0, head0.2, head1.0, head1.2.
// acts: x0[66] a1[64] a2[64] a3[64] a4[64] m[64] ua[64] wa[64]  (post-activation inputs of every layer)
constexpr int MLP_IN = 66, MLP_W = 64, MLP_ACTS = MLP_IN + 7 * MLP_W;
constexpr float MLP_SLOPE = 0.01f;

// y[0..out) = act(b + W x); 4 lanes per output (quad reduction), 256 threads
__device__ __forceinline__ void mlp_layer(const float *__restrict__ W, const float *__restrict__ b, const float *x, int in,
                                          int out, bool act, float *y) {
  const int j = threadIdx.x >> 2, q = threadIdx.x & 3;
  float acc = 0.f;
  if (j < out)
    for (int i = q; i < in; i += 4) acc = __builtin_fmaf(W[j * in + i], x[i], acc);
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  if (j < out && q == 0) {
    float v = acc + b[j];
    y[j] = act ? (v > 0.f ? v : MLP_SLOPE * v) : v;
  }
  __syncthreads();
}

struct MlpPtrs {
  const float *w[9];
  const float *b[9];
};
struct MlpGradPtrs {
  float *w[9];
  float *b[9];
};

__global__ void __launch_bounds__(256) k_move_mlp_fwd(const float *__restrict__ enc, MlpPtrs p, float *__restrict__ acts,
                                                      float *__restrict__ delta0, float *__restrict__ delta1) {
  __shared__ float sx[MLP_ACTS];
  __shared__ float sd[12];
  for (int i = threadIdx.x; i < MLP_IN; i += 256) sx[i] = enc[i];
  __syncthreads();
  float *a1 = sx + MLP_IN, *a2 = a1 + 64, *a3 = a2 + 64, *a4 = a3 + 64, *m = a4 + 64, *ua = m + 64, *wa = ua + 64;
  mlp_layer(p.w[0], p.b[0], sx, MLP_IN, 64, true, a1);
  mlp_layer(p.w[1], p.b[1], a1, 64, 64, true, a2);
  mlp_layer(p.w[2], p.b[2], a2, 64, 64, true, a3);
  mlp_layer(p.w[3], p.b[3], a3, 64, 64, true, a4);
  mlp_layer(p.w[4], p.b[4], a4, 64, 64, false, m);
  mlp_layer(p.w[5], p.b[5], m, 64, 64, true, ua);
  mlp_layer(p.w[7], p.b[7], m, 64, 64, true, wa);
  mlp_layer(p.w[6], p.b[6], ua, 64, 6, false, sd);
  mlp_layer(p.w[8], p.b[8], wa, 64, 6, false, sd + 6);
  for (int i = threadIdx.x; i < MLP_ACTS; i += 256) acts[i] = sx[i];
  if (threadIdx.x < 6) delta0[threadIdx.x] = sd[threadIdx.x];
  else if (threadIdx.x < 12) delta1[threadIdx.x - 6] = sd[threadIdx.x];
}

// one layer of the backward: gW = vz (x) x, gb = vz, vx = W^T vz (then times the LeakyReLU slope of x's layer)
__device__ __forceinline__ void mlp_layer_bwd(const float *__restrict__ W, const float *vz, const float *x, int in, int out,
                                              float *__restrict__ gW, float *__restrict__ gb, float *vx, bool add_vx,
                                              bool x_is_act) {
  for (int idx = threadIdx.x; idx < in * out; idx += 256) gW[idx] = vz[idx / in] * x[idx % in];
  if (threadIdx.x < out) gb[threadIdx.x] = vz[threadIdx.x];
  if (vx && threadIdx.x < in) {
    const int i = threadIdx.x;
    float acc = 0.f;
    for (int j = 0; j < out; j++) acc = __builtin_fmaf(W[j * in + i], vz[j], acc);
    if (x_is_act) acc *= x[i] > 0.f ? 1.f : MLP_SLOPE;  // a > 0 <=> z > 0
    vx[i] = add_vx ? vx[i] + acc : acc;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_move_mlp_bwd(const float *__restrict__ acts, MlpPtrs p, const float *__restrict__ v_delta0,
                                                      const float *__restrict__ v_delta1, MlpGradPtrs g,
                                                      float *__restrict__ v_enc) {
  __shared__ float sx[MLP_ACTS];
  __shared__ float va[64], vb[64], vd[12], ve[MLP_IN];
  for (int i = threadIdx.x; i < MLP_ACTS; i += 256) sx[i] = acts[i];
  if (threadIdx.x < 6) vd[threadIdx.x] = v_delta0[threadIdx.x];
  else if (threadIdx.x < 12) vd[threadIdx.x] = v_delta1[threadIdx.x - 6];
  __syncthreads();
  const float *a1 = sx + MLP_IN, *a2 = a1 + 64, *a3 = a2 + 64, *a4 = a3 + 64, *m = a4 + 64, *ua = m + 64, *wa = ua + 64;
  // heads: vd -> v_u (va) -> v_m (vb), then vd+6 -> v_w (va) -> v_m += 
  mlp_layer_bwd(p.w[6], vd, ua, 64, 6, g.w[6], g.b[6], va, false, true);
  mlp_layer_bwd(p.w[5], va, m, 64, 64, g.w[5], g.b[5], vb, false, false);
  mlp_layer_bwd(p.w[8], vd + 6, wa, 64, 6, g.w[8], g.b[8], va, false, true);
  mlp_layer_bwd(p.w[7], va, m, 64, 64, g.w[7], g.b[7], vb, true, false);
  // trunk: v_z5 = vb
  mlp_layer_bwd(p.w[4], vb, a4, 64, 64, g.w[4], g.b[4], va, false, true);
  mlp_layer_bwd(p.w[3], va, a3, 64, 64, g.w[3], g.b[3], vb, false, true);
  mlp_layer_bwd(p.w[2], vb, a2, 64, 64, g.w[2], g.b[2], va, false, true);
  mlp_layer_bwd(p.w[1], va, a1, 64, 64, g.w[1], g.b[1], vb, false, true);
  mlp_layer_bwd(p.w[0], vb, sx, MLP_IN, 64, g.w[0], g.b[0], v_enc ? ve : nullptr, false, false);
  if (v_enc && threadIdx.x < MLP_IN) v_enc[threadIdx.x] = ve[threadIdx.x];
}

}  // namespace

int d4gs_pose_encode_impl(const float *R, int32_t r_stride, const float *T, int32_t t_stride, float *enc,
                          hipStream_t stream) {
  ProfScope ps("k_pose_encode", stream);
  k_pose_encode<<<1, 64, 0, stream>>>(R, r_stride, T, t_stride, enc);
  return d4gs_check_launch("k_pose_encode");
}

int d4gs_pose_encode_bwd_impl(const float *R, int32_t r_stride, const float *T, int32_t t_stride, const float *v_enc,
                              float *v_R, float *v_T, hipStream_t stream) {
  ProfScope ps("k_pose_encode_bwd", stream);
  k_pose_encode_bwd<<<1, 64, 0, stream>>>(R, r_stride, T, t_stride, v_enc, v_R, v_T);
  return d4gs_check_launch("k_pose_encode_bwd");
}

int d4gs_camera_path_fwd_impl(const float *delta0, const float *delta1, int32_t S, const float *time_params,
                              int32_t index, float t, int32_t moving, float *RTs, float *jac, float *times,
                              float *dtimes, float *deltaT, hipStream_t stream) {
  ProfScope ps("k_camera_path", stream);
  int threads = S * 12;
  k_camera_path<<<(threads + 63) / 64, 64, 0, stream>>>(delta0, delta1, S, time_params, index, t, moving, RTs, jac,
                                                         times, dtimes, deltaT);
  return d4gs_check_launch("k_camera_path");
}

int d4gs_camera_path_bwd_impl(const float *jac, const float *dtimes, const float *deltaT, const float *v_RTs,
                              const float *v_times, const float *v_deltaT, int32_t S, int32_t index,
                              int32_t n_time_params, float *v_delta0, float *v_delta1, float *v_time_params,
                              hipStream_t stream) {
  ProfScope ps("k_camera_path_bwd", stream);
  k_camera_path_bwd<<<1, 64, 0, stream>>>(jac, dtimes, deltaT, v_RTs, v_times, v_deltaT, S, index, n_time_params,
                                          v_delta0, v_delta1, v_time_params);
  return d4gs_check_launch("k_camera_path_bwd");
}

int d4gs_move_model_fwd_impl(const float *R, int32_t r_stride, const float *T, int32_t t_stride, const float *const *w,
                             const float *const *b, int32_t S, const float *time_params, int32_t index, float t,
                             int32_t moving, float *enc, float *acts, float *delta, float *RTs, float *jac, float *times,
                             float *dtimes, float *deltaT, hipStream_t stream) {
  int rc = d4gs_pose_encode_impl(R, r_stride, T, t_stride, enc, stream);
  if (rc) return rc;
  MlpPtrs p;
  for (int l = 0; l < 9; l++) p.w[l] = w[l], p.b[l] = b[l];
  {
    ProfScope ps("k_move_mlp_fwd", stream);
    k_move_mlp_fwd<<<1, 256, 0, stream>>>(enc, p, acts, delta, delta + 6);
  }
  rc = d4gs_check_launch("k_move_mlp_fwd");
  if (rc) return rc;
  return d4gs_camera_path_fwd_impl(delta, delta + 6, S, time_params, index, t, moving, RTs, jac, times, dtimes, deltaT,
                                   stream);
}

int d4gs_move_model_bwd_impl(const float *jac, const float *dtimes, const float *deltaT, const float *acts,
                             const float *const *w, const float *const *b, const float *v_RTs, const float *v_times,
                             const float *v_deltaT, int32_t S, int32_t index, int32_t n_time_params, float *v_delta,
                             float *const *v_w, float *const *v_b, float *v_time_params, float *v_enc,
                             hipStream_t stream) {
  int rc = d4gs_camera_path_bwd_impl(jac, dtimes, deltaT, v_RTs, v_times, v_deltaT, S, index, n_time_params, v_delta,
                                     v_delta + 6, v_time_params, stream);
  if (rc) return rc;
  MlpPtrs p;
  MlpGradPtrs g;
  for (int l = 0; l < 9; l++) p.w[l] = w[l], p.b[l] = b[l], g.w[l] = v_w[l], g.b[l] = v_b[l];
  ProfScope ps("k_move_mlp_bwd", stream);
  k_move_mlp_bwd<<<1, 256, 0, stream>>>(acts, p, v_delta, v_delta + 6, g, v_enc);
  return d4gs_check_launch("k_move_mlp_bwd");
}
