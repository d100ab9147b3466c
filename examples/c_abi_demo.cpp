// c_abi_demo.cpp -- the drop-in boundary without PyTorch: a host program that links libd4gs.so and the HIP runtime only.
// It builds a small synthetic static scene, renders one frame forward + backward through the one-call entry points
// d4gs_forward / d4gs_backward (device pointers from hipMalloc, an explicit stream, one caller-provided workspace) and through
// their CPU twins d4gs_forward_cpu / d4gs_backward_cpu (host pointers), and compares the two products: image, alpha and the
// gradient of the means.  This is what a non-Python host of the reference's render path (flow3d/scene_model.py:313-397) binds.
// Then the threading contract of the boundary (the reference renders from its trainer thread and from its viewer thread,
// flow3d/trainer.py:204-207, flow3d/renderer.py:57-89): two host threads, each with its own stream, outputs and workspace, render the
// same scene concurrently ten times - every frame and gradient must be BITWISE the single-threaded one - while one of them also makes a
// failing call: its d4gs_last_error() names that failure, the other thread's stays empty (the string is thread-local).
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_abi_demo.cpp -Ldeblur4dgs_amd -ld4gs -Wl,-rpath,$PWD/deblur4dgs_amd -o /tmp/c_abi_demo
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "d4gs.h"

#define HIP(x)                                                                    \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 2;                                                                   \
    }                                                                             \
  } while (0)
#define D4(x)                                                                     \
  do {                                                                            \
    int rc_ = (x);                                                                \
    if (rc_ != D4GS_OK) {                                                         \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, d4gs_last_error());              \
      return 3;                                                                   \
    }                                                                             \
  } while (0)

static float frand(unsigned &s) {  // LCG in [0, 1)
  s = s * 1664525u + 1013904223u;
  return (float)(s >> 8) / 16777216.f;
}
static float nrand(unsigned &s) { return sqrtf(-2.f * logf(frand(s) + 1e-7f)) * cosf(6.2831853f * frand(s)); }

template <class T>
static T *to_dev(const std::vector<T> &h) {
  T *d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
static double max_abs(const std::vector<float> &a) {
  double m = 0;
  for (float v : a) m = fmax(m, fabs((double)v));
  return m;
}
static double max_diff(const std::vector<float> &a, const std::vector<float> &b) {
  double m = 0;
  for (size_t i = 0; i < a.size(); i++) m = fmax(m, fabs((double)a[i] - (double)b[i]));
  return m;
}
static double frac_off(const std::vector<float> &a, const std::vector<float> &b, double tol) {
  size_t n = 0;
  for (size_t i = 0; i < a.size(); i++) n += fabs((double)a[i] - (double)b[i]) > tol;
  return (double)n / (double)a.size();
}

// One host thread of the threading check: its own stream, outputs and workspace; the inputs are shared and read-only.
struct ThreadRun {
  std::vector<float> blend, v_means;
  std::string err_after_bad_call, err_otherwise;
  int rc = 0;
};
static int thread_body(const D4gsDims *dims, const D4gsProjIn *in_d, const float *bg_d, const float *d_wimg, const float *d_wacc, int64_t cap,
                       int reps, bool provoke, ThreadRun *out) {
  const int N = dims->N, S = dims->S, NCH = dims->D + 1;
  const size_t P = (size_t)dims->height * dims->width;
  hipStream_t stream;
  HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  float *d_blend, *d_acc, *d_rend, *d_alpha, *d_m2d, *dv_means, *dv_quats, *dv_scales, *dv_opac, *dv_colors, *dv_view, *dv_m2d;
  int32_t *d_radii;
  int64_t *d_n;
  HIP(hipMalloc(&d_blend, P * NCH * 4));
  HIP(hipMalloc(&d_acc, P * 4));
  HIP(hipMalloc(&d_rend, S * P * NCH * 4));
  HIP(hipMalloc(&d_alpha, S * P * 4));
  HIP(hipMalloc(&d_m2d, (size_t)S * N * 8));
  HIP(hipMalloc(&d_radii, (size_t)S * N * 4));
  HIP(hipMalloc(&d_n, 4 * sizeof(int64_t)));
  HIP(hipMalloc(&dv_means, N * 12));
  HIP(hipMalloc(&dv_quats, N * 16));
  HIP(hipMalloc(&dv_scales, N * 12));
  HIP(hipMalloc(&dv_opac, N * 4));
  HIP(hipMalloc(&dv_colors, (size_t)N * dims->D * 4));
  HIP(hipMalloc(&dv_view, 64));
  HIP(hipMalloc(&dv_m2d, (size_t)S * N * 8));
  const size_t ws_bytes = d4gs_frame_workspace_bytes(dims, cap);
  void *ws;
  HIP(hipMalloc(&ws, ws_bytes));
  D4gsFrameIO io;
  memset(&io, 0, sizeof io);
  io.blended = d_blend, io.acc = d_acc, io.renders = d_rend, io.alphas = d_alpha, io.means2d = d_m2d, io.radii = d_radii, io.n_isect = d_n;
  io.background = bg_d;
  D4gsFrameGrads fg;
  memset(&fg, 0, sizeof fg);
  fg.v_blended = d_wimg, fg.v_acc = d_wacc, fg.v_means2d = dv_m2d;
  D4gsLeafGrads lg;
  memset(&lg, 0, sizeof lg);
  lg.v_means = dv_means, lg.v_quats = dv_quats, lg.v_scales = dv_scales, lg.v_opacities = dv_opac, lg.v_colors = dv_colors, lg.v_viewmat = dv_view;
  for (int r = 0; r < reps; r++) {
    D4(d4gs_forward(dims, in_d, &io, ws, ws_bytes, cap, 0, stream));
    if (provoke && r == reps / 2) {  // a failing call in the middle of this thread's frames: only THIS thread's string may change
      if (d4gs_forward(nullptr, in_d, &io, ws, ws_bytes, cap, 0, stream) == D4GS_OK) return 5;
      out->err_after_bad_call = d4gs_last_error();
    }
    D4(d4gs_backward(dims, in_d, &io, &fg, &lg, ws, ws_bytes, cap, 0, stream));
  }
  HIP(hipStreamSynchronize(stream));
  if (!provoke) out->err_otherwise = d4gs_last_error();
  out->blend.resize(P * NCH), out->v_means.resize((size_t)N * 3);
  HIP(hipMemcpy(out->blend.data(), d_blend, out->blend.size() * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(out->v_means.data(), dv_means, out->v_means.size() * 4, hipMemcpyDeviceToHost));
  return 0;
}

int main() {
  const int N = 4000, W = 160, H = 96, S = 1, D = 3, NCH = D + 1;
  unsigned seed = 12345u;
  std::vector<float> means(N * 3), quats(N * 4), scales(N * 3), opac(N), colors(N * D);
  for (int g = 0; g < N; g++) {
    const float z = 2.f + 8.f * frand(seed);
    means[g * 3] = z * (frand(seed) * 1.1f - 0.55f), means[g * 3 + 1] = z * (frand(seed) * 1.1f - 0.55f) * (float)H / W, means[g * 3 + 2] = z;
    for (int j = 0; j < 4; j++) quats[g * 4 + j] = nrand(seed);
    for (int j = 0; j < 3; j++) scales[g * 3 + j] = logf(0.03f) + 0.5f * nrand(seed);  // raw (log) leaves
    opac[g] = 1.5f * nrand(seed);                                                        // raw (logit)
    for (int j = 0; j < D; j++) colors[g * D + j] = nrand(seed);
  }
  std::vector<float> V = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, K = {(float)W, 0, W / 2.f, 0, (float)W, H / 2.f, 0, 0, 1};
  std::vector<float> bg = {1, 1, 1};
  D4gsDims dims;
  memset(&dims, 0, sizeof dims);
  dims.N = N, dims.S = S, dims.D = D, dims.width = W, dims.height = H, dims.depth_mode = D4GS_DEPTH_ED;
  dims.flags = D4GS_RAW_PARAMS | D4GS_RAW_COLORS | D4GS_EXACT_CULL, dims.n_sigmoid = 3;
  dims.near_plane = 0.01f, dims.far_plane = 1e10f, dims.eps2d = 0.3f;
  const size_t P = (size_t)H * W;
  std::vector<float> wimg(P * NCH), wacc(P);
  for (auto &v : wimg) v = nrand(seed);
  for (auto &v : wacc) v = nrand(seed);

  // ---- CPU twin: host pointers, no stream, no workspace ----
  std::vector<float> c_blend(P * NCH), c_acc(P), c_rend(S * P * NCH), c_alpha(S * P), c_m2d((size_t)S * N * 2);
  std::vector<int32_t> c_radii((size_t)S * N);
  int64_t c_n[4];
  D4gsProjIn in_h;
  memset(&in_h, 0, sizeof in_h);
  in_h.means = means.data(), in_h.quats = quats.data(), in_h.scales = scales.data(), in_h.opacities = opac.data();
  in_h.colors = colors.data(), in_h.viewmat = V.data(), in_h.Kmat = K.data();
  D4gsFrameIO io_h;
  memset(&io_h, 0, sizeof io_h);
  io_h.blended = c_blend.data(), io_h.acc = c_acc.data(), io_h.renders = c_rend.data(), io_h.alphas = c_alpha.data();
  io_h.means2d = c_m2d.data(), io_h.radii = c_radii.data(), io_h.n_isect = c_n, io_h.background = bg.data();
  D4(d4gs_forward_cpu(&dims, &in_h, &io_h));
  std::vector<float> cv_means(N * 3), cv_quats(N * 4), cv_scales(N * 3), cv_opac(N), cv_colors(N * D), cv_view(16), cv_m2d((size_t)S * N * 2);
  D4gsFrameGrads fg_h;
  memset(&fg_h, 0, sizeof fg_h);
  fg_h.v_blended = wimg.data(), fg_h.v_acc = wacc.data(), fg_h.v_means2d = cv_m2d.data();
  D4gsLeafGrads lg_h;
  memset(&lg_h, 0, sizeof lg_h);
  lg_h.v_means = cv_means.data(), lg_h.v_quats = cv_quats.data(), lg_h.v_scales = cv_scales.data(), lg_h.v_opacities = cv_opac.data();
  lg_h.v_colors = cv_colors.data(), lg_h.v_viewmat = cv_view.data();
  D4(d4gs_backward_cpu(&dims, &in_h, &io_h, &fg_h, &lg_h));

  // ---- device: hipMalloc'ed buffers, an explicit stream, one workspace ----
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  D4gsProjIn in_d = in_h;
  in_d.means = to_dev(means), in_d.quats = to_dev(quats), in_d.scales = to_dev(scales), in_d.opacities = to_dev(opac);
  in_d.colors = to_dev(colors), in_d.viewmat = to_dev(V), in_d.Kmat = to_dev(K);
  float *d_blend, *d_acc, *d_rend, *d_alpha, *d_m2d, *d_wimg = to_dev(wimg), *d_wacc = to_dev(wacc);
  int32_t *d_radii;
  int64_t *d_n;
  HIP(hipMalloc(&d_blend, P * NCH * 4));
  HIP(hipMalloc(&d_acc, P * 4));
  HIP(hipMalloc(&d_rend, S * P * NCH * 4));
  HIP(hipMalloc(&d_alpha, S * P * 4));
  HIP(hipMalloc(&d_m2d, (size_t)S * N * 8));
  HIP(hipMalloc(&d_radii, (size_t)S * N * 4));
  HIP(hipMalloc(&d_n, 4 * sizeof(int64_t)));
  D4gsFrameIO io_d;
  memset(&io_d, 0, sizeof io_d);
  io_d.blended = d_blend, io_d.acc = d_acc, io_d.renders = d_rend, io_d.alphas = d_alpha, io_d.means2d = d_m2d, io_d.radii = d_radii;
  io_d.n_isect = d_n, io_d.background = to_dev(bg);
  // the list capacity is the caller's guess, checked on the device: here the CPU twin's exact count + the usual headroom
  const int64_t cap = c_n[0] + c_n[0] / 4 + 4096;
  const size_t ws_bytes = d4gs_frame_workspace_bytes(&dims, cap);
  void *ws;
  HIP(hipMalloc(&ws, ws_bytes));
  D4(d4gs_forward(&dims, &in_d, &io_d, ws, ws_bytes, cap, 0, stream));
  int64_t *g_n;  // pinned: d4gs_copy_counts is a kernel that stores the counts into host memory the device can address
  HIP(hipHostMalloc((void **)&g_n, 4 * sizeof(int64_t), hipHostMallocDefault));
  D4(d4gs_copy_counts(d_n, g_n, stream));
  HIP(hipStreamSynchronize(stream));
  if (g_n[0] > cap) {
    fprintf(stderr, "list capacity %lld too small for %lld intersections\n", (long long)cap, (long long)g_n[0]);
    return 4;
  }
  float *dv_means, *dv_quats, *dv_scales, *dv_opac, *dv_colors, *dv_view, *dv_m2d;
  HIP(hipMalloc(&dv_means, N * 12));
  HIP(hipMalloc(&dv_quats, N * 16));
  HIP(hipMalloc(&dv_scales, N * 12));
  HIP(hipMalloc(&dv_opac, N * 4));
  HIP(hipMalloc(&dv_colors, N * D * 4));
  HIP(hipMalloc(&dv_view, 64));
  HIP(hipMalloc(&dv_m2d, (size_t)S * N * 8));
  D4gsFrameGrads fg_d;
  memset(&fg_d, 0, sizeof fg_d);
  fg_d.v_blended = d_wimg, fg_d.v_acc = d_wacc, fg_d.v_means2d = dv_m2d;
  D4gsLeafGrads lg_d;
  memset(&lg_d, 0, sizeof lg_d);
  lg_d.v_means = dv_means, lg_d.v_quats = dv_quats, lg_d.v_scales = dv_scales, lg_d.v_opacities = dv_opac, lg_d.v_colors = dv_colors;
  lg_d.v_viewmat = dv_view;
  D4(d4gs_backward(&dims, &in_d, &io_d, &fg_d, &lg_d, ws, ws_bytes, cap, 0, stream));
  HIP(hipStreamSynchronize(stream));
  std::vector<float> g_blend(P * NCH), g_acc(P), gv_means(N * 3), gv_colors(N * D);
  HIP(hipMemcpy(g_blend.data(), d_blend, g_blend.size() * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(g_acc.data(), d_acc, g_acc.size() * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(gv_means.data(), dv_means, gv_means.size() * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(gv_colors.data(), dv_colors, gv_colors.size() * 4, hipMemcpyDeviceToHost));

  const double tol = 2e-4;
  const double e_img = max_diff(g_blend, c_blend) / max_abs(c_blend), e_acc = max_diff(g_acc, c_acc) / fmax(max_abs(c_acc), 1e-30);
  const double f_img = frac_off(g_blend, c_blend, tol * max_abs(c_blend));
  const double f_means = frac_off(gv_means, cv_means, tol * max_abs(cv_means)), f_col = frac_off(gv_colors, cv_colors, tol * max_abs(cv_colors));
  printf("d4gs %d: %lld intersections on the device, %lld on the CPU twin (longest tile list %lld)\n", d4gs_version(), (long long)g_n[0],
         (long long)c_n[0], (long long)g_n[1]);
  printf("blended max|dev - cpu| / max = %.2e (%.1e of the elements beyond %.0e), acc %.2e\n", e_img, f_img, tol, e_acc);
  printf("v_means: %.1e of the elements beyond %.0e x max, v_colors: %.1e\n", f_means, tol, f_col);
  bool ok = g_n[0] == c_n[0] && f_img <= 2e-3 && f_means <= 2e-3 && f_col <= 2e-3 && e_acc < 2e-2;

  // ---- the threading contract: two host threads, two streams, concurrently; thread 1 also makes a failing call ----
  const int T = 2, REPS = 10;
  ThreadRun runs[T];
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([&, t]() { runs[t].rc = thread_body(&dims, &in_d, io_d.background, d_wimg, d_wacc, cap, REPS, t == 1, &runs[t]); });
  for (auto &x : th) x.join();
  bool same = true;
  for (int t = 0; t < T; t++)
    same = same && runs[t].rc == 0 && runs[t].blend.size() == g_blend.size() &&
           memcmp(runs[t].blend.data(), g_blend.data(), g_blend.size() * 4) == 0 &&
           memcmp(runs[t].v_means.data(), gv_means.data(), gv_means.size() * 4) == 0;
  const bool tls = !runs[1].err_after_bad_call.empty() && runs[0].err_otherwise.empty();
  printf("threads: %d x %d frames (forward + backward) on %d streams, bitwise the single-threaded frame: %s; error strings thread-local: %s (\"%s\" / \"%s\")\n",
         T, REPS, T, same ? "yes" : "NO", tls ? "yes" : "NO", runs[1].err_after_bad_call.c_str(), runs[0].err_otherwise.c_str());
  ok = ok && same && tls;
  printf(ok ? "OK\n" : "MISMATCH\n");
  return ok ? 0 : 1;
}
