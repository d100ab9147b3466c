// microbench: operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products, one per QUAD of lanes) and its
// issue cost.  Assumed (and checked bit for bit here): lane l supplies A[i = l & 3] and B[j = l & 3] of block l >> 2; lane l receives
// D[i][j = l & 3] in register i:   out[l][i] = fma(a[(l & ~3) + i], b[l], c[l][i]).
// The wide (17-channel) composites use it for their per-(pixel, splat) channel contractions (raster_fwd.hip / raster_bwd.hip, round 6).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_layout(const float *a, const float *b, const float *c, float *o) {
  const int l = threadIdx.x;
  f32x4 acc = {c[l * 4], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int i = 0; i < 4; i++) o[l * 4 + i] = acc[i];
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_rate(float *o, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.f + 1e-6f * threadIdx.x;
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 16 / CHAINS; r++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  o[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same 16 multiply-adds per lane on the VALU, for comparison (4 independent chains of v_fma_f32)
__global__ void __launch_bounds__(256) k_rate_valu(float *o, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.f + 1e-6f * threadIdx.x, x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) asm volatile("v_fma_f32 %0,%4,%5,%0\n v_fma_f32 %1,%4,%5,%1\n v_fma_f32 %2,%4,%5,%2\n v_fma_f32 %3,%4,%5,%3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  }
  o[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <typename K> void run(const char *name, K kern, float *o, int waves_per_simd, double per_iter) {
  const int iters = 4000, blocks = 256 * waves_per_simd;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  kern<<<blocks, 256>>>(o, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); kern<<<blocks, 256>>>(o, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double n = (double)iters * per_iter * waves_per_simd;
  printf("%-28s waves/SIMD %d: %.3f ms -> %.2f cycles (at 2.4 GHz) per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / n * 2.4);
}
int main() {
  float ha[64], hb[64], hc[256], ho[256], *a, *b, *c, *o;
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
  for (int i = 0; i < 64; i++) ha[i] = rnd() * 3.f, hb[i] = rnd() * 0.7f;
  for (int i = 0; i < 256; i++) hc[i] = rnd() * 5.f;
  (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&c, 1024); (void)hipMalloc(&o, 4 * 256 * 256 * 8);
  (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(a, b, c, o);
  (void)hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int i = 0; i < 4; i++) {
      const float want = fmaf(ha[(l & ~3) + i], hb[l], hc[l * 4 + i]);
      if (memcmp(&want, &ho[l * 4 + i], 4)) { if (bad++ < 8) printf("lane %d reg %d: got %.9g want %.9g\n", l, i, ho[l * 4 + i], want); }
    }
  printf("layout out[l][i] = fma(a[(l & ~3) + i], b[l], c[l][i]): %s (%d of 256 differ bitwise)\n", bad ? "WRONG" : "confirmed bit for bit", bad);
  for (int w : {1, 4, 8}) {
    run("mfma 4x4x1 1 chain (dependent)", k_rate<1>, o, w, 16);
    run("mfma 4x4x1 2 chains", k_rate<2>, o, w, 16);
    run("mfma 4x4x1 4 chains", k_rate<4>, o, w, 16);
    run("v_fma_f32 4 chains", k_rate_valu, o, w, 16);
  }
  return bad != 0;
}
