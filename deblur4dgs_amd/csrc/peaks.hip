// peaks.hip -- d4gs_measure_peaks: the two MEASURED device ceilings bench.py's roofline objects are quoted against
// ("report roofline.achieved against *measured* HBM peak (device-to-device stream-copy kernel on the box) and,
// separately, against measured FP32 FMA peak").  Diagnostic entry point like d4gs_profile_*: it creates its own HIP events and
// WAITS for them, so it must not be called inside a timed region or a stream capture.  No reference counterpart (the reference
// has no measurement harness: SURVEY section 6).
#include "common.h"

namespace {

// device-to-device stream copy, 16 bytes per lane and access, grid-stride.  Shape picked by scripts/microbench/stream_copy.hip
// (profiles/r04_x_stream_copy.txt): non-temporal accesses, 8 loads in flight per lane, 256 workgroups per CU - 5.9 TB/s on 2 x 512 MiB,
// 5.3 TB/s on 2 x 2 GiB (the 256 MB MALL no longer helps), against 4.6 - 5.1 for plain accesses or grids of 8 - 16 workgroups per CU
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_peak_copy(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {  // eight independent 16-byte loads in flight per lane
    v4f r[8];
#pragma unroll
    for (int u = 0; u < 8; u++) r[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; u++) __builtin_nontemporal_store(r[u], dst + i + u * stride);
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

// FMA issue loop: NACC independent accumulator chains per lane, `iters` rounds.  PACKED: v_pk_fma_f32 (two fp32 FMAs per lane and
// instruction - the form the 157.3 TFLOP/s vector figure is quoted for: 1024 SIMDs x 16 lanes x 2 x 2 flop x 2.4 GHz); plain:
// v_fma_f32, the instruction the composite kernels actually issue.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NACC = 16;
template <bool PACKED>
__global__ void __launch_bounds__(256) k_peak_fma(float *out, int iters, float seed) {
  const float x0 = seed + 1e-7f * (float)threadIdx.x;
  if constexpr (PACKED) {
    f32x2 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = f32x2{x0 + (float)k, x0 - (float)k};
    const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < NACC; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(m), "v"(c));
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; k++) s += acc[k][0] + acc[k][1];
    if (s == 123.456f) out[0] = s;  // keeps the chains alive; never true in practice
  } else {
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = x0 + (float)k;
    for (int it = 0; it < iters; it++) {  // acc = acc * 0.5 + 1.0: inline constants, no operand-bank traffic besides the chain
#pragma unroll
      for (int k = 0; k < NACC; k++) asm volatile("v_fma_f32 %0, %0, 0.5, 1.0" : "+v"(acc[k]));
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; k++) s += acc[k];
    if (s == 123.456f) out[0] = s;
  }
}

struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
  bool ok;
  Timer() { ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
  ~Timer() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

}  // namespace

extern "C" int d4gs_measure_peaks(void *scratch, size_t scratch_bytes, double *out /* host [4] */, void *stream_) {
  if (!scratch || !out || scratch_bytes < (size_t)64 << 20) {
    d4gs_set_error("d4gs_measure_peaks: needs a device scratch buffer of >= 64 MiB and a host double[4]");
    return D4GS_EINVAL;
  }
  hipStream_t stream = (hipStream_t)stream_;
  Timer t;
  if (!t.ok) {
    d4gs_set_error("d4gs_measure_peaks: hipEventCreate failed");
    return D4GS_ELAUNCH;
  }
  int cus = 256;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // ---- stream copy: first half -> second half, best of 8 after 2 warm-ups
  const size_t half = (scratch_bytes / 2) & ~(size_t)4095, n4 = half / 16;
  const v4f *src = reinterpret_cast<const v4f *>(scratch);
  v4f *dst = reinterpret_cast<v4f *>(reinterpret_cast<char *>(scratch) + half);
  float best_copy = 1e30f;
  for (int rep = 0; rep < 10; rep++) {
    (void)hipEventRecord(t.a, stream);
    hipLaunchKernelGGL(k_peak_copy, dim3(cus * 256), dim3(256), 0, stream, src, dst, n4);
    (void)hipEventRecord(t.b, stream);
    if (hipEventSynchronize(t.b) != hipSuccess) break;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t.a, t.b);
    if (rep >= 2 && ms > 0.f && ms < best_copy) best_copy = ms;
  }
  // ---- FMA issue: 8 blocks of 4 waves per CU = 8 waves / SIMD, 2048 rounds of 16 chains
  const int iters = 2048, blocks = cus * 8;
  float best_pk = 1e30f, best_plain = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    float ms = 0.f;
    (void)hipEventRecord(t.a, stream);
    hipLaunchKernelGGL(k_peak_fma<true>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<float *>(scratch), iters, 1.0f);
    (void)hipEventRecord(t.b, stream);
    if (hipEventSynchronize(t.b) != hipSuccess) break;
    (void)hipEventElapsedTime(&ms, t.a, t.b);
    if (rep >= 1 && ms > 0.f && ms < best_pk) best_pk = ms;
    (void)hipEventRecord(t.a, stream);
    hipLaunchKernelGGL(k_peak_fma<false>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<float *>(scratch), iters, 1.0f);
    (void)hipEventRecord(t.b, stream);
    if (hipEventSynchronize(t.b) != hipSuccess) break;
    (void)hipEventElapsedTime(&ms, t.a, t.b);
    if (rep >= 1 && ms > 0.f && ms < best_plain) best_plain = ms;
  }
  int rc = d4gs_check_launch("d4gs_measure_peaks");
  if (rc) return rc;
  const double lanes = (double)blocks * 256.0;
  out[0] = best_copy < 1e29f ? 2.0 * (double)half / (best_copy * 1e-3) / 1e9 : 0.0;                          // GB/s, read + write
  out[1] = best_pk < 1e29f ? lanes * iters * NACC * 4.0 / (best_pk * 1e-3) / 1e12 : 0.0;                     // TFLOP/s, v_pk_fma_f32
  out[2] = best_plain < 1e29f ? lanes * iters * NACC * 2.0 / (best_plain * 1e-3) / 1e12 : 0.0;               // TFLOP/s, v_fma_f32
  out[3] = (double)half;                                                                                      // bytes copied per launch
  return D4GS_OK;
}
