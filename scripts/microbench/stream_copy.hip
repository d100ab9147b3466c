// stream_copy.hip -- which device-to-device copy kernel shape reaches the highest HBM rate on this box (feeds csrc/peaks.hip).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/stream_copy scripts/microbench/stream_copy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(const v4f *__restrict__ s, v4f *__restrict__ d, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    v4f r[U];
#pragma unroll
    for (int u = 0; u < U; u++) r[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (NT) __builtin_nontemporal_store(r[u], d + i + u * stride);
      else d[i + u * stride] = r[u];
    }
  }
  for (; i < n4; i += stride) d[i] = s[i];
}
// contiguous chunk per block (each block streams its own range)
template <bool NT>
__global__ void __launch_bounds__(256) k_copy_chunk(const v4f *__restrict__ s, v4f *__restrict__ d, size_t n4) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x, b0 = blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
    v4f r = NT ? __builtin_nontemporal_load(s + i) : s[i];
    if (NT) __builtin_nontemporal_store(r, d + i); else d[i] = r;
  }
}
template <typename F> float best_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float best = 1e30f;
  for (int r = 0; r < 8; r++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (r >= 2 && ms < best) best = ms; }
  return best;
}
int main() {
  for (size_t mib : {256, 512, 2048}) {
    const size_t bytes = mib << 20, n4 = bytes / 16;
    v4f *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
#define RUN(name, ...) { float ms = best_ms([&] { __VA_ARGS__; }); printf("%5zu MiB blocks %6d %-18s %8.1f GB/s\n", mib, blocks, name, 2.0 * bytes / (ms * 1e-3) / 1e9); }
      RUN("grid U1", hipLaunchKernelGGL((k_copy<1, false>), dim3(blocks), dim3(256), 0, 0, s, d, n4))
      RUN("grid U4", hipLaunchKernelGGL((k_copy<4, false>), dim3(blocks), dim3(256), 0, 0, s, d, n4))
      RUN("grid U4 nt", hipLaunchKernelGGL((k_copy<4, true>), dim3(blocks), dim3(256), 0, 0, s, d, n4))
      RUN("grid U8 nt", hipLaunchKernelGGL((k_copy<8, true>), dim3(blocks), dim3(256), 0, 0, s, d, n4))
      RUN("chunk", hipLaunchKernelGGL((k_copy_chunk<false>), dim3(blocks), dim3(256), 0, 0, s, d, n4))
      RUN("chunk nt", hipLaunchKernelGGL((k_copy_chunk<true>), dim3(blocks), dim3(256), 0, 0, s, d, n4))
    }
    { float ms = best_ms([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }); printf("%5zu MiB hipMemcpy D2D %8.1f GB/s\n", mib, 2.0 * bytes / (ms * 1e-3) / 1e9); }
    hipFree(s); hipFree(d);
  }
  return 0;
}
