// microbench: throughput of scattered int atomics on MI355X (decides the binning design)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(int *cnt, int C, int n, int *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int t = hsh(i) % C;
  int r = 0;
  if (MODE == 0) atomicAdd(cnt + t, 1);                                   // agent scope, no return
  if (MODE == 1) r = atomicAdd(cnt + t, 1);                               // agent scope, returning
  if (MODE == 2) r = __hip_atomic_fetch_add(cnt + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // wg scope ret
  if (MODE == 3) __hip_atomic_fetch_add(cnt + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // wg scope no ret
  if (MODE == 4) { r = cnt[t]; }                                          // plain gather (reference)
  if (MODE == 1 || MODE == 2 || MODE == 4) out[i] = r;
}
template <int MODE>
float run(int *cnt, int C, int n, int *out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipMemset(cnt, 0, sizeof(int) * C);
  k<MODE><<<(n + 255) / 256, 256>>>(cnt, C, n, out);
  hipDeviceSynchronize();
  hipMemset(cnt, 0, sizeof(int) * C);
  hipEventRecord(a);
  for (int it = 0; it < 5; it++) k<MODE><<<(n + 255) / 256, 256>>>(cnt, C, n, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}
int main() {
  const int n = 4300000;
  int *cnt, *out;
  hipMalloc(&cnt, sizeof(int) * (1 << 22)); hipMalloc(&out, sizeof(int) * n);
  int Cs[] = {576, 4608, 36864, 294912, 1 << 22};
  for (int C : Cs) {
    printf("C=%8d  noret %.3f ms  ret %.3f ms  wg-ret %.3f ms  wg-noret %.3f ms  gather %.3f ms\n", C, run<0>(cnt, C, n, out),
           run<1>(cnt, C, n, out), run<2>(cnt, C, n, out), run<3>(cnt, C, n, out), run<4>(cnt, C, n, out));
  }
  return 0;
}
