// microbench: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the composite kernels use.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                               \
  __global__ void __launch_bounds__(256) name(float *o, int iters) {                     \
    float a = threadIdx.x * 1e-3f, b = a + 1.f, c = a + 2.f, d = a + 3.f, e = a + 4.f, f = a + 5.f, g = a + 6.f, h = a + 7.f; \
    float2 A = {a, b}, B = {c, d}, C2 = {e, f}, D2 = {g, h};                              \
    for (int i = 0; i < iters; i++) { REP16(body) }                                       \
    o[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + A.x + A.y + B.x + B.y + C2.x + C2.y + D2.x + D2.y; \
  }
// 8 independent chains, 8 instrs per body -> 128 instrs per loop iteration
KERNEL(k_fma, asm volatile("v_fma_f32 %0,%0,%0,%0\n v_fma_f32 %1,%1,%1,%1\n v_fma_f32 %2,%2,%2,%2\n v_fma_f32 %3,%3,%3,%3\n v_fma_f32 %4,%4,%4,%4\n v_fma_f32 %5,%5,%5,%5\n v_fma_f32 %6,%6,%6,%6\n v_fma_f32 %7,%7,%7,%7" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0,%0,%0\n v_mul_f32 %1,%1,%1\n v_mul_f32 %2,%2,%2\n v_mul_f32 %3,%3,%3\n v_mul_f32 %4,%4,%4\n v_mul_f32 %5,%5,%5\n v_mul_f32 %6,%6,%6\n v_mul_f32 %7,%7,%7" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_pkfma, asm volatile("v_pk_fma_f32 %0,%0,%0,%0\n v_pk_fma_f32 %1,%1,%1,%1\n v_pk_fma_f32 %2,%2,%2,%2\n v_pk_fma_f32 %3,%3,%3,%3\n v_pk_fma_f32 %0,%0,%0,%0\n v_pk_fma_f32 %1,%1,%1,%1\n v_pk_fma_f32 %2,%2,%2,%2\n v_pk_fma_f32 %3,%3,%3,%3" : "+v"(A),"+v"(B),"+v"(C2),"+v"(D2));)
KERNEL(k_exp, asm volatile("v_exp_f32 %0,%0\n v_exp_f32 %1,%1\n v_exp_f32 %2,%2\n v_exp_f32 %3,%3\n v_exp_f32 %4,%4\n v_exp_f32 %5,%5\n v_exp_f32 %6,%6\n v_exp_f32 %7,%7" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0,%0\n v_rcp_f32 %1,%1\n v_rcp_f32 %2,%2\n v_rcp_f32 %3,%3\n v_rcp_f32 %4,%4\n v_rcp_f32 %5,%5\n v_rcp_f32 %6,%6\n v_rcp_f32 %7,%7" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_dpp, asm volatile("v_add_f32_dpp %0,%0,%0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1,%1,%1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %2,%2,%2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3,%3,%3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %4,%4,%4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %5,%5,%5 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %6,%6,%6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %7,%7,%7 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_cnd, asm volatile("v_cndmask_b32 %0,%0,%1,vcc\n v_cndmask_b32 %1,%1,%2,vcc\n v_cndmask_b32 %2,%2,%3,vcc\n v_cndmask_b32 %3,%3,%4,vcc\n v_cndmask_b32 %4,%4,%5,vcc\n v_cndmask_b32 %5,%5,%6,vcc\n v_cndmask_b32 %6,%6,%7,vcc\n v_cndmask_b32 %7,%7,%0,vcc" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");)
KERNEL(k_cmp, asm volatile("v_cmp_le_f32 vcc,%0,%1\n v_cmp_le_f32 vcc,%1,%2\n v_cmp_le_f32 vcc,%2,%3\n v_cmp_le_f32 vcc,%3,%4\n v_cmp_le_f32 vcc,%4,%5\n v_cmp_le_f32 vcc,%5,%6\n v_cmp_le_f32 vcc,%6,%7\n v_cmp_le_f32 vcc,%7,%0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");)
KERNEL(k_swap, asm volatile("v_permlane32_swap_b32 %0,%1\n v_permlane32_swap_b32 %2,%3\n v_permlane32_swap_b32 %4,%5\n v_permlane32_swap_b32 %6,%7\n v_permlane16_swap_b32 %0,%1\n v_permlane16_swap_b32 %2,%3\n v_permlane16_swap_b32 %4,%5\n v_permlane16_swap_b32 %6,%7" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_pkmul, asm volatile("v_pk_mul_f32 %0,%0,%0\n v_pk_mul_f32 %1,%1,%1\n v_pk_mul_f32 %2,%2,%2\n v_pk_mul_f32 %3,%3,%3\n v_pk_add_f32 %0,%0,%0\n v_pk_add_f32 %1,%1,%1\n v_pk_add_f32 %2,%2,%2\n v_pk_add_f32 %3,%3,%3" : "+v"(A),"+v"(B),"+v"(C2),"+v"(D2));)

KERNEL(k_cnd2, asm volatile("v_cndmask_b32 %0,%0,%0,vcc\n v_cndmask_b32 %1,%1,%1,vcc\n v_cndmask_b32 %2,%2,%2,vcc\n v_cndmask_b32 %3,%3,%3,vcc\n v_cndmask_b32 %4,%4,%4,vcc\n v_cndmask_b32 %5,%5,%5,vcc\n v_cndmask_b32 %6,%6,%6,vcc\n v_cndmask_b32 %7,%7,%7,vcc" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");)
KERNEL(k_cnd3, asm volatile("v_cndmask_b32_e64 %0,%0,%1,s[20:21]\n v_cndmask_b32_e64 %1,%1,%2,s[20:21]\n v_cndmask_b32_e64 %2,%2,%3,s[20:21]\n v_cndmask_b32_e64 %3,%3,%4,s[20:21]\n v_cndmask_b32_e64 %4,%4,%5,s[20:21]\n v_cndmask_b32_e64 %5,%5,%6,s[20:21]\n v_cndmask_b32_e64 %6,%6,%7,s[20:21]\n v_cndmask_b32_e64 %7,%7,%0,s[20:21]" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "s20","s21");)
KERNEL(k_cmp64, asm volatile("v_cmp_le_f32_e64 s[20:21],%0,%1\n v_cmp_le_f32_e64 s[22:23],%1,%2\n v_cmp_le_f32_e64 s[24:25],%2,%3\n v_cmp_le_f32_e64 s[26:27],%3,%4\n v_cmp_le_f32_e64 s[20:21],%4,%5\n v_cmp_le_f32_e64 s[22:23],%5,%6\n v_cmp_le_f32_e64 s[24:25],%6,%7\n v_cmp_le_f32_e64 s[26:27],%7,%0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "s20","s21","s22","s23","s24","s25","s26","s27");)
KERNEL(k_fmaclamp, asm volatile("v_fma_f32 %0,%0,%1,%2 clamp\n v_fma_f32 %1,%1,%2,%3 clamp\n v_fma_f32 %2,%2,%3,%4 clamp\n v_fma_f32 %3,%3,%4,%5 clamp\n v_fma_f32 %4,%4,%5,%6 clamp\n v_fma_f32 %5,%5,%6,%7 clamp\n v_fma_f32 %6,%6,%7,%0 clamp\n v_fma_f32 %7,%7,%0,%1 clamp" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_minmax, asm volatile("v_min_f32 %0,%0,%1\n v_max_f32 %1,%1,%2\n v_min_f32 %2,%2,%3\n v_max_f32 %3,%3,%4\n v_min_f32 %4,%4,%5\n v_max_f32 %5,%5,%6\n v_min_f32 %6,%6,%7\n v_max_f32 %7,%7,%0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
KERNEL(k_and, asm volatile("v_and_b32 %0,%0,%1\n v_and_b32 %1,%1,%2\n v_and_b32 %2,%2,%3\n v_and_b32 %3,%3,%4\n v_and_b32 %4,%4,%5\n v_and_b32 %5,%5,%6\n v_and_b32 %6,%6,%7\n v_and_b32 %7,%7,%0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h));)
template <typename K> void run(const char *name, K kern, float *o, int waves_per_simd) {
  const int iters = 2000, blocks = 256 * waves_per_simd;  // 4 waves per block -> waves_per_simd per SIMD
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  kern<<<blocks, 256>>>(o, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); kern<<<blocks, 256>>>(o, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)iters * 128.0 * waves_per_simd;
  printf("%-8s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (x2.4 GHz = %.2f cyc)\n", name, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
  float *o; (void)hipMalloc(&o, 4 * 256 * 256 * 8);
  for (int w : {4}) {
    run("cnd_indep", k_cnd2, o, w); run("cnd_e64", k_cnd3, o, w); run("cmp_e64", k_cmp64, o, w); run("fma_clamp", k_fmaclamp, o, w); run("minmax", k_minmax, o, w); run("and", k_and, o, w);
    run("fma", k_fma, o, w); run("mul", k_mul, o, w); run("pk_fma", k_pkfma, o, w); run("pk_mul+add", k_pkmul, o, w);
    run("exp", k_exp, o, w); run("rcp", k_rcp, o, w); run("add_dpp", k_dpp, o, w); run("cndmask", k_cnd, o, w);
    run("cmp", k_cmp, o, w); run("plswap", k_swap, o, w);
  }
  return 0;
}
