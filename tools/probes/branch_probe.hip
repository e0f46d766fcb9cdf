// Cost of scalar branches inside a VALU stream on gfx950, two waves per SIMD.
// build: hipcc -O2 --offload-arch=gfx950 branch_probe.hip -o ../../build/branch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "v0","v1","v2","v3","v20","v21","v22","v23","scc"
// 8 VALU + one branch construct per group; 16 groups per iteration
#define VALU8 "v_mul_f32 v20, v0, v1\n v_add_f32 v21, v0, v1\n v_mul_f32 v22, v2, v1\n v_sub_f32 v23, v3, v1\n v_mul_f32 v20, v0, v1\n v_add_f32 v21, v0, v1\n v_mul_f32 v22, v2, v1\n v_sub_f32 v23, v3, v1\n"
template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, int zero) {
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP16(asm volatile(VALU8 ::: CLOB);) }                                                                  // no branch
    if (MODE == 1) { REP16(asm volatile(VALU8 "s_cmp_eq_u32 %0, 1\n s_cbranch_scc1 1f\n s_nop 0\n1:\n" :: "s"(zero) : CLOB);) }   // not taken (falls into the nop)
    if (MODE == 2) { REP16(asm volatile(VALU8 "s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n" :: "s"(zero) : CLOB);) }   // taken, skips one instruction
    if (MODE == 3) { REP16(asm volatile(VALU8 "s_cmp_eq_u32 %0, 0\n s_nop 0\n s_nop 0\n" :: "s"(zero) : CLOB);) }                // same count, no branch
    if (MODE == 4) { REP16(asm volatile(VALU8 "s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n" VALU8 VALU8 VALU8 VALU8 "1:\n" :: "s"(zero) : CLOB);) }   // taken, skips 32 instructions
    if (MODE == 5) { REP16(asm volatile(VALU8 "s_branch 1f\n s_nop 0\n1:\n" ::: CLOB);) }                                     // unconditional
  }
}
template <int MODE>
void run(const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 512>>>(100, 0);
  hipEventRecord(e0);
  probe<MODE><<<256, 512>>>(iters, 0);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms  %.2f ns per group of 8 VALU (+ branch construct) per wave\n", name, ms, ms * 1e6 / (16.0 * iters));
}
int main() {
  run<0>("8 VALU"); run<3>("8 VALU + s_cmp + 2 s_nop"); run<1>("8 VALU + s_cmp + branch not taken + nop"); run<2>("8 VALU + s_cmp + branch TAKEN over 1");
  run<4>("8 VALU + s_cmp + branch TAKEN over 32"); run<5>("8 VALU + s_branch over 1"); run<0>("8 VALU");
  return 0;
}
