// Two waves per SIMD: how does the instruction mix of the partner wave change a wave's pace on gfx950?
// build: hipcc -O2 --offload-arch=gfx950 mix_probe.hip -o ../../build/mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "v0","v1","v2","v3","v4","v5","v20","v21","v22","v23"
#define F32X4 asm volatile("v_mul_f32 v20, v0, v1\n v_add_f32 v21, v0, v1\n v_mul_f32 v22, v2, v1\n v_sub_f32 v23, v3, v1" ::: CLOB);
#define F64X4 asm volatile("v_mul_f64 v[20:21], v[0:1], v[4:5]\n v_add_f64 v[22:23], v[0:1], v[4:5]\n v_fma_f64 v[20:21], v[2:3], v[4:5], v[0:1]\n v_cvt_f64_f32 v[22:23], v2" ::: CLOB);
// MODE 0: every wave: 64 f32 then 64 f64 per iteration (lock-step phases)
// MODE 1: waves 0..3 (the older wave of each SIMD) only f32 (128 per iteration), waves 4..7 only f64 (128)
// MODE 2: every wave: f32 block / f64 block, but waves 4..7 start with the f64 block (out of phase)
// MODE 3: every wave: alternating groups of 4 f32 / 4 f64
template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP16(F32X4) REP16(F64X4) }
    if (MODE == 1) { if (wave < 4) { REP16(F32X4) REP16(F32X4) } else { REP16(F64X4) REP16(F64X4) } }
    if (MODE == 2) { if (wave < 4) { REP16(F32X4) REP16(F64X4) } else { REP16(F64X4) REP16(F32X4) } }
    if (MODE == 3) { REP16(F32X4 F64X4) }
  }
}
template <int MODE>
void run(const char* name) {
  const int iters = 10000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 512>>>(100);
  hipEventRecord(e0);
  probe<MODE><<<256, 512>>>(iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms  %.3f ns per instruction per SIMD\n", name, ms, ms * 1e6 / (128.0 * iters * 2));
}
int main() {
  run<0>("lock-step: f32 block, f64 block"); run<2>("out of phase blocks"); run<1>("one wave f32 only, partner f64 only"); run<3>("groups of 4 alternating in every wave");
  run<0>("lock-step: f32 block, f64 block"); run<2>("out of phase blocks");
  return 0;
}
