// What does s_memtime count on gfx950?  Compares it with s_memrealtime (constant 100 MHz) around f32- and f64-heavy loops.
// build: hipcc -O2 --offload-arch=gfx950 clock_probe.hip -o ../../build/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "v0","v1","v2","v3","v4","v5","v20","v21","v22","v23"
#define F32X4 asm volatile("v_mul_f32 v20, v0, v1\n v_add_f32 v21, v0, v1\n v_mul_f32 v22, v2, v1\n v_sub_f32 v23, v3, v1" ::: CLOB);
#define F64X4 asm volatile("v_mul_f64 v[20:21], v[0:1], v[4:5]\n v_add_f64 v[22:23], v[0:1], v[4:5]\n v_fma_f64 v[20:21], v[2:3], v[4:5], v[0:1]\n v_fma_f64 v[22:23], v[2:3], v[4:5], v[0:1]" ::: CLOB);
template <int MODE>
__global__ __launch_bounds__(512) void probe(long long* out, int iters) {
  const long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP16(F32X4) }
    if (MODE == 1) { REP16(F64X4) }
    if (MODE == 2) { __builtin_amdgcn_s_sleep(127); }
  }
  const long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = m1 - m0; out[1] = r1 - r0; }
}
template <int MODE>
void run(const char* name, int blocks, int iters) {
  long long* d; hipMalloc(&d, 16); long long h[2];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<MODE><<<blocks, 512>>>(d, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-22s blocks %3d  %.3f ms  memtime %lld  memrealtime %lld  -> memtime = %.3f ticks per 10 ns, kernel %.3f ms by memrealtime\n", name, blocks, ms,
         h[0], h[1], (double)h[0] / h[1], h[1] * 1e-5);
  hipFree(d);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("f32", 256, 40000); run<1>("f64", 256, 40000); run<2>("sleep", 256, 400); run<1>("f64", 16, 40000); run<0>("f32", 16, 40000);
  }
  return 0;
}
