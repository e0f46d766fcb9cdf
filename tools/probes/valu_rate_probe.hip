// Issue cost of single VALU instructions on gfx950, measured with two waves per SIMD (the fused SparseImgAlign shape):
// every wave runs REP blocks of 64 independent copies of one instruction and stamps s_memtime around them.
// build: hipcc -O2 --offload-arch=gfx950 valu_rate_probe.hip -o /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OP>
__global__ __launch_bounds__(512) void probe(long long* out, int iters, float seed) {
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
  float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3;
  unsigned u0 = __float_as_uint(seed), u1 = u0 + 1;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %1, %1\n v_add_f32 %2, %3, %3\n v_add_f32 %0, %3, %1\n v_add_f32 %2, %1, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    if (OP == 1) { REP16(asm volatile("v_fma_f64 %0, %1, %1, %1\n v_fma_f64 %2, %3, %3, %3\n v_fma_f64 %0, %3, %3, %1\n v_fma_f64 %2, %1, %1, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 2) { REP16(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f64_f32 %1, %3\n v_cvt_f64_f32 %0, %3\n v_cvt_f64_f32 %1, %2" : "+v"(d0), "+v"(d1), "+v"(f0), "+v"(f1));) }
    if (OP == 3) { REP16(asm volatile("v_mul_f64 %0, %1, %1\n v_mul_f64 %2, %3, %3\n v_mul_f64 %0, %3, %1\n v_mul_f64 %2, %1, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 4) { REP16(asm volatile("v_add_f64 %0, %1, %1\n v_add_f64 %2, %3, %3\n v_add_f64 %0, %3, %1\n v_add_f64 %2, %1, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 5) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %2\n v_cvt_f32_ubyte1 %1, %3\n v_cvt_f32_ubyte2 %0, %3\n v_cvt_f32_ubyte3 %1, %2" : "+v"(f0), "+v"(f1), "+v"(u0), "+v"(u1));) }
    if (OP == 6) { REP16(asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3\n v_cvt_f32_f64 %0, %3\n v_cvt_f32_f64 %1, %2" : "+v"(f0), "+v"(f1), "+v"(d0), "+v"(d1));) }
    if (OP == 7) { REP16(asm volatile("v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %0, 5, %3\n v_lshlrev_b32 %2, 7, %1" : "+v"(u0), "+v"(u1), "+v"(f0), "+v"(f1));) }
    if (OP == 8) { REP16(asm volatile("v_mul_f32 %0, %1, %1\n v_mul_f32 %2, %3, %3\n v_mul_f32 %0, %3, %1\n v_mul_f32 %2, %1, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    if (OP == 9) { REP16(asm volatile("v_rcp_f64 %0, %1\n v_rcp_f64 %2, %3\n v_rcp_f64 %0, %3\n v_rcp_f64 %2, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 10) { REP16(asm volatile("v_cndmask_b32 %0, %1, %1, vcc\n v_cndmask_b32 %2, %3, %3, vcc\n v_cndmask_b32 %0, %3, %1, vcc\n v_cndmask_b32 %2, %1, %3, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : : "vcc");) }
    if (OP == 11) { REP16(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %2, %3\n v_mov_b64 %0, %3\n v_mov_b64 %2, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 12) { REP16(asm volatile("v_lshl_add_u64 %0, %1, 3, %1\n v_lshl_add_u64 %2, %3, 3, %3\n v_lshl_add_u64 %0, %3, 1, %1\n v_lshl_add_u64 %2, %1, 2, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 13) { REP16(asm volatile("v_fmac_f64 %0, %1, %1\n v_fmac_f64 %2, %3, %3\n v_fmac_f64 %0, %3, %3\n v_fmac_f64 %2, %1, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }

    if (OP == 20) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    if (OP == 21) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 22) { REP16(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 23) { REP16(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %2, %0\n v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %2, %0" : "+v"(d0), "+v"(d1), "+v"(f0), "+v"(f1));) }
    if (OP == 24) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    if (OP == 30) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    if (OP == 31) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 32) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    if (OP == 33) { REP16(asm volatile("v_cvt_f64_f32 %0, %2\n v_fma_f64 %1, %0, %0, %1\n v_cvt_f64_f32 %0, %3\n v_fma_f64 %1, %0, %0, %1" : "+v"(d0), "+v"(d1), "+v"(f0), "+v"(f1));) }
    if (OP == 34) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1\n v_fma_f64 %3, %3, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1\n v_fma_f64 %3, %3, %1, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }

    if (OP == 40) { REP16(asm volatile("v_fma_f64 v[20:21], v[0:1], v[4:5], v[8:9]\n v_fma_f64 v[22:23], v[0:1], v[4:5], v[8:9]\n v_fma_f64 v[24:25], v[0:1], v[4:5], v[8:9]\n v_fma_f64 v[26:27], v[0:1], v[4:5], v[8:9]" ::: "v0","v1","v4","v5","v8","v9","v20","v21","v22","v23","v24","v25","v26","v27");) }
    if (OP == 41) { REP16(asm volatile("v_fma_f64 v[20:21], v[0:1], v[2:3], v[6:7]\n v_fma_f64 v[22:23], v[0:1], v[2:3], v[6:7]\n v_fma_f64 v[24:25], v[0:1], v[2:3], v[6:7]\n v_fma_f64 v[26:27], v[0:1], v[2:3], v[6:7]" ::: "v0","v1","v2","v3","v6","v7","v20","v21","v22","v23","v24","v25","v26","v27");) }
    if (OP == 42) { REP16(asm volatile("v_fma_f32 v20, v0, v4, v8\n v_fma_f32 v21, v0, v4, v8\n v_fma_f32 v22, v0, v4, v8\n v_fma_f32 v23, v0, v4, v8" ::: "v0","v4","v8","v20","v21","v22","v23");) }
    if (OP == 43) { REP16(asm volatile("v_fma_f32 v20, v0, v1, v2\n v_fma_f32 v21, v0, v1, v2\n v_fma_f32 v22, v0, v1, v2\n v_fma_f32 v23, v0, v1, v2" ::: "v0","v1","v2","v20","v21","v22","v23");) }
    if (OP == 44) { REP16(asm volatile("v_mul_f64 v[20:21], v[0:1], v[4:5]\n v_mul_f64 v[22:23], v[0:1], v[4:5]\n v_add_f64 v[24:25], v[0:1], v[4:5]\n v_add_f64 v[26:27], v[0:1], v[4:5]" ::: "v0","v1","v4","v5","v20","v21","v22","v23","v24","v25","v26","v27");) }
    if (OP == 45) { REP16(asm volatile("v_mul_f64 v[20:21], v[0:1], v[2:3]\n v_mul_f64 v[22:23], v[0:1], v[2:3]\n v_add_f64 v[24:25], v[0:1], v[2:3]\n v_add_f64 v[26:27], v[0:1], v[2:3]" ::: "v0","v1","v2","v3","v20","v21","v22","v23","v24","v25","v26","v27");) }
    if (OP == 46) { REP16(asm volatile("v_mul_f32 v20, v0, v4\n v_fma_f64 v[22:23], v[0:1], v[2:3], v[6:7]\n v_add_f32 v24, v0, v5\n v_fma_f64 v[26:27], v[0:1], v[2:3], v[6:7]" ::: "v0","v1","v2","v3","v4","v5","v6","v7","v20","v22","v23","v24","v26","v27");) }
    if (OP == 47) { REP16(asm volatile("v_mul_f32 v20, v0, v4\n v_add_f32 v24, v0, v5\n v_fma_f64 v[22:23], v[0:1], v[2:3], v[6:7]\n v_fma_f64 v[26:27], v[0:1], v[2:3], v[6:7]" ::: "v0","v1","v2","v3","v4","v5","v6","v7","v20","v22","v23","v24","v26","v27");) }
    if (OP == 48) { REP16(asm volatile("v_mul_f32 v20, v0, v4\n v_readlane_b32 s20, v1, 3\n v_add_f32 v24, v0, v5\n v_readlane_b32 s21, v1, 5" ::: "v0","v1","v4","v5","v20","v24","s20","s21");) }
    if (OP == 49) { REP16(asm volatile("v_mul_f32 v20, v0, v4\n s_nop 0\n v_add_f32 v24, v0, v5\n s_nop 0" ::: "v0","v4","v5","v20","v24");) }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 + u0 + u1 == 12345.678) out[0] = 0;
}

template <int OP>
void run(const char* name, int waves_per_simd, double per_block = 64.0) {
  const int threads = 64 * 4 * waves_per_simd, blocks = 256, iters = 20000;
  long long* d;
  hipMalloc(&d, blocks * 8 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<blocks, threads>>>(d, iters, 1.5f);
  hipEventRecord(e0);
  probe<OP><<<blocks, threads>>>(d, iters, 1.5f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= h.size();
  // s_memtime counts at 100 MHz: report wall time per instruction per SIMD from the event time instead
  const double n_inst = per_block * iters;                       // per wave
  const double ns_per_inst_simd = ms * 1e6 / (n_inst * waves_per_simd);
  printf("%-18s waves/SIMD %d  %.3f ms  %.3f ns per instruction per SIMD  (memtime ticks/wave %.0f)\n", name, waves_per_simd, ms, ns_per_inst_simd, mean);
  hipFree(d);
}

int main() {
  for (int w = 2; w <= 2; ++w) {
    run<0>("v_add_f32", w); run<8>("v_mul_f32", w); run<5>("v_cvt_f32_ubyte", w); run<7>("v_lshlrev_b32", w);
    run<11>("v_mov_b64", w); run<12>("v_lshl_add_u64", w);
    run<1>("v_fma_f64", w); run<13>("v_fmac_f64", w); run<3>("v_mul_f64", w); run<4>("v_add_f64", w); run<2>("v_cvt_f64_f32", w); run<6>("v_cvt_f32_f64", w); run<9>("v_rcp_f64", w);
    run<40>("fma_f64 same banks", w); run<41>("fma_f64 spread banks", w); run<42>("fma_f32 same bank", w); run<43>("fma_f32 spread", w);
    run<44>("mul/add f64 same", w); run<45>("mul/add f64 spread", w); run<46>("f32,f64 alternating", w); run<47>("f32 f32 f64 f64", w);
    run<48>("f32 + readlane", w); run<49>("f32 + s_nop", w);
    run<20>("dep1 v_add_f32", w); run<30>("dep2 v_add_f32", w); run<32>("dep3 v_add_f32", w, 96.0); run<24>("dep1 mul/add f32", w);
    run<21>("dep1 v_fma_f64", w); run<31>("dep2 v_fma_f64", w); run<34>("dep3 v_fma_f64", w, 96.0); run<22>("dep1 v_add_f64", w); run<23>("dep1 cvt f64<->f32", w); run<33>("cvt->fma chain", w);
  }
  return 0;
}
