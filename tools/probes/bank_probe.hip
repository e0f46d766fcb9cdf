// Which VGPR pairs conflict as sources of one f32 VALU instruction on gfx950 (two waves per SIMD)?
// build: hipcc -O2 --offload-arch=gfx950 bank_probe.hip -o ../../build/bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v12","v16","v17","v32","v33","v64","v20","v21","v22","v23"
#define MUL2(K) asm volatile("v_mul_f32 v20, v0, v" #K "\n v_mul_f32 v21, v0, v" #K "\n v_mul_f32 v22, v0, v" #K "\n v_mul_f32 v23, v0, v" #K ::: CLOB);
#define FMA3(A,B) asm volatile("v_fma_f32 v20, v0, v" #A ", v" #B "\n v_fma_f32 v21, v0, v" #A ", v" #B "\n v_fma_f32 v22, v0, v" #A ", v" #B "\n v_fma_f32 v23, v0, v" #A ", v" #B ::: CLOB);
#define DST(K) asm volatile("v_mul_f32 v" #K ", v0, v1\n v_mul_f32 v" #K ", v0, v1\n v_mul_f32 v" #K ", v0, v1\n v_mul_f32 v" #K ", v0, v1" ::: CLOB);
template <int OP>
__global__ __launch_bounds__(512) void probe(long long* out, int iters) {
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(MUL2(1)) } if (OP == 1) { REP16(MUL2(2)) } if (OP == 2) { REP16(MUL2(3)) } if (OP == 3) { REP16(MUL2(4)) }
    if (OP == 4) { REP16(MUL2(5)) } if (OP == 5) { REP16(MUL2(8)) } if (OP == 6) { REP16(MUL2(16)) } if (OP == 7) { REP16(MUL2(32)) }
    if (OP == 8) { REP16(MUL2(64)) } if (OP == 9) { REP16(MUL2(0)) }
    if (OP == 10) { REP16(FMA3(1, 2)) } if (OP == 11) { REP16(FMA3(4, 1)) } if (OP == 12) { REP16(FMA3(1, 5)) } if (OP == 13) { REP16(FMA3(4, 8)) }
    if (OP == 14) { REP16(FMA3(0, 0)) } if (OP == 15) { REP16(FMA3(0, 1)) }
    if (OP == 16) { REP16(DST(4)) } if (OP == 17) { REP16(DST(5)) } if (OP == 18) { REP16(DST(2)) }
    if (OP == 19) { REP16(asm volatile("v_mul_f32 v20, s4, v1\n v_mul_f32 v21, s4, v1\n v_mul_f32 v22, s4, v1\n v_mul_f32 v23, s4, v1" ::: CLOB);) }
    if (OP == 20) { REP16(asm volatile("v_sub_f32 v20, v0, v4\n v_add_f32 v21, v0, v4\n v_sub_f32 v22, v0, v4\n v_add_f32 v23, v0, v4" ::: CLOB);) }
    if (OP == 21) { REP16(asm volatile("v_cvt_f64_f32 v[20:21], v0\n v_cvt_f64_f32 v[22:23], v4\n v_cvt_f64_f32 v[20:21], v1\n v_cvt_f64_f32 v[22:23], v5" ::: CLOB);) }
    if (OP == 30) { REP16(asm volatile("v_mul_f64 v[20:21], v[0:1], s[4:5]\n v_mul_f64 v[22:23], v[0:1], s[4:5]\n v_mul_f64 v[20:21], v[2:3], s[4:5]\n v_mul_f64 v[22:23], v[2:3], s[4:5]" ::: CLOB);) }
    if (OP == 31) { REP16(asm volatile("v_mul_f64 v[20:21], v[0:1], v[4:5]\n v_mul_f64 v[22:23], v[0:1], v[4:5]\n v_mul_f64 v[20:21], v[2:3], v[4:5]\n v_mul_f64 v[22:23], v[2:3], v[4:5]" ::: CLOB);) }
    if (OP == 32) { REP16(asm volatile("v_add_f64 v[20:21], v[0:1], s[4:5]\n v_add_f64 v[22:23], v[0:1], -v[4:5]\n v_add_f64 v[20:21], v[2:3], 1.0\n v_add_f64 v[22:23], -v[2:3], 1.0" ::: CLOB);) }
    if (OP == 33) { REP16(asm volatile("v_mul_f32_e64 v20, v0, v1\n v_mul_f32_e64 v21, v0, v1\n v_sub_f32_e64 v22, v0, -v1\n v_add_f32_e64 v23, |v0|, v1" ::: CLOB);) }
    if (OP == 34) { REP16(asm volatile("v_mul_f32 v20, 0.5, v1\n v_mul_f32 v21, 2.0, v1\n v_add_f32 v22, 1.0, v1\n v_add_f32 v23, 1.0, v1" ::: CLOB);) }
    if (OP == 35) { REP16(asm volatile("v_mul_f32 v20, 0x40490fdb, v1\n v_mul_f32 v21, 0x40490fdb, v1\n v_add_f32 v22, 0x40490fdb, v1\n v_add_f32 v23, 0x40490fdb, v1" ::: CLOB);) }
    if (OP == 36) { REP16(asm volatile("v_fma_f64 v[20:21], v[0:1], v[4:5], v[8:9]\n v_fmac_f64 v[22:23], v[0:1], v[4:5]\n v_fma_f64 v[20:21], -v[2:3], v[4:5], 1.0\n v_fmac_f64 v[22:23], v[2:3], v[4:5]" ::: CLOB);) }
    if (OP == 37) { REP16(asm volatile("v_mov_b32 v20, v0\n v_mov_b32 v21, v1\n v_mov_b32 v22, s4\n v_mov_b32 v23, 0" ::: CLOB);) }
    if (OP == 38) { REP16(asm volatile("v_cndmask_b32 v20, v0, v1, vcc\n v_cndmask_b32 v21, v0, v1, vcc\n v_cndmask_b32 v22, v2, v1, vcc\n v_cndmask_b32 v23, v3, v1, vcc" ::: CLOB);) }
    if (OP == 39) { REP16(asm volatile("v_cndmask_b32_e64 v20, v0, v1, s[6:7]\n v_cndmask_b32_e64 v21, v0, v1, s[6:7]\n v_cndmask_b32_e64 v22, v2, v1, s[6:7]\n v_cndmask_b32_e64 v23, v3, v1, s[6:7]" ::: CLOB);) }
    if (OP == 40) { REP16(asm volatile("v_add_u32 v20, v0, v1\n v_and_b32 v21, v0, v1\n v_add_u32 v22, v2, v1\n v_and_b32 v23, v3, v1" ::: CLOB);) }
    if (OP == 41) { REP16(asm volatile("v_cmp_gt_f32 vcc, v0, v1\n v_cmp_lt_i32 vcc, v0, v1\n v_cmp_gt_f32 vcc, v2, v1\n v_cmp_lt_i32 vcc, v3, v1" ::: CLOB, "vcc");) }
    if (OP == 42) { REP16(asm volatile("v_floor_f32 v20, v0\n v_cvt_i32_f32 v21, v0\n v_cvt_f32_i32 v22, v2\n v_floor_f32 v23, v3" ::: CLOB);) }
    if (OP == 43) { REP16(asm volatile("v_max_f32 v20, v0, v1\n v_min_f32 v21, v0, v1\n v_max_f32 v22, v2, v1\n v_min_f32 v23, v3, v1" ::: CLOB);) }
  }
  if ((threadIdx.x & 63) == 0 && out) out[0] = 1;
}
template <int OP>
void run(const char* name) {
  const int threads = 512, blocks = 256, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<blocks, threads>>>(nullptr, 100);
  hipEventRecord(e0);
  probe<OP><<<blocks, threads>>>(nullptr, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms  %.3f ns per instruction per SIMD\n", name, ms, ms * 1e6 / (64.0 * iters * 2));
}
int main() {
  run<9>("mul v0,v0"); run<0>("mul v0,v1"); run<1>("mul v0,v2"); run<2>("mul v0,v3"); run<3>("mul v0,v4"); run<4>("mul v0,v5"); run<5>("mul v0,v8");
  run<6>("mul v0,v16"); run<7>("mul v0,v32"); run<8>("mul v0,v64");
  run<14>("fma v0,v0,v0"); run<15>("fma v0,v0,v1"); run<10>("fma v0,v1,v2"); run<11>("fma v0,v4,v1"); run<12>("fma v0,v1,v5"); run<13>("fma v0,v4,v8");
  run<16>("mul dst v4 <- v0,v1"); run<17>("mul dst v5 <- v0,v1"); run<18>("mul dst v2 <- v0,v1"); run<19>("mul s4,v1"); run<20>("sub/add v0,v4"); run<21>("cvt_f64_f32");
  run<31>("mul_f64 v,v"); run<30>("mul_f64 v,s"); run<32>("add_f64 mixed mods"); run<36>("fma/fmac f64"); run<33>("f32 e64 / modifiers"); run<34>("f32 inline const"); run<35>("f32 literal");
  run<37>("v_mov_b32"); run<38>("cndmask vcc"); run<39>("cndmask e64 sgpr"); run<40>("int add/and"); run<41>("v_cmp"); run<42>("floor/cvt i32"); run<43>("max/min f32");
  return 0;
}
