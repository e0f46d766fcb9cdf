// svo_ordered_sum.h -- the f32 sum of a sequence of non-negative values EXACTLY as a scalar loop rounds it
//     float s = 0; for (k = 0; k < n; ++k) s += x[k];
// (the reference's chi2 += res*res*weight, S/sparse_img_align.cpp:266, and the sums of its scale estimators,
// S/robust_cost.cpp:53-60,83-85), computed by a 1024-thread workgroup instead of one lane's chain of n dependent additions.
//
// Why a sequential f32 sum of non-negative terms parallelises after all.  While the running sum s stays inside one
// binade [2^e, 2^(e+1)) its unit in the last place u = 2^(e-23) is fixed and s = m*u with an integer m in [2^23, 2^24).
// A term t = q*u + r (0 <= r < u) then adds an INTEGER number of units: fl(s + t) = (m + q + c)*u with c = 0 for r < u/2,
// c = 1 for r > u/2 and, on a tie, whatever makes m + q + c even.  So within a binade the only thing an addition needs to
// know about its predecessors is the PARITY of m: every element is a function {0,1} -> (units added, new parity), and such
// functions compose associatively -- a prefix scan.  The scan is wrong from the first addition whose result reaches
// 2^(e+1) (there the unit doubles): a second pass finds that element exactly (the running m is exact up to it), the one
// addition is made by the hardware adder, and the scan restarts from the next element in the new binade.  The sum never
// decreases, so there are at most ~280 such restarts however long the sequence (one per binade of the f32 range), and in
// practice log2(n) of them, nearly all at the very beginning where s is still small against the terms -- the first
// OS_SERIAL_HEAD elements are therefore simply added by one lane.
// Subnormal sums share the unit 2^-149 with the first normal binade.  Any window of values that holds a negative, infinite
// or NaN term is added by one lane in order (the scalar loop's own semantics, whatever they yield).
#pragma once
#include <climits>

#include "svo_device_math.h"

namespace svo_dev {

constexpr int OS_THREADS = 1024;               // the workgroup: 16 waves, four per SIMD (a lone wave issues one instruction every ~5 cycles)
constexpr int OS_WAVES = OS_THREADS / 64;
constexpr int OS_PER_THREAD = 4;
constexpr int OS_WINDOW = OS_THREADS * OS_PER_THREAD;
constexpr int OS_SERIAL_HEAD = 128;            // elements one lane adds before the scans start (the sum is still small there)
constexpr int OS_SERIAL_MAX = 4096;            // sequences up to this length are added by one lane altogether (cheaper)
constexpr unsigned OS_LIMIT = 1u << 24;        // m stays below: the binade's end in units of u
constexpr unsigned OS_SAT = 1u << 26;          // unit counts saturate here (anything >= OS_LIMIT only says "past the end")

struct OsShared {
  __attribute__((aligned(16))) float stage[OS_WINDOW];
  unsigned wave_fn[OS_WAVES][2];
  unsigned m_before[OS_THREADS];
  int cross[2];
  int invalid;
  float S;
};

// An element (or a run of elements) as a function of the incoming parity p: units added (saturating) and outgoing parity,
// one word per p: units in bits 0..26, parity in bit 31.
struct OsFn {
  unsigned f0, f1;
};
constexpr unsigned OS_PAR = 0x80000000u;

SVO_DEV unsigned os_sat_add(unsigned a, unsigned b) { const unsigned s = a + b; return s < OS_SAT ? s : OS_SAT; }

// first f, then g
SVO_DEV OsFn os_compose(const OsFn& f, const OsFn& g) {
  const unsigned g0 = (f.f0 & OS_PAR) ? g.f1 : g.f0, g1 = (f.f1 & OS_PAR) ? g.f1 : g.f0;
  OsFn h;
  h.f0 = os_sat_add(f.f0 & ~OS_PAR, g0 & ~OS_PAR) | (g0 & OS_PAR);
  h.f1 = os_sat_add(f.f1 & ~OS_PAR, g1 & ~OS_PAR) | (g1 & OS_PAR);
  return h;
}

// A word from another lane by DPP; a lane without a source keeps `keep` (the identity of the scan).
template <int CTRL, int ROW_MASK>
SVO_DEV unsigned os_dpp(unsigned keep, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)keep, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
SVO_DEV OsFn os_dpp(const OsFn& v) {
  OsFn p;
  p.f0 = os_dpp<CTRL, ROW_MASK>(0u, v.f0);          // identity: (0 units, parity kept)
  p.f1 = os_dpp<CTRL, ROW_MASK>(OS_PAR, v.f1);
  return p;
}

// inclusive scan of the composition over the lanes of a wave, in lane order: shifts inside the rows of 16 lanes, then the
// last lane of a row to the rows above it (DPP: no trip through the LDS crossbar, a handful of cycles per step)
SVO_DEV OsFn os_wave_scan(OsFn v) {
  v = os_compose(os_dpp<0x111, 0xf>(v), v);          // row_shr:1
  v = os_compose(os_dpp<0x112, 0xf>(v), v);          // row_shr:2
  v = os_compose(os_dpp<0x114, 0xf>(v), v);          // row_shr:4
  v = os_compose(os_dpp<0x118, 0xf>(v), v);          // row_shr:8
  v = os_compose(os_dpp<0x142, 0xa>(v), v);          // row_bcast:15 into rows 1 and 3
  v = os_compose(os_dpp<0x143, 0xc>(v), v);          // row_bcast:31 into rows 2 and 3
  return v;
}

// s >= 0 finite: s = m * 2^ue, the unit of its binade (2^-149 for subnormals and for [2^-126, 2^-125))
SVO_DEV void os_unit(float s, int* ue, unsigned* m) {
  const unsigned bits = (unsigned)__float_as_int(s) & 0x7fffffffu;
  const unsigned ef = bits >> 23, frac = bits & 0x7fffffu;
  if (ef <= 1) { *ue = -149; *m = ef ? (frac | 0x800000u) : frac; }
  else { *ue = (int)ef - 150; *m = frac | 0x800000u; }
}

// m * 2^ue for 0 <= m <= 2^24, built from its bits (exact)
SVO_DEV float os_make(unsigned m, int ue) {
  if (m < 0x800000u) return __int_as_float((int)m);                       // subnormal (ue is -149 then) or zero
  if (m == OS_LIMIT) return __int_as_float((int)((unsigned)(ue + 151) << 23));
  return __int_as_float((int)(((unsigned)(ue + 150) << 23) | (m & 0x7fffffu)));
}

// x >= 0 finite = M * 2^ex with a 24-bit M
SVO_DEV void os_split(float x, unsigned* M, int* ex) {
  const unsigned bits = (unsigned)__float_as_int(x) & 0x7fffffffu;
  const unsigned ef = bits >> 23;
  *M = bits & 0x7fffffu;
  *ex = -149;
  if (ef) { *M |= 0x800000u; *ex = (int)ef - 150; }
}

// ... against the unit 2^ue: whole units q (saturating) and the rounding class of the rest
// (0: below half a unit, 1: above, 2: exactly half)
SVO_DEV void os_decode(unsigned M, int ex, int ue, unsigned* q, unsigned* kind) {
  const int shift = ex - ue;
  *q = 0; *kind = 0;
  if (shift >= 0) {
    *q = shift > 3 ? (M ? OS_SAT : 0u) : (M << shift < OS_SAT ? M << shift : OS_SAT);
  } else if (shift > -26) {                                                // (further down: below a quarter of a unit, adds nothing)
    const int sh = -shift;
    *q = sh >= 24 ? 0u : M >> sh;
    const unsigned r = M & ((1u << sh) - 1u), half = 1u << (sh - 1);
    *kind = r < half ? 0u : (r > half ? 1u : 2u);
  }
}

// The sum of vals[0 .. N) in index order, every thread of the OS_THREADS-thread workgroup calling (all of them get the result).
// ok16 (optional): one flag per 16 values; a group whose flag is 0 is skipped (adds +0, which changes nothing).
SVO_DEV float os_block_sum(const float* __restrict__ vals, const uint8_t* __restrict__ ok16, int N, OsShared& sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float inf = __int_as_float(0x7f800000);
  float S = 0.0f;
  unsigned phase = 0;
  // a window's values are asked for while the window before it is being added
  auto fetch = [&](int base, float* x) {
    const int k0 = base + tid * OS_PER_THREAD;
    const bool live = k0 < N && (!ok16 || ok16[k0 >> 4]);            // (a thread's OS_PER_THREAD elements lie inside one group of 16)
#pragma unroll
    for (int j = 0; j < OS_PER_THREAD; ++j) x[j] = (live && k0 + j < N) ? vals[k0 + j] : 0.0f;
  };
  float xn[OS_PER_THREAD];
  fetch(0, xn);
  for (int base = 0; base < N; base += OS_WINDOW) {
    unsigned M[OS_PER_THREAD];
    int ex[OS_PER_THREAD];
    bool bad = false;
    float xc[OS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < OS_PER_THREAD; ++j) xc[j] = xn[j];
    if (base + OS_WINDOW < N) fetch(base + OS_WINDOW, xn);
#pragma unroll
    for (int j = 0; j < OS_PER_THREAD; ++j) {
      const float x = xc[j];
      bad = bad || x < 0.0f || !(x < inf);                                  // negative, infinite or NaN
      sh.stage[tid * OS_PER_THREAD + j] = x;
      os_split(x, &M[j], &ex[j]);
    }
    if (tid == 0) sh.invalid = 0;
    __syncthreads();
    if (bad) sh.invalid = 1;
    __syncthreads();
    // a window with such a term, a sum that is no longer a finite non-negative number, a short sequence: one lane, in order
    const bool serial_all = sh.invalid != 0 || !(S >= 0.0f && S < inf) || N <= OS_SERIAL_MAX;
    int lo = 0;                                   // my first element not added yet
    if (serial_all || base == 0) {
      const int left = N - base < OS_WINDOW ? N - base : OS_WINDOW;
      const int cnt = serial_all ? left : (left < OS_SERIAL_HEAD ? left : OS_SERIAL_HEAD);
      if (tid == 0) {
        // 16 values per step, the next 16 on their way out of LDS while this step's chain of additions runs (a lone lane
        // issues an instruction every ~5 cycles: the chain is as long as its instruction count)
        float a = S;
        const float4* q = reinterpret_cast<const float4*>(sh.stage);
        const int n16 = cnt / 16;
        float4 n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];
        for (int i = 0; i < n16; ++i) {
          const float4 v0 = n0, v1 = n1, v2 = n2, v3 = n3;
          const int nx = i + 1 < n16 ? i + 1 : i;
          n0 = q[4 * nx]; n1 = q[4 * nx + 1]; n2 = q[4 * nx + 2]; n3 = q[4 * nx + 3];
          a += v0.x; a += v0.y; a += v0.z; a += v0.w;
          a += v1.x; a += v1.y; a += v1.z; a += v1.w;
          a += v2.x; a += v2.y; a += v2.z; a += v2.w;
          a += v3.x; a += v3.y; a += v3.z; a += v3.w;
        }
        for (int k = 16 * n16; k < cnt; ++k) a += sh.stage[k];
        sh.S = a;
      }
      __syncthreads();
      S = sh.S;
      const int mine = cnt - tid * OS_PER_THREAD;
      lo = mine <= 0 ? 0 : (mine < OS_PER_THREAD ? mine : OS_PER_THREAD);
      if (serial_all) continue;                   // (the next window's first barrier separates this read of sh.S from the next write)
    }
    // ---- one pass of this loop per binade the sum visits inside the window
    for (;;) {
      if (!(S < inf)) break;                      // overflowed: inf + (finite, >= 0) stays inf
      ++phase;
      int ue;
      unsigned m0;
      os_unit(S, &ue, &m0);
      // this thread's elements as one function of the incoming parity
      unsigned q[OS_PER_THREAD], kinds = 0;
      unsigned i0 = 0, i1 = 0, p0 = 0, p1 = 1;
#pragma unroll
      for (int j = 0; j < OS_PER_THREAD; ++j) {
        unsigned kd;
        os_decode(M[j], ex[j], ue, &q[j], &kd);
        if (j < lo) { q[j] = 0; kd = 0; }
        kinds |= kd << (2 * j);
        const unsigned d0 = q[j] + (kd == 2 ? ((p0 + q[j]) & 1u) : kd), d1 = q[j] + (kd == 2 ? ((p1 + q[j]) & 1u) : kd);
        i0 = os_sat_add(i0, d0); p0 = (p0 + d0) & 1u;
        i1 = os_sat_add(i1, d1); p1 = (p1 + d1) & 1u;
      }
      // inclusive scan over the wave, the waves' totals through LDS
      const OsFn incl = os_wave_scan(OsFn{i0 | (p0 << 31), i1 | (p1 << 31)});
      if (lane == 63) { sh.wave_fn[wave][0] = incl.f0; sh.wave_fn[wave][1] = incl.f1; }
      int* cross = &sh.cross[phase & 1u];
      if (tid == 0) *cross = INT_MAX;
      __syncthreads();
      // every wave scans the waves' totals itself (lanes 0..OS_WAVES-1), takes what lies before it and the grand total
      OsFn wt = {0u, OS_PAR};
      if (lane < OS_WAVES) { wt.f0 = sh.wave_fn[lane][0]; wt.f1 = sh.wave_fn[lane][1]; }
      wt = os_wave_scan(wt);                        // (lanes past OS_WAVES - 1 hold the identity: they compose to the grand total)
      OsFn before = {0u, OS_PAR}, total;
      total.f0 = (unsigned)__builtin_amdgcn_readlane((int)wt.f0, OS_WAVES - 1); total.f1 = (unsigned)__builtin_amdgcn_readlane((int)wt.f1, OS_WAVES - 1);
      if (wave > 0) {                               // wave-uniform
        before.f0 = (unsigned)__builtin_amdgcn_readlane((int)wt.f0, wave - 1);
        before.f1 = (unsigned)__builtin_amdgcn_readlane((int)wt.f1, wave - 1);
      }
      {
        // what the lanes below mine add: the inclusive scan one lane down (wave_shr:1; lane 0 keeps the identity)
        const OsFn p = os_dpp<0x138, 0xf>(incl);
        before = os_compose(before, p);
      }
      const unsigned p_start = m0 & 1u;
      const unsigned bf = p_start ? before.f1 : before.f0;
      unsigned m = os_sat_add(m0, bf & ~OS_PAR), par = bf >> 31;
      const unsigned m_in = m;
      // walk my elements with the true running count: the first addition that leaves the binade
      int cross_j = -1;
      unsigned m_before = 0;
#pragma unroll
      for (int j = 0; j < OS_PER_THREAD; ++j) {
        const unsigned kd = (kinds >> (2 * j)) & 3u;
        const unsigned d = q[j] + (kd == 2 ? ((par + q[j]) & 1u) : kd);
        if (j >= lo && cross_j < 0) {
          if (m + d >= OS_LIMIT) { cross_j = j; m_before = m; }
          else { m += d; par = (par + d) & 1u; }
        }
      }
      // (every thread behind the crossing one sees a count past the end too: only the one whose own incoming count is
      // still inside the binade is the first -- one atomic per phase, not a thousand on one address)
      if (cross_j >= 0 && m_in < OS_LIMIT) { sh.m_before[tid] = m_before; atomicMin(cross, tid * OS_PER_THREAD + cross_j); }
      __syncthreads();
      const int kx = *cross;
      if (kx == INT_MAX) {                        // the window ends inside this binade
        S = os_make(m0 + ((p_start ? total.f1 : total.f0) & ~OS_PAR), ue);
        break;
      }
      const int owner = kx / OS_PER_THREAD, jx = kx % OS_PER_THREAD;
      S = os_make(sh.m_before[owner], ue) + sh.stage[kx];                   // the one addition the adder makes
      if (tid < owner) lo = OS_PER_THREAD;
      else if (tid == owner) lo = jx + 1;
    }
  }
  return S;
}

}  // namespace svo_dev
