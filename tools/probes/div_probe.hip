// Checks the claim behind div2_by (svo_sia.hip): for a denominator with |z| in [2^-900, 2^900) the shared-reciprocal
// quotients are bit-identical to the compiler's IEEE divisions x / z and y / z.  Random operands over the whole
// exponent range that the guard admits for z and a wide one for the numerators; counts mismatching bit patterns
// (quotients that are subnormal, infinite or NaN on either side are counted separately: outside every image).
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 div_probe.hip -o ../../build/div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void div2_by(double x, double y, double z, double& qx, double& qy) {
  double r = __builtin_amdgcn_rcp(z);
  double e = __builtin_fma(-z, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-z, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q0 = x * r, p0 = y * r;
  qx = __builtin_fma(__builtin_fma(-z, q0, x), r, q0);
  qy = __builtin_fma(__builtin_fma(-z, p0, y), r, p0);
}

__device__ uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ double make(uint64_t bits, int e_lo, int e_hi) {     // random sign / mantissa, biased exponent in [e_lo, e_hi]
  const uint64_t m = bits & 0x800fffffffffffffull;
  const uint64_t e = (uint64_t)(e_lo + (int)((bits >> 52) % (uint64_t)(e_hi - e_lo + 1)));
  return __longlong_as_double((long long)(m | (e << 52)));
}

__global__ void probe(unsigned long long* out, int per_thread, int mode) {
  uint64_t s = 0x1234567ull + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x632be59bd9b4e019ull;
  unsigned long long bad = 0, special = 0;
  for (int i = 0; i < per_thread; ++i) {
    double x, y, z;
    if (mode == 0) {            // the kernel's regime: metres
      z = make(splitmix(s), 1023 - 20, 1023 + 20); x = make(splitmix(s), 1023 - 30, 1023 + 20); y = make(splitmix(s), 1023 - 30, 1023 + 20);
    } else if (mode == 1) {     // everything the guard admits for z, numerators anywhere
      z = make(splitmix(s), 123, 1922); x = make(splitmix(s), 1, 2046); y = make(splitmix(s), 1, 2046);
    } else {                    // everything the guard admits for z, numerators in [2^-960, 2^960)
      z = make(splitmix(s), 123, 1922); x = make(splitmix(s), 63, 1982); y = make(splitmix(s), 63, 1982);
    }
    double qx, qy;
    div2_by(x, y, z, qx, qy);
    const double rx = x / z, ry = y / z;
    const double q[2] = {qx, qy}, r[2] = {rx, ry};
    for (int k = 0; k < 2; ++k) {
      const long long a = __double_as_longlong(q[k]), b = __double_as_longlong(r[k]);
      const int er = (int)((b >> 52) & 0x7ff);
      if (er == 0 || er == 2047) { if (a != b) ++special; }
      else if (a != b) ++bad;
    }
  }
  atomicAdd(&out[0], bad);
  atomicAdd(&out[1], special);
}

int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(d, 0, 16);
    probe<<<1024, 256>>>(d, 2000, mode);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mode %d: %llu quotient pairs, normal quotients with different bits: %llu, subnormal/inf/NaN quotients that differ: %llu\n", mode,
           1024ull * 256 * 2000 * 2, h[0], h[1]);
  }
  return 0;
}
