// Exhaustive check (all 2^32 binary32 operands) of cheap reciprocal sequences against the IEEE division 1.0f / x that the
// EXACT kernels are pinned to (GLSL `1.0 / x` correctly rounded, oracle/ref_glsl_shim.hpp).  Prints, per candidate, the number of
// operands whose result differs bitwise (NaN == NaN) and the exponent range of the offenders.
//   A: y = rcp(x); e = fma(-x, y, 1); y + y*e                       (one Newton step from v_rcp_f32)
//   B: A, then once more
//   C: A, then a residual correction r = fma(-x, y1, 1); fma(r, y1, y1)  (= B written on y1)
//   D: A with the operand's special classes (0, inf, NaN, and |x| outside [2^-126, 2^126]) passed through v_rcp_f32 unchanged
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float cand(int which, float x) {
  const float y = __builtin_amdgcn_rcpf(x);
  const float e = fmaf(-x, y, 1.0f);
  const float y1 = fmaf(y, e, y);
  if (which == 0) return y1;
  if (which == 1 || which == 2) { const float e2 = fmaf(-x, y1, 1.0f); return fmaf(y1, e2, y1); }
  // D: the Newton step, then v_div_fixup_f32 for the operand's special classes (0, inf, NaN, denormal)
  return __builtin_amdgcn_div_fixupf(y1, x, 1.0f);
}

__device__ unsigned long long g_hist[4][256];
__global__ void sweep(unsigned long long* bad, unsigned* lo, unsigned* hi) {
  const unsigned long long n = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = __uint_as_float((uint32_t)i);
    const float want = 1.0f / x;
    for (int c = 0; c < 4; ++c) {
      const float got = cand(c, x);
      const bool same = __float_as_uint(got) == __float_as_uint(want) || (got != got && want != want);
      if (!same) {
        atomicAdd(&bad[c], 1ull);
        const unsigned ex = ((uint32_t)i >> 23) & 0xffu;
        atomicAdd(&g_hist[c][ex], 1ull);
        atomicMin(&lo[c], ex);
        atomicMax(&hi[c], ex);
      }
    }
  }
}

int main() {
  unsigned long long* bad; unsigned *lo, *hi;
  CK(hipMalloc(&bad, 4 * 8)); CK(hipMalloc(&lo, 16)); CK(hipMalloc(&hi, 16));
  CK(hipMemset(bad, 0, 32)); CK(hipMemset(lo, 0xff, 16)); CK(hipMemset(hi, 0, 16));
  hipLaunchKernelGGL(sweep, dim3(256 * 32), dim3(256), 0, 0, bad, lo, hi);
  CK(hipDeviceSynchronize());
  unsigned long long hb[4]; unsigned hl[4], hh[4];
  CK(hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, lo, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hh, hi, 16, hipMemcpyDeviceToHost));
  const char* name[4] = {"A one Newton step", "B two Newton steps", "C (= B)", "D one step, special classes passed through"};
  for (int c = 0; c < 4; ++c)
    printf("%-44s mismatches %llu of 4294967296%s", name[c], hb[c], hb[c] ? "" : "\n"), hb[c] ? printf("  biased exponents %u .. %u\n", hl[c], hh[c]) : 0;
  static unsigned long long hist[4][256];
  CK(hipMemcpyFromSymbol(hist, HIP_SYMBOL(g_hist), sizeof hist));
  for (int c = 0; c < 4; c += 3) {
    printf("candidate %d, mismatches per biased exponent:", c);
    for (int e = 0; e < 256; ++e) if (hist[c][e]) printf(" %d:%llu", e, hist[c][e]);
    printf("\n");
  }
  return 0;
}
