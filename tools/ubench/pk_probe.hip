// Does packing two pixels per lane into v_pk_*_f32 pay at the power cap?  Same tap-weight arithmetic for two
// independent pixels per lane: scalar (2 x v_fma_f32 ...) vs packed (v_pk_fma_f32 ...).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2048;

template <class T> __device__ __forceinline__ T fma_(T a, T b, T c);
template <> __device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ f2 fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <class T>
__device__ __forceinline__ void taps(T q00, T ox, T s, T b, T k1, T k2, T k3, T cr, T cg, T cb, T& aR, T& aG, T& aB, T& aW) {
#pragma unroll
  for (int t = 0; t < 12; ++t) {
    const T one = (T)(1.0f), m1 = (T)(-1.0f);
    T u = fma_(ox, fma_(q00, ox, s), b);
    T base = fma_(fma_(k2, u, k1), u, one);
    T wa = fma_(k3, u, m1);
    T w = base * (wa * wa);
    aR = fma_(cr, w, aR); aG = fma_(cg, w, aG); aB = fma_(cb, w, aB); aW = aW + w;
    ox = ox + one;  // next tap
  }
}

__global__ void __launch_bounds__(256) k_scalar(float* out, float seed) {
  float aR0 = 0, aG0 = 0, aB0 = 0, aW0 = 0, aR1 = 0, aG1 = 0, aB1 = 0, aW1 = 0;
  float q = seed * 1e-3f, s = seed * 2e-3f, b = seed * 3e-3f;
  for (int it = 0; it < ITER; ++it) {
    taps<float>(q, -1.25f, s, b, -0.3f, 0.02f, 0.4f, seed, seed + 1, seed + 2, aR0, aG0, aB0, aW0);
    taps<float>(q + 1e-3f, -1.75f, s, b, -0.3f, 0.02f, 0.4f, seed, seed + 1, seed + 2, aR1, aG1, aB1, aW1);
    q += 1e-6f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = aR0 + aG0 + aB0 + aW0 + aR1 + aG1 + aB1 + aW1;
}
__global__ void __launch_bounds__(256) k_packed(float* out, float seed) {
  f2 aR = {0, 0}, aG = {0, 0}, aB = {0, 0}, aW = {0, 0};
  f2 q = {seed * 1e-3f, seed * 1e-3f + 1e-3f}, s = (f2)(seed * 2e-3f), b = (f2)(seed * 3e-3f);
  const f2 ox = {-1.25f, -1.75f};
  for (int it = 0; it < ITER; ++it) {
    taps<f2>(q, ox, s, b, (f2)(-0.3f), (f2)(0.02f), (f2)(0.4f), (f2)(seed), (f2)(seed + 1), (f2)(seed + 2), aR, aG, aB, aW);
    q = q + (f2)(1e-6f);
  }
  out[blockIdx.x * 256 + threadIdx.x] = aR.x + aG.x + aB.x + aW.x + aR.y + aG.y + aB.y + aW.y;
}
template <class K> float timeit(K k, int blocks, float* d) {
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / 20;
}
int main() {
  float* d; CK(hipMalloc(&d, 256 * 8 * 256 * 4));
  for (int bpc : {8, 4}) {
    const float ts = timeit(k_scalar, 256 * bpc, d), tp = timeit(k_packed, 256 * bpc, d);
    printf("%d waves/SIMD: scalar %.3f ms, packed %.3f ms (%.2fx)\n", bpc, ts, tp, ts / tp);
  }
  return 0;
}
