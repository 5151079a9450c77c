// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950: (1) operand/result lane layout, (2) issue cost next to VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float float4_t __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
  const int l = threadIdx.x;
  const float a = 1.0f + l;          // A: lane l
  const float b = 100.0f * (l + 1);  // B: lane l
  float4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  out[l * 4 + 0] = c.x; out[l * 4 + 1] = c.y; out[l * 4 + 2] = c.z; out[l * 4 + 3] = c.w;
}

constexpr int ITER = 4096;
// 12 "taps": 7 independent-ish VALU fmas for the weight + accumulate by 4 VALU fmas
__global__ void __launch_bounds__(256) k_valu(float* out, float seed) {
  float w = seed, u = seed * 0.5f, aR = 0, aG = 0, aB = 0, aW = 0;
  float cr = seed + 1, cg = seed + 2, cb = seed + 3;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      u = fmaf(u, 0.99f, 0.01f); float b0 = fmaf(u, 0.5f, -1.25f); b0 = fmaf(b0, u, 1.0f); float wa = fmaf(u, 0.7f, -1.0f);
      wa *= wa; w = b0 * wa; u = fmaf(w, 0.001f, u);
      aR = fmaf(cr, w, aR); aG = fmaf(cg, w, aG); aB = fmaf(cb, w, aB); aW += w;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = aR + aG + aB + aW;
}
__global__ void __launch_bounds__(256) k_mfma(float* out, float seed) {
  float w = seed, u = seed * 0.5f;
  float4_t acc = {0, 0, 0, 0};
  float ch = seed + (threadIdx.x & 3);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      u = fmaf(u, 0.99f, 0.01f); float b0 = fmaf(u, 0.5f, -1.25f); b0 = fmaf(b0, u, 1.0f); float wa = fmaf(u, 0.7f, -1.0f);
      wa *= wa; w = b0 * wa; u = fmaf(w, 0.001f, u);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(ch, w, acc, 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
__global__ void __launch_bounds__(256) k_none(float* out, float seed) {
  float w = seed, u = seed * 0.5f, s = 0;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      u = fmaf(u, 0.99f, 0.01f); float b0 = fmaf(u, 0.5f, -1.25f); b0 = fmaf(b0, u, 1.0f); float wa = fmaf(u, 0.7f, -1.0f);
      wa *= wa; w = b0 * wa; u = fmaf(w, 0.001f, u);
    }
    s += w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K> float timeit(K k, int blocks, float* d) {
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  float* d; CK(hipMalloc(&d, 256 * 8 * 256 * 4));
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
  std::vector<float> h(256);
  CK(hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost));
  // hypothesis: lane 4b+j, reg i holds A(lane 4b+i) * B(lane 4b+j)
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int b = l / 4, j = l % 4, i = r;
    const float want = (1.0f + 4 * b + i) * 100.0f * (4 * b + j + 1);
    if (h[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], want); ++bad; }
  }
  printf("layout D[i][j].blk b -> lane 4b+j, vgpr i with A row i from lane 4b+i, B col j from lane 4b+j: %s (%d mismatches)\n", bad ? "NO" : "CONFIRMED", bad);
  for (int bpc : {8, 4, 2}) {
    const int blocks = 256 * bpc;
    const float tn = timeit(k_none, blocks, d), tv = timeit(k_valu, blocks, d), tm = timeit(k_mfma, blocks, d);
    const double taps = (double)ITER * 12 * bpc;  // wave-taps per SIMD
    printf("%d waves/SIMD: weights only %.3f ms | +4 VALU fma accumulate %.3f ms | +1 MFMA 4x4x1 accumulate %.3f ms  (cycles/tap @2.4GHz: %.1f / %.1f / %.1f)\n",
           bpc, tn, tv, tm, tn * 2.4e6 / taps, tv * 2.4e6 / taps, tm * 2.4e6 / taps);
  }
  return 0;
}
