// Micro-benchmarks that size the MI355X budgets the EASU/RCAS kernels are designed against:
// VALU issue rate of the instructions the kernels lean on, and the achievable HBM stream rate.
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench     Run: ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr int ITER = 4096;
constexpr int UNROLL = 8;  // independent chains

struct OpFmaF32 { typedef float T; static __device__ T init(float s, int u) { return s + u; }
  static __device__ void op(T& a, float s) { float b = s, c = s * 0.5f; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }
  static __device__ float fold(T a) { return a; } };
struct OpPkFmaF32 { typedef float2_t T; static __device__ T init(float s, int u) { T t; t.x = s + u; t.y = s; return t; }
  static __device__ void op(T& a, float s) { T b; b.x = s; b.y = s; T c; c.x = s; c.y = 1.0f; asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }
  static __device__ float fold(T a) { return a.x + a.y; } };
struct OpPkMulF32 { typedef float2_t T; static __device__ T init(float s, int u) { T t; t.x = s + u; t.y = s; return t; }
  static __device__ void op(T& a, float s) { T b; b.x = s; b.y = s; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  static __device__ float fold(T a) { return a.x + a.y; } };
struct OpPkAddF32 { typedef float2_t T; static __device__ T init(float s, int u) { T t; t.x = s + u; t.y = s; return t; }
  static __device__ void op(T& a, float s) { T b; b.x = s; b.y = s; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  static __device__ float fold(T a) { return a.x + a.y; } };
struct OpPkFmaF16 { typedef half2_t T; static __device__ T init(float s, int u) { T t; t.x = (_Float16)(s + u); t.y = (_Float16)s; return t; }
  static __device__ void op(T& a, float s) { T b; b.x = (_Float16)s; b.y = (_Float16)s; asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
  static __device__ float fold(T a) { return (float)a.x + (float)a.y; } };
struct OpPkMinF16 { typedef half2_t T; static __device__ T init(float s, int u) { T t; t.x = (_Float16)(s + u); t.y = (_Float16)s; return t; }
  static __device__ void op(T& a, float s) { T b; b.x = (_Float16)s; b.y = (_Float16)s; asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a) : "v"(b)); }
  static __device__ float fold(T a) { return (float)a.x + (float)a.y; } };
struct OpFmaMixF32 { typedef float T; static __device__ T init(float s, int u) { return s + u; }
  static __device__ void op(T& a, float s) { unsigned b = __float_as_uint(s); float c = s * 0.5f; asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(b), "v"(c)); }
  static __device__ float fold(T a) { return a; } };
struct OpMin3F32 { typedef float T; static __device__ T init(float s, int u) { return s + u; }
  static __device__ void op(T& a, float s) { float b = s, c = s * 0.5f; asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }
  static __device__ float fold(T a) { return a; } };
struct OpMinF32 { typedef float T; static __device__ T init(float s, int u) { return s + u; }
  static __device__ void op(T& a, float s) { float b = s; asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
  static __device__ float fold(T a) { return a; } };
struct OpCvtF32F16 { typedef float T; static __device__ T init(float s, int u) { return s + u; }
  static __device__ void op(T& a, float s) { asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a)); }
  static __device__ float fold(T a) { return a; } };
struct OpCvtF16F32 { typedef float T; static __device__ T init(float s, int u) { return s + u; }
  static __device__ void op(T& a, float s) { asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a)); }
  static __device__ float fold(T a) { return a; } };
struct OpRcpF32 { typedef float T; static __device__ T init(float s, int u) { return s + u; }
  static __device__ void op(T& a, float s) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a)); }
  static __device__ float fold(T a) { return a; } };
struct OpSubU32 { typedef unsigned T; static __device__ T init(float s, int u) { return __float_as_uint(s) + u; }
  static __device__ void op(T& a, float s) { unsigned b = __float_as_uint(s); asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a) : "v"(b)); }
  static __device__ float fold(T a) { return (float)a; } };

template <class OP>
__global__ void __launch_bounds__(256) k_valu(float* out, float seed) {
  typename OP::T a[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) a[u] = OP::init(seed, u);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) OP::op(a[u], seed);
  }
  float r = 0;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) r += OP::fold(a[u]);
  if (r == 12345.678f) out[threadIdx.x] = r;
}

// LDS read throughput
template <int BYTES>
__global__ void __launch_bounds__(256) k_lds_read(float* out, int stride) {
  __shared__ float4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_float4(i, i, i, i);
  __syncthreads();
  float acc = 0;
  int idx = (threadIdx.x * stride) & 2047;
  for (int it = 0; it < 1024; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (BYTES == 16) { float4 v = lds[(idx + u * 256) & 4095]; acc += v.x + v.w; }
      else if (BYTES == 8) { float2 v = ((float2*)lds)[(idx + u * 256) & 8191]; acc += v.x + v.y; }
      else { float v = ((float*)lds)[(idx + u * 256) & 16383]; acc += v; }
    }
    idx = (idx + 1) & 2047;
  }
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ in, float* out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  for (; i < n; i += stride) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = make_float4(1, 2, 3, 4);
}

template <class F>
float time_ms(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d MHz  memclock=%d MHz  L2=%d  LDS/block=%zu\n", p.name, p.multiProcessorCount,
         p.clockRate / 1000, p.memoryClockRate / 1000, p.l2CacheSize, p.sharedMemPerBlock);
  float* out; CK(hipMalloc(&out, 1 << 20));
  const int blocks = p.multiProcessorCount * 8;  // 8 x 256 threads = 32 waves per CU
#define RUN(OP)                                                                                    \
  {                                                                                                \
    float ms = time_ms([&] { hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }, 5); \
    double insts = (double)blocks * 256 * ITER * UNROLL;                                          \
    printf("%-14s %8.3f ms  %8.2f T lane-inst/s  (%.2f cycles per wave-inst per SIMD @2.4GHz)\n", #OP, ms, \
           insts / ms * 1e-9, 2.4e9 * 4 * p.multiProcessorCount * 64 / (insts / (ms * 1e-3)));     \
  }
  RUN(OpFmaF32) RUN(OpPkFmaF32) RUN(OpPkMulF32) RUN(OpPkAddF32) RUN(OpPkFmaF16) RUN(OpPkMinF16) RUN(OpFmaMixF32)
  RUN(OpMin3F32) RUN(OpMinF32) RUN(OpCvtF32F16) RUN(OpCvtF16F32) RUN(OpRcpF32) RUN(OpSubU32)
  for (int stride : {1, 2}) {
    float ms;
    double bytes = (double)blocks * 256 * 1024 * 8;
    ms = time_ms([&] { hipLaunchKernelGGL(k_lds_read<16>, dim3(blocks), dim3(256), 0, 0, out, stride); }, 3);
    printf("lds_read_b128 stride %d: %.2f TB/s (%.1f B/clk/CU @2.4GHz)\n", stride, bytes * 16 / ms * 1e-9, bytes * 16 / (ms * 1e-3) / 2.4e9 / p.multiProcessorCount);
    ms = time_ms([&] { hipLaunchKernelGGL(k_lds_read<8>, dim3(blocks), dim3(256), 0, 0, out, stride); }, 3);
    printf("lds_read_b64  stride %d: %.2f TB/s (%.1f B/clk/CU)\n", stride, bytes * 8 / ms * 1e-9, bytes * 8 / (ms * 1e-3) / 2.4e9 / p.multiProcessorCount);
    ms = time_ms([&] { hipLaunchKernelGGL(k_lds_read<4>, dim3(blocks), dim3(256), 0, 0, out, stride); }, 3);
    printf("lds_read_b32  stride %d: %.2f TB/s (%.1f B/clk/CU)\n", stride, bytes * 4 / ms * 1e-9, bytes * 4 / (ms * 1e-3) / 2.4e9 / p.multiProcessorCount);
  }
  for (size_t mb : {64, 512, 2048}) {
    size_t n = mb * 1024 * 1024 / 16;
    float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 0, n * 16));
    for (int g : {2048, 8192}) {
      float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); }, 10);
      printf("copy  %5zu MiB grid %5d: %.3f ms  %.2f TB/s (read+write)\n", mb, g, ms, 2.0 * n * 16 / ms * 1e-9);
      ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, out, n); }, 10);
      printf("read  %5zu MiB grid %5d: %.3f ms  %.2f TB/s\n", mb, g, ms, 1.0 * n * 16 / ms * 1e-9);
      ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); }, 10);
      printf("write %5zu MiB grid %5d: %.3f ms  %.2f TB/s\n", mb, g, ms, 1.0 * n * 16 / ms * 1e-9);
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  // launch overhead: empty-ish kernel back to back
  {
    float ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, 0, (float4*)out, (size_t)1); }, 200);
    printf("back-to-back tiny launch: %.2f us\n", ms * 1e3);
  }
  return 0;
}
