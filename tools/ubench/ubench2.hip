// Cycle-accurate VALU/LDS issue costs on gfx950: every wave times its own instruction stream with
// s_memtime (shader clock) and the constant 100 MHz wall clock, at full occupancy (8 waves/SIMD) and at 1 wave/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2048;
constexpr int UNROLL = 16;

#define OPK(name, setup, stmt)                                                                    \
  __global__ void __launch_bounds__(256) name(unsigned long long* out, float seed) {              \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    float b0 = seed * 3, b1 = seed * 5;                                                           \
    setup;                                                                                        \
    unsigned long long t0 = __builtin_readcyclecounter();                                         \
    unsigned long long w0 = wall_clock64();                                                       \
    for (int it = 0; it < ITER; ++it) { stmt stmt }                                               \
    unsigned long long t1 = __builtin_readcyclecounter();                                         \
    unsigned long long w1 = wall_clock64();                                                       \
    asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));   \
    if ((threadIdx.x & 63) == 0) { size_t w = (blockIdx.x * 4 + (threadIdx.x >> 6)); out[2 * w] = t1 - t0; out[2 * w + 1] = w1 - w0; } \
  }

#define R8(op) asm volatile(op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));

#define I_FMA(x)  "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
#define GEN8(ins) ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n"
#define GEN8_3(ins) ins " %0, %0, %8, %9\n" ins " %1, %1, %8, %9\n" ins " %2, %2, %8, %9\n" ins " %3, %3, %8, %9\n" ins " %4, %4, %8, %9\n" ins " %5, %5, %8, %9\n" ins " %6, %6, %8, %9\n" ins " %7, %7, %8, %9\n"
#define GEN8_1(ins) ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7\n"
#define ASM8(str) asm volatile(str : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));

OPK(k_fma, , ASM8(GEN8_3("v_fma_f32")))
OPK(k_mul, , ASM8(GEN8("v_mul_f32")))
OPK(k_add, , ASM8(GEN8("v_add_f32")))
OPK(k_min, , ASM8(GEN8("v_min_f32")))
OPK(k_min3, , ASM8(GEN8_3("v_min3_f32")))
OPK(k_subu, , ASM8(GEN8("v_sub_u32")))
OPK(k_and, , ASM8(GEN8("v_and_b32")))
OPK(k_mov, , ASM8(GEN8_1("v_mov_b32")))
OPK(k_cvt_f32_f16, , ASM8(GEN8_1("v_cvt_f32_f16")))
OPK(k_cvt_f16_f32, , ASM8(GEN8_1("v_cvt_f16_f32")))
OPK(k_rcp, , ASM8(GEN8_1("v_rcp_f32")))
OPK(k_rsq, , ASM8(GEN8_1("v_rsq_f32")))
OPK(k_floor, , ASM8(GEN8_1("v_floor_f32")))
OPK(k_fma_mix, , ASM8(GEN8_3("v_fma_mix_f32")))
OPK(k_pk_fma_f16, , ASM8(GEN8_3("v_pk_fma_f16")))
OPK(k_pk_mul_f16, , ASM8(GEN8("v_pk_mul_f16")))
OPK(k_pk_min_f16, , ASM8(GEN8("v_pk_min_f16")))
OPK(k_dot2_f32_f16, , ASM8(GEN8_3("v_dot2_f32_f16")))
OPK(k_med3, , ASM8(GEN8_3("v_med3_f32")))
OPK(k_cndmask, , ASM8(GEN8("v_cndmask_b32")))

// packed fp32 needs 64-bit operands
typedef float float2_t __attribute__((ext_vector_type(2)));
#define OPK2(name, ins3)                                                                          \
  __global__ void __launch_bounds__(256) name(unsigned long long* out, float seed) {              \
    float2_t a0 = {seed, seed}, a1 = {seed + 1, seed}, a2 = {seed + 2, seed}, a3 = {seed + 3, seed}, a4 = {seed + 4, seed}, a5 = {seed + 5, seed}, a6 = {seed + 6, seed}, a7 = {seed + 7, seed}; \
    float2_t b0 = {seed * 3, seed}, b1 = {seed * 5, seed};                                        \
    unsigned long long t0 = __builtin_readcyclecounter();                                         \
    unsigned long long w0 = wall_clock64();                                                       \
    for (int it = 0; it < ITER; ++it) { ASM8(ins3) ASM8(ins3) }                                   \
    unsigned long long t1 = __builtin_readcyclecounter();                                         \
    unsigned long long w1 = wall_clock64();                                                       \
    asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));   \
    if ((threadIdx.x & 63) == 0) { size_t w = (blockIdx.x * 4 + (threadIdx.x >> 6)); out[2 * w] = t1 - t0; out[2 * w + 1] = w1 - w0; } \
  }
OPK2(k_pk_fma_f32, GEN8_3("v_pk_fma_f32"))
OPK2(k_pk_mul_f32, GEN8("v_pk_mul_f32"))
OPK2(k_pk_add_f32, GEN8("v_pk_add_f32"))
OPK2(k_pk_mov_b32, GEN8("v_pk_mov_b32"))

// LDS: 8 independent reads per statement, waited at the end of each statement
template <int BYTES>
__global__ void __launch_bounds__(256) k_lds(unsigned long long* out, float seed) {
  __shared__ float4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float4(i, i, i, i);
  __syncthreads();
  unsigned addr = (threadIdx.x & 63) * BYTES + (threadIdx.x >> 6) * 4096;
  float4 r0, r1, r2, r3, r4, r5, r6, r7;
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long w0 = wall_clock64();
  for (int it = 0; it < ITER; ++it) {
    if (BYTES == 16)
      asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:1024\nds_read_b128 %2, %8 offset:2048\nds_read_b128 %3, %8 offset:3072\n"
                   "ds_read_b128 %4, %8 offset:4096\nds_read_b128 %5, %8 offset:5120\nds_read_b128 %6, %8 offset:6144\nds_read_b128 %7, %8 offset:7168\ns_waitcnt lgkmcnt(0)\n"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr));
    else if (BYTES == 8)
      asm volatile("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:1024\nds_read_b64 %2, %8 offset:2048\nds_read_b64 %3, %8 offset:3072\n"
                   "ds_read_b64 %4, %8 offset:4096\nds_read_b64 %5, %8 offset:5120\nds_read_b64 %6, %8 offset:6144\nds_read_b64 %7, %8 offset:7168\ns_waitcnt lgkmcnt(0)\n"
                   : "=v"(*(double*)&r0), "=v"(*(double*)&r1), "=v"(*(double*)&r2), "=v"(*(double*)&r3), "=v"(*(double*)&r4), "=v"(*(double*)&r5), "=v"(*(double*)&r6), "=v"(*(double*)&r7) : "v"(addr));
    else
      asm volatile("ds_read_b32 %0, %8\nds_read_b32 %1, %8 offset:1024\nds_read_b32 %2, %8 offset:2048\nds_read_b32 %3, %8 offset:3072\n"
                   "ds_read_b32 %4, %8 offset:4096\nds_read_b32 %5, %8 offset:5120\nds_read_b32 %6, %8 offset:6144\nds_read_b32 %7, %8 offset:7168\ns_waitcnt lgkmcnt(0)\n"
                   : "=v"(r0.x), "=v"(r1.x), "=v"(r2.x), "=v"(r3.x), "=v"(r4.x), "=v"(r5.x), "=v"(r6.x), "=v"(r7.x) : "v"(addr));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long w1 = wall_clock64();
  asm volatile("" :: "v"(r0.x), "v"(r1.x), "v"(r2.x), "v"(r3.x), "v"(r4.x), "v"(r5.x), "v"(r6.x), "v"(r7.x));
  if ((threadIdx.x & 63) == 0) { size_t w = (blockIdx.x * 4 + (threadIdx.x >> 6)); out[2 * w] = t1 - t0; out[2 * w + 1] = w1 - w0; }
}

template <class K>
void run(const char* name, K k, int blocks_per_cu, int cus, unsigned long long* d, double insts_per_wave) {
  int blocks = cus * blocks_per_cu;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h(blocks * 4 * 2);
  CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks * 4; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  cyc /= blocks * 4; wall /= blocks * 4;
  int waves_per_simd = blocks_per_cu;  // 4 waves per block, 4 SIMDs
  double ghz = cyc / (wall / 100e6) * 1e-9;
  printf("%-16s %d waves/SIMD: %7.2f cycles per wave-inst per SIMD (s_memtime), shader clock %.2f GHz, kernel %.3f ms\n",
         name, waves_per_simd, cyc / (insts_per_wave * waves_per_simd), ghz, ms);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  unsigned long long* d; CK(hipMalloc(&d, cus * 8 * 4 * 2 * 8));
  const double n = (double)ITER * UNROLL;
  for (int bpc : {8, 1}) {
#define RUN(k) run(#k, k, bpc, cus, d, n);
    RUN(k_fma) RUN(k_mul) RUN(k_add) RUN(k_min) RUN(k_min3) RUN(k_med3) RUN(k_subu) RUN(k_and) RUN(k_mov) RUN(k_cndmask)
    RUN(k_cvt_f32_f16) RUN(k_cvt_f16_f32) RUN(k_floor) RUN(k_rcp) RUN(k_rsq) RUN(k_fma_mix) RUN(k_dot2_f32_f16)
    RUN(k_pk_fma_f16) RUN(k_pk_mul_f16) RUN(k_pk_min_f16) RUN(k_pk_fma_f32) RUN(k_pk_mul_f32) RUN(k_pk_add_f32) RUN(k_pk_mov_b32)
    run("ds_read_b128", k_lds<16>, bpc, cus, d, (double)ITER * 8);
    run("ds_read_b64", k_lds<8>, bpc, cus, d, (double)ITER * 8);
    run("ds_read_b32", k_lds<4>, bpc, cus, d, (double)ITER * 8);
  }
  return 0;
}
