// Device-side base of the FSR 1.0 HIP operator surface (gfx950 only): value types, the ffx_a.h helpers the
// filters are defined with (APrx*, min3 / max3, saturate), storage formats and the image view the tiled forms read.
//
// Part of the public, header-only device library (include/fsr1_device.hpp is the umbrella).  Translation units
// including these headers must be compiled with -ffp-contract=off: every fused multiply-add is an explicit
// fmaf(), so the EXACT variants keep the reference's operation order and rounding (ffx_fsr1.h) and the default
// variants fuse only where the filter is continuous in its inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fsr1_hip.h"

namespace fsr1 {

typedef _Float16 half_t;
typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// A batch of pitch-linear images in device memory (what fsr1_image describes on the host).
struct ImageView {
  char* base;
  int width, height;
  long long pitch;         // bytes between rows
  long long frame_stride;  // bytes between frames
};

__device__ __forceinline__ float as_f32(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __float_as_uint(f); }

// ffx_a.h:1843-1845 — integer-trick approximations; one v_sub_u32 (+ one shift) each.  They are part
// of the algorithm's definition (results differ from v_rcp_f32/v_rsq_f32), so they stay as they are.
__device__ __forceinline__ float APrxLoRcpF1(float a) { return as_f32(0x7ef07ebbu - as_u32(a)); }
__device__ __forceinline__ float APrxLoRsqF1(float a) { return as_f32(0x5f347d74u - (as_u32(a) >> 1)); }
template <bool EXACT>
__device__ __forceinline__ float APrxMedRcpF1(float a) {
  float b = as_f32(0x7ef19fffu - as_u32(a));
  return EXACT ? b * (-b * a + 2.0f) : b * fmaf(-b, a, 2.0f);
}
// min/max with IEEE minNum/maxNum semantics = v_min_f32/v_max_f32 in IEEE mode (fminf/fmaxf lower to them).
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(a, fminf(b, c)); }  // v_min3_f32
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }  // v_max3_f32
__device__ __forceinline__ float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }               // clamp modifier / v_med3

// a*b+c : two roundings when EXACT (reference order), one (v_fma_f32 / v_fma_mix_f32) otherwise.
template <bool EXACT>
__device__ __forceinline__ float mad(float a, float b, float c) { return EXACT ? a * b + c : fmaf(a, b, c); }

// The correctly rounded reciprocal 1.0f / x of the EXACT variants (GLSL `1.0 / x`, ARcpF1: the oracle's pin), in 6 instructions
// instead of the 11 of the general IEEE division: one Newton step from v_rcp_f32 (1 ulp) is the correctly rounded quotient for
// EVERY operand with a biased exponent in 1 .. 252 — checked bit for bit over all 2^32 operands on the device
// (fsr1_selftest, tools/ubench/rcp_exhaustive.hip) — v_div_fixup_f32 supplies the IEEE results for 0, inf and NaN, and the
// operands that are left (denormal x, and |x| >= 2^126 whose quotient is denormal) take the general division behind a branch no
// image value ever takes.
__device__ __forceinline__ float rcp_ieee(float x) {
  const float y = __builtin_amdgcn_rcpf(x);
  const float y1 = fmaf(y, fmaf(-x, y, 1.0f), y);
  float r = __builtin_amdgcn_div_fixupf(y1, x, 1.0f);
  if (__builtin_expect(__builtin_amdgcn_classf(x, 0x090) || fabsf(x) >= 0x1.0p+126f, 0)) r = 1.0f / x;  // 0x090: -denormal | +denormal
  return r;
}

// Optimisation barrier for a value that is about to be narrowed.  LLVM folds fptrunc(fmul) / fptrunc(fma)
// into v_fma_mixlo_f16 even with -ffp-contract=off, i.e. ONE rounding of the exact product to binary16
// instead of the reference's binary32 product followed by the store's rounding.  The EXACT variants pin
// the binary32 value first (costs no instruction).
// `volatile` on purpose: clang then treats the statement as touching memory, which also stops it from merging the LDS
// loads of consecutive pixels of one lane (the exact-2x EASU variant computes four pixels on the same 12-tap window).
// Merged, the window stays live in ~50 more VGPRs: 96 instead of 53, 5 waves per SIMD instead of 8, and the kernel
// takes 55.0 us instead of 48.7 us (capped at 64 VGPRs it spills: 87 us).  LDS bandwidth is not what EASU is short of.
__device__ __forceinline__ float pinned(float x) { asm volatile("" : "+v"(x)); return x; }

// Store of a pass's output: non-temporal when the image is the pipeline's last (FSR1_FLAG_OUTPUT_STREAMING), plain when a
// reader follows.  `streaming` is wave-uniform (a kernel argument), so this is a scalar branch around one store.
// ALIGN: what the address is known to be aligned to (a pair of texels is only texel-aligned).
template <int ALIGN, class T>
__device__ __forceinline__ void store_out(void* p, T v, bool streaming) {
  typedef T aligned_t __attribute__((aligned(ALIGN)));
  if (streaming) {
    // The empty statements keep this store distinct for the optimiser: it otherwise hoists / sinks the two branches'
    // stores into one and, intersecting their metadata, drops the non-temporal hint.  They emit nothing.
    asm volatile("");
    __builtin_nontemporal_store(v, reinterpret_cast<aligned_t*>(p));
    asm volatile("");
  } else {
    *reinterpret_cast<aligned_t*>(p) = v;
  }
}

// RTNE float -> binary16 (v_cvt_f16_f32 under the default rounding mode; never cvt_pkrtz).
__device__ __forceinline__ half_t to_half(float f) { return (half_t)f; }

// ARcpH1 (GLSL `1.0/x`, ffx_a.h:1005) — the binary16 reciprocal, correctly rounded: v_rcp_f32 (1 ulp in binary32)
// of the widened operand, narrowed RTNE.  That this equals the correctly rounded quotient for every one of the
// 65536 binary16 operands is checked on the device by fsr1_selftest() (an IEEE binary32 division narrowed
// to binary16 is correctly rounded because 24 >= 2*11+2; v_rcp_f16 and LLVM's f16 `1.0/x` are not).
__device__ __forceinline__ half_t half_rcp(half_t a) { return (half_t)__builtin_amdgcn_rcpf((float)a); }

template <int FMT> struct Pixel;  // FMT = fsr1_format
template <> struct Pixel<FSR1_FORMAT_RGBA16F> {
  typedef half4_t T;
  static __device__ __forceinline__ float4_t load(const T& p) { return float4_t{(float)p.x, (float)p.y, (float)p.z, (float)p.w}; }
  // two v_cvt_pk_f16_f32 (RTNE, two values per instruction): written as vector conversions, because from the element-wise
  // form the compiler builds three v_cvt_f16_f32 and two v_perm_b32 once the values have been through pinned()
  static __device__ __forceinline__ T store(float r, float g, float b, float a) {
    const half2_t rg = __builtin_convertvector(float2_t{r, g}, half2_t), ba = __builtin_convertvector(float2_t{b, a}, half2_t);
    return T{rg.x, rg.y, ba.x, ba.y};
  }
  static __device__ __forceinline__ T zero() { return T{(half_t)0, (half_t)0, (half_t)0, (half_t)0}; }
};
// UNORM decode: code / N correctly rounded (N = 255, 1023, 3) without a division: the product by the rounded
// reciprocal is within 1 ulp, one Newton residual step lands on the correctly rounded quotient (checked
// exhaustively for every code).  Encode: (uint) fma(clamp(x,0,1), N, 0.5), truncating; NaN -> 0.
template <int N>
__device__ __forceinline__ float unorm_decode(uint32_t code) {
  const float c = (float)code;
  const float r = 1.0f / (float)N;  // compile-time constant
  const float q = c * r;
  return fmaf(fmaf(-q, (float)N, c), r, q);
}
template <int N>
__device__ __forceinline__ uint32_t unorm_encode(float x) {
  return (uint32_t)fmaf(fminf(fmaxf(x, 0.0f), 1.0f), (float)N, 0.5f);
}
template <> struct Pixel<FSR1_FORMAT_RGBA8_UNORM> {
  typedef uint32_t T;
  static __device__ __forceinline__ float4_t load(const T& p) {
    return float4_t{unorm_decode<255>(p & 0xffu), unorm_decode<255>((p >> 8) & 0xffu), unorm_decode<255>((p >> 16) & 0xffu), unorm_decode<255>(p >> 24)};
  }
  static __device__ __forceinline__ T store(float r, float g, float b, float a) {
    return unorm_encode<255>(r) | (unorm_encode<255>(g) << 8) | (unorm_encode<255>(b) << 16) | (unorm_encode<255>(a) << 24);
  }
  static __device__ __forceinline__ T zero() { return 0u; }
};
template <> struct Pixel<FSR1_FORMAT_R10G10B10A2_UNORM> {
  typedef uint32_t T;
  static __device__ __forceinline__ float4_t load(const T& p) {
    return float4_t{unorm_decode<1023>(p & 0x3ffu), unorm_decode<1023>((p >> 10) & 0x3ffu), unorm_decode<1023>((p >> 20) & 0x3ffu), unorm_decode<3>(p >> 30)};
  }
  static __device__ __forceinline__ T store(float r, float g, float b, float a) {
    return unorm_encode<1023>(r) | (unorm_encode<1023>(g) << 10) | (unorm_encode<1023>(b) << 20) | (unorm_encode<3>(a) << 30);
  }
  static __device__ __forceinline__ T zero() { return 0u; }
};
template <> struct Pixel<FSR1_FORMAT_RGBA32F> {
  typedef float4_t T;
  static __device__ __forceinline__ float4_t load(const T& p) { return p; }
  static __device__ __forceinline__ T store(float r, float g, float b, float a) { return T{r, g, b, a}; }
  static __device__ __forceinline__ T zero() { return T{0.f, 0.f, 0.f, 0.f}; }
};

// Two horizontally adjacent texels as ONE value, so that a lane owning both moves them with a single access.
template <int FMT> struct TexelPair;
template <> struct TexelPair<FSR1_FORMAT_RGBA16F> {
  typedef half_t T __attribute__((ext_vector_type(8), aligned(8)));
  static __device__ __forceinline__ T make(half4_t a, half4_t b) { return T{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}; }
};
template <> struct TexelPair<FSR1_FORMAT_RGBA32F> {
  typedef float T __attribute__((ext_vector_type(8), aligned(16)));
  static __device__ __forceinline__ T make(float4_t a, float4_t b) { return T{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}; }
};
template <> struct TexelPair<FSR1_FORMAT_RGBA8_UNORM> {
  typedef uint32_t T __attribute__((ext_vector_type(2), aligned(4)));
  static __device__ __forceinline__ T make(uint32_t a, uint32_t b) { return T{a, b}; }
};
template <> struct TexelPair<FSR1_FORMAT_R10G10B10A2_UNORM> {
  typedef uint32_t T __attribute__((ext_vector_type(2), aligned(4)));
  static __device__ __forceinline__ T make(uint32_t a, uint32_t b) { return T{a, b}; }
};

}  // namespace fsr1
