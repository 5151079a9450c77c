// Shared device-side pieces of the FSR 1.0 HIP kernels (gfx950 only).
//
// The translation units including this header are compiled with -ffp-contract=off: every
// fused multiply-add in the kernels is an explicit fmaf()/__builtin_elementwise_fma, so the
// "EXACT" variants keep the reference's operation order and rounding (ffx_fsr1.h), and the
// default variants fuse only where the filter is continuous in its inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "fsr1_hip.h"

namespace fsr1 {

typedef _Float16 half_t;
typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// Output tile of one 256-thread workgroup (4 waves): 64 x 16 pixels, each wave owns 4 rows, a
// lane owns one column -> every global store instruction writes 64 consecutive pixels.
constexpr int kTileW = 64;
#ifndef FSR1_EASU_TILE_H
#define FSR1_EASU_TILE_H 16
#endif
constexpr int kTileH = FSR1_EASU_TILE_H;  // multiple of 4 (rows are split over the four waves)
constexpr int kThreads = 256;
#ifndef FSR1_FUSED_TILE_H
#define FSR1_FUSED_TILE_H 16
#endif
constexpr int kFusedTileH = FSR1_FUSED_TILE_H;  // output rows per fused-kernel tile (multiple of 4)
constexpr int kXcds = 8;  // MI355X: 8 XCDs, workgroup b is dispatched to XCD b % 8

struct ImageView {
  char* base;
  int width, height;
  long long pitch;         // bytes between rows
  long long frame_stride;  // bytes between frames
};

// Colour stages (fsr1_color_math.h); carried by every argument block, ignored by the plain kernels.
struct NoiseView {
  const char* base;  // slice already selected (frame % slices) by the host
  int width, height;
  long long pitch;
  int format;        // fsr1_format
  int off_x, off_y;  // noise_offset reduced to [0, width) x [0, height) by the host
  float rcp_width, rcp_height;  // 1.0f / width, 1.0f / height (wrap_mod)
};

struct ColorArgs {
  uint32_t stages;  // FSR1_COLOR_*
  float amount, bias;
  uint32_t frame;
  NoiseView noise;
};

struct ColorPassArgs {
  ImageView in, out;
  int tiles_x, tiles_y, frames;
  ColorArgs color;
};

struct EasuArgs {
  ImageView in, out;
  uint32_t con[16];
  int tiles_x, tiles_y, frames;
  int fp_w, fp_h;  // LDS footprint capacity (texels) per tile, >= the largest footprint of any tile
  uint32_t flags;
  ColorArgs color;
};

struct RcasArgs {
  ImageView in, out;
  uint32_t con[4];
  int tiles_x, tiles_y, frames;
  int rows;  // rows per strip (a multiple of the kernel's row ring), chosen by the launcher
  uint32_t flags;
  ColorArgs color;
};

struct FusedArgs {
  ImageView in, out;
  uint32_t easu_con[16];
  uint32_t rcas_con[4];
  int tiles_x, tiles_y, frames;
  int fp_w, fp_h;
  uint32_t flags;
  ColorArgs color;
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

// XCD-aware workgroup -> tile mapping.  Consecutive workgroup ids round-robin over the 8 XCDs
// (each with a private 4 MiB L2), so the ids that land on one XCD are given one contiguous range
// of tiles: neighbouring tiles, which share their input aprons, then share an L2.
__device__ __forceinline__ int xcd_swizzle(int b, int n) {
  const int q = n / kXcds, r = n % kXcds;
  const int xcd = b % kXcds, idx = b / kXcds;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// Kernels that may need more than the default 48 KiB of dynamic LDS ask for it through hipFuncSetAttribute — once per
// kernel, device and size: the attribute sticks, and re-issuing the call on every launch costs host time on the
// launch path (round 1 did).  The cache is keyed by (device, function) and only ever grows.
inline hipError_t ensure_dynamic_lds(const void* fn, size_t lds) {
  if (lds <= 48 * 1024) return hipSuccess;
  struct Slot { std::atomic<const void*> fn{nullptr}; std::atomic<size_t> bytes{0}; };
  constexpr int kDevices = 16, kSlots = 64;
  static Slot cache[kDevices][kSlots];
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  Slot* row = cache[dev >= 0 && dev < kDevices ? dev : 0];
  Slot* slot = nullptr;
  if (dev >= 0 && dev < kDevices) {
    for (int i = 0; i < kSlots && !slot; ++i) {
      const void* cur = row[i].fn.load(std::memory_order_acquire);
      if (cur == fn) slot = &row[i];
      else if (!cur) {
        const void* expected = nullptr;
        if (row[i].fn.compare_exchange_strong(expected, fn, std::memory_order_acq_rel) || expected == fn) slot = &row[i];
      }
    }
  }
  if (slot && slot->bytes.load(std::memory_order_acquire) >= lds) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
  if (slot) slot->bytes.store(lds, std::memory_order_release);
  return hipSuccess;
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
  static __device__ __forceinline__ T store(float r, float g, float b, float a) { return T{to_half(r), to_half(g), to_half(b), to_half(a)}; }
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

}  // namespace fsr1
