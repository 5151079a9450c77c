// Colour stages either side of the filters (ffx-fsr/ffx_fsr1.h:986-1199): FsrSrtmF / FsrSrtmInvF (simple
// reversible tone-mapper), FsrLfgaF (linear film grain applicator), FsrTepdDitF + FsrTepdC8F / FsrTepdC10F
// (temporal energy preserving dither).  Shared by the stand-alone colour pass (fsr1_color.hip) and by the
// prologue / epilogue variants of the EASU, RCAS and fused kernels.
//
// Numerics: binary32 in the reference's operation order; the translation units are built with
// -ffp-contract=off, so nothing here fuses.  The two ARcpF1 reciprocals (:1042, :1044) are IEEE divisions
// when EXACT and v_rcp_f32 (1 ulp) otherwise; everything else — including the whole of FsrTepd*F, whose
// floor() and greater-than-zero test are discontinuities — is evaluated identically in both modes.
#pragma once
#include "fsr1_device_base.hpp"

namespace fsr1 {

// Arguments of the colour stages (what fsr1_color_stages describes on the host).
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


constexpr uint32_t kColorPrologue = FSR1_COLOR_SRTM;
constexpr uint32_t kColorEpilogue = FSR1_COLOR_LFGA | FSR1_COLOR_SRTM_INV | FSR1_COLOR_TEPD_C8 | FSR1_COLOR_TEPD_C10;
constexpr uint32_t kColorNeedsNoise = FSR1_COLOR_LFGA | FSR1_COLOR_DITHER_FROM_NOISE;

struct rgb3_t { float r, g, b; };

template <bool EXACT>
__device__ __forceinline__ float color_rcp(float x) { return EXACT ? rcp_ieee(x) : __builtin_amdgcn_rcpf(x); }

// :1042 FsrSrtmF
template <bool EXACT>
__device__ __forceinline__ void FsrSrtmF(float& r, float& g, float& b) {
  const float k = color_rcp<EXACT>(max3f(r, g, b) + 1.0f);
  r *= k; g *= k; b *= k;
}
// :1044 FsrSrtmInvF ("the extra max solves the c=1.0 case")
template <bool EXACT>
__device__ __forceinline__ void FsrSrtmInvF(float& r, float& g, float& b) {
  const float k = color_rcp<EXACT>(fmaxf((float)(1.0 / 32768.0), 1.0f - max3f(r, g, b)));
  r *= k; g *= k; b *= k;
}
// :1012 FsrLfgaF  c += (t*a) * min(1-c, c)
__device__ __forceinline__ float FsrLfgaF1(float c, float t, float a) { return c + (t * a) * fminf(1.0f - c, c); }

// :1082-1091 FsrTepdDitF.  The header's constant expressions are folded in double and rounded once (the
// oracle's pin); x*a and y*b round separately, v_fract_f32 is x - floor(x).
__device__ __forceinline__ float FsrTepdDitF(uint32_t px, uint32_t py, uint32_t f) {
  float x = (float)(px + f);
  const float y = (float)py;
  const float a = 1.61803400516510009765625f;  // float((1 + sqrt(5))/2)
  const float b = (float)(1.0 / 3.69);
  x = x * a + (y * b);
  return x - floorf(x);
}

// ffx_a.h:1499 AGtZeroF1 = saturate(m * +INF): 1 for m > 0, else 0 (0*inf = NaN clamps to 0)
__device__ __forceinline__ float AGtZeroF1(float m) { return sat(m * __builtin_inff()); }

// :1097-1110 FsrTepdC8F (k = 255) / :1113-1120 FsrTepdC10F (k = 1023), one channel; rk = float(1.0 / k).
// k and rk are run-time values so that both widths share one copy of the code (it is inlined per pixel).
__device__ __forceinline__ float FsrTepdCF1(float c, float dit, float k, float rk) {
  float n = sqrtf(c);  // correctly rounded (hipcc default)
  n = floorf(n * k) * rk;
  const float a = n * n;
  float b = n + rk;
  b = b * b;
  const float r = (c - b) * APrxMedRcpF1<true>(a - b);
  return sat(n + AGtZeroF1(dit - r) * rk);
}

// v mod d for v < 2^23 without an integer division (a runtime-divisor `%` expands to ~30 instructions and this runs
// once or twice per pixel): the float quotient is within one of the true one, the remainder is fixed up.
__device__ __forceinline__ uint32_t wrap_mod(uint32_t v, uint32_t d, float rcp_d) {
  const uint32_t q = (uint32_t)((float)v * rcp_d);
  int32_t r = (int32_t)(v - q * d);
  r += r < 0 ? (int32_t)d : 0;
  r -= r >= (int32_t)d ? (int32_t)d : 0;
  return (uint32_t)r;
}

// One RGBA texel of the tiled noise texture for pixel (x, y); wrap addressing.
__device__ __forceinline__ float4_t noise_fetch(const NoiseView& nv, uint32_t x, uint32_t y) {
  const uint32_t nx = wrap_mod(x + (uint32_t)nv.off_x, (uint32_t)nv.width, nv.rcp_width);
  const uint32_t ny = wrap_mod(y + (uint32_t)nv.off_y, (uint32_t)nv.height, nv.rcp_height);
  const char* row = nv.base + (long long)ny * nv.pitch;
  switch (nv.format) {
    case FSR1_FORMAT_RGBA16F: return Pixel<FSR1_FORMAT_RGBA16F>::load(reinterpret_cast<const half4_t*>(row)[nx]);
    case FSR1_FORMAT_RGBA32F: return reinterpret_cast<const float4_t*>(row)[nx];
    case FSR1_FORMAT_RGBA8_UNORM: return Pixel<FSR1_FORMAT_RGBA8_UNORM>::load(reinterpret_cast<const uint32_t*>(row)[nx]);
    default: return Pixel<FSR1_FORMAT_R10G10B10A2_UNORM>::load(reinterpret_cast<const uint32_t*>(row)[nx]);
  }
}

// Prologue: FsrSrtmF on a texel as it is loaded.
template <bool EXACT>
__device__ __forceinline__ void color_prologue(const ColorArgs& ca, float& r, float& g, float& b) {
  if (ca.stages & FSR1_COLOR_SRTM) FsrSrtmF<EXACT>(r, g, b);
}
template <bool EXACT>
__device__ __forceinline__ float4_t color_prologue(const ColorArgs& ca, float4_t c) {
  float r = c.x, g = c.y, b = c.z;
  color_prologue<EXACT>(ca, r, g, b);
  return float4_t{r, g, b, c.w};
}
template <bool EXACT>
__device__ __forceinline__ float4_t color_epilogue(const ColorArgs& ca, uint32_t x, uint32_t y, float4_t c);

// Epilogue on the result for output pixel (x, y): LFGA -> SRTM_INV -> TEPD.
template <bool EXACT>
__device__ __forceinline__ void color_epilogue(const ColorArgs& ca, uint32_t x, uint32_t y, float& r, float& g, float& b) {
  const uint32_t st = ca.stages;
  float4_t n = {0.f, 0.f, 0.f, 0.f};
  if (st & kColorNeedsNoise) n = noise_fetch(ca.noise, x, y);
  if (st & FSR1_COLOR_LFGA) {
    r = FsrLfgaF1(r, n.x + ca.bias, ca.amount);
    g = FsrLfgaF1(g, n.y + ca.bias, ca.amount);
    b = FsrLfgaF1(b, n.z + ca.bias, ca.amount);
  }
  if (st & FSR1_COLOR_SRTM_INV) FsrSrtmInvF<EXACT>(r, g, b);
  if (st & (FSR1_COLOR_TEPD_C8 | FSR1_COLOR_TEPD_C10)) {
    const float dit = (st & FSR1_COLOR_DITHER_FROM_NOISE) ? sat(n.w) : FsrTepdDitF(x, y, ca.frame);
    const bool c8 = (st & FSR1_COLOR_TEPD_C8) != 0;
    const float k = c8 ? 255.0f : 1023.0f, rk = c8 ? (float)(1.0 / 255.0) : (float)(1.0 / 1023.0);
    r = FsrTepdCF1(r, dit, k, rk); g = FsrTepdCF1(g, dit, k, rk); b = FsrTepdCF1(b, dit, k, rk);
  }
  r = pinned(r); g = pinned(g); b = pinned(b);  // the narrowing that follows must round these binary32 values
}

template <bool EXACT>
__device__ __forceinline__ float4_t color_epilogue(const ColorArgs& ca, uint32_t x, uint32_t y, float4_t c) {
  float r = c.x, g = c.y, b = c.z;
  color_epilogue<EXACT>(ca, x, y, r, g, b);
  return float4_t{r, g, b, c.w};
}

}  // namespace fsr1
