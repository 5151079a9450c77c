// FsrRcasF per-pixel arithmetic (ffx-fsr/ffx_fsr1.h:684-769), shared by the RCAS kernel and the fused
// EASU->RCAS kernel.
#pragma once
#include "fsr1_device_base.hpp"

namespace fsr1 {

struct rgb_t { float r, g, b; };

// DPP wave shifts (GFX9 encodings): the value of the lane to the left / right; where that lane does not exist
// (lane 0 for shr, lane 63 for shl) the destination keeps `keep`.
constexpr int kDppWaveShr1 = 0x138;  // lane i <- lane i-1
constexpr int kDppWaveShl1 = 0x130;  // lane i <- lane i+1
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float keep, float v) {
  return as_f32((uint32_t)__builtin_amdgcn_update_dpp((int)as_u32(keep), (int)as_u32(v), CTRL, 0xf, 0xf, false));
}

// IEEE minNum / maxNum as single instructions on operands the compiler must not "canonicalise" first (see rcas_pixel)
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// One pixel of FsrRcasF from its 5 taps (b above, d left, e centre, f right, h below).
template <bool EXACT>
__device__ __forceinline__ rgb_t rcas_pixel(rgb_t b, rgb_t d, rgb_t e, rgb_t f, rgb_t h, float sharp, uint32_t flags) {  // flags: compile-time 0 in the plain variant
  // :741-746 min and max of the ring, per channel.  v_min3 / v_max3 / v_min / v_max written out (vmin3 ... below): the taps
  // reach this function from memory, from loop-carried registers and through DPP moves, where the compiler no longer
  // knows they are canonical and would spend a `v_max_f32 x, x, x` on every one of them before fminf / fmaxf (18 extra
  // instructions per two pixels in the streaming kernel).  The instructions themselves are IEEE minNum / maxNum.
  const float mn4R = vmin2(vmin3(b.r, d.r, f.r), h.r), mn4G = vmin2(vmin3(b.g, d.g, f.g), h.g), mn4B = vmin2(vmin3(b.b, d.b, f.b), h.b);
  const float mx4R = vmax2(vmax3(b.r, d.r, f.r), h.r), mx4G = vmax2(vmax3(b.g, d.g, f.g), h.g), mx4B = vmax2(vmax3(b.b, d.b, f.b), h.b);
  // :748-755 limiters; "these need to be high precision RCPs": the correctly rounded reciprocal when EXACT (rcp_ieee), v_rcp_f32 (1 ulp) otherwise.
  // 4*x and 4*x-4 are exact scalings, so fusing the latter does not change it (barring overflow).
  auto rcp = [](float x) { return EXACT ? rcp_ieee(x) : __builtin_amdgcn_rcpf(x); };
  const float hitMinR = vmin2(mn4R, e.r) * rcp(4.0f * mx4R);
  const float hitMinG = vmin2(mn4G, e.g) * rcp(4.0f * mx4G);
  const float hitMinB = vmin2(mn4B, e.b) * rcp(4.0f * mx4B);
  const float hitMaxR = (1.0f - vmax2(mx4R, e.r)) * rcp(fmaf(4.0f, mn4R, -4.0f));
  const float hitMaxG = (1.0f - vmax2(mx4G, e.g)) * rcp(fmaf(4.0f, mn4G, -4.0f));
  const float hitMaxB = (1.0f - vmax2(mx4B, e.b)) * rcp(fmaf(4.0f, mn4B, -4.0f));
  // :756-759  max() must return the non-NaN operand (0*inf on black pixels): v_max_f32 does.
  const float lobeR = fmaxf(-hitMinR, hitMaxR), lobeG = fmaxf(-hitMinG, hitMaxG), lobeB = fmaxf(-hitMinB, hitMaxB);
  float lobe = fmaxf(-(0.25f - (1.0f / 16.0f)), fminf(max3f(lobeR, lobeG, lobeB), 0.0f)) * sharp;
  if (flags & FSR1_FLAG_RCAS_DENOISE) {  // :731-739, :761-763
    const float bL = fmaf(b.b, 0.5f, fmaf(b.r, 0.5f, b.g)), dL = fmaf(d.b, 0.5f, fmaf(d.r, 0.5f, d.g));
    const float eL = fmaf(e.b, 0.5f, fmaf(e.r, 0.5f, e.g)), fL = fmaf(f.b, 0.5f, fmaf(f.r, 0.5f, f.g));
    const float hL = fmaf(h.b, 0.5f, fmaf(h.r, 0.5f, h.g));
    float nz = 0.25f * bL + 0.25f * dL + 0.25f * fL + 0.25f * hL - eL;
    nz = sat(fabsf(nz) * APrxMedRcpF1<EXACT>(max3f(max3f(bL, dL, eL), fL, hL) - min3f(min3f(bL, dL, eL), fL, hL)));
    nz = mad<EXACT>(-0.5f, nz, 1.0f);
    lobe *= nz;
  }
  // :765-768 resolve
  const float rcpL = APrxMedRcpF1<EXACT>(mad<EXACT>(4.0f, lobe, 1.0f));
  rgb_t p;
  if (EXACT) {
    p.r = (lobe * b.r + lobe * d.r + lobe * h.r + lobe * f.r + e.r) * rcpL;
    p.g = (lobe * b.g + lobe * d.g + lobe * h.g + lobe * f.g + e.g) * rcpL;
    p.b = (lobe * b.b + lobe * d.b + lobe * h.b + lobe * f.b + e.b) * rcpL;
  } else {
    p.r = fmaf(lobe, (b.r + h.r) + (d.r + f.r), e.r) * rcpL;
    p.g = fmaf(lobe, (b.g + h.g) + (d.g + f.g), e.g) * rcpL;
    p.b = fmaf(lobe, (b.b + h.b) + (d.b + f.b), e.b) * rcpL;
  }
  if (flags & FSR1_FLAG_HDR_SQUARE) { p.r *= p.r; p.g *= p.g; p.b *= p.b; }  // FSR_Pass.hlsl:92-93
  p.r = pinned(p.r); p.g = pinned(p.g); p.b = pinned(p.b);  // see easu_pixel: keep the narrowing a plain rounding
  return p;
}

}  // namespace fsr1
