// Device-side, source-level operator surface of the FSR 1.0 HIP library (gfx950): the reference's per-pixel entry
// points, callable from a user's own HIP kernel.
//
// The reference's plugin API is a header included into the integrator's shader after the integrator has defined a few
// load callbacks (ffx-fsr/ffx_fsr1.h:232-236, :445-449, :679-682, :777-780; includer side sample/src/DX12/FSR_Pass.hlsl:28-66).
// This header is that surface for HIP.  The entry points keep the reference's spellings and argument order; the
// callbacks, which a shading language takes as free functions, are the members of a functor passed last:
//
//     struct MyEasu {                                    // ffx_fsr1.h:234-236 (F) / :447-449 (H)
//       __device__ float4 FsrEasuRF(float2 p) const;     // gather4 of one channel at normalised coordinate p
//       __device__ float4 FsrEasuGF(float2 p) const;     //   .x=(i,j+1) .y=(i+1,j+1) .z=(i+1,j) .w=(i,j)
//       __device__ float4 FsrEasuBF(float2 p) const;     //   i=floor(p.x*W-0.5), j=floor(p.y*H-0.5)
//     };
//     struct MyRcas {                                    // ffx_fsr1.h:681-682 (F) / :779-780 (H)
//       __device__ float4 FsrRcasLoadF(int2 p) const;    // integer texel fetch, no filtering
//       __device__ void FsrRcasInputF(float& r, float& g, float& b) const;   // optional per-tap colour transform
//     };
//
//     fsr1::FsrEasuF<EXACT>(pix, ip, con0, con1, con2, con3, MyEasu{...});                  // ffx_fsr1.h:315-322
//     fsr1::FsrRcasF<EXACT>(r, g, b, ip, con, MyRcas{...});                                 // ffx_fsr1.h:684-690
//     fsr1::FsrRcasF<EXACT>(r, g, b, a, ip, con, MyRcas{...});                              //   FSR_RCAS_PASSTHROUGH_ALPHA
//     fsr1::FsrEasuH(pixh, ip, con0, con1, con2, con3, MyEasuH{...});                       // ffx_fsr1.h:505-512
//     fsr1::FsrRcasH(rh, gh, bh, ip, con, MyRcasH{...});                                    // ffx_fsr1.h:782-790
//     fsr1::FsrRcasHx2(r2, g2, b2, ip, con, MyRcasHx2{...});                                // ffx_fsr1.h:889-984: pixels ip and ip + (8, 0)
//     fsr1::FsrRcasDepackHx2(pix0, pix1, r2, g2, b2);                                       // ffx_fsr1.h:880-887
//     fsr1::ARmp8x8(lane)                                                                   // ffx_a.h:2304: the 64 -> 8x8 remap of mainCS
//
// con0..con3 / con are the words FsrEasuCon / FsrRcasCon (include/fsr1_hip.h, bit-exact with the reference) produce.
// EXACT = true follows the reference's operation order and is bit-identical to the reference's FsrEasuF / FsrRcasF
// evaluated on the CPU; EXACT = false (default) re-associates the continuous part of the filters (<= 1 binary16 ULP).
// The H entry points are always bit-exact against the reference's H path (one native binary16 operation per
// reference operation).  Compile with -ffp-contract=off (see fsr1_device_base.hpp).
//
// These per-pixel forms are the drop-in shape; they gather 12 x 3 channels per pixel like the reference shader does.
// The FAST form — what libfsr1_hip.so itself runs — stages a tile's input footprint in LDS once and filters from there:
// fsr1_device_easu.hpp (easu_stage_footprint / easu_pixel / easu_bounds / easu_resolve), fsr1_device_rcas.hpp
// (rcas_pixel) and fsr1_device_half.hpp, all included below; tests/device_api/ holds a kernel written against them.
#pragma once
#include "fsr1_device_base.hpp"
#include "fsr1_device_color.hpp"
#include "fsr1_device_easu.hpp"
#include "fsr1_device_half.hpp"
#include "fsr1_device_rcas.hpp"

namespace fsr1 {

typedef half_t half3_t __attribute__((ext_vector_type(3)));

namespace detail {
__device__ __forceinline__ float2 f2(uint32_t x, uint32_t y) { return float2{as_f32(x), as_f32(y)}; }
}  // namespace detail

// ---------------------------------------------------------------------------------------------------------------------
// FsrEasuF — ffx_fsr1.h:315-437.  pix: filtered RGB (alpha is the caller's: the sample writes 1, FSR_Pass.hlsl:80).
// ---------------------------------------------------------------------------------------------------------------------
template <bool EXACT = false, class Callbacks>
__device__ __forceinline__ void FsrEasuF(float3& pix, uint2 ip, uint4 con0, uint4 con1, uint4 con2, uint4 con3, const Callbacks& cb) {
  // :324-326 position of 'f' — product and sum rounded separately, like the reference (no contraction)
  float ppx = (float)ip.x * as_f32(con0.x) + as_f32(con0.z);
  float ppy = (float)ip.y * as_f32(con0.y) + as_f32(con0.w);
  const float fpx = floorf(ppx), fpy = floorf(ppy);
  ppx -= fpx;
  ppy -= fpy;
  // :344-348 the four gather positions
  const float2 p0 = {fpx * as_f32(con1.x) + as_f32(con1.z), fpy * as_f32(con1.y) + as_f32(con1.w)};
  const float2 p1 = {p0.x + as_f32(con2.x), p0.y + as_f32(con2.y)};
  const float2 p2 = {p0.x + as_f32(con2.z), p0.y + as_f32(con2.w)};
  const float2 p3 = {p0.x + as_f32(con3.x), p0.y + as_f32(con3.y)};
  // :349-360 — gather order .x=(i,j+1) .y=(i+1,j+1) .z=(i+1,j) .w=(i,j):  p0 -> b c . .   p1 -> i j f e   p2 -> k l h g   p3 -> . . o n
  const float4 bczzR = cb.FsrEasuRF(p0), bczzG = cb.FsrEasuGF(p0), bczzB = cb.FsrEasuBF(p0);
  const float4 ijfeR = cb.FsrEasuRF(p1), ijfeG = cb.FsrEasuGF(p1), ijfeB = cb.FsrEasuBF(p1);
  const float4 klhgR = cb.FsrEasuRF(p2), klhgG = cb.FsrEasuGF(p2), klhgB = cb.FsrEasuBF(p2);
  const float4 zzonR = cb.FsrEasuRF(p3), zzonG = cb.FsrEasuGF(p3), zzonB = cb.FsrEasuBF(p3);
  // the 4 x 4 window around 'f' (dx, dy in -1..2), corners unused; .w = luma*2 = B*0.5 + (R*0.5 + G)  (:363-366)
  auto texel = [](float r, float g, float b) { return float4_t{r, g, b, fmaf(b, 0.5f, fmaf(r, 0.5f, g))}; };
  const float4_t zero = {0.f, 0.f, 0.f, 0.f};
  const float4_t win[4][4] = {
      {zero, texel(bczzR.x, bczzG.x, bczzB.x), texel(bczzR.y, bczzG.y, bczzB.y), zero},                                          // . b c .
      {texel(ijfeR.w, ijfeG.w, ijfeB.w), texel(ijfeR.z, ijfeG.z, ijfeB.z), texel(klhgR.w, klhgG.w, klhgB.w), texel(klhgR.z, klhgG.z, klhgB.z)},  // e f g h
      {texel(ijfeR.x, ijfeG.x, ijfeB.x), texel(ijfeR.y, ijfeG.y, ijfeB.y), texel(klhgR.x, klhgG.x, klhgB.x), texel(klhgR.y, klhgG.y, klhgB.y)},  // i j k l
      {zero, texel(zzonR.w, zzonG.w, zzonB.w), texel(zzonR.z, zzonG.z, zzonB.z), zero}};                                         // . n o .
  auto tex = [&](int dx, int dy) { return win[dy + 1][dx + 1]; };
  auto lum = [&](int dx, int dy) { return win[dy + 1][dx + 1].w; };
  // :383-386 FsrEasuSetF on the '+' neighbourhoods of f, g, j, k
  auto ana = [&](int k) {
    const int x = k & 1, y = k >> 1;
    return easu_analysis<EXACT>(lum(x, y - 1), lum(x - 1, y), lum(x, y), lum(x + 1, y), lum(x, y + 1));
  };
  const rgbf_t c = easu_filter<EXACT>(tex, ana, ppx, ppy);
  // :416-419, :437 dering clamp to the 2x2 block f g / j k
  const rgbf_t q = easu_clamp<EXACT>(easu_bounds(tex(0, 0), tex(1, 0), tex(0, 1), tex(1, 1)), c, false);
  pix = float3{q.r, q.g, q.b};
}

// ---------------------------------------------------------------------------------------------------------------------
// FsrRcasF — ffx_fsr1.h:684-769.  DENOISE = FSR_RCAS_DENOISE (:647-651, :761-763).
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
template <bool EXACT, bool DENOISE, class Callbacks>
__device__ __forceinline__ rgb_t rcas_f(float* pixA, uint2 ip, uint4 con, const Callbacks& cb) {
  const int2 sp = {(int)ip.x, (int)ip.y};
  auto tap = [&](int dx, int dy, float* alpha) {
    const float4 t = cb.FsrRcasLoadF(int2{sp.x + dx, sp.y + dy});  // :697-707
    rgb_t c = {t.x, t.y, t.z};
    if (alpha) *alpha = t.w;
    cb.FsrRcasInputF(c.r, c.g, c.b);                                // :725-729
    return c;
  };
  const rgb_t b = tap(0, -1, nullptr), d = tap(-1, 0, nullptr), e = tap(0, 0, pixA), f = tap(1, 0, nullptr), h = tap(0, 1, nullptr);
  return rcas_pixel<EXACT>(b, d, e, f, h, as_f32(con.x), DENOISE ? (uint32_t)FSR1_FLAG_RCAS_DENOISE : 0u);
}
}  // namespace detail

template <bool EXACT = false, bool DENOISE = false, class Callbacks>
__device__ __forceinline__ void FsrRcasF(float& pixR, float& pixG, float& pixB, uint2 ip, uint4 con, const Callbacks& cb) {
  const rgb_t p = detail::rcas_f<EXACT, DENOISE>(nullptr, ip, con, cb);
  pixR = p.r; pixG = p.g; pixB = p.b;
}
// FSR_RCAS_PASSTHROUGH_ALPHA (:688-690, :700-702): pixA = the centre tap's alpha
template <bool EXACT = false, bool DENOISE = false, class Callbacks>
__device__ __forceinline__ void FsrRcasF(float& pixR, float& pixG, float& pixB, float& pixA, uint2 ip, uint4 con, const Callbacks& cb) {
  const rgb_t p = detail::rcas_f<EXACT, DENOISE>(&pixA, ip, con, cb);
  pixR = p.r; pixG = p.g; pixB = p.b;
}

// ---------------------------------------------------------------------------------------------------------------------
// FsrEasuH — ffx_fsr1.h:505-593.  Callbacks: half4 FsrEasuRH / GH / BH(float2 p) (:447-449).
// ---------------------------------------------------------------------------------------------------------------------
template <class Callbacks>
__device__ __forceinline__ void FsrEasuH(half3_t& pix, uint2 ip, uint4 con0, uint4 con1, uint4 con2, uint4 con3, const Callbacks& cb) {
  float ppx = (float)ip.x * as_f32(con0.x) + as_f32(con0.z);  // :513-515 (position arithmetic stays binary32)
  float ppy = (float)ip.y * as_f32(con0.y) + as_f32(con0.w);
  const float fpx = floorf(ppx), fpy = floorf(ppy);
  ppx -= fpx;
  ppy -= fpy;
  const half2_t ppp = h2((half_t)ppx, (half_t)ppy);  // :516
  const float2 p0 = {fpx * as_f32(con1.x) + as_f32(con1.z), fpy * as_f32(con1.y) + as_f32(con1.w)};
  const float2 p1 = {p0.x + as_f32(con2.x), p0.y + as_f32(con2.y)};
  const float2 p2 = {p0.x + as_f32(con2.z), p0.y + as_f32(con2.w)};
  const float2 p3 = {p0.x + as_f32(con3.x), p0.y + as_f32(con3.y)};
  const half4_t bczzR = cb.FsrEasuRH(p0), bczzG = cb.FsrEasuGH(p0), bczzB = cb.FsrEasuBH(p0);
  const half4_t ijfeR = cb.FsrEasuRH(p1), ijfeG = cb.FsrEasuGH(p1), ijfeB = cb.FsrEasuBH(p1);
  const half4_t klhgR = cb.FsrEasuRH(p2), klhgG = cb.FsrEasuGH(p2), klhgB = cb.FsrEasuBH(p2);
  const half4_t zzonR = cb.FsrEasuRH(p3), zzonG = cb.FsrEasuGH(p3), zzonB = cb.FsrEasuBH(p3);
  const half_t hlf = (half_t)0.5f;
  auto luma = [&](half_t r, half_t g, half_t b) { return (half_t)(b * hlf + (r * hlf + g)); };  // :535-538
  const half_t bL = luma(bczzR.x, bczzG.x, bczzB.x), cL = luma(bczzR.y, bczzG.y, bczzB.y);
  const half_t iL = luma(ijfeR.x, ijfeG.x, ijfeB.x), jL = luma(ijfeR.y, ijfeG.y, ijfeB.y), fL = luma(ijfeR.z, ijfeG.z, ijfeB.z), eL = luma(ijfeR.w, ijfeG.w, ijfeB.w);
  const half_t kL = luma(klhgR.x, klhgG.x, klhgB.x), lL = luma(klhgR.y, klhgG.y, klhgB.y), hL = luma(klhgR.z, klhgG.z, klhgB.z), gL = luma(klhgR.w, klhgG.w, klhgB.w);
  const half_t oL = luma(zzonR.z, zzonG.z, zzonB.z), nL = luma(zzonR.w, zzonG.w, zzonB.w);
  // :552-558 — FsrEasuSetH's lanes are the positions f, g (first call) and j, k (second call)
  const half4_t anas[4] = {easu_analysis_h(bL, eL, fL, gL, jL), easu_analysis_h(cL, fL, gL, hL, kL), easu_analysis_h(fL, iL, jL, kL, nL),
                           easu_analysis_h(gL, jL, kL, lL, oL)};
  const EasuHPair pairs[6] = {{h2(bczzR.x, bczzR.y), h2(bczzG.x, bczzG.y), h2(bczzB.x, bczzB.y)},   // bczz.xy
                              {h2(ijfeR.x, ijfeR.y), h2(ijfeG.x, ijfeG.y), h2(ijfeB.x, ijfeB.y)},   // ijfe.xy
                              {h2(ijfeR.z, ijfeR.w), h2(ijfeG.z, ijfeG.w), h2(ijfeB.z, ijfeB.w)},   // ijfe.zw
                              {h2(klhgR.x, klhgR.y), h2(klhgG.x, klhgG.y), h2(klhgB.x, klhgB.y)},   // klhg.xy
                              {h2(klhgR.z, klhgR.w), h2(klhgG.z, klhgG.w), h2(klhgB.z, klhgB.w)},   // klhg.zw
                              {h2(zzonR.z, zzonR.w), h2(zzonG.z, zzonG.w), h2(zzonB.z, zzonB.w)}};  // zzon.zw
  // :575-577 operand order of the reference: max(max(f, g), max(j, k)) on the (-x, x) pairs
  const half2_t boths[3] = {easu_both_h(ijfeR.z, klhgR.w, ijfeR.y, klhgR.x), easu_both_h(ijfeG.z, klhgG.w, ijfeG.y, klhgG.x),
                            easu_both_h(ijfeB.z, klhgB.w, ijfeB.y, klhgB.x)};
  const rgbh_t c = easu_filter_h([&](int k) { return anas[k]; }, [&](int i) { return pairs[i]; }, [&](int ch) { return boths[ch]; }, ppp);
  pix = half3_t{c.r, c.g, c.b};
}

// ---------------------------------------------------------------------------------------------------------------------
// FsrRcasH — ffx_fsr1.h:782-866.  Callbacks: half4 FsrRcasLoadH(short2 p), void FsrRcasInputH(half&, half&, half&) (:779-780).
// One pixel, scalar binary16 operations (rcas_pixel_h1): equal, bit for bit, to the matching lane of the two-pixel form.
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
template <bool DENOISE, class Callbacks>
__device__ __forceinline__ rgbh1_t rcas_h(half_t* pixA, uint2 ip, uint4 con, const Callbacks& cb) {
  const short2 sp = {(short)ip.x, (short)ip.y};  // :795 ASW2(ip)
  struct tap_t { half_t r, g, b; };
  auto tap = [&](int dx, int dy, half_t* alpha) {
    const half4_t t = cb.FsrRcasLoadH(short2{(short)(sp.x + dx), (short)(sp.y + dy)});
    half_t r = t.x, g = t.y, b = t.z;
    if (alpha) *alpha = t.w;
    cb.FsrRcasInputH(r, g, b);
    return tap_t{r, g, b};
  };
  const tap_t b = tap(0, -1, nullptr), d = tap(-1, 0, nullptr), e = tap(0, 0, pixA), f = tap(1, 0, nullptr), h = tap(0, 1, nullptr);
  const half_t sharp1 = __builtin_bit_cast(half_t, (u16)(con.y & 0xffffu));  // :857 AH2_AU1(con.y).x
  return rcas_pixel_h1(b.r, b.g, b.b, d.r, d.g, d.b, e.r, e.g, e.b, f.r, f.g, f.b, h.r, h.g, h.b, sharp1, DENOISE ? (uint32_t)FSR1_FLAG_RCAS_DENOISE : 0u);
}
}  // namespace detail

template <bool DENOISE = false, class Callbacks>
__device__ __forceinline__ void FsrRcasH(half_t& pixR, half_t& pixG, half_t& pixB, uint2 ip, uint4 con, const Callbacks& cb) {
  const rgbh1_t p = detail::rcas_h<DENOISE>(nullptr, ip, con, cb);
  pixR = p.r; pixG = p.g; pixB = p.b;
}
template <bool DENOISE = false, class Callbacks>
__device__ __forceinline__ void FsrRcasH(half_t& pixR, half_t& pixG, half_t& pixB, half_t& pixA, uint2 ip, uint4 con, const Callbacks& cb) {
  const rgbh1_t p = detail::rcas_h<DENOISE>(&pixA, ip, con, cb);
  pixR = p.r; pixG = p.g; pixB = p.b;
}

// ---------------------------------------------------------------------------------------------------------------------
// FsrRcasHx2 — ffx_fsr1.h:889-984 (FSR_RCAS_HX2): TWO pixels per call, `ip` and `ip + (8, 0)` — the left and the right 8x8 tile
// of a 16x8 region — in the .x / .y halves of every operand.  Callbacks (:876-877):
//   half4 FsrRcasLoadHx2(short2 p);  void FsrRcasInputHx2(half2& r, half2& g, half2& b);
// FsrRcasDepackHx2 (:880-887) turns the packed result back into two pixels for the store.
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
template <bool DENOISE, class Callbacks>
__device__ __forceinline__ rgbh2_t rcas_hx2(half2_t* pixA, uint2 ip, uint4 con, const Callbacks& cb) {
  const short2 sp0 = {(short)ip.x, (short)ip.y};            // :903 ASW2(ip)
  const short2 sp1 = {(short)(sp0.x + 8), sp0.y};           // :915
  struct tap_t { half2_t r, g, b; };
  auto tap = [&](int dx, int dy, half2_t* alpha) {           // :904-939 loads + AoS -> SoA
    const half4_t t0 = cb.FsrRcasLoadHx2(short2{(short)(sp0.x + dx), (short)(sp0.y + dy)});
    const half4_t t1 = cb.FsrRcasLoadHx2(short2{(short)(sp1.x + dx), (short)(sp1.y + dy)});
    if (alpha) *alpha = h2(t0.w, t1.w);
    return tap_t{h2(t0.x, t1.x), h2(t0.y, t1.y), h2(t0.z, t1.z)};
  };
  tap_t b = tap(0, -1, nullptr), d = tap(-1, 0, nullptr), e = tap(0, 0, pixA), f = tap(1, 0, nullptr), h = tap(0, 1, nullptr);
  cb.FsrRcasInputHx2(b.r, b.g, b.b);                         // :941-945, in the reference's order
  cb.FsrRcasInputHx2(d.r, d.g, d.b);
  cb.FsrRcasInputHx2(e.r, e.g, e.b);
  cb.FsrRcasInputHx2(f.r, f.g, f.b);
  cb.FsrRcasInputHx2(h.r, h.g, h.b);
  const half_t sharp1 = __builtin_bit_cast(half_t, (u16)(con.y & 0xffffu));  // :975 AH2_AU1(con.y).x
  return rcas_pixel_h2(b.r, b.g, b.b, d.r, d.g, d.b, e.r, e.g, e.b, f.r, f.g, f.b, h.r, h.g, h.b, sharp1, DENOISE ? (uint32_t)FSR1_FLAG_RCAS_DENOISE : 0u);
}
}  // namespace detail

template <bool DENOISE = false, class Callbacks>
__device__ __forceinline__ void FsrRcasHx2(half2_t& pixR, half2_t& pixG, half2_t& pixB, uint2 ip, uint4 con, const Callbacks& cb) {
  const rgbh2_t p = detail::rcas_hx2<DENOISE>(nullptr, ip, con, cb);
  pixR = p.r; pixG = p.g; pixB = p.b;
}
// FSR_RCAS_PASSTHROUGH_ALPHA (:897-899, :907-909, :919-921): pixA = the two centre taps' alpha
template <bool DENOISE = false, class Callbacks>
__device__ __forceinline__ void FsrRcasHx2(half2_t& pixR, half2_t& pixG, half2_t& pixB, half2_t& pixA, uint2 ip, uint4 con, const Callbacks& cb) {
  const rgbh2_t p = detail::rcas_hx2<DENOISE>(&pixA, ip, con, cb);
  pixR = p.r; pixG = p.g; pixB = p.b;
}
// :880-887 — packed structure-of-arrays back to two pixels; alpha = 0 like the reference's A_HLSL branch (its GLSL branch leaves
// it unwritten): the caller writes 1 or the passed-through alpha.
__device__ __forceinline__ void FsrRcasDepackHx2(half4_t& pix0, half4_t& pix1, half2_t pixR, half2_t pixG, half2_t pixB) {
  pix0 = half4_t{pixR.x, pixG.x, pixB.x, (half_t)0.0f};
  pix1 = half4_t{pixR.y, pixG.y, pixB.y, (half_t)0.0f};
}

// ---------------------------------------------------------------------------------------------------------------------
// ARmp8x8 — ffx_a.h:2304: "simple remap 64x1 to 8x8 with rotated 2x2 pixel quads in quad linear" — lane bits 543210 -> x = bits 3..1,
// y = bits 5,4 and 0.  The dispatch contract of the reference's shader shell (FSR_Pass.hlsl:110-117: a 64-lane workgroup covers
// 16 x 16 pixels as four 8x8 tiles, gxy = ARmp8x8(lane) + 16 * workgroup).  The library's own kernels use tile shapes of their own.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ uint2 ARmp8x8(uint32_t a) {
  return uint2{(a >> 1) & 7u,                       // ABfe(a, 1, 3)
               (((a >> 3) & 7u) & ~1u) | (a & 1u)};  // ABfiM(ABfe(a, 3, 3), a, 1): bit 0 of a inserted under bits 5..4
}

// ---------------------------------------------------------------------------------------------------------------------
// Ready-made callbacks over a pitch-linear image (what the sample's callbacks do with a texture and a CLAMP sampler,
// FSR_Pass.hlsl:39-46, :55-62): gather4 with clamp-to-edge for EASU, integer load with out-of-bounds = 0 for RCAS.
// ---------------------------------------------------------------------------------------------------------------------
template <int FMT>
struct ImageCallbacks {
  const char* base;  // frame already selected
  int width, height;
  long long pitch;

  __device__ __forceinline__ float4_t texel(int x, int y) const {
    typedef typename Pixel<FMT>::T texel_t;
    return Pixel<FMT>::load(*reinterpret_cast<const texel_t*>(base + (long long)y * pitch + (long long)x * (long long)sizeof(texel_t)));
  }
  // gather4 corner (i, j) of normalised coordinate p; every fetched coordinate is clamped to the image (CLAMP sampler)
  template <int CH>
  __device__ __forceinline__ float4 gather(float2 p) const {
    const int i = (int)floorf(p.x * (float)width - 0.5f), j = (int)floorf(p.y * (float)height - 0.5f);
    const int x0 = min(max(i, 0), width - 1), x1 = min(max(i + 1, 0), width - 1);
    const int y0 = min(max(j, 0), height - 1), y1 = min(max(j + 1, 0), height - 1);
    return float4{texel(x0, y1)[CH], texel(x1, y1)[CH], texel(x1, y0)[CH], texel(x0, y0)[CH]};
  }
  __device__ __forceinline__ float4 FsrEasuRF(float2 p) const { return gather<0>(p); }
  __device__ __forceinline__ float4 FsrEasuGF(float2 p) const { return gather<1>(p); }
  __device__ __forceinline__ float4 FsrEasuBF(float2 p) const { return gather<2>(p); }
  __device__ __forceinline__ half4_t FsrEasuRH(float2 p) const { const float4 g = gather<0>(p); return half4_t{(half_t)g.x, (half_t)g.y, (half_t)g.z, (half_t)g.w}; }
  __device__ __forceinline__ half4_t FsrEasuGH(float2 p) const { const float4 g = gather<1>(p); return half4_t{(half_t)g.x, (half_t)g.y, (half_t)g.z, (half_t)g.w}; }
  __device__ __forceinline__ half4_t FsrEasuBH(float2 p) const { const float4 g = gather<2>(p); return half4_t{(half_t)g.x, (half_t)g.y, (half_t)g.z, (half_t)g.w}; }

  __device__ __forceinline__ float4 FsrRcasLoadF(int2 p) const {
    if (p.x < 0 || p.y < 0 || p.x >= width || p.y >= height) return float4{0.f, 0.f, 0.f, 0.f};  // D3D `Load`: out of bounds reads 0
    const float4_t t = texel(p.x, p.y);
    return float4{t.x, t.y, t.z, t.w};
  }
  __device__ __forceinline__ void FsrRcasInputF(float&, float&, float&) const {}  // the sample's is empty (FSR_Pass.hlsl:46)
  __device__ __forceinline__ half4_t FsrRcasLoadH(short2 p) const {
    const float4 t = FsrRcasLoadF(int2{p.x, p.y});
    return half4_t{(half_t)t.x, (half_t)t.y, (half_t)t.z, (half_t)t.w};
  }
  __device__ __forceinline__ void FsrRcasInputH(half_t&, half_t&, half_t&) const {}
  __device__ __forceinline__ half4_t FsrRcasLoadHx2(short2 p) const { return FsrRcasLoadH(p); }
  __device__ __forceinline__ void FsrRcasInputHx2(half2_t&, half2_t&, half2_t&) const {}
};

}  // namespace fsr1
