// FsrEasuF building blocks (ffx-fsr/ffx_fsr1.h:315-437) shared by the EASU kernel and the fused
// EASU->RCAS kernel: footprint staging (phases 1-2) and the per-pixel filter (phase 3).
//
// MI355X cost model that shaped them (tools/ubench/ubench2.hip, measured): v_fma/v_mul/v_add_f32
// issue at ~2.4 cycles per wave64 instruction, while v_min/v_max/v_cvt/v_fma_mix and every packed
// (v_pk_*) instruction take ~4.3 and v_rcp/v_rsq ~8.5.  So: fp32 texels in LDS (no per-tap
// conversions), plain v_fma_f32 everywhere, the window clip done by the free `clamp` modifier
// instead of v_min_f32, min/max hoisted to the per-texel phase.
#pragma once
#include "fsr1_color_math.h"
#include "fsr1_device.h"

namespace fsr1 {

// LDS bytes per footprint texel: fp32 texel (R,G,B,luma) + analysis + dering bounds
constexpr int kEasuLdsPerTexel = 16 + 16 + 16;

struct EasuLds {
  float4_t* tex;  // [n] R G B luma*2
  float4_t* ana;  // [n] dirX dirY lenX^2 lenY^2 of FsrEasuSetF for the '+' around the texel
  uint4* mm;      // [n] RGBA16F only: packed binary16 min.RG min.B1 max.RG max.B1 of the 2x2 block at the texel
  int fw;         // row pitch (texels) = footprint width; arrays are dense
};

__device__ __forceinline__ EasuLds easu_lds_carve(char* smem, int capacity_texels) {
  EasuLds l;
  l.tex = reinterpret_cast<float4_t*>(smem);
  l.ana = reinterpret_cast<float4_t*>(smem + (size_t)capacity_texels * 16);
  l.mm = reinterpret_cast<uint4*>(smem + (size_t)capacity_texels * 32);
  l.fw = 0;
  return l;
}

// Phases 1 and 2 for the footprint [fx0, fx0+fw) x [fy0, fy0+fh) of input texels (unclamped
// coordinates; the sampler's clamp-to-edge, FSR_Filter.cpp:48-53, is applied while loading).
// Ends with a barrier: afterwards every thread may read any footprint entry.
// PRE: the colour prologue (FsrSrtmF, fsr1_color_math.h) is applied to every texel as it is loaded — the texels in
// LDS are then arbitrary binary32 values, so the packed binary16 dering bounds are not used (easu_resolve).
template <int FMT, bool PRE = false, bool EXACT = false>
__device__ __forceinline__ void easu_stage_footprint(const EasuLds& l, const ImageView& in, const char* in_frame, int fx0, int fy0,
                                                     int fw, int fh, int tid, const ColorArgs* color = nullptr) {
  typedef typename Pixel<FMT>::T texel_t;
  const int n = fw * fh;
  const float inv_fw = 1.0f / (float)fw;
  // ---- phase 1: HBM -> LDS, one coalesced pass, fp32 once per input texel ----
  for (int i = tid; i < n; i += kThreads) {
    const int ly = (int)(((float)i + 0.5f) * inv_fw);  // exact for the few thousand texels of a footprint
    const int lx = i - ly * fw;
    const int gy = min(max(fy0 + ly, 0), in.height - 1);
    const int gx = min(max(fx0 + lx, 0), in.width - 1);
    const texel_t px = *reinterpret_cast<const texel_t*>(in_frame + (long long)gy * in.pitch + (size_t)gx * sizeof(texel_t));
    float4_t c = Pixel<FMT>::load(px);
    if constexpr (PRE) c = color_prologue<EXACT>(*color, c);
    // :363-366  luma*2 = B*0.5 + (R*0.5 + G); the products by 0.5 are exact, so fusing them is too
    l.tex[i] = float4_t{c.x, c.y, c.z, fmaf(c.z, 0.5f, fmaf(c.x, 0.5f, c.y))};
  }
  __syncthreads();
  // ---- phase 2: per-texel terms, for the texels that are read as f/g/j/k of some pixel: columns 1..fw-2, rows
  //      1..fh-2 of the footprint (every neighbour of those lies inside it, so nothing is clamped). ----
  const int iw = fw - 2, m = iw * (fh - 2);
  const float inv_iw = 1.0f / (float)iw;
  for (int j = tid; j < m; j += kThreads) {
    const int y = (int)(((float)j + 0.5f) * inv_iw);
    const int i = (y + 1) * fw + (j - y * iw) + 1;
    const int iu = i - fw, id = i + fw, il = i - 1, ir = i + 1;
    const float4_t tc = l.tex[i], tr = l.tex[ir], td = l.tex[id];
    // FsrEasuSetF :295-313 — reference order, no contraction
    const float lA = l.tex[iu].w, lB = l.tex[il].w, lC = tc.w, lD = tr.w, lE = td.w;
    const float dc = lD - lC, cb = lC - lB;
    float lenX = APrxLoRcpF1(fmaxf(fabsf(dc), fabsf(cb)));
    const float dirX = lD - lB;
    lenX = sat(fabsf(dirX) * lenX);
    lenX *= lenX;
    const float ec = lE - lC, ca = lC - lA;
    float lenY = APrxLoRcpF1(fmaxf(fabsf(ec), fabsf(ca)));
    const float dirY = lE - lA;
    lenY = sat(fabsf(dirY) * lenY);
    lenY *= lenY;
    l.ana[i] = float4_t{dirX, dirY, lenX, lenY};
    if (FMT == FSR1_FORMAT_RGBA16F && !PRE) {
      // :416-419 min/max over the 2x2 block whose top-left texel is i (f g / j k).  The texels are binary16
      // values, so their min/max are too: keep them packed and clamp after the final rounding (rounding is
      // monotone, the bounds are representable, so both orders give the same binary16).  .w = 1 forces alpha.
      const float4_t tdr = l.tex[id + 1];
      const float mnR = fminf(min3f(tc.x, tr.x, td.x), tdr.x), mxR = fmaxf(max3f(tc.x, tr.x, td.x), tdr.x);
      const float mnG = fminf(min3f(tc.y, tr.y, td.y), tdr.y), mxG = fmaxf(max3f(tc.y, tr.y, td.y), tdr.y);
      const float mnB = fminf(min3f(tc.z, tr.z, td.z), tdr.z), mxB = fmaxf(max3f(tc.z, tr.z, td.z), tdr.z);
      const half2_t a0 = {(half_t)mnR, (half_t)mnG}, a1 = {(half_t)mnB, (half_t)1.0f};
      const half2_t b0 = {(half_t)mxR, (half_t)mxG}, b1 = {(half_t)mxB, (half_t)1.0f};
      l.mm[i] = uint4{__builtin_bit_cast(uint32_t, a0), __builtin_bit_cast(uint32_t, a1), __builtin_bit_cast(uint32_t, b0),
                      __builtin_bit_cast(uint32_t, b1)};
    }
  }
  __syncthreads();
}

struct rgbf_t { float r, g, b; };

// FsrEasuF for one output pixel whose 'f' texel sits at footprint index f_idx and whose sub-texel
// position is (ppx, ppy) (:324-326 done by the caller).  Returns aC * rcp(aW) (:437 before the
// dering clamp).  Everything up to the `dirR < 1/32768` decision is evaluated in the reference's exact
// operation order: that decision (and floor() in the caller) are the filter's only discontinuities.
template <bool EXACT>
__device__ __forceinline__ rgbf_t easu_pixel(const EasuLds& l, int f_idx, float ppx, float ppy) {
  const int fw = l.fw;
  const float omx = 1.0f - ppx, omy = 1.0f - ppy;
  // :381-386 bilinear accumulation of the 4 analyses (f,g,j,k), reference order:
  //   dir.x += dirX*w ; len += lenX*w ; dir.y += dirY*w ; len += lenY*w   for s,t,u,v in turn.
  const float4_t af = l.ana[f_idx], ag = l.ana[f_idx + 1], aj = l.ana[f_idx + fw], ak = l.ana[f_idx + fw + 1];
  const float wS = omx * omy, wT = ppx * omy, wU = omx * ppy, wV = ppx * ppy;
  float dirx = af.x * wS;  // 0 + x is exact, so the first add of each chain is dropped
  float diry = af.y * wS;
  dirx += ag.x * wT; diry += ag.y * wT;
  dirx += aj.x * wU; diry += aj.y * wU;
  dirx += ak.x * wV; diry += ak.y * wV;
  float len = af.z * wS;
  // len does not feed the zero test, so outside EXACT its multiply-adds are fused
  len = mad<EXACT>(af.w, wS, len);
  len = mad<EXACT>(ag.z, wT, len); len = mad<EXACT>(ag.w, wT, len);
  len = mad<EXACT>(aj.z, wU, len); len = mad<EXACT>(aj.w, wU, len);
  len = mad<EXACT>(ak.z, wV, len); len = mad<EXACT>(ak.w, wV, len);

  // :389-395 normalise; the zero test is the filter's only branch-like discontinuity
  const float dir2x = dirx * dirx, dir2y = diry * diry;
  float dirR = dir2x + dir2y;
  const bool zro = dirR < (1.0f / 32768.0f);
  dirR = zro ? 1.0f : APrxLoRsqF1(dirR);
  dirx = zro ? 1.0f : dirx;
  dirx *= dirR;
  diry *= dirR;
  // :397-409 kernel shape
  len = len * 0.5f;
  len *= len;
  const float stretch = mad<EXACT>(dirx, dirx, diry * diry) * APrxLoRcpF1(fmaxf(fabsf(dirx), fabsf(diry)));
  const float len2x = mad<EXACT>(stretch - 1.0f, len, 1.0f);
  const float len2y = mad<EXACT>(-0.5f, len, 1.0f);
  const float lob = mad<EXACT>((float)((1.0 / 4.0 - 0.04) - 0.5), len, 0.5f);
  const float clp = APrxLoRcpF1(lob);

  // :421-434 12 taps.  aC += c*w ; aW += w
  float aR = 0.f, aG = 0.f, aB = 0.f, aW = 0.f;
  const float oxm = -1.0f - ppx, ox0 = 0.0f - ppx, ox1 = 1.0f - ppx, ox2 = 2.0f - ppx;
  const float oym = -1.0f - ppy, oy0 = 0.0f - ppy, oy1 = 1.0f - ppy, oy2 = 2.0f - ppy;
  if (EXACT) {
    auto tap = [&](int dx, int dy, float offx, float offy) {
      const float4_t c = l.tex[f_idx + dy * fw + dx];
      float vx = (offx * dirx) + (offy * diry);
      float vy = (offx * (-diry)) + (offy * dirx);
      vx *= len2x;
      vy *= len2y;
      float d2 = vx * vx + vy * vy;
      d2 = fminf(d2, clp);
      float wB = (float)(2.0 / 5.0) * d2 + -1.0f;
      float wA = lob * d2 + -1.0f;
      wB *= wB;
      wA *= wA;
      wB = (float)(25.0 / 16.0) * wB + (float)(-(25.0 / 16.0 - 1.0));
      const float w = wB * wA;
      aR += c.x * w; aG += c.y * w; aB += c.z * w;
      aW += w;
    };
    // reference order: b c i j f e k l h g o n
    tap(0, -1, ox0, oym); tap(1, -1, ox1, oym); tap(-1, 1, oxm, oy1); tap(0, 1, ox0, oy1);
    tap(0, 0, ox0, oy0); tap(-1, 0, oxm, oy0); tap(1, 1, ox1, oy1); tap(2, 1, ox2, oy1);
    tap(2, 0, ox2, oy0); tap(1, 0, ox1, oy0); tap(1, 2, ox1, oy2); tap(0, 2, ox0, oy2);
  } else {
    // Re-formulated taps (continuous part of the filter; ~1e-6 relative from the reference order):
    //   v = M*off with M = [dir.x*len.x dir.y*len.x ; -dir.y*len.y dir.x*len.y]          (:250-253)
    //   u = min(|v|^2, clp)/clp = sat(off^T Q off),  Q = M^T M / clp  -> the clip is the fma's clamp bit
    //   off^T Q off = ox*(q00*ox + 2*q01*oy) + q11*oy^2      (per-row terms s = 2*q01*oy, b = q11*oy^2)
    //   base = 25/16*(2/5*d2-1)^2 - 9/16 = 1/4*d2^2 - 5/4*d2 + 1,  window = (lob*d2-1)^2,  d2 = clp*u
    const float rclp = __builtin_amdgcn_rcpf(clp);
    const float sx = len2x * len2x * rclp, sy = len2y * len2y * rclp;
    const float dxx = dirx * dirx, dyy = diry * diry, dxy2 = 2.0f * (dirx * diry);
    const float q00 = fmaf(dxx, sx, dyy * sy), q11 = fmaf(dyy, sx, dxx * sy), q01 = dxy2 * (sx - sy);
    const float sm = q01 * oym, s0 = q01 * oy0, s1 = q01 * oy1, s2 = q01 * oy2;
    const float bm = q11 * (oym * oym), b0 = q11 * (oy0 * oy0), b1 = q11 * (oy1 * oy1), b2 = q11 * (oy2 * oy2);
    const float k2 = 0.25f * clp * clp, k1 = -1.25f * clp, k3 = lob * clp;
    auto tap = [&](int dx, int dy, float ox, float s, float b) {
      const float4_t c = l.tex[f_idx + dy * fw + dx];
      const float u = sat(fmaf(ox, fmaf(q00, ox, s), b));
      const float base = fmaf(fmaf(k2, u, k1), u, 1.0f);
      const float wa = fmaf(k3, u, -1.0f);
      const float w = base * (wa * wa);
      aR = fmaf(c.x, w, aR); aG = fmaf(c.y, w, aG); aB = fmaf(c.z, w, aB);
      aW += w;
    };
    tap(0, -1, ox0, sm, bm); tap(1, -1, ox1, sm, bm);
    tap(-1, 0, oxm, s0, b0); tap(0, 0, ox0, s0, b0); tap(1, 0, ox1, s0, b0); tap(2, 0, ox2, s0, b0);
    tap(-1, 1, oxm, s1, b1); tap(0, 1, ox0, s1, b1); tap(1, 1, ox1, s1, b1); tap(2, 1, ox2, s1, b1);
    tap(0, 2, ox0, s2, b2); tap(1, 2, ox1, s2, b2);
  }
  // :437 normalise (dering clamp is applied by the caller, in the storage format)
  const float rW = EXACT ? 1.0f / aW : __builtin_amdgcn_rcpf(aW);
  // pinned in every variant: the narrowing that follows must round the binary32 product, not re-fuse it
  // (v_fma_mixlo_f16), or two kernels sharing this code could round the same pixel differently
  return rgbf_t{pinned(aR * rW), pinned(aG * rW), pinned(aB * rW)};
}

// Dering clamp (:416-419, :437) + alpha = 1 (FSR_Pass.hlsl:80) + optional `c *= c` (FSR_Pass.hlsl:78-79),
// producing the pixel in its storage format.
// Dering clamp in binary32 (:416-419, :437) + optional `c *= c`: the filter's result before the store conversion.
__device__ __forceinline__ rgbf_t easu_resolve_f(const EasuLds& l, int f_idx, rgbf_t p, bool hdr_square) {
  const int fw = l.fw;
  const float4_t cf = l.tex[f_idx], cg = l.tex[f_idx + 1], cj = l.tex[f_idx + fw], ck = l.tex[f_idx + fw + 1];
  float pr = fminf(fmaxf(max3f(cf.x, cg.x, cj.x), ck.x), fmaxf(fminf(min3f(cf.x, cg.x, cj.x), ck.x), p.r));
  float pg = fminf(fmaxf(max3f(cf.y, cg.y, cj.y), ck.y), fmaxf(fminf(min3f(cf.y, cg.y, cj.y), ck.y), p.g));
  float pb = fminf(fmaxf(max3f(cf.z, cg.z, cj.z), ck.z), fmaxf(fminf(min3f(cf.z, cg.z, cj.z), ck.z), p.b));
  if (hdr_square) { pr *= pr; pg *= pg; pb *= pb; }
  return rgbf_t{pinned(pr), pinned(pg), pinned(pb)};
}

template <int FMT, bool EXACT, bool PRE = false>
__device__ __forceinline__ typename Pixel<FMT>::T easu_resolve(const EasuLds& l, int f_idx, rgbf_t p, bool hdr_square) {
  typedef typename Pixel<FMT>::T texel_t;
  if constexpr (FMT == FSR1_FORMAT_RGBA16F && !PRE) if (!hdr_square) {
    const uint4 mm = l.mm[f_idx];
    half2_t rg = __builtin_convertvector(float2_t{p.r, p.g}, half2_t);  // v_cvt_pk_f16_f32, RTNE
    half2_t b1 = __builtin_convertvector(float2_t{p.b, 1.0f}, half2_t);
    // v_pk_max_f16 / v_pk_min_f16 written out: through the builtins the compiler first canonicalises the four bounds
    // it loaded from LDS (v_pk_max_f16 x, x, x each — it cannot know they were produced by a conversion), which
    // doubles the half-rate instructions of this clamp; the instructions themselves are IEEE maxNum/minNum.
    uint32_t rgu = __builtin_bit_cast(uint32_t, rg), b1u = __builtin_bit_cast(uint32_t, b1);
    asm("v_pk_max_f16 %0, %1, %0\n\tv_pk_min_f16 %0, %2, %0" : "+v"(rgu) : "v"(mm.x), "v"(mm.z));
    asm("v_pk_max_f16 %0, %1, %0\n\tv_pk_min_f16 %0, %2, %0" : "+v"(b1u) : "v"(mm.y), "v"(mm.w));
    const uint2 packed = {rgu, b1u};
    return __builtin_bit_cast(texel_t, packed);
  }
  const rgbf_t q = easu_resolve_f(l, f_idx, p, hdr_square);
  return Pixel<FMT>::store(q.r, q.g, q.b, 1.0f);
}

}  // namespace fsr1
