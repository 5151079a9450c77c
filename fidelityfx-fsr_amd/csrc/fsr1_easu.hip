// EASU — edge adaptive spatial upsampling (FsrEasuF, ffx-fsr/ffx_fsr1.h:315-437) for gfx950.
//
// One 256-thread workgroup produces a 64x16 output tile:
//   phase 1  the tile's input footprint (every texel any of its 12-tap windows can touch, with the
//            sampler's clamp-to-edge already applied) is read from HBM once, coalesced, converted to
//            fp32 once per *input* texel and parked in LDS as (R,G,B,luma);
//   phase 2  everything FsrEasuF recomputes per output pixel but that depends on the input only is
//            evaluated once per footprint texel: the direction/length terms of FsrEasuSetF
//            (:295-313: dirX, dirY, lenX^2, lenY^2 of the '+' neighbourhood) and the min/max of the
//            2x2 block used by the dering clamp (:416-419);
//   phase 3  each lane walks 4 output pixels of its column: bilinear accumulation of the analysis
//            in the reference's order, kernel shaping, 12 taps from LDS, dering clamp, one
//            row-contiguous store per wave.
//
// MI355X cost model that shaped phase 3 (tools/ubench/ubench2.hip, measured): v_fma/v_mul/v_add_f32
// issue at ~2.4 cycles per wave64 instruction, while v_min/v_max/v_cvt/v_fma_mix and every packed
// (v_pk_*) instruction take ~4.3 and v_rcp/v_rsq ~8.5.  So: fp32 texels in LDS (no per-tap
// conversions), plain v_fma_f32 everywhere, the window clip done by the free `clamp` modifier
// instead of v_min_f32, min/max hoisted to phase 2.
//
// Numerics: arithmetic is fp32 (FsrEasuF), storage is the image format.  Everything up to and
// including the `dirR < 1/32768` decision is evaluated in the reference's exact operation order
// (no contraction): that decision and floor() are the only discontinuities of the filter, so
// they must see bit-identical inputs.  With EXACT the rest follows the reference order as well and
// the result is bit-identical to the CPU-evaluated FsrEasuF; without it the continuous remainder
// is re-associated (see tap_weights below), which moves the fp32 result by ~1e-6 relative.
#include "fsr1_device.h"

namespace fsr1 {

// LDS bytes per footprint texel: fp32 texel + analysis + dering min/max
constexpr int kEasuLdsPerTexel = 16 + 16 + 16;

template <int FMT, bool EXACT>
__global__ void __launch_bounds__(kThreads) easu_kernel(const EasuArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = a.fp_w * a.fp_h;
  float4_t* const lds_tex = reinterpret_cast<float4_t*>(smem);                   // [n] R G B luma
  float4_t* const lds_ana = reinterpret_cast<float4_t*>(smem + (size_t)cap * 16); // [n] dirX dirY lenX2 lenY2
  uint4* const lds_mm = reinterpret_cast<uint4*>(smem + (size_t)cap * 32);         // [n] dering bounds (see phase 2)

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int ox0 = tx * kTileW, oy0 = ty * kTileH;

  const float c0x = as_f32(a.con[0]), c0y = as_f32(a.con[1]), c0z = as_f32(a.con[2]), c0w = as_f32(a.con[3]);

  // Footprint of this tile: fp(first pixel)-1 .. fp(last pixel)+2 per axis (ffx_fsr1.h:324-342).
  // Same arithmetic as the per-pixel position below, and x -> x*c+b is monotone under rounding.
  const int oxl = min(ox0 + kTileW, a.out.width) - 1, oyl = min(oy0 + kTileH, a.out.height) - 1;
  const int fx0 = (int)floorf((float)ox0 * c0x + c0z) - 1;
  const int fy0 = (int)floorf((float)oy0 * c0y + c0w) - 1;
  const int fw = min((int)floorf((float)oxl * c0x + c0z) + 2 - fx0 + 1, a.fp_w);
  const int fh = min((int)floorf((float)oyl * c0y + c0w) + 2 - fy0 + 1, a.fp_h);
  const int n = fw * fh;  // LDS arrays are dense: index = ly*fw + lx

  const int tid = threadIdx.x;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;
  const float inv_fw = 1.0f / (float)fw;

  // ---- phase 1: footprint -> LDS (fp32), clamp-to-edge applied here (FSR_Filter.cpp:48-53) ----
  for (int i = tid; i < n; i += kThreads) {
    const int ly = (int)(((float)i + 0.5f) * inv_fw);  // exact for the few thousand texels of a footprint
    const int lx = i - ly * fw;
    const int gy = min(max(fy0 + ly, 0), a.in.height - 1);
    const int gx = min(max(fx0 + lx, 0), a.in.width - 1);
    const texel_t px = *reinterpret_cast<const texel_t*>(in_frame + (long long)gy * a.in.pitch + (size_t)gx * sizeof(texel_t));
    const float4_t c = Pixel<FMT>::load(px);
    // :363-366  luma*2 = B*0.5 + (R*0.5 + G); the products by 0.5 are exact, so fusing them is too
    lds_tex[i] = float4_t{c.x, c.y, c.z, fmaf(c.z, 0.5f, fmaf(c.x, 0.5f, c.y))};
  }
  __syncthreads();

  // ---- phase 2: per-texel terms.  Border texels read clamped neighbours and produce values
  //      nobody uses (analysis is consumed for columns 1..fw-2 / rows 1..fh-2 only). ----
  for (int i = tid; i < n; i += kThreads) {
    const int iu = max(i - fw, 0), id = min(i + fw, n - 1), il = max(i - 1, 0), ir = min(i + 1, n - 1);
    const float4_t tc = lds_tex[i], tr = lds_tex[ir], td = lds_tex[id];
    // FsrEasuSetF :295-313 — reference order, no contraction
    const float lA = lds_tex[iu].w, lB = lds_tex[il].w, lC = tc.w, lD = tr.w, lE = td.w;
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
    lds_ana[i] = float4_t{dirX, dirY, lenX, lenY};
    // :416-419 min/max over the 2x2 block whose top-left texel is i (f g / j k)
    const float4_t tdr = lds_tex[min(id + 1, n - 1)];
    const float mnR = fminf(min3f(tc.x, tr.x, td.x), tdr.x), mxR = fmaxf(max3f(tc.x, tr.x, td.x), tdr.x);
    const float mnG = fminf(min3f(tc.y, tr.y, td.y), tdr.y), mxG = fmaxf(max3f(tc.y, tr.y, td.y), tdr.y);
    const float mnB = fminf(min3f(tc.z, tr.z, td.z), tdr.z), mxB = fmaxf(max3f(tc.z, tr.z, td.z), tdr.z);
    if (FMT == FSR1_FORMAT_RGBA16F) {
      // texels are binary16 values, so their min/max are too: keep them packed (min.RG min.B1 max.RG max.B1)
      // and clamp after the final rounding (rounding is monotone, the bounds are representable).
      const half2_t a0 = {(half_t)mnR, (half_t)mnG}, a1 = {(half_t)mnB, (half_t)1.0f};
      const half2_t b0 = {(half_t)mxR, (half_t)mxG}, b1 = {(half_t)mxB, (half_t)1.0f};
      lds_mm[i] = uint4{__builtin_bit_cast(uint32_t, a0), __builtin_bit_cast(uint32_t, a1),
                        __builtin_bit_cast(uint32_t, b0), __builtin_bit_cast(uint32_t, b1)};
    }  // (fp32 storage: the bounds are recomputed in phase 3 from the fp32 texels)
  }
  __syncthreads();

  // ---- phase 3: output pixels ----
  const int lane = tid & 63, wave = tid >> 6;
  const int ox = ox0 + lane;
  if (ox >= a.out.width) return;
  char* const out_frame = a.out.base + (long long)frame * a.out.frame_stride;
  // :324-326 (x part, shared by this lane's 4 rows)
  float ppx = (float)ox * c0x + c0z;
  const float fpx = floorf(ppx);
  ppx -= fpx;
  const int lx = (int)fpx - fx0;  // footprint column of texel 'f'
  const float omx = 1.0f - ppx;
  // tap x offsets (-1,0,1,2)-pp.x and their squares
  const float oxm = -1.0f - ppx, ox0f = 0.0f - ppx, ox1 = 1.0f - ppx, ox2 = 2.0f - ppx;

#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int oy = oy0 + wave * 4 + r;
    if (oy >= a.out.height) break;
    float ppy = (float)oy * c0y + c0w;
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const int ly = (int)fpy - fy0;
    const int f_idx = ly * fw + lx;
    const float omy = 1.0f - ppy;

    // :381-386 bilinear accumulation of the 4 analyses (f,g,j,k), reference order:
    //   dir += dirX*w ; len += lenX*w ; dir.y += dirY*w ; len += lenY*w   for s,t,u,v in turn.
    const float4_t af = lds_ana[f_idx], ag = lds_ana[f_idx + 1], aj = lds_ana[f_idx + fw], ak = lds_ana[f_idx + fw + 1];
    const float wS = omx * omy, wT = ppx * omy, wU = omx * ppy, wV = ppx * ppy;
    float dirx = af.x * wS;  // 0 + x is exact, so the first add of each chain is dropped
    float diry = af.y * wS;
    float len = af.z * wS;
    len += af.w * wS;
    dirx += ag.x * wT; len += ag.z * wT; diry += ag.y * wT; len += ag.w * wT;
    dirx += aj.x * wU; len += aj.z * wU; diry += aj.y * wU; len += aj.w * wU;
    dirx += ak.x * wV; len += ak.z * wV; diry += ak.y * wV; len += ak.w * wV;

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
    const float oym = -1.0f - ppy, oy0f = 0.0f - ppy, oy1 = 1.0f - ppy, oy2 = 2.0f - ppy;
    if (EXACT) {
      auto tap = [&](int dx, int dy, float offx, float offy) {
        const float4_t c = lds_tex[f_idx + dy * fw + dx];
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
      tap(0, -1, ox0f, oym); tap(1, -1, ox1, oym); tap(-1, 1, oxm, oy1); tap(0, 1, ox0f, oy1);
      tap(0, 0, ox0f, oy0f); tap(-1, 0, oxm, oy0f); tap(1, 1, ox1, oy1); tap(2, 1, ox2, oy1);
      tap(2, 0, ox2, oy0f); tap(1, 0, ox1, oy0f); tap(1, 2, ox1, oy2); tap(0, 2, ox0f, oy2);
    } else {
      // Re-formulated taps (continuous part of the filter; ~1e-6 relative from the reference order):
      //   v = M*off with M = [dir.x*len.x dir.y*len.x ; -dir.y*len.y dir.x*len.y]          (:250-253)
      //   u = min(|v|^2, clp)/clp = sat(off^T Q off),  Q = M^T M / clp  -> the clip is the fma's clamp bit
      //   off^T Q off = q00*ox^2 + q11*oy^2 + 2*q01*ox*oy   (per-column / per-row terms hoisted)
      //   base = 25/16*(2/5*d2-1)^2 - 9/16 = 1/4*d2^2 - 5/4*d2 + 1,  window = (lob*d2-1)^2,  d2 = clp*u
      const float rclp = __builtin_amdgcn_rcpf(clp);
      const float sx = len2x * len2x * rclp, sy = len2y * len2y * rclp;
      const float dxx = dirx * dirx, dyy = diry * diry, dxy2 = 2.0f * (dirx * diry);
      const float q00 = fmaf(dxx, sx, dyy * sy), q11 = fmaf(dyy, sx, dxx * sy), q01 = dxy2 * (sx - sy);
      const float axm = q00 * (oxm * oxm), ax0 = q00 * (ox0f * ox0f), ax1 = q00 * (ox1 * ox1), ax2 = q00 * (ox2 * ox2);
      const float cxm = q01 * oxm, cx0 = q01 * ox0f, cx1 = q01 * ox1, cx2 = q01 * ox2;
      const float bym = q11 * (oym * oym), by0 = q11 * (oy0f * oy0f), by1 = q11 * (oy1 * oy1), by2 = q11 * (oy2 * oy2);
      const float k2 = 0.25f * clp * clp, k1 = -1.25f * clp, k3 = lob * clp;
      auto tap = [&](int dx, int dy, float ax, float cx, float by, float offy) {
        const float4_t c = lds_tex[f_idx + dy * fw + dx];
        const float u = sat(fmaf(cx, offy, ax + by));
        const float base = fmaf(fmaf(k2, u, k1), u, 1.0f);
        const float wa = fmaf(k3, u, -1.0f);
        const float w = base * (wa * wa);
        aR = fmaf(c.x, w, aR); aG = fmaf(c.y, w, aG); aB = fmaf(c.z, w, aB);
        aW += w;
      };
      tap(0, -1, ax0, cx0, bym, oym); tap(1, -1, ax1, cx1, bym, oym);
      tap(-1, 0, axm, cxm, by0, oy0f); tap(0, 0, ax0, cx0, by0, oy0f); tap(1, 0, ax1, cx1, by0, oy0f); tap(2, 0, ax2, cx2, by0, oy0f);
      tap(-1, 1, axm, cxm, by1, oy1); tap(0, 1, ax0, cx0, by1, oy1); tap(1, 1, ax1, cx1, by1, oy1); tap(2, 1, ax2, cx2, by1, oy1);
      tap(0, 2, ax0, cx0, by2, oy2); tap(1, 2, ax1, cx1, by2, oy2);
    }

    // :437 normalise and dering with the min/max of the 4 nearest texels (:416-419)
    const float rW = EXACT ? 1.0f / aW : __builtin_amdgcn_rcpf(aW);
    float pr = aR * rW, pg = aG * rW, pb = aB * rW;
    if (EXACT) { pr = pinned(pr); pg = pinned(pg); pb = pinned(pb); }
    texel_t* const dst = reinterpret_cast<texel_t*>(out_frame + (long long)oy * a.out.pitch + (size_t)ox * sizeof(texel_t));
    if (FMT == FSR1_FORMAT_RGBA16F && !(a.flags & FSR1_FLAG_HDR_SQUARE)) {
      const uint4 mm = lds_mm[f_idx];
      half2_t rg = __builtin_convertvector(float2_t{pr, pg}, half2_t);  // v_cvt_pk_f16_f32, RTNE
      half2_t b1 = __builtin_convertvector(float2_t{pb, 1.0f}, half2_t);
      rg = __builtin_elementwise_min(__builtin_bit_cast(half2_t, mm.z), __builtin_elementwise_max(__builtin_bit_cast(half2_t, mm.x), rg));
      b1 = __builtin_elementwise_min(__builtin_bit_cast(half2_t, mm.w), __builtin_elementwise_max(__builtin_bit_cast(half2_t, mm.y), b1));
      *reinterpret_cast<uint2*>(dst) = uint2{__builtin_bit_cast(uint32_t, rg), __builtin_bit_cast(uint32_t, b1)};  // alpha = 1, FSR_Pass.hlsl:80
    } else {
      const float4_t cf = lds_tex[f_idx], cg = lds_tex[f_idx + 1], cj = lds_tex[f_idx + fw], ck = lds_tex[f_idx + fw + 1];
      pr = fminf(fmaxf(max3f(cf.x, cg.x, cj.x), ck.x), fmaxf(fminf(min3f(cf.x, cg.x, cj.x), ck.x), pr));
      pg = fminf(fmaxf(max3f(cf.y, cg.y, cj.y), ck.y), fmaxf(fminf(min3f(cf.y, cg.y, cj.y), ck.y), pg));
      pb = fminf(fmaxf(max3f(cf.z, cg.z, cj.z), ck.z), fmaxf(fminf(min3f(cf.z, cg.z, cj.z), ck.z), pb));
      if (a.flags & FSR1_FLAG_HDR_SQUARE) { pr *= pr; pg *= pg; pb *= pb; }  // FSR_Pass.hlsl:78-79
      if (EXACT) { pr = pinned(pr); pg = pinned(pg); pb = pinned(pb); }
      *dst = Pixel<FMT>::store(pr, pg, pb, 1.0f);
    }
  }
}

// Bytes of dynamic LDS the kernel needs for a footprint capacity of fp_w x fp_h texels.
size_t easu_lds_bytes(int fmt, int fp_w, int fp_h) {
  (void)fmt;
  return (size_t)fp_w * fp_h * kEasuLdsPerTexel;
}

hipError_t easu_launch(const EasuArgs& a, int fmt, bool exact, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = easu_lds_bytes(fmt, a.fp_w, a.fp_h);
#define FSR1_LAUNCH(F, E)                                                                               \
  do {                                                                                                  \
    if (lds > 48 * 1024) {                                                                              \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&easu_kernel<F, E>),             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
      if (e != hipSuccess) return e;                                                                    \
    }                                                                                                   \
    hipLaunchKernelGGL((easu_kernel<F, E>), grid, block, lds, stream, a);                               \
  } while (0)
  if (fmt == FSR1_FORMAT_RGBA16F) { if (exact) FSR1_LAUNCH(FSR1_FORMAT_RGBA16F, true); else FSR1_LAUNCH(FSR1_FORMAT_RGBA16F, false); }
  else { if (exact) FSR1_LAUNCH(FSR1_FORMAT_RGBA32F, true); else FSR1_LAUNCH(FSR1_FORMAT_RGBA32F, false); }
#undef FSR1_LAUNCH
  return hipGetLastError();
}

}  // namespace fsr1
