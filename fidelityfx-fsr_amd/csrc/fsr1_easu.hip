// EASU — edge adaptive spatial upsampling (FsrEasuF, ffx-fsr/ffx_fsr1.h:315-437) for gfx950.
//
// One 256-thread workgroup produces a 64x16 output tile:
//   phase 1  the tile's input footprint (every texel any of its 12-tap windows can touch, with the
//            sampler's clamp-to-edge already applied) is read from HBM once, coalesced, and parked
//            in LDS in the image's storage format, together with its luma (B*0.5+(R*0.5+G));
//   phase 2  the direction/length analysis of FsrEasuSetF (:295-313) depends only on the '+'
//            neighbourhood of an *input* texel, so it is evaluated once per footprint texel
//            (dirX, dirY, lenX^2, lenY^2) instead of four times per output pixel;
//   phase 3  each lane walks 4 output pixels of its column: bilinear accumulation of the analysis
//            in the reference's order, kernel shaping, 12 taps from LDS, dering clamp, one
//            row-contiguous store per wave.
//
// Numerics: arithmetic is fp32 (FsrEasuF), storage is the image format.  Everything up to and
// including the `dirR < 1/32768` decision is evaluated in the reference's exact operation order
// (no contraction): that decision and floor() are the only discontinuities of the filter, so
// they must see bit-identical inputs.  With EXACT the rest follows the reference order as well and
// the result is bit-identical to the CPU-evaluated FsrEasuF; without it the continuous remainder
// is re-associated/fused (see comments), which moves the fp32 result by a few 1e-7 relative.
#include "fsr1_device.h"

namespace fsr1 {

template <int FMT, bool EXACT>
__global__ void __launch_bounds__(kThreads) easu_kernel(const EasuArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = a.fp_w * a.fp_h;
  float4_t* const lds_ana = reinterpret_cast<float4_t*>(smem);                       // [cap] dirX dirY lenX2 lenY2
  texel_t* const lds_tex = reinterpret_cast<texel_t*>(smem + (size_t)cap * 16);       // [cap] RGBA as stored
  float* const lds_luma = reinterpret_cast<float*>(smem + (size_t)cap * (16 + sizeof(texel_t)));  // [cap]

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
  const int pitch = a.fp_w;  // LDS row pitch in texels

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;

  // ---- phase 1: footprint -> LDS, clamp-to-edge applied here (FSR_Filter.cpp:48-53) ----
  for (int ly = wave; ly < fh; ly += 4) {
    const int gy = min(max(fy0 + ly, 0), a.in.height - 1);
    const char* const row = in_frame + (long long)gy * a.in.pitch;
    for (int lx = lane; lx < fw; lx += 64) {
      const int gx = min(max(fx0 + lx, 0), a.in.width - 1);
      const texel_t px = *reinterpret_cast<const texel_t*>(row + (size_t)gx * sizeof(texel_t));
      const float4_t c = Pixel<FMT>::load(px);
      lds_tex[ly * pitch + lx] = px;
      // :363-366  luma*2 = B*0.5 + (R*0.5 + G); the products by 0.5 are exact, so fusing them is too
      lds_luma[ly * pitch + lx] = fmaf(c.z, 0.5f, fmaf(c.x, 0.5f, c.y));
    }
  }
  __syncthreads();

  // ---- phase 2: FsrEasuSetF's per-texel terms (:295-313), reference order, no contraction ----
  for (int ly = 1 + wave; ly < fh - 1; ly += 4) {
    for (int lx = 1 + lane; lx < fw - 1; lx += 64) {
      const float* L = lds_luma + ly * pitch + lx;
      const float lA = L[-pitch], lB = L[-1], lC = L[0], lD = L[1], lE = L[pitch];
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
      lds_ana[ly * pitch + lx] = float4_t{dirX, dirY, lenX, lenY};
    }
  }
  __syncthreads();

  // ---- phase 3: output pixels ----
  const int ox = ox0 + lane;
  if (ox >= a.out.width) return;
  char* const out_frame = a.out.base + (long long)frame * a.out.frame_stride;
  // :324-326 (x part, shared by this lane's 4 rows)
  float ppx = (float)ox * c0x + c0z;
  const float fpx = floorf(ppx);
  ppx -= fpx;
  const int lx = (int)fpx - fx0;  // footprint column of texel 'f'

#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int oy = oy0 + wave * 4 + r;
    if (oy >= a.out.height) break;
    float ppy = (float)oy * c0y + c0w;
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const int ly = (int)fpy - fy0;
    const int f_idx = ly * pitch + lx;

    // :381-386 bilinear accumulation of the 4 analyses (f,g,j,k), reference order:
    //   dir += dirX*w ; len += lenX*w ; dir.y += dirY*w ; len += lenY*w   for s,t,u,v in turn.
    const float4_t af = lds_ana[f_idx], ag = lds_ana[f_idx + 1], aj = lds_ana[f_idx + pitch], ak = lds_ana[f_idx + pitch + 1];
    const float wS = (1.0f - ppx) * (1.0f - ppy), wT = ppx * (1.0f - ppy), wU = (1.0f - ppx) * ppy, wV = ppx * ppy;
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
    // Non-EXACT: the rotate+scale of :250-253 is folded into one 2x2 matrix per pixel:
    //   v = M * off,  M = [dir.x*len.x  dir.y*len.x ; -dir.y*len.y  dir.x*len.y]
    const float m00 = dirx * len2x, m01 = diry * len2x, m10 = -diry * len2y, m11 = dirx * len2y;
    auto tap = [&](int dx, int dy) {
      const float4_t c = Pixel<FMT>::load(lds_tex[f_idx + dy * pitch + dx]);
      const float offx = (float)dx - ppx, offy = (float)dy - ppy;
      float d2;
      if (EXACT) {
        float vx = (offx * dirx) + (offy * diry);
        float vy = (offx * (-diry)) + (offy * dirx);
        vx *= len2x;
        vy *= len2y;
        d2 = vx * vx + vy * vy;
      } else {
        const float vx = fmaf(offy, m01, offx * m00);
        const float vy = fmaf(offy, m11, offx * m10);
        d2 = fmaf(vy, vy, vx * vx);
      }
      d2 = fminf(d2, clp);
      float wB = mad<EXACT>((float)(2.0 / 5.0), d2, -1.0f);
      float wA = mad<EXACT>(lob, d2, -1.0f);
      wB *= wB;
      wA *= wA;
      wB = mad<EXACT>((float)(25.0 / 16.0), wB, (float)(-(25.0 / 16.0 - 1.0)));
      const float w = wB * wA;
      aR = mad<EXACT>(c.x, w, aR);
      aG = mad<EXACT>(c.y, w, aG);
      aB = mad<EXACT>(c.z, w, aB);
      aW += w;
    };
    // reference order: b c i j f e k l h g o n
    tap(0, -1); tap(1, -1); tap(-1, 1); tap(0, 1); tap(0, 0); tap(-1, 0);
    tap(1, 1); tap(2, 1); tap(2, 0); tap(1, 0); tap(1, 2); tap(0, 2);

    // :416-419 min/max of the 4 nearest texels, :437 normalise and dering
    const float4_t cf = Pixel<FMT>::load(lds_tex[f_idx]), cg = Pixel<FMT>::load(lds_tex[f_idx + 1]);
    const float4_t cj = Pixel<FMT>::load(lds_tex[f_idx + pitch]), ck = Pixel<FMT>::load(lds_tex[f_idx + pitch + 1]);
    const float rW = EXACT ? 1.0f / aW : __builtin_amdgcn_rcpf(aW);
    float pr = fminf(fmaxf(max3f(cf.x, cg.x, cj.x), ck.x), fmaxf(fminf(min3f(cf.x, cg.x, cj.x), ck.x), aR * rW));
    float pg = fminf(fmaxf(max3f(cf.y, cg.y, cj.y), ck.y), fmaxf(fminf(min3f(cf.y, cg.y, cj.y), ck.y), aG * rW));
    float pb = fminf(fmaxf(max3f(cf.z, cg.z, cj.z), ck.z), fmaxf(fminf(min3f(cf.z, cg.z, cj.z), ck.z), aB * rW));
    if (a.flags & FSR1_FLAG_HDR_SQUARE) { pr *= pr; pg *= pg; pb *= pb; }  // FSR_Pass.hlsl:78-79
    *reinterpret_cast<texel_t*>(out_frame + (long long)oy * a.out.pitch + (size_t)ox * sizeof(texel_t)) =
        Pixel<FMT>::store(pr, pg, pb, 1.0f);  // alpha = 1, FSR_Pass.hlsl:80
  }
}

// Bytes of dynamic LDS the kernel needs for a footprint capacity of fp_w x fp_h texels.
size_t easu_lds_bytes(int fmt, int fp_w, int fp_h) {
  const size_t texel = fmt == FSR1_FORMAT_RGBA16F ? 8 : 16;
  return (size_t)fp_w * fp_h * (16 + texel + 4);
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
