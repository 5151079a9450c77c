// EASU, packed binary16 arithmetic — FsrEasuH (ffx-fsr/ffx_fsr1.h:445-593) for gfx950.
//
// Parity class "H": every arithmetic operation of the reference's half-precision path is one native
// binary16 operation here (v_pk_mul_f16 / v_pk_add_f16 / v_pk_max_f16 ..., contraction off), in the
// reference's operation order and with its two-taps-per-packed-op pairing, so results are
// bit-identical to the reference's FsrEasuH evaluated on the CPU with round-to-nearest-even after
// every operation (oracle/: ref_easu_h, oracle_easu_h).  It is *not* within 1 ULP of FsrEasuF — the
// reference's own H path is not (different magic constants 0x7784/0x59a3, true rcp in FsrEasuSetH).
//
// Same work decomposition as the fp32 kernel (fsr1_easu.hip): 64x16 output tile per 256-thread
// workgroup, footprint staged once into LDS with clamp-to-edge applied, and everything FsrEasuH recomputes per
// output pixel although it depends on the input only — luma (:535-538), the per-position terms of FsrEasuSetH
// (:486-502), the (-x, x) min/max pairs of the 2x2 block (:575-577) — evaluated once per input texel, as the very
// binary16 operation sequences the reference runs per lane.  Texel channels are kept as (texel, right neighbour)
// half2 pairs, the operand shape of the two-taps-per-op FsrEasuTapH.  Position arithmetic stays fp32 (:513-516).
#include "fsr1_device.h"

namespace fsr1 {

typedef unsigned short u16;

__device__ __forceinline__ half2_t h2(half_t a, half_t b) { return half2_t{a, b}; }
__device__ __forceinline__ half2_t h2s(half_t a) { return half2_t{a, a}; }
__device__ __forceinline__ half2_t habs2(half2_t a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ half2_t hmax2(half2_t a, half2_t b) { return __builtin_elementwise_max(a, b); }  // v_pk_max_f16 (maxNum)
__device__ __forceinline__ half2_t hmin2(half2_t a, half2_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ half_t hmax1(half_t a, half_t b) { return __builtin_fmaxf16(a, b); }
__device__ __forceinline__ half_t hmin1(half_t a, half_t b) { return __builtin_fminf16(a, b); }
__device__ __forceinline__ half_t habs1(half_t a) { return __builtin_fabsf16(a); }
// ASatH2, ffx_a.h:896 — clamp(x, 0, 1)
__device__ __forceinline__ half2_t hsat2(half2_t a) { return hmin2(hmax2(a, h2s((half_t)0.0f)), h2s((half_t)1.0f)); }
// ARcpH1/ARcpH2 (GLSL: 1.0/x): correctly rounded binary16 quotient
__device__ __forceinline__ half_t hrcp1(half_t a) { return half_rcp(a); }
__device__ __forceinline__ half2_t hrcp2(half2_t a) { return half2_t{hrcp1(a.x), hrcp1(a.y)}; }
// ffx_a.h:1808, :1820 — integer tricks on the binary16 pattern (16-bit wrap-around subtraction)
__device__ __forceinline__ half_t APrxLoRcpH1(half_t a) { return __builtin_bit_cast(half_t, (u16)(0x7784u - __builtin_bit_cast(u16, a))); }
__device__ __forceinline__ half_t APrxLoRsqH1(half_t a) { return __builtin_bit_cast(half_t, (u16)(0x59a3u - (__builtin_bit_cast(u16, a) >> 1))); }

struct EasuHAcc { half2_t dirPX, dirPY, lenP; };

// FsrEasuSetH :476-503 — two analysis positions per call (the AH2 lanes)
__device__ __forceinline__ void easu_set_h(EasuHAcc& s, half2_t w, half2_t lA, half2_t lB, half2_t lC, half2_t lD, half2_t lE) {
  const half2_t dc = lD - lC;
  const half2_t cb = lC - lB;
  half2_t lenX = hmax2(habs2(dc), habs2(cb));
  lenX = hrcp2(lenX);
  const half2_t dirX = lD - lB;
  s.dirPX = s.dirPX + dirX * w;
  lenX = hsat2(habs2(dirX) * lenX);
  lenX = lenX * lenX;
  s.lenP = s.lenP + lenX * w;
  const half2_t ec = lE - lC;
  const half2_t ca = lC - lA;
  half2_t lenY = hmax2(habs2(ec), habs2(ca));
  lenY = hrcp2(lenY);
  const half2_t dirY = lE - lA;
  s.dirPY = s.dirPY + dirY * w;
  lenY = hsat2(habs2(dirY) * lenY);
  lenY = lenY * lenY;
  s.lenP = s.lenP + lenY * w;
}

struct EasuHTaps { half2_t pR, pG, pB, pW; };

// FsrEasuTapH :452-473 — two taps per call
__device__ __forceinline__ void easu_tap_h(EasuHTaps& p, half2_t offX, half2_t offY, half2_t dir, half2_t len, half_t lob, half_t clp,
                                           half2_t cR, half2_t cG, half2_t cB) {
  half2_t vX = offX * h2s(dir.x) + offY * h2s(dir.y);
  half2_t vY = offX * h2s(-dir.y) + offY * h2s(dir.x);
  vX = vX * h2s(len.x);
  vY = vY * h2s(len.y);
  half2_t d2 = vX * vX + vY * vY;
  d2 = hmin2(d2, h2s(clp));
  half2_t wB = h2s((half_t)(2.0 / 5.0)) * d2 + h2s((half_t)-1.0f);
  half2_t wA = h2s(lob) * d2 + h2s((half_t)-1.0f);
  wB = wB * wB;
  wA = wA * wA;
  wB = h2s((half_t)(25.0 / 16.0)) * wB + h2s((half_t)(-(25.0 / 16.0 - 1.0)));
  const half2_t w = wB * wA;
  p.pR = p.pR + cR * w;
  p.pG = p.pG + cG * w;
  p.pB = p.pB + cB * w;
  p.pW = p.pW + w;
}

// LDS per footprint texel (48 bytes, the same budget as the fp32 kernel):
//   tex1  half4  R G B luma of the texel                                   (phase 1; luma = B*0.5 + (R*0.5 + G), :535-538)
//   texP  4 x half2  (R,G,B,luma) paired with the texel to the right       -> one ds_read_b128 per two-tap FsrEasuTapH call
//   ana1  half4  dirX dirY lenX lenY of FsrEasuSetH for the '+' around the texel (:476-503: they do not depend on the
//                output pixel, only their bilinear weights do), each a single binary16 operation sequence in the
//                reference's order, so the per-pixel accumulation that follows sees bit-identical operands
//   both  3 x half2  (max of -x, max of x) over the 2x2 block at the texel, per channel (:575-577)
struct EasuHLds {
  half4_t* tex1;
  uint4* texP;
  half4_t* ana1;
  uint4* both;
};

__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u(half2_t h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ half2_t swap2(half2_t a) { return __builtin_shufflevector(a, a, 1, 0); }

__global__ void __launch_bounds__(kThreads) easu_h_kernel(const EasuArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = a.fp_w * a.fp_h;
  EasuHLds l;
  l.texP = reinterpret_cast<uint4*>(smem);
  l.both = reinterpret_cast<uint4*>(smem + (size_t)cap * 16);
  l.tex1 = reinterpret_cast<half4_t*>(smem + (size_t)cap * 32);
  l.ana1 = reinterpret_cast<half4_t*>(smem + (size_t)cap * 40);

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int ox0 = tx * kTileW, oy0 = ty * kTileH;
  const float c0x = as_f32(a.con[0]), c0y = as_f32(a.con[1]), c0z = as_f32(a.con[2]), c0w = as_f32(a.con[3]);

  const int oxl = min(ox0 + kTileW, a.out.width) - 1, oyl = min(oy0 + kTileH, a.out.height) - 1;
  const int fx0 = (int)floorf((float)ox0 * c0x + c0z) - 1;
  const int fy0 = (int)floorf((float)oy0 * c0y + c0w) - 1;
  const int fw = min((int)floorf((float)oxl * c0x + c0z) + 2 - fx0 + 1, a.fp_w);
  const int fh = min((int)floorf((float)oyl * c0y + c0w) + 2 - fy0 + 1, a.fp_h);

  const int tid = threadIdx.x;
  const int n = fw * fh;
  const half_t one = (half_t)1.0f, zero = (half_t)0.0f;
  {  // ---- phase 1: HBM -> LDS (clamp-to-edge applied), luma once per input texel ----
    const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;
    const float inv_fw = 1.0f / (float)fw;
    for (int i = tid; i < n; i += kThreads) {
      const int ly = (int)(((float)i + 0.5f) * inv_fw);
      const int lx = i - ly * fw;
      const int gy = min(max(fy0 + ly, 0), a.in.height - 1);
      const int gx = min(max(fx0 + lx, 0), a.in.width - 1);
      const half4_t c = *reinterpret_cast<const half4_t*>(in_frame + (long long)gy * a.in.pitch + (size_t)gx * sizeof(half4_t));
      const half_t hlf = (half_t)0.5f;
      l.tex1[i] = half4_t{c.x, c.y, c.z, (half_t)(c.z * hlf + (c.x * hlf + c.y))};  // :535-538
    }
  }
  __syncthreads();
  // ---- phase 2: per-texel terms (border texels read clamped neighbours and produce values nobody uses) ----
  for (int i = tid; i < n; i += kThreads) {
    const int iu = max(i - fw, 0), id = min(i + fw, n - 1), il = max(i - 1, 0), ir = min(i + 1, n - 1), idr = min(id + 1, n - 1);
    const half4_t tc = l.tex1[i], tr = l.tex1[ir], td = l.tex1[id], tdr = l.tex1[idr];
    const half_t lA = l.tex1[iu].w, lB = l.tex1[il].w, lC = tc.w, lD = tr.w, lE = td.w;
    // FsrEasuSetH :486-502, one AH2 lane
    const half_t dc = lD - lC, cb = lC - lB;
    half_t lenX = hrcp1(hmax1(habs1(dc), habs1(cb)));
    const half_t dirX = lD - lB;
    lenX = hmin1(hmax1(habs1(dirX) * lenX, zero), one);
    lenX = lenX * lenX;
    const half_t ec = lE - lC, ca = lC - lA;
    half_t lenY = hrcp1(hmax1(habs1(ec), habs1(ca)));
    const half_t dirY = lE - lA;
    lenY = hmin1(hmax1(habs1(dirY) * lenY, zero), one);
    lenY = lenY * lenY;
    l.ana1[i] = half4_t{dirX, dirY, lenX, lenY};
    l.texP[i] = uint4{as_u(h2(tc.x, tr.x)), as_u(h2(tc.y, tr.y)), as_u(h2(tc.z, tr.z)), as_u(h2(tc.w, tr.w))};
    // :575-577 min and max of the 2x2 block f g / j k through max() of (-x, x) pairs
    const half2_t bR = hmax2(hmax2(h2(-tc.x, tc.x), h2(-tr.x, tr.x)), hmax2(h2(-td.x, td.x), h2(-tdr.x, tdr.x)));
    const half2_t bG = hmax2(hmax2(h2(-tc.y, tc.y), h2(-tr.y, tr.y)), hmax2(h2(-td.y, td.y), h2(-tdr.y, tdr.y)));
    const half2_t bB = hmax2(hmax2(h2(-tc.z, tc.z), h2(-tr.z, tr.z)), hmax2(h2(-td.z, td.z), h2(-tdr.z, tdr.z)));
    l.both[i] = uint4{as_u(bR), as_u(bG), as_u(bB), 0u};
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int ox = ox0 + lane;
  if (ox >= a.out.width) return;
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)ox * sizeof(half4_t);
  float ppx = (float)ox * c0x + c0z;  // :513-515
  const float fpx = floorf(ppx);
  ppx -= fpx;
  const int lx = (int)fpx - fx0;
  const bool hdr = (a.flags & FSR1_FLAG_HDR_SQUARE) != 0;

#pragma unroll 1
  for (int r = 0; r < kTileH / 4; ++r) {
    const int oy = oy0 + wave * (kTileH / 4) + r;
    if (oy >= a.out.height) break;
    float ppy = (float)oy * c0y + c0w;
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const half2_t ppp = h2((half_t)ppx, (half_t)ppy);  // :516 AH2(pp), RTNE
    const int f = ((int)fpy - fy0) * fw + lx;            // footprint index of texel 'f'
    //    b c
    //  e f g h
    //  i j k l
    //    n o
    // :552-558 the two FsrEasuSetH calls: lanes (f, g) with weights w1, lanes (j, k) with weights w2
    const half2_t wx = h2(one, zero) + h2(-ppp.x, ppp.x);  // :483-484
    const half2_t w1 = wx * h2s(one - ppp.y), w2 = wx * h2s(ppp.y);
    const half4_t af = l.ana1[f], ag = l.ana1[f + 1], aj = l.ana1[f + fw], ak = l.ana1[f + fw + 1];
    EasuHAcc s = {h2s(zero), h2s(zero), h2s(zero)};
    s.dirPX = s.dirPX + h2(af.x, ag.x) * w1;
    s.lenP = s.lenP + h2(af.z, ag.z) * w1;
    s.dirPY = s.dirPY + h2(af.y, ag.y) * w1;
    s.lenP = s.lenP + h2(af.w, ag.w) * w1;
    s.dirPX = s.dirPX + h2(aj.x, ak.x) * w2;
    s.lenP = s.lenP + h2(aj.z, ak.z) * w2;
    s.dirPY = s.dirPY + h2(aj.y, ak.y) * w2;
    s.lenP = s.lenP + h2(aj.w, ak.w) * w2;
    half2_t dir = h2(s.dirPX.x + s.dirPX.y, s.dirPY.x + s.dirPY.y);
    half_t len = s.lenP.x + s.lenP.y;
    // :560-572
    const half2_t dir2 = dir * dir;
    half_t dirR = dir2.x + dir2.y;
    const bool zro = dirR < (half_t)(1.0 / 32768.0);
    dirR = APrxLoRsqH1(dirR);
    dirR = zro ? one : dirR;
    dir.x = zro ? one : dir.x;
    dir = dir * h2s(dirR);
    len = len * (half_t)0.5f;
    len = len * len;
    const half_t stretch = (dir.x * dir.x + dir.y * dir.y) * APrxLoRcpH1(hmax1(habs1(dir.x), habs1(dir.y)));
    const half2_t len2 = h2(one + (stretch - one) * len, one + (half_t)-0.5f * len);
    const half_t lob = (half_t)0.5f + (half_t)((1.0 / 4.0 - 0.04) - 0.5) * len;
    const half_t clp = APrxLoRcpH1(lob);
    // :579-588 pairs in the reference's order: bc, ij, fe, kl, hg, on.  texP[t] holds (t, t+1), so bc / ij / kl come as
    // stored and fe / hg / on are the swapped (e,f) / (g,h) / (n,o)
    EasuHTaps p = {h2s(zero), h2s(zero), h2s(zero), h2s(zero)};
    const half2_t px2 = h2s(ppp.x), py2 = h2s(ppp.y);
    const uint4 bc = l.texP[f - fw], ij = l.texP[f + fw - 1], ef = l.texP[f - 1], kl = l.texP[f + fw + 1], gh = l.texP[f + 1], no = l.texP[f + 2 * fw];
    easu_tap_h(p, h2(zero, one) - px2, h2(-one, -one) - py2, dir, len2, lob, clp, as_h2(bc.x), as_h2(bc.y), as_h2(bc.z));
    easu_tap_h(p, h2(-one, zero) - px2, h2(one, one) - py2, dir, len2, lob, clp, as_h2(ij.x), as_h2(ij.y), as_h2(ij.z));
    easu_tap_h(p, h2(zero, -one) - px2, h2(zero, zero) - py2, dir, len2, lob, clp, swap2(as_h2(ef.x)), swap2(as_h2(ef.y)), swap2(as_h2(ef.z)));
    easu_tap_h(p, h2(one, (half_t)2.0f) - px2, h2(one, one) - py2, dir, len2, lob, clp, as_h2(kl.x), as_h2(kl.y), as_h2(kl.z));
    easu_tap_h(p, h2((half_t)2.0f, one) - px2, h2(zero, zero) - py2, dir, len2, lob, clp, swap2(as_h2(gh.x)), swap2(as_h2(gh.y)), swap2(as_h2(gh.z)));
    easu_tap_h(p, h2(one, zero) - px2, h2((half_t)2.0f, (half_t)2.0f) - py2, dir, len2, lob, clp, swap2(as_h2(no.x)), swap2(as_h2(no.y)), swap2(as_h2(no.z)));
    const half_t aR = p.pR.x + p.pR.y, aG = p.pG.x + p.pG.y, aB = p.pB.x + p.pB.y;
    const half_t aW = p.pW.x + p.pW.y;
    // :593
    const uint4 bo = l.both[f];
    const half2_t bothR = as_h2(bo.x), bothG = as_h2(bo.y), bothB = as_h2(bo.z);
    const half_t rW = hrcp1(aW);
    half_t pr = hmin1(bothR.y, hmax1(-bothR.x, aR * rW));
    half_t pg = hmin1(bothG.y, hmax1(-bothG.x, aG * rW));
    half_t pb = hmin1(bothB.y, hmax1(-bothB.x, aB * rW));
    if (hdr) { pr = pr * pr; pg = pg * pg; pb = pb * pb; }  // FSR_Pass.hlsl:78-79
    *reinterpret_cast<half4_t*>(out_col + (long long)oy * a.out.pitch) = half4_t{pr, pg, pb, one};  // alpha = 1, FSR_Pass.hlsl:80
  }
}

size_t easu_h_lds_bytes(int fp_w, int fp_h) { return (size_t)fp_w * fp_h * 48; }

hipError_t easu_h_launch(const EasuArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = easu_h_lds_bytes(a.fp_w, a.fp_h);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&easu_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(easu_h_kernel, grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace fsr1
