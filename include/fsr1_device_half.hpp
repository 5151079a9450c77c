// Packed-binary16 (parity class "H") building blocks of the FSR 1.0 HIP operator surface: FsrEasuH
// (ffx-fsr/ffx_fsr1.h:445-593) and FsrRcasH / FsrRcasHx2 (:777-984) as native _Float16 vector arithmetic.
//
// Every arithmetic operation of the reference's half-precision path is ONE native binary16 operation here, in the
// reference's order, with contraction off (-ffp-contract=off), so the results are bit-identical to the reference's H
// path evaluated with round-to-nearest-even after every operation.  ARcpH (GLSL `1.0/x`) is half_rcp() of
// fsr1_device_base.hpp: the correctly rounded quotient.
#pragma once
#include "fsr1_device_base.hpp"

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

// as / from the bit pattern of a pair; swap the two halves
__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u(half2_t h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ half2_t swap2(half2_t a) { return __builtin_shufflevector(a, a, 1, 0); }

// FsrEasuSetH's terms for ONE position (one AH2 lane of :486-502) of the '+' neighbourhood  a / b c d / e , before the
// bilinear weighting: (dirX, dirY, lenX^2, lenY^2).  They depend on the input image only, so the tiled kernel evaluates
// them once per input texel; the per-pixel accumulation then sees operands bit-identical to the reference's.
__device__ __forceinline__ half4_t easu_analysis_h(half_t lA, half_t lB, half_t lC, half_t lD, half_t lE) {
  const half_t one = (half_t)1.0f, zero = (half_t)0.0f;
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
  return half4_t{dirX, dirY, lenX, lenY};
}

// :575-577 min and max of a channel over the 2x2 block f g / j k through max() of (-x, x) pairs: (.x = -min, .y = max)
__device__ __forceinline__ half2_t easu_both_h(half_t f, half_t g, half_t j, half_t k) {
  return hmax2(hmax2(h2(-f, f), h2(-g, g)), hmax2(h2(-j, j), h2(-k, k)));
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

// FsrEasuH from the point where the taps are known (:552-593).  ppp = AH2(pp) (:516).
//   ana(k)   k = 0..3 for f, g, j, k -> easu_analysis_h of that texel
//   pair(i)  i = 0..5 -> the (cR, cG, cB) operands of the reference's six FsrEasuTapH calls, in its order:
//            (b,c) (i,j) (f,e) (k,l) (h,g) (o,n)      (:579-588)
//   both(c)  c = 0..2 -> easu_both_h of channel R, G, B
struct EasuHPair { half2_t r, g, b; };
struct rgbh_t { half_t r, g, b; };
template <class Ana, class PairFn, class Both>
__device__ __forceinline__ rgbh_t easu_filter_h(const Ana& ana, const PairFn& pair, const Both& both, half2_t ppp) {
  const half_t one = (half_t)1.0f, zero = (half_t)0.0f;
  // :552-558 the two FsrEasuSetH calls: lanes (f, g) with weights w1, lanes (j, k) with weights w2
  const half2_t wx = h2(one, zero) + h2(-ppp.x, ppp.x);  // :483-484
  const half2_t w1 = wx * h2s(one - ppp.y), w2 = wx * h2s(ppp.y);
  const half4_t af = ana(0), ag = ana(1), aj = ana(2), ak = ana(3);
  half2_t dirPX = h2s(zero), dirPY = h2s(zero), lenP = h2s(zero);
  dirPX = dirPX + h2(af.x, ag.x) * w1;
  lenP = lenP + h2(af.z, ag.z) * w1;
  dirPY = dirPY + h2(af.y, ag.y) * w1;
  lenP = lenP + h2(af.w, ag.w) * w1;
  dirPX = dirPX + h2(aj.x, ak.x) * w2;
  lenP = lenP + h2(aj.z, ak.z) * w2;
  dirPY = dirPY + h2(aj.y, ak.y) * w2;
  lenP = lenP + h2(aj.w, ak.w) * w2;
  half2_t dir = h2(dirPX.x + dirPX.y, dirPY.x + dirPY.y);
  half_t len = lenP.x + lenP.y;
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
  // :579-588
  EasuHTaps p = {h2s(zero), h2s(zero), h2s(zero), h2s(zero)};
  const half2_t px2 = h2s(ppp.x), py2 = h2s(ppp.y);
  const half_t two = (half_t)2.0f;
  const EasuHPair bc = pair(0), ij = pair(1), fe = pair(2), kl = pair(3), hg = pair(4), on = pair(5);
  easu_tap_h(p, h2(zero, one) - px2, h2(-one, -one) - py2, dir, len2, lob, clp, bc.r, bc.g, bc.b);
  easu_tap_h(p, h2(-one, zero) - px2, h2(one, one) - py2, dir, len2, lob, clp, ij.r, ij.g, ij.b);
  easu_tap_h(p, h2(zero, -one) - px2, h2(zero, zero) - py2, dir, len2, lob, clp, fe.r, fe.g, fe.b);
  easu_tap_h(p, h2(one, two) - px2, h2(one, one) - py2, dir, len2, lob, clp, kl.r, kl.g, kl.b);
  easu_tap_h(p, h2(two, one) - px2, h2(zero, zero) - py2, dir, len2, lob, clp, hg.r, hg.g, hg.b);
  easu_tap_h(p, h2(one, zero) - px2, h2(two, two) - py2, dir, len2, lob, clp, on.r, on.g, on.b);
  const half_t aR = p.pR.x + p.pR.y, aG = p.pG.x + p.pG.y, aB = p.pB.x + p.pB.y;
  const half_t aW = p.pW.x + p.pW.y;
  // :593
  const half2_t bothR = both(0), bothG = both(1), bothB = both(2);
  const half_t rW = hrcp1(aW);
  return rgbh_t{hmin1(bothR.y, hmax1(-bothR.x, aR * rW)), hmin1(bothG.y, hmax1(-bothG.x, aG * rW)), hmin1(bothB.y, hmax1(-bothB.x, aB * rW))};
}

// ---- the LDS-staged form of FsrEasuH (what the library's H kernels run) ----
// LDS per footprint texel (48 bytes):
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

// Phases 1 and 2 of the H kernel for the footprint [fx0, fx0+fw) x [fy0, fy0+fh).  FW, FH: compile-time extent (exact 2x).
// THREADS: threads of the workgroup, all of which must make the call (it contains two barriers).
template <int FW, int FH, int THREADS = 256>
__device__ __forceinline__ void easu_h_stage(const EasuHLds& l, const ImageView& in, const char* in_frame, int fx0, int fy0, int fw_rt, int fh_rt, int tid) {
  const int fw = FW ? FW : fw_rt, fh = FH ? FH : fh_rt;
  const int n = fw * fh - 1;  // the bottom-right corner texel is touched by no 12-tap window (no (2, 2) tap)
  const float inv_fw = 1.0f / (float)fw;
  auto row_of = [&](int i) { return FW ? i / FW : (int)(((float)i + 0.5f) * inv_fw); };
  {  // ---- phase 1: HBM -> LDS (clamp-to-edge applied), luma once per input texel; row base + 32-bit lane offsets ----
    const int gy0 = min(max(fy0, 0), in.height - 1);
    const char* const base = in_frame + (long long)gy0 * in.pitch;
    const uint32_t pitch = (uint32_t)in.pitch;
    const half_t hlf = (half_t)0.5f;
    auto stage = [&](int i, uint32_t off) {
      const half4_t c = *reinterpret_cast<const half4_t*>(base + (size_t)off);
      l.tex1[i] = half4_t{c.x, c.y, c.z, (half_t)(c.z * hlf + (c.x * hlf + c.y))};  // :535-538
    };
    if (fx0 >= 0 && fy0 >= 0 && fx0 + fw <= in.width && fy0 + fh <= in.height) {  // wave-uniform: nothing to clamp
      const uint32_t x_off = (uint32_t)fx0 * 8u;
      for (int i = tid; i < n; i += THREADS) {
        const int ly = row_of(i);
        stage(i, (uint32_t)ly * pitch + (uint32_t)(i - ly * fw) * 8u + x_off);
      }
    } else {
      for (int i = tid; i < n; i += THREADS) {
        const int ly = row_of(i);
        const int gy = min(max(fy0 + ly, 0), in.height - 1);
        const int gx = min(max(fx0 + (i - ly * fw), 0), in.width - 1);
        stage(i, (uint32_t)(gy - gy0) * pitch + (uint32_t)gx * 8u);
      }
    }
  }
  __syncthreads();
  // ---- phase 2a: (texel, right neighbour) pairs — read as the left element by taps in columns 0 .. fw-2 of every row
  //      (an entry in the last column pairs with the next row's first texel and is never read) ----
  for (int i = tid; i < n - 1; i += THREADS) {
    const half4_t tc = l.tex1[i], tr = l.tex1[i + 1];
    l.texP[i] = uint4{as_u(h2(tc.x, tr.x)), as_u(h2(tc.y, tr.y)), as_u(h2(tc.z, tr.z)), as_u(h2(tc.w, tr.w))};
  }
  // ---- phase 2b: per-texel terms of the texels read as f / g / j / k of some pixel: columns 1..fw-2, rows 1..fh-2
  //      (every neighbour of those lies inside the footprint, so nothing is clamped) ----
  const int iw = fw - 2, m = iw * (fh - 2);
  const float inv_iw = 1.0f / (float)iw;
  for (int j = tid; j < m; j += THREADS) {
    const int y = FW ? j / (FW - 2) : (int)(((float)j + 0.5f) * inv_iw);
    const int i = (y + 1) * fw + (j - y * iw) + 1;
    const half4_t tc = l.tex1[i], tr = l.tex1[i + 1], td = l.tex1[i + fw];
    l.ana1[i] = easu_analysis_h(l.tex1[i - fw].w, l.tex1[i - 1].w, tc.w, tr.w, td.w);  // FsrEasuSetH :486-502, one AH2 lane
    if (i + fw + 1 < n) {  // the 2x2 block at the last interior texel would reach the unstaged corner; no pixel has that texel as 'f'
      const half4_t tdr = l.tex1[i + fw + 1];
      // :575-577 min and max of the 2x2 block f g / j k through max() of (-x, x) pairs
      const half2_t bR = easu_both_h(tc.x, tr.x, td.x, tdr.x), bG = easu_both_h(tc.y, tr.y, td.y, tdr.y), bB = easu_both_h(tc.z, tr.z, td.z, tdr.z);
      l.both[i] = uint4{as_u(bR), as_u(bG), as_u(bB), 0u};
    }
  }
  __syncthreads();
}

// FsrEasuH for the pixel whose 'f' texel sits at footprint index f, sub-texel position ppp (= AH2(pp), :516).
__device__ __forceinline__ half4_t easu_h_pixel(const EasuHLds& l, int f, int fw, half2_t ppp, bool hdr) {
  // texP[t] holds (t, t+1), so the reference's pairs bc / ij / kl come as stored and fe / hg / on are the swapped
  // (e,f) / (g,h) / (n,o)
  const uint4 bo = l.both[f];
  const rgbh_t px = easu_filter_h(
      [&](int k) { return l.ana1[f + (k >> 1) * fw + (k & 1)]; },
      [&](int i) {
        const int at[6] = {f - fw, f + fw - 1, f - 1, f + fw + 1, f + 1, f + 2 * fw};
        const uint4 t = l.texP[at[i]];
        const bool sw = i == 2 || i == 4 || i == 5;
        return sw ? EasuHPair{swap2(as_h2(t.x)), swap2(as_h2(t.y)), swap2(as_h2(t.z))} : EasuHPair{as_h2(t.x), as_h2(t.y), as_h2(t.z)};
      },
      [&](int c) { return as_h2(c == 0 ? bo.x : (c == 1 ? bo.y : bo.z)); }, ppp);
  half_t pr = px.r, pg = px.g, pb = px.b;
  if (hdr) { pr = pr * pr; pg = pg * pg; pb = pb * pb; }  // FSR_Pass.hlsl:78-79
  return half4_t{pr, pg, pb, (half_t)1.0f};              // alpha = 1, FSR_Pass.hlsl:80
}

constexpr int kEasuHLdsPerTexel = 48;
__device__ __forceinline__ EasuHLds easu_h_lds_carve(char* smem, int capacity_texels) {
  EasuHLds l;
  l.texP = reinterpret_cast<uint4*>(smem);
  l.both = reinterpret_cast<uint4*>(smem + (size_t)capacity_texels * 16);
  l.tex1 = reinterpret_cast<half4_t*>(smem + (size_t)capacity_texels * 32);
  l.ana1 = reinterpret_cast<half4_t*>(smem + (size_t)capacity_texels * 40);
  return l;
}

// ---- RCAS, two pixels per call in the two halves of every operand (FsrRcasHx2's arithmetic, :913-984) ----
struct soa_t { half2_t r, g, b, a; };  // two pixels: .x = left (even column), .y = right

__device__ __forceinline__ half2_t mx2(half2_t a, half2_t b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ half2_t mn2(half2_t a, half2_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ half2_t ab2(half2_t a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ half2_t s2(float v) { return half2_t{(half_t)v, (half_t)v}; }
__device__ __forceinline__ half2_t rcp2(half2_t a) { return half2_t{half_rcp(a.x), half_rcp(a.y)}; }  // ARcpH2: correctly rounded
// AMin3H2(x,y,z) = min(x, min(y, z)), ffx_a.h:1150 ; AMax3H2 likewise
__device__ __forceinline__ half2_t mn3(half2_t x, half2_t y, half2_t z) { return mn2(x, mn2(y, z)); }
__device__ __forceinline__ half2_t mx3(half2_t x, half2_t y, half2_t z) { return mx2(x, mx2(y, z)); }
// APrxMedRcpH2, ffx_a.h:1816: b = bits(0x778d - bits(a)); b*(-b*a + 2)
__device__ __forceinline__ half2_t prx_med_rcp2(half2_t a) {
  typedef u16 u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 k = {0x778du, 0x778du};
  const half2_t b = __builtin_bit_cast(half2_t, (u16x2)(k - __builtin_bit_cast(u16x2, a)));
  return b * (-b * a + s2(2.0f));
}

// One FsrRcasHx2 evaluation from its taps (b above, d left, e centre, f right, h below; each a pair of pixels per
// channel).  `sharp1` = AH2_AU1(con.y).x, the packed half of con[1] (:857).  Returns the pair of sharpened pixels
// (alpha is the caller's: 1 or the centre tap's, :905-907).
struct rgbh2_t { half2_t r, g, b; };
__device__ __forceinline__ rgbh2_t rcas_pixel_h2(half2_t bR, half2_t bG, half2_t bB, half2_t dR, half2_t dG, half2_t dB, half2_t eR, half2_t eG,
                                                 half2_t eB, half2_t fR, half2_t fG, half2_t fB, half2_t hR, half2_t hG, half2_t hB, half_t sharp1,
                                                 uint32_t flags) {
  const half2_t sharp = {sharp1, sharp1};
  const half2_t hlf = s2(0.5f), qtr = s2(0.25f), four = s2(4.0f), one = s2(1.0f);
  // :946-951 min and max of ring
  const half2_t mn4R = mn2(mn3(bR, dR, fR), hR), mn4G = mn2(mn3(bG, dG, fG), hG), mn4B = mn2(mn3(bB, dB, fB), hB);
  const half2_t mx4R = mx2(mx3(bR, dR, fR), hR), mx4G = mx2(mx3(bG, dG, fG), hG), mx4B = mx2(mx3(bB, dB, fB), hB);
  // :953-961 limiters (peakC = (1, -4))
  const half2_t m4 = s2(-4.0f);
  const half2_t hitMinR = mn2(mn4R, eR) * rcp2(four * mx4R);
  const half2_t hitMinG = mn2(mn4G, eG) * rcp2(four * mx4G);
  const half2_t hitMinB = mn2(mn4B, eB) * rcp2(four * mx4B);
  const half2_t hitMaxR = (one - mx2(mx4R, eR)) * rcp2(four * mn4R + m4);
  const half2_t hitMaxG = (one - mx2(mx4G, eG)) * rcp2(four * mn4G + m4);
  const half2_t hitMaxB = (one - mx2(mx4B, eB)) * rcp2(four * mn4B + m4);
  const half2_t lobeR = mx2(-hitMinR, hitMaxR), lobeG = mx2(-hitMinG, hitMaxG), lobeB = mx2(-hitMinB, hitMaxB);
  half2_t lobe = mx2(s2(-(0.25f - (1.0f / 16.0f))), mn2(mx3(lobeR, lobeG, lobeB), s2(0.0f))) * sharp;
  if (flags & FSR1_FLAG_RCAS_DENOISE) {  // :935-944, :969-971
    const half2_t bL = bB * hlf + (bR * hlf + bG), dL = dB * hlf + (dR * hlf + dG), eL = eB * hlf + (eR * hlf + eG);
    const half2_t fL = fB * hlf + (fR * hlf + fG), hL = hB * hlf + (hR * hlf + hG);
    half2_t nz = qtr * bL + qtr * dL + qtr * fL + qtr * hL - eL;
    nz = mn2(mx2(ab2(nz) * prx_med_rcp2(mx3(mx3(bL, dL, eL), fL, hL) - mn3(mn3(bL, dL, eL), fL, hL)), s2(0.0f)), one);
    nz = s2(-0.5f) * nz + one;
    lobe = lobe * nz;
  }
  // :973-976 resolve
  const half2_t rcpL = prx_med_rcp2(four * lobe + one);
  half2_t pR = (lobe * bR + lobe * dR + lobe * hR + lobe * fR + eR) * rcpL;
  half2_t pG = (lobe * bG + lobe * dG + lobe * hG + lobe * fG + eG) * rcpL;
  half2_t pB = (lobe * bB + lobe * dB + lobe * hB + lobe * fB + eB) * rcpL;
  if (flags & FSR1_FLAG_HDR_SQUARE) { pR = pR * pR; pG = pG * pG; pB = pB * pB; }  // FSR_Pass.hlsl:92-93
  return rgbh2_t{pR, pG, pB};
}

// FsrRcasH for ONE pixel (:782-866): the same operation list as rcas_pixel_h2 on scalar binary16 operands (v_*_f16 instead of
// v_pk_*_f16 — each operation rounds to binary16 either way, so the value equals the corresponding lane of the two-pixel form,
// and an outside kernel that sharpens one pixel per lane does not pay for two).
__device__ __forceinline__ half_t prx_med_rcp1(half_t a) {  // APrxMedRcpH1, ffx_a.h:1815
  const half_t b = __builtin_bit_cast(half_t, (u16)(0x778du - __builtin_bit_cast(u16, a)));
  return b * (-b * a + (half_t)2.0f);
}
struct rgbh1_t { half_t r, g, b; };
__device__ __forceinline__ rgbh1_t rcas_pixel_h1(half_t bR, half_t bG, half_t bB, half_t dR, half_t dG, half_t dB, half_t eR, half_t eG, half_t eB,
                                                 half_t fR, half_t fG, half_t fB, half_t hR, half_t hG, half_t hB, half_t sharp, uint32_t flags) {
  const half_t hlf = (half_t)0.5f, qtr = (half_t)0.25f, four = (half_t)4.0f, one = (half_t)1.0f, zero = (half_t)0.0f, m4 = (half_t)-4.0f;
  auto mn3h = [](half_t x, half_t y, half_t z) { return hmin1(x, hmin1(y, z)); };  // AMin3H1, ffx_a.h:1149
  auto mx3h = [](half_t x, half_t y, half_t z) { return hmax1(x, hmax1(y, z)); };
  // :828-833 min and max of ring
  const half_t mn4R = hmin1(mn3h(bR, dR, fR), hR), mn4G = hmin1(mn3h(bG, dG, fG), hG), mn4B = hmin1(mn3h(bB, dB, fB), hB);
  const half_t mx4R = hmax1(mx3h(bR, dR, fR), hR), mx4G = hmax1(mx3h(bG, dG, fG), hG), mx4B = hmax1(mx3h(bB, dB, fB), hB);
  // :835-843 limiters (peakC = (1, -4))
  const half_t hitMinR = hmin1(mn4R, eR) * hrcp1(four * mx4R);
  const half_t hitMinG = hmin1(mn4G, eG) * hrcp1(four * mx4G);
  const half_t hitMinB = hmin1(mn4B, eB) * hrcp1(four * mx4B);
  const half_t hitMaxR = (one - hmax1(mx4R, eR)) * hrcp1(four * mn4R + m4);
  const half_t hitMaxG = (one - hmax1(mx4G, eG)) * hrcp1(four * mn4G + m4);
  const half_t hitMaxB = (one - hmax1(mx4B, eB)) * hrcp1(four * mn4B + m4);
  const half_t lobeR = hmax1(-hitMinR, hitMaxR), lobeG = hmax1(-hitMinG, hitMaxG), lobeB = hmax1(-hitMinB, hitMaxB);
  half_t lobe = hmax1((half_t)(-(0.25f - (1.0f / 16.0f))), hmin1(mx3h(lobeR, lobeG, lobeB), zero)) * sharp;  // :847
  if (flags & FSR1_FLAG_RCAS_DENOISE) {  // :817-826, :849-851
    const half_t bL = bB * hlf + (bR * hlf + bG), dL = dB * hlf + (dR * hlf + dG), eL = eB * hlf + (eR * hlf + eG);
    const half_t fL = fB * hlf + (fR * hlf + fG), hL = hB * hlf + (hR * hlf + hG);
    half_t nz = qtr * bL + qtr * dL + qtr * fL + qtr * hL - eL;
    nz = hmin1(hmax1(habs1(nz) * prx_med_rcp1(mx3h(mx3h(bL, dL, eL), fL, hL) - mn3h(mn3h(bL, dL, eL), fL, hL)), zero), one);
    nz = (half_t)-0.5f * nz + one;
    lobe = lobe * nz;
  }
  // :853-856 resolve
  const half_t rcpL = prx_med_rcp1(four * lobe + one);
  half_t pR = (lobe * bR + lobe * dR + lobe * hR + lobe * fR + eR) * rcpL;
  half_t pG = (lobe * bG + lobe * dG + lobe * hG + lobe * fG + eG) * rcpL;
  half_t pB = (lobe * bB + lobe * dB + lobe * hB + lobe * fB + eB) * rcpL;
  if (flags & FSR1_FLAG_HDR_SQUARE) { pR = pR * pR; pG = pG * pG; pB = pB * pB; }  // FSR_Pass.hlsl:92-93
  return rgbh1_t{pR, pG, pB};
}

}  // namespace fsr1
