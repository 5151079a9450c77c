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
#include "fsr1_device_half.hpp"

namespace fsr1 {

// S2: exactly 2x with the viewport covering the input (con0 = {1/2, 1/2, -1/4, -1/4}): a lane owns the 2x2 output quad
// that shares one 12-tap window, tiles are shifted by one pixel, the footprint is a compile-time 35 x 11 and the
// sub-texel positions are the constants 1/4 and 3/4 — see easu_kernel in fsr1_easu_kernel.h.  Same binary16 operations
// on the same values as the generic variant: bit-identical.
template <bool S2>
__global__ void __launch_bounds__(kThreads) easu_h_kernel(const EasuArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kS2W = kTileW / 2 + 3, kS2H = kTileH / 2 + 3;
  const int cap = S2 ? kS2W * kS2H : a.fp_w * a.fp_h;
  EasuHLds l;
  l.texP = reinterpret_cast<uint4*>(smem);
  l.both = reinterpret_cast<uint4*>(smem + (size_t)cap * 16);
  l.tex1 = reinterpret_cast<half4_t*>(smem + (size_t)cap * 32);
  l.ana1 = reinterpret_cast<half4_t*>(smem + (size_t)cap * 40);

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;
  const bool hdr = (a.flags & FSR1_FLAG_HDR_SQUARE) != 0;
  const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;

  if constexpr (S2) {
    static_assert(kTileW == 64 && kTileH == 16, "the exact-2x variant is built for 64 x 16 tiles");
    const int ox0 = tx * kTileW - 1, oy0 = ty * kTileH - 1;
    easu_h_stage<kS2W, kS2H>(l, a.in, in_frame, tx * (kTileW / 2) - 2, ty * (kTileH / 2) - 2, kS2W, kS2H, tid);
    const int W = a.out.width, H = a.out.height;
    const int qx = lane & 31, qy = 2 * wave + (lane >> 5);
    const int oxa = ox0 + 2 * qx, oya = oy0 + 2 * qy;  // odd: the quad is {oxa, oxa+1} x {oya, oya+1}
    const bool xin0 = oxa >= 0 && oxa < W, xin1 = oxa + 1 < W, yin0 = oya >= 0 && oya < H, yin1 = oya + 1 < H;
    if (!((xin0 || xin1) && (yin0 || yin1))) return;
    const int f = (qy + 1) * kS2W + (qx + 1);
    char* const o0 = a.out.base + (long long)frame * a.out.frame_stride + (long long)oya * a.out.pitch + (long long)oxa * 8;
    const half_t q1 = (half_t)0.25f, q3 = (half_t)0.75f;
    typedef TexelPair<FSR1_FORMAT_RGBA16F> pair;
    if (xin0 && xin1 && yin0 && yin1) {
      const half4_t p00 = easu_h_pixel(l, f, kS2W, h2(q1, q1), hdr), p10 = easu_h_pixel(l, f, kS2W, h2(q3, q1), hdr);
      const half4_t p01 = easu_h_pixel(l, f, kS2W, h2(q1, q3), hdr), p11 = easu_h_pixel(l, f, kS2W, h2(q3, q3), hdr);
      store_out<8>(o0, pair::make(p00, p10), stream);
      store_out<8>(o0 + a.out.pitch, pair::make(p01, p11), stream);
      return;
    }
    auto row = [&](char* o, bool yin, half_t ppy) {
      if (!yin) return;
      if (xin0) store_out<8>(o, easu_h_pixel(l, f, kS2W, h2(q1, ppy), hdr), stream);
      if (xin1) store_out<8>(o + 8, easu_h_pixel(l, f, kS2W, h2(q3, ppy), hdr), stream);
    };
    row(o0, yin0, q1);
    row(o0 + a.out.pitch, yin1, q3);
    return;
  }

  const int ox0 = tx * kTileW, oy0 = ty * kTileH;
  const float c0x = as_f32(a.con[0]), c0y = as_f32(a.con[1]), c0z = as_f32(a.con[2]), c0w = as_f32(a.con[3]);
  const int oxl = min(ox0 + kTileW, a.out.width) - 1, oyl = min(oy0 + kTileH, a.out.height) - 1;
  const int fx0 = (int)floorf((float)ox0 * c0x + c0z) - 1;
  const int fy0 = (int)floorf((float)oy0 * c0y + c0w) - 1;
  const int fw = min((int)floorf((float)oxl * c0x + c0z) + 2 - fx0 + 1, a.fp_w);
  const int fh = min((int)floorf((float)oyl * c0y + c0w) + 2 - fy0 + 1, a.fp_h);
  easu_h_stage<0, 0>(l, a.in, in_frame, fx0, fy0, fw, fh, tid);

  const int ox = ox0 + lane;
  if (ox >= a.out.width) return;
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)ox * sizeof(half4_t);
  float ppx = (float)ox * c0x + c0z;  // :513-515
  const float fpx = floorf(ppx);
  ppx -= fpx;
  const int lx = (int)fpx - fx0;

#pragma unroll 1
  for (int r = 0; r < kTileH / 4; ++r) {
    const int oy = oy0 + wave * (kTileH / 4) + r;
    if (oy >= a.out.height) break;
    float ppy = (float)oy * c0y + c0w;
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const half2_t ppp = h2((half_t)ppx, (half_t)ppy);  // :516 AH2(pp), RTNE
    const int f = ((int)fpy - fy0) * fw + lx;            // footprint index of texel 'f'
    store_out<8>(out_col + (long long)oy * a.out.pitch, easu_h_pixel(l, f, fw, ppp, hdr), stream);
  }
}

size_t easu_h_lds_bytes(int fp_w, int fp_h) { return (size_t)fp_w * fp_h * 48; }

// s2: launch the exact-2x variant (the caller has checked con0 and laid the grid out for the shifted tiles).
hipError_t easu_h_launch(const EasuArgs& a, bool s2, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = easu_h_lds_bytes(a.fp_w, a.fp_h);
  if (s2) {
    hipLaunchKernelGGL(easu_h_kernel<true>, grid, block, lds, stream, a);
    return hipGetLastError();
  }
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&easu_h_kernel<false>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(easu_h_kernel<false>, grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace fsr1
