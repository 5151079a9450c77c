// EASU -> RCAS in one launch with the packed-binary16 entry points (FsrEasuH, ffx_fsr1.h:505-593, then FsrRcasH,
// :782-866): the H twin of fsr1_fused_kernel.h.  Phases 1-2 stage the input footprint of a 64 x 16 tile plus a 1-pixel
// apron (include/fsr1_device_half.hpp), phase 3 runs FsrEasuH on the (64+2) x (16+2) apron tile and keeps the result in
// LDS as the RGBA16F texels the two-pass pipeline would have stored (0 outside the image: the `Load` rule RCAS sees,
// FSR_Pass.hlsl:61), phase 4 runs FsrRcasH from there.  Every arithmetic operation is the one native binary16 operation
// the two H kernels run, on the same values: the output equals fsr1_easu_dispatch + fsr1_rcas_dispatch with
// FSR1_FLAG_MATH_PACKED_FP16 bit for bit (tests/test_gpu_parity_h.py).  It is the launch-bound-frame pipeline of the H
// path; at 4K the apron recompute (+16 % of FsrEasuH) costs more than the intermediary's traffic saves.
#include "fsr1_device.h"
#include "fsr1_device_half.hpp"

namespace fsr1 {

namespace {
constexpr int kMidW = kTileW + 2;
constexpr int kMidH = kFusedTileH + 2;
}  // namespace

template <bool OPTS>
__global__ void __launch_bounds__(kThreads) fused_h_kernel(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = a.fp_w * a.fp_h;
  const EasuHLds l = easu_h_lds_carve(smem, cap);
  half4_t* const mid = reinterpret_cast<half4_t*>(smem + (size_t)cap * kEasuHLdsPerTexel);  // [kMidH][kMidW]

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int ox0 = tx * kTileW, oy0 = ty * kFusedTileH;
  const int W = a.out.width, H = a.out.height;
  const float c0x = as_f32(a.easu_con[0]), c0y = as_f32(a.easu_con[1]), c0z = as_f32(a.easu_con[2]), c0w = as_f32(a.easu_con[3]);

  // apron tile = output pixels [ox0-1, ox0+64] x [oy0-1, oy0+16], clipped to the image for the footprint
  const int ax0 = max(ox0 - 1, 0), ay0 = max(oy0 - 1, 0);
  const int axl = min(ox0 + kTileW, W - 1), ayl = min(oy0 + kFusedTileH, H - 1);
  const int fx0 = (int)floorf((float)ax0 * c0x + c0z) - 1;
  const int fy0 = (int)floorf((float)ay0 * c0y + c0w) - 1;
  const int fw = min((int)floorf((float)axl * c0x + c0z) + 2 - fx0 + 1, a.fp_w);
  const int fh = min((int)floorf((float)ayl * c0y + c0w) + 2 - fy0 + 1, a.fp_h);
  const int tid = threadIdx.x;
  easu_h_stage<0, 0>(l, a.in, a.in.base + (long long)frame * a.in.frame_stride, fx0, fy0, fw, fh, tid);

  // ---- phase 3: FsrEasuH on the apron tile -> LDS (EASU runs with Sample.x = 0 when RCAS follows, FSR_Filter.cpp:107) ----
  const int lane = tid & 63, wave = tid >> 6;
  auto easu_to_mid = [&](int mx, int my) {
    const int ox = ox0 - 1 + mx, oy = oy0 - 1 + my;
    half4_t px = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
    if (ox >= 0 && ox < W && oy >= 0 && oy < H) {
      float ppx = (float)ox * c0x + c0z, ppy = (float)oy * c0y + c0w;  // :513-515
      const float fpx = floorf(ppx), fpy = floorf(ppy);
      ppx -= fpx;
      ppy -= fpy;
      const int f = ((int)fpy - fy0) * fw + ((int)fpx - fx0);
      px = easu_h_pixel(l, f, fw, h2((half_t)ppx, (half_t)ppy), false);
    }
    mid[my * kMidW + mx] = px;
  };
#pragma unroll 1
  for (int my = wave; my < kMidH; my += 4) easu_to_mid(lane, my);
  if (wave == 3)  // the two columns left over (64, 65)
    for (int t2 = lane; t2 < 2 * kMidH; t2 += 64) easu_to_mid(kTileW + (t2 & 1), t2 >> 1);
  __syncthreads();

  // ---- phase 4: FsrRcasH from the LDS tile.  A lane owns an output column, a wave 4 rows; two vertically adjacent pixels
  //      ride in the two halves of every packed operand (the lanes of FsrRcasHx2 are independent, so the pairing does not
  //      enter the arithmetic). ----
  const uint32_t flags = OPTS ? a.flags : 0u;
  const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  const half_t sharp1 = __builtin_bit_cast(half_t, (u16)(a.rcas_con[1] & 0xffffu));  // :857
  const int ox = ox0 + lane;
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (long long)ox * 8;
  const half_t one = (half_t)1.0f;
#pragma unroll
  for (int p = 0; p < kFusedTileH / 8; ++p) {
    const int r = wave * (kFusedTileH / 4) + 2 * p;  // tile rows r, r + 1
    const half4_t* const c = mid + (r + 1) * kMidW + (lane + 1);  // centre texel of row r
    const half4_t up = c[-kMidW], e0 = c[0], e1 = c[kMidW], dn = c[2 * kMidW];
    const half4_t d0 = c[-1], f0 = c[1], d1 = c[kMidW - 1], f1 = c[kMidW + 1];
    const rgbh2_t px = rcas_pixel_h2(h2(up.x, e0.x), h2(up.y, e0.y), h2(up.z, e0.z),   // b: above
                                     h2(d0.x, d1.x), h2(d0.y, d1.y), h2(d0.z, d1.z),   // d: left
                                     h2(e0.x, e1.x), h2(e0.y, e1.y), h2(e0.z, e1.z),   // e: centre
                                     h2(f0.x, f1.x), h2(f0.y, f1.y), h2(f0.z, f1.z),   // f: right
                                     h2(e1.x, dn.x), h2(e1.y, dn.y), h2(e1.z, dn.z),   // h: below
                                     sharp1, flags);
    const bool alpha = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) != 0;  // :905-907 / FSR_Pass.hlsl:94
    const int oy = oy0 + r;
    if (ox < W && oy < H) store_out<8>(out_col + (long long)oy * a.out.pitch, half4_t{px.r.x, px.g.x, px.b.x, alpha ? e0.w : one}, stream);
    if (ox < W && oy + 1 < H) store_out<8>(out_col + (long long)(oy + 1) * a.out.pitch, half4_t{px.r.y, px.g.y, px.b.y, alpha ? e1.w : one}, stream);
  }
}

size_t fused_h_lds_bytes(int fp_w, int fp_h) { return (size_t)fp_w * fp_h * kEasuHLdsPerTexel + (size_t)kMidW * kMidH * sizeof(half4_t); }

hipError_t fused_h_launch(const FusedArgs& a, hipStream_t stream) {
  static_assert(kFusedTileH % 8 == 0, "a wave's rows are processed in vertical pairs");
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = fused_h_lds_bytes(a.fp_w, a.fp_h);
  const bool opts = (a.flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA | FSR1_FLAG_HDR_SQUARE)) != 0;
  if (opts) {
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fused_h_kernel<true>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(fused_h_kernel<true>, grid, block, lds, stream, a);
  } else {
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fused_h_kernel<false>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(fused_h_kernel<false>, grid, block, lds, stream, a);
  }
  return hipGetLastError();
}

}  // namespace fsr1
