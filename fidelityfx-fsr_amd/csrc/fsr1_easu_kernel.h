// EASU kernel template (see fsr1_easu.hip for the design notes); instantiated by fsr1_easu.hip (plain pass) and
// fsr1_easu_color.hip (colour prologue / epilogue variants).
#pragma once
#include "fsr1_device.h"
#include "fsr1_easu_math.h"

namespace fsr1 {

size_t easu_lds_bytes(int fmt, int fp_w, int fp_h);

// COLOR: colour stages fused in (fsr1_color_math.h) — FsrSrtmF on every input texel as it is loaded, and
// FsrLfgaF / FsrSrtmInvF / FsrTepdC*F on the result before it is stored as FOUT.  COLOR = false is the plain pass
// (FOUT == FMT), compiled without any of it.
template <int FMT, bool EXACT, bool COLOR = false, int FOUT = FMT>
__global__ void __launch_bounds__(kThreads) easu_kernel(const EasuArgs a) {
  typedef typename Pixel<FOUT>::T texel_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EasuLds l = easu_lds_carve(smem, a.fp_w * a.fp_h);

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
  l.fw = fw;

  const int tid = threadIdx.x;
  easu_stage_footprint<FMT, COLOR, EXACT>(l, a.in, a.in.base + (long long)frame * a.in.frame_stride, fx0, fy0, fw, fh, tid, &a.color);

  // ---- phase 3: output pixels; a lane owns a column, a wave 4 rows ----
  const int lane = tid & 63, wave = tid >> 6;
  const int ox = ox0 + lane;
  if (ox >= a.out.width) return;
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)ox * sizeof(texel_t);
  // :324-326 (x part, shared by this lane's 4 rows)
  float ppx = (float)ox * c0x + c0z;
  const float fpx = floorf(ppx);
  ppx -= fpx;
  const int lx = (int)fpx - fx0;  // footprint column of texel 'f'
  const bool hdr = (a.flags & FSR1_FLAG_HDR_SQUARE) != 0;

#pragma unroll 1
  for (int r = 0; r < kTileH / 4; ++r) {
    const int oy = oy0 + wave * (kTileH / 4) + r;
    if (oy >= a.out.height) break;
    float ppy = (float)oy * c0y + c0w;
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const int f_idx = ((int)fpy - fy0) * fw + lx;
    const rgbf_t p = easu_pixel<EXACT>(l, f_idx, ppx, ppy);
    texel_t* const dst = reinterpret_cast<texel_t*>(out_col + (long long)oy * a.out.pitch);
    if constexpr (COLOR) {
      rgbf_t q = easu_resolve_f(l, f_idx, p, hdr);
      color_epilogue<EXACT>(a.color, (uint32_t)ox, (uint32_t)oy, q.r, q.g, q.b);
      *dst = Pixel<FOUT>::store(q.r, q.g, q.b, 1.0f);
    } else {
      *dst = easu_resolve<FMT, EXACT>(l, f_idx, p, hdr);
    }
  }
}

template <int FMT, bool EXACT, bool COLOR, int FOUT>
hipError_t easu_launch_one(const EasuArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = easu_lds_bytes(FMT, a.fp_w, a.fp_h);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&easu_kernel<FMT, EXACT, COLOR, FOUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((easu_kernel<FMT, EXACT, COLOR, FOUT>), grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace fsr1
