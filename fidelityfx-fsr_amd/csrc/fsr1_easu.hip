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
// is re-associated (see easu_pixel in fsr1_easu_math.h), which moves the fp32 result by ~1e-6 relative.
#include "fsr1_device.h"
#include "fsr1_easu_math.h"

namespace fsr1 {

template <int FMT, bool EXACT>
__global__ void __launch_bounds__(kThreads) easu_kernel(const EasuArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
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
  easu_stage_footprint<FMT>(l, a.in, a.in.base + (long long)frame * a.in.frame_stride, fx0, fy0, fw, fh, tid);

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
  for (int r = 0; r < 4; ++r) {
    const int oy = oy0 + wave * 4 + r;
    if (oy >= a.out.height) break;
    float ppy = (float)oy * c0y + c0w;
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const int f_idx = ((int)fpy - fy0) * fw + lx;
    const rgbf_t p = easu_pixel<EXACT>(l, f_idx, ppx, ppy);
    *reinterpret_cast<texel_t*>(out_col + (long long)oy * a.out.pitch) = easu_resolve<FMT, EXACT>(l, f_idx, p, hdr);
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
#define FSR1_LAUNCH_E(F) do { if (exact) FSR1_LAUNCH(F, true); else FSR1_LAUNCH(F, false); } while (0)
  switch (fmt) {
    case FSR1_FORMAT_RGBA16F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F); break;
    case FSR1_FORMAT_RGBA32F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA32F); break;
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA8_UNORM); break;
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_R10G10B10A2_UNORM); break;
    default: return hipErrorInvalidValue;
  }
#undef FSR1_LAUNCH_E
#undef FSR1_LAUNCH
  return hipGetLastError();
}

}  // namespace fsr1
