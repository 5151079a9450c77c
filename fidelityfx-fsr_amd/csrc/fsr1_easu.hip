// EASU — edge adaptive spatial upsampling (FsrEasuF, ffx-fsr/ffx_fsr1.h:315-437) for gfx950.
//
// One 256-thread workgroup produces a 64x16 output tile (arithmetic: include/fsr1_device_easu.hpp):
//   phase 1  the tile's input footprint (every texel any of its 12-tap windows can touch, with the
//            sampler's clamp-to-edge already applied) is read from HBM once, coalesced, converted to
//            fp32 once per *input* texel and parked in LDS as (R,G,B,luma); addresses are a scalar row
//            base plus a 32-bit lane offset, tiles inside the image take a loop without clamps;
//   phase 2  the direction/length terms of FsrEasuSetF (:295-313: dirX, dirY, lenX^2, lenY^2 of the '+'
//            neighbourhood) — what FsrEasuF recomputes per output pixel although it depends on the input
//            only — once per footprint texel that some pixel reads as f/g/j/k.  32 B of LDS per texel;
//   phase 3  generic: each lane walks 4 output pixels of its column; exact 2x: each lane owns the 2x2 output
//            quad that shares one 12-tap window.  Bilinear accumulation of the analysis in the reference's
//            order, kernel shaping, 12 taps from LDS, dering clamp against bounds taken from the four
//            texels f g j k (once per quad in the exact-2x variant), store.
//
// MI355X cost model that shaped it (tools/ubench/ubench2.hip, measured): v_fma/v_mul/v_add_f32
// issue at ~2.4 cycles per wave64 instruction, while v_min/v_max/v_cvt/v_fma_mix, integer and every packed
// (v_pk_*) instruction take ~4.3 and v_rcp/v_rsq ~8.5.  So: fp32 texels in LDS (no per-tap
// conversions), plain v_fma_f32 everywhere, the window clip done by the free `clamp` modifier
// instead of v_min_f32, and as few 4-cycle instructions in staging as possible (DESIGN.md section 3.1 has the
// per-phase instruction budget: 944 per wave, 89 % of them the per-pixel filter itself).
//
// Numerics: arithmetic is fp32 (FsrEasuF), storage is the image format.  Everything up to and
// including the `dirR < 1/32768` decision is evaluated in the reference's exact operation order
// (no contraction): that decision and floor() are the only discontinuities of the filter, so
// they must see bit-identical inputs.  With EXACT the rest follows the reference order as well and
// the result is bit-identical to the CPU-evaluated FsrEasuF; without it the continuous remainder
// is re-associated (easu_filter in fsr1_device_easu.hpp), which moves the fp32 result by ~1e-6 relative.
#include "fsr1_easu_kernel.h"
#include "fsr1_overrides.h"

namespace fsr1 {

// Bytes of dynamic LDS the kernel needs for a footprint capacity of fp_w x fp_h texels.
size_t easu_lds_bytes(int fmt, int fp_w, int fp_h) {
  (void)fmt;
  return easu_lds_region_bytes((size_t)fp_w * fp_h);
}

// Compile-time LDS pitches of the generic kernel's row-interleaved layout (default arithmetic, plain pass).  A tile's footprint
// is 64 * (in / out) + 4 texels wide: 46 at 1.5x, 42 at 1.7x, 54 at 1.3x.  0: the dense run-time layout (any width).
int easu_lds_pitch(int fp_w, bool exact, bool color) {
  if (exact || color) return 0;
  return fp_w <= 48 ? 48 : (fp_w <= 56 ? 56 : (fp_w <= 64 ? 64 : 0));
}

// Exact-2x launches take 64 x 32 tiles (a 35 x 19 footprint per 2048 pixels instead of 35 x 11 per 1024: a seventh less staging; 21.3 KB
// of LDS, still seven workgroups per CU) when the launch has tiles to spare — batches — or runs beside other frames' launches
// (FSR1_FLAG_FRAMES_OVERLAP); a single 4K frame alone keeps 64 x 16: half as many workgroups lengthen its tail by more than the
// staging saves (round 4, profiles/ab_r04/r4c10_tile32.log, EASU us: one 4K frame 42.3 -> 44.5 alone; 16-frame 8K batch 2495 -> 2428;
// two dispatches on three streams 62.5 -> 60.1 per frame).
bool easu_s2_tall_tiles(int width, int height, int frames, bool overlapped, int cus) {
  if (const int forced = override_easu_s2_tall(); forced >= 0) return forced != 0;  // (csrc/fsr1_overrides.h: -1 in the product library)
  const long long tiles16 = (long long)((width + 1 + kTileW - 1) / kTileW) * ((height + 1 + kTileH - 1) / kTileH) * frames;
  return overlapped || tiles16 >= 16ll * 8 * (cus > 0 ? cus : 256);  // sixteen residencies of 64 x 16 tiles (four 4K frames) and up
}

// The generic kernel (default arithmetic, pitched LDS layout) on 64 x 32 tiles with 512-thread workgroups (easu_kernel<..., WAVES = 8>):
// whenever the taller footprint still lets a CU hold three workgroups (24 waves) and the launch has at least two workgroups per CU.
// `lds_tall`: bytes of the 32-row tile's footprint in the pitched layout; `lds_per_cu`: what a CU offers (hipDeviceAttributeMaxSharedMemoryPerMultiprocessor).
bool easu_generic_tall_tiles(int width, int height, int frames, int cus, size_t lds_tall, size_t lds_per_cu) {
  // (the test library's one tile-shape hook, fsr1_debug_easu_tall_tiles, forces BOTH this rule and easu_s2_tall_tiles: include/fsr1_hip_test.h)
  if (const int forced = override_easu_s2_tall(); forced >= 0) return forced != 0 && lds_tall <= lds_per_cu;
  const long long tiles32 = (long long)((width + kTileW - 1) / kTileW) * ((height + 31) / 32) * frames;
  return lds_tall * 3 <= lds_per_cu && tiles32 >= 2ll * (cus > 0 ? cus : 256);
}

// s2: launch the exact-2x variant (the caller has checked con0 and laid the grid out for the shifted tiles); tall: on 64 x 32 tiles
// (default arithmetic only: the EXACT variant's per-pixel form would spill at the seven-wave register budget with the longer loop).
hipError_t easu_strict_launch(const EasuArgs& a, int fmt, bool s2, bool tall, int pitch, hipStream_t stream);  // fsr1_easu_strict.hip

// strict: F-strict (FSR1_FLAG_MATH_STRICT; the host has routed RGBA32F and `c *= c` launches to EXACT): the default arithmetic's launch
// shape, decided exactly as for the default arithmetic.
hipError_t easu_launch(const EasuArgs& a, int fmt, bool exact, bool s2, bool tall, hipStream_t stream, bool strict) {
  const bool hdr = (a.flags & FSR1_FLAG_HDR_SQUARE) != 0;
  int pitch = s2 ? 0 : easu_lds_pitch(a.fp_w, exact, false);
  if (pitch && !tall && easu_lds_bytes(fmt, pitch, a.fp_h) > 40 * 1024) pitch = 0;  // (tall footprints of anisotropic ratios: keep the dense layout's occupancy)
  if (!s2 && tall && !pitch) return hipErrorInvalidValue;  // (the host pairs the 512-thread tile with a pitched layout: easu_generic_tall_tiles)
  if (strict) return (exact || hdr) ? hipErrorInvalidValue : easu_strict_launch(a, fmt, s2, tall, pitch, stream);
#define FSR1_LAUNCH_H(F, E, S, P) return hdr ? easu_launch_one<F, E, false, F, S, true, P>(a, stream) : easu_launch_one<F, E, false, F, S, false, P>(a, stream)
#define FSR1_LAUNCH_T(F, E) return hdr ? easu_launch_one<F, E, false, F, true, true, 0, 32>(a, stream) : easu_launch_one<F, E, false, F, true, false, 0, 32>(a, stream)
#define FSR1_LAUNCH_G8(F, P) return hdr ? easu_launch_one<F, false, false, F, false, true, P, 32, 8>(a, stream) : easu_launch_one<F, false, false, F, false, false, P, 32, 8>(a, stream)
#define FSR1_LAUNCH_E(F)                              \
  do {                                                \
    if (s2 && tall && !exact) FSR1_LAUNCH_T(F, false); \
    if (s2) {                                         \
      if (exact) FSR1_LAUNCH_H(F, true, true, 0);     \
      FSR1_LAUNCH_H(F, false, true, 0);               \
    }                                                 \
    if (exact) FSR1_LAUNCH_H(F, true, false, 0);      \
    if (tall && pitch == 48) FSR1_LAUNCH_G8(F, 48);   \
    if (tall && pitch == 56) FSR1_LAUNCH_G8(F, 56);   \
    if (tall && pitch == 64) FSR1_LAUNCH_G8(F, 64);   \
    if (pitch == 48) FSR1_LAUNCH_H(F, false, false, 48); \
    if (pitch == 56) FSR1_LAUNCH_H(F, false, false, 56); \
    if (pitch == 64) FSR1_LAUNCH_H(F, false, false, 64); \
    FSR1_LAUNCH_H(F, false, false, 0);                \
  } while (0)
  switch (fmt) {
    case FSR1_FORMAT_RGBA16F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F);
    case FSR1_FORMAT_RGBA32F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA32F);
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA8_UNORM);
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_R10G10B10A2_UNORM);
    default: return hipErrorInvalidValue;
  }
#undef FSR1_LAUNCH_E
#undef FSR1_LAUNCH_T
#undef FSR1_LAUNCH_G8
#undef FSR1_LAUNCH_H
}

}  // namespace fsr1
