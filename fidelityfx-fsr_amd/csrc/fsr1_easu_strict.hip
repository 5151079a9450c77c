// EASU, F-strict (FSR1_FLAG_MATH_STRICT): the default arithmetic's kernels (fsr1_easu_kernel.h) with the rounding-boundary test and the
// in-workgroup re-evaluation of the pixels that fail it — the stored image is bit-identical to FSR1_FLAG_MATH_EXACT's, i.e. to the
// CPU-evaluated FsrEasuF (ffx-fsr/ffx_fsr1.h:315-437), at close to the default arithmetic's speed.  Design and the measured bound the
// threshold rests on: include/fsr1_device_easu.hpp ("F-strict"), DESIGN.md 3.8.  A translation unit of its own: the instantiations
// compile beside fsr1_easu.hip's.
//
// Variants: the plain pass (no colour stages, no `c *= c`) for the storage formats with a conversion to test against — RGBA16F, RGBA8,
// R10G10B10A2.  RGBA32F stores the binary32 result itself (every value would be re-evaluated) and the colour / HDR variants are rare: the
// host routes those to the EXACT kernels, whose output F-strict promises anyway.
#include "fsr1_easu_kernel.h"

namespace fsr1 {

// LDS of an F-strict launch: the footprint region + the queue of the tile's pixels
size_t easu_strict_lds_bytes(int fmt, int fp_w, int fp_h, int tile_h) { return easu_lds_bytes(fmt, fp_w, fp_h) + easu_strict_queue_bytes((size_t)kTileW * tile_h); }

// s2 / tall / pitch: as easu_launch (fsr1_easu.hip) decided them for the default arithmetic
hipError_t easu_strict_launch(const EasuArgs& a, int fmt, bool s2, bool tall, int pitch, hipStream_t stream) {
#define FSR1_STRICT(F, S, P, T, W) return easu_launch_one<F, false, false, F, S, false, P, T, W, true>(a, stream)
#define FSR1_LAUNCH_E(F)                                 \
  do {                                                   \
    if (s2 && tall) FSR1_STRICT(F, true, 0, 32, 4);      \
    if (s2) FSR1_STRICT(F, true, 0, 16, 4);              \
    if (tall && pitch == 48) FSR1_STRICT(F, false, 48, 32, 8); \
    if (tall && pitch == 56) FSR1_STRICT(F, false, 56, 32, 8); \
    if (tall && pitch == 64) FSR1_STRICT(F, false, 64, 32, 8); \
    if (tall) return hipErrorInvalidValue;               \
    if (pitch == 48) FSR1_STRICT(F, false, 48, 16, 4);   \
    if (pitch == 56) FSR1_STRICT(F, false, 56, 16, 4);   \
    if (pitch == 64) FSR1_STRICT(F, false, 64, 16, 4);   \
    FSR1_STRICT(F, false, 0, 16, 4);                     \
  } while (0)
  switch (fmt) {
    case FSR1_FORMAT_RGBA16F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F);
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA8_UNORM);
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_R10G10B10A2_UNORM);
    default: return hipErrorInvalidValue;
  }
#undef FSR1_LAUNCH_E
#undef FSR1_STRICT
}

}  // namespace fsr1
