// EASU -> RCAS in one launch (BASELINE config 4; no counterpart in the reference, which always runs two
// dispatches with a UAV->SRV barrier in between: sample/src/DX12/FSR_Filter.cpp:121-131).
//
// A 256-thread workgroup owns a 64x16 tile of the final image:
//   phases 1-2  input footprint of the tile *plus a 1-pixel apron* -> LDS (shared with the EASU kernel);
//   phase 3     FsrEasuF for the (64+2)x(16+2) apron tile, rounded to the storage format exactly as the
//               two-pass pipeline's intermediary would be, kept in LDS (pixels outside the image are 0:
//               the `Load` rule RCAS sees in the two-pass pipeline, FSR_Pass.hlsl:45,61);
//   phase 4     FsrRcasF from that LDS tile, one row-contiguous store per wave.
// The result is bit-identical to fsr1_easu_dispatch + fsr1_rcas_dispatch with an intermediary of the
// output's format; the 2 x out bytes of HBM traffic of the intermediary are gone, at the price of
// recomputing EASU on the apron (+16 % pixels).
#include "fsr1_fused_kernel.h"

namespace fsr1 {

size_t fused_lds_bytes(int fmt, int fp_w, int fp_h) {
  const size_t texel = fmt == FSR1_FORMAT_RGBA32F ? 16 : (fmt == FSR1_FORMAT_RGBA16F ? 8 : 4);
  return easu_lds_region_bytes((size_t)fp_w * fp_h) + (size_t)kMidW * kMidH * texel;  // footprint + the intermediate tile
}

// F-strict: + the queue of the apron tile's pixels behind the intermediate tile (its size rounded up to 16 bytes)
size_t fused_strict_lds_bytes(int fmt, int fp_w, int fp_h) { return ((fused_lds_bytes(fmt, fp_w, fp_h) + 15) & ~(size_t)15) + easu_strict_queue_bytes((size_t)kMidW * kMidH); }

// strict: F-strict (the host has routed RGBA32F to EXACT)
hipError_t fused_launch(const FusedArgs& a, int fmt, bool exact, hipStream_t stream, bool strict) {
  if (strict) {
    if (exact) return hipErrorInvalidValue;
    switch (fmt) {
      case FSR1_FORMAT_RGBA16F: return fused_launch_one<FSR1_FORMAT_RGBA16F, false, false, FSR1_FORMAT_RGBA16F, true>(a, stream);
      case FSR1_FORMAT_RGBA8_UNORM: return fused_launch_one<FSR1_FORMAT_RGBA8_UNORM, false, false, FSR1_FORMAT_RGBA8_UNORM, true>(a, stream);
      case FSR1_FORMAT_R10G10B10A2_UNORM: return fused_launch_one<FSR1_FORMAT_R10G10B10A2_UNORM, false, false, FSR1_FORMAT_R10G10B10A2_UNORM, true>(a, stream);
      default: return hipErrorInvalidValue;
    }
  }
#define FSR1_LAUNCH_E(F) return exact ? fused_launch_one<F, true, false, F>(a, stream) : fused_launch_one<F, false, false, F>(a, stream)
  switch (fmt) {
    case FSR1_FORMAT_RGBA16F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F);
    case FSR1_FORMAT_RGBA32F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA32F);
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA8_UNORM);
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_R10G10B10A2_UNORM);
    default: return hipErrorInvalidValue;
  }
#undef FSR1_LAUNCH_E
}

}  // namespace fsr1
