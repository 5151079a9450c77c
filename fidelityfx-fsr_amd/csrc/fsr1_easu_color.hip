// EASU with colour stages fused in: FsrSrtmF on the input texels (prologue), FsrLfgaF / FsrSrtmInvF /
// FsrTepdC8F|C10F on the result (epilogue).  Same kernel template as the plain pass (fsr1_easu_kernel.h), in a
// translation unit of its own so the plain kernels' code is untouched by it.
// Output formats: the input's own, or — from RGBA16F — the two 32 bpp TEPD targets.
#include "fsr1_easu_kernel.h"

namespace fsr1 {

hipError_t easu_color_launch(const EasuArgs& a, int fin, int fout, bool exact, hipStream_t stream) {
#define FSR1_LAUNCH_E(F, O) return exact ? easu_launch_one<F, true, true, O>(a, stream) : easu_launch_one<F, false, true, O>(a, stream)
  if (fin == fout) {
    switch (fin) {
      case FSR1_FORMAT_RGBA16F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F, FSR1_FORMAT_RGBA16F);
      case FSR1_FORMAT_RGBA32F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA32F, FSR1_FORMAT_RGBA32F);
      case FSR1_FORMAT_RGBA8_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA8_UNORM, FSR1_FORMAT_RGBA8_UNORM);
      case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_R10G10B10A2_UNORM, FSR1_FORMAT_R10G10B10A2_UNORM);
      default: return hipErrorInvalidValue;
    }
  }
  if (fin == FSR1_FORMAT_RGBA16F && fout == FSR1_FORMAT_RGBA8_UNORM) FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F, FSR1_FORMAT_RGBA8_UNORM);
  if (fin == FSR1_FORMAT_RGBA16F && fout == FSR1_FORMAT_R10G10B10A2_UNORM) FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F, FSR1_FORMAT_R10G10B10A2_UNORM);
#undef FSR1_LAUNCH_E
  return hipErrorInvalidValue;
}

}  // namespace fsr1
