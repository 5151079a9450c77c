// RCAS with colour stages fused in: FsrSrtmF on every tap as it is loaded (FsrRcasInputF's role), FsrLfgaF /
// FsrSrtmInvF / FsrTepdC8F|C10F on the sharpened result.  Same kernel template as the plain pass
// (fsr1_rcas_kernel.h), in a translation unit of its own.
// Output formats: the input's own, or — from RGBA16F — the two 32 bpp TEPD targets.
#include "fsr1_rcas_kernel.h"

namespace fsr1 {

hipError_t rcas_color_launch(const RcasArgs& a, int fin, int fout, bool exact, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kRcasThreads);
  // OPTS = false compiles the denoise / alpha pass-through / HDR-square flags out (the faster body), as in the plain pass
  const bool opts = (a.flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA | FSR1_FLAG_HDR_SQUARE)) != 0;
#define FSR1_RCAS_K(F, E, P, O) hipLaunchKernelGGL((rcas_kernel<F, E, P, true, O>), grid, block, 0, stream, a)
#define FSR1_RCAS(F, O)                                                                  \
  do {                                                                                   \
    if (exact) { if (opts) FSR1_RCAS_K(F, true, true, O); else FSR1_RCAS_K(F, true, false, O); }       \
    else { if (opts) FSR1_RCAS_K(F, false, true, O); else FSR1_RCAS_K(F, false, false, O); }           \
    return hipGetLastError();                                                            \
  } while (0)
  if (fin == fout) {
    switch (fin) {
      case FSR1_FORMAT_RGBA16F: FSR1_RCAS(FSR1_FORMAT_RGBA16F, FSR1_FORMAT_RGBA16F);
      case FSR1_FORMAT_RGBA32F: FSR1_RCAS(FSR1_FORMAT_RGBA32F, FSR1_FORMAT_RGBA32F);
      case FSR1_FORMAT_RGBA8_UNORM: FSR1_RCAS(FSR1_FORMAT_RGBA8_UNORM, FSR1_FORMAT_RGBA8_UNORM);
      case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_RCAS(FSR1_FORMAT_R10G10B10A2_UNORM, FSR1_FORMAT_R10G10B10A2_UNORM);
      default: return hipErrorInvalidValue;
    }
  }
  if (fin == FSR1_FORMAT_RGBA16F && fout == FSR1_FORMAT_RGBA8_UNORM) FSR1_RCAS(FSR1_FORMAT_RGBA16F, FSR1_FORMAT_RGBA8_UNORM);
  if (fin == FSR1_FORMAT_RGBA16F && fout == FSR1_FORMAT_R10G10B10A2_UNORM) FSR1_RCAS(FSR1_FORMAT_RGBA16F, FSR1_FORMAT_R10G10B10A2_UNORM);
#undef FSR1_RCAS
#undef FSR1_RCAS_K
  return hipErrorInvalidValue;
}

}  // namespace fsr1
