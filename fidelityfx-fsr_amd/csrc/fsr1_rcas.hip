// RCAS — robust contrast adaptive sharpening (FsrRcasF, ffx-fsr/ffx_fsr1.h:684-769) for gfx950.
//
// 8 B read + 8 B written per pixel (RGBA16F) against 80 VALU instructions: on MI355X the pass moves its bytes
// at the chip's copy rate, so the kernel is built to touch every input byte as few
// times as it can, to move 16 bytes per lane per memory instruction, and to spend no instruction on staging:
//   * no LDS, no barrier: a wave owns a 128-column x 16-row strip and streams along it, a lane owning TWO
//     adjacent columns (one 16-byte load and one 16-byte store per row) — strips of even strip-rows from top to
//     bottom, those of odd ones from bottom to top (round 6), so that vertically adjacent strips, resident on one
//     XCD at the same time, reach the two rows they share at the same moment and the second request finds them in
//     the L2: a 4K frame fetches 1.014 x its bytes instead of 1.25 x (fsr1_rcas_kernel.h, UP);
//   * vertical neighbours (b above, h below) are the lane's own previous/next rows kept in registers;
//     of the horizontal neighbours, two are the lane's own other pixel and two are the adjacent lanes'
//     facing pixels, fetched in fp32 with DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1);
//   * lanes 0 and 63 additionally load the one texel left / right of the strip, which the DPP move
//     leaves in place for exactly those lanes (an invalid DPP source keeps the old destination);
//   * every texel is converted to fp32 once; loads run ahead of the arithmetic through a register ring;
//   * row addresses are a scalar 64-bit row base plus a 32-bit lane offset; the ring min/max are single
//     v_min3 / v_max3 instructions (include/fsr1_device_rcas.hpp);
//   * the output — the pipeline's last image — is stored non-temporally (FSR1_FLAG_OUTPUT_STREAMING), so it
//     does not displace the intermediary EASU just left in the Infinity Cache;
//   * strips that lie wholly inside the image (all but the image's border strips) run a branch-free body:
//     one basic block per strip, no per-lane predicates, no zero fills;
//   * texels outside the image are 0 (the D3D `Load` rule of the reference's callback, FSR_Pass.hlsl:45,61).
#include "fsr1_rcas_kernel.h"

namespace fsr1 {

constexpr int kRcasShallowRing = 2;  // rows in flight per lane when the launch is short of waves (a single 4K frame)

// Strip height.  Measured on MI355X at 3840x2160 (gpurun_out/): the pass runs at the same ~34 us for
// 8..16-row strips and slows down beyond (24 rows 39 us, 32 rows 42-46 us, 64 rows 67 us) even when the
// workgroups divide evenly over the CUs and however deep the per-lane prefetch ring is: what counts is the number
// of independent row streams in flight, and the apron rows that shorter strips re-read are cheap next to it
// (and since the strips walk alternately up and down, round 6, hits in the L2: profiles/ab_r06/r6d2_pmc_rcas_updown.log).
// 16 rows unless that leaves fewer than four
// waves per SIMD, then 8 — unless the launch runs beside other frames' launches (FSR1_FLAG_FRAMES_OVERLAP, fsr1_pipeline): then the
// neighbour's waves fill the chip and the bytes count again: 16 rows (round 4, two streams, us per frame: 1080p -> 4K 60.3-60.4 ->
// 59.4-59.6, 1440p -> 4K 70.9 -> 69.7-70.0; on one stream the same strips LOSE 5 %: profiles/ab_r04/r4c5_two_stream_rcas_rows.log).
void rcas_geometry(int width, int height, int frames, bool overlapped, int* tiles_x, int* tiles_y, int* rows) {
  const int tx = (width + kRcasCols - 1) / kRcasCols;
  int r = 16;
  if (!overlapped && (long long)tx * ((height + r - 1) / r) * frames * kRcasWaves < 4LL * 4 * 256) r = 8;
  *rows = r;
  *tiles_x = tx;
  *tiles_y = (height + r - 1) / r;
}

hipError_t rcas_launch(const RcasArgs& a, int fmt, bool exact, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kRcasThreads);
  const bool opts = (a.flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA | FSR1_FLAG_HDR_SQUARE)) != 0;
  // shallow ring when the image is short of waves: fewer than 6 per SIMD in 16-row strips (RGBA32F: always 4 rows)
  const bool shallow = (long long)a.tiles_x * ((a.in.height + 15) / 16) * a.frames * kRcasWaves < 6LL * 4 * 256;
#define FSR1_RCAS(F, E, O)                                                                                                                  \
  do {                                                                                                                                      \
    if (shallow) hipLaunchKernelGGL((rcas_kernel<F, E, O, false, F, (F == FSR1_FORMAT_RGBA32F ? 4 : kRcasShallowRing)>), grid, block, 0, stream, a);       \
    else hipLaunchKernelGGL((rcas_kernel<F, E, O, false, F, (F == FSR1_FORMAT_RGBA32F ? 4 : kRcasRing)>), grid, block, 0, stream, a);       \
  } while (0)
#define FSR1_RCAS_O(F, E) do { if (opts) FSR1_RCAS(F, E, true); else FSR1_RCAS(F, E, false); } while (0)
#define FSR1_RCAS_E(F) do { if (exact) FSR1_RCAS_O(F, true); else FSR1_RCAS_O(F, false); } while (0)
  switch (fmt) {
    case FSR1_FORMAT_RGBA16F: FSR1_RCAS_E(FSR1_FORMAT_RGBA16F); break;
    case FSR1_FORMAT_RGBA32F: FSR1_RCAS_E(FSR1_FORMAT_RGBA32F); break;
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_RCAS_E(FSR1_FORMAT_RGBA8_UNORM); break;
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_RCAS_E(FSR1_FORMAT_R10G10B10A2_UNORM); break;
    default: return hipErrorInvalidValue;
  }
#undef FSR1_RCAS_E
#undef FSR1_RCAS_O
#undef FSR1_RCAS
  return hipGetLastError();
}

}  // namespace fsr1
