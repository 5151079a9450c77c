// Stand-alone colour pass: out = stages(in), stages = FsrSrtmF -> FsrLfgaF -> FsrSrtmInvF -> FsrTepdC8F|C10F
// (ffx-fsr/ffx_fsr1.h:986-1199; the reference leaves the pass itself to the integration, e.g.
// sample/src/DX12/FSR_Tonemapping.hlsl:87).
//
// Pure streaming work, HBM bound (in + out bytes per pixel, ~60 VALU instructions with every stage on): a wave
// owns a 128-column x 8-row block, a lane two adjacent pixels (one 16-byte access per row for RGBA16F), all eight
// row loads are issued before the first use; the four waves of a workgroup stack vertically (128 x 32 tile).
// The noise tile (a few KiB) stays in L2.
#include "fsr1_color_math.h"
#include "fsr1_device.h"

namespace fsr1 {

constexpr int kColorCols = 128, kColorRowsPerWave = 8, kColorRows = 4 * kColorRowsPerWave;

template <int FIN, int FOUT, bool EXACT>
__global__ void __launch_bounds__(kThreads) color_kernel(const ColorPassArgs a) {
  typedef typename Pixel<FIN>::T in_t;
  typedef typename Pixel<FOUT>::T out_t;
  struct __attribute__((aligned(4))) in2_t { char bytes[2 * sizeof(in_t)]; };    // two adjacent texels, one access
  struct __attribute__((aligned(4))) out2_t { char bytes[2 * sizeof(out_t)]; };
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = tx * kColorCols + 2 * lane, y0 = ty * kColorRows + wave * kColorRowsPerWave;
  const int W = a.in.width, H = a.in.height;
  if (x >= W || y0 >= H) return;
  const bool pair = x + 1 < W;
  const char* src = a.in.base + (long long)frame * a.in.frame_stride + (size_t)x * sizeof(in_t);
  char* dst = a.out.base + (long long)frame * a.out.frame_stride + (size_t)x * sizeof(out_t);

  in_t q[kColorRowsPerWave][2];
#pragma unroll
  for (int r = 0; r < kColorRowsPerWave; ++r) {
    const int y = min(y0 + r, H - 1);
    const char* p = src + (long long)y * a.in.pitch;
    if (pair) {
      const in2_t v = *reinterpret_cast<const in2_t*>(p);
      __builtin_memcpy(&q[r][0], v.bytes, sizeof(in_t));
      __builtin_memcpy(&q[r][1], v.bytes + sizeof(in_t), sizeof(in_t));
    } else {
      q[r][0] = *reinterpret_cast<const in_t*>(p);
      q[r][1] = q[r][0];
    }
  }
#pragma unroll
  for (int r = 0; r < kColorRowsPerWave; ++r) {
    const int y = y0 + r;
    if (y >= H) break;
    out_t o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4_t c = Pixel<FIN>::load(q[r][i]);
      c = color_prologue<EXACT>(a.color, c);
      c = color_epilogue<EXACT>(a.color, (uint32_t)(x + i), (uint32_t)y, c);
      o[i] = Pixel<FOUT>::store(c.x, c.y, c.z, c.w);
    }
    char* p = dst + (long long)y * a.out.pitch;
    if (pair) {
      out2_t v;
      __builtin_memcpy(v.bytes, &o[0], sizeof(out_t));
      __builtin_memcpy(v.bytes + sizeof(out_t), &o[1], sizeof(out_t));
      *reinterpret_cast<out2_t*>(p) = v;
    } else {
      *reinterpret_cast<out_t*>(p) = o[0];
    }
  }
}

void color_geometry(int width, int height, int* tiles_x, int* tiles_y) {
  *tiles_x = (width + kColorCols - 1) / kColorCols;
  *tiles_y = (height + kColorRows - 1) / kColorRows;
}

hipError_t color_launch(const ColorPassArgs& a, int fin, int fout, bool exact, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
#define FSR1_COLOR(I, O, E) hipLaunchKernelGGL((color_kernel<I, O, E>), grid, block, 0, stream, a)
#define FSR1_COLOR_E(I, O) do { if (exact) FSR1_COLOR(I, O, true); else FSR1_COLOR(I, O, false); } while (0)
#define FSR1_COLOR_O(I)                                                                        \
  switch (fout) {                                                                              \
    case FSR1_FORMAT_RGBA16F: FSR1_COLOR_E(I, FSR1_FORMAT_RGBA16F); break;                     \
    case FSR1_FORMAT_RGBA32F: FSR1_COLOR_E(I, FSR1_FORMAT_RGBA32F); break;                     \
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_COLOR_E(I, FSR1_FORMAT_RGBA8_UNORM); break;             \
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_COLOR_E(I, FSR1_FORMAT_R10G10B10A2_UNORM); break; \
    default: return hipErrorInvalidValue;                                                      \
  }
  switch (fin) {
    case FSR1_FORMAT_RGBA16F: FSR1_COLOR_O(FSR1_FORMAT_RGBA16F); break;
    case FSR1_FORMAT_RGBA32F: FSR1_COLOR_O(FSR1_FORMAT_RGBA32F); break;
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_COLOR_O(FSR1_FORMAT_RGBA8_UNORM); break;
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_COLOR_O(FSR1_FORMAT_R10G10B10A2_UNORM); break;
    default: return hipErrorInvalidValue;
  }
#undef FSR1_COLOR_O
#undef FSR1_COLOR_E
#undef FSR1_COLOR
  return hipGetLastError();
}

}  // namespace fsr1
