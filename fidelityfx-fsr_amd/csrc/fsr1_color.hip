// Stand-alone colour pass: out = stages(in), stages = FsrSrtmF -> FsrLfgaF -> FsrSrtmInvF -> FsrTepdC8F|C10F
// (ffx-fsr/ffx_fsr1.h:986-1199; the reference leaves the pass itself to the integration, e.g.
// sample/src/DX12/FSR_Tonemapping.hlsl:87).
//
// Pure streaming work, HBM bound (in + out bytes per pixel, ~60 VALU instructions with every stage on): a wave
// owns a 128-column x 8-row block, a lane two adjacent pixels (one 16-byte access per row for RGBA16F), all eight
// row loads are issued before the first use; the four waves of a workgroup stack vertically (128 x 32 tile).
// The noise tile (a few KiB) stays in L2.
#include "fsr1_device_color.hpp"
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
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
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

// ---- packed binary16 arithmetic: FsrSrtmHx2 / FsrLfgaHx2 / FsrSrtmInvHx2 / FsrTepdC8Hx2 | C10Hx2 with FsrTepdDitHx2's
//      per-lane formula (ffx_fsr1.h:1017-1024, :1048-1056, :1124-1198).  A lane's two adjacent pixels are the two halves
//      of every AH2; each reference operation is one native binary16 operation (contraction off), so the result is
//      bit-identical to the CPU-evaluated H path (parity class "H").  RGBA16F in and out. ----
__device__ __forceinline__ half2_t c_h2(half_t a, half_t b) { return half2_t{a, b}; }
__device__ __forceinline__ half2_t c_h2s(half_t a) { return half2_t{a, a}; }
__device__ __forceinline__ half2_t c_max2(half2_t a, half2_t b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ half2_t c_min2(half2_t a, half2_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ half2_t c_sat2(half2_t a) { return c_min2(c_max2(a, c_h2s((half_t)0.0f)), c_h2s((half_t)1.0f)); }
__device__ __forceinline__ half2_t c_rcp2(half2_t a) { return half2_t{half_rcp(a.x), half_rcp(a.y)}; }  // ARcpH2: correctly rounded 1/x
// correctly rounded binary16 sqrt: the binary32 sqrt (correctly rounded) narrowed RTNE is exact-rounding because 24 >= 2*11+2
__device__ __forceinline__ half2_t c_sqrt2(half2_t a) { return half2_t{(half_t)sqrtf((float)a.x), (half_t)sqrtf((float)a.y)}; }
__device__ __forceinline__ half2_t c_floor2(half2_t a) { return half2_t{(half_t)__builtin_floorf16(a.x), (half_t)__builtin_floorf16(a.y)}; }
// ffx_a.h:1815 APrxMedRcpH2
__device__ __forceinline__ half2_t APrxMedRcpH2(half2_t a) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 ai = __builtin_bit_cast(u16x2, a);
  const u16x2 bi = {(unsigned short)(0x778du - ai.x), (unsigned short)(0x778du - ai.y)};
  const half2_t b = __builtin_bit_cast(half2_t, bi);
  return b * (-b * a + c_h2s((half_t)2.0f));
}
__device__ __forceinline__ half2_t AGtZeroH2(half2_t m) { return c_sat2(m * c_h2s((half_t)__builtin_inff())); }  // ffx_a.h:1525
__device__ __forceinline__ half2_t FsrTepdCHx2(half2_t c, half2_t dit, half_t k, half_t rk) {  // :1153-1198, one channel
  half2_t n = c_sqrt2(c);
  n = c_floor2(n * c_h2s(k)) * c_h2s(rk);
  const half2_t a = n * n;
  half2_t b = n + c_h2s(rk);
  b = b * b;
  const half2_t r = (c - b) * APrxMedRcpH2(a - b);
  return c_sat2(n + AGtZeroH2(dit - r) * c_h2s(rk));
}

__global__ void __launch_bounds__(kThreads) color_h_kernel(const ColorPassArgs a) {
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = tx * kColorCols + 2 * lane, y0 = ty * kColorRows + wave * kColorRowsPerWave;
  const int W = a.in.width, H = a.in.height;
  if (x >= W || y0 >= H) return;
  const bool pair = x + 1 < W;
  const char* src = a.in.base + (long long)frame * a.in.frame_stride + (size_t)x * sizeof(half4_t);
  char* dst = a.out.base + (long long)frame * a.out.frame_stride + (size_t)x * sizeof(half4_t);
  struct __attribute__((aligned(8))) px2_t { half4_t p[2]; };
  const uint32_t st = a.color.stages;
  const half_t one = (half_t)1.0f;
  const half2_t amount = c_h2s((half_t)a.color.amount), bias = c_h2s((half_t)a.color.bias);
  const bool c8 = (st & FSR1_COLOR_TEPD_C8) != 0;
  const half_t k = c8 ? (half_t)255.0f : (half_t)1023.0f, rk = c8 ? (half_t)(1.0 / 255.0) : (half_t)(1.0 / 1023.0);
#pragma unroll 1
  for (int r = 0; r < kColorRowsPerWave; ++r) {
    const int y = y0 + r;
    if (y >= H) break;
    px2_t q;
    if (pair) q = *reinterpret_cast<const px2_t*>(src + (long long)y * a.in.pitch);
    else { q.p[0] = *reinterpret_cast<const half4_t*>(src + (long long)y * a.in.pitch); q.p[1] = q.p[0]; }
    half2_t cR = c_h2(q.p[0].x, q.p[1].x), cG = c_h2(q.p[0].y, q.p[1].y), cB = c_h2(q.p[0].z, q.p[1].z);
    float4_t n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
    if (st & kColorNeedsNoise) { n0 = noise_fetch(a.color.noise, (uint32_t)x, (uint32_t)y); n1 = noise_fetch(a.color.noise, (uint32_t)x + 1u, (uint32_t)y); }
    if (st & FSR1_COLOR_SRTM) {  // :1052-1053
      const half2_t rcp = c_rcp2(c_max2(cR, c_max2(cG, cB)) + c_h2s(one));
      cR = cR * rcp; cG = cG * rcp; cB = cB * rcp;
    }
    if (st & FSR1_COLOR_LFGA) {  // :1022-1023
      const half2_t tR = c_h2((half_t)n0.x, (half_t)n1.x) + bias, tG = c_h2((half_t)n0.y, (half_t)n1.y) + bias, tB = c_h2((half_t)n0.z, (half_t)n1.z) + bias;
      cR = cR + (tR * amount) * c_min2(c_h2s(one) - cR, cR);
      cG = cG + (tG * amount) * c_min2(c_h2s(one) - cG, cG);
      cB = cB + (tB * amount) * c_min2(c_h2s(one) - cB, cB);
    }
    if (st & FSR1_COLOR_SRTM_INV) {  // :1054-1055
      const half2_t rcp = c_rcp2(c_max2(c_h2s((half_t)(1.0 / 32768.0)), c_h2s(one) - c_max2(cR, c_max2(cG, cB))));
      cR = cR * rcp; cG = cG * rcp; cB = cB * rcp;
    }
    if (st & (FSR1_COLOR_TEPD_C8 | FSR1_COLOR_TEPD_C10)) {
      // :1153-1160 FsrTepdDitHx2 is FsrTepdDitH per lane: binary32 arithmetic, narrowed once
      const half2_t dit = (st & FSR1_COLOR_DITHER_FROM_NOISE)
                              ? c_sat2(c_h2((half_t)n0.w, (half_t)n1.w))
                              : c_h2((half_t)FsrTepdDitF((uint32_t)x, (uint32_t)y, a.color.frame), (half_t)FsrTepdDitF((uint32_t)x + 1u, (uint32_t)y, a.color.frame));
      cR = FsrTepdCHx2(cR, dit, k, rk); cG = FsrTepdCHx2(cG, dit, k, rk); cB = FsrTepdCHx2(cB, dit, k, rk);
    }
    px2_t o;
    o.p[0] = half4_t{cR.x, cG.x, cB.x, q.p[0].w};
    o.p[1] = half4_t{cR.y, cG.y, cB.y, q.p[1].w};
    char* p = dst + (long long)y * a.out.pitch;
    if (pair) *reinterpret_cast<px2_t*>(p) = o;
    else *reinterpret_cast<half4_t*>(p) = o.p[0];
  }
}

hipError_t color_h_launch(const ColorPassArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  hipLaunchKernelGGL(color_h_kernel, grid, block, 0, stream, a);
  return hipGetLastError();
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
