// EASU, exact 2x, default arithmetic, with the 12-tap accumulation on the matrix pipe (gfx950 v_mfma_f32_4x4x1_16b_f32).
//
// Why: the EASU kernel is bound by VALU issue; of its 236 VALU instructions per pixel 48 are the
// multiply-adds `aC += c*w; aW += w` (ffx_fsr1.h:268-272) of the 12 taps — a [4 channels x 12 taps] x [12 taps x 4 pixels]
// product for every output quad, whose four pixels share one window at exactly 2x.  v_mfma_f32_4x4x1 computes sixteen such
// 4x4 outer-product updates per instruction (one per block of four lanes) on a pipe the kernel otherwise leaves idle, as an
// fmaf chain bit for bit, so the sums are those of the VALU kernel (fsr1_easu_kernel.h, S2) in the same tap order.
//
// Shape: the tile and its staging are the exact-2x kernel's (tile (tx, ty) = output [64 tx - 1, 64 tx + 62] x
// [16 ty - 1, 16 ty + 14] = 32 x 8 quads, footprint 35 x 11 texels) with the planar LDS layout of easu_stage_footprint_planar.
// A lane owns ONE pixel, a block of four lanes a quad (lane & 1 = column, lane & 2 = row of the pixel inside it), a wave
// iteration sixteen quads of one quad row; a wave walks its two quad rows in four iterations.  Per pixel: the bilinear
// analysis, shaping and the twelve weights on the VALU as before (sub-texel position is a per-lane constant, 1/4 or 3/4),
// twelve MFMAs, normalise, dering clamp against bounds that lanes 0-2 of the block took of their channel of f g j k
// (easu_clamp_quad_dpp), one 8-byte store (adjacent lanes: adjacent pixels).
#include "fsr1_easu_kernel.h"

namespace fsr1 {

constexpr int kMfmaFW = kTileW / 2 + 3, kMfmaFH = 16 / 2 + 3;  // 35 x 11

template <int FMT, bool HDR>
__global__ void __launch_bounds__(kThreads) easu_s2_mfma_kernel(const EasuArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
  static_assert(kTileW == 64, "64-wide tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const EasuLdsPlanar l = easu_lds_carve_planar(smem, kMfmaFW * kMfmaFH);

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;

  const int ox0 = tx * kTileW - 1, oy0 = ty * 16 - 1;
  const int fx0 = tx * (kTileW / 2) - 2 + (a.origin_x >> 1), fy0 = ty * 8 - 2 + (a.origin_y >> 1);  // (band origins are even here)
  easu_stage_footprint_planar<FMT, kMfmaFW, kMfmaFH>(l, a.in, in_frame, fx0, fy0, tid);

  const int W = a.out.width, H = a.out.height;
  const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  const int ch = lane & 3, blk = lane >> 2;
  const int px = lane & 1, py = (lane >> 1) & 1;          // this lane's pixel inside the quad
  const float ppx = px ? 0.75f : 0.25f, ppy = py ? 0.75f : 0.25f;  // ffx_fsr1.h:324-326 for con0 = {1/2, 1/2, -1/4, -1/4}
  // Everything that depends on the lane is formed once, for iteration (0, 0); the four iterations (quad row r of the wave's
  // two, left / right half h of the tile) are compile-time offsets from it.
  const int f0 = (2 * wave + 1) * kMfmaFW + (blk + 1);  // footprint index of the quad's texel 'f'
  const float* const cp0 = reinterpret_cast<const float*>(l.tex) + 4 * f0 + ch;  // this lane's channel (R, G, B, 1.0) of 'f'
  const float4_t* const ana0 = l.ana + f0;
  const int ox = ox0 + 2 * blk + px, oy = oy0 + 4 * wave + py;
  char* const o0 = a.out.base + (long long)frame * a.out.frame_stride + (long long)oy * a.out.pitch + (long long)ox * (long long)sizeof(texel_t);
  const bool xin[2] = {ox >= 0 && ox < W, ox + 32 < W}, yin[2] = {oy >= 0 && oy < H, oy + 2 < H};

#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // wave-uniform: the two pixel rows / thirty-two columns of this iteration lie outside the image (border tiles only)
      if (oy0 + 4 * wave + 2 * r >= H || ox0 + 32 * h >= W) continue;
      const int d = r * kMfmaFW + 16 * h;  // footprint offset of this iteration's quads
      auto chan = [&](int dx, int dy) { return cp0[4 * (d + dy * kMfmaFW + dx)]; };
      // dering bounds of this lane's channel over f g j k (lanes 0-2 of the block; lane 3 holds 1.0s, unused)
      const float cf = chan(0, 0), cg = chan(1, 0), cj = chan(0, 1), ck = chan(1, 1);
      const float mn = min4_asm(cf, cg, cj, ck), mx = max4_asm(cf, cg, cj, ck);
      EasuAccMfma<decltype(chan)> acc{chan, float4_t{0.f, 0.f, 0.f, 0.f}};
      auto no_tex = [](int, int) { return float4_t{0.f, 0.f, 0.f, 0.f}; };  // (the EXACT branch's texel source: not instantiated)
      rgbf_t p = easu_filter_acc<false>(no_tex, [&](int k) { return ana0[d + (k >> 1) * kMfmaFW + (k & 1)]; }, ppx, ppy, acc);
      easu_clamp_quad_dpp(p.r, p.g, p.b, mn, mx);
      if (HDR) { p.r *= p.r; p.g *= p.g; p.b *= p.b; }
      if (xin[h] && yin[r])
        store_out<sizeof(texel_t)>(o0 + (long long)(2 * r) * a.out.pitch + (long long)(32 * h) * (long long)sizeof(texel_t),
                                   Pixel<FMT>::store(pinned(p.r), pinned(p.g), pinned(p.b), 1.0f), stream);
    }
  }
}

size_t easu_mfma_lds_bytes() { return (size_t)kMfmaFW * kMfmaFH * kEasuPlanarLdsPerTexel; }

template <int FMT, bool HDR>
static hipError_t easu_mfma_launch_one(const EasuArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  hipLaunchKernelGGL((easu_s2_mfma_kernel<FMT, HDR>), grid, block, easu_mfma_lds_bytes(), stream, a);
  return hipGetLastError();
}

// The caller has checked con0 (exactly 2x) and laid the grid out for the shifted tiles, as for easu_launch(..., s2 = true).
hipError_t easu_mfma_launch(const EasuArgs& a, int fmt, hipStream_t stream) {
  const bool hdr = (a.flags & FSR1_FLAG_HDR_SQUARE) != 0;
  switch (fmt) {
#define FSR1_CASE(F) case F: return hdr ? easu_mfma_launch_one<F, true>(a, stream) : easu_mfma_launch_one<F, false>(a, stream)
    FSR1_CASE(FSR1_FORMAT_RGBA16F);
    FSR1_CASE(FSR1_FORMAT_RGBA32F);
    FSR1_CASE(FSR1_FORMAT_RGBA8_UNORM);
    FSR1_CASE(FSR1_FORMAT_R10G10B10A2_UNORM);
#undef FSR1_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fsr1
