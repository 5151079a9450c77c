// EASU -> RCAS in one launch, exact-2x variant (con0 = {1/2, 1/2, -1/4, -1/4}: 1080p -> 4K, 4K -> 8K, 540p -> 1080p).
//
// The generic fused kernel (fsr1_fused_kernel.h) runs EASU one pixel at a time on its apron tile; at exactly 2x the EASU
// kernel's quad form applies (fsr1_easu_kernel.h, S2): output pixels 2j+1 and 2j+2 share the texel f = j with sub-texel
// positions 1/4 and 3/4, so a 2x2 quad shares window address, analyses and dering bounds, and every position constant
// is a literal.  The tile is chosen so that its apron is a whole number of quads per lane and of columns per wave:
//   output tile   62 x (2 QH - 2) pixels at (62 tx, (2 QH - 2) ty)
//   apron tile    64 x 2 QH pixels from (62 tx - 1, (2 QH - 2) ty - 1): odd origin = quad-aligned, 32 x QH quads = QH / 8 per lane
//   footprint     35 x (QH + 3) texels from (31 tx - 2, (QH - 1) ty - 2)
// Phase 4 (RCAS) has lane L own apron column L: lanes 1 .. 62 are the tile's output columns and lanes 0 / 63 the apron
// columns themselves, so both horizontal neighbours of every output pixel arrive by DPP wave shifts and nothing but the
// lane's own column is read from LDS.  Same per-pixel functions on the same values as the generic kernel and as the two
// dispatches: bit-identical output (tests/test_gpu_parity.py, test_gpu_fullframe.py).  Row bands with an even origin
// (tests/test_gpu_bands.py) run this kernel too.
#include "fsr1_device.h"
#include "fsr1_device_easu.hpp"
#include "fsr1_device_rcas.hpp"

namespace fsr1 {

constexpr int kFs2OutW = 62, kFs2MidW = 64, kFs2FpW = 35;

size_t fused_s2_lds_bytes(int fmt, int qh) {
  const size_t texel = fmt == FSR1_FORMAT_RGBA32F ? 16 : (fmt == FSR1_FORMAT_RGBA16F ? 8 : 4);
  return easu_lds_region_bytes((size_t)kFs2FpW * (qh + 3)) + (size_t)kFs2MidW * 2 * qh * texel;
}

void fused_s2_geometry(int width, int height, int qh, int* tiles_x, int* tiles_y) {
  *tiles_x = (width + kFs2OutW - 1) / kFs2OutW;
  *tiles_y = (height + (2 * qh - 2) - 1) / (2 * qh - 2);
}

template <int FMT, bool EXACT, int QH>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(7, 8))) fused_s2_kernel(const FusedArgs a) {  // (<= 72 VGPRs: 7 workgroups per CU is what its LDS admits)
  typedef typename Pixel<FMT>::T texel_t;
  static_assert(QH % 8 == 0 && kThreads == 256, "32 x QH quads over 256 lanes");
  constexpr int kFpH = QH + 3, kOutH = 2 * QH - 2;  // (the apron tile is 2 QH rows tall)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EasuLds l = easu_lds_carve(smem, kFs2FpW * kFpH);
  texel_t* const mid = reinterpret_cast<texel_t*>(smem + easu_lds_region_bytes(kFs2FpW * kFpH));  // [kMidH][64]

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int W = a.out.width, H = a.out.height;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  l.fw = kFs2FpW;
  // (a band of a larger output image, fsr1_easu_rcas_fused_dispatch_band: output row 0 is row a.origin_y — even, the host
  //  checks — of the image the constants describe, and the rows just above / below the band exist there when rows_above /
  //  rows_below say so: the apron computes them instead of taking them as outside the image)
  easu_stage_footprint<FMT, false, EXACT, kFs2FpW, kFpH>(l, a.in, a.in.base + (long long)frame * a.in.frame_stride, 31 * tx - 2,
                                                         (QH - 1) * ty - 2 + (a.origin_y >> 1), kFs2FpW, kFpH, tid);
  const int ylo = -a.rows_above, yhi = H - 1 + a.rows_below;

  // ---- phase 3: EASU on the apron tile, a quad per lane and iteration, rounded to the storage format (EASU runs with
  //      Sample.x = 0 when RCAS follows: FSR_Filter.cpp:107); pixels outside the image are 0 (FSR_Pass.hlsl:45,61) ----
  const int ax0 = kFs2OutW * tx - 1, ay0 = kOutH * ty - 1;  // both odd
#pragma unroll 1
  for (int k = 0; k < QH / 8; ++k) {
    const int qx = lane & 31, qy = 8 * k + 2 * wave + (lane >> 5);
    const int oxa = ax0 + 2 * qx, oya = ay0 + 2 * qy;
    const bool xin0 = oxa >= 0 && oxa < W, xin1 = oxa + 1 < W, yin0 = oya >= ylo && oya <= yhi, yin1 = oya + 1 <= yhi;
    const int f_idx = (qy + 1) * kFs2FpW + (qx + 1);
    texel_t* const m0 = mid + (2 * qy) * kFs2MidW + 2 * qx;
    typedef typename TexelPair<FMT>::T pair_t;
    if (!EXACT && xin0 && xin1 && yin0 && yin1) {  // default arithmetic: two row pairs, every texel read once per row (see easu_kernel)
      const float4_t A[4] = {l.ana[f_idx], l.ana[f_idx + 1], l.ana[f_idx + kFs2FpW], l.ana[f_idx + kFs2FpW + 1]};
      EasuBounds m;
      rgbf_t q00, q10, q01, q11;
      easu_quad_row<true>(l, f_idx, 0.25f, A, m, q00, q10);
      *reinterpret_cast<pair_t*>(m0) = TexelPair<FMT>::make(easu_resolve<FMT, EXACT>(m, q00, false), easu_resolve<FMT, EXACT>(m, q10, false));
      easu_quad_row<false>(l, f_idx, 0.75f, A, m, q01, q11);
      *reinterpret_cast<pair_t*>(m0 + kFs2MidW) = TexelPair<FMT>::make(easu_resolve<FMT, EXACT>(m, q01, false), easu_resolve<FMT, EXACT>(m, q11, false));
    } else if (xin0 && xin1 && yin0 && yin1) {  // every quad but those on the image's border
      const EasuBounds m = easu_bounds(l, f_idx);
      const texel_t p00 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.25f), false);
      const texel_t p10 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.25f), false);
      const texel_t p01 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.75f), false);
      const texel_t p11 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.75f), false);
      *reinterpret_cast<pair_t*>(m0) = TexelPair<FMT>::make(p00, p10);
      *reinterpret_cast<pair_t*>(m0 + kFs2MidW) = TexelPair<FMT>::make(p01, p11);
    } else {
      texel_t p[4] = {Pixel<FMT>::zero(), Pixel<FMT>::zero(), Pixel<FMT>::zero(), Pixel<FMT>::zero()};
      if ((xin0 || xin1) && (yin0 || yin1)) {
        const EasuBounds m = easu_bounds(l, f_idx);
        if (xin0 && yin0) p[0] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.25f), false);
        if (xin1 && yin0) p[1] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.25f), false);
        if (xin0 && yin1) p[2] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.75f), false);
        if (xin1 && yin1) p[3] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.75f), false);
      }
      m0[0] = p[0]; m0[1] = p[1]; m0[kFs2MidW] = p[2]; m0[kFs2MidW + 1] = p[3];
    }
  }
  __syncthreads();

  // ---- phase 4: RCAS from the LDS tile.  Lane L owns apron column L; a wave walks down its share of the kOutH rows with
  //      b / e / h in registers, d and f are the neighbouring lanes' centre texels (DPP wave shifts).  Every lane stays
  //      active (DPP sources); lanes 0 / 63 and pixels outside the image store nothing. ----
  constexpr int kRowsLo = kOutH / 4, kExtra = kOutH % 4;  // the first kExtra waves take one row more
  const int nrows = kRowsLo + (wave < kExtra ? 1 : 0);
  const int ry0 = wave * kRowsLo + min(wave, kExtra);
  const int ox = ax0 + lane;
  const float sharp = as_f32(a.rcas_con[0]);
  const uint32_t flags = a.flags;
  const bool stream = (flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  const bool col_ok = lane >= 1 && lane <= kFs2OutW && ox < W;
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (long long)ox * (long long)sizeof(texel_t);
  auto rgb = [](const texel_t& p) { const float4_t c = Pixel<FMT>::load(p); return rgb_t{c.x, c.y, c.z}; };
  const texel_t* c = mid + (ry0 + 1) * kFs2MidW + lane;  // centre texel of this lane's first row
  rgb_t prev = rgb(c[-kFs2MidW]);
  texel_t e_raw = c[0];
  rgb_t cur = rgb(e_raw);
#pragma unroll
  for (int r = 0; r < kRowsLo + (kExtra ? 1 : 0); ++r, c += kFs2MidW) {
    if (r >= nrows) break;  // wave-uniform
    const int oy = ay0 + 1 + ry0 + r;
    const texel_t n_raw = c[kFs2MidW];
    const rgb_t next = rgb(n_raw);
    const rgb_t d = rgb_t{dpp_f32<kDppWaveShr1>(cur.r, cur.r), dpp_f32<kDppWaveShr1>(cur.g, cur.g), dpp_f32<kDppWaveShr1>(cur.b, cur.b)};
    const rgb_t f = rgb_t{dpp_f32<kDppWaveShl1>(cur.r, cur.r), dpp_f32<kDppWaveShl1>(cur.g, cur.g), dpp_f32<kDppWaveShl1>(cur.b, cur.b)};
    const rgb_t p = rcas_pixel<EXACT>(prev, d, cur, f, next, sharp, flags);
    if (col_ok && oy < H) {
      const float pa = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) ? Pixel<FMT>::load(e_raw).w : 1.0f;
      store_out<sizeof(texel_t)>(out_col + (long long)oy * a.out.pitch, Pixel<FMT>::store(p.r, p.g, p.b, pa), stream);
    }
    prev = cur; cur = next; e_raw = n_raw;
  }
}

template <int FMT, bool EXACT, int QH>
static hipError_t fused_s2_launch_one(const FusedArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = fused_s2_lds_bytes(FMT, QH);
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fused_s2_kernel<FMT, EXACT, QH>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((fused_s2_kernel<FMT, EXACT, QH>), grid, block, lds, stream, a);
  return hipGetLastError();
}

#ifndef FSR1_FUSED_S2_QH
#define FSR1_FUSED_S2_QH 8
#endif
int fused_s2_quad_rows() { return FSR1_FUSED_S2_QH; }

hipError_t fused_s2_launch(const FusedArgs& a, int fmt, bool exact, hipStream_t stream) {
#define FSR1_LAUNCH_E(F) return exact ? fused_s2_launch_one<F, true, FSR1_FUSED_S2_QH>(a, stream) : fused_s2_launch_one<F, false, FSR1_FUSED_S2_QH>(a, stream)
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
