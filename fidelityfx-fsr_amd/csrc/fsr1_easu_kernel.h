// EASU kernel template (see fsr1_easu.hip for the design notes); instantiated by fsr1_easu.hip (plain pass) and
// fsr1_easu_color.hip (colour prologue / epilogue variants).
#pragma once
#include "fsr1_device.h"
#include "fsr1_device_easu.hpp"

namespace fsr1 {

size_t easu_lds_bytes(int fmt, int fp_w, int fp_h);
int easu_lds_pitch(int fp_w, bool exact, bool color);

// COLOR: colour stages fused in (fsr1_device_color.hpp) — FsrSrtmF on every input texel as it is loaded, and
// FsrLfgaF / FsrSrtmInvF / FsrTepdC*F on the result before it is stored as FOUT.  COLOR = false is the plain pass
// (FOUT == FMT), compiled without any of it.
//
// S2: the scale is exactly 2x with the viewport covering the input (con0 = {1/2, 1/2, -1/4, -1/4}).  Output pixels
// 2j+1 and 2j+2 then share the input texel f = j, with sub-texel positions 1/4 and 3/4 (ffx_fsr1.h:324-326 is exact for
// these constants), per axis.  Tiles are shifted by one pixel — tile (tx, ty) covers [64 tx - 1, 64 tx + 62] x
// [16 ty - 1, 16 ty + 14] — so that each holds exactly 32 x 8 such quads, one per lane: window address and position
// arithmetic are done once per four pixels and (ppx, ppy) are compile-time constants, which folds the bilinear weights
// and the tap offsets of easu_pixel (-12 % VALU instructions, same four-pixels-per-lane balance).  The arithmetic per
// pixel is the same function on the same values: bit-identical to S2 = false (tests/test_gpu_parity.py).
// PITCH (generic variant only): 0 = dense LDS arrays of the tile's own footprint width; P = the row-interleaved layout with
// the compile-time pitch P >= a.fp_w (easu_lds_carve_pitched): no LDS address arithmetic per tap row.
// TH: rows of the output tile (16; 32 for exact-2x launches that are large or overlap other frames' launches, and for the generic
// kernel's 512-thread form, fsr1_easu.hip).
// WAVES: waves of the workgroup (4; 8 = the generic kernel on 64 x 32 tiles, round 5): a wave always owns TH / WAVES = 4 rows of the
// tile.  Why eight waves: the generic kernel is LDS-bound in workgroups per CU (a 1.5x tile's footprint is 23 KB for four waves: 7
// workgroups; 1.3x: 30 KB, 5) and loses 2-2.5 % per workgroup it cannot hold (profiles/ab_r05/r5c6_ab_ldspad.log).  A 64 x 32 tile shares
// its four apron rows between twice as many pixels — 40 KB for EIGHT waves at 1.5x, four workgroups = 8 waves per SIMD — and stages
// a fifth less per pixel.  (The same tile with four waves, round 4, halved the waves per CU instead and lost.)
// STRICT (F-strict, FSR1_FLAG_MATH_STRICT; plain pass, default arithmetic): the default arithmetic's pixels are tested against the
// store conversion's rounding boundaries, the ones that fail are queued in LDS and re-evaluated in the reference's operation order after
// the tile's default pass — the stored image is bit-identical to EXACT's (fsr1_device_easu.hpp, "F-strict").  The queue sits behind the
// footprint region (easu_strict_lds_bytes).
template <int FMT, bool EXACT, bool COLOR = false, int FOUT = FMT, bool S2 = false, bool HDR = false, int PITCH = 0, int TH = kTileH, int WAVES = 4, bool STRICT = false>
// amdgpu_waves_per_eu(7, 8): at least seven waves per SIMD, i.e. at most 72 VGPRs.  Only the exact-2x default-arithmetic
// variant is affected — its row-pair form would take 85 (five waves: 43.7 us) where 68 cost it nothing (41.6 us); every other
// variant needs fewer than 64 anyway.
// (F-strict on 64 x 32 exact-2x tiles: footprint + queue = 26.4 KB, six workgroups per CU — and six waves per SIMD's 80 registers, which its
//  re-evaluation loop beside the two-step pixel loop needs)
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(STRICT && S2 && TH == 32 ? 6 : 7, 8))) easu_kernel(const EasuArgs a) {
  typedef typename Pixel<FOUT>::T texel_t;
  static_assert(TH % 16 == 0, "a wave filters two quad rows per 16 tile rows");
  static_assert(WAVES == 4 || (WAVES == 8 && TH == 32 && !S2), "eight waves: the generic kernel's 64 x 32 tile");
  constexpr int kThreads = 64 * WAVES, kRowsPerWave = TH / WAVES;  // (shadows the default workgroup size)
  constexpr int kTileH = TH;  // (shadows the default tile height)
  constexpr bool kS2 = S2;
  constexpr int kS2W = kTileW / 2 + 3, kS2H = kTileH / 2 + 3;  // footprint of every exact-2x tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!(S2 && PITCH), "the exact-2x variant has a compile-time footprint of its own");
  static_assert(!STRICT || (!EXACT && !COLOR && !HDR && FOUT == FMT), "F-strict: the plain pass's default arithmetic");
  EasuLds l = PITCH ? easu_lds_carve_pitched<PITCH ? PITCH : 1>(smem) : easu_lds_carve(smem, kS2 ? kS2W * kS2H : a.fp_w * a.fp_h);
  // (STRICT) the queue behind the footprint region
  const EasuStrictQueue sq = easu_strict_queue_carve(smem + easu_lds_region_bytes(kS2 ? (size_t)kS2W * kS2H : (PITCH ? (size_t)PITCH * a.fp_h : (size_t)a.fp_w * a.fp_h)));

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;

  if constexpr (kS2) {
    // ---- exact 2x: tile (tx, ty) = output pixels [64 tx - 1, 64 tx + 62] x [16 ty - 1, 16 ty + 14] = 32 x 8 quads; quad
    //      (qx, qy) shares the texel f = (32 tx - 1 + qx, 8 ty - 1 + qy), so the footprint starts two texels before ----
    static_assert(!COLOR && kTileW == 64, "the exact-2x variant is built for the plain 64-wide tiles");
    const int ox0 = tx * kTileW - 1, oy0 = ty * kTileH - 1;
    const int fx0 = tx * (kTileW / 2) - 2 + (a.origin_x >> 1), fy0 = ty * (kTileH / 2) - 2 + (a.origin_y >> 1);  // (band origins are even here)
    l.fw = kS2W;
    easu_stage_footprint<FMT, false, EXACT, kS2W, kS2H, 256, 0, STRICT>(l, a.in, in_frame, fx0, fy0, kS2W, kS2H, tid);
    const int W = a.out.width, H = a.out.height;
    constexpr bool hdr = HDR;
    const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
    uint32_t redo_all = 0;  // (F-strict) bit 4 k + b: pixel b (= x + 2 y) of this lane's quad of pass k has to be re-evaluated
#pragma unroll 1
    for (int k = 0; k < kTileH / 16; ++k) {
      const int qx = lane & 31, qy = (kTileH / 8) * wave + 2 * k + (lane >> 5);
      const int oxa = ox0 + 2 * qx, oya = oy0 + 2 * qy;  // odd (ox0, oy0 are): the quad is {oxa, oxa+1} x {oya, oya+1}
      const bool xin0 = oxa >= 0 && oxa < W, xin1 = oxa + 1 < W, yin0 = oya >= 0 && oya < H, yin1 = oya + 1 < H;
      if (!((xin0 || xin1) && (yin0 || yin1))) continue;
      const int f_idx = (qy + 1) * kS2W + (qx + 1);
      char* const o0 = a.out.base + (long long)frame * a.out.frame_stride + (long long)oya * a.out.pitch + (long long)oxa * (long long)sizeof(texel_t);
      if constexpr (STRICT) {
        // F-strict: the default arithmetic's row pairs with the rounding-boundary test; bit b of `redo` = pixel b of the quad (x + 2 y)
        if (xin0 && xin1 && yin0 && yin1) {
          const float4_t A[4] = {l.ana[f_idx], l.ana[f_idx + 1], l.ana[f_idx + kS2W], l.ana[f_idx + kS2W + 1]};
          const float e1 = easu_strict_eps(A[0], A[1], A[2], A[3]);
          EasuBounds m;
          rgbf_t q00, q10, q01, q11;
          texel_t p00, p10, p01, p11;
          easu_quad_row<true>(l, f_idx, 0.25f, A, m, q00, q10);
          const EasuStrictEps e = easu_strict_eps_rgb(m, e1);
          uint32_t redo = easu_strict_resolve<FMT>(m, q00, e, p00) ? 1u : 0u;
          redo |= easu_strict_resolve<FMT>(m, q10, e, p10) ? 2u : 0u;
          store_out<sizeof(texel_t)>(o0, TexelPair<FOUT>::make(p00, p10), stream);
          easu_quad_row<false>(l, f_idx, 0.75f, A, m, q01, q11);
          redo |= easu_strict_resolve<FMT>(m, q01, e, p01) ? 4u : 0u;
          redo |= easu_strict_resolve<FMT>(m, q11, e, p11) ? 8u : 0u;
          store_out<sizeof(texel_t)>(o0 + a.out.pitch, TexelPair<FOUT>::make(p01, p11), stream);
          redo_all |= redo << (4 * k);
        } else {  // (border quads) straight to the queue
          redo_all |= ((xin0 && yin0 ? 1u : 0u) | (xin1 && yin0 ? 2u : 0u) | (xin0 && yin1 ? 4u : 0u) | (xin1 && yin1 ? 8u : 0u)) << (4 * k);
        }
        continue;
      }
      if constexpr (!EXACT) {
        // default arithmetic: a quad is filtered as two ROW PAIRS — the two pixels of a row share the 12-tap window, so every
        // texel is read from LDS once per row and used for both (28 instead of 60 ds_read_b128 per quad: the four analyses once
        // per quad, the dering bounds taken of the first row's tap values).  Same statements per pixel as easu_pixel: bit-identical.
        // LDS cycles matter well before the LDS is "the" limiter: 43.7 -> 41.6 us at 1080p -> 4K.  (EXACT keeps
        // the per-pixel form: its pair would need more than a hundred VGPRs.)
        if (xin0 && xin1 && yin0 && yin1) {
          const float4_t A[4] = {l.ana[f_idx], l.ana[f_idx + 1], l.ana[f_idx + kS2W], l.ana[f_idx + kS2W + 1]};
          EasuBounds m;
          rgbf_t q00, q10, q01, q11;
          easu_quad_row<true>(l, f_idx, 0.25f, A, m, q00, q10);
          store_out<sizeof(texel_t)>(o0, TexelPair<FOUT>::make(easu_resolve<FMT, EXACT>(m, q00, hdr), easu_resolve<FMT, EXACT>(m, q10, hdr)), stream);
          easu_quad_row<false>(l, f_idx, 0.75f, A, m, q01, q11);
          store_out<sizeof(texel_t)>(o0 + a.out.pitch, TexelPair<FOUT>::make(easu_resolve<FMT, EXACT>(m, q01, hdr), easu_resolve<FMT, EXACT>(m, q11, hdr)), stream);
          continue;
        }
      }
      if (EXACT && xin0 && xin1 && yin0 && yin1) {
        // EXACT: the per-pixel form, with the quad's four analyses loaded once and its bounds taken of the first pixel's taps (52
        // instead of 60 LDS reads per quad, 70 VGPRs: -1 .. -2 %, profiles/ab_r03/r3c16_exact_shared_analyses_ab.log); its row
        // pair would not fit the register budget.  (No predicates: the whole quad lies inside the image.)
        const float4_t A[4] = {l.ana[f_idx], l.ana[f_idx + 1], l.ana[f_idx + kS2W], l.ana[f_idx + kS2W + 1]};
        EasuBounds m;
        const rgbf_t q00 = easu_quad_pixel<EXACT, true>(l, f_idx, 0.25f, 0.25f, A, m);
        const texel_t p00 = easu_resolve<FMT, EXACT>(m, q00, hdr);
        const texel_t p10 = easu_resolve<FMT, EXACT>(m, easu_quad_pixel<EXACT, false>(l, f_idx, 0.75f, 0.25f, A, m), hdr);
        const texel_t p01 = easu_resolve<FMT, EXACT>(m, easu_quad_pixel<EXACT, false>(l, f_idx, 0.25f, 0.75f, A, m), hdr);
        const texel_t p11 = easu_resolve<FMT, EXACT>(m, easu_quad_pixel<EXACT, false>(l, f_idx, 0.75f, 0.75f, A, m), hdr);
        store_out<sizeof(texel_t)>(o0, TexelPair<FOUT>::make(p00, p10), stream);
        store_out<sizeof(texel_t)>(o0 + a.out.pitch, TexelPair<FOUT>::make(p01, p11), stream);
        continue;
      }
      const EasuBounds m = easu_bounds(l, f_idx);  // (border quads) one 2x2 block for the whole quad
      auto row = [&](char* o, bool yin, float ppy) {
        if (!yin) return;
        if (xin0) store_out<sizeof(texel_t)>(o, easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, ppy), hdr), stream);
        if (xin1) store_out<sizeof(texel_t)>(o + sizeof(texel_t), easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, ppy), hdr), stream);
      };
      row(o0, yin0, 0.25f);
      row(o0 + a.out.pitch, yin1, 0.75f);
    }
    if constexpr (STRICT) {
      // pixel id: bits 0-1 = position in the quad, 2-3 (one bit at 64 x 16) = pass k, then lane and wave.  The queued pixels are
      // re-evaluated densely over the workgroup's first lanes, in the reference's operation order
      easu_strict_rounds<kThreads, kTileH / 4>(sq, redo_all, [&](int b) { return tid * (kTileH / 4) + b; }, easu_strict_queue_capacity(kTileW * kTileH), tid, [&](int id) {
        const int sub = id & 3, k = (id >> 2) & (kTileH / 16 - 1), t = id / (kTileH / 4);
        const int qx = t & 31, qy = (kTileH / 8) * (t >> 6) + 2 * k + ((t >> 5) & 1);
        const int ox = ox0 + 2 * qx + (sub & 1), oy = oy0 + 2 * qy + (sub >> 1);
        const int f_idx = (qy + 1) * kS2W + (qx + 1);
        char* const o = a.out.base + (long long)frame * a.out.frame_stride + (long long)oy * a.out.pitch + (long long)ox * (long long)sizeof(texel_t);
        store_out<sizeof(texel_t)>(o, easu_strict_pixel<FMT>(l, f_idx, (sub & 1) ? 0.75f : 0.25f, (sub >> 1) ? 0.75f : 0.25f), stream);
      });
    }
    return;
  }

  const int ox0 = tx * kTileW, oy0 = ty * kTileH;
  const float c0x = as_f32(a.con[0]), c0y = as_f32(a.con[1]), c0z = as_f32(a.con[2]), c0w = as_f32(a.con[3]);
  // Footprint of this tile: fp(first pixel)-1 .. fp(last pixel)+2 per axis (ffx_fsr1.h:324-342).
  // Same arithmetic as the per-pixel position below, and x -> x*c+b is monotone under rounding.
  const int oxl = min(ox0 + kTileW, a.out.width) - 1, oyl = min(oy0 + kTileH, a.out.height) - 1;
  // (a.origin: `out` may be a band of the full output image — positions are those of the full image)
  const int gx0 = ox0 + a.origin_x, gy0 = oy0 + a.origin_y;
  const int fx0 = (int)floorf((float)gx0 * c0x + c0z) - 1;
  const int fy0 = (int)floorf((float)gy0 * c0y + c0w) - 1;
  const int fw = min((int)floorf((float)(oxl + a.origin_x) * c0x + c0z) + 2 - fx0 + 1, a.fp_w);
  const int fh = min((int)floorf((float)(oyl + a.origin_y) * c0y + c0w) + 2 - fy0 + 1, a.fp_h);
  if (!PITCH) l.fw = fw;
  const int row_stride = PITCH ? 2 * PITCH : fw;  // LDS records between footprint rows
  easu_stage_footprint<FMT, COLOR, EXACT, 0, 0, kThreads, PITCH, STRICT>(l, a.in, in_frame, fx0, fy0, fw, fh, tid, &a.color);

  // ---- phase 3: output pixels; a lane owns a column, a wave kRowsPerWave = 4 rows.  Which column: easu_lane_column — sixteen
  //      consecutive columns per LDS lane group, so that a group's texels stay within sixteen records (no bank conflicts) ----
  const int col = easu_lane_column(lane);
  const int ox = ox0 + col;
  const bool col_in = ox < a.out.width;
  if (!STRICT && !col_in) return;  // (F-strict: every lane stays for the barrier before the queue is drained)
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)ox * sizeof(texel_t);
  // :324-326 (x part, shared by this lane's rows)
  float ppx = (float)(ox + a.origin_x) * c0x + c0z;
  const float fpx = floorf(ppx);
  ppx -= fpx;
  const int lx = (int)fpx - fx0;  // footprint column of texel 'f'
  // `c *= c` (FSR_Pass.hlsl:78-79) is a template parameter of the plain kernels (a run-time flag costs three multiplies
  // and three selects per pixel for an option only the EASU-only HDR path uses); the colour variants read the flag
  const bool hdr = COLOR ? (a.flags & FSR1_FLAG_HDR_SQUARE) != 0 : HDR;
  const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;

  uint32_t redo = 0;  // (F-strict) bit r: this lane's pixel of row r has to be re-evaluated
#pragma unroll 1
  for (int r = 0; r < kRowsPerWave; ++r) {
    const int oy = oy0 + wave * kRowsPerWave + r;
    if (oy >= a.out.height || (STRICT && !col_in)) break;
    float ppy = (float)(oy + a.origin_y) * c0y + c0w;  // :324-326
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const EasuRowTerms yt = easu_row_terms(ppy);
    const int f_idx = ((int)fpy - fy0) * row_stride + lx;
    EasuBounds m;
    texel_t* const dst = reinterpret_cast<texel_t*>(out_col + (long long)oy * a.out.pitch);
    if constexpr (STRICT) {
      float e;
      const rgbf_t p = easu_pixel_with_bounds<false>(l, f_idx, ppx, yt, m, &e);
      texel_t px;
      redo |= easu_strict_resolve<FMT>(m, p, easu_strict_eps_rgb(m, e), px) ? 1u << r : 0u;
      store_out<sizeof(texel_t)>(dst, px, stream);
      continue;
    }
    const rgbf_t p = easu_pixel_with_bounds<EXACT>(l, f_idx, ppx, yt, m);
    if constexpr (COLOR) {
      rgbf_t q = easu_clamp<EXACT>(m, p, hdr);
      color_epilogue<EXACT>(a.color, (uint32_t)ox, (uint32_t)oy, q.r, q.g, q.b);
      store_out<sizeof(texel_t)>(dst, Pixel<FOUT>::store(q.r, q.g, q.b, 1.0f), stream);
    } else {
      store_out<sizeof(texel_t)>(dst, easu_resolve<FMT, EXACT>(m, p, hdr), stream);
    }
  }
  if constexpr (STRICT) {
    // group = (wave, column): the lane's four rows; pixel id = 4 * group + row.  The queued pixels densely over the workgroup's first
    // lanes, in the reference's operation order
    static_assert(S2 || kRowsPerWave == 4, "a group is a lane's four rows");
    easu_strict_rounds<kThreads, 4>(sq, redo, [&](int b) { return 4 * (wave * 64 + col) + b; }, easu_strict_queue_capacity(kTileW * kTileH), tid, [&](int id) {
      const int qx = ox0 + ((id >> 2) & 63), qy = oy0 + 4 * (id >> 8) + (id & 3);
      float px = (float)(qx + a.origin_x) * c0x + c0z, py = (float)(qy + a.origin_y) * c0y + c0w;  // :324-326, as above
      const float fx = floorf(px), fy = floorf(py);
      px -= fx;
      py -= fy;
      const int f_idx = ((int)fy - fy0) * row_stride + ((int)fx - fx0);
      char* const o = a.out.base + (long long)frame * a.out.frame_stride + (long long)qy * a.out.pitch + (size_t)qx * sizeof(texel_t);
      store_out<sizeof(texel_t)>(o, easu_strict_pixel<FMT>(l, f_idx, px, py), stream);
    });
  }
}

template <int FMT, bool EXACT, bool COLOR, int FOUT, bool S2 = false, bool HDR = false, int PITCH = 0, int TH = kTileH, int WAVES = 4, bool STRICT = false>
hipError_t easu_launch_one(const EasuArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(64 * WAVES);
  const size_t lds = easu_lds_bytes(FMT, PITCH ? PITCH : a.fp_w, a.fp_h) + (STRICT ? easu_strict_queue_bytes((size_t)kTileW * TH) : 0);
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&easu_kernel<FMT, EXACT, COLOR, FOUT, S2, HDR, PITCH, TH, WAVES, STRICT>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((easu_kernel<FMT, EXACT, COLOR, FOUT, S2, HDR, PITCH, TH, WAVES, STRICT>), grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace fsr1
