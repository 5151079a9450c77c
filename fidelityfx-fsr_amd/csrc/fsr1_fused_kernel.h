// Fused EASU->RCAS kernel template (see fsr1_fused.hip for the design notes); instantiated by fsr1_fused.hip (plain)
// and fsr1_fused_color.hip (colour prologue / epilogue variants).
#pragma once
#include "fsr1_device.h"
#include "fsr1_device_easu.hpp"
#include "fsr1_device_rcas.hpp"

namespace fsr1 {

size_t fused_lds_bytes(int fmt, int fp_w, int fp_h);

constexpr int kMidW = kTileW + 2;
constexpr int kMidH = kFusedTileH + 2;

// COLOR: colour stages fused in (fsr1_device_color.hpp) — FsrSrtmF on every input texel as it is loaded (prologue of
// EASU), FsrLfgaF / FsrSrtmInvF / FsrTepdC*F on the RCAS result before it is stored as FOUT.  The EASU->RCAS
// intermediary in LDS keeps the input's format FMT.  COLOR = false is the plain kernel (FOUT == FMT).
// STRICT (F-strict; plain kernel, default arithmetic): the apron tile's EASU pixels are tested against the store conversion's rounding
// boundaries and the ones that fail re-evaluated in the reference's operation order into the LDS tile before RCAS reads it; the RCAS half
// runs the default arithmetic (include/fsr1_hip.h, FSR1_FLAG_MATH_STRICT).  The queue sits behind the tile (fused_strict_lds_bytes).
template <int FMT, bool EXACT, bool COLOR = false, int FOUT = FMT, bool STRICT = false>
__global__ void __launch_bounds__(kThreads) fused_kernel(const FusedArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
  typedef typename Pixel<FOUT>::T out_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = a.fp_w * a.fp_h;
  EasuLds l = easu_lds_carve(smem, cap);
  texel_t* const mid = reinterpret_cast<texel_t*>(smem + easu_lds_region_bytes(cap));  // [kMidH][kMidW]
  static_assert(!STRICT || (!EXACT && !COLOR), "F-strict: the plain kernel's default arithmetic");
  const EasuStrictQueue sq = easu_strict_queue_carve(reinterpret_cast<char*>(mid) + ((sizeof(texel_t) * kMidW * kMidH + 15) & ~(size_t)15));  // (STRICT)

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int ox0 = tx * kTileW, oy0 = ty * kFusedTileH;
  const int W = a.out.width, H = a.out.height;

  const float c0x = as_f32(a.easu_con[0]), c0y = as_f32(a.easu_con[1]), c0z = as_f32(a.easu_con[2]), c0w = as_f32(a.easu_con[3]);

  // apron tile = output pixels [ox0-1, ox0+64] x [oy0-1, oy0+16], clipped to the image for the footprint (a band's image
  // reaches one row beyond the band where rows_above / rows_below say so; rows are positioned in the full image)
  const int ylo = -a.rows_above, yhi = H - 1 + a.rows_below, yorg = a.origin_y;
  const int ax0 = max(ox0 - 1, 0), ay0 = max(oy0 - 1, ylo);
  const int axl = min(ox0 + kTileW, W - 1), ayl = min(oy0 + kFusedTileH, yhi);
  const int fx0 = (int)floorf((float)ax0 * c0x + c0z) - 1;
  const int fy0 = (int)floorf((float)(ay0 + yorg) * c0y + c0w) - 1;
  const int fw = min((int)floorf((float)axl * c0x + c0z) + 2 - fx0 + 1, a.fp_w);
  const int fh = min((int)floorf((float)(ayl + yorg) * c0y + c0w) + 2 - fy0 + 1, a.fp_h);
  l.fw = fw;

  const int tid = threadIdx.x;
  easu_stage_footprint<FMT, COLOR, EXACT, 0, 0, 256, 0, STRICT>(l, a.in, a.in.base + (long long)frame * a.in.frame_stride, fx0, fy0, fw, fh, tid, &a.color);

  // ---- phase 3: EASU on the apron tile -> LDS, in the storage format (EASU runs with Sample.x = 0 when
  //      RCAS follows: FSR_Filter.cpp:107).  A lane owns apron column `lane` (its x position math is done once),
  //      the waves share the kMidH rows; the two columns left over (64, 65) are one extra partial pass. ----
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t redo = 0;  // (F-strict) this lane's pixels to re-evaluate
  int redo_col = 0;
  // (returns, F-strict only, whether the pixel has to be re-evaluated)
  auto easu_to_mid = [&](int mx, int my, float ppx, int lxf, bool x_ok) {
    const int oy = oy0 - 1 + my;
    texel_t px = Pixel<FMT>::zero();
    bool again = false;
    if (x_ok && oy >= ylo && oy <= yhi) {
      float ppy = (float)(oy + yorg) * c0y + c0w;  // :324-326
      const float fpy = floorf(ppy);
      ppy -= fpy;
      const int f_idx = ((int)fpy - fy0) * fw + lxf;
      EasuBounds m;  // taken of the taps as they arrive: no second read of f g j k
      if constexpr (STRICT) {
        float e;
        const rgbf_t p = easu_pixel_with_bounds<false>(l, f_idx, ppx, easu_row_terms(ppy), m, &e);
        again = easu_strict_resolve<FMT>(m, p, easu_strict_eps_rgb(m, e), px);
      } else {
        const rgbf_t p = easu_pixel_with_bounds<EXACT>(l, f_idx, ppx, easu_row_terms(ppy), m);
        px = easu_resolve<FMT, EXACT>(m, p, false);
      }
    }
    mid[my * kMidW + mx] = px;
    return again;
  };
  auto x_position = [&](int mx, float& ppx, int& lxf) {
    const int ox = ox0 - 1 + mx;
    ppx = (float)ox * c0x + c0z;
    const float fpx = floorf(ppx);
    ppx -= fpx;
    lxf = (int)fpx - fx0;
    return ox >= 0 && ox < W;
  };
  {
    float ppx;
    int lxf;
    const int mx = easu_lane_column(lane);  // sixteen consecutive columns per LDS lane group: no bank conflicts on the window reads
    const bool x_ok = x_position(mx, ppx, lxf);
#pragma unroll 1
    for (int my = wave; my < kMidH; my += 4) redo |= easu_to_mid(mx, my, ppx, lxf, x_ok) ? 1u << (my >> 2) : 0u;
    redo_col = mx;
  }
  // leftover columns: 2 x kMidH pixels, given to the last wave (it has the fewest rows above when kMidH % 4 == 2)
  static_assert(2 * kMidH <= 64 && kMidH <= 4 * 7, "one leftover pixel per lane of the last wave; bit 7 of the F-strict mask is free for it");
  if (wave == 3) {
    for (int t = lane; t < 2 * kMidH; t += 64) {
      float ppx;
      int lxf;
      const int mx = kTileW + (t & 1), my = t >> 1;
      const bool x_ok = x_position(mx, ppx, lxf);
      redo |= easu_to_mid(mx, my, ppx, lxf, x_ok) ? 1u << 7 : 0u;
    }
  }
  if constexpr (STRICT) {
    // pixel id = its index in the LDS tile.  Bit b < 7 of a lane's mask: row wave + 4 b of its column; bit 7: the last wave's leftover pixel
    easu_strict_rounds<kThreads, 8>(
        sq, redo, [&](int b) { return b == 7 ? (lane >> 1) * kMidW + kTileW + (lane & 1) : (wave + 4 * b) * kMidW + redo_col; },
        easu_strict_queue_capacity(kMidW * kMidH), tid, [&](int id) {
          const int my = id / kMidW, mx = id - my * kMidW;
          float ppx;
          int lxf;
          x_position(mx, ppx, lxf);
          float ppy = (float)(oy0 - 1 + my + yorg) * c0y + c0w;  // :324-326, as in easu_to_mid
          const float fpy = floorf(ppy);
          ppy -= fpy;
          mid[id] = easu_strict_pixel<FMT>(l, ((int)fpy - fy0) * fw + lxf, ppx, ppy);
        });
  }
  __syncthreads();

  // ---- phase 4: RCAS from the LDS tile.  Same streaming shape as the stand-alone pass (fsr1_rcas.hip): a wave
  //      walks down its rows with b/e/h in registers, d and f are the adjacent lanes' centre texels (DPP wave
  //      shifts), lanes 0 / 63 read the apron column.  Every lane stays active (DPP sources); stores are predicated. ----
  const int ox = ox0 + lane;
  const float sharp = as_f32(a.rcas_con[0]);
  const uint32_t flags = a.flags;
  const bool stream = (flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)ox * sizeof(out_t);
  auto rgb = [](const texel_t& p) { const float4_t c = Pixel<FMT>::load(p); return rgb_t{c.x, c.y, c.z}; };
  constexpr int kRowsPerWave = kFusedTileH / 4;
  const int ry0 = wave * kRowsPerWave;
  const texel_t* c = mid + (ry0 + 1) * kMidW + (lane + 1);  // centre texel of this lane's first row
  const int hoff = lane == 0 ? -1 : (lane == 63 ? 1 : 0);  // apron column for the wave's edge lanes
  rgb_t prev = rgb(c[-kMidW]);
  texel_t e_raw = c[0];
  rgb_t cur = rgb(e_raw);
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r, c += kMidW) {
    const int oy = oy0 + ry0 + r;
    const texel_t n_raw = c[kMidW];
    const rgb_t next = rgb(n_raw), hal = rgb(c[hoff]);
    const rgb_t d = rgb_t{dpp_f32<kDppWaveShr1>(hal.r, cur.r), dpp_f32<kDppWaveShr1>(hal.g, cur.g), dpp_f32<kDppWaveShr1>(hal.b, cur.b)};
    const rgb_t f = rgb_t{dpp_f32<kDppWaveShl1>(hal.r, cur.r), dpp_f32<kDppWaveShl1>(hal.g, cur.g), dpp_f32<kDppWaveShl1>(hal.b, cur.b)};
    rgb_t p = rcas_pixel<EXACT>(prev, d, cur, f, next, sharp, flags);
    if (ox < W && oy < H) {
      const float pa = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) ? Pixel<FMT>::load(e_raw).w : 1.0f;
      if constexpr (COLOR) color_epilogue<EXACT>(a.color, (uint32_t)ox, (uint32_t)oy, p.r, p.g, p.b);
      store_out<sizeof(out_t)>(out_col + (long long)oy * a.out.pitch, Pixel<FOUT>::store(p.r, p.g, p.b, pa), stream);
    }
    prev = cur; cur = next; e_raw = n_raw;
  }
}

size_t fused_strict_lds_bytes(int fmt, int fp_w, int fp_h);

template <int FMT, bool EXACT, bool COLOR, int FOUT, bool STRICT = false>
hipError_t fused_launch_one(const FusedArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = STRICT ? fused_strict_lds_bytes(FMT, a.fp_w, a.fp_h) : fused_lds_bytes(FMT, a.fp_w, a.fp_h);
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fused_kernel<FMT, EXACT, COLOR, FOUT, STRICT>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((fused_kernel<FMT, EXACT, COLOR, FOUT, STRICT>), grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace fsr1
