// RCAS kernel template (see fsr1_rcas.hip for the design notes); instantiated by fsr1_rcas.hip (plain pass) and
// fsr1_rcas_color.hip (colour prologue / epilogue variants).
#pragma once
#include "fsr1_device_color.hpp"
#include "fsr1_device.h"
#include "fsr1_device_rcas.hpp"

namespace fsr1 {

constexpr int kRcasWaveCols = 128;        // columns per wave (two per lane)
constexpr int kRcasWaves = 2;             // waves per workgroup, side by side
constexpr int kRcasThreads = 64 * kRcasWaves;
constexpr int kRcasCols = kRcasWaveCols * kRcasWaves;  // columns per workgroup
constexpr int kRcasRing = 8;              // rows in flight per lane (RGBA16F, batches); strips are a multiple of it tall
// fp32 texel of the lane to the left / right; `keep` stays where the neighbour lane does not exist
template <int CTRL>
__device__ __forceinline__ rgb_t neighbour(rgb_t keep, rgb_t v) {
  return rgb_t{dpp_f32<CTRL>(keep.r, v.r), dpp_f32<CTRL>(keep.g, v.g), dpp_f32<CTRL>(keep.b, v.b)};
}

template <int FMT> struct RcasPair;  // two adjacent texels, loaded / stored as one access (8-byte aligned)
template <> struct RcasPair<FSR1_FORMAT_RGBA16F> { typedef half_t T __attribute__((ext_vector_type(8), aligned(8))); };
template <> struct RcasPair<FSR1_FORMAT_RGBA32F> { typedef float T __attribute__((ext_vector_type(8), aligned(16))); };
template <> struct RcasPair<FSR1_FORMAT_RGBA8_UNORM> { typedef uint32_t T __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct RcasPair<FSR1_FORMAT_R10G10B10A2_UNORM> { typedef uint32_t T __attribute__((ext_vector_type(2), aligned(4))); };

// One 128-column x a.rows strip.  INTERIOR: every texel the strip reads (aprons included) lies inside the
// image, so nothing is predicated except the apron load of lanes 0 / 63.
// COLOR: colour stages fused in (fsr1_device_color.hpp) — FsrSrtmF on every tap as it is loaded (the role of the
// FsrRcasInputF callback, ffx_fsr1.h:682), FsrLfgaF / FsrSrtmInvF / FsrTepdC*F on the result before it is stored as FOUT.
// XEDGE (with INTERIOR): the strip's own 128 columns and every row it reads are inside the image, but it is the image's first
// and / or last wave-column: the apron column of lane 0 and / or lane 63 is outside, i.e. 0 (FSR_Pass.hlsl:45,61).  The interior
// body with that one texel replaced, instead of the fully predicated one (2 of 30 wave-columns at 3840 pixels).
// UP (with INTERIOR): the strip is walked from its last row to its first.  Strips of odd strip-rows walk up, those of even ones down, so
// that two vertically adjacent strips — resident on the same XCD at the same time (xcd_swizzle) — touch the two rows they share (each
// one's apron is the other's edge row) at the same moment: both at their start or both at their end.  The second request then finds the
// line in the XCD's L2 (or merges with the miss in flight) instead of fetching it again after 4 MB of other rows have passed through.
// The taps keep their places (b above, h below): the sums' operand order, and so every bit of the result, is that of the downward walk.
template <int FMT, bool EXACT, bool OPTS, bool INTERIOR, bool COLOR, int FOUT, int RING, bool XEDGE = false, bool UP = false>
__device__ __forceinline__ void rcas_strip(const RcasArgs& a, int frame, int x0, int y0, int lane) {
  typedef typename Pixel<FMT>::T texel_t;
  typedef typename RcasPair<FMT>::T pair_t;
  typedef typename Pixel<FOUT>::T out_t;
  typedef typename RcasPair<FOUT>::T out_pair_t;
  const uint32_t flags = OPTS ? a.flags : 0u;
  const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  const int W = a.in.width, H = a.in.height;
  const int col = x0 + 2 * lane;
  const bool ok0 = INTERIOR || col < W, ok1 = INTERIOR || col + 1 < W;
  const bool edge = lane == 0 || lane == 63;
  // apron column of lanes 0 / 63; the other lanes point at their own column so that the interior variant can load
  // unconditionally (no exec-masked branch: the compiler then counts vmcnt exactly and keeps the prefetch depth)
  const int hcol = lane == 0 ? x0 - 1 : (lane == 63 ? x0 + kRcasWaveCols : col);
  const bool halo_ok = edge && ((INTERIOR && !XEDGE) || (hcol >= 0 && hcol < W));
  const bool halo_out = XEDGE && edge && !halo_ok;  // (an outside apron texel is never addressed: its lane re-reads column 0 / W - 1)
  // a row's address = wave-uniform 64-bit row base (scalar arithmetic) + a 32-bit lane offset: no 64-bit vector arithmetic
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;
  char* const out_frame = a.out.base + (long long)frame * a.out.frame_stride;
  const uint32_t off = (uint32_t)col * (uint32_t)sizeof(texel_t), hoff = (uint32_t)min(max(hcol, 0), W - 1) * (uint32_t)sizeof(texel_t);
  const uint32_t ooff = (uint32_t)col * (uint32_t)sizeof(out_t);
  // keeps the zero-extension of a lane offset next to its use, where instruction selection can fold it into the
  // `global_load / global_store v_off, s[base]` addressing form (hoisted out of the row loop it becomes a 64-bit add per access)
  auto zext = [](uint32_t o) { asm("" : "+v"(o)); return (size_t)o; };

  const int rows = a.rows;
  static_assert(!UP || INTERIOR, "only strips that store every row walk up");
  auto row_y = [&](int k) { return UP ? y0 + rows - 1 - k : y0 + k; };  // k-th row of the walk, k = -1 .. rows
  struct row_t { texel_t p0, p1, halo; };
  auto load = [&](int y, row_t& r) {
    if (!INTERIOR) r.halo = Pixel<FMT>::zero();
    if (INTERIOR) {
      const pair_t pr = *reinterpret_cast<const pair_t*>(in_frame + (long long)y * a.in.pitch + zext(off));
      __builtin_memcpy(&r.p0, &pr, sizeof(texel_t));
      __builtin_memcpy(&r.p1, reinterpret_cast<const char*>(&pr) + sizeof(texel_t), sizeof(texel_t));
      r.halo = *reinterpret_cast<const texel_t*>(in_frame + (long long)y * a.in.pitch + zext(hoff));
      if constexpr (XEDGE) {
        if (halo_out) r.halo = Pixel<FMT>::zero();
      }
    } else {
      r.p0 = Pixel<FMT>::zero();
      r.p1 = Pixel<FMT>::zero();
      if (y >= -a.rows_above && y < H + a.rows_below) {  // wave-uniform; a band's neighbouring rows are real texels
        if (ok1) {
          const pair_t pr = *reinterpret_cast<const pair_t*>(in_frame + (long long)y * a.in.pitch + zext(off));
          __builtin_memcpy(&r.p0, &pr, sizeof(texel_t));
          __builtin_memcpy(&r.p1, reinterpret_cast<const char*>(&pr) + sizeof(texel_t), sizeof(texel_t));
        } else if (ok0) {
          r.p0 = *reinterpret_cast<const texel_t*>(in_frame + (long long)y * a.in.pitch + zext(off));
        }
        if (halo_ok) r.halo = *reinterpret_cast<const texel_t*>(in_frame + (long long)y * a.in.pitch + zext(hoff));
      }
    }
  };
  auto rgb = [&](const texel_t& p) {
    float4_t c = Pixel<FMT>::load(p);
    if constexpr (COLOR) c = color_prologue<EXACT>(a.color, c);
    return rgb_t{c.x, c.y, c.z};
  };
  // Ring of raw rows in registers: slot k holds row y0 + r with r % kRing == k; loads run kAhead = kRing - 1 rows
  // ahead of the arithmetic.  The row loop is unrolled by kRing, so every ring index is static, and rolled
  // beyond that so the body stays inside the instruction cache.
  // (the colour variants inline two epilogues per row: a shorter unroll keeps them inside the instruction cache)
  constexpr int kRing = RING, kAhead = kRing - 1;  // divides every strip height the launcher picks (multiples of 8)
  row_t q[kRing];
  rgb_t prev0, prev1, cur0, cur1;
  {
    row_t q_prev;
    load(row_y(-1), q_prev);
#pragma unroll
    for (int k = 0; k < kAhead; ++k) load(row_y(k), q[k]);
    prev0 = rgb(q_prev.p0); prev1 = rgb(q_prev.p1);
    cur0 = rgb(q[0].p0); cur1 = rgb(q[0].p1);
  }
  const float sharp = as_f32(a.con[0]);

#pragma unroll 1
  for (int r0 = 0; r0 < rows; r0 += kRing) {
#pragma unroll
    for (int k = 0; k < kRing; ++k) {
      const int y = row_y(r0 + k);
      load(row_y(min(r0 + k + kAhead, rows)), q[(k + kAhead) % kRing]);  // past the end: re-reads the last row (harmless, branch-free)
      const row_t& c = q[k];
      const row_t& n = q[(k + 1) % kRing];
      const rgb_t next0 = rgb(n.p0), next1 = rgb(n.p1);
      const rgb_t hal = rgb(c.halo);
      // horizontal neighbours: pixel 0's left = the left lane's pixel 1, pixel 1's right = the right lane's pixel 0;
      // lanes 0 / 63 keep their apron texel
      const rgb_t d0 = neighbour<kDppWaveShr1>(hal, cur1), f1 = neighbour<kDppWaveShl1>(hal, cur0);
      // b is the tap above, h the one below (ffx_fsr1.h:697-707), whichever way the strip is walked
      rgb_t o0 = rcas_pixel<EXACT>(UP ? next0 : prev0, d0, cur0, cur1, UP ? prev0 : next0, sharp, flags);
      rgb_t o1 = rcas_pixel<EXACT>(UP ? next1 : prev1, cur0, cur1, f1, UP ? prev1 : next1, sharp, flags);
      if (INTERIOR || y < H) {
        const bool alpha = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) != 0;  // :700-705 / FSR_Pass.hlsl:94
        if constexpr (COLOR) {
          color_epilogue<EXACT>(a.color, (uint32_t)col, (uint32_t)y, o0.r, o0.g, o0.b);
          color_epilogue<EXACT>(a.color, (uint32_t)col + 1u, (uint32_t)y, o1.r, o1.g, o1.b);
        }
        const out_t t0 = Pixel<FOUT>::store(o0.r, o0.g, o0.b, alpha ? Pixel<FMT>::load(c.p0).w : 1.0f);
        const out_t t1 = Pixel<FOUT>::store(o1.r, o1.g, o1.b, alpha ? Pixel<FMT>::load(c.p1).w : 1.0f);
        char* const dst = out_frame + (long long)y * a.out.pitch + zext(ooff);
        if (ok1) {
          out_pair_t pr;
          __builtin_memcpy(&pr, &t0, sizeof(out_t));
          __builtin_memcpy(reinterpret_cast<char*>(&pr) + sizeof(out_t), &t1, sizeof(out_t));
          store_out<sizeof(out_t)>(dst, pr, stream);
        } else if (ok0) {
          store_out<sizeof(out_t)>(dst, t0, stream);
        }
      }
      prev0 = cur0; prev1 = cur1; cur0 = next0; cur1 = next1;
    }
  }
}

// OPTS = false: the plain pass (no denoise / alpha pass-through / HDR square), flags compiled out.
// RING: rows in flight per lane.  A deep ring (8: loads 7 rows ahead) hides HBM latency when the launch has waves to
// spare — batches — but costs 102 VGPRs, 4 waves per SIMD; a single 4K frame is only 4 waves per SIMD to begin with, and
// there the pass is paced by its own arithmetic (dependent min/max/rcp chains), which a 64-VGPR body with a 2-row ring
// and twice the resident waves serves better (cold 4K frame: 37.6 -> 30.8 us).  rcas_launch picks by wave count.
template <int FMT, bool EXACT, bool OPTS, bool COLOR = false, int FOUT = FMT, int RING = ((FMT == FSR1_FORMAT_RGBA32F || COLOR) ? 4 : kRcasRing)>
__global__ void __launch_bounds__(kRcasThreads) rcas_kernel(const RcasArgs a) {
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = tx * kRcasCols + wave * kRcasWaveCols, y0 = ty * a.rows;
  if (x0 >= a.in.width) return;  // whole wave outside (no barriers in this kernel)
  const bool interior_y = y0 >= 1 - a.rows_above && y0 + a.rows + 1 <= a.in.height + a.rows_below &&
                          y0 + a.rows <= a.in.height;  // (the interior body stores every row of the strip)
  const bool interior = interior_y && x0 >= 1 && x0 + kRcasWaveCols + 1 <= a.in.width;
  const bool up = (ty & 1) != 0;
  if (interior && up) rcas_strip<FMT, EXACT, OPTS, true, COLOR, FOUT, RING, false, true>(a, frame, x0, y0, lane);
  else if (interior) rcas_strip<FMT, EXACT, OPTS, true, COLOR, FOUT, RING>(a, frame, x0, y0, lane);
  else if (interior_y && x0 + kRcasWaveCols <= a.in.width && up) rcas_strip<FMT, EXACT, OPTS, true, COLOR, FOUT, RING, true, true>(a, frame, x0, y0, lane);
  else if (interior_y && x0 + kRcasWaveCols <= a.in.width) rcas_strip<FMT, EXACT, OPTS, true, COLOR, FOUT, RING, true>(a, frame, x0, y0, lane);
  else rcas_strip<FMT, EXACT, OPTS, false, COLOR, FOUT, RING>(a, frame, x0, y0, lane);
}

}  // namespace fsr1
