// RCAS — robust contrast adaptive sharpening (FsrRcasF, ffx-fsr/ffx_fsr1.h:684-769) for gfx950.
//
// 8 B read + 8 B written per pixel (RGBA16F) against ~100 VALU instructions: on MI355X the pass sits
// right at the HBM/VALU balance point, so the kernel is built to touch every input byte once, to move
// 16 bytes per lane per memory instruction, and to spend no instruction on staging:
//   * no LDS, no barrier: a wave owns a 128-column x 16-row strip and streams down it, a lane owning TWO
//     adjacent columns (one 16-byte load and one 16-byte store per row);
//   * vertical neighbours (b above, h below) are the lane's own previous/next rows kept in registers;
//     of the horizontal neighbours, two are the lane's own other pixel and two are the adjacent lanes'
//     facing pixels, fetched in fp32 with DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1);
//   * lanes 0 and 63 additionally load the one texel left / right of the strip, which the DPP move
//     leaves in place for exactly those lanes (an invalid DPP source keeps the old destination);
//   * every texel is converted to fp32 once; loads run kAhead rows ahead of the arithmetic;
//   * strips that lie wholly inside the image (all but the image's border strips) run a branch-free body:
//     one basic block per strip, no per-lane predicates, no zero fills;
//   * texels outside the image are 0 (the D3D `Load` rule of the reference's callback, FSR_Pass.hlsl:45,61).
#include "fsr1_device.h"
#include "fsr1_rcas_math.h"

namespace fsr1 {

#ifndef FSR1_RCAS_WAVES
#define FSR1_RCAS_WAVES 2
#endif
#ifndef FSR1_RCAS_RING
#define FSR1_RCAS_RING 8
#endif
constexpr int kRcasWaveCols = 128;        // columns per wave (two per lane)
constexpr int kRcasWaves = FSR1_RCAS_WAVES;          // waves per workgroup, side by side
constexpr int kRcasThreads = 64 * kRcasWaves;
constexpr int kRcasCols = kRcasWaveCols * kRcasWaves;  // columns per workgroup
constexpr int kRcasRing = FSR1_RCAS_RING;            // rows in flight per lane (RGBA16F); strips are a multiple of it tall
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
template <int FMT, bool EXACT, bool OPTS, bool INTERIOR>
__device__ __forceinline__ void rcas_strip(const RcasArgs& a, int frame, int x0, int y0, int lane) {
  typedef typename Pixel<FMT>::T texel_t;
  typedef typename RcasPair<FMT>::T pair_t;
  const uint32_t flags = OPTS ? a.flags : 0u;
  const int W = a.in.width, H = a.in.height;
  const int col = x0 + 2 * lane;
  const bool ok0 = INTERIOR || col < W, ok1 = INTERIOR || col + 1 < W;
  const bool edge = lane == 0 || lane == 63;
  // apron column of lanes 0 / 63; the other lanes point at their own column so that the interior variant can load
  // unconditionally (no exec-masked branch: the compiler then counts vmcnt exactly and keeps the prefetch depth)
  const int hcol = lane == 0 ? x0 - 1 : (lane == 63 ? x0 + kRcasWaveCols : col);
  const bool halo_ok = edge && (INTERIOR || (hcol >= 0 && hcol < W));
  const char* const in_col = a.in.base + (long long)frame * a.in.frame_stride + (size_t)col * sizeof(texel_t);
  const char* const in_hcol = a.in.base + (long long)frame * a.in.frame_stride + (size_t)hcol * sizeof(texel_t);
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)col * sizeof(texel_t);

  struct row_t { texel_t p0, p1, halo; };
  auto load = [&](int y, row_t& r) {
    if (!INTERIOR) r.halo = Pixel<FMT>::zero();
    if (INTERIOR) {
      const pair_t pr = *reinterpret_cast<const pair_t*>(in_col + (long long)y * a.in.pitch);
      __builtin_memcpy(&r.p0, &pr, sizeof(texel_t));
      __builtin_memcpy(&r.p1, reinterpret_cast<const char*>(&pr) + sizeof(texel_t), sizeof(texel_t));
      r.halo = *reinterpret_cast<const texel_t*>(in_hcol + (long long)y * a.in.pitch);
    } else {
      r.p0 = Pixel<FMT>::zero();
      r.p1 = Pixel<FMT>::zero();
      if (y >= 0 && y < H) {  // wave-uniform
        if (ok1) {
          const pair_t pr = *reinterpret_cast<const pair_t*>(in_col + (long long)y * a.in.pitch);
          __builtin_memcpy(&r.p0, &pr, sizeof(texel_t));
          __builtin_memcpy(&r.p1, reinterpret_cast<const char*>(&pr) + sizeof(texel_t), sizeof(texel_t));
        } else if (ok0) {
          r.p0 = *reinterpret_cast<const texel_t*>(in_col + (long long)y * a.in.pitch);
        }
        if (halo_ok) r.halo = *reinterpret_cast<const texel_t*>(in_hcol + (long long)y * a.in.pitch);
      }
    }
  };
  auto rgb = [](const texel_t& p) { const float4_t c = Pixel<FMT>::load(p); return rgb_t{c.x, c.y, c.z}; };
  // Loop-carried fp32 values reach v_min/v_max through block boundaries, where the compiler no longer knows
  // they are canonical and would spend a v_max_f32 x,x,x (4.3 cycles) on each; x+0.0 (2.4 cycles) tells it the
  // same thing.  It turns -0 into +0, so the EXACT variant does not use it.
  auto known = [](rgb_t v) { return EXACT ? v : rgb_t{v.r + 0.0f, v.g + 0.0f, v.b + 0.0f}; };

  // Ring of raw rows in registers: slot k holds row y0 + r with r % kRing == k; loads run kAhead = kRing - 1 rows
  // ahead of the arithmetic.  The row loop is unrolled by kRing, so every ring index is static, and rolled
  // beyond that so the body stays inside the instruction cache.
  constexpr int kRing = FMT == FSR1_FORMAT_RGBA32F ? 4 : kRcasRing, kAhead = kRing - 1;  // both divide 8
  row_t q[kRing];
  rgb_t prev0, prev1, cur0, cur1;
  {
    row_t q_prev;
    load(y0 - 1, q_prev);
#pragma unroll
    for (int k = 0; k < kAhead; ++k) load(y0 + k, q[k]);
    prev0 = rgb(q_prev.p0); prev1 = rgb(q_prev.p1);
    cur0 = rgb(q[0].p0); cur1 = rgb(q[0].p1);
  }
  const float sharp = as_f32(a.con[0]);
  const int rows = a.rows;
  const int y_last = y0 + rows;  // the row below the strip is the last one read

#pragma unroll 1
  for (int r0 = 0; r0 < rows; r0 += kRing) {
#pragma unroll
    for (int k = 0; k < kRing; ++k) {
      const int y = y0 + r0 + k;
      load(min(y + kAhead, y_last), q[(k + kAhead) % kRing]);  // past the end: re-reads the last row (harmless, branch-free)
      const row_t& c = q[k];
      const row_t& n = q[(k + 1) % kRing];
      const rgb_t next0 = rgb(n.p0), next1 = rgb(n.p1);
      const rgb_t hal = rgb(c.halo);
      // horizontal neighbours: pixel 0's left = the left lane's pixel 1, pixel 1's right = the right lane's pixel 0;
      // lanes 0 / 63 keep their apron texel
      const rgb_t d0 = neighbour<kDppWaveShr1>(hal, cur1), f1 = neighbour<kDppWaveShl1>(hal, cur0);
#ifdef FSR1_RCAS_COPY_ONLY  // tuning experiment: memory pattern without the arithmetic
      const rgb_t o0 = {prev0.r + d0.r, cur0.g + next0.g, cur1.b}, o1 = {prev1.r + f1.r, cur1.g + next1.g, cur0.b};
      (void)sharp;
#else
      const rgb_t o0 = rcas_pixel<EXACT>(known(prev0), d0, known(cur0), known(cur1), next0, sharp, flags);
      const rgb_t o1 = rcas_pixel<EXACT>(known(prev1), known(cur0), known(cur1), f1, next1, sharp, flags);
#endif
      if (INTERIOR || y < H) {
        const bool alpha = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) != 0;  // :700-705 / FSR_Pass.hlsl:94
        const texel_t t0 = Pixel<FMT>::store(o0.r, o0.g, o0.b, alpha ? Pixel<FMT>::load(c.p0).w : 1.0f);
        const texel_t t1 = Pixel<FMT>::store(o1.r, o1.g, o1.b, alpha ? Pixel<FMT>::load(c.p1).w : 1.0f);
        char* const dst = out_col + (long long)y * a.out.pitch;
        if (ok1) {
          pair_t pr;
          __builtin_memcpy(&pr, &t0, sizeof(texel_t));
          __builtin_memcpy(reinterpret_cast<char*>(&pr) + sizeof(texel_t), &t1, sizeof(texel_t));
          *reinterpret_cast<pair_t*>(dst) = pr;
        } else if (ok0) {
          *reinterpret_cast<texel_t*>(dst) = t0;
        }
      }
      prev0 = cur0; prev1 = cur1; cur0 = next0; cur1 = next1;
    }
  }
}

// OPTS = false: the plain pass (no denoise / alpha pass-through / HDR square), flags compiled out.
template <int FMT, bool EXACT, bool OPTS>
__global__ void __launch_bounds__(kRcasThreads) rcas_kernel(const RcasArgs a) {
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = tx * kRcasCols + wave * kRcasWaveCols, y0 = ty * a.rows;
  if (x0 >= a.in.width) return;  // whole wave outside (no barriers in this kernel)
  const bool interior = x0 >= 1 && x0 + kRcasWaveCols + 1 <= a.in.width && y0 >= 1 && y0 + a.rows + 1 <= a.in.height;
  if (interior) rcas_strip<FMT, EXACT, OPTS, true>(a, frame, x0, y0, lane);
  else rcas_strip<FMT, EXACT, OPTS, false>(a, frame, x0, y0, lane);
}

// Strip height.  Measured on MI355X at 3840x2160 (gpurun_out/): the pass runs at the same ~34 us for
// 8..16-row strips and slows down beyond (24 rows 39 us, 32 rows 42-46 us, 64 rows 67 us) even when the
// workgroups divide evenly over the CUs and however deep the per-lane prefetch ring is: what counts is the number
// of independent row streams in flight, and the apron rows that shorter strips re-read are cheap next to it.
// 16 rows (two apron rows per 16 = 12.5 % extra reads, mostly L2/MALL hits) unless that leaves fewer than four
// waves per SIMD, then 8.
void rcas_geometry(int width, int height, int frames, int* tiles_x, int* tiles_y, int* rows) {
  const int tx = (width + kRcasCols - 1) / kRcasCols;
#ifdef FSR1_RCAS_ROWS
  int r = FSR1_RCAS_ROWS;
#else
  int r = 16;
  if ((long long)tx * ((height + r - 1) / r) * frames * kRcasWaves < 4LL * 4 * 256) r = 8;
#endif
  *rows = r;
  *tiles_x = tx;
  *tiles_y = (height + r - 1) / r;
}

hipError_t rcas_launch(const RcasArgs& a, int fmt, bool exact, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kRcasThreads);
  const bool opts = (a.flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA | FSR1_FLAG_HDR_SQUARE)) != 0;
#define FSR1_RCAS(F, E, O) hipLaunchKernelGGL((rcas_kernel<F, E, O>), grid, block, 0, stream, a)
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
