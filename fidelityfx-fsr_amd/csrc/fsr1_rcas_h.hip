// RCAS, packed binary16 arithmetic — FsrRcasHx2 (ffx-fsr/ffx_fsr1.h:888-984), whose per-element
// arithmetic is FsrRcasH (:782-866), for gfx950.
//
// Parity class "H": each lane sharpens TWO horizontally adjacent pixels in the two halves of packed
// binary16 registers (the reference pairs pixels 8 columns apart, :913; the pairing does not enter the
// arithmetic), one native binary16 operation per reference operation, contraction off — bit-identical
// to the reference's FsrRcasH evaluated on the CPU with round-to-nearest-even after every operation.
//
// Streaming structure of the fp32 kernel (fsr1_rcas.hip): no LDS, a wave owns a 128-column x 8-row strip,
// rows flow through registers (b = previous row, h = next row), the horizontal neighbours of a lane's
// pixel pair are its own other pixel and the adjacent lane's facing pixel (DPP wave shift); lanes 0 / 63
// also load the strip's apron column.  Loads outside the image are 0 (FSR_Pass.hlsl:61).
#include "fsr1_device.h"
#include "fsr1_device_half.hpp"

namespace fsr1 {

constexpr int kRcasHRows = 8;   // rows per strip, a multiple of the 4-row ring.  Measured at 4K, us (profiles/ab_r02/r2c9_ab.log): 8 rows 26.7, 16 rows 27.1, 32 rows 31.7; round 1's 24 rows x 4 waves, fully unrolled: 31.0
constexpr int kRcasHWaves = 2; // waves per workgroup, side by side: the F kernel's shape (fsr1_rcas_kernel.h)
constexpr int kRcasHThreads = 64 * kRcasHWaves;
constexpr int kRcasHCols = 128 * kRcasHWaves;  // columns per workgroup
constexpr int kShr1 = 0x138, kShl1 = 0x130;

namespace {

template <int CTRL>
__device__ __forceinline__ half2_t dpp2(half2_t keep, half2_t v) {
  return __builtin_bit_cast(half2_t, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

}  // namespace

// OPTS = false: plain pass, flags compiled out.
template <bool OPTS>
__global__ void __launch_bounds__(kRcasHThreads) rcas_h_kernel(const RcasArgs a) {
  const uint32_t flags = OPTS ? a.flags : 0u;
  const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = tx * kRcasHCols + wave * 128, y0 = ty * kRcasHRows;
  if (x0 >= a.in.width) return;

  const int W = a.in.width, H = a.in.height;
  const int col = x0 + 2 * lane;
  const bool ok0 = col < W, ok1 = col + 1 < W;
  const int hcol = lane == 0 ? x0 - 1 : x0 + 128;
  const bool halo_ok = (lane == 0 || lane == 63) && hcol >= 0 && hcol < W;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;
  char* const out_frame = a.out.base + (long long)frame * a.out.frame_stride;

  // a row's address = wave-uniform 64-bit row base + a 32-bit lane offset (see rcas_strip in fsr1_rcas_kernel.h)
  const uint32_t off = (uint32_t)col * 8u, hoff = (uint32_t)max(hcol, 0) * 8u;
  auto zext = [](uint32_t o) { asm("" : "+v"(o)); return (size_t)o; };
  typedef TexelPair<FSR1_FORMAT_RGBA16F>::T pair_t;
  // one row of this lane's pixel pair as structure-of-arrays, plus the apron texel broadcast to both halves; the pair is
  // ONE 16-byte access (8-byte accesses run at 0.5-0.7 of the 16-byte rate on this chip)
  auto load = [&](int y, soa_t& own, soa_t& halo) {
    half4_t p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0}, ph = {0, 0, 0, 0};
    if (y >= 0 && y < H) {  // wave-uniform
      const char* const row = in_frame + (long long)y * a.in.pitch;
      if (ok1) {
        const pair_t pr = *reinterpret_cast<const pair_t*>(row + zext(off));
        p0 = half4_t{pr[0], pr[1], pr[2], pr[3]};
        p1 = half4_t{pr[4], pr[5], pr[6], pr[7]};
      } else if (ok0) {
        p0 = *reinterpret_cast<const half4_t*>(row + zext(off));
      }
      if (halo_ok) ph = *reinterpret_cast<const half4_t*>(row + zext(hoff));
    }
    own = soa_t{half2_t{p0.x, p1.x}, half2_t{p0.y, p1.y}, half2_t{p0.z, p1.z}, half2_t{p0.w, p1.w}};
    halo = soa_t{half2_t{ph.x, ph.x}, half2_t{ph.y, ph.y}, half2_t{ph.z, ph.z}, half2_t{ph.w, ph.w}};
  };

  constexpr int kAhead = 3, kRing = kAhead + 1;
  soa_t q[kRing], g[kRing];
  soa_t prev;
  {
    soa_t g_prev;
    load(y0 - 1, prev, g_prev);
#pragma unroll
    for (int k = 0; k < kAhead; ++k) load(y0 + k, q[k], g[k]);
  }
  // :857 sharpness = AH2_AU1(con.y).x — the packed half of con[1]
  const half_t sharp1 = __builtin_bit_cast(half_t, (u16)(a.con[1] & 0xffffu));
  const half2_t one = s2(1.0f);

  static_assert(kRcasHRows % kRing == 0, "strips are a multiple of the ring tall");
  // the row loop is unrolled by the ring (static ring indices) and rolled beyond that, so the body stays inside the
  // instruction cache (24 fully unrolled rows were 44 KB of code)
#pragma unroll 1
  for (int r0 = 0; r0 < kRcasHRows; r0 += kRing)
#pragma unroll
  for (int k = 0; k < kRing; ++k) {
    const int r = r0 + k;
    const int y = y0 + r;
    // (past the row below the strip nothing is needed: re-read that row — a cache hit — instead of fetching two more rows per
    //  strip from memory, which round 3's kernel did: 99.6 MB fetched per 4K frame against the F kernel's 83.1 MB)
    load(min(y + kAhead, y0 + kRcasHRows), q[(k + kAhead) % kRing], g[(k + kAhead) % kRing]);
    const soa_t& e = q[k % kRing];
    const soa_t& eh = g[k % kRing];
    const soa_t& h = q[(k + 1) % kRing];
    const soa_t& b = prev;
    // d = (left neighbour's right pixel, own left pixel) ; f = (own right pixel, right neighbour's left pixel)
    const half2_t nlR = dpp2<kShr1>(eh.r, e.r), nlG = dpp2<kShr1>(eh.g, e.g), nlB = dpp2<kShr1>(eh.b, e.b);
    const half2_t nrR = dpp2<kShl1>(eh.r, e.r), nrG = dpp2<kShl1>(eh.g, e.g), nrB = dpp2<kShl1>(eh.b, e.b);
    const half2_t dR = {nlR.y, e.r.x}, dG = {nlG.y, e.g.x}, dB = {nlB.y, e.b.x};
    const half2_t fR = {e.r.y, nrR.x}, fG = {e.g.y, nrG.x}, fB = {e.b.y, nrB.x};
    const half2_t bR = b.r, bG = b.g, bB = b.b, eR = e.r, eG = e.g, eB = e.b, hR = h.r, hG = h.g, hB = h.b;

    const rgbh2_t px = rcas_pixel_h2(bR, bG, bB, dR, dG, dB, eR, eG, eB, fR, fG, fB, hR, hG, hB, sharp1, flags);
    const half2_t pR = px.r, pG = px.g, pB = px.b;
    const half2_t pA = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) ? e.a : one;       // :905-907 / FSR_Pass.hlsl:94
    if (y < H) {
      char* const row = out_frame + (long long)y * a.out.pitch + zext(off);
      // FsrRcasDepackHx2 :880-886
      const half4_t t0 = {pR.x, pG.x, pB.x, pA.x}, t1 = {pR.y, pG.y, pB.y, pA.y};
      if (ok1) store_out<8>(row, TexelPair<FSR1_FORMAT_RGBA16F>::make(t0, t1), stream);
      else if (ok0) store_out<8>(row, t0, stream);
    }
    prev = e;
  }
}

void rcas_h_geometry(int width, int height, int* tiles_x, int* tiles_y) {
  *tiles_x = (width + kRcasHCols - 1) / kRcasHCols;
  *tiles_y = (height + kRcasHRows - 1) / kRcasHRows;
}

hipError_t rcas_h_launch(const RcasArgs& a0, hipStream_t stream) {
  RcasArgs a = a0;
  rcas_h_geometry(a.in.width, a.in.height, &a.tiles_x, &a.tiles_y);  // its own strip shape (two pixels per lane)
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kRcasHThreads);
  const bool opts = (a.flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA | FSR1_FLAG_HDR_SQUARE)) != 0;
  if (opts) hipLaunchKernelGGL(rcas_h_kernel<true>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(rcas_h_kernel<false>, grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace fsr1
