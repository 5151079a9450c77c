// EASU -> RCAS in one launch with the packed-binary16 entry points (FsrEasuH, ffx_fsr1.h:505-593, then FsrRcasH / FsrRcasHx2,
// :782-984), exact-2x variant: the H twin of fsr1_fused_s2.hip (round 4; FSR_Pass.hlsl:81-87 makes FsrEasuH -> FsrRcasH the
// reference's shipping default, and BASELINE configs[1] / [3] are exactly 2x).
//
// Same geometry as the F kernel: a workgroup owns a 62-pixel column of the output and walks down it in steps of 16 EASU rows;
// a lane owns one 2 x 2 quad of the step's 64 x 16 EASU pixels (compile-time 35 x 11 footprint, sub-texel positions 1/4 and 3/4),
// the rows go to an LDS ring of 18 rows as the RGBA16F texels the two H dispatches would have stored, and the RCAS phase has lane L
// own apron column L, so both horizontal neighbours arrive by DPP.  The RCAS arithmetic is the two-pixel form (FsrRcasHx2's, lane by
// lane FsrRcasH's) with two VERTICALLY adjacent pixels in the halves of every operand.  Every arithmetic operation is the one
// native binary16 operation the two H kernels run, on the same values: bit-identical to fsr1_easu_dispatch + fsr1_rcas_dispatch
// with FSR1_FLAG_MATH_PACKED_FP16 (tests/test_gpu_parity_h.py, whole frames in tests/test_gpu_fullframe.py).
// LDS: 385 texels x 48 B (easu_h_stage's records) + 18 x 64 x 8 B = 27.7 KB: five workgroups per CU.
#include "fsr1_device.h"
#include "fsr1_device_half.hpp"

namespace fsr1 {

namespace {
constexpr int kShr1 = 0x138, kShl1 = 0x130;  // DPP wave shifts: lane i <- lane i - 1 / i + 1
template <int CTRL>
__device__ __forceinline__ half2_t dpp2(half2_t keep, half2_t v) {
  return __builtin_bit_cast(half2_t, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
}  // namespace

size_t fused_s2_h_lds_bytes() { return (size_t)kFs2FpW * kFs2FpH * kEasuHLdsPerTexel + (size_t)kFs2MidW * kFs2Ring * sizeof(half4_t); }

// OPTS = false: plain pass (no denoise / alpha pass-through / HDR square), flags compiled out.  RUN = false: the one-step launch.
template <bool OPTS, bool RUN>
__global__ void __launch_bounds__(kThreads) fused_s2_h_kernel(const FusedArgs a) {
  static_assert(kFs2QH == 8 && kThreads == 256, "32 x 8 quads over 256 lanes");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const EasuHLds l = easu_h_lds_carve(smem, kFs2FpW * kFs2FpH);
  half4_t* const mid = reinterpret_cast<half4_t*>(smem + (size_t)kFs2FpW * kFs2FpH * kEasuHLdsPerTexel);  // [kFs2Ring][64], a ring of rows

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int W = a.out.width, H = a.out.height;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int steps = RUN ? a.run_steps : 1;
  const int Y0 = (kFs2Step * steps - 2) * ty;  // the run's first output row (even)
  const int ax0 = kFs2OutW * tx - 1;           // odd, like every step's first EASU row
  const uint32_t flags = OPTS ? a.flags : 0u;
  const bool stream = (a.flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  const half_t sharp1 = __builtin_bit_cast(half_t, (u16)(a.rcas_con[1] & 0xffffu));  // :857 AH2_AU1(con.y).x
  char* const out_frame = a.out.base + (long long)frame * a.out.frame_stride;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;
  typedef TexelPair<FSR1_FORMAT_RGBA16F> pair;
  const half4_t zero4 = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
  const half_t q1 = (half_t)0.25f, q3 = (half_t)0.75f, one = (half_t)1.0f;

  // Ring row j of step s (j = 0, 1: the last two EASU rows of step s - 1; j = 2 .. 17: this step's) is EASU row ay0 + j - 2 and
  // sits in ring slot (base + j) mod 18, base = 16 s mod 18.
  int base = 0;
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int ay0 = Y0 - 1 + kFs2Step * s;  // first EASU row of this step
    const int tid = threadIdx.x, lane = tid & 63;
    easu_h_stage<kFs2FpW, kFs2FpH>(l, a.in, in_frame, 31 * tx - 2, ((ay0 + 1) >> 1) - 2, kFs2FpW, kFs2FpH, tid);
    // (its two barriers also separate this step's ring writes from the previous step's RCAS reads)

    // ---- phase 3: FsrEasuH on the step's 64 x 16 pixels, a quad per lane (EASU runs with Sample.x = 0 when RCAS follows:
    //      FSR_Filter.cpp:107); pixels outside the image are 0 (FSR_Pass.hlsl:45,61) ----
    {
      const int qx = lane & 31, qy = 2 * wave + (lane >> 5);
      const int oxa = ax0 + 2 * qx, oya = ay0 + 2 * qy;
      const bool xin0 = oxa >= 0 && oxa < W, xin1 = oxa + 1 < W, yin0 = oya >= 0 && oya < H, yin1 = oya + 1 < H;
      const int f = (qy + 1) * kFs2FpW + (qx + 1);
      int slot = base + 2 + 2 * qy;  // even, so the quad's two rows never straddle the wrap
      slot -= slot >= kFs2Ring ? kFs2Ring : 0;
      half4_t* const m0 = mid + slot * kFs2MidW + 2 * qx;
      half4_t p00 = zero4, p10 = zero4, p01 = zero4, p11 = zero4;
      if (xin0 && xin1 && yin0 && yin1) {  // every quad but those on the image's border
        p00 = easu_h_pixel(l, f, kFs2FpW, h2(q1, q1), false);
        p10 = easu_h_pixel(l, f, kFs2FpW, h2(q3, q1), false);
        p01 = easu_h_pixel(l, f, kFs2FpW, h2(q1, q3), false);
        p11 = easu_h_pixel(l, f, kFs2FpW, h2(q3, q3), false);
      } else if ((xin0 || xin1) && (yin0 || yin1)) {
        if (xin0 && yin0) p00 = easu_h_pixel(l, f, kFs2FpW, h2(q1, q1), false);
        if (xin1 && yin0) p10 = easu_h_pixel(l, f, kFs2FpW, h2(q3, q1), false);
        if (xin0 && yin1) p01 = easu_h_pixel(l, f, kFs2FpW, h2(q1, q3), false);
        if (xin1 && yin1) p11 = easu_h_pixel(l, f, kFs2FpW, h2(q3, q3), false);
      }
      *reinterpret_cast<pair::T*>(m0) = pair::make(p00, p10);
      *reinterpret_cast<pair::T*>(m0 + kFs2MidW) = pair::make(p01, p11);
    }
    __syncthreads();

    // ---- phase 4: FsrRcasH from the ring.  Lane L owns apron column L; wave w takes the ring rows 1 + 4 w .. 4 + 4 w as centres,
    //      two vertically adjacent rows per evaluation (step 0 has no rows above its row 2: its wave 0 leaves the first pair out);
    //      d and f are the neighbouring lanes' centre pairs (DPP wave shifts).  Every lane stays active (DPP sources); lanes 0 / 63
    //      and pixels outside the image store nothing. ----
    {
      const int ox = ax0 + lane;
      const bool col_ok = lane >= 1 && lane <= kFs2OutW && ox < W;
      const uint32_t col_off = (uint32_t)ox * 8u;  // (never used when ox < 0: lane 0 stores nothing)
      auto row = [&](int j) {  // (wave-uniform slot)
        int slot = base + j;
        slot -= slot >= kFs2Ring ? kFs2Ring : 0;
        return mid[slot * kFs2MidW + lane];
      };
      auto do_pair = [&](int j) {  // centres: ring rows j, j + 1
        const half4_t up = row(j - 1), e0 = row(j), e1 = row(j + 1), dn = row(j + 2);
        const half2_t eR = h2(e0.x, e1.x), eG = h2(e0.y, e1.y), eB = h2(e0.z, e1.z);
        const half2_t dR = dpp2<kShr1>(eR, eR), dG = dpp2<kShr1>(eG, eG), dB = dpp2<kShr1>(eB, eB);
        const half2_t fR = dpp2<kShl1>(eR, eR), fG = dpp2<kShl1>(eG, eG), fB = dpp2<kShl1>(eB, eB);
        const rgbh2_t px = rcas_pixel_h2(h2(up.x, e0.x), h2(up.y, e0.y), h2(up.z, e0.z),  // b: above
                                         dR, dG, dB, eR, eG, eB, fR, fG, fB,
                                         h2(e1.x, dn.x), h2(e1.y, dn.y), h2(e1.z, dn.z),  // h: below
                                         sharp1, flags);
        const bool alpha = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) != 0;  // :905-907 / FSR_Pass.hlsl:94
        const int oy = ay0 + j - 2;
        if (col_ok && oy < H)
          store_out<8>(out_frame + (long long)oy * a.out.pitch + col_off, half4_t{px.r.x, px.g.x, px.b.x, alpha ? e0.w : one}, stream);
        if (col_ok && oy + 1 < H)
          store_out<8>(out_frame + (long long)(oy + 1) * a.out.pitch + col_off, half4_t{px.r.y, px.g.y, px.b.y, alpha ? e1.w : one}, stream);
      };
      const int j0 = 1 + 4 * wave;
      if (!(s == 0 && wave == 0)) do_pair(j0);  // wave-uniform
      do_pair(j0 + 2);
    }
    if (!RUN || s + 1 == steps || ay0 + kFs2Step - 1 >= H) break;  // (the next step's first output row is ay0 + 15)
    base += kFs2Step;
    base -= base >= kFs2Ring ? kFs2Ring : 0;
  }
}

hipError_t fused_s2_h_launch(const FusedArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = fused_s2_h_lds_bytes();
  const bool opts = (a.flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA | FSR1_FLAG_HDR_SQUARE)) != 0;
  if (a.run_steps > 1) {
    if (opts) hipLaunchKernelGGL((fused_s2_h_kernel<true, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((fused_s2_h_kernel<false, true>), grid, block, lds, stream, a);
  } else {
    if (opts) hipLaunchKernelGGL((fused_s2_h_kernel<true, false>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((fused_s2_h_kernel<false, false>), grid, block, lds, stream, a);
  }
  return hipGetLastError();
}

}  // namespace fsr1
