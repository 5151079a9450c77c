// EASU, packed binary16 arithmetic — FsrEasuH (ffx-fsr/ffx_fsr1.h:445-593) for gfx950.
//
// Parity class "H": every arithmetic operation of the reference's half-precision path is one native
// binary16 operation here (v_pk_mul_f16 / v_pk_add_f16 / v_pk_max_f16 ..., contraction off), in the
// reference's operation order and with its two-taps-per-packed-op pairing, so results are
// bit-identical to the reference's FsrEasuH evaluated on the CPU with round-to-nearest-even after
// every operation (oracle/: ref_easu_h, oracle_easu_h).  It is *not* within 1 ULP of FsrEasuF — the
// reference's own H path is not (different magic constants 0x7784/0x59a3, true rcp in FsrEasuSetH).
//
// Same work decomposition as the fp32 kernel (fsr1_easu.hip): 64x16 output tile per 256-thread
// workgroup, footprint staged once into LDS with clamp-to-edge applied, and everything FsrEasuH recomputes per
// output pixel although it depends on the input only — luma (:535-538), the per-position terms of FsrEasuSetH
// (:486-502), the (-x, x) min/max pairs of the 2x2 block (:575-577) — evaluated once per input texel, as the very
// binary16 operation sequences the reference runs per lane.  Texel channels are kept as (texel, right neighbour)
// half2 pairs, the operand shape of the two-taps-per-op FsrEasuTapH.  Position arithmetic stays fp32 (:513-516).
#include "fsr1_device.h"
#include "fsr1_device_half.hpp"

namespace fsr1 {

// LDS per footprint texel (48 bytes, the same budget as the fp32 kernel):
//   tex1  half4  R G B luma of the texel                                   (phase 1; luma = B*0.5 + (R*0.5 + G), :535-538)
//   texP  4 x half2  (R,G,B,luma) paired with the texel to the right       -> one ds_read_b128 per two-tap FsrEasuTapH call
//   ana1  half4  dirX dirY lenX lenY of FsrEasuSetH for the '+' around the texel (:476-503: they do not depend on the
//                output pixel, only their bilinear weights do), each a single binary16 operation sequence in the
//                reference's order, so the per-pixel accumulation that follows sees bit-identical operands
//   both  3 x half2  (max of -x, max of x) over the 2x2 block at the texel, per channel (:575-577)
struct EasuHLds {
  half4_t* tex1;
  uint4* texP;
  half4_t* ana1;
  uint4* both;
};

__global__ void __launch_bounds__(kThreads) easu_h_kernel(const EasuArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = a.fp_w * a.fp_h;
  EasuHLds l;
  l.texP = reinterpret_cast<uint4*>(smem);
  l.both = reinterpret_cast<uint4*>(smem + (size_t)cap * 16);
  l.tex1 = reinterpret_cast<half4_t*>(smem + (size_t)cap * 32);
  l.ana1 = reinterpret_cast<half4_t*>(smem + (size_t)cap * 40);

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int ox0 = tx * kTileW, oy0 = ty * kTileH;
  const float c0x = as_f32(a.con[0]), c0y = as_f32(a.con[1]), c0z = as_f32(a.con[2]), c0w = as_f32(a.con[3]);

  const int oxl = min(ox0 + kTileW, a.out.width) - 1, oyl = min(oy0 + kTileH, a.out.height) - 1;
  const int fx0 = (int)floorf((float)ox0 * c0x + c0z) - 1;
  const int fy0 = (int)floorf((float)oy0 * c0y + c0w) - 1;
  const int fw = min((int)floorf((float)oxl * c0x + c0z) + 2 - fx0 + 1, a.fp_w);
  const int fh = min((int)floorf((float)oyl * c0y + c0w) + 2 - fy0 + 1, a.fp_h);

  const int tid = threadIdx.x;
  const int n = fw * fh;
  const half_t one = (half_t)1.0f;
  {  // ---- phase 1: HBM -> LDS (clamp-to-edge applied), luma once per input texel ----
    const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;
    const float inv_fw = 1.0f / (float)fw;
    for (int i = tid; i < n; i += kThreads) {
      const int ly = (int)(((float)i + 0.5f) * inv_fw);
      const int lx = i - ly * fw;
      const int gy = min(max(fy0 + ly, 0), a.in.height - 1);
      const int gx = min(max(fx0 + lx, 0), a.in.width - 1);
      const half4_t c = *reinterpret_cast<const half4_t*>(in_frame + (long long)gy * a.in.pitch + (size_t)gx * sizeof(half4_t));
      const half_t hlf = (half_t)0.5f;
      l.tex1[i] = half4_t{c.x, c.y, c.z, (half_t)(c.z * hlf + (c.x * hlf + c.y))};  // :535-538
    }
  }
  __syncthreads();
  // ---- phase 2: per-texel terms (border texels read clamped neighbours and produce values nobody uses) ----
  for (int i = tid; i < n; i += kThreads) {
    const int iu = max(i - fw, 0), id = min(i + fw, n - 1), il = max(i - 1, 0), ir = min(i + 1, n - 1), idr = min(id + 1, n - 1);
    const half4_t tc = l.tex1[i], tr = l.tex1[ir], td = l.tex1[id], tdr = l.tex1[idr];
    const half_t lA = l.tex1[iu].w, lB = l.tex1[il].w, lC = tc.w, lD = tr.w, lE = td.w;
    l.ana1[i] = easu_analysis_h(lA, lB, lC, lD, lE);  // FsrEasuSetH :486-502, one AH2 lane
    l.texP[i] = uint4{as_u(h2(tc.x, tr.x)), as_u(h2(tc.y, tr.y)), as_u(h2(tc.z, tr.z)), as_u(h2(tc.w, tr.w))};
    // :575-577 min and max of the 2x2 block f g / j k through max() of (-x, x) pairs
    const half2_t bR = easu_both_h(tc.x, tr.x, td.x, tdr.x), bG = easu_both_h(tc.y, tr.y, td.y, tdr.y), bB = easu_both_h(tc.z, tr.z, td.z, tdr.z);
    l.both[i] = uint4{as_u(bR), as_u(bG), as_u(bB), 0u};
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int ox = ox0 + lane;
  if (ox >= a.out.width) return;
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)ox * sizeof(half4_t);
  float ppx = (float)ox * c0x + c0z;  // :513-515
  const float fpx = floorf(ppx);
  ppx -= fpx;
  const int lx = (int)fpx - fx0;
  const bool hdr = (a.flags & FSR1_FLAG_HDR_SQUARE) != 0;

#pragma unroll 1
  for (int r = 0; r < kTileH / 4; ++r) {
    const int oy = oy0 + wave * (kTileH / 4) + r;
    if (oy >= a.out.height) break;
    float ppy = (float)oy * c0y + c0w;
    const float fpy = floorf(ppy);
    ppy -= fpy;
    const half2_t ppp = h2((half_t)ppx, (half_t)ppy);  // :516 AH2(pp), RTNE
    const int f = ((int)fpy - fy0) * fw + lx;            // footprint index of texel 'f'
    // texP[t] holds (t, t+1), so the reference's pairs bc / ij / kl come as stored and fe / hg / on are the swapped
    // (e,f) / (g,h) / (n,o)
    const uint4 bo = l.both[f];
    const rgbh_t px = easu_filter_h(
        [&](int k) { return l.ana1[f + (k >> 1) * fw + (k & 1)]; },
        [&](int i) {
          const int at[6] = {f - fw, f + fw - 1, f - 1, f + fw + 1, f + 1, f + 2 * fw};
          const uint4 t = l.texP[at[i]];
          const bool sw = i == 2 || i == 4 || i == 5;
          return sw ? EasuHPair{swap2(as_h2(t.x)), swap2(as_h2(t.y)), swap2(as_h2(t.z))} : EasuHPair{as_h2(t.x), as_h2(t.y), as_h2(t.z)};
        },
        [&](int c) { return as_h2(c == 0 ? bo.x : (c == 1 ? bo.y : bo.z)); }, ppp);
    half_t pr = px.r, pg = px.g, pb = px.b;
    if (hdr) { pr = pr * pr; pg = pg * pg; pb = pb * pb; }  // FSR_Pass.hlsl:78-79
    *reinterpret_cast<half4_t*>(out_col + (long long)oy * a.out.pitch) = half4_t{pr, pg, pb, one};  // alpha = 1, FSR_Pass.hlsl:80
  }
}

size_t easu_h_lds_bytes(int fp_w, int fp_h) { return (size_t)fp_w * fp_h * 48; }

hipError_t easu_h_launch(const EasuArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const size_t lds = easu_h_lds_bytes(a.fp_w, a.fp_h);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&easu_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(easu_h_kernel, grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace fsr1
