// RCAS — robust contrast adaptive sharpening (FsrRcasF, ffx-fsr/ffx_fsr1.h:684-769) for gfx950.
//
// HBM-bound pass: 8 B read + 8 B written per pixel (RGBA16F).  A 256-thread workgroup stages the
// (64+2)x(16+2) input window of its 64x16 output tile in LDS (texels outside the image are 0, the
// D3D `Load` rule of the reference's callback, FSR_Pass.hlsl:45,61), then each lane sharpens the 4
// pixels of its column and every wave issues row-contiguous stores.
#include "fsr1_device.h"

namespace fsr1 {

constexpr int kRcasW = kTileW + 2;
constexpr int kRcasH = kTileH + 2;

// One pixel of FsrRcasF from its 5 taps (b above, d left, e centre, f right, h below).
template <bool EXACT>
__device__ __forceinline__ void rcas_pixel(float4_t b, float4_t d, float4_t e, float4_t f, float4_t h, float sharp,
                                           uint32_t flags, float& pr, float& pg, float& pb) {
  // :741-746 min and max of the ring, per channel
  const float mn4R = fminf(min3f(b.x, d.x, f.x), h.x), mn4G = fminf(min3f(b.y, d.y, f.y), h.y), mn4B = fminf(min3f(b.z, d.z, f.z), h.z);
  const float mx4R = fmaxf(max3f(b.x, d.x, f.x), h.x), mx4G = fmaxf(max3f(b.y, d.y, f.y), h.y), mx4B = fmaxf(max3f(b.z, d.z, f.z), h.z);
  // :748-755 limiters; "these need to be high precision RCPs": IEEE division when EXACT, v_rcp_f32 (1 ulp) otherwise.
  // 4*x and 4*x-4 are exact scalings, so fusing the latter does not change it (barring overflow).
  auto rcp = [](float x) { return EXACT ? 1.0f / x : __builtin_amdgcn_rcpf(x); };
  const float hitMinR = fminf(mn4R, e.x) * rcp(4.0f * mx4R);
  const float hitMinG = fminf(mn4G, e.y) * rcp(4.0f * mx4G);
  const float hitMinB = fminf(mn4B, e.z) * rcp(4.0f * mx4B);
  const float hitMaxR = (1.0f - fmaxf(mx4R, e.x)) * rcp(4.0f * mn4R + -4.0f);
  const float hitMaxG = (1.0f - fmaxf(mx4G, e.y)) * rcp(4.0f * mn4G + -4.0f);
  const float hitMaxB = (1.0f - fmaxf(mx4B, e.z)) * rcp(4.0f * mn4B + -4.0f);
  // :756-759  max() must return the non-NaN operand (0*inf on black pixels): v_max_f32 does.
  const float lobeR = fmaxf(-hitMinR, hitMaxR), lobeG = fmaxf(-hitMinG, hitMaxG), lobeB = fmaxf(-hitMinB, hitMaxB);
  float lobe = fmaxf(-(0.25f - (1.0f / 16.0f)), fminf(max3f(lobeR, lobeG, lobeB), 0.0f)) * sharp;
  if (flags & FSR1_FLAG_RCAS_DENOISE) {  // :731-739, :761-763
    const float bL = fmaf(b.z, 0.5f, fmaf(b.x, 0.5f, b.y)), dL = fmaf(d.z, 0.5f, fmaf(d.x, 0.5f, d.y));
    const float eL = fmaf(e.z, 0.5f, fmaf(e.x, 0.5f, e.y)), fL = fmaf(f.z, 0.5f, fmaf(f.x, 0.5f, f.y));
    const float hL = fmaf(h.z, 0.5f, fmaf(h.x, 0.5f, h.y));
    float nz = 0.25f * bL + 0.25f * dL + 0.25f * fL + 0.25f * hL - eL;
    nz = sat(fabsf(nz) * APrxMedRcpF1<EXACT>(max3f(max3f(bL, dL, eL), fL, hL) - min3f(min3f(bL, dL, eL), fL, hL)));
    nz = mad<EXACT>(-0.5f, nz, 1.0f);
    lobe *= nz;
  }
  // :765-768 resolve
  const float rcpL = APrxMedRcpF1<EXACT>(mad<EXACT>(4.0f, lobe, 1.0f));
  if (EXACT) {
    pr = (lobe * b.x + lobe * d.x + lobe * h.x + lobe * f.x + e.x) * rcpL;
    pg = (lobe * b.y + lobe * d.y + lobe * h.y + lobe * f.y + e.y) * rcpL;
    pb = (lobe * b.z + lobe * d.z + lobe * h.z + lobe * f.z + e.z) * rcpL;
  } else {
    pr = fmaf(lobe, (b.x + d.x) + (h.x + f.x), e.x) * rcpL;
    pg = fmaf(lobe, (b.y + d.y) + (h.y + f.y), e.y) * rcpL;
    pb = fmaf(lobe, (b.z + d.z) + (h.z + f.z), e.z) * rcpL;
  }
  if (flags & FSR1_FLAG_HDR_SQUARE) { pr *= pr; pg *= pg; pb *= pb; }  // FSR_Pass.hlsl:92-93
  if (EXACT) { pr = pinned(pr); pg = pinned(pg); pb = pinned(pb); }
}

template <int FMT, bool EXACT>
__global__ void __launch_bounds__(kThreads) rcas_kernel(const RcasArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
  __shared__ texel_t tile[kRcasH][kRcasW];

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int ox0 = tx * kTileW, oy0 = ty * kTileH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;

  for (int ly = wave; ly < kRcasH; ly += 4) {
    const int gy = oy0 - 1 + ly;
    const bool rowok = gy >= 0 && gy < a.in.height;
    const char* const row = in_frame + (long long)gy * a.in.pitch;
    for (int lx = lane; lx < kRcasW; lx += 64) {
      const int gx = ox0 - 1 + lx;
      texel_t px = Pixel<FMT>::zero();
      if (rowok && gx >= 0 && gx < a.in.width) px = *reinterpret_cast<const texel_t*>(row + (size_t)gx * sizeof(texel_t));
      tile[ly][lx] = px;
    }
  }
  __syncthreads();

  const int ox = ox0 + lane;
  if (ox >= a.out.width) return;
  const float sharp = as_f32(a.con[0]);
  char* const out_frame = a.out.base + (long long)frame * a.out.frame_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ry = wave * 4 + r;
    const int oy = oy0 + ry;
    if (oy >= a.out.height) break;
    const float4_t b = Pixel<FMT>::load(tile[ry][lane + 1]);
    const float4_t d = Pixel<FMT>::load(tile[ry + 1][lane]);
    const float4_t e = Pixel<FMT>::load(tile[ry + 1][lane + 1]);
    const float4_t f = Pixel<FMT>::load(tile[ry + 1][lane + 2]);
    const float4_t h = Pixel<FMT>::load(tile[ry + 2][lane + 1]);
    float pr, pg, pb;
    rcas_pixel<EXACT>(b, d, e, f, h, sharp, a.flags, pr, pg, pb);
    const float pa = (a.flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) ? e.w : 1.0f;  // :700-705 / FSR_Pass.hlsl:94
    *reinterpret_cast<texel_t*>(out_frame + (long long)oy * a.out.pitch + (size_t)ox * sizeof(texel_t)) =
        Pixel<FMT>::store(pr, pg, pb, pa);
  }
}

hipError_t rcas_launch(const RcasArgs& a, int fmt, bool exact, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  if (fmt == FSR1_FORMAT_RGBA16F) {
    if (exact) hipLaunchKernelGGL((rcas_kernel<FSR1_FORMAT_RGBA16F, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rcas_kernel<FSR1_FORMAT_RGBA16F, false>), grid, block, 0, stream, a);
  } else {
    if (exact) hipLaunchKernelGGL((rcas_kernel<FSR1_FORMAT_RGBA32F, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rcas_kernel<FSR1_FORMAT_RGBA32F, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace fsr1
