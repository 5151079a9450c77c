// RCAS — robust contrast adaptive sharpening (FsrRcasF, ffx-fsr/ffx_fsr1.h:684-769) for gfx950.
//
// 8 B read + 8 B written per pixel (RGBA16F) against ~100 VALU instructions: on MI355X the pass sits
// right at the HBM/VALU balance point, so the kernel is built to touch every input byte once and to
// spend no instruction on staging:
//   * no LDS, no barrier: a wave owns a 64-column x 24-row strip and streams down it; the vertical
//     neighbours (b above, h below) are the lane's own previous/next rows kept in registers, the
//     horizontal neighbours (d left, f right) are the adjacent lanes' centre texels fetched with DPP
//     wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1);
//   * lanes 0 and 63 additionally load the one texel left / right of the strip, which the DPP move
//     leaves in place for exactly those lanes (an invalid DPP source keeps the old destination);
//   * every texel is converted to fp32 once; loads run 3 rows ahead of the arithmetic;
//   * texels outside the image are 0 (the D3D `Load` rule of the reference's callback, FSR_Pass.hlsl:45,61).
#include "fsr1_device.h"
#include "fsr1_rcas_math.h"

namespace fsr1 {

constexpr int kRcasRows = 24;             // rows per strip: 1080, 2160 and 4320 are multiples
constexpr int kRcasCols = 64 * 4;         // columns per 256-thread workgroup (4 waves side by side)
constexpr int kDppWaveShr1 = 0x138;       // lane i <- lane i-1
constexpr int kDppWaveShl1 = 0x130;       // lane i <- lane i+1

// raw texel of the lane to the left / right; `keep` stays where the neighbour lane does not exist
template <int CTRL>
__device__ __forceinline__ half4_t neighbour(const half4_t& keep, const half4_t& v) {
  const uint2 c = __builtin_bit_cast(uint2, v), k = __builtin_bit_cast(uint2, keep);
  const uint2 r = {(uint32_t)__builtin_amdgcn_update_dpp((int)k.x, (int)c.x, CTRL, 0xf, 0xf, false),
                   (uint32_t)__builtin_amdgcn_update_dpp((int)k.y, (int)c.y, CTRL, 0xf, 0xf, false)};
  return __builtin_bit_cast(half4_t, r);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float keep, float v) {  // by value: bit_cast of a vector-element lvalue reads element 0
  return as_f32((uint32_t)__builtin_amdgcn_update_dpp((int)as_u32(keep), (int)as_u32(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float4_t neighbour(const float4_t& keep, const float4_t& v) {
  const float kx = keep.x, ky = keep.y, kz = keep.z, vx = v.x, vy = v.y, vz = v.z;
  return float4_t{dpp_f32<CTRL>(kx, vx), dpp_f32<CTRL>(ky, vy), dpp_f32<CTRL>(kz, vz), 0.0f};
}

// OPTS = false: the plain pass (no denoise / alpha pass-through / HDR square), flags compiled out.
template <int FMT, bool EXACT, bool OPTS>
__global__ void __launch_bounds__(kThreads) rcas_kernel(const RcasArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
  const uint32_t flags = OPTS ? a.flags : 0u;
  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = tx * kRcasCols + wave * 64, y0 = ty * kRcasRows;
  if (x0 >= a.in.width) return;  // whole wave outside (no barriers in this kernel)

  const int W = a.in.width, H = a.in.height;
  const int col = x0 + lane;
  const bool col_ok = col < W;
  // the strip's left / right apron column, owned by lanes 0 / 63
  const int hcol = lane == 0 ? x0 - 1 : x0 + 64;
  const bool halo_ok = (lane == 0 || lane == 63) && hcol >= 0 && hcol < W;
  const char* const in_col = a.in.base + (long long)frame * a.in.frame_stride + (size_t)col * sizeof(texel_t);
  const char* const in_hcol = a.in.base + (long long)frame * a.in.frame_stride + (size_t)hcol * sizeof(texel_t);
  char* const out_col = a.out.base + (long long)frame * a.out.frame_stride + (size_t)col * sizeof(texel_t);

  auto load = [&](int y, texel_t& own, texel_t& halo) {
    own = Pixel<FMT>::zero();
    halo = Pixel<FMT>::zero();
    if (y >= 0 && y < H) {  // wave-uniform
      if (col_ok) own = *reinterpret_cast<const texel_t*>(in_col + (long long)y * a.in.pitch);
      if (halo_ok) halo = *reinterpret_cast<const texel_t*>(in_hcol + (long long)y * a.in.pitch);
    }
  };
  auto rgb = [](const texel_t& p) { const float4_t c = Pixel<FMT>::load(p); return rgb_t{c.x, c.y, c.z}; };
  // Loop-carried fp32 values reach v_min/v_max through a block boundary, where the compiler no longer
  // knows they are canonical and would spend a v_max_f32 x,x,x (4.3 cycles) on each; x+0.0 (2.4 cycles)
  // tells it the same thing.  It turns -0 into +0, so the EXACT variant does not use it.
  auto known = [](rgb_t v) { return EXACT ? v : rgb_t{v.r + 0.0f, v.g + 0.0f, v.b + 0.0f}; };

  // Ring of raw rows: slot (r+k) % kRing holds row y0+r+k, k = 0..kAhead; with the row loop fully
  // unrolled every index is static, so the ring lives in registers and loads run kAhead rows ahead.
  constexpr int kAhead = FMT == FSR1_FORMAT_RGBA16F ? 6 : 3, kRing = kAhead + 1;
  texel_t q[kRing], g[kRing];
  rgb_t prev, cur;
  {
    texel_t q_prev, g_prev;
    load(y0 - 1, q_prev, g_prev);
#pragma unroll
    for (int k = 0; k < kAhead; ++k) load(y0 + k, q[k], g[k]);
    prev = rgb(q_prev);
    cur = rgb(q[0]);
  }
  const float sharp = as_f32(a.con[0]);

#pragma unroll
  for (int r = 0; r < kRcasRows; ++r) {
    const int y = y0 + r;
    load(y + kAhead, q[(r + kAhead) % kRing], g[(r + kAhead) % kRing]);
    const texel_t& q_cur = q[r % kRing];
    const texel_t& g_cur = g[r % kRing];
    const rgb_t next = rgb(q[(r + 1) % kRing]);
    // horizontal neighbours: the adjacent lanes' raw centre texel; lanes 0 / 63 keep their apron texel
    const texel_t dq = neighbour<kDppWaveShr1>(g_cur, q_cur), fq = neighbour<kDppWaveShl1>(g_cur, q_cur);
    const rgb_t p = rcas_pixel<EXACT>(known(prev), rgb(dq), known(cur), rgb(fq), next, sharp, flags);
    if (col_ok && y < H) {
      const float pa = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) ? Pixel<FMT>::load(q_cur).w : 1.0f;  // :700-705 / FSR_Pass.hlsl:94
      *reinterpret_cast<texel_t*>(out_col + (long long)y * a.out.pitch) = Pixel<FMT>::store(p.r, p.g, p.b, pa);
    }
    prev = cur; cur = next;
  }
}

void rcas_geometry(int width, int height, int* tiles_x, int* tiles_y) {
  *tiles_x = (width + kRcasCols - 1) / kRcasCols;
  *tiles_y = (height + kRcasRows - 1) / kRcasRows;
}

hipError_t rcas_launch(const RcasArgs& a, int fmt, bool exact, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(kThreads);
  const bool opts = (a.flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA | FSR1_FLAG_HDR_SQUARE)) != 0;
#define FSR1_RCAS(F, E, O) hipLaunchKernelGGL((rcas_kernel<F, E, O>), grid, block, 0, stream, a)
#define FSR1_RCAS_O(F, E) do { if (opts) FSR1_RCAS(F, E, true); else FSR1_RCAS(F, E, false); } while (0)
  if (fmt == FSR1_FORMAT_RGBA16F) { if (exact) FSR1_RCAS_O(FSR1_FORMAT_RGBA16F, true); else FSR1_RCAS_O(FSR1_FORMAT_RGBA16F, false); }
  else { if (exact) FSR1_RCAS_O(FSR1_FORMAT_RGBA32F, true); else FSR1_RCAS_O(FSR1_FORMAT_RGBA32F, false); }
#undef FSR1_RCAS_O
#undef FSR1_RCAS
  return hipGetLastError();
}

}  // namespace fsr1
