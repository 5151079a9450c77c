// An integrator's own HIP translation unit, written against include/ ONLY (no file of the library's csrc/ is visible
// to this compile: see the Makefile's -I list).  It plays the role of the reference's includer, sample/src/DX12/FSR_Pass.hlsl:
// defines the load callbacks, calls the per-pixel entry points FsrEasuF / FsrRcasF / FsrEasuH / FsrRcasH from kernels of
// its own shape (one thread per output pixel, 16x16 blocks), runs the reference's own shader shell (64 lanes, ARmp8x8, four
// pixels per lane; FsrRcasHx2 + FsrRcasDepackHx2 as the packed form of the same shell), and — last — uses the LDS-staged fast
// form the library itself runs.  tests/test_device_api.py checks every result bit-for-bit against the golden vectors generated
// from the reference headers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fsr1_device.hpp"

using namespace fsr1;

namespace {

struct Con { uint4 c0, c1, c2, c3; };

// ---- per-pixel entry points, caller-supplied callbacks (here: the ready-made pitch-linear ones) ----
template <bool EXACT>
__global__ void easu_f_kernel(ImageCallbacks<FSR1_FORMAT_RGBA32F> cb, float4* out, int ow, int oh, Con k, int hdr) {
  const uint2 ip = {blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y};
  if ((int)ip.x >= ow || (int)ip.y >= oh) return;
  float3 c;
  FsrEasuF<EXACT>(c, ip, k.c0, k.c1, k.c2, k.c3, cb);
  if (hdr) { c.x *= c.x; c.y *= c.y; c.z *= c.z; }          // FSR_Pass.hlsl:78-79
  out[(size_t)ip.y * ow + ip.x] = float4{c.x, c.y, c.z, 1.0f};  // FSR_Pass.hlsl:80
}

// an integrator's own callbacks: an input transform on every RCAS tap (the FsrRcasInputF hook, ffx_fsr1.h:682)
struct ScaledRcasCallbacks : ImageCallbacks<FSR1_FORMAT_RGBA32F> {
  float scale;
  __device__ void FsrRcasInputF(float& r, float& g, float& b) const { r *= scale; g *= scale; b *= scale; }
};

template <bool EXACT, bool DENOISE, bool ALPHA>
__global__ void rcas_f_kernel(ScaledRcasCallbacks cb, float4* out, int w, int h, uint4 con) {
  const uint2 ip = {blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y};
  if ((int)ip.x >= w || (int)ip.y >= h) return;
  float r, g, b, a = 1.0f;
  if (ALPHA) FsrRcasF<EXACT, DENOISE>(r, g, b, a, ip, con, cb);
  else FsrRcasF<EXACT, DENOISE>(r, g, b, ip, con, cb);
  out[(size_t)ip.y * w + ip.x] = float4{r, g, b, a};
}

__global__ void easu_h_kernel(ImageCallbacks<FSR1_FORMAT_RGBA16F> cb, half4_t* out, int ow, int oh, Con k) {
  const uint2 ip = {blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y};
  if ((int)ip.x >= ow || (int)ip.y >= oh) return;
  half3_t c;
  FsrEasuH(c, ip, k.c0, k.c1, k.c2, k.c3, cb);
  out[(size_t)ip.y * ow + ip.x] = half4_t{c.x, c.y, c.z, (half_t)1.0f};
}

template <bool DENOISE, bool ALPHA>
__global__ void rcas_h_kernel(ImageCallbacks<FSR1_FORMAT_RGBA16F> cb, half4_t* out, int w, int h, uint4 con) {
  const uint2 ip = {blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y};
  if ((int)ip.x >= w || (int)ip.y >= h) return;
  half_t r, g, b, a = (half_t)1.0f;
  if (ALPHA) FsrRcasH<DENOISE>(r, g, b, a, ip, con, cb);
  else FsrRcasH<DENOISE>(r, g, b, ip, con, cb);
  out[(size_t)ip.y * w + ip.x] = half4_t{r, g, b, a};
}

// ---- the reference's shader shell, FSR_Pass.hlsl:106-118, written the way an integrator who copies it would: a 64-lane workgroup,
//      gxy = ARmp8x8(lane) + 16 * workgroup, CurrFilter on the four 8x8 tiles of a 16x16 region.  PASS 0: SAMPLE_EASU (FsrEasuH),
//      PASS 1: SAMPLE_RCAS (FsrRcasH), PASS 2: the packed two-pixel entry point the header offers for the same shell — FsrRcasHx2
//      covers gxy and gxy + (8, 0) in one call (ffx_fsr1.h:889-893), so a lane makes two calls instead of four — plus
//      FsrRcasDepackHx2 for the store.  A texture store outside the resource is dropped by the API; here it is a bounds check. ----
template <int PASS, bool DENOISE>
__global__ void __launch_bounds__(64) shader_shell_kernel(ImageCallbacks<FSR1_FORMAT_RGBA16F> cb, half4_t* out, int ow, int oh, Con k, int hdr) {
  auto store = [&](uint2 pos, half_t r, half_t g, half_t b) {
    if (hdr) { r = r * r; g = g * g; b = b * b; }                                      // `if (Sample.x == 1) c *= c;`
    if ((int)pos.x < ow && (int)pos.y < oh) out[(size_t)pos.y * ow + pos.x] = half4_t{r, g, b, (half_t)1.0f};  // OutputTexture[pos] = AH4(c, 1)
  };
  auto CurrFilter = [&](uint2 pos) {
    if (PASS == 0) {
      half3_t c;
      FsrEasuH(c, pos, k.c0, k.c1, k.c2, k.c3, cb);
      store(pos, c.x, c.y, c.z);
    } else {
      half_t r, g, b;
      FsrRcasH<DENOISE>(r, g, b, pos, k.c0, cb);
      store(pos, r, g, b);
    }
  };
  const uint2 rm = ARmp8x8(threadIdx.x);
  uint2 gxy = {rm.x + (blockIdx.x << 4u), rm.y + (blockIdx.y << 4u)};
  if (PASS == 2) {
    for (int half_tile = 0; half_tile < 2; ++half_tile) {
      half2_t r2, g2, b2;
      FsrRcasHx2<DENOISE>(r2, g2, b2, gxy, k.c0, cb);
      half4_t p0, p1;
      FsrRcasDepackHx2(p0, p1, r2, g2, b2);
      store(gxy, p0.x, p0.y, p0.z);
      store(uint2{gxy.x + 8u, gxy.y}, p1.x, p1.y, p1.z);
      gxy.y += 8u;
    }
    return;
  }
  CurrFilter(gxy);
  gxy.x += 8u;
  CurrFilter(gxy);
  gxy.y += 8u;
  CurrFilter(gxy);
  gxy.x -= 8u;
  CurrFilter(gxy);
}

__global__ void rmp8x8_kernel(uint32_t* xy) {
  const uint2 r = ARmp8x8(threadIdx.x);
  xy[2 * threadIdx.x] = r.x;
  xy[2 * threadIdx.x + 1] = r.y;
}

// ---- the LDS-staged fast form: a 32 x 32 output tile per 256-thread block (a shape of this file's own choosing) ----
constexpr int kTile = 32;
template <bool EXACT>
__global__ void __launch_bounds__(256) easu_tiled_kernel(ImageView in, float4* out, int ow, int oh, Con k, int cap_w, int cap_h) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EasuLds l = easu_lds_carve(smem, cap_w * cap_h);
  const int ox0 = blockIdx.x * kTile, oy0 = blockIdx.y * kTile;
  const float sx = as_f32(k.c0.x), sy = as_f32(k.c0.y), bx = as_f32(k.c0.z), by = as_f32(k.c0.w);
  const int oxl = min(ox0 + kTile, ow) - 1, oyl = min(oy0 + kTile, oh) - 1;
  // footprint of the tile: fp(first pixel) - 1 .. fp(last pixel) + 2 per axis (ffx_fsr1.h:324-342)
  const int fx0 = (int)floorf((float)ox0 * sx + bx) - 1, fy0 = (int)floorf((float)oy0 * sy + by) - 1;
  const int fw = min((int)floorf((float)oxl * sx + bx) + 2 - fx0 + 1, cap_w);
  const int fh = min((int)floorf((float)oyl * sy + by) + 2 - fy0 + 1, cap_h);
  l.fw = fw;
  easu_stage_footprint<FSR1_FORMAT_RGBA32F, false, EXACT>(l, in, in.base, fx0, fy0, fw, fh, (int)threadIdx.x);
  for (int p = threadIdx.x; p < kTile * kTile; p += 256) {
    const int ox = ox0 + (p % kTile), oy = oy0 + (p / kTile);
    if (ox >= ow || oy >= oh) continue;
    float ppx = (float)ox * sx + bx, ppy = (float)oy * sy + by;
    const float fpx = floorf(ppx), fpy = floorf(ppy);
    ppx -= fpx;
    ppy -= fpy;
    const int f_idx = ((int)fpy - fy0) * fw + ((int)fpx - fx0);
    const rgbf_t q = easu_clamp<EXACT>(easu_bounds(l, f_idx), easu_pixel<EXACT>(l, f_idx, ppx, ppy), false);
    out[(size_t)oy * ow + ox] = float4{q.r, q.g, q.b, 1.0f};
  }
}

Con make_con(const uint32_t* c) {
  return Con{uint4{c[0], c[1], c[2], c[3]}, uint4{c[4], c[5], c[6], c[7]}, uint4{c[8], c[9], c[10], c[11]}, uint4{c[12], c[13], c[14], c[15]}};
}
dim3 grid2(int w, int h) { return dim3((w + 15) / 16, (h + 15) / 16); }
int done() {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

// C entry points for the test (device pointers, tightly packed images)
extern "C" {

int outside_easu_f(const void* in, int iw, int ih, void* out, int ow, int oh, const uint32_t* con16, int exact, int hdr) {
  const ImageCallbacks<FSR1_FORMAT_RGBA32F> cb = {static_cast<const char*>(in), iw, ih, (long long)iw * 16};
  if (exact) hipLaunchKernelGGL(easu_f_kernel<true>, grid2(ow, oh), dim3(16, 16), 0, nullptr, cb, static_cast<float4*>(out), ow, oh, make_con(con16), hdr);
  else hipLaunchKernelGGL(easu_f_kernel<false>, grid2(ow, oh), dim3(16, 16), 0, nullptr, cb, static_cast<float4*>(out), ow, oh, make_con(con16), hdr);
  return done();
}

// variant bits: 1 = FSR_RCAS_DENOISE, 2 = FSR_RCAS_PASSTHROUGH_ALPHA; input_scale: the FsrRcasInputF hook (1 = none)
int outside_rcas_f(const void* in, int w, int h, void* out, const uint32_t* con4, int exact, int variant, float input_scale) {
  ScaledRcasCallbacks cb;
  cb.base = static_cast<const char*>(in); cb.width = w; cb.height = h; cb.pitch = (long long)w * 16; cb.scale = input_scale;
  const uint4 con = {con4[0], con4[1], con4[2], con4[3]};
  float4* o = static_cast<float4*>(out);
#define RUN(E, D, A) hipLaunchKernelGGL((rcas_f_kernel<E, D, A>), grid2(w, h), dim3(16, 16), 0, nullptr, cb, o, w, h, con)
  switch ((exact ? 4 : 0) | (variant & 3)) {
    case 0: RUN(false, false, false); break; case 1: RUN(false, true, false); break; case 2: RUN(false, false, true); break; case 3: RUN(false, true, true); break;
    case 4: RUN(true, false, false); break; case 5: RUN(true, true, false); break; case 6: RUN(true, false, true); break; default: RUN(true, true, true); break;
  }
#undef RUN
  return done();
}

int outside_easu_h(const void* in, int iw, int ih, void* out, int ow, int oh, const uint32_t* con16) {
  const ImageCallbacks<FSR1_FORMAT_RGBA16F> cb = {static_cast<const char*>(in), iw, ih, (long long)iw * 8};
  hipLaunchKernelGGL(easu_h_kernel, grid2(ow, oh), dim3(16, 16), 0, nullptr, cb, static_cast<half4_t*>(out), ow, oh, make_con(con16));
  return done();
}

int outside_rcas_h(const void* in, int w, int h, void* out, const uint32_t* con4, int variant) {
  const ImageCallbacks<FSR1_FORMAT_RGBA16F> cb = {static_cast<const char*>(in), w, h, (long long)w * 8};
  const uint4 con = {con4[0], con4[1], con4[2], con4[3]};
  half4_t* o = static_cast<half4_t*>(out);
#define RUN(D, A) hipLaunchKernelGGL((rcas_h_kernel<D, A>), grid2(w, h), dim3(16, 16), 0, nullptr, cb, o, w, h, con)
  switch (variant & 3) { case 0: RUN(false, false); break; case 1: RUN(true, false); break; case 2: RUN(false, true); break; default: RUN(true, true); break; }
#undef RUN
  return done();
}

// pass: 0 = EASU (con16 = FsrEasuCon's words), 1 = RCAS through FsrRcasH, 2 = RCAS through FsrRcasHx2 (con16[0..3] = FsrRcasCon's words);
// `in` is iw x ih, `out` ow x oh (RCAS: the same size); denoise = FSR_RCAS_DENOISE; hdr = Sample.x
int outside_shader_shell(const void* in, int iw, int ih, void* out, int ow, int oh, const uint32_t* con16, int pass, int denoise, int hdr) {
  const ImageCallbacks<FSR1_FORMAT_RGBA16F> cb = {static_cast<const char*>(in), iw, ih, (long long)iw * 8};
  const dim3 grid((ow + 15) / 16, (oh + 15) / 16);
  half4_t* o = static_cast<half4_t*>(out);
  Con k = make_con(con16);
#define RUN(P, D) hipLaunchKernelGGL((shader_shell_kernel<P, D>), grid, dim3(64), 0, nullptr, cb, o, ow, oh, k, hdr)
  switch (pass * 2 + (denoise ? 1 : 0)) {
    case 0: case 1: RUN(0, false); break;
    case 2: RUN(1, false); break; case 3: RUN(1, true); break;
    case 4: RUN(2, false); break; case 5: RUN(2, true); break;
    default: return -1;
  }
#undef RUN
  return done();
}

// ARmp8x8 of lanes 0 .. 63 -> xy[128] (device pointer)
int outside_rmp8x8(void* xy) {
  hipLaunchKernelGGL(rmp8x8_kernel, dim3(1), dim3(64), 0, nullptr, static_cast<uint32_t*>(xy));
  return done();
}

int outside_easu_tiled(const void* in, int iw, int ih, void* out, int ow, int oh, const uint32_t* con16, int exact) {
  ImageView v;
  v.base = const_cast<char*>(static_cast<const char*>(in));
  v.width = iw; v.height = ih; v.pitch = (long long)iw * 16; v.frame_stride = v.pitch * ih;
  float sx, sy;
  __builtin_memcpy(&sx, &con16[0], 4);
  __builtin_memcpy(&sy, &con16[1], 4);
  const int cap_w = (int)(kTile * sx) + 6, cap_h = (int)(kTile * sy) + 6;  // generous: fp(last) - fp(first) + 4 <= tile*scale + 5
  const size_t lds = (size_t)cap_w * cap_h * kEasuLdsPerTexel;
  if (lds > 64 * 1024) return -1;
  const dim3 grid((ow + kTile - 1) / kTile, (oh + kTile - 1) / kTile);
  if (exact) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&easu_tiled_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(easu_tiled_kernel<true>, grid, dim3(256), lds, nullptr, v, static_cast<float4*>(out), ow, oh, make_con(con16), cap_w, cap_h);
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&easu_tiled_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(easu_tiled_kernel<false>, grid, dim3(256), lds, nullptr, v, static_cast<float4*>(out), ow, oh, make_con(con16), cap_w, cap_h);
  }
  return done();
}

}  // extern "C"
