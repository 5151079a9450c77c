// TEST INFRASTRUCTURE — not product code.
//
// Harness around the reference headers compiled verbatim (oracle/build_ref.sh puts a mechanically
// qualifier-rewritten copy of /root/reference/ffx-fsr/{ffx_a.h,ffx_fsr1.h} in a temp include dir
// as ref_ffx_a.h / ref_ffx_fsr1.h; nothing of the reference is stored in this repository).
// The harness plays the role of the dispatch shell sample/src/DX12/FSR_Pass.hlsl:39-62 (callbacks)
// and :68-104 (CurrFilter): gather4 with clamp-to-edge for EASU, integer Load with zero outside
// the image for RCAS, alpha written as 1.
//
// Exposed as a C ABI over plain float buffers (RGBA interleaved, 4 floats per pixel) so the tests
// and bench.py's cpu_baseline leg can call it through ctypes.  "H" entry points take/return floats
// whose values are binary16-representable.
#include "ref_glsl_shim.hpp"

#include <cstdint>
#include <cstdlib>
#include <omp.h>

namespace glsl {

struct Image {
  const float* p;
  int w, h;
};
static thread_local Image g_src;

// gather4 at normalized coordinate p for channel c: returns texels
//   x=(i,j+1) y=(i+1,j+1) z=(i+1,j) w=(i,j),  i=floor(p.x*W-0.5), j=floor(p.y*H-0.5), clamp-to-edge
// (D3D/Vulkan gather semantics; sampler is CLAMP: sample/src/DX12/FSR_Filter.cpp:48-53).
static inline float texel(int x, int y, int c) {
  x = x < 0 ? 0 : (x >= g_src.w ? g_src.w - 1 : x);
  y = y < 0 ? 0 : (y >= g_src.h ? g_src.h - 1 : y);
  return g_src.p[((size_t)y * g_src.w + x) * 4 + c];
}
static inline vec4 gather(vec2 p, int c) {
  int i = (int)std::floor(p.x * (float)g_src.w - 0.5f);
  int j = (int)std::floor(p.y * (float)g_src.h - 0.5f);
  return vec4(texel(i, j + 1, c), texel(i + 1, j + 1, c), texel(i + 1, j, c), texel(i, j, c));
}
// integer load; outside the resource returns 0 (D3D Load semantics, FSR_Pass.hlsl:45,61)
static inline vec4 load(int x, int y) {
  if (x < 0 || y < 0 || x >= g_src.w || y >= g_src.h) return vec4(0.f, 0.f, 0.f, 0.f);
  const float* q = g_src.p + ((size_t)y * g_src.w + x) * 4;
  return vec4(q[0], q[1], q[2], q[3]);
}

#define A_GPU 1
#define A_GLSL 1
#define A_SKIP_EXT 1
#define A_HALF 1
#define FSR_EASU_F 1
#define FSR_EASU_H 1
#define FSR_RCAS_F 1
#define FSR_RCAS_H 1
#define FSR_RCAS_HX2 1 /* the packed two-pixel form (ffx_fsr1.h:874-984), compiled as well: tests pin FsrRcasHx2 to FsrRcasH with it */

#define REF_CALLBACKS                                                                                   \
  vec4 FsrEasuRF(vec2 p) { return gather(p, 0); }                                                       \
  vec4 FsrEasuGF(vec2 p) { return gather(p, 1); }                                                       \
  vec4 FsrEasuBF(vec2 p) { return gather(p, 2); }                                                       \
  f16vec4 FsrEasuRH(vec2 p) { return f16vec4(gather(p, 0)); }                                           \
  f16vec4 FsrEasuGH(vec2 p) { return f16vec4(gather(p, 1)); }                                           \
  f16vec4 FsrEasuBH(vec2 p) { return f16vec4(gather(p, 2)); }                                           \
  vec4 FsrRcasLoadF(ivec2 p) { return load(p.x, p.y); }                                                 \
  void FsrRcasInputF(float& r, float& g, float& b) {}                                                   \
  f16vec4 FsrRcasLoadH(i16vec2 p) { return f16vec4(load(p.x, p.y)); }                                   \
  void FsrRcasInputH(float16_t& r, float16_t& g, float16_t& b) {}                                       \
  f16vec4 FsrRcasLoadHx2(i16vec2 p) { return f16vec4(load(p.x, p.y)); }                                 \
  void FsrRcasInputHx2(f16vec2& r, f16vec2& g, f16vec2& b) {}

// Four builds of the same header: the RCAS feature macros are compile-time (ffx_fsr1.h:647-651).
namespace plain {
REF_CALLBACKS
#include "ref_ffx_a.h"
#include "ref_ffx_fsr1.h"
}  // namespace plain
namespace denoise {
#define FSR_RCAS_DENOISE 1
REF_CALLBACKS
#include "ref_ffx_a.h"
#include "ref_ffx_fsr1.h"
#undef FSR_RCAS_DENOISE
}  // namespace denoise
namespace alpha {
#define FSR_RCAS_PASSTHROUGH_ALPHA 1
REF_CALLBACKS
#include "ref_ffx_a.h"
#include "ref_ffx_fsr1.h"
}  // namespace alpha
namespace alpha_denoise {
#define FSR_RCAS_DENOISE 1
REF_CALLBACKS
#include "ref_ffx_a.h"
#include "ref_ffx_fsr1.h"
#undef FSR_RCAS_DENOISE
#undef FSR_RCAS_PASSTHROUGH_ALPHA
}  // namespace alpha_denoise

static inline uvec4 con4(const uint32_t* c) { return uvec4(c[0], c[1], c[2], c[3]); }

}  // namespace glsl

using namespace glsl;

enum { REF_RCAS_DENOISE = 1, REF_RCAS_ALPHA = 2, REF_HDR_SQUARE = 4 };
// colour stages around the filters (ffx_fsr1.h:986-1199); same bit values as FSR1_COLOR_* in include/fsr1_hip.h
enum { REF_COLOR_SRTM = 1, REF_COLOR_LFGA = 2, REF_COLOR_SRTM_INV = 4, REF_COLOR_TEPD_C8 = 8, REF_COLOR_TEPD_C10 = 16,
       REF_COLOR_DITHER_FROM_NOISE = 32 };

extern "C" {

// Rows [y0,y1) of the EASU output.  con16 = con0..con3 as produced by FsrEasuCon.
// flags & REF_HDR_SQUARE reproduces `if (Sample.x == 1) c *= c;` (FSR_Pass.hlsl:78-79).
void ref_easu_f(const float* in, int inW, int inH, float* out, int outW, int outH, const uint32_t* con16,
                int flags, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    g_src = Image{in, inW, inH};
    for (int x = 0; x < outW; ++x) {
      vec3 c;
      plain::FsrEasuF(c, uvec2((uint)x, (uint)y), con4(con16), con4(con16 + 4), con4(con16 + 8), con4(con16 + 12));
      if (flags & REF_HDR_SQUARE) c *= c;
      float* o = out + ((size_t)y * outW + x) * 4;
      o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = 1.0f;
    }
  }
}

void ref_easu_h(const float* in, int inW, int inH, float* out, int outW, int outH, const uint32_t* con16,
                int flags, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    g_src = Image{in, inW, inH};
    for (int x = 0; x < outW; ++x) {
      f16vec3 c;
      plain::FsrEasuH(c, uvec2((uint)x, (uint)y), con4(con16), con4(con16 + 4), con4(con16 + 8), con4(con16 + 12));
      if (flags & REF_HDR_SQUARE) c *= c;
      float* o = out + ((size_t)y * outW + x) * 4;
      o[0] = c.x.v; o[1] = c.y.v; o[2] = c.z.v; o[3] = 1.0f;
    }
  }
}

// Rows [y0,y1) of RCAS on a W x H image.  flags: REF_RCAS_DENOISE, REF_RCAS_ALPHA, REF_HDR_SQUARE.
void ref_rcas_f(const float* in, int W, int H, float* out, const uint32_t* con, int flags, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    g_src = Image{in, W, H};
    for (int x = 0; x < W; ++x) {
      vec3 c;
      float a = 1.0f;
      uvec2 ip((uint)x, (uint)y);
      switch (flags & 3) {
        case 0: plain::FsrRcasF(c.r, c.g, c.b, ip, con4(con)); break;
        case REF_RCAS_DENOISE: denoise::FsrRcasF(c.r, c.g, c.b, ip, con4(con)); break;
        case REF_RCAS_ALPHA: alpha::FsrRcasF(c.r, c.g, c.b, a, ip, con4(con)); break;
        default: alpha_denoise::FsrRcasF(c.r, c.g, c.b, a, ip, con4(con)); break;
      }
      if (flags & REF_HDR_SQUARE) c *= c;
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = a;
    }
  }
}

void ref_rcas_h(const float* in, int W, int H, float* out, const uint32_t* con, int flags, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    g_src = Image{in, W, H};
    for (int x = 0; x < W; ++x) {
      f16vec3 c;
      float16_t a(1.0);
      uvec2 ip((uint)x, (uint)y);
      switch (flags & 3) {
        case 0: plain::FsrRcasH(c.r, c.g, c.b, ip, con4(con)); break;
        case REF_RCAS_DENOISE: denoise::FsrRcasH(c.r, c.g, c.b, ip, con4(con)); break;
        case REF_RCAS_ALPHA: alpha::FsrRcasH(c.r, c.g, c.b, a, ip, con4(con)); break;
        default: alpha_denoise::FsrRcasH(c.r, c.g, c.b, a, ip, con4(con)); break;
      }
      if (flags & REF_HDR_SQUARE) c *= c;
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c.x.v; o[1] = c.y.v; o[2] = c.z.v; o[3] = a.v;
    }
  }
}

// The packed form FsrRcasHx2 (ffx_fsr1.h:888-984): one call sharpens the pixels ip and ip + (8, 0) — "2 8x8 tiles in a 16x8
// region" — in the two halves of every operand; FsrRcasDepackHx2 (:880-886) turns the result back into two pixels.  The image is
// walked in 16-pixel-wide column blocks, the call at column x of a block's left half producing x and x + 8.
void ref_rcas_hx2(const float* in, int W, int H, float* out, const uint32_t* con, int flags, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    g_src = Image{in, W, H};
    for (int bx = 0; bx < W; bx += 16) {
      for (int x = bx; x < bx + 8 && x < W; ++x) {
        f16vec2 r, g, b, a(float16_t(1.0), float16_t(1.0));
        uvec2 ip((uint)x, (uint)y);
        switch (flags & 3) {
          case 0: plain::FsrRcasHx2(r, g, b, ip, con4(con)); break;
          case REF_RCAS_DENOISE: denoise::FsrRcasHx2(r, g, b, ip, con4(con)); break;
          case REF_RCAS_ALPHA: alpha::FsrRcasHx2(r, g, b, a, ip, con4(con)); break;
          default: alpha_denoise::FsrRcasHx2(r, g, b, a, ip, con4(con)); break;
        }
        if (flags & REF_HDR_SQUARE) { r *= r; g *= g; b *= b; }
        f16vec4 p0, p1;
        plain::FsrRcasDepackHx2(p0, p1, r, g, b);
        float* o = out + ((size_t)y * W + x) * 4;
        o[0] = p0.x.v; o[1] = p0.y.v; o[2] = p0.z.v; o[3] = a.x.v;
        if (x + 8 < W) { o += 32; o[0] = p1.x.v; o[1] = p1.y.v; o[2] = p1.z.v; o[3] = a.y.v; }
      }
    }
  }
}

// Colour stages, in this fixed order, on rows [y0,y1) of a W x H RGBA image (alpha passes through):
//   FsrSrtmF (:1042) -> FsrLfgaF (:1012) -> FsrSrtmInvF (:1044) -> FsrTepdC8F | FsrTepdC10F (:1097, :1113)
// grain  t = noise[slice][(y+noy) mod nH][(x+nox) mod nW].rgb + bias   ("tiled blue noise", :1000; slice = frame mod nS,
//                                                                       the sample stacks its slices likewise: FSR_Tonemapping.hlsl:87)
// dither   = FsrTepdDitF(uvec2(x,y), frame) (:1082), or saturate(noise.a) as in FSR_Tonemapping.hlsl:87.
// The result is NOT quantised here: the store conversion belongs to the image format.
void ref_color_f(const float* in, int W, int H, float* out, int stages, float amount, float bias, uint32_t frame,
                 const float* noise, int nW, int nH, int nS, int nox, int noy, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    for (int x = 0; x < W; ++x) {
      const float* q = in + ((size_t)y * W + x) * 4;
      vec3 c(q[0], q[1], q[2]);
      vec4 n(0.f, 0.f, 0.f, 0.f);
      if (noise) {
        const int ny = (int)((((long long)y + noy) % nH + nH) % nH), nx = (int)((((long long)x + nox) % nW + nW) % nW);
        const float* t = noise + (((size_t)(frame % (uint32_t)nS) * nH + ny) * nW + nx) * 4;
        n = vec4(t[0], t[1], t[2], t[3]);
      }
      if (stages & REF_COLOR_SRTM) plain::FsrSrtmF(c);
      if (stages & REF_COLOR_LFGA) plain::FsrLfgaF(c, vec3(n.x + bias, n.y + bias, n.z + bias), amount);
      if (stages & REF_COLOR_SRTM_INV) plain::FsrSrtmInvF(c);
      if (stages & (REF_COLOR_TEPD_C8 | REF_COLOR_TEPD_C10)) {
        const float dit = (stages & REF_COLOR_DITHER_FROM_NOISE) ? plain::ASatF1(n.w)
                                                                  : plain::FsrTepdDitF(uvec2((uint)x, (uint)y), frame);
        if (stages & REF_COLOR_TEPD_C8) plain::FsrTepdC8F(c, dit); else plain::FsrTepdC10F(c, dit);
      }
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = q[3];
    }
  }
}

// The same chain through the reference's half-precision entry points: FsrSrtmH (:1049) -> FsrLfgaH (:1019) ->
// FsrSrtmInvH (:1050) -> FsrTepdC8H | FsrTepdC10H (:1134, :1143) with FsrTepdDitH (:1125).  (The Hx2 forms, :1022,
// :1052-1055, :1153-1198, run the same operations per lane.)  Values are binary16-representable floats.
void ref_color_h(const float* in, int W, int H, float* out, int stages, float amount, float bias, uint32_t frame,
                 const float* noise, int nW, int nH, int nS, int nox, int noy, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    for (int x = 0; x < W; ++x) {
      const float* q = in + ((size_t)y * W + x) * 4;
      f16vec3 c = f16vec3(float16_t(q[0]), float16_t(q[1]), float16_t(q[2]));
      float n[4] = {0.f, 0.f, 0.f, 0.f};
      if (noise) {
        const int ny = (int)((((long long)y + noy) % nH + nH) % nH), nx = (int)((((long long)x + nox) % nW + nW) % nW);
        const float* t = noise + (((size_t)(frame % (uint32_t)nS) * nH + ny) * nW + nx) * 4;
        n[0] = t[0]; n[1] = t[1]; n[2] = t[2]; n[3] = t[3];
      }
      const float16_t hb(bias), ha(amount);
      if (stages & REF_COLOR_SRTM) plain::FsrSrtmH(c);
      if (stages & REF_COLOR_LFGA) plain::FsrLfgaH(c, f16vec3(float16_t(n[0]) + hb, float16_t(n[1]) + hb, float16_t(n[2]) + hb), ha);
      if (stages & REF_COLOR_SRTM_INV) plain::FsrSrtmInvH(c);
      if (stages & (REF_COLOR_TEPD_C8 | REF_COLOR_TEPD_C10)) {
        const float16_t dit = (stages & REF_COLOR_DITHER_FROM_NOISE) ? plain::ASatH1(float16_t(n[3]))
                                                                      : plain::FsrTepdDitH(uvec2((uint)x, (uint)y), frame);
        if (stages & REF_COLOR_TEPD_C8) plain::FsrTepdC8H(c, dit); else plain::FsrTepdC10H(c, dit);
      }
      float* o = out + ((size_t)y * W + x) * 4;
      o[0] = c.x.v; o[1] = c.y.v; o[2] = c.z.v; o[3] = q[3];
    }
  }
}

// FsrTepdDitF alone (:1082-1091), for the known-answer tests.
float ref_tepd_dit_f(uint32_t x, uint32_t y, uint32_t f) { return plain::FsrTepdDitF(uvec2(x, y), f); }

// The A_GPU build of the constant setup (same source lines as the A_CPU build, ffx_fsr1.h:156-225,
// 662-672) — exported so the tests can confirm both builds of the reference agree.
void ref_easu_con_gpu(uint32_t* con16, float vpW, float vpH, float inW, float inH, float outW, float outH) {
  uvec4 c0, c1, c2, c3;
  plain::FsrEasuCon(c0, c1, c2, c3, vpW, vpH, inW, inH, outW, outH);
  for (int i = 0; i < 4; ++i) { con16[i] = c0[i]; con16[4 + i] = c1[i]; con16[8 + i] = c2[i]; con16[12 + i] = c3[i]; }
}

// ARmp8x8 lane -> (x,y) remap used by the dispatch shell (ffx_a.h:2304, FSR_Pass.hlsl:110).
void ref_rmp8x8(uint32_t lane, uint32_t* xy) {
  uvec2 r = plain::ARmp8x8(lane);
  xy[0] = r.x; xy[1] = r.y;
}

int ref_omp_threads(void) { return omp_get_max_threads(); }
}
