// FsrEasuF building blocks (ffx-fsr/ffx_fsr1.h:315-437) shared by the EASU kernel and the fused
// EASU->RCAS kernel: footprint staging (phases 1-2) and the per-pixel filter (phase 3).
//
// MI355X cost model that shaped them (tools/ubench/ubench2.hip, measured): v_fma/v_mul/v_add_f32
// issue at ~2.4 cycles per wave64 instruction, while v_min/v_max/v_cvt/v_fma_mix, integer and every packed
// (v_pk_*) instruction take ~4.3 and v_rcp/v_rsq ~8.5.  So: fp32 texels in LDS (no per-tap
// conversions), plain v_fma_f32 everywhere, the window clip done by the free `clamp` modifier
// instead of v_min_f32, and a staging pass that spends as few of the expensive integer / convert / min-max
// instructions per texel as it can (round 2: staging was a fifth of the kernel's issue time).
#pragma once
#include "fsr1_device_base.hpp"
#include "fsr1_device_color.hpp"

namespace fsr1 {

// LDS bytes per footprint texel: fp32 texel (R,G,B,luma*2) + analysis
// (a separate luma plane makes phase 2's five ds_read_b32 conflict-free — from the 16-byte records they are four dwords apart:
// 4-way bank conflicts — but its 4 bytes per texel cost a resident workgroup at 1.3x and the exact-2x kernels ran 1-2 % slower
// with it: profiles/ab_r03/r3c3_generic_easu_lane_columns_ab.log; phase 2 is a twentieth of the kernel)
constexpr int kEasuLdsPerTexel = 16 + 16;
// F-strict (see the end of this file): threshold of the rounding-boundary test in units of 2^-24 x the window's magnitude
constexpr float kEasuStrictK = 56.0f;
constexpr float kEasuStrictScale = kEasuStrictK * 0x1p-24f;
// bytes of the staged footprint (a multiple of 16: whatever a kernel carves behind it stays aligned)
__host__ __device__ constexpr size_t easu_lds_region_bytes(size_t capacity_texels) { return capacity_texels * kEasuLdsPerTexel; }

struct EasuLds {
  float4_t* tex;  // [n] R G B luma*2
  float4_t* ana;  // [n] FsrEasuSetF terms of the '+' around the texel: dirX dirY lenX^2 lenY^2 (EXACT) / dirX dirY lenX^2+lenY^2 - (default)
  int fw;         // row pitch of tex / ana in texels (dense layout: the footprint width)
};

__device__ __forceinline__ EasuLds easu_lds_carve(char* smem, int capacity_texels) {
  EasuLds l;
  l.tex = reinterpret_cast<float4_t*>(smem);
  l.ana = reinterpret_cast<float4_t*>(smem + (size_t)capacity_texels * 16);
  l.fw = 0;  // the caller sets the footprint width
  return l;
}

// Row-interleaved layout with a compile-time pitch P >= the footprint width: footprint row r is [P texels][P analyses], so
// `tex`, `ana` and the row pitch `fw` = 2 P are compile-time offsets from one base — every tap, analysis and bounds read of a
// pixel is then `ds_read_b128 v_base offset:imm`, with no address arithmetic per row (the dense layout of a run-time footprint
// width costs a v_add per tap row and array: 14 per pixel in the generic kernel).
template <int P>
__device__ __forceinline__ EasuLds easu_lds_carve_pitched(char* smem) {
  EasuLds l;
  l.tex = reinterpret_cast<float4_t*>(smem);
  l.ana = l.tex + P;
  l.fw = 2 * P;  // row r: [P texels][P analyses]
  return l;
}

// Lane -> pixel column inside a 64-column tile row for kernels whose lanes read ONE 16-byte LDS record each at a stride
// below one record per column (any ratio above 1x: column c reads texel floor(c * in/out + b)).  A wave64 ds_read_b128 is
// served in four groups of sixteen lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) — which makes unit-stride records conflict-free; at a stride of 2/3 record the sixteen lanes of a
// group, spread over 28 lane numbers, reach 19 texels and wrap around the 64 banks (2-way conflicts on half of the reads: the
// generic EASU kernel spent 50 % of its LDS cycles on them, profiles/r02_1440p_to_4k_two-pass.json).  Giving each group
// sixteen CONSECUTIVE columns keeps its texels within sixteen records for every ratio >= 1x.  A permutation inside the
// wave's 64 columns: a row's store still covers the same 512 contiguous bytes.
__device__ __forceinline__ int easu_lane_column(int lane) {
  const int l = lane & 31;
  //   lanes  0- 3 -> columns  0- 3      lanes  4-11 -> 16-23      lanes 12-15 ->  4- 7
  //   lanes 16-19 -> columns 24-27      lanes 20-27 ->  8-15      lanes 28-31 -> 28-31
  const int c = l < 4 ? l : (l < 12 ? l + 12 : (l < 16 ? l - 8 : (l < 20 ? l + 8 : (l < 28 ? l - 12 : l))));
  return (lane & 32) | c;
}

// max(|x|, |y|, |z|) of a texel record and a plain three-way maximum, as single v_max3_f32 (operands straight from LDS: see min4_asm)
__device__ __forceinline__ float absmax3(float4_t t) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(t.x), "v"(t.y), "v"(t.z));
  return r;
}
__device__ __forceinline__ float vmax_asm(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin_asm(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// FsrEasuSetF's terms for one position of the '+' neighbourhood  a / b c d / e  (ffx_fsr1.h:295-313), before the
// bilinear weighting: they depend on the input image only, so the tiled form evaluates them once per input texel.
// Reference order, no contraction.  Returns (dirX, dirY, lenX^2, lenY^2) when EXACT and (dirX, dirY, lenX^2 + lenY^2, 0)
// otherwise: `len` is continuous in its inputs and does not feed the zero test, so the default arithmetic adds the two
// squares here, once per texel, instead of once per pixel and position.
template <bool EXACT>
__device__ __forceinline__ float4_t easu_analysis(float lA, float lB, float lC, float lD, float lE) {
  const float dc = lD - lC, cb = lC - lB;
  float lenX = APrxLoRcpF1(fmaxf(fabsf(dc), fabsf(cb)));
  const float dirX = lD - lB;
  lenX = sat(fabsf(dirX) * lenX);
  lenX *= lenX;
  const float ec = lE - lC, ca = lC - lA;
  float lenY = APrxLoRcpF1(fmaxf(fabsf(ec), fabsf(ca)));
  const float dirY = lE - lA;
  lenY = sat(fabsf(dirY) * lenY);
  lenY *= lenY;
  return EXACT ? float4_t{dirX, dirY, lenX, lenY} : float4_t{dirX, dirY, lenX + lenY, 0.0f};
}

// Phases 1 and 2 for the footprint [fx0, fx0+fw) x [fy0, fy0+fh) of input texels (unclamped
// coordinates; the sampler's clamp-to-edge, FSR_Filter.cpp:48-53, is applied while loading).
// Ends with a barrier: afterwards every thread may read any footprint entry that a 12-tap window can touch
// (the bottom-right corner texel of the footprint is touched by none — no window has a (2, 2) tap — and is not staged:
// at exactly 2x that makes the 35 x 11 footprint 384 texels = six full wave-iterations instead of six and one lane).
// PRE: the colour prologue (FsrSrtmF, fsr1_device_color.hpp) is applied to every texel as it is loaded.
// FW, FH: compile-time footprint extent (the exact-2x variant: every tile has the same one), 0 = run-time.
// The host guarantees fh * pitch < 2^31 (fsr1_api.hip), so texel addresses are a wave-uniform 64-bit row base plus a
// 32-bit lane offset (global_load ... v_off, s[base]: no 64-bit vector arithmetic).
// THREADS: threads of the workgroup, all of which must make the call (it contains two barriers).
// PITCH: 0 = dense arrays of fw-texel rows; P = the row-interleaved layout of easu_lds_carve_pitched<P>.
// STRICT (default arithmetic only): phase 2 additionally leaves, in the analysis record's fourth component, the largest |R|, |G|, |B| of
// the texel's '+' neighbourhood — the four records of a pixel (f g j k) then cover exactly its 12 taps: the scale of the F-strict
// rounding-boundary test (easu_strict_* below).
template <int FMT, bool PRE = false, bool EXACT = false, int FW = 0, int FH = 0, int THREADS = 256, int PITCH = 0, bool STRICT = false>
__device__ __forceinline__ void easu_stage_footprint(const EasuLds& l, const ImageView& in, const char* in_frame, int fx0, int fy0,
                                                     int fw_rt, int fh_rt, int tid, const ColorArgs* color = nullptr) {
  typedef typename Pixel<FMT>::T texel_t;
  const int fw = FW ? FW : fw_rt, fh = FH ? FH : fh_rt;
  constexpr int kRow = PITCH ? 2 * PITCH : 0;  // LDS row pitch (records) of the interleaved layout
  auto texel_record = [&](const texel_t& px) {
    float4_t c = Pixel<FMT>::load(px);
    if constexpr (PRE) c = color_prologue<EXACT>(*color, c);
    // :363-366  luma*2 = B*0.5 + (R*0.5 + G); the products by 0.5 are exact, so fusing them is too
    return float4_t{c.x, c.y, c.z, fmaf(c.z, 0.5f, fmaf(c.x, 0.5f, c.y))};
  };
  // ---- phase 1: HBM -> LDS, one coalesced pass, fp32 once per input texel ----
  if (PITCH || FW || fw <= 64) {
    // Footprints up to 64 texels wide (every upscaling ratio; compile-time for the pitched and the exact-2x kernels): ONE LANE PER
    // FOOTPRINT COLUMN, ONE WAVE PER FOOTPRINT ROW (round 5).  A row is then wave-uniform — its clamp and its 64-bit base address
    // are scalar work, the column clamp is done once per lane, the LDS record of (row, lane) is a base register plus a scalar /
    // immediate offset — and a wave issues the loads of ALL its rows before it converts the first: up to kU requests per lane in
    // flight instead of one after the other.  The linear walk below spends 11 of its 19 VALU instructions per texel on index
    // arithmetic and waits for every load before it issues the next.  Lanes beyond the footprint's width idle (a row is one
    // wave-instruction either way).  Measured (profiles/ab_r05/r5c2_ab_stagerows.log, EASU us): 1440p -> 4K 53.8 -> 51.8-52.2,
    // 2954x1662 -> 4K 59.4 -> 56.0, 1477x831 -> 1080p 18.9 -> 17.2, 8-frame 1440p -> 4K batch 406 -> 395.
    static_assert(PITCH <= 64 && FW <= 64 && THREADS % 64 == 0, "one lane per footprint column");
    constexpr int kWaves = THREADS / 64;
    // rows per wave and trip: all of a compile-time footprint's, five otherwise (16-row tiles of any upscaling ratio have <= 20
    // rows), but no more than 48 bytes of raw texels per lane (RGBA32F: three rows — the exact-2x kernels have no registers to spare)
    constexpr int kWant = FH ? (FH + kWaves - 1) / kWaves : 5, kFit = 48 / (int)sizeof(texel_t);
    constexpr int kU = kWant < kFit ? kWant : kFit;
    const int rs = PITCH ? kRow : fw;                        // LDS records between footprint rows
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t xoff = (uint32_t)min(max(fx0 + lane, 0), in.width - 1) * (uint32_t)sizeof(texel_t);
    float4_t* const rec = l.tex + lane;
    if (lane < fw) {
#pragma unroll 1
      for (int ly0 = wave; ly0 < fh; ly0 += kWaves * kU) {
        texel_t px[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int ly = ly0 + kWaves * u;  // wave-uniform
          if (ly < fh) {
            const int gy = min(max(fy0 + ly, 0), in.height - 1);
            px[u] = *reinterpret_cast<const texel_t*>(in_frame + (long long)gy * in.pitch + (size_t)xoff);
          }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int ly = ly0 + kWaves * u;
          if (ly < fh) rec[ly * rs] = texel_record(px[u]);
        }
      }
    }
  } else {
    // wider footprints (minification): the linear walk, 256 texels per pass
    const int n = fw * fh - 1;
    // (v_rcp_f32, 1 ulp, instead of the IEEE division's dozen instructions: (i + 0.5) / fw lies at least 0.5 / fw from an integer, the
    //  product's error is below 1e-4 for a footprint's few thousand texels)
    const float inv_fw = __builtin_amdgcn_rcpf((float)fw);
    auto row_of = [&](int i) { return (int)(((float)i + 0.5f) * inv_fw); };
    const int gy0 = min(max(fy0, 0), in.height - 1);  // first row the footprint reads
    const char* const base = in_frame + (long long)gy0 * in.pitch;
    const uint32_t pitch = (uint32_t)in.pitch;
    for (int i = tid; i < n; i += THREADS) {
      const int ly = row_of(i);
      const int gy = min(max(fy0 + ly, 0), in.height - 1);
      const int gx = min(max(fx0 + (i - ly * fw), 0), in.width - 1);
      l.tex[i] = texel_record(*reinterpret_cast<const texel_t*>(base + (size_t)((uint32_t)(gy - gy0) * pitch + (uint32_t)gx * (uint32_t)sizeof(texel_t))));
    }
  }
  __syncthreads();
  // ---- phase 2: FsrEasuSetF's per-position terms, for the texels that are read as f/g/j/k of some pixel: columns
  //      1..fw-2, rows 1..fh-2 of the footprint (every neighbour of those lies inside it, so nothing is clamped). ----
  const int iw = fw - 2, m = iw * (fh - 2);
  const float inv_iw = __builtin_amdgcn_rcpf((float)iw);  // (1 ulp: see inv_fw above)
  const float* const lum = reinterpret_cast<const float*>(l.tex) + 3;  // luma of texel i at lum[4 * i]
  for (int j = tid; j < m; j += THREADS) {
    const int y = FW ? j / (FW - 2) : (int)(((float)j + 0.5f) * inv_iw);
    const int rs = PITCH ? kRow : fw;  // row stride of the layout
    const int i = (y + 1) * rs + (j - y * iw) + 1;
    if constexpr (STRICT) {
      static_assert(!EXACT, "F-strict stages the default arithmetic's records");
      // whole records instead of five lumas four dwords apart: the same LDS cycles (the strided ds_read_b32 are served four ways), and
      // the colours come along for the window magnitude
      const float4_t tA = l.tex[i - rs], tB = l.tex[i - 1], tC = l.tex[i], tD = l.tex[i + 1], tE = l.tex[i + rs];
      float4_t an = easu_analysis<false>(tA.w, tB.w, tC.w, tD.w, tE.w);
      an.w = max3_asm(max3_asm(absmax3(tA), absmax3(tB), absmax3(tC)), absmax3(tD), absmax3(tE));
      l.ana[i] = an;
    } else {
      l.ana[i] = easu_analysis<EXACT>(lum[4 * (i - rs)], lum[4 * (i - 1)], lum[4 * i], lum[4 * (i + 1)], lum[4 * (i + rs)]);
    }
  }
  __syncthreads();
}

struct rgbf_t { float r, g, b; };

// The FsrEasuF filter for one output pixel (ffx_fsr1.h:381-437 without the dering clamp): sub-texel position (ppx, ppy)
// (:324-326 done by the caller), the 12-tap window through `tex(dx, dy)` -> (R, G, B, .) with (0, 0) the texel 'f', and
// the four analyses through `ana(k)`, k = 0..3 for f, g, j, k (easu_analysis).  Returns aC * rcp(aW).  Everything up to
// the `dirR < 1/32768` decision is evaluated in the reference's exact operation order: that decision (and floor() in
// the caller) are the filter's only discontinuities.
// The terms of the filter that depend on the sub-texel ROW position only, grouped so that a caller whose pixels share a row
// may evaluate them once (easu_filter(..., ppx, ppy) evaluates them in place).  Parking them in LDS once per tile row was
// measured in round 3 and not kept: no gain in the EASU kernel, -2 % / +15 % (1.5x / 1.3x) in the generic fused kernel
// (profiles/ab_r03/r3c7_row_terms_table_ab.log).
struct EasuRowTerms {
  float ppy, omy, oym, oy2;    // sub-texel position (:324-326), 1 - ppy, and the tap-row offsets -1 - ppy, 2 - ppy
  float sqm, sq0, sq1, sq2;    // squares of the four tap-row offsets oym, oy0 = 0 - ppy, oy1 = omy, oy2
  float oy0;
};
__device__ __forceinline__ EasuRowTerms easu_row_terms(float ppy) {
  EasuRowTerms y;
  y.ppy = ppy; y.omy = 1.0f - ppy; y.oym = -1.0f - ppy; y.oy0 = 0.0f - ppy; y.oy2 = 2.0f - ppy;
  y.sqm = y.oym * y.oym; y.sq0 = y.oy0 * y.oy0; y.sq1 = y.omy * y.omy; y.sq2 = y.oy2 * y.oy2;
  return y;
}

// The default arithmetic's per-pixel tap terms (see the derivation in easu_filter): Q-form of the rotated, scaled offset with
// the row parts folded in, and the window polynomial's coefficients in u = d2 / clp.
struct EasuTapTerms { float q00, sm, s0, s1, s2, bm, b0, b1, b2, k1, k2, k3; };
__device__ __forceinline__ EasuTapTerms easu_tap_terms(float dirx, float diry, float len2x, float len2y, float lob, float clp, const EasuRowTerms& yt) {
  const float rclp = __builtin_amdgcn_rcpf(clp);
  const float sx = len2x * len2x * rclp, sy = len2y * len2y * rclp;
  const float dxx = dirx * dirx, dyy = diry * diry, dxy2 = 2.0f * (dirx * diry);
  const float q00 = fmaf(dxx, sx, dyy * sy), q11 = fmaf(dyy, sx, dxx * sy), q01 = dxy2 * (sx - sy);
  EasuTapTerms h;
  h.q00 = q00;
  h.sm = q01 * yt.oym; h.s0 = q01 * yt.oy0; h.s1 = q01 * yt.omy; h.s2 = q01 * yt.oy2;
  h.bm = q11 * yt.sqm; h.b0 = q11 * yt.sq0; h.b1 = q11 * yt.sq1; h.b2 = q11 * yt.sq2;
  h.k2 = 0.25f * clp * clp; h.k1 = -1.25f * clp; h.k3 = lob * clp;
  return h;
}
__device__ __forceinline__ float easu_tap_weight(const EasuTapTerms& h, float ox, float s, float b) {
  const float u = sat(fmaf(ox, fmaf(h.q00, ox, s), b));
  const float base = fmaf(fmaf(h.k2, u, h.k1), u, 1.0f);
  const float wa = fmaf(h.k3, u, -1.0f);
  return base * (wa * wa);
}

template <bool EXACT, class Tex, class Ana>
__device__ __forceinline__ rgbf_t easu_filter(const Tex& tex, const Ana& ana, float ppx, const EasuRowTerms& yt);

template <bool EXACT, class Tex, class Ana>
__device__ __forceinline__ rgbf_t easu_filter(const Tex& tex, const Ana& ana, float ppx, float ppy) {
  return easu_filter<EXACT>(tex, ana, ppx, easu_row_terms(ppy));
}

template <bool EXACT, class Tex, class Ana>
__device__ __forceinline__ rgbf_t easu_filter(const Tex& tex, const Ana& ana, float ppx, const EasuRowTerms& yt) {
  const float ppy = yt.ppy;
  const float omx = 1.0f - ppx, omy = yt.omy;
  // :381-386 bilinear accumulation of the 4 analyses (f,g,j,k), reference order:
  //   dir.x += dirX*w ; len += lenX*w ; dir.y += dirY*w ; len += lenY*w   for s,t,u,v in turn.
  const float4_t af = ana(0), ag = ana(1), aj = ana(2), ak = ana(3);
  const float wS = omx * omy, wT = ppx * omy, wU = omx * ppy, wV = ppx * ppy;
  float dirx = af.x * wS;  // 0 + x is exact, so the first add of each chain is dropped
  float diry = af.y * wS;
  dirx += ag.x * wT; diry += ag.y * wT;
  dirx += aj.x * wU; diry += aj.y * wU;
  dirx += ak.x * wV; diry += ak.y * wV;
  float len = af.z * wS;
  if (EXACT) {
    len = af.w * wS + len;
    len = ag.z * wT + len; len = ag.w * wT + len;
    len = aj.z * wU + len; len = aj.w * wU + len;
    len = ak.z * wV + len; len = ak.w * wV + len;
  } else {  // .z already holds lenX^2 + lenY^2
    len = fmaf(ag.z, wT, len);
    len = fmaf(aj.z, wU, len);
    len = fmaf(ak.z, wV, len);
  }

  // :389-395 normalise; the zero test is the filter's only branch-like discontinuity
  const float dir2x = dirx * dirx, dir2y = diry * diry;
  float dirR = dir2x + dir2y;
  const bool zro = dirR < (1.0f / 32768.0f);
  dirR = zro ? 1.0f : APrxLoRsqF1(dirR);
  dirx = zro ? 1.0f : dirx;
  dirx *= dirR;
  diry *= dirR;
  // :397-409 kernel shape
  len = len * 0.5f;
  len *= len;
  const float stretch = mad<EXACT>(dirx, dirx, diry * diry) * APrxLoRcpF1(fmaxf(fabsf(dirx), fabsf(diry)));
  const float len2x = mad<EXACT>(stretch - 1.0f, len, 1.0f);
  const float len2y = mad<EXACT>(-0.5f, len, 1.0f);
  const float lob = mad<EXACT>((float)((1.0 / 4.0 - 0.04) - 0.5), len, 0.5f);
  const float clp = APrxLoRcpF1(lob);

  // :421-434 12 taps.  aC += c*w ; aW += w
  float aR = 0.f, aG = 0.f, aB = 0.f, aW = 0.f;
  const float oxm = -1.0f - ppx, ox0 = 0.0f - ppx, ox1 = 1.0f - ppx, ox2 = 2.0f - ppx;
  const float oym = yt.oym, oy0 = yt.oy0, oy1 = yt.omy, oy2 = yt.oy2;
  if (EXACT) {
    auto tap = [&](int dx, int dy, float offx, float offy) {
      const float4_t c = tex(dx, dy);
      float vx = (offx * dirx) + (offy * diry);
      float vy = (offx * (-diry)) + (offy * dirx);
      vx *= len2x;
      vy *= len2y;
      float d2 = vx * vx + vy * vy;
      d2 = fminf(d2, clp);
      float wB = (float)(2.0 / 5.0) * d2 + -1.0f;
      float wA = lob * d2 + -1.0f;
      wB *= wB;
      wA *= wA;
      wB = (float)(25.0 / 16.0) * wB + (float)(-(25.0 / 16.0 - 1.0));
      const float w = wB * wA;
      aR += c.x * w; aG += c.y * w; aB += c.z * w;
      aW += w;
    };
    // reference order: b c i j f e k l h g o n
    tap(0, -1, ox0, oym); tap(1, -1, ox1, oym); tap(-1, 1, oxm, oy1); tap(0, 1, ox0, oy1);
    tap(0, 0, ox0, oy0); tap(-1, 0, oxm, oy0); tap(1, 1, ox1, oy1); tap(2, 1, ox2, oy1);
    tap(2, 0, ox2, oy0); tap(1, 0, ox1, oy0); tap(1, 2, ox1, oy2); tap(0, 2, ox0, oy2);
  } else {
    // Re-formulated taps (continuous part of the filter; ~1e-6 relative from the reference order):
    //   v = M*off with M = [dir.x*len.x dir.y*len.x ; -dir.y*len.y dir.x*len.y]          (:250-253)
    //   u = min(|v|^2, clp)/clp = sat(off^T Q off),  Q = M^T M / clp  -> the clip is the fma's clamp bit
    //   off^T Q off = ox*(q00*ox + 2*q01*oy) + q11*oy^2      (per-row terms s = 2*q01*oy, b = q11*oy^2)
    //   base = 25/16*(2/5*d2-1)^2 - 9/16 = 1/4*d2^2 - 5/4*d2 + 1,  window = (lob*d2-1)^2,  d2 = clp*u
    const EasuTapTerms h = easu_tap_terms(dirx, diry, len2x, len2y, lob, clp, yt);
    const float sm = h.sm, s0 = h.s0, s1 = h.s1, s2 = h.s2, bm = h.bm, b0 = h.b0, b1 = h.b1, b2 = h.b2;
    auto weight = [&](float ox, float s, float b) { return easu_tap_weight(h, ox, s, b); };
    auto tap = [&](int dx, int dy, float ox, float s, float b) {
      const float4_t c = tex(dx, dy);
      const float w = weight(ox, s, b);
      aR = fmaf(c.x, w, aR); aG = fmaf(c.y, w, aG); aB = fmaf(c.z, w, aB);
      aW += w;
    };
    {  // the first tap starts the sums
      const float4_t c = tex(0, -1);
      aW = weight(ox0, sm, bm);
      aR = c.x * aW; aG = c.y * aW; aB = c.z * aW;
    }
    tap(1, -1, ox1, sm, bm);
    tap(-1, 0, oxm, s0, b0); tap(0, 0, ox0, s0, b0); tap(1, 0, ox1, s0, b0); tap(2, 0, ox2, s0, b0);
    tap(-1, 1, oxm, s1, b1); tap(0, 1, ox0, s1, b1); tap(1, 1, ox1, s1, b1); tap(2, 1, ox2, s1, b1);
    tap(0, 2, ox0, s2, b2); tap(1, 2, ox1, s2, b2);
  }
  // :437 normalise (dering clamp is applied by the caller)
  // (EXACT: the general IEEE division, not rcp_ieee — one reciprocal per pixel saves five instructions of eight hundred, and its
  //  never-taken branch in the middle of the filter costs the exact-2x kernel 4 %: profiles/ab_r03/r3c15_exact_rcp_ieee_ab.log)
  const float rW = EXACT ? 1.0f / aW : __builtin_amdgcn_rcpf(aW);
  // pinned in every variant: the narrowing that follows must round the binary32 product, not re-fuse it
  // (v_fma_mixlo_f16), or two kernels sharing this code could round the same pixel differently
  return rgbf_t{pinned(aR * rW), pinned(aG * rW), pinned(aB * rW)};
}

// The filter on a staged footprint: 'f' sits at footprint index f_idx.
template <bool EXACT>
__device__ __forceinline__ rgbf_t easu_pixel(const EasuLds& l, int f_idx, float ppx, float ppy) {
  const int fw = l.fw;
  return easu_filter<EXACT>([&](int dx, int dy) { return l.tex[f_idx + dy * fw + dx]; },
                            [&](int k) { return l.ana[f_idx + (k >> 1) * fw + (k & 1)]; }, ppx, ppy);
}

template <bool EXACT>
__device__ __forceinline__ rgbf_t easu_pixel(const EasuLds& l, int f_idx, float ppx, const EasuRowTerms& yt) {
  const int fw = l.fw;
  // based at the window's top-left texel (-1, -1): with a compile-time row pitch every tap is base + a non-negative
  // immediate offset (the DS offset field is unsigned: taps above / left of 'f' would otherwise need bases of their own)
  const float4_t* const w0 = l.tex + (f_idx - fw - 1);
  const float4_t* const a0 = w0 + (l.ana - l.tex);  // (a compile-time distance in the pitched layout)
  return easu_filter<EXACT>([&](int dx, int dy) { return w0[(dy + 1) * fw + (dx + 1)]; },
                            [&](int k) { return a0[((k >> 1) + 1) * fw + (k & 1) + 1]; }, ppx, yt);
}

// Dering bounds (:416-419): per-channel min and max of the 2x2 block f g / j k whose top-left texel is f_idx.
// v_min3_f32 / v_max3_f32 written out: the operands come straight from LDS, where the compiler no longer knows they are
// canonical and would spend a `v_max_f32 x, x, x` on each of the twelve before handing them to fminf / fmaxf; the
// instructions themselves are IEEE minNum / maxNum, which is the oracle's pin for the shading languages' min / max.
struct EasuBounds { float mnR, mnG, mnB, mxR, mxG, mxB; };

__device__ __forceinline__ float min4_asm(float a, float b, float c, float d) {
  float t;
  asm("v_min3_f32 %0, %1, %2, %3\n\tv_min_f32 %0, %0, %4" : "=&v"(t) : "v"(a), "v"(b), "v"(c), "v"(d));
  return t;
}
__device__ __forceinline__ float max4_asm(float a, float b, float c, float d) {
  float t;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(t) : "v"(a), "v"(b), "v"(c), "v"(d));
  return t;
}

__device__ __forceinline__ EasuBounds easu_bounds(float4_t cf, float4_t cg, float4_t cj, float4_t ck) {
  return EasuBounds{min4_asm(cf.x, cg.x, cj.x, ck.x), min4_asm(cf.y, cg.y, cj.y, ck.y), min4_asm(cf.z, cg.z, cj.z, ck.z),
                    max4_asm(cf.x, cg.x, cj.x, ck.x), max4_asm(cf.y, cg.y, cj.y, ck.y), max4_asm(cf.z, cg.z, cj.z, ck.z)};
}

__device__ __forceinline__ EasuBounds easu_bounds(const EasuLds& l, int f_idx) {
  const int fw = l.fw;
  const float4_t* const w0 = l.tex + (f_idx - fw - 1);  // the window's top-left texel, as in easu_pixel: one base for every read
  return easu_bounds(w0[fw + 1], w0[fw + 2], w0[2 * fw + 1], w0[2 * fw + 2]);
}

// The filter and the dering bounds from ONE pass over the window: f g j k are four of the twelve taps, so the bounds are
// taken of the tap values as they arrive instead of reading the four texels from LDS a second time (the generic kernels
// are short of LDS cycles, not of registers).  Same min / max instructions on the same values as easu_bounds(l, f_idx).
// strict_eps (STRICT staging): receives the F-strict threshold of the pixel (easu_strict_eps of its four analysis records).
template <bool EXACT>
__device__ __forceinline__ rgbf_t easu_pixel_with_bounds(const EasuLds& l, int f_idx, float ppx, const EasuRowTerms& yt, EasuBounds& m, float* strict_eps = nullptr) {
  const int fw = l.fw;
  const float4_t* const w0 = l.tex + (f_idx - fw - 1);
  const float4_t* const a0 = w0 + (l.ana - l.tex);
  float4_t cf = {}, cg = {}, cj = {}, ck = {};
  float mag = 0.0f;
  const rgbf_t p = easu_filter<EXACT>(
      [&](int dx, int dy) {
        const float4_t v = w0[(dy + 1) * fw + (dx + 1)];
        if (dy == 0 && dx == 0) cf = v;       // (dx, dy are literals at every call site: these fold away)
        else if (dy == 0 && dx == 1) cg = v;
        else if (dy == 1 && dx == 0) cj = v;
        else if (dy == 1 && dx == 1) ck = v;
        return v;
      },
      [&](int k) {
        const float4_t v = a0[((k >> 1) + 1) * fw + (k & 1) + 1];
        if (strict_eps) mag = k ? vmax_asm(mag, v.w) : v.w;  // (a null literal at the other call sites: folds away)
        return v;
      },
      ppx, yt);
  m = easu_bounds(cf, cg, cj, ck);
  if (strict_eps) *strict_eps = kEasuStrictScale * mag;
  return p;
}

// The exact-2x quad form: the four pixels of a quad share the texel 'f', hence the four analyses and the dering bounds.  `A` holds
// the analyses of f g j k, loaded once per quad by the caller; WITH_BOUNDS additionally takes the bounds of the tap values
// (one pixel of the quad does that, the others reuse them).
template <bool EXACT, bool WITH_BOUNDS>
__device__ __forceinline__ rgbf_t easu_quad_pixel(const EasuLds& l, int f_idx, float ppx, float ppy, const float4_t (&A)[4], EasuBounds& m) {
  const int fw = l.fw;
  const float4_t* const w0 = l.tex + (f_idx - fw - 1);
  float4_t cf = {}, cg = {}, cj = {}, ck = {};
  const rgbf_t p = easu_filter<EXACT>(
      [&](int dx, int dy) {
        const float4_t v = w0[(dy + 1) * fw + (dx + 1)];
        if (WITH_BOUNDS) {
          if (dy == 0 && dx == 0) cf = v;
          else if (dy == 0 && dx == 1) cg = v;
          else if (dy == 1 && dx == 0) cj = v;
          else if (dy == 1 && dx == 1) ck = v;
        }
        return v;
      },
      [&](int k) { return A[k]; }, ppx, ppy);
  if (WITH_BOUNDS) m = easu_bounds(cf, cg, cj, ck);
  return p;
}

// Everything of the default arithmetic's filter before the taps, for one pixel: bilinear analysis (:381-386, reference order),
// normalise and zero test (:389-395), kernel shape (:397-409), tap terms.  The same statements as in easu_filter<false>.
template <class Ana>
__device__ __forceinline__ EasuTapTerms easu_pixel_terms(const Ana& ana, float ppx, const EasuRowTerms& yt) {
  const float ppy = yt.ppy;
  const float omx = 1.0f - ppx, omy = yt.omy;
  const float4_t af = ana(0), ag = ana(1), aj = ana(2), ak = ana(3);
  const float wS = omx * omy, wT = ppx * omy, wU = omx * ppy, wV = ppx * ppy;
  float dirx = af.x * wS;
  float diry = af.y * wS;
  dirx += ag.x * wT; diry += ag.y * wT;
  dirx += aj.x * wU; diry += aj.y * wU;
  dirx += ak.x * wV; diry += ak.y * wV;
  float len = af.z * wS;
  len = fmaf(ag.z, wT, len);
  len = fmaf(aj.z, wU, len);
  len = fmaf(ak.z, wV, len);
  const float dir2x = dirx * dirx, dir2y = diry * diry;
  float dirR = dir2x + dir2y;
  const bool zro = dirR < (1.0f / 32768.0f);
  dirR = zro ? 1.0f : APrxLoRsqF1(dirR);
  dirx = zro ? 1.0f : dirx;
  dirx *= dirR;
  diry *= dirR;
  len = len * 0.5f;
  len *= len;
  const float stretch = mad<false>(dirx, dirx, diry * diry) * APrxLoRcpF1(fmaxf(fabsf(dirx), fabsf(diry)));
  const float len2x = mad<false>(stretch - 1.0f, len, 1.0f);
  const float len2y = mad<false>(-0.5f, len, 1.0f);
  const float lob = mad<false>((float)((1.0 / 4.0 - 0.04) - 0.5), len, 0.5f);
  const float clp = APrxLoRcpF1(lob);
  return easu_tap_terms(dirx, diry, len2x, len2y, lob, clp, yt);
}

// Two pixels that share their 12-tap window and analyses (the two pixels of a row of an exact-2x quad), filtered together:
// every texel is read from LDS once and used for both — the default arithmetic's statements, pixel by pixel the same values in the
// same order as easu_filter<false>, so the results are bit-identical to two single calls.  `tex` is called once per tap.
template <class Tex, class Ana>
__device__ __forceinline__ void easu_filter_pair(const Tex& tex, const Ana& ana, float ppxA, float ppxB, float ppy, rgbf_t& outA, rgbf_t& outB) {
  const EasuRowTerms yt = easu_row_terms(ppy);
  const EasuTapTerms hA = easu_pixel_terms(ana, ppxA, yt), hB = easu_pixel_terms(ana, ppxB, yt);
  const float oxA[4] = {-1.0f - ppxA, 0.0f - ppxA, 1.0f - ppxA, 2.0f - ppxA}, oxB[4] = {-1.0f - ppxB, 0.0f - ppxB, 1.0f - ppxB, 2.0f - ppxB};
  float aR = 0.f, aG = 0.f, aB = 0.f, aW = 0.f, bR = 0.f, bG = 0.f, bB = 0.f, bW = 0.f;
  auto tap = [&](int dx, int dy, float sA, float tA, float sB, float tB) {
    const float4_t c = tex(dx, dy);
    const float wA = easu_tap_weight(hA, oxA[dx + 1], sA, tA), wB = easu_tap_weight(hB, oxB[dx + 1], sB, tB);
    aR = fmaf(c.x, wA, aR); aG = fmaf(c.y, wA, aG); aB = fmaf(c.z, wA, aB); aW += wA;
    bR = fmaf(c.x, wB, bR); bG = fmaf(c.y, wB, bG); bB = fmaf(c.z, wB, bB); bW += wB;
  };
  {  // the first tap starts the sums
    const float4_t c = tex(0, -1);
    aW = easu_tap_weight(hA, oxA[1], hA.sm, hA.bm);
    bW = easu_tap_weight(hB, oxB[1], hB.sm, hB.bm);
    aR = c.x * aW; aG = c.y * aW; aB = c.z * aW;
    bR = c.x * bW; bG = c.y * bW; bB = c.z * bW;
  }
  tap(1, -1, hA.sm, hA.bm, hB.sm, hB.bm);
  tap(-1, 0, hA.s0, hA.b0, hB.s0, hB.b0); tap(0, 0, hA.s0, hA.b0, hB.s0, hB.b0); tap(1, 0, hA.s0, hA.b0, hB.s0, hB.b0); tap(2, 0, hA.s0, hA.b0, hB.s0, hB.b0);
  tap(-1, 1, hA.s1, hA.b1, hB.s1, hB.b1); tap(0, 1, hA.s1, hA.b1, hB.s1, hB.b1); tap(1, 1, hA.s1, hA.b1, hB.s1, hB.b1); tap(2, 1, hA.s1, hA.b1, hB.s1, hB.b1);
  tap(0, 2, hA.s2, hA.b2, hB.s2, hB.b2); tap(1, 2, hA.s2, hA.b2, hB.s2, hB.b2);
  const float rA = __builtin_amdgcn_rcpf(aW), rB = __builtin_amdgcn_rcpf(bW);
  outA = rgbf_t{pinned(aR * rA), pinned(aG * rA), pinned(aB * rA)};
  outB = rgbf_t{pinned(bR * rB), pinned(bG * rB), pinned(bB * rB)};
}

// A row of an exact-2x quad (sub-texel columns 1/4 and 3/4) on a staged footprint; WITH_BOUNDS takes the quad's dering bounds
// of the tap values on the way.
template <bool WITH_BOUNDS>
__device__ __forceinline__ void easu_quad_row(const EasuLds& l, int f_idx, float ppy, const float4_t (&A)[4], EasuBounds& m, rgbf_t& p0, rgbf_t& p1) {
  const int fw = l.fw;
  const float4_t* const w0 = l.tex + (f_idx - fw - 1);
  float4_t cf = {}, cg = {}, cj = {}, ck = {};
  easu_filter_pair(
      [&](int dx, int dy) {
        const float4_t v = w0[(dy + 1) * fw + (dx + 1)];
        if (WITH_BOUNDS) {
          if (dy == 0 && dx == 0) cf = v;
          else if (dy == 0 && dx == 1) cg = v;
          else if (dy == 1 && dx == 0) cj = v;
          else if (dy == 1 && dx == 1) ck = v;
        }
        return v;
      },
      [&](int k) { return A[k]; }, 0.25f, 0.75f, ppy, p0, p1);
  if (WITH_BOUNDS) m = easu_bounds(cf, cg, cj, ck);
}

// Dering clamp in binary32 (:437 `min(max4, max(min4, pix))`) + optional `c *= c` (FSR_Pass.hlsl:78-79): the filter's
// result before the store conversion.  Clamping before or after the store's rounding gives the same stored value: rounding
// is monotone and the bounds are values of the storage format.  EXACT keeps the reference's two operations; the default
// arithmetic takes v_med3_f32, which returns min3 of its operands when one is a NaN — what max-then-min returns too
// when `pix` is the NaN (a window whose weights sum to zero).
template <bool EXACT>
__device__ __forceinline__ rgbf_t easu_clamp(const EasuBounds& m, rgbf_t p, bool hdr_square) {
  float pr, pg, pb;
  if (EXACT) {
    pr = fminf(m.mxR, fmaxf(m.mnR, p.r)); pg = fminf(m.mxG, fmaxf(m.mnG, p.g)); pb = fminf(m.mxB, fmaxf(m.mnB, p.b));
  } else {
    pr = __builtin_amdgcn_fmed3f(p.r, m.mnR, m.mxR); pg = __builtin_amdgcn_fmed3f(p.g, m.mnG, m.mxG); pb = __builtin_amdgcn_fmed3f(p.b, m.mnB, m.mxB);
  }
  if (hdr_square) { pr *= pr; pg *= pg; pb *= pb; }
  return rgbf_t{pinned(pr), pinned(pg), pinned(pb)};
}

// Dering clamp + alpha = 1 (FSR_Pass.hlsl:80), producing the pixel in its storage format.
template <int FMT, bool EXACT>
__device__ __forceinline__ typename Pixel<FMT>::T easu_resolve(const EasuBounds& m, rgbf_t p, bool hdr_square) {
  const rgbf_t q = easu_clamp<EXACT>(m, p, hdr_square);
  return Pixel<FMT>::store(q.r, q.g, q.b, 1.0f);
}

// ------------------------------------------------------------------------------------------------------------------------------
// F-strict (FSR1_FLAG_MATH_STRICT): the default arithmetic's speed with FsrEasuF's bits.
//
// The default arithmetic's binary32 result differs from the reference order's (EXACT) by a few binary32 ULPs of the WINDOW's
// magnitude M = the largest |R|, |G|, |B| among the pixel's 12 taps (relative to the VALUE the difference is unbounded: a dark pixel
// next to bright texels; the window's contrast, with or without a term in |x|, is no tighter a scale: profiles/ab_r06/
// r06_contrast_scale.json).  Measured, d = |default - EXACT| / (2^-24 M), tools/experiments_r06/strict_stress.py — 6.3e11 values of
// uniform / smooth / blocky / hard-edged / gradient / dark / HDR log-normal / text-like / natural content at ratios 1.25x .. 3x
// (profiles/ab_r06/r06_strict_stress.json): P(d > 4) 9e-3, P(d > 8) 4e-5, P(d > 16) 1.5e-9, max 25.0 (natural content: 17.6); the
// first 1.2e9 values had shown 14.6, a 1000-second run over 2.7e12 values (r06_strict_stress_long.json) 30.0 with none of them beyond 32.
// The tail thins by more than four decades from d > 8 to d > 16 and by at least three more to d > 32 (two generator seeds: 8.2e12 values, max 30.0).
// An ADVERSARY instead of random content (tools/experiments_r06/strict_adversarial.py: an evolution strategy over ~14 000 input tiles per
// ratio, twelve ratios from 0.75x to 4x, 2.3e11 values, profiles/ab_r06/r06_strict_adversarial*.json) lifts the typical d five- to tenfold
// (windows of HDR dynamic range, diagonal structure) and found max 37.2 (3x) and 35.6 (1.9x) once each — 29.2 / 25.4 when the same ratios were
// searched again, twice as long, from another seed; 18.5 .. 33.0 at the other ratios: d is rounding noise on top of a regime, which a search
// can choose but not climb.  A measured bound, not a proof (first order, twelve aligned weight errors would allow ~180): so the threshold is
//     e = kEasuStrictK * 2^-24 * M,   kEasuStrictK = 56     (1.5 x the largest d any search has produced, 1.9 x random content's; until
//                                                             the adversary's 37.2 it was 48; 32 / 48 / 64 cost EASU -1 % / 0 / +3 %, r6c9_strict_k.log)
// and the stored value of the default arithmetic is the stored value of FsrEasuF whenever the store conversion maps [x - e, x + e] to ONE
// code (rounding is monotone, and the dering clamp — applied to both — only ever moves a value onto a bound both share).  Pixels for
// which it does not (4-6 % of natural content at RGBA16F) are queued in LDS by the workgroup and re-evaluated in the reference's
// operation order by the first lanes of the workgroup, densely (easu_strict_pixel), before the tile's footprint leaves the LDS.
// The conversion is the format's own (Pixel<FMT>::store), so the test is exact for UNORM storage too.
// ------------------------------------------------------------------------------------------------------------------------------

template <class T> __device__ __forceinline__ bool texel_bits_differ(const T& a, const T& b);
template <> __device__ __forceinline__ bool texel_bits_differ<half4_t>(const half4_t& a, const half4_t& b) {
  return __builtin_bit_cast(unsigned long long, a) != __builtin_bit_cast(unsigned long long, b);
}
template <> __device__ __forceinline__ bool texel_bits_differ<uint32_t>(const uint32_t& a, const uint32_t& b) { return a != b; }
template <> __device__ __forceinline__ bool texel_bits_differ<float4_t>(const float4_t& a, const float4_t& b) {
  return as_u32(a.x) != as_u32(b.x) || as_u32(a.y) != as_u32(b.y) || as_u32(a.z) != as_u32(b.z);
}

// The window magnitude of the pixel whose texels f g j k have the analysis records a0 .. a3 (STRICT staging), times the threshold.
__device__ __forceinline__ float easu_strict_eps(const float4_t& af, const float4_t& ag, const float4_t& aj, const float4_t& ak) {
  return kEasuStrictScale * vmax_asm(max3_asm(af.w, ag.w, aj.w), ak.w);
}

// Per-channel thresholds of a pixel (of a quad: the four pixels share their bounds): e, but no more than the width of the dering
// interval — both arithmetics' results lie in [mn, mx], so they cannot differ by more than mx - mn whatever the window's magnitude.
// What it buys: a channel that is CONSTANT over f g j k (mn == mx: a black or saturated channel of flat colour, the zero channels of
// pure primaries) passes the test — both results are that constant — where e alone, scaled by the other channels' magnitude, spans
// dozens of binary16 steps around zero and would send every such pixel through the re-evaluation.
struct EasuStrictEps { float r, g, b; };
__device__ __forceinline__ EasuStrictEps easu_strict_eps_rgb(const EasuBounds& m, float e) {
  return EasuStrictEps{vmin_asm(e, m.mxR - m.mnR), vmin_asm(e, m.mxG - m.mnG), vmin_asm(e, m.mxB - m.mnB)};
}

// Dering clamp + store conversion of the default arithmetic's pixel `p`, and the test: true when [x - e, x + e] does not convert to
// one code in every channel, i.e. the pixel has to be re-evaluated in the reference's operation order.  `out` is the code of x - e:
// the code of x + e as well when the test passes, overwritten otherwise.
template <int FMT>
__device__ __forceinline__ bool easu_strict_resolve(const EasuBounds& m, rgbf_t p, const EasuStrictEps& e, typename Pixel<FMT>::T& out) {
  const rgbf_t q = easu_clamp<false>(m, p, false);
  out = Pixel<FMT>::store(q.r - e.r, q.g - e.g, q.b - e.b, 1.0f);
  const typename Pixel<FMT>::T hi = Pixel<FMT>::store(q.r + e.r, q.g + e.g, q.b + e.b, 1.0f);
  return texel_bits_differ<typename Pixel<FMT>::T>(out, hi);
}

// FsrEasuF in the reference's operation order for ONE pixel of a footprint staged for the default arithmetic (STRICT staging): the
// four analyses are re-derived from the lumas of the 12 taps (the tex records' fourth component) exactly as phase 2 of the EXACT
// kernels derives them — same function, same operands — so the result is bit-identical to easu_kernel<.., EXACT = true>'s.
template <int FMT>
__device__ __forceinline__ typename Pixel<FMT>::T easu_strict_pixel(const EasuLds& l, int f_idx, float ppx, float ppy) {
  const int fw = l.fw;
  const float4_t* const w0 = l.tex + (f_idx - fw - 1);
  const float* const lum = reinterpret_cast<const float*>(w0) + 3;  // luma of window texel (dx, dy) at lum[4 * ((dy + 1) * fw + dx + 1)]
  // (the bounds first, from reads of their own: taken of the tap values as they arrive they would hold sixteen registers through the
  //  whole filter, and this function runs beside the callers' loop state at their seven-wave budgets)
  const EasuBounds m = easu_bounds(l, f_idx);
  const rgbf_t p = easu_filter<true>(
      [&](int dx, int dy) { return w0[(dy + 1) * fw + (dx + 1)]; },
      [&](int k) {
        const int c = ((k >> 1) + 1) * fw + (k & 1) + 1;  // window index of f / g / j / k
        return easu_analysis<true>(lum[4 * (c - fw)], lum[4 * (c - 1)], lum[4 * c], lum[4 * (c + 1)], lum[4 * (c + fw)]);
      },
      ppx, ppy);
  return easu_resolve<FMT, true>(m, p, false);
}

// The workgroup's queue of pixels to re-evaluate: 16-bit pixel ids (the kernel's own numbering of its tile), one SEGMENT per wave.  A
// wave fills its own segment with ballots and lane counts (v_mbcnt) — no atomic, no LDS round trip, nothing serial — once, after its pixel
// loop; the re-evaluation then walks the segments as one dense list.  (Measured on the way, profiles/ab_r06: a shared queue behind an LDS
// atomic costs the 64 x 16 exact-2x kernel 2.8 us of 43.5 — the compiler combines the lanes' increments in a scalar loop over the pushing
// lanes, at the very end of each wave's life; a queue of lane groups expanded before the re-evaluation puts a serial step and a barrier on
// the workgroup's critical path and is slower still.)  The queue holds an EIGHTH of the tile's pixels, at least 256 (natural content
// queues 2-4 %): LDS per workgroup decides how many workgroups a CU holds, and a queue for every pixel of a 64 x 32 tile (4 KB) costs the
// generic kernel one of its four (1440p -> 4K: +7 % for the queue alone, r6c1_strict_breakdown.log).  What finds no room stays with its
// lane and is pushed again in the next round (easu_strict_rounds): content that queues every pixel runs tile_pixels / capacity rounds of
// fully occupied re-evaluation passes.
struct EasuStrictQueue {
  uint32_t* count;      // [waves] pushes of each wave in this round, may exceed the segment
  unsigned short* ids;  // [waves][capacity / waves]
};
__host__ __device__ constexpr int easu_strict_queue_capacity(int tile_pixels) { return tile_pixels / 8 < 256 ? 256 : tile_pixels / 8; }
// bytes of LDS behind the footprint region for a tile of `tile_pixels` pixels
__host__ __device__ constexpr size_t easu_strict_queue_bytes(size_t tile_pixels) { return 32 + (((size_t)easu_strict_queue_capacity((int)tile_pixels) * 2 + 15) & ~(size_t)15); }
__device__ __forceinline__ EasuStrictQueue easu_strict_queue_carve(char* p) {
  return EasuStrictQueue{reinterpret_cast<uint32_t*>(p), reinterpret_cast<unsigned short*>(p + 32)};
}
// One wave appends the pixels whose bits are set in its lanes' `mask` (bit b < NBITS = pixel id_of(b)) to its segment; returns, per lane,
// the bits that found no room.  Every lane of the wave takes part.
template <int NBITS, class IdOf>
__device__ __forceinline__ uint32_t easu_strict_push(const EasuStrictQueue& q, uint32_t mask, const IdOf& id_of, int wave, int segment) {
  unsigned short* const seg = q.ids + wave * segment;
  uint32_t used = 0;  // wave-uniform
#pragma unroll
  for (int b = 0; b < NBITS; ++b) {
    const bool mine = (mask >> b) & 1u;
    const unsigned long long vote = __ballot(mine);
    if (vote) {
      const uint32_t at = used + __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      if (mine && at < (uint32_t)segment) {
        seg[at] = (unsigned short)id_of(b);
        mask &= ~(1u << b);
      }
      used += (uint32_t)__builtin_popcountll(vote);
    }
  }
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) q.count[wave] = used;
  return mask;
}
// The re-evaluation rounds.  Every thread of the workgroup calls this once after its pixel loop with the pixels it collected
// (`mask`, bit b < NBITS = pixel id_of(b)); `redo(id)` re-evaluates and stores pixel `id`.  The first barrier inside also orders the
// caller's pixel loop before the re-evaluation; none follows the last round.
template <int THREADS, int NBITS, class IdOf, class Redo>
__device__ __forceinline__ void easu_strict_rounds(const EasuStrictQueue& q, uint32_t mask, const IdOf& id_of, int capacity, int tid, const Redo& redo) {
  constexpr int kWaves = THREADS / 64;
  const int segment = capacity / kWaves;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (;;) {
    mask = easu_strict_push<NBITS>(q, mask, id_of, wave, segment);
    __syncthreads();
    int held[kWaves], n = 0;
    bool more = false;  // (workgroup-uniform)
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const int c = __builtin_amdgcn_readfirstlane((int)q.count[w]);  // (uniform: keep the table in scalar registers)
      more |= c > segment;
      held[w] = c < segment ? c : segment;
      n += held[w];
    }
    for (int i = tid; i < n; i += THREADS) {
      int j = i, at = 0;
#pragma unroll
      for (int w = 0; w < kWaves - 1; ++w) {
        const bool beyond = j >= held[w];
        j -= beyond ? held[w] : 0;
        at += beyond ? segment : 0;
        if (!beyond) break;
      }
      redo((int)q.ids[at + j]);
    }
    if (!more) break;
    __syncthreads();  // everyone has read the counts and its ids
  }
}

}  // namespace fsr1
