// EASU -> RCAS in one launch, exact-2x variant (con0 = {1/2, 1/2, -1/4, -1/4}: 1080p -> 4K, 4K -> 8K, 540p -> 1080p).
//
// The generic fused kernel (fsr1_fused_kernel.h) runs EASU one pixel at a time on its apron tile; at exactly 2x the EASU
// kernel's quad form applies (fsr1_easu_kernel.h, S2): output pixels 2j+1 and 2j+2 share the texel f = j with sub-texel
// positions 1/4 and 3/4, so a 2x2 quad shares window address, analyses and dering bounds, and every position constant
// is a literal.  A workgroup owns a 62-pixel column of the output and walks DOWN it in steps of 16 EASU rows (round 3; before,
// every 16-row apron tile gave 14 output rows and the two apron rows were computed twice):
//   step s of run ty   EASU rows [Y0 - 1 + 16 s, Y0 + 14 + 16 s], Y0 = (16 S - 2) ty — 32 x 8 quads, one per lane, from an odd
//                      (= quad-aligned) origin, written to an LDS ring of 18 rows: the 16 new ones and the last two of step s - 1
//   footprint          35 x 11 texels from (31 tx - 2, Y0 / 2 + 8 s - 2)
//   RCAS               the 16 rows whose three EASU rows are in the ring (14 in step 0): four per wave
// so a run of S steps computes 16 S EASU rows for 16 S - 2 output rows: the vertical apron is paid once per run instead of once
// per tile (S = 1 is the old tile).  S comes from the host (FusedArgs::run_steps): as many steps as keep every CU supplied.
// Phase 4 (RCAS) has lane L own apron column L: lanes 1 .. 62 are the tile's output columns and lanes 0 / 63 the apron
// columns themselves, so both horizontal neighbours of every output pixel arrive by DPP wave shifts and nothing but the
// lane's own column is read from LDS.  Same per-pixel functions on the same values as the generic kernel and as the two
// dispatches: bit-identical output (tests/test_gpu_parity.py, test_gpu_fullframe.py).  Row bands with an even origin
// (tests/test_gpu_bands.py) run this kernel too.
#include "fsr1_device.h"
#include "fsr1_device_easu.hpp"
#include "fsr1_device_rcas.hpp"
#include "fsr1_overrides.h"

namespace fsr1 {

// (tile / step / ring geometry: kFs2* in fsr1_device.h, shared with the packed-fp16 twin fsr1_fused_s2_h.hip)

// `waves`: 4 = the 256-thread workgroup (16 EASU rows per step), 8 = the 512-thread one (32 rows per step: the one-step launch of
// a single large frame, see fused_s2_tall_tiles)
size_t fused_s2_lds_bytes(int fmt, int waves) {
  const size_t texel = fmt == FSR1_FORMAT_RGBA32F ? 16 : (fmt == FSR1_FORMAT_RGBA16F ? 8 : 4);
  return easu_lds_region_bytes((size_t)kFs2FpW * (2 * waves + 3)) + (size_t)kFs2MidW * (4 * waves + 2) * texel;
}
// F-strict (FSR1_FLAG_MATH_STRICT): + the queue of a step's EASU pixels behind the ring (fsr1_device_easu.hpp)
size_t fused_s2_strict_lds_bytes(int fmt, int waves) { return fused_s2_lds_bytes(fmt, waves) + easu_strict_queue_bytes((size_t)kFs2MidW * 4 * waves); }

// Steps per run: one workgroup per (column, run).  With S steps a run wastes 2 of 16 S EASU rows, so S wants to be large; but a
// launch wants every CU supplied to its end, and a run is S times as long as a tile: measured (profiles/ab_r03/r3c17..r3c19), a
// single 4K frame (9 610 one-step tiles, five residencies of 256 CUs x 7 workgroups) loses with any S > 1 (2: +3 %, 3: +8 %,
// 5: +18 %), four 4K frames or one 8K frame (38 440) gain 4.5 % at S = 4 (2: 1-2 %, 6: 2 %, 8: -1 %), sixteen 8K frames
// (613 000) gain 6.3 % at S = 8 (6: 5.7 %, 10: 5.9 %, 16: 4.5 %) — so: about five residencies of runs, at most 8 steps.
constexpr int kFs2MaxSteps = 8;
// (override_fused_s2_steps, csrc/fsr1_overrides.h: 0 in the product library; libfsr1_hip_test.so can force a number of steps per run —
//  any number gives the same image, tests/test_gpu_parity.py::test_fused_exact_2x_run_steps)
int fused_s2_run_steps(int width, int height, int frames, int cus, int wgs_per_cu, bool overlapped, bool strict) {
  if (const int forced = override_fused_s2_steps(); forced > 0) return forced;
  const long long tiles1 = (long long)((width + kFs2OutW - 1) / kFs2OutW) * ((height + kFs2Step - 3) / (kFs2Step - 2)) * frames;
  // (cus: the device's compute units — 256 on MI355X, where the rule was measured; wgs_per_cu: what the kernel's LDS admits, 7 for
  //  the F kernel, 5 for the packed-fp16 one)
  const long long slots = (long long)(cus > 0 ? cus : 256) * (wgs_per_cu > 0 ? wgs_per_cu : 7);
  // overlapped (FSR1_FLAG_FRAMES_OVERLAP: other frames run beside this launch on other streams, fsr1_pipeline): the tail of a launch
  // is filled by its neighbour's head, so the launch no longer has to keep every CU supplied to its own end and runs as long as
  // the apron saving asks for — one 4K frame walks 4 steps (profiles/ab_r04/r4c4_two_stream_walk.log, two streams, us per frame:
  // one-step 58.2-58.6, tall tile 56.6, S = 2 / 3 / 4 / 6 / 8: 55.8 / 54.8-55.1 / 53.9-54.2 / 54.2-54.3 / 55.9; 1440p output:
  // 26.3-26.6 one-step, 25.2-25.3 / 25.3-25.4 / 25.7 / 28.5 at 2 / 3 / 4 / 6): about 1.25 residencies of runs.
  const long long s = overlapped ? (4 * tiles1 + 5 * slots / 2) / (5 * slots) : tiles1 / (5 * slots);
  // (F-strict: every step ends with the serial re-evaluation of its queued pixels — at most two steps per run, fsr1_api.hip)
  const long long most = strict ? 2 : kFs2MaxSteps;
  return (int)(s < 1 ? 1 : s > most ? most : s);
}

// One-step launches of a frame that fills the chip take the TALL tile: a 512-thread workgroup (8 waves) whose single step is 32 EASU
// rows for 30 output rows — the vertical apron 2 of 32 rows instead of 2 of 16 (EASU work -6.7 %) with no serial steps, which is what
// makes walking lose on one frame (profiles/ab_r04/r4c1_fused_trace.log: a launch's first residency runs its steps in lock-step and
// its tail is as long as a run).  Three workgroups of eight waves per CU (38.7 KB of LDS each).  Small frames keep the 256-thread
// tile: twice as many workgroups to spread over the CUs.
bool fused_s2_tall_tiles(int width, int height, int frames, int steps, int cus, int fmt) {
  if (steps != 1 || fmt == FSR1_FORMAT_RGBA32F) return false;
  if (const int forced = override_fused_s2_tall(); forced >= 0) return forced != 0;  // (-1 in the product library)
  const long long tall = (long long)((width + kFs2OutW - 1) / kFs2OutW) * ((height + 29) / 30) * frames;
  return tall >= 4ll * 3 * (cus > 0 ? cus : 256);  // at least four residencies of tall tiles
}

void fused_s2_geometry(int width, int height, int steps, int* tiles_x, int* tiles_y, int step_rows) {
  const int run = step_rows * steps - 2;
  *tiles_x = (width + kFs2OutW - 1) / kFs2OutW;
  *tiles_y = (height + run - 1) / run;
}

// RUN = false: the one-step launch (run_steps == 1), compiled without the step loop and the ring arithmetic.
// WAVES: 4 (256 threads, 32 x 8 quads, 16 EASU rows per step) or 8 (512 threads, 32 x 16 quads, 32 rows per step).
// (register budget: <= 72 VGPRs for seven 4-wave workgroups per CU, which is what their LDS admits; the 8-wave workgroup's LDS admits
//  three per CU = six waves per SIMD: <= 80)
// STRICT (F-strict, default arithmetic only): the step's EASU pixels are tested against the store conversion's rounding boundaries and the
// ones that fail re-evaluated in the reference's operation order INTO THE RING before the RCAS phase reads it — the EASU half is
// bit-identical to EXACT's, the RCAS half runs the default arithmetic (include/fsr1_hip.h, FSR1_FLAG_MATH_STRICT).
template <int FMT, bool EXACT, bool RUN, int WAVES, bool STRICT = false>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 4 && !(STRICT && !RUN) ? 7 : 6, 8))) fused_s2_kernel(const FusedArgs a) {
  typedef typename Pixel<FMT>::T texel_t;
  static_assert(WAVES == 4 || WAVES == 8, "a wave filters two quad rows = four EASU rows of a step");
  constexpr int kFs2FpH = 2 * WAVES + 3, kFs2Step = 4 * WAVES, kFs2Ring = kFs2Step + 2;  // footprint rows, EASU rows per step, ring rows (shadow the 4-wave constants)
  constexpr int kThreads = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EasuLds l = easu_lds_carve(smem, kFs2FpW * kFs2FpH);
  texel_t* const mid = reinterpret_cast<texel_t*>(smem + easu_lds_region_bytes(kFs2FpW * kFs2FpH));  // [kFs2Ring][64], a ring of rows
  static_assert(!STRICT || !EXACT, "F-strict runs the default arithmetic");
  const EasuStrictQueue sq = easu_strict_queue_carve(reinterpret_cast<char*>(mid + kFs2Ring * kFs2MidW));  // (STRICT)

  const int tiles_per_frame = a.tiles_x * a.tiles_y;
  const int t = xcd_swizzle(blockIdx.x, tiles_per_frame * a.frames, a.xcd_shift);
  const int frame = t / tiles_per_frame;
  const int tf = t - frame * tiles_per_frame;
  const int ty = tf / a.tiles_x, tx = tf - ty * a.tiles_x;
  const int W = a.out.width, H = a.out.height;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  l.fw = kFs2FpW;
  // (a band of a larger output image, fsr1_easu_rcas_fused_dispatch_band: output row 0 is row a.origin_y — even, the host
  //  checks — of the image the constants describe, and the rows just above / below the band exist there when rows_above /
  //  rows_below say so: the apron computes them instead of taking them as outside the image)
  const int ylo = -a.rows_above, yhi = H - 1 + a.rows_below;
  const int steps = RUN ? a.run_steps : 1;
  const int Y0 = (kFs2Step * steps - 2) * ty;  // the run's first output row (even)
  const int ax0 = kFs2OutW * tx - 1;           // odd, like every step's first EASU row
  const float sharp = as_f32(a.rcas_con[0]);
  const uint32_t flags = a.flags;
  const bool stream = (flags & FSR1_FLAG_OUTPUT_STREAMING) != 0;
  char* const out_frame = a.out.base + (long long)frame * a.out.frame_stride;
  auto rgb = [](const texel_t& p) { const float4_t c = Pixel<FMT>::load(p); return rgb_t{c.x, c.y, c.z}; };
  typedef typename TexelPair<FMT>::T pair_t;

  const char* const in_frame = a.in.base + (long long)frame * a.in.frame_stride;

  // Ring row j of step s (j = 0, 1: the last two EASU rows of step s - 1; j = 2 .. 17: this step's) is EASU row ay0 + j - 2 and
  // sits in ring slot (base + j) mod 18, base = 16 s mod 18.
  int base = 0;
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int ay0 = Y0 - 1 + kFs2Step * s;  // first EASU row of this step
    // (step 0 is inside the image by the launch's geometry; a later step runs only if the previous one found it inside)
    int tid = threadIdx.x;
    if (RUN) asm volatile("" : "+v"(tid));  // per-lane addresses are re-derived in every step rather than kept in registers across the filter
    const int lane = tid & 63;
    easu_stage_footprint<FMT, false, EXACT, kFs2FpW, kFs2FpH, kThreads, 0, STRICT>(l, a.in, in_frame, 31 * tx - 2, ((ay0 + 1) >> 1) - 2 + (a.origin_y >> 1),
                                                                                  kFs2FpW, kFs2FpH, tid);
    // (its two barriers also separate this step's ring writes from the previous step's RCAS reads)

    // ---- phase 3: EASU on the step's 64 x 16 pixels, a quad per lane, rounded to the storage format (EASU runs with
    //      Sample.x = 0 when RCAS follows: FSR_Filter.cpp:107); pixels outside the image are 0 (FSR_Pass.hlsl:45,61) ----
    {
      const int qx = lane & 31, qy = 2 * wave + (lane >> 5);
      const int oxa = ax0 + 2 * qx, oya = ay0 + 2 * qy;
      const bool xin0 = oxa >= 0 && oxa < W, xin1 = oxa + 1 < W, yin0 = oya >= ylo && oya <= yhi, yin1 = oya + 1 <= yhi;
      const int f_idx = (qy + 1) * kFs2FpW + (qx + 1);
      int slot = base + 2 + 2 * qy;  // even, so the quad's two rows never straddle the wrap
      slot -= slot >= kFs2Ring ? kFs2Ring : 0;
      texel_t* const m0 = mid + slot * kFs2MidW + 2 * qx;
      if constexpr (STRICT) {
        // F-strict: the default arithmetic's row pairs + the rounding-boundary test; the pixels that fail (and the in-image pixels of
        // the image's border quads) are re-evaluated into the ring below.  Pixel id = 4 * thread + (x + 2 y of the quad)
        uint32_t redo = 0;
        if (xin0 && xin1 && yin0 && yin1) {
          const float4_t A[4] = {l.ana[f_idx], l.ana[f_idx + 1], l.ana[f_idx + kFs2FpW], l.ana[f_idx + kFs2FpW + 1]};
          const float e1 = easu_strict_eps(A[0], A[1], A[2], A[3]);
          EasuBounds m;
          rgbf_t q00, q10, q01, q11;
          texel_t p00, p10, p01, p11;
          easu_quad_row<true>(l, f_idx, 0.25f, A, m, q00, q10);
          const EasuStrictEps e = easu_strict_eps_rgb(m, e1);
          redo = easu_strict_resolve<FMT>(m, q00, e, p00) ? 1u : 0u;
          redo |= easu_strict_resolve<FMT>(m, q10, e, p10) ? 2u : 0u;
          *reinterpret_cast<pair_t*>(m0) = TexelPair<FMT>::make(p00, p10);
          easu_quad_row<false>(l, f_idx, 0.75f, A, m, q01, q11);
          redo |= easu_strict_resolve<FMT>(m, q01, e, p01) ? 4u : 0u;
          redo |= easu_strict_resolve<FMT>(m, q11, e, p11) ? 8u : 0u;
          *reinterpret_cast<pair_t*>(m0 + kFs2MidW) = TexelPair<FMT>::make(p01, p11);
        } else {
          m0[0] = m0[1] = m0[kFs2MidW] = m0[kFs2MidW + 1] = Pixel<FMT>::zero();  // outside the image: 0 (FSR_Pass.hlsl:45,61)
          redo = (xin0 && yin0 ? 1u : 0u) | (xin1 && yin0 ? 2u : 0u) | (xin0 && yin1 ? 4u : 0u) | (xin1 && yin1 ? 8u : 0u);
        }
        easu_strict_rounds<kThreads, 4>(sq, redo, [&](int b) { return 4 * tid + b; }, easu_strict_queue_capacity(kFs2MidW * kFs2Step), tid, [&](int id) {
          const int sub = id & 3, t = id >> 2, rqx = t & 31, rqy = 2 * (t >> 6) + ((t >> 5) & 1);
          int rslot = base + 2 + 2 * rqy;
          rslot -= rslot >= kFs2Ring ? kFs2Ring : 0;
          mid[(rslot + (sub >> 1)) * kFs2MidW + 2 * rqx + (sub & 1)] =
              easu_strict_pixel<FMT>(l, (rqy + 1) * kFs2FpW + (rqx + 1), (sub & 1) ? 0.75f : 0.25f, (sub >> 1) ? 0.75f : 0.25f);
        });
      } else if (!EXACT && xin0 && xin1 && yin0 && yin1) {  // default arithmetic: two row pairs, every texel read once per row (see easu_kernel)
        const float4_t A[4] = {l.ana[f_idx], l.ana[f_idx + 1], l.ana[f_idx + kFs2FpW], l.ana[f_idx + kFs2FpW + 1]};
        EasuBounds m;
        rgbf_t q00, q10, q01, q11;
        easu_quad_row<true>(l, f_idx, 0.25f, A, m, q00, q10);
        *reinterpret_cast<pair_t*>(m0) = TexelPair<FMT>::make(easu_resolve<FMT, EXACT>(m, q00, false), easu_resolve<FMT, EXACT>(m, q10, false));
        easu_quad_row<false>(l, f_idx, 0.75f, A, m, q01, q11);
        *reinterpret_cast<pair_t*>(m0 + kFs2MidW) = TexelPair<FMT>::make(easu_resolve<FMT, EXACT>(m, q01, false), easu_resolve<FMT, EXACT>(m, q11, false));
      } else if (xin0 && xin1 && yin0 && yin1) {  // every quad but those on the image's border
        const EasuBounds m = easu_bounds(l, f_idx);
        const texel_t p00 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.25f), false);
        const texel_t p10 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.25f), false);
        const texel_t p01 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.75f), false);
        const texel_t p11 = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.75f), false);
        *reinterpret_cast<pair_t*>(m0) = TexelPair<FMT>::make(p00, p10);
        *reinterpret_cast<pair_t*>(m0 + kFs2MidW) = TexelPair<FMT>::make(p01, p11);
      } else {
        texel_t p[4] = {Pixel<FMT>::zero(), Pixel<FMT>::zero(), Pixel<FMT>::zero(), Pixel<FMT>::zero()};
        if ((xin0 || xin1) && (yin0 || yin1)) {
          const EasuBounds m = easu_bounds(l, f_idx);
          if (xin0 && yin0) p[0] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.25f), false);
          if (xin1 && yin0) p[1] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.25f), false);
          if (xin0 && yin1) p[2] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.25f, 0.75f), false);
          if (xin1 && yin1) p[3] = easu_resolve<FMT, EXACT>(m, easu_pixel<EXACT>(l, f_idx, 0.75f, 0.75f), false);
        }
        m0[0] = p[0]; m0[1] = p[1]; m0[kFs2MidW] = p[2]; m0[kFs2MidW + 1] = p[3];
      }
    }
    __syncthreads();

    // ---- phase 4: RCAS from the ring.  Lane L owns apron column L; wave w takes the ring rows 1 + 4 w .. 4 + 4 w as centres
    //      (step 0 has no rows above its row 2: its first two are left out) with b / e / h in registers; d and f are the
    //      neighbouring lanes' centre texels (DPP wave shifts).  Every lane stays active (DPP sources); lanes 0 / 63 and pixels
    //      outside the image store nothing. ----
    {
      int lane = (int)threadIdx.x & 63;
      if (STRICT) asm volatile("" : "+v"(lane));  // (nothing per-lane lives across the re-evaluation: re-derived here)
      const int ox = ax0 + lane;
      const bool col_ok = lane >= 1 && lane <= kFs2OutW && ox < W;
      const uint32_t col_off = (uint32_t)ox * (uint32_t)sizeof(texel_t);  // (never used when ox < 0: lane 0 stores nothing)
      const int skip = (s == 0 && wave == 0) ? 2 : 0;
      const int j0 = 1 + 4 * wave + skip, nrows = 4 - skip;
      auto row = [&](int j) {  // (wave-uniform)
        int slot = base + j;
        slot -= slot >= kFs2Ring ? kFs2Ring : 0;
        return mid + slot * kFs2MidW + lane;
      };
      rgb_t prev = rgb(*row(j0 - 1));
      texel_t e_raw = *row(j0);
      rgb_t cur = rgb(e_raw);
      auto do_row = [&](int r) {
        const int j = j0 + r, oy = ay0 + j - 2;
        const texel_t n_raw = *row(j + 1);
        const rgb_t next = rgb(n_raw);
        const rgb_t d = rgb_t{dpp_f32<kDppWaveShr1>(cur.r, cur.r), dpp_f32<kDppWaveShr1>(cur.g, cur.g), dpp_f32<kDppWaveShr1>(cur.b, cur.b)};
        const rgb_t f = rgb_t{dpp_f32<kDppWaveShl1>(cur.r, cur.r), dpp_f32<kDppWaveShl1>(cur.g, cur.g), dpp_f32<kDppWaveShl1>(cur.b, cur.b)};
        const rgb_t p = rcas_pixel<EXACT>(prev, d, cur, f, next, sharp, flags);
        if (col_ok && oy < H) {
          const float pa = (flags & FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA) ? Pixel<FMT>::load(e_raw).w : 1.0f;
          store_out<sizeof(texel_t)>(out_frame + (long long)oy * a.out.pitch + col_off, Pixel<FMT>::store(p.r, p.g, p.b, pa), stream);
        }
        prev = cur; cur = next; e_raw = n_raw;
      };
      do_row(0); do_row(1);
      if (nrows == 4) { do_row(2); do_row(3); }  // wave-uniform
    }
    if (!RUN || s + 1 == steps || ay0 + kFs2Step - 1 >= H) break;  // (the next step's first output row is ay0 + 15)
    base += kFs2Step;
    base -= base >= kFs2Ring ? kFs2Ring : 0;
  }
}

template <int FMT, bool EXACT, bool RUN, int WAVES, bool STRICT = false>
static hipError_t fused_s2_launch_one(const FusedArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.frames)), block(64 * WAVES);
  const size_t lds = STRICT ? fused_s2_strict_lds_bytes(FMT, WAVES) : fused_s2_lds_bytes(FMT, WAVES);
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fused_s2_kernel<FMT, EXACT, RUN, WAVES, STRICT>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((fused_s2_kernel<FMT, EXACT, RUN, WAVES, STRICT>), grid, block, lds, stream, a);
  return hipGetLastError();
}

// tall: the 512-thread workgroup (one-step launches only: the host pairs it with run_steps == 1)
// strict: F-strict (the host has routed RGBA32F to EXACT: there is no store conversion to test against)
hipError_t fused_s2_launch(const FusedArgs& a, int fmt, bool exact, bool tall, hipStream_t stream, bool strict) {
  if (strict) {
    if (exact || fmt == FSR1_FORMAT_RGBA32F) return hipErrorInvalidValue;
#define FSR1_LAUNCH_S(F)                                                         \
  if (tall) return fused_s2_launch_one<F, false, false, 8, true>(a, stream);     \
  return a.run_steps > 1 ? fused_s2_launch_one<F, false, true, 4, true>(a, stream) : fused_s2_launch_one<F, false, false, 4, true>(a, stream)
    switch (fmt) {
      case FSR1_FORMAT_RGBA16F: FSR1_LAUNCH_S(FSR1_FORMAT_RGBA16F);
      case FSR1_FORMAT_RGBA8_UNORM: FSR1_LAUNCH_S(FSR1_FORMAT_RGBA8_UNORM);
      case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_LAUNCH_S(FSR1_FORMAT_R10G10B10A2_UNORM);
      default: return hipErrorInvalidValue;
    }
#undef FSR1_LAUNCH_S
  }
#define FSR1_LAUNCH_E(F)                                                                                                                  \
  if (tall && F != FSR1_FORMAT_RGBA32F)                                                                                                   \
    return exact ? fused_s2_launch_one<F == FSR1_FORMAT_RGBA32F ? FSR1_FORMAT_RGBA16F : F, true, false, 8>(a, stream)                     \
                 : fused_s2_launch_one<F == FSR1_FORMAT_RGBA32F ? FSR1_FORMAT_RGBA16F : F, false, false, 8>(a, stream);                   \
  return a.run_steps > 1 ? (exact ? fused_s2_launch_one<F, true, true, 4>(a, stream) : fused_s2_launch_one<F, false, true, 4>(a, stream)) \
                         : (exact ? fused_s2_launch_one<F, true, false, 4>(a, stream) : fused_s2_launch_one<F, false, false, 4>(a, stream))
  switch (fmt) {
    case FSR1_FORMAT_RGBA16F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA16F);
    case FSR1_FORMAT_RGBA32F: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA32F);
    case FSR1_FORMAT_RGBA8_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_RGBA8_UNORM);
    case FSR1_FORMAT_R10G10B10A2_UNORM: FSR1_LAUNCH_E(FSR1_FORMAT_R10G10B10A2_UNORM);
    default: return hipErrorInvalidValue;
  }
#undef FSR1_LAUNCH_E
}

}  // namespace fsr1
