/*
 * fsr1_hip.h — C ABI of the MI355X-native FSR 1.0 hot path (EASU upsample + RCAS sharpen).
 *
 * Drop-in boundary for the reference's source-level operator set:
 *   host   : FsrEasuCon / FsrEasuConOffset / FsrRcasCon — same spellings, argument order and
 *            bit-exact outputs as the A_CPU build of ffx-fsr/ffx_fsr1.h:156-225, :662-672.
 *   device : fsr1_easu_dispatch / fsr1_rcas_dispatch / fsr1_easu_rcas_fused_dispatch replace the
 *            compute dispatches of sample/src/DX12/FSR_Pass.hlsl:106-118 (mainCS -> CurrFilter ->
 *            FsrEasuF|H / FsrRcasF|H) issued by FSR_Filter::Upscale
 *            (sample/src/DX12/FSR_Filter.cpp:101-141); fsr1_upscale is Upscale() itself.
 *
 * Plain pointers and sizes only; device memory is caller-owned; every dispatch is asynchronous on
 * the HIP stream passed as an opaque void* (NULL = the default stream).  The library holds no
 * process-wide state that a call can set or that changes what a later call does: besides the per-thread
 * last-error string there are only idempotent per-device caches of device attributes (the CU count and
 * the dynamic-LDS limit already granted to a kernel).  The switches that force a launch shape for tests
 * live in a separate library, libfsr1_hip_test.so (include/fsr1_hip_test.h), not in this one.
 * All functions returning int return 0 on success and a negative fsr1_status on failure.
 */
#ifndef FSR1_HIP_H
#define FSR1_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSR1_HIP_VERSION 100 /* 1.0.0 */

/* ------------------------------------------------------------------------------------------------
 * Constant setup (host, pure, re-entrant).  Replaces ffx_fsr1.h:156-202, :205-225, :662-672.
 * Sizes are passed as floats exactly like the reference.  con0..con3 / con are caller-owned
 * uint32_t[4] each.
 * ---------------------------------------------------------------------------------------------- */
void FsrEasuCon(uint32_t* con0, uint32_t* con1, uint32_t* con2, uint32_t* con3,
                float inputViewportInPixelsX, float inputViewportInPixelsY,
                float inputSizeInPixelsX, float inputSizeInPixelsY,
                float outputSizeInPixelsX, float outputSizeInPixelsY);

void FsrEasuConOffset(uint32_t* con0, uint32_t* con1, uint32_t* con2, uint32_t* con3,
                      float inputViewportInPixelsX, float inputViewportInPixelsY,
                      float inputSizeInPixelsX, float inputSizeInPixelsY,
                      float outputSizeInPixelsX, float outputSizeInPixelsY,
                      float inputOffsetInPixelsX, float inputOffsetInPixelsY);

/* sharpness in stops: 0.0 = maximum, N>0 halves the sharpening N times (ffx_fsr1.h:645). */
void FsrRcasCon(uint32_t* con, float sharpness);

/* ffx_a.h:482-549 — float -> half by truncation, +-inf/NaN -> +-65504; used by FsrRcasCon. */
uint32_t AU1_AH1_AF1(float f);

/* ------------------------------------------------------------------------------------------------
 * Images.  Pixels are RGBA interleaved; rows are row_pitch_bytes apart, frames (independent
 * images of one batch, processed by one launch) frame_stride_bytes apart.
 * ---------------------------------------------------------------------------------------------- */
typedef enum fsr1_format {
  FSR1_FORMAT_RGBA16F = 0, /* 8 B/pixel; the view type the reference shader declares
                              (Texture2D<AH4> / RWTexture2D<AH4>, FSR_Pass.hlsl:51-52) */
  FSR1_FORMAT_RGBA32F = 1, /* 16 B/pixel; the SAMPLE_SLOW_FALLBACK view (FSR_Pass.hlsl:34-36) */
  /* 32 bpp formats — what the sample actually renders into and presents ("use 32bpp formats", ffx_fsr1.h:91;
   * DXGI_FORMAT_R8G8B8A8_UNORM / R10G10B10A2_UNORM swap chains, SampleRenderer.cpp:193).  The texture unit's
   * conversions, which the reference leaves to the API, are pinned here to the D3D11 functional-spec rules:
   *   load  : code / (2^n - 1), correctly rounded to binary32
   *   store : (uint) fma(clamp(x, 0, 1), 2^n - 1, 0.5) — one rounding, then truncation; NaN stores 0
   * Arithmetic between load and store is the F (binary32) path. */
  FSR1_FORMAT_RGBA8_UNORM = 2,        /* 4 B/pixel: R bits 0-7, G 8-15, B 16-23, A 24-31 */
  FSR1_FORMAT_R10G10B10A2_UNORM = 3   /* 4 B/pixel: R bits 0-9, G 10-19, B 20-29, A 30-31 */
} fsr1_format;

typedef struct fsr1_image {
  void* data;                 /* device pointer */
  int32_t width, height;      /* in pixels */
  int32_t format;             /* fsr1_format */
  int32_t frames;             /* >= 1 */
  int64_t row_pitch_bytes;    /* 0 = tightly packed */
  int64_t frame_stride_bytes; /* 0 = tightly packed */
} fsr1_image;

/* ------------------------------------------------------------------------------------------------
 * Dispatch flags
 * ---------------------------------------------------------------------------------------------- */
enum {
  /* `if (Sample.x == 1) c *= c;` after the filter (FSR_Pass.hlsl:78-79, :92-93): HDR gamma2 -> linear */
  FSR1_FLAG_HDR_SQUARE = 1u << 0,
  /* ffx_fsr1.h:651 FSR_RCAS_DENOISE */
  FSR1_FLAG_RCAS_DENOISE = 1u << 1,
  /* ffx_fsr1.h:648 FSR_RCAS_PASSTHROUGH_ALPHA (otherwise alpha is written as 1, FSR_Pass.hlsl:80) */
  FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA = 1u << 2,
  /* Arithmetic selection.  Default (neither bit): fp32 arithmetic with FMA contraction in the
   * continuous part of the filter — the FsrEasuF / FsrRcasF entry points ("F" parity class:
   * within 1 binary16 ULP of the reference's CPU-evaluated F path FOR INPUT IN THE RANGE THE
   * REFERENCE DEFINES THE FILTERS ON, {0 to 1} ("RCAS solves for 'w' by seeing where the signal
   * might clip out of the {0 to 1} input range", ffx_fsr1.h:618; linear HDR {0 to FP16_MAX} goes
   * through FsrSrtmF into {0 to 1} first, :1035-1040).  Outside
   * that range the class still holds for >= 99.9 % of the values, but where one 12-tap window mixes
   * magnitudes 1e4 ... 1e10 apart the negative lobes cancel terms far larger than the result and
   * about one value in 1e4 lands 2-8 ULP away (tests/test_gpu_special_values.py); EXACT has no such
   * boundary: it is bit-identical for every input, NaN / Inf / signed zeros included).
   * EXACT: fp32 arithmetic in the reference's exact operation order, no contraction, IEEE
   * division — bit-identical (fp32) to the CPU-evaluated FsrEasuF / FsrRcasF.
   * PACKED_FP16: the FsrEasuH / FsrRcasH entry points — packed binary16 arithmetic
   * (v_pk_*_f16), parity class "H" (vs the reference's CPU-evaluated H path). */
  FSR1_FLAG_MATH_EXACT = 1u << 4,
  FSR1_FLAG_MATH_PACKED_FP16 = 1u << 5,
  /* STRICT ("F-strict", round 6): EASU's stored image is BIT-IDENTICAL to FSR1_FLAG_MATH_EXACT's — i.e. to the CPU-evaluated
   * FsrEasuF rounded to the storage format — at close to the default arithmetic's speed: every pixel is evaluated with the default
   * arithmetic and tested against the store conversion's rounding boundaries with a margin that covers the default arithmetic's
   * measured distance from the reference order (|default - EXACT| <= 30 x 2^-24 x the 12-tap window's largest |R|,|G|,|B| over
   * 8e12 values of every kind of content, <= 37.2 under an adversarial search; threshold 56: include/fsr1_device_easu.hpp); the
   * 4-6 % of pixels that fail the test are
   * re-evaluated in the reference's operation order inside the same launch.  RCAS (as its own dispatch or as the second half of the
   * fused launch) runs the DEFAULT arithmetic under this flag — within 1 binary16 ULP of FsrRcasF on identical input — so the final
   * image of EASU -> RCAS is within 1 ULP of the reference chain FsrEasuF -> RTNE -> FsrRcasF end to end (the default arithmetic:
   * 99.99 % within 1 ULP, max 6).  Every pipeline (two dispatches, fused launch, fsr1_upscale, fsr1_pipeline) produces the same bits
   * under this flag.  Where EASU has no strict variant — RGBA32F storage (no conversion to test against), colour stages — EASU and
   * RCAS both take the EXACT kernels under this flag; an EASU-only launch with `c *= c` takes the EXACT kernel.  Content whose every
   * window mixes magnitudes (HDR noise) fails the test almost everywhere and runs at about the EXACT kernels' speed.  Exclusive with
   * MATH_EXACT and MATH_PACKED_FP16. */
  FSR1_FLAG_MATH_STRICT = 1u << 6,
  /* Diagnostics: never pick a shape-specialised kernel (e.g. the exact-2x variants, whose lanes own 2x2 output quads).
   * The specialised kernels run the same per-pixel arithmetic on the same values: results are bit-identical either way
   * (tests assert it); the flag exists so that this can be checked and the gain measured. */
  FSR1_FLAG_NO_FAST_PATHS = 1u << 8,
  /* Store policy of the pass's output image.  STREAMING: non-temporal stores — the image is not read again soon, so it
   * should not displace what is (measured on MI355X at 4K: RCAS 30.9 -> 25.7 us, two dispatches 70.2 -> 67.5 us).
   * CACHED: plain stores — a reader follows (the EASU -> RCAS intermediary).  Defaults: EASU CACHED (RCAS normally
   * follows; STREAMING when the image is larger than 512 MB — twice the Infinity Cache: nothing would keep it anyway), RCAS and the
   * fused launch STREAMING (their output is the pipeline's last image); fsr1_upscale sets
   * STREAMING on whichever pass writes `out`.  The pixels stored are the same either way. */
  FSR1_FLAG_OUTPUT_STREAMING = 1u << 9,
  FSR1_FLAG_OUTPUT_CACHED = 1u << 10,
  /* Scheduling hint: other, independent frames run beside this dispatch on other HIP streams (fsr1_pipeline sets it for its own
   * submissions when it has more than one stream).  Launch geometry is then chosen for the throughput of the overlapped stream of
   * frames instead of for the latency of this launch alone: the exact-2x fused launch walks its columns in longer runs (one 4K
   * frame: 4 steps instead of one-step tiles, 56.6 -> 54 us per frame on two streams — and 72 us if the hint is wrong and the
   * launch runs alone).  The pixels are the same either way. */
  FSR1_FLAG_FRAMES_OVERLAP = 1u << 11
};

typedef enum fsr1_status {
  FSR1_OK = 0,
  FSR1_ERR_INVALID_ARGUMENT = -1,
  FSR1_ERR_UNSUPPORTED = -2,
  FSR1_ERR_HIP = -3
} fsr1_status;

/* ------------------------------------------------------------------------------------------------
 * Device passes
 * ---------------------------------------------------------------------------------------------- */

/* EASU: out[frame] = FsrEasuF|H(in[frame]) for every output pixel; con = con0|con1|con2|con3
 * (16 words) from FsrEasuCon / FsrEasuConOffset.  Taps are clamped to the edge of the input
 * *resource* (in->width x in->height), like the CLAMP sampler (FSR_Filter.cpp:48-53). */
int fsr1_easu_dispatch(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags,
                       void* stream);

/* RCAS: in and out have the same size; loads outside the image return 0 (D3D Load, FSR_Pass.hlsl:45,61).
 * con = the 4 words from FsrRcasCon.  in and out must not alias. */
int fsr1_rcas_dispatch(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags,
                       void* stream);

/* EASU -> RCAS in one launch: the EASU result of each output tile (+1 pixel apron) is kept in LDS,
 * rounded to the intermediate format the two-pass pipeline would have stored (binary16 for
 * RGBA16F images), and sharpened from there — bit-identical to the two dispatches above with an
 * RGBA16F/RGBA32F intermediary of out's format, without its 2 x out bytes of HBM traffic. */
int fsr1_easu_rcas_fused_dispatch(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16],
                                  const uint32_t rcas_con[4], uint32_t flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Colour stages either side of the filters: the tone-mapping, film-grain and dither helpers
 * of ffx-fsr/ffx_fsr1.h:986-1199, as a stand-alone pass and as fused prologue / epilogue of the passes above
 * (the reference leaves them to the integration's own shaders, e.g. sample/src/DX12/FSR_Tonemapping.hlsl:87).
 *
 * Stages run in this fixed order, each only if its bit is set:
 *   FSR1_COLOR_SRTM      FsrSrtmF     (:1042)  c *= rcp(max3(c) + 1)            HDR {0..FP16_MAX} -> {0..1}
 *   FSR1_COLOR_LFGA      FsrLfgaF     (:1012)  c += (t*amount) * min(1 - c, c)  t = noise.rgb + grain_bias
 *   FSR1_COLOR_SRTM_INV  FsrSrtmInvF  (:1044)  c *= rcp(max(1/32768, 1 - max3(c)))
 *   FSR1_COLOR_TEPD_C8 | FSR1_COLOR_TEPD_C10   FsrTepdC8F / FsrTepdC10F (:1097, :1113): {0..1} linear -> dithered
 *                        gamma 2.0 on the 8-bit / 10-bit code grid, dither = FsrTepdDitF(pixel, frame) (:1082), or,
 *                        with FSR1_COLOR_DITHER_FROM_NOISE, saturate(noise.a) (FSR_Tonemapping.hlsl:87).
 * noise: tiled ("tiled blue noise", :1000) with wrap addressing,
 *   texel = noise[frame % noise->frames][(y + noise_offset_y) mod height][(x + noise_offset_x) mod width]
 * in any fsr1_format; signed grain {-0.5..0.5} as float texels with grain_bias 0, or UNORM texels with grain_bias -0.5.
 * Alpha is not touched by any stage.
 * Arithmetic: binary32 in the reference's operation order, no contraction; with FSR1_FLAG_MATH_EXACT the two
 * reciprocals (ARcpF1) are IEEE divisions and every stage is bit-identical to the CPU-evaluated reference,
 * otherwise they are v_rcp_f32 (1 ulp).  FsrTepd*F's sqrt is correctly rounded in both modes.
 * ---------------------------------------------------------------------------------------------- */
enum {
  FSR1_COLOR_SRTM = 1u << 0,
  FSR1_COLOR_LFGA = 1u << 1,
  FSR1_COLOR_SRTM_INV = 1u << 2,
  FSR1_COLOR_TEPD_C8 = 1u << 3,
  FSR1_COLOR_TEPD_C10 = 1u << 4,
  FSR1_COLOR_DITHER_FROM_NOISE = 1u << 5
};

typedef struct fsr1_color_stages {
  uint32_t stages;          /* FSR1_COLOR_* bits */
  float grain_amount;       /* FsrLfgaF `a`, {0..1} */
  float grain_bias;         /* added to noise.rgb before FsrLfgaF */
  uint32_t frame;           /* FsrTepdDitF `f`; also selects the noise slice */
  int32_t noise_offset_x, noise_offset_y;
  const fsr1_image* noise;  /* required by FSR1_COLOR_LFGA and FSR1_COLOR_DITHER_FROM_NOISE, else may be NULL */
} fsr1_color_stages;

/* Stand-alone pass: out[frame] = stages(in[frame]); in and out have the same size and may have different formats
 * (e.g. RGBA16F -> RGBA8_UNORM with FSR1_COLOR_TEPD_C8); in == out (in place) is allowed when the formats match.
 * flags: 0, FSR1_FLAG_MATH_EXACT, or FSR1_FLAG_MATH_PACKED_FP16 — the half-precision entry points FsrSrtmH /
 * FsrLfgaH / FsrSrtmInvH / FsrTepdC8H | C10H (+ their Hx2 forms; ffx_fsr1.h:1017-1024, :1048-1056, :1124-1198), RGBA16F
 * in and out, parity class "H" (bit-identical to the reference's CPU-evaluated H path). */
int fsr1_color_dispatch(const fsr1_image* in, const fsr1_image* out, const fsr1_color_stages* stages, uint32_t flags,
                        void* stream);

/* The passes above with colour stages fused in (no extra trip through HBM):
 *   prologue  FSR1_COLOR_SRTM is applied to every input texel as it is loaded — the role of a colour transform
 *             inside the FsrEasu{R,G,B}F gather callbacks (ffx_fsr1.h:234-236) / FsrRcasInputF (:682);
 *   epilogue  LFGA, SRTM_INV and TEPD are applied to the filter's result (after the optional `c *= c`) before
 *             it is stored.
 * fsr1_easu_dispatch_ex takes the prologue and, being the last pass when RCAS is off, the epilogue;
 * fsr1_rcas_dispatch_ex takes both; fsr1_easu_rcas_fused_dispatch_ex applies the prologue at the EASU loads and
 * the epilogue after RCAS.  `stages` == NULL (or no bits) is the plain pass.  out->format may differ from
 * in->format only as RGBA16F -> RGBA8_UNORM / R10G10B10A2_UNORM (the TEPD targets); the EASU->RCAS
 * intermediary of the fused kernel keeps in->format. */
int fsr1_easu_dispatch_ex(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags,
                          const fsr1_color_stages* stages, void* stream);
int fsr1_rcas_dispatch_ex(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags,
                          const fsr1_color_stages* stages, void* stream);
int fsr1_easu_rcas_fused_dispatch_ex(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16],
                                     const uint32_t rcas_con[4], uint32_t flags, const fsr1_color_stages* stages,
                                     void* stream);

/* ----------------------------------------------------------------------------------------------
 * Row bands of one frame (optional): a single frame split over several GPUs without any
 * exchange.  Every GPU holds the whole INPUT frame and produces a band of output rows:
 *   1. fsr1_easu_dispatch_band writes output rows [origin_y - 1, origin_y + rows + 1) (clipped to the image) of
 *      the EASU result into a band-sized intermediary: `out` is that band, (origin_x, origin_y) the position of its
 *      pixel (0, 0) in the full output image `con` was set up for (the position arithmetic of ffx_fsr1.h:324-326
 *      runs on full-image coordinates, so the band is bit-identical to the same rows of a full-frame dispatch);
 *   2. fsr1_rcas_dispatch_band sharpens the band's own rows: `in` describes them, rows_above / rows_below (0 or 1)
 *      say that the row just above / below exists in memory (the extra EASU rows of step 1) and is to be read
 *      instead of being treated as outside the image (= 0).
 * fsr1_easu_rcas_fused_dispatch_band is the single-launch form of the two steps: `out` holds output rows
 * [origin_y, origin_y + out->height) of the full image, and where rows_above / rows_below say that the full image
 * has a row above / below the band, the tile aprons compute that row with EASU (full-image coordinates) instead of
 * treating it as outside the image.  No intermediary; bit-identical to the same rows of a full-frame launch.
 * F arithmetic (default or FSR1_FLAG_MATH_EXACT), one frame per dispatch for the RCAS band.
 * ---------------------------------------------------------------------------------------------- */
int fsr1_easu_dispatch_band(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags,
                            int32_t origin_x, int32_t origin_y, void* stream);
int fsr1_rcas_dispatch_band(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags,
                            int32_t rows_above, int32_t rows_below, void* stream);
int fsr1_easu_rcas_fused_dispatch_band(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16],
                                       const uint32_t rcas_con[4], uint32_t flags, int32_t origin_y, int32_t rows_above,
                                       int32_t rows_below, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FSR_Filter::Upscale (sample/src/DX12/FSR_Filter.cpp:101-141) as one call.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fsr1_params {
  float render_width, render_height;   /* pState->renderWidth/Height: viewport == input resource size */
  int32_t use_rcas;                    /* pState->bUseRcas */
  float rcas_attenuation;              /* pState->rcasAttenuation, stops (sample default 0.25, SampleRenderer.h:49) */
  int32_t hdr;                         /* `hdr` argument of Upscale: Sample.x = hdr && !use_rcas for EASU, hdr for RCAS */
  int32_t fused;                       /* 0: EASU + RCAS as two dispatches; 1: the single fused launch (needs use_rcas);
                                          2: whichever is faster on MI355X, decided on round-2 measurements: at exactly
                                          2x with F arithmetic and no colour stages the fused launch, whose quad form wins
                                          at every size (4K: 63 vs 67 us; 8K x16: 3.90 vs 4.08 ms; 1440p: 31 vs 38 us);
                                          otherwise the two dispatches whenever the launch has more than 3 Mpixel of output
                                          (4K at 1.5x: 80.7 vs 88.8 us) and the fused launch below that, where a frame is
                                          launch-bound, for every arithmetic (FSR1_FLAG_MATH_PACKED_FP16 has a fused launch
                                          too; at exactly 2x its quad form is taken up to 4 Mpixel and from 60 Mpixel of output
                                          up — round 5: 720p -> 1440p 50.9 vs 47.0 us, sixteen 8K frames 5.45 vs 5.24 ms, one 4K
                                          frame the other way, 93.7 vs 95.6) — and only where the fused tile fits a CU's LDS (up to about 1.9x
                                          minification; beyond that auto keeps the two dispatches); with intermediary == NULL
                                          it is the fused launch */
  uint32_t flags;                      /* FSR1_FLAG_MATH_*, FSR1_FLAG_RCAS_DENOISE / _PASSTHROUGH_ALPHA, FSR1_FLAG_OUTPUT_* (of the pass that writes `out`) */
} fsr1_params;

/* in -> (intermediary) -> out.  `intermediary` may be NULL when use_rcas == 0 or fused == 1. */
int fsr1_upscale(const fsr1_image* in, const fsr1_image* intermediary, const fsr1_image* out,
                 const fsr1_params* params, void* stream);

/* fsr1_upscale with colour stages fused in (NULL = none): the prologue goes to EASU's loads, the epilogue to the
 * pass that writes `out` (RCAS, or EASU when use_rcas == 0), after the `hdr` square. */
int fsr1_upscale_ex(const fsr1_image* in, const fsr1_image* intermediary, const fsr1_image* out,
                    const fsr1_params* params, const fsr1_color_stages* stages, void* stream);

/* Which pipeline fsr1_upscale[_ex] would run for these arguments, without launching anything: 0 = EASU + RCAS as two
 * dispatches, 1 = the single fused launch, 2 = EASU only (use_rcas == 0); a negative fsr1_status for bad arguments.
 * have_intermediary / have_stages: whether the call would pass an intermediary image / colour stages.  Lets a host that
 * uses fused = 2 ("auto") account bytes and report timings for the path actually taken (runner/fsr1_runner.c). */
int fsr1_upscale_plan(const fsr1_image* in, int32_t have_intermediary, const fsr1_image* out, const fsr1_params* params,
                      int32_t have_stages);

/* ------------------------------------------------------------------------------------------------
 * Frame pipeline: independent frames on alternating HIP streams.
 *
 * FSR 1.0 keeps no history, so consecutive frames of a stream are independent — but dispatches on ONE HIP stream serialise,
 * and every kernel boundary costs the chip about 5 us in which it computes nothing (the last workgroups drain, caches write
 * back, the next dispatch ramps up; measured on MI355X, profiles/ab_r04/r4c1_fused_trace.log: a fused 4K launch is 59.6 us
 * from first to last instruction and 65 us per launch back to back).  A pipeline owns N non-blocking streams and N
 * EASU -> RCAS intermediaries and sends frame i to stream i mod N: the tail of one frame overlaps the head of the next.
 * Measured (profiles/ab_r04/r4c11_streams_sweep.log, us per frame, N = 1 / 2 / 3 / 4): 1080p -> 4K two dispatches 67.6 / 61.7 / 59.8 /
 * 64.0, fused launch 63.0 / 54.7 / 54.4 / 59.0, 1440p -> 4K 79.8 / 71.3 / 70.9 / 74.6, 540p -> 1080p two dispatches 28.7 / 20.4 / 17.1 /
 * 22.0: N = 3 is never worse than 2 and what bench.py and the runner default to; 4 loses.  The reference's sample has one graphics queue and
 * no counterpart; this is what its async-compute note (FSR_Filter.cpp:101 ff. run inside the frame's command list) leaves to
 * the engine.
 *
 * Ordering and aliasing — the rule the code enforces, no more: HIP streams are ordered only within themselves.  Submission i runs
 * on slot i mod N and is ordered after submission i - N, i - 2N, ... (the same slot) and after nothing else: a long submission on
 * one slot can still be running while several later ones on the other slots have come and gone.  Therefore
 *   - an image (input, output) may be reused by a LATER submission only if that submission lands on the SAME slot
 *     (fsr1_pipeline_next_slot tells which slot the next submission takes), or after fsr1_pipeline_join / _synchronize;
 *   - a ring of frame buffers that is a multiple of N long, walked in submission order, satisfies this by construction
 *     (bench.py and runner/fsr1_runner.c round their rings up to a multiple of N for that reason);
 *   - "the last N submissions" is NOT a safe window in general (N = 2: submission 4 may start while a long submission 1 still runs).
 * A pipeline is driven by one host thread at a time.  fork / join order the pipeline's streams against a stream of the caller's.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fsr1_pipeline fsr1_pipeline; /* opaque */
int fsr1_pipeline_create(fsr1_pipeline** pipeline, int32_t streams /* 1 .. 8 */);
/* fsr1_upscale_ex for one frame (or batch) on the pipeline's next stream, with that stream's own intermediary
 * (params->fused = 2 decides as if an intermediary was supplied).  A BATCH that takes the two dispatches is submitted frame by frame —
 * frame f on slot (next_slot + f) mod N, the slot counter advancing by the frame count — whenever the per-frame intermediaries of all
 * N slots fit the 256 MB Infinity Cache together (e.g. 4K RGBA16F frames on three streams): +6 % on an 8-frame 1440p -> 4K batch,
 * the same pixels (round 5); larger frames and fused launches stay one launch on one slot.  Asynchronous: when the slot's intermediary is too small it is
 * re-allocated in stream order on the slot's own stream (hipFreeAsync / hipMallocAsync — no host block, no device-wide
 * synchronisation).  While the stream is being captured into a hipGraph a growth is refused with FSR1_ERR_INVALID_ARGUMENT and a
 * message naming the size to reserve: call fsr1_pipeline_reserve before the capture.  A hipGraph captured EARLIER holds the
 * intermediary's address of that time: any later growth of the slot frees that buffer and INVALIDATES the graph (replaying it would
 * write into memory the pool may have handed to another slot) — reserve the largest size the pipeline will ever see before the first
 * capture.  On a device without memory pools (hipDeviceAttributeMemoryPoolsSupported == 0) a growth drains the slot's stream and uses
 * hipFree / hipMalloc instead.  Every frame of a split batch runs what was decided for the batch (two dispatches); all arguments are
 * validated before frame 0 is submitted, and if a HIP call fails mid-batch the error message says how many frames were submitted. */
int fsr1_pipeline_upscale(fsr1_pipeline* pipeline, const fsr1_image* in, const fsr1_image* out, const fsr1_params* params,
                          const fsr1_color_stages* stages);
/* Pre-sizes every slot's intermediary to at least bytes_per_stream (= out width x height x bytes per pixel of the largest frame the
 * two-dispatch pipeline will see; x frames for batches that are not split, see fsr1_pipeline_upscale), so that no later submission allocates: what a host does once before capturing frames
 * into a hipGraph, or to keep allocation out of its frame loop.  Intermediaries never shrink. */
int fsr1_pipeline_reserve(fsr1_pipeline* pipeline, size_t bytes_per_stream);
/* The slot (0 .. streams - 1) the next fsr1_pipeline_upscale will run on; -1 for a null pipeline.  See the aliasing rule above. */
int fsr1_pipeline_next_slot(const fsr1_pipeline* pipeline);
/* Work submitted to the pipeline after fork() starts only after what `stream` holds now (e.g. the producer of the inputs). */
int fsr1_pipeline_fork(fsr1_pipeline* pipeline, void* stream);
/* Work submitted to `stream` after join() starts only after everything the pipeline holds now (e.g. a consumer of the outputs). */
int fsr1_pipeline_join(fsr1_pipeline* pipeline, void* stream);
/* Blocks the host until the pipeline is empty. */
int fsr1_pipeline_synchronize(fsr1_pipeline* pipeline);
int fsr1_pipeline_streams(const fsr1_pipeline* pipeline);
void* fsr1_pipeline_stream(const fsr1_pipeline* pipeline, int32_t i); /* the i-th hipStream_t, for callers that record events of their own */
int fsr1_pipeline_destroy(fsr1_pipeline* pipeline);

/* ------------------------------------------------------------------------------------------------
 * Diagnostics
 * ---------------------------------------------------------------------------------------------- */
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* fsr1_last_error(void);
int fsr1_version(void);
/* Number of HIP devices visible, or a negative fsr1_status. */
int fsr1_device_count(void);

/* Exhaustive on-device check of the hardware assumptions behind the kernels' reciprocals: *failures = number of operands
 * for which (a) the packed-binary16 kernels' reciprocal (all 65536 binary16 operands) or (b) the EXACT variants' binary32
 * reciprocal (rcp_ieee, include/fsr1_device_base.hpp: all 2^32 binary32 operands) differs from the correctly rounded 1/x
 * that the reference's ARcpH1/ARcpH2/ARcpF1 (ffx_a.h:1005, GLSL `1.0/x`) are pinned to. */
int fsr1_selftest(uint32_t* failures);

/* The hash of the kernel sources this binary was built from (csrc/ + include/, first 16 hex digits of a SHA-256), baked in at build
 * time.  A host that also has the source tree compares it with the tree's hash (fidelityfx-fsr_amd/_lib.py:source_hash(); bench.py
 * prints both and says "binary_matches_sources": false when a stale prebuilt binary is running). */
const char* fsr1_build_id(void);

/* HIP-event stopwatch on a caller stream (used by the bench so that kernel time is measured on the
 * very stream the kernels run on).  Handles are opaque. */
int fsr1_timer_create(void** timer);
int fsr1_timer_start(void* timer, void* stream);
int fsr1_timer_stop(void* timer, void* stream);
/* Blocks until the stop event completed; returns elapsed milliseconds through *ms. */
int fsr1_timer_elapsed_ms(void* timer, float* ms);
int fsr1_timer_destroy(void* timer);

#ifdef __cplusplus
}
#endif
#endif /* FSR1_HIP_H */
