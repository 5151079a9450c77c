// C ABI of the FSR 1.0 HIP path (include/fsr1_hip.h): argument checking, launch geometry, error
// reporting.  The host logic mirrors FSR_Filter::Upscale (sample/src/DX12/FSR_Filter.cpp:101-141).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>

#include "fsr1_device.h"

namespace fsr1 {
hipError_t easu_launch(const EasuArgs& a, int fmt, bool exact, bool s2, bool tall, hipStream_t stream, bool strict);
size_t easu_strict_lds_bytes(int fmt, int fp_w, int fp_h, int tile_h);
bool easu_s2_tall_tiles(int width, int height, int frames, bool overlapped, int cus);
bool easu_generic_tall_tiles(int width, int height, int frames, int cus, size_t lds_tall, size_t lds_per_cu);
int easu_lds_pitch(int fp_w, bool exact, bool color);
size_t easu_lds_bytes(int fmt, int fp_w, int fp_h);
hipError_t rcas_launch(const RcasArgs& a, int fmt, bool exact, hipStream_t stream);
void rcas_geometry(int width, int height, int frames, bool overlapped, int* tiles_x, int* tiles_y, int* rows);
hipError_t fused_launch(const FusedArgs& a, int fmt, bool exact, hipStream_t stream, bool strict);
size_t fused_strict_lds_bytes(int fmt, int fp_w, int fp_h);
size_t fused_lds_bytes(int fmt, int fp_w, int fp_h);
hipError_t fused_h_launch(const FusedArgs& a, hipStream_t stream);
hipError_t fused_s2_launch(const FusedArgs& a, int fmt, bool exact, bool tall, hipStream_t stream, bool strict);
size_t fused_s2_lds_bytes(int fmt, int waves);
size_t fused_s2_strict_lds_bytes(int fmt, int waves);
bool fused_s2_tall_tiles(int width, int height, int frames, int steps, int cus, int fmt);
void fused_s2_geometry(int width, int height, int steps, int* tiles_x, int* tiles_y, int step_rows);
int fused_s2_run_steps(int width, int height, int frames, int cus, int wgs_per_cu, bool overlapped, bool strict);
hipError_t fused_s2_h_launch(const FusedArgs& a, hipStream_t stream);
size_t fused_s2_h_lds_bytes();
size_t fused_h_lds_bytes(int fp_w, int fp_h);
hipError_t easu_h_launch(const EasuArgs& a, bool s2, hipStream_t stream);
hipError_t easu_color_launch(const EasuArgs& a, int fin, int fout, bool exact, hipStream_t stream);
hipError_t rcas_color_launch(const RcasArgs& a, int fin, int fout, bool exact, hipStream_t stream);
hipError_t fused_color_launch(const FusedArgs& a, int fin, int fout, bool exact, hipStream_t stream);
hipError_t color_launch(const ColorPassArgs& a, int fin, int fout, bool exact, hipStream_t stream);
hipError_t color_h_launch(const ColorPassArgs& a, hipStream_t stream);
void color_geometry(int width, int height, int* tiles_x, int* tiles_y);
hipError_t rcas_h_launch(const RcasArgs& a, hipStream_t stream);
}  // namespace fsr1

using namespace fsr1;

// fsr1_selftest: hardware assumptions of the packed-binary16 kernels, checked exhaustively over all 65536 operands.
//   bit 0 of a failing operand's entry: half_rcp(x) != RTNE(IEEE binary32 1.0f/x)
__global__ void selftest_kernel(uint32_t* failures) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= 65536u) return;
  const half_t x = __builtin_bit_cast(half_t, (unsigned short)p);
  const half_t fast = half_rcp(x);
  const half_t ieee = (half_t)(1.0f / (float)x);  // correctly rounded binary32 division (hipcc default), then RTNE
  const unsigned short fb = __builtin_bit_cast(unsigned short, fast), ib = __builtin_bit_cast(unsigned short, ieee);
  const bool both_nan = (fast != fast) && (ieee != ieee);
  if (fb != ib && !both_nan) atomicAdd(failures, 1u);
}

//   rcp_ieee(x) != 1.0f / x for any of the 2^32 binary32 operands (NaN results compare equal)
__global__ void selftest_rcp_kernel(uint32_t* failures) {
  const unsigned long long n = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  uint32_t bad = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = __uint_as_float((uint32_t)i);
    const float got = rcp_ieee(x), want = 1.0f / x;
    if (__float_as_uint(got) != __float_as_uint(want) && !(got != got && want != want)) ++bad;
  }
  if (bad) atomicAdd(failures, bad);
}

// Tracing: with FSR1_ROCTX=1 in the environment every dispatch is wrapped in a roctx range named
// after the pass and its extents, so rocprofv3 --marker-trace / a timeline viewer shows the frame structure.  the roctx library
// (librocprofiler-sdk-roctx / libroctx64) is resolved with dlopen on first use: the library has no link-time dependency on it and pays one branch when off.
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = getenv("FSR1_ROCTX");
    if (!e || !*e || *e == '0') return;
    void* h = nullptr;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
      if ((h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
struct Range {
  bool on = false;
  Range(const char* pass, const fsr1_image* in, const fsr1_image* out) {
    static const Roctx r;
    if (!r.push || !in || !out) return;
    char name[96];
    snprintf(name, sizeof name, "fsr1_%s %dx%d->%dx%d x%d", pass, in->width, in->height, out->width, out->height, out->frames);
    r.push(name);
    pop_ = r.pop;
    on = true;
  }
  ~Range() { if (on) pop_(); }
  int (*pop_)() = nullptr;
};
}  // namespace

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
static int hip_fail(hipError_t e, const char* what) {
  return fail(FSR1_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

static size_t pixel_bytes(int fmt) { return fmt == FSR1_FORMAT_RGBA32F ? 16 : (fmt == FSR1_FORMAT_RGBA16F ? 8 : 4); }

static int check_image(const fsr1_image* im, const char* name, ImageView* v) {
  if (!im) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: null image descriptor", name);
  if (!im->data) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: null data pointer", name);
  if (im->width <= 0 || im->height <= 0 || im->frames <= 0)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: bad extent %dx%dx%d", name, im->width, im->height, im->frames);
  if (im->width > (1 << 20) || im->height > (1 << 20))  // pixel coordinates stay exact in binary32 with room to spare
    return fail(FSR1_ERR_UNSUPPORTED, "%s: extent %dx%d beyond 2^20 pixels per side", name, im->width, im->height);
  if (im->format < FSR1_FORMAT_RGBA16F || im->format > FSR1_FORMAT_R10G10B10A2_UNORM)
    return fail(FSR1_ERR_UNSUPPORTED, "%s: unsupported format %d", name, im->format);
  const size_t px = pixel_bytes(im->format);
  const long long pitch = im->row_pitch_bytes ? im->row_pitch_bytes : (long long)im->width * (long long)px;
  if (pitch < (long long)im->width * (long long)px || pitch % (long long)px)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: row pitch %lld too small or not a multiple of the pixel size", name, pitch);
  const long long fstride = im->frame_stride_bytes ? im->frame_stride_bytes : pitch * im->height;
  if (im->frames > 1 && fstride < pitch * im->height)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: frame stride %lld smaller than one frame", name, fstride);
  if (((uintptr_t)im->data % px) || (fstride % (long long)px))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: data pointer / frame stride not aligned to the pixel size", name);
  v->base = static_cast<char*>(im->data);
  v->width = im->width;
  v->height = im->height;
  v->pitch = pitch;
  v->frame_stride = fstride;
  return FSR1_OK;
}

static bool overlaps(const fsr1_image* a, const ImageView& va, const fsr1_image* b, const ImageView& vb) {
  const char* a0 = va.base;
  const char* a1 = va.base + va.frame_stride * (a->frames - 1) + va.pitch * a->height;
  const char* b0 = vb.base;
  const char* b1 = vb.base + vb.frame_stride * (b->frames - 1) + vb.pitch * b->height;
  return a0 < b1 && b0 < a1;
}

// Exact size of the largest input footprint any output tile touches along one axis, evaluated with the very
// float arithmetic the kernels use for it (fp = floor(o*scale + bias), product and sum rounded separately):
//   tile [o0, ol] reads texels fp(o0)-1 .. fp(ol)+2.
// `apron` = 1 for the fused kernel, whose tiles compute one extra pixel on each side (clipped to the image).
// Exactness matters: LDS per workgroup decides how many workgroups a CU holds (35x11 texels x 48 B lets 8 of them
// in at 2x; two texels of slack per axis would leave 6).
// `lo` / `hi`: rows (columns) the apron may reach beyond the image (a band with neighbouring rows in the full image).
static int footprint_extent(int out_size, int tile, int apron, float scale, float bias, int origin = 0, int lo = 0, int hi = 0) {
  if (!(scale > 0.0f) || !std::isfinite(scale) || !std::isfinite(bias)) return -1;
  if ((double)(tile + 2 * apron) * (double)scale > 4096.0) return -1;
  const int tiles = (out_size + tile - 1) / tile;
  int best = 0;
  for (int t = 0; t < tiles; ++t) {
    const int o0 = std::max(t * tile - apron, -lo) + origin;  // `origin`: the image is a band of a larger output image
    const int ol = std::min(t * tile + tile - 1 + apron, out_size - 1 + hi) + origin;
    const float p0 = (float)o0 * scale, pl = (float)ol * scale;  // two roundings each, as on the device
    const int f0 = (int)std::floor(p0 + bias), fl = (int)std::floor(pl + bias);
    best = std::max(best, fl - f0 + 4);
  }
  return best;
}

// Launch grids are one-dimensional: tiles_x * tiles_y * frames workgroups, formed in int on the device.
static int check_grid(const char* who, int tiles_x, int tiles_y, int frames) {
  const long long total = (long long)tiles_x * (long long)tiles_y * (long long)frames;
  if (total <= 0 || total > 0x7fffffffLL) return fail(FSR1_ERR_UNSUPPORTED, "%s: %lld workgroups exceed the launch grid limit; split the batch", who, total);
  return FSR1_OK;
}

static const uint32_t kKnownFlags = FSR1_FLAG_HDR_SQUARE | FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA |
                                    FSR1_FLAG_MATH_EXACT | FSR1_FLAG_MATH_PACKED_FP16 | FSR1_FLAG_MATH_STRICT | FSR1_FLAG_NO_FAST_PATHS |
                                    FSR1_FLAG_OUTPUT_STREAMING | FSR1_FLAG_OUTPUT_CACHED | FSR1_FLAG_FRAMES_OVERLAP;

static int check_flags(uint32_t flags) {
  if (flags & ~kKnownFlags) return fail(FSR1_ERR_INVALID_ARGUMENT, "unknown flag bits 0x%x", flags & ~kKnownFlags);
  if ((flags & FSR1_FLAG_MATH_EXACT) && (flags & FSR1_FLAG_MATH_PACKED_FP16))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "FSR1_FLAG_MATH_EXACT and FSR1_FLAG_MATH_PACKED_FP16 are exclusive");
  if ((flags & FSR1_FLAG_MATH_STRICT) && (flags & (FSR1_FLAG_MATH_EXACT | FSR1_FLAG_MATH_PACKED_FP16)))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "FSR1_FLAG_MATH_STRICT is exclusive with FSR1_FLAG_MATH_EXACT and FSR1_FLAG_MATH_PACKED_FP16");
  if ((flags & FSR1_FLAG_OUTPUT_STREAMING) && (flags & FSR1_FLAG_OUTPUT_CACHED))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "FSR1_FLAG_OUTPUT_STREAMING and FSR1_FLAG_OUTPUT_CACHED are exclusive");
  return FSR1_OK;
}

// The store policy the kernels see: exactly one of the two bits, resolved against the pass's default.
static uint32_t resolve_output_policy(uint32_t flags, bool streaming_by_default) {
  const bool streaming = (flags & FSR1_FLAG_OUTPUT_STREAMING) || (streaming_by_default && !(flags & FSR1_FLAG_OUTPUT_CACHED));
  return (flags & ~(uint32_t)(FSR1_FLAG_OUTPUT_STREAMING | FSR1_FLAG_OUTPUT_CACHED)) | (streaming ? FSR1_FLAG_OUTPUT_STREAMING : 0u);
}

// fsr1_color_stages -> the device-side ColorArgs; `allowed` = the stage bits this entry point can run.
// Returns FSR1_OK with c->stages == 0 for a NULL / empty descriptor.
static const uint32_t kColorStageBits = FSR1_COLOR_SRTM | FSR1_COLOR_LFGA | FSR1_COLOR_SRTM_INV | FSR1_COLOR_TEPD_C8 |
                                        FSR1_COLOR_TEPD_C10 | FSR1_COLOR_DITHER_FROM_NOISE;
static int check_color(const fsr1_color_stages* st, const char* who, ColorArgs* c) {
  memset(c, 0, sizeof *c);
  if (!st || !st->stages) return FSR1_OK;
  if (st->stages & ~kColorStageBits) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: unknown colour stage bits 0x%x", who, st->stages & ~kColorStageBits);
  if ((st->stages & FSR1_COLOR_TEPD_C8) && (st->stages & FSR1_COLOR_TEPD_C10))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: FSR1_COLOR_TEPD_C8 and FSR1_COLOR_TEPD_C10 are exclusive", who);
  if ((st->stages & FSR1_COLOR_DITHER_FROM_NOISE) && !(st->stages & (FSR1_COLOR_TEPD_C8 | FSR1_COLOR_TEPD_C10)))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: FSR1_COLOR_DITHER_FROM_NOISE needs a TEPD stage", who);
  c->stages = st->stages;
  c->amount = st->grain_amount;
  c->bias = st->grain_bias;
  c->frame = st->frame;
  if (st->stages & (FSR1_COLOR_LFGA | FSR1_COLOR_DITHER_FROM_NOISE)) {
    if (!st->noise) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: FSR1_COLOR_LFGA / FSR1_COLOR_DITHER_FROM_NOISE need a noise image", who);
    ImageView nv;
    int rc = check_image(st->noise, "noise", &nv);
    if (rc) return rc;
    const long long slice = (long long)(st->frame % (uint32_t)st->noise->frames);
    c->noise.base = nv.base + slice * nv.frame_stride;
    c->noise.width = nv.width;
    c->noise.height = nv.height;
    c->noise.pitch = nv.pitch;
    c->noise.format = st->noise->format;
    c->noise.off_x = (int)((((long long)st->noise_offset_x % nv.width) + nv.width) % nv.width);
    c->noise.off_y = (int)((((long long)st->noise_offset_y % nv.height) + nv.height) % nv.height);
    c->noise.rcp_width = 1.0f / (float)nv.width;
    c->noise.rcp_height = 1.0f / (float)nv.height;
  }
  return FSR1_OK;
}

// Format pairs the colour variants of the EASU / RCAS / fused kernels are built for.
static bool color_format_pair_ok(int fin, int fout) {
  return fin == fout || (fin == FSR1_FORMAT_RGBA16F && (fout == FSR1_FORMAT_RGBA8_UNORM || fout == FSR1_FORMAT_R10G10B10A2_UNORM));
}

// Compute units of the current device (256 on MI355X; 304 on an MI300X, 32 in CPX partitions), read once per device: launch
// rules that count residencies (fsr1_fused_s2.hip) scale with it instead of assuming the part they were measured on.
static int device_cus() {
  constexpr int kDevices = 64;
  static std::atomic<int> cache[kDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < kDevices) {
    if (const int c = cache[dev].load(std::memory_order_relaxed); c > 0) return c;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
  if (dev >= 0 && dev < kDevices) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

// The last-level cache in front of HBM ("Infinity Cache", 256 MB on MI355X) as HIP reports it (hipDeviceAttributeL2CacheSize is the
// per-XCD L2; the memory-side cache has no attribute of its own in this runtime): a constant for this family, in ONE place — the batch
// split of fsr1_pipeline_upscale and the store policy of the EASU intermediary compare sizes with it.
static size_t infinity_cache_bytes() { return (size_t)256 << 20; }

// LDS bytes a CU offers its resident workgroups (160 KB on gfx950), read from the device once instead of assumed: the
// launch rules that count workgroups per CU by their LDS (easu_generic_tall_tiles) and the "does it fit at all" checks use it.
static size_t device_lds_per_cu() {
  constexpr int kDevices = 64;
  static std::atomic<int> cache[kDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 160 * 1024;
  if (dev >= 0 && dev < kDevices) {
    if (const int c = cache[dev].load(std::memory_order_relaxed); c > 0) return (size_t)c;
  }
  int bytes = 0;
  if (hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || bytes < 64 * 1024) bytes = 160 * 1024;
  if (dev >= 0 && dev < kDevices) cache[dev].store(bytes, std::memory_order_relaxed);
  return (size_t)bytes;
}

// log2 of the XCDs behind the current device, for the kernels' workgroup -> tile mapping (xcd_swizzle): an XCD of this family has 32 CUs
// (MI355X: 256 CUs = 8 XCDs in SPX mode; the DPX / QPX / CPX partition modes expose devices of 128 / 64 / 32 CUs).  A CU count that is
// not 32 x a power of two (another part) rounds down: any value only permutes the tiles.
static int device_xcd_shift() {
  const int xcds = device_cus() / 32;
  int shift = 0;
  while ((2 << shift) <= xcds && shift < 3) ++shift;
  return shift;
}

// fsr1_params.fused = 2 ("auto"), on round-2 measurements: at exactly 2x the quad form of the fused
// launch (fsr1_fused_s2.hip) beats the two dispatches at every size; elsewhere the fused launch pays off only where a frame
// is launch-bound (<= 3 Mpixel of output per launch).  The fused kernels' tile (one-pixel apron) needs more LDS than EASU's:
// where it does not fit a CU's 160 KiB (about 1.9x minification and beyond) auto keeps the two dispatches the caller
// supplied an intermediary for, instead of failing.
static bool auto_takes_fused(const fsr1_image* in, bool have_intermediary, const fsr1_image* out, const uint32_t easu_con[16], uint32_t math,
                             bool have_stages) {
  if (!have_intermediary) return true;
  const bool packed = (math & FSR1_FLAG_MATH_PACKED_FP16) != 0;
  const bool con_2x = easu_con[0] == 0x3f000000u && easu_con[1] == 0x3f000000u && easu_con[2] == 0xbe800000u && easu_con[3] == 0xbe800000u;
  const bool quad_form = con_2x && !(math & (FSR1_FLAG_MATH_PACKED_FP16 | FSR1_FLAG_NO_FAST_PATHS)) && !have_stages;
  if (!quad_form) {
    float sx, sy, bx, by;
    memcpy(&sx, &easu_con[0], 4);
    memcpy(&sy, &easu_con[1], 4);
    memcpy(&bx, &easu_con[2], 4);
    memcpy(&by, &easu_con[3], 4);
    const int fp_w = footprint_extent(out->width, kTileW, 1, sx, bx), fp_h = footprint_extent(out->height, kFusedTileH, 1, sy, by);
    if (fp_w < 0 || fp_h < 0) return false;
    if ((packed ? fused_h_lds_bytes(fp_w, fp_h) : fused_lds_bytes(in->format, fp_w, fp_h)) > device_lds_per_cu()) return false;
  }
  const long long out_pixels = (long long)out->width * (long long)out->height * (long long)out->frames;
  // Packed fp16 at exactly 2x has a quad-form fused launch of its own (fsr1_fused_s2_h.hip, round 4) that walks longer runs the larger
  // the launch is.  Measured against the two H dispatches (round 5, profiles/ab_r05/r5c4_ab_h_auto.log, r5c5_ab_h_auto2.log, us per
  // launch, two dispatches / fused): 540p -> 1080p 34.8 / 29.8, 720p -> 1440p 50.9 / 47.0, one 4K frame 93.7 / 95.6, four 4K frames or
  // one 8K frame 348 / 359, eight 4K frames 708 / 685, four 8K frames 1347 / 1345, eight 2690 / 2640, sixteen 5450 / 5240: the fused
  // launch up to 4 Mpixel of output and from 60 Mpixel up, the two dispatches between.
  const bool packed_2x = con_2x && packed && !(math & FSR1_FLAG_NO_FAST_PATHS) && !have_stages;
  if (packed_2x && (out_pixels <= 4000000ll || out_pixels >= 60000000ll)) return true;
  return quad_form || out_pixels <= 3000000ll;
}

extern "C" {

const char* fsr1_last_error(void) { return g_err; }
int fsr1_version(void) { return FSR1_HIP_VERSION; }

int fsr1_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
  return n;
}

int fsr1_easu_dispatch(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags,
                       void* stream) {
  return fsr1_easu_dispatch_ex(in, out, con, flags, nullptr, stream);
}

static int easu_dispatch_impl(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags,
                              const fsr1_color_stages* stages, int origin_x, int origin_y, void* stream);

int fsr1_easu_dispatch_ex(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags,
                          const fsr1_color_stages* stages, void* stream) {
  return easu_dispatch_impl(in, out, con, flags, stages, 0, 0, stream);
}

// A band of the output image (a single frame split over GPUs as row bands, zero exchange): `out` holds output
// pixels [origin_x, origin_x + out->width) x [origin_y, origin_y + out->height) of the image `con` was set up for.
int fsr1_easu_dispatch_band(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags, int32_t origin_x,
                            int32_t origin_y, void* stream) {
  if (origin_x < 0 || origin_y < 0 || origin_x > (1 << 20) || origin_y > (1 << 20))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "easu band: origin (%d, %d) out of range", origin_x, origin_y);
  if (flags & FSR1_FLAG_MATH_PACKED_FP16) return fail(FSR1_ERR_UNSUPPORTED, "easu band: bands run with the F (binary32) arithmetic");
  return easu_dispatch_impl(in, out, con, flags, nullptr, origin_x, origin_y, stream);
}

static int easu_dispatch_impl(const fsr1_image* in, const fsr1_image* out, const uint32_t con[16], uint32_t flags,
                              const fsr1_color_stages* stages, int origin_x, int origin_y, void* stream) {
  const Range range("easu", in, out);
  EasuArgs a;
  a.xcd_shift = device_xcd_shift();
  int rc;
  if ((rc = check_flags(flags))) return rc;
  if ((rc = check_image(in, "easu input", &a.in))) return rc;
  if ((rc = check_image(out, "easu output", &a.out))) return rc;
  if ((rc = check_color(stages, "easu", &a.color))) return rc;
  if (!con) return fail(FSR1_ERR_INVALID_ARGUMENT, "easu: null constants");
  if (a.color.stages ? !color_format_pair_ok(in->format, out->format) : in->format != out->format)
    return fail(FSR1_ERR_UNSUPPORTED, "easu: unsupported input/output format pair %d -> %d", in->format, out->format);
  if (a.color.stages && (flags & FSR1_FLAG_MATH_PACKED_FP16)) return fail(FSR1_ERR_UNSUPPORTED, "easu: colour stages run with the F (binary32) arithmetic");
  if (in->frames != out->frames) return fail(FSR1_ERR_INVALID_ARGUMENT, "easu: frame counts differ (%d vs %d)", in->frames, out->frames);
  if (overlaps(in, a.in, out, a.out)) return fail(FSR1_ERR_INVALID_ARGUMENT, "easu: input and output overlap");
  // F-strict promises EXACT's stored image: where there is no conversion to test against (RGBA32F stores the binary32 result itself)
  // or no strict variant is built (`c *= c`, colour stages), the EXACT kernels deliver it directly
  if ((flags & FSR1_FLAG_MATH_STRICT) && (in->format == FSR1_FORMAT_RGBA32F || (flags & FSR1_FLAG_HDR_SQUARE) || a.color.stages))
    flags = (flags & ~(uint32_t)FSR1_FLAG_MATH_STRICT) | FSR1_FLAG_MATH_EXACT;
  memcpy(a.con, con, sizeof a.con);
  float sx, sy, bx, by;
  memcpy(&sx, &con[0], 4);
  memcpy(&sy, &con[1], 4);
  memcpy(&bx, &con[2], 4);
  memcpy(&by, &con[3], 4);
  a.origin_x = origin_x;
  a.origin_y = origin_y;
  a.fp_w = footprint_extent(out->width, kTileW, 0, sx, bx, origin_x);
  a.fp_h = footprint_extent(out->height, kTileH, 0, sy, by, origin_y);
  if (a.fp_w < 0 || a.fp_h < 0) return fail(FSR1_ERR_INVALID_ARGUMENT, "easu: scale constants con0.xy = (%g, %g) are not usable", sx, sy);
  if (easu_lds_bytes(in->format, a.fp_w, a.fp_h) > device_lds_per_cu())
    return fail(FSR1_ERR_UNSUPPORTED, "easu: input/output ratio (%g, %g) needs a %dx%d texel footprint per tile, beyond the LDS budget "
                                      "(EASU is an upscaler; ratios up to ~3x minification are supported)", sx, sy, a.fp_w, a.fp_h);
  if ((flags & FSR1_FLAG_MATH_STRICT) && easu_strict_lds_bytes(in->format, a.fp_w, a.fp_h, kTileH) > device_lds_per_cu())  // (no room for the queue: EXACT)
    flags = (flags & ~(uint32_t)FSR1_FLAG_MATH_STRICT) | FSR1_FLAG_MATH_EXACT;
  const bool strict = (flags & FSR1_FLAG_MATH_STRICT) != 0;
  if ((long long)(a.fp_h + 1) * a.in.pitch >= (1ll << 31))  // staging addresses texels as row base + 32-bit offset
    return fail(FSR1_ERR_UNSUPPORTED, "easu: input row pitch %lld too large for a %d-row footprint", a.in.pitch, a.fp_h);
  a.tiles_x = (out->width + kTileW - 1) / kTileW;
  a.tiles_y = (out->height + kTileH - 1) / kTileH;
  a.frames = out->frames;
  // RCAS normally follows: keep the intermediary cached — unless it is far larger than the 256 MB Infinity Cache anyway (batches):
  // then plain stores only leave dirty lines for the kernel's end to write back (round 4, 8-frame 1440p -> 4K batch, 531 MB: two
  // dispatches 610-613 -> 600-605 us; one 4K frame, 66 MB, the other way: 65.8 -> 70.5; profiles/ab_r04/r4c17_easu_streaming_stores.log)
  const bool beyond_cache = (unsigned long long)a.out.pitch * (unsigned long long)out->height * (unsigned long long)out->frames > 2 * (unsigned long long)infinity_cache_bytes();
  a.flags = resolve_output_policy(flags, beyond_cache);
  // Exact 2x with the viewport covering the input — con0 = {1/2, 1/2, -1/4, -1/4}, what FsrEasuCon gives for
  // out = 2 * in — takes the variant whose lanes own 2x2 output quads; its tiles are shifted by one pixel, hence one
  // more tile per axis when the size is a multiple of the tile, and the footprint of a tile is 64/2+3 x 16/2+3 texels.
  const bool s2 = con[0] == 0x3f000000u && con[1] == 0x3f000000u && con[2] == 0xbe800000u && con[3] == 0xbe800000u &&
                  !(flags & FSR1_FLAG_NO_FAST_PATHS) && !a.color.stages && kTileH % 16 == 0 && !((origin_x | origin_y) & 1);
  // (64 x 32 tiles for whole-image F launches that are large or overlapped: easu_s2_tall_tiles)
  bool tall = s2 && !(flags & (FSR1_FLAG_MATH_PACKED_FP16 | FSR1_FLAG_MATH_EXACT)) && !origin_x && !origin_y &&
              easu_s2_tall_tiles(out->width, out->height, out->frames, (flags & FSR1_FLAG_FRAMES_OVERLAP) != 0, device_cus());
  // Any other ratio, default arithmetic, plain pass, whole image: the generic kernel on 64 x 32 tiles with 512-thread workgroups when
  // its pitched layout applies and the launch is large enough (easu_generic_tall_tiles) — the footprint is re-derived for the 32-row tile
  if (!s2 && !(flags & (FSR1_FLAG_MATH_PACKED_FP16 | FSR1_FLAG_MATH_EXACT)) && !a.color.stages && !origin_x && !origin_y) {
    const int fp_h32 = footprint_extent(out->height, 2 * kTileH, 0, sy, by);
    const int pitch = easu_lds_pitch(a.fp_w, false, false);
    if (fp_h32 > 0 && pitch &&
        easu_generic_tall_tiles(out->width, out->height, out->frames, device_cus(),
                                strict ? easu_strict_lds_bytes(in->format, pitch, fp_h32, 2 * kTileH) : easu_lds_bytes(in->format, pitch, fp_h32), device_lds_per_cu()) &&
        (long long)(fp_h32 + 1) * a.in.pitch < (1ll << 31)) {
      tall = true;
      a.fp_h = fp_h32;
      a.tiles_y = (out->height + 2 * kTileH - 1) / (2 * kTileH);
    }
  }
  if (s2) {
    const int th = tall ? 2 * kTileH : kTileH;
    a.tiles_x = (out->width + 1 + kTileW - 1) / kTileW;
    a.tiles_y = (out->height + 1 + th - 1) / th;
    a.fp_w = kTileW / 2 + 3;
    a.fp_h = th / 2 + 3;
  }
  if ((rc = check_grid("easu", a.tiles_x, a.tiles_y, a.frames))) return rc;
  hipError_t e;
  if (flags & FSR1_FLAG_MATH_PACKED_FP16) {
    if (in->format != FSR1_FORMAT_RGBA16F) return fail(FSR1_ERR_UNSUPPORTED, "easu: packed-fp16 math needs RGBA16F images");
    e = easu_h_launch(a, s2, static_cast<hipStream_t>(stream));
  } else if (a.color.stages) {
    e = easu_color_launch(a, in->format, out->format, (flags & FSR1_FLAG_MATH_EXACT) != 0, static_cast<hipStream_t>(stream));
  } else {
    e = easu_launch(a, in->format, (flags & FSR1_FLAG_MATH_EXACT) != 0, s2, tall, static_cast<hipStream_t>(stream), strict);
  }
  if (e != hipSuccess) return hip_fail(e, "easu launch");
  return FSR1_OK;
}

int fsr1_rcas_dispatch(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags, void* stream) {
  return fsr1_rcas_dispatch_ex(in, out, con, flags, nullptr, stream);
}

static int rcas_dispatch_impl(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags,
                              const fsr1_color_stages* stages, int rows_above, int rows_below, void* stream);

int fsr1_rcas_dispatch_ex(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags,
                          const fsr1_color_stages* stages, void* stream) {
  return rcas_dispatch_impl(in, out, con, flags, stages, 0, 0, stream);
}

// RCAS on a band of a larger image: `in` describes the band's own rows, and rows_above / rows_below (0 or 1) say whether the
// row just above `in`'s first row / just below its last row exists IN MEMORY (same pitch) as part of the larger image —
// those taps are then read instead of being 0.  Columns are the full image width.
int fsr1_rcas_dispatch_band(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags, int32_t rows_above,
                            int32_t rows_below, void* stream) {
  if ((rows_above | rows_below) & ~1) return fail(FSR1_ERR_INVALID_ARGUMENT, "rcas band: rows_above / rows_below must be 0 or 1");
  if (flags & FSR1_FLAG_MATH_PACKED_FP16) return fail(FSR1_ERR_UNSUPPORTED, "rcas band: bands run with the F (binary32) arithmetic");
  return rcas_dispatch_impl(in, out, con, flags, nullptr, rows_above, rows_below, stream);
}

static int rcas_dispatch_impl(const fsr1_image* in, const fsr1_image* out, const uint32_t con[4], uint32_t flags,
                              const fsr1_color_stages* stages, int rows_above, int rows_below, void* stream) {
  const Range range("rcas", in, out);
  RcasArgs a;
  a.xcd_shift = device_xcd_shift();
  int rc;
  if ((rc = check_flags(flags))) return rc;
  // F-strict is EASU's property: RCAS runs the default arithmetic under it — except where EASU has no strict variant and runs EXACT
  // (RGBA32F storage, colour stages): then RCAS does too, so that the fused launch, which cannot mix arithmetics there, stays
  // bit-identical to the two dispatches (include/fsr1_hip.h)
  if (flags & FSR1_FLAG_MATH_STRICT)
    flags = (flags & ~(uint32_t)FSR1_FLAG_MATH_STRICT) | ((in && (in->format == FSR1_FORMAT_RGBA32F || (stages && stages->stages))) ? (uint32_t)FSR1_FLAG_MATH_EXACT : 0u);
  if ((rc = check_image(in, "rcas input", &a.in))) return rc;
  if ((rc = check_image(out, "rcas output", &a.out))) return rc;
  if ((rc = check_color(stages, "rcas", &a.color))) return rc;
  if (!con) return fail(FSR1_ERR_INVALID_ARGUMENT, "rcas: null constants");
  if (a.color.stages ? !color_format_pair_ok(in->format, out->format) : in->format != out->format)
    return fail(FSR1_ERR_UNSUPPORTED, "rcas: unsupported input/output format pair %d -> %d", in->format, out->format);
  if (a.color.stages && (flags & FSR1_FLAG_MATH_PACKED_FP16)) return fail(FSR1_ERR_UNSUPPORTED, "rcas: colour stages run with the F (binary32) arithmetic");
  if (in->width != out->width || in->height != out->height || in->frames != out->frames)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "rcas: input %dx%dx%d and output %dx%dx%d extents differ", in->width, in->height,
                in->frames, out->width, out->height, out->frames);
  {
    fsr1_image in_ext = *in;  // what the pass reads: the band plus the neighbouring rows it was told exist
    ImageView v_ext = a.in;
    in_ext.height += rows_above + rows_below;
    v_ext.base -= (long long)rows_above * a.in.pitch;
    if (overlaps(&in_ext, v_ext, out, a.out)) return fail(FSR1_ERR_INVALID_ARGUMENT, "rcas: input and output overlap (RCAS cannot run in place)");
  }
  if ((rows_above || rows_below) && in->frames != 1) return fail(FSR1_ERR_UNSUPPORTED, "rcas band: one frame per dispatch");
  memcpy(a.con, con, sizeof a.con);
  rcas_geometry(out->width, out->height, out->frames, (flags & FSR1_FLAG_FRAMES_OVERLAP) != 0, &a.tiles_x, &a.tiles_y, &a.rows);
  a.frames = out->frames;
  if ((rc = check_grid("rcas", a.tiles_x, a.tiles_y, a.frames))) return rc;
  a.flags = resolve_output_policy(flags, true);
  a.rows_above = rows_above;
  a.rows_below = rows_below;
  hipError_t e;
  if (flags & FSR1_FLAG_MATH_PACKED_FP16) {
    if (in->format != FSR1_FORMAT_RGBA16F) return fail(FSR1_ERR_UNSUPPORTED, "rcas: packed-fp16 math needs RGBA16F images");
    e = rcas_h_launch(a, static_cast<hipStream_t>(stream));
  } else if (a.color.stages) {
    e = rcas_color_launch(a, in->format, out->format, (flags & FSR1_FLAG_MATH_EXACT) != 0, static_cast<hipStream_t>(stream));
  } else {
    e = rcas_launch(a, in->format, (flags & FSR1_FLAG_MATH_EXACT) != 0, static_cast<hipStream_t>(stream));
  }
  if (e != hipSuccess) return hip_fail(e, "rcas launch");
  return FSR1_OK;
}

int fsr1_easu_rcas_fused_dispatch(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16],
                                  const uint32_t rcas_con[4], uint32_t flags, void* stream) {
  return fsr1_easu_rcas_fused_dispatch_ex(in, out, easu_con, rcas_con, flags, nullptr, stream);
}

static int fused_dispatch_impl(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16], const uint32_t rcas_con[4],
                               uint32_t flags, const fsr1_color_stages* stages, int origin_y, int rows_above, int rows_below, void* stream);

int fsr1_easu_rcas_fused_dispatch_ex(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16],
                                     const uint32_t rcas_con[4], uint32_t flags, const fsr1_color_stages* stages,
                                     void* stream) {
  return fused_dispatch_impl(in, out, easu_con, rcas_con, flags, stages, 0, 0, 0, stream);
}

// The single launch on a band of output rows: `out` holds rows [origin_y, origin_y + out->height) of the image
// `easu_con` was set up for; where that image has a row above / below the band (rows_above / rows_below = 1) the tile aprons
// compute it with EASU instead of taking it as outside the image, so the band equals the same rows of a full-frame launch.
int fsr1_easu_rcas_fused_dispatch_band(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16],
                                       const uint32_t rcas_con[4], uint32_t flags, int32_t origin_y, int32_t rows_above,
                                       int32_t rows_below, void* stream) {
  if (origin_y < 0 || origin_y > (1 << 20)) return fail(FSR1_ERR_INVALID_ARGUMENT, "fused band: origin_y %d out of range", origin_y);
  if ((rows_above | rows_below) & ~1) return fail(FSR1_ERR_INVALID_ARGUMENT, "fused band: rows_above / rows_below must be 0 or 1");
  if (rows_above && origin_y < 1) return fail(FSR1_ERR_INVALID_ARGUMENT, "fused band: rows_above = 1 needs origin_y >= 1");
  if (flags & FSR1_FLAG_MATH_PACKED_FP16) return fail(FSR1_ERR_UNSUPPORTED, "fused band: bands run with the F (binary32) arithmetic");
  return fused_dispatch_impl(in, out, easu_con, rcas_con, flags, nullptr, origin_y, rows_above, rows_below, stream);
}

static int fused_dispatch_impl(const fsr1_image* in, const fsr1_image* out, const uint32_t easu_con[16], const uint32_t rcas_con[4],
                               uint32_t flags, const fsr1_color_stages* stages, int origin_y, int rows_above, int rows_below, void* stream) {
  const Range range("easu_rcas_fused", in, out);
  FusedArgs a;
  a.xcd_shift = device_xcd_shift();
  int rc;
  if ((rc = check_flags(flags))) return rc;
  if ((rc = check_image(in, "fused input", &a.in))) return rc;
  if ((rc = check_image(out, "fused output", &a.out))) return rc;
  if ((rc = check_color(stages, "fused", &a.color))) return rc;
  if (!easu_con || !rcas_con) return fail(FSR1_ERR_INVALID_ARGUMENT, "fused: null constants");
  if (a.color.stages ? !color_format_pair_ok(in->format, out->format) : in->format != out->format)
    return fail(FSR1_ERR_UNSUPPORTED, "fused: unsupported input/output format pair %d -> %d", in->format, out->format);
  if (in->frames != out->frames) return fail(FSR1_ERR_INVALID_ARGUMENT, "fused: frame counts differ (%d vs %d)", in->frames, out->frames);
  if (overlaps(in, a.in, out, a.out)) return fail(FSR1_ERR_INVALID_ARGUMENT, "fused: input and output overlap");
  // F-strict: EASU's half bit-identical to EXACT's, RCAS's half in the default arithmetic.  Where EASU has no strict variant (RGBA32F:
  // no store conversion to test against; colour stages) BOTH halves run EXACT, as the two dispatches do under the flag
  if ((flags & FSR1_FLAG_MATH_STRICT) && (in->format == FSR1_FORMAT_RGBA32F || a.color.stages))
    flags = (flags & ~(uint32_t)FSR1_FLAG_MATH_STRICT) | FSR1_FLAG_MATH_EXACT;
  const bool packed = (flags & FSR1_FLAG_MATH_PACKED_FP16) != 0;
  if (packed && (in->format != FSR1_FORMAT_RGBA16F || a.color.stages))
    return fail(FSR1_ERR_UNSUPPORTED, "fused: packed-fp16 math needs RGBA16F images and runs without colour stages");
  memcpy(a.easu_con, easu_con, sizeof a.easu_con);
  memcpy(a.rcas_con, rcas_con, sizeof a.rcas_con);
  float sx, sy, bx, by;
  memcpy(&sx, &easu_con[0], 4);
  memcpy(&sy, &easu_con[1], 4);
  memcpy(&bx, &easu_con[2], 4);
  memcpy(&by, &easu_con[3], 4);
  a.fp_w = footprint_extent(out->width, kTileW, 1, sx, bx);
  a.fp_h = footprint_extent(out->height, kFusedTileH, 1, sy, by, origin_y, rows_above, rows_below);
  a.origin_y = origin_y;
  a.rows_above = rows_above;
  a.rows_below = rows_below;
  if (a.fp_w < 0 || a.fp_h < 0) return fail(FSR1_ERR_INVALID_ARGUMENT, "fused: scale constants con0.xy = (%g, %g) are not usable", sx, sy);
  const bool strict = (flags & FSR1_FLAG_MATH_STRICT) != 0;
  if ((packed ? fused_h_lds_bytes(a.fp_w, a.fp_h) : strict ? fused_strict_lds_bytes(in->format, a.fp_w, a.fp_h) : fused_lds_bytes(in->format, a.fp_w, a.fp_h)) > device_lds_per_cu())
    return fail(FSR1_ERR_UNSUPPORTED, "fused: input/output ratio (%g, %g) needs more LDS than a CU has", sx, sy);
  if ((long long)(a.fp_h + 1) * a.in.pitch >= (1ll << 31))
    return fail(FSR1_ERR_UNSUPPORTED, "fused: input row pitch %lld too large for a %d-row footprint", a.in.pitch, a.fp_h);
  a.tiles_x = (out->width + kTileW - 1) / kTileW;
  a.tiles_y = (out->height + kFusedTileH - 1) / kFusedTileH;
  a.frames = out->frames;
  // Exact 2x (con0 = {1/2, 1/2, -1/4, -1/4}), plain F arithmetic, a whole image or a band from an even row: the variant whose lanes own 2x2 quads
  // of the apron tile (fsr1_fused_s2.hip); its tiles are 62 pixels wide and 2 QH - 2 tall.
  // (packed fp16: the H twin, fsr1_fused_s2_h.hip — whole images only, five workgroups per CU)
  const bool s2 = easu_con[0] == 0x3f000000u && easu_con[1] == 0x3f000000u && easu_con[2] == 0xbe800000u && easu_con[3] == 0xbe800000u &&
                  !(flags & FSR1_FLAG_NO_FAST_PATHS) && !a.color.stages && !(origin_y & 1) &&
                  (packed ? origin_y == 0 && !rows_above && !rows_below : fused_s2_lds_bytes(in->format, 4) <= device_lds_per_cu());
  a.run_steps = s2 ? fused_s2_run_steps(out->width, out->height, out->frames, device_cus(), packed ? 5 : 7, (flags & FSR1_FLAG_FRAMES_OVERLAP) != 0, strict) : 0;
  // F-strict: every step ends with the re-evaluation of its queued pixels by the workgroup's first lanes while the other waves wait at
  // the barrier in front of the RCAS phase — the fewer waves wait and the fewer steps serialise it, the better: 256-thread tiles, runs
  // of at most two steps (fused_s2_run_steps) (one 4K frame alone, us: tall tile 77.2, one-step 74.1, 2 steps 74.5, 4 steps 78.3; profiles/ab_r06/r6c5)
  const bool tall = s2 && !packed && !strict && fused_s2_tall_tiles(out->width, out->height, out->frames, a.run_steps, device_cus(), in->format);
  if (s2) fused_s2_geometry(out->width, out->height, a.run_steps, &a.tiles_x, &a.tiles_y, tall ? 2 * kFs2Step : kFs2Step);
  if ((rc = check_grid("fused", a.tiles_x, a.tiles_y, a.frames))) return rc;
  a.flags = resolve_output_policy(flags, true);
  const bool exact = (flags & FSR1_FLAG_MATH_EXACT) != 0;
  hipError_t e = packed ? (s2 ? fused_s2_h_launch(a, static_cast<hipStream_t>(stream)) : fused_h_launch(a, static_cast<hipStream_t>(stream)))
                 : a.color.stages ? fused_color_launch(a, in->format, out->format, exact, static_cast<hipStream_t>(stream))
                 : s2             ? fused_s2_launch(a, in->format, exact, tall, static_cast<hipStream_t>(stream), strict)
                                  : fused_launch(a, in->format, exact, static_cast<hipStream_t>(stream), strict);
  if (e != hipSuccess) return hip_fail(e, "fused launch");
  return FSR1_OK;
}

// Stand-alone colour pass (ffx_fsr1.h:986-1199).
int fsr1_color_dispatch(const fsr1_image* in, const fsr1_image* out, const fsr1_color_stages* stages, uint32_t flags,
                        void* stream) {
  const Range range("color", in, out);
  ColorPassArgs a;
  a.xcd_shift = device_xcd_shift();
  int rc;
  if (flags & ~(uint32_t)(FSR1_FLAG_MATH_EXACT | FSR1_FLAG_MATH_PACKED_FP16))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "color: flags may only hold FSR1_FLAG_MATH_EXACT or FSR1_FLAG_MATH_PACKED_FP16");
  if ((rc = check_flags(flags))) return rc;
  if ((rc = check_image(in, "color input", &a.in))) return rc;
  if ((rc = check_image(out, "color output", &a.out))) return rc;
  if ((rc = check_color(stages, "color", &a.color))) return rc;
  if (in->width != out->width || in->height != out->height || in->frames != out->frames)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "color: input %dx%dx%d and output %dx%dx%d extents differ", in->width, in->height,
                in->frames, out->width, out->height, out->frames);
  if (overlaps(in, a.in, out, a.out) &&
      !(in->data == out->data && in->format == out->format && a.in.pitch == a.out.pitch && a.in.frame_stride == a.out.frame_stride))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "color: input and output overlap without being the same image");
  color_geometry(out->width, out->height, &a.tiles_x, &a.tiles_y);
  a.frames = out->frames;
  if ((rc = check_grid("color", a.tiles_x, a.tiles_y, a.frames))) return rc;
  hipError_t e;
  if (flags & FSR1_FLAG_MATH_PACKED_FP16) {
    if (in->format != FSR1_FORMAT_RGBA16F || out->format != FSR1_FORMAT_RGBA16F)
      return fail(FSR1_ERR_UNSUPPORTED, "color: packed-fp16 math needs RGBA16F images");
    e = color_h_launch(a, static_cast<hipStream_t>(stream));
  } else {
    e = color_launch(a, in->format, out->format, (flags & FSR1_FLAG_MATH_EXACT) != 0, static_cast<hipStream_t>(stream));
  }
  if (e != hipSuccess) return hip_fail(e, "color launch");
  return FSR1_OK;
}

// FSR_Filter::Upscale, sample/src/DX12/FSR_Filter.cpp:101-141.
int fsr1_upscale(const fsr1_image* in, const fsr1_image* intermediary, const fsr1_image* out, const fsr1_params* p,
                 void* stream) {
  return fsr1_upscale_ex(in, intermediary, out, p, nullptr, stream);
}

// Argument validation and the pipeline decision of fsr1_upscale_ex, shared with fsr1_upscale_plan so that the plan is by
// construction what the call does (and refuses what the call refuses).  `in` / `out` are checked as descriptors only (extents,
// formats, frame counts): the plan is asked without device pointers.
struct UpscalePlan {
  int pipeline;  // 0 = EASU + RCAS as two dispatches, 1 = the fused launch, 2 = EASU only
  uint32_t easu_con[16], rcas_con[4];
  uint32_t math, rcas_flags, out_policy;
};

static int upscale_decide(const char* who, const fsr1_image* in, bool have_intermediary, const fsr1_image* out, const fsr1_params* p,
                          bool have_stages, UpscalePlan* plan) {
  if (!in || !out || !p) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: null argument", who);
  for (const fsr1_image* im : {in, out}) {
    if (im->width <= 0 || im->height <= 0 || im->frames <= 0)
      return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: bad %s extent %dx%dx%d", who, im == in ? "input" : "output", im->width, im->height, im->frames);
    if (im->format < FSR1_FORMAT_RGBA16F || im->format > FSR1_FORMAT_R10G10B10A2_UNORM)
      return fail(FSR1_ERR_UNSUPPORTED, "%s: unsupported %s format %d", who, im == in ? "input" : "output", im->format);
  }
  if (in->frames != out->frames) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: frame counts differ (%d vs %d)", who, in->frames, out->frames);
  if (!(p->render_width >= 1.0f) || !(p->render_height >= 1.0f) || p->render_width > (float)in->width || p->render_height > (float)in->height)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: render size %gx%g must lie within the input image %dx%d", who, (double)p->render_width,
                (double)p->render_height, in->width, in->height);
  int rc;
  if ((rc = check_flags(p->flags))) return rc;
  plan->math = p->flags & (FSR1_FLAG_MATH_EXACT | FSR1_FLAG_MATH_PACKED_FP16 | FSR1_FLAG_MATH_STRICT | FSR1_FLAG_NO_FAST_PATHS);
  const uint32_t rcas_opts = p->flags & (FSR1_FLAG_RCAS_DENOISE | FSR1_FLAG_RCAS_PASSTHROUGH_ALPHA);
  plan->out_policy = p->flags & (FSR1_FLAG_OUTPUT_STREAMING | FSR1_FLAG_OUTPUT_CACHED);  // of the pass that writes `out`
  const uint32_t overlap = p->flags & FSR1_FLAG_FRAMES_OVERLAP;  // a scheduling hint: travels with the math bits to every pass
  if (p->flags & ~(plan->math | rcas_opts | plan->out_policy | overlap))
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: params.flags may only hold MATH_*, RCAS_*, OUTPUT_* and FRAMES_OVERLAP bits", who);
  plan->math |= overlap;
  if ((plan->math & FSR1_FLAG_MATH_PACKED_FP16) && (in->format != FSR1_FORMAT_RGBA16F || out->format != FSR1_FORMAT_RGBA16F || have_stages))
    return fail(FSR1_ERR_UNSUPPORTED, "%s: packed-fp16 math needs RGBA16F images and runs without colour stages", who);
  // :106 — viewport == input resource size == (renderWidth, renderHeight); output = display size
  FsrEasuCon(plan->easu_con, plan->easu_con + 4, plan->easu_con + 8, plan->easu_con + 12, p->render_width, p->render_height, p->render_width,
             p->render_height, (float)out->width, (float)out->height);
  float sx, sy, bx, by;
  memcpy(&sx, &plan->easu_con[0], 4);
  memcpy(&sy, &plan->easu_con[1], 4);
  memcpy(&bx, &plan->easu_con[2], 4);
  memcpy(&by, &plan->easu_con[3], 4);
  const int fp_w = footprint_extent(out->width, kTileW, 0, sx, bx), fp_h = footprint_extent(out->height, kTileH, 0, sy, by);
  if (fp_w < 0 || fp_h < 0) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: scale constants con0.xy = (%g, %g) are not usable", who, sx, sy);
  const bool packed = (plan->math & FSR1_FLAG_MATH_PACKED_FP16) != 0;
  if (!p->use_rcas || p->fused != 1) {  // an EASU dispatch will run (alone, or as the first of two, or as what `auto` may pick)
    if (easu_lds_bytes(in->format, fp_w, fp_h) > device_lds_per_cu())
      return fail(FSR1_ERR_UNSUPPORTED, "%s: input/output ratio (%g, %g) needs a %dx%d texel footprint per tile, beyond the LDS budget", who, sx, sy, fp_w, fp_h);
  }
  if (!p->use_rcas) {
    plan->pipeline = 2;
    return FSR1_OK;
  }
  FsrRcasCon(plan->rcas_con, p->rcas_attenuation);  // :124
  plan->rcas_flags = plan->math | rcas_opts | plan->out_policy | (p->hdr ? FSR1_FLAG_HDR_SQUARE : 0u);  // :125 Sample.x = hdr
  if (p->fused != 0 && p->fused != 1 && p->fused != 2) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: params.fused must be 0, 1 or 2 (auto)", who);
  const bool fused = p->fused == 1 || (p->fused == 2 && auto_takes_fused(in, have_intermediary, out, plan->easu_con, plan->math, have_stages));
  if (!fused && !have_intermediary) return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: two-pass EASU+RCAS needs an intermediary image", who);
  if (fused) {  // the fused launch's own LDS bound (fused_dispatch_impl)
    const int ffw = footprint_extent(out->width, kTileW, 1, sx, bx), ffh = footprint_extent(out->height, kFusedTileH, 1, sy, by);
    if (ffw < 0 || ffh < 0 || (packed ? fused_h_lds_bytes(ffw, ffh) : fused_lds_bytes(in->format, ffw, ffh)) > device_lds_per_cu())
      return fail(FSR1_ERR_UNSUPPORTED, "%s: input/output ratio (%g, %g) needs more LDS than a CU has for the fused launch", who, sx, sy);
  }
  plan->pipeline = fused ? 1 : 0;
  return FSR1_OK;
}

int fsr1_upscale_ex(const fsr1_image* in, const fsr1_image* intermediary, const fsr1_image* out, const fsr1_params* p,
                    const fsr1_color_stages* stages, void* stream) {
  const Range range("upscale", in, out);
  if (stages && !stages->stages) stages = nullptr;
  UpscalePlan plan;
  if (int rc = upscale_decide("upscale", in, intermediary != nullptr, out, p, stages != nullptr, &plan)) return rc;
  if (plan.pipeline == 2) {
    // :107 Sample.x = hdr && !bUseRcas ; :140 EASU straight into the output
    return fsr1_easu_dispatch_ex(in, out, plan.easu_con,
                                 plan.math | (p->hdr ? FSR1_FLAG_HDR_SQUARE : 0u) | (plan.out_policy ? plan.out_policy : (uint32_t)FSR1_FLAG_OUTPUT_STREAMING),
                                 stages, stream);
  }
  if (plan.pipeline == 1) return fsr1_easu_rcas_fused_dispatch_ex(in, out, plan.easu_con, plan.rcas_con, plan.rcas_flags, stages, stream);
  // two dispatches: the prologue belongs to EASU's loads, the epilogue to RCAS's stores
  fsr1_color_stages pre, post;
  const fsr1_color_stages* pre_p = nullptr;
  const fsr1_color_stages* post_p = nullptr;
  if (stages) {
    pre = *stages;
    post = *stages;
    pre.stages = stages->stages & (uint32_t)FSR1_COLOR_SRTM;
    post.stages = stages->stages & ~(uint32_t)FSR1_COLOR_SRTM;
    pre.noise = nullptr;
    if (pre.stages) pre_p = &pre;
    if (post.stages) post_p = &post;
  }
  int rc = fsr1_easu_dispatch_ex(in, intermediary, plan.easu_con, plan.math, pre_p, stream);  // :121 (Sample.x = 0 when RCAS follows)
  if (rc) return rc;
  // :130 the UAV->SRV barrier is stream order here
  return fsr1_rcas_dispatch_ex(intermediary, out, plan.rcas_con, plan.rcas_flags, post_p, stream);  // :131
}

// Which pipeline fsr1_upscale[_ex] runs for these arguments (no launch): 0 = EASU + RCAS as two dispatches, 1 = the single
// fused launch, 2 = EASU only (use_rcas == 0); the same validation as the call itself (upscale_decide).
int fsr1_upscale_plan(const fsr1_image* in, int32_t have_intermediary, const fsr1_image* out, const fsr1_params* p, int32_t have_stages) {
  UpscalePlan plan;
  if (int rc = upscale_decide("upscale_plan", in, have_intermediary != 0, out, p, have_stages != 0, &plan)) return rc;
  return plan.pipeline;
}

int fsr1_selftest(uint32_t* failures) {
  if (!failures) return fail(FSR1_ERR_INVALID_ARGUMENT, "selftest: null");
  uint32_t* d = nullptr;
  hipError_t e = hipMalloc(&d, sizeof(uint32_t));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc");
  e = hipMemset(d, 0, sizeof(uint32_t));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(selftest_kernel, dim3(256), dim3(256), 0, nullptr, d);
    hipLaunchKernelGGL(selftest_rcp_kernel, dim3(256 * 32), dim3(256), 0, nullptr, d);  // all 2^32 operands: well under a second
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(failures, d, sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return hip_fail(e, "selftest");
  return FSR1_OK;
}

// ---- frame pipeline: independent frames on alternating HIP streams (include/fsr1_hip.h) ----
struct fsr1_pipeline {
  static constexpr int kMax = 8;
  int n = 0, next = 0, device = 0;
  hipStream_t streams[kMax] = {};
  hipEvent_t done[kMax] = {};   // recorded after a slot's latest submission (join)
  hipEvent_t fork_ev = nullptr;
  void* mid[kMax] = {};         // per-slot EASU -> RCAS intermediary, tightly packed, grown on demand
  size_t mid_bytes[kMax] = {};
  bool stream_ordered_alloc = true;  // hipMallocAsync / hipFreeAsync available (hipDeviceAttributeMemoryPoolsSupported)
};

int fsr1_pipeline_create(fsr1_pipeline** out, int32_t streams) {
  if (!out) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_create: null");
  if (streams < 1 || streams > fsr1_pipeline::kMax) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_create: streams must be 1 .. %d", fsr1_pipeline::kMax);
  fsr1_pipeline* p = new fsr1_pipeline;
  hipError_t e = hipGetDevice(&p->device);
  if (e == hipSuccess) {  // stream-ordered growth of the intermediaries needs the device's memory pools; without them: hipMalloc + a stream sync
    int pools = 0;
    p->stream_ordered_alloc = hipDeviceGetAttribute(&pools, hipDeviceAttributeMemoryPoolsSupported, p->device) == hipSuccess && pools != 0;
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->fork_ev, hipEventDisableTiming);
  for (int i = 0; i < streams && e == hipSuccess; ++i) {
    e = hipStreamCreateWithFlags(&p->streams[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming);
    if (e == hipSuccess) p->n = i + 1;
  }
  if (e != hipSuccess) {
    const int rc = hip_fail(e, "pipeline_create");
    (void)fsr1_pipeline_destroy(p);
    return rc;
  }
  *out = p;
  return FSR1_OK;
}

// Grows slot `slot`'s intermediary to at least `need` bytes, in stream order on the slot's own stream (hipFreeAsync / hipMallocAsync:
// no host block, no device-wide synchronisation; the old buffer is released after the slot's last submission, the new one exists
// before its next).  A hipGraph captured EARLIER keeps the old pointer: any later growth invalidates it (include/fsr1_hip.h) — reserve the
// largest size before the first capture.  Refused with a clear error while the stream is being captured into a hipGraph: an allocation node inside a
// captured frame is not what the caller wants replayed — fsr1_pipeline_reserve() before the capture instead.
static int pipeline_grow(fsr1_pipeline* p, int slot, size_t need, const char* who) {
  if (p->mid_bytes[slot] >= need) return FSR1_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipError_t e = hipStreamIsCapturing(p->streams[slot], &cap); e != hipSuccess) return hip_fail(e, "hipStreamIsCapturing");
  if (cap != hipStreamCaptureStatusNone)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "%s: stream %d's intermediary (%zu bytes) must grow to %zu bytes while the stream is being captured; "
                "call fsr1_pipeline_reserve(pipeline, %zu) before the capture begins", who, slot, p->mid_bytes[slot], need, need);
  if (!p->stream_ordered_alloc) {
    // no memory pools on this device: the slot's stream drains, then a plain free / allocation (a host block, on growth only)
    if (hipError_t e = hipStreamSynchronize(p->streams[slot]); e != hipSuccess) return hip_fail(e, "hipStreamSynchronize before growing the intermediary");
    if (p->mid[slot]) (void)hipFree(p->mid[slot]);
    p->mid[slot] = nullptr;
    p->mid_bytes[slot] = 0;
    if (hipError_t e = hipMalloc(&p->mid[slot], need); e != hipSuccess) return hip_fail(e, "hipMalloc of the intermediary");
    p->mid_bytes[slot] = need;
    return FSR1_OK;
  }
  if (p->mid[slot]) {
    if (hipError_t e = hipFreeAsync(p->mid[slot], p->streams[slot]); e != hipSuccess) return hip_fail(e, "hipFreeAsync of the intermediary");
    p->mid[slot] = nullptr;
    p->mid_bytes[slot] = 0;
  }
  if (hipError_t e = hipMallocAsync(&p->mid[slot], need, p->streams[slot]); e != hipSuccess) return hip_fail(e, "hipMallocAsync of the intermediary");
  p->mid_bytes[slot] = need;
  return FSR1_OK;
}

int fsr1_pipeline_reserve(fsr1_pipeline* p, size_t bytes_per_stream) {
  if (!p || p->n < 1) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_reserve: null pipeline");
  if (int dev = -1; hipGetDevice(&dev) != hipSuccess || dev != p->device)
    return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_reserve: the pipeline belongs to device %d, the calling thread's current device is %d", p->device, dev);
  for (int i = 0; i < p->n; ++i)
    if (int rc = pipeline_grow(p, i, bytes_per_stream, "pipeline_reserve")) return rc;
  return FSR1_OK;
}

int fsr1_pipeline_upscale(fsr1_pipeline* p, const fsr1_image* in, const fsr1_image* out, const fsr1_params* params,
                          const fsr1_color_stages* stages) {
  if (!p || p->n < 1) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_upscale: null pipeline");
  if (int dev = -1; hipGetDevice(&dev) != hipSuccess || dev != p->device)  // its streams and intermediaries live on the device it was created on
    return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_upscale: the pipeline belongs to device %d, the calling thread's current device is %d", p->device, dev);
  if (stages && !stages->stages) stages = nullptr;
  UpscalePlan plan;
  if (int rc = upscale_decide("pipeline_upscale", in, true, out, params, stages != nullptr, &plan)) return rc;
  // A batch whose frames go through two dispatches is submitted FRAME BY FRAME over the pipeline's slots when the per-frame
  // intermediaries of all slots fit the 256 MB Infinity Cache together (three 4K RGBA16F frames: 199 MB): every frame's RCAS then
  // reads its intermediary from the cache, and frame f's RCAS runs beside frame f + 1's EASU, where one launch pair for the whole
  // batch sends a 531 MB intermediary through HBM and serialises the two passes (round 5, 8-frame 2560x1440 -> 3840x2160 batch,
  // profiles/ab_r05/r5c8_batch_vs_frames.log: 111.6 Gpix/s as launch pairs over three streams, 118.5 frame by frame).  Frame f of the batch
  // runs on slot (next_slot + f) mod N; the frames are the same kernels on the same values as in one launch: bit-identical
  // (tests/test_gpu_pipeline.py).  8K frames (265 MB each) and fused launches keep the single launch.
  if (plan.pipeline == 0 && out->frames > 1 && p->n > 1 &&
      (size_t)out->width * pixel_bytes(out->format) * (size_t)out->height * (size_t)p->n <= infinity_cache_bytes()) {
    ImageView vi, vo;
    if (int rc = check_image(in, "pipeline_upscale input", &vi)) return rc;
    if (int rc = check_image(out, "pipeline_upscale output", &vo)) return rc;
    if (overlaps(in, vi, out, vo)) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_upscale: input and output overlap");
    // every frame runs what was decided for the BATCH (decided again per frame, `auto` could take the fused launch for single
    // frames of a batch planned as two dispatches): the per-frame calls carry fused = 0
    fsr1_params per_frame = *params;
    per_frame.fused = 0;
    {  // everything a per-frame call could refuse is refused here, before frame 0 is submitted (the frames differ only in their base pointers)
      fsr1_image fi = *in, fo = *out;
      fi.frames = fo.frames = 1;
      fi.frame_stride_bytes = fo.frame_stride_bytes = 0;
      UpscalePlan one;
      if (int rc = upscale_decide("pipeline_upscale (frame of a batch)", &fi, true, &fo, &per_frame, stages != nullptr, &one)) return rc;
      if (one.pipeline != 0) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_upscale: a frame of a two-dispatch batch planned as pipeline %d", one.pipeline);
    }
    for (int f = 0; f < out->frames; ++f) {
      fsr1_image fi = *in, fo = *out;
      fi.data = vi.base + (long long)f * vi.frame_stride;
      fo.data = vo.base + (long long)f * vo.frame_stride;
      fi.frames = fo.frames = 1;
      fi.frame_stride_bytes = fo.frame_stride_bytes = 0;
      if (int rc = fsr1_pipeline_upscale(p, &fi, &fo, &per_frame, stages)) {
        // (a HIP failure mid-batch: frames 0 .. f-1 are submitted and the slot counter has advanced by f — say so)
        char first[384];
        snprintf(first, sizeof first, "%s", g_err);
        return fail(rc, "pipeline_upscale: frame %d of %d failed after %d frame(s) were submitted (next slot %d): %s", f, out->frames, f, p->next, first);
      }
    }
    return FSR1_OK;
  }
  const int slot = p->next;
  fsr1_image mid_img;
  const fsr1_image* mid_p = nullptr;
  if (plan.pipeline == 0) {  // two dispatches: this slot's own intermediary, out's format and extent
    const size_t need = (size_t)out->width * pixel_bytes(out->format) * (size_t)out->height * (size_t)out->frames;
    if (int rc = pipeline_grow(p, slot, need, "pipeline_upscale")) return rc;
    mid_img = fsr1_image{p->mid[slot], out->width, out->height, out->format, out->frames, 0, 0};
    mid_p = &mid_img;
  }
  // (the plan is re-derived by fsr1_upscale_ex from the same arguments: `auto` with an intermediary on offer decides as it did above)
  fsr1_params prm = *params;
  if (p->n > 1) prm.flags |= FSR1_FLAG_FRAMES_OVERLAP;         // the frames of this pipeline run beside each other
  if (plan.pipeline == 1) prm.fused = 1;                         // what was decided, whatever `auto` would say without an intermediary
  else if (plan.pipeline == 0) prm.fused = 0;
  if (int rc = fsr1_upscale_ex(in, mid_p, out, &prm, stages, p->streams[slot])) return rc;
  if (hipError_t e = hipEventRecord(p->done[slot], p->streams[slot]); e != hipSuccess) return hip_fail(e, "pipeline_upscale: hipEventRecord");
  p->next = (slot + 1) % p->n;
  return FSR1_OK;
}

int fsr1_pipeline_next_slot(const fsr1_pipeline* p) { return p && p->n > 0 ? p->next : -1; }

int fsr1_pipeline_fork(fsr1_pipeline* p, void* stream) {
  if (!p) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_fork: null pipeline");
  hipError_t e = hipEventRecord(p->fork_ev, static_cast<hipStream_t>(stream));
  for (int i = 0; i < p->n && e == hipSuccess; ++i) e = hipStreamWaitEvent(p->streams[i], p->fork_ev, 0);
  return e == hipSuccess ? FSR1_OK : hip_fail(e, "pipeline_fork");
}

int fsr1_pipeline_join(fsr1_pipeline* p, void* stream) {
  if (!p) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_join: null pipeline");
  hipError_t e = hipSuccess;
  for (int i = 0; i < p->n && e == hipSuccess; ++i) {
    e = hipEventRecord(p->done[i], p->streams[i]);
    if (e == hipSuccess) e = hipStreamWaitEvent(static_cast<hipStream_t>(stream), p->done[i], 0);
  }
  return e == hipSuccess ? FSR1_OK : hip_fail(e, "pipeline_join");
}

int fsr1_pipeline_synchronize(fsr1_pipeline* p) {
  if (!p) return fail(FSR1_ERR_INVALID_ARGUMENT, "pipeline_synchronize: null pipeline");
  for (int i = 0; i < p->n; ++i)
    if (hipError_t e = hipStreamSynchronize(p->streams[i]); e != hipSuccess) return hip_fail(e, "pipeline_synchronize");
  return FSR1_OK;
}

int fsr1_pipeline_streams(const fsr1_pipeline* p) { return p ? p->n : 0; }
void* fsr1_pipeline_stream(const fsr1_pipeline* p, int32_t i) { return p && i >= 0 && i < p->n ? p->streams[i] : nullptr; }

int fsr1_pipeline_destroy(fsr1_pipeline* p) {
  if (!p) return FSR1_OK;
  for (int i = 0; i < fsr1_pipeline::kMax; ++i) {
    if (p->streams[i]) (void)hipStreamSynchronize(p->streams[i]);
    if (p->mid[i]) (void)hipFree(p->mid[i]);  // (stream-ordered allocations may be released with hipFree once their stream is idle)
    if (p->streams[i]) (void)hipStreamDestroy(p->streams[i]);
    if (p->done[i]) (void)hipEventDestroy(p->done[i]);
  }
  if (p->fork_ev) (void)hipEventDestroy(p->fork_ev);
  delete p;
  return FSR1_OK;
}

// ---- HIP-event stopwatch ----
struct fsr1_timer {
  hipEvent_t start, stop;
};

int fsr1_timer_create(void** timer) {
  if (!timer) return fail(FSR1_ERR_INVALID_ARGUMENT, "timer_create: null");
  fsr1_timer* t = new fsr1_timer;
  hipError_t e = hipEventCreate(&t->start);
  if (e == hipSuccess) e = hipEventCreate(&t->stop);
  if (e != hipSuccess) { delete t; return hip_fail(e, "hipEventCreate"); }
  *timer = t;
  return FSR1_OK;
}
int fsr1_timer_start(void* timer, void* stream) {
  if (!timer) return fail(FSR1_ERR_INVALID_ARGUMENT, "timer_start: null");
  hipError_t e = hipEventRecord(static_cast<fsr1_timer*>(timer)->start, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? FSR1_OK : hip_fail(e, "hipEventRecord");
}
int fsr1_timer_stop(void* timer, void* stream) {
  if (!timer) return fail(FSR1_ERR_INVALID_ARGUMENT, "timer_stop: null");
  hipError_t e = hipEventRecord(static_cast<fsr1_timer*>(timer)->stop, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? FSR1_OK : hip_fail(e, "hipEventRecord");
}
int fsr1_timer_elapsed_ms(void* timer, float* ms) {
  if (!timer || !ms) return fail(FSR1_ERR_INVALID_ARGUMENT, "timer_elapsed_ms: null");
  fsr1_timer* t = static_cast<fsr1_timer*>(timer);
  hipError_t e = hipEventSynchronize(t->stop);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, t->start, t->stop);
  return e == hipSuccess ? FSR1_OK : hip_fail(e, "hipEventElapsedTime");
}
int fsr1_timer_destroy(void* timer) {
  if (!timer) return FSR1_OK;
  fsr1_timer* t = static_cast<fsr1_timer*>(timer);
  (void)hipEventDestroy(t->start);
  (void)hipEventDestroy(t->stop);
  delete t;
  return FSR1_OK;
}

}  // extern "C"
