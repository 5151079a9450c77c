// Private side of the FSR 1.0 HIP kernels (gfx950 only): launch geometry, kernel argument blocks, the XCD-aware
// workgroup -> tile mapping and host launch helpers.  The arithmetic lives in the public device headers
// (include/fsr1_device*.hpp), which these kernels include like any other user of the library would.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "fsr1_device_base.hpp"
#include "fsr1_device_color.hpp"

namespace fsr1 {

// Output tile of one 256-thread workgroup (4 waves): 64 x 16 pixels, each wave owns 4 rows, a
// lane owns one column -> every global store instruction writes 64 consecutive pixels.
constexpr int kTileW = 64;
constexpr int kTileH = 16;  // multiple of 4 (rows are split over the four waves); 8 / 12 / 24 / 32 measured slower at every ratio (profiles/ab_r02/c11_*)
constexpr int kThreads = 256;
constexpr int kFusedTileH = 16;  // output rows per fused-kernel tile (multiple of 4)
// Exact-2x fused launch (fsr1_fused_s2.hip, fsr1_fused_s2_h.hip): a workgroup owns a 62-pixel column and walks down it in steps of
// 16 EASU rows — 32 x 8 quads, one per lane — kept in an LDS ring of 18 rows (the 16 new ones and the last two of the step before).
constexpr int kFs2OutW = 62, kFs2MidW = 64, kFs2FpW = 35;
constexpr int kFs2QH = 8, kFs2FpH = kFs2QH + 3, kFs2Step = 2 * kFs2QH, kFs2Ring = kFs2Step + 2;  // quad rows, footprint rows, EASU rows per step, ring rows

struct ColorPassArgs {
  ImageView in, out;
  int xcd_shift;  // log2 of the XCDs behind the device (3 on an MI355X in SPX mode): xcd_swizzle
  int tiles_x, tiles_y, frames;
  ColorArgs color;
};

struct EasuArgs {
  ImageView in, out;
  int xcd_shift;  // log2 of the XCDs behind the device (3 on an MI355X in SPX mode): xcd_swizzle
  uint32_t con[16];
  int tiles_x, tiles_y, frames;
  int fp_w, fp_h;  // LDS footprint capacity (texels) per tile, >= the largest footprint of any tile
  uint32_t flags;
  int origin_x, origin_y;  // `out` is a window of the full output image: its pixel (0, 0) is output pixel (origin_x, origin_y)
  ColorArgs color;
};

struct RcasArgs {
  ImageView in, out;
  int xcd_shift;  // log2 of the XCDs behind the device (3 on an MI355X in SPX mode): xcd_swizzle
  uint32_t con[4];
  int tiles_x, tiles_y, frames;
  int rows;  // rows per strip (a multiple of the kernel's row ring), chosen by the launcher
  uint32_t flags;
  int rows_above, rows_below;  // 1: the image continues in memory above row 0 / below the last row (a band of a larger image): those taps are read, not 0
  ColorArgs color;
};

struct FusedArgs {
  ImageView in, out;
  int xcd_shift;  // log2 of the XCDs behind the device (3 on an MI355X in SPX mode): xcd_swizzle
  uint32_t easu_con[16];
  uint32_t rcas_con[4];
  int tiles_x, tiles_y, frames;
  int fp_w, fp_h;
  uint32_t flags;
  ColorArgs color;
  // band of a larger output image (fsr1_easu_rcas_fused_dispatch_band): output row 0 is row origin_y of the image the EASU
  // constants describe; rows_above / rows_below (0 or 1): that image has a row above / below the band, which the apron computes
  int origin_y, rows_above, rows_below;
  int run_steps;  // exact-2x kernel (fsr1_fused_s2.hip): 16-row steps a workgroup walks down its column
};

// XCD-aware workgroup -> tile mapping.  Consecutive workgroup ids round-robin over the device's XCDs
// (each with a private 4 MiB L2), so the ids that land on one XCD are given one contiguous range
// of tiles: neighbouring tiles, which share their input aprons, then share an L2.
// xcd_shift: log2 of the XCD count, from the host (fsr1_api.hip: the device's CU count / 32 — 8 XCDs on an MI355X in SPX mode, 4 / 2 / 1
// in the DPX / QPX / CPX partition modes, where the literal 8 of rounds 1-5 would have scattered neighbouring tiles over L2s that do
// not exist); any value gives a permutation of the tiles, i.e. the same image.
__device__ __forceinline__ int xcd_swizzle(int b, int n, int xcd_shift) {
  const int xcds = 1 << xcd_shift;
  const int q = n >> xcd_shift, r = n & (xcds - 1);
  const int xcd = b & (xcds - 1), idx = b >> xcd_shift;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// Kernels that may need more than the default 48 KiB of dynamic LDS ask for it through hipFuncSetAttribute — once per
// kernel, device and size: the attribute sticks, and re-issuing the call on every launch costs host time on the
// launch path (round 1 did).  The cache is keyed by (device, function); the attribute only ever GROWS: raising it and
// recording the new size happen under one mutex (two host threads launching the same kernel with different sizes could
// otherwise leave the smaller size applied and the larger one recorded), the common case — already large enough — is one
// atomic load.  Devices beyond the table (none exists: 64 per process) set the attribute on every launch, under the same mutex.
inline hipError_t ensure_dynamic_lds(const void* fn, size_t lds) {
  if (lds <= 48 * 1024) return hipSuccess;
  struct Slot { std::atomic<const void*> fn{nullptr}; std::atomic<size_t> bytes{0}; };
  constexpr int kDevices = 64, kSlots = 64;
  static Slot cache[kDevices][kSlots];
  static std::mutex raise;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  Slot* slot = nullptr;
  if (dev >= 0 && dev < kDevices) {
    Slot* row = cache[dev];
    for (int i = 0; i < kSlots && !slot; ++i) {
      const void* cur = row[i].fn.load(std::memory_order_acquire);
      if (cur == fn) slot = &row[i];
      else if (!cur) {
        const void* expected = nullptr;
        if (row[i].fn.compare_exchange_strong(expected, fn, std::memory_order_acq_rel) || expected == fn) slot = &row[i];
      }
    }
  }
  if (slot && slot->bytes.load(std::memory_order_acquire) >= lds) return hipSuccess;
  std::lock_guard<std::mutex> lock(raise);
  if (slot && slot->bytes.load(std::memory_order_relaxed) >= lds) return hipSuccess;  // another thread raised it meanwhile
  if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
  if (slot) slot->bytes.store(lds, std::memory_order_release);
  return hipSuccess;
}

}  // namespace fsr1
