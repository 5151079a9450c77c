// libfsr1_hip_test.so's side of csrc/fsr1_overrides.h: process-wide switches that force the launch shape, set through the
// fsr1_debug_* functions of include/fsr1_hip_test.h.  Not part of the product library (libfsr1_hip.so links
// fsr1_overrides_none.cpp instead and exports none of this).
#include <atomic>

#include "fsr1_hip_test.h"
#include "fsr1_overrides.h"

namespace {
std::atomic<int> g_fused_s2_steps{0};
std::atomic<int> g_fused_s2_tall{-1};
std::atomic<int> g_easu_s2_tall{-1};
int tri(int mode) { return mode < 0 ? -1 : (mode ? 1 : 0); }
}  // namespace

namespace fsr1 {
int override_fused_s2_steps() { return g_fused_s2_steps.load(std::memory_order_relaxed); }
int override_fused_s2_tall() { return g_fused_s2_tall.load(std::memory_order_relaxed); }
int override_easu_s2_tall() { return g_easu_s2_tall.load(std::memory_order_relaxed); }
}  // namespace fsr1

extern "C" {
void fsr1_debug_fused_run_steps(int32_t steps) { g_fused_s2_steps.store(steps < 0 ? 0 : (steps > 64 ? 64 : steps), std::memory_order_relaxed); }
void fsr1_debug_fused_tall_tiles(int32_t mode) { g_fused_s2_tall.store(tri(mode), std::memory_order_relaxed); }
void fsr1_debug_easu_tall_tiles(int32_t mode) { g_easu_s2_tall.store(tri(mode), std::memory_order_relaxed); }
}
