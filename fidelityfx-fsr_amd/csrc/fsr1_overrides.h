// Link-time seam for the launch-shape rules (fused_s2_run_steps, fused_s2_tall_tiles, easu_s2_tall_tiles).
//
// The rules are pure functions of the launch's extents and the device's CU count.  Tests (and tuning runs) need to force
// every shape a rule can pick, to show that all of them produce the same image; the PRODUCT library must not carry that
// switch: include/fsr1_hip.h promises no hidden process-wide state.  So the three functions below are defined twice:
//   fsr1_overrides_none.cpp   linked into libfsr1_hip.so        constants: "no override", no state, nothing exported
//   fsr1_test_hooks.cpp       linked into libfsr1_hip_test.so   atomics set through fsr1_debug_* (include/fsr1_hip_test.h)
// Same kernels, same object files otherwise: the two libraries differ in this one translation unit.
#pragma once

namespace fsr1 {
int override_fused_s2_steps();  // > 0: that many 16-row steps per run of the exact-2x fused launch; 0: the host's rule
int override_fused_s2_tall();   // 0 / 1: never / always the 512-thread tile for one-step launches; -1: the host's rule
int override_easu_s2_tall();    // 0 / 1: never / always 64 x 32 tiles for exact-2x F EASU launches; -1: the host's rule
}  // namespace fsr1
