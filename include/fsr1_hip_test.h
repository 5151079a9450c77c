/*
 * fsr1_hip_test.h — test hooks of libfsr1_hip_test.so.  NOT part of the product ABI.
 *
 * libfsr1_hip_test.so is libfsr1_hip.so (same object files, same kernels, every symbol of fsr1_hip.h) plus three
 * process-wide switches that force the launch SHAPE the host's rules would otherwise pick.  Every shape produces the
 * same image, bit for bit — which is exactly what the tests use the switches to show
 * (tests/test_gpu_parity.py::test_fused_exact_2x_run_steps, ::test_fused_exact_2x_tall_tile, ::test_easu_exact_2x_tall_tiles)
 * and what tuning runs use them to measure (tools/abtest.py).  The product library exports none of them and has no
 * process-wide state (csrc/fsr1_overrides.h is the link-time seam).
 */
#ifndef FSR1_HIP_TEST_H
#define FSR1_HIP_TEST_H

#include "fsr1_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of 16-row steps the workgroups of the exact-2x fused launch walk down their columns (fsr1_fused_s2.hip), clamped to
 * 0 .. 64; 0 restores the host's rule. */
void fsr1_debug_fused_run_steps(int32_t steps);
/* Tile shape of its one-step launches: -1 = the host's rule (the 62 x 30 tile of a 512-thread workgroup for frames that fill the
 * chip, the 62 x 14 tile of a 256-thread one otherwise), 0 = never the tall tile, 1 = always. */
void fsr1_debug_fused_tall_tiles(int32_t mode);
/* ... and of the EASU launches (F arithmetic, F-strict included), BOTH rules at once: exact 2x — 64 x 32 tiles for large or overlapped launches,
 * 64 x 16 otherwise (easu_s2_tall_tiles) — and any other ratio — the 512-thread workgroup on 64 x 32 tiles where its footprint leaves three
 * workgroups per CU, the 256-thread one on 64 x 16 otherwise (easu_generic_tall_tiles); -1 / 0 / 1 as above (1 is still refused where the
 * 32-row footprint does not fit a CU's LDS at all). */
void fsr1_debug_easu_tall_tiles(int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* FSR1_HIP_TEST_H */
