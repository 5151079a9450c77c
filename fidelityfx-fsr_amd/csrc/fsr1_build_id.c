/* fsr1_build_id(): the hash of the sources this binary was built from, baked in at build time (csrc/Makefile computes
 * FSR1_BUILD_ID_STR over the same files, in the same order, as fidelityfx-fsr_amd/_lib.py:source_hash() does at run time). */
#include "fsr1_hip.h"

const char* fsr1_build_id(void) { return FSR1_BUILD_ID_STR; }
