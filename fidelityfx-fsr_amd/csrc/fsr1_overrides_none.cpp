// The product library's side of csrc/fsr1_overrides.h: the launch-shape rules are never overridden.  No state.
#include "fsr1_overrides.h"

namespace fsr1 {
int override_fused_s2_steps() { return 0; }
int override_fused_s2_tall() { return -1; }
int override_easu_s2_tall() { return -1; }
}  // namespace fsr1
