// Entry points not built yet in this revision report hipErrorNotSupported through the C ABI.
#include "fsr1_device.h"
namespace fsr1 {
hipError_t easu_h_launch(const EasuArgs&, hipStream_t) { return hipErrorNotSupported; }
hipError_t rcas_h_launch(const RcasArgs&, hipStream_t) { return hipErrorNotSupported; }
}  // namespace fsr1
