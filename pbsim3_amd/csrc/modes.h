// modes.h -- constants shared by the host table builder and the kernels.
#pragma once
#include <stdint.h>
namespace pbsim {
// how an accuracy class relates to the ERRHMM model's own range (pbsim.cpp:3852-3926)
enum ErrMode : uint32_t { kModeInRange = 0, kModeBelow = 1, kModeAbove = 2, kModeVerbatim = 3 };
}  // namespace pbsim
