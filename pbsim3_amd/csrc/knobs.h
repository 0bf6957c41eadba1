// knobs.h -- environment knobs of CLOSED experiments (each measured as a same-box A/B, the outcome in DESIGN.md section 8d and
// profiles/): compiled out of the product, so that the parity suite covers every path the shipped library can take.  A build
// with -DPBSIM_EXPERIMENTAL (PBSIM_EXTRA_CFLAGS=-DPBSIM_EXPERIMENTAL python -m pbsim3_amd.build --force) reads them again, for
// the tools/ scripts that re-run an A/B.  The knobs that stay live (test hooks and operational settings) are listed in
// INTEGRATION.md with the test that covers each.
#pragma once
#include <stdlib.h>

namespace pbsim {
inline const char *exp_env(const char *name) {
#ifdef PBSIM_EXPERIMENTAL
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}  // namespace pbsim
