// Library identity + error strings for the C ABI declared in include/pocketflow_hip.h.
#include "pf_common.h"

extern "C" int pf_version(void) { return 100; }   // 0.1.0 (round 1)

extern "C" const char* pf_error_string(int err) {
  if (err == 0) return "ok";
  return hipGetErrorString((hipError_t)err);
}
