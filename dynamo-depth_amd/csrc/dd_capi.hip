// dd_capi.hip -- ABI housekeeping for libdynamo_hip.so (version + error decoding).
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"

extern "C" int dd_abi_version(void) { return DD_ABI_VERSION; }

extern "C" const char* dd_error_string(int code) {
  if (code == 0) return "success";
  return hipGetErrorString(static_cast<hipError_t>(code));
}
