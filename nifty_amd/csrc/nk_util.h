// nk_util.h -- error plumbing of libniftyk (never throw across the C ABI)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/niftyk.h"

int nk_set_error(int code, const char* msg);
int nk_set_hip_error(hipError_t e, const char* what);

static inline int nk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nk_set_hip_error(e, what);
  return NK_OK;
}

// Function attributes (dynamic LDS size) are per DEVICE: a launcher keeps one bit per device id in a static mask.
// Returns true the first time it is called for the current device (ids >= 64 are set every time).
static inline bool nk_first_on_device(unsigned long long& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  const unsigned long long bit = 1ULL << dev;
  if (mask & bit) return false;
  mask |= bit;
  return true;
}
