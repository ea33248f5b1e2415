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
