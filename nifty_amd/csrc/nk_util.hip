// nk_util.hip -- error state + version
#include "nk_util.h"

#include <cstdio>

static thread_local char g_err[512] = "";

int nk_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}

int nk_set_hip_error(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what ? what : "hip", hipGetErrorString(e));
  return e == hipErrorOutOfMemory ? NK_ERR_NOMEM : NK_ERR_RUNTIME;
}

extern "C" const char* nk_last_error(void) { return g_err; }
extern "C" int nk_version(void) { return 100; }
