// Error plumbing and version query of the C ABI (include/csmae.h).
#include "common.h"
#include <string>

static thread_local char g_err[512] = "";

void csmae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int csmae_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    csmae_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return CSMAE_ERR_LAUNCH;
  }
  return CSMAE_OK;
}

extern "C" const char* csmae_last_error(void) { return g_err; }
extern "C" int csmae_abi_version(void) { return 3; }
#ifndef CSMAE_SRC_HASH
#define CSMAE_SRC_HASH "unknown"
#endif
extern "C" const char* csmae_source_hash(void) { return CSMAE_SRC_HASH; }   // sha256 of csrc/ at build time (tools/csrc_hash.py)
