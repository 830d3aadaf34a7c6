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
extern "C" int csmae_abi_version(void) { return 7; }

// ---- an event attached to the next launch (common.h: CSMAE_LAUNCH).  The reference has no counterpart: torch records events behind kernels
// (autograd's stream hand-offs, DDP's bucket hooks — main_pretrain.py:417-421); here the weight-gradient stream of csmae_hip/engine.py waits for
// the main chain's dX GEMM / attention backward through the kernel's own completion signal.
thread_local hipEvent_t g_csmae_launch_event = nullptr;
extern "C" int csmae_next_launch_event(void* hip_event) { g_csmae_launch_event = (hipEvent_t)hip_event; return CSMAE_OK; }
// after the launch: if its launch site did not take the event (not every kernel goes through CSMAE_LAUNCH), record it the plain way
extern "C" int csmae_flush_launch_event(void* stream) {
  hipEvent_t ev = g_csmae_launch_event;
  if (!ev) return CSMAE_OK;
  g_csmae_launch_event = nullptr;
  hipError_t e = hipEventRecord(ev, (hipStream_t)stream);
  if (e != hipSuccess) { csmae_set_error("csmae_flush_launch_event: %s", hipGetErrorString(e)); return CSMAE_ERR_LAUNCH; }
  return CSMAE_OK;
}
// ---- a HIP stream confined to a subset of the compute units (include/csmae.h).  The reference's counterpart is the CUDA stream DDP's reducer and
// autograd's hand-offs run on (main_pretrain.py:417-421) — which cannot be confined; here the weight-gradient stream (and RCCL's) gets a fixed
// share of the CUs instead of time-slicing whole CUs away from the main chain's kernels.
extern "C" int csmae_stream_create_cu_mask(int words, const unsigned* mask, void** out) {
  CSMAE_REQUIRE(words > 0 && mask && out, "csmae_stream_create_cu_mask: bad args");
  int any = 0;
  for (int i = 0; i < words; ++i) any |= mask[i] != 0u;
  CSMAE_REQUIRE(any, "csmae_stream_create_cu_mask: empty mask (a queue without compute units never finishes)");
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
  if (e != hipSuccess) { csmae_set_error("csmae_stream_create_cu_mask: %s", hipGetErrorString(e)); return CSMAE_ERR_LAUNCH; }
  *out = (void*)st;
  return CSMAE_OK;
}
extern "C" int csmae_stream_destroy(void* stream) {
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) { csmae_set_error("csmae_stream_destroy: %s", hipGetErrorString(e)); return CSMAE_ERR_LAUNCH; }
  return CSMAE_OK;
}
#ifndef CSMAE_SRC_HASH
#define CSMAE_SRC_HASH "unknown"
#endif
extern "C" const char* csmae_source_hash(void) { return CSMAE_SRC_HASH; }   // sha256 of csrc/ at build time (tools/csrc_hash.py)

// CSMAE_DEBUG="key[=value],..." lookup (common.h).  The returned pointer refers to a per-key copy that lives for the life of the process.
#include <map>
#include <mutex>
#include <string>
#include <cstring>
#include <cstdlib>
const char* csmae_debug_opt(const char* key) {
  static std::mutex mu;
  static std::map<std::string, std::string> found;
  const char* env = getenv("CSMAE_DEBUG");
  if (!env) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  const size_t kl = strlen(key);
  for (const char* p = env; *p;) {
    const char* e = strchr(p, ',');
    size_t len = e ? (size_t)(e - p) : strlen(p);
    while (len && (*p == ' ' || *p == '\t')) { ++p; --len; }                      // the Python reader (csmae_hip.debug_opt) strips items: so does this one
    while (len && (p[len - 1] == ' ' || p[len - 1] == '\t')) --len;
    if (len >= kl && strncmp(p, key, kl) == 0 && (len == kl || p[kl] == '=')) {
      std::string& v = found[key];
      v = len == kl ? "1" : std::string(p + kl + 1, len - kl - 1);
      return v.c_str();
    }
    if (!e) break;
    p = e + 1;
  }
  return nullptr;
}
