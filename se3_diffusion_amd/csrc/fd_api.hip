// C-ABI plumbing shared by all kernels: error string, version, backend tag.
#include "fd_common.h"
#include "../../include/fd_hip.h"
#include <cstring>

static thread_local char g_err[512] = "";

void fd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fd_last_error(void) { return g_err; }
extern "C" int fd_abi_version(void) { return FD_ABI_VERSION; }
extern "C" const char* fd_backend(void) { return FD_BACKEND_NAME; }
// "" for a product build; the probe switch when tools/probes built this library (fd_probe.h: the timing / ablation hooks of the
// kernel sources, several of which compute wrong results by design, compile only under -DFD_PROBE_BUILD).  tests/test_abi.py
// asserts that the shipped library reports "".
extern "C" const char* fd_build_flags(void) {
#ifdef FD_PROBE_BUILD
  return "FD_PROBE_BUILD";
#else
  return "";
#endif
}
