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
