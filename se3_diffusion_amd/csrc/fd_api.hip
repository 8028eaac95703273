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

// kernel launches issued by this library since it was loaded (FD_CHECK_LAUNCH counts every successful one: also launches that are
// only RECORDED into a hipGraph capture -- which is how the sampler counts the kernels of its captured diffusion step)
#include <atomic>
static std::atomic<long> g_launches{0};
void fd_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" long fd_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

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
