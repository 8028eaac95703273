// Shared host-side helpers for the FrameDiff C-ABI (see include/fd_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <fd_intrin.h>
#include <cstdio>
#include <cstdarg>

#define FD_OK 0
#define FD_ERR_ARG -1
#define FD_ERR_LAUNCH -2
#define FD_ERR_UNSUPPORTED -3

void fd_set_error(const char* fmt, ...);
void fd_count_launch();      // one kernel launch issued by the library (fd_launch_count, include/fd_hip.h)

#define FD_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      fd_set_error(__VA_ARGS__);           \
      return FD_ERR_ARG;                   \
    }                                      \
  } while (0)

#define FD_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      fd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return FD_ERR_LAUNCH;                                                     \
    }                                                                           \
    fd_count_launch();                                                          \
  } while (0)

static inline int fd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool fd_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed-only assumption).
// Map the hardware block id to a logical id so that logically consecutive
// blocks (which share an operand panel) land on the same XCD's L2.
__device__ __forceinline__ int fd_xcd_swizzle(int bid, int nblk) {
  const int NX = 8;
  int q = nblk / NX, r = nblk % NX;
  int xcd = bid % NX, slot = bid / NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}
