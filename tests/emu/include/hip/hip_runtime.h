// TEST INFRASTRUCTURE ONLY -- never shipped, never loaded by the product package.
//
// A minimal SIMT interpreter that lets the gfx950 kernel sources under
// se3_diffusion_amd/csrc/ be compiled with g++ and executed on the host so the
// `-m "not gpu"` test tier can check kernel *logic* (indexing, tiling, MFMA
// fragment maps, reductions) against the oracle without a GPU.  Each HIP thread
// is a fiber; __syncthreads() and wave-level collectives (shuffles, MFMA) are
// rendezvous points.  The product build (hipcc --offload-arch=gfx950) never sees
// this header: it is only on the include path of tests/emu/build_emu.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
static const hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static const int hipMemcpyDeviceToDevice = 3;

namespace hipemu {

constexpr int kWave = 64;

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  dim3 tid;
  int linear = 0;
  int wave = 0;
  int lane = 0;
  bool done = false;
};

struct WaveXchg {
  // double-buffered exchange area for wave collectives
  alignas(16) unsigned char buf[2][kWave][64];
  int arrived[2] = {0, 0};
  int nlanes = 0;
};

extern Fiber* g_cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern long g_progress;

void yield();
void barrier();
WaveXchg* cur_wave();
// deposit `bytes` at this lane's slot, wait for the whole wave, return pointer
// to the [kWave][64] slot table of this collective.  Caller must call
// wave_done() after it finished reading.
unsigned char (*wave_exchange(const void* src, int bytes))[64];
void launch(dim3 grid, dim3 block, const std::function<void()>& body);

}  // namespace hipemu

#define threadIdx (hipemu::g_cur->tid)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)            \
  do {                                                                         \
    (void)(shmem); (void)(stream);                                             \
    hipemu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); });           \
  } while (0)

static inline void __syncthreads() { hipemu::barrier(); }

template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) <= 64, "too big");
  auto tab = hipemu::wave_exchange(&v, sizeof(T));
  int lane = hipemu::g_cur->lane;
  int base = lane & ~(width - 1);
  int s = base + (src & (width - 1));
  T r;
  memcpy(&r, tab[s], sizeof(T));
  return r;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  int lane = hipemu::g_cur->lane;
  return __shfl(v, (lane ^ mask), width);
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = hipemu::g_cur->lane;
  int l = (lane & (width - 1)) + (int)d;
  // out-of-range lanes keep their own value (HIP semantics)
  auto tab = hipemu::wave_exchange(&v, sizeof(T));
  T r = v;
  if (l < width) memcpy(&r, tab[(lane & ~(width - 1)) + l], sizeof(T));
  return r;
}

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline void sincosf_emu(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline unsigned __float_as_uint(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
static inline float __uint_as_float(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
