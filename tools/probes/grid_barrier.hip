// What a device-wide barrier inside ONE launch costs on this chip, against the ~7.7 us a dependent kernel costs in a replayed graph
// (the price of every node-level launch of lone-backbone sampling).  G blocks of 256 threads; between barriers every block writes a
// 4 KB tile and reads the tile another block (another XCD) wrote in the previous phase -- the read is CHECKED, so the probe also
// says whether the release / acquire pair used is enough across the eight L2s.  The spin is bounded (a lost block ends the kernel
// with an error count instead of hanging the GPU).
//   hipcc --offload-arch=gfx950 -O2 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ bool wait_ge(unsigned* ctr, unsigned target) {
  for (long spin = 0; spin < 20000000; ++spin) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// mode 0: thread 0 fences (release before the arrive, acquire after the wait); mode 1: every thread fences
template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target) {
  __shared__ int ok;
  if (MODE == 1) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) __threadfence();
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    ok = wait_ge(ctr, target) ? 1 : 0;
    if (MODE == 0) __threadfence();
  }
  __syncthreads();
  if (MODE == 1) __threadfence();
  return ok != 0;
}

// modes 2 / 3 / 4: nobody polls the arrival counter.  The LAST block to arrive (its fetch_add returns target - 1) releases the others by
// writing the epoch into one flag PER BLOCK (256 bytes apart: different channels), and every block polls its own flag.  mode 3 arrives
// on one of eight sub-counters (blockIdx & 7) whose last arrivers meet on a top counter; mode 4 = mode 2 without the fences (timing of
// the fences only: its reads may be stale).
template <int MODE>
__device__ __forceinline__ bool grid_barrier_f(unsigned* ctr, unsigned* flags, unsigned epoch, int G) {
  __shared__ int ok;
  __syncthreads();
  if (threadIdx.x < 64) {
    bool last = false;
    if (threadIdx.x == 0) {
      if (MODE != 4) __threadfence();
      if (MODE == 3) {
        const int grp = (int)blockIdx.x & 7;
        const unsigned ng = (unsigned)((G - grp + 7) >> 3);
        const unsigned old = __hip_atomic_fetch_add(ctr + 16 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == epoch * ng - 1) {
          const unsigned ngroups = G < 8 ? (unsigned)G : 8u;
          last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * ngroups - 1;
        }
      } else {
        last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * (unsigned)G - 1;
      }
    }
    last = __shfl((int)last, 0) != 0;
    if (last)
      for (int b = (int)threadIdx.x; b < G; b += 64) __hip_atomic_store(flags + 64 * b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
      ok = wait_ge(flags + 64 * blockIdx.x, epoch) ? 1 : 0;
      if (MODE != 4) __threadfence();
    }
  }
  __syncthreads();
  return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe_f(unsigned* ctr, unsigned* flags, float* buf, int iters, unsigned* err) {
  const int G = (int)gridDim.x, b = (int)blockIdx.x, t = (int)threadIdx.x;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    float* cur = buf + (long)(it & 1) * G * 1024;
    reinterpret_cast<float4*>(cur + (long)b * 1024)[t] = make_float4((float)it, (float)b, (float)t, 1.f);
    if (!grid_barrier_f<MODE>(ctr, flags, (unsigned)(it + 1), G)) { bad |= 0x80000000u; break; }
    const int src = (b + 1 + it) % G;
    const float4 v = reinterpret_cast<const float4*>(cur + (long)src * 1024)[t];
    if (v.x != (float)it || v.y != (float)src || v.z != (float)t) ++bad;
  }
  if (bad) atomicAdd(err, bad & 0x7fffffffu ? bad & 0x7fffffffu : 1u), atomicOr(err + 1, bad >> 31);
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned* ctr, float* buf, int iters, unsigned* err) {
  const int G = (int)gridDim.x, b = (int)blockIdx.x, t = (int)threadIdx.x;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    float* cur = buf + (long)(it & 1) * G * 1024;
    // this block's 4 KB tile of phase `it`
    reinterpret_cast<float4*>(cur + (long)b * 1024)[t] = make_float4((float)it, (float)b, (float)t, 1.f);
    if (!grid_barrier<MODE>(ctr, (unsigned)(it + 1) * (unsigned)G)) { bad |= 0x80000000u; break; }
    const int src = (b + 1 + it) % G;      // (consecutive blocks sit on different XCDs)
    const float4 v = reinterpret_cast<const float4*>(cur + (long)src * 1024)[t];
    if (v.x != (float)it || v.y != (float)src || v.z != (float)t) ++bad;
  }
  if (bad) atomicAdd(err, bad & 0x7fffffffu ? bad & 0x7fffffffu : 1u), atomicOr(err + 1, bad >> 31);
}

__global__ void empty_kernel(float* p) { if (p == nullptr) return; }

int main() {
  unsigned* ctr; float* buf; unsigned* err; unsigned* flags;
  hipMalloc(&flags, 512 * 256);
  hipMalloc(&ctr, 1024); hipMalloc(&buf, 2l * 512 * 1024 * 4); hipMalloc(&err, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 5; ++mode)
    for (int G : {8, 40, 64, 128, 256, 512}) {
      float best = 1e9f; unsigned herr[2] = {0, 0};
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ctr, 0, 1024); hipMemset(err, 0, 8); hipMemset(flags, 0, 512 * 256);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(256), 0, 0, ctr, buf, iters, err);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(G), dim3(256), 0, 0, ctr, buf, iters, err);
        else if (mode == 2) hipLaunchKernelGGL(probe_f<2>, dim3(G), dim3(256), 0, 0, ctr, flags, buf, iters, err);
        else if (mode == 3) hipLaunchKernelGGL(probe_f<3>, dim3(G), dim3(256), 0, 0, ctr, flags, buf, iters, err);
        else hipLaunchKernelGGL(probe_f<4>, dim3(G), dim3(256), 0, 0, ctr, flags, buf, iters, err);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        unsigned h[2]; hipMemcpy(h, err, 8, hipMemcpyDeviceToHost); herr[0] += h[0]; herr[1] |= h[1];
      }
      printf("mode %d (%s)  G=%3d blocks: %.3f us per (write 4 KB, barrier, read 4 KB)   stale reads %u   timed out %u\n", mode,
             mode == 0 ? "thread 0 fences" : mode == 1 ? "all threads fence" : mode == 2 ? "own flag per block" : mode == 3 ? "own flag, 8 sub-counters" : "own flag, NO fences", G, best * 1e3f / iters, herr[0], herr[1]);
      fflush(stdout);
    }
  // for scale: back-to-back dependent empty launches on one stream
  hipEventRecord(e0);
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(40), dim3(256), 0, 0, buf);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("2000 back-to-back empty launches (40 blocks): %.3f us each\n", ms * 1e3f / 2000);
  return 0;
}
