// Probe: how busy can TWO waves per SIMD keep the matrix pipe on the fused edge transition's instruction mix, when the two waves
// of a SIMD are (1) in lockstep, (2) deliberately in opposite phases ("ping-pong": one multiplies a weight unit while its partner
// reads the next unit's fragments from LDS, splits activations and issues the LDS-DMA copies), (0) MFMA only.
// Skeleton of csrc/fd_edge_mlp.hip's dataflow without its epilogues: a wave owns 16 rows, 96 accumulator registers, per weight
// unit (12 KB = 4 n-blocks x one 32-k step x 3 bf16 planes): 12 ds_read_b128 + 24 v_mfma_f32_16x16x32_bf16; every 6th unit a
// 3-way bf16 split of 8 values (44 VALU); the weight image streams through an LDS ring by LDS-DMA.  One 8-wave block per CU.
//   hipcc --offload-arch=gfx950 -O3 pp_probe.hip -o pp_probe && ./pp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PIECE = 1024, UNIT = 12 * PIECE, NUNITS = 128;
#ifndef RING
#define RING 5
#endif
#ifndef AHEAD
#define AHEAD 4
#endif

__device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void split8(const float (&x)[8], uint4& s0, uint4& s1, uint4& s2) {
  unsigned t0[4], t1[4], t2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = x[2 * j], v = x[2 * j + 1];
    const unsigned hh = pack_bf16(u, v);
    const float ru = u - __builtin_bit_cast(float, hh << 16), rv = v - __builtin_bit_cast(float, hh & 0xffff0000u);
    const unsigned mm = pack_bf16(ru, rv);
    const float qu = ru - __builtin_bit_cast(float, mm << 16), qv = rv - __builtin_bit_cast(float, mm & 0xffff0000u);
    t0[j] = hh; t1[j] = mm; t2[j] = pack_bf16(qu, qv);
  }
  s0 = make_uint4(t0[0], t0[1], t0[2], t0[3]);
  s1 = make_uint4(t1[0], t1[1], t1[2], t1[3]);
  s2 = make_uint4(t2[0], t2[1], t2[2], t2[3]);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
// one LDS-DMA piece (1 KB): global lane address -> LDS wave-uniform base + 16 * lane
__device__ __forceinline__ void glds16(const void* g_lane, void* lds_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_base));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g_lane), "s"(dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

// MODE 0: MFMA only.  MODE 1: lockstep (other(u); barrier; mfma(u); barrier -- all 8 waves the same).  MODE 2: ping-pong (second
// wave of every SIMD delayed by one interval).  MODE 3: free-running -- the shipped structure's order (reads of the next unit in
// front of the current unit's MFMAs, one barrier per unit) with 8 waves in one block.  MODE 4 (round 5): mode 3 with the COLUMN SPLIT at full
// size -- a wave owns 32 rows x half the columns: per unit it reads its half-unit (6 ds_read_b128 instead of 12) and feeds every
// fragment to two MFMAs (two 16-row tiles), two activation splits every sixth unit: what halving the fragment reads per MFMA buys.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const char* __restrict__ img, float* __restrict__ out, int tiles) {
  __shared__ __attribute__((aligned(16))) char ring[RING * UNIT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // role: second wave of its SIMD?  (a workgroup's waves go to the SIMDs in cyclic order, so waves w and w + 4 share one)
  const bool second = wave >= 4;
  f32x4 acc[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = 0.01f * (lane + e) + 0.5f;
  uint4 b[3], b2[3], H[12];
  split8(x, b[0], b[1], b[2]);
  split8(x, b2[0], b2[1], b2[2]);
  const int total = tiles * NUNITS;
  // pieces of unit u issued by this wave: piece p by wave p % 8 (waves 0..3: two pieces, waves 4..7: one)
  int iu = 0, islot = 0, isrc = 0;        // next unit to copy, its ring slot, its index in the image
  auto issue = [&]() __attribute__((always_inline)) {
    if (iu < total) {
      const char* src = img + isrc * UNIT + lane * 16;
      char* dst = ring + islot * UNIT;
      glds16(src + wave * PIECE, dst + wave * PIECE);
      if (wave < 4) glds16(src + (8 + wave) * PIECE, dst + (8 + wave) * PIECE);
    }
    ++iu;
    islot = islot + 1 == RING ? 0 : islot + 1;
    isrc = isrc + 1 == NUNITS ? 0 : isrc + 1;
  };
  int rslot = 0;
  auto read_unit = [&]() __attribute__((always_inline)) {
    const char* s = ring + rslot * UNIT + lane * 16;
    rslot = rslot + 1 == RING ? 0 : rslot + 1;
#pragma unroll
    for (int i = 0; i < 12; ++i) H[i] = *reinterpret_cast<const uint4*>(s + i * PIECE);
  };
  auto mma_unit = [&](int a0) __attribute__((always_inline)) {
    constexpr int PW[6] = {2, 1, 0, 1, 0, 0}, PX[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[a0 + nb] = mfma(H[3 * nb + PW[p]], b[PX[p]], acc[a0 + nb]);
  };
  auto read_half = [&]() __attribute__((always_inline)) {
    const char* s = ring + rslot * UNIT + ((wave & 1) * 6) * PIECE + lane * 16;
    rslot = rslot + 1 == RING ? 0 : rslot + 1;
#pragma unroll
    for (int i = 0; i < 6; ++i) H[i] = *reinterpret_cast<const uint4*>(s + i * PIECE);
  };
  auto mma_half2 = [&](int a0) __attribute__((always_inline)) {
    constexpr int PW[6] = {2, 1, 0, 1, 0, 0}, PX[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        acc[a0 + 2 * nb] = mfma(H[3 * nb + PW[p]], b[PX[p]], acc[a0 + 2 * nb]);
        acc[a0 + 2 * nb + 1] = mfma(H[3 * nb + PW[p]], b2[PX[p]], acc[a0 + 2 * nb + 1]);
      }
  };
  if (MODE == 0) {
    issue();
    wait_vm<0>();
    __syncthreads();
    read_unit();
    for (int u0 = 0; u0 < total; u0 += 6) {
#pragma unroll
      for (int g = 0; g < 6; ++g) mma_unit(4 * g);
    }
  } else {
    for (int a = 0; a < AHEAD; ++a) issue();
    wait_vm<0>();
    __syncthreads();
    if (MODE == 2 && second) __builtin_amdgcn_s_barrier();   // the delayed half
    for (int u0 = 0; u0 < total; u0 += 6) {
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        // ---- other(u) ----
        pin();
        if (MODE == 4) read_half(); else read_unit();
        if (g == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = acc[e >> 2][e & 3] * 0.5f + x[e] * 0.25f;
          split8(x, b[0], b[1], b[2]);
          if (MODE == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = acc[2 + (e >> 2)][e & 3] * 0.5f + x[e] * 0.25f;
            split8(x, b2[0], b2[1], b2[2]);
          }
        }
        issue();
        // pieces of unit u + 1 (issued AHEAD - 1 segments ago) have landed; the younger ones stay in flight
        if (wave < 4) wait_vm<2 * (AHEAD - 1)>(); else wait_vm<AHEAD - 1>();
        pin();
        if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_barrier();
        // ---- mfma(u) ----
        pin();
        if (MODE == 4) mma_half2(4 * g); else mma_unit(4 * g);
        pin();
        __builtin_amdgcn_s_barrier();
      }
    }
    if (MODE == 2 && !second) __builtin_amdgcn_s_barrier();
    wait_vm<0>();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + tid] = s + x[0];
}

int main() {
  const int blocks = 256, tiles = 30;      // 30 tiles of 128 units per wave ~ the B=30 x N=128 launch (491,520 rows / 16 / 2048 waves ... x2)
  char* img; float* out;
  hipMalloc(&img, NUNITS * UNIT); hipMalloc(&out, blocks * 512 * 4);
  std::vector<unsigned> h(NUNITS * UNIT / 4);
  for (auto& w : h) { unsigned lo = 0x3c00 | (rand() & 0x80ff), hi = 0x3c00 | (rand() & 0x80ff); w = lo | (hi << 16); }
  hipMemcpy(img, h.data(), NUNITS * UNIT, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[5] = {"mfma only          ", "lockstep (2 barr)  ", "ping-pong          ", "one barrier / unit ", "column split x 2 rt"};
  float ref = 0.f;
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<blocks, 512>>>(img, out, tiles);
      if (mode == 1) k<1><<<blocks, 512>>>(img, out, tiles);
      if (mode == 2) k<2><<<blocks, 512>>>(img, out, tiles);
      if (mode == 3) k<3><<<blocks, 512>>>(img, out, tiles);
      if (mode == 4) k<4><<<blocks, 512>>>(img, out, tiles);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
      const double nm = (double)blocks * 8 * tiles * NUNITS * 24;          // MFMAs
      const double tf = nm * 2.0 * 16 * 16 * 32 / ms / 1e9;
      if (mode == 0 && rep == 2) ref = ms;
      printf("%s %.3f ms  %.0f TFLOP/s bf16  (%.1f %% of the 2500 nominal; %.2f x the MFMA-only time)  [RING %d AHEAD %d]\n",
             names[mode], ms, tf, tf / 25.0, ref > 0 ? ms / ref : 1.0, RING, AHEAD);
    }
  }
  return 0;
}
