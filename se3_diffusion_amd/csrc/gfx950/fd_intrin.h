// gfx950 (CDNA4) intrinsics used by the FrameDiff kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#define FD_BACKEND_NAME "gfx950"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace fd {

// 16 bytes moved as one register quad (global_load_dwordx4 / ds_write_b128) without HIP's uint4 struct semantics
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: exact f32 (k-ordered fmaf chain), 64 cycles / SIMD.
//   A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r -> row 4*(l>>4)+r, col l&15
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], e = 0..7 packed two per dword
// (low half first); exact bf16 products, f32 accumulate; D as 32x32x2.  32 cycles / SIMD.
typedef __bf16 fd_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fd_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fd_bf16x8, a), __builtin_bit_cast(fd_bf16x8, b), c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15]; D reg r -> row 4*(l>>4)+r, col l&15.
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fd_bf16x8, a), __builtin_bit_cast(fd_bf16x8, b), c, 0, 0, 0);
}
// two f32 -> one dword of two bf16, round-to-nearest-even (v_cvt_pk_bf16_f32): lo in bits 0..15
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  // (asm rather than two __bf16 casts: the compiler otherwise re-derives each half with its own conversion when
  // the packed word is unpacked again by the split code)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf16lo_f32(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16hi_f32(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane straight from global memory into LDS, no VGPR round trip.
// The LDS destination is WAVE-UNIFORM base + lane * 16 (not a per-lane scatter); the source address is per lane.
// Completion is tracked by vmcnt like any load: __syncthreads() (vmcnt(0) + barrier) publishes it to the block.
// OFF (0..4095) is the instruction's immediate offset, applied to BOTH addresses: pieces at src + OFF -> dst + OFF share
// one M0 / address setup.
template <int OFF = 0>
__device__ __forceinline__ void glds16(const void* g_lane, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}

// s_setprio 3: this wave wins instruction arbitration on its SIMD
__device__ __forceinline__ void raise_wave_priority() { __builtin_amdgcn_s_setprio(3); }

// s_barrier without the LDS/memory fence of __syncthreads(): for a wave that has nothing to publish at the barrier
__device__ __forceinline__ void block_barrier_nofence() { __builtin_amdgcn_s_barrier(); }

// compiler-only fence: memory operations are not moved across it (bounds register live ranges in unrolled epilogues)
__device__ __forceinline__ void sched_fence() { asm volatile("" ::: "memory"); }

// Four LDS-DMA pieces (src + k KB -> dst + k KB, k = 0..3) behind ONE M0 setup, as inline asm: the compiler does not
// see these loads, so it neither drains them with a vmcnt(0) in front of the next ds_read (it cannot prove that an LDS
// read does not alias a pending LDS-DMA write) nor waits for them at __syncthreads().  The caller owns the wait:
// wait_vmem() before the barrier that publishes the copy.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
// ONE LDS-DMA piece (16 bytes per lane, lane-private source address, wave-uniform destination + 16 lane) as inline asm:
// like glds16x4 the compiler does not see the copy (no vmcnt(0) in front of the next ds_read, nothing at __syncthreads());
// the caller waits with wait_vmem() before the barrier that publishes it.  The vmcnt counter retires in issue order, so
// the compiler's own waits for loads issued AFTER this copy also cover it -- stricter than needed, never too weak.
__device__ __forceinline__ void glds16a(const void* g_lane, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_wave_base));
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g_lane), "s"(dst)
      : "memory");
}
__device__ __forceinline__ void glds16x4(const void* g_lane, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_wave_base));
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g_lane), "s"(dst)
      : "memory");
}
__device__ __forceinline__ void glds16x3(const void* g_lane, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_wave_base));
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g_lane), "s"(dst)
      : "memory");
}
__device__ __forceinline__ void glds16x2(const void* g_lane, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_wave_base));
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g_lane), "s"(dst)
      : "memory");
}
// ds_read_b64_tr_b16: LDS transpose read.  Every lane passes the address of 4 consecutive 16-bit elements (8-byte
// aligned); inside each group of 16 lanes the 16 x 4 elements are transposed: lane i receives element (i & 3) of lanes
// 4j + (i >> 2), j = 0..3.  With a [k][16 columns] image whose row k is covered by lanes 4k..4k+3 of the group, lane i
// gets column i for k = 0..3 -- the k-contiguous MFMA operand of a row-major (k-strided) tile.
typedef short fd_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_read_tr16(const void* p) {
  const fd_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (fd_s16x4 __attribute__((address_space(3)))*)(const __attribute__((address_space(3))) char*)p);
  return __builtin_bit_cast(uint2, v);
}

// s_waitcnt vmcnt(0): every vector-memory operation of this wave (loads, stores, LDS-DMA) has completed
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// s_waitcnt vmcnt(6): at most the six most recently issued vector-memory operations of this wave are still outstanding
// (gfx9 retires loads, stores and LDS-DMA in issue order on this counter): with the six LDS-DMA instructions of the NEXT
// stage issued last, the current stage's copy has landed while the next one stays in flight
__device__ __forceinline__ void wait_vmem_keep6() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmem_keep() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the instruction scheduler moves nothing across this point (pins a software-pipelined order)
__device__ __forceinline__ void sched_pin() { __builtin_amdgcn_sched_barrier(0); }

// one group of an instruction-scheduling pipeline (llvm.amdgcn.sched.group.barrier): the next SIZE instructions of class MASK
// (0x008 MFMA, 0x100 LDS read, 0x002 VALU, 0x020 VMEM read) in program order of the groups declared in this scheduling region
template <int MASK, int SIZE>
__device__ __forceinline__ void sched_group() { __builtin_amdgcn_sched_group_barrier(MASK, SIZE, 0); }

// a value the caller knows to be the same in every lane of the wave, as a scalar (v_readfirstlane): what is derived from it
// (per-head pointers, loop bounds) stays in SGPRs
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 16-byte store with the non-temporal hint (global_store_dwordx4 ... nt): data nobody re-reads before it has left the L2
__device__ __forceinline__ void store_nt4(float* p, float x, float y, float z, float w) {
  f32x4 v;
  v[0] = x; v[1] = y; v[2] = z; v[3] = w;
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// LDS float accumulate without a return value (ds_add_f32): p must point into __shared__ memory
__device__ __forceinline__ void lds_add(float* p, float v) {
  __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
}

// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), the total in every lane: four data-parallel-primitive adds
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), no LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  return v;
}
// the value of lane l ^ 8 (row_ror:8 within the 16-lane DPP row): no LDS crossbar
__device__ __forceinline__ float lane_xor8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
}
// float sums over the wave: the 16-lane rows by DPP, the four rows by two LDS-crossbar exchanges (6 exchanges in the generic
// form below -- a kernel that reduces 26 values per head spends more LDS instructions on them than on its data)
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o); v = v > u ? v : u; }
  return v;
}

}  // namespace fd
