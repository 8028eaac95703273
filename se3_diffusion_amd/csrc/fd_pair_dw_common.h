// Shared pieces of the grouped pair-row weight-gradient kernels (fd_pair_dw.hip; tools/experiments/fd_pair_dw_diag.hip):
// stage geometry, the fp32 -> 3 x bf16 split of a float4, the transposed LDS operand read.
#pragma once
#include "fd_common.h"

namespace {


constexpr int DW_THREADS = 512;
constexpr int DW_KS = 16;                          // pair rows per stage = one MFMA k-step
constexpr int DW_PSTRIDE = DW_KS * 64 + 64;        // bytes of a 32-column panel of one plane (+64: the two 8-lane halves
                                                   // of a ds_write_b64 group land on different bank halves)
constexpr int DW_PANELS = 16;                      // 12 of A (384 columns) + 4 of B (128 columns)
constexpr int DW_PLANE = DW_PANELS * DW_PSTRIDE;
constexpr int DW_STAGE = 3 * DW_PLANE;             // 52,224 B
constexpr int DW_RING = 2;
static_assert(DW_RING * DW_STAGE <= 160 * 1024, "LDS");

// four consecutive fp32 of one row -> three bf16 planes (x = p0 + p1 + p2 exactly, round-to-nearest at every stage)
__device__ __forceinline__ void dw_split4(const float4 v, uint2& s0, uint2& s1, uint2& s2) {
  const unsigned h0 = fd::pack_bf16(v.x, v.y), h1 = fd::pack_bf16(v.z, v.w);
  const float r0 = v.x - fd::bf16lo_f32(h0), r1 = v.y - fd::bf16hi_f32(h0);
  const float r2 = v.z - fd::bf16lo_f32(h1), r3 = v.w - fd::bf16hi_f32(h1);
  const unsigned m0 = fd::pack_bf16(r0, r1), m1 = fd::pack_bf16(r2, r3);
  const float q0 = r0 - fd::bf16lo_f32(m0), q1 = r1 - fd::bf16hi_f32(m0);
  const float q2 = r2 - fd::bf16lo_f32(m1), q3 = r3 - fd::bf16hi_f32(m1);
  s0 = make_uint2(h0, h1);
  s1 = make_uint2(m0, m1);
  s2 = make_uint2(fd::pack_bf16(q0, q1), fd::pack_bf16(q2, q3));
}

// the 8 consecutive k of one column: k 8kg..8kg+3 and 8kg+4..8kg+7 (four 64-byte rows further)
__device__ __forceinline__ uint4 dw_read8(const char* p) {
  const uint2 lo = fd::lds_read_tr16(p), hi = fd::lds_read_tr16(p + 256);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// the item's pointers come out of a dynamically indexed kernel-argument array: tell the compiler they are global
// (global_load / global_atomic instead of flat_*)
template <typename T>
__device__ __forceinline__ T* dw_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}

}  // namespace
