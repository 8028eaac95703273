// fd_pair_dw: the weight gradients of the pair-row MLPs (EdgeTransition, model/ipa_pytorch.py:194-233, autograd of its
// three Linear layers) as ONE grouped launch on MI355X (gfx950).
//
//   C_t[m, n] += sum_p A_t[p, m] * B_t[p, n]        t = 0 .. nitems-1,  p over the B*N*N pair rows
//
// Every item is a 384 x 128 output tile (m over the 384 columns of A_t, n over 128 columns of B_t); the 384 x 384
// gradient of the middle layer is three items that share A.  Both operands are row-major [rows, features] fp32
// tensors as the fused forward / backward kernels left them, so the reduction index p is the STRIDED one of both --
// the shape fd_gemm's split kernel stages with 8 dword loads per 8-k slot.  Here:
//   * a thread loads float4s along the feature axis (whole 128-byte lines per 8 lanes), splits them into the three
//     bf16 planes (fp32-accurate split-bf16 arithmetic, as fd_gemm tile 4) and writes each plane as ds_write_b64
//     into a [panel of 32 columns][k][32] image;
//   * the MFMA operands are read back with ds_read_b64_tr_b16 (the LDS transpose read of gfx950): two reads give a
//     lane its column's 8 consecutive k of a v_mfma_f32_32x32x16_bf16 operand -- no transposition in registers;
//   * all 8 waves of the block stage AND multiply (wave tile 96 x 64, 36 MFMAs per 16-k stage and wave), two-stage
//     LDS ring, one barrier per stage, global loads two stages ahead in registers;
//   * a block owns ONE item and one contiguous range of pair rows for its whole life: one prologue, one flush of the
//     384 x 128 accumulator tile with atomics (C accumulates; the caller's gradient buffer).  The blocks of a group
//     (the items of one row range) are placed on one XCD and walk the same rows in lockstep, so an operand shared by
//     several items (the three tiles of the middle layer) is fetched from HBM once and served by that XCD's L2.
// Options per item: A_add [rows,128] is added to columns 0..127 of A while staging (final_layer: its input is
// h2 + [z | e_i | e_j], so dWf[:, 0:128] = dy^T (h2[:, 0:128] + z) in the same pass); a_colsum accumulates the column
// sums of A (the bias gradient); trans stores C[n, m].
#include "fd_common.h"
#include "fd_pair_dw_common.h"
#include "../../include/fd_hip.h"

namespace {

// NA = 128-column bands of A (3: the 384 x 128 tile; 1: a 128 x 128 tile -- the 128-wide layers of the edge embedder --
// where wave (wm, wn) owns ONE A panel and 12 MFMAs per stage)
template <int NA, bool TRANS, bool HAS_ADD, bool HAS_CS>
__device__ __forceinline__ void dw_block(const FdPairDwItem& it, long row0, long row1, char* lds) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging slots: thread -> row k = tid / 32 of the 16-row stage, float4 column c = tid % 32 of the three
  // 128-column bands of A (slots 0..2) and of B (slot 3): 32 lanes cover 512 contiguous bytes of a row ----
  const int kk = tid >> 5, c4 = tid & 31;
  int wofs[4];
#pragma unroll
  for (int i = 0; i < 3; ++i) wofs[i] = (4 * i + (c4 >> 3)) * DW_PSTRIDE + kk * 64 + (c4 & 7) * 8;
  wofs[3] = (12 + (c4 >> 3)) * DW_PSTRIDE + kk * 64 + (c4 & 7) * 8;
  constexpr bool has_add = HAS_ADD, has_cs = HAS_CS;

  // B with fewer than 128 columns (b_cols: the 120 input features of the embedder's first layer): the lanes of the missing
  // columns read column 0 instead and stage zeros; their C columns are not written
  const int nb = it.b_cols > 0 ? it.b_cols : 128;
  const bool bok = 4 * c4 < nb;
  const int cb = bok ? 4 * c4 : 0;
  const float* __restrict__ A = dw_global(it.A) + row0 * it.lda;
  const float* __restrict__ B = dw_global(it.B) + row0 * it.ldb;
  const float* __restrict__ Ad = has_add ? dw_global(it.A_add) + row0 * it.ld_add : A;
  float* __restrict__ C = dw_global(it.C);
  float* __restrict__ colsum = dw_global(it.a_colsum);
  const long nrows = row1 - row0;
  const int nst = (int)((nrows + DW_KS - 1) / DW_KS);

  float4 rg[2][4], radd[2];      // slots 0 .. NA-1: the bands of A, slot 3: B
  float csum[NA][4];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) csum[i][e] = 0.f;

  // Branch-free: rows past the end of the range are read from its last row (a valid address) and replaced by zeros, so
  // the stage loop below is one basic block whatever the stage -- the scheduler interleaves the split / LDS writes of
  // stage s + 1 and the loads of stage s + 3 with the MFMAs of stage s.
  const long last = nrows - 1;
  auto load = [&](float4 (&r)[4], float4& ra, int st) __attribute__((always_inline)) {
    const long k = (long)st * DW_KS + kk;
    const bool ok = k <= last;
    const long kc = ok ? k : last;
    const float* a = A + kc * it.lda + 4 * c4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(a + 128 * i);
      r[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    {
      const float4 v = *reinterpret_cast<const float4*>(B + kc * it.ldb + cb);
      const bool okb = ok && bok;
      r[3] = make_float4(okb ? v.x : 0.f, okb ? v.y : 0.f, okb ? v.z : 0.f, okb ? v.w : 0.f);
    }
    if (has_add) {
      const float4 v = *reinterpret_cast<const float4*>(Ad + kc * it.ld_add + 4 * c4);
      ra = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
  };
  auto put = [&](float4 (&r)[4], const float4& ra, char* dst) __attribute__((always_inline)) {
    if (has_add) { r[0].x += ra.x; r[0].y += ra.y; r[0].z += ra.z; r[0].w += ra.w; }
    if (has_cs) {
#pragma unroll
      for (int i = 0; i < NA; ++i) { csum[i][0] += r[i].x; csum[i][1] += r[i].y; csum[i][2] += r[i].z; csum[i][3] += r[i].w; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= NA && i != 3) continue;
      uint2 s0, s1, s2;
      dw_split4(r[i], s0, s1, s2);
      *reinterpret_cast<uint2*>(dst + wofs[i]) = s0;
      *reinterpret_cast<uint2*>(dst + DW_PLANE + wofs[i]) = s1;
      *reinterpret_cast<uint2*>(dst + 2 * DW_PLANE + wofs[i]) = s2;
    }
  };

  // ---- MFMA side: wave (wm, wn) owns A panels NA wm .. NA wm + NA - 1 and B panels 2wn, 2wn+1 ----
  const int i16 = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5;
  const int lofs = (8 * kg + (i16 >> 2)) * 64 + half * 32 + (i16 & 3) * 8;
  const int a_rd = (NA * wm) * DW_PSTRIDE + lofs, b_rd = (12 + 2 * wn) * DW_PSTRIDE + lofs;
  f32x16 acc[NA][2];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&](const char* st) __attribute__((always_inline)) {
    uint4 fb[2][3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j][s] = dw_read8(st + b_rd + j * DW_PSTRIDE + s * DW_PLANE);
#pragma unroll
    for (int sa = 2; sa >= 0; --sa) {   // the small terms first
      uint4 fa[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) fa[i] = dw_read8(st + a_rd + i * DW_PSTRIDE + sa * DW_PLANE);
#pragma unroll
      for (int sb = 2 - sa; sb >= 0; --sb)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = TRANS ? fd::mfma_32x32x16_bf16(fb[j][sb], fa[i], acc[i][j])
                              : fd::mfma_32x32x16_bf16(fa[i], fb[j][sb], acc[i][j]);
    }
  };

  // Stages that lie entirely inside the row range are loaded through running pointers without the clamp / select of
  // load() (it costs ~35 of the ~135 VALU instructions a wave spends per stage, six of them quarter-rate 32-bit
  // multiplies, and the SIMD's issue port is shared with the MFMAs): the pointers always stand at the next stage.
  const float* pa = A + ((long)3 * DW_KS + kk) * it.lda + 4 * c4;
  const float* pb = B + ((long)3 * DW_KS + kk) * it.ldb + cb;
  const float* pd = Ad + ((long)3 * DW_KS + kk) * it.ld_add + 4 * c4;
  const long sa = (long)DW_KS * it.lda, sb = (long)DW_KS * it.ldb, sd = (long)DW_KS * it.ld_add;
  auto load_fast = [&](float4 (&r)[4], float4& ra) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) r[i] = *reinterpret_cast<const float4*>(pa + 128 * i);
    {
      const float4 v = *reinterpret_cast<const float4*>(pb);
      r[3] = make_float4(bok ? v.x : 0.f, bok ? v.y : 0.f, bok ? v.z : 0.f, bok ? v.w : 0.f);
    }
    if (has_add) ra = *reinterpret_cast<const float4*>(pd);
    pa += sa;
    pb += sb;
    if (has_add) pd += sd;
  };

  // ---- pipeline: stage s lives in register set s & 1 (loaded two stages ahead) and ring slot s & 1 ----
  load(rg[0], radd[0], 0);
  load(rg[1], radd[1], 1);
  put(rg[0], radd[0], lds);
  load(rg[0], radd[0], 2);
  __syncthreads();
  // multiply first, stage second: the compiler fills the MFMA shadow with the split of the next stage
  auto step_fast = [&](int s, float4 (&r)[4], float4& ra) __attribute__((always_inline)) {
    // r / ra: the register set of stage s + 1, refilled with stage s + 3
    mma(lds + (s & 1) * DW_STAGE);
    put(r, ra, lds + ((s + 1) & 1) * DW_STAGE);
    load_fast(r, ra);
    __syncthreads();
  };
  auto step = [&](int s, float4 (&r)[4], float4& ra) __attribute__((always_inline)) {
    mma(lds + (s & 1) * DW_STAGE);
    put(r, ra, lds + ((s + 1) & 1) * DW_STAGE);
    load(r, ra, s + 3);          // (stages past the end are zeros: see load)
    __syncthreads();
  };
  const int nfull = (int)(nrows / DW_KS);      // stages 0 .. nfull-1 are entirely inside the range
  int s = 0;
  for (; s + 5 <= nfull; s += 2) {             // both refills (stages s + 3, s + 4) are full stages
    step_fast(s, rg[1], radd[1]);
    step_fast(s + 1, rg[0], radd[0]);
  }
  for (; s < nst; s += 2) {
    step(s, rg[1], radd[1]);
    step(s + 1, rg[0], radd[0]);   // nst odd: one stage of zeros
  }

  // ---- flush: C (+)= acc, atomically (every row range adds its part) ----
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        // !TRANS: D[row = m][col = n]; TRANS: D[row = n][col = m]
        const int m = (NA * wm + i) * 32 + (TRANS ? l31 : rr);
        const int n = (2 * wn + j) * 32 + (TRANS ? rr : l31);
        float* c = TRANS ? C + (long)n * it.ldc + m : C + (long)m * it.ldc + n;
        if (n < nb) atomicAdd(c, acc[i][j][r]);
      }
  if (has_cs) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(colsum + 4 * (c4 + 32 * i) + e, csum[i][e]);
  }
}

__global__ __launch_bounds__(DW_THREADS, 1) void pair_dw_kernel(FdPairDwDesc d) {
  __shared__ __attribute__((aligned(16))) char lds[DW_RING * DW_STAGE];
  // logical block order: XCD-major (block b runs on XCD b % 8), so the items of a row range sit on one XCD
  const int G = (int)gridDim.x, per = G >> 3;
  const int L = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  const int ngroups = G / d.nitems;
  const int group = L / d.nitems, item = L - group * d.nitems;
  if (group >= ngroups) return;
  // row ranges in whole stages
  const long nst = (d.rows + DW_KS - 1) / DW_KS;
  const long s0 = nst * group / ngroups, s1 = nst * (group + 1) / ngroups;
  const long row0 = s0 * DW_KS, row1 = (s1 * DW_KS < d.rows) ? s1 * DW_KS : d.rows;
  if (row0 >= row1) return;
  // (a select chain, not d.item[item]: a dynamic index would spill the argument array to scratch)
  FdPairDwItem it = d.item[0];
#pragma unroll
  for (int t = 1; t < FD_PAIR_DW_MAX_ITEMS; ++t)
    if (item == t) it = d.item[t];
  if (it.a_bands == 1) {   // 128 x 128 tile: plain or with the bias gradient
    if (it.a_colsum)
      dw_block<1, false, false, true>(it, row0, row1, lds);
    else
      dw_block<1, false, false, false>(it, row0, row1, lds);
    return;
  }
  const int mode = (it.trans ? 4 : 0) | (it.A_add ? 2 : 0) | (it.a_colsum ? 1 : 0);
  switch (mode) {
    case 0: dw_block<3, false, false, false>(it, row0, row1, lds); break;
    case 1: dw_block<3, false, false, true>(it, row0, row1, lds); break;
    case 2: dw_block<3, false, true, false>(it, row0, row1, lds); break;
    case 3: dw_block<3, false, true, true>(it, row0, row1, lds); break;
    case 4: dw_block<3, true, false, false>(it, row0, row1, lds); break;
    case 5: dw_block<3, true, false, true>(it, row0, row1, lds); break;
    case 6: dw_block<3, true, true, false>(it, row0, row1, lds); break;
    default: dw_block<3, true, true, true>(it, row0, row1, lds); break;
  }
}

}  // namespace

extern "C" int fd_pair_dw(const FdPairDwDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FD_CHECK_ARG(desc != nullptr, "fd_pair_dw: null descriptor");
  const FdPairDwDesc& d = *desc;
  FD_CHECK_ARG(d.nitems >= 1 && d.nitems <= FD_PAIR_DW_MAX_ITEMS, "fd_pair_dw: 1..%d items", FD_PAIR_DW_MAX_ITEMS);
  FD_CHECK_ARG(d.rows >= 0, "fd_pair_dw: negative row count");
  if (d.rows == 0) return FD_OK;
  for (int t = 0; t < d.nitems; ++t) {
    const FdPairDwItem& it = d.item[t];
    FD_CHECK_ARG(it.A && it.B && it.C, "fd_pair_dw: item %d: null operand", t);
    FD_CHECK_ARG(it.a_bands == 0 || it.a_bands == 1 || it.a_bands == 3, "fd_pair_dw: item %d: a_bands must be 1 or 3", t);
    const int am = it.a_bands == 1 ? 128 : 384;
    FD_CHECK_ARG(it.a_bands != 1 || (!it.trans && !it.A_add), "fd_pair_dw: item %d: a 128-column A takes no trans / A_add", t);
    FD_CHECK_ARG(fd_aligned16(it.A) && fd_aligned16(it.B) && (it.lda & 3) == 0 && (it.ldb & 3) == 0 && it.lda >= am,
                 "fd_pair_dw: item %d: A [rows,%d] / B [rows,128] must be 16-byte aligned with row strides %% 4 == 0", t, am);
    FD_CHECK_ARG(!it.A_add || (fd_aligned16(it.A_add) && (it.ld_add & 3) == 0 && it.ld_add >= 128),
                 "fd_pair_dw: item %d: A_add [rows,128] must be 16-byte aligned with a row stride %% 4 == 0", t);
    FD_CHECK_ARG(it.b_cols >= 0 && it.b_cols <= 128 && (it.b_cols & 3) == 0, "fd_pair_dw: item %d: b_cols must be 0 (= 128) or a multiple of 4 up to 128", t);
    FD_CHECK_ARG(it.ldb >= (it.b_cols > 0 ? it.b_cols : 128), "fd_pair_dw: item %d: ldb too small", t);
    FD_CHECK_ARG(it.ldc >= (it.trans ? am : (it.b_cols > 0 ? it.b_cols : 128)), "fd_pair_dw: item %d: ldc too small", t);
  }
  int blocks = d.blocks > 0 ? d.blocks : 256;   // MI355X: one persistent block per CU
  blocks &= ~7;
  if (blocks < 8 * ((d.nitems + 7) / 8)) blocks = 8 * ((d.nitems + 7) / 8);
  // no more groups than 16-row stages
  const long nst = (d.rows + DW_KS - 1) / DW_KS;
  while (blocks > 8 && (long)(blocks / d.nitems) > nst) blocks -= 8;
  hipLaunchKernelGGL(pair_dw_kernel, dim3(blocks), dim3(DW_THREADS), 0, stream, d);
  FD_CHECK_LAUNCH("fd_pair_dw");
  return FD_OK;
}
