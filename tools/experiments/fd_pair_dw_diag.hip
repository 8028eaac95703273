// EXPERIMENT (not part of libfd_hip.so): block-diagonal form of the grouped pair-row weight-gradient kernel -- the three
// 128-wide layers of the edge embedder in one pass over the pair rows.  Correct on gfx950 (round 2) and SLOWER inside the
// training step than the fd_gemm launches it replaces (26.1 vs 25.3 ms per step): six operand streams per stage put the VALU
// split, the LDS pipe and the MFMAs at about the same time with nothing overlapping.  Kept for the record; built by
// tools/experiments/build.py into tools/experiments/libfd_experiments.so.
#include "fd_common.h"
#include "fd_pair_dw_common.h"
#include "fd_experiments.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Block-diagonal form: three INDEPENDENT 128 x 128 products over the same pair rows in one pass,
//     C_i[m, n] += sum_p A_i[p, m] * B_i[p, n]      i = 0, 1, 2;  m < 128;  n < b_cols_i <= 128
// -- the three Linear layers of the edge embedder (score_network.py:67-86: dW4 = dh3^T h2, dW2 = dh2^T h1, dW0 = dh1^T x with
// x the 120 input features).  Same staging, split and LDS images as dw_block (12 A panels + 12 B panels: 78 KB per stage,
// two stages), the same 36 MFMAs per stage and wave as the 384 x 128 tile: wave w owns, in EVERY band, A panel w / 2 and the
// B panels 2 (w % 2), 2 (w % 2) + 1.  A block owns one contiguous row range and all three bands.
constexpr int DD_PANELS = 24;
constexpr int DD_PLANE = DD_PANELS * DW_PSTRIDE;
constexpr int DD_STAGE = 3 * DD_PLANE;             // 78,336 B
static_assert(DW_RING * DD_STAGE <= 160 * 1024, "LDS");

__global__ __launch_bounds__(DW_THREADS, 1) void pair_dw_diag_kernel(FdPairDwDiagDesc d) {
  __shared__ __attribute__((aligned(16))) char lds[DW_RING * DD_STAGE];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = (int)gridDim.x;
  const long nst_all = (d.rows + DW_KS - 1) / DW_KS;
  const long s0 = nst_all * blockIdx.x / G, s1 = nst_all * (blockIdx.x + 1) / G;
  const long row0 = s0 * DW_KS, row1 = (s1 * DW_KS < d.rows) ? s1 * DW_KS : d.rows;
  if (row0 >= row1) return;
  const long nrows = row1 - row0, last = nrows - 1;
  const int nst = (int)((nrows + DW_KS - 1) / DW_KS);

  // ---- staging: thread -> row kk of the 16-row stage, float4 column c4 of each band's A and B ----
  const int kk = tid >> 5, c4 = tid & 31;
  const int wsub = (c4 >> 3) * DW_PSTRIDE + kk * 64 + (c4 & 7) * 8;
  const float* A[3];
  const float* B[3];
  long lda[3], ldb[3];
  bool bok[3];
  int nb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    nb[i] = d.b_cols[i] > 0 ? d.b_cols[i] : 128;
    bok[i] = 4 * c4 < nb[i];                     // (the lanes of a missing column read column 0 and stage zeros)
    lda[i] = d.lda[i];
    ldb[i] = d.ldb[i];
    A[i] = dw_global(d.A[i]) + row0 * lda[i] + 4 * c4;
    B[i] = dw_global(d.B[i]) + row0 * ldb[i] + (bok[i] ? 4 * c4 : 0);
  }
  const bool has_cs = d.a_colsum[0] != nullptr;
  float csum[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) csum[i][e] = 0.f;

  float4 rg[2][6];       // slots 0..2: A of band i, 3..5: B of band i
  auto load = [&](float4 (&r)[6], int st) __attribute__((always_inline)) {
    const long k = (long)st * DW_KS + kk;
    const bool ok = k <= last;
    const long kc = ok ? k : last;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 va = *reinterpret_cast<const float4*>(A[i] + kc * lda[i]);
      r[i] = make_float4(ok ? va.x : 0.f, ok ? va.y : 0.f, ok ? va.z : 0.f, ok ? va.w : 0.f);
      const float4 vb = *reinterpret_cast<const float4*>(B[i] + kc * ldb[i]);
      const bool okb = ok && bok[i];
      r[3 + i] = make_float4(okb ? vb.x : 0.f, okb ? vb.y : 0.f, okb ? vb.z : 0.f, okb ? vb.w : 0.f);
    }
  };
  auto put = [&](const float4 (&r)[6], char* dst) __attribute__((always_inline)) {
    if (has_cs) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { csum[i][0] += r[i].x; csum[i][1] += r[i].y; csum[i][2] += r[i].z; csum[i][3] += r[i].w; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      uint2 t0, t1, t2;
      dw_split4(r[i], t0, t1, t2);
      char* o = dst + (i < 3 ? 4 * i : 12 + 4 * (i - 3)) * DW_PSTRIDE + wsub;
      *reinterpret_cast<uint2*>(o) = t0;
      *reinterpret_cast<uint2*>(o + DD_PLANE) = t1;
      *reinterpret_cast<uint2*>(o + 2 * DD_PLANE) = t2;
    }
  };

  // ---- MFMA side ----
  const int i16 = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5;
  const int lofs = (8 * kg + (i16 >> 2)) * 64 + half * 32 + (i16 & 3) * 8;
  const int ap = wave >> 1, bp0 = 2 * (wave & 1);
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mma = [&](const char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const char* pa = st + (4 * i + ap) * DW_PSTRIDE + lofs;
      const char* pb = st + (12 + 4 * i + bp0) * DW_PSTRIDE + lofs;
      uint4 fb[2][3];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j][s] = dw_read8(pb + j * DW_PSTRIDE + s * DD_PLANE);
#pragma unroll
      for (int sa = 2; sa >= 0; --sa) {   // the small terms first
        const uint4 fa = dw_read8(pa + sa * DD_PLANE);
#pragma unroll
        for (int sb = 2 - sa; sb >= 0; --sb)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = fd::mfma_32x32x16_bf16(fa, fb[j][sb], acc[i][j]);
      }
    }
  };

  // ---- pipeline: as dw_block (stage s in register set s & 1, loaded two stages ahead, ring slot s & 1) ----
  load(rg[0], 0);
  load(rg[1], 1);
  put(rg[0], lds);
  load(rg[0], 2);
  __syncthreads();
  for (int s = 0; s < nst; s += 2) {
    mma(lds + (s & 1) * DD_STAGE);
    put(rg[1], lds + ((s + 1) & 1) * DD_STAGE);
    load(rg[1], s + 3);
    __syncthreads();
    mma(lds + ((s + 1) & 1) * DD_STAGE);     // (nst odd: one stage of zeros)
    put(rg[0], lds + (s & 1) * DD_STAGE);
    load(rg[0], s + 4);
    __syncthreads();
  }

  // ---- flush ----
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float* C = dw_global(d.C[i]);
    const long ldc = d.ldc[i];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = ap * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int n = (bp0 + j) * 32 + l31;
        if (n < nb[i]) atomicAdd(C + (long)m * ldc + n, acc[i][j][r]);
      }
    if (has_cs && d.a_colsum[i] != nullptr) {
      float* cs = dw_global(d.a_colsum[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(cs + 4 * c4 + e, csum[i][e]);
    }
  }
}

}  // namespace

extern "C" int fd_pair_dw_diag(const FdPairDwDiagDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FD_CHECK_ARG(desc != nullptr, "fd_pair_dw_diag: null descriptor");
  const FdPairDwDiagDesc& d = *desc;
  FD_CHECK_ARG(d.rows >= 0, "fd_pair_dw_diag: negative row count");
  if (d.rows == 0) return FD_OK;
  for (int i = 0; i < 3; ++i) {
    FD_CHECK_ARG(d.A[i] && d.B[i] && d.C[i], "fd_pair_dw_diag: band %d: null operand", i);
    const int nb = d.b_cols[i] > 0 ? d.b_cols[i] : 128;
    FD_CHECK_ARG(d.b_cols[i] >= 0 && nb <= 128 && (nb & 3) == 0, "fd_pair_dw_diag: band %d: b_cols must be 0 (= 128) or a multiple of 4 up to 128", i);
    FD_CHECK_ARG(fd_aligned16(d.A[i]) && fd_aligned16(d.B[i]) && (d.lda[i] & 3) == 0 && (d.ldb[i] & 3) == 0 && d.lda[i] >= 128 &&
                     d.ldb[i] >= nb && d.ldc[i] >= nb,
                 "fd_pair_dw_diag: band %d: A [rows,128] / B [rows,%d] must be 16-byte aligned with row strides %% 4 == 0", i, nb);
    FD_CHECK_ARG((d.a_colsum[i] != nullptr) == (d.a_colsum[0] != nullptr), "fd_pair_dw_diag: a_colsum for all bands or for none");
  }
  int blocks = d.blocks > 0 ? d.blocks : 256;   // MI355X: one persistent block per CU
  const long nst = (d.rows + DW_KS - 1) / DW_KS;
  if ((long)blocks > nst) blocks = (int)nst;
  hipLaunchKernelGGL(pair_dw_diag_kernel, dim3(blocks), dim3(DW_THREADS), 0, stream, d);
  FD_CHECK_LAUNCH("fd_pair_dw_diag");
  return FD_OK;
}
