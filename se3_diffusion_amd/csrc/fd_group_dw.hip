// fd_group_dw: every independent weight gradient  dW_t = dY_t^T X_t  (+ bias gradient) of a trunk block's node-level
// Linear layers in ONE launch on MI355X (gfx950) -- autograd of the nn.Linear layers of model/ipa_pytorch.py:169-191
// (StructureModuleTransition), :236-301,455-460 (IPA projections / linear_out), :584-595,632-644 (sequence transformer,
// skip embedding, post_tfmr), :194-233 (the per-residue halves of EdgeTransition) and the torsion / embedder heads.
//
//   C_t[m, n] += sum_r A_t[r, m] * B_t[r, n]        t = 0 .. nitems-1,  r over the B*N residue rows,
//   colsum_t[m] += sum_r A_t[r, m]                   (A_t = dY_t [rows, n_out], B_t = X_t [rows, k_in])
//
// These are ~25 GEMMs per trunk block with tiny outputs (64 x 256 .. 6816 x 256) and a short reduction (rows = 3,840 in
// training): as separate split-K launches of fd_gemm's 64 x 64 fp32 tile they cost 17-59 us each, ~100 launches per
// step, and re-read their operands once per split.  Here one persistent grid shares the sequence of
//   (item, 128 x 128 output tile, 16-row stage)
// in equal pieces (a block's piece spans one or two tiles: one prologue and one atomic flush each)
// Both operands are row-major [rows, features] fp32 activations, i.e. the reduction index is the strided one of both:
// the staging of fd_pair_dw (float4 loads along the feature axis, fp32 -> 3 bf16 planes in registers, [panel][k][32] LDS
// images, ds_read_b64_tr_b16 operand reads, 6 x v_mfma_f32_32x32x16_bf16 per fp32-accurate 16-row step) on a 4-wave
// block: wave (wm, wn) owns a 64 x 64 quarter of the tile = 24 MFMAs per stage, two blocks per CU.
// Column tails (n_out, k_in not multiples of 128) are staged as zeros and never stored.
#include "fd_common.h"
#include "fd_pair_dw_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int GD_THREADS = 256;
constexpr int GD_PANELS = 8;                        // 4 of A (128 columns of dY) + 4 of B (128 columns of X)
constexpr int GD_PLANE = GD_PANELS * DW_PSTRIDE;
constexpr int GD_STAGE = 3 * GD_PLANE;              // 26,112 B
constexpr int GD_RING = 2;
#ifndef GD_NSET
#define GD_NSET 4          // register sets of global loads: stage k is requested GD_NSET stages before it is split into the ring
#endif
static_assert(GD_NSET == 2 || GD_NSET == 4, "register sets");
static_assert(2 * GD_RING * GD_STAGE <= 160 * 1024, "LDS: two blocks per CU");

struct GdUnit {
  const float* A;
  const float* B;
  float* C;
  float* colsum;     // non-null only for the n-tile 0 of an item with a bias gradient
  long lda, ldb, ldc;
  int ma, nb;        // valid A / B columns of this tile (1..128, multiples of 4)
  long row0, row1;
};

__device__ __forceinline__ void gd_unit(const GdUnit& u, char* lds, const int dbg) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // staging: thread -> rows kk, kk + 8 of the 16-row stage, float4 column c4 of A and of B
  const int kk = tid >> 5, c4 = tid & 31;
  const int wA = (c4 >> 3) * DW_PSTRIDE + kk * 64 + (c4 & 7) * 8;
  const int wB = (4 + (c4 >> 3)) * DW_PSTRIDE + kk * 64 + (c4 & 7) * 8;
  const bool aok = 4 * c4 < u.ma, bok = 4 * c4 < u.nb;
  const int ca = aok ? 4 * c4 : 0, cb = bok ? 4 * c4 : 0;
  const float* __restrict__ A = dw_global(u.A) + u.row0 * u.lda;
  const float* __restrict__ B = dw_global(u.B) + u.row0 * u.ldb;
  const long nrows = u.row1 - u.row0, last = nrows - 1;
  const int nst = (int)((nrows + DW_KS - 1) / DW_KS);

  float4 rg[GD_NSET][4];     // [register set][A row kk, A row kk+8, B row kk, B row kk+8]
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  auto sel = [](const float4 v, bool ok) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); };
  // rows past the end of the range are read from its last row (a valid address) and replaced by zeros
  auto load = [&](float4 (&r)[4], int st) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long k = (long)st * DW_KS + kk + 8 * h;
      const bool ok = k <= last;
      const long kc = ok ? k : last;
      r[h] = sel(*reinterpret_cast<const float4*>(A + kc * u.lda + ca), ok && aok);
      r[2 + h] = sel(*reinterpret_cast<const float4*>(B + kc * u.ldb + cb), ok && bok);
    }
  };
  auto put = [&](float4 (&r)[4], char* dst) __attribute__((always_inline)) {
    csum[0] += r[0].x + r[1].x; csum[1] += r[0].y + r[1].y; csum[2] += r[0].z + r[1].z; csum[3] += r[0].w + r[1].w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 s0, s1, s2;
      dw_split4(r[i], s0, s1, s2);
      const int o = ((i & 2) ? wB : wA) + (i & 1) * 8 * 64;
      *reinterpret_cast<uint2*>(dst + o) = s0;
      *reinterpret_cast<uint2*>(dst + GD_PLANE + o) = s1;
      *reinterpret_cast<uint2*>(dst + 2 * GD_PLANE + o) = s2;
    }
  };

  // MFMA side: wave (wm, wn) owns A panels 2wm, 2wm+1 and B panels 2wn, 2wn+1
  const int i16 = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5;
  const int lofs = (8 * kg + (i16 >> 2)) * 64 + half * 32 + (i16 & 3) * 8;
  const int a_rd = (2 * wm) * DW_PSTRIDE + lofs, b_rd = (4 + 2 * wn) * DW_PSTRIDE + lofs;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&](const char* st) __attribute__((always_inline)) {
    uint4 fb[2][3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j][s] = dw_read8(st + b_rd + j * DW_PSTRIDE + s * GD_PLANE);
#pragma unroll
    for (int sa = 2; sa >= 0; --sa) {   // the small terms first
      uint4 fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = dw_read8(st + a_rd + i * DW_PSTRIDE + sa * GD_PLANE);
#pragma unroll
      for (int sb = 2 - sa; sb >= 0; --sb)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = fd::mfma_32x32x16_bf16(fa[i], fb[j][sb], acc[i][j]);
    }
  };

  // full stages through running pointers (no clamp / select of the row index)
  const float* pa = A + ((long)(GD_NSET + 1) * DW_KS + kk) * u.lda + ca;
  const float* pb = B + ((long)(GD_NSET + 1) * DW_KS + kk) * u.ldb + cb;
  const long sa = (long)DW_KS * u.lda, sb = (long)DW_KS * u.ldb, ha = 8 * u.lda, hb = 8 * u.ldb;
  auto load_fast = [&](float4 (&r)[4]) __attribute__((always_inline)) {
    r[0] = sel(*reinterpret_cast<const float4*>(pa), aok);
    r[1] = sel(*reinterpret_cast<const float4*>(pa + ha), aok);
    r[2] = sel(*reinterpret_cast<const float4*>(pb), bok);
    r[3] = sel(*reinterpret_cast<const float4*>(pb + hb), bok);
    pa += sa;
    pb += sb;
  };

  // pipeline: stage s lives in register set s % GD_NSET (requested GD_NSET stages before its split) and ring slot s & 1
#pragma unroll
  for (int k = 0; k < GD_NSET; ++k) load(rg[k], k);
  put(rg[0], lds);
  load(rg[0], GD_NSET);
  __syncthreads();
  // dbg (FD_GROUP_DW_DEBUG, measurements only): 1 = no flush, 2 = no MFMA phase, 4 = no split / LDS writes
  auto step_fast = [&](int s, float4 (&r)[4]) __attribute__((always_inline)) {
    if (!(dbg & 2)) mma(lds + (s & 1) * GD_STAGE);
    if (!(dbg & 4)) put(r, lds + ((s + 1) & 1) * GD_STAGE);
    load_fast(r);
    __syncthreads();
  };
  auto step = [&](int s, float4 (&r)[4]) __attribute__((always_inline)) {
    if (!(dbg & 2)) mma(lds + (s & 1) * GD_STAGE);
    if (!(dbg & 4)) put(r, lds + ((s + 1) & 1) * GD_STAGE);
    load(r, s + GD_NSET + 1);
    __syncthreads();
  };
  const int nfull = (int)(nrows / DW_KS);
  int s = 0;
  // (step s splits stage s + 1 out of set (s + 1) % GD_NSET and requests stage s + GD_NSET + 1 into it)
  for (; s + 2 * GD_NSET + 1 <= nfull; s += GD_NSET) {
#pragma unroll
    for (int k = 0; k < GD_NSET; ++k) step_fast(s + k, rg[(k + 1) % GD_NSET]);
  }
  for (; s < nst; s += GD_NSET) {
#pragma unroll
    for (int k = 0; k < GD_NSET; ++k) step(s + k, rg[(k + 1) % GD_NSET]);      // (stages past nst: zeros)
  }

  // flush: C += acc, atomically (every row range of the tile adds its part); lanes run along n (one line per 32 lanes)
  if (dbg & 1) {
    // keep the results alive without the atomics
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1.2345e-30f) dw_global(u.C)[0] = t + rg[0][0].x + rg[1][1].y + csum[0];
    return;
  }
  const int h = lane >> 5, l31 = lane & 31;
  float* __restrict__ C = dw_global(u.C);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = (2 * wn + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (2 * wm + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < u.ma && n < u.nb) atomicAdd(C + (long)m * u.ldc + n, acc[i][j][r]);
      }
    }
  if (u.colsum != nullptr && aok) {
    // the put() calls of the pipeline staged every row of the range exactly once (zeros beyond it); the eight threads
    // (kk = 0..7) that share the float4 column c4 add their parts atomically
    float* __restrict__ cs = dw_global(u.colsum);
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(cs + 4 * c4 + e, csum[e]);
  }
}

__global__ __launch_bounds__(GD_THREADS, 2) void group_dw_kernel(FdGroupDwDesc d, int ntiles, int dbg) {
  __shared__ __attribute__((aligned(16))) char lds[GD_RING * GD_STAGE];
  // The work is the (tile, 16-row stage) sequence, tile-major; block b takes the b-th of gridDim.x equal pieces of it and
  // walks it tile by tile: every block multiplies the same number of stages (+-1) whatever the mix of item sizes, and
  // flushes at most (piece length / stages per tile) + 2 partial tiles.
  const long nst = (d.rows + DW_KS - 1) / DW_KS;
  const long total = (long)ntiles * nst;
  // logical piece order XCD-major (block b runs on XCD b % 8): neighbouring pieces -- the n-tiles of one dY panel, the
  // m-tiles that share an X panel -- run on ONE XCD at about the same time, so a panel is fetched from HBM once and
  // served to its other tiles by that XCD's L2 (measured: the launch moves 1.1 GB through the CUs for ~0.2 GB of operands)
  const long piece = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  long lo = total * piece / gridDim.x;
  const long hi = total * (piece + 1) / gridDim.x;
  while (lo < hi) {
    const int tile = (int)(lo / nst);
    const long s0 = lo - (long)tile * nst;
    const long s1 = (hi - lo < nst - s0) ? s0 + (hi - lo) : nst;
    lo += s1 - s0;
    // tile -> item (a select chain over the by-value descriptor array: a dynamic index would spill it to scratch)
    FdGroupDwItem it = d.item[0];
    int first = 0, acc_tiles = 0;
#pragma unroll
    for (int t = 0; t < FD_GROUP_DW_MAX_ITEMS; ++t) {
      if (t < d.nitems) {
        const int nt = ((d.item[t].n_out + 127) >> 7) * ((d.item[t].k_in + 127) >> 7);
        if (tile >= acc_tiles) { it = d.item[t]; first = acc_tiles; }
        acc_tiles += nt;
      }
    }
    const int tn = (it.k_in + 127) >> 7;
    const int loc = tile - first, ti = loc / tn, tj = loc - ti * tn;
    GdUnit u;
    u.A = it.A + 128 * ti;
    u.B = it.B + 128 * tj;
    u.C = it.C + (long)(128 * ti) * it.ldc + 128 * tj;
    u.colsum = (it.a_colsum != nullptr && tj == 0) ? it.a_colsum + 128 * ti : nullptr;
    u.lda = it.lda; u.ldb = it.ldb; u.ldc = it.ldc;
    u.ma = it.n_out - 128 * ti < 128 ? it.n_out - 128 * ti : 128;
    u.nb = it.k_in - 128 * tj < 128 ? it.k_in - 128 * tj : 128;
    u.row0 = s0 * DW_KS;
    u.row1 = (s1 * DW_KS < d.rows) ? s1 * DW_KS : d.rows;
    gd_unit(u, lds, dbg);
    __syncthreads();             // the ring is reused by the next piece
  }
}

}  // namespace

extern "C" int fd_group_dw(const FdGroupDwDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FD_CHECK_ARG(desc != nullptr, "fd_group_dw: null descriptor");
  const FdGroupDwDesc& d = *desc;
  FD_CHECK_ARG(d.nitems >= 1 && d.nitems <= FD_GROUP_DW_MAX_ITEMS, "fd_group_dw: 1..%d items", FD_GROUP_DW_MAX_ITEMS);
  FD_CHECK_ARG(d.rows >= 0, "fd_group_dw: negative row count");
  if (d.rows == 0) return FD_OK;
  long tiles = 0;
  for (int t = 0; t < d.nitems; ++t) {
    const FdGroupDwItem& it = d.item[t];
    FD_CHECK_ARG(it.A && it.B && it.C, "fd_group_dw: item %d: null operand", t);
    FD_CHECK_ARG(it.n_out >= 4 && it.k_in >= 4 && (it.n_out & 3) == 0 && (it.k_in & 3) == 0,
                 "fd_group_dw: item %d: n_out / k_in must be positive multiples of 4 (got %d, %d)", t, it.n_out, it.k_in);
    FD_CHECK_ARG(fd_aligned16(it.A) && fd_aligned16(it.B) && (it.lda & 3) == 0 && (it.ldb & 3) == 0 && it.lda >= it.n_out &&
                     it.ldb >= it.k_in,
                 "fd_group_dw: item %d: A [rows,n_out] / B [rows,k_in] must be 16-byte aligned with row strides %% 4 == 0", t);
    FD_CHECK_ARG(it.ldc >= it.k_in, "fd_group_dw: item %d: ldc too small", t);
    tiles += (long)((it.n_out + 127) / 128) * ((it.k_in + 127) / 128);
  }
  int blocks = d.blocks > 0 ? d.blocks : 512;          // two blocks per CU on the 256 CUs of an MI355X
  // at least 8 stages per block: below that the prologue and the flush of a piece cost more than its stages
  const long nst = (d.rows + DW_KS - 1) / DW_KS;
  const long total = tiles * nst;
  FD_CHECK_ARG(tiles < (1L << 24), "fd_group_dw: too many tiles");
  if ((long)blocks > (total + 7) / 8) blocks = (int)((total + 7) / 8);
  if (blocks < 1) blocks = 1;
  static const int dbg = getenv("FD_GROUP_DW_DEBUG") ? atoi(getenv("FD_GROUP_DW_DEBUG")) : 0;
  hipLaunchKernelGGL(group_dw_kernel, dim3(blocks), dim3(GD_THREADS), 0, stream, d, (int)tiles, dbg);
  FD_CHECK_LAUNCH("fd_group_dw");
  return FD_OK;
}
