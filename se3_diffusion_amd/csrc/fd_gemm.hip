// fd_gemm: exact-fp32 MFMA GEMM with fused epilogues -- the dense workhorse of
// the FrameDiff hot path on MI355X (gfx950).
//
// Replaces every torch.nn.Linear / torch.matmul on the reference path
// (model/ipa_pytorch.py:101-166 Linear, :169-233 transitions, :334-374 IPA
// projections, :380-386 / :424-426 qk^T and a*v, score_network.py:67-86 embedder
// MLPs) and their autograd (dX = dY W, dW = dY^T X).
//
//   C[m,n] = epi( alpha * sum_k A(m,k) * B(k,n) )
//   A(m,k) = A[m*a_rs + k*a_cs],  B(k,n) = B[k*b_rs + n*b_cs]   (any strides)
//
// Design (CDNA4): 256 threads = 4 waves; v_mfma_f32_32x32x2_f32 (bit-exact f32
// fmaf chain, 157 TF/s peak).  BK = 32.  K-contiguous operands are staged
// row-major ([rows][32+4]) and read back with ds_read_b128 using a permuted
// k order (MFMA step (q,t) of lane-half h consumes k = 8q+4h+t), so one
// 16-byte LDS read feeds four MFMAs; strided operands are staged [k][rows+4]
// and read with ds_read_b32 in the same k order.  LDS is double-buffered (one
// barrier per k-tile); the next tile travels global -> registers -> LDS under
// the MFMAs of the current one.  The FAST instantiation (16-byte-aligned
// operands) keeps per-thread pointers and row predicates in registers so the
// staging code between MFMA groups is a handful of instructions; the generic
// instantiation handles any stride / alignment / tail element-wise.
// 1-D grid with an XCD-aware remap so that the column blocks that share an A
// row panel run on one XCD and hit its L2.
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int BK = 32;
constexpr int KPAD = 4;

struct GemmArgs {
  FdGemmDesc d;
  int nblk_m, nblk_n;
  int ksplit;
  int mtiles;   // consecutive M tiles pipelined by one block
  int epi_vec;  // 16-byte epilogue through LDS (needs mtiles == 1 and 4-element aligned C / gate / resid / pair)
  int raster;   // tiles 12-14: 1 = consecutive blocks walk the M tiles of one column block (see launch_w)
  int zbatch;   // > 0: the batch index rides in blockIdx.x (grid.x = tiles x zbatch), XCD-swizzled over the WHOLE grid so that
                // the tiles of one batch element -- which re-read its A rows / B columns -- share an XCD's L2
};

// Stages a ROWS x BK operand tile: global -> registers (load) -> LDS (store).
//   KC : global k-contiguous  -> LDS [ROWS][BK+KPAD]
//   !KC: global row-contiguous -> LDS [BK][ROWS+KPAD]
// `rs` = stride of the row index, `cs` = stride of k.
template <int ROWS, bool KC, bool FAST>
struct Stager {
  static constexpr int NV = ROWS / 32;  // float4 per thread (256 threads)
  const float* p[NV];   // FAST: per-thread source pointer of the current tile
  bool ok[NV];          // FAST: row (KC) / row-quad (!KC) in range
  int koff[NV];         // FAST: k offset of this thread's element inside the tile
  long kstep;           // FAST: BK * cs
  // generic path state
  const float* base;
  long rs, cs;
  int row0, nrows;

  __device__ __forceinline__ void init(const float* __restrict__ b, long rs_, long cs_, int row0_, int nrows_,
                                       int k0, int tid) {
    base = b; rs = rs_; cs = cs_; row0 = row0_; nrows = nrows_;
    if (FAST) {
      kstep = (long)BK * cs_;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int f = tid + 256 * i;
        if (KC) {
          const int row = f >> 3, kq = f & 7;
          koff[i] = 4 * kq;
          ok[i] = (row0_ + row) < nrows_;
          p[i] = b + (long)(row0_ + row) * rs_ + (long)(k0 + 4 * kq);
        } else {
          constexpr int RQ = ROWS / 4;
          const int k = f / RQ, rq = f % RQ;
          koff[i] = k;
          ok[i] = (row0_ + 4 * rq) < nrows_;
          p[i] = b + (long)(row0_ + 4 * rq) + (long)(k0 + k) * cs_;
        }
      }
    }
  }

  __device__ __forceinline__ void load(float4 (&r)[NV], int k0, int K, int tid) {
    if (FAST) {
      if (k0 + BK <= K) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok[i]) v = *reinterpret_cast<const float4*>(p[i]);
          r[i] = v;
          p[i] += kstep;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok[i] && k0 + koff[i] < K) v = *reinterpret_cast<const float4*>(p[i]);
          r[i] = v;
          p[i] += kstep;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        const int row = f >> 3, kq = f & 7;
        const int gr = row0 + row, gk = k0 + 4 * kq;
        if (gr < nrows) {
          const float* q = base + (long)gr * rs + (long)gk * cs;
          if (gk + 0 < K) v.x = q[0];
          if (gk + 1 < K) v.y = q[cs];
          if (gk + 2 < K) v.z = q[2 * cs];
          if (gk + 3 < K) v.w = q[3 * cs];
        }
      } else {
        constexpr int RQ = ROWS / 4;
        const int k = f / RQ, rq = f % RQ;
        const int gr = row0 + 4 * rq, gk = k0 + k;
        if (gk < K) {
          const float* q = base + (long)gr * rs + (long)gk * cs;
          if (gr + 0 < nrows) v.x = q[0];
          if (gr + 1 < nrows) v.y = q[rs];
          if (gr + 2 < nrows) v.z = q[2 * rs];
          if (gr + 3 < nrows) v.w = q[3 * rs];
        }
      }
      r[i] = v;
    }
  }

  __device__ __forceinline__ void store(const float4 (&r)[NV], float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + 256 * i;
      if (KC) {
        const int row = f >> 3, kq = f & 7;
        *reinterpret_cast<float4*>(&lds[row * (BK + KPAD) + 4 * kq]) = r[i];
      } else {
        constexpr int RQ = ROWS / 4;
        const int k = f / RQ, rq = f % RQ;
        *reinterpret_cast<float4*>(&lds[k * (ROWS + KPAD) + 4 * rq]) = r[i];
      }
    }
  }
};

template <int ROWS, bool KC>
__device__ __forceinline__ void read_frag(const float* __restrict__ lds, int row, int q, int h,
                                          float (&out)[4]) {
  if (KC) {
    float4 v = *reinterpret_cast<const float4*>(&lds[row * (BK + KPAD) + 8 * q + 4 * h]);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = lds[(8 * q + 4 * h + t) * (ROWS + KPAD) + row];
  }
}

// epilogue of one BM x BN output tile held in the MFMA accumulators
template <int TM, int TN>
__device__ __forceinline__ void store_tile(const FdGemmDesc& d, float* __restrict__ C, f32x16 (&acc)[TM][TN],
                                           int m_base, int n_base, int h, int l31, bool splitk) {
  if (splitk) {
    // split-K partial: C += alpha * partial (atomic; C holds the running sum)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= d.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n_base + j * 32 + l31;
          if (n < d.N) atomicAdd(C + (long)m * d.ldc + n, d.alpha * acc[i][j][r]);
        }
      }
    return;
  }
  // Branch-free operand fetch: row / column indices are CLAMPED into range for every epilogue load (gate, pair,
  // residual, old C) and only the final store is predicated, so the compiler can hoist and batch all loads of the
  // unrolled (row, column) loop instead of one exec-masked load-use pair at a time.  The pair (b,i,j) decode of a
  // row is one division per 32-row block, then carries.
  bool nok[TN];
  float bj[TN];
  int ncol[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n_base + j * 32 + l31;
    nok[j] = n < d.N;
    ncol[j] = nok[j] ? n : d.N - 1;
    bj[j] = d.bias ? d.bias[ncol[j]] : 0.f;
  }
  const int nres = d.nres;
  const bool has_pair = d.pair_p != nullptr, has_gate = d.gate != nullptr, has_res = d.resid != nullptr;
  const bool has_rs = d.rowscale != nullptr, has_beta = d.beta != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mi = m_base + i * 32;
    int rr0 = 0, ii0 = 0, bb0 = 0;
    if (has_pair) {
      const int q0 = mi / nres;
      rr0 = mi - q0 * nres;
      bb0 = q0 / nres;
      ii0 = q0 - bb0 * nres;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
      const bool mok = mi + o < d.M;
      const int m = mok ? mi + o : d.M - 1;
      const float* pp = C;
      const float* pq = C;
      if (has_pair) {
        int rr = rr0 + o, ii = ii0, bb = bb0;
        while (rr >= nres) {
          rr -= nres;
          if (++ii == nres) { ii = 0; ++bb; }
        }
        if (!mok) { rr = 0; ii = 0; bb = 0; }
        pp = d.pair_p + ((long)bb * nres + ii) * d.ld_pair;   // row m / nres
        pq = d.pair_q + ((long)bb * nres + rr) * d.ld_pair;   // row (m / nres^2) * nres + m % nres
      }
      const float rs = has_rs ? d.rowscale[m] : 1.f;
      float* crow = C + (long)m * d.ldc;
      const float* grow = has_gate ? d.gate + (long)m * d.ld_gate : C;
      const float* rrow = has_res ? d.resid + (long)m * d.ld_resid : C;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = ncol[j];
        float v = d.alpha * acc[i][j][r] + bj[j];
        if (has_pair) v += pp[n] + pq[n];
        if (d.relu) v = v > 0.f ? v : 0.f;
        if (has_gate) v = grow[n] > 0.f ? v : 0.f;
        v *= rs;
        if (has_res) v += rrow[n];
        if (has_beta) v += crow[n];
        if (mok && nok[j]) crow[n] = v;
      }
    }
  }
}

// Vector epilogue: the accumulator tile is transposed through LDS so that every thread owns 4 consecutive columns
// of a row -- C, gate, residual and pair operands then move as 16-byte accesses (4x fewer memory instructions than
// the fragment-shaped scalar epilogue, which is what bounds the short-K GEMMs of the backward pass).
template <int BM, int BN, int TM, int TN, int NTHR = 256>
__device__ __forceinline__ void store_tile_vec(const FdGemmDesc& d, float* __restrict__ C, f32x16 (&acc)[TM][TN],
                                               float* __restrict__ lds, int m0, int n0, int wm, int wn, int h, int l31,
                                               int tid) {
  constexpr int PITCH = BN + 4;
  __syncthreads();   // every wave is done reading operand fragments from LDS
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        lds[row * PITCH + (wn * TN + j) * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();
  constexpr int C4 = BN / 4;          // float4 columns per row
  constexpr int RPP = NTHR / C4;      // rows per pass
  const int c4 = tid % C4, r0 = tid / C4;
  const int n = n0 + 4 * c4;
  const bool nok = n < d.N;            // N % 4 == 0: the whole float4 is in or out
  const int nc = nok ? n : 0;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d.bias) b4 = *reinterpret_cast<const float4*>(d.bias + nc);
  const bool has_pair = d.pair_p != nullptr, has_gate = d.gate != nullptr, has_res = d.resid != nullptr;
  const bool has_rs = d.rowscale != nullptr, has_beta = d.beta != 0;
  const int nres = d.nres;
#pragma unroll 4
  for (int p = 0; p < BM / RPP; ++p) {
    const int row = p * RPP + r0;
    const bool mok = m0 + row < d.M;
    const int m = mok ? m0 + row : d.M - 1;
    const float4 a4 = *reinterpret_cast<const float4*>(&lds[row * PITCH + 4 * c4]);
    float4 v = make_float4(d.alpha * a4.x + b4.x, d.alpha * a4.y + b4.y, d.alpha * a4.z + b4.z, d.alpha * a4.w + b4.w);
    if (has_pair) {
      const int q = m / nres;
      const int jj = m - q * nres;
      const int bb = q / nres;
      const float4 pp = *reinterpret_cast<const float4*>(d.pair_p + (long)q * d.ld_pair + nc);
      const float4 pq = *reinterpret_cast<const float4*>(d.pair_q + ((long)bb * nres + jj) * d.ld_pair + nc);
      v.x += pp.x + pq.x; v.y += pp.y + pq.y; v.z += pp.z + pq.z; v.w += pp.w + pq.w;
    }
    if (d.relu) {
      v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    }
    if (has_gate) {
      const float4 g4 = *reinterpret_cast<const float4*>(d.gate + (long)m * d.ld_gate + nc);
      v.x = g4.x > 0.f ? v.x : 0.f; v.y = g4.y > 0.f ? v.y : 0.f; v.z = g4.z > 0.f ? v.z : 0.f; v.w = g4.w > 0.f ? v.w : 0.f;
    }
    if (has_rs) {
      const float rs = d.rowscale[m];
      v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
    }
    if (has_res) {
      const float4 r4 = *reinterpret_cast<const float4*>(d.resid + (long)m * d.ld_resid + nc);
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    float* cp = C + (long)m * d.ldc + nc;
    if (has_beta) {
      const float4 c4v = *reinterpret_cast<const float4*>(cp);
      v.x += c4v.x; v.y += c4v.y; v.z += c4v.z; v.w += c4v.w;
    }
    if (mok && nok) *reinterpret_cast<float4*>(cp) = v;
  }
}

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, bool FAST, bool ROWSUM = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int A_LDS = A_KC ? BM * (BK + KPAD) : BK * (BM + KPAD);
  constexpr int B_LDS = B_KC ? BN * (BK + KPAD) : BK * (BN + KPAD);
  constexpr int LDS_STAGE = A_LDS + B_LDS;
  __shared__ __attribute__((aligned(16))) float lds[2 * LDS_STAGE];
  float* As = lds;
  float* Bs = lds + A_LDS;

  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;

  // A block owns `mtiles` consecutive M tiles of one N column block and runs them as ONE software pipeline
  // (the first loads of tile i+1 are in flight while tile i finishes and stores), so short-K GEMMs do not pay
  // a load-latency bubble per output tile.
  const int T = g.mtiles;
  const int nblk_mg = (g.nblk_m + T - 1) / T;
  const int nblk = nblk_mg * g.nblk_n;
  int lid, z;
  if (g.zbatch > 0) {
    const int L = fd_xcd_swizzle((int)blockIdx.x, nblk * g.zbatch);
    z = L / nblk;
    lid = L % nblk;
  } else {
    lid = fd_xcd_swizzle((int)blockIdx.x, nblk);
    z = (int)blockIdx.y;
  }
  const int bmg = lid / g.nblk_n, bn = lid % g.nblk_n;
  const int mt0 = bmg * T;
  const int ntile = (g.nblk_m - mt0 < T) ? g.nblk_m - mt0 : T;
  const int n0 = bn * BN;

  const int zo = z / d.bdiv, zi = z % d.bdiv;
  const float* __restrict__ A = d.A + zo * d.a_so + zi * d.a_si;
  const float* __restrict__ B = d.B + zo * d.b_so + zi * d.b_si;
  float* __restrict__ C = d.C + zo * d.c_so + zi * d.c_si;

  // split-K: blockIdx.z owns k-tiles [kt0, nkt)
  const int nkt_all = (d.K + BK - 1) / BK;
  const int per = (nkt_all + g.ksplit - 1) / g.ksplit;
  const int kt0 = (int)blockIdx.z * per;
  const int nkt = (kt0 + per < nkt_all) ? kt0 + per : nkt_all;
  const int nk = nkt - kt0;
  if (nk <= 0 || ntile <= 0) return;
  const int total = ntile * nk;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<BM, A_KC, FAST> sa;
  Stager<BN, B_KC, FAST> sb;
  // two register sets: tile it+1 and tile it+2 are both in flight while tile it is multiplied (prefetch
  // distance ~1.75 k-tiles: cold HBM reads of a streamed A operand need more than one k-tile of cover)
  float4 ra[2][Stager<BM, A_KC, FAST>::NV], rb[2][Stager<BN, B_KC, FAST>::NV];
  int l_t = 0, l_k = 0;   // (tile, k-tile) of the next global load
  auto issue_load = [&](float4 (&xa)[Stager<BM, A_KC, FAST>::NV], float4 (&xb)[Stager<BN, B_KC, FAST>::NV]) {
    if (l_k == 0) {
      sa.init(A, d.a_rs, d.a_cs, (mt0 + l_t) * BM, d.M, kt0 * BK, tid);
      // B(k,n): the "row" index of the staged tile is n -> row stride b_cs, k stride b_rs
      sb.init(B, d.b_cs, d.b_rs, n0, d.N, kt0 * BK, tid);
    }
    sa.load(xa, (kt0 + l_k) * BK, d.K, tid);
    sb.load(xb, (kt0 + l_k) * BK, d.K, tid);
    if (++l_k == nk) { l_k = 0; ++l_t; }
  };

  // Software pipeline, one barrier per k-tile.  While tile `it` is multiplied out of LDS[cur]:
  //   after MFMA group 0: tile it+1 (registers, set (it+1)&1) -> LDS[cur^1]
  //   after MFMA group 1: global loads of tile it+3 -> the register set just freed
  // so LDS writes and global loads hide under the 64-cycle fp32 MFMAs of the SAME wave.  MFMA operand fragments
  // are double-buffered in registers (group q+1 is fetched before group q issues).
  issue_load(ra[0], rb[0]);
  sa.store(ra[0], As, tid);
  sb.store(rb[0], Bs, tid);
  if (total > 1) issue_load(ra[1], rb[1]);
  if (total > 2) issue_load(ra[0], rb[0]);
  __syncthreads();

  float a[2][TM][4], b[2][TN][4];
  int cur = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(As, (wm * TM + i) * 32 + l31, 0, h, a[0][i]);
#pragma unroll
  for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(Bs, (wn * TN + j) * 32 + l31, 0, h, b[0][j]);

  int c_t = 0, c_k = 0;   // (tile, k-tile) being multiplied
  const bool do_rowsum = ROWSUM && (d.a_rowsum != nullptr) && bn == 0 && wn == 0 && z == 0;
  float rsum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rsum[i] = 0.f;
  auto body = [&](int it, float4 (&xa)[Stager<BM, A_KC, FAST>::NV], float4 (&xb)[Stager<BN, B_KC, FAST>::NV]) {
    // xa/xb: the register set holding tile it+1 (stored here, then refilled with tile it+3)
    const float* Ac = As + cur * LDS_STAGE;
    const float* Bc = Bs + cur * LDS_STAGE;
    float* An = As + (cur ^ 1) * LDS_STAGE;
    float* Bn = Bs + (cur ^ 1) * LDS_STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(Ac, (wm * TM + i) * 32 + l31, q + 1, h, a[(q + 1) & 1][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(Bc, (wn * TN + j) * 32 + l31, q + 1, h, b[(q + 1) & 1][j]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = fd::mfma_32x32x2(a[q & 1][i][t], b[q & 1][j][t], acc[i][j]);
      if (q == 0 && it + 1 < total) {
        sa.store(xa, An, tid);
        sb.store(xb, Bn, tid);
      }
      if (q == 1 && it + 3 < total) issue_load(xa, xb);
      if (ROWSUM && do_rowsum) {
        // fused bias gradient: row sums of A over k (A = dY^T in dW = dY^T X), from the fragments already in registers
#pragma unroll
        for (int i = 0; i < TM; ++i) rsum[i] += (a[q & 1][i][0] + a[q & 1][i][1]) + (a[q & 1][i][2] + a[q & 1][i][3]);
      }
    }
    if (++c_k == nk) {
      if (g.epi_vec)
        store_tile_vec<BM, BN, TM, TN>(d, C, acc, lds, (mt0 + c_t) * BM, n0, wm, wn, h, l31, tid);
      else
        store_tile<TM, TN>(d, C, acc, (mt0 + c_t) * BM + wm * TM * 32, n0 + wn * TN * 32, h, l31, g.ksplit > 1);
      if (ROWSUM && do_rowsum) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float tot = rsum[i] + __shfl_xor(rsum[i], 32);   // the two lane halves hold disjoint k subsets
          const int m = (mt0 + c_t) * BM + (wm * TM + i) * 32 + l31;
          if (h == 0 && m < d.M) atomicAdd(d.a_rowsum + m, d.alpha * tot);
          rsum[i] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      c_k = 0;
      ++c_t;
    }
    __syncthreads();
    cur ^= 1;
    if (it + 1 < total) {
#pragma unroll
      for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(An, (wm * TM + i) * 32 + l31, 0, h, a[0][i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(Bn, (wn * TN + j) * 32 + l31, 0, h, b[0][j]);
    }
  };
  for (int it = 0; it < total; it += 2) {
    body(it, ra[1], rb[1]);
    if (it + 1 < total) body(it + 1, ra[0], rb[0]);
  }
}

#include "fd_gemm_split.h"   // gemm_bx3_kernel: tile code 4
#include "fd_gemm_direct.h"  // gemm_direct_kernel: tile code 5

// 16-byte staging is legal when the contiguous index is a multiple of 4 everywhere the kernel can touch it
bool operands_vectorisable(const FdGemmDesc& d) {
  auto al4 = [](long x) { return (x & 3) == 0; };
  bool va, vb;
  if (d.a_cs == 1)
    va = fd_aligned16(d.A) && al4(d.a_rs) && al4(d.K) && al4(d.a_so) && al4(d.a_si);
  else
    va = (d.a_rs == 1) && fd_aligned16(d.A) && al4(d.a_cs) && al4(d.M) && al4(d.a_so) && al4(d.a_si);
  if (d.b_rs == 1)
    vb = fd_aligned16(d.B) && al4(d.b_cs) && al4(d.K) && al4(d.b_so) && al4(d.b_si);
  else
    vb = (d.b_cs == 1) && fd_aligned16(d.B) && al4(d.b_rs) && al4(d.N) && al4(d.b_so) && al4(d.b_si);
  return va && vb;
}

// the LDS-transposed float4 epilogue needs 4-element aligned C / bias / gate / residual / pair operands
bool epilogue_vectorisable(const FdGemmDesc& d, int ksplit) {
  auto al4 = [](long x) { return (x & 3) == 0; };
  bool ok = ksplit == 1 && al4(d.N) && al4(d.ldc) && fd_aligned16(d.C) && al4(d.c_so) && al4(d.c_si);
  if (d.bias) ok = ok && fd_aligned16(d.bias);
  if (d.gate) ok = ok && fd_aligned16(d.gate) && al4(d.ld_gate);
  if (d.resid) ok = ok && fd_aligned16(d.resid) && al4(d.ld_resid);
  if (d.pair_p) ok = ok && fd_aligned16(d.pair_p) && fd_aligned16(d.pair_q) && al4(d.ld_pair);
  return ok;
}

#include "fd_gemm_s64.h"     // gemm_s64_kernel: tile code 10
#include "fd_gemm_w.h"       // gemm_w_kernel: tile codes 12 / 13 / 14 (pre-split weights)

// split-bf16 kernel: instantiated for A k-contiguous (activations [rows, features]) with either B layout
// (y = x W^T and dx = dy W) and for both operands row-contiguous (dW = dY^T X)
bool bx3_layout_ok(const FdGemmDesc& d) {
  const bool a_kc = (d.a_cs == 1), b_kc = (d.b_rs == 1);
  return a_kc || (!a_kc && !b_kc);
}

// persistent blocks of the split kernel: one per CU (0 = fresh block per tile; FD_GEMM_NOPERSIST=1,
// FD_GEMM_PERSIST_BLOCKS=n or fd_gemm_set_persistent_blocks)
int g_persist_blocks = -1;
int persist_blocks() {
  if (g_persist_blocks < 0) {
    const char* e = getenv("FD_GEMM_PERSIST_BLOCKS");
    g_persist_blocks = getenv("FD_GEMM_NOPERSIST") ? 0 : (e && atoi(e) >= 0 ? atoi(e) : 256);   // MI355X: 256 CUs
  }
  return g_persist_blocks;
}

template <int BM>
int launch_bx3(const FdGemmDesc& d, hipStream_t stream) {
  GemmArgs g{};
  g.d = d;
  g.nblk_m = fd_cdiv(d.M, BM);
  g.nblk_n = fd_cdiv(d.N, XBN);
  const int nb = d.batch > 0 ? d.batch : 1;
  g.ksplit = d.ksplit > 1 ? d.ksplit : 1;
  const int nkt_all = fd_cdiv(d.K, XBK);
  if (g.ksplit > nkt_all) g.ksplit = nkt_all > 0 ? nkt_all : 1;
  g.mtiles = 1;
  g.epi_vec = epilogue_vectorisable(d, g.ksplit);
  const bool a_kc = (d.a_cs == 1), b_kc = (d.b_rs == 1);
  dim3 grid(g.nblk_m * g.nblk_n, nb, g.ksplit), block(SplitCfg<BM>::NTHR, 1, 1);
  // the 256-row shape without split-K accumulates transposed and stores float4s straight from registers
  static const bool no_trans = getenv("FD_GEMM_NOTRANS") != nullptr;   // (A/B measurements)
  const bool trans = BM == 256 && g.ksplit == 1 && a_kc && !no_trans;
  // >= 2 tiles per CU: one persistent block per CU walks them with the next tile's prologue under the epilogue
  const int kCUs = persist_blocks();
  if (trans && kCUs > 0 && g.nblk_m * g.nblk_n >= 2 * kCUs) {
    dim3 pgrid(kCUs, nb, 1);
    if (b_kc)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_bx3p_kernel<true>), pgrid, block, 0, stream, g);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_bx3p_kernel<false>), pgrid, block, 0, stream, g);
    FD_CHECK_LAUNCH("fd_gemm(split-bf16, persistent)");
    return FD_OK;
  }
  if (a_kc && b_kc) {
    if (trans)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_bx3_kernel<BM, true, true, BM == 256>), grid, block, 0, stream, g);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_bx3_kernel<BM, true, true, false>), grid, block, 0, stream, g);
  } else if (a_kc && !b_kc) {
    if (trans)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_bx3_kernel<BM, true, false, BM == 256>), grid, block, 0, stream, g);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_bx3_kernel<BM, true, false, false>), grid, block, 0, stream, g);
  } else {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_bx3_kernel<BM, false, false, false>), grid, block, 0, stream, g);
  }
  FD_CHECK_LAUNCH("fd_gemm(split-bf16)");
  return FD_OK;
}

// FD_GEMM_EXACT_F32=1 (or fd_gemm_set_exact_f32) keeps every GEMM on the fp32-MFMA (fmaf-chain) kernels
int g_split_mode = -1;   // -1: not read yet, 0: exact fp32 only, 1: split-bf16 allowed
bool split_enabled() {
  if (g_split_mode < 0) {
    const char* e = getenv("FD_GEMM_EXACT_F32");
    g_split_mode = (e && e[0] && e[0] != '0') ? 0 : 1;
  }
  return g_split_mode != 0;
}

// FD_GEMM_NO_S64=1: the automatic choice keeps the 64x64 fp32 tile (A/B measurements)
bool s64_enabled() {
  static const bool on = getenv("FD_GEMM_NO_S64") == nullptr;
  return on;
}

// FD_GEMM_NO_W=1: the automatic choice ignores pre-split weight planes (A/B measurements)
bool w_enabled() {
  static const bool on = getenv("FD_GEMM_NO_W") == nullptr;
  return on;
}

// tile selection: 1 = 128x128, 2 = 64x64, 3 = 128x32 (fp32 MFMA); 4 = 256x128 split-bf16; 10 = 64x64 split-bf16.
// Wide tiles when the problem fills the chip, narrow otherwise.
int plan_tile(const FdGemmDesc& d, bool fast) {
  int cfg = d.tile;
  if (cfg == 0) {
    const long blocks128 = (long)fd_cdiv(d.M, 128) * fd_cdiv(d.N, 128) * (d.batch > 0 ? d.batch : 1);
    if (d.N <= 48) cfg = 3;
    else if (blocks128 >= 512 && d.N >= 96) cfg = 1;
    else cfg = 2;
    // the split-bf16 kernel wherever the wide fp32 tile would run, and for mid-size problems (single-backbone
    // sampling: 16k..64k pair rows) as soon as its 256x128 tiles cover ~40 % of the CUs
    const long blocks_x3 = (long)fd_cdiv(d.M, XBM) * fd_cdiv(d.N, XBN) * (d.batch > 0 ? d.batch : 1);
    if ((cfg == 1 || (d.N >= 96 && d.M >= 1024 && blocks_x3 >= 96)) && fast && d.K >= 64 && bx3_layout_ok(d) &&
        split_enabled())
      cfg = 4;
    // latency-bound launches (node-level GEMMs of sampling: fewer 64x64 tiles than CUs): 32x32 tiles, K split
    // over the four waves of the block
    const long blocks64 = (long)fd_cdiv(d.M, 64) * fd_cdiv(d.N, 64) * (d.batch > 0 ? d.batch : 1);
    // (measured: 128x320x320 8 vs 11 us, 128x256x2688 25 vs 68 us, 1024x320x320 9 vs 12 us; with >= ~200 tiles of
    // 64x64 the wider tile's operand reuse wins again)
    if (cfg != 4 && blocks64 <= 128 && d.M <= 1024 && direct_ok(d)) cfg = 5;
    // un-batched activations x weights with K >= 256 that would take the 64x64 fp32 tile: the 64x64 split-bf16 kernel
    // (a third of its MFMA cycles; measured 15.0 vs 18.3 us at 3840 x 320 x 320, 41 vs 56 us at K = 1280; the weight
    // gradients (both operands row-contiguous, split-K) and the batched attention products are not faster on it)
    if (cfg == 2 && fast && d.a_cs == 1 && d.batch <= 1 && d.K >= 256 && split_enabled() && s64_enabled()) cfg = 10;
  }
  if (d.tile == 0 && w_ok(d) && fast && split_enabled() && w_enabled() && (cfg == 10 || cfg == 4 || cfg == 2) && d.K >= 128 &&
      d.M <= 65536) {      // (residue rows: the pair-level GEMMs of the unfused paths keep the persistent 256 x 128 kernel)
    // activations x PRE-SPLIT weights (FdGemmDesc.b_planes: the host split the flat parameter buffer once per step).  Which tile,
    // measured at M = 3,840 rows (profiles/r06_node_gemm.log): many output tiles and a short reduction (IPA's projections,
    // N = 6816, K = 256) stay on tile 4, whose 256-row tile moves half the weight bytes per output; k-contiguous weights from
    // ~224 tiles of 128 x 128 up on tile 12, row-contiguous ones (dx = dy W) from ~480 tiles of 64 x 128 up on tile 13; split-K
    // launches (a long reduction over few tiles) on tile 12 when their blocks fill the chip; a long un-split reduction over few
    // tiles stays on tile 10; everything else on the 64 x 64 tile 14
    const int ks = d.ksplit > 1 ? d.ksplit : 1;
    const long t128 = (long)fd_cdiv(d.M, 128) * fd_cdiv(d.N, 128), t64 = (long)fd_cdiv(d.M, 64) * fd_cdiv(d.N, 128);
    const bool b_kc = d.b_rs == 1;
    if (d.N >= 4096 && d.K <= 256 && cfg == 4) {
    } else if (ks > 1) {
      cfg = t128 * ks >= 224 ? 12 : 14;
    } else if (b_kc && t128 >= 224) {
      cfg = 12;
    } else if (!b_kc && t64 >= 480) {
      cfg = 13;
    } else if (d.K > 1024 && t128 < 128 && cfg == 10) {
    } else {
      cfg = 14;
    }
  }
  if (cfg >= 12 && cfg <= 14 && !(w_ok(d) && fast && split_enabled())) cfg = (fast && split_enabled() && d.a_cs == 1 && d.batch <= 1) ? 10 : 2;
  if (cfg == 10 && !(fast && split_enabled())) cfg = 2;   // (explicit requests: exact-fp32 mode, unaligned operands)
  if ((cfg == 4 || cfg == 6) && !split_enabled()) {
    // FD_GEMM_EXACT_F32=1 also overrides explicit requests for the split-bf16 kernel (the host asks for it on the
    // weight gradients)
    const long blocks128 = (long)fd_cdiv(d.M, 128) * fd_cdiv(d.N, 128) * (d.batch > 0 ? d.batch : 1);
    cfg = (blocks128 >= 512 && d.N >= 96 && !d.a_rowsum) ? 1 : 2;
  }
  if (!fast && cfg == 1) cfg = 2;   // the element-wise staging path is only instantiated for the small tiles
  return cfg;
}

template <int BM, int BN, int WGM, int WGN, bool FAST>
int launch_cfg(const FdGemmDesc& d, hipStream_t stream) {
  GemmArgs g{};
  g.d = d;
  g.nblk_m = fd_cdiv(d.M, BM);
  g.nblk_n = fd_cdiv(d.N, BN);
  const bool a_kc = (d.a_cs == 1);
  const bool b_kc = (d.b_rs == 1);
  const int nb = d.batch > 0 ? d.batch : 1;
  g.ksplit = d.ksplit > 1 ? d.ksplit : 1;
  {
    const int nkt_all = fd_cdiv(d.K, BK);
    if (g.ksplit > nkt_all) g.ksplit = nkt_all > 0 ? nkt_all : 1;
  }
  // persistent M-tile pipelining for the large un-batched GEMMs (pair-level tensors): keep >= ~2300 blocks
  g.mtiles = 1;
  if (g.ksplit == 1 && nb == 1) {
    const long blocks = (long)g.nblk_m * g.nblk_n;
    long t = blocks / 2304;
    g.mtiles = (int)(t < 1 ? 1 : (t > 16 ? 16 : t));
    if (d.mtiles > 0) g.mtiles = d.mtiles;
  }
  g.epi_vec = epilogue_vectorisable(d, g.ksplit) && (d.mtiles <= 1);
  if (g.epi_vec) g.mtiles = 1;
  const int nblk_mg = (g.nblk_m + g.mtiles - 1) / g.mtiles;
  // batched launches (attention products: a handful of tiles per batch element): a [tiles, batch] grid puts tile t of EVERY batch
  // element on XCD t % 8 and each element's operands are fetched by several L2s (126 MB per launch for 25 MB of operands)
  g.zbatch = (nb > 1 && (long)nblk_mg * g.nblk_n * nb < (1l << 30) && getenv("FD_GEMM_BATCH_Y") == nullptr) ? nb : 0;
  dim3 grid(nblk_mg * g.nblk_n * (g.zbatch > 0 ? nb : 1), g.zbatch > 0 ? 1 : nb, g.ksplit), block(256, 1, 1);
  if constexpr (BM == 64 && BN == 64 && FAST) {
    if (d.a_rowsum && !a_kc && !b_kc) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, false, false, true, true>), grid, block, 0, stream, g);
      FD_CHECK_LAUNCH("fd_gemm");
      return FD_OK;
    }
  }
  if (a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, true, true, FAST>), grid, block, 0, stream, g);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, true, false, FAST>), grid, block, 0, stream, g);
  else if (!a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, false, true, FAST>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, false, false, FAST>), grid, block, 0, stream, g);
  FD_CHECK_LAUNCH("fd_gemm");
  return FD_OK;
}

}  // namespace

extern "C" int fd_gemm_set_exact_f32(int exact) {
  const int was = split_enabled() ? 0 : 1;
  g_split_mode = exact ? 0 : 1;
  return was;
}

extern "C" int fd_gemm_set_persistent_blocks(int blocks) {
  const int was = persist_blocks();
  g_persist_blocks = blocks > 0 ? blocks : 0;
  return was;
}

extern "C" int fd_split_planes(const float* x, long n, void* planes, void* stream_) {
  FD_CHECK_ARG(x != nullptr && planes != nullptr, "fd_split_planes: null operand");
  FD_CHECK_ARG(n >= 0 && (n % 8) == 0 && fd_aligned16(x) && fd_aligned16(planes), "fd_split_planes: n %% 8 == 0, 16-byte aligned buffers");
  if (n == 0) return FD_OK;
  const long n8 = n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, n8, n,
                     reinterpret_cast<unsigned short*>(planes));
  FD_CHECK_LAUNCH("fd_split_planes");
  return FD_OK;
}

extern "C" int fd_gemm_plan(const FdGemmDesc* desc) {
  FD_CHECK_ARG(desc != nullptr, "fd_gemm_plan: null descriptor");
  return plan_tile(*desc, operands_vectorisable(*desc));
}

extern "C" int fd_gemm(const FdGemmDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FD_CHECK_ARG(desc != nullptr, "fd_gemm: null descriptor");
  FdGemmDesc d = *desc;
  FD_CHECK_ARG(d.A && d.B && d.C, "fd_gemm: null operand");
  FD_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K >= 0, "fd_gemm: negative extent");
  if (d.M == 0 || d.N == 0) return FD_OK;
  if (d.bdiv <= 0) d.bdiv = 1;
  FD_CHECK_ARG((d.pair_p == nullptr) == (d.pair_q == nullptr), "fd_gemm: pair_p/pair_q must come together");
  FD_CHECK_ARG(d.pair_p == nullptr || d.nres > 0, "fd_gemm: pair epilogue needs nres");
  if (d.nres <= 0) d.nres = 1;
  if (d.ksplit > 1) {
    FD_CHECK_ARG(!d.bias && !d.pair_p && !d.resid && !d.gate && !d.rowscale && !d.relu,
                 "fd_gemm: split-K accumulates alpha*A*B into C; no other epilogue allowed");
  }
  const bool fast = operands_vectorisable(d);
  const int cfg = plan_tile(d, fast);
  if (cfg == 4 || cfg == 6)
    FD_CHECK_ARG(fast && bx3_layout_ok(d), "fd_gemm: tiles 4/6 (split-bf16) need 16-byte aligned operands and a k-contiguous A "
                                           "or both operands row-contiguous");
  if (d.a_rowsum && !((cfg == 2 || cfg == 4 || cfg == 6 || cfg == 10) && fast && d.a_cs != 1 && d.b_rs != 1 && d.batch <= 1)) {
    // the fused row-sum lives in the instantiations with both operands row-contiguous (the dW = dY^T X case) of
    // the 64x64 fp32 kernel and the split-bf16 kernel; anything else takes the stand-alone column-sum kernel
    FD_CHECK_ARG(d.a_rs == 1 && d.alpha == 1.0f && (d.batch <= 1), "fd_gemm: a_rowsum needs a row-contiguous A, alpha 1, no batch");
    int rc = fd_colsum_acc(d.A, d.a_cs, d.K, d.M, d.a_rowsum, stream_);
    if (rc != FD_OK) return rc;
    d.a_rowsum = nullptr;
  }
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2, true>(d, stream);
    case 2: return fast ? launch_cfg<64, 64, 2, 2, true>(d, stream) : launch_cfg<64, 64, 2, 2, false>(d, stream);
    case 3: return fast ? launch_cfg<128, 32, 4, 1, true>(d, stream) : launch_cfg<128, 32, 4, 1, false>(d, stream);
    case 4: return launch_bx3<256>(d, stream);
    case 6: return launch_bx3<128>(d, stream);
    case 10: return launch_s64(d, stream);
    case 12: return launch_w<128, 128>(d, stream);
    case 13: return launch_w<64, 128>(d, stream);
    case 14: return launch_w<64, 64>(d, stream);
    case 5:
      FD_CHECK_ARG(direct_ok(d), "fd_gemm: tile 5 (latency kernel) needs K %% 8 == 0, unit-stride 16-byte aligned operands, "
                                 "no pair epilogue / split-K / row sum");
      return launch_direct(d, stream);
    default: fd_set_error("fd_gemm: bad tile config %d", cfg); return FD_ERR_ARG;
  }
}
