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
// and read with ds_read_b32 in the same k order.  Global->register prefetch of
// tile kt+1 is issued before the MFMAs of tile kt (write-late staging).
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
  int vecA, vecB;
  int ksplit;
};

template <int ROWS, bool KC>
struct Stager {
  // ROWS x BK tile; KC: global k-contiguous -> LDS [ROWS][BK+KPAD]
  //                !KC: global row-contiguous -> LDS [BK][ROWS+KPAD]
  static constexpr int NV = ROWS / 32;  // float4 per thread (256 threads)
  float4 r[NV];

  __device__ __forceinline__ void load(const float* __restrict__ base, long rs, long cs,
                                       int row0, int k0, int nrows, int K, int vec, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        int row = f >> 3, kq = f & 7;
        int gr = row0 + row, gk = k0 + 4 * kq;
        if (gr < nrows) {
          const float* p = base + (long)gr * rs + (long)gk * cs;
          if (vec) {
            if (gk < K) v = *reinterpret_cast<const float4*>(p);
          } else {
            if (gk + 0 < K) v.x = p[0];
            if (gk + 1 < K) v.y = p[cs];
            if (gk + 2 < K) v.z = p[2 * cs];
            if (gk + 3 < K) v.w = p[3 * cs];
          }
        }
      } else {
        constexpr int RQ = ROWS / 4;
        int k = f / RQ, rq = f % RQ;
        int gr = row0 + 4 * rq, gk = k0 + k;
        if (gk < K) {
          const float* p = base + (long)gr * rs + (long)gk * cs;
          if (vec) {
            if (gr < nrows) v = *reinterpret_cast<const float4*>(p);
          } else {
            if (gr + 0 < nrows) v.x = p[0];
            if (gr + 1 < nrows) v.y = p[rs];
            if (gr + 2 < nrows) v.z = p[2 * rs];
            if (gr + 3 < nrows) v.w = p[3 * rs];
          }
        }
      }
      r[i] = v;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = tid + 256 * i;
      if (KC) {
        int row = f >> 3, kq = f & 7;
        *reinterpret_cast<float4*>(&lds[row * (BK + KPAD) + 4 * kq]) = r[i];
      } else {
        constexpr int RQ = ROWS / 4;
        int k = f / RQ, rq = f % RQ;
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

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int A_LDS = A_KC ? BM * (BK + KPAD) : BK * (BM + KPAD);
  constexpr int B_LDS = B_KC ? BN * (BK + KPAD) : BK * (BN + KPAD);
  __shared__ __attribute__((aligned(16))) float lds[A_LDS + B_LDS];
  float* As = lds;
  float* Bs = lds + A_LDS;

  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;

  const int nblk = g.nblk_m * g.nblk_n;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, nblk);
  const int bm = lid / g.nblk_n, bn = lid % g.nblk_n;
  const int m0 = bm * BM, n0 = bn * BN;

  const int z = (int)blockIdx.y;
  const int zo = z / d.bdiv, zi = z % d.bdiv;
  const float* __restrict__ A = d.A + zo * d.a_so + zi * d.a_si;
  const float* __restrict__ B = d.B + zo * d.b_so + zi * d.b_si;
  float* __restrict__ C = d.C + zo * d.c_so + zi * d.c_si;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<BM, A_KC> sa;
  Stager<BN, B_KC> sb;
  // split-K: blockIdx.z owns k-tiles [kt0, nkt)
  const int nkt_all = (d.K + BK - 1) / BK;
  const int per = (nkt_all + g.ksplit - 1) / g.ksplit;
  const int kt0 = (int)blockIdx.z * per;
  const int nkt = (kt0 + per < nkt_all) ? kt0 + per : nkt_all;

  sa.load(A, d.a_rs, d.a_cs, m0, kt0 * BK, d.M, d.K, g.vecA, tid);
  // B(k,n): "row" index of the staged tile is n -> row stride b_cs, k stride b_rs
  sb.load(B, d.b_cs, d.b_rs, n0, kt0 * BK, d.N, d.K, g.vecB, tid);
  sa.store(As, tid);
  sb.store(Bs, tid);
  __syncthreads();

  for (int kt = kt0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) {
      sa.load(A, d.a_rs, d.a_cs, m0, (kt + 1) * BK, d.M, d.K, g.vecA, tid);
      sb.load(B, d.b_cs, d.b_rs, n0, (kt + 1) * BK, d.N, d.K, g.vecB, tid);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        read_frag<BM, A_KC>(As, (wm * TM + i) * 32 + l31, q, h, a[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        read_frag<BN, B_KC>(Bs, (wn * TN + j) * 32 + l31, q, h, b[j]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = fd::mfma_32x32x2(a[i][t], b[j][t], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      sa.store(As, tid);
      sb.store(Bs, tid);
      __syncthreads();
    }
  }

  // ---- epilogue ----
  if (g.ksplit > 1) {
    // split-K partial: C += alpha * partial (atomic; C holds the running sum)
    if (kt0 >= nkt) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= d.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + (wn * TN + j) * 32 + l31;
          if (n < d.N) atomicAdd(C + (long)m * d.ldc + n, d.alpha * acc[i][j][r]);
        }
      }
    return;
  }
  const long nn = (long)d.nres * d.nres;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m >= d.M) continue;
      long prow = 0, qrow = 0;
      if (d.pair_p) {
        prow = m / d.nres;
        qrow = (m / nn) * d.nres + (m % d.nres);
      }
      const float rs = d.rowscale ? d.rowscale[m] : 1.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        if (n >= d.N) continue;
        float v = d.alpha * acc[i][j][r];
        if (d.bias) v += d.bias[n];
        if (d.pair_p) v += d.pair_p[prow * d.ld_pair + n] + d.pair_q[qrow * d.ld_pair + n];
        if (d.relu) v = v > 0.f ? v : 0.f;
        if (d.gate) v = d.gate[(long)m * d.ld_gate + n] > 0.f ? v : 0.f;
        v *= rs;
        if (d.resid) v += d.resid[(long)m * d.ld_resid + n];
        float* cp = C + (long)m * d.ldc + n;
        if (d.beta) v += *cp;
        *cp = v;
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN>
int launch_cfg(const FdGemmDesc& d, hipStream_t stream) {
  GemmArgs g;
  g.d = d;
  g.nblk_m = fd_cdiv(d.M, BM);
  g.nblk_n = fd_cdiv(d.N, BN);
  const bool a_kc = (d.a_cs == 1);
  const bool b_kc = (d.b_rs == 1);
  const int nb = d.batch > 0 ? d.batch : 1;
  auto al4 = [](long x) { return (x & 3) == 0; };
  // vector (16 B) staging is legal when the contiguous index is a multiple of 4
  // everywhere the kernel can touch it.
  if (a_kc)
    g.vecA = fd_aligned16(d.A) && al4(d.a_rs) && al4(d.K) && al4(d.a_so) && al4(d.a_si);
  else
    g.vecA = (d.a_rs == 1) && fd_aligned16(d.A) && al4(d.a_cs) && al4(d.M) && al4(d.a_so) && al4(d.a_si);
  if (b_kc)
    g.vecB = fd_aligned16(d.B) && al4(d.b_cs) && al4(d.K) && al4(d.b_so) && al4(d.b_si);
  else
    g.vecB = (d.b_cs == 1) && fd_aligned16(d.B) && al4(d.b_rs) && al4(d.N) && al4(d.b_so) && al4(d.b_si);
  g.ksplit = d.ksplit > 1 ? d.ksplit : 1;
  {
    const int nkt_all = fd_cdiv(d.K, BK);
    if (g.ksplit > nkt_all) g.ksplit = nkt_all > 0 ? nkt_all : 1;
  }
  dim3 grid(g.nblk_m * g.nblk_n, nb, g.ksplit), block(256, 1, 1);
  if (a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, true, true>), grid, block, 0, stream, g);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, true, false>), grid, block, 0, stream, g);
  else if (!a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, false, true>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<BM, BN, WGM, WGN, false, false>), grid, block, 0, stream, g);
  FD_CHECK_LAUNCH("fd_gemm");
  return FD_OK;
}

}  // namespace

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
  // tile selection: wide tiles when the problem fills the chip, narrow otherwise
  const long blocks128 = (long)fd_cdiv(d.M, 128) * fd_cdiv(d.N, 128) * (d.batch > 0 ? d.batch : 1);
  int cfg = d.tile;
  if (cfg == 0) {
    if (d.N <= 48) cfg = 3;
    else if (blocks128 >= 512 && d.N >= 96) cfg = 1;
    else cfg = 2;
  }
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2>(d, stream);
    case 2: return launch_cfg<64, 64, 2, 2>(d, stream);
    case 3: return launch_cfg<128, 32, 4, 1>(d, stream);
    default: fd_set_error("fd_gemm: bad tile config %d", cfg); return FD_ERR_ARG;
  }
}
