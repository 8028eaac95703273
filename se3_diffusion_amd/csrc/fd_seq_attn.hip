// Sequence-transformer self-attention of one (batch, head, 32-query tile) in ONE kernel: scores, key-padding bias,
// softmax and the value product (torch.nn.TransformerEncoderLayer's self_attn inside IpaScore, model/ipa_pytorch.py:
// 584-593 with nhead = 4, d_model = 320; torch.nn.functional.multi_head_attention_forward's
// softmax(q k^T / sqrt(d) + mask) v).
//
// Replaces three launches per layer (batched q k^T GEMM, row softmax, batched a v GEMM) whose operands are a few hundred
// KB: at sampling sizes they were 18 us of launch latency per layer for ~3 us of work.  fp32 MFMA (v_mfma_f32_32x32x2_f32,
// exact fp32 products and sums) for both contractions, operand fragments straight from global memory in MFMA layout
// (as fd_gemm tile 5), the 32 x N score tile only in LDS.  The probabilities are written to HBM only when the caller
// asks for them (training: the backward needs A).
//
//   qkv [B*N, 960] = [q (4 x 80) | k (4 x 80) | v (4 x 80)] as in_proj leaves them; key_add [B, N] additive mask or null;
//   out [B*N, 320]: head h in columns 80h .. 80h+79.
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int TH = 4, THD = 80, TD = 320, LDQ = 3 * TD;
constexpr int QT = 32;            // query rows per block
constexpr int NG = THD / 8;       // 8-k groups of the score contraction

template <int NMAX>
__global__ __launch_bounds__(256) void seq_attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ key_add,
                                                           float* __restrict__ out, float* __restrict__ A_out, float scale,
                                                           int N) {
  constexpr int SP = NMAX + 4;                       // floats per score row (16-byte aligned rows)
  constexpr int PARTP = 97;                          // floats per row of a wave's partial output tile (96 columns)
  constexpr int LDSF = QT * SP > 4 * QT * PARTP ? QT * SP : 4 * QT * PARTP;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  float* S = lds;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  // 1-D grid, XCD-swizzled as a whole: the query tiles of one (batch, head) -- which all read its K and V -- run on one XCD
  // (a [tiles, heads, batch] grid puts tile t of every head on XCD t % 8: eight L2s fetch the same K / V)
  const int nqt = (N + QT - 1) / QT;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int i0 = (lid % nqt) * QT, hd = (lid / nqt) % TH, b = lid / (nqt * TH);
  const float* base = qkv + (long)b * N * LDQ + hd * THD;

  // ---- scores: S[i][j] = scale * q_i . k_j + key_add[j]; wave w takes the key tiles w, w + 4, ... ----
  const int qi = (i0 + l31 < N) ? i0 + l31 : N - 1;
  float4 qf[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) qf[g] = *reinterpret_cast<const float4*>(base + (long)qi * LDQ + 8 * g + 4 * h);
  const int njt = (N + 31) / 32;
  for (int jt = wave; jt < njt; jt += 4) {
    const int j0 = jt * 32;
    const int kj = (j0 + l31 < N) ? j0 + l31 : N - 1;
    const float* kp = base + TD + (long)kj * LDQ + 4 * h;
    // four independent accumulation chains (a dependent v_mfma_f32_32x32x2_f32 issues every 64 cycles: one chain of 40
    // would be 1.2 us of latency on its own)
    f32x16 pa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) pa[c][r] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float4 kf = *reinterpret_cast<const float4*>(kp + 8 * g);
      pa[0] = fd::mfma_32x32x2(qf[g].x, kf.x, pa[0]);
      pa[1] = fd::mfma_32x32x2(qf[g].y, kf.y, pa[1]);
      pa[2] = fd::mfma_32x32x2(qf[g].z, kf.z, pa[2]);
      pa[3] = fd::mfma_32x32x2(qf[g].w, kf.w, pa[3]);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (pa[0][r] + pa[1][r]) + (pa[2][r] + pa[3][r]);
    // D: reg r -> row (r & 3) + 8 (r >> 2) + 4 h, column l31
    const bool jok = j0 + l31 < N;
    const float ka = (key_add != nullptr && jok) ? key_add[(long)b * N + j0 + l31] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      S[row * SP + j0 + l31] = jok ? acc[r] * scale + ka : -INFINITY;
    }
  }
  __syncthreads();

  // ---- softmax over j: eight lanes per row, all 32 rows at once (reductions stay inside the 8-lane group) ----
  const int n8 = (N + 7) / 8 * 8;                    // the value product walks j in groups of 8: zero the padding
  {
    const int row = tid >> 3, sub = tid & 7;
    float* s = S + row * SP;
    float mx = -INFINITY;
    for (int j = sub; j < N; j += 8) mx = fmaxf(mx, s[j]);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = sub; j < N; j += 8) {
      const float e = expf(s[j] - mx);
      s[j] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
    const bool rok = i0 + row < N;
    float* arow = (A_out != nullptr && rok) ? A_out + (((long)b * TH + hd) * N + i0 + row) * N : nullptr;
    for (int j = sub; j < n8; j += 8) {
      const float a = j < N ? s[j] / sum : 0.f;
      s[j] = a;
      if (arow != nullptr && j < N) arow[j] = a;
    }
  }
  __syncthreads();

  // ---- out[i][c] = sum_j A[i][j] v[j][c]: wave w takes the j groups g = w (mod 4); 3 column tiles (80 of 96 used).
  // The value fragments of a group are 12 dependent-latency loads: two groups are kept in flight. ----
  f32x16 o[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  const float* vbase = base + 2 * TD;
  const int ngroups = n8 / 8;
  int cc[3];
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) cc[ct] = (32 * ct + l31 < THD) ? 32 * ct + l31 : THD - 1;
  auto vload = [&](float (&v)[4][3], int g) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = 8 * g + 4 * h + t;
      const float* vrow = vbase + (long)(j < N ? j : N - 1) * LDQ;     // (A is zero for j >= N)
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) v[t][ct] = vrow[cc[ct]];
    }
  };
  auto vmma = [&](const float (&v)[4][3], int g) __attribute__((always_inline)) {
    const float4 af = *reinterpret_cast<const float4*>(S + l31 * SP + 8 * g + 4 * h);
    const float a4[4] = {af.x, af.y, af.z, af.w};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) o[ct] = fd::mfma_32x32x2(a4[t], v[t][ct], o[ct]);
  };
  float v0[4][3], v1[4][3];
  int g = wave;
  if (g < ngroups) vload(v0, g);
  for (; g < ngroups; g += 8) {
    const bool more = g + 4 < ngroups;
    if (more) vload(v1, g + 4);
    vmma(v0, g);
    if (!more) break;
    if (g + 8 < ngroups) vload(v0, g + 8);
    vmma(v1, g + 4);
  }
  __syncthreads();                                   // every wave is done reading the probabilities
  float* part = lds + wave * (QT * PARTP);
#pragma unroll
  for (int ct = 0; ct < 3; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[((r & 3) + 8 * (r >> 2) + 4 * h) * PARTP + 32 * ct + l31] = o[ct][r];
  __syncthreads();
  for (int e = tid; e < QT * THD; e += 256) {
    const int row = e / THD, c = e % THD;
    if (i0 + row >= N) continue;
    const float v = (lds[row * PARTP + c] + lds[QT * PARTP + row * PARTP + c]) +
                    (lds[2 * QT * PARTP + row * PARTP + c] + lds[3 * QT * PARTP + row * PARTP + c]);
    out[((long)b * N + i0 + row) * TD + hd * THD + c] = v;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Backward of that attention in two launches instead of five (dA = dO V^T GEMM, dV = A^T dO GEMM, row softmax backward,
// dQ = dS K GEMM, dK = dS^T Q GEMM: 120 batches of 128 x 80 x 128 at B=30 x N=128 -- 15 us of launch latency each for ~2 us
// of work).  One wave owns a tile of 16 rows of one (batch, head); no LDS, no barrier; fp32 MFMA 16x16x4 (exact fp32).
//   query side (a wave per 16 query rows): dP^T = V dO^T, dS = A (dP - D) with D_i = dO_i . o_i (the saved attention
//     output: no pass over the keys), dS written once (in place of dA), dQ^T += K^T dS^T
//   key side (a wave per 16 key rows): dV^T += dO^T A, dK^T += Q^T dS over the query tiles
// Operand layouts as in fd_ipa_flash.hip: "K layout" = lane (row l & 15, k group l >> 4) holds 4 consecutive channels of a
// 16-channel chunk, the MFMA k-steps of a chunk contract the channels {4 kk' + s}; "V layout" = lane (m = l & 15, kk) holds
// the channels 64 cb + 4 m .. + 3 of row 4 kk + r, output tile q of block cb has channel 64 cb + 4 m + q in its row m.
constexpr int TQ = 16;
constexpr int NCH = THD / 16;      // 5 chunks of 16 channels
__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ f32x4 zero_f4() { f32x4 z; z[0] = z[1] = z[2] = z[3] = 0.f; return z; }
__device__ __forceinline__ int mini(int a, int b) { return a < b ? a : b; }

__global__ __launch_bounds__(256) void seq_attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ A,
                                                             const float* __restrict__ dout, const float* __restrict__ out,
                                                             float* __restrict__ dS, float* __restrict__ dqkv, float scale,
                                                             int N, int units) {
  const int nti = (N + TQ - 1) / TQ;
  const int unit = (int)blockIdx.x * 4 + fd::uniform(fd::wave_id());
  if (unit >= units) return;
  const int it = unit % nti, hd = (unit / nti) % TH, b = unit / (nti * TH);
  const int lane = fd::lane_id(), n = lane & 15, kk = lane >> 4;
  const int i0 = it * TQ;
  const long rb = (long)b * N;
  const long ri = rb + mini(i0 + n, N - 1);
  const bool row_ok = i0 + n < N;
  const bool vec = (N & 3) == 0;
  // dO^T as B operand, D = dO . o
  float4 dOf[NCH];
  float D = 0.f;
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc) {
    dOf[cc] = ldf4(dout + ri * TD + hd * THD + 16 * cc + 4 * kk);
    const float4 o = ldf4(out + ri * TD + hd * THD + 16 * cc + 4 * kk);
    D += dOf[cc].x * o.x + dOf[cc].y * o.y + dOf[cc].z * o.z + dOf[cc].w * o.w;
  }
  D += __shfl_xor(D, 16);
  D += __shfl_xor(D, 32);
  const float* __restrict__ kb = qkv + rb * LDQ + TD + hd * THD;          // K rows of the head
  const float* __restrict__ vb = qkv + rb * LDQ + 2 * TD + hd * THD;      // V rows
  const long arow = (((long)b * TH + hd) * N + mini(i0 + n, N - 1)) * N;
  f32x4 dQ[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) dQ[c] = zero_f4();
  for (int jt = 0; jt < nti; ++jt) {
    const int j0 = jt * TQ;
    float p[4];
    {
      const int j = j0 + 4 * kk;
      if (vec && j + 3 < N) {
        const float4 v = ldf4(A + arow + j);
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = j + r < N ? A[arow + j + r] : 0.f;
      }
    }
    // dP^T[j][i] = V_j . dO_i
    f32x4 s0 = zero_f4(), s1 = zero_f4();
    {
      const float* vr = vb + (long)mini(j0 + n, N - 1) * LDQ + 4 * kk;
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) {
        const float4 v = ldf4(vr + 16 * cc);
        f32x4& acc = (cc & 1) ? s1 : s0;
        acc = fd::mfma_16x16x4(v.x, dOf[cc].x, acc);
        acc = fd::mfma_16x16x4(v.y, dOf[cc].y, acc);
        acc = fd::mfma_16x16x4(v.z, dOf[cc].z, acc);
        acc = fd::mfma_16x16x4(v.w, dOf[cc].w, acc);
      }
    }
    float ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[r] = p[r] * ((s0[r] + s1[r]) - D);
    if (row_ok) {
      const int j = j0 + 4 * kk;
      if (vec && j + 3 < N) {
        *reinterpret_cast<float4*>(dS + arow + j) = make_float4(ds[0], ds[1], ds[2], ds[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j + r < N) dS[arow + j + r] = ds[r];
      }
    }
    // dQ^T[c][i] += K[j][c] dS^T[j][i]: k-step r contracts the keys {j0 + 4 kk' + r}
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* kr = kb + (long)mini(j0 + 4 * kk + r, N - 1) * LDQ + 4 * n;
      const float4 k0 = ldf4(kr);
      const float4 k1 = n < 4 ? ldf4(kr + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
      dQ[0] = fd::mfma_16x16x4(k0.x, ds[r], dQ[0]);
      dQ[1] = fd::mfma_16x16x4(k0.y, ds[r], dQ[1]);
      dQ[2] = fd::mfma_16x16x4(k0.z, ds[r], dQ[2]);
      dQ[3] = fd::mfma_16x16x4(k0.w, ds[r], dQ[3]);
      dQ[4] = fd::mfma_16x16x4(k1.x, ds[r], dQ[4]);
      dQ[5] = fd::mfma_16x16x4(k1.y, ds[r], dQ[5]);
      dQ[6] = fd::mfma_16x16x4(k1.z, ds[r], dQ[6]);
      dQ[7] = fd::mfma_16x16x4(k1.w, ds[r], dQ[7]);
    }
  }
  // C layout: lane (n = query row, kk), register r -> operand row m = 4 kk + r -> channels 64 cb + 4 m + q
  if (row_ok) {
    float* __restrict__ dq = dqkv + ri * LDQ + hd * THD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<float4*>(dq + 16 * kk + 4 * r) =
          make_float4(scale * dQ[0][r], scale * dQ[1][r], scale * dQ[2][r], scale * dQ[3][r]);
      if (kk == 0)
        *reinterpret_cast<float4*>(dq + 64 + 4 * r) =
            make_float4(scale * dQ[4][r], scale * dQ[5][r], scale * dQ[6][r], scale * dQ[7][r]);
    }
  }
}

__global__ __launch_bounds__(256) void seq_attn_bwd_k_kernel(const float* __restrict__ qkv, const float* __restrict__ A,
                                                             const float* __restrict__ dout, const float* __restrict__ dS,
                                                             float* __restrict__ dqkv, float scale, int N, int units) {
  const int nti = (N + TQ - 1) / TQ;
  const int unit = (int)blockIdx.x * 4 + fd::uniform(fd::wave_id());
  if (unit >= units) return;
  const int jt = unit % nti, hd = (unit / nti) % TH, b = unit / (nti * TH);
  const int lane = fd::lane_id(), n = lane & 15, kk = lane >> 4;
  const int j0 = jt * TQ;
  const long rb = (long)b * N;
  const bool key_ok = j0 + n < N;
  const int jc = mini(j0 + n, N - 1);
  const float* __restrict__ qb = qkv + rb * LDQ + hd * THD;               // Q rows of the head
  const float* __restrict__ db = dout + rb * TD + hd * THD;               // dO rows
  const long abase = ((long)b * TH + hd) * N * N + jc;                    // column jc of the (batch, head) matrix
  f32x4 dV[8], dK[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { dV[c] = zero_f4(); dK[c] = zero_f4(); }
  for (int it = 0; it < nti; ++it) {
    const int i0 = it * TQ;
    // B operands [k = query row i0 + 4 kk + r][n = key]: A and dS
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * kk + r;
      const bool ok = i < N && key_ok;
      const int ic = mini(i, N - 1);
      const float a = ok ? A[abase + (long)ic * N] : 0.f;
      const float g = ok ? dS[abase + (long)ic * N] : 0.f;
      const float* dr = db + (long)ic * TD + 4 * n;
      const float* qr = qb + (long)ic * LDQ + 4 * n;
      const float4 d0 = ldf4(dr), q0 = ldf4(qr);
      const float4 d1 = n < 4 ? ldf4(dr + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 q1 = n < 4 ? ldf4(qr + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
      dV[0] = fd::mfma_16x16x4(d0.x, a, dV[0]);
      dV[1] = fd::mfma_16x16x4(d0.y, a, dV[1]);
      dV[2] = fd::mfma_16x16x4(d0.z, a, dV[2]);
      dV[3] = fd::mfma_16x16x4(d0.w, a, dV[3]);
      dV[4] = fd::mfma_16x16x4(d1.x, a, dV[4]);
      dV[5] = fd::mfma_16x16x4(d1.y, a, dV[5]);
      dV[6] = fd::mfma_16x16x4(d1.z, a, dV[6]);
      dV[7] = fd::mfma_16x16x4(d1.w, a, dV[7]);
      dK[0] = fd::mfma_16x16x4(q0.x, g, dK[0]);
      dK[1] = fd::mfma_16x16x4(q0.y, g, dK[1]);
      dK[2] = fd::mfma_16x16x4(q0.z, g, dK[2]);
      dK[3] = fd::mfma_16x16x4(q0.w, g, dK[3]);
      dK[4] = fd::mfma_16x16x4(q1.x, g, dK[4]);
      dK[5] = fd::mfma_16x16x4(q1.y, g, dK[5]);
      dK[6] = fd::mfma_16x16x4(q1.z, g, dK[6]);
      dK[7] = fd::mfma_16x16x4(q1.w, g, dK[7]);
    }
  }
  if (key_ok) {
    float* __restrict__ dk = dqkv + (rb + j0 + n) * LDQ + TD + hd * THD;
    float* __restrict__ dv = dqkv + (rb + j0 + n) * LDQ + 2 * TD + hd * THD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<float4*>(dk + 16 * kk + 4 * r) =
          make_float4(scale * dK[0][r], scale * dK[1][r], scale * dK[2][r], scale * dK[3][r]);
      *reinterpret_cast<float4*>(dv + 16 * kk + 4 * r) = make_float4(dV[0][r], dV[1][r], dV[2][r], dV[3][r]);
      if (kk == 0) {
        *reinterpret_cast<float4*>(dk + 64 + 4 * r) =
            make_float4(scale * dK[4][r], scale * dK[5][r], scale * dK[6][r], scale * dK[7][r]);
        *reinterpret_cast<float4*>(dv + 64 + 4 * r) = make_float4(dV[4][r], dV[5][r], dV[6][r], dV[7][r]);
      }
    }
  }
}

}  // namespace

extern "C" int fd_seq_attn_fwd(const float* qkv, const float* key_add, float* out, float* A_out, float scale, int B, int N,
                               void* stream) {
  FD_CHECK_ARG(qkv && out, "fd_seq_attn_fwd: null operand");
  FD_CHECK_ARG(fd_aligned16(qkv), "fd_seq_attn_fwd: qkv must be 16-byte aligned");
  FD_CHECK_ARG(N <= 1024, "fd_seq_attn_fwd: N=%d exceeds 1024", N);
  if (B == 0 || N == 0) return FD_OK;
  const dim3 grid((unsigned)((N + QT - 1) / QT) * TH * (unsigned)B);
  if (N <= 256)
    hipLaunchKernelGGL(seq_attn_fwd_kernel<256>, grid, dim3(256), 0, (hipStream_t)stream, qkv, key_add, out, A_out, scale, N);
  else
    hipLaunchKernelGGL(seq_attn_fwd_kernel<1024>, grid, dim3(256), 0, (hipStream_t)stream, qkv, key_add, out, A_out, scale, N);
  FD_CHECK_LAUNCH("fd_seq_attn_fwd");
  return FD_OK;
}

extern "C" int fd_seq_attn_bwd(const float* qkv, const float* A, const float* dout, const float* out, float* dS,
                               float* dqkv, float scale, int B, int N, void* stream) {
  FD_CHECK_ARG(qkv && A && dout && out && dS && dqkv, "fd_seq_attn_bwd: null operand");
  FD_CHECK_ARG(fd_aligned16(qkv) && fd_aligned16(A) && fd_aligned16(dout) && fd_aligned16(out) && fd_aligned16(dS) &&
                   fd_aligned16(dqkv),
               "fd_seq_attn_bwd: operands must be 16-byte aligned");
  FD_CHECK_ARG(N <= 1024, "fd_seq_attn_bwd: N=%d exceeds 1024", N);
  if (B == 0 || N == 0) return FD_OK;
  const int units = B * TH * ((N + TQ - 1) / TQ);
  const dim3 grid((unsigned)((units + 3) / 4));
  hipLaunchKernelGGL(seq_attn_bwd_q_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, A, dout, out, dS, dqkv, scale, N,
                     units);
  FD_CHECK_LAUNCH("fd_seq_attn_bwd (query side)");
  hipLaunchKernelGGL(seq_attn_bwd_k_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, A, dout, (const float*)dS, dqkv,
                     scale, N, units);
  FD_CHECK_LAUNCH("fd_seq_attn_bwd (key side)");
  return FD_OK;
}
