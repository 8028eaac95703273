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
