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


// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the same attention in ONE launch (round 6): autograd of softmax(q k^T / sqrt(d) + mask) v with respect to q, k, v
// (torch.nn.functional.multi_head_attention_forward inside nn.TransformerEncoderLayer, model/ipa_pytorch.py:584-593) from the saved
// probabilities A and the saved output o.  Replaces dA = dO V^T, dV = A^T dO (batched GEMMs), the row-softmax backward, dQ = dS K and
// dK = dS^T Q (batched GEMMs): five launches of ~14 us per layer for ~3 us of work, and dA / dS never reach HBM.
//   dS_ij = A_ij (dO_i . V_j - D_i),  D_i = dO_i . o_i;   dQ_i = scale sum_j dS_ij K_j;   dK_j = scale sum_i dS_ij Q_i;   dV_j = sum_i A_ij dO_i
// Decomposition (the key-side kernel of fd_ipa_flash.hip with both roles): a block owns FOUR consecutive 16-row tiles of one
// (batch, head), one wave each, either as QUERY tiles (role 0: walks the key tiles, dS^T = A^T o (V dO^T - D), dQ^T += K^T dS^T) or as
// KEY tiles (role 1: walks the query tiles, recomputes dS = A o (dO V^T - D) -- 20 MFMAs -- and accumulates dV^T += dO^T A,
// dK^T += Q^T dS).  The rows of the OTHER side (16 x 80 floats per operand) go global -> registers -> LDS once per block and step, in
// two images: "K layout" (row stride 544 B: a lane reads 16 bytes of ITS row per 16-channel chunk -- the strides that keep gfx950's
// ds_read_b128 lane groups conflict-free are = 32 mod 256) for the operand whose rows are the MFMA's rows, and "V layout" (row stride
// 512 B: lanes 0..15 read 256 contiguous bytes of row 4 s + kk) for the transposed operand.  v_mfma_f32_16x16x4_f32 (exact fp32).
// The C layout of dS (rows 4 kk + r, column n) IS the B-operand layout of the product that consumes it: no exchange.
constexpr int SB_KS = 136, SB_VS = 128;            // floats per row of the K-layout / V-layout images
constexpr int SB_T = 16;
constexpr int SB_NCH = THD / 16;                   // 5 chunks of 16 channels

struct SeqBwdArgs {
  const float *qkv, *A, *dout, *out;
  float* dqkv;
  float scale;
  int B, N;
};

__global__ __launch_bounds__(256, 2) void seq_attn_bwd_kernel(SeqBwdArgs a) {
  // [buffer][image][row][..]: role 1 uses 4 images (dO K-layout, dO V-layout, Q V-layout, o K-layout), role 0 two (V K-layout, K V-layout)
  __shared__ __attribute__((aligned(16))) float img[2][2 * SB_T * SB_KS + 2 * SB_T * SB_VS];
  const int N = a.N;
  const int nti = (N + SB_T - 1) / SB_T, ngr = (nti + 3) / 4;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int role = lid & 1, gr = (lid >> 1) % ngr, hd = ((lid >> 1) / ngr) % TH, b = (lid >> 1) / (ngr * TH);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kk = lane >> 4;
  const int t0 = (gr * 4 + wave) * SB_T;             // first row of this wave's tile (past N: computes on clamped rows, stores nothing)
  const long rb = (long)b * N;
  const bool vec = (N & 3) == 0;
  auto ldv = [](const float* p) -> f32x4 { return *reinterpret_cast<const f32x4*>(p); };
  auto zero4 = []() -> f32x4 { f32x4 z; z[0] = z[1] = z[2] = z[3] = 0.f; return z; };
  const float* __restrict__ Ab = a.A + ((long)b * TH + hd) * N * N;
  // staging: float4 f = tid (+ 256): row f / 20, 16-byte chunk f % 20 of a 16 x 80 tile
  const int f0r = tid / 20, f0c = tid % 20, f1r = (tid + 256) / 20, f1c = (tid + 256) % 20;
  const bool has1 = tid + 256 < SB_T * 20;
  float* const KI0 = img[0];                           // image offsets inside a buffer
  constexpr int OFF_K0 = 0, OFF_K1 = SB_T * SB_KS, OFF_V0 = 2 * SB_T * SB_KS, OFF_V1 = 2 * SB_T * SB_KS + SB_T * SB_VS;
  (void)KI0;

  if (role == 0) {
    // ------------------------------------------------------------------ query tiles: dQ
    const int i0 = t0;
    const long ri = rb + (i0 + n < N ? i0 + n : N - 1);
    // dO_i^T as B operand (K layout: lane (row n, kk) holds channels 16 cc + 4 kk ..), D_i = dO_i . o_i
    f32x4 dOf[SB_NCH];
    float D = 0.f;
#pragma unroll
    for (int cc = 0; cc < SB_NCH; ++cc) {
      dOf[cc] = ldv(a.dout + ri * TD + hd * THD + 16 * cc + 4 * kk);
      const f32x4 o = ldv(a.out + ri * TD + hd * THD + 16 * cc + 4 * kk);
      D += dOf[cc][0] * o[0] + dOf[cc][1] * o[1] + dOf[cc][2] * o[2] + dOf[cc][3] * o[3];
    }
    D += __shfl_xor(D, 16);
    D += __shfl_xor(D, 32);
    const float* __restrict__ Arow = Ab + (long)(i0 + n < N ? i0 + n : N - 1) * N;
    const bool row_ok = i0 + n < N;
    f32x4 dQ[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dQ[c] = zero4();
    // staging of a key tile: V rows (K layout), K rows (V layout)
    const float* __restrict__ vsrc = a.qkv + rb * LDQ + 2 * TD + hd * THD;
    const float* __restrict__ ksrc = a.qkv + rb * LDQ + TD + hd * THD;
    f32x4 sv[2], sk[2];
    auto fetch = [&](int jt) __attribute__((always_inline)) {
      const int j0 = jt * SB_T;
      const long r0 = (long)(j0 + f0r < N ? j0 + f0r : N - 1) * LDQ;
      sv[0] = ldv(vsrc + r0 + 4 * f0c);
      sk[0] = ldv(ksrc + r0 + 4 * f0c);
      if (has1) {
        const long r1 = (long)(j0 + f1r < N ? j0 + f1r : N - 1) * LDQ;
        sv[1] = ldv(vsrc + r1 + 4 * f1c);
        sk[1] = ldv(ksrc + r1 + 4 * f1c);
      }
    };
    auto deposit = [&](int buf) __attribute__((always_inline)) {
      float* im = img[buf];
      *reinterpret_cast<f32x4*>(im + OFF_K0 + f0r * SB_KS + 4 * f0c) = sv[0];
      *reinterpret_cast<f32x4*>(im + OFF_V0 + f0r * SB_VS + 4 * f0c) = sk[0];
      if (has1) {
        *reinterpret_cast<f32x4*>(im + OFF_K0 + f1r * SB_KS + 4 * f1c) = sv[1];
        *reinterpret_cast<f32x4*>(im + OFF_V0 + f1r * SB_VS + 4 * f1c) = sk[1];
      }
    };
    fetch(0);
    deposit(0);
    __syncthreads();
#pragma unroll 1
    for (int jt = 0; jt < nti; ++jt) {
      const int buf = jt & 1, j0 = jt * SB_T;
      if (jt + 1 < nti) fetch(jt + 1);
      const float* im = img[buf];
      // dP^T[j, i] = V_j . dO_i  (A operand: V rows from the K-layout image)
      f32x4 p0 = zero4(), p1 = zero4();
#pragma unroll
      for (int cc = 0; cc < SB_NCH; ++cc) {
        const f32x4 vf = ldv(im + OFF_K0 + n * SB_KS + 16 * cc + 4 * kk);
        p0 = fd::mfma_16x16x4(vf[0], dOf[cc][0], p0);
        p1 = fd::mfma_16x16x4(vf[1], dOf[cc][1], p1);
        p0 = fd::mfma_16x16x4(vf[2], dOf[cc][2], p0);
        p1 = fd::mfma_16x16x4(vf[3], dOf[cc][3], p1);
      }
      // dS^T[j = j0 + 4 kk + r, i = n] = A[i, j] (dP - D_i)
      float pa[4];
      {
        const int j = j0 + 4 * kk;
        if (vec && j + 3 < N) {
          const f32x4 v = ldv(Arow + j);
          pa[0] = v[0]; pa[1] = v[1]; pa[2] = v[2]; pa[3] = v[3];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) pa[r] = j + r < N ? Arow[j + r] : 0.f;
        }
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[r] = row_ok ? pa[r] * ((p0[r] + p1[r]) - D) : 0.f;
      // dQ^T[c, i] += K^T[c, j] dS^T[j, i]  (A operand: K rows 4 kk' + r from the V-layout image, channels 64 cb + 4 m + q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* krow = im + OFF_V0 + (4 * kk + r) * SB_VS + 4 * n;
        const f32x4 k0 = ldv(krow);
        dQ[0] = fd::mfma_16x16x4(k0[0], ds[r], dQ[0]);
        dQ[1] = fd::mfma_16x16x4(k0[1], ds[r], dQ[1]);
        dQ[2] = fd::mfma_16x16x4(k0[2], ds[r], dQ[2]);
        dQ[3] = fd::mfma_16x16x4(k0[3], ds[r], dQ[3]);
        f32x4 k1 = zero4();
        if (n < 4) k1 = ldv(krow + 64);
        dQ[4] = fd::mfma_16x16x4(k1[0], ds[r], dQ[4]);
        dQ[5] = fd::mfma_16x16x4(k1[1], ds[r], dQ[5]);
        dQ[6] = fd::mfma_16x16x4(k1[2], ds[r], dQ[6]);
        dQ[7] = fd::mfma_16x16x4(k1[3], ds[r], dQ[7]);
      }
      if (jt + 1 < nti) deposit(buf ^ 1);
      __syncthreads();
    }
    // D layout: register r of tile 4 cb + q = channel 64 cb + 4 (4 kk + r) + q of query n
    if (row_ok) {
      float* __restrict__ dq = a.dqkv + (rb + i0 + n) * LDQ + hd * THD + 16 * kk;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(dq + 4 * r) = make_float4(a.scale * dQ[0][r], a.scale * dQ[1][r], a.scale * dQ[2][r], a.scale * dQ[3][r]);
      if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float4*>(dq + 64 + 4 * r) = make_float4(a.scale * dQ[4][r], a.scale * dQ[5][r], a.scale * dQ[6][r], a.scale * dQ[7][r]);
      }
    }
    return;
  }

  // -------------------------------------------------------------------- key tiles: dV, dK
  const int j0 = t0;
  const int jn = j0 + n < N ? j0 + n : N - 1;
  const bool key_ok = j0 + n < N;
  // V_j^T as B operand (K layout)
  f32x4 Vf[SB_NCH];
#pragma unroll
  for (int cc = 0; cc < SB_NCH; ++cc) Vf[cc] = ldv(a.qkv + (rb + jn) * LDQ + 2 * TD + hd * THD + 16 * cc + 4 * kk);
  f32x4 dV[8], dK[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { dV[c] = zero4(); dK[c] = zero4(); }
  const float* __restrict__ dosrc = a.dout + rb * TD + hd * THD;
  const float* __restrict__ osrc = a.out + rb * TD + hd * THD;
  const float* __restrict__ qsrc = a.qkv + rb * LDQ + hd * THD;
  f32x4 sd[2], so[2], sq[2];
  auto fetch = [&](int it) __attribute__((always_inline)) {
    const int i0 = it * SB_T;
    const long r0 = (long)(i0 + f0r < N ? i0 + f0r : N - 1);
    sd[0] = ldv(dosrc + r0 * TD + 4 * f0c);
    so[0] = ldv(osrc + r0 * TD + 4 * f0c);
    sq[0] = ldv(qsrc + r0 * LDQ + 4 * f0c);
    if (has1) {
      const long r1 = (long)(i0 + f1r < N ? i0 + f1r : N - 1);
      sd[1] = ldv(dosrc + r1 * TD + 4 * f1c);
      so[1] = ldv(osrc + r1 * TD + 4 * f1c);
      sq[1] = ldv(qsrc + r1 * LDQ + 4 * f1c);
    }
  };
  auto deposit = [&](int buf) __attribute__((always_inline)) {
    float* im = img[buf];
    *reinterpret_cast<f32x4*>(im + OFF_K0 + f0r * SB_KS + 4 * f0c) = sd[0];
    *reinterpret_cast<f32x4*>(im + OFF_K1 + f0r * SB_KS + 4 * f0c) = so[0];
    *reinterpret_cast<f32x4*>(im + OFF_V0 + f0r * SB_VS + 4 * f0c) = sd[0];
    *reinterpret_cast<f32x4*>(im + OFF_V1 + f0r * SB_VS + 4 * f0c) = sq[0];
    if (has1) {
      *reinterpret_cast<f32x4*>(im + OFF_K0 + f1r * SB_KS + 4 * f1c) = sd[1];
      *reinterpret_cast<f32x4*>(im + OFF_K1 + f1r * SB_KS + 4 * f1c) = so[1];
      *reinterpret_cast<f32x4*>(im + OFF_V0 + f1r * SB_VS + 4 * f1c) = sd[1];
      *reinterpret_cast<f32x4*>(im + OFF_V1 + f1r * SB_VS + 4 * f1c) = sq[1];
    }
  };
  fetch(0);
  deposit(0);
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < nti; ++it) {
    const int buf = it & 1, i0 = it * SB_T;
    if (it + 1 < nti) fetch(it + 1);
    const float* im = img[buf];
    // dP[i, j] = dO_i . V_j (A operand: dO rows from the K-layout image) and D_i = dO_i . o_i of the tile's rows
    f32x4 p0 = zero4(), p1 = zero4();
    float Dn = 0.f;
#pragma unroll
    for (int cc = 0; cc < SB_NCH; ++cc) {
      const f32x4 df = ldv(im + OFF_K0 + n * SB_KS + 16 * cc + 4 * kk);
      const f32x4 of = ldv(im + OFF_K1 + n * SB_KS + 16 * cc + 4 * kk);
      Dn += df[0] * of[0] + df[1] * of[1] + df[2] * of[2] + df[3] * of[3];
      p0 = fd::mfma_16x16x4(df[0], Vf[cc][0], p0);
      p1 = fd::mfma_16x16x4(df[1], Vf[cc][1], p1);
      p0 = fd::mfma_16x16x4(df[2], Vf[cc][2], p0);
      p1 = fd::mfma_16x16x4(df[3], Vf[cc][3], p1);
    }
    Dn += __shfl_xor(Dn, 16);
    Dn += __shfl_xor(Dn, 32);                 // D of row i0 + n, in every lane with that n
    // A[i = i0 + 4 kk + r, j = j0 + n]; dS = A (dP - D_i)
    float pa[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * kk + r;
      const float v = Ab[(long)(i < N ? i : N - 1) * N + jn];
      pa[r] = (i < N && key_ok) ? v : 0.f;
      const float Di = __shfl(Dn, 4 * kk + r);
      ds[r] = pa[r] * ((p0[r] + p1[r]) - Di);
    }
    // dV^T[c, j] += dO^T[c, i] A[i, j];  dK^T[c, j] += Q^T[c, i] dS[i, j]   (A operands: rows 4 kk' + r of the V-layout images)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* drow = im + OFF_V0 + (4 * kk + r) * SB_VS + 4 * n;
      const float* qrow = im + OFF_V1 + (4 * kk + r) * SB_VS + 4 * n;
      const f32x4 d0 = ldv(drow), q0 = ldv(qrow);
      f32x4 d1 = zero4(), q1 = zero4();
      if (n < 4) { d1 = ldv(drow + 64); q1 = ldv(qrow + 64); }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dV[q] = fd::mfma_16x16x4(d0[q], pa[r], dV[q]);
        dK[q] = fd::mfma_16x16x4(q0[q], ds[r], dK[q]);
        dV[4 + q] = fd::mfma_16x16x4(d1[q], pa[r], dV[4 + q]);
        dK[4 + q] = fd::mfma_16x16x4(q1[q], ds[r], dK[4 + q]);
      }
    }
    if (it + 1 < nti) deposit(buf ^ 1);
    __syncthreads();
  }
  if (key_ok) {
    float* __restrict__ dk = a.dqkv + (rb + j0 + n) * LDQ + TD + hd * THD + 16 * kk;
    float* __restrict__ dv = dk + TD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<float4*>(dk + 4 * r) = make_float4(a.scale * dK[0][r], a.scale * dK[1][r], a.scale * dK[2][r], a.scale * dK[3][r]);
      *reinterpret_cast<float4*>(dv + 4 * r) = make_float4(dV[0][r], dV[1][r], dV[2][r], dV[3][r]);
    }
    if (kk == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<float4*>(dk + 64 + 4 * r) = make_float4(a.scale * dK[4][r], a.scale * dK[5][r], a.scale * dK[6][r], a.scale * dK[7][r]);
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

extern "C" int fd_seq_attn_bwd(const float* qkv, const float* A, const float* dout, const float* out, float* dqkv, float scale, int B,
                               int N, void* stream) {
  FD_CHECK_ARG(qkv && A && dout && out && dqkv, "fd_seq_attn_bwd: null operand");
  FD_CHECK_ARG(fd_aligned16(qkv) && fd_aligned16(A) && fd_aligned16(dout) && fd_aligned16(out) && fd_aligned16(dqkv),
               "fd_seq_attn_bwd: tensor arguments must be 16-byte aligned");
  FD_CHECK_ARG(N <= 1024, "fd_seq_attn_bwd: N=%d exceeds 1024", N);
  if (B == 0 || N == 0) return FD_OK;
  const int ngr = ((N + SB_T - 1) / SB_T + 3) / 4;
  SeqBwdArgs a{qkv, A, dout, out, dqkv, scale, B, N};
  hipLaunchKernelGGL(seq_attn_bwd_kernel, dim3((unsigned)((long)B * TH * ngr * 2)), dim3(256), 0, (hipStream_t)stream, a);
  FD_CHECK_LAUNCH("fd_seq_attn_bwd");
  return FD_OK;
}
