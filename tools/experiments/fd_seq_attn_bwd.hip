// PARKED (round 4 -> removed from the product library in round 5): the sequence-transformer attention backward in two launches
// (query side: dS = A (dO V^T - D), dQ; key side: dV, dK).  Parity-green, measured SLOWER than the four batched GEMMs + row-softmax
// backward it replaced (58 against 54 us per layer at B=30 x N=128, 99 against 52 at B=7 x N=256: one wave per SIMD, a dependent
// load chain per tile; profiles/r04_ab.txt).  Kept here for the record; it needs the helpers of csrc/fd_seq_attn.hip to compile.

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
