// LayerNorm folded into the Linear that consumes it, for the node-level launches of sampling (M = B*N <= 1024 rows):
//   out = epi( (rowscale * (LayerNorm(x) * gamma + beta)) W^T + bias ),   epi: ReLU, + residual
// -- norm1 -> linear1 and norm2 -> (next layer's in_proj | post_tfmr) of the sequence transformer (torch.nn.TransformerEncoderLayer,
// post-norm, built at model/ipa_pytorch.py:584-595; post_tfmr :638).  At these sizes a LayerNorm launch is ~5 us of dependent
// latency for ~1 us of work, 16 of the ~165 launches of a forward.
//
// Same shape as the latency GEMM (fd_gemm tile 5, fd_gemm_direct.h): one block = one 32 x 32 output tile, its four waves split K
// (wave w takes the 8-k groups g = w mod 4), operands go global -> registers in MFMA layout, v_mfma_f32_32x32x2_f32 (exact fp32
// products and sums), partial tiles meet in LDS.  A block reads its 32 rows of x over ALL of K anyway, so it also forms their
// statistics: every wave keeps its K quarter of the rows in registers (K <= 320: at most 10 float4 per lane), the row sums
// meet in LDS (two-pass: mean, then centred squares -- the arithmetic of fd_layernorm_fwd), the registers are normalised in
// place and multiplied.  The column blocks repeat the statistics of their rows (32 x K floats from L2); block column 0 also
// writes the normalised rows out when a later launch needs them (the residual of the layer behind).
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int LG_MAXG = 10;     // 8-k groups per wave: K <= 4 * 8 * 10 = 320

__global__ __launch_bounds__(256) void ln_gemm_kernel(FdLnGemmDesc d, int nblk_n) {
  __shared__ float part[4][32][33];
  __shared__ float red[2][4][32];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int bm = (int)blockIdx.x / nblk_n, bn = (int)blockIdx.x % nblk_n;
  const int m0 = bm * 32, n0 = bn * 32;
  // rows / columns past the end are clamped: their results are never stored
  const int ra = (m0 + l31 < d.M) ? m0 + l31 : d.M - 1;
  const int rb = (n0 + l31 < d.N) ? n0 + l31 : d.N - 1;
  const float* pa = d.x + (long)ra * d.ldx + 4 * h;
  const float* pb = d.W + (long)rb * d.ldw + 4 * h;

  // epilogue operands of this thread's outputs (thread -> row tid >> 3, columns 4 (tid & 7) ..), fetched first
  const int erow = tid >> 3, ec4 = tid & 7;
  const int em = m0 + erow;
  const bool erow_ok = em < d.M;
  float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_res[4] = {0.f, 0.f, 0.f, 0.f};
  if (erow_ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + 4 * ec4 + e;
      if (n < d.N) {
        if (d.bias) e_bias[e] = d.bias[n];
        if (d.resid) e_res[e] = d.resid[(long)em * d.ld_resid + n];
      }
    }
  }

  const int ngroups = d.K / 8;                       // K % 8 == 0, K <= 320 (checked by the host)
  const int nmine = (ngroups - wave + 3) >> 2;       // this wave's groups g = wave, wave + 4, ...: 0..10 of them
  float av[LG_MAXG][4], bv[LG_MAXG][4], gm[LG_MAXG][4], bt[LG_MAXG][4];
#pragma unroll
  for (int u = 0; u < LG_MAXG; ++u)
    if (u < nmine) {
      const int k = 8 * (wave + 4 * u);
      const float4 a = *reinterpret_cast<const float4*>(pa + k);
      const float4 b = *reinterpret_cast<const float4*>(pb + k);
      float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.ln_cols <= 0 || k < d.ln_cols) {           // (gamma / beta are [ln_cols])
        g4 = *reinterpret_cast<const float4*>(d.gamma + k + 4 * h);
        b4 = *reinterpret_cast<const float4*>(d.beta + k + 4 * h);
      }
      av[u][0] = a.x; av[u][1] = a.y; av[u][2] = a.z; av[u][3] = a.w;
      bv[u][0] = b.x; bv[u][1] = b.y; bv[u][2] = b.z; bv[u][3] = b.w;
      gm[u][0] = g4.x; gm[u][1] = g4.y; gm[u][2] = g4.z; gm[u][3] = g4.w;
      bt[u][0] = b4.x; bt[u][1] = b4.y; bt[u][2] = b4.z; bt[u][3] = b4.w;
    }
  const float rsl = d.ln_rowscale ? d.ln_rowscale[ra] : 1.f;
  // ln_cols < K: only the first ln_cols columns of x are normalised (statistics over them alone), the rest pass through -- the
  // concatenation [LayerNorm(ipa output) | skip_embed] that the first transformer layer reads (ipa_pytorch.py:632-636)
  const int lnc = d.ln_cols > 0 ? d.ln_cols : d.K;
  const int nln = (lnc / 8 - wave + 3) >> 2;         // this wave's groups inside the normalised range (ln_cols % 8 == 0)

  // ---- row statistics: lane (row l31, half h) of wave w holds 4 * nln of the row's normalised K values ----
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < LG_MAXG; ++u)
    if (u < nln) s += (av[u][0] + av[u][1]) + (av[u][2] + av[u][3]);
  s += __shfl_xor(s, 32);
  if (h == 0) red[0][wave][l31] = s;
  __syncthreads();
  const float mean = ((red[0][0][l31] + red[0][1][l31]) + (red[0][2][l31] + red[0][3][l31])) / (float)lnc;
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < LG_MAXG; ++u)
    if (u < nln) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        av[u][e] -= mean;
        q += av[u][e] * av[u][e];
      }
    }
  q += __shfl_xor(q, 32);
  if (h == 0) red[1][wave][l31] = q;
  __syncthreads();
  const float var = ((red[1][0][l31] + red[1][1][l31]) + (red[1][2][l31] + red[1][3][l31])) / (float)lnc;
  const float rstd = 1.0f / sqrtf(var + d.eps);

  // ---- normalise in place, write the rows out (block column 0), multiply ----
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int u = 0; u < LG_MAXG; ++u)
    if (u < nmine) {
      if (u < nln) {
#pragma unroll
        for (int e = 0; e < 4; ++e) av[u][e] = (av[u][e] * rstd * gm[u][e] + bt[u][e]) * rsl;
      }
      if (d.ln_out != nullptr && bn == 0 && m0 + l31 < d.M)
        *reinterpret_cast<float4*>(d.ln_out + (long)ra * d.ld_ln_out + 8 * (wave + 4 * u) + 4 * h) =
            make_float4(av[u][0], av[u][1], av[u][2], av[u][3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fd::mfma_32x32x2(av[u][e], bv[u][e], acc);
    }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
  __syncthreads();

  if (!erow_ok) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = 4 * ec4 + e;
    const int n = n0 + col;
    if (n >= d.N) continue;
    float x = (part[0][erow][col] + part[1][erow][col]) + (part[2][erow][col] + part[3][erow][col]);
    if (d.bias) x += e_bias[e];
    if (d.relu) x = x > 0.f ? x : 0.f;
    if (d.resid) x += e_res[e];
    d.out[(long)em * d.ldo + n] = x;
  }
}

}  // namespace

extern "C" int fd_ln_gemm(const FdLnGemmDesc* desc, void* stream) {
  FD_CHECK_ARG(desc != nullptr, "fd_ln_gemm: null descriptor");
  const FdLnGemmDesc& d = *desc;
  FD_CHECK_ARG(d.x && d.W && d.out && d.gamma && d.beta, "fd_ln_gemm: x / W / out / gamma / beta are required");
  FD_CHECK_ARG(d.M >= 0 && d.N > 0 && d.K > 0 && d.K % 8 == 0 && d.K <= 32 * LG_MAXG,
               "fd_ln_gemm: K=%d must be a multiple of 8, at most 320", d.K);
  FD_CHECK_ARG(d.ln_cols >= 0 && d.ln_cols <= d.K && d.ln_cols % 8 == 0, "fd_ln_gemm: ln_cols=%d must be a multiple of 8, at most K",
               d.ln_cols);
  FD_CHECK_ARG((d.ldx & 3) == 0 && (d.ldw & 3) == 0 && (d.ld_ln_out & 3) == 0 && fd_aligned16(d.x) && fd_aligned16(d.W) &&
                   fd_aligned16(d.gamma) && fd_aligned16(d.beta) && fd_aligned16(d.ln_out),
               "fd_ln_gemm: x / W / gamma / beta / ln_out must be 16-byte aligned with row strides that are multiples of 4");
  if (d.M == 0) return FD_OK;
  const int nblk_m = fd_cdiv(d.M, 32), nblk_n = fd_cdiv(d.N, 32);
  hipLaunchKernelGGL(ln_gemm_kernel, dim3(nblk_m * nblk_n), dim3(256), 0, (hipStream_t)stream, d, nblk_n);
  FD_CHECK_LAUNCH("fd_ln_gemm");
  return FD_OK;
}
