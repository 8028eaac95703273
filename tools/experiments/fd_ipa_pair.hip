// EXPERIMENT (not part of libfd_hip.so): the IPA pair pass in one kernel per direction.  Parity-tested at N <= 512 and SLOWER
// than the launch sequence it replaces (forward 325 vs 260 us, backward 758 vs 450 us per block at B=30 x N=128, round 2).
// Invariant Point Attention, the pair pass (model/ipa_pytorch.py:380-422 logits + softmax, :455-457 o_pair, and their
// autograd): everything that touches the pair tensor z of one query row (b, i) in ONE kernel per direction, so z is
// read once for BOTH linear_b (128 -> 8 bias) and down_z (128 -> 32 values) and the [P, 40] projections `zb` / `dzb`
// never exist in HBM.
//
//   forward  (block = one (b, i)):  zb[j, 0:40] = W40 z[i, j, :] + b40 into LDS (fp32 MFMA, exact);
//            logits[h, j] = qk[h, i, j] + sqrt(1/3) zb[j, h] - 1/2 gamma_h sum_p |q_p - k_p|^2 + 1e5 (m_i m_j - 1),
//            softmax over j (logits staged in LDS, wave reductions), probabilities back to HBM for the a.v GEMMs;
//            o_pair[h, c] = sum_j a[h, j] zb[j, 8 + c].
//   backward (persistent blocks walk the (b, i) rows): recomputes the down_z part of zb from z; dA += dout . pair_z;
//            softmax backward (dLogits over dA, d q-points, d head weights as ipa_softmax_bwd_kernel); dzb in LDS;
//            dz[i, j, :] (+)= dzb W40 (MFMA); dW40 += dzb^T z accumulated in MFMA registers over the block's rows and
//            flushed with one atomic pass per block; db40 likewise.
// W40 = [linear_b.weight (8) ; down_z.weight (32)] x 128, b40 likewise.  N <= 512.
#include "fd_common.h"
#include "fd_experiments.h"

namespace {

constexpr int H = 8, PQ = 8, CZ4 = 32, ZB = 40, CZ = 128;
constexpr int LDF = 2688, F_PAIR = 2432;
constexpr int CH = 64;        // pair rows (j) per staged chunk
constexpr int ZSP = 132;      // floats per staged z row (16-byte aligned rows, 2-way LDS conflicts at most)
constexpr int ZP = 41;        // floats per zb row in LDS
constexpr int DP = 48;        // floats per dzb row in LDS (columns 40..47 stay zero: MFMA row blocks of 16)

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// z[i, j0 .. j0+63, :] -> zs (rows past N zero-filled)
__device__ __forceinline__ void stage_z(float* __restrict__ zs, const float* __restrict__ zrow, int j0, int N, int tid) {
#pragma unroll
  for (int rep = 0; rep < 8; ++rep) {
    const int row = rep * 8 + (tid >> 5), c4 = tid & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 + row < N) v = *reinterpret_cast<const float4*>(zrow + (long)(j0 + row) * CZ + 4 * c4);
    *reinterpret_cast<float4*>(zs + row * ZSP + 4 * c4) = v;
  }
}

template <int NMAX>
__global__ __launch_bounds__(256) void ipa_pair_fwd_kernel(float* __restrict__ S, const float* __restrict__ z,
                                                           const float* __restrict__ W40, const float* __restrict__ b40,
                                                           const float* __restrict__ qp, const float* __restrict__ kp,
                                                           const float* __restrict__ head_w, const float* __restrict__ mask,
                                                           float* __restrict__ feats, int N) {
  __shared__ __attribute__((aligned(16))) float zs[CH * ZSP];
  __shared__ float zb_s[NMAX * ZP];
  __shared__ float lg[H * NMAX];
  const long bi = blockIdx.x;
  const int b = (int)(bi / N), i = (int)(bi % N);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;

  // W40 as MFMA B fragments (B[k][n] = W40[n][k]): lane (n = l & 15, kq) holds W40[16 nb + n][4 ks + kq]
  float wreg[3][32], bz[3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) {
    const int n = 16 * nb + l15;
    bz[nb] = n < ZB ? b40[n] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wreg[nb][ks] = n < ZB ? W40[n * CZ + 4 * ks + kq] : 0.f;
  }

  // ---- zb = W40 z + b40 for all j of this row, 64 rows per pass ----
  const float* zrow = z + bi * (long)N * CZ;
  for (int j0 = 0; j0 < N; j0 += CH) {
    __syncthreads();
    stage_z(zs, zrow, j0, N, tid);
    __syncthreads();
    f32x4 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nb][r] = bz[nb];
    const float* arow = zs + (16 * wave + l15) * ZSP + kq;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const float a = arow[4 * ks];
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) acc[nb] = fd::mfma_16x16x4(a, wreg[nb][ks], acc[nb]);
    }
    // D: lane (n = l & 15, rq = l >> 4) holds rows 4 rq + r
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
      const int n = 16 * nb + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + 16 * wave + 4 * kq + r;
        if (n < ZB && j < N) zb_s[j * ZP + n] = acc[nb][r];
      }
    }
  }
  __syncthreads();

  // ---- logits + softmax (two heads per wave), as ipa_softmax_fwd_kernel with the bias read from LDS ----
  const float mi = mask[bi];
  const float sq13 = sqrtf(1.0f / 3.0f);
  const float gscale = sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
  for (int hh = 0; hh < 2; ++hh) {
    const int h = wave * 2 + hh;
    const float gamma = softplus_f(head_w[h]) * gscale;
    float q[PQ * 3];
    const float* qsrc = qp + (bi * H + h) * (PQ * 3);
#pragma unroll
    for (int k = 0; k < PQ * 3; ++k) q[k] = qsrc[k];
    float* Srow = S + (((long)b * H + h) * N + i) * N;
    float* lrow = lg + h * NMAX;
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
      const long bj = (long)b * N + j;
      const float* ksrc = kp + (bj * H + h) * (PQ * 3);
      float pt = 0.f;
#pragma unroll
      for (int p = 0; p < PQ; ++p) {
        const float dx = q[3 * p] - ksrc[3 * p], dy = q[3 * p + 1] - ksrc[3 * p + 1], dz = q[3 * p + 2] - ksrc[3 * p + 2];
        pt += (dx * dx + dy * dy + dz * dz) * gamma;
      }
      float a = Srow[j] + sq13 * zb_s[j * ZP + h];
      a = a + pt * (-0.5f);
      a = a + 1e5f * (mi * mask[bj] - 1.f);
      lrow[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = fd::wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float e = expf(lrow[j] - mx);
      lrow[j] = e;
      sum += e;
    }
    sum = fd::wave_sum(sum);
    for (int j = lane; j < N; j += 64) {
      const float a = lrow[j] / sum;
      lrow[j] = a;
      Srow[j] = a;
    }
  }
  __syncthreads();

  // ---- o_pair[h, c] = sum_j a[h, j] zb[j, 8 + c] ----
  {
    const int h = tid / CZ4, c = tid % CZ4;
    const float* lrow = lg + h * NMAX;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc += lrow[j] * zb_s[j * ZP + H + c];
    feats[bi * LDF + F_PAIR + h * CZ4 + c] = acc;
  }
}

template <int NMAX>
__global__ __launch_bounds__(256) void ipa_pair_bwd_kernel(const float* __restrict__ A, float* __restrict__ dA,
                                                           const float* __restrict__ z, const float* __restrict__ W40,
                                                           const float* __restrict__ b40, const float* __restrict__ dfeats,
                                                           const float* __restrict__ qp, const float* __restrict__ kp,
                                                           const float* __restrict__ head_w, float* __restrict__ dz,
                                                           int dz_accumulate, float* __restrict__ dqp,
                                                           float* __restrict__ hw_part, float* __restrict__ dW40,
                                                           float* __restrict__ db40, int B, int N) {
  __shared__ __attribute__((aligned(16))) float zs[CH * ZSP];
  __shared__ float pz_s[NMAX * (CZ4 + 1)];   // down_z part of zb (pitch 33)
  __shared__ float dl_s[H * NMAX];
  __shared__ float Ai_s[H * NMAX];
  __shared__ float dzc[CH * DP];
  __shared__ float dout[H][CZ4];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  constexpr int PP = CZ4 + 1;

  // down_z rows of W40 as B fragments for the recomputation: lane (n, kq) holds W40[8 + 16 nb + n][4 ks + kq]
  float wpz[2][32], bpz[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = H + 16 * nb + l15;
    bpz[nb] = b40[n];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wpz[nb][ks] = W40[n * CZ + 4 * ks + kq];
  }
  // W40 as B fragments of dz = dzb W40 (B[k][n] = W40[k][n], k = zb column): lane (n, kq) holds W40[4 ks + kq][16 nb + n]
  float wt[8][10];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) wt[nb][ks] = W40[(4 * ks + kq) * CZ + 16 * nb + l15];
  // dW40 accumulators: wave w owns z-column blocks 2 w, 2 w + 1 for all three zb-row blocks
  f32x4 accw[3][2];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r) accw[mb][nn][r] = 0.f;
  float dbias = 0.f;   // thread m < 40: sum over rows of dzb[:, m]

  const float sq13 = sqrtf(1.0f / 3.0f);
  const float gscale = sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
  const long nrows = (long)B * N;
  for (long bi = blockIdx.x; bi < nrows; bi += gridDim.x) {
    const int b = (int)(bi / N), i = (int)(bi % N);
    const float* zrow = z + bi * (long)N * CZ;
    __syncthreads();
    dout[tid / CZ4][tid % CZ4] = dfeats[bi * LDF + F_PAIR + tid];

    // ---- pair_z = down_z(z) recomputed into LDS ----
    for (int j0 = 0; j0 < N; j0 += CH) {
      __syncthreads();
      stage_z(zs, zrow, j0, N, tid);
      __syncthreads();
      f32x4 acc[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nb][r] = bpz[nb];
      const float* arow = zs + (16 * wave + l15) * ZSP + kq;
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        const float a = arow[4 * ks];
        acc[0] = fd::mfma_16x16x4(a, wpz[0][ks], acc[0]);
        acc[1] = fd::mfma_16x16x4(a, wpz[1][ks], acc[1]);
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + 16 * wave + 4 * kq + r;
          if (j < N) pz_s[j * PP + 16 * nb + l15] = acc[nb][r];
        }
    }
    __syncthreads();

    // ---- softmax backward (two heads per wave): dA_total = dA + dout . pair_z ; dL = A (dA_total - sum_j A dA_total) ----
    for (int hh = 0; hh < 2; ++hh) {
      const int h = wave * 2 + hh;
      const float w = head_w[h];
      const float gamma = softplus_f(w) * gscale;
      const long rowoff = (((long)b * H + h) * N + i) * N;
      const float* Arow = A + rowoff;
      float* dArow = dA + rowoff;
      float* dlr = dl_s + h * NMAX;
      float* air = Ai_s + h * NMAX;
      float dot = 0.f;
      for (int j = lane; j < N; j += 64) {
        float t = dArow[j];
        const float* pz = pz_s + j * PP;
#pragma unroll
        for (int c = 0; c < CZ4; ++c) t += dout[h][c] * pz[c];
        const float a = Arow[j];
        air[j] = a;
        dlr[j] = t;
        dot += a * t;
      }
      dot = fd::wave_sum(dot);
      float q[PQ * 3], dq[PQ * 3];
      const float* qsrc = qp + (bi * H + h) * (PQ * 3);
#pragma unroll
      for (int k = 0; k < PQ * 3; ++k) { q[k] = qsrc[k]; dq[k] = 0.f; }
      float dgam = 0.f;
      for (int j = lane; j < N; j += 64) {
        const float dl = air[j] * (dlr[j] - dot);
        dArow[j] = dl;
        dlr[j] = dl;
        const float* ksrc = kp + (((long)b * N + j) * H + h) * (PQ * 3);
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < PQ * 3; ++k) {
          const float df = q[k] - ksrc[k];
          d2 += df * df;
          dq[k] -= gamma * dl * df;
        }
        dgam -= 0.5f * dl * d2;
      }
#pragma unroll
      for (int k = 0; k < PQ * 3; ++k) dq[k] = fd::wave_sum(dq[k]);
      dgam = fd::wave_sum(dgam);
      if (lane == 0) {
        float* dst = dqp + (bi * H + h) * (PQ * 3);
#pragma unroll
        for (int k = 0; k < PQ * 3; ++k) dst[k] = dq[k];
        const float sig = w > 20.f ? 1.f : 1.f / (1.f + expf(-w));
        hw_part[bi * H + h] = dgam * gscale * sig;     // column-summed by the host wrapper (no same-line atomics)
      }
    }

    // ---- per chunk: dzb in LDS, dW40 += dzb^T z, dz (+)= dzb W40 ----
    float* dzrow = dz + bi * (long)N * CZ;
    for (int j0 = 0; j0 < N; j0 += CH) {
      __syncthreads();
      stage_z(zs, zrow, j0, N, tid);
      for (int e = tid; e < CH * DP; e += 256) {
        const int row = e / DP, m = e % DP, j = j0 + row;
        float v = 0.f;
        if (j < N && m < ZB) {
          if (m < H) {
            v = sq13 * dl_s[m * NMAX + j];
          } else {
#pragma unroll
            for (int h = 0; h < H; ++h) v += Ai_s[h * NMAX + j] * dout[h][m - H];
          }
        }
        dzc[e] = v;
      }
      __syncthreads();
      if (tid < ZB) {
        float s = 0.f;
        for (int row = 0; row < CH; ++row) s += dzc[row * DP + tid];
        dbias += s;
      }
      // dW40[m][n] += sum_j dzb[j][m] z[j][n]: A[i = m][k = j], B[k = j][n]
#pragma unroll 4
      for (int ks = 0; ks < CH / 4; ++ks) {
        const int jr = 4 * ks + kq;
        const float b0 = zs[jr * ZSP + 32 * wave + l15], b1 = zs[jr * ZSP + 32 * wave + 16 + l15];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
          const float a = dzc[jr * DP + 16 * mb + l15];
          accw[mb][0] = fd::mfma_16x16x4(a, b0, accw[mb][0]);
          accw[mb][1] = fd::mfma_16x16x4(a, b1, accw[mb][1]);
        }
      }
      // dz rows 16 wave .. +15 of the chunk: A[i = j][k = zb column], B = wt
      f32x4 acc[8];
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) {
        const float a = dzc[(16 * wave + l15) * DP + 4 * ks + kq];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = fd::mfma_16x16x4(a, wt[nb][ks], acc[nb]);
      }
      __syncthreads();     // every wave is done with zs as an MFMA operand: it becomes the staging buffer of the dz tile
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) zs[(16 * wave + 4 * kq + r) * ZSP + 16 * nb + l15] = acc[nb][r];
      __syncthreads();
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        const int row = rep * 8 + (tid >> 5), c4 = tid & 31;
        if (j0 + row < N) {
          float4 v = *reinterpret_cast<const float4*>(zs + row * ZSP + 4 * c4);
          float* dst = dzrow + (long)(j0 + row) * CZ + 4 * c4;
          if (dz_accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(dst);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
    }
  }

  // ---- flush the weight-gradient accumulators: D lane (n = l & 15, rq) holds rows m = 16 mb + 4 rq + r ----
#pragma unroll
  for (int mb = 0; mb < 3; ++mb)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * mb + 4 * kq + r;
        if (m < ZB) atomicAdd(dW40 + m * CZ + 32 * wave + 16 * nn + l15, accw[mb][nn][r]);
      }
  if (tid < ZB) atomicAdd(db40 + tid, dbias);
}

}  // namespace

#define PAIR_DISPATCH(KERNEL, GRID, ...)                                                                     \
  do {                                                                                                       \
    if (N <= 128)                                                                                            \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<128>), GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (N <= 256)                                                                                       \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<256>), GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else                                                                                                     \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<512>), GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
  } while (0)

extern "C" int fd_ipa_pair_fwd(float* S, const float* z, const float* W40, const float* b40, const float* qp,
                               const float* kp, const float* head_w, const float* mask, float* feats, int B, int N,
                               void* stream) {
  FD_CHECK_ARG(N <= 512, "fd_ipa_pair_fwd: N=%d exceeds 512", N);
  FD_CHECK_ARG(fd_aligned16(z), "fd_ipa_pair_fwd: z must be 16-byte aligned");
  if (B == 0 || N == 0) return FD_OK;
  PAIR_DISPATCH(ipa_pair_fwd_kernel, dim3((unsigned)((long)B * N)), S, z, W40, b40, qp, kp, head_w, mask, feats, N);
  FD_CHECK_LAUNCH("fd_ipa_pair_fwd");
  return FD_OK;
}

extern "C" int fd_ipa_kpts_bwd(const float* dL, const float* qp, const float* kp, const float* head_w, float* dkp, int B,
                               int N, void* stream);

extern "C" int fd_ipa_pair_bwd(const float* A, float* dA, const float* z, const float* W40, const float* b40,
                               const float* dfeats, const float* qp, const float* kp, const float* head_w, float* dz,
                               int dz_accumulate, float* dqp, float* dkp, float* dhead_w, float* hw_part, float* dW40,
                               float* db40, int B, int N, void* stream) {
  FD_CHECK_ARG(N <= 512, "fd_ipa_pair_bwd: N=%d exceeds 512", N);
  FD_CHECK_ARG(fd_aligned16(z) && fd_aligned16(dz), "fd_ipa_pair_bwd: z / dz must be 16-byte aligned");
  if (B == 0 || N == 0) return FD_OK;
  const long nrows = (long)B * N;
  // persistent blocks: the dW40 accumulators live in registers across a block's rows; one atomic flush per block
  const unsigned grid = (unsigned)(nrows < 512 ? nrows : 512);
  PAIR_DISPATCH(ipa_pair_bwd_kernel, dim3(grid), A, dA, z, W40, b40, dfeats, qp, kp, head_w, dz, dz_accumulate, dqp,
                hw_part, dW40, db40, B, N);
  FD_CHECK_LAUNCH("fd_ipa_pair_bwd");
  {
    int rc = fd_colsum_acc(hw_part, H, nrows, H, dhead_w, stream);
    if (rc != FD_OK) return rc;
  }
  return fd_ipa_kpts_bwd(dA, qp, kp, head_w, dkp, B, N, stream);
}
