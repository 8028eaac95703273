// Invariant Point Attention: the non-GEMM pieces (HBM / VALU bound), forward and backward.
//
// Reference: model/ipa_pytorch.py:303-471 (InvariantPointAttention.forward) with the frame
// algebra of openfold/utils/rigid_utils.py (:82-106 rot_vec_mul, :173-205 quat_to_rot,
// :1104-1130 Rigid.apply / invert_apply).  Dense contractions (q k^T, a v, a v_pts,
// linear_b/down_z, linear_out) run on fd_gemm; this file holds
//   points   : raw Linear outputs ([x|y|z] blocks, :351-352,:364-374) -> global-frame q/k/v points
//   softmax  : logits = qk*sqrt(1/3C) + sqrt(1/3) b - 0.5 gamma_h sum_p |q_p - k_p|^2 + 1e5 (m_i m_j - 1)
//              (:380-422), softmax over j
//   opt      : o_pt = R^T (sum_j a v_pts - t), |o_pt| (:432-449)
//   opair    : o_pair = sum_j a_ij (down_z z)_ij (:455-457)
// Layouts (fp32): proj [R,6816] = [q 2048 | kv 8x(256 k,256 v) | qp raw 3x64 | kvp raw 3x160];
// qp,kp [R,8,8,3]; vp [R,8,12,3]; S/A [B,8,N,N]; zb [P,40] = [linear_b 8 | down_z 32];
// feats [R,2688] = [o 2048 | o_pt.x 96 | .y 96 | .z 96 | |o_pt| 96 | o_pair 256];
// dframe [R,12] = dL/dR (row-major 3x3) followed by dL/dt, accumulated (+=).
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int H = 8, C = 256, PQ = 8, PV = 12, CZ4 = 32, ZB = 40;
constexpr int LDP = H * C * 3 + H * PQ * 3 + H * (PQ + PV) * 3;  // 6816
constexpr int QP_OFF = H * C * 3;                                 // 6144
constexpr int KVP_OFF = QP_OFF + H * PQ * 3;                      // 6336
constexpr int NQP = H * PQ;                                       // 64
constexpr int NKVP = H * (PQ + PV);                               // 160
constexpr int LDF = H * (C + 4 * PV + CZ4);                       // 2688
constexpr int F_PT = H * C;                                       // 2048
constexpr int F_NORM = F_PT + 3 * H * PV;                         // 2336
constexpr int F_PAIR = F_NORM + H * PV;                           // 2432
constexpr int MAXN = 1024;

struct Rot { float r[9]; };

__device__ __forceinline__ Rot quat_to_rot(const float* __restrict__ q) {
  const float a = q[0], b = q[1], c = q[2], d = q[3];
  Rot R;
  R.r[0] = a * a + b * b - c * c - d * d;
  R.r[1] = 2.f * (b * c - a * d);
  R.r[2] = 2.f * (b * d + a * c);
  R.r[3] = 2.f * (b * c + a * d);
  R.r[4] = a * a - b * b + c * c - d * d;
  R.r[5] = 2.f * (c * d - a * b);
  R.r[6] = 2.f * (b * d - a * c);
  R.r[7] = 2.f * (c * d + a * b);
  R.r[8] = a * a - b * b - c * c + d * d;
  return R;
}

// ---------------------------------------------------------------- points
__global__ __launch_bounds__(256) void ipa_points_fwd_kernel(const float* __restrict__ proj,
                                                             const float* __restrict__ quat,
                                                             const float* __restrict__ trans,
                                                             float* __restrict__ qp, float* __restrict__ kp,
                                                             float* __restrict__ vp, float* __restrict__ kp_soa, int N,
                                                             long R_) {
  // kp_soa != nullptr: a second copy of the key points as [B, H, 24, N] -- the attention kernels walk the keys with one
  // lane per j, and 24 floats at a 768-byte lane stride cost them half their time (64 lines per 16-byte load)
  const long total = R_ * (NQP + NKVP);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / (NQP + NKVP);
    const int pi = (int)(e % (NQP + NKVP));
    const Rot R = quat_to_rot(quat + r * 4);
    const float* pr = proj + r * LDP;
    float x, y, z;
    float* dst;
    int soa_h = 0, soa_p = -1;
    if (pi < NQP) {
      x = pr[QP_OFF + pi]; y = pr[QP_OFF + NQP + pi]; z = pr[QP_OFF + 2 * NQP + pi];
      dst = qp + (r * NQP + pi) * 3;
    } else {
      const int pk = pi - NQP;
      x = pr[KVP_OFF + pk]; y = pr[KVP_OFF + NKVP + pk]; z = pr[KVP_OFF + 2 * NKVP + pk];
      const int h = pk / (PQ + PV), p = pk % (PQ + PV);
      dst = p < PQ ? kp + ((r * H + h) * PQ + p) * 3 : vp + ((r * H + h) * PV + (p - PQ)) * 3;
      if (p < PQ) { soa_h = h; soa_p = p; }
    }
    const float* t = trans + r * 3;
    const float g0 = R.r[0] * x + R.r[1] * y + R.r[2] * z + t[0];
    const float g1 = R.r[3] * x + R.r[4] * y + R.r[5] * z + t[1];
    const float g2 = R.r[6] * x + R.r[7] * y + R.r[8] * z + t[2];
    dst[0] = g0; dst[1] = g1; dst[2] = g2;
    if (kp_soa != nullptr && soa_p >= 0) {
      const long b = r / N, j = r % N;
      float* d = kp_soa + ((b * H + soa_h) * (PQ * 3) + 3 * soa_p) * N + j;
      d[0] = g0; d[N] = g1; d[2 * (long)N] = g2;
    }
  }
}

// block per residue: d raw = R^T d glob ; dR[a][c] += dg[a]*raw[c] ; dt += dg
__global__ __launch_bounds__(256) void ipa_points_bwd_kernel(const float* __restrict__ proj,
                                                             const float* __restrict__ quat,
                                                             const float* __restrict__ dqp,
                                                             const float* __restrict__ dkp,
                                                             const float* __restrict__ dvp,
                                                             float* __restrict__ dproj, float* __restrict__ dframe) {
  __shared__ float red[4][12];
  const long r = blockIdx.x;
  const int pi = (int)threadIdx.x;
  const Rot R = quat_to_rot(quat + r * 4);
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  if (pi < NQP + NKVP) {
    const float* pr = proj + r * LDP;
    float* dpr = dproj + r * LDP;
    float x, y, z;
    const float* dg;
    int cx, cy, cz;
    if (pi < NQP) {
      cx = QP_OFF + pi; cy = cx + NQP; cz = cy + NQP;
      dg = dqp + (r * NQP + pi) * 3;
    } else {
      const int pk = pi - NQP;
      cx = KVP_OFF + pk; cy = cx + NKVP; cz = cy + NKVP;
      const int h = pk / (PQ + PV), p = pk % (PQ + PV);
      dg = p < PQ ? dkp + ((r * H + h) * PQ + p) * 3 : dvp + ((r * H + h) * PV + (p - PQ)) * 3;
    }
    x = pr[cx]; y = pr[cy]; z = pr[cz];
    const float g0 = dg[0], g1 = dg[1], g2 = dg[2];
    dpr[cx] = R.r[0] * g0 + R.r[3] * g1 + R.r[6] * g2;
    dpr[cy] = R.r[1] * g0 + R.r[4] * g1 + R.r[7] * g2;
    dpr[cz] = R.r[2] * g0 + R.r[5] * g1 + R.r[8] * g2;
    acc[0] = g0 * x; acc[1] = g0 * y; acc[2] = g0 * z;
    acc[3] = g1 * x; acc[4] = g1 * y; acc[5] = g1 * z;
    acc[6] = g2 * x; acc[7] = g2 * y; acc[8] = g2 * z;
    acc[9] = g0; acc[10] = g1; acc[11] = g2;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = fd::wave_sum(acc[k]);
  const int lane = fd::lane_id(), wave = fd::wave_id();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
  __syncthreads();
  if (pi < 12) dframe[r * 12 + pi] += red[0][pi] + red[1][pi] + red[2][pi] + red[3][pi];
}

// ---------------------------------------------------------------- softmax
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// the 8 query / key points of one (residue, head): 24 contiguous floats at a 96-byte stride -> six 16-byte loads
__device__ __forceinline__ void load_pts(float (&v)[PQ * 3], const float* __restrict__ src) {
#pragma unroll
  for (int i = 0; i < PQ * 3 / 4; ++i) {
    const float4 t = *reinterpret_cast<const float4*>(src + 4 * i);
    v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
  }
}

// the same 24 floats of key j from the [B, H, 24, N] copy: 24 loads, each one contiguous run over the lanes of a wave
__device__ __forceinline__ void load_pts_soa(float (&v)[PQ * 3], const float* __restrict__ src, int N) {
#pragma unroll
  for (int k = 0; k < PQ * 3; ++k) v[k] = src[(long)k * N];
}

// SLAB (N <= NMAX = 256): the row's [N, 40] block of zb -- one contiguous 160 N bytes -- is copied to LDS with whole-line
// float4 loads and both the bias column of the logits and the 32 o_pair columns are served from there; the direct reads
// (4 bytes at a 160-byte stride for the bias, 128 of every 160 bytes for o_pair) fetched about twice the bytes they used.
constexpr long FD_IPA_ATTN_WIDE_MAX_ROWS = 512;     // fd_ipa_attn_fwd: up to this many query rows on the 512-thread form
constexpr int ZPAD = ZB + 1;    // floats per zb row in LDS (41: conflict-free for lanes along j and along the columns)
// NT (round 5): 256 threads = four waves x two heads each, or 512 = one wave per head with the o_pair sums split in two halves of the
// keys -- for launches with few query rows (a lone backbone: 128 / 256 blocks on 256 CUs), where a row's serial chain is the launch
template <int NMAX, bool SLAB, int NT = 256>
__global__ __launch_bounds__(NT) void ipa_softmax_fwd_kernel(float* __restrict__ S, const float* __restrict__ zb,
                                                              const float* __restrict__ qp,
                                                              const float* __restrict__ kp,
                                                              const float* __restrict__ kp_soa,
                                                              const float* __restrict__ head_w,
                                                              const float* __restrict__ mask, float* __restrict__ feats,
                                                              int N) {
  // feats != nullptr: o_pair of the same (b, i) row is taken from the probabilities while they are in LDS
  // (fd_ipa_attn_fwd: one launch and one pass over A less than softmax + opair)
  constexpr int HPW = H / (NT / 64);      // heads per wave
  __shared__ float lg[H][NMAX];
  __shared__ float zs[SLAB ? NMAX * ZPAD : 1];
  __shared__ float opart[NT > 256 ? 256 : 1];
  const long bi = blockIdx.x;
  const int b = (int)(bi / N), i = (int)(bi % N);
  const int lane = fd::lane_id(), wave = fd::wave_id();
  if (SLAB) {
    const float4* src = reinterpret_cast<const float4*>(zb + bi * N * ZB);     // (N * 40 floats: a multiple of 4, 16-byte aligned)
    for (int e = (int)threadIdx.x; e < N * (ZB / 4); e += NT) {
      const float4 v = src[e];
      const int j = e / (ZB / 4), c = 4 * (e % (ZB / 4));
      float* d = zs + j * ZPAD + c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
  }
  const float mi = mask[bi];
  const float sq13 = sqrtf(1.0f / 3.0f);
  const float gscale = sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
  for (int hh = 0; hh < HPW; ++hh) {
    const int h = wave * HPW + hh;
    const float gamma = softplus_f(head_w[h]) * gscale;
    float q[PQ * 3];
    const float* qsrc = qp + (bi * H + h) * (PQ * 3);
#pragma unroll
    for (int k = 0; k < PQ * 3; ++k) q[k] = qsrc[k];
    float* Srow = S + (((long)b * H + h) * N + i) * N;
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
      const long bj = (long)b * N + j;
      float ksrc[PQ * 3];
      if (kp_soa != nullptr)
        load_pts_soa(ksrc, kp_soa + ((long)b * H + h) * (PQ * 3) * N + j, N);
      else
        load_pts(ksrc, kp + (bj * H + h) * (PQ * 3));
      float pt = 0.f;
#pragma unroll
      for (int p = 0; p < PQ; ++p) {
        const float dx = q[3 * p] - ksrc[3 * p], dy = q[3 * p + 1] - ksrc[3 * p + 1], dz = q[3 * p + 2] - ksrc[3 * p + 2];
        pt += (dx * dx + dy * dy + dz * dz) * gamma;
      }
      float a = Srow[j] + sq13 * (SLAB ? zs[j * ZPAD + h] : zb[(bi * N + j) * ZB + h]);
      a = a + pt * (-0.5f);
      a = a + 1e5f * (mi * mask[bj] - 1.f);
      lg[h][j] = a;
      mx = fmaxf(mx, a);
    }
    mx = fd::wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float e = expf(lg[h][j] - mx);
      lg[h][j] = e;
      sum += e;
    }
    sum = fd::wave_sum(sum);
    for (int j = lane; j < N; j += 64) {
      const float a = lg[h][j] / sum;
      Srow[j] = a;
      lg[h][j] = a;
    }
  }
  if (feats != nullptr) {
    __syncthreads();
    const int t = (int)threadIdx.x & 255, half = (int)threadIdx.x >> 8;
    const int h = t / CZ4, c = t % CZ4;
    // (NT = 512: the upper half of the block takes the second half of the keys)
    const int j0 = NT > 256 ? half * ((N + 1) / 2) : 0;
    const int j1 = NT > 256 ? (half == 0 ? (N + 1) / 2 : N) : N;
    float acc = 0.f;
    if (SLAB) {
      const float* z = zs + H + c;
      for (int j = j0; j < j1; ++j) acc += lg[h][j] * z[j * ZPAD];
    } else {
      const float* z = zb + bi * N * ZB + H + c;
      for (int j = j0; j < j1; ++j) acc += lg[h][j] * z[(long)j * ZB];
    }
    if (NT > 256) {
      if (half == 1) opart[t] = acc;
      __syncthreads();
      if (half == 0) feats[bi * LDF + F_PAIR + h * CZ4 + c] = acc + opart[t];
    } else {
      feats[bi * LDF + F_PAIR + h * CZ4 + c] = acc;
    }
  }
}

// dL = A * (dA - sum_j A dA) written over dA; d(zb bias) = sqrt(1/3) dL; dqp_i; d head_w
// FUSED (fd_ipa_attn_bwd): the o_pair backward of the same (b, i) row first -- dzb[:, 8:40] = sum_h A dout and
// dA += dout . pair_z -- with A and the updated dA held in LDS (one launch, one pass over A and one over dA less).
template <bool FUSED, int NMAX, bool SLAB = false>
__global__ __launch_bounds__(256) void ipa_softmax_bwd_kernel(const float* __restrict__ A, float* __restrict__ dA,
                                                              const float* __restrict__ qp,
                                                              const float* __restrict__ kp,
                                                              const float* __restrict__ kp_soa,
                                                              const float* __restrict__ head_w,
                                                              float* __restrict__ dzb, float* __restrict__ dqp,
                                                              float* __restrict__ hw_part, const float* __restrict__ zb,
                                                              const float* __restrict__ dfeats, int N) {
  // (NMAX = 256 for N <= 256: 17 KB of LDS instead of 66 KB, three blocks per CU instead of two)
  __shared__ float dl_s[H][NMAX];
  __shared__ float Ai[FUSED ? H : 1][FUSED ? NMAX : 1];
  __shared__ float dout[H][CZ4];
  // SLAB (FUSED, N <= NMAX): the row's [N, 40] block of zb comes in, and its block of dzb goes out, as ONE contiguous
  // run of whole lines through this LDS image (pitch 41 floats); zb's o_pair columns are consumed before dzb overwrites them
  __shared__ float zs[SLAB ? NMAX * ZPAD : 1];
  const long bi = blockIdx.x;
  const int b = (int)(bi / N), i = (int)(bi % N);
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const float sq13 = sqrtf(1.0f / 3.0f);
  const float gscale = sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
  if (FUSED) {
    // (thread -> head tid >> 5, keys tid & 31 + 32 k: no division by the run-time N in the element loops)
    const int eh = (int)threadIdx.x >> 5, ej = (int)threadIdx.x & 31;
    for (int j = ej; j < N; j += 32) Ai[eh][j] = A[(((long)b * H + eh) * N + i) * N + j];
    dout[threadIdx.x / CZ4][threadIdx.x % CZ4] = dfeats[bi * LDF + F_PAIR + threadIdx.x];
    if (SLAB) {
      const float4* src = reinterpret_cast<const float4*>(zb + bi * N * ZB);
      for (int e = (int)threadIdx.x; e < N * (ZB / 4); e += 256) {
        const float4 v = src[e];
        float* d = zs + (e / (ZB / 4)) * ZPAD + 4 * (e % (ZB / 4));
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    __syncthreads();
    // dA[b,h,i,j] + sum_c dout[h][c] * pair_z[b,i,j,c]
    for (int j = ej; j < N; j += 32) {
      const float* z = SLAB ? zs + j * ZPAD + H : zb + (bi * N + j) * ZB + H;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < CZ4; ++c) acc += dout[eh][c] * z[c];
      dl_s[eh][j] = dA[(((long)b * H + eh) * N + i) * N + j] + acc;
    }
    if (SLAB) __syncthreads();   // every read of the zb image is done before dzb goes into it
    // d pair_z[b,i,j,c] = sum_h A[h][j] * dout[h][c]
    for (int e = (int)threadIdx.x; e < N * CZ4; e += 256) {
      const int j = e / CZ4, c = e % CZ4;
      float acc = 0.f;
#pragma unroll
      for (int h = 0; h < H; ++h) acc += Ai[h][j] * dout[h][c];
      if (SLAB)
        zs[j * ZPAD + H + c] = acc;
      else
        dzb[(bi * N + j) * ZB + H + c] = acc;
    }
    __syncthreads();
  }
  for (int hh = 0; hh < 2; ++hh) {
    const int h = wave * 2 + hh;
    const float w = head_w[h];
    const float gamma = softplus_f(w) * gscale;
    const long rowoff = (((long)b * H + h) * N + i) * N;
    const float* Arow = FUSED ? &Ai[h][0] : A + rowoff;
    float* dArow = dA + rowoff;
    const float* dAin = FUSED ? &dl_s[h][0] : dArow;
    float dot = 0.f;
    for (int j = lane; j < N; j += 64) dot += Arow[j] * dAin[j];
    dot = fd::wave_sum(dot);
    float q[PQ * 3], dq[PQ * 3];
    const float* qsrc = qp + (bi * H + h) * (PQ * 3);
#pragma unroll
    for (int k = 0; k < PQ * 3; ++k) { q[k] = qsrc[k]; dq[k] = 0.f; }
    float dgam = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float dl = Arow[j] * (dAin[j] - dot);
      dArow[j] = dl;
      dl_s[h][j] = dl;
      float ksrc[PQ * 3];
      if (kp_soa != nullptr)
        load_pts_soa(ksrc, kp_soa + ((long)b * H + h) * (PQ * 3) * N + j, N);
      else
        load_pts(ksrc, kp + (((long)b * N + j) * H + h) * (PQ * 3));
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
      // gamma = softplus(w) * gscale ; d softplus = sigmoid
      const float sig = w > 20.f ? 1.f : 1.f / (1.f + expf(-w));
      // per-(b,i) partial: B*N blocks x 8 single-lane atomics on ONE cache line serialise at ~10 ns each (the
      // kernel ran 377 us with them, 78 us without); fd_ipa_softmax_bwd column-sums the partials instead
      hw_part[bi * H + h] = dgam * gscale * sig;
    }
  }
  __syncthreads();
  for (int e = (int)threadIdx.x; e < N * H; e += 256) {
    const int j = e / H, h = e % H;
    if (SLAB)
      zs[j * ZPAD + h] = sq13 * dl_s[h][j];
    else
      dzb[(bi * N + j) * ZB + h] = sq13 * dl_s[h][j];
  }
  if (SLAB) {
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(dzb + bi * N * ZB);
    for (int e = (int)threadIdx.x; e < N * (ZB / 4); e += 256) {
      const float* d = zs + (e / (ZB / 4)) * ZPAD + 4 * (e % (ZB / 4));
      dst[e] = make_float4(d[0], d[1], d[2], d[3]);
    }
  }
}

// dkp[b,j,h,:] = gamma_h * sum_i dL[b,h,i,j] * (qp[b,i,h,:] - kp[b,j,h,:]); block = (j-tile of 64, h, b)
__global__ __launch_bounds__(256) void ipa_kpts_bwd_kernel(const float* __restrict__ dL, const float* __restrict__ qp,
                                                           const float* __restrict__ kp,
                                                           const float* __restrict__ head_w, float* __restrict__ dkp,
                                                           int N) {
  __shared__ float red[4][64][PQ * 3 + 1];
  const int jt = (int)blockIdx.x, h = (int)blockIdx.y, b = (int)blockIdx.z;
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const int j = jt * 64 + lane;
  const float gscale = sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
  const float gamma = softplus_f(head_w[h]) * gscale;
  float k[PQ * 3], acc[PQ * 3];
#pragma unroll
  for (int c = 0; c < PQ * 3; ++c) { k[c] = 0.f; acc[c] = 0.f; }
  if (j < N) {
    const float* ksrc = kp + (((long)b * N + j) * H + h) * (PQ * 3);
#pragma unroll
    for (int c = 0; c < PQ * 3; ++c) k[c] = ksrc[c];
  }
  for (int i = wave; i < N; i += 4) {
    const float dl = j < N ? dL[(((long)b * H + h) * N + i) * N + j] : 0.f;
    const float* qsrc = qp + (((long)b * N + i) * H + h) * (PQ * 3);
#pragma unroll
    for (int c = 0; c < PQ * 3; ++c) acc[c] += dl * (qsrc[c] - k[c]);
  }
#pragma unroll
  for (int c = 0; c < PQ * 3; ++c) red[wave][lane][c] = acc[c];
  __syncthreads();
  if (wave == 0 && j < N) {
    float* dst = dkp + (((long)b * N + j) * H + h) * (PQ * 3);
#pragma unroll
    for (int c = 0; c < PQ * 3; ++c)
      dst[c] = gamma * (red[0][lane][c] + red[1][lane][c] + red[2][lane][c] + red[3][lane][c]);
  }
}

// ---------------------------------------------------------------- o_pt
__global__ __launch_bounds__(256) void ipa_opt_fwd_kernel(const float* __restrict__ optg,
                                                          const float* __restrict__ quat,
                                                          const float* __restrict__ trans, float* __restrict__ feats,
                                                          long R_) {
  const long total = R_ * H * PV;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / (H * PV);
    const int hp = (int)(e % (H * PV));
    const Rot R = quat_to_rot(quat + r * 4);
    const float* t = trans + r * 3;
    const float* g = optg + e * 3;
    const float ux = g[0] - t[0], uy = g[1] - t[1], uz = g[2] - t[2];
    const float lx = R.r[0] * ux + R.r[3] * uy + R.r[6] * uz;
    const float ly = R.r[1] * ux + R.r[4] * uy + R.r[7] * uz;
    const float lz = R.r[2] * ux + R.r[5] * uy + R.r[8] * uz;
    float* f = feats + r * LDF;
    f[F_PT + hp] = lx;
    f[F_PT + H * PV + hp] = ly;
    f[F_PT + 2 * H * PV + hp] = lz;
    f[F_NORM + hp] = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
  }
}

// block per residue (128 threads, 96 active)
// ptdot (may be null; needs trans): [R, 8] = sum_p d(o_pt global) . (o_pt global) per head -- the o_pt term of the softmax
// backward's row constant sum_j a_ij dP_ij (fd_ipa_flash_bwd takes it from here instead of a pass over the keys)
__global__ __launch_bounds__(128) void ipa_opt_bwd_kernel(const float* __restrict__ dfeats,
                                                          const float* __restrict__ feats,
                                                          const float* __restrict__ quat,
                                                          float* __restrict__ doptg, float* __restrict__ dframe,
                                                          const float* __restrict__ trans, float* __restrict__ ptdot) {
  __shared__ float red[2][12];
  __shared__ float dots[H * PV];
  const long r = blockIdx.x;
  const int hp = (int)threadIdx.x;
  const Rot R = quat_to_rot(quat + r * 4);
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  if (hp < H * PV) {
    const float* f = feats + r * LDF;
    const float* df = dfeats + r * LDF;
    const float lx = f[F_PT + hp], ly = f[F_PT + H * PV + hp], lz = f[F_PT + 2 * H * PV + hp];
    const float nrm = f[F_NORM + hp];
    const float dn = df[F_NORM + hp] / nrm;
    const float dlx = df[F_PT + hp] + dn * lx;
    const float dly = df[F_PT + H * PV + hp] + dn * ly;
    const float dlz = df[F_PT + 2 * H * PV + hp] + dn * lz;
    // o_local = R^T u  =>  du = R dl ; dR[a][c] += u[a] dl[c] ; dt = -du
    const float dux = R.r[0] * dlx + R.r[1] * dly + R.r[2] * dlz;
    const float duy = R.r[3] * dlx + R.r[4] * dly + R.r[5] * dlz;
    const float duz = R.r[6] * dlx + R.r[7] * dly + R.r[8] * dlz;
    const float ux = R.r[0] * lx + R.r[1] * ly + R.r[2] * lz;
    const float uy = R.r[3] * lx + R.r[4] * ly + R.r[5] * lz;
    const float uz = R.r[6] * lx + R.r[7] * ly + R.r[8] * lz;
    float* dg = doptg + (r * H * PV + hp) * 3;
    dg[0] = dux; dg[1] = duy; dg[2] = duz;
    if (ptdot != nullptr) {
      const float* t = trans + r * 3;
      dots[hp] = dux * (ux + t[0]) + duy * (uy + t[1]) + duz * (uz + t[2]);
    }
    acc[0] = ux * dlx; acc[1] = ux * dly; acc[2] = ux * dlz;
    acc[3] = uy * dlx; acc[4] = uy * dly; acc[5] = uy * dlz;
    acc[6] = uz * dlx; acc[7] = uz * dly; acc[8] = uz * dlz;
    acc[9] = -dux; acc[10] = -duy; acc[11] = -duz;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = fd::wave_sum(acc[k]);
  const int lane = fd::lane_id(), wave = fd::wave_id();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
  __syncthreads();
  if (hp < 12) dframe[r * 12 + hp] += red[0][hp] + red[1][hp];
  if (ptdot != nullptr && hp < H) {
    float d = 0.f;
#pragma unroll
    for (int p = 0; p < PV; ++p) d += dots[hp * PV + p];
    ptdot[r * H + hp] = d;
  }
}

// ---------------------------------------------------------------- o_pair
__global__ __launch_bounds__(256) void ipa_opair_fwd_kernel(const float* __restrict__ A, const float* __restrict__ zb,
                                                            float* __restrict__ feats, int N) {
  __shared__ float Ai[H][MAXN];
  const long bi = blockIdx.x;
  const int b = (int)(bi / N), i = (int)(bi % N);
  for (int e = (int)threadIdx.x; e < H * N; e += 256) {
    const int h = e / N, j = e % N;
    Ai[h][j] = A[(((long)b * H + h) * N + i) * N + j];
  }
  __syncthreads();
  const int h = (int)threadIdx.x / CZ4, c = (int)threadIdx.x % CZ4;
  const float* z = zb + bi * N * ZB + H + c;
  float acc = 0.f;
  for (int j = 0; j < N; ++j) acc += Ai[h][j] * z[(long)j * ZB];
  feats[bi * LDF + F_PAIR + h * CZ4 + c] = acc;
}

__global__ __launch_bounds__(256) void ipa_opair_bwd_kernel(const float* __restrict__ A, const float* __restrict__ zb,
                                                            const float* __restrict__ dfeats,
                                                            float* __restrict__ dA, float* __restrict__ dzb, int N) {
  __shared__ float Ai[H][MAXN];
  __shared__ float dout[H][CZ4];
  const long bi = blockIdx.x;
  const int b = (int)(bi / N), i = (int)(bi % N);
  for (int e = (int)threadIdx.x; e < H * N; e += 256) {
    const int h = e / N, j = e % N;
    Ai[h][j] = A[(((long)b * H + h) * N + i) * N + j];
  }
  dout[threadIdx.x / CZ4][threadIdx.x % CZ4] = dfeats[bi * LDF + F_PAIR + threadIdx.x];
  __syncthreads();
  // d pair_z[b,i,j,c] = sum_h A[h][j] * dout[h][c]
  for (int e = (int)threadIdx.x; e < N * CZ4; e += 256) {
    const int j = e / CZ4, c = e % CZ4;
    float acc = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) acc += Ai[h][j] * dout[h][c];
    dzb[(bi * N + j) * ZB + H + c] = acc;
  }
  // dA[b,h,i,j] += sum_c dout[h][c] * pair_z[b,i,j,c]
  for (int e = (int)threadIdx.x; e < H * N; e += 256) {
    const int h = e / N, j = e % N;
    const float* z = zb + (bi * N + j) * ZB + H;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CZ4; ++c) acc += dout[h][c] * z[c];
    dA[(((long)b * H + h) * N + i) * N + j] += acc;
  }
}

// ---------------------------------------------------------------- dz (+)= dzb W40
// The pair-bias / down_z part of IPA's gradient w.r.t. the pair tensor: dz[p, :] (+)= dzb[p, 0:40] W40[0:40, 0:128] over
// the B*N*N pair rows (autograd of linear_b and down_z, ipa_pytorch.py:380-386,455-457).  583 MB of traffic for 5 GFLOP:
// a streaming kernel.  Every wave owns 32-row tiles; W40 (20 KB) lives in its registers as MFMA B fragments for the whole
// launch, the dzb tile goes global -> registers directly in MFMA A layout (as fd_gemm tile 5), v_mfma_f32_32x32x2_f32
// (exact fp32), and the 32 x 128 result is added to dz with 128-byte row segments per lane half.  No LDS, no barrier.
__global__ __launch_bounds__(256) void ipa_dz_acc_kernel(const float* __restrict__ dzb, const float* __restrict__ W40,
                                                         float* __restrict__ dz, long rows, int accumulate) {
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const int h = lane >> 5, l31 = lane & 31;
  // B fragments: W40[k = 8 g + 4 h + t][n = 32 nt + l31]
  float w[ZB / 8][4][4];
#pragma unroll
  for (int g = 0; g < ZB / 8; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) w[g][t][nt] = W40[(8 * g + 4 * h + t) * 128 + 32 * nt + l31];
  const long ntiles = (rows + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long r0 = tile * 32;
    const long ra = (r0 + l31 < rows) ? r0 + l31 : rows - 1;
    float4 a[ZB / 8];
#pragma unroll
    for (int g = 0; g < ZB / 8; ++g) a[g] = *reinterpret_cast<const float4*>(dzb + ra * ZB + 8 * g + 4 * h);
    // the accumulators start from the old values of dz (fetched beside the dzb tile, under the MFMAs of the previous
    // tile's tail): the read-modify-write costs no dependent load -> add -> store chain per element
    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float* o = dz + (row < rows ? row : rows - 1) * 128 + l31;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt][r] = accumulate ? o[32 * nt] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < ZB / 8; ++g) {
      const float av[4] = {a[g].x, a[g].y, a[g].z, a[g].w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = fd::mfma_32x32x2(av[t], w[g][t][nt], acc[nt]);
    }
    // D: reg r -> row (r & 3) + 8 (r >> 2) + 4 h, column l31 of n-tile nt
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < rows) {
        float* o = dz + row * 128 + l31;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[32 * nt] = acc[nt][r];
      }
    }
  }
}

}  // namespace

extern "C" int fd_ipa_dz_acc(const float* dzb, const float* W40, float* dz, long rows, int accumulate, void* stream) {
  FD_CHECK_ARG(dzb && W40 && dz, "fd_ipa_dz_acc: null operand");
  FD_CHECK_ARG(fd_aligned16(dzb), "fd_ipa_dz_acc: dzb must be 16-byte aligned");
  if (rows == 0) return FD_OK;
  const long tiles = (rows + 31) / 32;
  long g = (tiles + 3) / 4;
  if (g > 2048) g = 2048;                      // 8 blocks per CU: the waves walk their tiles, W40 stays in registers
  hipLaunchKernelGGL(ipa_dz_acc_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dzb, W40, dz, rows, accumulate);
  FD_CHECK_LAUNCH("fd_ipa_dz_acc");
  return FD_OK;
}

#define CHECK_DIMS(fn)                                                                                         \
  FD_CHECK_ARG(nheads == H && c_hidden == C && n_qk == PQ && n_v == PV,                                         \
               fn ": built for no_heads=8 c_hidden=256 no_qk_points=8 no_v_points=12 (config/base.yaml), got " \
                  "%d/%d/%d/%d", nheads, c_hidden, n_qk, n_v)

static unsigned grid1d(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int fd_ipa_points_fwd(const float* proj, const float* quat, const float* trans, float* qp, float* kp,
                                 float* vp, float* kp_soa, int n_res, long R_, int nheads, int c_hidden, int n_qk,
                                 int n_v, void* stream) {
  CHECK_DIMS("fd_ipa_points_fwd");
  if (R_ == 0) return FD_OK;
  FD_CHECK_ARG(kp_soa == nullptr || (n_res > 0 && R_ % n_res == 0),
               "fd_ipa_points_fwd: kp_soa needs the residue count per backbone (R=%ld, n_res=%d)", R_, n_res);
  hipLaunchKernelGGL(ipa_points_fwd_kernel, dim3(grid1d(R_ * (NQP + NKVP))), dim3(256), 0, (hipStream_t)stream, proj,
                     quat, trans, qp, kp, vp, kp_soa, n_res > 0 ? n_res : 1, R_);
  FD_CHECK_LAUNCH("fd_ipa_points_fwd");
  return FD_OK;
}

extern "C" int fd_ipa_points_bwd(const float* proj, const float* quat, const float* dqp, const float* dkp,
                                 const float* dvp, float* dproj, float* dframe, long R_, int nheads, int c_hidden,
                                 int n_qk, int n_v, void* stream) {
  CHECK_DIMS("fd_ipa_points_bwd");
  if (R_ == 0) return FD_OK;
  hipLaunchKernelGGL(ipa_points_bwd_kernel, dim3((unsigned)R_), dim3(256), 0, (hipStream_t)stream, proj, quat, dqp,
                     dkp, dvp, dproj, dframe);
  FD_CHECK_LAUNCH("fd_ipa_points_bwd");
  return FD_OK;
}

extern "C" int fd_ipa_softmax_fwd(float* S, const float* zb, const float* qp, const float* kp, const float* head_w,
                                  const float* mask, int B, int N, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_softmax_fwd: N=%d exceeds %d", N, MAXN);
  if (B == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_fwd_kernel<MAXN, false>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                     (hipStream_t)stream, S, zb, qp, kp, (const float*)nullptr, head_w, mask, (float*)nullptr, N);
  FD_CHECK_LAUNCH("fd_ipa_softmax_fwd");
  return FD_OK;
}

extern "C" int fd_ipa_attn_fwd(float* S, const float* zb, const float* qp, const float* kp, const float* kp_soa,
                               const float* head_w, const float* mask, float* feats, int B, int N, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_attn_fwd: N=%d exceeds %d", N, MAXN);
  FD_CHECK_ARG(feats != nullptr, "fd_ipa_attn_fwd: feats is required");
  if (B == 0 || N == 0) return FD_OK;
  // few query rows (a lone backbone): one wave per head (see the kernel)
  static const int nt_env = getenv("FD_IPA_ATTN_THREADS") ? atoi(getenv("FD_IPA_ATTN_THREADS")) : 0;
  const bool wide = nt_env ? nt_env == 512 : (long)B * N <= FD_IPA_ATTN_WIDE_MAX_ROWS;
  if (N <= 128 && fd_aligned16(zb) && wide)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_fwd_kernel<128, true, 512>), dim3((unsigned)((long)B * N)), dim3(512), 0,
                       (hipStream_t)stream, S, zb, qp, kp, kp_soa, head_w, mask, feats, N);
  else if (N <= 256 && fd_aligned16(zb) && wide)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_fwd_kernel<256, true, 512>), dim3((unsigned)((long)B * N)), dim3(512), 0,
                       (hipStream_t)stream, S, zb, qp, kp, kp_soa, head_w, mask, feats, N);
  else if (N <= 128 && fd_aligned16(zb))
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_fwd_kernel<128, true>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, S, zb, qp, kp, kp_soa, head_w, mask, feats, N);
  else if (N <= 256 && fd_aligned16(zb))
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_fwd_kernel<256, true>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, S, zb, qp, kp, kp_soa, head_w, mask, feats, N);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_fwd_kernel<MAXN, false>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, S, zb, qp, kp, kp_soa, head_w, mask, feats, N);
  FD_CHECK_LAUNCH("fd_ipa_attn_fwd");
  return FD_OK;
}

extern "C" int fd_ipa_kpts_bwd(const float* dL, const float* qp, const float* kp, const float* head_w, float* dkp, int B,
                               int N, void* stream);

extern "C" int fd_ipa_softmax_bwd(const float* A, float* dA, const float* qp, const float* kp, const float* head_w,
                                  float* dzb, float* dqp, float* dkp, float* dhead_w, float* hw_part, int B, int N,
                                  void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_softmax_bwd: N=%d exceeds %d", N, MAXN);
  if (B == 0 || N == 0) return FD_OK;
  if (N <= 256)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_bwd_kernel<false, 256>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, A, dA, qp, kp, (const float*)nullptr, head_w, dzb, dqp, hw_part, (const float*)nullptr,
                       (const float*)nullptr, N);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_bwd_kernel<false, MAXN>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, A, dA, qp, kp, (const float*)nullptr, head_w, dzb, dqp, hw_part, (const float*)nullptr,
                       (const float*)nullptr, N);
  FD_CHECK_LAUNCH("fd_ipa_softmax_bwd");
  {
    int rc = fd_colsum_acc(hw_part, H, (long)B * N, H, dhead_w, stream);
    if (rc != FD_OK) return rc;
  }
  return fd_ipa_kpts_bwd(dA, qp, kp, head_w, dkp, B, N, stream);
}

extern "C" int fd_ipa_attn_bwd(const float* A, float* dA, const float* zb, const float* dfeats, const float* qp,
                               const float* kp, const float* kp_soa, const float* head_w, float* dzb, float* dqp,
                               float* dkp, float* dhead_w, float* hw_part, int B, int N, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_attn_bwd: N=%d exceeds %d", N, MAXN);
  FD_CHECK_ARG(zb && dfeats, "fd_ipa_attn_bwd: zb / dfeats are required");
  if (B == 0 || N == 0) return FD_OK;
  const bool al = fd_aligned16(zb) && fd_aligned16(dzb);
  if (N <= 128 && al)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_bwd_kernel<true, 128, true>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, A, dA, qp, kp, kp_soa, head_w, dzb, dqp, hw_part, zb, dfeats, N);
  else if (N <= 256 && al)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_bwd_kernel<true, 256, true>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, A, dA, qp, kp, kp_soa, head_w, dzb, dqp, hw_part, zb, dfeats, N);
  else if (N <= 256)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_bwd_kernel<true, 256>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, A, dA, qp, kp, kp_soa, head_w, dzb, dqp, hw_part, zb, dfeats, N);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_softmax_bwd_kernel<true, MAXN>), dim3((unsigned)((long)B * N)), dim3(256), 0,
                       (hipStream_t)stream, A, dA, qp, kp, kp_soa, head_w, dzb, dqp, hw_part, zb, dfeats, N);
  FD_CHECK_LAUNCH("fd_ipa_attn_bwd");
  {
    int rc = fd_colsum_acc(hw_part, H, (long)B * N, H, dhead_w, stream);
    if (rc != FD_OK) return rc;
  }
  return fd_ipa_kpts_bwd(dA, qp, kp, head_w, dkp, B, N, stream);
}

extern "C" int fd_ipa_kpts_bwd(const float* dL, const float* qp, const float* kp, const float* head_w, float* dkp, int B,
                               int N, void* stream) {
  if (B == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(ipa_kpts_bwd_kernel, dim3((unsigned)((N + 63) / 64), H, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, dL, qp, kp, head_w, dkp, N);
  FD_CHECK_LAUNCH("fd_ipa_kpts_bwd");
  return FD_OK;
}

extern "C" int fd_ipa_opt_fwd(const float* optg, const float* quat, const float* trans, float* feats, long R_,
                              void* stream) {
  if (R_ == 0) return FD_OK;
  hipLaunchKernelGGL(ipa_opt_fwd_kernel, dim3(grid1d(R_ * H * PV)), dim3(256), 0, (hipStream_t)stream, optg, quat,
                     trans, feats, R_);
  FD_CHECK_LAUNCH("fd_ipa_opt_fwd");
  return FD_OK;
}

extern "C" int fd_ipa_opt_bwd(const float* dfeats, const float* feats, const float* quat, float* doptg,
                              float* dframe, long R_, void* stream) {
  if (R_ == 0) return FD_OK;
  hipLaunchKernelGGL(ipa_opt_bwd_kernel, dim3((unsigned)R_), dim3(128), 0, (hipStream_t)stream, dfeats, feats, quat,
                     doptg, dframe, (const float*)nullptr, (float*)nullptr);
  FD_CHECK_LAUNCH("fd_ipa_opt_bwd");
  return FD_OK;
}

extern "C" int fd_ipa_opt_bwd_dot(const float* dfeats, const float* feats, const float* quat, const float* trans,
                                  float* doptg, float* dframe, float* ptdot, long R_, void* stream) {
  FD_CHECK_ARG(trans != nullptr && ptdot != nullptr, "fd_ipa_opt_bwd_dot: trans and ptdot are required");
  if (R_ == 0) return FD_OK;
  hipLaunchKernelGGL(ipa_opt_bwd_kernel, dim3((unsigned)R_), dim3(128), 0, (hipStream_t)stream, dfeats, feats, quat,
                     doptg, dframe, trans, ptdot);
  FD_CHECK_LAUNCH("fd_ipa_opt_bwd_dot");
  return FD_OK;
}

extern "C" int fd_ipa_opair_fwd(const float* A, const float* zb, float* feats, int B, int N, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_opair_fwd: N=%d exceeds %d", N, MAXN);
  if (B == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(ipa_opair_fwd_kernel, dim3((unsigned)((long)B * N)), dim3(256), 0, (hipStream_t)stream, A, zb,
                     feats, N);
  FD_CHECK_LAUNCH("fd_ipa_opair_fwd");
  return FD_OK;
}

extern "C" int fd_ipa_opair_bwd(const float* A, const float* zb, const float* dfeats, float* dA, float* dzb, int B,
                                int N, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_opair_bwd: N=%d exceeds %d", N, MAXN);
  if (B == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(ipa_opair_bwd_kernel, dim3((unsigned)((long)B * N)), dim3(256), 0, (hipStream_t)stream, A, zb,
                     dfeats, dA, dzb, N);
  FD_CHECK_LAUNCH("fd_ipa_opair_bwd");
  return FD_OK;
}
