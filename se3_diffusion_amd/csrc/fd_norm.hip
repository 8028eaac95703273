// LayerNorm forward/backward and column / pair reductions (HBM-bound, one wave per row).
//
// Replaces torch.nn.LayerNorm on the path: score_network.py:73,85 (embedder),
// ipa_pytorch.py:189 (node transition), :231 (edge transition), :577,632 (ipa_ln),
// TransformerEncoderLayer.norm1/norm2 (ipa_pytorch.py:584-593); eps = 1e-5,
// biased variance.  Optional per-row output scale fuses the `* node_mask` /
// `* edge_mask` that always follows (ipa_pytorch.py:641,649; score_network.py:194-195).
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int MAXC_PER_LANE = 8;  // C <= 512

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ rowscale, float* __restrict__ y, long ldy, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, long rows, int C, float eps) {
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const int npl = C / 64;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* xr = x + row * ldx;
    float v[MAXC_PER_LANE], gm[MAXC_PER_LANE], bt[MAXC_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) { v[k] = xr[lane + 64 * k]; s += v[k]; }
    // (the scale / shift / row-scale operands are fetched beside x, not behind the two reductions: at sampling sizes a launch of
    // this kernel is one row per wave, i.e. a chain of dependent latencies)
    const float rs = rowscale ? rowscale[row] : 1.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) { gm[k] = gamma[lane + 64 * k]; bt[k] = beta[lane + 64 * k]; }
    const float mean = fd::wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) { float d = v[k] - mean; q += d * d; }
    const float var = fd::wave_sum(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    float* yr = y + row * ldy;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) {
        int c = lane + 64 * k;
        yr[c] = ((v[k] - mean) * rstd * gm[k] + bt[k]) * rs;
      }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*rowscale*gamma
// dgamma += sum_rows dy*rowscale*xhat ; dbeta += sum_rows dy*rowscale
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
    const float* __restrict__ gamma, const float* __restrict__ rowscale, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, long lddx, int dx_accum,
    float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int C) {
  __shared__ float red[2][4][512];
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const int npl = C / 64;
  float ag[MAXC_PER_LANE], ab[MAXC_PER_LANE];
#pragma unroll
  for (int k = 0; k < MAXC_PER_LANE; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float m = mean[row], r = rstd[row];
    const float rs = rowscale ? rowscale[row] : 1.f;
    const float* xr = x + row * ldx;
    const float* dyr = dy + row * lddy;
    float xh[MAXC_PER_LANE], g[MAXC_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) {
        int c = lane + 64 * k;
        float d = dyr[c] * rs;
        xh[k] = (xr[c] - m) * r;
        g[k] = d * gamma[c];
        ag[k] += d * xh[k];
        ab[k] += d;
        s1 += g[k];
        s2 += g[k] * xh[k];
      }
    s1 = fd::wave_sum(s1) / (float)C;
    s2 = fd::wave_sum(s2) / (float)C;
    float* dxr = dx + row * lddx;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) {
        int c = lane + 64 * k;
        float v = r * (g[k] - s1 - xh[k] * s2);
        dxr[c] = dx_accum ? dxr[c] + v : v;
      }
  }
  if (dgamma) {
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) { red[0][wave][lane + 64 * k] = ag[k]; red[1][wave][lane + 64 * k] = ab[k]; }
    __syncthreads();
    for (int c = (int)threadIdx.x; c < C; c += 256) {
      float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
      float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
      atomicAdd(&dgamma[c], a);
      atomicAdd(&dbeta[c], b);
    }
  }
}

// Node-level form of the backward (rows = B * N residues, C = 256 / 320): a wave takes LNB_U rows per trip with float4 accesses and
// ALL of their loads in flight before the first reduction; at 3,840 rows every wave makes exactly one trip.  (The one-row-per-trip
// kernel above walks its four rows as four dependent load -> reduce -> store chains: 21 us per launch in the training step,
// 25 launches per step -- round 4's serialised profile.)  NCH = float4 chunks per lane (1: C <= 256, 2: C <= 512).
constexpr int LNB_U = 4;
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd4_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
    const float* __restrict__ gamma, const float* __restrict__ rowscale, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, long lddx, int dx_accum,
    float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int C) {
  __shared__ float red[2][4][512];
  const int lane = fd::lane_id(), wave = fd::wave_id();
  int col[NCH];
  bool cok[NCH];
  float4 gm[NCH], ag[NCH], ab[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    col[k] = 4 * (lane + 64 * k);
    cok[k] = col[k] < C;
    if (!cok[k]) col[k] = 0;                        // (lanes past the row read column 0 and contribute zeros)
    gm[k] = *reinterpret_cast<const float4*>(gamma + col[k]);
    ag[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[k] = ag[k];
  }
  const float invc = 1.0f / (float)C;
  const long wslot = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
  for (long base = wslot * LNB_U; base < rows; base += nwaves * LNB_U) {
    float4 xv[LNB_U][NCH], dv[LNB_U][NCH], old[LNB_U][NCH];
    float m[LNB_U], rr[LNB_U], rs[LNB_U];
#pragma unroll
    for (int u = 0; u < LNB_U; ++u) {
      const long r = base + u < rows ? base + u : rows - 1;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        xv[u][k] = *reinterpret_cast<const float4*>(x + r * ldx + col[k]);
        dv[u][k] = *reinterpret_cast<const float4*>(dy + r * lddy + col[k]);
        if (dx_accum) old[u][k] = *reinterpret_cast<const float4*>(dx + r * lddx + col[k]);
      }
      m[u] = mean[r];
      rr[u] = rstd[r];
      rs[u] = rowscale ? rowscale[r] : 1.f;
    }
    float4 xh[LNB_U][NCH], g[LNB_U][NCH];
    float s1[LNB_U], s2[LNB_U];
#pragma unroll
    for (int u = 0; u < LNB_U; ++u) {
      const bool ok = base + u < rows;
      s1[u] = 0.f;
      s2[u] = 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const float sc = (ok && cok[k]) ? rs[u] : 0.f;   // rows / columns past the end contribute nothing
        const float4 d = make_float4(dv[u][k].x * sc, dv[u][k].y * sc, dv[u][k].z * sc, dv[u][k].w * sc);
        xh[u][k] = make_float4((xv[u][k].x - m[u]) * rr[u], (xv[u][k].y - m[u]) * rr[u], (xv[u][k].z - m[u]) * rr[u],
                               (xv[u][k].w - m[u]) * rr[u]);
        g[u][k] = make_float4(d.x * gm[k].x, d.y * gm[k].y, d.z * gm[k].z, d.w * gm[k].w);
        ag[k].x += d.x * xh[u][k].x; ag[k].y += d.y * xh[u][k].y; ag[k].z += d.z * xh[u][k].z; ag[k].w += d.w * xh[u][k].w;
        ab[k].x += d.x; ab[k].y += d.y; ab[k].z += d.z; ab[k].w += d.w;
        s1[u] += (g[u][k].x + g[u][k].y) + (g[u][k].z + g[u][k].w);
        s2[u] += (g[u][k].x * xh[u][k].x + g[u][k].y * xh[u][k].y) + (g[u][k].z * xh[u][k].z + g[u][k].w * xh[u][k].w);
      }
    }
#pragma unroll
    for (int u = 0; u < LNB_U; ++u) {
      s1[u] = fd::wave_sum(s1[u]) * invc;
      s2[u] = fd::wave_sum(s2[u]) * invc;
    }
#pragma unroll
    for (int u = 0; u < LNB_U; ++u) {
      if (base + u >= rows) continue;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (!cok[k]) continue;
        float4 o;
        o.x = rr[u] * (g[u][k].x - s1[u] - xh[u][k].x * s2[u]);
        o.y = rr[u] * (g[u][k].y - s1[u] - xh[u][k].y * s2[u]);
        o.z = rr[u] * (g[u][k].z - s1[u] - xh[u][k].z * s2[u]);
        o.w = rr[u] * (g[u][k].w - s1[u] - xh[u][k].w * s2[u]);
        if (dx_accum) { o.x += old[u][k].x; o.y += old[u][k].y; o.z += old[u][k].z; o.w += old[u][k].w; }
        *reinterpret_cast<float4*>(dx + (base + u) * lddx + col[k]) = o;
      }
    }
  }
  if (dgamma) {
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      if (cok[k]) {
        *reinterpret_cast<float4*>(&red[0][wave][col[k]]) = ag[k];
        *reinterpret_cast<float4*>(&red[1][wave][col[k]]) = ab[k];
      }
    __syncthreads();
    for (int c = (int)threadIdx.x; c < C; c += 256) {
      float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
      float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
      atomicAdd(&dgamma[c], a);
      atomicAdd(&dbeta[c], b);
    }
  }
}

// ---- C == 128 fast path (the pair tensor z: every edge-transition / edge-embedder LayerNorm) --------------------
// A row is 512 B = 32 lanes x float4, so a wave holds two rows side by side and each half-wave walks LN_U rows per
// iteration with all of their loads issued before the first reduction (the generic one-row-per-wave kernels keep
// only two 4-byte loads per lane in flight, ~40 % of the HBM rate on the 252 MB pair tensor).
constexpr int LN_U = 4;

__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void layernorm_fwd128_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ rowscale, float* __restrict__ y, long ldy, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, long rows, float eps) {
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const int hl = lane & 31, half = lane >> 5;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * hl);
  const float4 bt = *reinterpret_cast<const float4*>(beta + 4 * hl);
  // wave-uniform trip count (both halves shuffle in lock step); rows past the end are clamped on load, masked on store
  const long wslot = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
  for (long base = wslot * (2 * LN_U); base < rows; base += nwaves * (2 * LN_U)) {
    const long row0 = base + half * LN_U;
    float4 v[LN_U];
    float rs[LN_U];
#pragma unroll
    for (int u = 0; u < LN_U; ++u) {
      const long r = row0 + u < rows ? row0 + u : rows - 1;
      v[u] = *reinterpret_cast<const float4*>(x + r * ldx + 4 * hl);
      rs[u] = rowscale ? rowscale[r] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < LN_U; ++u) {
      const float mean = half_wave_sum((v[u].x + v[u].y) + (v[u].z + v[u].w)) * (1.0f / 128.0f);
      const float dx = v[u].x - mean, dy = v[u].y - mean, dz = v[u].z - mean, dw = v[u].w - mean;
      const float var = half_wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 128.0f);
      const float rstd = 1.0f / sqrtf(var + eps);
      const long r = row0 + u;
      if (r < rows) {
        float4 o;
        o.x = (dx * rstd * gm.x + bt.x) * rs[u];
        o.y = (dy * rstd * gm.y + bt.y) * rs[u];
        o.z = (dz * rstd * gm.z + bt.z) * rs[u];
        o.w = (dw * rstd * gm.w + bt.w) * rs[u];
        *reinterpret_cast<float4*>(y + r * ldy + 4 * hl) = o;
        if (hl == 0) {
          if (mean_out) mean_out[r] = mean;
          if (rstd_out) rstd_out[r] = rstd;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd128_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
    const float* __restrict__ gamma, const float* __restrict__ rowscale, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, long lddx, int dx_accum,
    float* __restrict__ dgamma, float* __restrict__ dbeta, long rows) {
  __shared__ float4 red[2][8][32];
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const int hl = lane & 31, half = lane >> 5;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * hl);
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
  // wave-uniform trip count (both halves shuffle in lock step); rows past the end are clamped on load, masked on store
  const long wslot = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
  for (long base = wslot * (2 * LN_U); base < rows; base += nwaves * (2 * LN_U)) {
    const long row0 = base + half * LN_U;
    float4 xv[LN_U], dv[LN_U], old[LN_U];
    float m[LN_U], rr[LN_U], rs[LN_U];
#pragma unroll
    for (int u = 0; u < LN_U; ++u) {
      const long r = row0 + u < rows ? row0 + u : rows - 1;
      xv[u] = *reinterpret_cast<const float4*>(x + r * ldx + 4 * hl);
      dv[u] = *reinterpret_cast<const float4*>(dy + r * lddy + 4 * hl);
      if (dx_accum) old[u] = *reinterpret_cast<const float4*>(dx + r * lddx + 4 * hl);
      m[u] = mean[r];
      rr[u] = rstd[r];
      rs[u] = rowscale ? rowscale[r] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < LN_U; ++u) {
      const bool ok = row0 + u < rows;
      const float sc = ok ? rs[u] : 0.f;   // rows past the end contribute nothing to dgamma / dbeta
      const float4 d = make_float4(dv[u].x * sc, dv[u].y * sc, dv[u].z * sc, dv[u].w * sc);
      const float4 xh = make_float4((xv[u].x - m[u]) * rr[u], (xv[u].y - m[u]) * rr[u], (xv[u].z - m[u]) * rr[u],
                                    (xv[u].w - m[u]) * rr[u]);
      const float4 g = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      ag.x += d.x * xh.x; ag.y += d.y * xh.y; ag.z += d.z * xh.z; ag.w += d.w * xh.w;
      ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
      const float s1 = half_wave_sum((g.x + g.y) + (g.z + g.w)) * (1.0f / 128.0f);
      const float s2 = half_wave_sum((g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w)) * (1.0f / 128.0f);
      if (ok) {
        float4 o;
        o.x = rr[u] * (g.x - s1 - xh.x * s2);
        o.y = rr[u] * (g.y - s1 - xh.y * s2);
        o.z = rr[u] * (g.z - s1 - xh.z * s2);
        o.w = rr[u] * (g.w - s1 - xh.w * s2);
        if (dx_accum) { o.x += old[u].x; o.y += old[u].y; o.z += old[u].z; o.w += old[u].w; }
        *reinterpret_cast<float4*>(dx + (row0 + u) * lddx + 4 * hl) = o;
      }
    }
  }
  if (dgamma) {
    red[0][wave * 2 + half][hl] = ag;
    red[1][wave * 2 + half][hl] = ab;
    __syncthreads();
    // thread t: channel t & 127 of dgamma (t < 128) or dbeta: consecutive lanes hit consecutive addresses
    const int t = (int)threadIdx.x;
    const int w = t >> 7, ch = t & 127;
    const float* r0 = reinterpret_cast<const float*>(&red[w][0][0]);
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += r0[k * 128 + ch];
    atomicAdd((w == 0 ? dgamma : dbeta) + ch, a);
  }
}

// out[n] += sum_m X[m*ld + n]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, long rows, int ncols,
                                                      float* __restrict__ out) {
  // thread t owns column (t % 64) + 64*cc, row phase t/64; grid-stride over row chunks
  __shared__ float red[4][64];
  const int lane = fd::lane_id(), wave = fd::wave_id();
  for (int c0 = 0; c0 < ncols; c0 += 64) {
    const int c = c0 + lane;
    float acc = 0.f;
    if (c < ncols)
      for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) acc += X[r * ld + c];
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < ncols) atomicAdd(&out[c], red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    __syncthreads();
  }
}

// X is [B, n, n, C] (pair tensor).  rowsum[b,i,c] (+)= sum_j X[b,i,j,c]; colsum[b,j,c] (+)= sum_i X[b,i,j,c].
// One block per (b, i): row sums are block-local, column sums go through atomics.
__global__ __launch_bounds__(256) void pair_reduce_kernel(const float* __restrict__ X, int n, int C,
                                                          float* __restrict__ rowsum, float* __restrict__ colsum,
                                                          long ld_out) {
  const long bi = blockIdx.x;            // b*n + i
  const long b = bi / n;
  const float* xb = X + bi * (long)n * C;
  for (int c = (int)threadIdx.x; c < C; c += (int)blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < n; ++j) {
      float v = xb[(long)j * C + c];
      acc += v;
      if (colsum) atomicAdd(&colsum[(b * n + j) * ld_out + c], v);
    }
    if (rowsum) rowsum[bi * ld_out + c] += acc;
  }
}

// colsum without atomics: one block per (b, j) looping over i (strided rows of C contiguous floats)
__global__ __launch_bounds__(256) void pair_colsum_kernel(const float* __restrict__ X, int n, int C,
                                                          float* __restrict__ colsum, long ld_out) {
  const long bj = blockIdx.x;
  const long b = bj / n, j = bj % n;
  const float* xb = X + (b * n * (long)n + j) * C;
  for (int c = (int)threadIdx.x; c < C; c += (int)blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += xb[(long)i * n * C + c];
    colsum[bj * ld_out + c] += acc;
  }
}

// Both pair reductions in ONE pass over X [nbatch, n, n, C] (the two-kernel path reads X twice).  A block owns
// (b, a chunk of 32 i, 32 channels, a tile of 128 j): thread = (float4 channel group c4 of 8, j group jg of 32).  The
// column sums over the block's i stay in registers (4 float4 per thread) and leave as one atomicAdd per element and
// block (n/32 partials per output); the row sums over j are reduced with shuffles inside a wave and through LDS
// across the four waves.  30 x 4 x 12 = 1440 blocks at B=30, N=128, C=384.
constexpr int PR_I = 32, PR_J = 128, PR_C = 32;
__global__ __launch_bounds__(256) void pair_reduce2_kernel(const float* __restrict__ X, int n, int C,
                                                           float* __restrict__ rowsum, float* __restrict__ colsum,
                                                           long ld_out, int nich, int ncc) {
  __shared__ float4 red[PR_I][4][8];   // 16 KB
  const int tid = (int)threadIdx.x;
  const int c4 = tid & 7, jg = tid >> 3, wave = tid >> 6;
  int bx = (int)blockIdx.x;
  const int cc = bx % ncc; bx /= ncc;
  const int ic = bx % nich;
  const long b = bx / nich;
  const int j0 = (int)blockIdx.y * PR_J;
  const int c = cc * PR_C + 4 * c4;
  const int i0 = ic * PR_I;
  float4 cs[PR_J / 32];
#pragma unroll
  for (int q = 0; q < PR_J / 32; ++q) cs[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int il = 0; il < PR_I; ++il) {
    const int i = i0 + il;
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {   // (uniform per block)
      const float* xi = X + ((b * n + i) * (long)n) * C + c;
#pragma unroll
      for (int q = 0; q < PR_J / 32; ++q) {
        const int j = j0 + jg + 32 * q;
        if (j < n) {
          const float4 v = *reinterpret_cast<const float4*>(xi + (long)j * C);
          rs.x += v.x; rs.y += v.y; rs.z += v.z; rs.w += v.w;
          cs[q].x += v.x; cs[q].y += v.y; cs[q].z += v.z; cs[q].w += v.w;
        }
      }
    }
    // lanes of a wave with equal c4 differ in lane bits 3..5
    rs.x += fd::lane_xor8(rs.x); rs.y += fd::lane_xor8(rs.y); rs.z += fd::lane_xor8(rs.z); rs.w += fd::lane_xor8(rs.w);   // (DPP)
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      rs.x += __shfl_xor(rs.x, o); rs.y += __shfl_xor(rs.y, o); rs.z += __shfl_xor(rs.z, o); rs.w += __shfl_xor(rs.w, o);
    }
    if ((tid & 63) < 8) red[il][wave][c4] = rs;
  }
#pragma unroll
  for (int q = 0; q < PR_J / 32; ++q) {
    const int j = j0 + jg + 32 * q;
    if (j < n) {
      float* o = colsum + (b * n + j) * ld_out + c;
      atomicAdd(o + 0, cs[q].x); atomicAdd(o + 1, cs[q].y); atomicAdd(o + 2, cs[q].z); atomicAdd(o + 3, cs[q].w);
    }
  }
  __syncthreads();
  // row sums: 32 i x 8 channel groups = 256 float4 outputs, one per thread
  {
    const int il = tid >> 3, g = tid & 7;
    const int i = i0 + il;
    if (i < n) {
      float4 a = red[il][0][g];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float4 v = red[il][k][g];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      float* o = rowsum + (b * n + i) * ld_out + cc * PR_C + 4 * g;
      if (gridDim.y == 1) {   // this block owns the whole j range of its rows
        o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
      } else {
        atomicAdd(o + 0, a.x); atomicAdd(o + 1, a.y); atomicAdd(o + 2, a.z); atomicAdd(o + 3, a.w);
      }
    }
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float a,
                                                    float b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a * x[i] + b * y[i];
}

// y[r, c] = x[r, c] * rowscale[r]  (strided 2-D, used for masks)
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ rs, float* __restrict__ y, long ldy,
                                                       long rows, int C) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i / C;
    int c = (int)(i % C);
    y[r * ldy + c] = x[r * ldx + c] * rs[r];
  }
}

// vectorised variant (ld % 4 == 0, ncols % 4 == 0, 16-byte aligned): lanes span 256 columns with float4 loads,
// four independent row streams per wave keep loads in flight
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ X, long ld, long rows, int ncols,
                                                       float* __restrict__ out) {
  __shared__ float4 red[4][64];
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const long wg = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
  for (int c0 = 0; c0 < ncols; c0 += 256) {
    const int c = c0 + lane * 4;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0, acc2 = acc0, acc3 = acc0;
    if (c < ncols) {
      long r = wg;
      for (; r + 3 * nw < rows; r += 4 * nw) {
        const float4 v0 = *reinterpret_cast<const float4*>(X + r * ld + c);
        const float4 v1 = *reinterpret_cast<const float4*>(X + (r + nw) * ld + c);
        const float4 v2 = *reinterpret_cast<const float4*>(X + (r + 2 * nw) * ld + c);
        const float4 v3 = *reinterpret_cast<const float4*>(X + (r + 3 * nw) * ld + c);
        acc0.x += v0.x; acc0.y += v0.y; acc0.z += v0.z; acc0.w += v0.w;
        acc1.x += v1.x; acc1.y += v1.y; acc1.z += v1.z; acc1.w += v1.w;
        acc2.x += v2.x; acc2.y += v2.y; acc2.z += v2.z; acc2.w += v2.w;
        acc3.x += v3.x; acc3.y += v3.y; acc3.z += v3.z; acc3.w += v3.w;
      }
      for (; r < rows; r += nw) {
        const float4 v0 = *reinterpret_cast<const float4*>(X + r * ld + c);
        acc0.x += v0.x; acc0.y += v0.y; acc0.z += v0.z; acc0.w += v0.w;
      }
    }
    float4 s;
    s.x = (acc0.x + acc1.x) + (acc2.x + acc3.x);
    s.y = (acc0.y + acc1.y) + (acc2.y + acc3.y);
    s.z = (acc0.z + acc1.z) + (acc2.z + acc3.z);
    s.w = (acc0.w + acc1.w) + (acc2.w + acc3.w);
    red[wave][lane] = s;
    __syncthreads();
    {
      // thread t adds column c0 + t: one atomic instruction per wave over 64 consecutive floats (2 cache lines)
      // instead of four stride-4 instructions over 8 lines each (same-line atomics serialise in the L2)
      const int t = (int)threadIdx.x, cc = c0 + t;
      if (cc < ncols) {
        const float* r0 = reinterpret_cast<const float*>(&red[0][0]);
        atomicAdd(&out[cc], (r0[t] + r0[256 + t]) + (r0[512 + t] + r0[768 + t]));
      }
    }
    __syncthreads();
  }
}

// dst[r*ldd + c] += a * src[r*lds + c]   (gradient accumulation into column blocks / views)
__global__ __launch_bounds__(256) void add2d_kernel(float* __restrict__ dst, long ldd, const float* __restrict__ src,
                                                    long lds, long rows, int cols, float a) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i / cols;
    int c = (int)(i % cols);
    dst[r * ldd + c] += a * src[r * lds + c];
  }
}

}  // namespace

static int ln_grid(long rows) {
  long g = (rows + 3) / 4;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

static bool ln128_ok(const float* p, long ld) { return fd_aligned16(p) && (ld & 3) == 0; }

extern "C" int fd_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta,
                                const float* rowscale, float* y, long ldy, float* mean, float* rstd, long rows,
                                int C, float eps, void* stream) {
  FD_CHECK_ARG(C % 64 == 0 && C <= 64 * MAXC_PER_LANE, "fd_layernorm_fwd: C=%d must be a multiple of 64 <= 512", C);
  if (rows == 0) return FD_OK;
  if (C == 128 && ln128_ok(x, ldx) && ln128_ok(y, ldy) && fd_aligned16(gamma) && fd_aligned16(beta)) {
    long g = (rows + 8 * LN_U - 1) / (8 * LN_U);
    int grid = (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
    hipLaunchKernelGGL(layernorm_fwd128_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                       rowscale, y, ldy, mean, rstd, rows, eps);
    FD_CHECK_LAUNCH("fd_layernorm_fwd(128)");
    return FD_OK;
  }
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma,
                     beta, rowscale, y, ldy, mean, rstd, rows, C, eps);
  FD_CHECK_LAUNCH("fd_layernorm_fwd");
  return FD_OK;
}

extern "C" int fd_layernorm_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                                const float* rowscale, const float* mean, const float* rstd, float* dx, long lddx,
                                int dx_accum, float* dgamma, float* dbeta, long rows, int C, void* stream) {
  FD_CHECK_ARG(C % 64 == 0 && C <= 64 * MAXC_PER_LANE, "fd_layernorm_bwd: C=%d must be a multiple of 64 <= 512", C);
  FD_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "fd_layernorm_bwd: dgamma/dbeta come together");
  if (rows == 0) return FD_OK;
  if (C == 128 && ln128_ok(x, ldx) && ln128_ok(dy, lddy) && ln128_ok(dx, lddx) && fd_aligned16(gamma)) {
    long g128 = (rows + 8 * LN_U - 1) / (8 * LN_U);
    int grid128 = (int)(g128 < 1 ? 1 : (g128 > 512 ? 512 : g128));
    hipLaunchKernelGGL(layernorm_bwd128_kernel, dim3(grid128), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx,
                       gamma, rowscale, mean, rstd, dx, lddx, dx_accum, dgamma, dbeta, rows);
    FD_CHECK_LAUNCH("fd_layernorm_bwd(128)");
    return FD_OK;
  }
  // every block ends with 2*C atomics on the same dgamma / dbeta addresses: keep the block count at ~ one per CU
  // for the node-level calls (rows = B*N), more only when there is real streaming work
  long g = (rows + 15) / 16;
  int grid = (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
  if (ln128_ok(x, ldx) && ln128_ok(dy, lddy) && ln128_ok(dx, lddx) && fd_aligned16(gamma)) {
    // (16-byte addressable rows: LNB_U rows per wave and trip, every load in flight before the first reduction)
    if (C <= 256)
      hipLaunchKernelGGL(layernorm_bwd4_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, gamma,
                         rowscale, mean, rstd, dx, lddx, dx_accum, dgamma, dbeta, rows, C);
    else
      hipLaunchKernelGGL(layernorm_bwd4_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, gamma,
                         rowscale, mean, rstd, dx, lddx, dx_accum, dgamma, dbeta, rows, C);
    FD_CHECK_LAUNCH("fd_layernorm_bwd(x4)");
    return FD_OK;
  }
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, gamma,
                     rowscale, mean, rstd, dx, lddx, dx_accum, dgamma, dbeta, rows, C);
  FD_CHECK_LAUNCH("fd_layernorm_bwd");
  return FD_OK;
}

extern "C" int fd_colsum_acc(const float* X, long ld, long rows, int ncols, float* out, void* stream) {
  if (rows == 0 || ncols == 0) return FD_OK;
  long g = (rows + 15) / 16;
  int grid = (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
  if ((ld & 3) == 0 && (ncols & 3) == 0 && fd_aligned16(X)) {
    hipLaunchKernelGGL(colsum4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ld, rows, ncols, out);
    FD_CHECK_LAUNCH("fd_colsum_acc");
    return FD_OK;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ld, rows, ncols, out);
  FD_CHECK_LAUNCH("fd_colsum_acc");
  return FD_OK;
}

extern "C" int fd_pair_reduce_acc(const float* X, int nbatch, int n, int C, float* rowsum, float* colsum,
                                  long ld_out, void* stream) {
  if (nbatch == 0 || n == 0) return FD_OK;
  if (rowsum && colsum && (C % PR_C) == 0 && fd_aligned16(X)) {
    const int nich = fd_cdiv(n, PR_I), ncc = C / PR_C;
    hipLaunchKernelGGL(pair_reduce2_kernel, dim3(nbatch * nich * ncc, fd_cdiv(n, PR_J)), dim3(256), 0, (hipStream_t)stream,
                       X, n, C, rowsum, colsum, ld_out, nich, ncc);
    FD_CHECK_LAUNCH("fd_pair_reduce_acc(fused)");
    return FD_OK;
  }
  int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
  if (rowsum) {
    hipLaunchKernelGGL(pair_reduce_kernel, dim3(nbatch * n), dim3(threads), 0, (hipStream_t)stream, X, n, C, rowsum,
                       (float*)nullptr, ld_out);
    FD_CHECK_LAUNCH("fd_pair_reduce_acc(row)");
  }
  if (colsum) {
    hipLaunchKernelGGL(pair_colsum_kernel, dim3(nbatch * n), dim3(threads), 0, (hipStream_t)stream, X, n, C, colsum,
                       ld_out);
    FD_CHECK_LAUNCH("fd_pair_reduce_acc(col)");
  }
  return FD_OK;
}

extern "C" int fd_axpby(float* y, const float* x, float a, float b, long n, void* stream) {
  if (n == 0) return FD_OK;
  long g = (n + 255) / 256;
  int grid = (int)(g > 4096 ? 4096 : g);
  hipLaunchKernelGGL(axpby_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, a, b, n);
  FD_CHECK_LAUNCH("fd_axpby");
  return FD_OK;
}

extern "C" int fd_rowscale(const float* x, long ldx, const float* rs, float* y, long ldy, long rows, int C,
                           void* stream) {
  if (rows == 0 || C == 0) return FD_OK;
  long g = (rows * C + 255) / 256;
  int grid = (int)(g > 4096 ? 4096 : g);
  hipLaunchKernelGGL(rowscale_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, rs, y, ldy, rows, C);
  FD_CHECK_LAUNCH("fd_rowscale");
  return FD_OK;
}

extern "C" int fd_add2d(float* dst, long ldd, const float* src, long lds, long rows, int cols, float a,
                        void* stream) {
  if (rows == 0 || cols == 0) return FD_OK;
  long g = (rows * cols + 255) / 256;
  int grid = (int)(g > 4096 ? 4096 : g);
  hipLaunchKernelGGL(add2d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, rows, cols, a);
  FD_CHECK_LAUNCH("fd_add2d");
  return FD_OK;
}
