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
    float v[MAXC_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) { v[k] = xr[lane + 64 * k]; s += v[k]; }
    const float mean = fd::wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) { float d = v[k] - mean; q += d * d; }
    const float var = fd::wave_sum(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float rs = rowscale ? rowscale[row] : 1.f;
    float* yr = y + row * ldy;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k)
      if (k < npl) {
        int c = lane + 64 * k;
        yr[c] = ((v[k] - mean) * rstd * gamma[c] + beta[c]) * rs;
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
    if (wave == 0 && c < ncols) {
      const float4 a = red[0][lane], b = red[1][lane], e = red[2][lane], f = red[3][lane];
      atomicAdd(&out[c + 0], (a.x + b.x) + (e.x + f.x));
      atomicAdd(&out[c + 1], (a.y + b.y) + (e.y + f.y));
      atomicAdd(&out[c + 2], (a.z + b.z) + (e.z + f.z));
      atomicAdd(&out[c + 3], (a.w + b.w) + (e.w + f.w));
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

extern "C" int fd_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta,
                                const float* rowscale, float* y, long ldy, float* mean, float* rstd, long rows,
                                int C, float eps, void* stream) {
  FD_CHECK_ARG(C % 64 == 0 && C <= 64 * MAXC_PER_LANE, "fd_layernorm_fwd: C=%d must be a multiple of 64 <= 512", C);
  if (rows == 0) return FD_OK;
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
  long g = (rows + 3) / 4;
  int grid = (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
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
