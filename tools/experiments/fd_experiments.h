/* EXPERIMENTS -- kernels that were built, parity-tested on gfx950 and measured SLOWER than the launch sequences they were meant to
 * replace.  They are NOT part of the product library (libfd_hip.so), its C ABI (include/fd_hip.h) or its tests; this header
 * and tools/experiments/build.py exist so that they keep compiling and can be re-measured.  See tools/experiments/README.md. */
#ifndef FD_EXPERIMENTS_H
#define FD_EXPERIMENTS_H
#include "../../include/fd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Block-diagonal form of the same kernel: three independent products over the same pair rows in one pass,
 *   C_i[m * ldc_i + n] += sum_p A_i[p, m] * B_i[p, n]     i = 0..2, m < 128, n < b_cols_i (0 = 128)
 * -- the weight gradients of the edge embedder's three Linear layers (autograd of score_network.py:67-86, 194-195);
 * a_colsum_i [128] (all three or none): += sum_p A_i[p, :], the bias gradients. */
typedef struct FdPairDwDiagDesc {
  const float* A[3];    /* [rows, 128], row stride lda[i] */
  long lda[3];
  const float* B[3];    /* [rows, b_cols[i]], row stride ldb[i] */
  long ldb[3];
  float* C[3];          /* [128, b_cols[i]], row stride ldc[i] */
  long ldc[3];
  float* a_colsum[3];
  int b_cols[3];
  long rows;            /* B * nres * nres */
  int blocks;           /* 0 = one persistent block per CU (256) */
} FdPairDwDiagDesc;
int fd_pair_dw_diag(const FdPairDwDiagDesc* desc, void* stream);

/* zb[p, 0:40] = W40[0:40, 0:128] z[p, 0:128] + b40 over the pair rows (linear_b and down_z in one streaming pass over z,
 * ipa_pytorch.py:380-386,455): W40 resident in registers; b40 may be null */
int fd_ipa_zb(const float* z, const float* W40, const float* b40, float* zb, long rows, void* stream);

/* The pair pass of IPA in one kernel per direction (se3_diffusion_amd/csrc/fd_ipa_pair.hip): z is read once for both
 * linear_b and down_z; the [P,40] projections zb / dzb stay in LDS.  W40 = [linear_b.weight ; down_z.weight] [40,128],
 * b40 likewise.  N <= 512.
 *   fwd: S [B,8,N,N] holds sqrt(1/(3C)) q k^T on entry and the attention probabilities on return; feats[:, o_pair] written.
 *   bwd: dA holds dO V^T + d(o_pt) v_pts^T on entry and dLogits on return; dz (+)= dzb W40 (dz_accumulate 0: assign);
 *        dW40 [40,128] and db40 [40] accumulated atomically; dqp / dkp / dhead_w as fd_ipa_softmax_bwd. */
int fd_ipa_pair_fwd(float* S, const float* z, const float* W40, const float* b40, const float* qp, const float* kp,
                    const float* head_w, const float* mask, float* feats, int B, int N, void* stream);
int fd_ipa_pair_bwd(const float* A, float* dA, const float* z, const float* W40, const float* b40, const float* dfeats,
                    const float* qp, const float* kp, const float* head_w, float* dz, int dz_accumulate, float* dqp,
                    float* dkp, float* dhead_w, float* hw_part, float* dW40, float* db40, int B, int N, void* stream);

#ifdef __cplusplus
}
#endif
#endif
