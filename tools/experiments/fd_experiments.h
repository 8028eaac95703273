/* EXPERIMENTS -- kernels that were built, parity-tested on gfx950 and measured SLOWER than the launch sequences they were meant to
 * replace.  They are NOT part of the product library (libfd_hip.so), its C ABI (include/fd_hip.h) or its tests; this header
 * and tools/experiments/build.py exist so that they keep compiling and can be re-measured.  See tools/experiments/README.md. */
#ifndef FD_EXPERIMENTS_H
#define FD_EXPERIMENTS_H
#include "../../include/fd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- residual MLP blocks of the node level, one launch per block and direction (tools/experiments/fd_node_chain.hip) ----
 * (width, nlayers) = (256, 3): StructureModuleTransition.forward (model/ipa_pytorch.py:169-191) followed by the block's
 *   node mask (:644);  (320, 2): the feed-forward half of the sequence transformer's TransformerEncoderLayer (post-norm:
 *   linear1 - ReLU - linear2 - residual - norm2; built at ipa_pytorch.py:584-595) -- and the input-gradient chains of both.
 * forward:   h_0 = x;  h_l = relu(W_l h_{l-1} + b_l) (l < nlayers);  t = W_nl h_{nl-1} + b_nl + x;
 *            out = rowscale * (LayerNorm(t) * gamma + beta).   Optional outputs (training): save[l-1] = h_l, pre = t, mean, rstd.
 * backward:  x = gradient of out;  dt = LayerNorm-backward(rowscale * x; ln_in = t, mean, rstd, gamma) -> pre (required),
 *            dgamma / dbeta accumulated;  d_{nl-1} = [h_{nl-1} > 0] (dt W_nl) -> save[0], gate[0] = h_{nl-1};  ... ;
 *            out = dt + d_1 W_1.   (The weight gradients are dY^T X products over pre / save / the forward's saves: fd_group_dw.)
 * Image: nlayers layers of FD_NODE_CHAIN_LAYER_BYTES(width), each written by fd_node_chain_pack(A, rs, cs, ...) with
 * A[n][k] = A[n * rs + k * cs] the layer's [out, in] matrix -- forward: W_1 (chained = 0), W_2.. (chained = 1);
 * backward: W_nl^T, ..., W_1^T (rs = 1, cs = ld; all chained = 1).  Split-bf16 arithmetic (fp32-accurate, as fd_gemm tile 4). */
#define FD_NODE_CHAIN_LAYER_BYTES(width) (((width) / 32) * ((width) / 64) * 12288)
int fd_node_chain_pack(const float* A, long rs, long cs, int width, int chained, void* img_layer, void* stream);
typedef struct FdNodeChainDesc {
  const float* x;         /* [rows, width] dense */
  const void* img;
  float* out;             /* [rows, width] dense */
  const float* bias[3];   /* forward: b_l [width] (nullable) */
  float* save[2];         /* forward (optional): h_1, h_2;  backward (required): d_{nl-1}, d_{nl-2} */
  const float* gate[2];   /* backward: h_{nl-1}, h_{nl-2} */
  float* pre;             /* forward (optional): t;  backward (required): dt */
  const float* ln_in;     /* backward: t */
  const float* gamma;     /* [width] */
  const float* beta;      /* [width] (forward) */
  const float* rowscale;  /* optional [rows] */
  float* mean;            /* [rows]: forward out (optional), backward in */
  float* rstd;
  float* dgamma;          /* backward, optional [width], accumulated */
  float* dbeta;
  long rows;
  int width;              /* 256 | 320 */
  int nlayers;            /* 3 | 2 */
  int backward;
  float eps;
  int blocks;             /* 0 = one persistent block per CU (256) */
} FdNodeChainDesc;
int fd_node_chain(const FdNodeChainDesc* desc, void* stream);

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
