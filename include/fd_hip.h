/* fd_hip.h -- C ABI of libfd_hip.so, the MI355X (gfx950) FrameDiff hot path.
 *
 * The reference (jasonkyuyim/se3_diffusion) has no FFI: its hot path is eager
 * PyTorch.  This header is the seam a maintainer binds (ctypes stub in
 * INTEGRATION.md) to route model/score_network.py:ScoreNetwork.forward and
 * data/se3_diffuser.py:SE3Diffuser.{calc_rot_score,calc_trans_score,reverse,
 * sample_ref,forward_marginal} through hand-written HIP kernels.
 *
 * Conventions: plain device pointers (fp32 unless stated), extents as int,
 * strides in ELEMENTS, `stream` is a hipStream_t passed as void*.  Every entry
 * point is asynchronous on `stream`, re-entrant per stream, takes no ownership,
 * and returns 0 on success or a negative FD_ERR_* code; fd_last_error() gives
 * the message of the last failure on the calling thread.
 */
#ifndef FD_HIP_H_
#define FD_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define FD_ABI_VERSION 2

const char* fd_last_error(void);
int fd_abi_version(void);
/* "gfx950" for the product library; "emu" for the test-only host interpreter. */
const char* fd_backend(void);
/* "" for a product build; "FD_PROBE_BUILD" when the library was built by tools/probes with the timing / ablation hooks of the
 * kernel sources enabled (csrc/fd_probe.h: those hooks do not compile in a product build). */
const char* fd_build_flags(void);
/* kernel launches this library has issued since it was loaded, including launches recorded into a hipGraph capture: the
 * difference across one captured diffusion step (experiments/train_se3_diffusion.py:746-781: one ScoreNetwork forward + one reverse
 * step) is the number of kernels a replay of that step runs. */
long fd_launch_count(void);

/* ---- dense: C = epi(alpha * A*B) --------------------------------------
 * Replaces torch Linear/matmul on the path: model/ipa_pytorch.py:101-166
 * (Linear), :169-233 (transitions), :334-374 (IPA projections), :380-386 and
 * :424-426 (qk^T, a*v); model/score_network.py:67-86 (embedder MLPs); and
 * their autograd.  A(m,k)=A[m*a_rs+k*a_cs], B(k,n)=B[k*b_rs+n*b_cs].
 * Batch index z in [0,batch): zo=z/bdiv, zi=z%bdiv; operand X is offset by
 * zo*x_so + zi*x_si.  Epilogue order: alpha, +bias[n], +pair_p/q, relu,
 * gate (zero where gate<=0), *rowscale[m], +resid, then C = v (+ C if beta). */
typedef struct FdGemmDesc {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long a_rs, a_cs;
  long b_rs, b_cs;
  long ldc;
  int batch, bdiv;
  long a_so, a_si, b_so, b_si, c_so, c_si;
  float alpha;
  int beta;
  const float* bias;     /* [N] */
  const float* pair_p;   /* [B*nres, ld_pair]: row (m / nres)            */
  const float* pair_q;   /* [B*nres, ld_pair]: row (m/nres^2)*nres+m%nres */
  long ld_pair;
  int nres;
  const float* resid;    /* [M, ld_resid] */
  long ld_resid;
  const float* gate;     /* [M, ld_gate] */
  long ld_gate;
  const float* rowscale; /* [M] */
  int relu;
  int tile;              /* 0 = auto; fp32 MFMA (bitwise an fmaf chain over k): 1: 128x128, 2: 64x64, 3: 128x32;
                            4: 256x128 split-bf16 (each fp32 operand = 3 exact bf16 terms, 6 bf16-MFMA products,
                            fp32 accumulate: fp32 accuracy, not bitwise the fmaf chain).  Auto picks 4 for the
                            large pair-level GEMMs unless FD_GEMM_EXACT_F32=1 is set in the environment.
                            5: latency kernel (fp32 MFMA, 32x32 tiles, K split over the waves of a block) that auto
                            picks when the problem has fewer 64x64 tiles than CUs (node-level GEMMs of sampling);
                            needs K % 8 == 0 and unit-stride 16-byte aligned operands.  A long K on few tiles (K >= 640 up
                            to 40 tiles, K >= 960 up to 80; K % 16 == 0) runs on 16x16 tiles: four times the blocks.
                            6: the split-bf16 kernel with a 128x128 block tile (two blocks per CU); never picked
                            automatically by fd_gemm (slower than 4 except for 128-row outputs), the host uses it
                            for the N_out = 128 weight gradients. */
  int ksplit;            /* >1: split K over blocks, C += alpha*A*B atomically
                            (weight gradients: tiny MxN, huge K); epilogue-free */
  int mtiles;            /* 0 = auto; >0: consecutive M tiles pipelined per block (un-batched, ksplit 1) */
  float* a_rowsum;       /* optional [M]: += alpha * sum_k A(m,k), accumulated atomically (fused bias gradient
                            of dW = dY^T X: A = dY^T, so this is sum over rows of dY) */
  const void* b_planes;  /* optional: B pre-split into its three exact bf16 planes (fd_split_planes of the buffer B lives in):
                            the address of plane 0's element of B[0]; plane p of B(k,n) is b_planes[p * b_plane_stride +
                            k * b_rs + n * b_cs] (16-bit elements).  With it fd_gemm may run the node-level layers (nn.Linear
                            forward y = x W^T and activation gradient dx = dy W, model/ipa_pytorch.py:101-166) on tiles
                            12: 128x128, 13: 64x128, 14: 64x64 -- split-bf16 arithmetic as tile 4, the weight operand moved as
                            6-byte elements with no split work in the kernel.  Needs A k-contiguous, K % 16 == 0, B unit-stride
                            along k or along n (then N % 8 == 0), un-batched, 16-byte aligned operands. */
  long b_plane_stride;   /* 16-bit elements between two planes (% 8 == 0) */
} FdGemmDesc;

int fd_gemm(const FdGemmDesc* desc, void* stream);
/* planes[p * n + e] = p-th bf16 term of x[e] (x = t0 + t1 + t2 exactly, round-to-nearest at every stage), p = 0..2: the
 * weight operand format of fd_gemm tiles 12-14.  One launch over the flat parameter buffer per optimiser step
 * (experiments/train_se3_diffusion.py:139: the parameters change once per step).  n % 8 == 0. */
int fd_split_planes(const float* x, long n, void* planes, void* stream);
/* the tile code (1..6) fd_gemm would run this descriptor with; no launch */
int fd_gemm_plan(const FdGemmDesc* desc);
/* exact != 0: every later fd_gemm runs on the fp32-MFMA kernels (as FD_GEMM_EXACT_F32=1); returns the previous mode */
int fd_gemm_set_exact_f32(int exact);
/* the split-bf16 kernel runs as `blocks` persistent blocks (default 256 = one per MI355X CU) that walk the output
 * tiles whenever a launch has at least 2 * blocks tiles; 0 = one fresh block per tile.  Returns the previous value. */
int fd_gemm_set_persistent_blocks(int blocks);

/* ---- LayerNorm / reductions (HBM-bound) -------------------------------
 * torch.nn.LayerNorm (eps 1e-5, biased variance) at score_network.py:73,85,
 * ipa_pytorch.py:189,231,577,632 and TransformerEncoderLayer.norm1/2; the
 * optional rowscale fuses the `* mask` that follows (ipa_pytorch.py:641,649). */
int fd_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, const float* rowscale,
                     float* y, long ldy, float* mean, float* rstd, long rows, int C, float eps, void* stream);
int fd_layernorm_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                     const float* rowscale, const float* mean, const float* rstd, float* dx, long lddx,
                     int dx_accum, float* dgamma, float* dbeta, long rows, int C, void* stream);
/* out[n] += sum_m X[m*ld+n] (bias gradients) */
int fd_colsum_acc(const float* X, long ld, long rows, int ncols, float* out, void* stream);
/* X [nbatch,n,n,C]: rowsum[b,i,:] += sum_j X[b,i,j,:], colsum[b,j,:] += sum_i X[b,i,j,:] (either may be null) */
int fd_pair_reduce_acc(const float* X, int nbatch, int n, int C, float* rowsum, float* colsum, long ld_out,
                       void* stream);
int fd_axpby(float* y, const float* x, float a, float b, long n, void* stream);
/* dst[r*ldd+c] += a*src[r*lds+c] */
int fd_add2d(float* dst, long ldd, const float* src, long lds, long rows, int cols, float a, void* stream);
int fd_rowscale(const float* x, long ldx, const float* rs, float* y, long ldy, long rows, int C, void* stream);

/* ---- edge transition, fused (model/ipa_pytorch.py:194-233 EdgeTransition.forward :218-233 and its autograd) ----
 * One kernel per direction for the whole 128 -> 384 -> 384 -> 128 chain of a pair row (se3_diffusion_amd/csrc/
 * fd_edge_mlp.hip): the hidden activations stay in the registers of the wave that owns the row, the weights stream
 * through LDS from a pre-packed bf16-plane image (fd_edge_mlp_pack, once per optimiser step, FD_EDGE_MLP_IMAGE_BYTES).
 * Split-bf16 arithmetic (fp32-accurate, as fd_gemm tile 4).  x [rows,128] = z.
 *   forward : h1 = relu(W1z x + p1[b,i] + q1[b,j]); h2 = relu(W2 h1 + bias2); y = Wf (h2 + [x | 0 | 0]) + pf[b,i] + qf[b,j]
 *             (W1z = W1[:, 0:128]; the residual through the final layer, ipa_pytorch.py:231, costs no product of its own);
 *             out = rowscale * LayerNorm(y; gamma, beta, eps).  Training outputs (all four or none): save1 = h1,
 *             save2 = h2 + [x | 0 | 0] (the operand of the final layer's weight gradient), mask1 / mask2 = the packed signs of
 *             h1 / h2 (the backward's ReLU gates); optionally y, mean, rstd of the LayerNorm.
 *   backward: x = dy; u = Wf^T x; d2 = [gmask1] u; d1 = [gmask2] (W2^T d2); out = u[0:128] + W1z^T d1, with the image packed
 *             from the transposed weights (fd_edge_mlp_pack_bwd; gmask1 / gmask2 = the forward's mask2 / mask1);
 *             save1 = d2, save2 = d1 (operands of the weight-gradient GEMMs), both required.
 * Optional fourth forward layer (zb_out != NULL): zb_out[rows,40] = W40 out + zb_bias, W40 = [linear_b.weight ; down_z.weight]
 * of the NEXT trunk block's IPA (ipa_pytorch.py:380-386,455) -- the pair bias and the down-projected pair features of its
 * attention, taken from the output while it is in registers; the image then carries four more units
 * (fd_edge_mlp_pack_zb, after fd_edge_mlp_pack). */
#define FD_EDGE_MLP_IMAGE_BYTES (124 * 12288)
int fd_edge_mlp_pack_zb(const float* W40, void* image, void* stream);
/* forward image: trunk.0.weight W1 [384,384], trunk.2.weight W2 [384,384], final_layer.weight Wf [128,384], row-major with row
 * stride ld (ipa_pytorch.py:204-216) */
int fd_edge_mlp_pack(const float* W1, const float* W2, const float* Wf, long ld, void* image, void* stream);
/* backward image (the transposed chain, hidden chunks in the order the backward kernel walks them); W40 [40,128] of the IPA block
 * BEHIND the transition or null: four leading units for the dzb term of the fused prologue (FdEdgeMlpDesc.dzb) */
int fd_edge_mlp_pack_bwd(const float* Wf, const float* W2, const float* W1, long ld, const float* W40, void* image, void* stream);
typedef struct FdEdgeMlpDesc {
  const float* x;        /* [rows,128] */
  const void* img;       /* weight image of fd_edge_mlp_pack */
  const float* p1;       /* forward: [B*nres,384] term of residue i */
  const float* q1;       /* forward: [B*nres,384] term of residue j (carries the layer-1 bias) */
  const float* bias2;    /* forward: [384] */
  float* save1;          /* [rows,384]: forward (training) h1; backward d2 */
  float* save2;          /* [rows,384]: forward (training) h2 + [x | 0 | 0]; backward d1 */
  const float* pf;       /* forward: [B*nres,128] */
  const float* qf;       /* forward: [B*nres,128] (carries the final bias) */
  const float* gamma;    /* forward: LayerNorm weight [128] */
  const float* beta;     /* forward: LayerNorm bias [128] */
  const float* rowscale; /* forward, optional: [rows] pair mask */
  float* y;              /* forward (training), optional: [rows,128] pre-LayerNorm values */
  float* mean;           /* forward (training), optional: [rows] */
  float* rstd;           /* forward (training), optional: [rows] */
  float* out;            /* [rows,128] */
  long rows;             /* B * nres * nres */
  int nres;
  int backward;
  float eps;
  int blocks;            /* 0 = persistent blocks that fill the 256 CUs (512 of shape 4, 256 of shape 8) */
  long ld_pq;            /* row stride of p1 / q1 (0 = 384) */
  long ld_pqf;           /* row stride of pf / qf (0 = 128) */
  float* zb_out;         /* forward, optional: [rows,40] (see above) */
  const float* zb_bias;  /* forward, optional: [40] */
  unsigned* mask1;       /* forward (training): [rows,12] packed signs of h1: bit 4 nb + e of word 4 c + g <-> unit 128 c + 16 nb + 4 g + e */
  unsigned* mask2;       /* forward (training): [rows,12] packed signs of h2 */
  const unsigned* gmask1; /* backward: the forward's mask2 (48 B per row instead of the 1536 B of h2) */
  const unsigned* gmask2; /* backward: the forward's mask1 */
  /* Backward, optional -- fused prologue (image from fd_edge_mlp_pack_bwd): x is then the UPSTREAM gradient of the transition's
   * output (may be null with dzb), the kernel's own input dy = LayerNorm-backward(x [+ dzb W40]; ln_y, ln_mean, ln_rstd,
   * ln_gamma, ln_rowscale) is formed in registers, written to dy_out (optional) and ln_dgamma / ln_dbeta (+=, optional). */
  const float* ln_y;        /* [rows,128] the forward's pre-LayerNorm save */
  const float* ln_mean;     /* [rows] */
  const float* ln_rstd;     /* [rows] */
  const float* ln_gamma;    /* [128] */
  const float* ln_rowscale; /* optional [rows] */
  float* dy_out;            /* optional [rows,128] */
  float* ln_dgamma;         /* optional [128], accumulated */
  float* ln_dbeta;          /* optional [128], accumulated */
  const float* dzb;         /* optional [rows,40]: adds dzb W40 to the upstream gradient (autograd of the next IPA block's linear_b /
                               down_z w.r.t. this transition's output; W40 = the four leading units of the image) */
  unsigned* sched;          /* optional: one word of device scratch -- a launch whose blocks walk four or more 64-row tiles each then
                               hands the tiles out dynamically (one atomic per tile and block); fd_edge_mlp zeroes the word on
                               the launch's stream; the words of launches that may run at the same time must differ */
  int shape;                /* 0 = by size (8 from FD_EDGE_MLP_W8_MIN_ROWS rows up; 2 for an inference forward of at most
                               FD_EDGE_MLP_PAIR_MAX_ROWS rows), 4 = 4 waves x 64-row tiles on two blocks per CU, 8 = 8 waves x 128-row
                               tiles on one block per CU (4 and 8: same results bit for bit), 2 = two waves per 16-row group, 64-row
                               tiles on one block per CU (inference forward only; equal to fp32 rounding) */
} FdEdgeMlpDesc;
#define FD_EDGE_MLP_W8_MIN_ROWS 65536L
#define FD_EDGE_MLP_PAIR_MAX_ROWS 16384L   /* 256 CUs x 4 SIMDs x 16 rows: at most one 16-row group per SIMD */
int fd_edge_mlp(const FdEdgeMlpDesc* desc, void* stream);

/* ---- edge embedder, fused (model/score_network.py:97-101,129-153 Embedder edge path, data/utils.py:570-580) ----
 * One kernel builds the 120-d pair feature on the fly and runs the MLP 120 -> 128 -> 128 -> 128 + LayerNorm + pair mask
 * in the registers of the wave that owns the pair row (se3_diffusion_amd/csrc/fd_edge_embed.hip).  The part of the first
 * layer that depends on one residue only arrives as p[b,i] = W0[:, 0:33] pt_i + b0 and q[b,j] = W0[:, 33:66] pt_j
 * (pt = [t-emb(32) | fixed]); the kernel adds W0[:, 66:120] [relpos sincos(32) | distogram(22)].  The image packs
 * W0[:, 66:120], W2, W4 ([128,120], [128,128], [128,128] row-major) as bf16 planes (FD_EDGE_EMBED_IMAGE_BYTES). */
#define FD_EDGE_EMBED_IMAGE_BYTES (24 * 12288)
/* optional fourth layer, as fd_edge_mlp's: zb_out[rows,40] = W40 out + zb_bias for the FIRST trunk block's IPA; the image then
 * carries four more units (fd_edge_embed_pack_zb after fd_edge_embed_pack) */
int fd_edge_embed_pack_zb(const float* W40, void* image, void* stream);
int fd_edge_embed_pack(const float* W0, const float* W2, const float* W4, void* image, void* stream);
typedef struct FdEdgeEmbedDesc {
  const long* seq_idx;    /* [B*nres] */
  const float* sc_ca;     /* [B*nres,3] self-conditioning CA positions (A) */
  const float* idenom;    /* [16] index-embedding denominators (host table, score_network.py:26-29) */
  const float* dg_lower;  /* [22] distogram bin edges (data/utils.py:573-578) */
  const float* dg_upper;  /* [22] */
  const void* img;
  const float* p;         /* [B*nres,128] */
  const float* q;         /* [B*nres,128] */
  const float* bias2;     /* [128] */
  const float* bias3;     /* [128] */
  const float* gamma;     /* LayerNorm weight [128] */
  const float* beta;      /* LayerNorm bias [128] */
  const float* rowscale;  /* optional [rows] pair mask */
  float* h1;              /* optional saves for the backward: [rows,128] post-ReLU layer 1 */
  float* h2;              /* ... layer 2 */
  float* h3;              /* ... layer 3 (pre-LayerNorm) */
  float* mean;            /* optional [rows] */
  float* rstd;            /* optional [rows] */
  float* out;             /* [rows,128] */
  long rows;              /* B * nres * nres */
  int nres;
  float eps;
  int blocks;             /* 0 = one persistent block per CU (256) */
  float* zb_out;          /* optional [rows,40] */
  const float* zb_bias;   /* optional [40] */
  unsigned* mask1;        /* optional [rows,4]: packed signs of h1 (bit 4 nb + e of word g <-> unit 16 nb + 4 g + e) */
  unsigned* mask2;        /* optional [rows,4]: packed signs of h2 */
  long ld_pq;             /* row stride of p / q (0 = 128): both as column blocks of one GEMM's output */
} FdEdgeEmbedDesc;
int fd_edge_embed(const FdEdgeEmbedDesc* desc, void* stream);

/* Backward dX chain of the same MLP (autograd of model/score_network.py:67-86,194-195 w.r.t. its hidden activations) in one
 * launch (se3_diffusion_amd/csrc/fd_edge_embed_bwd.hip): LayerNorm backward (dgamma / dbeta accumulated), dh2 = [h2 > 0]
 * (dh3 W4), dh1 = [h1 > 0] (dh2 W2).  h1, h2, h3, mean, rstd are fd_edge_embed's saves; dh3 / dh2 / dh1 [rows,128] are the
 * operands of the weight gradients (fd_group_dw).  Image: fd_edge_embed_bwd_pack(W2 = edge_embedder.2.weight, W4 = .4.weight). */
#define FD_EDGE_EMBED_BWD_IMAGE_BYTES (16 * 12288)
int fd_edge_embed_bwd_pack(const float* W2, const float* W4, void* image, void* stream);
typedef struct FdEdgeEmbedBwdDesc {
  const float* dy;        /* [rows,128] gradient of the embedder output */
  const float* h3;        /* [rows,128] pre-LayerNorm save */
  const float* mean;      /* [rows] */
  const float* rstd;      /* [rows] */
  const float* gamma;     /* LayerNorm weight [128] */
  const float* rowscale;  /* optional [rows] pair mask */
  const float* h2;        /* [rows,128] post-ReLU layer 2 */
  const float* h1;        /* [rows,128] post-ReLU layer 1 */
  const void* img;
  float* dh3;             /* [rows,128] out */
  float* dh2;
  float* dh1;
  float* dgamma;          /* optional [128], accumulated */
  float* dbeta;           /* optional [128], accumulated */
  long rows;
  int blocks;             /* 0 = two persistent blocks per CU (512) */
  const unsigned* gmask2; /* optional [rows,4]: fd_edge_embed's mask2 -- replaces the read of h2 (h2 may then be null) */
  const unsigned* gmask1; /* optional [rows,4]: fd_edge_embed's mask1 -- replaces the read of h1 */
  unsigned* sched;        /* optional: one word of device scratch for a dynamic tile hand-out (as FdEdgeMlpDesc.sched) */
} FdEdgeEmbedBwdDesc;
int fd_edge_embed_bwd(const FdEdgeEmbedBwdDesc* desc, void* stream);

/* ---- LayerNorm folded into the Linear that consumes it (se3_diffusion_amd/csrc/fd_ln_gemm.hip), sampling sizes (M <= ~1024) ----
 *   out[m, n] = epi( sum_k y[m, k] W[n, k] + bias[n] ),   y = ln_rowscale[m] * (LayerNorm(x[m, :]) * gamma + beta)
 *   epi: ReLU (relu != 0), then + resid[m, n].   ln_out (optional): y itself, for a later launch (a residual branch).
 * norm1 -> linear1 and norm2 -> the next consumer (self_attn.in_proj of the next layer, post_tfmr) of the sequence transformer's
 * TransformerEncoderLayer (post-norm; built at model/ipa_pytorch.py:584-595, post_tfmr :638).  LayerNorm arithmetic as
 * fd_layernorm_fwd (two-pass, eps inside the square root); exact fp32 products (v_mfma_f32_32x32x2_f32).
 * K % 8 == 0, K <= 320; x, W, gamma, beta, ln_out 16-byte aligned, row strides multiples of 4. */
typedef struct FdLnGemmDesc {
  const float* x;          /* [M, K], row stride ldx */
  long ldx;
  const float* gamma;      /* [K] */
  const float* beta;       /* [K] */
  const float* ln_rowscale;/* optional [M] */
  float* ln_out;           /* optional [M, K], row stride ld_ln_out */
  long ld_ln_out;
  const float* W;          /* [N, K], row stride ldw */
  long ldw;
  const float* bias;       /* optional [N] */
  const float* resid;      /* optional [M, N], row stride ld_resid */
  long ld_resid;
  float* out;              /* [M, N], row stride ldo */
  long ldo;
  int M, N, K;
  int relu;
  float eps;
  int ln_cols;             /* 0 = K; else only columns [0, ln_cols) of x are normalised (gamma / beta [ln_cols]), the rest pass through */
} FdLnGemmDesc;
int fd_ln_gemm(const FdLnGemmDesc* desc, void* stream);

/* ---- weight gradients of the pair-row MLPs, grouped (autograd of the Linear layers of EdgeTransition,
 * model/ipa_pytorch.py:194-233: dW = dY^T X with the B*N*N pair rows as the reduction index) ----
 * One launch for up to FD_PAIR_DW_MAX_ITEMS output tiles of 384 x 128 that share the row count
 * (se3_diffusion_amd/csrc/fd_pair_dw.hip):  C[m, n] += sum_p (A[p, m] + [m < 128] A_add[p, m]) * B[p, n].
 * Split-bf16 arithmetic (fp32-accurate, as fd_gemm tile 4); C accumulates atomically. */
#define FD_PAIR_DW_MAX_ITEMS 8
typedef struct FdPairDwItem {
  const float* A;       /* [rows, 384] ([rows, 128] when a_bands == 1), row stride lda: its columns index m */
  long lda;
  const float* A_add;   /* optional [rows, 128], row stride ld_add: added to columns 0..127 of A */
  long ld_add;
  const float* B;       /* [rows, 128] ([rows, b_cols] when b_cols > 0), row stride ldb: its columns index n */
  long ldb;
  float* C;             /* trans == 0: C[m * ldc + n] += ...; trans != 0: C[n * ldc + m] += ... */
  long ldc;
  float* a_colsum;      /* optional [384]: += sum_p A[p, :] (+ A_add) -- the bias gradient when A is dY */
  int trans;
  int a_bands;          /* 0 or 3: A has 384 columns; 1: A has 128 columns (a 128 x 128 tile; no trans / A_add) */
  int b_cols;           /* 0: B has 128 columns; else its column count (a multiple of 4, <= 128): C has b_cols columns */
} FdPairDwItem;
typedef struct FdPairDwDesc {
  FdPairDwItem item[FD_PAIR_DW_MAX_ITEMS];
  int nitems;
  long rows;            /* B * nres * nres */
  int blocks;           /* 0 = one persistent block per CU (256) */
} FdPairDwDesc;
int fd_pair_dw(const FdPairDwDesc* desc, void* stream);

/* ---- grouped node-level weight gradients (se3_diffusion_amd/csrc/fd_group_dw.hip): autograd of the per-residue nn.Linear
 * layers of a trunk block w.r.t. their weights and biases -- IPA's projections and linear_out (model/ipa_pytorch.py:236-301,
 * 455-460), skip_embed / the sequence transformer / post_tfmr (:584-595,632-638), StructureModuleTransition (:169-191),
 * the per-residue halves of EdgeTransition (:194-233), the torsion head (:560-582) and the node embedder
 * (model/score_network.py:57-66) -- as ONE launch instead of one split-K GEMM each:
 *   C_t[m * ldc_t + n] += sum_r A_t[r, m] * B_t[r, n]   (m < n_out_t, n < k_in_t, r < rows)      dW = dY^T X
 *   a_colsum_t[m]      += sum_r A_t[r, m]                (optional)                                db = sum_r dY
 * A_t = dY_t [rows, n_out_t] (row stride lda_t), B_t = X_t [rows, k_in_t] (row stride ldb_t); 16-byte aligned, strides,
 * n_out, k_in multiples of 4.  fp32 in / out, products as 3-term bf16 splits (fp32-accurate, as fd_gemm tile 4). */
#define FD_GROUP_DW_MAX_ITEMS 32
typedef struct FdGroupDwItem {
  const float* A;
  const float* B;
  float* C;
  float* a_colsum;
  int lda, ldb, ldc;
  int n_out, k_in;
} FdGroupDwItem;
typedef struct FdGroupDwDesc {
  FdGroupDwItem item[FD_GROUP_DW_MAX_ITEMS];
  int nitems;
  long rows;            /* B * nres */
  int blocks;           /* 0 = two persistent blocks per CU (512) */
} FdGroupDwDesc;
int fd_group_dw(const FdGroupDwDesc* desc, void* stream);

/* ---- sequence-transformer self-attention, fused (torch.nn.TransformerEncoderLayer.self_attn inside IpaScore,
 * model/ipa_pytorch.py:584-593; nhead 4, d_model 320): out = softmax(scale * q k^T + key_add) v per (batch, head) in one
 * launch (se3_diffusion_amd/csrc/fd_seq_attn.hip).  qkv [B*N, 960] = in_proj output [q | k | v]; key_add [B, N] additive
 * key mask or null; out [B*N, 320]; A_out [B, 4, N, N] optional (the probabilities, for the backward).  N <= 1024. */
int fd_seq_attn_fwd(const float* qkv, const float* key_add, float* out, float* A_out, float scale, int B, int N,
                    void* stream);
/* Backward of that attention in ONE launch (autograd of softmax(q k^T / sqrt(d) + mask) v, model/ipa_pytorch.py:584-593, with
 * respect to q, k, v): dqkv [B*N, 960] = [dQ | dK | dV] from qkv, the saved probabilities A [B,4,N,N], the saved output out [B*N,320]
 * and its gradient dout; replaces two batched fd_gemm launches, fd_row_softmax_bwd and two more batched fd_gemm launches; dA / dS
 * never reach HBM.  16-byte aligned tensors. */
int fd_seq_attn_bwd(const float* qkv, const float* A, const float* dout, const float* out, float* dqkv, float scale, int B, int N,
                    void* stream);

/* ---- embedder features: score_network.py:14-47,97-148; data/utils.py:570-580 ----
 * tscaled = (t*1e4) as fp32 [B]; tfreq[16], idenom[16], dg_lower[22], dg_upper[22] are the
 * host-computed tables of the reference's own op sequence. */
int fd_node_feats(const long* seq_idx, const float* tscaled, const float* fixed, const float* tfreq,
                  const float* idenom, float* out /*[B*N,65]*/, int B, int N, void* stream);
/* the same features with a row stride ld >= 65, columns 65 .. ld-1 written as zeros: K padded for the consumers' latency GEMM */
int fd_node_feats_ld(const long* seq_idx, const float* tscaled, const float* fixed, const float* tfreq,
                     const float* idenom, float* out /*[B*N,ld]*/, long ld, int B, int N, void* stream);
int fd_edge_feats(const long* seq_idx, const float* tscaled, const float* fixed, const float* sc_ca,
                  const float* tfreq, const float* idenom, const float* dg_lower, const float* dg_upper,
                  float* out /*[B*N*N,120]*/, int B, int N, void* stream);

/* ---- Invariant Point Attention, non-GEMM parts: ipa_pytorch.py:303-471 ----
 * layouts in se3_diffusion_amd/csrc/fd_ipa.hip.  Built for base.yaml dims
 * (no_heads 8, c_hidden 256, no_qk_points 8, no_v_points 12). */
/* kp_soa (may be null): a second copy of the key points as [B, 8, 24, n_res] (R = B * n_res rows), the layout the
 * attention kernels read with one lane per key */
int fd_ipa_points_fwd(const float* proj, const float* quat, const float* trans, float* qp, float* kp, float* vp,
                      float* kp_soa, int n_res, long R, int nheads, int c_hidden, int n_qk, int n_v, void* stream);
int fd_ipa_points_bwd(const float* proj, const float* quat, const float* dqp, const float* dkp, const float* dvp,
                      float* dproj, float* dframe, long R, int nheads, int c_hidden, int n_qk, int n_v,
                      void* stream);
int fd_ipa_softmax_fwd(float* S, const float* zb, const float* qp, const float* kp, const float* head_w,
                       const float* mask, int B, int N, void* stream);
/* hw_part: [B*N, 8] scratch (per-query-row partials of the head-weight gradient, column-summed into dhead_w) */
int fd_ipa_softmax_bwd(const float* A, float* dA, const float* qp, const float* kp, const float* head_w,
                       float* dzb, float* dqp, float* dkp, float* dhead_w, float* hw_part, int B, int N,
                       void* stream);
int fd_ipa_opt_fwd(const float* optg, const float* quat, const float* trans, float* feats, long R, void* stream);
int fd_ipa_opt_bwd(const float* dfeats, const float* feats, const float* quat, float* doptg, float* dframe,
                   long R, void* stream);
int fd_ipa_opair_fwd(const float* A, const float* zb, float* feats, int B, int N, void* stream);
int fd_ipa_opair_bwd(const float* A, const float* zb, const float* dfeats, float* dA, float* dzb, int B, int N,
                     void* stream);

/* softmax + o_pair of a query row in one launch (fd_ipa_softmax_fwd followed by fd_ipa_opair_fwd: the probabilities bit-identical,
 * o_pair equal to fp32 rounding -- up to 512 query rows the block is 512 threads, one wave per head, and sums o_pair in two halves
 * of the keys), and
 * their backward (fd_ipa_opair_bwd followed by fd_ipa_softmax_bwd): the probabilities / the updated dA stay in LDS.
 * kp_soa (may be null): fd_ipa_points_fwd's [B, 8, 24, N] copy of kp; when given, the key points are read from it */
int fd_ipa_attn_fwd(float* S, const float* zb, const float* qp, const float* kp, const float* kp_soa,
                    const float* head_w, const float* mask, float* feats, int B, int N, void* stream);
int fd_ipa_attn_bwd(const float* A, float* dA, const float* zb, const float* dfeats, const float* qp, const float* kp,
                    const float* kp_soa, const float* head_w, float* dzb, float* dqp, float* dkp, float* dhead_w,
                    float* hw_part, int B, int N, void* stream);

/* IPA attention of a block of the trunk in ONE launch (model/ipa_pytorch.py:380-457: logits :380-417, softmax :422,
 * o :424-428, o_pt + norm :432-449, o_pair :455-457) -- replaces  fd_gemm (q k^T) -> fd_ipa_attn_fwd -> fd_gemm (a v) ->
 * fd_gemm (a v_pts) -> fd_ipa_opt_fwd.  A block owns 16 query rows and heads_per_block heads (one wave each); keys walk in
 * tiles of 16 with a running maximum / denominator; the zb rows of the tile stream through LDS by LDS-DMA; q k^T, a v, a v_pts
 * and o_pair ride the exact-fp32 MFMA; the logits / probabilities never reach HBM.
 *   proj [R, 6816] (fd_ipa_points_fwd's input: q | kv | raw points), zb [B N N, 40], qp / kp [R, 8, 24] and vp [R, 8, 36]
 *   (global-frame points from fd_ipa_points_fwd), head_w [8], mask [R], quat [R, 4], trans [R, 3] ->
 *   feats [R, 2688] = [o 2048 | o_pt x,y,z 288 | |o_pt| 96 | o_pair 256]  (every column written).
 * A (may be null): [B, 8, N, N] receives the probabilities (training: the backward kernels read them).
 * heads_per_block: 8, 4, 2, or 0 = chosen from the number of query tiles.  All tensor arguments 16-byte aligned. */
int fd_ipa_flash_fwd(const float* proj, const float* zb, const float* qp, const float* kp, const float* vp,
                     const float* head_w, const float* mask, const float* quat, const float* trans, float* feats,
                     float* A, int B, int N, int heads_per_block, void* stream);
/* The same with the KEYS of a query tile split over key_splits blocks (inference on a lone backbone: B ceil(N/16) query tiles do
 * not fill 256 CUs): block ks walks the key tiles [ks nt / key_splits, (ks + 1) nt / key_splits) and leaves unnormalised sums
 * + (running maximum, denominator) per (split, residue, head) in `part` -- key_splits * B * N * 8 * FD_IPA_FLASH_PART_LD floats,
 * 16-byte aligned -- and a second launch (one block per residue) combines the splits and writes feats.  key_splits == 1 is
 * fd_ipa_flash_fwd; A must be null when key_splits > 1. */
#define FD_IPA_FLASH_PART_LD 328
int fd_ipa_flash_fwd_split(const float* proj, const float* zb, const float* qp, const float* kp, const float* vp,
                           const float* head_w, const float* mask, const float* quat, const float* trans, float* feats,
                           float* A, int B, int N, int heads_per_block, int key_splits, float* part, void* stream);

/* Query side of the IPA attention backward in ONE launch (autograd of model/ipa_pytorch.py:380-457 with the probabilities A
 * saved by the forward) -- replaces  fd_gemm (dA = dO V^T) -> fd_gemm (dA += dOpt vpts^T) -> fd_ipa_attn_bwd's per-row kernel;
 * dA never exists.  dL = A (dP - D) with dP = dO.v + dOpt.vpts + dout.zd on the exact-fp32 MFMA, D from the forward's outputs
 * (dO.o + dout.opair + ptdot) ->
 *   dL [B, 8, N, N] (written once: the dQ / dK GEMMs and the key-point gradient that follow read it),
 *   dzb [B N N, 40] = [sqrt(1/3) dL | sum_h A dout], dqp [R, 8, 24], dkp [R, 8, 24] (fd_ipa_kpts_bwd on dL), dhead_w [8] (+=;
 *   hw_part [R, 8] is scratch).
 * proj [R, 6816], A [B, 8, N, N], zb [B N N, 40], dfeats / feats [R, 2688], doptg [R, 8, 36] and ptdot [R, 8] from
 * fd_ipa_opt_bwd_dot, qp / kp [R, 8, 24], vp [R, 8, 36] (global frame), head_w [8], trans [R, 3].  16-byte aligned tensors. */
int fd_ipa_flash_bwd(const float* proj, const float* A, const float* zb, const float* dfeats, const float* feats,
                     const float* doptg, const float* ptdot, const float* qp, const float* kp, const float* vp,
                     const float* head_w, const float* trans, float* dL, float* dzb, float* dqp, float* dkp,
                     float* dhead_w, float* hw_part, int B, int N, void* stream);
/* KEY side of the same backward in one launch (se3_diffusion_amd/csrc/fd_ipa_flash.hip, ipa_flash_bwd_keys_kernel): autograd of
 * model/ipa_pytorch.py:380-457 with respect to keys / values and their points, from the probabilities A and the logit gradient dL
 * ([B,8,N,N] each, read once):  dproj[:, 2048 + 512 h + 256 ..] = dV = A^T dO;  dproj[:, 2048 + 512 h ..] = dK = sqrt(1/3C) dL^T Q;
 * dvp [R,8,36] = A^T doptg;  dkp [R,8,24] = gamma_h sum_i dL_ij (qp_i - kp_j).  Replaces three batched fd_gemm launches and
 * fd_ipa_kpts_bwd (pass dkp = NULL to fd_ipa_flash_bwd to skip its own fd_ipa_kpts_bwd).  heads_per_block: 0 = by size, 2 / 4 / 8. */
int fd_ipa_flash_bwd_keys(const float* A, const float* dL, const float* proj, const float* dfeats, const float* doptg,
                          const float* qp, const float* kp, const float* head_w, float* dproj, float* dvp, float* dkp,
                          int B, int N, int heads_per_block, void* stream);
/* fd_ipa_opt_bwd that also returns ptdot [R, 8] = sum_p d(o_pt, global frame) . (o_pt, global frame) per head (the o_pt term
 * of the softmax backward's row constant; model/ipa_pytorch.py:432-449) */
int fd_ipa_opt_bwd_dot(const float* dfeats, const float* feats, const float* quat, const float* trans, float* doptg,
                       float* dframe, float* ptdot, long R, void* stream);

/* dz[p, 0:128] (+)= dzb[p, 0:40] W40[0:40, 0:128] over the pair rows (autograd of linear_b / down_z w.r.t. z,
 * ipa_pytorch.py:380-386,455-457): streaming kernel, W40 resident in registers */
int fd_ipa_dz_acc(const float* dzb, const float* W40, float* dz, long rows, int accumulate, void* stream);

/* dkp[b,j,h,:] = gamma_h sum_i dLogits[b,h,i,j] (qp[b,i,h,:] - kp[b,j,h,:]) (the key-side point gradient) */
int fd_ipa_kpts_bwd(const float* dL, const float* qp, const float* kp, const float* head_w, float* dkp, int B, int N,
                    void* stream);

/* ---- sequence-transformer softmax (nn.MultiheadAttention, ipa_pytorch.py:584-593) ---- */
int fd_row_softmax_fwd(float* S, const float* key_add, long rows, int N, int rows_per_batch, void* stream);
int fd_row_softmax_bwd(const float* A, float* dA, long rows, int N, void* stream);

/* ---- input frames: rigids_t [R,7] -> quat [R,4], trans [R,3] * scale (scale_rigids, model/score_network.py:190-193) and,
 * optionally, tscaled[b] = t[b] * tscale for the B examples (the embedder's timestep argument, score_network.py:38,43) ---- */
int fd_split_rigids(const float* rig7, float scale, const float* t, float tscale, float* quat, float* trans, float* tscaled,
                    long R, int B, void* stream);

/* ---- backbone update: ipa_pytorch.py:530-557,641-644; rigid_utils.py:266-275,587-616,1039-1063 ---- */
int fd_bb_update_fwd(const float* node, long ldn, int cs, const float* dmask, const float* W6, const float* b6,
                     const float* quat, const float* trans, float* upd, float* quat_out, float* trans_out,
                     long R, void* stream);
int fd_bb_update_bwd(const float* dquat_out, const float* dtrans_out, const float* dframe, const float* dmask,
                     const float* upd, const float* quat, float* dquat, float* dtrans, float* dupd, float* dupd_s,
                     long R, void* stream);

/* ---- score heads + psi + backbone atoms: ipa_pytorch.py:650-672, se3_diffuser.py:115-125,
 * so3_diffuser.py:9-117,182-213,274-305, r3_diffuser.py:42-43,148-166, all_atom.py:152-174 ---- */
typedef struct FdHeadConst {
  float atoms[15];       /* N, CA, C, CB, O idealised local coordinates (ALA) */
  float Rd[9];           /* psi rigid-group default frame rotation */
  float td[3];           /* ... translation */
  float coord_scale;     /* 0.1 */
  double exp_max_sigma, exp_min_sigma;
  float min_b, max_b;
  int L;                 /* IGSO(3) truncation (1000) */
  /* so3.use_cached_score=True (config/icml_published.yaml; so3_diffuser.py:293-299): device table [ng, n_omega] of
   * score norms + its omega grid [n_omega]; the rotation score is then the bucketised lookup instead of the series
   * (and carries no gradient through the lookup, like torch.gather of a constant).  NULL = series. */
  const double* score_norms;
  const double* omega_grid;
  int n_omega;
} FdHeadConst;
int fd_heads_fwd(const float* rig0, const float* quatF, const float* transF, const float* upsi,
                 const float* gt_psi, long gt_stride, const float* fixed, const float* mask, const float* t,
                 const double* sigma_grid, int ng, const FdHeadConst* c, double* rot_score, float* trans_score,
                 float* rigids, float* psi_out, float* atom37, float* atom14,
                 float* sc_ca_out /* optional [R,3]: the predicted CA positions (= rigids[..., 4:7]) once more, where the sampling
                 loop keeps its self-conditioning input (experiments/train_se3_diffusion.py:763-765) */,
                 int B, int N, void* stream);
/* data/all_atom.py:152-174 compute_backbone as one kernel: rigids [R,7] (A), psi [R,2] (sin,cos) */
int fd_backbone_atoms(const float* rigids, const float* psi, const FdHeadConst* c, float* atom37, float* atom14,
                      long R, void* stream);
int fd_heads_bwd(const float* rig0, const float* quatF, const float* transF, const float* upsi,
                 const float* psi_out, const float* fixed, const float* mask, const float* t,
                 const double* sigma_grid, int ng, const FdHeadConst* c, const double* d_rot,
                 const float* d_trans_score, const float* d_rigids, const float* d_psi, const float* d_atom37,
                 float* dquatF, float* dtransF, float* dupsi, int B, int N, void* stream);

/* ---- SE(3) diffuser (fp64 arithmetic, fp32 [..,7] frames in Angstrom) ----------------
 * Random draws are inputs in the reference's call order, so identical noise gives identical frames.
 * fd_igso3_tables     data/so3_diffuser.py:122-180   pdf/cdf/score_norms [ns, no] (8 MB each at 1000^2)
 * fd_sample_ref       data/se3_diffuser.py:216-268   prior: IGSO3(t=1) x N(0, I) (scaled units -> A)
 * fd_forward_marginal data/se3_diffuser.py:43-110    noised frames + DSM score targets at time t
 * fd_se3_reverse_step data/se3_diffuser.py:160-214   one Euler-Maruyama / geodesic-random-walk step */
int fd_igso3_tables(const double* sigma, const double* omega, int ns, int no, int L, double* pdf, double* cdf,
                    double* score_norms, void* stream);
int fd_sample_ref(const double* z_axis, const double* u, const double* z_trans, const double* cdf_row,
                  const double* omega, int no, double coord_scale, float* out, long n, void* stream);
int fd_forward_marginal(const float* rig0, const double* z_axis, const double* u, const double* z_trans,
                        const double* cdf_row, const double* omega, int no,
                        const double* score_row /* optional: this sigma's score_norms row (use_cached_score) */,
                        double sigma, double beta, double coord_scale, int L, const float* mask, float* rig_t,
                        double* rot_score, double* trans_score, long n, void* stream);
/* a training batch in one launch (pdb_data_loader.py:240-262 calls forward_marginal once per example, each with its
 * own t): cdf / score_norms are the full [ns, no] tables, tparams [B,3] (device, fp64) = {sigma bin = t_to_idx(t_b),
 * discrete sigma of that bin, marginal_b_t(t_b)}; everything else is [B,N,...] as above. */
int fd_forward_marginal_batch(const float* rig0, const double* z_axis, const double* u, const double* z_trans,
                              const double* cdf, const double* omega, int no,
                              const double* score_norms /* optional, use_cached_score */, const double* tparams,
                              double coord_scale, int L, const float* mask, float* rig_t, double* rot_score,
                              double* trans_score, int B, int N, void* stream);
int fd_se3_reverse_step(const float* rig_t, const double* rot_score, const double* trans_score,
                        const double* z_rot, const double* z_trans, const float* mask, int B, int N, double g_rot,
                        double b_t, const double* tparams /* optional device {g_rot, b_t}: overrides the scalars,
                        lets one captured hipGraph serve every t */, double dt, double noise_scale,
                        double coord_scale, int center, int diffuse_rot, int diffuse_trans, float* out, void* stream);
/* the same step with the scores as the network stores them (float32, widened in registers: the same numbers the reference
 * passes after its .to(float64)); out may be rig_t (in place) in both forms */
int fd_se3_reverse_step_f32(const float* rig_t, const float* rot_score, const float* trans_score,
                        const double* z_rot, const double* z_trans, const float* mask, int B, int N, double g_rot,
                        double b_t, const double* tparams /* optional device {g_rot, b_t}: overrides the scalars,
                        lets one captured hipGraph serve every t */, double dt, double noise_scale,
                        double coord_scale, int center, int diffuse_rot, int diffuse_trans, float* out, void* stream);
/* ... and as ScoreNetwork.forward returns them: rot_score float64, trans_score float32 */
int fd_se3_reverse_step_net(const float* rig_t, const double* rot_score, const float* trans_score,
                        const double* z_rot, const double* z_trans, const float* mask, int B, int N, double g_rot,
                        double b_t, const double* tparams, double dt, double noise_scale,
                        double coord_scale, int center, int diffuse_rot, int diffuse_trans, float* out, void* stream);
/* The per-step host glue of the reverse loop (experiments/train_se3_diffusion.py:746-781: t = reverse_steps[i], the draws of
 * diffuser.reverse) as the FIRST NODE of a captured step: idx = *counter; t_out[0..B) = all_t[idx]; tparams[0..2) =
 * all_tp[2 idx ..] ({g_rot(t), b(t)} of fd_se3_reverse_step); z_out[0..nz) = z_all[(idx % K) nz ..] (K steps of normal draws,
 * rotation block then translation block per step, refilled by the caller every K steps); *counter = idx + 1.  One block. */
int fd_sample_advance(int* counter, const float* all_t, const double* all_tp, const double* z_all, int K, long nz, float* t_out,
                      int B, double* tparams, double* z_out, void* stream);

/* ---- training loss (the step next to the hot path): experiments/train_se3_diffusion.py:524-693 ----
 * Experiment.loss_fn, both rotation branches: translation score / x0 loss, rotation axis + angle (or joint MSE) loss,
 * backbone-atom loss and the 5N x 5N distance-matrix loss with their per-example normalisers and t filters, value AND
 * gradient w.r.t. the network outputs in three launches.  loss[0] = sum_b final_b / #non-empty examples.
 * terms[b] = {trans_score_loss, trans_x0_loss, axis_loss, angle_loss (weighted), bb_atom_loss (weighted),
 * dist_mat_loss (weighted), final_b, sum of the loss mask}.  scratch: B*N*15 + 2*B floats. */
typedef struct FdLossDesc {
  int B, N;
  const float* res_mask;            /* [B,N] */
  const float* fixed_mask;          /* [B,N] */
  const float* t;                   /* [B]   */
  const float* gt_trans_score;      /* [B,N,3] */
  const double* gt_rot_score;       /* [B,N,3] */
  const float* trans_score_scaling; /* [B] */
  const float* rot_score_scaling;   /* [B] */
  const float* gt_rigids;           /* [B,N,7] rigids_0 */
  const float* gt_atom37;           /* [B,N,37,3] */
  const double* rot_score;          /* network outputs: [B,N,3] fp64 */
  const float* trans_score;         /* [B,N,3] */
  const float* rigids;              /* [B,N,7] */
  const float* atom37;              /* [B,N,37,3] */
  float coordinate_scaling, trans_x0_threshold, trans_loss_weight, rot_loss_weight, rot_loss_t_threshold;
  float bb_atom_loss_weight, bb_atom_loss_t_filter, aux_loss_weight, dist_mat_loss_weight, dist_mat_loss_t_filter;
  double* d_rot_score;              /* gradients of loss[0], same shapes as the outputs */
  float* d_trans_score;
  float* d_rigids;
  float* d_atom37;
  float* terms;                     /* [B,8] */
  float* loss;                      /* [1] */
  float* scratch;
  int joint_rot_loss;               /* 0: separate_rot_loss=True (axis + angle, base.yaml); 1: joint rot-score MSE
                                       (train_se3_diffusion.py:597-604, icml_published.yaml), reported in terms[:,3] */
} FdLossDesc;

int fd_dsm_loss(const FdLossDesc* desc, void* stream);

/* Adam (torch.optim.Adam defaults, experiments/train_se3_diffusion.py:139) over flat fp32 buffers of n elements
 * (n % 4 == 0, 16-byte aligned): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps) with bc_i = 1 - b_i^t supplied by the host. */
int fd_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                 float bc1, float bc2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FD_HIP_H_ */
