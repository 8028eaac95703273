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

#define FD_ABI_VERSION 1

const char* fd_last_error(void);
int fd_abi_version(void);
/* "gfx950" for the product library; "emu" for the test-only host interpreter. */
const char* fd_backend(void);

/* ---- dense: C = epi(alpha * A*B) --------------------------------------
 * Replaces torch Linear/matmul on the path: model/ipa_pytorch.py:101-166
 * (Linear), :169-233 (transitions), :334-374 (IPA projections), :380-386 and
 * :424-426 (qk^T, a*v); model/score_network.py:67-86 (embedder MLPs); and
 * their autograd.  A(m,k)=A[m*a_rs+k*a_cs], B(k,n)=B[k*b_rs+n*b_cs].
 * Batch index z in [0,batch): zo=z/bdiv, zi=z%bdiv; operand X is offset by
 * zo*x_so + zi*x_si.  Epilogue order: alpha, +bias[n], +pair_p/q, +resid,
 * relu, gate (zero where gate<=0), *rowscale[m], then C = v (+ C if beta). */
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
  int tile;              /* 0 = auto; 1: 128x128, 2: 64x64, 3: 128x32 */
  int ksplit;            /* >1: split K over blocks, C += alpha*A*B atomically
                            (weight gradients: tiny MxN, huge K); epilogue-free */
} FdGemmDesc;

int fd_gemm(const FdGemmDesc* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FD_HIP_H_ */
