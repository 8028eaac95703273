// Edge transition (model/ipa_pytorch.py:194-233) as ONE kernel per direction: the 128 -> 384 -> 384 -> 128 chain of a
// pair row never leaves the CU.
//
//   forward :  h1 = relu(W1z z + P1_i + Q1_j)      h2 = relu(W2 h1 + b2)      y = Wf (h2 + [z | 0 | 0]) + Pf_i + Qf_j
//              z' = mask * LayerNorm(y)                          (P/Q: the node halves of W [z | e_i | e_j], trunk.py)
//   backward:  u = Wf^T dy        d2 = [h2 > 0] u        d1 = [h1 > 0] W2^T d2        dz = u[0:128] + W1z^T d1
//              (the same dataflow with transposed weights and ReLU gates instead of bias + ReLU)
//
// The residual through the final layer (reference: final_layer(trunk(x) + x), ipa_pytorch.py:231) costs NO product of its own
// (round 5; rounds 2-4 ran Wf[:, :128] z and its transpose as an eight-unit region per direction = 6.25 % of the MFMAs and of
// the weight stream): the input z is held in the registers in layer-OUTPUT layout, so the forward adds it to the first 128
// units of h2 before the last layer, and the backward takes the z gradient's first term from the ungated accumulator of the
// first layer's chunk that covers hidden units 0..127 (that chunk is processed LAST, so the accumulator is simply kept).
//
// Arithmetic: the split-bf16 scheme of fd_gemm_split.h -- every fp32 operand is three exact bf16 terms, six bf16 MFMA
// products per k-step, fp32 accumulation: fp32-accurate (not bitwise an fmaf chain).
//
// Register-chained layout.  A wave owns 16 pair rows for the whole chain and accumulates TRANSPOSED
// (D[n][m] = sum_k W[n][k] x[m][k]: the weights are the MFMA's row operand, v_mfma_f32_16x16x32_bf16), so lane
// (m = l & 15, g = l >> 4) ends up with hidden units n = 16 nb + 4 g + r of ITS OWN row.  Two consecutive 16-blocks of
// those registers are exactly a B-operand fragment of the next GEMM's 32-k step if the next layer's weights are stored
// with the k order permuted to match (slot (g, e') <-> k = 16 (e' >> 2) + 4 g + (e' & 3)): activations go accumulator
// -> relu -> bf16 split -> MFMA operand without touching LDS, without a barrier, without leaving the wave.  The kernel's
// input is loaded in the same layout (a lane reads columns 16 nb + 4 g .. + 3 of its row).  LDS only
// streams the weights, packed ONCE per optimiser step (fd_edge_mlp_pack) into bf16-plane fragments in the exact order
// the kernel consumes them: staging is a straight LDS-DMA copy (global_load_lds_dwordx4, no VGPRs, no VALU) and every
// fragment read a conflict-free ds_read_b128 of a lane-linear 1 KB piece (PMC: SQ_LDS_BANK_CONFLICT = 0).
//
// Per tile the weight stream is 120 units of 12 KB (4 n-blocks of 16 x one 32-k step x 3 planes) through a two-stage LDS ring;
// stage s+1 is in flight while stage s is multiplied:
//     for c in 0..2:   8 units  W1z[n in chunk c]        (layer 1, K = 128)      -> h1 chunk c (32 registers)
//                     24 units  W2[all n][k in chunk c]  (layer 2, partial K)    -> acc2 (96 registers) += ...
//     24 units Wf                                        (layer 3, K = 384)
// (backward: the chunks in the order 1, 2, 0 of the hidden units -- see above.)
// Algorithmic HBM bytes per pair row: 512 read + 512 written (PMC: 1.1 KB) (+ h1, h2 + z, y saved for the backward in
// training).
// Two shapes of the same kernel are built (round 4):
//   fd_edge_mlp.hip itself      4 waves x 64-row tiles, two blocks per CU, 24 KB stages (two units): launches with
//                               few tiles per CU (single backbones)
//   fd_edge_mlp_w8.hip          8 waves x 128-row tiles, ONE block per CU, 48 KB stages (four units): half the barriers
//                               per MFMA and one weight stream per CU instead of two (half the L2 -> LDS traffic): -5 ... -8 % per
//                               launch from 65,536 pair rows up (profiles/r04_edge_variants_*.log)
// (that file defines EM_SHAPE_W8 + the shape macros and includes this one; the pack kernels and the C entry points live here only)
// Shape history (4 waves x 32 rows on 32x32x16, deeper rings, stepped per-residue terms, lumped saves): DESIGN.md
// section 6 -- the variants that lost are no longer in this source.
#if defined(EM_PHASE_TIMING) || defined(EM_ABLATE_PQ) || defined(EM_PLAIN_SAVES)
#include "fd_probe.h"      // timing / ablation hooks: tools/probes builds only (-DFD_PROBE_BUILD)
#endif
#include "fd_common.h"
#include "../../include/fd_hip.h"

#ifdef EM_SHAPE_W8
#define EM_LAUNCH fd_edge_mlp_launch_w8
#else
#define EM_LAUNCH fd_edge_mlp_launch_w4
#endif
int fd_edge_mlp_launch_w4(const FdEdgeMlpDesc& d, hipStream_t st);
int fd_edge_mlp_launch_w8(const FdEdgeMlpDesc& d, hipStream_t st);
int fd_edge_mlp_launch_pair(const FdEdgeMlpDesc& d, hipStream_t st);      // csrc/fd_edge_mlp_pair.hip

namespace {

#include "fd_chain.h"

__device__ __forceinline__ float fd_f4(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

constexpr int EM_UNITS = 120;              // units per tile
constexpr int EM_ZB_UNITS = 4;             // + the next IPA block's [linear_b ; down_z] (40 <- 128: 4 k-steps x one n-group)
constexpr int EM_ZB = 40;
// LDS ring of the weight stream: two stages, the copy of stage s + 1 in flight while stage s is multiplied (deeper rings measured
// in rounds 2 and 4: no gain on the one-block-per-CU shape, and 144 KB per CU cost the step its overlap with the gradient stream)
constexpr int EM_RING = 2;
constexpr int EM_H = 384, EM_C = 128;
// The training saves (h1 / h2z forward, d2 / d1 backward: 3 KB per pair row that only the weight-gradient launch reads, much later)
// are stored with the NON-TEMPORAL hint (round 5): plain stores pushed the input rows that the forward re-reads per k-step out of the
// XCD's L2.  Same box, variant libraries swapped (profiles/r05_ab.txt): training forward launch 1.60-1.62 against 1.69-1.80 ms,
// average fused launch of the step 1.310 against 1.346 ms, training step 21.65 against 21.85 ms.  The other outputs (y, dy, z', dz)
// measured the same with either policy and keep plain stores.  (Rounds 2 and 4 tried the hint when nothing was re-read: neutral.)
// -DEM_PLAIN_SAVES (probe build) restores the plain stores for that A/B.
#ifdef EM_PLAIN_SAVES
#define EM_SAVE4(p, a, b, c, e) (*reinterpret_cast<float4*>(p) = make_float4((a), (b), (c), (e)))
#else
#define EM_SAVE4(p, a, b, c, e) fd::store_nt4((p), (a), (b), (c), (e))
#endif

// Probe build only (tools/probes/edge_phases.py compiles this file with -DEM_PHASE_TIMING into its own library): wave-level
// cycle counts per phase of a tile, read with s_memtime at stage boundaries (where no LDS read is outstanding) and summed over
// all waves into em_phase[]: 0 tile prologue, 1 stage waits (vmcnt + barrier), 2 layer-1 stages, 3 epilogue 1, 4 layer-2 stages,
// 5 epilogue 2, 6 layer-3 stages, 7 final epilogue (backward), 8 final epilogue + fourth layer (forward), 9 fused backward prologue.
#ifdef EM_PHASE_TIMING
__device__ unsigned long long em_phase[12];
#define EM_TICK_TO(var)                                                \
  {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const long long t_ = (long long)__builtin_amdgcn_s_memtime();      \
    __builtin_amdgcn_sched_barrier(0);                                 \
    var += t_ - tl_;                                                   \
    tl_ = t_;                                                          \
  }
#define EM_BODY_TO(i) { EM_TICK_TO(tb_); em_add(i, tb_); tb_ = 0; }
#else
#define EM_TICK_TO(var)
#define EM_BODY_TO(i)
#endif

#ifndef EM_SHAPE_W8
struct EmMat {
  const float* p;
  long rs, cs;
};

// weight image: 120 units in consumption order; a unit = [4 n-blocks of 16][3 planes][64 lanes] x 16 B for one 32-k
// step; the fragment of lane (n = l & 15, g = l >> 4) holds slots e' = 0..7 in CHAINED k order (every operand of the chain,
// the kernel's input included, sits in the registers in layer-output layout): k = k0 + 16 (e' >> 2) + 4 g + (e' & 3).
// Unit order inside a region: k-step major, n-group minor (the activation planes of a k-step are split once and reused by its
// n-groups).  A1 [384,128] layer 1, A2 [384,384] layer 2, A4 [128,384] layer 3.
// rot: the image's chunk c holds the hidden units of chunk (c + rot) % 3 (backward: 1 -- hidden units 0..127 last, see the top)
__global__ __launch_bounds__(256) void edge_mlp_pack16_kernel(EmMat A1, EmMat A2, EmMat A4, char* __restrict__ img, int rot) {
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (unit, n-block, lane)
  if (gid >= EM_UNITS * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, u = gid >> 8;
  const int m = lane & 15, g = lane >> 4;
  EmMat M;
  int n, k0;
  if (u < 96) {
    const int c = u / 32, r = u % 32, hc = (c + rot) % 3;
    if (r < 8) {                    // layer 1: k-step r >> 1, n-group r & 1 of hidden chunk hc
      M = A1; n = 128 * hc + 64 * (r & 1) + 16 * i + m; k0 = 32 * (r >> 1);
    } else {                        // layer 2: k-step (r - 8) / 6 of hidden chunk hc, n-group (r - 8) % 6
      const int r2 = r - 8;
      M = A2; n = 64 * (r2 % 6) + 16 * i + m; k0 = 128 * hc + 32 * (r2 / 6);
    }
  } else {                          // layer 3
    const int v = u - 96;
    M = A4; n = 64 * (v & 1) + 16 * i + m; k0 = 32 * (v >> 1);
  }
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + 16 * (e >> 2) + 4 * g + (e & 3);
    x[e] = M.p[(long)n * M.rs + (long)k * M.cs];
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)u * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

// units 120..123 of the forward image: W40 = [linear_b.weight ; down_z.weight] [40,128] of the NEXT trunk block's IPA
// (ipa_pytorch.py:380-386,455), rows 40..63 zero, chained k order (its operand is the LayerNorm output in registers)
__global__ __launch_bounds__(256) void edge_mlp_pack_zb_kernel(const float* __restrict__ W40, char* __restrict__ img) {
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (k-step, n-block, lane)
  if (gid >= EM_ZB_UNITS * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, ks = gid >> 8;
  const int m = lane & 15, g = lane >> 4;
  const int n = 16 * i + m;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * ks + 16 * (e >> 2) + 4 * g + (e & 3);
    x[e] = n < EM_ZB ? W40[n * EM_C + k] : 0.f;
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)(EM_UNITS + ks) * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

// four LEADING units of a backward image with the dzb W40 prologue: W40^T -- n = the 128 columns of z (two n-groups), k = the 40
// columns of dzb in natural order (operand loaded from memory), zero beyond k = 40; unit = (k-step, n-group), n-group minor
__global__ __launch_bounds__(256) void edge_mlp_pack_zbw_kernel(const float* __restrict__ W40, char* __restrict__ img) {
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (unit, n-block, lane)
  if (gid >= EM_ZB_UNITS * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, u = gid >> 8;
  const int m = lane & 15, g = lane >> 4;
  const int n = 64 * (u & 1) + 16 * i + m, k0 = 32 * (u >> 1);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + 8 * g + e;
    x[e] = k < EM_ZB ? W40[k * EM_C + n] : 0.f;
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)u * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

#endif  // !EM_SHAPE_W8

// ZB (forward only): a fourth chained layer on the kernel's own output -- zb = [linear_b ; down_z] z' + b40 of the next
// trunk block's IPA -- so that block needs no pass over z' [P,128] for it (fd_gemm: 119 us per block at B=30 x N=128, 252 MB
// read); +3 % of the chain's MFMAs, 160 B more written per pair row
// TRAIN (forward): the saves of the backward -- save1 = h1, save2 = h2 + [z | 0 | 0] (the operand of the final layer's weight
// gradient; the ReLU gates of the backward come from the masks, so nobody needs h2 alone), the packed signs of h1 / h2 (mask1 /
// mask2), optionally y / mean / rstd of the LayerNorm.  The saves leave the wave two 16-blocks at a time where the next layer
// splits them.  The backward always saves (save1 = d2, save2 = d1: the operands of the weight gradients) and always gates on
// the packed masks.
// LNB (backward): the kernel's input dy is formed HERE from the upstream gradient of the transition's output -- the LayerNorm
// backward of ipa_pytorch.py:232 (dgamma / dbeta accumulated in LDS, one atomic per column and block).
// ZBW (with LNB): and the upstream gradient first gets the IPA pair-projection term  dz += dzb W40  of the block behind this
// transition (autograd of linear_b / down_z w.r.t. z, ipa_pytorch.py:380-386,455) as a K = 40 product on four leading units.
template <bool BWD, bool ZB = false, bool TRAIN = false, bool LNB = false, bool ZBW = false>
__global__ __launch_bounds__(64 * EM_WAVES, 2) void edge_mlp16_kernel(FdEdgeMlpDesc d) {
  static_assert(!BWD || (!ZB && TRAIN), "backward: no zb layer, always with the saves");
  static_assert(BWD || (!LNB && !ZBW), "the fused prologue is a backward option");
  constexpr int EM_NSTAGE = (EM_UNITS + (ZB ? EM_ZB_UNITS : 0) + (ZBW ? EM_ZB_UNITS : 0)) / EM_UPS;
  __shared__ __attribute__((aligned(16))) char lds[EM_RING * EM_STAGE];
  __shared__ float lnacc[LNB ? 2 * EM_C : 1];      // dgamma | dbeta of the fused LayerNorm backward
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const long rows = d.rows;
#ifdef EM_PHASE_TIMING
  long long p0_ = 0, p1_ = 0, p2_ = 0, p3_ = 0, p4_ = 0, p5_ = 0, p6_ = 0, p7_ = 0, p8_ = 0, p9_ = 0;
  long long tb_ = 0, tl_ = (long long)__builtin_amdgcn_s_memtime();
  auto em_add = [&](int i, long long v) {       // (i is a literal at every call site)
    switch (i) {
      case 0: p0_ += v; break;
      case 2: p2_ += v; break;
      case 3: p3_ += v; break;
      case 4: p4_ += v; break;
      case 5: p5_ += v; break;
      case 6: p6_ += v; break;
      case 7: p7_ += v; break;
      case 8: p8_ += v; break;
      default: p9_ += v; break;
    }
  };
#endif
  // row strides of the per-residue terms (0 = dense [B*nres, 384] / [B*nres, 128]; a caller that forms all four with one
  // GEMM passes the width of that GEMM's output)
  const long ld_pq = d.ld_pq > 0 ? d.ld_pq : EM_H, ld_pqf = d.ld_pqf > 0 ? d.ld_pqf : EM_C;
  const int ntiles = (int)((rows + EM_ROWS - 1) / EM_ROWS);
  const int G = (int)gridDim.x, first = (int)blockIdx.x;
  if (first >= ntiles) return;
  // Tiles: block b starts with tile b; after that either the static stride (b + G, b + 2 G, ...) or -- d.sched, when a launch has
  // more tiles than blocks -- the next tile nobody has taken (one atomic per tile and block).  With the static stride a block that
  // becomes resident late (another stream's kernel still holds its CU's LDS when the launch starts) finishes its equal share late
  // and the whole launch waits for it: in the training step the backward ran 1.5 ms alone and 1.9 ms behind a 0.3 ms grouped
  // weight-gradient launch of the side stream.
  unsigned* const sched = d.sched;      // (the host entry zeroes the word in front of every launch that uses it)
  const bool dyn = sched != nullptr;
  __shared__ int s_tile;
  const int nmine = (ntiles - first + G - 1) / G;
  const int total_stages = dyn ? 0x7fffffff : nmine * EM_NSTAGE;   // (dyn: the weight stream keeps one stage ahead to the end)

  // ---- weight stream: every wave copies its share (EM_PPW pieces) of each stage; stage s lives in buffer s & 1 ----
  const char* __restrict__ img_lane = static_cast<const char*>(d.img) + wave * (EM_STAGE / EM_WAVES) + lane * 16;
  char* const lds_wave = lds + wave * (EM_STAGE / EM_WAVES);
  int issued = 0, consumed = 0;
  auto issue_stage = [&]() __attribute__((always_inline)) {
    const char* src = img_lane + (long)(issued % EM_NSTAGE) * EM_STAGE;
    char* dst = lds_wave + (issued % EM_RING) * EM_STAGE;
    if (EM_PPW == 6) {
      fd::glds16x4(src, dst);
      fd::glds16x2(src + 4096, dst + 4096);
    } else {
      fd::glds16x3(src, dst);
    }
    ++issued;
  };
  // begin the next stage: its copy (issued one stage ago) has landed and is visible to the block; every wave is done
  // with the previous stage, whose buffer takes the copy after next.  Returns the stage's LDS address + 16 * lane.
  auto stage_begin = [&]() __attribute__((always_inline)) -> const char* {
    EM_TICK_TO(tb_);
    fd::wait_vmem();
    __syncthreads();
    EM_TICK_TO(p1_);
    const char* cur = lds + (consumed % EM_RING) * EM_STAGE + lane * 16;
    ++consumed;
    return cur;
  };
  // the copy of the stage after this one goes out behind the first unit's fragment reads and MFMAs (it has the other
  // units' time to land; issuing it first would put ~500 cycles of LDS-DMA issue in front of every stage)
  auto stage_prefetch = [&]() __attribute__((always_inline)) {
    if (issued < total_stages) issue_stage();
  };
  issue_stage();
  if (LNB) {
    if (tid < 2 * EM_C) lnacc[tid] = 0.f;           // (ordered before its first use by the first stage barrier / the one below)
    __syncthreads();
  }

  int tile = first, nxt = 0;
  for (int ti = 0;;) {
    // (dyn) the tile after this one is taken now: the atomic's round trip runs under this tile's work
    if (dyn && tid == 0) nxt = G + (int)atomicAdd(&sched[0], 1u);
    const long row = (long)tile * EM_ROWS + wave * 16 + m;
    const bool rok = row < rows;
    const long rc = rok ? row : rows - 1;         // rows past the end are clamped on load, masked on store
    const long qi = rc / d.nres;                  // (b, i)
    const long qj = (qi / d.nres) * d.nres + (rc - qi * d.nres);   // (b, j)

    // the kernel's input row in layer-output layout: lane (m, g) holds columns 16 nb + 4 g + r -- two consecutive blocks are the
    // B operand of a 32-k step in chained k order
    // forward: the input row is NOT held in registers (32 of them, beside the 96-register accumulator of layer 2, the 32 of layer 1
    // and 48 of weight fragments: the compiler answered with 80 - 140 spilled VGPRs and a scratch round trip of the accumulator
    // in every chunk) -- the two 16-blocks of a k-step of layer 1 are fetched one k-step ahead (L1 / L2 hits after the first
    // chunk), and the residual add of epilogue 2 fetches the row once more.  backward: the row (dy) is held: with the fused
    // prologue it exists only in registers.
    f32x4 X[BWD ? 8 : 1], Xk[2][2];
    auto load_xk = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(d.x + rc * EM_C + 16 * (2 * ks + i) + 4 * g);
        Xk[buf][i][0] = v.x; Xk[buf][i][1] = v.y; Xk[buf][i][2] = v.z; Xk[buf][i][3] = v.w;
      }
    };
    if (BWD) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!LNB || d.x != nullptr) v = *reinterpret_cast<const float4*>(d.x + rc * EM_C + 16 * nb + 4 * g);
        X[nb][0] = v.x; X[nb][1] = v.y; X[nb][2] = v.z; X[nb][3] = v.w;
      }
    } else {
      load_xk(0, 0);
    }
    if (LNB) {
      // ---- fused prologue of the backward: dz (upstream) [+ dzb W40]  ->  LayerNorm backward  ->  dy ----
      if (ZBW) {
        // X += dzb W40: operand = the row's 40 values of dzb in natural k order (two 32-k steps, zero beyond k = 40), four
        // leading units of the image = W40^T (n = 128 columns in two n-groups), the upstream gradient as the initial value
        float kz[2][8];
        {
          const float* zp = d.dzb + rc * EM_ZB + 8 * g;
          const float4 v = *reinterpret_cast<const float4*>(zp);
          const float4 w = *reinterpret_cast<const float4*>(zp + 4);
          kz[0][0] = v.x; kz[0][1] = v.y; kz[0][2] = v.z; kz[0][3] = v.w;
          kz[0][4] = w.x; kz[0][5] = w.y; kz[0][6] = w.z; kz[0][7] = w.w;
          float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = v1;
          if (g == 0) {
            v1 = *reinterpret_cast<const float4*>(d.dzb + rc * EM_ZB + 32);
            w1 = *reinterpret_cast<const float4*>(d.dzb + rc * EM_ZB + 36);
          }
          kz[1][0] = v1.x; kz[1][1] = v1.y; kz[1][2] = v1.z; kz[1][3] = v1.w;
          kz[1][4] = w1.x; kz[1][5] = w1.y; kz[1][6] = w1.z; kz[1][7] = w1.w;
        }
        uint4 bz[3];
        Em16Half Hz[2];
#pragma clang loop unroll(full)
        for (int sg = 0; sg < EM_ZB_UNITS / EM_UPS; ++sg) {
          const char* st = stage_begin();
          em16_read_half(Hz[0], st);
#pragma clang loop unroll(full)
          for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
            const int r = EM_UPS * sg + (hh >> 1), g2 = r & 1, a = 4 * g2 + 2 * (hh & 1);
            EM_PIN_TOP();
            if (hh + 1 < 2 * EM_UPS) em16_read_half(Hz[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
            if (g2 == 0 && (hh & 1) == 0) em_split8(kz[r >> 1], bz[0], bz[1], bz[2]);
            em16_mma_half(X[a], X[a + 1], Hz[hh & 1], bz);
            EM_GROUPS(hh + 1 < 2 * EM_UPS);
            if (hh == 1) stage_prefetch();
          }
        }
      }
      {
        const float rs = (d.ln_rowscale != nullptr ? d.ln_rowscale[rc] : 1.f) * (rok ? 1.f : 0.f);
        const float mean = d.ln_mean[rc], rstd = d.ln_rstd[rc];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          const int col = 16 * nb + 4 * g;
          const float4 yv = *reinterpret_cast<const float4*>(d.ln_y + rc * EM_C + col);
          const float4 gm = *reinterpret_cast<const float4*>(d.ln_gamma + col);
          const float hv[4] = {yv.x, yv.y, yv.z, yv.w}, gv[4] = {gm.x, gm.y, gm.z, gm.w};
          float cg[4], cb[4];      // this tile's 16-row sums for dgamma / dbeta of columns col .. col + 3 (DPP row reduction)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (hv[e] - mean) * rstd;
            const float gy = X[nb][e] * rs;
            cg[e] = fd::row16_sum(gy * xh);
            cb[e] = fd::row16_sum(gy);
            const float t = gy * gv[e];
            X[nb][e] = t;
            s1 += t;
            s2 += t * xh;
          }
          if (m == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fd::lds_add(&lnacc[col + e], cg[e]);
              fd::lds_add(&lnacc[EM_C + col + e], cb[e]);
            }
          }
        }
        s1 += __shfl_xor(s1, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16);
        s2 += __shfl_xor(s2, 32);
        const float m1 = s1 * (1.0f / 128.0f), m2 = s2 * (1.0f / 128.0f);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          const int col = 16 * nb + 4 * g;
          const float4 yv = *reinterpret_cast<const float4*>(d.ln_y + rc * EM_C + col);   // (second read: an L1 / L2 hit)
          const float hv[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) X[nb][e] = rstd * (X[nb][e] - m1 - (hv[e] - mean) * rstd * m2);
          if (d.dy_out != nullptr && rok)
            *reinterpret_cast<float4*>(d.dy_out + row * EM_C + col) = make_float4(X[nb][0], X[nb][1], X[nb][2], X[nb][3]);
        }
      }
    }

    EM_BODY_TO(LNB ? 9 : 0);
    // backward: packed ReLU gates (gmask1 / gmask2 = the forward's mask2 / mask1 outputs): bit 4 nb + e of word (row, chunk c, g)
    // says whether hidden unit 128 c + 16 nb + 4 g + e was positive -- 6 dwords per lane and tile, fetched here
    unsigned gm1[3] = {0u, 0u, 0u}, gm2[3] = {0u, 0u, 0u};
    if (BWD) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gm1[c] = d.gmask1[rc * 12 + 4 * c + g];
        gm2[c] = d.gmask2[rc * 12 + 4 * c + g];
      }
    }
    uint4 b[3];          // activation planes (B operand) of the current k-step
    Em16Half H[2];       // fragments of the current / next half-unit
    f32x4 acc2[24];
#pragma unroll
    for (int nb = 0; nb < 24; ++nb) {
      // forward: the layer-2 bias is the accumulator's initial value
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!BWD) a = *reinterpret_cast<const float4*>(d.bias2 + 16 * nb + 4 * g);
      acc2[nb][0] = a.x; acc2[nb][1] = a.y; acc2[nb][2] = a.z; acc2[nb][3] = a.w;
    }

    f32x4 acc1[8], acc1g[2];
    float4 zres[8];          // forward: the input row once more, for the residual into the final layer (requested under the last
                             // k-step of layer 2, when the layer-1 accumulator has freed its registers)
#pragma clang loop unroll(full)
    for (int c = 0; c < 3; ++c) {
      // hidden chunk of this pass: the backward walks 1, 2, 0 so that the UNGATED layer-1 accumulator of hidden units 0..127 -- the
      // first term of dz = (Wf^T dy)[0:128] + W1z^T d1 -- is the one still in the registers when layer 3 starts
      const int hc = BWD ? (c + 1) % 3 : c;
      // ---- layer 1, chunk hc: 128 hidden units x K = 128: units (k-step r >> 1, n-group r & 1) ----
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[nb][r] = 0.f;
#pragma clang loop unroll(full)
      for (int sg = 0; sg < 8 / EM_UPS; ++sg) {
        const char* st = stage_begin();
        em16_read_half(H[0], st);
#pragma clang loop unroll(full)
        for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
          const int r = EM_UPS * sg + (hh >> 1), g2 = r & 1, a = 4 * g2 + 2 * (hh & 1);
          EM_PIN_TOP();
          if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
          if (g2 == 0 && (hh & 1) == 0) {
            const int ks = r >> 1;
            if (BWD) {
              em16_split2(X[BWD ? 2 * ks : 0], X[BWD ? 2 * ks + 1 : 0], b[0], b[1], b[2]);
            } else {
              em16_split2(Xk[ks & 1][0], Xk[ks & 1][1], b[0], b[1], b[2]);
              if (!(c == 2 && ks == 3)) load_xk((ks + 1) & 1, (ks + 1) & 3);     // (the next chunk's first k-step behind the last)
            }
          }
          em16_mma_half(acc1[a], acc1[a + 1], H[hh & 1], b);
          EM_GROUPS(hh + 1 < 2 * EM_UPS);
          if (hh == 1) stage_prefetch();
        }
      }
      // epilogue 1, forward: h1 = relu(acc + P1_i + Q1_j)   (backward: the gate is applied where layer 2 splits the blocks)
      EM_BODY_TO(2);
      if (!BWD) {
        // (the sixteen per-residue fetches stay behind the chunk's last MFMAs: hoisted above them -- 64 registers in flight beside
        //  the fragments of the last half-unit -- they cost the inference variants 90 - 140 spilled VGPRs)
        fd::sched_pin();
        unsigned bits1 = 0u;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          const int col = 128 * hc + 16 * nb + 4 * g;
#ifdef EM_ABLATE_PQ      // (probe build only: what the kernel would take if the per-residue terms cost no fetch)
          const float4 a = make_float4(0.1f, 0.2f, -0.1f, 0.f), bq = make_float4(0.f, 0.1f, 0.f, -0.2f);
#else
          const float4 a = *reinterpret_cast<const float4*>(d.p1 + qi * ld_pq + col);
          const float4 bq = *reinterpret_cast<const float4*>(d.q1 + qj * ld_pq + col);
#endif
          float v[4] = {acc1[nb][0] + (a.x + bq.x), acc1[nb][1] + (a.y + bq.y), acc1[nb][2] + (a.z + bq.z),
                        acc1[nb][3] + (a.w + bq.w)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (TRAIN) bits1 |= (v[e] > 0.f ? 1u : 0u) << (4 * nb + e);
            acc1[nb][e] = v[e] > 0.f ? v[e] : 0.f;
          }
        }
        if (TRAIN && rok) d.mask1[row * 12 + 4 * hc + g] = bits1;
      }
      EM_BODY_TO(3);
      // ---- layer 2, k in chunk hc: units (k-step u2 / 6, n-group u2 % 6) ----
#pragma clang loop unroll(full)
      for (int sg = 0; sg < 24 / EM_UPS; ++sg) {
        const char* st = stage_begin();
        em16_read_half(H[0], st);
#pragma clang loop unroll(full)
        for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
          const int u2 = EM_UPS * sg + (hh >> 1), ks = u2 / 6, g6 = u2 % 6, a = 4 * g6 + 2 * (hh & 1);
          EM_PIN_TOP();
          if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
          if (g6 == 0 && (hh & 1) == 0) {
            // the two 16-blocks this k-step consumes: (backward) gated, split, and (training) saved HERE -- two stores per six
            // units of MFMAs instead of eight back to back in an epilogue
            if (BWD) {
#pragma unroll
              for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  acc1g[i][e] = ((gm1[hc] >> (4 * (2 * ks + i) + e)) & 1u) ? acc1[2 * ks + i][e] : 0.f;
              em16_split2(acc1g[0], acc1g[1], b[0], b[1], b[2]);
            } else {
              em16_split2(acc1[2 * ks], acc1[2 * ks + 1], b[0], b[1], b[2]);
            }
            const f32x4 (&t)[2] = BWD ? acc1g : reinterpret_cast<const f32x4 (&)[2]>(acc1[2 * ks]);
            if (!BWD && c == 2 && ks == 3) {
#pragma unroll
              for (int nb = 0; nb < 8; ++nb) zres[nb] = *reinterpret_cast<const float4*>(d.x + rc * EM_C + 16 * nb + 4 * g);
            }
            if (TRAIN && rok) {
#pragma unroll
              for (int i = 0; i < 2; ++i)
                EM_SAVE4(d.save1 + row * EM_H + 128 * hc + 16 * (2 * ks + i) + 4 * g, t[i][0], t[i][1], t[i][2], t[i][3]);
            }
          }
          em16_mma_half(acc2[a], acc2[a + 1], H[hh & 1], b);
          EM_GROUPS(hh + 1 < 2 * EM_UPS);
          if (hh == 1) stage_prefetch();
        }
      }
    }

    // epilogue 2: forward  h2 = relu(acc2), + z on the first 128 units (the residual into the final layer);  backward  d1 = acc2
    // gated by h1 > 0
    EM_BODY_TO(4);
    unsigned bits2 = 0u;
#pragma unroll
    for (int nb = 0; nb < 24; ++nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc2[nb][e];
        if (!BWD) {
          if (TRAIN) bits2 |= (v > 0.f ? 1u : 0u) << (4 * (nb & 7) + e);
          acc2[nb][e] = (v > 0.f ? v : 0.f) + (nb < 8 ? fd_f4(zres[nb & 7], e) : 0.f);
        } else {
          acc2[nb][e] = ((gm2[nb >> 3] >> (4 * (nb & 7) + e)) & 1u) ? v : 0.f;
        }
      }
      if (!BWD && TRAIN && (nb & 7) == 7) {
        if (rok) d.mask2[row * 12 + 4 * (nb >> 3) + g] = bits2;
        bits2 = 0u;
      }
    }

    EM_BODY_TO(5);
    // ---- layer 3: 128 outputs x K = 384 of the hidden layer; the backward starts from (Wf^T dy)[0:128], ungated ----
    f32x4 acc3[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc3[nb][r] = BWD ? acc1[nb][r] : 0.f;
#pragma clang loop unroll(full)
    for (int sg = 0; sg < 24 / EM_UPS; ++sg) {
      const char* st = stage_begin();
      em16_read_half(H[0], st);
#pragma clang loop unroll(full)
      for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
        const int v = EM_UPS * sg + (hh >> 1), ks = v >> 1, g2 = v & 1, a = 4 * g2 + 2 * (hh & 1);
        EM_PIN_TOP();
        if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
        if (g2 == 0 && (hh & 1) == 0) {
          em16_split2(acc2[2 * ks], acc2[2 * ks + 1], b[0], b[1], b[2]);
          if (TRAIN && rok) {      // the save of h2 + [z | 0 | 0] / d1, two blocks per k-step of layer 3
#pragma unroll
            for (int i = 0; i < 2; ++i)
              EM_SAVE4(d.save2 + row * EM_H + 16 * (2 * ks + i) + 4 * g, acc2[2 * ks + i][0], acc2[2 * ks + i][1], acc2[2 * ks + i][2],
                       acc2[2 * ks + i][3]);
          }
        }
        em16_mma_half(acc3[a], acc3[a + 1], H[hh & 1], b);
        EM_GROUPS(hh + 1 < 2 * EM_UPS);
        if (hh == 1) stage_prefetch();
      }
    }

    // ---- final epilogue ----
    EM_BODY_TO(6);
    if (!BWD) {
      // y = acc + Pf_i + Qf_j ; z' = rowscale * LayerNorm(y).  A row's 128 values sit in four lanes (l & 15 fixed).
      float s = 0.f;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const int col = 16 * nb + 4 * g;
#ifdef EM_ABLATE_PQ
        const float4 pa = make_float4(0.1f, 0.2f, -0.1f, 0.f), qa = make_float4(0.f, 0.1f, 0.f, -0.2f);
#else
        const float4 pa = *reinterpret_cast<const float4*>(d.pf + qi * ld_pqf + col);
        const float4 qa = *reinterpret_cast<const float4*>(d.qf + qj * ld_pqf + col);
#endif
        acc3[nb][0] += pa.x + qa.x; acc3[nb][1] += pa.y + qa.y; acc3[nb][2] += pa.z + qa.z; acc3[nb][3] += pa.w + qa.w;
        s += (acc3[nb][0] + acc3[nb][1]) + (acc3[nb][2] + acc3[nb][3]);
        if (TRAIN && d.y != nullptr && rok)
          *reinterpret_cast<float4*>(d.y + row * EM_C + col) = make_float4(acc3[nb][0], acc3[nb][1], acc3[nb][2], acc3[nb][3]);
      }
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / 128.0f);
      float vs = 0.f;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dlt = acc3[nb][r] - mean;
          acc3[nb][r] = dlt;
          vs += dlt * dlt;
        }
      vs += __shfl_xor(vs, 16);
      vs += __shfl_xor(vs, 32);
      const float rstd = 1.0f / sqrtf(vs * (1.0f / 128.0f) + d.eps);
      const float rs = d.rowscale != nullptr ? d.rowscale[rc] : 1.f;
      if (TRAIN && rok && g == 0) {
        if (d.mean != nullptr) d.mean[row] = mean;
        if (d.rstd != nullptr) d.rstd[row] = rstd;
      }
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const int col = 16 * nb + 4 * g;
        const float4 gm = *reinterpret_cast<const float4*>(d.gamma + col);
        const float4 bt = *reinterpret_cast<const float4*>(d.beta + col);
        float4 o;
        o.x = (acc3[nb][0] * rstd * gm.x + bt.x) * rs;
        o.y = (acc3[nb][1] * rstd * gm.y + bt.y) * rs;
        o.z = (acc3[nb][2] * rstd * gm.z + bt.z) * rs;
        o.w = (acc3[nb][3] * rstd * gm.w + bt.w) * rs;
        if (rok) *reinterpret_cast<float4*>(d.out + row * EM_C + col) = o;
        if (ZB) { acc3[nb][0] = o.x; acc3[nb][1] = o.y; acc3[nb][2] = o.z; acc3[nb][3] = o.w; }
      }
      if (ZB) {
        // ---- layer 4: zb[0:40] = W40 z' + b40 (n-blocks 0..2 of one n-group; columns >= 40 are zero weights) ----
        f32x4 acc4[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int col = 16 * nb + 4 * g;
          float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
          if (col < EM_ZB && d.zb_bias != nullptr) a = *reinterpret_cast<const float4*>(d.zb_bias + col);
          acc4[nb][0] = a.x; acc4[nb][1] = a.y; acc4[nb][2] = a.z; acc4[nb][3] = a.w;
        }
#pragma clang loop unroll(full)
        for (int sg = 0; sg < EM_ZB_UNITS / EM_UPS; ++sg) {
          const char* st = stage_begin();
          em16_read_half(H[0], st);
#pragma clang loop unroll(full)
          for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
            const int ks = EM_UPS * sg + (hh >> 1), a = 2 * (hh & 1);
            EM_PIN_TOP();
            if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
            if ((hh & 1) == 0) em16_split2(acc3[2 * ks], acc3[2 * ks + 1], b[0], b[1], b[2]);
            em16_mma_half(acc4[a], acc4[a + 1], H[hh & 1], b);
            EM_GROUPS(hh + 1 < 2 * EM_UPS);
            if (hh == 1) stage_prefetch();
          }
        }
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
          const int col = 16 * nb + 4 * g;
          if (rok && col < EM_ZB)
            *reinterpret_cast<float4*>(d.zb_out + row * EM_ZB + col) = make_float4(acc4[nb][0], acc4[nb][1], acc4[nb][2], acc4[nb][3]);
        }
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
        if (rok)
          *reinterpret_cast<float4*>(d.out + row * EM_C + 16 * nb + 4 * g) =
              make_float4(acc3[nb][0], acc3[nb][1], acc3[nb][2], acc3[nb][3]);
    }
    EM_BODY_TO(BWD ? 7 : 8);
    if (!dyn) {
      if (++ti >= nmine) break;
      tile = first + ti * G;
    } else {
      if (tid == 0) s_tile = nxt;
      __syncthreads();
      tile = s_tile;
      __syncthreads();          // (thread 0 writes s_tile again only after every thread has read it)
      if (tile >= ntiles) break;
    }
  }
#ifdef EM_PHASE_TIMING
  if (lane == 0) {
    const long long pv[10] = {p0_, p1_, p2_, p3_, p4_, p5_, p6_, p7_, p8_, p9_};
#pragma unroll
    for (int i = 0; i < 10; ++i)
      __hip_atomic_fetch_add(&em_phase[i], (unsigned long long)pv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) __hip_atomic_fetch_add(&em_phase[10], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  if (dyn) fd::wait_vmem();     // the stage copied ahead for a tile that does not exist must land before this block's LDS is freed
  if (LNB) {
    __syncthreads();
    if (tid < EM_C) {
      if (d.ln_dgamma != nullptr) atomicAdd(d.ln_dgamma + tid, lnacc[tid]);
    } else if (tid < 2 * EM_C) {
      if (d.ln_dbeta != nullptr) atomicAdd(d.ln_dbeta + (tid - EM_C), lnacc[tid]);
    }
  }
}

}  // namespace

#ifdef EM_PHASE_TIMING
extern "C" int fd_edge_mlp_phases(unsigned long long* host12, int reset) {
  if (host12 != nullptr && hipMemcpyFromSymbol(host12, HIP_SYMBOL(em_phase), sizeof(em_phase)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[12] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(em_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// the launch of THIS translation unit's shape (tile = 16 * EM_WAVES rows, 256 * EM_BLOCKS_PER_CU persistent blocks)
int EM_LAUNCH(const FdEdgeMlpDesc& d, hipStream_t st) {
  const long ntiles = (d.rows + EM_ROWS - 1) / EM_ROWS;
  const int blocks = d.blocks > 0 ? d.blocks : 256 * EM_BLOCKS_PER_CU;   // MI355X: persistent blocks fill the 256 CUs
  const int grid = (int)(ntiles < blocks ? ntiles : blocks);
  const dim3 g3(grid), b3(64 * EM_WAVES);
  // dynamic tile hand-out: only for launches whose blocks walk several tiles each; the counter word is zeroed here, on the launch's
  // own stream, so a launch never depends on how an earlier one left it
  FdEdgeMlpDesc dd = d;
  if (dd.sched != nullptr && ntiles >= 4L * grid) {
    FD_CHECK_ARG(hipMemsetAsync(dd.sched, 0, sizeof(unsigned), st) == hipSuccess, "fd_edge_mlp: zeroing the tile counter failed");
  } else {
    dd.sched = nullptr;
  }
  const bool zbv = d.zb_out != nullptr, train = d.save1 != nullptr;
#define EM_GO(...) hipLaunchKernelGGL(HIP_KERNEL_NAME(edge_mlp16_kernel<__VA_ARGS__>), g3, b3, 0, st, dd)
  if (d.backward) {
    if (d.ln_y != nullptr && d.dzb != nullptr) EM_GO(true, false, true, true, true);
    else if (d.ln_y != nullptr) EM_GO(true, false, true, true, false);
    else EM_GO(true, false, true, false, false);
  } else if (zbv) {
    if (train) EM_GO(false, true, true); else EM_GO(false, true, false);
  } else {
    if (train) EM_GO(false, false, true); else EM_GO(false, false, false);
  }
#undef EM_GO
  FD_CHECK_LAUNCH("fd_edge_mlp");
  return FD_OK;
}

#ifndef EM_SHAPE_W8
extern "C" int fd_edge_mlp_pack(const float* W1, const float* W2, const float* Wf, long ld, void* img, void* stream) {
  // forward image: A1 = W1[:, 0:128] [384,128], A2 = W2 [384,384], A4 = Wf [128,384], chunks in natural order
  FD_CHECK_ARG(W1 && W2 && Wf && img, "fd_edge_mlp_pack: null operand");
  FD_CHECK_ARG(fd_aligned16(img), "fd_edge_mlp_pack: image must be 16-byte aligned");
  EmMat m1{W1, ld, 1}, m2{W2, ld, 1}, m4{Wf, ld, 1};
  hipLaunchKernelGGL(edge_mlp_pack16_kernel, dim3(EM_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, m1, m2, m4,
                     static_cast<char*>(img), 0);
  FD_CHECK_LAUNCH("fd_edge_mlp_pack");
  return FD_OK;
}

extern "C" int fd_edge_mlp_pack_bwd(const float* Wf, const float* W2, const float* W1, long ld, const float* W40, void* img,
                                    void* stream) {
  // backward image: [4 units W40^T (optional, when W40 != null)] + 120 units of the transposed chain (A1 = Wf^T [384,128],
  // A2 = W2^T, A4 = W1[:, 0:128]^T [128,384]) with the hidden chunks in the order 1, 2, 0
  FD_CHECK_ARG(Wf && W2 && W1 && img, "fd_edge_mlp_pack_bwd: null operand");
  FD_CHECK_ARG(fd_aligned16(img), "fd_edge_mlp_pack_bwd: image must be 16-byte aligned");
  char* base = static_cast<char*>(img);
  if (W40 != nullptr) {
    hipLaunchKernelGGL(edge_mlp_pack_zbw_kernel, dim3(EM_ZB_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, W40, base);
    base += (long)EM_ZB_UNITS * EM_UNIT;
  }
  EmMat m1{Wf, 1, ld}, m2{W2, 1, ld}, m4{W1, 1, ld};
  hipLaunchKernelGGL(edge_mlp_pack16_kernel, dim3(EM_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, m1, m2, m4, base, 1);
  FD_CHECK_LAUNCH("fd_edge_mlp_pack_bwd");
  return FD_OK;
}

extern "C" int fd_edge_mlp_pack_zb(const float* W40, void* img, void* stream) {
  FD_CHECK_ARG(W40 && img, "fd_edge_mlp_pack_zb: null operand");
  FD_CHECK_ARG(fd_aligned16(img), "fd_edge_mlp_pack_zb: image must be 16-byte aligned");
  hipLaunchKernelGGL(edge_mlp_pack_zb_kernel, dim3(EM_ZB_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, W40,
                     static_cast<char*>(img));
  FD_CHECK_LAUNCH("fd_edge_mlp_pack_zb");
  return FD_OK;
}

extern "C" int fd_edge_mlp(const FdEdgeMlpDesc* desc, void* stream) {
  FD_CHECK_ARG(desc != nullptr, "fd_edge_mlp: null descriptor");
  const FdEdgeMlpDesc& d = *desc;
  FD_CHECK_ARG(d.img && d.out && (d.x || (d.backward && d.ln_y && d.dzb)), "fd_edge_mlp: x / img / out are required");
  FD_CHECK_ARG(d.ln_y == nullptr || (d.backward && d.ln_mean && d.ln_rstd && d.ln_gamma),
               "fd_edge_mlp: the fused LayerNorm backward (ln_y) is a backward option and needs ln_mean / ln_rstd / ln_gamma");
  FD_CHECK_ARG(d.dzb == nullptr || d.ln_y != nullptr, "fd_edge_mlp: dzb needs the fused LayerNorm-backward prologue (ln_y)");
  FD_CHECK_ARG(d.nres > 0 && d.rows >= 0, "fd_edge_mlp: bad extents");
  if (d.backward) {
    FD_CHECK_ARG(d.gmask1 && d.gmask2 && d.save1 && d.save2,
                 "fd_edge_mlp(backward): the packed sign masks of h2 (gmask1) and h1 (gmask2) and the outputs d2 (save1) / d1 (save2) "
                 "are required");
    FD_CHECK_ARG(!d.mask1 && !d.mask2 && !d.zb_out, "fd_edge_mlp(backward): mask1 / mask2 / zb_out are forward outputs");
  } else {
    FD_CHECK_ARG(d.p1 && d.q1 && d.bias2 && d.pf && d.qf && d.gamma && d.beta,
                 "fd_edge_mlp(forward): p1 / q1 / bias2 / pf / qf / gamma / beta are required");
    const int nsave = (d.save1 != nullptr) + (d.save2 != nullptr) + (d.mask1 != nullptr) + (d.mask2 != nullptr);
    FD_CHECK_ARG(nsave == 0 || nsave == 4,
                 "fd_edge_mlp(forward): the training outputs save1 (h1), save2 (h2 + [z | 0 | 0]), mask1, mask2 come together");
    FD_CHECK_ARG(nsave == 4 || (!d.y && !d.mean && !d.rstd), "fd_edge_mlp(forward): y / mean / rstd are training outputs (with the saves)");
    FD_CHECK_ARG(!d.gmask1 && !d.gmask2, "fd_edge_mlp(forward): gmask1 / gmask2 are backward inputs");
  }
  const void* ptrs[] = {d.x, d.img, d.out, d.p1, d.q1, d.bias2, d.save1, d.save2, d.pf, d.qf,
                        d.gamma, d.beta, d.y, d.ln_y, d.ln_gamma, d.dy_out, d.dzb};
  for (const void* p : ptrs) FD_CHECK_ARG(fd_aligned16(p), "fd_edge_mlp: operands must be 16-byte aligned");
  if (d.rows == 0) return FD_OK;
  FD_CHECK_ARG(d.zb_out == nullptr || (fd_aligned16(d.zb_out) && fd_aligned16(d.zb_bias)),
               "fd_edge_mlp: zb_out / zb_bias must be 16-byte aligned (the image must carry the fd_edge_mlp_pack_zb units)");
  FD_CHECK_ARG(d.shape == 0 || d.shape == 2 || d.shape == 4 || d.shape == 8,
               "fd_edge_mlp: shape is 0 (by size), 2 (two waves per 16 rows: inference forward), 4 or 8 (waves per block)");
  // shape by size: the one-block-per-CU shape from two of its 128-row tiles per CU up (measured fwd / fwd + saves / bwd: 16,384 rows
  // 0.088 vs 0.063 ms; 65,536 rows 0.171-0.179 / 0.216-0.240 / 0.196-0.213 vs 0.190-0.193 / 0.225-0.253 / 0.210-0.231 ms; 458,752 rows
  // 1.21 / 1.42 / 1.33 vs 1.23 / 1.66 / 1.47 ms; profiles/r04_edge_variants_*); an inference forward that gives a SIMD at most one
  // 16-row group (a lone backbone of N <= 128) takes the column-split kernel: two waves per group (fd_edge_mlp_pair.hip)
  const bool pair_ok = !d.backward && d.save1 == nullptr;
  FD_CHECK_ARG(d.shape != 2 || pair_ok, "fd_edge_mlp: shape 2 is an inference forward (no backward, no training saves)");
  const int shape = d.shape != 0 ? d.shape
                                 : (d.rows >= FD_EDGE_MLP_W8_MIN_ROWS ? 8 : (pair_ok && d.rows <= FD_EDGE_MLP_PAIR_MAX_ROWS ? 2 : 4));
  hipStream_t st = (hipStream_t)stream;
  if (shape == 2) return fd_edge_mlp_launch_pair(d, st);
  return shape == 8 ? fd_edge_mlp_launch_w8(d, st) : fd_edge_mlp_launch_w4(d, st);
}
#endif  // !EM_SHAPE_W8
