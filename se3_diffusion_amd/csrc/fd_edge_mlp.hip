// Edge transition (model/ipa_pytorch.py:194-233) as ONE kernel per direction: the 128 -> 384 -> 384 -> 128 chain of a
// pair row never leaves the CU.
//
//   forward :  h1 = relu(W1z z + P1_i + Q1_j)      h2 = relu(W2 h1 + b2)      y = Wf h2 + Wfz z + Pf_i + Qf_j
//              z' = mask * LayerNorm(y)                          (P/Q: the node halves of W [z | e_i | e_j], trunk.py)
//   backward:  d2 = [h2 > 0] Wf^T dy               d1 = [h1 > 0] W2^T d2      dz = Wfz^T dy + W1z^T d1
//              (the same dataflow with transposed weights and ReLU gates instead of bias + ReLU)
//
// Arithmetic: the split-bf16 scheme of fd_gemm_split.h -- every fp32 operand is three exact bf16 terms, six
// v_mfma_f32_32x32x16_bf16 products per 16-k step, fp32 accumulation: fp32-accurate (not bitwise an fmaf chain).
//
// Register-chained layout.  A wave owns 32 pair rows for the whole chain and accumulates TRANSPOSED
// (D[n][m] = sum_k W[n][k] x[m][k]: the weights are the MFMA's row operand), so lane (m = l & 31, h = l >> 5) ends up
// with 16 hidden units n = 32 nb + 8 q + 4 h + e of ITS OWN row per 32-unit block.  Those registers are exactly a
// B-operand fragment of the next GEMM if the next layer's weights are stored with the k order permuted to match
// (slot (h, e') <-> k = 8 (e' >> 2) + 4 h + (e' & 3) inside a 16-k step): activations go accumulator -> relu ->
// bf16 split -> MFMA operand without touching LDS, without a barrier, without leaving the wave.  LDS only streams
// the weights, which are packed ONCE per optimiser step (fd_edge_mlp_pack) into bf16-plane fragments in the exact
// order the kernel consumes them, so staging is a straight LDS-DMA copy (global_load_lds_dwordx4, no VGPRs, no VALU)
// and every fragment read is a conflict-free ds_read_b128 of a lane-linear 1 KB piece.
//
// Block = 4 waves (one per SIMD, up to 512 registers each: the 32 x 384 fp32 accumulator of layer 2 alone is 192) x
// 32 rows = 128-row tiles, persistent (one block per CU walks the tiles).  Per tile the weight stream is 128 units of
// 12 KB (4 n-blocks x one 16-k step x 3 planes), grouped in 32 stages of 48 KB through a two-stage LDS ring; stage s+1
// is in flight while stage s (96 MFMAs per wave, ~3k cycles) is multiplied:
//     for c in 0..2:   8 units  W1z[n in chunk c]        (layer 1, K = 128)      -> h1 chunk c (64 registers)
//                     24 units  W2[all n][k in chunk c]  (layer 2, partial K)    -> acc2 += ...
//     8 units Wfz, 24 units Wf                           (layer 3, K = 128 + 384)
// Algorithmic HBM bytes per pair row: 512 read + 512 written (+ h1, h2, y saved for the backward in training).
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int EM_ROWS = 128;               // rows per block tile (4 waves x 32)
constexpr int EM_PIECE = 1024;             // one fragment: 64 lanes x 16 B
constexpr int EM_UNIT = 12 * EM_PIECE;     // 4 n-blocks x 3 planes
constexpr int EM_UPS = 4;                  // units per stage
constexpr int EM_STAGE = EM_UPS * EM_UNIT; // 48 KB
constexpr int EM_UNITS = 128;              // units per tile
constexpr int EM_NSTAGE = EM_UNITS / EM_UPS;
constexpr int EM_H = 384, EM_C = 128;

__device__ __forceinline__ void em_split8(const float (&x)[8], uint4& s0, uint4& s1, uint4& s2) {
  unsigned t0[4], t1[4], t2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = x[2 * j], v = x[2 * j + 1];
    const unsigned hh = fd::pack_bf16(u, v);
    const float ru = u - fd::bf16lo_f32(hh), rv = v - fd::bf16hi_f32(hh);
    const unsigned mm = fd::pack_bf16(ru, rv);
    const float qu = ru - fd::bf16lo_f32(mm), qv = rv - fd::bf16hi_f32(mm);
    t0[j] = hh;
    t1[j] = mm;
    t2[j] = fd::pack_bf16(qu, qv);
  }
  s0 = make_uint4(t0[0], t0[1], t0[2], t0[3]);
  s1 = make_uint4(t1[0], t1[1], t1[2], t1[3]);
  s2 = make_uint4(t2[0], t2[1], t2[2], t2[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// weight image: 128 units in consumption order; a unit = [4 n-blocks][3 planes][64 lanes] x 16 B, the fragment of lane
// (l31 = n, h) holding slots e' = 0..7 of a 16-k step.  natural k order (operand loaded from memory):
// k = k0 + 8 h + e'; chained k order (operand = the previous layer's accumulator): k = k0 + 8 (e' >> 2) + 4 h + (e' & 3).
struct EmMat {
  const float* p;
  long rs, cs;
};

__global__ __launch_bounds__(256) void edge_mlp_pack_kernel(EmMat A1, EmMat A2, EmMat A3, EmMat A4, char* __restrict__ img) {
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (unit, n-block, lane)
  if (gid >= EM_UNITS * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, u = gid >> 8;
  const int l31 = lane & 31, h = lane >> 5;
  EmMat M;
  int n, k0;
  bool chained;
  if (u < 96) {
    const int c = u / 32, r = u % 32;
    if (r < 8) {                    // layer 1: rows of chunk c, k-step r
      M = A1; n = 128 * c + 32 * i + l31; k0 = 16 * r; chained = false;
    } else {                        // layer 2: all rows (group g), k-step ks of chunk c
      const int r2 = r - 8, ks = r2 / 3, g = r2 % 3;
      M = A2; n = 32 * (4 * g + i) + l31; k0 = 128 * c + 16 * ks; chained = true;
    }
  } else if (u < 104) {             // layer 3, x part
    M = A3; n = 32 * i + l31; k0 = 16 * (u - 96); chained = false;
  } else {                          // layer 3, hidden part
    M = A4; n = 32 * i + l31; k0 = 16 * (u - 104); chained = true;
  }
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = chained ? k0 + 8 * (e >> 2) + 4 * h + (e & 3) : k0 + 8 * h + e;
    x[e] = M.p[(long)n * M.rs + (long)k * M.cs];
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)u * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

// Half a unit = 2 n-blocks x one 16-k step: 6 fragments (24 registers), 12 MFMAs.  The kernel is software-pipelined at
// this granularity: while half-unit i is multiplied out of one fragment set, the fragments of half-unit i+1 are read
// into the other (prefetch distance 12 MFMAs = 384 cycles, LDS latency ~130).
struct EmHalf {
  uint4 w[2][3];
};

__device__ __forceinline__ void em_read_half(EmHalf& f, const char* __restrict__ u) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s) f.w[i][s] = *reinterpret_cast<const uint4*>(u + (i * 3 + s) * EM_PIECE);
}

__device__ __forceinline__ void em_mma_half(f32x16& a0, f32x16& a1, const EmHalf& f, const uint4 (&b)[3]) {
  // products (weight plane, activation plane) with i + j <= 2, smallest first
  constexpr int PW[6] = {2, 1, 0, 1, 0, 0};
  constexpr int PX[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    a0 = fd::mfma_32x32x16_bf16(f.w[0][PW[p]], b[PX[p]], a0);
    a1 = fd::mfma_32x32x16_bf16(f.w[1][PW[p]], b[PX[p]], a1);
  }
}

__device__ __forceinline__ void em_zero(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

__device__ __forceinline__ void em_load_x(float (&xr)[8][8], const float* __restrict__ xp) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const float4 v = *reinterpret_cast<const float4*>(xp + 16 * ks);
    const float4 w = *reinterpret_cast<const float4*>(xp + 16 * ks + 4);
    xr[ks][0] = v.x; xr[ks][1] = v.y; xr[ks][2] = v.z; xr[ks][3] = v.w;
    xr[ks][4] = w.x; xr[ks][5] = w.y; xr[ks][6] = w.z; xr[ks][7] = w.w;
  }
}

template <bool BWD>
__global__ __launch_bounds__(256, 1) void edge_mlp_kernel(FdEdgeMlpDesc d) {
  __shared__ __attribute__((aligned(16))) char lds[2 * EM_STAGE];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const long rows = d.rows;
  const int ntiles = (int)((rows + EM_ROWS - 1) / EM_ROWS);
  const int G = (int)gridDim.x, first = (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nmine = (ntiles - first + G - 1) / G;
  const int total_stages = nmine * EM_NSTAGE;

  // ---- weight stream: every wave copies a quarter (12 pieces) of each stage; stage s lives in buffer s & 1 ----
  const char* __restrict__ img_lane = static_cast<const char*>(d.img) + wave * (EM_STAGE / 4) + lane * 16;
  char* const lds_wave = lds + wave * (EM_STAGE / 4);
  const char* const lds_lane = lds + lane * 16;
  int issued = 0;      // stages whose copy has been issued
  auto issue_stage = [&]() {
    const char* src = img_lane + (long)(issued % EM_NSTAGE) * EM_STAGE;
    char* dst = lds_wave + (issued & 1) * EM_STAGE;
#pragma unroll
    for (int g4 = 0; g4 < 3; ++g4)   // four pieces per address setup (the immediate offset reaches 4095)
      fd::glds16x4(src + g4 * 4096, dst + g4 * 4096);
    ++issued;
  };
  // Called when the LAST half-unit of a stage has its fragments in registers: the next stage's copy (issued one stage
  // ago) has landed and is visible to the block, every wave has finished READING the stage that ends, so its buffer
  // takes the copy of the stage after the next.
  auto stage_advance = [&]() {
    fd::wait_vmem();     // the LDS-DMA is issued from inline asm: the compiler does not wait for it
    __syncthreads();
    if (issued < total_stages) issue_stage();
  };

  EmHalf H[2];
  uint4 bq[2][3];      // activation planes (B operand) of a k-step, ping-pong: every region has an even number of k-steps
  float xr[8][8];      // x in B-operand layout: k = 16 ks + 8 h + e
  float4 pre[4][4];    // forward: P1_i + Q1_j of the COMING chunk (loaded under layer 2); backward: this chunk's gate (h2)

  auto row_of = [&](int ti) -> long { return ((long)first + (long)ti * G) * EM_ROWS + wave * 32 + l31; };
  // one n-block (16 values per lane) of the chunk's pair terms / gates.  Issued in four groups spread over the
  // previous phase's stages: 8 loads in flight cost 32 registers, all 32 at once would cost 128.
  auto load_pre = [&](long rcx, int c, int nb) {
    const long qix = rcx / d.nres, qjx = (qix / d.nres) * d.nres + (rcx - qix * d.nres);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 128 * c + 32 * nb + 8 * q + 4 * h;
      if (!BWD) {
        const float4 a = *reinterpret_cast<const float4*>(d.p1 + qix * EM_H + col);
        const float4 bq_ = *reinterpret_cast<const float4*>(d.q1 + qjx * EM_H + col);
        pre[nb][q] = make_float4(a.x + bq_.x, a.y + bq_.y, a.z + bq_.z, a.w + bq_.w);
      } else {
        pre[nb][q] = *reinterpret_cast<const float4*>(d.gate1 + rcx * EM_H + col);
      }
    }
  };

  // half-unit HJ of a region (static index; 16 half-units fill the two stage buffers)
#define EM_HALF(HJ, A0, A1, BCUR)                                                          \
  do {                                                                                     \
    if ((((HJ) + 1) & 7) == 0) stage_advance();                                            \
    em_read_half(H[((HJ) + 1) & 1], lds_lane + (((HJ) + 1) & 15) * (EM_UNIT / 2));         \
    fd::sched_pin(); /* the prefetch stays ahead of this half-unit's MFMAs */               \
    em_mma_half(A0, A1, H[(HJ) & 1], BCUR);                                                \
  } while (0)

  // ---- prologue ----
  issue_stage();
  {
    const long r = row_of(0);
    const long rc = r < rows ? r : rows - 1;
    em_load_x(xr, d.x + rc * EM_C + 8 * h);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      load_pre(rc, 0, nb);
      fd::sched_fence();
    }
  }
  fd::wait_vmem();
  __syncthreads();
  issue_stage();
  em_read_half(H[0], lds_lane);
  em_split8(xr[0], bq[0][0], bq[0][1], bq[0][2]);

  for (int ti = 0; ti < nmine; ++ti) {
    const long row = row_of(ti);
    const bool rok = row < rows;
    const long rc = rok ? row : rows - 1;         // rows past the end are clamped on load, masked on store
    const long rown = row_of(ti + 1);
    const long rcn = rown < rows ? rown : rows - 1;   // the next tile's row (any valid row when there is none)

    f32x16 acc2[12];
#pragma unroll
    for (int nb = 0; nb < 12; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // forward: the layer-2 bias is the accumulator's initial value
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!BWD) a = *reinterpret_cast<const float4*>(d.bias2 + 32 * nb + 8 * q + 4 * h);
        acc2[nb][4 * q + 0] = a.x; acc2[nb][4 * q + 1] = a.y; acc2[nb][4 * q + 2] = a.z; acc2[nb][4 * q + 3] = a.w;
      }

    for (int c = 0; c < 3; ++c) {
      // ---- layer 1, chunk c: 128 hidden units x K = 128 (8 units) ----
      f32x16 acc1[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // forward: the pair terms are the accumulator's initial value
          acc1[nb][4 * q + 0] = BWD ? 0.f : pre[nb][q].x; acc1[nb][4 * q + 1] = BWD ? 0.f : pre[nb][q].y;
          acc1[nb][4 * q + 2] = BWD ? 0.f : pre[nb][q].z; acc1[nb][4 * q + 3] = BWD ? 0.f : pre[nb][q].w;
        }
#pragma clang loop unroll(full)
      for (int j = 0; j < 8; ++j) {
        if (j + 1 < 8) em_split8(xr[j + 1], bq[(j + 1) & 1][0], bq[(j + 1) & 1][1], bq[(j + 1) & 1][2]);
        if (BWD && (j & 1) == 0) load_pre(rc, c, j >> 1);   // this chunk's gates (h2), used by epilogue 1
        EM_HALF(2 * j, acc1[0], acc1[1], bq[j & 1]);
        EM_HALF(2 * j + 1, acc1[2], acc1[3], bq[j & 1]);
      }
      // epilogue 1: forward  h1 = relu(acc);  backward  d2 = acc gated by h2 > 0
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc1[nb][4 * q + e];
          if (!BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          } else {
            const float4 g = pre[nb][q];
            v[0] = g.x > 0.f ? v[0] : 0.f; v[1] = g.y > 0.f ? v[1] : 0.f;
            v[2] = g.z > 0.f ? v[2] : 0.f; v[3] = g.w > 0.f ? v[3] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc1[nb][4 * q + e] = v[e];
          if (d.save1 != nullptr && rok)
            *reinterpret_cast<float4*>(d.save1 + row * EM_H + 128 * c + 32 * nb + 8 * q + 4 * h) =
                make_float4(v[0], v[1], v[2], v[3]);
        }
      {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = acc1[0][e];
        em_split8(t, bq[0][0], bq[0][1], bq[0][2]);
      }
      // ---- layer 2, k in chunk c: 24 units (k-step ks, row group g) ----
#pragma clang loop unroll(full)
      for (int u2 = 0; u2 < 24; ++u2) {
        const int ks = u2 / 3, g = u2 % 3, j = 8 + u2;
        if (g == 2 && ks + 1 < 8) {        // planes of the next k-step
          float t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = acc1[(ks + 1) >> 1][8 * ((ks + 1) & 1) + e];
          em_split8(t, bq[(ks + 1) & 1][0], bq[(ks + 1) & 1][1], bq[(ks + 1) & 1][2]);
        }
        if (u2 == 23) em_split8(xr[0], bq[0][0], bq[0][1], bq[0][2]);   // first k-step of what follows (layer 1 or layer 3)
        if (!BWD && (u2 & 3) == 2 && u2 < 16) {   // pair terms of the next chunk (the next tile's first after c = 2),
          if (c < 2) load_pre(rc, c + 1, u2 >> 2); else load_pre(rcn, 0, u2 >> 2);   // one n-block per stage
        }
        EM_HALF(2 * j, acc2[4 * g + 0], acc2[4 * g + 1], bq[ks & 1]);
        EM_HALF(2 * j + 1, acc2[4 * g + 2], acc2[4 * g + 3], bq[ks & 1]);
      }
    }

    f32x16 acc3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) em_zero(acc3[i]);
    // epilogue 2: forward  h2 = relu(acc2);  backward  d1 = acc2 gated by h1 > 0
#pragma unroll
    for (int nb = 0; nb < 12; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * nb + 8 * q + 4 * h;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc2[nb][4 * q + e];
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        } else {
          const float4 g = *reinterpret_cast<const float4*>(d.gate2 + rc * EM_H + col);
          v[0] = g.x > 0.f ? v[0] : 0.f; v[1] = g.y > 0.f ? v[1] : 0.f;
          v[2] = g.z > 0.f ? v[2] : 0.f; v[3] = g.w > 0.f ? v[3] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2[nb][4 * q + e] = v[e];
        if (d.save2 != nullptr && rok)
          *reinterpret_cast<float4*>(d.save2 + row * EM_H + col) = make_float4(v[0], v[1], v[2], v[3]);
      }

    // ---- layer 3: 128 outputs x (K = 128 of x, then K = 384 of the hidden layer) ----
#pragma clang loop unroll(full)
    for (int j = 0; j < 8; ++j) {
      if (j + 1 < 8) {
        em_split8(xr[j + 1], bq[(j + 1) & 1][0], bq[(j + 1) & 1][1], bq[(j + 1) & 1][2]);
      } else {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = acc2[0][e];
        em_split8(t, bq[0][0], bq[0][1], bq[0][2]);
      }
      EM_HALF(2 * j, acc3[0], acc3[1], bq[j & 1]);
      EM_HALF(2 * j + 1, acc3[2], acc3[3], bq[j & 1]);
    }
#pragma clang loop unroll(full)
    for (int v = 0; v < 24; ++v) {
      const int j = 8 + v;
      if (v == 0) em_load_x(xr, d.x + rcn * EM_C + 8 * h);     // x of the next tile (this tile's is consumed)
      if (v + 1 < 24) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = acc2[(v + 1) >> 1][8 * ((v + 1) & 1) + e];
        em_split8(t, bq[(v + 1) & 1][0], bq[(v + 1) & 1][1], bq[(v + 1) & 1][2]);
      } else {
        em_split8(xr[0], bq[0][0], bq[0][1], bq[0][2]);        // first k-step of the next tile
      }
      EM_HALF(2 * j, acc3[0], acc3[1], bq[v & 1]);
      EM_HALF(2 * j + 1, acc3[2], acc3[3], bq[v & 1]);
    }

    // ---- final epilogue ----
    if (!BWD) {
      // z' = rowscale * LayerNorm(y).  A row's 128 values sit in two lanes (l, l ^ 32).
      float s = 0.f;
      const long qi = rc / d.nres, qj = (qi / d.nres) * d.nres + (rc - qi * d.nres);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * nb + 8 * q + 4 * h;
          const float4 pa = *reinterpret_cast<const float4*>(d.pf + qi * EM_C + col);
          const float4 qa = *reinterpret_cast<const float4*>(d.qf + qj * EM_C + col);
          acc3[nb][4 * q + 0] += pa.x + qa.x; acc3[nb][4 * q + 1] += pa.y + qa.y;
          acc3[nb][4 * q + 2] += pa.z + qa.z; acc3[nb][4 * q + 3] += pa.w + qa.w;
        }
        fd::sched_fence();
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s += (acc3[nb][4 * q + 0] + acc3[nb][4 * q + 1]) + (acc3[nb][4 * q + 2] + acc3[nb][4 * q + 3]);
          if (d.y != nullptr && rok)
            *reinterpret_cast<float4*>(d.y + row * EM_C + 32 * nb + 8 * q + 4 * h) =
                make_float4(acc3[nb][4 * q + 0], acc3[nb][4 * q + 1], acc3[nb][4 * q + 2], acc3[nb][4 * q + 3]);
        }
      s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / 128.0f);
      float vs = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dlt = acc3[nb][r] - mean;
          acc3[nb][r] = dlt;
          vs += dlt * dlt;
        }
      vs += __shfl_xor(vs, 32);
      const float rstd = 1.0f / sqrtf(vs * (1.0f / 128.0f) + d.eps);
      const float rs = d.rowscale != nullptr ? d.rowscale[rc] : 1.f;
      if (rok && h == 0) {
        if (d.mean != nullptr) d.mean[row] = mean;
        if (d.rstd != nullptr) d.rstd[row] = rstd;
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        fd::sched_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * nb + 8 * q + 4 * h;
          const float4 gm = *reinterpret_cast<const float4*>(d.gamma + col);
          const float4 bt = *reinterpret_cast<const float4*>(d.beta + col);
          float4 o;
          o.x = (acc3[nb][4 * q + 0] * rstd * gm.x + bt.x) * rs;
          o.y = (acc3[nb][4 * q + 1] * rstd * gm.y + bt.y) * rs;
          o.z = (acc3[nb][4 * q + 2] * rstd * gm.z + bt.z) * rs;
          o.w = (acc3[nb][4 * q + 3] * rstd * gm.w + bt.w) * rs;
          if (rok) *reinterpret_cast<float4*>(d.out + row * EM_C + col) = o;
        }
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (rok)
            *reinterpret_cast<float4*>(d.out + row * EM_C + 32 * nb + 8 * q + 4 * h) =
                make_float4(acc3[nb][4 * q + 0], acc3[nb][4 * q + 1], acc3[nb][4 * q + 2], acc3[nb][4 * q + 3]);
        }
    }
  }
#undef EM_HALF
}

}  // namespace

extern "C" int fd_edge_mlp_pack(const float* A1, long rs1, long cs1, const float* A2, long rs2, long cs2,
                                const float* A3, long rs3, long cs3, const float* A4, long rs4, long cs4, void* img,
                                void* stream) {
  FD_CHECK_ARG(A1 && A2 && A3 && A4 && img, "fd_edge_mlp_pack: null operand");
  FD_CHECK_ARG(fd_aligned16(img), "fd_edge_mlp_pack: image must be 16-byte aligned");
  EmMat m1{A1, rs1, cs1}, m2{A2, rs2, cs2}, m3{A3, rs3, cs3}, m4{A4, rs4, cs4};
  hipLaunchKernelGGL(edge_mlp_pack_kernel, dim3(EM_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, m1, m2, m3,
                     m4, static_cast<char*>(img));
  FD_CHECK_LAUNCH("fd_edge_mlp_pack");
  return FD_OK;
}

extern "C" int fd_edge_mlp(const FdEdgeMlpDesc* desc, void* stream) {
  FD_CHECK_ARG(desc != nullptr, "fd_edge_mlp: null descriptor");
  const FdEdgeMlpDesc& d = *desc;
  FD_CHECK_ARG(d.x && d.img && d.out, "fd_edge_mlp: x / img / out are required");
  FD_CHECK_ARG(d.nres > 0 && d.rows >= 0, "fd_edge_mlp: bad extents");
  if (d.backward) {
    FD_CHECK_ARG(d.gate1 && d.gate2, "fd_edge_mlp(backward): the saved activations h2 (gate1) and h1 (gate2) are required");
  } else {
    FD_CHECK_ARG(d.p1 && d.q1 && d.bias2 && d.pf && d.qf && d.gamma && d.beta,
                 "fd_edge_mlp(forward): p1 / q1 / bias2 / pf / qf / gamma / beta are required");
  }
  const void* ptrs[] = {d.x, d.img, d.out, d.p1, d.q1, d.bias2, d.gate1, d.gate2, d.save1, d.save2, d.pf, d.qf,
                        d.gamma, d.beta, d.y};
  for (const void* p : ptrs) FD_CHECK_ARG(fd_aligned16(p), "fd_edge_mlp: operands must be 16-byte aligned");
  if (d.rows == 0) return FD_OK;
  const long ntiles = (d.rows + EM_ROWS - 1) / EM_ROWS;
  const int blocks = d.blocks > 0 ? d.blocks : 256;   // MI355X: one persistent block per CU
  const int grid = (int)(ntiles < blocks ? ntiles : blocks);
  if (d.backward)
    hipLaunchKernelGGL(edge_mlp_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d);
  else
    hipLaunchKernelGGL(edge_mlp_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d);
  FD_CHECK_LAUNCH("fd_edge_mlp");
  return FD_OK;
}
