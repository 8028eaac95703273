// Invariant Point Attention, forward, as ONE kernel per block of the trunk ("flash" form): the logits, the probabilities and
// the three value products of a tile of query rows never leave the CU.
//
// Reference: model/ipa_pytorch.py:380-457 --
//   a_ij^h = sqrt(1/3C) q_i.k_j + sqrt(1/3) b_ij^h - 0.5 gamma_h sum_p |T_i q_p - T_j k_p|^2 + 1e5 (m_i m_j - 1)   (:380-417)
//   a = softmax_j(a)                                                                                              (:422)
//   o = sum_j a v_j (:424-428) ; o_pt = T_i^-1 (sum_j a T_j v_p), |o_pt| (:432-449) ; o_pair = sum_j a_ij down_z(z)_ij (:455-457)
// and replaces the launch sequence  q k^T GEMM -> fd_ipa_attn_fwd -> a v GEMM -> a v_pts GEMM -> fd_ipa_opt_fwd  of
// network.ipa_fwd, whose [B, 8, N, N] logits / probabilities round-trip HBM twice.
//
// Decomposition.  A block owns a tile of TI = 16 query rows of one backbone and HPB heads, one wave per head (HPB = 8: all
// heads, 512 threads, one block per CU; HPB = 4 / 2 for launches with few query tiles -- a lone backbone).  The keys are
// walked in tiles of 16 with a running maximum / denominator per (row, head) ("online" softmax), so any N fits and no
// per-row buffer scales with N.  Per key tile and wave (fp32 MFMA 16x16x4, exact fp32 products, fmaf chains):
//   S^T[j, i]   = K_h[j, :] Q_h[i, :]^T                                          64 MFMAs   (K straight from L2 as A operand,
//                                                                                           Q^T resident as B operand: 64 VGPRs)
//   + the point term as gamma (q'.k' - 0.5 |k'|^2) with q' = T_i q_p - c, k' = T_j k_p - c, c = the tile's own translation:
//     |q' - k'|^2 = |q'|^2 + |k'|^2 - 2 q'.k', the |q'|^2 part is constant along j and cancels in the softmax; the
//     norm rides as a 25th contraction channel (K side -0.5 |k'|^2, Q side gamma)               8 MFMAs
//     (centred on the tile: |q'| is a few residues wide, so the rounding error of the rewrite is that of the direct form,
//      eps * gamma * d^2 -- measured against float64 in tests/test_ipa_flash.py)
//   + sqrt(1/3) b_ij^h from the LDS image of the tile's zb rows, + mask term, online-softmax update (VALU, 2 lane exchanges)
//   O^T[c, i]  += V_h[j, c]^T P^T[j, i]                                            64 MFMAs   (the C layout of S^T IS the B layout
//   O_pt^T     += V_pts[j, :]^T P^T                                                16 MFMAs    of P^T: no transpose, no LDS)
//   o_pair[i][h, c] += E_i[h, j] ZD_i[j, c]  (rows split over the waves; E through LDS)       16 MFMAs per wave
// The zb rows of the tile ([16 rows, 16 keys, 40] fp32 = 40 KB per key tile; 160 N bytes per query row in total -- the
// per-pair operand that makes IPA HBM-bound) stream through a two-stage LDS ring by LDS-DMA (global_load_lds_dwordx4, no
// VGPR round trip), one stage ahead; the image pads every 4 keys by 64 bytes and every row by 16 so that the o_pair
// operand reads (16 channels x 4 keys per instruction) hit 32 distinct banks.
// Epilogue: o / l, R_i^T (o_pt - t_i) and its norm (fd_ipa_opt_fwd's arithmetic), o_pair / l -> feats [R, 2688].
// Training (A != nullptr): the logits are written to A [B, 8, N, N] on the way and turned into probabilities by the lane that
// wrote them once the row's maximum and denominator are final -- the backward kernels read A as before.
// Key split (fd_ipa_flash_fwd_split, inference on a long lone backbone): KS blocks share a query tile, each walks 1 / KS of the
// key tiles and leaves its unnormalised sums with (maximum, denominator) in a workspace; ipa_flash_merge_kernel (one block per
// residue) combines the splits with the usual exp(m_k - M) weights and applies the epilogue.
// The second kernel of this file, ipa_flash_bwd_kernel, is the query side of the backward on the same skeleton (see there).
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

constexpr int H = 8, C = 256, PQ = 8, PV = 12, CZ4 = 32, ZB = 40;
constexpr int LDP = H * C * 3 + H * PQ * 3 + H * (PQ + PV) * 3;  // 6816
constexpr int KV_OFF = H * C;                                     // 2048: [k 256 | v 256] per head
constexpr int LDF = H * (C + 4 * PV + CZ4);                       // 2688
constexpr int F_PT = H * C;                                       // 2048
constexpr int F_NORM = F_PT + 3 * H * PV;                         // 2336
constexpr int F_PAIR = F_NORM + H * PV;                           // 2432
constexpr int MAXN = 1024;

constexpr int TI = 16;                         // query rows per block = key rows per tile (one 16x16 MFMA tile)
constexpr int GRP_PIECES = 44;                 // 4 keys x 10 pieces of 16 bytes + 4 pad pieces
constexpr int ROW_PIECES = 4 * GRP_PIECES + 1; // 177 (one pad piece per row)
constexpr int STAGE_PIECES = TI * ROW_PIECES;  // 2832
constexpr int ROW_F = ROW_PIECES * 4;          // 708 floats
constexpr int GRP_F = GRP_PIECES * 4;          // 176 floats
constexpr int STAGE_INSTR = (STAGE_PIECES + 63) / 64;   // 45 wave-wide copies per stage
// timing-only ablations of tools/bench_ipa_flash.py --variants (WRONG RESULTS by design; they compile only in a probe build:
// csrc/fd_probe.h)
#if defined(FL_ABL_NODMA) || defined(FL_ABL_NOBAR) || defined(FL_ABL_NOKV) || defined(FL_ABL_KVB0) || defined(FL_ABL_NOSM) || \
    defined(FL_ABL_NOPAIR) || defined(FL_ABL_NODZB) || defined(FL_ABL_NOPTS)
#include "fd_probe.h"
#endif
#ifdef FL_ABL_NODMA
#define FL_DMA(src, dst)
#else
#define FL_DMA(src, dst) fd::glds16a(src, dst)
#endif
#ifdef FL_ABL_NOBAR
#define FL_SYNC()
#else
#define FL_SYNC() __syncthreads()
#endif
#ifndef FL_KPF
#define FL_KPF 4    // K chunks (one 16-byte load per lane each) requested ahead of their MFMAs (probe: -DFL_KPF=..)
#endif
#ifndef FL_VPF
#define FL_VPF 2    // V loads requested ahead
#endif

struct FlashArgs {
  const float *proj, *zb, *qp, *kp, *vp, *head_w, *mask, *quat, *trans;
  float *feats, *A;
  int B, N;
  // key split (a lone backbone has too few query tiles to fill the chip): KS > 1 blocks share a query tile, block ks walks
  // the key tiles [ks nti / KS, (ks + 1) nti / KS) and leaves its unnormalised sums + (max, denominator) in
  // part [KS][R][8][PART_LD]; ipa_flash_merge_kernel combines them.  KS == 1: part unused, the kernel writes feats itself.
  float* part;
  int KS;
};
constexpr int PART_O = 0, PART_PT = C, PART_PAIR = C + PV * 3, PART_ML = C + PV * 3 + CZ4, PART_LD = 328;

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

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#ifdef FL_ABL_NOKV
__device__ __forceinline__ float4 ldkv(const float* p) { const float x = (float)((unsigned long long)p & 15u); return make_float4(x, x, x, x); }
#else
__device__ __forceinline__ float4 ldkv(const float* p) { return ld4(p); }
#endif
__device__ __forceinline__ float f4(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ void scale4(f32x4& v, float f) { v[0] *= f; v[1] *= f; v[2] *= f; v[3] *= f; }
__device__ __forceinline__ f32x4 zero4() { f32x4 z; z[0] = z[1] = z[2] = z[3] = 0.f; return z; }

// LDS piece P of a stage <- (row, key, 16-byte piece) of the zb rows i0 .. i0+15 of the backbone: the float offset of the
// piece for key tile 0 (tile t adds 640 t).  Pad pieces repeat the row's first piece; rows beyond N repeat row N - 1; keys
// beyond N run into the next row (finite values whose weight is exactly zero) and the caller clamps the offset against
// the backbone's last piece, so nothing is read outside the tensor.
__device__ __forceinline__ unsigned piece_off(int P, int i0, int N) {
  const int row = P < STAGE_PIECES ? P / ROW_PIECES : 0;
  const int w = P % ROW_PIECES;
  const int u = w % GRP_PIECES;
  const bool pad = P >= STAGE_PIECES || w == ROW_PIECES - 1 || u >= 40;
  const int key = pad ? 0 : 4 * (w / GRP_PIECES) + u / 10;
  const int q = pad ? 0 : u % 10;
  const int i = imin(i0 + row, N - 1);
  return ((unsigned)i * (unsigned)N + (unsigned)key) * ZB + 4u * (unsigned)q;
}

template <int HPB>
__global__ __launch_bounds__(HPB * 64) void ipa_flash_fwd_kernel(FlashArgs a) {
  constexpr int NI = (STAGE_INSTR + HPB - 1) / HPB;   // LDS-DMA instructions per wave and stage
  constexpr int STAGE_BYTES = HPB * NI * 1024;
  constexpr int RPW = TI / HPB;                        // o_pair rows per wave
  constexpr int NG = H / HPB;                          // head groups
  __shared__ __attribute__((aligned(16))) char slab[2 * STAGE_BYTES];
  __shared__ __attribute__((aligned(16))) float Es[HPB][TI][16];     // unnormalised probabilities of the key tile [head][i][key]
  __shared__ __attribute__((aligned(16))) float Fs[TI][16];          // rescale factor of the tile [i][head] (pad heads: 1)
  __shared__ __attribute__((aligned(16))) float Ls[TI][16];          // 1 / denominator [i][head]
  __shared__ __attribute__((aligned(16))) float mask_s[MAXN + TI];

  const int N = a.N;
  const int nti = (N + TI - 1) / TI;
  const int lid0 = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int KS = a.KS, ksi = lid0 % KS, lid = lid0 / KS;
  const int g = lid % NG, it = (lid / NG) % nti, b = lid / (NG * nti);
  const int t0 = (int)((long)ksi * nti / KS), t1 = (int)((long)(ksi + 1) * nti / KS);      // this block's key tiles
  const int lane = fd::lane_id();
  const int wave = fd::uniform(fd::wave_id());     // (wave-uniform: the head's pointers live in SGPRs)
  const int n = lane & 15, kk = lane >> 4;
  const int h = g * HPB + wave;
  const int i0 = it * TI;
  const long rb = (long)b * N;                               // first residue row of the backbone
  const long rg = rb + imin(i0 + n, N - 1);                   // this lane's query row (clamped)
  const float* __restrict__ zb_b = a.zb + rb * N * ZB;

  // ---- block setup
  for (int j = (int)threadIdx.x; j < nti * TI; j += HPB * 64) mask_s[j] = j < N ? a.mask[rb + j] : 0.f;
  for (int e = (int)threadIdx.x; e < TI * 16; e += HPB * 64) { (&Fs[0][0])[e] = 1.f; (&Ls[0][0])[e] = 1.f; }
  // first stage of the zb image
  unsigned poff[NI];
  const unsigned plim = (unsigned)N * (unsigned)N * ZB - 4u;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int inst = wave * NI + k;
    poff[k] = piece_off(inst * 64 + lane, i0, N);
    if (inst < STAGE_INSTR) FL_DMA(zb_b + umin(poff[k] + (unsigned)t0 * (TI * ZB), plim), slab + (t0 & 1) * STAGE_BYTES + inst * 1024);
  }

  // ---- per-wave operands that stay in registers
  const float sc = sqrtf(1.0f / (3.0f * (float)C));
  const float sq13 = sqrtf(1.0f / 3.0f);
  const float gamma = softplus_f(a.head_w[h]) * sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
  const float mi = a.mask[rg];
  // Q^T as B operand: lane (n = query row, kk) holds channels 16 cc + 4 kk .. + 3 of chunk cc; k-step s of a chunk contracts
  // the channels {16 cc + 4 kk' + s}, the same set on the K side
  float4 Qf[C / 16];
  {
    const float* q = a.proj + rg * LDP + h * C + 4 * kk;
#pragma unroll
    for (int cc = 0; cc < C / 16; ++cc) Qf[cc] = ld4(q + 16 * cc);
  }
  // point operands: 24 floats (8 points x xyz) per (row, head) -> chunk 0 = floats 0..15, chunk 1 = floats 16..23 | norm | 0.
  // The centre component of float f is f % 3; lane kk holds floats 4 kk + e of chunk 0 (component (kk + e) % 3) and
  // 16 + 4 kk + e of chunk 1 (component (kk + e + 1) % 3): three registers d0..d2 = c[(kk + 0..2) % 3] cover both.
  const float* tc = a.trans + (rb + imin(i0 + TI / 2, N - 1)) * 3;
  const float c0 = tc[0], c1 = tc[1], c2 = tc[2];
  const int k3 = kk % 3;
  const float d0 = k3 == 0 ? c0 : k3 == 1 ? c1 : c2;
  const float d1 = k3 == 0 ? c1 : k3 == 1 ? c2 : c0;
  const float d2 = k3 == 0 ? c2 : k3 == 1 ? c0 : c1;
  const float* __restrict__ qpr = a.qp + (rg * H + h) * (PQ * 3) + 4 * kk;

  f32x4 O[C / 16], OP[4], PA[RPW][2];
#pragma unroll
  for (int c = 0; c < C / 16; ++c) O[c] = zero4();
#pragma unroll
  for (int c = 0; c < 4; ++c) OP[c] = zero4();
#pragma unroll
  for (int r = 0; r < RPW; ++r) { PA[r][0] = zero4(); PA[r][1] = zero4(); }
  float m_run = -INFINITY, l_run = 0.f;
  const bool row_ok = i0 + n < N;
  float* __restrict__ Arow = a.A != nullptr ? a.A + (((long)b * H + h) * N + imin(i0 + n, N - 1)) * N : nullptr;
  const bool avec = (N & 3) == 0 && (((unsigned long long)a.A) & 15ull) == 0;       // rows of A are 16-byte aligned

  // K fragments are requested KPF chunks ahead of the MFMAs that consume them, V fragments VPF loads ahead (a wave has one
  // partner on its SIMD and an L2 round trip is ~20 MFMA issue slots); the first KPF chunks of the NEXT key tile are
  // requested before the o_pair phase of this one.
  constexpr int KPF = HPB == 8 ? FL_KPF : 16, VPF = HPB == 8 ? FL_VPF : 10;     // (HPB < 8: one wave per SIMD, 512 registers)
  float4 kf[C / 16];
#ifdef FL_ABL_KVB0
  const float* __restrict__ kbase = a.proj + KV_OFF + h * 2 * C;                 // (probe: every backbone reads backbone 0's K / V)
#else
  const float* __restrict__ kbase = a.proj + rb * LDP + KV_OFF + h * 2 * C;      // (wave-uniform; lane offsets are 32-bit)
#endif
  const float* __restrict__ vpb = a.vp + (rb * H + h) * (PV * 3);
  {
    const float* kr = kbase + (unsigned)(imin(TI * t0 + n, N - 1) * LDP + 4 * kk);
#pragma unroll
    for (int cc = 0; cc < KPF; ++cc) kf[cc] = ldkv(kr + 16 * cc);
  }

  // the copy of the next key tile's zb image (its stage was last read in the o_pair phase of tile t - 1)
  auto next_stage = [&](int t) {
    if (t + 1 < t1) {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int inst = wave * NI + k;
        if (inst < STAGE_INSTR)
          FL_DMA(zb_b + umin(poff[k] + (unsigned)(t + 1) * (TI * ZB), plim), slab + ((t + 1) & 1) * STAGE_BYTES + inst * 1024);
      }
    }
  };
#define FL_NEXT_STAGE() next_stage(t)

#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
    fd::wait_vmem();
    FL_SYNC();                 // stage t of the image has landed; every wave is done with tile t - 1 (its stage, Es, Fs)
    FL_NEXT_STAGE();      // (behind the tile's last K request instead: measured slower, 122 against 109 us at B=30 x N=128)
    const float* __restrict__ sl = reinterpret_cast<const float*>(slab + (t & 1) * STAGE_BYTES);
    const int j0 = TI * t;
    // ---- S^T = K Q^T (two accumulator chains) and the point term
    f32x4 s0 = zero4(), s1 = zero4(), spt = zero4();
    {
      const int jl = imin(j0 + n, N - 1);
      const float* kr = kbase + (unsigned)(jl * LDP + 4 * kk);
      const float* kpr = a.kp + ((rb + jl) * H + h) * (PQ * 3) + 4 * kk;
      float4 u, v = make_float4(0.f, 0.f, 0.f, 0.f), qa, qb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int cc = 0; cc < C / 16; cc += 2) {
        if (cc == C / 16 - 2 * KPF || (C / 16 <= 2 * KPF && cc == 0)) {       // the point operands, as far ahead as a K chunk
          u = ld4(kpr); qa = ld4(qpr);
          if (kk < 2) { v = ld4(kpr + 16); qb = ld4(qpr + 16); }
        }
        if (cc + KPF < C / 16) { kf[cc + KPF] = ldkv(kr + 16 * (cc + KPF)); kf[cc + KPF + 1] = ldkv(kr + 16 * (cc + KPF + 1)); }
        const float4 k0 = kf[cc], k1 = kf[cc + 1];
        s0 = fd::mfma_16x16x4(k0.x, Qf[cc].x, s0);
        s1 = fd::mfma_16x16x4(k1.x, Qf[cc + 1].x, s1);
        s0 = fd::mfma_16x16x4(k0.y, Qf[cc].y, s0);
        s1 = fd::mfma_16x16x4(k1.y, Qf[cc + 1].y, s1);
        s0 = fd::mfma_16x16x4(k0.z, Qf[cc].z, s0);
        s1 = fd::mfma_16x16x4(k1.z, Qf[cc + 1].z, s1);
        s0 = fd::mfma_16x16x4(k0.w, Qf[cc].w, s0);
        s1 = fd::mfma_16x16x4(k1.w, Qf[cc + 1].w, s1);
      }
      u = make_float4(u.x - d0, u.y - d1, u.z - d2, u.w - d0);
      qa = make_float4(gamma * (qa.x - d0), gamma * (qa.y - d1), gamma * (qa.z - d2), gamma * (qa.w - d0));
      if (kk < 2) {
        v = make_float4(v.x - d1, v.y - d2, v.z - d0, v.w - d1);
        qb = make_float4(gamma * (qb.x - d1), gamma * (qb.y - d2), gamma * (qb.z - d0), gamma * (qb.w - d1));
      }
      float nsq = u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w + v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      nsq += __shfl_xor(nsq, 16);
      nsq += __shfl_xor(nsq, 32);
      if (kk == 2) { v.x = -0.5f * nsq; qb.x = gamma; }
      spt = fd::mfma_16x16x4(u.x, qa.x, spt);
      spt = fd::mfma_16x16x4(u.y, qa.y, spt);
      spt = fd::mfma_16x16x4(u.z, qa.z, spt);
      spt = fd::mfma_16x16x4(u.w, qa.w, spt);
      spt = fd::mfma_16x16x4(v.x, qb.x, spt);
      spt = fd::mfma_16x16x4(v.y, qb.y, spt);
      spt = fd::mfma_16x16x4(v.z, qb.z, spt);
      spt = fd::mfma_16x16x4(v.w, qb.w, spt);
    }
    // ---- the first V fragments are on their way while the softmax runs.  Load l = 4 r + cb4 of the tile: keys
    // {j0 + 4 kk' + r}, channels 64 cb4 + 4 m + q (m = lane & 15) -> O tiles 4 cb4 + q
    // V work items l = 5 r + c: c < 4 -> channels 64 c + 4 m + q of the keys {j0 + 4 kk' + r} (O tiles 4 c + q), c = 4 -> the
    // 36 point floats (4 m + q, m < 9) of the same keys (O_pt tiles q)
    const float* __restrict__ vbase = kbase + C + 4 * n;
    float4 vf[20];
    unsigned voff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) voff[r] = (unsigned)imin(j0 + 4 * kk + r, N - 1);
    auto vload = [&](int l) -> float4 {
      const int r = l / 5, c = l % 5;
      if (c < 4) return ldkv(vbase + voff[r] * LDP + 64 * c);
      return n < 9 ? ld4(vpb + voff[r] * (H * PV * 3) + 4 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int l = 0; l < VPF; ++l) vf[l] = vload(l);
    // ---- logits of (row n, keys j0 + 4 kk + r), online softmax
    const float4 mj = ld4(mask_s + j0 + 4 * kk);
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = (s0[r] + s1[r]) * sc + sq13 * sl[n * ROW_F + kk * GRP_F + r * ZB + h];
      x = x + spt[r];
      x = x + 1e5f * (mi * f4(mj, r) - 1.f);
      s[r] = j0 + 4 * kk + r < N ? x : -INFINITY;
    }
    if (Arow != nullptr && row_ok) {
      const int j = j0 + 4 * kk;
      if (avec && j + 3 < N) {
        *reinterpret_cast<float4*>(Arow + j) = make_float4(s[0], s[1], s[2], s[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j + r < N) Arow[j + r] = s[r];
      }
    }
    float tmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
#ifdef FL_ABL_NOSM
    const float fsc = 1.f + 1e-9f * m_new;
#else
    const float fsc = expf(m_run - m_new);
#endif
    float p[4];
#pragma unroll
#ifdef FL_ABL_NOSM
    for (int r = 0; r < 4; ++r) p[r] = s[r] * 1e-3f;
#else
    for (int r = 0; r < 4; ++r) p[r] = expf(s[r] - m_new);
#endif
    float ps = (p[0] + p[1]) + (p[2] + p[3]);
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run = l_run * fsc + ps;
    m_run = m_new;
    *reinterpret_cast<float4*>(&Es[wave][n][4 * kk]) = make_float4(p[0], p[1], p[2], p[3]);
    if (kk == 0) Fs[n][wave] = fsc;
#pragma unroll
    for (int c = 0; c < C / 16; ++c) scale4(O[c], fsc);
#pragma unroll
    for (int c = 0; c < 4; ++c) scale4(OP[c], fsc);
    // ---- O^T += V^T P^T, O_pt^T += V_pts^T P^T: k-step r contracts the keys {j0 + 4 kk' + r}
#pragma unroll
    for (int l = 0; l < 20; ++l) {
      if (l + VPF < 20) vf[l + VPF] = vload(l + VPF);
      const int r = l / 5, c = l % 5;
      if (c < 4) {
        O[4 * c + 0] = fd::mfma_16x16x4(vf[l].x, p[r], O[4 * c + 0]);
        O[4 * c + 1] = fd::mfma_16x16x4(vf[l].y, p[r], O[4 * c + 1]);
        O[4 * c + 2] = fd::mfma_16x16x4(vf[l].z, p[r], O[4 * c + 2]);
        O[4 * c + 3] = fd::mfma_16x16x4(vf[l].w, p[r], O[4 * c + 3]);
      } else {
        OP[0] = fd::mfma_16x16x4(vf[l].x, p[r], OP[0]);
        OP[1] = fd::mfma_16x16x4(vf[l].y, p[r], OP[1]);
        OP[2] = fd::mfma_16x16x4(vf[l].z, p[r], OP[2]);
        OP[3] = fd::mfma_16x16x4(vf[l].w, p[r], OP[3]);
      }
    }
    // the next key tile's first K fragments (in flight across the barrier and the o_pair phase)
    if (t + 1 < t1) {
      const float* kr = kbase + (unsigned)(imin(j0 + TI + n, N - 1) * LDP + 4 * kk);
#pragma unroll
      for (int cc = 0; cc < KPF; ++cc) kf[cc] = ldkv(kr + 16 * cc);
    }
#ifndef FL_ABL_NOPAIR
    FL_SYNC();                 // Es, Fs of this tile are visible
    // ---- o_pair of the wave's rows: [16 (heads, HPB used) x 16 keys] x [16 keys x 32]
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int i = wave * RPW + rr;
      const float4 fr = ld4(&Fs[i][4 * kk]);
      float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < HPB) e4 = ld4(&Es[n][i][4 * kk]);
      const float* zd = sl + i * ROW_F + kk * GRP_F + H + n;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 acc = PA[rr][ct];
        acc[0] *= fr.x; acc[1] *= fr.y; acc[2] *= fr.z; acc[3] *= fr.w;
        acc = fd::mfma_16x16x4(e4.x, zd[0 * ZB + 16 * ct], acc);
        acc = fd::mfma_16x16x4(e4.y, zd[1 * ZB + 16 * ct], acc);
        acc = fd::mfma_16x16x4(e4.z, zd[2 * ZB + 16 * ct], acc);
        acc = fd::mfma_16x16x4(e4.w, zd[3 * ZB + 16 * ct], acc);
        PA[rr][ct] = acc;
      }
    }
#endif
  }

  // ---- key split: unnormalised sums + (max, denominator) to the workspace; the merge launch finishes the job
  if (KS > 1) {
    if (row_ok) {
      float* __restrict__ po = a.part + (((long)ksi * a.B * N + rg) * H + h) * PART_LD;
#pragma unroll
      for (int cb4 = 0; cb4 < 4; ++cb4)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float4*>(po + PART_O + 64 * cb4 + 16 * kk + 4 * r) =
              make_float4(O[4 * cb4][r], O[4 * cb4 + 1][r], O[4 * cb4 + 2][r], O[4 * cb4 + 3][r]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * kk + r < 9)
          *reinterpret_cast<float4*>(po + PART_PT + 16 * kk + 4 * r) = make_float4(OP[0][r], OP[1][r], OP[2][r], OP[3][r]);
      if (kk == 0) { po[PART_ML] = m_run; po[PART_ML + 1] = l_run; }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int i = wave * RPW + rr;
      if (i0 + i < N) {
        float* __restrict__ pp = a.part + (((long)ksi * a.B * N + rb + i0 + i) * H + g * HPB) * PART_LD + PART_PAIR + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int hl = 4 * kk + r;
          if (hl < HPB) {
            pp[hl * PART_LD] = PA[rr][0][r];
            pp[hl * PART_LD + 16] = PA[rr][1][r];
          }
        }
      }
    }
    return;
  }
  // ---- epilogue
  const float inv = 1.0f / l_run;
  if (kk == 0) Ls[n][wave] = inv;
  if (row_ok) {
    float* __restrict__ fo = a.feats + rg * LDF + h * C + 16 * kk;
#pragma unroll
    for (int cb4 = 0; cb4 < 4; ++cb4)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(fo + 64 * cb4 + 4 * r) =
            make_float4(O[4 * cb4][r] * inv, O[4 * cb4 + 1][r] * inv, O[4 * cb4 + 2][r] * inv, O[4 * cb4 + 3][r] * inv);
  }
  __syncthreads();             // the last tile's o_pair reads of the image are done; Ls is visible
  float* __restrict__ og = reinterpret_cast<float*>(slab);       // [HPB][16][36] global-frame sums, normalised
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * kk + r < 9)
      *reinterpret_cast<float4*>(og + (wave * TI + n) * (PV * 3) + 16 * kk + 4 * r) =
          make_float4(OP[0][r] * inv, OP[1][r] * inv, OP[2][r] * inv, OP[3][r] * inv);
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int i = wave * RPW + rr;
    if (i0 + i < N) {
      const float4 li = ld4(&Ls[i][4 * kk]);
      float* __restrict__ fp = a.feats + (rb + i0 + i) * LDF + F_PAIR + g * HPB * CZ4 + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int hl = 4 * kk + r;
        if (hl < HPB) {
          fp[hl * CZ4] = PA[rr][0][r] * f4(li, r);
          fp[hl * CZ4 + 16] = PA[rr][1][r] * f4(li, r);
        }
      }
    }
  }
  __syncthreads();
  for (int e = (int)threadIdx.x; e < HPB * TI * PV; e += HPB * 64) {
    const int hl = e / (TI * PV), i = (e / PV) % TI, pt = e % PV;
    if (i0 + i >= N) continue;
    const long r = rb + i0 + i;
    const Rot R = quat_to_rot(a.quat + r * 4);
    const float* tt = a.trans + r * 3;
    const float* gs = og + (hl * TI + i) * (PV * 3) + 3 * pt;
    const float ux = gs[0] - tt[0], uy = gs[1] - tt[1], uz = gs[2] - tt[2];
    const float lx = R.r[0] * ux + R.r[3] * uy + R.r[6] * uz;
    const float ly = R.r[1] * ux + R.r[4] * uy + R.r[7] * uz;
    const float lz = R.r[2] * ux + R.r[5] * uy + R.r[8] * uz;
    float* f = a.feats + r * LDF;
    const int hp = (g * HPB + hl) * PV + pt;
    f[F_PT + hp] = lx;
    f[F_PT + H * PV + hp] = ly;
    f[F_PT + 2 * H * PV + hp] = lz;
    f[F_NORM + hp] = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
  }
  // ---- training: logits -> probabilities, by the lane that wrote them
  if (Arow != nullptr && row_ok) {
    for (int t = 0; t < nti; ++t) {
      const int j = TI * t + 4 * kk;
      if (avec && j + 3 < N) {
        const float4 v = ld4(Arow + j);
        *reinterpret_cast<float4*>(Arow + j) =
            make_float4(expf(v.x - m_run) * inv, expf(v.y - m_run) * inv, expf(v.z - m_run) * inv, expf(v.w - m_run) * inv);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j + r < N) Arow[j + r] = expf(Arow[j + r] - m_run) * inv;
      }
    }
  }
}


// Key-split epilogue: one block per residue, thread groups of 32 per head.  part [KS][R][8][PART_LD] ->
// feats [R, 2688] (o, o_pt + norm in the residue's frame, o_pair).
__global__ __launch_bounds__(256) void ipa_flash_merge_kernel(const float* __restrict__ part, const float* __restrict__ quat,
                                                              const float* __restrict__ trans, float* __restrict__ feats,
                                                              long R_, int KS) {
  __shared__ float og[H][PV * 3];
  const long r = blockIdx.x;
  const int h = (int)threadIdx.x >> 5, l = (int)threadIdx.x & 31;
  const float* __restrict__ p0 = part + (r * H + h) * PART_LD;
  const long ks_stride = R_ * H * PART_LD;
  float M = -INFINITY;
  for (int k = 0; k < KS; ++k) M = fmaxf(M, p0[k * ks_stride + PART_ML]);
  float L = 0.f;
  for (int k = 0; k < KS; ++k) L += expf(p0[k * ks_stride + PART_ML] - M) * p0[k * ks_stride + PART_ML + 1];
  const float inv = 1.0f / L;
  float* __restrict__ f = feats + r * LDF;
  // 324 values per head (256 o | 36 o_pt sums | 32 o_pair): lane l takes l, l + 32, ...
  for (int e = l; e < PART_ML; e += 32) {
    float acc = 0.f;
    for (int k = 0; k < KS; ++k) acc += expf(p0[k * ks_stride + PART_ML] - M) * p0[k * ks_stride + e];
    acc *= inv;
    if (e < PART_PT) f[h * C + e] = acc;
    else if (e < PART_PAIR) og[h][e - PART_PT] = acc;
    else f[F_PAIR + h * CZ4 + (e - PART_PAIR)] = acc;
  }
  __syncthreads();
  if ((int)threadIdx.x < H * PV) {
    const int hp = (int)threadIdx.x, hh = hp / PV, pt = hp % PV;
    const Rot Rm = quat_to_rot(quat + r * 4);
    const float* tt = trans + r * 3;
    const float ux = og[hh][3 * pt] - tt[0], uy = og[hh][3 * pt + 1] - tt[1], uz = og[hh][3 * pt + 2] - tt[2];
    const float lx = Rm.r[0] * ux + Rm.r[3] * uy + Rm.r[6] * uz;
    const float ly = Rm.r[1] * ux + Rm.r[4] * uy + Rm.r[7] * uz;
    const float lz = Rm.r[2] * ux + Rm.r[5] * uy + Rm.r[8] * uz;
    f[F_PT + hp] = lx;
    f[F_PT + H * PV + hp] = ly;
    f[F_PT + 2 * H * PV + hp] = lz;
    f[F_NORM + hp] = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward, query side (reference: autograd of ipa_pytorch.py:380-457).  With the probabilities A saved by the forward,
//   dP_ij^h = dO_i . v_j + dOpt_i . vpts_j + dout_i^h . zd_ij          (gradient w.r.t. a_ij^h of o, o_pt (global frame), o_pair)
//   dL_ij^h = A_ij^h (dP_ij^h - D_i^h),   D_i^h = sum_j A dP = dO_i . o_i + dOpt_i . opt_i + dout_i^h . opair_i^h
// (D from the forward's own outputs: no second pass over the keys), and from dL
//   dzb[i, j, h] = sqrt(1/3) dL,  dzb[i, j, 8 + c] = sum_h A_ij^h dout_i^h[c],  dqp_i = gamma sum_j dL_ij (k'_j - q'_i),
//   d gamma_h (per-row partial) = -1/2 sum_j dL_ij |q'_i - k'_j|^2
// in ONE launch with the forward kernel's skeleton (16 query rows x HPB heads per block, key tiles of 16, the tile's zb rows
// through the LDS image) -- it replaces the dA GEMM (dO V^T), the dA += GEMM (dOpt vpts^T) and fd_ipa_attn_bwd's per-row
// kernel; dA never exists, dL is written once (for the dK GEMM, fd_ipa_kpts_bwd and dQ GEMM that follow).
// Per key tile and wave: dP^T = V dO^T (64 MFMAs, dO^T resident as B operand) + vpts dOpt^T (12) + the o_pair term as a
// per-row product [heads x 32] x [32 x 16 keys] of the wave's own two rows (16, results handed to the head waves through
// LDS); sum_j dL k'_j with |k'_j|^2 riding as a 25th row (16); dzb's o_pair columns as [16 keys x heads] x [heads x 32] per
// row (8, probabilities of all heads through LDS).  The per-tile exchange buffers are double-buffered so that a tile needs
// two block barriers.
struct FlashBwdArgs {
  const float *proj, *A, *zb, *dfeats, *feats, *doptg, *ptdot, *qp, *kp, *vp, *head_w, *trans;
  float *dL, *dzb, *dqp, *hw_part;
  int B, N;
};

template <int HPB>
__global__ __launch_bounds__(HPB * 64) void ipa_flash_bwd_kernel(FlashBwdArgs a) {
  constexpr int NI = (STAGE_INSTR + HPB - 1) / HPB;
  constexpr int STAGE_BYTES = HPB * NI * 1024;
  constexpr int RPW = TI / HPB;
  constexpr int NG = H / HPB;
  static_assert(HPB == H, "dzb's o_pair columns sum over all heads: one block owns them");
  __shared__ __attribute__((aligned(16))) char slab[2 * STAGE_BYTES];
  __shared__ __attribute__((aligned(16))) float Es[2][HPB][TI][16];     // probabilities of the key tile [head][i][key]
  __shared__ __attribute__((aligned(16))) float Xs[2][HPB][TI][16];     // o_pair part of dP [head][i][key]
  __shared__ __attribute__((aligned(16))) float Ys[2][HPB][TI][16];     // sqrt(1/3) dL [head][i][key]

  const int N = a.N;
  const int nti = (N + TI - 1) / TI;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int g = lid % NG, it = (lid / NG) % nti, b = lid / (NG * nti);
  const int lane = fd::lane_id();
  const int wave = fd::uniform(fd::wave_id());
  const int n = lane & 15, kk = lane >> 4;
  const int h = g * HPB + wave;
  const int i0 = it * TI;
  const long rb = (long)b * N;
  const long rg = rb + imin(i0 + n, N - 1);
  const bool row_ok = i0 + n < N;
  const bool vec = (N & 3) == 0;                             // rows of A / dL are 16-byte aligned
  const float* __restrict__ zb_b = a.zb + rb * N * ZB;

  unsigned poff[NI];
  const unsigned plim = (unsigned)N * (unsigned)N * ZB - 4u;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int inst = wave * NI + k;
    poff[k] = piece_off(inst * 64 + lane, i0, N);
    if (inst < STAGE_INSTR) FL_DMA(zb_b + umin(poff[k], plim), slab + inst * 1024);
  }

  const float sq13 = sqrtf(1.0f / 3.0f);
  const float gscale = sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
  const float hwv = a.head_w[h];
  const float gamma = softplus_f(hwv) * gscale;
  // ---- resident operands
  // dO^T as B operand (the forward's Q^T layout), and D's first term dO . o on the way
  float4 dOf[C / 16];
  float dsum = 0.f;
  {
    const float* d = a.dfeats + rg * LDF + h * C + 4 * kk;
    const float* o = a.feats + rg * LDF + h * C + 4 * kk;
#pragma unroll
    for (int cc = 0; cc < C / 16; ++cc) {
      dOf[cc] = ld4(d + 16 * cc);
      const float4 ov = ld4(o + 16 * cc);
      dsum += dOf[cc].x * ov.x + dOf[cc].y * ov.y + dOf[cc].z * ov.z + dOf[cc].w * ov.w;
    }
    // dout . opair: 32 floats per (row, head), 8 per lane
    const float* dp = a.dfeats + rg * LDF + F_PAIR + h * CZ4 + 8 * kk;
    const float* op = a.feats + rg * LDF + F_PAIR + h * CZ4 + 8 * kk;
    const float4 d0 = ld4(dp), d1 = ld4(dp + 4), o0 = ld4(op), o1 = ld4(op + 4);
    dsum += d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w;
  }
  dsum += __shfl_xor(dsum, 16);
  dsum += __shfl_xor(dsum, 32);
  const float D = dsum + a.ptdot[rg * H + h];
  // dOpt^T as B operand: 36 floats per (row, head) in chunks of 16 | 16 | 4
  float4 dgf[3];
  {
    const float* d = a.doptg + (rg * H + h) * (PV * 3) + 4 * kk;
    dgf[0] = ld4(d);
    dgf[1] = ld4(d + 16);
    dgf[2] = kk == 0 ? ld4(d + 32) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // dout of the wave's own rows: as A operand [m = head][k = channel] for the o_pair part of dP, and as B operand
  // [k = head][n = channel] for dzb's o_pair columns
  float4 doutA[RPW][2];
  float doutB[RPW][2][2];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const long ri = rb + imin(i0 + wave * RPW + rr, N - 1);
    const float* d = a.dfeats + ri * LDF + F_PAIR + g * HPB * CZ4;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      doutA[rr][cb] = n < HPB ? ld4(d + n * CZ4 + 16 * cb + 4 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) doutB[rr][ks][ct] = 4 * ks + kk < HPB ? d[(4 * ks + kk) * CZ4 + 16 * ct + n] : 0.f;
  }
  // centre of the point coordinates (as the forward); lane m = n of the key-point operand holds floats 4 m + e
  const float* tc = a.trans + (rb + imin(i0 + TI / 2, N - 1)) * 3;
  const float c0 = tc[0], c1 = tc[1], c2 = tc[2];
  const int n3 = n % 3;                                       // component of float 4 n + e is (n + e) % 3
  const float e0 = n3 == 0 ? c0 : n3 == 1 ? c1 : c2;
  const float e1 = n3 == 0 ? c1 : n3 == 1 ? c2 : c0;
  const float e2 = n3 == 0 ? c2 : n3 == 1 ? c0 : c1;

  f32x4 dqk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dqk[q] = zero4();
  float sds = 0.f;
  const float* __restrict__ vbase = a.proj + rb * LDP + KV_OFF + h * 2 * C + C;      // V rows of the head
  const float* __restrict__ kpb = a.kp + (rb * H + h) * (PQ * 3);
  const float* __restrict__ vpb = a.vp + (rb * H + h) * (PV * 3);
  const float* __restrict__ Arow = a.A + (((long)b * H + h) * N + imin(i0 + n, N - 1)) * N;
  float* __restrict__ dLrow = a.dL + (((long)b * H + h) * N + imin(i0 + n, N - 1)) * N;

  // ---- operands of a key tile that are requested one tile ahead (their first use is late in the tile, or a dependent
  // chain of its own): the probabilities, the key points (lanes n >= 6 read lane n - 6's piece and drop it: no branch around
  // the loads) and the value points
  float4 pA, kpn[4], vpn[3];
  auto request = [&](int t) {
    const int j0 = TI * t;
    {
      const int j = imin(j0 + 4 * kk, N - 4 < 0 ? 0 : N - 4);      // (a row tail is re-read scalar below)
      pA = vec ? ld4(Arow + (j & ~3)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int m6 = n < 6 ? n : n - 6 < 6 ? n - 6 : n - 12;
#pragma unroll
    for (int r = 0; r < 4; ++r) kpn[r] = ld4(kpb + (unsigned)(imin(j0 + 4 * kk + r, N - 1) * (H * PQ * 3) + 4 * m6));
    const float* pr = vpb + (unsigned)(imin(j0 + n, N - 1) * (H * PV * 3));
    vpn[0] = ld4(pr + 4 * kk);
    vpn[1] = ld4(pr + 16 + 4 * kk);
    vpn[2] = ld4(pr + 32);                                          // (used by kk == 0)
  };
  request(0);

  // dzb of the wave's rows for key tile t (exchange buffers t & 1): o_pair columns by MFMA, bias columns from Ys
  auto emit_dzb = [&](int t) {
    const int bf = t & 1;
    const int j0 = TI * t;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int i = wave * RPW + rr;
      // transposed product [32 channels x heads] x [heads x 16 keys]: lane (n = key, kk) ends up with the channels 4 kk .. + 3
      // (+ 16) of ITS key -- 16 contiguous bytes of the dzb row per store
      f32x4 z0 = zero4(), z1 = zero4();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float e = 4 * ks + kk < HPB ? Es[bf][4 * ks + kk][i][n] : 0.f;     // B operand [k = head 4 ks + kk][n = key]
        z0 = fd::mfma_16x16x4(doutB[rr][ks][0], e, z0);
        z1 = fd::mfma_16x16x4(doutB[rr][ks][1], e, z1);
      }
#ifdef FL_ABL_NODZB
      if (i0 + i < N && z0[0] == 12345.f) {
#else
      if (i0 + i < N && j0 + n < N) {
#endif
        float* __restrict__ dz = a.dzb + ((rb + i0 + i) * N + j0 + n) * ZB;
        *reinterpret_cast<float4*>(dz + H + 4 * kk) = make_float4(z0[0], z0[1], z0[2], z0[3]);
        *reinterpret_cast<float4*>(dz + H + 16 + 4 * kk) = make_float4(z1[0], z1[1], z1[2], z1[3]);
        // bias columns of the key: heads 2 kk, 2 kk + 1
        *reinterpret_cast<float2*>(dz + 2 * kk) = make_float2(Ys[bf][2 * kk][i][n], Ys[bf][2 * kk + 1][i][n]);
      }
    }
  };

#pragma unroll 1
  for (int t = 0; t < nti; ++t) {
    fd::wait_vmem();
    FL_SYNC();                 // stage t of the image has landed; exchange buffers of tile t - 1 are complete
    if (t + 1 < nti) {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int inst = wave * NI + k;
        if (inst < STAGE_INSTR)
          FL_DMA(zb_b + umin(poff[k] + (unsigned)(t + 1) * (TI * ZB), plim), slab + ((t + 1) & 1) * STAGE_BYTES + inst * 1024);
      }
    }
    const int bf = t & 1;
    const int j0 = TI * t;
    // the first V chunks of this tile are on their way while dzb of the tile before and the o_pair part of dP are formed
    constexpr int VPFB = 6;
    float4 vfr[C / 16];
    const float* vr = vbase + (unsigned)(imin(j0 + n, N - 1) * LDP + 4 * kk);
#pragma unroll
    for (int cc = 0; cc < VPFB; ++cc) vfr[cc] = ldkv(vr + 16 * cc);
    if (t > 0) emit_dzb(t - 1);
    const float* __restrict__ sl = reinterpret_cast<const float*>(slab + (t & 1) * STAGE_BYTES);
    // ---- o_pair part of dP for the wave's rows: [heads x 32] x [32 x 16 keys]; k-step s of channel block cb contracts the
    // channels {16 cb + 4 kk' + s}
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int i = wave * RPW + rr;
      const float* zr = sl + i * ROW_F + (n >> 2) * GRP_F + (n & 3) * ZB + H + 4 * kk;     // lane (kk, key n)
      f32x4 x = zero4();
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const float4 zv = ld4(zr + 16 * cb);
        x = fd::mfma_16x16x4(doutA[rr][cb].x, zv.x, x);
        x = fd::mfma_16x16x4(doutA[rr][cb].y, zv.y, x);
        x = fd::mfma_16x16x4(doutA[rr][cb].z, zv.z, x);
        x = fd::mfma_16x16x4(doutA[rr][cb].w, zv.w, x);
      }
      // C[m = head][n = key]: lane (n, kk) holds heads 4 kk + r
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * kk + r < HPB) Xs[bf][4 * kk + r][i][n] = x[r];
    }
    // ---- probabilities of (row n, keys j0 + 4 kk + r)
    float p[4];
    {
      const int j = j0 + 4 * kk;
      if (vec && j + 3 < N) {
        p[0] = pA.x; p[1] = pA.y; p[2] = pA.z; p[3] = pA.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = j + r < N ? Arow[j + r] : 0.f;
      }
    }
    *reinterpret_cast<float4*>(&Es[bf][wave][n][4 * kk]) = make_float4(p[0], p[1], p[2], p[3]);
    // ---- dP^T = V dO^T + vpts dOpt^T
    f32x4 s0 = zero4(), s1 = zero4(), spt = zero4();
    {
#pragma unroll
      for (int cc = 0; cc < C / 16; cc += 2) {
        if (cc + VPFB < C / 16) { vfr[cc + VPFB] = ldkv(vr + 16 * (cc + VPFB)); vfr[cc + VPFB + 1] = ldkv(vr + 16 * (cc + VPFB + 1)); }
        const float4 v0 = vfr[cc], v1 = vfr[cc + 1];
        s0 = fd::mfma_16x16x4(v0.x, dOf[cc].x, s0);
        s1 = fd::mfma_16x16x4(v1.x, dOf[cc + 1].x, s1);
        s0 = fd::mfma_16x16x4(v0.y, dOf[cc].y, s0);
        s1 = fd::mfma_16x16x4(v1.y, dOf[cc + 1].y, s1);
        s0 = fd::mfma_16x16x4(v0.z, dOf[cc].z, s0);
        s1 = fd::mfma_16x16x4(v1.z, dOf[cc + 1].z, s1);
        s0 = fd::mfma_16x16x4(v0.w, dOf[cc].w, s0);
        s1 = fd::mfma_16x16x4(v1.w, dOf[cc + 1].w, s1);
      }
      const float4 u0 = vpn[0], u1 = vpn[1];
      const float4 u2 = kk == 0 ? vpn[2] : make_float4(0.f, 0.f, 0.f, 0.f);
      spt = fd::mfma_16x16x4(u0.x, dgf[0].x, spt);
      spt = fd::mfma_16x16x4(u0.y, dgf[0].y, spt);
      spt = fd::mfma_16x16x4(u0.z, dgf[0].z, spt);
      spt = fd::mfma_16x16x4(u0.w, dgf[0].w, spt);
      spt = fd::mfma_16x16x4(u1.x, dgf[1].x, spt);
      spt = fd::mfma_16x16x4(u1.y, dgf[1].y, spt);
      spt = fd::mfma_16x16x4(u1.z, dgf[1].z, spt);
      spt = fd::mfma_16x16x4(u1.w, dgf[1].w, spt);
      spt = fd::mfma_16x16x4(u2.x, dgf[2].x, spt);
      spt = fd::mfma_16x16x4(u2.y, dgf[2].y, spt);
      spt = fd::mfma_16x16x4(u2.z, dgf[2].z, spt);
      spt = fd::mfma_16x16x4(u2.w, dgf[2].w, spt);
    }
    FL_SYNC();                 // Xs (and Es) of this tile are visible
    // ---- dL
    const float4 xv = ld4(&Xs[bf][wave][n][4 * kk]);
    float dl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dP = (s0[r] + s1[r]) + spt[r] + f4(xv, r);
      dl[r] = p[r] * (dP - D);
    }
    if (row_ok) {
      const int j = j0 + 4 * kk;
      if (vec && j + 3 < N) {
        *reinterpret_cast<float4*>(dLrow + j) = make_float4(dl[0], dl[1], dl[2], dl[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j + r < N) dLrow[j + r] = dl[r];
      }
    }
    *reinterpret_cast<float4*>(&Ys[bf][wave][n][4 * kk]) = make_float4(sq13 * dl[0], sq13 * dl[1], sq13 * dl[2], sq13 * dl[3]);
    sds += (dl[0] + dl[1]) + (dl[2] + dl[3]);
    // ---- sum_j dL k'_j (24 floats) and sum_j dL |k'_j|^2 (row 6 of tile 0): A operand lane (m = n, kk) = floats 4 m + q of
    // key j0 + 4 kk + r, m < 6
#ifndef FL_ABL_NOPTS
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float4 kf = kpn[r];
      kf = n < 6 ? make_float4(kf.x - e0, kf.y - e1, kf.z - e2, kf.w - e0) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float nsq = fd::row16_sum(kf.x * kf.x + kf.y * kf.y + kf.z * kf.z + kf.w * kf.w);
      if (n == 6) kf.x = nsq;
      dqk[0] = fd::mfma_16x16x4(kf.x, dl[r], dqk[0]);
      dqk[1] = fd::mfma_16x16x4(kf.y, dl[r], dqk[1]);
      dqk[2] = fd::mfma_16x16x4(kf.z, dl[r], dqk[2]);
      dqk[3] = fd::mfma_16x16x4(kf.w, dl[r], dqk[3]);
    }
#endif
    if (t + 1 < nti) request(t + 1);
  }
  FL_SYNC();
  emit_dzb(nti - 1);

  // ---- epilogue: dqp = gamma (sum_j dL k'_j - q'_i sum_j dL), head-weight partial
  sds += __shfl_xor(sds, 16);
  sds += __shfl_xor(sds, 32);
  // C layout of dqk[q]: lane (n = row i, kk), register r -> operand row m = 4 kk + r -> float 4 m + q
  float qdot = 0.f, qsq = 0.f;
  const float* qsrc = a.qp + (rg * H + h) * (PQ * 3);
  float* qdst = a.dqp + (rg * H + h) * (PQ * 3);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * kk + r;
    if (m < 6) {
      float4 q = ld4(qsrc + 4 * m);
      // component of float 4 m + e is (m + e) % 3
      const int m3 = m % 3;
      const float g0 = m3 == 0 ? c0 : m3 == 1 ? c1 : c2;
      const float g1 = m3 == 0 ? c1 : m3 == 1 ? c2 : c0;
      const float g2 = m3 == 0 ? c2 : m3 == 1 ? c0 : c1;
      q = make_float4(q.x - g0, q.y - g1, q.z - g2, q.w - g0);
      const float k0 = dqk[0][r], k1 = dqk[1][r], k2 = dqk[2][r], k3 = dqk[3][r];
      qdot += q.x * k0 + q.y * k1 + q.z * k2 + q.w * k3;
      qsq += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
      if (row_ok)
        *reinterpret_cast<float4*>(qdst + 4 * m) =
            make_float4(gamma * (k0 - q.x * sds), gamma * (k1 - q.y * sds), gamma * (k2 - q.z * sds), gamma * (k3 - q.w * sds));
    }
  }
  qdot += __shfl_xor(qdot, 16);
  qdot += __shfl_xor(qdot, 32);
  qsq += __shfl_xor(qsq, 16);
  qsq += __shfl_xor(qsq, 32);
  const float skn = __shfl(dqk[0][2], n + 16);              // row m = 6 (kk = 1, r = 2) of tile q = 0
  if (kk == 0 && row_ok) {
    const float dgam = -0.5f * (qsq * sds + skn - 2.f * qdot);
    const float sig = hwv > 20.f ? 1.f : 1.f / (1.f + expf(-hwv));
    a.hw_part[rg * H + h] = dgam * gscale * sig;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// KEY side of the IPA attention backward as one kernel (round 6): autograd of model/ipa_pytorch.py:380-457 with respect to the
// keys, the values and their points --
//   dV_j   = sum_i A_ij dO_i              (:424-428)          dv_pts_j = sum_i A_ij dO_pt_i          (:432-441, global frame)
//   dK_j   = sqrt(1/3C) sum_i dL_ij Q_i   (:380-384)          dk_pts_j = gamma sum_i dL_ij (q_pts_i - k_pts_j)   (:398-412)
// from the probabilities A (the forward's) and the logit gradient dL (the query-side kernel's), both [B, 8, N, N].  Replaces three
// batched fp32 GEMMs over those tensors (A^T dO, A^T dO_pt, dL^T Q) and fd_ipa_kpts_bwd of network.ipa_bwd: A and dL are read
// once here instead of three times each, dO / Q stream from L2 as MFMA operands, nothing is staged through LDS.
// Decomposition: a block owns a tile of 16 KEYS of one backbone and HPB heads, one wave per head (the key side has no sum over
// heads, so any HPB works); it walks the query tiles.  Per query tile and wave (v_mfma_f32_16x16x4_f32, exact fp32 products,
// transposed accumulation X^T[channel, key] += operand^T[channel, query] P[query, key] -- the forward's O^T += V^T P^T with the
// roles of queries and keys exchanged):
//   dV^T 64 + dV_pts^T 16 + dK^T 64 + [sum_i dL q_pts ; sum_i dL] 16 MFMAs; the operand of k-step r is the query row 4 kk + r of
//   the tile, channels 64 c + 4 m + q of lane (m = lane & 15, kk = lane >> 4) feed output tile 4 c + q, so every operand load is a
//   float4 along the channels and every lane ends up with 16 consecutive channels of its key per 64-channel group (float4 stores).
// 160 accumulator registers + one operand set in flight: two waves per SIMD.
struct FlashKeysArgs {
  const float *A, *dL, *proj, *dfeats, *doptg, *qp, *kp, *head_w;
  float *dproj, *dvp, *dkp;
  int B, N;
};

template <int HPB>
__global__ __launch_bounds__(HPB * 64, 2) void ipa_flash_bwd_keys_kernel(FlashKeysArgs a) {
  constexpr int NG = H / HPB;
  const int N = a.N;
  const int nti = (N + TI - 1) / TI;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int g = lid % NG, jt = (lid / NG) % nti, b = lid / (NG * nti);
  const int lane = fd::lane_id();
  const int wave = fd::uniform(fd::wave_id());
  const int n = lane & 15, kk = lane >> 4;
  const int h = g * HPB + wave;
  const int j0 = jt * TI;
  const long rb = (long)b * N;
  const int jn = imin(j0 + n, N - 1);                         // this lane's key (clamped; keys past N are not stored)
  const bool key_ok = j0 + n < N;
  const float sc = sqrtf(1.0f / (3.0f * (float)C));
  const float gamma = softplus_f(a.head_w[h]) * sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));

  f32x4 dV[C / 16], dK[C / 16], dP[4], dX[4];
#pragma unroll
  for (int c = 0; c < C / 16; ++c) { dV[c] = zero4(); dK[c] = zero4(); }
#pragma unroll
  for (int c = 0; c < 4; ++c) { dP[c] = zero4(); dX[c] = zero4(); }

  const float* __restrict__ Ab = a.A + ((long)b * H + h) * N * N + jn;        // column jn of the head's [N, N] block
  const float* __restrict__ Lb = a.dL + ((long)b * H + h) * N * N + jn;
  const float* __restrict__ dOb = a.dfeats + rb * LDF + h * C + 4 * n;         // channels 64 c + 4 n + q of a query row
  const float* __restrict__ Qb = a.proj + rb * LDP + h * C + 4 * n;
  const float* __restrict__ dGb = a.doptg + (rb * H + h) * (PV * 3) + 4 * n;   // 36 floats per (row, head): lanes n < 9
  const float* __restrict__ qpb = a.qp + (rb * H + h) * (PQ * 3) + 4 * n;      // 24 floats: lanes n < 6; lane 6 carries the 1 of sum_i dL

  // operands of one k-step = one query row per lane group, requested a row ahead of the MFMAs that consume them (two named
  // register sets of native vectors: a struct of HIP float4s copied whole goes through scratch memory)
  auto ldv = [](const float* p) -> f32x4 { return *reinterpret_cast<const f32x4*>(p); };
  auto request = [&](int i, f32x4 (&o)[4], f32x4 (&q)[4], f32x4& gp, f32x4& pp, float& pa, float& pl) __attribute__((always_inline)) {
    const bool ok = i < N;                                  // (a row past N: clamped address, weight zero)
    const unsigned ic = (unsigned)imin(i, N - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o[c] = ldv(dOb + ic * LDF + 64 * c);
      q[c] = ldv(Qb + ic * LDP + 64 * c);
    }
    f32x4 z = zero4();
    gp = n < 9 ? ldv(dGb + ic * (H * PV * 3)) : z;
    z[0] = n == 6 ? 1.f : 0.f;
    pp = n < 6 ? ldv(qpb + ic * (H * PQ * 3)) : z;
    const float va = Ab[(long)ic * N], vl = Lb[(long)ic * N];
    pa = ok ? va : 0.f;
    pl = ok ? vl : 0.f;
  };
  auto multiply = [&](const f32x4 (&o)[4], const f32x4 (&q)[4], const f32x4& gp, const f32x4& pp, float pa, float pl)
                      __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dV[4 * c + e] = fd::mfma_16x16x4(o[c][e], pa, dV[4 * c + e]);
        dK[4 * c + e] = fd::mfma_16x16x4(q[c][e], pl, dK[4 * c + e]);
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dP[e] = fd::mfma_16x16x4(gp[e], pa, dP[e]);
      dX[e] = fd::mfma_16x16x4(pp[e], pl, dX[e]);
    }
  };

  // k-step s of the walk contracts the query rows {4 s + kk' : kk' = 0..3}: lane group kk supplies row 4 s + kk
  const int nsteps = 4 * nti;                               // (even)
  f32x4 oA[4], qA[4], gA, pA, oB[4], qB[4], gB, pB;
  float paA, plA, paB, plB;
  request(kk, oA, qA, gA, pA, paA, plA);
#pragma unroll 1
  for (int s = 0; s < nsteps; s += 2) {
    request(4 * (s + 1) + kk, oB, qB, gB, pB, paB, plB);
    multiply(oA, qA, gA, pA, paA, plA);
    request(4 * (s + 2) + kk, oA, qA, gA, pA, paA, plA);
    multiply(oB, qB, gB, pB, paB, plB);
  }

  // ---- epilogue.  D layout: register r of tile 4 c + q = channel 64 c + 4 (4 kk + r) + q of key n: the four tiles of a group give
  // 16 consecutive channels 64 c + 16 kk + 4 r + q per lane
  if (key_ok) {
    float* __restrict__ dk = a.dproj + (rb + j0 + n) * LDP + KV_OFF + h * 2 * C + 16 * kk;
    float* __restrict__ dv = dk + C;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<float4*>(dk + 64 * c + 4 * r) =
            make_float4(sc * dK[4 * c][r], sc * dK[4 * c + 1][r], sc * dK[4 * c + 2][r], sc * dK[4 * c + 3][r]);
        *reinterpret_cast<float4*>(dv + 64 * c + 4 * r) = make_float4(dV[4 * c][r], dV[4 * c + 1][r], dV[4 * c + 2][r], dV[4 * c + 3][r]);
      }
  }
  // value points: floats 16 kk + 4 r + q < 36
  if (key_ok) {
    float* __restrict__ dvp = a.dvp + ((rb + j0 + n) * H + h) * (PV * 3) + 16 * kk;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * kk + 4 * r < PV * 3) *reinterpret_cast<float4*>(dvp + 4 * r) = make_float4(dP[0][r], dP[1][r], dP[2][r], dP[3][r]);
  }
  // key points: dk_pts = gamma (sum_i dL q_pts - k_pts sum_i dL); the column sum is float 24 = tile 0, register 2 of lane group 1
  const float csum = __shfl(dX[0][2], n + 16);
  if (key_ok) {
    const float* __restrict__ kpr = a.kp + ((rb + j0 + n) * H + h) * (PQ * 3) + 16 * kk;
    float* __restrict__ dkp = a.dkp + ((rb + j0 + n) * H + h) * (PQ * 3) + 16 * kk;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * kk + 4 * r < PQ * 3) {
        const float4 kv = ld4(kpr + 4 * r);
        *reinterpret_cast<float4*>(dkp + 4 * r) = make_float4(gamma * (dX[0][r] - kv.x * csum), gamma * (dX[1][r] - kv.y * csum),
                                                              gamma * (dX[2][r] - kv.z * csum), gamma * (dX[3][r] - kv.w * csum));
      }
  }
}


// The same key-side backward with the query operands SHARED: a block owns FOUR consecutive key tiles of one head (one wave each), so
// the 16 dO rows and 16 Q rows of a query tile -- 32 KB that every key tile of the head multiplies -- are fetched from L2 once per
// block (global -> registers -> LDS, two stages) and read by all four waves as MFMA operands (ds_read_b128 of a row-major [16][256]
// image: lanes 0..15 of a lane group cover 256 contiguous bytes, the four row groups of an instruction fall into disjoint bank
// windows).  The per-key-tile kernel above moved 503 MB through the CUs per launch at B=30 x N=128 for 63 MB of operands.
__global__ __launch_bounds__(256, 2) void ipa_flash_bwd_keys4_kernel(FlashKeysArgs a) {
  constexpr int KT = 4;                                        // key tiles (= waves) of a block
  __shared__ __attribute__((aligned(16))) float stage[2][2][TI][C];      // [buffer][dO | Q][query row][channel]
  const int N = a.N;
  const int nti = (N + TI - 1) / TI;
  const int njg = (nti + KT - 1) / KT;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int jg = lid % njg, h = (lid / njg) % H, b = lid / (njg * H);
  const int lane = fd::lane_id();
  const int wave = fd::uniform(fd::wave_id());
  const int n = lane & 15, kk = lane >> 4;
  const int jt = jg * KT + wave;                               // (a wave past the last key tile computes on clamped rows, stores nothing)
  const int j0 = jt * TI;
  const long rb = (long)b * N;
  const int jn = imin(j0 + n, N - 1);
  const bool key_ok = j0 + n < N;
  const float sc = sqrtf(1.0f / (3.0f * (float)C));
  const float gamma = softplus_f(a.head_w[h]) * sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));

  f32x4 dV[C / 16], dK[C / 16], dP[4], dX[4];
#pragma unroll
  for (int c = 0; c < C / 16; ++c) { dV[c] = zero4(); dK[c] = zero4(); }
#pragma unroll
  for (int c = 0; c < 4; ++c) { dP[c] = zero4(); dX[c] = zero4(); }

  const float* __restrict__ Ab = a.A + ((long)b * H + h) * N * N + jn;
  const float* __restrict__ Lb = a.dL + ((long)b * H + h) * N * N + jn;
  const float* __restrict__ dGb = a.doptg + (rb * H + h) * (PV * 3) + 4 * n;
  const float* __restrict__ qpb = a.qp + (rb * H + h) * (PQ * 3) + 4 * n;
  // staging: wave w copies rows 4 w .. 4 w + 3 of the tile's dO and Q, lane l the 16 bytes at channel 4 l
  const float* __restrict__ dOs = a.dfeats + rb * LDF + h * C + 4 * lane;
  const float* __restrict__ Qs = a.proj + rb * LDP + h * C + 4 * lane;
  auto ldv = [](const float* p) -> f32x4 { return *reinterpret_cast<const f32x4*>(p); };
  f32x4 so[4], sq[4];
  auto fetch = [&](int it) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned ic = (unsigned)imin(TI * it + 4 * wave + r, N - 1);
      so[r] = ldv(dOs + ic * LDF);
      sq[r] = ldv(Qs + ic * LDP);
    }
  };
  auto deposit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<f32x4*>(&stage[buf][0][4 * wave + r][4 * lane]) = so[r];
      *reinterpret_cast<f32x4*>(&stage[buf][1][4 * wave + r][4 * lane]) = sq[r];
    }
  };
  fetch(0);
  deposit(0);
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < nti; ++it) {
    const int buf = it & 1;
    if (it + 1 < nti) fetch(it + 1);                          // (lands under this tile's 160 MFMAs)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int i = TI * it + 4 * s + kk;                     // this lane group's query row of the k-step
      const bool ok = i < N;
      const unsigned ic = (unsigned)imin(i, N - 1);
      f32x4 z = zero4();
      const f32x4 gp = n < 9 ? ldv(dGb + ic * (H * PV * 3)) : z;
      z[0] = n == 6 ? 1.f : 0.f;
      const f32x4 pp = n < 6 ? ldv(qpb + ic * (H * PQ * 3)) : z;
      const float va = Ab[(long)ic * N], vl = Lb[(long)ic * N];
      const float pa = ok ? va : 0.f, pl = ok ? vl : 0.f;
      const float* orow = &stage[buf][0][4 * s + kk][4 * n];
      const float* qrow = &stage[buf][1][4 * s + kk][4 * n];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 o = ldv(orow + 64 * c), q = ldv(qrow + 64 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dV[4 * c + e] = fd::mfma_16x16x4(o[e], pa, dV[4 * c + e]);
          dK[4 * c + e] = fd::mfma_16x16x4(q[e], pl, dK[4 * c + e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dP[e] = fd::mfma_16x16x4(gp[e], pa, dP[e]);
        dX[e] = fd::mfma_16x16x4(pp[e], pl, dX[e]);
      }
    }
    if (it + 1 < nti) deposit(buf ^ 1);                        // (buffer buf ^ 1 was last read in tile it - 1: a barrier ago)
    __syncthreads();
  }

  if (key_ok) {
    float* __restrict__ dk = a.dproj + (rb + j0 + n) * LDP + KV_OFF + h * 2 * C + 16 * kk;
    float* __restrict__ dv = dk + C;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<float4*>(dk + 64 * c + 4 * r) =
            make_float4(sc * dK[4 * c][r], sc * dK[4 * c + 1][r], sc * dK[4 * c + 2][r], sc * dK[4 * c + 3][r]);
        *reinterpret_cast<float4*>(dv + 64 * c + 4 * r) = make_float4(dV[4 * c][r], dV[4 * c + 1][r], dV[4 * c + 2][r], dV[4 * c + 3][r]);
      }
    float* __restrict__ dvp = a.dvp + ((rb + j0 + n) * H + h) * (PV * 3) + 16 * kk;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * kk + 4 * r < PV * 3) *reinterpret_cast<float4*>(dvp + 4 * r) = make_float4(dP[0][r], dP[1][r], dP[2][r], dP[3][r]);
  }
  const float csum = __shfl(dX[0][2], n + 16);
  if (key_ok) {
    const float* __restrict__ kpr = a.kp + ((rb + j0 + n) * H + h) * (PQ * 3) + 16 * kk;
    float* __restrict__ dkp = a.dkp + ((rb + j0 + n) * H + h) * (PQ * 3) + 16 * kk;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * kk + 4 * r < PQ * 3) {
        const float4 kv = ld4(kpr + 4 * r);
        *reinterpret_cast<float4*>(dkp + 4 * r) = make_float4(gamma * (dX[0][r] - kv.x * csum), gamma * (dX[1][r] - kv.y * csum),
                                                              gamma * (dX[2][r] - kv.z * csum), gamma * (dX[3][r] - kv.w * csum));
      }
  }
}

}  // namespace

extern "C" int fd_ipa_flash_fwd(const float* proj, const float* zb, const float* qp, const float* kp, const float* vp,
                                const float* head_w, const float* mask, const float* quat, const float* trans,
                                float* feats, float* A, int B, int N, int heads_per_block, void* stream) {
  return fd_ipa_flash_fwd_split(proj, zb, qp, kp, vp, head_w, mask, quat, trans, feats, A, B, N, heads_per_block, 1,
                                nullptr, stream);
}

extern "C" int fd_ipa_flash_fwd_split(const float* proj, const float* zb, const float* qp, const float* kp, const float* vp,
                                      const float* head_w, const float* mask, const float* quat, const float* trans,
                                      float* feats, float* A, int B, int N, int heads_per_block, int key_splits,
                                      float* part, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_flash_fwd: N=%d exceeds %d", N, MAXN);
  FD_CHECK_ARG(feats != nullptr, "fd_ipa_flash_fwd: feats is required");
  FD_CHECK_ARG(fd_aligned16(proj) && fd_aligned16(zb) && fd_aligned16(qp) && fd_aligned16(kp) && fd_aligned16(vp) &&
                   fd_aligned16(feats),
               "fd_ipa_flash_fwd: proj, zb, qp, kp, vp and feats must be 16-byte aligned");
  FD_CHECK_ARG(heads_per_block == 0 || heads_per_block == 2 || heads_per_block == 4 || heads_per_block == 8,
               "fd_ipa_flash_fwd: heads_per_block must be 0 (pick), 2, 4 or 8, got %d", heads_per_block);
  const int nti = (N + TI - 1) / TI;
  FD_CHECK_ARG(key_splits >= 1 && (key_splits == 1 || (part != nullptr && fd_aligned16(part) && A == nullptr)),
               "fd_ipa_flash_fwd_split: key_splits > 1 needs a 16-byte aligned workspace `part` of key_splits * B * N * 8 * FD_IPA_FLASH_PART_LD "
               "(= %d) floats and A == NULL", FD_IPA_FLASH_PART_LD);
  if (B == 0 || N == 0) return FD_OK;
  if (key_splits > nti) key_splits = nti;
  const long tiles = (long)B * nti;
  int hpb = heads_per_block;
  if (hpb == 0) hpb = tiles >= 128 ? 8 : tiles >= 32 || key_splits > 1 ? 4 : 2;      // (a lone N = 128 backbone: 8 query tiles)
  FlashArgs a{proj, zb, qp, kp, vp, head_w, mask, quat, trans, feats, A, B, N, part, key_splits};
  const dim3 grid((unsigned)(tiles * (H / hpb) * key_splits));
  if (hpb == 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_fwd_kernel<8>), grid, dim3(512), 0, (hipStream_t)stream, a);
  else if (hpb == 4)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_fwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_fwd_kernel<2>), grid, dim3(128), 0, (hipStream_t)stream, a);
  FD_CHECK_LAUNCH("fd_ipa_flash_fwd");
  if (key_splits > 1) {
    hipLaunchKernelGGL(ipa_flash_merge_kernel, dim3((unsigned)((long)B * N)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)part, quat, trans, feats, (long)B * N, key_splits);
    FD_CHECK_LAUNCH("fd_ipa_flash_fwd (merge)");
  }
  return FD_OK;
}

extern "C" int fd_ipa_flash_bwd(const float* proj, const float* A, const float* zb, const float* dfeats, const float* feats,
                                const float* doptg, const float* ptdot, const float* qp, const float* kp, const float* vp,
                                const float* head_w, const float* trans, float* dL, float* dzb, float* dqp, float* dkp,
                                float* dhead_w, float* hw_part, int B, int N, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_flash_bwd: N=%d exceeds %d", N, MAXN);
  FD_CHECK_ARG(fd_aligned16(proj) && fd_aligned16(A) && fd_aligned16(zb) && fd_aligned16(dfeats) && fd_aligned16(feats) &&
                   fd_aligned16(doptg) && fd_aligned16(qp) && fd_aligned16(kp) && fd_aligned16(vp) && fd_aligned16(dL) &&
                   fd_aligned16(dzb) && fd_aligned16(dqp),
               "fd_ipa_flash_bwd: tensor arguments must be 16-byte aligned");
  if (B == 0 || N == 0) return FD_OK;
  const long tiles = (long)B * ((N + TI - 1) / TI);
  FlashBwdArgs a{proj, A, zb, dfeats, feats, doptg, ptdot, qp, kp, vp, head_w, trans, dL, dzb, dqp, hw_part, B, N};
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_bwd_kernel<8>), dim3((unsigned)tiles), dim3(512), 0, (hipStream_t)stream, a);
  FD_CHECK_LAUNCH("fd_ipa_flash_bwd");
  {
    int rc = fd_colsum_acc(hw_part, H, (long)B * N, H, dhead_w, stream);
    if (rc != FD_OK) return rc;
  }
  if (dkp == nullptr) return FD_OK;        // (the key-side kernel, fd_ipa_flash_bwd_keys, forms dk_pts with dK / dV)
  return fd_ipa_kpts_bwd(dL, qp, kp, head_w, dkp, B, N, stream);
}

extern "C" int fd_ipa_flash_bwd_keys(const float* A, const float* dL, const float* proj, const float* dfeats, const float* doptg,
                                     const float* qp, const float* kp, const float* head_w, float* dproj, float* dvp, float* dkp,
                                     int B, int N, int heads_per_block, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_flash_bwd_keys: N=%d exceeds %d", N, MAXN);
  FD_CHECK_ARG(A && dL && proj && dfeats && doptg && qp && kp && head_w && dproj && dvp && dkp, "fd_ipa_flash_bwd_keys: null operand");
  FD_CHECK_ARG(fd_aligned16(proj) && fd_aligned16(dfeats) && fd_aligned16(doptg) && fd_aligned16(qp) && fd_aligned16(kp) &&
                   fd_aligned16(dproj) && fd_aligned16(dvp) && fd_aligned16(dkp),
               "fd_ipa_flash_bwd_keys: tensor arguments must be 16-byte aligned");
  FD_CHECK_ARG(heads_per_block == 0 || heads_per_block == 1 || heads_per_block == 2 || heads_per_block == 4 || heads_per_block == 8,
               "fd_ipa_flash_bwd_keys: heads_per_block must be 0 (pick), 1 (four key tiles of one head per block), 2, 4 or 8, got %d",
               heads_per_block);
  if (B == 0 || N == 0) return FD_OK;
  const long tiles = (long)B * ((N + TI - 1) / TI);
  if (heads_per_block == 0 || heads_per_block == 1) {
    // four key tiles of one head per block: the query operands go through LDS once per block
    FlashKeysArgs a4{A, dL, proj, dfeats, doptg, qp, kp, head_w, dproj, dvp, dkp, B, N};
    const int njg = ((N + TI - 1) / TI + 3) / 4;
    hipLaunchKernelGGL(ipa_flash_bwd_keys4_kernel, dim3((unsigned)((long)B * H * njg)), dim3(256), 0, (hipStream_t)stream, a4);
    FD_CHECK_LAUNCH("fd_ipa_flash_bwd_keys");
    return FD_OK;
  }
  int hpb = heads_per_block;
  if (hpb == 0) hpb = N >= 256 ? 4 : 2;      // (measured, tools/bench_ipa_keys.py: B=12 x N=200 113 against 127 us, B=7 x N=256 114 against 132)
  FlashKeysArgs a{A, dL, proj, dfeats, doptg, qp, kp, head_w, dproj, dvp, dkp, B, N};
  const dim3 grid((unsigned)(tiles * (H / hpb)));
  if (hpb == 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_bwd_keys_kernel<8>), grid, dim3(512), 0, (hipStream_t)stream, a);
  else if (hpb == 4)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_bwd_keys_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_bwd_keys_kernel<2>), grid, dim3(128), 0, (hipStream_t)stream, a);
  FD_CHECK_LAUNCH("fd_ipa_flash_bwd_keys");
  return FD_OK;
}
