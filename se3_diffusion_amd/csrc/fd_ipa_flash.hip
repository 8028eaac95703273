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
#ifdef FL_DOPIN
#define FL_PIN() fd::sched_pin()      // (probe: the load-ahead order below kept exactly as written -- slower, more spills)
#else
#define FL_PIN()
#endif
// timing-only ablations of tools/bench_ipa_flash.py --variants (wrong results; no product build defines them)
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
};

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
  const int lid = fd_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int g = lid % NG, it = (lid / NG) % nti, b = lid / (NG * nti);
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
    if (inst < STAGE_INSTR) FL_DMA(zb_b + umin(poff[k], plim), slab + inst * 1024);
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
    const float* kr = kbase + (unsigned)(imin(n, N - 1) * LDP + 4 * kk);
#pragma unroll
    for (int cc = 0; cc < KPF; ++cc) kf[cc] = ldkv(kr + 16 * cc);
  }

  // the copy of the next key tile's zb image (its stage was last read in the o_pair phase of tile t - 1)
  auto next_stage = [&](int t) {
    if (t + 1 < nti) {
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
  for (int t = 0; t < nti; ++t) {
    fd::wait_vmem();
    FL_SYNC();                 // stage t of the image has landed; every wave is done with tile t - 1 (its stage, Es, Fs)
#ifndef FL_DMA_MID
    FL_NEXT_STAGE();
#endif
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
#ifdef FL_DMA_MID
        // (probe: the image copy behind the tile's LAST K request instead of at the top of the tile -- measured SLOWER,
        //  122 against 109 us at B=30 x N=128, 464 against 398 at B=8 x N=512: profiles/r04_ipa_flash_variants.log)
        if (cc + KPF == C / 16 - 2 || (KPF >= C / 16 && cc == 0)) FL_NEXT_STAGE();
#endif
        FL_PIN();
        const float4 k0 = kf[cc], k1 = kf[cc + 1];
        s0 = fd::mfma_16x16x4(k0.x, Qf[cc].x, s0);
        s1 = fd::mfma_16x16x4(k1.x, Qf[cc + 1].x, s1);
        s0 = fd::mfma_16x16x4(k0.y, Qf[cc].y, s0);
        s1 = fd::mfma_16x16x4(k1.y, Qf[cc + 1].y, s1);
        s0 = fd::mfma_16x16x4(k0.z, Qf[cc].z, s0);
        s1 = fd::mfma_16x16x4(k1.z, Qf[cc + 1].z, s1);
        s0 = fd::mfma_16x16x4(k0.w, Qf[cc].w, s0);
        s1 = fd::mfma_16x16x4(k1.w, Qf[cc + 1].w, s1);
        FL_PIN();
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
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j0 + 4 * kk + r < N) Arow[j0 + 4 * kk + r] = s[r];
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
      FL_PIN();
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
      FL_PIN();
    }
    // the next key tile's first K fragments (in flight across the barrier and the o_pair phase)
    if (t + 1 < nti) {
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
    for (int t = 0; t < nti; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = TI * t + 4 * kk + r;
        if (j < N) Arow[j] = expf(Arow[j] - m_run) * inv;
      }
  }
}

}  // namespace

extern "C" int fd_ipa_flash_fwd(const float* proj, const float* zb, const float* qp, const float* kp, const float* vp,
                                const float* head_w, const float* mask, const float* quat, const float* trans,
                                float* feats, float* A, int B, int N, int heads_per_block, void* stream) {
  FD_CHECK_ARG(N <= MAXN, "fd_ipa_flash_fwd: N=%d exceeds %d", N, MAXN);
  FD_CHECK_ARG(feats != nullptr, "fd_ipa_flash_fwd: feats is required");
  FD_CHECK_ARG(fd_aligned16(proj) && fd_aligned16(zb) && fd_aligned16(qp) && fd_aligned16(kp) && fd_aligned16(vp) &&
                   fd_aligned16(feats),
               "fd_ipa_flash_fwd: proj, zb, qp, kp, vp and feats must be 16-byte aligned");
  FD_CHECK_ARG(heads_per_block == 0 || heads_per_block == 2 || heads_per_block == 4 || heads_per_block == 8,
               "fd_ipa_flash_fwd: heads_per_block must be 0 (pick), 2, 4 or 8, got %d", heads_per_block);
  if (B == 0 || N == 0) return FD_OK;
  const long tiles = (long)B * ((N + TI - 1) / TI);
  int hpb = heads_per_block;
  if (hpb == 0) hpb = tiles >= 128 ? 8 : tiles >= 32 ? 4 : 2;      // (a lone N = 128 backbone: 8 query tiles -> 32 blocks)
  FlashArgs a{proj, zb, qp, kp, vp, head_w, mask, quat, trans, feats, A, B, N};
  const dim3 grid((unsigned)(tiles * (H / hpb)));
  if (hpb == 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_fwd_kernel<8>), grid, dim3(512), 0, (hipStream_t)stream, a);
  else if (hpb == 4)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_fwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ipa_flash_fwd_kernel<2>), grid, dim3(128), 0, (hipStream_t)stream, a);
  FD_CHECK_LAUNCH("fd_ipa_flash_fwd");
  return FD_OK;
}
