// Edge embedder (model/score_network.py:97-101,129-153 + data/utils.py:570-580) as ONE kernel: the 120-d pair feature is
// built on the fly in the registers of the wave that owns the pair row, the MLP 120 -> 128 -> 128 -> 128, its LayerNorm
// and the pair mask follow in the same registers (fd_chain.h: register-chained split-bf16 layers, weights streamed by
// LDS-DMA).  Nothing but z [P,128] reaches HBM (training also saves h1, h2, h3 for the backward).
//
// The feature of pair (b, i, j) is [t-emb(32) | fixed_i | t-emb(32) | fixed_j | sincos((idx_i - idx_j) pi / d_k)(32) |
// distogram(22)].  Its first 66 entries depend on ONE residue each, so their share of the first layer is node-level:
//     W0 x = (W0[:, 0:33] pt_i + b0) + W0[:, 33:66] pt_j + W0[:, 66:120] [relpos | distogram]
// = P[b,i] + Q[b,j] (two [B N, 128] GEMMs on the host side) + a K = 54 product per pair (padded to two 32-k steps), whose
// operand -- 8 sin/cos values and 8 distogram bits per lane -- is generated in MFMA B-operand layout directly.  The
// sin / cos ARGUMENTS are the reference's own op sequence (fd_feats.hip).
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

#include "fd_chain.h"

constexpr int EE_UNITS = 20;               // layer 1: 2 k-steps x 2 n-groups; layers 2, 3: 4 x 2 each
constexpr int EE_ZB_UNITS = 4;             // + the first IPA block's [linear_b ; down_z] (40 <- 128: 4 k-steps x one n-group)
constexpr int EE_ZB = 40;
constexpr int EE_C = 128;
constexpr float kPi = 3.14159265358979323846f;

// weight image, units in consumption order (fd_chain.h): layer 1 = W0[:, 66:120] (natural k order, zero beyond k = 54),
// layers 2 / 3 = W2 / W4 (chained k order); unit = (k-step, n-group) with the n-group minor.
__global__ __launch_bounds__(256) void edge_embed_pack_kernel(const float* __restrict__ W0, const float* __restrict__ W2,
                                                              const float* __restrict__ W4, char* __restrict__ img) {
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (unit, n-block, lane)
  if (gid >= EE_UNITS * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, u = gid >> 8;
  const int m = lane & 15, g = lane >> 4;
  const int layer = u < 4 ? 0 : (u < 12 ? 1 : 2);
  const int r = u - (layer == 0 ? 0 : (layer == 1 ? 4 : 12));
  const int n = 64 * (r & 1) + 16 * i + m, k0 = 32 * (r >> 1);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (layer == 0) {
      const int k = k0 + 8 * g + e;
      x[e] = k < 54 ? W0[n * 120 + 66 + k] : 0.f;
    } else {
      const int k = k0 + 16 * (e >> 2) + 4 * g + (e & 3);
      x[e] = (layer == 1 ? W2 : W4)[n * EE_C + k];
    }
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)u * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

// units 20..23 of the image: W40 = [linear_b.weight ; down_z.weight] [40,128] of the first trunk block's IPA
// (ipa_pytorch.py:380-386,455), rows 40..63 zero, chained k order
__global__ __launch_bounds__(256) void edge_embed_pack_zb_kernel(const float* __restrict__ W40, char* __restrict__ img) {
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (k-step, n-block, lane)
  if (gid >= EE_ZB_UNITS * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, ks = gid >> 8;
  const int m = lane & 15, g = lane >> 4;
  const int n = 16 * i + m;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * ks + 16 * (e >> 2) + 4 * g + (e & 3);
    x[e] = n < EE_ZB ? W40[n * EE_C + k] : 0.f;
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)(EE_UNITS + ks) * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

// ZB: a fourth chained layer on the kernel's own output -- zb = [linear_b ; down_z] z + b40 of the first trunk block's IPA
template <bool ZB>
__global__ __launch_bounds__(64 * EM_WAVES, 2) void edge_embed_kernel(FdEdgeEmbedDesc d) {
  constexpr int EE_NSTAGE = (EE_UNITS + (ZB ? EE_ZB_UNITS : 0)) / EM_UPS;
  __shared__ __attribute__((aligned(16))) char lds[2 * EM_STAGE];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const long rows = d.rows;
  const long ld_pq = d.ld_pq > 0 ? d.ld_pq : EE_C;      // row stride of p / q (a caller that forms both with one GEMM passes its width)
  const int ntiles = (int)((rows + EM_ROWS - 1) / EM_ROWS);
  const int G = (int)gridDim.x, first = (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nmine = (ntiles - first + G - 1) / G;
  const int total_stages = nmine * EE_NSTAGE;

  // ---- weight stream (as fd_edge_mlp.hip): every wave copies an eighth of each stage; stage s lives in buffer s & 1 ----
  const char* __restrict__ img_lane = static_cast<const char*>(d.img) + wave * (EM_STAGE / EM_WAVES) + lane * 16;
  char* const lds_wave = lds + wave * (EM_STAGE / EM_WAVES);
  int issued = 0, consumed = 0;
  auto issue_stage = [&]() {
    const char* src = img_lane + (long)(issued % EE_NSTAGE) * EM_STAGE;
    char* dst = lds_wave + (issued & 1) * EM_STAGE;
    fd::glds16x4(src, dst);
    fd::glds16x2(src + 4096, dst + 4096);
    ++issued;
  };
  auto stage_begin = [&]() -> const char* {
    fd::wait_vmem();
    __syncthreads();
    const char* cur = lds + (consumed & 1) * EM_STAGE + lane * 16;
    ++consumed;
    return cur;
  };
  auto stage_prefetch = [&]() {
    if (issued < total_stages) issue_stage();
  };
  issue_stage();

  // lane-constant tables: the 8 index-embedding denominators and the 8 distogram bins of this lane's k slots
  float den[8], lo[8], up[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    den[e] = d.idenom[(8 * g + e) & 15];
    const int kb = 8 * g + e;
    lo[e] = kb < 22 ? d.dg_lower[kb] : 3.0e38f;      // an empty bin: d > 3e38 never holds
    up[e] = kb < 22 ? d.dg_upper[kb] : 0.f;
  }

  for (int ti = 0; ti < nmine; ++ti) {
    const long row = ((long)first + (long)ti * G) * EM_ROWS + wave * 16 + m;
    const bool rok = row < rows;
    const long rc = rok ? row : rows - 1;         // rows past the end are clamped on load, masked on store
    const long qi = rc / d.nres;                  // (b, i)
    const long qj = (qi / d.nres) * d.nres + (rc - qi * d.nres);   // (b, j)

    // ---- the pair part of the feature, in B-operand layout (k = 8 g + e of each 32-k step) ----
    float x0[8], x1[8];
    {
      const float rel = (float)(d.seq_idx[qi] - d.seq_idx[qj]) * kPi;
      const float dx = d.sc_ca[qi * 3 + 0] - d.sc_ca[qj * 3 + 0], dy = d.sc_ca[qi * 3 + 1] - d.sc_ca[qj * 3 + 1],
                  dz = d.sc_ca[qi * 3 + 2] - d.sc_ca[qj * 3 + 2];
      const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // slots 0..15 sin, 16..31 cos of the same 16 arguments: lanes g and g + 2 evaluate the same sincos and keep
        // one half each (a divergent sinf / cosf pair would cost the wave both calls anyway)
        float sn, cs;
        sincosf(rel / den[e], &sn, &cs);
        x0[e] = g < 2 ? sn : cs;
        x1[e] = (dist > lo[e] && dist < up[e]) ? 1.f : 0.f;
      }
    }
    uint4 b[3];
    Em16Half H[2];

    // ---- layer 1: h1 = relu(P_i + Q_j + W0[:, 66:120] x): the node terms are the accumulator's initial value ----
    f32x4 acc1[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const float4 pa = *reinterpret_cast<const float4*>(d.p + qi * ld_pq + 16 * nb + 4 * g);
      const float4 qa = *reinterpret_cast<const float4*>(d.q + qj * ld_pq + 16 * nb + 4 * g);
      acc1[nb][0] = pa.x + qa.x; acc1[nb][1] = pa.y + qa.y; acc1[nb][2] = pa.z + qa.z; acc1[nb][3] = pa.w + qa.w;
    }
#pragma clang loop unroll(full)
    for (int sg = 0; sg < 4 / EM_UPS; ++sg) {
      const char* st = stage_begin();
      em16_read_half(H[0], st);
#pragma clang loop unroll(full)
      for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
        const int r = EM_UPS * sg + (hh >> 1), g2 = r & 1, a = 4 * g2 + 2 * (hh & 1);
        EM_PIN_TOP();
        if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
        EM_PIN_MID();
        if (g2 == 0 && (hh & 1) == 0) {
          if ((r >> 1) == 0) em_split8(x0, b[0], b[1], b[2]); else em_split8(x1, b[0], b[1], b[2]);
        }
        em16_mma_half(acc1[a], acc1[a + 1], H[hh & 1], b);
        EM_GROUPS(hh + 1 < 2 * EM_UPS);
        if (hh == 1) stage_prefetch();
      }
    }
    unsigned bits = 0u;       // packed signs of the layer's 32 units of this lane: bit 4 nb + e <-> unit 16 nb + 4 g + e
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bits |= (acc1[nb][e] > 0.f ? 1u : 0u) << (4 * nb + e);
        acc1[nb][e] = acc1[nb][e] > 0.f ? acc1[nb][e] : 0.f;
      }
      if (d.h1 != nullptr && rok)
        *reinterpret_cast<float4*>(d.h1 + row * EE_C + 16 * nb + 4 * g) =
            make_float4(acc1[nb][0], acc1[nb][1], acc1[nb][2], acc1[nb][3]);
    }
    if (d.mask1 != nullptr && rok) d.mask1[row * 4 + g] = bits;

    // ---- layers 2 and 3: K = 128 of the previous layer's registers, bias as the initial value ----
    f32x4 acc2[8], acc3[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const float4 b2 = *reinterpret_cast<const float4*>(d.bias2 + 16 * nb + 4 * g);
      const float4 b3 = *reinterpret_cast<const float4*>(d.bias3 + 16 * nb + 4 * g);
      acc2[nb][0] = b2.x; acc2[nb][1] = b2.y; acc2[nb][2] = b2.z; acc2[nb][3] = b2.w;
      acc3[nb][0] = b3.x; acc3[nb][1] = b3.y; acc3[nb][2] = b3.z; acc3[nb][3] = b3.w;
    }
#pragma clang loop unroll(full)
    for (int sg = 0; sg < 8 / EM_UPS; ++sg) {
      const char* st = stage_begin();
      em16_read_half(H[0], st);
#pragma clang loop unroll(full)
      for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
        const int r = EM_UPS * sg + (hh >> 1), ks = r >> 1, g2 = r & 1, a = 4 * g2 + 2 * (hh & 1);
        EM_PIN_TOP();
        if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
        EM_PIN_MID();
        if (g2 == 0 && (hh & 1) == 0) em16_split2(acc1[2 * ks], acc1[2 * ks + 1], b[0], b[1], b[2]);
        em16_mma_half(acc2[a], acc2[a + 1], H[hh & 1], b);
        EM_GROUPS(hh + 1 < 2 * EM_UPS);
        if (hh == 1) stage_prefetch();
      }
    }
    bits = 0u;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bits |= (acc2[nb][e] > 0.f ? 1u : 0u) << (4 * nb + e);
        acc2[nb][e] = acc2[nb][e] > 0.f ? acc2[nb][e] : 0.f;
      }
      if (d.h2 != nullptr && rok)
        *reinterpret_cast<float4*>(d.h2 + row * EE_C + 16 * nb + 4 * g) =
            make_float4(acc2[nb][0], acc2[nb][1], acc2[nb][2], acc2[nb][3]);
    }
    if (d.mask2 != nullptr && rok) d.mask2[row * 4 + g] = bits;
#pragma clang loop unroll(full)
    for (int sg = 0; sg < 8 / EM_UPS; ++sg) {
      const char* st = stage_begin();
      em16_read_half(H[0], st);
#pragma clang loop unroll(full)
      for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
        const int r = EM_UPS * sg + (hh >> 1), ks = r >> 1, g2 = r & 1, a = 4 * g2 + 2 * (hh & 1);
        EM_PIN_TOP();
        if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
        EM_PIN_MID();
        if (g2 == 0 && (hh & 1) == 0) em16_split2(acc2[2 * ks], acc2[2 * ks + 1], b[0], b[1], b[2]);
        em16_mma_half(acc3[a], acc3[a + 1], H[hh & 1], b);
        EM_GROUPS(hh + 1 < 2 * EM_UPS);
        if (hh == 1) stage_prefetch();
      }
    }

    // ---- z = rowscale * LayerNorm(h3).  A row's 128 values sit in four lanes (l & 15 fixed). ----
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      s += (acc3[nb][0] + acc3[nb][1]) + (acc3[nb][2] + acc3[nb][3]);
      if (d.h3 != nullptr && rok)
        *reinterpret_cast<float4*>(d.h3 + row * EE_C + 16 * nb + 4 * g) =
            make_float4(acc3[nb][0], acc3[nb][1], acc3[nb][2], acc3[nb][3]);
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
    if (rok && g == 0) {
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
      if (rok) *reinterpret_cast<float4*>(d.out + row * EE_C + col) = o;
      if (ZB) { acc3[nb][0] = o.x; acc3[nb][1] = o.y; acc3[nb][2] = o.z; acc3[nb][3] = o.w; }
    }
    if (ZB) {
      // ---- layer 4: zb[0:40] = W40 z + b40 (n-blocks 0..2 of one n-group; columns >= 40 are zero weights) ----
      f32x4 acc4[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int col = 16 * nb + 4 * g;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < EE_ZB && d.zb_bias != nullptr) a = *reinterpret_cast<const float4*>(d.zb_bias + col);
        acc4[nb][0] = a.x; acc4[nb][1] = a.y; acc4[nb][2] = a.z; acc4[nb][3] = a.w;
      }
#pragma clang loop unroll(full)
      for (int sg = 0; sg < EE_ZB_UNITS / EM_UPS; ++sg) {
        const char* st = stage_begin();
        em16_read_half(H[0], st);
#pragma clang loop unroll(full)
        for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
          const int ks = EM_UPS * sg + (hh >> 1), a = 2 * (hh & 1);
          EM_PIN_TOP();
          if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
          EM_PIN_MID();
          if ((hh & 1) == 0) em16_split2(acc3[2 * ks], acc3[2 * ks + 1], b[0], b[1], b[2]);
          em16_mma_half(acc4[a], acc4[a + 1], H[hh & 1], b);
          EM_GROUPS(hh + 1 < 2 * EM_UPS);
          if (hh == 1) stage_prefetch();
        }
      }
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) {
        const int col = 16 * nb + 4 * g;
        if (rok && col < EE_ZB)
          *reinterpret_cast<float4*>(d.zb_out + row * EE_ZB + col) = make_float4(acc4[nb][0], acc4[nb][1], acc4[nb][2], acc4[nb][3]);
      }
    }
  }
}

}  // namespace

extern "C" int fd_edge_embed_pack(const float* W0, const float* W2, const float* W4, void* img, void* stream) {
  FD_CHECK_ARG(W0 && W2 && W4 && img, "fd_edge_embed_pack: null operand");
  FD_CHECK_ARG(fd_aligned16(img), "fd_edge_embed_pack: image must be 16-byte aligned");
  hipLaunchKernelGGL(edge_embed_pack_kernel, dim3(EE_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, W0, W2, W4,
                     static_cast<char*>(img));
  FD_CHECK_LAUNCH("fd_edge_embed_pack");
  return FD_OK;
}

extern "C" int fd_edge_embed_pack_zb(const float* W40, void* img, void* stream) {
  FD_CHECK_ARG(W40 && img, "fd_edge_embed_pack_zb: null operand");
  FD_CHECK_ARG(fd_aligned16(img), "fd_edge_embed_pack_zb: image must be 16-byte aligned");
  hipLaunchKernelGGL(edge_embed_pack_zb_kernel, dim3(EE_ZB_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, W40,
                     static_cast<char*>(img));
  FD_CHECK_LAUNCH("fd_edge_embed_pack_zb");
  return FD_OK;
}

extern "C" int fd_edge_embed(const FdEdgeEmbedDesc* desc, void* stream) {
  FD_CHECK_ARG(desc != nullptr, "fd_edge_embed: null descriptor");
  const FdEdgeEmbedDesc& d = *desc;
  FD_CHECK_ARG(d.seq_idx && d.sc_ca && d.idenom && d.dg_lower && d.dg_upper && d.img && d.p && d.q && d.bias2 && d.bias3 &&
                   d.gamma && d.beta && d.out,
               "fd_edge_embed: a required operand is null");
  FD_CHECK_ARG(d.nres > 0 && d.rows >= 0, "fd_edge_embed: bad extents");
  const void* ptrs[] = {d.img, d.p, d.q, d.bias2, d.bias3, d.gamma, d.beta, d.h1, d.h2, d.h3, d.out};
  for (const void* p : ptrs) FD_CHECK_ARG(fd_aligned16(p), "fd_edge_embed: operands must be 16-byte aligned");
  FD_CHECK_ARG(d.ld_pq == 0 || (d.ld_pq >= 128 && (d.ld_pq & 3) == 0), "fd_edge_embed: ld_pq must be 0 or a multiple of 4 >= 128");
  if (d.rows == 0) return FD_OK;
  const long ntiles = (d.rows + EM_ROWS - 1) / EM_ROWS;
  const int blocks = d.blocks > 0 ? d.blocks : 256 * EM_BLOCKS_PER_CU;   // MI355X: persistent blocks fill the 256 CUs
  FD_CHECK_ARG(d.zb_out == nullptr || (fd_aligned16(d.zb_out) && fd_aligned16(d.zb_bias)),
               "fd_edge_embed: zb_out / zb_bias must be 16-byte aligned (and the image carry the fd_edge_embed_pack_zb units)");
  if (d.zb_out != nullptr)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(edge_embed_kernel<true>), dim3((unsigned)(ntiles < blocks ? ntiles : blocks)), dim3(64 * EM_WAVES),
                       0, (hipStream_t)stream, d);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(edge_embed_kernel<false>), dim3((unsigned)(ntiles < blocks ? ntiles : blocks)), dim3(64 * EM_WAVES),
                       0, (hipStream_t)stream, d);
  FD_CHECK_LAUNCH("fd_edge_embed");
  return FD_OK;
}
