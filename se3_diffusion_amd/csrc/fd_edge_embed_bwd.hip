// Backward dX chain of the edge embedder (autograd of model/score_network.py:67-86,194-195 w.r.t. the hidden activations)
// as ONE kernel on MI355X (gfx950), the 128-wide sibling of edge_mlp16_kernel<true>:
//
//     g   = rowscale * dz                                   (z = rowscale * LayerNorm(h3))
//     dh3 = rstd (g gamma - mean_c(g gamma) - xhat mean_c(g gamma xhat)),   dgamma += sum_rows g xhat,  dbeta += sum_rows g
//     dh2 = [h2 > 0] (dh3 W4)
//     dh1 = [h1 > 0] (dh2 W2)
//
// per pair row, in the registers of the wave that owns the row (fd_chain.h: register-chained split-bf16 layers on
// v_mfma_f32_16x16x32_bf16, the transposed weights streamed by LDS-DMA as pre-split bf16 planes).  It replaces
// fd_layernorm_bwd + two gated dX GEMMs (and their HBM round trips of dh3 / dh2) at the tail of the training step; dh3,
// dh2, dh1 are written once, for the weight-gradient launch (fd_group_dw over the pair rows).  The LayerNorm parameter
// gradients are kept in registers across a persistent block's tiles (64 per lane) and flushed once per block.
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

#include "fd_chain.h"

constexpr int EB_UNITS = 16;               // 2 layers x 4 k-steps x 2 n-groups
constexpr int EB_NSTAGE = EB_UNITS / EM_UPS;
constexpr int EB_C = 128;

// weight image, units in consumption order: layer A = W4^T (dh2 = dh3 W4: unit n = input index of Linear 4, k = its output
// index), layer B = W2^T; both in chained k order (their operands are the previous stage's registers).
__global__ __launch_bounds__(256) void edge_embed_bwd_pack_kernel(const float* __restrict__ W2, const float* __restrict__ W4,
                                                                  char* __restrict__ img) {
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (unit, n-block, lane)
  if (gid >= EB_UNITS * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, u = gid >> 8;
  const int m = lane & 15, g = lane >> 4;
  const int layer = u >> 3, r = u & 7;
  const int n = 64 * (r & 1) + 16 * i + m, k0 = 32 * (r >> 1);
  const float* __restrict__ W = layer == 0 ? W4 : W2;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + 16 * (e >> 2) + 4 * g + (e & 3);
    x[e] = W[k * EB_C + n];
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)u * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

__global__ __launch_bounds__(64 * EM_WAVES, 2) void edge_embed_bwd_kernel(FdEdgeEmbedBwdDesc d) {
  __shared__ __attribute__((aligned(16))) char lds[2 * EM_STAGE];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const long rows = d.rows;
  const int ntiles = (int)((rows + EM_ROWS - 1) / EM_ROWS);
  const int G = (int)gridDim.x, first = (int)blockIdx.x;
  if (first >= ntiles) return;
  // tiles: block b starts with tile b, then the static stride or (d.sched, more tiles than blocks) the next tile nobody has taken --
  // as in fd_edge_mlp.hip: a block that becomes resident late no longer finishes an equal share late
  unsigned* const sched = d.sched;      // (zeroed by the host entry in front of the launch)
  const bool dyn = sched != nullptr;
  __shared__ int s_tile;
  const int nmine = (ntiles - first + G - 1) / G;
  const int total_stages = dyn ? 0x7fffffff : nmine * EB_NSTAGE;

  // ---- weight stream (as fd_edge_embed.hip) ----
  const char* __restrict__ img_lane = static_cast<const char*>(d.img) + wave * (EM_STAGE / EM_WAVES) + lane * 16;
  char* const lds_wave = lds + wave * (EM_STAGE / EM_WAVES);
  int issued = 0, consumed = 0;
  auto issue_stage = [&]() {
    const char* src = img_lane + (long)(issued % EB_NSTAGE) * EM_STAGE;
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

  // LayerNorm parameter gradients of this lane's 32 columns (16 nb + 4 g + r), summed over every row the lane sees
  f32x4 dgam[8], dbet[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) dgam[nb][e] = dbet[nb][e] = 0.f;

  int tile = first, nxt = 0;
  for (int ti = 0;;) {
    if (dyn && tid == 0) nxt = G + (int)atomicAdd(&sched[0], 1u);     // (the tile after this one)
    const long row = (long)tile * EM_ROWS + wave * 16 + m;
    const bool rok = row < rows;
    const long rc = rok ? row : rows - 1;         // rows past the end are clamped on load, masked on store / in the sums

    // packed ReLU gates (the forward's mask outputs: bit 4 nb + e of word (row, g) <-> unit 16 nb + 4 g + e), fetched up front
    const bool packed = d.gmask2 != nullptr;
    unsigned gm2 = 0u, gm1 = 0u;
    if (packed) {
      gm2 = d.gmask2[rc * 4 + g];
      gm1 = d.gmask1[rc * 4 + g];
    }

    // ---- LayerNorm backward: dh3 in the register layout of a layer output (lane (m, g): columns 16 nb + 4 g + r) ----
    f32x4 d3[8];
    {
      const float rs = (d.rowscale != nullptr ? d.rowscale[rc] : 1.f) * (rok ? 1.f : 0.f);
      const float mean = d.mean[rc], rstd = d.rstd[rc];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const int col = 16 * nb + 4 * g;
        const float4 dy = *reinterpret_cast<const float4*>(d.dy + rc * EB_C + col);
        const float4 h = *reinterpret_cast<const float4*>(d.h3 + rc * EB_C + col);
        const float4 gm = *reinterpret_cast<const float4*>(d.gamma + col);
        const float dyv[4] = {dy.x, dy.y, dy.z, dy.w}, hv[4] = {h.x, h.y, h.z, h.w}, gv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = (hv[e] - mean) * rstd;
          const float gy = dyv[e] * rs;
          dgam[nb][e] += gy * x;
          dbet[nb][e] += gy;
          const float t = gy * gv[e];
          d3[nb][e] = t;
          s1 += t;
          s2 += t * x;
        }
      }
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16);
      s2 += __shfl_xor(s2, 32);
      const float m1 = s1 * (1.0f / 128.0f), m2 = s2 * (1.0f / 128.0f);
      // (xhat is recomputed from a second read of h3 -- an L1 / L2 hit -- instead of 32 more live registers)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const float4 h = *reinterpret_cast<const float4*>(d.h3 + rc * EB_C + 16 * nb + 4 * g);
        const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) d3[nb][e] = rstd * (d3[nb][e] - m1 - (hv[e] - mean) * rstd * m2);
        if (rok)
          *reinterpret_cast<float4*>(d.dh3 + row * EB_C + 16 * nb + 4 * g) =
              make_float4(d3[nb][0], d3[nb][1], d3[nb][2], d3[nb][3]);
      }
    }
    uint4 b[3];
    Em16Half H[2];

    // ---- dh2 = [h2 > 0] (dh3 W4) ----
    f32x4 a2[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) a2[nb][e] = 0.f;
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
        if (g2 == 0 && (hh & 1) == 0) em16_split2(d3[2 * ks], d3[2 * ks + 1], b[0], b[1], b[2]);
        em16_mma_half(a2[a], a2[a + 1], H[hh & 1], b);
        EM_GROUPS(hh + 1 < 2 * EM_UPS);
        if (hh == 1) stage_prefetch();
      }
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      if (packed) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a2[nb][e] = ((gm2 >> (4 * nb + e)) & 1u) ? a2[nb][e] : 0.f;
      } else {
        const float4 h = *reinterpret_cast<const float4*>(d.h2 + rc * EB_C + 16 * nb + 4 * g);
        a2[nb][0] = h.x > 0.f ? a2[nb][0] : 0.f;
        a2[nb][1] = h.y > 0.f ? a2[nb][1] : 0.f;
        a2[nb][2] = h.z > 0.f ? a2[nb][2] : 0.f;
        a2[nb][3] = h.w > 0.f ? a2[nb][3] : 0.f;
      }
      if (rok)
        *reinterpret_cast<float4*>(d.dh2 + row * EB_C + 16 * nb + 4 * g) = make_float4(a2[nb][0], a2[nb][1], a2[nb][2], a2[nb][3]);
    }

    // ---- dh1 = [h1 > 0] (dh2 W2) ----
    f32x4 a1[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) a1[nb][e] = 0.f;
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
        if (g2 == 0 && (hh & 1) == 0) em16_split2(a2[2 * ks], a2[2 * ks + 1], b[0], b[1], b[2]);
        em16_mma_half(a1[a], a1[a + 1], H[hh & 1], b);
        EM_GROUPS(hh + 1 < 2 * EM_UPS);
        if (hh == 1) stage_prefetch();
      }
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      float4 o;
      if (packed) {
        o = make_float4(((gm1 >> (4 * nb)) & 1u) ? a1[nb][0] : 0.f, ((gm1 >> (4 * nb + 1)) & 1u) ? a1[nb][1] : 0.f,
                        ((gm1 >> (4 * nb + 2)) & 1u) ? a1[nb][2] : 0.f, ((gm1 >> (4 * nb + 3)) & 1u) ? a1[nb][3] : 0.f);
      } else {
        const float4 h = *reinterpret_cast<const float4*>(d.h1 + rc * EB_C + 16 * nb + 4 * g);
        o = make_float4(h.x > 0.f ? a1[nb][0] : 0.f, h.y > 0.f ? a1[nb][1] : 0.f, h.z > 0.f ? a1[nb][2] : 0.f,
                        h.w > 0.f ? a1[nb][3] : 0.f);
      }
      if (rok) *reinterpret_cast<float4*>(d.dh1 + row * EB_C + 16 * nb + 4 * g) = o;
    }
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
  if (dyn) fd::wait_vmem();     // the stage copied ahead for a tile that does not exist lands before the ring is reused below

  // ---- LayerNorm parameter gradients: over the 16 rows of the wave (lanes l & 15), over the waves (LDS), one atomic per
  // column and block ----
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = dgam[nb][e], c = dbet[nb][e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o);
        c += __shfl_xor(c, o);
      }
      dgam[nb][e] = a;
      dbet[nb][e] = c;
    }
  __syncthreads();                                   // every wave is done with the weight ring
  float* red = reinterpret_cast<float*>(lds);        // [EM_WAVES][2][128]
  if (m == 0) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(wave * 2 + 0) * EB_C + 16 * nb + 4 * g + e] = dgam[nb][e];
        red[(wave * 2 + 1) * EB_C + 16 * nb + 4 * g + e] = dbet[nb][e];
      }
  }
  __syncthreads();
  if (tid < 2 * EB_C) {
    const int which = tid >> 7, col = tid & 127;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < EM_WAVES; ++w) a += red[(w * 2 + which) * EB_C + col];
    float* dst = which == 0 ? d.dgamma : d.dbeta;
    if (dst != nullptr) atomicAdd(dst + col, a);
  }
}

}  // namespace

extern "C" int fd_edge_embed_bwd_pack(const float* W2, const float* W4, void* img, void* stream) {
  FD_CHECK_ARG(W2 && W4 && img, "fd_edge_embed_bwd_pack: null operand");
  FD_CHECK_ARG(fd_aligned16(img), "fd_edge_embed_bwd_pack: image must be 16-byte aligned");
  hipLaunchKernelGGL(edge_embed_bwd_pack_kernel, dim3(EB_UNITS * 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, W2, W4,
                     static_cast<char*>(img));
  FD_CHECK_LAUNCH("fd_edge_embed_bwd_pack");
  return FD_OK;
}

extern "C" int fd_edge_embed_bwd(const FdEdgeEmbedBwdDesc* desc, void* stream) {
  FD_CHECK_ARG(desc != nullptr, "fd_edge_embed_bwd: null descriptor");
  const FdEdgeEmbedBwdDesc& d = *desc;
  FD_CHECK_ARG(d.dy && d.h3 && d.mean && d.rstd && d.gamma && d.img && d.dh3 && d.dh2 && d.dh1,
               "fd_edge_embed_bwd: a required operand is null");
  FD_CHECK_ARG((d.h2 && d.h1) || (d.gmask2 && d.gmask1), "fd_edge_embed_bwd: h2 / h1 or their packed sign masks are required");
  FD_CHECK_ARG(d.rows >= 0, "fd_edge_embed_bwd: negative row count");
  const void* ptrs[] = {d.dy, d.h3, d.gamma, d.h2, d.h1, d.img, d.dh3, d.dh2, d.dh1};
  for (const void* p : ptrs) FD_CHECK_ARG(fd_aligned16(p), "fd_edge_embed_bwd: operands must be 16-byte aligned");
  if (d.rows == 0) return FD_OK;
  const long ntiles = (d.rows + EM_ROWS - 1) / EM_ROWS;
  const int blocks = d.blocks > 0 ? d.blocks : 256 * EM_BLOCKS_PER_CU;   // MI355X: persistent blocks fill the 256 CUs
  const long grid = ntiles < blocks ? ntiles : blocks;
  FdEdgeEmbedBwdDesc dd = d;
  if (dd.sched != nullptr && ntiles >= 4 * grid) {       // (as fd_edge_mlp: zeroed on the launch's own stream)
    FD_CHECK_ARG(hipMemsetAsync(dd.sched, 0, sizeof(unsigned), (hipStream_t)stream) == hipSuccess,
                 "fd_edge_embed_bwd: zeroing the tile counter failed");
  } else {
    dd.sched = nullptr;
  }
  hipLaunchKernelGGL(edge_embed_bwd_kernel, dim3((unsigned)grid), dim3(64 * EM_WAVES), 0, (hipStream_t)stream, dd);
  FD_CHECK_LAUNCH("fd_edge_embed_bwd");
  return FD_OK;
}
