// The fused edge transition (csrc/fd_edge_mlp.hip; EdgeTransition.forward, model/ipa_pytorch.py:218-233) for launches that give the
// chip less than one 16-row group per SIMD -- a lone backbone in sampling (N <= 128, B = 1: 16,384 pair rows = 1,024 groups on 1,024
// SIMDs).  There the register-chained kernel is ONE wave per SIMD walking a 2,952-MFMA chain in order: nothing overlaps its LDS
// reads, stage barriers and epilogues, and the matrix pipe runs at a quarter of its rate (73 us per launch, three launches per
// diffusion step).  This kernel gives every SIMD a second instruction stream by letting TWO waves own a 16-row group:
//
//   block = 8 waves = 4 row groups x 2 halves (wave = 2 rg + p), 64-row tiles, one block per CU, the same weight image and the same
//   two-stage LDS ring of 48 KB stages (four 12 KB units) as the 8-wave shape; wave p multiplies half-unit p (n-blocks 2p, 2p + 1)
//   of every unit it takes part in, i.e. half the MFMAs and half the fragment reads of the chain:
//     layer 1   columns split: wave p owns the 16-blocks {2p, 2p+1, 4+2p, 5+2p} of each 128-unit chunk = the k-steps {p, 2+p} of
//               layer 2 in that chunk.  Both waves split the input row themselves (48 resident registers instead of an exchange).
//     layer 2   columns split (12 of the 24 16-blocks: a 48-register accumulator).  A k-step's operand is the owner's two blocks:
//               every wave leaves its four blocks of h1 (fp32, 4 KB) in an LDS exchange slot after epilogue 1 -- the stage barrier
//               in front of layer 2 orders the exchange -- and a k-step's operand is read from its owner's slot and split there.
//     layer 3   K split: the blocks of h2 a wave owns are the k-steps {p, 2+p, 4+p, ...} of layer 3, so it multiplies the units of ITS
//               k-steps for all 128 outputs and the two partial sums meet through the exchange slot (the half the partner finishes).
//     LayerNorm each wave finishes 64 columns of its rows: (mean, centred sum of squares) of the halves are combined exactly
//               (Chan's pairwise update, n = 64 + 64).
//     zb layer  K split again (a wave's 64 output columns are two of its four k-steps); the partner's partial sums of the 40 outputs
//               are added by wave 0 of the pair.
//   Three block barriers per tile beyond the stage barriers (partial sums, LayerNorm statistics, zb partial sums).
// Inference forward only (no saves / masks: fd_edge_mlp() picks it for forward launches without training outputs of at most
// FD_EDGE_MLP_PAIR_MAX_ROWS rows).  Per pair row the arithmetic of layers 1-2 is the 16-row kernel's; layer 3, the LayerNorm
// statistics and zb add their halves in a different order, so outputs agree to fp32 rounding, not bit for bit.
#define EM_WAVES 8
#define EM_UPS 4
#include "fd_common.h"
#include "../../include/fd_hip.h"

int fd_edge_mlp_launch_pair(const FdEdgeMlpDesc& d, hipStream_t st);

namespace {

#include "fd_chain.h"

// (the image layout of fd_edge_mlp.hip: 120 units + 4 of the zb layer, 384 hidden units, 128 channels)
constexpr int EM_UNITS = 120, EM_ZB_UNITS = 4, EM_ZB = 40, EM_RING = 2, EM_H = 384, EM_C = 128;
constexpr int XSLOT = 4096;             // exchange slot of a wave: four 16-blocks of its 16 rows, fp32
static_assert(EM_ROWS == 128 && EM_STAGE == 4 * EM_UNIT, "fd_chain.h: 8 waves, four units per stage");
constexpr int PAIR_ROWS = 64;           // rows per tile: 4 row groups (fd_chain.h's EM_ROWS counts 16 rows per WAVE)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ f32x4 lds_f4(const char* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  f32x4 r;
  r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  return r;
}
__device__ __forceinline__ void st_lds4(char* p, const f32x4& v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

template <bool ZB>
__global__ __launch_bounds__(512, 1) void edge_mlp_pair_kernel(FdEdgeMlpDesc d) {
  constexpr int NSTAGE = (EM_UNITS + (ZB ? EM_ZB_UNITS : 0)) / EM_UPS;
  __shared__ __attribute__((aligned(16))) char lds[EM_RING * EM_STAGE];
  __shared__ __attribute__((aligned(16))) char xch[8 * XSLOT];
  __shared__ __attribute__((aligned(16))) float2 xst[8 * 64];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int rg = wave >> 1, p = wave & 1;
  const long rows = d.rows;
  const long ld_pq = d.ld_pq > 0 ? d.ld_pq : EM_H, ld_pqf = d.ld_pqf > 0 ? d.ld_pqf : EM_C;
  const int ntiles = (int)((rows + PAIR_ROWS - 1) / PAIR_ROWS);
  const int G = (int)gridDim.x, first = (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nmine = (ntiles - first + G - 1) / G;
  const int total_stages = nmine * NSTAGE;

  // ---- weight stream (as fd_edge_mlp.hip's 8-wave shape): every wave copies six 1 KB pieces of each 48 KB stage ----
  const char* __restrict__ img_lane = static_cast<const char*>(d.img) + wave * (EM_STAGE / EM_WAVES) + lane * 16;
  char* const lds_wave = lds + wave * (EM_STAGE / EM_WAVES);
  int issued = 0, consumed = 0;
  auto issue_stage = [&]() __attribute__((always_inline)) {
    const char* src = img_lane + (long)(issued % NSTAGE) * EM_STAGE;
    char* dst = lds_wave + (issued % EM_RING) * EM_STAGE;
    fd::glds16x4(src, dst);
    fd::glds16x2(src + 4096, dst + 4096);
    ++issued;
  };
  auto stage_begin = [&]() __attribute__((always_inline)) -> const char* {
    fd::wait_vmem();
    __syncthreads();
    const char* cur = lds + (consumed % EM_RING) * EM_STAGE + lane * 16;
    ++consumed;
    return cur;
  };
  auto stage_prefetch = [&]() __attribute__((always_inline)) {
    if (issued < total_stages) issue_stage();
  };
  issue_stage();

  char* const my_x = xch + wave * XSLOT + lane * 16;
  const char* const pa_x = xch + (wave ^ 1) * XSLOT + lane * 16;
  const char* const pair_x = xch + (wave & ~1) * XSLOT + lane * 16;     // slot of wave 0 of this pair; wave 1's is XSLOT further

  for (int ti = 0; ti < nmine; ++ti) {
    const int tile = first + ti * G;
    const long row = (long)tile * PAIR_ROWS + rg * 16 + m;
    const bool rok = row < rows;
    const long rc = rok ? row : rows - 1;
    const long qi = rc / d.nres;
    const long qj = (qi / d.nres) * d.nres + (rc - qi * d.nres);

    // the input row, all four k-steps of layer 1 as bf16 planes (lane (m, g) holds columns 16 nb + 4 g + r: chained k order)
    uint4 Xb[4][3];
    {
      float4 xv[8];
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) xv[nb] = ld4(d.x + rc * EM_C + 16 * nb + 4 * g);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float t[8] = {xv[2 * ks].x, xv[2 * ks].y, xv[2 * ks].z, xv[2 * ks].w,
                      xv[2 * ks + 1].x, xv[2 * ks + 1].y, xv[2 * ks + 1].z, xv[2 * ks + 1].w};
        em_split8(t, Xb[ks][0], Xb[ks][1], Xb[ks][2]);
      }
    }

    Em16Half H[2];
    uint4 b[3];
    // layer 2: the wave's 12 blocks -- local block 2 g6 + e is block 4 g6 + 2 p + e of the 24; the bias is the initial value
    f32x4 acc2[12];
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      const float4 v = ld4(d.bias2 + 16 * (4 * (a >> 1) + 2 * p + (a & 1)) + 4 * g);
      acc2[a][0] = v.x; acc2[a][1] = v.y; acc2[a][2] = v.z; acc2[a][3] = v.w;
    }
    f32x4 acc1[4];
    float4 zres[4];
#pragma clang loop unroll(full)
    for (int c = 0; c < 3; ++c) {
      // ---- layer 1, chunk c: local block 2 ng + e is block 4 ng + 2 p + e of the chunk's eight ----
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[a][r] = 0.f;
#pragma clang loop unroll(full)
      for (int sg = 0; sg < 2; ++sg) {
        const char* st = stage_begin() + p * (EM_UNIT / 2);
        em16_read_half(H[0], st);
#pragma clang loop unroll(full)
        for (int hh = 0; hh < 4; ++hh) {
          const int r = 4 * sg + hh, ks = r >> 1, a = 2 * (r & 1);
          EM_PIN_TOP();
          if (hh + 1 < 4) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * EM_UNIT);
          em16_mma_half(acc1[a], acc1[a + 1], H[hh & 1], Xb[ks]);
          EM_GROUPS(hh + 1 < 4);
          if (hh == 0) stage_prefetch();
        }
      }
      // epilogue 1: h1 = relu(acc + P1_i + Q1_j) on the wave's four blocks, left in its exchange slot
      fd::sched_pin();
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int col = 128 * c + 16 * (4 * (a >> 1) + 2 * p + (a & 1)) + 4 * g;
        const float4 pa = ld4(d.p1 + qi * ld_pq + col);
        const float4 qa = ld4(d.q1 + qj * ld_pq + col);
        const float v[4] = {acc1[a][0] + (pa.x + qa.x), acc1[a][1] + (pa.y + qa.y), acc1[a][2] + (pa.z + qa.z),
                            acc1[a][3] + (pa.w + qa.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1[a][e] = v[e] > 0.f ? v[e] : 0.f;
        st_lds4(my_x + a * 1024, acc1[a]);
      }
      // ---- layer 2, k in chunk c: k-step ks = blocks (2 ks, 2 ks + 1) of the chunk, owned by wave ks & 1 of the pair as its local
      // blocks 2 (ks >> 1), 2 (ks >> 1) + 1 (the first stage barrier below orders the slots' writes before these reads; the next
      // writes come two stage barriers after the chunk's last read) ----
#pragma clang loop unroll(full)
      for (int sg = 0; sg < 6; ++sg) {
        const char* st = stage_begin() + p * (EM_UNIT / 2);
        em16_read_half(H[0], st);
#pragma clang loop unroll(full)
        for (int hh = 0; hh < 4; ++hh) {
          const int u2 = 4 * sg + hh, ks = u2 / 6, g6 = u2 % 6, a = 2 * g6;
          EM_PIN_TOP();
          if (hh + 1 < 4) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * EM_UNIT);
          if (g6 == 0) {
            const char* src = pair_x + (ks & 1) * XSLOT + (2 * (ks >> 1)) * 1024;
            em16_split2(lds_f4(src), lds_f4(src + 1024), b[0], b[1], b[2]);
            if (c == 2 && ks == 3) {      // the residual into the final layer: the wave's four blocks of the first 128 hidden units
#pragma unroll
              for (int a4 = 0; a4 < 4; ++a4) zres[a4] = ld4(d.x + rc * EM_C + 16 * (4 * (a4 >> 1) + 2 * p + (a4 & 1)) + 4 * g);
            }
          }
          em16_mma_half(acc2[a], acc2[a + 1], H[hh & 1], b);
          EM_GROUPS(hh + 1 < 4);
          if (hh == 0) stage_prefetch();
        }
      }
    }
    // epilogue 2: h2 = relu(acc2) (+ z on the first 128 units)
#pragma unroll
    for (int a = 0; a < 12; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc2[a][e];
        acc2[a][e] = (v > 0.f ? v : 0.f) + (a < 4 ? (e == 0 ? zres[a & 3].x : e == 1 ? zres[a & 3].y : e == 2 ? zres[a & 3].z : zres[a & 3].w) : 0.f);
      }

    // ---- layer 3, K split: the wave's local blocks (2 s, 2 s + 1) are k-step 2 s + p; a stage holds the k-steps 2 sg (units 0, 1)
    // and 2 sg + 1 (units 2, 3), so the wave multiplies units 2 p, 2 p + 1 of stage sg with its blocks (2 sg, 2 sg + 1) ----
    f32x4 acc3[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc3[nb][r] = 0.f;
#pragma clang loop unroll(full)
    for (int sg = 0; sg < 6; ++sg) {
      const char* st = stage_begin() + (2 * p) * EM_UNIT;
      em16_read_half(H[0], st);
#pragma clang loop unroll(full)
      for (int q = 0; q < 4; ++q) {
        const int a = 4 * (q >> 1) + 2 * (q & 1);
        EM_PIN_TOP();
        if (q + 1 < 4) em16_read_half(H[(q + 1) & 1], st + (q + 1) * (EM_UNIT / 2));
        if (q == 0) em16_split2(acc2[2 * sg], acc2[2 * sg + 1], b[0], b[1], b[2]);
        em16_mma_half(acc3[a], acc3[a + 1], H[q & 1], b);
        EM_GROUPS(q + 1 < 4);
        if (q == 0) stage_prefetch();
      }
    }

    // ---- final epilogue: the wave finishes blocks {2p, 2p+1, 4+2p, 5+2p}; the partner's share of the partial sums goes through the slot
    f32x4 fin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nb0 = 4 * (i >> 1) + (i & 1);                  // wave 0's block; wave 1's is nb0 + 2
#pragma unroll
      for (int e = 0; e < 4; ++e) fin[i][e] = p ? acc3[nb0 + 2][e] : acc3[nb0][e];
      f32x4 oth;
#pragma unroll
      for (int e = 0; e < 4; ++e) oth[e] = p ? acc3[nb0][e] : acc3[nb0 + 2][e];
      st_lds4(my_x + i * 1024, oth);
    }
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = 16 * (4 * (i >> 1) + 2 * p + (i & 1)) + 4 * g;
      const f32x4 o = lds_f4(pa_x + i * 1024);
      const float4 pa = ld4(d.pf + qi * ld_pqf + col);
      const float4 qa = ld4(d.qf + qj * ld_pqf + col);
      fin[i][0] = (fin[i][0] + o[0]) + (pa.x + qa.x);
      fin[i][1] = (fin[i][1] + o[1]) + (pa.y + qa.y);
      fin[i][2] = (fin[i][2] + o[2]) + (pa.z + qa.z);
      fin[i][3] = (fin[i][3] + o[3]) + (pa.w + qa.w);
      s += (fin[i][0] + fin[i][1]) + (fin[i][2] + fin[i][3]);
    }
    // LayerNorm over the row's 128 values: this wave's 64 sit in four lanes (l & 15 fixed)
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean_h = s * (1.0f / 64.0f);
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dlt = fin[i][r] - mean_h;
        vs += dlt * dlt;
      }
    vs += __shfl_xor(vs, 16);
    vs += __shfl_xor(vs, 32);
    xst[wave * 64 + lane] = make_float2(mean_h, vs);
    __syncthreads();
    const float2 ps = xst[(wave ^ 1) * 64 + lane];
    const float dm = mean_h - ps.x;
    const float mean = 0.5f * (mean_h + ps.x);
    const float m2 = (vs + ps.y) + 32.0f * dm * dm;          // Chan et al.: n_a n_b / (n_a + n_b) = 32
    const float rstd = 1.0f / sqrtf(m2 * (1.0f / 128.0f) + d.eps);
    const float rs = d.rowscale != nullptr ? d.rowscale[rc] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = 16 * (4 * (i >> 1) + 2 * p + (i & 1)) + 4 * g;
      const float4 gm = ld4(d.gamma + col);
      const float4 bt = ld4(d.beta + col);
      float4 o;
      o.x = ((fin[i][0] - mean) * rstd * gm.x + bt.x) * rs;
      o.y = ((fin[i][1] - mean) * rstd * gm.y + bt.y) * rs;
      o.z = ((fin[i][2] - mean) * rstd * gm.z + bt.z) * rs;
      o.w = ((fin[i][3] - mean) * rstd * gm.w + bt.w) * rs;
      if (rok) *reinterpret_cast<float4*>(d.out + row * EM_C + col) = o;
      if (ZB) { fin[i][0] = o.x; fin[i][1] = o.y; fin[i][2] = o.z; fin[i][3] = o.w; }
    }
    if (ZB) {
      // ---- layer 4: zb[0:40] = W40 z' + b40, K split: the wave's blocks (2 t, 2 t + 1) are k-step 2 t + p = unit 2 t + p of the stage ----
      f32x4 acc4[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[nb][r] = 0.f;
      const char* st = stage_begin() + p * EM_UNIT;
      em16_read_half(H[0], st);
#pragma clang loop unroll(full)
      for (int q = 0; q < 4; ++q) {
        const int a = 2 * (q & 1);
        EM_PIN_TOP();
        if (q + 1 < 4) em16_read_half(H[(q + 1) & 1], st + ((q + 1) >> 1) * 2 * EM_UNIT + ((q + 1) & 1) * (EM_UNIT / 2));
        if ((q & 1) == 0) em16_split2(fin[q], fin[q + 1], b[0], b[1], b[2]);
        em16_mma_half(acc4[a], acc4[a + 1], H[q & 1], b);
        EM_GROUPS(q + 1 < 4);
        if (q == 0) stage_prefetch();
      }
      if (p == 1) {
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) st_lds4(my_x + nb * 1024, acc4[nb]);
      }
      __syncthreads();
      if (p == 0) {
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
          const int col = 16 * nb + 4 * g;
          if (col < EM_ZB) {
            const f32x4 o = lds_f4(pa_x + nb * 1024);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d.zb_bias != nullptr) a = ld4(d.zb_bias + col);
            if (rok)
              *reinterpret_cast<float4*>(d.zb_out + row * EM_ZB + col) =
                  make_float4(a.x + (acc4[nb][0] + o[0]), a.y + (acc4[nb][1] + o[1]), a.z + (acc4[nb][2] + o[2]), a.w + (acc4[nb][3] + o[3]));
          }
        }
      }
    }
  }
}

}  // namespace

int fd_edge_mlp_launch_pair(const FdEdgeMlpDesc& d, hipStream_t st) {
  const long ntiles = (d.rows + PAIR_ROWS - 1) / PAIR_ROWS;
  const int blocks = d.blocks > 0 ? d.blocks : 256;        // one 8-wave block per CU
  const int grid = (int)(ntiles < blocks ? ntiles : blocks);
  FdEdgeMlpDesc dd = d;
  dd.sched = nullptr;                                       // static tile stride (the shape is for launches of one tile per block)
  if (d.zb_out != nullptr)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(edge_mlp_pair_kernel<true>), dim3(grid), dim3(512), 0, st, dd);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(edge_mlp_pair_kernel<false>), dim3(grid), dim3(512), 0, st, dd);
  FD_CHECK_LAUNCH("fd_edge_mlp(pair)");
  return FD_OK;
}
