// Register-chained residual MLP blocks of the node level (one launch per block and direction instead of one per layer):
//   StructureModuleTransition (model/ipa_pytorch.py:169-191):   n3 = mask * LN(n2 + W3 relu(W2 relu(W1 n2 + b1) + b2) + b3)
//   the feed-forward half of a TransformerEncoderLayer (torch.nn, post-norm, ipa_pytorch.py:584-595):
//                                                                y2 = LN(y1 + W2 relu(W1 y1 + b1) + b2)
// and their input-gradient chains (LayerNorm backward -> gated transposed products -> residual).  Same machinery as the fused
// pair-level kernels (fd_chain.h): a wave owns 16 rows for the whole chain and accumulates transposed with
// v_mfma_f32_16x16x32_bf16 on 3-term bf16 splits (fp32-accurate products, fp32 sums), the layer outputs never leave the registers
// (two accumulator sets of width W ping-pong), the weights stream as pre-split 12 KB units through an LDS ring by
// LDS-DMA (a six-stage ring here).  At M = B*N = 3,840 rows the 4 / 3 launches this replaces are 64x64-tile GEMMs of 240-300 blocks with a serial
// K loop each plus a LayerNorm launch (~20-35 us each, latency-bound); the chain is one 60-block launch whose length is one wave's
// MFMA chain.  Every layer is W x W (W = 256: transition, W = 320: transformer), NL = 3 / 2 layers.
#include "fd_common.h"
#include "fd_experiments.h"

namespace {

#include "fd_chain.h"

// Stages of the weight ring.  A block has its CU to itself here (60 blocks at M = 3,840 rows, 2 at M = 128), so a stage is ~770
// cycles of one wave's MFMAs and a copy issued one stage ahead (the two-stage ring of the pair-level kernels, where eight waves
// share the CU and a stage takes ~3,000 cycles) would expose the ~1.2 us LDS-DMA latency on EVERY stage: 65 us per launch at
// M = 128.  Six stages (144 KB) keep five copies in flight.
constexpr int NC_RING = 6;

struct NcMat {
  const float* p;     // A[n][k] = p[n * rs + k * cs]
  long rs, cs;
};

// One layer (W x W) of an image: (W/32) k-steps x (W/64) n-groups of units, k-step major; unit = [4 n-blocks][3 planes][64 lanes]
// x 16 B.  chained = 0: k = k0 + 8 g + e' (operand from memory); 1: k = k0 + 16 (e' >> 2) + 4 g + (e' & 3) (operand = the
// previous layer's accumulators).
__global__ __launch_bounds__(256) void node_chain_pack_kernel(NcMat A, int W, int chained, char* __restrict__ img) {
  const int NG = W / 64, NU = (W / 32) * NG;
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (unit, n-block, lane)
  if (gid >= NU * 4 * 64) return;
  const int lane = gid & 63, i = (gid >> 6) & 3, u = gid >> 8;
  const int m = lane & 15, g = lane >> 4;
  const int n = 64 * (u % NG) + 16 * i + m, k0 = 32 * (u / NG);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = chained ? k0 + 16 * (e >> 2) + 4 * g + (e & 3) : k0 + 8 * g + e;
    x[e] = A.p[(long)n * A.rs + (long)k * A.cs];
  }
  uint4 s0, s1, s2;
  em_split8(x, s0, s1, s2);
  char* dst = img + (long)u * EM_UNIT + (i * 3) * EM_PIECE + lane * 16;
  *reinterpret_cast<uint4*>(dst) = s0;
  *reinterpret_cast<uint4*>(dst + EM_PIECE) = s1;
  *reinterpret_cast<uint4*>(dst + 2 * EM_PIECE) = s2;
}

template <int W, int NL, bool BWD>
__global__ __launch_bounds__(64 * EM_WAVES, 1) void node_chain_kernel(FdNodeChainDesc d) {
  constexpr int NB = W / 16, KS = W / 32, NG = W / 64, UL = KS * NG;     // n-blocks, k-steps, n-groups, units per layer
  constexpr int NSTAGE = NL * UL / EM_UPS;
  static_assert(UL % EM_UPS == 0, "a layer is a whole number of stages");
  __shared__ __attribute__((aligned(16))) char lds[NC_RING * EM_STAGE];
  __shared__ float lnacc[BWD ? 2 * W : 1];           // dgamma | dbeta of the LayerNorm backward
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const long rows = d.rows;
  const int ntiles = (int)((rows + EM_ROWS - 1) / EM_ROWS);
  const int G = (int)gridDim.x, first = (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nmine = (ntiles - first + G - 1) / G;
  const int total_stages = nmine * NSTAGE;

  // ---- weight stream (as fd_edge_mlp.hip): every wave copies a quarter (6 pieces) of each stage ----
  const char* __restrict__ img_lane = static_cast<const char*>(d.img) + wave * (EM_STAGE / EM_WAVES) + lane * 16;
  char* const lds_wave = lds + wave * (EM_STAGE / EM_WAVES);
  int issued = 0, consumed = 0;
  auto issue_stage = [&]() __attribute__((always_inline)) {
    const char* src = img_lane + (long)(issued % NSTAGE) * EM_STAGE;
    char* dst = lds_wave + (issued % NC_RING) * EM_STAGE;
    fd::glds16x4(src, dst);
    fd::glds16x2(src + 4096, dst + 4096);
    ++issued;
  };
  auto stage_begin = [&]() __attribute__((always_inline)) -> const char* {
    // the copies of the stages after this one (6 LDS-DMA instructions per wave and stage, issued after this stage's) may stay in
    // flight: vmcnt retires in issue order, so "at most 6 k outstanding" means this stage's copy has landed (any other memory
    // operation issued since only makes the wait stricter)
    switch (issued - consumed - 1) {
      case 0: fd::wait_vmem(); break;
      case 1: fd::wait_vmem_keep<6>(); break;
      case 2: fd::wait_vmem_keep<12>(); break;
      case 3: fd::wait_vmem_keep<18>(); break;
      default: fd::wait_vmem_keep<24>(); break;
    }
    __syncthreads();
    const char* cur = lds + (consumed % NC_RING) * EM_STAGE + lane * 16;
    ++consumed;
    return cur;
  };
  auto stage_prefetch = [&]() __attribute__((always_inline)) {
    if (issued < total_stages) issue_stage();
  };
  for (int i = 0; i < NC_RING - 1 && i < total_stages; ++i) issue_stage();
  if (BWD) {
    for (int i = tid; i < 2 * W; i += 64 * EM_WAVES) lnacc[i] = 0.f;
    __syncthreads();
  }

  uint4 b[3];          // activation planes (B operand) of the current k-step
  Em16Half H[2];       // fragments of the current / next half-unit

  // one W x W layer whose operand is the previous layer's accumulators (chained k order)
  auto layer_chained = [&](const f32x4 (&in)[NB], f32x4 (&out)[NB]) __attribute__((always_inline)) {
#pragma clang loop unroll(full)
    for (int sg = 0; sg < UL / EM_UPS; ++sg) {
      const char* st = stage_begin();
      em16_read_half(H[0], st);
#pragma clang loop unroll(full)
      for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
        const int u = EM_UPS * sg + (hh >> 1), ks = u / NG, ng = u % NG, a = 4 * ng + 2 * (hh & 1);
        if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
        fd::sched_pin();
        if (ng == 0 && (hh & 1) == 0) em16_split2(in[2 * ks], in[2 * ks + 1], b[0], b[1], b[2]);
        em16_mma_half(out[a], out[a + 1], H[hh & 1], b);
        if (hh == 1) stage_prefetch();
      }
    }
  };
  auto init_acc = [&](f32x4 (&acc)[NB], const float* __restrict__ bias) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias != nullptr) a = *reinterpret_cast<const float4*>(bias + 16 * nb + 4 * g);
      acc[nb][0] = a.x; acc[nb][1] = a.y; acc[nb][2] = a.z; acc[nb][3] = a.w;
    }
  };

  for (int ti = 0; ti < nmine; ++ti) {
    const long row = ((long)first + (long)ti * G) * EM_ROWS + wave * 16 + m;
    const bool rok = row < rows;
    const long rc = rok ? row : rows - 1;         // rows past the end are clamped on load, masked on store
    f32x4 P[NB], Q[NB];

    if (!BWD) {
      // ---- layer 1: operand from memory, natural k order (k = 32 ks + 8 g + e) ----
      {
        float xr[KS][8];
        const float* xp = d.x + rc * W + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 v = *reinterpret_cast<const float4*>(xp + 32 * ks);
          const float4 w = *reinterpret_cast<const float4*>(xp + 32 * ks + 4);
          xr[ks][0] = v.x; xr[ks][1] = v.y; xr[ks][2] = v.z; xr[ks][3] = v.w;
          xr[ks][4] = w.x; xr[ks][5] = w.y; xr[ks][6] = w.z; xr[ks][7] = w.w;
        }
        init_acc(P, d.bias[0]);
#pragma clang loop unroll(full)
        for (int sg = 0; sg < UL / EM_UPS; ++sg) {
          const char* st = stage_begin();
          em16_read_half(H[0], st);
#pragma clang loop unroll(full)
          for (int hh = 0; hh < 2 * EM_UPS; ++hh) {
            const int u = EM_UPS * sg + (hh >> 1), ks = u / NG, ng = u % NG, a = 4 * ng + 2 * (hh & 1);
            if (hh + 1 < 2 * EM_UPS) em16_read_half(H[(hh + 1) & 1], st + (hh + 1) * (EM_UNIT / 2));
            fd::sched_pin();
            if (ng == 0 && (hh & 1) == 0) em_split8(xr[ks], b[0], b[1], b[2]);
            em16_mma_half(P[a], P[a + 1], H[hh & 1], b);
            if (hh == 1) stage_prefetch();
          }
        }
      }
      // hidden layers: ReLU, optional save (training)
      auto relu_save = [&](f32x4 (&acc)[NB], float* __restrict__ save) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nb][e] = acc[nb][e] > 0.f ? acc[nb][e] : 0.f;
          if (save != nullptr && rok)
            *reinterpret_cast<float4*>(save + row * W + 16 * nb + 4 * g) = make_float4(acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]);
        }
      };
      // last layer: + x (residual), optional save of the pre-LayerNorm row, LayerNorm, row scale
      auto final_ln = [&](f32x4 (&acc)[NB]) __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = 16 * nb + 4 * g;
          const float4 xa = *reinterpret_cast<const float4*>(d.x + rc * W + col);
          acc[nb][0] += xa.x; acc[nb][1] += xa.y; acc[nb][2] += xa.z; acc[nb][3] += xa.w;
          s += (acc[nb][0] + acc[nb][1]) + (acc[nb][2] + acc[nb][3]);
          if (d.pre != nullptr && rok)
            *reinterpret_cast<float4*>(d.pre + row * W + col) = make_float4(acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / W);
        float vs = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dlt = acc[nb][r] - mean;
            acc[nb][r] = dlt;
            vs += dlt * dlt;
          }
        vs += __shfl_xor(vs, 16);
        vs += __shfl_xor(vs, 32);
        const float rstd = 1.0f / sqrtf(vs * (1.0f / W) + d.eps);
        const float rs = d.rowscale != nullptr ? d.rowscale[rc] : 1.f;
        if (rok && g == 0) {
          if (d.mean != nullptr) d.mean[row] = mean;
          if (d.rstd != nullptr) d.rstd[row] = rstd;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = 16 * nb + 4 * g;
          const float4 gm = *reinterpret_cast<const float4*>(d.gamma + col);
          const float4 bt = *reinterpret_cast<const float4*>(d.beta + col);
          float4 o;
          o.x = (acc[nb][0] * rstd * gm.x + bt.x) * rs;
          o.y = (acc[nb][1] * rstd * gm.y + bt.y) * rs;
          o.z = (acc[nb][2] * rstd * gm.z + bt.z) * rs;
          o.w = (acc[nb][3] * rstd * gm.w + bt.w) * rs;
          if (rok) *reinterpret_cast<float4*>(d.out + row * W + col) = o;
        }
      };
      relu_save(P, d.save[0]);
      init_acc(Q, d.bias[1]);
      layer_chained(P, Q);
      if (NL == 2) {
        final_ln(Q);
      } else {
        relu_save(Q, d.save[1]);
        init_acc(P, d.bias[2]);
        layer_chained(Q, P);
        final_ln(P);
      }
    } else {
      // ---- prologue: LayerNorm backward of the upstream gradient, in layer-output layout (lane (m, g): columns 16 nb + 4 g + r)
      // dt = rstd * (gy*gamma - mean_c(gy*gamma) - xhat * mean_c(gy*gamma*xhat)),  gy = rowscale * dy,  xhat = (t - mean) * rstd ----
      {
        const float rs = (d.rowscale != nullptr ? d.rowscale[rc] : 1.f) * (rok ? 1.f : 0.f);
        const float mean = d.mean[rc], rstd = d.rstd[rc];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = 16 * nb + 4 * g;
          const float4 dv = *reinterpret_cast<const float4*>(d.x + rc * W + col);
          const float4 tv = *reinterpret_cast<const float4*>(d.ln_in + rc * W + col);
          const float4 gm = *reinterpret_cast<const float4*>(d.gamma + col);
          const float up[4] = {dv.x, dv.y, dv.z, dv.w}, hv[4] = {tv.x, tv.y, tv.z, tv.w}, gv[4] = {gm.x, gm.y, gm.z, gm.w};
          float cg[4], cb[4];      // this tile's 16-row sums for dgamma / dbeta of columns col .. col + 3 (DPP row reduction)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (hv[e] - mean) * rstd;
            const float gy = up[e] * rs;
            cg[e] = fd::row16_sum(gy * xh);
            cb[e] = fd::row16_sum(gy);
            const float t = gy * gv[e];
            P[nb][e] = t;
            s1 += t;
            s2 += t * xh;
          }
          if (m == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fd::lds_add(&lnacc[col + e], cg[e]);
              fd::lds_add(&lnacc[W + col + e], cb[e]);
            }
          }
        }
        s1 += __shfl_xor(s1, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16);
        s2 += __shfl_xor(s2, 32);
        const float m1 = s1 * (1.0f / W), m2 = s2 * (1.0f / W);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = 16 * nb + 4 * g;
          const float4 tv = *reinterpret_cast<const float4*>(d.ln_in + rc * W + col);     // (second read: a cache hit)
          const float hv[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) P[nb][e] = rstd * (P[nb][e] - m1 - (hv[e] - mean) * rstd * m2);
          if (rok) *reinterpret_cast<float4*>(d.pre + row * W + col) = make_float4(P[nb][0], P[nb][1], P[nb][2], P[nb][3]);
        }
      }
      // hidden layers of the backward: gate on the forward's saved activation, save (operand of the weight gradient)
      auto gate_save = [&](f32x4 (&acc)[NB], const float* __restrict__ gate, float* __restrict__ save) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = 16 * nb + 4 * g;
          const float4 gt = *reinterpret_cast<const float4*>(gate + rc * W + col);
          acc[nb][0] = gt.x > 0.f ? acc[nb][0] : 0.f; acc[nb][1] = gt.y > 0.f ? acc[nb][1] : 0.f;
          acc[nb][2] = gt.z > 0.f ? acc[nb][2] : 0.f; acc[nb][3] = gt.w > 0.f ? acc[nb][3] : 0.f;
          if (rok) *reinterpret_cast<float4*>(save + row * W + col) = make_float4(acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]);
        }
      };
      // last layer: dx = acc + dt (the residual path; dt re-read from this lane's own store above)
      auto final_resid = [&](f32x4 (&acc)[NB]) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = 16 * nb + 4 * g;
          if (rok) {
            const float4 tv = *reinterpret_cast<const float4*>(d.pre + row * W + col);
            *reinterpret_cast<float4*>(d.out + row * W + col) =
                make_float4(acc[nb][0] + tv.x, acc[nb][1] + tv.y, acc[nb][2] + tv.z, acc[nb][3] + tv.w);
          }
        }
      };
      init_acc(Q, nullptr);
      layer_chained(P, Q);
      gate_save(Q, d.gate[0], d.save[0]);
      init_acc(P, nullptr);
      layer_chained(Q, P);
      if (NL == 2) {
        final_resid(P);
      } else {
        gate_save(P, d.gate[1], d.save[1]);
        init_acc(Q, nullptr);
        layer_chained(P, Q);
        final_resid(Q);
      }
    }
  }
  if (BWD) {
    __syncthreads();
    for (int i = tid; i < W; i += 64 * EM_WAVES) {
      if (d.dgamma != nullptr) atomicAdd(d.dgamma + i, lnacc[i]);
      if (d.dbeta != nullptr) atomicAdd(d.dbeta + i, lnacc[W + i]);
    }
  }
}

}  // namespace

extern "C" int fd_node_chain_pack(const float* A, long rs, long cs, int width, int chained, void* img_layer, void* stream) {
  FD_CHECK_ARG(A && img_layer, "fd_node_chain_pack: null operand");
  FD_CHECK_ARG(width == 256 || width == 320, "fd_node_chain_pack: width must be 256 or 320");
  FD_CHECK_ARG(fd_aligned16(img_layer), "fd_node_chain_pack: image must be 16-byte aligned");
  const int nu = (width / 32) * (width / 64);
  NcMat m{A, rs, cs};
  hipLaunchKernelGGL(node_chain_pack_kernel, dim3((nu * 4 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, m, width,
                     chained, static_cast<char*>(img_layer));
  FD_CHECK_LAUNCH("fd_node_chain_pack");
  return FD_OK;
}

extern "C" int fd_node_chain(const FdNodeChainDesc* desc, void* stream) {
  FD_CHECK_ARG(desc != nullptr, "fd_node_chain: null descriptor");
  const FdNodeChainDesc& d = *desc;
  FD_CHECK_ARG((d.width == 256 && d.nlayers == 3) || (d.width == 320 && d.nlayers == 2),
               "fd_node_chain: (width, nlayers) must be (256, 3) or (320, 2)");
  FD_CHECK_ARG(d.x && d.img && d.out && d.gamma && d.rows >= 0, "fd_node_chain: x / img / out / gamma are required");
  if (d.backward) {
    FD_CHECK_ARG(d.ln_in && d.mean && d.rstd && d.pre, "fd_node_chain(backward): ln_in / mean / rstd / pre are required");
    for (int l = 0; l + 1 < d.nlayers; ++l)
      FD_CHECK_ARG(d.gate[l] && d.save[l], "fd_node_chain(backward): gate[l] / save[l] are required for every hidden layer");
  } else {
    FD_CHECK_ARG(d.beta != nullptr, "fd_node_chain(forward): beta is required");
  }
  const void* ptrs[] = {d.x, d.img, d.out, d.bias[0], d.bias[1], d.bias[2], d.save[0], d.save[1], d.gate[0], d.gate[1],
                        d.pre, d.ln_in, d.gamma, d.beta};
  for (const void* p : ptrs) FD_CHECK_ARG(fd_aligned16(p), "fd_node_chain: operands must be 16-byte aligned");
  if (d.rows == 0) return FD_OK;
  const long ntiles = (d.rows + EM_ROWS - 1) / EM_ROWS;
  const int blocks = d.blocks > 0 ? d.blocks : 256;    // one block per CU (144 KB of LDS)
  const int grid = (int)(ntiles < blocks ? ntiles : blocks);
  const dim3 g3(grid), b3(64 * EM_WAVES);
  hipStream_t st = (hipStream_t)stream;
  if (d.width == 256 && !d.backward)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(node_chain_kernel<256, 3, false>), g3, b3, 0, st, d);
  else if (d.width == 256)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(node_chain_kernel<256, 3, true>), g3, b3, 0, st, d);
  else if (!d.backward)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(node_chain_kernel<320, 2, false>), g3, b3, 0, st, d);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(node_chain_kernel<320, 2, true>), g3, b3, 0, st, d);
  FD_CHECK_LAUNCH("fd_node_chain");
  return FD_OK;
}
