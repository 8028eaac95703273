// Split-bf16 GEMM against PRE-SPLIT weights (fd_gemm tile codes 12 / 13 / 14) for the node-level layers of a training step:
// M = B*N residue rows of a few thousand against the per-residue nn.Linear weights (model/ipa_pytorch.py:334-374 IPA
// projections, :465-469 linear_out, :169-191 node transition, :584-595 sequence transformer, :625-649 block body) -- the
// forward y = x W^T and the activation gradient dx = dy W.  Included by fd_gemm.hip inside its anonymous namespace (uses
// GemmArgs, store_tile, store_tile_t, s64_split4).
//
// What is different from tiles 4 / 10 (which split BOTH operands in registers on the way to LDS):
//   * the weight operand arrives already split: fd_split_planes writes the three exact bf16 planes of the whole flat parameter
//     buffer once per step (FdGemmDesc.b_planes / b_plane_stride address the plane element of B's first element), so a
//     block moves 6 bytes per weight element L2 -> registers -> LDS and spends no VALU on them; the SAME planes serve both
//     directions -- k-contiguous for y = x W^T ([row][16 k] images read with ds_read_b128), row-contiguous for dx = dy W
//     ([k][32 columns] images read with ds_read_b64_tr_b16, the LDS transpose read);
//   * 128 x 128 / 64 x 128 / 64 x 64 block tiles on 2 x 2 waves (wave tile up to 64 x 64: 24 MFMAs per 12 fragment reads and
//     16-k stage instead of 6 per 6), two blocks per CU so that one block's prologue / epilogue runs under the other's MFMA loop;
//   * NO LDS WRITE from registers anywhere in the loop: both operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4,
//     lane-private source, wave-linear destination) -- the weight planes as they are, the activations as raw fp32 -- and the
//     activation FRAGMENT is split into its three planes in the registers of the wave that multiplies it.  (Measured on the first
//     form of this kernel, which staged through registers like tiles 4 / 10: with the ds_write instructions removed a launch took
//     34 instead of 59 us; the split VALU alone was free, and two co-resident blocks did not overlap at all.)
//   * LDS ring of THREE 16-k stages, copies issued TWO stages ahead; the loop issues no other vector-memory operation, so
//     s_waitcnt vmcnt(copies per stage) at the end of a stage says exactly "the next stage has landed";
//   * conflict-free without padding: the 16-byte chunks of a row are placed by XOR with row bits -- bit 3 for a plane's 32-byte
//     slice, bits 2-3 for the 64-byte fp32 slice -- so that the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27},
//     {4-11,16-19,28-31}, ...) cover all 64 banks;
//   * transposed accumulation (the weight fragment is the MFMA's row operand): a lane owns four consecutive columns of an output
//     row, the epilogue (bias / ReLU / gate / row scale / residual / old C) moves float4s between registers and memory
//     (store_tile_t); split-K launches accumulate row-major and add their partial tiles atomically (store_tile).
// Needs: A k-contiguous, K % 16 == 0, 16-byte aligned operands, un-batched, B unit-stride along k (KC) or along n (N % 8 == 0).
#ifdef GW_RING
#include "fd_probe.h"
#endif
#ifdef GW_ABL               // timing-only ablations (WRONG RESULTS by design): 1 no loop loads, 2 no LDS writes, 4 no split, 8 no MFMAs,
#include "fd_probe.h"       // 16 no epilogue, 32 no fragment reads -- tools/probes builds only (-DFD_PROBE_BUILD)
#else
#define GW_ABL 0
#endif
constexpr int W_BK = 16;
typedef fd::u32x4 w_u32x4;   // (a native 16-byte vector: HIP's uint4 struct, copied whole between pointers, kept the plane
                             //  registers in scratch memory)
constexpr int W_TR_PST = W_BK * 64 + 64;      // row-contiguous B: one 32-column panel of one plane, [16 k][64 B] + pad

template <int BM, int BN, bool B_KC>
struct WCfg {
  static constexpr int TM = BM / 64, TN = BN / 64;          // 32 x 32 MFMA blocks of a wave (2 x 2 waves)
  static constexpr int A_BYTES = BM * 64;                   // raw fp32 [row][16 k], the row's four 16-byte chunks swizzled
  static constexpr int B_PLANE = B_KC ? BN * 32 : (BN / 32) * W_TR_PST;
  static constexpr int B_BYTES = 3 * B_PLANE;
  static constexpr int STAGE = A_BYTES + B_BYTES;
#ifdef GW_RING
  static constexpr int RING = GW_RING;                      // (probe builds: ring depth A/B)
#else
  static constexpr int RING = 3;
#endif
  static constexpr int LDS = RING * STAGE;
  static constexpr int NIA = BM / 16;                       // LDS-DMA wave-instructions (1 KB each) of A per stage
  static constexpr int NIB = 3 * BN / 32;                   // ... of the three B planes
  static constexpr int NPW = (NIA + NIB + 3) / 4;           // per wave (a wave whose share is short repeats a piece)
  static constexpr int BLOCKS_PER_CU = 2 * LDS <= 160 * 1024 ? 2 : 1;
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <int BM, int BN, bool B_KC, bool TRANS>
__global__ __launch_bounds__(256, (WCfg<BM, BN, B_KC>::BLOCKS_PER_CU)) void gemm_w_kernel(GemmArgs g) {
  using Cfg = WCfg<BM, BN, B_KC>;
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NPW = Cfg::NPW, NIA = Cfg::NIA, NIB = Cfg::NIB;
  constexpr int B_PLANE = Cfg::B_PLANE, A_BYTES = Cfg::A_BYTES, STAGE = Cfg::STAGE;
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS];
  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int nblk = g.nblk_m * g.nblk_n;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, nblk);
  // logically consecutive blocks run on one XCD (fd_xcd_swizzle) and share its 4 MB L2: raster 0 walks the column blocks of an
  // M tile (the activation panel stays, the weights stream), raster 1 the M tiles of a column block (the weight panel stays)
  const int bm = g.raster ? lid % g.nblk_m : lid / g.nblk_n, bn = g.raster ? lid / g.nblk_m : lid % g.nblk_n;
  const int m0 = bm * BM, n0 = bn * BN;

  // split-K: blockIdx.z owns stages [kt0, kt0 + nk)
  const int nkt_all = d.K / W_BK;
  const int per = (nkt_all + g.ksplit - 1) / g.ksplit;
  const int kt0 = (int)blockIdx.z * per;
  const int nkt = (kt0 + per < nkt_all) ? kt0 + per : nkt_all;
  const int nk = nkt - kt0;
  if (nk <= 0) return;

  // ---- the LDS-DMA pieces of this wave: piece ii = (wave + 4 i) mod (NIA + NIB) of a stage ----
  //   A piece (ii < NIA): rows 16 ii .. 16 ii + 15, lane -> row + (lane >> 2), chunk position lane & 3 holds k chunk
  //                       (lane & 3) ^ ((row >> 2) & 3)   (rows past M are clamped: their outputs are never stored)
  //   B piece, k-contiguous weights: plane pl, rows 32 rb .. + 31, lane -> row + (lane >> 1), position lane & 1 holds k half
  //                       (lane & 1) ^ ((row >> 3) & 1)
  //   B piece, row-contiguous weights: plane pl, 32-column panel, lane -> k = lane >> 2, columns 8 (lane & 3) .. + 7
  const char* src[NPW];      // per-lane source of the NEXT stage to copy
  int dst[NPW];              // wave-uniform destination inside a stage
  long kstep[NPW];
  const unsigned short* __restrict__ Bp = reinterpret_cast<const unsigned short*>(d.b_planes);
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int ii = (wave + 4 * i) % (NIA + NIB);
    if (ii < NIA) {
      const int row = 16 * ii + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      const int gr = (m0 + row < d.M) ? m0 + row : d.M - 1;
      src[i] = reinterpret_cast<const char*>(d.A + (long)gr * d.a_rs + (long)(kt0 * W_BK + 4 * c));
      dst[i] = ii * 1024;
      kstep[i] = W_BK * 4;
    } else {
      const int jb = ii - NIA, pl = jb / (BN / 32), rb = jb % (BN / 32);
      if (B_KC) {
        const int row = 32 * rb + (lane >> 1);
        const int hs = (lane & 1) ^ ((row >> 3) & 1);
        const int gn = (n0 + row < d.N) ? n0 + row : d.N - 1;
        src[i] = reinterpret_cast<const char*>(Bp + (long)pl * d.b_plane_stride + (long)gn * d.b_cs + (long)(kt0 * W_BK + 8 * hs));
        dst[i] = A_BYTES + pl * B_PLANE + rb * 1024;
        kstep[i] = W_BK * 2;
      } else {
        const int k = lane >> 2, c = lane & 3;
        const int gn = (n0 + 32 * rb + 8 * c < d.N) ? n0 + 32 * rb + 8 * c : d.N - 8;     // (N % 8 == 0)
        src[i] = reinterpret_cast<const char*>(Bp + (long)pl * d.b_plane_stride + (long)(kt0 * W_BK + k) * d.b_rs + gn);
        dst[i] = A_BYTES + pl * B_PLANE + rb * W_TR_PST;
        kstep[i] = (long)W_BK * d.b_rs * 2;
      }
    }
  }
  // copies past the end of the k range re-read the last stage (the sources stop advancing) into a ring slot nobody reads again
  int cp = 0;          // stage index of the next copy
  auto copy = [&](char* st) __attribute__((always_inline)) {
    const bool adv = cp + 1 < nk;
    ++cp;
    if ((GW_ABL & 1) && cp > 2) return;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      fd::glds16a(src[i], st + dst[i]);
      src[i] += adv ? kstep[i] : 0;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses of this lane.  A: row l31 of a 32-row block, k half h = chunks 2h, 2h + 1 at positions c ^ ((row >> 2) & 3)
  const int sa = (l31 >> 2) & 3;
  const int fa_row = (wm * TM * 32 + l31) * 64;
  const int fa_p0 = fa_row + (((2 * h) ^ sa) * 16), fa_p1 = fa_row + (((2 * h + 1) ^ sa) * 16);
  const int fb_off = A_BYTES + (B_KC ? (wn * TN * 32 + l31) * 32 + ((h ^ ((l31 >> 3) & 1)) * 16)
                                     : (wn * TN) * W_TR_PST + (8 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
  auto step = [&](const char* st) __attribute__((always_inline)) {
    uint4 fa[TM][3], fb[TN][3];
    float4 xa[TM][2];
    if (!(GW_ABL & 32)) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        xa[i][0] = *reinterpret_cast<const float4*>(st + fa_p0 + i * 32 * 64);
        xa[i][1] = *reinterpret_cast<const float4*>(st + fa_p1 + i * 32 * 64);
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (B_KC) {
            fb[j][pl] = *reinterpret_cast<const uint4*>(st + fb_off + pl * B_PLANE + j * 32 * 32);
          } else {
            const char* q = st + fb_off + pl * B_PLANE + j * W_TR_PST;
            const uint2 lo = fd::lds_read_tr16(q), hi = fd::lds_read_tr16(q + 256);
            fb[j][pl] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
    }
    // the activation fragment is split HERE, in the registers of the wave that multiplies it (no LDS write on the path)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      uint2 a0, a1, a2, b0, b1, b2;
      if (GW_ABL & 4) {
        a0 = make_uint2(__float_as_uint(xa[i][0].x), __float_as_uint(xa[i][0].y)); a1 = a0; a2 = a0;
        b0 = make_uint2(__float_as_uint(xa[i][1].x), __float_as_uint(xa[i][1].y)); b1 = b0; b2 = b0;
      } else {
        s64_split4(xa[i][0], a0, a1, a2);
        s64_split4(xa[i][1], b0, b1, b2);
      }
      fa[i][0] = make_uint4(a0.x, a0.y, b0.x, b0.y);
      fa[i][1] = make_uint4(a1.x, a1.y, b1.x, b1.y);
      fa[i][2] = make_uint4(a2.x, a2.y, b2.x, b2.y);
    }
    if (GW_ABL & 8) {
      acc[0][0][0] += __uint_as_float(fa[0][0].x ^ fb[0][0].x ^ fa[TM - 1][2].w ^ fb[TN - 1][2].w);
      return;
    }
    // the six products with i + j <= 2, small terms first; the TM x TN accumulators interleaved
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
      constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = TRANS ? fd::mfma_32x32x16_bf16(fb[j][PB[p]], fa[i][PA[p]], acc[i][j])
                            : fd::mfma_32x32x16_bf16(fa[i][PA[p]], fb[j][PB[p]], acc[i][j]);
    }
  };

  // stage s lives in ring slot s % RING and is copied RING - 1 stages before it is multiplied.  The vector-memory counter retires in
  // issue order and the loop issues nothing but these copies: vmcnt((RING - 2) NPW) at the end of a stage = the next stage has
  // landed, the RING - 2 behind it stay in flight; __syncthreads() (lgkmcnt(0) + s_barrier) publishes it and frees the slot the
  // next copy overwrites
  constexpr int RING = Cfg::RING;
  int wslot = 0, rslot = 0;
#pragma unroll
  for (int u = 0; u < RING - 1; ++u) {
    copy(lds + wslot * STAGE);
    wslot = wslot + 1 == RING ? 0 : wslot + 1;
  }
  fd::wait_vmem_keep<(RING - 2) * NPW>();
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < nk; ++s) {
    copy(lds + wslot * STAGE);
    wslot = wslot + 1 == RING ? 0 : wslot + 1;
    step(lds + rslot * STAGE);
    rslot = rslot + 1 == RING ? 0 : rslot + 1;
    fd::wait_vmem_keep<(RING - 2) * NPW>();
    __syncthreads();
  }
  fd::wait_vmem();

  float* __restrict__ C = d.C;
  if ((GW_ABL & 16) && acc[0][0][0] != 12345.678f) return;
  if (TRANS)
    store_tile_t<TM, TN>(d, C, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, h, l31, g.epi_vec != 0);
  else
    store_tile<TM, TN>(d, C, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, h, l31, g.ksplit > 1);
}

// element-wise split of a flat fp32 buffer into its three exact bf16 planes: planes[p * n + e], p = 0 (leading) .. 2
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long n8, long n,
                                                           unsigned short* __restrict__ planes) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(x + 8 * i), b = *reinterpret_cast<const float4*>(x + 8 * i + 4);
    uint2 a0, a1, a2, b0, b1, b2;
    s64_split4(a, a0, a1, a2);
    s64_split4(b, b0, b1, b2);
    *reinterpret_cast<uint4*>(planes + 8 * i) = make_uint4(a0.x, a0.y, b0.x, b0.y);
    *reinterpret_cast<uint4*>(planes + n + 8 * i) = make_uint4(a1.x, a1.y, b1.x, b1.y);
    *reinterpret_cast<uint4*>(planes + 2 * n + 8 * i) = make_uint4(a2.x, a2.y, b2.x, b2.y);
  }
}

// can tiles 12 / 13 / 14 run this descriptor?
bool w_ok(const FdGemmDesc& d) {
  if (d.b_planes == nullptr || d.a_cs != 1 || (d.K % W_BK) != 0 || d.batch > 1 || d.a_rowsum != nullptr || d.M < 1) return false;
  auto al4 = [](long x) { return (x & 3) == 0; };
  auto al8 = [](long x) { return (x & 7) == 0; };
  if (!(fd_aligned16(d.A) && al4(d.a_rs) && fd_aligned16(d.b_planes) && al8(d.b_plane_stride))) return false;
  if (d.b_rs == 1) return al8(d.b_cs);                                  // k-contiguous weights
  return d.b_cs == 1 && al8(d.b_rs) && al8(d.N) && d.N >= 8;            // row-contiguous weights (dx = dy W)
}

template <int BM, int BN>
int launch_w(const FdGemmDesc& d, hipStream_t stream) {
  GemmArgs g{};
  g.d = d;
  g.nblk_m = fd_cdiv(d.M, BM);
  g.nblk_n = fd_cdiv(d.N, BN);
  g.ksplit = d.ksplit > 1 ? d.ksplit : 1;
  const int nkt_all = d.K / W_BK;
  if (g.ksplit > nkt_all) g.ksplit = nkt_all > 0 ? nkt_all : 1;
  // (non-temporal stores of a large C -- 105 MB for IPA's projections -- measured: 181 against 115 us; the float4-per-row epilogue
  //  lives on the L2's write combining)
  g.mtiles = 1;
  g.epi_vec = epilogue_vectorisable(d, g.ksplit);
  const bool b_kc = (d.b_rs == 1);
  // which operand an XCD keeps in its L2 while the other streams past: the larger one (weights as 6-byte plane elements against
  // fp32 activations); measured at 3,840 x 6,816 x 256: 122 against 124 us on tile 12, 137 against 150 on tile 13
  static const int raster_env = getenv("FD_GEMM_W_RASTER") ? atoi(getenv("FD_GEMM_W_RASTER")) : -1;       // (A/B measurements)
  g.raster = raster_env >= 0 ? raster_env : ((long)d.N * 6 > (long)d.M * 4 ? 1 : 0);
  dim3 grid(g.nblk_m * g.nblk_n, 1, g.ksplit), block(256, 1, 1);
  if (g.ksplit > 1) {
    if (b_kc) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_w_kernel<BM, BN, true, false>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_w_kernel<BM, BN, false, false>), grid, block, 0, stream, g);
  } else {
    if (b_kc) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_w_kernel<BM, BN, true, true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_w_kernel<BM, BN, false, true>), grid, block, 0, stream, g);
  }
  FD_CHECK_LAUNCH("fd_gemm(pre-split weights)");
  return FD_OK;
}
