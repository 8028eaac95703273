// 64 x 64 split-bf16 GEMM kernel (fd_gemm tile code 10) for the node-level and attention GEMMs: M = B*N rows of a few
// thousand, N_out / K of a few hundred, 100..1000 blocks per launch.  Included by fd_gemm.hip inside its anonymous
// namespace (uses GemmArgs, store_tile, store_tile_vec).
//
// These launches ran on the 64 x 64 fp32-MFMA kernel (tile 2): one 32 x 32 wave tile costs 16 v_mfma_f32_32x32x2_f32
// = 1024 matrix-pipe cycles per 32 k.  Here the same wave tile costs 12 v_mfma_f32_32x32x16_bf16 = 384 cycles (fp32
// operands as three exact bf16 planes, six products per k-step: fp32-accurate, the arithmetic of tile 4), so a launch
// whose CUs hold one or two blocks each is no longer bound by its serial MFMA chain.
//   * every thread stages AND multiplies (4 waves, 2 x 2 wave tiles of 32 x 32); two-stage LDS ring of 32-k stages, one
//     barrier per stage, global loads two stages ahead in registers;
//   * operands are loaded as float4s along whichever index is contiguous in memory, split into planes in registers and
//     written with ds_write_b64.  A k-contiguous operand is read back with ds_read_b128 ([row][16 k] images, the row
//     format of tile 4); a row-contiguous one ([k][32 columns] images) with ds_read_b64_tr_b16, the LDS transpose
//     read -- all four layout combinations share the code;
//   * epilogues, batching and split-K are those of the fp32 kernel (store_tile / store_tile_vec); the fused bias
//     gradient (a_rowsum) is taken from the staged A operand when it is row-contiguous (dW = dY^T X).
constexpr int S_BK = 32;
constexpr int S_NSET = 4;                           // register sets = stages of global loads in flight
constexpr int S_KC_ROWB = 112;                        // three 32-byte planes of 16 k + 16 B pad
constexpr int S_KC_SUB = 64 * S_KC_ROWB + 32;         // one 16-k half of a stage (+32: the halves start on different banks)
constexpr int S_TR_PST = S_BK * 64 + 64;              // a 32-column panel of one plane: [32 k][64 B]
constexpr int S_TR_PLANE = 2 * S_TR_PST;
constexpr int S_OP_BYTES = 2 * S_KC_SUB;              // 14,400 B >= 3 * S_TR_PLANE = 12,672 B
constexpr int S_STAGE = 2 * S_OP_BYTES;
constexpr int S_LDS = 2 * S_STAGE;                    // 57,600 B: two blocks per CU
static_assert(3 * S_TR_PLANE <= S_OP_BYTES, "operand image");
static_assert(64 * (64 + 4) * 4 <= S_LDS, "epilogue scratch");

__device__ __forceinline__ void s64_split4(const float4 v, uint2& s0, uint2& s1, uint2& s2) {
  const unsigned h0 = fd::pack_bf16(v.x, v.y), h1 = fd::pack_bf16(v.z, v.w);
  const float r0 = v.x - fd::bf16lo_f32(h0), r1 = v.y - fd::bf16hi_f32(h0);
  const float r2 = v.z - fd::bf16lo_f32(h1), r3 = v.w - fd::bf16hi_f32(h1);
  const unsigned m0 = fd::pack_bf16(r0, r1), m1 = fd::pack_bf16(r2, r3);
  const float q0 = r0 - fd::bf16lo_f32(m0), q1 = r1 - fd::bf16hi_f32(m0);
  const float q2 = r2 - fd::bf16lo_f32(m1), q3 = r3 - fd::bf16hi_f32(m1);
  s0 = make_uint2(h0, h1);
  s1 = make_uint2(m0, m1);
  s2 = make_uint2(fd::pack_bf16(q0, q1), fd::pack_bf16(q2, q3));
}

// One 64-row operand of a 32-k stage.  `rs` = stride of the row index (m of A, n of B), `ks` = stride of k.
template <bool KC>
struct S64Stager {
  const float* p[2];   // per-slot source of the current stage (row / column clamped into range)
  int kofs[2];         // k of the slot inside the stage (first of four when KC)
  int wofs[2];         // LDS byte offset of the slot in plane 0
  long kstep;

  __device__ __forceinline__ void init(const float* __restrict__ b, long rs, long ks, int row0, int nrows, int k0, int tid) {
    kstep = (long)S_BK * ks;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      if (KC) {
        const int row = f >> 3, kq = f & 7;
        const int gr = (row0 + row < nrows) ? row0 + row : nrows - 1;
        kofs[i] = 4 * kq;
        p[i] = b + (long)gr * rs + (long)(k0 + 4 * kq);
        wofs[i] = (kq >> 2) * S_KC_SUB + row * S_KC_ROWB + (kq & 3) * 8;
      } else {
        const int k = f >> 4, c4 = f & 15;
        // (nrows % 4 == 0: a float4 of rows is entirely inside or outside; outside -> the last one inside)
        const int gr = (row0 + 4 * c4 < nrows) ? row0 + 4 * c4 : nrows - 4;
        kofs[i] = k;
        p[i] = b + (long)gr * rs + (long)(k0 + k) * ks;
        wofs[i] = (c4 >> 3) * S_TR_PST + k * 64 + (c4 & 7) * 8;
      }
    }
  }
  // stage starting at k0 of a k range ending at kend: slots past the end read the range's last valid k and become zeros
  __device__ __forceinline__ void load(float4 (&r)[2], int k0, int kend, long ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int over = k0 + kofs[i] - (kend - (KC ? 4 : 1));   // > 0: past the end by `over`
      const bool ok = over <= 0;
      const float4 v = *reinterpret_cast<const float4*>(ok ? p[i] : p[i] - (long)over * (KC ? 1 : ks));
      r[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
      p[i] += kstep;
    }
  }
  __device__ __forceinline__ void store(const float4 (&r)[2], char* __restrict__ img) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint2 s0, s1, s2;
      s64_split4(r[i], s0, s1, s2);
      constexpr int PL = KC ? 32 : S_TR_PLANE;
      *reinterpret_cast<uint2*>(img + wofs[i]) = s0;
      *reinterpret_cast<uint2*>(img + wofs[i] + PL) = s1;
      *reinterpret_cast<uint2*>(img + wofs[i] + 2 * PL) = s2;
    }
  }
};

// the 32-row fragment `w` (0 / 1) of an operand image for 16-k step s: one uint4 (8 consecutive k) per plane
template <bool KC>
__device__ __forceinline__ void s64_frag(uint4 (&f)[3], const char* img, int w, int s, int lane) {
  if (KC) {
    const char* q = img + s * S_KC_SUB + (w * 32 + (lane & 31)) * S_KC_ROWB + (lane >> 5) * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const uint4*>(q + 32 * pl);
  } else {
    const int i16 = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5;
    const char* q = img + w * S_TR_PST + (16 * s + 8 * kg + (i16 >> 2)) * 64 + half * 32 + (i16 & 3) * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const uint2 lo = fd::lds_read_tr16(q + pl * S_TR_PLANE), hi = fd::lds_read_tr16(q + pl * S_TR_PLANE + 256);
      f[pl] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_s64_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[S_LDS];
  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int nblk = g.nblk_m * g.nblk_n;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, nblk);
  const int bm = lid / g.nblk_n, bn = lid % g.nblk_n;
  const int m0 = bm * 64, n0 = bn * 64;
  const int z = (int)blockIdx.y;
  const int zo = z / d.bdiv, zi = z % d.bdiv;
  const float* __restrict__ A = d.A + zo * d.a_so + zi * d.a_si;
  const float* __restrict__ B = d.B + zo * d.b_so + zi * d.b_si;
  float* __restrict__ C = d.C + zo * d.c_so + zi * d.c_si;

  // split-K: blockIdx.z owns stages [kt0, kt0 + nk)
  const int nkt_all = (d.K + S_BK - 1) / S_BK;
  const int per = (nkt_all + g.ksplit - 1) / g.ksplit;
  const int kt0 = (int)blockIdx.z * per;
  const int nkt = (kt0 + per < nkt_all) ? kt0 + per : nkt_all;
  const int nk = nkt - kt0;
  if (nk <= 0) return;
  const int kbeg = kt0 * S_BK;
  const int kend = (nkt * S_BK < d.K) ? nkt * S_BK : d.K;

  S64Stager<A_KC> sa;
  S64Stager<B_KC> sb;
  sa.init(A, d.a_rs, d.a_cs, m0, d.M, kbeg, tid);
  sb.init(B, d.b_cs, d.b_rs, n0, d.N, kbeg, tid);   // the staged "row" of B is n
  float4 ra[S_NSET][2], rb[S_NSET][2];
  // fused bias gradient of dW = dY^T X: row sums of the row-contiguous A over k, from the registers on their way to LDS
  const bool do_rowsum = !A_KC && d.a_rowsum != nullptr && bn == 0 && z == 0;
  float rsum[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) rsum[i][e] = 0.f;

  auto load = [&](float4 (&xa)[2], float4 (&xb)[2], int st) __attribute__((always_inline)) {
    sa.load(xa, kbeg + st * S_BK, kend, d.a_cs);
    sb.load(xb, kbeg + st * S_BK, kend, d.b_rs);
  };
  auto put = [&](float4 (&xa)[2], float4 (&xb)[2], char* dst) __attribute__((always_inline)) {
    if (!A_KC && do_rowsum) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { rsum[i][0] += xa[i].x; rsum[i][1] += xa[i].y; rsum[i][2] += xa[i].z; rsum[i][3] += xa[i].w; }
    }
    sa.store(xa, dst);
    sb.store(xb, dst + S_OP_BYTES);
  };
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  auto mma = [&](const char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 fa[3], fb[3];
      s64_frag<A_KC>(fa, st, wm, s, lane);
      s64_frag<B_KC>(fb, st + S_OP_BYTES, wn, s, lane);
      // the six products with i + j <= 2, small terms first
      acc[0][0] = fd::mfma_32x32x16_bf16(fa[2], fb[0], acc[0][0]);
      acc[0][0] = fd::mfma_32x32x16_bf16(fa[1], fb[1], acc[0][0]);
      acc[0][0] = fd::mfma_32x32x16_bf16(fa[0], fb[2], acc[0][0]);
      acc[0][0] = fd::mfma_32x32x16_bf16(fa[1], fb[0], acc[0][0]);
      acc[0][0] = fd::mfma_32x32x16_bf16(fa[0], fb[1], acc[0][0]);
      acc[0][0] = fd::mfma_32x32x16_bf16(fa[0], fb[0], acc[0][0]);
    }
  };

  // stage s lives in register set s % S_NSET (loaded S_NSET stages ahead: these launches have 5..120 stages of ~0.2 us
  // of MFMA work each, so the global-load latency has to be covered by depth, not by the stage time) and ring slot
  // s & 1; stages past the end of the k range are zeros (see S64Stager::load), so the loop body is branch-free
#pragma unroll
  for (int u = 0; u < S_NSET; ++u) load(ra[u], rb[u], u);
  put(ra[0], rb[0], lds);
  load(ra[0], rb[0], S_NSET);
  __syncthreads();
  auto step = [&](int s, float4 (&xa)[2], float4 (&xb)[2]) __attribute__((always_inline)) {
    // xa / xb: the register set of stage s + 1, refilled with stage s + 1 + S_NSET
    mma(lds + (s & 1) * S_STAGE);
    put(xa, xb, lds + ((s + 1) & 1) * S_STAGE);
    load(xa, xb, s + 1 + S_NSET);
    __syncthreads();
  };
  for (int s = 0; s < nk; s += S_NSET) {
#pragma unroll
    for (int u = 0; u < S_NSET; ++u) {
      if (s + u >= nk) break;
      step(s + u, ra[(u + 1) % S_NSET], rb[(u + 1) % S_NSET]);
    }
  }

  if (!A_KC && d.a_rowsum != nullptr) {   // (block-uniform)
    // 16 threads x 2 slots hold partial sums of the same four rows: meet in LDS, one global atomic per row
    float* rs_lds = reinterpret_cast<float*>(lds);
    if (tid < 64) rs_lds[tid] = 0.f;
    __syncthreads();
    if (do_rowsum) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = (tid + 256 * i) & 15;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(rs_lds + 4 * c4 + e, rsum[i][e]);
      }
    }
    __syncthreads();
    if (do_rowsum && tid < 64 && m0 + tid < d.M) atomicAdd(d.a_rowsum + m0 + tid, d.alpha * rs_lds[tid]);
    __syncthreads();
  }
  if (g.epi_vec)
    store_tile_vec<64, 64, 1, 1>(d, C, acc, reinterpret_cast<float*>(lds), m0, n0, wm, wn, h, l31, tid);
  else
    store_tile<1, 1>(d, C, acc, m0 + wm * 32, n0 + wn * 32, h, l31, g.ksplit > 1);
}

int launch_s64(const FdGemmDesc& d, hipStream_t stream) {
  GemmArgs g{};
  g.d = d;
  g.nblk_m = fd_cdiv(d.M, 64);
  g.nblk_n = fd_cdiv(d.N, 64);
  const int nb = d.batch > 0 ? d.batch : 1;
  g.ksplit = d.ksplit > 1 ? d.ksplit : 1;
  const int nkt_all = fd_cdiv(d.K, S_BK);
  if (g.ksplit > nkt_all) g.ksplit = nkt_all > 0 ? nkt_all : 1;
  g.mtiles = 1;
  g.epi_vec = epilogue_vectorisable(d, g.ksplit);
  const bool a_kc = (d.a_cs == 1), b_kc = (d.b_rs == 1);
  dim3 grid(g.nblk_m * g.nblk_n, nb, g.ksplit), block(256, 1, 1);
  if (a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_s64_kernel<true, true>), grid, block, 0, stream, g);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_s64_kernel<true, false>), grid, block, 0, stream, g);
  else if (!a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_s64_kernel<false, true>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_s64_kernel<false, false>), grid, block, 0, stream, g);
  FD_CHECK_LAUNCH("fd_gemm(64x64 split-bf16)");
  return FD_OK;
}
