// Latency GEMM (fd_gemm tile code 5) for the node-level launches of sampling: M = B*N rows is 128..1024, so the
// 64x64 kernel runs 10..80 blocks and each of them walks the whole K range alone (12.5 us per launch at M = 128,
// 60 % of a sampling forward).  Included by fd_gemm.hip inside its anonymous namespace.
//
// One block = one 32 x 32 output tile; its four waves SPLIT K (wave w takes the 8-k groups g = w mod 4), so the serial
// MFMA chain is a quarter of K.  Operand fragments go global -> registers directly in MFMA layout (no LDS, no
// barrier in the loop): lane (i = l & 31, h = l >> 5) loads the float4 A[i][8g+4h .. +3] (k-contiguous operand) or
// four coalesced dwords (row-contiguous operand); four v_mfma_f32_32x32x2_f32 consume it.  DPF groups are kept in
// flight.  The four partial tiles meet in LDS and every thread finishes one float4 of the tile with the full fused
// epilogue.  fp32 MFMA: exact fp32 products and sums (the k order differs from the 64x64 kernel's).
// gemm_direct16_kernel (round 5): the same kernel on 16 x 16 tiles (v_mfma_f32_16x16x4_f32, 16-k groups) for LONG-K products with few
// tiles.  A 32 x 32 tile of IPA's linear_out on a lone backbone (128 x 256 x 2688: 32 tiles on 256 CUs) is 5.5 MFLOP on one CU's fp32
// matrix pipe (256 flop/cycle: 9 us) behind a load path in which every float4 instruction touches 32 cache lines -- 22 us per launch,
// four per forward, and neither a K split over blocks (device-scope fences) nor more waves per tile (same pipe) helped
// (profiles/r05_gemm_direct_splitk_experiment.txt).  A quarter of the tile per block puts four times as many CUs on the product.
constexpr int DG = 8;   // k per group
constexpr int DPF = 8;  // groups in flight per wave

template <bool KC>
__device__ __forceinline__ void direct_load(float (&v)[4], const float* __restrict__ p, long kstride) {
  if (KC) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = p[t * kstride];
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_direct_kernel(GemmArgs g) {
  __shared__ float part[4][32][33];
  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int bm = (int)blockIdx.x / g.nblk_n, bn = (int)blockIdx.x % g.nblk_n;
  const int m0 = bm * 32, n0 = bn * 32;
  const int z = (int)blockIdx.y;
  const int zo = z / d.bdiv, zi = z % d.bdiv;
  const float* __restrict__ A = d.A + zo * d.a_so + zi * d.a_si;
  const float* __restrict__ B = d.B + zo * d.b_so + zi * d.b_si;
  float* __restrict__ C = d.C + zo * d.c_so + zi * d.c_si;

  // rows / columns past the end are clamped: their results are never stored
  const int ra = (m0 + l31 < d.M) ? m0 + l31 : d.M - 1;
  const int rb = (n0 + l31 < d.N) ? n0 + l31 : d.N - 1;
  const float* pa = A + (long)ra * d.a_rs + (long)(4 * h) * d.a_cs;
  const float* pb = B + (long)rb * d.b_cs + (long)(4 * h) * d.b_rs;
  const long ag = (long)DG * d.a_cs, bg = (long)DG * d.b_rs;   // pointer step per k group

  // epilogue operands of this thread's four outputs (thread -> row tid >> 3, columns 4 (tid & 7) ..): fetched HERE, in front of
  // the operand loads, so that their round trip runs under the K loop instead of after the reduction barrier (a launch of this
  // kernel is a chain of dependent latencies; this removes one of them)
  const int erow = tid >> 3, ec4 = tid & 7;
  const int em = m0 + erow;
  const bool erow_ok = em < d.M;
  float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_res[4] = {0.f, 0.f, 0.f, 0.f}, e_old[4] = {0.f, 0.f, 0.f, 0.f};
  float e_gate[4] = {1.f, 1.f, 1.f, 1.f};
  float e_rs = 1.f;
  if (erow_ok) {
    if (d.rowscale) e_rs = d.rowscale[em];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + 4 * ec4 + e;
      if (n < d.N) {
        if (d.bias) e_bias[e] = d.bias[n];
        if (d.gate) e_gate[e] = d.gate[(long)em * d.ld_gate + n];
        if (d.resid) e_res[e] = d.resid[(long)em * d.ld_resid + n];
        if (d.beta) e_old[e] = C[(long)em * d.ldc + n];
      }
    }
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ngroups = d.K / DG;   // K % 8 == 0 (checked by the host)
  // This wave's groups are g = wave, wave + 4, ...: nmine of them.  DPF groups (one float4 of A and of B each) are kept in
  // flight in a register ring: a launch of this kernel is a chain of dependent latencies (operand fetch -> MFMA chain ->
  // reduction -> store), and with two groups in flight a K = 320 product paid five memory round trips per wave (8.5 us per
  // launch, 45 % of a sampling forward at N = 128); with eight its ten groups are one round trip and a quarter.
  const int nmine = (ngroups - wave + 3) >> 2;
  if (nmine > 0) {
    // (loads are unconditional with the group index clamped to the wave's last group -- a few redundant L1 hits at the end
    // instead of branches around loads: the count of loads in flight is then static and every wait is an exact vmcnt(n))
    const int lastg = nmine - 1;
    float av[DPF][4], bv[DPF][4];
#pragma unroll
    for (int u = 0; u < DPF; ++u) {
      const int gq = wave + 4 * (u < lastg ? u : lastg);
      direct_load<A_KC>(av[u], pa + (long)gq * ag, d.a_cs);
      direct_load<B_KC>(bv[u], pb + (long)gq * bg, d.b_rs);
    }
    int t0 = 0;
    for (; t0 + DPF <= nmine; t0 += DPF) {
#pragma unroll
      for (int u = 0; u < DPF; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fd::mfma_32x32x2(av[u][e], bv[u][e], acc);
        const int tn = t0 + u + DPF;
        const int gq = wave + 4 * (tn < lastg ? tn : lastg);
        direct_load<A_KC>(av[u], pa + (long)gq * ag, d.a_cs);
        direct_load<B_KC>(bv[u], pb + (long)gq * bg, d.b_rs);
      }
    }
    const int rem = nmine - t0;     // < DPF groups left, already in the ring
#pragma unroll
    for (int u = 0; u < DPF - 1; ++u)
      if (u < rem) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fd::mfma_32x32x2(av[u][e], bv[u][e], acc);
      }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
  __syncthreads();

  // thread -> (row, 4 consecutive columns) of the 32 x 32 tile
  const int row = erow, c4 = ec4;
  const int m = em;
  if (!erow_ok) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = 4 * c4 + e;
    const int n = n0 + col;
    if (n >= d.N) continue;
    float x = d.alpha * ((part[0][row][col] + part[1][row][col]) + (part[2][row][col] + part[3][row][col]));
    if (d.bias) x += e_bias[e];
    if (d.relu) x = x > 0.f ? x : 0.f;
    if (d.gate) x = e_gate[e] > 0.f ? x : 0.f;
    x *= e_rs;
    if (d.resid) x += e_res[e];
    if (d.beta) x += e_old[e];
    C[(long)m * d.ldc + n] = x;
  }
}


template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_direct16_kernel(GemmArgs g) {
  constexpr int NW = 4;                      // waves sharing the tile's K range (eight measured the same or slower: 12.6 / 9.1 / 7.7 us
                                             // against 12.6 / 7.4 / 6.0 at 128 x 256 x 2688, 128 x 320 x 1280, 128 x 320 x 960)
  constexpr int G16 = 16;                    // k per group: lane (i = l & 15, q = l >> 4) holds k = 16 grp + 4 q .. + 3 of row i
  __shared__ float part[NW][16][17];
  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, l15 = lane & 15;
  const int bm = (int)blockIdx.x / g.nblk_n, bn = (int)blockIdx.x % g.nblk_n;
  const int m0 = bm * 16, n0 = bn * 16;
  const int z = (int)blockIdx.y;
  const int zo = z / d.bdiv, zi = z % d.bdiv;
  const float* __restrict__ A = d.A + zo * d.a_so + zi * d.a_si;
  const float* __restrict__ B = d.B + zo * d.b_so + zi * d.b_si;
  float* __restrict__ C = d.C + zo * d.c_so + zi * d.c_si;
  const int ra = (m0 + l15 < d.M) ? m0 + l15 : d.M - 1;
  const int rb = (n0 + l15 < d.N) ? n0 + l15 : d.N - 1;
  const float* pa = A + (long)ra * d.a_rs + (long)(4 * q) * d.a_cs;
  const float* pb = B + (long)rb * d.b_cs + (long)(4 * q) * d.b_rs;
  const long ag = (long)G16 * d.a_cs, bg = (long)G16 * d.b_rs;

  // epilogue operands of this thread's output (row tid >> 4, column tid & 15), requested in front of the K loop
  const int erow = tid >> 4, ecol = tid & 15;
  const int em = m0 + erow, en = n0 + ecol;
  const bool eok = em < d.M && en < d.N;
  float e_bias = 0.f, e_res = 0.f, e_old = 0.f, e_gate = 1.f, e_rs = 1.f;
  if (eok) {
    if (d.rowscale) e_rs = d.rowscale[em];
    if (d.bias) e_bias = d.bias[en];
    if (d.gate) e_gate = d.gate[(long)em * d.ld_gate + en];
    if (d.resid) e_res = d.resid[(long)em * d.ld_resid + en];
    if (d.beta) e_old = C[(long)em * d.ldc + en];
  }

  f32x4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = 0.f;
  const int ngroups = d.K / G16;             // K % 16 == 0 (checked by the host)
  const int nmine = wave < ngroups ? (ngroups - wave + NW - 1) / NW : 0;
  if (nmine > 0) {
    const int lastg = nmine - 1;
    float av[DPF][4], bv[DPF][4];
#pragma unroll
    for (int u = 0; u < DPF; ++u) {
      const int gq = wave + NW * (u < lastg ? u : lastg);
      direct_load<A_KC>(av[u], pa + (long)gq * ag, d.a_cs);
      direct_load<B_KC>(bv[u], pb + (long)gq * bg, d.b_rs);
    }
    int t0 = 0;
    for (; t0 + DPF <= nmine; t0 += DPF) {
#pragma unroll
      for (int u = 0; u < DPF; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fd::mfma_16x16x4(av[u][e], bv[u][e], acc);
        const int tn = t0 + u + DPF;
        const int gq = wave + NW * (tn < lastg ? tn : lastg);
        direct_load<A_KC>(av[u], pa + (long)gq * ag, d.a_cs);
        direct_load<B_KC>(bv[u], pb + (long)gq * bg, d.b_rs);
      }
    }
    const int rem = nmine - t0;
#pragma unroll
    for (int u = 0; u < DPF - 1; ++u)
      if (u < rem) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fd::mfma_16x16x4(av[u][e], bv[u][e], acc);
      }
  }
  // D: lane (column l15, g = q): register r <-> row 4 q + r
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][4 * q + r][l15] = acc[r];
  __syncthreads();
  if (!eok) return;
  float x = d.alpha * ((part[0][erow][ecol] + part[1][erow][ecol]) + (part[2][erow][ecol] + part[3][erow][ecol]));
  if (d.bias) x += e_bias;
  if (d.relu) x = x > 0.f ? x : 0.f;
  if (d.gate) x = e_gate > 0.f ? x : 0.f;
  x *= e_rs;
  if (d.resid) x += e_res;
  if (d.beta) x += e_old;
  C[(long)em * d.ldc + en] = x;
}

bool direct_ok(const FdGemmDesc& d) {
  // unit-stride operands with 16-byte aligned k groups where they are read as float4; no pair epilogue, no split-K
  if (d.K <= 0 || (d.K % DG) != 0 || d.pair_p || d.ksplit > 1 || d.a_rowsum) return false;
  auto al4 = [](long x) { return (x & 3) == 0; };
  const bool a_kc = (d.a_cs == 1), b_kc = (d.b_rs == 1);
  if (a_kc) { if (!(fd_aligned16(d.A) && al4(d.a_rs) && al4(d.a_so) && al4(d.a_si))) return false; }
  else if (d.a_rs != 1) return false;
  if (b_kc) { if (!(fd_aligned16(d.B) && al4(d.b_cs) && al4(d.b_so) && al4(d.b_si))) return false; }
  else if (d.b_cs != 1) return false;
  return true;
}

int launch_direct(const FdGemmDesc& d, hipStream_t stream) {
  GemmArgs g{};
  g.d = d;
  g.nblk_m = fd_cdiv(d.M, 32);
  g.nblk_n = fd_cdiv(d.N, 32);
  g.ksplit = 1;
  g.mtiles = 1;
  g.epi_vec = 0;
  const int nb = d.batch > 0 ? d.batch : 1;
  const bool a_kc = (d.a_cs == 1), b_kc = (d.b_rs == 1);
  // 16 x 16 tiles for a long K on few tiles (see the top; FD_GEMM_DIRECT_T16 = 0 / 1 forces it off / on where it applies: measurements)
  static const int t16_env = getenv("FD_GEMM_DIRECT_T16") ? atoi(getenv("FD_GEMM_DIRECT_T16")) : -1;
  const long tiles32 = (long)g.nblk_m * g.nblk_n * nb;
  const bool t16_ok = (d.K % 16) == 0;
  // (measured, profiles/r05_gemm_direct_t16.txt: wins for K >= 640 up to 40 tiles of 32 x 32 -- M = 128 --, for K >= 960 up to 80 -- M = 256)
  const bool t16 = t16_ok && (t16_env >= 0 ? t16_env != 0 : ((d.K >= 640 && tiles32 <= 40) || (d.K >= 960 && tiles32 <= 80)));
  if (t16) {
    g.nblk_m = fd_cdiv(d.M, 16);
    g.nblk_n = fd_cdiv(d.N, 16);
    dim3 grid16(g.nblk_m * g.nblk_n, nb, 1), block16(256, 1, 1);
    if (a_kc && b_kc)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct16_kernel<true, true>), grid16, block16, 0, stream, g);
    else if (a_kc && !b_kc)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct16_kernel<true, false>), grid16, block16, 0, stream, g);
    else if (!a_kc && b_kc)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct16_kernel<false, true>), grid16, block16, 0, stream, g);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct16_kernel<false, false>), grid16, block16, 0, stream, g);
    FD_CHECK_LAUNCH("fd_gemm(direct, 16 x 16)");
    return FD_OK;
  }
  dim3 grid(g.nblk_m * g.nblk_n, nb, 1), block(256, 1, 1);
  if (a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct_kernel<true, true>), grid, block, 0, stream, g);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct_kernel<true, false>), grid, block, 0, stream, g);
  else if (!a_kc && b_kc)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct_kernel<false, true>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_direct_kernel<false, false>), grid, block, 0, stream, g);
  FD_CHECK_LAUNCH("fd_gemm(direct)");
  return FD_OK;
}
