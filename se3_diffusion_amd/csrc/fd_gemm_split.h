// Split-bf16 GEMM kernel (fd_gemm tile codes 4 and 6) for the pair-level GEMMs.  Included by fd_gemm.hip inside its
// anonymous namespace (uses GemmArgs, store_tile, store_tile_vec, fd_xcd_swizzle).
//
// gfx950 runs the bf16 MFMA at 16x the fp32-MFMA rate.  An fp32 value splits EXACTLY into three bf16 terms
// x = x0 + x1 + x2 (round-to-nearest at every stage: 3 x (8 bits + sign of the next residual) covers the 24-bit
// significand), so  a*b = sum_{i,j} a_i*b_j  with every term an exact bf16 product.  The six terms with i+j <= 2
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the three dropped terms are <= 2^-26 |a*b|, below the fp32
// rounding of the product itself.  The result is an fp32-accurate GEMM (same error class as the fmaf-chain kernel;
// tests/test_gemm.py measures both against fp64) at up to 16/6 = 2.7x its peak rate.
//
// Operands stay fp32 in HBM; the split happens in registers on the way global -> LDS (11 VALU ops per element
// pair).  The block is wave-specialised so that this VALU work never sits in the instruction stream of a wave that
// feeds the matrix pipe:
//   * consumer waves (wave tile 64 x 64) issue only ds_read_b128 + MFMA;
//   * producer waves load later stages from global memory, split them and write them to LDS.
// A stage is 16 k (one MFMA step) of every row: three 32-byte bf16 planes + 16 B pad = 112 B per row (7 x 16 B:
// conflict-free ds_read_b128; the producer's row order makes every 8-lane ds_write_b128 group tile one 128-byte
// bank window).
//
// Two block shapes (BM x 128 output tile, BM/64 x 2 consumer waves, BM/64 producer waves):
//   BM = 256 (tile 4): 12 waves, LDS ring of THREE stages = 129 KB -> one block per CU.  While the consumers multiply
//     stage s out of registers they prefetch the fragments of stage s+1 (complete since the previous barrier) and
//     the producers fill stage s+2: the single barrier per stage never has an LDS read waiting behind it.
//   BM = 128 (tile 6): 6 waves, ring of TWO stages, 66 KB -> two blocks per CU, so the prologue and the (HBM-heavy)
//     epilogue of one block run under the MFMA loop of the other; the natural shape for the N_out = 128 weight
//     gradients too.
constexpr int XBK = 16;
constexpr int XROWB = 3 * XBK * 2 + 16;   // bytes per staged row
constexpr int XBN = 128;
constexpr int XBM = 256;                  // the wide shape (tile 4)

template <int BM>
struct SplitCfg {
  static constexpr int NCONS = BM * 2;           // consumer threads: (BM / 64) x 2 waves
  static constexpr int NPROD = BM;               // producer threads: BM / 64 waves
  static constexpr int NTHR = NCONS + NPROD;
  static constexpr int STAGE = (BM + XBN) * XROWB;
  static constexpr int RING = BM == 256 ? 3 : 2;
  static constexpr int EPI = BM * (XBN + 4) * 4;  // the vector epilogue transposes the C tile through LDS
  static constexpr int LDS = RING * STAGE > EPI ? RING * STAGE : EPI;
  static constexpr int BLOCKS_PER_CU = BM == 256 ? 1 : 2;
  static_assert(LDS * BLOCKS_PER_CU <= 160 * 1024, "LDS");
};

// LDS-free epilogue for the TRANSPOSED accumulation (the B fragment is the MFMA's row operand): a lane holds four
// consecutive COLUMNS of one output row per accumulator group, so bias / pair / gate / residual / old C / C move as
// float4s straight between registers and global memory -- no LDS round trip, no barrier.
template <int TM, int TN>
__device__ __forceinline__ void store_tile_t(const FdGemmDesc& d, float* __restrict__ C, f32x16 (&acc)[TM][TN],
                                             int m_base, int n_base, int h, int l31, bool vec) {
  // acc[i][j][r]: row m = m_base + 32 i + l31, column n = n_base + 32 j + 8 (r >> 2) + 4 h + (r & 3)
  const bool has_pair = d.pair_p != nullptr, has_gate = d.gate != nullptr, has_res = d.resid != nullptr;
  const bool has_rs = d.rowscale != nullptr, has_beta = d.beta != 0, has_bias = d.bias != nullptr;
  const int nres = d.nres;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mr = m_base + i * 32 + l31;
    const bool mok = mr < d.M;
    const int m = mok ? mr : d.M - 1;
    const float rs = has_rs ? d.rowscale[m] : 1.f;
    const float* pp = C;
    const float* pq = C;
    if (has_pair) {
      const int q = m / nres;
      const int jj = m - q * nres;
      const int bb = q / nres;
      pp = d.pair_p + (long)q * d.ld_pair;
      pq = d.pair_q + ((long)bb * nres + jj) * d.ld_pair;
    }
    float* crow = C + (long)m * d.ldc;
    const float* grow = has_gate ? d.gate + (long)m * d.ld_gate : C;
    const float* rrow = has_res ? d.resid + (long)m * d.ld_resid : C;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int nr = n_base + j * 32 + 8 * gq + 4 * h;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = d.alpha * acc[i][j][4 * gq + e];
        if (vec) {
          // N % 4 == 0: the float4 is entirely inside or outside
          const bool nok = nr < d.N;
          const int n = nok ? nr : 0;
          if (has_bias) {
            const float4 t = *reinterpret_cast<const float4*>(d.bias + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
          }
          if (has_pair) {
            const float4 a = *reinterpret_cast<const float4*>(pp + n);
            const float4 b = *reinterpret_cast<const float4*>(pq + n);
            v[0] += a.x + b.x; v[1] += a.y + b.y; v[2] += a.z + b.z; v[3] += a.w + b.w;
          }
          if (d.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (has_gate) {
            const float4 t = *reinterpret_cast<const float4*>(grow + n);
            v[0] = t.x > 0.f ? v[0] : 0.f; v[1] = t.y > 0.f ? v[1] : 0.f;
            v[2] = t.z > 0.f ? v[2] : 0.f; v[3] = t.w > 0.f ? v[3] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= rs;
          if (has_res) {
            const float4 t = *reinterpret_cast<const float4*>(rrow + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
          }
          if (has_beta) {
            const float4 t = *reinterpret_cast<const float4*>(crow + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
          }
          if (mok && nok) *reinterpret_cast<float4*>(crow + n) = make_float4(v[0], v[1], v[2], v[3]);
          fd::sched_fence();   // keep the 16 float4 groups from being software-pipelined into one register spike
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ne = nr + e;
            const bool nok = ne < d.N;
            const int n = nok ? ne : 0;
            float x = v[e];
            if (has_bias) x += d.bias[n];
            if (has_pair) x += pp[n] + pq[n];
            if (d.relu) x = x > 0.f ? x : 0.f;
            if (has_gate) x = grow[n] > 0.f ? x : 0.f;
            x *= rs;
            if (has_res) x += rrow[n];
            if (has_beta) x += crow[n];
            if (mok && nok) crow[n] = x;
          }
        }
      }
  }
}

__device__ __forceinline__ void split8(const float (&x)[8], uint4& s0, uint4& s1, uint4& s2) {
  unsigned t0[4], t1[4], t2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = x[2 * j], v = x[2 * j + 1];
    const unsigned h = fd::pack_bf16(u, v);
    const float ru = u - fd::bf16lo_f32(h), rv = v - fd::bf16hi_f32(h);
    const unsigned m = fd::pack_bf16(ru, rv);
    const float qu = ru - fd::bf16lo_f32(m), qv = rv - fd::bf16hi_f32(m);
    t0[j] = h;
    t1[j] = m;
    t2[j] = fd::pack_bf16(qu, qv);
  }
  s0 = make_uint4(t0[0], t0[1], t0[2], t0[3]);
  s1 = make_uint4(t1[0], t1[1], t1[2], t1[3]);
  s2 = make_uint4(t2[0], t2[1], t2[2], t2[3]);
}

// Stages a ROWS x 16 operand tile with NPROD producer threads.  A slot = (row, group of 8 consecutive k).
//   KC : k contiguous in memory   -> 2 x float4 per slot, 2 lanes per row
//   !KC: row contiguous in memory -> 8 dword loads per slot (coalesced across the lanes = rows)
template <int ROWS, bool KC, int NPROD>
struct SplitStager {
  static constexpr int NS = ROWS * 2 / NPROD;
  const float* p[NS];
  int kofs[NS];   // first k of the slot inside the stage
  int lofs[NS];   // LDS byte offset of the slot (plane 0)
  long kstep, cs;

  // slot i of producer thread tid -> (row inside the tile, k group)
  static __device__ __forceinline__ void slot(int tid, int i, int& row, int& kg) {
    const int f = tid + NPROD * i;   // tid = producer thread index
    // KC: two lanes cover the 64 contiguous bytes a row contributes to a stage; the four rows of an 8-lane
    // ds_write_b128 group are taken 2 apart (2 * 112 B = 96 mod 128: the group tiles one 32-bank window)
    const int q = f >> 1;
    row = KC ? ((q & ~7) | ((q & 3) << 1) | ((q & 7) >> 2)) : (f % ROWS);
    kg = KC ? (f & 1) : (f / ROWS);
  }

  // point the slots at rows row0.. of the operand, first k = k0 (the persistent kernel calls this once per tile)
  __device__ __forceinline__ void target(const float* __restrict__ b, long rs_, int row0, int nrows, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      int row, kg;
      slot(tid, i, row, kg);
      // rows past the end are CLAMPED, not zero-filled: an output row/column depends only on its own operand
      // row, and the epilogue never stores rows >= M or columns >= N
      const int grow = (row0 + row < nrows) ? row0 + row : nrows - 1;
      p[i] = b + (long)grow * rs_ + (long)(k0 + 8 * kg) * cs;
    }
  }

  __device__ __forceinline__ void init(const float* __restrict__ b, long rs_, long cs_, int row0, int nrows, int k0,
                                       int tid) {
    cs = cs_;
    kstep = (long)XBK * cs_;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      int row, kg;
      slot(tid, i, row, kg);
      kofs[i] = 8 * kg;
      lofs[i] = row * XROWB + kg * 16;
    }
    target(b, rs_, row0, nrows, k0, tid);
  }

  __device__ __forceinline__ void load(float (&r)[NS][8], int k0, int K) {
    if (k0 + XBK <= K) {   // (uniform) whole stage inside the k range: unconditional loads
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (KC) {
          const float4 v = *reinterpret_cast<const float4*>(p[i]);
          const float4 w = *reinterpret_cast<const float4*>(p[i] + 4);
          r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
          r[i][4] = w.x; r[i][5] = w.y; r[i][6] = w.z; r[i][7] = w.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) r[i][e] = p[i][e * cs];
        }
        p[i] += kstep;
      }
      return;
    }
    // k tail: zeros beyond K
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[i][e] = 0.f;
      if (KC) {
        // K % 4 == 0 (operands_vectorisable): a float4 is entirely inside or outside the k range
        if (k0 + kofs[i] < K) {
          const float4 v = *reinterpret_cast<const float4*>(p[i]);
          r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
        }
        if (k0 + kofs[i] + 4 < K) {
          const float4 v = *reinterpret_cast<const float4*>(p[i] + 4);
          r[i][4] = v.x; r[i][5] = v.y; r[i][6] = v.z; r[i][7] = v.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + kofs[i] + e < K) r[i][e] = p[i][e * cs];
      }
      p[i] += kstep;
    }
  }

  __device__ __forceinline__ void store(const float (&r)[NS][8], char* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      uint4 s0, s1, s2;
      split8(r[i], s0, s1, s2);
      *reinterpret_cast<uint4*>(lds + lofs[i]) = s0;
      *reinterpret_cast<uint4*>(lds + lofs[i] + 2 * XBK) = s1;
      *reinterpret_cast<uint4*>(lds + lofs[i] + 4 * XBK) = s2;
    }
  }
};

// TRANS: transposed accumulation + LDS-free float4 epilogue (needs ksplit == 1: its split-K atomics would touch one
// cache line per lane); !TRANS: row-major accumulation, LDS-transposed epilogue or split-K atomics.
template <int BM, bool A_KC, bool B_KC, bool TRANS>
__global__ __launch_bounds__(SplitCfg<BM>::NTHR, SplitCfg<BM>::BLOCKS_PER_CU) void gemm_bx3_kernel(GemmArgs g) {
  using Cfg = SplitCfg<BM>;
  constexpr int BN = XBN, TM = 2, TN = 2;
  constexpr int NCONS = Cfg::NCONS, NPROD = Cfg::NPROD, STAGE = Cfg::STAGE, RING = Cfg::RING;
  constexpr int A_BYTES = BM * XROWB;
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS];

  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;

  const int nblk = g.nblk_m * g.nblk_n;
  const int lid = fd_xcd_swizzle((int)blockIdx.x, nblk);
  const int bm = lid / g.nblk_n, bn = lid % g.nblk_n;
  const int m0 = bm * BM, n0 = bn * BN;

  const int z = (int)blockIdx.y;
  const int zo = z / d.bdiv, zi = z % d.bdiv;

  // split-K: blockIdx.z owns stages [kt0, kt0 + nk)
  const int nkt_all = (d.K + XBK - 1) / XBK;
  const int per = (nkt_all + g.ksplit - 1) / g.ksplit;
  const int kt0 = (int)blockIdx.z * per;
  const int nkt = (kt0 + per < nkt_all) ? kt0 + per : nkt_all;
  const int nk = nkt - kt0;
  if (nk <= 0) return;

  if (tid >= NCONS) {
    // ---- producer waves: global -> registers -> (split) -> LDS ring ----
    const int ptid = tid - NCONS;
    fd::raise_wave_priority();   // the producer's instruction stream must never wait for an issue slot
    const float* __restrict__ A = d.A + zo * d.a_so + zi * d.a_si;
    const float* __restrict__ B = d.B + zo * d.b_so + zi * d.b_si;
    SplitStager<BM, A_KC, NPROD> sa;
    SplitStager<BN, B_KC, NPROD> sb;
    constexpr int NSA = SplitStager<BM, A_KC, NPROD>::NS, NSB = SplitStager<BN, B_KC, NPROD>::NS;
    // NSET register sets = NSET stages of global loads in flight per producer thread (stage s lives in set s % NSET).
    // (Four sets were measured on the HBM-streamed pair tensor: no change against two -- 1.06 ms on the K = 384
    // edge shape either way -- so the load latency is not what separates it from an L2-resident operand.)
    constexpr int NSET = 2;
    float ra[NSET][NSA][8], rb[NSET][NSB][8];
    sa.init(A, d.a_rs, d.a_cs, m0, d.M, kt0 * XBK, ptid);
    sb.init(B, d.b_cs, d.b_rs, n0, d.N, kt0 * XBK, ptid);   // the staged "row" of B is n
    int lk = kt0;   // stage of the next global load
    auto issue = [&](float (&xa)[NSA][8], float (&xb)[NSB][8]) {
      sa.load(xa, lk * XBK, d.K);
      sb.load(xb, lk * XBK, d.K);
      ++lk;
    };
    // fused bias gradient of dW = dY^T X (A = dY^T row-contiguous: producer thread t stages row m0 + t in all of
    // its slots, NPROD == BM): row sums of A over k, taken from the registers on their way to LDS
    const bool do_rowsum = !A_KC && d.a_rowsum != nullptr && bn == 0 && z == 0;
    float rsum = 0.f;
    int wbuf = 0;   // ring slot of the next LDS store
    auto put = [&](float (&xa)[NSA][8], float (&xb)[NSB][8]) {
      if (!A_KC && do_rowsum) {
#pragma unroll
        for (int i = 0; i < NSA; ++i)
          rsum += ((xa[i][0] + xa[i][1]) + (xa[i][2] + xa[i][3])) + ((xa[i][4] + xa[i][5]) + (xa[i][6] + xa[i][7]));
      }
      char* dst = lds + wbuf * STAGE;
      sa.store(xa, dst);
      sb.store(xb, dst + A_BYTES);
      wbuf = (wbuf == RING - 1) ? 0 : wbuf + 1;
    };
    // AHEAD = how many stages beyond the one being multiplied are complete in LDS at a barrier
    constexpr int AHEAD = RING - 1;
#pragma unroll
    for (int u = 0; u < NSET; ++u)
      if (u < nk) issue(ra[u], rb[u]);
#pragma unroll
    for (int u = 0; u < AHEAD; ++u)
      if (u < nk) {
        put(ra[u], rb[u]);
        if (u + NSET < nk) issue(ra[u], rb[u]);
      }
    __syncthreads();   // the first AHEAD stages are in the ring
    // during stage `it`: stage it+AHEAD goes registers -> its ring slot (last read AHEAD barriers ago) and the loads
    // of stage it+AHEAD+NSET refill the register set
    auto step = [&](int it, float (&xa)[NSA][8], float (&xb)[NSB][8]) {
      if (it + AHEAD < nk) {
        put(xa, xb);
        if (it + AHEAD + NSET < nk) issue(xa, xb);
      }
      __syncthreads();
    };
    for (int it = 0; it < nk; it += NSET) {
#pragma unroll
      for (int u = 0; u < NSET; ++u)
        if (it + u < nk) step(it + u, ra[(AHEAD + u) % NSET], rb[(AHEAD + u) % NSET]);
    }
    if (!A_KC && do_rowsum && m0 + ptid < d.M) atomicAdd(d.a_rowsum + m0 + ptid, d.alpha * rsum);
    if (!TRANS && g.epi_vec) {   // the two barriers of the consumers' LDS-transposed epilogue
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  // ---- consumer waves: LDS -> registers -> MFMA ----
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  float* __restrict__ C = d.C + zo * d.c_so + zi * d.c_si;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane fragment base: row (l31); lane-half h takes k 8h..8h+7 of the 16-k MFMA step
  const int a_frag = ((wm * TM) * 32 + l31) * XROWB + h * 16;
  const int b_frag = A_BYTES + ((wn * TN) * 32 + l31) * XROWB + h * 16;
  uint4 fa[2][TM][3], fb[2][TN][3];
  int rbuf = 0;   // ring slot of the next fragment read
  auto read_frags = [&](uint4 (&xa)[TM][3], uint4 (&xb)[TN][3]) {
    const char* st = lds + rbuf * STAGE;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        xa[i][s] = *reinterpret_cast<const uint4*>(st + a_frag + i * 32 * XROWB + s * 2 * XBK);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        xb[j][s] = *reinterpret_cast<const uint4*>(st + b_frag + j * 32 * XROWB + s * 2 * XBK);
    }
    rbuf = (rbuf == RING - 1) ? 0 : rbuf + 1;
  };
  // term pairs (i, j) with i + j <= 2, smallest first; the four accumulators are interleaved so that dependent
  // MFMAs are four issues apart
  auto mfmas = [&](uint4 (&xa)[TM][3], uint4 (&xb)[TN][3]) {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
      constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = TRANS ? fd::mfma_32x32x16_bf16(xb[j][PB[p]], xa[i][PA[p]], acc[i][j])
                            : fd::mfma_32x32x16_bf16(xa[i][PA[p]], xb[j][PB[p]], acc[i][j]);
    }
  };

  __syncthreads();   // the first stages are in the ring
  if (RING == 3) {
    // stage it is multiplied out of registers while the fragments of stage it+1 (complete in the ring since the
    // last barrier) are fetched; the barrier itself carries no LDS wait on this side
    auto step = [&](int it, uint4 (&ca)[TM][3], uint4 (&cb)[TN][3], uint4 (&na)[TM][3], uint4 (&nb)[TN][3]) {
      if (it + 1 < nk) read_frags(na, nb);
      mfmas(ca, cb);
      fd::block_barrier_nofence();
    };
    read_frags(fa[0], fb[0]);
    for (int it = 0; it < nk; it += 2) {
      step(it, fa[0], fb[0], fa[1], fb[1]);
      if (it + 1 < nk) step(it + 1, fa[1], fb[1], fa[0], fb[0]);
    }
  } else {
    // two-stage ring: stage it+1 is only complete at the barrier that ends stage it, so its fragments are fetched
    // after the barrier; the co-resident second block covers that latency
    for (int it = 0; it < nk; ++it) {
      read_frags(fa[0], fb[0]);
      mfmas(fa[0], fb[0]);
      fd::block_barrier_nofence();
    }
  }

  if (TRANS)
    store_tile_t<TM, TN>(d, C, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, h, l31, g.epi_vec != 0);
  else if (g.epi_vec)
    store_tile_vec<BM, BN, TM, TN, NCONS>(d, C, acc, reinterpret_cast<float*>(lds), m0, n0, wm, wn, h, l31, tid);
  else
    store_tile<TM, TN>(d, C, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, h, l31, g.ksplit > 1);
}


// Persistent form of the 256 x 128 TRANS kernel (A k-contiguous, no split-K): gridDim.x = one block per CU, block b
// walks the tiles b, b + gridDim.x, ...  The producers treat the (tile, k stage) sequence as ONE stream, so while the
// consumers store a finished tile the producers already have the next tile's first stages in the ring (and two more
// in registers): the global-load latency of a tile's prologue, the block launch and the descriptor setup no longer
// sit between two MFMA loops.  Measured on the fresh-block kernel: 10.5 us of un-overlapped prologue + epilogue per tile
// against 29 us of MFMA loop at K = 384 (52 % of a K = 128 launch); persistent: 9.3 us.  What remains is the epilogue
// itself (no epilogue: ~0; the same bytes as lane-contiguous 1 KB stores: -2.7 us): a store instruction of the
// transposed layout touches 32 different lines (a lane owns a row) and the CU retires about one line per clock.
// Re-transposing through a wave-private LDS scratch to store whole lines was measured and gives the 2.7 us back to
// the LDS round trip.  The consumers keep no fragment alive across the epilogue (stage 0 of the next tile is read from
// the ring after it), so the register budget is the fresh-block kernel's (166 VGPRs, no spills).
template <bool B_KC>
__global__ __launch_bounds__(SplitCfg<XBM>::NTHR, 1) void gemm_bx3p_kernel(GemmArgs g) {
  using Cfg = SplitCfg<XBM>;
  constexpr int BM = XBM, BN = XBN, TM = 2, TN = 2;
  constexpr int NCONS = Cfg::NCONS, NPROD = Cfg::NPROD, STAGE = Cfg::STAGE, RING = 3;
  constexpr int A_BYTES = BM * XROWB;
  static_assert(Cfg::RING == 3, "ring");
  __shared__ __attribute__((aligned(16))) char lds[RING * STAGE];

  const FdGemmDesc& d = g.d;
  const int tid = (int)threadIdx.x;
  const int nblk = g.nblk_m * g.nblk_n;
  const int G = (int)gridDim.x, first = (int)blockIdx.x;       // G <= nblk
  const int ntiles = (nblk - first + G - 1) / G;
  const int z = (int)blockIdx.y;
  const int zo = z / d.bdiv, zi = z % d.bdiv;
  const int nk = (d.K + XBK - 1) / XBK;
  const int total = ntiles * nk;

  if (tid >= NCONS) {
    const int ptid = tid - NCONS;
    fd::raise_wave_priority();
    const float* __restrict__ A = d.A + zo * d.a_so + zi * d.a_si;
    const float* __restrict__ B = d.B + zo * d.b_so + zi * d.b_si;
    SplitStager<BM, true, NPROD> sa;
    SplitStager<BN, B_KC, NPROD> sb;
    constexpr int NSA = SplitStager<BM, true, NPROD>::NS, NSB = SplitStager<BN, B_KC, NPROD>::NS;
    constexpr int NSET = 2;
    float ra[NSET][NSA][8], rb[NSET][NSB][8];
    int lt = first, lk = 0;   // tile and k stage of the next global load
    {
      const int lid = fd_xcd_swizzle(lt, nblk);
      sa.init(A, d.a_rs, d.a_cs, (lid / g.nblk_n) * BM, d.M, 0, ptid);
      sb.init(B, d.b_cs, d.b_rs, (lid % g.nblk_n) * BN, d.N, 0, ptid);
    }
    auto issue = [&](float (&xa)[NSA][8], float (&xb)[NSB][8]) {
      sa.load(xa, lk * XBK, d.K);
      sb.load(xb, lk * XBK, d.K);
      if (++lk == nk) {
        lk = 0;
        lt += G;
        if (lt < nblk) {
          const int lid = fd_xcd_swizzle(lt, nblk);
          sa.target(A, d.a_rs, (lid / g.nblk_n) * BM, d.M, 0, ptid);
          sb.target(B, d.b_cs, (lid % g.nblk_n) * BN, d.N, 0, ptid);
        }
      }
    };
    int wbuf = 0;
    auto put = [&](float (&xa)[NSA][8], float (&xb)[NSB][8]) {
      char* dst = lds + wbuf * STAGE;
      sa.store(xa, dst);
      sb.store(xb, dst + A_BYTES);
      wbuf = (wbuf == RING - 1) ? 0 : wbuf + 1;
    };
    constexpr int AHEAD = RING - 1;
#pragma unroll
    for (int u = 0; u < NSET; ++u)
      if (u < total) issue(ra[u], rb[u]);
#pragma unroll
    for (int u = 0; u < AHEAD; ++u)
      if (u < total) {
        put(ra[u], rb[u]);
        if (u + NSET < total) issue(ra[u], rb[u]);
      }
    __syncthreads();
    auto step = [&](int it, float (&xa)[NSA][8], float (&xb)[NSB][8]) {
      if (it + AHEAD < total) {
        put(xa, xb);
        if (it + AHEAD + NSET < total) issue(xa, xb);
      }
      __syncthreads();
    };
    for (int it = 0; it < total; it += NSET) {
#pragma unroll
      for (int u = 0; u < NSET; ++u)
        if (it + u < total) step(it + u, ra[(AHEAD + u) % NSET], rb[(AHEAD + u) % NSET]);
    }
    return;
  }

  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  float* __restrict__ C = d.C + zo * d.c_so + zi * d.c_si;
  const int a_frag = ((wm * TM) * 32 + l31) * XROWB + h * 16;
  const int b_frag = A_BYTES + ((wn * TN) * 32 + l31) * XROWB + h * 16;
  int rbuf = 0;
  __syncthreads();   // the first stages are in the ring
  for (int ti = 0; ti < ntiles; ++ti) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    uint4 fa[2][TM][3], fb[2][TN][3];
    auto read_frags = [&](uint4 (&xa)[TM][3], uint4 (&xb)[TN][3]) {
      const char* st = lds + rbuf * STAGE;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          xa[i][s] = *reinterpret_cast<const uint4*>(st + a_frag + i * 32 * XROWB + s * 2 * XBK);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          xb[j][s] = *reinterpret_cast<const uint4*>(st + b_frag + j * 32 * XROWB + s * 2 * XBK);
      }
      rbuf = (rbuf == RING - 1) ? 0 : rbuf + 1;
    };
    auto mfmas = [&](uint4 (&xa)[TM][3], uint4 (&xb)[TN][3]) {
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = fd::mfma_32x32x16_bf16(xb[j][PB[p]], xa[i][PA[p]], acc[i][j]);
      }
    };
    auto step = [&](int it, uint4 (&ca)[TM][3], uint4 (&cb)[TN][3], uint4 (&na)[TM][3], uint4 (&nb)[TN][3]) {
      if (it + 1 < nk) read_frags(na, nb);
      mfmas(ca, cb);
      fd::block_barrier_nofence();
    };
    read_frags(fa[0], fb[0]);
    for (int it = 0; it < nk; it += 2) {
      step(it, fa[0], fb[0], fa[1], fb[1]);
      if (it + 1 < nk) step(it + 1, fa[1], fb[1], fa[0], fb[0]);
    }
    const int lid = fd_xcd_swizzle(first + ti * G, nblk);
    const int mb = (lid / g.nblk_n) * BM + wm * TM * 32, nb = (lid % g.nblk_n) * BN + wn * TN * 32;
    store_tile_t<TM, TN>(d, C, acc, mb, nb, h, l31, g.epi_vec != 0);
  }
}
