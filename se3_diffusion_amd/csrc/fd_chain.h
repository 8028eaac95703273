// Register-chained split-bf16 MLP machinery shared by the fused pair-level kernels (fd_edge_mlp.hip: edge transition,
// fd_edge_embed.hip: edge embedder).  Included inside the including file's anonymous namespace.
//
// A wave owns 16 rows for a whole chain of linear layers and accumulates TRANSPOSED with v_mfma_f32_16x16x32_bf16
// (weights = row operand), so lane (m = l & 15, g = l >> 4) holds units n = 16 nb + 4 g + r of its own row; two
// consecutive 16-blocks are the B-operand fragment of the next layer's 32-k step when that layer's weights are packed
// in "chained" k order (slot (g, e') <-> k = k0 + 16 (e' >> 2) + 4 g + (e' & 3)); operands that come from memory use
// the "natural" order k = k0 + 8 g + e'.  Weights are pre-split into three bf16 planes and packed as units of
// [4 n-blocks of 16][3 planes][64 lanes] x 16 B (12 KB, one 32-k step) in consumption order; stages of four units
// stream through a two-stage LDS ring by LDS-DMA.
#pragma once

#ifndef EM_WAVES
#define EM_WAVES 4                         // waves per block (8 or 4); a stage holds EM_WAVES / 2 units so that every wave
#endif                                     // copies 6 KB (six 1 KB LDS-DMA pieces) of it
constexpr int EM_ROWS = 16 * EM_WAVES;     // rows per block tile
constexpr int EM_PIECE = 1024;             // one fragment: 64 lanes x 16 B
constexpr int EM_UNIT = 12 * EM_PIECE;     // 4 n-blocks x 3 planes
#ifndef EM_UPS
#define EM_UPS (EM_WAVES / 2)              // units per stage (a stage = one barrier interval of the weight stream)
#endif
constexpr int EM_STAGE = EM_UPS * EM_UNIT; // 24 KB with two units
constexpr int EM_BLOCKS_PER_CU = 8 / EM_WAVES;
constexpr int EM_PPW = 12 * EM_UPS / EM_WAVES;   // 1 KB LDS-DMA pieces of a stage per wave: 6 (4 waves x 2 units) or 3 (8 waves x 2 units)
static_assert(EM_PPW * EM_WAVES == 12 * EM_UPS && (EM_PPW == 6 || EM_PPW == 3), "a stage must split into 3 or 6 whole pieces per wave");

__device__ __forceinline__ void em_split8(const float (&x)[8], uint4& s0, uint4& s1, uint4& s2) {
  unsigned t0[4], t1[4], t2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = x[2 * j], v = x[2 * j + 1];
    const unsigned hh = fd::pack_bf16(u, v);
    const float ru = u - fd::bf16lo_f32(hh), rv = v - fd::bf16hi_f32(hh);
    const unsigned mm = fd::pack_bf16(ru, rv);
    const float qu = ru - fd::bf16lo_f32(mm), qv = rv - fd::bf16hi_f32(mm);
    t0[j] = hh;
    t1[j] = mm;
    t2[j] = fd::pack_bf16(qu, qv);
  }
  s0 = make_uint4(t0[0], t0[1], t0[2], t0[3]);
  s1 = make_uint4(t1[0], t1[1], t1[2], t1[3]);
  s2 = make_uint4(t2[0], t2[1], t2[2], t2[3]);
}


__device__ __forceinline__ void em16_split2(const f32x4& lo, const f32x4& hi, uint4& s0, uint4& s1, uint4& s2) {
  float t[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    t[e] = lo[e];
    t[4 + e] = hi[e];
  }
  em_split8(t, s0, s1, s2);
}

// half a unit (2 n-blocks x one 32-k step: 6 fragments, 12 MFMAs): the stage loop reads half-unit i + 1 while half-unit i
// is multiplied, so a wave's MFMAs never wait for its own LDS reads inside a stage (the two waves of a SIMD run the same
// phase at the same time: without the prefetch both sit in the LDS latency together and the pipe idles)
struct Em16Half {
  uint4 w[2][3];
};
__device__ __forceinline__ void em16_read_half(Em16Half& f, const char* __restrict__ u) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s) f.w[i][s] = *reinterpret_cast<const uint4*>(u + (i * 3 + s) * EM_PIECE);
}
__device__ __forceinline__ void em16_mma_half(f32x4& a0, f32x4& a1, const Em16Half& f, const uint4 (&b)[3]) {
  constexpr int PW[6] = {2, 1, 0, 1, 0, 0};
  constexpr int PX[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    a0 = fd::mfma_16x16x32_bf16(f.w[0][PW[p]], b[PX[p]], a0);
    a1 = fd::mfma_16x16x32_bf16(f.w[1][PW[p]], b[PX[p]], a1);
  }
}

// Order of a half-unit's instructions in the stage loops (the fragment reads of half-unit i + 1 are in flight while half-unit i is
// multiplied): one scheduling region per half-unit in which the six ds_read_b128 ride between its first six MFMAs -- a wave's own
// MFMA stream has no read-issue gap and the compiler keeps the fragments' live ranges short: -3 ... -8 % per launch against the
// six reads standing in a block in front of the twelve MFMAs (round 3's order; profiles/r04_edge_variants_*.log).
#define EM_PIN_TOP() fd::sched_pin()
#define EM_PIN_MID()
// (the six reads ride behind the FIRST six MFMAs -- one each --, so the last one has six MFMAs = ~100 cycles to return before the
// next half-unit needs it; spread over all twelve, the next half-unit opened with an lgkmcnt(0) stall)
#define EM_GROUPS(has_read)                                                                     \
  if (has_read) {                                                                               \
    fd::sched_group<0x008, 1>(); fd::sched_group<0x100, 1>(); fd::sched_group<0x008, 1>(); fd::sched_group<0x100, 1>(); \
    fd::sched_group<0x008, 1>(); fd::sched_group<0x100, 1>(); fd::sched_group<0x008, 1>(); fd::sched_group<0x100, 1>(); \
    fd::sched_group<0x008, 1>(); fd::sched_group<0x100, 1>(); fd::sched_group<0x008, 1>(); fd::sched_group<0x100, 1>(); \
    fd::sched_group<0x008, 6>();                                                                \
  }
