// TEST INFRASTRUCTURE ONLY -- host emulation of the gfx950 intrinsics wrapped by
// se3_diffusion_amd/csrc/gfx950/fd_intrin.h (same names, same semantics).
// Fragment maps follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//                 D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
//                 D reg r -> row 4*(l>>4)+r, col l&15
// Result is a k-ordered fmaf chain (bitwise the hardware behaviour).
#pragma once
#include <hip/hip_runtime.h>

#define FD_BACKEND_NAME "emu"

struct f32x16 {
  float v[16];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
struct f32x4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};

namespace fd {

// 16 bytes moved as one unit (the device header: a native vector type)
struct alignas(16) u32x4 { unsigned v[4]; };

static inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
  struct AB { float a, b; } ab{a, b};
  auto tab = hipemu::wave_exchange(&ab, sizeof(ab));
  int l = hipemu::g_cur->lane;
  int col = l & 31, hi = l >> 5;
  f32x16 d;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      AB x, y;
      memcpy(&x, tab[row + 32 * k], sizeof(AB));   // A[row][k]
      memcpy(&y, tab[col + 32 * k], sizeof(AB));   // B[k][col]
      acc = fmaf(x.a, y.b, acc);
    }
    d[r] = acc;
  }
  return d;
}

static inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
  struct AB { float a, b; } ab{a, b};
  auto tab = hipemu::wave_exchange(&ab, sizeof(ab));
  int l = hipemu::g_cur->lane;
  int col = l & 15, g = l >> 4;
  f32x4 d;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * g + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      AB x, y;
      memcpy(&x, tab[row + 16 * k], sizeof(AB));
      memcpy(&y, tab[col + 16 * k], sizeof(AB));
      acc = fmaf(x.a, y.b, acc);
    }
    d[r] = acc;
  }
  return d;
}

static inline float bf16lo_f32(unsigned w) { unsigned u = w << 16; float f; memcpy(&f, &u, 4); return f; }
static inline float bf16hi_f32(unsigned w) { unsigned u = w & 0xffff0000u; float f; memcpy(&f, &u, 4); return f; }
static inline unsigned bf16_rne(float x) {
  unsigned u; memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline unsigned pack_bf16(float lo, float hi) { return (bf16_rne(lo) & 0xffffu) | (bf16_rne(hi) << 16); }

// 32x32x16 bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31]; exact products, f32 accumulate (k-ordered here)
static inline f32x16 mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) {
  struct AB { unsigned a[4], b[4]; } ab{{a.x, a.y, a.z, a.w}, {b.x, b.y, b.z, b.w}};
  auto tab = hipemu::wave_exchange(&ab, sizeof(ab));
  int l = hipemu::g_cur->lane;
  int col = l & 31, hi = l >> 5;
  f32x16 d;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int kh = 0; kh < 2; ++kh) {
      AB x, y;
      memcpy(&x, tab[row + 32 * kh], sizeof(AB));
      memcpy(&y, tab[col + 32 * kh], sizeof(AB));
      for (int e = 0; e < 4; ++e) {
        acc = fmaf(bf16lo_f32(x.a[e]), bf16lo_f32(y.b[e]), acc);
        acc = fmaf(bf16hi_f32(x.a[e]), bf16hi_f32(y.b[e]), acc);
      }
    }
    d[r] = acc;
  }
  return d;
}

// LDS-DMA: lane l copies 16 bytes from its own global address to (wave-uniform base) + 16 l
template <int OFF = 0>
static inline void glds16(const void* g_lane, void* lds_wave_base) {
  memcpy(static_cast<char*>(lds_wave_base) + OFF + 16 * hipemu::g_cur->lane, static_cast<const char*>(g_lane) + OFF, 16);
}

// 16x16x32 bf16: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15]; D reg r -> row 4*(l>>4)+r, col l&15
static inline f32x4 mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
  struct AB { unsigned a[4], b[4]; } ab{{a.x, a.y, a.z, a.w}, {b.x, b.y, b.z, b.w}};
  auto tab = hipemu::wave_exchange(&ab, sizeof(ab));
  int l = hipemu::g_cur->lane;
  int col = l & 15, g = l >> 4;
  f32x4 d;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * g + r;
    float acc = c[r];
    for (int kg = 0; kg < 4; ++kg) {
      AB x, y;
      memcpy(&x, tab[row + 16 * kg], sizeof(AB));
      memcpy(&y, tab[col + 16 * kg], sizeof(AB));
      for (int e = 0; e < 4; ++e) {
        acc = fmaf(bf16lo_f32(x.a[e]), bf16lo_f32(y.b[e]), acc);
        acc = fmaf(bf16hi_f32(x.a[e]), bf16hi_f32(y.b[e]), acc);
      }
    }
    d[r] = acc;
  }
  return d;
}

// ds_read_b64_tr_b16: within each group of 16 lanes, lane i receives element (i & 3) of lanes 4j + (i >> 2), j = 0..3
static inline uint2 lds_read_tr16(const void* p) {
  unsigned short own[4];
  memcpy(own, p, 8);
  auto tab = hipemu::wave_exchange(own, 8);
  const int lane = hipemu::g_cur->lane, base = lane & ~15, i = lane & 15;
  unsigned short r[4];
  for (int j = 0; j < 4; ++j) memcpy(&r[j], tab[base + 4 * j + (i >> 2)] + 2 * (i & 3), 2);
  uint2 out;
  out.x = (unsigned)r[0] | ((unsigned)r[1] << 16);
  out.y = (unsigned)r[2] | ((unsigned)r[3] << 16);
  return out;
}

static inline void glds16a(const void* g_lane, void* lds_wave_base) {
  memcpy(static_cast<char*>(lds_wave_base) + 16 * hipemu::g_cur->lane, g_lane, 16);
}
static inline void glds16x2(const void* g_lane, void* lds_wave_base) {
  for (int k = 0; k < 2; ++k)
    memcpy(static_cast<char*>(lds_wave_base) + 1024 * k + 16 * hipemu::g_cur->lane, static_cast<const char*>(g_lane) + 1024 * k, 16);
}
static inline void glds16x3(const void* g_lane, void* lds_wave_base) {
  for (int k = 0; k < 3; ++k)
    memcpy(static_cast<char*>(lds_wave_base) + 1024 * k + 16 * hipemu::g_cur->lane, static_cast<const char*>(g_lane) + 1024 * k, 16);
}
static inline void glds16x4(const void* g_lane, void* lds_wave_base) {
  for (int k = 0; k < 4; ++k)
    memcpy(static_cast<char*>(lds_wave_base) + 1024 * k + 16 * hipemu::g_cur->lane, static_cast<const char*>(g_lane) + 1024 * k, 16);
}
template <int MASK, int SIZE>
static inline void sched_group() {}
static inline void wait_vmem() {}
static inline void wait_vmem_keep6() {}
template <int N>
static inline void wait_vmem_keep() {}

static inline void raise_wave_priority() {}

static inline void block_barrier_nofence() { hipemu::barrier(); }

static inline void sched_fence() {}
static inline void sched_pin() {}


static inline void store_nt4(float* p, float x, float y, float z, float w) { p[0] = x; p[1] = y; p[2] = z; p[3] = w; }
static inline int uniform(int v) { return v; }
static inline int lane_id() { return hipemu::g_cur->lane; }
static inline int wave_id() { return hipemu::g_cur->wave; }

static inline void lds_add(float* p, float v) { *p += v; }      // (fibers of a block run one at a time)

// (the same four exchanges as the DPP controls of the device version, so the additions pair up identically)
static inline float row16_sum(float v) {
  const int l = hipemu::g_cur->lane;
  v += __shfl(v, l ^ 1);
  v += __shfl(v, l ^ 2);
  v += __shfl(v, (l & ~7) | (7 - (l & 7)));
  v += __shfl(v, (l & ~15) | (15 - (l & 15)));
  return v;
}
static inline float lane_xor8(float v) { return __shfl(v, hipemu::g_cur->lane ^ 8); }
static inline float wave_sum(float v) {       // (same order of additions as the device version)
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
template <typename T>
static inline T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <typename T>
static inline T wave_max(T v) {
  for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o); v = v > u ? v : u; }
  return v;
}

}  // namespace fd
