// Sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip with register-resident operands (no memory traffic in the
// loop): the practical ceiling the split-bf16 kernels are priced against.  Operands: N(0,1)-like random bit patterns or
// zeros (data-dependent power).   hipcc --offload-arch=gfx950 -O2 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(const uint4* in, float* out, int iters) {
  uint4 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), c3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int blocks = 256 * 4, iters = 20000;
  uint4* in; float* out;
  hipMalloc(&in, 1024 * 16); hipMalloc(&out, blocks * 256 * 4);
  unsigned h[4096];
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 4096; ++i) {
      // two bf16 per word: random sign / mantissa, exponent around 1.0 (mode 0) or all zero (mode 1)
      unsigned lo = 0x3f00 | (rand() & 0x80ff), hi = 0x3f00 | (rand() & 0x80ff);
      h[i] = mode == 0 ? (lo | (hi << 16)) : 0u;
    }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      k<<<blocks, 256>>>(in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 /*waves*/ * iters * 4 * 2.0 * 32 * 32 * 16;
      printf("%s operands: %.3f ms  %.1f TFLOP/s bf16 (= %.1f TFLOP/s of fp32-accurate split products)\n",
             mode == 0 ? "random" : "zero  ", ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
    }
  }
  return 0;
}
