// EXPERIMENT (not part of libfd_hip.so): zb = z W40^T + b40 by the streaming sibling of ipa_dz_acc_kernel.  130 us against
// 119 us for the 128 x 32-tile GEMM of fd_gemm at B=30 x N=128 (round 2): with K = 128 the fp32 MFMA chain and two waves per
// SIMD do not cover the loads.
#include "fd_common.h"
#include "fd_experiments.h"

namespace {

constexpr int ZB = 40;

// ---------------------------------------------------------------- zb = z W40^T + b40
// linear_b and down_z of the pair tensor in one streaming pass (ipa_pytorch.py:380-386,455): zb[p, 0:40] =
// W40[0:40, 0:128] z[p, :] + b40.  Same structure as ipa_dz_acc_kernel: W40 as MFMA B fragments in registers for the whole
// launch (two column tiles: 32 + 8 of 32 used), the z tile straight from global memory in A layout, fp32 MFMA.
__global__ __launch_bounds__(256) void ipa_zb_kernel(const float* __restrict__ z, const float* __restrict__ W40,
                                                     const float* __restrict__ b40, float* __restrict__ zb, long rows) {
  const int lane = fd::lane_id(), wave = fd::wave_id();
  const int h = lane >> 5, l31 = lane & 31;
  // B fragments: B[k][n] = W40[n][k], k = 8 g + 4 h + t contiguous in memory -> one float4 per (g, column tile)
  float4 w[16][2];
  const int n1 = 32 + (l31 < ZB - 32 ? l31 : ZB - 33);        // second tile: columns 32..39, the other lanes duplicate 39
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    w[g][0] = *reinterpret_cast<const float4*>(W40 + l31 * 128 + 8 * g + 4 * h);
    w[g][1] = *reinterpret_cast<const float4*>(W40 + n1 * 128 + 8 * g + 4 * h);
  }
  const float bias0 = b40 ? b40[l31] : 0.f, bias1 = b40 ? b40[n1] : 0.f;
  const long ntiles = (rows + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long r0 = tile * 32;
    const long ra = (r0 + l31 < rows) ? r0 + l31 : rows - 1;
    float4 a[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) a[g] = *reinterpret_cast<const float4*>(z + ra * 128 + 8 * g + 4 * h);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = bias0; acc1[r] = bias1; }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc0 = fd::mfma_32x32x2(a[g].x, w[g][0].x, acc0); acc1 = fd::mfma_32x32x2(a[g].x, w[g][1].x, acc1);
      acc0 = fd::mfma_32x32x2(a[g].y, w[g][0].y, acc0); acc1 = fd::mfma_32x32x2(a[g].y, w[g][1].y, acc1);
      acc0 = fd::mfma_32x32x2(a[g].z, w[g][0].z, acc0); acc1 = fd::mfma_32x32x2(a[g].z, w[g][1].z, acc1);
      acc0 = fd::mfma_32x32x2(a[g].w, w[g][0].w, acc0); acc1 = fd::mfma_32x32x2(a[g].w, w[g][1].w, acc1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < rows) {
        zb[row * ZB + l31] = acc0[r];
        if (l31 < ZB - 32) zb[row * ZB + 32 + l31] = acc1[r];
      }
    }
  }
}

}  // namespace

extern "C" int fd_ipa_zb(const float* z, const float* W40, const float* b40, float* zb, long rows, void* stream) {
  FD_CHECK_ARG(z && W40 && zb, "fd_ipa_zb: null operand");
  FD_CHECK_ARG(fd_aligned16(z) && fd_aligned16(W40), "fd_ipa_zb: z / W40 must be 16-byte aligned");
  if (rows == 0) return FD_OK;
  long g = ((rows + 31) / 32 + 3) / 4;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(ipa_zb_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, z, W40, b40, zb, rows);
  FD_CHECK_LAUNCH("fd_ipa_zb");
  return FD_OK;
}
