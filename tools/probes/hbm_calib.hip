// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access patterns of the fused
// pair-level kernels (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"):
//   copy16        every lane moves one float4, lanes contiguous (a wave covers 1 KB): streaming read + streaming write
//   copy16_seg64  lane (m = l & 15, g = l >> 4) moves the float4 at row m, column 16 nb + 4 g of a [rows,128] tensor, one
//                 instruction per nb: 16 rows x 64 contiguous bytes per instruction -- how edge_mlp16 / edge_embed load x and
//                 store h1 / h2 / y
//   read16        reads only (sum kept in a register, one store per wave)
// Each kernel moves BYTES = 1 GiB in and (the copies) 1 GiB out, 4x the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O2 hbm_calib.hip -o hbm_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_calib     (and a second run with --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void copy16(const float4* __restrict__ in, float4* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i];
}

__global__ __launch_bounds__(256) void copy16_seg64(const float* __restrict__ in, float* __restrict__ out, long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  for (long t = (long)blockIdx.x * 4 + wave; t * 16 < rows; t += (long)gridDim.x * 4) {
    const long row = t * 16 + m;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const long o = row * 128 + 16 * nb + 4 * g;
      *reinterpret_cast<float4*>(out + o) = *reinterpret_cast<const float4*>(in + o);
    }
  }
}

__global__ __launch_bounds__(256) void read16(const float4* __restrict__ in, float* __restrict__ out, long n) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float4 v = in[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 1.2345e-30f) out[blockIdx.x] = s;
}

int main() {
  const long bytes = 1L << 30, n4 = bytes / 16, rows = bytes / 512;
  float *a, *b;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMemset(a, 0x3c, bytes);
  hipMemset(b, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(copy16, dim3(4096), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4);
    hipLaunchKernelGGL(copy16_seg64, dim3(2048), dim3(256), 0, 0, a, b, rows);
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const float4*)a, b, n4);
  }
  hipDeviceSynchronize();
  printf("hbm_calib: 3 x (copy16, copy16_seg64, read16) over %ld bytes each way\n", bytes);
  return 0;
}
