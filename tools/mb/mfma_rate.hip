// Sustained v_mfma_f32_16x16x4_f32 rate per SIMD with 1 and 2 waves per SIMD, 4 or 8 independent accumulators (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16 / NACC * 4; ++s)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 r = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) r += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}
int main() {
  float* out; (void)hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4096;                                 // x 64 MFMAs per iteration
  for (int nacc : {4, 8})
    for (int threads : {256, 512})
      for (int grid : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(e0, 0);
          if (nacc == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
          else hipLaunchKernelGGL(k<8>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
          (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)grid * (threads / 64) * iters * 64, fl = mfma * 2048;
        printf("acc %d, %d threads x %d WGs: %.3f ms, %.1f TFLOP/s (%.0f %% of 157.3); cycles per MFMA per SIMD at 2.4 GHz: %.1f\n", nacc, threads, grid, ms,
               fl / ms * 1e-9, fl / ms * 1e-9 / 157.3 * 100, ms * 1e-3 * 2.4e9 / (mfma / 1024));
      }
  return 0;
}
