// Does a wave's own non-MFMA work issue in the shadow of its MFMAs?  16 independent v_mfma_f32_16x16x4_f32 per iteration with
// NV v_add_f32 / NS s_add_u32 sprinkled between them, 1 or 2 waves per SIMD (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int NS>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0, v0 = a0, v1 = b0, v2 = 1.f, v3 = 2.f;
  unsigned s0 = iters;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          if (q & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v2)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v1) : "v"(v3));
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
      }
  }
  f32x4 r = acc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) r += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + v0 + v1 + (float)s0;
}
template <int NV, int NS> void run(float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 4096;
  for (int threads : {256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL((k<NV, NS>), dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)(threads / 256) * iters * 16;
    printf("%d v_add + %d s_add per MFMA, %d wave(s) per SIMD: %.1f cycles per MFMA per SIMD (2.4 GHz)\n", NV, NS, threads / 256, ms * 1e-3 * 2.4e9 / mfma_per_simd);
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  run<0, 0>(out, e0, e1); run<1, 0>(out, e0, e1); run<2, 0>(out, e0, e1); run<4, 0>(out, e0, e1); run<8, 0>(out, e0, e1);
  run<0, 2>(out, e0, e1); run<0, 4>(out, e0, e1); run<0, 8>(out, e0, e1);
  return 0;
}
