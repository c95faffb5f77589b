// standalone check of the 4-row MFMA job (exec_group_m + gsum of ode_fast.hip) against a scalar reference (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 gsum(f32x4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = v[r];
    x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
    // lane ^ 32: v_permlane32_swap exchanges a[32:63] with b[0:31] in place in BOTH registers.  Through inline assembly: the
    // builtin's second result is miscompiled by hipcc 7.2 (p[0] + p[1] came out as v_pk_add v, v, v of the FIRST result; found
    // with tools/mb/m4job.hip), and inline assembly gets no hazard padding from the compiler, hence the s_nop
    float xa = x, xb = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(xa), "+v"(xb));
    v[r] = xa + xb;
  }
  return v;
}
// A: [4][K] row-major (lda), Wp: packed [kb][lane][4] for ONE column tile; out: [64 lanes][4 rows]
__global__ void job(const float* A, int lda, const f32x4* Wp, int KB, float* out, float* raw) {
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const float* arow = A + (c & 3) * lda + 4 * g;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int kb = 0; kb < KB; ++kb) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(arow + kb * 16);
    const f32x4 b = Wp[kb * 64 + lane];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = mfma1(a[s], b[s], acc[s]);
  }
  const f32x4 part = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  for (int r = 0; r < 4; ++r) raw[lane * 4 + r] = part[r];
  const f32x4 pre = gsum(part);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = pre[r];
}
int main() {
  const int K = 128, KB = K / 16, lda = K + 8;
  std::vector<float> A(4 * lda), W(K * 16), Wp(KB * 64 * 4);
  for (auto& v : A) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : W) v = (rand() % 2001 - 1000) / 1000.f;
  for (int kb = 0; kb < KB; ++kb) for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) Wp[(kb * 64 + l) * 4 + s] = W[(16 * kb + 4 * (l >> 4) + s) * 16 + (l & 15)];
  float *dA, *dW, *dO, *dR;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dW, Wp.size() * 4); (void)hipMalloc(&dO, 1024); (void)hipMalloc(&dR, 1024);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dW, Wp.data(), Wp.size() * 4, hipMemcpyHostToDevice);
  job<<<1, 64>>>(dA, lda, (const f32x4*)dW, KB, dO, dR);
  float h[256], hr[256]; (void)hipMemcpy(h, dO, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, dR, 1024, hipMemcpyDeviceToHost);
  double worst = 0, worst_raw = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    double ref = 0, refp = 0;
    for (int k = 0; k < K; ++k) { const double t = (double)A[r * lda + k] * W[k * 16 + (l & 15)]; ref += t; if (((k & 15) >> 2) == (l >> 4)) refp += t; }
    worst = fmax(worst, fabs(h[l * 4 + r] - ref)); worst_raw = fmax(worst_raw, fabs(hr[l * 4 + r] - refp));
  }
  for (int c = 0; c < 2; ++c) { double t = 0; for (int g = 0; g < 4; ++g) t += hr[(16 * g + c) * 4]; printf("col %d row 0: partials %g %g %g %g sum %g | gsum lanes %g %g %g %g\n", c, hr[c * 4], hr[(16 + c) * 4], hr[(32 + c) * 4], hr[(48 + c) * 4], t, h[c * 4], h[(16 + c) * 4], h[(32 + c) * 4], h[(48 + c) * 4]); }
  { double ref = 0; for (int k = 0; k < K; ++k) ref += (double)A[k] * W[k * 16]; printf("ref col 0 row 0: %g\n", ref); }
  printf("4-row job: max |partial - ref| %.3g, max |gsum - ref| %.3g; lane 17: %g %g (ref col 1)\n", worst_raw, worst, h[17 * 4], h[1 * 4]);
  return 0;
}
