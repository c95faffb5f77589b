// micro-benchmark / layout probe for v_mfma_f32_4x4x1_16b_f32 (development aid): hipcc --offload-arch=gfx950 -O3 -o m4 m4.hip
// Finds, for every output lane l and register r, WHICH lane's A value and which lane's B value were multiplied.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
template <int NACC>
__global__ void rate(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x + blockDim.x * blockIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float *d, *da, *db; unsigned long long* c;
  (void)hipMalloc(&d, 1 << 20); (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&c, 64);
  // primes: a product a_i * b_j identifies (i, j) uniquely
  float ha[64], hb[64], h[256];
  int p = 2, n = 0; int primes[128];
  while (n < 128) { bool ok = true; for (int q = 2; q * q <= p; ++q) if (p % q == 0) ok = false; if (ok) primes[n++] = p; ++p; }
  for (int i = 0; i < 64; ++i) { ha[i] = (float)primes[i]; hb[i] = (float)primes[64 + i]; }
  (void)hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, d);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int okA = 0, okB = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int fi = -1, fj = -1;
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) if (ha[i] * hb[j] == h[l * 4 + r]) { fi = i; fj = j; }
    if (l < 8 || l == 21 || l == 63) printf("D[lane %2d][reg %d] = a[lane %2d] * b[lane %2d]\n", l, r, fi, fj);
    okA += fi == 4 * (l / 4) + r; okB += fj == l;
  }
  printf("A from lane 4 (l / 4) + r: %d / 256; B from own lane: %d / 256\n", okA, okB);
  return 0;
}
