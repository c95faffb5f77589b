#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* buf, int nbytes, float* out, int nslots) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)blockIdx.x * nbytes / 4, 0, nbytes, 0x00020000);
  int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int s = wave; s < nslots; s += 8) {
    f32x4 v = {(float)(s * 1000 + lane) + 0.1f, (float)(s * 1000 + lane) + 0.2f, (float)(s * 1000 + lane) + 0.3f, (float)(s * 1000 + lane) + 0.4f};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane * 16, s * 1024, 0);
  }
  __syncthreads();
  for (int s = wave; s < nslots; s += 8) {
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, s * 1024, 0));
    for (int i = 0; i < 4; ++i) out[((size_t)blockIdx.x * nslots + s) * 256 + lane * 4 + i] = v[i];
  }
}
int main() {
  const int nslots = 120, nbytes = nslots * 1024, nb = 4;
  float *buf, *out;
  hipMalloc(&buf, (size_t)nb * nbytes); hipMalloc(&out, (size_t)nb * nslots * 256 * 4);
  hipMemset(buf, 0, (size_t)nb * nbytes); hipMemset(out, 0, (size_t)nb * nslots * 256 * 4);
  k<<<nb, 512>>>(buf, nbytes, out, nslots);
  std::vector<float> h((size_t)nb * nslots * 256), hb((size_t)nb * nbytes / 4);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), buf, hb.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0, badmem = 0;
  for (int b = 0; b < nb; ++b) for (int s = 0; s < nslots; ++s) for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    float e = (float)(s * 1000 + l) + 0.1f * (i + 1);
    float g = h[((size_t)b * nslots + s) * 256 + l * 4 + i], m = hb[(size_t)b * nbytes / 4 + s * 256 + l * 4 + i];
    if (g != e) { if (bad < 8) printf("load mismatch b%d s%d l%d i%d got %f exp %f\n", b, s, l, i, g, e); ++bad; }
    if (m != e) { if (badmem < 8) printf("mem mismatch b%d s%d l%d i%d got %f exp %f\n", b, s, l, i, m, e); ++badmem; }
  }
  printf("bad loads %d bad mem %d\n", bad, badmem);
  return 0;
}
