// What does a tile's TAIL evaluation (<= 2 live rows, 4-row MFMA path of ode_fast.hip: eval_m) cost as a function of how many
// weight fragments a wave keeps in flight?  A stand-alone model of the x branch of one field evaluation at the headline shape
// (d = 256, hidden 128): five barrier-separated layers, 8 waves x one 16-column tile each, weights streamed L2 -> VGPR through a
// buffer descriptor, v_mfma_f32_4x4x1 on a 4-row LDS image, gsum + activation + one LDS store per lane, workgroup barrier.
// 56 fragments (56 KB) per wave and evaluation; ring of R fragments (R in {8, 14, 28, 56}); the stream runs ACROSS layers,
// barriers and evaluations (the address sequence is static).        hipcc --offload-arch=gfx950 -O3 micro_eval.hip -o micro_eval
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 256, H = 128, LDX = D + 8, LDH = H + 8, NW = 8;
constexpr int W2 = 0, W3 = W2 + D * H, W5 = W3 + H * H, W6 = W5 + 2 * H * H, W7 = W6 + H * H, WTOT = W7 + H * D;
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 gsum(f32x4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = v[r];
    x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
    float xa = x, xb = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(xa), "+v"(xb));
    v[r] = xa + xb;
  }
  return v;
}
// fragment f of an evaluation (0..55) for wave w: byte offset in the packed weights.  x1: 16 k-blocks of tile w; x2 / j1 / j2: 8
// each; out: tiles w and w + 8, interleaved per k-block (f = 40 + 2 kb + tile)
__device__ __forceinline__ constexpr int frag_const(int f) {
  return f < 16 ? W2 * 4 + f * 1024 : f < 24 ? W3 * 4 + (f - 16) * 1024 : f < 32 ? W5 * 4 + (f - 24) * 1024 : f < 40 ? W6 * 4 + (f - 32) * 1024
         : W7 * 4 + ((f - 40) >> 1) * 1024 + ((f - 40) & 1) * 8 * 8 * 1024;
}
__device__ __forceinline__ constexpr int frag_wave(int f) {       // bytes per wave index
  return f < 16 ? 16 * 1024 : f < 24 ? 8 * 1024 : f < 32 ? 16 * 1024 : 8 * 1024;
}

template <int R, bool BARRIERS>
__global__ __launch_bounds__(512) void eval_kernel(const float* Wp, int evals, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* X = lds;                    // [4][LDX]
  float* A1 = X + 4 * LDX;           // [4][LDH] x 4 images
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp), 0, WTOT * 4, 0x00020000);
  for (int i = threadIdx.x; i < 4 * LDX + 16 * LDH; i += 512) lds[i] = 0.01f * (float)((i * 37) % 101 - 50);
  __syncthreads();
  f32x4 ring[R];
  const int voff = lane * 16;
#pragma unroll
  for (int f = 0; f < R; ++f) ring[f] = bload(wr, voff, frag_const(f % 56) + wave * frag_wave(f % 56));
  const float* ax = X + (c & 3) * LDX + 4 * g;
  const float* ah = A1 + (c & 3) * LDH + 4 * g;
  float* eh = A1 + g * LDH + 16 * wave + c;
  const int rank = g & 1;
  const bool is_t = g >= 2;
  float accum = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int e = 0; e < evals; ++e) {
    f32x4 acc[2][4];
    auto zero = [&]() {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t][s] = f32x4{0, 0, 0, 0};
    };
    auto finish = [&](int t, float* dst) {
      const f32x4 pre = gsum((acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]));
      const float pv = rank ? pre[1] : pre[0], pt = rank ? pre[3] : pre[2];
      *dst = is_t ? (pv > 0.f ? pt : 0.f) : fmaxf(pv, 0.f);
    };
    // one fragment = one k-block of one tile: a from LDS, four 4x4x1 MFMAs; then the ring slot is refilled R fragments ahead
#define FRAG(F, AROW, KBL, T)                                                                                           \
    {                                                                                                                  \
      const f32x4 a = *reinterpret_cast<const f32x4*>((AROW) + (KBL) * 16);                                            \
      const f32x4 b = ring[(F) % R];                                                                                   \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[T][s] = mfma1(a[s], b[s], acc[T][s]);                        \
      ring[(F) % R] = bload(wr, voff, frag_const(((F) + R) % 56) + wave * frag_wave(((F) + R) % 56));                  \
      asm volatile("" ::: "memory");                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
    zero();
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) FRAG(kb, ax, kb, 0)
    finish(0, eh);
    if (BARRIERS) __syncthreads();
    zero();
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) FRAG(16 + kb, ah, kb, 0)
    finish(0, eh + 4 * LDH);
    if (BARRIERS) __syncthreads();
    zero();
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) FRAG(24 + kb, ah + 4 * LDH, kb, 0)
    finish(0, eh + 8 * LDH);
    if (BARRIERS) __syncthreads();
    zero();
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) FRAG(32 + kb, ah + 8 * LDH, kb, 0)
    finish(0, eh + 12 * LDH);
    if (BARRIERS) __syncthreads();
    zero();
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) { FRAG(40 + 2 * kb, ah + 12 * LDH, kb, 0) FRAG(41 + 2 * kb, ah + 12 * LDH, kb, 1) }
    {
      const f32x4 p0 = gsum((acc[0][0] + acc[0][1]) + (acc[0][2] + acc[0][3])), p1 = gsum((acc[1][0] + acc[1][1]) + (acc[1][2] + acc[1][3]));
      const float v0 = rank ? p0[1] : p0[0], v1 = rank ? p1[1] : p1[0];
      accum += v0 + v1;
      if (!is_t) { X[rank * LDX + 4 + 16 * wave + c] = 0.001f * v0; X[rank * LDX + 4 + 128 + 16 * wave + c] = 0.001f * v1; }
    }
    if (BARRIERS) __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 r = ring[0];
#pragma unroll
  for (int f = 1; f < R; ++f) r += ring[f];
  out[blockIdx.x * 512 + threadIdx.x] = accum + r[0] + r[1] + r[2] + r[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, bool B>
static void run(const float* dW, float* dO, unsigned long long* dC, int wgs) {
  const int evals = 2000;
  const size_t sm = (4 * LDX + 16 * LDH) * 4;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((eval_kernel<R, B>), dim3(wgs), dim3(512), sm, 0, dW, evals, dO, dC);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> c(wgs);
  (void)hipMemcpy(c.data(), dC, wgs * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : c) m += (double)v; m /= wgs;
  printf("ring %2d fragments (%3d VGPRs), barriers %d, %3d workgroups: %8.0f shader cycles (s_memtime) per evaluation\n", R, 4 * R, (int)B, wgs, m / evals);
}
int main() {
  std::vector<float> W(WTOT);
  for (size_t i = 0; i < W.size(); ++i) W[i] = 0.01f * (float)((i * 2654435761u >> 20) % 201) - 1.0f;
  float *dW, *dO; unsigned long long* dC;
  (void)hipMalloc(&dW, WTOT * 4); (void)hipMalloc(&dO, 256 * 512 * 4); (void)hipMalloc(&dC, 256 * 8);
  (void)hipMemcpy(dW, W.data(), WTOT * 4, hipMemcpyHostToDevice);
  for (int wgs : {1, 8, 256}) {
    run<8, true>(dW, dO, dC, wgs); run<14, true>(dW, dO, dC, wgs); run<28, true>(dW, dO, dC, wgs); run<56, true>(dW, dO, dC, wgs);
    run<8, false>(dW, dO, dC, wgs); run<28, false>(dW, dO, dC, wgs);
  }
  return 0;
}
