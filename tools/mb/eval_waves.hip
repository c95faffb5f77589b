// Should the shape-specialised flow step (mfm_amd/csrc/ode_fast.hip) run its FULL-mode field evaluation on 4 waves of up to 512
// registers (one per SIMD) instead of 8 waves of up to 256 (two per SIMD)?  A stand-alone model of the x branch of one evaluation at
// the headline shape (d = 256, hidden 128, 16 chains per workgroup = 16 value rows + 16 tangent rows), one workgroup per CU on every CU:
//   x1  (M = 16, K = 256 -> 128; tangent rows = relu' * (z W_x1), a per-lane constant)      64 MFMA tiles-k per column tile
//   x2, j1, j2  (M = 32, K = 128 -> 128; j1's accumulators start from the time batch's contribution)
//   out (M = 16, K = 128 -> 256; + the gate * clip(grad log pi) terms: ~20 vector instructions per element)
// five barrier-separated layers, weights streamed L2 -> VGPR through a buffer descriptor as ready B fragments with a one-group
// look-ahead that crosses layers and barriers, activations in LDS (row-major, leading dimension = 8 mod 64), exact-f32 MFMA 16x16x4.
// NW = 8: wave w owns column tile w (two tiles of the out layer); NW = 4: wave w owns tiles 2w, 2w + 1 (four of the out layer) and,
// with PIPE, finishes tile 2w before it starts 2w + 1 so that the first tile's epilogue (bias, ReLU, tangent mask, LDS stores) can
// issue in the shadow of the second tile's MFMAs.  Same arithmetic volume per CU in every variant: 2,560 MFMAs per evaluation.
//        hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-sched-strategy=max-ilp eval_waves.hip -o eval_waves && ./eval_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 256, H = 128, LDX = D + 8, LDH = H + 8;
constexpr int W2 = 0, W3 = W2 + D * H, W5 = W3 + H * H, W6 = W5 + H * H, W7 = W6 + H * H, WX = W7 + H * D;      // (j1: the sx half only, K = 128)
constexpr int W0 = WX, W1 = W0 + 2 * H * H, W4 = W1 + H * H, W5T = W4 + H * D, WTOT = W5T + H * H;      // time branch: t1 (K = 256), t2, gate (N = 256), the st half of j1
constexpr int LDF = 2 * H + 8;
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// acc[t][m] += A[m-th 16 rows][16 KB] * W[.., column tile ct0 + t]; fragments of (tile t, k-block kb) at woff + ((ct0 + t) * KB + kb) * 1024.
// Four k-blocks per group, the next group (or the next job's first) requested before the current one executes.
template <int MT, int NT, int KB, int LDA>
__device__ __forceinline__ void job(const float* arow, __amdgpu_buffer_rsrc_t wr, int woff, int ct0, int lane, f32x4 (&acc)[NT][MT]) {
  f32x4 bf[2][NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) bf[0][t][u] = bload(wr, lane * 16, woff + ((ct0 + t) * KB + u) * 1024);
#pragma unroll
  for (int gi = 0; gi < KB / 4; ++gi) {
    if (gi + 1 < KB / 4) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) bf[(gi + 1) & 1][t][u] = bload(wr, lane * 16, woff + ((ct0 + t) * KB + (gi + 1) * 4 + u) * 1024);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4 a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * LDA + (gi * 4 + u) * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][m] = mfma4(a[m][s], bf[gi & 1][t][u][s], acc[t][m]);
    }
  }
}

// TB = 1: the time branch of the NEXT stage (16 rows: Fourier features -> t1 -> t2 -> gate, st half of j1) rides in the phases of this
// evaluation -- t1 beside x1, t2 beside x2, gate + st beside j1 -- instead of one M = 80 batch per attempt (TB = 2: that batch alone, per attempt)
template <int NW, bool PIPE, int TB = 0>
__global__ __launch_bounds__(NW * 64) void eval_kernel(const float* Wp, const float* bias, int evals, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CT = 8 / NW;              // column tiles per wave of a 128-wide layer
  float* X = lds;                         // [16][LDX] stage input (values); halo at +4
  float* A1 = X + 16 * LDX;               // [32][LDH] x1 out (values, tangents)
  float* SX = A1 + 32 * LDH;              // x2 out
  float* J1 = SX + 32 * LDH;
  float* J2 = J1 + 32 * LDH;
  float* FF = TB == 2 ? lds : J2 + 32 * LDH;      // [16 or 80][LDF] Fourier features of the next stage(s)
  float* T1 = FF + (TB == 2 ? 80 : 16) * LDF;     // [16 or 80][LDH]
  float* ST = TB == 2 ? FF : T1 + 16 * LDH;       // (the batch writes st over the consumed Fourier image, as ode_fast.hip does)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp), 0, WTOT * 4, 0x00020000);
  for (int i = threadIdx.x; i < (TB == 2 ? 80 * (LDF + LDH) : 16 * LDX + 4 * 32 * LDH + (TB ? 16 * (LDF + 2 * LDH) : 0)); i += NW * 64) lds[i] = 0.01f * (float)((i * 37) % 101 - 50);
  __syncthreads();
  float tz1[CT][4], bs[5][CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tz1[t][i] = 0.01f * (float)(lane + i + t);
#pragma unroll
    for (int l = 0; l < 5; ++l) bs[l][t] = bias[l * 256 + (wave * CT + t) * 16 + c];
  }
  const float* ax = X + (lane & 15) * LDX + 4 * g + 4;
  float accum = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (TB == 2) {      // the time batch alone: five stage times x 16 rows = M 80, four layers, accumulator-layout results to a global scratch
    static_assert(NW == 8, "");
#pragma unroll 1
    for (int e = 0; e < evals; ++e) {
      f32x4 acc[1][5];
      auto z5 = [&]() { for (int m = 0; m < 5; ++m) acc[0][m] = f32x4{0, 0, 0, 0}; };
      auto st5 = [&](float* dst, float b) {
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[(m * 16 + 4 * g + i) * LDH + wave * 16 + c] = fmaxf(acc[0][m][i] + b, 0.f);
      };
      z5(); job<5, 1, 16, LDF>(FF + (lane & 15) * LDF + 4 * g, wr, W0 * 4, wave, lane, acc); st5(T1, bs[0][0]);
      __syncthreads();
      z5(); job<5, 1, 8, LDH>(T1 + (lane & 15) * LDH + 4 * g, wr, W1 * 4, wave, lane, acc); st5(ST, bs[1][0]);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        z5(); job<5, 1, 8, LDH>(ST + (lane & 15) * LDH + 4 * g, wr, (q < 2 ? W4 : W5T) * 4, q < 2 ? wave + 8 * q : wave, lane, acc);
#pragma unroll
        for (int m = 0; m < 5; ++m) __builtin_nontemporal_store(acc[0][m] + bs[2][0], reinterpret_cast<f32x4*>(out) + ((size_t)blockIdx.x * 15 + q * 5 + m) * 512 + threadIdx.x);
      }
      __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    return;
  }
#pragma unroll 1
  for (int e = 0; e < evals; ++e) {
    // epilogue of a hidden layer: bias, ReLU, tangent mask, 2 x 4 LDS stores per column tile
    auto epi32 = [&](float* dst, f32x4 v, f32x4 tn, float b, int ct) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = v[i] + b;
        dst[(4 * g + i) * LDH + ct * 16 + c] = fmaxf(pre, 0.f);
        dst[(16 + 4 * g + i) * LDH + ct * 16 + c] = pre > 0.f ? tn[i] : 0.f;
      }
    };
    auto hidden = [&](const float* src, float* dst, int woff, int layer, float seed) {      // M = 32, K = 128 -> 128
      const float* ar = src + (lane & 15) * LDH + 4 * g;
      if constexpr (PIPE && CT == 2) {
        f32x4 a0[1][2] = {{{seed, seed, seed, seed}, {0, 0, 0, 0}}}, a1[1][2] = {{{seed, seed, seed, seed}, {0, 0, 0, 0}}};
        job<2, 1, 8, LDH>(ar, wr, woff, wave * 2, lane, a0);
        job<2, 1, 8, LDH>(ar, wr, woff, wave * 2 + 1, lane, a1);      // (tile 0's epilogue below has no dependence on these MFMAs: the scheduler may sink it among them)
        epi32(dst, a0[0][0], a0[0][1], bs[layer][0], wave * 2);
        epi32(dst, a1[0][0], a1[0][1], bs[layer][1], wave * 2 + 1);
      } else {
        f32x4 acc[CT][2];
#pragma unroll
        for (int t = 0; t < CT; ++t) { acc[t][0] = f32x4{seed, seed, seed, seed}; acc[t][1] = f32x4{0, 0, 0, 0}; }
        job<2, CT, 8, LDH>(ar, wr, woff, wave * CT, lane, acc);
#pragma unroll
        for (int t = 0; t < CT; ++t) epi32(dst, acc[t][0], acc[t][1], bs[layer][t], wave * CT + t);
      }
    };
    // x1: values only; tangent rows are the mask times tz1
    {
      f32x4 acc[CT][1];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t][0] = f32x4{0, 0, 0, 0};
      job<1, CT, 16, LDX>(ax, wr, W2 * 4, wave * CT, lane, acc);
#pragma unroll
      for (int t = 0; t < CT; ++t) epi32(A1, acc[t][0], f32x4{tz1[t][0], tz1[t][1], tz1[t][2], tz1[t][3]}, bs[0][t], wave * CT + t);
    }
    if constexpr (TB == 1) {      // t1 of the next stage: M = 16, K = 256
      f32x4 acc[CT][1];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t][0] = f32x4{0, 0, 0, 0};
      job<1, CT, 16, LDF>(FF + (lane & 15) * LDF + 4 * g, wr, W0 * 4, wave * CT, lane, acc);
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) T1[(4 * g + i) * LDH + (wave * CT + t) * 16 + c] = fmaxf(acc[t][0][i] + bs[0][t], 0.f);
    }
    __syncthreads();
    hidden(A1, SX, W3 * 4, 1, 0.f);
    if constexpr (TB == 1) {      // t2
      f32x4 acc[CT][1];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t][0] = f32x4{0, 0, 0, 0};
      job<1, CT, 8, LDH>(T1 + (lane & 15) * LDH + 4 * g, wr, W1 * 4, wave * CT, lane, acc);
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) ST[(4 * g + i) * LDH + (wave * CT + t) * 16 + c] = fmaxf(acc[t][0][i] + bs[1][t], 0.f);
    }
    __syncthreads();
    hidden(SX, J1, W5 * 4, 2, 0.125f);                  // (the time batch's st contribution seeds the accumulators)
    if constexpr (TB == 1) {      // gate (N = 256) and the st half of j1 of the next stage: kept in registers for the next evaluation
      f32x4 acc[3 * CT][1];
#pragma unroll
      for (int t = 0; t < 3 * CT; ++t) acc[t][0] = f32x4{0, 0, 0, 0};
      f32x4 (&ag)[2 * CT][1] = reinterpret_cast<f32x4 (&)[2 * CT][1]>(acc[0]);
      f32x4 (&as)[CT][1] = reinterpret_cast<f32x4 (&)[CT][1]>(acc[2 * CT]);
      job<1, 2 * CT, 8, LDH>(ST + (lane & 15) * LDH + 4 * g, wr, W4 * 4, wave * 2 * CT, lane, ag);
      job<1, CT, 8, LDH>(ST + (lane & 15) * LDH + 4 * g, wr, W5T * 4, wave * CT, lane, as);
#pragma unroll
      for (int t = 0; t < 3 * CT; ++t) accum += acc[t][0][0] * 1e-9f + acc[t][0][3] * 1e-9f;
    }
    __syncthreads();
    hidden(J1, J2, W6 * 4, 3, 0.f);
    __syncthreads();
    {   // out: M = 16, 2 CT column tiles per wave; v = out + gate * clip(grad log pi): the target terms per element
      f32x4 acc[2 * CT][1];
#pragma unroll
      for (int t = 0; t < 2 * CT; ++t) acc[t][0] = f32x4{0, 0, 0, 0};
      job<1, 2 * CT, 8, LDH>(J2 + (lane & 15) * LDH + 4 * g, wr, W7 * 4, wave * 2 * CT, lane, acc);
#pragma unroll
      for (int t = 0; t < 2 * CT; ++t) {
        const int col = (wave * 2 * CT + t) * 16 + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* xr = X + (4 * g + i) * LDX + 4 + col;
          const float x = xr[0], lap = 2.f * x - xr[-1] - xr[1];
          const float graw = -20.f * (0.5f * lap - x * (1.f - x * x) * 2.f);
          const float gc = fminf(fmaxf(graw, -1.f), 1.f);
          const float hv = fabsf(graw) <= 1.f ? -20.f * (0.5f * lap - (1.f - 3.f * x * x) * 2.f) : 0.f;
          const float v = acc[t][0][i] + bs[4][t & (CT - 1)] + 0.37f * gc;
          accum += v * 1e-6f + hv * 1e-7f;
          if (e + 1 < evals) X[(4 * g + i) * LDX + 4 + col] = x + 1e-3f * v;      // the next stage input
        }
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * NW * 64 + threadIdx.x] = accum;
}

template <int NW, bool PIPE, int TB = 0>
static void run(const char* name, const float* dW, const float* dB, float* dout, unsigned long long* dcyc, int evals) {
  const size_t sm = (TB == 2 ? 80 * (LDF + LDH) : 16 * LDX + 4 * 32 * LDH + (TB ? 16 * (LDF + 2 * LDH) : 0)) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)eval_kernel<NW, PIPE, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  std::vector<unsigned long long> h(256);
  double best = 1e30, mean = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((eval_kernel<NW, PIPE, TB>), dim3(256), dim3(NW * 64), sm, 0, dW, dB, evals, dout, dcyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dcyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v; m /= 256.0 * evals;
    if (rep) { best = std::min(best, m); mean += m / 4; }
  }
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)eval_kernel<NW, PIPE, TB>);
  printf("%-34s %8.0f cycles per evaluation (best of 4; mean %8.0f)   registers %d, scratch %zu B   [matrix-pipe floor %d]\n", name, best, mean, fa.numRegs, (size_t)fa.localSizeBytes, TB == 2 ? 61440 : (TB == 1 ? 32768 : 20480));
}

int main() {
  float *dW, *dB, *dout; unsigned long long* dcyc;
  std::vector<float> w(WTOT), b(5 * 256);
  for (int i = 0; i < WTOT; ++i) w[i] = 0.02f * (float)((i * 131) % 97 - 48) / 48.f;
  for (size_t i = 0; i < b.size(); ++i) b[i] = 0.01f * (float)((int)(i % 13) - 6);
  (void)hipMalloc(&dW, WTOT * 4); (void)hipMalloc(&dB, b.size() * 4); (void)hipMalloc(&dout, (size_t)256 * 15 * 512 * 16); (void)hipMalloc(&dcyc, 256 * 8);
  (void)hipMemcpy(dW, w.data(), WTOT * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  const int evals = 400;
  run<8, false>("8 waves x <= 256 registers", dW, dB, dout, dcyc, evals);
  run<4, false>("4 waves x <= 512, two tiles jointly", dW, dB, dout, dcyc, evals);
  run<4, true>("4 waves x <= 512, tile by tile", dW, dB, dout, dcyc, evals);
  run<8, false, 1>("8 waves, next stage's time branch inside", dW, dB, dout, dcyc, evals);
  run<8, false, 2>("8 waves, the M = 80 time batch alone", dW, dB, dout, dcyc, evals);
  printf("an attempt = 6 evaluations + 1 time batch (today), or 5 evaluations with the time branch inside + 1 without\n");
  return 0;
}
