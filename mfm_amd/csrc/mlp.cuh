// VectorFieldNet on CDNA4: layer descriptors, the MFMA-operand-ready ("packed") weight / activation layouts and
// the per-workgroup tile GEMM every MLP kernel is built from.
//
// Network (exe_flow_matching.py:56-90), layers in flax creation order:
//   0 t1 (2F -> ht1)   1 t2 (ht1 -> ht2)   2 x1 (d -> hx1)   3 x2 (hx1 -> hx2)
//   4 gate (ht2 -> d, zero-init)   5 j1 (hx2 + ht2 -> hj1)   6 j2 (hj1 -> hj2)   7 out (hj2 -> d, zero-init)
//   v = out + gate * clip(grad log pi(x))
//
// One workgroup = 4 wavefronts owns a tile of 16 chains (one MFMA M-tile; 32 rows when value + tangent are pushed
// together).  Activations of the tile stay in LDS between layers; weights are streamed L2 -> VGPR once per
// workgroup as ready-made B operands of v_mfma_f32_16x16x4_f32 (exact f32, SURVEY.md "fp32 tolerance").
//
// Packed layouts (all tiles are 16 x 16, lane = 16 g + c, g = lane >> 4, c = lane & 15):
//   weights  W[K][N]  : Wp [nt][kb][lane][s] = W[16 kb + 4 g + s][16 nt + c]     (one float4 per lane per 16 k)
//   transposed (dgrad): WpT[kt][nb][lane][s] = W[16 kt + c][16 nb + 4 g + s]     (= pack of W^T)
//   activations [B][F]: P  [ft][bb][lane][s] = Act[16 bb + 4 g + s][16 ft + c]  (bb = chain block)
// With k(s, g) = 16 kb + 4 g + s the A operand of k-step s is element s of ONE ds_read_b128 from the row-major LDS
// tile, the B operand is element s of ONE coalesced global float4, and the f32 accumulator (col = c,
// row = 4 g + reg) IS the packed activation fragment, so epilogues store it with one float4 per lane and the
// weight-gradient GEMM (reduction over chains) loads both of its operands as float4 with no transposes.
#pragma once
#include "common.cuh"
#include "targets.cuh"

#define MLP_NLAYER 8
#define MLP_WAVES 4
#define MLP_THREADS (MLP_WAVES * 64)
#define MLP_ROWS 16

struct LayerDesc {
  int K, N;        // true sizes
  int Kp, Np;      // padded to 16
  int w_off;       // float offset of the packed forward weights of this layer in NetDev::Wp (and of WpT)
  int b_off;       // float offset of the (padded) bias in NetDev::bias
  int m_w, m_b;    // offsets of kernel / bias in the canonical flat parameter vector (kernel [K][N] row-major, bias [N])
};

struct NetDev {
  int d, dp, F, F2p;
  int ht1, ht2, hx1, hx2, hj1, hj2;   // multiples of 16
  LayerDesc L[MLP_NLAYER];
  int n_params;          // canonical flat size
  int n_packed;          // floats in Wp (= in WpT)
  int n_bias;            // floats in bias
  const float* Wp;
  const float* WpT;
  const float* bias;
  const float* fourier;  // [F]
  float grad_clip;       // 0: no clip  (exe_flow_matching.py:351: gradient_clip if dim > 128 else None)
  TargetDev T;           // UNTEMPERED target for grad log pi (exe_flow_matching.py:351)
};

__host__ __device__ __forceinline__ int pack_index(int k, int n, int KB) {
  int nt = n >> 4, c = n & 15, kb = k >> 4, r = k & 15, g = r >> 2, s = r & 3;
  return (((nt * KB + kb) * 64) + g * 16 + c) * 4 + s;
}
__host__ __device__ __forceinline__ int pack_index_T(int k, int n, int NB) {
  int kt = k >> 4, c = k & 15, nb = n >> 4, r = n & 15, g = r >> 2, s = r & 3;
  return (((kt * NB + nb) * 64) + g * 16 + c) * 4 + s;
}

// ---- the tile GEMM -------------------------------------------------------------------------------------------
// acc[m][j] (+)= A[m-tile rows][K] * W[K][tile nt], for the n-tiles nt = wave + 4 q owned by this wave.
// A: LDS, row-major, MT*16 rows, leading dimension lda (multiple of 4 floats).  Wp: packed weights of the layer.
// epi(q, nt, m, acc): called once per finished tile; acc[i] is (row = 16 m + 4 g + i, col = 16 nt + c).
template <int MT, int NTB, typename Epi>
__device__ __forceinline__ void layer_gemm(const float* A, int lda, const float* __restrict__ Wp_, int KB, int NT,
                                           int wave, int lane, Epi epi) {
  const int r = lane & 15, g = lane >> 4;
  const f32x4* Wp = reinterpret_cast<const f32x4*>(Wp_);
  const float* arow = A + r * lda + 4 * g;
  for (int q0 = 0; wave + 4 * q0 < NT; q0 += NTB) {
    f32x4 acc[MT][NTB];
    const f32x4* wp[NTB];
    bool ok[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
      int nt = wave + 4 * (q0 + j);
      ok[j] = nt < NT;
      wp[j] = Wp + (size_t)(ok[j] ? nt : wave) * KB * 64 + lane;
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // software pipeline: B fragments two k-blocks ahead (ring of 3), A fragment one ahead
    f32x4 bq[3][NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
      bq[0][j] = wp[j][0];
      if (KB > 1) bq[1][j] = wp[j][64];
    }
    for (int kb0 = 0; kb0 < KB; kb0 += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int kb = kb0 + u;
        if (kb < KB) {
          if (kb + 2 < KB) {
#pragma unroll
            for (int j = 0; j < NTB; ++j) bq[(u + 2) % 3][j] = wp[j][(size_t)(kb + 2) * 64];
          }
          f32x4 a[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * lda + kb * 16);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int j = 0; j < NTB; ++j)
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], bq[u][j][s], acc[m][j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NTB; ++j)
      if (ok[j]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) epi(q0 + j, wave + 4 * (q0 + j), m, acc[m][j]);
      }
  }
}

// ---- grad log pi of the (untempered) target on an LDS row with zero pads, clipped ----------------------------
__device__ __forceinline__ float clipf(float v, float c) { return c > 0.f ? fminf(fmaxf(v, -c), c) : v; }
