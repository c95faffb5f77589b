// VectorFieldNet on CDNA4: layer descriptors, the MFMA-operand-ready ("packed") weight / activation layouts and
// the per-workgroup tile GEMM every MLP kernel is built from.
//
// Network (exe_flow_matching.py:56-90), layers in flax creation order:
//   0 t1 (2F -> ht1)   1 t2 (ht1 -> ht2)   2 x1 (d -> hx1)   3 x2 (hx1 -> hx2)
//   4 gate (ht2 -> d, zero-init)   5 j1 (hx2 + ht2 -> hj1)   6 j2 (hj1 -> hj2)   7 out (hj2 -> d, zero-init)
//   v = out + gate * clip(grad log pi(x))
// (hidden lists of another length -- :74-85 loop over them -- keep the order t.., x.., gate, joint.., out: NetDev::nT / nX / nJ;
//  only the wide family, which launches one GEMM per layer, runs them)
//
// One workgroup = 8 wavefronts (2 per SIMD) owns a tile of 16 chains (one MFMA M-tile; 32 rows when value + tangent are pushed
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
#include "common.hip.h"
#include "targets.hip.h"

#define MLP_NLAYER 8              // layers of the network with TWO hidden layers per branch: what the fused tile kernels are written for
#define MLP_MAX_DEPTH 3           // hidden layers per branch the wide family takes (exe_flow_matching.py:74-85 loops over lists of any length)
#define MLP_MAXL (3 * MLP_MAX_DEPTH + 2)
#define MLP_WAVES_FM 8            // waves per workgroup of the flow-matching kernels (2 per SIMD)
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
  int ht1, ht2, hx1, hx2, hj1, hj2;   // multiples of 16 (padded; the true widths are the layers' N): FIRST and LAST hidden width of the t / x / joint branch (the two widths of a two-layer branch)
  LayerDesc L[MLP_MAXL];              // flax creation order: t[0..nT), x[0..nX), gate, joint[0..nJ), out; unused slots: K = N = 0, m_w = m_b = n_params
  int nT, nX, nJ;                     // hidden layers per branch (2, 2, 2: the layer numbering above; anything else: wide family only)
  int n_params;          // canonical flat size
  int n_packed;          // floats in Wp (= in WpT)
  int n_bias;            // floats in bias
  const float* Wp;
  const float* WpT;
  const float* bias;
  const float* fourier;  // [F]
  float grad_clip;       // 0: no clip  (exe_flow_matching.py:351: gradient_clip if dim > 128 else None)
  int act;               // MFM_ACT_*: the hidden non-linearity (exe_flow_matching.py:39-45, multi_modal.py:177)
  TargetDev T;           // UNTEMPERED target for grad log pi (exe_flow_matching.py:351)
};

// ---- hidden non-linearities (exe_flow_matching.py:39-45: jax.nn.relu / tanh / elu / gelu (tanh approximation, the jax
//      default) / swish) with the derivatives the forward-mode tangent (from the pre-activation) and the backward pass
//      (from the stored OUTPUT where the activation is invertible: relu, tanh, elu) need ----
#include "../../include/mfm.h"     // MFM_ACT_*
__device__ __forceinline__ float act_f(float z, int k) {
  if (k == MFM_ACT_RELU) return fmaxf(z, 0.f);
  if (k == MFM_ACT_TANH) return tanhf(z);
  if (k == MFM_ACT_ELU) return z > 0.f ? z : expm1f(z);
  if (k == MFM_ACT_GELU) return 0.5f * z * (1.f + tanhf(0.7978845608028654f * (z + 0.044715f * z * z * z)));
  return z / (1.f + expf(-z));
}
__device__ __forceinline__ float dact_pre(float z, int k) {
  if (k == MFM_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (k == MFM_ACT_TANH) { const float t = tanhf(z); return 1.f - t * t; }
  if (k == MFM_ACT_ELU) return z > 0.f ? 1.f : expf(z);
  if (k == MFM_ACT_GELU) {
    const float c = 0.7978845608028654f, u = c * (z + 0.044715f * z * z * z), t = tanhf(u);
    return 0.5f * (1.f + t) + 0.5f * z * (1.f - t * t) * c * (1.f + 3.f * 0.044715f * z * z);
  }
  const float s = 1.f / (1.f + expf(-z));
  return s * (1.f + z * (1.f - s));
}
__device__ __forceinline__ float dact_out(float y, int k) {      // relu / tanh / elu only (f' as a function of f)
  if (k == MFM_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (k == MFM_ACT_TANH) return 1.f - y * y;
  return y > 0.f ? 1.f : y + 1.f;
}
// tangent / gradient through the activation: relu keeps the exact select of the tuned kernels (no 0 * inf)
__device__ __forceinline__ float mask_pre(float pre, float t, int k) { return k == MFM_ACT_RELU ? (pre > 0.f ? t : 0.f) : t * dact_pre(pre, k); }
__device__ __forceinline__ float mask_out(float y, float t, int k) { return k == MFM_ACT_RELU ? (y > 0.f ? t : 0.f) : t * dact_out(y, k); }

__host__ __device__ __forceinline__ int pack_index(int k, int n, int KB) {
  int nt = n >> 4, c = n & 15, kb = k >> 4, r = k & 15, g = r >> 2, s = r & 3;
  return (((nt * KB + kb) * 64) + g * 16 + c) * 4 + s;
}
__host__ __device__ __forceinline__ int pack_index_T(int k, int n, int NB) {
  int kt = k >> 4, c = k & 15, nb = n >> 4, r = n & 15, g = r >> 2, s = r & 3;
  return (((kt * NB + nb) * 64) + g * 16 + c) * 4 + s;
}

// Row of a layer's PACKED K axis that canonical input row k feeds.  Hidden widths that are not multiples of 16 are zero-padded to one
// (wide family only); the first joint layer reads [sx | st] with BOTH halves padded, so its st rows sit behind the padded sx half.
__host__ __device__ __forceinline__ int packed_row(const NetDev& n, int layer, int k) {
  if (layer != n.nT + n.nX + 1) return k;
  const LayerDesc& lx = n.L[n.nT + n.nX - 1];
  return k < lx.N ? k : k - lx.N + lx.Np;
}

// a hidden width that is not a multiple of 16 (zero-padded: K / N of the layers differ from Kp / Np beyond the d- and 2F-wide edges)
__host__ __device__ __forceinline__ bool net_ragged(const NetDev& n) {
  const int gate = n.nT + n.nX, out = gate + n.nJ + 1;
  for (int l = 0; l < out; ++l)
    if (l != gate && n.L[l].N != n.L[l].Np) return true;
  return false;
}

// ---- the tile GEMM -------------------------------------------------------------------------------------------
// acc[m] (+)= A[m-tile rows][K] * W[K][tile nt], for the n-tiles nt = wave + NW * q owned by this wave (NW waves per workgroup).
// A: LDS, row-major, MT*16 rows, leading dimension lda (multiple of 4 floats).  Wp: packed weights of the layer.
// epi(q, nt, m, acc, b): called once per finished tile; acc[i] is (row = 16 m + 4 g + i, col = 16 nt + c), b the
// layer bias of that column (0 when `bias` is null).
// 8 waves per workgroup = 2 per SIMD: while one wave sits in an epilogue / barrier / load wait the other keeps the
// SIMD's matrix pipe busy; a wave's own chain of dependent MFMAs (40-cycle latency vs 32-cycle issue) is hidden too.
// PIPE selects the weight-streaming schedule: 2 = ping-pong register sets, counted waits, no copies (fastest when the
// kernel has registers to spare: the flow-matching kernels); 1 = single look-ahead set (smaller live range: the solver
// kernels, which keep seven Runge-Kutta stages in registers).
template <int MT, int NW, int PIPE = 2, typename Epi>
__device__ __forceinline__ void layer_gemm(const float* A, int lda, const float* __restrict__ Wp_,
                                           const float* __restrict__ bias, int KB, int NT, int wave, int lane, Epi epi) {
  const int r = lane & 15, g = lane >> 4;
  const float* arow = A + r * lda + 4 * g;
  for (int q = 0; wave + NW * q < NT; ++q) {
    const int nt = wave + NW * q;
    // the tile's packed fragments through a buffer descriptor: the per-lane offset is a loop constant and the k-block a SCALAR
    // offset, so no vector address arithmetic sits between the MFMAs (a vector instruction of the wave's own stream is not
    // free next to its MFMAs: tools/mb/mfma_coissue.hip); the tile index is wave-uniform (made a scalar here: a descriptor
    // built from a per-lane value is loaded under a waterfall loop); reads past the tile's last block return zeros
    const int nt_s = __builtin_amdgcn_readfirstlane(nt), kb_s = __builtin_amdgcn_readfirstlane(KB);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp_) + (size_t)nt_s * kb_s * 256, 0, kb_s * 1024, 0x00020000);
    const int wvo = lane * 16;
    auto wload = [&](int kb) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wvo, kb * 1024, 0)); };
    const float bv = bias ? bias[nt * 16 + (lane & 15)] : 0.f;   // issued ahead of the K loop: its latency hides there
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE == 2) {
    // Software pipeline over groups of 4 k-blocks with two PING-PONG register sets (no register copies at the loop
    // back-edge: a copy of a load destination would wait for that load).  The next group's B fragments are issued
    // before the current group's MFMAs, so the compiler's counted wait (vmcnt(4)) only covers loads issued a whole
    // group earlier.
    auto group = [&](const f32x4 (&bf)[4], int kb0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * lda + (kb0 + u) * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], bf[u][s], acc[m], 0, 0, 0);
      }
    };
    const int KG = KB >> 2;                 // full groups
    f32x4 ba[4], bb[4];
    if (KG > 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) ba[u] = wload(u);
    }
    int gi = 0;
    for (; gi + 1 < KG; gi += 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bb[u] = wload(4 * gi + 4 + u);
      __builtin_amdgcn_sched_barrier(0);
      group(ba, 4 * gi);
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 2 < KG) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ba[u] = wload(4 * gi + 8 + u);
      }
      __builtin_amdgcn_sched_barrier(0);
      group(bb, 4 * gi + 4);
    }
    if (gi < KG) group(ba, 4 * gi);
    for (int kb = 4 * KG; kb < KB; ++kb) {          // tail (KB not a multiple of 4: tiny layers only)
      const f32x4 bf = wload(kb);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + m * 16 * lda + kb * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bf[s], acc[m], 0, 0, 0);
      }
    }
    } else {
    // Software pipeline at 4-k-block granularity with two register sets: the next group's B fragments are issued
    // right AFTER the first use of the current group, so the wait the compiler places before that use (it emits
    // vmcnt(0) across the loop back-edge) only ever covers loads issued a whole group (>= 12 MT MFMAs) earlier.
    f32x4 cur[4], nxt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = wload(u);
    for (int kb0 = 0; kb0 < KB; kb0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kb = kb0 + u;
        if (kb < KB) {
          f32x4 a[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * lda + kb * 16);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], cur[u][s], acc[m], 0, 0, 0);
        }
        if (u == 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            nxt[v] = wload(kb0 + 4 + v);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) cur[v] = nxt[v];
    }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) epi(q, nt, m, acc[m], bv);
  }
}

// ---- the same tile GEMM with the weight stream CHAINED across tiles, layers and barriers -------------------------------------
// A workgroup's eight waves start every tile together, right after a barrier, and each asks for its first two fragment groups
// (8 KB per wave, 64 KB per CU) at once: the CU's 64 B/clk vector L1 needs ~1,000 cycles to deliver them, on top of the L2
// latency, and nothing can be multiplied meanwhile -- 1.3 - 1.6 k cycles per tile against 2.0 - 4.1 k cycles of MFMAs
// (tools/fm_stamps.py --lg).  Here the first group of the NEXT tile is requested while the LAST group of the current tile is
// being multiplied (its register set is free by then), so it streams in under those MFMAs, the epilogue and the barrier.
// Measured in the benchmarked loop (tools/fm_stamps.py --loop): 98.2 k -> 94.8 k cycles per workgroup of the training kernel
// (part of the wait only moves: the tile's SECOND group is now the one requested a bare group of MFMAs ahead); the rocprofv3
// average of the kernel moves by less than a microsecond (45.9 -> 45.0 us together with the streaming stores).
// Requires an even number of 4-k-block groups per tile (K a multiple of 128) and NT >= NW.  `nx`: the packed fragments of the
// tile this wave processes after the call (null: none); `ch` carries the requested group from call to call.
struct WNext { const float* tile; int KB; };
struct WChain { f32x4 b[4]; bool have; };
template <int MT, int NW, typename Epi>
__device__ __forceinline__ void layer_gemm_chain(const float* A, int lda, const float* __restrict__ Wp_, const float* __restrict__ bias,
                                                 int KB, int NT, int wave, int lane, Epi epi, WChain& ch, WNext nx) {
  const int r = lane & 15, g = lane >> 4;
  const float* arow = A + r * lda + 4 * g;
  const int wvo = lane * 16;
  auto desc = [](const float* base, int kb) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, __builtin_amdgcn_readfirstlane(kb) * 1024, 0x00020000);
  };
  for (int q = 0; wave + NW * q < NT; ++q) {
    const int nt = __builtin_amdgcn_readfirstlane(wave + NW * q), kb_s = __builtin_amdgcn_readfirstlane(KB);
    const bool more = nt + NW < NT;
    const float* ntile = more ? Wp_ + (size_t)(nt + NW) * kb_s * 256 : nx.tile;
    const __amdgpu_buffer_rsrc_t wr = desc(Wp_ + (size_t)nt * kb_s * 256, kb_s);
    const __amdgpu_buffer_rsrc_t wn = desc(ntile ? ntile : Wp_, more ? kb_s : nx.KB);
    auto wload = [&](__amdgpu_buffer_rsrc_t d, int kb) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d, wvo, kb * 1024, 0)); };
    const float bv = bias ? bias[nt * 16 + (lane & 15)] : 0.f;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto group = [&](const f32x4 (&bf)[4], int kb0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * lda + (kb0 + u) * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], bf[u][s], acc[m], 0, 0, 0);
      }
    };
    const int KG = KB >> 2;                 // even
    f32x4 bb[4];
    if (!ch.have) {
#pragma unroll
      for (int u = 0; u < 4; ++u) ch.b[u] = wload(wr, u);
    }
    for (int gi = 0; gi < KG; gi += 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bb[u] = wload(wr, 4 * gi + 4 + u);
      __builtin_amdgcn_sched_barrier(0);
      group(ch.b, 4 * gi);
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 2 < KG) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ch.b[u] = wload(wr, 4 * gi + 8 + u);
      } else if (ntile) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ch.b[u] = wload(wn, u);      // the next tile's first group
      }
      __builtin_amdgcn_sched_barrier(0);
      group(bb, 4 * gi + 4);
    }
    ch.have = ntile != nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) epi(q, nt, m, acc[m], bv);
  }
}

// ---- grad log pi of the (untempered) target on an LDS row with zero pads, clipped ----------------------------
__device__ __forceinline__ float clipf(float v, float c) { return c > 0.f ? fminf(fmaxf(v, -c), c) : v; }
