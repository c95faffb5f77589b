// K3 + K4 (+ K9): flow-matching batch construction, vector-field forward, loss, and the data-gradient half of the
// backward pass, fused in ONE kernel per tile of 16 chains; then the weight-gradient GEMMs (reduction over chains)
// as a second kernel on the packed activations the first one leaves behind.
//
// Replaces exe_flow_matching.py:151-178 (cond_flow_fn / flow_fn + flow_matching_loss) and the XLA backward of
// jax.value_and_grad(loss_fn, argnums=2) at :364-365.  K3 is fused into the prologue (no HBM round trip for t, x0,
// eps, cond, target); the loss is the SUM over chains and dims (:178).
#include <type_traits>
#include "mlp.hip.h"
#include "prng.hip.h"


struct WsLayout {            // tile-row offsets (units: NBB * 256 floats) into the packed activation workspaces
  int a_ffat, a_t1, a_st, a_cond, a_x1, a_sx, a_j1, a_j2, a_tiles;
  int z_t1, z_t2, z_x1, z_x2, z_gate, z_j1, z_j2, z_out, z_tiles;
};

// The MALA step of the same iteration, run by the training kernel's workgroup on ITS 16 chains before it builds their batch
// (mfm_train_iter on one rank: exe_flow_matching.py:300-314 directly followed by :362-368; mala.hip: mala_chain_step)
struct FmMala {
  int on, textbook;
  Key2 key;
  double beta, eps;
  double* logp; float* grad; float* acc_prob;      // chain state beside `pos` (updated in place), acceptance probability (may be null)
  const draw_t* pre_n; const double* pre_u;        // draws produced ahead of time (noise.hip), or null
};

struct FmArgs {
  NetDev net;
  WsLayout ws;
  Key2 key_time, key_ref, key_gauss;
  uint32_t n_total, chain_offset;
  int B;                 // samples handled by this launch (multiple of 16)
  int n_valid;           // rows >= n_valid are padding of the chain shard: residual 0, i.e. no loss and no gradient (B: none)
  float sigma;
  int cond_flow;
  const float* pos;      // [B][d] samples x1
  float* acts;           // packed activations (TRAIN only)
  float* dzs;            // packed pre-activation gradients (TRAIN only)
  float* dacts;          // gelu / swish (TRAIN only): f'(pre-activation) of every hidden unit, packed like `acts` -- their backward
                         // pass cannot recover f' from the stored output (relu, tanh, elu can), and LDS has no room for it
  double* loss_part;     // [gridDim.x] partial sums of squared residuals
  double ref_std;        // reference distribution of the flow: x0 = ref_std * normal (IndepGaussian(dim, var), distributions.py:93-97)
  const draw_t* pre_x0; const draw_t* pre_eps; const float* pre_t;   // non-null: the batch's draws, produced ahead of time by noise_kernel
  int stagger_cycles;    // fm_eval_kernel<2, .>: start delay of the second workgroup of every CU (0: none)
  FmMala mala;           // fm_fwd_bwd_kernel<.., MALA = true> only
  int* flags_clear;      // non-null (TRAIN): flag words [0], [3], [4] of the optimizer's scratch, cleared here for the weight-gradient kernel
                         // and the one-launch reduction + optimizer that follow (optim.hip: reduce_adamw_kernel)
  int* sus_set;          // non-null (TRAIN): raised when a value stored for the weight-gradient kernel is not <= FM_SAFE in magnitude (NaN included):
  int* sus_clear;        // clear, no sum of <= 2^20 of their products overflows (wgrad_sk.hip).  sus_clear: the OTHER iteration parity's word, reset here
};
constexpr float FM_SAFE = 1.0e15f;

struct FmLds {           // float offsets into dynamic LDS
  int ff, ldff, x, ldx, t1, ldt1, cat, ldcat, x1, ldx1, j1, ldj1, j2, ldj2, g, ldg;
  int dv, lddv, d1, ldd1, d2, ldd2, dcat, gcs, red, gc, total;
};

__host__ __device__ inline FmLds fm_lds_layout(const NetDev& n, bool train) {
  FmLds L;
  int o = 0;
  auto take = [&](int rows, int ld) { int r = o; o += rows * ld; return r; };
  // leading dimensions = 8 (mod 64) dwords for the usual widths (multiples of 64): the 16-lane groups of a ds_read_b128
  // A-fragment read (row = lane & 15, k offset 4 (lane >> 4)) then touch 64 distinct banks; K + 4 put two lanes of every
  // group on the same four banks (2-way conflict on every A read: 1.44e6 conflict cycles per launch in the round-1 profile)
  L.ldff = n.F2p + 8;            L.ff = take(16, L.ldff);
  L.ldx = n.dp + 8;              L.x = take(16, L.ldx);          // data starts at col 4: x[-1] and x[d] pads exist
  L.ldt1 = n.ht1 + 8;            L.t1 = take(16, L.ldt1);
  L.ldcat = n.hx2 + n.ht2 + 8;   L.cat = take(16, L.ldcat);
  L.ldx1 = n.hx1 + 8;            L.x1 = take(16, L.ldx1);
  L.ldj1 = n.hj1 + 8;            L.j1 = take(16, L.ldj1);
  L.ldj2 = n.hj2 + 8;            L.j2 = take(16, L.ldj2);
  L.ldg = n.dp + 8;              L.g = take(16, L.ldg);
  L.lddv = n.dp + 8; L.ldd1 = n.hj2 + 8; L.ldd2 = n.hj1 + 8;
  L.dv = L.d1 = L.d2 = L.dcat = 0;
  if (train) {
    L.dv = take(16, L.lddv);
    L.d1 = take(16, L.ldd1);
    L.d2 = take(16, L.ldd2);
    L.dcat = take(16, L.ldcat);
  }
  L.gcs = take(16, 8);
  L.red = take(1, 32);
  L.gc = n.T.kind == MFM_TARGET_LGCP ? take(16, L.ldg) : 0;      // grad log pi of the tile (needs the K^-1 GEMM)
  L.total = o;
  return L;
}

// Streaming stores: the packed activations (44 MB per training step at the headline shape) are read next by ANOTHER kernel; written
// with the default policy they push the network's weights out of the XCD's 4 MB L2 while this kernel is still reading them
// (rocprofv3, 440 launches each: training kernel 46.8 -> 45.0 us, the weight-gradient kernel that reads them 26.0 -> 26.5 us).
// Measured and dropped next to it: pulling the weights into each XCD's L2 at kernel start (one dword per 128-byte line, shared
// out over the XCD's workgroups): 44.6 us without against 45.0 with.
__device__ __forceinline__ void store_packed(float* base, int tile_row, int nbb, int bb, int lane, f32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base) + ((size_t)tile_row * nbb + bb) * 64 + lane);
}

// grad log pi(x)[row][col], clipped, for the tile whose positions sit in LDS `xrow0` (row stride ldx, data at +4)
__device__ __forceinline__ float target_gclip(const NetDev& n, const float* xbuf, int ldx, const float* gcs, const float* gcl, int ldg,
                                              int row, int col) {
  float gv;
  if (n.T.kind == MFM_TARGET_PHI4) gv = phi4_grad(n.T, xbuf + row * ldx + 4, col);
  else if (n.T.kind == MFM_TARGET_LGCP) gv = gcl[row * ldg + col];
  else gv = gcs[row * 8 + col];
  return clipf(gv, n.grad_clip);
}

// STATIC: the headline configuration (F = 128, every hidden width 128, relu, phi-four d = 256: multi_modal.py:156,177-180)
// with its dimensions, activation and target kind as compile-time constants -- loop bounds, tile counts and the LDS layout fold, the K loops
// unroll; offsets into the parameter buffers stay the host's.  Same arithmetic as the generic instance (the compiler's multiply-add
// contraction may differ in the last bits).
// MALA: the workgroup first advances its 16 chains by one MALA step (two chains per wave, mala_chain_step: the arithmetic of
// mala_step_kernel, bit for bit) and builds the batch from the new positions left in LDS -- one launch less per iteration, and
// the positions do not travel through HBM between the two.
// The six leading pointer arguments repeat fields of `a` (first layer's packed weights; position, gradient, prefetched MALA draws,
// log-density and prefetched uniforms of the chains): scalar arguments at the head of the list are preloaded into SGPRs (-amdgpu-kernarg-preload-count,
// mfm_amd/build.py), so the first loads of the prologue -- the weight prefetch and the MALA step's -- go out before the read of the
// 850-byte argument struct has returned.
template <int TPW, bool TRAIN, bool STATIC = false, int ACT = -1, bool MALA = false>      // ACT >= 0: the activation as a compile-time constant (the five-way
__global__ __launch_bounds__((MLP_WAVES_FM * 64)) void fm_fwd_bwd_kernel(const float* pl_w0, const float* pl_pos, const float* pl_grad, const draw_t* pl_pren,
                                                                         const double* pl_logp, const double* pl_preu, FmArgs a) {      // run-time selection in every epilogue triples the code)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  NetDev nloc = a.net;
  if constexpr (ACT >= 0) nloc.act = ACT;
  if constexpr (STATIC) {
    nloc.d = 256; nloc.dp = 256; nloc.F = 128; nloc.F2p = 256;
    nloc.ht1 = nloc.ht2 = nloc.hx1 = nloc.hx2 = nloc.hj1 = nloc.hj2 = 128;
    constexpr int K_[MLP_NLAYER] = {256, 128, 256, 128, 128, 256, 128, 128}, N_[MLP_NLAYER] = {128, 128, 128, 128, 256, 128, 128, 256};
#pragma unroll
    for (int l = 0; l < MLP_NLAYER; ++l) { nloc.L[l].K = nloc.L[l].Kp = K_[l]; nloc.L[l].N = nloc.L[l].Np = N_[l]; }
    nloc.act = MFM_ACT_RELU; nloc.T.kind = MFM_TARGET_PHI4;      // the epilogues' activation / target selections fold too
  }
  const NetDev& n = nloc;
  const FmLds L = fm_lds_layout(n, TRAIN);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  WChain wch; wch.have = false;
  if constexpr (STATIC) {      // the first tile's first fragment group: requested before everything else -- from a preloaded pointer,
                               // ahead of the first read of the argument struct --, it arrives under the prologue
    const __amdgpu_buffer_rsrc_t w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pl_w0) + (size_t)wave * (n.L[0].Kp / 16) * 256, 0,
                                                                        (n.L[0].Kp / 16) * 1024, 0x00020000);
#pragma unroll
    for (int u = 0; u < 4; ++u) wch.b[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w0, lane * 16, u * 1024, 0));
    wch.have = true;
  }
  __builtin_amdgcn_sched_barrier(0);
  const int bb = blockIdx.x, b0 = bb * 16, nbb = a.B / 16;
  const int d = n.d;

  float* bFF = lds + L.ff;  float* bX = lds + L.x;   float* bT1 = lds + L.t1; float* bCat = lds + L.cat;
  float* bX1 = lds + L.x1;  float* bJ1 = lds + L.j1; float* bJ2 = lds + L.j2; float* bG = lds + L.g;
  float* bDV = lds + L.dv;  float* bD1 = lds + L.d1; float* bD2 = lds + L.d2; float* bDC = lds + L.dcat;
  float* gcs = lds + L.gcs; double* red = reinterpret_cast<double*>(lds + L.red); float* bGC = lds + L.gc;

  if (TRAIN && a.flags_clear && blockIdx.x == 0 && threadIdx.x == 0) { a.flags_clear[0] = 0; a.flags_clear[3] = 0; a.flags_clear[4] = 0; if (a.sus_clear) *a.sus_clear = 0; }
  // running sum of squares of everything this lane stores for the weight-gradient kernel: NaN and infinity propagate, an overflow of the
  // sum itself only errs on the safe side; one comparison at the end of the kernel (FmArgs::sus_set)
  float sq_acc = 0.f;
  auto store_chk = [&](float* base, int tile_row, f32x4 v) {
    store_packed(base, tile_row, nbb, bb, lane, v);
    sq_acc = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[3], v[3], sq_acc))));
  };
  FM_STAMP(0);
  // ---------------- prologue: K3 batch construction (exe_flow_matching.py:151-169 / :139-147) ----------------
  // draws produced ahead of time (noise.hip): every load of the tile is issued in one go, well before the first use (behind the
  // per-element `drawn ? load : threefry + erfinv` selection each load used to wait for the previous one: 24 HBM round trips) --
  // with the MALA step in this kernel, right behind that step's own loads (mala.hip: after_loads)
  const bool drawn = a.cond_flow && a.pre_x0;
  draw_t x0d[TPW][4], ned[TPW][4]; float x1f[TPW][4], tpre[4] = {0.f, 0.f, 0.f, 0.f};
  auto issue_batch_loads = [&]() {
    if (drawn) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int nt = wave + MLP_WAVES_FM * q, col = nt * 16 + c;
        const bool live = nt * 16 < n.dp && col < d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const size_t po = (size_t)(b0 + 4 * g + i) * d + (live ? col : 0);
          x0d[q][i] = a.pre_x0[po]; ned[q][i] = a.pre_eps[po];
          if constexpr (!MALA) x1f[q][i] = a.pos[po];
        }
      }
    }
  };
  // The times of the tile's rows and everything that depends on them alone -- the Fourier features (:70-71) -- come FIRST: their
  // loads (64 B of times, one frequency per lane) are issued ahead of the MALA step's and the batch's, and the features are formed
  // while those are in flight (`early_work`: the hook of mala_chain_step, or straight away); the zero pads of the x buffer too.
  if (a.pre_t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tpre[i] = a.pre_t[b0 + 4 * g + i];
  }
  // the wave's first frequency tile: an UNCONDITIONAL load kept in its storage type until `early_work` (behind a branch, or converted
  // here, the compiler waits for it -- and for the weight fragments requested above -- on the spot: s_waitcnt vmcnt(0) before any
  // other load of the prologue has been issued)
  const float f_first_raw = n.fourier[(wave * 16 + c) < n.F ? wave * 16 + c : 0];
  float tt[4];
  Key2 kref[4];
  uint32_t bglob[4];
  auto early_work = [&]() {
    issue_batch_loads();
    for (int i = threadIdx.x; i < 16 * L.ldx; i += (MLP_WAVES_FM * 64)) bX[i] = 0.f;      // pads (incl. x[-1], x[d..])
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bglob[i] = a.chain_offset + (uint32_t)(b0 + 4 * g + i);
      if (a.pre_t) { tt[i] = tpre[i]; kref[i] = Key2{0, 0}; continue; }
      tt[i] = (float)uniform01(a.key_time, bglob[i], a.n_total);                  // :154 / :142
      kref[i] = split_at(a.key_ref, a.n_total, bglob[i]);                         // :155
    }
    // Fourier features of t (:70-71): cos block then sin block
    if (n.F % 16 == 0) {          // tile-aligned halves: one sincos per (row, frequency) feeds both
      const int FT = n.F / 16;
      for (int nt = wave; nt < FT; nt += MLP_WAVES_FM) {
        const int col = nt * 16 + c;
        const double f = nt == wave ? (double)f_first_raw : (double)n.fourier[col];
        f32x4 cs, sn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          double ft = f * (double)tt[i];
          ft -= rint(ft);
          float sv, cvv;
          sincospif(2.f * (float)ft, &sv, &cvv);
          cs[i] = cvv; sn[i] = sv;
          bFF[(4 * g + i) * L.ldff + col] = cvv;
          bFF[(4 * g + i) * L.ldff + n.F + col] = sv;
        }
        if (TRAIN) {
          store_chk(a.acts, a.ws.a_ffat + nt, cs);
          store_chk(a.acts, a.ws.a_ffat + FT + nt, sn);
        }
      }
    } else {
      for (int nt = wave; nt * 16 < n.F2p; nt += MLP_WAVES_FM) {
        const int col = nt * 16 + c;
        f32x4 fv = {0.f, 0.f, 0.f, 0.f};
        if (col < 2 * n.F) {
          const bool is_sin = col >= n.F;
          const double f = n.fourier[is_sin ? col - n.F : col];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            double ft = f * (double)tt[i];
            ft -= rint(ft);
            float sv, cvv;
            sincospif(2.f * (float)ft, &sv, &cvv);
            fv[i] = is_sin ? sv : cvv;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bFF[(4 * g + i) * L.ldff + col] = fv[i];
        if (TRAIN) store_chk(a.acts, a.ws.a_ffat + nt, fv);
      }
    }
  };
  if constexpr (!MALA) early_work();
  if constexpr (MALA) {
    // rows `wave` and `wave + 8` of the tile; their LDS rows live in the dv buffer (first written by the output layer's epilogue)
    MalaArgs m;
    m.T = n.T; m.T.dim = d; m.key = a.mala.key; m.keys = nullptr; m.n_total = a.n_total; m.chain_offset = a.chain_offset; m.B = a.B;
    m.beta = a.mala.beta; m.eps = a.mala.eps; m.textbook = a.mala.textbook;
    m.pos = const_cast<float*>(pl_pos); m.logp = const_cast<double*>(pl_logp); m.grad = const_cast<float*>(pl_grad);
    m.acc_prob = a.mala.acc_prob; m.accepted = nullptr; m.proposed = nullptr; m.prop_weight = nullptr;
    m.pre_n = pl_pren; m.pre_u = pl_preu;
    const int bs[2] = {b0 + wave, b0 + wave + 8};
    float* const xs[2] = {bDV + wave * L.lddv + 4, bDV + (wave + 8) * L.lddv + 4};
    float* const gsm[2] = {gcs + wave * 8, gcs + (wave + 8) * 8};
    FM_STAMP(6);
    mala_chain_step<2 * TPW, 2>(m, bs, xs, gsm, lane, early_work);
    FM_STAMP(7);
  }
  __syncthreads();
  FM_STAMP(8);
  // x1 of (tile row, column): the chain's position -- from HBM, or what the MALA step above left in LDS
  auto x1_at = [&](int row, int col) -> float { return MALA ? bDV[row * L.lddv + 4 + col] : a.pos[(size_t)(b0 + row) * d + col]; };
  float tgt[TPW][4];
  if (drawn) {
    if constexpr (MALA) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int nt = wave + MLP_WAVES_FM * q, col = nt * 16 + c;
        const bool live = nt * 16 < n.dp && col < d;
#pragma unroll
        for (int i = 0; i < 4; ++i) x1f[q][i] = x1_at(4 * g + i, live ? col : 0);
      }
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int nt = wave + MLP_WAVES_FM * q, col = nt * 16 + c;
      const bool live = nt * 16 < n.dp && col < d;
      f32x4 cv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tgt[q][i] = 0.f;
        if (live) {
          const double x1v = x1f[q][i], t = tt[i], x0 = a.ref_std * (double)x0d[q][i];
          cv[i] = (float)((double)a.sigma * (double)ned[q][i] + t * x1v + (1.0 - t) * x0);      // :167
          tgt[q][i] = (float)(x1v - x0);                                              // :168
          bX[(4 * g + i) * L.ldx + 4 + col] = cv[i];
        }
      }
      if (TRAIN && nt * 16 < n.dp) store_chk(a.acts, a.ws.a_cond + nt, cv);
    }
  } else
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int nt = wave + MLP_WAVES_FM * q, col = nt * 16 + c;
    f32x4 cv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tgt[q][i] = 0.f;
      if (nt * 16 < n.dp && col < d) {
        const int row = 4 * g + i;
        const double x1v = x1_at(row, col);
        const double t = tt[i];
        double cnd, tg;
        if (a.cond_flow) {
          const size_t po = (size_t)(b0 + row) * d + col;
          const double x0 = a.ref_std * (double)(a.pre_x0 ? a.pre_x0[po] : (draw_t)normal64(kref[i], (uint32_t)col, (uint32_t)d));
          const double ne = (double)(a.pre_x0 ? a.pre_eps[po] : (draw_t)normal64(a.key_gauss, bglob[i] * (uint32_t)d + (uint32_t)col, a.n_total * (uint32_t)d));  // :166
          cnd = (double)a.sigma * ne + t * x1v + (1.0 - t) * x0;               // :167
          tg = x1v - x0;                                                       // :168
        } else {
          const double x0 = normal64(a.key_ref, bglob[i] * (uint32_t)d + (uint32_t)col, a.n_total * (uint32_t)d);   // :143
          const double sds = 1.0 - (1.0 - (double)a.sigma) * t;                // :144
          cnd = t * x1v + sds * x0;                                            // :145
          tg = x1v - (1.0 - (double)a.sigma) * x0;                             // :146
        }
        cv[i] = (float)cnd;
        tgt[q][i] = (float)tg;
        bX[row * L.ldx + 4 + col] = cv[i];
      }
    }
    if (TRAIN && nt * 16 < n.dp) store_chk(a.acts, a.ws.a_cond + nt, cv);
  }
  FM_STAMP(1);
  __syncthreads();
  if (n.T.kind == MFM_TARGET_GMM && (n.T.n_modes <= 16 ? threadIdx.x < 256 : threadIdx.x < 16)) {      // one mode per lane (targets.hip.h)
    double lp; float gg[8];
    const int row = n.T.n_modes <= 16 ? (int)(threadIdx.x >> 4) : (int)threadIdx.x;
    if (n.T.n_modes <= 16) gmm_eval_lanes16<8>(n.T, bX + row * L.ldx + 4, threadIdx.x & 15, &lp, gg);
    else gmm_eval<8>(n.T, bX + row * L.ldx + 4, &lp, gg);
    if (n.T.n_modes > 16 || (threadIdx.x & 15) == 0)
      for (int j = 0; j < d; ++j) gcs[row * 8 + j] = gg[j];
  }

  FM_STAMP(2);
  // ---------------- forward ----------------------------------------------------------------------------------
  auto relu_store = [&](const LayerDesc& ld, float* out, int ldo, int coff, int a_tile) {
    return [&, out, ldo, coff, a_tile](int q, int nt, int m, f32x4 acc, float bias) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = act_f(acc[i] + bias, n.act);
        out[(4 * g + i) * ldo + coff + nt * 16 + c] = v[i];
      }
      if (TRAIN) store_chk(a.acts, a_tile + nt, v);
      if (TRAIN && n.act >= MFM_ACT_GELU) {          // the derivative the backward epilogue of this tile will need (same lane, same slots)
        f32x4 dv;
#pragma unroll
        for (int i = 0; i < 4; ++i) dv[i] = dact_pre(acc[i] + bias, n.act);
        store_packed(a.dacts, a_tile + nt, nbb, bb, lane, dv);
      }
    };
  };
  // the network's 14 layer GEMMs: on the headline shape with the weight stream chained from tile to tile (mlp.hip.h: layer_gemm_chain)
  auto LG = [&](const float* A, int lda, const float* W, const float* bias, int KB, int NT, auto epi, const float* Wnext, int KBnext) {
    if constexpr (STATIC)
      layer_gemm_chain<1, MLP_WAVES_FM>(A, lda, W, bias, KB, NT, wave, lane, epi, wch, WNext{Wnext ? Wnext + (size_t)wave * KBnext * 256 : nullptr, KBnext});
    else
      layer_gemm<1, MLP_WAVES_FM>(A, lda, W, bias, KB, NT, wave, lane, epi);
  };
  if (n.T.kind == MFM_TARGET_LGCP)      // grad log pi(cond) = c - a exp(cond) - K^-1 (cond - mu)
    layer_gemm<1, MLP_WAVES_FM>(bX + 4, L.ldx, n.T.KinvP, n.T.kbias, n.dp / 16, n.dp / 16, wave, lane,
                                [&](int q, int nt, int m, f32x4 acc, float kb) {
                                  const int col = nt * 16 + c;
#pragma unroll
                                  for (int i = 0; i < 4; ++i) {
                                    const int row = 4 * g + i;
                                    const float xv = bX[row * L.ldx + 4 + col];
                                    bGC[row * L.ldg + col] = col < d ? n.T.counts[col] - n.T.poisson_a * expf(xv) - (acc[i] + kb) : 0.f;
                                  }
                                });
  LG(bFF, L.ldff, n.Wp + n.L[0].w_off, n.bias + n.L[0].b_off, n.L[0].Kp / 16, n.L[0].Np / 16,
                   relu_store(n.L[0], bT1, L.ldt1, 0, a.ws.a_t1), n.Wp + n.L[2].w_off, n.L[2].Kp / 16);
  LG(bX + 4, L.ldx, n.Wp + n.L[2].w_off, n.bias + n.L[2].b_off, n.L[2].Kp / 16, n.L[2].Np / 16,
                   relu_store(n.L[2], bX1, L.ldx1, 0, a.ws.a_x1), n.Wp + n.L[1].w_off, n.L[1].Kp / 16);
  __syncthreads();
  LG(bT1, L.ldt1, n.Wp + n.L[1].w_off, n.bias + n.L[1].b_off, n.L[1].Kp / 16, n.L[1].Np / 16,
                   relu_store(n.L[1], bCat, L.ldcat, n.hx2, a.ws.a_st), n.Wp + n.L[3].w_off, n.L[3].Kp / 16);
  LG(bX1, L.ldx1, n.Wp + n.L[3].w_off, n.bias + n.L[3].b_off, n.L[3].Kp / 16, n.L[3].Np / 16,
                   relu_store(n.L[3], bCat, L.ldcat, 0, a.ws.a_sx), n.Wp + n.L[4].w_off, n.L[4].Kp / 16);
  __syncthreads();
  LG(bCat + n.hx2, L.ldcat, n.Wp + n.L[4].w_off, n.bias + n.L[4].b_off, n.L[4].Kp / 16, n.L[4].Np / 16,
                   [&](int q, int nt, int m, f32x4 acc, float bias) {
#pragma unroll
                     for (int i = 0; i < 4; ++i) bG[(4 * g + i) * L.ldg + nt * 16 + c] = acc[i] + bias;
                   }, n.Wp + n.L[5].w_off, n.L[5].Kp / 16);
  LG(bCat, L.ldcat, n.Wp + n.L[5].w_off, n.bias + n.L[5].b_off, n.L[5].Kp / 16, n.L[5].Np / 16,
                   relu_store(n.L[5], bJ1, L.ldj1, 0, a.ws.a_j1), n.Wp + n.L[6].w_off, n.L[6].Kp / 16);
  __syncthreads();
  LG(bJ1, L.ldj1, n.Wp + n.L[6].w_off, n.bias + n.L[6].b_off, n.L[6].Kp / 16, n.L[6].Np / 16,
                   relu_store(n.L[6], bJ2, L.ldj2, 0, a.ws.a_j2), n.Wp + n.L[7].w_off, n.L[7].Kp / 16);
  __syncthreads();
  FM_STAMP(3);
  // output layer + loss (:88-90, :177-178); dv = 2 (v - target), dgate = dv * clip(grad log pi)
  float loss_loc = 0.f;
  LG(bJ2, L.ldj2, n.Wp + n.L[7].w_off, n.bias + n.L[7].b_off, n.L[7].Kp / 16, n.L[7].Np / 16,
                   [&](int q, int nt, int m, f32x4 acc, float bias) {
                     const int col = nt * 16 + c;
                     f32x4 dv = {0.f, 0.f, 0.f, 0.f}, dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                     for (int i = 0; i < 4; ++i) {
                       const int row = 4 * g + i;
                       if (col < d) {
                         const float gc = target_gclip(n, bX, L.ldx, gcs, bGC, L.ldg, row, col);
                         const float v = acc[i] + bias + bG[row * L.ldg + col] * gc;
                         // tgt is indexed by the static slot q: select without dynamic register indexing
                         float tg = 0.f;
#pragma unroll
                         for (int qq = 0; qq < TPW; ++qq) tg = (qq == q) ? tgt[qq][i] : tg;
                         const float r = b0 + row < a.n_valid ? v - tg : 0.f;
                         loss_loc += r * r;
                         dv[i] = 2.f * r;
                         dg[i] = dv[i] * gc;
                       }
                       if (TRAIN) { bDV[row * L.lddv + col] = dv[i]; bG[row * L.ldg + col] = dg[i]; }
                     }
                     if (TRAIN) {
                       store_chk(a.dzs, a.ws.z_out + nt, dv);
                       store_chk(a.dzs, a.ws.z_gate + nt, dg);
                     }
                   }, TRAIN ? n.WpT + n.L[7].w_off : nullptr, n.L[7].Np / 16);
  // the tile's loss: per-wave sums now (forward only), or after the backward pass (training: the shuffles, the barrier's single-lane
  // tail and the store leave the path between the output layer and the first data-gradient GEMM)
  auto loss_total = [&]() {
    const double lw = wave_sum((double)loss_loc);
    if (lane == 0) red[wave] = lw;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < MLP_WAVES_FM; ++w) tot += red[w];
      a.loss_part[blockIdx.x] = tot;
    }
  };
  if (!TRAIN) { loss_total(); FM_STAMP(4); return; }
  __syncthreads();      // dv / dgate of every wave are in LDS
  FM_STAMP(4);

  // ---------------- backward (data gradients only; weight gradients: wgrad_kernel) -----------------------------
  // tangent of the loss through a hidden activation: from the stored OUTPUT (relu / tanh / elu), or times the derivative the
  // forward epilogue of the same tile left in `dacts` (gelu / swish)
  auto dmask = [&](float y, float t, const f32x4& dd, int i) { return n.act >= MFM_ACT_GELU ? t * dd[i] : mask_out(y, t, n.act); };
  auto dload = [&](int a_tile, int nt) {
    return n.act >= MFM_ACT_GELU ? reinterpret_cast<const f32x4*>(a.dacts)[((size_t)(a_tile + nt) * nbb + bb) * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // d j2
  LG(bDV, L.lddv, n.WpT + n.L[7].w_off, nullptr, n.L[7].Np / 16, n.L[7].Kp / 16,
                   [&](int q, int nt, int m, f32x4 acc, float) {
                     f32x4 z;
                     const f32x4 dd = dload(a.ws.a_j2, nt);
#pragma unroll
                     for (int i = 0; i < 4; ++i) {
                       const int row = 4 * g + i, col = nt * 16 + c;
                       z[i] = dmask(bJ2[row * L.ldj2 + col], acc[i], dd, i);
                       bD1[row * L.ldd1 + col] = z[i];
                     }
                     store_chk(a.dzs, a.ws.z_j2 + nt, z);
                   }, n.WpT + n.L[6].w_off, n.L[6].Np / 16);
  __syncthreads();
  // d j1
  LG(bD1, L.ldd1, n.WpT + n.L[6].w_off, nullptr, n.L[6].Np / 16, n.L[6].Kp / 16,
                   [&](int q, int nt, int m, f32x4 acc, float) {
                     f32x4 z;
                     const f32x4 dd = dload(a.ws.a_j1, nt);
#pragma unroll
                     for (int i = 0; i < 4; ++i) {
                       const int row = 4 * g + i, col = nt * 16 + c;
                       z[i] = dmask(bJ1[row * L.ldj1 + col], acc[i], dd, i);
                       bD2[row * L.ldd2 + col] = z[i];
                     }
                     store_chk(a.dzs, a.ws.z_j1 + nt, z);
                   }, n.WpT + n.L[5].w_off, n.L[5].Np / 16);
  __syncthreads();
  // d [sx | st] through j1; the sx half is finished here (-> dz of x2), the st half waits for the gate path
  LG(bD2, L.ldd2, n.WpT + n.L[5].w_off, nullptr, n.L[5].Np / 16, n.L[5].Kp / 16,
                   [&](int q, int nt, int m, f32x4 acc, float) {
                     const int col = nt * 16 + c;
                     const bool is_sx = col < n.hx2;
                     f32x4 z;
                     const f32x4 dd = is_sx ? dload(a.ws.a_sx, nt) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                     for (int i = 0; i < 4; ++i) {
                       const int row = 4 * g + i;
                       z[i] = is_sx ? dmask(bCat[row * L.ldcat + col], acc[i], dd, i) : acc[i];
                       bDC[row * L.ldcat + col] = z[i];
                     }
                     if (is_sx) store_chk(a.dzs, a.ws.z_x2 + nt, z);
                   }, n.WpT + n.L[4].w_off, n.L[4].Np / 16);
  __syncthreads();
  // d st += dgate . W_gate^T ; then relu mask -> dz of t2
  LG(bG, L.ldg, n.WpT + n.L[4].w_off, nullptr, n.L[4].Np / 16, n.L[4].Kp / 16,
                   [&](int q, int nt, int m, f32x4 acc, float) {
                     f32x4 z;
                     const f32x4 dd = dload(a.ws.a_st, nt);
#pragma unroll
                     for (int i = 0; i < 4; ++i) {
                       const int row = 4 * g + i, col = n.hx2 + nt * 16 + c;
                       const float ds = acc[i] + bDC[row * L.ldcat + col];
                       z[i] = dmask(bCat[row * L.ldcat + col], ds, dd, i);
                       bDC[row * L.ldcat + col] = z[i];
                     }
                     store_chk(a.dzs, a.ws.z_t2 + nt, z);
                   }, n.WpT + n.L[3].w_off, n.L[3].Np / 16);
  __syncthreads();
  // d x1 (only needed by wgrad) and d t1
  LG(bDC, L.ldcat, n.WpT + n.L[3].w_off, nullptr, n.L[3].Np / 16, n.L[3].Kp / 16,
                   [&](int q, int nt, int m, f32x4 acc, float) {
                     f32x4 z;
                     const f32x4 dd = dload(a.ws.a_x1, nt);
#pragma unroll
                     for (int i = 0; i < 4; ++i) z[i] = dmask(bX1[(4 * g + i) * L.ldx1 + nt * 16 + c], acc[i], dd, i);
                     store_chk(a.dzs, a.ws.z_x1 + nt, z);
                   }, n.WpT + n.L[1].w_off, n.L[1].Np / 16);
  LG(bDC + n.hx2, L.ldcat, n.WpT + n.L[1].w_off, nullptr, n.L[1].Np / 16, n.L[1].Kp / 16,
                   [&](int q, int nt, int m, f32x4 acc, float) {
                     f32x4 z;
                     const f32x4 dd = dload(a.ws.a_t1, nt);
#pragma unroll
                     for (int i = 0; i < 4; ++i) z[i] = dmask(bT1[(4 * g + i) * L.ldt1 + nt * 16 + c], acc[i], dd, i);
                     store_chk(a.dzs, a.ws.z_t1 + nt, z);
                   }, nullptr, 0);
  if (a.sus_set && __ballot(!(sq_acc <= FM_SAFE * FM_SAFE)) != 0ull && lane == 0) atomicOr(a.sus_set, 1);
  loss_total();
  FM_STAMP(5);
}

// ---- K9: eval_step's loss on LARGE sample sets of a low-dimensional target (exe_flow_matching.py:370-374) -------------------
// The mixture examples evaluate the flow-matching loss on eval_iter * num_chain = 409,600 exact samples EVERY iteration
// (94.6 GFLOP forward: it dominates their iteration).  One workgroup per 16 samples streams the
// whole network (461 KB) per 16 rows: 11.8 GB of L2 -> CU weight traffic per call, the 16-row kernel's bound (2.24 ms = 27 % of
// the f32-MFMA peak).  This forward-only kernel takes 64 samples per workgroup -- four MFMA row tiles per streamed weight
// fragment, a quarter of the weight traffic -- and aliases the layer buffers (Fourier features / concatenated branch outputs,
// t1 / j1, x1 / j2) so that 64 rows fit in LDS.  Same per-sample arithmetic as fm_fwd_bwd_kernel<.., TRAIN = false> (same
// k-order per accumulator); the per-workgroup loss partial covers 64 samples instead of 16.  For dp <= 16 (the mixtures).
struct FmEvalLds { int a, lda, b, ldb, c, ldc, x, ldx, g, ldg, tgt, tt, gcs, red, gmm, total; };
// the same draw as a CALL: for paths that are compiled into a kernel but do not carry its time (the eval kernel's general batch
// construction beside the two-threads-per-element one the mixtures take), so that they do not carry its code size either
__device__ __attribute__((noinline)) double normal64_call(Key2 key, uint32_t idx, uint32_t size) { return normal64(key, idx, size); }
__host__ __device__ inline FmEvalLds fm_eval_lds_layout(const NetDev& n, int R = 64) {      // R: samples per workgroup
  FmEvalLds L; int o = 0;
  auto take = [&](int cnt) { int r = o; o += cnt; return r; };
  auto mx = [](int a, int b) { return a > b ? a : b; };
  L.lda = mx(n.F2p, n.hx2 + n.ht2) + 8; L.a = take(R * L.lda);       // Fourier features, then [sx | st]
  L.ldb = mx(n.ht1, n.hj1) + 8;         L.b = take(R * L.ldb);       // t1, then j1
  L.ldc = mx(n.hx1, n.hj2) + 8;         L.c = take(R * L.ldc);       // x1, then j2
  L.ldx = n.dp + 8;                     L.x = take(R * L.ldx);       // cond (data at column 4)
  L.ldg = n.dp + 8;                     L.g = take(R * L.ldg);       // gate
  L.tgt = take(R * n.dp); L.tt = take(R); L.gcs = take(R * 8); L.red = take(32);
  L.gmm = take(n.T.kind == MFM_TARGET_GMM ? n.T.n_modes * (2 * n.d + 1) : 0);      // mixture parameters, staged once per workgroup
  L.total = o;
  return L;
}
// MT row tiles of 16 samples per workgroup.  MT = 4: every streamed weight fragment feeds four MFMA tiles, one workgroup per CU
// (137 KB of LDS).  MT = 2: half the LDS and <= 128 registers, so TWO workgroups share a CU and one's batch construction
// (threefry + erfinv draws, sincos, the mixture's gradient: 15 % of a workgroup's cycles, all vector ALU), epilogues and
// barriers run under the other's MFMAs -- the measured section stamps of the MT = 4 kernel put its matrix pipe at 52 %.
template <int MT, int ACT, bool CHAIN = false>       // ACT: the hidden non-linearity as a compile-time constant (MFM_ACT_*), or -1: read from the network
__global__ __launch_bounds__((MLP_WAVES_FM * 64), (MT == 2 ? 4 : 2)) void fm_eval_kernel(FmArgs a) {      // (threads, waves per SIMD)
  constexpr int R = 16 * MT;
  const int act = ACT >= 0 ? ACT : a.net.act;
  if constexpr (MT == 2) {
    // Two workgroups share a CU so that one's vector-ALU phases (batch construction, epilogues) run under the other's MFMAs --
    // but workgroups dispatched together and equally long stay IN PHASE (both in their prologue, both in their GEMMs).  The
    // second workgroup of every CU of the first dispatch round starts half a workgroup's time late; slots freed later inherit
    // the stagger.  (Dispatch order: one workgroup per CU over the whole chip, then the second.)
    const int stagger = a.stagger_cycles;
    if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)stagger) __builtin_amdgcn_s_sleep(64);
    }
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const NetDev& n = a.net;
  const FmEvalLds L = fm_eval_lds_layout(n, R);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * R, d = n.d;
  constexpr int NT_ = MLP_WAVES_FM * 64;
  float* bA = lds + L.a; float* bB = lds + L.b; float* bC = lds + L.c; float* bX = lds + L.x; float* bG = lds + L.g;
  float* bT = lds + L.tgt; float* btt = lds + L.tt; float* gcs = lds + L.gcs; double* red = reinterpret_cast<double*>(lds + L.red);
  FM_STAMP(0);
  // ---- batch construction (:151-169 / :139-147), rows past the end of the batch are zero rows that do not count ----
  for (int i = threadIdx.x; i < R * L.ldx; i += NT_) bX[i] = 0.f;
  for (int i = threadIdx.x; i < R * n.dp; i += NT_) bT[i] = 0.f;
  for (int r = threadIdx.x; r < R; r += NT_)
    btt[r] = b0 + r < a.B ? (float)uniform01(a.key_time, a.chain_offset + (uint32_t)(b0 + r), a.n_total) : 0.f;      // :154 / :142
  TargetDev Tl = n.T;                       // the mixture's parameters from LDS: gmm_eval walks them twice per row, one row per thread
  if (n.T.kind == MFM_TARGET_GMM) {
    float* gm = lds + L.gmm;
    const int K = n.T.n_modes;
    for (int i = threadIdx.x; i < K * d; i += NT_) { gm[i] = n.T.gmm_mode[i]; gm[K * d + i] = n.T.gmm_std[i]; }
    for (int i = threadIdx.x; i < K; i += NT_) gm[2 * K * d + i] = n.T.gmm_logw[i];
    Tl.gmm_mode = gm; Tl.gmm_std = gm + K * d; Tl.gmm_logw = gm + 2 * K * d;
  }
  __syncthreads();
  // only the d live columns draw (the pads of cond / target stay zero).  With few columns (the d = 2 mixtures: 2 R elements on
  // 512 threads) the two Gaussian draws of an element go to two threads and meet in LDS (float64, in the t1 buffer, which is
  // free until the first GEMM): one threefry + erfinv chain per thread instead of two in sequence
  if (a.cond_flow && 2 * R * d <= NT_) {
    double* stage = reinterpret_cast<double*>(bB);
    const int e = threadIdx.x >> 1, which = threadIdx.x & 1;
    if (e < R * d) {
      const int row = e / d, col = e - row * d;
      const uint32_t bg = a.chain_offset + (uint32_t)(b0 + row);
      double v = 0.0;
      if (b0 + row < a.B) {
        // ONE inlined threefry + erfinv chain for both kinds of draw (key / counter / scale selected first): two copies made the
        // prologue 7.6 k instructions of a kernel whose code then exceeded the 64 KB instruction cache two CUs share
        const Key2 kr = split_at(a.key_ref, a.n_total, bg);
        const Key2 kk = which == 0 ? kr : a.key_gauss;
        const uint32_t idx = which == 0 ? (uint32_t)col : bg * (uint32_t)d + (uint32_t)col, size = which == 0 ? (uint32_t)d : a.n_total * (uint32_t)d;
        v = (which == 0 ? a.ref_std : 1.0) * normal64(kk, idx, size);                                                         // :155 / :166
      }
      stage[2 * e + which] = v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < R * d; q += NT_) {
      const int row = q / d, col = q - row * d;
      if (b0 + row < a.B) {
        const double x1v = a.pos[(size_t)(b0 + row) * d + col], t = btt[row], x0 = stage[2 * q], ne = stage[2 * q + 1];
        bX[row * L.ldx + 4 + col] = (float)((double)a.sigma * ne + t * x1v + (1.0 - t) * x0);                                   // :167
        bT[row * n.dp + col] = (float)(x1v - x0);                                                                               // :168
      }
    }
  } else
  for (int e = threadIdx.x; e < R * d; e += NT_) {
    const int row = e / d, col = e - row * d;
    if (b0 + row < a.B) {
      const uint32_t bg = a.chain_offset + (uint32_t)(b0 + row);
      const double x1v = a.pos[(size_t)(b0 + row) * d + col], t = btt[row];
      double cnd, tgd;
      if (a.cond_flow) {
        const double x0 = a.ref_std * normal64_call(split_at(a.key_ref, a.n_total, bg), (uint32_t)col, (uint32_t)d);            // :155
        const double ne = normal64_call(a.key_gauss, bg * (uint32_t)d + (uint32_t)col, a.n_total * (uint32_t)d);               // :166
        cnd = (double)a.sigma * ne + t * x1v + (1.0 - t) * x0; tgd = x1v - x0;                                                  // :167-168
      } else {
        const double x0 = normal64_call(a.key_ref, bg * (uint32_t)d + (uint32_t)col, a.n_total * (uint32_t)d);                 // :143
        cnd = t * x1v + (1.0 - (1.0 - (double)a.sigma) * t) * x0; tgd = x1v - (1.0 - (double)a.sigma) * x0;                      // :144-146
      }
      bX[row * L.ldx + 4 + col] = (float)cnd; bT[row * n.dp + col] = (float)tgd;
    }
  }
  FM_STAMP(1);
  // Fourier features (:70-71): cos block, then sin block; ONE sincos per (row, frequency), and a thread keeps its frequency
  // while it walks the rows (the workgroup size is a multiple of the usual F: the frequency load leaves the loop)
  if (NT_ % n.F == 0) {
    const int col = threadIdx.x % n.F;
    const double f = n.fourier[col];
    for (int row = threadIdx.x / n.F; row < R; row += NT_ / n.F) {
      double ft = f * (double)btt[row];
      ft -= rint(ft);
      float sv, cvv;
      sincospif(2.f * (float)ft, &sv, &cvv);
      bA[row * L.lda + col] = cvv; bA[row * L.lda + n.F + col] = sv;
    }
  } else {
    for (int e = threadIdx.x; e < R * n.F; e += NT_) {
      const int row = e / n.F, col = e - row * n.F;
      double ft = (double)n.fourier[col] * (double)btt[row];
      ft -= rint(ft);
      float sv, cvv;
      sincospif(2.f * (float)ft, &sv, &cvv);
      bA[row * L.lda + col] = cvv; bA[row * L.lda + n.F + col] = sv;
    }
  }
  for (int e = threadIdx.x; e < R * (n.F2p - 2 * n.F); e += NT_) {      // pad columns of the first layer's K (F2p = ceil16(2 F))
    const int w = n.F2p - 2 * n.F, row = e / w;
    bA[row * L.lda + 2 * n.F + (e - row * w)] = 0.f;
  }
  __syncthreads();
  FM_STAMP(2);
  // grad log pi of the mixture (distributions.py:58-61 in its log-sum-exp form).  With K <= 16 modes: ONE MODE PER LANE, 16 lanes
  // per row -- component log-weights, their maximum, the responsibilities and the d gradient sums by DPP row reductions, all
  // 512 threads busy (the serial per-row walk over the modes was ~3.4 k instructions on one wave, 8-14 k cycles on the critical
  // path of every workgroup).  The sums run over the modes in another order than gmm_eval's loop: float rounding only.
  if (n.T.kind == MFM_TARGET_GMM && n.T.n_modes <= 16) {
    for (int p = threadIdx.x; p < R * 16; p += NT_) {
      const int row = p >> 4;
      double lp; float gg[8];
      if (d == 2) gmm_eval_lanes16<2>(Tl, bX + row * L.ldx + 4, p & 15, &lp, gg);      // (the same sums; the unrolled walk over 8 - d dead columns dropped)
      else gmm_eval_lanes16<8>(Tl, bX + row * L.ldx + 4, p & 15, &lp, gg);
      if ((p & 15) == 0)
        for (int j = 0; j < d; ++j) gcs[row * 8 + j] = gg[j];
    }
  } else if (n.T.kind == MFM_TARGET_GMM && wave == MLP_WAVES_FM - 1 && lane < R) {
    const int row = lane;
    double lp; float gg[8];
    gmm_eval<8>(Tl, bX + row * L.ldx + 4, &lp, gg);
    for (int j = 0; j < d; ++j) gcs[row * 8 + j] = gg[j];
  }
  FM_STAMP(3);
  auto relu_store = [&](float* out, int ldo, int coff) {
    return [&, out, ldo, coff](int q, int nt, int m, f32x4 acc, float bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(16 * m + 4 * g + i) * ldo + coff + nt * 16 + c] = act_f(acc[i] + bias, act);
    };
  };
  auto L_ = [&](int l) -> const LayerDesc& { return n.L[l]; };
  // the five full-width layers with the weight stream chained from tile to tile (mlp.hip.h: layer_gemm_chain) where every K is a
  // multiple of 128 (an even number of fragment groups per tile) and each wave owns exactly one column tile per layer
  const bool chain_ok = CHAIN;
  WChain wch; wch.have = false;
  auto LGE = [&](const float* A, int lda, int l, auto epi, int lnext) {
    const LayerDesc& ld = n.L[l];
    if (chain_ok)
      layer_gemm_chain<MT, MLP_WAVES_FM>(A, lda, n.Wp + ld.w_off, n.bias + ld.b_off, ld.Kp / 16, ld.Np / 16, wave, lane, epi, wch,
                                         WNext{lnext >= 0 ? n.Wp + n.L[lnext < 0 ? 0 : lnext].w_off + (size_t)wave * (n.L[lnext < 0 ? 0 : lnext].Kp / 16) * 256 : nullptr,
                                               lnext >= 0 ? n.L[lnext].Kp / 16 : 0});
    else
      layer_gemm<MT, MLP_WAVES_FM>(A, lda, n.Wp + ld.w_off, n.bias + ld.b_off, ld.Kp / 16, ld.Np / 16, wave, lane, epi);
  };
  LGE(bA, L.lda, 0, relu_store(bB, L.ldb, 0), 1);       // t1
  FM_STAMP(4);
  if (d == 2) {        // x1 with K = 2: two multiply-adds per output on the vector ALU (as a GEMM job its 16-deep padded K and the
                       // wait of seven idle waves cost 7 k cycles per workgroup)
    const float* W2 = n.Wp + L_(2).w_off;
    if (NT_ % n.hx1 == 0) {          // a thread keeps its column (two weights and the bias in registers) while it walks the rows
      const int col = threadIdx.x % n.hx1;
      const float w0 = W2[pack_index(0, col, 1)], w1 = W2[pack_index(1, col, 1)], bb = n.bias[L_(2).b_off + col];
      for (int row = threadIdx.x / n.hx1; row < R; row += NT_ / n.hx1)
        bC[row * L.ldc + col] = act_f(fmaf(bX[row * L.ldx + 5], w1, bX[row * L.ldx + 4] * w0) + bb, act);
    } else
    for (int e = threadIdx.x; e < R * n.hx1; e += NT_) {
      const int row = e / n.hx1, col = e - row * n.hx1;
      const float pre = fmaf(bX[row * L.ldx + 5], W2[pack_index(1, col, 1)], bX[row * L.ldx + 4] * W2[pack_index(0, col, 1)]) + n.bias[L_(2).b_off + col];
      bC[row * L.ldc + col] = act_f(pre, act);
    }
  } else
  layer_gemm<MT, MLP_WAVES_FM>(bX + 4, L.ldx, n.Wp + L_(2).w_off, n.bias + L_(2).b_off, L_(2).Kp / 16, L_(2).Np / 16, wave, lane, relu_store(bC, L.ldc, 0));   // x1
  __syncthreads();
  FM_STAMP(5);
  LGE(bB, L.ldb, 1, relu_store(bA, L.lda, n.hx2), 3);   // st
  LGE(bC, L.ldc, 3, relu_store(bA, L.lda, 0), 5);       // sx
  __syncthreads();
  FM_STAMP(6);
  // gate and out have dp / 16 = 1 column tile: instead of ONE wave pushing the four row tiles through it (the other seven
  // waiting at the barrier), four waves take one row tile each
  const bool narrow = n.dp == 16;
  if (narrow) {
    if (wave < MT)
      layer_gemm<1, 1>(bA + n.hx2 + wave * 16 * L.lda, L.lda, n.Wp + L_(4).w_off, n.bias + L_(4).b_off, L_(4).Kp / 16, 1, 0, lane,
                       [&](int q, int nt, int m, f32x4 acc, float bias) {
#pragma unroll
                         for (int i = 0; i < 4; ++i) bG[(16 * wave + 4 * g + i) * L.ldg + c] = acc[i] + bias;
                       });
  } else
  layer_gemm<MT, MLP_WAVES_FM>(bA + n.hx2, L.lda, n.Wp + L_(4).w_off, n.bias + L_(4).b_off, L_(4).Kp / 16, L_(4).Np / 16, wave, lane,                         // gate
                              [&](int q, int nt, int m, f32x4 acc, float bias) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) bG[(16 * m + 4 * g + i) * L.ldg + nt * 16 + c] = acc[i] + bias;
                              });
  LGE(bA, L.lda, 5, relu_store(bB, L.ldb, 0), 6);       // j1
  __syncthreads();
  FM_STAMP(7);
  LGE(bB, L.ldb, 6, relu_store(bC, L.ldc, 0), -1);      // j2
  __syncthreads();
  FM_STAMP(8);
  float loss_loc = 0.f;        // out + loss (:88-90, :177-178)
  auto out_epi = [&](int mt, int nt, f32x4 acc, float bias) {
    const int col = nt * 16 + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * mt + 4 * g + i;
      if (col < d && b0 + row < a.B) {
        const float gc = target_gclip(n, bX, L.ldx, gcs, nullptr, 0, row, col);
        const float r = acc[i] + bias + bG[row * L.ldg + col] * gc - bT[row * n.dp + col];
        loss_loc += r * r;
      }
    }
  };
  if (narrow) {
    if (wave < MT)
      layer_gemm<1, 1>(bC + wave * 16 * L.ldc, L.ldc, n.Wp + L_(7).w_off, n.bias + L_(7).b_off, L_(7).Kp / 16, 1, 0, lane,
                       [&](int q, int nt, int m, f32x4 acc, float bias) { out_epi(wave, 0, acc, bias); });
  } else
  layer_gemm<MT, MLP_WAVES_FM>(bC, L.ldc, n.Wp + L_(7).w_off, n.bias + L_(7).b_off, L_(7).Kp / 16, L_(7).Np / 16, wave, lane,
                              [&](int q, int nt, int m, f32x4 acc, float bias) { out_epi(m, nt, acc, bias); });
  FM_STAMP(9);
  const double lw = wave_sum((double)loss_loc);
  if (lane == 0) red[wave] = lw;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < MLP_WAVES_FM; ++w) tot += red[w];
    a.loss_part[blockIdx.x] = tot;
  }
  FM_STAMP(10);
}

// ---- weight gradients: dW[k][n] = sum_b A[b][k] dZ[b][n], db[n] = sum_b dZ[b][n] ------------------------------
// One workgroup (4 waves) per 64 x 64 output block, one wave per 32 x 32 quadrant (2 x 2 MFMA tiles), SPLIT slices of the
// chain axis.  Both operands come from the packed workspaces as float4 per lane; partial sums go to slab[split][n_params] in the
// canonical flat layout (deterministic: no atomics), reduced by reduce_slabs_kernel / the AdamW kernel.
struct WgradJob { int layer, kt0, nt0; };   // kt0, nt0 in units of 16; covers tiles kt0..kt0+3, nt0..nt0+3 (one workgroup)

struct WgradArgs {
  NetDev net;
  WsLayout ws;
  const float* acts; const float* dzs;
  const WgradJob* jobs; int n_jobs;
  int nbb, split;
  float* slabs;            // [split][n_params]
  int* flag_reset;         // non-null: the non-finite flag reduce_slabs_kernel will raise for THIS gradient (cleared here)
  int* flag_partial;       // non-null (one launch for reduction + optimizer, optim.hip: reduce_adamw_kernel): raised when a partial sum
                           // is non-finite or so large that the sum over the slices could overflow; cleared by the training kernel
};

__host__ __device__ __forceinline__ int wgrad_a_tile(const NetDev& n, const WsLayout& w, int layer, int kt) {
  switch (layer) {
    case 0: return w.a_ffat + kt;
    case 1: return w.a_t1 + kt;
    case 2: return w.a_cond + kt;
    case 3: return w.a_x1 + kt;
    case 4: return w.a_st + kt;
    case 5: return kt < n.hx2 / 16 ? w.a_sx + kt : w.a_st + (kt - n.hx2 / 16);
    case 6: return w.a_j1 + kt;
    default: return w.a_j2 + kt;
  }
}
__host__ __device__ __forceinline__ int wgrad_z_tile(const WsLayout& w, int layer, int nt) {
  switch (layer) {
    case 0: return w.z_t1 + nt;
    case 1: return w.z_t2 + nt;
    case 2: return w.z_x1 + nt;
    case 3: return w.z_x2 + nt;
    case 4: return w.z_gate + nt;
    case 5: return w.z_j1 + nt;
    case 6: return w.z_j2 + nt;
    default: return w.z_out + nt;
  }
}

// Workgroup = 4 waves = one 64 x 64 block of dW (wave (wk, wn) owns the 32 x 32 quadrant); the block's four A tiles and
// four dZ tiles of a chain tile are fetched ONCE per workgroup into LDS (2 chain tiles per stage, double-buffered) and read
// from there by the two waves that need each: with every wave fetching its own 2 + 2 tiles the kernel moved 218 MB through
// L2 -> L1 per launch (7.3 TB/s: its bound), now 109 MB.
#ifndef MFM_WG_BB
#define MFM_WG_BB 2
#endif
constexpr int WG_BB = MFM_WG_BB;                           // chain tiles per LDS stage
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  __shared__ f32x4 sh[2][WG_BB][8][64];                   // [stage][chain tile][A0..A3, Z0..Z3][lane]: 32 KB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const int wk = wave >> 1, wn = wave & 1;
  if (a.flag_reset && blockIdx.x == 0 && threadIdx.x == 0) *a.flag_reset = 0;
  // Workgroup -> (block, chain slice), XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.
  // id % split = slice puts (with split = 8) ALL blocks of one chain slice on one XCD, where they run concurrently and walk the
  // slice's chain tiles together: an operand tile that 2 - 4 blocks of its layer need is then fetched into that L2 once.  With
  // (block, slice) = (blockIdx.x, blockIdx.y) the blocks of a layer were spread over all XCDs: 90 % of the kernel's L2 requests
  // missed (TCC_MISS 0.87 M of 0.97 M per launch, 111 MB from beyond L2 -- its bound, not the MFMAs: profiles/r03_pmc_summary.json).
  const int lin = blockIdx.x, sp = lin % a.split;
  const WgradJob J = a.jobs[lin / a.split];
  const NetDev& n = a.net;
  const LayerDesc& ld = n.L[J.layer];
  const int KT = ld.Kp / 16, NT = ld.Np / 16;
  const int bb_lo = (int)((long long)a.nbb * sp / a.split), bb_hi = (int)((long long)a.nbb * (sp + 1) / a.split);
  // this wave FETCHES tiles `wave` (an A tile) and 4 + `wave` (a dZ tile) of the block; tiles past the layer's edge are
  // clamped to a valid one (fetched, never used)
  const int kt_f = J.kt0 + wave < KT ? J.kt0 + wave : J.kt0, nt_f = J.nt0 + wave < NT ? J.nt0 + wave : J.nt0;
  const f32x4* Af = reinterpret_cast<const f32x4*>(a.acts) + (size_t)wgrad_a_tile(n, a.ws, J.layer, kt_f) * a.nbb * 64 + lane;
  const f32x4* Zf = reinterpret_cast<const f32x4*>(a.dzs) + (size_t)wgrad_z_tile(a.ws, J.layer, nt_f) * a.nbb * 64 + lane;
  const int kq = J.kt0 + 2 * wk, nq = J.nt0 + 2 * wn;     // this wave's quadrant
  const bool k0 = kq < KT, k1 = kq + 1 < KT, n0 = nq < NT, n1 = nq + 1 < NT;
  f32x4 acc00 = {0, 0, 0, 0}, acc01 = acc00, acc10 = acc00, acc11 = acc00, bs0 = acc00, bs1 = acc00;
  f32x4 fa[WG_BB], fz[WG_BB];
  auto fetch = [&](int bb) {
#pragma unroll
    for (int u = 0; u < WG_BB; ++u) {
      const int bq = bb + u < bb_hi ? bb + u : bb_hi - 1;
      fa[u] = Af[(size_t)bq * 64]; fz[u] = Zf[(size_t)bq * 64];
    }
  };
  auto stage = [&](int st) {
#pragma unroll
    for (int u = 0; u < WG_BB; ++u) { sh[st][u][wave][lane] = fa[u]; sh[st][u][4 + wave][lane] = fz[u]; }
  };
  if (bb_lo < bb_hi) { fetch(bb_lo); stage(0); }
  __syncthreads();
  int st = 0;
  for (int bb = bb_lo; bb < bb_hi; bb += WG_BB, st ^= 1) {
    const bool more = bb + WG_BB < bb_hi;
    if (more) fetch(bb + WG_BB);
#pragma unroll
    for (int u = 0; u < WG_BB; ++u) {
      if (bb + u < bb_hi) {
        const f32x4 a0 = sh[st][u][2 * wk][lane], a1 = sh[st][u][2 * wk + 1][lane];
        const f32x4 z0 = sh[st][u][4 + 2 * wn][lane], z1 = sh[st][u][4 + 2 * wn + 1][lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], z0[s], acc00, 0, 0, 0);
          acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], z1[s], acc01, 0, 0, 0);
          acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], z0[s], acc10, 0, 0, 0);
          acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], z1[s], acc11, 0, 0, 0);
        }
        bs0 += z0; bs1 += z1;
      }
    }
    if (more) stage(st ^ 1);
    __syncthreads();
  }
  float* slab = a.slabs + (size_t)sp * n.n_params;
  const float big = 3.0e38f / (float)a.split;      // |partial| <= big for every slice: the sum over the slices cannot overflow
  bool suspicious = false;
  // the joint layer reads [sx | st] with both halves padded to 16: packed rows [ks_true, ks_pad) are padding, row r >= ks_pad is
  // canonical row r - ks_pad + ks_true (mlp.hip.h: packed_row); every other layer -- and every network whose hidden widths are multiples
  // of 16 -- has no gap and takes the branch-free store
  const int ks_true = J.layer == 5 ? n.L[3].N : 0, ks_pad = J.layer == 5 ? n.L[3].Np : 0;
  auto put = [&](f32x4 acc, int kt, int nt, auto gap) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = kt * 16 + 4 * g + i;
      const int nn = nt * 16 + c;
      if constexpr (decltype(gap)::value) {
        if (k >= ks_true) {
          if (k < ks_pad) continue;
          k -= ks_pad - ks_true;
        }
      }
      if (k < ld.K && nn < ld.N) { slab[ld.m_w + k * ld.N + nn] = acc[i]; suspicious |= !(fabsf(acc[i]) <= big); }
    }
  };
  if (ks_pad == ks_true) {      // (wave-uniform)
    const std::false_type ng;
    if (k0 && n0) put(acc00, kq, nq, ng);
    if (k0 && n1) put(acc01, kq, nq + 1, ng);
    if (k1 && n0) put(acc10, kq + 1, nq, ng);
    if (k1 && n1) put(acc11, kq + 1, nq + 1, ng);
  } else {
    const std::true_type wg;
    if (k0 && n0) put(acc00, kq, nq, wg);
    if (k0 && n1) put(acc01, kq, nq + 1, wg);
    if (k1 && n0) put(acc10, kq + 1, nq, wg);
    if (k1 && n1) put(acc11, kq + 1, nq + 1, wg);
  }
  if (kq == 0) {        // bias gradient: sum over chains of dZ, for the n-tiles of this quadrant
    float s0 = bs0[0] + bs0[1] + bs0[2] + bs0[3], s1 = bs1[0] + bs1[1] + bs1[2] + bs1[3];
    s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    if (g == 0) {
      const int nn0 = nq * 16 + c, nn1 = (nq + 1) * 16 + c;
      if (n0 && nn0 < ld.N) { slab[ld.m_b + nn0] = s0; suspicious |= !(fabsf(s0) <= big); }
      if (n1 && nn1 < ld.N) { slab[ld.m_b + nn1] = s1; suspicious |= !(fabsf(s1) <= big); }
    }
  }
  if (a.flag_partial && __ballot(suspicious) != 0ull && lane == 0) atomicOr(a.flag_partial, 1);
}

// grads[p] = sum_s slabs[s][p]; the last workgroup of the grid also totals the per-tile loss partials of the forward
// kernel (same fixed order as reduce_loss_kernel), which saves a launch per training step.
// With `bad` (single-rank training: the summed gradient IS the one the optimizer will see) it also raises the non-finite
// flag the AdamW kernel decides on, which saves the check kernel of the optimizer step.
__global__ void reduce_slabs_kernel(const float* slabs, int split, int n, float* out, const double* loss_part, int n_part, double* loss_out, int* bad) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  bool nf = false;
  if (p < n) {
    float s = 0.f;
    for (int k = 0; k < split; ++k) s += slabs[(size_t)k * n + p];
    out[p] = s;
    nf = !isfinite(s);
  }
  if (bad) {                                   // uniform over the grid
    const int any = __syncthreads_or(nf ? 1 : 0);
    if (any && threadIdx.x == 0) atomicOr(bad, 1);
  }
  if (loss_part && blockIdx.x == gridDim.x - 1) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) s += loss_part[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = sm[0];
  }
}

__global__ void reduce_loss_kernel(const double* part, int n, double* out, int accumulate) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = accumulate ? *out + sm[0] : sm[0];
}

// ---- launchers ---------------------------------------------------------------------------------------------
// number of per-workgroup loss partials the forward-only launch of `n` samples leaves in loss_part
static int fm_eval_rows(const NetDev& n, int B) {      // samples per workgroup of the forward-only kernel (0: the 16-chain training tile)
  if (n.dp > 16 || n.T.kind == MFM_TARGET_LGCP || B < 64 * 256 || g_sw.eval16) return 0;
  const int r = g_sw.eval_rows;
  if ((r != 32 && r != 64) || (size_t)fm_eval_lds_layout(n, r).total * sizeof(float) > (r == 32 ? 80 : 160) * 1024) return 0;
  return r;
}
int fm_eval_parts(const NetDev& n, int B) { const int r = fm_eval_rows(n, B); return r ? (B + r - 1) / r : B / 16; }

// configurations whose MALA step rides in the training kernel: the tile family's relu instances on the phi-four target, whose
// value and gradient one wave evaluates from its LDS row (rocprofv3 at the headline shape: 42.6 + 9.3 us as two kernels, 45.0 us
// as one).  The 2-d mixtures could (one mode per lane; tests/test_gpu_loop.py ran them bit-identical) but gain nothing: two
// chains per wave on 2 of 64 lanes each, in-line draws -- 44.1 + 10.2 -> 53.4 us at 4096 chains, 39.0 + 6.6 -> 47.1 us at 512;
// the Cox process needs the K^-1 GEMM.
bool fm_mala_fusable(const NetDev& n) {
  const int tpw = (n.dp / 16 + MLP_WAVES_FM - 1) / MLP_WAVES_FM;
  return n.act == MFM_ACT_RELU && n.T.kind == MFM_TARGET_PHI4 && tpw <= 2 && n.d <= 128 * tpw;
}

int launch_fm(const FmArgs& a, bool train, hipStream_t stream) {
  if (const int r = train ? 0 : fm_eval_rows(a.net, a.B)) {
    const size_t smr = (size_t)fm_eval_lds_layout(a.net, r).total * sizeof(float);
#define FM_EVAL_LAUNCH(MT_, ACT_)                                                                                     \
  do {                                                                                                                \
    (void)hipFuncSetAttribute((const void*)fm_eval_kernel<MT_, ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smr); \
    hipLaunchKernelGGL((fm_eval_kernel<MT_, ACT_>), dim3((a.B + 16 * MT_ - 1) / (16 * MT_)), dim3(MLP_WAVES_FM * 64), smr, stream, a); \
  } while (0)
    FmArgs a2 = a;
    a2.stagger_cycles = g_sw.eval_stagger;       // measured: 1.023 -> 0.994 ms on 409,600 samples (any delay of 20 k .. 70 k cycles)
    const FmArgs& a = a2;
    const bool relu = a.net.act == MFM_ACT_RELU;
    // the 32-sample relu instance with its five full-width layers chained (0.997 -> 0.985 ms on 409,600 samples; same arithmetic)
    bool chain = a.net.d == 2 && a.net.dp == 16 && !g_sw.eval_no_chain;
    for (int l : {0, 1, 3, 5, 6}) chain &= a.net.L[l].Kp % 128 == 0 && a.net.L[l].Np == 16 * MLP_WAVES_FM;
    if (r == 32 && relu && chain) {
      (void)hipFuncSetAttribute((const void*)fm_eval_kernel<2, MFM_ACT_RELU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smr);
      hipLaunchKernelGGL((fm_eval_kernel<2, MFM_ACT_RELU, true>), dim3((a.B + 31) / 32), dim3(MLP_WAVES_FM * 64), smr, stream, a);
    } else
    if (r == 32) { if (relu) FM_EVAL_LAUNCH(2, MFM_ACT_RELU); else FM_EVAL_LAUNCH(2, -1); }
    else { if (relu) FM_EVAL_LAUNCH(4, MFM_ACT_RELU); else FM_EVAL_LAUNCH(4, -1); }
#undef FM_EVAL_LAUNCH
    return 0;
  }
  const FmLds L = fm_lds_layout(a.net, train);
  const size_t sm = (size_t)L.total * sizeof(float);
  if (sm > 160 * 1024) return -3;
  const int tpw = (a.net.dp / 16 + MLP_WAVES_FM - 1) / MLP_WAVES_FM;
  dim3 grid(a.B / 16), block((MLP_WAVES_FM * 64));
#define FM_LAUNCH_A(T, TR, ACT_)                                                                           \
  do {                                                                                                     \
    (void)hipFuncSetAttribute((const void*)fm_fwd_bwd_kernel<T, TR, false, ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
    hipLaunchKernelGGL((fm_fwd_bwd_kernel<T, TR, false, ACT_>), grid, block, sm, stream, a.net.Wp + a.net.L[0].w_off, a.pos, a.mala.grad, a.mala.pre_n, a.mala.logp, a.mala.pre_u, a);              \
  } while (0)
#define FM_LAUNCH(T, TR) do { if (a.net.act == MFM_ACT_RELU) FM_LAUNCH_A(T, TR, MFM_ACT_RELU); else FM_LAUNCH_A(T, TR, -1); } while (0)
#define FM_LAUNCH_M(T, STATIC_, ACT_)                                                                      \
  do {                                                                                                     \
    (void)hipFuncSetAttribute((const void*)fm_fwd_bwd_kernel<T, true, STATIC_, ACT_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
    hipLaunchKernelGGL((fm_fwd_bwd_kernel<T, true, STATIC_, ACT_, true>), grid, block, sm, stream, a.net.Wp + a.net.L[0].w_off, a.pos, a.mala.grad, a.mala.pre_n, a.mala.logp, a.mala.pre_u, a);    \
  } while (0)
  const NetDev& n = a.net;
  bool headline = n.d == 256 && n.dp == 256 && n.F == 128 && n.F2p == 256 && n.ht1 == 128 && n.ht2 == 128 && n.hx1 == 128 &&
                  n.hx2 == 128 && n.hj1 == 128 && n.hj2 == 128 && n.T.kind == MFM_TARGET_PHI4 && n.act == MFM_ACT_RELU && !g_sw.generic_fm;
  static const int Kh[MLP_NLAYER] = {256, 128, 256, 128, 128, 256, 128, 128}, Nh[MLP_NLAYER] = {128, 128, 128, 128, 256, 128, 128, 256};
  for (int l = 0; l < MLP_NLAYER; ++l) headline &= n.L[l].K == Kh[l] && n.L[l].Kp == Kh[l] && n.L[l].N == Nh[l] && n.L[l].Np == Nh[l];
#define FM_LAUNCH_S(TR)                                                                                    \
  do {                                                                                                     \
    (void)hipFuncSetAttribute((const void*)fm_fwd_bwd_kernel<2, TR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
    hipLaunchKernelGGL((fm_fwd_bwd_kernel<2, TR, true>), grid, block, sm, stream, a.net.Wp + a.net.L[0].w_off, a.pos, a.mala.grad, a.mala.pre_n, a.mala.logp, a.mala.pre_u, a);                     \
  } while (0)
  if (a.mala.on) {      // the iteration's MALA step in the same launch (callers ask fm_mala_fusable first)
    if (!train || !fm_mala_fusable(n)) return -3;
    if (headline) FM_LAUNCH_M(2, true, -1); else if (tpw <= 1) FM_LAUNCH_M(1, false, MFM_ACT_RELU); else FM_LAUNCH_M(2, false, MFM_ACT_RELU);
  } else if (headline) {
    if (train) FM_LAUNCH_S(true); else FM_LAUNCH_S(false);
  } else if (train) {
    if (tpw <= 1) FM_LAUNCH(1, true); else if (tpw <= 2) FM_LAUNCH(2, true); else return -3;
  } else {
    if (tpw <= 1) FM_LAUNCH(1, false); else if (tpw <= 2) FM_LAUNCH(2, false); else return -3;
  }
#undef FM_LAUNCH_S
#undef FM_LAUNCH_M
#undef FM_LAUNCH
#undef FM_LAUNCH_A
  return 0;
}

int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
  dim3 grid(a.n_jobs * a.split), block(256);
  hipLaunchKernelGGL(wgrad_kernel, grid, block, 0, stream, a);
  return 0;
}

void launch_reduce_slabs(const float* slabs, int split, int n, float* out, const double* loss_part, int n_part, double* loss_out, int* bad, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, slabs, split, n, out, loss_part, n_part, loss_out, bad);
}
void launch_reduce_loss(const double* part, int n, double* out, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, stream, part, n, out, accumulate);
}
