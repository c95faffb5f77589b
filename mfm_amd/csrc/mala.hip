// K1: fused MALA step over parallel chains (one wavefront per chain), K2: target value & gradient.
//
// Replaces the XLA computation of `jax.vmap(kernel)(keys, states)` at exe_flow_matching.py:313, i.e.
// bblackjax/mcmc/mala.py:86-118 + diffusions.py:19-34 + proposal.py:104-112,157-159,178-186, and
// `jax.vmap(init)` at exe_flow_matching.py:316 (mala.py:51-54).  The acceptance rule is reproduced AS WRITTEN
// (p = min(1, exp(prev_E - new_E)), the inverse of the textbook ratio); `textbook` flips it.
//
// Layout: position / gradient [B, d] float32 row-major, logdensity [B] float64 (|logp| ~ 4e4 for phi-four at
// d = 256, where a float32 ulp is 4e-3 -- too coarse for the energy difference).  One wave owns one chain: lanes
// stride the row (coalesced 256 B per wave instruction), the proposal row is staged in LDS with one zero pad on
// each side for the stencil, energies are reduced in float64 with wavefront shuffles.  Noise is drawn in-kernel
// (threefry + float64 erfinv), so the only HBM traffic is the algorithmic 4*(5d+5) bytes per chain.
#include "prng.hip.h"
#include "targets.hip.h"

#define MALA_WAVES 4
#define MALA_MAXD_SMALL 8

struct MalaArgs {
  TargetDev T;
  Key2 key;
  const uint32_t* keys;   // non-null: one key per chain [B][2] (a caller that vmaps over its own keys: bblackjax/smc/base.py:122-123)
  uint32_t n_total, chain_offset;
  int B;
  double beta;      // annealing temperature: logprob = beta * loglik + logprior
  double eps;       // step size
  int textbook;
  float* pos; double* logp; float* grad;                    // state, updated in place
  float* acc_prob; uint8_t* accepted; float* proposed; float* prop_weight;   // info (may be null)
  const draw_t* pre_n; const double* pre_u;   // non-null: the step's Gaussian / uniform draws, produced ahead of time by noise_kernel
};

// value (float64, wave-reduced) and gradient of the tempered target for the row staged in `xs`.
// grad is returned per lane for elements j = lane + 64*it in gout[it].
template <int MAXIT>
__device__ __forceinline__ double row_value_grad(const TargetDev& T, double beta, const float* xs, int d, int lane,
                                                 float (&gout)[MAXIT], float* gsm) {
  double acc = 0.0;
  if (T.kind == MFM_TARGET_PHI4) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int j = lane + 64 * it;
      if (j < d) {
        acc += phi4_term(T, xs, j);
        gout[it] = (float)beta * phi4_grad(T, xs, j);
      }
    }
    return beta * wave_sum(acc);
  } else if (T.kind == MFM_TARGET_LGCP) {   // likelihood part only (loglik_kernel); value/grad with the prior: lgcp.hip
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int j = lane + 64 * it;
      if (j < d) { acc += (double)xs[j] * (double)T.counts[j] - (double)T.poisson_a * (double)expf(xs[j]); gout[it] = 0.f; }
    }
    return beta * wave_sum(acc);
  } else if (T.n_modes <= 16) {  // GMM, one mode per lane: every 16-lane group of the wave evaluates the row (targets.hip.h)
    double lp = 0.0;
    float g[MALA_MAXD_SMALL];
    gmm_eval_lanes16<MALA_MAXD_SMALL>(T, xs, lane & 15, &lp, g);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      float gj = 0.f;
#pragma unroll
      for (int jj = 0; jj < MALA_MAXD_SMALL; ++jj) gj = (jj == j) ? g[jj] : gj;
      if (j < d) gout[it] = (float)beta * gj;
    }
    return beta * lp;
  } else {  // GMM with more than 16 modes: lane 0 evaluates the row, gradient broadcast through LDS scratch
    double lp = 0.0;
    if (lane == 0) {
      float g[MALA_MAXD_SMALL];
      gmm_eval<MALA_MAXD_SMALL>(T, xs, &lp, g);
      for (int j = 0; j < d; ++j) gsm[j] = g[j];
    }
    lp = __shfl(lp, 0, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int j = lane + 64 * it;
      if (j < d) gout[it] = (float)beta * gsm[j];
    }
    return beta * lp;
  }
}

template <int MAXIT>
__global__ __launch_bounds__(MALA_WAVES * 64) void mala_init_kernel(MalaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.T.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowlen = d + 2;
  float* xs = smem + wave * rowlen + 1;
  float* gsm = smem + MALA_WAVES * rowlen + wave * MALA_MAXD_SMALL;
  const int b = blockIdx.x * MALA_WAVES + wave;
  const bool live = b < a.B;
  if (lane == 0) { xs[-1] = 0.f; xs[d] = 0.f; }
  if (live)
    for (int j = lane; j < d; j += 64) xs[j] = a.pos[(size_t)b * d + j];
  __syncthreads();
  if (!live) return;
  float g[MAXIT];
  double lp = row_value_grad<MAXIT>(a.T, a.beta, xs, d, lane, g, gsm);
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    int j = lane + 64 * it;
    if (j < d) a.grad[(size_t)b * d + j] = g[it];
  }
  if (lane == 0) a.logp[b] = lp;
}

// loglik only (beta_fn input, exe_flow_matching.py:413,418)
template <int MAXIT>
__global__ __launch_bounds__(MALA_WAVES * 64) void loglik_kernel(MalaArgs a, double* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.T.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowlen = d + 2;
  float* xs = smem + wave * rowlen + 1;
  float* gsm = smem + MALA_WAVES * rowlen + wave * MALA_MAXD_SMALL;
  const int b = blockIdx.x * MALA_WAVES + wave;
  const bool live = b < a.B;
  if (lane == 0) { xs[-1] = 0.f; xs[d] = 0.f; }
  if (live)
    for (int j = lane; j < d; j += 64) xs[j] = a.pos[(size_t)b * d + j];
  __syncthreads();
  if (!live) return;
  float g[MAXIT];
  double lp = row_value_grad<MAXIT>(a.T, 1.0, xs, d, lane, g, gsm);
  if (lane == 0) out[b] = lp;
}

// One MALA step of the NCH chains b[0..NCH) by ONE wave (diffusions.py:19-34, mala.py:86-118, proposal.py:104-112,157-159,178-186).
// xs[c]: the wave's LDS row for chain c (d floats, zero pads at [-1] and [d] written here); on return it holds the chain's NEW
// position (proposal if accepted, the old position otherwise), which the fused MALA + training kernel (fm.hip) reads instead of
// going back to HBM.  The chains of a wave are independent: their loads and float64 butterfly sums interleave.  Only the wave's
// own lanes touch xs[c], so the staging needs no workgroup barrier.
// `after_loads()` runs once the step's own loads are in flight and before their first use: a caller with loads of its own issues
// them there, BEHIND these in the (in-order) memory queue, so that the step's data comes first and theirs arrives while it computes
// -- and does there whatever needs none of the loads.  (Every workgroup of the grid is in this prologue at once: the memory
// system delivers ~10 B/clk/CU and the ISSUE of a load stalls behind it; issuing the caller's loads later, after the proposal, stalled
// just as long there and was slower: tools/fm_stamps.py --loop, 8.3 k cycles for that section.)
struct MalaNoHook { __device__ __forceinline__ void operator()() const {} };
template <int MAXIT, int NCH, typename Hook = MalaNoHook>
__device__ __forceinline__ void mala_chain_step(const MalaArgs& a, const int (&b)[NCH], float* const (&xs)[NCH], float* const (&gsm)[NCH], int lane,
                                                Hook after_loads = Hook()) {
  // No multiply-add contraction in this function's own arithmetic: it is instantiated in two kernels (stand-alone, and inside the
  // training kernel where unused outputs fold away), and whether `-(beta * S) + c * t` becomes one fused operation depended on how
  // many uses beta * S had left -- acceptance probabilities differed in the last bit between the two at beta < 1.
#pragma clang fp contract(off)
  const int d = a.T.dim;
  float x[NCH][MAXIT], g[NCH][MAXIT], xn[NCH][MAXIT];
  double th1[NCH];                        // |x' - x - eps g|^2 = 2 eps |noise|^2
  Key2 k_int[NCH], k_rmh[NCH];
  const double s2e = sqrt(2.0 * a.eps);
  draw_t nz[NCH][MAXIT];                  // prefetched draws: all loads in flight before the first use
  double lp0[NCH], u0[NCH];               // the accept step's two scalars: requested here, a whole HBM round trip before their use
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const size_t row = (size_t)b[c] * d;
    th1[c] = 0.0; k_int[c] = Key2{0, 0}; k_rmh[c] = Key2{0, 0};
    lp0[c] = a.logp[b[c]];
    u0[c] = a.pre_u ? a.pre_u[b[c]] : 0.0;
    if (lane == 0) { xs[c][-1] = 0.f; xs[c][d] = 0.f; }
    if (!a.pre_n) {
      const Key2 kb = a.keys ? Key2{a.keys[2 * b[c]], a.keys[2 * b[c] + 1]} : split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b[c]);     // exe_flow_matching.py:303
      k_int[c] = split_at(kb, 2, 0);                                                 // mala.py:93
      k_rmh[c] = split_at(kb, 2, 1);
    } else {
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) { const int j = lane + 64 * it; nz[c][it] = j < d ? a.pre_n[row + j] : (draw_t)0; }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      if (j < d) { x[c][it] = a.pos[row + j]; g[c][it] = a.grad[row + j]; }
    }
  }
  FM_STAMP(10);
  __builtin_amdgcn_sched_barrier(0);
  after_loads();
  __builtin_amdgcn_sched_barrier(0);
  FM_STAMP(11);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      if (j < d) {
        const double n = a.pre_n ? (double)nz[c][it] : (double)(draw_t)normal64(k_int[c], (uint32_t)j, (uint32_t)d);   // util.py:80-82
        const double th = s2e * n;
        th1[c] += th * th;
        xn[c][it] = (float)((double)x[c][it] + a.eps * (double)g[c][it] + th);     // diffusions.py:25-30
        xs[c][j] = xn[c][it];
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // the stencil reads its neighbours' proposal elements from this wave's row
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  FM_STAMP(12);
  float gn[NCH][MAXIT];
  double lpn[NCH], th2[NCH];              // th2 = |x - x' - eps g'|^2
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    lpn[c] = row_value_grad<MAXIT>(a.T, a.beta, xs[c], d, lane, gn[c], gsm[c]);    // diffusions.py:32
    th2[c] = 0.0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      if (j < d) {
        const double t = (double)x[c][it] - (double)xn[c][it] - a.eps * (double)gn[c][it];
        th2[c] += t * t;
      }
    }
  }
  FM_STAMP(13);
#pragma unroll
  for (int c = 0; c < NCH; ++c) { th1[c] = wave_sum(th1[c]); th2[c] = wave_sum(th2[c]); }
  FM_STAMP(14);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const size_t row = (size_t)b[c] * d;
    const double lp = lp0[c];
    const double inv4e = 0.25 / a.eps;
    const double new_E = -lp + inv4e * th1[c];                                       // mala.py:68-79, proposal.py:157
    const double prev_E = -lpn[c] + inv4e * th2[c];                                  // proposal.py:158
    double delta = prev_E - new_E;                                                   // proposal.py:104
    if (a.textbook) delta = -delta;
    if (isnan(delta)) delta = -INFINITY;                                             // proposal.py:105
    const double p = fmin(exp(delta), 1.0);                                          // proposal.py:178
    const double u = a.pre_u ? u0[c] : uniform01(k_rmh[c], 0, 1);
    const bool acc = u < p;                                                          // proposal.py:179
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      if (j < d) {
        if (a.proposed) a.proposed[row + j] = xn[c][it];
        if (acc) { a.pos[row + j] = xn[c][it]; a.grad[row + j] = gn[c][it]; }
        else xs[c][j] = x[c][it];
      }
    }
    if (lane == 0) {
      if (acc) a.logp[b[c]] = lpn[c];
      if (a.acc_prob) a.acc_prob[b[c]] = (float)p;
      if (a.accepted) a.accepted[b[c]] = acc ? 1 : 0;
      if (a.prop_weight) a.prop_weight[b[c]] = (float)exp(lpn[c] + inv4e * th2[c]);   // mala.py:104-113 (diagnostic)
    }
  }
}

template <int MAXIT>
__global__ __launch_bounds__(MALA_WAVES * 64) void mala_step_kernel(MalaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.T.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowlen = d + 2;
  const int b = blockIdx.x * MALA_WAVES + wave;
  if (b >= a.B) return;
  const int bs[1] = {b};
  float* const xs[1] = {smem + wave * rowlen + 1};
  float* const gsm[1] = {smem + MALA_WAVES * rowlen + wave * MALA_MAXD_SMALL};
  mala_chain_step<MAXIT, 1>(a, bs, xs, gsm, lane);
}

// ---- launchers (called from the C ABI in api.hip) ---------------------------------------------------------
static inline size_t mala_smem(int d) { return (size_t)(MALA_WAVES * (d + 2) + MALA_WAVES * MALA_MAXD_SMALL) * sizeof(float); }

#define MALA_DISPATCH(KERN, ...)                                                             \
  do {                                                                                       \
    int nit = (a.T.dim + 63) / 64;                                                           \
    dim3 grid((a.B + MALA_WAVES - 1) / MALA_WAVES), block(MALA_WAVES * 64);                  \
    size_t sm = mala_smem(a.T.dim);                                                          \
    if (nit <= 1) hipLaunchKernelGGL(KERN<1>, grid, block, sm, stream, __VA_ARGS__);         \
    else if (nit <= 4) hipLaunchKernelGGL(KERN<4>, grid, block, sm, stream, __VA_ARGS__);    \
    else if (nit <= 16) hipLaunchKernelGGL(KERN<16>, grid, block, sm, stream, __VA_ARGS__);  \
    else if (nit <= 32) hipLaunchKernelGGL(KERN<32>, grid, block, sm, stream, __VA_ARGS__);  \
    else return -2;                                                                          \
  } while (0)

int launch_mala_init(const MalaArgs& a, hipStream_t stream) { MALA_DISPATCH(mala_init_kernel, a); return 0; }
int launch_mala_step(const MalaArgs& a, hipStream_t stream) { MALA_DISPATCH(mala_step_kernel, a); return 0; }
int launch_loglik(const MalaArgs& a, double* out, hipStream_t stream) { MALA_DISPATCH(loglik_kernel, a, out); return 0; }
