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
#include "prng.cuh"
#include "targets.cuh"

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
  const double* pre_n; const double* pre_u;   // non-null: the step's Gaussian / uniform draws, produced ahead of time by noise_kernel
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
  } else if (T.n_modes <= 16) {  // GMM, one mode per lane: every 16-lane group of the wave evaluates the row (targets.cuh)
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

template <int MAXIT>
__global__ __launch_bounds__(MALA_WAVES * 64) void mala_step_kernel(MalaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.T.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowlen = d + 2;
  float* xs = smem + wave * rowlen + 1;
  float* gsm = smem + MALA_WAVES * rowlen + wave * MALA_MAXD_SMALL;
  const int b = blockIdx.x * MALA_WAVES + wave;
  const bool live = b < a.B;
  const size_t row = (size_t)b * d;

  float x[MAXIT], g[MAXIT], xn[MAXIT];
  double th1 = 0.0;                       // |x' - x - eps g|^2 = 2 eps |noise|^2
  Key2 k_int = {0, 0}, k_rmh = {0, 0};
  if (lane == 0) { xs[-1] = 0.f; xs[d] = 0.f; }
  if (live) {
    if (!a.pre_n) {
      const Key2 kb = a.keys ? Key2{a.keys[2 * b], a.keys[2 * b + 1]} : split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b);     // exe_flow_matching.py:303
      k_int = split_at(kb, 2, 0);                                                    // mala.py:93
      k_rmh = split_at(kb, 2, 1);
    }
    const double s2e = sqrt(2.0 * a.eps);
    double nz[MAXIT];                      // prefetched draws: all loads in flight before the first use
    if (a.pre_n) {
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) { const int j = lane + 64 * it; nz[it] = j < d ? a.pre_n[row + j] : 0.0; }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int j = lane + 64 * it;
      if (j < d) {
        x[it] = a.pos[row + j];
        g[it] = a.grad[row + j];
        double n = a.pre_n ? nz[it] : normal64(k_int, (uint32_t)j, (uint32_t)d);   // util.py:80-82
        double th = s2e * n;
        th1 += th * th;
        xn[it] = (float)((double)x[it] + a.eps * (double)g[it] + th);              // diffusions.py:25-30
        xs[j] = xn[it];
      }
    }
  }
  __syncthreads();
  if (!live) return;

  float gn[MAXIT];
  const double lpn = row_value_grad<MAXIT>(a.T, a.beta, xs, d, lane, gn, gsm);     // diffusions.py:32
  double th2 = 0.0;                       // |x - x' - eps g'|^2
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    int j = lane + 64 * it;
    if (j < d) {
      double t = (double)x[it] - (double)xn[it] - a.eps * (double)gn[it];
      th2 += t * t;
    }
  }
  th1 = wave_sum(th1);
  th2 = wave_sum(th2);
  const double lp = a.logp[b];
  const double inv4e = 0.25 / a.eps;
  const double new_E = -lp + inv4e * th1;                                          // mala.py:68-79, proposal.py:157
  const double prev_E = -lpn + inv4e * th2;                                        // proposal.py:158
  double delta = prev_E - new_E;                                                   // proposal.py:104
  if (a.textbook) delta = -delta;
  if (isnan(delta)) delta = -INFINITY;                                             // proposal.py:105
  const double p = fmin(exp(delta), 1.0);                                          // proposal.py:178
  const double u = a.pre_u ? a.pre_u[b] : uniform01(k_rmh, 0, 1);
  const bool acc = u < p;                                                          // proposal.py:179

#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    int j = lane + 64 * it;
    if (j < d) {
      if (a.proposed) a.proposed[row + j] = xn[it];
      if (acc) { a.pos[row + j] = xn[it]; a.grad[row + j] = gn[it]; }
    }
  }
  if (lane == 0) {
    if (acc) a.logp[b] = lpn;
    if (a.acc_prob) a.acc_prob[b] = (float)p;
    if (a.accepted) a.accepted[b] = acc ? 1 : 0;
    if (a.prop_weight) a.prop_weight[b] = (float)exp(lpn + inv4e * th2);           // mala.py:104-113 (diagnostic)
  }
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
