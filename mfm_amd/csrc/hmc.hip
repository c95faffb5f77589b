// Hamiltonian Monte Carlo step over parallel chains, one wavefront per chain -- a BUILD-SIDE MODE: BASELINE.json's north star names a
// "MALA/HMC log-density-and-grad step", the reference's MFM loop has MALA only and vendors no hmc.py (SURVEY.md note 7).  The kernel
// follows blackjax's HMC (the package the reference's bblackjax was cut from), restated in oracle/hmc.py: momentum ~ N(0, I) from
// split(key, 2)[0], num_steps velocity-Verlet steps (half kick, drift, value-and-gradient, half kick), H = -logp + |p|^2 / 2,
// accept with min(1, exp(H_0 - H_end)) against uniform(split(key, 2)[1]) (proposal.py:105,178-179).
// Layout as the MALA kernel (mala.hip): positions / gradients float32, log-density float64; the momentum lives in float64 registers,
// the trajectory's position is rounded to float32 after every drift (it is what the target is evaluated at); energies are summed in
// float64 over the wave.  Targets: phi-four and the Gaussian mixtures (row_value_grad); the Cox process needs the K^-1 GEMM tile
// (lgcp.hip) and is not served.
#pragma once
// (included by api.hip after mala.hip: MalaArgs' helpers row_value_grad, MALA_DISPATCH, mala_smem)

struct HmcArgs {
  TargetDev T;
  Key2 key;
  uint32_t n_total, chain_offset;
  int B, num_steps;
  double beta, eps;
  float* pos; double* logp; float* grad;              // state, updated in place
  float* acc_prob; uint8_t* accepted;                 // info (may be null)
};

template <int MAXIT>
__global__ __launch_bounds__(MALA_WAVES * 64) void hmc_step_kernel(HmcArgs a) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.T.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowlen = d + 2;
  const int b = blockIdx.x * MALA_WAVES + wave;
  if (b >= a.B) return;
  float* xs = smem + wave * rowlen + 1;
  float* gsm = smem + MALA_WAVES * rowlen + wave * MALA_MAXD_SMALL;
  const size_t row = (size_t)b * d;
  const Key2 kb = split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b);
  const Key2 k_mom = split_at(kb, 2, 0), k_acc = split_at(kb, 2, 1);
  float x0[MAXIT], g0[MAXIT], x[MAXIT], g[MAXIT];
  double p[MAXIT];
  double kin = 0.0;
  if (lane == 0) { xs[-1] = 0.f; xs[d] = 0.f; }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = lane + 64 * it;
    x0[it] = 0.f; g0[it] = 0.f; p[it] = 0.0;
    if (j < d) {
      x0[it] = a.pos[row + j]; g0[it] = a.grad[row + j];
      p[it] = normal64(k_mom, (uint32_t)j, (uint32_t)d);
      kin += p[it] * p[it];
    }
    x[it] = x0[it]; g[it] = g0[it];
  }
  const double lp0 = a.logp[b];
  const double h0 = -lp0 + 0.5 * wave_sum(kin);
  double lp = lp0;
  for (int s = 0; s < a.num_steps; ++s) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      if (j < d) {
        p[it] = p[it] + 0.5 * a.eps * (double)g[it];                       // half kick
        x[it] = (float)((double)x[it] + a.eps * p[it]);                    // drift
        xs[j] = x[it];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // the stencil reads its neighbours' elements from this wave's row
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    lp = row_value_grad<MAXIT>(a.T, a.beta, xs, d, lane, g, gsm);
    __builtin_amdgcn_wave_barrier();                             // (the next drift overwrites the row)
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      if (j < d) p[it] = p[it] + 0.5 * a.eps * (double)g[it];             // half kick
    }
  }
  double kin1 = 0.0;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) kin1 += (lane + 64 * it < d) ? p[it] * p[it] : 0.0;
  const double h1 = -lp + 0.5 * wave_sum(kin1);
  double delta = h0 - h1;
  if (isnan(delta)) delta = -INFINITY;                                      // proposal.py:105
  const double pa = fmin(exp(delta), 1.0);                                  // proposal.py:178
  const bool acc = uniform01(k_acc, 0, 1) < pa;                             // proposal.py:179
  if (acc) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = lane + 64 * it;
      if (j < d) { a.pos[row + j] = x[it]; a.grad[row + j] = g[it]; }
    }
  }
  if (lane == 0) {
    if (acc) a.logp[b] = lp;
    if (a.acc_prob) a.acc_prob[b] = (float)pa;
    if (a.accepted) a.accepted[b] = acc ? 1 : 0;
  }
}

int launch_hmc_step(const HmcArgs& a, hipStream_t stream) { MALA_DISPATCH(hmc_step_kernel, a); return 0; }
