// N3: selection step of conditional importance sampling (exe_flow_matching.py:280-296) and the row-wise Gaussian
// sampler of the reference distribution (distributions.py:93-97, vmapped at exe_flow_matching.py:285, :389, :453).
//
// The expensive parts of a CIS step -- one inverse solve per chain and num_importance_samples forward solves per chain
// with log-det, and the target log-density of every flow sample -- are the ODE / target kernels the other flow steps
// use (mfm_ode_transform, mfm_mala_init).  What remains is per chain: importance weights
//   w_0 = exp(logp(x) - log q0(u0) - vol0),  w_j = exp(logp(x_j) - log q0(u_j) - vol_j)       (:283, :289)
// a categorical draw over [w_0, w_1 .. w_n] (:290-292; jax.random.choice = inverse CDF at cumsum[-1] (1 - U)) and the
// state update (:293-295; the gradient of the previous state is kept as is, a quirk of the reference).
// One wavefront per chain, float64 weights (they span hundreds of orders of magnitude before normalisation), weights
// are normalised by their sum exactly as the reference does (no max subtraction: overflow -> NaN -> the reference's
// behaviour is reproduced, index 0 is chosen by searchsorted on an all-NaN table).
#include "prng.hip.h"

struct CisArgs {
  Key2 key; uint32_t n_total, chain_offset;
  int B, d, n_is;
  const float* u0; const float* vol0;          // [B, d], [B]: pull-back of the current positions
  const float* refs; const float* xs;          // [B n_is, d]: reference draws and their flow samples
  const float* vols; const double* lps;        // [B n_is]: log-dets, tempered target log-densities
  float* pos; double* logp;                    // state (in / out); logdensity_grad is untouched (:295)
  float* acc_prob; uint8_t* accepted; float* proposed; float* weight;
  double ref_std;                              // std of the flow's reference distribution IndepGaussian(dim, var) (distributions.py:80-97)
};

__global__ __launch_bounds__(256) void cis_select_kernel(CisArgs a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int d = a.d, n = a.n_is;
  const double c0 = -0.5 * (double)d * 1.8378770664093453 - (double)d * log(a.ref_std);          // -d/2 log(2 pi) - d log std   (distributions.py:90)
  auto refl = [&](const float* u) {
    double s = 0.0;
    for (int j = lane; j < d; j += 64) { const double v = (double)u[j] / a.ref_std; s += v * v; }
    return -0.5 * wave_sum(s) + c0;
  };
  // weights; the running (unnormalised) cumulative sum is recomputed in the second pass instead of being stored
  const double w0 = exp(a.logp[b] - refl(a.u0 + (size_t)b * d) - (double)a.vol0[b]);                    // :283
  double tot = w0;
  for (int j = 0; j < n; ++j) {
    const size_t r = (size_t)b * n + j;
    tot += exp(a.lps[r] - refl(a.refs + r * d) - (double)a.vols[r]);                                   // :289-290
  }
  const Key2 kb = split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b);                            // :303
  const double u = uniform01(split_at(kb, 4, 3), 0, 1);                                                 // key_choice (:281, :292)
  // p = w / tot; p_cuml = cumsum(p); r = p_cuml[-1] (1 - u); first index with p_cuml[idx] >= r
  double cum = w0 / tot, cum_last = 0.0;
  {
    double c = w0 / tot;
    for (int j = 0; j < n; ++j) { const size_t r = (size_t)b * n + j; c += exp(a.lps[r] - refl(a.refs + r * d) - (double)a.vols[r]) / tot; }
    cum_last = c;
  }
  const double rr = cum_last * (1.0 - u);
  int choice = 0; double wsel = w0 / tot;
  if (!(cum >= rr)) {
    choice = n;                                         // searchsorted returns len(p) when nothing qualifies -> clamp (gather clamps)
    for (int j = 0; j < n; ++j) {
      const size_t r = (size_t)b * n + j;
      const double pj = exp(a.lps[r] - refl(a.refs + r * d) - (double)a.vols[r]) / tot;
      cum += pj;
      if (cum >= rr) { choice = j + 1; wsel = pj; break; }
      if (j == n - 1) wsel = pj;
    }
  }
  const bool acc = choice != 0;
  const size_t pick = (size_t)b * n + (choice > 0 ? choice - 1 : 0);
  for (int j = lane; j < d; j += 64) {
    const float v = acc ? a.xs[pick * d + j] : a.pos[(size_t)b * d + j];
    if (a.proposed) a.proposed[(size_t)b * d + j] = v;                                                  // :293 / :294
    if (acc) a.pos[(size_t)b * d + j] = v;
  }
  if (lane == 0) {
    if (acc) a.logp[b] = a.lps[pick];
    if (a.acc_prob) a.acc_prob[b] = (float)wsel;
    if (a.accepted) a.accepted[b] = acc ? 1 : 0;
    if (a.weight) a.weight[b] = (float)wsel;
  }
}

static void launch_cis_select(const CisArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(cis_select_kernel, dim3((a.B + 3) / 4), dim3(256), 0, stream, a);
}
