// Target densities evaluated on the device: value, gradient and Hessian-vector product, per chain row.
// Follows distributions.py:114-165 (PhiFour, Dirichlet b.c.), :42-77 (GaussianMixture, diagonal) and
// :231-314 (LogGaussianCoxPines, unwhitened).  The reference differentiates with jax.grad / jax.jvp; these are
// the closed forms, checked against the CPU oracle in tests/test_gpu_*.py.
//
// Row convention: a chain's position lives in an LDS row `xs` with one zero pad on EACH side
// (xs[-1] = xs[d] = 0), so the PhiFour stencil needs no branches.
#pragma once
#include "common.hip.h"

enum { MFM_TARGET_PHI4 = 0, MFM_TARGET_GMM = 1, MFM_TARGET_LGCP = 2 };

#define MFM_GMM_MAX_MODES 64

struct TargetDev {
  int kind;
  int dim;
  // phi4
  float coef;      // a * dim
  float tbeta;     // the model's own beta (20), NOT the annealing temperature
  // gmm (dim == 2 as the reference forces, distributions.py:53; general small dim supported)
  int n_modes;
  const float* gmm_mode;   // [K, d]
  const float* gmm_std;    // [K, d]
  const float* gmm_logw;   // [K]  log w_k - sum_j log s_kj - d/2 log 2pi
  // lgcp (unwhitened): loglik = sum(x c - a e^x), logprior = -1/2 (x-mu)^T K^-1 (x-mu) + log_norm
  const float* counts;     // [dp] (zero padded)
  const float* KinvP;      // K^-1 packed like an MLP layer (K = N = d; symmetric, so one packing serves both uses)
  const float* kbias;      // [dp]: -mu * rowsum(K^-1), so that x . K^-1 + kbias = K^-1 (x - mu)
  const float* kdiag;      // [dp]: diag(K^-1) (the Hessian diagonal of the exact-trace log-det, wide.hip)
  float mu, poisson_a, log_norm;
};

// ---- PhiFour --------------------------------------------------------------------------------------
// loglik terms of element j (to be summed over j): -beta (coef/2 (x_{j+1}-x_j)^2 [+ left edge] + (1-x^2)^2/(4 coef))
// The multiply-adds are written as explicit fused operations: left to the compiler's contraction, the SAME source rounded
// differently in two kernels that inline it (the MALA step stand-alone and inside the training kernel: gradients one float32 ulp
// apart, acceptance probabilities 1e-6 apart).
__device__ __forceinline__ float phi4_grad(const TargetDev& T, const float* xs, int j) {
  float x = xs[j];
  float lap = 2.f * x - xs[j - 1] - xs[j + 1];
  return -T.tbeta * __builtin_fmaf(T.coef, lap, -(x * __builtin_fmaf(-x, x, 1.f) / T.coef));
}
__device__ __forceinline__ double phi4_term(const TargetDev& T, const float* xs, int j) {
  // each bond (j, j+1) counted once, plus the left boundary bond for j == 0
  double x = xs[j];
  double dr = (double)xs[j + 1] - x;
  double u = dr * dr;
  if (j == 0) u = __builtin_fma(x, x, u);
  double q = __builtin_fma(-x, x, 1.0);
  return -(double)T.tbeta * __builtin_fma(0.5 * (double)T.coef, u, q * q / (4.0 * (double)T.coef));
}
__device__ __forceinline__ float phi4_hvp(const TargetDev& T, const float* xs, const float* vs, int j) {
  float x = xs[j], v = vs[j];
  float lap = 2.f * v - vs[j - 1] - vs[j + 1];
  return -T.tbeta * (T.coef * lap - (1.f - 3.f * x * x) * v / T.coef);
}

// ---- Gaussian mixture (evaluated by ONE lane for a whole row; d <= MAXD is tiny) ------------------------------
// log-sum-exp form: algebraically identical to the reference's log(sum_k w_k prod_j pdf) (distributions.py:59-61)
// but does not underflow in float32 ("fp32 vs the reference's fp64").
// Written without dynamically indexed local arrays (no scratch): component log-weights are recomputed in the
// second pass instead of being stored.
template <int MAXD>
__device__ __forceinline__ float gmm_comp(const TargetDev& T, const float (&x)[MAXD], int k) {
  float s = T.gmm_logw[k];
#pragma unroll
  for (int j = 0; j < MAXD; ++j)
    if (j < T.dim) {
      const float z = (x[j] - T.gmm_mode[k * T.dim + j]) / T.gmm_std[k * T.dim + j];
      s -= 0.5f * z * z;
    }
  return s;
}

// The same evaluation with ONE MODE PER LANE: the 16 lanes that share (lane >> 4) evaluate one row of a mixture of
// K <= 16 modes -- component log-weights, their maximum, the responsibilities and the gradient / Hessian-vector sums by DPP
// row reductions.  gmm_eval's serial walk over the modes (two passes, ~3.4 k instructions at K = 16) sat on one lane per row;
// here every lane of the group returns the row's results after ~150 instructions.  `k` = lane & 15.  The sums run over the
// modes in the reduction tree's order instead of the loop's: float rounding only.
template <int MAXD>
__device__ __forceinline__ void gmm_eval_lanes16(const TargetDev& T, const float* xp, int k, double* logp, float* grad,
                                                 const float* vp = nullptr, float* hv = nullptr) {
  const int d = T.dim;
  const bool live = k < T.n_modes;
  float comp = -INFINITY, a[MAXD], w[MAXD], av = 0.f;
  if (live) comp = T.gmm_logw[k];
#pragma unroll
  for (int j = 0; j < MAXD; ++j) {
    a[j] = 0.f; w[j] = 0.f;
    if (j < d && live) {
      const float sd = T.gmm_std[k * d + j], dx = xp[j] - T.gmm_mode[k * d + j], z = dx / sd;
      comp -= 0.5f * z * z;
      a[j] = -dx / (sd * sd);
      if (vp) { av += a[j] * vp[j]; w[j] = vp[j] / (sd * sd); }
    }
  }
  const float m = group16_max_dpp(comp);
  const float e = live ? expf(comp - m) : 0.f;
  const float se = group16_sum_dpp(e), inv = 1.f / se;
  *logp = (double)m + (double)logf(se);
  float gv = 0.f, t1[MAXD];
#pragma unroll
  for (int j = 0; j < MAXD; ++j) {
    grad[j] = 0.f; t1[j] = 0.f;
    if (j < d) {
      grad[j] = group16_sum_dpp(e * a[j]) * inv;
      if (vp) { t1[j] = group16_sum_dpp(e * (a[j] * av - w[j])) * inv; gv += grad[j] * vp[j]; }
    }
  }
  if (vp) {
#pragma unroll
    for (int j = 0; j < MAXD; ++j)
      if (j < d) hv[j] = t1[j] - grad[j] * gv;
  }
}

template <int MAXD>
__device__ __forceinline__ void gmm_eval(const TargetDev& T, const float* xp, double* logp, float* grad,
                                         const float* vp = nullptr, float* hv = nullptr) {
  const int d = T.dim, K = T.n_modes;
  float x[MAXD], v[MAXD], g[MAXD], t1[MAXD];
#pragma unroll
  for (int j = 0; j < MAXD; ++j) { x[j] = j < d ? xp[j] : 0.f; v[j] = (vp && j < d) ? vp[j] : 0.f; g[j] = 0.f; t1[j] = 0.f; }
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, gmm_comp<MAXD>(T, x, k));
  float se = 0.f;
  for (int k = 0; k < K; ++k) {
    const float e = expf(gmm_comp<MAXD>(T, x, k) - m);     // unnormalised responsibility
    se += e;
    float a[MAXD], av = 0.f;
#pragma unroll
    for (int j = 0; j < MAXD; ++j) {
      a[j] = 0.f;
      if (j < d) {
        const float sd = T.gmm_std[k * d + j];
        a[j] = -(x[j] - T.gmm_mode[k * d + j]) / (sd * sd);
        g[j] += e * a[j];
        av += a[j] * v[j];
      }
    }
    if (vp) {
#pragma unroll
      for (int j = 0; j < MAXD; ++j)
        if (j < d) {
          const float sd = T.gmm_std[k * d + j];
          t1[j] += e * (a[j] * av - v[j] / (sd * sd));
        }
    }
  }
  *logp = (double)m + (double)logf(se);
  const float inv = 1.f / se;
  float gv = 0.f;
#pragma unroll
  for (int j = 0; j < MAXD; ++j) { g[j] *= inv; t1[j] *= inv; gv += g[j] * v[j]; }
#pragma unroll
  for (int j = 0; j < MAXD; ++j)
    if (j < d) {
      grad[j] = g[j];
      if (vp) hv[j] = t1[j] - g[j] * gv;
    }
}
