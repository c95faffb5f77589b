// K8: the next annealing temperature by bisection on the effective sample size (exe_flow_matching.py:391-402).
// jaxopt.Bisection(lower = prev_beta, upper = 1, maxiter = 30, tol = 1e-5, check_bracket = False) restated
// (SURVEY.md section 8c): returns the LAST midpoint evaluated.  One workgroup; <= 32 passes over the n log-likelihoods
// (float64, L2-resident).
#include "common.hip.h"

#define BETA_THREADS 1024

__device__ double block_reduce(double v, double* sm, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = sm[0];
  for (int w = 1; w < BETA_THREADS / 64; ++w) r = is_max ? fmax(r, sm[w]) : r + sm[w];
  return r;
}

// ess_zero(beta) = 1 / sum(w^2) - alpha * n,  w = softmax(loglik * (beta - prev_beta))      (:393-399)
__device__ double ess_zero(const double* ll, int n, double beta, double prev, double alpha, double* sm) {
  const double db = beta - prev;
  double m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += BETA_THREADS) m = fmax(m, ll[i] * db);
  m = block_reduce(m, sm, true);
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += BETA_THREADS) { double e = exp(ll[i] * db - m); s1 += e; s2 += e * e; }
  s1 = block_reduce(s1, sm, false);
  s2 = block_reduce(s2, sm, false);
  return s1 * s1 / s2 - alpha * (double)n;
}

__global__ __launch_bounds__(BETA_THREADS) void beta_kernel(double prev, const double* ll, int n, double alpha, double* out) {
  __shared__ double sm[BETA_THREADS / 64];
  double low = prev, high = 1.0;
  const double fl = ess_zero(ll, n, low, prev, alpha, sm), fh = ess_zero(ll, n, high, prev, alpha, sm);
  const int sign = (fl < 0 && fh >= 0) ? 1 : ((fl > 0 && fh <= 0) ? -1 : 0);
  double params = 0.5 * (low + high), err = INFINITY;
  for (int it = 0; it < 30 && err > 1e-5; ++it) {
    params = 0.5 * (high + low);
    const double value = ess_zero(ll, n, params, prev, alpha, sm);
    if (sign * value > 0) high = params; else low = params;
    err = fabs(value);
  }
  if (threadIdx.x == 0) out[0] = params;
}

void launch_beta(double prev, const double* ll, int n, double alpha, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(beta_kernel, dim3(1), dim3(BETA_THREADS), 0, stream, prev, ll, n, alpha, out);
}
