// N4: the device pieces of the adaptive tempered SMC baseline (exe_others.py:79-111 -> bblackjax/smc/*), which drives the
// same MALA kernels as the MFM loop from a second caller.  Everything here is O(n) float64 work on one workgroup
// (n = number of particles, a few thousand): latency-bound, kept on the device so a step needs no host round trip
// beyond the scalar temperature.
//   smc_delta_kernel    ess.py:46-89 (ess_solver; AS WRITTEN the weights are exp(-delta * loglik)) + solver.py:20-82 (dichotomy)
//   smc_weights_kernel  base.py:125-128 (normalised weights, log normalising constant) with weigh_fn = delta * loglik (tempered.py:118-119)
//   smc_resample_kernel resampling.py:50-52,124-135 (systematic: ONE uniform, cumsum, searchsorted(left), clip)
//   gather_rows_kernel  base.py:120 (particles[resampling_idx])
#include "common.hip.h"
#include "prng.hip.h"

#define SMC_THREADS 1024

__device__ double smc_block_reduce(double v, double* sm, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = sm[0];
  for (int w = 1; w < SMC_THREADS / 64; ++w) r = is_max ? fmax(r, sm[w]) : r + sm[w];
  return r;
}
__device__ __forceinline__ double nan_to_num(double v) {        // jnp.nan_to_num defaults
  if (isnan(v)) return 0.0;
  if (isinf(v)) return v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
  return v;
}
// log_ess(nan_to_num(-delta * ll)) - log(n * target)      (ess.py:28-43,82-86)
__device__ double smc_fun(const double* ll, int n, double delta, double target_val, double* sm) {
  double m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) m = fmax(m, nan_to_num(-delta * ll[i]));
  m = smc_block_reduce(m, sm, true);
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) { const double lw = nan_to_num(-delta * ll[i]) - m; s1 += exp(lw); s2 += exp(2.0 * lw); }
  s1 = smc_block_reduce(s1, sm, false);
  s2 = smc_block_reduce(s2, sm, false);
  return 2.0 * (m + log(s1)) - (2.0 * m + log(s2)) - target_val;
}
__global__ __launch_bounds__(SMC_THREADS) void smc_delta_kernel(const double* ll, int n, double target_ess, double max_delta, double* out) {
  __shared__ double sm[SMC_THREADS / 64];
  const double target_val = log((double)n * target_ess);
  double a = 0.0, b = max_delta;
  double f_a = smc_fun(ll, n, a, target_val, sm), f_b = smc_fun(ll, n, b, target_val, sm);
  double res;
  if (f_b > 0) res = max_delta;                                   // solver.py:76-81
  else if (f_a > 0) {
    for (int i = 0; i < 100 && f_a - f_b > 1e-4; ++i) {           // solver.py:45-62 (eps = 1e-4, max_iter = 100)
      const double mid = 0.5 * (a + b);
      const double f_mid = smc_fun(ll, n, mid, target_val, sm);
      if (f_mid < 0) { b = mid; f_b = f_mid; } else { a = mid; f_a = f_mid; }
    }
    res = a;
  } else res = NAN;
  if (threadIdx.x == 0) out[0] = fmin(fmax(res, 0.0), max_delta);      // adaptive_tempered.py:70 (clip; NaN propagates)
}

__global__ __launch_bounds__(SMC_THREADS) void smc_weights_kernel(const double* ll, int n, double delta, double* weights, double* lognorm) {
  __shared__ double sm[SMC_THREADS / 64];
  double m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) m = fmax(m, delta * ll[i]);
  m = smc_block_reduce(m, sm, true);
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) s += exp(delta * ll[i] - m);
  s = smc_block_reduce(s, sm, false);
  const double logsum = m + log(s);
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) weights[i] = exp(delta * ll[i] - logsum);
  if (threadIdx.x == 0) lognorm[0] = logsum - log((double)n);
}

// cumsum by ONE thread in index order (bit-identical to a sequential numpy cumsum: the resampling indices are integer
// outputs and must not depend on a scan tree), then one binary search per output.
__global__ __launch_bounds__(SMC_THREADS) void smc_resample_kernel(Key2 key, const double* weights, int n, double* cum, int* idx) {
  if (threadIdx.x == 0) {
    double c = 0.0;
    for (int i = 0; i < n; ++i) { c += weights[i]; cum[i] = c; }
  }
  __syncthreads();
  const double u = uniform01(key, 0, 1);                          // jax.random.uniform(rng_key, ())
  for (int j = threadIdx.x; j < n; j += SMC_THREADS) {
    const double v = ((double)j + u) / (double)n;                 // resampling.py:133
    int lo = 0, hi = n;                                           // searchsorted, side = 'left': first i with cum[i] >= v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] < v) lo = mid + 1; else hi = mid;
    }
    idx[j] = lo < n - 1 ? lo : n - 1;                             // :135
  }
}

// The other cumulative-sum schemes of resampling.py: scheme 1 = stratified (:55-57: one uniform PER output, uniform(key, (n,))),
// scheme 2 = multinomial (:60-80: Chopin's sorted uniforms z[:-1] / z[-1], z = cumsum(-log uniform(key, (n + 1,)))).  cum: 2 n + 2
// doubles (cumulative weights, then the sorted uniforms' cumulative sum); both sums by ONE thread in index order, as above.
__global__ __launch_bounds__(SMC_THREADS) void smc_resample2_kernel(int scheme, Key2 key, const double* weights, int n, double* cum, int* idx) {
  double* z = cum + n;
  if (scheme == 2)
    for (int j = threadIdx.x; j <= n; j += SMC_THREADS) z[j] = -log(uniform01(key, (uint32_t)j, (uint32_t)(n + 1)));      // :149
  __syncthreads();
  if (threadIdx.x == 0) {
    double c = 0.0;
    for (int i = 0; i < n; ++i) { c += weights[i]; cum[i] = c; }
    if (scheme == 2) { double zz = 0.0; for (int j = 0; j <= n; ++j) { zz += z[j]; z[j] = zz; } }                         // :150
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += SMC_THREADS) {
    const double v = scheme == 2 ? z[j] / z[n]                                                          // :151
                                 : ((double)j + uniform01(key, (uint32_t)j, (uint32_t)n)) / (double)n;  // :131-133
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] < v) lo = mid + 1; else hi = mid;
    }
    idx[j] = lo < n - 1 ? lo : n - 1;
  }
}

// jax.random.choice(key, n, (m,), replace=True, p = exp(logw - max logw)) -- the self-normalised importance resampling of the
// final flow samples (exe_flow_matching.py:458-459).  jax: p_cuml = cumsum(p); r = p_cuml[-1] * (1 - uniform(key, (m,)));
// searchsorted(p_cuml, r) (side = 'left').  Sequential cumulative sum for the same reason as above.
__global__ __launch_bounds__(SMC_THREADS) void choice_logw_kernel(Key2 key, const double* logw, int n, int m, double* cum, int* idx) {
  __shared__ double sm[SMC_THREADS / 64];
  double mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) mx = fmax(mx, logw[i]);          // jnp.max: a NaN weight poisons the draw there too
  mx = smc_block_reduce(mx, sm, true);
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) cum[i] = exp(logw[i] - mx);      // :458
  __syncthreads();
  if (threadIdx.x == 0) {
    double c = 0.0;
    for (int i = 0; i < n; ++i) { c += cum[i]; cum[i] = c; }
  }
  __syncthreads();
  const double tot = cum[n - 1];
  for (int j = threadIdx.x; j < m; j += SMC_THREADS) {
    const double r = tot * (1.0 - uniform01(key, (uint32_t)j, (uint32_t)m));
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] < r) lo = mid + 1; else hi = mid;
    }
    idx[j] = lo < n - 1 ? lo : n - 1;
  }
}

// sum and sum of squares (float64) of a per-chain float32 quantity: the acceptance statistics logged every iteration
// (exe_flow_matching.py:442-443) without a handful of framework launches per iteration
__global__ __launch_bounds__(SMC_THREADS) void acc_stats_kernel(const float* x, int n, double* out) {
  __shared__ double sm[SMC_THREADS / 64];
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += SMC_THREADS) { const double v = x[i]; s1 += v; s2 += v * v; }
  s1 = smc_block_reduce(s1, sm, false);
  s2 = smc_block_reduce(s2, sm, false);
  if (threadIdx.x == 0) { out[0] = s1; out[1] = s2; }
}

__global__ void gather_rows_kernel(const float* src, const int* idx, int n, int d, float* dst) {
  const size_t tot = (size_t)n * d;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const size_t r = i / d;
    dst[i] = src[(size_t)idx[r] * d + (i - r * d)];
  }
}
