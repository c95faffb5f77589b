// K1/K2 for the log-Gaussian Cox process target (distributions.py:231-314, cox_process_utils.py:98-165): the one target
// whose gradient needs a dense contraction, K^-1 (x - mu).  One workgroup per tile of 16 chains; the proposal tile
// sits in LDS and is multiplied by the packed K^-1 with the same MFMA tile GEMM the MLP kernels use (mlp.hip.h), so
// K^-1 is streamed once per 16 chains; everything else (noise, energies, accept) is fused around it.
//   loglik(x)   = sum_i (x_i c_i - a exp(x_i))                      a = 1/d              (cox_process_utils.py:113-115)
//   logprior(x) = -1/2 (x - mu)^T K^-1 (x - mu) + log_norm                              (distributions.py:299-303)
//   grad        = beta (c - a exp(x)) - K^-1 (x - mu)
// mode 0: mala_init (value and gradient at the given positions); 1: one MALA step (mala.py:86-118, as written).
#include "mlp.hip.h"
#include "prng.hip.h"

#define LGCP_NW 8

struct LgcpArgs {
  TargetDev T; int dp;
  int mode;
  Key2 key; const uint32_t* keys; uint32_t n_total, chain_offset;
  int B; double beta, eps; int textbook;
  float* pos; double* logp; float* grad;
  float* acc_prob; uint8_t* accepted; float* proposed; float* prop_weight;
};

template <int TPW>
__global__ __launch_bounds__(LGCP_NW * 64) void mala_lgcp_kernel(LgcpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const int d = a.T.dim, dp = a.dp, ld = dp + 4, b0 = blockIdx.x * 16;
  float* bU = lds;                                   // [16][ld] proposal positions
  double* red = reinterpret_cast<double*>(lds + 16 * ld);   // [3][LGCP_NW][16]
  float x[TPW][4], gr[TPW][4], xn[TPW][4];
  double th1[4] = {0, 0, 0, 0};
  Key2 k_int[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bi = b0 + 4 * g + i;
    const Key2 kb = a.keys ? Key2{a.keys[2 * bi], a.keys[2 * bi + 1]} : split_at(a.key, a.n_total, a.chain_offset + (uint32_t)bi);     // exe_flow_matching.py:303
    k_int[i] = split_at(kb, 2, 0);                                                                // mala.py:93
  }
  double lp_old[4];        // read BEFORE any wave can publish an accepted log-density for the same chain
#pragma unroll
  for (int i = 0; i < 4; ++i) lp_old[i] = a.mode == 1 ? a.logp[b0 + 4 * g + i] : 0.0;
  const double s2e = sqrt(2.0 * a.eps);
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (wave + LGCP_NW * q) * 16 + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      x[q][i] = gr[q][i] = xn[q][i] = 0.f;
      if (col < d) {
        const size_t o = (size_t)(b0 + row) * d + col;
        x[q][i] = a.pos[o];
        if (a.mode == 1) {
          gr[q][i] = a.grad[o];
          const double th = s2e * normal64(k_int[i], (uint32_t)col, (uint32_t)d);                 // util.py:80-82
          th1[i] += th * th;
          xn[q][i] = (float)((double)x[q][i] + a.eps * (double)gr[q][i] + th);                    // diffusions.py:25-30
        } else {
          xn[q][i] = x[q][i];
        }
      }
      if (col < dp) bU[row * ld + col] = xn[q][i];
    }
  }
  __syncthreads();
  // y = K^-1 (x' - mu) through the tile GEMM; the epilogue finishes gradient, likelihood and both quadratic forms
  float gn[TPW][4];
  double lik[4] = {0, 0, 0, 0}, quad[4] = {0, 0, 0, 0}, th2[4] = {0, 0, 0, 0};
  layer_gemm<1, LGCP_NW, 2>(bU, ld, a.T.KinvP, a.T.kbias, dp / 16, dp / 16, wave, lane,
                            [&](int q, int nt, int m, f32x4 acc, float kb) {
                              const int col = nt * 16 + c;
#pragma unroll
                              for (int i = 0; i < 4; ++i) {
                                float xv = 0.f, xo = 0.f, gv = 0.f;
#pragma unroll
                                for (int qq = 0; qq < TPW; ++qq)
                                  if (qq == q) { xv = xn[qq][i]; xo = x[qq][i]; }
                                if (col < d) {
                                  const float y = acc[i] + kb;
                                  const float ex = expf(xv);
                                  gv = (float)a.beta * (a.T.counts[col] - a.T.poisson_a * ex) - y;
                                  lik[i] += (double)xv * (double)a.T.counts[col] - (double)a.T.poisson_a * (double)ex;
                                  quad[i] += (double)(xv - a.T.mu) * (double)y;
                                  const double t = (double)xo - (double)xv - a.eps * (double)gv;
                                  th2[i] += t * t;
                                }
#pragma unroll
                                for (int qq = 0; qq < TPW; ++qq)
                                  if (qq == q) gn[qq][i] = gv;
                              }
                            });
  // row sums over the tile's columns (float64)
  auto reduce3 = [&](double (&v)[4], int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
      if (c == 0) red[(slot * LGCP_NW + wave) * 16 + 4 * g + i] = v[i];
    }
  };
  reduce3(lik, 0); reduce3(quad, 1); reduce3(th2, 2);
  double t1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) th1[i] += __shfl_xor(th1[i], o, 64);
    t1[i] = th1[i];
  }
  __syncthreads();
  double* red1 = red + 3 * LGCP_NW * 16;
  if (c == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red1[wave * 16 + 4 * g + i] = t1[i];
  }
  __syncthreads();
  bool acc[4];
  double lpn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * g + i, b = b0 + row;
    double sl = 0, sq = 0, s2 = 0, s1 = 0;
#pragma unroll
    for (int w = 0; w < LGCP_NW; ++w) {
      sl += red[(0 * LGCP_NW + w) * 16 + row]; sq += red[(1 * LGCP_NW + w) * 16 + row];
      s2 += red[(2 * LGCP_NW + w) * 16 + row]; s1 += red1[w * 16 + row];
    }
    lpn[i] = a.beta * sl - 0.5 * sq + (double)a.T.log_norm;
    acc[i] = true;
    if (a.mode == 1) {
      const double lp = lp_old[i], inv4e = 0.25 / a.eps;
      const double new_E = -lp + inv4e * s1, prev_E = -lpn[i] + inv4e * s2;          // mala.py:68-79, proposal.py:157-158
      double delta = prev_E - new_E;                                                 // proposal.py:104
      if (a.textbook) delta = -delta;
      if (isnan(delta)) delta = -INFINITY;                                           // proposal.py:105
      const double p = fmin(exp(delta), 1.0);                                        // proposal.py:178
      const Key2 kb = a.keys ? Key2{a.keys[2 * b], a.keys[2 * b + 1]} : split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b);
      acc[i] = uniform01(split_at(kb, 2, 1), 0, 1) < p;                              // proposal.py:179
      if (wave == 0 && c == 0) {
        if (a.acc_prob) a.acc_prob[b] = (float)p;
        if (a.accepted) a.accepted[b] = acc[i] ? 1 : 0;
        if (a.prop_weight) a.prop_weight[b] = (float)exp(lpn[i] + inv4e * s2);       // mala.py:104-113
      }
    }
    if (wave == 0 && c == 0 && acc[i]) a.logp[b] = lpn[i];
  }
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (wave + LGCP_NW * q) * 16 + c;
    if (col < d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t o = (size_t)(b0 + 4 * g + i) * d + col;
        if (a.mode == 1 && a.proposed) a.proposed[o] = xn[q][i];
        if (acc[i]) { if (a.mode == 1) a.pos[o] = xn[q][i]; a.grad[o] = gn[q][i]; }
      }
    }
  }
}

int launch_mala_lgcp(const LgcpArgs& a, hipStream_t stream) {
  const int tpw = (a.dp / 16 + LGCP_NW - 1) / LGCP_NW;
  const size_t sm = (size_t)(16 * (a.dp + 4)) * 4 + (size_t)(4 * LGCP_NW * 16) * 8;
  if (sm > 160 * 1024 || a.B % 16) return -3;
  dim3 grid(a.B / 16), block(LGCP_NW * 64);
#define LG_LAUNCH(T) do { (void)hipFuncSetAttribute((const void*)mala_lgcp_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                          hipLaunchKernelGGL(mala_lgcp_kernel<T>, grid, block, sm, stream, a); } while (0)
  if (tpw <= 1) LG_LAUNCH(1); else if (tpw <= 2) LG_LAUNCH(2); else if (tpw <= 4) LG_LAUNCH(4); else if (tpw <= 8) LG_LAUNCH(8); else return -3;
#undef LG_LAUNCH
  return 0;
}
