// N2: sample-quality metrics on the device -- kernelised Stein discrepancy (inverse multi-quadric kernel, U- and
// V-statistic) and maximum mean discrepancy (RBF kernel).  Replaces mcmc_utils.py:28-85 (stein_disc) and :88-111
// (max_mean_disc), called by exe_flow_matching.py:469-487 on the flow samples / resampled samples.
//
// Both are sums of a scalar function of (x_i, x_j[, g_i, g_j]) over all N^2 ordered pairs: an all-pairs tile kernel.
// One workgroup owns a 64 x 64 tile of pairs at a time (256 threads x 4 x 4 pairs); the coordinates are staged through
// LDS in k-chunks of 32, TRANSPOSED ([k][row]) so that the inner loop reads the 4 rows of a thread with one
// ds_read_b128 (the i side is a broadcast).  Squared distances are accumulated directly as sum (x_i - x_j)^2 -- no
// Gram-matrix expansion, so no cancellation -- in float32; pair terms are summed in float32 per thread and tile (16
// terms), then in float64 across tiles, lanes and workgroups (fixed order: deterministic).  The reference evaluates
// everything in float64; stated tolerance of the parity tests: 1e-5 relative.
//
// Bound: VALU (5 flop-instructions per pair and coordinate for the Stein terms); HBM / L2 traffic is negligible
// (N d floats re-read N / 64 times from L2).
#include "common.hip.h"

#define PAIR_T 64          // tile edge (pairs)
#define PAIR_KC 32         // coordinates per LDS chunk

template <int MODE>        // 0: Stein (x and grad log p), 1: RBF (x only)
__global__ __launch_bounds__(256) void pair_sum_kernel(const float* __restrict__ A, const float* __restrict__ GA,
                                                        const float* __restrict__ B, const float* __restrict__ GB, int na, int nb,
                                                        int d, int tiles_per_chunk, float beta, double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float sXi[PAIR_KC][PAIR_T], sXj[PAIR_KC][PAIR_T];
  __shared__ __attribute__((aligned(16))) float sGi[MODE == 0 ? PAIR_KC : 1][PAIR_T], sGj[MODE == 0 ? PAIR_KC : 1][PAIR_T];
  __shared__ double red[8];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int i0 = blockIdx.x * PAIR_T;
  const int jt_lo = blockIdx.y * tiles_per_chunk, nj_tiles = (nb + PAIR_T - 1) / PAIR_T;
  const int jt_hi = jt_lo + tiles_per_chunk < nj_tiles ? jt_lo + tiles_per_chunk : nj_tiles;
  const int lrow = t & 63, lkq = t >> 6;        // loader role: row of the tile, k quad
  double s_all = 0.0, s_diag = 0.0;
  for (int jt = jt_lo; jt < jt_hi; ++jt) {
    const int j0 = jt * PAIR_T;
    float r2[4][4], gx[4][4], gg[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) { r2[a][b] = 0.f; gx[a][b] = 0.f; gg[a][b] = 0.f; }
    for (int k0 = 0; k0 < d; k0 += PAIR_KC) {
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = (lkq + 4 * h) * 4, k = k0 + kk;
        float xi[4] = {0, 0, 0, 0}, xj[4] = {0, 0, 0, 0}, gi[4] = {0, 0, 0, 0}, gj[4] = {0, 0, 0, 0};
        const int ri = i0 + lrow, rj = j0 + lrow;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < d) {
            if (ri < na) { xi[e] = A[(size_t)ri * d + k + e]; if (MODE == 0) gi[e] = GA[(size_t)ri * d + k + e]; }
            if (rj < nb) { xj[e] = B[(size_t)rj * d + k + e]; if (MODE == 0) gj[e] = GB[(size_t)rj * d + k + e]; }
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sXi[kk + e][lrow] = xi[e]; sXj[kk + e][lrow] = xj[e];
          if (MODE == 0) { sGi[kk + e][lrow] = gi[e]; sGj[kk + e][lrow] = gj[e]; }
        }
      }
      __syncthreads();
#pragma unroll 8
      for (int kk = 0; kk < PAIR_KC; ++kk) {
        const f32x4 xi = *reinterpret_cast<const f32x4*>(&sXi[kk][4 * ty]), xj = *reinterpret_cast<const f32x4*>(&sXj[kk][4 * tx]);
        f32x4 gi = {0, 0, 0, 0}, gj = {0, 0, 0, 0};
        if (MODE == 0) { gi = *reinterpret_cast<const f32x4*>(&sGi[kk][4 * ty]); gj = *reinterpret_cast<const f32x4*>(&sGj[kk][4 * tx]); }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const float df = xi[a] - xj[b];
            r2[a][b] = fmaf(df, df, r2[a][b]);
            if (MODE == 0) {
              gx[a][b] = fmaf(gi[a] - gj[b], df, gx[a][b]);
              gg[a][b] = fmaf(gi[a], gj[b], gg[a][b]);
            }
          }
      }
    }
    float tile_all = 0.f, tile_diag = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = i0 + 4 * ty + a, j = j0 + 4 * tx + b;
        if (i < na && j < nb) {
          float term;
          if (MODE == 0) {
            // mcmc_utils.py:69-77 with beta <- -beta (:54):  -4b(b+1) r2 / (1+r2)^(b+2) + 2b (d + (g-g').(x-x')) / (1+r2)^(1+b)
            //                                               + g.g' / (1+r2)^b
            const float base = 1.f + r2[a][b];
            const float pb = powf(base, -beta);               // (1 + r2)^-b
            const float inv = 1.f / base;
            term = -4.f * beta * (beta + 1.f) * r2[a][b] * pb * inv * inv + 2.f * beta * ((float)d + gx[a][b]) * pb * inv + gg[a][b] * pb;
            if (i == j) tile_diag += term;
          } else {
            term = expf(-0.5f * r2[a][b]);                    // mcmc_utils.py:98-100, sigma2 = 1
          }
          tile_all += term;
        }
      }
    s_all += (double)tile_all; s_diag += (double)tile_diag;
  }
  // workgroup reduction in a fixed order
  s_all = wave_sum(s_all); s_diag = wave_sum(s_diag);
  __syncthreads();
  if ((t & 63) == 0) { red[t >> 6] = s_all; red[4 + (t >> 6)] = s_diag; }
  __syncthreads();
  if (t == 0) {
    const size_t w = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    part[2 * w] = (red[0] + red[1]) + (red[2] + red[3]);
    part[2 * w + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

__global__ void pair_reduce_kernel(const double* part, int n, double* out) {
  __shared__ double sm[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
  sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sm[0][threadIdx.x] += sm[0][threadIdx.x + o]; sm[1][threadIdx.x] += sm[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sm[0][0]; out[1] = sm[1][0]; }
}

// sum over all (i, j) of the pair function -> out[0] (all pairs), out[1] (i == j pairs; Stein mode only).  `part` must hold
// 2 * pair_sum_parts(na, nb) doubles.
static void pair_grid(int na, int nb, dim3& grid, int& tiles_per_chunk) {
  const int ni = (na + PAIR_T - 1) / PAIR_T, nj = (nb + PAIR_T - 1) / PAIR_T;
  int chunks = 2048 / ni; if (chunks < 1) chunks = 1; if (chunks > nj) chunks = nj;
  tiles_per_chunk = (nj + chunks - 1) / chunks;
  chunks = (nj + tiles_per_chunk - 1) / tiles_per_chunk;
  grid = dim3(ni, chunks);
}
static size_t pair_sum_parts(int na, int nb) { dim3 g; int t; pair_grid(na, nb, g, t); return (size_t)g.x * g.y; }
static void launch_pair_sum(int mode, const float* A, const float* GA, const float* B, const float* GB, int na, int nb, int d, float beta,
                            double* part, double* out, hipStream_t stream) {
  dim3 grid; int tpc;
  pair_grid(na, nb, grid, tpc);
  if (mode == 0) hipLaunchKernelGGL(pair_sum_kernel<0>, grid, dim3(256), 0, stream, A, GA, B, GB, na, nb, d, tpc, beta, part);
  else hipLaunchKernelGGL(pair_sum_kernel<1>, grid, dim3(256), 0, stream, A, GA, B, GB, na, nb, d, tpc, beta, part);
  hipLaunchKernelGGL(pair_reduce_kernel, dim3(1), dim3(256), 0, stream, part, (int)(grid.x * grid.y), out);
}
