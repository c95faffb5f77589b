// The d = 2 exact-trace solver on FOUR-chain tiles (the mixture examples with few chains: BASELINE configs[0], 512 chains).
//
// Same arithmetic per entry and the same Dormand-Prince state machine as the generic tile (ode.hip: OdeTile::eval_x2 /
// ode_solve; exe_flow_matching.py:206-242, :246-278, jax.experimental.ode.odeint restated in oracle/ode.py); what changes
// is the tile.  The generic tile puts 16 chains into a workgroup and pushes values + the tangents of both basis vectors
// through every layer as M = 48 rows: 512 chains are 32 workgroups on 256 CUs, each three MFMA row tiles deep per streamed
// weight fragment in a latency-bound adaptive solve.  Here a workgroup owns FOUR chains and ONE 16-row MFMA tile:
//
//   M-row 4 j + 0 : value row of chain j          M-row 4 j + 2 : tangent of e_2
//   M-row 4 j + 1 : tangent of e_1                M-row 4 j + 3 : unused (zero)
//
// In the f32 accumulator layout lane (g, c) holds rows 4 g .. 4 g + 3 of column c: the value's pre-activation and both
// tangents of the SAME chain sit in one lane, so the activation masks need no cross-lane traffic.  512 chains = 128
// workgroups, a third of the matrix work per evaluation.  With d = 2 the whole Runge-Kutta state of a chain (y, log-det, seven
// stage derivatives of each, t, dt, counters) is ~40 registers, so EVERY lane of group g carries chain g's state redundantly
// and runs its step-size controller itself: no LDS row-state block, no cross-wave reductions for the error norms.  The two
// narrow layers (d -> hx1 with K = 2; hj2 -> d and the gate ht2 -> d with N = 2) do not go through full MFMA tiles: the first
// is two FMAs per output on the vector ALU, the last two are split over the waves along K (one k-block of 16 per wave) and
// summed through LDS in a fixed order.
//
// Differences to the generic tile are float reassociations only (K-split sums of the out / gate layers, FMA of the K = 2
// layer); the replay instrumentation (mfm_debug_replay) is carried so the same step-for-step parity tests run on both.
namespace d2 {

constexpr int NW = 8;

struct Lds { int ff, ldff, t1, ldt1, x1, ldx1, cat, ldcat, j1, ldj1, j2, ldj2, part, total; };
__host__ __device__ inline Lds layout(const NetDev& n) {
  Lds L; int o = 0;
  auto take = [&](int cnt) { int r = o; o += cnt; return r; };
  L.ldff = n.F2p + 8; L.ff = take(16 * L.ldff);
  L.ldt1 = n.ht1 + 8; L.t1 = take(16 * L.ldt1);
  L.ldx1 = n.hx1 + 8; L.x1 = take(16 * L.ldx1);
  L.ldcat = n.hx2 + n.ht2 + 8; L.cat = take(16 * L.ldcat);
  L.ldj1 = n.hj1 + 8; L.j1 = take(16 * L.ldj1);
  L.ldj2 = n.hj2 + 8; L.j2 = take(16 * L.ldj2);
  L.part = take(4 * NW * 8);          // [chain][wave][8]: K-split partial sums of the out / gate layers
  L.total = o;
  return L;
}

static bool shape_ok(const NetDev& n, int hutch) {
  if (n.d != 2 || hutch || n.T.kind != MFM_TARGET_GMM || net_ragged(n)) return false;
  if (n.F % 16 || n.F2p != 2 * n.F) return false;
  if (n.hx1 / 16 > 2 * NW) return false;
  return (size_t)layout(n).total * sizeof(float) <= 160 * 1024;
}

struct Tile {
  static constexpr int CHAINS = 4;                 // chains per workgroup
#ifdef MFM_STAMPS
  unsigned long long cyc[4] = {0, 0, 0, 0};
#endif
  __device__ __forceinline__ int chain() const { return blockIdx.x * 4 + g; }        // the chain this lane carries
  __device__ __forceinline__ bool writer() const { return wave == 0 && c == 0; }     // ONE lane per chain
  const NetDev* n;
  Lds L;
  float* lds;
  int lane, wave, g, c, sign;
  float w1[2][2], b1[2];       // x1 layer (K = 2): both rows of W_x1 and the bias at this lane's column of the wave's tiles
  float b7[2], b4[2];          // out / gate bias of both columns
  float gate[2];               // nn_t of the last evaluated stage time (stages 6 and 7 share it)

  __device__ __forceinline__ void init(const NetDev* net, float* l) {
    n = net; L = layout(*net); lds = l;
    lane = threadIdx.x & 63; wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); g = lane >> 4; c = lane & 15; sign = 1;
    for (int i = threadIdx.x; i < L.total; i += NW * 64) lds[i] = 0.f;       // unused rows / tangent rows of st stay zero
    const LayerDesc& l2 = n->L[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int nt = wave + NW * q;
      const bool in = nt < l2.Np / 16;
      w1[q][0] = in ? n->Wp[l2.w_off + pack_index(0, nt * 16 + c, l2.Kp / 16)] : 0.f;
      w1[q][1] = in ? n->Wp[l2.w_off + pack_index(1, nt * 16 + c, l2.Kp / 16)] : 0.f;
      b1[q] = in ? n->bias[l2.b_off + nt * 16 + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { b7[j] = n->bias[n->L[7].b_off + j]; b4[j] = n->bias[n->L[4].b_off + j]; gate[j] = 0.f; }
    __syncthreads();
  }

  // value + two tangent rows of one chain through a hidden layer: everything a lane needs is in its own accumulator
  __device__ __forceinline__ void three(const float* A, int lda, int layer, float* out, int ldo) {
    const NetDev& N = *n;
    layer_gemm<1, NW, 1>(A, lda, N.Wp + N.L[layer].w_off, N.bias + N.L[layer].b_off, N.L[layer].Kp / 16, N.L[layer].Np / 16, wave, lane,
                         [&](int q, int nt, int m, f32x4 acc, float b) {
                           const float pre = acc[0] + b;
                           float* o = out + (4 * g) * ldo + nt * 16 + c;
                           o[0] = act_f(pre, N.act);
                           o[ldo] = mask_pre(pre, acc[1], N.act);
                           o[2 * ldo] = mask_pre(pre, acc[2], N.act);
                         });
  }

  // (hooks of the resident-weight tile below: the streamed tile evaluates its time branch inside every evaluation)
  __device__ __forceinline__ void prepare(const float (&)[5]) {}

  // One evaluation of the augmented field for chain g at x = (x0, x1), time tt.  kv = dx/dt, dl = d(logdet)/dt.
  __device__ __forceinline__ void eval(float x0, float x1, float tt, float (&kv)[2], float& dl, int phase) {
    const NetDev& N = *n;
    const bool reuse_time = phase == 7;       // stages 6 and 7 are both at t + dt
    // ---- Fourier features of the value rows (:70-71), the K = 2 layer on the vector ALU, the mixture's gradient ----
    if (!reuse_time) {
      for (int nt = wave; nt < N.F / 16; nt += NW) {
        const int col = nt * 16 + c;
        const double te = sign > 0 ? (double)tt : 1.0 - (double)tt;          // :229
        double ft = (double)N.fourier[col] * te;
        ft -= rint(ft);
        float sv, cv;
        sincospif(2.f * (float)ft, &sv, &cv);
        lds[L.ff + (4 * g) * L.ldff + col] = cv;
        lds[L.ff + (4 * g) * L.ldff + N.F + col] = sv;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int nt = wave + NW * q;
      if (nt < N.hx1 / 16) {
        const float pre = fmaf(x1, w1[q][1], x0 * w1[q][0]) + b1[q];
        float* o = lds + L.x1 + (4 * g) * L.ldx1 + nt * 16 + c;
        o[0] = act_f(pre, N.act);
        o[L.ldx1] = mask_pre(pre, w1[q][0], N.act);
        o[2 * L.ldx1] = mask_pre(pre, w1[q][1], N.act);
      }
    }
    float gc[2], hd[2];              // clip(grad log pi), masked diagonal of its Jacobian: H_11, H_22
    {
      const float xr[2] = {x0, x1}, e1[2] = {1.f, 0.f}, e2[2] = {0.f, 1.f};
      double lp; float gg[2], h1[2], h2[2];
      if (N.T.n_modes <= 16) { gmm_eval_lanes16<2>(N.T, xr, c, &lp, gg, e1, h1); gmm_eval_lanes16<2>(N.T, xr, c, &lp, gg, e2, h2); }
      else { gmm_eval<2>(N.T, xr, &lp, gg, e1, h1); gmm_eval<2>(N.T, xr, &lp, gg, e2, h2); }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool inside = !(N.grad_clip > 0.f) || fabsf(gg[j]) <= N.grad_clip;
        gc[j] = clipf(gg[j], N.grad_clip);
        hd[j] = inside ? (j == 0 ? h1[0] : h2[1]) : 0.f;
      }
    }
    __syncthreads();
    // ---- t1 ; x2 ----
    if (!reuse_time)
      layer_gemm<1, NW, 1>(lds + L.ff, L.ldff, N.Wp + N.L[0].w_off, N.bias + N.L[0].b_off, N.L[0].Kp / 16, N.L[0].Np / 16, wave, lane,
                           [&](int q, int nt, int m, f32x4 acc, float b) { lds[L.t1 + (4 * g) * L.ldt1 + nt * 16 + c] = act_f(acc[0] + b, N.act); });
    three(lds + L.x1, L.ldx1, 3, lds + L.cat, L.ldcat);
    __syncthreads();
    // ---- t2 -> st ----
    if (!reuse_time) {
      layer_gemm<1, NW, 1>(lds + L.t1, L.ldt1, N.Wp + N.L[1].w_off, N.bias + N.L[1].b_off, N.L[1].Kp / 16, N.L[1].Np / 16, wave, lane,
                           [&](int q, int nt, int m, f32x4 acc, float b) { lds[L.cat + (4 * g) * L.ldcat + N.hx2 + nt * 16 + c] = act_f(acc[0] + b, N.act); });
      __syncthreads();
    }
    three(lds + L.cat, L.ldcat, 5, lds + L.j1, L.ldj1);
    __syncthreads();
    three(lds + L.j1, L.ldj1, 6, lds + L.j2, L.ldj2);
    __syncthreads();
    // ---- out (and the gate): one k-block of 16 per wave, partial sums through LDS ----
    {
      const int r = lane & 15;
      auto ksplit = [&](const float* A, int lda, int layer) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const f32x4* W = reinterpret_cast<const f32x4*>(N.Wp + N.L[layer].w_off);
        for (int kb = wave; kb < N.L[layer].Kp / 16; kb += NW) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(A + r * lda + 4 * g + kb * 16);
          const f32x4 b = W[kb * 64 + lane];
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
        }
        return acc;
      };
      const f32x4 ao = ksplit(lds + L.j2, L.ldj2, 7);
      float* p = lds + L.part + (g * NW + wave) * 8;
      if (c == 0) { p[0] = ao[0]; p[2] = ao[1]; }
      if (c == 1) { p[1] = ao[0]; p[3] = ao[2]; }
      if (!reuse_time) {
        const f32x4 ag = ksplit(lds + L.cat + N.hx2, L.ldcat, 4);
        if (c < 2) p[4 + c] = ag[0];
      }
    }
    __syncthreads();
    float o0 = 0.f, o1 = 0.f, j11 = 0.f, j22 = 0.f, g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + L.part + (g * NW + w) * 8);
      const f32x4 pb = *reinterpret_cast<const f32x4*>(lds + L.part + (g * NW + w) * 8 + 4);
      o0 += pa[0]; o1 += pa[1]; j11 += pa[2]; j22 += pa[3]; g0 += pb[0]; g1 += pb[1];
    }
    if (!reuse_time) { gate[0] = g0 + b4[0]; gate[1] = g1 + b4[1]; }
    // v = nn_xt + nn_t * clip(grad log pi(x)) (:88-90);  trace J = sum_j (d nn_xt e_j)_j + nn_t_j 1[|g_j| <= clip] H_jj
    const float v0 = o0 + b7[0] + gate[0] * gc[0], v1 = o1 + b7[1] + gate[1] * gc[1];
    const float tr = (j11 + gate[0] * hd[0]) + (j22 + gate[1] * hd[1]);
    kv[0] = sign > 0 ? v0 : -v0; kv[1] = sign > 0 ? v1 : -v1;
    dl = sign > 0 ? -tr : tr;                                                 // :218 / :239
  }
};


// ---- the resident-weight tile ------------------------------------------------------------------------------------------------
// For the reference's default widths (F = 128 Fourier frequencies, every hidden layer 128 wide: multi_modal.py:159,178-180) the
// eight waves of the workgroup own one 16-column tile of each layer.  Two consequences:
//  * the B fragments of the three hidden layers an evaluation passes on its x path (x2, the x half of j1, j2: 3 x 8 float4)
//    and the wave's k-block of the out layer stay IN REGISTERS for the whole kernel (100 VGPRs): an evaluation issues no
//    global loads at all -- LDS reads, MFMAs, five barriers;
//  * everything that depends on t only -- Fourier features, t1, t2, the gate and the st half of j1's pre-activation (with
//    its bias) -- is evaluated ONCE PER ATTEMPT for the five distinct stage times (t + c_s dt, known when the attempt starts)
//    of the four chains as one M = 20 (two row tiles) batch with streamed weights, and kept in LDS: 10 KB of j1 accumulator
//    seeds, 40 gate values.  An evaluation is then x1 (vector ALU) -> x2 -> j1 (K = 128, seeded) -> j2 -> out (K split).
// Per attempt: 6 x 100 + 260 MFMAs per wave instead of 6 x 232.  Float reassociations against the streamed tile: the j1
// pre-activation sums its st half (and bias) first.
struct LdsR { int ff, t1, st, ct, gate, gp, x1, cat, j1, j2, part, gc, total; };
constexpr int R_LD = 136, R_LDF = 264;      // leading dimensions (128 + 8, 256 + 8: conflict-free ds_read_b128 fragments)
__host__ __device__ inline LdsR layout_r() {
  LdsR L; int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 3) & ~3; return r; };
  L.ff = take(32 * R_LDF); L.t1 = take(32 * R_LD); L.st = take(32 * R_LD);
  L.ct = take(20 * 128); L.gate = take(20 * 2); L.gp = take(NW * 32 * 2);
  L.x1 = take(16 * R_LD); L.cat = take(16 * R_LD); L.j1 = take(16 * R_LD); L.j2 = take(16 * R_LD);
  L.part = take(4 * NW * 4); L.gc = take(2 * 4 * 4);
  L.total = o;
  return L;
}
static bool shape_ok_r(const NetDev& n, int hutch) {
  return shape_ok(n, hutch) && n.F == 128 && n.ht1 == 128 && n.ht2 == 128 && n.hx1 == 128 && n.hx2 == 128 && n.hj1 == 128 && n.hj2 == 128 &&
         n.T.n_modes <= 16;
}

// grad log pi and the DIAGONAL of its Jacobian for a d = 2 mixture, one mode per lane (k = lane & 15): the two calls of
// gmm_eval_lanes16 with v = e_1, e_2 (targets.hip.h) folded into one pass -- (H e_j)_j = sum_k r_k (a_kj^2 - 1/s_kj^2) - g_j^2.
__device__ __forceinline__ void gmm_grad_hdiag2(const TargetDev& T, float x0, float x1, int k, float (&gg)[2], float (&hd)[2]) {
  const bool live = k < T.n_modes;
  float comp = -INFINITY, a[2] = {0.f, 0.f}, iv[2] = {0.f, 0.f};
  if (live) {
    comp = T.gmm_logw[k];
    const float xs[2] = {x0, x1};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float sd = T.gmm_std[k * 2 + j], dx = xs[j] - T.gmm_mode[k * 2 + j], z = dx / sd;
      comp -= 0.5f * z * z;
      a[j] = -dx / (sd * sd);
      iv[j] = 1.f / (sd * sd);
    }
  }
  const float m = group16_max_dpp(comp);
  const float e = live ? expf(comp - m) : 0.f;
  const float inv = 1.f / group16_sum_dpp(e);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    gg[j] = group16_sum_dpp(e * a[j]) * inv;
    hd[j] = group16_sum_dpp(e * (a[j] * a[j] - iv[j])) * inv - gg[j] * gg[j];
  }
}

template <int ACT>          // the hidden non-linearity as a compile-time constant (MFM_ACT_*), or -1: read from the network
struct TileR {
  static constexpr int CHAINS = 4;
#ifdef MFM_STAMPS
  unsigned long long cyc[4] = {0, 0, 0, 0};
#endif
  __device__ __forceinline__ int chain() const { return blockIdx.x * 4 + g; }
  __device__ __forceinline__ bool writer() const { return wave == 0 && c == 0; }
  __device__ __forceinline__ int act() const { return ACT >= 0 ? ACT : n->act; }
  const NetDev* n;
  LdsR L;
  float* lds;
  int lane, wave, g, c, sign, par;
  float w1[2], b1x, b3, b5, b6, b7[2], b4[2];
  f32x4 W3f[8], W5f[8], W6f[8];
  float w7c[2];                // W_out[col][0..1] at this lane's column of the last hidden layer (D2_OUT_MFMA: its k-block as an MFMA fragment instead)
#ifdef D2_OUT_MFMA
  f32x4 W7f;
#endif

  __device__ __forceinline__ void init(const NetDev* net, float* l) {
    n = net; L = layout_r(); lds = l; par = 0;
    lane = threadIdx.x & 63; wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); g = lane >> 4; c = lane & 15; sign = 1;
    for (int i = threadIdx.x; i < L.total; i += NW * 64) lds[i] = 0.f;       // unused rows stay zero
    const NetDev& N = *n;
    const int col = wave * 16 + c;
    w1[0] = N.Wp[N.L[2].w_off + pack_index(0, col, 1)]; w1[1] = N.Wp[N.L[2].w_off + pack_index(1, col, 1)];
    b1x = N.bias[N.L[2].b_off + col]; b3 = N.bias[N.L[3].b_off + col]; b5 = N.bias[N.L[5].b_off + col]; b6 = N.bias[N.L[6].b_off + col];
#pragma unroll
    for (int j = 0; j < 2; ++j) { b7[j] = N.bias[N.L[7].b_off + j]; b4[j] = N.bias[N.L[4].b_off + j]; }
    const f32x4* P3 = reinterpret_cast<const f32x4*>(N.Wp + N.L[3].w_off);
    const f32x4* P5 = reinterpret_cast<const f32x4*>(N.Wp + N.L[5].w_off);
    const f32x4* P6 = reinterpret_cast<const f32x4*>(N.Wp + N.L[6].w_off);
    const f32x4* P7 = reinterpret_cast<const f32x4*>(N.Wp + N.L[7].w_off);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      W3f[kb] = P3[(wave * 8 + kb) * 64 + lane];
      W5f[kb] = P5[(wave * 16 + kb) * 64 + lane];          // rows 0..127 of the [sx | st] input: the x half
      W6f[kb] = P6[(wave * 8 + kb) * 64 + lane];
    }
#ifdef D2_OUT_MFMA
    W7f = P7[wave * 64 + lane];
#else
    (void)P7;
#endif
    w7c[0] = N.Wp[N.L[7].w_off + pack_index(col, 0, N.L[7].Kp / 16)]; w7c[1] = N.Wp[N.L[7].w_off + pack_index(col, 1, N.L[7].Kp / 16)];
    __syncthreads();
  }

  // The time branch for five stage times of chain g (row 4 s + g of the batch is (slot s, chain g)).
  __device__ __forceinline__ void prepare(const float (&ts)[5]) {
    const NetDev& N = *n;
    const int col = wave * 16 + c;
    {
      const double f = (double)N.fourier[col];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const double te = sign > 0 ? (double)ts[s] : 1.0 - (double)ts[s];    // :229
        double ft = f * te;
        ft -= rint(ft);
        float sv, cv;
        sincospif(2.f * (float)ft, &sv, &cv);                                 // :70-71
        lds[L.ff + (4 * s + g) * R_LDF + col] = cv;
        lds[L.ff + (4 * s + g) * R_LDF + 128 + col] = sv;
      }
    }
    __syncthreads();
    layer_gemm<2, NW, 1>(lds + L.ff, R_LDF, N.Wp + N.L[0].w_off, N.bias + N.L[0].b_off, 16, 8, wave, lane,
                         [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                           for (int i = 0; i < 4; ++i) lds[L.t1 + (16 * m + 4 * g + i) * R_LD + nt * 16 + c] = act_f(acc[i] + b, act());
                         });
    __syncthreads();
    layer_gemm<2, NW, 1>(lds + L.t1, R_LD, N.Wp + N.L[1].w_off, N.bias + N.L[1].b_off, 8, 8, wave, lane,
                         [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                           for (int i = 0; i < 4; ++i) lds[L.st + (16 * m + 4 * g + i) * R_LD + nt * 16 + c] = act_f(acc[i] + b, act());
                         });
    __syncthreads();
    {
      // st half of j1's pre-activation (+ bias) and the wave's k-block of the gate
      const f32x4* P5 = reinterpret_cast<const f32x4*>(N.Wp + N.L[5].w_off) + (wave * 16 + 8) * 64 + lane;
      const f32x4 wg = reinterpret_cast<const f32x4*>(N.Wp + N.L[4].w_off)[wave * 64 + lane];
      f32x4 wf[8];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) wf[kb] = P5[kb * 64];
      const float* arow = lds + L.st + c * R_LD + 4 * g;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ag[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(arow + m * 16 * R_LD + kb * 16);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wf[kb][s], acc[m], 0, 0, 0);
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + m * 16 * R_LD + wave * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) ag[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wg[s], ag[m], 0, 0, 0);
      }
      // the eight (m, i) stores of each kind from ONE base per kind + immediate offsets; the bases are made opaque here, inside the
      // solver loop: as loop invariants the sixteen full addresses were hoisted out of it, spilled, and came back through ten
      // serialized scratch reloads per attempt (each with its own vmcnt(0))
      float* pct = lds + L.ct + (4 * g) * 128 + col;
      float* pgp = lds + L.gp + (wave * 32 + 4 * g) * 2 + c;
      asm volatile("" : "+v"(pct), "+v"(pgp));
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int R = 16 * m + 4 * g + i;
          if (R < 20) {
            pct[(16 * m + i) * 128] = acc[m][i] + b5;
            if (c < 2) pgp[(16 * m + i) * 2] = ag[m][i];
          }
        }
    }
    __syncthreads();
    if (threadIdx.x < 40) {
      const int R = threadIdx.x >> 1, j = threadIdx.x & 1;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += lds[L.gp + (w * 32 + R) * 2 + j];
      lds[L.gate + R * 2 + j] = s + (j ? b4[1] : b4[0]);
    }
    // (visible to everyone after the first barrier of the next evaluation)
  }

  // value + two tangent rows through a resident hidden layer
  __device__ __forceinline__ void hidden(const float* A, const f32x4 (&W)[8], f32x4 acc, float b, float* out) {
    const float* arow = A + c * R_LD + 4 * g;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(arow + kb * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], W[kb][s], acc, 0, 0, 0);
    }
    const float pre = acc[0] + b;
    float* o = out + (4 * g) * R_LD + wave * 16 + c;
    o[0] = act_f(pre, act());
    o[R_LD] = mask_pre(pre, acc[1], act());
    o[2 * R_LD] = mask_pre(pre, acc[2], act());
  }

  __device__ __forceinline__ void eval(float x0, float x1, float tt, float (&kv)[2], float& dl, int phase) {
    const NetDev& N = *n;
    const int slot = phase < 2 ? 0 : (phase - 2 < 4 ? phase - 2 : 4);       // stages 6 and 7 are both at t + dt
    const int col = wave * 16 + c;
    {
      const float pre = fmaf(x1, w1[1], x0 * w1[0]) + b1x;
      float* o = lds + L.x1 + (4 * g) * R_LD + col;
      o[0] = act_f(pre, act());
      o[R_LD] = mask_pre(pre, w1[0], act());
      o[2 * R_LD] = mask_pre(pre, w1[1], act());
    }
    par ^= 1;
    if (wave == NW - 1) {            // the mixture's gradient: 4 chains x 16 modes = the lanes of ONE wave
      float gg[2], hh[2];
      gmm_grad_hdiag2(N.T, x0, x1, c, gg, hh);
      if (c == 0) {
        float* o = lds + L.gc + (par * 4 + g) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool inside = !(N.grad_clip > 0.f) || fabsf(gg[j]) <= N.grad_clip;
          o[j] = clipf(gg[j], N.grad_clip);
          o[2 + j] = inside ? hh[j] : 0.f;
        }
      }
    }
    __syncthreads();
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    hidden(lds + L.x1, W3f, zero, b3, lds + L.cat);
    __syncthreads();
    hidden(lds + L.cat, W5f, f32x4{lds[L.ct + (4 * slot + g) * 128 + col], 0.f, 0.f, 0.f}, 0.f, lds + L.j1);
    __syncthreads();
#ifdef D2_OUT_MFMA
    hidden(lds + L.j1, W6f, zero, b6, lds + L.j2);
    __syncthreads();
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(lds + L.j2 + c * R_LD + 4 * g + wave * 16);
      f32x4 acc = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], W7f[s], acc, 0, 0, 0);
      float* p = lds + L.part + (g * NW + wave) * 4;
      if (c == 0) { p[0] = acc[0]; p[2] = acc[1]; }
      if (c == 1) { p[1] = acc[0]; p[3] = acc[2]; }
    }
    __syncthreads();
#else
    {
      // Round 4: the out layer (hj2 -> 2) rides in the LAST hidden layer's epilogue.  Each lane holds value / e1-tangent / e2-tangent
      // of chain g at ONE column of j2: its share of out_0, out_1, d out_0 / d e1, d out_1 / d e2 is a product with W_out[col][.],
      // the wave's 16 columns are summed over the 16 lanes of the group (DPP row sums), the eight waves through LDS as before.
      // One barrier and one matrix phase less per evaluation (a float reassociation of the out layer's k-sum).
      const float* arow = lds + L.j1 + c * R_LD + 4 * g;
      f32x4 acc = zero;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + kb * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], W6f[kb][s], acc, 0, 0, 0);
      }
      const float pre = acc[0] + b6;
      const float v = act_f(pre, act()), t1v = mask_pre(pre, acc[1], act()), t2v = mask_pre(pre, acc[2], act());
      const float s0 = group16_sum_dpp(v * w7c[0]), s1 = group16_sum_dpp(v * w7c[1]);
      const float s2 = group16_sum_dpp(t1v * w7c[0]), s3 = group16_sum_dpp(t2v * w7c[1]);
      if (c == 0) *reinterpret_cast<f32x4*>(lds + L.part + (g * NW + wave) * 4) = f32x4{s0, s1, s2, s3};
    }
    __syncthreads();
#endif
    float o0 = 0.f, o1 = 0.f, j11 = 0.f, j22 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + L.part + (g * NW + w) * 4);
      o0 += pa[0]; o1 += pa[1]; j11 += pa[2]; j22 += pa[3];
    }
    const f32x4 gch = *reinterpret_cast<const f32x4*>(lds + L.gc + (par * 4 + g) * 4);
    const float g0 = lds[L.gate + (4 * slot + g) * 2], g1 = lds[L.gate + (4 * slot + g) * 2 + 1];
    // v = nn_xt + nn_t * clip(grad log pi(x)) (:88-90);  trace J = sum_j (d nn_xt e_j)_j + nn_t_j 1[|g_j| <= clip] H_jj
    const float v0 = o0 + b7[0] + g0 * gch[0], v1 = o1 + b7[1] + g1 * gch[1];
    const float tr = (j11 + g0 * gch[2]) + (j22 + g1 * gch[3]);
    kv[0] = sign > 0 ? v0 : -v0; kv[1] = sign > 0 ? v1 : -v1;
    dl = sign > 0 ? -tr : tr;                                                 // :218 / :239
  }
};


// (Measured and dropped, round 3: a FOUR-ROW tile -- two chains per workgroup as two v_mfma_f32_4x4x1 passes sharing the resident
// fragments, a quarter of the matrix-pipe time per chain and layer.  Parity-green on the first run, and no faster: in-kernel stamps
// (tools/d2_cycles.py) put the resident tile at 9.7 k cycles per evaluation for 6.4 k of matrix pipe and the four-row tile at 10.0 k
// for 3.6 k -- an evaluation is five barrier-separated phases of ~2 k cycles whose length is a dependent chain (LDS read -> MFMA
// chain -> fold / epilogue -> barrier), not matrix throughput; with twice the workgroups it lost on large launches: 4096-chain
// flow step 43.6 ms against 24.9.)

// Integrate chain g's augmented ODE from t = 0 to 1 (every lane of group g holds the same state).  The state machine of
// ode.hip: ode_solve -- phase 0: f0, phase 1: the extra evaluation of the initial-step heuristic, phases 2..7: the six stages.
template <typename TILE>
__device__ __forceinline__ void solve(TILE& T, float rtol, float atol, int max_attempts, float (&y)[2], float& ell, int& natt,
                                      const Replay& rp, int rp_solve, int rp_row) {
  const float inv_n = 1.f / 3.f;                    // d + 1 components
  float k[7][2], kl[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) { k[j][0] = 0.f; k[j][1] = 0.f; kl[j] = 0.f; }
  float t = 0.f, dt = 0.f, h0 = 0.f, d1 = 0.f, na = 0.f;
  bool done = false;
  ell = 0.f;
  int phase = 0;
#pragma unroll 1
  for (;;) {
    float cf[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) cf[j] = DP_TAB[phase][j];
    const float hs = phase == 1 ? h0 : dt;
    const float ts = t + hs * cf[6];
    float xin[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) acc += cf[j] * k[j][q];
      xin[q] = y[q] + hs * acc;
    }
    float kv[2], dlv;
#ifdef MFM_STAMPS
    const unsigned long long c0_ = __builtin_amdgcn_s_memtime();
#endif
    if (phase <= 2) {          // time branch: of the evaluations of the initial-step heuristic (phases 0, 1), of the five distinct
      float ts5[5];            // stage times of the attempt that starts (phase 2)
#pragma unroll
      for (int q = 0; q < 5; ++q) ts5[q] = phase < 2 ? ts : t + dt * DP_TAB[2 + q][6];
      T.prepare(ts5);
    }
#ifdef MFM_STAMPS
    const unsigned long long c1_ = __builtin_amdgcn_s_memtime();
#endif
    T.eval(xin[0], xin[1], ts, kv, dlv, phase);
#ifdef MFM_STAMPS
    { const unsigned long long c2_ = __builtin_amdgcn_s_memtime(); T.cyc[0] += c1_ - c0_; T.cyc[1] += c2_ - c1_; T.cyc[2] += 1; if (phase <= 2) T.cyc[3] += 1; }      // time batch, evaluation, counts (tools/d2_cycles.py)
#endif
    const int dst = phase == 0 ? 0 : phase - 1 + (phase == 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j == dst) { k[j][0] = kv[0]; k[j][1] = kv[1]; kl[j] = dlv; }

    if (phase == 0) {
      // ---- initial step size, part 1 (Hairer II.4, order 4) ----
      float p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float sc = atol + fabsf(y[q]) * rtol;
        const float a0 = y[q] / sc, a1 = k[0][q] / sc;
        p0 += a0 * a0; p1 += a1 * a1;
      }
      const float a1 = dlv / atol;                                     // ell0 = 0 -> scale = atol
      const float d0 = sqrtf(p0); d1 = sqrtf(p1 + a1 * a1);
      h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
      phase = 1;
    } else if (phase == 1) {
      float p2 = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float sc = atol + fabsf(y[q]) * rtol;
        const float a2 = (k[1][q] - k[0][q]) / sc;
        p2 += a2 * a2;
      }
      const float a2 = (dlv - kl[0]) / atol;
      const float d2 = sqrtf(p2 + a2 * a2) / h0;
      const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
      dt = fminf(100.f * h0, h1);
      if (rp.dt) {
        const size_t o = rp.at(rp_solve, rp_row, 0);
        if (T.writer()) rp.dt_own[o] = dt;
        dt = rp.dt[o];
      }
      phase = 2;
      if (!__syncthreads_or(dt > 0.f ? 1 : 0)) break;
    } else if (phase < 7) {
      phase += 1;
    } else {
      // ---- end of an attempted step: xin holds y1 (row 7 of the table = 5th-order weights) ----
      float e2 = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float er = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) er += DP_E[j] * k[j][q];
        er *= hs;
        const float tol = atol + rtol * fmaxf(fabsf(y[q]), fabsf(xin[q]));
        const float rr = er / tol;
        e2 += rr * rr;
      }
      const float dti = hs;
      const bool active = !done && na < (float)max_attempts && dti > 0.f;
      float sl = 0.f, el = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) sl += DP_TAB[7][j] * kl[j];
#pragma unroll
      for (int j = 0; j < 7; ++j) el += DP_E[j] * kl[j];
      const float l1 = ell + dti * sl;
      el *= dti;
      const float tol = atol + rtol * fmaxf(fabsf(ell), fabsf(l1));
      const float rr = el / tol;
      const float ratio = sqrtf((e2 + rr * rr) * inv_n);
      bool acc = active && ratio <= 1.f;
      const float dfac = ratio < 1.f ? 1.f : 0.2f;
      const float fac = fminf(10.f, fmaxf(0.9f * powf(ratio, -0.2f), dfac));
      float ndt = fmaxf(ratio == 0.f ? dti * 10.f : dti * fac, 0.f);
      if (rp.dt && active) {
        const int j = (int)na;
        const bool in = j < rp.cap, nx = j + 1 < rp.cap;
        const size_t o = rp.at(rp_solve, rp_row, in ? j : 0);
        if (T.writer() && in) { rp.ratio[o] = ratio; if (nx) rp.dt_own[o + 1] = ndt; }
        acc = in && rp.acc[o] != 0;
        ndt = nx ? rp.dt[o + 1] : 0.f;
      }
      if (acc) {
        const float tn = t + dti;
        if (tn >= 1.f) {
          // final output: 4th-order interpolant of this step evaluated at t = 1
          const float sfrac = (1.f - t) / (tn - t);
          float lm = 0.f;
#pragma unroll
          for (int j = 0; j < 7; ++j) lm += DP_M[j] * kl[j];
          const float y0 = ell, y1 = l1, ym = y0 + dti * lm, f0 = dti * kl[0], f1 = dti * kl[6];
          const float pa = -2.f * f0 + 2.f * f1 - 8.f * y0 - 8.f * y1 + 16.f * ym;
          const float pb = 5.f * f0 - 3.f * f1 + 18.f * y0 + 14.f * y1 - 32.f * ym;
          const float pc = -4.f * f0 + f1 - 11.f * y0 - 5.f * y1 + 16.f * ym;
          ell = (((pa * sfrac + pb) * sfrac + pc) * sfrac + f0) * sfrac + y0;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float km = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) km += DP_M[j] * k[j][q];
            const float x0 = y[q], x1 = xin[q], xm = x0 + dti * km, g0 = dti * k[0][q], g1 = dti * k[6][q];
            const float qa = -2.f * g0 + 2.f * g1 - 8.f * x0 - 8.f * x1 + 16.f * xm;
            const float qb = 5.f * g0 - 3.f * g1 + 18.f * x0 + 14.f * x1 - 32.f * xm;
            const float qc = -4.f * g0 + g1 - 11.f * x0 - 5.f * x1 + 16.f * xm;
            y[q] = (((qa * sfrac + qb) * sfrac + qc) * sfrac + g0) * sfrac + x0;
          }
          done = true;
        } else {
          ell = l1;
#pragma unroll
          for (int q = 0; q < 2; ++q) { y[q] = xin[q]; k[0][q] = k[6][q]; }
          kl[0] = kl[6];
        }
        t = tn;
      }
      if (active) { dt = ndt; na += 1.f; }
      const bool more = !done && na < (float)max_attempts && dt > 0.f;
      // the loop condition must be uniform over the workgroup: a lane only knows its own chain
      if (!__syncthreads_or(more ? 1 : 0)) break;
      phase = 2;
    }
  }
  natt = (int)na;
}

template <typename TILE>
__global__ __launch_bounds__(NW * 64) void transform_kernel(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  TILE T;
  T.init(&a.net, lds);
  T.sign = a.direction;
  const int b = T.chain();
  float y[2] = {a.in[(size_t)b * 2], a.in[(size_t)b * 2 + 1]}, ell; int natt;
  solve(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, 0, b);
  if (T.writer()) {
    a.out[(size_t)b * 2] = y[0]; a.out[(size_t)b * 2 + 1] = y[1];
    a.ldj[b] = ell;
    if (a.nsteps) a.nsteps[b] = natt;
  }
}

// One flow-based MH step per chain (random-walk in latent space :264-278, or independent :246-260).
template <typename TILE>
__global__ __launch_bounds__(NW * 64) void flow_kernel(OdeArgs a, FlowArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef MFM_STAMPS
  const unsigned long long fc0_ = __builtin_amdgcn_s_memtime();
#endif
  TILE T;
  T.init(&a.net, lds);
  const NetDev& N = a.net;
  const int b = T.chain();
  const Key2 kb = split_at(f.key, f.n_total, f.chain_offset + (uint32_t)b);                         // :303
  const double lp_old = f.logp[b];
  float y[2] = {f.pos[(size_t)b * 2], f.pos[(size_t)b * 2 + 1]}, ell, vol0, lq_ref = 0.f;
  int natt, natt_tot;
  natt_tot = 0; vol0 = 0.f;
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {         // ONE call site of the solver: inverse solve (:267 / :251), then the proposal's forward solve
    if (ph == 1) {
      // ---- proposal in latent space ----
      vol0 = ell;
      const float scale = 2.38f / sqrtf(2.f);                                                       // :262
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float nz = a.zgen[(size_t)b * 2 + q];
        if (f.mode == MFM_FLOW_RWMH) y[q] = y[q] + scale * nz;                                      // :268
        else { const float up = f.ref_std * nz; r0 += y[q] * y[q]; y[q] = up; r1 += up * up; }      // :249
      }
      if (f.mode == MFM_FLOW_IMH) lq_ref = -0.5f * (r0 - r1) / (f.ref_std * f.ref_std);             // :254-255
    }
    T.sign = ph == 0 ? -1 : 1;
    solve(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, ph, b);
    natt_tot += natt;
  }
  // ---- target at the proposal (:270 / :252), tempered: beta * loglik + logprior (the mixture has no prior term) ----
  double lp; float gg[2];
  if (N.T.n_modes <= 16) gmm_eval_lanes16<2>(N.T, y, T.c, &lp, gg);
  else gmm_eval<2>(N.T, y, &lp, gg);
  const double lpn = f.beta * lp;
  // ---- accept / reject (:271-278 / :253-260); the acceptance probability is NOT clipped ----
  const double la = lpn - (double)ell - lp_old - (double)vol0 + (double)lq_ref;
  const double ap = exp(la);
  const double u = uniform01(split_at(kb, 4, 1), 0, 1);
  const bool acc = u <= ap;                       // NaN compares false -> reject
  if (T.writer()) {
    if (a.rp.diag) { double* o = a.rp.diag + 4 * (size_t)b; o[0] = vol0; o[1] = ell; o[2] = lpn; o[3] = la; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const size_t o = (size_t)b * 2 + q;
      if (f.proposed) f.proposed[o] = y[q];
      if (acc) { f.pos[o] = y[q]; f.grad[o] = (float)f.beta * gg[q]; }
    }
    if (acc) f.logp[b] = lpn;
    if (f.acc_prob) f.acc_prob[b] = (float)ap;
    if (f.accepted) f.accepted[b] = acc ? 1 : 0;
    if (f.nsteps) f.nsteps[b] = natt_tot;
  }
#ifdef MFM_STAMPS
  if (g_flow_dbg && threadIdx.x == 0) {
    unsigned long long* o = g_flow_dbg + blockIdx.x * 64;
    o[0] = __builtin_amdgcn_s_memtime() - fc0_; o[2] = T.cyc[2]; o[3] = T.cyc[1]; o[5] = T.cyc[3]; o[6] = T.cyc[0];
  }
#endif
}

// Which tiling serves a launch of `rows` samples.  The resident-weight 4-chain tile wherever its shape fits (measured against the
// 16-chain tile: flow step of 512 chains 14.3 -> 2.75 ms, of 4096 chains 35.6 -> 24.1 ms, transform of 409,600 draws 568 -> 400 ms);
// the streamed 4-chain tile for other widths while the 16-chain tiling would leave CUs without a workgroup (it pays 4 x the
// weight traffic per chain: 868 ms on the 409,600 draws).  MFM_D2_TILE (development / tests): 16 forces the generic tile, 4 the
// 4-chain tiles, 4s the streamed instance of them even where the resident one fits.
static int pick(const NetDev& n, int hutch, int rows) {       // 0: generic 16-chain tile, 1: streamed 4-chain, 2: resident 4-chain
  if (!shape_ok(n, hutch)) return 0;
  const int e = g_sw.d2_tile;
  if (e == 16) return 0;
  if (e == 5) return 1;
  if (shape_ok_r(n, hutch)) return 2;
  return (e || rows / 16 < 256) ? 1 : 0;
}
static bool use_for(const NetDev& n, int hutch, int rows) { return pick(n, hutch, rows) != 0; }
template <typename TILE>
static int launch_transform_t(const OdeArgs& a, size_t sm, hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)transform_kernel<TILE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(transform_kernel<TILE>, dim3(a.n / TILE::CHAINS), dim3(NW * 64), sm, stream, a);
  return 0;
}
template <typename TILE>
static int launch_flow_t(const OdeArgs& a, const FlowArgs& f, size_t sm, hipStream_t stream) {
  (void)hipFuncSetAttribute((const void*)flow_kernel<TILE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(flow_kernel<TILE>, dim3(a.n / TILE::CHAINS), dim3(NW * 64), sm, stream, a, f);
  return 0;
}
static int launch_transform(const OdeArgs& a, hipStream_t stream) {
  if (pick(a.net, a.hutch, a.n) == 2) {
    const size_t sm = (size_t)layout_r().total * sizeof(float);
    return a.net.act == MFM_ACT_RELU ? launch_transform_t<TileR<MFM_ACT_RELU>>(a, sm, stream) : launch_transform_t<TileR<-1>>(a, sm, stream);
  }

  return launch_transform_t<Tile>(a, (size_t)layout(a.net).total * sizeof(float), stream);
}
static int launch_flow(const OdeArgs& a, const FlowArgs& f, hipStream_t stream) {
  if (pick(a.net, a.hutch, a.n) == 2) {
    const size_t sm = (size_t)layout_r().total * sizeof(float);
    return a.net.act == MFM_ACT_RELU ? launch_flow_t<TileR<MFM_ACT_RELU>>(a, f, sm, stream) : launch_flow_t<TileR<-1>>(a, f, sm, stream);
  }

  return launch_flow_t<Tile>(a, f, (size_t)layout(a.net).total * sizeof(float), stream);
}

}  // namespace d2
