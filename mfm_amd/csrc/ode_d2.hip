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
  if (n.d != 2 || hutch || n.T.kind != MFM_TARGET_GMM) return false;
  if (n.F % 16 || n.F2p != 2 * n.F) return false;
  if (n.hx1 / 16 > 2 * NW) return false;
  return (size_t)layout(n).total * sizeof(float) <= 160 * 1024;
}

struct Tile {
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

  // One evaluation of the augmented field for chain g at x = (x0, x1), time tt.  kv = dx/dt, dl = d(logdet)/dt.
  __device__ __forceinline__ void eval(float x0, float x1, float tt, float (&kv)[2], float& dl, bool reuse_time) {
    const NetDev& N = *n;
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

// Integrate chain g's augmented ODE from t = 0 to 1 (every lane of group g holds the same state).  The state machine of
// ode.hip: ode_solve -- phase 0: f0, phase 1: the extra evaluation of the initial-step heuristic, phases 2..7: the six stages.
__device__ __forceinline__ void solve(Tile& T, float rtol, float atol, int max_attempts, float (&y)[2], float& ell, int& natt,
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
    T.eval(xin[0], xin[1], ts, kv, dlv, phase == 7);
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
        if (T.wave == 0 && T.c == 0) rp.dt_own[o] = dt;
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
        if (T.wave == 0 && T.c == 0 && in) { rp.ratio[o] = ratio; if (nx) rp.dt_own[o + 1] = ndt; }
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

__global__ __launch_bounds__(NW * 64) void transform_kernel(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  Tile T;
  T.init(&a.net, lds);
  T.sign = a.direction;
  const int b = blockIdx.x * 4 + T.g;
  float y[2] = {a.in[(size_t)b * 2], a.in[(size_t)b * 2 + 1]}, ell; int natt;
  solve(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, 0, b);
  if (T.wave == 0 && T.c == 0) {
    a.out[(size_t)b * 2] = y[0]; a.out[(size_t)b * 2 + 1] = y[1];
    a.ldj[b] = ell;
    if (a.nsteps) a.nsteps[b] = natt;
  }
}

// One flow-based MH step per chain (random-walk in latent space :264-278, or independent :246-260).
__global__ __launch_bounds__(NW * 64) void flow_kernel(OdeArgs a, FlowArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  Tile T;
  T.init(&a.net, lds);
  const NetDev& N = a.net;
  const int b = blockIdx.x * 4 + T.g;
  const Key2 kb = split_at(f.key, f.n_total, f.chain_offset + (uint32_t)b);                         // :303
  const double lp_old = f.logp[b];
  float y[2] = {f.pos[(size_t)b * 2], f.pos[(size_t)b * 2 + 1]}, ell, vol0, lq_ref = 0.f;
  int natt, natt_tot;
  // ---- inverse solve from the current position (:267 / :251), key_hutch2 unused (exact trace) ----
  T.sign = -1;
  solve(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, 0, b);
  vol0 = ell; natt_tot = natt;
  // ---- proposal in latent space ----
  {
    const float scale = 2.38f / sqrtf(2.f);                                                         // :262
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float nz = a.zgen[(size_t)b * 2 + q];
      if (f.mode == MFM_FLOW_RWMH) y[q] = y[q] + scale * nz;                                        // :268
      else { const float up = f.ref_std * nz; r0 += y[q] * y[q]; y[q] = up; r1 += up * up; }        // :249
    }
    if (f.mode == MFM_FLOW_IMH) lq_ref = -0.5f * (r0 - r1) / (f.ref_std * f.ref_std);               // :254-255
  }
  T.sign = 1;
  solve(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, 1, b);
  natt_tot += natt;
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
  if (T.wave == 0 && T.c == 0) {
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
}

// 4-chain tiles while the 16-chain tiling would leave CUs without a workgroup (or under MFM_D2_TILE=4); MFM_D2_TILE=16 keeps
// the generic tile.
static bool use_for(const NetDev& n, int hutch, int rows) {
  if (!shape_ok(n, hutch)) return false;
  if (const char* e = getenv("MFM_D2_TILE")) return atoi(e) == 4;
  return rows / 16 < 256;
}
static int launch_transform(const OdeArgs& a, hipStream_t stream) {
  const size_t sm = (size_t)layout(a.net).total * sizeof(float);
  (void)hipFuncSetAttribute((const void*)transform_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(transform_kernel, dim3(a.n / 4), dim3(NW * 64), sm, stream, a);
  return 0;
}
static int launch_flow(const OdeArgs& a, const FlowArgs& f, hipStream_t stream) {
  const size_t sm = (size_t)layout(a.net).total * sizeof(float);
  (void)hipFuncSetAttribute((const void*)flow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(flow_kernel, dim3(a.n / 4), dim3(NW * 64), sm, stream, a, f);
  return 0;
}

}  // namespace d2
