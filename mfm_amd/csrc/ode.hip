// K5 + K6: continuous-normalising-flow transforms with log-det (adaptive Dormand-Prince 5(4), Hutchinson or exact
// trace) and the flow-based Metropolis-Hastings steps, one persistent workgroup per tile of 16 chains.
//
// Replaces exe_flow_matching.py:206-221 (transform_and_logdet), :223-242 (inverse_and_logdet), :246-260 / :264-278
// (independent / random-walk MH in latent space) and the third-party integrator they call,
// jax.experimental.ode.odeint (:13,345-349; restated in oracle/ode.py): same tableau, same
// controller (RMS error ratio over d+1 components, accept <= 1, factor clip(0.9 r^-1/5, dfactor, 10)), same initial
// step heuristic, output at t = 1 by the 4th-order interpolant of the last accepted step.  Each chain runs its OWN
// adaptive sequence (its own t, dt); the 16 chains of a tile execute in masked lock-step like a vmapped while_loop.
//
// Per RHS evaluation the tile pushes value AND tangent rows through the MLP together (forward-mode, M = 32 rows for
// the x-path layers), so every streamed weight fragment is used twice.  z is drawn once per solve,
// so the first x-layer's tangent pre-activation z W_x1 is computed once per solve and only re-masked afterwards.
// The Runge-Kutta stages k1..k7 never leave registers: they are kept in the MFMA accumulator layout of the output
// layer, which is also the layout the next stage's input is assembled in.
//
// The intermediate output times of the "4-mode" example (n_ts = 5) do not change the step sequence (steps are not
// clamped to output times); they only reset odeint's attempted-step counter, so mxstep is applied per segment as
// mxstep * (n_ts - 1) attempted steps in total.
#include "mlp.hip.h"
#include "prng.hip.h"

// Gaussian draws the solver kernels consume (Hutchinson probes of both solves, latent random-walk noise) are
// produced by a small separate kernel into this workspace: the float64 erfinv code would otherwise sit inside the
// persistent solver kernel and cost it ~200 spilled registers.
// padded images for the shape-specialised kernels serving a NARROWER lattice than their tile width (ode_fast.hip: dispatch):
// the parameter pack in the tile's shapes and six [rows][D] images (position, gradient, proposal / output, three probes)
struct PadWs { float* Wp; float* WpT; float* bias; float* img[6]; size_t rows; int D; };
struct OdeWs { float* noise; size_t rows; f32x4* fast_scr; size_t fast_wgs; PadWs pad; };
static thread_local const PadWs* t_pad = nullptr;      // the padded images of the context whose OdeArgs were formed last (ode_args): host-side only,
                                                        // kept out of the kernel arguments (two more words there cost the headline kernel 2 %)
#define ODE_FAST_MAX_WGS 1024                 // grid cap of the shape-specialised transform kernel (it loops over tiles)
#define ODE_FAST_SCR_F4 (5 * 8 * 3 * 64)      // float4 of time-branch scratch per workgroup (ode_fast.hip)
// Chains per workgroup of the flow step.  A tile of 16 chains per workgroup is the cheapest per chain (233 k cycles per attempted step
// for 16 rows), but a launch with fewer tiles than CUs leaves the rest of the chip idle while every tile walks its ~330 attempts in
// lock-step: the same chains spread over more workgroups, 8 / 4 / 2 per tile with the other rows marked done, run in the tile's
// compact (198 k per attempt), two-pass (152 k) or one-pass (92 k) layout from the second attempt on.  The reference's phi-four default
// is 1024 chains (multi_modal.py:52-55): 64 tiles -> 256 workgroups of 4.  A row's arithmetic depends on the layout its tile runs in
// (float reassociations), so the choice is a function of the chain count only.  MFM_FLOW_LIVE forces it (tests, A/B).
static int flow_live_rows(int n_chains) {
  if (g_sw.flow_live == 16 || g_sw.flow_live == 8 || g_sw.flow_live == 4 || g_sw.flow_live == 2) return g_sw.flow_live;
  int live = 16;
  while (live > 2 && n_chains / (live / 2) <= 256) live /= 2;
  return live;
}

static int ode_ws_alloc(const NetDev& n, const mfm_config& c, OdeWs& w) {
  w.rows = (size_t)(c.max_eval_samples > c.n_chain_local ? c.max_eval_samples : c.n_chain_local);
  if (hipMalloc((void**)&w.noise, 3 * w.rows * n.d * sizeof(float)) != hipSuccess) return -4;
  const size_t t_all = w.rows / 16, t_chain = (size_t)c.n_chain_local / 16;
  w.fast_wgs = t_all < ODE_FAST_MAX_WGS ? t_all : ODE_FAST_MAX_WGS;
  const size_t t_flow = (size_t)c.n_chain_local / flow_live_rows(c.n_chain_local);      // the flow step runs one workgroup per 16 (8 / 4 / 2) chains
  if (w.fast_wgs < t_flow) w.fast_wgs = t_flow;
  memset(&w.pad, 0, sizeof w.pad);
  if (c.hutch && n.nT == 2 && n.nX == 2 && n.nJ == 2 && !net_ragged(n) && n.T.kind == MFM_TARGET_PHI4 && n.act == MFM_ACT_RELU && n.F == 128 && n.ht1 == 128 && n.ht2 == 128 && n.hx1 == 128 &&
      n.hx2 == 128 && n.hj1 == 128 && n.hj2 == 128 && n.d >= 16 && n.d < 256 && n.d != 128 && n.d % 16 == 0) {
    const int D = n.d < 128 ? 128 : 256;
    const size_t wtot = (size_t)2 * 128 * 128 + 5 * 128 * 128 + 3 * (size_t)D * 128;      // FS<D>::WTOT
    w.pad.D = D; w.pad.rows = w.rows;
    if (hipMalloc((void**)&w.pad.Wp, wtot * sizeof(float)) != hipSuccess || hipMalloc((void**)&w.pad.WpT, wtot * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&w.pad.bias, (6 * 128 + 2 * D) * sizeof(float)) != hipSuccess) return -4;
    if (hipMemset(w.pad.WpT, 0, wtot * sizeof(float)) != hipSuccess) return -4;
    for (int i = 0; i < 6; ++i)
      if (hipMalloc((void**)&w.pad.img[i], w.rows * D * sizeof(float)) != hipSuccess) return -4;
  }
  return hipMalloc((void**)&w.fast_scr, w.fast_wgs * ODE_FAST_SCR_F4 * sizeof(f32x4)) == hipSuccess ? 0 : -4;
}
static void ode_ws_free(OdeWs& w) {
  if (w.noise) (void)hipFree(w.noise);
  if (w.fast_scr) (void)hipFree(w.fast_scr);
  for (void* p_ : {(void*)w.pad.Wp, (void*)w.pad.WpT, (void*)w.pad.bias, (void*)w.pad.img[0], (void*)w.pad.img[1], (void*)w.pad.img[2],
                   (void*)w.pad.img[3], (void*)w.pad.img[4], (void*)w.pad.img[5]})
    if (p_) (void)hipFree(p_);
  memset(&w.pad, 0, sizeof w.pad);
  w.noise = nullptr; w.fast_scr = nullptr;
}

// PARITY INSTRUMENTATION (mfm_debug_replay): a prescribed Dormand-Prince step sequence per sample.  With dt != nullptr the
// solver takes dt[j] as the step size of attempt j (j = 0: the initial step) and acc[j] as its accept decision instead of
// its controller's; the controller is still evaluated and what it computed is recorded (ratio[j]: error ratio of attempt j,
// dt_own[0]: its initial step, dt_own[j + 1]: the step it proposed after attempt j).  Element (solve s, sample r, attempt j)
// of every array sits at ((s * n + r) * cap + j); a flow step has two solves (0: inverse, 1: forward), a transform one.
struct Replay {
  const float* dt; const uint8_t* acc; float* ratio; float* dt_own; int cap, n;
  double* diag;             // flow step only, may be null: [n][4] = {vol0 (inverse log-det), log-det of the forward solve,
                            // tempered log-density at the proposal, log acceptance ratio} -- the terms of :271-274 / :253-256
  __device__ __forceinline__ size_t at(int solve, int row, int j) const { return ((size_t)solve * n + row) * cap + j; }
};

struct OdeArgs {
  NetDev net;
  int hutch;
  float rtol, atol;
  int max_attempts;
  int direction, per_chain_keys;
  const uint32_t* keys; Key2 key;
  const float* in; float* out; float* ldj; int* nsteps;
  int n;
  const float* z1;          // probe of the (first / only) solve, [n][d]
  const float* z2;          // probe of the second solve (flow step)
  const float* zgen;        // latent proposal noise (flow step)
  f32x4* fast_scr;          // time-branch scratch of the shape-specialised kernels (ode_fast.hip)
  Replay rp;                // rp.dt == nullptr: off (production)
  int fixed_method, fixed_steps;      // fixed_steps > 0: the fixed-step mode (ode_fixed.hip; mfm_config.ode_method / ode_steps)
};

struct FlowArgs {
  int mode; Key2 key; uint32_t n_total, chain_offset;
  double beta;
  float ref_std;            // reference distribution IndepGaussian(dim, var): independent-MH proposals and their density (:249-255)
  float* pos; double* logp; float* grad;
  float* acc_prob; uint8_t* accepted; float* proposed; int* nsteps;
};

static OdeArgs ode_args(const NetDev& n, const mfm_config& c, const OdeWs& w) {
  OdeArgs a; memset(&a, 0, sizeof a);
  a.z1 = w.noise; a.z2 = w.noise + w.rows * n.d; a.zgen = w.noise + 2 * w.rows * n.d; a.fast_scr = w.fast_scr; t_pad = &w.pad;
  a.net = n; a.hutch = c.hutch; a.rtol = (float)c.rtol; a.atol = (float)c.atol;
  a.max_attempts = c.mxstep * (c.n_ts > 1 ? c.n_ts - 1 : 1);
  a.fixed_method = c.ode_method; a.fixed_steps = c.ode_method != MFM_ODE_DOPRI5 ? c.ode_steps : 0;
  return a;
}

struct OdeLds {   // float offsets
  int ff_j1, ldff, ldj1, x, ldx, z, cat, ldcat, x1, ldx1, j2_t1, ldj2, ldt1, red, gcs, rs, gc, hz, kz, ldgc, total;
};
__host__ __device__ inline OdeLds ode_lds_layout(const NetDev& n, int NW) {
  OdeLds L; int o = 0;
  auto take = [&](int cnt) { int r = o; o += cnt; return r; };
  // rows of the value + tangent images: 32 (values + one tangent), 48 for d = 2, whose exact trace pushes the tangents of
  // BOTH basis vectors through in one pass (OdeTile::eval_x2)
  const int RT = n.d == 2 ? 48 : 32;
  L.ldff = n.F2p + 8; L.ldj1 = n.hj1 + 8;
  { int a = 16 * L.ldff, b = RT * L.ldj1; L.ff_j1 = take(a > b ? a : b); }
  L.ldx = n.dp + 8; L.x = take(16 * L.ldx); L.z = take(16 * L.ldx);
  L.ldcat = n.hx2 + n.ht2 + 8; L.cat = take(RT * L.ldcat);
  L.ldx1 = n.hx1 + 8; L.x1 = take(RT * L.ldx1);
  L.ldj2 = n.hj2 + 8; L.ldt1 = n.ht1 + 8;
  { int a = RT * L.ldj2, b = 16 * L.ldt1; L.j2_t1 = take(a > b ? a : b); }
  L.red = take(8 * 16 * NW);      // 8 reduction slots of [NW][16 rows]
  L.gcs = take(16 * 24);     // small-d targets: grad[8], hvp[8], inside-mask[8] per row
  L.rs = take(16 * 16);
  L.ldgc = n.dp + 8; L.gc = L.hz = L.kz = 0;
  if (n.T.kind == MFM_TARGET_LGCP) { L.gc = take(16 * L.ldgc); L.hz = take(16 * L.ldgc); L.kz = take(16 * L.ldgc); }   // grad, masked H z, K^-1 z      // per-row solver state (t, dt, h0, d1, ell, kl[7], natt, done): 16 arrays of 16 rows
  L.total = o;
  return L;
}

// Dormand-Prince tableau (oracle/ode.py)
__device__ static const float DP_E[7] = {(float)(35.0 / 384 - 1951.0 / 21600), 0.f, (float)(500.0 / 1113 - 22642.0 / 50085),
                                         (float)(125.0 / 192 - 451.0 / 720), (float)(-2187.0 / 6784 + 12231.0 / 42400),
                                         (float)(11.0 / 84 - 649.0 / 6300), (float)(-1.0 / 60)};
__device__ static const float DP_M[7] = {(float)(6025192743.0 / 30085553152.0 / 2), 0.f, (float)(51252292925.0 / 65400821598.0 / 2),
                                         (float)(-2691868925.0 / 45128329728.0 / 2), (float)(187940372067.0 / 1594534317056.0 / 2),
                                         (float)(-1776094331.0 / 19743644256.0 / 2), (float)(11237099.0 / 235043384.0 / 2)};

#ifdef MFM_STAMPS
__device__ unsigned long long* g_flow_dbg = nullptr;     // [WG][8]: cycles, realtime ticks, field evaluations, ...
#define MFM_STAMP(id) do { if (stamps && lane == 0) stamps[(blockIdx.x * NW + wave) * 16 + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFM_STAMP(id) do {} while (0)
#endif

template <int TPW, int NW>
struct OdeTile {
  unsigned long long* stamps = nullptr;
  unsigned long long n_eval = 0, cyc_eval = 0;
  const NetDev* n;
  OdeLds L;
  float* lds;
  int lane, wave, g, c;
  bool hutch;               // tangent rows are pushed through the MLP (always on in the solver kernels)
  bool exact;               // exact trace: d basis probes per RHS evaluation instead of one Hutchinson probe
  int sign;                 // +1 forward (:208-218), -1 inverse (:225-239)
  float tz1[2][4];          // z W_x1 for this lane's x1-layer tiles (<= 2 tiles per wave)
  float tze[2][2];          // exact trace, d = 2: e_j W_x1 = row j of W_x1 at this lane's column of its x1-layer tiles
  float gate[TPW][4];       // nn_t of the last evaluated stage time (kept for stages that share it)

  __device__ __forceinline__ float* bFF() { return lds + L.ff_j1; }
  __device__ __forceinline__ float* bJ1() { return lds + L.ff_j1; }
  __device__ __forceinline__ float* bX() { return lds + L.x; }
  __device__ __forceinline__ float* bZ() { return lds + L.z; }
  __device__ __forceinline__ float* bCat() { return lds + L.cat; }
  __device__ __forceinline__ float* bX1() { return lds + L.x1; }
  __device__ __forceinline__ float* bJ2() { return lds + L.j2_t1; }
  __device__ __forceinline__ float* bT1() { return lds + L.j2_t1; }
  __device__ __forceinline__ float* red(int slot) { return lds + L.red + slot * 16 * NW; }
  __device__ __forceinline__ float* gcs() { return lds + L.gcs; }
  // per-row solver state: every lane needs the values of its 4 rows (4g..4g+3) -> one ds_read_b128 per field;
  // they are written by ONE lane per row group (wave 0, c == 0) and become visible at the next barrier
  __device__ __forceinline__ f32x4 rs_get(int field) { return *reinterpret_cast<const f32x4*>(lds + L.rs + field * 16 + 4 * g); }
  __device__ __forceinline__ void rs_put(int field, const float (&v)[4]) {
    if (wave == 0 && c == 0) *reinterpret_cast<f32x4*>(lds + L.rs + field * 16 + 4 * g) = f32x4{v[0], v[1], v[2], v[3]};
  }

  // sum over the tile's columns of per-lane partials (rows 4g..4g+3); uses reduction slot `slot`.
  // The caller guarantees a barrier between two uses of the same slot.
  __device__ __forceinline__ void row_reduce(float (&p)[4], int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = group16_sum(p[i]);
    float* r = red(slot);
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[wave * 16 + 4 * g + i] = p[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += r[w * 16 + row];
      p[i] = t;
    }
  }

  // z W_x1 (no bias), once per solve.  Requires bZ filled and a barrier before.  LGCP: also K^-1 z (the dense part of
  // the Hessian-vector product), read back by the SAME lanes in eval.
  __device__ __forceinline__ void precompute_tz1() {
    if (n->T.kind == MFM_TARGET_LGCP)
      layer_gemm<1, NW, 1>(bZ() + 4, L.ldx, n->T.KinvP, nullptr, n->dp / 16, n->dp / 16, wave, lane,
                           [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                             for (int i = 0; i < 4; ++i) lds[L.kz + (4 * g + i) * L.ldgc + nt * 16 + c] = acc[i];
                           });
    const LayerDesc& l2 = n->L[2];
    layer_gemm<1, NW, 1>(bZ() + 4, L.ldx, n->Wp + l2.w_off, nullptr, l2.Kp / 16, l2.Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int i = 0; i < 4; ++i) { if (q == 0) tz1[0][i] = acc[i]; else tz1[1][i] = acc[i]; }
                     });
  }

  // One evaluation of the augmented field at the positions in bX (value rows) and times tt[i] (this lane's rows).
  // Outputs: kv[q][i] = dx/dt for (row 4g+i, col 16(wave+4q)+c); dl[i] = d(logdet)/dt for row 4g+i (all lanes agree).
  template <bool WANT_JZ = false>
  __device__ __forceinline__ void eval(const float (&tt)[4], float (&kv)[TPW][4], float (&dl)[4], int red_slot,
                                       float (*jzo)[4] = nullptr, bool reuse_time = false) {
    const NetDev& N = *n;
    const int d = N.d;
    MFM_STAMP(0);
    // Fourier features (:70-71); skipped when this stage shares its time with the previous one (st and the gate
    // output are still valid: Dopri5 stages 6 and 7 are both at t + dt)
    if (!reuse_time) {
      if (N.F % 16 == 0) {        // one sincos per (row, frequency) feeds the cos and the sin block
        for (int nt = wave; nt < N.F / 16; nt += NW) {
          const int col = nt * 16 + c;
          const double f = (double)N.fourier[col];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double te = sign > 0 ? (double)tt[i] : 1.0 - (double)tt[i];          // :229
            double ft = f * te;
            ft -= rint(ft);
            float sv, cv;
            sincospif(2.f * (float)ft, &sv, &cv);
            bFF()[(4 * g + i) * L.ldff + col] = cv;
            bFF()[(4 * g + i) * L.ldff + N.F + col] = sv;
          }
        }
      } else {
        for (int nt = wave; nt * 16 < N.F2p; nt += NW) {
          const int col = nt * 16 + c;
          const bool is_sin = col >= N.F;
          const double f = col < 2 * N.F ? (double)N.fourier[is_sin ? col - N.F : col] : 0.0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = 0.f;
            if (col < 2 * N.F) {
              const double te = sign > 0 ? (double)tt[i] : 1.0 - (double)tt[i];          // :229
              double ft = f * te;
              ft -= rint(ft);
              float sv, cv;
              sincospif(2.f * (float)ft, &sv, &cv);
              v = is_sin ? sv : cv;
            }
            bFF()[(4 * g + i) * L.ldff + col] = v;
          }
        }
      }
    }
    if (N.T.kind == MFM_TARGET_GMM && (N.T.n_modes <= 16 ? threadIdx.x < 256 : threadIdx.x < 16)) {
      // small-d target: grad / hvp per row -- one mode per lane, 16 lanes per row (targets.hip.h); > 16 modes: one thread per row
      double lp; float gg[8], hv[8];
      const int row = N.T.n_modes <= 16 ? (int)(threadIdx.x >> 4) : (int)threadIdx.x;
      const float* xr = bX() + row * L.ldx + 4;
      const float* zr = bZ() + row * L.ldx + 4;
      if (N.T.n_modes <= 16) gmm_eval_lanes16<8>(N.T, xr, threadIdx.x & 15, &lp, gg, hutch ? zr : nullptr, hv);
      else gmm_eval<8>(N.T, xr, &lp, gg, hutch ? zr : nullptr, hv);
      if (N.T.n_modes > 16 || (threadIdx.x & 15) == 0) {
        float* o = gcs() + row * 24;
        for (int j = 0; j < d; ++j) {
          const bool inside = !(N.grad_clip > 0.f) || fabsf(gg[j]) <= N.grad_clip;
          o[j] = clipf(gg[j], N.grad_clip);
          o[8 + j] = (hutch && inside) ? hv[j] : 0.f;
        }
      }
    }
    if (N.T.kind == MFM_TARGET_LGCP)      // grad log pi(x) = c - a exp(x) - K^-1 (x - mu);  H z = -a exp(x) z - K^-1 z
      layer_gemm<1, NW, 1>(bX() + 4, L.ldx, N.T.KinvP, N.T.kbias, N.dp / 16, N.dp / 16, wave, lane,
                           [&](int q, int nt, int m, f32x4 acc, float kb) {
                             const int col = nt * 16 + c;
#pragma unroll
                             for (int i = 0; i < 4; ++i) {
                               const int row = 4 * g + i;
                               float gcv = 0.f, hzv = 0.f;
                               if (col < d) {
                                 const float ex = N.T.poisson_a * expf(bX()[row * L.ldx + 4 + col]);
                                 const float graw = N.T.counts[col] - ex - (acc[i] + kb);
                                 gcv = clipf(graw, N.grad_clip);
                                 const bool inside = !(N.grad_clip > 0.f) || fabsf(graw) <= N.grad_clip;
                                 if (hutch && inside) hzv = -ex * bZ()[row * L.ldx + 4 + col] - lds[L.kz + row * L.ldgc + col];
                               }
                               lds[L.gc + row * L.ldgc + col] = gcv;
                               lds[L.hz + row * L.ldgc + col] = hzv;
                             }
                           });
    MFM_STAMP(2);
    __syncthreads();
    MFM_STAMP(3);
    // t1 ; x1 (value rows; tangent rows = relu' * (z W_x1))
    if (!reuse_time)
    layer_gemm<1, NW, 1>(bFF(), L.ldff, N.Wp + N.L[0].w_off, N.bias + N.L[0].b_off, N.L[0].Kp / 16, N.L[0].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int i = 0; i < 4; ++i) bT1()[(4 * g + i) * L.ldt1 + nt * 16 + c] = act_f(acc[i] + b, N.act);
                     });
    layer_gemm<1, NW, 1>(bX() + 4, L.ldx, N.Wp + N.L[2].w_off, N.bias + N.L[2].b_off, N.L[2].Kp / 16, N.L[2].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int i = 0; i < 4; ++i) {
                         const float pre = acc[i] + b;
                         const int o = (4 * g + i) * L.ldx1 + nt * 16 + c;
                         bX1()[o] = act_f(pre, N.act);
                         const float tz = (q == 0) ? tz1[0][i] : tz1[1][i];
                         bX1()[16 * L.ldx1 + o] = mask_pre(pre, tz, N.act);
                       }
                     });
    MFM_STAMP(4);
    __syncthreads();
    MFM_STAMP(5);
    // t2 -> st (value rows only) ; x2 on value + tangent rows
    if (!reuse_time)
    layer_gemm<1, NW, 1>(bT1(), L.ldt1, N.Wp + N.L[1].w_off, N.bias + N.L[1].b_off, N.L[1].Kp / 16, N.L[1].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int i = 0; i < 4; ++i) bCat()[(4 * g + i) * L.ldcat + N.hx2 + nt * 16 + c] = act_f(acc[i] + b, N.act);
                     });
    {
      f32x4 keep = {0, 0, 0, 0};   // value pre-activation of the same tile, handed from m = 0 to m = 1
      layer_gemm<2, NW, 1>(bX1(), L.ldx1, N.Wp + N.L[3].w_off, N.bias + N.L[3].b_off, N.L[3].Kp / 16, N.L[3].Np / 16, wave, lane,
                       [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                         for (int i = 0; i < 4; ++i) {
                           const int o = (4 * g + i) * L.ldcat + nt * 16 + c;
                           if (m == 0) { keep[i] = acc[i] + b; bCat()[o] = act_f(keep[i], N.act); }
                           else bCat()[16 * L.ldcat + o] = mask_pre(keep[i], acc[i], N.act);
                         }
                       });
    }
    MFM_STAMP(6);
    __syncthreads();
    MFM_STAMP(7);
    // gate (registers) ; j1
    if (!reuse_time)
    layer_gemm<1, NW, 1>(bCat() + N.hx2, L.ldcat, N.Wp + N.L[4].w_off, N.bias + N.L[4].b_off, N.L[4].Kp / 16, N.L[4].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int qq = 0; qq < TPW; ++qq)
                         if (qq == q) {
#pragma unroll
                           for (int i = 0; i < 4; ++i) gate[qq][i] = acc[i] + b;
                         }
                     });
    {
      f32x4 keep = {0, 0, 0, 0};
      layer_gemm<2, NW, 1>(bCat(), L.ldcat, N.Wp + N.L[5].w_off, N.bias + N.L[5].b_off, N.L[5].Kp / 16, N.L[5].Np / 16, wave, lane,
                       [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                         for (int i = 0; i < 4; ++i) {
                           const int o = (4 * g + i) * L.ldj1 + nt * 16 + c;
                           if (m == 0) { keep[i] = acc[i] + b; bJ1()[o] = act_f(keep[i], N.act); }
                           else bJ1()[16 * L.ldj1 + o] = mask_pre(keep[i], acc[i], N.act);
                         }
                       });
    }
    MFM_STAMP(8);
    __syncthreads();
    MFM_STAMP(9);
    {
      f32x4 keep = {0, 0, 0, 0};
      layer_gemm<2, NW, 1>(bJ1(), L.ldj1, N.Wp + N.L[6].w_off, N.bias + N.L[6].b_off, N.L[6].Kp / 16, N.L[6].Np / 16, wave, lane,
                       [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                         for (int i = 0; i < 4; ++i) {
                           const int o = (4 * g + i) * L.ldj2 + nt * 16 + c;
                           if (m == 0) { keep[i] = acc[i] + b; bJ2()[o] = act_f(keep[i], N.act); }
                           else bJ2()[16 * L.ldj2 + o] = mask_pre(keep[i], acc[i], N.act);
                         }
                       });
    }
    MFM_STAMP(10);
    __syncthreads();
    MFM_STAMP(11);
    // out: v = nn_xt + nn_t * clip(grad log pi(x)) (:88-90);  J z = d nn_xt . z + nn_t * 1[|g| <= clip] * (H z)
    float dpart[4] = {0.f, 0.f, 0.f, 0.f};
    {
      f32x4 hzk = {0, 0, 0, 0};      // masked Hessian-vector product of the tile, handed from m = 0 to m = 1
      layer_gemm<2, NW, 1>(bJ2(), L.ldj2, N.Wp + N.L[7].w_off, N.bias + N.L[7].b_off, N.L[7].Kp / 16, N.L[7].Np / 16, wave, lane,
                       [&](int q, int nt, int m, f32x4 acc, float b) {
                         const int col = nt * 16 + c;
#pragma unroll
                         for (int i = 0; i < 4; ++i) {
                           const int row = 4 * g + i;
                           float gt = 0.f;
#pragma unroll
                           for (int qq = 0; qq < TPW; ++qq) gt = (qq == q) ? gate[qq][i] : gt;
                           if (m == 0) {
                             float gc = 0.f, hz = 0.f;
                             if (col < d) {
                               if (N.T.kind == MFM_TARGET_PHI4) {
                                 const float* xr = bX() + row * L.ldx + 4;
                                 const float graw = phi4_grad(N.T, xr, col);
                                 gc = clipf(graw, N.grad_clip);
                                 const bool inside = !(N.grad_clip > 0.f) || fabsf(graw) <= N.grad_clip;
                                 if (hutch && inside) hz = phi4_hvp(N.T, xr, bZ() + row * L.ldx + 4, col);
                               } else if (N.T.kind == MFM_TARGET_LGCP) {
                                 gc = lds[L.gc + row * L.ldgc + col];
                                 hz = lds[L.hz + row * L.ldgc + col];
                               } else {
                                 gc = gcs()[row * 24 + col];
                                 hz = gcs()[row * 24 + 8 + col];
                               }
                             }
                             hzk[i] = hz;
                             const float v = col < d ? acc[i] + b + gt * gc : 0.f;
#pragma unroll
                             for (int qq = 0; qq < TPW; ++qq)
                               if (qq == q) kv[qq][i] = sign > 0 ? v : -v;
                           } else if (col < d) {
                             const float jz = acc[i] + gt * hzk[i];
                             dpart[i] += bZ()[row * L.ldx + 4 + col] * jz;
                             if (WANT_JZ) {
#pragma unroll
                               for (int qq = 0; qq < TPW; ++qq)
                                 if (qq == q) jzo[qq][i] = jz;
                             }
                           }
                         }
                       });
    }
    MFM_STAMP(12);
    row_reduce(dpart, red_slot);
    MFM_STAMP(13);
#pragma unroll
    for (int i = 0; i < 4; ++i) dl[i] = sign > 0 ? -dpart[i] : dpart[i];     // :218 / :239
  }

  // ---- exact trace for d = 2 (the mixture examples, no --hutch: trace(jacfwd(v)), :216-217 / :236-237) in ONE pass ----------
  // The generic exact mode runs the whole evaluation once per basis vector (value rows recomputed, 12 barriers, the target
  // evaluated twice).  With two basis vectors the tangent rows of BOTH ride behind the value rows: M = 48 through x2 / j1 / j2 /
  // out (every weight fragment feeds three row tiles), the x1 tangent is row j of W_x1 itself (no probe image), the
  // Hessian-vector products of e_1 and e_2 come from two calls of the mode-per-lane mixture evaluation, and
  // trace J = (J e_1)_1 + (J e_2)_2 is picked out of the out layer's tangent tiles.  Same arithmetic per entry as `eval`.
  __device__ __forceinline__ void precompute_tze() {
    const LayerDesc& l2 = n->L[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int nt = wave + NW * q;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        tze[q][j] = nt < l2.Np / 16 ? n->Wp[l2.w_off + pack_index(j, nt * 16 + c, l2.Kp / 16)] : 0.f;
    }
  }
  __device__ __forceinline__ void eval_x2(const float (&tt)[4], float (&kv)[TPW][4], float (&dl)[4], int red_slot, bool reuse_time) {
    const NetDev& N = *n;
    const int d = N.d;
    if (!reuse_time) {
      for (int nt = wave; nt * 16 < N.F2p; nt += NW) {
        const int col = nt * 16 + c;
        const bool is_sin = col >= N.F;
        const double f = col < 2 * N.F ? (double)N.fourier[is_sin ? col - N.F : col] : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = 0.f;
          if (col < 2 * N.F) {
            const double te = sign > 0 ? (double)tt[i] : 1.0 - (double)tt[i];          // :229
            double ft = f * te;
            ft -= rint(ft);
            float sv, cv;
            sincospif(2.f * (float)ft, &sv, &cv);
            v = is_sin ? sv : cv;
          }
          bFF()[(4 * g + i) * L.ldff + col] = v;
        }
      }
    }
    if (N.T.n_modes <= 16 ? threadIdx.x < 256 : threadIdx.x < 16) {          // grad log pi, H e_1, H e_2 per row
      const bool l16 = N.T.n_modes <= 16;
      const int row = l16 ? (int)(threadIdx.x >> 4) : (int)threadIdx.x;
      const float* xr = bX() + row * L.ldx + 4;
      const float e1[8] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, e2[8] = {0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      double lp; float gg[8], h1[8], h2[8];
      if (l16) { gmm_eval_lanes16<8>(N.T, xr, threadIdx.x & 15, &lp, gg, e1, h1); gmm_eval_lanes16<8>(N.T, xr, threadIdx.x & 15, &lp, gg, e2, h2); }
      else { gmm_eval<8>(N.T, xr, &lp, gg, e1, h1); gmm_eval<8>(N.T, xr, &lp, gg, e2, h2); }
      if (!l16 || (threadIdx.x & 15) == 0) {
        float* o = gcs() + row * 24;
        for (int j = 0; j < d; ++j) {
          const bool inside = !(N.grad_clip > 0.f) || fabsf(gg[j]) <= N.grad_clip;
          o[j] = clipf(gg[j], N.grad_clip);
          o[8 + j] = inside ? h1[j] : 0.f;
          o[16 + j] = inside ? h2[j] : 0.f;
        }
      }
    }
    __syncthreads();
    if (!reuse_time)
    layer_gemm<1, NW, 1>(bFF(), L.ldff, N.Wp + N.L[0].w_off, N.bias + N.L[0].b_off, N.L[0].Kp / 16, N.L[0].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int i = 0; i < 4; ++i) bT1()[(4 * g + i) * L.ldt1 + nt * 16 + c] = act_f(acc[i] + b, N.act);
                     });
    layer_gemm<1, NW, 1>(bX() + 4, L.ldx, N.Wp + N.L[2].w_off, N.bias + N.L[2].b_off, N.L[2].Kp / 16, N.L[2].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int i = 0; i < 4; ++i) {
                         const float pre = acc[i] + b;
                         const int o = (4 * g + i) * L.ldx1 + nt * 16 + c;
                         bX1()[o] = act_f(pre, N.act);
                         bX1()[16 * L.ldx1 + o] = mask_pre(pre, q == 0 ? tze[0][0] : tze[1][0], N.act);
                         bX1()[32 * L.ldx1 + o] = mask_pre(pre, q == 0 ? tze[0][1] : tze[1][1], N.act);
                       }
                     });
    __syncthreads();
    if (!reuse_time)
    layer_gemm<1, NW, 1>(bT1(), L.ldt1, N.Wp + N.L[1].w_off, N.bias + N.L[1].b_off, N.L[1].Kp / 16, N.L[1].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int i = 0; i < 4; ++i) bCat()[(4 * g + i) * L.ldcat + N.hx2 + nt * 16 + c] = act_f(acc[i] + b, N.act);
                     });
    // value + two tangent row tiles through one layer: m = 0 keeps the value's pre-activation for the masks of m = 1, 2
    auto three = [&](const float* A, int lda, int layer, float* out, int ldo) {
      f32x4 keep = {0, 0, 0, 0};
      layer_gemm<3, NW, 1>(A, lda, N.Wp + N.L[layer].w_off, N.bias + N.L[layer].b_off, N.L[layer].Kp / 16, N.L[layer].Np / 16, wave, lane,
                       [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                         for (int i = 0; i < 4; ++i) {
                           const int o = (4 * g + i) * ldo + nt * 16 + c;
                           if (m == 0) { keep[i] = acc[i] + b; out[o] = act_f(keep[i], N.act); }
                           else out[16 * m * ldo + o] = mask_pre(keep[i], acc[i], N.act);
                         }
                       });
    };
    three(bX1(), L.ldx1, 3, bCat(), L.ldcat);
    __syncthreads();
    if (!reuse_time)
    layer_gemm<1, NW, 1>(bCat() + N.hx2, L.ldcat, N.Wp + N.L[4].w_off, N.bias + N.L[4].b_off, N.L[4].Kp / 16, N.L[4].Np / 16, wave, lane,
                     [&](int q, int nt, int m, f32x4 acc, float b) {
#pragma unroll
                       for (int qq = 0; qq < TPW; ++qq)
                         if (qq == q) {
#pragma unroll
                           for (int i = 0; i < 4; ++i) gate[qq][i] = acc[i] + b;
                         }
                     });
    three(bCat(), L.ldcat, 5, bJ1(), L.ldj1);
    __syncthreads();
    three(bJ1(), L.ldj1, 6, bJ2(), L.ldj2);
    __syncthreads();
    float dpart[4] = {0.f, 0.f, 0.f, 0.f};
    {
      f32x4 hz1 = {0, 0, 0, 0}, hz2 = {0, 0, 0, 0};
      layer_gemm<3, NW, 1>(bJ2(), L.ldj2, N.Wp + N.L[7].w_off, N.bias + N.L[7].b_off, N.L[7].Kp / 16, N.L[7].Np / 16, wave, lane,
                       [&](int q, int nt, int m, f32x4 acc, float b) {
                         const int col = nt * 16 + c;
#pragma unroll
                         for (int i = 0; i < 4; ++i) {
                           const int row = 4 * g + i;
                           float gt = 0.f;
#pragma unroll
                           for (int qq = 0; qq < TPW; ++qq) gt = (qq == q) ? gate[qq][i] : gt;
                           if (m == 0) {
                             float gc = 0.f;
                             if (col < d) { gc = gcs()[row * 24 + col]; hz1[i] = gcs()[row * 24 + 8 + col]; hz2[i] = gcs()[row * 24 + 16 + col]; }
                             const float v = col < d ? acc[i] + b + gt * gc : 0.f;
#pragma unroll
                             for (int qq = 0; qq < TPW; ++qq)
                               if (qq == q) kv[qq][i] = sign > 0 ? v : -v;
                           } else if (col == m - 1) {                    // (J e_m)_m: the diagonal entry of this basis vector
                             dpart[i] += acc[i] + gt * (m == 1 ? hz1[i] : hz2[i]);
                           }
                         }
                       });
    }
    row_reduce(dpart, red_slot);
#pragma unroll
    for (int i = 0; i < 4; ++i) dl[i] = sign > 0 ? -dpart[i] : dpart[i];     // :218 / :239
  }
};

// Integrate the augmented ODE from t = 0 to 1 for the tile whose start positions are in y[][] (accumulator layout);
// on return y holds the interpolated positions at t = 1, ell the log-det, natt the attempted steps per row.
// Requires: bZ filled (probe, zero where col >= d), pads of bX / bZ zero, tangent rows of bCat[:, hx2:] zero.
//
// ONE call site of the field evaluation, driven by a small state machine (phase 0: f0, phase 1: the extra
// evaluation of the initial-step heuristic, phases 2..7: the six Runge-Kutta stages), so the kernel carries one copy
// of the MLP code; stage results are routed into k[.] with predicated moves (static register indices).
__device__ static const float DP_TAB[8][7] = {   // [phase][j]: input = y + h * sum_j TAB[phase][j] k_j ; last column: time fraction
    {0, 0, 0, 0, 0, 0, 0.f},
    {1, 0, 0, 0, 0, 0, 1.f},
    {1.f / 5, 0, 0, 0, 0, 0, 1.f / 5},
    {3.f / 40, 9.f / 40, 0, 0, 0, 0, 3.f / 10},
    {44.f / 45, -56.f / 15, 32.f / 9, 0, 0, 0, 4.f / 5},
    {19372.f / 6561, -25360.f / 2187, 64448.f / 6561, -212.f / 729, 0, 0, 8.f / 9},
    {9017.f / 3168, -355.f / 33, 46732.f / 5247, 49.f / 176, -5103.f / 18656, 0, 1.f},
    {35.f / 384, 0, 500.f / 1113, 125.f / 192, -2187.f / 6784, 11.f / 84, 1.f}};

enum { RS_T = 0, RS_DT = 1, RS_H0 = 2, RS_D1 = 3, RS_ELL = 4, RS_KL = 5 /* ..11 */, RS_NATT = 12, RS_DONE = 13, RS_FLAG = 14, RS_SFRAC = 15 };

template <int TPW, int NW>
__device__ __forceinline__ void ode_solve(OdeTile<TPW, NW>& T, float rtol, float atol, int max_attempts,
                                          float (&y)[TPW][4], float (&ell)[4], int (&natt)[4], const Replay& rp, int rp_solve, int rp_row0) {
  const NetDev& N = *T.n;
  const int d = N.d, g = T.g, c = T.c, wave = T.wave;
  const float inv_n = 1.f / (float)(d + 1);
  auto colmask = [&](int q) { return (wave + NW * q) * 16 + c < d; };

  // The seven stage derivatives of x stay in registers (accumulator layout).  Everything that is per ROW (time,
  // step, log-det and its stage derivatives, counters) lives in the LDS row-state block: all lanes would compute
  // identical copies, and at two waves per SIMD the registers are needed for the MFMA pipeline instead.
  float k[7][TPW][4];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) k[j][q][i] = 0.f;
  {
    const float z4[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                       // previous users of the row-state block / bZ writers are done
#pragma unroll
    for (int fld = 0; fld < 14; ++fld) T.rs_put(fld, z4);
  }
  __syncthreads();
  const bool ex2 = T.exact && N.d == 2 && N.T.kind == MFM_TARGET_GMM;      // exact trace of the d = 2 mixtures: one fused pass
  if (ex2) T.precompute_tze();
  else if (!T.exact) T.precompute_tz1();     // Hutchinson probe: z W_x1 once per solve (reads bZ only)

  int phase = 0;
#pragma unroll 1
  for (;;) {
    // ---- stage input: y + h * sum_j TAB[phase][j] k_j, written to bX ----
    float xin[TPW][4], ts[4], hs[4];
    float cf[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) cf[j] = DP_TAB[phase][j];
    {
      const f32x4 t4 = T.rs_get(RS_T), h4 = T.rs_get(phase == 1 ? RS_H0 : RS_DT);
#pragma unroll
      for (int i = 0; i < 4; ++i) { hs[i] = h4[i]; ts[i] = t4[i] + hs[i] * cf[6]; }
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = (wave + NW * q) * 16 + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += cf[j] * k[j][q][i];
        xin[q][i] = y[q][i] + hs[i] * acc;
      }
      if (col < N.dp) {
#pragma unroll
        for (int i = 0; i < 4; ++i) T.bX()[(4 * g + i) * T.L.ldx + 4 + col] = col < d ? xin[q][i] : 0.f;
      }
    }
    __syncthreads();
    float kv[TPW][4], dlv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) kv[q][i] = 0.f;      // waves that own no output tile never write kv
    // Hutchinson: one probe z (drawn once per solve).  Exact trace (:216-217 / :236-237): trace J = sum_j e_j . J e_j,
    // one tangent pass per basis vector; passes after the first reuse the time branch of the first.
    const int nprobe = (T.exact && !ex2) ? d : 1;
#pragma unroll 1
    for (int pj = 0; pj < nprobe; ++pj) {
      if (T.exact && !ex2) {
        for (int idx = threadIdx.x; idx < 16 * N.dp; idx += NW * 64) {
          const int row = idx / N.dp, col = idx - row * N.dp;
          T.bZ()[row * T.L.ldx + 4 + col] = col == pj ? 1.f : 0.f;
        }
        __syncthreads();
        T.precompute_tz1();
      }
      float dl1[4];
#ifdef MFM_STAMPS
      const unsigned long long c0_ = __builtin_amdgcn_s_memtime();
#endif
      if (ex2) T.eval_x2(ts, kv, dl1, phase & 1, phase == 7);
      else T.eval(ts, kv, dl1, (phase + pj) & 1, nullptr, phase == 7 || pj > 0);
#ifdef MFM_STAMPS
      T.cyc_eval += __builtin_amdgcn_s_memtime() - c0_; T.n_eval += 1;
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) dlv[i] += dl1[i];
    }
    // ---- route the result: phase 0 -> k[0], phase 1 -> k[1], phase p >= 2 -> k[p - 1] ----
    const int dst = phase == 0 ? 0 : phase - 1 + (phase == 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j == dst) {
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) k[j][q][i] = kv[q][i];
      }
    T.rs_put(RS_KL + dst, dlv);            // read again only after later barriers (phases 1 and 7)

    if (phase == 0) {
      // ---- initial step size, part 1 (Hairer II.4, order 4) ----
      float p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < TPW; ++q)
        if (colmask(q)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float sc = atol + fabsf(y[q][i]) * rtol;
            const float a0 = y[q][i] / sc, a1 = k[0][q][i] / sc;
            p0[i] += a0 * a0; p1[i] += a1 * a1;
          }
        }
      T.row_reduce(p0, 2); T.row_reduce(p1, 3);
      float h0[4], d1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a1 = dlv[i] / atol;                                // ell0 = 0 -> scale = atol
        const float d0 = sqrtf(p0[i]); d1[i] = sqrtf(p1[i] + a1 * a1);
        h0[i] = (d0 < 1e-5f || d1[i] < 1e-5f) ? 1e-6f : 0.01f * d0 / d1[i];
      }
      T.rs_put(RS_H0, h0); T.rs_put(RS_D1, d1);
      __syncthreads();
      phase = 1;
    } else if (phase == 1) {
      float p2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < TPW; ++q)
        if (colmask(q)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float sc = atol + fabsf(y[q][i]) * rtol;
            const float a2 = (k[1][q][i] - k[0][q][i]) / sc;
            p2[i] += a2 * a2;
          }
        }
      T.row_reduce(p2, 2);
      const f32x4 h04 = T.rs_get(RS_H0), d14 = T.rs_get(RS_D1), kl0 = T.rs_get(RS_KL + 0);
      float dt[4];
      bool any = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a2 = (dlv[i] - kl0[i]) / atol;
        const float d2 = sqrtf(p2[i] + a2 * a2) / h04[i];
        const float h1 = (d14[i] <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h04[i] * 1e-3f)
                                                            : powf(0.01f / fmaxf(d14[i], d2), 0.2f);
        dt[i] = fminf(100.f * h04[i], h1);
        if (rp.dt) {
          const size_t o = rp.at(rp_solve, rp_row0 + 4 * g + i, 0);
          if (wave == 0 && c == 0) rp.dt_own[o] = dt[i];
          dt[i] = rp.dt[o];
        }
        any |= (dt[i] > 0.f);
      }
      T.rs_put(RS_DT, dt);
      phase = 2;
      if (!__syncthreads_or(any ? 1 : 0)) break;
    } else if (phase < 7) {
      phase += 1;
    } else {
      // ---- end of an attempted step: xin holds y1 (row 7 of the table = 5th-order weights) ----
      const f32x4 t4 = T.rs_get(RS_T), ell4 = T.rs_get(RS_ELL);
      const f32x4 na4 = T.rs_get(RS_NATT), dn4 = T.rs_get(RS_DONE);
      f32x4 kl[7];
#pragma unroll
      for (int j = 0; j < 6; ++j) kl[j] = T.rs_get(RS_KL + j);
      kl[6] = f32x4{dlv[0], dlv[1], dlv[2], dlv[3]};
      float e2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < TPW; ++q)
        if (colmask(q)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float er = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) er += DP_E[j] * k[j][q][i];
            er *= hs[i];
            const float tol = atol + rtol * fmaxf(fabsf(y[q][i]), fabsf(xin[q][i]));
            const float rr = er / tol;
            e2[i] += rr * rr;
          }
        }
      T.row_reduce(e2, 2);
      bool any = false;
      float t_n[4], dt_n[4], ell_n[4], kl0_n[4], na_n[4], dn_n[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dti = hs[i];
        const bool was_done = dn4[i] != 0.f;
        const bool active = !was_done && na4[i] < (float)max_attempts && dti > 0.f;
        float sl = 0.f, el = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) sl += DP_TAB[7][j] * kl[j][i];
#pragma unroll
        for (int j = 0; j < 7; ++j) el += DP_E[j] * kl[j][i];
        const float l1 = ell4[i] + dti * sl;
        el *= dti;
        const float tol = atol + rtol * fmaxf(fabsf(ell4[i]), fabsf(l1));
        const float rr = el / tol;
        const float ratio = sqrtf((e2[i] + rr * rr) * inv_n);
        bool acc = active && ratio <= 1.f;
        const float dfac = ratio < 1.f ? 1.f : 0.2f;
        const float fac = fminf(10.f, fmaxf(0.9f * powf(ratio, -0.2f), dfac));
        float ndt = fmaxf(ratio == 0.f ? dti * 10.f : dti * fac, 0.f);
        if (rp.dt && active) {
          const int j = (int)na4[i];
          const bool in = j < rp.cap, nx = j + 1 < rp.cap;
          const size_t o = rp.at(rp_solve, rp_row0 + 4 * g + i, in ? j : 0);
          if (wave == 0 && c == 0 && in) { rp.ratio[o] = ratio; if (nx) rp.dt_own[o + 1] = ndt; }
          acc = in && rp.acc[o] != 0;
          ndt = nx ? rp.dt[o + 1] : 0.f;
        }
        t_n[i] = t4[i]; ell_n[i] = ell4[i]; kl0_n[i] = kl[0][i]; dn_n[i] = dn4[i];
        if (acc) {
          const float tn = t4[i] + dti;
          if (tn >= 1.f) {
            // final output: 4th-order interpolant of this step evaluated at t = 1
            const float sfrac = (1.f - t4[i]) / (tn - t4[i]);
            float lm = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) lm += DP_M[j] * kl[j][i];
            const float y0 = ell4[i], y1 = l1, ym = y0 + dti * lm, f0 = dti * kl[0][i], f1 = dti * kl[6][i];
            const float pa = -2.f * f0 + 2.f * f1 - 8.f * y0 - 8.f * y1 + 16.f * ym;
            const float pb = 5.f * f0 - 3.f * f1 + 18.f * y0 + 14.f * y1 - 32.f * ym;
            const float pc = -4.f * f0 + f1 - 11.f * y0 - 5.f * y1 + 16.f * ym;
            ell_n[i] = (((pa * sfrac + pb) * sfrac + pc) * sfrac + f0) * sfrac + y0;
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
              float km = 0.f;
#pragma unroll
              for (int j = 0; j < 7; ++j) km += DP_M[j] * k[j][q][i];
              const float x0 = y[q][i], x1 = xin[q][i], xm = x0 + dti * km, g0 = dti * k[0][q][i], g1 = dti * k[6][q][i];
              const float qa = -2.f * g0 + 2.f * g1 - 8.f * x0 - 8.f * x1 + 16.f * xm;
              const float qb = 5.f * g0 - 3.f * g1 + 18.f * x0 + 14.f * x1 - 32.f * xm;
              const float qc = -4.f * g0 + g1 - 11.f * x0 - 5.f * x1 + 16.f * xm;
              y[q][i] = (((qa * sfrac + qb) * sfrac + qc) * sfrac + g0) * sfrac + x0;
            }
            dn_n[i] = 1.f;
          } else {
            ell_n[i] = l1;
#pragma unroll
            for (int q = 0; q < TPW; ++q) { y[q][i] = xin[q][i]; k[0][q][i] = k[6][q][i]; }
            kl0_n[i] = kl[6][i];
          }
          t_n[i] = tn;
        }
        dt_n[i] = active ? ndt : dti;
        na_n[i] = active ? na4[i] + 1.f : na4[i];
        any |= (dn_n[i] == 0.f && na_n[i] < (float)max_attempts && dt_n[i] > 0.f);
      }
      T.rs_put(RS_T, t_n); T.rs_put(RS_DT, dt_n); T.rs_put(RS_ELL, ell_n); T.rs_put(RS_KL + 0, kl0_n);
      T.rs_put(RS_NATT, na_n); T.rs_put(RS_DONE, dn_n);
      // the loop condition must be uniform over the workgroup: every lane only sees its own 4 rows
      if (!__syncthreads_or(any ? 1 : 0)) break;
      phase = 2;
    }
  }
  {
    const f32x4 ell4 = T.rs_get(RS_ELL), na4 = T.rs_get(RS_NATT);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ell[i] = ell4[i]; natt[i] = (int)na4[i]; }
  }
}

// Gaussian draws for the solver kernels.  mode 0: per-sample keys (uint32[n][2]); 1: ONE shared key (:455);
// 2: flow step -- chain b uses split(split(key, n_total)[chain_offset + b], 4)[sub] (:265 / :247).
__global__ __launch_bounds__(256) void probe_kernel(int mode, const uint32_t* keys, Key2 key, uint32_t n_total, uint32_t chain_offset,
                                                   int sub, int n, int d, float* out) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n) return;
  Key2 k = key;
  if (mode == 0) k = Key2{keys[2 * b], keys[2 * b + 1]};
  else if (mode == 2) k = split_at(split_at(key, n_total, chain_offset + (uint32_t)b), 4, (uint32_t)sub);
  for (int j = lane; j < d; j += 64) out[(size_t)b * d + j] = (float)normal64(k, (uint32_t)j, (uint32_t)d);   // :212 / :232 / :268
}
static void launch_probe(int mode, const uint32_t* keys, Key2 key, uint32_t n_total, uint32_t chain_offset, int sub, int n, int d,
                         float* out, hipStream_t stream) {
  hipLaunchKernelGGL(probe_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, mode, keys, key, n_total, chain_offset, sub, n, d, out);
}

// fill bZ with the Hutchinson probe of each row (normal(key, (d,)), :212 / :232), drawn by probe_kernel, or zero it
template <int TPW, int NW>
__device__ __forceinline__ void fill_probe(OdeTile<TPW, NW>& T, const float* z, int b0, bool hutch) {
  const NetDev& N = *T.n;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (T.wave + NW * q) * 16 + T.c;
    if (col < N.dp) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        T.bZ()[(4 * T.g + i) * T.L.ldx + 4 + col] = (hutch && col < N.d) ? z[(size_t)(b0 + 4 * T.g + i) * N.d + col] : 0.f;
    }
  }
}

template <int TPW, int NW>
__device__ __forceinline__ void tile_init(OdeTile<TPW, NW>& T, const NetDev* n, float* lds, bool hutch) {
  T.n = n; T.L = ode_lds_layout(*n, NW); T.lds = lds;
  T.lane = threadIdx.x & 63; T.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); T.g = T.lane >> 4; T.c = T.lane & 15;
  T.hutch = hutch; T.exact = false; T.sign = 1;
  for (int i = threadIdx.x; i < T.L.total; i += (NW * 64)) lds[i] = 0.f;     // pads, tangent rows of st, scratch
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T.tz1[0][i] = 0.f; T.tz1[1][i] = 0.f;
#pragma unroll
    for (int q = 0; q < TPW; ++q) T.gate[q][i] = 0.f;
  }
  __syncthreads();
}

template <int TPW, int NW>
__global__ __launch_bounds__(NW * 64) void ode_transform_kernel(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  OdeTile<TPW, NW> T;
  tile_init(T, &a.net, lds, true);
  T.exact = a.hutch == 0;
  T.sign = a.direction;
  const int b0 = blockIdx.x * 16, d = a.net.d;
  fill_probe(T, a.z1, b0, !T.exact);
  float y[TPW][4], ell[4]; int natt[4];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (T.wave + NW * q) * 16 + T.c;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[q][i] = col < d ? a.in[(size_t)(b0 + 4 * T.g + i) * d + col] : 0.f;
  }
  ode_solve<TPW, NW>(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, 0, b0);
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (T.wave + NW * q) * 16 + T.c;
    if (col < d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a.out[(size_t)(b0 + 4 * T.g + i) * d + col] = y[q][i];
    }
  }
  if (T.wave == 0 && T.c == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a.ldj[b0 + 4 * T.g + i] = ell[i];
      if (a.nsteps) a.nsteps[b0 + 4 * T.g + i] = natt[i];
    }
  }
}

// v(x, t) and J z for n samples (mfm_vf_apply): one field evaluation per tile.
template <int TPW, int NW>
__global__ __launch_bounds__(NW * 64) void vf_apply_kernel(NetDev net, const float* x, const float* t, const float* tan, int n,
                                                               float* v, float* jvp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  OdeTile<TPW, NW> T;
  tile_init(T, &net, lds, tan != nullptr);
  const int b0 = blockIdx.x * 16, d = net.d;
  float tt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) tt[i] = t[b0 + 4 * T.g + i];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (T.wave + NW * q) * 16 + T.c;
    if (col < d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t o = (size_t)(b0 + 4 * T.g + i) * d + col;
        T.bX()[(4 * T.g + i) * T.L.ldx + 4 + col] = x[o];
        if (tan) T.bZ()[(4 * T.g + i) * T.L.ldx + 4 + col] = tan[o];
      }
    }
  }
  __syncthreads();
  if (tan) T.precompute_tz1();
  float kv[TPW][4], dl[4], jz[TPW][4];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) jz[q][i] = 0.f;
  T.template eval<true>(tt, kv, dl, 0, jz);
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (T.wave + NW * q) * 16 + T.c;
    if (col < d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t o = (size_t)(b0 + 4 * T.g + i) * d + col;
        v[o] = kv[q][i];
        if (jvp) jvp[o] = jz[q][i];
      }
    }
  }
}

#ifdef MFM_STAMPS
template <int TPW, int NW>
__global__ __launch_bounds__(NW * 64) void eval_stamps_kernel(NetDev net, const float* x, const float* t, const float* tan, int reps,
                                                              unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  OdeTile<TPW, NW> T;
  tile_init(T, &net, lds, true);
  const int b0 = blockIdx.x * 16, d = net.d;
  float tt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) tt[i] = t[b0 + 4 * T.g + i];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (T.wave + NW * q) * 16 + T.c;
    if (col < d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t o = (size_t)(b0 + 4 * T.g + i) * d + col;
        T.bX()[(4 * T.g + i) * T.L.ldx + 4 + col] = x[o];
        T.bZ()[(4 * T.g + i) * T.L.ldx + 4 + col] = tan[o];
      }
    }
  }
  __syncthreads();
  T.precompute_tz1();
  float kv[TPW][4], dl[4];
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    if (r == reps - 1) T.stamps = stamps;
#pragma unroll
    for (int i = 0; i < 4; ++i) tt[i] += 1e-3f;
    __syncthreads();
    T.eval(tt, kv, dl, r & 1);
  }
  if (T.lane == 0) stamps[(blockIdx.x * NW + T.wave) * 16 + 15] = (unsigned long long)(kv[0][0] + dl[0]);   // keep results live
}
int launch_eval_stamps(const NetDev& n, const float* x, const float* t, const float* tan, int cnt, int reps, unsigned long long* stamps, hipStream_t stream);
#endif

// One flow-based MH step per chain (random-walk in latent space :264-278, or independent :246-260).
template <int TPW, int NW>
__global__ __launch_bounds__(NW * 64) void flow_step_kernel(OdeArgs a, FlowArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  OdeTile<TPW, NW> T;
#ifdef MFM_STAMPS
  const unsigned long long fc0_ = __builtin_amdgcn_s_memtime(), fr0_ = __builtin_amdgcn_s_memrealtime();
#endif
  tile_init(T, &a.net, lds, true);
  T.exact = a.hutch == 0;
  const NetDev& N = a.net;
  const int b0 = blockIdx.x * 16, d = N.d, g = T.g, c = T.c, wave = T.wave;
  Key2 kb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    kb[i] = split_at(f.key, f.n_total, f.chain_offset + (uint32_t)(b0 + 4 * g + i));             // :303
  // sub-keys of :265 / :247 are re-derived where they are used: key_gen 0, key_acc 1, key_hutch1 2, key_hutch2 3
  float y[TPW][4], ell[4], vol0[4] = {0, 0, 0, 0}, lq_ref[4] = {0, 0, 0, 0};
  int natt[4], natt_tot[4] = {0, 0, 0, 0};
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {       // (ONE call site of the solver: inverse solve, then forward solve of the proposal
    if (ph == 0) {
      // ---- inverse solve from the current position (:267 / :251) ----
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int col = (wave + NW * q) * 16 + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[q][i] = col < d ? f.pos[(size_t)(b0 + 4 * g + i) * d + col] : 0.f;
      }
    } else {
      // ---- proposal in latent space ----
      float r0[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0};
      const float scale = 2.38f / sqrtf((float)d);                                                  // :262
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int col = (wave + NW * q) * 16 + c;
        if (col < d) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float nz = a.zgen[(size_t)(b0 + 4 * g + i) * d + col];
            if (f.mode == MFM_FLOW_RWMH) y[q][i] = y[q][i] + scale * nz;                            // :268
            else { const float up = f.ref_std * nz; r0[i] += y[q][i] * y[q][i]; y[q][i] = up; r1[i] += up * up; }   // :249
          }
        }
      }
      if (f.mode == MFM_FLOW_IMH) {     // ref.logprob(u0) - ref.logprob(up) = -(|u0|^2 - |up|^2) / 2   (:254-255)
        __syncthreads();
        T.row_reduce(r0, 5); T.row_reduce(r1, 6);
#pragma unroll
        for (int i = 0; i < 4; ++i) lq_ref[i] = -0.5f * (r0[i] - r1[i]) / (f.ref_std * f.ref_std);
      }
      __syncthreads();
    }
    fill_probe(T, ph == 0 ? a.z1 : a.z2, b0, !T.exact);       // key_hutch2 for the inverse, key_hutch1 for the forward solve
    T.sign = ph == 0 ? -1 : 1;
    ode_solve<TPW, NW>(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, ph, b0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ph == 0) vol0[i] = ell[i];
      natt_tot[i] += natt[i];
    }
  }
  // ---- target at the proposal (:270 / :252), tempered: beta * loglik + logprior ----
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (wave + NW * q) * 16 + c;
    if (col < N.dp) {
#pragma unroll
      for (int i = 0; i < 4; ++i) T.bX()[(4 * g + i) * T.L.ldx + 4 + col] = col < d ? y[q][i] : 0.f;
    }
  }
  __syncthreads();
  double lpn[4];
  float gnew[TPW][4];
  if (N.T.kind == MFM_TARGET_PHI4) {
    double part[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = (wave + NW * q) * 16 + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gnew[q][i] = 0.f;
        if (col < d) {
          const float* xr = T.bX() + (4 * g + i) * T.L.ldx + 4;
          part[i] += phi4_term(N.T, xr, col);
          gnew[q][i] = (float)f.beta * phi4_grad(N.T, xr, col);
        }
      }
    }
    double* rd = reinterpret_cast<double*>(T.red(0));      // slots 0..1 as [NW][16 rows] doubles
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) part[i] += __shfl_xor(part[i], o, 64);
      if (c == 0) rd[wave * 16 + 4 * g + i] = part[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += rd[w * 16 + row];
      lpn[i] = f.beta * t;
    }
  } else if (N.T.kind == MFM_TARGET_LGCP) {
    double lik[4] = {0, 0, 0, 0}, quad[4] = {0, 0, 0, 0};
    layer_gemm<1, NW, 1>(T.bX() + 4, T.L.ldx, N.T.KinvP, N.T.kbias, N.dp / 16, N.dp / 16, wave, T.lane,
                         [&](int q, int nt, int m, f32x4 acc, float kb) {
                           const int col = nt * 16 + c;
#pragma unroll
                           for (int i = 0; i < 4; ++i) {
                             float gv = 0.f;
                             if (col < d) {
                               const float xv = T.bX()[(4 * g + i) * T.L.ldx + 4 + col], yv = acc[i] + kb, ex = expf(xv);
                               gv = (float)f.beta * (N.T.counts[col] - N.T.poisson_a * ex) - yv;
                               lik[i] += (double)xv * (double)N.T.counts[col] - (double)N.T.poisson_a * (double)ex;
                               quad[i] += (double)(xv - N.T.mu) * (double)yv;
                             }
#pragma unroll
                             for (int qq = 0; qq < TPW; ++qq)
                               if (qq == q) gnew[qq][i] = gv;
                           }
                         });
    double* rd = reinterpret_cast<double*>(T.red(0));      // slots 0..1: lik, 2..3: quad, as [NW][16] doubles
    double* rq = reinterpret_cast<double*>(T.red(2));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { lik[i] += __shfl_xor(lik[i], o, 64); quad[i] += __shfl_xor(quad[i], o, 64); }
      if (c == 0) { rd[wave * 16 + 4 * g + i] = lik[i]; rq[wave * 16 + 4 * g + i] = quad[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      double sl = 0.0, sq = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { sl += rd[w * 16 + row]; sq += rq[w * 16 + row]; }
      lpn[i] = f.beta * sl - 0.5 * sq + (double)N.T.log_norm;
    }
  } else {
    double* rd = reinterpret_cast<double*>(T.red(0));
    const bool lanes16 = N.T.n_modes <= 16;
    if (lanes16 ? threadIdx.x < 256 : threadIdx.x < 16) {
      double lp; float gg[8];
      const int row = lanes16 ? (int)(threadIdx.x >> 4) : (int)threadIdx.x;
      if (lanes16) gmm_eval_lanes16<8>(N.T, T.bX() + row * T.L.ldx + 4, threadIdx.x & 15, &lp, gg);
      else gmm_eval<8>(N.T, T.bX() + row * T.L.ldx + 4, &lp, gg);
      if (!lanes16 || (threadIdx.x & 15) == 0) {
        rd[row] = lp;
        for (int j = 0; j < d; ++j) T.gcs()[row * 24 + j] = gg[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) lpn[i] = f.beta * rd[4 * g + i];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = (wave + NW * q) * 16 + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) gnew[q][i] = col < d ? (float)f.beta * T.gcs()[(4 * g + i) * 24 + col] : 0.f;
    }
  }
  // ---- accept / reject (:271-278 / :253-260); the acceptance probability is NOT clipped ----
  bool acc[4];
  float aprob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = b0 + 4 * g + i;
    const double lp_old = f.logp[b];
    const double la = lpn[i] - (double)ell[i] - lp_old - (double)vol0[i] + (double)lq_ref[i];
    const double ap = exp(la);
    const double u = uniform01(split_at(kb[i], 4, 1), 0, 1);
    acc[i] = u <= ap;                     // NaN compares false -> reject
    aprob[i] = (float)ap;
    if (a.rp.diag && wave == 0 && c == 0) { double* o = a.rp.diag + 4 * (size_t)b; o[0] = vol0[i]; o[1] = ell[i]; o[2] = lpn[i]; o[3] = la; }
  }
  __syncthreads();      // every wave has read the OLD log-densities before wave 0 publishes the accepted ones
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = (wave + NW * q) * 16 + c;
    if (col < d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t o = (size_t)(b0 + 4 * g + i) * d + col;
        if (f.proposed) f.proposed[o] = y[q][i];
        if (acc[i]) { f.pos[o] = y[q][i]; f.grad[o] = gnew[q][i]; }
      }
    }
  }
  if (wave == 0 && c == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = b0 + 4 * g + i;
      if (acc[i]) f.logp[b] = lpn[i];
      if (f.acc_prob) f.acc_prob[b] = aprob[i];
      if (f.accepted) f.accepted[b] = acc[i] ? 1 : 0;
      if (f.nsteps) f.nsteps[b] = natt_tot[i];
    }
  }
#ifdef MFM_STAMPS
  if (g_flow_dbg && threadIdx.x == 0) {
    unsigned long long* o = g_flow_dbg + blockIdx.x * 64;
    o[0] = __builtin_amdgcn_s_memtime() - fc0_; o[1] = __builtin_amdgcn_s_memrealtime() - fr0_;
    o[2] = T.n_eval; o[3] = T.cyc_eval; o[4] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // XCC_ID
  }
#endif
}

#include "ode_fast.hip"
#include "ode_fixed.hip"
#include "ode_d2.hip"

// ---- launchers -----------------------------------------------------------------------------------------------
// ODE_NW: waves per workgroup of the solver kernels.  4 = one wave per SIMD with a 512-register budget (the seven
// Runge-Kutta stages stay in registers without spilling); 8 = two per SIMD with 256 registers each.
#ifndef ODE_NW
#define ODE_NW 8
#endif
static int ode_check(const NetDev& n, size_t& sm, int& tpw) {
  const OdeLds L = ode_lds_layout(n, ODE_NW);
  sm = (size_t)L.total * sizeof(float);
  tpw = (n.dp / 16 + ODE_NW - 1) / ODE_NW;
  if (sm > 160 * 1024 || tpw > 2 * (8 / ODE_NW) || n.hx1 > 16 * 2 * ODE_NW) return -3;
  return 0;
}
#define ODE_LAUNCH_T(KERN, T, GRID, ...)                                                                               \
  do {                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)KERN<T, ODE_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);      \
    hipLaunchKernelGGL((KERN<T, ODE_NW>), GRID, dim3(ODE_NW * 64), sm, stream, __VA_ARGS__);                           \
  } while (0)
#if ODE_NW == 8
#define ODE_LAUNCH(KERN, GRID, ...)                                                                                    \
  do { if (tpw <= 1) ODE_LAUNCH_T(KERN, 1, GRID, __VA_ARGS__); else ODE_LAUNCH_T(KERN, 2, GRID, __VA_ARGS__); } while (0)
#else
#define ODE_LAUNCH(KERN, GRID, ...)                                                                                    \
  do { if (tpw <= 1) ODE_LAUNCH_T(KERN, 1, GRID, __VA_ARGS__); else ODE_LAUNCH_T(KERN, 4, GRID, __VA_ARGS__); } while (0)
#endif

int launch_ode_transform(const OdeArgs& a, hipStream_t stream) {
  size_t sm; int tpw;
  if (ode_check(a.net, sm, tpw)) return -3;
  if (a.hutch) launch_probe(a.per_chain_keys ? 0 : 1, a.keys, a.key, 0, 0, 0, a.n, a.net.d, const_cast<float*>(a.z1), stream);
  if (a.fixed_steps > 0) {
    if (!fast::shape_ok(a.net, a.hutch) || a.rp.dt) return -4;
    return fast::launch_transform_fixed(a, a.fixed_method, a.fixed_steps, a.fast_scr, stream);
  }
  if (d2::use_for(a.net, a.hutch, a.n)) return d2::launch_transform(a, stream);
  if (fast::shape_ok(a.net, a.hutch) && !g_sw.generic_ode)
    return fast::tile_width(a.net) == 256 ? fast::launch_transform_t<256>(a, a.fast_scr, stream) : fast::launch_transform_t<128>(a, a.fast_scr, stream);
  ODE_LAUNCH(ode_transform_kernel, dim3(a.n / 16), a);
  return 0;
}
int launch_vf_apply(const NetDev& n, const float* x, const float* t, const float* tan, int cnt, float* v, float* jvp, hipStream_t stream) {
  size_t sm; int tpw;
  if (ode_check(n, sm, tpw)) return -3;
  ODE_LAUNCH(vf_apply_kernel, dim3(cnt / 16), n, x, t, tan, cnt, v, jvp);
  return 0;
}
int launch_flow_step(const OdeArgs& a, const FlowArgs& f, const NoiseArgs& nz, hipStream_t stream) {
  size_t sm; int tpw;
  if (ode_check(a.net, sm, tpw)) return -3;
  launch_probe(2, nullptr, f.key, f.n_total, f.chain_offset, 0, a.n, a.net.d, const_cast<float*>(a.zgen), stream);     // key_gen
  if (a.hutch) {
    launch_probe(2, nullptr, f.key, f.n_total, f.chain_offset, 3, a.n, a.net.d, const_cast<float*>(a.z1), stream);     // key_hutch2
    launch_probe(2, nullptr, f.key, f.n_total, f.chain_offset, 2, a.n, a.net.d, const_cast<float*>(a.z2), stream);     // key_hutch1
  }
  if (a.fixed_steps > 0) {
    if (!fast::shape_ok(a.net, a.hutch) || a.rp.dt || (f.mode & 0xFF) != MFM_FLOW_RWMH) return -4;
    return fast::launch_flow_fixed(a, f, a.fixed_method, a.fixed_steps, a.fast_scr, stream);
  }
  if (d2::use_for(a.net, a.hutch, a.n)) return d2::launch_flow(a, f, stream);
  if (fast::shape_ok(a.net, a.hutch) && !g_sw.generic_ode)
    return fast::tile_width(a.net) == 256 ? fast::launch_flow_t<256>(a, f, nz, a.fast_scr, stream) : fast::launch_flow_t<128>(a, f, nz, a.fast_scr, stream);
  ODE_LAUNCH(flow_step_kernel, dim3(a.n / 16), a, f);
  return 0;
}

#ifdef MFM_STAMPS
int launch_eval_stamps(const NetDev& n, const float* x, const float* t, const float* tan, int cnt, int reps, unsigned long long* stamps, hipStream_t stream) {
  size_t sm; int tpw;
  if (ode_check(n, sm, tpw)) return -3;
  ODE_LAUNCH(eval_stamps_kernel, dim3(cnt / 16), n, x, t, tan, reps, stamps);
  return 0;
}
#endif
