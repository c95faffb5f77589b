/* libmfm_ref: float64 C / OpenMP restatement of the MFM inner loop at the headline configuration -- ORACLE, TEST INFRASTRUCTURE.
 *
 * Same standing as the numpy package around it (oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, as the checker or as the timed CPU port; nothing under mfm_amd/ does.  PARITY UNPINNED: the
 * reference (pure Python on JAX) cannot run in the build container and holds no golden vectors for this path; this file follows
 * the reference's source lines cited at every function and is itself checked against the numpy restatement
 * (tests/test_oracle_cref.py: values to 1e-10, attempted-step counts equal).
 *
 * Scope: what BASELINE configs[2] exercises -- the PhiFour target (distributions.py:131-164), the MALA step
 * (bblackjax/mcmc/mala.py:57-120, diffusions.py:19-34, proposal.py:104-112,157-159,178-186), VectorFieldNet forward / x-JVP /
 * parameter gradient with relu (exe_flow_matching.py:56-90), the flow-matching loss (:171-179) and the adaptive Dopri5 CNF solves
 * with the Hutchinson log-det (:206-242, jax.experimental.ode.odeint restated as in oracle/ode.py).  Random draws are INPUTS
 * (made by oracle/prng.py); other targets, activations and the exact trace stay with the numpy oracle.
 *
 * One chain per OpenMP task; every chain runs its own adaptive step sequence, as under jax.vmap. */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define MAXL 16

typedef struct {
  int d, F, lt, lx, lxt;  /* dimension, Fourier frequencies, hidden layers of the time / x / joint branch */
  const int* shapes;      /* [n_layers][2] = (fan_in, fan_out), flax creation order (exe_flow_matching.py:74-86) */
  const float* flat;      /* canonical flat float32 parameters: per layer kernel [in][out], then bias */
  const double* fourier;  /* [F] (:70) */
  double grad_clip;       /* 0: none (:88-89: clip only when dim > 128) */
  double coef, beta;      /* PhiFour: coef = a * dim (distributions.py:132,150), beta (:157) */
} mfmref_net;

typedef struct {
  int n, maxw;
  int fin[MAXL], fout[MAXL];
  double* W[MAXL]; double* b[MAXL];
  size_t off_w[MAXL], off_b[MAXL], n_params;
} netd;

static int net_build(const mfmref_net* N, netd* P) {
  P->n = N->lt + N->lx + 1 + N->lxt + 1;
  if (P->n > MAXL || N->lt < 1 || N->lx < 1 || N->lxt < 1) return -1;
  size_t o = 0; P->maxw = N->d > 2 * N->F ? N->d : 2 * N->F;
  for (int l = 0; l < P->n; ++l) {
    const int fi = N->shapes[2 * l], fo = N->shapes[2 * l + 1];
    P->fin[l] = fi; P->fout[l] = fo;
    if (fi > P->maxw) P->maxw = fi;
    if (fo > P->maxw) P->maxw = fo;
    P->off_w[l] = o; o += (size_t)fi * fo; P->off_b[l] = o; o += fo;
    P->W[l] = (double*)malloc(sizeof(double) * (size_t)fi * fo);
    P->b[l] = (double*)malloc(sizeof(double) * fo);
    for (size_t i = 0; i < (size_t)fi * fo; ++i) P->W[l][i] = (double)N->flat[P->off_w[l] + i];   /* float32 params promoted (multi_modal.py:14) */
    for (int i = 0; i < fo; ++i) P->b[l][i] = (double)N->flat[P->off_b[l] + i];
  }
  P->n_params = o;
  return 0;
}
static void net_free(netd* P) { for (int l = 0; l < P->n; ++l) { free(P->W[l]); free(P->b[l]); } }

int mfmref_threads(void) { return omp_get_max_threads(); }
void mfmref_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* ---- PhiFour (distributions.py:131-164; Dirichlet boundary :144) ------------------------------------------------------------ */
static double phi4_loglik(const double* x, int d, double coef, double beta) {
  double U = 0.0, V = 0.0, prev = 0.0;
  for (int i = 0; i <= d; ++i) { const double cur = i < d ? x[i] : 0.0, df = cur - prev; U += df * df; prev = cur; }   /* :148 */
  for (int i = 0; i < d; ++i) { const double q = 1.0 - x[i] * x[i]; V += q * q; }                                     /* :133 */
  return -beta * (U / 2.0 * coef + V / 4.0 / coef);                                                                  /* :134,149-151,157 */
}
static void phi4_grad(const double* x, int d, double coef, double beta, double* g) {
  for (int i = 0; i < d; ++i) {
    const double l = i > 0 ? x[i - 1] : 0.0, r = i + 1 < d ? x[i + 1] : 0.0, lap = 2.0 * x[i] - l - r;
    g[i] = -beta * (coef * lap - x[i] * (1.0 - x[i] * x[i]) / coef);
  }
}
static void phi4_hvp(const double* x, const double* v, int d, double coef, double beta, double* h) {
  for (int i = 0; i < d; ++i) {
    const double l = i > 0 ? v[i - 1] : 0.0, r = i + 1 < d ? v[i + 1] : 0.0, lap = 2.0 * v[i] - l - r;
    h[i] = -beta * (coef * lap - (1.0 - 3.0 * x[i] * x[i]) * v[i] / coef);
  }
}

/* tempered target beta_t * loglik + logprior (exe_flow_matching.py:301; logprior = 0: distributions.py:159-160) */
int mfmref_phi4_value_grad(const double* x, int B, int d, double coef, double beta, double temper, double* logp, double* grad) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    logp[b] = temper * phi4_loglik(x + (size_t)b * d, d, coef, beta);
    phi4_grad(x + (size_t)b * d, d, coef, beta, grad + (size_t)b * d);
    for (int i = 0; i < d; ++i) grad[(size_t)b * d + i] *= temper;
  }
  return 0;
}

/* ---- MALA step, state updated in place (mala.py:86-118 as written: p = min(1, exp(prev_E - new_E))) --------------------------- */
int mfmref_mala_step(double* x, double* logp, double* grad, const double* noise, const double* u, int B, int d, double step,
                     double coef, double beta, double temper, int textbook, double* p_accept, unsigned char* accepted) {
#pragma omp parallel
  {
    double* xn = (double*)malloc(sizeof(double) * 2 * d); double* gn = xn + d;
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      double* xb = x + (size_t)b * d; double* gb = grad + (size_t)b * d; const double* nb = noise + (size_t)b * d;
      const double s2e = sqrt(2.0 * step);
      for (int i = 0; i < d; ++i) xn[i] = xb[i] + step * gb[i] + s2e * nb[i];                      /* diffusions.py:25-30 */
      const double lpn = temper * phi4_loglik(xn, d, coef, beta);                                  /* diffusions.py:32 */
      phi4_grad(xn, d, coef, beta, gn);
      for (int i = 0; i < d; ++i) gn[i] *= temper;
      double th1 = 0.0, th2 = 0.0;
      for (int i = 0; i < d; ++i) {
        const double a = xn[i] - xb[i] - step * gb[i], c = xb[i] - xn[i] - step * gn[i];
        th1 += a * a; th2 += c * c;
      }
      const double new_E = -logp[b] + 0.25 * (1.0 / step) * th1;                                   /* mala.py:68-79, proposal.py:157 */
      const double prev_E = -lpn + 0.25 * (1.0 / step) * th2;                                      /* proposal.py:158 */
      double delta = prev_E - new_E;                                                               /* proposal.py:104 */
      if (textbook) delta = -delta;
      if (isnan(delta)) delta = -INFINITY;                                                         /* proposal.py:105 */
      const double p = fmin(exp(delta), 1.0);                                                      /* proposal.py:178 */
      const int acc = u[b] < p;                                                                    /* proposal.py:179 */
      if (acc) { memcpy(xb, xn, sizeof(double) * d); memcpy(gb, gn, sizeof(double) * d); logp[b] = lpn; }
      if (p_accept) p_accept[b] = p;
      if (accepted) accepted[b] = (unsigned char)acc;
    }
    free(xn);
  }
  return 0;
}

/* ---- VectorFieldNet (exe_flow_matching.py:56-90) ------------------------------------------------------------------------------ */
/* out = in @ W + b, W [fin][fout] (flax Dense) */
static void dense(const double* W, const double* b, int fin, int fout, const double* in, double* out) {
  for (int j = 0; j < fout; ++j) out[j] = 0.0;
  for (int k = 0; k < fin; ++k) {
    const double a = in[k];
    if (a == 0.0) continue;                       /* (relu outputs: exact zeros add nothing) */
    const double* w = W + (size_t)k * fout;
    for (int j = 0; j < fout; ++j) out[j] += a * w[j];
  }
  if (b) for (int j = 0; j < fout; ++j) out[j] += b[j];
}

/* the same for a value row and a tangent row at once (no bias on the tangent): one pass over W */
static void dense2(const double* W, const double* b, int fin, int fout, const double* in, const double* tin, double* out, double* tout) {
  for (int j = 0; j < fout; ++j) { out[j] = 0.0; tout[j] = 0.0; }
  for (int k = 0; k < fin; ++k) {
    const double a = in[k], c = tin[k];
    const double* w = W + (size_t)k * fout;
    if (a != 0.0 && c != 0.0) for (int j = 0; j < fout; ++j) { out[j] += a * w[j]; tout[j] += c * w[j]; }
    else if (a != 0.0) for (int j = 0; j < fout; ++j) out[j] += a * w[j];
    else if (c != 0.0) for (int j = 0; j < fout; ++j) tout[j] += c * w[j];
  }
  for (int j = 0; j < fout; ++j) out[j] += b[j];
}

typedef struct {         /* per-thread scratch: activations entering each layer and pre-activations (the backward pass needs both) */
  double* in[MAXL]; double* pre[MAXL]; double* tin; double* tout; double* g; double* hv; double* buf; double* st; double* sx;
} ws_t;
static void ws_alloc(ws_t* w, const netd* P, int d) {
  for (int l = 0; l < P->n; ++l) { w->in[l] = (double*)malloc(sizeof(double) * P->fin[l]); w->pre[l] = (double*)malloc(sizeof(double) * P->fout[l]); }
  w->tin = (double*)malloc(sizeof(double) * 2 * P->maxw); w->tout = (double*)malloc(sizeof(double) * 2 * P->maxw);
  w->g = (double*)malloc(sizeof(double) * d); w->hv = (double*)malloc(sizeof(double) * d); w->buf = (double*)malloc(sizeof(double) * 2 * P->maxw);
  w->st = (double*)malloc(sizeof(double) * P->maxw); w->sx = (double*)malloc(sizeof(double) * P->maxw);
}
static void ws_free(ws_t* w, const netd* P) {
  for (int l = 0; l < P->n; ++l) { free(w->in[l]); free(w->pre[l]); }
  free(w->tin); free(w->tout); free(w->g); free(w->hv); free(w->buf); free(w->st); free(w->sx);
}

/* v(x, t) [d]; with z: also (J_x v) z in jv [d].  Leaves the layer inputs / pre-activations and the clipped gradient term in w. */
static void field_eval(const mfmref_net* N, const netd* P, ws_t* w, const double* x, double t, const double* z, double* v, double* jv) {
  const int d = N->d, F = N->F, lt = N->lt, lx = N->lx, lxt = N->lxt;
  int li = 0;
  /* time branch (:70-75) */
  double* s = w->in[0];
  for (int k = 0; k < F; ++k) { const double deg = 2.0 * M_PI * N->fourier[k] * t; s[k] = cos(deg); s[F + k] = sin(deg); }
  const double* cur = s;
  for (int k = 0; k < lt; ++k, ++li) {
    if (cur != w->in[li]) memcpy(w->in[li], cur, sizeof(double) * P->fin[li]);
    dense(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], w->pre[li]);
    double* o = w->buf; for (int j = 0; j < P->fout[li]; ++j) o[j] = w->pre[li][j] > 0.0 ? w->pre[li][j] : 0.0;
    cur = o;
    if (k + 1 < lt) { memcpy(w->in[li + 1], o, sizeof(double) * P->fout[li]); cur = w->in[li + 1]; }
  }
  const int ht = lt ? P->fout[lt - 1] : 2 * F;
  double* st = w->st; memcpy(st, cur, sizeof(double) * ht);
  /* x branch (:78-79), tangent alongside */
  double* ts = z ? w->tin : NULL;
  if (z) memcpy(ts, z, sizeof(double) * d);
  cur = x;
  int wx = d;
  for (int k = 0; k < lx; ++k, ++li) {
    if (cur != w->in[li]) memcpy(w->in[li], cur, sizeof(double) * P->fin[li]);
    if (z) {
      dense2(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], ts, w->pre[li], w->tout);
      for (int j = 0; j < P->fout[li]; ++j) ts[j] = w->pre[li][j] > 0.0 ? w->tout[j] : 0.0;
    } else dense(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], w->pre[li]);
    double* o = w->buf; for (int j = 0; j < P->fout[li]; ++j) o[j] = w->pre[li][j] > 0.0 ? w->pre[li][j] : 0.0;
    cur = o; wx = P->fout[li];
    if (k + 1 < lx) { memcpy(w->in[li + 1], o, sizeof(double) * wx); cur = w->in[li + 1]; }
  }
  double* sx = w->sx; memcpy(sx, cur, sizeof(double) * wx);
  /* gate (:81) */
  const int lg = li;
  memcpy(w->in[li], st, sizeof(double) * ht);
  dense(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], w->pre[li]);
  ++li;
  /* joint branch (:83-85) */
  memcpy(w->in[li], sx, sizeof(double) * wx); memcpy(w->in[li] + wx, st, sizeof(double) * ht);
  if (z) for (int j = wx; j < wx + ht; ++j) ts[j] = 0.0;
  for (int k = 0; k < lxt; ++k, ++li) {
    if (z) {
      dense2(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], ts, w->pre[li], w->tout);
      for (int j = 0; j < P->fout[li]; ++j) ts[j] = w->pre[li][j] > 0.0 ? w->tout[j] : 0.0;
    } else dense(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], w->pre[li]);
    for (int j = 0; j < P->fout[li]; ++j) w->in[li + 1][j] = w->pre[li][j] > 0.0 ? w->pre[li][j] : 0.0;
  }
  /* output (:86) and the gate term (:88-90) */
  if (z) {
    dense2(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], ts, w->pre[li], w->tout);
    phi4_hvp(x, z, d, N->coef, N->beta, w->hv);
  } else dense(P->W[li], P->b[li], P->fin[li], P->fout[li], w->in[li], w->pre[li]);
  phi4_grad(x, d, N->coef, N->beta, w->g);
  const double* nn_t = w->pre[lg]; const double* nn_xt = w->pre[li];
  for (int j = 0; j < d; ++j) {
    double g = w->g[j]; int inside = 1;
    if (N->grad_clip > 0.0) { inside = fabs(g) <= N->grad_clip; g = g > N->grad_clip ? N->grad_clip : g < -N->grad_clip ? -N->grad_clip : g; }
    w->g[j] = g;
    v[j] = nn_xt[j] + nn_t[j] * g;
    if (z) jv[j] = w->tout[j] + nn_t[j] * (inside ? w->hv[j] : 0.0);
  }
}

int mfmref_vfield(const mfmref_net* N, const double* x, const double* t, const double* tangent, int B, double* v, double* jv) {
  netd P; if (net_build(N, &P)) return -1;
  const int d = N->d;
#pragma omp parallel
  {
    ws_t w; ws_alloc(&w, &P, d);
#pragma omp for schedule(dynamic, 4)
    for (int b = 0; b < B; ++b)
      field_eval(N, &P, &w, x + (size_t)b * d, t[b], tangent ? tangent + (size_t)b * d : NULL, v + (size_t)b * d, jv ? jv + (size_t)b * d : NULL);
    ws_free(&w, &P);
  }
  net_free(&P);
  return 0;
}

/* flow-matching loss = SUM of squared residuals and its parameter gradient (exe_flow_matching.py:171-178, :364-365); `cond`,
 * `target`, `t` are the batch of :151-169 (built by the caller from its draws).  grads: canonical flat layout, float32 like the params. */
int mfmref_fm_loss_grad(const mfmref_net* N, const double* cond, const double* target, const double* t, int B, double* loss, float* grads) {
  netd P; if (net_build(N, &P)) return -1;
  const int d = N->d, lt = N->lt, lx = N->lx, lxt = N->lxt, nth = omp_get_max_threads();
  enum { R = 16 };         /* chains per block: a layer's gradient rows are touched once per block, not once per chain */
  double* acc = (double*)calloc((size_t)nth * P.n_params, sizeof(double));
  double* lpart = (double*)calloc(nth, sizeof(double));
  const int nblk = (B + R - 1) / R;
#pragma omp parallel
  {
    ws_t w; ws_alloc(&w, &P, d);
    double* G = acc + (size_t)omp_get_thread_num() * P.n_params;
    double* v = (double*)malloc(sizeof(double) * d); double* dz = (double*)malloc(sizeof(double) * 2 * P.maxw);
    double* ds = (double*)malloc(sizeof(double) * 2 * P.maxw); double* dst = (double*)malloc(sizeof(double) * 2 * P.maxw);
    double* dv = (double*)malloc(sizeof(double) * d);
    double* IN[MAXL]; double* DZ[MAXL];        /* [R][fin], [R][fout] of every layer */
    for (int l = 0; l < P.n; ++l) { IN[l] = (double*)malloc(sizeof(double) * R * P.fin[l]); DZ[l] = (double*)malloc(sizeof(double) * R * P.fout[l]); }
    double lsum = 0.0;
#pragma omp for schedule(dynamic, 1)
    for (int blk = 0; blk < nblk; ++blk) {
      const int b0 = blk * R, nr = B - b0 < R ? B - b0 : R;
      for (int r = 0; r < nr; ++r) {
        const int b = b0 + r;
        field_eval(N, &P, &w, cond + (size_t)b * d, t[b], NULL, v, NULL);
        for (int j = 0; j < d; ++j) { const double q = v[j] - target[(size_t)b * d + j]; lsum += q * q; dv[j] = 2.0 * q; }      /* :177-178 */
        for (int l = 0; l < P.n; ++l) memcpy(IN[l] + (size_t)r * P.fin[l], w.in[l], sizeof(double) * P.fin[l]);
        /* backward (oracle/vfield.py: backward): dz of every layer, ds = dz W^T */
#define KEEP(l, dzv) memcpy(DZ[l] + (size_t)r * P.fout[l], (dzv), sizeof(double) * P.fout[l])
#define BACK(l, dzv, out) do { const int fi_ = P.fin[l], fo_ = P.fout[l]; \
          for (int k_ = 0; k_ < fi_; ++k_) { double s_ = 0.0; const double* wr_ = P.W[l] + (size_t)k_ * fo_; for (int j_ = 0; j_ < fo_; ++j_) s_ += (dzv)[j_] * wr_[j_]; (out)[k_] = s_; } } while (0)
        int li = P.n - 1;
        KEEP(li, dv); BACK(li, dv, ds);
        for (int k = 0; k < lxt; ++k) {
          --li;
          for (int j = 0; j < P.fout[li]; ++j) dz[j] = w.pre[li][j] > 0.0 ? ds[j] : 0.0;
          KEEP(li, dz); BACK(li, dz, ds);
        }
        const int hx = P.fout[lt + lx - 1], ht = P.fout[lt - 1];
        --li;                                                  /* gate layer: dz = dv * clip(grad log pi); ds = [d_sx (hx), d_st (ht)] */
        for (int j = 0; j < d; ++j) dz[j] = dv[j] * w.g[j];
        KEEP(li, dz); BACK(li, dz, dst);
        for (int j = 0; j < ht; ++j) dst[j] += ds[hx + j];
        for (int k = 0; k < lx; ++k) {                         /* x branch, reversed */
          --li;
          for (int j = 0; j < P.fout[li]; ++j) dz[j] = w.pre[li][j] > 0.0 ? ds[j] : 0.0;
          KEEP(li, dz);
          if (k + 1 < lx) BACK(li, dz, ds);
        }
        memcpy(ds, dst, sizeof(double) * ht);
        for (int k = 0; k < lt; ++k) {                         /* time branch, reversed */
          --li;
          for (int j = 0; j < P.fout[li]; ++j) dz[j] = w.pre[li][j] > 0.0 ? ds[j] : 0.0;
          KEEP(li, dz);
          if (k + 1 < lt) BACK(li, dz, ds);
        }
      }
      /* dW += in^T dz, db += sum dz over the block's chains (in chain order) */
      for (int l = 0; l < P.n; ++l) {
        const int fi = P.fin[l], fo = P.fout[l];
        double* gw = G + P.off_w[l]; double* gb = G + P.off_b[l];
        for (int k = 0; k < fi; ++k) {
          double* row = gw + (size_t)k * fo;
          for (int r = 0; r < nr; ++r) {
            const double a_ = IN[l][(size_t)r * fi + k];
            if (a_ == 0.0) continue;
            const double* dzr = DZ[l] + (size_t)r * fo;
            for (int j = 0; j < fo; ++j) row[j] += a_ * dzr[j];
          }
        }
        for (int r = 0; r < nr; ++r) for (int j = 0; j < fo; ++j) gb[j] += DZ[l][(size_t)r * fo + j];
      }
    }
    lpart[omp_get_thread_num()] = lsum;
    free(v); free(dz); free(ds); free(dst); free(dv);
    for (int l = 0; l < P.n; ++l) { free(IN[l]); free(DZ[l]); }
    ws_free(&w, &P);
  }
  double L = 0.0; for (int i = 0; i < nth; ++i) L += lpart[i];
  *loss = L;
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < (long long)P.n_params; ++p) { double s = 0.0; for (int i = 0; i < nth; ++i) s += acc[(size_t)i * P.n_params + p]; grads[p] = (float)s; }
  free(acc); free(lpart);
  net_free(&P);
  return 0;
}

/* ---- adaptive Dormand-Prince 5(4) as jax.experimental.ode.odeint (restated in oracle/ode.py; exe_flow_matching.py:345-349) ------ */
static const double ALPHA[6] = {1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
static const double BETA[6][6] = {
  {1.0 / 5},
  {3.0 / 40, 9.0 / 40},
  {44.0 / 45, -56.0 / 15, 32.0 / 9},
  {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
  {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
  {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
static const double C_SOL[7] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0};
static const double C_ERR[7] = {35.0 / 384 - 1951.0 / 21600, 0, 500.0 / 1113 - 22642.0 / 50085, 125.0 / 192 - 451.0 / 720,
                                -2187.0 / 6784 - -12231.0 / 42400, 11.0 / 84 - 649.0 / 6300, -1.0 / 60.0};
static const double C_MID[7] = {6025192743.0 / 30085553152.0 / 2, 0, 51252292925.0 / 65400821598.0 / 2, -2691868925.0 / 45128329728.0 / 2,
                                187940372067.0 / 1594534317056.0 / 2, -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};

typedef struct { const mfmref_net* N; const netd* P; ws_t* w; const double* z; int sign; double* v; double* jv; long long evals; } rhs_t;

/* RHS of the augmented ODE (:208-218 forward, :225-239 inverse), Hutchinson estimator z^T (J z) with z fixed for the solve (:212-214) */
static void rhs(rhs_t* R, const double* y, double t, double* out) {
  const int d = R->N->d;
  const double tt = R->sign > 0 ? t : 1.0 - t;                                        /* :229 */
  field_eval(R->N, R->P, R->w, y, tt, R->z, R->v, R->jv);
  double q = 0.0; for (int j = 0; j < d; ++j) q += R->z[j] * R->jv[j];
  if (R->sign > 0) { for (int j = 0; j < d; ++j) out[j] = R->v[j]; out[d] = -q; }       /* :218 */
  else { for (int j = 0; j < d; ++j) out[j] = -R->v[j]; out[d] = q; }                   /* :230,239 */
  R->evals++;
}
static double norm2(const double* a, const double* scale, int n) { double s = 0.0; for (int i = 0; i < n; ++i) { const double q = a[i] / scale[i]; s += q * q; } return sqrt(s); }

/* one chain: y0 = (x0, 0) at t = 0 to t = 1; returns the attempted-step count */
/* rp_dt / rp_acc (replay, PARITY INSTRUMENTATION as in oracle/ode.py: odeint): the solve takes the prescribed step sizes [cap] and accept
 * decisions [cap] instead of its controller's.  rec_dt / rec_acc (record): rec_dt[0] = the initial step, rec_dt[j + 1] = the step size
 * after attempt j (0 once the chain has reached the end), rec_acc[j] = attempt j accepted -- the `dt_seq` / `acc_seq` of oracle/ode.py. */
static long long solve_chain(rhs_t* R, const double* x0, double rtol, double atol, int mxstep, double* xout, double* ldj,
                             const double* rp_dt, const unsigned char* rp_acc, double* rec_dt, unsigned char* rec_acc, int cap) {
  const int d = R->N->d, n = d + 1;
  double* mem = (double*)malloc(sizeof(double) * n * 24);
  double *y = mem, *f = mem + n, *y1 = mem + 2 * n, *err = mem + 3 * n, *yi = mem + 4 * n, *scale = mem + 5 * n, *tmp = mem + 6 * n;
  double* k[7]; for (int i = 0; i < 7; ++i) k[i] = mem + (7 + i) * n;
  double* co[5]; for (int i = 0; i < 5; ++i) co[i] = mem + (14 + i) * n;
  double* nco[5]; for (int i = 0; i < 5; ++i) nco[i] = mem + (19 + i) * n;
  memcpy(y, x0, sizeof(double) * d); y[d] = 0.0;                                       /* :220 / :241 */
  double t = 0.0;
  rhs(R, y, t, f);
  /* initial step (Hairer-Norsett-Wanner II.4, order 4): oracle/ode.py: initial_step_size */
  for (int i = 0; i < n; ++i) scale[i] = atol + fabs(y[i]) * rtol;
  const double d0 = norm2(y, scale, n), d1 = norm2(f, scale, n);
  const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
  for (int i = 0; i < n; ++i) yi[i] = y[i] + h0 * f[i];
  rhs(R, yi, t + h0, tmp);
  for (int i = 0; i < n; ++i) tmp[i] -= f[i];
  const double d2 = norm2(tmp, scale, n) / h0;
  const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
  double dt = fmin(100.0 * h0, h1);
  if (!(dt >= 0.0) && !isnan(dt)) dt = 0.0;
  if (rp_dt) dt = rp_dt[0];
  if (rec_dt) rec_dt[0] = dt;
  double last_t = t;
  for (int c = 0; c < 5; ++c) memcpy(co[c], y, sizeof(double) * n);
  long long natt = 0;
  while (t < 1.0 && natt < mxstep && dt > 0.0) {
    memcpy(k[0], f, sizeof(double) * n);
    for (int s = 0; s < 6; ++s) {
      for (int i = 0; i < n; ++i) { double a = 0.0; for (int m = 0; m <= s; ++m) a += BETA[s][m] * k[m][i]; yi[i] = y[i] + dt * a; }
      rhs(R, yi, t + dt * ALPHA[s], k[s + 1]);
    }
    double r2 = 0.0;
    for (int i = 0; i < n; ++i) {
      double a = 0.0, e = 0.0;
      for (int m = 0; m < 7; ++m) { a += C_SOL[m] * k[m][i]; e += C_ERR[m] * k[m][i]; }
      y1[i] = dt * a + y[i]; err[i] = dt * e;
      const double tol = atol + rtol * fmax(fabs(y[i]), fabs(y1[i])), q = err[i] / tol;
      r2 += q * q;
    }
    const double ratio = sqrt(r2 / n);
    /* interpolation coefficients of this step (interp_fit_dopri) */
    for (int i = 0; i < n; ++i) {
      double a = 0.0; for (int m = 0; m < 7; ++m) a += C_MID[m] * k[m][i];
      const double ymid = y[i] + dt * a, dy0 = k[0][i], dy1 = k[6][i];
      nco[0][i] = -2.0 * dt * dy0 + 2.0 * dt * dy1 - 8.0 * y[i] - 8.0 * y1[i] + 16.0 * ymid;
      nco[1][i] = 5.0 * dt * dy0 - 3.0 * dt * dy1 + 18.0 * y[i] + 14.0 * y1[i] - 32.0 * ymid;
      nco[2][i] = -4.0 * dt * dy0 + dt * dy1 - 11.0 * y[i] - 5.0 * y1[i] + 16.0 * ymid;
      nco[3][i] = dt * dy0; nco[4][i] = y[i];
    }
    /* controller (optimal_step_size): dt * clip(0.9 ratio^(-1/5), dfactor, 10); NaN propagates (numpy maximum / minimum / clip do) */
    double ndt;
    if (ratio == 0.0) ndt = dt * 10.0;
    else {
      const double fr = pow(ratio, -1.0 / 5.0) * 0.9, dfac = ratio < 1.0 ? 1.0 : 0.2;
      ndt = isnan(fr) ? NAN : dt * fmin(10.0, fmax(fr, dfac));
    }
    if (!isnan(ndt) && ndt < 0.0) ndt = 0.0;
    int accept = ratio <= 1.0;                              /* (NaN: reject) */
    if (rp_dt) {
      const long long j = natt < cap - 1 ? natt : cap - 1, j1 = natt + 1 < cap - 1 ? natt + 1 : cap - 1;
      accept = rp_acc[j] != 0; ndt = rp_dt[j1];
    }
    if (accept) {
      for (int c = 0; c < 5; ++c) memcpy(co[c], nco[c], sizeof(double) * n);
      last_t = t; memcpy(y, y1, sizeof(double) * n); memcpy(f, k[6], sizeof(double) * n); t = t + dt;
    }
    dt = ndt;
    if (rec_acc && natt < cap) rec_acc[natt] = (unsigned char)accept;
    if (rec_dt && natt + 1 < cap) rec_dt[natt + 1] = t < 1.0 ? dt : 0.0;
    ++natt;
  }
  const double s = (1.0 - last_t) / (t - last_t);          /* value at the output time: the last accepted step's 4th-order interpolant */
  for (int i = 0; i < n; ++i) { const double o = (((co[0][i] * s + co[1][i]) * s + co[2][i]) * s + co[3][i]) * s + co[4][i]; if (i < d) xout[i] = o; else *ldj = o; }
  free(mem);
  return natt;
}

/* sign = +1: transform_and_logdet (:206-221); sign = -1: inverse_and_logdet (:223-242).  z [B][d]: the Hutchinson probes. */
int mfmref_cnf_solve(const mfmref_net* N, const double* x0, const double* z, int sign, double rtol, double atol, int mxstep, int B,
                     double* xout, double* ldj, long long* n_att, long long* n_evals,
                     const double* rp_dt, const unsigned char* rp_acc, double* rec_dt, unsigned char* rec_acc, int cap) {
  netd P; if (net_build(N, &P)) return -1;
  const int d = N->d;
  long long ev = 0;
#pragma omp parallel reduction(+ : ev)
  {
    ws_t w; ws_alloc(&w, &P, d);
    double* v = (double*)malloc(sizeof(double) * 2 * d);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      rhs_t R = {N, &P, &w, z + (size_t)b * d, sign, v, v + d, 0};
      n_att[b] = solve_chain(&R, x0 + (size_t)b * d, rtol, atol, mxstep, xout + (size_t)b * d, ldj + b,
                             rp_dt ? rp_dt + (size_t)b * cap : NULL, rp_acc ? rp_acc + (size_t)b * cap : NULL,
                             rec_dt ? rec_dt + (size_t)b * cap : NULL, rec_acc ? rec_acc + (size_t)b * cap : NULL, cap);
      ev += R.evals;
    }
    free(v);
    ws_free(&w, &P);
  }
  if (n_evals) *n_evals = ev;
  net_free(&P);
  return 0;
}
