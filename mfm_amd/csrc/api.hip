// C ABI of libmfm_hip (include/mfm.h).  Unity build: the kernel translation units are included here so one hipcc
// invocation produces the shared library.
#include "../../include/mfm.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>     // types and enums only: the library is resolved at run time (rccl_api), never linked

#include "mala.hip"
#include "lgcp.hip"
#include "hmc.hip"
#include "fm.hip"
#include "optim.hip"
#include "wgrad_sk.hip"
#include "noise.hip"
#include "ode.hip"
#include "anneal.hip"
#include "metrics.hip"
#include "cis.hip"
#include "wide.hip"
#include "smc.hip"

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(x)                                                                                        \
  do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MFM_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define LAUNCHCHK()                                                                                      \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(MFM_EHIP, "kernel launch: %s", hipGetErrorString(e_)); } while (0)

// ---- optional per-kernel timing with HIP events on the context's stream (bench.py roofline) -----------------------
enum { PROF_MALA = 0, PROF_FM = 1, PROF_WGRAD = 2, PROF_ADAM = 3, PROF_FLOW = 4, PROF_EVAL = 5, PROF_REDUCE = 6, PROF_NCLS = 8 };
struct Prof {
  bool on = false;
  unsigned mask = ~0u;            // classes that are bracketed by events
  std::vector<hipEvent_t> ev;     // pairs
  std::vector<int> cls;
  size_t used = 0;
};

// draws of the coming MALA + training iterations, produced in the tail of the flow-step kernel (noise.hip)
struct NoiseWs {
  int cap = 0;                                  // slots allocated
  draw_t *mala_n = nullptr, *fm_x0 = nullptr, *fm_eps = nullptr; double* mala_u = nullptr;
  float* fm_t = nullptr; uint32_t* d_keys = nullptr;      // [2][cap][2]
  int* counter = nullptr;
  std::vector<uint32_t> h_keys;                 // staging copy of the keys
  std::vector<Key2> gn, st;                     // keys of the slots
  int n_armed = 0;                              // slots the NEXT flow step will fill
  int n_valid = 0, cur_gn = 0, cur_st = 0;      // slots filled by the last flow step, consumption cursors
  bool no_mala0 = false;                        // slot 0 was keyed by the flow step that filled it: its MALA draws were not produced
};

// slots of mfm_get_counters
enum { CTR_MALA = 0, CTR_FM_TRAIN = 1, CTR_FM_EVAL = 2, CTR_SOLVES = 3, CTR_ATTEMPTS = 4, CTR_FIELD_EVALS = 5, CTR_OPT_STEPS = 6, CTR_MALA_BYTES = 7 };

struct mfm_ctx {
  Prof* prof;
  NoiseWs* noise;
  mfm_config cfg;
  hipStream_t stream;
  NetDev net;
  WsLayout ws;
  bool has_target, has_fourier;
  float *master, *mu, *nu, *Wp, *WpT, *bias, *fourier;
  float *acts, *dzs, *slabs, *dacts;
  double* loss_part; int loss_cap;
  float* wsk_partials = nullptr; int* wsk_tickets = nullptr;      // stream-K weight gradients (wgrad_sk.hip): partial blocks, arrival counters [2][n_jobs]
  WskConst* wsk_const = nullptr; WskWg* wsk_wg = nullptr;           // its tables, in device memory
  int wsk_G = 0, wsk_par = 0;                                       // its grid (0: the slab kernels serve this context), parity of the arrival counters
  int sus_par = 0;                                                  // parity of the training kernel's "suspicious values" word (flag[5 + parity])
  float* eval_pad = nullptr;   // [16][dim]: the last, partial 16-row tile of a mfm_fm_loss call whose n is not a multiple of 16
  WgradJob* jobs; int n_jobs, split;
  std::vector<WgradJob> h_jobs;          // host copy (wgrad_sk.hip takes the table in its launch arguments)
  OptState* opt; int* flag;      // opt: the CURRENT optimizer scalars (opt_alt: the buffer the one-launch reduction + optimizer writes next)
  OptState* opt_alt;
  const float* checked_grads = nullptr;   // gradient whose finite check already sits in flag[0] (single-rank mfm_fm_loss_grad)
  float *gmm_mode, *gmm_std, *gmm_logw, *counts, *Kinv, *kbias, *kdiag;
  OdeWs ode;
  Replay replay;               // armed by mfm_debug_replay for the NEXT mfm_ode_transform / mfm_flow_step (dt == nullptr: off)
  int64_t ctr[8];              // mfm_get_counters: host-side tallies (slot 4, the attempted Dopri5 steps, is summed on the device)
  unsigned long long* d_att;   // device tally of attempted steps
  int* att_buf; size_t att_cap;   // per-sample attempt counts of the last solve when the caller passed no d_nsteps
  double* beta_out;
  wide::Ctx* wide;             // non-null: the wide kernel family serves the network kernels (wide.hip)
  wide::Ctx* wide_ex = nullptr;   // fused family, exact trace, d >= 16: the SOLVES run on the wide family's solver (see create_impl)
  // context-owned RCCL communicator for the gradient all-reduce (mfm_comm_init; null: the caller reduces the gradient itself)
  ncclComm_t comm = nullptr; int comm_nranks = 0;
  hipStream_t comm_stream = nullptr; hipEvent_t ev_grads = nullptr, ev_comm = nullptr;
  const float* comm_pending = nullptr;   // gradient buffer whose all-reduce is in flight on comm_stream
  Switches sw;                           // the A/B switches as read at mfm_create: installed by every entry point (use_ctx)
  FmMala fuse_mala = {};                 // on != 0 during a mfm_train_iter whose MALA step rides in the training kernel
  int opt_resident_wgs = 0;              // workgroups of reduce_adamw_kernel this device holds at once (occupancy query at create)
};

// The development switches (common.hip.h) are read once per mfm_create and belong to THAT context: every entry point installs its
// context's snapshot before it sizes or launches anything, so a second context created under another environment cannot change what an
// earlier one launches (its workspaces were sized from its own snapshot: e.g. the flow step's chains per workgroup, MFM_FLOW_LIVE).
static inline void use_ctx(const mfm_ctx* x) { if (x) g_sw = x->sw; }

struct ProfScope {
  mfm_ctx* x; bool active;
  ProfScope(mfm_ctx* x_, int cls);
  ~ProfScope();
};

extern "C" const char* mfm_last_error(void) { return g_err; }
extern "C" int mfm_version(void) { return 1; }
extern "C" int mfm_pack_index(int k, int n, int KB) { return pack_index(k, n, KB); }
extern "C" int mfm_pack_index_T(int k, int n, int NB) { return pack_index_T(k, n, NB); }
extern "C" int mfm_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]) {
  threefry2x32(Key2{k0, k1}, c0, c1, out[0], out[1]);
  return 0;
}

ProfScope::ProfScope(mfm_ctx* x_, int cls) : x(x_), active(false) {
  Prof* p = x->prof;
  if (!p || !p->on || !(p->mask >> cls & 1u) || p->used + 2 > p->ev.size()) return;
  active = true;
  p->cls.push_back(cls);
  (void)hipEventRecord(p->ev[p->used], x->stream);
}
ProfScope::~ProfScope() {
  if (!active) return;
  Prof* p = x->prof;
  (void)hipEventRecord(p->ev[p->used + 1], x->stream);
  p->used += 2;
}

extern "C" int mfm_profile(mfm_ctx* x, int enable) { use_ctx(x);
  if (!x) return fail(MFM_EINVAL, "null ctx");
  if (!x->prof) x->prof = new Prof();
  Prof* p = x->prof;
  if (enable && p->ev.empty()) {
    p->ev.resize(2 * 16384);
    for (auto& e : p->ev) HIPCHK(hipEventCreate(&e));
  }
  p->on = enable != 0;
  p->mask = (unsigned)enable;
  if (enable) { p->used = 0; p->cls.clear(); }
  return MFM_OK;
}

extern "C" int mfm_profile_read(mfm_ctx* x, double ms[8], int64_t counts[8]) { use_ctx(x);
  if (!x || !ms || !counts) return fail(MFM_EINVAL, "null argument");
  for (int i = 0; i < 8; ++i) { ms[i] = 0.0; counts[i] = 0; }
  if (!x->prof) return MFM_OK;
  HIPCHK(hipStreamSynchronize(x->stream));
  Prof* p = x->prof;
  for (size_t i = 0; i < p->cls.size(); ++i) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, p->ev[2 * i], p->ev[2 * i + 1]));
    ms[p->cls[i]] += t; counts[p->cls[i]] += 1;
  }
  return MFM_OK;
}

// hidden widths of one branch as a list: depth 0 = the two-element array of the configuration (include/mfm.h)
static int branch_widths(int depth, const int32_t two[2], int32_t third, int out[MLP_MAX_DEPTH]) {
  const int n = depth == 0 ? 2 : depth;
  for (int i = 0; i < MLP_MAX_DEPTH; ++i) out[i] = 0;
  if (n < 1 || n > MLP_MAX_DEPTH) return -1;
  for (int i = 0; i < n && i < 2; ++i) out[i] = two[i];
  if (n > 2) out[2] = third;
  return n;
}

static void build_net(const mfm_config& c, NetDev& n) {
  memset(&n, 0, sizeof n);
  n.d = c.dim; n.dp = ceil16(c.dim); n.F = c.fourier_dim; n.F2p = ceil16(2 * c.fourier_dim);
  int ht[MLP_MAX_DEPTH], hx[MLP_MAX_DEPTH], hj[MLP_MAX_DEPTH];
  n.nT = branch_widths(c.depth_t, c.hidden_t, c.hidden_t3, ht);
  n.nX = branch_widths(c.depth_x, c.hidden_x, c.hidden_x3, hx);
  n.nJ = branch_widths(c.depth_xt, c.hidden_xt, c.hidden_xt3, hj);
  // buffer widths: padded to 16 (a hidden unit of the padding has zero weights and bias on both sides: act(0) = 0 for all five
  // activations, and its tangent / gradient meet zero weights -- exact)
  n.ht1 = ceil16(ht[0]); n.ht2 = ceil16(ht[n.nT - 1]); n.hx1 = ceil16(hx[0]); n.hx2 = ceil16(hx[n.nX - 1]); n.hj1 = ceil16(hj[0]); n.hj2 = ceil16(hj[n.nJ - 1]);
  // flax creation order (exe_flow_matching.py:74-86): time branch, x branch, gate, joint branch, output
  int K[MLP_MAXL], N[MLP_MAXL], nl = 0, prev = 2 * c.fourier_dim;
  for (int i = 0; i < n.nT; ++i) { K[nl] = prev; N[nl] = ht[i]; prev = ht[i]; ++nl; }
  prev = c.dim;
  for (int i = 0; i < n.nX; ++i) { K[nl] = prev; N[nl] = hx[i]; prev = hx[i]; ++nl; }
  K[nl] = ht[n.nT - 1]; N[nl] = c.dim; ++nl;
  prev = hx[n.nX - 1] + ht[n.nT - 1];
  const int joint0 = nl;
  for (int i = 0; i < n.nJ; ++i) { K[nl] = prev; N[nl] = hj[i]; prev = hj[i]; ++nl; }
  K[nl] = hj[n.nJ - 1]; N[nl] = c.dim; ++nl;
  int wo = 0, bo = 0, mo = 0;
  for (int l = 0; l < nl; ++l) {
    LayerDesc& L = n.L[l];
    L.K = K[l]; L.N = N[l]; L.Kp = l == joint0 ? n.hx2 + n.ht2 : ceil16(K[l]); L.Np = ceil16(N[l]);      // [sx | st]: both halves padded (mlp.hip.h: packed_row)
    L.w_off = wo; wo += L.Kp * L.Np;
    L.b_off = bo; bo += L.Np;
    L.m_w = mo; mo += K[l] * N[l];
    L.m_b = mo; mo += N[l];
  }
  for (int l = nl; l < MLP_MAXL; ++l) { n.L[l].m_w = n.L[l].m_b = mo; n.L[l].w_off = wo; n.L[l].b_off = bo; }      // never selected by a parameter index
  n.n_packed = wo; n.n_bias = bo; n.n_params = mo;
  n.grad_clip = c.grad_clip;
  n.act = c.activation;
}

static void build_ws(const NetDev& n, WsLayout& w) {
  int o = 0;
  auto take = [&](int feat) { int r = o; o += feat / 16; return r; };
  w.a_ffat = take(n.F2p); w.a_t1 = take(n.ht1); w.a_st = take(n.ht2); w.a_cond = take(n.dp);
  w.a_x1 = take(n.hx1); w.a_sx = take(n.hx2); w.a_j1 = take(n.hj1); w.a_j2 = take(n.hj2); w.a_tiles = o;
  o = 0;
  w.z_t1 = take(n.ht1); w.z_t2 = take(n.ht2); w.z_x1 = take(n.hx1); w.z_x2 = take(n.hx2);
  w.z_gate = take(n.dp); w.z_j1 = take(n.hj1); w.z_j2 = take(n.hj2); w.z_out = take(n.dp); w.z_tiles = o;
}

// Everything mfm_create allocates hangs off the context, so ONE exit path frees it: a failure below returns its status and
// mfm_create destroys the partially built context (mfm_destroy tolerates null members) -- an out-of-memory create can be
// retried in-process without leaking what was allocated before the failure.
static int create_impl(const mfm_config& c, mfm_ctx* x);

extern "C" int mfm_create(const mfm_config* cfg, mfm_ctx** out) {
  if (!cfg || !out) return fail(MFM_EINVAL, "null argument");
  *out = nullptr;
  mfm_ctx* x = new mfm_ctx();          // value-initialised: every pointer null, every counter zero
  const int rc = create_impl(*cfg, x);
  if (rc != MFM_OK) { (void)mfm_destroy(x); return rc; }
  *out = x;
  return MFM_OK;
}

static int create_impl(const mfm_config& c, mfm_ctx* x) {
  if (c.dim <= 0 || c.fourier_dim <= 0) return fail(MFM_EINVAL, "dim / fourier_dim must be positive");
  {
    int hs[3][MLP_MAX_DEPTH];
    const int nd[3] = {branch_widths(c.depth_t, c.hidden_t, c.hidden_t3, hs[0]), branch_widths(c.depth_x, c.hidden_x, c.hidden_x3, hs[1]),
                       branch_widths(c.depth_xt, c.hidden_xt, c.hidden_xt3, hs[2])};
    for (int b = 0; b < 3; ++b) {
      if (nd[b] < 0) return fail(MFM_EUNSUPPORTED, "a branch has 1 to %d hidden layers (depth_t / depth_x / depth_xt = %d / %d / %d)", MLP_MAX_DEPTH, c.depth_t, c.depth_x, c.depth_xt);
      for (int i = 0; i < nd[b]; ++i) {
        if (hs[b][i] <= 0) return fail(MFM_EINVAL, "hidden widths must be positive (got %d)", hs[b][i]);      // (not multiples of 16: zero-padded, mlp.hip.h)
      }
    }
  }
  if (c.n_chain_local <= 0 || c.n_chain_local % 16)
    return fail(MFM_EUNSUPPORTED, "n_chain_local must be a positive multiple of 16 (got %d)", c.n_chain_local);
  const int n_valid = c.n_chain_valid > 0 ? c.n_chain_valid : c.n_chain_local;
  if (n_valid > c.n_chain_local) return fail(MFM_EINVAL, "n_chain_valid=%d exceeds n_chain_local=%d", n_valid, c.n_chain_local);
  if (c.chain_offset < 0 || c.chain_offset + n_valid > c.n_chain_total)
    return fail(MFM_EINVAL, "chain shard [%d, %d) outside n_chain_total=%d", c.chain_offset,
                c.chain_offset + n_valid, c.n_chain_total);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MFM_EHIP, "no HIP device available");
  switches_read(); x->sw = g_sw;                     // development / A-B switches: fixed from here until the next mfm_create (common.hip.h)
  {
    // reduce_adamw_kernel's grid-wide exchange (taken when a gradient partial is huge or non-finite) spins until every workgroup
    // of its grid has arrived: all of them must be RESIDENT at once.  Ask the runtime what this device (a full MI355X, a CPX
    // partition, a CU-masked queue ...) and this build of the kernel actually hold, instead of assuming 256 CUs x 8 blocks.
    int dev = 0, per_cu = 0, cus = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reduce_adamw_kernel, 256, 0));
    x->opt_resident_wgs = per_cu * cus;
  }
  x->cfg = c;
  build_net(c, x->net);
  build_ws(x->net, x->ws);
  NetDev& n = x->net;
  if (c.activation < MFM_ACT_RELU || c.activation > MFM_ACT_SWISH) return fail(MFM_EINVAL, "unknown activation %d", c.activation);
  if (x->cfg.ref_std == 0.0) x->cfg.ref_std = 1.0;           // zero-initialised config: the default 'stdgauss'
  if (!(x->cfg.ref_std > 0.0)) return fail(MFM_EINVAL, "ref_std must be positive");
  bool use_wide = c.kernel_family == MFM_FAMILY_WIDE;
  const bool two_layer = n.nT == 2 && n.nX == 2 && n.nJ == 2;
  if (!two_layer) {             // the fused tile kernels are written layer by layer for two hidden layers per branch
    if (c.kernel_family == MFM_FAMILY_TILE)
      return fail(MFM_EUNSUPPORTED, "hidden lists of %d / %d / %d layers (t / x / xt) run on the wide kernel family only", n.nT, n.nX, n.nJ);
    use_wide = true;
  } else {
    const FmLds L = fm_lds_layout(n, true);
    size_t sm_ode; int tpw_ode;
    const bool fits = (size_t)L.total * 4 <= 160 * 1024 && (n.dp / 16 + MLP_WAVES_FM - 1) / MLP_WAVES_FM <= 2 && ode_check(n, sm_ode, tpw_ode) == 0;
    if (!fits && c.kernel_family == MFM_FAMILY_TILE)
      return fail(MFM_ETOOLARGE, "network does not fit the fused 16-chain tile kernel (LDS %zu B, dim %d)", (size_t)L.total * 4, c.dim);
    if (!fits) use_wide = true;
  }
  if (c.kernel_family < 0 || c.kernel_family > MFM_FAMILY_WIDE) return fail(MFM_EINVAL, "unknown kernel_family %d", c.kernel_family);
  if (c.ode_method < MFM_ODE_DOPRI5 || c.ode_method > MFM_ODE_EULER) return fail(MFM_EINVAL, "unknown ode_method %d", c.ode_method);
  if (c.ode_method != MFM_ODE_DOPRI5 && (c.ode_steps < 1 || (c.n_ts > 2 && c.ode_steps % (c.n_ts - 1)))) return fail(MFM_EINVAL, "ode_steps=%d: a positive number of steps (a multiple of n_ts - 1) is needed with a fixed-step ode_method", c.ode_steps);
  const int nbb = c.n_chain_local / 16;
  x->split = nbb < 8 ? nbb : 8;
  if (const char* e = getenv("MFM_WGRAD_SPLIT")) { const int v = atoi(e); if (v >= 1 && v <= nbb) x->split = v; }      // development: A/B
  // chain slices of the weight-gradient GEMM: 8 measured best at 4096 x 256 (wgrad + slab reduction 36.9 us against 41.4 at 16)
  x->loss_cap = (c.max_eval_samples > c.n_chain_local ? c.max_eval_samples : c.n_chain_local) / 16 + 1;
  // wgrad job table
  std::vector<WgradJob> jobs;
  for (int l = 0; l < 8; ++l)
    for (int nt = 0; nt < n.L[l].Np / 16; nt += 4)
      for (int kt = 0; kt < n.L[l].Kp / 16; kt += 4) jobs.push_back(WgradJob{l, kt, nt});
  x->n_jobs = (int)jobs.size(); x->h_jobs = jobs;
#define ALLOC(p, cnt) HIPCHK(hipMalloc((void**)&(p), (size_t)(cnt) * sizeof(*(p))))
  ALLOC(x->master, n.n_params); ALLOC(x->mu, n.n_params); ALLOC(x->nu, n.n_params);
  ALLOC(x->Wp, n.n_packed); ALLOC(x->WpT, n.n_packed); ALLOC(x->bias, n.n_bias); ALLOC(x->fourier, n.F);
  if (!use_wide) {       // packed activation / gradient workspaces of the fused family
    ALLOC(x->acts, (size_t)x->ws.a_tiles * nbb * 256); ALLOC(x->dzs, (size_t)x->ws.z_tiles * nbb * 256);
    if (c.activation >= MFM_ACT_GELU) ALLOC(x->dacts, (size_t)x->ws.a_tiles * nbb * 256);      // gelu / swish: f'(pre-activation), see FmArgs
    ALLOC(x->slabs, (size_t)x->split * n.n_params);
  }
  ALLOC(x->loss_part, x->loss_cap);
  ALLOC(x->eval_pad, (size_t)16 * c.dim);
  if (!use_wide && x->n_jobs <= WSK_MAXJOBS && !getenv("MFM_WGRAD_SLABS")) {      // (MFM_WGRAD_SLABS=1: the round-1..4 slab kernels, for A/B)
    // stream-K weight gradients + optimizer in one launch (wgrad_sk.hip).  Its workgroups WAIT for the other contributors of their
    // block, so every workgroup of the grid must be resident at once: the grid never exceeds what the runtime says this device (a
    // full MI355X, a partition ...) holds of this kernel; a device that cannot hold one workgroup per block keeps the slab kernels.
    int dev = 0, per_cu = 0, cus = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_sk_kernel, 256, 0));
    WskConst h; memset(&h, 0, sizeof h);
    int G = 0;
    if (per_cu * cus >= x->n_jobs) wsk_plan(x->n_jobs, nbb, per_cu * cus, G, h.upw_q, h.upw_r);
    if (G >= x->n_jobs && h.upw_q >= 1) {
      h.net = n; h.nbb = nbb; h.n_jobs = x->n_jobs; h.G = G;
      ALLOC(x->wsk_partials, (size_t)2 * G * WSK_PSZ); ALLOC(x->wsk_tickets, 2 * x->n_jobs); ALLOC(x->wsk_const, 1);
      HIPCHK(hipMemset(x->wsk_tickets, 0, (size_t)2 * x->n_jobs * sizeof(int)));
      h.acts = x->acts; h.dzs = x->dzs; h.partials = x->wsk_partials;
      h.master = x->master; h.mu = x->mu; h.nu = x->nu; h.Wp = x->Wp; h.WpT = x->WpT; h.bias = x->bias;
      h.lr0 = c.learning_rate; h.learning_iter = c.learning_iter; h.warmup = c.warmup_steps;
      h.b1 = c.adam_b1; h.b2 = c.adam_b2; h.eps = (float)c.adam_eps; h.wd = (float)c.weight_decay; h.clip = (float)c.update_clip; h.max_err = 10;
      for (int j = 0; j < x->n_jobs; ++j) {
        const WgradJob& J = jobs[j];
        WskJob& o = h.jobs[j];
        o.layer = J.layer; o.kt0 = J.kt0; o.nt0 = J.nt0;
        const int KT = n.L[J.layer].Kp / 16, NT = n.L[J.layer].Np / 16;
        for (int wv = 0; wv < 4; ++wv) {
          o.a_row[wv] = wgrad_a_tile(n, x->ws, J.layer, J.kt0 + wv < KT ? J.kt0 + wv : J.kt0);
          o.z_row[wv] = wgrad_z_tile(x->ws, J.layer, J.nt0 + wv < NT ? J.nt0 + wv : J.nt0);
        }
        h.n_slices += wsk_wg_of(h.upw_q, h.upw_r, (j + 1) * nbb - 1) - wsk_wg_of(h.upw_q, h.upw_r, j * nbb) + 1;
      }
      HIPCHK(hipMemcpy(x->wsk_const, &h, sizeof h, hipMemcpyHostToDevice));
      std::vector<WskWg> wg(G);
      for (int wq = 0; wq < G; ++wq) {
        WskWg& d = wg[wq];
        const int u0 = wsk_start(h.upw_q, h.upw_r, wq);
        d.cnt = h.upw_q + (wq < h.upw_r ? 1 : 0); d.j0 = u0 / nbb; d.bb0 = u0 - d.j0 * nbb; d.n0 = d.cnt < nbb - d.bb0 ? d.cnt : nbb - d.bb0;
        for (int sg = 0; sg < 2; ++sg) {
          const WskJob& J = h.jobs[d.j0 + sg < x->n_jobs ? d.j0 + sg : d.j0];
          for (int wv = 0; wv < 4; ++wv) { d.a_row[sg][wv] = J.a_row[wv]; d.z_row[sg][wv] = J.z_row[wv]; }
        }
      }
      ALLOC(x->wsk_wg, G);
      HIPCHK(hipMemcpy(x->wsk_wg, wg.data(), sizeof(WskWg) * G, hipMemcpyHostToDevice));
      x->wsk_G = G;
    }
  }
  ALLOC(x->jobs, x->n_jobs); ALLOC(x->opt, 1); ALLOC(x->opt_alt, 1); ALLOC(x->flag, 8); ALLOC(x->beta_out, 4);
  ALLOC(x->d_att, 1); HIPCHK(hipMemset(x->d_att, 0, sizeof(unsigned long long)));
  HIPCHK(hipMemcpy(x->jobs, jobs.data(), jobs.size() * sizeof(WgradJob), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(x->master, 0, n.n_params * 4)); HIPCHK(hipMemset(x->mu, 0, n.n_params * 4));
  HIPCHK(hipMemset(x->nu, 0, n.n_params * 4)); HIPCHK(hipMemset(x->Wp, 0, n.n_packed * 4));
  HIPCHK(hipMemset(x->WpT, 0, n.n_packed * 4)); HIPCHK(hipMemset(x->bias, 0, n.n_bias * 4));
  HIPCHK(hipMemset(x->opt, 0, sizeof(OptState))); HIPCHK(hipMemset(x->opt_alt, 0, sizeof(OptState))); HIPCHK(hipMemset(x->flag, 0, 32));
  n.Wp = x->Wp; n.WpT = x->WpT; n.bias = x->bias; n.fourier = x->fourier;
  int rc = ode_ws_alloc(n, c, x->ode);
  if (rc) return fail(rc, "ODE workspace allocation failed");
  if (use_wide) {
    rc = wide::create(n, c.n_chain_local, &x->wide);
    if (rc) return fail(rc, "workspace allocation of the wide kernel family failed");
    x->wide->exact = !c.hutch;           // exact-trace log-det (exe_flow_matching.py:216-217,236-237): wide.hip, exact_trace()
    x->wide->master = x->master;         // ... which reads two kernels in their canonical [in][out] layout
  } else if (!c.hutch && c.dim >= 16 && !g_sw.tile_exact) {
    // Exact trace (the reference's default: no --hutch) on the fused family: its generic tile pushes the d basis tangents of 16 chains
    // through the network one 16-row tile at a time -- 7.4 ms per attempted step at d = 64, whatever the chain count.  The wide family's
    // exact trace needs hx1 tangent rows per chain in a few large GEMMs over all chains (wide.hip: exact_trace): 1.2 ms per attempt at
    // 1024 chains, 0.4 ms at 64.  So the SOLVES (flow steps, transforms) of such a context go to a wide workspace kept beside the tile
    // kernels, which keep MALA, training and evaluation.  Both read the same packed weights.  MFM_TILE_EXACT=1 keeps the tile's own.
    rc = wide::create(n, c.n_chain_local, &x->wide_ex);
    if (rc) return fail(rc, "workspace allocation of the exact-trace solver failed");
    x->wide_ex->exact = true;
    x->wide_ex->master = x->master;
  }
  return MFM_OK;
}

extern "C" int mfm_destroy(mfm_ctx* x) { use_ctx(x);
  if (!x) return MFM_OK;
  hipDeviceSynchronize();
  void* ps[] = {x->master, x->mu, x->nu, x->Wp, x->WpT, x->bias, x->fourier, x->acts, x->dzs, x->dacts, x->slabs, x->loss_part, x->eval_pad, x->wsk_partials, x->wsk_tickets, x->wsk_const, x->wsk_wg,
                x->jobs, x->opt, x->opt_alt, x->flag, x->gmm_mode, x->gmm_std, x->gmm_logw, x->counts, x->Kinv, x->kbias, x->kdiag, x->beta_out,
                x->d_att, x->att_buf};
  for (void* p : ps) if (p) hipFree(p);
  (void)mfm_comm_destroy(x);
  ode_ws_free(x->ode);
  wide::destroy(x->wide);
  wide::destroy(x->wide_ex);
  if (x->noise) {
    NoiseWs* w = x->noise;
    for (void* p : {(void*)w->mala_n, (void*)w->mala_u, (void*)w->fm_x0, (void*)w->fm_eps, (void*)w->fm_t, (void*)w->d_keys, (void*)w->counter}) if (p) (void)hipFree(p);
    delete w;
  }
  if (x->prof) { for (auto& e : x->prof->ev) (void)hipEventDestroy(e); delete x->prof; }
  delete x;
  return MFM_OK;
}

extern "C" int mfm_set_stream(mfm_ctx* x, void* s) { use_ctx(x); if (!x) return fail(MFM_EINVAL, "null ctx"); x->stream = (hipStream_t)s; return MFM_OK; }
extern "C" int mfm_sync(mfm_ctx* x) { use_ctx(x); if (!x) return fail(MFM_EINVAL, "null ctx"); HIPCHK(hipStreamSynchronize(x->stream)); return MFM_OK; }
extern "C" int mfm_num_params(const mfm_ctx* x) { use_ctx(x); return x ? x->net.n_params : MFM_EINVAL; }

extern "C" int mfm_set_target(mfm_ctx* x, int kind, const double* p, size_t np) { use_ctx(x);
  if (!x || !p) return fail(MFM_EINVAL, "null argument");
  TargetDev& T = x->net.T;
  const int d = x->cfg.dim;
  memset(&T, 0, sizeof T);
  T.kind = kind; T.dim = d;
  if (kind == MFM_PHI4) {
    if (np != 2) return fail(MFM_EINVAL, "phi4 target takes {a, beta}");
    T.coef = (float)(p[0] * d); T.tbeta = (float)p[1];
  } else if (kind == MFM_GMM) {
    const int K = (int)p[0];
    if (K <= 0 || K > MFM_GMM_MAX_MODES || np != (size_t)(1 + 2 * K * d + K)) return fail(MFM_EINVAL, "bad GMM parameter block");
    if (d > 8) return fail(MFM_EUNSUPPORTED, "GMM targets support dim <= 8 (the reference forces dim = 2)");
    std::vector<float> mode(K * d), sd(K * d), lw(K);
    for (int i = 0; i < K * d; ++i) { mode[i] = (float)p[1 + i]; sd[i] = (float)p[1 + K * d + i]; }
    for (int k = 0; k < K; ++k) {
      double s = std::log(p[1 + 2 * K * d + k]) - 0.5 * d * std::log(2.0 * M_PI);
      for (int j = 0; j < d; ++j) s -= std::log(p[1 + K * d + k * d + j]);
      lw[k] = (float)s;
    }
    for (float** q : {&x->gmm_mode, &x->gmm_std, &x->gmm_logw}) if (*q) { hipFree(*q); *q = nullptr; }
    ALLOC(x->gmm_mode, K * d); ALLOC(x->gmm_std, K * d); ALLOC(x->gmm_logw, K);
    HIPCHK(hipMemcpy(x->gmm_mode, mode.data(), K * d * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(x->gmm_std, sd.data(), K * d * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(x->gmm_logw, lw.data(), K * 4, hipMemcpyHostToDevice));
    T.n_modes = K; T.gmm_mode = x->gmm_mode; T.gmm_std = x->gmm_std; T.gmm_logw = x->gmm_logw;
  } else if (kind == MFM_LGCP) {
    // {mu, poisson_a, log_norm, counts[d], Kinv[d*d]} (distributions.py:249-274; Kinv = inverse of the Gram matrix :265)
    if (np != (size_t)(3 + d + (size_t)d * d)) return fail(MFM_EINVAL, "bad LGCP parameter block");
    const int dp = x->net.dp, KB = dp / 16;
    std::vector<float> cnt(dp, 0.f), kb(dp, 0.f), kp((size_t)dp * dp, 0.f);
    const double mu = p[0];
    const double* Kinv = p + 3 + d;
    for (int j = 0; j < d; ++j) cnt[j] = (float)p[3 + j];
    for (int k = 0; k < d; ++k)
      for (int j = 0; j < d; ++j) kp[pack_index(k, j, KB)] = (float)Kinv[(size_t)k * d + j];
    for (int j = 0; j < d; ++j) {
      double rs = 0.0;
      for (int k = 0; k < d; ++k) rs += Kinv[(size_t)k * d + j];
      kb[j] = (float)(-mu * rs);
    }
    for (float** q : {&x->counts, &x->Kinv, &x->kbias, &x->kdiag}) if (*q) { hipFree(*q); *q = nullptr; }
    ALLOC(x->counts, dp); ALLOC(x->kbias, dp); ALLOC(x->kdiag, dp); ALLOC(x->Kinv, (size_t)dp * dp);
    {
      std::vector<float> kd(dp, 0.f);
      for (int j = 0; j < d; ++j) kd[j] = (float)Kinv[(size_t)j * d + j];
      HIPCHK(hipMemcpy(x->kdiag, kd.data(), dp * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(x->counts, cnt.data(), dp * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(x->kbias, kb.data(), dp * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(x->Kinv, kp.data(), (size_t)dp * dp * 4, hipMemcpyHostToDevice));
    T.counts = x->counts; T.KinvP = x->Kinv; T.kbias = x->kbias; T.kdiag = x->kdiag;
    T.mu = (float)mu; T.poisson_a = (float)p[1]; T.log_norm = (float)p[2];
    // the K^-1 tile buffers must fit next to the MLP tiles
    const FmLds Lf = fm_lds_layout(x->net, true);
    const OdeLds Lo = ode_lds_layout(x->net, ODE_NW);
    if (!x->wide && ((size_t)Lf.total * 4 > 160 * 1024 || (size_t)Lo.total * 4 > 160 * 1024)) {
      memset(&T, 0, sizeof T);
      return fail(MFM_ETOOLARGE, "LGCP target with dim %d does not fit the 16-chain LDS tile next to this network", d);
    }
  } else {
    return fail(MFM_EINVAL, "unknown target kind %d", kind);
  }
  x->has_target = true;
  return MFM_OK;
}

extern "C" int mfm_set_fourier(mfm_ctx* x, const float* h) { use_ctx(x);
  if (!x || !h) return fail(MFM_EINVAL, "null argument");
  HIPCHK(hipMemcpy(x->fourier, h, x->net.F * 4, hipMemcpyHostToDevice));
  x->has_fourier = true;
  return MFM_OK;
}
extern "C" int mfm_set_params(mfm_ctx* x, const float* h) { use_ctx(x);
  if (!x || !h) return fail(MFM_EINVAL, "null argument");
  HIPCHK(hipMemcpyAsync(x->master, h, (size_t)x->net.n_params * 4, hipMemcpyHostToDevice, x->stream));
  launch_pack(x->net, x->master, x->Wp, x->WpT, x->bias, x->stream);
  LAUNCHCHK();
  HIPCHK(hipStreamSynchronize(x->stream));
  return MFM_OK;
}
extern "C" int mfm_get_params(mfm_ctx* x, float* h) { use_ctx(x);
  if (!x || !h) return fail(MFM_EINVAL, "null argument");
  HIPCHK(hipStreamSynchronize(x->stream));
  HIPCHK(hipMemcpy(h, x->master, (size_t)x->net.n_params * 4, hipMemcpyDeviceToHost));
  return MFM_OK;
}
extern "C" int mfm_reset_optimizer(mfm_ctx* x) { use_ctx(x);
  if (!x) return fail(MFM_EINVAL, "null ctx");
  HIPCHK(hipMemsetAsync(x->mu, 0, (size_t)x->net.n_params * 4, x->stream));
  HIPCHK(hipMemsetAsync(x->nu, 0, (size_t)x->net.n_params * 4, x->stream));
  HIPCHK(hipMemsetAsync(x->opt, 0, sizeof(OptState), x->stream));
  return MFM_OK;
}

static int noise_take(mfm_ctx* x, Key2 key, bool step);
#define NEED_TARGET() do { if (!x) return fail(MFM_EINVAL, "null ctx"); if (!x->has_target) return fail(MFM_ENOTARGET, "mfm_set_target has not been called"); } while (0)

static MalaArgs mala_args(mfm_ctx* x, double beta) {
  MalaArgs a; memset(&a, 0, sizeof a);
  a.T = x->net.T; a.n_total = x->cfg.n_chain_total; a.chain_offset = x->cfg.chain_offset; a.B = x->cfg.n_chain_local;
  a.beta = beta;
  return a;
}

// LGCP dimensions beyond the fused tile kernel (d > 1024) go through the wide family's propose / K^-1 GEMM / accept split
static int wide_mala_lgcp(mfm_ctx* x, const LgcpArgs& l) {
  if (!x->wide) return -3;
  wide::LgcpMala m; memset(&m, 0, sizeof m);
  m.T = l.T; m.mode = l.mode; m.key = l.key; m.keys = l.keys; m.n_total = l.n_total; m.chain_offset = l.chain_offset; m.rows = l.B; m.d = l.T.dim; m.dp = l.dp;
  m.beta = l.beta; m.eps = l.eps; m.textbook = l.textbook;
  m.pos = l.pos; m.logp = l.logp; m.grad = l.grad; m.acc_prob = l.acc_prob; m.accepted = l.accepted; m.proposed = l.proposed; m.prop_weight = l.prop_weight;
  return wide::mala_lgcp(x->wide, x->net, m, x->stream);
}

// The fused LGCP tile kernel runs one workgroup per 16 chains and streams K^-1 once per workgroup: right for few chains,
// but 1024 chains are only 64 workgroups on 256 CUs.  With the wide family present and >= 128 chains the K^-1 contraction
// goes through the wide GEMM (every CU busy) between a propose and an accept kernel; it also serves d > 1024.
static int lgcp_mala_dispatch(mfm_ctx* x, const LgcpArgs& l) {
  if (x->wide && l.B >= 128 && wide_mala_lgcp(x, l) == 0) return 0;
  if (launch_mala_lgcp(l, x->stream) == 0) return 0;
  return wide_mala_lgcp(x, l);
}

extern "C" int mfm_mala_init(mfm_ctx* x, const float* d_pos, double beta, double* d_logp, float* d_grad) { use_ctx(x);
  NEED_TARGET();
  if (!d_pos || !d_logp || !d_grad) return fail(MFM_EINVAL, "null device pointer");
  if (x->net.T.kind == MFM_TARGET_LGCP) {
    LgcpArgs l; memset(&l, 0, sizeof l);
    l.T = x->net.T; l.dp = x->net.dp; l.mode = 0; l.n_total = x->cfg.n_chain_total; l.chain_offset = x->cfg.chain_offset;
    l.B = x->cfg.n_chain_local; l.beta = beta; l.eps = 1.0; l.pos = const_cast<float*>(d_pos); l.logp = d_logp; l.grad = d_grad;
    if (lgcp_mala_dispatch(x, l)) return fail(MFM_ETOOLARGE, "dim %d too large for the LGCP MALA kernel", x->cfg.dim);
    LAUNCHCHK();
    return MFM_OK;
  }
  MalaArgs a = mala_args(x, beta);
  a.pos = const_cast<float*>(d_pos); a.logp = d_logp; a.grad = d_grad;
  if (launch_mala_init(a, x->stream)) return fail(MFM_ETOOLARGE, "dim %d too large for the MALA kernel", x->cfg.dim);
  LAUNCHCHK();
  return MFM_OK;
}

static int mala_step_common(mfm_ctx* x, uint32_t k0, uint32_t k1, const uint32_t* d_keys, double beta, double step, int textbook, float* d_pos,
                            double* d_logp, float* d_grad, float* d_acc, uint8_t* d_isacc, float* d_prop, float* d_pw) {
  NEED_TARGET();
  if (!d_pos || !d_logp || !d_grad) return fail(MFM_EINVAL, "null device pointer");
  if (!(step > 0)) return fail(MFM_EINVAL, "step_size must be positive");
  if (x->net.T.kind == MFM_TARGET_LGCP) {
    LgcpArgs l; memset(&l, 0, sizeof l);
    l.T = x->net.T; l.dp = x->net.dp; l.mode = 1; l.key = Key2{k0, k1}; l.keys = d_keys; l.n_total = x->cfg.n_chain_total;
    l.chain_offset = x->cfg.chain_offset; l.B = x->cfg.n_chain_local; l.beta = beta; l.eps = step; l.textbook = textbook;
    l.pos = d_pos; l.logp = d_logp; l.grad = d_grad; l.acc_prob = d_acc; l.accepted = d_isacc; l.proposed = d_prop; l.prop_weight = d_pw;
    ProfScope ps_(x, PROF_MALA);
    if (lgcp_mala_dispatch(x, l)) return fail(MFM_ETOOLARGE, "dim %d too large for the LGCP MALA kernel", x->cfg.dim);
    LAUNCHCHK();
    x->ctr[CTR_MALA] += x->cfg.n_chain_local; x->ctr[CTR_MALA_BYTES] += (int64_t)x->cfg.n_chain_local * 4 * (5 * x->cfg.dim + 5);
    return MFM_OK;
  }
  MalaArgs a = mala_args(x, beta);
  a.key = Key2{k0, k1}; a.keys = d_keys; a.eps = step; a.textbook = textbook;
  if (!d_keys) {
    const int slot = noise_take(x, Key2{k0, k1}, false);
    if (slot >= 0) {
      const size_t B = (size_t)x->cfg.n_chain_local;
      a.pre_n = x->noise->mala_n + (size_t)slot * B * x->cfg.dim; a.pre_u = x->noise->mala_u + (size_t)slot * B;
    }
  }
  a.pos = d_pos; a.logp = d_logp; a.grad = d_grad;
  a.acc_prob = d_acc; a.accepted = d_isacc; a.proposed = d_prop; a.prop_weight = d_pw;
  ProfScope ps_(x, PROF_MALA);
  if (launch_mala_step(a, x->stream)) return fail(MFM_ETOOLARGE, "dim %d too large for the MALA kernel", x->cfg.dim);
  LAUNCHCHK();
  x->ctr[CTR_MALA] += x->cfg.n_chain_local; x->ctr[CTR_MALA_BYTES] += (int64_t)x->cfg.n_chain_local * 4 * (5 * x->cfg.dim + 5);
  return MFM_OK;
}

extern "C" int mfm_mala_step(mfm_ctx* x, uint32_t k0, uint32_t k1, double beta, double step, int textbook, float* d_pos,
                             double* d_logp, float* d_grad, float* d_acc, uint8_t* d_isacc, float* d_prop, float* d_pw) { use_ctx(x);
  return mala_step_common(x, k0, k1, nullptr, beta, step, textbook, d_pos, d_logp, d_grad, d_acc, d_isacc, d_prop, d_pw);
}
extern "C" int mfm_mala_step_keys(mfm_ctx* x, const uint32_t* d_keys, double beta, double step, int textbook, float* d_pos,
                                  double* d_logp, float* d_grad, float* d_acc, uint8_t* d_isacc, float* d_prop, float* d_pw) { use_ctx(x);
  if (!d_keys) return fail(MFM_EINVAL, "null key array");
  return mala_step_common(x, 0, 0, d_keys, beta, step, textbook, d_pos, d_logp, d_grad, d_acc, d_isacc, d_prop, d_pw);
}

// Build-side mode (hmc.hip; not on the reference's MFM path): one HMC step of every local chain, state updated in place like mfm_mala_step
extern "C" int mfm_hmc_step(mfm_ctx* x, uint32_t k0, uint32_t k1, double beta, double step, int num_steps, float* d_pos, double* d_logp,
                            float* d_grad, float* d_acc, uint8_t* d_isacc) { use_ctx(x);
  NEED_TARGET();
  if (!d_pos || !d_logp || !d_grad) return fail(MFM_EINVAL, "null device pointer");
  if (!(step > 0)) return fail(MFM_EINVAL, "step_size must be positive");
  if (num_steps < 1 || num_steps > 100000) return fail(MFM_EINVAL, "num_steps must be in [1, 100000] (got %d)", num_steps);
  if (x->net.T.kind == MFM_TARGET_LGCP) return fail(MFM_EUNSUPPORTED, "the HMC step serves the phi-four and mixture targets (the Cox process needs the K^-1 GEMM tile)");
  HmcArgs a; memset(&a, 0, sizeof a);
  a.T = x->net.T; a.key = Key2{k0, k1}; a.n_total = x->cfg.n_chain_total; a.chain_offset = x->cfg.chain_offset; a.B = x->cfg.n_chain_local;
  a.num_steps = num_steps; a.beta = beta; a.eps = step;
  a.pos = d_pos; a.logp = d_logp; a.grad = d_grad; a.acc_prob = d_acc; a.accepted = d_isacc;
  ProfScope ps_(x, PROF_MALA);
  if (launch_hmc_step(a, x->stream)) return fail(MFM_ETOOLARGE, "dim %d too large for the HMC kernel", x->cfg.dim);
  LAUNCHCHK();
  x->ctr[CTR_MALA] += x->cfg.n_chain_local;
  return MFM_OK;
}

extern "C" int mfm_loglik(mfm_ctx* x, const float* d_pos, double* d_out) { use_ctx(x);
  NEED_TARGET();
  if (!d_pos || !d_out) return fail(MFM_EINVAL, "null device pointer");
  MalaArgs a = mala_args(x, 1.0);
  a.pos = const_cast<float*>(d_pos);
  if (launch_loglik(a, d_out, x->stream)) return fail(MFM_ETOOLARGE, "dim %d too large", x->cfg.dim);
  LAUNCHCHK();
  return MFM_OK;
}

// slot of the prefetched draws for `key` (consumed in order), or -1; the first consumer after a prefetch makes the
// context's stream wait for the side stream
static int noise_take(mfm_ctx* x, Key2 key, bool step) {
  NoiseWs* w = x->noise;
  if (!w || w->n_valid == 0) return -1;
  int& cur = step ? w->cur_st : w->cur_gn;
  const std::vector<Key2>& ks = step ? w->st : w->gn;
  for (int j = cur; j < w->n_valid; ++j)
    if (ks[j].k0 == key.k0 && ks[j].k1 == key.k1 && !(!step && j == 0 && w->no_mala0)) { cur = j + 1; return j; }      // same stream as the producer: ordered
  return -1;
}

// eval_valid > 0 (forward only): rows >= eval_valid of the launch are padding (no loss); accumulate: add to *d_loss
static int fm_common(mfm_ctx* x, uint32_t k0, uint32_t k1, const float* d_samples, int n, int n_total, int offset, bool train,
                     double* d_loss, float* d_grads = nullptr, int eval_valid = 0, bool accumulate = false) {
  if (!x->has_fourier) return fail(MFM_EINVAL, "mfm_set_fourier has not been called");
  if (n <= 0 || n % 16) return fail(MFM_EUNSUPPORTED, "sample count must be a positive multiple of 16 (got %d)", n);
  if (!x->wide && n / 16 > x->loss_cap) return fail(MFM_ETOOLARGE, "n=%d exceeds max_eval_samples given at mfm_create", n);
  x->ctr[train ? CTR_FM_TRAIN : CTR_FM_EVAL] += n;
  FmArgs a; memset(&a, 0, sizeof a);
  a.net = x->net; a.ws = x->ws;
  const Key2 key{k0, k1};
  if (x->cfg.cond_flow) {                       // exe_flow_matching.py:153
    a.key_time = split_at(key, 4, 0); a.key_ref = split_at(key, 4, 1); a.key_gauss = split_at(key, 4, 2);
  } else {                                      // :141
    a.key_time = split_at(key, 2, 0); a.key_ref = split_at(key, 2, 1);
  }
  a.n_total = n_total; a.chain_offset = offset; a.B = n; a.sigma = x->cfg.sigma; a.cond_flow = x->cfg.cond_flow;
  a.ref_std = x->cfg.ref_std;
  a.n_valid = train && x->cfg.n_chain_valid > 0 ? x->cfg.n_chain_valid : n;      // (padding rows of the chain shard: no loss, no gradient)
  if (!train && eval_valid > 0) a.n_valid = eval_valid;
  a.pos = d_samples; a.acts = x->acts; a.dzs = x->dzs; a.dacts = x->dacts; a.loss_part = x->loss_part;
  if (x->wide) {      // R rows per pass; the loss is accumulated over the passes
    wide::Ctx* w = x->wide;
    for (int r0 = 0; r0 < n; r0 += w->R) {
      wide::FmCall c;
      c.key_time = a.key_time; c.key_ref = a.key_ref; c.key_gauss = a.key_gauss; c.n_total = (uint32_t)n_total; c.chain_offset = (uint32_t)(offset + r0);
      c.sigma = a.sigma; c.cond_flow = a.cond_flow; c.ref_std = a.ref_std; c.pos = d_samples + (size_t)r0 * x->cfg.dim; c.rows = n - r0 < w->R ? n - r0 : w->R;
      c.rows_valid = a.n_valid - r0 < 0 ? 0 : (a.n_valid - r0 < c.rows ? a.n_valid - r0 : c.rows);
      // one rank, one pass: this IS the gradient the optimizer will see, so its finite check rides in the weight-gradient kernel
      const bool inline_check = train && n <= w->R && x->cfg.n_chain_total == x->cfg.n_chain_local;
      c.bad = inline_check ? x->flag : nullptr;
      int rcw;
      { ProfScope ps_(x, train ? PROF_FM : PROF_EVAL); rcw = wide::fm(w, x->net, c, train, train ? d_grads : nullptr, x->stream); }
      if (rcw) return fail(rcw, "wide fm kernels cannot be launched for this configuration");
      LAUNCHCHK();
      ProfScope ps2_(x, PROF_REDUCE);
      launch_reduce_loss(w->loss_part, (c.rows + 3) / 4, d_loss, r0 > 0 || accumulate, x->stream);
      LAUNCHCHK();
      if (inline_check) x->checked_grads = d_grads;
    }
    return MFM_OK;
  }
  if (train && x->fuse_mala.on) a.mala = x->fuse_mala;      // mfm_train_iter: the iteration's MALA step in the same launch
  if (train) { a.flags_clear = x->flag; a.sus_set = x->flag + 5 + x->sus_par; a.sus_clear = x->flag + 5 + (x->sus_par ^ 1); }
  if (train && x->cfg.cond_flow) {
    const int slot = noise_take(x, key, true);
    if (slot >= 0) {
      const size_t B = (size_t)x->cfg.n_chain_local;
      a.pre_x0 = x->noise->fm_x0 + (size_t)slot * B * x->cfg.dim; a.pre_eps = x->noise->fm_eps + (size_t)slot * B * x->cfg.dim;
      a.pre_t = x->noise->fm_t + (size_t)slot * B;
    }
  }
  int rc;
  { ProfScope ps_(x, train ? PROF_FM : PROF_EVAL); rc = launch_fm(a, train, x->stream); }
  if (rc) return fail(rc, "fm kernel cannot be launched for this configuration");
  LAUNCHCHK();
  if (train) return MFM_OK;            // the training path totals the loss partials in its slab-reduction kernel
  ProfScope ps2_(x, PROF_REDUCE);
  launch_reduce_loss(x->loss_part, fm_eval_parts(x->net, n), d_loss, accumulate ? 1 : 0, x->stream);
  LAUNCHCHK();
  return MFM_OK;
}

static AdamArgs adam_args(mfm_ctx* x, const float* grads, int n_slabs) {
  const mfm_config& c = x->cfg;
  AdamArgs a; memset(&a, 0, sizeof a);
  a.net = x->net; a.grads = grads; a.n_slabs = n_slabs;
  a.master = x->master; a.mu = x->mu; a.nu = x->nu; a.Wp = x->Wp; a.WpT = x->WpT; a.bias = x->bias;
  a.st = x->opt; a.flag = x->flag;
  a.lr0 = c.learning_rate; a.learning_iter = c.learning_iter; a.warmup = c.warmup_steps;
  a.b1 = c.adam_b1; a.b2 = c.adam_b2; a.eps = (float)c.adam_eps; a.wd = (float)c.weight_decay; a.clip = (float)c.update_clip;
  a.max_err = 10;
  return a;
}

// one rank, tile family: slab reduction + apply_if_finite + AdamW as ONE launch behind the weight-gradient kernel (optim.hip:
// reduce_adamw_kernel; every workgroup of its grid must be resident at once, hence the bound on the parameter count)
static bool opt_fusable(mfm_ctx* x) {
  // the grid (one parameter per thread) may take at most HALF of the workgroup slots the occupancy query reported at create:
  // margin for other queues' kernels and for a CU mask narrower than the device (headline: 837 of 2048 on a full MI355X; a
  // 32-CU partition holds 256 and falls back to reduce_slabs + adamw)
  const int grid = (x->net.n_params + 255) / 256;
  if (x->wsk_G) return !g_sw.no_fused_opt && !x->wide && !x->comm && x->cfg.n_chain_total == x->cfg.n_chain_local;      // (wgrad_sk.hip: no workgroup waits on another)
  return !g_sw.no_fused_opt && !x->wide && !x->comm && x->cfg.n_chain_total == x->cfg.n_chain_local && 2 * grid <= x->opt_resident_wgs;
}

static int fm_loss_grad_impl(mfm_ctx* x, uint32_t k0, uint32_t k1, const float* d_pos, double* d_loss, float* d_grads, bool with_optimizer) {
  NEED_TARGET();
  if (!d_pos || !d_loss || !d_grads) return fail(MFM_EINVAL, "null device pointer");
  x->checked_grads = nullptr;
  int rc = fm_common(x, k0, k1, d_pos, x->cfg.n_chain_local, x->cfg.n_chain_total, x->cfg.chain_offset, true, d_loss, d_grads);
  if (rc || x->wide) return rc;
  const int sus_par = x->sus_par; x->sus_par ^= 1;      // (the training kernel just wrote flag[5 + sus_par] and cleared the other word)
  if (x->wsk_G) {
    // stream-K weight gradients; the last arriver of every block totals it and, with the optimizer, updates it (wgrad_sk.hip)
    WskArgs k; memset(&k, 0, sizeof k);
    k.C = x->wsk_const; k.wg = x->wsk_wg; k.acts = x->acts; k.dzs = x->dzs; k.nbb = x->cfg.n_chain_local / 16; k.G = x->wsk_G;
    k.xcd_remap = g_sw.wsk_xcd ? 1 : 0;
    k.tickets = x->wsk_tickets + x->wsk_par * x->n_jobs; k.tickets_clear = x->wsk_tickets + (x->wsk_par ^ 1) * x->n_jobs; x->wsk_par ^= 1;
    k.out = d_grads;
    k.loss_part = x->loss_part; k.n_part = x->cfg.n_chain_local / 16; k.loss_out = d_loss;
    const bool single1 = x->cfg.n_chain_total == x->cfg.n_chain_local;
    if (with_optimizer) {
      k.fuse = 1;
      k.st = x->opt; k.st_next = x->opt_alt; k.flag = x->flag; k.suspicious = x->flag + 5 + sus_par;
      k.force_exchange = g_sw.force_exchange ? 1 : 0;
    } else {
      k.bad = single1 ? x->flag : nullptr;      // flag[0] was cleared by the training kernel (FmArgs::flags_clear)
    }
    { ProfScope ps_(x, PROF_WGRAD); launch_wgrad_sk(k, x->wsk_G, x->stream); }
    LAUNCHCHK();
    if (with_optimizer) {
      OptState* t = x->opt; x->opt = x->opt_alt; x->opt_alt = t;
      x->ctr[CTR_OPT_STEPS] += 1;
    } else x->checked_grads = single1 ? d_grads : nullptr;
    return MFM_OK;
  }
  WgradArgs w; memset(&w, 0, sizeof w);
  w.net = x->net; w.ws = x->ws; w.acts = x->acts; w.dzs = x->dzs; w.jobs = x->jobs; w.n_jobs = x->n_jobs;
  w.nbb = x->cfg.n_chain_local / 16; w.split = x->split; w.slabs = x->slabs;
  // One rank: this sum is the gradient the optimizer will see, so its finite check rides in the reduction and
  // mfm_adamw_step(d_grads) skips its check kernel.  With more ranks the check has to follow the all-reduce.
  const bool single = x->cfg.n_chain_total == x->cfg.n_chain_local;
  if (with_optimizer) {
    w.flag_partial = x->flag;                  // cleared by the training kernel (FmArgs::flags_clear)
    { ProfScope ps_(x, PROF_WGRAD); launch_wgrad(w, x->stream); }
    LAUNCHCHK();
    const int force = g_sw.force_exchange ? 1 : 0;      // tests: the grid-wide decision path on ordinary gradients
    ProfScope ps2_(x, PROF_ADAM);
    launch_reduce_adamw(adam_args(x, x->slabs, x->split), x->opt_alt, d_grads, x->loss_part, x->cfg.n_chain_local / 16, d_loss, force, x->stream);
    LAUNCHCHK();
    OptState* t = x->opt; x->opt = x->opt_alt; x->opt_alt = t;
    x->ctr[CTR_OPT_STEPS] += 1;
    return MFM_OK;
  }
  w.flag_reset = single ? x->flag : nullptr;
  { ProfScope ps_(x, PROF_WGRAD); launch_wgrad(w, x->stream); }
  LAUNCHCHK();
  ProfScope ps2_(x, PROF_REDUCE);
  launch_reduce_slabs(x->slabs, x->split, x->net.n_params, d_grads, x->loss_part, x->cfg.n_chain_local / 16, d_loss, single ? x->flag : nullptr, x->stream);
  LAUNCHCHK();
  x->checked_grads = single ? d_grads : nullptr;
  return MFM_OK;
}
extern "C" int mfm_fm_loss_grad(mfm_ctx* x, uint32_t k0, uint32_t k1, const float* d_pos, double* d_loss, float* d_grads) { use_ctx(x);
  return fm_loss_grad_impl(x, k0, k1, d_pos, d_loss, d_grads, false);
}

// rows [0, rem) of `src` followed by copies of row rem - 1: the 16-row image of a partial tile (finite padding, never counted)
__global__ void pad_tile_kernel(const float* src, int rem, int d, float* dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 16 * d; i += gridDim.x * blockDim.x) {
    const int r = i / d, j = i - r * d;
    dst[i] = src[(size_t)(r < rem ? r : rem - 1) * d + j];
  }
}

extern "C" int mfm_fm_loss(mfm_ctx* x, uint32_t k0, uint32_t k1, const float* d_samples, int n, int n_total, int offset, double* d_loss) { use_ctx(x);
  NEED_TARGET();
  if (n <= 0) return fail(MFM_EINVAL, "sample count must be positive (got %d)", n);
  if (!d_samples || !d_loss) return fail(MFM_EINVAL, "null device pointer");
  // Any n: the kernels work on 16-row tiles, so the last n % 16 samples go through a 16-row staging tile of the context whose other
  // rows are padding with residual 0 (FmArgs::n_valid).  Every draw is indexed by the sample's GLOBAL index out of n_total, so the
  // two launches see the draws one launch over n rows would (eval_step on num_chain * eval_iter samples for ANY --num_chain,
  // exe_flow_matching.py:370-374, and for any split of them over ranks).
  const int n_main = n & ~15, rem = n - n_main;
  if (n_main) {
    const int rc = fm_common(x, k0, k1, d_samples, n_main, n_total, offset, false, d_loss);
    if (rc || !rem) return rc;
  }
  hipLaunchKernelGGL(pad_tile_kernel, dim3(16), dim3(256), 0, x->stream, d_samples + (size_t)n_main * x->cfg.dim, rem, x->cfg.dim, x->eval_pad);
  LAUNCHCHK();
  const int rc = fm_common(x, k0, k1, x->eval_pad, 16, n_total, offset + n_main, false, d_loss, nullptr, rem, n_main > 0);
  x->ctr[CTR_FM_EVAL] -= 16 - rem;      // (fm_common counted the staging tile's 16 rows)
  return rc;
}

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------------
// The library carries no link-time dependency on RCCL (a host that never calls mfm_comm_* needs none): the first call looks for
// a copy the process has already loaded (PyTorch ships one under the same soname), then for the system's.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int*);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
};
static const char* g_rccl_why = "";
static RcclApi* rccl_api() {
  // resolved once, whichever thread comes first (std::call_once); the reason of a failed load is kept for the error text
  static RcclApi api; static bool ok = false; static std::once_flag once; static char why[256] = "";
  std::call_once(once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) { const char* e = dlerror(); snprintf(why, sizeof why, "%s", e ? e : "dlopen failed"); return; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
    api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommCount && api.AllReduce && api.GetErrorString;
    if (!ok) snprintf(why, sizeof why, "librccl.so.1 lacks an expected symbol");
  });
  g_rccl_why = why;
  return ok ? &api : nullptr;
}
#define RCCLCHK(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(MFM_EHIP, "RCCL: %s", R->GetErrorString(r_)); } while (0)

extern "C" int mfm_comm_unique_id(uint8_t out[MFM_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == MFM_COMM_ID_BYTES, "ncclUniqueId size");
  RcclApi* R = rccl_api();
  if (!R) return fail(MFM_EUNSUPPORTED, "librccl.so.1 cannot be loaded: %s", g_rccl_why);
  if (!out) return fail(MFM_EINVAL, "null argument");
  ncclUniqueId id;
  RCCLCHK(R->GetUniqueId(&id));
  memcpy(out, &id, sizeof id);
  return MFM_OK;
}

extern "C" int mfm_comm_destroy(mfm_ctx* x) { use_ctx(x);
  if (!x) return fail(MFM_EINVAL, "null ctx");
  // keyed on what exists, not on the communicator: a mfm_comm_init that failed half-way leaves a stream / events behind
  if (x->comm_stream) (void)hipStreamSynchronize(x->comm_stream);
  if (x->comm) { RcclApi* R = rccl_api(); if (R) (void)R->CommDestroy(x->comm); }
  if (x->ev_grads) (void)hipEventDestroy(x->ev_grads);
  if (x->ev_comm) (void)hipEventDestroy(x->ev_comm);
  if (x->comm_stream) (void)hipStreamDestroy(x->comm_stream);
  x->comm = nullptr; x->comm_stream = nullptr; x->ev_grads = x->ev_comm = nullptr; x->comm_pending = nullptr; x->comm_nranks = 0;
  return MFM_OK;
}

extern "C" int mfm_comm_init(mfm_ctx* x, int nranks, int rank, const uint8_t id_bytes[MFM_COMM_ID_BYTES]) { use_ctx(x);
  if (!x || !id_bytes) return fail(MFM_EINVAL, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MFM_EINVAL, "bad rank %d of %d", rank, nranks);
  if (x->comm) return fail(MFM_EINVAL, "the context already owns a communicator (mfm_comm_destroy first)");
  if ((long long)x->cfg.n_chain_local * nranks != (long long)x->cfg.n_chain_total)
    return fail(MFM_EINVAL, "n_chain_total (%d) is not nranks (%d) x n_chain_local (%d)", x->cfg.n_chain_total, nranks, x->cfg.n_chain_local);
  RcclApi* R = rccl_api();
  if (!R) return fail(MFM_EUNSUPPORTED, "librccl.so.1 cannot be loaded: %s", g_rccl_why);
  ncclUniqueId id; memcpy(&id, id_bytes, sizeof id);
  int rc = MFM_OK;
  if (hipStreamCreateWithFlags(&x->comm_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&x->ev_grads, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&x->ev_comm, hipEventDisableTiming) != hipSuccess)
    rc = fail(MFM_EHIP, "communication stream / events: %s", hipGetErrorString(hipGetLastError()));
  if (rc == MFM_OK) {
    const ncclResult_t r = R->CommInitRank(&x->comm, nranks, id, rank);
    if (r != ncclSuccess) { x->comm = nullptr; rc = fail(MFM_EHIP, "RCCL: %s", R->GetErrorString(r)); }
  }
  if (rc != MFM_OK) { (void)mfm_comm_destroy(x); return rc; }      // nothing of a failed init survives: a retry starts clean
  x->comm_nranks = nranks;
  return MFM_OK;
}

// Ranks of the context's communicator AS RCCL REPORTS THEM (ncclCommCount), 0 without a communicator: what a scaling record
// cites to show that the all-reduce spanned N ranks.
extern "C" int mfm_comm_count(mfm_ctx* x, int32_t* out) { use_ctx(x);
  if (!x || !out) return fail(MFM_EINVAL, "null argument");
  *out = 0;
  if (!x->comm) return MFM_OK;
  RcclApi* R = rccl_api();
  int n = 0;
  RCCLCHK(R->CommCount(x->comm, &n));
  *out = n;
  return MFM_OK;
}

// Starts the SUM all-reduce of the flow-matching gradient over the context's communicator on the context's communication
// stream, ordered after everything queued on the context's stream so far; returns at once.  Work the caller queues next that
// touches neither the gradient nor the parameters (the MALA step of the following iteration) overlaps it; the next
// mfm_adamw_step on the same buffer waits for it.
extern "C" int mfm_grad_allreduce_begin(mfm_ctx* x, float* d_grads) { use_ctx(x);
  if (!x || !d_grads) return fail(MFM_EINVAL, "null argument");
  if (!x->comm) return fail(MFM_EINVAL, "no communicator (mfm_comm_init)");
  if (x->comm_pending) return fail(MFM_EINVAL, "an all-reduce is already in flight: apply it with mfm_adamw_step first");
  RcclApi* R = rccl_api();
  HIPCHK(hipEventRecord(x->ev_grads, x->stream));
  HIPCHK(hipStreamWaitEvent(x->comm_stream, x->ev_grads, 0));
  RCCLCHK(R->AllReduce(d_grads, d_grads, (size_t)x->net.n_params, ncclFloat32, ncclSum, x->comm, x->comm_stream));
  HIPCHK(hipEventRecord(x->ev_comm, x->comm_stream));
  x->comm_pending = d_grads;
  return MFM_OK;
}

extern "C" int mfm_adamw_step(mfm_ctx* x, const float* d_grads) { use_ctx(x);
  if (!x || !d_grads) return fail(MFM_EINVAL, "null argument");
  AdamArgs a = adam_args(x, d_grads, 1);
  a.inline_decide = (x->checked_grads == d_grads) ? 1 : 0;
  x->checked_grads = nullptr;
  if (x->comm) {
    // the gradient every rank applies is the SUM over the communicator (exe_flow_matching.py:178 sums over ALL chains): started
    // earlier by mfm_grad_allreduce_begin, or here
    if (x->comm_pending != d_grads) {
      if (x->comm_pending) return fail(MFM_EINVAL, "an all-reduce of another buffer is in flight");
      if (!g_sw.rccl_comm_stream) {     // nothing to overlap with: the collective in line on the context's stream, no event hops
        RcclApi* R = rccl_api();
        RCCLCHK(R->AllReduce(d_grads, const_cast<float*>(d_grads), (size_t)x->net.n_params, ncclFloat32, ncclSum, x->comm, x->stream));
      } else {
        const int rc = mfm_grad_allreduce_begin(x, const_cast<float*>(d_grads));
        if (rc) return rc;
      }
    }
    if (x->comm_pending) HIPCHK(hipStreamWaitEvent(x->stream, x->ev_comm, 0));
    x->comm_pending = nullptr;
    if (x->comm_nranks > 1) a.inline_decide = 0;      // the finite check must see the reduced gradient
  }
  ProfScope ps_(x, PROF_ADAM);
  launch_adamw(a, x->stream);
  LAUNCHCHK();
  x->ctr[CTR_OPT_STEPS] += 1;
  return MFM_OK;
}

extern "C" int mfm_flow_step(mfm_ctx* x, int mode, uint32_t k0, uint32_t k1, double beta, float* d_pos, double* d_logp, float* d_grad,
                             float* d_acc, uint8_t* d_is_acc, float* d_prop, int32_t* d_nsteps);

// exe_flow_matching.py:432-439 as one call: generator (:300-314) + train_step (:362-368); see include/mfm.h
extern "C" int mfm_train_iter(mfm_ctx* x, int64_t count, int K, int flow_mode, uint32_t gk0, uint32_t gk1, uint32_t tk0, uint32_t tk1,
                              double beta, double step_size, float* d_pos, double* d_logp, float* d_grad, float* d_acc,
                              int32_t* d_nsteps, double* d_loss, float* d_grads, int apply_update) { use_ctx(x);
  if (!x) return fail(MFM_EINVAL, "null context");
  if (K < 1) return fail(MFM_EINVAL, "mfm_train_iter serves mcmc_per_flow_steps >= 1; compose the other schedules from the separate calls");
  if (count < 0) return fail(MFM_EINVAL, "count must be non-negative");
  int rc;
  const bool no_fuse = g_sw.no_fused_mala;
  if (count % ((int64_t)K + 1) == 0) rc = mfm_flow_step(x, flow_mode, gk0, gk1, beta, d_pos, d_logp, d_grad, d_acc, nullptr, nullptr, d_nsteps);
  else if (x->has_target && !x->wide && !no_fuse && fm_mala_fusable(x->net)) {
    // the MALA step rides in the training kernel (fm.hip: fm_fwd_bwd_kernel<.., MALA>): same arithmetic, same draws, one launch less
    if (!d_pos || !d_logp || !d_grad) return fail(MFM_EINVAL, "null device pointer");
    if (!(step_size > 0)) return fail(MFM_EINVAL, "step_size must be positive");
    // everything fm_loss_grad_impl / fm_common would reject is rejected HERE, before the prefetched slot of this MALA step is
    // consumed: an early error must leave the prefetch cursor where it was (a retry then finds its draws)
    if (!d_loss || !d_grads) return fail(MFM_EINVAL, "null device pointer");
    if (!x->has_fourier) return fail(MFM_EINVAL, "mfm_set_fourier has not been called");
    FmMala m; memset(&m, 0, sizeof m);
    m.on = 1; m.key = Key2{gk0, gk1}; m.beta = beta; m.eps = step_size; m.logp = d_logp; m.grad = d_grad; m.acc_prob = d_acc;
    const int slot = noise_take(x, m.key, false);
    if (slot >= 0) {
      const size_t B = (size_t)x->cfg.n_chain_local;
      m.pre_n = x->noise->mala_n + (size_t)slot * B * x->cfg.dim; m.pre_u = x->noise->mala_u + (size_t)slot * B;
    }
    x->fuse_mala = m;
    const bool with_opt = apply_update && opt_fusable(x);
    rc = fm_loss_grad_impl(x, tk0, tk1, d_pos, d_loss, d_grads, with_opt);
    x->fuse_mala.on = 0;
    // the fused step reads x, g, logp and writes x, g, logp, acc: the proposed position never leaves the workgroup (4 (4d + 4) B)
    if (rc == MFM_OK) { x->ctr[CTR_MALA] += x->cfg.n_chain_local; x->ctr[CTR_MALA_BYTES] += (int64_t)x->cfg.n_chain_local * 4 * (4 * x->cfg.dim + 4); }
    if (rc || !apply_update || with_opt) return rc;
    return mfm_adamw_step(x, d_grads);
  }
  else rc = mfm_mala_step(x, gk0, gk1, beta, step_size, 0, d_pos, d_logp, d_grad, d_acc, nullptr, nullptr, nullptr);
  if (rc) return rc;
  const bool with_opt = apply_update && x->has_target && opt_fusable(x);
  rc = fm_loss_grad_impl(x, tk0, tk1, d_pos, d_loss, d_grads, with_opt);
  if (rc || !apply_update || with_opt) return rc;
  return mfm_adamw_step(x, d_grads);
}

extern "C" int mfm_opt_state(mfm_ctx* x, int32_t out[4], float* lr) { use_ctx(x);
  if (!x || !out) return fail(MFM_EINVAL, "null argument");
  OptState s;
  HIPCHK(hipStreamSynchronize(x->stream));
  HIPCHK(hipMemcpy(&s, x->opt, sizeof s, hipMemcpyDeviceToHost));
  out[0] = s.step; out[1] = s.count; out[2] = s.notfinite_count; out[3] = s.last_applied;
  if (lr) *lr = s.last_lr;
  return MFM_OK;
}

// ---- algorithmic counters (mfm_get_counters) ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tally_attempts_kernel(const int* nsteps, int n, unsigned long long* out) {
  __shared__ unsigned long long part[4];
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += (unsigned long long)nsteps[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
// the solver kernels report attempted steps per sample; where the caller did not ask for them the context lends a buffer
static int* att_buffer(mfm_ctx* x, int32_t* d_nsteps, int n) {
  if (d_nsteps) return d_nsteps;
  if ((size_t)n > x->att_cap) {
    if (x->att_buf) { (void)hipStreamSynchronize(x->stream); (void)hipFree(x->att_buf); x->att_buf = nullptr; x->att_cap = 0; }
    if (hipMalloc((void**)&x->att_buf, (size_t)n * sizeof(int)) != hipSuccess) return nullptr;
    x->att_cap = (size_t)n;
  }
  return x->att_buf;
}
static void tally_solves(mfm_ctx* x, const int* nsteps, int n, int solves_per_sample) {
  x->ctr[CTR_SOLVES] += (int64_t)n * solves_per_sample;
  if (nsteps) hipLaunchKernelGGL(tally_attempts_kernel, dim3(1), dim3(256), 0, x->stream, nsteps, n, x->d_att);
}

extern "C" int mfm_get_counters(mfm_ctx* x, int64_t h_out[8]) { use_ctx(x);
  if (!x || !h_out) return fail(MFM_EINVAL, "null argument");
  unsigned long long att = 0;
  HIPCHK(hipStreamSynchronize(x->stream));
  HIPCHK(hipMemcpy(&att, x->d_att, sizeof att, hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; ++i) h_out[i] = x->ctr[i];
  h_out[CTR_ATTEMPTS] = (int64_t)att;
  h_out[CTR_FIELD_EVALS] = 2 * x->ctr[CTR_SOLVES] + 6 * (int64_t)att;      // odeint: f(y0), the initial-step probe, six stages per attempt
  return MFM_OK;
}
extern "C" int mfm_reset_counters(mfm_ctx* x) { use_ctx(x);
  if (!x) return fail(MFM_EINVAL, "null ctx");
  for (int i = 0; i < 8; ++i) x->ctr[i] = 0;
  HIPCHK(hipMemsetAsync(x->d_att, 0, sizeof(unsigned long long), x->stream));
  return MFM_OK;
}

extern "C" int mfm_debug_replay(mfm_ctx* x, int cap, const float* d_dt, const uint8_t* d_acc, float* d_ratio, float* d_dt_own, double* d_diag) { use_ctx(x);
  if (!x) return fail(MFM_EINVAL, "null ctx");
  if (!d_dt) { memset(&x->replay, 0, sizeof x->replay); return MFM_OK; }          // disarm
  if (!d_acc || !d_ratio || !d_dt_own || cap < 2) return fail(MFM_EINVAL, "mfm_debug_replay needs all four arrays and cap >= 2");
  x->replay.dt = d_dt; x->replay.acc = d_acc; x->replay.ratio = d_ratio; x->replay.dt_own = d_dt_own; x->replay.cap = cap; x->replay.n = 0; x->replay.diag = d_diag;
  return MFM_OK;
}

extern "C" int mfm_vf_apply(mfm_ctx* x, const float* d_x, const float* d_t, const float* d_tan, int n, float* d_v, float* d_jvp) { use_ctx(x);
  NEED_TARGET();
  if (!x->has_fourier) return fail(MFM_EINVAL, "mfm_set_fourier has not been called");
  if (!d_x || !d_t || !d_v || ((d_tan == nullptr) != (d_jvp == nullptr))) return fail(MFM_EINVAL, "bad pointer arguments");
  if (n <= 0 || n % 16) return fail(MFM_EUNSUPPORTED, "n must be a positive multiple of 16");
  if (x->wide) {
    if (wide::vf_apply(x->wide, x->net, d_x, d_t, d_tan, n, d_v, d_jvp, x->stream)) return fail(MFM_EHIP, "wide vf_apply failed");
    LAUNCHCHK();
    return MFM_OK;
  }
  int rc = launch_vf_apply(x->net, d_x, d_t, d_tan, n, d_v, d_jvp, x->stream);
  if (rc) return fail(rc, "vf_apply cannot be launched for this configuration");
  LAUNCHCHK();
  return MFM_OK;
}

extern "C" int mfm_ode_transform(mfm_ctx* x, int direction, int per_chain, const uint32_t* d_keys, uint32_t k0, uint32_t k1,
                                 const float* d_in, int n, float* d_out, float* d_ldj, int32_t* d_nsteps) { use_ctx(x);
  NEED_TARGET();
  if (!x->has_fourier) return fail(MFM_EINVAL, "mfm_set_fourier has not been called");
  if (!d_in || !d_out || !d_ldj || (per_chain && !d_keys)) return fail(MFM_EINVAL, "null device pointer");
  if (direction != 1 && direction != -1) return fail(MFM_EINVAL, "direction must be +1 or -1");
  if (n <= 0 || n % 16) return fail(MFM_EUNSUPPORTED, "n must be a positive multiple of 16");
  if ((size_t)n > x->ode.rows) return fail(MFM_ETOOLARGE, "n=%d exceeds max_eval_samples given at mfm_create", n);
  OdeArgs a = ode_args(x->net, x->cfg, x->ode);
  a.direction = direction; a.per_chain_keys = per_chain; a.keys = d_keys; a.key = Key2{k0, k1};
  a.in = d_in; a.out = d_out; a.ldj = d_ldj; a.n = n;
  a.nsteps = d_nsteps = att_buffer(x, d_nsteps, n);
  a.rp = x->replay; a.rp.n = n; memset(&x->replay, 0, sizeof x->replay);      // one-shot
  if ((x->wide || x->wide_ex) && a.fixed_steps > 0) return fail(MFM_EUNSUPPORTED, "fixed-step mode (ode_method / ode_steps): built for the shape-specialised solver");
  if (wide::Ctx* ws = x->wide ? x->wide : x->wide_ex) {
    launch_probe(per_chain ? 0 : 1, d_keys, a.key, 0, 0, 0, n, x->net.d, const_cast<float*>(a.z1), x->stream);
    wide::WReplay wr{a.rp.dt, a.rp.acc, a.rp.ratio, a.rp.dt_own, a.rp.cap, a.rp.n, 0, 0, nullptr};
    const int rcw = wide::transform(ws, x->net, direction, a.rtol, a.atol, a.max_attempts, a.z1, d_in, n, d_out, d_ldj, d_nsteps, x->stream, wr);
    if (rcw) return fail(rcw, "wide ODE transform failed: %s", hipGetErrorString(hipGetLastError()));
    LAUNCHCHK();
    tally_solves(x, d_nsteps, n, 1);
    return MFM_OK;
  }
  int rc = launch_ode_transform(a, x->stream);
  if (rc == -4) return fail(MFM_EUNSUPPORTED, "fixed-step mode (ode_method / ode_steps): built for the shape-specialised solver -- default widths, PhiFour, relu, hutch");
  if (rc) return fail(rc, "ODE kernel cannot be launched for this configuration");
  LAUNCHCHK();
  tally_solves(x, d_nsteps, n, 1);
  return MFM_OK;
}

extern "C" int mfm_flow_step(mfm_ctx* x, int mode, uint32_t k0, uint32_t k1, double beta, float* d_pos, double* d_logp,
                             float* d_grad, float* d_acc, uint8_t* d_isacc, float* d_prop, int32_t* d_nsteps) { use_ctx(x);
  NEED_TARGET();
  if (!x->has_fourier) return fail(MFM_EINVAL, "mfm_set_fourier has not been called");
  if (!d_pos || !d_logp || !d_grad) return fail(MFM_EINVAL, "null device pointer");
  if (mode != MFM_FLOW_RWMH && mode != MFM_FLOW_IMH) return fail(MFM_EINVAL, "unknown flow step mode %d", mode);
  OdeArgs a = ode_args(x->net, x->cfg, x->ode);
  a.n = x->cfg.n_chain_local;
  FlowArgs f; memset(&f, 0, sizeof f);
  f.mode = mode; f.key = Key2{k0, k1}; f.n_total = x->cfg.n_chain_total; f.chain_offset = x->cfg.chain_offset;
  f.beta = beta; f.ref_std = (float)x->cfg.ref_std; f.pos = d_pos; f.logp = d_logp; f.grad = d_grad; f.acc_prob = d_acc; f.accepted = d_isacc;
  f.proposed = d_prop; f.nsteps = d_nsteps = att_buffer(x, d_nsteps, a.n);
  a.rp = x->replay; a.rp.n = a.n; memset(&x->replay, 0, sizeof x->replay);    // one-shot
  // draws of the following iterations, produced by the workgroups of this launch whose tile is done (noise.hip); only the
  // shape-specialised kernel carries that tail: elsewhere the request is dropped and the consumers draw in line
  NoiseArgs nz; memset(&nz, 0, sizeof nz);
  if (x->noise) {
    NoiseWs* w = x->noise;
    w->n_valid = 0;
    const bool fast_path = !x->wide && fast::shape_ok(x->net, x->cfg.hutch) && !g_sw.generic_ode && a.fixed_steps == 0;
    if (w->n_armed > 0 && fast_path) {
      const int B = x->cfg.n_chain_local;
      nz.gn = w->d_keys; nz.st = w->d_keys + 2 * (size_t)w->n_armed; nz.n_slots = w->n_armed;
      nz.n_total = (uint32_t)x->cfg.n_chain_total; nz.chain_offset = (uint32_t)x->cfg.chain_offset; nz.B = B; nz.d = x->cfg.dim;
      nz.mala_n = w->mala_n; nz.mala_u = w->mala_u; nz.fm_x0 = w->fm_x0; nz.fm_eps = w->fm_eps; nz.fm_t = w->fm_t;
      nz.counter = w->counter; nz.groups = (B + 7) / 8; nz.n_items = nz.groups * w->n_armed;
      nz.skip_mala0 = w->gn[0].k0 == k0 && w->gn[0].k1 == k1; w->no_mala0 = nz.skip_mala0 != 0;      // (slot 0 = this flow iteration's own keys: its MALA draws are never used)
      HIPCHK(hipMemsetAsync(w->counter, 0, sizeof(int), x->stream));
      w->n_valid = w->n_armed; w->cur_gn = w->cur_st = 0;
    }
    w->n_armed = 0;
  }
  int rc = 0;
  {
    ProfScope ps_(x, PROF_FLOW);
    if ((x->wide || x->wide_ex) && a.fixed_steps > 0) return fail(MFM_EUNSUPPORTED, "fixed-step mode (ode_method / ode_steps): built for the shape-specialised solver");
    if (wide::Ctx* ws = x->wide ? x->wide : x->wide_ex) {
      launch_probe(2, nullptr, f.key, f.n_total, f.chain_offset, 0, a.n, x->net.d, const_cast<float*>(a.zgen), x->stream);     // key_gen
      launch_probe(2, nullptr, f.key, f.n_total, f.chain_offset, 3, a.n, x->net.d, const_cast<float*>(a.z1), x->stream);       // key_hutch2
      launch_probe(2, nullptr, f.key, f.n_total, f.chain_offset, 2, a.n, x->net.d, const_cast<float*>(a.z2), x->stream);       // key_hutch1
      wide::FlowCall c; memset(&c, 0, sizeof c);
      c.mode = mode; c.key = f.key; c.n_total = f.n_total; c.chain_offset = f.chain_offset; c.beta = beta; c.rows = a.n; c.ref_std = f.ref_std;
      c.rtol = a.rtol; c.atol = a.atol; c.max_attempts = a.max_attempts; c.z_inv = a.z1; c.z_fwd = a.z2; c.zgen = a.zgen;
      c.pos = d_pos; c.logp = d_logp; c.grad = d_grad; c.acc_prob = d_acc; c.accepted = d_isacc; c.proposed = d_prop; c.nsteps = d_nsteps;
      c.rp = wide::WReplay{a.rp.dt, a.rp.acc, a.rp.ratio, a.rp.dt_own, a.rp.cap, a.rp.n, 0, 0, a.rp.diag};
      const int rcw = wide::flow_step(ws, x->net, c, x->stream);
      if (rcw) return fail(rcw, "wide flow step failed: %s", hipGetErrorString(hipGetLastError()));
    } else {
      rc = launch_flow_step(a, f, nz, x->stream);
      if (rc == -4) return fail(MFM_EUNSUPPORTED, "fixed-step mode (ode_method / ode_steps): built for the shape-specialised solver -- default widths, PhiFour, relu, hutch, random-walk flow step");
      if (rc) return fail(rc, "flow step cannot be launched for this configuration");
    }
    LAUNCHCHK();
  }
  // (rows >= n_chain_valid of a padded shard are integrated like any other but are no chains: not counted)
  tally_solves(x, d_nsteps, x->cfg.n_chain_valid > 0 && x->cfg.n_chain_valid < a.n ? x->cfg.n_chain_valid : a.n, 2);
  return MFM_OK;
}

extern "C" int mfm_beta_update(mfm_ctx* x, double prev_beta, const double* d_ll, int n, double alpha, double* h_out) { use_ctx(x);
  if (!x || !d_ll || !h_out) return fail(MFM_EINVAL, "null argument");
  if (n <= 0) return fail(MFM_EINVAL, "n must be positive");
  launch_beta(prev_beta, d_ll, n, alpha, x->beta_out, x->stream);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(h_out, x->beta_out, sizeof(double), hipMemcpyDeviceToHost, x->stream));
  HIPCHK(hipStreamSynchronize(x->stream));
  return MFM_OK;
}

// ---- reference-distribution rows and the CIS selection step ---------------------------------------------------------
extern "C" int mfm_normal_rows(mfm_ctx* x, const uint32_t* d_keys, int n, float* d_out) { use_ctx(x);
  if (!x || !d_keys || !d_out) return fail(MFM_EINVAL, "null argument");
  if (n <= 0) return fail(MFM_EINVAL, "n must be positive");
  launch_probe(0, d_keys, Key2{0, 0}, 0, 0, 0, n, x->cfg.dim, d_out, x->stream);
  LAUNCHCHK();
  return MFM_OK;
}

extern "C" int mfm_cis_select(mfm_ctx* x, uint32_t k0, uint32_t k1, int n_is, const float* d_u0, const float* d_vol0, const float* d_refs,
                              const float* d_xs, const float* d_vols, const double* d_lps, float* d_pos, double* d_logp, float* d_acc,
                              uint8_t* d_isacc, float* d_prop, float* d_weight) { use_ctx(x);
  if (!x || !d_u0 || !d_vol0 || !d_refs || !d_xs || !d_vols || !d_lps || !d_pos || !d_logp) return fail(MFM_EINVAL, "null argument");
  if (n_is <= 0) return fail(MFM_EINVAL, "num_importance_samples must be positive");
  CisArgs a; memset(&a, 0, sizeof a);
  a.key = Key2{k0, k1}; a.n_total = x->cfg.n_chain_total; a.chain_offset = x->cfg.chain_offset;
  a.B = x->cfg.n_chain_local; a.d = x->cfg.dim; a.n_is = n_is; a.ref_std = x->cfg.ref_std;
  a.u0 = d_u0; a.vol0 = d_vol0; a.refs = d_refs; a.xs = d_xs; a.vols = d_vols; a.lps = d_lps;
  a.pos = d_pos; a.logp = d_logp; a.acc_prob = d_acc; a.accepted = d_isacc; a.proposed = d_prop; a.weight = d_weight;
  ProfScope ps_(x, PROF_FLOW);
  launch_cis_select(a, x->stream);
  LAUNCHCHK();
  return MFM_OK;
}

// ---- sample-quality metrics (mcmc_utils.py:28-111) ----------------------------------------------------------------
static int pair_call(mfm_ctx* x, int mode, const float* A, const float* GA, const float* B, const float* GB, int na, int nb, float beta, double h[2]) {
  double* ws = nullptr;
  const size_t parts = pair_sum_parts(na, nb);
  HIPCHK(hipMalloc((void**)&ws, (2 * parts + 2) * sizeof(double)));
  launch_pair_sum(mode, A, GA, B, GB, na, nb, x->cfg.dim, beta, ws + 2, ws, x->stream);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(h, ws, 2 * sizeof(double), hipMemcpyDeviceToHost, x->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
  (void)hipFree(ws);
  if (e != hipSuccess) return fail(MFM_EHIP, "pair-sum kernel: %s", hipGetErrorString(e));
  return MFM_OK;
}

extern "C" int mfm_stein_disc(mfm_ctx* x, const float* d_x, const float* d_grad, int n, double beta, double h_out[2]) { use_ctx(x);
  if (!x || !d_x || !d_grad || !h_out) return fail(MFM_EINVAL, "null argument");
  if (n < 2) return fail(MFM_EINVAL, "stein_disc needs at least 2 samples");
  double h[2];
  int rc = pair_call(x, 0, d_x, d_grad, d_x, d_grad, n, n, (float)(-beta), h);      /* mcmc_utils.py:54: beta = -beta */
  if (rc) return rc;
  h_out[0] = (h[0] - h[1]) / ((double)n * (double)(n - 1));                         /* :85 U-statistic */
  h_out[1] = h[0] / ((double)n * (double)n);                                        /*     V-statistic */
  return MFM_OK;
}

extern "C" int mfm_max_mean_disc(mfm_ctx* x, const float* d_x, const float* d_y, int m, double* h_out) { use_ctx(x);
  if (!x || !d_x || !d_y || !h_out) return fail(MFM_EINVAL, "null argument");
  if (m < 2) return fail(MFM_EINVAL, "max_mean_disc needs at least 2 samples");
  double xx[2], yy[2], xy[2];
  int rc = pair_call(x, 1, d_x, nullptr, d_x, nullptr, m, m, 0.f, xx); if (rc) return rc;
  rc = pair_call(x, 1, d_y, nullptr, d_y, nullptr, m, m, 0.f, yy); if (rc) return rc;
  rc = pair_call(x, 1, d_x, nullptr, d_y, nullptr, m, m, 0.f, xy); if (rc) return rc;
  const double m2 = (double)m * (double)m;
  *h_out = (xx[0] - m) / (m2 - m) - 2.0 * xy[0] / m2 + (yy[0] - m) / (m2 - m);      /* mcmc_utils.py:106-110 */
  return MFM_OK;
}

// ---- noise prefetch (noise.hip) -------------------------------------------------------------------------------------------
extern "C" int mfm_noise_prefetch(mfm_ctx* x, int n_slots, const uint32_t* h_keys_gn, const uint32_t* h_keys_step) { use_ctx(x);
  NEED_TARGET();
  if (!h_keys_gn || !h_keys_step || n_slots <= 0) return fail(MFM_EINVAL, "bad arguments");
  if (x->wide || !x->cfg.cond_flow || !fast::shape_ok(x->net, x->cfg.hutch) || (x->cfg.ode_method != MFM_ODE_DOPRI5 && x->cfg.ode_steps > 0))
    return fail(MFM_EUNSUPPORTED, "the noise prefetch rides in the tail of the shape-specialised flow-step kernel (headline network shape, PhiFour, --hutch)");
  if (!x->noise) x->noise = new NoiseWs();
  NoiseWs* w = x->noise;
  const size_t B = (size_t)x->cfg.n_chain_local, d = (size_t)x->cfg.dim;
  if (n_slots > w->cap) {
    HIPCHK(hipStreamSynchronize(x->stream));
    for (void* p : {(void*)w->mala_n, (void*)w->mala_u, (void*)w->fm_x0, (void*)w->fm_eps, (void*)w->fm_t, (void*)w->d_keys}) if (p) (void)hipFree(p);
    w->cap = n_slots; w->n_valid = 0;
    ALLOC(w->mala_n, (size_t)n_slots * B * d); ALLOC(w->mala_u, (size_t)n_slots * B);
    ALLOC(w->fm_x0, (size_t)n_slots * B * d); ALLOC(w->fm_eps, (size_t)n_slots * B * d); ALLOC(w->fm_t, (size_t)n_slots * B);
    ALLOC(w->d_keys, (size_t)4 * n_slots);
    if (!w->counter) ALLOC(w->counter, 4);
  }
  // the key upload is ordered on the context's stream before the flow step that consumes it; the staging copy must outlive
  // the asynchronous copy of a pageable source, which HIP stages synchronously
  w->h_keys.assign(h_keys_gn, h_keys_gn + 2 * (size_t)n_slots);
  w->h_keys.insert(w->h_keys.end(), h_keys_step, h_keys_step + 2 * (size_t)n_slots);
  HIPCHK(hipMemcpyAsync(w->d_keys, w->h_keys.data(), w->h_keys.size() * sizeof(uint32_t), hipMemcpyHostToDevice, x->stream));
  w->gn.resize(n_slots); w->st.resize(n_slots);
  for (int j = 0; j < n_slots; ++j) {
    w->gn[j] = Key2{h_keys_gn[2 * j], h_keys_gn[2 * j + 1]};
    w->st[j] = Key2{h_keys_step[2 * j], h_keys_step[2 * j + 1]};
  }
  w->n_armed = n_slots; w->n_valid = 0;
  return MFM_OK;
}
extern "C" int mfm_noise_drop(mfm_ctx* x) { use_ctx(x);
  if (!x) return fail(MFM_EINVAL, "null ctx");
  if (x->noise) { x->noise->n_valid = 0; x->noise->n_armed = 0; }
  return MFM_OK;
}

// ---- N4: adaptive tempered SMC pieces (bblackjax/smc; exe_others.py:79-111) -------------------------------------------
extern "C" int mfm_smc_delta(mfm_ctx* x, const double* d_ll, int n, double target_ess, double max_delta, double* h_delta) { use_ctx(x);
  if (!x || !d_ll || !h_delta) return fail(MFM_EINVAL, "null argument");
  if (n <= 0) return fail(MFM_EINVAL, "n must be positive");
  hipLaunchKernelGGL(smc_delta_kernel, dim3(1), dim3(SMC_THREADS), 0, x->stream, d_ll, n, target_ess, max_delta, x->beta_out);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(h_delta, x->beta_out, sizeof(double), hipMemcpyDeviceToHost, x->stream));
  HIPCHK(hipStreamSynchronize(x->stream));
  return MFM_OK;
}
extern "C" int mfm_smc_weights(mfm_ctx* x, const double* d_ll, int n, double delta, double* d_weights, double* h_lognorm) { use_ctx(x);
  if (!x || !d_ll || !d_weights) return fail(MFM_EINVAL, "null argument");
  if (n <= 0) return fail(MFM_EINVAL, "n must be positive");
  hipLaunchKernelGGL(smc_weights_kernel, dim3(1), dim3(SMC_THREADS), 0, x->stream, d_ll, n, delta, d_weights, x->beta_out + 1);
  LAUNCHCHK();
  if (h_lognorm) {
    HIPCHK(hipMemcpyAsync(h_lognorm, x->beta_out + 1, sizeof(double), hipMemcpyDeviceToHost, x->stream));
    HIPCHK(hipStreamSynchronize(x->stream));
  }
  return MFM_OK;
}
extern "C" int mfm_smc_resample(mfm_ctx* x, uint32_t k0, uint32_t k1, const double* d_weights, int n, double* d_scratch, int32_t* d_idx) { use_ctx(x);
  if (!x || !d_weights || !d_scratch || !d_idx) return fail(MFM_EINVAL, "null argument");
  if (n <= 0) return fail(MFM_EINVAL, "n must be positive");
  hipLaunchKernelGGL(smc_resample_kernel, dim3(1), dim3(SMC_THREADS), 0, x->stream, Key2{k0, k1}, d_weights, n, d_scratch, d_idx);
  LAUNCHCHK();
  return MFM_OK;
}
extern "C" int mfm_smc_resample_scheme(mfm_ctx* x, int scheme, uint32_t k0, uint32_t k1, const double* d_weights, int n, double* d_scratch, int32_t* d_idx) { use_ctx(x);
  if (scheme == MFM_RESAMPLE_SYSTEMATIC) return mfm_smc_resample(x, k0, k1, d_weights, n, d_scratch, d_idx);
  if (!x || !d_weights || !d_scratch || !d_idx) return fail(MFM_EINVAL, "null argument");
  if (n <= 0) return fail(MFM_EINVAL, "n must be positive");
  if (scheme != MFM_RESAMPLE_STRATIFIED && scheme != MFM_RESAMPLE_MULTINOMIAL) return fail(MFM_EINVAL, "unknown resampling scheme %d", scheme);
  hipLaunchKernelGGL(smc_resample2_kernel, dim3(1), dim3(SMC_THREADS), 0, x->stream, scheme, Key2{k0, k1}, d_weights, n, d_scratch, d_idx);
  LAUNCHCHK();
  return MFM_OK;
}
extern "C" int mfm_choice_logw(mfm_ctx* x, uint32_t k0, uint32_t k1, const double* d_logw, int n, int m, double* d_scratch, int32_t* d_idx) { use_ctx(x);
  if (!x || !d_logw || !d_scratch || !d_idx) return fail(MFM_EINVAL, "null argument");
  if (n <= 0 || m <= 0) return fail(MFM_EINVAL, "n and m must be positive");
  hipLaunchKernelGGL(choice_logw_kernel, dim3(1), dim3(SMC_THREADS), 0, x->stream, Key2{k0, k1}, d_logw, n, m, d_scratch, d_idx);
  LAUNCHCHK();
  return MFM_OK;
}
extern "C" int mfm_acc_stats(mfm_ctx* x, const float* d_x, int n, double* d_out) { use_ctx(x);
  if (!x || !d_x || !d_out) return fail(MFM_EINVAL, "null argument");
  if (n <= 0) return fail(MFM_EINVAL, "n must be positive");
  hipLaunchKernelGGL(acc_stats_kernel, dim3(1), dim3(SMC_THREADS), 0, x->stream, d_x, n, d_out);
  LAUNCHCHK();
  return MFM_OK;
}
extern "C" int mfm_gather_rows(mfm_ctx* x, const float* d_src, const int32_t* d_idx, int n, int dim, float* d_dst) { use_ctx(x);
  if (!x || !d_src || !d_idx || !d_dst) return fail(MFM_EINVAL, "null argument");
  if (n <= 0 || dim <= 0) return fail(MFM_EINVAL, "n and dim must be positive");
  if (d_src == d_dst) return fail(MFM_EINVAL, "gather_rows is not in-place");
  const size_t tot = (size_t)n * dim;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096)), dim3(256), 0, x->stream, d_src, d_idx, n, dim, d_dst);
  LAUNCHCHK();
  return MFM_OK;
}

#ifdef MFM_FM_STAMPS
extern "C" int mfm_debug_fm_buffer(unsigned long long* d_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fm_dbg), &d_buf, sizeof d_buf) == hipSuccess ? 0 : MFM_EHIP;
}
#endif
#ifdef MFM_WSK_STAMPS
extern "C" int mfm_debug_wsk_buffer(unsigned long long* d_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wsk_dbg), &d_buf, sizeof d_buf) == hipSuccess ? 0 : MFM_EHIP;
}
#endif
#ifdef MFM_STAMPS
extern "C" int mfm_debug_flow_buffer(unsigned long long* d_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_flow_dbg), &d_buf, sizeof d_buf) == hipSuccess ? 0 : MFM_EHIP;
}
extern "C" int mfm_debug_eval_stamps(mfm_ctx* x, const float* d_x, const float* d_t, const float* d_tan, int n, int reps, unsigned long long* d_stamps) { use_ctx(x);
  int rc = launch_eval_stamps(x->net, d_x, d_t, d_tan, n, reps, d_stamps, x->stream);
  if (rc) return fail(rc, "eval_stamps cannot be launched");
  LAUNCHCHK();
  return MFM_OK;
}
#endif
