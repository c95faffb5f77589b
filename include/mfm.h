/* libmfm_hip -- C ABI of the MI355X-native Markovian Flow Matching inner loop.
 *
 * This is the drop-in boundary for the hot path of albcab/mfm (reference paths relative to /root/reference):
 * every entry point replaces one jit-compiled XLA executable (or a piece of one) that the reference's `run()` loop
 * (exe_flow_matching.py:432-449) calls.  The reference has no FFI of its own (it is pure Python on JAX); the
 * binding a maintainer adds is the ctypes layer in mfm_amd/_lib.py.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0 on success and a negative
 *    MFM_E* code on failure; mfm_last_error() gives the message.  No exceptions or aborts cross the boundary.
 *  - Pointers named d_* are DEVICE pointers owned by the caller (e.g. torch.Tensor.data_ptr()); h_* are host
 *    pointers.  Chain state is row-major [n_chain_local, dim] float32; log-densities are float64 [n_chain_local].
 *  - The context owns the network parameters (canonical flat float32 vector + MFMA-packed copies), the AdamW
 *    state, all workspaces and remembers the HIP stream work is queued on.  Calls are asynchronous on that stream;
 *    mfm_sync() or any h_* output synchronises.  One context per GPU per process; not thread-safe.
 *  - PRNG keys are jax-style uint32[2] passed by value as two words; draws are indexed by GLOBAL chain id
 *    (chain_offset + local row) out of n_chain_total, so results do not depend on how chains are sharded.
 *
 * Canonical flat parameter layout (mfm_set_params / mfm_get_params / gradients): for Dense_0 .. Dense_7 in flax
 * creation order (exe_flow_matching.py:74-86): kernel [in][out] row-major, then bias [out].
 */
#ifndef MFM_H
#define MFM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfm_ctx mfm_ctx;

enum { MFM_OK = 0, MFM_EINVAL = -1, MFM_EUNSUPPORTED = -2, MFM_ETOOLARGE = -3, MFM_EHIP = -4, MFM_ENOTARGET = -5 };

enum { MFM_PHI4 = 0, MFM_GMM = 1, MFM_LGCP = 2 };                 /* distributions.py:114,42,231 */
enum { MFM_FLOW_RWMH = 0, MFM_FLOW_IMH = 1 };                      /* exe_flow_matching.py:264-278 / :246-260 */
enum { MFM_FAMILY_AUTO = 0, MFM_FAMILY_TILE = 1, MFM_FAMILY_WIDE = 2 };
enum { MFM_ACT_RELU = 0, MFM_ACT_TANH = 1, MFM_ACT_ELU = 2, MFM_ACT_GELU = 3, MFM_ACT_SWISH = 4 };   /* exe_flow_matching.py:39-45 */

#define MFM_MAX_DEPTH 3
typedef struct mfm_config {
  int32_t dim;                 /* args.dim */
  int32_t fourier_dim;         /* args.fourier_dim  (multi_modal.py:156) */
  int32_t hidden_t[2];         /* args.hidden_t     (:179)  any positive width (zero-padded to 16 inside the library) */
  int32_t hidden_x[2];         /* args.hidden_x     (:178) */
  int32_t hidden_xt[2];        /* args.hidden_xt    (:180) */
  int32_t n_chain_local;       /* chains resident on this GPU (multiple of 16) */
  int32_t n_chain_total;       /* args.num_chain over all GPUs */
  int32_t chain_offset;        /* global id of local chain 0 */
  float grad_clip;             /* args.gradient_clip if dim > 128 else 0   (exe_flow_matching.py:351) */
  float sigma;                 /* args.sigma        (:155) */
  int32_t cond_flow;           /* args.cond_flow    (:162-163) */
  int32_t hutch;               /* args.hutchs       (:158) */
  double rtol, atol;           /* args.rtol / atol  (:207-208) */
  int32_t mxstep;              /* args.mxstep       (:209) */
  int32_t n_ts;                /* 5 for "4-mode", else 2 (exe_flow_matching.py:347) */
  /* optimizer (multi_modal.py:199-205) */
  double learning_rate, adam_b1, adam_b2, adam_eps, weight_decay, update_clip;
  int32_t learning_iter, warmup_steps;
  int32_t max_eval_samples;    /* largest n passed to mfm_fm_loss / mfm_vf_apply (0: n_chain_local) */
  int32_t kernel_family;       /* MFM_FAMILY_AUTO: fused 16-chain LDS tile kernels when the network fits them, else the wide
                                  family (per-layer MFMA GEMMs on HBM-resident activations: the "pines" widths of
                                  multi_modal.py:89-96); _TILE / _WIDE force one (ETOOLARGE if _TILE does not fit) */
  int32_t activation;          /* MFM_ACT_*: args.non_linearity (multi_modal.py:177; table exe_flow_matching.py:39-45).  relu, tanh and elu
                                  run on either kernel family; gelu and swish need stored pre-activations: wide family */
  double ref_std;              /* std of the flow's reference distribution IndepGaussian(dim, var): 1 for args.ref_dist = 'stdgauss',
                                  sqrt(5) for 'widegauss' (exe_flow_matching.py:48-54, distributions.py:80-97); 0 = 1 */
  int32_t n_chain_valid;       /* 0 = n_chain_local.  args.num_chain takes any integer (multi_modal.py:169); the kernels work on tiles of
                                  16 chains, so the host pads its shard to n_chain_local (a multiple of 16) and names here how many of
                                  those rows are chains: rows >= n_chain_valid are integrated like any other but contribute nothing to
                                  the flow-matching loss and its gradient (exe_flow_matching.py:171-178) */
  /* args.hidden_t / hidden_x / hidden_xt are lists of any length (multi_modal.py:178-180, nargs='+'; the loops at
     exe_flow_matching.py:74-85).  depth_* = 0 reads the two widths above (a two-layer branch).  1 .. MFM_MAX_DEPTH: that many hidden
     layers, widths hidden_*[0], hidden_*[1], then hidden_*3.  Anything but (2, 2, 2) runs on the wide family (one GEMM launch per
     layer): MFM_FAMILY_TILE then fails with MFM_EUNSUPPORTED. */
  int32_t depth_t, depth_x, depth_xt;
  int32_t hidden_t3, hidden_x3, hidden_xt3;
  /* BUILD-SIDE MODE, not in the reference (whose integrator is the adaptive Dopri5 of jax.experimental.ode.odeint,
     exe_flow_matching.py:345-349 -- ode_method = MFM_ODE_DOPRI5, the default): the CNF solves on ode_steps EQUAL steps of classical
     RK4 (MFM_ODE_RK4) or forward Euler (MFM_ODE_EULER), the "RK4/Euler ODE integrator" BASELINE.json's north star names -- no step-size
     controller, every chain takes the same steps.  Built for the shape-specialised solver (default widths, PhiFour, relu, hutch = 1,
     random-walk flow step and the transforms); other configurations fail with MFM_EUNSUPPORTED at the first solve.  rtol / atol /
     mxstep are ignored in this mode; n_ts > 2 needs ode_steps % (n_ts - 1) == 0. */
  int32_t ode_method, ode_steps;
} mfm_config;

#define MFM_ODE_DOPRI5 0
#define MFM_ODE_RK4 1
#define MFM_ODE_EULER 2

const char* mfm_last_error(void);
int mfm_version(void);

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
int mfm_create(const mfm_config* cfg, mfm_ctx** out);
int mfm_destroy(mfm_ctx* ctx);
int mfm_set_stream(mfm_ctx* ctx, void* hip_stream);            /* hipStream_t; NULL = default stream */
int mfm_sync(mfm_ctx* ctx);
int mfm_num_params(const mfm_ctx* ctx);                        /* P_w + P_b */

/* ---- target (distributions.py) ---------------------------------------------------------------------------------
 * MFM_PHI4: h_params = {a, beta}                                    (PhiFour.__init__, :115-129)
 * MFM_GMM : h_params = {n_modes, modes[K*d], stds[K*d], weights[K]} (GaussianMixture, :43-56; stds = sqrt(covs))
 * MFM_LGCP: h_params = {mu, poisson_a, log_norm, counts[d], Kinv[d*d]}  (LogGaussianCoxPines, :233-281) */
int mfm_set_target(mfm_ctx* ctx, int kind, const double* h_params, size_t n);

/* ---- network parameters (VectorFieldNet, exe_flow_matching.py:56-90,350-353) --------------------------------- */
int mfm_set_fourier(mfm_ctx* ctx, const float* h_fourier_random);              /* [fourier_dim] */
int mfm_set_params(mfm_ctx* ctx, const float* h_flat);                          /* canonical flat layout */
int mfm_get_params(mfm_ctx* ctx, float* h_flat);
int mfm_reset_optimizer(mfm_ctx* ctx);

/* ---- K1/K2: MALA kernel API (bblackjax/mcmc/mala.py) ------------------------------------------------------------ */
/* init (mala.py:51-54) vmapped as init_fn (exe_flow_matching.py:316): logdensity and gradient of the tempered target
 * beta * loglik + logprior at d_pos. */
int mfm_mala_init(mfm_ctx* ctx, const float* d_pos, double beta, double* d_logp, float* d_grad);
/* kernel (mala.py:86-118) vmapped with per-chain keys split(key, n_chain_total) (exe_flow_matching.py:303,313).
 * State is updated in place; the MALAInfo outputs may be NULL. */
int mfm_mala_step(mfm_ctx* ctx, uint32_t key0, uint32_t key1, double beta, double step_size, int textbook,
                  float* d_pos, double* d_logp, float* d_grad,
                  float* d_acceptance_rate, uint8_t* d_is_accepted, float* d_proposed_position,
                  float* d_proposed_weight);
/* the same kernel for a caller that vmaps over its OWN per-chain keys (bblackjax/smc/base.py:122-123 ->
 * tempered.py:126-137): d_keys is uint32[n_chain_local][2], chain b uses d_keys[b] where mfm_mala_step uses
 * split(key, n_chain_total)[chain_offset + b]. */
int mfm_mala_step_keys(mfm_ctx* ctx, const uint32_t* d_keys, double beta, double step_size, int textbook,
                       float* d_pos, double* d_logp, float* d_grad,
                       float* d_acceptance_rate, uint8_t* d_is_accepted, float* d_proposed_position,
                       float* d_proposed_weight);
/* BUILD-SIDE MODE, not on the reference's MFM path (BASELINE.json's north star names a "MALA/HMC" step; the reference's loop uses
 * MALA only and vendors no hmc.py: SURVEY.md note 7): one Hamiltonian Monte Carlo step of every local chain as in blackjax's hmc
 * kernel (oracle/hmc.py restates it): momentum ~ N(0, I) from split(key, n_chain_total)[chain_offset + b] -> split(., 2)[0],
 * num_steps velocity-Verlet steps of size step_size on beta * loglik + logprior, accept with min(1, exp(H_0 - H_end)).  State
 * updated in place like mfm_mala_step; d_acceptance_rate / d_is_accepted may be null.  phi-four and mixture targets
 * (MFM_EUNSUPPORTED for the Cox process). */
int mfm_hmc_step(mfm_ctx* ctx, uint32_t key0, uint32_t key1, double beta, double step_size, int32_t num_steps,
                 float* d_pos, double* d_logp, float* d_grad, float* d_acceptance_rate, uint8_t* d_is_accepted);
/* vmap(dist.loglik) (exe_flow_matching.py:418) */
int mfm_loglik(mfm_ctx* ctx, const float* d_pos, double* d_out);

/* ---- K3/K4/K9: flow-matching loss and gradient (exe_flow_matching.py:151-178,362-365) ------------------------------ */
/* loss and parameter gradient on the local chains; d_grads [num_params] (canonical layout) and d_loss [1] are caller
 * owned so a multi-GPU host can all-reduce them (SUM) before mfm_adamw_step. */
int mfm_fm_loss_grad(mfm_ctx* ctx, uint32_t key0, uint32_t key1, const float* d_pos, double* d_loss, float* d_grads);
/* loss only on n samples (eval_step, :370-374); any n > 0 (a last partial 16-row tile is staged inside the library), draws
 * indexed out of n_total starting at offset */
int mfm_fm_loss(mfm_ctx* ctx, uint32_t key0, uint32_t key1, const float* d_samples, int n, int n_total, int offset,
                double* d_loss);

/* ---- K7: optimizer step (exe_flow_matching.py:129-137,184,366) ------------------------------------------------
 * apply_if_finite: on a single-rank context the finite check of the gradient that mfm_fm_loss_grad just wrote to d_grads
 * rides in its reduction, and mfm_adamw_step(d_grads) with the SAME pointer reuses that verdict.  A caller that changes the
 * buffer's contents in between (its own all-reduce, accumulation, clipping) must pass the result through a DIFFERENT buffer
 * (or call on a multi-rank context): any other pointer is checked afresh by the optimizer's own check kernel. */
int mfm_adamw_step(mfm_ctx* ctx, const float* d_grads);

/* ---- context-owned RCCL communicator for the gradient all-reduce ------------------------------
 * The reference has no distributed code: its loss sums over ALL chains (exe_flow_matching.py:178) inside one process.  With
 * the chains sharded over ranks that sum becomes ONE all-reduce(SUM) of the flow-matching gradient per train_step (:362-368),
 * over RCCL / xGMI.  A host without a collective layer of its own hands the context a communicator:
 *   rank 0: mfm_comm_unique_id(id) and ships the 128 bytes to the other ranks out of band (a file, a socket, MPI, ...);
 *   every rank, with its GPU current: mfm_comm_init(ctx, nranks, rank, id)  -- collective (ncclCommInitRank);
 *   per iteration: mfm_fm_loss_grad(...); mfm_grad_allreduce_begin(ctx, d_grads)  -- asynchronous, on the context's own
 *     communication stream, ordered after the work queued on the context's stream; the caller may queue the next MALA step
 *     (which touches neither gradient nor parameters) before mfm_adamw_step(ctx, d_grads), which waits for the all-reduce.
 *   mfm_adamw_step on a context with a communicator and no all-reduce in flight reduces first, IN LINE on the context's stream
 *     (no event hop; what mfm_train_iter does at more than one rank: training kernel with the MALA step inside, weight-gradient kernel,
 *     ncclAllReduce, AdamW -- measured 5 us per iteration above the one-rank sequence on a one-rank communicator, against 25 - 30 us
 *     for the overlapped form, whose two cross-stream event hops cost more than the 9 us MALA step they hide).
 * With more than one rank the apply_if_finite decision is taken on the REDUCED gradient, so every rank decides alike.
 * RCCL is resolved at run time (dlopen of librccl.so.1, preferring a copy the process already loaded): the library has no
 * link-time dependency on it and a single-GPU host needs none.  mfm_destroy destroys the communicator.
 * (The Python host, mfm_amd/engine.py, uses this path whenever its process group is RCCL; MFM_TORCH_ALLREDUCE=1 keeps torch.distributed's.) */
#define MFM_COMM_ID_BYTES 128
int mfm_comm_unique_id(uint8_t out[MFM_COMM_ID_BYTES]);
int mfm_comm_init(mfm_ctx* ctx, int nranks, int rank, const uint8_t id[MFM_COMM_ID_BYTES]);
int mfm_comm_destroy(mfm_ctx* ctx);
/* ranks of the context's communicator as RCCL reports them (ncclCommCount); 0 when the context owns none */
int mfm_comm_count(mfm_ctx* ctx, int32_t* h_out);
int mfm_grad_allreduce_begin(mfm_ctx* ctx, float* d_grads);
/* host copies of {step, count, notfinite_count, last_applied} and the learning rate logged at :367 */
int mfm_opt_state(mfm_ctx* ctx, int32_t h_out[4], float* h_last_lr);

/* ---- K5/K6: CNF transforms and flow-MH step (exe_flow_matching.py:206-242,246-278) ------------------------------ */
/* v(x, t) and optionally the x-JVP (d_tangent, d_jvp may be NULL); n multiple of 16 */
int mfm_vf_apply(mfm_ctx* ctx, const float* d_x, const float* d_t, const float* d_tangent, int n,
                 float* d_v, float* d_jvp);
/* direction +1: transform_and_logdet (:206-221); -1: inverse_and_logdet (:223-242).  Hutchinson keys: per_chain_keys
 * != 0 -> d_keys is uint32[n][2] (one key per sample, flow-MH steps); else key0/key1 is ONE key shared by all samples
 * (final sampling, :455).  d_nsteps (may be NULL) receives the attempted Dopri5 steps per sample. */
int mfm_ode_transform(mfm_ctx* ctx, int direction, int per_chain_keys, const uint32_t* d_keys, uint32_t key0,
                      uint32_t key1, const float* d_in, int n, float* d_out, float* d_ldj, int32_t* d_nsteps);
/* one flow-based MH step for every local chain with keys split(key, n_chain_total) (:303,312); state in place */
int mfm_flow_step(mfm_ctx* ctx, int mode, uint32_t key0, uint32_t key1, double beta,
                  float* d_pos, double* d_logp, float* d_grad,
                  float* d_acceptance_rate, uint8_t* d_is_accepted, float* d_proposed_position, int32_t* d_nsteps);

/* ---- L1: one loop iteration in one call (exe_flow_matching.py:432-439) ------------------------------------------ */
/* train_data_generator(key_gen, states, count, params, beta) (:300-314) followed by train_step(key_train, positions,
 * state) (:362-368): the flow-MH step of `flow_mode` when count % (mcmc_per_flow_steps + 1) == 0, the MALA step otherwise
 * (mcmc_per_flow_steps >= 1; the fractional / negative schedules of :304-310 are composed by the host from the separate
 * entry points), then loss and gradient on the NEW positions and, with apply_update != 0, the optimizer step.  A
 * multi-GPU host passes apply_update = 0, all-reduces d_grads (SUM) and calls mfm_adamw_step itself.  d_acceptance_rate
 * and d_nsteps may be NULL; d_nsteps is written by flow iterations only.  Returns MFM_OK or the first failing step's status.
 * On a MALA iteration of a phi-four target (relu network on the tile family) the step runs INSIDE the training kernel's workgroups
 * -- the arithmetic and draws of mfm_mala_step, bit for bit, one launch less (MFM_NO_FUSED_MALA=1 in the environment keeps the
 * two launches).  With apply_update != 0 on one rank (tile family, no context-owned communicator) the weight gradients, their
 * reduction over the chain slices, the apply_if_finite decision and AdamW are ONE launch (wgrad_sk.hip), bit-identical with
 * mfm_fm_loss_grad + mfm_adamw_step including skipped (non-finite) updates; MFM_NO_FUSED_OPT=1 keeps the optimizer apart.  That
 * launch's workgroups wait for one another: one context per device at a time (two contexts of one process launching it concurrently
 * on different streams could starve each other of workgroup slots). */
int mfm_train_iter(mfm_ctx* ctx, int64_t count, int mcmc_per_flow_steps, int flow_mode,
                   uint32_t gen_key0, uint32_t gen_key1, uint32_t train_key0, uint32_t train_key1,
                   double beta, double step_size, float* d_pos, double* d_logp, float* d_grad,
                   float* d_acceptance_rate, int32_t* d_nsteps, double* d_loss, float* d_grads, int apply_update);

/* ---- K8: annealing (exe_flow_matching.py:391-417) ------------------------------------------------------------- */
/* Bisection for the next beta on n (global) log-likelihoods; h_beta_out gets the new beta (synchronises). */
int mfm_beta_update(mfm_ctx* ctx, double prev_beta, const double* d_logliks, int n, double alpha, double* h_beta_out);

/* ---- measurement (bench.py): HIP-event timing of the kernels, recorded on the context's stream ------------------- */
/* class ids: 0 mala_step, 1 fm_fwd_bwd, 2 wgrad, 3 adamw (4 small kernels), 4 flow_step, 5 fm eval, 6 reductions */
/* ---- rows of the flow's reference distribution: out[i] = normal(keys[i], (dim,))  (distributions.py:93-97 vmapped at
 *      exe_flow_matching.py:285, :389, :453).  d_keys: uint32 [n][2]. ---- */
int mfm_normal_rows(mfm_ctx* ctx, const uint32_t* d_keys, int n, float* d_out);

/* ---- selection step of conditional importance sampling (exe_flow_matching.py:280-296, chosen when
 *      num_importance_samples > 0): given, per chain b, the pull-back of the current position (u0, vol0 from
 *      mfm_ode_transform direction -1 with key split(keys[b],4)[1]) and n_is flow samples (refs = normal rows of
 *      split(split(keys[b],4)[0], n_is), xs / vols = mfm_ode_transform of them with keys split(split(keys[b],4)[2], n_is),
 *      lps = tempered target log-density of xs; sample (b, j) at row b * n_is + j), draw the categorical choice with
 *      split(keys[b],4)[3] and update position / logdensity in place (the gradient is left as is, :295). ---- */
int mfm_cis_select(mfm_ctx* ctx, uint32_t key0, uint32_t key1, int n_is, const float* d_u0, const float* d_vol0,
                   const float* d_refs, const float* d_xs, const float* d_vols, const double* d_lps, float* d_pos,
                   double* d_logp, float* d_acc_prob, uint8_t* d_is_accepted, float* d_proposed, float* d_weight);

/* ---- sample-quality metrics: mcmc_utils.py:28-85 (stein_disc) and :88-111 (max_mean_disc), called at
 *      exe_flow_matching.py:469-487.  d_grad = grad log p of the UNTEMPERED target at d_x (mfm_mala_init at beta = 1
 *      returns it).  beta is the reference's argument (default -1/2).  Synchronise; results on the host. ---- */
int mfm_stein_disc(mfm_ctx* ctx, const float* d_x, const float* d_grad, int n, double beta, double h_u_v[2]);
int mfm_max_mean_disc(mfm_ctx* ctx, const float* d_x, const float* d_y, int m, double* h_out);

/* ---- draws of the coming iterations, produced ahead of time (noise.hip) -------------------------------------------------
 * Slot j holds the Gaussian / uniform draws of the MALA step keyed h_keys_gn[j] and of the flow-matching batch keyed
 * h_keys_step[j] (uint32 [n_slots][2] each, the keys later passed to mfm_mala_step / mfm_fm_loss_grad).  The request arms
 * the NEXT mfm_flow_step: the workgroups of its kernel whose tile of chains has finished produce the draws until the slowest
 * tile is done (the launch lasts as long as its slowest chain), so the work rides in otherwise idle CU time;
 * mfm_mala_step / mfm_fm_loss_grad then recognise their key and read the stored draws (the values they would draw in line:
 * results are bit-identical).  Call it right BEFORE mfm_flow_step with the keys of the following K iterations.  Served by the
 * shape-specialised flow-step kernel (headline network shape, PhiFour, --hutch); otherwise EUNSUPPORTED and nothing changes.
 * mfm_noise_drop forgets armed / stored draws (the kernels draw in line again). */
int mfm_noise_prefetch(mfm_ctx* ctx, int n_slots, const uint32_t* h_keys_gn, const uint32_t* h_keys_step);
int mfm_noise_drop(mfm_ctx* ctx);

/* ---- adaptive tempered SMC baseline on the same MALA kernels (exe_others.py:79-111 -> bblackjax/smc) ----------------
 * mfm_smc_delta   : ess.ess_solver + solver.dichotomy (ess.py:46-89, solver.py:20-82) on n log-likelihoods, clipped to
 *                   [0, max_delta] (adaptive_tempered.py:61-72); synchronises, result on the host.
 * mfm_smc_weights : weights = softmax(delta * loglik) (float64, [n]) and log normalising constant (base.py:125-128).
 * mfm_smc_resample: resampling.systematic (resampling.py:50-52,124-135); d_scratch: n doubles (the cumulative sum).
 * mfm_gather_rows : particles[idx] (base.py:120); dst != src. */
int mfm_smc_delta(mfm_ctx* ctx, const double* d_loglik, int n, double target_ess, double max_delta, double* h_delta);
int mfm_smc_weights(mfm_ctx* ctx, const double* d_loglik, int n, double delta, double* d_weights, double* h_lognorm);
int mfm_smc_resample(mfm_ctx* ctx, uint32_t key0, uint32_t key1, const double* d_weights, int n, double* d_scratch,
                     int32_t* d_idx);
/* the other cumulative-sum schemes of resampling.py: stratified (:55-57) and multinomial (:60-80, sorted uniforms);
 * d_scratch: 2 n + 2 doubles.  (residual, :83-121, is composed on the host from the multinomial one: mfm_amd/bblackjax/smc/resampling.py) */
#define MFM_RESAMPLE_SYSTEMATIC 0
#define MFM_RESAMPLE_STRATIFIED 1
#define MFM_RESAMPLE_MULTINOMIAL 2
int mfm_smc_resample_scheme(mfm_ctx* ctx, int scheme, uint32_t key0, uint32_t key1, const double* d_weights, int n,
                            double* d_scratch, int32_t* d_idx);
int mfm_gather_rows(mfm_ctx* ctx, const float* d_src, const int32_t* d_idx, int n, int dim, float* d_dst);
/* ---- N1: self-normalised importance resampling of the final flow samples (exe_flow_matching.py:458-459):
 *      d_idx[j] = jax.random.choice(key, n, (m,), p = exp(d_logw - max d_logw))[j]; d_scratch: n doubles (the cumulative sum,
 *      taken in index order).  Combine with mfm_gather_rows. ---- */
int mfm_choice_logw(mfm_ctx* ctx, uint32_t key0, uint32_t key1, const double* d_logw, int n, int m, double* d_scratch,
                    int32_t* d_idx);
/* d_out[0..1] = sum and sum of squares (float64) of d_x[0..n): the per-iteration acceptance statistics
 * (exe_flow_matching.py:442-443: infos.acceptance_rate.mean() / .std()) without a host round trip */
int mfm_acc_stats(mfm_ctx* ctx, const float* d_x, int n, double* d_out);

/* ---- algorithmic counters (the figures roofline numbers are computed from) ----------------------
 * h_out[0] MALA chain-steps, [1] flow-matching training samples, [2] flow-matching evaluation samples (mfm_fm_loss),
 * [3] Dopri5 solves (a flow step is two per chain), [4] attempted Dopri5 steps over all solves (summed on the device),
 * [5] vector-field evaluations = 2 * [3] + 6 * [4] (odeint: f(y0), the initial-step probe, six stages per attempt),
 * [6] optimizer steps, [7] algorithmic HBM bytes of the MALA steps (4 (5 dim + 5) per chain-step).  Synchronises. */
int mfm_get_counters(mfm_ctx* ctx, int64_t h_out[8]);
int mfm_reset_counters(mfm_ctx* ctx);

/* ---- parity instrumentation: Dormand-Prince on a PRESCRIBED step sequence -------------------------------------------------
 * Arms the NEXT mfm_ode_transform (one solve per sample) or mfm_flow_step (two: 0 inverse, 1 forward) on this context: the
 * solver takes d_dt[j] as the step size of attempt j (j = 0: the initial step) and d_acc[j] as its accept decision instead
 * of its controller's, and records what the controller computed: d_ratio[j] = error ratio of attempt j, d_dt_own[0] = its
 * initial step, d_dt_own[j + 1] = the step it proposed after attempt j.  Element (solve s, sample r, attempt j) of each
 * array is at ((s * n + r) * cap + j), n = samples of the armed call; a zero d_dt entry ends the solve.  Lets two
 * implementations be compared stage for stage on the same step sequence (tests/test_gpu_replay.py); no counterpart in the
 * reference.  d_diag (may be NULL; flow step only): float64 [n][4] = {inverse log-det, forward log-det, tempered log-density
 * at the proposal, log acceptance ratio}, the terms of exe_flow_matching.py:271-274.  d_dt == NULL disarms.  Both kernel
 * families (the wide family's host-driven solver takes the same hooks in its row kernels). */
int mfm_debug_replay(mfm_ctx* ctx, int cap, const float* d_dt, const uint8_t* d_acc, float* d_ratio, float* d_dt_own,
                     double* d_diag);

/* class_mask: 0 = off; otherwise bit c enables class c (-1: all) and the record is reset.  Two event records per launch
 * cost ~6 us of stream time each side on this runtime, so a caller that is itself being timed enables only the class it
 * needs (bench.py: the flow step during the timed region, every class in a separate instrumented pass). */
int mfm_profile(mfm_ctx* ctx, int class_mask);
int mfm_profile_read(mfm_ctx* ctx, double ms_total[8], int64_t launches[8]);   /* synchronises */

/* ---- test helpers (host only, no GPU needed) ------------------------------------------------------------------- */
int mfm_pack_index(int k, int n, int KB);       /* float index of W[k][n] inside a packed layer */
int mfm_pack_index_T(int k, int n, int NB);
int mfm_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]);

#ifdef __cplusplus
}
#endif
#endif /* MFM_H */
