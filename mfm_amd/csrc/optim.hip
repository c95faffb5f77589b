// K7: apply_if_finite(chain(adamw(mask = not-bias), clip(+-c)), 10) as a check/decide kernel + ONE elementwise kernel that also re-emits the
// MFMA-operand-ready packed weights (forward and transposed) the MLP kernels stream.
//
// Replaces state.apply_gradients at exe_flow_matching.py:366 with the optimizer of :129-137,184 (optax 0.1.9 rules:
// SURVEY.md section 8c).  Quirks kept: the clip acts on the final UPDATE (Q5); the schedule value is
// lr * (1 - count / learning_iter) at the inner optimizer's pre-increment count (Q6), which only advances on accepted
// updates; a non-finite gradient zeroes the update and leaves the inner state untouched, up to 10 in a row.
//
// The finite check and the step counters live on the device (OptState) so an iteration needs no host sync.
#include "mlp.hip.h"

struct OptState {        // device-resident scalars
  int step;              // TrainState.step (every call)
  int count;             // inner adam / schedule count (accepted updates only)
  int notfinite_count;   // consecutive non-finite gradients
  int last_applied;      // 1 if the last call changed the parameters
  float last_lr;         // learning_rate_fn(state.step) as logged at :367 (pre-increment step)
  int bc_for;            // wgrad_sk.hip: bc1 / bc2 below are Adam's bias corrections 1 - b^bc_for (0: not computed; they depend on nothing else,
  float bc1, bc2;        // so a writer that leaves them alone leaves them valid or visibly stale)
};

struct AdamArgs {
  NetDev net;
  const float* grads; int n_slabs;     // grads = sum over n_slabs slabs of n_params floats
  float* master; float* mu; float* nu; // canonical flat layout
  float* Wp; float* WpT; float* bias;  // packed outputs
  OptState* st;
  int* flag;                           // scratch: [0] non-finite flag, [1] / [2] arrival tickets of the two kernels (self-resetting)
  double lr0; int learning_iter, warmup;
  double b1, b2; float eps, wd, clip;
  int max_err;
  int inline_decide;                   // the non-finite flag was raised by the kernel that produced `grads`: decide here
};

// apply_if_finite decision + step scalars, identical in every thread.  With inline_decide every thread derives the decision
// from the flag and the (not yet advanced) state; the last workgroup commits the state once all have read it.
struct AdamDecision { bool apply; int count; int nf_new; };
struct AdamRaw { int flag0, nf, last_applied, count; };          // the state words as loaded: no use, hence no wait, yet
__device__ __forceinline__ AdamRaw adam_load(const AdamArgs& a) {
  AdamRaw r;
  r.count = a.st->count; r.nf = a.st->notfinite_count; r.last_applied = a.st->last_applied;
  r.flag0 = a.inline_decide ? a.flag[0] : 0;
  return r;
}
__device__ __forceinline__ AdamDecision adam_resolve(const AdamArgs& a, const AdamRaw& r) {
  AdamDecision d;
  d.count = r.count;
  if (a.inline_decide) {
    const bool finite = r.flag0 == 0;
    d.nf_new = finite ? 0 : r.nf + 1;
    d.apply = finite || d.nf_new > a.max_err;
  } else {
    d.nf_new = r.nf;
    d.apply = r.last_applied != 0;
  }
  return d;
}
__device__ __forceinline__ AdamDecision adam_decide(const AdamArgs& a) { return adam_resolve(a, adam_load(a)); }

__device__ __forceinline__ float lr_schedule(double lr0, int learning_iter, int warmup, int count) {
  // join_schedules([linear(0 -> lr, warmup), linear(lr -> 0, learning_iter - warmup)], [warmup]) (:189-198)
  if (warmup > 0 && count < warmup) return (float)(lr0 * ((double)count / (double)warmup));
  const int ts = learning_iter - warmup;
  if (ts <= 0) return (float)lr0;
  int cc = count - warmup; cc = cc < 0 ? 0 : (cc > ts ? ts : cc);
  return (float)(lr0 * (1.0 - (double)cc / (double)ts));
}

// Finite check over the (summed) gradient + the apply_if_finite decision.  Every thread of the update kernel must see the
// same decision, so the decision is taken by the LAST workgroup of this kernel to arrive (self-resetting ticket): one
// launch instead of a check kernel and a single-thread decision kernel.
__global__ void finite_decide_kernel(const float* grads, int n_slabs, int n, OptState* st, int* flag, double lr0, int learning_iter,
                                     int warmup, int max_err) {
  bool bad = false;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {     // few workgroups: one ticket each
    float s = 0.f;
    for (int k = 0; k < n_slabs; ++k) s += grads[(size_t)k * n + p];
    bad |= !isfinite(s);
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (threadIdx.x == 0) {
    // device-scope atomics are performed at the memory side; the ticket is issued only after the flag update has RETURNED
    // (data dependence), so the last arrival sees every flag update without an L2 write-back fence (3.5 us each)
    unsigned int dep = any_bad ? (unsigned int)atomicOr(flag, 1) : 0u;
    if (atomicInc(reinterpret_cast<unsigned int*>(flag + 1) + (dep & 0x40000000u), gridDim.x - 1) == gridDim.x - 1) {      // last arrival
      const bool finite = atomicOr(flag, 0) == 0;
      st->notfinite_count = finite ? 0 : st->notfinite_count + 1;
      st->last_lr = lr_schedule(lr0, learning_iter, warmup, st->step);
      st->step += 1;
      st->last_applied = (finite || st->notfinite_count > max_err) ? 1 : 0;
      atomicExch(flag, 0);
    }
  }
}

__device__ __forceinline__ void adamw_element(const AdamArgs& a, int p, bool apply, float bc1, float bc2, float lr);

// One parameter's AdamW + clip update (optax 0.1.9: scale_by_adam, add_decayed_weights on kernels, scale by -lr, clip).  Every
// multiply-add is an explicit fused operation: the three update kernels (scalar, 4 x 4 blocks, reduction + update) inline this
// one function and must round alike -- left to the compiler's contraction they need not (targets.hip.h: phi4_grad, round 3).
__device__ __forceinline__ float adam_update(float w, float g, float& m, float& v, float b1, float b2, float bc1, float bc2, float eps, float wd,
                                             bool decay, float lr, float clip) {
  m = __builtin_fmaf(b1, m, (1.f - b1) * g);
  v = __builtin_fmaf(b2, v, (1.f - b2) * g * g);
  float u = (m / bc1) / (sqrtf(v / bc2) + eps);
  if (decay) u = __builtin_fmaf(wd, w, u);
  u = fminf(fmaxf(-lr * u, -clip), clip);
  return w + u;
}

// Last workgroup of an update kernel: advance the inner count, and with inline_decide the bookkeeping finite_decide_kernel
// would have done (every workgroup has read the old state by the time the last ticket is drawn).
__device__ __forceinline__ void adam_commit(const AdamArgs& a, const AdamDecision& d) {
  if (atomicInc(reinterpret_cast<unsigned int*>(a.flag + 2), gridDim.x - 1) != gridDim.x - 1) return;
  if (a.inline_decide) {
    a.st->notfinite_count = d.nf_new;
    a.st->last_lr = lr_schedule(a.lr0, a.learning_iter, a.warmup, a.st->step);
    a.st->step += 1;
    a.st->last_applied = d.apply ? 1 : 0;
    atomicExch(a.flag, 0);
  }
  if (d.apply) a.st->count = d.count + 1;
}

// The update + repack; the LAST workgroup to finish advances the inner optimizer count (every workgroup has read it by
// then), which used to be a kernel of its own.
__global__ void adamw_kernel(AdamArgs a) {
  const AdamDecision dec = adam_decide(a);
  const bool apply = dec.apply;
  const int count = dec.count;
  // per-step scalars once per thread (two float64 pow calls used to be evaluated per parameter)
  const int c1 = count + 1;
  const float bc1 = (float)(1.0 - pow(a.b1, (double)c1)), bc2 = (float)(1.0 - pow(a.b2, (double)c1));
  const float lr = lr_schedule(a.lr0, a.learning_iter, a.warmup, count);
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.net.n_params; p += gridDim.x * blockDim.x) adamw_element(a, p, apply, bc1, bc2, lr);
  __syncthreads();
  if (threadIdx.x == 0) adam_commit(a, dec);
}

// `gsum_in`: the element's (summed) gradient when the caller already holds it; otherwise it is summed from a.grads here
template <bool HAVE_G>
__device__ __forceinline__ void adamw_element_t(const AdamArgs& a, int p, bool apply, float bc1, float bc2, float lr, float gsum_in) {
  const NetDev& n = a.net;
  float w = a.master[p];
  // locate (layer, kernel/bias, k, nn)
  int layer = 0;
#pragma unroll
  for (int l = 1; l < MLP_MAXL; ++l) if (p >= n.L[l].m_w) layer = l;
  const LayerDesc& ld = n.L[layer];
  const bool is_bias = p >= ld.m_b;
  if (apply) {
    float gsum = gsum_in;
    if constexpr (!HAVE_G) {
      gsum = 0.f;
      for (int k = 0; k < a.n_slabs; ++k) gsum += a.grads[(size_t)k * n.n_params + p];
    }
    float m = a.mu[p], v = a.nu[p];
    w = adam_update(w, gsum, m, v, (float)a.b1, (float)a.b2, bc1, bc2, a.eps, a.wd, !is_bias, lr, a.clip);
    a.mu[p] = m; a.nu[p] = v;
    a.master[p] = w;
  }
  // re-emit packed copies (also on rejected updates: cheap, keeps the kernel branch-free for the packer)
  if (is_bias) {
    a.bias[ld.b_off + (p - ld.m_b)] = w;
  } else {
    const int e = p - ld.m_w, k = e / ld.N, nn = e - k * ld.N;
    const int kk = packed_row(n, layer, k);
    a.Wp[ld.w_off + pack_index(kk, nn, ld.Kp / 16)] = w;
    a.WpT[ld.w_off + pack_index_T(kk, nn, ld.Np / 16)] = w;
  }
}

__device__ __forceinline__ void adamw_element(const AdamArgs& a, int p, bool apply, float bc1, float bc2, float lr) {
  adamw_element_t<false>(a, p, apply, bc1, bc2, lr, 0.f);
}

// ---- one rank, one launch: slab reduction + apply_if_finite decision + AdamW (mfm_train_iter) ---------------------------------
// reduce_slabs_kernel and the update kernel as ONE kernel, one parameter per thread.  What kept them apart is the decision:
// apply_if_finite skips the WHOLE update when ANY element of the gradient is non-finite, which a kernel that sums and updates
// element by element only knows after a grid-wide exchange (tried in round 3 as an arrival ticket inside the update kernel: 26 us
// against 11.7 + 7.1).  Here the weight-gradient kernel raises flag[0] when one of its PARTIAL sums (one of `n_slabs` slices of
// the chain axis) is non-finite or exceeds FLT_MAX / n_slabs in magnitude: with flag[0] == 0 every total is a sum of n_slabs
// finite terms that cannot overflow, i.e. finite, and the update is applied without any exchange.  With flag[0] != 0 (gradients
// beyond 1e37: practically a diverged run) the kernel decides on the TOTALS exactly as before, through a grid-wide arrival
// counter -- every workgroup of this grid is resident at once (one parameter per thread: a few hundred workgroups), so the
// wait cannot deadlock.  The optimizer state is double-buffered (read `st`, block 0 writes `st_next`): no workgroup has to be the
// last to read it.  Also totals the loss partials of the training kernel (last block), like reduce_slabs_kernel.
__global__ __launch_bounds__(256) void reduce_adamw_kernel(AdamArgs a, OptState* st_next, float* out, const double* loss_part, int n_part, double* loss_out,
                                                           int force_exchange) {
  __shared__ int sh_finite;
  const int n = a.net.n_params, p = blockIdx.x * 256 + threadIdx.x;
  const bool live = p < n;
  const OptState st = *a.st;
  const int suspicious = a.flag[0] | force_exchange;
  float s = 0.f;
  if (live) {
    for (int k = 0; k < a.n_slabs; ++k) s += a.grads[(size_t)k * n + p];
    out[p] = s;
  }
  bool finite = true;
  if (suspicious) {                            // uniform over the grid
    const int any = __syncthreads_or(live && !isfinite(s) ? 1 : 0);
    if (threadIdx.x == 0) {
      // the arrival is counted only after this workgroup's flag update has RETURNED (data dependence), and device-scope
      // atomics are performed at the memory side: the last arrival sees every update
      const unsigned int dep = any ? (unsigned int)atomicOr(a.flag + 3, 1) : 0u;
      atomicAdd(a.flag + 4, 1 + (int)(dep & 0x40000000u));
      while (atomicAdd(a.flag + 4, 0) < (int)gridDim.x) __builtin_amdgcn_s_sleep(8);
      sh_finite = atomicOr(a.flag + 3, 0) == 0;
    }
    __syncthreads();
    finite = sh_finite != 0;
  }
  const int nf_new = finite ? 0 : st.notfinite_count + 1;
  const bool apply = finite || nf_new > a.max_err;
  const int count = st.count, c1 = count + 1;
  const float bc1 = (float)(1.0 - pow(a.b1, (double)c1)), bc2 = (float)(1.0 - pow(a.b2, (double)c1));
  const float lr = lr_schedule(a.lr0, a.learning_iter, a.warmup, count);
  if (live) adamw_element_t<true>(a, p, apply, bc1, bc2, lr, s);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    OptState nx;
    nx.notfinite_count = nf_new;
    nx.last_lr = lr_schedule(a.lr0, a.learning_iter, a.warmup, st.step);
    nx.step = st.step + 1;
    nx.last_applied = apply ? 1 : 0;
    nx.count = apply ? count + 1 : count;
    *st_next = nx;
  }
  if (loss_part && blockIdx.x == gridDim.x - 1) {
    __shared__ double sm[256];
    double t = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) t += loss_part[i];
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = sm[0];
  }
}
void launch_reduce_adamw(const AdamArgs& a, OptState* st_next, float* out, const double* loss_part, int n_part, double* loss_out, int force_exchange, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_adamw_kernel, dim3((a.net.n_params + 255) / 256), dim3(256), 0, stream, a, st_next, out, loss_part, n_part, loss_out, force_exchange);
}

// The same update for networks whose every kernel has in / out widths that are multiples of 4 (all of the reference's
// examples): one thread per 4 x 4 block of a kernel matrix.  The canonical rows (4 consecutive `out`), the forward packing
// (4 consecutive `in` at one `out`: one B fragment element run) and the transposed packing (4 consecutive `out` at one `in`)
// are then all 128-bit accesses, where adamw_kernel scatters three 4-byte stores per parameter -- that matters for the
// pines widths (8.65 M parameters, 34.6 MB per copy).  Biases are handled one element per thread after the blocks.
struct AdamBlocks { int first[MLP_MAXL + 1]; int n_blocks; int n_bias_items; };
__global__ __launch_bounds__(256) void adamw_vec_kernel(AdamArgs a, AdamBlocks bl) {
  const NetDev& n = a.net;
  const AdamRaw raw = adam_load(a);                 // state reads issued here, first used after the operand loads below
  const float b1 = (float)a.b1, b2 = (float)a.b2;
  const int total = bl.n_blocks + bl.n_bias_items;
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    if (it >= bl.n_blocks) {                       // a bias element: scalar path
      const AdamDecision dec = adam_resolve(a, raw);
      const bool apply = dec.apply;
      const int count = dec.count, c1 = count + 1;
      const float bc1 = (float)(1.0 - pow(a.b1, (double)c1)), bc2 = (float)(1.0 - pow(a.b2, (double)c1));
      const float lr = lr_schedule(a.lr0, a.learning_iter, a.warmup, count);
      int p = it - bl.n_blocks, layer = 0;
      for (int l = 0; l < MLP_MAXL; ++l) { if (p < n.L[l].N) { layer = l; break; } p -= n.L[l].N; }
      adamw_element(a, n.L[layer].m_b + p, apply, bc1, bc2, lr);
      continue;
    }
    int layer = 0;
#pragma unroll
    for (int l = 1; l < MLP_MAXL; ++l) if (it >= bl.first[l]) layer = l;
    const LayerDesc& ld = n.L[layer];
    const int e = it - bl.first[layer], nb4 = ld.N >> 2, kb4 = e / nb4, k0 = 4 * kb4, n0 = 4 * (e - kb4 * nb4);
    // every operand is loaded unconditionally and up front: the loads then fly together with the optimizer-state reads the
    // decision above waits for (a rejected update -- rare -- reads moments it does not use)
    f32x4 w[4], g4[4], m4[4], v4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t p = (size_t)ld.m_w + (size_t)(k0 + r) * ld.N + n0;
      w[r] = *reinterpret_cast<const f32x4*>(a.master + p);
      g4[r] = *reinterpret_cast<const f32x4*>(a.grads + p);
      m4[r] = *reinterpret_cast<const f32x4*>(a.mu + p); v4[r] = *reinterpret_cast<const f32x4*>(a.nu + p);
    }
    __builtin_amdgcn_sched_barrier(0);             // keep the loads above the first use of the decision
    const AdamDecision dec = adam_resolve(a, raw);
    const bool apply = dec.apply;
    const int count = dec.count, c1 = count + 1;
    const float bc1 = (float)(1.0 - pow(a.b1, (double)c1)), bc2 = (float)(1.0 - pow(a.b2, (double)c1));
    const float lr = lr_schedule(a.lr0, a.learning_iter, a.warmup, count);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t p = (size_t)ld.m_w + (size_t)(k0 + r) * ld.N + n0;
      if (apply) {
        f32x4 g = g4[r];
        for (int sl = 1; sl < a.n_slabs; ++sl) g += *reinterpret_cast<const f32x4*>(a.grads + (size_t)sl * n.n_params + p);
        f32x4 m = m4[r], v = v4[r];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float mj = m[j], vj = v[j];
          w[r][j] = adam_update(w[r][j], g[j], mj, vj, b1, b2, bc1, bc2, a.eps, a.wd, true, lr, a.clip);
          m[j] = mj; v[j] = vj;
        }
        *reinterpret_cast<f32x4*>(a.mu + p) = m; *reinterpret_cast<f32x4*>(a.nu + p) = v;
        *reinterpret_cast<f32x4*>(a.master + p) = w[r];
      }
      // transposed packing: 4 consecutive `out` at in = k0 + r
      *reinterpret_cast<f32x4*>(a.WpT + ld.w_off + pack_index_T(k0 + r, n0, ld.Np / 16)) = w[r];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)                    // forward packing: 4 consecutive `in` at out = n0 + j
      *reinterpret_cast<f32x4*>(a.Wp + ld.w_off + pack_index(k0, n0 + j, ld.Kp / 16)) = f32x4{w[0][j], w[1][j], w[2][j], w[3][j]};
  }
  __syncthreads();
  if (threadIdx.x == 0) adam_commit(a, adam_resolve(a, raw));
}

void launch_adamw(const AdamArgs& a, hipStream_t stream) {
  const int n = a.net.n_params;
  const int nb = (n + 255) / 256;
  // a ticket atomic per workgroup: ~11 ns each on one word, keep them few -- unless the network is large enough (the pines
  // widths: 8.65 M parameters) for the elementwise pass itself to need the whole chip
  const int cap = n > (1 << 21) ? 2048 : 256;
  dim3 grid(nb < cap ? nb : cap), block(256);
  if (!a.inline_decide) {
    // flag[0] may still hold the verdict of an earlier mfm_fm_loss_grad whose gradient was never applied (another buffer was
    // passed here): the check below must start from a clean flag
    (void)hipMemsetAsync(a.flag, 0, sizeof(int), stream);
  }
  if (!a.inline_decide)
    hipLaunchKernelGGL(finite_decide_kernel, grid, block, 0, stream, a.grads, a.n_slabs, n, a.st, a.flag, a.lr0, a.learning_iter, a.warmup, a.max_err);
  bool vec = true;
  AdamBlocks bl; memset(&bl, 0, sizeof bl);
  for (int l = 0; l < MLP_MAXL; ++l) {
    vec &= (a.net.L[l].K % 4 == 0) && (a.net.L[l].N % 4 == 0);
    bl.first[l] = bl.n_blocks; bl.n_blocks += (a.net.L[l].K / 4) * (a.net.L[l].N / 4); bl.n_bias_items += a.net.L[l].N;
  }
  bl.first[MLP_MAXL] = bl.n_blocks;
  vec &= packed_row(a.net, a.net.nT + a.net.nX + 1, a.net.L[a.net.nT + a.net.nX + 1].K - 1) == a.net.L[a.net.nT + a.net.nX + 1].K - 1;      // no row remap (mlp.hip.h)
  if (vec) {
    const int items = bl.n_blocks + bl.n_bias_items, nbv = (items + 255) / 256;
    hipLaunchKernelGGL(adamw_vec_kernel, dim3(nbv < cap ? nbv : cap), block, 0, stream, a, bl);
  } else {
    hipLaunchKernelGGL(adamw_kernel, grid, block, 0, stream, a);
  }
}

// pack only (after mfm_set_params)
__global__ void pack_kernel(NetDev n, const float* master, float* Wp, float* WpT, float* bias) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n.n_params) return;
  int layer = 0;
#pragma unroll
  for (int l = 1; l < MLP_MAXL; ++l) if (p >= n.L[l].m_w) layer = l;
  const LayerDesc& ld = n.L[layer];
  const float w = master[p];
  if (p >= ld.m_b) bias[ld.b_off + (p - ld.m_b)] = w;
  else {
    const int e = p - ld.m_w, k = e / ld.N, nn = e - k * ld.N;
    const int kk = packed_row(n, layer, k);
    Wp[ld.w_off + pack_index(kk, nn, ld.Kp / 16)] = w;
    WpT[ld.w_off + pack_index_T(kk, nn, ld.Np / 16)] = w;
  }
}
void launch_pack(const NetDev& n, const float* master, float* Wp, float* WpT, float* bias, hipStream_t stream) {
  hipLaunchKernelGGL(pack_kernel, dim3((n.n_params + 255) / 256), dim3(256), 0, stream, n, master, Wp, WpT, bias);
}
