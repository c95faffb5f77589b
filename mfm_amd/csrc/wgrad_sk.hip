// K4b + K7 in ONE launch: the weight gradients dW = A^T dZ, db = sum dZ of the fused tile family (exe_flow_matching.py:364-365, the
// parameter half of jax.value_and_grad) as a STREAM-K GEMM over the chain axis, and -- on one rank -- the optimizer step of
// state.apply_gradients (:366; optax chain of :129-137,184: optim.hip) applied by the LAST workgroup to arrive at each 64 x 64 block.
//
// Why a second form of wgrad_kernel (fm.hip).  That kernel gives every 64 x 64 block of dW to `split` = 8 workgroups, one per slice
// of the chain axis: 52 blocks x 8 = 416 workgroups on 256 CUs, so 160 CUs carry two and the launch lasts 2 x 32 chain tiles of MFMAs
// where a balanced one would last 2 x 26; each workgroup keeps ONE stage of operands in flight (its loads come from beyond L2: the
// packed activations were written by the training kernel a launch ago), and a third launch then sums the eight slabs and runs AdamW.
// Here:
//  * the work is cut into UNITS (block, chain tile) -- 52 x 256 = 13,312 at the headline shape -- and dealt out evenly and
//    contiguously: 512 workgroups x 26 units, two workgroups per CU on every CU.  A workgroup's range crosses at most one block
//    boundary (its range is never longer than a block's chain axis), so it produces at most two partial blocks;
//  * operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: one packed 16 x 16 tile = 1 KB = one wave instruction,
//    no staging registers, no ds_write pass) into a ring of WSK_NS stages of WSK_TB chain tiles, WSK_NS - 1 stages in flight
//    behind a COUNTED vmcnt and a raw s_barrier (a __syncthreads() would drain the ring: cdna_hip_programming.md section 5);
//  * partial blocks are published in ACCUMULATOR layout (one float4 per lane and MFMA tile: coalesced) with write-through (sc1)
//    stores, each storing wave drains its stores, the workgroup meets at a barrier and ONE lane adds to the block's arrival
//    counter; the workgroup whose add came last reads the block's partials back with sc1 loads, sums them in a FIXED order
//    (deterministic: no float atomics) and
//      - writes the summed gradient in the canonical layout (a multi-rank host all-reduces it), and
//      - with `fuse`: runs apply_if_finite(adamw, clip) on the block's 4,096 parameters and re-emits their packed copies --
//        in this layout a lane's accumulator IS one float4 of the forward pack.
//    No agent-scope release / acquire fences (an L2 write-back per workgroup: the round-3 attempt at this seam cost 50-80 us);
//    the hand-off is MI355X_MICROARCH.md's "stores all sc1, one lane's agent-scope add per storing workgroup, the workgroup whose
//    add came last loads with sc1" row.
//  * apply_if_finite needs "is ANY element of the gradient non-finite" before the FIRST parameter is touched.  The training kernel
//    raises a flag when a value it stores for this kernel (activation or pre-activation gradient) is not <= 1e15 in magnitude (NaN
//    included): with the flag clear every partial and every total is a sum of at most 2^20 products below 1e30, i.e. finite, and the
//    blocks update independently.  With the flag raised (a diverged run) no block updates on its own: every last arriver leaves its
//    totals in the canonical gradient (write-through), reports whether they are finite and draws a second ticket; the workgroup
//    that draws the LAST of those knows the verdict on the whole gradient and runs the optimizer over all parameters itself -- slow
//    (one workgroup), rare, and free of any wait: no workgroup of this kernel ever spins on another, so nothing here depends on
//    how many of them the device holds at once.
#include "mlp.hip.h"

#ifdef MFM_WSK_STAMPS
__device__ unsigned long long* g_wsk_dbg = nullptr;      // [WG][16] section time stamps (development build only)
#define WSK_STAMP(id) do { if (g_wsk_dbg && threadIdx.x == 0) g_wsk_dbg[blockIdx.x * 16 + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSK_STAMP(id) do {} while (0)
#endif
#define WSK_MAXJOBS 64
#ifndef WSK_NS
#define WSK_NS 4          // LDS stages
#endif
#ifndef WSK_TB
#define WSK_TB 2          // chain tiles per stage
#endif
constexpr int WSK_PSZ = 4 * 4 * 64 * 4 + 64;      // floats per partial block: [wave][tile of the wave's quadrant][lane] float4 + 64 bias sums
constexpr float WSK_SAFE = 1.0e15f;               // |activation|, |dZ| bound under which no sum of <= 2^20 products can overflow

struct WskOpt {            // the optimizer half (fuse != 0)
  float *master, *mu, *nu, *Wp, *WpT, *bias;
  const OptState* st; OptState* st_next;
  int* flag;               // [3] / [4]: non-finite flag and arrival counter of the exchange among the last arrivers (cleared by the training kernel)
  const int* suspicious;   // the training kernel's "a stored value was huge or NaN" word for THIS iteration
  double lr0; int learning_iter, warmup;
  double b1, b2; float eps, wd, clip;
  int max_err, force_exchange;
};

struct WskArgs {
  NetDev net;
  WsLayout ws;
  const float* acts; const float* dzs;
  int nbb, n_jobs, G;      // chain tiles, 64 x 64 blocks, workgroups
  int upw_q, upw_r;        // units per workgroup: the first upw_r take upw_q + 1, the others upw_q
  int xcd_remap;           // != 0: consecutive unit ranges go to workgroups of ONE XCD (blockIdx.x % 8 names the workgroups that share one)
  float* partials;         // [2 G][WSK_PSZ]
  int* tickets;            // [n_jobs] arrival counters, zero between launches (the last arriver of a block resets its own)
  float* out;              // [n_params] the summed gradient, canonical layout (null: not wanted)
  int* bad;                // non-null (fuse == 0): raised when a TOTAL is non-finite (the verdict mfm_adamw_step reuses on one rank)
  const double* loss_part; int n_part; double* loss_out;      // non-null: workgroup 0 totals the training kernel's loss partials
  int fuse;
  WskOpt opt;
  WgradJob jobs[WSK_MAXJOBS];
};

__device__ __forceinline__ int wsk_start(const WskArgs& a, int w) { return w * a.upw_q + (w < a.upw_r ? w : a.upw_r); }
__device__ __forceinline__ int wsk_wg_of(const WskArgs& a, int u) {
  const int big = a.upw_r * (a.upw_q + 1);
  return u < big ? u / (a.upw_q + 1) : a.upw_r + (u - big) / a.upw_q;
}

typedef unsigned int wsk_u32x4 __attribute__((ext_vector_type(4)));
constexpr int WSK_SC1 = 16;      // cache-policy bit of the raw buffer intrinsics that sets sc1 on gfx940+ (write-through store / L1-bypassing load)

__global__ __launch_bounds__(256) void wgrad_sk_kernel(WskArgs a) {
  // ONE __shared__ object: with a second one beside the LDS-DMA ring hipcc waits vmcnt(0) before the first ds_read of every stage
  // and the ring drains (cdna_hip_programming.md section 5, trap 4(a)); the three flag words live behind the ring
  __shared__ f32x4 sh_all[WSK_NS * WSK_TB * 8 * 64 + 1];
  f32x4 (*sh)[WSK_TB][8][64] = reinterpret_cast<f32x4 (*)[WSK_TB][8][64]>(sh_all);      // [stage][chain tile][A0..A3, Z0..Z3][lane]
  int* const sh_last = reinterpret_cast<int*>(sh_all + WSK_NS * WSK_TB * 8 * 64);         // [2]
  int& sh_fin = sh_last[2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const int wk = wave >> 1, wn = wave & 1;
  const NetDev& n = a.net;
  // workgroup -> unit range
  int w = blockIdx.x;
  if (a.xcd_remap) {           // bijective for any G (cdna_hip_programming.md, XCD swizzle): the blockIdx.x % 8 class x gets a contiguous run of w
    const int q = a.G >> 3, r = a.G & 7, x = w & 7, i = w >> 3;
    w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  WSK_STAMP(0);
  const int u0 = wsk_start(a, w), cnt = a.upw_q + (w < a.upw_r ? 1 : 0);
  const int j0 = u0 / a.nbb, bb0 = u0 - j0 * a.nbb;
  const int n0 = cnt < a.nbb - bb0 ? cnt : a.nbb - bb0;              // tiles of the first segment (block j0); the rest belong to block j0 + 1
  const int nseg = n0 < cnt ? 2 : 1;
  // optimizer scalars and flags: read here, first used after the main loop
  OptState st0; int sus = 0;
  if (a.fuse) { st0 = *a.opt.st; sus = *a.opt.suspicious | a.opt.force_exchange; }

  // per segment: the tiles this wave FETCHES (A tile `wave`, dZ tile `wave` of the block; clamped at the layer's edge: fetched, never used)
  const f32x4* Af[2]; const f32x4* Zf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = j0 + s < a.n_jobs ? j0 + s : j0;
    const WgradJob J = a.jobs[j];
    const LayerDesc& ld = n.L[J.layer];
    const int KT = ld.Kp / 16, NT = ld.Np / 16;
    const int kt_f = J.kt0 + wave < KT ? J.kt0 + wave : J.kt0, nt_f = J.nt0 + wave < NT ? J.nt0 + wave : J.nt0;
    Af[s] = reinterpret_cast<const f32x4*>(a.acts) + (size_t)wgrad_a_tile(n, a.ws, J.layer, kt_f) * a.nbb * 64 + lane;
    Zf[s] = reinterpret_cast<const f32x4*>(a.dzs) + (size_t)wgrad_z_tile(a.ws, J.layer, nt_f) * a.nbb * 64 + lane;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  auto issue = [&](int stg) {      // the WSK_TB chain tiles of stage `stg` (tiles past the range re-fetch the last one: the count stays static)
    const int slot = stg % WSK_NS;
#pragma unroll
    for (int u = 0; u < WSK_TB; ++u) {
      int i = stg * WSK_TB + u; i = i < cnt ? i : cnt - 1;
      const int s = i >= n0 ? 1 : 0;
      const int bb = s ? i - n0 : bb0 + i;
      __builtin_amdgcn_global_load_lds((glb_void*)(Af[s] + (size_t)bb * 64), (lds_void*)&sh[slot][u][wave][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(Zf[s] + (size_t)bb * 64), (lds_void*)&sh[slot][u][4 + wave][0], 16, 0, 0);
    }
  };
  const int nst = (cnt + WSK_TB - 1) / WSK_TB;
#pragma unroll
  for (int s = 0; s < WSK_NS - 1; ++s) issue(s);

  f32x4 acc00 = {0, 0, 0, 0}, acc01 = acc00, acc10 = acc00, acc11 = acc00, bs0 = acc00, bs1 = acc00;
  float* part = a.partials + (size_t)(2 * w) * WSK_PSZ;
  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(a.partials, 0, 0x7fffffff, 0x00020000);
  // publish the accumulators as partial block `seg` of this workgroup (write-through stores) and clear them
  auto publish = [&](int seg) {
    const int base = ((2 * w + seg) * WSK_PSZ) * 4 + ((wave * 4) * 64 + lane) * 16;      // bytes
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wsk_u32x4, acc00), pr, base, 0, WSK_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wsk_u32x4, acc01), pr, base + 1024, 0, WSK_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wsk_u32x4, acc10), pr, base + 2048, 0, WSK_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wsk_u32x4, acc11), pr, base + 3072, 0, WSK_SC1);
    if (wk == 0) {               // bias partials: sum over this workgroup's chains of dZ, n-tiles 2 wn and 2 wn + 1 of the block
      float s0 = bs0[0] + bs0[1] + bs0[2] + bs0[3], s1 = bs1[0] + bs1[1] + bs1[2] + bs1[3];
      s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      if (g == 0) {
        const int bo = ((2 * w + seg) * WSK_PSZ + 4096 + (2 * wn) * 16 + c) * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, s0), pr, bo, 0, WSK_SC1);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, s1), pr, bo + 64, 0, WSK_SC1);
      }
    }
    acc00 = f32x4{0, 0, 0, 0}; acc01 = acc00; acc10 = acc00; acc11 = acc00; bs0 = acc00; bs1 = acc00;
  };
  (void)part;
  const __amdgpu_buffer_rsrc_t outr = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);

  WSK_STAMP(1);
  for (int it = 0; it < nst; ++it) {
    if (it == 1) WSK_STAMP(2);
    // stage `it` has landed for this wave once at most the (WSK_NS - 2) younger stages are outstanding; the barrier makes that true for
    // every wave's part of it -- and says every wave is done READING stage it - 1, whose slot the next issue overwrites
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WSK_NS - 2) * 2 * WSK_TB) : "memory");
    asm volatile("s_barrier" ::: "memory");
    issue(it + WSK_NS - 1);
    const int slot = it % WSK_NS;
#pragma unroll
    for (int u = 0; u < WSK_TB; ++u) {
      const int i = it * WSK_TB + u;
      if (i < cnt) {
        if (i == n0) publish(0);                 // (wave-uniform) the range crosses into its second block here
        // fragment reads by inline assembly: a compiler-visible ds_read of an array that LDS-DMA writes is preceded by s_waitcnt vmcnt(0)
        // (the waitcnt pass cannot tell the stage being read from the stages in flight), which drains the ring every stage
        f32x4 a0, a1, z0, z1;
        const unsigned int la = (unsigned int)(size_t)(lds_void*)&sh[slot][u][2 * wk][lane], lz = (unsigned int)(size_t)(lds_void*)&sh[slot][u][4 + 2 * wn][lane];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(z0), "=&v"(z1) : "v"(la), "v"(lz) : "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], z0[s], acc00, 0, 0, 0);
          acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], z1[s], acc01, 0, 0, 0);
          acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], z0[s], acc10, 0, 0, 0);
          acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], z1[s], acc11, 0, 0, 0);
        }
        bs0 += z0; bs1 += z1;
      }
    }
  }
  WSK_STAMP(3);
  publish(nseg - 1);
  // every storing wave drains its stores (and the ring's surplus fetches), the workgroup meets, ONE lane signals for all of it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WSK_STAMP(4);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      sh_last[s] = 0;
      if (s < nseg) {
        const int j = j0 + s;
        const int nc = wsk_wg_of(a, (j + 1) * a.nbb - 1) - wsk_wg_of(a, j * a.nbb) + 1;
        const int old = __hip_atomic_fetch_add(a.tickets + j, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nc - 1) { sh_last[s] = 1; __hip_atomic_store(a.tickets + j, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
    }
  }
  __syncthreads();

  WSK_STAMP(5);
  // workgroup 0: the training kernel's loss partials (same fixed order as reduce_loss_kernel)
  if (a.loss_part && blockIdx.x == 0) {
    double* sm = reinterpret_cast<double*>(&sh[0][0][0][0]);
    double t = 0.0;
    for (int i = threadIdx.x; i < a.n_part; i += 256) t += a.loss_part[i];
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *a.loss_out = sm[0];
    __syncthreads();
  }

#pragma unroll 1
  for (int s = 0; s < nseg; ++s) {
    if (!sh_last[s]) continue;                   // (workgroup-uniform)
    // ---- last arriver of block j: total its partials in contributor order ----
    const int j = j0 + s;
    const WgradJob J = a.jobs[j];
    const LayerDesc& ld = n.L[J.layer];
    const int w_lo = wsk_wg_of(a, j * a.nbb), w_hi = wsk_wg_of(a, (j + 1) * a.nbb - 1);
    // Every load of this section is issued in BATCHES with nothing that depends on one of them in between: a dependent load under
    // load takes 2 - 3 us from beyond L2 (MI355X_MICROARCH.md, handoff-payload), so ten contributors summed one after the other, or
    // sixteen parameters updated one after the other (each waits for the previous one's stores: vmcnt counts in order), cost the
    // launch 30 us.  Loads past the last contributor re-read it and are dropped by a select (a branch around a load serialises too).
    const bool bias_lane = J.kt0 == 0 && threadIdx.x < 64;           // thread = (n-tile of the block) * 16 + column
    const int ks_true = J.layer == n.nT + n.nX + 1 ? n.L[n.nT + n.nX - 1].N : 0, ks_pad = J.layer == n.nT + n.nX + 1 ? n.L[n.nT + n.nX - 1].Np : 0;
    const int KB = ld.Kp / 16, NB = ld.Np / 16;
    // this lane's 16 elements: canonical index (or -1: padding / past the layer's edge)
    int pidx[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kt = J.kt0 + 2 * wk + (q >> 1), nt = J.nt0 + 2 * wn + (q & 1), nn = nt * 16 + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kp = kt * 16 + 4 * g + i;      // packed row
        int k = kp; bool ok = kt < KB && nt < NB && nn < ld.N;
        if (ks_pad != ks_true && kp >= ks_true) { if (kp < ks_pad) ok = false; else k = kp - (ks_pad - ks_true); }
        pidx[q][i] = ok && k < ld.K ? ld.m_w + k * ld.N + nn : -1;
      }
    }
    const int nb_ = (J.nt0 + (int)((threadIdx.x & 63) >> 4)) * 16 + c, pb = bias_lane && nb_ < ld.N ? ld.m_b + nb_ : -1;
    const bool upd_maybe = a.fuse && !sus;       // (uniform) the optimizer's operands are wanted: fetch them beside the partials
    float w0[4][4], m0[4][4], v0[4][4], wb = 0.f, mb = 0.f, vb = 0.f;
    if (upd_maybe) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = pidx[q][i] < 0 ? 0 : pidx[q][i];
          w0[q][i] = a.opt.master[p]; m0[q][i] = a.opt.mu[p]; v0[q][i] = a.opt.nu[p];
        }
      const int p = pb < 0 ? 0 : pb;
      wb = a.opt.master[p]; mb = a.opt.mu[p]; vb = a.opt.nu[p];
    }
    f32x4 t[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float tb = 0.f;
    constexpr int NCB = 6;                       // contributors per batch (24 float4 in flight per lane)
    for (int c0 = w_lo; c0 <= w_hi; c0 += NCB) {
      f32x4 pv[NCB][4]; float pbv[NCB];
#pragma unroll
      for (int e = 0; e < NCB; ++e) {
        const int cw = c0 + e <= w_hi ? c0 + e : w_hi;
        const int seg = wsk_start(a, cw) < j * a.nbb ? 1 : 0;
        const int base = ((2 * cw + seg) * WSK_PSZ) * 4 + ((wave * 4) * 64 + lane) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[e][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, base + q * 1024, 0, WSK_SC1));
        pbv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, ((2 * cw + seg) * WSK_PSZ + 4096 + (int)(threadIdx.x & 63)) * 4, 0, WSK_SC1));
      }
#pragma unroll
      for (int e = 0; e < NCB; ++e) {
        const bool live = c0 + e <= w_hi;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) t[q][i] += live ? pv[e][q][i] : 0.f;
        tb += live ? pbv[e] : 0.f;
      }
    }
    WSK_STAMP(6);
    // ---- the apply_if_finite decision ----
    bool apply = true; int nf_new = 0;
    bool nf = false;
    if ((a.fuse && sus) || (!a.fuse && a.bad)) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) nf |= pidx[q][i] >= 0 && !isfinite(t[q][i]);
      if (pb >= 0) nf |= !isfinite(tb);
    }
    if (!a.fuse && a.bad) { if (__syncthreads_or(nf ? 1 : 0) && threadIdx.x == 0) atomicOr(a.bad, 1); }
    const bool defer = a.fuse && sus;            // (grid-uniform; rare) the verdict needs every block's totals: see the end of the kernel
    // ---- per-step scalars ----
    float bc1 = 1.f, bc2 = 1.f, lr = 0.f;
    const float b1 = (float)a.opt.b1, b2 = (float)a.opt.b2;
    if (a.fuse) {
      const int c1 = st0.count + 1;
      if (st0.bc_for == c1) { bc1 = st0.bc1; bc2 = st0.bc2; }
      else { bc1 = (float)(1.0 - pow(a.opt.b1, (double)c1)); bc2 = (float)(1.0 - pow(a.opt.b2, (double)c1)); }
      lr = lr_schedule(a.opt.lr0, a.opt.learning_iter, a.opt.warmup, st0.count);
    }
    const bool upd = a.fuse && apply && !defer;
    // ---- this lane's 4 tiles x 4 rows: compute everything, then store everything ----
    if (upd) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) w0[q][i] = adam_update(w0[q][i], t[q][i], m0[q][i], v0[q][i], b1, b2, bc1, bc2, a.opt.eps, a.opt.wd, true, lr, a.opt.clip);
      wb = adam_update(wb, tb, mb, vb, b1, b2, bc1, bc2, a.opt.eps, a.opt.wd, false, lr, a.opt.clip);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kt = J.kt0 + 2 * wk + (q >> 1), nt = J.nt0 + 2 * wn + (q & 1), nn = nt * 16 + c;
      const bool all = pidx[q][0] >= 0 && pidx[q][1] >= 0 && pidx[q][2] >= 0 && pidx[q][3] >= 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = pidx[q][i];
        if (p < 0) continue;
        if (defer) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, t[q][i]), outr, p * 4, 0, WSK_SC1);
        else if (a.out) a.out[p] = t[q][i];
        if (upd) {
          a.opt.mu[p] = m0[q][i]; a.opt.nu[p] = v0[q][i]; a.opt.master[p] = w0[q][i];
          a.opt.WpT[ld.w_off + pack_index_T(kt * 16 + 4 * g + i, nn, NB)] = w0[q][i];
          if (!all) a.opt.Wp[ld.w_off + pack_index(kt * 16 + 4 * g + i, nn, KB)] = w0[q][i];
        }
      }
      if (upd && all) *reinterpret_cast<f32x4*>(a.opt.Wp + ld.w_off + ((size_t)(nt * KB + kt) * 64 + lane) * 4) = f32x4{w0[q][0], w0[q][1], w0[q][2], w0[q][3]};      // the accumulator IS the pack's float4
    }
    if (pb >= 0) {
      if (defer) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, tb), outr, pb * 4, 0, WSK_SC1);
      else if (a.out) a.out[pb] = tb;
      if (upd) { a.opt.mu[pb] = mb; a.opt.nu[pb] = vb; a.opt.master[pb] = wb; a.opt.bias[ld.b_off + nb_] = wb; }
    }
    // ---- the optimizer's scalars for the next call: by the last arriver of block 0 ----
    auto commit = [&](bool app, int nfn) {
      OptState nx;
      nx.notfinite_count = nfn;
      nx.last_lr = lr_schedule(a.opt.lr0, a.opt.learning_iter, a.opt.warmup, st0.step);
      nx.step = st0.step + 1;
      nx.last_applied = app ? 1 : 0;
      nx.count = app ? st0.count + 1 : st0.count;
      nx.bc_for = nx.count + 1;                  // the bias corrections the NEXT accepted update will use (two float64 pow calls off its path)
      nx.bc1 = (float)(1.0 - pow(a.opt.b1, (double)nx.bc_for)); nx.bc2 = (float)(1.0 - pow(a.opt.b2, (double)nx.bc_for));
      *a.opt.st_next = nx;
    };
    if (a.fuse && !defer && j == 0 && threadIdx.x == 0) commit(true, 0);
    WSK_STAMP(7);
    if (!defer) continue;
    // ---- deferred (suspicious gradient): totals are in a.out; report, draw the second ticket ----
    const int any = __syncthreads_or(nf ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int dep = any ? (unsigned int)atomicOr(a.opt.flag + 3, 1) : 0u;       // the ticket is drawn only after the flag update has returned
      const int old = atomicAdd(a.opt.flag + 4, 1 + (int)(dep & 0x40000000u));
      sh_fin = old == a.n_jobs - 1 ? 1 + (atomicOr(a.opt.flag + 3, 0) == 0 ? 1 : 0) : 0;  // 0: not the last; 1: last, non-finite; 2: last, finite
    }
    __syncthreads();
    if (sh_fin == 0) continue;
    const bool finite = sh_fin == 2;
    nf_new = finite ? 0 : st0.notfinite_count + 1;
    apply = finite || nf_new > a.opt.max_err;
    if (apply) {
      for (int p = threadIdx.x; p < n.n_params; p += 256) {
        const float gv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(outr, p * 4, 0, WSK_SC1));
        int layer = 0;
#pragma unroll
        for (int l = 1; l < MLP_MAXL; ++l) if (p >= n.L[l].m_w) layer = l;
        const LayerDesc& lq = n.L[layer];
        const bool is_bias = p >= lq.m_b;
        float m = a.opt.mu[p], v = a.opt.nu[p];
        const float wnew = adam_update(a.opt.master[p], gv, m, v, b1, b2, bc1, bc2, a.opt.eps, a.opt.wd, !is_bias, lr, a.opt.clip);
        a.opt.mu[p] = m; a.opt.nu[p] = v; a.opt.master[p] = wnew;
        if (is_bias) a.opt.bias[lq.b_off + (p - lq.m_b)] = wnew;
        else {
          const int e = p - lq.m_w, k = e / lq.N, nn = e - k * lq.N, kk = packed_row(n, layer, k);
          a.opt.Wp[lq.w_off + pack_index(kk, nn, lq.Kp / 16)] = wnew;
          a.opt.WpT[lq.w_off + pack_index_T(kk, nn, lq.Np / 16)] = wnew;
        }
      }
    }
    if (threadIdx.x == 0) commit(apply, nf_new);
  }
  WSK_STAMP(8);
}

static void wsk_plan(int n_jobs, int nbb, int& G, int& q, int& r) {
  const int U = n_jobs * nbb;
  G = U < 512 ? U : 512;
  if (G < n_jobs) G = n_jobs;                    // a range never longer than one block's chain axis: at most two partial blocks per workgroup
  q = U / G; r = U - q * G;
}
int launch_wgrad_sk(const WskArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(wgrad_sk_kernel, dim3(a.G), dim3(256), 0, stream, a);
  return 0;
}
