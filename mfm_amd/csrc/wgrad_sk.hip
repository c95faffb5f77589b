// K4b + K7 in ONE launch: the weight gradients dW = A^T dZ, db = sum dZ of the fused tile family (exe_flow_matching.py:364-365, the
// parameter half of jax.value_and_grad) as a STREAM-K GEMM over the chain axis, and -- on one rank -- the optimizer step of
// state.apply_gradients (:366; optax chain of :129-137,184: optim.hip) applied slice by slice by the workgroups that produced the block.
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
//    behind a COUNTED vmcnt and a raw s_barrier (a __syncthreads() would drain the ring: cdna_hip_programming.md section 5); the
//    fragments are read back by inline-assembly ds_reads (a compiler-visible read of an array that LDS-DMA writes is preceded by
//    s_waitcnt vmcnt(0)), and ALL LDS is one __shared__ object (trap 4(a) of the same section);
//  * everything the kernel needs beside its operands -- layer table, block table with the tile rows each wave fetches, optimizer
//    hyper-parameters -- sits in ONE device-memory struct written at mfm_create (WskConst); the launch arguments are a dozen
//    words.  (As 2 KB of launch arguments the same tables cost every workgroup 4.7 us of dependent scalar loads before its first
//    fetch: tools/wsk_stamps.py.)
//  * partial blocks are published in ACCUMULATOR layout (one float4 per lane and MFMA tile: coalesced) with write-through (sc1)
//    stores; each storing wave drains its stores, the workgroup meets at a barrier and ONE lane adds to the block's arrival
//    counter.  The partial buffer is touched by sc1 accesses ONLY (never a plain load, store or memset: a line that a plain access
//    left in some XCD's L2 is what an sc1 load may be served from);
//  * EVERY contributor then combines its own SLICE of the block (1 / n_contributors of its 1,040 float4): one lane polls the
//    arrival counter (all workgroups of the grid are resident: mfm_create checks; see the guard in api.hip), the slice's float4 of
//    every contributor's partial is read back with sc1 loads -- ONE batch, all in flight at once -- and summed in contributor
//    order (deterministic: no float atomics), and the slice's parameters are
//      - written to the canonical gradient (a multi-rank host all-reduces it), and
//      - with `fuse`: updated by apply_if_finite(adamw, clip), their packed copies re-emitted.
//    (The first form of this seam let the LAST workgroup to arrive combine the whole block: 176 KB of dependent reads by one
//    workgroup, 11 us after the last ticket, 15 us for the slowest block; spread over the contributors it is 16 KB each.)
//  * apply_if_finite needs "is ANY element of the gradient non-finite" before the FIRST parameter is touched.  The training kernel
//    raises a flag when a value it stores for this kernel (activation or pre-activation gradient) is not <= 1e15 in magnitude (NaN
//    included): with the flag clear every partial and every total is a sum of at most 2^20 products below 1e30, i.e. finite, and the
//    slices update independently.  With the flag raised (a diverged run) no slice updates on its own: every slice leaves its totals
//    in the canonical gradient, reports whether they are finite and draws a second ticket (fenced hand-off: the gradient buffer is
//    ordinary memory); the workgroup that draws the LAST of those knows the verdict on the whole gradient and runs the optimizer
//    over all parameters itself -- slow (one workgroup) and rare.
#include "mlp.hip.h"

#ifdef MFM_WSK_STAMPS
__device__ unsigned long long* g_wsk_dbg = nullptr;      // [WG][16] section time stamps (development build only)
#define WSK_STAMP(id) do { if (g_wsk_dbg && threadIdx.x == 0) g_wsk_dbg[blockIdx.x * 16 + (id)] = __builtin_amdgcn_s_memrealtime(); } while (0)      /* 100 MHz, one clock for the whole device */
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
constexpr int WSK_PF4 = 4 * 4 * 64 + 16;          // float4 per partial block: [wave][tile of the wave's quadrant][lane] + 64 bias sums
constexpr int WSK_PSZ = 4 * WSK_PF4;              // floats
constexpr int WSK_NCB = 8;                        // contributors read per batch

struct WskJob { int layer, kt0, nt0, a_row[4], z_row[4]; };      // a_row / z_row: packed tile rows wave 0..3 FETCHES (clamped at the layer's edge)

struct WskConst {          // device memory, written once per context
  NetDev net;
  const float* acts; const float* dzs;
  int nbb, n_jobs, G;      // chain tiles, 64 x 64 blocks, workgroups
  int upw_q, upw_r;        // units per workgroup: the first upw_r take upw_q + 1, the others upw_q
  int n_slices;            // sum over the blocks of their contributor counts
  float* partials;         // [2 G][WSK_PSZ]
  float *master, *mu, *nu, *Wp, *WpT, *bias;
  double lr0; int learning_iter, warmup;
  double b1, b2; float eps, wd, clip;
  int max_err;
  WskJob jobs[WSK_MAXJOBS];
};

// one workgroup's unit range and the tile rows its waves fetch, precomputed: ONE dependent load between the launch arguments and the first fetch
struct WskWg { int j0, bb0, cnt, n0; int a_row[2][4], z_row[2][4]; };

struct WskArgs {           // per launch
  const WskConst* C;
  const WskWg* wg;         // [G]
  const float* acts; const float* dzs; int nbb, G;
  int xcd_remap;           // != 0: consecutive unit ranges go to workgroups of ONE XCD (blockIdx.x % 8 names the workgroups that share one)
  int* tickets;            // [n_jobs] arrival counters of THIS launch (zero at its start)
  int* tickets_clear;      // [n_jobs] the other parity's counters: zeroed here for the next launch
  float* out;              // [n_params] the summed gradient, canonical layout
  int* bad;                // non-null (fuse == 0): raised when a TOTAL is non-finite (the verdict mfm_adamw_step reuses on one rank)
  const double* loss_part; int n_part; double* loss_out;      // non-null: workgroup 0 totals the training kernel's loss partials
  int fuse;                // != 0: the optimizer step rides here
  const OptState* st; OptState* st_next;
  int* flag;               // [3] / [4]: non-finite flag and second-ticket counter of the suspicious path (cleared by the training kernel)
  const int* suspicious;   // the training kernel's "a stored value was huge or NaN" word for THIS iteration
  int force_exchange;      // tests: every launch through the suspicious path
};

__host__ __device__ __forceinline__ int wsk_start(int q, int r, int w) { return w * q + (w < r ? w : r); }
__host__ __device__ __forceinline__ int wsk_wg_of(int q, int r, int u) {
  const int big = r * (q + 1);
  return u < big ? u / (q + 1) : r + (u - big) / q;
}

typedef unsigned int wsk_u32x4 __attribute__((ext_vector_type(4)));
constexpr int WSK_SC1 = 16;      // cache-policy bit of the raw buffer intrinsics that sets sc1 on gfx940+ (write-through store / L1-bypassing load)

// float4 `f` of a block's partial layout -> the canonical indices of its four elements (-1: padding / past the layer's edge), the packed
// row / column of element 0 and whether it is a bias float4
struct WskElem { int p[4]; int kp0, nn0; bool bias; };
__device__ __forceinline__ WskElem wsk_decode(const NetDev& n, const WskJob& J, int f) {
  WskElem e;
  const LayerDesc& ld = n.L[J.layer];
  const int KB = ld.Kp / 16, NB = ld.Np / 16;
  if (f >= 1024) {                     // bias: float e of the 64 is column (nt0 + e / 16) * 16 + e % 16
    e.bias = true; e.kp0 = 0;
    const int e0 = 4 * (f - 1024);
    e.nn0 = (J.nt0 + (e0 >> 4)) * 16 + (e0 & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) e.p[i] = J.kt0 == 0 && J.nt0 + (e0 >> 4) < NB && e.nn0 + i < ld.N ? ld.m_b + e.nn0 + i : -1;
    return e;
  }
  e.bias = false;
  const int wv = f >> 8, q = (f >> 6) & 3, ln = f & 63, g = ln >> 4, c = ln & 15;
  const int kt = J.kt0 + 2 * (wv >> 1) + (q >> 1), nt = J.nt0 + 2 * (wv & 1) + (q & 1);
  e.kp0 = kt * 16 + 4 * g; e.nn0 = nt * 16 + c;
  const int joint = n.nT + n.nX + 1;
  const int ks_true = J.layer == joint ? n.L[joint - 2].N : 0, ks_pad = J.layer == joint ? n.L[joint - 2].Np : 0;      // (mlp.hip.h: packed_row)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kp = e.kp0 + i;
    int k = kp; bool ok = kt < KB && nt < NB && e.nn0 < ld.N;
    if (ks_pad != ks_true && kp >= ks_true) { if (kp < ks_pad) ok = false; else k = kp - (ks_pad - ks_true); }
    e.p[i] = ok && k < ld.K ? ld.m_w + k * ld.N + e.nn0 : -1;
  }
  return e;
}

// The first six arguments repeat fields of `a`: scalar arguments at the head of the argument list are PRELOADED into SGPRs by the
// command processor (-mllvm -amdgpu-kernarg-preload-count, mfm_amd/build.py), so the unit-range lookup and the first fetches of the
// ring do not wait for a read of the kernel-argument segment first (three dependent scalar-load round trips in the by-value form:
// xcd flag -> G -> pointers; tools/wsk_stamps.py: 3.6 us from the workgroup's start to the ring's last prime fetch).  On a stack
// without the preload the compiler's compatibility prologue loads them the usual way.
__global__ __launch_bounds__(256, 2) void wgrad_sk_kernel(const WskWg* wg_tab, const float* acts_p, const float* dzs_p, int nbb_p, int G_remap,
                                                          WskArgs a) {      // (two workgroups per CU: 2 waves per SIMD, at most 256 registers per lane)
  // ONE __shared__ object: with a second one beside the LDS-DMA ring hipcc waits vmcnt(0) before the first ds_read of every stage
  // and the ring drains (cdna_hip_programming.md section 5, trap 4(a)); the flag word lives behind the ring
  __shared__ f32x4 sh_all[WSK_NS * WSK_TB * 8 * 64 + 1];
  f32x4 (*sh)[WSK_TB][8][64] = reinterpret_cast<f32x4 (*)[WSK_TB][8][64]>(sh_all);      // [stage][chain tile][A0..A3, Z0..Z3][lane]
  int& sh_fin = *reinterpret_cast<int*>(sh_all + WSK_NS * WSK_TB * 8 * 64);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const int wk = wave >> 1, wn = wave & 1;
  WSK_STAMP(0);
  // workgroup -> unit range
  int w = blockIdx.x;
  if (G_remap > 0) {           // (G_remap: G, negated when the remap is off) bijective for any G (cdna_hip_programming.md, XCD swizzle): the blockIdx.x % 8 class x gets a contiguous run of w
    const int q = G_remap >> 3, r = G_remap & 7, x = w & 7, i = w >> 3;
    w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const WskWg& D = wg_tab[w];
  const int nbb = nbb_p, j0 = D.j0, bb0 = D.bb0, cnt = D.cnt;
  const int n0 = D.n0;                                               // tiles of the first segment (block j0); the rest belong to block j0 + 1
  const int nseg = n0 < cnt ? 2 : 1;
  // per segment: the tiles this wave FETCHES (A tile `wave`, dZ tile `wave` of the block)
  const f32x4* Af[2]; const f32x4* Zf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Af[s] = reinterpret_cast<const f32x4*>(acts_p) + (size_t)D.a_row[s][wave] * nbb * 64 + lane;
    Zf[s] = reinterpret_cast<const f32x4*>(dzs_p) + (size_t)D.z_row[s][wave] * nbb * 64 + lane;
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
  WSK_STAMP(1);
  const WskConst& C = *a.C;
  const NetDev& n = C.net;
  const int upq = C.upw_q, upr = C.upw_r;
  // optimizer scalars and flags: read here, first used after the main loop
  OptState st0; int sus = 0;
  if (a.fuse) { st0 = *a.st; sus = *a.suspicious | a.force_exchange; }
  if (blockIdx.x == 0 && (int)threadIdx.x < C.n_jobs) a.tickets_clear[threadIdx.x] = 0;

  f32x4 acc00 = {0, 0, 0, 0}, acc01 = acc00, acc10 = acc00, acc11 = acc00, bs0 = acc00, bs1 = acc00;
  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(C.partials, 0, 0x7fffffff, 0x00020000);
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

  // (Tried: the fragments of tile i + 1 requested from LDS, and the stage hand-over done, between the first four and the other twelve
  // MFMAs of tile i, so that one wave per SIMD would keep the matrix pipe busy and 256 equal workgroups could replace 416 -- slower in
  // every decomposition: 38.1 us at 416 workgroups against 36.5, 40.3 - 40.8 at 256, 43.6 at 512; tools/dbg/wsk_sweep.sh.)
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
    for (int s = 0; s < 2; ++s)
      if (s < nseg) __hip_atomic_fetch_add(a.tickets + j0 + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- per-step scalars; workgroup 0: the optimizer's scalars for the next call and the training kernel's loss partials ----
  float bc1 = 1.f, bc2 = 1.f, lr = 0.f;
  const float b1 = (float)C.b1, b2 = (float)C.b2;
  auto commit = [&](bool app, int nfn) {
    OptState nx;
    nx.notfinite_count = nfn;
    nx.last_lr = lr_schedule(C.lr0, C.learning_iter, C.warmup, st0.step);
    nx.step = st0.step + 1;
    nx.last_applied = app ? 1 : 0;
    nx.count = app ? st0.count + 1 : st0.count;
    nx.bc_for = nx.count + 1;                    // the bias corrections the NEXT accepted update will use (two float64 pow calls off its path)
    nx.bc1 = (float)(1.0 - pow(C.b1, (double)nx.bc_for)); nx.bc2 = (float)(1.0 - pow(C.b2, (double)nx.bc_for));
    *a.st_next = nx;
  };
  if (a.fuse) {
    const int c1 = st0.count + 1;
    if (st0.bc_for == c1) { bc1 = st0.bc1; bc2 = st0.bc2; }
    else { bc1 = (float)(1.0 - pow(C.b1, (double)c1)); bc2 = (float)(1.0 - pow(C.b2, (double)c1)); }
    lr = lr_schedule(C.lr0, C.learning_iter, C.warmup, st0.count);
  }
  const bool defer = a.fuse && sus;              // (grid-uniform; rare) the verdict needs every slice's totals: see the end of the kernel
  const bool upd = a.fuse && !defer;
  if (blockIdx.x == 0 && upd && threadIdx.x == 0) commit(true, 0);
  WSK_STAMP(5);

  // ---- every contributor combines its slice of the block ----
  bool nf_any = false;
#pragma unroll 1
  for (int s = 0; s < nseg; ++s) {
    const int j = j0 + s;
    const WskJob& J = C.jobs[j];
    const LayerDesc& ld = n.L[J.layer];
    const int w_lo = wsk_wg_of(upq, upr, j * nbb), w_hi = wsk_wg_of(upq, upr, (j + 1) * nbb - 1), nc = w_hi - w_lo + 1;
    const int per = (WSK_PF4 + nc - 1) / nc, f_lo = (w - w_lo) * per, f_hi = f_lo + per < WSK_PF4 ? f_lo + per : WSK_PF4;
    int f = f_lo + (int)threadIdx.x;
    // the optimizer's operands of the first float4 this thread combines: requested BEFORE the wait for the other contributors
    WskElem e = wsk_decode(n, J, f < f_hi ? f : 0);
    float w0[4], m0[4], v0[4];
    auto fetch_state = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int p = e.p[i] < 0 ? 0 : e.p[i]; w0[i] = C.master[p]; m0[i] = C.mu[p]; v0[i] = C.nu[p]; }
    };
    if (upd) fetch_state();
    // wait until every contributor of block j has published (one lane polls; the counter only grows during a launch)
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(a.tickets + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nc) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (s == 0) WSK_STAMP(6);
    for (; f < f_hi; f += 256) {                 // (one pass unless the block has fewer than 5 contributors)
      // ---- this float4 of every contributor's partial, in batches that are in flight together; summed in contributor order ----
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = w_lo; c0 <= w_hi; c0 += WSK_NCB) {
        f32x4 pv[WSK_NCB];
#pragma unroll
        for (int q = 0; q < WSK_NCB; ++q) {      // (past the last contributor: re-read it, dropped by the select below -- a branch around a load serialises)
          const int cw = c0 + q <= w_hi ? c0 + q : w_hi;
          const int seg = wsk_start(upq, upr, cw) < j * nbb ? 1 : 0;
          pv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, ((2 * cw + seg) * WSK_PF4 + f) * 16, 0, WSK_SC1));
        }
#pragma unroll
        for (int q = 0; q < WSK_NCB; ++q) {
          const bool live = c0 + q <= w_hi;
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] += live ? pv[q][i] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) nf_any |= e.p[i] >= 0 && !isfinite(t[i]);
      // ---- canonical gradient, optimizer, packed copies ----
      if (upd) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w0[i] = adam_update(w0[i], t[i], m0[i], v0[i], b1, b2, bc1, bc2, C.eps, C.wd, !e.bias, lr, C.clip);
      }
      const bool all = e.p[0] >= 0 && e.p[1] >= 0 && e.p[2] >= 0 && e.p[3] >= 0;
      const int KB = ld.Kp / 16, NB = ld.Np / 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = e.p[i];
        if (p < 0) continue;
        a.out[p] = t[i];
        if (upd) {
          C.mu[p] = m0[i]; C.nu[p] = v0[i]; C.master[p] = w0[i];
          if (e.bias) C.bias[ld.b_off + e.nn0 + i] = w0[i];
          else {
            C.WpT[ld.w_off + pack_index_T(e.kp0 + i, e.nn0, NB)] = w0[i];
            if (!all) C.Wp[ld.w_off + pack_index(e.kp0 + i, e.nn0, KB)] = w0[i];
          }
        }
      }
      if (upd && all && !e.bias)                 // the accumulator float4 IS a float4 of the forward pack
        *reinterpret_cast<f32x4*>(C.Wp + ld.w_off + pack_index(e.kp0, e.nn0, KB)) = f32x4{w0[0], w0[1], w0[2], w0[3]};
      if (f + 256 < f_hi) { e = wsk_decode(n, J, f + 256); if (upd) fetch_state(); }
    }
  }
  WSK_STAMP(7);
  if (a.loss_part && blockIdx.x == 0) {         // workgroup 0: the training kernel's loss partials, same fixed order as reduce_loss_kernel
    double* sm = reinterpret_cast<double*>(sh_all);
    double t = 0.0;
    for (int i = threadIdx.x; i < a.n_part; i += 256) t += a.loss_part[i];
    __syncthreads();
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *a.loss_out = sm[0];
    __syncthreads();
  }
  if (!a.fuse && a.bad) { if (__syncthreads_or(nf_any ? 1 : 0) && threadIdx.x == 0) atomicOr(a.bad, 1); }
  if (!defer) { WSK_STAMP(8); return; }
  // ---- suspicious gradient: the totals are in a.out; report, draw the second ticket(s) ----
  // a.out is ordinary memory (plain stores; lines of it may sit in any XCD's L2 from earlier kernels): this hand-off is the fenced
  // form -- every storing wave drains, the workgroup meets, ONE lane releases at agent scope (L2 write-back) before the ticket; the
  // workgroup that draws the last ticket acquires before it reads.
  const int any = __syncthreads_or(nf_any ? 1 : 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int dep = any ? (unsigned int)atomicOr(a.flag + 3, 1) : 0u;             // the tickets are drawn only after the flag update has returned
    const int old = atomicAdd(a.flag + 4, nseg + (int)(dep & 0x40000000u));
    sh_fin = old + nseg == C.n_slices ? 1 + (atomicOr(a.flag + 3, 0) == 0 ? 1 : 0) : 0;    // 0: not the last; 1: last, non-finite; 2: last, finite
    if (sh_fin) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  __syncthreads();
  if (sh_fin == 0) return;
  const bool finite = sh_fin == 2;
  const int nf_new = finite ? 0 : st0.notfinite_count + 1;
  const bool apply = finite || nf_new > C.max_err;
  if (apply) {
    for (int p = threadIdx.x; p < n.n_params; p += 256) {
      const float gv = a.out[p];
      int layer = 0;
#pragma unroll
      for (int l = 1; l < MLP_MAXL; ++l) if (p >= n.L[l].m_w) layer = l;
      const LayerDesc& lq = n.L[layer];
      const bool is_bias = p >= lq.m_b;
      float m = C.mu[p], v = C.nu[p];
      const float wnew = adam_update(C.master[p], gv, m, v, b1, b2, bc1, bc2, C.eps, C.wd, !is_bias, lr, C.clip);
      C.mu[p] = m; C.nu[p] = v; C.master[p] = wnew;
      if (is_bias) C.bias[lq.b_off + (p - lq.m_b)] = wnew;
      else {
        const int el = p - lq.m_w, k = el / lq.N, nn = el - k * lq.N, kk = packed_row(n, layer, k);
        C.Wp[lq.w_off + pack_index(kk, nn, lq.Kp / 16)] = wnew;
        C.WpT[lq.w_off + pack_index_T(kk, nn, lq.Np / 16)] = wnew;
      }
    }
  }
  if (threadIdx.x == 0) commit(apply, nf_new);
}

// Grid and units per workgroup: U = n_jobs * nbb units over G workgroups, as evenly as integers allow.  Default: EIGHT chain slices per
// block, workgroup w = block w / 8, slice w % 8 -- consecutive workgroup ids go round-robin to the 8 XCDs, so all blocks of one chain
// slice run on one XCD and walk its chain tiles together: an operand tile that 2 - 4 blocks of its layer need comes from beyond that
// XCD's L2 once.  This kernel is bound by exactly that traffic: measured at the headline shape (tools/iter_time.py, wgrad + optimizer)
//   512 workgroups x 26 units, block-major (every CU two workgroups, ~95 MB from beyond L2)   42.0 us
//   the same, consecutive ranges on one XCD                                                   39.7 us
//   416 workgroups = 52 blocks x 8 slices of 32 tiles, slice = XCD (~49 MB; 160 CUs carry two workgroups)   37.0 us
// (the slab kernels it replaces: 28.5 + 12.5 us).  Eight slices also make the sums bit-identical with those kernels' eight slabs.
static void wsk_plan(int n_jobs, int nbb, int cap, int& G, int& q, int& r) {
  const int U = n_jobs * nbb;
  int S = nbb < 8 ? nbb : 8;
  while (S > 1 && n_jobs * S > cap) --S;         // (every workgroup must be resident: api.hip)
  G = n_jobs * S;
  if (const char* e = getenv("MFM_WSK_G")) { const int v = atoi(e); if (v >= n_jobs && v <= U && v <= cap) G = v; }      // development: A/B of the decomposition
  q = U / G; r = U - q * G;                      // a range is never longer than one block's chain axis: at most two partial blocks per workgroup
}
int launch_wgrad_sk(const WskArgs& a, int G, hipStream_t stream) {
  hipLaunchKernelGGL(wgrad_sk_kernel, dim3(G), dim3(256), 0, stream, a.wg, a.acts, a.dzs, a.nbb, a.xcd_remap ? a.G : -a.G, a);
  return 0;
}
