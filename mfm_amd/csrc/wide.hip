// The WIDE kernel family: networks / dimensions that do not fit the fused 16-chain LDS tile (mlp.hip.h) -- the
// "pines" configuration of the reference (multi_modal.py:89-96: hidden widths 1024, d = 1024 / 1600; SURVEY.md
// section 8d config C4).  Same arithmetic and the same C ABI as the fused family; what changes is the data movement:
//
//   * activations live in HBM / MALL as plain row-major [rows][features] float32 (a 1024-wide layer of 1024 chains is
//     4 MB: L2-resident between the producing and the consuming kernel), one GEMM kernel launch per layer;
//   * the GEMM is computed TRANSPOSED on the matrix cores: D = W^T-tile (A operand, the packed weights exactly as the
//     fused family streams them) x activation-tile^T (B operand: one float4 per lane from a row-major activation row).
//     The f32 accumulator of v_mfma_f32_16x16x4_f32 then holds 4 consecutive FEATURES of one chain per lane, so
//     bias / ReLU / masks / residual adds and the store are float4 wide and row-major again: no LDS staging, no
//     transposes, no packed activation copies;
//   * forward-mode (value + tangent) evaluation shares every weight fragment between the value rows and the tangent
//     rows of the same chains (two accumulator sets), with the ReLU mask applied to the tangent in the epilogue;
//   * the weight-gradient GEMM (reduction over chains) loads BOTH operands as row-major float4 and issues the 4 x 4
//     outer product of their elements as 16 MFMAs (a 64 x 64 block of dW per wave, k-depth 4 chains per MFMA); the
//     accumulators come out as float4 runs of the canonical [in][out] gradient layout;
//   * the adaptive Dopri5 state machine (oracle/ode.py) is the same per-chain masked lock-step
//     as ode.hip, but driven from the host: one attempt = 6 field evaluations of ~10 launches each on the context's
//     stream, then a 4-byte read-back of the number of chains still integrating.
//
// Replaces the same reference code as fm.hip / ode.hip: exe_flow_matching.py:56-90 (VectorFieldNet), :151-178 (loss),
// :206-242 (CNF transforms), :246-278 (flow-MH steps), jax.value_and_grad at :364-365.
#include <type_traits>
#include "mlp.hip.h"
#include "prng.hip.h"

namespace wide {

// ---------------------------------------------------------------------------------------------------------------------
// GEMM:  Y[r][n] = epi( sum_k X[r][k] W[k][n] ),  optionally also the tangent rows YT = epi'( XT W ).
// ---------------------------------------------------------------------------------------------------------------------
struct Gemm {
  const float* W; int KB, NT;            // packed weights [NT][KB][64 lanes] float4 (mlp.hip.h "Wp" layout), K blocks, N tiles
  const float* bias;                     // [16 NT] or null
  const float* X; int ldx;               // value rows (row-major, ld multiple of 4)
  const float* XT; int KBT;              // tangent rows (same ld) and the K blocks they span (<= KB); null: no tangent GEMM
  const float* TS; int ldts;             // tangent taken from a precomputed buffer instead of a GEMM (x1 layer: z W_x1)
  float* Y; int ldy; int ycol;           // value output (+ column offset, e.g. the st half of [sx | st])
  float* YT;                             // tangent output (same ld / offset) or null
  int rows;                              // multiple of 16
  int act;                               // 0: linear; k + 1: y = act_k(pre), tangent scaled by act_k'(pre)  (MFM_ACT_*)
  float* P;                              // optional copy of the pre-activations (same ld / offset as Y): what the backward
                                         // pass differentiates through for the non-invertible activations (gelu, swish)
  const float* mask; int ldm, mcol;      // backward: y = pre * act'(.) with act' from the stored OUTPUT (relu / tanh / elu)
  int mask_kind, mask_is_pre;            // ... or from the stored PRE-ACTIVATION (mask_is_pre)
  const float* add; int lda, acol;       // backward: pre += add (second contribution to the same activation)
  int KBW;                               // k-blocks per N tile in the PACKED weights when the product spans only the first KB of them
                                         // (the sx half of the joint layer's input); 0: = KB
  int mask_rdiv;                         // > 1: mask row = row / mask_rdiv (exact trace: hx1 tangent rows per chain share the chain's mask row)
};

// Epilogue shared by the GEMM kernels: lane (g, c) holds features 16 nt + 4 g .. + 3 of row 16 mt + c.
template <int MTW, int NTW>
__device__ __forceinline__ void gemm_epilogue(const Gemm& a, f32x4 (&acc)[NTW][MTW], f32x4 (&acT)[NTW][MTW], int mt0, int nt0, int MT, int g, int c) {
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int nt = nt0 + j;
    if (nt >= a.NT) continue;
    const int f0 = nt * 16 + 4 * g;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + f0);
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      if (mt0 + m >= MT) continue;
      const size_t row = (size_t)(mt0 + m) * 16 + c;
      f32x4 pre = acc[j][m] + bv;
      if (a.add) pre += *reinterpret_cast<const f32x4*>(a.add + row * a.lda + a.acol + f0);
      if (a.mask) {
        const size_t mrow = a.mask_rdiv > 1 ? row / (size_t)a.mask_rdiv : row;
        const f32x4 mk = *reinterpret_cast<const f32x4*>(a.mask + mrow * a.ldm + a.mcol + f0);
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = a.mask_is_pre ? mask_pre(mk[i], pre[i], a.mask_kind) : mask_out(mk[i], pre[i], a.mask_kind);
      }
      f32x4 y = pre;
      if (a.act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = act_f(pre[i], a.act - 1);
        if (a.P) *reinterpret_cast<f32x4*>(a.P + row * a.ldy + a.ycol + f0) = pre;
      }
      *reinterpret_cast<f32x4*>(a.Y + row * a.ldy + a.ycol + f0) = y;
      if (a.YT) {
        f32x4 t = acT[j][m];
        if (a.TS) t = *reinterpret_cast<const f32x4*>(a.TS + row * a.ldts + f0);
        if (a.act) {
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] = mask_pre(pre[i], t[i], a.act - 1);
        }
        *reinterpret_cast<f32x4*>(a.YT + row * a.ldy + a.ycol + f0) = t;
      }
    }
  }
}

// Keeps a group of prefetch requests where the source puts them: (1) a compiler memory barrier -- without it the optimizer
// folds "phi of loads" into a load of a phi of addresses at the top of the iteration that USES the data; (2) a scheduling
// barrier -- without it the machine scheduler sinks the requests below the MFMAs of the blocks in front of them (both seen in
// the ISA of round 1's kernels: vmcnt(0) two instructions after the request, 50 % / 28 % MFMA occupancy)
#define WIDE_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int MTW, int NTW, bool DUAL, int RING = 4>
__global__ __launch_bounds__(256) void gemm_kernel(Gemm a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const int mt0 = (blockIdx.y * 2 + (wave >> 1)) * MTW, nt0 = (blockIdx.x * 2 + (wave & 1)) * NTW;
  const int MT = a.rows >> 4;
  if (mt0 >= MT || nt0 >= a.NT) return;                       // no barriers in this kernel
  const f32x4* wp[NTW];
  const float* xp[MTW];
  const float* xtp[MTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) wp[j] = reinterpret_cast<const f32x4*>(a.W) + (size_t)(nt0 + j < a.NT ? nt0 + j : nt0) * (a.KBW ? a.KBW : a.KB) * 64 + lane;
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const size_t row = (size_t)(mt0 + m < MT ? mt0 + m : mt0) * 16 + c;
    xp[m] = a.X + row * a.ldx + 4 * g;
    xtp[m] = DUAL ? a.XT + row * a.ldx + 4 * g : nullptr;
  }
  f32x4 acc[NTW][MTW], acT[NTW][MTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int m = 0; m < MTW; ++m) { acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f}; acT[j][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  struct Frag { f32x4 w[NTW], x[MTW], t[MTW]; };
  using WithT = std::integral_constant<bool, true>;
  using NoT = std::integral_constant<bool, false>;
  auto load = [&](Frag& f, int kb, auto with_t) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) f.w[j] = wp[j][(size_t)kb * 64];
#pragma unroll
    for (int m = 0; m < MTW; ++m) f.x[m] = *reinterpret_cast<const f32x4*>(xp[m] + kb * 16);
    if constexpr (DUAL && decltype(with_t)::value) {
      const int kt = kb < a.KBT ? kb : a.KBT - 1;
#pragma unroll
      for (int m = 0; m < MTW; ++m) f.t[m] = *reinterpret_cast<const f32x4*>(xtp[m] + kt * 16);
    }
  };
  auto mac = [&](const Frag& f, auto with_t) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int m = 0; m < MTW; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[j][s], f.x[m][s], acc[j][m], 0, 0, 0);
    if constexpr (DUAL && decltype(with_t)::value) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int m = 0; m < MTW; ++m) acT[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[j][s], f.t[m][s], acT[j][m], 0, 0, 0);
    }
  };
  // A ring of RING fragment sets: the loads of block kb + RING - 1 are issued before block kb is multiplied, i.e. RING - 1 blocks
  // ((RING - 1) x 16 MTW NTW MFMA issue slots) of latency cover for a wave that is often alone on its SIMD (the launch is ~one
  // workgroup per CU at the pines shape).  Every load of the loop is UNCONDITIONAL (block indices clamped to the last block,
  // the tangent's to its last block): with the loads under `if (kb + 3 < KB)` the compiler could not count them and waited
  // with vmcnt(0) in front of the first MFMA of every round -- for the block it had just requested (round 1: 50 % MFMA
  // occupancy on the 1024^3 layer).  Tangent blocks come first (K blocks [0, KBT)), then the value-only remainder.
  Frag f[RING];
  const int kl = a.KB - 1;
  auto ck = [&](int kb) { return kb < kl ? kb : kl; };
  const int kt_end = DUAL ? a.KBT : 0;                         // a multiple of RING on the path that uses it
  if ((DUAL && (a.KBT % RING)) || (a.KB % RING)) {
    // general case (never taken by the reference's widths): one block at a time, tangent under a uniform branch
    for (int kb = 0; kb < a.KB; ++kb) {
      load(f[0], kb, WithT{});
      if (kb < a.KBT) mac(f[0], WithT{}); else mac(f[0], NoT{});
    }
  } else {
#pragma unroll
    for (int j = 0; j < RING - 1; ++j) load(f[j], ck(j), WithT{});
    WIDE_PIN();
    int kb = 0;
    for (; kb < kt_end; kb += RING) {
#pragma unroll
      for (int j = 0; j < RING; ++j) { load(f[(j + RING - 1) % RING], ck(kb + j + RING - 1), WithT{}); WIDE_PIN(); mac(f[j], WithT{}); }
    }
    for (; kb < a.KB; kb += RING) {
#pragma unroll
      for (int j = 0; j < RING; ++j) { load(f[(j + RING - 1) % RING], ck(kb + j + RING - 1), NoT{}); WIDE_PIN(); mac(f[j], NoT{}); }
    }
  }
  gemm_epilogue<MTW, NTW>(a, acc, acT, mt0, nt0, MT, g, c);
}

// The same GEMM with the activation tile staged through LDS: workgroup tile 64 rows x 64 features, K in steps of 64.
// Why: the fragment-shaped activation loads of gemm_kernel (16 rows x 64 B per instruction) use half of every 128-byte line
// they touch -- twice the texture-addresser time for the same bytes -- and four waves each load their own copy.  Here the
// 256 threads load the [64 rows][64 k] tile ONCE per step in full lines (16 lanes x 16 B per row), park it in registers for
// one step (2,048 MFMA cycles of latency cover), write it to a [64][72] LDS image (leading dimension = 8 mod 64 dwords:
// the ds_read_b128 A-fragment reads -- 16 rows x 4 k-groups -- touch all 64 banks once) and every wave reads its fragments
// from there.  Waves are arranged 1 (rows) x 4 (features): a wave owns one 16-feature tile for all 64 rows, so each packed
// weight fragment is loaded by exactly one wave (2 x 2 waves loaded every fragment twice); the weights stay on the
// register ring, three k-blocks ahead.  One barrier per step, placed in front of the step's LAST k-block: by then every
// wave has written the next tile (at the step's start) and has issued its last fragment read of this one.
constexpr int GL_LD = 72;
__device__ __forceinline__ f32x4 wide_bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
#ifdef WIDE_DBG_CLOCK
__device__ unsigned long long wide_dbg_clk[4];
#endif
// WM = 1: four waves, each a 16-feature tile x all 64 rows.  WM = 2: eight waves (two per SIMD), each 16 features x 32 rows:
// every weight fragment is then loaded by two waves, but a SIMD has a second wave to issue from while the first waits.
template <bool DUAL, int WM>
__global__ __launch_bounds__(256 * WM) void gemm_lds_kernel(Gemm a) {
  constexpr int MTW = 4 / WM, NP = 4 / WM;                     // M tiles per wave; staging passes per thread
  __shared__ __attribute__((aligned(16))) float Xs[2][64][GL_LD];
  __shared__ __attribute__((aligned(16))) float Ts[DUAL ? 2 : 1][DUAL ? 64 : 1][DUAL ? GL_LD : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c = lane & 15;
  const int wn = wave & 3, wm = wave >> 2;
  const int MT = a.rows >> 4;
  // XCD-aware tile order: consecutive workgroup ids go to consecutive XCDs (id % 8), each with its own L2.  Give XCD q the
  // compact block of tiles (q % QX, q / QX) so that the activation rows and weight columns it streams are shared by its 32 CUs
  int bx = blockIdx.x, by = blockIdx.y;
#ifndef WIDE_DBG_NO_REMAP
  if ((gridDim.x & 7) == 0 && (gridDim.y & 3) == 0) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, q = id & 7, j = id >> 3;
    const int bw = gridDim.x >> 1, bh = gridDim.y >> 2;        // block of tiles per XCD: (gx / 2) x (gy / 4)
    bx = (q & 1) * bw + j % bw; by = (q >> 1) * bh + j / bw;
  }
#endif
  const int mt0 = by * 4 + wm * MTW, nt = bx * 4 + wn;
  // Operands through buffer descriptors: per-lane offsets are loop constants, the k position is a SCALAR offset (no vector
  // address arithmetic between the MFMAs) and reads past the end of a descriptor return zeros (no clamps)
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.W) + (size_t)(nt < a.NT ? nt : a.NT - 1) * (a.KBW ? a.KBW : a.KB) * 256, 0, a.KB * 1024, 0x00020000);
  const int wvo = lane * 16;
  // staging role of this thread: rows sr + 16 WM p (p = 0..NP-1), 4 floats at column sc of the step's 64
  const int sr = tid >> 4, sc = (tid & 15) * 4;
  const int trows = a.rows - by * 64 < 64 ? a.rows - by * 64 : 64;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X) + (size_t)by * 64 * a.ldx, 0, trows * a.ldx * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? a.XT : a.X) + (size_t)by * 64 * a.ldx, 0, trows * a.ldx * 4, 0x00020000);
  int xvo[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) xvo[p] = ((sr + 16 * WM * p) * a.ldx + sc) * 4;
  const int NS = a.KB >> 2, NST = DUAL ? a.KBT >> 2 : 0;       // steps (an even number); steps that carry tangent rows (the first NST)
  // Prefetch distances: the activation tile TWO steps ahead (two register sets: tile t waits in set t & 1 until it is written
  // to LDS at the start of step t - 1), the weights SEVEN k-blocks ahead (ring of eight).  One step is 2,048 MFMA cycles
  // (~0.9 us): the first workgroup of an XCD to touch a line waits for HBM / MALL, not for L2.
  f32x4 xr[2][NP], tr[2][NP];
  auto gload = [&](int st, auto setc) {
    constexpr int S = decltype(setc)::value;
    const int sx = st < NS ? st : NS - 1;                       // (a row is ldx wide: the next row's data, not zeros, lies past K)
#pragma unroll
    for (int p = 0; p < NP; ++p) xr[S][p] = wide_bload(xrs, xvo[p], sx * 256);
    if constexpr (DUAL) {
      const int stt = st < NST ? st : NST - 1;
#pragma unroll
      for (int p = 0; p < NP; ++p) tr[S][p] = wide_bload(trs, xvo[p], stt * 256);
    }
  };
  auto swrite = [&](int buf, auto setc) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(&Xs[buf][sr + 16 * WM * p][sc]) = xr[S][p];
    if constexpr (DUAL) {
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(&Ts[buf][sr + 16 * WM * p][sc]) = tr[S][p];
    }
  };
  struct XF { f32x4 x[MTW], t[MTW]; };
  auto fread = [&](XF& f, int buf, int kbl) {
#pragma unroll
    for (int m = 0; m < MTW; ++m) f.x[m] = *reinterpret_cast<const f32x4*>(&Xs[buf][16 * (wm * MTW + m) + c][16 * kbl + 4 * g]);
    if constexpr (DUAL) {
#pragma unroll
      for (int m = 0; m < MTW; ++m) f.t[m] = *reinterpret_cast<const f32x4*>(&Ts[buf][16 * (wm * MTW + m) + c][16 * kbl + 4 * g]);
    }
  };
  f32x4 acc[1][MTW], acT[1][MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) { acc[0][m] = f32x4{0.f, 0.f, 0.f, 0.f}; acT[0][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto mac = [&](const f32x4& w, const XF& f, bool with_t) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < MTW; ++m) acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], f.x[m][s], acc[0][m], 0, 0, 0);
    if constexpr (DUAL) {
      if (with_t) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < MTW; ++m) acT[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], f.t[m][s], acT[0][m], 0, 0, 0);
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  f32x4 w[8];
  XF xf[2];
#ifdef WIDE_DBG_CLOCK
  const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // request order = the order the loop consumes in (the compiler's counted waits merge the loop entry with the back edge)
  gload(0, C0{});
  swrite(0, C0{});
  gload(1, C1{});
  gload(2, C0{});
#pragma unroll
  for (int j = 0; j < 7; ++j) w[j] = wide_bload(wr, wvo, j * 1024);
  WIDE_PIN();
  __syncthreads();
  fread(xf[0], 0, 0);
  asm volatile("" ::: "memory");       // (as in WIDE_PIN: keeps this read and the read-ahead of the loop's last block two loads)
  auto step = [&](int st, auto pc) {
    constexpr int P = decltype(pc)::value;                     // st & 1: LDS buffer of tile st, register set of tile st + 2
    const bool tt = DUAL && st < NST;
#ifndef WIDE_DBG_NO_X
    swrite(P ^ 1, std::integral_constant<int, P ^ 1>{});       // tile st + 1 (requested two steps ago)
    gload(st + 3, std::integral_constant<int, P ^ 1>{});
#endif
    WIDE_PIN();
#pragma unroll
    for (int kbl = 0; kbl < 4; ++kbl) {
#ifndef WIDE_DBG_NO_BARRIER
      if (kbl == 3) __syncthreads();
#endif
      // One k-block: MTW MFMAs per k-sub-step s, and ONE memory instruction after each group -- the weight request after the
      // first, one fragment read of the NEXT block after each.  Issued in a cluster at the block boundary they cost the matrix
      // pipe ~120 cycles per block (a wave issues in order and the pipe holds one MFMA: five memory instructions between two
      // MFMAs drain it); spread out, each one issues in the shadow of the MFMA before it.
      const f32x4& wk = w[4 * P + kbl];
      const XF& fc = xf[kbl & 1];
      XF& fn = xf[(kbl + 1) & 1];
      const int nbuf = kbl < 3 ? P : P ^ 1, nk = (kbl + 1) & 3;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[s4], fc.x[m][s4], acc[0][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef WIDE_DBG_NO_W
        if (s4 == 0) w[(4 * P + kbl + 7) & 7] = wide_bload(wr, wvo, (4 * st + kbl + 7) * 1024);
#endif
#ifndef WIDE_DBG_NO_FREAD
        if (s4 < MTW) fn.x[s4] = *reinterpret_cast<const f32x4*>(&Xs[nbuf][16 * (wm * MTW + s4) + c][16 * nk + 4 * g]);
#endif
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DUAL) {
        if (tt) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) acT[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[s4], fc.t[m][s4], acT[0][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s4 < MTW) fn.t[s4] = *reinterpret_cast<const f32x4*>(&Ts[nbuf][16 * (wm * MTW + s4) + c][16 * nk + 4 * g]);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  };
  for (int st = 0; st < NS; st += 2) { step(st, C0{}); step(st + 1, C1{}); }
#ifdef WIDE_DBG_CLOCK
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    wide_dbg_clk[0] = __builtin_readcyclecounter() - dbg_c0; wide_dbg_clk[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
  }
#endif
  gemm_epilogue<MTW, 1>(a, acc, acT, mt0, nt, MT, g, c);
}

static void launch_gemm(const Gemm& a, hipStream_t s) {
  const bool dual = a.XT != nullptr;
  const int MT = a.rows / 16;
  // 2 x 2 waves per workgroup; wave tile (16 MTW) rows x (16 NTW) features.  Small problems (this path runs ~1024 rows)
  // take the 32 x 32 wave tile so that the launch still covers the 1024 SIMDs.
  // tall launches (the batched time branch: 5 x rows) without whole 128-wide K steps keep round 1's 128 x 64 register-only tile;
  // wherever the LDS-staged kernel applies it is the faster one at every height (pines flow step 54.4 -> 51.8 ms when the batched
  // time-branch GEMMs moved to it)
  const bool no_lds_any = g_sw.wide_nolds;
  const bool big = (long long)MT * a.NT >= 4 * 4096 && !dual && (no_lds_any || (a.KB & 7) != 0);
  if (big) {
    dim3 grid((a.NT + 3) / 4, (MT + 7) / 8);
    hipLaunchKernelGGL((gemm_kernel<4, 2, false>), grid, dim3(256), 0, s, a);
  } else if (a.rows <= 512 && !g_sw.wide_no_small_tiles) {
    // few rows (the compacted attempts of a solve's tail, small batches): with 64 x 64 workgroup tiles a 256-row layer is 64
    // workgroups on 256 CUs and lasts as long as the 1024-row one.  Smaller tiles of the register-only kernel keep the chip
    // covered: 32 rows x 64 features up to 512 rows, 32 x 32 up to 256.
    if (a.rows <= 256) {
      dim3 grid((a.NT + 1) / 2, (MT + 1) / 2);
      if (dual) hipLaunchKernelGGL((gemm_kernel<1, 1, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_kernel<1, 1, false>), grid, dim3(256), 0, s, a);
    } else {
      dim3 grid((a.NT + 3) / 4, (MT + 1) / 2);
      if (dual) hipLaunchKernelGGL((gemm_kernel<1, 2, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_kernel<1, 2, false>), grid, dim3(256), 0, s, a);
    }
  } else {
    dim3 grid((a.NT + 3) / 4, (MT + 3) / 4);
    // the LDS-staged kernel needs an even number of whole 64-wide K steps (every width of the reference's pines networks); MFM_WIDE_NOLDS=1
    // keeps the register-only kernel for A/B measurements
    const bool no_lds = g_sw.wide_nolds;
    const bool lds = !no_lds && (a.KB & 7) == 0 && (!dual || ((a.KBT & 3) == 0 && a.KBT >= 4));
    const bool wm2 = !g_sw.wide_wm1;    // eight waves (two per SIMD) by default: 37.0 k vs 40.6 k cycles on 1024^3
    if (lds && wm2) {
      if (dual) hipLaunchKernelGGL((gemm_lds_kernel<true, 2>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gemm_lds_kernel<false, 2>), grid, dim3(512), 0, s, a);
    } else if (lds) {
      if (dual) hipLaunchKernelGGL((gemm_lds_kernel<true, 1>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_lds_kernel<false, 1>), grid, dim3(256), 0, s, a);
    } else if (dual) {
      if (a.KB % 8 == 0 && a.KBT % 8 == 0 && g_sw.wide_ring8) hipLaunchKernelGGL((gemm_kernel<2, 2, true, 8>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_kernel<2, 2, true>), grid, dim3(256), 0, s, a);
    } else {
      if (a.KB % 8 == 0 && g_sw.wide_ring8) hipLaunchKernelGGL((gemm_kernel<2, 2, false, 8>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_kernel<2, 2, false>), grid, dim3(256), 0, s, a);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradients: dW[k][n] = sum_r A[r][k] dZ[r][n], db[n] = sum_r dZ[r][n]; one wave per 64 x 64 block of dW.
// ---------------------------------------------------------------------------------------------------------------------
struct WgLayer { const float* A; int lda; const float* Z; int ldz; int K, N, Kp, Np, m_w, m_b;
                 int ks_true, ks_pad; };      // input columns [ks_true, ks_pad) are padding, column c >= ks_pad is canonical row c - ks_pad + ks_true (the first joint layer, mlp.hip.h: packed_row)
struct WgJob { int layer, kt, nt; };
struct WgArgs { WgLayer L[MLP_MAXL]; const WgJob* jobs; int n_jobs; int rows; float* grads;
                int* bad;      // non-null: raise *bad when a gradient element is not finite (the optimizer's apply_if_finite check)
                int n_full; }; // jobs [0, n_full) run one per wave (4 per workgroup); jobs [n_full, n_jobs) one per WORKGROUP, rows split over its waves

__global__ __launch_bounds__(256) void wgrad_kernel(WgArgs a) {
  // The jobs are equal (one 64 x 64 block over all rows) and two waves share a SIMD, so 2048 of them fill the chip exactly; the
  // pines widths have 2112.  Run one per wave, the 64 left over started when the first 2048 ended and ran ALONE on 16 CUs for a
  // whole job's time (3 job times where 2.06 are needed).  They are therefore given a workgroup each, their rows split over the
  // four waves and the partial blocks summed through LDS in a fixed order (deterministic): 2 + 1/4 job times.
  __shared__ f32x4 red[3][17][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
  const int n_full_wg = a.n_full >> 2;
  const bool split = (int)blockIdx.x >= n_full_wg;
  const int job = split ? a.n_full + ((int)blockIdx.x - n_full_wg) : (int)blockIdx.x * 4 + wave;
  if (job >= a.n_jobs) return;
  // The job and its layer descriptor are wave-uniform: read them as scalars ONCE.  Indexing the kernel-argument array with a
  // per-lane value made the compiler re-read `lda` / `ldz` from memory -- and wait with vmcnt(0) -- in front of every operand
  // load of the loop (round 1: 28 % MFMA occupancy).
  const int jl = __builtin_amdgcn_readfirstlane(a.jobs[job].layer), jkt = __builtin_amdgcn_readfirstlane(a.jobs[job].kt),
            jnt = __builtin_amdgcn_readfirstlane(a.jobs[job].nt);
  struct { int layer, kt, nt; } J = {jl, jkt, jnt};
  const WgLayer L = a.L[jl];
  const int lda = L.lda, ldz = L.ldz, rows = split ? a.rows >> 2 : a.rows, row0 = split ? wave * (a.rows >> 2) : 0;
  const int ac = 64 * J.kt + 4 * c, zc = 64 * J.nt + 4 * c;
  const bool av = ac < L.Kp, zv = zc < L.Np;
  // lanes beyond the padded widths read column 0 (valid memory): what they accumulate lands in rows / columns of the 64 x 64
  // block that are never stored (an MFMA keeps the M rows and the N columns of its operands apart)
  const float* ap = L.A + (size_t)(row0 + g) * lda + (av ? ac : 0);
  const float* zp = L.Z + (size_t)(row0 + g) * ldz + (zv ? zc : 0);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][4], bs = zero;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[s][u] = zero;
  auto mac = [&](const f32x4& af, const f32x4& zf) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[s][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], zf[u], acc[s][u], 0, 0, 0);
    bs += zf;
  };
  // rows is a multiple of 16: four steps of 4 chains per 16-row block.  Two operand sets in ping-pong (no register copies
  // at the end of an iteration: a copy would wait for the request just issued): the block after the one being multiplied is
  // requested first, unconditionally (the last request re-reads valid rows), so the wait in front of the MFMAs is a counted one
  struct Set { f32x4 a[4], z[4]; };
  auto request = [&](Set& q, int r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) q.a[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(r + 4 * i) * lda);
#pragma unroll
    for (int i = 0; i < 4; ++i) q.z[i] = *reinterpret_cast<const f32x4*>(zp + (size_t)(r + 4 * i) * ldz);
  };
  auto macs = [&](const Set& q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) mac(q.a[i], q.z[i]);
  };
  Set q0, q1;
  request(q0, 0);
  WIDE_PIN();
  int r = 0;
  for (; r + 32 <= rows; r += 32) {
    request(q1, r + 16);
    WIDE_PIN();                        // see gemm_kernel: keeps the requests one block ahead of their use
    macs(q0);
    request(q0, r + 32 < rows ? r + 32 : r + 16);
    WIDE_PIN();
    macs(q1);
  }
  if (r < rows) macs(q0);              // odd number of 16-row blocks
  if (split) {                         // (uniform over the workgroup)
    if (wave > 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) red[wave - 1][4 * s + u][lane] = acc[s][u];
      red[wave - 1][16][lane] = bs;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[s][u] += red[w][4 * s + u][lane];
      bs += red[w][16][lane];
    }
  }
  // acc[s][u][i] = dW[64 kt + 4 (4 g + i) + s][64 nt + 4 c + u]
  float* gw = a.grads + L.m_w;
  const bool vec = (L.N & 3) == 0;
  float chk = 0.f;                             // sum of 0 * element: NaN iff some element of this wave's block is not finite
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = 64 * J.kt + 4 * (4 * g + i) + s;
      if (k >= L.ks_true) {
        if (k < L.ks_pad) continue;
        k -= L.ks_pad - L.ks_true;
      }
      if (k >= L.K) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (zc + u < L.N) chk += 0.f * acc[s][u][i];
      if (vec) {
        if (zc < L.N) *reinterpret_cast<f32x4*>(gw + (size_t)k * L.N + zc) = f32x4{acc[s][0][i], acc[s][1][i], acc[s][2][i], acc[s][3][i]};
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (zc + u < L.N) gw[(size_t)k * L.N + zc + u] = acc[s][u][i];
      }
    }
  if (J.kt == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { bs[u] += __shfl_xor(bs[u], 16, 64); bs[u] += __shfl_xor(bs[u], 32, 64); }
    if (g == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (zc + u < L.N) { a.grads[L.m_b + zc + u] = bs[u]; chk += 0.f * bs[u]; }
    }
  }
  if (a.bad && __any(chk != chk) && lane == 0) atomicOr(a.bad, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// Row kernels (one wavefront per chain row, 4 rows per workgroup)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fourier_row(const float* __restrict__ fr, int F, int F2p, double te, float* out, int lane) {
  for (int col = lane; col < F; col += 64) {                 // :70-71: cos block then sin block
    double ft = (double)fr[col] * te;
    ft -= rint(ft);
    float sv, cv;
    sincospif(2.f * (float)ft, &sv, &cv);
    out[col] = cv; out[F + col] = sv;
  }
  for (int col = 2 * F + lane; col < F2p; col += 64) out[col] = 0.f;
}

struct FmPro {
  Key2 key_time, key_ref, key_gauss; uint32_t n_total, chain_offset;
  int rows, d, dp, F, F2p; float sigma; int cond_flow; double ref_std;
  const float* pos; const float* fourier;
  float* cond; float* tgt; float* ffat;
  int* flag_reset;       // non-null: the non-finite flag wgrad_kernel raises for this gradient, cleared here
};
// K3 batch construction (exe_flow_matching.py:151-169 / :139-147), same draws as fm.hip's prologue
__global__ __launch_bounds__(256) void fm_prologue_kernel(FmPro a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (a.flag_reset && blockIdx.x == 0 && threadIdx.x == 0) *a.flag_reset = 0;
  if (b >= a.rows) return;
  const uint32_t bg = a.chain_offset + (uint32_t)b, d = (uint32_t)a.d;
  const float tf = (float)uniform01(a.key_time, bg, a.n_total);                 // :154 / :142
  const double t = tf;
  const Key2 kref = split_at(a.key_ref, a.n_total, bg);                         // :155
  for (int col = lane; col < a.dp; col += 64) {
    float cv = 0.f, tv = 0.f;
    if (col < a.d) {
      const double x1v = a.pos[(size_t)b * a.d + col];
      double cnd, tg;
      if (a.cond_flow) {
        const double x0 = a.ref_std * normal64(kref, (uint32_t)col, d);                      // ref_dist.sample_model (:155)
        const double ne = normal64(a.key_gauss, bg * d + (uint32_t)col, a.n_total * d);       // :166
        cnd = (double)a.sigma * ne + t * x1v + (1.0 - t) * x0;                                // :167
        tg = x1v - x0;                                                                        // :168
      } else {
        const double x0 = normal64(a.key_ref, bg * d + (uint32_t)col, a.n_total * d);         // :143
        const double sds = 1.0 - (1.0 - (double)a.sigma) * t;                                 // :144
        cnd = t * x1v + sds * x0;                                                             // :145
        tg = x1v - (1.0 - (double)a.sigma) * x0;                                              // :146
      }
      cv = (float)cnd; tv = (float)tg;
    }
    a.cond[(size_t)b * a.dp + col] = cv;
    a.tgt[(size_t)b * a.dp + col] = tv;
  }
  fourier_row(a.fourier, a.F, a.F2p, t, a.ffat + (size_t)b * a.F2p, lane);
}

// clip(grad log pi(x)) of the UNTEMPERED target (:351) and, with a tangent, the masked Hessian-vector product.
// LGCP: KV = K^-1 (x - mu) and KZ = K^-1 z come from GEMMs with the packed K^-1.
struct TgtArgs {
  TargetDev T; float clip; int rows, d, dp;
  const float* X; const float* Z; const float* KV; const float* KZ;
  float* GC; float* HZ;
  const int* cmap;          // non-null: X / KV / GC / HZ rows are COMPACT (row j = chain cmap[j]); Z / KZ stay indexed by chain
  int diag;                 // exact trace: HZ = 1[|g| <= clip] * H_ii, the DIAGONAL of the target's Hessian (Z, KZ unused)
};
__global__ __launch_bounds__(256) void target_kernel(TgtArgs a) {
  const size_t n = (size_t)a.rows * a.dp;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    const int col = (int)(idx % a.dp);
    const size_t r0 = idx - col;
    const size_t zi = a.cmap ? (size_t)a.cmap[idx / a.dp] * a.dp + col : idx;       // the chain's row of the per-solve constants
    float gc = 0.f, hz = 0.f;
    if (col < a.d) {
      const float x = a.X[idx];
      float graw, hraw = 0.f;
      if (a.T.kind == MFM_TARGET_PHI4) {
        const float xl = col > 0 ? a.X[idx - 1] : 0.f, xr = col + 1 < a.d ? a.X[idx + 1] : 0.f;
        graw = -a.T.tbeta * (a.T.coef * (2.f * x - xl - xr) - x * (1.f - x * x) / a.T.coef);
        if (a.diag) hraw = -a.T.tbeta * (a.T.coef * 2.f - (1.f - 3.f * x * x) / a.T.coef);
        else if (a.Z) {
          const float v = a.Z[zi], vl = col > 0 ? a.Z[zi - 1] : 0.f, vr = col + 1 < a.d ? a.Z[zi + 1] : 0.f;
          hraw = -a.T.tbeta * (a.T.coef * (2.f * v - vl - vr) - (1.f - 3.f * x * x) * v / a.T.coef);
        }
      } else {
        const float ex = a.T.poisson_a * expf(x);
        graw = a.T.counts[col] - ex - a.KV[idx];
        if (a.diag) hraw = -ex - a.T.kdiag[col];
        else if (a.Z) hraw = -ex * a.Z[zi] - a.KZ[zi];
      }
      gc = clipf(graw, a.clip);
      const bool inside = !(a.clip > 0.f) || fabsf(graw) <= a.clip;
      hz = inside ? hraw : 0.f;
    }
    (void)r0;
    a.GC[idx] = gc;
    if (a.HZ) a.HZ[idx] = hz;
  }
}

// The Gaussian mixtures (distributions.py:42-77; d <= 8, the reference forces 2): one thread per chain row walks the modes
// (targets.hip.h: gmm_eval).  Tangent mode: H z by the closed form; diag mode: H_jj from one unit tangent per coordinate.
__global__ __launch_bounds__(256) void target_gmm_kernel(TgtArgs a) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= a.rows) return;
  const size_t o = (size_t)row * a.dp;
  const size_t zo = a.cmap ? (size_t)a.cmap[row] * a.dp : o;
  float x[8], z[8], g[8], hv[8], hd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = j < a.d ? a.X[o + j] : 0.f; z[j] = (a.Z && !a.diag && j < a.d) ? a.Z[zo + j] : 0.f; g[j] = hv[j] = hd[j] = 0.f; }
  double lp;
  gmm_eval<8>(a.T, x, &lp, g, (a.Z && !a.diag) ? z : nullptr, hv);
  if (a.diag) {
    for (int j = 0; j < a.d; ++j) {
      float e[8], t[8], g2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { e[i] = i == j ? 1.f : 0.f; t[i] = 0.f; }
      gmm_eval<8>(a.T, x, &lp, g2, e, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) if (i == j) hd[i] = t[i];
    }
  }
  for (int col = 0; col < a.dp; ++col) {
    float gc = 0.f, hz = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j == col && col < a.d) {
        gc = clipf(g[j], a.clip);
        const bool inside = !(a.clip > 0.f) || fabsf(g[j]) <= a.clip;
        hz = inside ? (a.diag ? hd[j] : hv[j]) : 0.f;
      }
    a.GC[o + col] = gc;
    if (a.HZ) a.HZ[o + col] = hz;
  }
}

// x[r][c] *= act'(.) for c < cols (in place, float4 wide): the sx half of d[sx | st]
struct ElemMask { int rows, cols, ld; float* x; const float* m; int kind, is_pre; };
__global__ void elem_mask_kernel(ElemMask a) {
  const int c4 = a.cols / 4;
  const size_t tot = (size_t)a.rows * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const size_t o = (i / c4) * a.ld + (i % c4) * 4;
    f32x4 v = *reinterpret_cast<f32x4*>(a.x + o);
    const f32x4 mk = *reinterpret_cast<const f32x4*>(a.m + o);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = a.is_pre ? mask_pre(mk[j], v[j], a.kind) : mask_out(mk[j], v[j], a.kind);
    *reinterpret_cast<f32x4*>(a.x + o) = v;
  }
}

// loss (:177-178) and the two output-side gradients: dv = 2 (v - target), dgate = dv * clip(grad log pi)
struct LossArgs { int rows, rows_valid, d, dp; const float* out; const float* gate; const float* gc; const float* tgt; float* dv; float* dg; double* part; };
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
  __shared__ double sm[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x * 4 + w;
  float loc = 0.f;
  if (b < a.rows) {
    for (int col = lane; col < a.dp; col += 64) {
      const size_t o = (size_t)b * a.dp + col;
      float dv = 0.f, dg = 0.f;
      if (col < a.d) {
        const float gc = a.gc[o];
        const float r = b < a.rows_valid ? a.out[o] + a.gate[o] * gc - a.tgt[o] : 0.f;      // (padding rows of the chain shard)
        loc += r * r;
        dv = 2.f * r; dg = dv * gc;
      }
      if (a.dv) { a.dv[o] = dv; a.dg[o] = dg; }
    }
  }
  const double s = wave_sum((double)loc);
  if (lane == 0) sm[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) a.part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// ---------------------------------------------------------------------------------------------------------------------
// Adaptive Dopri5, per-row state in global memory
// ---------------------------------------------------------------------------------------------------------------------
struct RowState { float *t, *dt, *h0, *d1, *ell, *kl /* [7][rows] */; int *natt, *done; };

__device__ static const float W_TAB[8][7] = {   // [phase][j]: input = y + h sum_j TAB[phase][j] k_j ; last column: time fraction
    {0, 0, 0, 0, 0, 0, 0.f},
    {1, 0, 0, 0, 0, 0, 1.f},
    {1.f / 5, 0, 0, 0, 0, 0, 1.f / 5},
    {3.f / 40, 9.f / 40, 0, 0, 0, 0, 3.f / 10},
    {44.f / 45, -56.f / 15, 32.f / 9, 0, 0, 0, 4.f / 5},
    {19372.f / 6561, -25360.f / 2187, 64448.f / 6561, -212.f / 729, 0, 0, 8.f / 9},
    {9017.f / 3168, -355.f / 33, 46732.f / 5247, 49.f / 176, -5103.f / 18656, 0, 1.f},
    {35.f / 384, 0, 500.f / 1113, 125.f / 192, -2187.f / 6784, 11.f / 84, 1.f}};
__device__ static const float W_E[7] = {(float)(35.0 / 384 - 1951.0 / 21600), 0.f, (float)(500.0 / 1113 - 22642.0 / 50085),
                                        (float)(125.0 / 192 - 451.0 / 720), (float)(-2187.0 / 6784 + 12231.0 / 42400),
                                        (float)(11.0 / 84 - 649.0 / 6300), (float)(-1.0 / 60)};
__device__ static const float W_M[7] = {(float)(6025192743.0 / 30085553152.0 / 2), 0.f, (float)(51252292925.0 / 65400821598.0 / 2),
                                        (float)(-2691868925.0 / 45128329728.0 / 2), (float)(187940372067.0 / 1594534317056.0 / 2),
                                        (float)(-1776094331.0 / 19743644256.0 / 2), (float)(11237099.0 / 235043384.0 / 2)};

// Parity instrumentation (mfm_debug_replay): the solve runs on a PRESCRIBED step sequence and records its own controller's
// values; same meaning as `Replay` in ode.hip (dt == nullptr: off, production).
struct WReplay {
  const float* dt; const uint8_t* acc; float* ratio; float* dt_own; int cap, n, solve, row0;
  double* diag;             // flow step only, may be null: [n][4] as in ode.hip
  __device__ __forceinline__ size_t at(int row, int j) const { return ((size_t)solve * n + row0 + row) * cap + j; }
};

constexpr int JT_SLICES = 8;          // partial sums per chain of the exact Jacobian trace (jt_trace_kernel), added in index order
struct OdeBuf {
  int rows, d, dp, F, F2p, sign;
  float rtol, atol; int max_attempts;
  RowState rs;
  float* Y;            // [rows][dp] state x
  float* K;            // [7][rows][dp] stage derivatives of x
  float* X;            // [rows][dp] stage input (value rows of the field evaluation)
  const float* Z;      // [rows][dp] probe (zero padded)
  float* ffat;         // [rows][F2p]
  const float* fourier;
  const float* out; const float* outT; const float* gate; const float* gc; const float* hz;   // results of the evaluation
  int* n_active;
  const int* cpos;     // non-null: chain b's row in the COMPACT evaluation buffers (X, ffat, out, outT, gate, gc, hz), -1: not integrating
  const float* trp;    // non-null: EXACT trace (exe_flow_matching.py:216-217) -- trp[cp * JT_SLICES + s] = partial sums of trace(d nn_xt / d x)
                       // of compact row cp (jt_trace_kernel), hz = the masked Hessian DIAGONAL; Z is null
  int tb_rows;         // > 0: time-branch batching -- at phase 2 stage_prep writes the Fourier rows of the attempt's FIVE distinct stage
                       // times to ffat[(s * tb_rows + row)] (slot s), and no Fourier rows in the other phases
  WReplay rp;
};

// The row kernels walk a chain's d columns with one wavefront.  With d a multiple of 4 (every example of the reference) a lane takes
// FOUR consecutive columns per step -- 128-bit loads and stores of the seven stage derivatives, the state and the stage input --
// where the one-column walk of round 2 issued sixteen dependent 4-byte accesses per array and lane (stage_prep 23.5 us,
// stage_finish 16.7 us per call at 1024 x 1024: 14 % of a field evaluation's time).  Rows that have reached t = 1 are skipped:
// their state no longer changes and nothing reads what an evaluation leaves for them.
template <typename T> struct RowV;
template <> struct RowV<float> {
  static constexpr int W = 1;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float sum(float v) { return v; }
  static __device__ __forceinline__ float absv(float v) { return fabsf(v); }
  static __device__ __forceinline__ float maxv(float x, float y) { return fmaxf(x, y); }
  static __device__ __forceinline__ float zero() { return 0.f; }
};
template <> struct RowV<f32x4> {
  static constexpr int W = 4;
  static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  static __device__ __forceinline__ float sum(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
  static __device__ __forceinline__ f32x4 absv(f32x4 v) { return f32x4{fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3])}; }
  static __device__ __forceinline__ f32x4 maxv(f32x4 x, f32x4 y) { return f32x4{fmaxf(x[0], y[0]), fmaxf(x[1], y[1]), fmaxf(x[2], y[2]), fmaxf(x[3], y[3])}; }
  static __device__ __forceinline__ f32x4 zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
};

// stage input of phase p: X = y + h sum_j TAB[p][j] k_j, Fourier features of the stage time (:70-71, :229)
template <typename T>
__global__ __launch_bounds__(256) void stage_prep_kernel(OdeBuf a, int phase) {
  using V = RowV<T>;
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.rows) return;
  if (phase >= 2 && a.rs.done[b]) return;
  const int cp = a.cpos ? a.cpos[b] : b;
  if (cp < 0) return;
  float cf[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) cf[j] = W_TAB[phase][j];
  const float h = phase == 1 ? a.rs.h0[b] : a.rs.dt[b];
  const float ts = a.rs.t[b] + h * cf[6];
  const size_t o0 = (size_t)b * a.dp, ks = (size_t)a.rows * a.dp, oc = (size_t)cp * a.dp;
  for (int col = lane * V::W; col < a.dp; col += 64 * V::W) {
    T v = V::zero();
    if (col < a.d) {
      T acc = V::zero();
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (cf[j] != 0.f) acc += cf[j] * V::ld(a.K + j * ks + o0 + col);
      v = V::ld(a.Y + o0 + col) + h * acc;
    }
    V::st(a.X + oc + col, v);
  }
  if (a.tb_rows > 0) {
    if (phase == 2) {
      const float t0 = a.rs.t[b];
#pragma unroll 1
      for (int sl = 0; sl < 5; ++sl) {
        const float tsl = t0 + h * W_TAB[2 + sl][6];
        const double te = a.sign > 0 ? (double)tsl : 1.0 - (double)tsl;
        fourier_row(a.fourier, a.F, a.F2p, te, a.ffat + ((size_t)sl * a.tb_rows + cp) * a.F2p, lane);
      }
    }
  } else if (phase != 7) {        // stages 6 and 7 share t + dt: the time branch of stage 6 is still valid
    const double te = a.sign > 0 ? (double)ts : 1.0 - (double)ts;
    fourier_row(a.fourier, a.F, a.F2p, te, a.ffat + (size_t)cp * a.F2p, lane);
  }
}

// end of a field evaluation: k_dst = +-v, kl_dst = -+ z . J z, then the phase-specific part of the state machine
// (phase 0 / 1: initial step size, Hairer II.4 order 4; phase 7: error norm, accept / reject, step-size controller,
// 4th-order interpolation at t = 1, FSAL) -- the same float32 arithmetic as ode_solve() in ode.hip.
template <typename T>
__global__ __launch_bounds__(256) void stage_finish_kernel(OdeBuf a, int phase) {
  using V = RowV<T>;
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.rows) return;
  if (phase >= 2 && a.rs.done[b]) return;
  const int cp = a.cpos ? a.cpos[b] : b;
  if (cp < 0) return;
  const int dst = phase == 0 ? 0 : phase - 1 + (phase == 1 ? 1 : 0);
  const size_t o0 = (size_t)b * a.dp, ks = (size_t)a.rows * a.dp, oc = (size_t)cp * a.dp;      // state row / compact evaluation row
  const int R = a.rows;
  float dpart = 0.f;
  for (int col = lane * V::W; col < a.d; col += 64 * V::W) {
    const size_t o = o0 + col, e = oc + col;
    const T gt = V::ld(a.gate + e);
    const T v = V::ld(a.out + e) + gt * V::ld(a.gc + e);
    V::st(a.K + dst * ks + o, a.sign > 0 ? v : -v);
    if (a.Z) dpart += V::sum(V::ld(a.Z + o) * (V::ld(a.outT + e) + gt * V::ld(a.hz + e)));
    else if (a.trp) dpart += V::sum(gt * V::ld(a.hz + e));      // d/dx_i [gate_i clip(g_i(x))] = gate_i 1[|g_i| <= clip] H_ii
  }
  dpart = wave_sum(dpart);
  if (a.trp) {
#pragma unroll
    for (int sl = 0; sl < JT_SLICES; ++sl) dpart += a.trp[(size_t)cp * JT_SLICES + sl];      // fixed order: deterministic
  }
  const float dl = a.sign > 0 ? -dpart : dpart;                 // :218 / :239
  if (lane == 0) a.rs.kl[dst * R + b] = dl;
  const float atol = a.atol, rtol = a.rtol;
  if (phase == 0) {
    float p0 = 0.f, p1 = 0.f;
    for (int col = lane * V::W; col < a.d; col += 64 * V::W) {
      const T y = V::ld(a.Y + o0 + col), k0 = V::ld(a.K + o0 + col);      // same lane wrote K[0] above
      const T sc = atol + V::absv(y) * rtol;
      const T a0 = y / sc, a1 = k0 / sc;
      p0 += V::sum(a0 * a0); p1 += V::sum(a1 * a1);
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1);
    if (lane == 0) {
      const float a1 = dl / atol;
      const float d0 = sqrtf(p0), d1 = sqrtf(p1 + a1 * a1);
      a.rs.h0[b] = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
      a.rs.d1[b] = d1;
    }
  } else if (phase == 1) {
    float p2 = 0.f;
    for (int col = lane * V::W; col < a.d; col += 64 * V::W) {
      const T y = V::ld(a.Y + o0 + col);
      const T sc = atol + V::absv(y) * rtol;
      const T a2 = (V::ld(a.K + ks + o0 + col) - V::ld(a.K + o0 + col)) / sc;
      p2 += V::sum(a2 * a2);
    }
    p2 = wave_sum(p2);
    if (lane == 0) {
      const float h0 = a.rs.h0[b], d1 = a.rs.d1[b];
      const float a2 = (dl - a.rs.kl[b]) / atol;
      const float d2 = sqrtf(p2 + a2 * a2) / h0;
      const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
      float dt = fminf(100.f * h0, h1);
      if (a.rp.dt) { const size_t o = a.rp.at(b, 0); a.rp.dt_own[o] = dt; dt = a.rp.dt[o]; }
      a.rs.dt[b] = dt;
      if (dt > 0.f) atomicAdd(a.n_active, 1);
    }
  } else if (phase == 7) {
    const float dti = a.rs.dt[b], t0 = a.rs.t[b], ell0 = a.rs.ell[b];
    const int na = a.rs.natt[b], dn = a.rs.done[b];
    float kl[7];
#pragma unroll
    for (int j = 0; j < 6; ++j) kl[j] = a.rs.kl[j * R + b];
    kl[6] = dl;
    float e2 = 0.f;
    for (int col = lane * V::W; col < a.d; col += 64 * V::W) {
      const size_t o = o0 + col;
      T er = V::zero();
#pragma unroll
      for (int j = 0; j < 7; ++j) er += W_E[j] * V::ld(a.K + j * ks + o);
      er *= dti;
      const T tol = atol + rtol * V::maxv(V::absv(V::ld(a.Y + o)), V::absv(V::ld(a.X + oc + col)));
      const T rr = er / tol;
      e2 += V::sum(rr * rr);
    }
    e2 = wave_sum(e2);
    const bool active = !dn && na < a.max_attempts && dti > 0.f;
    float sl = 0.f, el = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) sl += W_TAB[7][j] * kl[j];
#pragma unroll
    for (int j = 0; j < 7; ++j) el += W_E[j] * kl[j];
    const float l1 = ell0 + dti * sl;
    el *= dti;
    const float tol = atol + rtol * fmaxf(fabsf(ell0), fabsf(l1));
    const float rr = el / tol;
    const float ratio = sqrtf((e2 + rr * rr) / (float)(a.d + 1));
    bool acc = active && ratio <= 1.f;
    const float dfac = ratio < 1.f ? 1.f : 0.2f;
    const float fac = fminf(10.f, fmaxf(0.9f * powf(ratio, -0.2f), dfac));
    float ndt = fmaxf(ratio == 0.f ? dti * 10.f : dti * fac, 0.f);
    if (a.rp.dt && active) {               // every lane of the wavefront takes the same (uniform) decision
      const bool in = na < a.rp.cap, nx = na + 1 < a.rp.cap;
      const size_t o = a.rp.at(b, in ? na : 0);
      if (lane == 0 && in) { a.rp.ratio[o] = ratio; if (nx) a.rp.dt_own[o + 1] = ndt; }
      acc = in && a.rp.acc[o] != 0;
      ndt = nx ? a.rp.dt[o + 1] : 0.f;
    }
    float t_n = t0, ell_n = ell0, kl0_n = kl[0];
    int dn_n = dn;
    if (acc) {
      const float tn = t0 + dti;
      if (tn >= 1.f) {
        const float sfrac = (1.f - t0) / (tn - t0);
        float lm = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) lm += W_M[j] * kl[j];
        {
          const float y0 = ell0, y1 = l1, ym = y0 + dti * lm, f0 = dti * kl[0], f1 = dti * kl[6];
          const float pa = -2.f * f0 + 2.f * f1 - 8.f * y0 - 8.f * y1 + 16.f * ym;
          const float pb = 5.f * f0 - 3.f * f1 + 18.f * y0 + 14.f * y1 - 32.f * ym;
          const float pc = -4.f * f0 + f1 - 11.f * y0 - 5.f * y1 + 16.f * ym;
          ell_n = (((pa * sfrac + pb) * sfrac + pc) * sfrac + f0) * sfrac + y0;
        }
        for (int col = lane * V::W; col < a.d; col += 64 * V::W) {
          const size_t o = o0 + col;
          T km = V::zero();
#pragma unroll
          for (int j = 0; j < 7; ++j) km += W_M[j] * V::ld(a.K + j * ks + o);
          const T x0 = V::ld(a.Y + o), x1 = V::ld(a.X + oc + col), xm = x0 + dti * km, g0 = dti * V::ld(a.K + o), g1 = dti * V::ld(a.K + 6 * ks + o);
          const T qa = -2.f * g0 + 2.f * g1 - 8.f * x0 - 8.f * x1 + 16.f * xm;
          const T qb = 5.f * g0 - 3.f * g1 + 18.f * x0 + 14.f * x1 - 32.f * xm;
          const T qc = -4.f * g0 + g1 - 11.f * x0 - 5.f * x1 + 16.f * xm;
          V::st(a.Y + o, (((qa * sfrac + qb) * sfrac + qc) * sfrac + g0) * sfrac + x0);
        }
        dn_n = 1;
      } else {
        ell_n = l1;
        for (int col = lane * V::W; col < a.d; col += 64 * V::W) {
          const size_t o = o0 + col;
          V::st(a.Y + o, V::ld(a.X + oc + col));
          V::st(a.K + o, V::ld(a.K + 6 * ks + o));               // FSAL
        }
        kl0_n = kl[6];
      }
      t_n = tn;
    }
    if (lane == 0) {
      const float dt_n = active ? ndt : dti;
      const int na_n = active ? na + 1 : na;
      a.rs.t[b] = t_n; a.rs.dt[b] = dt_n; a.rs.ell[b] = ell_n; a.rs.kl[b] = kl0_n; a.rs.natt[b] = na_n; a.rs.done[b] = dn_n;
      if (!dn_n && na_n < a.max_attempts && dt_n > 0.f) atomicAdd(a.n_active, 1);
    }
  }
}

// Compaction map of an attempt: the chains still integrating (the predicate stage_finish counts n_active with), in chain order.
// One workgroup; cmap[j] = chain of compact row j, cpos[chain] = its compact row or -1.
__global__ __launch_bounds__(1024) void compact_map_kernel(RowState rs, int rows, int max_attempts, int* cmap, int* cpos) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int r0 = 0; r0 < rows; r0 += 1024) {
    const int b = r0 + (int)threadIdx.x;
    const bool act = b < rows && !rs.done[b] && rs.natt[b] < max_attempts && rs.dt[b] > 0.f;
    const unsigned long long m = __ballot(act);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    const int j = off + __popcll(m & ((1ull << lane) - 1ull));
    if (b < rows) cpos[b] = act ? j : -1;
    if (act) cmap[j] = b;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base += t; }
    __syncthreads();
  }
}
// dst[j][:] = src[cmap[j]][:] for j < n (128-bit copies; ld multiple of 4)
__global__ void gather_rows_c_kernel(const float* src, const int* cmap, int n, int ld, float* dst) {
  const int l4 = ld >> 2;
  const size_t tot = (size_t)n * l4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const size_t j = i / l4, c = i - j * l4;
    reinterpret_cast<f32x4*>(dst + j * ld)[c] = reinterpret_cast<const f32x4*>(src + (size_t)cmap[j] * ld)[c];
  }
}

__global__ void ode_init_kernel(RowState rs, int rows) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= rows) return;
  rs.t[b] = 0.f; rs.dt[b] = 0.f; rs.h0[b] = 0.f; rs.d1[b] = 0.f; rs.ell[b] = 0.f; rs.natt[b] = 0; rs.done[b] = 0;
  for (int j = 0; j < 7; ++j) rs.kl[j * rows + b] = 0.f;
}

// rows [n][d] (caller layout) <-> [n][dp] zero-padded work layout
__global__ void pad_rows_kernel(const float* src, int n, int d, int dp, float* dst) {
  const size_t tot = (size_t)n * dp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const int col = (int)(i % dp);
    dst[i] = col < d ? src[(i / dp) * d + col] : 0.f;
  }
}
__global__ void unpad_rows_kernel(const float* src, int n, int d, int dp, float* dst) {
  const size_t tot = (size_t)n * d;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const int col = (int)(i % d);
    dst[i] = src[(i / d) * dp + col];
  }
}
// v = out + gate * gc ; J z = outT + gate * hz  (mfm_vf_apply)
__global__ void vf_out_kernel(int n, int d, int dp, const float* out, const float* outT, const float* gate, const float* gc, const float* hz,
                              float* v, float* jvp) {
  const size_t tot = (size_t)n * d;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const size_t o = (i / d) * dp + (i % d);
    v[i] = out[o] + gate[o] * gc[o];
    if (jvp) jvp[i] = outT[o] + gate[o] * hz[o];
  }
}

// ---- flow-MH step glue (exe_flow_matching.py:246-278) ----------------------------------------------------------------
struct FlowGlue {
  int mode; Key2 key; uint32_t n_total, chain_offset; double beta; float ref_std;
  int rows, d, dp;
  TargetDev T;
  float* Y;                  // [rows][dp]: u0 after the inverse solve -> proposal -> x' after the forward solve
  const float* zgen;         // [rows][d]
  const float* ell;          // log-det of the solve that just finished
  float* vol0; float* lqref; int* natt_tot; const int* natt;
  const float* KV;           // LGCP: K^-1 (x' - mu)
  float* pos; double* logp; float* grad; float* acc_prob; uint8_t* accepted; float* proposed; int* nsteps;
  double* diag;              // replay instrumentation (may be null)
};
// after the inverse solve: keep vol0, build the latent proposal (:268 random walk / :249 independent)
__global__ __launch_bounds__(256) void flow_propose_kernel(FlowGlue a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.rows) return;
  const float scale = 2.38f / sqrtf((float)a.d);                                                  // :262
  float r0 = 0.f, r1 = 0.f;
  for (int col = lane; col < a.d; col += 64) {
    const size_t o = (size_t)b * a.dp + col;
    const float nz = a.zgen[(size_t)b * a.d + col], u0 = a.Y[o];
    if (a.mode == MFM_FLOW_RWMH) a.Y[o] = u0 + scale * nz;
    else { const float up = a.ref_std * nz; r0 += u0 * u0; r1 += up * up; a.Y[o] = up; }             // ref_dist.sample_model (:249)
  }
  r0 = wave_sum(r0); r1 = wave_sum(r1);
  if (lane == 0) {
    a.vol0[b] = a.ell[b];
    a.lqref[b] = a.mode == MFM_FLOW_IMH ? -0.5f * (r0 - r1) / (a.ref_std * a.ref_std) : 0.f;      // :254-255
    a.natt_tot[b] = a.natt[b];
  }
}
// after the forward solve: tempered target at the proposal (:270 / :252), unclipped acceptance ratio, accept / reject
__global__ __launch_bounds__(256) void flow_accept_kernel(FlowGlue a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.rows) return;
  const float* y = a.Y + (size_t)b * a.dp;
  double lpn;
  float gmm_g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (a.T.kind == MFM_TARGET_PHI4) {
    double part = 0.0;
    for (int col = lane; col < a.d; col += 64) {
      const double x = y[col];
      const double xr = col + 1 < a.d ? (double)y[col + 1] : 0.0;
      const double dr = xr - x;
      double u = dr * dr;
      if (col == 0) u += x * x;
      const double q = 1.0 - x * x;
      part += -(double)a.T.tbeta * (0.5 * (double)a.T.coef * u + q * q / (4.0 * (double)a.T.coef));
    }
    lpn = a.beta * wave_sum(part);
  } else if (a.T.kind == MFM_TARGET_GMM) {      // every lane walks the modes of its chain's row (d <= 8): tempered density beta * logprob
    float xg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xg[j] = j < a.d ? y[j] : 0.f;
    double lp;
    gmm_eval<8>(a.T, xg, &lp, gmm_g);
    lpn = a.beta * lp;
  } else {
    double lik = 0.0, quad = 0.0;
    for (int col = lane; col < a.d; col += 64) {
      const float xv = y[col], yv = a.KV[(size_t)b * a.dp + col], ex = expf(xv);
      lik += (double)xv * (double)a.T.counts[col] - (double)a.T.poisson_a * (double)ex;
      quad += (double)(xv - a.T.mu) * (double)yv;
    }
    lpn = a.beta * wave_sum(lik) - 0.5 * wave_sum(quad) + (double)a.T.log_norm;
  }
  const Key2 kb = split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b);                      // :303
  const double la = lpn - (double)a.ell[b] - a.logp[b] - (double)a.vol0[b] + (double)a.lqref[b];
  const double ap = exp(la);
  const double u = uniform01(split_at(kb, 4, 1), 0, 1);
  const bool acc = u <= ap;                      // NaN compares false -> reject
  for (int col = lane; col < a.d; col += 64) {
    const size_t o = (size_t)b * a.d + col;
    const float xv = y[col];
    if (a.proposed) a.proposed[o] = xv;
    if (acc) {
      float gv;
      if (a.T.kind == MFM_TARGET_PHI4) {
        const float xl = col > 0 ? y[col - 1] : 0.f, xr = col + 1 < a.d ? y[col + 1] : 0.f;
        gv = (float)a.beta * (-a.T.tbeta * (a.T.coef * (2.f * xv - xl - xr) - xv * (1.f - xv * xv) / a.T.coef));
      } else if (a.T.kind == MFM_TARGET_GMM) {
        gv = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j == col) gv = (float)a.beta * gmm_g[j];
      } else {
        gv = (float)a.beta * (a.T.counts[col] - a.T.poisson_a * expf(xv)) - a.KV[(size_t)b * a.dp + col];
      }
      a.pos[o] = xv; a.grad[o] = gv;
    }
  }
  if (lane == 0) {
    if (acc) a.logp[b] = lpn;
    if (a.acc_prob) a.acc_prob[b] = (float)ap;
    if (a.accepted) a.accepted[b] = acc ? 1 : 0;
    if (a.nsteps) a.nsteps[b] = a.natt_tot[b] + a.natt[b];
    if (a.diag) { double* o = a.diag + 4 * (size_t)b; o[0] = a.vol0[b]; o[1] = a.ell[b]; o[2] = lpn; o[3] = la; }
  }
}

// ---- MALA step / init for the LGCP target beyond the fused tile kernel's dimension (lgcp.hip handles d <= 1024; the
//      reference's own pines default is the 40 x 40 grid, multi_modal.py:89).  Same arithmetic as mala_lgcp_kernel, split
//      around the K^-1 GEMM: propose -> K^-1 (x' - mu) -> energies / accept. ----
struct LgcpMala {
  TargetDev T; int mode; Key2 key; const uint32_t* keys; uint32_t n_total, chain_offset; int rows, d, dp; double beta, eps; int textbook;
  float* Y; const float* KV; double* th1;
  float* pos; double* logp; float* grad; float* acc_prob; uint8_t* accepted; float* proposed; float* prop_weight;
};
__global__ __launch_bounds__(256) void lgcp_propose_kernel(LgcpMala a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.rows) return;
  const Key2 kb = a.keys ? Key2{a.keys[2 * b], a.keys[2 * b + 1]} : split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b);           // exe_flow_matching.py:303
  const Key2 k_int = split_at(kb, 2, 0);                                              // mala.py:93
  const double s2e = sqrt(2.0 * a.eps);
  double th1 = 0.0;
  for (int col = lane; col < a.dp; col += 64) {
    float xn = 0.f;
    if (col < a.d) {
      const size_t o = (size_t)b * a.d + col;
      const float x = a.pos[o];
      if (a.mode == 1) {
        const double th = s2e * normal64(k_int, (uint32_t)col, (uint32_t)a.d);        // util.py:80-82
        th1 += th * th;
        xn = (float)((double)x + a.eps * (double)a.grad[o] + th);                     // diffusions.py:25-30
      } else {
        xn = x;
      }
    }
    a.Y[(size_t)b * a.dp + col] = xn;
  }
  th1 = wave_sum(th1);
  if (lane == 0) a.th1[b] = th1;
}
__global__ __launch_bounds__(256) void lgcp_accept_kernel(LgcpMala a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.rows) return;
  double lik = 0.0, quad = 0.0, th2 = 0.0;
  for (int col = lane; col < a.d; col += 64) {
    const float xv = a.Y[(size_t)b * a.dp + col], y = a.KV[(size_t)b * a.dp + col], ex = expf(xv);
    const float gv = (float)a.beta * (a.T.counts[col] - a.T.poisson_a * ex) - y;
    lik += (double)xv * (double)a.T.counts[col] - (double)a.T.poisson_a * (double)ex;
    quad += (double)(xv - a.T.mu) * (double)y;
    const double t = (double)a.pos[(size_t)b * a.d + col] - (double)xv - a.eps * (double)gv;
    th2 += t * t;
  }
  lik = wave_sum(lik); quad = wave_sum(quad); th2 = wave_sum(th2);
  const double lpn = a.beta * lik - 0.5 * quad + (double)a.T.log_norm;
  bool acc = true;
  if (a.mode == 1) {
    const double lp = a.logp[b], inv4e = 0.25 / a.eps;
    const double new_E = -lp + inv4e * a.th1[b], prev_E = -lpn + inv4e * th2;      // mala.py:68-79, proposal.py:157-158
    double delta = prev_E - new_E;                                                 // proposal.py:104
    if (a.textbook) delta = -delta;
    if (isnan(delta)) delta = -INFINITY;                                           // proposal.py:105
    const double p = fmin(exp(delta), 1.0);                                        // proposal.py:178
    const Key2 kb = a.keys ? Key2{a.keys[2 * b], a.keys[2 * b + 1]} : split_at(a.key, a.n_total, a.chain_offset + (uint32_t)b);
    acc = uniform01(split_at(kb, 2, 1), 0, 1) < p;                                 // proposal.py:179
    if (lane == 0) {
      if (a.acc_prob) a.acc_prob[b] = (float)p;
      if (a.accepted) a.accepted[b] = acc ? 1 : 0;
      if (a.prop_weight) a.prop_weight[b] = (float)exp(lpn + inv4e * th2);         // mala.py:104-113
    }
  }
  for (int col = lane; col < a.d; col += 64) {
    const size_t o = (size_t)b * a.d + col;
    const float xv = a.Y[(size_t)b * a.dp + col];
    if (a.mode == 1 && a.proposed) a.proposed[o] = xv;
    if (acc) {
      const float gv = (float)a.beta * (a.T.counts[col] - a.T.poisson_a * expf(xv)) - a.KV[(size_t)b * a.dp + col];
      if (a.mode == 1) a.pos[o] = xv;
      a.grad[o] = gv;
    }
  }
  if (lane == 0 && acc) a.logp[b] = lpn;
}

// ---- EXACT trace of the Jacobian (exe_flow_matching.py:216-217, :236-237: jnp.trace(jax.jacfwd(v)(x))) --------------------------
// The reference pushes the d basis vectors through the network.  With row vectors, d nn_xt / d x = W_x1 D1 W_x2 D2 W_j1x D3 W_j2 D4
// W_out (D_l = diag act'(pre-activation of layer l) of the chain, W_j1x = the sx rows of the joint layer), and the trace is cyclic:
//     trace = tr( D1 W_x2 D2 W_j1x D3 W_j2 D4 E ),   E = W_out W_x1  [hj2 x hx1], computed once per solve,
// i.e. hx1 tangent rows per chain instead of d, none of them through the two d-wide layers: row i starts as D1[i] W_x2[i, :] D2
// (jt_seed_kernel: element-wise), takes two GEMMs with the chain's masks in the epilogue (the layer GEMM kernels; mask_rdiv maps
// the hx1 rows of a chain to its mask row) and is contracted with column i of E (jt_trace_kernel).  Per chain and evaluation
// 2 hx1 (hx2 hj1 + hj1 hj2) flop -- 4.3 GFLOP at hidden 1024, whatever d is (d-tangent form: 10 GFLOP at d = 1600).  The gate term
// of the field, gate_i clip(g_i(x)), adds gate_i 1[|g_i| <= clip] H_ii (target_kernel's diag mode; stage_finish sums it).
struct JtSeed { int chains, hx1, hx2, row0; const float* W; int rows_w, cols_w;      // canonical kernel [rows_w][cols_w] (true sizes; hx1 / hx2: padded)
                const float* m1; int ld1; const float* m2; int ld2; int kind, is_pre; float* X; };
__global__ __launch_bounds__(256) void jt_seed_kernel(JtSeed a) {
  const int c4 = a.hx2 / 4;
  const size_t tot = (size_t)a.chains * a.hx1 * c4;
  const bool vec = (a.cols_w & 3) == 0;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * 256) {
    const int j4 = (int)(idx % c4);
    const size_t r = idx / c4;
    const int i = (int)(r % a.hx1), b = (int)(r / a.hx1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < a.rows_w) {           // a padded unit of x1 has no row (and act'(0) is not 0 for every activation)
      if (vec && 4 * j4 + 3 < a.cols_w) v = *reinterpret_cast<const f32x4*>(a.W + (size_t)i * a.cols_w + 4 * j4);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * j4 + j < a.cols_w) v[j] = a.W[(size_t)i * a.cols_w + 4 * j4 + j];
      }
    }
    const float s1 = a.m1[(size_t)(a.row0 + b) * a.ld1 + i];
    const f32x4 s2 = *reinterpret_cast<const f32x4*>(a.m2 + (size_t)(a.row0 + b) * a.ld2 + 4 * j4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = a.is_pre ? mask_pre(s1, v[j], a.kind) : mask_out(s1, v[j], a.kind);
      v[j] = a.is_pre ? mask_pre(s2[j], v[j], a.kind) : mask_out(s2[j], v[j], a.kind);
    }
    *reinterpret_cast<f32x4*>(a.X + r * a.hx2 + 4 * j4) = v;
  }
}
struct JtTrace { int hx1, hj2, row0; const float* Y; const float* ET; float* trp; };
__global__ __launch_bounds__(256) void jt_trace_kernel(JtTrace a) {      // grid (JT_SLICES, chains of the chunk)
  __shared__ double sm[4];
  const int b = blockIdx.y, sl = blockIdx.x, c4 = a.hj2 / 4;
  const int i0 = (int)((long long)a.hx1 * sl / JT_SLICES), i1 = (int)((long long)a.hx1 * (sl + 1) / JT_SLICES);
  float acc = 0.f;
  for (size_t idx = threadIdx.x; idx < (size_t)(i1 - i0) * c4; idx += 256) {
    const int i = i0 + (int)(idx / c4), k4 = (int)(idx % c4);
    const f32x4 y = *reinterpret_cast<const f32x4*>(a.Y + ((size_t)b * a.hx1 + i) * a.hj2 + 4 * k4);
    const f32x4 e = *reinterpret_cast<const f32x4*>(a.ET + (size_t)i * a.hj2 + 4 * k4);
    acc += (y[0] * e[0] + y[1] * e[1]) + (y[2] * e[2] + y[3] * e[3]);
  }
  const double s = wave_sum((double)acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) a.trp[(size_t)(a.row0 + b) * JT_SLICES + sl] = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
}
__global__ void transpose_kernel(const float* src, int R, int C, float* dst) {      // dst[c][r] = src[r][c]
  __shared__ float t[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < R && c0 + tx < C) t[j][tx] = src[(size_t)(r0 + j) * C + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < C && r0 + tx < R) dst[(size_t)(c0 + j) * R + r0 + tx] = t[tx][j];
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------------------------------
struct Ctx {
  int R;                                   // row capacity of one pass (multiple of 16)
  int d, dp, F2p, cat;
  // topology (NetDev::nT / nX / nJ hidden layers per branch): ids of the layers in NetDev::L and their widths
  int nT, nX, nJ, nl, lt[MLP_MAX_DEPTH], lx[MLP_MAX_DEPTH], lj[MLP_MAX_DEPTH], l_gate, l_out;
  int ht[MLP_MAX_DEPTH], hx[MLP_MAX_DEPTH], hj[MLP_MAX_DEPTH];
  float* pool = nullptr; size_t pool_floats = 0;
  // forward activations.  The LAST hidden layer of the t branch and of the x branch write their half of catv = [sx | st]
  // (exe_flow_matching.py:83); ta[i] / xa[i] hold the layers in front of it (i < nT - 1, i < nX - 1), ja[i] every joint layer
  float *ffat, *ta[MLP_MAX_DEPTH], *catv, *cond, *xa[MLP_MAX_DEPTH], *ja[MLP_MAX_DEPTH], *gate, *out, *tgt, *gc, *kv;
  // tangent twins
  float *zp, *tz1, *kz, *hz, *xaT[MLP_MAX_DEPTH], *catT, *jaT[MLP_MAX_DEPTH], *outT;
  // backward (gradients with respect to the pre-activations, same shapes as the activations)
  float *dv, *dg, *dja[MLP_MAX_DEPTH], *dcat, *dxa[MLP_MAX_DEPTH], *dta[MLP_MAX_DEPTH];
  // pre-activations of the hidden layers (allocated for gelu / swish only)
  float *pta[MLP_MAX_DEPTH] = {}, *pcat = nullptr, *pxa[MLP_MAX_DEPTH] = {}, *pja[MLP_MAX_DEPTH] = {};
  // ODE
  float *Y, *K, *rsf; int* rsi; RowState rs;
  float *vol0, *lqref; int* natt_tot;
  double* loss_part; int n_loss_part;
  WgJob* jobs = nullptr; int n_jobs = 0;
  int* n_active = nullptr; int* h_active = nullptr;
  // row compaction of the host-driven solver (solve): map, inverse map, the compact copy of z W_x1, the probe constants in use
  int *cmap = nullptr, *cpos = nullptr; float* tz1c = nullptr; const float* tz1_use = nullptr; const int* cmap_use = nullptr;
  // time-branch batch of an attempt (five stage times): Fourier rows, the t layers, [sx | st], gate, each 5 R rows
  float *ffat5 = nullptr, *ta5[MLP_MAX_DEPTH] = {}, *cat5 = nullptr, *gate5 = nullptr;
  // exact-trace log-det (no --hutch): set by mfm_create; buffers allocated by the first solve that needs them
  bool exact = false; const float* master = nullptr;
  float *jtA = nullptr, *jtB = nullptr, *jtE = nullptr, *jtET = nullptr, *jtWo = nullptr, *jtP = nullptr; int jt_chains = 0;
};

static int create(const NetDev& n, int rows_cap, Ctx** out) {
  Ctx* w = new Ctx();
  *out = w;            // handed over at once: whatever a failure below leaves allocated is freed by the caller's destroy()
  w->R = (rows_cap + 15) & ~15;
  w->d = n.d; w->dp = n.dp; w->F2p = n.F2p;
  w->nT = n.nT; w->nX = n.nX; w->nJ = n.nJ;
  {
    int l = 0;
    for (int i = 0; i < n.nT; ++i) { w->lt[i] = l; w->ht[i] = n.L[l].Np; ++l; }
    for (int i = 0; i < n.nX; ++i) { w->lx[i] = l; w->hx[i] = n.L[l].Np; ++l; }
    w->l_gate = l++;
    for (int i = 0; i < n.nJ; ++i) { w->lj[i] = l; w->hj[i] = n.L[l].Np; ++l; }
    w->l_out = l++;
    w->nl = l;
  }
  w->cat = n.hx2 + n.ht2;
  const size_t R = w->R;
  size_t o = 0;
  auto take = [&](size_t cnt) { size_t r = o; o += (cnt + 63) & ~(size_t)63; return r; };
  const bool need_pre = n.act >= MFM_ACT_GELU;
  struct Slot { float** p; size_t off; };
  std::vector<Slot> slots;
  auto want = [&](float*& p, size_t width) { slots.push_back(Slot{&p, take(R * width)}); };
  want(w->ffat, n.F2p); want(w->catv, w->cat); want(w->cond, n.dp); want(w->gate, n.dp); want(w->out, n.dp); want(w->tgt, n.dp);
  want(w->gc, n.dp); want(w->kv, n.dp); want(w->zp, n.dp); want(w->tz1, w->hx[0]); want(w->kz, n.dp); want(w->hz, n.dp);
  want(w->catT, w->cat); want(w->outT, n.dp); want(w->dv, n.dp); want(w->dg, n.dp); want(w->dcat, w->cat);
  if (need_pre) want(w->pcat, w->cat);
  for (int i = 0; i + 1 < n.nT; ++i) { want(w->ta[i], w->ht[i]); want(w->dta[i], w->ht[i]); if (need_pre) want(w->pta[i], w->ht[i]); }
  for (int i = 0; i + 1 < n.nX; ++i) { want(w->xa[i], w->hx[i]); want(w->xaT[i], w->hx[i]); want(w->dxa[i], w->hx[i]); if (need_pre) want(w->pxa[i], w->hx[i]); }
  for (int i = 0; i < n.nJ; ++i) { want(w->ja[i], w->hj[i]); want(w->jaT[i], w->hj[i]); want(w->dja[i], w->hj[i]); if (need_pre) want(w->pja[i], w->hj[i]); }
  want(w->Y, n.dp); want(w->K, 7 * (size_t)n.dp); want(w->rsf, 12); want(w->vol0, 1); want(w->lqref, 1);
  w->pool_floats = o;
  if (hipMalloc((void**)&w->pool, o * sizeof(float)) != hipSuccess) return -4;
  (void)hipMemset(w->pool, 0, o * sizeof(float));
  for (const Slot& sl : slots) *sl.p = w->pool + sl.off;
  w->rs.t = w->rsf; w->rs.dt = w->rsf + R; w->rs.h0 = w->rsf + 2 * R; w->rs.d1 = w->rsf + 3 * R; w->rs.ell = w->rsf + 4 * R; w->rs.kl = w->rsf + 5 * R;
  if (hipMalloc((void**)&w->rsi, 3 * R * sizeof(int)) != hipSuccess) return -4;
  w->rs.natt = w->rsi; w->rs.done = w->rsi + R; w->natt_tot = w->rsi + 2 * R;
  w->n_loss_part = (int)(R / 4 + 1);
  if (hipMalloc((void**)&w->loss_part, w->n_loss_part * sizeof(double)) != hipSuccess) return -4;
  if (hipMalloc((void**)&w->n_active, 16) != hipSuccess) return -4;
  if (hipHostMalloc((void**)&w->h_active, 16, hipHostMallocDefault) != hipSuccess) return -4;
  if (hipMalloc((void**)&w->cmap, 2 * R * sizeof(int)) != hipSuccess) return -4;
  w->cpos = w->cmap + R;
  (void)hipMemset(w->cmap, 0, 2 * R * sizeof(int));
  if (hipMalloc((void**)&w->tz1c, R * w->hx[0] * sizeof(float)) != hipSuccess) return -4;
  w->tz1_use = nullptr;
  {
    size_t wt = 0;
    for (int i = 0; i + 1 < n.nT; ++i) wt += w->ht[i];
    const size_t n5 = 5 * R * ((size_t)n.F2p + wt + w->cat + n.dp);
    if (hipMalloc((void**)&w->ffat5, n5 * sizeof(float)) != hipSuccess) return -4;
    (void)hipMemset(w->ffat5, 0, n5 * sizeof(float));
    float* p = w->ffat5 + 5 * R * n.F2p;
    for (int i = 0; i + 1 < n.nT; ++i) { w->ta5[i] = p; p += 5 * R * w->ht[i]; }
    w->cat5 = p; w->gate5 = w->cat5 + 5 * R * w->cat;
  }
  std::vector<WgJob> jobs;
  for (int l = 0; l < w->nl; ++l)
    for (int nt = 0; nt * 64 < n.L[l].Np; ++nt)
      for (int kt = 0; kt * 64 < n.L[l].Kp; ++kt) jobs.push_back(WgJob{l, kt, nt});
  w->n_jobs = (int)jobs.size();
  if (hipMalloc((void**)&w->jobs, jobs.size() * sizeof(WgJob)) != hipSuccess) return -4;
  (void)hipMemcpy(w->jobs, jobs.data(), jobs.size() * sizeof(WgJob), hipMemcpyHostToDevice);
  return 0;
}

static void destroy(Ctx* w) {
  if (!w) return;
  if (w->pool) (void)hipFree(w->pool);
  if (w->rsi) (void)hipFree(w->rsi);
  if (w->loss_part) (void)hipFree(w->loss_part);
  if (w->n_active) (void)hipFree(w->n_active);
  if (w->h_active) (void)hipHostFree(w->h_active);
  if (w->jobs) (void)hipFree(w->jobs);
  if (w->cmap) (void)hipFree(w->cmap);
  if (w->tz1c) (void)hipFree(w->tz1c);
  if (w->ffat5) (void)hipFree(w->ffat5);
  for (float* p : {w->jtA, w->jtB, w->jtE, w->jtET, w->jtWo, w->jtP}) if (p) (void)hipFree(p);
  delete w;
}

static Gemm fwd(const NetDev& n, int layer, const float* X, int ldx, float* Y, int ldy, int ycol, int rows, int activate) {
  Gemm g; memset(&g, 0, sizeof g);
  const LayerDesc& L = n.L[layer];
  g.W = n.Wp + L.w_off; g.KB = L.Kp / 16; g.NT = L.Np / 16; g.bias = n.bias + L.b_off;
  g.X = X; g.ldx = ldx; g.Y = Y; g.ldy = ldy; g.ycol = ycol; g.rows = rows; g.act = activate ? n.act + 1 : 0;
  return g;
}
static Gemm bwd(const NetDev& n, int layer, const float* dZ, int ldz, float* dA, int lda, int acol, int rows) {
  Gemm g; memset(&g, 0, sizeof g);
  const LayerDesc& L = n.L[layer];
  g.W = n.WpT + L.w_off; g.KB = L.Np / 16; g.NT = L.Kp / 16;          // dA = dZ W^T: the transposed packing is the "weight" here
  g.X = dZ; g.ldx = ldz; g.Y = dA; g.ldy = lda; g.ycol = acol; g.rows = rows;
  return g;
}
static Gemm kinv(const NetDev& n, const float* X, float* Y, int rows, bool with_bias) {
  Gemm g; memset(&g, 0, sizeof g);
  g.W = n.T.KinvP; g.KB = n.dp / 16; g.NT = n.dp / 16; g.bias = with_bias ? n.T.kbias : nullptr;
  g.X = X; g.ldx = n.dp; g.Y = Y; g.ldy = n.dp; g.rows = rows;
  return g;
}
static int grid4(int rows) { return (rows + 3) / 4; }
static int grid_el(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 4096 ? b : 4096); }

static void target_eval(Ctx* w, const NetDev& n, const float* X, const float* Z, int rows, hipStream_t s, bool diag = false) {
  if (n.T.kind == MFM_TARGET_LGCP) launch_gemm(kinv(n, X, w->kv, rows, true), s);
  TgtArgs t; memset(&t, 0, sizeof t);
  t.T = n.T; t.clip = n.grad_clip; t.rows = rows; t.d = n.d; t.dp = n.dp; t.X = X; t.Z = Z; t.KV = w->kv; t.KZ = w->kz; t.GC = w->gc; t.HZ = (Z || diag) ? w->hz : nullptr;
  t.diag = diag ? 1 : 0;
  t.cmap = w->cmap_use;
  if (n.T.kind == MFM_TARGET_GMM) { hipLaunchKernelGGL(target_gmm_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, t); return; }
  hipLaunchKernelGGL(target_kernel, dim3(grid_el((size_t)rows * n.dp)), dim3(256), 0, s, t);
}

// time branch (exe_flow_matching.py:73-75,81): Fourier rows -> the t layers (the last one into the st half of [sx | st]) -> gate.
// `ta`: the buffers of the layers in front of the last; `pre`: keep the pre-activations (gelu / swish backward pass)
static void time_branch_on(Ctx* w, const NetDev& n, const float* ffat, float* const* ta, float* cat, float* gate, bool pre, int rows, hipStream_t s) {
  for (int i = 0; i < w->nT; ++i) {
    const bool last = i + 1 == w->nT;
    Gemm g = fwd(n, w->lt[i], i == 0 ? ffat : ta[i - 1], i == 0 ? n.F2p : w->ht[i - 1], last ? cat : ta[i], last ? w->cat : w->ht[i], last ? n.hx2 : 0, rows, 1);
    if (pre) g.P = last ? w->pcat : w->pta[i];
    launch_gemm(g, s);
  }
  launch_gemm(fwd(n, w->l_gate, cat + n.hx2, w->cat, gate, n.dp, 0, rows, 0), s);
}
static void time_branch(Ctx* w, const NetDev& n, int rows, hipStream_t s) { time_branch_on(w, n, w->ffat, w->ta, w->catv, w->gate, true, rows, s); }
// x branch + joint layers (:77-79,83-86) on value rows X (and tangent rows: z in w->zp, z W_x1 in w->tz1)
static void x_branch(Ctx* w, const NetDev& n, const float* X, bool tangent, int rows, hipStream_t s) {
  for (int i = 0; i < w->nX; ++i) {
    const bool last = i + 1 == w->nX;
    Gemm g = fwd(n, w->lx[i], i == 0 ? X : w->xa[i - 1], i == 0 ? n.dp : w->hx[i - 1], last ? w->catv : w->xa[i], last ? w->cat : w->hx[i], 0, rows, 1);
    g.P = last ? w->pcat : w->pxa[i];
    if (tangent) {
      g.YT = last ? w->catT : w->xaT[i];
      if (i == 0) { g.TS = w->tz1_use ? w->tz1_use : w->tz1; g.ldts = w->hx[0]; }
      else { g.XT = w->xaT[i - 1]; g.KBT = g.KB; }
    }
    launch_gemm(g, s);
  }
  for (int i = 0; i < w->nJ; ++i) {
    Gemm g = fwd(n, w->lj[i], i == 0 ? w->catv : w->ja[i - 1], i == 0 ? w->cat : w->hj[i - 1], w->ja[i], w->hj[i], 0, rows, 1);
    g.P = w->pja[i];
    if (tangent) { g.XT = i == 0 ? w->catT : w->jaT[i - 1]; g.KBT = i == 0 ? n.hx2 / 16 : g.KB; g.YT = w->jaT[i]; }      // the st half of the tangent is zero
    launch_gemm(g, s);
  }
  const int jl = w->nJ - 1;
  Gemm g = fwd(n, w->l_out, w->ja[jl], w->hj[jl], w->out, n.dp, 0, rows, 0);
  if (tangent) { g.XT = w->jaT[jl]; g.KBT = g.KB; g.YT = w->outT; }
  launch_gemm(g, s);
}
// once per solve: z W_x1 (no bias) and, LGCP, K^-1 z
static void probe_setup(Ctx* w, const NetDev& n, int rows, hipStream_t s) {
  Gemm g = fwd(n, w->lx[0], w->zp, n.dp, w->tz1, w->hx[0], 0, rows, 0);
  g.bias = nullptr;
  launch_gemm(g, s);
  if (n.T.kind == MFM_TARGET_LGCP) launch_gemm(kinv(n, w->zp, w->kz, rows, false), s);
}

struct FmCall {
  Key2 key_time, key_ref, key_gauss; uint32_t n_total, chain_offset; float sigma; int cond_flow; double ref_std;
  const float* pos; int rows;
  int rows_valid;        // rows >= rows_valid: padding of the chain shard (no loss, no gradient)
  int* bad;              // non-null (training, one rank): finite check of the gradient rides in the weight-gradient kernel
};
// loss (+ gradient into d_grads, canonical layout) on `rows` samples; the per-workgroup loss partials land in w->loss_part
static int fm(Ctx* w, const NetDev& n, const FmCall& c, bool train, float* d_grads, hipStream_t s) {
  const int rows = c.rows;
  if (rows > w->R || rows % 16) return -3;
  FmPro p; memset(&p, 0, sizeof p);
  p.key_time = c.key_time; p.key_ref = c.key_ref; p.key_gauss = c.key_gauss; p.n_total = c.n_total; p.chain_offset = c.chain_offset;
  p.rows = rows; p.d = n.d; p.dp = n.dp; p.F = n.F; p.F2p = n.F2p; p.sigma = c.sigma; p.cond_flow = c.cond_flow; p.ref_std = c.ref_std;
  p.pos = c.pos; p.fourier = n.fourier; p.cond = w->cond; p.tgt = w->tgt; p.ffat = w->ffat; p.flag_reset = train ? c.bad : nullptr;
  hipLaunchKernelGGL(fm_prologue_kernel, dim3(grid4(rows)), dim3(256), 0, s, p);
  time_branch(w, n, rows, s);
  target_eval(w, n, w->cond, nullptr, rows, s);
  x_branch(w, n, w->cond, false, rows, s);
  LossArgs l; memset(&l, 0, sizeof l);
  l.rows = rows; l.rows_valid = c.rows_valid; l.d = n.d; l.dp = n.dp; l.out = w->out; l.gate = w->gate; l.gc = w->gc; l.tgt = w->tgt; l.part = w->loss_part;
  if (train) { l.dv = w->dv; l.dg = w->dg; }
  hipLaunchKernelGGL(loss_kernel, dim3(grid4(rows)), dim3(256), 0, s, l);
  if (!train) return 0;
  // ---- backward: data gradients (the reverse sweep jax.value_and_grad performs at :364-365) ----
  // the derivative of the activation comes from the stored output (relu / tanh / elu) or, for gelu / swish, from the
  // stored pre-activation
  const bool pre = n.act >= MFM_ACT_GELU;
  auto masked = [&](Gemm g, const float* out, const float* prebuf, int ld, int col) {
    g.mask = pre ? prebuf : out; g.ldm = ld; g.mcol = col; g.mask_kind = n.act; g.mask_is_pre = pre ? 1 : 0;
    return g;
  };
  const int jl = w->nJ - 1, xl = w->nX - 1, tl = w->nT - 1;
  launch_gemm(masked(bwd(n, w->l_out, w->dv, n.dp, w->dja[jl], w->hj[jl], 0, rows), w->ja[jl], w->pja[jl], w->hj[jl], 0), s);
  for (int i = jl; i >= 1; --i)
    launch_gemm(masked(bwd(n, w->lj[i], w->dja[i], w->hj[i], w->dja[i - 1], w->hj[i - 1], 0, rows), w->ja[i - 1], w->pja[i - 1], w->hj[i - 1], 0), s);
  // d [sx | st] through the first joint layer (not yet through the activations of sx / st)
  launch_gemm(bwd(n, w->lj[0], w->dja[0], w->hj[0], w->dcat, w->cat, 0, rows), s);
  // st half: += dgate W_gate^T, then through the activation of st (GEMM epilogue); sx half: through the activation of sx
  // by one thin elementwise pass (it is both the input of the data-gradient GEMM of the x branch's last layer and the dZ of that
  // layer's weight gradient)
  {
    Gemm g = bwd(n, w->l_gate, w->dg, n.dp, w->dcat, w->cat, n.hx2, rows);
    g.add = w->dcat; g.lda = w->cat; g.acol = n.hx2;
    launch_gemm(masked(g, w->catv, w->pcat, w->cat, n.hx2), s);
  }
  {
    ElemMask e; e.rows = rows; e.cols = n.hx2; e.ld = w->cat; e.x = w->dcat; e.m = pre ? w->pcat : w->catv; e.kind = n.act; e.is_pre = pre ? 1 : 0;
    hipLaunchKernelGGL(elem_mask_kernel, dim3(grid_el((size_t)rows * n.hx2 / 4)), dim3(256), 0, s, e);
  }
  for (int i = xl; i >= 1; --i)          // x branch, towards its first layer (the gradient with respect to x itself is not needed)
    launch_gemm(masked(bwd(n, w->lx[i], i == xl ? w->dcat : w->dxa[i], i == xl ? w->cat : w->hx[i], w->dxa[i - 1], w->hx[i - 1], 0, rows),
                       w->xa[i - 1], w->pxa[i - 1], w->hx[i - 1], 0), s);
  for (int i = tl; i >= 1; --i)          // t branch
    launch_gemm(masked(bwd(n, w->lt[i], i == tl ? w->dcat + n.hx2 : w->dta[i], i == tl ? w->cat : w->ht[i], w->dta[i - 1], w->ht[i - 1], 0, rows),
                       w->ta[i - 1], w->pta[i - 1], w->ht[i - 1], 0), s);
  // ---- weight gradients, straight into the canonical flat gradient vector ----
  WgArgs a; memset(&a, 0, sizeof a);
  auto wg = [&](int l, const float* A, int lda, const float* Z, int ldz) {
    const LayerDesc& L = n.L[l];
    const bool j0 = l == w->lj[0];
    const LayerDesc& Lx = n.L[w->lx[w->nX - 1]];
    a.L[l] = WgLayer{A, lda, Z, ldz, L.K, L.N, L.Kp, L.Np, L.m_w, L.m_b, j0 ? Lx.N : L.Kp, j0 ? Lx.Np : L.Kp};
  };
  for (int i = 0; i <= tl; ++i) wg(w->lt[i], i == 0 ? w->ffat : w->ta[i - 1], i == 0 ? n.F2p : w->ht[i - 1], i == tl ? w->dcat + n.hx2 : w->dta[i], i == tl ? w->cat : w->ht[i]);
  for (int i = 0; i <= xl; ++i) wg(w->lx[i], i == 0 ? w->cond : w->xa[i - 1], i == 0 ? n.dp : w->hx[i - 1], i == xl ? w->dcat : w->dxa[i], i == xl ? w->cat : w->hx[i]);
  wg(w->l_gate, w->catv + n.hx2, w->cat, w->dg, n.dp);
  for (int i = 0; i <= jl; ++i) wg(w->lj[i], i == 0 ? w->catv : w->ja[i - 1], i == 0 ? w->cat : w->hj[i - 1], w->dja[i], w->hj[i]);
  wg(w->l_out, w->ja[jl], w->hj[jl], w->dv, n.dp);
  a.jobs = w->jobs; a.n_jobs = w->n_jobs; a.rows = rows; a.grads = d_grads; a.bad = c.bad;
  // what does not fill the chip's 2048 wave slots a whole number of times, if it is a tail of at most one workgroup per CU
  const int rem = w->n_jobs % 2048;
  const bool tail = w->n_jobs > 2048 && rem > 0 && rem <= 256 && rows % 64 == 0 && !g_sw.wide_wgrad_nosplit;
  a.n_full = tail ? w->n_jobs - rem : (w->n_jobs + 3) / 4 * 4;
  hipLaunchKernelGGL(wgrad_kernel, dim3(a.n_full / 4 + (tail ? rem : 0)), dim3(256), 0, s, a);
  return 0;
}

// ---- exact trace of d nn_xt / d x for the `rows` (compact) rows of the evaluation x_branch has just run: see jt_seed_kernel ----
constexpr int JT_ROWS_CAP = 131072;      // tangent rows per pass (128 chains x hx1 = 1024): two buffers of rows x (widest layer behind x1) floats
static int jt_alloc(Ctx* w, const NetDev& n) {
  if (w->jtA) return 0;
  int chains = JT_ROWS_CAP / n.hx1;
  if (chains < 1) chains = 1;
  if (chains > w->R) chains = w->R;
  const size_t rows = (size_t)chains * n.hx1;
  size_t hmax = 0;
  for (int i = 1; i < w->nX; ++i) if ((size_t)w->hx[i] > hmax) hmax = w->hx[i];
  for (int i = 0; i < w->nJ; ++i) if ((size_t)w->hj[i] > hmax) hmax = w->hj[i];
  if (hipMalloc((void**)&w->jtA, rows * hmax * sizeof(float)) != hipSuccess) return -4;
  if (hipMalloc((void**)&w->jtB, rows * hmax * sizeof(float)) != hipSuccess) return -4;
  if (hipMalloc((void**)&w->jtE, (size_t)n.hj2 * n.hx1 * sizeof(float)) != hipSuccess) return -4;
  if (hipMalloc((void**)&w->jtET, (size_t)n.hj2 * n.hx1 * sizeof(float)) != hipSuccess) return -4;
  if (hipMalloc((void**)&w->jtWo, (size_t)n.hj2 * n.dp * sizeof(float)) != hipSuccess) return -4;
  (void)hipMemset(w->jtWo, 0, (size_t)n.hj2 * n.dp * sizeof(float));
  if (hipMalloc((void**)&w->jtP, (size_t)w->R * JT_SLICES * sizeof(float)) != hipSuccess) return -4;
  w->jt_chains = chains;
  return 0;
}
// E^T, E = W_out W_x1 (the parameters do not change inside a solve): rows of the canonical out kernel [hj2][d] as the "activations"
// of the x1 layer's packed weights, then a transpose so that jt_trace_kernel reads row i of E^T beside tangent row i
static void jt_setup(Ctx* w, const NetDev& n, hipStream_t s) {
  const int hj_true = n.L[w->l_out].K;          // rows of the canonical out kernel (jtWo's rows beyond them stay zero: jt_alloc)
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_el((size_t)hj_true * n.dp)), dim3(256), 0, s, w->master + n.L[w->l_out].m_w, hj_true, n.d, n.dp, w->jtWo);
  Gemm g = fwd(n, w->lx[0], w->jtWo, n.dp, w->jtE, n.hx1, 0, n.hj2, 0);
  g.bias = nullptr;
  launch_gemm(g, s);
  hipLaunchKernelGGL(transpose_kernel, dim3((n.hx1 + 31) / 32, (n.hj2 + 31) / 32), dim3(256), 0, s, w->jtE, n.hj2, n.hx1, w->jtET);
}
// The layers a tangent of x1's output passes on its way to the output layer: the rest of the x branch, then the joint branch (whose
// first layer sees the tangent through its sx rows only).  The FIRST of them is applied element-wise by jt_seed_kernel (tangent row
// i of a chain starts as act'(x1)_i times row i of that layer's kernel), the others are GEMMs with the chain's mask in the epilogue.
static void exact_trace(Ctx* w, const NetDev& n, int rows, hipStream_t s) {
  const bool pre = n.act >= MFM_ACT_GELU;       // act' from the stored pre-activations (gelu, swish) or from the stored outputs
  struct Hop { int layer; const float* m; int ldm; int width; bool joint0; };
  Hop hops[2 * MLP_MAX_DEPTH]; int nh = 0;
  for (int i = 1; i < w->nX; ++i) {
    const bool last = i + 1 == w->nX;
    hops[nh++] = Hop{w->lx[i], last ? (pre ? w->pcat : w->catv) : (pre ? w->pxa[i] : w->xa[i]), last ? w->cat : w->hx[i], w->hx[i], false};
  }
  for (int i = 0; i < w->nJ; ++i) hops[nh++] = Hop{w->lj[i], pre ? w->pja[i] : w->ja[i], w->hj[i], w->hj[i], i == 0};
  // act'(x1): the first x layer's output sits in [sx | st] when it is the branch's only layer
  const float* m1 = w->nX == 1 ? (pre ? w->pcat : w->catv) : (pre ? w->pxa[0] : w->xa[0]);
  const int ld1 = w->nX == 1 ? w->cat : w->hx[0];
  for (int r0 = 0; r0 < rows; r0 += w->jt_chains) {
    const int C = rows - r0 < w->jt_chains ? rows - r0 : w->jt_chains;
    const int trows = C * n.hx1;
    float *cur = w->jtA, *nxt = w->jtB;
    JtSeed sd; memset(&sd, 0, sizeof sd);
    sd.chains = C; sd.hx1 = n.hx1; sd.hx2 = hops[0].width; sd.row0 = r0; sd.W = w->master + n.L[hops[0].layer].m_w;      // rows [0, hx1) of the canonical kernel
    sd.rows_w = n.L[w->lx[0]].N; sd.cols_w = n.L[hops[0].layer].N;
    sd.m1 = m1; sd.ld1 = ld1; sd.m2 = hops[0].m; sd.ld2 = hops[0].ldm; sd.kind = n.act; sd.is_pre = pre; sd.X = cur;
    hipLaunchKernelGGL(jt_seed_kernel, dim3(grid_el((size_t)trows * hops[0].width / 4)), dim3(256), 0, s, sd);
    for (int h = 1; h < nh; ++h) {
      const LayerDesc& L = n.L[hops[h].layer];
      Gemm g; memset(&g, 0, sizeof g);           // tangent rows x the layer (the sx rows of the first joint layer), masked by the chain's act'
      g.W = n.Wp + L.w_off; g.KB = L.Kp / 16; g.NT = L.Np / 16;
      if (hops[h].joint0) { g.KBW = L.Kp / 16; g.KB = n.hx2 / 16; }
      g.X = cur; g.ldx = hops[h - 1].width; g.Y = nxt; g.ldy = hops[h].width; g.rows = trows;
      g.mask = hops[h].m + (size_t)r0 * hops[h].ldm; g.ldm = hops[h].ldm; g.mask_kind = n.act; g.mask_is_pre = pre; g.mask_rdiv = n.hx1;
      launch_gemm(g, s);
      float* t = cur; cur = nxt; nxt = t;
    }
    JtTrace t; t.hx1 = n.hx1; t.hj2 = n.hj2; t.row0 = r0; t.Y = cur; t.ET = w->jtET; t.trp = w->jtP;
    hipLaunchKernelGGL(jt_trace_kernel, dim3(JT_SLICES, C), dim3(256), 0, s, t);
  }
}

// ---- one evaluation of the augmented field on the rows in w->X-like buffer `X` (times already in w->ffat) ----
// mode 0: value rows only; 1: + the tangent rows of the probe (Hutchinson / mfm_vf_apply's JVP); 2: + the exact Jacobian trace
// (`live`: the rows of the evaluation that carry a chain -- compact evaluations are padded to a multiple of 16)
static void field_eval(Ctx* w, const NetDev& n, const float* X, int mode, bool time_too, int rows, hipStream_t s, int live = -1) {
  if (time_too) time_branch(w, n, rows, s);
  target_eval(w, n, X, mode == 1 ? w->zp : nullptr, rows, s, mode == 2);
  x_branch(w, n, X, mode == 1, rows, s);
  if (mode == 2) exact_trace(w, n, live < 0 ? rows : live, s);
}

struct SolveArgs { int sign; float rtol, atol; int max_attempts; int rows; WReplay rp; };

// Integrate rows of w->Y (padded [rows][dp]) from t = 0 to 1 with the probe in w->zp; results: w->Y, w->rs.ell, w->rs.natt.
// Synchronises the stream once per attempted step (4-byte read-back of the number of rows still integrating).
static int solve(Ctx* w, const NetDev& n, const SolveArgs& c, float* xstage, hipStream_t s) {
  const int rows = c.rows;
  OdeBuf o; memset(&o, 0, sizeof o);
  o.rows = rows; o.d = n.d; o.dp = n.dp; o.F = n.F; o.F2p = n.F2p; o.sign = c.sign; o.rtol = c.rtol; o.atol = c.atol; o.max_attempts = c.max_attempts;
  const bool exact = w->exact;
  const int fmode = exact ? 2 : 1;
  if (exact) {
    if (!w->master || jt_alloc(w, n)) return -4;
    jt_setup(w, n, s);
  }
  o.rs = w->rs; o.Y = w->Y; o.K = w->K; o.X = xstage; o.Z = exact ? nullptr : w->zp; o.trp = exact ? w->jtP : nullptr; o.ffat = w->ffat; o.fourier = n.fourier;
  o.out = w->out; o.outT = w->outT; o.gate = w->gate; o.gc = w->gc; o.hz = w->hz; o.n_active = w->n_active;
  o.rp = c.rp;
  hipLaunchKernelGGL(ode_init_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, w->rs, rows);
  if (hipMemsetAsync(w->K, 0, (size_t)7 * rows * n.dp * sizeof(float), s) != hipSuccess) return -4;
  if (!exact) probe_setup(w, n, rows, s);
  const bool vec4 = (n.d & 3) == 0 && (n.dp & 3) == 0;
  auto stage_prep = [&](const OdeBuf& ob, int phase) {
    if (vec4) hipLaunchKernelGGL(stage_prep_kernel<f32x4>, dim3(grid4(rows)), dim3(256), 0, s, ob, phase);
    else hipLaunchKernelGGL(stage_prep_kernel<float>, dim3(grid4(rows)), dim3(256), 0, s, ob, phase);
  };
  auto stage_finish = [&](const OdeBuf& ob, int phase) {
    if (vec4) hipLaunchKernelGGL(stage_finish_kernel<f32x4>, dim3(grid4(rows)), dim3(256), 0, s, ob, phase);
    else hipLaunchKernelGGL(stage_finish_kernel<float>, dim3(grid4(rows)), dim3(256), 0, s, ob, phase);
  };
  auto read_active = [&](int& v) -> int {
    if (hipMemcpyAsync(w->h_active, w->n_active, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return -4;
    if (hipStreamSynchronize(s) != hipSuccess) return -4;
    v = *w->h_active;
    return 0;
  };
  for (int phase = 0; phase < 2; ++phase) {                   // f0 and the extra evaluation of the initial-step heuristic
    if (phase == 1 && hipMemsetAsync(w->n_active, 0, sizeof(int), s) != hipSuccess) return -4;
    stage_prep(o, phase);
    field_eval(w, n, xstage, fmode, true, rows, s);
    stage_finish(o, phase);
  }
  int active = 0;
  if (read_active(active)) return -4;
  // every row stops after max_attempts attempted steps, so the loop is bounded even if the read-back misbehaves
  // Row compaction: the chains of a solve end after different numbers of attempts (13.2 on average, 15.8 for the slowest, at the
  // pines shape), and the layer GEMMs are the cost.  From the attempt on in which fewer chains than rows are integrating, the
  // evaluation buffers are COMPACT: stage_prep writes chain b's stage input and Fourier row to row cpos[b], the GEMMs run on
  // ceil16(active) rows, stage_finish reads chain b's results from row cpos[b]; per-solve constants the GEMMs take per row
  // (z W_x1) are gathered once per attempt, the element-wise target kernel follows the map.  The number of rows is the
  // read-back the loop already makes.
  const bool no_compact = g_sw.wide_nocompact;
  const bool no_tbatch = g_sw.wide_no_tbatch;
  for (int it = 0; active > 0 && it < c.max_attempts; ++it) {
    if (hipMemsetAsync(w->n_active, 0, sizeof(int), s) != hipSuccess) return -4;
    int rc = rows, live = rows;
    if (!no_compact && active < rows) {
      live = active;
      hipLaunchKernelGGL(compact_map_kernel, dim3(1), dim3(1024), 0, s, w->rs, rows, c.max_attempts, w->cmap, w->cpos);
      if (!exact) hipLaunchKernelGGL(gather_rows_c_kernel, dim3(grid_el((size_t)active * n.hx1 / 4)), dim3(256), 0, s, w->tz1, w->cmap, active, n.hx1, w->tz1c);
      rc = (active + 15) & ~15;
      o.cpos = w->cpos; w->tz1_use = w->tz1c; w->cmap_use = w->cmap;
    }
    if (!no_tbatch) {
      // Time-branch batching: Fourier features -> t1 -> st -> gate depend on t only and the six stage times of an attempt
      // (five distinct) are known when it starts: ONE chain of three GEMMs on 5 rc rows (slot-major) instead of five chains
      // on rc rows -- 12 launches fewer per attempt, and GEMMs tall enough to keep every CU busy past their first tile
      OdeBuf ob = o;
      ob.tb_rows = rc; ob.ffat = w->ffat5;
      stage_prep(ob, 2);
      {
        const int r5 = 5 * rc;
        time_branch_on(w, n, w->ffat5, w->ta5, w->cat5, w->gate5, false, r5, s);
      }
      float* const cat_keep = w->catv;
      for (int phase = 2; phase < 8; ++phase) {
        const int slot = phase - 2 < 4 ? phase - 2 : 4;
        ob.gate = w->gate5 + (size_t)slot * rc * n.dp;
        w->catv = w->cat5 + (size_t)slot * rc * w->cat;        // x2 writes its sx half into the slot's [sx | st] rows, j1 reads them
        if (phase > 2) stage_prep(ob, phase);
        field_eval(w, n, xstage, fmode, false, rc, s, live);
        stage_finish(ob, phase);
      }
      w->catv = cat_keep;
    } else
    for (int phase = 2; phase < 8; ++phase) {
      stage_prep(o, phase);
      field_eval(w, n, xstage, fmode, phase != 7, rc, s, live);
      stage_finish(o, phase);
    }
    if (read_active(active)) { w->tz1_use = nullptr; w->cmap_use = nullptr; return -4; }
  }
  w->tz1_use = nullptr; w->cmap_use = nullptr;
  return hipGetLastError() == hipSuccess ? 0 : -4;
}


static void pad_rows(const float* src, int n, int d, int dp, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_el((size_t)n * dp)), dim3(256), 0, s, src, n, d, dp, dst);
}
static void unpad_rows(const float* src, int n, int d, int dp, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(unpad_rows_kernel, dim3(grid_el((size_t)n * d)), dim3(256), 0, s, src, n, d, dp, dst);
}

// transform_and_logdet / inverse_and_logdet (:206-242) on n samples, R rows per pass.  z: Hutchinson probes [n][d].
static int transform(Ctx* w, const NetDev& n, int direction, float rtol, float atol, int max_attempts, const float* z, const float* in,
                     int cnt, float* out, float* ldj, int* nsteps, hipStream_t s, WReplay rp = WReplay{}) {
  for (int r0 = 0; r0 < cnt; r0 += w->R) {
    const int rows = cnt - r0 < w->R ? cnt - r0 : w->R;
    pad_rows(in + (size_t)r0 * n.d, rows, n.d, n.dp, w->Y, s);
    pad_rows(z + (size_t)r0 * n.d, rows, n.d, n.dp, w->zp, s);
    rp.solve = 0; rp.row0 = r0;
    SolveArgs c{direction, rtol, atol, max_attempts, rows, rp};
    const int rc = solve(w, n, c, w->cond, s);
    if (rc) return rc;
    unpad_rows(w->Y, rows, n.d, n.dp, out + (size_t)r0 * n.d, s);
    if (hipMemcpyAsync(ldj + r0, w->rs.ell, rows * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return -4;
    if (nsteps && hipMemcpyAsync(nsteps + r0, w->rs.natt, rows * sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess) return -4;
  }
  return 0;
}

struct FlowCall {
  int mode; Key2 key; uint32_t n_total, chain_offset; double beta; int rows; float ref_std;
  float rtol, atol; int max_attempts;
  const float *z_inv, *z_fwd, *zgen;          // [rows][d]: key_hutch2, key_hutch1, key_gen draws (:265 / :247)
  float* pos; double* logp; float* grad; float* acc_prob; uint8_t* accepted; float* proposed; int* nsteps;
  WReplay rp;
};
static int flow_step(Ctx* w, const NetDev& n, const FlowCall& c, hipStream_t s) {
  const int rows = c.rows;
  if (rows > w->R || rows % 16) return -3;
  FlowGlue f; memset(&f, 0, sizeof f);
  f.mode = c.mode; f.key = c.key; f.n_total = c.n_total; f.chain_offset = c.chain_offset; f.beta = c.beta; f.ref_std = c.ref_std;
  f.rows = rows; f.d = n.d; f.dp = n.dp; f.T = n.T; f.Y = w->Y; f.zgen = c.zgen; f.ell = w->rs.ell; f.vol0 = w->vol0; f.lqref = w->lqref;
  f.natt_tot = w->natt_tot; f.natt = w->rs.natt; f.KV = w->kv;
  f.pos = c.pos; f.logp = c.logp; f.grad = c.grad; f.acc_prob = c.acc_prob; f.accepted = c.accepted; f.proposed = c.proposed; f.nsteps = c.nsteps;
  // inverse solve from the current position (:267 / :251)
  pad_rows(c.pos, rows, n.d, n.dp, w->Y, s);
  pad_rows(c.z_inv, rows, n.d, n.dp, w->zp, s);
  SolveArgs sa{-1, c.rtol, c.atol, c.max_attempts, rows, c.rp};
  sa.rp.solve = 0; sa.rp.row0 = 0;
  f.diag = c.rp.diag;
  int rc = solve(w, n, sa, w->cond, s);
  if (rc) return rc;
  hipLaunchKernelGGL(flow_propose_kernel, dim3(grid4(rows)), dim3(256), 0, s, f);
  // forward solve of the proposal (:269 / :250)
  pad_rows(c.z_fwd, rows, n.d, n.dp, w->zp, s);
  sa.sign = 1; sa.rp.solve = 1;
  rc = solve(w, n, sa, w->cond, s);
  if (rc) return rc;
  if (n.T.kind == MFM_TARGET_LGCP) launch_gemm(kinv(n, w->Y, w->kv, rows, true), s);
  hipLaunchKernelGGL(flow_accept_kernel, dim3(grid4(rows)), dim3(256), 0, s, f);
  return 0;
}

// v(x, t) and J z for cnt samples (mfm_vf_apply)
__global__ void fourier_rows_kernel(const float* fr, int F, int F2p, const float* t, int rows, float* ffat) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b < rows) fourier_row(fr, F, F2p, (double)t[b], ffat + (size_t)b * F2p, lane);
}
static int vf_apply(Ctx* w, const NetDev& n, const float* x, const float* t, const float* tan, int cnt, float* v, float* jvp, hipStream_t s) {
  for (int r0 = 0; r0 < cnt; r0 += w->R) {
    const int rows = cnt - r0 < w->R ? cnt - r0 : w->R;
    pad_rows(x + (size_t)r0 * n.d, rows, n.d, n.dp, w->cond, s);
    if (tan) { pad_rows(tan + (size_t)r0 * n.d, rows, n.d, n.dp, w->zp, s); probe_setup(w, n, rows, s); }
    hipLaunchKernelGGL(fourier_rows_kernel, dim3(grid4(rows)), dim3(256), 0, s, n.fourier, n.F, n.F2p, t + r0, rows, w->ffat);
    field_eval(w, n, w->cond, tan != nullptr ? 1 : 0, true, rows, s);
    hipLaunchKernelGGL(vf_out_kernel, dim3(grid_el((size_t)rows * n.d)), dim3(256), 0, s, rows, n.d, n.dp, w->out, w->outT, w->gate, w->gc, w->hz,
                       v + (size_t)r0 * n.d, jvp ? jvp + (size_t)r0 * n.d : nullptr);
  }
  return 0;
}


static int mala_lgcp(Ctx* w, const NetDev& n, LgcpMala a, hipStream_t s) {
  if (a.rows > w->R) return -3;
  a.Y = w->Y; a.KV = w->kv; a.th1 = reinterpret_cast<double*>(w->K);          // K (stage buffer) is free outside a solve
  hipLaunchKernelGGL(lgcp_propose_kernel, dim3(grid4(a.rows)), dim3(256), 0, s, a);
  launch_gemm(kinv(n, w->Y, w->kv, (a.rows + 15) & ~15, true), s);
  hipLaunchKernelGGL(lgcp_accept_kernel, dim3(grid4(a.rows)), dim3(256), 0, s, a);
  return 0;
}

}  // namespace wide
