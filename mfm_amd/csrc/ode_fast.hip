// K5f / K6f: the flow-MH step and the CNF transforms for the HEADLINE network shape (fourier_dim = 128, all hidden
// widths 128, dim 128 or 256, Hutchinson log-det, PhiFour target), as a shape-specialised sibling of the generic solver
// tile in ode.hip.  Same algorithm, same arithmetic per field evaluation (exe_flow_matching.py:206-242, :246-278;
// jax.experimental.ode.odeint restated in oracle/ode.py); what changes is the schedule:
//
//  * TIME-BRANCH BATCHING.  The time branch of the vector field (Fourier features -> t1 -> t2 -> gate, and the
//    st-half of the first joint layer) depends on t only, and all six stage times of a Dormand-Prince attempt are
//    known when the attempt starts.  It is therefore evaluated ONCE per attempt for the five distinct stage times as
//    an M = 80 GEMM chain (5 x 16 rows; each streamed weight fragment feeds five MFMA row tiles instead of one).
//    Its outputs (gate and the st contribution to j1's pre-activation, per stage) are kept in accumulator layout in a
//    per-workgroup global scratch; every lane later reads back exactly the elements it wrote.
//  * A field evaluation is then the x branch only: x1 (M = 16), x2, j1 (K = 128 instead of 256, accumulators
//    initialised with the st contribution), j2, out (two column tiles per wave sharing their A fragments), M = 32
//    (value + tangent rows), five workgroup barriers.
//  * Weight fragments are prefetched ACROSS layers and barriers: two ping-pong register sets of four fragments; the
//    first group of the next layer is issued before the last group of the current one executes, so no layer starts
//    by waiting for L2.
//  * The per-stage divergence partials are left per wave in LDS and only summed when the attempt is judged; stage
//    inputs are double buffered, so the out-layer epilogue (which reads x for grad log pi) never races the next write.
//  * All shapes are compile-time constants: no layer-descriptor loads, no dynamic loops.
//
// Everything that is not the solver core (probes, proposal, target evaluation at the proposal, accept / reject) follows
// flow_step_kernel in ode.hip.  Other shapes / targets / the exact-trace mode keep using the generic kernels.

namespace fast {

constexpr int NW = 8, H = 128, F = 128;
constexpr int SCR_F4_PER_WG = ODE_FAST_SCR_F4;      // float4 per workgroup: [slot 5][wave 8][gate q0, gate q1, j1t][lane 64]

template <int D>
struct FS {                                         // float offsets (LDS, packed weights, biases)
  static constexpr int TPW = D / 128;
  static constexpr int LDX = D + 8, LDH = H + 4;
  static constexpr int XB0 = 0, XB1 = 16 * LDX, ZB = 32 * LDX, R = 48 * LDX;
  static constexpr int A1 = R, SX = R + 32 * LDH, J1 = R + 64 * LDH, J2 = R + 96 * LDH;     // x branch (32 rows each)
  static constexpr int FH = R, T1 = R + 80 * LDH;                                             // time batch (80 rows each); ST = FH
  static constexpr int RS = R + 160 * LDH, RED = RS + 256, DLP = RED + 4 * 128, BIAS = DLP + 8 * 128;
  static constexpr int BTOT = 6 * H + 2 * D, TOTAL = BIAS + BTOT;
  static constexpr int W0 = 0, W1 = W0 + 2 * F * H, W2 = W1 + H * H, W3 = W2 + D * H, W4 = W3 + H * H, W5 = W4 + H * D,
                       W6 = W5 + 2 * H * H, W7 = W6 + H * H, WTOT = W7 + H * D;
  static constexpr int B0 = 0, B1 = H, B2 = 2 * H, B3 = 3 * H, B4 = 4 * H, B5 = 4 * H + D, B6 = 5 * H + D, B7 = 6 * H + D;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// four B fragments: NTL == 1: four consecutive k-blocks of one column tile; NTL == 2: two k-blocks x two column tiles
// Weights and the time-branch scratch are read through raw buffer loads: a wave-uniform descriptor (SGPRs), ONE shared
// per-lane byte offset (lane * 16) and a wave-uniform scalar offset per fragment group -- there are no per-fragment
// address registers for the compiler to hoist out of the solver loop (64-bit per-lane pointers did exactly that and
// spilled ~400 registers).  `soff` values are BYTE offsets derived from constants and the readfirstlane'd wave index.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
template <int NTL, int T1OFF>
__device__ __forceinline__ void load_group(f32x4 (&bf)[4], __amdgpu_buffer_rsrc_t r, int soff, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bf[j] = NTL == 1 ? bload(r, lane * 16 + j * 1024, soff) : bload(r, lane * 16 + (j >> 1) * 1024, soff + (j & 1) * T1OFF);
}

template <int MT, int NTL, int LDA>
__device__ __forceinline__ void exec_group(const float* arow, const f32x4 (&bf)[4], f32x4 (&acc)[NTL][MT]) {
  constexpr int KPG = 4 / NTL;
  // A fragments one k-block ahead (two register sets, statically renamed by the unroll): the LDS latency of block u + 1
  // hides behind the MFMAs of block u, and the scheduler cannot hoist a whole group's reads (MT = 5: 80 registers)
  f32x4 a[2][MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * LDA);
#pragma unroll
  for (int u = 0; u < KPG; ++u) {
    if (u + 1 < KPG) {
#pragma unroll
      for (int m = 0; m < MT; ++m) a[(u + 1) & 1][m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * LDA + (u + 1) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = mfma4(a[u & 1][m][s], bf[u * NTL + t][s], acc[t][m]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// single tile, single row block: two accumulators (even / odd k-blocks) so the 40-cycle dependent latency never stalls
template <int LDA>
__device__ __forceinline__ void exec_group_11(const float* arow, const f32x4 (&bf)[4], f32x4 (&acc)[2]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(arow + u * 16);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[u & 1] = mfma4(a[s], bf[u][s], acc[u & 1]);
  }
}

// One layer for this wave: acc[t][m] += A[m-th 16 rows][K] W[K][tile t].  Entry: P holds the first fragment group.
// Exit: P holds the first group of the NEXT job (`wnext`, of kind NTLN) -- its loads fly over the epilogue and barrier.
template <int MT, int NTL, int KB, int LDA, int T1OFF, int NTLN, int T1OFFN, bool SPLIT11 = false, typename ACC>
__device__ __forceinline__ void run_job(const float* arow, __amdgpu_buffer_rsrc_t wr, int w, int wnext, int lane,
                                        f32x4 (&P)[4], f32x4 (&Q)[4], ACC& acc) {
  constexpr int KPG = 4 / NTL, G = KB / KPG;
  static_assert(G % 2 == 0, "even number of fragment groups per job");
#pragma unroll
  for (int gi = 0; gi < G; gi += 2) {
    load_group<NTL, T1OFF>(Q, wr, w + (gi + 1) * KPG * 1024, lane);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPLIT11) exec_group_11<LDA>(arow + gi * KPG * 16, P, acc); else exec_group<MT, NTL, LDA>(arow + gi * KPG * 16, P, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (gi + 2 < G) load_group<NTL, T1OFF>(P, wr, w + (gi + 2) * KPG * 1024, lane);
    else load_group<NTLN, T1OFFN>(P, wr, wnext, lane);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPLIT11) exec_group_11<LDA>(arow + (gi + 1) * KPG * 16, Q, acc); else exec_group<MT, NTL, LDA>(arow + (gi + 1) * KPG * 16, Q, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ static const float C5[5] = {1.f / 5, 3.f / 10, 4.f / 5, 8.f / 9, 1.f};     // stage times of DP_TAB rows 2..6

template <int D>
struct FTile {
  using S = FS<D>;
  static constexpr int TPW = S::TPW, LDX = S::LDX, LDH = S::LDH;
  static constexpr int OUT_T1OFF = 8 * (H / 16) * 1024;     // byte distance between column tiles w and w + 8 of a K = 128 layer
  float* lds;
  __amdgpu_buffer_rsrc_t wr;    // packed weights
  __amdgpu_buffer_rsrc_t sr;    // this workgroup's time-branch scratch
  int lane, wave, g, c, sign;
  float ffreq, coef, tbeta, clip;
  float tz1[4];

  __device__ __forceinline__ int W(int off_floats, int nt, int KB, int kb = 0) const { return off_floats * 4 + (nt * KB + kb) * 1024; }   // byte offset of a fragment
  __device__ __forceinline__ float bias(int off) const { return lds[S::BIAS + off]; }
  __device__ __forceinline__ f32x4 rs_get(int field) const { return *reinterpret_cast<const f32x4*>(lds + S::RS + field * 16 + 4 * g); }
  __device__ __forceinline__ void rs_put(int field, const float (&v)[4]) {
    if (wave == 0 && c == 0) *reinterpret_cast<f32x4*>(lds + S::RS + field * 16 + 4 * g) = f32x4{v[0], v[1], v[2], v[3]};
  }
  // partial sums of this lane's 4 rows over its 16 columns -> LDS [slot][row][wave]; totals after a barrier
  __device__ __forceinline__ void part_put(float* base, float (&p)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = group16_sum(p[i]);
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) base[(4 * g + i) * 8 + wave] = p[i];
    }
  }
  __device__ __forceinline__ void part_get(const float* base, float (&p)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(base + (4 * g + i) * 8), b = *reinterpret_cast<const f32x4*>(base + (4 * g + i) * 8 + 4);
      p[i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
    }
  }
  __device__ __forceinline__ void row_reduce(float (&p)[4], int slot) {
    part_put(lds + S::RED + slot * 128, p);
    __syncthreads();
    part_get(lds + S::RED + slot * 128, p);
  }

  // z W_x1 (no bias), once per solve.  Entry: P = first group of W2 tile `wave`; exit: P = first group of W0 tile `wave`.
  __device__ __forceinline__ void precompute_tz1(f32x4 (&P)[4], f32x4 (&Q)[4]) {
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    run_job<1, 1, D / 16, LDX, 0, 1, 0, true>(lds + S::ZB + 4 + (lane & 15) * LDX + 4 * g, wr, W(S::W2, wave, D / 16), W(S::W0, wave, 2 * F / 16), lane, P, Q, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) tz1[i] = acc[0][i] + acc[1][i];
  }

  // ---- the time branch for the five stage times of an attempt (phase 2), or one time replicated (phases 0, 1) ----
  // Entry: P = first group of W0 tile `wave`; row state visible.  Exit: P = first group of W2 tile `wave` (x1), and a
  // barrier has passed since every LDS access of this routine (region R is free for the x branch; a stage input written
  // by the caller BEFORE this call is visible).
  __device__ __forceinline__ void tbatch(int phase, f32x4 (&P)[4], f32x4 (&Q)[4]) {
    const int r = lane & 15;
    float sv[5][4];
    {
      const f32x4 t4 = rs_get(RS_T), h4 = rs_get(phase == 1 ? RS_H0 : RS_DT);
      const double f = (double)ffreq;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const float cs = phase == 0 ? 0.f : (phase == 1 ? 1.f : C5[s]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float tt = t4[i] + h4[i] * cs;
          const double te = sign > 0 ? (double)tt : 1.0 - (double)tt;          // :229
          double ft = f * te;
          ft -= rint(ft);
          float cv;
          sincospif(2.f * (float)ft, &sv[s][i], &cv);                          // :70-71
          lds[S::FH + (s * 16 + 4 * g + i) * LDH + 16 * wave + c] = cv;
        }
      }
    }
    __syncthreads();
    f32x4 acc[1][5];
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[0][m] = f32x4{0, 0, 0, 0};
    const float* afh = lds + S::FH + r * LDH + 4 * g;
    run_job<5, 1, 8, LDH, 0, 1, 0>(afh, wr, W(S::W0, wave, 16, 0), W(S::W0, wave, 16, 8), lane, P, Q, acc);      // cos half
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) lds[S::FH + (s * 16 + 4 * g + i) * LDH + 16 * wave + c] = sv[s][i];
    __syncthreads();
    run_job<5, 1, 8, LDH, 0, 1, 0>(afh, wr, W(S::W0, wave, 16, 8), W(S::W1, wave, 8), lane, P, Q, acc);           // sin half
    {
      const float b = bias(S::B0 + 16 * wave + c);
#pragma unroll
      for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) lds[S::T1 + (m * 16 + 4 * g + i) * LDH + 16 * wave + c] = fmaxf(acc[0][m][i] + b, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[0][m] = f32x4{0, 0, 0, 0};
    run_job<5, 1, 8, LDH, 0, 1, 0>(lds + S::T1 + r * LDH + 4 * g, wr, W(S::W1, wave, 8), W(S::W4, wave, 8), lane, P, Q, acc);
    {
      const float b = bias(S::B1 + 16 * wave + c);
#pragma unroll
      for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) lds[S::FH + (m * 16 + 4 * g + i) * LDH + 16 * wave + c] = fmaxf(acc[0][m][i] + b, 0.f);   // st
    }
    __syncthreads();
    const float* ast = afh;
#pragma unroll
    for (int q = 0; q < 3; ++q) {        // gate tiles wave, wave + 8 (D = 256; D = 128: one tile), then the st half of j1
      if (q == 1 && TPW == 1) continue;
#pragma unroll
      for (int m = 0; m < 5; ++m) acc[0][m] = f32x4{0, 0, 0, 0};
      float b;
      if (q < 2) {
        const int nxt = (q == 0 && TPW == 2) ? W(S::W4, wave + 8, 8) : W(S::W5, wave, 16, 8);
        run_job<5, 1, 8, LDH, 0, 1, 0>(ast, wr, W(S::W4, wave + 8 * q, 8), nxt, lane, P, Q, acc);
        b = bias(S::B4 + 16 * (wave + 8 * q) + c);
      } else {
        run_job<5, 1, 8, LDH, 0, 1, 0>(ast, wr, W(S::W5, wave, 16, 8), W(S::W2, wave, D / 16), lane, P, Q, acc);
        b = bias(S::B5 + 16 * wave + c);
      }
#pragma unroll
      for (int m = 0; m < 5; ++m) bstore(sr, lane * 16, ((m * NW + wave) * 3 + q) * 1024, f32x4{acc[0][m][0] + b, acc[0][m][1] + b, acc[0][m][2] + b, acc[0][m][3] + b});
    }
    __syncthreads();
  }

  // ---- one field evaluation (x branch) at the stage input in X buffer `cur`, time slot `slot` ----------------------
  // Entry: X[cur] visible to the workgroup, P = first group of W2 tile `wave`.  Exit: kv = dx/dt of this lane's
  // elements (row 4g+i, col 16 (wave + 8 q) + c); this wave's divergence partials in DLP[dst]; P = first group of
  // `wnext` (W2: another evaluation follows, W0: a time batch follows).
  __device__ __forceinline__ void eval(int slot, int cur, int dst, bool next_is_tbatch, f32x4 (&P)[4], f32x4 (&Q)[4], float (&kv)[TPW][4]) {
    const int r = lane & 15;
    const float* xb = lds + (cur ? S::XB1 : S::XB0);
    // stage-time inputs of this lane, written by itself in tbatch
    f32x4 gt[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) gt[q] = bload(sr, lane * 16, ((slot * NW + wave) * 3 + q) * 1024);
    const f32x4 j1t = bload(sr, lane * 16, ((slot * NW + wave) * 3 + 2) * 1024);
    {   // x1: value rows; tangent rows = relu' * (z W_x1)
      f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      run_job<1, 1, D / 16, LDX, 0, 1, 0, true>(xb + 4 + r * LDX + 4 * g, wr, W(S::W2, wave, D / 16), W(S::W3, wave, 8), lane, P, Q, acc);
      const float b = bias(S::B2 + 16 * wave + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = (acc[0][i] + acc[1][i]) + b;
        const int o = S::A1 + (4 * g + i) * LDH + 16 * wave + c;
        lds[o] = fmaxf(pre, 0.f);
        lds[o + 16 * LDH] = pre > 0.f ? tz1[i] : 0.f;
      }
    }
    __syncthreads();
    {   // x2
      f32x4 acc[1][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}};
      run_job<2, 1, 8, LDH, 0, 1, 0>(lds + S::A1 + r * LDH + 4 * g, wr, W(S::W3, wave, 8), W(S::W5, wave, 16, 0), lane, P, Q, acc);
      const float b = bias(S::B3 + 16 * wave + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = acc[0][0][i] + b;
        const int o = S::SX + (4 * g + i) * LDH + 16 * wave + c;
        lds[o] = fmaxf(pre, 0.f);
        lds[o + 16 * LDH] = pre > 0.f ? acc[0][1][i] : 0.f;
      }
    }
    __syncthreads();
    {   // j1: sx half of the concatenated input (:83); the st half + bias arrive as the initial accumulator
      f32x4 acc[1][2] = {{j1t, {0, 0, 0, 0}}};
      run_job<2, 1, 8, LDH, 0, 1, 0>(lds + S::SX + r * LDH + 4 * g, wr, W(S::W5, wave, 16, 0), W(S::W6, wave, 8), lane, P, Q, acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = acc[0][0][i];
        const int o = S::J1 + (4 * g + i) * LDH + 16 * wave + c;
        lds[o] = fmaxf(pre, 0.f);
        lds[o + 16 * LDH] = pre > 0.f ? acc[0][1][i] : 0.f;
      }
    }
    __syncthreads();
    {   // j2
      f32x4 acc[1][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}};
      run_job<2, 1, 8, LDH, 0, TPW, OUT_T1OFF>(lds + S::J1 + r * LDH + 4 * g, wr, W(S::W6, wave, 8), W(S::W7, wave, 8), lane, P, Q, acc);
      const float b = bias(S::B6 + 16 * wave + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = acc[0][0][i] + b;
        const int o = S::J2 + (4 * g + i) * LDH + 16 * wave + c;
        lds[o] = fmaxf(pre, 0.f);
        lds[o + 16 * LDH] = pre > 0.f ? acc[0][1][i] : 0.f;
      }
    }
    __syncthreads();
    {   // out: v = nn_xt + nn_t * clip(grad log pi(x)) (:88-90);  J z = d nn_xt . z + nn_t * 1[|g| <= clip] * (H z)
      float gc[TPW][4], hz[TPW][4], zz[TPW][4];
      const float* zb = lds + S::ZB;
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int col = 16 * (wave + NW * q) + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* xr = xb + (4 * g + i) * LDX + 4 + col;
          const float* zr = zb + (4 * g + i) * LDX + 4 + col;
          const float x = xr[0], z = zr[0];
          const float graw = -tbeta * (coef * (2.f * x - xr[-1] - xr[1]) - x * (1.f - x * x) / coef);
          const float hv = -tbeta * (coef * (2.f * z - zr[-1] - zr[1]) - (1.f - 3.f * x * x) * z / coef);
          gc[q][i] = clip > 0.f ? fminf(fmaxf(graw, -clip), clip) : graw;
          hz[q][i] = (!(clip > 0.f) || fabsf(graw) <= clip) ? hv : 0.f;
          zz[q][i] = z;
        }
      }
      f32x4 acc[TPW][2];
#pragma unroll
      for (int q = 0; q < TPW; ++q) { acc[q][0] = f32x4{0, 0, 0, 0}; acc[q][1] = f32x4{0, 0, 0, 0}; }
      const int wnext = next_is_tbatch ? W(S::W0, wave, 16) : W(S::W2, wave, D / 16);
      run_job<2, TPW, 8, LDH, OUT_T1OFF, 1, 0>(lds + S::J2 + r * LDH + 4 * g, wr, W(S::W7, wave, 8), wnext, lane, P, Q, acc);
      float dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const float b = bias(S::B7 + 16 * (wave + NW * q) + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = acc[q][0][i] + b + gt[q][i] * gc[q][i];
          const float jz = acc[q][1][i] + gt[q][i] * hz[q][i];
          dp[i] += zz[q][i] * jz;
          kv[q][i] = sign > 0 ? v : -v;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) dp[i] = sign > 0 ? -dp[i] : dp[i];          // :218 / :239
      part_put(lds + S::DLP + dst * 128, dp);
    }
  }
};

// Integrate the augmented ODE from t = 0 to 1 (see ode_solve in ode.hip: same state machine, same controller).
// Requires: Z filled (probe), halo pads of X0 / X1 / Z zero, biases in LDS.
template <int D>
__device__ __forceinline__ void solve(FTile<D>& T, float rtol, float atol, int max_attempts, float (&y)[FTile<D>::TPW][4],
                                      float (&ell)[4], int (&natt)[4]) {
  using S = FS<D>;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX;
  const int g = T.g, c = T.c, wave = T.wave;
  float* lds = T.lds;
  const float inv_n = 1.f / (float)(D + 1);
  float k[7][TPW][4];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) k[j][q][i] = 0.f;
  {
    const float z4[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                       // previous users of the row state / Z writers are done
#pragma unroll
    for (int fld = 0; fld < 14; ++fld) T.rs_put(fld, z4);
  }
  f32x4 P[4], Q[4];
  load_group<1, 0>(P, T.wr, T.W(S::W2, wave, D / 16), T.lane);
  __syncthreads();
  T.precompute_tz1(P, Q);

  int phase = 0, cur = 0;
#pragma unroll 1
  for (;;) {
    // ---- stage input: y + h * sum_j TAB[phase][j] k_j -> X[cur] ----
    float hs[4];
    {
      float cf[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) cf[j] = DP_TAB[phase][j];
      const f32x4 h4 = T.rs_get(phase == 1 ? RS_H0 : RS_DT);
#pragma unroll
      for (int i = 0; i < 4; ++i) hs[i] = h4[i];
      float* xw = lds + (cur ? S::XB1 : S::XB0);
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int col = 16 * (wave + NW * q) + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j) acc += cf[j] * k[j][q][i];
          xw[(4 * g + i) * LDX + 4 + col] = y[q][i] + hs[i] * acc;
        }
      }
    }
    if (phase <= 2) T.tbatch(phase, P, Q); else __syncthreads();
    float kv[TPW][4];
    const int dst = phase == 0 ? 0 : phase - 1 + (phase == 1 ? 1 : 0);
    T.eval(phase < 2 ? 0 : (phase == 7 ? 4 : phase - 2), cur, dst, phase == 7 || phase < 2, P, Q, kv);
    cur ^= 1;
    // ---- route the result: phase 0 -> k[0], phase 1 -> k[1], phase p >= 2 -> k[p - 1] ----
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j == dst) {
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) k[j][q][i] = kv[q][i];
      }

    if (phase == 0) {
      // ---- initial step size, part 1 (Hairer II.4, order 4) ----
      float p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sc = atol + fabsf(y[q][i]) * rtol;
          const float a0 = y[q][i] / sc, a1 = k[0][q][i] / sc;
          p0[i] += a0 * a0; p1[i] += a1 * a1;
        }
      T.part_put(lds + S::RED + 0 * 128, p0); T.part_put(lds + S::RED + 1 * 128, p1);
      __syncthreads();
      T.part_get(lds + S::RED + 0 * 128, p0); T.part_get(lds + S::RED + 1 * 128, p1);
      float dlv[4];
      T.part_get(lds + S::DLP + 0 * 128, dlv);
      float h0[4], d1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a1 = dlv[i] / atol;                                // ell0 = 0 -> scale = atol
        const float d0 = sqrtf(p0[i]); d1[i] = sqrtf(p1[i] + a1 * a1);
        h0[i] = (d0 < 1e-5f || d1[i] < 1e-5f) ? 1e-6f : 0.01f * d0 / d1[i];
      }
      T.rs_put(RS_H0, h0); T.rs_put(RS_D1, d1); T.rs_put(RS_KL + 0, dlv);
      __syncthreads();
      phase = 1;
    } else if (phase == 1) {
      float p2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sc = atol + fabsf(y[q][i]) * rtol;
          const float a2 = (k[1][q][i] - k[0][q][i]) / sc;
          p2[i] += a2 * a2;
        }
      T.part_put(lds + S::RED + 2 * 128, p2);
      __syncthreads();
      T.part_get(lds + S::RED + 2 * 128, p2);
      float dlv[4];
      T.part_get(lds + S::DLP + 1 * 128, dlv);
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
        any |= (dt[i] > 0.f);
      }
      T.rs_put(RS_DT, dt);
      phase = 2;
      if (!__syncthreads_or(any ? 1 : 0)) break;
    } else if (phase < 7) {
      phase += 1;
    } else {
      // ---- end of an attempted step ----
      float y1[TPW][4], e2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float acc = 0.f, er = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j) acc += DP_TAB[7][j] * k[j][q][i];
          y1[q][i] = y[q][i] + hs[i] * acc;                    // the stage-7 input (5th-order solution), same arithmetic
#pragma unroll
          for (int j = 0; j < 7; ++j) er += DP_E[j] * k[j][q][i];
          er *= hs[i];
          const float tol = atol + rtol * fmaxf(fabsf(y[q][i]), fabsf(y1[q][i]));
          const float rr = er / tol;
          e2[i] += rr * rr;
        }
      T.part_put(lds + S::RED + 3 * 128, e2);
      __syncthreads();
      T.part_get(lds + S::RED + 3 * 128, e2);
      const f32x4 t4 = T.rs_get(RS_T), ell4 = T.rs_get(RS_ELL);
      const f32x4 na4 = T.rs_get(RS_NATT), dn4 = T.rs_get(RS_DONE), kl04 = T.rs_get(RS_KL + 0);
      float kl[7][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kl[0][i] = kl04[i];
#pragma unroll
      for (int j = 1; j < 7; ++j) T.part_get(lds + S::DLP + j * 128, kl[j]);
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
        const bool acc = active && ratio <= 1.f;
        const float dfac = ratio < 1.f ? 1.f : 0.2f;
        const float fac = fminf(10.f, fmaxf(0.9f * powf(ratio, -0.2f), dfac));
        const float ndt = fmaxf(ratio == 0.f ? dti * 10.f : dti * fac, 0.f);
        t_n[i] = t4[i]; ell_n[i] = ell4[i]; kl0_n[i] = kl[0][i]; dn_n[i] = dn4[i];
        if (acc) {
          const float tn = t4[i] + dti;
          if (tn >= 1.f) {
            // final output: 4th-order interpolant of this step evaluated at t = 1
            const float sfrac = (1.f - t4[i]) / (tn - t4[i]);
            float lm = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) lm += DP_M[j] * kl[j][i];
            const float y0 = ell4[i], yy1 = l1, ym = y0 + dti * lm, f0 = dti * kl[0][i], f1 = dti * kl[6][i];
            const float pa = -2.f * f0 + 2.f * f1 - 8.f * y0 - 8.f * yy1 + 16.f * ym;
            const float pb = 5.f * f0 - 3.f * f1 + 18.f * y0 + 14.f * yy1 - 32.f * ym;
            const float pc = -4.f * f0 + f1 - 11.f * y0 - 5.f * yy1 + 16.f * ym;
            ell_n[i] = (((pa * sfrac + pb) * sfrac + pc) * sfrac + f0) * sfrac + y0;
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
              float km = 0.f;
#pragma unroll
              for (int j = 0; j < 7; ++j) km += DP_M[j] * k[j][q][i];
              const float x0 = y[q][i], x1 = y1[q][i], xm = x0 + dti * km, g0 = dti * k[0][q][i], g1 = dti * k[6][q][i];
              const float qa = -2.f * g0 + 2.f * g1 - 8.f * x0 - 8.f * x1 + 16.f * xm;
              const float qb = 5.f * g0 - 3.f * g1 + 18.f * x0 + 14.f * x1 - 32.f * xm;
              const float qc = -4.f * g0 + g1 - 11.f * x0 - 5.f * x1 + 16.f * xm;
              y[q][i] = (((qa * sfrac + qb) * sfrac + qc) * sfrac + g0) * sfrac + x0;
            }
            dn_n[i] = 1.f;
          } else {
            ell_n[i] = l1;
#pragma unroll
            for (int q = 0; q < TPW; ++q) { y[q][i] = y1[q][i]; k[0][q][i] = k[6][q][i]; }
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

template <int D>
__device__ __forceinline__ void tile_init(FTile<D>& T, const NetDev& n, float* lds, f32x4* scr_wg) {
  using S = FS<D>;
  T.lds = lds;
  T.lane = threadIdx.x & 63; T.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); T.g = T.lane >> 4; T.c = T.lane & 15;
  T.wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(n.Wp), 0, S::WTOT * 4, 0x00020000);
  T.sr = __builtin_amdgcn_make_buffer_rsrc(scr_wg, 0, SCR_F4_PER_WG * 16, 0x00020000);
  T.sign = 1;
  T.ffreq = n.fourier[16 * T.wave + T.c];
  T.coef = n.T.coef; T.tbeta = n.T.tbeta; T.clip = n.grad_clip;
  for (int i = threadIdx.x; i < S::BIAS; i += NW * 64) lds[i] = 0.f;          // halo pads, row state, scratch
  for (int i = threadIdx.x; i < S::BTOT; i += NW * 64) lds[S::BIAS + i] = n.bias[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) T.tz1[i] = 0.f;
  __syncthreads();
}

template <int D>
__device__ __forceinline__ void fill_probe(FTile<D>& T, const float* z, int b0) {
  using S = FS<D>;
#pragma unroll
  for (int q = 0; q < FTile<D>::TPW; ++q) {
    const int col = 16 * (T.wave + NW * q) + T.c;
#pragma unroll
    for (int i = 0; i < 4; ++i) T.lds[S::ZB + (4 * T.g + i) * S::LDX + 4 + col] = z[(size_t)(b0 + 4 * T.g + i) * D + col];
  }
}

template <int D>
__global__ __launch_bounds__(NW * 64) void ode_transform_fast_kernel(OdeArgs a, f32x4* scratch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TPW = FTile<D>::TPW;
  FTile<D> T;
  tile_init(T, a.net, lds, scratch + (size_t)blockIdx.x * SCR_F4_PER_WG);
  T.sign = a.direction;
#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.n / 16; tile += gridDim.x) {
    const int b0 = tile * 16;
    __syncthreads();
    fill_probe(T, a.z1, b0);
    float y[TPW][4], ell[4]; int natt[4];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = 16 * (T.wave + NW * q) + T.c;
#pragma unroll
      for (int i = 0; i < 4; ++i) y[q][i] = a.in[(size_t)(b0 + 4 * T.g + i) * D + col];
    }
    solve<D>(T, a.rtol, a.atol, a.max_attempts, y, ell, natt);
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = 16 * (T.wave + NW * q) + T.c;
#pragma unroll
      for (int i = 0; i < 4; ++i) a.out[(size_t)(b0 + 4 * T.g + i) * D + col] = y[q][i];
    }
    if (T.wave == 0 && T.c == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a.ldj[b0 + 4 * T.g + i] = ell[i];
        if (a.nsteps) a.nsteps[b0 + 4 * T.g + i] = natt[i];
      }
    }
  }
}

// One flow-based MH step per chain (random-walk in latent space :264-278, or independent :246-260), PhiFour target.
template <int D>
__global__ __launch_bounds__(NW * 64) void flow_step_fast_kernel(OdeArgs a, FlowArgs f, f32x4* scratch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using S = FS<D>;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX;
  FTile<D> T;
#ifdef MFM_STAMPS
  const unsigned long long fc0_ = __builtin_amdgcn_s_memtime(), fr0_ = __builtin_amdgcn_s_memrealtime();
#endif
  tile_init(T, a.net, lds, scratch + (size_t)blockIdx.x * SCR_F4_PER_WG);
  const int b0 = blockIdx.x * 16, g = T.g, c = T.c, wave = T.wave;
  float y[TPW][4], ell[4], vol0[4] = {0, 0, 0, 0}, lq_ref[4] = {0, 0, 0, 0};
  int natt[4], natt_tot[4] = {0, 0, 0, 0};
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {       // ONE call site of the solver: inverse solve, then forward solve of the proposal
    if (ph == 0) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int col = 16 * (wave + NW * q) + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[q][i] = f.pos[(size_t)(b0 + 4 * g + i) * D + col];                 // :267 / :251
      }
    } else {
      float r0[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0};
      const float scale = 2.38f / sqrtf((float)D);                                                  // :262
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int col = 16 * (wave + NW * q) + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float nz = a.zgen[(size_t)(b0 + 4 * g + i) * D + col];
          if (f.mode == MFM_FLOW_RWMH) y[q][i] = y[q][i] + scale * nz;                              // :268
          else { r0[i] += y[q][i] * y[q][i]; y[q][i] = nz; r1[i] += nz * nz; }                      // :249
        }
      }
      if (f.mode == MFM_FLOW_IMH) {     // ref.logprob(u0) - ref.logprob(up) = -(|u0|^2 - |up|^2) / 2   (:254-255)
        __syncthreads();
        T.part_put(lds + S::RED + 0 * 128, r0); T.part_put(lds + S::RED + 1 * 128, r1);
        __syncthreads();
        T.part_get(lds + S::RED + 0 * 128, r0); T.part_get(lds + S::RED + 1 * 128, r1);
#pragma unroll
        for (int i = 0; i < 4; ++i) lq_ref[i] = -0.5f * (r0[i] - r1[i]);
      }
      __syncthreads();
    }
    fill_probe(T, ph == 0 ? a.z1 : a.z2, b0);       // key_hutch2 for the inverse, key_hutch1 for the forward solve
    T.sign = ph == 0 ? -1 : 1;
    solve<D>(T, a.rtol, a.atol, a.max_attempts, y, ell, natt);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ph == 0) vol0[i] = ell[i];
      natt_tot[i] += natt[i];
    }
  }
  // ---- target at the proposal (:270 / :252), tempered: beta * loglik (logprior = 0) ----
  __syncthreads();
  float* xw = lds + S::XB0;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = 16 * (wave + NW * q) + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) xw[(4 * g + i) * LDX + 4 + col] = y[q][i];
  }
  __syncthreads();
  double lpn[4];
  float gnew[TPW][4];
  {
    double part[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = 16 * (wave + NW * q) + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* xr = xw + (4 * g + i) * LDX + 4;
        part[i] += phi4_term(a.net.T, xr, col);
        gnew[q][i] = (float)f.beta * phi4_grad(a.net.T, xr, col);
      }
    }
    double* rd = reinterpret_cast<double*>(lds + S::RED);      // [NW][16 rows] doubles = 2 slots
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) part[i] += __shfl_xor(part[i], o, 64);
      if (c == 0) rd[wave * 16 + 4 * g + i] = part[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += rd[w * 16 + 4 * g + i];
      lpn[i] = f.beta * t;
    }
  }
  // ---- accept / reject (:271-278 / :253-260); the acceptance probability is NOT clipped ----
  bool acc[4];
  float aprob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = b0 + 4 * g + i;
    const Key2 kb = split_at(f.key, f.n_total, f.chain_offset + (uint32_t)b);             // :303
    const double lp_old = f.logp[b];
    const double la = lpn[i] - (double)ell[i] - lp_old - (double)vol0[i] + (double)lq_ref[i];
    const double ap = exp(la);
    const double u = uniform01(split_at(kb, 4, 1), 0, 1);
    acc[i] = u <= ap;                     // NaN compares false -> reject
    aprob[i] = (float)ap;
  }
  __syncthreads();      // every wave has read the OLD log-densities before wave 0 publishes the accepted ones
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = 16 * (wave + NW * q) + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t o = (size_t)(b0 + 4 * g + i) * D + col;
      if (f.proposed) f.proposed[o] = y[q][i];
      if (acc[i]) { f.pos[o] = y[q][i]; f.grad[o] = gnew[q][i]; }
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
    unsigned long long* o = g_flow_dbg + blockIdx.x * 8;
    o[0] = __builtin_amdgcn_s_memtime() - fc0_; o[1] = __builtin_amdgcn_s_memrealtime() - fr0_;
    o[2] = 0; o[3] = 0;
  }
#endif
}

// ---- dispatch --------------------------------------------------------------------------------------------------
static bool shape_ok(const NetDev& n, int hutch) {
  if (!hutch || n.T.kind != MFM_TARGET_PHI4) return false;
  if (n.F != F || n.ht1 != H || n.ht2 != H || n.hx1 != H || n.hx2 != H || n.hj1 != H || n.hj2 != H) return false;
  return n.d == 256 || n.d == 128;
}
static int max_wgs() { return ODE_FAST_MAX_WGS; }

template <int D>
static int launch_flow_t(const OdeArgs& a, const FlowArgs& f, f32x4* scratch, hipStream_t stream) {
  const size_t sm = (size_t)FS<D>::TOTAL * sizeof(float);
  (void)hipFuncSetAttribute((const void*)flow_step_fast_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL((flow_step_fast_kernel<D>), dim3(a.n / 16), dim3(NW * 64), sm, stream, a, f, scratch);
  return 0;
}
template <int D>
static int launch_transform_t(const OdeArgs& a, f32x4* scratch, hipStream_t stream) {
  const size_t sm = (size_t)FS<D>::TOTAL * sizeof(float);
  (void)hipFuncSetAttribute((const void*)ode_transform_fast_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  const int tiles = a.n / 16, grid = tiles < max_wgs() ? tiles : max_wgs();
  hipLaunchKernelGGL((ode_transform_fast_kernel<D>), dim3(grid), dim3(NW * 64), sm, stream, a, scratch);
  return 0;
}

}  // namespace fast
