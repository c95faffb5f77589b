// K5f / K6f: the flow-MH step and the CNF transforms for the HEADLINE network shape (fourier_dim = 128, all hidden
// widths 128, dim 128 or 256 -- narrower lattices zero-padded to those, see the dispatch section --, Hutchinson log-det, PhiFour target), as a shape-specialised sibling of the generic solver
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
//  * THE OUT LAYER CARRIES NO TANGENT ROWS.  The Hutchinson integrand only needs the SCALAR z . (J z), and the last layer is
//    linear: z . (W_out^T tj2) = (W_out z) . tj2.  w7z = W_out z (128 values per row) is formed once per solve, next to
//    z W_x1, and the tangent's contribution to the divergence becomes a 128-term dot product in the j2 epilogue; the out
//    layer runs on the 16 value rows only (64 instead of 128 MFMAs per wave: 320 instead of 384 per evaluation).
//
// Everything that is not the solver core (probes, proposal, target evaluation at the proposal, accept / reject) follows
// flow_step_kernel in ode.hip.  Other shapes / targets / the exact-trace mode keep using the generic kernels.

namespace fast {

constexpr int NW = 8, H = 128, F = 128;
constexpr int SCR_F4_PER_WG = ODE_FAST_SCR_F4;      // float4 per workgroup: [slot 5][wave 8][gate q0, gate q1, j1t][lane 64]

template <int D>
struct FS {                                         // float offsets (LDS, packed weights, biases)
  static constexpr int TPW = D / 128;
  // leading dimensions = 8 mod 64 dwords: the 16-lane groups of a ds_read_b128 A-fragment read (row = lane & 15, k offset
  // 4 (lane >> 4)) then touch 64 distinct banks; K + 4 (= 4 mod 64) put two lanes of every group on the same four banks
  static constexpr int LDX = D + 8, LDH = H + 8;
  // Small, often-addressed structures first: every LDS access below is (one of a handful of per-lane base registers) +
  // a constant that fits the 16-bit DS offset field.  (Constants beyond 64 KB each cost a register, and the compiler
  // hoists all of them out of the solver loop: with the row state at 135 KB that alone spilled ~250 registers.)
  static constexpr int BTOT = 6 * H + 2 * D;
  static constexpr int RS = 0, RED = RS + 384, DLP = RED + 6 * 128, BIAS = DLP + 8 * 128;   // 24 row-state fields, 6 reduction slots
  static constexpr int XB0 = BIAS + BTOT, XB1 = XB0 + 16 * LDX, ZB = XB1 + 16 * LDX, R = ZB + 16 * LDX;
  // region R, x branch (32 rows each): A1 (x1 out), SX at R; J1, J2 at R2 = R + 64 LDH.  Time batch (80 rows each):
  // FH (Fourier features, later st) at R, T1 at R2 + 16 LDH
  static constexpr int R2 = R + 64 * LDH, TOTAL = R + 160 * LDH;
  static_assert((ZB + 16 * LDX) * 4 <= 65536 && 96 * LDH * 4 + 16 * LDH * 4 <= 65536, "DS offset field");
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
// Stores keep the scalar offset OUT of the soffset field (it is added into the per-lane offset instead): a
// buffer_store_dwordx4 with an SGPR soffset followed at once by a VALU write of one of its data registers stored the
// NEW value of that register on gfx950 (one element of the float4 wrong, which element depending on register
// allocation).  hipcc only pads that write-after-read hazard when soffset is not a register, so that is the form used.
// The per-lane offset is formed IN PLACE by a volatile add: as a plain `voff + soff` every store's offset is a loop invariant
// the compiler hoists out of the solver loop, and with 256 VGPRs in use it spilled them all (a scratch reload and a
// vmcnt(0) in front of each store of the time batch).
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, f32x4 v) {
  int off;
  asm volatile("v_add_u32 %0, %1, %2" : "=v"(off) : "s"(soff), "v"(voff));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
template <int NTL, int T1OFF>
__device__ __forceinline__ void load_group(f32x4 (&bf)[4], __amdgpu_buffer_rsrc_t r, int soff, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bf[j] = NTL == 1 ? bload(r, lane * 16 + j * 1024, soff) : bload(r, lane * 16 + (j >> 1) * 1024, soff + (j & 1) * T1OFF);
}

#ifndef FAST_DBUF_MAX_MT
#define FAST_DBUF_MAX_MT 2      // (round 4: 1 -> 2, x2 / j1 / j2 read their two A fragments one k-block ahead: -0.5 % on the launch)
#endif
template <int MT, int NTL, int LDA>
__device__ __forceinline__ void exec_group(const float* arow, const f32x4 (&bf)[4], f32x4 (&acc)[NTL][MT]) {
  constexpr int KPG = 4 / NTL;
  if constexpr (MT <= FAST_DBUF_MAX_MT) {
    // A fragments one k-block ahead (two register sets, statically renamed by the unroll): the LDS latency of block
    // u + 1 hides behind the MFMAs of block u
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
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTL; ++t) acc[t][m] = mfma4(a[u & 1][m][s], bf[u * NTL + t][s], acc[t][m]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // ONE set of A fragments, refreshed in place: a[m] is reloaded for k-block u + 1 right after its last use in block u
    // (k-step 3), i.e. MT - 1 MFMAs before block u + 1 needs it.  The scheduling barriers pin that order: left alone,
    // the scheduler hoists a whole group's reads (MT = 5: 80 registers).
    f32x4 a[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * LDA);
#pragma unroll
    for (int u = 0; u < KPG; ++u) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int t = 0; t < NTL; ++t) acc[t][m] = mfma4(a[m][s], bf[u * NTL + t][s], acc[t][m]);
          if (s == 3 && u + 1 < KPG) {
            __builtin_amdgcn_sched_barrier(0);
            a[m] = *reinterpret_cast<const f32x4*>(arow + m * 16 * LDA + (u + 1) * 16);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
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

// ---- the same job on FOUR rows (v_mfma_f32_4x4x1_16b_f32), for a tile's last one or two chains -------------------------------
// A tile whose other chains have finished still pushes 16-row MFMA tiles through every layer: the launch lasts as long as its
// slowest chain, and that chain runs its last 100-230 attempts alone or with one neighbour (tools/tail_stats.py).  The 4x4x1
// MFMA computes 16 independent 4 x 4 outer products per instruction (block b = lanes 4b .. 4b + 3: D[i][j] += A_b[i] B_b[j], lane
// 4b + j supplies A_b[j] and B_b[j] and holds column j of D in its four registers; layout and rate measured in tools/mb/m4.hip:
// ~9.6 cycles per instruction per wave at two waves per SIMD, against 32 for the 16 x 16 x 4 form).  Fed with the SAME streamed
// fragment as the 16 x 16 x 4 path -- lane (g, c) holds W[16 kb + 4 g + s][16 nt + c] -- block (g, c >> 2) multiplies rows
// A[c & 3][16 kb + 4 g + s] into columns 16 nt + c: each lane accumulates, for its own column, the k-subset of its g over the four
// M-rows, and the four g-groups are summed at the end of the job (gsum).  Same weights, same LDS images, a quarter of the
// matrix-pipe time per k-block.  M-rows: values of ranks 0, 1, then their tangent rows.
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ static const float C5[5] = {1.f / 5, 3.f / 10, 4.f / 5, 8.f / 9, 1.f};     // stage times of DP_TAB rows 2..6

// sin(pi x), cos(pi x) for |x| <= 1 (the Fourier features' argument is reduced to half a turn in float64 before it gets here):
// x = k / 2 + r with |r| <= 1 / 4, degree-9 / degree-10 Taylor polynomials in r (truncation < 2e-9), quadrant by k.  Max error
// 9e-8 on [-1, 1] (0.75 ulp of 1: as the library's sincospif, which carries the general range reduction and special cases).
#ifndef MFM_LIB_SINCOSPI
__device__ __forceinline__ void sincospi_half_turn(float x, float* sn, float* cs) {
  const float k = rintf(2.f * x);
  const float r = __builtin_fmaf(-0.5f, k, x), r2 = r * r;
  float sp = 0.08214588661112823f;                                  // pi^9 / 9!
  sp = __builtin_fmaf(sp, r2, -0.5992645293207921f);                // -pi^7 / 7!
  sp = __builtin_fmaf(sp, r2, 2.5501640398773455f);                 // pi^5 / 5!
  sp = __builtin_fmaf(sp, r2, -5.16771278004997f);                  // -pi^3 / 3!
  sp = __builtin_fmaf(sp, r2, 3.141592653589793f) * r;
  float cp = -0.02580689139001406f;                                 // -pi^10 / 10!
  cp = __builtin_fmaf(cp, r2, 0.23533063035889312f);                // pi^8 / 8!
  cp = __builtin_fmaf(cp, r2, -1.3352627688545893f);                // -pi^6 / 6!
  cp = __builtin_fmaf(cp, r2, 4.058712126416768f);                  // pi^4 / 4!
  cp = __builtin_fmaf(cp, r2, -4.934802200544679f);                 // -pi^2 / 2!
  cp = __builtin_fmaf(cp, r2, 1.f);
  const int q = (int)k & 3;
  *sn = q == 0 ? sp : (q == 1 ? cp : (q == 2 ? -sp : -cp));
  *cs = q == 0 ? cp : (q == 1 ? -sp : (q == 2 ? -cp : sp));
}
#else
__device__ __forceinline__ void sincospi_half_turn(float x, float* sn, float* cs) { sincospif(x, sn, cs); }
#endif

// Row-state fields beyond ode.hip's RS_* (0..15), used by the flow step's per-row solve phases (solve2): every chain of a
// tile runs its OWN sequence inverse solve -> proposal -> forward solve; the tile only shares the attempt clock.
enum { RS_MODE = 16, RS_SOLVE = 17, RS_SIGN = 18, RS_VOL0 = 19, RS_NTOT = 20, RS_LQ = 21, RS_SW = 22, RS_TILE = 23,
       RS_RANK = 13 /* = RS_DONE, unused by solve2: rank of the row among the rows still integrating, -1 otherwise */ };
enum { RM_INIT0 = 0, RM_INIT1 = 1, RM_ATT = 2, RM_DONE = 3 };
#ifndef TAIL_PASSES
#define TAIL_PASSES 2         // a tile enters the tail with <= 2 TAIL_PASSES live rows (1: the round-3 two-row tail)
#endif

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
  float w7z[4];                 // (W_out z)[row 4g + i][col 16 wave + c]: the probe pulled back through the (linear) out layer
  __amdgpu_buffer_rsrc_t wtr;   // transposed packed weights (pack of W^T), for that pull-back
  // per-lane LDS base offsets in BYTES, opaque to the optimiser (see FS): row state / partial-sum reads (rows 4g..),
  // partial-sum writes, bias column, A-fragment reads and owned-element accesses in the X / Z buffers, A-fragment reads
  // and epilogue writes in region R and in R2 = R + 64 LDH
  int o_rs, o_pg, o_pp, o_bias, o_xa, o_xo, o_ha, o_ha2, o_he, o_he2, o_l8, o_l1;    // o_l8 / o_l1: row leaders (row = lane)
  int o_hc;                     // region R, row 0, this lane's column (compact time batch: the row is data dependent)
  int o_xc;                     // X buffers, row 0, this lane's column (compact evaluation: rows are ranks)
  __device__ __forceinline__ float* at(int off_bytes, int cfloats) const { return reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + off_bytes) + cfloats; }
#ifdef MFM_STAMPS
  unsigned long long n_em = 0, cyc_em = 0, n_t1 = 0, cyc_t1 = 0;      // micro evaluations, single-tile time batches
  unsigned long long cyc_si = 0, cyc_n1 = 0, cyc_n2 = 0, cyc_n3 = 0;      // main loop outside the batch and the evaluations: stage inputs, norms, leaders' barrier, decision
  unsigned long long cyc_tail = 0, cyc_done = 0;      // time in the tail loops; kernel start -> this tile done (before the noise work)
  unsigned long long n_tc = 0, cyc_tc = 0;      // compacted time batches
  unsigned long long n_ec = 0, cyc_ec = 0;      // compact evaluations
  unsigned long long n_eval = 0, cyc_eval = 0, n_tb = 0, cyc_tb = 0, cyc_sec[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sec_t0 = 0;
  unsigned long long cyc_csec[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // sections of the compact evaluation
#ifdef MFM_STAMPS_FINE      // section stamps INSIDE the batch and the evaluations: every stamp waits for the wave's outstanding LDS / scalar
                            // operations, so this build runs ~15 % slower and inflates short sections; the coarse stamps alone cost < 1 %
#define FSEC(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cyc_sec[i] += t_ - sec_t0; sec_t0 = t_; } while (0)
#define CSEC(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cyc_csec[i] += t_ - sec_t0; sec_t0 = t_; } while (0)
#else
#define FSEC(i) do {} while (0)
#define CSEC(i) do {} while (0)
#endif
#else
#define FSEC(i) do {} while (0)
#define CSEC(i) do {} while (0)
#endif

  __device__ __forceinline__ int W(int off_floats, int nt, int KB, int kb = 0) const { return off_floats * 4 + (nt * KB + kb) * 1024; }   // byte offset of a fragment
  __device__ __forceinline__ float bias(int off) const { return *at(o_bias, S::BIAS + off); }      // column 16 wave + c (+ 128 q) of a layer
  __device__ __forceinline__ f32x4 rs_get(int field) const { return *reinterpret_cast<const f32x4*>(at(o_rs, S::RS + field * 16)); }
  __device__ __forceinline__ void rs_put(int field, const float (&v)[4]) {
    if (wave == 0 && c == 0) *reinterpret_cast<f32x4*>(at(o_rs, S::RS + field * 16)) = f32x4{v[0], v[1], v[2], v[3]};
  }
  // partial sums of this lane's 4 rows over its 16 columns -> LDS [slot][row][wave]; totals after a barrier
  __device__ __forceinline__ void part_put(int base, float (&p)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = group16_sum_dpp(p[i]);
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *at(o_pp, base + i * 8) = p[i];
    }
  }
  __device__ __forceinline__ void part_get(int base, float (&p)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(at(o_pg, base + i * 8)), b = *reinterpret_cast<const f32x4*>(at(o_pg, base + i * 8 + 4));
      p[i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
    }
  }
  __device__ __forceinline__ void row_reduce(float (&p)[4], int slot) {
    part_put(S::RED + slot * 128, p);
    __syncthreads();
    part_get(S::RED + slot * 128, p);
  }

  // z W_x1 (no bias), once per solve.  Entry: P = first group of W2 tile `wave`; exit: P = first group of W0 tile `wave`.
  __device__ __forceinline__ void precompute_tz1(f32x4 (&P)[4], f32x4 (&Q)[4]) {
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    run_job<1, 1, D / 16, LDX, 0, 1, 0, true>(at(o_xa, S::ZB), wr, W(S::W2, wave, D / 16), W(S::W0, wave, 2 * F / 16), lane, P, Q, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) tz1[i] = acc[0][i] + acc[1][i];
  }

  // W_out z = z W_out^T (no bias), once per solve: the transposed pack of layer 7 IS a [D -> 128] layer like W_x1.
  // Entry: P = first group of WpT layer 7 tile `wave` (from wtr); exit: P = first group of W2 tile `wave` (precompute_tz1 next).
  __device__ __forceinline__ void precompute_w7z(f32x4 (&P)[4], f32x4 (&Q)[4]) {
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    run_job<1, 1, D / 16, LDX, 0, 1, 0, true>(at(o_xa, S::ZB), wtr, W(S::W7, wave, D / 16), W(S::W7, wave, D / 16), lane, P, Q, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) w7z[i] = acc[0][i] + acc[1][i];
    load_group<1, 0>(P, wr, W(S::W2, wave, D / 16), lane);
  }

  // ---- the time branch for the five stage times of an attempt (phase 2), or one time replicated (phases 0, 1) ----
  // Entry: P = first group of W0 tile `wave`; row state visible.  Exit: P = first group of W2 tile `wave` (x1), and a
  // barrier has passed since every LDS access of this routine (region R is free for the x branch; a stage input written
  // by the caller BEFORE this call is visible).
  // MTB = 5: the 80 (stage, row) pairs of the tile, stage-major (M-row 16 s + row).  MTB = 1 (flow step, <= 3 rows still
  // integrating): the <= 15 pairs of the ACTIVE rows compacted into ONE M tile (M-row 3 s + rank(row)) -- a fifth of the
  // matrix work of the time branch for the attempts of a tile's tail, which is what the launch waits for.
  // Fourier features of the stage times (:70-71): ONE copy of the sincos code, shared by every time-batch variant (the
  // three variants of the matrix part below used to carry it each: 3 x ~10 KB of a 64 KB instruction cache shared by two CUs)
  template <bool ROWMODE>
  __device__ __forceinline__ void tb_trig(int phase, float (&cv)[5][4], float (&sv)[5][4]) {
    const f32x4 t4 = rs_get(RS_T), h4 = rs_get(phase == 1 ? RS_H0 : RS_DT);
    f32x4 md4 = {2.f, 2.f, 2.f, 2.f}, sg4 = {(float)sign, (float)sign, (float)sign, (float)sign};
    if constexpr (ROWMODE) { md4 = rs_get(RS_MODE); sg4 = rs_get(RS_SIGN); }
    const double f = (double)ffreq;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float cs = phase == 0 ? 0.f : (phase == 1 ? 1.f : C5[s]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // a row in its initial-step phases rides along the attempt: INIT0 has dt = 0 (every slot is t0), INIT1 has
        // dt = h0 and takes the extra evaluation of the step-size heuristic in slot 0 at t0 + h0
        const float csr = ROWMODE ? (md4[i] == (float)RM_INIT1 ? (s == 0 ? 1.f : 0.f) : cs) : cs;
        const float tt = t4[i] + h4[i] * csr;
        const double te = sg4[i] > 0.f ? (double)tt : 1.0 - (double)tt;          // :229
        double ft = f * te;
        ft -= rint(ft);
        sincospi_half_turn(2.f * (float)ft, &sv[s][i], &cv[s][i]);             // :70-71
      }
    }
  }

  template <bool ROWMODE = false, int MTB = 5>
  __device__ __forceinline__ void tbatch(int phase, f32x4 (&P)[4], f32x4 (&Q)[4], const float (&cv)[5][4], const float (&sv)[5][4]) {
    constexpr int RPS = MTB == 1 ? 3 : 8;   // compact modes: rows per stage (M-row RPS * s + rank)
#ifdef MFM_STAMPS
    sec_t0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr bool CMP = MTB != 5;
    // MTB = 3, 5 (round 4): cos | sin SIDE BY SIDE in one [16 MTB][264] image and ONE K = 256 job over it (the same k order as the
    // two half jobs it replaces: bit-identical).  The sine block used to wait in the global scratch while the cosine half ran,
    // with two more barriers and a second write pass.  The image overlaps the place of t1, hence the barrier after the job.
    constexpr bool SIDE = MTB != 1;
    constexpr int LDF = 2 * F + 8;
    static_assert(!SIDE || 16 * MTB * LDF <= 160 * LDH, "Fourier image");
    int mrow[5][4];                         // compact mode: LDS row of (stage s, this lane's row i), -1: row not integrating
    {
      f32x4 rk4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (CMP) rk4 = rs_get(RS_RANK);
      const int o_hf = o_hc + g * (4 * LDF * 4);      // image row 4 g, this lane's column
#pragma unroll
      for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (CMP) {
            mrow[s][i] = rk4[i] >= 0.f ? RPS * s + (int)rk4[i] : -1;
            if constexpr (SIDE) { if (mrow[s][i] >= 0) { *at(o_hc, mrow[s][i] * LDF) = cv[s][i]; *at(o_hc, mrow[s][i] * LDF + F) = sv[s][i]; } }
            else if (mrow[s][i] >= 0) *at(o_hc, mrow[s][i] * LDH) = cv[s][i];
          } else {
            if constexpr (SIDE) { *at(o_hf, (s * 16 + i) * LDF) = cv[s][i]; *at(o_hf, (s * 16 + i) * LDF + F) = sv[s][i]; }
            else *at(o_he, (s * 16 + i) * LDH) = cv[s][i];
          }
        }
        // the sine block waits in this lane's scratch slot of the stage (rewritten by the gate epilogue afterwards)
        if constexpr (!SIDE) bstore(sr, lane * 16, ((s * NW + wave) * 3 + 0) * 1024, f32x4{sv[s][0], sv[s][1], sv[s][2], sv[s][3]});
      }
    }
    FSEC(0);
    __syncthreads();
    FSEC(1);
    const float* afh = at(o_ha, 0);
    // MTB = 1 runs its single tile with two accumulators (even / odd k-blocks): one dependent MFMA chain per wave would leave
    // the matrix pipe waiting on its own latency.  That reassociates the k-sum of the time branch (float rounding only): a
    // row's attempts after its tile compacts can differ in the last bits from the stage-major schedule.  Tiles are fixed
    // groups of 16 consecutive global chains, so results stay deterministic and independent of the sharding.  (Giving the
    // MTB = 5 mode the same even / odd order -- 10 accumulators -- cost it 6 %: measured, not kept.)
    f32x4 acc[1][MTB], ac2[2];
    auto zero = [&]() {
#pragma unroll
      for (int m = 0; m < MTB; ++m) acc[0][m] = f32x4{0, 0, 0, 0};
      ac2[0] = f32x4{0, 0, 0, 0}; ac2[1] = f32x4{0, 0, 0, 0};
    };
    auto job = [&](const float* arow, int w, int wnext) {
      if constexpr (MTB == 1) run_job<1, 1, 8, LDH, 0, 1, 0, true>(arow, wr, w, wnext, lane, P, Q, ac2);
      else run_job<MTB, 1, 8, LDH, 0, 1, 0>(arow, wr, w, wnext, lane, P, Q, acc);
    };
    auto fold = [&]() { if constexpr (MTB == 1) acc[0][0] = ac2[0] + ac2[1]; };
    zero();
    if constexpr (SIDE) {
      const int o_hfa = (S::R + (lane & 15) * LDF + 4 * g) * 4;
      run_job<MTB, 1, 16, LDF, 0, 1, 0>(at(o_hfa, 0), wr, W(S::W0, wave, 16, 0), W(S::W1, wave, 8), lane, P, Q, acc);
      FSEC(2);
      __syncthreads();                    // every wave has read the image: t1 may take its place
      FSEC(3);
    } else {
    job(afh, W(S::W0, wave, 16, 0), W(S::W0, wave, 16, 8));      // cos half
    FSEC(2);
    __syncthreads();
    {
      // all five read-backs in flight before the first is used (left to itself the compiler reuses ONE destination and
      // serialises five L2 round trips here)
      f32x4 sn[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) sn[s] = bload(sr, lane * 16, ((s * NW + wave) * 3 + 0) * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (CMP) { if (mrow[s][i] >= 0) *at(o_hc, mrow[s][i] * LDH) = sn[s][i]; }
          else *at(o_he, (s * 16 + i) * LDH) = sn[s][i];
        }
      }
    }
    __syncthreads();
    FSEC(3);
    job(afh, W(S::W0, wave, 16, 8), W(S::W1, wave, 8));           // sin half (accumulates on the cos half)
    fold();
    }
    {
      const float b = bias(S::B0);
#pragma unroll
      for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) *at(o_he2, (16 + m * 16 + i) * LDH) = fmaxf(acc[0][m][i] + b, 0.f);
    }
    __syncthreads();
    FSEC(4);
    zero();
    job(at(o_ha2, 16 * LDH), W(S::W1, wave, 8), W(S::W4, wave, 8));
    fold();
    {
      const float b = bias(S::B1);
#pragma unroll
      for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) *at(o_he, (m * 16 + i) * LDH) = fmaxf(acc[0][m][i] + b, 0.f);   // st
    }
    __syncthreads();
    FSEC(5);
    const float* ast = afh;
#pragma unroll
    for (int q = 0; q < 3; ++q) {        // gate tiles wave, wave + 8 (D = 256; D = 128: one tile), then the st half of j1
      if (q == 1 && TPW == 1) continue;
      zero();
      float b;
      if (q < 2) {
        const int nxt = (q == 0 && TPW == 2) ? W(S::W4, wave + 8, 8) : W(S::W5, wave, 16, 8);
        job(ast, W(S::W4, wave + 8 * q, 8), nxt);
        b = bias(S::B4 + 128 * q);
      } else {
        job(ast, W(S::W5, wave, 16, 8), W(S::W2, wave, D / 16));
        b = bias(S::B5);
      }
      fold();
#pragma unroll
      for (int m = 0; m < MTB; ++m) bstore(sr, lane * 16, ((m * NW + wave) * 3 + q) * 1024, f32x4{acc[0][m][0] + b, acc[0][m][1] + b, acc[0][m][2] + b, acc[0][m][3] + b});
    }
    FSEC(6);
    __syncthreads();
    FSEC(7);
  }

  // ---- one field evaluation (x branch) at the stage input in X buffer `cur`, time slot `slot` ----------------------
  // Entry: X[cur] visible to the workgroup, P = first group of W2 tile `wave`.  Exit: kv = dx/dt of this lane's
  // elements (row 4g+i, col 16 (wave + 8 q) + c); this wave's divergence partials in DLP[dst]; P = first group of
  // `wnext` (W2: another evaluation follows, W0: a time batch follows).
  // gate (q < TPW) / st contribution to j1 (q = 2) of (stage `slot`, rank `rank`) from a compacted time batch: M-row
  // mm = rps * slot + rank, i.e. element (mm & 3) of the float4 that lane 16 ((mm & 15) >> 2) + c of this wave stored for M tile mm >> 4
  __device__ __forceinline__ float tgather(int slot, int rank, int rps, int q) const {
    const int mm = rps * slot + rank;
    const int vo = (mm >> 4) * (NW * 3 * 1024) + (((mm & 15) >> 2) * 16 + c) * 16 + (mm & 3) * 4;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, vo, (wave * 3 + q) * 1024, 0));
  }

  __device__ __forceinline__ void eval(int slot, int cur, int dst, bool next_is_tbatch, f32x4 (&P)[4], f32x4 (&Q)[4], float (&kv)[TPW][4],
                                       const f32x4 sg, int rps = 0, const f32x4 rk = f32x4{0.f, 0.f, 0.f, 0.f}) {
    const bool compact = rps != 0;
#ifdef MFM_STAMPS
    sec_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int xsel = cur ? S::XB1 * 4 : S::XB0 * 4;
    // stage-time inputs of this lane, written by itself in tbatch
    f32x4 gt[TPW], j1t;
    if (!compact) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) gt[q] = bload(sr, lane * 16, ((slot * NW + wave) * 3 + q) * 1024);
      j1t = bload(sr, lane * 16, ((slot * NW + wave) * 3 + 2) * 1024);
    } else {
      // compact time batch: (stage, row) sits in M-row 3 slot + rank(row) of the single tile, i.e. in element (mm & 3) of the
      // float4 that lane 16 (mm >> 2) + c of this wave stored
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rank = rk[i] >= 0.f ? (int)rk[i] : 0;
#pragma unroll
        for (int q = 0; q < TPW; ++q) gt[q][i] = tgather(slot, rank, rps, q);
        j1t[i] = tgather(slot, rank, rps, 2);
      }
    }
    // grad log pi(x) (clipped), the masked Hessian-vector product and z for this lane's out-layer elements.  Pure VALU
    // + LDS work on inputs known when the evaluation starts: waves 0-3 do it before their x1 job, waves 4-7 after theirs,
    // so on every SIMD one wave's VALU phase runs beside its partner's MFMAs instead of both stalling the matrix pipe
    // at the head of the out job.
    float gc[TPW][4], hz[TPW][4], zz[TPW][4];
    const float icoef = 1.f / coef;
    auto tt_elem = [&](int q, int i) {
      const float* xr = at(o_xo + xsel, i * LDX + 128 * q);
      const float* zr = at(o_xo, S::ZB + i * LDX + 128 * q);
      const float x = xr[0], z = zr[0];
      const float graw = -tbeta * (coef * (2.f * x - xr[-1] - xr[1]) - x * (1.f - x * x) * icoef);
      const float hv = -tbeta * (coef * (2.f * z - zr[-1] - zr[1]) - (1.f - 3.f * x * x) * z * icoef);
      gc[q][i] = clip > 0.f ? fminf(fmaxf(graw, -clip), clip) : graw;
      hz[q][i] = (!(clip > 0.f) || fabsf(graw) <= clip) ? hv : 0.f;
      zz[q][i] = z;
    };
    auto target_terms = [&]() {
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tt_elem(q, i);
      }
    };
#ifndef MFM_TT_MODE
#define MFM_TT_MODE 3        // 0: waves 0-3 before their x1 job, waves 4-7 after theirs (stagger); 1: every wave before; 2: every wave after
                             // (same-trajectory A/B, tools/flow_ab.py, round 2: 52.75 / 52.95 / 53.03 ms: none of them overlaps anything --
                             // a wave issues in order, so its vector work only runs in the shadow of ITS OWN queued MFMAs);
                             // 3 (round 4): every wave, INSIDE its x1 job: two elements per fragment group, the scheduler told to put
                             // vector and LDS instructions between the group's sixteen MFMAs (sched_group_barrier)
#endif
    if (MFM_TT_MODE == 1 || (MFM_TT_MODE == 0 && wave < NW / 2)) target_terms();
    {   // x1: value rows; tangent rows = relu' * (z W_x1)
      f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      if constexpr (MFM_TT_MODE == 3) {
        // One PAIR of elements (q, i0), (q, i0 + 1) per fragment group of sixteen MFMAs; its six pieces sit between the group's MFMAs,
        // fenced so they stay there, and the operands of the next pair are fetched one group ahead.  The arithmetic is spelled with
        // explicit fused multiply-adds in the form the compiler chose for target_terms(): the two are bit-identical.
        constexpr int G = D / 64;                 // fragment groups of four k-blocks = pairs of elements (4 TPW / 2)
        const float* arow = at(o_xa + xsel, 0);
        const int w0 = W(S::W2, wave, D / 16), wnx = W(S::W3, wave, 8);
        const float ntb = -tbeta;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 ox[2][6];                           // [buffer][x, x-, x+, z, z-, z+] of the pair's two elements
        auto tt_fetch = [&](int p, f32x2 (&o)[6]) {
          const int q = p >> 1, i0 = 2 * (p & 1);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float* xr = at(o_xo + xsel, (i0 + e) * LDX + 128 * q);
            const float* zr = at(o_xo, S::ZB + (i0 + e) * LDX + 128 * q);
            o[0][e] = xr[0]; o[1][e] = xr[-1]; o[2][e] = xr[1]; o[3][e] = zr[0]; o[4][e] = zr[-1]; o[5][e] = zr[1];
          }
        };
        // two elements per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): beside MFMAs every vector instruction costs
        // the SIMD its issue cycles whether or not the matrix pipe is busy (measured: the scalar form of these pieces, 240
        // instructions, took as long inside the job as the packed 110 had taken in front of it)
        f32x2 lx, lz, ax, bz, gr, hv;
        const f32x2 two = {2.f, 2.f}, one = {1.f, 1.f}, three = {3.f, 3.f}, ic2 = {icoef, icoef}, co2 = {coef, coef}, nt2 = {ntb, ntb};
        auto piece = [&](int p, int k, const f32x2 (&o)[6]) {
          const int q = p >> 1, i0 = 2 * (p & 1);
          const f32x2 x = o[0], z = o[3];
          if (k == 0) { lx = __builtin_elementwise_fma(x, two, -o[1]) - o[2]; lz = __builtin_elementwise_fma(z, two, -o[4]) - o[5]; }
          if (k == 1) ax = ic2 * (x * __builtin_elementwise_fma(-x, x, one));
          if (k == 2) { gr = __builtin_elementwise_fma(co2, lx, -ax) * nt2; bz = ic2 * (z * __builtin_elementwise_fma(-x, x * three, one)); }
          if (k == 3) hv = __builtin_elementwise_fma(co2, lz, -bz) * nt2;
          if (k == 4) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              gc[q][i0 + e] = clip > 0.f ? fminf(fmaxf(gr[e], -clip), clip) : gr[e];
              hz[q][i0 + e] = (!(clip > 0.f) || fabsf(gr[e]) <= clip) ? hv[e] : 0.f;
              zz[q][i0 + e] = z[e];
            }
          }
          // pin the piece where it stands: pure arithmetic is otherwise sunk to its first use (the last piece) before the
          // instruction scheduler ever sees it, whatever the scheduling fences say
          if (k == 0) asm volatile("" : "+v"(lx), "+v"(lz));
          if (k == 1) asm volatile("" : "+v"(ax));
          if (k == 2) asm volatile("" : "+v"(gr), "+v"(bz));
          if (k == 3) asm volatile("" : "+v"(hv));
          if (k == 4) asm volatile("" : "+v"(gc[q][i0]), "+v"(gc[q][i0 + 1]), "+v"(hz[q][i0]), "+v"(hz[q][i0 + 1]));
        };
        tt_fetch(0, ox[0]);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          f32x4 (&B)[4] = (gi & 1) ? Q : P;
          f32x4 (&N)[4] = (gi & 1) ? P : Q;
          if (gi + 1 < G) load_group<1, 0>(N, wr, w0 + (gi + 1) * 4 * 1024, lane);
          else load_group<1, 0>(N, wr, wnx, lane);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 an = *reinterpret_cast<const f32x4*>(arow + gi * 64);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 a = an;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              acc[u & 1] = mfma4(a[s4], B[u][s4], acc[u & 1]);
              const int slot = u * 4 + s4;                              // 0..15 inside the group
              __builtin_amdgcn_sched_barrier(0);
              if (s4 == 0 && u < 3) { an = *reinterpret_cast<const f32x4*>(arow + gi * 64 + (u + 1) * 16); asm volatile("" ::: "memory"); }   // A fragment one k-block ahead
              if (slot == 1 && gi + 1 < G) { tt_fetch(gi + 1, ox[(gi + 1) & 1]); asm volatile("" ::: "memory"); }      // issued here, awaited a group later
              if (slot == 3) piece(gi, 0, ox[gi & 1]);
              if (slot == 5) piece(gi, 1, ox[gi & 1]);
              if (slot == 7) piece(gi, 2, ox[gi & 1]);
              if (slot == 9) piece(gi, 3, ox[gi & 1]);
              if (slot == 12) piece(gi, 4, ox[gi & 1]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if constexpr (G & 1) {                    // (never: G = 2 or 4) the job must end with its next group in P
#pragma unroll
          for (int j = 0; j < 4; ++j) P[j] = Q[j];
        }
      } else {
        run_job<1, 1, D / 16, LDX, 0, 1, 0, true>(at(o_xa + xsel, 0), wr, W(S::W2, wave, D / 16), W(S::W3, wave, 8), lane, P, Q, acc);
      }
      FSEC(8);
      if (MFM_TT_MODE == 2 || (MFM_TT_MODE == 0 && wave >= NW / 2)) target_terms();
      const float b = bias(S::B2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = (acc[0][i] + acc[1][i]) + b;
        *at(o_he, i * LDH) = fmaxf(pre, 0.f);
        *at(o_he, (16 + i) * LDH) = pre > 0.f ? tz1[i] : 0.f;
      }
    }
    FSEC(13);
    __syncthreads();
    FSEC(18);
    {   // x2
      f32x4 acc[1][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}};
      run_job<2, 1, 8, LDH, 0, 1, 0>(at(o_ha, 0), wr, W(S::W3, wave, 8), W(S::W5, wave, 16, 0), lane, P, Q, acc);
      FSEC(9);
      const float b = bias(S::B3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = acc[0][0][i] + b;
        *at(o_he, (32 + i) * LDH) = fmaxf(pre, 0.f);
        *at(o_he, (48 + i) * LDH) = pre > 0.f ? acc[0][1][i] : 0.f;
      }
    }
    FSEC(14);
    __syncthreads();
    FSEC(18);
    {   // j1: sx half of the concatenated input (:83); the st half + bias arrive as the initial accumulator
      f32x4 acc[1][2] = {{j1t, {0, 0, 0, 0}}};
      run_job<2, 1, 8, LDH, 0, 1, 0>(at(o_ha, 32 * LDH), wr, W(S::W5, wave, 16, 0), W(S::W6, wave, 8), lane, P, Q, acc);
      FSEC(10);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = acc[0][0][i];
        *at(o_he2, i * LDH) = fmaxf(pre, 0.f);
        *at(o_he2, (16 + i) * LDH) = pre > 0.f ? acc[0][1][i] : 0.f;
      }
    }
    FSEC(15);
    __syncthreads();
    FSEC(18);
    float dp[4];        // this lane's share of z . (J z) per row: the j2-tangent part here, the gate part in the out epilogue
    {   // j2
      f32x4 acc[1][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}};
      run_job<2, 1, 8, LDH, 0, TPW, OUT_T1OFF>(at(o_ha2, 0), wr, W(S::W6, wave, 8), W(S::W7, wave, 8), lane, P, Q, acc);
      FSEC(11);
      const float b = bias(S::B6);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = acc[0][0][i] + b;
        *at(o_he2, (32 + i) * LDH) = fmaxf(pre, 0.f);
        // z . (W_out^T tj2) = (W_out z) . tj2: the tangent rows stop here (see the header)
        dp[i] = (pre > 0.f ? acc[0][1][i] : 0.f) * w7z[i];
      }
    }
    FSEC(16);
    __syncthreads();
    FSEC(18);
    {   // out: v = nn_xt + nn_t * clip(grad log pi(x)) (:88-90);  z . J z = (W_out z) . tj2 + z . (nn_t * 1[|g| <= clip] * (H z))
      float bo[TPW];
      f32x4 acc[TPW][1];
#pragma unroll
      for (int q = 0; q < TPW; ++q) { acc[q][0] = f32x4{0, 0, 0, 0}; bo[q] = bias(S::B7 + 128 * q); }
      const int wnext = next_is_tbatch ? W(S::W0, wave, 16) : W(S::W2, wave, D / 16);
      run_job<1, TPW, 8, LDH, OUT_T1OFF, 1, 0>(at(o_ha2, 32 * LDH), wr, W(S::W7, wave, 8), wnext, lane, P, Q, acc);
      FSEC(12);
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const float b = bo[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = acc[q][0][i] + b + gt[q][i] * gc[q][i];
          dp[i] += zz[q][i] * (gt[q][i] * hz[q][i]);
          kv[q][i] = sg[i] > 0.f ? v : -v;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) dp[i] = sg[i] > 0.f ? -dp[i] : dp[i];          // :218 / :239
      part_put(S::DLP + dst * 128, dp);
      FSEC(17);
    }
  }

  // lanes 32..63 receive the value of lane - 32 (v_permlane32_swap: lanes [32:63] of vdst <-> lanes [0:31] of src)
  __device__ __forceinline__ float from_lower_half(float v) const {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, r[0]);
  }

  // ---- COMPACT field evaluation: at most 8 rows of the tile still integrate ----------------------------------------
  // The full evaluation pushes 16 value rows and 16 tangent rows (two M tiles) through the x branch whatever the number of
  // rows still integrating.  Here the value rows of the <= 8 active chains (by rank) and their tangent rows share ONE M
  // tile: M-rows 0..7 = values of ranks 0..7, M-rows 8..15 = tangents of ranks 0..7 -- half the matrix work per evaluation
  // for the attempts of a tile's tail, which is what the launch waits for.  In the accumulator, lane (g, c) holds M-rows
  // 4g..4g+3: lanes 0..31 values, lanes 32..63 the tangents of the same (rank, column), so the activation mask crosses the
  // wave halves with one v_permlane32_swap per element.  Inputs: X[cur] rows 0..7 = stage inputs by rank, rows 8..15 = the
  // probes by rank (written by the owners: solve2).  The first x layer recomputes z W_x1 as its rows 8..15.  Results go
  // back to the lanes that own the chain rows (Runge-Kutta registers) through rows 0..7 of the OTHER X buffer.
  __device__ __forceinline__ void eval_c(int slot, int cur, int dst, bool next_is_tbatch, f32x4 (&P)[4], f32x4 (&Q)[4], float (&kv)[TPW][4],
                                         const f32x4 sg, int rps, const f32x4 rk) {
#ifdef MFM_STAMPS
    sec_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int xsel = cur ? S::XB1 * 4 : S::XB0 * 4, xoth = cur ? S::XB0 * 4 : S::XB1 * 4;
    const bool is_t = g >= 2;
    const int jr0 = 4 * (g & 1);                       // this lane's M-rows 4g + i carry rank jr0 + i
    f32x4 gt[TPW], j1t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) gt[q][i] = tgather(slot, jr0 + i, rps, q);
      const float jt = tgather(slot, jr0 + i, rps, 2);
      j1t[i] = is_t ? 0.f : jt;                        // the st half of j1's input has no tangent
    }
    float gc[TPW][4], hz[TPW][4], zz[TPW][4];
    auto target_terms = [&]() {
      const float icoef = 1.f / coef;
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* xr = at(o_xc + xsel, (jr0 + i) * LDX + 128 * q);
          const float* zr = xr + 8 * LDX;
          const float x = xr[0], z = zr[0];
          const float graw = -tbeta * (coef * (2.f * x - xr[-1] - xr[1]) - x * (1.f - x * x) * icoef);
          const float hv = -tbeta * (coef * (2.f * z - zr[-1] - zr[1]) - (1.f - 3.f * x * x) * z * icoef);
          gc[q][i] = clip > 0.f ? fminf(fmaxf(graw, -clip), clip) : graw;
          hz[q][i] = (!(clip > 0.f) || fabsf(graw) <= clip) ? hv : 0.f;
          zz[q][i] = z;
        }
      }
    };
    // one layer's epilogue: value lanes activate, tangent lanes take the mask from their partner's pre-activation
    auto act_store = [&](const f32x4& pre, float b, float* dstp) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pv = pre[i] + b;
        const float pp = from_lower_half(pv);
        dstp[i * LDH] = is_t ? (pp > 0.f ? pre[i] : 0.f) : fmaxf(pv, 0.f);
      }
    };
    CSEC(0);
    if (wave < NW / 2) target_terms();
    CSEC(1);
    {   // x1 on [values ; probes]
      f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      run_job<1, 1, D / 16, LDX, 0, 1, 0, true>(at(o_xa + xsel, 0), wr, W(S::W2, wave, D / 16), W(S::W3, wave, 8), lane, P, Q, acc);
      CSEC(2);
      if (wave >= NW / 2) target_terms();
      act_store(acc[0] + acc[1], bias(S::B2), at(o_he, 0));
    }
    CSEC(3);
    __syncthreads();
    CSEC(4);
    {   // x2
      f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      run_job<1, 1, 8, LDH, 0, 1, 0, true>(at(o_ha, 0), wr, W(S::W3, wave, 8), W(S::W5, wave, 16, 0), lane, P, Q, acc);
      CSEC(5);
      act_store(acc[0] + acc[1], bias(S::B3), at(o_he, 32 * LDH));
    }
    CSEC(6);
    __syncthreads();
    CSEC(4);
    {   // j1: the st half + bias arrive as the initial accumulator of the value rows
      f32x4 acc[2] = {j1t, {0, 0, 0, 0}};
      run_job<1, 1, 8, LDH, 0, 1, 0, true>(at(o_ha, 32 * LDH), wr, W(S::W5, wave, 16, 0), W(S::W6, wave, 8), lane, P, Q, acc);
      CSEC(7);
      act_store(acc[0] + acc[1], 0.f, at(o_he2, 0));
    }
    CSEC(6);
    __syncthreads();
    CSEC(4);
    {   // j2
      f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      run_job<1, 1, 8, LDH, 0, TPW, OUT_T1OFF, true>(at(o_ha2, 0), wr, W(S::W6, wave, 8), W(S::W7, wave, 8), lane, P, Q, acc);
      CSEC(8);
      act_store(acc[0] + acc[1], bias(S::B6), at(o_he2, 32 * LDH));
    }
    CSEC(6);
    __syncthreads();
    CSEC(4);
    {   // out
      float bo[TPW];
      f32x4 acc[TPW][1];
#pragma unroll
      for (int q = 0; q < TPW; ++q) { acc[q][0] = f32x4{0, 0, 0, 0}; bo[q] = bias(S::B7 + 128 * q); }
      const int wnext = next_is_tbatch ? W(S::W0, wave, 16) : W(S::W2, wave, D / 16);
      run_job<1, TPW, 8, LDH, OUT_T1OFF, 1, 0>(at(o_ha2, 32 * LDH), wr, W(S::W7, wave, 8), wnext, lane, P, Q, acc);
      float dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (is_t) dp[i] += zz[q][i] * (acc[q][0][i] + gt[q][i] * hz[q][i]);                   // z . J z of rank jr0 + i
          else *at(o_xc + xoth, (jr0 + i) * LDX + 128 * q) = acc[q][0][i] + bo[q] + gt[q][i] * gc[q][i];   // v of rank jr0 + i
        }
      }
      CSEC(9);
      part_put(S::DLP + dst * 128, dp);            // M-row 8 + rank; direction sign applied by the row leaders
    }
    // no workgroup barrier: an owner lane reads what a lane of its OWN wave (same column tile, another row group) stored just
    // above, and a wave's LDS instructions complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = rk[i] >= 0.f ? *at(o_xc + xoth, (int)rk[i] * LDX + 128 * q) : 0.f;
        kv[q][i] = sg[i] > 0.f ? v : -v;
      }
    CSEC(10);
  }

  // sum over the four 16-lane groups; every lane receives the same (bit-identical: float addition commutes) total
  __device__ __forceinline__ f32x4 gsum(f32x4 v) const {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = v[r];
      x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));      // lane ^ 16
      // lane ^ 32: v_permlane32_swap exchanges a[32:63] with b[0:31] in place in BOTH registers.  Through inline assembly: the
      // builtin's second result is miscompiled by hipcc 7.2 (p[0] + p[1] came out as v_pk_add v, v, v of the FIRST result; found
      // with tools/mb/m4job.hip), and inline assembly gets no hazard padding from the compiler, hence the s_nop
      float xa = x, xb = x;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(xa), "+v"(xb));
      v[r] = xa + xb;
    }
    return v;
  }

};

// Integrate the augmented ODE from t = 0 to 1 (see ode_solve in ode.hip: same state machine, same controller).
// Requires: Z filled (probe), halo pads of X0 / X1 / Z zero, biases in LDS.
// RP: the parity-instrumentation instance (Replay, ode.hip); the production instance (RP = false) carries none of it.
template <int D, bool RP>
__device__ __forceinline__ void solve(FTile<D>& T, float rtol, float atol, int max_attempts, float (&y)[FTile<D>::TPW][4],
                                      float (&ell)[4], int (&natt)[4], const Replay& rp, int rp_row0, int d_true = D) {
  using S = FS<D>;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX;
  const int g = T.g, c = T.c, wave = T.wave;
  float* lds = T.lds;
  const float inv_n = 1.f / (float)(d_true + 1);      // (d_true < D: a network zero-padded to this instance's tile width, see launch_*)
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
    for (int fld = 0; fld < 16; ++fld) T.rs_put(fld, z4);
  }
  f32x4 P[4], Q[4];
  load_group<1, 0>(P, T.wtr, T.W(S::W7, wave, D / 16), T.lane);
  __syncthreads();
  T.precompute_w7z(P, Q);
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
      const int xsel = cur ? S::XB1 * 4 : S::XB0 * 4;
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j) acc += cf[j] * k[j][q][i];
          *T.at(T.o_xo + xsel, i * LDX + 128 * q) = y[q][i] + hs[i] * acc;
        }
      }
    }
#ifdef MFM_STAMPS
    const unsigned long long c0_ = __builtin_amdgcn_s_memtime();
#endif
    if (phase <= 2) { float cvv[5][4], svv[5][4]; T.template tb_trig<false>(phase, cvv, svv); T.tbatch(phase, P, Q, cvv, svv); } else __syncthreads();
#ifdef MFM_STAMPS
    const unsigned long long c1_ = __builtin_amdgcn_s_memtime();
    if (phase <= 2) { T.cyc_tb += c1_ - c0_; T.n_tb += 1; }
#endif
    float kv[TPW][4];
    const int dst = phase == 0 ? 0 : phase - 1 + (phase == 1 ? 1 : 0);
    T.eval(phase < 2 ? 0 : (phase == 7 ? 4 : phase - 2), cur, dst, phase == 7 || phase < 2, P, Q, kv,
           f32x4{(float)T.sign, (float)T.sign, (float)T.sign, (float)T.sign});
#ifdef MFM_STAMPS
    T.cyc_eval += __builtin_amdgcn_s_memtime() - c1_; T.n_eval += 1;
#endif
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
      T.part_put(S::RED + 0 * 128, p0); T.part_put(S::RED + 1 * 128, p1);
      __syncthreads();
      if (wave == 0 && T.lane < 16) {          // row leaders (see the end-of-step block below)
        auto sum8 = [&](int base) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(T.at(T.o_l8, base)), b = *reinterpret_cast<const f32x4*>(T.at(T.o_l8, base + 4));
          return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
        };
        auto R1 = [&](int field) -> float& { return *T.at(T.o_l1, S::RS + field * 16); };
        const float dl0 = sum8(S::DLP + 0 * 128);
        const float a1 = dl0 / atol;                                   // ell0 = 0 -> scale = atol
        const float d0 = sqrtf(sum8(S::RED + 0 * 128)), d1 = sqrtf(sum8(S::RED + 1 * 128) + a1 * a1);
        R1(RS_H0) = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
        R1(RS_D1) = d1; R1(RS_KL + 0) = dl0;
      }
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
      T.part_put(S::RED + 2 * 128, p2);
      __syncthreads();
      int any = 0;
      if (wave == 0 && T.lane < 16) {
        auto sum8 = [&](int base) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(T.at(T.o_l8, base)), b = *reinterpret_cast<const f32x4*>(T.at(T.o_l8, base + 4));
          return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
        };
        auto R1 = [&](int field) -> float& { return *T.at(T.o_l1, S::RS + field * 16); };
        const float h0 = R1(RS_H0), d1 = R1(RS_D1);
        const float a2 = (sum8(S::DLP + 1 * 128) - R1(RS_KL + 0)) / atol;
        const float d2 = sqrtf(sum8(S::RED + 2 * 128) + a2 * a2) / h0;
        const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
        float dt = fminf(100.f * h0, h1);
        if constexpr (RP) { const size_t o = rp.at(0, rp_row0 + T.lane, 0); rp.dt_own[o] = dt; dt = rp.dt[o]; }
        R1(RS_DT) = dt;
        any = dt > 0.f ? 1 : 0;
      }
      phase = 2;
      if (!__syncthreads_or(any)) break;
    } else if (phase < 7) {
      phase += 1;
    } else {
      // ---- end of an attempted step ----
      // every lane: this wave's share of the squared error norm of its rows
      {
        float e2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float acc = 0.f, er = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) acc += DP_TAB[7][j] * k[j][q][i];
            const float y1 = y[q][i] + hs[i] * acc;               // the stage-7 input (5th-order solution), same arithmetic
#pragma unroll
            for (int j = 0; j < 7; ++j) er += DP_E[j] * k[j][q][i];
            er *= hs[i];
            const float tol = atol + rtol * fmaxf(fabsf(y[q][i]), fabsf(y1));
            const float rr = er / tol;
            e2[i] += rr * rr;
          }
        T.part_put(S::RED + 3 * 128, e2);
      }
      __syncthreads();
      // ONE lane per row (the row leaders: lanes 0..15 of wave 0) totals the partials and runs the step-size controller;
      // it publishes the new row state, a decision flag (0 keep, 1 advance, 2 finish) and the interpolation abscissa
      int any = 0;
      if (wave == 0 && T.lane < 16) {
        auto sum8 = [&](int base) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(T.at(T.o_l8, base)), b = *reinterpret_cast<const f32x4*>(T.at(T.o_l8, base + 4));
          return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
        };
        auto R1 = [&](int field) -> float& { return *T.at(T.o_l1, S::RS + field * 16); };
        float kl[7];
        kl[0] = R1(RS_KL + 0);
#pragma unroll
        for (int j = 1; j < 7; ++j) kl[j] = sum8(S::DLP + j * 128);
        const float e2 = sum8(S::RED + 3 * 128);
        const float t0 = R1(RS_T), dti = R1(RS_DT), ell0 = R1(RS_ELL), na = R1(RS_NATT), dn = R1(RS_DONE);
        const bool active = !(dn != 0.f) && na < (float)max_attempts && dti > 0.f;
        float sl = 0.f, el = 0.f, lm = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) sl += DP_TAB[7][j] * kl[j];
#pragma unroll
        for (int j = 0; j < 7; ++j) el += DP_E[j] * kl[j];
#pragma unroll
        for (int j = 0; j < 7; ++j) lm += DP_M[j] * kl[j];
        const float l1 = ell0 + dti * sl;
        el *= dti;
        const float tol = atol + rtol * fmaxf(fabsf(ell0), fabsf(l1));
        const float rr = el / tol;
        const float ratio = sqrtf((e2 + rr * rr) * inv_n);
        bool acc = active && ratio <= 1.f;
        const float dfac = ratio < 1.f ? 1.f : 0.2f;
#ifdef MFM_LIB_POW
        const float fac = fminf(10.f, fmaxf(0.9f * powf(ratio, -0.2f), dfac));
#else
    // ratio^(-1/5) through the hardware's log2 / exp2 (1 ulp each; ratio is a non-negative finite number or NaN here): the library's
    // powf spends ~150 dependent instructions on cases this call cannot meet, on ONE wave while the other seven wait at the barrier
        const float fac = fminf(10.f, fmaxf(0.9f * __builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(ratio)), dfac));
#endif
        float ndt = fmaxf(ratio == 0.f ? dti * 10.f : dti * fac, 0.f);
        if constexpr (RP) {
          if (active) {
            const int j = (int)na;
            const bool in = j < rp.cap, nx = j + 1 < rp.cap;
            const size_t o = rp.at(0, rp_row0 + T.lane, in ? j : 0);
            if (in) { rp.ratio[o] = ratio; if (nx) rp.dt_own[o + 1] = ndt; }
            acc = in && rp.acc[o] != 0;
            ndt = nx ? rp.dt[o + 1] : 0.f;
          }
        }
        const float tn = t0 + dti;
        const bool fin = acc && tn >= 1.f, adv = acc && !(tn >= 1.f);
        const float sfrac = (1.f - t0) / (tn - t0);
        // log-det: 4th-order interpolant of this step evaluated at t = 1 when the step reaches the end
        const float y0 = ell0, ym = y0 + dti * lm, f0 = dti * kl[0], f1 = dti * kl[6];
        const float pa = -2.f * f0 + 2.f * f1 - 8.f * y0 - 8.f * l1 + 16.f * ym;
        const float pb = 5.f * f0 - 3.f * f1 + 18.f * y0 + 14.f * l1 - 32.f * ym;
        const float pc = -4.f * f0 + f1 - 11.f * y0 - 5.f * l1 + 16.f * ym;
        const float li = (((pa * sfrac + pb) * sfrac + pc) * sfrac + f0) * sfrac + y0;
        const float dt_n = active ? ndt : dti, na_n = active ? na + 1.f : na, dn_n = fin ? 1.f : dn;
        R1(RS_T) = acc ? tn : t0; R1(RS_DT) = dt_n; R1(RS_ELL) = fin ? li : (adv ? l1 : ell0);
        R1(RS_KL + 0) = adv ? kl[6] : kl[0]; R1(RS_NATT) = na_n; R1(RS_DONE) = dn_n;
        R1(RS_FLAG) = fin ? 2.f : (adv ? 1.f : 0.f); R1(RS_SFRAC) = sfrac;
        any = (dn_n == 0.f && na_n < (float)max_attempts && dt_n > 0.f) ? 1 : 0;
      }
      const int go = __syncthreads_or(any);
      // every lane: apply the decision to its elements (branch-free selects)
      {
        const f32x4 fl4 = T.rs_get(RS_FLAG), sf4 = T.rs_get(RS_SFRAC);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dti = hs[i], sfrac = sf4[i];
          const bool fin = fl4[i] == 2.f, adv = fl4[i] == 1.f;
#pragma unroll
          for (int q = 0; q < TPW; ++q) {
            float acc = 0.f, km = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) acc += DP_TAB[7][j] * k[j][q][i];
#pragma unroll
            for (int j = 0; j < 7; ++j) km += DP_M[j] * k[j][q][i];
            const float x0 = y[q][i], x1 = x0 + dti * acc, xm = x0 + dti * km, g0 = dti * k[0][q][i], g1 = dti * k[6][q][i];
            const float qa = -2.f * g0 + 2.f * g1 - 8.f * x0 - 8.f * x1 + 16.f * xm;
            const float qb = 5.f * g0 - 3.f * g1 + 18.f * x0 + 14.f * x1 - 32.f * xm;
            const float qc = -4.f * g0 + g1 - 11.f * x0 - 5.f * x1 + 16.f * xm;
            const float xi = (((qa * sfrac + qb) * sfrac + qc) * sfrac + g0) * sfrac + x0;
            y[q][i] = fin ? xi : (adv ? x1 : x0);
            k[0][q][i] = adv ? k[6][q][i] : k[0][q][i];
          }
        }
      }
      if (!go) break;
      // the next attempt's stage input is built at the top of the loop from the row state the leaders wrote: make sure
      // every lane has passed the reads above before a leader can overwrite the flags again (next barrier is in tbatch)
      phase = 2;
    }
  }
  {
    const f32x4 ell4 = T.rs_get(RS_ELL), na4 = T.rs_get(RS_NATT);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ell[i] = ell4[i]; natt[i] = (int)na4[i]; }
  }
}

// ---- end of an attempt, row leaders (lanes 0..15 of wave 0, one per row): totals of the partial sums, initial-step heuristic /
// step-size controller / solve switch of the row, in its mode; publishes the new row state, a decision flag (0 keep, 1 advance,
// 2 finish, 3 initial slope) and the interpolation abscissa.  Shared by the tile's main loop (solve2) and its tail (solve2_tail).
// STICKY: the rows keep the slots they were given when the tail began (RS_RANK = slot or -1); otherwise ranks and the tile's
// evaluation mode are recomputed from the rows that take part in the next attempt.  Returns "this row takes part in it".
__device__ __forceinline__ int true_dim(const OdeArgs& a) { return a.net.d; }
__device__ __forceinline__ int true_dim(const struct TailArgs& a);
template <int D, bool RP, bool STICKY, bool PAD, typename AA, typename FA>
__device__ __forceinline__ int leaders_end_of_attempt(FTile<D>& T, const AA& a, const FA& f, int b0, int cmode) {
  using S = FS<D>;
  const float rtol = a.rtol, atol = a.atol;
  const int max_attempts = a.max_attempts;
  const float inv_n = 1.f / (float)((PAD ? true_dim(a) : D) + 1);
  int any = 0;
  // (the tail derives its lane coordinates locally: see solve2_tail)
  int o_l8 = T.o_l8, o_l1 = T.o_l1, ln = T.lane;
  if constexpr (STICKY) { int t_ = threadIdx.x; asm volatile("" : "+v"(t_)); ln = t_ & 15; o_l8 = ln * 32; o_l1 = ln * 4; }
      auto sum8 = [&](int base) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(T.at(o_l8, base)), v = *reinterpret_cast<const f32x4*>(T.at(o_l8, base + 4));
    return ((u[0] + u[1]) + (u[2] + u[3])) + ((v[0] + v[1]) + (v[2] + v[3]));
  };
  auto R1 = [&](int field) -> float& { return *T.at(o_l1, S::RS + field * 16); };
  const float mode = R1(RS_MODE);
  float flag = 0.f, sw = 0.f;
  // divergence partials of this row: full evaluation: M-row = row, direction already applied; compact evaluation: M-row
  // 8 + rank(row), raw
  const int dl_off = cmode == 0 ? o_l8 : (8 + (int)R1(RS_RANK)) * 32;
  const float dl_sg = cmode == 0 ? 1.f : (R1(RS_SIGN) > 0.f ? -1.f : 1.f);
  auto dlsum = [&](int base) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(T.at(dl_off, base)), v = *reinterpret_cast<const f32x4*>(T.at(dl_off, base + 4));
    return dl_sg * (((u[0] + u[1]) + (u[2] + u[3])) + ((v[0] + v[1]) + (v[2] + v[3])));
  };
  if (mode == (float)RM_INIT0) {
    // f0 sits in k[1] (slot 0 of the attempt): initial step size, part 1
    if (R1(RS_SOLVE) != 0.f && (f.mode & 0xFF) == MFM_FLOW_IMH)        // ref.logprob(u0) - ref.logprob(up)  (:254-255); (bits 8..: flow_live_rows)
      R1(RS_LQ) = -0.5f * (sum8(S::RED + 4 * 128) - sum8(S::RED + 5 * 128)) / (f.ref_std * f.ref_std);
    const float dl0 = dlsum(S::DLP + 1 * 128);
    const float a1 = dl0 / atol;
    const float d0 = sqrtf(sum8(S::RED + 0 * 128)), d1 = sqrtf(sum8(S::RED + 1 * 128) + a1 * a1);
    const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    R1(RS_H0) = h0; R1(RS_D1) = d1; R1(RS_KL + 0) = dl0;
    R1(RS_DT) = h0; R1(RS_MODE) = (float)RM_INIT1;
    flag = 3.f;
  } else if (mode == (float)RM_INIT1) {
    const float h0 = R1(RS_H0), d1 = R1(RS_D1);
    const float a2 = (dlsum(S::DLP + 1 * 128) - R1(RS_KL + 0)) / atol;
    const float d2 = sqrtf(sum8(S::RED + 2 * 128) + a2 * a2) / h0;
    const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
    float dt = fminf(100.f * h0, h1);
    if constexpr (RP) { const size_t o = a.rp.at((int)R1(RS_SOLVE), b0 + ln, 0); a.rp.dt_own[o] = dt; dt = a.rp.dt[o]; }
    R1(RS_DT) = dt; R1(RS_MODE) = (float)RM_ATT;
    flag = 0.f;
  } else if (mode == (float)RM_ATT) {
    float kl[7];
    kl[0] = R1(RS_KL + 0);
#pragma unroll
    for (int j = 1; j < 7; ++j) kl[j] = dlsum(S::DLP + j * 128);
    const float e2 = sum8(S::RED + 3 * 128);
    const float t0 = R1(RS_T), dti = R1(RS_DT), ell0 = R1(RS_ELL), na = R1(RS_NATT);
    const bool active = na < (float)max_attempts && dti > 0.f;
    float sl = 0.f, el = 0.f, lm = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) sl += DP_TAB[7][j] * kl[j];
#pragma unroll
    for (int j = 0; j < 7; ++j) el += DP_E[j] * kl[j];
#pragma unroll
    for (int j = 0; j < 7; ++j) lm += DP_M[j] * kl[j];
    const float l1 = ell0 + dti * sl;
    el *= dti;
    const float tol = atol + rtol * fmaxf(fabsf(ell0), fabsf(l1));
    const float rr = el / tol;
    const float ratio = sqrtf((e2 + rr * rr) * inv_n);
    bool acc = active && ratio <= 1.f;
    const float dfac = ratio < 1.f ? 1.f : 0.2f;
#ifdef MFM_LIB_POW
    const float fac = fminf(10.f, fmaxf(0.9f * powf(ratio, -0.2f), dfac));
#else
    const float fac = fminf(10.f, fmaxf(0.9f * __builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(ratio)), dfac));      // (see solve)
#endif
    float ndt = fmaxf(ratio == 0.f ? dti * 10.f : dti * fac, 0.f);
    if constexpr (RP) {
      if (active) {
        const int j = (int)na;
        const bool in = j < a.rp.cap, nx = j + 1 < a.rp.cap;
        const size_t o = a.rp.at((int)R1(RS_SOLVE), b0 + ln, in ? j : 0);
        if (in) { a.rp.ratio[o] = ratio; if (nx) a.rp.dt_own[o + 1] = ndt; }
        acc = in && a.rp.acc[o] != 0;
        ndt = nx ? a.rp.dt[o + 1] : 0.f;
      }
    }
    const float tn = t0 + dti;
    const bool fin = acc && tn >= 1.f, adv = acc && !(tn >= 1.f);
    const float sfrac = (1.f - t0) / (tn - t0);
    const float y0 = ell0, ym = y0 + dti * lm, f0 = dti * kl[0], f1 = dti * kl[6];
    const float pa = -2.f * f0 + 2.f * f1 - 8.f * y0 - 8.f * l1 + 16.f * ym;
    const float pb = 5.f * f0 - 3.f * f1 + 18.f * y0 + 14.f * l1 - 32.f * ym;
    const float pc = -4.f * f0 + f1 - 11.f * y0 - 5.f * l1 + 16.f * ym;
    const float li = (((pa * sfrac + pb) * sfrac + pc) * sfrac + f0) * sfrac + y0;
    const float dt_n = active ? ndt : dti, na_n = active ? na + 1.f : na;
    const float ell_n = fin ? li : (adv ? l1 : ell0);
    // the solve ends when t reaches 1, or (as in odeint's while_loop) when the step budget / step size runs out
    const bool over = fin || !(na_n < (float)max_attempts && dt_n > 0.f);
    flag = fin ? 2.f : (adv ? 1.f : 0.f);
    R1(RS_SFRAC) = sfrac;
    if (!over) {
      R1(RS_T) = acc ? tn : t0; R1(RS_DT) = dt_n; R1(RS_ELL) = ell_n; R1(RS_KL + 0) = adv ? kl[6] : kl[0]; R1(RS_NATT) = na_n;
    } else if (R1(RS_SOLVE) == 0.f) {
      // inverse solve done: keep vol0, start this row's forward solve (:268-269 / :249-250)
      R1(RS_VOL0) = ell_n; R1(RS_NTOT) = na_n;
      R1(RS_T) = 0.f; R1(RS_DT) = 0.f; R1(RS_ELL) = 0.f; R1(RS_KL + 0) = 0.f; R1(RS_NATT) = 0.f;
      R1(RS_SOLVE) = 1.f; R1(RS_SIGN) = 1.f; R1(RS_MODE) = (float)RM_INIT0;
      sw = 1.f;
      *T.at(0, S::RS + RS_TILE * 16) = 1.f;
    } else {
      R1(RS_ELL) = ell_n; R1(RS_NATT) = na_n; R1(RS_MODE) = (float)RM_DONE;
    }
  }
  R1(RS_FLAG) = flag; R1(RS_SW) = sw;
  any = R1(RS_MODE) != (float)RM_DONE ? 1 : 0;
  {
    // tile-wide: "a row will be in an initial-step phase during the next attempt" (only then are its three extra norms needed) and
    // "a row ends a solve with this attempt" (only then is the interpolant needed); the other lanes skip that arithmetic otherwise
    const float mn = R1(RS_MODE);
    const unsigned long long bi = __ballot(mn == (float)RM_INIT0 || mn == (float)RM_INIT1), bf = __ballot(flag == 2.f);
    if (ln == 0) { *T.at(0, S::RS + RS_TILE * 16 + 2) = bi ? 1.f : 0.f; *T.at(0, S::RS + RS_TILE * 16 + 3) = bf ? 1.f : 0.f; }
  }
  if constexpr (STICKY) {
    if (!any) R1(RS_RANK) = -1.f;
  } else {
    // rank of this row among the rows that take part in the next attempt: with <= 3 of them the time batch is compacted
    const unsigned long long bal = __ballot(any != 0);
    R1(RS_RANK) = any ? (float)__popcll(bal & ((1ull << ln) - 1ull)) : -1.f;
    if (ln == 0) { const int na = __popcll(bal); *T.at(0, S::RS + RS_TILE * 16 + 1) = na <= 2 * TAIL_PASSES ? 3.f : ((TAIL_PASSES < 2 && na <= 3) ? 2.f : (na <= 8 ? 1.f : 0.f)); }
  }
  return any;
}

// ---- THE TAIL OF A TILE: at most two rows still integrate (solve2_tail) ------------------------------------------------------
// The launch lasts as long as its slowest tile, and that tile runs its last 110-230 attempts with one or two rows (tools/tail_stats.py).
// Inside the main loop those attempts carried the full layout's register state (seven stage derivatives of 16 rows: 64 VGPRs of
// which 4-8 are live), its spills (five scratch reloads per evaluation, each a vmcnt(0) that drains the weight stream) and a time
// batch that pushes every weight fragment through a 16-row tile and its results through a global scratch.  A stand-alone model of
// the 4-row evaluation (tools/mb/micro_eval.hip) runs in 9.3 k cycles against the 14.4 k measured in the main loop -- with the SAME
// eight fragments in flight: the stream must simply never be drained.  Hence a loop of its own, entered once (the number of live
// rows never grows) with the state of the two rows moved to a compact layout:
//   * slot r (0, 1) = the row that had rank r when the tail began (sticky: a row that ends leaves its slot empty); lane (g, c) of
//     wave w carries slot g & 1, columns 16 (w + 8 q) + c: y and k_1..7 of 2 elements = 16 registers (lane groups g and g ^ 2 hold
//     copies: after the g-sum of the 4 x 4 x 1 MFMA every lane has all four M-rows of its column, so every lane can finish the
//     value AND the divergence of its slot -- no LDS round trip for the evaluation's result);
//   * ONE weight ring of eight fragments for the whole attempt -- time batch (32 + 8 TPW + 8 fragments) and six evaluations (D / 16
//     + 24 + 8 TPW each), refilled one fragment at a time right after its use, across layers, barriers and attempts (the address
//     sequence is static); no scratch, no spills in the loop;
//   * the time batch of the <= 10 (stage, slot) pairs as one M tile with K = 256 (cos | sin side by side in a [16][264] image: no
//     two-pass accumulation, no sine stash), its results (gate, st contribution to j1) in LDS, read back by plain ds_read;
//     three barriers instead of six; sincos only for the pairs that exist (<= 3 per lane instead of 20).
// Arithmetic: per-row controller, interpolation and step sizes are those of the main loop bit for bit (leaders_end_of_attempt); the
// evaluation runs on the 4 x 4 x 1 MFMA (k-sum per 16-lane group, then over the groups), the time batch sums K = 256 in the even /
// odd k-block order of the single-tile batch, now over both halves at once (a float reassociation of the first time layer).
template <int D, int P>
struct TailMap {
  using S = FS<D>;
  static constexpr int NS = 2 * P;                            // slots
  static constexpr int MT = (10 * P + 15) / 16;               // M tiles of the time batch: M-row NS * stage + slot
  static constexpr int LDF = 2 * F + 8;                       // Fourier image [16 MT][cos 128 | sin 128]
  static constexpr int LDT = D + H + 8;                       // time-batch results [10 P][gate D | j1t H]
  static constexpr int IMG = 4 * P * S::LDH;                  // one image of the x branch: pass p owns rows 4 p .. 4 p + 3
  // area U (dead once the batch is done, then the four x-branch images): the Fourier image, later t1 | st side by side
  static constexpr int USZ = (16 * MT * LDF > 2 * 16 * MT * S::LDH ? 16 * MT * LDF : 2 * 16 * MT * S::LDH) > 4 * IMG
                           ? (16 * MT * LDF > 2 * 16 * MT * S::LDH ? 16 * MT * LDF : 2 * 16 * MT * S::LDH) : 4 * IMG;
  static constexpr int FH2 = S::R, T1I = S::R, STI = S::R + 16 * MT * S::LDH;
  static constexpr int A1 = S::R, SX = A1 + IMG, J1 = SX + IMG, J2 = J1 + IMG;
  static constexpr int TBR = S::R + USZ;                      // lives through the attempt
  static constexpr int STG = S::R;                            // [slot][y, k_1..7][D]: state transfer at entry and exit
  static_assert(TBR + 10 * P * LDT <= S::TOTAL && NS * 8 * D <= S::TOTAL - S::R && 4 * P <= 16, "tail LDS map");
  // fragment i of the attempt's time batch / of an evaluation, wave w: byte offset in the packed weights
  static __device__ __forceinline__ int tb_soff(int i, int w) {
    constexpr int G = 8 * FTile<D>::TPW;
    return i < 16 ? S::W0 * 4 + (w * 16 + i) * 1024
         : i < 24 ? S::W1 * 4 + (w * 8 + i - 16) * 1024
         : i < 24 + G ? S::W4 * 4 + ((w + 8 * ((i - 24) >> 3)) * 8 + ((i - 24) & 7)) * 1024
         : S::W5 * 4 + (w * 16 + 8 + i - 24 - G) * 1024;
  }
  static constexpr int NTB = 32 + 8 * FTile<D>::TPW;
  // the out layer's fragments come column tile by column tile (q-major): one accumulator set per pass is live at a time
  static __device__ __forceinline__ int ev_soff(int i, int w) {
    constexpr int K1 = D / 16;
    return i < K1 ? S::W2 * 4 + (w * K1 + i) * 1024
         : i < K1 + 8 ? S::W3 * 4 + (w * 8 + i - K1) * 1024
         : i < K1 + 16 ? S::W5 * 4 + (w * 16 + i - K1 - 8) * 1024
         : i < K1 + 24 ? S::W6 * 4 + (w * 8 + i - K1 - 16) * 1024
         : S::W7 * 4 + ((w + 8 * ((i - K1 - 24) >> 3)) * 8 + ((i - K1 - 24) & 7)) * 1024;
  }
  static constexpr int NEV = D / 16 + 24 + 8 * FTile<D>::TPW;
  static_assert(NTB % 8 == 0 && NEV % 8 == 0, "the ring positions are static");
};

// Runs the tile from <= 2 P live rows until <= 2 (P - 1) are left (P = 1: until none is).  Entry: RS_RANK = rank of the row among
// the live ones (-1: not live), the Runge-Kutta state of live row r in STG[rank r] (y, k_1..7: written by the caller, or by the loop
// that just ended), a barrier since.  Exit: the same for the rows that are still live, ranked again; the final y of a row that
// ended here in its (now free) row of the probe image.  Nothing is passed in registers: the full layout's sixty-four state
// registers must not stay live across these loops (when they did, the allocator spilled them inside the MAIN loop).
// (what the loops need of the kernel's arguments, by value: the loops are functions of their own -- see below -- and a reference to
// the kernel's argument structs would pin those to memory for the main loop as well)
struct TailArgs {
  float rtol, atol; int max_attempts; int d_true;
  float coef, tbeta, clip; const float* fourier; const float* Wp;
  const float* zgen; const float* z2;
  int mode; float ref_std;                                    // FlowArgs::mode, ::ref_std
  Replay rp;
};
__device__ __forceinline__ int true_dim(const TailArgs& a) { return a.d_true; }
// NOT inlined: inlined into solve2, the three loops changed the register allocation of the MAIN loop (its Runge-Kutta stages went
// to scratch: 50 -> 63 ms).  As functions they get an allocation of their own and the main loop keeps the one it had.
template <int D, bool RP, int P, bool PAD>
__device__ __noinline__ void solve2_tail(TailArgs a, int b0) {
  using S = FS<D>;
  using M = TailMap<D, P>;
  // The workgroup's LDS and the weight descriptor are formed HERE: as arguments they arrive as a flat pointer (every LDS address
  // a 64-bit sum of its own, no immediate offsets) and as a descriptor in vector registers (every buffer load in a waterfall loop).
  extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
  FTile<D> T;
  T.lds = lds_dyn_;
  {
    const unsigned long long wp = (unsigned long long)a.Wp;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)wp), hi = __builtin_amdgcn_readfirstlane((unsigned)(wp >> 32));
    T.wr = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, S::WTOT * 4, 0x00020000);
  }
  const TailArgs& f = a;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX, LDH = S::LDH, LDF = M::LDF, LDT = M::LDT, NTB = M::NTB, NEV = M::NEV, K1 = D / 16;
  constexpr int NS = M::NS, MT = M::MT;
  // Every per-lane quantity of this loop is derived HERE from an opaque copy of the thread index: values computed before the main
  // loop (the tile's lane coordinates, its LDS offsets) are live across it, where all 256 registers are taken, so the allocator
  // spills their whole live range and every use in this loop would be a scratch reload -- a vector-memory load whose wait drains
  // the weight ring (seen in the ISA of the first version: one reload per layer for the bias column alone).
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int lane = tid_ & 63, g = lane >> 4, c = lane & 15, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int sl2 = g & 1;                                      // slot of this lane inside a pass
  const bool is_t = g >= 2;
  float* const lds = T.lds;
  const float scale = 2.38f / sqrtf((float)(PAD ? a.d_true : D));                               // :262
  const float rtol = a.rtol, atol = a.atol;
  const float t_coef = a.coef, t_beta = a.tbeta, t_clip = a.clip;
  const float ffreq = a.fourier[16 * wave + c];
  const float* const biasp = lds + S::BIAS + 16 * wave + c;    // column 16 wave + c (+ 128 q) of a layer's bias
  auto RSF = [&](int field, int row) -> float& { return lds[S::RS + field * 16 + row]; };
  // ---- slots: the row that holds rank r now keeps slot r ----
  int rowof[NS];
  {
    const float rkv = RSF(RS_RANK, lane & 15);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const unsigned long long m = __ballot(rkv == (float)s_) & 0xFFFFull;
      rowof[s_] = m ? __builtin_ctzll(m) : -1;
    }
  }
  int myrow[P], mr[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { myrow[p] = sl2 ? rowof[2 * p + 1] : rowof[2 * p]; mr[p] = myrow[p] < 0 ? 0 : myrow[p]; }   // (an empty slot computes on row 0's numbers and stores nothing)
  float ym[P][TPW], km[P][7][TPW];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = 2 * p + sl2;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = 16 * (wave + NW * q) + c;
      ym[p][q] = lds[M::STG + (slot * 8 + 0) * D + col];
#pragma unroll
      for (int j = 0; j < 7; ++j) km[p][j][q] = lds[M::STG + (slot * 8 + 1 + j) * D + col];
      if (!is_t && myrow[p] >= 0) {                           // probe rows 4 p + 2, 4 p + 3 of both stage-input images
        const float z = lds[S::ZB + myrow[p] * LDX + 4 + col];
        lds[S::XB0 + (4 * p + 2 + sl2) * LDX + 4 + col] = z; lds[S::XB1 + (4 * p + 2 + sl2) * LDX + 4 + col] = z;
      }
    }
  }
  __syncthreads();
  // ---- the weight ring ----
  f32x4 ring[8];
  const int voff = lane * 16;
#pragma unroll
  for (int i = 0; i < 8; ++i) ring[i] = bload(T.wr, voff, M::tb_soff(i, wave));
  // one fragment = one k-block of one column tile: per pass (per M tile) A from LDS and four MFMAs, then the ring slot is refilled
  // eight fragments ahead
#define TAIL_F4(I, AP, PSTRIDE, ACC, NEXT)                                                                              \
  {                                                                                                                    \
    const f32x4 b_ = ring[(I) & 7];                                                                                    \
    _Pragma("unroll") for (int p_ = 0; p_ < P; ++p_) {                                                                 \
      const f32x4 a_ = *reinterpret_cast<const f32x4*>((AP) + p_ * (PSTRIDE));                                        \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) (ACC)[p_][s_] = mfma1(a_[s_], b_[s_], (ACC)[p_][s_]);          \
    }                                                                                                                  \
    ring[(I) & 7] = bload(T.wr, voff, (NEXT));                                                                         \
    asm volatile("" ::: "memory");                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }
#define TAIL_F16(I, AP, ACC, EO, NEXT)                                                                                  \
  {                                                                                                                    \
    const f32x4 b_ = ring[(I) & 7];                                                                                    \
    _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                                                \
      const f32x4 a_ = *reinterpret_cast<const f32x4*>((AP) + m_ * 16 * (LDA_));                                       \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) (ACC)[m_][EO] = mfma4(a_[s_], b_[s_], (ACC)[m_][EO]);          \
    }                                                                                                                  \
    ring[(I) & 7] = bload(T.wr, voff, (NEXT));                                                                         \
    asm volatile("" ::: "memory");                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }
  const int colw = 16 * wave + c;                             // this lane's column inside a 128-wide layer
  int cur = 0;
#ifdef MFM_STAMPS
  const unsigned long long tp0_ = __builtin_amdgcn_s_memtime();
  unsigned long long tp_att_ = 0;
#endif
#pragma unroll 1
  for (;;) {
#ifdef MFM_STAMPS
    tp_att_ += 1;
#endif
    float mode[P], hs[P], sgn[P];
    bool alive[P], att[P], in1[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      mode[p] = RSF(RS_MODE, mr[p]); hs[p] = RSF(RS_DT, mr[p]); sgn[p] = RSF(RS_SIGN, mr[p]);
      alive[p] = myrow[p] >= 0 && RSF(RS_RANK, mr[p]) >= 0.f;
      att[p] = mode[p] == (float)RM_ATT; in1[p] = mode[p] == (float)RM_INIT1;
    }
#pragma unroll 1
    for (int phase = 2; phase < 8; ++phase) {
      // ---- stage inputs of my slots -> X[cur] rows 4 p + slot ----
      {
        float cf[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) cf[j] = DP_TAB[phase][j];
        float* const X = lds + (cur ? S::XB1 : S::XB0);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float c0 = (in1[p] && phase == 2) ? 1.f : cf[0];
          const float he = (att[p] || phase == 2) ? hs[p] : 0.f;
#pragma unroll
          for (int q = 0; q < TPW; ++q) {
            float acc = c0 * km[p][0][q];
#pragma unroll
            for (int j = 1; j < 6; ++j) acc += cf[j] * km[p][j][q];
            const float xin = ym[p][q] + he * acc;
            if (!is_t && alive[p]) X[(4 * p + sl2) * LDX + 4 + colw + 128 * q] = xin;
          }
        }
      }
#ifdef MFM_STAMPS
      const unsigned long long tb0_ = __builtin_amdgcn_s_memtime();
#endif
      if (phase == 2) {
        // ---- time batch: M-row NS s + slot, s = 0..4 ----
        {
          const double fq = (double)ffreq;
#pragma unroll
          for (int pp = 0; pp < (10 * P + 3) / 4; ++pp) {
            const int m = g + 4 * pp;                         // pair (stage m / NS, slot m % NS), wave-uniform per lane group
            if (m < 10 * P) {
              const int so = m % NS, st = m / NS;
              int rw = rowof[0];
#pragma unroll
              for (int s_ = 1; s_ < NS; ++s_) rw = so == s_ ? rowof[s_] : rw;
              if (rw >= 0 && RSF(RS_RANK, rw < 0 ? 0 : rw) >= 0.f) {
                const float md = RSF(RS_MODE, rw), t0 = RSF(RS_T, rw), h = RSF(RS_DT, rw), sg = RSF(RS_SIGN, rw);
                // (stage fractions by selects: a table indexed per lane would be a vector-memory load, whose wait drains the weight ring)
                const float c5 = st == 0 ? 1.f / 5 : (st == 1 ? 3.f / 10 : (st == 2 ? 4.f / 5 : (st == 3 ? 8.f / 9 : 1.f)));
                const float csr = md == (float)RM_INIT1 ? (st == 0 ? 1.f : 0.f) : c5;
                const float tt = t0 + h * csr;
                const double te = sg > 0.f ? (double)tt : 1.0 - (double)tt;          // :229
                double ft = fq * te;
                ft -= rint(ft);
                float sn, cs;
                sincospi_half_turn(2.f * (float)ft, &sn, &cs);                       // :70-71
                lds[M::FH2 + m * LDF + colw] = cs; lds[M::FH2 + m * LDF + F + colw] = sn;
              }
            }
          }
        }
        __syncthreads();
        f32x4 ac[MT][2];
        auto zero_b = [&]() {
#pragma unroll
          for (int m_ = 0; m_ < MT; ++m_) { ac[m_][0] = f32x4{0, 0, 0, 0}; ac[m_][1] = f32x4{0, 0, 0, 0}; }
        };
        const float* const af = lds + M::FH2 + (lane & 15) * LDF + 4 * g;
        zero_b();
#define LDA_ LDF
#pragma unroll
        for (int i = 0; i < 16; ++i) TAIL_F16(i, af + i * 16, ac, i & 1, M::tb_soff(i + 8, wave))
#undef LDA_
        __syncthreads();                                      // the Fourier image is dead: t1 takes its place
        {
          const float b = biasp[S::B0];
#pragma unroll
          for (int m_ = 0; m_ < MT; ++m_) {
            const f32x4 t1 = ac[m_][0] + ac[m_][1];
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[M::T1I + (16 * m_ + 4 * g + i) * LDH + colw] = fmaxf(t1[i] + b, 0.f);
          }
        }
        __syncthreads();
        const float* const at1 = lds + M::T1I + (lane & 15) * LDH + 4 * g;
        zero_b();
#define LDA_ LDH
#pragma unroll
        for (int i = 0; i < 8; ++i) TAIL_F16(16 + i, at1 + i * 16, ac, i & 1, M::tb_soff(24 + i, wave))
        {
          const float b = biasp[S::B1];
#pragma unroll
          for (int m_ = 0; m_ < MT; ++m_) {
            const f32x4 t2 = ac[m_][0] + ac[m_][1];
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[M::STI + (16 * m_ + 4 * g + i) * LDH + colw] = fmaxf(t2[i] + b, 0.f);          // st
          }
        }
        __syncthreads();
        const float* const ast = lds + M::STI + (lane & 15) * LDH + 4 * g;
#pragma unroll
        for (int q = 0; q < TPW + 1; ++q) {                   // gate tiles wave (+ 8), then the st half of j1
          zero_b();
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int fi = 24 + 8 * q + i;                    // fragment index in the batch; its refill is 8 ahead: the batch, then x1
            TAIL_F16(fi, ast + i * 16, ac, i & 1, fi + 8 < NTB ? M::tb_soff(fi + 8, wave) : M::ev_soff(fi + 8 - NTB, wave))
          }
          const float b = q < TPW ? biasp[S::B4 + 128 * q] : biasp[S::B5];
          const int co = q < TPW ? 128 * q : D;
#pragma unroll
          for (int m_ = 0; m_ < MT; ++m_) {
            const f32x4 r = ac[m_][0] + ac[m_][1];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (16 * m_ + 4 * g + i < 10 * P) lds[M::TBR + (16 * m_ + 4 * g + i) * LDT + co + colw] = r[i] + b;
          }
        }
#undef LDA_
        // The x-branch images take the place of t1 | st: every wave must have finished reading st before the first
        // evaluation's epilogue writes there
        __syncthreads();
      } else {
        __syncthreads();
      }
#ifdef MFM_STAMPS
      const unsigned long long ev0_ = __builtin_amdgcn_s_memtime();
      if (phase == 2) { T.cyc_t1 += ev0_ - tb0_; T.n_t1 += 1; }
#endif
      // ---- one field evaluation of the slots on the 4-row images ----
      float kv[P][TPW];
      {
        const int ss = phase == 7 ? 4 : phase - 2;            // time slot of the stage
        const bool next_tb = phase == 7;
        const float* const X = lds + (cur ? S::XB1 : S::XB0);
        float gt[P][TPW], gc[P][TPW], hz[P][TPW], zz[P][TPW];
        auto target_terms = [&]() {
          const float icoef = 1.f / t_coef;
#pragma unroll
          for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
              const float* xr = X + (4 * p + sl2) * LDX + 4 + colw + 128 * q;
              const float* zr = xr + 2 * LDX;
              const float x = xr[0], z = zr[0];
              const float graw = -t_beta * (t_coef * (2.f * x - xr[-1] - xr[1]) - x * (1.f - x * x) * icoef);
              const float hv = -t_beta * (t_coef * (2.f * z - zr[-1] - zr[1]) - (1.f - 3.f * x * x) * z * icoef);
              gc[p][q] = t_clip > 0.f ? fminf(fmaxf(graw, -t_clip), t_clip) : graw;
              hz[p][q] = (!(t_clip > 0.f) || fabsf(graw) <= t_clip) ? hv : 0.f;
              zz[p][q] = z;
            }
          }
        };
        f32x4 acc[P][4];
        auto zero = [&]() {
#pragma unroll
          for (int p = 0; p < P; ++p)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) acc[p][s_] = f32x4{0, 0, 0, 0};
        };
        // finish one layer: M-row g of this lane's column -> row 4 p + g of the next layer's image
        auto act_store = [&](int p, float add, float* dst) {
          const f32x4 pre = T.gsum((acc[p][0] + acc[p][1]) + (acc[p][2] + acc[p][3]));
          const float pv = (sl2 ? pre[1] : pre[0]) + add, pt = sl2 ? pre[3] : pre[2];
          dst[(4 * p + g) * LDH + colw] = is_t ? (pv > 0.f ? pt : 0.f) : fmaxf(pv, 0.f);
        };
        if (wave < NW / 2) target_terms();
        {   // x1 on [values ; probes]
          const float* const ax = X + (c & 3) * LDX + 4 + 4 * g;
          zero();
#pragma unroll
          for (int i = 0; i < K1; ++i) TAIL_F4(i, ax + i * 16, 4 * LDX, acc, M::ev_soff(i + 8, wave))
          if (wave >= NW / 2) target_terms();
          const float b = biasp[S::B2];
#pragma unroll
          for (int p = 0; p < P; ++p) act_store(p, b, lds + M::A1);
        }
        __syncthreads();
        float j1t[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int mrow = NS * ss + 2 * p + sl2;
#pragma unroll
          for (int q = 0; q < TPW; ++q) gt[p][q] = lds[M::TBR + mrow * LDT + 128 * q + colw];
          j1t[p] = lds[M::TBR + mrow * LDT + D + colw];
        }
        {   // x2
          const float* const ah = lds + M::A1 + (c & 3) * LDH + 4 * g;
          zero();
#pragma unroll
          for (int i = 0; i < 8; ++i) TAIL_F4(K1 + i, ah + i * 16, 4 * LDH, acc, M::ev_soff(K1 + i + 8, wave))
          const float b = biasp[S::B3];
#pragma unroll
          for (int p = 0; p < P; ++p) act_store(p, b, lds + M::SX);
        }
        __syncthreads();
        {   // j1: the st half + bias of the value rows come from the time batch
          const float* const ah = lds + M::SX + (c & 3) * LDH + 4 * g;
          zero();
#pragma unroll
          for (int i = 0; i < 8; ++i) TAIL_F4(K1 + 8 + i, ah + i * 16, 4 * LDH, acc, M::ev_soff(K1 + 16 + i, wave))
#pragma unroll
          for (int p = 0; p < P; ++p) act_store(p, j1t[p], lds + M::J1);
        }
        __syncthreads();
        {   // j2
          const float* const ah = lds + M::J1 + (c & 3) * LDH + 4 * g;
          zero();
#pragma unroll
          for (int i = 0; i < 8; ++i) TAIL_F4(K1 + 16 + i, ah + i * 16, 4 * LDH, acc, M::ev_soff(K1 + 24 + i, wave))
          const float b = biasp[S::B6];
#pragma unroll
          for (int p = 0; p < P; ++p) act_store(p, b, lds + M::J2);
        }
        __syncthreads();
        {   // out, one column tile after the other: every lane ends with all four M-rows of its columns -> value and divergence
            // of its slot, no LDS round trip
          const float* const ah = lds + M::J2 + (c & 3) * LDH + 4 * g;
          float dpv[P];
#pragma unroll
          for (int p = 0; p < P; ++p) dpv[p] = 0.f;
#pragma unroll
          for (int q = 0; q < TPW; ++q) {
            zero();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int fi = K1 + 24 + 8 * q + i;
              const int nx = fi + 8 < NEV ? M::ev_soff(fi + 8, wave) : (next_tb ? M::tb_soff(fi + 8 - NEV, wave) : M::ev_soff(fi + 8 - NEV, wave));
              TAIL_F4(fi, ah + i * 16, 4 * LDH, acc, nx)
            }
            const float b = biasp[S::B7 + 128 * q];
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const f32x4 pre = T.gsum((acc[p][0] + acc[p][1]) + (acc[p][2] + acc[p][3]));
              const float pv = sl2 ? pre[1] : pre[0], pt = sl2 ? pre[3] : pre[2];
              dpv[p] += zz[p][q] * (pt + gt[p][q] * hz[p][q]);                                      // z . J z of the slot
              const float v = pv + b + gt[p][q] * gc[p][q];                                         // v of the slot
              kv[p][q] = sgn[p] > 0.f ? v : -v;
            }
          }
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const float d_ = group16_sum_dpp(dpv[p]);
            if (is_t && c == 0) lds[S::DLP + (phase - 1) * 128 + (8 + 2 * p + sl2) * 8 + wave] = d_;    // M-row 8 + slot of the partial sums (as eval_c); raw
          }
        }
      }
#ifdef MFM_STAMPS
      T.cyc_em += __builtin_amdgcn_s_memtime() - ev0_; T.n_em += 1;
#endif
      cur ^= 1;
#pragma unroll
      for (int j = 1; j < 7; ++j)
        if (j == phase - 1) {
#pragma unroll
          for (int p = 0; p < P; ++p)
#pragma unroll
            for (int q = 0; q < TPW; ++q) km[p][j][q] = kv[p][q];
        }
    }
    // ---- end of the attempt: norms of my slots' rows (lane groups 0, 1; groups 2, 3 hold copies) ----
    const bool some_init = RSF(RS_TILE, 2) != 0.f;             // (published by the leaders of the previous attempt)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float p0 = 0.f, p1 = 0.f, p2 = 0.f, e2 = 0.f;
      if (some_init) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          const float sc = atol + fabsf(ym[p][q]) * rtol;                   // initial-step norms (Hairer II.4)
          const float a0 = ym[p][q] / sc, a1 = km[p][1][q] / sc, a2 = (km[p][1][q] - km[p][0][q]) / sc;
          p0 += a0 * a0; p1 += a1 * a1; p2 += a2 * a2;
        }
        p0 = group16_sum_dpp(p0); p1 = group16_sum_dpp(p1); p2 = group16_sum_dpp(p2);
      }
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        float acc = 0.f, er = 0.f;                                          // error norm of the attempted step
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += DP_TAB[7][j] * km[p][j][q];
        const float y1 = ym[p][q] + hs[p] * acc;
#pragma unroll
        for (int j = 0; j < 7; ++j) er += DP_E[j] * km[p][j][q];
        er *= hs[p];
        const float tol = atol + rtol * fmaxf(fabsf(ym[p][q]), fabsf(y1));
        const float rr = er / tol;
        e2 += rr * rr;
      }
      e2 = group16_sum_dpp(e2);
      if (!is_t && c == 0 && myrow[p] >= 0) {
        float* const rd = lds + S::RED + myrow[p] * 8 + wave;
        if (some_init) { rd[0] = p0; rd[128] = p1; rd[256] = p2; }
        rd[384] = e2;
      }
    }
    if (threadIdx.x == 0) RSF(RS_TILE, 0) = 0.f;              // "some row switched solves in this attempt"
    __syncthreads();
    int any = 0;
    if (wave == 0 && lane < 16) any = leaders_end_of_attempt<D, RP, true, PAD>(T, a, f, b0, 3);
    (void)__syncthreads_or(any);
    // ---- apply the decision of my slots' rows ----
    const bool some_fin = RSF(RS_TILE, 3) != 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float fl = RSF(RS_FLAG, mr[p]), sfrac = RSF(RS_SFRAC, mr[p]);
      const bool swr = RSF(RS_SW, mr[p]) != 0.f && RSF(RS_TILE, 0) != 0.f;
      const bool fin = fl == 2.f, adv = fl == 1.f, ini = fl == 3.f;
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += DP_TAB[7][j] * km[p][j][q];
        const float x0 = ym[p][q], x1 = x0 + hs[p] * acc;
        float xi = x1;
        if (some_fin) {
          float kmid = 0.f;
#pragma unroll
          for (int j = 0; j < 7; ++j) kmid += DP_M[j] * km[p][j][q];
          const float xm = x0 + hs[p] * kmid, g0 = hs[p] * km[p][0][q], g1 = hs[p] * km[p][6][q];
          const float qa = -2.f * g0 + 2.f * g1 - 8.f * x0 - 8.f * x1 + 16.f * xm;
          const float qb = 5.f * g0 - 3.f * g1 + 18.f * x0 + 14.f * x1 - 32.f * xm;
          const float qc = -4.f * g0 + g1 - 11.f * x0 - 5.f * x1 + 16.f * xm;
          xi = (((qa * sfrac + qb) * sfrac + qc) * sfrac + g0) * sfrac + x0;
        }
        float yn = fin ? xi : (adv ? x1 : x0);
        km[p][0][q] = ini ? km[p][1][q] : (adv ? km[p][6][q] : km[p][0][q]);
        if (swr && myrow[p] >= 0) {          // this row starts its forward solve: latent proposal, forward probe
          const int col = colw + 128 * q;
          const size_t o = (size_t)(b0 + myrow[p]) * D + col;
          const float nz = a.zgen[o];
          if (f.mode == MFM_FLOW_RWMH) yn = yn + scale * nz;                                        // :268
          else { const float up = f.ref_std * nz; r0 += yn * yn; r1 += up * up; yn = up; }          // :249
          const float z2 = a.z2[o];                                                                 // key_hutch1
          if (!is_t) {
            lds[S::XB0 + (4 * p + 2 + sl2) * LDX + 4 + col] = z2; lds[S::XB1 + (4 * p + 2 + sl2) * LDX + 4 + col] = z2;
            lds[S::ZB + myrow[p] * LDX + 4 + col] = z2;       // the full layout's probe row (a row that leaves this loop alive needs it)
          }
#pragma unroll
          for (int j = 0; j < 7; ++j) km[p][j][q] = 0.f;
        }
        ym[p][q] = yn;
      }
      if (RSF(RS_TILE, 0) != 0.f && f.mode == MFM_FLOW_IMH) {      // tile-uniform: ref.logprob terms of the rows that switched
        r0 = group16_sum_dpp(r0); r1 = group16_sum_dpp(r1);
        if (!is_t && c == 0 && swr && myrow[p] >= 0) { lds[S::RED + 4 * 128 + myrow[p] * 8 + wave] = r0; lds[S::RED + 5 * 128 + myrow[p] * 8 + wave] = r1; }
      }
    }
    // leave when the rows that are left fit a smaller loop (P = 1: when none is left)
    {
      const unsigned long long lv = __ballot(RSF(RS_RANK, lane & 15) >= 0.f) & 0xFFFFull;
      if (__popcll(lv) <= 2 * (P - 1)) break;
    }
    // (the next attempt's stage input is visible after the time batch's first barrier; the leaders rewrite the row state only
    // after the end-of-attempt barrier of that attempt)
  }
#undef TAIL_F4
#undef TAIL_F16
#ifdef MFM_STAMPS
  if (g_flow_dbg && threadIdx.x == 0) {      // cumulative over the launches of a process: tools/flow_cycles.py takes differences
    unsigned long long* o = g_flow_dbg + blockIdx.x * 64;
    o[54 + 2 * (P - 1)] += __builtin_amdgcn_s_memtime() - tp0_; o[55 + 2 * (P - 1)] += tp_att_;
    if (P == 1) {                             // inside the one-pass tail: its evaluations and time batches (second half of the buffer)
      unsigned long long* o4 = g_flow_dbg + (blockIdx.x + gridDim.x) * 64;
      o4[54] += T.n_em; o4[55] += T.cyc_em; o4[56] += T.n_t1; o4[57] += T.cyc_t1;
    }
  }
#endif
  // ---- leave: the rows that are still live are ranked again and their state goes to STG[new rank]; a row that ended here leaves
  // its final y in its row of the probe image (free now) ----
  __syncthreads();
  {
    const unsigned long long lv = __ballot(RSF(RS_RANK, lane & 15) >= 0.f) & 0xFFFFull;
    if (!is_t) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (myrow[p] >= 0) {
          const bool live = (lv >> myrow[p]) & 1ull;
          const int nr = __popcll(lv & ((1ull << myrow[p]) - 1ull));
#pragma unroll
          for (int q = 0; q < TPW; ++q) {
            const int col = colw + 128 * q;
            if (live) {
              lds[M::STG + (nr * 8 + 0) * D + col] = ym[p][q];
#pragma unroll
              for (int j = 0; j < 7; ++j) lds[M::STG + (nr * 8 + 1 + j) * D + col] = km[p][j][q];
            } else {
              lds[S::ZB + myrow[p] * LDX + 4 + col] = ym[p][q];
            }
          }
        }
      }
    }
    __syncthreads();                                          // every lane has read the old ranks
    if (wave == 0 && lane < 16) RSF(RS_RANK, lane) = ((lv >> lane) & 1ull) ? (float)__popcll(lv & ((1ull << lane) - 1ull)) : -1.f;
    __syncthreads();
  }
}

// ---- the flow step's two solves with PER-ROW phases ------------------------------------------------------------------
// flow_step = inverse solve of the current position, latent proposal, forward solve of the proposal (:264-278 / :246-260).
// Run as two tile-wide solves, a tile waits for its slowest inverse solve AND then for its slowest forward solve
// (max_i inv_i + max_i fwd_i attempted steps; the two are uncorrelated across chains: tools/natt_split.py).  Here every
// row carries its own (solve, mode) and switches to its forward solve as soon as ITS inverse solve ends, so a tile takes
// max_i (inv_i + fwd_i) attempts (+ ~4 per row: see below).  The tile shares only the attempt clock: the loop body is
// always one Dormand-Prince attempt (time batch + six x-branch evaluations).  A row that starts a solve spends two
// attempts on jax's initial-step heuristic instead of two evaluations: INIT0 rides with dt = 0 (slot 0 evaluates f(y, t0)),
// INIT1 with dt = h0 and unit stage coefficients in slot 0 (the extra evaluation at t0 + h0); their remaining slots are
// ignored.  Per-row arithmetic (controller, interpolation, step sizes) is that of solve() bit for bit.
// On return: y = proposal x' at t = 1 of the forward solve, row state holds ell (forward), vol0 (inverse), lq, counts.
template <int D, bool RP, bool PAD>
__device__ __forceinline__ void solve2(FTile<D>& T, const OdeArgs& a, const FlowArgs& f, int b0, float (&y)[FTile<D>::TPW][4], int live, int fmode) {
  using S = FS<D>;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX;
  const int g = T.g, c = T.c, wave = T.wave;
  const float rtol = a.rtol, atol = a.atol;
  const int max_attempts = a.max_attempts;
  const float scale = 2.38f / sqrtf((float)(PAD ? a.net.d : D));                                // :262
  float k[7][TPW][4];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) k[j][q][i] = 0.f;
  {
    const float z4[4] = {0.f, 0.f, 0.f, 0.f}, m1[4] = {-1.f, -1.f, -1.f, -1.f};
    __syncthreads();
#pragma unroll
    for (int fld = 0; fld < 24; ++fld) T.rs_put(fld, z4);      // mode INIT0, solve 0, t = 0, dt = 0
    T.rs_put(RS_SIGN, m1);                                     // inverse solve first (:267 / :251)
    if (live < 16) {          // rows without a chain: done from the start, so that the tile runs in the layout of its live rows
      float md[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) md[i] = 4 * g + i < live ? 0.f : (float)RM_DONE;
      T.rs_put(RS_MODE, md);
    }
    if (threadIdx.x == 0) *T.at(0, S::RS + RS_TILE * 16 + 2) = 1.f;      // every row starts in an initial-step phase
  }
  f32x4 P[4], Q[4];
  load_group<1, 0>(P, T.wtr, T.W(S::W7, wave, D / 16), T.lane);
  __syncthreads();
  T.precompute_w7z(P, Q);
  T.precompute_tz1(P, Q);

  int phase = 2, cur = 0;
#pragma unroll 1
  for (;;) {
    // ---- stage input -> X[cur] ----
#ifdef MFM_STAMPS
    const unsigned long long si0_ = __builtin_amdgcn_s_memtime();
#endif
    float hs[4];
    const f32x4 md4 = T.rs_get(RS_MODE), rk4 = T.rs_get(RS_RANK);
    const int cmode = (int)*T.at(0, S::RS + RS_TILE * 16 + 1);             // 0: > 8 rows of the tile still integrate, 1: <= 8, 2: <= 3, 3: <= 2
    const int rps = cmode >= 2 ? 3 : (cmode == 1 ? 8 : 0);
    const int prow = 8;                                                    // first probe row of the compact input image
    {
      float cf[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) cf[j] = DP_TAB[phase][j];
      const f32x4 h4 = T.rs_get(RS_DT);
      const int xsel = cur ? S::XB1 * 4 : S::XB0 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hs[i] = h4[i];
        const bool att = md4[i] == (float)RM_ATT, in1 = md4[i] == (float)RM_INIT1;
        const float c0 = (in1 && phase == 2) ? 1.f : cf[0];
        const float he = (att || phase == 2) ? hs[i] : 0.f;      // rows in their initial-step phases only use slot 0
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          float acc = c0 * k[0][q][i];
#pragma unroll
          for (int j = 1; j < 6; ++j) acc += cf[j] * k[j][q][i];
          const float xin = y[q][i] + he * acc;
          if (cmode == 0) *T.at(T.o_xo + xsel, i * LDX + 128 * q) = xin;
          else if (rk4[i] >= 0.f) {                              // compact evaluation: value rows by rank, probes by rank below them
            *T.at(T.o_xc + xsel, (int)rk4[i] * LDX + 128 * q) = xin;
            if (phase == 2) {
              const float z = *T.at(T.o_xo, S::ZB + i * LDX + 128 * q);
              *T.at(T.o_xc + S::XB0 * 4, (prow + (int)rk4[i]) * LDX + 128 * q) = z;
              *T.at(T.o_xc + S::XB1 * 4, (prow + (int)rk4[i]) * LDX + 128 * q) = z;
            }
          }
        }
      }
    }
#ifdef MFM_STAMPS
    const unsigned long long tb0_ = __builtin_amdgcn_s_memtime();
    T.cyc_si += tb0_ - si0_;
#endif
    if (phase == 2) {
      float cvv[5][4], svv[5][4];
      T.template tb_trig<true>(2, cvv, svv);
      if (TAIL_PASSES < 2 && cmode >= 2) T.template tbatch<true, 1>(2, P, Q, cvv, svv);
      else if (cmode == 1) T.template tbatch<true, 3>(2, P, Q, cvv, svv);
      else T.template tbatch<true, 5>(2, P, Q, cvv, svv);
    } else __syncthreads();
#ifdef MFM_STAMPS
    if (phase == 2) { const unsigned long long d_ = __builtin_amdgcn_s_memtime() - tb0_;
      if (cmode == 0) { T.cyc_tb += d_; T.n_tb += 1; } else if (cmode == 1) { T.cyc_tc += d_; T.n_tc += 1; } else { T.cyc_t1 += d_; T.n_t1 += 1; } }
#endif
    float kv[TPW][4];
    const int dst = phase - 1;
#ifdef MFM_STAMPS
    const unsigned long long ce0_ = __builtin_amdgcn_s_memtime();
#endif
    if (cmode == 0) T.eval(phase == 7 ? 4 : phase - 2, cur, dst, phase == 7, P, Q, kv, T.rs_get(RS_SIGN), 0, rk4);
    else T.eval_c(phase == 7 ? 4 : phase - 2, cur, dst, phase == 7, P, Q, kv, T.rs_get(RS_SIGN), rps, rk4);
#ifdef MFM_STAMPS
    { const unsigned long long d_ = __builtin_amdgcn_s_memtime() - ce0_;
      if (cmode == 0) { T.cyc_eval += d_; T.n_eval += 1; } else if (cmode == 3) { T.cyc_em += d_; T.n_em += 1; } else { T.cyc_ec += d_; T.n_ec += 1; } }
#endif
    cur ^= 1;
#pragma unroll
    for (int j = 1; j < 7; ++j)
      if (j == dst) {
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) k[j][q][i] = kv[q][i];
      }
    if (phase < 7) { phase += 1; continue; }

    // ---- end of an attempt: per-lane partials of every norm a row may need in its mode ----
#ifdef MFM_STAMPS
    const unsigned long long n0_ = __builtin_amdgcn_s_memtime();
#endif
    {
      float p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0}, p2[4] = {0, 0, 0, 0}, e2[4] = {0, 0, 0, 0};
      const bool some_init = *T.at(0, S::RS + RS_TILE * 16 + 2) != 0.f;     // (published by the leaders of the previous attempt)
      if (some_init) {
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float sc = atol + fabsf(y[q][i]) * rtol;                  // initial-step norms (Hairer II.4)
            const float a0 = y[q][i] / sc, a1 = k[1][q][i] / sc, a2 = (k[1][q][i] - k[0][q][i]) / sc;
            p0[i] += a0 * a0; p1[i] += a1 * a1; p2[i] += a2 * a2;
          }
        T.part_put(S::RED + 0 * 128, p0); T.part_put(S::RED + 1 * 128, p1); T.part_put(S::RED + 2 * 128, p2);
      }
#pragma unroll
      for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float acc = 0.f, er = 0.f;                                       // error norm of the attempted step
#pragma unroll
          for (int j = 0; j < 6; ++j) acc += DP_TAB[7][j] * k[j][q][i];
          const float y1 = y[q][i] + hs[i] * acc;
#pragma unroll
          for (int j = 0; j < 7; ++j) er += DP_E[j] * k[j][q][i];
          er *= hs[i];
          const float tol = atol + rtol * fmaxf(fabsf(y[q][i]), fabsf(y1));
          const float rr = er / tol;
          e2[i] += rr * rr;
        }
      T.part_put(S::RED + 3 * 128, e2);
    }
    if (threadIdx.x == 0) *T.at(0, S::RS + RS_TILE * 16) = 0.f;            // "some row switched solves in this attempt"
    __syncthreads();
#ifdef MFM_STAMPS
    const unsigned long long n1_ = __builtin_amdgcn_s_memtime();
#endif
    int any = 0;
    if (wave == 0 && T.lane < 16) any = leaders_end_of_attempt<D, RP, false, PAD>(T, a, f, b0, cmode);
    const int go = __syncthreads_or(any);
#ifdef MFM_STAMPS
    const unsigned long long n2_ = __builtin_amdgcn_s_memtime();
    T.cyc_n1 += n1_ - n0_; T.cyc_n2 += n2_ - n1_;
#endif
    // ---- every lane: apply the decision of its rows (branch-free selects) ----
    const bool tile_sw = *T.at(0, S::RS + RS_TILE * 16) != 0.f;
    const bool some_fin = *T.at(0, S::RS + RS_TILE * 16 + 3) != 0.f;       // tile-uniform: a row reaches t = 1 with this attempt
    {
      const f32x4 fl4 = T.rs_get(RS_FLAG), sf4 = T.rs_get(RS_SFRAC), sw4 = T.rs_get(RS_SW);
      float r0[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dti = hs[i], sfrac = sf4[i];
        const bool fin = fl4[i] == 2.f, adv = fl4[i] == 1.f, ini = fl4[i] == 3.f, swr = sw4[i] != 0.f;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j) acc += DP_TAB[7][j] * k[j][q][i];
          const float x0 = y[q][i], x1 = x0 + dti * acc;
          float xi = x1;
          if (some_fin) {                     // the interpolant at t = 1 (2 of a row's ~330 attempts need it)
            float km = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) km += DP_M[j] * k[j][q][i];
            const float xm = x0 + dti * km, g0 = dti * k[0][q][i], g1 = dti * k[6][q][i];
            const float qa = -2.f * g0 + 2.f * g1 - 8.f * x0 - 8.f * x1 + 16.f * xm;
            const float qb = 5.f * g0 - 3.f * g1 + 18.f * x0 + 14.f * x1 - 32.f * xm;
            const float qc = -4.f * g0 + g1 - 11.f * x0 - 5.f * x1 + 16.f * xm;
            xi = (((qa * sfrac + qb) * sfrac + qc) * sfrac + g0) * sfrac + x0;
          }
          float yn = fin ? xi : (adv ? x1 : x0);
          k[0][q][i] = ini ? k[1][q][i] : (adv ? k[6][q][i] : k[0][q][i]);
          if (tile_sw && swr) {              // this row starts its forward solve: latent proposal, forward probe
            const int col = 16 * (wave + NW * q) + c;
            const size_t o = (size_t)(b0 + 4 * g + i) * D + col;
            const float nz = a.zgen[o];
            if (fmode == MFM_FLOW_RWMH) yn = yn + scale * nz;                                         // :268
            else { const float up = f.ref_std * nz; r0[i] += yn * yn; r1[i] += up * up; yn = up; }      // :249
            *T.at(T.o_xo, S::ZB + i * LDX + 128 * q) = a.z2[o];                                        // key_hutch1
#pragma unroll
            for (int j = 0; j < 7; ++j) k[j][q][i] = 0.f;
          }
          y[q][i] = yn;
        }
      }
      if (tile_sw) {                         // tile-uniform
        if (fmode == MFM_FLOW_IMH) { T.part_put(S::RED + 4 * 128, r0); T.part_put(S::RED + 5 * 128, r1); }
        __syncthreads();                     // the new probe rows are visible
        load_group<1, 0>(P, T.wtr, T.W(S::W7, wave, D / 16), T.lane);
        T.precompute_w7z(P, Q);              // W_out z and z W_x1 of every row (unchanged rows recompute the same values)
        T.precompute_tz1(P, Q);
      }
    }
#ifdef MFM_STAMPS
    T.cyc_n3 += __builtin_amdgcn_s_memtime() - n2_;
#endif
    if (!go) break;
    if (TAIL_PASSES > 0 && *T.at(0, S::RS + RS_TILE * 16 + 1) == 3.f) {      // <= 2 TAIL_PASSES rows left: the tile's tail runs in loops of its own
      using M1 = TailMap<D, 1>;
      unsigned intail = 0;
      {
        const f32x4 rk4n = T.rs_get(RS_RANK);                               // rank among the live rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (rk4n[i] >= 0.f) {
            intail |= 1u << i;
            const int r = (int)rk4n[i];
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
              const int col = 16 * (wave + NW * q) + c;
              T.lds[M1::STG + (r * 8 + 0) * D + col] = y[q][i];
#pragma unroll
              for (int j = 0; j < 7; ++j) T.lds[M1::STG + (r * 8 + 1 + j) * D + col] = k[j][q][i];
            }
          }
        }
      }
      __syncthreads();
#ifdef MFM_STAMPS
      const unsigned long long tl0_ = __builtin_amdgcn_s_memtime();
#endif
      TailArgs ta;
      ta.rtol = a.rtol; ta.atol = a.atol; ta.max_attempts = a.max_attempts; ta.d_true = a.net.d;
      ta.coef = a.net.T.coef; ta.tbeta = a.net.T.tbeta; ta.clip = a.net.grad_clip; ta.fourier = a.net.fourier; ta.Wp = a.net.Wp;
      ta.zgen = a.zgen; ta.z2 = a.z2; ta.mode = fmode; ta.ref_std = f.ref_std; ta.rp = a.rp;
#pragma unroll 1
      for (;;) {
        const int live = __popcll(__ballot(*T.at((T.lane & 15) * 4, S::RS + RS_RANK * 16) >= 0.f) & 0xFFFFull);
        if (live == 0) break;
        if (TAIL_PASSES >= 3 && live > 4) solve2_tail<D, RP, TAIL_PASSES >= 3 ? 3 : 1, PAD>(ta, b0);
        else if (TAIL_PASSES >= 2 && live > 2) solve2_tail<D, RP, TAIL_PASSES >= 2 ? 2 : 1, PAD>(ta, b0);
        else solve2_tail<D, RP, 1, PAD>(ta, b0);
      }
#ifdef MFM_STAMPS
      T.cyc_tail += __builtin_amdgcn_s_memtime() - tl0_;
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (intail >> i & 1u) {
#pragma unroll
          for (int q = 0; q < TPW; ++q) y[q][i] = *T.at(T.o_xo, S::ZB + i * LDX + 128 * q);
        }
      break;
    }
    phase = 2;
  }
}

template <int D>
__device__ __forceinline__ void tile_init(FTile<D>& T, const NetDev& n, float* lds, f32x4* scr_wg) {
  using S = FS<D>;
  T.lds = lds;
  T.lane = threadIdx.x & 63; T.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); T.g = T.lane >> 4; T.c = T.lane & 15;
  T.wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(n.Wp), 0, S::WTOT * 4, 0x00020000);
  T.sr = __builtin_amdgcn_make_buffer_rsrc(scr_wg, 0, SCR_F4_PER_WG * 16, 0x00020000);
  T.wtr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(n.WpT), 0, S::WTOT * 4, 0x00020000);
  T.sign = 1;
  T.ffreq = n.fourier[16 * T.wave + T.c];
  T.coef = n.T.coef; T.tbeta = n.T.tbeta; T.clip = n.grad_clip;
  {
    const int r = T.lane & 15, g = T.g, c = T.c, w = T.wave;
    T.o_rs = 16 * g; T.o_pg = 128 * g; T.o_pp = (32 * g + w) * 4; T.o_bias = (16 * w + c) * 4;
    T.o_xa = (r * S::LDX + 4 * g + 4) * 4; T.o_xo = (4 * g * S::LDX + 4 + 16 * w + c) * 4;
    T.o_ha = (S::R + r * S::LDH + 4 * g) * 4; T.o_ha2 = T.o_ha + 64 * S::LDH * 4;
    T.o_he = (S::R + 4 * g * S::LDH + 16 * w + c) * 4; T.o_he2 = T.o_he + 64 * S::LDH * 4;
    asm volatile("" : "+v"(T.o_rs), "+v"(T.o_pg), "+v"(T.o_pp), "+v"(T.o_bias), "+v"(T.o_xa));
    asm volatile("" : "+v"(T.o_xo), "+v"(T.o_ha), "+v"(T.o_ha2), "+v"(T.o_he), "+v"(T.o_he2));
    T.o_l8 = (T.lane & 15) * 32; T.o_l1 = (T.lane & 15) * 4;
    T.o_hc = (S::R + 16 * w + c) * 4; T.o_xc = (4 + 16 * w + c) * 4;
    asm volatile("" : "+v"(T.o_l8), "+v"(T.o_l1), "+v"(T.o_hc), "+v"(T.o_xc));
  }
  for (int i = threadIdx.x; i < S::TOTAL; i += NW * 64)
    if (i < S::BIAS || i >= S::BIAS + S::BTOT) lds[i] = 0.f;                   // halo pads, row state, scratch
  for (int i = threadIdx.x; i < S::BTOT; i += NW * 64) lds[S::BIAS + i] = n.bias[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) { T.tz1[i] = 0.f; T.w7z[i] = 0.f; }
  __syncthreads();
}

// (`live` < 16: the tile carries chains in its first `live` rows only; the other rows read the last of them -- finite values, never used)
template <int D>
__device__ __forceinline__ void fill_probe(FTile<D>& T, const float* z, int b0, int live = 16) {
  using S = FS<D>;
#pragma unroll
  for (int q = 0; q < FTile<D>::TPW; ++q) {
    const int col = 16 * (T.wave + NW * q) + T.c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * T.g + i;
      *T.at(T.o_xo, S::ZB + i * S::LDX + 128 * q) = z[(size_t)(b0 + (r < live ? r : live - 1)) * D + col];
    }
  }
}

template <int D, bool RP, bool PAD = false>
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
    solve<D, RP>(T, a.rtol, a.atol, a.max_attempts, y, ell, natt, a.rp, b0, PAD ? a.net.d : D);
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
template <int D, bool RP, bool PAD = false>
__global__ __launch_bounds__(NW * 64) void flow_step_fast_kernel(OdeArgs a, FlowArgs f, NoiseArgs nz, f32x4* scratch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using S = FS<D>;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX;
  FTile<D> T;
#ifdef MFM_STAMPS
  const unsigned long long fc0_ = __builtin_amdgcn_s_memtime(), fr0_ = __builtin_amdgcn_s_memrealtime();
#endif
  tile_init(T, a.net, lds, scratch + (size_t)blockIdx.x * SCR_F4_PER_WG);
  // f.mode = flow kernel (bits 0..7) | chains per workgroup (bits 8..; 0 = 16): with fewer tiles than CUs the launch gives every
  // workgroup 8 / 4 / 2 chains in the first rows of its tile and marks the other rows done (flow_live_rows)
  const int live = (f.mode >> 8) ? (f.mode >> 8) : 16, fmode = f.mode & 0xFF;
  const int b0 = blockIdx.x * live, g = T.g, c = T.c, wave = T.wave;
  int rsrc[4];                           // chain a row READS (rows without a chain: the last one of the workgroup)
#pragma unroll
  for (int i = 0; i < 4; ++i) rsrc[i] = b0 + (4 * g + i < live ? 4 * g + i : live - 1);
  float y[TPW][4], ell[4], vol0[4], lq_ref[4];
  int natt_tot[4];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = 16 * (wave + NW * q) + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[q][i] = f.pos[(size_t)rsrc[i] * D + col];                          // :267 / :251
  }
  fill_probe(T, a.z1, b0, live);         // key_hutch2: probe of the inverse solve (the forward probe is loaded per row)
  solve2<D, RP, PAD>(T, a, f, b0, y, live, fmode);    // inverse solve -> proposal -> forward solve, per row
  {
    const f32x4 e4 = T.rs_get(RS_ELL), v4 = T.rs_get(RS_VOL0), l4 = T.rs_get(RS_LQ), n0 = T.rs_get(RS_NTOT), n1 = T.rs_get(RS_NATT);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ell[i] = e4[i]; vol0[i] = v4[i]; lq_ref[i] = l4[i]; natt_tot[i] = (int)n0[i] + (int)n1[i]; }
  }
  // ---- target at the proposal (:270 / :252), tempered: beta * loglik (logprior = 0) ----
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) *T.at(T.o_xo, S::XB0 + i * LDX + 128 * q) = y[q][i];
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
        const float* xr = T.at(T.o_xo, S::XB0 + i * LDX + 128 * q) - col;      // row base: xr[col] is this lane's element
        if (!PAD || col < a.net.d) part[i] += phi4_term(a.net.T, xr, col);     // (PAD: columns >= d are the zero padding of a narrower lattice)
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
    const int b = rsrc[i];
    const Key2 kb = split_at(f.key, f.n_total, f.chain_offset + (uint32_t)b);             // :303
    const double lp_old = f.logp[b];
    const double la = lpn[i] - (double)ell[i] - lp_old - (double)vol0[i] + (double)lq_ref[i];
    const double ap = exp(la);
    const double u = uniform01(split_at(kb, 4, 1), 0, 1);
    acc[i] = u <= ap;                     // NaN compares false -> reject
    aprob[i] = (float)ap;
    if constexpr (RP) {
      if (a.rp.diag && wave == 0 && c == 0 && 4 * g + i < live) { double* o = a.rp.diag + 4 * (size_t)b; o[0] = vol0[i]; o[1] = ell[i]; o[2] = lpn[i]; o[3] = la; }
    }
  }
  __syncthreads();      // every wave has read the OLD log-densities before wave 0 publishes the accepted ones
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = 16 * (wave + NW * q) + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * g + i >= live) continue;
      const size_t o = (size_t)(b0 + 4 * g + i) * D + col;
      if (f.proposed) f.proposed[o] = y[q][i];
      if (acc[i]) { f.pos[o] = y[q][i]; f.grad[o] = gnew[q][i]; }
    }
  }
  if (wave == 0 && c == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * g + i >= live) continue;
      const int b = b0 + 4 * g + i;
      if (acc[i]) f.logp[b] = lpn[i];
      if (f.acc_prob) f.acc_prob[b] = aprob[i];
      if (f.accepted) f.accepted[b] = acc[i] ? 1 : 0;
      if (f.nsteps) f.nsteps[b] = natt_tot[i];
    }
  }
  // this tile is done: until the slowest tile of the launch finishes, fill the idle CU with the draws of the coming iterations
#ifdef MFM_STAMPS
  T.cyc_done = __builtin_amdgcn_s_memtime() - fc0_;
#endif
  if (nz.n_items > 0) noise_tail(nz, reinterpret_cast<volatile int*>(lds + S::RS));
#ifdef MFM_STAMPS
  if (g_flow_dbg && (threadIdx.x == 0 || threadIdx.x == 256)) {
    unsigned long long* o = g_flow_dbg + (blockIdx.x + (threadIdx.x ? gridDim.x : 0)) * 64;
    o[0] = __builtin_amdgcn_s_memtime() - fc0_; o[1] = __builtin_amdgcn_s_memrealtime() - fr0_;
    o[2] = T.n_eval; o[3] = T.cyc_eval; o[5] = T.n_tb; o[6] = T.cyc_tb;
    for (int i = 0; i < 20; ++i) o[8 + i] = T.cyc_sec[i];
    o[28] = T.n_tc; o[29] = T.cyc_tc; o[30] = T.n_ec; o[31] = T.cyc_ec;
    for (int i = 0; i < 16; ++i) o[32 + i] = T.cyc_csec[i];
    o[48] = T.n_em; o[49] = T.cyc_em; o[50] = T.n_t1; o[51] = T.cyc_t1; o[52] = T.cyc_tail; o[53] = T.cyc_done; o[60] = T.cyc_si; o[61] = T.cyc_n1; o[62] = T.cyc_n2; o[63] = T.cyc_n3;
  }
#endif
}

// ---- dispatch --------------------------------------------------------------------------------------------------
// The two instances (tile width D = 128, 256) also serve NARROWER lattices (d a multiple of 16 below D: the reference's own
// phi-four default is d = 64, multi_modal.py:52-55) by zero padding: x1 gets D - d zero input rows, the gate and the out layer D - d
// zero output columns (weights AND biases), so the padded columns of the state stay exactly zero through both solves (their field
// is 0 + 0 * clip(...)), contribute nothing to any norm, divergence or proposal, and the kernels only need the true d in three
// places: the error norm's 1 / (d + 1), the random-walk scale 2.38 / sqrt(d) and the target sum at the proposal.  The padded
// weight / bias pack is re-formed from the network's own pack before every launch (0.7 MB: microseconds per 101 iterations);
// chain states and probes are staged through [n][D] images.  The generic tile needs 85.9 k cycles per evaluation at these widths.
static int tile_width(const NetDev& n) { return n.d <= 128 ? 128 : 256; }
static bool shape_ok(const NetDev& n, int hutch) {
  if (!hutch || n.T.kind != MFM_TARGET_PHI4 || n.act != MFM_ACT_RELU) return false;
  if (n.nT != 2 || n.nX != 2 || n.nJ != 2 || net_ragged(n)) return false;
  if (n.F != F || n.ht1 != H || n.ht2 != H || n.hx1 != H || n.hx2 != H || n.hj1 != H || n.hj2 != H) return false;
  return n.d >= 16 && n.d <= 256 && n.d % 16 == 0;
}
static int max_wgs() { return ODE_FAST_MAX_WGS; }

// padded parameter pack of a narrower network for instance D: every layer with (true -> padded) shapes, from the network's own pack
template <int D>
__global__ void pad_pack_kernel(NetDev n, float* Wp, float* WpT, float* bias) {
  using S = FS<D>;
  const int Kpad[8] = {2 * F, H, D, H, H, 2 * H, H, H}, Npad[8] = {H, H, H, H, D, H, H, D};
  const int woff[8] = {S::W0, S::W1, S::W2, S::W3, S::W4, S::W5, S::W6, S::W7};
  const int boff[8] = {S::B0, S::B1, S::B2, S::B3, S::B4, S::B5, S::B6, S::B7};
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S::WTOT; e += gridDim.x * blockDim.x) {
    int l = 7;
#pragma unroll
    for (int j = 7; j >= 1; --j) if (e < woff[j]) l = j - 1;
    const int r = e - woff[l], k = r / Npad[l], nn = r - k * Npad[l];
    const LayerDesc& ld = n.L[l];
    const float w = (k < ld.Kp && nn < ld.Np) ? n.Wp[ld.w_off + pack_index(k, nn, ld.Kp / 16)] : 0.f;
    Wp[woff[l] + pack_index(k, nn, Kpad[l] / 16)] = w;
    if (l == 7) WpT[woff[7] + pack_index_T(k, nn, Npad[7] / 16)] = w;
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S::BTOT; e += gridDim.x * blockDim.x) {
    int l = 7;
#pragma unroll
    for (int j = 7; j >= 1; --j) if (e < boff[j]) l = j - 1;
    const int nn = e - boff[l];
    bias[e] = nn < n.L[l].Np ? n.bias[n.L[l].b_off + nn] : 0.f;
  }
}
// [n][d] -> [n][D] (zero columns) and back
__global__ void pad_rows_kernel(const float* src, float* dst, int n, int d, int D) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n * D; e += (size_t)gridDim.x * blockDim.x) {
    const size_t r = e / D; const int c = (int)(e - r * D);
    dst[e] = c < d ? src[r * d + c] : 0.f;
  }
}
__global__ void unpad_rows_kernel(const float* src, float* dst, int n, int d, int D) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n * d; e += (size_t)gridDim.x * blockDim.x) {
    const size_t r = e / d; const int c = (int)(e - r * d);
    dst[e] = src[r * D + c];
  }
}
static void pad_rows(const float* src, float* dst, int n, int d, int D, hipStream_t s) {
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(((size_t)n * D + 255) / 256 < 4096 ? ((size_t)n * D + 255) / 256 : 4096)), dim3(256), 0, s, src, dst, n, d, D);
}
static void unpad_rows(const float* src, float* dst, int n, int d, int D, hipStream_t s) {
  hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)(((size_t)n * d + 255) / 256 < 4096 ? ((size_t)n * d + 255) / 256 : 4096)), dim3(256), 0, s, src, dst, n, d, D);
}
// arguments of a launch on the padded images (the caller's stay as they are); false: no workspace for this shape / row count
template <int D>
static bool pad_args(OdeArgs& a, hipStream_t s) {
  if (!t_pad) return false;
  const PadWs& w = *t_pad;
  if (!w.Wp || w.D != D || (size_t)a.n > w.rows) return false;
  hipLaunchKernelGGL((pad_pack_kernel<D>), dim3(256), dim3(256), 0, s, a.net, w.Wp, w.WpT, w.bias);
  a.net.Wp = w.Wp; a.net.WpT = w.WpT; a.net.bias = w.bias;
  const int d = a.net.d;
  if (a.z1) { pad_rows(a.z1, w.img[3], a.n, d, D, s); a.z1 = w.img[3]; }
  if (a.z2) { pad_rows(a.z2, w.img[4], a.n, d, D, s); a.z2 = w.img[4]; }
  if (a.zgen) { pad_rows(a.zgen, w.img[5], a.n, d, D, s); a.zgen = w.img[5]; }
  return true;
}

template <int D, bool RP>
static int launch_flow_tr(const OdeArgs& a0, const FlowArgs& f0, const NoiseArgs& nz, f32x4* scratch, hipStream_t stream) {
  const size_t sm = (size_t)FS<D>::TOTAL * sizeof(float);
  (void)hipFuncSetAttribute((const void*)flow_step_fast_kernel<D, RP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  OdeArgs a = a0; FlowArgs f = f0;
  const int d = a.net.d;
  if (d != D) {
    if (!pad_args<D>(a, stream)) return -3;
    const PadWs& w = *t_pad;
    pad_rows(f0.pos, w.img[0], a.n, d, D, stream); pad_rows(f0.grad, w.img[1], a.n, d, D, stream);
    f.pos = w.img[0]; f.grad = w.img[1]; f.proposed = f0.proposed ? w.img[2] : nullptr;
  }
  const int live = flow_live_rows(a.n);
  f.mode = (f0.mode & 0xFF) | (live == 16 ? 0 : live << 8);
  if (d != D) {
    (void)hipFuncSetAttribute((const void*)flow_step_fast_kernel<D, RP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((flow_step_fast_kernel<D, RP, true>), dim3(a.n / live), dim3(NW * 64), sm, stream, a, f, nz, scratch);
  } else {
    hipLaunchKernelGGL((flow_step_fast_kernel<D, RP, false>), dim3(a.n / live), dim3(NW * 64), sm, stream, a, f, nz, scratch);
  }
  if (d != D) {
    unpad_rows(f.pos, f0.pos, a.n, d, D, stream); unpad_rows(f.grad, f0.grad, a.n, d, D, stream);
    if (f0.proposed) unpad_rows(f.proposed, f0.proposed, a.n, d, D, stream);
  }
  return 0;
}
template <int D>
static int launch_flow_t(const OdeArgs& a, const FlowArgs& f, const NoiseArgs& nz, f32x4* scratch, hipStream_t stream) {
  return a.rp.dt ? launch_flow_tr<D, true>(a, f, nz, scratch, stream) : launch_flow_tr<D, false>(a, f, nz, scratch, stream);
}
template <int D, bool RP>
static int launch_transform_tr(const OdeArgs& a0, f32x4* scratch, hipStream_t stream) {
  const size_t sm = (size_t)FS<D>::TOTAL * sizeof(float);
  (void)hipFuncSetAttribute((const void*)ode_transform_fast_kernel<D, RP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  OdeArgs a = a0;
  const int d = a.net.d;
  if (d != D) {
    if (!pad_args<D>(a, stream)) return -3;
    pad_rows(a0.in, t_pad->img[0], a.n, d, D, stream);
    a.in = t_pad->img[0]; a.out = t_pad->img[2];
  }
  const int tiles = a.n / 16, grid = tiles < max_wgs() ? tiles : max_wgs();
  if (d != D) {
    (void)hipFuncSetAttribute((const void*)ode_transform_fast_kernel<D, RP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((ode_transform_fast_kernel<D, RP, true>), dim3(grid), dim3(NW * 64), sm, stream, a, scratch);
    unpad_rows(a.out, a0.out, a.n, d, D, stream);
  } else {
    hipLaunchKernelGGL((ode_transform_fast_kernel<D, RP, false>), dim3(grid), dim3(NW * 64), sm, stream, a, scratch);
  }
  return 0;
}
template <int D>
static int launch_transform_t(const OdeArgs& a, f32x4* scratch, hipStream_t stream) {
  return a.rp.dt ? launch_transform_tr<D, true>(a, scratch, stream) : launch_transform_tr<D, false>(a, scratch, stream);
}

}  // namespace fast
