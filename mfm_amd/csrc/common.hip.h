// Shared device/host helpers for the MFM HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Storage type of the Gaussian draws of the MALA step and of the flow-matching batch in the fused-tile family (mala.hip, fm.hip,
// noise.hip): drawn in float64 (threefry + erfinv, as the reference under jax_enable_x64), ROUNDED to this type at the point of use,
// whether they were produced ahead of time (noise.hip) or in line -- the chain positions they are added to are float32, and the
// prefetched draws are the largest HBM stream of the MALA + training iteration (float64: 6 KB of its 8 KB per chain).
typedef float draw_t;

// section time stamps of the training kernel and of the MALA step inside it (development build only: tools/fm_stamps.py)
#ifdef MFM_FM_STAMPS
__device__ unsigned long long* g_fm_dbg = nullptr;      // [WG][32] section time stamps (development build only)
#define FM_STAMP(id) do { if (g_fm_dbg && threadIdx.x == 0) g_fm_dbg[blockIdx.x * 32 + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FM_STAMP(id) do {} while (0)
#endif

#define MFM_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- development / A-B switches (environment) -----------------------------------------------------------------------------
// Read ONCE PER mfm_create (api.hip: create_impl) and fixed from then until the next context is created: a switch flipped in
// the environment between two calls on the same context has no effect, one flipped before mfm_create always does.  (They used
// to be a mix of per-call getenv and function-local statics latched by the first call of the process, so an in-process A/B
// could silently run the same arm twice.)  One context per process and GPU is the deployment model (include/mfm.h).
#include <cstdlib>
struct Switches {
  bool no_fused_opt = false, force_exchange = false, no_fused_mala = false, generic_ode = false, generic_fm = false;
  bool eval16 = false, eval_no_chain = false;
  int eval_rows = 32, eval_stagger = 30000;
  int d2_tile = 0;          // 0: automatic; 16: generic tile; 4: 4-chain tiles; 5: "4s", the streamed 4-chain tile
  bool wide_nolds = false, wide_no_small_tiles = false, wide_wm1 = false, wide_ring8 = false, wide_wgrad_nosplit = false;
  bool wide_nocompact = false, wide_no_tbatch = false;
  bool tile_exact = false;  // exact-trace solves of the fused family on its own generic tile instead of the wide family's solver (api.hip: wide_ex)
  bool wsk_xcd = false;     // wgrad_sk.hip: consecutive unit ranges on one XCD instead of workgroup w = blockIdx.x (A/B, with MFM_WSK_G)
  bool rccl_comm_stream = false;   // api.hip: the gradient all-reduce of an mfm_adamw_step that has none in flight goes through the communication stream
                                   // (two event hops) instead of in line on the context's own stream (A/B)
  int flow_live = 0;        // chains per workgroup of the shape-specialised flow step: 0 automatic, 16 / 8 / 4 / 2 forced (ode_fast.hip: flow_live_rows)
};
static Switches g_sw;
static void switches_read() {
  Switches s;
  auto on = [](const char* k) { return getenv(k) != nullptr; };
  s.no_fused_opt = on("MFM_NO_FUSED_OPT"); s.force_exchange = on("MFM_DEBUG_FORCE_EXCHANGE"); s.no_fused_mala = on("MFM_NO_FUSED_MALA");
  s.generic_ode = on("MFM_GENERIC_ODE"); s.generic_fm = on("MFM_GENERIC_FM");
  s.eval16 = on("MFM_EVAL16"); s.eval_no_chain = on("MFM_EVAL_NO_CHAIN");
  if (const char* e = getenv("MFM_EVAL_ROWS")) s.eval_rows = atoi(e);
  if (const char* e = getenv("MFM_EVAL_STAGGER")) s.eval_stagger = atoi(e);
  if (const char* e = getenv("MFM_D2_TILE")) s.d2_tile = (e[0] == '4' && e[1] == 's') ? 5 : atoi(e);
  s.wide_nolds = on("MFM_WIDE_NOLDS"); s.wide_no_small_tiles = on("MFM_WIDE_NO_SMALL_TILES"); s.wide_wm1 = on("MFM_WIDE_WM1");
  s.wide_ring8 = on("MFM_WIDE_RING8"); s.wide_wgrad_nosplit = on("MFM_WIDE_WGRAD_NOSPLIT");
  s.wide_nocompact = on("MFM_WIDE_NOCOMPACT"); s.wide_no_tbatch = on("MFM_WIDE_NO_TBATCH");
  if (const char* e = getenv("MFM_FLOW_LIVE")) s.flow_live = atoi(e);
  s.tile_exact = on("MFM_TILE_EXACT"); s.wsk_xcd = on("MFM_WSK_XCD"); s.rccl_comm_stream = on("MFM_RCCL_COMM_STREAM");
  g_sw = s;
}

// ---- wave-level reductions (64 lanes) -----------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// the same sum by four DPP adds inside the 16-lane row (quad butterflies, then half-row mirror and row mirror once the
// quads / halves are uniform), ~40 cycles: __shfl_xor lowers to ds_bpermute_b32, i.e. four dependent LDS round trips
// (~500 cycles, measured in the solver's out-layer epilogue).  The association order of the adds differs from
// group16_sum, so the two are not bit-identical; used by the shape-specialised solver (ode_fast.hip).
__device__ __forceinline__ float group16_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

__device__ __forceinline__ float group16_max_dpp(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  return v;
}

__host__ __device__ __forceinline__ int ceil16(int x) { return (x + 15) & ~15; }
