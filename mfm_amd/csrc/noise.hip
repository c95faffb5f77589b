// Random draws of the MALA + training iterations of a whole (K + 1)-cycle, produced AHEAD of time in the tail of the flow
// step: a flow step lasts as long as its slowest chain while the mean tile needs ~70 % of that, so most workgroups of the
// persistent flow kernel finish early and their CUs would idle.  Instead, a workgroup whose tile is done pulls items of
// this work from a global counter until none is left (noise_tail, called at the end of flow_step_fast_kernel): the tail is
// filled by exactly the CUs that are free, with no second stream and no scheduling assumptions.  The float64
// threefry + erfinv work is ~1/4 of a MALA + training iteration (mala_step 20.6 -> 11 us, fm_fwd_bwd 77 -> 63 us).
// The draws are the ones the kernels would make in line (same keys, same counters, float64 arithmetic, rounded to `draw_t` -- float32 --
// as the consuming kernels round theirs: common.hip.h), so results are bit-identical with and without the prefetch.  Slot j holds the draws of the MALA step keyed gn[j] (mala.py:93, util.py:80-82,
// proposal.py:179) and of the flow-matching batch keyed st[j] (exe_flow_matching.py:153-155,166).
#pragma once
#include "common.hip.h"
#include "prng.hip.h"

struct NoiseArgs {
  const uint32_t* gn; const uint32_t* st;     // [n_slots][2] device copies of the keys
  int n_slots; uint32_t n_total, chain_offset; int B, d;
  draw_t* mala_n; double* mala_u;             // [slot][B][d], [slot][B]
  draw_t* fm_x0; draw_t* fm_eps; float* fm_t; // [slot][B][d] x 2, [slot][B]
  int* counter; int n_items, groups;          // work items of 8 chains (one wavefront each): item = slot * groups + group
  int skip_mala0;                             // slot 0 is keyed by the flow step that produces it (its own iteration's training batch): nobody
                                              // will ask for that key's MALA draws
};

// all draws of chain row b in slot `slot`, by one wavefront
__device__ __forceinline__ void noise_row(const NoiseArgs& a, int slot, int b, int lane) {
  const uint32_t bg = a.chain_offset + (uint32_t)b, d = (uint32_t)a.d;
  const size_t row = ((size_t)slot * a.B + b) * a.d, one = (size_t)slot * a.B + b;
  if (!(a.skip_mala0 && slot == 0)) {
    const Key2 kg{a.gn[2 * slot], a.gn[2 * slot + 1]};
    const Key2 kb = split_at(kg, a.n_total, bg);                                  // exe_flow_matching.py:303
    const Key2 k_int = split_at(kb, 2, 0), k_rmh = split_at(kb, 2, 1);            // mala.py:93
    for (int j = lane; j < a.d; j += 64) a.mala_n[row + j] = (draw_t)normal64(k_int, (uint32_t)j, d);      // util.py:80-82
    if (lane == 0) a.mala_u[one] = uniform01(k_rmh, 0, 1);                        // proposal.py:179
  }
  {
    const Key2 ks{a.st[2 * slot], a.st[2 * slot + 1]};
    const Key2 key_time = split_at(ks, 4, 0), key_ref = split_at(ks, 4, 1), key_gauss = split_at(ks, 4, 2);   // :153
    const Key2 kref = split_at(key_ref, a.n_total, bg);                           // :155
    for (int j = lane; j < a.d; j += 64) {
      a.fm_x0[row + j] = (draw_t)normal64(kref, (uint32_t)j, d);
      a.fm_eps[row + j] = (draw_t)normal64(key_gauss, bg * d + (uint32_t)j, a.n_total * d);               // :166
    }
    if (lane == 0) a.fm_t[one] = (float)uniform01(key_time, bg, a.n_total);       // :154
  }
}

// Called by every thread of an 8-wave workgroup once its own work is done.  `slot_word` is an LDS word the workgroup no
// longer uses.  Every workgroup leaves when the counter passes n_items, so the grid always drains.
__device__ __noinline__ void noise_tail(const NoiseArgs& a, volatile int* slot_word) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) *slot_word = atomicAdd(a.counter, 1);
    __syncthreads();
    const int item = *slot_word;
    if (item >= a.n_items) break;
    const int slot = item / a.groups, b = (item - slot * a.groups) * 8 + wave;
    if (b < a.B) noise_row(a, slot, b, lane);
  }
}
