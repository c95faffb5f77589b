"""Development aid: instrument mlp.hip.h's layer_gemm with four time stamps per tile and wave (entry, first weight group arrived,
last MFMA issued, epilogue done) for tools/fm_stamps.py --lg.  Usage:
    python tools/dbg/lg_patch.py && bash tools/build_variant.sh lgdbg -DMFM_FM_STAMPS; git checkout mfm_amd/csrc/{mlp.hip.h,fm.hip,api.hip}
(the patch is never committed: it adds a file-scope __shared__ counter and waits that the product build must not carry)."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def sub(path, old, new):
    p = os.path.join(R, path); s = open(p).read(); assert old in s, (path, old[:60]); open(p, "w").write(s.replace(old, new, 1))
sub("mfm_amd/csrc/mlp.hip.h", '''template <int MT, int NW, int PIPE = 2, typename Epi>
__device__ __forceinline__ void layer_gemm(''', '''__device__ unsigned long long* g_lg_dbg = nullptr;      // [WG][8 waves][128]: 4 stamps per layer_gemm tile, in call order
__shared__ int lg_cnt[8];
#define LG_STAMP(k) do { if (g_lg_dbg && (threadIdx.x & 63) == 0) { const int w_ = threadIdx.x >> 6; g_lg_dbg[((size_t)blockIdx.x * 8 + w_) * 128 + lg_cnt[w_] * 4 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
template <int MT, int NW, int PIPE = 2, typename Epi>
__device__ __forceinline__ void layer_gemm(''')
sub("mfm_amd/csrc/mlp.hip.h", '''    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE == 2) {''', '''    LG_STAMP(0);
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE == 2) {''')
sub("mfm_amd/csrc/mlp.hip.h", '''      __builtin_amdgcn_sched_barrier(0);
      group(ba, 4 * gi);
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 2 < KG) {''', '''      __builtin_amdgcn_sched_barrier(0);
      if (gi == 0) { asm volatile("s_waitcnt vmcnt(4)"); LG_STAMP(1); }
      group(ba, 4 * gi);
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 2 < KG) {''')
sub("mfm_amd/csrc/mlp.hip.h", '''#pragma unroll
    for (int m = 0; m < MT; ++m) epi(q, nt, m, acc[m], bv);
  }
}''', '''    LG_STAMP(2);
#pragma unroll
    for (int m = 0; m < MT; ++m) epi(q, nt, m, acc[m], bv);
    LG_STAMP(3);
    if ((threadIdx.x & 63) == 0 && lg_cnt[threadIdx.x >> 6] < 31) lg_cnt[threadIdx.x >> 6]++;
  }
}''')
# the chained variant (layer_gemm_chain): same four stamps
sub("mfm_amd/csrc/mlp.hip.h", '''    const int KG = KB >> 2;                 // even
    f32x4 bb[4];
    if (!ch.have) {''', '''    const int KG = KB >> 2;                 // even
    f32x4 bb[4];
    LG_STAMP(0);
    if (!ch.have) {''')
sub("mfm_amd/csrc/mlp.hip.h", '''      __builtin_amdgcn_sched_barrier(0);
      group(ch.b, 4 * gi);''', '''      __builtin_amdgcn_sched_barrier(0);
      if (gi == 0) { asm volatile("s_waitcnt vmcnt(4)"); LG_STAMP(1); }
      group(ch.b, 4 * gi);''')
sub("mfm_amd/csrc/mlp.hip.h", '''    ch.have = ntile != nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) epi(q, nt, m, acc[m], bv);
  }
}''', '''    ch.have = ntile != nullptr;
    LG_STAMP(2);
#pragma unroll
    for (int m = 0; m < MT; ++m) epi(q, nt, m, acc[m], bv);
    LG_STAMP(3);
    if ((threadIdx.x & 63) == 0 && lg_cnt[threadIdx.x >> 6] < 31) lg_cnt[threadIdx.x >> 6]++;
  }
}''')
sub("mfm_amd/csrc/fm.hip", '''  FM_STAMP(0);
  // ---------------- prologue: K3 batch construction''', '''  if (threadIdx.x < 8) lg_cnt[threadIdx.x] = 0;
  __syncthreads();
  FM_STAMP(0);
  // ---------------- prologue: K3 batch construction''')
sub("mfm_amd/csrc/api.hip", '#ifdef MFM_FM_STAMPS\nextern "C" int mfm_debug_fm_buffer', 'extern "C" int mfm_debug_lg_buffer(unsigned long long* d_buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_lg_dbg), &d_buf, sizeof d_buf) == hipSuccess ? 0 : MFM_EHIP; }\n#ifdef MFM_FM_STAMPS\nextern "C" int mfm_debug_fm_buffer')
print("patched")
