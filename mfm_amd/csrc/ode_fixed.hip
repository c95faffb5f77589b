// NS1: the FIXED-STEP mode of the CNF solver on the shape-specialised tile (ode_fast.hip) -- classical RK4 or forward Euler on N equal
// steps of [0, 1].  NOT in the reference, which integrates with the adaptive Dopri5 of jax.experimental.ode.odeint
// (exe_flow_matching.py:345-349: every other solver path of this library restates that); BASELINE.json's north star names an
// "RK4/Euler ODE integrator", and this is that mode: every chain of every tile takes the same N steps, so there is no step-size
// controller, no initial-step heuristic, no dense output, no per-row solve phases and no lock-step tail -- a tile's run time does not
// depend on its slowest chain.  Same field evaluation (FTile::eval: x branch, Hutchinson integrand, target terms) and the same time
// batch (FTile::tbatch: five stage times per M = 80 GEMM chain) as the adaptive kernels:
//   RK4   : one batch serves TWO steps (times t, t + h/2, t + h, t + 3h/2, t + 2h), eight evaluations;
//   Euler : one batch serves FIVE steps, five evaluations.
// Oracle: oracle/ode.py: odeint_fixed (float64, checked against closed forms, scipy and the oracle's own Dopri5); parity:
// tests/test_gpu_fixed.py.  mfm_config.ode_method / ode_steps select it (include/mfm.h); the Hutchinson log-det, the flow-MH
// acceptance (exe_flow_matching.py:264-278) and every draw are those of the adaptive path.
namespace fast {

enum { ODE_RK4 = 1, ODE_EULER = 2 };

template <int D, int METHOD>
__device__ __forceinline__ void solve_fixed(FTile<D>& T, int nsteps, float (&y)[FTile<D>::TPW][4], float (&ell)[4]) {
  using S = FS<D>;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX;
  constexpr int NK = METHOD == ODE_RK4 ? 4 : 1, SPB = METHOD == ODE_RK4 ? 2 : 5;      // stages per step, steps per time batch
  const int wave = T.wave;
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
  const float h = 1.f / (float)nsteps;
  const f32x4 sg = f32x4{(float)T.sign, (float)T.sign, (float)T.sign, (float)T.sign};
  float k[NK][TPW][4];
  float el[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  auto write_x = [&](const float (&x)[TPW][4]) {
    const int xsel = cur ? S::XB1 * 4 : S::XB0 * 4;
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) *T.at(T.o_xo + xsel, i * LDX + 128 * q) = x[q][i];
  };
  // log-det of a finished step from the per-wave divergence partials of its stages (DLP slots 4 par .. 4 par + NK - 1); read after a barrier
  auto add_ell = [&](int par) {
    float l[NK][4];
#pragma unroll
    for (int j = 0; j < NK; ++j) T.part_get(S::DLP + (4 * par + j) * 128, l[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      el[i] += METHOD == ODE_RK4 ? (h / 6.f) * ((l[0][i] + l[NK - 1][i]) + 2.f * (l[NK > 1 ? 1 : 0][i] + l[NK > 2 ? 2 : 0][i])) : h * l[0][i];
  };
  int par = 0; bool pending = false;
#pragma unroll 1
  for (int n0 = 0; n0 < nsteps; n0 += SPB) {
    // ---- Fourier features of the batch's five stage times (:70-71): slot s is t = (n0 + s / 2) h (RK4) or (n0 + s) h (Euler) ----
    float cv[5][4], sv[5][4];
    {
      const double f = (double)T.ffreq;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const float tt = METHOD == ODE_RK4 ? ((float)(2 * n0 + s) * 0.5f) * h : (float)(n0 + s) * h;
        const double te = T.sign > 0 ? (double)tt : 1.0 - (double)tt;          // :229
        double ft = f * te;
        ft -= rint(ft);
        float sn, cs;
        sincospi_half_turn(2.f * (float)ft, &sn, &cs);
#pragma unroll
        for (int i = 0; i < 4; ++i) { cv[s][i] = cs; sv[s][i] = sn; }
      }
    }
    write_x(y);                            // the first stage input of the batch; tbatch's barriers make it visible
    T.tbatch(2, P, Q, cv, sv);
    if (pending) { add_ell(par ^ 1); pending = false; }
#pragma unroll 1
    for (int ss = 0; ss < SPB && n0 + ss < nsteps; ++ss) {
      const bool last_of_batch = ss + 1 == SPB || n0 + ss + 1 == nsteps;
      if (ss > 0) {
        write_x(y);
        __syncthreads();
        if (pending) { add_ell(par ^ 1); pending = false; }
      }
      float kv[TPW][4];
      if constexpr (METHOD == ODE_EULER) {
        T.eval(ss, cur, 4 * par, last_of_batch, P, Q, kv, sg);
        cur ^= 1;
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) y[q][i] = __builtin_fmaf(h, kv[q][i], y[q][i]);
      } else {
        float x[TPW][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // stage j at slot 2 ss + {0, 1, 1, 2}, input y + {0, h/2 k1, h/2 k2, h k3}
          if (j > 0) {
            const float a = j == 3 ? h : 0.5f * h;
#pragma unroll
            for (int q = 0; q < TPW; ++q)
#pragma unroll
              for (int i = 0; i < 4; ++i) x[q][i] = __builtin_fmaf(a, k[j - 1][q][i], y[q][i]);
            write_x(x);
            __syncthreads();
          }
          T.eval(2 * ss + (j == 0 ? 0 : (j == 3 ? 2 : 1)), cur, 4 * par + j, last_of_batch && j == 3, P, Q, kv, sg);
          cur ^= 1;
#pragma unroll
          for (int q = 0; q < TPW; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) k[j][q][i] = kv[q][i];
        }
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            y[q][i] = __builtin_fmaf(h / 6.f, (k[0][q][i] + k[3][q][i]) + 2.f * (k[1][q][i] + k[2][q][i]), y[q][i]);
      }
      pending = true; par ^= 1;            // this step's log-det increment is added after the next barrier
    }
  }
  __syncthreads();
  if (pending) add_ell(par ^ 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) ell[i] = el[i];
}

// CNF transform / inverse with log-det on N fixed steps (exe_flow_matching.py:206-242 with the integrator swapped)
template <int D, int METHOD, bool PAD = false>
__global__ __launch_bounds__(NW * 64) void ode_transform_fixed_kernel(OdeArgs a, int nsteps, f32x4* scratch) {
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
    float y[TPW][4], ell[4];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int col = 16 * (T.wave + NW * q) + T.c;
#pragma unroll
      for (int i = 0; i < 4; ++i) y[q][i] = a.in[(size_t)(b0 + 4 * T.g + i) * D + col];
    }
    solve_fixed<D, METHOD>(T, nsteps, y, ell);
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
        if (a.nsteps) a.nsteps[b0 + 4 * T.g + i] = nsteps;
      }
    }
  }
}

// One random-walk flow-MH step per chain (exe_flow_matching.py:264-278) with both solves on N fixed steps: inverse solve ->
// latent proposal -> forward solve, tile-wide (every row is in the same phase), then the target at the proposal and the accept
// step exactly as flow_step_fast_kernel.
template <int D, int METHOD, bool PAD = false>
__global__ __launch_bounds__(NW * 64) void flow_step_fixed_kernel(OdeArgs a, FlowArgs f, int nsteps, f32x4* scratch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using S = FS<D>;
  constexpr int TPW = FTile<D>::TPW, LDX = S::LDX;
  FTile<D> T;
  tile_init(T, a.net, lds, scratch + (size_t)blockIdx.x * SCR_F4_PER_WG);
  const int b0 = blockIdx.x * 16, g = T.g, c = T.c, wave = T.wave;
  float y[TPW][4], ell[4], vol0[4];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = 16 * (wave + NW * q) + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[q][i] = f.pos[(size_t)(b0 + 4 * g + i) * D + col];              // :267
  }
  fill_probe(T, a.z1, b0);                 // key_hutch2
  T.sign = -1;
  solve_fixed<D, METHOD>(T, nsteps, y, vol0);                                                      // :267 inverse_and_logdet
  const float scale = 2.38f / sqrtf((float)(PAD ? a.net.d : D));                                   // :262
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = 16 * (wave + NW * q) + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[q][i] = y[q][i] + scale * a.zgen[(size_t)(b0 + 4 * g + i) * D + col];      // :268
  }
  __syncthreads();
  fill_probe(T, a.z2, b0);                 // key_hutch1
  T.sign = 1;
  solve_fixed<D, METHOD>(T, nsteps, y, ell);                                                       // :269 transform_and_logdet
  // ---- target at the proposal (:270), tempered: beta * loglik (logprior = 0) ----
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
        const float* xr = T.at(T.o_xo, S::XB0 + i * LDX + 128 * q) - col;
        if (!PAD || col < a.net.d) part[i] += phi4_term(a.net.T, xr, col);
        gnew[q][i] = (float)f.beta * phi4_grad(a.net.T, xr, col);
      }
    }
    double* rd = reinterpret_cast<double*>(lds + S::RED);
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
  // ---- accept / reject (:271-278); the acceptance probability is NOT clipped ----
  bool acc[4];
  float aprob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = b0 + 4 * g + i;
    const Key2 kb = split_at(f.key, f.n_total, f.chain_offset + (uint32_t)b);                      // :303
    const double la = lpn[i] - (double)ell[i] - f.logp[b] - (double)vol0[i];
    const double ap = exp(la);
    const double u = uniform01(split_at(kb, 4, 1), 0, 1);
    acc[i] = u <= ap;
    aprob[i] = (float)ap;
    if (a.rp.diag && wave == 0 && c == 0) { double* o = a.rp.diag + 4 * (size_t)b; o[0] = vol0[i]; o[1] = ell[i]; o[2] = lpn[i]; o[3] = la; }
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
      if (f.nsteps) f.nsteps[b] = 2 * nsteps;
    }
  }
}

template <int D, int METHOD>
static int launch_flow_fixed_m(const OdeArgs& a0, const FlowArgs& f0, int nsteps, f32x4* scratch, hipStream_t stream) {
  const size_t sm = (size_t)FS<D>::TOTAL * sizeof(float);
  OdeArgs a = a0; FlowArgs f = f0;
  const int d = a.net.d;
  if (d != D) {
    if (!pad_args<D>(a, stream)) return -3;
    const PadWs& w = *t_pad;
    pad_rows(f0.pos, w.img[0], a.n, d, D, stream); pad_rows(f0.grad, w.img[1], a.n, d, D, stream);
    f.pos = w.img[0]; f.grad = w.img[1]; f.proposed = f0.proposed ? w.img[2] : nullptr;
    (void)hipFuncSetAttribute((const void*)flow_step_fixed_kernel<D, METHOD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((flow_step_fixed_kernel<D, METHOD, true>), dim3(a.n / 16), dim3(NW * 64), sm, stream, a, f, nsteps, scratch);
    unpad_rows(f.pos, f0.pos, a.n, d, D, stream); unpad_rows(f.grad, f0.grad, a.n, d, D, stream);
    if (f0.proposed) unpad_rows(f.proposed, f0.proposed, a.n, d, D, stream);
  } else {
    (void)hipFuncSetAttribute((const void*)flow_step_fixed_kernel<D, METHOD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((flow_step_fixed_kernel<D, METHOD, false>), dim3(a.n / 16), dim3(NW * 64), sm, stream, a, f, nsteps, scratch);
  }
  return 0;
}
template <int D, int METHOD>
static int launch_transform_fixed_m(const OdeArgs& a0, int nsteps, f32x4* scratch, hipStream_t stream) {
  const size_t sm = (size_t)FS<D>::TOTAL * sizeof(float);
  OdeArgs a = a0;
  const int d = a.net.d;
  const int tiles = a.n / 16, grid = tiles < max_wgs() ? tiles : max_wgs();
  if (d != D) {
    if (!pad_args<D>(a, stream)) return -3;
    pad_rows(a0.in, t_pad->img[0], a.n, d, D, stream);
    a.in = t_pad->img[0]; a.out = t_pad->img[2];
    (void)hipFuncSetAttribute((const void*)ode_transform_fixed_kernel<D, METHOD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((ode_transform_fixed_kernel<D, METHOD, true>), dim3(grid), dim3(NW * 64), sm, stream, a, nsteps, scratch);
    unpad_rows(a.out, a0.out, a.n, d, D, stream);
  } else {
    (void)hipFuncSetAttribute((const void*)ode_transform_fixed_kernel<D, METHOD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((ode_transform_fixed_kernel<D, METHOD, false>), dim3(grid), dim3(NW * 64), sm, stream, a, nsteps, scratch);
  }
  return 0;
}
static int launch_flow_fixed(const OdeArgs& a, const FlowArgs& f, int method, int nsteps, f32x4* scratch, hipStream_t stream) {
  const bool w256 = tile_width(a.net) == 256;
  if (method == ODE_RK4) return w256 ? launch_flow_fixed_m<256, ODE_RK4>(a, f, nsteps, scratch, stream) : launch_flow_fixed_m<128, ODE_RK4>(a, f, nsteps, scratch, stream);
  return w256 ? launch_flow_fixed_m<256, ODE_EULER>(a, f, nsteps, scratch, stream) : launch_flow_fixed_m<128, ODE_EULER>(a, f, nsteps, scratch, stream);
}
static int launch_transform_fixed(const OdeArgs& a, int method, int nsteps, f32x4* scratch, hipStream_t stream) {
  const bool w256 = tile_width(a.net) == 256;
  if (method == ODE_RK4) return w256 ? launch_transform_fixed_m<256, ODE_RK4>(a, nsteps, scratch, stream) : launch_transform_fixed_m<128, ODE_RK4>(a, nsteps, scratch, stream);
  return w256 ? launch_transform_fixed_m<256, ODE_EULER>(a, nsteps, scratch, stream) : launch_transform_fixed_m<128, ODE_EULER>(a, nsteps, scratch, stream);
}

}  // namespace fast
