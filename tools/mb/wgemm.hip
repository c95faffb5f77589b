// Timing of the wide family's layer GEMM (wide::gemm_kernel through wide::launch_gemm) and weight-gradient kernel at the
// pines shapes, outside the library (development aid; values are random, only durations matter).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mb/wgemm tools/mb/wgemm.hip && tools/mb/wgemm
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "../../mfm_amd/csrc/wide.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
static float* dev_rand(size_t n) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}
int main() {
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int shapes[][4] = {{1024, 1024, 1024, 0}, {1024, 2048, 1024, 0}, {1024, 256, 1024, 0}, {1024, 1024, 1024, 1}, {1024, 2048, 1024, 1},
                           {4096, 1024, 1024, 0}, {1024, 1024, 2048, 0}};
  for (auto& sh : shapes) {
    const int rows = sh[0], K = sh[1], N = sh[2], dual = sh[3];
    wide::Gemm g; memset(&g, 0, sizeof g);
    g.W = dev_rand((size_t)K * N); g.KB = K / 16; g.NT = N / 16; g.bias = dev_rand(N);
    const int pad = getenv("XPAD") ? atoi(getenv("XPAD")) : 0;
    g.X = dev_rand((size_t)rows * (K + pad)); g.ldx = K + pad; g.rows = rows; g.act = 1;
    g.Y = dev_rand((size_t)rows * N); g.ldy = N;
    if (dual) { g.XT = dev_rand((size_t)rows * (K + pad)); g.KBT = (dual && K == 2048) ? K / 32 : K / 16; g.YT = dev_rand((size_t)rows * N); }
    for (int i = 0; i < 20; ++i) wide::launch_gemm(g, s);
    const int it = 200;
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < it; ++i) wide::launch_gemm(g, s);
    (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * rows * N * (K + (dual ? g.KBT * 16 : 0));
#ifdef WIDE_DBG_CLOCK
    unsigned long long clk[4]; (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(wide::wide_dbg_clk), sizeof clk);
    printf("   main loop of workgroup 0: %llu shader cycles in %llu x 10 ns = %.3f GHz; %.1f us\n", clk[0], clk[1], clk[0] / (clk[1] * 10.0), clk[1] * 0.01);
#endif
    printf("gemm rows %d K %d N %d dual %d: %.2f us per launch (back to back), %.1f TFLOP/s = %.0f %% of 157.3\n", rows, K, N, dual, ms * 1e3 / it,
           fl / (ms * 1e-3 / it) * 1e-12, fl / (ms * 1e-3 / it) * 1e-12 / 157.3 * 100);
  }
  return 0;
}
