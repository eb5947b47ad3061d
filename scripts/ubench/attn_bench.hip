// Stand-alone timing harness of the 32x32x16 attention forward (diagnostic): the kernel source is compiled straight in,
// optionally with an ablation mask, so that one GPU call can time many variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I scenesplat_amd/csrc [-DFA32_ABL=m] \
//         scripts/ubench/attn_bench.hip -o attn_bench
// FA32_ABL bits: 1 no global loads / LDS staging writes, 2 no barrier, 4 exp -> mul, 8 no max / rescale,
//                16 no PV MFMAs, 32 no QK^T MFMAs.   -DFA32_CLOCK reads the in-kernel clock stamps.
//                (ablation results are wrong by construction; only the time is read)
#include "../../scenesplat_amd/csrc/attention_mfma32.hip"
#define FWD_LAUNCH ss_attn_fwd_mfma32
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

int main(int argc, char** argv) {
  const int W = 100, K = 1024, H = 16, D = argc > 1 ? atoi(argv[1]) : 48, C = H * D;
  const int64_t n = (int64_t)W * K;
  std::vector<unsigned short> q((size_t)n * 3 * C);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : q) { float f = nd(rng); unsigned int u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  std::vector<int32_t> gidx(n), win(W + 1);
  std::iota(gidx.begin(), gidx.end(), 0);
  std::shuffle(gidx.begin(), gidx.end(), rng);          // rows in memory order != window order, as on the real path
  for (int w = 0; w <= W; ++w) win[w] = w * K;
  unsigned short *dq, *dout; int32_t *dg, *dw; float* dlse;
  hipMalloc(&dq, q.size() * 2); hipMalloc(&dout, (size_t)n * C * 2); hipMalloc(&dg, n * 4); hipMalloc(&dw, (W + 1) * 4);
  hipMalloc(&dlse, (size_t)n * H * 4);
  hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dg, gidx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, win.data(), (W + 1) * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const float scale = 1.f / sqrtf((float)D);
  for (int i = 0; i < 3; ++i) FWD_LAUNCH(dq, dg, dg, dw, W, K, dout, dlse, C, H, scale, 0);
  hipDeviceSynchronize();
  const int iters = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) FWD_LAUNCH(dq, dg, dg, dw, W, K, dout, dlse, C, H, scale, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double fl = (double)W * H * 4.0 * K * K * D;
  printf("attn fwd32 d=%d ABL=%d: %.3f ms  %.0f TFLOP/s (%.1f %% of 2.5 PF)\n", D, (int)FA32_ABL, ms, fl / ms / 1e9, fl / ms / 1e9 / 25.0);
#ifdef FA32_CLOCK
  {
    // keep the chip loaded for ~2 s first (the clock the chip HOLDS under this kernel), then read the stamps of the last launch
    for (int i = 0; i < 4000; ++i) FWD_LAUNCH(dq, dg, dg, dw, W, K, dout, dlse, C, H, scale, 0);
    hipDeviceSynchronize();
    const int nwg = std::min(32768, W * H * ((K + FA32_BQ - 1) / FA32_BQ));
    std::vector<unsigned long long> st(4 * nwg);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(fa32_stamp), st.size() * 8);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < nwg; ++i) {
      double dc = (double)(st[4 * i + 2] - st[4 * i]), dr = (double)(st[4 * i + 3] - st[4 * i + 1]);
      if (dr > 0) { ghz.push_back(dc / dr * 0.1); cyc.push_back(dc); }
    }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    printf("  in-kernel clock (median over %zu workgroups): %.2f GHz; tile loop %.0f shader cycles per workgroup (16 tiles)\n", ghz.size(), ghz[ghz.size() / 2], cyc[cyc.size() / 2]);
  }
#endif
  return 0;
}
