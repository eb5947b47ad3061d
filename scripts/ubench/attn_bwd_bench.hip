// Stand-alone timing harness of the MFMA attention BACKWARD kernels (dQ, dK/dV) with ablation masks (diagnostic).
//   FA_ABL bits: 1 no tile loads / LDS staging, 64 no epilogue stores, 128 no prologue row loads
#include "../../scenesplat_amd/csrc/attention_mfma.hip"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
int ss_attn_fwd_mfma32(const void*, const int32_t*, const int32_t*, const int32_t*, int, int, void*, float*, int, int, float, hipStream_t) { return 0; }

int main(int argc, char** argv) {
  const int W = 100, K = 1024, H = 16, D = argc > 1 ? atoi(argv[1]) : 48, C = H * D;
  const int64_t n = (int64_t)W * K;
  std::vector<unsigned short> q((size_t)n * 3 * C), go((size_t)n * C);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto bf = [&](float f) { unsigned int u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
  for (auto& v : q) v = bf(nd(rng));
  for (auto& v : go) v = bf(nd(rng));
  std::vector<int32_t> gidx(n), win(W + 1);
  std::iota(gidx.begin(), gidx.end(), 0);
  std::shuffle(gidx.begin(), gidx.end(), rng);
  for (int w = 0; w <= W; ++w) win[w] = w * K;
  std::vector<float> lse((size_t)n * H, 3.0f);
  unsigned short *dq, *dgo, *dout, *ddq, *dex; int32_t *dg, *dw; float *dlse, *ddelta;
  hipMalloc(&dq, q.size() * 2); hipMalloc(&dgo, go.size() * 2); hipMalloc(&dout, go.size() * 2); hipMalloc(&ddq, q.size() * 2);
  hipMalloc(&dex, 1 << 20); hipMalloc(&dg, n * 4); hipMalloc(&dw, (W + 1) * 4); hipMalloc(&dlse, lse.size() * 4); hipMalloc(&ddelta, lse.size() * 4);
  hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dgo, go.data(), go.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dout, go.data(), go.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dg, gidx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, win.data(), (W + 1) * 4, hipMemcpyHostToDevice);
  hipMemcpy(dlse, lse.data(), lse.size() * 4, hipMemcpyHostToDevice);
  const float scale = 1.f / sqrtf((float)D);
  const int chunks = (K + FA_BQ - 1) / FA_BQ, chunks2 = (K + DKV_BKEYS - 1) / DKV_BKEYS;
  dim3 g1(W * H * chunks), b1(FA_THREADS), g2(W * H * chunks2), b2(DKV_THREADS);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  auto run_dq = [&]() { hipLaunchKernelGGL((k_attn_bwd_dq_mfma<48>), g1, b1, 0, 0, dq, dgo, dout, dlse, ddelta, dg, dg, dw, ddq, C, H, scale, chunks); };
  auto run_kv = [&]() { hipLaunchKernelGGL((k_attn_bwd_dkv_mfma<48>), g2, b2, 0, 0, dq, dgo, dlse, (const float*)ddelta, dg, dg, dw, ddq, dex, C, H, scale, chunks2); };
  for (int i = 0; i < 3; ++i) { run_dq(); run_kv(); }
  hipDeviceSynchronize();
  const int iters = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) run_dq();
  hipEventRecord(e1, 0);
  for (int i = 0; i < iters; ++i) run_kv();
  hipEventRecord(e2, 0); hipEventSynchronize(e2);
  float m1, m2; hipEventElapsedTime(&m1, e0, e1); hipEventElapsedTime(&m2, e1, e2); m1 /= iters; m2 /= iters;
  printf("attn bwd d=48 ABL=%d: dQ %.3f ms  dK/dV %.3f ms  total %.3f ms\n", (int)FA_ABL, m1, m2, m1 + m2);
  return 0;
}
