// Stand-alone timing of the 32x32x16 dK / dV kernel (attention_mfma32.hip) at the dec0 shape, same data as attn_bwd_bench.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I scenesplat_amd/csrc [-DFB32_WAVES=4] scripts/ubench/attn_dkv32_bench.hip -o ...
#include "../../scenesplat_amd/csrc/attention_mfma32.hip"
#include "attn_dkv32_rejected.inc"   // the rejected kernel (see its header)
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
int main(int argc, char** argv) {
  const int W = 100, K = 1024, H = 16, D = 48, C = H * D;
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
  std::vector<float> lse((size_t)n * H, 3.0f), dl((size_t)n * H, 0.1f);
  unsigned short *dq, *dgo, *ddq, *dex; int32_t *dg, *dw; float *dlse, *ddelta;
  hipMalloc(&dq, q.size() * 2); hipMalloc(&dgo, go.size() * 2); hipMalloc(&ddq, q.size() * 2);
  hipMalloc(&dex, 1 << 20); hipMalloc(&dg, n * 4); hipMalloc(&dw, (W + 1) * 4); hipMalloc(&dlse, lse.size() * 4); hipMalloc(&ddelta, lse.size() * 4);
  hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dgo, go.data(), go.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dg, gidx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, win.data(), (W + 1) * 4, hipMemcpyHostToDevice);
  hipMemcpy(dlse, lse.data(), lse.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ddelta, dl.data(), dl.size() * 4, hipMemcpyHostToDevice);
  const float scale = 1.f / sqrtf((float)D);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) ss_attn_bwd_dkv_mfma32(dq, dgo, dlse, ddelta, dg, dg, dw, W, K, ddq, dex, C, H, scale, 0);
  hipDeviceSynchronize();
  const int iters = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) ss_attn_bwd_dkv_mfma32(dq, dgo, dlse, ddelta, dg, dg, dw, W, K, ddq, dex, C, H, scale, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  printf("dK/dV 32x32x16 (%d waves): %.3f ms\n", (int)FB32_WAVES, ms);
  return 0;
}
