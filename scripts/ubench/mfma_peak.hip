// Sustained dense bf16 MFMA rate of the whole chip and the shader clock it holds while doing so (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_peak.hip -o scripts/ubench/bin/mfma_peak
// Every wave issues independent v_mfma_f32_32x32x16_bf16 on register operands (no memory traffic): 4 waves per SIMD x 4
// accumulators.  FLOP/s = launched MFMAs x 32768 / wall time (HIP events); the clock = s_memtime / s_memrealtime deltas
// (the latter ticks at 100 MHz).  usage: mfma_peak [iters] [random 0|1]: constant operands draw little power; random bf16
// operands that alternate between two sets make the multipliers toggle as real data does.  The 2.5 PFLOP/s roofline peak corresponds to 256 CUs x 4 SIMDs x 1024 FLOP/cycle x 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;

__device__ __forceinline__ float rnd(unsigned x) {             // ~N(0,1)-ish pseudo-random value per (lane, slot)
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return ((float)(x & 0xffff) + (float)((x >> 16) & 0xffff) - 65535.f) * (1.f / 26754.f);
}
__global__ void __launch_bounds__(256) k_mfma(int iters, float* sink, unsigned long long* stamp, int random) {
  bf8_t a, b, a2, b2;
  for (int i = 0; i < 8; ++i) {
    const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 32 + i;
    a[i] = (__bf16)(random ? rnd(id) : (float)(threadIdx.x & 3)); b[i] = (__bf16)(random ? rnd(id + 8) : 1.0f);
    a2[i] = (__bf16)(random ? rnd(id + 16) : (float)(threadIdx.x & 3)); b2[i] = (__bf16)(random ? rnd(id + 24) : 1.0f);
  }
  f16_t c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c1, 0, 0, 0);     // operands alternate: the multiplier inputs toggle
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { stamp[2 * blockIdx.x] = t1 - t0; stamp[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * 4, iters = argc > 1 ? atoi(argv[1]) : 200000;     // 4 blocks x 4 waves per CU = 4 waves per SIMD
  const int random = argc > 2 ? atoi(argv[2]) : 0;
  float* sink; unsigned long long* stamp;
  hipMalloc(&sink, 4); hipMalloc(&stamp, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    k_mfma<<<blocks, 256>>>(iters, sink, stamp, random);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(2 * blocks);
    hipMemcpy(st.data(), stamp, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < blocks; ++i) if (st[2 * i + 1]) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("%s operands, rep %d: %.1f ms  %.0f TFLOP/s dense bf16 (%.1f %% of 2.5 PF)  shader clock %.2f GHz (median of %zu workgroups)\n", random ? "random" : "constant", rep, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 25.0, ghz.empty() ? 0.0 : ghz[ghz.size() / 2], ghz.size());
  }
  return 0;
}
