// Micro-benchmark (gfx950): how VALU, transcendental and MFMA work of co-resident waves shares a SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates
// Each test runs a loop of N groups of independent instructions in every wave of a 256-thread block (one wave per
// SIMD per block), with B blocks per CU (= waves per SIMD), and reports shader cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(16))) float f16v;

#define REP 256
template <int MODE, int THREADS = 256>
__global__ void __launch_bounds__(THREADS) k(float* out, unsigned long long* cyc, int role_split) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  f16v acc = {0}; f16v acc2 = {0};
  bf8_t x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i + 1); }
  // role_split: with 2 blocks per CU, odd blocks (by launch order) take the "other" role in MODE 5
  int mode = MODE;
  if (MODE == 5) mode = (threadIdx.x >= 256) ? role_split : 3;   // 512-thread block: waves 0-3 MFMA, waves 4-7 the partner role
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REP; ++it) {
    if (mode == 1) {          // 16 independent v_exp_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    } else if (mode == 2) {   // 16 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
    } else if (mode == 3) {   // 4 MFMA 32x32x16 on two accumulators
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc2, 0, 0, 0);
    } else if (mode == 4) {   // 4 MFMA + 16 exp + 16 fma in one wave
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i + 8] = __builtin_fmaf(a[i + 8], 1.0001f, 0.5f); }
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc2, 0, 0, 0);
#pragma unroll
      for (int i = 4; i < 8; ++i) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i + 8] = __builtin_fmaf(a[i + 8], 1.0001f, 0.5f); }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i + 8] = __builtin_fmaf(a[i + 8], 1.0001f, 0.5f); }
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc2, 0, 0, 0);
#pragma unroll
      for (int i = 4; i < 8; ++i) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i + 8] = __builtin_fmaf(a[i + 8], 1.0001f, 0.5f); }
    } else if (mode == 6) {   // 8 exp + 8 fma (softmax-like mix)
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i + 8] = __builtin_fmaf(a[i + 8], 1.0001f, 0.5f); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + acc[i] + acc2[i];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> double run(int blocks_per_cu, const char* name, double insts_per_iter) {
  int ncu = 256, nb = ncu * blocks_per_cu;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, nb * 256 * sizeof(float)); hipMalloc(&cyc, nb * 4 * sizeof(unsigned long long));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nb * 4);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  double med = (double)h[h.size() / 2];
  // per SIMD: blocks_per_cu waves share a SIMD; time per wave loop = med; instructions issued per SIMD = waves * REP * insts
  double per_inst = med / (REP * insts_per_iter);             // cycles per instruction as seen by ONE wave
  double per_simd = med / (REP * insts_per_iter * blocks_per_cu);   // cycles per instruction per SIMD (aggregate)
  printf("%-34s waves/SIMD %d: %8.0f cyc/loop  %6.2f cyc/inst/wave  %6.2f cyc/inst/SIMD\n", name, blocks_per_cu, med / REP, per_inst, per_simd);
  hipFree(out); hipFree(cyc);
  return med;
}

int main() {
  for (int b : {1, 2, 3, 4}) run<1>(b, "v_exp_f32 x16", 16);
  for (int b : {1, 2, 3, 4}) run<2>(b, "v_fma_f32 x16", 16);
  for (int b : {1, 2, 3}) run<3>(b, "mfma32x32x16 x4", 4);
  for (int b : {1, 2, 3}) run<6>(b, "8 exp + 8 fma", 16);
  for (int b : {1, 2, 3}) run<4>(b, "4 mfma + 16 exp + 16 fma (1 wave)", 36);
  // two waves per SIMD with split roles: waves 0-3 run MFMA x4 per iteration, waves 4-7 run exp x16 (1) / fma x16 (2) / exp+fma (6)
  for (int partner : {1, 2, 6}) {
    int nb = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nb * 512 * sizeof(float)); hipMalloc(&cyc, nb * 8 * sizeof(unsigned long long));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<5, 512>), dim3(nb), dim3(512), 0, 0, out, cyc, partner);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b;
    for (int i = 0; i < nb; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b).push_back(h[i * 8 + w]);
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("split roles (partner mode %d): MFMA waves %6.0f cyc/loop (4 mfma: alone = 128)   partner waves %6.0f cyc/loop (16 inst)\n",
           partner, (double)a[a.size() / 2] / REP, (double)b[b.size() / 2] / REP);
    hipFree(out); hipFree(cyc);
  }
  return 0;
}
