// Open-vocabulary scan (reference pointcept/engines/hooks/evaluator.py:793-800 and
// pointcept/engines/test.py:335-351): logits = feat (N, D) @ text^T (D, C), probs = sigmoid(logits),
// then either the per-Gaussian max / argmax (evaluator) or the full probabilities accumulated into
// pred[idx[i]] (fragment voting in the tester).  One pass over the features: bf16 MFMA GEMM with the
// sigmoid / max / arg-max / scatter fused into the epilogue; C <= 256 classes sit in one N tile.
#include "common.h"
#include "../../include/scenesplat_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#ifndef SC_WAVES
#define SC_WAVES 4     // 4 waves x 32 rows, two workgroups per CU (one stages while the other multiplies): 0.401 ms per 1 M x 768 x 160 scan
                       // against 0.448 ms with 8 waves x 32 rows and one workgroup per CU (sustained, scripts/power_probe.py scan)
#endif
#define SC_THREADS (64 * SC_WAVES)
#define SC_BM (32 * SC_WAVES)      // rows per workgroup
#define SC_BK 64

__device__ __forceinline__ int sc_row_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int NT>   // NT sixteen-wide class tiles (C <= 16 NT)
__global__ void __launch_bounds__(SC_THREADS)
k_feat_text_scan(const unsigned short* __restrict__ feat, const unsigned short* __restrict__ text, int64_t n, int D, int C,
                 float* __restrict__ max_prob, int32_t* __restrict__ argmax, const int32_t* __restrict__ idx,
                 float* __restrict__ pred_accum) {
  constexpr int BN = 16 * NT;
  constexpr int AIMG = SC_BM * SC_BK * 2, BIMG = BN * SC_BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * (AIMG + BIMG)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * SC_BM;
  f32x4_t acc[2][NT];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int NLA = (SC_BM * 8) / SC_THREADS, NLB = (BN * 8 + SC_THREADS - 1) / SC_THREADS;
  uint4 sa[NLA], sb[NLB];
  auto stage_load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      int c = i * SC_THREADS + tid; int64_t r = m0 + (c >> 3); int k = k0 + (c & 7) * 8;
      sa[i] = (r < n && k < D) ? *reinterpret_cast<const uint4*>(feat + r * D + k) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      int c = i * SC_THREADS + tid; int cl = c >> 3, k = k0 + (c & 7) * 8;
      sb[i] = (c < BN * 8 && cl < C && k < D) ? *reinterpret_cast<const uint4*>(text + (int64_t)cl * D + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stage_write = [&](int b) {
    char* A = smem + b * (AIMG + BIMG); char* B = A + AIMG;
#pragma unroll
    for (int i = 0; i < NLA; ++i) { int c = i * SC_THREADS + tid; *reinterpret_cast<uint4*>(A + sc_row_off(c >> 3, c & 7)) = sa[i]; }
#pragma unroll
    for (int i = 0; i < NLB; ++i) { int c = i * SC_THREADS + tid; if (c < BN * 8) *reinterpret_cast<uint4*>(B + sc_row_off(c >> 3, c & 7)) = sb[i]; }
  };
  const int ksteps = (D + SC_BK - 1) / SC_BK;
  stage_load(0); stage_write(0);
  __syncthreads();
  for (int it = 0; it < ksteps; ++it) {
    const int b = it & 1;
    if (it + 1 < ksteps) stage_load((it + 1) * SC_BK);
    const char* A = smem + b * (AIMG + BIMG); const char* B = A + AIMG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf8_t af[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        af[mi] = __builtin_bit_cast(bf8_t, *reinterpret_cast<const uint4*>(A + sc_row_off(32 * wave + 16 * mi + lq, 4 * ks + g)));
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        bf8_t bf = __builtin_bit_cast(bf8_t, *reinterpret_cast<const uint4*>(B + sc_row_off(16 * ni + lq, 4 * ks + g)));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = MFMA16(bf, af[mi], acc[mi][ni]);   // rows = classes, cols = Gaussians
      }
    }
    if (it + 1 < ksteps) stage_write(b ^ 1);
    __syncthreads();
  }
  // epilogue: lane (lq, g) holds, for Gaussian 32*wave + 16*mi + lq, the classes 16*ni + 4*g + r
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int64_t row = m0 + 32 * wave + 16 * mi + lq;
    if (pred_accum && row < n) {
      const int64_t dst = idx ? idx[row] : row;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int cl = 16 * ni + 4 * g + r;
          if (cl < C) pred_accum[dst * C + cl] += 1.f / (1.f + __expf(-acc[mi][ni][r]));
        }
    }
    if (max_prob) {
      float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int cl = 16 * ni + 4 * g + r;
          float v = acc[mi][ni][r];
          if (cl < C && (v > best || (v == best && cl < bi))) { best = v; bi = cl; }
        }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        float ob = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (g == 0 && row < n) { max_prob[row] = 1.f / (1.f + __expf(-best)); argmax[row] = bi; }
    }
  }
}

extern "C" int ss_feat_text_scan(const void* feat_bf16, const void* text_bf16, int64_t n, int dim, int num_classes,
                                 float* max_prob, int32_t* argmax, const int32_t* idx, float* pred_accum, hipStream_t stream) {
  if (n < 0 || dim <= 0 || (dim & 7) || num_classes < 1 || num_classes > 256) return SS_ERR_ARG;
  if ((max_prob == nullptr) != (argmax == nullptr)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  dim3 g(ss_div_up(n, SC_BM)), b(SC_THREADS);
  const unsigned short* f = (const unsigned short*)feat_bf16; const unsigned short* t = (const unsigned short*)text_bf16;
  const int nt = (num_classes + 15) / 16;
#define SS_SCAN(NTN) SS_LAUNCH(k_feat_text_scan<NTN>, g, b, 0, stream, f, t, n, dim, num_classes, max_prob, argmax, idx, pred_accum)
  // exact tile counts for the label sets of the reference (20 / 21 classes: 2, 100: 7, 160: 10, 200: 13)
  if (nt <= 2) SS_SCAN(2); else if (nt <= 4) SS_SCAN(4); else if (nt <= 7) SS_SCAN(7); else if (nt <= 8) SS_SCAN(8);
  else if (nt <= 10) SS_SCAN(10); else if (nt <= 13) SS_SCAN(13); else SS_SCAN(16);
#undef SS_SCAN
  return SS_OK;
}
