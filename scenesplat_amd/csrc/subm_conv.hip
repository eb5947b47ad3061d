// Submanifold 3-D convolution on a precomputed rulebook, bf16 MFMA implicit GEMM (gfx950).
// Replaces spconv.SubMConv3d (ptv3:278-284 CPE k=3, ptv3:499-506 stem k=5) forward, dgrad, wgrad.
//
//   forward / dgrad   out[i][co] = b[co] + sum_t sum_ci in[nbr[t][i]][ci] * W[co][t][ci]
//       one kernel (k_subm_gemm).  GEMM view: M = sites, N = Cout, K = taps x Cin; the A tile is
//       GATHERED through the rulebook while it is staged (128-byte row pieces, zero rows for
//       missing neighbours), taps with no neighbour in the whole 128-site tile are skipped --
//       sites are walked in z-order (rowperm) so tiles are spatially compact and on surfaces
//       roughly two thirds of the taps drop out.  dgrad = the same kernel on dout with the
//       tap-mirrored, transposed weights (rulebook symmetry nbr[i][t]=j <=> nbr[j][T-1-t]=i).
//   wgrad             dW[co][t][ci] = sum_i dout[i][co] * in[nbr[t][i]][ci]
//       k_subm_wgrad: M = Cout, N = Cin, K = sites (split across workgroups, fp32 atomics into a
//       zeroed dW); both operands are [k][m]-major, so fragments come from plain LDS images via
//       ds_read_b64_tr_b16 with the same k permutation on both sides.
// MFMA: v_mfma_f32_16x16x32_bf16, 128x128 workgroup tile, 4 waves (2x2) of 64x64, BK = 64.
#include "common.h"
#include "../../include/scenesplat_hip.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

#define CV_THREADS 256
#define CV_BM 128
#define CV_BN 128
#define CV_BK 64
#ifndef SS_CONV_BIG_MIN_TILES
#define SS_CONV_BIG_MIN_TILES 64    // uniform-102400 level 3 (105 tiles): pipeline 144 us vs 175 us
#endif
#ifndef SS_CONV_DMA_SMALL
#define SS_CONV_DMA_SMALL false
#endif
#ifndef SS_CONV_DMA_BIG
#define SS_CONV_DMA_BIG false
#endif
#define CV_TG 27   // taps per rulebook group held in LDS (27 keeps the workgroup under 80 KB: 2 per CU)

__device__ __forceinline__ bf8_t cv_as_bf8(uint4 v) { return __builtin_bit_cast(bf8_t, v); }
__device__ __forceinline__ int cv_row_off(int row, int chunk) {   // [row][64 bf16] image, 8 chunks of 16 B, XOR swizzle
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ bf8_t cv_lds_b128(const char* base, int off) {
  return cv_as_bf8(*reinterpret_cast<const uint4*>(base + off));
}
__device__ __forceinline__ s4_t cv_lds_tr(const char* addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(addr));
}
__device__ __forceinline__ bf8_t cv_cat(s4_t lo, s4_t hi) {
  s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8_t, v);
}
// plain [k][128 cols] bf16 image (256-B rows, 16 chunks), swizzled for conflict-free transposed reads
__device__ __forceinline__ int cv_tr_off(int row, int chunk) {
  return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// WM x WN waves, each TM x TN sixteen-wide tiles: BM = 16*TM*WM sites, BN = 16*TN*WN channels per workgroup.
//   <2,2,4,4>  128 x 128, 256 threads, ~79 KB LDS (2 workgroups / CU)   -- narrow layers (Cout < 256)
//   <4,2,4,8>  256 x 256, 512 threads, ~156 KB LDS (1 workgroup / CU)   -- wide layers: half the L2->LDS bytes
//              per FLOP (the 128 x 128 form moved 20 GB per dec0 call = 8.5 TB/s and was bound by it)
__device__ uint4 g_zero_row[8];   // 128 zero bytes: the source of every missing-neighbour / out-of-range LDS-DMA piece

__device__ __forceinline__ void cv_glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// DMA = true: tiles are filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction = 8 rows x 128 B):
// the gather is the per-lane SOURCE address, the XOR swizzle is applied to the source chunk index, the LDS image
// stays lane-linear; no staging VGPRs and no ds_write traffic (the register path was limited by the ~79 B/clk
// VGPR->LDS store path).  The __syncthreads() that ends a K-step drains vmcnt, so the next tile has landed.
template <typename OutT, int WM, int WN, int TM, int TN, bool DMA>
__global__ void __launch_bounds__(64 * WM * WN)
k_subm_gemm(const unsigned short* __restrict__ in, const unsigned short* __restrict__ W, const float* __restrict__ bias,
            const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowperm, OutT* __restrict__ out, int n,
            int Cin, int Cout, int taps, float* __restrict__ acc32) {
  // gridDim.z > 1: split-K over tap ranges for small levels (a 1,600-site level has 26 tiles for 256 CUs and a
  // serial 27-tap loop); partial sums go to the zeroed fp32 buffer acc32 with atomics, the caller converts
  constexpr int THREADS = 64 * WM * WN, BM = 16 * TM * WM, BN = 16 * TN * WN;
  constexpr int AIMG = BM * CV_BK * 2, BIMG = BN * CV_BK * 2;
  constexpr int NLA = (BM * 8) / THREADS, NLB = (BN * 8) / THREADS;      // 16-byte chunks per thread per K-step
  __shared__ __attribute__((aligned(16))) char smem[2 * (AIMG + BIMG)];   // A0 B0 A1 B1
  __shared__ int32_t nbr_s[CV_TG][BM];
  __shared__ int32_t rowid_s[BM];
  __shared__ int act_s[CV_TG];
  __shared__ unsigned mask_s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  for (int r = tid; r < BM; r += THREADS) {
    int k = m0 + r;
    rowid_s[r] = k < n ? (rowperm ? rowperm[k] : k) : -1;
  }
  f32x4_t acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int ksteps = (Cin + CV_BK - 1) / CV_BK;
  uint4 sa[DMA ? 1 : NLA], sb[DMA ? 1 : NLB];
  auto stage_dma = [&](int b, int tt, int tap, int ci0) {
    char* A = smem + b * (AIMG + BIMG); char* B = A + AIMG;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      int c = i * THREADS + tid;
      int r = c >> 3, ci = ci0 + (((c & 7) ^ ((r >> 1) & 7)) << 3);      // LDS slot (r, c&7) holds logical chunk slot ^ swz(r)
      int src = nbr_s[tt][r];
      const void* g = (src >= 0 && ci < Cin) ? (const void*)(in + (int64_t)src * Cin + ci) : (const void*)g_zero_row;
      cv_glds16(g, A + (c & ~63) * 16);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      int c = i * THREADS + tid;
      int r = c >> 3, co = n0 + r, ci = ci0 + (((c & 7) ^ ((r >> 1) & 7)) << 3);
      const void* g = (co < Cout && ci < Cin) ? (const void*)(W + ((int64_t)co * taps + tap) * Cin + ci) : (const void*)g_zero_row;
      cv_glds16(g, B + (c & ~63) * 16);
    }
  };
  auto stage_load = [&](int tt, int tap, int ci0) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      int c = i * THREADS + tid;
      int r = c >> 3, ci = ci0 + (c & 7) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      int src = nbr_s[tt][r];
      if (src >= 0 && ci < Cin) v = *reinterpret_cast<const uint4*>(in + (int64_t)src * Cin + ci);
      sa[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      int c = i * THREADS + tid;
      int co = n0 + (c >> 3), ci = ci0 + (c & 7) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (co < Cout && ci < Cin) v = *reinterpret_cast<const uint4*>(W + ((int64_t)co * taps + tap) * Cin + ci);
      sb[i] = v;
    }
  };
  auto stage_write = [&](int b) {
    char* A = smem + b * (AIMG + BIMG); char* B = A + AIMG;
#pragma unroll
    for (int i = 0; i < NLA; ++i) { int c = i * THREADS + tid; *reinterpret_cast<uint4*>(A + cv_row_off(c >> 3, c & 7)) = sa[i]; }
#pragma unroll
    for (int i = 0; i < NLB; ++i) { int c = i * THREADS + tid; *reinterpret_cast<uint4*>(B + cv_row_off(c >> 3, c & 7)) = sb[i]; }
  };
  const int taps_per_z = (taps + gridDim.z - 1) / gridDim.z;
  const int tap_beg = blockIdx.z * taps_per_z, tap_end = min(taps, tap_beg + taps_per_z);
  // a split-K slice can be EMPTY (27 taps over 8 slices of 4: slice 7 starts at tap 28).  Its workgroups used to fall through to
  // the epilogue without ever passing a barrier, read rowid_s before the threads that fill it had written it, and added zeros at
  // row = whatever LDS held: an out-of-bounds atomic (memory access fault at n = 4,096, round 3).  Nothing to add: leave.
  if (tap_beg >= tap_end) return;
  __syncthreads();                       // rowid_s is complete for every path below
  for (int tg = tap_beg; tg < tap_end; tg += CV_TG) {
    const int nt = min(CV_TG, tap_end - tg);
    __syncthreads();                     // previous group's LDS reads are done
    if (tid == 0) mask_s = 0u;
    __syncthreads();
    for (int e = tid; e < nt * BM; e += THREADS) {
      int tt = e / BM, r = e - tt * BM;
      int row = rowid_s[r];
      int v = row >= 0 ? nbr[(int64_t)(tg + tt) * n + row] : -1;
      nbr_s[tt][r] = v;
      if (v >= 0) atomicOr(&mask_s, 1u << tt);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned mk = mask_s; int c = 0;
      for (int tt = 0; tt < nt; ++tt) if (mk >> tt & 1u) act_s[c++] = tt;
      mask_s = (unsigned)c;
    }
    __syncthreads();
    const int nact = (int)mask_s;
    const int iters = nact * ksteps;
    if (iters == 0) continue;
    {
      int tt = act_s[0];
      if (DMA) stage_dma(0, tt, tg + tt, 0);
      else { stage_load(tt, tg + tt, 0); stage_write(0); }
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      const int b = it & 1;
      if (it + 1 < iters) {
        // channel chunk OUTER, tap INNER: the ~9-13 active taps of a tile gather (almost) the same neighbour
        // rows, so one 128-byte column slice of them (tile rows x 128 B ~ 40 KB per CU) stays L2-resident
        // across the taps instead of streaming every tap's full rows from the Infinity Cache
        int k_ = (it + 1) / nact, a_ = (it + 1) - k_ * nact;
        int tt = act_s[a_];
        if (DMA) stage_dma(b ^ 1, tt, tg + tt, k_ * CV_BK);
        else stage_load(tt, tg + tt, k_ * CV_BK);
      }
      const char* A = smem + b * (AIMG + BIMG); const char* B = A + AIMG;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf8_t af[TM], bf[TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af[mi] = cv_lds_b128(A, cv_row_off(16 * TM * wm + 16 * mi + lq, 4 * ks + g));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) bf[ni] = cv_lds_b128(B, cv_row_off(16 * TN * wn + 16 * ni + lq, 4 * ks + g));
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = MFMA16(bf[ni], af[mi], acc[mi][ni]);   // C^T: rows = channels
      }
      if (!DMA && it + 1 < iters) stage_write(b ^ 1);
      __syncthreads();
    }
  }
  // epilogue: acc[mi][ni] holds C^T: element r = channel n0 + 16TN*wn + 16ni + 4g + r of site 16TM*wm + 16mi + lq
  // -> one 8-byte (bf16) / 16-byte (f32) store per tile instead of four 2-byte ones
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row = rowid_s[16 * TM * wm + 16 * mi + lq];
    if (row < 0) continue;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = n0 + 16 * TN * wn + 16 * ni + 4 * g;
      if (acc32) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r < Cout) atomicAdd(acc32 + (int64_t)row * Cout + col + r, acc[mi][ni][r] + ((bias && blockIdx.z == 0) ? bias[col + r] : 0.f));
      } else if (col + 3 < Cout && (Cout & 3) == 0) {
        float4 v = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
        if (bias) { float4 bv = *reinterpret_cast<const float4*>(bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
        OutT* op = out + (int64_t)row * Cout + col;
        if (sizeof(OutT) == 2) {
          uint2 u; u.x = pack_bf16x2(v.x, v.y); u.y = pack_bf16x2(v.z, v.w);
          *reinterpret_cast<uint2*>(op) = u;
        } else {
          *reinterpret_cast<float4*>(op) = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r < Cout) ElemIO<OutT>::store(out + (int64_t)row * Cout + col + r, acc[mi][ni][r] + (bias ? bias[col + r] : 0.f));
      }
    }
  }
}

// active 64-site blocks per tap (in rowperm order): blk_list[t][0 .. blk_count[t]).
// Two launches: one wave per (tap, block) writes a 0/1 flag INTO blk_list, then one workgroup per tap compacts its
// row in place (1024 flags per round; a round reads its flags before anything is written, and writes land at
// positions <= the round's first flag).
__global__ void k_subm_block_flags(const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowperm, int n, int nblocks,
                                   int32_t* __restrict__ blk_list) {
  const int t = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * 4 + wave;
  if (blk >= nblocks) return;
  const int k = blk * 64 + lane;
  int j = -1;
  if (k < n) j = nbr[(int64_t)t * n + (rowperm ? rowperm[k] : k)];
  unsigned long long m = __ballot(j >= 0);
  if (lane == 0) blk_list[(int64_t)t * nblocks + blk] = m != 0ULL;
}

__global__ void __launch_bounds__(1024) k_subm_block_compact(int nblocks, int32_t* __restrict__ blk_count,
                                                              int32_t* __restrict__ blk_list) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  int32_t* row = blk_list + (int64_t)blockIdx.x * nblocks;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nblocks; c0 += 1024) {
    const int b = c0 + tid;
    const int f = b < nblocks ? row[b] : 0;
    unsigned long long m = __ballot(f != 0);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();                          // every flag of this round is in registers
    int before = base_s;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (f) row[before + __popcll(m & ((1ULL << lane) - 1ULL))] = b;
    __syncthreads();
    if (tid == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wsum[w]; base_s += s; }
    __syncthreads();
  }
  if (tid == 0) blk_count[blockIdx.x] = base_s;
}

extern "C" int ss_subm_block_lists(const int32_t* nbr, const int32_t* rowperm, int64_t n, int taps, int32_t* blk_count,
                                   int32_t* blk_list, hipStream_t stream) {
  if (n <= 0 || taps <= 0 || n >= (1LL << 31)) return SS_ERR_ARG;
  const int nblocks = ss_div_up(n, 64);
  SS_LAUNCH(k_subm_block_flags, dim3(ss_div_up(nblocks, 4), taps), dim3(256), 0, stream, nbr, rowperm, (int)n, nblocks, blk_list);
  SS_LAUNCH(k_subm_block_compact, dim3(taps), dim3(1024), 0, stream, nblocks, blk_count, blk_list);
  return SS_OK;
}

// dW[co][t][ci] += sum over this workgroup's share of the tap's ACTIVE 64-site blocks.
// WM x WN waves of TM x TN tiles: <2,2,4,4> = 128 x 128 (narrow layers), <2,4,8,4> = 256 x 256 / 8 waves (wide).
template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(64 * WM * WN)
k_subm_wgrad(const unsigned short* __restrict__ in, const unsigned short* __restrict__ dout,
             const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowperm, const int32_t* __restrict__ blk_count,
             const int32_t* __restrict__ blk_list, float* __restrict__ dW, int n, int Cin, int Cout, int taps, int ntiles_n,
             int nblocks_total) {
  constexpr int THREADS = 64 * WM * WN, BMc = 16 * TM * WM, BNc = 16 * TN * WN;
  constexpr int ARB = BMc * 2, BRB = BNc * 2;              // row bytes of the [site][col] images
  constexpr int AIMG = 64 * ARB, BIMG = 64 * BRB;
  constexpr int ACH = BMc / 8, BCH = BNc / 8;               // 16-byte chunks per row
  constexpr int NLA = (64 * ACH) / THREADS, NLB = (64 * BCH) / THREADS;
  __shared__ __attribute__((aligned(16))) char smem[2 * (AIMG + BIMG)];   // A0 B0 A1 B1
  __shared__ int32_t isite_s[2][64], jsite_s[2][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = (blockIdx.x / ntiles_n) * BMc, n0 = (blockIdx.x % ntiles_n) * BNc;
  const int tap = blockIdx.y;
  const int cnt = blk_count[tap];
  const int per = (cnt + gridDim.z - 1) / gridDim.z;
  const int beg = blockIdx.z * per, end = min(cnt, beg + per);
  const int nblk = end - beg;
  if (nblk <= 0) return;
  const int32_t* list = blk_list + (int64_t)tap * nblocks_total + beg;
  // plain [site][cols] image, 16-byte chunks XOR-swizzled inside each group of 16 for conflict-free transposed reads
  auto img_off = [](int row, int ch, int rb) { return row * rb + ((((ch & 15) ^ ((row & 7) << 1)) | (ch & ~15)) << 4); };
  f32x4_t acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 sa[NLA], sb[NLB];
  auto index_load = [&](int b, int blk) {      // wave 0: site ids of the block
    if (tid < 64) {
      int k = list[blk] * 64 + tid;
      int i = k < n ? (rowperm ? rowperm[k] : k) : -1;
      isite_s[b][tid] = i;
      jsite_s[b][tid] = i >= 0 ? nbr[(int64_t)tap * n + i] : -1;
    }
  };
  auto stage_load = [&](int b) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      int c = i * THREADS + tid;
      int r = c / ACH, co = m0 + (c % ACH) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (jsite_s[b][r] >= 0 && co < Cout) v = *reinterpret_cast<const uint4*>(dout + (int64_t)isite_s[b][r] * Cout + co);
      sa[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      int c = i * THREADS + tid;
      int r = c / BCH, ci = n0 + (c % BCH) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      int sj = jsite_s[b][r];
      if (sj >= 0 && ci < Cin) v = *reinterpret_cast<const uint4*>(in + (int64_t)sj * Cin + ci);
      sb[i] = v;
    }
  };
  auto stage_write = [&](int b) {
    char* A = smem + b * (AIMG + BIMG); char* B = A + AIMG;
#pragma unroll
    for (int i = 0; i < NLA; ++i) { int c = i * THREADS + tid; *reinterpret_cast<uint4*>(A + img_off(c / ACH, c % ACH, ARB)) = sa[i]; }
#pragma unroll
    for (int i = 0; i < NLB; ++i) { int c = i * THREADS + tid; *reinterpret_cast<uint4*>(B + img_off(c / BCH, c % BCH, BRB)) = sb[i]; }
  };
  index_load(0, 0);
  __syncthreads();
  stage_load(0); stage_write(0);
  if (nblk > 1) index_load(1, 1);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    const int b = blk & 1;
    const bool have_next = blk + 1 < nblk;
    if (have_next) stage_load(b ^ 1);
    {
      const char* A = smem + b * (AIMG + BIMG); const char* B = A + AIMG;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf8_t af[TM], bf[TN];
        const int r0 = 32 * kk + 4 * g + (lq >> 2);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
          int ch = (16 * TM * wm + 16 * mi) / 8 + ((lq & 3) >> 1);
          af[mi] = cv_cat(cv_lds_tr(A + img_off(r0, ch, ARB) + 8 * (lq & 1)), cv_lds_tr(A + img_off(r0 + 16, ch, ARB) + 8 * (lq & 1)));
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          int ch = (16 * TN * wn + 16 * ni) / 8 + ((lq & 3) >> 1);
          bf[ni] = cv_cat(cv_lds_tr(B + img_off(r0, ch, BRB) + 8 * (lq & 1)), cv_lds_tr(B + img_off(r0 + 16, ch, BRB) + 8 * (lq & 1)));
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = MFMA16(af[mi], bf[ni], acc[mi][ni]);
      }
    }
    __syncthreads();                    // all reads of buffer b and of index slot b are done
    if (have_next) stage_write(b ^ 1);
    if (blk + 2 < nblk) index_load(b, blk + 2);
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int co = m0 + 16 * TM * wm + 16 * mi + 4 * g + r;
      if (co < Cout) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          int ci = n0 + 16 * TN * wn + 16 * ni + lq;
          if (ci < Cin) atomicAdd(dW + ((int64_t)co * taps + tap) * Cin + ci, acc[mi][ni][r]);
        }
      }
    }
}

template <typename OutT>
static int subm_gemm_launch(const unsigned short* x, const unsigned short* w, const float* bias, const int32_t* nbr,
                            const int32_t* rowperm, OutT* out, int n, int cin, int cout, int taps, hipStream_t stream,
                            float* acc32 = nullptr, int splits = 1) {
  if (acc32) {
    dim3 g(ss_div_up(n, 128), ss_div_up(cout, 128), splits), b(256);
    SS_LAUNCH((k_subm_gemm<OutT, 2, 2, 4, 4, SS_CONV_DMA_SMALL>), g, b, 0, stream, x, w, bias, nbr, rowperm, out, n, cin, cout, taps, acc32);
    return SS_OK;
  }
  // the 256 x 256 form needs enough tiles to fill 256 CUs at one workgroup each
  if (cout >= 256 && (int64_t)ss_div_up(n, 256) * ss_div_up(cout, 256) >= SS_CONV_BIG_MIN_TILES) {
    dim3 g(ss_div_up(n, 256), ss_div_up(cout, 256)), b(512);
    SS_LAUNCH((k_subm_gemm<OutT, 4, 2, 4, 8, SS_CONV_DMA_BIG>), g, b, 0, stream, x, w, bias, nbr, rowperm, out, n, cin, cout, taps, (float*)nullptr);
  } else {
    dim3 g(ss_div_up(n, 128), ss_div_up(cout, 128)), b(256);
    SS_LAUNCH((k_subm_gemm<OutT, 2, 2, 4, 4, SS_CONV_DMA_SMALL>), g, b, 0, stream, x, w, bias, nbr, rowperm, out, n, cin, cout, taps, (float*)nullptr);
  }
  return SS_OK;
}

// SS_CONV_PIPE=0/1 overrides the choice between k_subm_gemm and the LDS-DMA pipeline kernel (gemm8.hip)
static int conv_pipe_mode() {
  static int mode = -2;
  if (mode == -2) { const char* e = getenv("SS_CONV_PIPE"); mode = e ? atoi(e) : -1; }
  return mode;
}

extern "C" int ss_subm_conv_fwd(const void* in, const void* weight, const float* bias, const int32_t* nbr,
                                const int32_t* rowperm, void* out, int64_t n, int cin, int cout, int taps, int out_dtype,
                                hipStream_t stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || taps <= 0 || (cin & 7) || n >= (1LL << 31)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  if (ss_gemm8_ok(n, cin, cout, taps) && conv_pipe_mode() != 0 &&
      (conv_pipe_mode() == 1 || (cout >= 256 && (int64_t)ss_div_up(n, 256) * ss_div_up(cout, 256) >= SS_CONV_BIG_MIN_TILES)))
    return ss_subm_conv_fwd_pipe(in, weight, bias, nbr, rowperm, out, n, cin, cout, taps, out_dtype, stream);
  const unsigned short* x = (const unsigned short*)in; const unsigned short* w = (const unsigned short*)weight;
  if (out_dtype == SS_BF16) return subm_gemm_launch<unsigned short>(x, w, bias, nbr, rowperm, (unsigned short*)out, (int)n, cin, cout, taps, stream);
  if (out_dtype == SS_F32) return subm_gemm_launch<float>(x, w, bias, nbr, rowperm, (float*)out, (int)n, cin, cout, taps, stream);
  return SS_ERR_ARG;
}

static bool conv_fwd_uses_pipe(int64_t n, int cin, int cout, int taps) {
  return ss_gemm8_ok(n, cin, cout, taps) && conv_pipe_mode() != 0 &&
         (conv_pipe_mode() == 1 || (cout >= 256 && (int64_t)ss_div_up(n, 256) * ss_div_up(cout, 256) >= SS_CONV_BIG_MIN_TILES));
}
extern "C" int ss_subm_conv_fwd_uses_pipe(int64_t n, int cin, int cout, int taps) { return conv_fwd_uses_pipe(n, cin, cout, taps) ? 1 : 0; }

// ss_subm_conv_fwd with the walk-order rulebook beside the plain one: the pipeline kernel (wide, large levels) reads nbr_walk,
// the other kernels nbr.  nbr_walk NULL: ss_subm_conv_fwd.
extern "C" int ss_subm_conv_fwd_walk(const void* in, const void* weight, const float* bias, const int32_t* nbr,
                                     const int32_t* nbr_walk, const int32_t* rowperm, void* out, int64_t n, int cin, int cout,
                                     int taps, int out_dtype, hipStream_t stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || taps <= 0 || (cin & 7) || n >= (1LL << 31)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  if (nbr_walk && conv_fwd_uses_pipe(n, cin, cout, taps))
    return ss_subm_conv_fwd_pipe_walk(in, weight, bias, nbr_walk, rowperm, out, n, cin, cout, taps, out_dtype, stream);
  return ss_subm_conv_fwd(in, weight, bias, nbr, rowperm, out, n, cin, cout, taps, out_dtype, stream);
}

// split-K form for small levels: acc32 (n, cout) f32 must be zero on entry and receives out (+ bias); `splits`
// tap ranges run as separate workgroups.  ss_subm_conv_splits() is the recommended count (1 = use ss_subm_conv_fwd).
extern "C" int ss_subm_conv_splits(int64_t n, int cout, int taps) {
  int64_t tiles = (int64_t)ss_div_up(n, 128) * ss_div_up(cout, 128);
  if (tiles >= 192) return 1;
  int z = (int)((256 + tiles - 1) / tiles);
  if (z > taps) z = taps;
  if (z > 9) z = 9;
  return z < 2 ? 1 : z;
}
extern "C" int ss_subm_conv_fwd_splitk(const void* in, const void* weight, const float* bias, const int32_t* nbr,
                                       const int32_t* rowperm, float* acc32, int64_t n, int cin, int cout, int taps,
                                       int splits, hipStream_t stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || taps <= 0 || (cin & 7) || n >= (1LL << 31) || splits < 1 || !acc32) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  return subm_gemm_launch<float>((const unsigned short*)in, (const unsigned short*)weight, bias, nbr, rowperm, acc32,
                                 (int)n, cin, cout, taps, stream, acc32, splits);
}

static bool conv_wgrad_uses_pipe(int64_t n, int cin, int cout, int taps) {
  static int mode = -2;                 // SS_WGRAD_PIPE=0/1 overrides the kernel choice
  if (mode == -2) { const char* e = getenv("SS_WGRAD_PIPE"); mode = e ? atoi(e) : -1; }
  return mode != 0 && ss_wgrad8_ok(n, cin, cout, taps) && (mode == 1 || (cout >= 128 && cin >= 128));
}
extern "C" int ss_subm_conv_wgrad_uses_pipe(int64_t n, int cin, int cout, int taps) { return conv_wgrad_uses_pipe(n, cin, cout, taps) ? 1 : 0; }

extern "C" int ss_subm_conv_wgrad_walk(const void* in, const void* dout, const int32_t* nbr, const int32_t* nbr_walk,
                                       const int32_t* rowperm, const int32_t* blk_count, const int32_t* blk_list, float* dweight,
                                       int64_t n, int cin, int cout, int taps, hipStream_t stream);

extern "C" int ss_subm_conv_wgrad(const void* in, const void* dout, const int32_t* nbr, const int32_t* rowperm,
                                  const int32_t* blk_count, const int32_t* blk_list, float* dweight, int64_t n, int cin,
                                  int cout, int taps, hipStream_t stream) {
  return ss_subm_conv_wgrad_walk(in, dout, nbr, nullptr, rowperm, blk_count, blk_list, dweight, n, cin, cout, taps, stream);
}

// nbr_walk (may be NULL): the rulebook in walk order, read by the pipeline kernel when the shape runs on it
extern "C" int ss_subm_conv_wgrad_walk(const void* in, const void* dout, const int32_t* nbr, const int32_t* nbr_walk,
                                       const int32_t* rowperm, const int32_t* blk_count, const int32_t* blk_list, float* dweight,
                                       int64_t n, int cin, int cout, int taps, hipStream_t stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || taps <= 0 || (cin & 7) || (cout & 7) || n >= (1LL << 31)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  if (conv_wgrad_uses_pipe(n, cin, cout, taps)) {
    if (nbr_walk) return ss_subm_conv_wgrad_pipe_walk(in, dout, nbr_walk, rowperm, blk_count, blk_list, dweight, n, cin, cout, taps, stream);
    return ss_subm_conv_wgrad_pipe(in, dout, nbr, rowperm, blk_count, blk_list, dweight, n, cin, cout, taps, stream);
  }
  const int nblocks = ss_div_up(n, 64);
  // the 256 x 256 form pays off only with long K loops per workgroup (its fp32-atomic epilogue is 4x larger)
  const bool big = cout >= 256 && cin >= 256 && nblocks >= 1024;
  const int T = big ? 256 : 128;
  const int tm = ss_div_up(cout, T), tn = ss_div_up(cin, T);
  // taps are very unevenly populated on surfaces (about a third carry almost all pairs): size the K split for the
  // busy ones so that enough workgroups of useful work exist
  int splits = (big ? 3072 : 6144) / (tm * tn * taps);
  if (splits > nblocks / 8) splits = nblocks / 8;
  if (splits < 1) splits = 1;
  dim3 g(tm * tn, taps, splits);
  const unsigned short* x = (const unsigned short*)in; const unsigned short* go = (const unsigned short*)dout;
  if (big)
    SS_LAUNCH((k_subm_wgrad<2, 4, 8, 4>), g, dim3(512), 0, stream, x, go, nbr, rowperm, blk_count, blk_list, dweight, (int)n, cin, cout, taps, tn, nblocks);
  else
    SS_LAUNCH((k_subm_wgrad<2, 2, 4, 4>), g, dim3(256), 0, stream, x, go, nbr, rowperm, blk_count, blk_list, dweight, (int)n, cin, cout, taps, tn, nblocks);
  return SS_OK;
}
