// Serialized-window attention on the gfx950 matrix cores (impl = SS_ATTN_MFMA), bf16 in /
// fp32 accumulate, flash-style (scores never leave registers).  Replaces
// flash_attn_varlen_qkvpacked_func AND the qkv[order] / feat[inverse] row gathers around it
// (ptv3:184-216): rows are fetched through gidx and written through sidx inside the kernels.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16 everywhere.
//   forward   S^T = K Q^T (key rows, query on the lane)  -> online softmax lane-local per query
//             O^T = V^T P^T: P^T is taken straight from the S^T accumulators as the B operand
//             (the k index inside a 32-key step is permuted identically for A = V^T, which is read
//             from the row-major V tile with ds_read_b64_tr_b16), so P never touches LDS.
//   backward  two kernels, no atomics: dQ (query-stationary, same structure as forward) and
//             dK/dV (key-stationary: keys on the lane, Q/dO tiles streamed through LDS).
// Head dims 16/32/48/64: the QK^T contraction is zero-padded to 32/64, the PV side uses D/16
// exact 16-wide tiles (d = 48 = 3 tiles: no padding waste there).
// LDS images: "row" images ([row][DP] bf16, XOR-swizzled 16-B chunks, conflict-free
// ds_read_b128) and "tr" images ([row][D] plain, for the transposed reads).
#include "attention_internal.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int D> struct ACfg {
  static constexpr int DP = (D <= 32) ? 32 : 64;   // padded contraction width of QK^T / dO V^T
  static constexpr int NKS = DP / 32;              // 32-wide k steps over d
  static constexpr int NDT = D / 16;               // 16-wide d tiles
  static constexpr int CH = D / 8;                 // 16-byte chunks per global row
  static constexpr int CHP = DP / 8;               // 16-byte chunks per padded LDS row
  static constexpr int ROWB = DP * 2;              // bytes per row of a "row" image
  static constexpr int TRB = D * 2;                // bytes per row of a "tr" image
};

template <int D> __device__ __forceinline__ int row_img_off(int row, int chunk) {
  if (ACfg<D>::DP == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  return row * 64 + ((chunk ^ ((3 * (row >> 2)) & 3)) << 4);
}

__device__ __forceinline__ bf8_t as_bf8(uint4 v) { return __builtin_bit_cast(bf8_t, v); }
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ bf8_t lds_b128(const char* base, int off) {
  return as_bf8(*reinterpret_cast<const uint4*>(base + off));
}
// transposed read: lane i of each 16-lane group receives column i of a 4-row x 16-col block;
// lane 4q+p supplies the address of (row q, cols 4p..4p+3)
__device__ __forceinline__ s4_t lds_tr(const char* addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(addr));
}
__device__ __forceinline__ bf8_t cat_tr(s4_t lo, s4_t hi) {
  s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8_t, v);
}
// 8 f32 -> bf16x8 operand: elements 0..3 from a, 4..7 from b
__device__ __forceinline__ bf8_t pack8(f32x4_t a, f32x4_t b) {
  uint4 v;
  v.x = pack_bf16x2(a[0], a[1]); v.y = pack_bf16x2(a[2], a[3]);
  v.z = pack_bf16x2(b[0], b[1]); v.w = pack_bf16x2(b[2], b[3]);
  return as_bf8(v);
}
// 8 bf16 (one uint4) times a scalar, rounded back to bf16 (RNE)
__device__ __forceinline__ uint4 scale_bf16x8(uint4 v, float c) {
  uint4 r;
  r.x = pack_bf16x2(__uint_as_float(v.x << 16) * c, __uint_as_float(v.x & 0xffff0000u) * c);
  r.y = pack_bf16x2(__uint_as_float(v.y << 16) * c, __uint_as_float(v.y & 0xffff0000u) * c);
  r.z = pack_bf16x2(__uint_as_float(v.z << 16) * c, __uint_as_float(v.z & 0xffff0000u) * c);
  r.w = pack_bf16x2(__uint_as_float(v.w << 16) * c, __uint_as_float(v.w & 0xffff0000u) * c);
  return r;
}
__device__ __forceinline__ float bf16_round(float x) { return __uint_as_float(pack_bf16x2(x, 0.f) << 16); }
// x ~= hi + lo with both exactly representable in bf16 (relative error ~2^-17): packed {hi, lo}
__device__ __forceinline__ unsigned int bf16_hi_lo(float x) {
  float hi = bf16_round(x);
  return pack_bf16x2(hi, x - hi);
}
__device__ __forceinline__ float xmax4(float v) {   // max over the 4 lane groups (lanes l, l^16, l^32, l^48)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xsum4(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ int xcd_remap(int bid, int nb) {   // bijective: blocks sharing an XCD get adjacent logical ids
  int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// 8 waves per workgroup, each owning FA_NT = 2 sixteen-wide tiles (32 queries, or 32 keys in the
// dK/dV kernel): ~100-130 VGPRs per wave so several waves share a SIMD and one wave's softmax VALU
// work overlaps another's MFMAs (a 4-tile / 4-wave variant needed >256 registers and spent half
// its instructions on v_accvgpr copies: 7.7k cycles per 64-key tile against 0.9k of MFMA).
#define FA_NT 2
#ifndef FA_WAVES
#define FA_WAVES 4
#endif
#define FA_THREADS (64 * FA_WAVES)
#define FA_WQ (16 * FA_NT)            // rows (queries / keys) per wave
#define FA_BQ (FA_WQ * FA_WAVES)      // 256 rows per workgroup
#define FA_BK 64
#ifndef FA_FWD_PAD
#define FA_FWD_PAD 0
#endif
#ifndef FA_PREFETCH2
#define FA_PREFETCH2 0   // 1: two register stages (tile requested two steps ahead); measured neutral, costs 12-18 VGPRs
#endif
#define FA_IDX_CAP SS_ATTN_MFMA_MAX_WINDOW
#ifndef FA_ABL
#define FA_ABL 0   // ablation bitmask of scripts/ubench/attn_bwd_bench.hip (diagnostic builds only)
#endif   // longest window whose gather rows fit the LDS copy (launchers refuse longer ones)

// stage a 64-row K/V tile (rows gidx[p0+r0 .. +63], column block `colofs`) into registers
// gidx_w: the window's gather rows, copied to LDS once per workgroup (FA_IDX_CAP) -- a per-step global index load
// followed by the dependent row load put a full memory round trip (s_waitcnt vmcnt(0)) inside every step
template <int D, int NLD>
__device__ __forceinline__ void tile_load(uint4 (&reg)[NLD], const unsigned short* __restrict__ qkv,
                                          const int32_t* gidx_w, int r0, int L, int64_t C3,
                                          int colofs_a, int colofs_b, int tid) {
  constexpr int CH = ACfg<D>::CH;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int c = i * FA_THREADS + tid;
    int second = c >= 64 * CH;
    int cc = second ? c - 64 * CH : c;
    int r = cc / CH, ch = cc - r * CH;
    // rows past the window end are clamped to its last row: finite duplicates whose scores are masked
    // (K) or multiplied by p = 0 (V), so no zero fill and no divergent branch
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((2 * 64 * CH) % FA_THREADS == 0 || c < 2 * 64 * CH) {
      // gidx_w holds row * (3C / 8): the row's offset in 16-byte units, so the address is one shift-add
      const uint64_t o16 = (uint32_t)gidx_w[min(r0 + r, L - 1)];
      v = ld16(reinterpret_cast<const char*>(qkv + (second ? colofs_b : colofs_a) + ch * 8) + (o16 << 4));
    }
    reg[i] = v;
  }
}

// =====================================================================================
// forward
// =====================================================================================
// d <= 32 fits 128 VGPRs (4 waves/SIMD, +5 %); forcing d = 48 under 128 spills and loses 7 %, it runs at 144 / 3 waves
template <int D>
__global__ void __launch_bounds__(FA_THREADS, (D <= 32 ? 4 : 1))
k_attn_fwd_mfma(const unsigned short* __restrict__ qkv, const int32_t* __restrict__ gidx,
                const int32_t* __restrict__ sidx, const int32_t* __restrict__ win_start,
                unsigned short* __restrict__ out, float* __restrict__ lse, int C, int H, float scale, int qchunks) {
  using A = ACfg<D>;
  constexpr int NLD = (2 * 64 * A::CH + FA_THREADS - 1) / FA_THREADS;     // 16-B loads per thread per K+V tile
  constexpr int KIMG = 64 * A::ROWB, VIMG = 64 * A::TRB;
  __shared__ __attribute__((aligned(16))) char smem[2 * (KIMG + VIMG)];
  __shared__ int32_t gidx_s[FA_IDX_CAP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qc = lid % qchunks; const int t_ = lid / qchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int q0 = qc * FA_BQ;
  if (q0 >= L) return;
  for (int i = tid; i < L; i += FA_THREADS) gidx_s[i] = (int32_t)((uint32_t)gidx[p0 + i] * (uint32_t)(3 * C >> 3));
  __syncthreads();
  const int64_t C3 = 3 * (int64_t)C;
  const float c2 = scale * 1.44269504088896340736f;
  auto Kbuf = [&](int b_) { return smem + b_ * (KIMG + VIMG); };
  auto Vbuf = [&](int b_) { return smem + b_ * (KIMG + VIMG) + KIMG; };
  // zero the contraction padding of the K images once (chunks CH..CHP-1)
  if (A::CHP > A::CH) {
    for (int e = tid; e < 2 * 64 * (A::CHP - A::CH); e += FA_THREADS) {
      int b = e / (64 * (A::CHP - A::CH)); int r = (e / (A::CHP - A::CH)) % 64; int ch = A::CH + e % (A::CHP - A::CH);
      // FA_FWD_PAD: contraction column D carries a constant 1; with the Q side holding -shift there, the MFMA itself
      // subtracts the softmax shift.  Measured SLOWER in the forward (0.68 -> 0.77 ms at dec0): some query of the wave
      // sees a new maximum on almost every tile, and that path then pays the subtraction plus the fragment update;
      // the backward kernels, whose shifts (lse, delta) are known up front, keep the trick.
      *reinterpret_cast<uint4*>(Kbuf(b) + row_img_off<D>(r, ch)) = make_uint4((FA_FWD_PAD && ch == A::CH) ? 0x3F80u : 0u, 0, 0, 0);
    }
  }
  constexpr bool PAD = FA_FWD_PAD && (A::CHP > A::CH);   // free contraction columns (d = 16, 48)
  constexpr int PADKS = D / 32, PADG = (D % 32) / 8; // fragment position of contraction column D
  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q = lq][d = 32ks + 8g .. +7]; with PAD they are pre-scaled
  // by scale * log2(e) so that S^T comes out of the MFMA in exp2 units
  bf8_t qf[FA_NT][A::NKS];
  int qslot[FA_NT];
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt) {
    int slot = q0 + wave * FA_WQ + qt * 16 + lq;
    qslot[qt] = slot;
    int64_t row = slot < L ? gidx[p0 + slot] : -1;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      int d0 = 32 * ks + 8 * g;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row >= 0 && d0 < D) v = ld16(qkv + row * C3 + h * D + d0);
      if (PAD) v = scale_bf16x8(v, c2);
      qf[qt][ks] = as_bf8(v);
    }
  }
  // row sums ride the matrix pipe: an extra A tile whose row 0 is all ones makes
  // lsum[qt][0] (lane group 0) = sum_k P[k][q] of the same bf16-rounded P the PV product uses;
  // the softmax is VALU-bound at d = 48, so 2 MFMAs per step are cheaper than 32 v_add per tile
  const bf8_t ones = as_bf8(lq == 0 ? make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u) : make_uint4(0, 0, 0, 0));
  float m[FA_NT];
  f32x4_t o[A::NDT][FA_NT], lsum[FA_NT];
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt) {
    m[qt] = PAD ? 0.f : -1e30f; lsum[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};   // PAD: m = current shift in exp2 units
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt) o[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // two register stages: the tile written to LDS at the bottom of step t was requested two steps earlier
  // (one step of ~30 MFMAs does not cover a gathered-row fetch under load)
  uint4 stA[NLD], stB[NLD];
  auto stage_write = [&](uint4 (&stage)[NLD], int b) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int c = i * FA_THREADS + tid;
      int second = c >= 64 * A::CH;
      int cc = second ? c - 64 * A::CH : c;
      int r = cc / A::CH, ch = cc - r * A::CH;
      if (c >= 2 * 64 * A::CH) continue;
      if (second) *reinterpret_cast<uint4*>(Vbuf(b) + r * A::TRB + ch * 16) = stage[i];
      else *reinterpret_cast<uint4*>(Kbuf(b) + row_img_off<D>(r, ch)) = stage[i];
    }
  };
  const int ntiles = (L + FA_BK - 1) / FA_BK;
  tile_load<D, NLD>(stA, qkv, gidx_s, 0, L, C3, C + h * D, 2 * C + h * D, tid);
  stage_write(stA, 0);
  if (FA_PREFETCH2 && ntiles > 1) tile_load<D, NLD>(stA, qkv, gidx_s, FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
  if (FA_PREFETCH2 && ntiles > 2) tile_load<D, NLD>(stB, qkv, gidx_s, 2 * FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
  __syncthreads();
  auto step = [&](const int t, uint4 (&stage)[NLD]) {      // FA_PREFETCH2: stage holds tile t+1
    const int b = t & 1, kv0 = t * FA_BK;
    if (!FA_PREFETCH2 && t + 1 < ntiles) tile_load<D, NLD>(stage, qkv, gidx_s, kv0 + FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
    // ---- S^T = K Q^T : s[kt][qt], rows = keys 16kt + 4g + r, col = query lq
    f32x4_t s[4][FA_NT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < FA_NT; ++qt) s[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) {
        bf8_t a = lds_b128(Kbuf(b), row_img_off<D>(16 * kt + lq, 4 * ks + g));
#pragma unroll
        for (int qt = 0; qt < FA_NT; ++qt) s[kt][qt] = MFMA16(a, qf[qt][ks], s[kt][qt]);
      }
    }
    if (kv0 + FA_BK > L) {   // mask the keys past the window end (last tile only; wave-uniform)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kv0 + 16 * kt + 4 * g + r >= L) {
#pragma unroll
            for (int qt = 0; qt < FA_NT; ++qt) s[kt][qt][r] = -INFINITY;
          }
    }
    // ---- online softmax, lane-local per query column
#pragma unroll
    for (int qt = 0; qt < FA_NT; ++qt) {
      float mx = s[0][qt][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
      mx = xmax4(mx);
      if (PAD) {
        // s is already c2 * S - shift (shift = m[qt], a bf16-exact value held in Q's contraction column D).  Only when a
        // score exceeds the shift (or on the first tile) does the shift move; then this tile pays the subtraction.
        const bool move = (t == 0) || (mx > 0.f);
        if (__any(move)) {
          const float ns = move ? bf16_round(m[qt] + mx) : m[qt];
          const float dlt = ns - m[qt];
          const float alpha = __builtin_amdgcn_exp2f(-dlt);
          lsum[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < A::NDT; ++dt) o[dt][qt] *= alpha;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][qt][r] = __builtin_amdgcn_exp2f(s[kt][qt][r] - dlt);
          m[qt] = ns;
          uint4 u = __builtin_bit_cast(uint4, qf[qt][PADKS]);
          if (g == PADG) u.x = (u.x & 0xffff0000u) | (pack_bf16x2(-ns, 0.f) & 0xffffu);
          qf[qt][PADKS] = as_bf8(u);
        } else {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][qt][r] = __builtin_amdgcn_exp2f(s[kt][qt][r]);
        }
      } else {
      float mn = fmaxf(m[qt], mx);
      // exact skip: when no lane's running max moved, alpha == 1 for the whole wave
      if (__any(mn > m[qt])) {
        float alpha = __builtin_amdgcn_exp2f((m[qt] - mn) * c2);
        lsum[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < A::NDT; ++dt) o[dt][qt] *= alpha;
      }
      m[qt] = mn;
      float mc = mn * c2;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][qt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][r], c2, -mc));
        }
      }
    }
    // ---- O^T += V^T P^T ; k index (g, j) of step kk <-> key 32kk + 16(j>>2) + 4g + (j&3)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf8_t pf[FA_NT];
#pragma unroll
      for (int qt = 0; qt < FA_NT; ++qt) {
        pf[qt] = pack8(s[2 * kk][qt], s[2 * kk + 1][qt]);
        lsum[qt] = MFMA16(ones, pf[qt], lsum[qt]);
      }
      const char* vbase = Vbuf(b) + (32 * kk + 4 * g + (lq >> 2)) * A::TRB + (lq & 3) * 8;
#pragma unroll
      for (int dt = 0; dt < A::NDT; ++dt) {
        bf8_t vf = cat_tr(lds_tr(vbase + dt * 32), lds_tr(vbase + 16 * A::TRB + dt * 32));
#pragma unroll
        for (int qt = 0; qt < FA_NT; ++qt) o[dt][qt] = MFMA16(vf, pf[qt], o[dt][qt]);
      }
    }
    if (t + 1 < ntiles) stage_write(stage, b ^ 1);
    if (FA_PREFETCH2 && t + 3 < ntiles) tile_load<D, NLD>(stage, qkv, gidx_s, kv0 + 3 * FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
    __syncthreads();
  };
  if (FA_PREFETCH2) {
    for (int t = 0; t < ntiles; t += 2) {
      step(t, stA);
      if (t + 1 < ntiles) step(t + 1, stB);
    }
  } else {
    for (int t = 0; t < ntiles; ++t) step(t, stA);
  }
  // ---- epilogue
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt) {
    float lt = __shfl(lsum[qt][0], lq, 64);   // row 0 of the ones tile lives in lane group 0
    int slot = qslot[qt];
    if (slot < L) {
      if (g == 0) lse[(int64_t)(p0 + slot) * H + h] = (PAD ? m[qt] * 0.69314718055994530942f : m[qt] * scale) + __logf(lt);
      int32_t srow = sidx[p0 + slot];
      if (srow >= 0) {
        float inv = 1.f / lt;
        unsigned short* op = out + (int64_t)srow * C + h * D + 4 * g;
#pragma unroll
        for (int dt = 0; dt < A::NDT; ++dt) {
          uint2 v;
          v.x = pack_bf16x2(o[dt][qt][0] * inv, o[dt][qt][1] * inv);
          v.y = pack_bf16x2(o[dt][qt][2] * inv, o[dt][qt][3] * inv);
          *reinterpret_cast<uint2*>(op + 16 * dt) = v;
        }
      }
    }
  }
}

// =====================================================================================
// backward, dQ: query-stationary.  S^T and dP^T = V dO^T (key rows, query on the lane),
// dS^T = P^T o (dP^T - delta_q), dQ^T += K^T dS^T with K^T read transposed from a plain image.
// =====================================================================================
#ifndef FA_BWD_MIN_BLOCKS
#define FA_BWD_MIN_BLOCKS 1   // 3 would cap the backward kernels at 168 VGPRs (3 waves / SIMD)
#endif
template <int D>
__global__ void __launch_bounds__(FA_THREADS, FA_BWD_MIN_BLOCKS)
k_attn_bwd_dq_mfma(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dout,
                   const unsigned short* __restrict__ outp, const float* __restrict__ lse, float* __restrict__ delta,
                   const int32_t* __restrict__ gidx,
                   const int32_t* __restrict__ sidx, const int32_t* __restrict__ win_start,
                   unsigned short* __restrict__ dqkv, int C, int H, float scale, int qchunks) {
  using A = ACfg<D>;
  constexpr int NLD = (2 * 64 * A::CH + FA_THREADS - 1) / FA_THREADS;
  constexpr int RIMG = 64 * A::ROWB, TIMG = 64 * A::TRB;
  constexpr int BUF = 2 * RIMG + TIMG;      // K row image, V row image, K tr image
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  __shared__ int32_t gidx_s[FA_IDX_CAP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qc = lid % qchunks; const int t_ = lid / qchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int q0 = qc * FA_BQ;
  if (q0 >= L) return;
  // ---- prologue in TWO dependent rounds of global loads (it was five: index copy | barrier | query indices | query rows
  // | first tile): a workgroup lives ~40 us and spent a third of it here (dQ 0.84 -> 0.57 ms with the row loads removed).
  // round 1 -- everything addressed by p0 alone: the window's gather offsets, this lane's query indices and lse, and the
  // row indices of this thread's chunks of the first K / V tile
  const int64_t C3 = 3 * (int64_t)C;
  const float c2 = scale * 1.44269504088896340736f;
  constexpr int NFILL = FA_IDX_CAP / FA_THREADS;
  int32_t fillv[NFILL];
#pragma unroll
  for (int k = 0; k < NFILL; ++k) { const int i = tid + k * FA_THREADS; fillv[k] = i < L ? gidx[p0 + i] : 0; }
  int64_t qrow[FA_NT];
  float lse2[FA_NT], dl[FA_NT];
  int32_t srow[FA_NT];
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt) {
    const int slot = q0 + wave * FA_WQ + qt * 16 + lq;
    const bool ok = slot < L;
    qrow[qt] = ok ? gidx[p0 + slot] : -1;
    srow[qt] = ok ? sidx[p0 + slot] : -1;
    lse2[qt] = ok ? lse[(int64_t)(p0 + slot) * H + h] * 1.44269504088896340736f : 0.f;
  }
  uint32_t t0off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = i * FA_THREADS + tid;
    const int cc = c >= 64 * A::CH ? c - 64 * A::CH : c;
    t0off[i] = (uint32_t)gidx[p0 + min(cc / A::CH, L - 1)] * (uint32_t)(3 * C >> 3);
  }
  // round 2 -- the rows those indices name: Q, dO, O of the wave's queries and the first K / V tile
  uint4 qv[FA_NT][A::NKS], gv[FA_NT][A::NKS], ov[FA_NT][A::NKS];
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt)
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      const int d0 = 32 * ks + 8 * g;
      qv[qt][ks] = gv[qt][ks] = ov[qt][ks] = make_uint4(0, 0, 0, 0);
      if (qrow[qt] >= 0 && d0 < D && !(FA_ABL & 128)) qv[qt][ks] = ld16(qkv + qrow[qt] * C3 + h * D + d0);
      if (srow[qt] >= 0 && d0 < D && !(FA_ABL & 128)) {
        gv[qt][ks] = ld16(dout + (int64_t)srow[qt] * C + h * D + d0); ov[qt][ks] = ld16(outp + (int64_t)srow[qt] * C + h * D + d0);
      }
    }
  uint4 stA[NLD], stB[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = i * FA_THREADS + tid;
    const int second = c >= 64 * A::CH;
    const int cc = second ? c - 64 * A::CH : c;
    const int ch = cc - (cc / A::CH) * A::CH;
    stA[i] = make_uint4(0, 0, 0, 0);
    if ((2 * 64 * A::CH) % FA_THREADS == 0 || c < 2 * 64 * A::CH)
      stA[i] = ld16(reinterpret_cast<const char*>(qkv + (second ? 2 * C : C) + h * D + ch * 8) + ((uint64_t)t0off[i] << 4));
  }
  // LDS: gather offsets (16-byte units), contraction padding; the barrier also retires round 2
#pragma unroll
  for (int k = 0; k < NFILL; ++k) { const int i = tid + k * FA_THREADS; if (i < L) gidx_s[i] = (int32_t)((uint32_t)fillv[k] * (uint32_t)(3 * C >> 3)); }
  if (A::CHP > A::CH) {
    for (int e = tid; e < 4 * 64 * (A::CHP - A::CH); e += FA_THREADS) {
      int img = e / (64 * (A::CHP - A::CH)); int r = (e / (A::CHP - A::CH)) % 64; int ch = A::CH + e % (A::CHP - A::CH);
      char* base = smem + (img >> 1) * BUF + (img & 1) * RIMG;
      // contraction columns D, D+1 of the K and V row images carry 1, 1: the query side holds -(lse, delta) there as
      // bf16 hi + lo pairs, so S^T comes out as c2*S - lse2 and dP^T as dP - delta (two VALU ops per score saved)
      *reinterpret_cast<uint4*>(base + row_img_off<D>(r, ch)) = make_uint4(ch == A::CH ? 0x3F803F80u : 0u, 0, 0, 0);
    }
  }
  __syncthreads();
  constexpr bool PAD = A::CHP > A::CH;
  constexpr int PADKS = D / 32, PADG = (D % 32) / 8;
  bf8_t qf[FA_NT][A::NKS], gf[FA_NT][A::NKS];
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt) {
    const int slot = q0 + wave * FA_WQ + qt * 16 + lq;
    const bool ok = slot < L;
    // delta = rowsum(O o dO) of this query is computed HERE (the dO fragment is loaded anyway) and published for the
    // dK/dV kernel, which runs after this one: no separate delta pass over O and dO
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      const uint4 v = qv[qt][ks], u = gv[qt][ks], o = ov[qt][ks];
      qf[qt][ks] = as_bf8(PAD ? scale_bf16x8(v, c2) : v); gf[qt][ks] = as_bf8(u);
      const unsigned int* uu = reinterpret_cast<const unsigned int*>(&u);
      const unsigned int* uo = reinterpret_cast<const unsigned int*>(&o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dsum += __uint_as_float(uu[j] << 16) * __uint_as_float(uo[j] << 16);
        dsum += __uint_as_float(uu[j] & 0xffff0000u) * __uint_as_float(uo[j] & 0xffff0000u);
      }
    }
    dl[qt] = xsum4(dsum);
    if (ok && g == 0) delta[(int64_t)(p0 + slot) * H + h] = dl[qt];
    if (PAD && g == PADG) {
      uint4 u = __builtin_bit_cast(uint4, qf[qt][PADKS]); u.x = bf16_hi_lo(-lse2[qt]); qf[qt][PADKS] = as_bf8(u);
      u = __builtin_bit_cast(uint4, gf[qt][PADKS]); u.x = bf16_hi_lo(-dl[qt]); gf[qt][PADKS] = as_bf8(u);
    }
  }
  f32x4_t dq[A::NDT][FA_NT];
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt)
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt) dq[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto stage_write = [&](uint4 (&stage)[NLD], int b) {
    char* base = smem + b * BUF;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int c = i * FA_THREADS + tid;
      int second = c >= 64 * A::CH;
      int cc = second ? c - 64 * A::CH : c;
      int r = cc / A::CH, ch = cc - r * A::CH;
      if (c >= 2 * 64 * A::CH) continue;
      if (second) *reinterpret_cast<uint4*>(base + RIMG + row_img_off<D>(r, ch)) = stage[i];
      else {
        *reinterpret_cast<uint4*>(base + row_img_off<D>(r, ch)) = stage[i];
        *reinterpret_cast<uint4*>(base + 2 * RIMG + r * A::TRB + ch * 16) = stage[i];
      }
    }
  };
  const int ntiles = (L + FA_BK - 1) / FA_BK;
  stage_write(stA, 0);                         // tile 0 was fetched in round 2 of the prologue
  if (FA_PREFETCH2 && ntiles > 1) tile_load<D, NLD>(stA, qkv, gidx_s, FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
  if (FA_PREFETCH2 && ntiles > 2) tile_load<D, NLD>(stB, qkv, gidx_s, 2 * FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
  __syncthreads();
  auto step = [&](const int t, uint4 (&stage)[NLD]) {        // FA_PREFETCH2: stage holds tile t+1
    const int b = t & 1, kv0 = t * FA_BK;
    const char* Kr = smem + b * BUF; const char* Vr = Kr + RIMG; const char* Kt = Kr + 2 * RIMG;
    if (!FA_PREFETCH2 && t + 1 < ntiles && !(FA_ABL & 1)) tile_load<D, NLD>(stage, qkv, gidx_s, kv0 + FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
    f32x4_t s[4][FA_NT], dp[4][FA_NT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < FA_NT; ++qt) { s[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) {
        int off = row_img_off<D>(16 * kt + lq, 4 * ks + g);
        bf8_t ka = lds_b128(Kr, off), va = lds_b128(Vr, off);
#pragma unroll
        for (int qt = 0; qt < FA_NT; ++qt) {
          s[kt][qt] = MFMA16(ka, qf[qt][ks], s[kt][qt]);
          dp[kt][qt] = MFMA16(va, gf[qt][ks], dp[kt][qt]);
        }
      }
    }
    const bool tail = kv0 + FA_BK > L;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < FA_NT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p, ds;
          if (PAD) { p = __builtin_amdgcn_exp2f(s[kt][qt][r]); ds = p * dp[kt][qt][r]; }
          else { p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][r], c2, -lse2[qt])); ds = p * (dp[kt][qt][r] - dl[qt]); }
          if (tail && kv0 + 16 * kt + 4 * g + r >= L) ds = 0.f;
          s[kt][qt][r] = ds;
        }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf8_t df[FA_NT];
#pragma unroll
      for (int qt = 0; qt < FA_NT; ++qt) df[qt] = pack8(s[2 * kk][qt], s[2 * kk + 1][qt]);
      const char* kbase = Kt + (32 * kk + 4 * g + (lq >> 2)) * A::TRB + (lq & 3) * 8;
#pragma unroll
      for (int dt = 0; dt < A::NDT; ++dt) {
        bf8_t kf = cat_tr(lds_tr(kbase + dt * 32), lds_tr(kbase + 16 * A::TRB + dt * 32));
#pragma unroll
        for (int qt = 0; qt < FA_NT; ++qt) dq[dt][qt] = MFMA16(kf, df[qt], dq[dt][qt]);
      }
    }
    if (t + 1 < ntiles && !(FA_ABL & 1)) stage_write(stage, b ^ 1);
    if (FA_PREFETCH2 && t + 3 < ntiles) tile_load<D, NLD>(stage, qkv, gidx_s, kv0 + 3 * FA_BK, L, C3, C + h * D, 2 * C + h * D, tid);
    __syncthreads();
  };
  if (FA_PREFETCH2) {
    for (int t = 0; t < ntiles; t += 2) {
      step(t, stA);
      if (t + 1 < ntiles) step(t + 1, stB);
    }
  } else {
    for (int t = 0; t < ntiles; ++t) step(t, stA);
  }
#pragma unroll
  for (int qt = 0; qt < FA_NT; ++qt) {
    if (srow[qt] >= 0 && !(FA_ABL & 64)) {
      unsigned short* op = dqkv + (int64_t)srow[qt] * C3 + h * D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < A::NDT; ++dt) {
        uint2 v;
        v.x = pack_bf16x2(dq[dt][qt][0] * scale, dq[dt][qt][1] * scale);
        v.y = pack_bf16x2(dq[dt][qt][2] * scale, dq[dt][qt][3] * scale);
        *reinterpret_cast<uint2*>(op + 16 * dt) = v;
      }
    }
  }
}

// =====================================================================================
// backward, dK/dV: key-stationary.  Each wave owns 64 keys (K, V fragments in registers, key on
// the lane); 32-query tiles of Q and dO stream through LDS (row image for S / dP, tr image for
// dV^T += dO^T P and dK^T += Q^T dS).
// =====================================================================================
#ifndef FA_BQ2
#define FA_BQ2 64     // queries per step: two 32-query halves between barriers
#endif
#ifndef FA_DKV_WAVES
#define FA_DKV_WAVES 4      // waves (32 keys each) per dK/dV workgroup
#endif
#define DKV_THREADS (64 * FA_DKV_WAVES)
#define DKV_BKEYS (FA_WQ * FA_DKV_WAVES)
template <int D>
__global__ void __launch_bounds__(DKV_THREADS, (FA_DKV_WAVES == 4 ? FA_BWD_MIN_BLOCKS : 1))
k_attn_bwd_dkv_mfma(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dout,
                    const float* __restrict__ lse, const float* __restrict__ delta, const int32_t* __restrict__ gidx,
                    const int32_t* __restrict__ sidx, const int32_t* __restrict__ win_start,
                    unsigned short* __restrict__ dqkv, unsigned short* __restrict__ extra, int C, int H, float scale,
                    int kchunks) {
  using A = ACfg<D>;
  constexpr int TOT = 2 * FA_BQ2 * A::CH;                     // 16-B chunks per (Q, dO) tile
  constexpr int NLD = (TOT + DKV_THREADS - 1) / DKV_THREADS;
  constexpr int RIMG = FA_BQ2 * A::ROWB, TIMG = FA_BQ2 * A::TRB;
  constexpr int BUF = 2 * RIMG + 2 * TIMG + 2 * FA_BQ2 * 4;   // Q row, dO row, Q tr, dO tr, lse2[32], delta[32]
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  __shared__ int32_t gidx_s[FA_IDX_CAP], sidx_s[FA_IDX_CAP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int kc = lid % kchunks; const int t_ = lid / kchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int k0 = kc * DKV_BKEYS;
  if (k0 >= L) return;
  // ---- prologue in TWO dependent rounds of global loads (was four: index copies | barrier | key indices | K / V rows,
  // first Q / dO tile): round 1 = everything addressed by p0 alone, round 2 = the rows those indices name
  const int64_t C3 = 3 * (int64_t)C;
  const float c2 = scale * 1.44269504088896340736f;
  constexpr bool PAD = A::CHP > A::CH;
  constexpr int PADKS = D / 32, PADG = (D % 32) / 8;
  constexpr int NFILL = FA_IDX_CAP / DKV_THREADS;
  int32_t fillg[NFILL], fills[NFILL];
#pragma unroll
  for (int k = 0; k < NFILL; ++k) {
    const int i = tid + k * DKV_THREADS;
    fillg[k] = i < L ? gidx[p0 + i] : 0; fills[k] = i < L ? sidx[p0 + i] : -1;
  }
  int kslot[FA_NT];
  int32_t ksr[FA_NT];
  int64_t krow[FA_NT];
#pragma unroll
  for (int kt = 0; kt < FA_NT; ++kt) {
    kslot[kt] = k0 + wave * FA_WQ + kt * 16 + lq;
    krow[kt] = kslot[kt] < L ? gidx[p0 + kslot[kt]] : -1;
    ksr[kt] = kslot[kt] < L ? sidx[p0 + kslot[kt]] : -1;      // destination row of the key (read in the epilogue): fetched now
  }
  struct Stage { uint4 v[NLD]; float l, d; };
  Stage stA, stB;
  stA.l = stA.d = stB.l = stB.d = 0.f;
  int32_t s0idx[NLD];                       // first (Q, dO) tile: row index of this thread's chunks (Q: gather row, dO: scatter row)
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = i * DKV_THREADS + tid;
    const int second = c >= FA_BQ2 * A::CH;
    const int r = (second ? c - FA_BQ2 * A::CH : c) / A::CH;
    s0idx[i] = (c < TOT && r < L) ? (second ? sidx[p0 + r] : gidx[p0 + r]) : -1;
  }
  if (tid < FA_BQ2) {
    const bool ok = tid < L;
    stA.l = ok ? lse[(int64_t)(p0 + tid) * H + h] * 1.44269504088896340736f : 1e30f;   // p = 0 for rows past the window
    stA.d = ok ? delta[(int64_t)(p0 + tid) * H + h] : 0.f;
  }
  // round 2
  uint4 ka[FA_NT][A::NKS], vb[FA_NT][A::NKS];
#pragma unroll
  for (int kt = 0; kt < FA_NT; ++kt)
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      const int d0 = 32 * ks + 8 * g;
      ka[kt][ks] = vb[kt][ks] = make_uint4(0, 0, 0, 0);
      if (krow[kt] >= 0 && d0 < D && !(FA_ABL & 128)) {
        ka[kt][ks] = ld16(qkv + krow[kt] * C3 + C + h * D + d0); vb[kt][ks] = ld16(qkv + krow[kt] * C3 + 2 * C + h * D + d0);
      }
    }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = i * DKV_THREADS + tid;
    const int second = c >= FA_BQ2 * A::CH;
    const int cc = second ? c - FA_BQ2 * A::CH : c;
    const int ch = cc - (cc / A::CH) * A::CH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (s0idx[i] >= 0) {
      if (!second) v = ld16(qkv + (int64_t)s0idx[i] * C3 + h * D + ch * 8);
      else v = ld16(dout + (int64_t)s0idx[i] * C + h * D + ch * 8);
    }
    stA.v[i] = v;
  }
  // LDS: row offsets in 16-byte units (row * 3C/8, row * C/8; borrowed slots stay < 0), contraction padding
#pragma unroll
  for (int k = 0; k < NFILL; ++k) {
    const int i = tid + k * DKV_THREADS;
    if (i < L) { gidx_s[i] = (int32_t)((uint32_t)fillg[k] * (uint32_t)(3 * C >> 3)); sidx_s[i] = fills[k] >= 0 ? fills[k] * (C >> 3) : -1; }
  }
  if (A::CHP > A::CH) {
    for (int e = tid; e < 4 * FA_BQ2 * (A::CHP - A::CH); e += DKV_THREADS) {
      int img = e / (FA_BQ2 * (A::CHP - A::CH)); int r = (e / (A::CHP - A::CH)) % FA_BQ2; int ch = A::CH + e % (A::CHP - A::CH);
      char* base = smem + (img >> 1) * BUF + (img & 1) * RIMG;
      *reinterpret_cast<uint4*>(base + row_img_off<D>(r, ch)) = make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  // K / V fragments as B operands: lane holds K[key = lq][d = 32ks + 8g ..]
  bf8_t kf[FA_NT][A::NKS], vf[FA_NT][A::NKS];
#pragma unroll
  for (int kt = 0; kt < FA_NT; ++kt)
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      uint4 a = ka[kt][ks], b = vb[kt][ks];
      if (PAD) {
        // K pre-scaled by scale*log2(e); contraction columns D, D+1 of both fragments carry 1, 1: the streamed Q / dO rows
        // hold -(lse2) / -(delta) there as bf16 hi + lo, so the MFMAs deliver c2*S - lse2 and dP - delta directly
        a = scale_bf16x8(a, c2);
        if (ks == PADKS && g == PADG) { a.x = 0x3F803F80u; b.x = 0x3F803F80u; }
      }
      kf[kt][ks] = as_bf8(a); vf[kt][ks] = as_bf8(b);
    }
  f32x4_t dk[A::NDT][FA_NT], dv[A::NDT][FA_NT];
#pragma unroll
  for (int kt = 0; kt < FA_NT; ++kt)
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt) { dk[dt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  // two register stages (tile written at the bottom of step t was requested two steps earlier): a 32-query step
  // is ~28 MFMAs per wave, far shorter than a gathered-row fetch, and only 2 workgroups fit a CU
  auto stage_load = [&](Stage& st, int qb) {
    uint4 (&stage)[NLD] = st.v; float& st_l = st.l; float& st_d = st.d;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int c = i * DKV_THREADS + tid;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c < TOT) {
        int second = c >= FA_BQ2 * A::CH;
        int cc = second ? c - FA_BQ2 * A::CH : c;
        int r = cc / A::CH, ch = cc - r * A::CH;
        if (qb + r < L) {
          if (!second) v = ld16(reinterpret_cast<const char*>(qkv + h * D + ch * 8) + ((uint64_t)(uint32_t)gidx_s[qb + r] << 4));
          else { int32_t sr = sidx_s[qb + r]; if (sr >= 0) v = ld16(reinterpret_cast<const char*>(dout + h * D + ch * 8) + ((uint64_t)(uint32_t)sr << 4)); }
        }
      }
      stage[i] = v;
    }
    if (tid < FA_BQ2) {
      bool ok = qb + tid < L;
      st_l = ok ? lse[(int64_t)(p0 + qb + tid) * H + h] * 1.44269504088896340736f : 1e30f;   // p = 0 for rows past the window
      st_d = ok ? delta[(int64_t)(p0 + qb + tid) * H + h] : 0.f;
    }
  };
  auto stage_write = [&](Stage& st, int b) {
    uint4 (&stage)[NLD] = st.v; const float st_l = st.l, st_d = st.d;
    char* base = smem + b * BUF;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int c = i * DKV_THREADS + tid;
      if (c < TOT) {
        int second = c >= FA_BQ2 * A::CH;
        int cc = second ? c - FA_BQ2 * A::CH : c;
        int r = cc / A::CH, ch = cc - r * A::CH;
        *reinterpret_cast<uint4*>(base + second * RIMG + row_img_off<D>(r, ch)) = stage[i];
        *reinterpret_cast<uint4*>(base + 2 * RIMG + second * TIMG + r * A::TRB + ch * 16) = stage[i];
      }
    }
    if (tid < FA_BQ2) {
      if (PAD) {      // first pad chunk of the row's Q / dO images: {-lse2 hi, lo} / {-delta hi, lo}
        *reinterpret_cast<uint4*>(base + row_img_off<D>(tid, A::CH)) = make_uint4(bf16_hi_lo(-st_l), 0, 0, 0);
        *reinterpret_cast<uint4*>(base + RIMG + row_img_off<D>(tid, A::CH)) = make_uint4(bf16_hi_lo(-st_d), 0, 0, 0);
      } else {
        float* f = reinterpret_cast<float*>(base + 2 * RIMG + 2 * TIMG);
        f[tid] = st_l; f[FA_BQ2 + tid] = st_d;
      }
    }
  };
  const int ntiles = (L + FA_BQ2 - 1) / FA_BQ2;
  stage_write(stA, 0);                         // the first tile was fetched in round 2 of the prologue
  if (FA_PREFETCH2 && ntiles > 1) stage_load(stA, FA_BQ2);
  if (FA_PREFETCH2 && ntiles > 2) stage_load(stB, 2 * FA_BQ2);
  __syncthreads();
  auto step = [&](const int t, Stage& st) {                  // FA_PREFETCH2: st holds tile t+1
    const int b = t & 1;
    if (!FA_PREFETCH2 && t + 1 < ntiles && !(FA_ABL & 1)) stage_load(st, (t + 1) * FA_BQ2);
    const char* Qr = smem + b * BUF; const char* Gr = Qr + RIMG; const char* Qt = Qr + 2 * RIMG; const char* Gt = Qt + TIMG;
    const float* fl = reinterpret_cast<const float*>(Qr + 2 * RIMG + 2 * TIMG);
#pragma unroll
    for (int hq = 0; hq < FA_BQ2 / 32; ++hq) {
    const char* Qr_h = Qr + hq * 32 * A::ROWB; const char* Gr_h = Gr + hq * 32 * A::ROWB;
    // S[q][key], dP[q][key]: rows = queries 32hq + 16qt + 4g + r, col = key lq (tile kt)
    f32x4_t s[2][FA_NT], dp[2][FA_NT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int kt = 0; kt < FA_NT; ++kt) { s[qt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[qt][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) {
        int off = row_img_off<D>(16 * qt + lq, 4 * ks + g);
        bf8_t qa = lds_b128(Qr_h, off), ga = lds_b128(Gr_h, off);
#pragma unroll
        for (int kt = 0; kt < FA_NT; ++kt) {
          s[qt][kt] = MFMA16(qa, kf[kt][ks], s[qt][kt]);
          dp[qt][kt] = MFMA16(ga, vf[kt][ks], dp[qt][kt]);
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float l2 = 0.f, dd = 0.f;
        if (!PAD) { l2 = fl[32 * hq + 16 * qt + 4 * g + r]; dd = fl[FA_BQ2 + 32 * hq + 16 * qt + 4 * g + r]; }
#pragma unroll
        for (int kt = 0; kt < FA_NT; ++kt) {
          float p = PAD ? __builtin_amdgcn_exp2f(s[qt][kt][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][kt][r], c2, -l2));
          s[qt][kt][r] = p;
          dp[qt][kt][r] = PAD ? p * dp[qt][kt][r] : p * (dp[qt][kt][r] - dd);
        }
      }
    // dV^T += dO^T P ; dK^T += Q^T dS ; k index (g, j) <-> query 16(j>>2) + 4g + (j&3)
    {
      const char* gbase = Gt + (32 * hq + 4 * g + (lq >> 2)) * A::TRB + (lq & 3) * 8;
      const char* qbase = Qt + (32 * hq + 4 * g + (lq >> 2)) * A::TRB + (lq & 3) * 8;
      bf8_t pf[FA_NT], df[FA_NT];
#pragma unroll
      for (int kt = 0; kt < FA_NT; ++kt) { pf[kt] = pack8(s[0][kt], s[1][kt]); df[kt] = pack8(dp[0][kt], dp[1][kt]); }
#pragma unroll
      for (int dt = 0; dt < A::NDT; ++dt) {
        bf8_t ga = cat_tr(lds_tr(gbase + dt * 32), lds_tr(gbase + 16 * A::TRB + dt * 32));
        bf8_t qa = cat_tr(lds_tr(qbase + dt * 32), lds_tr(qbase + 16 * A::TRB + dt * 32));
#pragma unroll
        for (int kt = 0; kt < FA_NT; ++kt) {
          dv[dt][kt] = MFMA16(ga, pf[kt], dv[dt][kt]);
          dk[dt][kt] = MFMA16(qa, df[kt], dk[dt][kt]);
        }
      }
    }
    }   // hq
    if (t + 1 < ntiles && !(FA_ABL & 1)) stage_write(st, b ^ 1);
    if (FA_PREFETCH2 && t + 3 < ntiles) stage_load(st, (t + 3) * FA_BQ2);
    __syncthreads();
  };
  if (FA_PREFETCH2) {
    for (int t = 0; t < ntiles; t += 2) {
      step(t, stA);
      if (t + 1 < ntiles) step(t + 1, stB);
    }
  } else {
    for (int t = 0; t < ntiles; ++t) step(t, stA);
  }
#pragma unroll
  for (int kt = 0; kt < FA_NT; ++kt) {
    int slot = kslot[kt];
    if (slot < L && !(FA_ABL & 64)) {
      int32_t sr = ksr[kt];
      unsigned short* kp; unsigned short* vp;
      if (sr >= 0) { kp = dqkv + (int64_t)sr * C3 + C + h * D + 4 * g; vp = kp + C; }
      else { kp = extra + (int64_t)(-1 - sr) * 2 * C + h * D + 4 * g; vp = kp + C; }
#pragma unroll
      for (int dt = 0; dt < A::NDT; ++dt) {
        uint2 a, b2;
        a.x = pack_bf16x2(dk[dt][kt][0] * scale, dk[dt][kt][1] * scale);
        a.y = pack_bf16x2(dk[dt][kt][2] * scale, dk[dt][kt][3] * scale);
        b2.x = pack_bf16x2(dv[dt][kt][0], dv[dt][kt][1]);
        b2.y = pack_bf16x2(dv[dt][kt][2], dv[dt][kt][3]);
        *reinterpret_cast<uint2*>(kp + 16 * dt) = a;
        *reinterpret_cast<uint2*>(vp + 16 * dt) = b2;
      }
    }
  }
}

// =====================================================================================
// host launchers
// =====================================================================================
int ss_attn_fwd_mfma(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int W,
                     int max_window, void* out, float* lse, int C, int H, float scale, hipStream_t st) {
  const int D = C / H;
  if ((C & 7) || max_window <= 0 || max_window > FA_IDX_CAP) return SS_ERR_ARG;
  // SS_ATTN_FWD32=0 (diagnostic A/B switch, read once) keeps the 16x16x32 forward
  static const int use32 = [] { const char* e = getenv("SS_ATTN_FWD32"); return e ? atoi(e) : 1; }();
  if (use32) return ss_attn_fwd_mfma32(qkv, gidx, sidx, win_start, W, max_window, out, lse, C, H, scale, st);
  const int qchunks = (max_window + FA_BQ - 1) / FA_BQ;
  dim3 g((unsigned)(W * H * qchunks)), b(FA_THREADS);
  const unsigned short* q = (const unsigned short*)qkv; unsigned short* o = (unsigned short*)out;
  switch (D) {
    case 16: SS_LAUNCH((k_attn_fwd_mfma<16>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    case 32: SS_LAUNCH((k_attn_fwd_mfma<32>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    case 48: SS_LAUNCH((k_attn_fwd_mfma<48>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    case 64: SS_LAUNCH((k_attn_fwd_mfma<64>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    default: return SS_ERR_ARG;
  }
  return SS_OK;
}

int ss_attn_bwd_mfma(const void* qkv, const void* dout, const void* out, const float* lse, float* delta, const int32_t* gidx,
                     const int32_t* sidx, const int32_t* win_start, int W, int max_window, void* dqkv, void* extra,
                     int C, int H, float scale, hipStream_t st) {
  const int D = C / H;
  if ((C & 7) || max_window <= 0 || max_window > FA_IDX_CAP) return SS_ERR_ARG;
  const int chunks = (max_window + FA_BQ - 1) / FA_BQ;
  dim3 g((unsigned)(W * H * chunks)), b(FA_THREADS);
  const int chunks2 = (max_window + DKV_BKEYS - 1) / DKV_BKEYS;
  dim3 g2((unsigned)(W * H * chunks2)), b2(DKV_THREADS);
  const unsigned short* q = (const unsigned short*)qkv; const unsigned short* go = (const unsigned short*)dout;
  unsigned short* dq = (unsigned short*)dqkv; unsigned short* ex = (unsigned short*)extra;
#define SS_MB_CASE(DD)                                                                                              \
  case DD:                                                                                                          \
    SS_LAUNCH((k_attn_bwd_dq_mfma<DD>), g, b, 0, st, q, go, (const unsigned short*)out, lse, delta, gidx, sidx, win_start, dq, C, H, scale, chunks); \
    SS_LAUNCH((k_attn_bwd_dkv_mfma<DD>), g2, b2, 0, st, q, go, lse, (const float*)delta, gidx, sidx, win_start, dq, ex, C, H, scale, chunks2); \
    break;
  switch (D) {
    SS_MB_CASE(16) SS_MB_CASE(32) SS_MB_CASE(48) SS_MB_CASE(64)
    default: return SS_ERR_ARG;
  }
#undef SS_MB_CASE
  return SS_OK;
}
