// Serialized-window attention forward on v_mfma_f32_32x32x16_bf16 (gfx950).
//
// Same ABI and the same dataflow as attention_mfma.hip (S^T = K Q^T with the query on the lane, the online
// softmax lane-local, P^T handed to O^T += V^T P^T straight from the accumulators), re-tiled for the 32x32
// matrix instruction because the 16x16x32 kernel is bound by vector ISSUE, not by the matrix pipe (PMC: the wave's
// issue slots are ~100 % busy while the pipe is 39 % busy): a 32x32x16 MFMA does twice the work per 8 issue cycles.
//   * QK^T contracts over d in steps of 16: d = 48 is three exact steps (the 16x16x32 form padded 48 -> 64);
//   * PV has M = d in tiles of 32, so d = 48 pays for 64 rows -- the spare rows are put to work: column 48 of the V
//     image holds 1.0, which makes row 48 of O^T the softmax denominator (sum_k P) of exactly the bf16-rounded P the
//     product uses; no v_add chain, no extra MFMA, and the lazy rescale of O^T rescales it too;
//   * the running maximum moves only when a tile's maximum exceeds it by more than FA32_THR (exp2 units): after the
//     first tiles the O^T rescale (32 multiplies) is skipped almost always; P <= 2^THR keeps bf16's relative precision;
//   * one accumulator tile = 32 queries x 32 keys per wave; FA32_WAVES = 8 waves = 256 queries per workgroup, 64 keys per barrier;
//   * K image rows padded to an odd number of 16-byte chunks (conflict-free ds_read_b128 without a swizzle), V image
//     rows at a stride of 192 B (64 B for d <= 32): the four rows of a transposed read land in distinct 64-byte bank
//     groups (ds_read_b64_tr_b16 conflict-free);
//   * staging addresses: per thread three (row, column) pairs fixed before the loop, the window's gather offsets
//     sit in LDS padded to a multiple of 64 rows: one ds_read + one 64-bit shift-add per 16-byte load.
#include "attention_internal.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;   // native 16-byte vector: staging registers

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#ifndef FA32_THR
#define FA32_THR 6.0f          // lazy-rescale threshold in exp2 units (P <= 64)
#endif
#ifndef FA32_WAVES
#define FA32_WAVES 8           // waves per workgroup: 256 queries share one staging of every K / V tile
#endif
#ifndef FA32_NQB
#define FA32_NQB 1             // 32-query blocks per wave (K / V fragments are reused across them)
#endif
#ifndef FA32_PIPE
#define FA32_PIPE 0            // 1: S(t+1) = K Q^T is issued beside the softmax of tile t (software pipeline inside the wave)
#endif
#ifndef FA32_MINWAVES
#define FA32_MINWAVES 4        // __launch_bounds__ second argument for d <= 48 (waves per SIMD the register allocation must allow)
#endif
// Measured at dec0 (100 windows x 16 heads, K = 1024, d = 48; scripts/ubench/attn_bench.hip, one device, ms per launch):
//   4 waves x 32 q, plain loop 0.611 | + in-wave pipeline 0.645 | 8 waves x 32 q, plain 0.559 (default) | 8 waves, pipeline 0.724
//   4 waves x 64 q (2 q-blocks per wave) 0.566 | 8 waves x 64 q 0.591 | 16x16x32 kernel of round 1: 0.657
// Ablations of the pipelined 4-wave build are ADDITIVE (no component hides under another): global gather + LDS staging
// 0.14, PV MFMAs + transposed reads 0.13, QK^T 0.085, exp 0.044, max 0.034, loop skeleton + prologue / epilogue 0.22 ms.
// The gather is the largest separable piece: every (window, head) workgroup fetches 96-byte slices of 4,608-byte rows
// (1-2 cache lines each, ~45 % of every line used) and the q-chunk workgroups of a window fetch them again.
#define FA32_THREADS (64 * FA32_WAVES)
#define FA32_WQ (32 * FA32_NQB)       // queries per wave
#define FA32_BQ (FA32_WQ * FA32_WAVES) // queries per workgroup
#define FA32_BK 64                    // keys per barrier
#define FA32_IDX_CAP SS_ATTN_MFMA_MAX_WINDOW
#ifndef FA32_ABL
#define FA32_ABL 0     // ablation bitmask of scripts/ubench/attn_bench.hip (diagnostic builds only; 0 in the library)
#endif

#ifdef FA32_CLOCK
// diagnostic build only (scripts/ubench/attn_bench.hip -DFA32_CLOCK): shader-clock and 100 MHz real-time stamps around the
// tile loop of wave 0 of every workgroup -> in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
__device__ unsigned long long fa32_stamp[4 * 32768];
#endif

template <int D> struct A32 {
  static constexpr int NKS = D / 16;                      // QK^T contraction steps
  static constexpr int NMT = (D + 31) / 32;               // 32-row tiles of O^T
  static constexpr int DV = NMT * 32;                     // padded PV rows
  static constexpr bool PADCOL = DV > D;                  // spare row D of O^T carries the row sums
  static constexpr int CH = D / 8;                        // 16-byte chunks per global row
  static constexpr int KROW = (CH + 1 + (CH & 1)) * 16;   // odd chunk count: conflict-free b128 reads
  static constexpr int VROW = DV == 32 ? 64 : 192;        // odd multiple of 64 B
  static constexpr int KIMG = FA32_BK * KROW, VIMG = FA32_BK * VROW;
};

__device__ __forceinline__ bf8_t as_bf8_(uint4 v) { return __builtin_bit_cast(bf8_t, v); }
__device__ __forceinline__ s4_t lds_tr_(const char* addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(addr)); }
__device__ __forceinline__ bf8_t cat_tr_(s4_t lo, s4_t hi) {
  s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8_t, v);
}
__device__ __forceinline__ float max3_(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float swap32_(float v) {     // value of lane ^ 32
  unsigned int u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ int xcd_remap_(int bid, int nb) {
  int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// issue the 16-byte loads of a K tile (rows rK ..) and a V tile (rows rV ..) into registers; they are written to LDS
// after the compute phases (global latency hides under the MFMAs)
template <int NLD>
__device__ __forceinline__ void fa32_tile_load(u32x4_t (&stg)[NLD], const int32_t* gidx_s, int rK, int rV, const int (&st_row)[NLD],
                                               const int (&st_isv)[NLD], const char* const (&st_src)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const uint64_t o16 = (uint32_t)gidx_s[(st_isv[i] ? rV : rK) + st_row[i]];
    stg[i] = *reinterpret_cast<const u32x4_t*>(st_src[i] + (o16 << 4));
  }
}
template <int NLD>
__device__ __forceinline__ void fa32_tile_write(const u32x4_t (&stg)[NLD], char* kimg, char* vimg, const int (&st_isv)[NLD],
                                                const int (&st_lds)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i)
    if (st_lds[i] >= 0) *reinterpret_cast<u32x4_t*>((st_isv[i] ? vimg : kimg) + st_lds[i]) = stg[i];
}

// Software pipeline inside a wave (the 3 co-resident waves alone left the matrix pipe 30 % busy: each wave's tile was one
// serial chain QK^T -> max -> exp -> PV, SQ counters: 1/3 of the wave's cycles parked, 1/3 issue-stalled): iteration t
//     phase 1   S(t+1) = K(t+1) Q^T   [6 MFMA]   beside   exp / pack of block 0 of tile t
//     phase 2   O^T += V(t)^T P_0^T   [4 MFMA]   beside   exp / pack of block 1 of tile t
//     phase 3   O^T += V(t)^T P_1^T   [4 MFMA]   beside   max of S(t+1), the (rare) rescale decision
// so every MFMA group has independent vector work next to it.  K is staged two tiles ahead, V one (K image t&1 and V image
// (t+1)&1 are free during iteration t), one barrier per iteration.
template <int D>
__global__ void __launch_bounds__(FA32_THREADS, (D <= 48 ? FA32_MINWAVES : 2))
k_attn_fwd_mfma32(const unsigned short* __restrict__ qkv, const int32_t* __restrict__ gidx,
                  const int32_t* __restrict__ sidx, const int32_t* __restrict__ win_start,
                  unsigned short* __restrict__ out, float* __restrict__ lse, int C, int H, float scale, int qchunks) {
  using A = A32<D>;
  constexpr int TOT = 2 * FA32_BK * A::CH;                                  // 16-B chunks of one K + V tile
  constexpr int NLD = (TOT + FA32_THREADS - 1) / FA32_THREADS;
  __shared__ __attribute__((aligned(16))) char smem[2 * (A::KIMG + A::VIMG)];
  __shared__ int32_t gidx_s[FA32_IDX_CAP + FA32_BK];
#ifdef FA32_EXTRA_LDS
  __shared__ int32_t occupancy_pad[FA32_EXTRA_LDS / 4];      // diagnostic: lowers the workgroups per CU
  if (threadIdx.x == 0 && qchunks < 0) occupancy_pad[0] = 1;
#endif
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hh = lane >> 5;
  const int lid = xcd_remap_(blockIdx.x, gridDim.x);
  const int qc = lid % qchunks; const int t_ = lid / qchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int q0 = qc * FA32_BQ;
  if (q0 >= L) return;
  const int Lpad = (L + FA32_BK - 1) / FA32_BK * FA32_BK;
  // row offsets in 16-byte units; slots past the window end repeat its last row (finite rows whose scores are masked)
  for (int i = tid; i < Lpad; i += FA32_THREADS) gidx_s[i] = (int32_t)((uint32_t)gidx[p0 + min(i, L - 1)] * (uint32_t)(3 * C >> 3));
  auto Kimg = [&](int b_) { return smem + b_ * A::KIMG; };
  auto Vimg = [&](int b_) { return smem + 2 * A::KIMG + b_ * A::VIMG; };
  if (A::PADCOL) {   // V columns D .. DV-1: column D = 1.0 (row sums), the rest 0; written once, staging never touches them
    constexpr int PCH = (A::DV - D) / 8;
    for (int e = tid; e < 2 * FA32_BK * PCH; e += FA32_THREADS) {
      int b = e / (FA32_BK * PCH), r = (e / PCH) % FA32_BK, ch = e % PCH;
      *reinterpret_cast<uint4*>(Vimg(b) + r * A::VROW + D * 2 + ch * 16) = make_uint4(ch == 0 ? 0x3F80u : 0u, 0, 0, 0);
    }
  }
  const int64_t C3 = 3 * (int64_t)C;
  const float c2 = scale * 1.44269504088896340736f;
  // ---- staging plan of this thread: chunk c = i * THREADS + tid -> (K | V, row, 16-byte column)
  int st_row[NLD], st_lds[NLD], st_isv[NLD];
  const char* st_src[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int c = i * FA32_THREADS + tid;
    bool live = c < TOT;
    int second = c >= FA32_BK * A::CH;
    int cc = second ? c - FA32_BK * A::CH : c;
    int r = live ? cc / A::CH : 0, ch = live ? cc - r * A::CH : 0;
    st_row[i] = r; st_isv[i] = second;
    st_lds[i] = live ? (second ? r * A::VROW + ch * 16 : r * A::KROW + ch * 16) : -1;
    st_src[i] = reinterpret_cast<const char*>(qkv + (second ? 2 * C : C) + h * D + ch * 8);
  }
  u32x4_t stg[NLD];
  // ---- Q fragments (B operand of S^T = K Q^T): lane (query lr, half hh) of q-block qb holds Q[q][16 ks + 8 hh .. +7]
  bf8_t qf[FA32_NQB][A::NKS];
  int qslot[FA32_NQB];
  int32_t srow_[FA32_NQB];            // destination row of the query, fetched now: a load in the epilogue would be a full
                                      // memory round trip at the end of every workgroup
#pragma unroll
  for (int qb = 0; qb < FA32_NQB; ++qb) {
    qslot[qb] = q0 + wave * FA32_WQ + 32 * qb + lr;
    const int64_t row = qslot[qb] < L ? gidx[p0 + qslot[qb]] : -1;
    srow_[qb] = qslot[qb] < L ? sidx[p0 + qslot[qb]] : -1;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row >= 0 && !(FA32_ABL & 128)) v = *reinterpret_cast<const uint4*>(qkv + row * C3 + h * D + 16 * ks + 8 * hh);
      else if (FA32_ABL & 128) v = make_uint4(0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u + ks, 0x3c003c00u);
      qf[qb][ks] = as_bf8_(v);
    }
  }
  // first K / V tile: its row indices come straight from global memory in the same round as the query indices, its rows in
  // the same round as the Q rows -- the prologue is two dependent memory round trips, not three
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const uint64_t o16 = (uint32_t)gidx[p0 + min(st_row[i], L - 1)] * (uint32_t)(3 * C >> 3);
    stg[i] = *reinterpret_cast<const u32x4_t*>(st_src[i] + (o16 << 4));
  }
  f32x16_t o[FA32_NQB][A::NMT];
  float m2[FA32_NQB], lsum[FA32_NQB];     // shift in exp2 units (scale * log2e * score); lsum only without a spare row
#pragma unroll
  for (int qb = 0; qb < FA32_NQB; ++qb) {
    m2[qb] = -1e30f; lsum[qb] = 0.f;
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][mt][r] = 0.f;
  }
  const int ntiles = Lpad / FA32_BK;
  // per-lane read bases: K rows lr (+32 for the second block), chunk hh; V transposed reads: 16-lane group (hh, lr>>4)
  const int koff = lr * A::KROW + hh * 16;
  const int voff = (4 * hh + ((lane & 15) >> 2)) * A::VROW + ((lr >> 4) * 16 + (lane & 3) * 4) * 2;

  // S^T of one tile: two 32-key blocks per q-block; s[qb][blk][r]: key 32 blk + (r&3) + 8 (r>>2) + 4 hh, query lr
  auto qk_tile = [&](f32x16_t (&s)[FA32_NQB][2], const char* Kb) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int qb = 0; qb < FA32_NQB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qb][blk][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) {
        bf8_t a = as_bf8_(*reinterpret_cast<const uint4*>(Kb + koff + blk * 32 * A::KROW + ks * 32));
#pragma unroll
        for (int qb = 0; qb < FA32_NQB; ++qb) s[qb][blk] = MFMA32(a, qf[qb][ks], s[qb][blk]);
      }
    }
  };
  auto mask_tail = [&](f32x16_t (&s)[FA32_NQB][2], int kv0) {   // keys past the window end (last tile only; wave-uniform branch)
    if (kv0 + FA32_BK > L) {
      asm volatile("; tail tile: mask keys past the window end" ::: "memory");   // keeps this a branch (no if-conversion into selects)
#pragma unroll
      for (int qb = 0; qb < FA32_NQB; ++qb)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kv0 + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[qb][blk][r] = -INFINITY;
    }
  };
  // tile maximum (exp2 units) of the query whose 64 scores sit in lanes lr and lr + 32
  auto tile_max = [&](const f32x16_t (&s)[2]) {
    float mx = max3_(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3_(mx, s[0][r], s[0][r + 1]);
    mx = max3_(mx, s[0][15], s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3_(mx, s[1][r], s[1][r + 1]);
    mx = fmaxf(mx, s[1][15]);
    return fmaxf(mx, swap32_(mx)) * c2;
  };
  // the running shift moves only past the threshold; then everything accumulated so far is rescaled exactly once
  auto maybe_rescale = [&](int qb, float mx) {
    if (__any(mx > m2[qb] + FA32_THR)) {
      const float mn = mx > m2[qb] + FA32_THR ? mx : m2[qb];
      const float alpha = __builtin_amdgcn_exp2f(m2[qb] - mn);
      m2[qb] = mn;
      lsum[qb] *= alpha;
#pragma unroll
      for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][mt][r] *= alpha;
    }
  };
  auto max_and_rescale = [&](f32x16_t (&s)[FA32_NQB][2], bool live) {
    if (FA32_ABL & 8) return;
#pragma unroll
    for (int qb = 0; qb < FA32_NQB; ++qb) {
      const float mx = tile_max(s[qb]);
      maybe_rescale(qb, live ? mx : -1e30f);
    }
  };
  // exp + pack of one 32-key block -> the two B fragments (contraction steps) of P^T
  auto exp_pack = [&](int qb, f32x16_t& sb, bf8_t (&pf)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (FA32_ABL & 4) sb[r] = __builtin_fmaf(sb[r], c2, -m2[qb]) * 0.001f;
      else sb[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sb[r], c2, -m2[qb]));
    }
    if (!A::PADCOL) {
#pragma unroll
      for (int r = 0; r < 16; ++r) lsum[qb] += sb[r];
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      uint4 pk;
      pk.x = pack_bf16x2(sb[8 * ss + 0], sb[8 * ss + 1]); pk.y = pack_bf16x2(sb[8 * ss + 2], sb[8 * ss + 3]);
      pk.z = pack_bf16x2(sb[8 * ss + 4], sb[8 * ss + 5]); pk.w = pack_bf16x2(sb[8 * ss + 6], sb[8 * ss + 7]);
      pf[ss] = as_bf8_(pk);
    }
  };
  // O^T += V^T P^T of one key block for every q-block; element j of lane half hh in step ss <-> key 32 blk + 16 ss + 8 (j>>2)
  // + 4 hh + (j&3); the V^T fragment of a (step, tile) is read once and used by all q-blocks
  auto pv_block = [&](const bf8_t (&pf)[FA32_NQB][2], const char* Vb, int blk) {
    if (FA32_ABL & 16) {
#pragma unroll
      for (int qb = 0; qb < FA32_NQB; ++qb) asm volatile("" :: "v"(pf[qb][0]), "v"(pf[qb][1]));
      return;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const char* vb = Vb + voff + (32 * blk + 16 * ss) * A::VROW;
#pragma unroll
      for (int mt = 0; mt < A::NMT; ++mt) {
        bf8_t vf = cat_tr_(lds_tr_(vb + mt * 64), lds_tr_(vb + 8 * A::VROW + mt * 64));
#pragma unroll
        for (int qb = 0; qb < FA32_NQB; ++qb) o[qb][mt] = MFMA32(vf, pf[qb][ss], o[qb][mt]);
      }
    }
  };
  auto softmax_pv = [&](f32x16_t (&sc)[FA32_NQB][2], const char* Vb) {
    bf8_t pf[FA32_NQB][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int qb = 0; qb < FA32_NQB; ++qb) exp_pack(qb, sc[qb][blk], pf[qb]);
      pv_block(pf, Vb, blk);
    }
  };

  __syncthreads();                   // gidx_s ready
#ifdef FA32_CLOCK
  unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rk0 = __builtin_amdgcn_s_memrealtime();
#endif
#if FA32_PIPE
  // prologue: K(0), V(0) -> images 0; K(1) -> K image 1 (V(0) is simply written twice); S(0) and its maximum
  fa32_tile_write<NLD>(stg, Kimg(0), Vimg(0), st_isv, st_lds);
  fa32_tile_load<NLD>(stg, gidx_s, ntiles > 1 ? FA32_BK : 0, 0, st_row, st_isv, st_src);
  __syncthreads();
  fa32_tile_write<NLD>(stg, Kimg(1), Vimg(0), st_isv, st_lds);
  f32x16_t sa[FA32_NQB][2], sb_[FA32_NQB][2];
  qk_tile(sa, Kimg(0));
  mask_tail(sa, 0);
  max_and_rescale(sa, true);
  __syncthreads();
  auto step = [&](f32x16_t (&sc)[FA32_NQB][2], f32x16_t (&sn)[FA32_NQB][2], const int t) {
    const bool more = t + 1 < ntiles;
    if (!(FA32_ABL & 1)) fa32_tile_load<NLD>(stg, gidx_s, min(t + 2, ntiles - 1) * FA32_BK, min(t + 1, ntiles - 1) * FA32_BK, st_row, st_isv, st_src);
    // phase 1 (branch-free: in the last iteration it multiplies a stale but finite K image and the result is dropped)
    if (!(FA32_ABL & 32)) qk_tile(sn, Kimg((t + 1) & 1));
    else {
#pragma unroll
      for (int qb = 0; qb < FA32_NQB; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { sn[qb][0][r] = sc[qb][0][r] * 0.5f; sn[qb][1][r] = sc[qb][1][r] * 0.5f; }
    }
    softmax_pv(sc, Vimg(t & 1));                       // phases 2, 3
    mask_tail(sn, (t + 1) * FA32_BK);
    max_and_rescale(sn, more);
    if (!(FA32_ABL & 1)) fa32_tile_write<NLD>(stg, Kimg(t & 1), Vimg((t + 1) & 1), st_isv, st_lds);
    if (!(FA32_ABL & 2)) __syncthreads();
  };
  for (int t = 0; t < ntiles; t += 2) {                // two named score states: no runtime-indexed registers
    step(sa, sb_, t);
    if (t + 1 < ntiles) step(sb_, sa, t + 1);
  }
#else
  // plain loop: tile t+1 is loaded into registers at the top and written to the other image pair at the bottom
  fa32_tile_write<NLD>(stg, Kimg(0), Vimg(0), st_isv, st_lds);      // tile 0 was fetched beside the Q rows
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int b = t & 1;
    if (t + 1 < ntiles && !(FA32_ABL & 1)) fa32_tile_load<NLD>(stg, gidx_s, (t + 1) * FA32_BK, (t + 1) * FA32_BK, st_row, st_isv, st_src);
    f32x16_t sc[FA32_NQB][2];
    qk_tile(sc, Kimg(b));
    mask_tail(sc, t * FA32_BK);
    max_and_rescale(sc, true);
    softmax_pv(sc, Vimg(b));
    if (t + 1 < ntiles && !(FA32_ABL & 1)) fa32_tile_write<NLD>(stg, Kimg(b ^ 1), Vimg(b ^ 1), st_isv, st_lds);
    if (!(FA32_ABL & 2)) __syncthreads();
  }
#endif
#ifdef FA32_CLOCK
  if (tid == 0 && blockIdx.x < 32768) {
    fa32_stamp[4 * blockIdx.x + 0] = ck0; fa32_stamp[4 * blockIdx.x + 1] = rk0;
    fa32_stamp[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime(); fa32_stamp[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  // ---- epilogue: row sums, lse, normalised output rows
#pragma unroll
  for (int qb = 0; qb < FA32_NQB; ++qb) {
    float lt;
    if (A::PADCOL) {
      // row D of O^T = local row D % 32 of tile D / 32: register ((D%32)>>3)*4 of the hh = ((D%32)>>2)&1 half
      constexpr int LR = D % 32, REG = (LR >> 3) * 4 + (LR & 3), HF = (LR >> 2) & 1;
      float mine = o[qb][D / 32][REG], other = swap32_(mine);
      lt = (hh == HF) ? mine : other;
    } else {
      lt = lsum[qb] + swap32_(lsum[qb]);
    }
    if (qslot[qb] < L) {
      if (hh == 0) lse[(int64_t)(p0 + qslot[qb]) * H + h] = m2[qb] * 0.69314718055994530942f + __logf(lt);
      const int32_t srow = srow_[qb];
      if (srow >= 0 && !(FA32_ABL & 64)) {
        // Lane (q, hh) holds columns 8 g4 + 4 hh .. +3 of its row for every 8-column group g4: 8-byte pieces.  Pairs of
        // groups are exchanged between the two half-waves (v_permlane32_swap) so that every lane owns 16 contiguous bytes:
        // hh = 0 gets columns 16 j .. 16 j + 7, hh = 1 columns 16 j + 8 .. + 15 -- half the store instructions (the row
        // pieces are scattered by row, the tail is store-issue-bound: -0.09 ms of 0.54 at dec0 with the stores removed)
        const float inv = 1.f / lt;
        unsigned short* op = out + (int64_t)srow * C + h * D + 8 * hh;
        constexpr int NG = D / 8;                                  // 8-column groups that hold real columns
#pragma unroll
        for (int j = 0; j < (NG + 1) / 2; ++j) {
          const int ga = 2 * j, gb = 2 * j + 1;                    // group index = 4 mt + g4
          uint2 a, b;
          a.x = pack_bf16x2(o[qb][ga >> 2][4 * (ga & 3) + 0] * inv, o[qb][ga >> 2][4 * (ga & 3) + 1] * inv);
          a.y = pack_bf16x2(o[qb][ga >> 2][4 * (ga & 3) + 2] * inv, o[qb][ga >> 2][4 * (ga & 3) + 3] * inv);
          if (gb < NG) {
            b.x = pack_bf16x2(o[qb][gb >> 2][4 * (gb & 3) + 0] * inv, o[qb][gb >> 2][4 * (gb & 3) + 1] * inv);
            b.y = pack_bf16x2(o[qb][gb >> 2][4 * (gb & 3) + 2] * inv, o[qb][gb >> 2][4 * (gb & 3) + 3] * inv);
            // swap: upper half of a <-> lower half of b; afterwards (a.x, a.y, b.x, b.y) are 16 contiguous bytes in both halves
            auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
            // lanes < 32: own group ga (cols +0..3) then the upper half's group ga (cols +4..7); lanes >= 32: the lower
            // half's group gb (cols +8..11) then own group gb (cols +12..15)
            *reinterpret_cast<uint4*>(op + 16 * j) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
          } else {
            *reinterpret_cast<uint2*>(out + (int64_t)srow * C + h * D + 8 * ga + 4 * hh) = a;      // odd group count (d = 8 mod 16)
          }
        }
      }
    }
  }
}

int ss_attn_fwd_mfma32(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int W,
                       int max_window, void* out, float* lse, int C, int H, float scale, hipStream_t st) {
  const int D = C / H;
  if ((C & 7) || max_window <= 0 || max_window > FA32_IDX_CAP) return SS_ERR_ARG;
  const int qchunks = (max_window + FA32_BQ - 1) / FA32_BQ;
  dim3 g((unsigned)(W * H * qchunks)), b(FA32_THREADS);
  const unsigned short* q = (const unsigned short*)qkv; unsigned short* o = (unsigned short*)out;
  switch (D) {
    case 16: SS_LAUNCH((k_attn_fwd_mfma32<16>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    case 32: SS_LAUNCH((k_attn_fwd_mfma32<32>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    case 48: SS_LAUNCH((k_attn_fwd_mfma32<48>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    case 64: SS_LAUNCH((k_attn_fwd_mfma32<64>), g, b, 0, st, q, gidx, sidx, win_start, o, lse, C, H, scale, qchunks); break;
    default: return SS_ERR_ARG;
  }
  return SS_OK;
}
