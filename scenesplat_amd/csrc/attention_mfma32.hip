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
//   * one accumulator tile = 32 queries x 32 keys per wave; 4 waves = 128 queries per workgroup, 64 keys per barrier;
//   * K image rows padded to an odd number of 16-byte chunks (conflict-free ds_read_b128 without a swizzle), V image
//     rows at a stride of 192 B (64 B for d <= 32): the four rows of a transposed read land in distinct 64-byte bank
//     groups (ds_read_b64_tr_b16 conflict-free);
//   * staging addresses: per thread three (row, column) pairs fixed before the loop, the window's gather offsets
//     sit in LDS padded to a multiple of 64 rows: one ds_read + one 64-bit shift-add per 16-byte load.
#include "attention_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;   // native 16-byte vector: staging registers

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#ifndef FA32_THR
#define FA32_THR 6.0f          // lazy-rescale threshold in exp2 units (P <= 64)
#endif
#define FA32_WAVES 4
#define FA32_THREADS (64 * FA32_WAVES)
#define FA32_BQ (32 * FA32_WAVES)     // queries per workgroup
#define FA32_BK 64                    // keys per barrier
#define FA32_IDX_CAP SS_ATTN_MFMA_MAX_WINDOW

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

// issue the 16-byte loads of the next K + V tile into registers (written to LDS after the compute phase)
template <int NLD>
__device__ __forceinline__ void fa32_tile_load(u32x4_t (&stg)[NLD], const int32_t* gidx_s, int r0, const int (&st_row)[NLD],
                                               const char* const (&st_src)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const uint64_t o16 = (uint32_t)gidx_s[r0 + st_row[i]];
    stg[i] = *reinterpret_cast<const u32x4_t*>(st_src[i] + (o16 << 4));
  }
}
template <int NLD>
__device__ __forceinline__ void fa32_tile_write(const u32x4_t (&stg)[NLD], char* base, const int (&st_lds)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i)
    if (st_lds[i] >= 0) *reinterpret_cast<u32x4_t*>(base + st_lds[i]) = stg[i];
}

template <int D>
__global__ void __launch_bounds__(FA32_THREADS, 2)
k_attn_fwd_mfma32(const unsigned short* __restrict__ qkv, const int32_t* __restrict__ gidx,
                  const int32_t* __restrict__ sidx, const int32_t* __restrict__ win_start,
                  unsigned short* __restrict__ out, float* __restrict__ lse, int C, int H, float scale, int qchunks) {
  using A = A32<D>;
  constexpr int TOT = 2 * FA32_BK * A::CH;                                  // 16-B chunks of one K + V tile
  constexpr int NLD = (TOT + FA32_THREADS - 1) / FA32_THREADS;
  __shared__ __attribute__((aligned(16))) char smem[2 * (A::KIMG + A::VIMG)];
  __shared__ int32_t gidx_s[FA32_IDX_CAP + FA32_BK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hh = lane >> 5;
  const int lid = xcd_remap_(blockIdx.x, gridDim.x);
  const int qc = lid % qchunks; const int t_ = lid / qchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int q0 = qc * FA32_BQ;
  if (q0 >= L) return;
  const int Lpad = (L + FA32_BK - 1) / FA32_BK * FA32_BK;
  // row offsets in 16-byte units; slots past the window end repeat its last row (finite rows whose scores are masked)
  for (int i = tid; i < Lpad; i += FA32_THREADS) gidx_s[i] = (int32_t)((uint32_t)gidx[p0 + min(i, L - 1)] * (uint32_t)(3 * C >> 3));
  auto Kbuf = [&](int b_) { return smem + b_ * (A::KIMG + A::VIMG); };
  auto Vbuf = [&](int b_) { return smem + b_ * (A::KIMG + A::VIMG) + A::KIMG; };
  if (A::PADCOL) {   // V columns D .. DV-1: column D = 1.0 (row sums), the rest 0; written once, staging never touches them
    constexpr int PCH = (A::DV - D) / 8;
    for (int e = tid; e < 2 * FA32_BK * PCH; e += FA32_THREADS) {
      int b = e / (FA32_BK * PCH), r = (e / PCH) % FA32_BK, ch = e % PCH;
      *reinterpret_cast<uint4*>(Vbuf(b) + r * A::VROW + D * 2 + ch * 16) = make_uint4(ch == 0 ? 0x3F80u : 0u, 0, 0, 0);
    }
  }
  const int64_t C3 = 3 * (int64_t)C;
  const float c2 = scale * 1.44269504088896340736f;
  // ---- staging plan of this thread: chunk c = i * THREADS + tid -> (K | V, row, 16-byte column)
  int st_row[NLD], st_lds[NLD];
  const char* st_src[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int c = i * FA32_THREADS + tid;
    bool live = c < TOT;
    int second = c >= FA32_BK * A::CH;
    int cc = second ? c - FA32_BK * A::CH : c;
    int r = live ? cc / A::CH : 0, ch = live ? cc - r * A::CH : 0;
    st_row[i] = r;
    st_lds[i] = live ? (second ? A::KIMG + r * A::VROW + ch * 16 : r * A::KROW + ch * 16) : -1;
    st_src[i] = reinterpret_cast<const char*>(qkv + (second ? 2 * C : C) + h * D + ch * 8);
  }
  u32x4_t stg[NLD];
  // ---- Q fragments (B operand of S^T = K Q^T): lane (query lr, half hh) holds Q[q][16 ks + 8 hh .. +7]
  bf8_t qf[A::NKS];
  const int qslot = q0 + wave * 32 + lr;
  {
    const int64_t row = qslot < L ? gidx[p0 + qslot] : -1;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row >= 0) v = *reinterpret_cast<const uint4*>(qkv + row * C3 + h * D + 16 * ks + 8 * hh);
      qf[ks] = as_bf8_(v);
    }
  }
  f32x16_t o[A::NMT];
#pragma unroll
  for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mt][r] = 0.f;
  float m2 = -1e30f;                 // current shift, exp2 units (scale * log2e * score)
  float lsum = 0.f;                  // only when there is no spare row (D % 32 == 0)
  __syncthreads();                   // gidx_s ready
  const int ntiles = Lpad / FA32_BK;
  fa32_tile_load<NLD>(stg, gidx_s, 0, st_row, st_src);
  fa32_tile_write<NLD>(stg, Kbuf(0), st_lds);
  __syncthreads();
  // per-lane read bases: K rows lr (+32 for the second block), chunk hh; V transposed reads: 16-lane group (hh, lr>>4)
  const int koff = lr * A::KROW + hh * 16;
  const int voff = (4 * hh + ((lane & 15) >> 2)) * A::VROW + ((lr >> 4) * 16 + (lane & 3) * 4) * 2;
  for (int t = 0; t < ntiles; ++t) {
    const int b = t & 1, kv0 = t * FA32_BK;
    if (t + 1 < ntiles) fa32_tile_load<NLD>(stg, gidx_s, kv0 + FA32_BK, st_row, st_src);
    const char* Kb = Kbuf(b); const char* Vb = Vbuf(b);
    // ---- S^T = K Q^T: two 32-key blocks; s[blk][r]: key 32 blk + (r&3) + 8 (r>>2) + 4 hh, query lr
    f32x16_t s[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < A::NKS; ++ks) {
        bf8_t a = as_bf8_(*reinterpret_cast<const uint4*>(Kb + koff + blk * 32 * A::KROW + ks * 32));
        s[blk] = MFMA32(a, qf[ks], s[blk]);
      }
    }
    if (kv0 + FA32_BK > L) {           // keys past the window end (last tile only; wave-uniform branch)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[blk][r] = -INFINITY;
    }
    // ---- online softmax; the query's 64 scores of this tile sit in lanes lr and lr + 32
    float mx = max3_(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3_(mx, s[0][r], s[0][r + 1]);
    mx = max3_(mx, s[0][15], s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3_(mx, s[1][r], s[1][r + 1]);
    mx = fmaxf(mx, s[1][15]);
    mx = fmaxf(mx, swap32_(mx)) * c2;
    if (__any(mx > m2 + FA32_THR)) {
      const float mn = mx > m2 + FA32_THR ? mx : m2;
      const float alpha = __builtin_amdgcn_exp2f(m2 - mn);
      m2 = mn;
      lsum *= alpha;
#pragma unroll
      for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[blk][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[blk][r], c2, -m2));
    if (!A::PADCOL) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) lsum += s[blk][r];
    }
    // ---- O^T += V^T P^T; element j of lane half hh in contraction step ss <-> key 32 blk + 16 ss + 8 (j>>2) + 4 hh + (j&3)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        uint4 pk;
        pk.x = pack_bf16x2(s[blk][8 * ss + 0], s[blk][8 * ss + 1]); pk.y = pack_bf16x2(s[blk][8 * ss + 2], s[blk][8 * ss + 3]);
        pk.z = pack_bf16x2(s[blk][8 * ss + 4], s[blk][8 * ss + 5]); pk.w = pack_bf16x2(s[blk][8 * ss + 6], s[blk][8 * ss + 7]);
        const bf8_t pf = as_bf8_(pk);
        const char* vb = Vb + voff + (32 * blk + 16 * ss) * A::VROW;
#pragma unroll
        for (int mt = 0; mt < A::NMT; ++mt) {
          bf8_t vf = cat_tr_(lds_tr_(vb + mt * 64), lds_tr_(vb + 8 * A::VROW + mt * 64));
          o[mt] = MFMA32(vf, pf, o[mt]);
        }
      }
    if (t + 1 < ntiles) fa32_tile_write<NLD>(stg, Kbuf(b ^ 1), st_lds);
    __syncthreads();
  }
  // ---- epilogue: row sums, lse, normalised output rows
  float lt;
  if (A::PADCOL) {
    // row D of O^T = local row D % 32 of tile D / 32: register ((D%32)>>3)*4 of the hh = ((D%32)>>2)&1 half
    constexpr int LR = D % 32, REG = (LR >> 3) * 4 + (LR & 3), HF = (LR >> 2) & 1;
    float mine = o[D / 32][REG], other = swap32_(mine);
    lt = (hh == HF) ? mine : other;
  } else {
    lt = lsum + swap32_(lsum);
  }
  if (qslot < L) {
    if (hh == 0) lse[(int64_t)(p0 + qslot) * H + h] = m2 * 0.69314718055994530942f + __logf(lt);
    const int32_t srow = sidx[p0 + qslot];
    if (srow >= 0) {
      const float inv = 1.f / lt;
      unsigned short* op = out + (int64_t)srow * C + h * D + 4 * hh;
#pragma unroll
      for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * mt + 8 * g4;
          if (d0 < D) {
            uint2 v;
            v.x = pack_bf16x2(o[mt][4 * g4 + 0] * inv, o[mt][4 * g4 + 1] * inv);
            v.y = pack_bf16x2(o[mt][4 * g4 + 2] * inv, o[mt][4 * g4 + 3] * inv);
            *reinterpret_cast<uint2*>(op + d0) = v;
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
