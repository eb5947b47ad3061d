// Serialized-window attention on HEAD-MAJOR, WINDOW-ORDERED operands (gfx950), round 3.
//
// The qkv projection (gemm8.hip, ss_qkv_headmajor_fwd) writes q / k / v as  hm[3][H][NP][D]  bf16: row p of a head is padded
// slot p of the curve order (plan.WindowIndex), so the 128 keys of a K / V tile of one (window, head) are ONE contiguous
// block of 128 * 2D bytes.  Replaces flash_attn_varlen_qkvpacked_func and the qkv[order] gather in front of it
// (ptv3:184-216); rounds 1-2 fetched 2D-byte head slices of 6D-byte rows through per-row indices into registers and
// wrote them to LDS (25-35 % of the kernels, profiles/r02_attn_fwd_variants.md).
//
// Every tile is staged by LDS-DMA (global_load_lds_dwordx4, no VGPR hop, no ds_write, no index arithmetic) into PLANE images:
//     row operand   [16-byte chunk c][key][16 B]     lane (row r, half / group) reads chunk plane c at r * 16:
//                                                     the 16 lanes of every ds_read_b128 group hit 16 different rows -> 16
//                                                     different 16-byte slots of the 256-byte bank row: conflict-free, no swizzle
//     tr  operand   [16-column block cb][key][32 B]  plane stride = 32 * keys + 128: the two 16-lane groups of a transposed
//                                                     read (column blocks 2 mt, 2 mt + 1) land on opposite 128-byte halves
// A DMA piece (one wave instruction, 1 KiB) fills 64 rows of a chunk plane or 32 rows of a column-block plane; the per-lane
// SOURCE address does the re-arrangement (rows past the window end clamp to its last row).
// Pipeline: 3-slot ring, tile t+2 is issued while tile t is computed; per tile ONE s_barrier behind a counted
// s_waitcnt vmcnt(pieces of one tile): nothing but the DMA pieces is on the vector-memory counter inside the loops.
//
// Forward math: attention_mfma32.hip's (S^T = K Q^T on v_mfma_f32_32x32x16_bf16 with the query on the lane, lane-local online
// softmax with a lazily moved maximum, P^T handed to O^T += V^T P^T from the accumulators, row sums through a ones column).
#include "attention_internal.h"
#include "../../include/scenesplat_hip.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 hbf8_t;
typedef __attribute__((ext_vector_type(4))) short hs4_t;
typedef __attribute__((ext_vector_type(8))) short hs8_t;
typedef __attribute__((address_space(3))) hs4_t hlds_s4_t;

#define HM_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define HM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

#ifndef HM_THR
#define HM_THR 6.0f            // lazy-rescale threshold in exp2 units (P <= 64)
#endif
#define HM_WAVES 8
#define HM_THREADS (64 * HM_WAVES)
#define HM_BQ (32 * HM_WAVES)    // queries per forward workgroup
#define HM_BK 128                // keys per ring slot / barrier
#ifndef HM_FWD_NQ_DEFAULT
#define HM_FWD_NQ_DEFAULT 1     // query sub-tiles per wave of the forward.  2 = the 4-wave x 64-query geometry (every K / V fragment read feeds two
                                // MFMAs, 222 VGPRs, 2 waves / SIMD): measured 0.394 ms against 0.379 ms for 1 at the dec0 shape (round 4) -- LDS
                                // reads are not what bounds the kernel; kept as a switch (SS_ATTN_FWD_NQ=2)
#endif
#ifndef HM_ABL
#define HM_ABL 0                 // ablation mask of scripts/ubench/attn_hm_bench.hip (diagnostic builds only; 0 in the library)
#endif

template <int D, int NW = HM_WAVES> struct HMC {
  static constexpr int CH = D / 8;                         // 16-byte chunks per row
  static constexpr int NKS = D / 16;                       // QK^T contraction steps
  static constexpr int NMT = (D + 31) / 32;                // 32-row tiles of O^T
  static constexpr int DV = NMT * 32;
  static constexpr bool PADCOL = DV > D;                   // spare row D of O^T carries the row sums
  static constexpr int NCB = D / 16;                       // real 16-column blocks of V
  static constexpr int KPL = HM_BK * 16;                   // chunk plane
  static constexpr int VPL = HM_BK * 32 + 128;             // column-block plane (+128: bank stagger)
  static constexpr int KIMG = CH * KPL;
  static constexpr int BUF = (KIMG + NCB * VPL + 255) & ~255;
  static constexpr int PADPL = PADCOL ? VPL : 0;           // shared ones plane, based at 128 mod 256
  static constexpr int LDS = 3 * BUF + (PADCOL ? 128 + PADPL : 0);
  static constexpr int NPW = (D / 2) / NW;                 // DMA pieces per wave and tile (D / 2 pieces of 1 KiB per 128-key tile)
  static constexpr int NKP = CH * (HM_BK / 64);            // K pieces of a tile
};

__device__ __forceinline__ hbf8_t hm_bf8(uint4 v) { return __builtin_bit_cast(hbf8_t, v); }
__device__ __forceinline__ hs4_t hm_tr(const char* addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(addr)); }
__device__ __forceinline__ hbf8_t hm_cat(hs4_t lo, hs4_t hi) {
  hs8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(hbf8_t, v);
}
__device__ __forceinline__ float hm_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float hm_swap32(float v) {     // value of lane ^ 32
  unsigned int u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ int hm_xcd(int bid, int nb) {
  int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}
// LDS-DMA as inline asm (wgrad8.hip: with the builtin hipcc drains vmcnt in front of every transposed read)
__device__ __forceinline__ void hm_glds16(const void* gsrc, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_wave_base) : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset
__device__ __forceinline__ void hm_glds16s(const void* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ const char* hm_uniform_ptr(const void* p) {
  uint64_t u = (uint64_t)(uintptr_t)p;
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ float hm_bf16_round(float x) { return __uint_as_float(pack_bf16x2(x, 0.f) << 16); }
__device__ __forceinline__ unsigned int hm_hi_lo(float x) {     // x ~= hi + lo, both bf16: packed {hi, lo}
  float hi = hm_bf16_round(x);
  return pack_bf16x2(hi, x - hi);
}

// ---- DMA plan of one wave: NPW pieces per (row operand with CH chunk planes | tr operand with NCB column-block planes) tile ----
// piece pi = wave * NPW + j; pi < NKP: chunk plane pi / 2, key half pi % 2 (lane l <-> key 64 half + l); else column-block plane
// (pi - NKP) / 4, key quarter (pi - NKP) % 4 (lane l <-> key 32 quarter + l / 2, 16-byte half l & 1).  Everything here is
// wave-uniform (SGPRs): isv 0 = row operand, 1 = tr operand; rbase first key; cbase byte column; lds offset inside a ring slot.
template <int D, int NW = HM_WAVES> struct HMPlan { int isv[HMC<D, NW>::NPW], rbase[HMC<D, NW>::NPW], cbase[HMC<D, NW>::NPW]; unsigned lds[HMC<D, NW>::NPW]; };
template <int D, int NW>
__device__ __forceinline__ void hm_plan(int wave_u, HMPlan<D, NW>& P) {
  using A = HMC<D, NW>;
#pragma unroll
  for (int j = 0; j < A::NPW; ++j) {
    const int pi = wave_u * A::NPW + j;
    const bool isv = pi >= A::NKP;
    const int vi = pi - A::NKP;
    P.isv[j] = isv;
    P.rbase[j] = isv ? 32 * (vi & 3) : 64 * (pi & 1);
    P.cbase[j] = isv ? (vi >> 2) * 32 : (pi >> 1) * 16;
    P.lds[j] = isv ? A::KIMG + (vi >> 2) * A::VPL + (vi & 3) * 1024 : (pi >> 1) * A::KPL + (pi & 1) * 1024;
  }
}

// =====================================================================================
// forward.  NW waves per workgroup, NQ 32-query sub-tiles per wave (round 4: <4, 2> = every K / V fragment read from LDS feeds
// TWO MFMAs and the exponentials of one sub-tile sit between the MFMAs of the other inside ONE instruction stream; <8, 1> = the
// round-3 geometry).  Both cover 256 queries per workgroup and stage the same 128-key ring slots.
// =====================================================================================
template <int D, int NW, int NQ>
__global__ void __launch_bounds__(64 * NW, (NQ == 1 ? (D <= 48 ? 4 : 2) : 2))
k_attn_hm_fwd(const unsigned short* __restrict__ hm, int64_t NP, const int32_t* __restrict__ sidx,
              const int32_t* __restrict__ win_start, unsigned short* __restrict__ out, float* __restrict__ nlse2, int C, int H,
              float scale, int qchunks) {
  using A = HMC<D, NW>;
  constexpr int THREADS = 64 * NW, BQ = 32 * NW * NQ;
  __shared__ __attribute__((aligned(256))) char smem[A::LDS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hh = lane >> 5;
  const int lid = hm_xcd(blockIdx.x, gridDim.x);
  const int qc = lid % qchunks; const int t_ = lid / qchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int q0 = qc * BQ;
  if (q0 >= L) return;
  const int64_t sec = (int64_t)H * NP * D;                          // elements per q / k / v section
  const unsigned short* const qbase = hm + ((int64_t)h * NP + p0) * D;
  const unsigned short* const kbase = qbase + sec;
  const unsigned short* const vbase = kbase + sec;
  (void)scale;        // folded into section 0 of hm by the projection (q * scale * log2 e)
  const int ntiles = (L + HM_BK - 1) / HM_BK;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
  char* const padpl = smem + 3 * A::BUF + 128;

  // ---- DMA plan ----
  HMPlan<D, NW> P;
  hm_plan<D, NW>(__builtin_amdgcn_readfirstlane(wave), P);
  const char* const kb_u = hm_uniform_ptr(kbase);
  const char* const vb_u = hm_uniform_ptr(vbase);
  auto issue_tile = [&](int t, unsigned slot_bytes) {
    const int key0 = min(t, ntiles - 1) * HM_BK;                   // past the end: re-stage the last tile (keeps vmcnt uniform)
#pragma unroll
    for (int j = 0; j < A::NPW; ++j) {
      const int row = min(key0 + P.rbase[j] + (P.isv[j] ? (lane >> 1) : lane), L - 1);
      const unsigned off = __umul24((unsigned)row, 2u * D) + P.cbase[j] + (P.isv[j] ? (lane & 1) * 16 : 0);
      if (!(HM_ABL & 1)) hm_glds16s(P.isv[j] ? vb_u : kb_u, off, lds0 + slot_bytes + P.lds[j]);
    }
  };
  issue_tile(0, 0u);
  issue_tile(1, (unsigned)A::BUF);

  // ---- ones plane (column D = 1.0: row D of O^T = sum_k P); written once, the DMA never touches it ----
  if (A::PADCOL) {
    for (int e = tid; e < HM_BK * 2; e += THREADS)
      *reinterpret_cast<uint4*>(padpl + e * 16) = make_uint4((e & 1) ? 0u : 0x3F80u, 0, 0, 0);
  }
  // ---- Q fragments (B operand of S^T = K Q^T): lane (query lr, half hh) of sub-tile u holds Q[q][16 ks + 8 hh .. +7] ----
  int qslot[NQ];
  int32_t srow[NQ];
  hbf8_t qf[NQ][A::NKS];
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    qslot[u] = q0 + (wave * NQ + u) * 32 + lr;
    const unsigned short* qp = qbase + (int64_t)min(qslot[u], L - 1) * D + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) qf[u][ks] = hm_bf8(*reinterpret_cast<const uint4*>(qp + 16 * ks));
    srow[u] = qslot[u] < L ? sidx[p0 + qslot[u]] : -1;
  }

  f32x16_t o[NQ][A::NMT];
  // running shift m2 (exp2 units; section 0 of hm holds q * scale * log2(e), written by the projection's epilogue in fp32
  // before the bf16 rounding) and MNEG = sixteen registers of -m2: the C operand of the first QK^T MFMA of every block, so
  // the scores leave the matrix pipe already shifted and the softmax is exp2 + pack alone (no fma per score)
  float m2[NQ], lsum[NQ];
  f32x16_t mneg[NQ];
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    m2[u] = 0.f; lsum[u] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) mneg[u][r] = 0.f;
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][mt][r] = 0.f;
    asm volatile("" : "+v"(mneg[u]));
  }

  // per-lane read bases (inside a ring slot): K chunk plane hh (+ 2 ks), row lr; V: 16-lane group (hh, lr >> 4) reads rows
  // 4 hh + q (+8), columns 4 p .. of column block 2 mt + (lr >> 4)
  const unsigned lds_lane0 = (unsigned)(uintptr_t)smem;
  const unsigned koff = lds_lane0 + hh * A::KPL + lr * 16;
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg = lr >> 4;
  unsigned voff[A::NMT];
  bool vpad[A::NMT];
#pragma unroll
  for (int mt = 0; mt < A::NMT; ++mt) {
    const int cb = 2 * mt + vg;
    vpad[mt] = cb >= A::NCB;
    voff[mt] = lds_lane0 + (vpad[mt] ? 3 * A::BUF + 128 : A::KIMG + cb * A::VPL) + (4 * hh + vq) * 32 + vp * 8;
  }
  typedef __attribute__((ext_vector_type(4))) unsigned int hu32x4_t;
  typedef const __attribute__((address_space(3))) hu32x4_t lds_u4_t;

  // ---- one 32-key block = rows IMM .. IMM + 31 of the ring slot whose per-lane bases are kb / vb[], in three pieces so that
  // the tile loop can issue QK^T of block b+1 BEFORE the exponentials of block b (matrix and vector work of one wave overlap);
  // a K / V fragment is read ONCE and multiplied with the operands of all NQ sub-tiles
  auto qk_block = [&](f32x16_t (&s)[NQ], const unsigned kb, const int IMM) {
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      hbf8_t a = __builtin_bit_cast(hbf8_t, *(lds_u4_t*)(kb + ks * 2 * A::KPL + IMM * 16));
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        if (!(HM_ABL & 32)) s[u] = (ks == 0) ? HM_MFMA32(a, qf[u][ks], mneg[u]) : HM_MFMA32(a, qf[u][ks], s[u]);
        else if (ks == 0) s[u] = mneg[u];
      }
    }
  };
  // tail mask, block maximum, (rare) move of the running shift
  auto max_block = [&](f32x16_t (&sv)[NQ], const int kv0, const bool first) {
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      f32x16_t& s = sv[u];
      if (kv0 + 32 > L) {               // keys past the window end (last tile only; wave-uniform branch)
        asm volatile("; tail: mask keys past the window end" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -INFINITY;
      }
      if (HM_ABL & 8) continue;
      float mx = hm_max3(s[0], s[1], s[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = hm_max3(mx, s[r], s[r + 1]);
      mx = fmaxf(mx, s[15]);
      {   // maximum over the two half-waves: after the swap one of (a, b) is this lane's value, the other its partner's
        unsigned int uu = __float_as_uint(mx);
        auto r = __builtin_amdgcn_permlane32_swap(uu, uu, false, false);
        mx = hm_max3(mx, __uint_as_float(r[0]), __uint_as_float(r[1]));
      }
      const float thr = first ? -INFINITY : HM_THR;
      if (__any(mx > thr)) {        // the shift moves only past the threshold; everything so far is rescaled exactly once
        const float dlt = mx > thr ? mx : 0.f;
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-dlt);
        m2[u] += dlt;
        lsum[u] *= alpha;
#pragma unroll
        for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[u][mt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] -= dlt; mneg[u][r] = -m2[u]; }
        asm volatile("" : "+v"(mneg[u]));
      }
    }
  };
  auto pv_block = [&](f32x16_t (&sv)[NQ], const unsigned (&vb)[A::NMT], const int IMM) {
    hbf8_t pf[NQ][2];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      f32x16_t& s = sv[u];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (HM_ABL & 4) s[r] = s[r] * 0.001f;
        else s[r] = __builtin_amdgcn_exp2f(s[r]);
      }
      if (!A::PADCOL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) lsum[u] += s[r];
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        uint4 pk;
        pk.x = pack_bf16x2(s[8 * ss + 0], s[8 * ss + 1]); pk.y = pack_bf16x2(s[8 * ss + 2], s[8 * ss + 3]);
        pk.z = pack_bf16x2(s[8 * ss + 4], s[8 * ss + 5]); pk.w = pack_bf16x2(s[8 * ss + 6], s[8 * ss + 7]);
        pf[u][ss] = hm_bf8(pk);
      }
    }
    // O^T += V^T P^T; element j of lane half hh in step ss <-> key 16 ss + 8 (j>>2) + 4 hh + (j&3) of the block
    if (HM_ABL & 16) {
#pragma unroll
      for (int u = 0; u < NQ; ++u) asm volatile("" :: "v"(pf[u][0]), "v"(pf[u][1]));
      return;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
      for (int mt = 0; mt < A::NMT; ++mt) {
        const unsigned va = vb[mt] + (IMM + 16 * ss) * 32;
        hbf8_t vf = hm_cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(va)),
                           __builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(va + 8 * 32)));
#pragma unroll
        for (int u = 0; u < NQ; ++u) o[u][mt] = HM_MFMA32(vf, pf[u][ss], o[u][mt]);
      }
    }
  };

  // every compiler-visible load retires HERE: inside the loop the vector-memory counter belongs to the DMA pieces alone (a
  // pending ordinary load would make hipcc drain vmcnt(0) at its first use inside the loop, every iteration)
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    asm volatile("" : "+v"(srow[u]));
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) asm volatile("" : "+v"(qf[u][ks]));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ones plane is written before the first barrier releases its readers
  unsigned slot_off = 0;                                   // byte offset of tile t's ring slot
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed (own pieces: all but the NPW youngest), then everybody's; the barrier also says that every wave is
    // done with tile t-1, whose slot tile t+2 goes to
    if (A::NPW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (A::NPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (A::NPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (A::NPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (A::NPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (!(HM_ABL & 2)) { __builtin_amdgcn_s_barrier(); }
    __builtin_amdgcn_sched_barrier(0);
    issue_tile(t + 2, slot_off == 0 ? 2u * A::BUF : slot_off - A::BUF);
    unsigned kb = koff + slot_off, vb[A::NMT];
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt) vb[mt] = voff[mt] + (vpad[mt] ? 0u : slot_off);
    asm volatile("" : "+v"(kb));
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt) asm volatile("" : "+v"(vb[mt]));
    const int kv0 = t * HM_BK;
    // blocks past the window end (last tile of a short window) run on the clamped rows with every score masked: P = 0
    f32x16_t sa[NQ], sb[NQ];
    qk_block(sa, kb, 0);
    max_block(sa, kv0, t == 0);
    qk_block(sb, kb, 32);
    pv_block(sa, vb, 0);
    max_block(sb, kv0 + 32, false);
    qk_block(sa, kb, 64);
    pv_block(sb, vb, 32);
    max_block(sa, kv0 + 64, false);
    qk_block(sb, kb, 96);
    pv_block(sa, vb, 64);
    max_block(sb, kv0 + 96, false);
    pv_block(sb, vb, 96);
    slot_off = slot_off == 2 * A::BUF ? 0u : slot_off + A::BUF;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-staged pieces of the last two issues (LDS must outlive them)

  // ---- epilogue: row sums, lse, normalised output rows ----
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    float lt;
    if (A::PADCOL) {
      constexpr int LR = D % 32, REG = (LR >> 3) * 4 + (LR & 3), HF = (LR >> 2) & 1;
      float mine = o[u][D / 32][REG], other = hm_swap32(mine);
      lt = (hh == HF) ? mine : other;
    } else {
      lt = lsum[u] + hm_swap32(lsum[u]);
    }
    if (qslot[u] < L) {
      // -lse in exp2 units: the backward kernels start their score accumulators from it (C operand of the first MFMA).  A BORROWED
      // slot (duplicate padding, sidx < 0) publishes -inf: its recomputed probabilities are exactly 0 in the backward, so the
      // dK / dV kernel may stream any finite dO row for it (round 4: dO is read in place, no head-major copy)
      if (hh == 0) nlse2[(int64_t)h * NP + p0 + qslot[u]] = srow[u] >= 0 ? -(m2[u] + __log2f(lt)) : -INFINITY;
      if (srow[u] >= 0 && !(HM_ABL & 64)) {
        const float inv = 1.f / lt;
        unsigned short* op = out + (int64_t)srow[u] * C + h * D + 8 * hh;
        constexpr int NG = D / 8;
#pragma unroll
        for (int j = 0; j < (NG + 1) / 2; ++j) {
          const int ga = 2 * j, gb = 2 * j + 1;
          uint2 a, b;
          a.x = pack_bf16x2(o[u][ga >> 2][4 * (ga & 3) + 0] * inv, o[u][ga >> 2][4 * (ga & 3) + 1] * inv);
          a.y = pack_bf16x2(o[u][ga >> 2][4 * (ga & 3) + 2] * inv, o[u][ga >> 2][4 * (ga & 3) + 3] * inv);
          if (gb < NG) {
            b.x = pack_bf16x2(o[u][gb >> 2][4 * (gb & 3) + 0] * inv, o[u][gb >> 2][4 * (gb & 3) + 1] * inv);
            b.y = pack_bf16x2(o[u][gb >> 2][4 * (gb & 3) + 2] * inv, o[u][gb >> 2][4 * (gb & 3) + 3] * inv);
            auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
            *reinterpret_cast<uint4*>(op + 16 * j) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
          } else {
            *reinterpret_cast<uint2*>(out + (int64_t)srow[u] * C + h * D + 8 * ga + 4 * hh) = a;
          }
        }
      }
    }
  }
}

// geometry of the forward: 8 waves x 32 queries (default) or, SS_ATTN_FWD_NQ=2, 4 waves x 64 queries
static int hm_fwd_nq() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SS_ATTN_FWD_NQ"); v = e ? (atoi(e) == 2 ? 2 : 1) : HM_FWD_NQ_DEFAULT; }
  return v;
}

int ss_attn_hm_fwd(const void* hm, int64_t np, const int32_t* sidx, const int32_t* win_start, int W, int max_window,
                   void* out, float* nlse2, int C, int H, float scale, hipStream_t st) {
  const int D = C / H;
  if ((C & 7) || max_window <= 0 || W <= 0) return SS_ERR_ARG;
  const int qchunks = (max_window + HM_BQ - 1) / HM_BQ;
  const unsigned short* q = (const unsigned short*)hm; unsigned short* o = (unsigned short*)out;
  dim3 g((unsigned)(W * H * qchunks));
  if (hm_fwd_nq() == 2) {
    dim3 b(256);
    switch (D) {
      case 16: SS_LAUNCH((k_attn_hm_fwd<16, 4, 2>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
      case 32: SS_LAUNCH((k_attn_hm_fwd<32, 4, 2>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
      case 48: SS_LAUNCH((k_attn_hm_fwd<48, 4, 2>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
      case 64: SS_LAUNCH((k_attn_hm_fwd<64, 4, 2>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
      default: return SS_ERR_ARG;
    }
    return SS_OK;
  }
  dim3 b(HM_THREADS);
  switch (D) {
    case 16: SS_LAUNCH((k_attn_hm_fwd<16, 8, 1>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
    case 32: SS_LAUNCH((k_attn_hm_fwd<32, 8, 1>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
    case 48: SS_LAUNCH((k_attn_hm_fwd<48, 8, 1>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
    case 64: SS_LAUNCH((k_attn_hm_fwd<64, 8, 1>), g, b, 0, st, q, np, sidx, win_start, o, nlse2, C, H, scale, qchunks); break;
    default: return SS_ERR_ARG;
  }
  return SS_OK;
}

// =====================================================================================
// backward.  Two kernels, no atomics, bitwise reproducible.  Both on v_mfma_f32_32x32x16_bf16 (rounds 1-2: 16x16x32, which
// costs 8 issue cycles per 16 pipe cycles and made the kernels vector-ISSUE-bound); the per-row constants of the recomputed
// probabilities enter through the C operand of the first MFMA of a chain, so a score costs exp2 + one multiply:
//     S' = -lse2 + q~ k      (exp2 units)        P = exp2(S')        dP' = -delta + dO v        dS = P o dP'
// DUAL image of a row tile (K, V in dQ; q~, dO in dK/dV): chunk planes [c][row][16 B] at a stride of 16 * rows + 64 bytes,
// i.e. plane c starts at (c & 3) * 64 modulo the 256-byte bank row.  Row reads (ds_read_b128) stay conflict-free (a rotation of the bank row) and
// the SAME image serves the transposed reads: the 4 x 16 block of a 16-lane group lies in planes 2 cb, 2 cb + 1 (64 B
// apart) and the two groups of a half-wave in planes 128 B apart -- 256 distinct bank bytes.
// =====================================================================================
template <int D, int BR> struct HMD {                      // BR rows per ring slot image
  static constexpr int CH = D / 8, NKS = D / 16, NMT = (D + 31) / 32, NCB = D / 16;
  static constexpr int PL = BR * 16 + 64;                  // chunk plane stride (BR * 16 is a multiple of 256)
  static constexpr int IMG = (CH * PL + 255) & ~255;
  static __device__ __forceinline__ constexpr int plane(int c) { return c * PL; }
};

#define HMQ_WAVES 4
#define HMQ_THREADS (64 * HMQ_WAVES)
#define HMQ_BQ (32 * HMQ_WAVES)     // queries per dQ workgroup
#define HMQ_BK 64                   // keys per ring slot

// dQ: query-stationary (query on the lane).  S'^T = -lse2 + K q~^T, dP'^T = -delta + V dO^T, dS^T = P^T o dP'^T,
// dQ^T += K^T dS^T (K^T by transposed reads of the same K image; dS^T straight from the accumulators).
// Also publishes what the dK/dV kernel streams: -delta (H, NP) and dO head-major (H, NP, D), both read here anyway.
template <int D>
__global__ void __launch_bounds__(HMQ_THREADS, (D <= 48 ? 3 : 2))
k_attn_hm_dq(const unsigned short* __restrict__ hm, int64_t NP, const unsigned short* __restrict__ dout,
             const unsigned short* __restrict__ outp, const float* __restrict__ nlse2, float* __restrict__ ndelta,
             unsigned short* __restrict__ doh, const int32_t* __restrict__ sidx, const int32_t* __restrict__ win_start,
             unsigned short* __restrict__ dqkv, int C, int H, float scale, int qchunks) {
  using A = HMD<D, HMQ_BK>;
  constexpr int NPW = D / 16;                               // DMA pieces per wave and tile: 2 CH planes of 1 KiB over 4 waves
  constexpr int BUF = 2 * A::IMG;
  __shared__ __attribute__((aligned(256))) char smem[3 * BUF];
  typedef __attribute__((ext_vector_type(4))) unsigned int hu32x4_t;
  typedef const __attribute__((address_space(3))) hu32x4_t lds_u4_t;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hh = lane >> 5;
  const int lid = hm_xcd(blockIdx.x, gridDim.x);
  const int qc = lid % qchunks; const int t_ = lid / qchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int q0 = qc * HMQ_BQ;
  if (q0 >= L) return;
  const int64_t sec = (int64_t)H * NP * D;
  const int64_t hrow = (int64_t)h * NP + p0;
  const unsigned short* const qbase = hm + hrow * D;
  const char* const kb_u = hm_uniform_ptr(qbase + sec);
  const char* const vb_u = hm_uniform_ptr(qbase + 2 * sec);
  const int ntiles = (L + HMQ_BK - 1) / HMQ_BK;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_tile = [&](int t, unsigned slot_bytes) {
    const int key0 = min(t, ntiles - 1) * HMQ_BK;
    const int row = min(key0 + lane, L - 1);
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pi = wave_u * NPW + j;
      const int isv = pi >= A::CH, c = isv ? pi - A::CH : pi;
      const unsigned off = __umul24((unsigned)row, 2u * D) + c * 16;
      hm_glds16s(isv ? vb_u : kb_u, off, lds0 + slot_bytes + isv * A::IMG + A::plane(c));
    }
  };
  issue_tile(0, 0u);
  issue_tile(1, (unsigned)BUF);

  // ---- this lane's query: q~ and dO fragments (B operands), -lse2, delta = rowsum(O o dO) ----
  const int qslot = q0 + wave * 32 + lr;
  const bool qok = qslot < L;
  int32_t srow = qok ? sidx[p0 + qslot] : -1;
  float nl = qok ? nlse2[hrow + qslot] : 0.f;
  hbf8_t qf[A::NKS], gf[A::NKS];
  float dsum = 0.f;
  {
    const unsigned short* qp = qbase + (int64_t)min(qslot, L - 1) * D + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      qf[ks] = hm_bf8(*reinterpret_cast<const uint4*>(qp + 16 * ks));
      uint4 g = make_uint4(0, 0, 0, 0), o = make_uint4(0, 0, 0, 0);
      if (srow >= 0) {
        g = *reinterpret_cast<const uint4*>(dout + (int64_t)srow * C + h * D + 16 * ks + 8 * hh);
        o = *reinterpret_cast<const uint4*>(outp + (int64_t)srow * C + h * D + 16 * ks + 8 * hh);
      }
      gf[ks] = hm_bf8(g);
      const unsigned int* ug = reinterpret_cast<const unsigned int*>(&g);
      const unsigned int* uo = reinterpret_cast<const unsigned int*>(&o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dsum += __uint_as_float(ug[j] << 16) * __uint_as_float(uo[j] << 16);
        dsum += __uint_as_float(ug[j] & 0xffff0000u) * __uint_as_float(uo[j] & 0xffff0000u);
      }
      if (qok && doh) *reinterpret_cast<uint4*>(doh + (hrow + qslot) * D + 16 * ks + 8 * hh) = g;    // head-major copy for dK/dV (round-3 path)
    }
  }
  float ndl = -(dsum + hm_swap32(dsum));
  if (qok && hh == 0) ndelta[hrow + qslot] = ndl;
  f32x16_t cl, cd;                       // C operands: sixteen registers of -lse2 / -delta of this lane's query
#pragma unroll
  for (int r = 0; r < 16; ++r) { cl[r] = nl; cd[r] = ndl; }
  asm volatile("" : "+v"(cl));
  asm volatile("" : "+v"(cd));
  asm volatile("" : "+v"(srow));
#pragma unroll
  for (int ks = 0; ks < A::NKS; ++ks) { asm volatile("" : "+v"(qf[ks])); asm volatile("" : "+v"(gf[ks])); }

  f32x16_t dq[A::NMT];
#pragma unroll
  for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[mt][r] = 0.f;

  // per-lane read bases: row reads = chunk plane 2 ks + hh, row lr; transposed reads = 16-lane group (hh, vg): rows 4 hh + q
  // (+8), columns 4 p .. of column block 2 mt + vg (a block past the last real one aliases the block before it: same
  // addresses, rows >= D of dQ^T are never stored)
  const unsigned lds_lane0 = (unsigned)(uintptr_t)smem;
  const unsigned roff = lds_lane0 + hh * A::PL + lr * 16;
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg = lr >> 4;
  unsigned toff[A::NMT];
#pragma unroll
  for (int mt = 0; mt < A::NMT; ++mt) {
    const int cb = min(2 * mt + vg, A::NCB - 1);
    const int c = 2 * cb + (vp >> 1);
    toff[mt] = lds_lane0 + c * A::PL + (4 * hh + vq) * 16 + (vp & 1) * 8;
  }
  auto block = [&](const unsigned rb, const unsigned (&tb)[A::NMT], const int kv0, const int IMM) {
    f32x16_t s, dp;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      const int po = 2 * ks * A::PL + IMM * 16;
      hbf8_t ka = __builtin_bit_cast(hbf8_t, *(lds_u4_t*)(rb + po));
      hbf8_t va = __builtin_bit_cast(hbf8_t, *(lds_u4_t*)(rb + A::IMG + po));
      s = (ks == 0) ? HM_MFMA32(ka, qf[ks], cl) : HM_MFMA32(ka, qf[ks], s);
      dp = (ks == 0) ? HM_MFMA32(va, gf[ks], cd) : HM_MFMA32(va, gf[ks], dp);
    }
    if (kv0 + 32 > L) {               // keys past the window end: P = 0 (last tile only; wave-uniform branch)
      asm volatile("; tail: mask keys past the window end" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]) * dp[r];
    hbf8_t df[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      uint4 pk;
      pk.x = pack_bf16x2(s[8 * ss + 0], s[8 * ss + 1]); pk.y = pack_bf16x2(s[8 * ss + 2], s[8 * ss + 3]);
      pk.z = pack_bf16x2(s[8 * ss + 4], s[8 * ss + 5]); pk.w = pack_bf16x2(s[8 * ss + 6], s[8 * ss + 7]);
      df[ss] = hm_bf8(pk);
    }
    // dQ^T += K^T dS^T; element j of lane half hh in step ss <-> key 16 ss + 8 (j>>2) + 4 hh + (j&3) of the block
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int mt = 0; mt < A::NMT; ++mt) {
        const unsigned ta = tb[mt] + (IMM + 16 * ss) * 16;
        hbf8_t kt = hm_cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(ta)),
                           __builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(ta + 8 * 16)));
        dq[mt] = HM_MFMA32(kt, df[ss], dq[mt]);
      }
  };

  unsigned slot_off = 0;
  for (int t = 0; t < ntiles; ++t) {
    if (NPW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (NPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (NPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue_tile(t + 2, slot_off == 0 ? 2u * BUF : slot_off - BUF);
    unsigned rb = roff + slot_off, tb[A::NMT];
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt) tb[mt] = toff[mt] + slot_off;
    asm volatile("" : "+v"(rb));
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt) asm volatile("" : "+v"(tb[mt]));
    const int kv0 = t * HMQ_BK;
    block(rb, tb, kv0, 0);
    block(rb, tb, kv0 + 32, 32);
    slot_off = slot_off == 2 * BUF ? 0u : slot_off + BUF;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: dq = scale * dQ^T of this lane's query, 16-byte pieces after a half-wave exchange (as the forward) ----
  if (srow >= 0) {
    unsigned short* op = dqkv + (int64_t)srow * 3 * C + h * D + 8 * hh;
    constexpr int NG = D / 8;
#pragma unroll
    for (int j = 0; j < (NG + 1) / 2; ++j) {
      const int ga = 2 * j, gb = 2 * j + 1;
      uint2 a, b;
      a.x = pack_bf16x2(dq[ga >> 2][4 * (ga & 3) + 0] * scale, dq[ga >> 2][4 * (ga & 3) + 1] * scale);
      a.y = pack_bf16x2(dq[ga >> 2][4 * (ga & 3) + 2] * scale, dq[ga >> 2][4 * (ga & 3) + 3] * scale);
      if (gb < NG) {
        b.x = pack_bf16x2(dq[gb >> 2][4 * (gb & 3) + 0] * scale, dq[gb >> 2][4 * (gb & 3) + 1] * scale);
        b.y = pack_bf16x2(dq[gb >> 2][4 * (gb & 3) + 2] * scale, dq[gb >> 2][4 * (gb & 3) + 3] * scale);
        auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        *reinterpret_cast<uint4*>(op + 16 * j) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
      } else {
        *reinterpret_cast<uint2*>(dqkv + (int64_t)srow * 3 * C + h * D + 8 * ga + 4 * hh) = a;
      }
    }
  }
}

int ss_attn_hm_dq(const void* hm, int64_t np, const void* dout, const void* out, const float* nlse2, float* ndelta, void* doh,
                  const int32_t* sidx, const int32_t* win_start, int W, int max_window, void* dqkv, int C, int H, float scale,
                  hipStream_t st) {
  const int D = C / H;
  if ((C & 7) || max_window <= 0 || W <= 0) return SS_ERR_ARG;
  const int qchunks = (max_window + HMQ_BQ - 1) / HMQ_BQ;
  dim3 g((unsigned)(W * H * qchunks)), b(HMQ_THREADS);
  const unsigned short* q = (const unsigned short*)hm; const unsigned short* go = (const unsigned short*)dout;
  const unsigned short* oo = (const unsigned short*)out;
#define SS_HQ_CASE(DD) case DD: SS_LAUNCH((k_attn_hm_dq<DD>), g, b, 0, st, q, np, go, oo, nlse2, ndelta, (unsigned short*)doh, sidx, win_start, (unsigned short*)dqkv, C, H, scale, qchunks); break;
  switch (D) {
    SS_HQ_CASE(16) SS_HQ_CASE(32) SS_HQ_CASE(48) SS_HQ_CASE(64)
    default: return SS_ERR_ARG;
  }
#undef SS_HQ_CASE
  return SS_OK;
}

// dK/dV: key-stationary (key on the lane, K / V fragments in registers for the whole sweep); 64-query tiles of q~ and dO stream
// through the ring as DUAL images, -lse2 / -delta of the tile as two fp32 vectors (one global_load_lds_dword each) that are read
// back with broadcast ds_read_b128 straight INTO the score accumulators (C operand).
//     S' = -lse2 + q~ K^T   dP' = -delta + dO V^T   (rows = queries in the registers, column = key on the lane)
//     dV^T += dO^T P        dK^T += q~^T dS          (P, dS straight from the accumulators: they sum over the row index)
#define HMK_WAVES 4
#define HMK_THREADS (64 * HMK_WAVES)
#define HMK_BKEYS (32 * HMK_WAVES)   // keys per workgroup
#define HMK_BQ 64                    // queries per ring slot

__device__ __forceinline__ void hm_glds4s(const void* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave_base) : "memory");
}

#define HMK_MAXW 1024                // longest window whose row map fits the in-place path's LDS table

// INPLACE (round 4): dO is streamed straight from the (n, C) gradient tensor -- the DMA's per-lane source offset is
// sidx[slot] * 2 C + h * 2 D + chunk, the window's row map sits in LDS -- instead of from a head-major copy the dQ kernel had to
// write (157 MB written + 157 MB read per dec0 launch).  Borrowed slots stream row 0: their -lse2 is -inf (forward), so P = 0.
template <int D, bool INPLACE>
__global__ void __launch_bounds__(HMK_THREADS, 2)
k_attn_hm_dkv(const unsigned short* __restrict__ hm, int64_t NP, const unsigned short* __restrict__ doh,
              const float* __restrict__ nlse2, const float* __restrict__ ndelta, const int32_t* __restrict__ sidx,
              const int32_t* __restrict__ win_start, unsigned short* __restrict__ dqkv, unsigned short* __restrict__ extra,
              int C, int H, int kchunks) {
  using A = HMD<D, HMK_BQ>;
  constexpr int NPW = D / 16;                               // 16-byte DMA pieces per wave and tile (+ 1 dword piece)
  constexpr int OFF_L = 2 * A::IMG, OFF_D = OFF_L + 256, OFF_X = OFF_D + 256;   // -lse2[64], -delta[64], dummy target
  constexpr int BUF = OFF_X + 512;
  __shared__ __attribute__((aligned(256))) char smem[3 * BUF + (INPLACE ? HMK_MAXW * 4 : 0)];
  typedef __attribute__((ext_vector_type(4))) unsigned int hu32x4_t;
  typedef const __attribute__((address_space(3))) hu32x4_t lds_u4_t;
  typedef __attribute__((ext_vector_type(4))) float hf32x4_t;
  typedef const __attribute__((address_space(3))) hf32x4_t lds_f4_t;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hh = lane >> 5;
  const int lid = hm_xcd(blockIdx.x, gridDim.x);
  const int kc = lid % kchunks; const int t_ = lid / kchunks; const int h = t_ % H; const int w = t_ / H;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int k0 = kc * HMK_BKEYS;
  if (k0 >= L) return;
  const int64_t sec = (int64_t)H * NP * D;
  const int64_t hrow = (int64_t)h * NP + p0;
  const char* const qb_u = hm_uniform_ptr(hm + hrow * D);
  const char* const gb_u = hm_uniform_ptr(INPLACE ? doh + h * D : doh + hrow * D);       // INPLACE: doh = the (n, C) dO tensor
  const char* const lb_u = hm_uniform_ptr(nlse2 + hrow);
  const char* const db_u = hm_uniform_ptr(ndelta + hrow);
  const int ntiles = (L + HMK_BQ - 1) / HMK_BQ;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int32_t* const srow_s = reinterpret_cast<int32_t*>(smem + 3 * BUF);
  // tile t, ring slot at slot_bytes; grow = the (n, C) row of this lane's query slot (INPLACE; borrowed slots: row 0)
  auto issue_tile = [&](int t, unsigned slot_bytes, int grow) {
    const int q0 = min(t, ntiles - 1) * HMK_BQ;
    const int row = min(q0 + lane, L - 1);
    const unsigned goff = INPLACE ? (unsigned)grow * (2u * (unsigned)C) : 0u;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pi = wave_u * NPW + j;
      const int isg = pi >= A::CH, c = isg ? pi - A::CH : pi;
      const unsigned off = (INPLACE && isg) ? goff + c * 16 : __umul24((unsigned)row, 2u * D) + c * 16;
      hm_glds16s(isg ? gb_u : qb_u, off, lds0 + slot_bytes + isg * A::IMG + A::plane(c));
    }
    // waves 0 / 1: -lse2 / -delta of the 64 queries; waves 2 / 3 repeat them into a dummy target (uniform vmcnt)
    hm_glds4s((wave_u & 1) ? db_u : lb_u, (unsigned)row * 4u,
              lds0 + slot_bytes + (wave_u >= 2 ? OFF_X + (wave_u & 1) * 256 : ((wave_u & 1) ? OFF_D : OFF_L)));
  };
  if (INPLACE) {
    // The window's row map goes to LDS for tiles 2 ..; tiles 0 and 1 take their rows straight from registers, so the first DMA
    // pieces wait for ONE global round trip of a 4-byte load instead of a staging pass + workgroup barrier (the q~ pieces -- waves
    // 0 / 1 -- do not wait at all: the row map only addresses dO).  The loop's first barrier publishes the table.
    constexpr int NST = HMK_MAXW / HMK_THREADS;
    int32_t st[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) { const int e = tid + i * HMK_THREADS; st[i] = e < L ? sidx[p0 + e] : 0; }
    const int g0 = max(sidx[p0 + min(lane, L - 1)], 0);
    const int g1 = max(sidx[p0 + min(min(1, ntiles - 1) * HMK_BQ + lane, L - 1)], 0);
    issue_tile(0, 0u, g0);
    issue_tile(1, (unsigned)BUF, g1);
#pragma unroll
    for (int i = 0; i < NST; ++i) { const int e = tid + i * HMK_THREADS; if (e < L) srow_s[e] = max(st[i], 0); }
  } else {
    issue_tile(0, 0u, 0);
    issue_tile(1, (unsigned)BUF, 0);
  }

  // ---- this lane's key: K and V fragments (B operands): lane (key lr, half hh) holds K[key][16 ks + 8 hh .. +7] ----
  const int kslot = k0 + wave * 32 + lr;
  int32_t ksr = kslot < L ? sidx[p0 + kslot] : 0;
  hbf8_t kf[A::NKS], vf[A::NKS];
  {
    const unsigned short* kp = hm + sec + (hrow + min(kslot, L - 1)) * D + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      kf[ks] = hm_bf8(*reinterpret_cast<const uint4*>(kp + 16 * ks));
      vf[ks] = hm_bf8(*reinterpret_cast<const uint4*>(kp + sec + 16 * ks));
    }
  }
  asm volatile("" : "+v"(ksr));
#pragma unroll
  for (int ks = 0; ks < A::NKS; ++ks) { asm volatile("" : "+v"(kf[ks])); asm volatile("" : "+v"(vf[ks])); }
  f32x16_t dk[A::NMT], dv[A::NMT];
#pragma unroll
  for (int mt = 0; mt < A::NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[mt][r] = 0.f; dv[mt][r] = 0.f; }

  const unsigned lds_lane0 = (unsigned)(uintptr_t)smem;
  const unsigned roff = lds_lane0 + hh * A::PL + lr * 16;
  const unsigned coff = lds_lane0 + OFF_L + hh * 16;                  // -lse2 rows 8 g + 4 hh .. + 3 (broadcast inside a half-wave)
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg = lr >> 4;
  unsigned toff[A::NMT];
#pragma unroll
  for (int mt = 0; mt < A::NMT; ++mt) {
    const int cb = min(2 * mt + vg, A::NCB - 1);
    const int c = 2 * cb + (vp >> 1);
    toff[mt] = lds_lane0 + c * A::PL + (4 * hh + vq) * 16 + (vp & 1) * 8;
  }
  auto block = [&](const unsigned rb, const unsigned cb_, const unsigned (&tb)[A::NMT], const int qv0, const int IMM) {
    f32x16_t s, dp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {       // accumulator register 4 g + i <-> query row IMM + 8 g + 4 hh + i
      hf32x4_t a = *(lds_f4_t*)(cb_ + (IMM + 8 * g) * 4);
      hf32x4_t b = *(lds_f4_t*)(cb_ + 256 + (IMM + 8 * g) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[4 * g + i] = a[i]; dp[4 * g + i] = b[i]; }
    }
    if (qv0 + 32 > L) {               // query rows past the window end: P = 0 (last tile only; wave-uniform branch)
      asm volatile("; tail: mask queries past the window end" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (qv0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -INFINITY;
    }
#pragma unroll
    for (int ks = 0; ks < A::NKS; ++ks) {
      const int po = 2 * ks * A::PL + IMM * 16;
      hbf8_t qa = __builtin_bit_cast(hbf8_t, *(lds_u4_t*)(rb + po));
      hbf8_t ga = __builtin_bit_cast(hbf8_t, *(lds_u4_t*)(rb + A::IMG + po));
      s = HM_MFMA32(qa, kf[ks], s);
      dp = HM_MFMA32(ga, vf[ks], dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r]); dp[r] = s[r] * dp[r]; }
    hbf8_t pf[2], df[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      uint4 pk, dk_;
      pk.x = pack_bf16x2(s[8 * ss + 0], s[8 * ss + 1]); pk.y = pack_bf16x2(s[8 * ss + 2], s[8 * ss + 3]);
      pk.z = pack_bf16x2(s[8 * ss + 4], s[8 * ss + 5]); pk.w = pack_bf16x2(s[8 * ss + 6], s[8 * ss + 7]);
      dk_.x = pack_bf16x2(dp[8 * ss + 0], dp[8 * ss + 1]); dk_.y = pack_bf16x2(dp[8 * ss + 2], dp[8 * ss + 3]);
      dk_.z = pack_bf16x2(dp[8 * ss + 4], dp[8 * ss + 5]); dk_.w = pack_bf16x2(dp[8 * ss + 6], dp[8 * ss + 7]);
      pf[ss] = hm_bf8(pk); df[ss] = hm_bf8(dk_);
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int mt = 0; mt < A::NMT; ++mt) {
        const unsigned ta = tb[mt] + (IMM + 16 * ss) * 16;
        hbf8_t qt = hm_cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(ta)),
                           __builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(ta + 8 * 16)));
        hbf8_t gt = hm_cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(ta + A::IMG)),
                           __builtin_amdgcn_ds_read_tr16_b64_v4i16((hlds_s4_t*)(ta + A::IMG + 8 * 16)));
        dv[mt] = HM_MFMA32(gt, pf[ss], dv[mt]);
        dk[mt] = HM_MFMA32(qt, df[ss], dk[mt]);
      }
  };

  if (INPLACE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row map is written before the first barrier releases its readers
  unsigned slot_off = 0;
  for (int t = 0; t < ntiles; ++t) {
    if (NPW == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (NPW == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (NPW == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue_tile(t + 2, slot_off == 0 ? 2u * BUF : slot_off - BUF, INPLACE ? srow_s[min(min(t + 2, ntiles - 1) * HMK_BQ + lane, L - 1)] : 0);
    unsigned rb = roff + slot_off, cb_ = coff + slot_off, tb[A::NMT];
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt) tb[mt] = toff[mt] + slot_off;
    asm volatile("" : "+v"(rb));
    asm volatile("" : "+v"(cb_));
#pragma unroll
    for (int mt = 0; mt < A::NMT; ++mt) asm volatile("" : "+v"(tb[mt]));
    const int qv0 = t * HMK_BQ;
    block(rb, cb_, tb, qv0, 0);
    block(rb, cb_, tb, qv0 + 32, 32);
    slot_off = slot_off == 2 * BUF ? 0u : slot_off + BUF;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: dK = ln 2 * dK^T (q~ carries scale * log2 e), dV = dV^T of this lane's key; borrowed slots -> side buffer ----
  if (kslot < L) {
    unsigned short* kp = ksr >= 0 ? dqkv + (int64_t)ksr * 3 * C + C + h * D : extra + (int64_t)(-1 - ksr) * 2 * C + h * D;
    unsigned short* vp_ = kp + C;
    constexpr int NG = D / 8;
    constexpr float LN2 = 0.69314718055994530942f;
#pragma unroll
    for (int j = 0; j < (NG + 1) / 2; ++j) {
      const int ga = 2 * j, gb = 2 * j + 1;
      uint2 a, b, va, vb2;
      a.x = pack_bf16x2(dk[ga >> 2][4 * (ga & 3) + 0] * LN2, dk[ga >> 2][4 * (ga & 3) + 1] * LN2);
      a.y = pack_bf16x2(dk[ga >> 2][4 * (ga & 3) + 2] * LN2, dk[ga >> 2][4 * (ga & 3) + 3] * LN2);
      va.x = pack_bf16x2(dv[ga >> 2][4 * (ga & 3) + 0], dv[ga >> 2][4 * (ga & 3) + 1]);
      va.y = pack_bf16x2(dv[ga >> 2][4 * (ga & 3) + 2], dv[ga >> 2][4 * (ga & 3) + 3]);
      if (gb < NG) {
        b.x = pack_bf16x2(dk[gb >> 2][4 * (gb & 3) + 0] * LN2, dk[gb >> 2][4 * (gb & 3) + 1] * LN2);
        b.y = pack_bf16x2(dk[gb >> 2][4 * (gb & 3) + 2] * LN2, dk[gb >> 2][4 * (gb & 3) + 3] * LN2);
        vb2.x = pack_bf16x2(dv[gb >> 2][4 * (gb & 3) + 0], dv[gb >> 2][4 * (gb & 3) + 1]);
        vb2.y = pack_bf16x2(dv[gb >> 2][4 * (gb & 3) + 2], dv[gb >> 2][4 * (gb & 3) + 3]);
        auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        auto sx = __builtin_amdgcn_permlane32_swap(va.x, vb2.x, false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(va.y, vb2.y, false, false);
        *reinterpret_cast<uint4*>(kp + 8 * hh + 16 * j) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        *reinterpret_cast<uint4*>(vp_ + 8 * hh + 16 * j) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      } else {
        *reinterpret_cast<uint2*>(kp + 8 * ga + 4 * hh) = a;
        *reinterpret_cast<uint2*>(vp_ + 8 * ga + 4 * hh) = va;
      }
    }
  }
}

// inplace != 0: `doh` is the (n, C) dO tensor itself (see k_attn_hm_dkv)
int ss_attn_hm_dkv(const void* hm, int64_t np, const void* doh, const float* nlse2, const float* ndelta, const int32_t* sidx,
                   const int32_t* win_start, int W, int max_window, void* dqkv, void* extra, int C, int H, int inplace, hipStream_t st) {
  const int D = C / H;
  if ((C & 7) || max_window <= 0 || W <= 0 || (inplace && max_window > HMK_MAXW)) return SS_ERR_ARG;
  const int kchunks = (max_window + HMK_BKEYS - 1) / HMK_BKEYS;
  dim3 g((unsigned)(W * H * kchunks)), b(HMK_THREADS);
  const unsigned short* q = (const unsigned short*)hm;
#define SS_HK_CASE(DD) case DD: if (inplace) SS_LAUNCH((k_attn_hm_dkv<DD, true>), g, b, 0, st, q, np, (const unsigned short*)doh, nlse2, ndelta, sidx, win_start, (unsigned short*)dqkv, (unsigned short*)extra, C, H, kchunks); \
  else SS_LAUNCH((k_attn_hm_dkv<DD, false>), g, b, 0, st, q, np, (const unsigned short*)doh, nlse2, ndelta, sidx, win_start, (unsigned short*)dqkv, (unsigned short*)extra, C, H, kchunks); break;
  switch (D) {
    SS_HK_CASE(16) SS_HK_CASE(32) SS_HK_CASE(48) SS_HK_CASE(64)
    default: return SS_ERR_ARG;
  }
#undef SS_HK_CASE
  return SS_OK;
}

// dO in place is possible while the window's row map fits the LDS table and a row offset fits the DMA's 32-bit lane offset
static int hm_do_inplace(int64_t n, int channels, int max_window) {
  static int env = -1;
  // default OFF (measured, round 4, dec0 shape, one box, interleaved): the head-major copy costs 314 MB of traffic per launch pair
  // but its contiguous 1-KiB DMA pieces stream faster than 16-byte pieces of 64 scattered 1,536-byte rows -- dQ 0.641 -> 0.617 ms,
  // dK/dV 0.720 -> 0.766 ms, the pair 1.361 -> 1.384 ms.  SS_ATTN_DO_INPLACE=1 selects the in-place path (less traffic, 1.7 % slower).
  if (env < 0) { const char* e = getenv("SS_ATTN_DO_INPLACE"); env = (e && atoi(e) == 1) ? 1 : 0; }
  return env && max_window <= HMK_MAXW && (uint64_t)n * (uint64_t)channels * 2ull < (1ull << 32);
}

// =====================================================================================
// head-major packing of an existing (n, sections * C) projection (fallback for levels whose projection is not eligible
// for the fused epilogue of gemm8.hip: k < 64 or not a multiple of 64).  in_dtype SS_F32: the library GEMM's fp32 result, so
// q is scaled and rounded ONCE, exactly as in the fused epilogue.  One thread per 16-byte output chunk, output order.
// =====================================================================================
template <typename InT>
__global__ void k_hm_pack(const InT* __restrict__ src, const int32_t* __restrict__ gidx, unsigned short* __restrict__ hm,
                          int64_t NP, int C, int H, int D, int nsec, float sec0_scale) {
  const int CH = D / 8;
  const int64_t total = (int64_t)nsec * H * NP * CH;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(e % CH); int64_t r = e / CH;
    const int64_t p = r % NP; r /= NP;
    const int h = (int)(r % H); const int sec = (int)(r / H);
    const InT* sp = src + (int64_t)gidx[p] * nsec * C + (int64_t)sec * C + h * D + ch * 8;
    const float sc = sec == 0 ? sec0_scale : 1.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ElemIO<InT>::load(sp + i) * sc;
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(hm + e * 8) = o;
  }
}

extern "C" int ss_headmajor_pack(const void* src, int in_dtype, const int32_t* gidx, void* hm, int64_t n_pad, int channels,
                                 int num_heads, int sections, float sec0_scale, hipStream_t stream) {
  if (channels <= 0 || num_heads <= 0 || channels % num_heads || ((channels / num_heads) & 7) || sections <= 0) return SS_ERR_ARG;
  if (n_pad == 0) return SS_OK;
  const int D = channels / num_heads;
  const int64_t total = (int64_t)sections * n_pad * channels / 8;
  dim3 g((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), b(256);
  if (in_dtype == SS_F32)
    SS_LAUNCH((k_hm_pack<float>), g, b, 0, stream, (const float*)src, gidx, (unsigned short*)hm, n_pad, channels, num_heads, D, sections, sec0_scale);
  else if (in_dtype == SS_BF16)
    SS_LAUNCH((k_hm_pack<unsigned short>), g, b, 0, stream, (const unsigned short*)src, gidx, (unsigned short*)hm, n_pad, channels, num_heads, D, sections, sec0_scale);
  else
    return SS_ERR_ARG;
  return SS_OK;
}

// =====================================================================================
// C-ABI of the head-major window attention (include/scenesplat_hip.h)
// =====================================================================================
static inline size_t hm_al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int ss_window_attn_hm_fwd(const void* hm, const int32_t* sidx, const int32_t* win_start, int num_windows,
                                     int max_window, int64_t n, int64_t n_pad, int channels, int num_heads, void* out,
                                     float* neg_lse2, hipStream_t stream) {
  if (num_windows < 0 || channels <= 0 || num_heads <= 0 || channels % num_heads || n_pad < n) return SS_ERR_ARG;
  if (num_windows == 0) return SS_OK;
  return ss_attn_hm_fwd(hm, n_pad, sidx, win_start, num_windows, max_window, out, neg_lse2, channels, num_heads, 1.f, stream);
}

extern "C" size_t ss_window_attn_hm_bwd_workspace_bytes(int64_t n, int64_t n_pad, int channels, int num_heads) {
  return hm_al256((size_t)n_pad * num_heads * 4) + hm_al256((size_t)n_pad * channels * 2) + hm_al256((size_t)(n_pad - n) * 2 * channels * 2);
}

extern "C" int ss_window_attn_hm_bwd(const void* hm, const void* out, const void* dout, const float* neg_lse2,
                                     const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int num_windows,
                                     int max_window, int64_t n, int64_t n_pad, int channels, int num_heads, float scale,
                                     void* dqkv, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (num_windows < 0 || channels <= 0 || num_heads <= 0 || channels % num_heads || n_pad < n) return SS_ERR_ARG;
  if (workspace_bytes < ss_window_attn_hm_bwd_workspace_bytes(n, n_pad, channels, num_heads)) return SS_ERR_WORKSPACE;
  if (num_windows == 0) return SS_OK;
  float* ndelta = (float*)workspace;
  char* doh = (char*)workspace + hm_al256((size_t)n_pad * num_heads * 4);
  char* extra = doh + hm_al256((size_t)n_pad * channels * 2);
  const int inplace = hm_do_inplace(n, channels, max_window);
  int rc = ss_attn_hm_dq(hm, n_pad, dout, out, neg_lse2, ndelta, inplace ? nullptr : doh, sidx, win_start, num_windows, max_window, dqkv,
                         channels, num_heads, scale, stream);
  if (rc) return rc;
  rc = ss_attn_hm_dkv(hm, n_pad, inplace ? dout : (const void*)doh, neg_lse2, ndelta, sidx, win_start, num_windows, max_window, dqkv, extra,
                      channels, num_heads, inplace, stream);
  if (rc) return rc;
  if (n_pad > n) rc = ss_attn_fix_borrowed(gidx, sidx, n_pad, extra, dqkv, channels, SS_BF16, stream);
  return rc;
}
