// Weight-gradient GEMM on the LDS-DMA pipeline of gemm8.hip (gfx950):  256 x 256 tile, K = sites.
//
//   dW[co][tap][ci] += sum_i dout[i][co] * in[row_tap(i)][ci]        (fp32 atomics into a zeroed dW)
//
//   GATHER = true   spconv.SubMConv3d weight gradient (ptv3:278-284): row_tap(i) = nbr[tap][i] (pair skipped when
//                   missing); per tap only the ACTIVE 64-site blocks (ss_subm_block_lists) are walked, split over
//                   `nshares` workgroups
//   GATHER = false  nn.Linear weight gradient dW = dy^T x: taps = 1, row(i) = i, every block active
//
// Both operands are [site][channel]-major, i.e. K-strided for the MFMA.  The LDS images stay plain [site][256 B]
// rows -- what LDS-DMA writes, 4 site rows per 1-KiB wave instruction -- and the fragments are read with
// ds_read_b64_tr_b16 (the same k permutation on both operands).  The 16-byte chunk index is XOR-swizzled per row,
// chunk ^= (row & 7) << 1 (applied to the DMA SOURCE column, and again on the read): a transposed read is served in
// two 32-lane groups of 8 rows x 32 B, and the 8 rows must fall on 8 different 32-byte bank segments
// (SQ_LDS_BANK_CONFLICT went from 48 % of the LDS cycles to 0).
// Pipeline, phases, staggered wave rows and hazard rules: exactly gemm8.hip's (see its header).
// GATHER: the (site, neighbour) ids of up to W8_CHUNK K-tiles live in LDS; longer shares are walked chunk by chunk.
#include "common.h"
#include "../../include/scenesplat_hip.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 w8_bf8_t;
typedef __attribute__((ext_vector_type(4))) short w8_s4_t;
typedef __attribute__((ext_vector_type(8))) short w8_s8_t;
typedef __attribute__((address_space(3))) w8_s4_t w8_lds_s4_t;
#define W8_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

#define W8_BUF 65536
#define W8_HALF 16384
#define W8_OFF_B 32768
#define W8_CHUNK 60
#define W8_OFF_ISITE 131072
#define W8_OFF_JSITE (W8_OFF_ISITE + W8_CHUNK * 64 * 4)
#define W8_LDS_BYTES (W8_OFF_JSITE + W8_CHUNK * 64 * 4)
#define W8_ZERO_ELEMS 4096

__device__ uint4 w8_zero[W8_ZERO_ELEMS / 8];

// LDS-DMA issued as inline asm: with the builtin, hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every
// ds_read_b64_tr_b16 (it cannot tell the transposed read from a reader of the pending DMA) and the pipeline drains
// every phase.  Ordering is by the counted vmcnt + barriers of the schedule, as documented in gemm8.hip.
__device__ __forceinline__ void w8_glds16(const void* gsrc, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ w8_bf8_t w8_tr2(const char* p) {       // k rows r0 and r0+16 of one 16-column group
  w8_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w8_lds_s4_t*)(p));
  w8_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w8_lds_s4_t*)(p + 16 * 256));
  w8_s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(w8_bf8_t, v);
}

// The whole workgroup program; L = the workgroup's id inside its problem (a plain launch passes blockIdx.x, the grouped
// launch blockIdx.x minus the problem's first workgroup).
template <bool GATHER>
__device__ __forceinline__ void
w8_body(const unsigned short* __restrict__ X, const unsigned short* __restrict__ DY, const int32_t* __restrict__ nbr,
        const int32_t* __restrict__ rowperm, const int32_t* __restrict__ blk_count, const int32_t* __restrict__ blk_list,
        float* __restrict__ dW, float* __restrict__ dbias, int n, int Cin, int Cout, int taps, int ntn, int nblocks_total,
        int min_per, int ntiles, int nshares, const int L, const int walk = 0) {
  __shared__ __attribute__((aligned(16))) char smem[GATHER ? W8_LDS_BYTES : 2 * W8_BUF];
  int32_t* isite_s = reinterpret_cast<int32_t*>(smem + W8_OFF_ISITE);
  int32_t* jsite_s = reinterpret_cast<int32_t*>(smem + W8_OFF_JSITE);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  // ids in (share, tap, tile) order, tile fastest: the tiles of one (tap, share) -- which stream the SAME site rows -- are
  // dealt round-robin over the XCDs.  Measured: giving them to ONE XCD (to share its L2) is 20-45 % SLOWER (768 -> 768
  // Linear 191 -> 232 us, conv dec0 1.20 -> 1.64 ms): eight L2s and the Infinity Cache serve the re-reads better than one.
  const int tile = L % ntiles, group = L / ntiles;                             // group = share * taps + tap
  const int tap = group % taps, share = group / taps;
  const int m0 = (tile / ntn) * 256, n0 = (tile % ntn) * 256;                  // Cout, Cin origins
  const int cnt = GATHER ? blk_count[tap] : nblocks_total;
  // a share is at least min_per K-tiles (the 256 KiB fp32-atomic epilogue is paid per share); surplus workgroups exit
  const int per = max((cnt + nshares - 1) / nshares, min_per);
  const int beg = share * per, end = min(cnt, beg + per);
  if (end <= beg) return;
  const int32_t* list = GATHER ? blk_list + (int64_t)tap * nblocks_total : nullptr;

  f32x4_t acc[2][4][2][2];                      // [A half][mi][B half][ni]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[a][mi][b][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // column sums of dout (the bias gradient) ride along in the workgroups of the first Cin tile: wave column wc adds
  // its 16-channel group mi = wc of both A halves with an all-ones B fragment (4 extra MFMAs per K-tile)
  const bool do_bias = !GATHER && dbias != nullptr && (tile % ntn) == 0;
  f32x4_t accb[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  w8_bf8_t ones;
  {
    w8_s8_t o = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ones = __builtin_bit_cast(w8_bf8_t, o);
  }

  // ---- staging: wave-instruction (j, wave) fills site rows (j*8+wave)*4 + (lane>>4), 16-byte slot lane&15 ----
  const int srow = wave * 4 + (lane >> 4);                         // + 32 j
  const int ssw = ((srow & 7) << 1);                               // swizzle of those rows (same for j = 0, 1)
  const int lch = (lane & 15) ^ ssw;                               // logical chunk held by the slot
  int colA[2], colB[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    colA[h] = min(m0 + (lch >> 3) * 128 + h * 64 + (lch & 7) * 8, Cout - 8);
    colB[h] = min(n0 + (lch >> 2) * 64 + h * 32 + (lch & 3) * 8, Cin - 8);
  }
  const unsigned short* const zrow = reinterpret_cast<const unsigned short*>(w8_zero);
  const unsigned short* pA[2];
  const unsigned short* pB[2];
  int chunk_beg = beg, T = 0, staged = 0;
  int nis[2] = {-1, -1}, njs[2] = {-1, -1};    // site / neighbour ids of the NEXT tile to stage (GATHER)
  // read one phase early (ph4, which has no fragment reads) so ph2 only does pointer arithmetic
  auto prefetch_idx = [&]() {
    if (!GATHER || staged + 1 >= T) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      nis[j] = isite_s[(staged + 1) * 64 + srow + 32 * j];
      njs[j] = jsite_s[(staged + 1) * 64 + srow + 32 * j];
    }
  };
  auto make_rows = [&](int s) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (GATHER) {
        pA[j] = njs[j] >= 0 ? DY + (int64_t)nis[j] * Cout : zrow;
        pB[j] = njs[j] >= 0 ? X + (int64_t)njs[j] * Cin : zrow;
      } else {
        int64_t k = (int64_t)(chunk_beg + s) * 64 + srow + 32 * j;
        pA[j] = k < n ? DY + k * Cout : zrow;
        pB[j] = k < n ? X + k * Cin : zrow;
      }
    }
  };
  auto advance = [&]() {
    if (staged + 1 >= T) return;
    ++staged;
    make_rows(staged);
  };
  const unsigned dst0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + (unsigned)wave * 1024u);
  auto stageA = [&](int buf, int h) {
    unsigned d = dst0 + buf * W8_BUF + h * W8_HALF;
    w8_glds16(pA[0] + colA[h], d);
    w8_glds16(pA[1] + colA[h], d + 8192);
  };
  auto stageB = [&](int buf, int h) {
    unsigned d = dst0 + buf * W8_BUF + W8_OFF_B + h * W8_HALF;
    w8_glds16(pB[0] + colB[h], d);
    w8_glds16(pB[1] + colB[h], d + 8192);
  };

  // ---- fragment reads: k rows 32kk + 4g + (lq>>2) (+16), 16-column group -> chunks 2i, 2i+1; swizzle is per lane ----
  const int rsw = ((4 * (g & 1) + (lq >> 2)) << 1);                // (row & 7) << 1
  const int rbase = (4 * g + (lq >> 2)) * 256 + 8 * (lq & 1);
  int oA[4], oB[2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) oA[mi] = rbase + (((wr * 8 + 2 * mi + ((lq & 3) >> 1)) ^ rsw) << 4);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) oB[ni] = W8_OFF_B + rbase + (((wc * 4 + 2 * ni + ((lq & 3) >> 1)) ^ rsw) << 4);
  w8_bf8_t af[4][2], b0f[2][2], b1f[2][2];
  auto readA = [&](int buf, int h) {
    const char* base = smem + buf * W8_BUF + h * W8_HALF;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      af[mi][0] = w8_tr2(base + oA[mi]);
      af[mi][1] = w8_tr2(base + oA[mi] + 32 * 256);
    }
  };
  auto readB = [&](int buf, int h, w8_bf8_t (&bf)[2][2]) {
    const char* base = smem + buf * W8_BUF + h * W8_HALF;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      bf[ni][0] = w8_tr2(base + oB[ni]);
      bf[ni][1] = w8_tr2(base + oB[ni] + 32 * 256);
    }
  };
#define W8_MM(HA, HB, BF)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                \
      acc[HA][mi][HB][ni] = W8_MFMA(af[mi][ks], BF[ni][ks], acc[HA][mi][HB][ni]);
#define W8_BIAS(HA)                                                                               \
  if (do_bias) {                                                                                  \
    if (wc == 0) { accb[HA] = W8_MFMA(af[0][0], ones, accb[HA]); accb[HA] = W8_MFMA(af[0][1], ones, accb[HA]); }       \
    else if (wc == 1) { accb[HA] = W8_MFMA(af[1][0], ones, accb[HA]); accb[HA] = W8_MFMA(af[1][1], ones, accb[HA]); }  \
    else if (wc == 2) { accb[HA] = W8_MFMA(af[2][0], ones, accb[HA]); accb[HA] = W8_MFMA(af[2][1], ones, accb[HA]); }  \
    else { accb[HA] = W8_MFMA(af[3][0], ones, accb[HA]); accb[HA] = W8_MFMA(af[3][1], ones, accb[HA]); }              \
  }
#define W8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W8_COMPUTE_BEGIN() do { W8_BAR(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); } while (0)
#define W8_COMPUTE_END() do { __builtin_amdgcn_s_setprio(0); W8_BAR(); } while (0)

  for (; chunk_beg < end; chunk_beg += (GATHER ? W8_CHUNK : (1 << 30))) {
    T = GATHER ? min(W8_CHUNK, end - chunk_beg) : end - chunk_beg;
    if (GATHER) {
      __syncthreads();                           // previous chunk's index reads are done
      // the chunk's (site, neighbour) pairs: block id -> walk position k -> site rowperm[k] -> neighbour nbr[tap][site], a load +
      // use per element = 24 serial round trips per chunk (15-19 % of a chunk's time at dec0).  walk != 0: nbr is in WALK order
      // (nbr_walk[t][k] = nbr[t][rowperm[k]]): the neighbour does not wait for the site, a wave reads 256 contiguous bytes of it,
      // and the two levels are batched four elements deep (more costs registers the K loop needs)
      if (walk) {
        constexpr int NH = 4;                                      // 4 x 512 elements per batch, two batches per chunk
        for (int e0 = 0; e0 < T * 64; e0 += NH * 512) {
          int kk[NH], is_[NH], js_[NH];
          const int laste = T * 64 - 1;
#pragma unroll
          for (int it = 0; it < NH; ++it) {
            const int e = min(e0 + it * 512 + tid, laste);
            kk[it] = list[chunk_beg + (e >> 6)] * 64 + (e & 63);
          }
#pragma unroll
          for (int it = 0; it < NH; ++it) {
            const int kc = min(kk[it], n - 1);
            is_[it] = rowperm ? rowperm[kc] : kc;
            js_[it] = nbr[(int64_t)tap * n + kc];
          }
#pragma unroll
          for (int it = 0; it < NH; ++it) {
            const int e = e0 + it * 512 + tid;
            if (e <= laste) {
              const bool ok = kk[it] < n;
              isite_s[e] = ok ? is_[it] : -1;
              jsite_s[e] = ok ? js_[it] : -1;
            }
          }
        }
      } else {
        for (int e = tid; e < T * 64; e += 512) {
          int k = list[chunk_beg + (e >> 6)] * 64 + (e & 63);
          int is = k < n ? (rowperm ? rowperm[k] : k) : -1;
          isite_s[e] = is;
          jsite_s[e] = is >= 0 ? nbr[(int64_t)tap * n + is] : -1;
        }
      }
      __syncthreads();
    }
    staged = -1;
    prefetch_idx();
    staged = 0;
    make_rows(0);
    prefetch_idx();
    // prologue: tile 0 complete + B0 A0 B1 of tile 1
    stageB(0, 0); stageA(0, 0); stageB(0, 1); stageA(0, 1);
    advance();
    prefetch_idx();                              // tile 2, consumed by ph2 of the first iteration
    stageB(1, 0); stageA(1, 0); stageB(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    W8_BAR();
    if (wr == 1) W8_BAR();                       // wave row 1 runs one barrier behind
    for (int t = 0; t < T; ++t) {
      const int buf = t & 1;
      // ph1
      readB(buf, 0, b0f);
      __builtin_amdgcn_sched_barrier(0);
      readA(buf, 0);
      stageA(buf ^ 1, 1);                        // A1 of tile t+1
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");   // 24 reads issued, B0's 8 first: retired before the barrier
      W8_COMPUTE_BEGIN();
      W8_MM(0, 0, b0f)
      W8_COMPUTE_END();
      // ph2
      advance();                                 // state -> tile t+2
      readB(buf, 1, b1f);
      stageB(buf, 0);
      W8_COMPUTE_BEGIN();
      W8_MM(0, 1, b1f)
      W8_BIAS(0)
      W8_COMPUTE_END();
      // ph3
      readA(buf, 1);
      stageA(buf, 0);
      W8_COMPUTE_BEGIN();
      W8_MM(1, 1, b1f)
      W8_COMPUTE_END();
      // ph4
      stageB(buf, 1);
      prefetch_idx();                            // ids of tile staged+1, retired by this phase's lgkmcnt(0)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      W8_COMPUTE_BEGIN();
      W8_MM(1, 0, b0f)
      W8_BIAS(1)
      W8_COMPUTE_END();
    }
    if (wr == 0) W8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // ---- epilogue: acc[ha][mi][hb][ni][r] = dW[co = 128wr + 64ha + 16mi + 4g + r][ci = 64wc + 32hb + 16ni + lq] ----
#pragma unroll
  for (int ha = 0; ha < 2; ++ha)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = m0 + 128 * wr + 64 * ha + 16 * mi + 4 * g + r;
        if (co >= Cout) continue;
        float* rowp = dW + ((int64_t)co * taps + tap) * Cin;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int ci = n0 + 64 * wc + 32 * hb + 16 * ni + lq;
            if (ci < Cin) atomicAdd(rowp + ci, acc[ha][mi][hb][ni][r]);
          }
      }
  if (do_bias && lq == 0) {
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = m0 + 128 * wr + 64 * ha + 16 * wc + 4 * g + r;
        if (co < Cout) atomicAdd(dbias + co, accb[ha][r]);
      }
  }
}

#undef W8_BIAS

// XCD-aware order of the output tiles (round 4).  Workgroup id i is dispatched to XCD i % 8.  Tiles that are consecutive in L (Cin-tile
// fastest) stream the SAME dy rows; dealt round-robin over the XCDs (rounds 1-3) every one of them fetches those rows through its own
// L2.  Here runs of W8_XG = 3 consecutive tiles go to ONE XCD in consecutive slots, the runs round-robin over the XCDs: measured on
// the dec0 conv weight gradient 1.077 -> 1.041 ms stand-alone, but +0.25 ms on the whole step (see ss_wgrad_xcd_order: off by default;
// all NINE tiles of a (tap, share) on one XCD was 20-45 % slower, round 1).  `first` = the id of the problem's first workgroup (a multiple of 8), nb = its workgroup count.
#define W8_XG 3
__device__ __forceinline__ int w8_xcd_order(int L, int nb) {
  const int row = 8 * W8_XG, full = nb / row * row;
  if (L >= full) return L;
  const int xcd = L & 7, slot = L >> 3;
  return ((slot / W8_XG) * 8 + xcd) * W8_XG + slot % W8_XG;
}

template <bool GATHER>
__global__ void __launch_bounds__(512)
k_wgrad8(const unsigned short* __restrict__ X, const unsigned short* __restrict__ DY, const int32_t* __restrict__ nbr,
         const int32_t* __restrict__ rowperm, const int32_t* __restrict__ blk_count, const int32_t* __restrict__ blk_list,
         float* __restrict__ dW, float* __restrict__ dbias, int n, int Cin, int Cout, int taps, int ntn, int nblocks_total,
         int min_per, int ntiles, int nshares, int walk) {
  // walk bit 1: XCD-aware tile order (w8_xcd_order; SS_WGRAD_XCD_TRIPLE=0 restores the round-robin order)
  int L = (int)blockIdx.x;
  if (walk & 2) L = w8_xcd_order(L, (int)gridDim.x);
  w8_body<GATHER>(X, DY, nbr, rowperm, blk_count, blk_list, dW, dbias, n, Cin, Cout, taps, ntn, nblocks_total, min_per, ntiles,
                  nshares, L, walk & 1);
}

// Grouped nn.Linear weight gradients: ONE launch for many independent problems (the six identical blocks of a pooled
// stage produce thirty weight gradients of 20-30 us each, latency-bound on 12-20 workgroups apiece).  desc: 8 int64 words
// per problem = {x, dy, dW, dbias, m, k_in | n_out << 32, ntn | ntiles << 32, nshares | min_per << 32}; wg_start
// (nprob + 1): first workgroup of every problem.
__global__ void __launch_bounds__(512)
k_wgrad8_group(const int64_t* __restrict__ desc, const int32_t* __restrict__ wg_start, int nprob, int nprob_flags) {
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < nprob && wg_start[p + 1] <= b) ++p;          // nprob <= 128: a scalar scan
  const int64_t* d = desc + (int64_t)p * 8;
  const int64_t m = d[4];
  const uint64_t w5 = (uint64_t)d[5], w6 = (uint64_t)d[6], w7 = (uint64_t)d[7];
  const int k_in = (int)(w5 & 0xffffffffu), n_out = (int)(w5 >> 32);
  const int ntn = (int)(w6 & 0xffffffffu), ntiles = (int)(w6 >> 32);
  const int nshares = (int)(w7 & 0xffffffffu), min_per = (int)(w7 >> 32);
  // XCD-aware tile order inside the problem: the workgroups before the first id that is a multiple of 8 keep their place, the rest is
  // ordered as if the problem started there (no padding workgroups: a launch is planned as ONE round of <= 256 workgroups, and ids that
  // leave at once would skew the XCDs' shares of the real ones -- measured +2.6 ms per step with padded problems)
  int L = b - wg_start[p];
  if (nprob_flags & 1) {
    const int a = (8 - (wg_start[p] & 7)) & 7;
    if (L >= a) L = a + w8_xcd_order(L - a, nshares * ntiles - a);
  }
  w8_body<false>((const unsigned short*)d[0], (const unsigned short*)d[1], nullptr, nullptr, nullptr, nullptr, (float*)d[2],
                 (float*)d[3], (int)m, k_in, n_out, 1, ntn, (int)((m + 63) / 64), min_per, ntiles, nshares, L);
}

// XCD-aware tile order of the conv weight gradient: default OFF.  Stand-alone it wins (dec0 1.077 -> 1.041 ms, dec1 188 -> 184 us, two
// repetitions), inside the step it loses (in-process interleaved A/B, scripts/ab_step.py wgrad_xcd: 37.48 ms with, 37.23 without).
static int w8_xcd_flag = -1;
extern "C" int ss_wgrad_xcd_order(void) {
  if (w8_xcd_flag < 0) { const char* e = getenv("SS_WGRAD_XCD_TRIPLE"); w8_xcd_flag = (e && atoi(e) == 1) ? 1 : 0; }
  return w8_xcd_flag;
}
// A/B switch of scripts/ab_step.py (a process-wide tuning knob, like the environment variable it overrides)
extern "C" int ss_wgrad_set_xcd_order(int on) { w8_xcd_flag = on ? 1 : 0; return SS_OK; }

extern "C" int ss_wgrad8_ok(int64_t n, int cin, int cout, int taps) {
  return n > 0 && n < (1LL << 31) && cin >= 8 && (cin & 7) == 0 && cin <= W8_ZERO_ELEMS && cout >= 8 && (cout & 7) == 0 &&
         cout <= W8_ZERO_ELEMS && taps >= 1;
}

// K-tiles per share.  Measured (scripts/tune_wgrad.py): the 256-KiB atomic epilogue and the pipeline fill make about
// 200 shares the sweet spot until shares reach ~160 K-tiles; est_blocks = expected active blocks of a busy tile.
static int w8_min_per(int64_t busy_tiles, int64_t est_blocks) {
  int64_t per = busy_tiles * est_blocks / 200;
  if (per < 8) per = 8;
  if (per > 160) per = 160;
  return (int)per;
}

static int w8_conv_launch(const void* in, const void* dout, const int32_t* nbr, const int32_t* rowperm, const int32_t* blk_count,
                          const int32_t* blk_list, float* dweight, int64_t n, int cin, int cout, int taps, int walk, hipStream_t stream);

extern "C" int ss_subm_conv_wgrad_pipe(const void* in, const void* dout, const int32_t* nbr, const int32_t* rowperm,
                                       const int32_t* blk_count, const int32_t* blk_list, float* dweight, int64_t n,
                                       int cin, int cout, int taps, hipStream_t stream) {
  return w8_conv_launch(in, dout, nbr, rowperm, blk_count, blk_list, dweight, n, cin, cout, taps, 0, stream);
}
// the same kernel reading the rulebook in WALK order (nbr_walk[t][k] = nbr[t][rowperm[k]]; blk_* as before)
extern "C" int ss_subm_conv_wgrad_pipe_walk(const void* in, const void* dout, const int32_t* nbr_walk, const int32_t* rowperm,
                                            const int32_t* blk_count, const int32_t* blk_list, float* dweight, int64_t n,
                                            int cin, int cout, int taps, hipStream_t stream) {
  return w8_conv_launch(in, dout, nbr_walk, rowperm, blk_count, blk_list, dweight, n, cin, cout, taps, 1, stream);
}

static int w8_conv_launch(const void* in, const void* dout, const int32_t* nbr, const int32_t* rowperm, const int32_t* blk_count,
                          const int32_t* blk_list, float* dweight, int64_t n, int cin, int cout, int taps, int walk, hipStream_t stream) {
  if (n == 0) return SS_OK;
  if (!ss_wgrad8_ok(n, cin, cout, taps) || !blk_count || !blk_list || !nbr) return SS_ERR_ARG;
  const int nblocks = ss_div_up(n, 64);
  const int tm = ss_div_up(cout, 256), tn = ss_div_up(cin, 256);
  // the active-block counts live on the device: launch enough shares for a fully active tap and let the kernel
  // size them (>= min_per K-tiles each); on surfaces about a third of the taps carry almost all pairs
  static int env_per = -2;
  if (env_per == -2) { const char* e = getenv("SS_WGRAD_MINPER"); env_per = e ? atoi(e) : -1; }
  int min_per = env_per > 0 ? env_per : w8_min_per((int64_t)tm * tn * ((taps + 2) / 3), (int64_t)nblocks * 7 / 10);
  int splits = ss_div_up(nblocks, min_per);
  if (ss_wgrad_xcd_order()) walk |= 2;
  dim3 g((unsigned)((int64_t)taps * splits * tm * tn));
  SS_LAUNCH((k_wgrad8<true>), g, dim3(512), 0, stream, (const unsigned short*)in, (const unsigned short*)dout, nbr, rowperm,
            blk_count, blk_list, dweight, (float*)nullptr, (int)n, cin, cout, taps, tn, nblocks, min_per, tm * tn, splits, walk);
  return SS_OK;
}

// share sizing of one nn.Linear weight gradient: every block is active here, so the shares are exact: one round of
// <= 256 workgroups (one per CU), shares of at least 8 K-tiles (270 workgroups on 256 CUs ran two rounds: qkv 768->2304
// took 622 us, 243 workgroups 440 us)
static void w8_linear_plan(int64_t m, int k_in, int n_out, int* tm_, int* tn_, int* min_per_, int* splits_, int launch_tiles = 0) {
  const int nblocks = ss_div_up(m, 64);
  const int tm = ss_div_up(n_out, 256), tn = ss_div_up(k_in, 256);
  static int env_per = -2;
  if (env_per == -2) { const char* e = getenv("SS_WGRAD_MINPER"); env_per = e ? atoi(e) : -1; }
  // shares per tile = floor(CUs / tiles): rounding the share LENGTH instead (ceil(nblocks tiles / 256)) let the workgroup
  // count overshoot the CU count (27 tiles x 10 shares = 270, 36 x 8 = 288, 9 x 29 = 261) and every such launch ran a
  // second, almost empty round: one workgroup per CU (128 KiB of LDS each)
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  // launch_tiles: output tiles of ALL problems of a grouped launch (0: this problem is the whole launch) -- the CUs are shared out over them
  const int tiles_all = launch_tiles > tm * tn ? launch_tiles : tm * tn;
  const int shares = cus / tiles_all > 0 ? cus / tiles_all : 1;
  int min_per = env_per > 0 ? env_per : (int)((nblocks + shares - 1) / shares);
  if (env_per <= 0 && min_per < 8) min_per = 8;
  if (min_per > nblocks) min_per = nblocks;
  *tm_ = tm; *tn_ = tn; *min_per_ = min_per; *splits_ = ss_div_up(nblocks, min_per);
}

// Host half of the grouped launch: fills desc words 5..7 of one problem (words 0..4 = x, dy, dW, dbias, m are the
// caller's) and returns the number of workgroups the problem takes (0 = not eligible).
extern "C" int ss_linear_wgrad_tiles(int k_in, int n_out) { return ss_div_up(n_out, 256) * ss_div_up(k_in, 256); }
// launch_tiles = sum of ss_linear_wgrad_tiles over the problems of the launch: one round of <= CUs workgroups for the GROUP (planning every
// problem as if it had the chip to itself gave ten problems 240 workgroups each, 20-K-tile shares and 630 MB of fp32 atomics per launch)
extern "C" int ss_linear_wgrad_group_plan2(int64_t m, int k_in, int n_out, int launch_tiles, int64_t* desc_words) {
  if (m <= 0 || !ss_wgrad8_ok(m, k_in, n_out, 1) || !desc_words) return 0;
  int tm, tn, min_per, splits;
  w8_linear_plan(m, k_in, n_out, &tm, &tn, &min_per, &splits, launch_tiles);
  desc_words[5] = (int64_t)((uint64_t)(uint32_t)k_in | ((uint64_t)(uint32_t)n_out << 32));
  desc_words[6] = (int64_t)((uint64_t)(uint32_t)tn | ((uint64_t)(uint32_t)(tm * tn) << 32));
  desc_words[7] = (int64_t)((uint64_t)(uint32_t)splits | ((uint64_t)(uint32_t)min_per << 32));
  return splits * tm * tn;
}
extern "C" int ss_linear_wgrad_group_plan(int64_t m, int k_in, int n_out, int64_t* desc_words) {
  return ss_linear_wgrad_group_plan2(m, k_in, n_out, 0, desc_words);
}

// desc (nprob, 8) int64 and wg_start (nprob + 1) int32 in DEVICE memory (see k_wgrad8_group); every dW / dbias zeroed.
extern "C" int ss_linear_wgrad_group(const int64_t* desc, const int32_t* wg_start, int nprob, int total_workgroups,
                                     hipStream_t stream) {
  if (nprob <= 0 || total_workgroups <= 0) return SS_OK;
  if (!desc || !wg_start || nprob > 128) return SS_ERR_ARG;
  // (the XCD-aware order is NOT applied to the Linear problems: in-process A/B of the whole step 36.79 ms with it, 36.58 without -- their
  // launches are ONE round of <= 256 workgroups, every tile of a run would be resident at once on neighbouring CUs of one XCD)
  SS_LAUNCH(k_wgrad8_group, dim3((unsigned)total_workgroups), dim3(512), 0, stream, desc, wg_start, nprob, 0);
  return SS_OK;
}

extern "C" int ss_linear_wgrad(const void* x, const void* dy, float* dweight, float* dbias, int64_t m, int k_in, int n_out,
                               hipStream_t stream) {
  if (m == 0) return SS_OK;
  if (!ss_wgrad8_ok(m, k_in, n_out, 1)) return SS_ERR_ARG;
  const int nblocks = ss_div_up(m, 64);
  int tm, tn, min_per, splits;
  w8_linear_plan(m, k_in, n_out, &tm, &tn, &min_per, &splits);
  dim3 g((unsigned)(splits * tm * tn));
  SS_LAUNCH((k_wgrad8<false>), g, dim3(512), 0, stream, (const unsigned short*)x, (const unsigned short*)dy,
            (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, dweight,
            dbias, (int)m, k_in, n_out, 1, tn, nblocks, min_per, tm * tn, splits, 0);
  return SS_OK;
}
