// 256 x 256 bf16 MFMA GEMM with an LDS-DMA pipeline that stays in flight across barriers (gfx950).
//
//   C[m][n] = bias[n] + sum_k A[row(m)][k] * B[n][k]            (both operands K-contiguous)
//
// One kernel serves the two GEMM-shaped operators of the PTv3 hot path:
//   GATHER = true   submanifold conv forward / dgrad (spconv.SubMConv3d, ptv3:278-284): K = active taps x Cin,
//                   row(m) = the tap's neighbour of site rowperm[m] (zero row when missing), B = W[co][tap][ci]
//   GATHER = false  nn.Linear (ptv3 qkv / proj / fc1 / fc2, torch F.linear): row(m) = m
//   HM              (GATHER = false) the qkv projection in front of the head-major window attention (attention_hm.hip,
//                   ptv3:172-183): output row m = padded slot m of a curve order, its A row = rowperm[m] (the slot's
//                   point), and the epilogue writes out[section][head][slot][d] with section 0 (q) multiplied by
//                   hm_scale = softmax scale * log2(e) in fp32 BEFORE the one bf16 rounding
//
// Structure (512 threads = 8 waves as 2 (M) x 4 (N), one workgroup per CU, wave tile 128 x 64):
//   * LDS: two K-tile buffers (BK = 64) of four 16-KiB HALF-tiles each: A0 A1 B0 B1.  Half h of A holds, for
//     both wave rows, the 64 sites its waves read in one phase; half h of B the 32 channels of each wave column.
//   * a K-tile is computed in 4 phases (one C quadrant = 16 MFMA 16x16x32 each):
//         ph1 read B0 A0 | stage A1(t+1) | A0xB0        ph2 read B1 | stage B0(t+2) | A0xB1
//         ph3 read A1    | stage A0(t+2) | A1xB1        ph4          | stage B1(t+2) | A1xB0 , s_waitcnt vmcnt(6)
//     every phase = {ds_reads, 2 x global_load_lds_dwordx4 per thread} s_barrier {MFMAs} s_barrier.
//     The only VMEM wait in the loop is the counted vmcnt(6) of ph4: it retires tile t+1 completely and leaves
//     the three newest half-tiles (tile t+2) in flight across the barriers.
//   * the two wave rows run staggered by one barrier (row 1 executes one extra s_barrier up front, row 0 one at
//     the end), so on every SIMD one wave is in its MFMA section while the other reads LDS / issues DMA.
//   * hazards: a half is read >= 1 barrier after BOTH rows' vmcnt wait that retired its DMA (row 1 waits one
//     interval later -> tile t+1 is first read two barriers after row 0's wait); a half is re-staged only after
//     both rows retired their reads of it: A0/B1/A1 are re-staged two phases after the reading phase, B0 one
//     phase after -- its four reads are issued first in ph1 and retired by lgkmcnt(8) BEFORE ph1's barrier.
//   * LDS images are lane-linear (LDS-DMA writes base + lane*16); the XOR swizzle that makes ds_read_b128
//     conflict-free is applied to the per-lane SOURCE chunk and again on the read.
#include "common.h"
#include <stdlib.h>
#include "../../include/scenesplat_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 g8_bf8_t;
#define G8_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define G8_TAPS_MAX 27
#define G8_ZERO_ELEMS 4096

#ifndef G8_STAGGER
#define G8_STAGGER 1
#endif
#define G8_BUF 65536
#define G8_HALF 16384
#define G8_OFF_B 32768
#define G8_OFF_NBR 131072                                   // int32 [27][256]
#define G8_OFF_ROWID (G8_OFF_NBR + G8_TAPS_MAX * 256 * 4)   // int32 [256]
#define G8_OFF_MASK (G8_OFF_ROWID + 1024)
#define G8_LDS_BYTES (G8_OFF_MASK + 16)

__device__ uint4 g8_zero[G8_ZERO_ELEMS / 8];   // source of every missing-neighbour / out-of-range row piece

__device__ __forceinline__ void g8_glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ g8_bf8_t g8_lds(const char* p) {
  return __builtin_bit_cast(g8_bf8_t, *reinterpret_cast<const uint4*>(p));
}

template <bool GATHER, typename OutT, bool HM = false>
__global__ void __launch_bounds__(512)
k_gemm8(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, const float* __restrict__ bias,
        const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowperm, OutT* __restrict__ out, int M, int K,
        int N, int taps, int ntn, int hm_c = 0, int hm_d = 0, float hm_scale = 1.f, int walk = 0) {
  __shared__ __attribute__((aligned(16))) char smem[G8_LDS_BYTES];   // ONE object: tiles + rulebook slice
  int32_t* nbr_s = reinterpret_cast<int32_t*>(smem + G8_OFF_NBR);
  int32_t* rowid_s = reinterpret_cast<int32_t*>(smem + G8_OFF_ROWID);
  unsigned* mask_s = reinterpret_cast<unsigned*>(smem + G8_OFF_MASK);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  // XCD-aware, bijective: the workgroups an XCD runs are consecutive tiles (all N tiles of an M tile together)
  int L;
  {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = (L / ntn) * 256, n0 = (L % ntn) * 256;

  unsigned rem = 0u;       // taps still to do after the current one (GATHER)
  int tap = 0;
  if (GATHER) {
    if (tid < 256) {
      int k = m0 + tid;
      rowid_s[tid] = k < M ? (rowperm ? rowperm[k] : k) : -1;
    }
    if (tid == 0) *mask_s = 0u;
    __syncthreads();
    // the tile's rulebook slice: taps x 256 words, ALL loads in flight before the first use (a load + ballot per iteration made
    // this prologue 14 dependent round trips, 6-10 % of a workgroup's life at dec0).  walk != 0: nbr is the rulebook in WALK order
    // (nbr_walk[t][k] = nbr[t][rowperm[k]], ScenePlan.neighbors_walk): a wave reads 256 contiguous bytes instead of 64 cache lines
    constexpr int NIT = (G8_TAPS_MAX * 256 + 511) / 512;
    int vals[NIT];
    const int last = taps * 256 - 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = min(it * 512 + tid, last), tt = e >> 8, r = e & 255;
      if (walk) {
        const int k = m0 + r;
        const int v = nbr[(int64_t)tt * M + min(k, M - 1)];
        vals[it] = k < M ? v : -1;
      } else {
        const int row = rowid_s[r];
        const int v = nbr[(int64_t)tt * M + max(row, 0)];
        vals[it] = row >= 0 ? v : -1;
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = it * 512 + tid;              // e >> 8 is wave-uniform
      if (e <= last) {
        nbr_s[e] = vals[it];
        if (__ballot(vals[it] >= 0) != 0ULL && lane == 0) atomicOr(mask_s, 1u << (e >> 8));
      }
    }
    __syncthreads();
    rem = __builtin_amdgcn_readfirstlane(*mask_s);
  }
  const int ksteps = K >> 6;
  int T;                                        // K-tiles of this workgroup
  if (GATHER) {
    T = __builtin_popcount(rem) * ksteps;
    if (rem) { tap = __builtin_ctz(rem); rem &= rem - 1u; }
  } else {
    T = ksteps;
  }

  f32x4_t acc[2][4][2][2];                      // [A half][mi][B half][ni]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[a][mi][b][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- staging state: thread (tid) fills LDS row j*64 + (tid>>3), 16-byte slot tid&7, of each half ----
  const int lc8 = (((tid & 7) ^ ((tid >> 4) & 7)) << 3);          // logical chunk (elements) held by that slot
  const int srow = tid >> 3;                                       // 0..63
  const unsigned short* pA[4];                                     // row q = 2j + h  <->  tile row 128j + 64h + srow
  const unsigned short* pB[2][2];                                  // [h][j] <-> tile col 128j + 64(tid>>8) + 32h + (srow&31)
  auto load_rows = [&](int tp) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int R = 128 * (q >> 1) + 64 * (q & 1) + srow;
      if (GATHER) {
        int src = nbr_s[tp * 256 + R];
        pA[q] = (src >= 0 ? A + (int64_t)src * K : reinterpret_cast<const unsigned short*>(g8_zero)) + lc8;
      } else {
        int row = min(m0 + R, M - 1);
        if (HM) row = rowperm[row];
        pA[q] = A + (int64_t)row * K + lc8;
      }
    }
  };
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = min(n0 + 128 * j + 64 * (tid >> 8) + 32 * h + (srow & 31), N - 1);
      pB[h][j] = W + (int64_t)col * taps * K + lc8;
    }
  int kc = 0;                                    // channel offset of the tile being staged
  int64_t boff = (int64_t)tap * K;               // tap offset in W rows
  load_rows(tap);
  int staged = 0;                                // index of the tile the state describes
  auto advance = [&]() {
    if (staged + 1 >= T) return;                 // past the end: keep re-staging the last tile (harmless, keeps vmcnt uniform)
    ++staged;
    kc += 64;
    if (GATHER) {
      // K walk of the conv: the tile's active taps outermost, a tap's whole rows (64-channel K-tiles) innermost.  (Walking channel
      // chunks outermost, so that a row piece wanted by up to nine (site, tap) pairs stays L2-resident, was built in round 2 and is
      // monotonically slower the smaller the chunk: long contiguous row reads matter more than the re-use -- DESIGN.md section 4.)
      if (kc == K) {
        kc = 0;
        tap = __builtin_ctz(rem); rem &= rem - 1u;
        boff = (int64_t)tap * K;
        load_rows(tap);
      }
    } else if (kc == K) {
      kc = 0;
    }
  };
  char* const dst0 = smem + wave * 1024;
  auto stageA = [&](int buf, int h) {
    char* d = dst0 + buf * G8_BUF + h * G8_HALF;
    g8_glds16(pA[h] + kc, d);
    g8_glds16(pA[2 + h] + kc, d + 8192);
  };
  auto stageB = [&](int buf, int h) {
    char* d = dst0 + buf * G8_BUF + G8_OFF_B + h * G8_HALF;
    g8_glds16(pB[h][0] + boff + kc, d);
    g8_glds16(pB[h][1] + boff + kc, d + 8192);
  };

  // ---- fragment read addresses: row = wave offset + 16*i + lq, logical chunk 4*ks + g, swizzle (row>>1)&7 = (lq>>1)&7 ----
  const int sw = (lq >> 1) & 7;
  const int oA = (wr * 64 + lq) * 128 + ((g ^ sw) << 4);          // ks = 0; ks = 1 is ^ 64
  const int oB = G8_OFF_B + (wc * 32 + lq) * 128 + ((g ^ sw) << 4);
  g8_bf8_t af[4][2], b0f[2][2], b1f[2][2];
  auto readA = [&](int buf, int h) {
    const char* base = smem + buf * G8_BUF + h * G8_HALF;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      af[mi][0] = g8_lds(base + oA + mi * 2048);
      af[mi][1] = g8_lds(base + (oA ^ 64) + mi * 2048);
    }
  };
  auto readB = [&](int buf, int h, g8_bf8_t (&bf)[2][2]) {
    const char* base = smem + buf * G8_BUF + h * G8_HALF;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      bf[ni][0] = g8_lds(base + oB + ni * 2048);
      bf[ni][1] = g8_lds(base + (oB ^ 64) + ni * 2048);
    }
  };
#define G8_MM(HA, HB, BF)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                \
      acc[HA][mi][HB][ni] = G8_MFMA(BF[ni][ks], af[mi][ks], acc[HA][mi][HB][ni]);   /* C^T: rows = channels */
#define G8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G8_COMPUTE_BEGIN() do { G8_BAR(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); } while (0)
#define G8_COMPUTE_END() do { __builtin_amdgcn_s_setprio(0); G8_BAR(); } while (0)

  if (T > 0) {
    // prologue: tile 0 complete + B0 A0 B1 of tile 1
    stageB(0, 0); stageA(0, 0); stageB(0, 1); stageA(0, 1);
    advance();
    stageB(1, 0); stageA(1, 0); stageB(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G8_BAR();
#if G8_STAGGER
    if (wr == 1) G8_BAR();                       // wave row 1 runs one barrier behind: its reads overlap row 0's MFMAs
#endif
    for (int t = 0; t < T; ++t) {
      const int buf = t & 1;
      // ph1
      readB(buf, 0, b0f);
      __builtin_amdgcn_sched_barrier(0);
      readA(buf, 0);
      stageA(buf ^ 1, 1);                        // A1 of tile t+1 (state = t+1)
#if G8_STAGGER
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // B0 reads retired before the barrier: B0 is re-staged next phase
#endif
      G8_COMPUTE_BEGIN();
      G8_MM(0, 0, b0f)
      G8_COMPUTE_END();
      // ph2
      advance();                                 // state -> tile t+2
      readB(buf, 1, b1f);
      stageB(buf, 0);
      G8_COMPUTE_BEGIN();
      G8_MM(0, 1, b1f)
      G8_COMPUTE_END();
      // ph3
      readA(buf, 1);
      stageA(buf, 0);
      G8_COMPUTE_BEGIN();
      G8_MM(1, 1, b1f)
      G8_COMPUTE_END();
      // ph4
      stageB(buf, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      G8_COMPUTE_BEGIN();
      G8_MM(1, 0, b0f)
      G8_COMPUTE_END();
    }
#if G8_STAGGER
    if (wr == 0) G8_BAR();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // ---- epilogue: acc[ha][mi][hb][ni][r] = C[site 128wr + 64ha + 16mi + lq][channel 64wc + 32hb + 16ni + 4g + r] ----
  // HM: the (section, head, d) split of a lane's four column groups is fixed for the whole tile: computed once (the divisions
  // by the runtime head width cost ~40 instructions each)
  int64_t hm_base[2][2];
  float hm_mul[2][2];
  if (HM) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = min(n0 + 64 * wc + 32 * hb + 16 * ni + 4 * g, N - 4);
        const int sec = col / hm_c, cc = col - sec * hm_c, hd = cc / hm_d, dd = cc - hd * hm_d;
        hm_base[hb][ni] = ((int64_t)sec * (hm_c / hm_d) + hd) * M * hm_d + dd;
        hm_mul[hb][ni] = sec == 0 ? hm_scale : 1.f;
      }
  }
#pragma unroll
  for (int ha = 0; ha < 2; ++ha)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int R = 128 * wr + 64 * ha + 16 * mi + lq;
      int64_t row;
      if (GATHER) { row = rowid_s[R]; } else { row = m0 + R < M ? m0 + R : -1; }
      if (row < 0) continue;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int col = n0 + 64 * wc + 32 * hb + 16 * ni + 4 * g;
          if (col >= N) continue;                // N is a multiple of 4 (checked on the host)
          f32x4_t v = acc[ha][mi][hb][ni];
          if (bias) { float4 bv = *reinterpret_cast<const float4*>(bias + col); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
          OutT* op = out + row * N + col;
          if (HM) {       // (section, head, slot, d): 4 consecutive channels never straddle a head (hm_d % 4 == 0)
            const float m_ = hm_mul[hb][ni];
            v[0] *= m_; v[1] *= m_; v[2] *= m_; v[3] *= m_;
            op = out + hm_base[hb][ni] + row * hm_d;
          }
          if (sizeof(OutT) == 2) {
            uint2 u; u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(op) = u;
          } else {
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
    }
}

// eligibility of the pipeline kernel (callers fall back to k_subm_gemm / hipBLASLt otherwise)
extern "C" int ss_gemm8_ok(int64_t m, int k, int n, int taps) {
  return m > 0 && m < (1LL << 31) && k >= 64 && (k & 63) == 0 && k <= G8_ZERO_ELEMS && n >= 4 && (n & 3) == 0 &&
         taps >= 1 && taps <= G8_TAPS_MAX;
}

template <bool GATHER>
static int g8_launch(const void* a, const void* w, const float* bias, const int32_t* nbr, const int32_t* rowperm, void* out,
                     int64_t m, int k, int n, int taps, int out_dtype, hipStream_t stream, int walk = 0) {
  if (!ss_gemm8_ok(m, k, n, taps)) return SS_ERR_ARG;
  const int ntm = ss_div_up(m, 256), ntn = ss_div_up(n, 256);
  dim3 grid(ntm * ntn), block(512);
  const unsigned short* A = (const unsigned short*)a; const unsigned short* Wp = (const unsigned short*)w;
  if (out_dtype == SS_BF16)
    SS_LAUNCH((k_gemm8<GATHER, unsigned short>), grid, block, 0, stream, A, Wp, bias, nbr, rowperm, (unsigned short*)out, (int)m, k, n, taps, ntn, 0, 0, 1.f, walk);
  else if (out_dtype == SS_F32)
    SS_LAUNCH((k_gemm8<GATHER, float>), grid, block, 0, stream, A, Wp, bias, nbr, rowperm, (float*)out, (int)m, k, n, taps, ntn, 0, 0, 1.f, walk);
  else
    return SS_ERR_ARG;
  return SS_OK;
}

extern "C" int ss_subm_conv_fwd_pipe(const void* in, const void* weight, const float* bias, const int32_t* nbr,
                                     const int32_t* rowperm, void* out, int64_t n, int cin, int cout, int taps,
                                     int out_dtype, hipStream_t stream) {
  if (n == 0) return SS_OK;
  return g8_launch<true>(in, weight, bias, nbr, rowperm, out, n, cin, cout, taps, out_dtype, stream);
}

// the same kernel reading the rulebook in WALK order: nbr_walk[t][k] = nbr[t][rowperm[k]] (rowperm NULL: nbr itself)
extern "C" int ss_subm_conv_fwd_pipe_walk(const void* in, const void* weight, const float* bias, const int32_t* nbr_walk,
                                          const int32_t* rowperm, void* out, int64_t n, int cin, int cout, int taps,
                                          int out_dtype, hipStream_t stream) {
  if (n == 0) return SS_OK;
  return g8_launch<true>(in, weight, bias, nbr_walk, rowperm, out, n, cin, cout, taps, out_dtype, stream, 1);
}

extern "C" int ss_linear_fwd(const void* x, const void* weight, const float* bias, void* out, int64_t m, int k, int n,
                             int out_dtype, hipStream_t stream) {
  if (m == 0) return SS_OK;
  return g8_launch<false>(x, weight, bias, nullptr, nullptr, out, m, k, n, 1, out_dtype, stream);
}

// qkv projection with the head-major, window-ordered epilogue (see the header): x (n, k) bf16, row_index (m) = the point of
// padded slot p, weight (n_out, k) bf16 with n_out = sections * channels, out (sections, channels / head_dim, m, head_dim) bf16
extern "C" int ss_linear_fwd_headmajor(const void* x, const int32_t* row_index, const void* weight, const float* bias,
                                       void* out_hm, int64_t m, int k, int n_out, int channels, int head_dim,
                                       float sec0_scale, hipStream_t stream) {
  if (m == 0) return SS_OK;
  if (!ss_gemm8_ok(m, k, n_out, 1) || channels <= 0 || head_dim <= 0 || (head_dim & 3) || channels % head_dim || n_out % channels ||
      !row_index)
    return SS_ERR_ARG;
  const int ntm = ss_div_up(m, 256), ntn = ss_div_up(n_out, 256);
  dim3 grid(ntm * ntn), block(512);
  SS_LAUNCH((k_gemm8<false, unsigned short, true>), grid, block, 0, stream, (const unsigned short*)x, (const unsigned short*)weight,
            bias, (const int32_t*)nullptr, row_index, (unsigned short*)out_hm, (int)m, k, n_out, 1, ntn, channels, head_dim,
            sec0_scale);
  return SS_OK;
}
