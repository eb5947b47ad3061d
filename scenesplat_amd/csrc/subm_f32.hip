// Submanifold convolution of the NARROW first stage in exact fp32 on the matrix cores (gfx950), round 3.
//
// The reference keeps spconv.SubMConv3d in fp32 under AMP (pointcept/models/modules.py:64-75).  Rounding the conv operands of the
// 32-channel first stage (stem k = 5, 11 -> 32; two enc0 blocks k = 3, 32 -> 32) to bf16 costs the per-Gaussian cosine budget
// (it propagates through all 22 blocks), so rounds 1-2 ran that stage on hi/lo-split bf16 operands
// ("bf16x3": 3x the products on 128-column tiles for 32 output channels, 1.6 ms per step, bound by the neighbour gather).
// Here: v_mfma_f32_32x32x2_f32 -- fp32 in, fp32 accumulate, bit-for-bit a k-ordered fmaf chain -- on 32-site x 32-channel tiles.
// The stage is 11.4 GFLOP per pass: at the fp32 matrix rate (157 TFLOP/s, 1/16 of bf16) that is 12-16 us of pipe time per conv
// over the chip; what matters is the gather, so the kernels are built for occupancy (one 32-site tile per WAVE, no workgroup
// barrier, 3 waves per SIMD), not for tile reuse.  Measured (scripts/f32conv_probe.py, 102,400 sites): 32 -> 32, k = 3 forward
// 46 us, weight gradient 54-65 us; stem 11 -> 32, k = 5 forward 62 us, weight gradient 200 us.
//
//   forward / dgrad   out[s][co] = bias[co] + sum_tap sum_ci in[nbr[tap][s]][ci] * W[co][tap][ci]     (dgrad: mirrored weights)
//   weight gradient   dW[co][tap][ci] += sum_s g[s][co] * in[nbr[tap][s]][ci]       (fp32 atomics, one flush per (tap, share))
//
// Operand maps of v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; a 16-byte LDS /
// global read gives a lane four consecutive channels, which serve four MFMA steps when the k order is permuted identically on both
// sides (step (q, e), lane half kk  <->  channel 8 q + 4 kk + e).
#include "common.h"
#include "../../include/scenesplat_hip.h"

#define F32MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SF_TG 32          // taps per group held in the wave's LDS table

// row chunk of a neighbour that may be missing (index < 0 -> zeros): an UNCONDITIONAL load of a clamped row + selects, so that a
// tap's gathers are issued back to back instead of one exec-masked branch per load
__device__ __forceinline__ float4 sf_row_or_zero(const float* __restrict__ base, int row, int width, int col) {
  float4 v = *reinterpret_cast<const float4*>(base + (int64_t)max(row, 0) * width + col);
  const bool ok = row >= 0;
  v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Rulebook in WALK order: nbr_walk[tap][k] = neighbour (row of `in`) of the k-th site of the walk, i.e. nbr[tap][rowperm[k]] -- a
// wave's 32 sites read 128 contiguous bytes per tap instead of 32 cache lines (measured: the scattered 4-byte rulebook reads, not the
// row gathers, were what bound the first version: 1.65 GB of line traffic for the 125-tap stem).  rowperm == NULL: nbr itself.
//
// forward (and dgrad with mirrored weights).  wq: weights re-laid as [tap][CI / 8][2][32 co][4] fp32 (one coalesced 1-KiB read per
// (tap, q) and wave); in (n, CI) fp32 with CI = 16 or 32 (zero-padded); out (n, 32) fp32.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int CI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_subm_f32_fwd(const float* __restrict__ in, const float* __restrict__ wq, const float* __restrict__ bias,
               const int32_t* __restrict__ nbr_walk, const int32_t* __restrict__ rowperm, float* __restrict__ out, int n, int taps,
               int winner_tap) {
  constexpr int NCH = CI / 4;                  // 16-byte chunks per input row
  constexpr int NQ = CI / 8;                   // k groups of 8 channels
  constexpr int NA = (32 * NCH) / 64;          // staging loads per lane and tap
  constexpr int PL = 32 * 16 + 16;             // chunk plane [chunk][site][16 B], +16: the 8 chunk lanes of a row hit 8 different banks
  __shared__ __attribute__((aligned(16))) char smem[4 * (NCH * PL + SF_TG * 32 * 4)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, kk = lane >> 5;
  char* const aimg = smem + wave * (NCH * PL + SF_TG * 32 * 4);
  int32_t* const nb_s = reinterpret_cast<int32_t*>(aimg + NCH * PL);              // [tap in group][site]
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  if (m0 >= n) return;                                                              // wave-uniform
  const int site = (m0 + i < n) ? (rowperm ? rowperm[m0 + i] : m0 + i) : -1;
  // duplicate voxels, adjoint pass (winner_tap = the centre tap, else -1): only the WINNER row of a voxel is ever read by the forward,
  // so only winners receive a gradient; the centre tap of the rulebook names each site's winner
  const int keep = (winner_tap < 0 || site < 0) ? 1 : (nbr_walk[(int64_t)winner_tap * n + m0 + i] == site ? 1 : 0);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int tg = 0; tg < taps; tg += SF_TG) {
    const int nt = min(SF_TG, taps - tg);
    // the group's rulebook slice of this wave's 32 sites (all loads in flight at once: the two lane halves take alternate taps)
    // + the mask of taps that reach at least one neighbour
    int nbv[SF_TG / 2];
#pragma unroll
    for (int u = 0; u < SF_TG / 2; ++u) {
      const int tt = 2 * u + kk;
      const int v = nbr_walk[(int64_t)(tg + min(tt, nt - 1)) * n + min(m0 + i, n - 1)];
      nbv[u] = (tt < nt && site >= 0) ? v : -1;
    }
    unsigned mask = 0u;
#pragma unroll
    for (int u = 0; u < SF_TG / 2; ++u) {
      nb_s[(2 * u + kk) * 32 + i] = nbv[u];
      const unsigned long long b = __ballot(nbv[u] >= 0);
      mask |= ((unsigned)(b & 0xffffffffULL) != 0u ? 1u : 0u) << (2 * u);
      mask |= ((unsigned)(b >> 32) != 0u ? 1u : 0u) << (2 * u + 1);
    }
    mask = __builtin_amdgcn_readfirstlane(mask);
    // per active tap: neighbour rows -> registers (chunk e = r * 64 + lane -> site e / NCH, chunk e % NCH; zeros where the pair
    // is missing) and the tap's weights; the NEXT tap's loads are issued before this tap's LDS write + MFMAs
    auto issue = [&](int t, float4 (&ra)[NA], float4 (&rb)[NQ]) {
#pragma unroll
      for (int r = 0; r < NA; ++r) {
        const int e = r * 64 + lane, si = e / NCH, c = e - si * NCH;
        const int nb = nb_s[t * 32 + si];
        ra[r] = sf_row_or_zero(in, nb, CI, c * 4);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) rb[q] = *reinterpret_cast<const float4*>(wq + ((((int64_t)(tg + t) * NQ + q) * 2 + kk) * 32 + i) * 4);
    };
    auto consume = [&](const float4 (&ra)[NA], const float4 (&rb)[NQ]) {
#pragma unroll
      for (int r = 0; r < NA; ++r) {
        const int e = r * 64 + lane, si = e / NCH, c = e - si * NCH;
        *reinterpret_cast<float4*>(aimg + c * PL + si * 16) = ra[r];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(aimg + (2 * q + kk) * PL + i * 16);
        acc = F32MFMA(a.x, rb[q].x, acc); acc = F32MFMA(a.y, rb[q].y, acc);
        acc = F32MFMA(a.z, rb[q].z, acc); acc = F32MFMA(a.w, rb[q].w, acc);
      }
    };
    if (mask) {
      float4 ra0[NA], rb0[NQ], ra1[NA], rb1[NQ];
      int t = __builtin_ctz(mask); mask &= mask - 1u;
      issue(t, ra0, rb0);
      while (true) {
        if (!mask) { consume(ra0, rb0); break; }
        t = __builtin_ctz(mask); mask &= mask - 1u;
        issue(t, ra1, rb1); consume(ra0, rb0);
        if (!mask) { consume(ra1, rb1); break; }
        t = __builtin_ctz(mask); mask &= mask - 1u;
        issue(t, ra0, rb0); consume(ra1, rb1);
      }
    }
  }
  // D: column (lane & 31) = output channel, register r of half kk = site row (r & 3) + 8 (r >> 2) + 4 kk
  const float bj = bias ? bias[i] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
    const int s = __shfl(site, row, 64);
    const int kp = __shfl(keep, row, 64);
    if (s >= 0) out[(int64_t)s * 32 + i] = kp ? acc[r] + bj : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient: one wave per (tap, share of that tap's ACTIVE 64-site blocks -- ss_subm_block_lists, rowperm order), dW in the
// checkpoint layout (32 co, taps, CI) fp32, zero on entry.  A = g^T (co on the lane, site = k), B = gathered input rows.
// ---------------------------------------------------------------------------------------------------------------------------------
#define SF_WPER 4          // active 64-site blocks per wave (8 sub-blocks of 32 sites); many short waves hide the dependent gathers

template <int CI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_subm_f32_wgrad(const float* __restrict__ in, const float* __restrict__ g, const int32_t* __restrict__ nbr_walk,
                 const int32_t* __restrict__ rowperm, const int32_t* __restrict__ blk_count, const int32_t* __restrict__ blk_list,
                 float* __restrict__ dW, int n, int taps, int cin, int nblocks) {
  constexpr int NCH = CI / 4;
  constexpr int NX = (32 * NCH) / 64;                                   // input-row loads per lane and sub-block
  constexpr int WAVE_LDS = 2 * SF_WPER * 64 * 4 + 32 * 128 + 32 * CI * 4;
  __shared__ __attribute__((aligned(16))) char smem[4 * WAVE_LDS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kk = lane >> 5;
  int32_t* const site_s = reinterpret_cast<int32_t*>(smem + wave * WAVE_LDS);       // [SF_WPER * 64] walk-order sites of the share
  int32_t* const nb_s = site_s + SF_WPER * 64;                                        // their neighbour under this tap
  char* const gimg = reinterpret_cast<char*>(nb_s + SF_WPER * 64);                    // [32 sites][32 co] fp32
  char* const ximg = gimg + 32 * 128;                                                 // [32 sites][CI] fp32
  const int tap = blockIdx.y, share = blockIdx.x * 4 + wave;
  const int cnt = blk_count[tap];
  const int beg = share * SF_WPER, end = min(cnt, beg + SF_WPER);
  if (beg >= end) return;                                               // wave-uniform
  const int32_t* list = blk_list + (int64_t)tap * nblocks;
  // indices of the whole share first (two dependent rounds for all SF_WPER blocks at once)
#pragma unroll
  for (int r = 0; r < SF_WPER; ++r) {
    const int b = beg + r;
    int site = -1, nb = -1;
    if (b < end) {
      const int k = list[b] * 64 + lane;
      if (k < n) { site = rowperm ? rowperm[k] : k; nb = nbr_walk[(int64_t)tap * n + k]; }
    }
    site_s[r * 64 + lane] = nb >= 0 ? site : -1;                        // a site without the pair contributes nothing
    nb_s[r * 64 + lane] = nb;
  }
  const int nsub = (end - beg) * 2;
  auto issue = [&](int sb, float4 (&rg)[4], float4 (&rx)[NX]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = r * 64 + lane, si = e >> 3, c = e & 7;
      const int s2 = site_s[sb * 32 + si];
      rg[r] = sf_row_or_zero(g, s2, 32, c * 4);
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = r * 64 + lane, si = e / NCH, c = e - si * NCH;
      const int n2 = nb_s[sb * 32 + si];
      rx[r] = sf_row_or_zero(in, n2, CI, c * 4);
    }
  };
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto consume = [&](const float4 (&rg)[4], const float4 (&rx)[NX]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = r * 64 + lane;
      *reinterpret_cast<float4*>(gimg + (e >> 3) * 128 + (e & 7) * 16) = rg[r];
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = r * 64 + lane, si = e / NCH, c = e - si * NCH;
      *reinterpret_cast<float4*>(ximg + si * (CI * 4) + c * 16) = rx[r];
    }
    // dW^T tile: D[co][ci] += sum over the 32 sites; MFMA step u covers the site pair (2 u, 2 u + 1), lane half kk takes 2 u + kk
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float a = *reinterpret_cast<const float*>(gimg + (2 * u + kk) * 128 + j * 4);
      const float x = *reinterpret_cast<const float*>(ximg + (2 * u + kk) * (CI * 4) + (j & (CI - 1)) * 4);
      acc = F32MFMA(a, x, acc);
    }
  };
  {
    float4 rg0[4], rx0[NX], rg1[4], rx1[NX];
    int sb = 0;
    issue(0, rg0, rx0);
    while (true) {
      if (++sb >= nsub) { consume(rg0, rx0); break; }
      issue(sb, rg1, rx1); consume(rg0, rx0);
      if (++sb >= nsub) { consume(rg1, rx1); break; }
      issue(sb, rg0, rx0); consume(rg1, rx1);
    }
  }
  // D: column (lane & 31) = input channel ci, register r of half kk = output channel (r & 3) + 8 (r >> 2) + 4 kk
  if (j < cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * kk;
      atomicAdd(dW + ((int64_t)co * taps + tap) * cin + j, acc[r]);
    }
  }
}

extern "C" int ss_subm_f32_ok(int cin_padded, int cout) { return cout == 32 && (cin_padded == 16 || cin_padded == 32); }

static int subm_f32_launch(const float* in, const float* wq, const float* bias, const int32_t* nbr_walk, const int32_t* rowperm,
                           float* out, int64_t n, int cin_padded, int cout, int taps, int winner_tap, hipStream_t stream) {
  if (n < 0 || n >= (1LL << 31) || taps <= 0 || !ss_subm_f32_ok(cin_padded, cout)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  dim3 g(ss_div_up(n, 128)), b(256);
  if (cin_padded == 16) SS_LAUNCH((k_subm_f32_fwd<16>), g, b, 0, stream, in, wq, bias, nbr_walk, rowperm, out, (int)n, taps, winner_tap);
  else SS_LAUNCH((k_subm_f32_fwd<32>), g, b, 0, stream, in, wq, bias, nbr_walk, rowperm, out, (int)n, taps, winner_tap);
  return SS_OK;
}

extern "C" int ss_subm_f32_fwd(const float* in, const float* wq, const float* bias, const int32_t* nbr_walk, const int32_t* rowperm,
                               float* out, int64_t n, int cin_padded, int cout, int taps, hipStream_t stream) {
  return subm_f32_launch(in, wq, bias, nbr_walk, rowperm, out, n, cin_padded, cout, taps, -1, stream);
}

// dgrad on a level with duplicate voxels: dout_folded = ss_dup_fold_rows(dout) (gradients of a voxel's sites summed onto its winner),
// wq = the tap-mirrored transposed weights; rows that are not the winner of their voxel come out as zeros (taps must be odd: the
// centre tap of the rulebook is each site's winner)
extern "C" int ss_subm_f32_dgrad_dup(const float* dout_folded, const float* wq, const int32_t* nbr_walk, const int32_t* rowperm,
                                     float* din, int64_t n, int cin_padded, int cout, int taps, hipStream_t stream) {
  if (!(taps & 1)) return SS_ERR_ARG;
  return subm_f32_launch(dout_folded, wq, nullptr, nbr_walk, rowperm, din, n, cin_padded, cout, taps, taps / 2, stream);
}

// dweight (32, taps, cin) fp32 ZEROED by the caller; blk_count / blk_list from ss_subm_block_lists (same rowperm)
extern "C" int ss_subm_f32_wgrad(const float* in, const float* gout, const int32_t* nbr_walk, const int32_t* rowperm,
                                 const int32_t* blk_count, const int32_t* blk_list, float* dweight, int64_t n, int cin_padded,
                                 int cin, int cout, int taps, hipStream_t stream) {
  if (n < 0 || n >= (1LL << 31) || taps <= 0 || !ss_subm_f32_ok(cin_padded, cout) || cin <= 0 || cin > cin_padded) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  const int nblocks = ss_div_up(n, 64);
  dim3 g(ss_div_up(ss_div_up(nblocks, SF_WPER), 4), taps), b(256);
  if (cin_padded == 16)
    SS_LAUNCH((k_subm_f32_wgrad<16>), g, b, 0, stream, in, gout, nbr_walk, rowperm, blk_count, blk_list, dweight, (int)n, taps, cin, nblocks);
  else
    SS_LAUNCH((k_subm_f32_wgrad<32>), g, b, 0, stream, in, gout, nbr_walk, rowperm, blk_count, blk_list, dweight, (int)n, taps, cin, nblocks);
  return SS_OK;
}
