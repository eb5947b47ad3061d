// Fused residual-add (+ DropPath row scale) + LayerNorm, forward and backward (gfx950, HBM-bound).
// One pass per pre-norm seam of a PTv3 Block (ptv3:318-338) instead of PyTorch's separate
// add / cast / layer_norm / layer_norm-backward / cast kernels:
//     v     = x + rowscale * y          (y, rowscale optional)
//     xout  = v                         (fp32 residual stream; optional extra bf16 copy for the next conv)
//     h     = LN(v) * gamma + beta      (optional; written in bf16 for the following GEMM, or fp32)
// backward:  g_v = g_xout + LN'(g_h);  g_x = g_v;  g_y = rowscale * g_v;  dgamma/dbeta as per-block
// partial sums (reduced by the caller), fp32 statistics throughout.
// One wave per row, 4 elements per lane per step (16-byte fp32 / 8-byte bf16 accesses, fully
// coalesced 1 KiB / 512 B per wave instruction); the row lives in registers between the two passes.
#include "common.h"
#include "../../include/scenesplat_hip.h"

#define LN_THREADS 256
#define LN_MAXIT 4   // C <= 1024

__device__ __forceinline__ float4 ln_ld4(const void* p, int dtype, int64_t idx) {
  if (dtype == SS_F32) return *reinterpret_cast<const float4*>((const float*)p + idx);
  uint2 u = *reinterpret_cast<const uint2*>((const unsigned short*)p + idx);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void ln_st4(void* p, int dtype, int64_t idx, float4 v) {
  if (dtype == SS_F32) { *reinterpret_cast<float4*>((float*)p + idx) = v; return; }
  uint2 u; u.x = pack_bf16x2(v.x, v.y); u.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>((unsigned short*)p + idx) = u;
}

template <int IT>
__global__ void __launch_bounds__(LN_THREADS)
k_add_ln_fwd(const void* __restrict__ x, int x_dt, const void* __restrict__ y, int y_dt, const float* __restrict__ rowscale,
             const float* __restrict__ gamma, const float* __restrict__ beta, float eps, void* __restrict__ xout, int xout_dt,
             void* __restrict__ xcopy, void* __restrict__ h, int h_dt, float* __restrict__ mean, float* __restrict__ rstd,
             int64_t n, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  const float rs = rowscale ? rowscale[row] : 1.f;
  float4 v[IT];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < C) {
      float4 a = ln_ld4(x, x_dt, row * C + j);
      if (y) { float4 b = ln_ld4(y, y_dt, row * C + j); a.x += rs * b.x; a.y += rs * b.y; a.z += rs * b.z; a.w += rs * b.w; }
      v[i] = a;
      if (xout) ln_st4(xout, xout_dt, row * C + j, a);
      if (xcopy) ln_st4(xcopy, SS_BF16, row * C + j, a);
      sum += a.x + a.y + a.z + a.w;
    }
  }
  if (!gamma) return;
  const float mu = wave_reduce_sum(sum) / C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) {
      float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
      sq += dx * dx + dy * dy + dz * dz + dw * dw;
    }
  }
  const float r = rsqrtf(wave_reduce_sum(sq) / C + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = r; }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) {
      float4 g = *reinterpret_cast<const float4*>(gamma + j), b = *reinterpret_cast<const float4*>(beta + j);
      float4 o = make_float4((v[i].x - mu) * r * g.x + b.x, (v[i].y - mu) * r * g.y + b.y, (v[i].z - mu) * r * g.z + b.z,
                             (v[i].w - mu) * r * g.w + b.w);
      ln_st4(h, h_dt, row * C + j, o);
    }
  }
}

template <int IT>
__global__ void __launch_bounds__(LN_THREADS)
k_add_ln_bwd(const void* __restrict__ g_xout, int gxo_dt, const void* __restrict__ g_xcopy, int gxc_dt,
             const void* __restrict__ g_h, int gh_dt,
             const void* __restrict__ v_in, int v_dt, const float* __restrict__ mean, const float* __restrict__ rstd,
             const float* __restrict__ gamma, const float* __restrict__ rowscale, void* __restrict__ g_x, int gx_dt,
             void* __restrict__ g_y, int gy_dt, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, int64_t n,
             int C) {
  __shared__ float red[2][LN_THREADS / 64][LN_MAXIT * 256 / 64][64];   // [gamma|beta][wave][i*4+e][lane] -- only IT slices used
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int waves_total = gridDim.x * (LN_THREADS / 64);
  const bool ln = g_h != nullptr;
  float4 dg[IT], db[IT], gm[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i]; gm[i] = dg[i];
    int j = i * 256 + lane * 4;
    if (ln && j < C) gm[i] = *reinterpret_cast<const float4*>(gamma + j);
  }
  for (int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + wave; row < n; row += waves_total) {
    float4 gv[IT];
    if (ln) {
      const float mu = mean[row], r = rstd[row];
      float4 xh[IT], gy[IT];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int j = i * 256 + lane * 4;
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); gy[i] = xh[i];
        if (j < C) {
          float4 a = ln_ld4(v_in, v_dt, row * C + j), g = ln_ld4(g_h, gh_dt, row * C + j);
          xh[i] = make_float4((a.x - mu) * r, (a.y - mu) * r, (a.z - mu) * r, (a.w - mu) * r);
          gy[i] = make_float4(g.x * gm[i].x, g.y * gm[i].y, g.z * gm[i].z, g.w * gm[i].w);
          c1 += gy[i].x + gy[i].y + gy[i].z + gy[i].w;
          c2 += gy[i].x * xh[i].x + gy[i].y * xh[i].y + gy[i].z * xh[i].z + gy[i].w * xh[i].w;
          dg[i].x += g.x * xh[i].x; dg[i].y += g.y * xh[i].y; dg[i].z += g.z * xh[i].z; dg[i].w += g.w * xh[i].w;
          db[i].x += g.x; db[i].y += g.y; db[i].z += g.z; db[i].w += g.w;
        }
      }
      c1 = wave_reduce_sum(c1) / C; c2 = wave_reduce_sum(c2) / C;
#pragma unroll
      for (int i = 0; i < IT; ++i)
        gv[i] = make_float4(r * (gy[i].x - c1 - xh[i].x * c2), r * (gy[i].y - c1 - xh[i].y * c2),
                            r * (gy[i].z - c1 - xh[i].z * c2), r * (gy[i].w - c1 - xh[i].w * c2));
    } else {
#pragma unroll
      for (int i = 0; i < IT; ++i) gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float rs = rowscale ? rowscale[row] : 1.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      if (j < C) {
        if (g_xout) { float4 e = ln_ld4(g_xout, gxo_dt, row * C + j); gv[i].x += e.x; gv[i].y += e.y; gv[i].z += e.z; gv[i].w += e.w; }
        if (g_xcopy) { float4 e = ln_ld4(g_xcopy, gxc_dt, row * C + j); gv[i].x += e.x; gv[i].y += e.y; gv[i].z += e.z; gv[i].w += e.w; }
        if (g_x) ln_st4(g_x, gx_dt, row * C + j, gv[i]);
        if (g_y) ln_st4(g_y, gy_dt, row * C + j, make_float4(rs * gv[i].x, rs * gv[i].y, rs * gv[i].z, rs * gv[i].w));
      }
    }
  }
  if (!ln) return;
  // block reduction of the per-wave column partials, then one row of partials per block
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    red[0][wave][i * 4 + 0][lane] = dg[i].x; red[0][wave][i * 4 + 1][lane] = dg[i].y;
    red[0][wave][i * 4 + 2][lane] = dg[i].z; red[0][wave][i * 4 + 3][lane] = dg[i].w;
    red[1][wave][i * 4 + 0][lane] = db[i].x; red[1][wave][i * 4 + 1][lane] = db[i].y;
    red[1][wave][i * 4 + 2][lane] = db[i].z; red[1][wave][i * 4 + 3][lane] = db[i].w;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int j = i * 256 + lane * 4 + e;
        if (j < C) {
          float a = 0.f, b = 0.f;
          for (int w = 0; w < LN_THREADS / 64; ++w) { a += red[0][w][i * 4 + e][lane]; b += red[1][w][i * 4 + e][lane]; }
          dgamma_part[(int64_t)blockIdx.x * C + j] = a;
          dbeta_part[(int64_t)blockIdx.x * C + j] = b;
        }
      }
  }
}

extern "C" int ss_add_layernorm_fwd(const void* x, int x_dtype, const void* y, int y_dtype, const float* rowscale,
                                    const float* gamma, const float* beta, float eps, void* xout, int xout_dtype,
                                    void* xcopy_bf16, void* h, int h_dtype, float* mean, float* rstd, int64_t n, int channels,
                                    hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256) return SS_ERR_ARG;
  if (gamma && (!beta || !h || !mean || !rstd)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  dim3 g(ss_div_up(n, LN_THREADS / 64)), b(LN_THREADS);
  const int it = (channels + 255) / 256;
#define SS_LN_FWD(ITN) SS_LAUNCH(k_add_ln_fwd<ITN>, g, b, 0, stream, x, x_dtype, y, y_dtype, rowscale, gamma, beta, eps, xout, xout_dtype, xcopy_bf16, h, h_dtype, mean, rstd, n, channels)
  switch (it) { case 1: SS_LN_FWD(1); break; case 2: SS_LN_FWD(2); break; case 3: SS_LN_FWD(3); break; default: SS_LN_FWD(4); break; }
#undef SS_LN_FWD
  return SS_OK;
}

extern "C" int ss_add_layernorm_bwd_blocks(int64_t n) {
  // rows per wave: a wave walks its rows one after the other (each a dependent load -> reduce -> store chain of 1-2 us), so the
  // pooled levels (1,600 / 6,400 rows, launches of 13-17 us) get 2 rows per wave; large levels are capped at 1,024 workgroups anyway
  const int rows_per_wave = n >= 32768 ? 8 : 2;
  int64_t b = (n + (LN_THREADS / 64) * rows_per_wave - 1) / ((LN_THREADS / 64) * rows_per_wave);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int ss_add_layernorm_bwd(const void* g_xout, int g_xout_dtype, const void* g_xcopy, int g_xcopy_dtype,
                                    const void* g_h, int g_h_dtype, const void* v,
                                    int v_dtype, const float* mean, const float* rstd, const float* gamma,
                                    const float* rowscale, void* g_x, int g_x_dtype, void* g_y, int g_y_dtype,
                                    float* dgamma_part, float* dbeta_part, int64_t n, int channels, int nblocks,
                                    hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256 || nblocks < 1) return SS_ERR_ARG;
  if (g_h && (!v || !mean || !rstd || !gamma || !dgamma_part || !dbeta_part)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  dim3 g(nblocks), b(LN_THREADS);
  const int it = (channels + 255) / 256;
#define SS_LN_BWD(ITN) SS_LAUNCH(k_add_ln_bwd<ITN>, g, b, 0, stream, g_xout, g_xout_dtype, g_xcopy, g_xcopy_dtype, g_h, g_h_dtype, v, v_dtype, mean, rstd, gamma, rowscale, g_x, g_x_dtype, g_y, g_y_dtype, dgamma_part, dbeta_part, n, channels)
  switch (it) { case 1: SS_LN_BWD(1); break; case 2: SS_LN_BWD(2); break; case 3: SS_LN_BWD(3); break; default: SS_LN_BWD(4); break; }
#undef SS_LN_BWD
  return SS_OK;
}

// =====================================================================================
// First seam of a pre-norm Block (ptv3:318-325): x += LN0(t); h = LN1(x) in ONE pass:
//     u = LN0(t) * gamma0 + beta0 (kept in fp32 registers, never written);  v = x + u;  xout = v;  h = LN1(v)*gamma1 + beta1
// backward: g_v = g_xout + LN1'(g_h);  g_x = g_v;  g_t = LN0'(g_v);  per-block partials of the four affine gradients.
// Saves the LN0 output round trip (2 x n x C bf16 forward, 2 x backward) and one launch each way per Block.
// =====================================================================================
template <int IT>
__global__ void __launch_bounds__(LN_THREADS)
k_ln_add_ln_fwd(const void* __restrict__ x, int x_dt, const void* __restrict__ t, int t_dt, const float* __restrict__ gamma0,
                const float* __restrict__ beta0, float eps0, const float* __restrict__ gamma1, const float* __restrict__ beta1,
                float eps1, float* __restrict__ xout, void* __restrict__ h, int h_dt, float* __restrict__ stats, int64_t n,
                int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  float4 v[IT];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < C) { v[i] = ln_ld4(t, t_dt, row * C + j); sum += v[i].x + v[i].y + v[i].z + v[i].w; }
  }
  const float mu0 = wave_reduce_sum(sum) / C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) { float a = v[i].x - mu0, b = v[i].y - mu0, c = v[i].z - mu0, d = v[i].w - mu0; sq += a * a + b * b + c * c + d * d; }
  }
  const float r0 = rsqrtf(wave_reduce_sum(sq) / C + eps0);
  sum = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) {
      float4 g = *reinterpret_cast<const float4*>(gamma0 + j), b = *reinterpret_cast<const float4*>(beta0 + j);
      float4 a = ln_ld4(x, x_dt, row * C + j);
      a.x += (v[i].x - mu0) * r0 * g.x + b.x; a.y += (v[i].y - mu0) * r0 * g.y + b.y;
      a.z += (v[i].z - mu0) * r0 * g.z + b.z; a.w += (v[i].w - mu0) * r0 * g.w + b.w;
      v[i] = a;
      *reinterpret_cast<float4*>(xout + row * C + j) = a;
      sum += a.x + a.y + a.z + a.w;
    }
  }
  const float mu1 = wave_reduce_sum(sum) / C;
  sq = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) { float a = v[i].x - mu1, b = v[i].y - mu1, c = v[i].z - mu1, d = v[i].w - mu1; sq += a * a + b * b + c * c + d * d; }
  }
  const float r1 = rsqrtf(wave_reduce_sum(sq) / C + eps1);
  if (lane == 0) *reinterpret_cast<float4*>(stats + row * 4) = make_float4(mu0, r0, mu1, r1);
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) {
      float4 g = *reinterpret_cast<const float4*>(gamma1 + j), b = *reinterpret_cast<const float4*>(beta1 + j);
      ln_st4(h, h_dt, row * C + j, make_float4((v[i].x - mu1) * r1 * g.x + b.x, (v[i].y - mu1) * r1 * g.y + b.y,
                                               (v[i].z - mu1) * r1 * g.z + b.z, (v[i].w - mu1) * r1 * g.w + b.w));
    }
  }
}

// LayerNorm input gradient of one row held as xh (normalised input) and gy (= g_out * gamma): r * (gy - mean(gy) - xh * mean(gy*xh))
#define LN_ROW_BWD(XH, GY, R, OUT)                                                                        \
  {                                                                                                       \
    float c1_ = 0.f, c2_ = 0.f;                                                                           \
    _Pragma("unroll") for (int i = 0; i < IT; ++i) {                                                      \
      c1_ += GY[i].x + GY[i].y + GY[i].z + GY[i].w;                                                       \
      c2_ += GY[i].x * XH[i].x + GY[i].y * XH[i].y + GY[i].z * XH[i].z + GY[i].w * XH[i].w;               \
    }                                                                                                     \
    c1_ = wave_reduce_sum(c1_) / C; c2_ = wave_reduce_sum(c2_) / C;                                       \
    _Pragma("unroll") for (int i = 0; i < IT; ++i)                                                        \
      OUT[i] = make_float4(R * (GY[i].x - c1_ - XH[i].x * c2_), R * (GY[i].y - c1_ - XH[i].y * c2_),      \
                           R * (GY[i].z - c1_ - XH[i].z * c2_), R * (GY[i].w - c1_ - XH[i].w * c2_));     \
  }

template <int IT>
__global__ void __launch_bounds__(LN_THREADS, (IT <= 3 ? 4 : 2))      // <= 128 VGPRs at C <= 768: 4 waves / SIMD for an HBM-bound kernel
k_ln_add_ln_bwd(const float* __restrict__ g_xout, const void* __restrict__ g_h, int gh_dt, const float* __restrict__ xout,
                const void* __restrict__ t, int t_dt, const float* __restrict__ stats, const float* __restrict__ gamma0,
                const float* __restrict__ gamma1, void* __restrict__ g_x, int gx_dt, void* __restrict__ g_t, int gt_dt,
                float* __restrict__ part /* [4][nblocks][C]: dgamma0 dbeta0 dgamma1 dbeta1 */, int64_t n, int C) {
  __shared__ float red[2][LN_THREADS / 64][IT * 4][64];   // two rounds of two partial arrays: 24 KB at C = 768 (48 KB capped the kernel at 3 waves / SIMD)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int waves_total = gridDim.x * (LN_THREADS / 64);
  // the two gamma vectors live in LDS (6 KB at C = 768), not in 24 VGPRs: the kernel is HBM-bound and needs the waves
  __shared__ __attribute__((aligned(16))) float gam[2][IT * 256];
  for (int j = threadIdx.x; j < IT * 256; j += LN_THREADS) { gam[0][j] = j < C ? gamma0[j] : 0.f; gam[1][j] = j < C ? gamma1[j] : 0.f; }
  __syncthreads();
  float4 dg0[IT], db0[IT], dg1[IT], db1[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) { dg0[i] = make_float4(0.f, 0.f, 0.f, 0.f); db0[i] = dg0[i]; dg1[i] = dg0[i]; db1[i] = dg0[i]; }
  for (int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + wave; row < n; row += waves_total) {
    const float4 st = *reinterpret_cast<const float4*>(stats + row * 4);
    float4 xh[IT], gy[IT], gv[IT];
    // ---- LN1 backward on v = xout
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); gy[i] = xh[i];
      if (j < C) {
        float4 a = *reinterpret_cast<const float4*>(xout + row * C + j), g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_h) g = ln_ld4(g_h, gh_dt, row * C + j);
        xh[i] = make_float4((a.x - st.z) * st.w, (a.y - st.z) * st.w, (a.z - st.z) * st.w, (a.w - st.z) * st.w);
        const float4 gm1 = *reinterpret_cast<const float4*>(&gam[1][j]);
        gy[i] = make_float4(g.x * gm1.x, g.y * gm1.y, g.z * gm1.z, g.w * gm1.w);
        dg1[i].x += g.x * xh[i].x; dg1[i].y += g.y * xh[i].y; dg1[i].z += g.z * xh[i].z; dg1[i].w += g.w * xh[i].w;
        db1[i].x += g.x; db1[i].y += g.y; db1[i].z += g.z; db1[i].w += g.w;
      }
    }
    LN_ROW_BWD(xh, gy, st.w, gv)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      if (j < C) {
        if (g_xout) { float4 e = *reinterpret_cast<const float4*>(g_xout + row * C + j); gv[i].x += e.x; gv[i].y += e.y; gv[i].z += e.z; gv[i].w += e.w; }
        ln_st4(g_x, gx_dt, row * C + j, gv[i]);
      } else {
        gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // ---- LN0 backward on t with upstream gradient g_v
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); gy[i] = xh[i];
      if (j < C) {
        float4 a = ln_ld4(t, t_dt, row * C + j);
        xh[i] = make_float4((a.x - st.x) * st.y, (a.y - st.x) * st.y, (a.z - st.x) * st.y, (a.w - st.x) * st.y);
        const float4 gm0 = *reinterpret_cast<const float4*>(&gam[0][j]);
        gy[i] = make_float4(gv[i].x * gm0.x, gv[i].y * gm0.y, gv[i].z * gm0.z, gv[i].w * gm0.w);
        dg0[i].x += gv[i].x * xh[i].x; dg0[i].y += gv[i].y * xh[i].y; dg0[i].z += gv[i].z * xh[i].z; dg0[i].w += gv[i].w * xh[i].w;
        db0[i].x += gv[i].x; db0[i].y += gv[i].y; db0[i].z += gv[i].z; db0[i].w += gv[i].w;
      }
    }
    LN_ROW_BWD(xh, gy, st.y, gv)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      if (j < C) ln_st4(g_t, gt_dt, row * C + j, gv[i]);
    }
  }
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (round) __syncthreads();                    // round 0's reads are done
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const float4 a = round ? dg1[i] : dg0[i], b = round ? db1[i] : db0[i];
      red[0][wave][i * 4 + 0][lane] = a.x; red[0][wave][i * 4 + 1][lane] = a.y; red[0][wave][i * 4 + 2][lane] = a.z; red[0][wave][i * 4 + 3][lane] = a.w;
      red[1][wave][i * 4 + 0][lane] = b.x; red[1][wave][i * 4 + 1][lane] = b.y; red[1][wave][i * 4 + 2][lane] = b.z; red[1][wave][i * 4 + 3][lane] = b.w;
    }
    __syncthreads();
    if (wave < 2) {                                // wave q reduces partial array 2*round + q
      const int q = wave;
#pragma unroll
      for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int j = i * 256 + lane * 4 + e;
          if (j < C) {
            float acc = 0.f;
            for (int w = 0; w < LN_THREADS / 64; ++w) acc += red[q][w][i * 4 + e][lane];
            part[((int64_t)(2 * round + q) * gridDim.x + blockIdx.x) * C + j] = acc;
          }
        }
    }
  }
}
#undef LN_ROW_BWD

extern "C" int ss_ln_add_ln_fwd(const void* x, int x_dtype, const void* t, int t_dtype, const float* gamma0, const float* beta0,
                                float eps0, const float* gamma1, const float* beta1, float eps1, float* xout, void* h,
                                int h_dtype, float* stats, int64_t n, int channels, hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256) return SS_ERR_ARG;
  if (!x || !t || !gamma0 || !beta0 || !gamma1 || !beta1 || !xout || !h || !stats) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  dim3 g(ss_div_up(n, LN_THREADS / 64)), b(LN_THREADS);
#define SS_LN2_FWD(ITN) SS_LAUNCH(k_ln_add_ln_fwd<ITN>, g, b, 0, stream, x, x_dtype, t, t_dtype, gamma0, beta0, eps0, gamma1, beta1, eps1, xout, h, h_dtype, stats, n, channels)
  switch ((channels + 255) / 256) { case 1: SS_LN2_FWD(1); break; case 2: SS_LN2_FWD(2); break; case 3: SS_LN2_FWD(3); break; default: SS_LN2_FWD(4); break; }
#undef SS_LN2_FWD
  return SS_OK;
}

extern "C" int ss_ln_add_ln_bwd(const float* g_xout, const void* g_h, int g_h_dtype, const float* xout, const void* t, int t_dtype,
                                const float* stats, const float* gamma0, const float* gamma1, void* g_x, int g_x_dtype, void* g_t,
                                int g_t_dtype, float* part, int64_t n, int channels, int nblocks, hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256 || nblocks < 1) return SS_ERR_ARG;
  if (!xout || !t || !stats || !gamma0 || !gamma1 || !g_x || !g_t || !part || (!g_xout && !g_h)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  dim3 g(nblocks), b(LN_THREADS);
#define SS_LN2_BWD(ITN) SS_LAUNCH(k_ln_add_ln_bwd<ITN>, g, b, 0, stream, g_xout, g_h, g_h_dtype, xout, t, t_dtype, stats, gamma0, gamma1, g_x, g_x_dtype, g_t, g_t_dtype, part, n, channels)
  switch ((channels + 255) / 256) { case 1: SS_LN2_BWD(1); break; case 2: SS_LN2_BWD(2); break; case 3: SS_LN2_BWD(3); break; default: SS_LN2_BWD(4); break; }
#undef SS_LN2_BWD
  return SS_OK;
}

// =====================================================================================
// Fused BatchNorm1d (+ exact erf GELU) over (n, C) rows: pooling / unpooling / stem norms
// (ptv3:581 eps 1e-3; ptv3:439-442, 461-467, 508-511).  Column statistics use the same
// wave-per-row, lane-per-4-columns mapping as the LayerNorm backward: per-lane register partials
// over a grid-stride row loop, LDS reduction across the block's waves, one partial row per block.
// =====================================================================================
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float z) {
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}

template <int IT>
__device__ __forceinline__ void col_partials_store(float4 (&a)[IT], float4 (&b)[IT], float* pa, float* pb, int C) {
  __shared__ float red[2][LN_THREADS / 64][IT * 4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    red[0][wave][i * 4 + 0][lane] = a[i].x; red[0][wave][i * 4 + 1][lane] = a[i].y;
    red[0][wave][i * 4 + 2][lane] = a[i].z; red[0][wave][i * 4 + 3][lane] = a[i].w;
    red[1][wave][i * 4 + 0][lane] = b[i].x; red[1][wave][i * 4 + 1][lane] = b[i].y;
    red[1][wave][i * 4 + 2][lane] = b[i].z; red[1][wave][i * 4 + 3][lane] = b[i].w;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int j = i * 256 + lane * 4 + e;
        if (j < C) {
          float s0 = 0.f, s1 = 0.f;
          for (int w = 0; w < LN_THREADS / 64; ++w) { s0 += red[0][w][i * 4 + e][lane]; s1 += red[1][w][i * 4 + e][lane]; }
          pa[(int64_t)blockIdx.x * C + j] = s0;
          pb[(int64_t)blockIdx.x * C + j] = s1;
        }
      }
  }
}

// partial column sums of x and x^2 (shifted by `shift[c]` for conditioning; shift may be NULL)
template <int IT>
__global__ void __launch_bounds__(LN_THREADS)
k_col_stats(const void* __restrict__ x, int x_dt, const float* __restrict__ shift, float* __restrict__ psum,
            float* __restrict__ psq, int64_t n, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int waves_total = gridDim.x * (LN_THREADS / 64);
  float4 s[IT], q[IT], sh[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    s[i] = make_float4(0.f, 0.f, 0.f, 0.f); q[i] = s[i]; sh[i] = s[i];
    int j = i * 256 + lane * 4;
    if (shift && j < C) sh[i] = *reinterpret_cast<const float4*>(shift + j);
  }
  for (int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + wave; row < n; row += waves_total) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      if (j < C) {
        float4 a = ln_ld4(x, x_dt, row * C + j);
        a.x -= sh[i].x; a.y -= sh[i].y; a.z -= sh[i].z; a.w -= sh[i].w;
        s[i].x += a.x; s[i].y += a.y; s[i].z += a.z; s[i].w += a.w;
        q[i].x += a.x * a.x; q[i].y += a.y * a.y; q[i].z += a.z * a.z; q[i].w += a.w * a.w;
      }
    }
  }
  col_partials_store<IT>(s, q, psum, psq, C);
}

// y = act((x - mean) * rstd * gamma + beta)
template <int IT>
__global__ void __launch_bounds__(LN_THREADS)
k_bn_act_fwd(const void* __restrict__ x, int x_dt, const float* __restrict__ mean, const float* __restrict__ rstd,
             const float* __restrict__ gamma, const float* __restrict__ beta, int act, void* __restrict__ y, int y_dt,
             int64_t n, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) {
      float4 a = ln_ld4(x, x_dt, row * C + j);
      float4 m = *reinterpret_cast<const float4*>(mean + j), r = *reinterpret_cast<const float4*>(rstd + j);
      float4 g = *reinterpret_cast<const float4*>(gamma + j), b = *reinterpret_cast<const float4*>(beta + j);
      float4 z = make_float4((a.x - m.x) * r.x * g.x + b.x, (a.y - m.y) * r.y * g.y + b.y, (a.z - m.z) * r.z * g.z + b.z,
                             (a.w - m.w) * r.w * g.w + b.w);
      if (act) z = make_float4(gelu_f(z.x), gelu_f(z.y), gelu_f(z.z), gelu_f(z.w));
      ln_st4(y, y_dt, row * C + j, z);
    }
  }
}

// backward pass 1: dz = dy * act'(z); partial column sums of dz and dz * xhat
template <int IT>
__global__ void __launch_bounds__(LN_THREADS)
k_bn_act_bwd_reduce(const void* __restrict__ dy, int dy_dt, const void* __restrict__ x, int x_dt,
                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                    const float* __restrict__ beta, int act, float* __restrict__ pdz, float* __restrict__ pdzx, int64_t n,
                    int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int waves_total = gridDim.x * (LN_THREADS / 64);
  float4 s[IT], q[IT], m[IT], r[IT], g[IT], b[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    s[i] = make_float4(0.f, 0.f, 0.f, 0.f); q[i] = s[i]; m[i] = s[i]; r[i] = s[i]; g[i] = s[i]; b[i] = s[i];
    int j = i * 256 + lane * 4;
    if (j < C) {
      m[i] = *reinterpret_cast<const float4*>(mean + j); r[i] = *reinterpret_cast<const float4*>(rstd + j);
      g[i] = *reinterpret_cast<const float4*>(gamma + j); b[i] = *reinterpret_cast<const float4*>(beta + j);
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + wave; row < n; row += waves_total) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      if (j < C) {
        float4 a = ln_ld4(x, x_dt, row * C + j), d = ln_ld4(dy, dy_dt, row * C + j);
        float4 xh = make_float4((a.x - m[i].x) * r[i].x, (a.y - m[i].y) * r[i].y, (a.z - m[i].z) * r[i].z, (a.w - m[i].w) * r[i].w);
        if (act) {
          d.x *= dgelu_f(xh.x * g[i].x + b[i].x); d.y *= dgelu_f(xh.y * g[i].y + b[i].y);
          d.z *= dgelu_f(xh.z * g[i].z + b[i].z); d.w *= dgelu_f(xh.w * g[i].w + b[i].w);
        }
        s[i].x += d.x; s[i].y += d.y; s[i].z += d.z; s[i].w += d.w;
        q[i].x += d.x * xh.x; q[i].y += d.y * xh.y; q[i].z += d.z * xh.z; q[i].w += d.w * xh.w;
      }
    }
  }
  col_partials_store<IT>(s, q, pdz, pdzx, C);
}

// backward pass 2: dx = gamma * rstd * (dz - c1 - xhat * c2); training: c1 = sum_dz / n, c2 = sum_dzx / n; eval: 0
template <int IT>
__global__ void __launch_bounds__(LN_THREADS)
k_bn_act_bwd_apply(const void* __restrict__ dy, int dy_dt, const void* __restrict__ x, int x_dt,
                   const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                   const float* __restrict__ beta, int act, const float* __restrict__ c1, const float* __restrict__ c2,
                   void* __restrict__ dx, int dx_dt, int64_t n, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int j = i * 256 + lane * 4;
    if (j < C) {
      float4 a = ln_ld4(x, x_dt, row * C + j), d = ln_ld4(dy, dy_dt, row * C + j);
      float4 m = *reinterpret_cast<const float4*>(mean + j), r = *reinterpret_cast<const float4*>(rstd + j);
      float4 g = *reinterpret_cast<const float4*>(gamma + j), b = *reinterpret_cast<const float4*>(beta + j);
      float4 k1 = c1 ? *reinterpret_cast<const float4*>(c1 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 k2 = c2 ? *reinterpret_cast<const float4*>(c2 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 xh = make_float4((a.x - m.x) * r.x, (a.y - m.y) * r.y, (a.z - m.z) * r.z, (a.w - m.w) * r.w);
      if (act) {
        d.x *= dgelu_f(xh.x * g.x + b.x); d.y *= dgelu_f(xh.y * g.y + b.y);
        d.z *= dgelu_f(xh.z * g.z + b.z); d.w *= dgelu_f(xh.w * g.w + b.w);
      }
      float4 o = make_float4(g.x * r.x * (d.x - k1.x - xh.x * k2.x), g.y * r.y * (d.y - k1.y - xh.y * k2.y),
                             g.z * r.z * (d.z - k1.z - xh.z * k2.z), g.w * r.w * (d.w - k1.w - xh.w * k2.w));
      ln_st4(dx, dx_dt, row * C + j, o);
    }
  }
}

// ---- BatchNorm: the small vector work between the passes, one launch each way ---------------------------------------
// Forward (training): per-block partial sums of (x - shift), (x - shift)^2  ->  batch mean / rstd, the running-statistics
// update of nn.BatchNorm1d (momentum, unbiased variance) and the batch counter.  PyTorch spells this as ~15 elementwise
// launches on C-element vectors (s / n, shift + d, q / n - d^2, clamp, rsqrt, mul_, add_ ...) per BatchNorm layer.
// (a block = 32 channels x 32 row groups: the nb <= 1024 partials of a channel are summed by 32 threads, then through LDS)
#define BNF_THREADS 1024
__device__ __forceinline__ void bnf_reduce(const float* __restrict__ part, int nb, int C, int c, int rg, float& s, float& q) {
  __shared__ float sh[2][32][33];
  float a = 0.f, b_ = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int b = rg; b < nb; b += 32) { a += part[(int64_t)b * C + c]; b_ += part[((int64_t)nb + b) * C + c]; }
  }
  const int cl = threadIdx.x & 31;
  sh[0][rg][cl] = a; sh[1][rg][cl] = b_;
  __syncthreads();
  s = 0.f; q = 0.f;
  if (rg == 0) {
#pragma unroll
    for (int r = 0; r < 32; ++r) { s += sh[0][r][cl]; q += sh[1][r][cl]; }
  }
}
__global__ void __launch_bounds__(BNF_THREADS)
k_bn_stats_finish(const float* __restrict__ part, const float* shift /* may alias running_mean */, int nb, int C, float n,
                  float unbias, float momentum, float eps, float* running_mean, float* __restrict__ running_var,
                  long long* __restrict__ num_batches, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
  float s, q;
  bnf_reduce(part, nb, C, c, rg, s, q);
  if (rg != 0 || c >= C) return;
  const float d = s / n;
  const float m = (shift ? shift[c] : 0.f) + d;
  const float var = fmaxf(q / n - d * d, 0.f);
  mean[c] = m;
  rstd[c] = rsqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = running_mean[c] * (1.f - momentum) + m * momentum;
    running_var[c] = running_var[c] * (1.f - momentum) + (var * unbias) * momentum;
  }
}
// Backward: partial sums of dz, dz * xhat -> sums (2, C) = (dbeta, dgamma) and, in training, coef (2, C) = sums / n.
__global__ void __launch_bounds__(BNF_THREADS)
k_bn_bwd_finish(const float* __restrict__ part, int nb, int C, float n, float* __restrict__ sums, float* __restrict__ coef) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  float a, b_;
  bnf_reduce(part, nb, C, c, rg, a, b_);
  if (rg != 0 || c >= C) return;
  sums[c] = a; sums[C + c] = b_;
  if (coef) { coef[c] = a / n; coef[C + c] = b_ / n; }
}

#define SS_IT_SWITCH(MACRO) switch (it) { case 1: MACRO(1); break; case 2: MACRO(2); break; case 3: MACRO(3); break; default: MACRO(4); break; }

extern "C" int ss_col_stats(const void* x, int x_dtype, const float* shift, float* psum, float* psq, int64_t n, int channels,
                            int nblocks, hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256 || nblocks < 1) return SS_ERR_ARG;
  const int it = (channels + 255) / 256;
#define SS_CS(ITN) SS_LAUNCH(k_col_stats<ITN>, dim3(nblocks), dim3(LN_THREADS), 0, stream, x, x_dtype, shift, psum, psq, n, channels)
  SS_IT_SWITCH(SS_CS)
#undef SS_CS
  return SS_OK;
}
extern "C" int ss_bn_act_fwd(const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, int act, void* y, int y_dtype, int64_t n, int channels, hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  const int it = (channels + 255) / 256;
  dim3 g(ss_div_up(n, LN_THREADS / 64)), b(LN_THREADS);
#define SS_BF(ITN) SS_LAUNCH(k_bn_act_fwd<ITN>, g, b, 0, stream, x, x_dtype, mean, rstd, gamma, beta, act, y, y_dtype, n, channels)
  SS_IT_SWITCH(SS_BF)
#undef SS_BF
  return SS_OK;
}
extern "C" int ss_bn_act_bwd_reduce(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, int act, float* pdz, float* pdzx,
                                    int64_t n, int channels, int nblocks, hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256 || nblocks < 1) return SS_ERR_ARG;
  const int it = (channels + 255) / 256;
#define SS_BR(ITN) SS_LAUNCH(k_bn_act_bwd_reduce<ITN>, dim3(nblocks), dim3(LN_THREADS), 0, stream, dy, dy_dtype, x, x_dtype, mean, rstd, gamma, beta, act, pdz, pdzx, n, channels)
  SS_IT_SWITCH(SS_BR)
#undef SS_BR
  return SS_OK;
}
extern "C" int ss_bn_act_bwd_apply(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, int act, const float* c1,
                                   const float* c2, void* dx, int dx_dtype, int64_t n, int channels, hipStream_t stream) {
  if (n < 0 || channels <= 0 || (channels & 3) || channels > LN_MAXIT * 256) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  const int it = (channels + 255) / 256;
  dim3 g(ss_div_up(n, LN_THREADS / 64)), b(LN_THREADS);
#define SS_BA(ITN) SS_LAUNCH(k_bn_act_bwd_apply<ITN>, g, b, 0, stream, dy, dy_dtype, x, x_dtype, mean, rstd, gamma, beta, act, c1, c2, dx, dx_dtype, n, channels)
  SS_IT_SWITCH(SS_BA)
#undef SS_BA
  return SS_OK;
}


// ---- grouped partial-sum reduction ------------------------------------------------------------------------------------
// The backward seams above leave per-block partial sums of the affine gradients, part (K, nb, C) f32 (K = 2 or 4 parameter
// vectors).  Reducing each with its own launch is 44 tiny kernels per step on the pooled levels; the stage identity node
// (functional._StageParams) queues them and this kernel reduces ALL of a stage's partials in one launch.
// desc: 4 int64 words per problem = {part, dst (K*C f32), nb, C | (K*C) << 32}; wg_start (nprob + 1): first workgroup.
// A workgroup owns GPS_COLS consecutive outputs and splits the nb partial rows over 8 row groups (one thread per (output, group), LDS
// reduce): the first form gave each output ONE thread that walked all nb <= 1024 rows -- a 256-step dependent chain on 60 workgroups,
// 65 us per stage and 0.45 ms per step for a few MB of reads.
#define GPS_COLS 32
__global__ void __launch_bounds__(256)
k_group_partial_sums(const int64_t* __restrict__ desc, const int32_t* __restrict__ wg_start, int nprob) {
  __shared__ float red[8][GPS_COLS];
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < nprob && wg_start[p + 1] <= b) ++p;
  const int64_t* d = desc + (int64_t)p * 4;
  const float* part = reinterpret_cast<const float*>(d[0]);
  float* dst = reinterpret_cast<float*>(d[1]);
  const int nb = (int)d[2];
  const int C = (int)((uint64_t)d[3] & 0xffffffffu), KC = (int)((uint64_t)d[3] >> 32);
  const int cg = threadIdx.x & (GPS_COLS - 1), rg = threadIdx.x / GPS_COLS;
  const int j = (b - wg_start[p]) * GPS_COLS + cg;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (j < KC) {
    const int k = j / C, c = j - k * C;
    const float* src = part + (int64_t)k * nb * C + c;
    int bb = rg;
    for (; bb + 24 < nb; bb += 32) {
      a0 += src[(int64_t)bb * C]; a1 += src[(int64_t)(bb + 8) * C]; a2 += src[(int64_t)(bb + 16) * C]; a3 += src[(int64_t)(bb + 24) * C];
    }
    for (; bb < nb; bb += 8) a0 += src[(int64_t)bb * C];
  }
  red[rg][cg] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rg == 0 && j < KC) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][cg];
    dst[j] = s;
  }
}

extern "C" int ss_group_partial_sums_outputs_per_workgroup(void) { return GPS_COLS; }
extern "C" int ss_group_partial_sums(const int64_t* desc, const int32_t* wg_start, int nprob, int total_workgroups,
                                     ss_stream_t stream) {
  if (nprob <= 0 || total_workgroups <= 0) return SS_OK;
  if (!desc || !wg_start) return SS_ERR_ARG;
  SS_LAUNCH(k_group_partial_sums, dim3((unsigned)total_workgroups), dim3(256), 0, stream, desc, wg_start, nprob);
  return SS_OK;
}

extern "C" int ss_bn_stats_finish(const float* part, const float* shift, int nblocks, int channels, int64_t n, float momentum,
                                  float eps, float* running_mean, float* running_var, int64_t* num_batches, float* mean,
                                  float* rstd, hipStream_t stream) {
  if (!part || !mean || !rstd || nblocks < 1 || channels <= 0 || n <= 0 || ((running_mean == nullptr) != (running_var == nullptr)))
    return SS_ERR_ARG;
  const float unbias = (float)((double)n / (double)(n > 1 ? n - 1 : 1));
  SS_LAUNCH(k_bn_stats_finish, dim3(ss_div_up(channels, 32)), dim3(BNF_THREADS), 0, stream, part, shift, nblocks, channels, (float)n,
            unbias, momentum, eps, running_mean, running_var, (long long*)num_batches, mean, rstd);
  return SS_OK;
}
extern "C" int ss_bn_bwd_finish(const float* part, int nblocks, int channels, int64_t n, float* sums, float* coef,
                                hipStream_t stream) {
  if (!part || !sums || nblocks < 1 || channels <= 0 || n <= 0) return SS_ERR_ARG;
  SS_LAUNCH(k_bn_bwd_finish, dim3(ss_div_up(channels, 32)), dim3(BNF_THREADS), 0, stream, part, nblocks, channels, (float)n, sums, coef);
  return SS_OK;
}


// =====================================================================================
// Stand-alone exact-erf GELU of the MLP (ptv3:225-248: Linear -> nn.GELU() -> Linear), forward and backward, on (n, 4C) bf16 or
// fp32 tensors as flat arrays: y = 0.5 x (1 + erf(x / sqrt 2)), dx = dy (Phi(x) + x phi(x)) -- the fp32 formulas torch evaluates
// (at::native GeluCUDAKernelImpl / GeluBackwardCUDAKernelImpl, approximate = 'none'), one rounding to the storage type.
// HBM-bound: 16-byte lanes, eight (bf16) or four (fp32) elements per thread.  Why it is not an epilogue of the fc1 GEMM: DESIGN.md
// section 4 ("Why the exact-erf GELU is not in a GEMM epilogue").
// =====================================================================================
__device__ __forceinline__ float dgelu_exact(float z) {
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
}
__device__ __forceinline__ void bf8_unpack(const uint4& u, float (&v)[8]) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u); v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u); v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 bf8_pack(const float (&v)[8]) {
  uint4 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}

template <bool BWD>
__global__ void __launch_bounds__(256)
k_gelu_bf16(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy, unsigned short* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float a[8], g[8], r[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(x + i), a);
    if (BWD) bf8_unpack(*reinterpret_cast<const uint4*>(dy + i), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = BWD ? g[e] * dgelu_exact(a[e]) : gelu_f(a[e]);
    *reinterpret_cast<uint4*>(out + i) = bf8_pack(r);
  } else {
    for (int64_t k = i; k < n; ++k) {
      const float a = bf16_to_f32(x[k]);
      out[k] = f32_to_bf16(BWD ? bf16_to_f32(dy[k]) * dgelu_exact(a) : gelu_f(a));
    }
  }
}
template <bool BWD>
__global__ void __launch_bounds__(256)
k_gelu_f32(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) {
    const float4 a = *reinterpret_cast<const float4*>(x + i);
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
    if (BWD) g = *reinterpret_cast<const float4*>(dy + i);
    float4 r;
    r.x = BWD ? g.x * dgelu_exact(a.x) : gelu_f(a.x); r.y = BWD ? g.y * dgelu_exact(a.y) : gelu_f(a.y);
    r.z = BWD ? g.z * dgelu_exact(a.z) : gelu_f(a.z); r.w = BWD ? g.w * dgelu_exact(a.w) : gelu_f(a.w);
    *reinterpret_cast<float4*>(out + i) = r;
  } else {
    for (int64_t k = i; k < n; ++k) out[k] = BWD ? dy[k] * dgelu_exact(x[k]) : gelu_f(x[k]);
  }
}

// x, dy (NULL = forward), out: flat arrays of `numel` elements of `dtype` (SS_F32 | SS_BF16), 16-byte aligned bases
extern "C" int ss_gelu(const void* x, const void* dy, void* out, int64_t numel, int dtype, hipStream_t stream) {
  if (numel < 0 || (dtype != SS_F32 && dtype != SS_BF16)) return SS_ERR_ARG;
  if (numel == 0) return SS_OK;
  if (!x || !out || (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15)) return SS_ERR_ARG;
  const int per = dtype == SS_BF16 ? 8 : 4;
  dim3 g((unsigned)ss_div_up(ss_div_up(numel, per), 256)), b(256);
  if (dtype == SS_BF16) {
    if (dy) SS_LAUNCH((k_gelu_bf16<true>), g, b, 0, stream, (const unsigned short*)x, (const unsigned short*)dy, (unsigned short*)out, numel);
    else SS_LAUNCH((k_gelu_bf16<false>), g, b, 0, stream, (const unsigned short*)x, (const unsigned short*)nullptr, (unsigned short*)out, numel);
  } else {
    if (dy) SS_LAUNCH((k_gelu_f32<true>), g, b, 0, stream, (const float*)x, (const float*)dy, (float*)out, numel);
    else SS_LAUNCH((k_gelu_f32<false>), g, b, 0, stream, (const float*)x, (const float*)nullptr, (float*)out, numel);
  }
  return SS_OK;
}
