// Vision-language distillation head in one pass each way (gfx950, HBM-bound).
//
// Reference (pointcept/models/default.py:98-109, pointcept/models/losses/misc.py:254-295): F.normalize(feat), then
// CosineSimilarity and L2Loss each gather pred[valid] / target[valid] (two boolean-mask copies of 768-wide rows per
// loss) and reduce -- about six passes over two (N, 768) fp32 tensors plus a device->host sync per mask.  Here one
// kernel reads every feature / target row ONCE and emits
//     p     = f / max(|f|, 1e-12)                                  (optional; the contrastive loss and eval consume it)
//     sums  = [ sum_valid (1 - cos(p, t)),  sum_valid |p - t|^2,  #valid ]
// with cos(p, t) = p.t / (max(|p|, 1e-8) max(|t|, 1e-8)) (torch.nn.CosineSimilarity), and one backward kernel turns
// (d sums, optional d p) into d f.  One wave per row, the row lives in registers (C % 4 == 0, C <= 2048); per-wave
// partial sums -> LDS -> per-block partials -> a single-block finish kernel (deterministic, no atomics).
#include "common.h"
#include "../../include/scenesplat_hip.h"

#define HD_THREADS 256
#define HD_MAX_BLOCKS 1024

__device__ __forceinline__ float4 hd_ld4(const void* p, int dtype, int64_t idx) {
  if (dtype == SS_F32) return *reinterpret_cast<const float4*>((const float*)p + idx);
  uint2 u = *reinterpret_cast<const uint2*>((const unsigned short*)p + idx);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void hd_st4(void* p, int dtype, int64_t idx, float4 v) {
  if (dtype == SS_F32) { *reinterpret_cast<float4*>((float*)p + idx) = v; return; }
  uint2 u; u.x = pack_bf16x2(v.x, v.y); u.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>((unsigned short*)p + idx) = u;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// rowstat (n,4) f32: [ s = 1/max(|f|,1e-12) (1 when !normalize), a = 1/max(|p|,1e-8), b = 1/max(|t|,1e-8), p.t ]
template <int IT>
__global__ void __launch_bounds__(HD_THREADS)
k_head_fwd(const void* __restrict__ feat, int f_dt, const void* __restrict__ target, int t_dt,
           const unsigned char* __restrict__ mask, int normalize, void* __restrict__ p_out, int p_dt,
           float* __restrict__ rowstat, float* __restrict__ part, int64_t n, int C) {
  __shared__ float red[HD_THREADS / 64][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float acc_cos = 0.f, acc_l2 = 0.f, acc_cnt = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * (HD_THREADS / 64) + wv; row < n; row += (int64_t)gridDim.x * (HD_THREADS / 64)) {
    float4 f[IT];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      f[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < C) { f[i] = hd_ld4(feat, f_dt, row * C + j); ss += dot4(f[i], f[i]); }
    }
    ss = wave_reduce_sum(ss);
    const float nf = sqrtf(ss);
    const float s = normalize ? 1.f / fmaxf(nf, 1e-12f) : 1.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      f[i].x *= s; f[i].y *= s; f[i].z *= s; f[i].w *= s;
      if (p_out && j < C) hd_st4(p_out, p_dt, row * C + j, f[i]);
    }
    if (!target) { if (lane == 0) rowstat[row * 4] = s; continue; }
    float pt = 0.f, tt = 0.f, l2 = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      if (j < C) {
        float4 t = hd_ld4(target, t_dt, row * C + j);
        pt += dot4(f[i], t); tt += dot4(t, t);
        float dx = f[i].x - t.x, dy = f[i].y - t.y, dz = f[i].z - t.z, dw = f[i].w - t.w;
        l2 += dx * dx + dy * dy + dz * dz + dw * dw;
      }
    }
    pt = wave_reduce_sum(pt); tt = wave_reduce_sum(tt); l2 = wave_reduce_sum(l2);
    const float np_ = nf * s;                                 // |p| (1 up to rounding when normalised)
    const float a = 1.f / fmaxf(np_, 1e-8f), b = 1.f / fmaxf(sqrtf(tt), 1e-8f);
    if (lane == 0) *reinterpret_cast<float4*>(rowstat + row * 4) = make_float4(s, a, b, pt);
    if (mask[row]) { acc_cos += 1.f - a * b * pt; acc_l2 += l2; acc_cnt += 1.f; }
  }
  if (!target) return;
  if (lane == 0) { red[wv][0] = acc_cos; red[wv][1] = acc_l2; red[wv][2] = acc_cnt; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < HD_THREADS / 64; ++w) v += red[w][threadIdx.x];
    part[(int64_t)blockIdx.x * 3 + threadIdx.x] = v;
  }
}

// sums[k] = sum_b part[b][k] in double (one block; fixed order)
__global__ void __launch_bounds__(256) k_head_finish(const float* __restrict__ part, int nblocks, float* __restrict__ sums) {
  __shared__ double red[256];
  for (int k = 0; k < 3; ++k) {
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) v += (double)part[(int64_t)b * 3 + k];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) sums[k] = (float)red[0];
    __syncthreads();
  }
}

// d f from d sums (coef[0] = dL/d sums[0], coef[1] = dL/d sums[1], device memory: no host sync) and an optional
// gradient dp_extra arriving at p from other consumers:
//   d p = m [ -coef0 (a b t - cos a^2 p) + 2 coef1 (p - t) ] + dp_extra,   d f = s (d p - p (p . d p))   (normalised)
template <int IT>
__global__ void __launch_bounds__(HD_THREADS)
k_head_bwd(const void* __restrict__ feat, int f_dt, const void* __restrict__ target, int t_dt,
           const unsigned char* __restrict__ mask, int normalize, const float* __restrict__ rowstat,
           const float* __restrict__ coef, const void* __restrict__ dp_extra, int dp_dt, void* __restrict__ dfeat, int df_dt,
           int64_t n, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float c0 = coef ? coef[0] : 0.f, c1 = coef ? coef[1] : 0.f;
  for (int64_t row = (int64_t)blockIdx.x * (HD_THREADS / 64) + wv; row < n; row += (int64_t)gridDim.x * (HD_THREADS / 64)) {
    float s = 1.f, a = 0.f, b = 0.f, pt = 0.f;
    if (target) { float4 st = *reinterpret_cast<const float4*>(rowstat + row * 4); s = st.x; a = st.y; b = st.z; pt = st.w; }
    else if (normalize) s = rowstat[row * 4];
    const float m = (target && mask[row]) ? 1.f : 0.f;
    const float cs = a * b * pt;
    const float kt = -m * c0 * a * b, kp = m * c0 * cs * a * a, kl = 2.f * m * c1;
    float4 p[IT], dp[IT];
    float pdp = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      p[i] = dp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < C) {
        float4 f = hd_ld4(feat, f_dt, row * C + j);
        p[i] = make_float4(f.x * s, f.y * s, f.z * s, f.w * s);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dp_extra) g = hd_ld4(dp_extra, dp_dt, row * C + j);
        if (target) {
          float4 t = hd_ld4(target, t_dt, row * C + j);
          g.x += kt * t.x + kp * p[i].x + kl * (p[i].x - t.x); g.y += kt * t.y + kp * p[i].y + kl * (p[i].y - t.y);
          g.z += kt * t.z + kp * p[i].z + kl * (p[i].z - t.z); g.w += kt * t.w + kp * p[i].w + kl * (p[i].w - t.w);
        }
        dp[i] = g;
        pdp += dot4(p[i], g);
      }
    }
    if (normalize) pdp = wave_reduce_sum(pdp);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int j = i * 256 + lane * 4;
      if (j < C) {
        float4 g = dp[i];
        if (normalize) g = make_float4(s * (g.x - p[i].x * pdp), s * (g.y - p[i].y * pdp), s * (g.z - p[i].z * pdp), s * (g.w - p[i].w * pdp));
        hd_st4(dfeat, df_dt, row * C + j, g);
      }
    }
  }
}

static int head_blocks(int64_t n) {
  int64_t b = (n + HD_THREADS / 64 - 1) / (HD_THREADS / 64);
  return (int)(b < 1 ? 1 : (b > HD_MAX_BLOCKS ? HD_MAX_BLOCKS : b));
}

extern "C" int ss_lang_head_blocks(int64_t n) { return head_blocks(n); }

extern "C" int ss_lang_head_fwd(const void* feat, int feat_dtype, const void* target, int target_dtype, const unsigned char* mask,
                                int normalize, void* p_out, int p_dtype, float* rowstat, float* part, float* sums, int64_t n,
                                int channels, ss_stream_t stream) {
  if (n < 0 || channels <= 0 || channels % 4 || channels > 2048) return SS_ERR_ARG;
  if (!feat || (target && (!mask || !part || !sums)) || !rowstat) return SS_ERR_ARG;
  if (n == 0) { if (sums) hipMemsetAsync(sums, 0, 3 * sizeof(float), stream); return SS_OK; }
  const int nb = head_blocks(n), it = (channels + 255) / 256;
#define HD_FWD(IT) SS_LAUNCH(k_head_fwd<IT>, dim3(nb), dim3(HD_THREADS), 0, stream, feat, feat_dtype, target, target_dtype, mask, \
                             normalize, p_out, p_dtype, rowstat, part, n, channels)
  switch (it) {
    case 1: HD_FWD(1); break; case 2: HD_FWD(2); break; case 3: HD_FWD(3); break; case 4: HD_FWD(4); break;
    case 5: HD_FWD(5); break; case 6: HD_FWD(6); break; case 7: HD_FWD(7); break; default: HD_FWD(8); break;
  }
#undef HD_FWD
  if (target) SS_LAUNCH(k_head_finish, dim3(1), dim3(256), 0, stream, part, nb, sums);
  return SS_OK;
}

extern "C" int ss_lang_head_bwd(const void* feat, int feat_dtype, const void* target, int target_dtype, const unsigned char* mask,
                                int normalize, const float* rowstat, const float* coef, const void* dp_extra, int dp_dtype,
                                void* dfeat, int dfeat_dtype, int64_t n, int channels, ss_stream_t stream) {
  if (n < 0 || channels <= 0 || channels % 4 || channels > 2048) return SS_ERR_ARG;
  if (!feat || !dfeat || !rowstat || (target && (!mask || !coef))) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  const int nb = head_blocks(n), it = (channels + 255) / 256;
#define HD_BWD(IT) SS_LAUNCH(k_head_bwd<IT>, dim3(nb), dim3(HD_THREADS), 0, stream, feat, feat_dtype, target, target_dtype, mask, \
                             normalize, rowstat, coef, dp_extra, dp_dtype, dfeat, dfeat_dtype, n, channels)
  switch (it) {
    case 1: HD_BWD(1); break; case 2: HD_BWD(2); break; case 3: HD_BWD(3); break; case 4: HD_BWD(4); break;
    case 5: HD_BWD(5); break; case 6: HD_BWD(6); break; case 7: HD_BWD(7); break; default: HD_BWD(8); break;
  }
#undef HD_BWD
  return SS_OK;
}
