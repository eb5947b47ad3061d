// Serialized-window attention, SIMT fp32-math reference implementation (impl = SS_ATTN_SIMT).
// Same C-ABI, indexing and padding semantics as the MFMA kernels in attention_mfma.hip; kept as
// the in-library cross-check and for head dims the MFMA path does not cover.
//
// Semantics (ptv3:172-222 with flash_attn_varlen, ptv3:208-214): window w = padded slots
// [win_start[w], win_start[w+1]); slot p reads row gidx[p] of qkv (n, 3C) laid out
// [q(H,d) | k(H,d) | v(H,d)]; softmax(q k^T * scale) v, no mask, no bias.  Output rows are
// written only for canonical slots (sidx[p] >= 0); borrowed (duplicate-padding) slots take part
// as keys/values and as queries whose result is discarded (feat[inverse], ptv3:216).
// Backward: dqkv rows of canonical slots are written directly; the dK/dV of borrowed slots go
// to a side buffer and are added to their rows by a fix-up kernel (no atomics, deterministic).
#include "common.h"
#include "../../include/scenesplat_hip.h"

#define AT_THREADS 256
#define AT_KT 64  // keys (or queries) staged per LDS tile

template <typename T, int D>
__global__ void __launch_bounds__(AT_THREADS)
k_attn_fwd_simt(const T* __restrict__ qkv, const int32_t* __restrict__ gidx, const int32_t* __restrict__ sidx,
                const int32_t* __restrict__ win_start, T* __restrict__ out, float* __restrict__ lse, int C, int H,
                float scale) {
  __shared__ float Ks[AT_KT][D + 1];
  __shared__ float Vs[AT_KT][D + 1];
  const int w = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int64_t C3 = 3 * (int64_t)C;
  for (int qb = 0; qb < L; qb += AT_THREADS) {
    const int qi = qb + tid;
    const bool active = qi < L;
    float q[D], acc[D];
    float m = -INFINITY, l = 0.f;
    if (active) {
      const T* qp = qkv + (int64_t)gidx[p0 + qi] * C3 + h * D;
#pragma unroll
      for (int d = 0; d < D; ++d) { q[d] = ElemIO<T>::load(qp + d) * scale; acc[d] = 0.f; }
    }
    for (int kt = 0; kt < L; kt += AT_KT) {
      const int nk = min(AT_KT, L - kt);
      __syncthreads();
      for (int e = tid; e < nk * D; e += AT_THREADS) {
        int j = e / D, d = e - j * D;
        const T* kp = qkv + (int64_t)gidx[p0 + kt + j] * C3 + C + h * D + d;
        Ks[j][d] = ElemIO<T>::load(kp);
        Vs[j][d] = ElemIO<T>::load(kp + C);
      }
      __syncthreads();
      if (active) {
        for (int j0 = 0; j0 < nk; j0 += 8) {
          float s[8], mx = m;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            float a = -INFINITY;
            if (j0 + jj < nk) {
              a = 0.f;
#pragma unroll
              for (int d = 0; d < D; ++d) a += q[d] * Ks[j0 + jj][d];
            }
            s[jj] = a; mx = fmaxf(mx, a);
          }
          float alpha = __expf(m - mx);
          l *= alpha;
#pragma unroll
          for (int d = 0; d < D; ++d) acc[d] *= alpha;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            if (j0 + jj < nk) {
              float p = __expf(s[jj] - mx);
              l += p;
#pragma unroll
              for (int d = 0; d < D; ++d) acc[d] += p * Vs[j0 + jj][d];
            }
          }
          m = mx;
        }
      }
    }
    if (active) {
      const int p = p0 + qi;
      lse[(int64_t)p * H + h] = m + __logf(l);
      const int32_t row = sidx[p];
      if (row >= 0) {
        T* op = out + (int64_t)row * C + h * D;
        float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < D; ++d) ElemIO<T>::store(op + d, acc[d] * inv);
      }
    }
  }
}

// delta[p][h] = dO . O over the head's channels for canonical slots, 0 for borrowed slots
template <typename T>
__global__ void k_attn_delta(const T* __restrict__ out, const T* __restrict__ dout, const int32_t* __restrict__ sidx,
                             float* __restrict__ delta, int64_t n_pad, int C, int H) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_pad * H) return;
  int64_t p = gid / H; int h = (int)(gid - p * H);
  int32_t row = sidx[p];
  float s = 0.f;
  if (row >= 0) {
    int D = C / H;
    const T* o = out + (int64_t)row * C + h * D;
    const T* g = dout + (int64_t)row * C + h * D;
    for (int d = 0; d < D; ++d) s += ElemIO<T>::load(o + d) * ElemIO<T>::load(g + d);
  }
  delta[gid] = s;
}

template <typename T, int D>
__global__ void __launch_bounds__(AT_THREADS)
k_attn_bwd_dq_simt(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                   const float* __restrict__ delta, const int32_t* __restrict__ gidx, const int32_t* __restrict__ sidx,
                   const int32_t* __restrict__ win_start, T* __restrict__ dqkv, int C, int H, float scale) {
  __shared__ float Ks[AT_KT][D + 1];
  __shared__ float Vs[AT_KT][D + 1];
  const int w = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int64_t C3 = 3 * (int64_t)C;
  for (int qb = 0; qb < L; qb += AT_THREADS) {
    const int qi = qb + tid;
    const bool active = qi < L;
    float q[D], go[D], dq[D];
    float ls = 0.f, dl = 0.f;
    int32_t row = -1;
    if (active) {
      const int p = p0 + qi;
      row = sidx[p];
      const T* qp = qkv + (int64_t)gidx[p] * C3 + h * D;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        q[d] = ElemIO<T>::load(qp + d) * scale; dq[d] = 0.f;
        go[d] = row >= 0 ? ElemIO<T>::load(dout + (int64_t)row * C + h * D + d) : 0.f;
      }
      ls = lse[(int64_t)p * H + h]; dl = delta[(int64_t)p * H + h];
    }
    for (int kt = 0; kt < L; kt += AT_KT) {
      const int nk = min(AT_KT, L - kt);
      __syncthreads();
      for (int e = tid; e < nk * D; e += AT_THREADS) {
        int j = e / D, d = e - j * D;
        const T* kp = qkv + (int64_t)gidx[p0 + kt + j] * C3 + C + h * D + d;
        Ks[j][d] = ElemIO<T>::load(kp);
        Vs[j][d] = ElemIO<T>::load(kp + C);
      }
      __syncthreads();
      if (active && row >= 0) {
        for (int j = 0; j < nk; ++j) {
          float s = 0.f, dp = 0.f;
#pragma unroll
          for (int d = 0; d < D; ++d) { s += q[d] * Ks[j][d]; dp += go[d] * Vs[j][d]; }
          float ds = __expf(s - ls) * (dp - dl);
#pragma unroll
          for (int d = 0; d < D; ++d) dq[d] += ds * Ks[j][d];
        }
      }
    }
    if (active && row >= 0) {
      T* dp_ = dqkv + (int64_t)row * C3 + h * D;
#pragma unroll
      for (int d = 0; d < D; ++d) ElemIO<T>::store(dp_ + d, dq[d] * scale);
    }
  }
}

template <typename T, int D>
__global__ void __launch_bounds__(AT_THREADS)
k_attn_bwd_dkv_simt(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                    const float* __restrict__ delta, const int32_t* __restrict__ gidx, const int32_t* __restrict__ sidx,
                    const int32_t* __restrict__ win_start, T* __restrict__ dqkv, T* __restrict__ extra, int C, int H,
                    float scale) {
  __shared__ float Qs[AT_KT][D + 1];
  __shared__ float Gs[AT_KT][D + 1];
  __shared__ float Ls[AT_KT], Dl[AT_KT];
  const int w = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int p0 = win_start[w], L = win_start[w + 1] - p0;
  const int64_t C3 = 3 * (int64_t)C;
  for (int kb = 0; kb < L; kb += AT_THREADS) {
    const int kj = kb + tid;
    const bool active = kj < L;
    float k[D], v[D], dk[D], dv[D];
    if (active) {
      const T* kp = qkv + (int64_t)gidx[p0 + kj] * C3 + C + h * D;
#pragma unroll
      for (int d = 0; d < D; ++d) { k[d] = ElemIO<T>::load(kp + d); v[d] = ElemIO<T>::load(kp + C + d); dk[d] = 0.f; dv[d] = 0.f; }
    }
    for (int qt = 0; qt < L; qt += AT_KT) {
      const int nq = min(AT_KT, L - qt);
      __syncthreads();
      for (int e = tid; e < nq * D; e += AT_THREADS) {
        int i = e / D, d = e - i * D;
        const int p = p0 + qt + i;
        const int32_t row = sidx[p];
        Qs[i][d] = ElemIO<T>::load(qkv + (int64_t)gidx[p] * C3 + h * D + d) * scale;
        Gs[i][d] = row >= 0 ? ElemIO<T>::load(dout + (int64_t)row * C + h * D + d) : 0.f;
      }
      if (tid < nq) { Ls[tid] = lse[(int64_t)(p0 + qt + tid) * H + h]; Dl[tid] = delta[(int64_t)(p0 + qt + tid) * H + h]; }
      __syncthreads();
      if (active) {
        for (int i = 0; i < nq; ++i) {
          float s = 0.f, dp = 0.f;
#pragma unroll
          for (int d = 0; d < D; ++d) { s += Qs[i][d] * k[d]; dp += Gs[i][d] * v[d]; }
          float p = __expf(s - Ls[i]);
          float ds = p * (dp - Dl[i]);
#pragma unroll
          for (int d = 0; d < D; ++d) { dv[d] += p * Gs[i][d]; dk[d] += ds * Qs[i][d]; }  // Qs already carries scale
        }
      }
    }
    if (active) {
      const int32_t sr = sidx[p0 + kj];
      T* dkp; T* dvp;
      if (sr >= 0) { dkp = dqkv + (int64_t)sr * C3 + C + h * D; dvp = dkp + C; }
      else { dkp = extra + (int64_t)(-1 - sr) * 2 * C + h * D; dvp = dkp + C; }
#pragma unroll
      for (int d = 0; d < D; ++d) { ElemIO<T>::store(dkp + d, dk[d]); ElemIO<T>::store(dvp + d, dv[d]); }
    }
  }
}

// dqkv[gidx[p]][C:3C] += extra[x] for every borrowed slot p (x = -1 - sidx[p]); also zero dQ is implied
template <typename T>
__global__ void k_attn_fix_borrowed(const int32_t* __restrict__ gidx, const int32_t* __restrict__ sidx, int64_t n_pad,
                                    const T* __restrict__ extra, T* __restrict__ dqkv, int C) {
  int64_t p = blockIdx.x;
  if (p >= n_pad) return;
  int32_t sr = sidx[p];
  if (sr >= 0) return;
  const T* e = extra + (int64_t)(-1 - sr) * 2 * C;
  T* d = dqkv + (int64_t)gidx[p] * 3 * C + C;
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x)
    ElemIO<T>::store(d + c, ElemIO<T>::load(d + c) + ElemIO<T>::load(e + c));
}

template <typename T>
static int launch_fwd(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* ws, int W, void* out,
                      float* lse, int C, int H, float scale, hipStream_t st) {
  dim3 g(W, H), b(AT_THREADS);
  const T* q = (const T*)qkv; T* o = (T*)out;
  switch (C / H) {
    case 16: SS_LAUNCH((k_attn_fwd_simt<T, 16>), g, b, 0, st, q, gidx, sidx, ws, o, lse, C, H, scale); break;
    case 32: SS_LAUNCH((k_attn_fwd_simt<T, 32>), g, b, 0, st, q, gidx, sidx, ws, o, lse, C, H, scale); break;
    case 48: SS_LAUNCH((k_attn_fwd_simt<T, 48>), g, b, 0, st, q, gidx, sidx, ws, o, lse, C, H, scale); break;
    case 64: SS_LAUNCH((k_attn_fwd_simt<T, 64>), g, b, 0, st, q, gidx, sidx, ws, o, lse, C, H, scale); break;
    default: return SS_ERR_ARG;
  }
  return SS_OK;
}

int ss_attn_fwd_simt(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int W,
                     void* out, float* lse, int C, int H, float scale, int dtype, hipStream_t st) {
  int rc = dtype == SS_F32 ? launch_fwd<float>(qkv, gidx, sidx, win_start, W, out, lse, C, H, scale, st)
                           : launch_fwd<unsigned short>(qkv, gidx, sidx, win_start, W, out, lse, C, H, scale, st);
  if (rc) return rc;
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// bf16 fast path: 16-byte loads, one thread per (slot, head); D % 8 == 0
__global__ void k_attn_delta_bf16v(const unsigned short* __restrict__ out, const unsigned short* __restrict__ dout,
                                   const int32_t* __restrict__ sidx, float* __restrict__ delta, int64_t n_pad, int C, int H) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_pad * H) return;
  int64_t p = gid / H; int h = (int)(gid - p * H);
  int32_t row = sidx[p];
  float s = 0.f;
  if (row >= 0) {
    const int D = C / H;
    const uint4* o = reinterpret_cast<const uint4*>(out + (int64_t)row * C + h * D);
    const uint4* g = reinterpret_cast<const uint4*>(dout + (int64_t)row * C + h * D);
    for (int i = 0; i < D / 8; ++i) {
      uint4 a = o[i], b = g[i];
      const unsigned int* ua = reinterpret_cast<const unsigned int*>(&a);
      const unsigned int* ub = reinterpret_cast<const unsigned int*>(&b);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += __uint_as_float(ua[j] << 16) * __uint_as_float(ub[j] << 16);
        s += __uint_as_float(ua[j] & 0xffff0000u) * __uint_as_float(ub[j] & 0xffff0000u);
      }
    }
  }
  delta[gid] = s;
}

template <typename T>
int ss_attn_delta_t(const void* out, const void* dout, const int32_t* sidx, float* delta, int64_t n_pad, int C, int H,
                    hipStream_t st) {
  SS_LAUNCH(k_attn_delta<T>, dim3(ss_div_up(n_pad * H, 256)), dim3(256), 0, st, (const T*)out, (const T*)dout,
                     sidx, delta, n_pad, C, H);
  return SS_OK;
}
int ss_attn_delta(const void* out, const void* dout, const int32_t* sidx, float* delta, int64_t n_pad, int C, int H,
                  int dtype, hipStream_t st) {
  if (dtype == SS_BF16 && ((C / H) & 7) == 0) {
    SS_LAUNCH(k_attn_delta_bf16v, dim3(ss_div_up(n_pad * H, 256)), dim3(256), 0, st, (const unsigned short*)out,
              (const unsigned short*)dout, sidx, delta, n_pad, C, H);
    return SS_OK;
  }
  return dtype == SS_F32 ? ss_attn_delta_t<float>(out, dout, sidx, delta, n_pad, C, H, st)
                         : ss_attn_delta_t<unsigned short>(out, dout, sidx, delta, n_pad, C, H, st);
}
template <typename T>
int ss_attn_fix_t(const int32_t* gidx, const int32_t* sidx, int64_t n_pad, const void* extra, void* dqkv, int C,
                  hipStream_t st) {
  SS_LAUNCH(k_attn_fix_borrowed<T>, dim3((unsigned)n_pad), dim3(256), 0, st, gidx, sidx, n_pad, (const T*)extra,
                     (T*)dqkv, C);
  return SS_OK;
}
int ss_attn_fix_borrowed(const int32_t* gidx, const int32_t* sidx, int64_t n_pad, const void* extra, void* dqkv, int C,
                         int dtype, hipStream_t st) {
  return dtype == SS_F32 ? ss_attn_fix_t<float>(gidx, sidx, n_pad, extra, dqkv, C, st)
                         : ss_attn_fix_t<unsigned short>(gidx, sidx, n_pad, extra, dqkv, C, st);
}

template <typename T>
static int launch_bwd(const void* qkv, const void* dout, const float* lse, const float* delta, const int32_t* gidx,
                      const int32_t* sidx, const int32_t* ws, int W, void* dqkv, void* extra, int C, int H, float scale,
                      hipStream_t st) {
  dim3 g(W, H), b(AT_THREADS);
  const T* q = (const T*)qkv; const T* go = (const T*)dout; T* dq = (T*)dqkv; T* ex = (T*)extra;
#define SS_BWD_CASE(DD)                                                                                              \
  case DD:                                                                                                           \
    SS_LAUNCH((k_attn_bwd_dq_simt<T, DD>), g, b, 0, st, q, go, lse, delta, gidx, sidx, ws, dq, C, H, scale); \
    SS_LAUNCH((k_attn_bwd_dkv_simt<T, DD>), g, b, 0, st, q, go, lse, delta, gidx, sidx, ws, dq, ex, C, H, scale); \
    break;
  switch (C / H) {
    SS_BWD_CASE(16) SS_BWD_CASE(32) SS_BWD_CASE(48) SS_BWD_CASE(64)
    default: return SS_ERR_ARG;
  }
#undef SS_BWD_CASE
  return SS_OK;
}
int ss_attn_bwd_simt(const void* qkv, const void* dout, const float* lse, const float* delta, const int32_t* gidx,
                     const int32_t* sidx, const int32_t* win_start, int W, void* dqkv, void* extra, int C, int H,
                     float scale, int dtype, hipStream_t st) {
  int rc = dtype == SS_F32
               ? launch_bwd<float>(qkv, dout, lse, delta, gidx, sidx, win_start, W, dqkv, extra, C, H, scale, st)
               : launch_bwd<unsigned short>(qkv, dout, lse, delta, gidx, sidx, win_start, W, dqkv, extra, C, H, scale, st);
  if (rc) return rc;
  SS_CHECK_LAUNCH();
  return SS_OK;
}
