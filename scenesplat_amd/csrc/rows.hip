// Row movement kernels (HBM-bound): indexed gather / scatter of feature rows, the serialized
// grid-pool segment reduce and its broadcast backward, and the unpool gather-add.
// Replaces, on the PTv3 path, feat[indices] / feat[inverse] indexing (ptv3:188,216,417,478) and
// torch_scatter.segment_csr (ptv3:416-421).  Every source row and index is read once and every
// destination row written once, 16 bytes per lane where the row width allows.
#include "common.h"
#include "../../include/scenesplat_hip.h"

template <typename V>
__global__ void k_gather_rows(const V* __restrict__ src, const int32_t* __restrict__ idx, V* __restrict__ dst,
                              int64_t n_dst, int chunks) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_dst * chunks) return;
  int64_t r = gid / chunks; int c = (int)(gid - r * chunks);
  int32_t s = idx[r];
  V v; 
  if (s >= 0) v = src[(int64_t)s * chunks + c]; else v = V{};
  dst[gid] = v;
}
template <typename V>
__global__ void k_scatter_rows(const V* __restrict__ src, const int32_t* __restrict__ idx, V* __restrict__ dst,
                               int64_t n_src, int chunks) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_src * chunks) return;
  int64_t r = gid / chunks; int c = (int)(gid - r * chunks);
  int32_t d = idx[r];
  if (d >= 0) dst[(int64_t)d * chunks + c] = src[gid];
}

// dgrad weight of a submanifold conv: wt[ci][T-1-t][co] = w[co][t][ci] (tap-mirrored transpose, bf16), one pass through
// 32 x 32 LDS tiles so both the reads (ci fastest) and the writes (co fastest) are coalesced.
__global__ void k_subm_weight_mirror(const unsigned short* __restrict__ w, unsigned short* __restrict__ wt, int cout, int taps,
                                     int cin) {
  __shared__ unsigned short tile[32][33];
  const int t = blockIdx.z, co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 256 threads: 8 rows per pass
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < cout && ci < cin) ? w[((int64_t)co * taps + t) * cin + ci] : (unsigned short)0;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    int ci = ci0 + r, co = co0 + tx;
    if (ci < cin && co < cout) wt[((int64_t)ci * taps + (taps - 1 - t)) * cout + co] = tile[tx][r];
  }
}

extern "C" int ss_subm_weight_mirror(const void* w, void* wt, int cout, int taps, int cin, hipStream_t stream) {
  if (cout <= 0 || taps <= 0 || cin <= 0 || taps > 65535) return SS_ERR_ARG;
  SS_LAUNCH(k_subm_weight_mirror, dim3(ss_div_up(cin, 32), ss_div_up(cout, 32), taps), dim3(256), 0, stream,
            (const unsigned short*)w, (unsigned short*)wt, cout, taps, cin);
  return SS_OK;
}

// im2col of a submanifold conv: dst[i][t][:] = src[nbr[t][i]][:] (zero row when the neighbour is missing).
// Small levels (n <~ 8k sites) turn the conv into ONE long-K library GEMM instead of a serial 27-tap loop.
__global__ void k_subm_im2col(const uint4* __restrict__ src, const int32_t* __restrict__ nbr, uint4* __restrict__ dst,
                              int64_t n, int taps, int chunks) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * taps * chunks) return;
  int64_t it = gid / chunks; int c = (int)(gid - it * chunks);
  int64_t i = it / taps; int t = (int)(it - i * taps);
  int32_t s = nbr[(int64_t)t * n + i];
  uint4 v = make_uint4(0, 0, 0, 0);
  if (s >= 0) v = src[(int64_t)s * chunks + c];
  dst[gid] = v;
}

extern "C" int ss_subm_im2col(const void* src, const int32_t* nbr, void* dst, int64_t n, int taps, int64_t row_bytes,
                              hipStream_t stream) {
  if (n < 0 || taps <= 0 || row_bytes <= 0 || (row_bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  int chunks = (int)(row_bytes >> 4);
  SS_LAUNCH(k_subm_im2col, dim3(ss_div_up(n * taps * chunks, 256)), dim3(256), 0, stream, (const uint4*)src, nbr, (uint4*)dst,
            n, taps, chunks);
  return SS_OK;
}

// A HIP stream whose kernels may only use the CUs set in `mask` (bit i of word i/32 = CU i).  The deferred
// weight-gradient launches run on such a stream (every other CU) so the latency-bound main chain always finds free CUs.
extern "C" int ss_stream_create_cu_mask(int nwords, const uint32_t* mask, void** stream_out) {
  if (nwords <= 0 || !mask || !stream_out) return SS_ERR_ARG;
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask) != hipSuccess) { (void)hipGetLastError(); return SS_ERR_LAUNCH; }
  *stream_out = (void*)st;
  return SS_OK;
}

extern "C" int ss_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n_dst, int64_t row_bytes,
                              hipStream_t stream) {
  if (n_dst < 0 || row_bytes <= 0 || (row_bytes & 1)) return SS_ERR_ARG;
  if (n_dst == 0) return SS_OK;
  if ((row_bytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    int chunks = (int)(row_bytes >> 4);
    SS_LAUNCH(k_gather_rows<uint4>, dim3(ss_div_up(n_dst * chunks, 256)), dim3(256), 0, stream,
                       (const uint4*)src, idx, (uint4*)dst, n_dst, chunks);
  } else if ((row_bytes & 3) == 0) {
    int chunks = (int)(row_bytes >> 2);
    SS_LAUNCH(k_gather_rows<uint32_t>, dim3(ss_div_up(n_dst * chunks, 256)), dim3(256), 0, stream,
                       (const uint32_t*)src, idx, (uint32_t*)dst, n_dst, chunks);
  } else {
    int chunks = (int)(row_bytes >> 1);
    SS_LAUNCH(k_gather_rows<uint16_t>, dim3(ss_div_up(n_dst * chunks, 256)), dim3(256), 0, stream,
                       (const uint16_t*)src, idx, (uint16_t*)dst, n_dst, chunks);
  }
  SS_CHECK_LAUNCH();
  return SS_OK;
}
extern "C" int ss_scatter_rows(const void* src, const int32_t* idx, void* dst, int64_t n_src, int64_t row_bytes,
                               hipStream_t stream) {
  if (n_src < 0 || row_bytes <= 0 || (row_bytes & 1)) return SS_ERR_ARG;
  if (n_src == 0) return SS_OK;
  if ((row_bytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    int chunks = (int)(row_bytes >> 4);
    SS_LAUNCH(k_scatter_rows<uint4>, dim3(ss_div_up(n_src * chunks, 256)), dim3(256), 0, stream,
                       (const uint4*)src, idx, (uint4*)dst, n_src, chunks);
  } else if ((row_bytes & 3) == 0) {
    int chunks = (int)(row_bytes >> 2);
    SS_LAUNCH(k_scatter_rows<uint32_t>, dim3(ss_div_up(n_src * chunks, 256)), dim3(256), 0, stream,
                       (const uint32_t*)src, idx, (uint32_t*)dst, n_src, chunks);
  } else {
    int chunks = (int)(row_bytes >> 1);
    SS_LAUNCH(k_scatter_rows<uint16_t>, dim3(ss_div_up(n_src * chunks, 256)), dim3(256), 0, stream,
                       (const uint16_t*)src, idx, (uint16_t*)dst, n_src, chunks);
  }
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// out[s][c] = reduce_{j in [ptr[s], ptr[s+1])} src[indices ? indices[j] : j][c]   (fp32 accumulate)
template <typename T>
__global__ void k_segment_reduce(const T* __restrict__ src, const int32_t* __restrict__ indices,
                                 const int32_t* __restrict__ ptr, T* __restrict__ out, int64_t n_seg, int C, int mean) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_seg * C) return;
  int64_t s = gid / C; int c = (int)(gid - s * C);
  int b = ptr[s], e = ptr[s + 1];
  float acc = 0.f;
  for (int j = b; j < e; ++j) {
    int64_t r = indices ? indices[j] : j;
    acc += ElemIO<T>::load(src + r * C + c);
  }
  if (mean && e > b) acc /= (float)(e - b);
  ElemIO<T>::store(out + gid, acc);
}
// out[s][c] = min / max over the segment, arg[s][c] = the source ROW that attains it (first one on ties; -1 and 0 for an
// empty segment, as torch_scatter's base value)
template <typename T>
__global__ void k_segment_minmax(const T* __restrict__ src, const int32_t* __restrict__ indices,
                                 const int32_t* __restrict__ ptr, T* __restrict__ out, int32_t* __restrict__ arg,
                                 int64_t n_seg, int C, int is_max) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_seg * C) return;
  int64_t s = gid / C; int c = (int)(gid - s * C);
  int b = ptr[s], e = ptr[s + 1];
  float best = 0.f; int32_t bi = -1;
  for (int j = b; j < e; ++j) {
    int32_t r = indices ? indices[j] : j;
    float v = ElemIO<T>::load(src + (int64_t)r * C + c);
    if (bi < 0 || (is_max ? v > best : v < best) || v != v) { if (!(best != best)) { best = v; bi = r; } }
  }
  ElemIO<T>::store(out + gid, best);
  if (arg) arg[gid] = bi;
}
// dsrc (zeroed) [arg[s][c]][c] = dout[s][c]
template <typename T>
__global__ void k_segment_minmax_bwd(const T* __restrict__ dout, const int32_t* __restrict__ arg, T* __restrict__ dsrc,
                                     int64_t n_seg, int C) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_seg * C) return;
  int c = (int)(gid % C);
  int32_t r = arg[gid];
  if (r >= 0) dsrc[(int64_t)r * C + c] = dout[gid];
}
// dsrc[i][c] = dout[cluster[i]][c] * (mean ? 1/count : 1)
template <typename T>
__global__ void k_segment_bcast(const T* __restrict__ dout, const int32_t* __restrict__ cluster,
                                const int32_t* __restrict__ ptr, T* __restrict__ dsrc, int64_t n, int C, int mean) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * C) return;
  int64_t i = gid / C; int c = (int)(gid - i * C);
  int s = cluster[i];
  float v = ElemIO<T>::load(dout + (int64_t)s * C + c);
  if (mean) v /= (float)(ptr[s + 1] - ptr[s]);
  ElemIO<T>::store(dsrc + gid, v);
}
// dst[i][c] = a[i][c] + b[idx[i]][c]
template <typename T>
__global__ void k_gather_add(const T* __restrict__ a, const T* __restrict__ b, const int32_t* __restrict__ idx,
                             T* __restrict__ dst, int64_t n, int C) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * C) return;
  int64_t i = gid / C; int c = (int)(gid - i * C);
  float v = ElemIO<T>::load(a + gid) + ElemIO<T>::load(b + (int64_t)idx[i] * C + c);
  ElemIO<T>::store(dst + gid, v);
}

// ---- 16-byte-lane forms of the three kernels the step runs at every pooling / unpooling seam (round 3) -------------------
// One thread per 16-byte chunk of a row (8 bf16 / 4 fp32), fp32 accumulate: every load and store is a dwordx4 and the chunks of a
// row are adjacent lanes (the one-element-per-thread forms above moved 2 bytes per lane: 0.14-0.26 of the HBM roof at dec0,
// profiles/r02_kernel_stats.md).  Used whenever the row is a multiple of 16 bytes and the pointers are 16-byte aligned.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec16<unsigned short> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned int* u = reinterpret_cast<const unsigned int*>(&t);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[8]) {
    uint4 t;
    t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]); t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
  }
};
template <typename T>
__global__ void k_segment_reduce_v(const T* __restrict__ src, const int32_t* __restrict__ indices, const int32_t* __restrict__ ptr,
                                   T* __restrict__ out, int64_t n_seg, int C, int chunks, int mean) {
  constexpr int N = Vec16<T>::N;
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_seg * chunks) return;
  const int64_t s = gid / chunks; const int c = (int)(gid - s * chunks) * N;
  const int b = ptr[s], e = ptr[s + 1];
  float acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.f;
  for (int j = b; j < e; ++j) {
    const int64_t r = indices ? indices[j] : j;
    float v[N];
    Vec16<T>::load(src + r * C + c, v);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] += v[i];
  }
  if (mean && e > b) {
    const float inv = (float)(e - b);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] /= inv;
  }
  Vec16<T>::store(out + s * C + c, acc);
}
template <typename T>
__global__ void k_segment_bcast_v(const T* __restrict__ dout, const int32_t* __restrict__ cluster, const int32_t* __restrict__ ptr,
                                  T* __restrict__ dsrc, int64_t n, int C, int chunks, int mean) {
  constexpr int N = Vec16<T>::N;
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * chunks) return;
  const int64_t i = gid / chunks; const int c = (int)(gid - i * chunks) * N;
  const int s = cluster[i];
  float v[N];
  Vec16<T>::load(dout + (int64_t)s * C + c, v);
  if (mean) {
    const float cnt = (float)(ptr[s + 1] - ptr[s]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] /= cnt;
  }
  Vec16<T>::store(dsrc + i * C + c, v);
}
template <typename T>
__global__ void k_gather_add_v(const T* __restrict__ a, const T* __restrict__ b, const int32_t* __restrict__ idx, T* __restrict__ dst,
                               int64_t n, int C, int chunks) {
  constexpr int N = Vec16<T>::N;
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * chunks) return;
  const int64_t i = gid / chunks; const int c = (int)(gid - i * chunks) * N;
  float x[N], y[N];
  Vec16<T>::load(a + i * C + c, x);
  Vec16<T>::load(b + (int64_t)idx[i] * C + c, y);
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] += y[k];
  Vec16<T>::store(dst + i * C + c, x);
}
static inline bool rows_vec16_ok(const void* a, const void* b, const void* c, int channels, int dtype) {
  const int64_t rb = (int64_t)channels * (dtype == SS_F32 ? 4 : 2);
  return (rb & 15) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

extern "C" int ss_segment_reduce(const void* src, const int32_t* indices, const int32_t* idx_ptr, void* out,
                                 int64_t n_seg, int channels, int dtype, int mean, hipStream_t stream) {
  if (n_seg < 0 || channels <= 0) return SS_ERR_ARG;
  if (n_seg == 0) return SS_OK;
  if (rows_vec16_ok(src, out, nullptr, channels, dtype)) {
    const int chunks = channels / (dtype == SS_F32 ? 4 : 8);
    dim3 gv(ss_div_up(n_seg * chunks, 256)), bv(256);
    if (dtype == SS_F32)
      SS_LAUNCH(k_segment_reduce_v<float>, gv, bv, 0, stream, (const float*)src, indices, idx_ptr, (float*)out, n_seg, channels, chunks, mean);
    else if (dtype == SS_BF16)
      SS_LAUNCH(k_segment_reduce_v<unsigned short>, gv, bv, 0, stream, (const unsigned short*)src, indices, idx_ptr, (unsigned short*)out, n_seg, channels, chunks, mean);
    else return SS_ERR_ARG;
    return SS_OK;
  }
  dim3 g(ss_div_up(n_seg * channels, 256)), b(256);
  if (dtype == SS_F32)
    SS_LAUNCH(k_segment_reduce<float>, g, b, 0, stream, (const float*)src, indices, idx_ptr, (float*)out, n_seg, channels, mean);
  else if (dtype == SS_BF16)
    SS_LAUNCH(k_segment_reduce<unsigned short>, g, b, 0, stream, (const unsigned short*)src, indices, idx_ptr, (unsigned short*)out, n_seg, channels, mean);
  else return SS_ERR_ARG;
  SS_CHECK_LAUNCH();
  return SS_OK;
}
extern "C" int ss_segment_minmax(const void* src, const int32_t* indices, const int32_t* idx_ptr, void* out, int32_t* arg,
                                 int64_t n_seg, int channels, int dtype, int is_max, hipStream_t stream) {
  if (n_seg < 0 || channels <= 0) return SS_ERR_ARG;
  if (n_seg == 0) return SS_OK;
  dim3 g(ss_div_up(n_seg * channels, 256)), b(256);
  if (dtype == SS_F32)
    SS_LAUNCH(k_segment_minmax<float>, g, b, 0, stream, (const float*)src, indices, idx_ptr, (float*)out, arg, n_seg, channels, is_max);
  else if (dtype == SS_BF16)
    SS_LAUNCH(k_segment_minmax<unsigned short>, g, b, 0, stream, (const unsigned short*)src, indices, idx_ptr, (unsigned short*)out, arg, n_seg, channels, is_max);
  else return SS_ERR_ARG;
  return SS_OK;
}
extern "C" int ss_segment_minmax_bwd(const void* dout, const int32_t* arg, void* dsrc, int64_t n_seg, int channels, int dtype,
                                     hipStream_t stream) {
  if (n_seg < 0 || channels <= 0 || !arg) return SS_ERR_ARG;
  if (n_seg == 0) return SS_OK;
  dim3 g(ss_div_up(n_seg * channels, 256)), b(256);
  if (dtype == SS_F32)
    SS_LAUNCH(k_segment_minmax_bwd<float>, g, b, 0, stream, (const float*)dout, arg, (float*)dsrc, n_seg, channels);
  else if (dtype == SS_BF16)
    SS_LAUNCH(k_segment_minmax_bwd<unsigned short>, g, b, 0, stream, (const unsigned short*)dout, arg, (unsigned short*)dsrc, n_seg, channels);
  else return SS_ERR_ARG;
  return SS_OK;
}
extern "C" int ss_segment_bcast(const void* dout, const int32_t* cluster, const int32_t* idx_ptr, void* dsrc, int64_t n,
                                int channels, int dtype, int mean, hipStream_t stream) {
  if (n < 0 || channels <= 0) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  if (rows_vec16_ok(dout, dsrc, nullptr, channels, dtype)) {
    const int chunks = channels / (dtype == SS_F32 ? 4 : 8);
    dim3 gv(ss_div_up(n * chunks, 256)), bv(256);
    if (dtype == SS_F32)
      SS_LAUNCH(k_segment_bcast_v<float>, gv, bv, 0, stream, (const float*)dout, cluster, idx_ptr, (float*)dsrc, n, channels, chunks, mean);
    else if (dtype == SS_BF16)
      SS_LAUNCH(k_segment_bcast_v<unsigned short>, gv, bv, 0, stream, (const unsigned short*)dout, cluster, idx_ptr, (unsigned short*)dsrc, n, channels, chunks, mean);
    else return SS_ERR_ARG;
    return SS_OK;
  }
  dim3 g(ss_div_up(n * channels, 256)), b(256);
  if (dtype == SS_F32)
    SS_LAUNCH(k_segment_bcast<float>, g, b, 0, stream, (const float*)dout, cluster, idx_ptr, (float*)dsrc, n, channels, mean);
  else if (dtype == SS_BF16)
    SS_LAUNCH(k_segment_bcast<unsigned short>, g, b, 0, stream, (const unsigned short*)dout, cluster, idx_ptr, (unsigned short*)dsrc, n, channels, mean);
  else return SS_ERR_ARG;
  SS_CHECK_LAUNCH();
  return SS_OK;
}
extern "C" int ss_gather_add_rows(const void* a, const void* b, const int32_t* idx, void* dst, int64_t n, int channels,
                                  int dtype, hipStream_t stream) {
  if (n < 0 || channels <= 0) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  if (rows_vec16_ok(a, b, dst, channels, dtype)) {
    const int chunks = channels / (dtype == SS_F32 ? 4 : 8);
    dim3 gv(ss_div_up(n * chunks, 256)), bv(256);
    if (dtype == SS_F32)
      SS_LAUNCH(k_gather_add_v<float>, gv, bv, 0, stream, (const float*)a, (const float*)b, idx, (float*)dst, n, channels, chunks);
    else if (dtype == SS_BF16)
      SS_LAUNCH(k_gather_add_v<unsigned short>, gv, bv, 0, stream, (const unsigned short*)a, (const unsigned short*)b, idx, (unsigned short*)dst, n, channels, chunks);
    else return SS_ERR_ARG;
    return SS_OK;
  }
  dim3 g(ss_div_up(n * channels, 256)), bl(256);
  if (dtype == SS_F32)
    SS_LAUNCH(k_gather_add<float>, g, bl, 0, stream, (const float*)a, (const float*)b, idx, (float*)dst, n, channels);
  else if (dtype == SS_BF16)
    SS_LAUNCH(k_gather_add<unsigned short>, g, bl, 0, stream, (const unsigned short*)a, (const unsigned short*)b, idx, (unsigned short*)dst, n, channels);
  else return SS_ERR_ARG;
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// ---- grouped 2-byte transpose: dst (cols, rows) = src (rows, cols)^T for many matrices in ONE launch -----------------------------------
// The dgrad GEMM of nn.Linear, dx = dy @ W, runs 12-17 % faster on hipBLASLt when W arrives as a transposed contiguous copy (its NT
// form; NN 0.41 / 0.45 / 0.50 ms vs NT 0.36 / 0.37 / 0.42 ms at the dec0 shapes, bit-identical results).  The copies are refreshed
// together with the bf16 parameter shadows, all matrices of a model in this one launch.
// desc: 4 int64 words per problem = {src, dst, rows, cols}; wg_start (nprob + 1): first workgroup; a workgroup owns one 64 x 64 tile.
__global__ void __launch_bounds__(256)
k_transpose16_group(const int64_t* __restrict__ desc, const int32_t* __restrict__ wg_start, int nprob) {
  __shared__ unsigned short tile[64][66];
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < nprob && wg_start[p + 1] <= b) ++p;
  const int64_t* d = desc + (int64_t)p * 4;
  const unsigned short* src = reinterpret_cast<const unsigned short*>(d[0]);
  unsigned short* dst = reinterpret_cast<unsigned short*>(d[1]);
  const int rows = (int)d[2], cols = (int)d[3];
  const int tc = (cols + 63) >> 6, t = b - wg_start[p];
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * i][tx] = src[(int64_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (r < rows && c < cols) dst[(int64_t)c * rows + r] = tile[tx][ty + 4 * i];
  }
}

extern "C" int ss_transpose16_group(const int64_t* desc, const int32_t* wg_start, int nprob, int total_workgroups, hipStream_t stream) {
  if (nprob <= 0 || total_workgroups <= 0) return SS_OK;
  if (!desc || !wg_start) return SS_ERR_ARG;
  SS_LAUNCH(k_transpose16_group, dim3((unsigned)total_workgroups), dim3(256), 0, stream, desc, wg_start, nprob);
  return SS_OK;
}

// ---- grouped fp32 -> bf16 cast: the bf16 shadows of ALL GEMM / conv weights of a model in ONE launch (round 3) ----------------
// torch._foreach_copy_ with a dtype change falls back to one copy kernel per tensor: 206 launches and 1.2 ms per step for the
// lang-pretrain model (profiles/r03_kernel_stats.md, first take).  desc: 3 int64 words per tensor = {src f32, dst bf16, numel};
// wg_start (nprob + 1); a workgroup owns 8,192 consecutive elements of one tensor (tensor bases are allocator-aligned).
#define CASTG_PER_WG 8192
__global__ void __launch_bounds__(256)
k_cast_bf16_group(const int64_t* __restrict__ desc, const int32_t* __restrict__ wg_start, int nprob) {
  const int b = blockIdx.x;
  int lo = 0, hi = nprob - 1;                       // last problem whose first workgroup is <= b
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (wg_start[mid] <= b) lo = mid; else hi = mid - 1; }
  const int64_t* d = desc + (int64_t)lo * 3;
  const float* src = reinterpret_cast<const float*>(d[0]);
  unsigned short* dst = reinterpret_cast<unsigned short*>(d[1]);
  const int64_t numel = d[2];
  const int64_t e0 = (int64_t)(b - wg_start[lo]) * CASTG_PER_WG;
#pragma unroll
  for (int i = 0; i < CASTG_PER_WG / (256 * 8); ++i) {
    const int64_t e = e0 + ((int64_t)i * 256 + threadIdx.x) * 8;
    if (e + 8 <= numel) {
      const float4 a = *reinterpret_cast<const float4*>(src + e), c = *reinterpret_cast<const float4*>(src + e + 4);
      uint4 o;
      o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w); o.z = pack_bf16x2(c.x, c.y); o.w = pack_bf16x2(c.z, c.w);
      *reinterpret_cast<uint4*>(dst + e) = o;
    } else {
      for (int64_t k = e; k < numel; ++k) dst[k] = f32_to_bf16(src[k]);
    }
  }
}

extern "C" int ss_cast_bf16_group_elems_per_workgroup(void) { return CASTG_PER_WG; }
extern "C" int ss_cast_bf16_group(const int64_t* desc, const int32_t* wg_start, int nprob, int total_workgroups, hipStream_t stream) {
  if (nprob <= 0 || total_workgroups <= 0) return SS_OK;
  if (!desc || !wg_start) return SS_ERR_ARG;
  SS_LAUNCH(k_cast_bf16_group, dim3((unsigned)total_workgroups), dim3(256), 0, stream, desc, wg_start, nprob);
  return SS_OK;
}

// ---- DropPath row scales of ALL residual seams of a forward in one launch ------------------------------------------------------
// timm DropPath on (n, C) rows (ptv3:333-336): scale[i] = Bernoulli(keep[i]) / keep[i], one draw per row and seam.  The uniform
// draws come from Philox4x32-10 keyed by a 64-bit seed that lives in DEVICE memory (drawn by torch's generator: graph-safe, fresh per
// replay) with the row index as the counter -- torch.rand over the 1.09 M rows of the lang-pretrain model cost 0.21 ms per step
// plus two elementwise launches; this is one ~4 us launch.  Not torch's random stream: the reference's masks are not reproducible
// across libraries either (parity fixtures run with drop_path = 0).
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void __launch_bounds__(256)
k_row_keep_scales(const int64_t* __restrict__ seed, const float* __restrict__ keep, float* __restrict__ out, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;           // four consecutive rows per thread
  if (q * 4 >= n) return;
  const uint64_t sd = (uint64_t)seed[0];
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), 0u, 0u};
  philox4x32_10(c, (uint32_t)sd, (uint32_t)(sd >> 32));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = q * 4 + j;
    if (i < n) {
      const float k = keep[i], u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
      out[i] = u < k ? 1.0f / k : 0.0f;
    }
  }
}

extern "C" int ss_row_keep_scales(const int64_t* seed, const float* keep, float* out, int64_t n, hipStream_t stream) {
  if (n < 0 || (n > 0 && (!seed || !keep || !out))) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  SS_LAUNCH(k_row_keep_scales, dim3((unsigned)ss_div_up(ss_div_up(n, 4), 256)), dim3(256), 0, stream, seed, keep, out, n);
  return SS_OK;
}

// ---- grouped form of k_subm_weight_mirror: the dgrad weights of ALL convs of a model in one launch (refreshed with the bf16 shadows) ----
// desc: 5 int64 words per problem = {w, wt, cout, taps, cin}; wg_start (nprob + 1); a workgroup owns one (tap, WMG_TILE x WMG_TILE) tile.
#define WMG_TILE 64      // 64 x 64 elements: 128-byte rows on both the read and the write side (32 x 32 moved 64-byte half lines: 142 us per refresh)
__global__ void __launch_bounds__(256)
k_subm_weight_mirror_group(const int64_t* __restrict__ desc, const int32_t* __restrict__ wg_start, int nprob) {
  __shared__ unsigned short tile[WMG_TILE][WMG_TILE + 2];
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < nprob && wg_start[p + 1] <= b) ++p;
  const int64_t* d = desc + (int64_t)p * 5;
  const unsigned short* w = reinterpret_cast<const unsigned short*>(d[0]);
  unsigned short* wt = reinterpret_cast<unsigned short*>(d[1]);
  const int cout = (int)d[2], taps = (int)d[3], cin = (int)d[4];
  const int tci = (cin + WMG_TILE - 1) / WMG_TILE, tco = (cout + WMG_TILE - 1) / WMG_TILE;
  int l = b - wg_start[p];
  const int ci0 = (l % tci) * WMG_TILE; l /= tci;
  const int co0 = (l % tco) * WMG_TILE; const int t = l / tco;
  const int tx = threadIdx.x & (WMG_TILE - 1), ty = threadIdx.x / WMG_TILE;
#pragma unroll 4
  for (int r = ty; r < WMG_TILE; r += 256 / WMG_TILE) {
    int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < cout && ci < cin) ? w[((int64_t)co * taps + t) * cin + ci] : (unsigned short)0;
  }
  __syncthreads();
#pragma unroll 4
  for (int r = ty; r < WMG_TILE; r += 256 / WMG_TILE) {
    int ci = ci0 + r, co = co0 + tx;
    if (ci < cin && co < cout) wt[((int64_t)ci * taps + (taps - 1 - t)) * cout + co] = tile[tx][r];
  }
}

extern "C" int ss_subm_weight_mirror_group_tile(void) { return WMG_TILE; }
extern "C" int ss_subm_weight_mirror_group(const int64_t* desc, const int32_t* wg_start, int nprob, int total_workgroups,
                                           hipStream_t stream) {
  if (nprob <= 0 || total_workgroups <= 0) return SS_OK;
  if (!desc || !wg_start) return SS_ERR_ARG;
  SS_LAUNCH(k_subm_weight_mirror_group, dim3((unsigned)total_workgroups), dim3(256), 0, stream, desc, wg_start, nprob);
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Duplicate voxels (Mix3D batches: two samples share one batch element, pointcept/datasets/utils.py:43-47).  Every site of a voxel
// reads the voxel's WINNER row (its lowest row: what the rulebook resolves to), so the adjoint of a submanifold conv first folds the
// gradients of all sites of a voxel onto the winner and leaves the other rows -- which nobody reads -- with a zero gradient.
// The runs come from the plan: sorted_keys (n) = the codes of one curve in sorted order, order (n) = its STABLE argsort, so a run of
// equal keys lists the rows of one voxel in ascending order and its first entry is the winner.  One thread per (sorted position,
// 16-byte chunk): run heads sum their run in fp32 (in run order: deterministic), every other position writes zeros; each destination
// row is written exactly once.
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ void k_dup_fold_rows(const uint4* __restrict__ src, const int64_t* __restrict__ keys, const int32_t* __restrict__ order,
                                uint4* __restrict__ dst, int64_t n, int chunks) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * chunks) return;
  const int64_t p = gid / chunks; const int c = (int)(gid - p * chunks);
  const int64_t key = keys[p];
  const int64_t row = order[p];
  uint4 v = make_uint4(0, 0, 0, 0);
  if (p == 0 || keys[p - 1] != key) {
    v = src[row * chunks + c];
    if (p + 1 < n && keys[p + 1] == key) {
      float a[8];
      if (BF16) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[2 * e] = __uint_as_float(w[e] << 16); a[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
      } else {
        a[0] = __uint_as_float(v.x); a[1] = __uint_as_float(v.y); a[2] = __uint_as_float(v.z); a[3] = __uint_as_float(v.w);
      }
      for (int64_t q = p + 1; q < n && keys[q] == key; ++q) {
        const uint4 u = src[(int64_t)order[q] * chunks + c];
        if (BF16) {
          const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[2 * e] += __uint_as_float(w[e] << 16); a[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        } else {
          a[0] += __uint_as_float(u.x); a[1] += __uint_as_float(u.y); a[2] += __uint_as_float(u.z); a[3] += __uint_as_float(u.w);
        }
      }
      if (BF16) { v.x = pack_bf16x2(a[0], a[1]); v.y = pack_bf16x2(a[2], a[3]); v.z = pack_bf16x2(a[4], a[5]); v.w = pack_bf16x2(a[6], a[7]); }
      else { v.x = __float_as_uint(a[0]); v.y = __float_as_uint(a[1]); v.z = __float_as_uint(a[2]); v.w = __float_as_uint(a[3]); }
    }
  }
  dst[row * chunks + c] = v;
}

// rows of x that are not the winner of their voxel := 0 (in place)
__global__ void k_dup_zero_rows(const int64_t* __restrict__ keys, const int32_t* __restrict__ order, uint4* __restrict__ x, int64_t n,
                                int chunks) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * chunks) return;
  const int64_t p = gid / chunks; const int c = (int)(gid - p * chunks);
  if (p > 0 && keys[p - 1] == keys[p]) x[(int64_t)order[p] * chunks + c] = make_uint4(0, 0, 0, 0);
}

extern "C" int ss_dup_fold_rows(const void* src, const int64_t* sorted_keys, const int32_t* order, void* dst, int64_t n, int channels,
                                int dtype, hipStream_t stream) {
  const int64_t row_bytes = (int64_t)channels * (dtype == SS_BF16 ? 2 : 4);
  if (n < 0 || n >= (1LL << 31) || channels <= 0 || (dtype != SS_F32 && dtype != SS_BF16) || (row_bytes & 15) ||
      (((uintptr_t)src | (uintptr_t)dst) & 15) || src == dst)
    return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  const int chunks = (int)(row_bytes >> 4);
  if (dtype == SS_BF16)
    SS_LAUNCH(k_dup_fold_rows<true>, dim3(ss_div_up(n * chunks, 256)), dim3(256), 0, stream, (const uint4*)src, sorted_keys, order,
              (uint4*)dst, n, chunks);
  else
    SS_LAUNCH(k_dup_fold_rows<false>, dim3(ss_div_up(n * chunks, 256)), dim3(256), 0, stream, (const uint4*)src, sorted_keys, order,
              (uint4*)dst, n, chunks);
  return SS_OK;
}

extern "C" int ss_dup_zero_rows(const int64_t* sorted_keys, const int32_t* order, void* x, int64_t n, int64_t row_bytes,
                                hipStream_t stream) {
  if (n < 0 || n >= (1LL << 31) || row_bytes <= 0 || (row_bytes & 15) || ((uintptr_t)x & 15)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  const int chunks = (int)(row_bytes >> 4);
  SS_LAUNCH(k_dup_zero_rows, dim3(ss_div_up(n * chunks, 256)), dim3(256), 0, stream, sorted_keys, order, (uint4*)x, n, chunks);
  return SS_OK;
}
