// C-ABI entry points of the serialized-window attention; dispatches to the SIMT reference
// kernels (attention_simt.hip) or the MFMA kernels (attention_mfma.hip).
#include "attention_internal.h"
#include "../../include/scenesplat_hip.h"

extern "C" int ss_version(void) { return 100; }

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int ss_window_attn_fwd(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* win_start,
                                  int num_windows, int max_window, int64_t n, int64_t n_pad, int channels,
                                  int num_heads, float scale, int dtype, int impl, void* out, float* lse,
                                  hipStream_t stream) {
  if (num_windows < 0 || channels <= 0 || num_heads <= 0 || channels % num_heads || n_pad < n) return SS_ERR_ARG;
  if (dtype != SS_F32 && dtype != SS_BF16) return SS_ERR_ARG;
  if (num_windows == 0) return SS_OK;
  // the MFMA kernels keep the window's row offsets (16-byte units, 32 bits) in LDS
  if (impl == SS_ATTN_MFMA && (max_window > SS_ATTN_MFMA_MAX_WINDOW || n * (int64_t)(3 * channels / 8) >= (1LL << 31))) impl = SS_ATTN_SIMT;
  if (impl == SS_ATTN_SIMT)
    return ss_attn_fwd_simt(qkv, gidx, sidx, win_start, num_windows, out, lse, channels, num_heads, scale, dtype, stream);
  if (impl == SS_ATTN_MFMA && dtype == SS_BF16)
    return ss_attn_fwd_mfma(qkv, gidx, sidx, win_start, num_windows, max_window, out, lse, channels, num_heads, scale, stream);
  return SS_ERR_ARG;
}

extern "C" size_t ss_window_attn_bwd_workspace_bytes(int64_t n, int64_t n_pad, int channels, int num_heads, int dtype) {
  size_t es = dtype == SS_F32 ? 4 : 2;
  return al256((size_t)n_pad * num_heads * 4) + al256((size_t)(n_pad - n) * 2 * channels * es);
}

extern "C" int ss_window_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                  const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int num_windows,
                                  int max_window, int64_t n, int64_t n_pad, int channels, int num_heads, float scale,
                                  int dtype, int impl, void* dqkv, void* workspace, size_t workspace_bytes,
                                  hipStream_t stream) {
  if (num_windows < 0 || channels <= 0 || num_heads <= 0 || channels % num_heads || n_pad < n) return SS_ERR_ARG;
  if (dtype != SS_F32 && dtype != SS_BF16) return SS_ERR_ARG;
  if (workspace_bytes < ss_window_attn_bwd_workspace_bytes(n, n_pad, channels, num_heads, dtype)) return SS_ERR_WORKSPACE;
  if (num_windows == 0) return SS_OK;
  float* delta = (float*)workspace;
  void* extra = (char*)workspace + al256((size_t)n_pad * num_heads * 4);
  int rc = SS_OK;
  if (impl == SS_ATTN_MFMA && (max_window > SS_ATTN_MFMA_MAX_WINDOW || n * (int64_t)(3 * channels / 8) >= (1LL << 31))) impl = SS_ATTN_SIMT;
  if (!(impl == SS_ATTN_MFMA && dtype == SS_BF16))      // the MFMA dQ kernel computes delta itself
    rc = ss_attn_delta(out, dout, sidx, delta, n_pad, channels, num_heads, dtype, stream);
  if (rc) return rc;
  if (impl == SS_ATTN_SIMT)
    rc = ss_attn_bwd_simt(qkv, dout, lse, delta, gidx, sidx, win_start, num_windows, dqkv, extra, channels, num_heads,
                          scale, dtype, stream);
  else if (impl == SS_ATTN_MFMA && dtype == SS_BF16)
    rc = ss_attn_bwd_mfma(qkv, dout, out, lse, delta, gidx, sidx, win_start, num_windows, max_window, dqkv, extra, channels,
                          num_heads, scale, stream);
  else
    return SS_ERR_ARG;
  if (rc) return rc;
  if (n_pad > n) rc = ss_attn_fix_borrowed(gidx, sidx, n_pad, extra, dqkv, channels, dtype, stream);
  if (rc) return rc;
  SS_CHECK_LAUNCH();
  return SS_OK;
}
