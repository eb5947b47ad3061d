#pragma once
#include "common.h"
#define SS_ATTN_MFMA_MAX_WINDOW 2048   // = FA_IDX_CAP of attention_mfma.hip; longer windows run on the SIMT kernels
int ss_attn_fwd_simt(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int W,
                     void* out, float* lse, int C, int H, float scale, int dtype, hipStream_t st);
int ss_attn_delta(const void* out, const void* dout, const int32_t* sidx, float* delta, int64_t n_pad, int C, int H,
                  int dtype, hipStream_t st);
int ss_attn_fix_borrowed(const int32_t* gidx, const int32_t* sidx, int64_t n_pad, const void* extra, void* dqkv, int C,
                         int dtype, hipStream_t st);
int ss_attn_bwd_simt(const void* qkv, const void* dout, const float* lse, const float* delta, const int32_t* gidx,
                     const int32_t* sidx, const int32_t* win_start, int W, void* dqkv, void* extra, int C, int H,
                     float scale, int dtype, hipStream_t st);
int ss_attn_fwd_mfma(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int W,
                     int max_window, void* out, float* lse, int C, int H, float scale, hipStream_t st);
// delta (n_pad, H) is an OUTPUT of the dQ kernel here (rowsum(out o dout), consumed by the dK/dV kernel): no ss_attn_delta pass
int ss_attn_bwd_mfma(const void* qkv, const void* dout, const void* out, const float* lse, float* delta, const int32_t* gidx,
                     const int32_t* sidx, const int32_t* win_start, int W, int max_window, void* dqkv, void* extra,
                     int C, int H, float scale, hipStream_t st);
// 32x32x16 re-tiling of the forward (attention_mfma32.hip); same contract as ss_attn_fwd_mfma
int ss_attn_fwd_mfma32(const void* qkv, const int32_t* gidx, const int32_t* sidx, const int32_t* win_start, int W,
                       int max_window, void* out, float* lse, int C, int H, float scale, hipStream_t st);
