// Integer structure kernels of the PTv3 hot path (gfx950): space-filling-curve keys,
// LSD radix argsort, grid-pool partition, window (patch) index, submanifold rulebook.
// Replaces the pure-torch int64 op chains of
//   pointcept/models/utils/serialization/{default,z_order,hilbert}.py
//   pointcept/models/utils/structure.py:47-102 (Point.serialization)
//   point_transformer_v3m1_base.py:114-170 (padding) and :384-398 (pool partition)
// and the spconv indice-pair build behind SubMConv3d (ptv3:278-284, 499-506).
// All work is HBM/L2-bound integer arithmetic: coalesced 8/16-byte accesses, one thread per
// point, LDS only for the per-tile digit ranking of the radix sort.
#include "common.h"
#include "../../include/scenesplat_hip.h"

// ------------------------------------------------------------------------------------------
// curve keys
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread3(uint64_t x) {  // 21 bits -> every 3rd bit
  x &= 0x1fffffULL;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
}
__device__ __forceinline__ uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  return (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
}
// Skilling transpose -> Hilbert index (hilbert.py:156-181)
__device__ __forceinline__ uint64_t hilbert3(uint32_t x, uint32_t y, uint32_t z, int depth) {
  if (depth <= 0) return 0;
  uint32_t X0 = x, X1 = y, X2 = z;
  for (uint32_t q = 1u << (depth - 1); q > 1; q >>= 1) {
    uint32_t p = q - 1;
    if (X0 & q) X0 ^= p;
    if (X1 & q) X0 ^= p; else { uint32_t t = (X0 ^ X1) & p; X0 ^= t; X1 ^= t; }
    if (X2 & q) X0 ^= p; else { uint32_t t = (X0 ^ X2) & p; X0 ^= t; X2 ^= t; }
  }
  uint64_t h = morton3(X0, X1, X2);
  for (int s = 1; s < 3 * depth; s <<= 1) h ^= h >> s;
  return h;
}

struct EncodeArgs { int order[SS_MAX_ORDERS]; int num_orders; };

__global__ void k_encode(const int32_t* __restrict__ gc, const int32_t* __restrict__ batch, int64_t n,
                         int depth, EncodeArgs a, int64_t* __restrict__ codes) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)gc[3 * i], y = (uint32_t)gc[3 * i + 1], z = (uint32_t)gc[3 * i + 2];
  uint64_t b = batch ? ((uint64_t)(uint32_t)batch[i]) << (3 * depth) : 0;
  for (int k = 0; k < a.num_orders; ++k) {
    uint64_t key;
    switch (a.order[k]) {
      case SS_ORDER_Z: key = morton3(x, y, z); break;
      case SS_ORDER_Z_TRANS: key = morton3(y, x, z); break;
      case SS_ORDER_HILBERT: key = hilbert3(x, y, z, depth); break;
      default: key = hilbert3(y, x, z, depth); break;
    }
    if (depth < 21) key &= (1ULL << (3 * depth)) - 1;
    codes[(int64_t)k * n + i] = (int64_t)(b | key);
  }
}

extern "C" int ss_serialize_encode(const int32_t* grid_coord, const int32_t* batch, int64_t n, int depth,
                                   const int* orders, int num_orders, int64_t* codes, hipStream_t stream) {
  if (num_orders < 1 || num_orders > SS_MAX_ORDERS || depth < 0 || depth > 16 || n < 0) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  EncodeArgs a; a.num_orders = num_orders;
  for (int k = 0; k < num_orders; ++k) {
    if (orders[k] < 0 || orders[k] > 3) return SS_ERR_ARG;
    a.order[k] = orders[k];
  }
  SS_LAUNCH(k_encode, dim3(ss_div_up(n, 256)), dim3(256), 0, stream, grid_coord, batch, n, depth, a, codes);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

__global__ void k_grid_max(const int32_t* __restrict__ gc, int64_t n3, int32_t* out) {
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, gc[i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  // one global atomic per workgroup (4,096 same-address atomics, one per wave of 1,024 workgroups, took 50 us)
  __shared__ int wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, max(max(wm[0], wm[1]), max(wm[2], wm[3])));
}
extern "C" int ss_grid_coord_max(const int32_t* grid_coord, int64_t n, int32_t* out_max, hipStream_t stream) {
  if (n < 0) return SS_ERR_ARG;
  hipMemsetAsync(out_max, 0, sizeof(int32_t), stream);
  if (n == 0) return SS_OK;
  int blocks = min(256, ss_div_up(n * 3, 256));
  SS_LAUNCH(k_grid_max, dim3(blocks), dim3(256), 0, stream, grid_coord, n * 3, out_max);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// ------------------------------------------------------------------------------------------
// LSD radix argsort, 8-bit digits, stable; K independent segments of n keys each.
// per pass: tile histogram -> digit-major exclusive scan -> ranked scatter.
// ------------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ITEMS 8
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_STRIDE(num_tiles) (((num_tiles) + 3) & ~3)      // counters per digit row of the tile histogram

__global__ void k_rs_hist(const int64_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ tile_hist,
                          int num_tiles) {
  __shared__ uint32_t h[256];
  const int k = blockIdx.y, tile = blockIdx.x;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t* kk = keys + (int64_t)k * n;
  int64_t base = (int64_t)tile * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    int64_t i = base + r * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[((uint64_t)kk[i] >> shift) & 255], 1u);
  }
  __syncthreads();
  tile_hist[((int64_t)k * 256 + threadIdx.x) * RS_STRIDE(num_tiles) + tile] = h[threadIdx.x];
}

// digit-major rows of `stride` = num_tiles rounded up to 4 counters (16-byte loads; the pad entries are never written and are masked
// here).  One thread per digit walks its row: with one 4-byte load per step this was 2 x 50 dependent-ish loads on ONE workgroup per
// segment, 19 us per pass and 0.36 ms of every plan build; four counters per load and the loads of a row unrolled cut it to a third.
__global__ void k_rs_scan(uint32_t* __restrict__ tile_hist, int num_tiles) {
  __shared__ uint32_t tot[256];
  const int n4 = RS_STRIDE(num_tiles) >> 2;
  uint4* row = reinterpret_cast<uint4*>(tile_hist + ((int64_t)blockIdx.x * 256 + threadIdx.x) * RS_STRIDE(num_tiles));
  uint32_t s = 0;
#pragma unroll 8
  for (int q = 0; q < n4; ++q) {
    const uint4 c = row[q];
    const int t = q * 4;
    s += c.x + (t + 1 < num_tiles ? c.y : 0u) + (t + 2 < num_tiles ? c.z : 0u) + (t + 3 < num_tiles ? c.w : 0u);
  }
  tot[threadIdx.x] = s;
  __syncthreads();
  // exclusive scan of 256 digit totals (Hillis-Steele)
  uint32_t v = s;
  for (int o = 1; o < 256; o <<= 1) {
    uint32_t add = (threadIdx.x >= (unsigned)o) ? tot[threadIdx.x - o] : 0;
    __syncthreads();
    v += add; tot[threadIdx.x] = v;
    __syncthreads();
  }
  uint32_t run = v - s;
#pragma unroll 8
  for (int q = 0; q < n4; ++q) {
    const uint4 c = row[q];
    const int t = q * 4;
    uint4 o;
    o.x = run; run += c.x;
    o.y = run; run += (t + 1 < num_tiles ? c.y : 0u);
    o.z = run; run += (t + 2 < num_tiles ? c.z : 0u);
    o.w = run; run += (t + 3 < num_tiles ? c.w : 0u);
    row[q] = o;
  }
}

__global__ void k_rs_scatter(const int64_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                             int64_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                             int32_t* __restrict__ inverse_out, int64_t n, int shift,
                             const uint32_t* __restrict__ tile_hist, int num_tiles) {
  __shared__ uint32_t digit_base[256];
  __shared__ uint32_t wave_cnt[RS_THREADS / 64][256];
  const int k = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int64_t seg = (int64_t)k * n;
  digit_base[tid] = tile_hist[((int64_t)k * 256 + tid) * RS_STRIDE(num_tiles) + tile];
  const uint64_t lt_mask = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  int64_t base = (int64_t)tile * RS_TILE;
  for (int r = 0; r < RS_ITEMS; ++r) {
#pragma unroll
    for (int j = 0; j < RS_THREADS / 64; ++j) wave_cnt[j][tid] = 0;
    __syncthreads();
    int64_t i = base + r * RS_THREADS + tid;
    bool valid = i < n;
    int64_t key = valid ? keys_in[seg + i] : 0;
    int32_t val = valid ? (vals_in ? vals_in[seg + i] : (int32_t)i) : 0;
    uint32_t d = ((uint64_t)key >> shift) & 255;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      uint64_t m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    uint32_t rank = __popcll(peers & lt_mask);
    if (valid && rank == 0) wave_cnt[w][d] = __popcll(peers);
    __syncthreads();
    {  // thread tid owns digit tid: exclusive prefix over waves, advance the running base
      uint32_t run = digit_base[tid];
#pragma unroll
      for (int j = 0; j < RS_THREADS / 64; ++j) { uint32_t c = wave_cnt[j][tid]; wave_cnt[j][tid] = run; run += c; }
      digit_base[tid] = run;
    }
    __syncthreads();
    if (valid) {
      uint32_t pos = wave_cnt[w][d] + rank;
      keys_out[seg + pos] = key;
      vals_out[seg + pos] = val;
      if (inverse_out) inverse_out[seg + val] = (int32_t)pos;
    }
    __syncthreads();
  }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t ss_argsort_workspace_bytes(int64_t n, int num_segments) {
  int num_tiles = ss_div_up(n > 0 ? n : 1, RS_TILE);
  return align256((size_t)num_segments * n * 8) * 2 + align256((size_t)num_segments * n * 4) * 2 +
         align256((size_t)num_segments * 256 * RS_STRIDE(num_tiles) * 4);
}

extern "C" int ss_argsort_i64(const int64_t* keys, int num_segments, int64_t n, int key_bits, int32_t* order_out,
                              int32_t* inverse_out, int64_t* sorted_keys_out, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
  if (n < 0 || num_segments < 1 || key_bits < 1 || key_bits > 64 || n >= (1LL << 31)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  if (workspace_bytes < ss_argsort_workspace_bytes(n, num_segments)) return SS_ERR_WORKSPACE;
  const int num_tiles = ss_div_up(n, RS_TILE);
  char* ws = (char*)workspace;
  int64_t* kbuf[2]; int32_t* vbuf[2];
  kbuf[0] = (int64_t*)ws; ws += align256((size_t)num_segments * n * 8);
  kbuf[1] = (int64_t*)ws; ws += align256((size_t)num_segments * n * 8);
  vbuf[0] = (int32_t*)ws; ws += align256((size_t)num_segments * n * 4);
  vbuf[1] = (int32_t*)ws; ws += align256((size_t)num_segments * n * 4);
  uint32_t* hist = (uint32_t*)ws;
  const int passes = (key_bits + 7) / 8;
  const int64_t* kin = keys; const int32_t* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const bool last = (p == passes - 1);
    int64_t* kout = last && sorted_keys_out ? sorted_keys_out : kbuf[p & 1];
    int32_t* vout = last ? order_out : vbuf[p & 1];
    SS_LAUNCH(k_rs_hist, dim3(num_tiles, num_segments), dim3(RS_THREADS), 0, stream, kin, n, p * 8, hist, num_tiles);
    SS_LAUNCH(k_rs_scan, dim3(num_segments), dim3(256), 0, stream, hist, num_tiles);
    SS_LAUNCH(k_rs_scatter, dim3(num_tiles, num_segments), dim3(RS_THREADS), 0, stream, kin, vin, kout, vout,
                       last ? inverse_out : (int32_t*)nullptr, n, p * 8, hist, num_tiles);
    kin = kout; vin = vout;
  }
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// ------------------------------------------------------------------------------------------
// device-wide exclusive scan of uint32 (3 phases), used by the pool partition
// ------------------------------------------------------------------------------------------
#define SC_THREADS 256
#define SC_ITEMS 8
#define SC_TILE (SC_THREADS * SC_ITEMS)

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* sh /*[SC_THREADS]*/, uint32_t* total) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  uint32_t acc = v;
  for (int o = 1; o < SC_THREADS; o <<= 1) {
    uint32_t add = (tid >= o) ? sh[tid - o] : 0;
    __syncthreads();
    acc += add; sh[tid] = acc;
    __syncthreads();
  }
  if (total) *total = sh[SC_THREADS - 1];
  return acc - v;
}

__global__ void k_scan_tile_sums(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t sh[SC_THREADS];
  int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) if (base + j < n) s += in[base + j];
  uint32_t tot;
  block_exclusive_scan(s, sh, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ void k_scan_tile_prefix(uint32_t* __restrict__ tile_sums, int num_tiles) {  // single block
  __shared__ uint32_t sh[SC_THREADS];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int b = 0; b < num_tiles; b += SC_THREADS) {
    int i = b + threadIdx.x;
    uint32_t v = i < num_tiles ? tile_sums[i] : 0, tot;
    uint32_t ex = block_exclusive_scan(v, sh, &tot);
    uint32_t carry = carry_s;
    if (i < num_tiles) tile_sums[i] = ex + carry;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
}
__global__ void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n,
                             const uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t sh[SC_THREADS];
  int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
  uint32_t v[SC_ITEMS], s = 0;
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) { v[j] = (base + j < n) ? in[base + j] : 0; s += v[j]; }
  uint32_t ex = block_exclusive_scan(s, sh, nullptr) + tile_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) { if (base + j < n) out[base + j] = ex; ex += v[j]; }
}
static int device_exclusive_scan(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* tile_sums, hipStream_t stream) {
  int tiles = ss_div_up(n, SC_TILE);
  SS_LAUNCH(k_scan_tile_sums, dim3(tiles), dim3(SC_THREADS), 0, stream, in, n, tile_sums);
  SS_LAUNCH(k_scan_tile_prefix, dim3(1), dim3(SC_THREADS), 0, stream, tile_sums, tiles);
  SS_LAUNCH(k_scan_apply, dim3(tiles), dim3(SC_THREADS), 0, stream, in, out, n, tile_sums);
  return SS_OK;
}

// ------------------------------------------------------------------------------------------
// grid-pool partition (ptv3:384-398): clusters = runs of equal (code0 >> shift) in sorted order
// ------------------------------------------------------------------------------------------
__global__ void k_pp_flags(const int64_t* __restrict__ code0, const int32_t* __restrict__ order0, int64_t n, int shift,
                           uint32_t* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t c = code0[order0[i]] >> shift;
  flags[i] = (i == 0 || (code0[order0[i - 1]] >> shift) != c) ? 1u : 0u;
}
__global__ void k_pp_scatter(const int32_t* __restrict__ order0, const uint32_t* __restrict__ flags,
                             const uint32_t* __restrict__ excl, int64_t n, int32_t* __restrict__ cluster,
                             int32_t* __restrict__ idx_ptr, int32_t* __restrict__ head, int32_t* __restrict__ n_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t f = flags[i];
  int32_t cid = (int32_t)(excl[i] + f) - 1;
  int32_t row = order0[i];
  cluster[row] = cid;
  if (f) { idx_ptr[cid] = (int32_t)i; head[cid] = row; }
  if (i == n - 1) { idx_ptr[cid + 1] = (int32_t)n; *n_out = cid + 1; }
}

extern "C" size_t ss_pool_partition_workspace_bytes(int64_t n) {
  return align256((size_t)n * 4) * 2 + align256((size_t)(ss_div_up(n > 0 ? n : 1, SC_TILE) + 1) * 4);
}

extern "C" int ss_pool_partition(const int64_t* code0, const int32_t* order0, int64_t n, int shift_bits,
                                 int32_t* cluster, int32_t* idx_ptr, int32_t* head, int32_t* n_out, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
  if (n <= 0 || shift_bits < 0 || shift_bits > 62) return SS_ERR_ARG;
  if (workspace_bytes < ss_pool_partition_workspace_bytes(n)) return SS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  uint32_t* flags = (uint32_t*)ws; ws += align256((size_t)n * 4);
  uint32_t* excl = (uint32_t*)ws; ws += align256((size_t)n * 4);
  uint32_t* tile_sums = (uint32_t*)ws;
  int blocks = ss_div_up(n, 256);
  SS_LAUNCH(k_pp_flags, dim3(blocks), dim3(256), 0, stream, code0, order0, n, shift_bits, flags);
  device_exclusive_scan(flags, excl, n, tile_sums, stream);
  SS_LAUNCH(k_pp_scatter, dim3(blocks), dim3(256), 0, stream, order0, flags, excl, n, cluster, idx_ptr, head, n_out);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// pooled level attributes: grid_coord[head] >> d, batch[head], code[:, head] >> 3d  (ptv3:398,422,427)
__global__ void k_pool_level(const int32_t* __restrict__ head, int64_t n_out, int64_t n_in,
                             const int32_t* __restrict__ gc, const int32_t* __restrict__ batch,
                             const int64_t* __restrict__ codes, int num_orders, int pool_depth,
                             int32_t* __restrict__ gc_out, int32_t* __restrict__ batch_out, int64_t* __restrict__ codes_out) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_out) return;
  int32_t h = head[c];
  gc_out[3 * c] = gc[3 * h] >> pool_depth;
  gc_out[3 * c + 1] = gc[3 * h + 1] >> pool_depth;
  gc_out[3 * c + 2] = gc[3 * h + 2] >> pool_depth;
  batch_out[c] = batch[h];
  for (int k = 0; k < num_orders; ++k) codes_out[(int64_t)k * n_out + c] = codes[(int64_t)k * n_in + h] >> (3 * pool_depth);
}
extern "C" int ss_pool_level_attrs(const int32_t* head, int64_t n_out, int64_t n_in, const int32_t* grid_coord,
                                   const int32_t* batch, const int64_t* codes, int num_orders, int pool_depth,
                                   int32_t* grid_coord_out, int32_t* batch_out, int64_t* codes_out, hipStream_t stream) {
  if (n_out < 0 || n_in < n_out || num_orders < 1 || pool_depth < 0) return SS_ERR_ARG;
  if (n_out == 0) return SS_OK;
  SS_LAUNCH(k_pool_level, dim3(ss_div_up(n_out, 256)), dim3(256), 0, stream, head, n_out, n_in, grid_coord,
                     batch, codes, num_orders, pool_depth, grid_coord_out, batch_out, codes_out);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// offsets[b] = #rows with batch id <= b, for a batch array that is non-decreasing along `perm`
__global__ void k_batch_offsets(const int32_t* __restrict__ batch, const int32_t* __restrict__ perm, int64_t n,
                                int num_batches, int32_t* __restrict__ offsets) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= num_batches) return;
  int64_t lo = 0, hi = n;  // first position with batch > b
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    int32_t v = batch[perm ? perm[mid] : mid];
    if (v <= b) lo = mid + 1; else hi = mid;
  }
  offsets[b] = (int32_t)lo;
}
extern "C" int ss_batch_offsets(const int32_t* batch, const int32_t* perm, int64_t n, int num_batches, int32_t* offsets,
                                hipStream_t stream) {
  if (n < 0 || num_batches < 1) return SS_ERR_ARG;
  SS_LAUNCH(k_batch_offsets, dim3(ss_div_up(num_batches, 64)), dim3(64), 0, stream, batch, perm, n, num_batches, offsets);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// ------------------------------------------------------------------------------------------
// window (patch) index (ptv3:114-170, 184-185): for padded position p of batch element e,
// local q = p - off_pad[e]: row = order[off[e] + (q < c ? q : q - K)]; q >= c are the slots
// borrowed from the previous window.  sidx[p] = row for canonical slots, -1-x for the x-th
// borrowed slot (x indexes the side buffer the backward uses for dK/dV of borrowed rows).
// ------------------------------------------------------------------------------------------
__global__ void k_window_index(const int32_t* __restrict__ order, const int32_t* __restrict__ off /*B+1*/,
                               const int32_t* __restrict__ off_pad /*B+1*/, int num_batches, int patch,
                               int32_t* __restrict__ gidx, int32_t* __restrict__ sidx) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n_pad = off_pad[num_batches];
  if (p >= n_pad) return;
  int lo = 0, hi = num_batches;  // find e with off_pad[e] <= p < off_pad[e+1]
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (off_pad[mid] <= p) lo = mid; else hi = mid; }
  int e = lo;
  int q = (int)(p - off_pad[e]);
  int c = off[e + 1] - off[e];
  bool canon = q < c;
  int32_t row = order[off[e] + (canon ? q : q - patch)];
  gidx[p] = row;
  sidx[p] = canon ? row : -1 - ((off_pad[e] - off[e]) + (q - c));
}
extern "C" int ss_window_index(const int32_t* order, const int32_t* offsets, const int32_t* offsets_pad, int num_batches,
                               int patch_size, int64_t n_pad, int32_t* gidx, int32_t* sidx, hipStream_t stream) {
  if (num_batches < 1 || patch_size < 1 || n_pad < 0) return SS_ERR_ARG;
  if (n_pad == 0) return SS_OK;
  SS_LAUNCH(k_window_index, dim3(ss_div_up(n_pad, 256)), dim3(256), 0, stream, order, offsets, offsets_pad,
                     num_batches, patch_size, gidx, sidx);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// ------------------------------------------------------------------------------------------
// submanifold rulebook: nbr[i][t] = row of the site at grid(i)+delta_t in the same batch element
// (or -1).  Lookup = lower_bound in the z-order keys sorted by the serialization (stable, so
// duplicate voxels resolve to the lowest row index).  Tap t = (ix*k+iy)*k+iz over (x,y,z).
// Output layout nbr[t][i] (k^3 rows of n).
// ------------------------------------------------------------------------------------------
__global__ void k_rulebook(const int32_t* __restrict__ gc, const int32_t* __restrict__ batch, int64_t n, int depth,
                           const int64_t* __restrict__ zkeys_sorted, const int32_t* __restrict__ zorder, int swap_xy,
                           int ksize, int32_t* __restrict__ nbr) {
  const int taps = ksize * ksize * ksize;
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * taps) return;
  int t = (int)(gid / n); int64_t i = gid - (int64_t)t * n;   // layout nbr[t][i]: per-tap index rows are contiguous
  int h = ksize >> 1;
  int iz = t % ksize, iy = (t / ksize) % ksize, ix = t / (ksize * ksize);
  int x = gc[3 * i] + ix - h, y = gc[3 * i + 1] + iy - h, z = gc[3 * i + 2] + iz - h;
  int lim = 1 << depth;
  int32_t res = -1;
  if (x >= 0 && y >= 0 && z >= 0 && x < lim && y < lim && z < lim) {
    uint64_t key = swap_xy ? morton3(y, x, z) : morton3(x, y, z);
    key |= ((uint64_t)(uint32_t)batch[i]) << (3 * depth);
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if ((uint64_t)zkeys_sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < n && (uint64_t)zkeys_sorted[lo] == key) res = zorder[lo];
  }
  nbr[gid] = res;
}
extern "C" int ss_subm_rulebook(const int32_t* grid_coord, const int32_t* batch, int64_t n, int depth,
                                const int64_t* zkeys_sorted, const int32_t* zorder, int swap_xy, int kernel_size,
                                int32_t* nbr, hipStream_t stream) {
  if (n < 0 || (kernel_size != 3 && kernel_size != 5) || depth < 0 || depth > 16) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  int64_t total = n * kernel_size * kernel_size * kernel_size;
  SS_LAUNCH(k_rulebook, dim3(ss_div_up(total, 256)), dim3(256), 0, stream, grid_coord, batch, n, depth,
                     zkeys_sorted, zorder, swap_xy, kernel_size, nbr);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// Hashed rulebook: the same table as ss_subm_rulebook from an open-addressing hash of the voxel keys (load <= 0.5,
// linear probing, ~1.3 probes per lookup) instead of a 17-step binary search per (site, tap): the 125-tap stem rulebook
// of a 102,400-site level took ~0.4 ms per step.  Duplicate voxels resolve to the lowest row (atomicMin), as the sorted
// lookup does.  workspace = T x (8 + 4) bytes, T = ss_subm_rulebook_table_size(n).
__device__ __forceinline__ uint32_t rb_hash(uint64_t key, int log2t) { return (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (64 - log2t)); }
__global__ void k_rulebook_insert(const int32_t* __restrict__ gc, const int32_t* __restrict__ batch, int64_t n, int depth,
                                  unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tvals, int log2t) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = morton3(gc[3 * i], gc[3 * i + 1], gc[3 * i + 2]) | (((uint64_t)(uint32_t)batch[i]) << (3 * depth));
  const uint32_t mask = (1u << log2t) - 1u;
  uint32_t slot = rb_hash(key, log2t);
  for (;;) {
    unsigned long long old = atomicCAS(&tkeys[slot], ~0ULL, (unsigned long long)key);
    if (old == ~0ULL || old == (unsigned long long)key) { atomicMin(&tvals[slot], (int32_t)i); return; }
    slot = (slot + 1) & mask;
  }
}
__global__ void k_rulebook_lookup(const int32_t* __restrict__ gc, const int32_t* __restrict__ batch, int64_t n, int depth,
                                  const unsigned long long* __restrict__ tkeys, const int32_t* __restrict__ tvals, int log2t,
                                  int ksize, int32_t* __restrict__ nbr) {
  const int taps = ksize * ksize * ksize;
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * taps) return;
  int t = (int)(gid / n); int64_t i = gid - (int64_t)t * n;
  int h = ksize >> 1;
  int iz = t % ksize, iy = (t / ksize) % ksize, ix = t / (ksize * ksize);
  int x = gc[3 * i] + ix - h, y = gc[3 * i + 1] + iy - h, z = gc[3 * i + 2] + iz - h;
  int lim = 1 << depth;
  int32_t res = -1;
  if (x >= 0 && y >= 0 && z >= 0 && x < lim && y < lim && z < lim) {
    uint64_t key = morton3(x, y, z) | (((uint64_t)(uint32_t)batch[i]) << (3 * depth));
    const uint32_t mask = (1u << log2t) - 1u;
    uint32_t slot = rb_hash(key, log2t);
    for (;;) {
      unsigned long long k = tkeys[slot];
      if (k == (unsigned long long)key) { res = tvals[slot]; break; }
      if (k == ~0ULL) break;
      slot = (slot + 1) & mask;
    }
  }
  nbr[gid] = res;
}
static int rb_log2t(int64_t n) { int l = 10; while ((1LL << l) < 2 * n) ++l; return l; }
extern "C" int64_t ss_subm_rulebook_table_size(int64_t n) { return n < 0 ? 0 : (1LL << rb_log2t(n)); }
extern "C" int ss_subm_rulebook_hashed(const int32_t* grid_coord, const int32_t* batch, int64_t n, int depth, int kernel_size,
                                       int32_t* nbr, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n < 0 || (kernel_size != 3 && kernel_size != 5) || depth < 0 || depth > 16 || n >= (1LL << 30)) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  const int l2 = rb_log2t(n);
  const size_t T = (size_t)1 << l2;
  if (!workspace || workspace_bytes < T * 12) return SS_ERR_WORKSPACE;
  unsigned long long* tkeys = (unsigned long long*)workspace;
  int32_t* tvals = (int32_t*)((char*)workspace + T * 8);
  if (hipMemsetAsync(tkeys, 0xFF, T * 8, stream) != hipSuccess || hipMemsetAsync(tvals, 0x7F, T * 4, stream) != hipSuccess) return SS_ERR_LAUNCH;
  SS_LAUNCH(k_rulebook_insert, dim3(ss_div_up(n, 256)), dim3(256), 0, stream, grid_coord, batch, n, depth, tkeys, tvals, l2);
  int64_t total = n * kernel_size * kernel_size * kernel_size;
  SS_LAUNCH(k_rulebook_lookup, dim3(ss_div_up(total, 256)), dim3(256), 0, stream, grid_coord, batch, n, depth, tkeys, tvals, l2,
            kernel_size, nbr);
  return SS_OK;
}

// sort key of the mask-grouped conv walk: key[p] = tapmask(order[p]) | (p >> coarse_bits) << taps, tapmask bit t = site has
// its tap-t neighbour (taps <= 27)
__global__ void k_tap_mask_keys(const int32_t* __restrict__ nbr, const int32_t* __restrict__ order, int64_t n, int taps,
                                int coarse_bits, int64_t* __restrict__ keys) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int64_t i = order ? order[p] : p;
  uint32_t m = 0;
  for (int t = 0; t < taps; ++t) m |= (nbr[(int64_t)t * n + i] >= 0 ? 1u : 0u) << t;
  keys[p] = (int64_t)m | ((p >> coarse_bits) << taps);
}
extern "C" int ss_subm_tap_mask_keys(const int32_t* nbr, const int32_t* order, int64_t n, int taps, int coarse_bits,
                                     int64_t* keys, hipStream_t stream) {
  if (n < 0 || taps < 1 || taps > 27 || coarse_bits < 0 || coarse_bits > 30) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  SS_LAUNCH(k_tap_mask_keys, dim3(ss_div_up(n, 256)), dim3(256), 0, stream, nbr, order, n, taps, coarse_bits, keys);
  return SS_OK;
}

// batch[i] = #offsets <= i  (offset2batch, pointcept/models/utils/misc.py:19-23); offsets (B) inclusive ends
__global__ void k_offsets_to_batch(const int32_t* __restrict__ offsets, int num_batches, int64_t n, int32_t* __restrict__ batch) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = num_batches;  // first b with offsets[b] > i
  while (lo < hi) { int mid = (lo + hi) >> 1; if (offsets[mid] <= i) lo = mid + 1; else hi = mid; }
  batch[i] = lo;
}
extern "C" int ss_offsets_to_batch(const int32_t* offsets, int num_batches, int64_t n, int32_t* batch, hipStream_t stream) {
  if (n < 0 || num_batches < 1) return SS_ERR_ARG;
  if (n == 0) return SS_OK;
  SS_LAUNCH(k_offsets_to_batch, dim3(ss_div_up(n, 256)), dim3(256), 0, stream, offsets, num_batches, n, batch);
  SS_CHECK_LAUNCH();
  return SS_OK;
}

// count[0] += #adjacent equal keys in a sorted key row (duplicate voxels)
__global__ void k_count_dups(const int64_t* __restrict__ sorted_keys, int64_t n, int32_t* __restrict__ count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int d = (i + 1 < n && sorted_keys[i] == sorted_keys[i + 1]) ? 1 : 0;
  uint64_t m = __ballot(d);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}
extern "C" int ss_count_duplicates(const int64_t* sorted_keys, int64_t n, int32_t* count, hipStream_t stream) {
  if (n < 0) return SS_ERR_ARG;
  hipMemsetAsync(count, 0, sizeof(int32_t), stream);
  if (n < 2) return SS_OK;
  SS_LAUNCH(k_count_dups, dim3(ss_div_up(n, 256)), dim3(256), 0, stream, sorted_keys, n, count);
  SS_CHECK_LAUNCH();
  return SS_OK;
}
