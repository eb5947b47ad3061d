// How fast can a [n][768] bf16 matrix be streamed when every workgroup walks its 256-row block in K steps of P bytes per row (the access
// pattern of the open-vocabulary scan: P = 128), compared with wider pieces and whole rows?  (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/row_piece_stream.hip -o scripts/ubench/bin/row_piece_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int P>   // bytes of a row read per step (multiple of 16)
__global__ void __launch_bounds__(512) k_stream(const uint4* __restrict__ x, int64_t n, int row_bytes, float* sink) {
  constexpr int CH = P / 16;                                   // 16-byte chunks per row piece
  const int64_t m0 = (int64_t)blockIdx.x * 256;
  const int steps = row_bytes / P;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    // one barrier per step as in the scan (keeps the waves of the workgroup on the same step)
    for (int c = threadIdx.x; c < 256 * CH; c += 512) {
      const int64_t r = m0 + c / CH;
      if (r < n) { uint4 v = x[(r * row_bytes + (int64_t)s * P) / 16 + c % CH]; acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w); }
    }
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}
int main() {
  const int64_t n = 1000000; const int row_bytes = 1536;
  uint4* x; float* sink;
  hipMalloc(&x, n * row_bytes); hipMalloc(&sink, 4);
  hipMemset(x, 1, n * row_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (int)((n + 255) / 256);
  auto run = [&](int P) {
    for (int rep = 0; rep < 12; ++rep) {
      if (rep == 2) hipEventRecord(e0, 0);
      switch (P) {
        case 128: k_stream<128><<<grid, 512>>>(x, n, row_bytes, sink); break;
        case 256: k_stream<256><<<grid, 512>>>(x, n, row_bytes, sink); break;
        case 512: k_stream<512><<<grid, 512>>>(x, n, row_bytes, sink); break;
        default: k_stream<1536><<<grid, 512>>>(x, n, row_bytes, sink); break;
      }
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("piece %4d B per row and step: %.3f ms  %.0f GB/s\n", P, ms, (double)n * row_bytes / ms / 1e6);
  };
  run(128); run(256); run(512); run(1536);
  return 0;
}
