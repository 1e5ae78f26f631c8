// micro-benchmark: v_mfma_f32_32x32x16_f16 issue rate vs number of independent accumulators (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(float* d, int waves_per_simd) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * waves_per_simd), block(256);
  k<NACC><<<grid, block>>>(d, 10);
  hipEventRecord(e0);
  k<NACC><<<grid, block>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)grid.x * 4 * iters * 8 * NACC;          // MFMAs
  const double tf = nm * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  const double cyc_per_mfma_per_simd = (ms * 1e-3 * 2.4e9) / ((double)iters * 8 * NACC * waves_per_simd);
  printf("NACC=%d waves/SIMD=%d: %.3f ms  %.0f TFLOP/s  ~%.1f cyc/MFMA/SIMD @2.4GHz\n", NACC, waves_per_simd, ms, tf, cyc_per_mfma_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  for (int w = 1; w <= 2; ++w) { run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); }
  return 0;
}
