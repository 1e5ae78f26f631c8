// micro-benchmark: how many VALU instructions hide in the gap of v_mfma_f32_32x32x2_f32 (64 cycles) for ONE wave per SIMD
// (the Winograd kernels' situation: 256 accumulator registers leave room for one wave).  K plain f32 VALU ops between
// consecutive MFMAs on 16 different accumulators; FEED: the VALU chain produces the next MFMA's A operand.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K, bool FEED, int WAVES, int NACC = 16>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, long long* cyc) {
  f32x16 acc[NACC];
  float x[16], y[4];
  for (int n = 0; n < 16; ++n) x[n] = threadIdx.x * 0.001f + n;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int i = 0; i < 4; ++i) y[i] = 0.25f * (i + 1) + threadIdx.x;
  const float b = 0.5f, a0 = 1.0f + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      acc[xi % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(FEED ? x[xi] : a0, b, acc[xi % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) x[(xi + 1) & 15] = x[(xi + 1) & 15] - y[q & 3];
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int n = 0; n < 16; ++n) s += x[n];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int K, bool FEED, int WAVES, int NACC = 16> void run(float* d, long long* dc) {
  const int iters = 400;
  k<K, FEED, WAVES, NACC><<<256, 64 * WAVES>>>(d, 4, dc);
  k<K, FEED, WAVES, NACC><<<256, 64 * WAVES>>>(d, iters, dc);
  hipDeviceSynchronize();
  long long c = 0; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("K=%d feed=%d waves/CU=%d acc=%d: %.1f cycles per MFMA (one wave's view; two waves per SIMD share the pipe: 129 = both at full rate)\n", K, (int)FEED, WAVES, NACC, (double)c / (iters * 16.0));
}
int main() {
  float* d; long long* dc;
  hipMalloc(&d, 256 * 512 * sizeof(float)); hipMalloc(&dc, 8);
  run<0, false, 4>(d, dc); run<1, false, 4>(d, dc); run<2, false, 4>(d, dc); run<3, false, 4>(d, dc); run<4, false, 4>(d, dc); run<6, false, 4>(d, dc); run<8, false, 4>(d, dc);
  // two waves per SIMD (8 accumulators = 128 registers each): does the partner wave's MFMA stream hide this wave's VALU?
  run<0, false, 8, 8>(d, dc); run<1, false, 8, 8>(d, dc); run<2, false, 8, 8>(d, dc); run<4, false, 8, 8>(d, dc);
  run<0, false, 4, 8>(d, dc); run<2, false, 4, 8>(d, dc);
  run<1, true, 4>(d, dc); run<2, true, 4>(d, dc); run<3, true, 4>(d, dc); run<4, true, 4>(d, dc);
  return 0;
}
