// micro-benchmark: the fp16x2 consumer inner loop piece by piece (one wave per SIMD, 4 waves per block, 1 block per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CPH = 72, PW = 34, NPIX = 204, STEPS = 36;
template <int VAR>   // bit0: LDS A reads, bit1: global B loads (ring 10), bit2: sched_barrier
__global__ __launch_bounds__(256) void k(const f16x8* __restrict__ w, float* out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* hi = lds; _Float16* lo = lds + NPIX * CPH;
  for (int i = threadIdx.x; i < 2 * NPIX * CPH; i += 256) lds[i] = (_Float16)(0.001f * (i & 255));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  int aoff[2];
  for (int m = 0; m < 2; ++m) aoff[m] = ((wm * 2 + m) * PW + (lane & 31)) * CPH + 8 * (lane >> 5);
  const f16x8* wb = w + (size_t)wn * STEPS * 128 + lane;
  f32x16 acc[2];
  for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  f16x8 ca, cb; for (int i = 0; i < 8; ++i) { ca[i] = (_Float16)(0.01f * i); cb[i] = (_Float16)(0.02f * i); }
  for (int t = 0; t < tiles; ++t) {
    constexpr int R = 10;
    f16x8 ring[R][2];
    if (VAR & 2) {
#pragma unroll
      for (int s = 0; s < R - 1; ++s) { ring[s][0] = wb[s * 128]; ring[s][1] = wb[s * 128 + 64]; }
    }
    f16x8 ah[2][2], al[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { ah[0][m] = ca; al[0][m] = cb; ah[1][m] = ca; al[1][m] = cb; }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if ((VAR & 2) && s + R - 1 < STEPS) { ring[(s + R - 1) % R][0] = wb[(s + R - 1) * 128]; ring[(s + R - 1) % R][1] = wb[(s + R - 1) * 128 + 64]; }
      if ((VAR & 1) && s + 1 < STEPS) {
        const int tap = (s + 1) / 4, ks = (s + 1) % 4;
        const int o = ((tap / 3) * PW + (tap % 3)) * CPH + ks * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          ah[(s + 1) & 1][m] = *reinterpret_cast<const f16x8*>(hi + aoff[m] + o);
          al[(s + 1) & 1][m] = *reinterpret_cast<const f16x8*>(lo + aoff[m] + o);
        }
      }
      if (VAR & 4) __builtin_amdgcn_sched_barrier(0);
      const f16x8 bh = (VAR & 2) ? ring[s % R][0] : ca, bl = (VAR & 2) ? ring[s % R][1] : cb;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s & 1][m], bh, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s & 1][m], bl, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s & 1][m], bh, acc[m], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int VAR> void run(const f16x8* w, float* d, int blocks_per_cu) {
  const int tiles = 40;
  const size_t lds = 2 * NPIX * CPH * sizeof(_Float16);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  dim3 grid(256 * blocks_per_cu), block(256);
  k<VAR><<<grid, block, lds>>>(w, d, 2);
  (void)hipEventRecord(e0);
  k<VAR><<<grid, block, lds>>>(w, d, tiles);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)grid.x * 4 * tiles * STEPS * 6;
  printf("VAR=%d blocks/CU=%d: %.3f ms  %.0f TFLOP/s (MFMA-executed)  %.1f us per tile\n", VAR, blocks_per_cu, ms,
         nm * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12, ms * 1e3 / tiles / blocks_per_cu * blocks_per_cu);
}
int main() {
  f16x8* w; (void)hipMalloc(&w, 2 * STEPS * 128 * sizeof(f16x8)); (void)hipMemset(w, 0, 2 * STEPS * 128 * sizeof(f16x8));
  float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  for (int b = 1; b <= 2; ++b) { run<0>(w, d, b); run<1>(w, d, b); run<5>(w, d, b); run<2>(w, d, b); run<6>(w, d, b); run<3>(w, d, b); run<7>(w, d, b); }
  return 0;
}
