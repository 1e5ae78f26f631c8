// conv_common.h -- tile geometry and epilogue shared by the fp32 and fp16x2 conv kernels (internal).
#pragma once
#include "kernels.h"

#include <cstdlib>

namespace d2fe {

// experiment knob: D2FE_CONV64_TILE: 1 (default) = 4x32-pixel tile, 2x2 waves, ~53 KB LDS, 3 blocks/CU; 0 = 8x32 tile, 4x1 waves, 1 block/CU
static inline int tune_ablate() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("D2FE_ABLATE"); v = e ? atoi(e) : 0; }
  return v;
}
static inline int tune_conv_pc() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("D2FE_CONV_PC"); v = e ? atoi(e) : 2; }   // 0 off, 1 every layer, 2 every layer but the fused conv1a+conv1b (measured best)
  return v;
}
static inline int tune_conv64() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("D2FE_CONV64_TILE"); v = e ? atoi(e) : 1; }
  return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// -----------------------------------------------------------------------------------------------------
// Tile geometry shared by both precisions.
//   m-tile = 32 pixels: TW >= 32 -> 32 consecutive x of one row; TW == 16 -> 2 rows x 16.
// -----------------------------------------------------------------------------------------------------
template <int TW>
__device__ __forceinline__ void mtile_pixel(int mt, int i, int& py, int& px) {
  if (TW >= 32) {
    constexpr int TPR = TW / 32;
    py = mt / TPR;
    px = (mt % TPR) * 32 + i;
  } else {
    constexpr int MTR = 32 / TW;
    py = mt * MTR + i / TW;
    px = i % TW;
  }
}


// conv1a (1 -> 64 channels, 3x3, pad 1) + ReLU for ONE pixel, evaluated exactly as conv1a_kernel / the oracle:
// u8 -> f32 * (1/255)  (SuperPoint::processInput, superpoint_tensorrt.cpp:185-198), chain acc0 = bias, taps in (ky,kx)
// order.  `oct` (group of 8 output channels) is wave-uniform, so the 72 weights + 8 biases come in through scalar loads.
__device__ __forceinline__ void conv1a_load_taps(const uint8_t* __restrict__ img, int stride, int H, int W, int gy, int gx,
                                                 float (&v)[9]) {
  const float scale = (float)(1.0 / 255.0);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = gy + ky - 1, xx = gx + kx - 1;
      v[ky * 3 + kx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)img[(size_t)yy * stride + xx] * scale : 0.f;
    }
}
__device__ __forceinline__ void conv1a_octet(const float (&v)[9], const float* __restrict__ w9x64,
                                             const float* __restrict__ bias, int oct, bool valid, float (&o)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = bias[oct * 8 + c];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = __builtin_fmaf(v[t], w9x64[t * 64 + oct * 8 + c], o[c]);
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = (valid && o[c] > 0.f) ? o[c] : 0.f;
}

// Epilogue: optional power-of-two rescale, ReLU, optional fused 2x2 max pool, NHWC store.
// C layout of v_mfma_*_32x32: col = lane&31 (output channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel of the m-tile).
// Pooling is lane-local: horizontal neighbours are registers (r, r+1), vertical neighbours are the wave's two m-tiles.
template <int TW, int MT, int NT, bool POOL, bool RELU>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], float scale, int img, int ty0, int tx0,
                                              int wm, int ntile0, int lane) {
  float* out = a.out + (size_t)img * a.out_img_stride + a.out_coff;
  if constexpr (!POOL) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int py, px;
        mtile_pixel<TW>(wm * MT + m, i, py, px);
        const int oy = ty0 + py, ox = tx0 + px;
        if (oy < a.H && ox < a.W) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int co = (ntile0 + n) * 32 + (lane & 31);
            float v = acc[m][n][r] * scale;
            if (RELU) v = v > 0.f ? v : 0.f;
            if (co < a.cout_real && !(a.ablate & 4)) out[((size_t)oy * a.W + ox) * a.out_cstride + co] = v;
          }
        }
      }
  } else {
    static_assert(MT == 2 && TW == 32, "pool epilogue expects 2 vertically adjacent 32-px m-tiles");
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    const int oy = (ty0 + wm * 2) >> 1;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // even
      const int ox = (tx0 + i) >> 1;
      if (oy < Ho && ox < Wo) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = (ntile0 + n) * 32 + (lane & 31);
          const float m0 = fmaxf(acc[0][n][r], acc[0][n][r + 1]);
          const float m1 = fmaxf(acc[1][n][r], acc[1][n][r + 1]);
          float v = fmaxf(m0, m1) * scale;
          if (RELU) v = v > 0.f ? v : 0.f;
          if (co < a.cout_real && !(a.ablate & 4)) out[((size_t)oy * Wo + ox) * a.out_cstride + co] = v;
        }
      }
    }
  }
}

}  // namespace d2fe
