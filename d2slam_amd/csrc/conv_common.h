// conv_common.h -- tile geometry and epilogue shared by the fp32 and fp16x2 conv kernels (internal).
#pragma once
#include "kernels.h"

#include <cstdlib>

namespace d2fe {

// development-library switches (d2fe_dev_env: constants in the product library).  D2FE_CONV64_TILE: 1 (default) = 4x32-pixel tile, 2x2 waves, ~53 KB LDS, 3 blocks/CU; 0 = 8x32 tile, 4x1 waves, 1 block/CU
static inline int tune_ablate() {
  static int v = -1;
  if (v < 0) v = d2fe_dev_env("D2FE_ABLATE", 0);
  return v;
}
static inline int tune_conv_pc() {
  static int v = -1;
  if (v < 0) v = d2fe_dev_env("D2FE_CONV_PC", 2);   // 0 off, 1 every layer, 2 every layer but the fused conv1a+conv1b (measured best)
  return v;
}
static inline int tune_conv64() {
  static int v = -1;
  if (v < 0) v = d2fe_dev_env("D2FE_CONV64_TILE", 1);
  return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
  // unconditional loads from clamped coordinates (all nine in flight at once; a branch per tap made hipcc wait for each
  // byte separately: ~9 serialized global round trips per tile), then the zero padding is applied by a select
  unsigned char raw[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = gy + ky - 1, xx = gx + kx - 1;
      const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
      raw[ky * 3 + kx] = img[(size_t)yc * stride + xc];
    }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = gy + ky - 1, xx = gx + kx - 1;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      v[ky * 3 + kx] = in ? (float)raw[ky * 3 + kx] * scale : 0.f;
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

// conv1a on the matrix pipe, written straight into an LDS patch (used by every fused conv1a+conv1b kernel).
// A [32 patch pixels x 10] x [10 x 64] problem per m-tile (K = 9 taps padded to 10): v_mfma_f32_32x32x2_f32 IS the oracle's
// fmaf chain in (ky,kx) order started from the bias, so the values are bit-identical to conv1a_kernel.  c1w/c1b are this
// lane's B fragments / bias (conv1a_mfma_load_weights), loaded once per workgroup.  MODE 0 writes fp32 (pixel stride CPF
// floats), MODE 1 writes the 2^SA-scaled fp16 hi/lo planes (pixel stride CPH halves).  wv/nw: this wave's index / wave count.
__device__ __forceinline__ void conv1a_mfma_load_weights(const float* __restrict__ w9x64, const float* __restrict__ bias, int lane,
                                                         float (&c1w)[5][2], float (&c1b)[2]) {
#pragma unroll
  for (int st = 0; st < 5; ++st) {
    const int k = 2 * st + (lane >> 5);
#pragma unroll
    for (int n = 0; n < 2; ++n) c1w[st][n] = k < 9 ? w9x64[k * 64 + n * 32 + (lane & 31)] : 0.f;
  }
  c1b[0] = bias[lane & 31];
  c1b[1] = bias[32 + (lane & 31)];
}

template <int MODE, int NPIX, int PW, int CPF, int CPH, int SA>
__device__ __forceinline__ void conv1a_mfma_stage(const uint8_t* __restrict__ ip, int img_stride_b, int aH, int aW, int ty0, int tx0,
                                                  const float (&c1w)[5][2], const float (&c1b)[2], int lane, int wv, int nw,
                                                  float* patch, _Float16* hi, _Float16* lo) {
  constexpr int NMT = (NPIX + 31) / 32;
  const float scale = (float)(1.0 / 255.0);
  const float sa = (float)(1 << SA);
  for (int mt = wv; mt < NMT; mt += nw) {
    const int p = mt * 32 + (lane & 31);
    const int gy = ty0 + p / PW - 1, gx = tx0 + p % PW - 1;       // image coordinates of the patch pixel (conv1b halo incl.)
    const bool pvalid = p < NPIX && gy >= 0 && gy < aH && gx >= 0 && gx < aW;
    const unsigned long long vmask = __ballot(pvalid);              // bit i (< 32): patch pixel mt*32+i lies inside the image
    unsigned char raw[5];
    bool tin[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int k = 2 * st + (lane >> 5);
      const int yy = gy + k / 3 - 1, xx = gx + k % 3 - 1;
      tin[st] = k < 9 && pvalid && yy >= 0 && yy < aH && xx >= 0 && xx < aW;
      const int yc = yy < 0 ? 0 : (yy >= aH ? aH - 1 : yy), xc = xx < 0 ? 0 : (xx >= aW ? aW - 1 : xx);
      raw[st] = ip[(size_t)yc * img_stride_b + xc];
    }
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = c1b[0]; c1[r] = c1b[1]; }
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const float av = tin[st] ? (float)raw[st] * scale : 0.f;
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, c1w[st][0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, c1w[st][1], c1, 0, 0, 0);
    }
    // D layout: column = lane&31 (channel), row i = (r&3) + 8*(r>>2) + 4*(lane>>5) (patch pixel mt*32 + i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int pp = mt * 32 + i;
      const bool ok = (vmask >> i) & 1ull;
      const float o0 = (ok && c0[r] > 0.f) ? c0[r] : 0.f;
      const float o1 = (ok && c1[r] > 0.f) ? c1[r] : 0.f;
      if (pp < NPIX) {
        if constexpr (MODE == 0) {
          patch[pp * CPF + (lane & 31)] = o0;
          patch[pp * CPF + 32 + (lane & 31)] = o1;
        } else {
          const float x0 = fminf(o0 * sa, 65000.f), x1 = fminf(o1 * sa, 65000.f);
          const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
          hi[pp * CPH + (lane & 31)] = h0;
          hi[pp * CPH + 32 + (lane & 31)] = h1;
          lo[pp * CPH + (lane & 31)] = (_Float16)(x0 - (float)h0);
          lo[pp * CPH + 32 + (lane & 31)] = (_Float16)(x1 - (float)h1);
        }
      }
    }
  }
}

// Epilogue: optional power-of-two rescale, ReLU, optional fused 2x2 max pool, NHWC store.
// C layout of v_mfma_*_32x32: col = lane&31 (output channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel of the m-tile).
// Pooling is lane-local: horizontal neighbours are registers (r, r+1), vertical neighbours are the wave's two m-tiles.
// Fast path (tile inside the image, all 32 channels of the n-tile real -- a wave-uniform test): the address of a store is
// a UNIFORM element offset (SALU) plus one per-lane 32-bit offset computed once, i.e. 2-3 VALU per store instead of
// the ~10 of the bounds-checked path (measured: the epilogue was 1.5-1.9 us of a 8.5 us fp16x2 tile).
template <int TW, int MT, int NT, bool POOL, bool RELU>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], float scale, int img, int ty0, int tx0,
                                              int wm, int ntile0, int lane) {
  float* out = a.out + (size_t)img * a.out_img_stride + a.out_coff;
  const int cs = a.out_cstride;
  const bool cfull = (ntile0 + NT) * 32 <= a.cout_real;
  if constexpr (!POOL) {
    // byte offsets inside one image fit 32 bits (<= 2^30 floats per activation tensor, checked at create)
    const unsigned lane_off = ((unsigned)(4 * (lane >> 5)) * (unsigned)cs + (unsigned)(lane & 31)) * 4u;
    const unsigned cs4 = (unsigned)cs * 4u, wcs4 = (unsigned)a.W * cs4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      int pyl, pxl, py0, px0;
      mtile_pixel<TW>(wm * MT + m, 31, pyl, pxl);
      mtile_pixel<TW>(wm * MT + m, 0, py0, px0);
      if (cfull && ty0 + pyl < a.H && tx0 + pxl < a.W) {
        if (a.ablate & 4) continue;
        const unsigned base = (unsigned)(ty0 + py0) * wcs4 + (unsigned)(tx0 + px0) * cs4 + (unsigned)ntile0 * 128u;   // uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int py, px;
          mtile_pixel<TW>(0, (r & 3) + 8 * (r >> 2), py, px);   // offsets inside the m-tile: compile-time
          const unsigned so = base + (unsigned)py * wcs4 + (unsigned)px * cs4;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            float v = acc[m][n][r] * scale;
            if (RELU) v = v > 0.f ? v : 0.f;
            *reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (so + n * 128u + lane_off)) = v;
          }
        }
        continue;
      }
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
            if (co < a.cout_real && !(a.ablate & 4)) out[((size_t)oy * a.W + ox) * cs + co] = v;
          }
        }
      }
    }
  } else {
    static_assert(MT == 2 && TW == 32, "pool epilogue expects 2 vertically adjacent 32-px m-tiles");
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    const int oy = (ty0 + wm * 2) >> 1;
    if (cfull && ty0 + wm * 2 + 1 < a.H && tx0 + 31 < a.W) {
      if (a.ablate & 4) return;
      const unsigned cs4 = (unsigned)cs * 4u;
      const unsigned lane_off = (unsigned)(2 * (lane >> 5)) * cs4 + (unsigned)(lane & 31) * 4u;
      const unsigned base = ((unsigned)oy * (unsigned)Wo + (unsigned)(tx0 >> 1)) * cs4 + (unsigned)ntile0 * 128u;   // uniform
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int iu = (r & 3) + 8 * (r >> 2);   // even
        const unsigned so = base + (unsigned)(iu >> 1) * cs4;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float m0 = fmaxf(acc[0][n][r], acc[0][n][r + 1]);
          const float m1 = fmaxf(acc[1][n][r], acc[1][n][r + 1]);
          float v = fmaxf(m0, m1) * scale;
          if (RELU) v = v > 0.f ? v : 0.f;
          *reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (so + n * 128u + lane_off)) = v;
        }
      }
      return;
    }
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
          if (co < a.cout_real && !(a.ablate & 4)) out[((size_t)oy * Wo + ox) * cs + co] = v;
        }
      }
    }
  }
}

}  // namespace d2fe
