// conv.hip -- SuperPoint convolution stack for gfx950 (MI355X), implicit GEMM on MFMA.
//
// Replaces the TensorRT engine execution at d2frontend/src/CNN/superpoint_tensorrt.cpp:150
// (network definition: d2frontend/superpoint.ipynb:300-374).
//
// GEMM view: M = output pixels, N = output channels, K = (ky, kx, ci).  Activations are NHWC fp32 in HBM.
// A block stages the (TH+2)x(TW+2)xCin input patch of its TH x TW pixel tile in LDS once (halo included,
// zero-filled outside the image == the conv's zero padding) and then walks K with the patch as the A operand;
// B fragments (weights) are pre-packed on the host in MFMA lane order and stream from L2 as one coalesced
// 1 KiB load per 32-wide N tile per 8 input channels.
//
// Exact mode (precision 0): v_mfma_f32_32x32x2_f32.  Per output the accumulation is bit-for-bit the fp32
// fmaf chain acc0 = bias; for ky, kx, ci ascending: acc = fmaf(x, w, acc)  -- the order the oracle uses
// (oracle/d2fe_oracle.c orc_conv), so activations compare bitwise.
// Fast mode (precision 1): operands split into fp16 hi + lo (x = hi + lo to ~2^-22), three
// v_mfma_f32_32x32x16_f16 per k-step (hi*hi + hi*lo + lo*hi) into one fp32 accumulator.
#include "conv_common.h"

namespace d2fe {

// =====================================================================================================
// Exact fp32 kernel
// =====================================================================================================
// waves/SIMD the LDS footprint allows (160 KiB per CU, 4 waves per block on 4 SIMDs): used as the register-allocation target
constexpr int f32_min_waves(int CIN, int KS, int TH, int TW) {
  const int lds = (TH + KS - 1) * (TW + KS - 1) * (CIN + 1) * 4;
  return 163840 / lds >= 3 ? 3 : (163840 / lds >= 2 ? 2 : 1);
}

template <int CIN, int KS, int TH, int TW, int WM, int WN, int MT, int NT, bool POOL, bool RELU, bool FUSE1A = false>
__global__ __launch_bounds__(WM * WN * 64, f32_min_waves(CIN, KS, TH, TW)) void conv_f32_kernel(ConvArgs a) {
  constexpr int P = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int CP = CIN + 1;  // odd pixel stride: 32 pixels x same channel hit 32 distinct banks
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int TAPS = KS * KS;
  constexpr int C8 = CIN / 8;
  static_assert(TH * TW == WM * MT * 32, "tile / wave mismatch");
  extern __shared__ __attribute__((aligned(16))) float patch[];

  const int tiles_x = (a.W + TW - 1) / TW;
  const int tx0 = (blockIdx.x % tiles_x) * TW;
  const int ty0 = (blockIdx.x / tiles_x) * TH;
  const int img = blockIdx.z;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // ---- stage the input patch -------------------------------------------------------------------------
  if constexpr (FUSE1A) {
    // conv1a is evaluated on the fly for the (TH+2)x(TW+2) patch: the 78.6 MB/image conv1a activation never exists.  Same staging as
    // the Winograd and fp16x2 kernels: frame bytes through LDS, a unit = 32 patch pixels x 32 channels as a chain of five
    // v_mfma_f32_32x32x2_f32 with the weights as the A operand (k = 0: bias against 1.0, then the taps in (ky,kx) order = the fmaf chain
    // of conv1a_octet, bit for bit); a patch pixel outside the image gets all taps and the bias slot zeroed, i.e. +0.
    static_assert(CIN == 64 && KS == 3 && NTHREADS == 256, "fused prologue is conv1a -> conv1b, four waves");
    constexpr int NPIX = PH * PW, FR = PH + 2, FC = PW + 2, NMT = (NPIX + 31) / 32;
    unsigned char* u8p = reinterpret_cast<unsigned char*>(patch + NPIX * CP);
    const uint8_t* ip = a.img + (size_t)img * a.img_istride;
    for (int i = tid; i < FR * FC; i += NTHREADS) {
      const int r = i / FC, c = i - r * FC;
      const int gy = ty0 - P - 1 + r, gx = tx0 - P - 1 + c;
      u8p[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? ip[(size_t)gy * a.img_stride + gx] : (unsigned char)0;
    }
    const int hh = lane >> 5, nt = wave & 1;
    float c1a[5];
    int koff[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int k = 2 * st + hh;
      c1a[st] = k == 0 ? a.b1a[nt * 32 + (lane & 31)] : a.w1a[(k - 1) * 64 + nt * 32 + (lane & 31)];
      koff[st] = k == 0 ? 0 : ((k - 1) / 3) * FC + (k - 1) % 3;
    }
    const float scale = (float)(1.0 / 255.0);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int u = wave; u < 2 * NMT; u += 4) {             // u & 1 == wave & 1: a wave only ever needs one half of the weights
      const int pidx = (u >> 1) * 32 + (lane & 31);
      const int py = pidx / PW, px = pidx - py * PW;
      const int gy = ty0 + py - P, gx = tx0 + px - P;
      const bool inpatch = pidx < NPIX;
      const bool pvalid = inpatch && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int tb = inpatch ? py * FC + px : 0;
      float tap[5];
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        float v = (float)u8p[tb + koff[st]] * scale;
        if (st == 0) v = hh ? v : 1.0f;
        tap[st] = pvalid ? v : 0.f;
      }
      f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(c1a[0], tap[0], zero, 0, 0, 0);
#pragma unroll
      for (int st = 1; st < 5; ++st) d = __builtin_amdgcn_mfma_f32_32x32x2f32(c1a[st], tap[st], d, 0, 0, 0);
      if (inpatch) {
        float* dst = patch + pidx * CP + nt * 32 + 4 * hh;      // rows (channels) 8 q + 4 hh + (0..3) of half nt; odd pixel stride: scalar stores
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[8 * (r >> 2) + (r & 3)] = d[r] > 0.f ? d[r] : 0.f;
      }
    }
  } else {
    const float* in = a.in + (size_t)img * a.in_img_stride + a.in_coff;
    constexpr int C4 = CIN / 4;
    constexpr int TOTAL = PH * PW * C4;
    constexpr int ITERS = (TOTAL + NTHREADS - 1) / NTHREADS;
    constexpr int UNR = 8;
    for (int it0 = 0; it0 < ITERS; it0 += UNR) {
      f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = (it0 + u) * NTHREADS + tid;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (it0 + u < ITERS && idx < TOTAL) {
          const int pix = idx / C4, c4 = idx % C4;
          const int gy = ty0 + pix / PW - P, gx = tx0 + pix % PW - P;
          if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && !D2FE_ABL(a, 1))
            v[u] = *reinterpret_cast<const f32x4*>(in + ((size_t)gy * a.W + gx) * a.in_cstride + c4 * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = (it0 + u) * NTHREADS + tid;
        if (it0 + u < ITERS && idx < TOTAL) {
          const int pix = idx / C4, c4 = idx % C4;
          float* d = patch + pix * CP + c4 * 4;
          d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
        }
      }
    }
  }
  __syncthreads();

  // ---- accumulators start from the bias ----------------------------------------------------------------
  const int ntile0 = blockIdx.y * (WN * NT) + wn * NT;  // global 32-wide N tile index
  f32x16 acc[MT][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float b = a.bias[(ntile0 + n) * 32 + (lane & 31)];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = b;
  }

  // per-lane A base offsets (floats) of this wave's m-tiles at tap (0,0), channel k-half
  int aoff[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int py, px;
    mtile_pixel<TW>(wm * MT + m, lane & 31, py, px);
    aoff[m] = (py * PW + px) * CP + (lane >> 5);
  }

  // packed weights: [ntile][tap][c8][lane] float4;  float4[q] = W[co = ntile*32 + (lane&31)][ci = c8*8 + 2q + (lane>>5)][tap]
  const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack);
  const f32x4* wbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) wbase[n] = wp + (size_t)(ntile0 + n) * TAPS * C8 * 64 + lane;

  // B fragments are double-buffered in registers in groups of G k-steps: group g+1 is requested from L2 before the
  // MFMAs of group g start (sched_barrier keeps hipcc from sinking the loads next to their uses), so one L2 round
  // trip is covered by G*4*MT*NT MFMAs (>= 64 x 64 cycles).
  constexpr int G = 8;                 // k-steps (of 8 channels) per group; divides C8
  constexpr int GPT = C8 / G;          // groups per tap
  constexpr int NG = TAPS * GPT;
  static_assert(C8 % G == 0, "group size must divide the steps per tap");
  f32x4 bq[2][G][NT];
  auto load_grp = [&](int buf, int grp) {
    if (D2FE_ABL(a, 2)) grp = 0;
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) bq[buf][j][n] = wbase[n][(size_t)(grp * G + j) * 64];
  };
  auto compute_grp = [&](int buf, int grp) {
    const int tap = grp / GPT, c80 = (grp % GPT) * G;
    const int tap_off = ((tap / KS) * PW + (tap % KS)) * CP + c80 * 8;
#pragma unroll
    for (int j = 0; j < G; ++j) {
      float av[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float* ap = patch + aoff[m] + tap_off + j * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) av[m][q] = ap[2 * q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][q], bq[buf][j][n][q], acc[m][n], 0, 0, 0);
    }
  };
  load_grp(0, 0);
#pragma unroll 1
  for (int g = 0; g + 1 < NG; g += 2) {
    load_grp(1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_grp(0, g);
    load_grp(0, g + 2 < NG ? g + 2 : NG - 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_grp(1, g + 1);
  }
  if constexpr (NG & 1) compute_grp(0, NG - 1);

  conv_epilogue<TW, MT, NT, POOL, RELU>(a, acc, 1.0f, img, ty0, tx0, wm, ntile0, lane);
}

template <int CIN, int KS, int TH, int TW, int WM, int WN, int MT, int NT>
static hipError_t launch_f32(bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s) {
  constexpr int BN = WN * NT * 32;
  constexpr size_t lds = (size_t)(TH + KS - 1) * (TW + KS - 1) * (CIN + 1) * sizeof(float);
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  dim3 grid(tiles_x * tiles_y, cout_pad / BN, a.n_img), block(WM * WN * 64);
  if (cout_pad % BN) return hipErrorInvalidValue;
#define D2FE_LAUNCH(PL, RL)                                                                          \
  do {                                                                                               \
    auto k = conv_f32_kernel<CIN, KS, TH, TW, WM, WN, MT, NT, PL, RL>;                               \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
    if (e != hipSuccess) return e;                                                                   \
    hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                   \
  } while (0)
  if constexpr (MT == 2 && TW == 32) {
    if (pool) { if (relu) D2FE_LAUNCH(true, true); else D2FE_LAUNCH(true, false); return hipGetLastError(); }
  } else {
    if (pool) return hipErrorInvalidValue;
  }
  if (relu) D2FE_LAUNCH(false, true); else D2FE_LAUNCH(false, false);
#undef D2FE_LAUNCH
  return hipGetLastError();
}

static hipError_t launch_f32_fused1b(int cout_pad, const ConvArgs& a, hipStream_t s) {
  constexpr int TH = 4, TW = 32;
  constexpr size_t lds = (size_t)(TH + 2) * (TW + 2) * 65 * sizeof(float) + (TH + 4) * (TW + 4);      // the patch + the frame bytes
  if (cout_pad != 64) return hipErrorInvalidValue;
  auto k = conv_f32_kernel<64, 3, TH, TW, 2, 2, 2, 1, true, true, true>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  dim3 grid(((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH), 1, a.n_img), block(256);
  hipLaunchKernelGGL(k, grid, block, lds, s, a);
  return hipGetLastError();
}

hipError_t launch_conv_f16x2(ConvShape shape, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s);

hipError_t launch_conv(ConvShape shape, int precision, bool pool, bool relu, int cout_pad, const ConvArgs& a,
                       hipStream_t s) {
  if (tune_conv_pc() == 1 || (tune_conv_pc() == 2 && shape != CONV1B_FUSED)) return launch_conv_pc(shape, precision, pool, relu, cout_pad, a, s);
  if (precision == 1) return launch_conv_f16x2(shape, pool, relu, cout_pad, a, s);
  switch (shape) {
    case CONV_64_T8x32:
      if (tune_conv64() == 1) return launch_f32<64, 3, 4, 32, 2, 2, 2, 1>(pool, relu, cout_pad, a, s);
      return launch_f32<64, 3, 8, 32, 4, 1, 2, 2>(pool, relu, cout_pad, a, s);
    case CONV_128_T4x32:     return launch_f32<128, 3, 4, 32, 2, 2, 2, 2>(pool, relu, cout_pad, a, s);
    case CONV_128_T4x16:     return launch_f32<128, 3, 4, 16, 1, 4, 2, 1>(pool, relu, cout_pad, a, s);
    case CONV_256_1x1_T4x16: return launch_f32<256, 1, 4, 16, 1, 4, 2, 1>(pool, relu, cout_pad, a, s);
    case CONV1B_FUSED:       return launch_f32_fused1b(cout_pad, a, s);
  }
  return hipErrorInvalidValue;
}

// -----------------------------------------------------------------------------------------------------
// host-side weight packing (fp32 fragment order)
// -----------------------------------------------------------------------------------------------------
size_t packed_weight_floats_f32(int cout_pad, int cin, int ks) { return (size_t)cout_pad * cin * ks * ks; }

void pack_weights_f32(const float* w, int cout, int cin, int ks, int cout_pad, float* dst) {
  const int taps = ks * ks, c8n = cin / 8;
  for (int nt = 0; nt < cout_pad / 32; ++nt)
    for (int tap = 0; tap < taps; ++tap)
      for (int c8 = 0; c8 < c8n; ++c8)
        for (int lane = 0; lane < 64; ++lane)
          for (int q = 0; q < 4; ++q) {
            const int co = nt * 32 + (lane & 31);
            const int ci = c8 * 8 + 2 * q + (lane >> 5);
            const float v = co < cout ? w[((size_t)co * cin + ci) * taps + tap] : 0.f;
            dst[((((size_t)nt * taps + tap) * c8n + c8) * 64 + lane) * 4 + q] = v;
          }
}

// =====================================================================================================
// conv1a: 1 -> 64 channels, K = 9.  VALU kernel (K too small for MFMA to matter), fused u8 -> f32 prep
// (SuperPoint::processInput, superpoint_tensorrt.cpp:185-198: convertTo(CV_32FC1, 1/255)).
// Thread = (pixel, group of 16 output channels); chain order (ky,kx), acc0 = bias: bitwise == oracle.
// =====================================================================================================
__global__ __launch_bounds__(256) void conv1a_kernel(const uint8_t* __restrict__ img, int stride, long img_stride,
                                                     int H, int W, const float* __restrict__ w9x64,
                                                     const float* __restrict__ bias, float* __restrict__ out) {
  __shared__ float wsm[9 * 64 + 64];
  for (int i = threadIdx.x; i < 9 * 64 + 64; i += 256) wsm[i] = i < 576 ? w9x64[i] : bias[i - 576];
  __syncthreads();
  const int n = blockIdx.z;
  const int y = blockIdx.y;
  const int x = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int g = threadIdx.x & 3;
  if (x >= W) return;
  const uint8_t* ip = img + (size_t)n * img_stride;
  const float scale = (float)(1.0 / 255.0);
  float v[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      v[ky * 3 + kx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)ip[(size_t)yy * stride + xx] * scale : 0.f;
    }
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = wsm[576 + g * 16 + c];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = __builtin_fmaf(v[t], wsm[t * 64 + g * 16 + c], acc[c]);
  float* op = out + (((size_t)n * H + y) * W + x) * 64 + g * 16;
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float t = acc[c4 * 4 + j]; o[j] = t > 0.f ? t : 0.f; }
    *reinterpret_cast<f32x4*>(op + c4 * 4) = o;
  }
}

// conv1a on the matrix pipe (the stand-alone form of conv1a_mfma_stage): a wave walks 8 consecutive m-tiles of 32 pixels of one
// image row; per m-tile a [32 px x 10] x [10 x 64] problem = 10 v_mfma_f32_32x32x2_f32 started from the bias (the same fmaf chain
// in (ky,kx) order as conv1a_kernel, K = 9 padded with a zero tap), ReLU, 32 stores of 2 x 128 contiguous bytes.  The B fragments
// and the bias stay in 12 registers.  Used by the Winograd mode, whose conv1b reads the materialised activation.
constexpr int C1A_TILES_PER_WAVE = 8;
__global__ __launch_bounds__(256) void conv1a_mfma_kernel(const uint8_t* __restrict__ img, int stride, long img_stride, int H, int W,
                                                          int tiles_x, long total_tiles, const float* __restrict__ w9x64,
                                                          const float* __restrict__ bias, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  float c1w[5][2], c1b[2];
  conv1a_mfma_load_weights(w9x64, bias, lane, c1w, c1b);
  const float scale = (float)(1.0 / 255.0);
#pragma unroll 1
  for (int u = 0; u < C1A_TILES_PER_WAVE; ++u) {
    const long t = wave * C1A_TILES_PER_WAVE + u;
    if (t >= total_tiles) return;
    const int xt = (int)(t % tiles_x);
    const long row = t / tiles_x;                 // n * H + y
    const int y = (int)(row % H);
    const uint8_t* ip = img + (size_t)(row / H) * img_stride;
    const int gx = xt * 32 + (lane & 31);
    unsigned char raw[5];
    bool tin[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int k = 2 * st + (lane >> 5);
      const int yy = y + k / 3 - 1, xx = gx + k % 3 - 1;
      tin[st] = k < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
      raw[st] = ip[(size_t)yc * stride + xc];
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
    float* op = out + ((size_t)row * W + xt * 32 + 4 * (lane >> 5)) * 64 + (lane & 31);
    const bool fullw = xt * 32 + 32 <= W;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2);       // + 4 * (lane >> 5): pixel of the m-tile
      if (fullw || xt * 32 + i + 4 * (lane >> 5) < W) {
        op[(size_t)i * 64] = c0[r] > 0.f ? c0[r] : 0.f;
        op[(size_t)i * 64 + 32] = c1[r] > 0.f ? c1[r] : 0.f;
      }
    }
  }
}

hipError_t launch_conv1a(const uint8_t* img, int stride, long img_stride_bytes, int H, int W, int n,
                         const float* w9x64, const float* bias, float* out, hipStream_t s) {
  static int valu = -1;
  if (valu < 0) valu = d2fe_dev_env("D2FE_CONV1A_VALU", 0);   // 1: the VALU kernel (A/B timing)
  if (valu) {
    dim3 grid((W + 63) / 64, H, n), block(256);
    hipLaunchKernelGGL(conv1a_kernel, grid, block, 0, s, img, stride, img_stride_bytes, H, W, w9x64, bias, out);
    return hipGetLastError();
  }
  const int tiles_x = (W + 31) / 32;
  const long total = (long)tiles_x * H * n;
  const long waves = (total + C1A_TILES_PER_WAVE - 1) / C1A_TILES_PER_WAVE;
  hipLaunchKernelGGL(conv1a_mfma_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, img, stride, img_stride_bytes, H, W,
                     tiles_x, total, w9x64, bias, out);
  return hipGetLastError();
}

}  // namespace d2fe
