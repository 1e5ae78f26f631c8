// conv_f16.hip -- "fast" precision of the SuperPoint conv stack: fp16 hi/lo split operands on the
// double-rate gfx950 matrix instruction v_mfma_f32_32x32x16_f16, fp32 accumulation.
//
// Every fp32 operand x is represented as hi + lo with hi = fp16(x), lo = fp16(x - hi)  (|x - hi - lo| <~ 2^-22 |x|);
// a product is evaluated as hi_a*hi_b + hi_a*lo_b + lo_a*hi_b (the dropped lo*lo term is <= 2^-22 relative), i.e.
// three MFMAs per k-step at 16x the fp32-MFMA rate => 5.3x the exact mode's throughput ceiling at ~fp32 accuracy.
// To keep the lo parts out of fp16's subnormal range, activations are scaled by 2^SA and weights by 2^SW before
// the split (exact power-of-two scalings); the epilogue multiplies by 2^-(SA+SW).
//
// Same tiling and staging as conv.hip: the (TH+2)x(TW+2)xCin patch is converted once per block while it is
// staged (global fp32 -> LDS fp16 hi plane + lo plane), and is then read as ds_read_b128 A fragments
// (pixel stride Cin+8 halves -> the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots).
#include "conv_common.h"

namespace d2fe {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int F16_SA = 4;   // activation scale 2^4  (|x| <= 4094 representable; clamped)
constexpr int F16_SW = 8;   // weight scale 2^8

template <int CIN, int KS, int TH, int TW, int WM, int WN, int MT, int NT, bool POOL, bool RELU, bool FUSE1A = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_f16x2_kernel(ConvArgs a) {
  constexpr int P = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int CPH = CIN + 8;  // halves per pixel (16-byte slots per pixel = CIN/8 + 1, odd)
  constexpr int NPIX = PH * PW;
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int TAPS = KS * KS;
  constexpr int KST = CIN / 16;
  static_assert(TH * TW == WM * MT * 32, "tile / wave mismatch");
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* hi = lds;
  _Float16* lo = lds + NPIX * CPH;

  const int tiles_x = (a.W + TW - 1) / TW;
  const int tx0 = (blockIdx.x % tiles_x) * TW;
  const int ty0 = (blockIdx.x / tiles_x) * TH;
  const int img = blockIdx.z;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // ---- stage + split the input patch ---------------------------------------------------------------------
  if constexpr (FUSE1A) {
    // conv1a on the fp32 matrix pipe (the Winograd kernel's fused staging, conv_wino.hip): the (PH+2) x (PW+2) frame bytes under the
    // patch go through LDS once (zero outside the image = conv1a's padding), a unit = 32 patch pixels x 32 channels is a chain of five
    // v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand (k = 0: the bias against a tap of 1.0, then the taps in (ky,kx) order --
    // bit for bit the fmaf chain of conv1a_octet), so a lane (= pixel) ends up with 4 x 4 consecutive channels: scaled, split into
    // hi/lo and stored as 8-byte runs.  A patch pixel outside the image (conv1b's padding) gets all taps and the bias slot zeroed.
    static_assert(CIN == 64 && KS == 3 && NTHREADS == 256, "fused prologue is conv1a -> conv1b, four waves");
    constexpr int FR = PH + 2, FC = PW + 2, NMT = (NPIX + 31) / 32;
    unsigned char* u8p = reinterpret_cast<unsigned char*>(lds + 2 * NPIX * CPH);
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
    const float sa = (float)(1 << F16_SA), scale = (float)(1.0 / 255.0);
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
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // rows (channels) 8 q + 4 hh + (0..3) of half nt
          f16x4 h4, l4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float o = d[4 * q + e] > 0.f ? d[4 * q + e] : 0.f;
            const float x = fminf(o * sa, 65000.f);
            const _Float16 h = (_Float16)x;
            h4[e] = h;
            l4[e] = (_Float16)(x - (float)h);
          }
          const int off = pidx * CPH + nt * 32 + 8 * q + 4 * hh;
          *reinterpret_cast<f16x4*>(hi + off) = h4;
          *reinterpret_cast<f16x4*>(lo + off) = l4;
        }
      }
    }
  } else {
    const float* in = a.in + (size_t)img * a.in_img_stride + a.in_coff;
    constexpr int C4 = CIN / 4;
    constexpr int TOTAL = NPIX * C4;
    constexpr int ITERS = (TOTAL + NTHREADS - 1) / NTHREADS;
    constexpr int UNR = 8;
    const float sa = (float)(1 << F16_SA);
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
          f16x4 h4, l4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = v[u][j] * sa;
            x = fminf(fmaxf(x, -65000.f), 65000.f);
            const _Float16 h = (_Float16)x;
            h4[j] = h;
            l4[j] = (_Float16)(x - (float)h);
          }
          *reinterpret_cast<f16x4*>(hi + pix * CPH + c4 * 4) = h4;
          *reinterpret_cast<f16x4*>(lo + pix * CPH + c4 * 4) = l4;
        }
      }
    }
  }
  __syncthreads();

  const int ntile0 = blockIdx.y * (WN * NT) + wn * NT;
  const float bscale = (float)(1 << (F16_SA + F16_SW));
  f32x16 acc[MT][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float b = a.bias[(ntile0 + n) * 32 + (lane & 31)] * bscale;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = b;
  }

  int aoff[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int py, px;
    mtile_pixel<TW>(wm * MT + m, lane & 31, py, px);
    aoff[m] = (py * PW + px) * CPH + 8 * (lane >> 5);
  }

  // packed weights: [ntile][tap][kstep][hi|lo][lane] f16x8;  element i = W[co = ntile*32 + (lane&31)][ci = kstep*16 + 8*(lane>>5) + i][tap]
  const f16x8* wp = reinterpret_cast<const f16x8*>(a.wpack);
  const f16x8* wbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) wbase[n] = wp + (size_t)(ntile0 + n) * TAPS * KST * 128 + lane;

  // B fragments are double-buffered in registers in groups of G k-steps (hi and lo): group g+1 is requested from L2
  // before the MFMAs of group g start (sched_barrier keeps hipcc from sinking the loads next to their uses).
  constexpr int G = 4;                 // k-steps (of 16 channels) per group; divides KST
  constexpr int GPT = KST / G;
  constexpr int NG = TAPS * GPT;
  static_assert(KST % G == 0, "group size must divide the steps per tap");
  f16x8 bq[2][G][NT][2];
  auto load_grp = [&](int buf, int grp) {
    if (D2FE_ABL(a, 2)) grp = 0;
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bq[buf][j][n][0] = wbase[n][(size_t)(grp * G + j) * 128];
        bq[buf][j][n][1] = wbase[n][(size_t)(grp * G + j) * 128 + 64];
      }
  };
  auto compute_grp = [&](int buf, int grp) {
    const int tap = grp / GPT, ks0 = (grp % GPT) * G;
    const int tap_off = ((tap / KS) * PW + (tap % KS)) * CPH + ks0 * 16;
#pragma unroll
    for (int j = 0; j < G; ++j) {
      f16x8 ah[MT], al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ah[m] = *reinterpret_cast<const f16x8*>(hi + aoff[m] + tap_off + j * 16);
        al[m] = *reinterpret_cast<const f16x8*>(lo + aoff[m] + tap_off + j * 16);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bq[buf][j][n][0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bq[buf][j][n][1], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bq[buf][j][n][0], acc[m][n], 0, 0, 0);
        }
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

  conv_epilogue<TW, MT, NT, POOL, RELU>(a, acc, 1.0f / bscale, img, ty0, tx0, wm, ntile0, lane);
}

template <int CIN, int KS, int TH, int TW, int WM, int WN, int MT, int NT>
static hipError_t launch_f16(bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s) {
  constexpr int BN = WN * NT * 32;
  constexpr size_t lds = (size_t)(TH + KS - 1) * (TW + KS - 1) * (CIN + 8) * sizeof(_Float16) * 2;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  dim3 grid(tiles_x * tiles_y, cout_pad / BN, a.n_img), block(WM * WN * 64);
  if (cout_pad % BN) return hipErrorInvalidValue;
#define D2FE_LAUNCH(PL, RL)                                                                          \
  do {                                                                                               \
    auto k = conv_f16x2_kernel<CIN, KS, TH, TW, WM, WN, MT, NT, PL, RL>;                             \
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

static hipError_t launch_f16_fused1b(int cout_pad, const ConvArgs& a, hipStream_t s) {
  constexpr int TH = 4, TW = 32;
  constexpr size_t lds = (size_t)(TH + 2) * (TW + 2) * 72 * sizeof(_Float16) * 2 + (TH + 4) * (TW + 4);      // hi/lo patch planes + the frame bytes
  if (cout_pad != 64 || !a.img || !a.w1a || !a.b1a) return hipErrorInvalidValue;
  auto k = conv_f16x2_kernel<64, 3, TH, TW, 2, 2, 2, 1, true, true, true>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  dim3 grid(((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH), 1, a.n_img), block(256);
  hipLaunchKernelGGL(k, grid, block, lds, s, a);
  return hipGetLastError();
}

hipError_t launch_conv_f16x2(ConvShape shape, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s) {
  switch (shape) {
    case CONV1B_FUSED:       return launch_f16_fused1b(cout_pad, a, s);
    case CONV_64_T8x32:
      if (tune_conv64() == 1) return launch_f16<64, 3, 4, 32, 2, 2, 2, 1>(pool, relu, cout_pad, a, s);
      return launch_f16<64, 3, 8, 32, 4, 1, 2, 2>(pool, relu, cout_pad, a, s);
    case CONV_128_T4x32:     return launch_f16<128, 3, 4, 32, 2, 2, 2, 2>(pool, relu, cout_pad, a, s);
    case CONV_128_T4x16:     return launch_f16<128, 3, 4, 16, 1, 4, 2, 1>(pool, relu, cout_pad, a, s);
    case CONV_256_1x1_T4x16: return launch_f16<256, 1, 4, 16, 1, 4, 2, 1>(pool, relu, cout_pad, a, s);
  }
  return hipErrorInvalidValue;
}

// host-side packing: hi/lo fp16 fragments of 2^SW * w
size_t packed_weight_halfs_f16x2(int cout_pad, int cin, int ks) { return (size_t)cout_pad * cin * ks * ks * 2; }

void pack_weights_f16x2(const float* w, int cout, int cin, int ks, int cout_pad, uint16_t* dst) {
  const int taps = ks * ks, kst = cin / 16;
  const float sw = (float)(1 << F16_SW);
  _Float16* d = reinterpret_cast<_Float16*>(dst);
  for (int nt = 0; nt < cout_pad / 32; ++nt)
    for (int tap = 0; tap < taps; ++tap)
      for (int k = 0; k < kst; ++k)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 8; ++i) {
            const int co = nt * 32 + (lane & 31);
            const int ci = k * 16 + 8 * (lane >> 5) + i;
            float v = co < cout ? w[((size_t)co * cin + ci) * taps + tap] * sw : 0.f;
            if (v > 65000.f) v = 65000.f;
            if (v < -65000.f) v = -65000.f;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            const size_t base = (((size_t)nt * taps + tap) * kst + k) * 128;
            d[(base + lane) * 8 + i] = h;
            d[(base + 64 + lane) * 8 + i] = l;
          }
}

}  // namespace d2fe
