// conv_pc.hip -- persistent producer/consumer version of the SuperPoint conv kernels (both precisions).
//
// PMC on the one-tile-per-block kernels (conv.hip / conv_f16.hip) showed the matrix pipe busy only 72 % (fp32) and
// 44 % (fp16x2) of the time: a block alternates a VALU/VMEM-heavy phase (stage the input patch: loads, conv1a
// evaluation, hi/lo split, LDS writes; then the epilogue) with its MFMA phase, and the co-resident blocks of a CU run in
// lock-step, so the phases do not overlap.  Here a workgroup is persistent (one per CU, grid-stride over tiles) and
// specialised by wave: waves [0, NC) are CONSUMERS (MFMA main loop + epilogue of tile k from LDS buffer k&1), waves
// [NC, 2NC) are PRODUCERS (stage tile k+1 into buffer (k+1)&1), one __syncthreads() per tile.  Consumer wave i and
// producer wave NC+i land on the same SIMD, whose matrix and vector pipes then run concurrently.
//
// Arithmetic is unchanged: the fp32 path is still the oracle's (ky,kx,ci) fmaf chain (bitwise), the fp16x2 path the
// same hi/lo split.  Tiling: 4x32 pixels x 64 channels (Cin 64, waves 2x2), 4x16 x 128 (Cin 128 and the 1x1 heads,
// waves 1x4); 2x2 max-pool stays lane-local for both tile widths.
#include "conv_common.h"
#include <cstdio>
#include <cstdlib>

namespace d2fe {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int PC_SA = 4, PC_SW = 8;   // fp16x2 power-of-two operand scalings (same as conv_f16.hip)

struct PcTile { int img, ty0, tx0, cg; };

// D2FE_ABLATE bit 256: workgroup 0 records s_memrealtime (100 MHz) marks of its first consumer and producer wave for the
// first 64 tiles -- [wave kind][tile][0 start, 1 main loop issued, 2 epilogue/staging issued, 3 past the barrier].
#ifdef D2FE_DEVTOOLS
__device__ unsigned long long g_pc_trace[2][64][4];
__device__ unsigned long long g_pc_trace_clk[4];   // (s_memtime, s_memrealtime) at tiles 2 and 10 -> shader clock
#endif

template <int TH, int TW>
__device__ __forceinline__ PcTile pc_decode(int t, int tiles_x, int tiles_y, int ncg) {
  PcTile r;
  r.cg = t % ncg; t /= ncg;
  r.tx0 = (t % tiles_x) * TW; t /= tiles_x;
  r.ty0 = (t % tiles_y) * TH;
  r.img = t / tiles_y;
  return r;
}

// pooled epilogue for 16-wide tiles: an m-tile is 2 rows x 16 columns, so both pooling directions stay inside one
// accumulator (horizontal: registers r, r+1; vertical: r, r+8).
template <int MT, int NT, bool RELU>
__device__ __forceinline__ void pc_epilogue_pool16(const ConvArgs& a, f32x16 (&acc)[MT][NT], float scale, int img, int ty0,
                                                   int tx0, int wm, int ntile0, int lane) {
  float* out = a.out + (size_t)img * a.out_img_stride + a.out_coff;
  const int Ho = a.H >> 1, Wo = a.W >> 1;
  const int cs = a.out_cstride;
  const bool cfull = (ntile0 + NT) * 32 <= a.cout_real;
  const unsigned cs4 = (unsigned)cs * 4u;
  const unsigned lane_off = (unsigned)(2 * (lane >> 5)) * cs4 + (unsigned)(lane & 31) * 4u;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int oy = (ty0 + (wm * MT + m) * 2) >> 1;
    if (cfull && ty0 + (wm * MT + m) * 2 + 1 < a.H && tx0 + 15 < a.W) {   // wave-uniform fast path, see conv_epilogue
      const unsigned base = ((unsigned)oy * (unsigned)Wo + (unsigned)(tx0 >> 1)) * cs4 + (unsigned)ntile0 * 128u;   // uniform
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        const int cu = (r & 3) + 8 * (r >> 2);
        const unsigned so = base + (unsigned)(cu >> 1) * cs4;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float m0 = fmaxf(acc[m][n][r], acc[m][n][r + 1]);
          const float m1 = fmaxf(acc[m][n][r + 8], acc[m][n][r + 9]);
          float v = fmaxf(m0, m1) * scale;
          if (RELU) v = v > 0.f ? v : 0.f;
          *reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (so + n * 128u + lane_off)) = v;
        }
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      const int col = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);   // even, < 16
      const int ox = (tx0 + col) >> 1;
      if (oy < Ho && ox < Wo) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = (ntile0 + n) * 32 + (lane & 31);
          const float m0 = fmaxf(acc[m][n][r], acc[m][n][r + 1]);
          const float m1 = fmaxf(acc[m][n][r + 8], acc[m][n][r + 9]);
          float v = fmaxf(m0, m1) * scale;
          if (RELU) v = v > 0.f ? v : 0.f;
          if (co < a.cout_real) out[((size_t)oy * Wo + ox) * cs + co] = v;
        }
      }
    }
  }
}

template <int MODE, int CIN, int KS, int TH, int TW, int WM, int WN, int MT, int NT, bool POOL, bool RELU, bool FUSE1A>
__global__ __launch_bounds__(WM * WN * 128) void conv_pc_kernel(ConvArgs a, int tiles_x, int tiles_y, int ncg, int total) {
  constexpr int NC = WM * WN;            // consumer waves (== producer waves)
  constexpr int NPT = NC * 64;           // producer threads
  constexpr int P = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1, NPIX = PH * PW;
  constexpr int TAPS = KS * KS;
  constexpr int CPF = CIN + 1;           // fp32 pixel stride (floats)
  constexpr int CPH = CIN + 8;           // fp16 pixel stride (halves), per plane
  constexpr int BUF_BYTES = MODE == 0 ? NPIX * CPF * 4 : NPIX * CPH * 2 * 2;
  static_assert(TH * TW == WM * MT * 32, "tile / wave mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < NC;
  // kernel-argument pointers copied to locals: inside the by-reference lambdas hipcc otherwise loses their
  // uniform/read-only provenance and turns the conv1a weight fetches into per-lane vector loads (measured: 480 VMEM/wave/tile)
  const float* __restrict__ w1a = a.w1a;
  const float* __restrict__ b1a = a.b1a;
  const uint8_t* __restrict__ img_base = a.img;
  const int img_stride_b = a.img_stride;
  const long img_istride = a.img_istride;
  const int aH = a.H, aW = a.W;

  // Fused conv1a (FUSE1A): the producers evaluate conv1a for the patch on the matrix pipe as well -- a [32 px x 10] x
  // [10 x 64] fp32 MFMA problem per 32 patch pixels (K = 9 taps padded to 10; v_mfma_f32_32x32x2_f32 IS the oracle's fmaf
  // chain in (ky,kx) order, so the values stay bit-exact).  The B fragments (weights) and the bias are loaded ONCE per
  // persistent workgroup and stay in 12 registers; the A fragment is 5 byte loads per lane per m-tile.
  float c1w[5][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  float c1b[2] = {0.f, 0.f};
  if constexpr (FUSE1A) {
    if (!consumer) conv1a_mfma_load_weights(w1a, b1a, lane, c1w, c1b);
  }

  int trace_k = 0;
#ifdef D2FE_DEVTOOLS
  const bool tracing = D2FE_ABL(a, 256) && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == NC);
  auto mark = [&](int slot) {
    if (tracing && trace_k < 64) g_pc_trace[wave == 0 ? 0 : 1][trace_k][slot] = wall_clock64();
  };
#else
  auto mark = [&](int) {};
#endif

  // ------------------------------------------------------------------------------------------------ producer
  auto stage = [&](const PcTile& T, int buf) {
    const int ptid = tid - NPT;
    unsigned char* base = lds_raw + (size_t)buf * BUF_BYTES;
    float* patch = reinterpret_cast<float*>(base);
    _Float16* hi = reinterpret_cast<_Float16*>(base);
    _Float16* lo = hi + NPIX * CPH;
    const float sa = (float)(1 << PC_SA);
    if constexpr (FUSE1A) {
      conv1a_mfma_stage<MODE, NPIX, PW, CPF, CPH, PC_SA>(img_base + (size_t)T.img * img_istride, img_stride_b, aH, aW, T.ty0, T.tx0,
                                                         c1w, c1b, lane, wave - NC, NC, patch, hi, lo);
    } else if (MODE == 0 && !D2FE_ABL(a, 2048)) {
      // fp32 patch = a plain copy: one LDS-DMA instruction (global_load_lds_dword, 64 lanes x 4 B) per 64 channels of a patch
      // pixel, no VGPR round trip and almost no VALU work -- every producer instruction costs the consumer wave of the same
      // SIMD issue time (measured: ~3.4 us per tile for ~1-2 k instructions).  The LDS destination of an LDS-DMA is
      // wave-uniform base + lane * 4, which is exactly one pixel's channel run at the odd pixel stride; pixels outside the
      // image (the conv's zero padding) are copied from a page of zeros.
      typedef __attribute__((address_space(1))) const void gvoid_t;
      typedef __attribute__((address_space(3))) void lvoid_t;
      const float* in = a.in + (size_t)T.img * a.in_img_stride + a.in_coff;
      const float* zeros = a.zeros;
      const int in_cs = a.in_cstride;
      for (int pix = wave - NC; pix < NPIX; pix += NC) {          // wave-uniform: the address arithmetic stays on the SALU
        const int gy = T.ty0 + pix / PW - P, gx = T.tx0 + pix % PW - P;
        const bool valid = gy >= 0 && gy < aH && gx >= 0 && gx < aW;
        const float* src = valid ? in + ((size_t)gy * aW + gx) * in_cs : zeros;
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += 64)
          __builtin_amdgcn_global_load_lds((gvoid_t*)(src + c0 + lane), (lvoid_t*)(patch + pix * CPF + c0), 4, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the copies must have landed before the tile barrier
    } else {
      const float* in = a.in + (size_t)T.img * a.in_img_stride + a.in_coff;
      constexpr int C4 = CIN / 4;
      constexpr int TOTAL = NPIX * C4;
      constexpr int ITERS = (TOTAL + NPT - 1) / NPT;
      constexpr int UNR = 8;
      for (int it0 = 0; it0 < ITERS; it0 += UNR) {
        f32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int idx = (it0 + u) * NPT + ptid;
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (it0 + u < ITERS && idx < TOTAL) {
            const int pix = idx / C4, c4 = idx % C4;
            const int gy = T.ty0 + pix / PW - P, gx = T.tx0 + pix % PW - P;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && !D2FE_ABL(a, 1))
              v[u] = *reinterpret_cast<const f32x4*>(in + ((size_t)gy * a.W + gx) * a.in_cstride + c4 * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int idx = (it0 + u) * NPT + ptid;
          if (it0 + u < ITERS && idx < TOTAL) {
            const int pix = idx / C4, c4 = idx % C4;
            if constexpr (MODE == 0) {
              float* d = patch + pix * CPF + c4 * 4;
              d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
            } else {
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
    }
  };

  // ------------------------------------------------------------------------------------------------ consumer
  // The weight stream is the same for every tile of a workgroup (same channel group), so the B-fragment buffers are
  // PERSISTENT: the loads wrap around at the end of a tile and the first k-steps of the next tile are already in
  // registers when it starts (no L2 round trip exposed per tile).  `wcg` = channel group the buffers currently hold.
  constexpr int F32_C8 = CIN / 8, F32_G = F32_C8 >= 16 ? 8 : 4, F32_GPT = F32_C8 / F32_G, F32_NG = TAPS * F32_GPT;
  static_assert(F32_C8 % F32_G == 0 && F32_NG % 2 == 0, "fp32 weight groups must tile the k loop evenly");
  constexpr int F16_KST = CIN / 16, F16_S = TAPS * F16_KST, F16_R = (F16_S % 9 == 0) ? 9 : 8;
  static_assert(F16_S % F16_R == 0, "ring depth must divide the k-steps of a tile");
  f32x4 bq[2][MODE == 0 ? F32_G : 1][NT];
  f16x8 ring[MODE == 1 ? F16_R : 1][NT][2];
  int wcg = -1;

  auto compute = [&](const PcTile& T, int buf) {
    const int wm = wave / WN, wn = wave % WN;
    const int ntile0 = T.cg * (WN * NT) + wn * NT;
    unsigned char* base = lds_raw + (size_t)buf * BUF_BYTES;
    f32x16 acc[MT][NT];
    const float bscale = MODE == 0 ? 1.0f : (float)(1 << (PC_SA + PC_SW));
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const float b = a.bias[(ntile0 + n) * 32 + (lane & 31)] * bscale;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = b;
    }
    if constexpr (MODE == 0) {
      const float* patch = reinterpret_cast<const float*>(base);
      constexpr int C8 = F32_C8, G = F32_G, GPT = F32_GPT, NG = F32_NG;
      int aoff[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int py, px;
        mtile_pixel<TW>(wm * MT + m, lane & 31, py, px);
        aoff[m] = (py * PW + px) * CPF + (lane >> 5);
      }
      const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack);
      const f32x4* wbase[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) wbase[n] = wp + (size_t)(ntile0 + n) * TAPS * C8 * 64 + lane;
      auto load_grp = [&](int b2, int grp) {
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n) bq[b2][j][n] = wbase[n][(size_t)(grp * G + j) * 64];
      };
      auto compute_grp = [&](int b2, int grp) {
        const int tap = grp / GPT, c80 = (grp % GPT) * G;
        const int tap_off = ((tap / KS) * PW + (tap % KS)) * CPF + c80 * 8;
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
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][q], bq[b2][j][n][q], acc[m][n], 0, 0, 0);
        }
      };
      if (wcg != T.cg) { load_grp(0, 0); wcg = T.cg; }
#pragma unroll 1
      for (int g = 0; g < NG; g += 2) {
        load_grp(1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_grp(0, g);
        load_grp(0, g + 2 < NG ? g + 2 : 0);      // wraps: group 0 of the NEXT tile is in flight during the epilogue
        __builtin_amdgcn_sched_barrier(0);
        compute_grp(1, g + 1);
      }
    } else {
      const _Float16* hi = reinterpret_cast<const _Float16*>(base);
      const _Float16* lo = hi + NPIX * CPH;
      constexpr int KST = F16_KST, S = F16_S, R = F16_R;
      int aoff[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int py, px;
        mtile_pixel<TW>(wm * MT + m, lane & 31, py, px);
        aoff[m] = (py * PW + px) * CPH + 8 * (lane >> 5);
      }
      const f16x8* wp = reinterpret_cast<const f16x8*>(a.wpack);
      const f16x8* wbase[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) wbase[n] = wp + (size_t)(ntile0 + n) * TAPS * KST * 128 + lane;
      // B fragments (hi, lo per n-tile) travel through a register ring R k-steps deep that wraps around tile boundaries:
      // the load for step (s+R-1) mod S is issued before the MFMAs of step s; A fragments are read one step ahead.
      auto load_step = [&](int slot, int st) {
        if (D2FE_ABL(a, 2)) st = 0;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          ring[slot][n][0] = wbase[n][(size_t)st * 128];
          ring[slot][n][1] = wbase[n][(size_t)st * 128 + 64];
        }
      };
      auto a_off = [&](int st) { const int tap = st / KST, ks = st % KST; return ((tap / KS) * PW + (tap % KS)) * CPH + ks * 16; };
      f16x8 ah[2][MT], al[2][MT];
      auto load_a = [&](int slot, int st) {
        const int o = a_off(st);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          ah[slot][m] = *reinterpret_cast<const f16x8*>(hi + aoff[m] + o);
          al[slot][m] = *reinterpret_cast<const f16x8*>(lo + aoff[m] + o);
        }
      };
      if (wcg != T.cg) {
#pragma unroll
        for (int st = 0; st < R - 1; ++st) load_step(st, st);
        wcg = T.cg;
      }
      load_a(0, 0);
#pragma unroll
      for (int st = 0; st < S; ++st) {
        load_step((st + R - 1) % R, (st + R - 1) % S);
        if (st + 1 < S) load_a((st + 1) & 1, st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st & 1][m], ring[st % R][n][0], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st & 1][m], ring[st % R][n][1], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st & 1][m], ring[st % R][n][0], acc[m][n], 0, 0, 0);
          }
      }
    }
    mark(1);
    const float oscale = MODE == 0 ? 1.0f : 1.0f / bscale;
    if constexpr (POOL && TW == 16)
      pc_epilogue_pool16<MT, NT, RELU>(a, acc, oscale, T.img, T.ty0, T.tx0, wm, ntile0, lane);
    else
      conv_epilogue<TW, MT, NT, POOL, RELU>(a, acc, oscale, T.img, T.ty0, T.tx0, wm, ntile0, lane);
  };

  // producers outrank consumers at the issue arbiter: their few VALU/VMEM/LDS instructions slot in between the consumer's
  // MFMAs instead of waiting for the consumer wave to stall (D2FE_ABLATE bit 512 switches this off for A/B measurements)
  if (!consumer && !D2FE_ABL(a, 512)) __builtin_amdgcn_s_setprio(3);

  // ------------------------------------------------------------------------------------------------ tile loop
  int t = blockIdx.x;
  if (t >= total) return;
  PcTile cur = pc_decode<TH, TW>(t, tiles_x, tiles_y, ncg);
  if (!consumer) stage(cur, 0);
  __syncthreads();
  int buf = 0;
  for (;;) {
    const int tn = t + gridDim.x;
    const bool more = tn < total;
    PcTile nxt = cur;
    if (more) nxt = pc_decode<TH, TW>(tn, tiles_x, tiles_y, ncg);
    mark(0);
#ifdef D2FE_DEVTOOLS
    if (tracing && wave == 0 && (trace_k == 2 || trace_k == 10)) {
      g_pc_trace_clk[trace_k == 2 ? 0 : 2] = clock64();
      g_pc_trace_clk[trace_k == 2 ? 1 : 3] = wall_clock64();
    }
#endif
    if (consumer) { if (!D2FE_ABL(a, 16)) compute(cur, buf); }
    else if (more && !D2FE_ABL(a, 8)) stage(nxt, buf ^ 1);
    mark(2);
    __syncthreads();
    mark(3);
    ++trace_k;
    if (!more) break;
    t = tn; cur = nxt; buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int MODE, int CIN, int KS, int TH, int TW, int WM, int WN, int MT, int NT, bool POOL, bool RELU, bool FUSE1A>
static hipError_t launch_pc_one(int cout_pad, const ConvArgs& a, hipStream_t s) {
  constexpr int BN = WN * NT * 32;
  constexpr int NPIX = (TH + KS - 1) * (TW + KS - 1);
  constexpr size_t buf = MODE == 0 ? (size_t)NPIX * (CIN + 1) * 4 : (size_t)NPIX * (CIN + 8) * 4;
  constexpr size_t lds = 2 * buf;
  static_assert(lds <= 163840, "double-buffered patch does not fit in LDS");
  if (cout_pad % BN) return hipErrorInvalidValue;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH, ncg = cout_pad / BN;
  const int total = tiles_x * tiles_y * ncg * a.n_img;
  const int ncu = a.ncu > 0 ? a.ncu : 256;     // ConvArgs::ncu: the handle's device (no process-wide cache: one process may drive several GPUs)
  auto k = conv_pc_kernel<MODE, CIN, KS, TH, TW, WM, WN, MT, NT, POOL, RELU, FUSE1A>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int grid = total < ncu ? total : ncu;
  hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 128), lds, s, a, tiles_x, tiles_y, ncg, total);
#ifdef D2FE_DEVTOOLS
  if (D2FE_ABL(a, 256)) {
    static int dumped = 0;
    const int sel = d2fe_dev_env("D2FE_PC_TRACE_CIN", 0);
    if (dumped < 2 && (!sel || sel == CIN * 10 + KS + (FUSE1A ? 100000 : 0)) && a.n_img > 1) {
      ++dumped;
      unsigned long long tr[2][64][4];
      if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_pc_trace), sizeof(tr)) == hipSuccess) {
        fprintf(stderr, "pc trace MODE %d CIN %d KS %d tile %dx%d fuse %d (10 ns ticks: start->loop, loop->end, end->barrier)\n", MODE, CIN, KS, TH, TW, (int)FUSE1A);
        unsigned long long ck[4];
        if (hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_pc_trace_clk), sizeof(ck)) == hipSuccess && ck[3] > ck[1])
          fprintf(stderr, "  s_memtime / s_memrealtime over tiles 2..10: %.3f cycles per 10 ns\n", (double)(ck[2] - ck[0]) / (double)(ck[3] - ck[1]));
        for (int k = 0; k < 24 && k * grid < total; ++k)
          fprintf(stderr, "  tile %2d  consumer %5lld %5lld %5lld | producer %5lld       %5lld | tile period %5lld\n", k,
                  (long long)(tr[0][k][1] - tr[0][k][0]), (long long)(tr[0][k][2] - tr[0][k][1]), (long long)(tr[0][k][3] - tr[0][k][2]),
                  (long long)(tr[1][k][2] - tr[1][k][0]), (long long)(tr[1][k][3] - tr[1][k][2]),
                  k ? (long long)(tr[0][k][0] - tr[0][k - 1][0]) : 0ll);
      }
    }
  }
#endif
  return hipGetLastError();
}

template <int MODE>
static hipError_t launch_pc_mode(ConvShape shape, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s) {
  switch (shape) {
    case CONV1B_FUSED:
      return launch_pc_one<MODE, 64, 3, 4, 32, 2, 2, 2, 1, true, true, true>(cout_pad, a, s);
    case CONV_64_T8x32:
      if (pool && relu) return launch_pc_one<MODE, 64, 3, 4, 32, 2, 2, 2, 1, true, true, false>(cout_pad, a, s);
      if (!pool && relu) return launch_pc_one<MODE, 64, 3, 4, 32, 2, 2, 2, 1, false, true, false>(cout_pad, a, s);
      break;
    case CONV_128_T4x32:   // conv3b: 2x2 pool on 4x16 tiles
      if (pool && relu) return launch_pc_one<MODE, 128, 3, 4, 16, 1, 4, 2, 1, true, true, false>(cout_pad, a, s);
      break;
    case CONV_128_T4x16:
      if (!pool && relu) return launch_pc_one<MODE, 128, 3, 4, 16, 1, 4, 2, 1, false, true, false>(cout_pad, a, s);
      break;
    case CONV_256_1x1_T4x16:
      if (!pool && !relu) return launch_pc_one<MODE, 256, 1, 4, 16, 1, 4, 2, 1, false, false, false>(cout_pad, a, s);
      break;
  }
  return hipErrorInvalidValue;
}

hipError_t launch_conv_pc(ConvShape shape, int precision, bool pool, bool relu, int cout_pad, const ConvArgs& a, hipStream_t s) {
  return precision == 0 ? launch_pc_mode<0>(shape, pool, relu, cout_pad, a, s)
                        : launch_pc_mode<1>(shape, pool, relu, cout_pad, a, s);
}

}  // namespace d2fe
