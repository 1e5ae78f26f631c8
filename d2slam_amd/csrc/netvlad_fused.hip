// netvlad_fused.hip -- MobileNetV2 inverted-residual blocks of the NetVLAD trunk as ONE kernel per block (gfx950).
//
// Replaces the per-layer launches of netvlad.hip for the layer patterns  [pw expand -> dw 3x3 -> pw project (+ residual)]  and
// [dw 3x3 -> pw project]  (optionally with the network's first 3x3 convolution from the u8 frame evaluated inside the block's
// input staging), and for the trunk's last 1x1 chained with the NetVLAD pre-projection.  Reference boundary:
// MobileNetVLADONNX::inference, d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:49-74 (one ONNX Runtime session run);
// the layer list is whatever d2fe_load_netvlad() was given.
//
// Why: the expanded tensor (6x the block's input) is the largest thing the trunk touches -- 472 MB for 32 images after the first
// expand at 640x480 -- and it exists only between two layers of the same block.  Here it lives in LDS, 16 channels at a time:
//
//   workgroup = 8 x 16 output pixels x all output channels, 4 waves.
//   X  [Cin + 4][XP]  input patch ((8-1)s+3) x ((16-1)s+3) pixels, channel-major in LDS, zeros outside the image (TF "SAME");
//                     row Cin is the in-image mask: the expand bias enters the GEMM as one more k-step against it, so that
//                     out-of-image pixels expand to act(0) = 0 without a compare per element
//   per chunk of 16 hidden channels:
//     expand  E[16][EP] = act([X; mask]^T [We; be])                      v_mfma_f32_16x16x4_f32, M = patch pixels, K = Cin + 4
//     dw      d = act(sum_9 E[c][p + tap] wd[tap][c] + bd[c])           v_pk_fma_f32 (two output rows per instruction); each lane
//                                                                       computes exactly the A-fragment elements (pixel = lane & 15,
//                                                                       channel = 4 ks + lane >> 4) it feeds to
//     project acc[pixel][cout] += d * Wp                                v_mfma_f32_16x16x4_f32, M = 128 output pixels, K = 16
//   epilogue: + bias (+ residual), activation, NHWC store.
//
// The kernels are instruction-bound, not memory- or MFMA-bound (rocprofv3 PMC, profiles/): everything that is not an MFMA or one of
// the 9 multiply-adds per depthwise output is overhead, so the code below keeps index arithmetic out of the loops (per-pixel
// staging loop, 32-bit offsets, weights as whole-workgroup coalesced records through LDS, v_med3 activations).
//
// LDS pitches: XP = 16 (mod 32) floats so that the two channel rows a 32-lane group of ds_read_b32 touches fall on
// disjoint bank halves; E uses the same pitch for stride 1 and an odd pitch for stride 2 (the 16 pixels of a lane group are
// then 2 apart: even banks for one channel row, odd for the other).
#include <algorithm>

#include "kernels.h"

namespace d2fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// act 0 none / 1 ReLU / 2 ReLU6 as one v_med3_f32 with the bounds in registers
__device__ __forceinline__ float nvf_lo(int act) { return act >= 1 ? 0.f : -__builtin_inff(); }
__device__ __forceinline__ float nvf_hi(int act) { return act == 2 ? 6.f : __builtin_inff(); }
__device__ __forceinline__ float nvf_clamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

int nv_block_ntiles(int cout);
constexpr int NVB_TH = 8, NVB_TW = 16;
// MODE 0: spatial block (depthwise 3x3 between expand and project); 1: the same with the network's first conv evaluated into X from
// the u8 frame; 2: no depthwise stage -- two chained 1x1 convolutions over a flat list of pixels (the trunk's last 1x1 + the
// NetVLAD pre-projection), 128 consecutive pixels per workgroup
__host__ __device__ constexpr int nvb_ih(int s) { return (NVB_TH - 1) * s + 3; }
__host__ __device__ constexpr int nvb_iw(int s) { return (NVB_TW - 1) * s + 3; }
__host__ __device__ constexpr int nvb_mt_in(int s, int mode) { return mode == 2 ? 8 : (nvb_ih(s) * nvb_iw(s) + 15) / 16; }
__host__ __device__ constexpr int nvb_xp(int s, int mode) { return (nvb_mt_in(s, mode) * 16 + 31) / 32 * 32 + 16; }
__host__ __device__ constexpr int nvb_ep(int s, int mode) { return (s == 1 || mode == 2) ? nvb_xp(s, mode) : nvb_mt_in(s, mode) * 16 + 1; }
// E is double-buffered (one barrier per chunk instead of two) when that still leaves room for two workgroups per CU
constexpr size_t NVB_LDS_2PER_CU = 80 * 1024;
constexpr int NVB_U8_PITCH = 40;      // MODE 1: u8 patch rows ((10-1)*2+3 = 21 rows x 37 bytes for a stride-2 first conv)
// per-chunk weight records as the host packs them and as they sit in LDS, sizes rounded up to 256 floats (one per thread and pass):
//   WE record: [Cin/4 + 1][64] expand B fragments; the last k-step carries the bias (k = 0 row: be[16], rows 1..3: zeros)
//   WD record: wd[9][16] (tap-major), bd[16], pad to 256, then [4][NT][64] project B fragments
__host__ __device__ constexpr int nvb_we_rec(int cin) { return ((cin / 4 + 1) * 64 + 255) / 256 * 256; }
__host__ __device__ constexpr int nvb_wd_rec(int nt) { return 256 + nt * 256; }
static size_t nvb_lds_bytes(int cin, int s, int mode, bool expand, int nbuf, int nt) {
  size_t fl = (size_t)(cin + (expand ? 4 : 0)) * nvb_xp(s, mode) + 2 * (size_t)nvb_wd_rec(nt) +
              (expand ? (size_t)nbuf * 16 * nvb_ep(s, mode) + 2 * (size_t)nvb_we_rec(cin) : 0);
  if (mode == 1) fl += 24 * NVB_U8_PITCH / 4 + 3 * 2 * 64;     // u8 patch + the first conv's B fragments
  return fl * sizeof(float);
}

template <bool EXPAND, int MODE, int S, int NT, int NBUF>
__global__ __launch_bounds__(256) void nv_block_kernel(NvBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IH = nvb_ih(S), IW = nvb_iw(S), NPX = MODE == 2 ? 128 : IH * IW, MT_IN = nvb_mt_in(S, MODE), XP = nvb_xp(S, MODE),
                EP = nvb_ep(S, MODE);
  constexpr int MPW = MT_IN / 4;                 // input m-tiles per wave (3, 9 or 2): exact, the wave loop has no remainder
  constexpr int GI = MPW % 3 == 0 ? 3 : 2;       // independent MFMA chains in flight in the expand stage
  static_assert(MT_IN % 4 == 0 && MPW % GI == 0, "m-tile split");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lp = lane & 15;
  const int Cin = a.Cin;
  float* X = lds;                                         // [Cin (+ 4: mask row, 3 zero rows)][XP]
  float* E = X + (Cin + (EXPAND ? 4 : 0)) * XP;           // [NBUF][16][EP]         (EXPAND only)
  float* WE = E + (EXPAND ? NBUF * 16 * EP : 0);          // [2][nvb_we_rec(Cin)]   (EXPAND only)
  float* WD = WE + (EXPAND ? 2 * nvb_we_rec(Cin) : 0);    // [2][nvb_wd_rec(NT)]

  const int tiles_x = (a.Wo + NVB_TW - 1) / NVB_TW;
  const int n = blockIdx.y;
  const int ty = MODE == 2 ? 0 : (int)blockIdx.x / tiles_x;
  const int oy0 = ty * NVB_TH, ox0 = MODE == 2 ? 0 : ((int)blockIdx.x - ty * tiles_x) * NVB_TW;
  const int iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;
  const int p0 = (int)blockIdx.x * 128;            // MODE 2: first pixel of the flat list (images folded into the list, gridDim.y == 1)

  // ---- weights: no per-lane global loads.  A chunk's weights are one contiguous record (packed by the host in exactly the layout the
  // lanes read), fetched by the whole workgroup with coalesced loads one chunk ahead and parked in two LDS double buffers; the
  // barrier that already separates expand from depthwise publishes them:
  //   top(ch): fetch WE(ch+1), WD(ch+1) -> regs | expand(ch) | regs -> WE[next] | barrier | dw+project(ch) | regs -> WD[next]
  // (WD[next] was last read by dw+project(ch-1), which every wave left before this chunk's barrier.)
  const int we_n = nvb_we_rec(Cin);
  constexpr int WD_N = nvb_wd_rec(NT);
  constexpr int WER = 8, WDR = WD_N / 256;                  // staged floats per thread (WER * 256 >= we_n: Cin <= 124, host check)
  const int wer_n = we_n >> 8;
  float wes[WER], wds[WDR];
  const int nchunk = a.Chid >> 4;
  const int ch0 = blockIdx.z * a.cpg, ch1 = min(nchunk, ch0 + a.cpg);      // this workgroup's share of the hidden channels
  auto fetch_w = [&](int ch) {
    if (EXPAND) {
      const float* wb = a.we + (size_t)ch * we_n + tid;
#pragma unroll
      for (int i = 0; i < WER; ++i) if (i < wer_n) wes[i] = wb[256 * i];
    }
    const float* wb = a.wp + (size_t)ch * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wds[i] = wb[256 * i];
  };
  auto store_we = [&](int buf) {
    if (!EXPAND) return;
    float* wl = WE + buf * we_n + tid;
#pragma unroll
    for (int i = 0; i < WER; ++i) if (i < wer_n) wl[256 * i] = wes[i];
  };
  auto store_wd = [&](int buf) {
    float* wl = WD + buf * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wl[256 * i] = wds[i];
  };
  if (ch0 < ch1) fetch_w(ch0);

  // ---- stage the input patch -----------------------------------------------------------------------------------------
  if (MODE == 1) {
    // first conv on the matrix pipe: M = patch pixels, K = 9 taps + 1 bias row (+ 2 zero rows), N = Cin channels.  A = (u8 - 128)/128
    // read from a u8 copy of the patch's receptive field in LDS; a pixel outside the conv's output zeroes its whole A row (bias row
    // included), so X is act(0) = 0 there -- the depthwise stage's zero padding.
    uint8_t* u8p = reinterpret_cast<uint8_t*>(WD + 2 * WD_N);
    float* w0l = reinterpret_cast<float*>(u8p + 24 * NVB_U8_PITCH);      // [3][2][64]
    const uint8_t* ip = a.img + (size_t)n * a.img_istride;
    const int cs = a.c0_stride;
    const int uy0 = iy0 * cs - a.c0_pt, ux0 = ix0 * cs - a.c0_pl;
    const int UH = (IH - 1) * cs + 3, UW = (IW - 1) * cs + 3;       // 21 x 37 for cs = 2 (host checks UH <= 24, UW <= NVB_U8_PITCH)
    for (int i = tid; i < UH * NVB_U8_PITCH; i += 256) {
      const int uy = i / NVB_U8_PITCH, ux = i - uy * NVB_U8_PITCH;
      const int yy = uy0 + uy, xx = ux0 + ux;
      u8p[i] = (ux < UW && yy >= 0 && yy < a.H0 && xx >= 0 && xx < a.W0) ? ip[yy * a.img_stride + xx] : (uint8_t)128;   // 128 -> exactly 0 after (x-128)/128
    }
    for (int i = tid; i < 3 * 2 * 64; i += 256) w0l[i] = a.w0[i];
    __syncthreads();
    const int nt0 = Cin >> 4;                     // 1 or 2 n-tiles of 16 channels
    const float lo0 = nvf_lo(a.act0), hi0 = nvf_hi(a.act0);
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      const int mt = wave + 4 * j;
      const int p = mt * 16 + lp;
      const int iy = p / IW, ix = p - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool ok = p < NPX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const uint8_t* up = u8p + (p < NPX ? (iy * cs) * NVB_U8_PITCH + ix * cs : 0);
      // this lane's three k-steps: k = 4 ks + lq; k < 9 a tap, k == 9 the bias row (A = 1), k > 9 nothing
      float av[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int k = ks * 4 + lq;
        const int toff = k < 9 ? (k / 3) * NVB_U8_PITCH + (k % 3) : 0;
        const float t = ((float)up[toff] - 128.0f) * 0.0078125f;
        av[ks] = ok ? (k < 9 ? t : (k == 9 ? 1.f : 0.f)) : 0.f;
      }
      for (int t = 0; t < nt0; ++t) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], w0l[(ks * 2 + t) * 64 + lane], c, 0, 0, 0);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = nvf_clamp(c[r], lo0, hi0);
        *reinterpret_cast<f32x4*>(X + (t * 16 + lp) * XP + mt * 16 + lq * 4) = o;
      }
    }
  } else {
    // one pixel per thread and pass, its channels in an inner loop: no division by a run-time channel count anywhere.
    // The producer may have split its hidden channels over `in_slabs` workgroup groups: the input is the sum of its partial slabs.
    const int c4n = Cin >> 2;
    const float* ip = MODE == 2 ? a.in : a.in + (size_t)n * a.H * a.W * Cin;
    constexpr int PXT = MODE == 2 ? 128 : MT_IN * 16;
    // MODE 2: 128 pixels, two threads per pixel (each half of the channels)
    const int half = MODE == 2 ? (tid >> 7) : 0;
    const int cbeg = MODE == 2 ? half * (c4n >> 1) : 0, cend = MODE == 2 ? (half ? c4n : (c4n >> 1)) : c4n;
    for (int p = MODE == 2 ? (tid & 127) : tid; p < PXT; p += 256) {
      bool ok; int off;
      if (MODE == 2) {
        ok = p0 + p < a.P; off = (p0 + p) * Cin;
      } else {
        const int iy = p / IW, ix = p - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        ok = p < NPX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        off = (gy * a.W + gx) * Cin;
      }
      const float* src = ip + (ok ? off : 0);
      float* d = X + p;
      for (int c4 = cbeg; c4 < cend; ++c4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + c4 * 4);
        for (int sl = 1; sl < a.in_slabs; ++sl) v += *reinterpret_cast<const f32x4*>(src + (size_t)sl * a.in_slab_stride + c4 * 4);
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        float* dc = d + (c4 * 4) * XP;
        dc[0] = v[0]; dc[XP] = v[1]; dc[2 * XP] = v[2]; dc[3 * XP] = v[3];
      }
      if (EXPAND && half == 0) {
        float* dm = d + Cin * XP;
        dm[0] = ok ? 1.f : 0.f; dm[XP] = 0.f; dm[2 * XP] = 0.f; dm[3 * XP] = 0.f;
      }
    }
  }
  if (ch0 < ch1) { store_we(0); store_wd(0); }
  __syncthreads();                                   // X (with its mask row) and the first chunk's weights are complete

  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kse1 = (Cin >> 2) + 1;                   // k-steps of the expand GEMM incl. the bias step
  const float lo_e = nvf_lo(a.act_e), hi_e = nvf_hi(a.act_e), lo_d = nvf_lo(a.act_d), hi_d = nvf_hi(a.act_d);

  for (int ch = ch0; ch < ch1; ++ch) {
    const int wb_i = (ch - ch0) & 1;
    const float* Esrc;
    if (ch + 1 < ch1) fetch_w(ch + 1);
    if (EXPAND) {
      float* Eb = E + (NBUF == 2 ? wb_i : 0) * 16 * EP;
      const float* wl = WE + wb_i * we_n + lane;
      if (NBUF == 1 && ch > ch0) __syncthreads();            // everybody is done reading the previous chunk
#pragma unroll
      for (int g = 0; g < MPW / GI; ++g) {
        f32x4 c[GI];
        const float* xa[GI];
#pragma unroll
        for (int j = 0; j < GI; ++j) {
          c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          xa[j] = X + lq * XP + (wave + 4 * (g * GI + j)) * 16 + lp;
        }
#pragma unroll 2
        for (int ks = 0; ks < kse1; ++ks) {
          const float wv = wl[ks * 64];
#pragma unroll
          for (int j = 0; j < GI; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j][ks * 4 * XP], wv, c[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < GI; ++j) {
          const int mt = wave + 4 * (g * GI + j);
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = nvf_clamp(c[j][r], lo_e, hi_e);
          float* e = Eb + lp * EP + mt * 16 + lq * 4;
          if (EP % 4 == 0) *reinterpret_cast<f32x4*>(e) = o;
          else { e[0] = o[0]; e[1] = o[1]; e[2] = o[2]; e[3] = o[3]; }
        }
      }
      if (ch + 1 < ch1) store_we(wb_i ^ 1);
      __syncthreads();
      Esrc = Eb;
    } else {
      Esrc = X + ch * 16 * XP;
    }
    constexpr int PITCH = EXPAND ? EP : XP;
    const float* wd = WD + wb_i * WD_N;
    const float* wpl = wd + 256 + lane;
    // depthwise on v_pk_fma_f32 with two HIDDEN CHANNELS (k-steps 2 kp, 2 kp + 1) of one pixel as the halves: their E values are a
    // constant 4 * PITCH apart and their weights 4 floats, so both operands are real register pairs straight out of the LDS reads.
    // (Pairing the wave's two output rows instead makes the rows they share -- row ky + 1 of the first is row ky of the second -- one
    // load feeding two different pair positions: two v_mov per FMA in a loop that is VALU-bound.)  Same fma chain per channel.
    const float* eb0 = MODE == 2 ? Esrc + (wave * 2) * 16 + lp : Esrc + (wave * 2 * S) * IW + lp * S;
    constexpr int ROW2 = MODE == 2 ? 16 : S * IW;          // distance between the two rows' patch pixels
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      const float* ea = eb0 + (kp * 8 + lq) * PITCH;
      const float* ec = ea + 4 * PITCH;
      f32x2 d0, d1;
      if (MODE == 2) {
        d0 = f32x2{ea[0], ec[0]}; d1 = f32x2{ea[ROW2], ec[ROW2]};
      } else {
        const float* wk = wd + kp * 8 + lq;
        d0 = f32x2{wk[144], wk[148]}; d1 = d0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const f32x2 w = {wk[(ky * 3 + kx) * 16], wk[(ky * 3 + kx) * 16 + 4]};
            d0 = __builtin_elementwise_fma(f32x2{ea[ky * IW + kx], ec[ky * IW + kx]}, w, d0);
            d1 = __builtin_elementwise_fma(f32x2{ea[ROW2 + ky * IW + kx], ec[ROW2 + ky * IW + kx]}, w, d1);
          }
        d0[0] = nvf_clamp(d0[0], lo_d, hi_d); d0[1] = nvf_clamp(d0[1], lo_d, hi_d);
        d1[0] = nvf_clamp(d1[0], lo_d, hi_d); d1[1] = nvf_clamp(d1[1], lo_d, hi_d);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float wv = wpl[((kp * 2 + h) * NT + t) * 64];
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d0[h], wv, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[h], wv, acc[1][t], 0, 0, 0);
        }
    }
    if (ch + 1 < ch1) store_wd(wb_i ^ 1);
    if (!EXPAND && ch + 1 < ch1) __syncthreads();   // no expand stage, hence no barrier of its own between the chunks
  }

  // ---- epilogue: C layout col = lane & 15 (channel within the n-tile), row = 4 (lane >> 4) + r (pixel = column ox of row oy) ------
  // hidden-channel group g > 0 stores its bare partial sum into slab g; group 0 adds the bias and the residual.  32-bit offsets.
  const bool lead = blockIdx.z == 0;
  const int Cout = a.Cout;
  float* op = a.out + (size_t)blockIdx.z * a.out_slab_stride + (MODE == 2 ? 0 : (size_t)n * a.Ho * a.Wo * Cout);
  const float* rp = (a.res && lead) ? a.res + (size_t)n * a.Ho * a.Wo * Cout : nullptr;
  const float lo_p = nvf_lo(a.act_p), hi_p = nvf_hi(a.act_p);
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) {
    int base, nvalid;
    if (MODE == 2) {
      const int p = p0 + (wave * 2 + m2) * 16 + lq * 4;
      nvalid = (int)min((long)4, a.P - p);
      base = p * Cout + lp;
    } else {
      const int gy = oy0 + wave * 2 + m2, gx = ox0 + lq * 4;
      nvalid = gy < a.Ho ? min(4, a.Wo - gx) : 0;
      base = (gy * a.Wo + gx) * Cout + lp;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t * 16 + lp >= Cout) continue;
      const float bv = lead ? a.bp[t * 16 + lp] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r >= nvalid) continue;
        const int o = base + r * Cout + t * 16;
        float v = acc[m2][t][r] + bv;
        if (rp) { for (int sl = 0; sl < a.res_slabs; ++sl) v += rp[(size_t)sl * a.res_slab_stride + o]; }
        op[o] = nvf_clamp(v, lo_p, hi_p);
      }
    }
  }
}

template <bool EXPAND, int MODE, int S, int NT, int NBUF>
static hipError_t launch_block_b(const NvBlockArgs& a, int n, int groups, hipStream_t s) {
  const size_t lds = nvb_lds_bytes(a.Cin, S, MODE, EXPAND, NBUF, NT);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  auto k = nv_block_kernel<EXPAND, MODE, S, NT, NBUF>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const long tiles = MODE == 2 ? (a.P + 127) / 128 : (long)((a.Wo + NVB_TW - 1) / NVB_TW) * ((a.Ho + NVB_TH - 1) / NVB_TH);
  hipLaunchKernelGGL(k, dim3((unsigned)tiles, MODE == 2 ? 1 : n, groups), dim3(256), lds, s, a);
  return hipGetLastError();
}
template <bool EXPAND, int MODE, int S, int NT>
static hipError_t launch_block_t(const NvBlockArgs& a, int n, int groups, hipStream_t s) {
  if (EXPAND && (S == 1 || MODE == 2) && nvb_lds_bytes(a.Cin, S, MODE, true, 2, NT) <= NVB_LDS_2PER_CU) return launch_block_b<EXPAND, MODE, S, NT, 2>(a, n, groups, s);
  return launch_block_b<EXPAND, MODE, S, NT, 1>(a, n, groups, s);
}

template <bool EXPAND, int MODE, int S>
static hipError_t launch_block_nt(const NvBlockArgs& a, int n, int groups, hipStream_t s) {
  const int nt = (a.Cout + 15) / 16;
  if (nt == 1) return launch_block_t<EXPAND, MODE, S, 1>(a, n, groups, s);
  if (nt == 2) return launch_block_t<EXPAND, MODE, S, 2>(a, n, groups, s);
  if (nt <= 4) return launch_block_t<EXPAND, MODE, S, 4>(a, n, groups, s);
  if (nt <= 8) return launch_block_t<EXPAND, MODE, S, 8>(a, n, groups, s);
  return hipErrorInvalidValue;
}

__host__ __device__ constexpr int nvx_min_waves(int s, int nt, int nj) { return (s == 2 && nt == 2 && nj == 2) ? 3 : 1; }      // <2, 4, 2> at three waves spills 18 registers: 49 vs 44.6 us (32 images), 28.7 vs 20.4 (one)
// ---- expand -> depthwise -> project with the block INPUT in registers ------------------------------------------------------------------
// What limits the LDS-resident form above on the low-resolution layers is occupancy: the input patch (Cin x 208 floats) plus two
// E buffers leave room for two workgroups per CU, i.e. ~1.3 waves per SIMD on average, and every LDS round trip and barrier is
// exposed (PMC: waves wait two thirds of their lifetime).  A wave only ever multiplies ITS OWN patch m-tiles in the expand stage, so
// the input it needs is MPW x ceil(Cin/16) float4 per lane, loaded once, straight from HBM/L2 into registers -- no staging pass, no
// A-operand LDS reads, and the workgroup's LDS shrinks to E plus the weight double buffers (~40 KB: four workgroups per CU).
// K order: a float4 hands a lane 4 consecutive channels of its pixel, so k-step (j, e) carries channel (lq + 4 j) * 4 + e
// (zero rows beyond Cin); pack_nv_expand_perm packs the B fragments to match.
// Tile shape is a run-time choice (th x tw <= 128 output pixels, row-major flat index -> m-tiles), so that 15x20 or 30x40 maps are
// cut into 5x20 / 6x20 tiles without padding instead of 47 % / 78 % useful 8x16 tiles; the lane -> pixel maps are computed once.
template <int S, int NT, int NJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(nvx_min_waves(S, NT, NJ)))) void nv_xblock_kernel(NvBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT_IN = nvb_mt_in(S, 0), EP = nvb_ep(S, 0), NBUF = S == 1 ? 2 : 1;
  constexpr int MPW = MT_IN / 4, GI = 3;
  // NJ = 0: Cin = 8, a float2 per lane (channels 2 lq, 2 lq + 1) and two k-steps instead of a float4 with half the lane groups zero and four
  constexpr int KS = NJ ? NJ * 4 : 2;
  constexpr int KSE1 = KS + 1, WE_N = (KSE1 * 64 + 255) / 256 * 256, WD_N = nvb_wd_rec(NT), WER = WE_N / 256, WDR = WD_N / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lp = lane & 15;
  float* E = lds;                         // [NBUF][16][EP]
  float* WE = E + NBUF * 16 * EP;         // [2][WE_N]
  float* WD = WE + 2 * WE_N;              // [2][WD_N]
  const int Cin = a.Cin, th = a.th, tw = a.tw;
  const int iw = (tw - 1) * S + 3, npx = ((th - 1) * S + 3) * iw;
  const int tiles_x = (a.Wo + tw - 1) / tw;
  const int n = blockIdx.y;
  const int ty = (int)blockIdx.x / tiles_x;
  const int oy0 = ty * th, ox0 = ((int)blockIdx.x - ty * tiles_x) * tw;
  const int iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;
  const int nchunk = a.Chid >> 4;
  const int ch0 = blockIdx.z * a.cpg, ch1 = min(nchunk, ch0 + a.cpg);
  unsigned long long* stamp = a.stamps ? a.stamps + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 : nullptr;
  int stamp_i = 0;
  auto STAMP = [&]() { if (stamp && tid == 0 && stamp_i < 32) stamp[stamp_i] = wall_clock64(); ++stamp_i; };
  STAMP();
  // the project bias and the residual, first thing in the kernel instead of in the epilogue where their latency has nothing to hide behind
  // (NT = 8: no registers for it)
  const bool lead = blockIdx.z == 0;
  const int Cout = a.Cout;
  const float* rp = (a.res && lead) ? a.res + (size_t)n * a.Ho * a.Wo * Cout : nullptr;
  constexpr bool RES_EARLY = NT <= 4 && S == 1;        // a stride-2 block has no residual (shapes differ)
  float resv[2][4][RES_EARLY ? NT : 1], res1[2][4][RES_EARLY ? NT : 1], bvv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bvv[t] = a.bp[t * 16 + lp];           // padded to the n-tiles by the host
  if (RES_EARLY && rp) {
    int rb[2][4];
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = (wave * 2 + m2) * 16 + lq * 4 + r;
        const int oy = (int)(((unsigned)q * a.inv_tw) >> 20), ox = q - oy * tw;
        const int gy = oy0 + oy, gx = ox0 + ox;
        rb[m2][r] = (q < th * tw && gy < a.Ho && gx < a.Wo) ? (gy * a.Wo + gx) * Cout + lp : -1;
      }
    auto batch = [&](float (&dst)[2][4][RES_EARLY ? NT : 1], int sl, bool add) {
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < (RES_EARLY ? NT : 1); ++t) {
            const float v = rp[(size_t)sl * a.res_slab_stride + ((rb[m2][r] >= 0 && t * 16 + lp < Cout) ? rb[m2][r] + t * 16 : 0)];
            dst[m2][r][t] = add ? dst[m2][r][t] + v : v;
          }
    };
    batch(resv, 0, false);
    if (a.res_slabs > 1) batch(res1, 1, false);
    for (int sl = 2; sl < a.res_slabs; ++sl) batch(res1, sl, true);
  }
  float wes[WER], wds[WDR];
  // weight records travel global -> registers -> LDS; the loads for chunk ch + 2 are issued right after the registers holding chunk
  // ch + 1 have been written to LDS, so a full chunk of work covers their L2 latency (a chunk's MFMA time is ~0.5 us, a miss ~1-2 us)
  auto fetch_we = [&](int ch) {
    const float* wb = a.we + (size_t)ch * WE_N + tid;
#pragma unroll
    for (int i = 0; i < WER; ++i) wes[i] = wb[256 * i];
  };
  auto fetch_wd = [&](int ch) {
    const float* wc = a.wp + (size_t)ch * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wds[i] = wc[256 * i];
  };
  auto store_we = [&](int buf) {
    float* wl = WE + buf * WE_N + tid;
#pragma unroll
    for (int i = 0; i < WER; ++i) wl[256 * i] = wes[i];
  };
  auto store_wd = [&](int buf) {
    float* wm = WD + buf * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wm[256 * i] = wds[i];
  };
  if (ch0 < ch1) { fetch_we(ch0); fetch_wd(ch0); }

  // ---- the wave's MPW patch m-tiles: pixel mt*16 + lp, channels (lq + 4 j) * 4 .. + 3, summed over the producer's partial slabs ----
  // Every global load of the prologue is issued before anything waits on one, in batches per slab: a run-time loop over the partial slabs
  // around each load costs one memory round trip per iteration (tools/nv_stamps.py: 3 us of prologue, 3 us of epilogue per workgroup).
  float xr[MPW][KS];
  float one[MPW];
  {
    const float* ip = a.in + (size_t)n * a.H * a.W * Cin;        // one slab (the launcher rejects a split input: run_netvlad sums the slabs first)
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int p = (wave + 4 * m) * 16 + lp;
      const int iy = (int)(((unsigned)p * a.inv_iw) >> 20), ix = p - iy * iw;     // p / iw without the 25-instruction division
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool ok = p < npx && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      one[m] = ok ? 1.f : 0.f;
      const float* src = ip + (ok ? (gy * a.W + gx) * Cin : 0);
      if constexpr (NJ == 0) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(src + lq * 2);
        xr[m][0] = v[0]; xr[m][1] = v[1];
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((lq + 4 * j) * 4 < Cin ? (lq + 4 * j) * 4 : 0));
          xr[m][j * 4] = v[0]; xr[m][j * 4 + 1] = v[1]; xr[m][j * 4 + 2] = v[2]; xr[m][j * 4 + 3] = v[3];
        }
      }
    }
  }
  // ---- the wave's two output m-tiles (flat index q = (2 wave + m2) * 16 + lp over the th x tw tile) -> patch pixel of tap (0,0) ----
  int ebase[2];
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) {
    const int q = (wave * 2 + m2) * 16 + lp;
    const int oy = (int)(((unsigned)q * a.inv_tw) >> 20), ox = q - oy * tw;
    ebase[m2] = q < th * tw ? (oy * S) * iw + ox * S : 0;
  }
  STAMP();
  if (ch0 < ch1) { store_we(0); store_wd(0); }
  if (ch0 + 1 < ch1) { fetch_we(ch0 + 1); fetch_wd(ch0 + 1); }
  STAMP();
  __syncthreads();
  STAMP();

  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float lo_e = nvf_lo(a.act_e), hi_e = nvf_hi(a.act_e), lo_d = nvf_lo(a.act_d), hi_d = nvf_hi(a.act_d);
  if (RES_EARLY && rp && a.res_slabs > 1) {
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < (RES_EARLY ? NT : 1); ++t) resv[m2][r][t] += res1[m2][r][t];
  }
#pragma unroll
  for (int m = 0; m < MPW; ++m)
#pragma unroll
    for (int k = 0; k < KS; ++k)
      if (!(one[m] != 0.f && (NJ == 0 || (lq + 4 * (k >> 2)) * 4 < Cin))) xr[m][k] = 0.f;

  for (int ch = ch0; ch < ch1; ++ch) {
    const int wb_i = (ch - ch0) & 1;
    float* Eb = E + (NBUF == 2 ? wb_i : 0) * 16 * EP;
    const float* wl = WE + wb_i * WE_N + lane;
    if (NBUF == 1 && ch > ch0) __syncthreads();            // everybody is done reading the previous chunk
#pragma unroll
    for (int g = 0; g < MPW / GI; ++g) {
      f32x4 c[GI];
#pragma unroll
      for (int i = 0; i < GI; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const float wv = wl[k * 64];
#pragma unroll
        for (int i = 0; i < GI; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[g * GI + i][k], wv, c[i], 0, 0, 0);
      }
      {
        const float wv = wl[(KSE1 - 1) * 64];          // bias step against the in-image mask
#pragma unroll
        for (int i = 0; i < GI; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(one[g * GI + i], wv, c[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < GI; ++i) {
        const int mt = wave + 4 * (g * GI + i);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = nvf_clamp(c[i][r], lo_e, hi_e);
        float* e = Eb + lp * EP + mt * 16 + lq * 4;
        if (EP % 4 == 0) *reinterpret_cast<f32x4*>(e) = o;
        else { e[0] = o[0]; e[1] = o[1]; e[2] = o[2]; e[3] = o[3]; }
      }
    }
    STAMP();
    if (ch + 1 < ch1) store_we(wb_i ^ 1);              // WE[wb_i ^ 1] was last read by the previous chunk's expand stage
    if (ch + 2 < ch1) fetch_we(ch + 2);
    STAMP();
    __syncthreads();
    STAMP();
    const float* wd = WD + wb_i * WD_N;
    const float* wpl = wd + 256 + lane;
    // depthwise: two hidden channels (k-steps 2 kp, 2 kp + 1) of one pixel as the halves of v_pk_fma_f32 (see nv_block_kernel); three row
    // pointers per m-tile, kx and the channel as immediates
    const float* e0 = Eb + lq * EP + ebase[0];
    const float* e1 = Eb + lq * EP + ebase[1];
    const float* r0[3] = {e0, e0 + iw, e0 + 2 * iw};
    const float* r1[3] = {e1, e1 + iw, e1 + 2 * iw};
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      // this lane group's two channels (kp * 8 + lq, + 4): [tap][2] + bias[2] = 20 floats, five ds_read_b128 (pack_nv_dwproj_x) instead of 20 ds_read_b32
      float wq[20];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wd + (lq * 2 + kp) * 20 + i * 4);
        wq[i * 4] = v[0]; wq[i * 4 + 1] = v[1]; wq[i * 4 + 2] = v[2]; wq[i * 4 + 3] = v[3];
      }
      f32x2 d0 = {wq[18], wq[19]}, d1 = d0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x2 w = {wq[(ky * 3 + kx) * 2], wq[(ky * 3 + kx) * 2 + 1]};
          d0 = __builtin_elementwise_fma(f32x2{r0[ky][kp * 8 * EP + kx], r0[ky][(kp * 8 + 4) * EP + kx]}, w, d0);
          d1 = __builtin_elementwise_fma(f32x2{r1[ky][kp * 8 * EP + kx], r1[ky][(kp * 8 + 4) * EP + kx]}, w, d1);
        }
      d0[0] = nvf_clamp(d0[0], lo_d, hi_d); d0[1] = nvf_clamp(d0[1], lo_d, hi_d);
      d1[0] = nvf_clamp(d1[0], lo_d, hi_d); d1[1] = nvf_clamp(d1[1], lo_d, hi_d);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float wv = wpl[((kp * 2 + h) * NT + t) * 64];
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d0[h], wv, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[h], wv, acc[1][t], 0, 0, 0);
        }
    }
    STAMP();
    if (ch + 1 < ch1) store_wd(wb_i ^ 1);              // WD[wb_i ^ 1]: everybody passed this chunk's barrier, so chunk ch - 1 is done with it
    if (ch + 2 < ch1) fetch_wd(ch + 2);
    STAMP();
  }

  // ---- epilogue: C row = pixel q = (2 wave + m2) * 16 + 4 lq + r of the flat tile, col = channel lp of n-tile t ----------------------
  float* op = a.out + (size_t)blockIdx.z * a.out_slab_stride + (size_t)n * a.Ho * a.Wo * Cout;
  const float lo_p = nvf_lo(a.act_p), hi_p = nvf_hi(a.act_p);
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = (wave * 2 + m2) * 16 + lq * 4 + r;
      const int oy = (int)(((unsigned)q * a.inv_tw) >> 20), ox = q - oy * tw;
      const int gy = oy0 + oy, gx = ox0 + ox;
      if (q >= th * tw || gy >= a.Ho || gx >= a.Wo) continue;
      const int base = (gy * a.Wo + gx) * Cout + lp;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t * 16 + lp >= Cout) continue;
        const int o = base + t * 16;
        float v = acc[m2][t][r] + (lead ? bvv[t] : 0.f);
        if constexpr (RES_EARLY) { if (rp) v += resv[m2][r][t]; }
        else if (rp) { for (int sl = 0; sl < a.res_slabs; ++sl) v += rp[(size_t)sl * a.res_slab_stride + o]; }
        op[o] = nvf_clamp(v, lo_p, hi_p);
      }
    }
  }
  STAMP();
}

// tile shape for a th x tw <= 128 pixel output tile whose patch fits the kernel's m-tiles: maximise the useful fraction of the MFMA rows
void nv_xblock_tile(int Ho, int Wo, int stride, int* th_out, int* tw_out) {
  const int lim = nvb_mt_in(stride, 0) * 16;
  double best = -1; int bth = 8, btw = 16;
  for (int th = 1; th <= 32; ++th)
    for (int tw = 4; tw <= 64; ++tw) {
      if (th * tw > 128) continue;
      if (((th - 1) * stride + 3) * ((tw - 1) * stride + 3) > lim) continue;
      const long tiles = (long)((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
      const double eff = (double)Ho * Wo / (double)(tiles * 128);     // a workgroup always pays for 8 output m-tiles
      const double halo = (double)(th * tw) / (double)(((th - 1) * stride + 3) * ((tw - 1) * stride + 3));
      const double score = eff * (0.75 + 0.25 * halo);
      if (score > best + 1e-9) { best = score; bth = th; btw = tw; }
    }
  *th_out = bth; *tw_out = btw;
}
int nv_xblock_nj(int cin) { if (cin == 8) return 0; const int nj = (cin + 15) / 16; return nj <= 2 ? nj : nj <= 4 ? 4 : nj <= 7 ? 7 : -1; }     // 0: the Cin = 8 form
bool nv_xblock_supported(int cin, int chid, int cout, int stride) {
  return (stride == 1 || stride == 2) && !(cin & 3) && !(chid & 15) && cin >= 4 && nv_xblock_nj(cin) >= 0 && nv_block_ntiles(cout) > 0 &&
         (stride == 1 || nv_xblock_nj(cin) <= 2);      // stride 2 keeps 9 m-tiles of input per lane: Cin <= 32
}
// expand record for nv_xblock_kernel: k-step (j, e) holds input channel (lq + 4 j) * 4 + e for lane group lq (zero beyond cin); bias step last
size_t pack_nv_expand_perm_floats(int chid, int cin) { const int nj = nv_xblock_nj(cin), ks = nj ? nj * 4 : 2; return (size_t)(chid / 16) * (((ks + 1) * 64 + 255) / 256 * 256); }
void pack_nv_expand_perm(const float* w /*[chid][cin]*/, const float* b, int chid, int cin, float* dst) {
  const int nj = nv_xblock_nj(cin), ks = nj ? nj * 4 : 2, rec = ((ks + 1) * 64 + 255) / 256 * 256;
  for (int ch = 0; ch < chid / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < rec; ++i) d[i] = 0.f;
    if (nj == 0)         // Cin = 8: k-step e holds channel 2 lq + e
      for (int e = 0; e < 2; ++e)
        for (int l = 0; l < 64; ++l) d[e * 64 + l] = w[(size_t)(ch * 16 + (l & 15)) * cin + (l >> 4) * 2 + e];
    for (int j = 0; j < nj; ++j)
      for (int e = 0; e < 4; ++e)
        for (int l = 0; l < 64; ++l) {
          const int k = ((l >> 4) + 4 * j) * 4 + e;
          d[(j * 4 + e) * 64 + l] = k < cin ? w[(size_t)(ch * 16 + (l & 15)) * cin + k] : 0.f;
        }
    for (int c = 0; c < 16; ++c) d[ks * 64 + c] = b[ch * 16 + c];
  }
}
template <int S, int NT, int NJ>
static hipError_t launch_xblock_t(const NvBlockArgs& a, int n, int groups, hipStream_t s) {
  constexpr int EP = nvb_ep(S, 0), NBUF = S == 1 ? 2 : 1, WE_N = (((NJ ? NJ * 4 : 2) + 1) * 64 + 255) / 256 * 256;
  const size_t lds = sizeof(float) * ((size_t)NBUF * 16 * EP + 2 * WE_N + 2 * nvb_wd_rec(NT));
  auto k = nv_xblock_kernel<S, NT, NJ>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const long tiles = (long)((a.Wo + a.tw - 1) / a.tw) * ((a.Ho + a.th - 1) / a.th);
  hipLaunchKernelGGL(k, dim3((unsigned)tiles, n, groups), dim3(256), lds, s, a);
  return hipGetLastError();
}
template <int S, int NJ>
static hipError_t launch_xblock_nt(const NvBlockArgs& a, int n, int groups, hipStream_t s) {
  switch (nv_block_ntiles(a.Cout)) {
    case 1: return launch_xblock_t<S, 1, NJ>(a, n, groups, s);
    case 2: return launch_xblock_t<S, 2, NJ>(a, n, groups, s);
    case 4: return launch_xblock_t<S, 4, NJ>(a, n, groups, s);
    case 8: return launch_xblock_t<S, 8, NJ>(a, n, groups, s);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_nv_xblock(const NvBlockArgs& a_in, int n, int groups, hipStream_t s) {
  NvBlockArgs a = a_in;
  const int iw = (a.tw - 1) * a.stride + 3;
  if (a.tw < 1 || a.tw > 128 || iw > 1024 || a.in_slabs > 1) return hipErrorInvalidValue;
  a.inv_iw = ((1u << 20) + iw - 1) / iw; a.inv_tw = ((1u << 20) + a.tw - 1) / a.tw;     // exact for n < 2^20 / d: n < 1024 here
  if ((long)a.H * a.W * a.Cin >= (1l << 31) || (long)a.Ho * a.Wo * a.Cout >= (1l << 31)) return hipErrorInvalidValue;
  const int nj = nv_xblock_nj(a.Cin);
  if (a.stride == 1) {
    if (nj == 0) return launch_xblock_nt<1, 0>(a, n, groups, s);
    if (nj == 1) return launch_xblock_nt<1, 1>(a, n, groups, s);
    if (nj == 2) return launch_xblock_nt<1, 2>(a, n, groups, s);
    if (nj == 4) return launch_xblock_nt<1, 4>(a, n, groups, s);
    if (nj == 7) return launch_xblock_nt<1, 7>(a, n, groups, s);
  } else {
    if (nj == 0) return launch_xblock_nt<2, 0>(a, n, groups, s);
    if (nj == 1) return launch_xblock_nt<2, 1>(a, n, groups, s);
    if (nj == 2) return launch_xblock_nt<2, 2>(a, n, groups, s);
  }
  return hipErrorInvalidValue;
}

// ---- the trunk's last 1x1 + the NetVLAD pre-projection, register-resident input ------------------------------------------------------
// Same arithmetic as MODE 2 above, organised for the shape it really has (Cin = 112 -> 1280 hidden -> 128): a wave owns 32 pixels for
// BOTH GEMMs, so the input (2 m-tiles x Cin values per lane group = NJ float4 per m-tile and lane) stays in registers for the whole
// kernel, the expanded chunk only crosses LDS inside the wave (C layout -> A layout, no workgroup barrier), and the workgroup's LDS is
// just the two weight double buffers (35 KB: four workgroups per CU instead of one).  A float4 load hands a lane 4 CONSECUTIVE input
// channels of its pixel, so the K dimension is walked in the permuted order k = (lq + 4 j) * 4 + e (j = load, e = element); the host
// packs the expand fragments in the same order (pack_nv_expand_tail).
// NJ = Cin / 16: 7 / 8 (MobileNetV2 x 0.35 / 0.4: three workgroups per CU) and 15 (x 0.75, Cin = 240: 120 input registers per lane, two per CU).
template <int NJ, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NJ > 8 ? 2 : 3, NJ > 8 ? 2 : 3))) void nv_tail_kernel(NvBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CIN = NJ * 16, KSE1 = CIN / 4 + 1, WE_N = nvb_we_rec(CIN), WD_N = nvb_wd_rec(NT);
  constexpr int WER = WE_N / 256, WDR = WD_N / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lp = lane & 15;
  float* WE = lds;                    // [2][WE_N]
  float* WD = WE + 2 * WE_N;          // [2][WD_N]
  const int p0 = (int)blockIdx.x * 128 + wave * 32;
  const int nchunk = a.Chid >> 4;
  const int ch0 = blockIdx.z * a.cpg, ch1 = min(nchunk, ch0 + a.cpg);
  float wes[WER], wds[WDR];
  auto fetch_w = [&](int ch) {
    const float* wb = a.we + (size_t)ch * WE_N + tid;
#pragma unroll
    for (int i = 0; i < WER; ++i) wes[i] = wb[256 * i];
    const float* wc = a.wp + (size_t)ch * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wds[i] = wc[256 * i];
  };
  auto store_w = [&](int buf) {
    float* wl = WE + buf * WE_N + tid;
#pragma unroll
    for (int i = 0; i < WER; ++i) wl[256 * i] = wes[i];
    float* wm = WD + buf * WD_N + tid;
#pragma unroll
    for (int i = 0; i < WDR; ++i) wm[256 * i] = wds[i];
  };
  if (ch0 < ch1) fetch_w(ch0);
  // input: pixel p0 + mt*16 + lp, channels (lq + 4 j) * 4 .. + 3, summed over the producer's partial slabs
  f32x4 xr[2][NJ];
  float one[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = p0 + mt * 16 + lp;
    const bool ok = p < a.P;
    one[mt] = ok ? 1.f : 0.f;
    const float* src = a.in + (ok ? p * CIN : 0) + lq * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + j * 16);        // one slab (launch_nv_tail rejects a split input)
      xr[mt][j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (ch0 < ch1) store_w(0);
  __syncthreads();
  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float lo_e = nvf_lo(a.act_e), hi_e = nvf_hi(a.act_e);
  for (int ch = ch0; ch < ch1; ++ch) {
    const int b = (ch - ch0) & 1;
    if (ch + 1 < ch1) fetch_w(ch + 1);
    const float* wl = WE + b * WE_N + lane;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float wv = wl[(j * 4 + e) * 64];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xr[0][j][e], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xr[1][j][e], c1, 0, 0, 0);
      }
    {
      const float wv = wl[(KSE1 - 1) * 64];          // bias step: bias[hidden] (k = 0 row) x mask[pixel]
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, one[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, one[1], c1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { c0[r] = nvf_clamp(c0[r], lo_e, hi_e); c1[r] = nvf_clamp(c1[r], lo_e, hi_e); }
    // The expand GEMM is evaluated TRANSPOSED (A = weights, B = pixels): its C layout -- lane (pixel lp, group lq) holds hidden channels 4 lq + r --
    // is already the A layout of the project GEMM's k-step r (pack_nv_proj_t orders the project fragments accordingly): no trip through LDS
    const float* wpl = WD + b * WD_N + 256 + lane;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float d0 = c0[ks], d1 = c1[ks];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float wv = wpl[(ks * NT + t) * 64];
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d0, wv, acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, wv, acc[1][t], 0, 0, 0);
      }
    }
    if (ch + 1 < ch1) store_w(b ^ 1);
    __syncthreads();          // the next chunk's weights are in place; everybody is done with this chunk's
  }
  const bool lead = blockIdx.z == 0;
  const int Cout = a.Cout;
  float* op = a.out + (size_t)blockIdx.z * a.out_slab_stride;
  const float lo_p = nvf_lo(a.act_p), hi_p = nvf_hi(a.act_p);
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) {
    const int p = p0 + m2 * 16 + lq * 4;
    const int nvalid = (int)min((long)4, a.P - p);
    const int base = p * Cout + lp;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t * 16 + lp >= Cout) continue;
      const float bv = lead ? a.bp[t * 16 + lp] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r >= nvalid) continue;
        op[base + r * Cout + t * 16] = nvf_clamp(acc[m2][t][r] + bv, lo_p, hi_p);
      }
    }
  }
}
// expand record for nv_tail_kernel: k-step (j, e) holds input channel (lq + 4 j) * 4 + e for lane group lq; bias step last
void pack_nv_expand_tail(const float* w /*[chid][cin]*/, const float* b, int chid, int cin, float* dst) {
  const int rec = nvb_we_rec(cin), nj = cin / 16;
  for (int ch = 0; ch < chid / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < rec; ++i) d[i] = 0.f;
    for (int j = 0; j < nj; ++j)
      for (int e = 0; e < 4; ++e)
        for (int l = 0; l < 64; ++l) d[(j * 4 + e) * 64 + l] = w[(size_t)(ch * 16 + (l & 15)) * cin + ((l >> 4) + 4 * j) * 4 + e];
    for (int c = 0; c < 16; ++c) d[(cin / 4) * 64 + c] = b[ch * 16 + c];
  }
}
bool nv_tail_supported(int cin, int cout) { return (cin == 112 || cin == 128 || cin == 240) && nv_block_ntiles(cout) == 8; }
hipError_t launch_nv_tail(const NvBlockArgs& a, int groups, hipStream_t s) {
  if (a.P * (long)std::max(a.Cin, a.Cout) >= (1l << 31) || a.in_slabs > 1) return hipErrorInvalidValue;
  const size_t lds = sizeof(float) * (2 * (size_t)nvb_we_rec(a.Cin) + 2 * (size_t)nvb_wd_rec(8));
  auto k = a.Cin == 112 ? nv_tail_kernel<7, 8> : a.Cin == 128 ? nv_tail_kernel<8, 8> : nv_tail_kernel<15, 8>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((unsigned)((a.P + 127) / 128), 1, groups), dim3(256), lds, s, a);
  return hipGetLastError();
}

int nv_block_ntiles(int cout) { const int nt = (cout + 15) / 16; return nt <= 2 ? nt : nt <= 4 ? 4 : nt <= 8 ? 8 : -1; }

// mode: 0 spatial block, 1 spatial block behind the first conv (front), 2 two chained 1x1 convolutions (flat)
bool nv_block_supported(int cin, int chid, int cout, int stride, bool expand, int mode) {
  if (stride != 1 && stride != 2) return false;
  if ((cin & 3) || (chid & 15) || cin < 4 || cout < 1 || nv_block_ntiles(cout) < 0) return false;
  if (!expand && chid != cin) return false;
  if (mode == 1 && (expand || stride != 1 || (cin != 16 && cin != 32))) return false;
  if (mode == 2 && (!expand || stride != 1 || (cin & 7))) return false;
  if (expand && nvb_we_rec(cin) > 8 * 256) return false;       // the expand-weight staging covers 8 x 256 floats per chunk (Cin <= 124)
  return nvb_lds_bytes(cin, stride, mode, expand, 1, nv_block_ntiles(cout)) <= 160 * 1024;
}

// groups: the hidden channels are split over `groups` workgroup groups (grid.z); group g writes slab g of a.out (a.cpg chunks of 16 each)
hipError_t launch_nv_block(const NvBlockArgs& a, bool expand, int mode, int n, int groups, hipStream_t s) {
  // 32-bit element offsets inside the kernels
  if ((long)a.H * a.W * a.Cin >= (1l << 31) || (long)a.Ho * a.Wo * a.Cout >= (1l << 31) || a.P * (long)std::max(a.Cin, a.Cout) >= (1l << 31))
    return hipErrorInvalidValue;
  if (mode == 1) {
    if ((nvb_ih(1) - 1) * a.c0_stride + 3 > 24 || (nvb_iw(1) - 1) * a.c0_stride + 3 > NVB_U8_PITCH) return hipErrorInvalidValue;
    return launch_block_nt<false, 1, 1>(a, n, groups, s);
  }
  if (mode == 2) return launch_block_nt<true, 2, 1>(a, n, groups, s);
  if (expand) return a.stride == 1 ? launch_block_nt<true, 0, 1>(a, n, groups, s) : launch_block_nt<true, 0, 2>(a, n, groups, s);
  return a.stride == 1 ? launch_block_nt<false, 0, 1>(a, n, groups, s) : launch_block_nt<false, 0, 2>(a, n, groups, s);
}

// ---- partial slabs -> one tensor ------------------------------------------------------------------------------------------------------
// A block that split its hidden channels over g groups leaves g partial slabs; every consumer workgroup group would re-read all of them
// (measured: the tail kernel's prologue alone moved 215 MB of L2 traffic for 21 MB of input at g = 5 producers x 10 consumer groups).
// For g >= 3 the slabs are summed once, in place into slab 0, in slab order (deterministic), by this kernel.
// `tree` > 1: runs of `tree` consecutive slabs are summed first, then the runs in order -- the order in which a launch of merged groups (NvBlockArgs::gmerge = tree) adds
// the same partial sums, so that a batch (merged) and one image (not merged) give the same bits.
__global__ __launch_bounds__(256) void nv_slab_sum_kernel(float* __restrict__ t, int slabs, long slab_stride, long n4, int tree) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4* p = reinterpret_cast<f32x4*>(t) + i;
  f32x4 v = *p;
  if (tree <= 1) {
    for (int sl = 1; sl < slabs; ++sl) v += *reinterpret_cast<const f32x4*>(t + (size_t)sl * slab_stride + i * 4);
  } else {
    for (int s0 = 0; s0 < slabs; s0 += tree) {
      const int s1 = s0 + tree < slabs ? s0 + tree : slabs;
      f32x4 r = *reinterpret_cast<const f32x4*>(t + (size_t)s0 * slab_stride + i * 4);
      for (int sl = s0 + 1; sl < s1; ++sl) r += *reinterpret_cast<const f32x4*>(t + (size_t)sl * slab_stride + i * 4);
      v = s0 == 0 ? r : v + r;
    }
  }
  *p = v;
}
hipError_t launch_nv_slab_sum(float* t, int slabs, long slab_stride, long count, hipStream_t s, int tree) {
  if (slabs < 2) return hipSuccess;
  if ((count & 3) || (slab_stride & 3)) return hipErrorInvalidValue;
  const long n4 = count >> 2;
  hipLaunchKernelGGL(nv_slab_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, t, slabs, slab_stride, n4, tree);
  return hipGetLastError();
}

// ---- host-side packing: one record per chunk of 16 hidden channels, B fragments in v_mfma_f32_16x16x4_f32 order B[k = lane >> 4][n = lane & 15] ----
// expand record (nvb_we_rec(cin) floats): [ks = cin/4][lane] = W[chunk*16 + (lane & 15)][ks*4 + (lane >> 4)], then the bias k-step
// (k = 0 row: b[chunk*16 + (lane & 15)], rows 1..3: 0), zero padding
void pack_nv_expand(const float* w /*[chid][cin]*/, const float* b /*[chid]*/, int chid, int cin, float* dst) {
  const int rec = nvb_we_rec(cin);
  for (int ch = 0; ch < chid / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < rec; ++i) d[i] = 0.f;
    for (int ks = 0; ks < cin / 4; ++ks)
      for (int l = 0; l < 64; ++l) d[ks * 64 + l] = w[(size_t)(ch * 16 + (l & 15)) * cin + ks * 4 + (l >> 4)];
    for (int c = 0; c < 16; ++c) d[(cin / 4) * 64 + c] = b[ch * 16 + c];
  }
}
size_t pack_nv_expand_floats(int chid, int cin) { return (size_t)(chid / 16) * nvb_we_rec(cin); }
// depthwise + project record (nvb_wd_rec(nt) floats): wd[tap][c] (c = channel within the chunk), bd[16], padding to 256,
// then [ks = 4][t = nt][lane] = Wp[t*16 + (lane & 15)][chunk*16 + ks*4 + (lane >> 4)] (0 beyond cout).  wd/bd may be null (mode 2).
void pack_nv_dwproj(const float* wd /*[chid][9]*/, const float* bd, const float* wp /*[cout][chid]*/, int cout, int chid, int nt, float* dst) {
  const int rec = nvb_wd_rec(nt);
  for (int ch = 0; ch < chid / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < 256; ++i) d[i] = 0.f;
    for (int t = 0; t < 9; ++t)
      for (int c = 0; c < 16; ++c) d[t * 16 + c] = wd ? wd[(size_t)(ch * 16 + c) * 9 + t] : 0.f;
    for (int c = 0; c < 16; ++c) d[144 + c] = bd ? bd[ch * 16 + c] : 0.f;
    for (int ks = 0; ks < 4; ++ks)
      for (int t = 0; t < nt; ++t)
        for (int l = 0; l < 64; ++l) {
          const int co = t * 16 + (l & 15), k = ch * 16 + ks * 4 + (l >> 4);
          d[256 + (ks * nt + t) * 64 + l] = co < cout ? wp[(size_t)co * chid + k] : 0.f;
        }
  }
}
size_t pack_nv_dwproj_floats(int chid, int nt) { return (size_t)(chid / 16) * nvb_wd_rec(nt); }
// the record of nv_tail_kernel (no depthwise part): project fragments [ks][t][lane] = Wp[t*16 + (lane & 15)][chunk*16 + (lane >> 4) * 4 + ks] -- lane group lq
// owns hidden channels 4 lq .. 4 lq + 3 of the chunk (the C layout of the transposed expand GEMM), k-step ks takes the ks-th of them
void pack_nv_proj_t(const float* wp /*[cout][chid]*/, int cout, int chid, int nt, float* dst) {
  const int rec = nvb_wd_rec(nt);
  for (int ch = 0; ch < chid / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < 256; ++i) d[i] = 0.f;
    for (int ks = 0; ks < 4; ++ks)
      for (int t = 0; t < nt; ++t)
        for (int l = 0; l < 64; ++l) {
          const int co = t * 16 + (l & 15), k = ch * 16 + (l >> 4) * 4 + ks;
          d[256 + (ks * nt + t) * 64 + l] = co < cout ? wp[(size_t)co * chid + k] : 0.f;
        }
  }
}
// the same record for nv_xblock_kernel: the depthwise part as [lane group lq][kp][tap][2] + bias[2] (20 floats per (lq, kp): channels kp*8 + lq and + 4,
// the two halves of the kernel's v_pk_fma_f32), read as five ds_read_b128; project part unchanged
void pack_nv_dwproj_x(const float* wd /*[chid][9]*/, const float* bd, const float* wp /*[cout][chid]*/, int cout, int chid, int nt, float* dst) {
  pack_nv_dwproj(wd, bd, wp, cout, chid, nt, dst);
  const int rec = nvb_wd_rec(nt);
  for (int ch = 0; ch < chid / 16; ++ch) {
    float* d = dst + (size_t)ch * rec;
    for (int i = 0; i < 256; ++i) d[i] = 0.f;
    for (int lq = 0; lq < 4; ++lq)
      for (int kp = 0; kp < 2; ++kp)
        for (int h = 0; h < 2; ++h) {
          const int c = ch * 16 + kp * 8 + lq + 4 * h;
          for (int t = 0; t < 9; ++t) d[(lq * 2 + kp) * 20 + t * 2 + h] = wd[(size_t)c * 9 + t];
          d[(lq * 2 + kp) * 20 + 18 + h] = bd[c];
        }
  }
}
// first conv (mode 1) as B fragments [ks = 3][t = 2][lane]: k = ks*4 + (lane >> 4): k < 9 tap weight w[c][k], k == 9 the bias, else 0
void pack_nv_conv0(const float* w /*[cout][9]*/, const float* b, int cout, float* dst /*[384]*/) {
  for (int ks = 0; ks < 3; ++ks)
    for (int t = 0; t < 2; ++t)
      for (int l = 0; l < 64; ++l) {
        const int k = ks * 4 + (l >> 4), c = t * 16 + (l & 15);
        dst[(ks * 2 + t) * 64 + l] = c < cout ? (k < 9 ? w[c * 9 + k] : k == 9 ? b[c] : 0.f) : 0.f;
      }
}

}  // namespace d2fe
