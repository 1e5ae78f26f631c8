// netvlad.hip -- global image descriptor (MobileNetVLAD) on gfx950.
//
// Replaces MobileNetVLADONNX::inference (d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:49-74: ONNX Runtime
// session run + optional PCA).  The reference's graph is NOT in its tree; the layer kinds below
// execute whatever flat layer list d2fe_load_netvlad() is given -- the documented stand-in lives in
// d2slam_amd/netvlad.py and oracle/d2fe_oracle.c (A9 block).  All of it is 0.33 GMAC per 640x480 image (0.6 % of
// SuperPoint), HBM/latency bound: 1x1 convs as fp32-MFMA GEMMs, depthwise/first conv on the VALU, activations NHWC.
#include "kernels.h"

namespace d2fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float nv_act(float v, int act) {
  if (act >= 1) v = v > 0.f ? v : 0.f;
  if (act == 2) v = v < 6.f ? v : 6.f;
  return v;
}
__device__ __forceinline__ int same_pad(int in, int k, int stride, int out) {
  int pt = (out - 1) * stride + k - in;
  return pt > 0 ? pt / 2 : 0;
}

// ---- first layer: u8 gray -> (x-128)/128 -> 3x3 conv, stride s, TF-SAME, COUT <= 32 ------------------------------------
__global__ __launch_bounds__(256) void nv_conv0_kernel(const uint8_t* __restrict__ img, int stride_b, long img_stride, int H,
                                                       int W, int Ho, int Wo, int cstride, int cout, int act,
                                                       const float* __restrict__ w /*[9][32]*/, const float* __restrict__ b,
                                                       float* __restrict__ out) {
  const int n = blockIdx.z;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Ho * Wo) return;
  const int y = p / Wo, x = p % Wo;
  const int pt = same_pad(H, 3, cstride, Ho), pl = same_pad(W, 3, cstride, Wo);
  const uint8_t* ip = img + (size_t)n * img_stride;
  float v[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y * cstride + ky - pt, xx = x * cstride + kx - pl;
      v[ky * 3 + kx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? ((float)ip[(size_t)yy * stride_b + xx] - 128.0f) / 128.0f : 0.f;
    }
  float* op = out + ((size_t)n * Ho * Wo + p) * cout;
  for (int c0 = 0; c0 < cout; c0 += 8) {
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = b[c0 + c];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_fmaf(v[t], w[t * 32 + c0 + c], acc[c]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c0 + c < cout) op[c0 + c] = nv_act(acc[c], act);
  }
}

// ---- depthwise 3x3, NHWC, 4 channels per thread ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nv_dw_kernel(const float* __restrict__ in, int H, int W, int C, int Ho, int Wo,
                                                    int cstride, int act, const float* __restrict__ w /*[9][C]*/,
                                                    const float* __restrict__ b, float* __restrict__ out) {
  const int n = blockIdx.z;
  const int C4 = C / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)Ho * Wo * C4) return;
  const int c4 = (int)(i % C4);
  const int p = (int)(i / C4);
  const int y = p / Wo, x = p % Wo;
  const int pt = same_pad(H, 3, cstride, Ho), pl = same_pad(W, 3, cstride, Wo);
  const float* ip = in + (size_t)n * H * W * C + c4 * 4;
  f32x4 acc = *reinterpret_cast<const f32x4*>(b + c4 * 4);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y * cstride + ky - pt, xx = x * cstride + kx - pl;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ip + ((size_t)yy * W + xx) * C);
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + (ky * 3 + kx) * C + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v[j], ww[j], acc[j]);
      }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = nv_act(acc[j], act);
  *reinterpret_cast<f32x4*>(out + ((size_t)n * Ho * Wo + p) * C + c4 * 4) = acc;
}

// ---- pointwise (1x1) conv as a GEMM on fp32 MFMA: [P][Cin] x [Cin][CoutPad] ---------------------------------------------------
// Block = 128 consecutive pixels (one contiguous NHWC span -> coalesced staging) x up to 128 output channels; wave w owns
// pixels [32w, 32w+32) and all NT 32-wide channel tiles (v_mfma_f32_32x32x2_f32, NT*16 accumulators).  Cin is staged through
// LDS in chunks of 64 channels (row stride 65 floats: conflict-free column reads).  Weights are pre-packed on the host in the
// same fragment order as the SuperPoint convs: [ntile][cin/8][lane] float4, element q = W[co = ntile*32 + (lane&31)][ci = c8*8 + 2q + (lane>>5)].
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PW_CK = 64;
template <int NT>
__global__ __launch_bounds__(256) void nv_pw_mfma_kernel(const float* __restrict__ in, long P, int Cin, int Cout, int act,
                                                         const f32x4* __restrict__ wpack, const float* __restrict__ b,
                                                         const float* __restrict__ res, float* __restrict__ out) {
  __shared__ float xs[128 * (PW_CK + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long p0 = (long)blockIdx.x * 128;
  const int nt0 = blockIdx.y * NT;
  const int c8n = (Cin + 7) / 8;                      // Cin is a multiple of 8 for every MobileNetV2 width used here
  const int npix = (int)((P - p0) < 128 ? (P - p0) : 128);
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float bv = b[(nt0 + n) * 32 + (lane & 31)];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = bv;
  }
  // Both operand streams run one step ahead of the MFMAs: the next chunk of the input span is fetched into registers while this chunk is multiplied (a
  // launch of a dozen workgroups -- one image at 15 x 20 -- has nothing else to hide that latency behind), and the weight fragments of step c8 + 1 are
  // in flight during step c8.  Summation order per output is unchanged: chunk by chunk, c8 by c8.
  f32x4 pre[8];
  auto fetch = [&](int c0) {
    const int cc4 = ((Cin - c0) < PW_CK ? (Cin - c0) : PW_CK) / 4;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + k * 256;
      const int pp = cc4 == 16 ? (i >> 4) : i / cc4, q = cc4 == 16 ? (i & 15) : i % cc4;
      pre[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 128 * cc4 && pp < npix) pre[k] = *reinterpret_cast<const f32x4*>(in + (size_t)(p0 + pp) * Cin + c0 + q * 4);
    }
  };
  auto stage = [&](int cc4) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + k * 256;
      if (i < 128 * cc4) {
        const int pp = cc4 == 16 ? (i >> 4) : i / cc4, q = cc4 == 16 ? (i & 15) : i % cc4;
        float* d = xs + pp * (PW_CK + 1) + q * 4;
        d[0] = pre[k][0]; d[1] = pre[k][1]; d[2] = pre[k][2]; d[3] = pre[k][3];
      }
    }
    __syncthreads();
  };
  const float* ap = xs + (wave * 32 + (lane & 31)) * (PW_CK + 1) + (lane >> 5);
  fetch(0);
  if constexpr (NT == 1) {
    // one channel tile per workgroup is what the small launches get (launch_nv_pw): the weight fragments of the WHOLE next chunk are in flight as well
    f32x4 wc[8], wn[8];
    auto wfetch = [&](int c0, f32x4 (&w)[8]) {
      const int n8 = ((Cin - c0) < PW_CK ? (Cin - c0) : PW_CK) / 8;
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8)
        if (c8 < n8) w[c8] = wpack[((size_t)nt0 * c8n + (c0 / 8 + c8)) * 64 + lane];
    };
    wfetch(0, wc);
    for (int c0 = 0; c0 < Cin; c0 += PW_CK) {
      const int cc = (Cin - c0) < PW_CK ? (Cin - c0) : PW_CK;
      stage(cc / 4);
      if (c0 + PW_CK < Cin) { fetch(c0 + PW_CK); wfetch(c0 + PW_CK, wn); }
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8)
        if (c8 < cc / 8) {
          float av[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) av[q] = ap[c8 * 8 + 2 * q];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], wc[c8][q], acc[0], 0, 0, 0);
        }
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) wc[c8] = wn[c8];
    }
  } else {
    for (int c0 = 0; c0 < Cin; c0 += PW_CK) {
      const int cc = (Cin - c0) < PW_CK ? (Cin - c0) : PW_CK;
      const int n8 = cc / 8;
      f32x4 bw[NT], bwn[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) bw[n] = wpack[((size_t)(nt0 + n) * c8n + c0 / 8) * 64 + lane];
      stage(cc / 4);
      if (c0 + PW_CK < Cin) fetch(c0 + PW_CK);
      for (int c8 = 0; c8 < n8; ++c8) {
        if (c8 + 1 < n8) {
#pragma unroll
          for (int n = 0; n < NT; ++n) bwn[n] = wpack[((size_t)(nt0 + n) * c8n + (c0 / 8 + c8 + 1)) * 64 + lane];
        }
        float av[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = ap[c8 * 8 + 2 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bw[n][q], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) bw[n] = bwn[n];
      }
    }
  }
  // C layout: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel of this wave's 32)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pp = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (pp < npix) {
      const long p = p0 + pp;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = (nt0 + n) * 32 + (lane & 31);
        if (co < Cout) {
          float v = acc[n][r];
          if (res) v += res[(size_t)p * Cout + co];
          out[(size_t)p * Cout + co] = nv_act(v, act);
        }
      }
    }
  }
}

// ---- NetVLAD head: soft-assignment, residual aggregation, intra + global L2 ----------------------------------------------------
// x: [np][D] pre-projected features per image.  Stage 1 (grid: position chunks of 64 x images): features and assignment
// weights live in LDS; memberships a = softmax_k(x W_a^T + b_a); partial V[k][d] = sum_p a[p][k] (c[k][d] - x[p][d]) written
// per chunk (deterministic: no float atomics).  Stage 2 (one block per image): chunk sum, intra-normalisation per cluster,
// flatten k-major, global L2.  K <= 64, D <= 128*2, K*D <= 8192.
constexpr int VL_PCH = 64;
__global__ __launch_bounds__(256) void nv_vlad_partial_kernel(const float* __restrict__ x, int slabs, long slab_stride, int np, int D, int K,
                                                              const float* __restrict__ aw, const float* __restrict__ ab,
                                                              const float* __restrict__ cen, float* __restrict__ part,
                                                              int nchunk) {
  extern __shared__ float sm[];
  const int DS = D + 1;
  float* xs = sm;                    // [VL_PCH][D+1]
  float* ws = xs + VL_PCH * DS;      // [K][D+1]
  float* a = ws + K * DS;            // [VL_PCH][K+1]
  const int img = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int p0 = chunk * VL_PCH;
  const int pn = (np - p0) < VL_PCH ? (np - p0) : VL_PCH;
  const float* xi = x + ((size_t)img * np + p0) * D;
  // x = sum of `slabs` partial tensors (the fused tail splits its hidden channels over workgroup groups); D is a multiple of 4
  {
    const int D4 = D >> 2;
    for (int i = tid; i < VL_PCH * D4; i += 256) {
      const int p = i / D4, j4 = i - p * D4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < pn) {
        const float* s0 = xi + (size_t)p * D + j4 * 4;
        v = *reinterpret_cast<const f32x4*>(s0);
        for (int sl = 1; sl < slabs; ++sl) v += *reinterpret_cast<const f32x4*>(s0 + (size_t)sl * slab_stride);
      }
      float* d = xs + p * DS + j4 * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
  }
  for (int i = tid; i < K * D; i += 256) ws[(i / D) * DS + i % D] = aw[i];
  __syncthreads();
  for (int i = tid; i < pn * K; i += 256) {
    const int p = i % pn, k = i / pn;          // consecutive lanes -> consecutive positions (distinct LDS rows)
    float s = ab[k];
    for (int j = 0; j < D; ++j) s = __builtin_fmaf(xs[p * DS + j], ws[k * DS + j], s);
    a[p * (K + 1) + k] = s;
  }
  __syncthreads();
  if (tid < pn) {
    float* ap = a + tid * (K + 1);
    float m = -__builtin_inff();
    for (int k = 0; k < K; ++k) m = ap[k] > m ? ap[k] : m;
    float sum = 0.f;
    for (int k = 0; k < K; ++k) { const float e = __expf(ap[k] - m); ap[k] = e; sum += e; }
    for (int k = 0; k < K; ++k) ap[k] = ap[k] / sum;
  }
  __syncthreads();
  const int KD = K * D;
  float* po = part + ((size_t)img * nchunk + chunk) * KD;
  for (int e = tid; e < KD; e += 256) {
    const int k = e / D, j = e % D;
    const float c = cen[e];
    float acc = 0.f;
    for (int p = 0; p < pn; ++p) acc = __builtin_fmaf(a[p * (K + 1) + k], c - xs[p * DS + j], acc);
    po[e] = acc;
  }
}

// Every sum below has a fixed order (no float atomics): the descriptor is reproducible run to run for any K x D, as it is for the 32 x 128 head.
__global__ __launch_bounds__(1024) void nv_vlad_final_kernel(const float* __restrict__ part, int nchunk, int D, int K,
                                                             float* __restrict__ out) {
  __shared__ float sq[8192];         // v^2 per element, then reused for nothing else (K * D <= 8192)
  __shared__ float nk[64];           // squared norm per cluster
  __shared__ float red2[16];
  const int img = blockIdx.x, tid = threadIdx.x;
  const int KD = K * D;
  float v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = tid + 1024 * r;
    float s = 0.f;
    if (e < KD) {
      for (int c = 0; c < nchunk; ++c) s += part[((size_t)img * nchunk + c) * KD + e];
      sq[e] = s * s;
    }
    v[r] = s;
  }
  __syncthreads();
  // one 16-lane group per cluster: lane j sums channels j, j + 16, .. in order, then a fixed butterfly over the 16 lanes
  for (int k = tid >> 4; k < K; k += 64) {
    float s = 0.f;
    for (int j = tid & 15; j < D; j += 16) s += sq[k * D + j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
    if ((tid & 15) == 0) nk[k] = s;
  }
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = tid + 1024 * r;
    if (e < KD) {
      const float n = __builtin_sqrtf(nk[e / D]);
      v[r] = v[r] / (n > 1e-12f ? n : 1e-12f);
      tot += v[r] * v[r];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  if ((tid & 63) == 0) red2[tid >> 6] = tot;
  __syncthreads();
  float gsum = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) gsum += red2[w];
  const float nt = __builtin_sqrtf(gsum);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = tid + 1024 * r;
    if (e < KD) out[(size_t)img * KD + e] = v[r] / (nt > 1e-12f ? nt : 1e-12f);
  }
}

// ---- the same head on the matrix pipe (K = 32 clusters, D = 128: the shape the 4096-D descriptor implies) ------------------------------
// One workgroup per (64-position chunk, image).  Both contractions are small GEMMs on v_mfma_f32_16x16x4_f32:
//   logits  L[64 px][32] = X[64][128] Wa^T + ba                    M = px (one m-tile per wave), N = 32, K = 128
//   sums    P[32][128 + 1] = A^T[32][64] [X | 1]                    M = clusters, N = 128 channels + a column of ones (-> S[k] = sum_p a[p][k])
// so that V[k][d] = sum_p a[p][k] (c[k][d] - x[p][d]) = c[k][d] S[k] - P[k][d] is finished by the final kernel over the chunks.
// X sits pixel-major in LDS with an odd pitch (145): the A fragments of the first GEMM (lanes = pixels) and the B fragments of the
// second (lanes = channels) both read it conflict-free.  Partial record per chunk: [32][144] (128 channels, S, zero padding).
constexpr int VM_DP = 145, VM_PP = 144;
__global__ __launch_bounds__(256) void nv_vlad_mfma_kernel(const float* __restrict__ x, int slabs, long slab_stride, int np,
                                                           const float* __restrict__ wa_pack, const float* __restrict__ ab,
                                                           float* __restrict__ part, int nchunk) {
  __shared__ float Xr[64 * VM_DP];
  __shared__ float As[64 * 33];
  const int img = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lp = lane & 15;
  const int p0 = chunk * 64;
  const int pn = (np - p0) < 64 ? (np - p0) : 64;
  const float* xi = x + ((size_t)img * np + p0) * 128;
  {
    // x = the sum of the tail kernel's partial slabs, in slab order.  Thread (p8, j4) owns the float4 j4 of pixels p8, p8 + 8, ..: the eight loads of a slab
    // (of two slabs) are issued as one batch -- a run-time loop over the slabs around EACH load would cost one memory round trip per slab and element
    const int p8 = tid >> 5, j4 = tid & 31;
    f32x4 v[8];
    unsigned off[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int p = p8 + 8 * it;
      off[it] = (unsigned)((p < pn ? p : 0) * 128 + j4 * 4);
      v[it] = *reinterpret_cast<const f32x4*>(xi + off[it]);
    }
    int sl = 1;
    for (; sl + 1 < slabs; sl += 2) {
      const float* xa = xi + (size_t)sl * slab_stride;
      const float* xb = xa + slab_stride;
      f32x4 a[8], b[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) a[it] = *reinterpret_cast<const f32x4*>(xa + off[it]);
#pragma unroll
      for (int it = 0; it < 8; ++it) b[it] = *reinterpret_cast<const f32x4*>(xb + off[it]);
#pragma unroll
      for (int it = 0; it < 8; ++it) { v[it] += a[it]; v[it] += b[it]; }
    }
    if (sl < slabs) {
      const float* xa = xi + (size_t)sl * slab_stride;
      f32x4 a[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) a[it] = *reinterpret_cast<const f32x4*>(xa + off[it]);
#pragma unroll
      for (int it = 0; it < 8; ++it) v[it] += a[it];
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int p = p8 + 8 * it;
      float* d = Xr + p * VM_DP + j4 * 4;
      const bool ok = p < pn;
      d[0] = ok ? v[it][0] : 0.f; d[1] = ok ? v[it][1] : 0.f; d[2] = ok ? v[it][2] : 0.f; d[3] = ok ? v[it][3] : 0.f;
    }
  }
  for (int i = tid; i < 64 * 17; i += 256) { const int p = i / 17, c = i - p * 17; Xr[p * VM_DP + 128 + c] = (c == 0 && p < pn) ? 1.f : 0.f; }
  __syncthreads();
  // ---- logits + softmax: wave w owns pixels 16 w .. 16 w + 15 ----
  {
    f32x4 c0, c1;
    const float b0 = ab[lp], b1 = ab[16 + lp];
    c0 = f32x4{b0, b0, b0, b0}; c1 = f32x4{b1, b1, b1, b1};
    const float* xa = Xr + (wave * 16 + lp) * VM_DP + lq;
    const float* wb = wa_pack + lane;
#pragma unroll 4
    for (int ks = 0; ks < 32; ++ks) {
      const float a = xa[ks * 4];
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[(ks * 2) * 64], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[(ks * 2 + 1) * 64], c1, 0, 0, 0);
    }
    // row r of the C tiles = pixel 16 w + 4 lq + r; its 32 logits sit in lanes lp = 0..15 of this lq group (c0: clusters 0-15, c1: 16-31)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = fmaxf(c0[r], c1[r]);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
      const float e0 = __expf(c0[r] - m), e1 = __expf(c1[r] - m);
      float sum = e0 + e1;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 16);
      const int p = wave * 16 + lq * 4 + r;
      const bool ok = p < pn;
      As[p * 33 + lp] = ok ? e0 / sum : 0.f;
      As[p * 33 + 16 + lp] = ok ? e1 / sum : 0.f;
    }
  }
  __syncthreads();
  // ---- P = A^T [X | 1]: 2 m-tiles (clusters) x 9 n-tiles (128 channels + the ones column), K = 64 pixels ----
  float* po = part + ((size_t)img * nchunk + chunk) * 32 * VM_PP;
  for (int t = wave; t < 18; t += 4) {
    const int mt = t / 9, nt = t - mt * 9;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float* aa = As + lq * 33 + mt * 16 + lp;
    const float* bb = Xr + lq * VM_DP + nt * 16 + lp;
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ks * 4 * 33], bb[ks * 4 * VM_DP], c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) po[(mt * 16 + lq * 4 + r) * VM_PP + nt * 16 + lp] = c[r];
  }
}

// V[k][d] = c[k][d] * sum_chunks S[k] - sum_chunks P[k][d], then intra-normalisation per cluster, flatten k-major, global L2
__global__ __launch_bounds__(1024) void nv_vlad_final_mfma_kernel(const float* __restrict__ part, int nchunk, const float* __restrict__ cen,
                                                                  float* __restrict__ out) {
  __shared__ float red[34];
  __shared__ float red2[16];
  __shared__ float S[32];
  __shared__ float half[64];         // per 64 consecutive elements (one wave, half a cluster): sum of squares
  const int img = blockIdx.x, tid = threadIdx.x;
  const float* pp = part + (size_t)img * nchunk * 32 * VM_PP;
  if (tid < 32) {
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += pp[((size_t)c * 32 + tid) * VM_PP + 128];
    S[tid] = s;
  }
  __syncthreads();
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = tid + 1024 * r, k = e >> 7, d = e & 127;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += pp[((size_t)c * 32 + k) * VM_PP + d];
    v[r] = cen[e] * S[k] - s;
    float q = v[r] * v[r];           // the 128 channels of cluster k are 2 waves' worth of lanes: reduce per wave, the two halves added in wave order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((tid & 63) == 0) half[e >> 6] = q;
  }
  __syncthreads();
  if (tid < 32) red[tid] = half[2 * tid] + half[2 * tid + 1];
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = (tid + 1024 * r) >> 7;
    const float n = __builtin_sqrtf(red[k]);
    v[r] = v[r] / (n > 1e-12f ? n : 1e-12f);
    tot += v[r] * v[r];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  // 16 wave partials summed in wave order by every thread: no float atomic, the descriptor is reproducible run to run
  if ((tid & 63) == 0) red2[tid >> 6] = tot;
  __syncthreads();
  float gsum = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) gsum += red2[w];
  const float nt = __builtin_sqrtf(gsum);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(size_t)img * 4096 + tid + 1024 * r] = v[r] / (nt > 1e-12f ? nt : 1e-12f);
}

// ---- PCA: y = comp (x - mean); y /= |y|  (mobilenetvlad_onnx.h:66-71).  One wave per output row, then a normalise pass.
__global__ __launch_bounds__(256) void nv_pca_kernel(const float* __restrict__ x, int n, const float* __restrict__ comp,
                                                     const float* __restrict__ mean, int m, float* __restrict__ y) {
  const int img = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  const float* xr = x + (size_t)img * n;
  const float* cr = comp + (size_t)row * n;
  float s = 0.f;
  for (int j = lane * 4; j < n; j += 256) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(cr + j);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + j);
    const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + j);
#pragma unroll
    for (int q = 0; q < 4; ++q) s = __builtin_fmaf(c[q], xv[q] - mv[q], s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) y[(size_t)img * m + row] = s;
}
__global__ __launch_bounds__(1024) void nv_l2norm_kernel(float* __restrict__ y, int m) {
  __shared__ float red[16];
  float* yr = y + (size_t)blockIdx.x * m;
  float s = 0.f;
  for (int i = threadIdx.x; i < m; i += 1024) s += yr[i] * yr[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < 16; ++i) t += red[i];
  const float nrm = __builtin_sqrtf(t);
  for (int i = threadIdx.x; i < m; i += 1024) yr[i] = yr[i] / nrm;
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
hipError_t launch_nv_conv0(const uint8_t* img, int stride_b, long img_stride, int H, int W, int Ho, int Wo, int cstride,
                           int cout, int act, const float* w, const float* b, float* out, int n, hipStream_t s) {
  dim3 grid((Ho * Wo + 255) / 256, 1, n);
  hipLaunchKernelGGL(nv_conv0_kernel, grid, dim3(256), 0, s, img, stride_b, img_stride, H, W, Ho, Wo, cstride, cout, act, w, b, out);
  return hipGetLastError();
}
hipError_t launch_nv_dw(const float* in, int H, int W, int C, int Ho, int Wo, int cstride, int act, const float* w, const float* b,
                        float* out, int n, hipStream_t s) {
  const long tot = (long)Ho * Wo * (C / 4);
  dim3 grid((unsigned)((tot + 255) / 256), 1, n);
  hipLaunchKernelGGL(nv_dw_kernel, grid, dim3(256), 0, s, in, H, W, C, Ho, Wo, cstride, act, w, b, out);
  return hipGetLastError();
}
hipError_t launch_nv_pw(const float* in, long P, int Cin, int Cout, int CoutPad, int act, const float* w, const float* b,
                        const float* res, float* out, hipStream_t s) {
  const int ntiles = CoutPad / 32;
  const f32x4* wp = reinterpret_cast<const f32x4*>(w);
  const unsigned gx = (unsigned)((P + 127) / 128);
  // channel tiles per workgroup: as many as divide the layer (each staged input chunk feeds more MFMAs) -- but a launch of fewer than one workgroup per CU
  // (15 x 20 layers: 75 pixel spans per 32 images) trades that for more workgroups.  An output's sum runs over Cin in the same order either way: same bits
  int nt = ntiles % 4 == 0 ? 4 : ntiles == 3 ? 3 : ntiles % 2 == 0 ? 2 : 1;
  while (nt > 1 && nt != 3 && (long)gx * (ntiles / nt) < 256) nt /= 2;
  if (nt == 4) hipLaunchKernelGGL(nv_pw_mfma_kernel<4>, dim3(gx, ntiles / 4), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  else if (nt == 3) hipLaunchKernelGGL(nv_pw_mfma_kernel<3>, dim3(gx, 1), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  else if (nt == 2) hipLaunchKernelGGL(nv_pw_mfma_kernel<2>, dim3(gx, ntiles / 2), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  else hipLaunchKernelGGL(nv_pw_mfma_kernel<1>, dim3(gx, ntiles), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  return hipGetLastError();
}
// aw_pack: the soft-assignment weights as B fragments [ks = D/4][nt = K/16][lane] = Wa[nt*16 + (lane & 15)][ks*4 + (lane >> 4)] (or null)
void pack_nv_assign(const float* aw /*[K][D]*/, int K, int D, float* dst) {
  for (int ks = 0; ks < D / 4; ++ks)
    for (int nt = 0; nt < K / 16; ++nt)
      for (int l = 0; l < 64; ++l) dst[((size_t)ks * (K / 16) + nt) * 64 + l] = aw[(size_t)(nt * 16 + (l & 15)) * D + ks * 4 + (l >> 4)];
}
size_t nv_vlad_part_floats(int np, int D, int K) { return (size_t)((np + VL_PCH - 1) / VL_PCH) * K * (D + 16); }
hipError_t launch_nv_vlad(const float* x, int slabs, long slab_stride, int np, int D, int K, const float* aw, const float* aw_pack,
                          const float* ab, const float* cen, float* part, float* out, int n, hipStream_t s) {
  if (K > 64 || D > 256 || K * D > 8192) return hipErrorInvalidValue;
  const int nchunk = (np + VL_PCH - 1) / VL_PCH;
  if (K == 32 && D == 128 && aw_pack) {
    hipLaunchKernelGGL(nv_vlad_mfma_kernel, dim3(nchunk, n), dim3(256), 0, s, x, slabs, slab_stride, np, aw_pack, ab, part, nchunk);
    hipLaunchKernelGGL(nv_vlad_final_mfma_kernel, dim3(n), dim3(1024), 0, s, part, nchunk, cen, out);
    return hipGetLastError();
  }
  const size_t lds = sizeof(float) * ((size_t)VL_PCH * (D + 1) + (size_t)K * (D + 1) + (size_t)VL_PCH * (K + 1));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nv_vlad_partial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nv_vlad_partial_kernel, dim3(nchunk, n), dim3(256), lds, s, x, slabs, slab_stride, np, D, K, aw, ab, cen, part, nchunk);
  hipLaunchKernelGGL(nv_vlad_final_kernel, dim3(n), dim3(1024), 0, s, part, nchunk, D, K, out);
  return hipGetLastError();
}
hipError_t launch_nv_pca(const float* x, int nfeat, const float* comp, const float* mean, int m, float* y, int n, hipStream_t s) {
  dim3 grid((m + 3) / 4, n);
  hipLaunchKernelGGL(nv_pca_kernel, grid, dim3(256), 0, s, x, nfeat, comp, mean, m, y);
  hipLaunchKernelGGL(nv_l2norm_kernel, dim3(n), dim3(1024), 0, s, y, m);
  return hipGetLastError();
}

}  // namespace d2fe
