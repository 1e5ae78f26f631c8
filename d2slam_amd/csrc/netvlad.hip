// netvlad.hip -- global image descriptor (MobileNetVLAD) on gfx950.
//
// Replaces MobileNetVLADONNX::inference (d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:49-74: ONNX Runtime
// session run + optional PCA).  The reference's graph is NOT in its tree; the layer kinds below
// execute whatever flat layer list d2fe_load_netvlad() is given -- the documented stand-in lives in
// d2slam_amd/netvlad.py and oracle/d2fe_oracle.c (A9 block).  All of it is 0.33 GMAC per 640x480 image (0.6 % of
// SuperPoint), HBM/latency bound: plain fp32 VALU kernels, weights through the scalar cache, activations NHWC.
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

// ---- pointwise (1x1) conv: [P][Cin] x [Cin][CoutPad] ----------------------------------------------------------------------------
// Block = 64 pixels x 128 output channels; lane = pixel, wave = 32 output channels held as 32 accumulators.  The pixel tile
// is one contiguous span of NHWC memory (coalesced float4 staging into LDS, Cin chunked by 256); weights for a wave's 32
// channels are wave-uniform and arrive through scalar loads (v_fma with an SGPR operand).
constexpr int PW_CH = 256;
__global__ __launch_bounds__(256) void nv_pw_kernel(const float* __restrict__ in, long P, int Cin, int Cout, int CoutPad,
                                                    int act, const float* __restrict__ w /*[Cin][CoutPad]*/,
                                                    const float* __restrict__ b /*[CoutPad]*/,
                                                    const float* __restrict__ res /*nullable [P][Cout]*/,
                                                    float* __restrict__ out) {
  __shared__ float xs[64 * (PW_CH + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long p0 = (long)blockIdx.x * 64;
  const int co0 = blockIdx.y * 128 + wave * 32;
  float acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  const int npix = (int)((P - p0) < 64 ? (P - p0) : 64);
  for (int c0 = 0; c0 < Cin; c0 += PW_CH) {
    const int cc = (Cin - c0) < PW_CH ? (Cin - c0) : PW_CH;
    __syncthreads();
    // stage [npix][cc] (Cin and cc are multiples of 4)
    const int cc4 = cc / 4;
    for (int i = tid; i < npix * cc4; i += 256) {
      const int pp = i / cc4, q = i % cc4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(in + (size_t)(p0 + pp) * Cin + c0 + q * 4);
      float* d = xs + pp * (PW_CH + 1) + q * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    if (co0 < CoutPad) {
      const float* wp = w + (size_t)c0 * CoutPad + co0;
      const float* xp = xs + lane * (PW_CH + 1);
      for (int ci = 0; ci < cc; ++ci) {
        const float xv = xp[ci];
        const float* wr = wp + (size_t)ci * CoutPad;
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
      }
    }
  }
  if (co0 >= CoutPad || lane >= npix) return;
  const long p = p0 + lane;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int co = co0 + j;
    if (co < Cout) {
      float v = acc[j] + b[co];
      if (res) v += res[(size_t)p * Cout + co];
      out[(size_t)p * Cout + co] = nv_act(v, act);
    }
  }
}

// ---- NetVLAD head: soft-assignment, residual aggregation, intra + global L2 ----------------------------------------------------
// One 1024-thread block per image.  x: [np][D] (pre-projected features).  out: [K*D].  K <= 64, D <= 256, K*D <= 8192.
constexpr int VL_PCH = 128;  // positions per chunk kept in LDS
__global__ __launch_bounds__(1024) void nv_vlad_kernel(const float* __restrict__ x, int np, int D, int K,
                                                       const float* __restrict__ aw /*[K][D]*/, const float* __restrict__ ab,
                                                       const float* __restrict__ cen /*[K][D]*/, float* __restrict__ out) {
  extern __shared__ float sm[];
  float* a = sm;                       // [VL_PCH][K] memberships of the current chunk
  float* red = a + VL_PCH * 64;        // [64] per-cluster norms, [64] scratch
  const int img = blockIdx.x, tid = threadIdx.x;
  const float* xi = x + (size_t)img * np * D;
  const int KD = K * D;
  float v[8];                          // this thread's V entries: e = tid + 1024*r
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = 0.f;
  for (int pc = 0; pc < np; pc += VL_PCH) {
    const int pn = (np - pc) < VL_PCH ? (np - pc) : VL_PCH;
    __syncthreads();
    // memberships: thread (p, k) = one logit
    for (int i = tid; i < pn * K; i += 1024) {
      const int p = i / K, k = i % K;
      float s = ab[k];
      const float* xp = xi + (size_t)(pc + p) * D;
      const float* wk = aw + (size_t)k * D;
      for (int j = 0; j < D; ++j) s = __builtin_fmaf(xp[j], wk[j], s);
      a[p * 64 + k] = s;
    }
    __syncthreads();
    for (int p = tid; p < pn; p += 1024) {
      float m = -__builtin_inff();
      for (int k = 0; k < K; ++k) m = a[p * 64 + k] > m ? a[p * 64 + k] : m;
      float sum = 0.f;
      for (int k = 0; k < K; ++k) { const float e = __expf(a[p * 64 + k] - m); a[p * 64 + k] = e; sum += e; }
      for (int k = 0; k < K; ++k) a[p * 64 + k] = a[p * 64 + k] / sum;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int e = tid + 1024 * r;
      if (e < KD) {
        const int k = e / D, j = e % D;
        const float c = cen[e];
        float acc = v[r];
        for (int p = 0; p < pn; ++p) acc = __builtin_fmaf(a[p * 64 + k], c - xi[(size_t)(pc + p) * D + j], acc);
        v[r] = acc;
      }
    }
  }
  // intra-normalisation (per cluster over D), then global L2
  __syncthreads();
  for (int i = tid; i < 128; i += 1024) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = tid + 1024 * r;
    if (e < KD) atomicAdd(&red[e / D], v[r] * v[r]);
  }
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = tid + 1024 * r;
    if (e < KD) {
      const float n = __builtin_sqrtf(red[e / D]);
      v[r] = v[r] / (n > 1e-12f ? n : 1e-12f);
      tot += v[r] * v[r];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  if ((tid & 63) == 0) atomicAdd(&red[64], tot);
  __syncthreads();
  const float nt = __builtin_sqrtf(red[64]);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = tid + 1024 * r;
    if (e < KD) out[(size_t)img * KD + e] = v[r] / (nt > 1e-12f ? nt : 1e-12f);
  }
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
  dim3 grid((unsigned)((P + 63) / 64), (CoutPad + 127) / 128);
  hipLaunchKernelGGL(nv_pw_kernel, grid, dim3(256), 0, s, in, P, Cin, Cout, CoutPad, act, w, b, res, out);
  return hipGetLastError();
}
hipError_t launch_nv_vlad(const float* x, int np, int D, int K, const float* aw, const float* ab, const float* cen, float* out,
                          int n, hipStream_t s) {
  if (K > 64 || D > 256 || K * D > 8192) return hipErrorInvalidValue;
  const size_t lds = sizeof(float) * (VL_PCH * 64 + 128);
  hipLaunchKernelGGL(nv_vlad_kernel, dim3(n), dim3(1024), lds, s, x, np, D, K, aw, ab, cen, out);
  return hipGetLastError();
}
hipError_t launch_nv_pca(const float* x, int nfeat, const float* comp, const float* mean, int m, float* y, int n, hipStream_t s) {
  dim3 grid((m + 3) / 4, n);
  hipLaunchKernelGGL(nv_pca_kernel, grid, dim3(256), 0, s, x, nfeat, comp, mean, m, y);
  hipLaunchKernelGGL(nv_l2norm_kernel, dim3(n), dim3(1024), 0, s, y, m);
  return hipGetLastError();
}

}  // namespace d2fe
