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
  for (int c0 = 0; c0 < Cin; c0 += PW_CK) {
    const int cc = (Cin - c0) < PW_CK ? (Cin - c0) : PW_CK;
    const int cc4 = cc / 4;
    __syncthreads();
    for (int i = tid; i < 128 * cc4; i += 256) {
      const int pp = i / cc4, q = i % cc4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pp < npix) v = *reinterpret_cast<const f32x4*>(in + (size_t)(p0 + pp) * Cin + c0 + q * 4);
      float* d = xs + pp * (PW_CK + 1) + q * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const float* ap = xs + (wave * 32 + (lane & 31)) * (PW_CK + 1) + (lane >> 5);
    for (int c8 = 0; c8 < cc / 8; ++c8) {
      f32x4 bw[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) bw[n] = wpack[((size_t)(nt0 + n) * c8n + (c0 / 8 + c8)) * 64 + lane];
      float av[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) av[q] = ap[c8 * 8 + 2 * q];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bw[n][q], acc[n], 0, 0, 0);
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

__global__ __launch_bounds__(1024) void nv_vlad_final_kernel(const float* __restrict__ part, int nchunk, int D, int K,
                                                             float* __restrict__ out) {
  __shared__ float red[130];
  const int img = blockIdx.x, tid = threadIdx.x;
  const int KD = K * D;
  float v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = tid + 1024 * r;
    float s = 0.f;
    if (e < KD)
      for (int c = 0; c < nchunk; ++c) s += part[((size_t)img * nchunk + c) * KD + e];
    v[r] = s;
  }
  for (int i = tid; i < 130; i += 1024) red[i] = 0.f;
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
  if ((tid & 63) == 0) atomicAdd(&red[128], tot);
  __syncthreads();
  const float nt = __builtin_sqrtf(red[128]);
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
  const int ntiles = CoutPad / 32;
  const f32x4* wp = reinterpret_cast<const f32x4*>(w);
  const unsigned gx = (unsigned)((P + 127) / 128);
  if (ntiles % 4 == 0) hipLaunchKernelGGL(nv_pw_mfma_kernel<4>, dim3(gx, ntiles / 4), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  else if (ntiles == 3) hipLaunchKernelGGL(nv_pw_mfma_kernel<3>, dim3(gx, 1), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  else if (ntiles % 2 == 0) hipLaunchKernelGGL(nv_pw_mfma_kernel<2>, dim3(gx, ntiles / 2), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  else hipLaunchKernelGGL(nv_pw_mfma_kernel<1>, dim3(gx, ntiles), dim3(256), 0, s, in, P, Cin, Cout, act, wp, b, res, out);
  return hipGetLastError();
}
hipError_t launch_nv_vlad(const float* x, int slabs, long slab_stride, int np, int D, int K, const float* aw, const float* ab,
                          const float* cen, float* part, float* out, int n, hipStream_t s) {
  if (K > 64 || D > 256 || K * D > 8192) return hipErrorInvalidValue;
  const int nchunk = (np + VL_PCH - 1) / VL_PCH;
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
