// next.hip -- the components either side of the hot path that SURVEY.md section 8(f) ranks "next":
//  (f)-1 fisheye undistort + photometric gain  (FisheyeUndist::undist_id_cuda, d2common/include/d2common/fisheye_undistort.h:152-176)
//  (f)-2 NetVLAD database add/search + gate     (faiss::IndexFlatIP in d2frontend/src/loop_detector.cpp:254-263,300-350)
//  (f)-3 int8 wire codec of descriptors         (d2common/include/d2common/d2frontend_types.h:228-237,260-268,313-351)
// All three are HBM/latency-bound byte and float work: coalesced loads, wave reductions, no MFMA.
#include "kernels.h"

namespace d2fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ unsigned sat_u8(float v) {   // saturate_cast<uchar>(float): round-to-nearest-even, clamp
  if (!(v > 0.f)) return 0u;
  if (v >= 255.f) return 255u;
  return (unsigned)__builtin_rintf(v);
}

// ---- (f)-1: one fused pass instead of remap -> convertTo -> multiply -> convertTo (4 kernels, 3 round trips) -----------------
// The two roundings of the reference (after the remap and after the gain) are both reproduced, in registers.
__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t* __restrict__ src, int sh, int sw, int sstride,
                                                        long src_istride, const float* __restrict__ mapx,
                                                        const float* __restrict__ mapy, const float* __restrict__ gain,
                                                        int npix, uint8_t* __restrict__ dst) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const uint8_t* s = src + (size_t)n * src_istride;
  const float x = mapx[i], y = mapy[i];
  const int x1 = (int)__builtin_floorf(x), y1 = (int)__builtin_floorf(y), x2 = x1 + 1, y2 = y1 + 1;
  auto S = [&](int yy, int xx) -> float {
    return (yy >= 0 && yy < sh && xx >= 0 && xx < sw) ? (float)s[(size_t)yy * sstride + xx] : 0.f;
  };
  float out = 0.f;
  out = out + S(y1, x1) * (((float)x2 - x) * ((float)y2 - y));
  out = out + S(y1, x2) * ((x - (float)x1) * ((float)y2 - y));
  out = out + S(y2, x1) * (((float)x2 - x) * (y - (float)y1));
  out = out + S(y2, x2) * ((x - (float)x1) * (y - (float)y1));
  unsigned u = sat_u8(out);
  if (gain) u = sat_u8((float)u * gain[i]);
  dst[(size_t)n * npix + i] = (uint8_t)u;
}

hipError_t launch_undistort(const uint8_t* src, int sh, int sw, int sstride, long src_istride, const float* mapx,
                            const float* mapy, const float* gain, int dh, int dw, int n, uint8_t* dst, hipStream_t s) {
  const int npix = dh * dw;
  hipLaunchKernelGGL(undistort_kernel, dim3((npix + 255) / 256, n), dim3(256), 0, s, src, sh, sw, sstride, src_istride, mapx,
                     mapy, gain, npix, dst);
  return hipGetLastError();
}

// ---- (f)-1, map generation: FisheyeUndist::generateCylinderMap / genOneUndistMap (fisheye_undistort.h:458-500,559-660) --------
// One thread per map pixel, fp64 like the reference (camodocal works in double); the maps are born in HBM where
// undistort_kernel reads them.  cam = xi k1 k2 p1 p2 gamma1 gamma2 u0 v0 (CataCamera, CataCamera.cc:495-515,617-633).
struct MapArgs { double cam[9]; double q[4]; double f; double iK13, iK23; int width, height, mode; float* mapx; float* mapy; };

__device__ __forceinline__ void mei_space_to_plane(const double* cam, double X, double Y, double Z, double& u, double& v) {
  const double nrm = __builtin_sqrt(X * X + (Y * Y + Z * Z));
  const double z = Z + cam[0] * nrm;
  const double pu0 = X / z, pu1 = Y / z;
  const double mx2 = pu0 * pu0, my2 = pu1 * pu1, mxy = pu0 * pu1, rho2 = mx2 + my2;
  const double rad = cam[1] * rho2 + cam[2] * rho2 * rho2;
  const double d0 = pu0 * rad + 2.0 * cam[3] * mxy + cam[4] * (rho2 + 2.0 * mx2);
  const double d1 = pu1 * rad + 2.0 * cam[4] * mxy + cam[3] * (rho2 + 2.0 * my2);
  u = cam[5] * (pu0 + d0) + cam[7];
  v = cam[6] * (pu1 + d1) + cam[8];
}

__global__ __launch_bounds__(256) void gen_map_kernel(MapArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.width * a.height) return;
  const int x = i % a.width, y = i / a.width;
  double X, Y, Z;
  if (a.mode == 0) {           // cylinder: CylindricalCamera::liftProjective (CylindricalCamera.cc:207-220), m_inv_K11 = m_inv_K22 = 1/f
    const double ik = 1.0 / a.f;
    const double phi = ik * (double)x + a.iK13;
    const double ybr = ik * (double)y + a.iK23;
    Z = __builtin_fabs(phi) > 1.5707963267948966 ? -1.0 : 1.0;
    X = Z * tan(phi);
    Y = ybr * __builtin_sqrt(X * X + Z * Z);
  } else {                     // pinhole: rotation * (x - w/2, y - h/2, f), Eigen quaternion-vector product
    const double w = a.q[0], qx = a.q[1], qy = a.q[2], qz = a.q[3];
    const double vx = (double)x - (double)(unsigned)a.width / 2, vy = (double)y - (double)(unsigned)a.height / 2, vz = a.f;
    double ux = qy * vz - qz * vy, uy = qz * vx - qx * vz, uz = qx * vy - qy * vx;
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const double cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
    X = (vx + w * ux) + cx; Y = (vy + w * uy) + cy; Z = (vz + w * uz) + cz;
  }
  double u, v;
  mei_space_to_plane(a.cam, X, Y, Z, u, v);
  a.mapx[i] = (float)u;
  a.mapy[i] = (float)v;
}

hipError_t launch_gen_map(const double* cam9, const double* q4, int mode, int width, int height, double f, float* mapx,
                          float* mapy, hipStream_t s) {
  MapArgs a;
  for (int i = 0; i < 9; ++i) a.cam[i] = cam9[i];
  for (int i = 0; i < 4; ++i) a.q[i] = q4 ? q4[i] : (i == 0 ? 1.0 : 0.0);
  a.f = f; a.width = width; a.height = height; a.mode = mode; a.mapx = mapx; a.mapy = mapy;
  a.iK13 = -(double)((unsigned)width / 2) / f;     // -cx/fx with cx = imgWidth / 2 (unsigned division, fisheye_undistort.h:494-495)
  a.iK23 = -(double)((unsigned)height / 2) / f;
  hipLaunchKernelGGL(gen_map_kernel, dim3((width * height + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---- A1, variant A / NetVLAD prep: cv::cvtColor(COLOR_BGR2GRAY) + cv::resize(INTER_LINEAR) fused (superpoint_onnx.cpp:76-83,
// mobilenetvlad_onnx.h:51-59).  Integer arithmetic of OpenCV 4.10 as restated in the oracle (orc_bgr2gray, orc_resize_linear_u8):
// one thread per destination pixel, the 4 taps converted to gray on the fly.  HBM-bound: reads <= 4 * channels bytes per pixel.
__device__ __forceinline__ int prep_px(const uint8_t* __restrict__ row, int x, int ch) {
  if (ch == 1) return row[x];
  const uint8_t* p = row + 3 * x;
  return (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
}
__device__ __forceinline__ void resize_coef(int d, int ssize, int dsize, bool clamp_ofs, int& ofs, int& c0, int& c1) {
  const double scale = 1.0 / ((double)dsize / (double)ssize);
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)__builtin_floorf(f);
  f -= (float)s;
  if (clamp_ofs) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  ofs = s;
  c0 = (int)__builtin_rintf((1.f - f) * 2048.f);
  c1 = (int)__builtin_rintf(f * 2048.f);
}
__global__ __launch_bounds__(256) void prep_gray_kernel(const uint8_t* __restrict__ src, int ch, int sw, int sh, int sstride,
                                                        long src_istride, int dw, int dh, uint8_t* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
  if (x >= dw || y >= dh) return;
  const uint8_t* s = src + (size_t)n * src_istride;
  uint8_t* d = dst + (size_t)n * dw * dh;
  if (sw == dw && sh == dh) { d[(size_t)y * dw + x] = (uint8_t)prep_px(s + (size_t)y * sstride, x, ch); return; }
  if (sw == 2 * dw && sh == 2 * dh) {     // cv::resize turns an exact 2x INTER_LINEAR decimation into INTER_AREA
    const uint8_t* r0 = s + (size_t)(2 * y) * sstride; const uint8_t* r1 = r0 + sstride;
    d[(size_t)y * dw + x] = (uint8_t)((prep_px(r0, 2 * x, ch) + prep_px(r0, 2 * x + 1, ch) + prep_px(r1, 2 * x, ch) + prep_px(r1, 2 * x + 1, ch) + 2) >> 2);
    return;
  }
  int sy, b0, b1, sx, a0, a1;
  resize_coef(y, sh, dh, false, sy, b0, b1);
  resize_coef(x, sw, dw, true, sx, a0, a1);
  const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
  const int x1 = sx + 1 < sw ? sx + 1 : sx;
  const uint8_t* r0 = s + (size_t)y0 * sstride; const uint8_t* r1 = s + (size_t)y1 * sstride;
  const int S0 = prep_px(r0, sx, ch) * a0 + prep_px(r0, x1, ch) * a1;
  const int S1 = prep_px(r1, sx, ch) * a0 + prep_px(r1, x1, ch) * a1;
  d[(size_t)y * dw + x] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}
hipError_t launch_prep_gray(const uint8_t* src, int ch, int sw, int sh, int sstride, long src_istride, int n, int dw, int dh,
                            uint8_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(prep_gray_kernel, dim3((dw + 63) / 64, (dh + 3) / 4, n), dim3(256), 0, s, src, ch, sw, sh, sstride, src_istride,
                     dw, dh, dst);
  return hipGetLastError();
}

// ---- (f)-2: flat inner-product database ------------------------------------------------------------------------------------------
// sims[q][i] = <db[i], query[q]>: one wave per database row, the row is read once (coalesced float4) and reused for every
// query of the batch (queries staged in LDS).  HBM-bound: ntotal*dim*4 bytes per search.
__global__ __launch_bounds__(256) void db_sims_kernel(const float* __restrict__ db, int ntotal, int dim,
                                                      const float* __restrict__ q, int nq, float* __restrict__ sims) {
  extern __shared__ float qs[];   // [nq][dim]
  for (int i = threadIdx.x; i < nq * dim; i += 256) qs[i] = q[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= ntotal) return;
  const float* r = db + (size_t)row * dim;
  for (int qi = 0; qi < nq; ++qi) {
    float a = 0.f;
    for (int j = lane * 4; j < dim; j += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(r + j);
      const float* qq = qs + qi * dim + j;
      a = __builtin_fmaf(v[0], qq[0], a); a = __builtin_fmaf(v[1], qq[1], a);
      a = __builtin_fmaf(v[2], qq[2], a); a = __builtin_fmaf(v[3], qq[3], a);
    }
    a = wsum(a);
    if (lane == 0) sims[(size_t)qi * ntotal + row] = a;
  }
}

// top-k per query by (similarity desc, label asc): k rounds of a block-wide arg-max over keys (sim bits, ~label).
__global__ __launch_bounds__(1024) void db_topk_kernel(float* __restrict__ sims, int ntotal, int k, int32_t* __restrict__ labels,
                                                       float* __restrict__ out_sims) {
  __shared__ unsigned long long red[16];
  const int qi = blockIdx.x, tid = threadIdx.x;
  float* s = sims + (size_t)qi * ntotal;
  for (int r = 0; r < k; ++r) {
    unsigned long long best = 0;
    for (int i = tid; i < ntotal; i += 1024) {
      const float v = s[i];
      unsigned b = __float_as_uint(v);
      b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // order-preserving map of float to unsigned
      const unsigned long long key = ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
      if (__float_as_uint(v) != 0xFFFFFFFFu && key > best) best = key;   // all-ones NaN pattern marks "already taken"
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_xor(best, o, 64);
      best = t > best ? t : best;
    }
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long b = 0;
      for (int w = 0; w < 16; ++w) b = red[w] > b ? red[w] : b;
      const int idx = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull));
      labels[(size_t)qi * k + r] = b ? idx : -1;
      out_sims[(size_t)qi * k + r] = b ? s[idx] : 0.f;
      if (b) s[idx] = __uint_as_float(0xFFFFFFFFu);
    }
    __syncthreads();
  }
}

hipError_t launch_db_search(const float* db, int ntotal, int dim, const float* q, int nq, int k, float* sims_scratch,
                            int32_t* labels, float* out_sims, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)nq * dim;
  if (lds > 64 * 1024 || (dim & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(db_sims_kernel, dim3((ntotal + 3) / 4), dim3(256), lds, s, db, ntotal, dim, q, nq, sims_scratch);
  hipLaunchKernelGGL(db_topk_kernel, dim3(nq), dim3(1024), 0, s, sims_scratch, ntotal, k, labels, out_sims);
  return hipGetLastError();
}

// ---- (f)-3: int8 codec ----------------------------------------------------------------------------------------------------------------
// quantise: q = (int8)(x / max|x| * 127) (C cast = truncation); max over the whole tensor (one block per tensor).
__global__ __launch_bounds__(1024) void quant_int8_kernel(const float* __restrict__ x, int n, int double_max,
                                                          int8_t* __restrict__ out) {
  __shared__ float red[16];
  const int tid = threadIdx.x;
  float m = 0.f;
  for (int i = tid; i < n; i += 1024) m = fmaxf(m, __builtin_fabsf(x[i]));
  m = wmax(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = 0.f;
  for (int w = 0; w < 16; ++w) m = fmaxf(m, red[w]);
  if (double_max) {
    const double md = (double)m;
    for (int i = tid; i < n; i += 1024) out[i] = (int8_t)(int)((double)x[i] / md * 127.0);
  } else {
    for (int i = tid; i < n; i += 1024) out[i] = (int8_t)(int)(x[i] / m * 127.0f);
  }
}
// dequantise: x = (float)(q / 127.0); landmark descriptors: every 32-float segment i < landmark_num re-normalised
// (the reference's hard-coded 32, d2frontend_types.h:326-328); global descriptor (landmark_num < 0): whole-vector L2.
__global__ __launch_bounds__(1024) void dequant_int8_kernel(const int8_t* __restrict__ q, int n, int landmark_num,
                                                            float* __restrict__ out) {
  __shared__ float red[16];
  const int tid = threadIdx.x;
  if (landmark_num >= 0) {
    for (int i = tid; i < n; i += 1024) {
      const float v = (float)((double)q[i] / 127.0);
      const int seg = i >> 5;
      // 32 consecutive lanes hold one segment: reduce within the half-wave
      float s = v * v;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const bool full = (seg + 1) * 32 <= n;
      // Eigen's normalize(): z = squaredNorm(); if (z > 0) derived() /= sqrt(z) -- an all-zero segment stays zero
      out[i] = (seg < landmark_num && full && s > 0.f) ? v / __builtin_sqrtf(s) : v;
    }
  } else {
    float s = 0.f;
    for (int i = tid; i < n; i += 1024) { const float v = (float)((double)q[i] / 127.0); s += v * v; }
    s = wsum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    const float nr = __builtin_sqrtf(t);
    for (int i = tid; i < n; i += 1024) { const float v = (float)((double)q[i] / 127.0); out[i] = t > 0.f ? v / nr : v; }
  }
}

hipError_t launch_quant_int8(const float* x, int n, int double_max, int8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(quant_int8_kernel, dim3(1), dim3(1024), 0, s, x, n, double_max, out);
  return hipGetLastError();
}
hipError_t launch_dequant_int8(const int8_t* q, int n, int landmark_num, float* out, hipStream_t s) {
  hipLaunchKernelGGL(dequant_int8_kernel, dim3(1), dim3(1024), 0, s, q, n, landmark_num, out);
  return hipGetLastError();
}

}  // namespace d2fe
