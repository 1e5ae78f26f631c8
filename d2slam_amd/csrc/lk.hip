// lk.hip -- SURVEY.md section 8(f)-4: the LK optical-flow tracker of d2frontend on the device.
//   buildImagePyramid   d2frontend/src/opticaltrack_utils.cpp:526-542 (cv::cuda::pyrDown)
//   opticalflowTrackPyr :173-279 (cv::cuda::SparsePyrLKOpticalFlow forward + reverse, 0.5 px check, inBorder)
//   detectFastByRegion  :444-493 (cv::cuda::FastFeatureDetector per region), detectPoints :375-442 (good features)
// The arithmetic is OpenCV 4.10's (not in the reference tree); the evaluation orders fixed by the oracle
// (oracle/d2fe_oracle_lk.c, header) are reproduced operation by operation, so every output is compared bitwise.
//
// MI355X mapping: one 64-lane wave per tracked point = OpenCV's 8x8 thread block; the whole bidirectional track of a
// point (3 pyramid levels forward, 3 back, the 0.5 px test and the border test) is ONE kernel launch instead of
// 6 launches + 4 PCIe round trips; the 21x21 window lives in 27 registers per lane; reductions are 6 DPP/shuffle steps.
// All of it is latency-bound byte/float work (150 points x 441 px): no MFMA, no LDS staging needed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/d2fe.h"
#include "kernels.h"

namespace d2fe {

namespace {

__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

// ---- pyrDown: 5x5 [1 4 6 4 1]^2 / 256, reflect-101 border, round half to even --------------------------------------------
__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t* __restrict__ src, int w, int h, uint8_t* __restrict__ dst,
                                                       int dw, int dh) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const int k[5] = {1, 4, 6, 4, 1};
  int xs[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) xs[j] = reflect101(2 * x + j - 2, w);
  int s = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const uint8_t* row = src + (size_t)reflect101(2 * y + i - 2, h) * w;
    int r = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) r += k[j] * row[xs[j]];
    s += k[i] * r;
  }
  int q = s >> 8;
  const int rem = s & 255;
  if (rem > 128 || (rem == 128 && (q & 1))) ++q;
  dst[(size_t)y * dw + x] = (uint8_t)(q > 255 ? 255 : q);
}

// ---- sparse pyramidal LK ---------------------------------------------------------------------------------------------------------
// one (previous frame, current frame) pair of a batched call; a point carries the index of its pair
struct LkPairDev {
  const uint8_t* prev; const uint8_t* cur;
  int off[8], ws[8], hs[8];
  int levels, w, h, type;
  float move_cols;
  int pad_[3];
};
struct LkArgs {
  const LkPairDev* pairs; const int* pair_of;
  int n, win, iters;
  const float* prev_pts; const float* cur_init;
  float* cur_pts; uint8_t* status;
};

__device__ __forceinline__ float tex(const uint8_t* __restrict__ im, int w, int h, float x, float y) {
  const float xs = x - 0.5f, ys = y - 0.5f;
  const float xf = __builtin_floorf(xs), yf = __builtin_floorf(ys);
  const float fx = xs - xf, fy = ys - yf;
  int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), w - 1); x1 = min(max(x1, 0), w - 1);
  y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
  const float s = 1.0f / 255.0f;
  const float p00 = (float)im[(size_t)y0 * w + x0] * s, p10 = (float)im[(size_t)y0 * w + x1] * s;
  const float p01 = (float)im[(size_t)y1 * w + x0] * s, p11 = (float)im[(size_t)y1 * w + x1] * s;
  const float gx = 1.0f - fx, gy = 1.0f - fy;
  float v = (gx * gy) * p00;
  v = v + (fx * gy) * p10;
  v = v + (gx * fy) * p01;
  v = v + (fx * fy) * p11;
  return v;
}

// the shared-memory tree of OpenCV's block reduce, v[t] += v[t+s] for s = 32..1, result broadcast from lane 0
__device__ __forceinline__ float tree64(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = v + __shfl_down(v, s, 64);
  return __shfl(v, 0, 64);
}

// one pyramid level for one point, executed by a whole wave with uniform control flow
__device__ void lk_level(const uint8_t* __restrict__ I, const uint8_t* __restrict__ J, int cols, int rows, int level, int win,
                         int iters, float ppx, float ppy, float& npx, float& npy, int& status, int lane) {
  const float half = (float)((win - 1) / 2);
  float px = ppx * (1.0f / (float)(1 << level)), py = ppy * (1.0f / (float)(1 << level));
  if (px < 0 || px >= (float)cols || py < 0 || py >= (float)rows) {
    if (level == 0) status = 0;
    return;
  }
  px -= half; py -= half;
  const int tx = lane & 7, ty = lane >> 3;
  float Ip[3][3], Dx[3][3], Dy[3][3];
  float s11 = 0.f, s12 = 0.f, s22 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int xb = tx + 8 * j, yb = ty + 8 * i;
      Ip[i][j] = 0.f; Dx[i][j] = 0.f; Dy[i][j] = 0.f;
      if (xb < win && yb < win) {
        const float x = px + (float)xb + 0.5f, y = py + (float)yb + 0.5f;
        Ip[i][j] = tex(I, cols, rows, x, y);
        const float tmm = tex(I, cols, rows, x - 1, y - 1), tpm = tex(I, cols, rows, x + 1, y - 1);
        const float tmp = tex(I, cols, rows, x - 1, y + 1), tpp = tex(I, cols, rows, x + 1, y + 1);
        float dx = 3.0f * tpm;
        dx = dx + 10.0f * tex(I, cols, rows, x + 1, y);
        dx = dx + 3.0f * tpp;
        float mx = 3.0f * tmm;
        mx = mx + 10.0f * tex(I, cols, rows, x - 1, y);
        mx = mx + 3.0f * tmp;
        dx = dx - mx;
        float dy = 3.0f * tmp;
        dy = dy + 10.0f * tex(I, cols, rows, x, y + 1);
        dy = dy + 3.0f * tpp;
        float my = 3.0f * tmm;
        my = my + 10.0f * tex(I, cols, rows, x, y - 1);
        my = my + 3.0f * tpm;
        dy = dy - my;
        Dx[i][j] = dx; Dy[i][j] = dy;
        s11 = s11 + dx * dx; s12 = s12 + dx * dy; s22 = s22 + dy * dy;
      }
    }
  float A11 = tree64(s11), A12 = tree64(s12), A22 = tree64(s22);
  float D = A11 * A22 - A12 * A12;
  if (D < 1.1920928955078125e-07f) {
    if (level == 0) status = 0;
    return;
  }
  D = 1.0f / D;
  A11 = A11 * D; A12 = A12 * D; A22 = A22 * D;
  float nx = npx * 2.0f, ny = npy * 2.0f;
  nx -= half; ny -= half;
  for (int k = 0; k < iters; ++k) {
    if (nx < -half || nx >= (float)cols || ny < -half || ny >= (float)rows) {
      if (level == 0) status = 0;
      return;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int xb = tx + 8 * j, yb = ty + 8 * i;
        if (xb < win && yb < win) {
          const float Jv = tex(J, cols, rows, nx + (float)xb + 0.5f, ny + (float)yb + 0.5f);
          const float diff = (Jv - Ip[i][j]) * 32.0f;
          s1 = s1 + diff * Dx[i][j];
          s2 = s2 + diff * Dy[i][j];
        }
      }
    const float B1 = tree64(s1), B2 = tree64(s2);
    const float ddx = A12 * B2 - A22 * B1;
    const float ddy = A12 * B1 - A11 * B2;
    nx = nx + ddx; ny = ny + ddy;
    if (__builtin_fabsf(ddx) < 0.01f && __builtin_fabsf(ddy) < 0.01f) break;
  }
  npx = nx + half; npy = ny + half;
}

__device__ __forceinline__ void lk_calc(const LkArgs& a, const LkPairDev& P, const uint8_t* Ip, const uint8_t* Jp, float ppx,
                                        float ppy, float& npx, float& npy, int& status, int lane) {
  const float sc = (float)(1.0 / (double)(1 << P.levels) / 2.0);
  npx = npx * sc; npy = npy * sc;
  status = 1;
  for (int l = P.levels; l >= 0; --l)
    lk_level(Ip + P.off[l], Jp + P.off[l], P.ws[l], P.hs[l], l, a.win, a.iters, ppx, ppy, npx, npy, status, lane);
}

__global__ __launch_bounds__(256) void lk_track_kernel(LkArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= a.n) return;
  const LkPairDev& P = a.pairs[__builtin_amdgcn_readfirstlane(a.pair_of[i])];   // wave-uniform: scalar loads
  const float ppx = a.prev_pts[2 * i], ppy = a.prev_pts[2 * i + 1];
  float cx = a.cur_init[2 * i], cy = a.cur_init[2 * i + 1];
  int st = 1, rst = 1;
  lk_calc(a, P, P.prev, P.cur, ppx, ppy, cx, cy, st, lane);
  float rx = cx, ry = cy;
  if (P.type == 1 && st == 1) rx -= P.move_cols;
  if (P.type == 2 && st == 1) rx += P.move_cols;
  lk_calc(a, P, P.cur, P.prev, cx, cy, rx, ry, rst, lane);
  const float dx = ppx - rx, dy = ppy - ry;
  const double nrm = __builtin_sqrt((double)dx * dx + (double)dy * dy);
  int ok = (st && rst && nrm <= 0.5) ? 1 : 0;
  if (ok) {
    const int ix = (int)__builtin_rint((double)cx), iy = (int)__builtin_rint((double)cy);
    if (!(1 <= ix && ix < P.w - 1 && 1 <= iy && iy < P.h - 1)) ok = 0;
  }
  if (lane == 0) {
    a.cur_pts[2 * i] = cx; a.cur_pts[2 * i + 1] = cy;
    a.status[i] = (uint8_t)ok;
  }
}

// ---- FAST-9/16 ---------------------------------------------------------------------------------------------------------------------
// m = max over the 16 arcs of 9 contiguous circle pixels of min(q - v) (bright) and of min(v - q) (dark):
// the pixel is a corner at threshold t iff m > t, and OpenCV's CUDA cornerScore (largest passing threshold) is m - 1.
__device__ __forceinline__ int fast_strength(const uint8_t* __restrict__ p, int stride) {
  const int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  const int DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const int v = p[0];
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) d[k] = (int)p[DY[k] * stride + DX[k]] - v;
  int best = -256;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    int mn = 256, mx = -256;
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int q = d[(s + k) & 15]; mn = min(mn, q); mx = max(mx, q); }
    best = max(best, max(mn, -mx));
  }
  return best;
}

// one workgroup per region: raster scan of the region interior in chunks of 1024 pixels; the first `features` corners (raster
// order -- the deterministic stand-in for the CUDA detector's atomics-ordered max_npoints cap) get their score written
__global__ __launch_bounds__(1024) void fast_region_kernel(const uint8_t* __restrict__ img, int stride, int sw, int sh, int rows,
                                                           int features, int threshold, int* __restrict__ score, int w) {
  __shared__ int wsum[16];
  __shared__ int s_base;
  const int region = blockIdx.x, ri = region / rows, rj = region % rows;
  const int x0 = sw * ri, y0 = sh * rj;
  const int iw = sw - 6, ih = sh - 6;
  if (iw <= 0 || ih <= 0) return;
  const int total = iw * ih;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < total; c0 += 1024) {
    const int idx = c0 + tid;
    int m = -256, gx = 0, gy = 0;
    if (idx < total) {
      gx = x0 + 3 + idx % iw; gy = y0 + 3 + idx / iw;
      m = fast_strength(img + (size_t)gy * stride + gx, stride);
    }
    const bool flag = m > threshold;
    const unsigned long long bal = __ballot(flag);
    const int inw = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int c = wsum[k]; pre += k < wv ? c : 0; tot += c; }
    const int base = s_base;
    if (flag && base + pre + inw < features) score[(size_t)gy * w + gx] = m - 1;
    __syncthreads();
    if (tid == 0) s_base = base + tot;
    __syncthreads();
    if (base + tot >= features) break;
  }
}

// non-max suppression (strictly greater than the 8 neighbours) + emit {x, y, response, order}
__global__ __launch_bounds__(256) void fast_nonmax_kernel(const int* __restrict__ score, int w, int h, int sw, int sh, int cols,
                                                          int rows, int4* __restrict__ out, int cap, int* __restrict__ count) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) return;
  const int s = score[(size_t)y * w + x];
  if (s <= 0) return;
  const int* r0 = score + (size_t)(y - 1) * w + x; const int* r1 = r0 + w; const int* r2 = r1 + w;
  if (s > r0[-1] && s > r0[0] && s > r0[1] && s > r1[-1] && s > r1[1] && s > r2[-1] && s > r2[0] && s > r2[1]) {
    const int ri = x / sw, rj = y / sh;
    if (ri >= cols || rj >= rows) return;
    const int order = (ri * rows + rj) * (sw * sh) + (y - sh * rj) * sw + (x - sw * ri);
    const int k = atomicAdd(count, 1);
    if (k < cap) out[k] = int4{x, y, s, order};
  }
}

// ---- good features to track -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sobel_kernel(const uint8_t* __restrict__ img, int stride, int w, int h, float k0, float k1,
                                                    float* __restrict__ dx, float* __restrict__ dy) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const int ys[3] = {reflect101(y - 1, h), y, reflect101(y + 1, h)};
  float t0[3], t1[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const uint8_t* row = img + (size_t)ys[r] * stride;
    const float a = (float)row[xm], b = (float)row[x], c = (float)row[xp];
    t0[r] = c - a;
    t1[r] = (a + 2.0f * b) + c;
  }
  float s = k0 * t0[0];
  s = s + k1 * t0[1];
  s = s + k0 * t0[2];
  dx[(size_t)y * w + x] = s;
  dy[(size_t)y * w + x] = k0 * t1[2] - k0 * t1[0];
}

__global__ __launch_bounds__(256) void min_eigen_kernel(const float* __restrict__ dx, const float* __restrict__ dy, int w, int h,
                                                        float* __restrict__ eig, unsigned* __restrict__ maxbits) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  float e = 0.f;
  if (x < w && y < h) {
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
      const size_t ro = (size_t)reflect101(y + i, h) * w;
#pragma unroll
      for (int j = -1; j <= 1; ++j) {
        const size_t o = ro + reflect101(x + j, w);
        const float gx = dx[o], gy = dy[o];
        a = a + gx * gx; b = b + gx * gy; c = c + gy * gy;
      }
    }
    a = a * 0.5f; c = c * 0.5f;
    e = (a + c) - __builtin_sqrtf((a - c) * (a - c) + b * b);
    eig[(size_t)y * w + x] = e;
  }
  // max over the image: non-negative floats order like their bit patterns (negative / NaN values never win against 0)
  float m = e > 0.f ? e : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(maxbits, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void corners_kernel(const float* __restrict__ eig, int w, int h, const unsigned* __restrict__ maxbits,
                                                      double quality, unsigned long long* __restrict__ out, int cap,
                                                      int* __restrict__ count) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) return;
  const float thr = (float)((double)__uint_as_float(*maxbits) * quality);
  const float v = eig[(size_t)y * w + x];
  if (!(v > thr)) return;
  float m = v;
#pragma unroll
  for (int i = -1; i <= 1; ++i)
#pragma unroll
    for (int j = -1; j <= 1; ++j) m = fmaxf(m, eig[(size_t)(y + i) * w + x + j]);
  if (v == m) {
    const int k = atomicAdd(count, 1);
    // v > thr >= 0: the bit pattern orders like the value; low word = ~raster index (ties: lower index first)
    if (k < cap) out[k] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(y * w + x));
  }
}

}  // namespace

// internal accessors of the context (api.hip)
int ctx_fail(int code, const std::string& msg);
int ctx_device(d2fe_handle h);
hipStream_t ctx_stream(d2fe_handle h);
int ctx_scratch(d2fe_handle h, size_t bytes, void** out);

}  // namespace d2fe

using namespace d2fe;

struct d2fe_lk_frame_s {
  int device = 0;                 // the frame outlives nothing but must not touch the handle after creation
  int w = 0, hgt = 0, levels = 0, total = 0;
  int off[8], ws[8], hs[8];
  uint8_t* pyr = nullptr;
};

#define LK_TRY(expr)                                                                                                         \
  do {                                                                                                                       \
    hipError_t e_ = (expr);                                                                                                  \
    if (e_ != hipSuccess) return ctx_fail(D2FE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " @lk.hip:" + std::to_string(__LINE__)); \
  } while (0)

namespace {

int frame_alloc(d2fe_handle h, int width, int height, int levels, d2fe_lk_frame* out) {
  if (!h || !out) return ctx_fail(D2FE_ERR_INVALID, "null argument");
  *out = nullptr;
  if (width < 16 || height < 16 || levels < 0 || levels > 7) return ctx_fail(D2FE_ERR_INVALID, "bad pyramid geometry");
  LK_TRY(hipSetDevice(ctx_device(h)));
  d2fe_lk_frame f = new d2fe_lk_frame_s();
  f->device = ctx_device(h); f->w = width; f->hgt = height; f->levels = levels;
  int o = 0, w = width, hh = height;
  for (int l = 0; l <= levels; ++l) {
    f->off[l] = o; f->ws[l] = w; f->hs[l] = hh;
    o += w * hh; w = (w + 1) / 2; hh = (hh + 1) / 2;
  }
  f->total = o;
  if (hipMalloc(&f->pyr, (size_t)o) != hipSuccess) { delete f; return ctx_fail(D2FE_ERR_HIP, "hipMalloc pyramid"); }
  *out = f;
  return D2FE_OK;
}

hipError_t frame_build(d2fe_lk_frame f, hipStream_t s) {
  for (int l = 1; l <= f->levels; ++l) {
    hipLaunchKernelGGL(pyr_down_kernel, dim3((f->ws[l] + 63) / 64, (f->hs[l] + 3) / 4), dim3(256), 0, s, f->pyr + f->off[l - 1],
                       f->ws[l - 1], f->hs[l - 1], f->pyr + f->off[l], f->ws[l], f->hs[l]);
  }
  return hipGetLastError();
}

}  // namespace

extern "C" {

int d2fe_lk_frame_create(d2fe_handle h, const uint8_t* gray, int width, int height, int stride, int levels, d2fe_lk_frame* out) {
  if (!gray || stride < width) return ctx_fail(D2FE_ERR_INVALID, "null image / bad stride");
  const int rc = frame_alloc(h, width, height, levels, out);
  if (rc != D2FE_OK) return rc;
  d2fe_lk_frame f = *out;
  hipStream_t s = ctx_stream(h);
  hipError_t e = hipMemcpy2DAsync(f->pyr, (size_t)width, gray, (size_t)stride, (size_t)width, (size_t)height, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = frame_build(f, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { d2fe_lk_frame_destroy(f); *out = nullptr; return ctx_fail(D2FE_ERR_HIP, std::string("lk frame: ") + hipGetErrorString(e)); }
  return D2FE_OK;
}

int d2fe_lk_frame_create_device(d2fe_handle h, const uint8_t* d_gray, int width, int height, int stride, int levels, void* stream,
                                d2fe_lk_frame* out) {
  if (!d_gray || stride < width) return ctx_fail(D2FE_ERR_INVALID, "null image / bad stride");
  const int rc = frame_alloc(h, width, height, levels, out);
  if (rc != D2FE_OK) return rc;
  d2fe_lk_frame f = *out;
  hipStream_t s = stream ? (hipStream_t)stream : ctx_stream(h);
  hipError_t e = hipMemcpy2DAsync(f->pyr, (size_t)width, d_gray, (size_t)stride, (size_t)width, (size_t)height, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = frame_build(f, s);
  if (e != hipSuccess) { d2fe_lk_frame_destroy(f); *out = nullptr; return ctx_fail(D2FE_ERR_HIP, std::string("lk frame: ") + hipGetErrorString(e)); }
  return D2FE_OK;
}

void d2fe_lk_frame_destroy(d2fe_lk_frame f) {
  if (!f) return;
  hipSetDevice(f->device);
  if (f->pyr) hipFree(f->pyr);
  delete f;
}

long d2fe_lk_frame_read_level(d2fe_lk_frame f, int level, uint8_t* dst, size_t max_bytes, int* width, int* height) {
  if (!f || !dst || level < 0 || level > f->levels) return ctx_fail(D2FE_ERR_INVALID, "bad argument");
  const size_t bytes = (size_t)f->ws[level] * f->hs[level];
  if (bytes > max_bytes) return ctx_fail(D2FE_ERR_TRUNCATED, "destination too small");
  LK_TRY(hipSetDevice(f->device));
  LK_TRY(hipDeviceSynchronize());
  LK_TRY(hipMemcpy(dst, f->pyr + f->off[level], bytes, hipMemcpyDeviceToHost));
  if (width) *width = f->ws[level];
  if (height) *height = f->hs[level];
  return (long)bytes;
}

int d2fe_lk_track_batch(d2fe_handle h, const d2fe_lk_pair* pairs, int npairs, const float* prev_pts, const float* cur_init,
                        int n_total, int win, int iters, float* cur_pts, uint8_t* status) {
  if (!h || (npairs > 0 && !pairs) || (n_total > 0 && (!prev_pts || !cur_init || !cur_pts || !status))) return ctx_fail(D2FE_ERR_INVALID, "null argument");
  if (npairs < 0 || npairs > 4096 || n_total < 0 || win < 3 || win > 24 || !(win & 1) || iters < 1) return ctx_fail(D2FE_ERR_INVALID, "bad LK parameters (win odd, 3..23)");
  if (n_total == 0 || npairs == 0) return D2FE_OK;
  // host staging block: [pairs table | pair index per point | prev_pts | cur_init]  -> one H2D; [cur_pts | status] <- one D2H
  const size_t n = (size_t)n_total;
  const size_t o_pairs = 0, o_idx = sizeof(LkPairDev) * (size_t)npairs, o_prev = o_idx + sizeof(int) * n, o_init = o_prev + sizeof(float) * 2 * n;
  const size_t in_bytes = o_init + sizeof(float) * 2 * n;
  const size_t o_cur = (in_bytes + 15) / 16 * 16, o_st = o_cur + sizeof(float) * 2 * n, total = o_st + n;
  std::vector<char> stage(in_bytes);
  LkPairDev* tp = reinterpret_cast<LkPairDev*>(stage.data() + o_pairs);
  int* ti = reinterpret_cast<int*>(stage.data() + o_idx);
  std::vector<char> seen(n, 0);
  for (int p = 0; p < npairs; ++p) {
    const d2fe_lk_pair& q = pairs[p];
    if (!q.prev || !q.cur) return ctx_fail(D2FE_ERR_INVALID, "null frame in pair");
    if (q.prev->w != q.cur->w || q.prev->hgt != q.cur->hgt || q.prev->levels != q.cur->levels) return ctx_fail(D2FE_ERR_INVALID, "pyramid geometry mismatch");
    if (q.type < 0 || q.type > 2 || q.first < 0 || q.count < 0 || (size_t)q.first + (size_t)q.count > n) return ctx_fail(D2FE_ERR_INVALID, "bad pair (type / point range)");
    LkPairDev& d = tp[p];
    memset(&d, 0, sizeof(d));
    d.prev = q.prev->pyr; d.cur = q.cur->pyr;
    for (int l = 0; l < 8; ++l) { d.off[l] = q.prev->off[l]; d.ws[l] = q.prev->ws[l]; d.hs[l] = q.prev->hs[l]; }
    d.levels = q.prev->levels; d.w = q.prev->w; d.h = q.prev->hgt; d.type = q.type; d.move_cols = q.move_cols;
    for (int i = q.first; i < q.first + q.count; ++i) { ti[i] = p; seen[i] = 1; }
  }
  for (size_t i = 0; i < n; ++i) if (!seen[i]) return ctx_fail(D2FE_ERR_INVALID, "point not covered by any pair");
  memcpy(stage.data() + o_prev, prev_pts, sizeof(float) * 2 * n);
  memcpy(stage.data() + o_init, cur_init, sizeof(float) * 2 * n);
  LK_TRY(hipSetDevice(ctx_device(h)));
  hipStream_t s = ctx_stream(h);
  void* raw = nullptr;
  { const int rc0 = ctx_scratch(h, total + 16, &raw); if (rc0 != D2FE_OK) return rc0; }
  char* dev = static_cast<char*>(raw);
  LkArgs a;
  a.pairs = reinterpret_cast<const LkPairDev*>(dev + o_pairs); a.pair_of = reinterpret_cast<const int*>(dev + o_idx);
  a.n = n_total; a.win = win; a.iters = iters;
  a.prev_pts = reinterpret_cast<const float*>(dev + o_prev); a.cur_init = reinterpret_cast<const float*>(dev + o_init);
  a.cur_pts = reinterpret_cast<float*>(dev + o_cur); a.status = reinterpret_cast<uint8_t*>(dev + o_st);
  std::vector<char> back(total - o_cur);
  int rc = D2FE_OK;
  auto chk = [&](hipError_t e, const char* w) { if (e != hipSuccess && rc == D2FE_OK) rc = ctx_fail(D2FE_ERR_HIP, std::string(w) + ": " + hipGetErrorString(e)); };
  chk(hipMemcpyAsync(dev, stage.data(), in_bytes, hipMemcpyHostToDevice, s), "H2D");
  if (rc == D2FE_OK) {
    hipLaunchKernelGGL(lk_track_kernel, dim3((n_total + 3) / 4), dim3(256), 0, s, a);
    chk(hipGetLastError(), "lk_track_kernel");
  }
  chk(hipMemcpyAsync(back.data(), dev + o_cur, back.size(), hipMemcpyDeviceToHost, s), "D2H");
  chk(hipStreamSynchronize(s), "sync");
  if (rc == D2FE_OK) {
    memcpy(cur_pts, back.data(), sizeof(float) * 2 * n);
    memcpy(status, back.data() + (o_st - o_cur), n);
  }
  return rc;
}

int d2fe_lk_track(d2fe_handle h, d2fe_lk_frame prev, d2fe_lk_frame cur, const float* prev_pts, const float* cur_init, int n,
                  int type, float move_cols, int win, int iters, float* cur_pts, uint8_t* status) {
  if (!prev || !cur) return ctx_fail(D2FE_ERR_INVALID, "null argument");
  d2fe_lk_pair p;
  p.prev = prev; p.cur = cur; p.first = 0; p.count = n; p.type = type; p.move_cols = move_cols;
  return d2fe_lk_track_batch(h, &p, 1, prev_pts, cur_init, n, win, iters, cur_pts, status);
}

int d2fe_detect_fast_by_region(d2fe_handle h, d2fe_lk_frame f, int features, int cols, int rows, int threshold, float* pts_xy,
                               int32_t* response, int cap, int* n_out) {
  if (n_out) *n_out = 0;
  if (!h || !f || !pts_xy || !n_out) return ctx_fail(D2FE_ERR_INVALID, "null argument");
  if (features < 1 || cols < 1 || rows < 1 || cols * rows > 4096 || threshold < 0 || threshold > 254 || cap < 1) return ctx_fail(D2FE_ERR_INVALID, "bad FAST parameters");
  const int w = f->w, hh = f->hgt, sw = w / cols, sh = hh / rows;
  LK_TRY(hipSetDevice(ctx_device(h)));
  hipStream_t s = ctx_stream(h);
  const size_t ocap = (size_t)cols * rows * features;
  const size_t sbytes = (sizeof(int) * (size_t)w * hh + 15) / 16 * 16;   // keeps the int4 list 16-byte aligned
  void* raw = nullptr;
  { const int rc0 = ctx_scratch(h, sbytes + sizeof(int4) * ocap + 16, &raw); if (rc0 != D2FE_OK) return rc0; }
  char* buf = static_cast<char*>(raw);
  int* d_score = reinterpret_cast<int*>(buf);
  int4* d_out = reinterpret_cast<int4*>(buf + sbytes);
  int* d_cnt = reinterpret_cast<int*>(buf + sbytes + sizeof(int4) * ocap);
  int rc = D2FE_OK;
  auto chk = [&](hipError_t e, const char* wh) { if (e != hipSuccess && rc == D2FE_OK) rc = ctx_fail(D2FE_ERR_HIP, std::string(wh) + ": " + hipGetErrorString(e)); };
  chk(hipMemsetAsync(buf, 0, sbytes, s), "memset score");
  chk(hipMemsetAsync(d_cnt, 0, sizeof(int), s), "memset count");
  int cnt = 0;
  std::vector<int4> host;
  if (rc == D2FE_OK && sw > 6 && sh > 6) {
    hipLaunchKernelGGL(fast_region_kernel, dim3(cols * rows), dim3(1024), 0, s, f->pyr, w, sw, sh, rows, features, threshold, d_score, w);
    hipLaunchKernelGGL(fast_nonmax_kernel, dim3((w + 63) / 64, (hh + 3) / 4), dim3(256), 0, s, d_score, w, hh, sw, sh, cols, rows, d_out, (int)ocap, d_cnt);
    chk(hipGetLastError(), "fast kernels");
    chk(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, s), "D2H count");
    chk(hipStreamSynchronize(s), "sync");
    if (rc == D2FE_OK && cnt > 0) {
      cnt = std::min<int>(cnt, (int)ocap);
      host.resize(cnt);
      chk(hipMemcpy(host.data(), d_out, sizeof(int4) * (size_t)cnt, hipMemcpyDeviceToHost), "D2H keypoints");
    }
  }
  if (rc != D2FE_OK) return rc;
  // host part of detectFastByRegion (:476-492): sort by response, keep the top `features`
  std::sort(host.begin(), host.end(), [](const int4& p, const int4& q) { return p.z != q.z ? p.z > q.z : p.w < q.w; });
  int n = 0;
  for (const int4& k : host) {
    if (n >= features) break;
    if (n >= cap) { *n_out = n; return ctx_fail(D2FE_ERR_TRUNCATED, "output capacity too small"); }
    pts_xy[2 * n] = (float)k.x; pts_xy[2 * n + 1] = (float)k.y;
    if (response) response[n] = k.z;
    ++n;
  }
  *n_out = n;
  return D2FE_OK;
}

int d2fe_good_features_to_track(d2fe_handle h, d2fe_lk_frame f, int max_corners, double quality, double min_dist, float* pts_xy,
                                int cap, int* n_out) {
  if (n_out) *n_out = 0;
  if (!h || !f || !pts_xy || !n_out) return ctx_fail(D2FE_ERR_INVALID, "null argument");
  if (!(quality > 0) || cap < 1) return ctx_fail(D2FE_ERR_INVALID, "bad parameters");
  const int w = f->w, hh = f->hgt;
  LK_TRY(hipSetDevice(ctx_device(h)));
  hipStream_t s = ctx_stream(h);
  const size_t npix = (size_t)w * hh, np = (npix + 3) / 4 * 4, ccap = npix / 2 + 1024;   // np: plane stride, keeps the 64-bit list aligned
  void* raw = nullptr;
  { const int rc0 = ctx_scratch(h, sizeof(float) * 3 * np + sizeof(unsigned long long) * ccap + 16, &raw); if (rc0 != D2FE_OK) return rc0; }
  char* buf = static_cast<char*>(raw);
  float* d_dx = reinterpret_cast<float*>(buf); float* d_dy = d_dx + np; float* d_eig = d_dy + np;
  unsigned long long* d_c = reinterpret_cast<unsigned long long*>(d_eig + np);
  unsigned* d_max = reinterpret_cast<unsigned*>(d_c + ccap);
  int* d_cnt = reinterpret_cast<int*>(d_max + 1);
  int rc = D2FE_OK;
  auto chk = [&](hipError_t e, const char* wh) { if (e != hipSuccess && rc == D2FE_OK) rc = ctx_fail(D2FE_ERR_HIP, std::string(wh) + ": " + hipGetErrorString(e)); };
  chk(hipMemsetAsync(d_max, 0, 8, s), "memset");
  const double scd = 1.0 / ((double)(1 << 2) * 3.0 * 255.0);
  const dim3 grid((w + 63) / 64, (hh + 3) / 4);
  int cnt = 0;
  std::vector<unsigned long long> keys;
  if (rc == D2FE_OK) {
    hipLaunchKernelGGL(sobel_kernel, grid, dim3(256), 0, s, f->pyr, w, w, hh, (float)scd, (float)(2.0 * scd), d_dx, d_dy);
    hipLaunchKernelGGL(min_eigen_kernel, grid, dim3(256), 0, s, d_dx, d_dy, w, hh, d_eig, d_max);
    hipLaunchKernelGGL(corners_kernel, grid, dim3(256), 0, s, d_eig, w, hh, d_max, quality, d_c, (int)ccap, d_cnt);
    chk(hipGetLastError(), "good-features kernels");
    chk(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, s), "D2H count");
    chk(hipStreamSynchronize(s), "sync");
    if (rc == D2FE_OK && cnt > (int)ccap) rc = ctx_fail(D2FE_ERR_TRUNCATED, "corner candidate buffer overflow");
    if (rc == D2FE_OK && cnt > 0) {
      keys.resize(cnt);
      chk(hipMemcpy(keys.data(), d_c, sizeof(unsigned long long) * (size_t)cnt, hipMemcpyDeviceToHost), "D2H corners");
    }
  }
  if (rc != D2FE_OK) return rc;
  std::sort(keys.begin(), keys.end(), [](unsigned long long p, unsigned long long q) { return p > q; });
  // host min-distance filter of cv::goodFeaturesToTrack / cv::cuda::GoodFeaturesToTrackDetector (grid of cell = round(minDistance))
  int n = 0;
  auto emit = [&](int x, int y) -> bool {
    if (n >= cap) return false;
    pts_xy[2 * n] = (float)x; pts_xy[2 * n + 1] = (float)y;
    ++n;
    return true;
  };
  if (min_dist >= 1) {
    const int cell = (int)std::lrint(min_dist);
    const int gw = (w + cell - 1) / cell, gh = (hh + cell - 1) / cell;
    std::vector<std::vector<int>> grid2((size_t)gw * gh);
    const double md2 = min_dist * min_dist;
    for (unsigned long long k : keys) {
      const int idx = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
      const int y = idx / w, x = idx % w;
      const int xc = x / cell, yc = y / cell;
      const int x1 = std::max(xc - 1, 0), y1 = std::max(yc - 1, 0), x2 = std::min(xc + 1, gw - 1), y2 = std::min(yc + 1, gh - 1);
      bool good = true;
      for (int yy = y1; yy <= y2 && good; ++yy)
        for (int xx = x1; xx <= x2 && good; ++xx)
          for (int e : grid2[(size_t)yy * gw + xx]) {
            const float ddx = (float)x - (float)(e % w), ddy = (float)y - (float)(e / w);
            if ((double)(ddx * ddx + ddy * ddy) < md2) { good = false; break; }
          }
      if (good) {
        grid2[(size_t)yc * gw + xc].push_back(idx);
        if (!emit(x, y)) { *n_out = n; return ctx_fail(D2FE_ERR_TRUNCATED, "output capacity too small"); }
        if (max_corners > 0 && n == max_corners) break;
      }
    }
  } else {
    for (unsigned long long k : keys) {
      if (max_corners > 0 && n >= max_corners) break;
      const int idx = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
      if (!emit(idx % w, idx / w)) { *n_out = n; return ctx_fail(D2FE_ERR_TRUNCATED, "output capacity too small"); }
    }
  }
  *n_out = n;
  return D2FE_OK;
}

}  // extern "C"
