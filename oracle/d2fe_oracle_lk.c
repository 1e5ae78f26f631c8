/*
 * d2fe_oracle_lk.c -- CPU restatement ("oracle") of SURVEY.md section 8(f)-4: the LK optical-flow tracker of
 * d2frontend (opticalflowTrackPyr d2frontend/src/opticaltrack_utils.cpp:173-279, detectPoints :375-442,
 * detectFastByRegion :444-493, buildImagePyramid :508-542).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as d2fe_oracle.c).  PARITY STATUS: **parity unpinned**, twice:
 * the reference has no fixtures for this path, and the arithmetic lives in a dependency that is not in /root/reference:
 * OpenCV 4.10.0 CUDA modules (docker/Dockerfile.x86:6,114-117) -- cv::cuda::pyrDown, cv::cuda::SparsePyrLKOpticalFlow,
 * cv::cuda::FastFeatureDetector, cv::cuda::createGoodFeaturesToTrackDetector.  What follows restates the published
 * algorithms of those classes as called by the reference (window 21x21, PYR_LEVEL 2, 30 iterations, useInitialFlow;
 * FAST-9/16 threshold 10 with non-max suppression; min-eigenvalue corners blockSize 3, Sobel 3, quality 0.01) and
 * FIXES every evaluation order the CUDA sources leave to the hardware, so that the HIP kernels can be compared bitwise:
 *   - pyrDown: 5x5 [1 4 6 4 1]/16 separable, BORDER_REFLECT_101, exact integer sum / 256 rounded half-to-even
 *     (what the CUDA kernel's exact float arithmetic + saturate_cast does; the CPU cv::pyrDown rounds half up);
 *   - LK: one 64-thread block (8x8) per point, thread (tx,ty) owns window pixels (tx+8j, ty+8i); per-thread sums in
 *     (i,j) order, then the shared-memory tree v[t] += v[t+s], s = 32..1; texture reads = fp32 bilinear interpolation
 *     of u8/255 with clamp-to-edge (a CUDA texture unit interpolates with 1.8 fixed-point weights -- not reproducible
 *     and not reproduced); no fused multiply-add anywhere;
 *   - FAST: candidates in raster order per region, sort by (response desc, region order, raster) -- the reference's
 *     std::sort is unstable and the CUDA detector's candidate order is atomics-dependent;
 *   - good features: Sobel as separable [-1 0 1] x scale*[1 2 1] in fp32 (row pass first), corners sorted by
 *     (eigenvalue desc, raster index asc), min-distance grid filter of cv::goodFeaturesToTrack on the host.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

/* ---- pyramid ------------------------------------------------------------------------------------------------------- */
/* packed pyramid of one image: level l is a tight (w_l x h_l) u8 image at byte offset off[l]; w_{l+1} = (w_l+1)/2 */
ORC_API int orc_pyr_layout(int w, int h, int levels, int* off, int* ws, int* hs) {
  int o = 0;
  for (int l = 0; l <= levels; ++l) {
    off[l] = o; ws[l] = w; hs[l] = h;
    o += w * h;
    w = (w + 1) / 2; h = (h + 1) / 2;
  }
  return o;
}

/* cv::cuda::pyrDown on CV_8UC1 (buildImagePyramid, opticaltrack_utils.cpp:526-542) */
ORC_API void orc_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) {
  static const int k[5] = {1, 4, 6, 4, 1};
  const int dw = (w + 1) / 2, dh = (h + 1) / 2;
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      int s = 0;
      for (int i = -2; i <= 2; ++i) {
        const int yy = reflect101(2 * y + i, h);
        int r = 0;
        for (int j = -2; j <= 2; ++j) r += k[j + 2] * src[(size_t)yy * w + reflect101(2 * x + j, w)];
        s += k[i + 2] * r;
      }
      int q = s >> 8;
      const int rem = s & 255;
      if (rem > 128 || (rem == 128 && (q & 1))) ++q;   /* round half to even */
      dst[(size_t)y * dw + x] = (uint8_t)(q > 255 ? 255 : q);
    }
}

ORC_API void orc_pyr_build(const uint8_t* img, int w, int h, int stride, int levels, uint8_t* pyr) {
  int off[16], ws[16], hs[16];
  orc_pyr_layout(w, h, levels, off, ws, hs);
  for (int y = 0; y < h; ++y) memcpy(pyr + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
  for (int l = 1; l <= levels; ++l) orc_pyr_down(pyr + off[l - 1], ws[l - 1], hs[l - 1], pyr + off[l]);
}

/* ---- sparse pyramidal LK --------------------------------------------------------------------------------------------- */
/* tex2D(x, y), linear filter, clamp addressing, normalised-float read mode: sample centre at (x-0.5, y-0.5) */
static float tex(const uint8_t* im, int w, int h, float x, float y) {
  const float xs = x - 0.5f, ys = y - 0.5f;
  const float xf = floorf(xs), yf = floorf(ys);
  const float fx = xs - xf, fy = ys - yf;
  int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
  x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
  y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
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

static float tree64(float* v) {
  for (int s = 32; s > 0; s >>= 1)
    for (int t = 0; t < s; ++t) v[t] = v[t] + v[t + s];
  return v[0];
}

/* one level of cv::cuda::SparsePyrLKOpticalFlow for one point (the `sparseKernel` of OpenCV's pyrlk.cu): updates *np and
 * *status exactly where the CUDA kernel writes nextPts[i] / status[i] */
static void lk_level(const uint8_t* I, const uint8_t* J, int cols, int rows, int level, int win, int iters, float ppx, float ppy,
                     float* npx, float* npy, uint8_t* status) {
  const float half = (float)((win - 1) / 2);
  float px = ppx * (1.0f / (float)(1 << level)), py = ppy * (1.0f / (float)(1 << level));
  if (px < 0 || px >= (float)cols || py < 0 || py >= (float)rows) {
    if (level == 0) *status = 0;
    return;
  }
  px -= half; py -= half;
  float Ip[64][3][3], Dx[64][3][3], Dy[64][3][3];
  float a11[64], a12[64], a22[64];
  for (int t = 0; t < 64; ++t) {
    const int tx = t & 7, ty = t >> 3;
    float s11 = 0.f, s12 = 0.f, s22 = 0.f;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const int xb = tx + 8 * j, yb = ty + 8 * i;
        if (xb >= win || yb >= win) continue;
        const float x = px + (float)xb + 0.5f, y = py + (float)yb + 0.5f;
        Ip[t][i][j] = tex(I, cols, rows, x, y);
        float dx = 3.0f * tex(I, cols, rows, x + 1, y - 1);
        dx = dx + 10.0f * tex(I, cols, rows, x + 1, y);
        dx = dx + 3.0f * tex(I, cols, rows, x + 1, y + 1);
        float mx = 3.0f * tex(I, cols, rows, x - 1, y - 1);
        mx = mx + 10.0f * tex(I, cols, rows, x - 1, y);
        mx = mx + 3.0f * tex(I, cols, rows, x - 1, y + 1);
        dx = dx - mx;
        float dy = 3.0f * tex(I, cols, rows, x - 1, y + 1);
        dy = dy + 10.0f * tex(I, cols, rows, x, y + 1);
        dy = dy + 3.0f * tex(I, cols, rows, x + 1, y + 1);
        float my = 3.0f * tex(I, cols, rows, x - 1, y - 1);
        my = my + 10.0f * tex(I, cols, rows, x, y - 1);
        my = my + 3.0f * tex(I, cols, rows, x + 1, y - 1);
        dy = dy - my;
        Dx[t][i][j] = dx; Dy[t][i][j] = dy;
        s11 = s11 + dx * dx; s12 = s12 + dx * dy; s22 = s22 + dy * dy;
      }
    a11[t] = s11; a12[t] = s12; a22[t] = s22;
  }
  float A11 = tree64(a11), A12 = tree64(a12), A22 = tree64(a22);
  float D = A11 * A22 - A12 * A12;
  if (D < 1.1920928955078125e-07f) {   /* numeric_limits<float>::epsilon() */
    if (level == 0) *status = 0;
    return;
  }
  D = 1.0f / D;
  A11 = A11 * D; A12 = A12 * D; A22 = A22 * D;
  float nx = *npx * 2.0f, ny = *npy * 2.0f;
  nx -= half; ny -= half;
  for (int k = 0; k < iters; ++k) {
    if (nx < -half || nx >= (float)cols || ny < -half || ny >= (float)rows) {
      if (level == 0) *status = 0;
      return;
    }
    float b1[64], b2[64];
    for (int t = 0; t < 64; ++t) {
      const int tx = t & 7, ty = t >> 3;
      float s1 = 0.f, s2 = 0.f;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          const int xb = tx + 8 * j, yb = ty + 8 * i;
          if (xb >= win || yb >= win) continue;
          const float Jv = tex(J, cols, rows, nx + (float)xb + 0.5f, ny + (float)yb + 0.5f);
          const float diff = (Jv - Ip[t][i][j]) * 32.0f;
          s1 = s1 + diff * Dx[t][i][j];
          s2 = s2 + diff * Dy[t][i][j];
        }
      b1[t] = s1; b2[t] = s2;
    }
    const float B1 = tree64(b1), B2 = tree64(b2);
    const float ddx = A12 * B2 - A22 * B1;
    const float ddy = A12 * B1 - A11 * B2;
    nx = nx + ddx; ny = ny + ddy;
    if (fabsf(ddx) < 0.01f && fabsf(ddy) < 0.01f) break;
  }
  *npx = nx + half; *npy = ny + half;
}

/* SparsePyrLKOpticalFlow::calc(prevPyr, nextPyr, prevPts, nextPts, status) with useInitialFlow = true */
static void lk_calc(const uint8_t* Ipyr, const uint8_t* Jpyr, const int* off, const int* ws, const int* hs, int levels, int win,
                    int iters, float ppx, float ppy, float* npx, float* npy, uint8_t* status) {
  const float sc = (float)(1.0 / (double)(1 << levels) / 2.0);
  *npx = *npx * sc; *npy = *npy * sc;
  *status = 1;
  for (int l = levels; l >= 0; --l)
    lk_level(Ipyr + off[l], Jpyr + off[l], ws[l], hs[l], l, win, iters, ppx, ppy, npx, npy, status);
}

/* cv::cuda::SparsePyrLKOpticalFlow(win, levels, iters, useInitialFlow = true)->calc(prevPyr, nextPyr, prevPts, nextPts, status): ONE direction,
 * next_pts holds the initial flow on entry.  Exposed for oracle/ref_shim/spref_api4.cpp, where the reference's own opticalflowTrackPyr calls it
 * through a stand-in cv::cuda class. */
ORC_API void orc_lk_calc(const uint8_t* prev_pyr, const uint8_t* next_pyr, int w, int h, int levels, const float* prev_pts, float* next_pts, int n,
                         int win, int iters, uint8_t* status) {
  int off[16], ws[16], hs[16];
  orc_pyr_layout(w, h, levels, off, ws, hs);
  for (int i = 0; i < n; ++i)
    lk_calc(prev_pyr, next_pyr, off, ws, hs, levels, win, iters, prev_pts[2 * i], prev_pts[2 * i + 1], &next_pts[2 * i], &next_pts[2 * i + 1], &status[i]);
}

/* the tracking block of opticalflowTrackPyr (opticaltrack_utils.cpp:236-272): forward LK prev->cur from cur_init, reverse
 * LK cur->prev from the (shifted) forward result, accept iff both succeed, |prev - reverse| <= 0.5 and inBorder(cur).
 * type: 0 WHOLE_IMG_MATCH, 1 LEFT_RIGHT_IMG_MATCH, 2 RIGHT_LEFT_IMG_MATCH.  Outputs cur_pts[n][2], status[n]. */
ORC_API void orc_lk_track(const uint8_t* prev_pyr, const uint8_t* cur_pyr, int w, int h, int levels, const float* prev_pts,
                          const float* cur_init, int n, int type, float move_cols, int win, int iters, float* cur_pts,
                          uint8_t* status) {
  int off[16], ws[16], hs[16];
  orc_pyr_layout(w, h, levels, off, ws, hs);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n; ++i) {
    const float ppx = prev_pts[2 * i], ppy = prev_pts[2 * i + 1];
    float cx = cur_init[2 * i], cy = cur_init[2 * i + 1];
    uint8_t st = 1, rst = 1;
    lk_calc(prev_pyr, cur_pyr, off, ws, hs, levels, win, iters, ppx, ppy, &cx, &cy, &st);
    float rx = cx, ry = cy;
    if (type == 1 && st == 1) rx -= move_cols;
    if (type == 2 && st == 1) rx += move_cols;
    lk_calc(cur_pyr, prev_pyr, off, ws, hs, levels, win, iters, cx, cy, &rx, &ry, &rst);
    const float dx = ppx - rx, dy = ppy - ry;
    const double nrm = sqrt((double)dx * dx + (double)dy * dy);   /* cv::norm(Point2f) */
    uint8_t ok = (st && rst && nrm <= 0.5) ? 1 : 0;
    if (ok) {   /* inBorder, :35-41: cvRound = round half to even */
      const int ix = (int)lrint((double)cx), iy = (int)lrint((double)cy);
      if (!(1 <= ix && ix < w - 1 && 1 <= iy && iy < h - 1)) ok = 0;
    }
    cur_pts[2 * i] = cx; cur_pts[2 * i + 1] = cy;
    status[i] = ok;
  }
}

/* ---- FAST-9/16 by region ------------------------------------------------------------------------------------------------- */
static const int FAST_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int FAST_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_is_corner(const uint8_t* p, int stride, int t) {
  const int v = p[0];
  int br = 0, dk = 0;   /* bit masks over the circle */
  for (int k = 0; k < 16; ++k) {
    const int q = p[FAST_DY[k] * stride + FAST_DX[k]];
    if (q > v + t) br |= 1 << k;
    if (q < v - t) dk |= 1 << k;
  }
  for (int s = 0; s < 16; ++s) {
    int m = 0;
    for (int k = 0; k < 9; ++k) m |= 1 << ((s + k) & 15);
    if ((br & m) == m || (dk & m) == m) return 1;
  }
  return 0;
}

/* cornerScore of OpenCV's CUDA FAST: the largest threshold for which the pixel is still a corner (binary search) */
static int fast_score(const uint8_t* p, int stride, int threshold) {
  int lo = threshold + 1, hi = 255;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (fast_is_corner(p, stride, mid)) lo = mid + 1; else hi = mid - 1;
  }
  return lo - 1;
}

typedef struct { float x, y; int resp; int order; } fast_kp;
static int fast_cmp(const void* a, const void* b) {
  const fast_kp* p = (const fast_kp*)a; const fast_kp* q = (const fast_kp*)b;
  if (p->resp != q->resp) return p->resp > q->resp ? -1 : 1;
  return p->order < q->order ? -1 : (p->order > q->order ? 1 : 0);
}

/* detectFastByRegion(img, mask = empty, features, cols, rows) (opticaltrack_utils.cpp:444-493): FAST(10, nonmax, 9_16) on
 * every (img.cols/cols) x (img.rows/rows) region, regions in (i over cols, j over rows) order, sort by response, top `features`.
 * The CUDA detector's max_npoints (= features) caps the candidates of a region BEFORE non-max suppression: kept = the
 * first `features` in raster order (the CUDA kernel keeps an atomics-dependent subset). */
ORC_API int orc_fast_by_region(const uint8_t* img, int w, int h, int stride, int features, int cols, int rows, int threshold,
                               float* out_xy, int* out_resp, int cap) {
  const int sw = w / cols, sh = h / rows;
  fast_kp* all = (fast_kp*)malloc(sizeof(fast_kp) * (size_t)(w * h + 1));
  int* score = (int*)malloc(sizeof(int) * (size_t)(sw * sh + 1));
  int nall = 0;
  for (int i = 0; i < cols; ++i)
    for (int j = 0; j < rows; ++j) {
      const uint8_t* roi = img + (size_t)(sh * j) * stride + sw * i;
      memset(score, 0, sizeof(int) * (size_t)(sw * sh + 1));
      int ncand = 0;
      for (int y = 3; y < sh - 3; ++y)
        for (int x = 3; x < sw - 3; ++x) {
          const uint8_t* p = roi + (size_t)y * stride + x;
          if (ncand < features && fast_is_corner(p, stride, threshold)) {
            score[y * sw + x] = fast_score(p, stride, threshold);
            ++ncand;
          }
        }
      for (int y = 3; y < sh - 3; ++y)
        for (int x = 3; x < sw - 3; ++x) {
          const int s = score[y * sw + x];
          if (!s) continue;
          if (s > score[(y - 1) * sw + x - 1] && s > score[(y - 1) * sw + x] && s > score[(y - 1) * sw + x + 1] &&
              s > score[y * sw + x - 1] && s > score[y * sw + x + 1] && s > score[(y + 1) * sw + x - 1] &&
              s > score[(y + 1) * sw + x] && s > score[(y + 1) * sw + x + 1]) {
            all[nall].x = (float)(x + sw * i); all[nall].y = (float)(y + sh * j); all[nall].resp = s; all[nall].order = nall;
            ++nall;
          }
        }
    }
  qsort(all, (size_t)nall, sizeof(fast_kp), fast_cmp);
  int n = 0;
  for (int k = 0; k < nall && n < features && n < cap; ++k, ++n) {
    out_xy[2 * n] = all[k].x; out_xy[2 * n + 1] = all[k].y;
    if (out_resp) out_resp[n] = all[k].resp;
  }
  free(all); free(score);
  return n;
}

/* ---- good features to track ------------------------------------------------------------------------------------------------ */
/* cv::cuda::cornerMinEigenVal(u8, blockSize 3, ksize 3, BORDER_REFLECT101) */
ORC_API void orc_min_eigen(const uint8_t* img, int w, int h, int stride, float* eig) {
  const double scd = 1.0 / ((double)(1 << 2) * 3.0 * 255.0);
  const float k0 = (float)scd, k1 = (float)(2.0 * scd);
  float* dx = (float*)malloc(sizeof(float) * (size_t)w * h);
  float* dy = (float*)malloc(sizeof(float) * (size_t)w * h);
  float* t0 = (float*)malloc(sizeof(float) * (size_t)w * h);
  float* t1 = (float*)malloc(sizeof(float) * (size_t)w * h);
  /* row pass: t0 = [-1 0 1] (for Dx), t1 = [1 2 1] (for Dy), exact small integers */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float a = (float)img[(size_t)y * stride + reflect101(x - 1, w)], b = (float)img[(size_t)y * stride + x],
                  c = (float)img[(size_t)y * stride + reflect101(x + 1, w)];
      t0[(size_t)y * w + x] = c - a;
      t1[(size_t)y * w + x] = (a + 2.0f * b) + c;
    }
  /* column pass: Dx = scale*[1 2 1], Dy = scale*[-1 0 1] */
  for (int y = 0; y < h; ++y) {
    const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    for (int x = 0; x < w; ++x) {
      float s = k0 * t0[(size_t)ym * w + x];
      s = s + k1 * t0[(size_t)y * w + x];
      s = s + k0 * t0[(size_t)yp * w + x];
      dx[(size_t)y * w + x] = s;
      dy[(size_t)y * w + x] = k0 * t1[(size_t)yp * w + x] - k0 * t1[(size_t)ym * w + x];
    }
  }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float a = 0.f, b = 0.f, c = 0.f;
      for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j) {
          const size_t o = (size_t)reflect101(y + i, h) * w + reflect101(x + j, w);
          const float gx = dx[o], gy = dy[o];
          a = a + gx * gx; b = b + gx * gy; c = c + gy * gy;
        }
      a = a * 0.5f; c = c * 0.5f;
      eig[(size_t)y * w + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
    }
  free(dx); free(dy); free(t0); free(t1);
}

typedef struct { float v; int idx; } eig_c;
static int eig_cmp(const void* a, const void* b) {
  const eig_c* p = (const eig_c*)a; const eig_c* q = (const eig_c*)b;
  if (p->v != q->v) return p->v > q->v ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}

/* cv::cuda::GoodFeaturesToTrackDetector::detect (detectPoints, opticaltrack_utils.cpp:404-412): corners = interior pixels
 * with eig > quality*max and eig == max of their 3x3 neighbourhood, sorted by eig, then the host min-distance grid filter */
ORC_API int orc_good_features(const uint8_t* img, int w, int h, int stride, int max_corners, double quality, double min_dist,
                              float* out_xy, int cap) {
  float* eig = (float*)malloc(sizeof(float) * (size_t)w * h);
  orc_min_eigen(img, w, h, stride, eig);
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)w * h; ++i) mx = eig[i] > mx ? eig[i] : mx;
  const float thr = (float)((double)mx * quality);
  eig_c* c = (eig_c*)malloc(sizeof(eig_c) * (size_t)w * h);
  int nc = 0;
  for (int y = 1; y < h - 1; ++y)
    for (int x = 1; x < w - 1; ++x) {
      const float v = eig[(size_t)y * w + x];
      if (!(v > thr)) continue;
      float m = v;
      for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j) m = fmaxf(m, eig[(size_t)(y + i) * w + x + j]);
      if (v == m) { c[nc].v = v; c[nc].idx = y * w + x; ++nc; }
    }
  qsort(c, (size_t)nc, sizeof(eig_c), eig_cmp);
  int n = 0;
  if (min_dist >= 1) {
    const int cell = (int)lrint(min_dist);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    int* head = (int*)malloc(sizeof(int) * (size_t)gw * gh);
    int* next = (int*)malloc(sizeof(int) * (size_t)(nc + 1));
    for (int i = 0; i < gw * gh; ++i) head[i] = -1;
    const double md2 = min_dist * min_dist;
    for (int k = 0; k < nc; ++k) {
      const int y = c[k].idx / w, x = c[k].idx % w;
      const int xc = x / cell, yc = y / cell;
      int x1 = xc - 1, y1 = yc - 1, x2 = xc + 1, y2 = yc + 1;
      x1 = x1 < 0 ? 0 : x1; y1 = y1 < 0 ? 0 : y1; x2 = x2 > gw - 1 ? gw - 1 : x2; y2 = y2 > gh - 1 ? gh - 1 : y2;
      int good = 1;
      for (int yy = y1; yy <= y2 && good; ++yy)
        for (int xx = x1; xx <= x2 && good; ++xx)
          for (int e = head[yy * gw + xx]; e >= 0; e = next[e]) {
            const float ddx = (float)x - (float)(c[e].idx % w), ddy = (float)y - (float)(c[e].idx / w);
            if ((double)(ddx * ddx + ddy * ddy) < md2) { good = 0; break; }
          }
      if (good) {
        next[k] = head[yc * gw + xc]; head[yc * gw + xc] = k;
        if (n < cap) { out_xy[2 * n] = (float)x; out_xy[2 * n + 1] = (float)y; }
        ++n;
        if (max_corners > 0 && n == max_corners) break;
      }
    }
    free(head); free(next);
  } else {
    for (int k = 0; k < nc && (max_corners <= 0 || n < max_corners); ++k, ++n)
      if (n < cap) { out_xy[2 * n] = (float)(c[k].idx % w); out_xy[2 * n + 1] = (float)(c[k].idx / w); }
  }
  free(eig); free(c);
  return n < cap ? n : cap;
}
