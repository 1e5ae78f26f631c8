// Stand-in for <opencv2/opencv.hpp> (test infrastructure, see ../README.md).  Only what the compiled reference line ranges touch:
// superpoint_common.cpp:8-40,101-177 (getKeyPoints, NMS2), feature_matcher.cpp (matchKNN), d2featuretracker.cpp:1051-1075,1146-1181.
// Semantics follow OpenCV 4.10.0 (docker/Dockerfile.x86:6).  cv::BFMatcher is a restatement of third-party arithmetic
// (modules/features2d matchers.cpp knnMatchImpl -> modules/core batch_distance.cpp BatchDistInvoker -> norm.cpp normL2Sqr_).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_16UC1 CV_16U
#define CV_32FC1 CV_32F

namespace cv {
template <class T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
  Point_ operator-(const Point_& o) const { return Point_((T)(x - o.x), (T)(y - o.y)); }
  // core/types.hpp: template<typename _Tp2> operator Point_<_Tp2>() const = Point_<_Tp2>(saturate_cast<_Tp2>(x), saturate_cast<_Tp2>(y)); float -> int
  // is cvRound (round half to even, lrintf) -- what img.at<uchar>(cv::Point2f) does with a keypoint (loop_utils.cpp:54-63, round 6)
  template <class T2> operator Point_<T2>() const { return Point_<T2>(shim_sat<T2>(x), shim_sat<T2>(y)); }
  template <class T2, class S> static T2 shim_sat(S v) { if (std::is_integral<T2>::value && std::is_floating_point<S>::value) return (T2)std::lrint(v); return (T2)v; }
};
typedef Point_<int> Point;
typedef Point_<float> Point2f;
// cv::norm(Point_<T>) = std::sqrt((double)pt.x*pt.x + (double)pt.y*pt.y)   (core/types.hpp)
template <class T> inline double norm(const Point_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }
struct Size { int width, height; Size(int w, int h) : width(w), height(h) {} };
// round 6 (spref_loopcam.cpp: LoopCam::extractorImgDescDeepnet, loop_cam.cpp:589-648, and extractColor, loop_utils.cpp:54-63)
struct Rect { int x, y, width, height; Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
struct Vec3b { unsigned char v[3]; Vec3b() : v{0, 0, 0} {} Vec3b(unsigned char a, unsigned char b, unsigned char c) : v{a, b, c} {} unsigned char operator[](int i) const { return v[i]; } };

class Mat {
 public:
  int rows = 0, cols = 0, type_ = CV_8U;
  unsigned char* data = nullptr;
  size_t step = 0;
  std::shared_ptr<std::vector<unsigned char>> own;
  static size_t esz(int t) { return t == CV_8U ? 1 : t == CV_16U ? 2 : 4; }
  Mat() {}
  Mat(int r, int c, int t) { create(r, c, t); }
  Mat(Size s, int t) { create(s.height, s.width, t); }
  Mat(int r, int c, int t, void* d) : rows(r), cols(c), type_(t), data((unsigned char*)d), step((size_t)c * esz(t)) {}   // borrowed
  void create(int r, int c, int t) {
    rows = r; cols = c; type_ = t; step = (size_t)c * esz(t);
    own = std::make_shared<std::vector<unsigned char>>((size_t)r * step + 16);
    data = own->data();
  }
  template <class T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <class T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <class T> const T* ptr(int r) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  void setTo(int v) { if (data) std::memset(data, v, (size_t)rows * step); }   // only ever called with 0
  Size size() const { return Size(cols, rows); }
  // round 6: single-channel images only (the stand-in has no channel count: CV_8UC1 frames are what the path extracts from)
  int channels() const { return 1; }
  bool empty() const { return !data || rows <= 0 || cols <= 0; }
  Mat operator()(const Rect& r) const { Mat m; m.rows = r.height; m.cols = r.width; m.type_ = type_; m.step = step; m.own = own; m.data = data + (size_t)r.y * step + (size_t)r.x * esz(type_); return m; }   // a view, as in OpenCV
  void setTo(const Scalar& s) { for (int y = 0; data && y < rows; ++y) std::memset(data + (size_t)y * step, (int)s.v[0], (size_t)cols * esz(type_)); }      // CV_8U views (the STEREO_FISHEYE mask, loop_cam.cpp:601-604)
  template <class T> const T& at(Point_<int> p) const { return *reinterpret_cast<const T*>(data + (size_t)p.y * step + (size_t)p.x * sizeof(T)); }
};
// (prob > threshold): CV_8U mask, 255 where true (cv::compare CMP_GT with a scalar, single precision)
inline Mat operator>(const Mat& m, float thr) {
  Mat r(m.rows, m.cols, CV_8U);
  for (int y = 0; y < m.rows; ++y)
    for (int x = 0; x < m.cols; ++x) r.at<unsigned char>(y, x) = m.at<float>(y, x) > thr ? 255 : 0;
  return r;
}
// cv::findNonZero: row-major scan, Point(x = column, y = row)
inline void findNonZero(const Mat& m, std::vector<Point>& out) {
  out.clear();
  for (int y = 0; y < m.rows; ++y)
    for (int x = 0; x < m.cols; ++x)
      if (m.at<unsigned char>(y, x)) out.push_back(Point(x, y));
}

struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = FLT_MAX;
  DMatch() {}
  DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
};
enum { NORM_L2 = 4 };

// modules/core/src/norm.cpp normL2Sqr_(const float*, const float*, int), x86-64 baseline build (128-bit universal intrinsics):
// four 4-lane accumulators over 16-element strides, v_muladd without FMA = mul then add, v_reduce_sum, scalar tail.
inline float shim_normL2Sqr(const float* a, const float* b, int n) {
  float acc[4][4] = {{0}};
  int j = 0;
  for (; j <= n - 16; j += 16)
    for (int v = 0; v < 4; ++v)
      for (int l = 0; l < 4; ++l) { const float t = a[j + 4 * v + l] - b[j + 4 * v + l]; const float tt = t * t; acc[v][l] = acc[v][l] + tt; }
  float r[4];
  for (int l = 0; l < 4; ++l) r[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
  float d = (r[0] + r[2]) + (r[1] + r[3]);
  for (; j < n; ++j) { const float t = a[j] - b[j]; d += t * t; }
  return d;
}

class BFMatcher {
 public:
  int normType; bool crossCheck;
  explicit BFMatcher(int nt = NORM_L2, bool cc = false) : normType(nt), crossCheck(cc) {}
  // batchDistance(query, train, dist, CV_32F, nidx, NORM_L2, K, noArray(), 0, false): per query row i, all distances
  // dist_j = std::sqrt(normL2Sqr(q_i, t_j)) into a buffer, then the K-best insertion of BatchDistInvoker (bit patterns of
  // non-negative floats compared as ints; strict '<': among equal distances the lower train index stays first; slots start at
  // FLT_MAX / index -1).  knnMatchImpl then emits the entries with index >= 0 in that order.
  void knnMatch(const Mat& q, const Mat& t, std::vector<std::vector<DMatch>>& matches, int K) const {
    matches.clear();
    matches.reserve(q.rows);
    std::vector<float> buf((size_t)std::max(t.rows, 1));
    for (int i = 0; i < q.rows; ++i) {
      std::vector<int> nidx((size_t)K, -1);
      std::vector<float> dist((size_t)K, FLT_MAX);
      for (int j = 0; j < t.rows; ++j) buf[j] = std::sqrt(shim_normL2Sqr(q.ptr<float>(i), t.ptr<float>(j), q.cols));
      for (int j = 0; j < t.rows; ++j) {
        int32_t di; std::memcpy(&di, &buf[j], 4);
        int32_t last; std::memcpy(&last, &dist[K - 1], 4);
        if (di < last) {
          int k;
          for (k = K - 2; k >= 0; --k) {
            int32_t dk; std::memcpy(&dk, &dist[k], 4);
            if (!(dk > di)) break;
            nidx[k + 1] = nidx[k]; dist[k + 1] = dist[k];
          }
          nidx[k + 1] = j; dist[k + 1] = buf[j];
        }
      }
      std::vector<DMatch> row;
      for (int kk = 0; kk < K; ++kk) if (nidx[kk] >= 0) row.push_back(DMatch(i, nidx[kk], 0, dist[kk]));
      matches.push_back(row);
    }
  }
  // match() with crossCheck = true: knnMatch(k = 1) with batchDistance's crossCheck -- a pair (i, j) survives iff j is i's
  // nearest train row and i is j's nearest query row (first minimum on ties, in both directions)
  void match(const Mat& q, const Mat& t, std::vector<DMatch>& out) const {
    out.clear();
    if (q.rows == 0 || t.rows == 0) return;
    std::vector<int> bq((size_t)q.rows, -1), bt((size_t)t.rows, -1);
    std::vector<float> dq((size_t)q.rows, FLT_MAX), dt((size_t)t.rows, FLT_MAX);
    for (int i = 0; i < q.rows; ++i)
      for (int j = 0; j < t.rows; ++j) {
        const float d = std::sqrt(shim_normL2Sqr(q.ptr<float>(i), t.ptr<float>(j), q.cols));
        if (d < dq[i]) { dq[i] = d; bq[i] = j; }
        if (d < dt[j]) { dt[j] = d; bt[j] = i; }
      }
    for (int i = 0; i < q.rows; ++i)
      if (bq[i] >= 0 && (!crossCheck || bt[bq[i]] == i)) out.push_back(DMatch(i, bq[i], 0, dq[i]));
  }
};
}  // namespace cv
