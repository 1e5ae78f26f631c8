// d2fe.hpp -- header-only C++ mirror of the D2SLAM d2frontend interfaces this library stands behind, written on top of the
// C ABI (d2fe.h).  Same names, argument meaning, ownership and error behaviour as the reference, so that a call site in
// D2SLAM compiles against either:
//   SuperPointConfig / SuperPoint::build / SuperPoint::infer   d2frontend/include/d2frontend/CNN/superpoint_tensorrt.h:17-93,
//                                                              d2frontend/src/CNN/superpoint_tensorrt.cpp:161-183
//   MobileNetVLADONNX::inference                               d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:49-74
//   matchKNN                                                   d2frontend/include/d2frontend/feature_matcher.h:6-11
//   cv::BFMatcher(NORM_L2, true).match                         loop_cam.cpp:167-170, d2featuretracker.cpp:1141-1142
//   getFeatureHalfImg                                          d2frontend/src/d2featuretracker.cpp:1051-1075
//   LKImageInfo, buildImagePyramid, opticalflowTrackPyr, detectPoints
//                                                              d2frontend/include/d2frontend/opticaltrack_utils.h:16-36,
//                                                              d2frontend/src/opticaltrack_utils.cpp:173-279,375-442
// OpenCV is not required: Point2f / DMatch / the image and descriptor views are layout-compatible stand-ins for cv::Point2f,
// cv::DMatch and the (data, rows, cols, step) fields of a cv::Mat; with -DD2FE_WITH_OPENCV the cv:: types convert implicitly.
#ifndef D2FE_HPP_
#define D2FE_HPP_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "d2fe.h"

#ifdef D2FE_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace D2FrontEnd {

struct Point2f {                      // == cv::Point2f
  float x = 0.f, y = 0.f;
  Point2f() = default;
  Point2f(float x_, float y_) : x(x_), y(y_) {}
#ifdef D2FE_WITH_OPENCV
  Point2f(const cv::Point2f& p) : x(p.x), y(p.y) {}
  operator cv::Point2f() const { return cv::Point2f(x, y); }
#endif
};
static_assert(sizeof(Point2f) == 2 * sizeof(float), "Point2f must be two packed floats");

struct DMatch {                       // == cv::DMatch
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0.f;
  DMatch() = default;
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
};

struct ImageView {                    // borrowed view of a CV_8UC1 / CV_8UC3 cv::Mat
  const uint8_t* data = nullptr;
  int rows = 0, cols = 0, channels = 1;
  size_t step = 0;
  ImageView() = default;
  ImageView(const uint8_t* d, int rows_, int cols_, size_t step_ = 0, int channels_ = 1)
      : data(d), rows(rows_), cols(cols_), channels(channels_), step(step_ ? step_ : (size_t)cols_ * channels_) {}
#ifdef D2FE_WITH_OPENCV
  ImageView(const cv::Mat& m) : data(m.data), rows(m.rows), cols(m.cols), channels(m.channels()), step(m.step) {}
#endif
  bool empty() const { return !data || rows <= 0 || cols <= 0; }
};

struct DescView {                     // borrowed view of a continuous CV_32F cv::Mat [rows x cols]
  const float* data = nullptr;
  int rows = 0, cols = 0;
  DescView() = default;
  DescView(const float* d, int rows_, int cols_) : data(d), rows(rows_), cols(cols_) {}
#ifdef D2FE_WITH_OPENCV
  DescView(const cv::Mat& m) : data(m.ptr<float>()), rows(m.rows), cols(m.cols) {}
#endif
};

// superpoint_tensorrt.h:17-33 (the fields the path uses) plus what the TensorRT engine carried implicitly
struct SuperPointConfig {
  int32_t max_keypoints = 100;           // -1: keep every keypoint above the threshold, raster order (topKeypoints with k == -1, superpoint_tensorrt.cpp:241-253)
  int32_t keep_all_capacity = 4096;      // max_keypoints == -1 only: the buffers infer() starts with; an image with more keypoints is re-run with 4x the
                                         // capacity (up to width * height) until everything fits, so the result is the reference's, whatever the count
  int32_t remove_borders = 1;
  float keypoint_threshold = 0.015f;
  int32_t input_width = 640, input_height = 480;
  int32_t dla_core = -1;                 // unused here, kept for source compatibility
  std::string onnx_path, engine_path;    // the adapter's weight loader reads these; this class takes the tensors in build()
  int32_t device_id = 0;
  int32_t max_batch = 2;
  bool fast_mode = false;                // D2FE_PREC_F16X2 instead of the bit-exact fp32 mode
  bool winograd = false;                 // D2FE_PREC_F32_WINO: fp32, 3x3 layers as Winograd F(2x2,3x3) (1.76x the direct mode's throughput)
};

class SuperPoint {
 public:
  explicit SuperPoint(const SuperPointConfig& cfg) : cfg_(cfg) {}
  ~SuperPoint() { if (h_) d2fe_destroy(h_); }
  SuperPoint(const SuperPoint&) = delete;
  SuperPoint& operator=(const SuperPoint&) = delete;

  // reference: build() parses the ONNX / deserialises the engine (superpoint_tensorrt.cpp:22-107); here the 12 conv layers
  // are passed in (d2slam_amd/weights.py reads .pth / .npz / .onnx; any loader that fills d2fe_superpoint_weights works)
  bool build(const d2fe_superpoint_weights& w) {
    d2fe_config c;
    d2fe_default_config(&c);
    c.device_id = cfg_.device_id;
    c.max_width = cfg_.input_width; c.max_height = cfg_.input_height; c.max_batch = cfg_.max_batch;
    c.max_keypoints = cfg_.max_keypoints; c.remove_borders = cfg_.remove_borders; c.keypoint_threshold = cfg_.keypoint_threshold;
    c.postproc = D2FE_POSTPROC_B;
    c.precision = cfg_.fast_mode ? D2FE_PREC_F16X2 : (cfg_.winograd ? D2FE_PREC_F32_WINO : D2FE_PREC_F32);
    if (d2fe_create(&c, &h_) != D2FE_OK) { report("d2fe_create"); h_ = nullptr; return false; }
    if (d2fe_load_superpoint(h_, &w) != D2FE_OK) { report("d2fe_load_superpoint"); return false; }
    return true;
  }

  // superpoint_tensorrt.cpp:161-183: keypoints are APPENDED (not cleared) on success, all three outputs cleared on failure
  bool infer(const ImageView& input, std::vector<Point2f>& keypoints, std::vector<float>& local_descriptors,
             std::vector<float>& scores) {
    // never throws (the reference's contract is "return false", :164-170): sizes come from the configuration, not from max_keypoints == -1
    const long most = input.empty() ? 1 : (long)input.rows * input.cols;
    long cap = cfg_.max_keypoints > 0 ? cfg_.max_keypoints : (cfg_.keep_all_capacity > 0 ? cfg_.keep_all_capacity : 4096);
    if (cap > most) cap = most;
    int n = 0, rc = D2FE_ERR_INVALID;
    for (;;) {
      if (!reserve(cap)) break;
      rc = (!h_ || input.channels != 1) ? (int)D2FE_ERR_INVALID
                                        : d2fe_superpoint_extract(h_, input.data, input.cols, input.rows, (int)input.step, kp_.data(), sc_.data(), de_.data(), (int)cap, &n);
      // keep-all with more keypoints than the buffers hold: the library has written the strongest `cap`; the reference returns ALL of them, so run again with room
      if (rc != D2FE_ERR_TRUNCATED || cfg_.max_keypoints > 0 || cap >= most) break;
      cap = cap * 4 < most ? cap * 4 : most;
    }
    if (rc != D2FE_OK) {
      keypoints.clear(); local_descriptors.clear(); scores.clear();
      report("superpoint infer failed");
      return false;
    }
    for (int i = 0; i < n; ++i) keypoints.emplace_back(kp_[2 * i], kp_[2 * i + 1]);
    local_descriptors.insert(local_descriptors.end(), de_.begin(), de_.begin() + (size_t)n * d2fe_desc_dim(h_));
    scores.assign(sc_.begin(), sc_.begin() + n);
    return true;
  }

  d2fe_handle handle() const { return h_; }

 private:
  static void report(const char* what) { std::fprintf(stderr, "[d2fe] %s: %s\n", what, d2fe_last_error()); }
  bool reserve(long cap) {               // grow-only staging for `cap` keypoints; false instead of an exception when memory is short
    try {
      if (kp_.size() < 2 * (size_t)cap) kp_.resize(2 * (size_t)cap);
      if (sc_.size() < (size_t)cap) sc_.resize((size_t)cap);
      if (de_.size() < (size_t)cap * 256) de_.resize((size_t)cap * 256);
    } catch (...) { return false; }
    return true;
  }
  SuperPointConfig cfg_;
  d2fe_handle h_ = nullptr;
  std::vector<float> kp_, sc_, de_;
};

// MobileNetVLADONNX (mobilenetvlad_onnx.h:18-74) on an existing handle: colour conversion / resize / PCA as in inference()
class MobileNetVLAD {
 public:
  MobileNetVLAD(d2fe_handle h, int width, int height) : h_(h), width_(width), height_(height) {}
  bool load(const d2fe_netvlad_weights& w) { return d2fe_load_netvlad(h_, &w) == D2FE_OK; }
  bool setPCA(const float* comp, const float* mean, int rows) { return d2fe_set_netvlad_pca(h_, comp, mean, rows) == D2FE_OK; }
  std::vector<float> inference(const ImageView& input) {
    std::vector<float> out((size_t)d2fe_netvlad_dim(h_));
    const uint8_t* img = input.data;
    int stride = (int)input.step;
    if (input.channels != 1 || input.rows != height_ || input.cols != width_) {   // :51-59 cvtColor + resize
      tmp_.resize((size_t)width_ * height_);
      if (d2fe_prepare_gray(h_, input.data, input.channels, input.cols, input.rows, (int)input.step, width_, height_, tmp_.data()) != D2FE_OK) return {};
      img = tmp_.data(); stride = width_;
    }
    if (d2fe_netvlad(h_, img, width_, height_, stride, out.data()) != D2FE_OK) return {};
    return out;
  }

 private:
  d2fe_handle h_;
  int width_, height_;
  std::vector<uint8_t> tmp_;
};

// The per-frame work of D2Frontend::processStereoframe (d2frontend.cpp:155-169: LoopCam::generateStereoImageDescriptor, loop_cam.cpp:440-470, then
// D2FeatureTracker::trackLocalFrames, d2featuretracker.cpp:403-456,658-695) with several frames in flight: d2fe_pipe_* of include/d2fe.h behind the
// containers the reference's call sites use.  submit() from the image callback, wait() from the tracker thread; results bit-identical to
// SuperPoint::infer + MobileNetVLAD::inference + matchKNN on the same frames.
struct StereoFrameResult {
  std::vector<Point2f> kps_left, kps_right;
  std::vector<float> scores_left, scores_right;
  std::vector<float> desc_left, desc_right;          // [n][256]
  std::vector<float> netvlad;                        // empty when the pipe runs without NetVLAD
  std::vector<DMatch> left_right;                    // queryIdx: left keypoint, trainIdx: right keypoint
  std::vector<DMatch> left_prev;                     // queryIdx: left keypoint, trainIdx: keypoint of the PREVIOUS left frame
};
class StereoPipe {
 public:
  // cfg: d2fe_pipe_default_config() + the fields the caller sets (lanes, width, height, cap, netvlad, coalesce, coalesce_depth, ...); frames is forced to 1
  StereoPipe(d2fe_handle h, d2fe_pipe_config cfg) {
    cfg.frames = 1;
    if (d2fe_pipe_create(h, &cfg, &p_) != D2FE_OK) { std::fprintf(stderr, "[d2fe] d2fe_pipe_create: %s\n", d2fe_last_error()); p_ = nullptr; }
  }
  ~StereoPipe() { if (p_) d2fe_pipe_destroy(p_); }
  StereoPipe(const StereoPipe&) = delete;
  StereoPipe& operator=(const StereoPipe&) = delete;
  bool ok() const { return p_ != nullptr; }
  // returns the ticket (>= 0) or -1
  int64_t submit(const ImageView& left, const ImageView& right) {
    int64_t t = -1;
    if (!p_ || left.channels != 1 || right.channels != 1 || left.step != right.step) return -1;
    if (d2fe_pipe_submit(p_, left.data, right.data, (int)left.step, 0, &t) != D2FE_OK) { std::fprintf(stderr, "[d2fe] d2fe_pipe_submit: %s\n", d2fe_last_error()); return -1; }
    return t;
  }
  bool wait(int64_t ticket, StereoFrameResult& out) {
    d2fe_pipe_result r;
    if (!p_ || d2fe_pipe_wait(p_, ticket, &r) != D2FE_OK) { std::fprintf(stderr, "[d2fe] d2fe_pipe_wait: %s\n", d2fe_last_error()); return false; }
    auto side = [&](int i, std::vector<Point2f>& k, std::vector<float>& s, std::vector<float>& d) {
      const int n = r.n_kp[i];
      k.clear(); for (int j = 0; j < n; ++j) k.emplace_back(r.kps_xy[((size_t)i * r.cap + j) * 2], r.kps_xy[((size_t)i * r.cap + j) * 2 + 1]);
      s.assign(r.scores + (size_t)i * r.cap, r.scores + (size_t)i * r.cap + n);
      d.assign(r.desc + (size_t)i * r.cap * r.desc_dim, r.desc + ((size_t)i * r.cap + n) * r.desc_dim);
    };
    side(0, out.kps_left, out.scores_left, out.desc_left);
    side(1, out.kps_right, out.scores_right, out.desc_right);
    out.netvlad.clear();
    if (r.netvlad) out.netvlad.assign(r.netvlad, r.netvlad + r.netvlad_dim);
    auto matches = [&](const int32_t* q, const int32_t* t, const float* d, const int32_t* n, std::vector<DMatch>& m) {
      m.clear();
      if (q) for (int j = 0; j < n[0]; ++j) m.emplace_back(q[j], t[j], d[j]);
    };
    matches(r.lr_q, r.lr_t, r.lr_dist, r.lr_n, out.left_right);
    matches(r.prev_q, r.prev_t, r.prev_dist, r.prev_n, out.left_prev);
    return true;
  }
  // d2fe_pipe_stream_placement: the hardware-pipe class measured for every lane's (own, second) stream; the return value = classes told apart (0: not measured)
  int streamPlacement(std::vector<std::pair<int, int>>& lanes) const {
    lanes.clear();
    if (!p_) return 0;
    const int K = d2fe_pipe_lanes(p_);
    std::vector<int32_t> c((size_t)2 * (K > 0 ? K : 0));
    int32_t n = 0;
    if (K <= 0 || d2fe_pipe_stream_placement(p_, c.data(), &n) != D2FE_OK) return 0;
    for (int k = 0; k < K; ++k) lanes.emplace_back(c[2 * k], c[2 * k + 1]);
    return n;
  }

 private:
  d2fe_pipe p_ = nullptr;
};

// feature_matcher.h:6-11.  `h` replaces the implicit global state of cv::BFMatcher; everything else as in the reference.
inline std::vector<DMatch> matchKNN(d2fe_handle h, const DescView& desc_a, const DescView& desc_b, double knn_match_ratio = 0.8,
                                    const std::vector<Point2f>& pts_a = std::vector<Point2f>(),
                                    const std::vector<Point2f>& pts_b = std::vector<Point2f>(), double search_local_dist = -1) {
  const int na = desc_a.rows, nb = desc_b.rows, cap = na > 0 ? na : 1;
  std::vector<int32_t> q(cap), t(cap);
  std::vector<float> d(cap);
  int n = 0;
  const bool gate = search_local_dist > 0 && (int)pts_a.size() == na && (int)pts_b.size() == nb && na > 0 && nb > 0;
  std::vector<DMatch> out;
  if (d2fe_match_knn(h, desc_a.data, na, desc_b.data, nb, desc_a.cols, knn_match_ratio, gate ? &pts_a[0].x : nullptr,
                     gate ? &pts_b[0].x : nullptr, search_local_dist, q.data(), t.data(), d.data(), cap, &n) != D2FE_OK)
    return out;
  out.reserve(n);
  for (int i = 0; i < n; ++i) out.emplace_back(q[i], t[i], d[i]);
  return out;
}

// cv::BFMatcher(cv::NORM_L2, true).match(desc_a, desc_b, matches)
inline std::vector<DMatch> matchCrossCheck(d2fe_handle h, const DescView& desc_a, const DescView& desc_b) {
  const int na = desc_a.rows, cap = na > 0 ? na : 1;
  std::vector<int32_t> q(cap), t(cap);
  std::vector<float> d(cap);
  int n = 0;
  std::vector<DMatch> out;
  if (d2fe_match_crosscheck(h, desc_a.data, na, desc_b.data, desc_b.rows, desc_a.cols, q.data(), t.data(), d.data(), cap, &n) != D2FE_OK) return out;
  for (int i = 0; i < n; ++i) out.emplace_back(q[i], t[i], d[i]);
  return out;
}

// getFeatureHalfImg (d2featuretracker.cpp:1051-1075): returns the kept descriptors; pts / tmp_to_idx are filled like the reference's
inline std::vector<float> getFeatureHalfImg(const std::vector<Point2f>& pts, const DescView& desc, bool require_left, int width_undistort,
                                            double undistort_fov, std::vector<Point2f>& pts_out, std::vector<int>& tmp_to_idx) {
  std::vector<int32_t> map(pts.size() ? pts.size() : 1);
  int n = 0;
  pts_out.clear(); tmp_to_idx.clear();
  std::vector<float> out;
  if (pts.empty() || d2fe_half_image_filter(&pts[0].x, (int)pts.size(), require_left ? 1 : 0, width_undistort, undistort_fov, map.data(), &n) != D2FE_OK)
    return out;
  out.reserve((size_t)n * desc.cols);
  for (int i = 0; i < n; ++i) {
    tmp_to_idx.push_back(map[i]);
    pts_out.push_back(pts[map[i]]);
    out.insert(out.end(), desc.data + (size_t)map[i] * desc.cols, desc.data + (size_t)(map[i] + 1) * desc.cols);
  }
  return out;
}

// ---- A8: struct fill of LoopCam::extractorImgDescDeepnet (loop_cam.cpp:619-645) -------------------------------------------------
// Host work on <= max_keypoints points: liftProjective of the camera the keypoints live in, normalisation, NaN rejection.
// The three camera models the shipped configurations use are restated from the reference's camera_models package
// (camodocal): pinhole + radtan (realsense_d435), MEI / "omni" + radtan (raw fisheye), cylindrical (undistorted quadcam views).
struct Vec3d { double x = 0, y = 0, z = 0; };

struct RadTan { double k1 = 0, k2 = 0, p1 = 0, p2 = 0; };
inline void radtanDistortion(const RadTan& d, double mx, double my, double& dx, double& dy) {   // PinholeCamera/CataCamera::distortion
  const double mx2 = mx * mx, my2 = my * my, mxy = mx * my, rho2 = mx2 + my2;
  const double rad = d.k1 * rho2 + d.k2 * rho2 * rho2;
  dx = mx * rad + 2.0 * d.p1 * mxy + d.p2 * (rho2 + 2.0 * mx2);
  dy = my * rad + 2.0 * d.p2 * mxy + d.p1 * (rho2 + 2.0 * my2);
}
inline void radtanUndistort(const RadTan& d, double mx_d, double my_d, double& mx_u, double& my_u) {   // the "recursive distortion model", n = 8
  double dx, dy;
  radtanDistortion(d, mx_d, my_d, dx, dy);
  mx_u = mx_d - dx; my_u = my_d - dy;
  for (int i = 1; i < 8; ++i) {
    radtanDistortion(d, mx_u, my_u, dx, dy);
    mx_u = mx_d - dx; my_u = my_d - dy;
  }
}
// PinholeCamera::liftProjective (camera_models/src/camera_models/PinholeCamera.cc:337-395)
inline Vec3d liftProjectivePinhole(double fx, double fy, double cx, double cy, const RadTan& d, const Point2f& p) {
  const double mx_d = (1.0 / fx) * p.x + (-cx / fx), my_d = (1.0 / fy) * p.y + (-cy / fy);
  double mx_u = mx_d, my_u = my_d;
  if (d.k1 != 0 || d.k2 != 0 || d.p1 != 0 || d.p2 != 0) radtanUndistort(d, mx_d, my_d, mx_u, my_u);
  return Vec3d{mx_u, my_u, 1.0};
}
// CataCamera::liftProjective (CataCamera.cc:425-487); cam as in d2fe_mei_camera
inline Vec3d liftProjectiveMEI(const d2fe_mei_camera& c, const Point2f& p) {
  const double mx_d = (1.0 / c.gamma1) * p.x + (-c.u0 / c.gamma1), my_d = (1.0 / c.gamma2) * p.y + (-c.v0 / c.gamma2);
  double mx_u = mx_d, my_u = my_d;
  const RadTan d{c.k1, c.k2, c.p1, c.p2};
  if (d.k1 != 0 || d.k2 != 0 || d.p1 != 0 || d.p2 != 0) radtanUndistort(d, mx_d, my_d, mx_u, my_u);
  const double xi = c.xi;
  if (xi == 1.0) return Vec3d{mx_u, my_u, (1.0 - mx_u * mx_u - my_u * my_u) / 2.0};
  const double rho2 = mx_u * mx_u + my_u * my_u;
  return Vec3d{mx_u, my_u, 1.0 - xi * (rho2 + 1.0) / (xi + std::sqrt(1.0 + (1.0 - xi * xi) * rho2))};
}
// CylindricalCamera::liftProjective (CylindricalCamera.cc:207-220)
inline Vec3d liftProjectiveCylindrical(double fx, double fy, double cx, double cy, const Point2f& p) {
  const double phi = (1.0 / fx) * p.x + (-cx / fx), y_by_rho = (1.0 / fy) * p.y + (-cy / fy);
  const double z = std::fabs(phi) > M_PI / 2 ? -1.0 : 1.0;
  const double x = z * std::tan(phi);
  return Vec3d{x, y_by_rho * std::sqrt(x * x + z * z), z};
}

struct Landmark { Point2f pt2d; Vec3d pt3d_norm; int index = -1; };   // index = position of the keypoint (and of its descriptor)
// loop_cam.cpp:619-645: lift, normalise, SKIP on NaN.  The reference does not remove the skipped keypoint's descriptor (its own
// warning, :627-629), so landmark i and descriptor i go out of step after a NaN; `index` keeps the association here.
template <typename Lift>
inline std::vector<Landmark> fillLandmarks(const std::vector<Point2f>& keypoints, Lift lift) {
  std::vector<Landmark> out;
  out.reserve(keypoints.size());
  for (size_t i = 0; i < keypoints.size(); ++i) {
    Vec3d P = lift(keypoints[i]);
    const double n = std::sqrt(P.x * P.x + P.y * P.y + P.z * P.z);
    P.x /= n; P.y /= n; P.z /= n;
    if (std::isnan(P.x) || std::isnan(P.y) || std::isnan(P.z)) continue;
    Landmark lm; lm.pt2d = keypoints[i]; lm.pt3d_norm = P; lm.index = (int)i;
    out.push_back(lm);
  }
  return out;
}

// ---- LK tracker (opticaltrack_utils.h / .cpp) ---------------------------------------------------------------------------------
enum TrackLRType { WHOLE_IMG_MATCH = 0, LEFT_RIGHT_IMG_MATCH = 1, RIGHT_LEFT_IMG_MATCH = 2 };
constexpr int PYR_LEVEL = 2;          // opticaltrack_utils.h:10
constexpr int LK_WIN_SIZE = 21;       // WIN_SIZE, opticaltrack_utils.cpp:25
constexpr int LK_ITERS = 30;          // opticaltrack_utils.cpp:239

struct LKImageInfo {                  // LKImageInfoGPU (opticaltrack_utils.h:16-23); pyr is owned by the caller of buildImagePyramid
  std::vector<Point2f> lk_pts;
  std::vector<int64_t> lk_ids;
  std::vector<int> lk_local_index;
  std::vector<int> lk_types;
  d2fe_lk_frame pyr = nullptr;
};

inline d2fe_lk_frame buildImagePyramid(d2fe_handle h, const ImageView& img, int maxLevel = PYR_LEVEL) {
  d2fe_lk_frame f = nullptr;
  if (d2fe_lk_frame_create(h, img.data, img.cols, img.rows, (int)img.step, maxLevel, &f) != D2FE_OK) return nullptr;
  return f;
}

template <typename T>
inline void reduceVector(std::vector<T>& v, const std::vector<uint8_t>& status) {
  size_t j = 0;
  for (size_t i = 0; i < v.size() && i < status.size(); ++i)
    if (status[i]) v[j++] = v[i];
  v.resize(j);
}

// opticalflowTrackPyr (opticaltrack_utils.cpp:173-279); undistort_fov = params->undistort_fov
inline LKImageInfo opticalflowTrackPyr(d2fe_handle h, const ImageView& cur_img, const LKImageInfo& prev_lk, TrackLRType type,
                                       double undistort_fov) {
  LKImageInfo ret;
  ret.pyr = buildImagePyramid(h, cur_img, PYR_LEVEL);
  std::vector<Point2f> prev_pts = prev_lk.lk_pts, cur_pts;
  std::vector<int64_t> ids = prev_lk.lk_ids;
  std::vector<int> types = prev_lk.lk_types, local = prev_lk.lk_local_index;
  if (prev_pts.empty() || !ret.pyr || !prev_lk.pyr) return ret;
  const float move_cols = (float)(cur_img.cols * 90.0 / undistort_fov);
  if (type == WHOLE_IMG_MATCH) {
    cur_pts = prev_pts;
  } else {
    std::vector<uint8_t> keep(prev_pts.size(), 0);
    for (size_t i = 0; i < prev_pts.size(); ++i) {
      Point2f pt = prev_pts[i];
      if (type == LEFT_RIGHT_IMG_MATCH ? pt.x < cur_img.cols - move_cols : pt.x >= move_cols) {
        pt.x += type == LEFT_RIGHT_IMG_MATCH ? move_cols : -move_cols;
        keep[i] = 1;
        cur_pts.push_back(pt);
      }
    }
    reduceVector(prev_pts, keep); reduceVector(types, keep); reduceVector(local, keep); reduceVector(ids, keep);
  }
  if (cur_pts.empty()) return ret;
  std::vector<uint8_t> status(cur_pts.size());
  if (d2fe_lk_track(h, prev_lk.pyr, ret.pyr, &prev_pts[0].x, &cur_pts[0].x, (int)cur_pts.size(), (int)type, move_cols, LK_WIN_SIZE,
                    LK_ITERS, &cur_pts[0].x, status.data()) != D2FE_OK)
    return ret;
  reduceVector(cur_pts, status); reduceVector(ids, status); reduceVector(types, status); reduceVector(local, status);
  ret.lk_pts = cur_pts; ret.lk_ids = ids; ret.lk_types = types; ret.lk_local_index = local;
  return ret;
}

// detectPoints (opticaltrack_utils.cpp:375-442); feature_min_dist = params->feature_min_dist
inline void detectPoints(d2fe_handle h, d2fe_lk_frame frame, std::vector<Point2f>& n_pts, const std::vector<Point2f>& cur_pts,
                         int require_pts, bool use_fast = false, int fast_rows = 3, int fast_cols = 4, double feature_min_dist = 20) {
  const int lack_up_top_pts = require_pts - (int)cur_pts.size();
  n_pts.clear();
  if (!(lack_up_top_pts > require_pts / 4)) return;
  const int num_to_detect = cur_pts.empty() ? lack_up_top_pts : lack_up_top_pts * 2;
  std::vector<Point2f> n_pts_tmp((size_t)num_to_detect);
  int n = 0;
  const int rc = use_fast ? d2fe_detect_fast_by_region(h, frame, num_to_detect, fast_rows, fast_cols, 10, &n_pts_tmp[0].x, nullptr, num_to_detect, &n)
                          : d2fe_good_features_to_track(h, frame, num_to_detect, 0.01, feature_min_dist, &n_pts_tmp[0].x, num_to_detect, &n);
  if (rc != D2FE_OK) return;
  n_pts_tmp.resize(n);
  std::vector<Point2f> all_pts = cur_pts;
  for (const Point2f& pt : n_pts_tmp) {
    bool has_nearby = false;
    for (const Point2f& pt_j : all_pts) {
      const float dx = pt.x - pt_j.x, dy = pt.y - pt_j.y;
      if (std::sqrt((double)dx * dx + (double)dy * dy) < feature_min_dist) { has_nearby = true; break; }
    }
    if (!has_nearby) { n_pts.push_back(pt); all_pts.push_back(pt); }
    if ((int)n_pts.size() >= lack_up_top_pts) break;
  }
}

}  // namespace D2FrontEnd

#endif  // D2FE_HPP_
