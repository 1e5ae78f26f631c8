// spref_api.cpp -- extern "C" entry points of oracle/_ref/libspref.so (test infrastructure; never linked into the product).
// The function bodies that do the work are the REFERENCE'S OWN line ranges, compiled where they lie under /root/reference:
// oracle/build_ref.py writes them into the SPREF_GEN_* include files of a temporary build directory (never into this
// repository) and compiles this file against the stand-in headers of oracle/ref_shim/.  The class declaration
// (D2FrontEnd::SuperPoint, SuperPointConfig) is the reference's own header, included in place.
#define USE_CUDA 1
#define private public          /* the wrappers below call SuperPoint::processOutput and set semi_dims_/desc_dims_ */
#include "d2frontend/CNN/superpoint_tensorrt.h"      /* /root/reference/d2frontend/include (found via -I) */
#undef private
#include <spdlog/spdlog.h>
#include <numeric>
#include "d2frontend/d2frontend_params.h"
#include "d2common/utils.hpp"
using D2Common::Utility::TicToc;

namespace D2FrontEnd {
D2FrontendParams* params = new D2FrontendParams();
const int32_t kSuperPointDescDim = 256;              /* superpoint_tensorrt.cpp:14 */
// the one constructor the wrappers need (superpoint_tensorrt.cpp:16-19 without the TensorRT logger call)
SuperPoint::SuperPoint(const SuperPointConfig& c) : super_point_config_(c), engine_(nullptr), context_(nullptr) {}
// stand-in for the TensorRT half of infer() (superpoint_tensorrt.cpp:128-159): the network outputs are handed in
static thread_local const tensorrt_buffer::BufferManager* g_buffers = nullptr;
bool SuperPoint::infer(const cv::Mat&, std::vector<Eigen::Vector2f>& keypoints, std::vector<Eigen::VectorXf>& descriptors,
                       std::vector<float>& scores) {
  return processOutput(*g_buffers, keypoints, descriptors, scores);
}
}  // namespace D2FrontEnd

namespace D2FrontEnd {
#include SPREF_GEN_TENSORRT_INFER      /* superpoint_tensorrt.cpp:161-183  SuperPoint::infer (the D2SLAM middle level) */
#include SPREF_GEN_TENSORRT_POST       /* superpoint_tensorrt.cpp:200-350  findHighScoreIndex .. processOutput */
}
#include SPREF_GEN_COMMON_KPS          /* superpoint_common.cpp:8-40       namespace open, getKeyPoints */
#include SPREF_GEN_COMMON_NMS          /* superpoint_common.cpp:101-178    pt_conf_comp, NMS2, namespace close */
#include "d2frontend/feature_matcher.h"               /* the reference's own declaration (default arguments) */
#include SPREF_GEN_MATCHER             /* feature_matcher.cpp:3-43 */

namespace D2FrontEnd {
enum TrackLRType { WHOLE_IMG_MATCH = 0, LEFT_RIGHT_IMG_MATCH, RIGHT_LEFT_IMG_MATCH };   /* d2featuretracker.h:18-22 */
struct ShimFTConfig { bool enable_knn_match = true; double knn_match_ratio = 0.8; };
struct ShimMatchParams { bool enable_search_in_local = false; TrackLRType type = WHOLE_IMG_MATCH; };
#include SPREF_GEN_HALFIMG             /* d2featuretracker.cpp:1051-1075   getFeatureHalfImg */
// the quadcam neighbour branch of D2FeatureTracker::matchLocalFeatures, d2featuretracker.cpp:1146-1181, with the locals it reads
static bool neighbour_branch(const std::vector<cv::Point2f>& pts_a, const std::vector<float>& raw_desc_a,
                             const std::vector<cv::Point2f>& pts_b, const std::vector<float>& raw_desc_b, const ShimMatchParams& param,
                             const ShimFTConfig& _config, double search_radius, std::vector<cv::DMatch>& _matches) {
#include SPREF_GEN_NEIGHBOUR
  return true;
}
}  // namespace D2FrontEnd

extern "C" {
#define SPREF_API __attribute__((visibility("default")))

// SuperPoint::infer(cv::Mat, vector<Point2f>&, vector<float>&, vector<float>&) on given network outputs:
// semi [h][w] fp32, desc CHW [dim][hc][wc] fp32 (the TensorRT output layouts).  Returns the keypoint count, -1 if cap is too small.
SPREF_API int spref_superpoint_post(const float* semi, int h, int w, const float* desc, int dim, int hc, int wc, float threshold,
                                    int remove_borders, int max_keypoints, float* kps_xy, float* scores, float* desc_out, int cap) {
  D2FrontEnd::SuperPointConfig cfg;
  cfg.max_keypoints = max_keypoints; cfg.remove_borders = remove_borders; cfg.keypoint_threshold = threshold;
  cfg.output_tensor_names = {"scores", "descriptors"};
  D2FrontEnd::SuperPoint sp(cfg);
  sp.semi_dims_.nbDims = 3; sp.semi_dims_.d[0] = 1; sp.semi_dims_.d[1] = h; sp.semi_dims_.d[2] = w;
  sp.desc_dims_.nbDims = 4; sp.desc_dims_.d[0] = 1; sp.desc_dims_.d[1] = dim; sp.desc_dims_.d[2] = hc; sp.desc_dims_.d[3] = wc;
  tensorrt_buffer::BufferManager buffers;
  buffers.host["scores"] = const_cast<float*>(semi);
  buffers.host["descriptors"] = const_cast<float*>(desc);
  D2FrontEnd::g_buffers = &buffers;
  std::vector<cv::Point2f> kps; std::vector<float> d, s;
  cv::Mat dummy;
  if (!sp.infer(dummy, kps, d, s)) return -2;
  const int n = (int)kps.size();
  if (n > cap) return -1;
  if ((size_t)n * dim != d.size() || (size_t)n != s.size()) return -3;
  for (int i = 0; i < n; ++i) { kps_xy[2 * i] = kps[i].x; kps_xy[2 * i + 1] = kps[i].y; scores[i] = s[i]; }
  std::copy(d.begin(), d.end(), desc_out);
  return n;
}

// getKeyPoints(prob, threshold, nms_dist, keypoints, scores, width, height, max_num)  (superpoint_common.cpp:12-40 -> NMS2)
SPREF_API int spref_get_keypoints(const float* prob, int h, int w, float threshold, int nms_dist, int max_num, float* kps_xy,
                                  float* scores, int cap) {
  cv::Mat p(h, w, CV_32F, const_cast<float*>(prob));
  std::vector<cv::Point2f> kps; std::vector<float> sc;
  D2FrontEnd::getKeyPoints(p, threshold, nms_dist, kps, sc, w, h, max_num);
  const int n = (int)kps.size();
  if (n > cap) return -1;
  for (int i = 0; i < n; ++i) { kps_xy[2 * i] = kps[i].x; kps_xy[2 * i + 1] = kps[i].y; scores[i] = sc[i]; }
  return n;
}

static std::vector<cv::Point2f> to_pts(const float* p, int n) {
  std::vector<cv::Point2f> v;
  if (p) for (int i = 0; i < n; ++i) v.push_back(cv::Point2f(p[2 * i], p[2 * i + 1]));
  return v;
}

// matchKNN(desc_a, desc_b, ratio, pts_a, pts_b, search_local_dist)  (feature_matcher.cpp:4-42)
SPREF_API int spref_match_knn(const float* a, int na, const float* b, int nb, int dim, double ratio, const float* pts_a,
                              const float* pts_b, double radius, int32_t* q_idx, int32_t* t_idx, float* dist, int cap) {
  const cv::Mat da(na, dim, CV_32F, const_cast<float*>(a)), db(nb, dim, CV_32F, const_cast<float*>(b));
  const std::vector<cv::DMatch> m = D2FrontEnd::matchKNN(da, db, ratio, to_pts(pts_a, na), to_pts(pts_b, nb), radius);
  if ((int)m.size() > cap) return -1;
  for (size_t i = 0; i < m.size(); ++i) { q_idx[i] = m[i].queryIdx; t_idx[i] = m[i].trainIdx; dist[i] = m[i].distance; }
  return (int)m.size();
}

// getFeatureHalfImg(pts, desc, require_left, tmp_to_idx)  (d2featuretracker.cpp:1051-1075); map_out[c] = original index
SPREF_API int spref_half_image(const float* pts, int n, int dim, int require_left, int width_undistort, double undistort_fov,
                               int32_t* map_out) {
  D2FrontEnd::params->width_undistort = width_undistort; D2FrontEnd::params->undistort_fov = undistort_fov;
  D2FrontEnd::params->superpoint_dims = dim;
  std::vector<float> desc((size_t)n * dim, 0.f);
  std::map<int, int> m;
  D2FrontEnd::getFeatureHalfImg(to_pts(pts, n), desc, require_left != 0, m);
  for (auto& kv : m) map_out[kv.first] = kv.second;
  return (int)m.size();
}

// the LEFT_RIGHT / RIGHT_LEFT branch of matchLocalFeatures (d2featuretracker.cpp:1146-1181): half-image filter, +-move_cols shift,
// matchKNN with the radius gate (or cross-check match), index remap.  type: 1 = LEFT_RIGHT_IMG_MATCH, 2 = RIGHT_LEFT_IMG_MATCH.
// Returns the match count, -2 when the branch returns false (one side empty), -1 if cap is too small.
SPREF_API int spref_match_neighbour(const float* pts_a, const float* desc_a, int na, const float* pts_b, const float* desc_b, int nb,
                                    int dim, int type, int enable_knn_match, double ratio, int enable_search_in_local,
                                    double search_radius, int width_undistort, double undistort_fov, int32_t* q_idx, int32_t* t_idx,
                                    float* dist, int cap) {
  D2FrontEnd::params->width_undistort = width_undistort; D2FrontEnd::params->undistort_fov = undistort_fov;
  D2FrontEnd::params->superpoint_dims = dim;
  D2FrontEnd::ShimFTConfig cfg; cfg.enable_knn_match = enable_knn_match != 0; cfg.knn_match_ratio = ratio;
  D2FrontEnd::ShimMatchParams prm; prm.enable_search_in_local = enable_search_in_local != 0;
  prm.type = type == 1 ? D2FrontEnd::LEFT_RIGHT_IMG_MATCH : D2FrontEnd::RIGHT_LEFT_IMG_MATCH;
  std::vector<float> da(desc_a, desc_a + (size_t)na * dim), db(desc_b, desc_b + (size_t)nb * dim);
  std::vector<cv::DMatch> m;
  // matchLocalFeatures :1107-1112: without prediction and without enable_search_in_local the radius is disabled
  if (!prm.enable_search_in_local) search_radius = -1;
  if (!D2FrontEnd::neighbour_branch(to_pts(pts_a, na), da, to_pts(pts_b, nb), db, prm, cfg, search_radius, m)) return -2;
  if ((int)m.size() > cap) return -1;
  for (size_t i = 0; i < m.size(); ++i) { q_idx[i] = m[i].queryIdx; t_idx[i] = m[i].trainIdx; dist[i] = m[i].distance; }
  return (int)m.size();
}
}
