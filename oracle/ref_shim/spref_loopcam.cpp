// spref_loopcam.cpp -- oracle/_ref/libspref_loopcam_hip.so and libspref_loopcam_ref.so (test infrastructure; never linked into the product):
// LoopCam::extractorImgDescDeepnet, d2frontend/src/loop_cam.cpp:589-648, compiled UNCHANGED where it lies under /root/reference (the line range is written
// into a temporary include file by oracle/build_ref.py, nothing is copied into this repository) -- twice:
//
//   -DLOOPCAM_SIDE_HIP   superpoint_ptr / netvlad_onnx = include/d2fe_adapter.cpp over libd2fe_hip.so (what a D2SLAM maintainer adds)
//   -DLOOPCAM_SIDE_REF   superpoint_ptr = the reference's own SuperPoint::infer + processOutput (superpoint_tensorrt.cpp:161-183,200-350, compiled in place)
//                        fed by network outputs handed in from the test (the oracle's), netvlad_onnx = a holder of the oracle's global descriptor
//
// so that the two VisualImageDesc the reference's OWN caller fills can be compared field by field (tests/test_loopcam_adapter.py).  Also compiled in place:
// VisualImageDesc (d2common/include/d2common/d2frontend_types.h:85-110), LandmarkPerFrame (d2landmarks.h:28-70), the id typedefs and CameraConfig
// (d2basetypes.h:18-20,40-46), extractColor (d2frontend/src/loop_utils.cpp:54-63), camodocal's CataCamera::liftProjective (CataCamera.cc:221-224,425-487).
// Stand-ins (mine, restating third-party types): ros::Time, Swarm::Pose, spdlog macros, the cv:: types of ref_shim/opencv2, double-precision Eigen vectors
// below (Vector3d::normalize = Eigen 3.4 Dot.h: z = squaredNorm(); if (z > 0) *this /= sqrt(z), squaredNorm summed (v0^2 + v1^2) + v2^2 as the SSE2 packet
// reduction of a 3-vector does), and a LoopCam class declaration with exactly the members the function touches (loop_cam.h:66-104 needs ROS).
#define USE_CUDA 1
#include <chrono>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <numeric>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include <opencv2/opencv.hpp>            /* ref_shim */
#include <Eigen/Dense>                   /* ref_shim: the float types superpoint_tensorrt.h names */

#include <spdlog/spdlog.h>               /* ref_shim */
#undef SPDLOG_WARN
#undef SPDLOG_INFO
#define SPREF_API __attribute__((visibility("default")))
#define SPDLOG_WARN(...) (++g_nan_warnings)          /* loop_cam.cpp:628 "NaN detected!!!": counted */
#define SPDLOG_INFO(...) ((void)0)
static int g_nan_warnings = 0;
typedef unsigned char uchar;             /* OpenCV's global typedef (core/hal/interface.h), used unqualified by loop_utils.cpp:59 */

namespace Eigen {
struct Vector2d {
  double v[2];
  Vector2d() : v{0, 0} {}
  Vector2d(double a, double b) : v{a, b} {}
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  struct Comma { double* v; int i; Comma& operator,(double x) { v[i++] = x; return *this; } };
  Comma operator<<(double a) { v[0] = a; return Comma{v, 1}; }
  Vector2d operator+(const Vector2d& o) const { return Vector2d(v[0] + o.v[0], v[1] + o.v[1]); }
};
struct Vector3d {
  double v[3];
  struct Comma { double* v; int i; Comma& operator,(double x) { v[i++] = x; return *this; } };
  Comma operator<<(double a) { v[0] = a; return Comma{v, 1}; }
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  static Vector3d Zero() { return Vector3d(0, 0, 0); }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
  void normalize() { const double z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; if (z > 0) { const double s = std::sqrt(z); v[0] /= s; v[1] /= s; v[2] /= s; } }
  bool hasNaN() const { return !(v[0] == v[0] && v[1] == v[1] && v[2] == v[2]); }
};
}  // namespace Eigen

namespace ros { struct Time { double t = 0; Time() {} explicit Time(double s) : t(s) {} double toSec() const { return t; } }; }
namespace Swarm { struct Pose { double p[7] = {0, 0, 0, 1, 0, 0, 0}; }; }
struct Landmark_t;       // LCM type named by constructors outside the compiled ranges

namespace camodocal {
struct Camera {
  virtual ~Camera() {}
  virtual void liftProjective(const Eigen::Vector2d& p, Eigen::Vector3d& P) const = 0;
};
typedef std::shared_ptr<Camera> CameraPtr;
class CataCamera : public Camera {
 public:
  struct Parameters {
    double xi_, k1_, k2_, p1_, p2_, g1_, g2_, u0_, v0_;
    double xi() const { return xi_; } double k1() const { return k1_; } double k2() const { return k2_; } double p1() const { return p1_; }
    double p2() const { return p2_; } double gamma1() const { return g1_; } double gamma2() const { return g2_; }
    double u0() const { return u0_; } double v0() const { return v0_; }
  };
  Parameters mParameters;
  bool m_noDistortion = false;
  double m_inv_K11 = 1, m_inv_K13 = 0, m_inv_K22 = 1, m_inv_K23 = 0;
  void set_inverse_K() {
#include SPREF_GEN_CATA_INVK           /* CataCamera.cc:221-224 */
  }
  void liftProjective(const Eigen::Vector2d& p, Eigen::Vector3d& P) const override;
  void distortion(const Eigen::Vector2d& p_u, Eigen::Vector2d& d_u) const;
};
#include SPREF_GEN_CATA_LIFT           /* CataCamera.cc:425-487 */
#include SPREF_GEN_CATA_DIST           /* CataCamera.cc:617-633 (liftProjective's recursive distortion removal calls it) */
// pinhole camera without distortion (the d435 configuration's PinholeCamera::liftProjective with k1 = k2 = p1 = p2 = 0, PinholeCamera.cc: mx_u = m_inv_K11 * p(0) +
// m_inv_K13 ...): restated, two multiply-adds; only the MEI camera above is the reference's own code
class PinholeNoDist : public Camera {
 public:
  double fx, fy, cx, cy;
  PinholeNoDist(double fx_, double fy_, double cx_, double cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
  void liftProjective(const Eigen::Vector2d& p, Eigen::Vector3d& P) const override {
    const double i11 = 1.0 / fx, i13 = -cx / fx, i22 = 1.0 / fy, i23 = -cy / fy;
    P << i11 * p(0) + i13, i22 * p(1) + i23, 1.0;
  }
};
}  // namespace camodocal

namespace D2Common {
#include SPREF_GEN_BASETYPES_IDS       /* d2basetypes.h:18-20   FrameIdType, LandmarkIdType, CamIdType */
#include SPREF_GEN_BASETYPES_CAMCFG    /* d2basetypes.h:40-46   enum CameraConfig */
enum LandmarkFlag { UNINITIALIZED = 0, INITIALIZED = 1, ESTIMATED = 2, OUTLIER = 3 };       /* d2landmarks.h:11-16 */
enum LandmarkType { SuperPointLandmark, FlowLandmark };                                      /* d2landmarks.h:23-26 */
#include SPREF_GEN_LANDMARK_PER_FRAME  /* d2landmarks.h:28-70   struct LandmarkPerFrame { ... default constructor */
};
#include SPREF_GEN_VISUAL_IMAGE_DESC   /* d2frontend_types.h:85-110   struct VisualImageDesc { ... double cur_td = 0; */
};
}  // namespace D2Common
using namespace D2Common;
using namespace std::chrono;

#include "d2frontend/d2frontend_params.h"      /* ref_shim: params->superpoint_dims */

#ifdef LOOPCAM_SIDE_HIP
#include "../../include/d2fe_adapter.cpp"      /* D2FrontEnd::SuperPoint (constructor, build, infer) and D2FrontEnd::MobileNetVLADONNX over libd2fe_hip.so */
namespace D2FrontEnd { D2FrontendParams* params = new D2FrontendParams(); }
#else
#define private public
#include "d2frontend/CNN/superpoint_tensorrt.h"
#undef private
#include "d2common/utils.hpp"
using D2Common::Utility::TicToc;
namespace D2FrontEnd {
D2FrontendParams* params = new D2FrontendParams();
const int32_t kSuperPointDescDim = 256;              /* superpoint_tensorrt.cpp:14 */
SuperPoint::SuperPoint(const SuperPointConfig& c) : super_point_config_(c), engine_(nullptr), context_(nullptr) {}      /* :17-20 without the TensorRT logger */
// stand-in for the TensorRT half of the low-level infer() (superpoint_tensorrt.cpp:128-159): the network outputs are handed in by the test
static const tensorrt_buffer::BufferManager* g_buffers = nullptr;
bool SuperPoint::infer(const cv::Mat&, std::vector<Eigen::Vector2f>& keypoints, std::vector<Eigen::VectorXf>& descriptors, std::vector<float>& scores) {
  return processOutput(*g_buffers, keypoints, descriptors, scores);
}
#include SPREF_GEN_TENSORRT_INFER      /* superpoint_tensorrt.cpp:161-183  the middle level LoopCam calls */
#include SPREF_GEN_TENSORRT_POST       /* superpoint_tensorrt.cpp:200-350  findHighScoreIndex .. processOutput */
// holder of the global descriptor the test computed with the oracle (the reference's class needs ONNX Runtime and its missing model file)
class MobileNetVLADONNX {
 public:
  std::vector<float> preset;
  std::vector<float> inference(const cv::Mat&) { return preset; }
};
}  // namespace D2FrontEnd
#endif

namespace D2FrontEnd {
#include SPREF_GEN_EXTRACT_COLOR       /* loop_utils.cpp:54-63 */
struct LoopCamConfig { int superpoint_max_num = 200; bool cnn_use_onnx = true; bool OUTPUT_RAW_SUPERPOINT_DESC = false; };     /* loop_cam.h:35-64, the fields read */
class LoopCam {                                                                                                                   /* loop_cam.h:66-104, the members read */
 public:
  LoopCamConfig _config;
  int self_id = 0;
  CameraConfig camera_configuration = STEREO_PINHOLE;
  std::fstream fsp;
  MobileNetVLADONNX* netvlad_onnx = nullptr;
  std::unique_ptr<SuperPoint> superpoint_ptr;
  std::vector<camodocal::CameraPtr> cams;
  VisualImageDesc extractorImgDescDeepnet(ros::Time stamp, cv::Mat img, int index, int camera_id, bool superpoint_mode = false);
};
#include SPREF_GEN_LOOPCAM_EXTRACT     /* loop_cam.cpp:589-648  LoopCam::extractorImgDescDeepnet */
}  // namespace D2FrontEnd

// ---- C interface for the test -------------------------------------------------------------------------------------------------------------------------
namespace {
struct Session {
  D2FrontEnd::LoopCam cam;
  std::unique_ptr<D2FrontEnd::MobileNetVLADONNX> nv;
#ifndef LOOPCAM_SIDE_HIP
  tensorrt_buffer::BufferManager buffers;
#endif
};
}

extern "C" {
// cam_kind 0: pinhole {fx, fy, cx, cy}; 1: MEI / CataCamera {xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0}.  camera_configuration: D2Common::CameraConfig.
// HIP side: sp_path / nv_path = D2FW containers, precision = d2fe precision code.  REF side: paths ignored.
SPREF_API void* spref_loopcam_create(int width, int height, int max_keypoints, float threshold, int remove_borders, int self_id, int camera_configuration,
                                     int n_cams, const int* cam_kind, const double* cam_params /*[n_cams][9]*/, const char* sp_path, const char* nv_path, int precision) {
  auto* s = new Session();
  s->cam.self_id = self_id;
  s->cam.camera_configuration = (D2Common::CameraConfig)camera_configuration;
  s->cam._config.superpoint_max_num = max_keypoints;
  for (int i = 0; i < n_cams; ++i) {
    const double* c = cam_params + 9 * i;
    if (cam_kind[i] == 1) {
      auto cc = std::make_shared<camodocal::CataCamera>();
      cc->mParameters = camodocal::CataCamera::Parameters{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]};
      cc->m_noDistortion = c[1] == 0.0 && c[2] == 0.0 && c[3] == 0.0 && c[4] == 0.0;       /* CataCamera::setParameters, CataCamera.cc:208-219 */
      cc->set_inverse_K();
      s->cam.cams.push_back(cc);
    } else {
      s->cam.cams.push_back(std::make_shared<camodocal::PinholeNoDist>(c[0], c[1], c[2], c[3]));
    }
  }
  D2FrontEnd::SuperPointConfig cfg;
  cfg.max_keypoints = max_keypoints; cfg.remove_borders = remove_borders; cfg.keypoint_threshold = threshold;
  cfg.input_width = width; cfg.input_height = height;
  cfg.output_tensor_names = {"scores", "descriptors"};
#ifdef LOOPCAM_SIDE_HIP
  cfg.onnx_path = sp_path ? sp_path : "";
  D2FrontEnd::hip_adapter::precision() = precision;
  s->cam.superpoint_ptr.reset(new D2FrontEnd::SuperPoint(cfg));
  if (!s->cam.superpoint_ptr->build()) { delete s; return nullptr; }
  if (nv_path && nv_path[0]) { s->nv.reset(new D2FrontEnd::MobileNetVLADONNX(nv_path, width, height)); s->cam.netvlad_onnx = s->nv.get(); }
#else
  (void)sp_path; (void)nv_path; (void)precision;
  s->cam.superpoint_ptr.reset(new D2FrontEnd::SuperPoint(cfg));
  s->cam.superpoint_ptr->semi_dims_.nbDims = 3; s->cam.superpoint_ptr->semi_dims_.d[0] = 1; s->cam.superpoint_ptr->semi_dims_.d[1] = height; s->cam.superpoint_ptr->semi_dims_.d[2] = width;
  s->cam.superpoint_ptr->desc_dims_.nbDims = 4; s->cam.superpoint_ptr->desc_dims_.d[0] = 1; s->cam.superpoint_ptr->desc_dims_.d[1] = 256;
  s->cam.superpoint_ptr->desc_dims_.d[2] = height / 8; s->cam.superpoint_ptr->desc_dims_.d[3] = width / 8;
  s->nv.reset(new D2FrontEnd::MobileNetVLADONNX()); s->cam.netvlad_onnx = s->nv.get();
#endif
  s->cam._config.cnn_use_onnx = s->cam.netvlad_onnx != nullptr;
  return s;
}
SPREF_API void spref_loopcam_destroy(void* p) {
  auto* s = static_cast<Session*>(p);
#ifdef LOOPCAM_SIDE_HIP
  if (s && s->cam.superpoint_ptr) D2FrontEnd::hip_adapter::release(s->cam.superpoint_ptr.get());
#endif
  delete s;
}
// REF side only: the network outputs of the NEXT call (semi [H][W], desc CHW [256][H/8][W/8]: the TensorRT output layouts) and its global descriptor
SPREF_API void spref_loopcam_set_network_outputs(void* p, const float* semi, const float* desc, const float* netvlad, int netvlad_dim) {
#ifndef LOOPCAM_SIDE_HIP
  auto* s = static_cast<Session*>(p);
  s->buffers.host["scores"] = const_cast<float*>(semi);
  s->buffers.host["descriptors"] = const_cast<float*>(desc);
  D2FrontEnd::g_buffers = &s->buffers;
  s->nv->preset.assign(netvlad, netvlad + (netvlad ? netvlad_dim : 0));
#else
  (void)p; (void)semi; (void)desc; (void)netvlad; (void)netvlad_dim;
#endif
}
// One call of LoopCam::extractorImgDescDeepnet(stamp, img, camera_index, camera_id, superpoint_mode); `img` is modified in place where the reference does
// (the STEREO_FISHEYE mask).  Every field of the returned VisualImageDesc the function sets is flattened:
//   head[8]  = stamp, camera_index, camera_id, drone_id, landmarks.size(), landmark_descriptor.size(), landmark_scores.size(), image_desc.size()
//   per landmark i (cap_lm of them at most): pt2d[i][2], pt3d_norm[i][3], lm_meta[i][4] = camera_index, camera_id, stamp, stamp_discover, color[i][3]
// returns the landmark count, or -1 if a capacity is too small; *nan_warnings = SPDLOG_WARN calls (NaN-skipped keypoints)
SPREF_API int spref_loopcam_extract(void* p, double stamp, uint8_t* img, int width, int height, int stride, int camera_index, int camera_id, int superpoint_mode,
                                    double* head, float* pt2d, double* pt3d_norm, double* lm_meta, uint8_t* color, int cap_lm,
                                    float* landmark_descriptor, long cap_desc, float* landmark_scores, long cap_scores, float* image_desc, long cap_gdesc,
                                    int* nan_warnings) {
  auto* s = static_cast<Session*>(p);
  g_nan_warnings = 0;
  cv::Mat m(height, width, CV_8UC1, img);
  m.step = (size_t)stride;
  const D2Common::VisualImageDesc v = s->cam.extractorImgDescDeepnet(ros::Time(stamp), m, camera_index, camera_id, superpoint_mode != 0);
  if (nan_warnings) *nan_warnings = g_nan_warnings;
  head[0] = v.stamp; head[1] = v.camera_index; head[2] = v.camera_id; head[3] = v.drone_id; head[4] = (double)v.landmarks.size();
  head[5] = (double)v.landmark_descriptor.size(); head[6] = (double)v.landmark_scores.size(); head[7] = (double)v.image_desc.size();
  if ((long)v.landmarks.size() > cap_lm || (long)v.landmark_descriptor.size() > cap_desc || (long)v.landmark_scores.size() > cap_scores || (long)v.image_desc.size() > cap_gdesc) return -1;
  for (size_t i = 0; i < v.landmarks.size(); ++i) {
    const auto& lm = v.landmarks[i];
    pt2d[2 * i] = lm.pt2d.x; pt2d[2 * i + 1] = lm.pt2d.y;
    for (int k = 0; k < 3; ++k) { pt3d_norm[3 * i + k] = lm.pt3d_norm(k); color[3 * i + k] = lm.color[k]; }
    lm_meta[4 * i] = lm.camera_index; lm_meta[4 * i + 1] = lm.camera_id; lm_meta[4 * i + 2] = lm.stamp; lm_meta[4 * i + 3] = lm.stamp_discover;
  }
  std::copy(v.landmark_descriptor.begin(), v.landmark_descriptor.end(), landmark_descriptor);
  std::copy(v.landmark_scores.begin(), v.landmark_scores.end(), landmark_scores);
  std::copy(v.image_desc.begin(), v.image_desc.end(), image_desc);
  return (int)v.landmarks.size();
}
}
