// d2fe_adapter.cpp -- the translation unit a D2SLAM maintainer adds to d2frontend (e.g. as d2frontend/src/CNN/superpoint_hip.cpp) so that
// LoopCam keeps its members and call sites and runs on libd2fe_hip.so (compiled and tested since round 6):
//
//   std::unique_ptr<SuperPoint> superpoint_ptr;   loop_cam.h:78-82    ->  D2FrontEnd::SuperPoint, the reference's OWN class declaration
//                                                                         (d2frontend/include/d2frontend/CNN/superpoint_tensorrt.h:36-93); this file
//                                                                         defines its constructor, build() and the "middle level" infer() of
//                                                                         d2frontend/src/CNN/superpoint_tensorrt.cpp:17-20,22-107,161-183
//   MobileNetVLADONNX* netvlad_onnx;              loop_cam.h:77       ->  D2FrontEnd::MobileNetVLADONNX below: the constructor signature and
//                                                                         inference() of d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:18-74
//                                                                         (that header needs ONNX Runtime; this class replaces it under USE_HIP)
//
// LoopCam::extractorImgDescDeepnet (d2frontend/src/loop_cam.cpp:589-648) is compiled UNCHANGED over this file by oracle/build_ref.py and held field by
// field to the same function over the reference's own SuperPoint::infer post-processing (tests/test_loopcam_adapter.py).
//
// The TensorRT members of the class declaration stay untouched (a maintainer would drop them under USE_HIP); the d2fe handle of an object lives in a
// table keyed by its address.  Model FILES: super_point_config_.onnx_path / the NetVLAD engine_path name a D2FW container (include/d2fe_weights_file.hpp,
// written by d2slam_amd/weights.py from the .pth / .onnx files D2SLAM ships links to); PCA of the global descriptor through params->pca_netvlad as before.
//
// Compile with the include paths of d2frontend (for superpoint_tensorrt.h, opencv2, Eigen) and -DUSE_CUDA (the class declaration is inside that #ifdef).
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "d2frontend/CNN/superpoint_tensorrt.h"

#include "d2fe.h"
#include "d2fe_weights_file.hpp"

namespace D2FrontEnd {
namespace hip_adapter {
struct Entry { d2fe_handle h = nullptr; d2fe_weights::File file; };
inline std::map<const void*, Entry>& table() { static std::map<const void*, Entry> t; return t; }
inline std::mutex& table_mutex() { static std::mutex m; return m; }
inline d2fe_handle handle_of(const void* obj) {
  std::lock_guard<std::mutex> g(table_mutex());
  auto it = table().find(obj);
  return it == table().end() ? nullptr : it->second.h;
}
// precision of the convolutions: D2FE_PREC_F32 (direct fmaf chains, the ABI default), D2FE_PREC_F32_WINO (fp32 Winograd, bench.py's headline mode) or
// D2FE_PREC_F16X2; one switch for the process, set before build()
inline int& precision() { static int p = D2FE_PREC_F32; return p; }
inline void release(const void* obj) {
  std::lock_guard<std::mutex> g(table_mutex());
  auto it = table().find(obj);
  if (it != table().end()) { if (it->second.h) d2fe_destroy(it->second.h); table().erase(it); }
}
}  // namespace hip_adapter

// superpoint_tensorrt.cpp:17-20
SuperPoint::SuperPoint(const SuperPointConfig& super_point_config) : super_point_config_(super_point_config), engine_(nullptr), context_(nullptr) {}

// superpoint_tensorrt.cpp:22-107 (parse the ONNX / deserialise the engine) -> create the handle, load the 12 layers
bool SuperPoint::build() {
  d2fe_config c;
  d2fe_default_config(&c);
  c.max_width = super_point_config_.input_width;           // SuperPointConfig fields map 1:1 (superpoint_tensorrt.h:17-33)
  c.max_height = super_point_config_.input_height;
  c.max_batch = 1;
  c.max_keypoints = super_point_config_.max_keypoints;
  c.remove_borders = super_point_config_.remove_borders;
  c.keypoint_threshold = super_point_config_.keypoint_threshold;
  c.postproc = D2FE_POSTPROC_B;                            // the live USE_CUDA path
  c.precision = hip_adapter::precision();
  hip_adapter::Entry e;
  std::string err;
  if (!e.file.load(super_point_config_.onnx_path)) { std::fprintf(stderr, "[d2fe] SuperPoint::build: %s\n", e.file.error.c_str()); return false; }
  d2fe_superpoint_weights w;
  if (!d2fe_weights::superpoint(e.file, &w, &err)) { std::fprintf(stderr, "[d2fe] SuperPoint::build: %s\n", err.c_str()); return false; }
  if (d2fe_create(&c, &e.h) != D2FE_OK) { std::fprintf(stderr, "[d2fe] d2fe_create: %s\n", d2fe_last_error()); return false; }
  if (d2fe_load_superpoint(e.h, &w) != D2FE_OK) { std::fprintf(stderr, "[d2fe] d2fe_load_superpoint: %s\n", d2fe_last_error()); d2fe_destroy(e.h); return false; }
  e.file.t.clear();                                        // the library copied and re-packed the weights
  hip_adapter::release(this);
  std::lock_guard<std::mutex> g(hip_adapter::table_mutex());
  hip_adapter::table()[this] = std::move(e);
  return true;
}

// superpoint_tensorrt.cpp:161-183, the "middle level for D2SLAM": keypoints are APPENDED (not cleared) on success (:172-174), the descriptors appended
// (:175-178), the scores assigned (:179); all three cleared and false on failure (:164-170)
bool SuperPoint::infer(const cv::Mat& input, std::vector<cv::Point2f>& keypoints, std::vector<float>& local_descriptors, std::vector<float>& scores) {
  d2fe_handle h = hip_adapter::handle_of(this);
  // max_keypoints == -1 is legal (topKeypoints keeps everything, :241-253): the buffers are sized from a stated capacity, never from -1; on
  // D2FE_ERR_TRUNCATED (more keypoints than the buffers hold) the call is repeated with room -- infer() returns false, it never throws
  const long most = (long)input.rows * input.cols;
  long cap = super_point_config_.max_keypoints > 0 ? super_point_config_.max_keypoints : (most < 4096 ? most : 4096);
  std::vector<float> kp, sc, de;
  int n = 0, rc = D2FE_ERR_INVALID;
  const int dim = h ? d2fe_desc_dim(h) : 256;
  while (h && input.data && input.channels() == 1 && cap > 0) {
    kp.resize(2 * (size_t)cap); sc.resize((size_t)cap); de.resize((size_t)dim * cap);
    rc = d2fe_superpoint_extract(h, input.data, input.cols, input.rows, (int)input.step, kp.data(), sc.data(), de.data(), (int)cap, &n);
    if (rc != D2FE_ERR_TRUNCATED || super_point_config_.max_keypoints > 0 || cap >= most) break;
    cap = 4 * cap < most ? 4 * cap : most;
  }
  if (rc != D2FE_OK) {
    keypoints.clear(); local_descriptors.clear(); scores.clear();
    std::fprintf(stderr, "[d2fe] superpoint infer failed: %s\n", d2fe_last_error());
    return false;
  }
  for (int i = 0; i < n; ++i) keypoints.emplace_back(kp[2 * i], kp[2 * i + 1]);
  local_descriptors.insert(local_descriptors.end(), de.begin(), de.begin() + (size_t)dim * n);
  scores.assign(sc.begin(), sc.begin() + n);
  return true;
}

// mobilenetvlad_onnx.h:18-74 under USE_HIP: same constructor arguments (the TensorRT / precision switches are accepted and ignored), same inference()
class MobileNetVLADONNX {
 public:
  const int descriptor_size = 4096;
  int width, height;
  MobileNetVLADONNX(std::string engine_path, int _width, int _height, bool /*use_tensorrt*/ = true, bool /*use_fp16*/ = true, bool /*use_int8*/ = false,
                    std::string /*int8_calib_table_name*/ = "")
      : width(_width), height(_height) {
    d2fe_config c;
    d2fe_default_config(&c);
    c.max_width = _width; c.max_height = _height; c.max_batch = 1;
    d2fe_weights::File f;
    std::vector<d2fe_nv_layer> layers;
    d2fe_netvlad_weights w;
    std::string err;
    if (!f.load(engine_path) || !d2fe_weights::netvlad(f, &layers, &w, &err)) { std::fprintf(stderr, "[d2fe] MobileNetVLADONNX: %s%s\n", f.error.c_str(), err.c_str()); return; }
    if (d2fe_create(&c, &h_) != D2FE_OK || d2fe_load_netvlad(h_, &w) != D2FE_OK) {
      std::fprintf(stderr, "[d2fe] MobileNetVLADONNX: %s\n", d2fe_last_error());
      if (h_) d2fe_destroy(h_);
      h_ = nullptr;
    }
  }
  ~MobileNetVLADONNX() { if (h_) d2fe_destroy(h_); }
  MobileNetVLADONNX(const MobileNetVLADONNX&) = delete;
  MobileNetVLADONNX& operator=(const MobileNetVLADONNX&) = delete;
  // the CSV of the reference (row 0 = mean, rows 1.. = components, :35-41) already parsed: comp [m][4096], mean [4096]
  bool setPCA(const float* comp, const float* mean, int m) { return h_ && d2fe_set_netvlad_pca(h_, comp, mean, m) == D2FE_OK; }
  // :49-74: BGR -> gray and resize to the network's size where needed (d2fe_prepare_gray), NO scaling of the pixel values (the graph holds the
  // (x - 128) / 128), PCA + L2 inside the library when set; an empty vector on failure
  std::vector<float> inference(const cv::Mat& input) {
    if (!h_ || !input.data) return {};
    std::vector<float> out((size_t)d2fe_netvlad_dim(h_));
    const uint8_t* img = input.data;
    int stride = (int)input.step;
    if (input.channels() != 1 || input.rows != height || input.cols != width) {
      tmp_.resize((size_t)width * height);
      if (d2fe_prepare_gray(h_, input.data, input.channels(), input.cols, input.rows, (int)input.step, width, height, tmp_.data()) != D2FE_OK) return {};
      img = tmp_.data(); stride = width;
    }
    if (d2fe_netvlad(h_, img, width, height, stride, out.data()) != D2FE_OK) return {};
    return out;
  }

 private:
  d2fe_handle h_ = nullptr;
  std::vector<uint8_t> tmp_;
};
}  // namespace D2FrontEnd
