// spref_api2.cpp -- second translation unit of oracle/_ref/libspref.so (test infrastructure; never linked into the product): the
// reference-OWNED code either side of the hot path, compiled where it lies under /root/reference (oracle/build_ref.py writes the line
// ranges into the SPREF_GEN_* include files of a temporary directory, nothing is copied into this repository):
//   int8 wire codec        d2common/include/d2common/d2frontend_types.h:230-236, 262-267 (VisualImageDesc::toLCM), 319-341 (LCM ctor)
//   NetVLAD database gate  d2frontend/src/loop_detector.cpp:300-350   LoopDetector::queryIndexFromDatabase
//   tracker gate           d2frontend/src/d2featuretracker.cpp:166-235 getMatchedPrevKeyframe, :270-284 the view pairing of trackRemoteFrames
//   stereo cross-check     d2frontend/src/loop_cam.cpp:156-191        matchLocalFeatures
// The stand-in types below carry exactly the members those ranges touch.
#include <Eigen/Dense>
#include <opencv2/opencv.hpp>
#include <spdlog/spdlog.h>
#include <cstdint>
#include <iostream>
#include <map>
#include <mutex>
#include <vector>

#define SPREF_API __attribute__((visibility("default")))
#define ROS_ERROR(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define SEARCH_NEAREST_NUM 5             /* d2frontend_params.h:22 */
#define ACCEPT_SP_MATCH_DISTANCE 0.7     /* d2frontend_params.h:31 */
#define REMOTE_MAGIN_NUMBER 1000000      /* loop_detector.h:14 */

using namespace Eigen;

// ---- int8 codec ----------------------------------------------------------------------------------------------------------------
namespace {
struct ShimHeader { int32_t image_desc_size_int8 = 0, image_desc_size = 0; std::vector<int8_t> image_desc_int8; std::vector<float> image_desc; };
struct ShimImageDescriptor_t {
  ShimHeader header;
  int32_t landmark_descriptor_size_int8 = 0, landmark_descriptor_size = 0, landmark_num = 0, landmark_scores_size = 0;
  std::vector<int8_t> landmark_descriptor_int8;
  std::vector<float> landmark_descriptor;
};
}  // namespace

extern "C" {
// VisualImageDesc::toLCM, landmark descriptors (compress_int8): float max
SPREF_API void spref_quant_landmarks(const float* x, int n, int8_t* out) {
  std::vector<float> landmark_descriptor(x, x + n);
  ShimImageDescriptor_t img_desc;
#include SPREF_GEN_CODEC_Q_LM          /* d2frontend_types.h:230-236 */
  std::copy(img_desc.landmark_descriptor_int8.begin(), img_desc.landmark_descriptor_int8.end(), out);
}
// VisualImageDesc::toLCM, NetVLAD descriptor: double max
SPREF_API void spref_quant_netvlad(const float* x, int n, int8_t* out) {
  std::vector<float> image_desc(x, x + n);
  ShimImageDescriptor_t img_desc;
#include SPREF_GEN_CODEC_Q_NV          /* d2frontend_types.h:262-267 */
  std::copy(img_desc.header.image_desc_int8.begin(), img_desc.header.image_desc_int8.end(), out);
}
// VisualImageDesc(const ImageDescriptor_t&): q/127.0, per-landmark 32-float renormalisation, whole-vector NetVLAD normalisation
SPREF_API void spref_dequant(const int8_t* lm, int n_lm, int landmark_num, const int8_t* nv, int n_nv, float* lm_out, float* nv_out) {
  ShimImageDescriptor_t desc;
  desc.landmark_descriptor_int8.assign(lm, lm + n_lm); desc.landmark_num = landmark_num;
  desc.header.image_desc_int8.assign(nv, nv + n_nv); desc.header.image_desc_size_int8 = n_nv;
  std::vector<float> landmark_descriptor, image_desc;
#include SPREF_GEN_CODEC_DEQ           /* d2frontend_types.h:319-341 */
  std::copy(landmark_descriptor.begin(), landmark_descriptor.end(), lm_out);
  std::copy(image_desc.begin(), image_desc.end(), nv_out);
}
}

// ---- LoopDetector::queryIndexFromDatabase over a stand-in faiss::IndexFlatIP ------------------------------------------------------
namespace faiss {
typedef int64_t idx_t;
// IndexFlatIP::search restated (faiss 1.7.4 is third-party and absent): k largest inner products, descending; the summation order of
// faiss's SIMD kernel and its heap's order among equal similarities are unspecified -- sequential fp32 sum, lower label first here
struct IndexFlatIP {
  int d; idx_t ntotal = 0; std::vector<float> xb;
  explicit IndexFlatIP(int d_) : d(d_) {}
  void add(idx_t n, const float* x) { xb.insert(xb.end(), x, x + (size_t)n * d); ntotal += n; }
  void search(idx_t nq, const float* q, idx_t k, float* sims, idx_t* labels) const {
    std::vector<float> s((size_t)ntotal);
    for (idx_t i = 0; i < ntotal; ++i) { float a = 0.f; for (int j = 0; j < d; ++j) a += xb[(size_t)i * d + j] * q[j]; s[(size_t)i] = a; }
    std::vector<char> used((size_t)ntotal, 0);
    for (idx_t r = 0; r < k; ++r) {
      idx_t best = -1;
      for (idx_t i = 0; i < ntotal; ++i) if (!used[(size_t)i] && (best < 0 || s[(size_t)i] > s[(size_t)best])) best = i;
      if (best < 0) { labels[r] = -1; sims[r] = -3.4e38f; continue; }
      used[(size_t)best] = 1; labels[r] = best; sims[r] = s[(size_t)best];
    }
    (void)nq;
  }
};
}  // namespace faiss
namespace D2FrontEnd {
struct VisualImageDesc {
  std::vector<float> image_desc; int drone_id = 0; long frame_id = 0;
  int sp_num = 0; int spLandmarkNum() const { return sp_num; }
};
struct VisualImageDescArray { std::vector<VisualImageDesc> images; long frame_id = 0; };
struct LoopDetector {
  std::map<int, long> index_to_frame_id;
  std::map<long, VisualImageDescArray> keyframe_database_arr;
  struct KF { int drone_id = 0; };
  std::map<long, KF> keyframe_database;
  std::mutex keyframe_database_mutex;
  int queryIndexFromDatabase(const VisualImageDesc& img_desc, faiss::IndexFlatIP& index, bool remote_db, double thres, int max_index,
                             double& similarity);
};
#include SPREF_GEN_DB_QUERY            /* loop_detector.cpp:300-350 */
}  // namespace D2FrontEnd

extern "C" SPREF_API int spref_db_query(const float* db, int ntotal, int dim, const float* q, int max_index, double thres, float* sim_out) {
  faiss::IndexFlatIP index(dim);
  if (ntotal > 0) index.add(ntotal, db);
  D2FrontEnd::LoopDetector ld;
  for (int i = 0; i < ntotal; ++i) { ld.index_to_frame_id[i] = i; ld.keyframe_database[i] = D2FrontEnd::LoopDetector::KF(); }
  D2FrontEnd::VisualImageDesc d; d.image_desc.assign(q, q + dim);
  double sim = -1.0;
  const int r = ld.queryIndexFromDatabase(d, index, false, thres, max_index, sim);
  if (sim_out) *sim_out = (float)sim;
  return r;
}

// ---- D2FeatureTracker::getMatchedPrevKeyframe + the view pairing of trackRemoteFrames ----------------------------------------------
namespace D2FrontEnd {
enum CameraConfig { STEREO_PINHOLE = 0, STEREO_FISHEYE = 1, PINHOLE_DEPTH = 2, FOURCORNER_FISHEYE = 3, MONOCULAR = 4 };
struct ShimParams2 { int camera_configuration = STEREO_PINHOLE; int netvlad_dims = 4096; double track_remote_netvlad_thres = 0.5; int self_id = 0; };
static ShimParams2* params = new ShimParams2();
struct D2FeatureTracker {
  typedef std::lock_guard<std::recursive_mutex> Guard;
  std::recursive_mutex keyframe_lock;
  std::vector<VisualImageDescArray> current_keyframes;
  bool getMatchedPrevKeyframe(const VisualImageDescArray& frame_a, VisualImageDescArray& prev, int& dir_a, int& dir_b);
};
#include SPREF_GEN_TRACKER_GATE        /* d2featuretracker.cpp:166-235 */
// trackRemoteFrames' FOURCORNER_FISHEYE view pairing (:270-284) with the locals it reads
static void remote_view_pairs(const VisualImageDescArray& frames, const VisualImageDescArray& prev, int dir_cur, int dir_prev,
                              std::vector<int>& dirs_cur_out, std::vector<int>& dirs_prev_out) {
#include SPREF_GEN_TRACKER_DIRS
  dirs_cur_out = dirs_cur; dirs_prev_out = dirs_prev;
}
}  // namespace D2FrontEnd

// remote: [n_views][dim] NetVLAD vectors of the remote frame; keyframes: [n_kf][n_views][dim] of the local keyframes (oldest first);
// sp_remote / sp_kf: SuperPoint landmark counts per view (the pairing skips empty views).  config: 0 stereo/mono branch, 3 quadcam branch.
// Returns 1 when a keyframe matched: *kf_idx, *dir_a, *dir_b and the (remote view, local view) pairs in pairs_cur/pairs_prev (*n_pairs).
extern "C" SPREF_API int spref_tracker_gate(int config, const float* remote, const int* sp_remote, const float* keyframes, const int* sp_kf,
                                            int n_kf, int n_views, int dim, double thres, int* kf_idx, int* dir_a, int* dir_b,
                                            int* pairs_cur, int* pairs_prev, int* n_pairs) {
  using namespace D2FrontEnd;
  params->camera_configuration = config; params->netvlad_dims = dim; params->track_remote_netvlad_thres = thres;
  D2FeatureTracker ft;
  for (int k = 0; k < n_kf; ++k) {
    VisualImageDescArray a; a.frame_id = 100 + k;
    for (int v = 0; v < n_views; ++v) {
      VisualImageDesc d; d.image_desc.assign(keyframes + ((size_t)k * n_views + v) * dim, keyframes + ((size_t)k * n_views + v + 1) * dim);
      d.sp_num = sp_kf ? sp_kf[k * n_views + v] : 1; a.images.push_back(d);
    }
    ft.current_keyframes.push_back(a);
  }
  VisualImageDescArray fr; fr.frame_id = 7;
  for (int v = 0; v < n_views; ++v) {
    VisualImageDesc d; d.image_desc.assign(remote + (size_t)v * dim, remote + (size_t)(v + 1) * dim);
    d.sp_num = sp_remote ? sp_remote[v] : 1; fr.images.push_back(d);
  }
  VisualImageDescArray prev; int da = 0, db = 0;
  *n_pairs = 0; *kf_idx = -1;
  if (!ft.getMatchedPrevKeyframe(fr, prev, da, db)) return 0;
  *kf_idx = (int)(prev.frame_id - 100); *dir_a = da; *dir_b = db;
  if (config == FOURCORNER_FISHEYE) {
    std::vector<int> dc, dp;
    remote_view_pairs(fr, prev, da, db, dc, dp);
    for (size_t i = 0; i < dc.size(); ++i) { pairs_cur[i] = dc[i]; pairs_prev[i] = dp[i]; }
    *n_pairs = (int)dc.size();
  }
  return 1;
}

// ---- matchLocalFeatures of loop_cam.cpp (stereo up/down cross-check matching) ---------------------------------------------------------
namespace D2FrontEnd {
struct ShimParams3 { int superpoint_dims = 256; };
namespace lc { static ShimParams3* params = new ShimParams3();
#include SPREF_GEN_LOOPCAM_MATCH       /* loop_cam.cpp:156-191 */
} }
extern "C" SPREF_API int spref_loopcam_match(const float* pts_up, const float* desc_up, int n_up, const float* pts_down, const float* desc_down,
                                             int n_down, int dim, int32_t* ids_up, int32_t* ids_down, float* pts_up_out, float* pts_down_out) {
  D2FrontEnd::lc::params->superpoint_dims = dim;
  std::vector<cv::Point2f> pu, pd;
  for (int i = 0; i < n_up; ++i) pu.push_back(cv::Point2f(pts_up[2 * i], pts_up[2 * i + 1]));
  for (int i = 0; i < n_down; ++i) pd.push_back(cv::Point2f(pts_down[2 * i], pts_down[2 * i + 1]));
  std::vector<float> du(desc_up, desc_up + (size_t)n_up * dim), dd(desc_down, desc_down + (size_t)n_down * dim);
  std::vector<int> iu, id;
  D2FrontEnd::lc::matchLocalFeatures(pu, pd, du, dd, iu, id);
  for (size_t i = 0; i < iu.size(); ++i) {
    ids_up[i] = iu[i]; ids_down[i] = id[i];
    pts_up_out[2 * i] = pu[i].x; pts_up_out[2 * i + 1] = pu[i].y; pts_down_out[2 * i] = pd[i].x; pts_down_out[2 * i + 1] = pd[i].y;
  }
  return (int)iu.size();
}
