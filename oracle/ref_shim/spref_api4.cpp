// spref_api4.cpp -- fourth translation unit of oracle/_ref/libspref.so (test infrastructure; never linked into the product): the LK tracker's GLUE,
// the reference's own code compiled where it lies under /root/reference:
//   opticalflowTrackPyr (GPU form)   d2frontend/src/opticaltrack_utils.cpp:173-278  (half-image pre-filter and +-move_cols shift, forward track, the
//                                    reverse track from the shifted result, the 0.5 px forward/backward test, inBorder, reduceVector compaction)
//   inBorder                         d2frontend/src/opticaltrack_utils.cpp:35-41
//   LKImageInfo                      d2frontend/include/d2frontend/opticaltrack_utils.h:16-26
//   reduceVector                     d2frontend/include/d2frontend/utils.h:21-28
// The optical flow itself is OpenCV-CUDA (absent): cv::cuda::SparsePyrLKOpticalFlow::calc and buildImagePyramid are stand-ins that call the
// oracle's restatement (oracle/d2fe_oracle_lk.c) through callbacks the test installs -- what this pins is the reference's control flow around them.
#include <opencv2/opencv.hpp>
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#define SPREF_API __attribute__((visibility("default")))
typedef unsigned char uchar;
inline int cvRound(double v) { return (int)std::lrint(v); }      /* OpenCV: round half to even */

extern "C" {
typedef void (*spref_pyr_build_fn)(const uint8_t* img, int w, int h, int stride, int levels, uint8_t* pyr);
typedef int (*spref_pyr_layout_fn)(int w, int h, int levels, int* off, int* ws, int* hs);
typedef void (*spref_lk_calc_fn)(const uint8_t* prev_pyr, const uint8_t* next_pyr, int w, int h, int levels, const float* prev_pts, float* next_pts,
                                 int n, int win, int iters, uint8_t* status);
static spref_pyr_build_fn g_pyr_build = nullptr;
static spref_pyr_layout_fn g_pyr_layout = nullptr;
static spref_lk_calc_fn g_lk_calc = nullptr;
SPREF_API void spref_set_lk_backend(spref_pyr_build_fn a, spref_pyr_layout_fn b, spref_lk_calc_fn c) { g_pyr_build = a; g_pyr_layout = b; g_lk_calc = c; }
}

namespace cv {
template <class T> using Ptr = std::shared_ptr<T>;
namespace cuda {
// one class for what the reference keeps in GpuMats: an image (pyramid level 0 carries the oracle's packed pyramid), a point list, a status list
struct GpuMat {
  int rows = 0, cols = 0;
  std::shared_ptr<std::vector<uint8_t>> bytes;      // image or packed pyramid
  std::vector<cv::Point2f> pts;
  std::vector<uchar> st;
  GpuMat() {}
  GpuMat(const cv::Mat& m) : rows(m.rows), cols(m.cols), bytes(std::make_shared<std::vector<uint8_t>>((size_t)m.rows * m.cols)) {
    for (int y = 0; y < m.rows; ++y) std::copy(m.data + (size_t)y * m.step, m.data + (size_t)y * m.step + m.cols, bytes->data() + (size_t)y * m.cols);
  }
  GpuMat(const std::vector<cv::Point2f>& p) : pts(p) {}
  void download(std::vector<uchar>& o) const { o = st; }
  void download(std::vector<cv::Point2f>& o) const { o = pts; }
};
struct SparsePyrLKOpticalFlow {
  int win, levels, iters;
  static Ptr<SparsePyrLKOpticalFlow> create(cv::Size w, int maxLevel, int it, bool /*useInitialFlow*/) {
    auto p = std::make_shared<SparsePyrLKOpticalFlow>(); p->win = w.width; p->levels = maxLevel; p->iters = it; return p;
  }
  void calc(const std::vector<GpuMat>& prevPyr, const std::vector<GpuMat>& nextPyr, GpuMat& prevPts, GpuMat& nextPts, GpuMat& status) const {
    const int n = (int)prevPts.pts.size();
    status.st.assign((size_t)n, 0);
    if (!n) return;
    g_lk_calc(prevPyr[0].bytes->data(), nextPyr[0].bytes->data(), prevPyr[0].cols, prevPyr[0].rows, levels, &prevPts.pts[0].x, &nextPts.pts[0].x, n, win,
              iters, status.st.data());
  }
};
}  // namespace cuda
}  // namespace cv

namespace Eigen { struct Vector3d { double v[3]; }; }
namespace D2Common { typedef long LandmarkIdType; enum LandmarkType { SuperPointLandmark = 0, FlowLandmark = 1 };
namespace Utility { struct TicToc { double toc() { return 0; } }; } }
using D2Common::Utility::TicToc;
#define PYR_LEVEL 2                    /* opticaltrack_utils.h:10 */
#define WIN_SIZE cv::Size(21, 21)      /* opticaltrack_utils.cpp:25 */

namespace D2FrontEnd {
using D2Common::LandmarkIdType;
using LandmarkType = D2Common::LandmarkType;
enum TrackLRType { WHOLE_IMG_MATCH = 0, LEFT_RIGHT_IMG_MATCH, RIGHT_LEFT_IMG_MATCH };     /* d2featuretracker.h:18-22 */
struct ShimLKParams { double undistort_fov = 200.0; };
static ShimLKParams* params = new ShimLKParams();
#include SPREF_GEN_LK_INFO             /* opticaltrack_utils.h:16-26  LKImageInfo, LKImageInfoCPU / GPU */
#include SPREF_GEN_REDUCE_VECTOR       /* utils.h:21-28 */
// stand-in for buildImagePyramid (opticaltrack_utils.cpp:526-542, cv::cuda::pyrDown): the oracle's packed u8 pyramid rides in level 0
std::vector<cv::cuda::GpuMat> buildImagePyramid(const cv::cuda::GpuMat& img, int maxLevel_) {
  int off[16], ws[16], hs[16];
  const int total = g_pyr_layout(img.cols, img.rows, maxLevel_, off, ws, hs);
  cv::cuda::GpuMat l0; l0.rows = img.rows; l0.cols = img.cols;
  l0.bytes = std::make_shared<std::vector<uint8_t>>((size_t)total);
  g_pyr_build(img.bytes->data(), img.cols, img.rows, img.cols, maxLevel_, l0.bytes->data());
  return std::vector<cv::cuda::GpuMat>(1, l0);
}
#include SPREF_GEN_LK_INBORDER         /* opticaltrack_utils.cpp:35-41 */
#include SPREF_GEN_LK_TRACKPYR         /* opticaltrack_utils.cpp:173-278 */
}  // namespace D2FrontEnd

// opticalflowTrackPyr(cur_img, prev_lk, type) with prev_lk = {pyramid of prev_img, prev_pts, ids 0..n-1, local index 0..n-1, types}.
// Returns the number of surviving points; out_pts / out_ids hold them in order.
extern "C" SPREF_API int spref_lk_track_pyr(const uint8_t* prev_img, const uint8_t* cur_img, int w, int h, const float* prev_pts, int n, int type,
                                            double undistort_fov, float* out_pts, int32_t* out_ids) {
  using namespace D2FrontEnd;
  params->undistort_fov = undistort_fov;
  cv::Mat prev(h, w, CV_8U, const_cast<uint8_t*>(prev_img)), cur(h, w, CV_8U, const_cast<uint8_t*>(cur_img));
  LKImageInfoGPU lk;
  lk.pyr = buildImagePyramid(cv::cuda::GpuMat(prev), PYR_LEVEL);
  for (int i = 0; i < n; ++i) {
    lk.lk_pts.push_back(cv::Point2f(prev_pts[2 * i], prev_pts[2 * i + 1]));
    lk.lk_ids.push_back(i); lk.lk_local_index.push_back(i); lk.lk_types.push_back(D2Common::FlowLandmark);
  }
  const LKImageInfoGPU r = opticalflowTrackPyr(cur, lk, (TrackLRType)type);
  for (size_t i = 0; i < r.lk_pts.size(); ++i) { out_pts[2 * i] = r.lk_pts[i].x; out_pts[2 * i + 1] = r.lk_pts[i].y; out_ids[i] = (int32_t)r.lk_ids[i]; }
  return (int)r.lk_pts.size();
}
