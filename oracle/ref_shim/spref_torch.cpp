// spref_torch.cpp -- oracle/_ref/libspref_torch.so (test infrastructure; never linked into the product): the reference's variant-A descriptor
// sampling, computeDescriptors (d2frontend/src/CNN/superpoint_common.cpp:42-99), compiled where it lies against the REAL libtorch of this image
// (torch::grid_sampler / norm / div are ATen's own kernels: the library the reference links, here at the image's version) and the stand-in
// OpenCV / Eigen headers of oracle/ref_shim.  Built only when torch's C++ headers are present (oracle/build_ref.py); tests skip otherwise.
#define USE_CUDA 1
#include <iostream>
#include <opencv2/opencv.hpp>
#include <Eigen/Eigen>
#include <torch/csrc/api/include/torch/types.h>
#include <ATen/ATen.h>
#include <spdlog/spdlog.h>

#define SPREF_API __attribute__((visibility("default")))
namespace D2Common { namespace Utility { struct TicToc { double toc() { return 0.0; } }; } }
using D2Common::Utility::TicToc;
namespace D2FrontEnd {
struct ShimParamsT { bool enable_perf_output = false; };
static ShimParamsT* params = new ShimParamsT();
#include SPREF_GEN_COMPUTE_DESC        /* superpoint_common.cpp:42-99 */
}  // namespace D2FrontEnd

// desc_chw: the network's raw descriptor map [dim][hc][wc] (the `desc` output of the ONNX graph is channel-normalised already: pass that);
// kps [n][2] (x, y); comp_T [dim][pca] column-major (= the reference's pca_comp_T, transpose of the CSV) or NULL; out [n][pca or dim].
extern "C" SPREF_API int spref_compute_descriptors(const float* desc_chw, int dim, int hc, int wc, const float* kps, int n, int width, int height,
                                                   const float* comp_T, const float* mean, int pca, float* out) {
  torch::NoGradGuard guard;
  at::Tensor mDesc = at::from_blob(const_cast<float*>(desc_chw), {1, dim, hc, wc}, torch::kFloat);
  at::Tensor mProb;
  std::vector<cv::Point2f> keypoints;
  for (int i = 0; i < n; ++i) keypoints.push_back(cv::Point2f(kps[2 * i], kps[2 * i + 1]));
  Eigen::MatrixXf c; Eigen::RowVectorXf m;
  if (comp_T && pca > 0) { c.r = dim; c.c = pca; c.v.assign(comp_T, comp_T + (size_t)dim * pca); m.v.assign(mean, mean + dim); }
  std::vector<float> d;
  D2FrontEnd::computeDescriptors(mProb, mDesc, keypoints, d, width, height, c, m);
  std::copy(d.begin(), d.end(), out);
  return (int)d.size();
}
