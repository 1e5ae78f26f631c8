// Stand-in for d2frontend/d2frontend_params.h (test infrastructure): the fields the extracted line ranges read.
#pragma once
namespace D2FrontEnd {
struct D2FrontendParams {
  bool enable_perf_output = false;
  int width_undistort = 800;
  double undistort_fov = 200.0;
  int superpoint_dims = 256;
};
extern D2FrontendParams* params;
}
