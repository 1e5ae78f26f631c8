// extern "C" view of the host-side lift helpers of include/d2fe.hpp (A8: what an adapter without camodocal uses in place of
// camera->liftProjective, loop_cam.cpp:619-623), for tests/test_ref_pin.py: compiled with g++ at test time, no GPU and no link against the library.
#include "d2fe.hpp"

extern "C" void lift_mei(const double* c, const float* pts, int n, double* out) {
  d2fe_mei_camera cam{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]};
  for (int i = 0; i < n; ++i) {
    const D2FrontEnd::Vec3d P = D2FrontEnd::liftProjectiveMEI(cam, D2FrontEnd::Point2f(pts[2 * i], pts[2 * i + 1]));
    out[3 * i] = P.x; out[3 * i + 1] = P.y; out[3 * i + 2] = P.z;
  }
}
extern "C" void lift_cyl(double fx, double fy, double cx, double cy, const float* pts, int n, double* out) {
  for (int i = 0; i < n; ++i) {
    const D2FrontEnd::Vec3d P = D2FrontEnd::liftProjectiveCylindrical(fx, fy, cx, cy, D2FrontEnd::Point2f(pts[2 * i], pts[2 * i + 1]));
    out[3 * i] = P.x; out[3 * i + 1] = P.y; out[3 * i + 2] = P.z;
  }
}
