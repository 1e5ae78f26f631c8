// spref_api3.cpp -- third translation unit of oracle/_ref/libspref.so (test infrastructure; never linked into the product): the undistortion
// MAP GENERATION of the reference, compiled where it lies under /root/reference (line ranges written into SPREF_GEN_* include files by
// oracle/build_ref.py, nothing copied into this repository):
//   camodocal (vendored: camera_models/)   CataCamera::spaceToPlane  CataCamera.cc:495-515,  CataCamera::distortion  :617-633,
//                                          CylindricalCamera::liftProjective  CylindricalCamera.cc:207-220, the inverse-K assignments :144-147
//   FisheyeUndist::genOneUndistMap         d2common/include/d2common/fisheye_undistort.h:571-579 (virtual-camera form: lift, project, store)
//                                          and :627-638 (rotated-pinhole form)
// Stand-in double-precision Eigen types below carry the members those ranges touch, with Eigen 3.4's arithmetic
// (Quaternion * Vector3: uv = 2 (q.vec x v); v + w uv + q.vec x uv -- Quaternion.h _transformVector; Vector3d::norm = sqrt of the sum of squares).
#include <cmath>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>

#define SPREF_API __attribute__((visibility("default")))

namespace Eigen {
struct Vector2d;
struct Comma2 { double* v; int i; Comma2& operator,(double x) { v[i++] = x; return *this; } };
struct Vector2d {
  double v[2];
  Vector2d() : v{0, 0} {}
  Vector2d(double a, double b) : v{a, b} {}
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  Comma2 operator<<(double a) { v[0] = a; return Comma2{v, 1}; }
  Vector2d operator+(const Vector2d& o) const { return Vector2d(v[0] + o.v[0], v[1] + o.v[1]); }
};
struct Comma3 { double* v; int i; Comma3& operator,(double x) { v[i++] = x; return *this; } };
struct Vector3d {
  double v[3];
  Comma3 operator<<(double a) { v[0] = a; return Comma3{v, 1}; }
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};
struct Quaterniond {
  double w_, x_, y_, z_;
  Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
  Vector3d operator*(const Vector3d& p) const {
    double ux = y_ * p.v[2] - z_ * p.v[1], uy = z_ * p.v[0] - x_ * p.v[2], uz = x_ * p.v[1] - y_ * p.v[0];
    ux += ux; uy += uy; uz += uz;
    const double cx = y_ * uz - z_ * uy, cy = z_ * ux - x_ * uz, cz = x_ * uy - y_ * ux;
    return Vector3d((p.v[0] + w_ * ux) + cx, (p.v[1] + w_ * uy) + cy, (p.v[2] + w_ * uz) + cz);
  }
};
}  // namespace Eigen

namespace camodocal {
struct Camera {
  virtual ~Camera() {}
  virtual void liftProjective(const Eigen::Vector2d& p, Eigen::Vector3d& P) const = 0;
  virtual void spaceToPlane(const Eigen::Vector3d& P, Eigen::Vector2d& p) const = 0;
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
  void spaceToPlane(const Eigen::Vector3d& P, Eigen::Vector2d& p) const override;
  void distortion(const Eigen::Vector2d& p_u, Eigen::Vector2d& d_u) const;
};
#include SPREF_GEN_CATA_LIFT           /* CataCamera.cc:425-487  liftProjective: inverse K, 8-step recursive distortion removal, the MEI ray */
#include SPREF_GEN_CATA_SPACE          /* CataCamera.cc:495-515 */
#include SPREF_GEN_CATA_DIST           /* CataCamera.cc:617-633 */

class CylindricalCamera : public Camera {
 public:
  struct Parameters { double fx_, fy_, cx_, cy_; double fx() const { return fx_; } double fy() const { return fy_; } double cx() const { return cx_; } double cy() const { return cy_; } };
  Parameters mParameters;
  double m_inv_K11, m_inv_K13, m_inv_K22, m_inv_K23;
  // the constructor the reference calls (fisheye_undistort.h:489-493): ("cylindrical", w, h, fx, fy, cx, cy)
  CylindricalCamera(const std::string&, int, int, double fx, double fy, double cx, double cy) : mParameters{fx, fy, cx, cy} {
#include SPREF_GEN_CYL_INVK            /* CylindricalCamera.cc:144-147 */
  }
  void liftProjective(const Eigen::Vector2d& p, Eigen::Vector3d& P) const override;
  void spaceToPlane(const Eigen::Vector3d&, Eigen::Vector2d&) const override {}
};
typedef std::shared_ptr<CylindricalCamera> CylindricalCameraPtr;
#include SPREF_GEN_CYL_LIFT            /* CylindricalCamera.cc:207-220 */
}  // namespace camodocal

namespace cv {
struct Vec2f { float a, b; Vec2f(float x, float y) : a(x), b(y) {} };
struct Point { int x, y; Point(int x_, int y_) : x(x_), y(y_) {} };
struct MapMat {
  float* mx; float* my; int w;
  template <class T> struct Ref { float* px; float* py; void operator=(const Vec2f& v) { *px = v.a; *py = v.b; } };
  template <class T> Ref<T> at(const Point& p) { return Ref<T>{mx + (size_t)p.y * w + p.x, my + (size_t)p.y * w + p.x}; }
};
}  // namespace cv

#define DEG_TO_RAD (M_PI / 180.0)      /* fisheye_undistort.h:27 */

static std::shared_ptr<camodocal::CataCamera> make_cata(const double* c) {
  auto cam = std::make_shared<camodocal::CataCamera>();
  cam->mParameters = camodocal::CataCamera::Parameters{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]};
  cam->set_inverse_K();
  return cam;
}

extern "C" {
// CataCamera::liftProjective for n image points (what LoopCam::extractorImgDescDeepnet calls per keypoint, loop_cam.cpp:619-623): out[n][3]
SPREF_API void spref_cata_lift(const double* cam9, const float* pts, int n, double* out) {
  auto cam = make_cata(cam9);
  for (int i = 0; i < n; ++i) {
    Eigen::Vector3d P;
    cam->liftProjective(Eigen::Vector2d(pts[2 * i], pts[2 * i + 1]), P);
    out[3 * i] = P(0); out[3 * i + 1] = P(1); out[3 * i + 2] = P(2);
  }
}
// generateCylinderMap (fisheye_undistort.h:458-500): cylinderFov = fov * DEG_TO_RAD (:465), f_center = imgWidth / cylinderFov (:469),
// CylindricalCamera("cylindrical", imgWidth, imgHeight, f_center, f_center, imgWidth / 2, imgHeight / 2) (:489-493, unsigned divisions),
// then genOneUndistMap's loop.  cam9 = {xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0}.
SPREF_API void spref_gen_cylinder_map(const double* cam9, int width, int height, double fov, float* mapx, float* mapy) {
  const unsigned imgWidth = (unsigned)width, imgHeight = (unsigned)height;
  double cylinderFov = fov * DEG_TO_RAD;
  double f_center = imgWidth / cylinderFov;
  camodocal::CameraPtr p_cam = make_cata(cam9);
  camodocal::CameraPtr p_vcam = camodocal::CylindricalCameraPtr(new camodocal::CylindricalCamera(
      "cylindrical", imgWidth, imgHeight, f_center, f_center, imgWidth / 2, imgHeight / 2));
  cv::MapMat map{mapx, mapy, width};
#include SPREF_GEN_MAP_LOOP_VCAM       /* fisheye_undistort.h:571-579, up to the store; the bookkeeping `if` behind it is not part of the map */
            }
}
// genOneUndistMap(_id, p_cam, rotation, imgWidth, imgHeight, f_center) (fisheye_undistort.h:615-660): the rotated-pinhole form
SPREF_API void spref_gen_pinhole_map(const double* cam9, const double* q_wxyz, int width, int height, double f, float* mapx, float* mapy) {
  const unsigned imgWidth = (unsigned)width, imgHeight = (unsigned)height;
  const double f_center = f;
  camodocal::CameraPtr p_cam = make_cata(cam9);
  Eigen::Quaterniond rotation(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]);
  cv::MapMat map{mapx, mapy, width};
#include SPREF_GEN_MAP_LOOP_PINHOLE    /* fisheye_undistort.h:627-638 */
            }
}
}
