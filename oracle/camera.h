// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// camera.h: restatement of the rig geometry model, Eigen/folly-free:
//   surround360_render/source/render/Camera.h:133-284, Camera.cpp:16-83,144-167
//   surround360_render/source/render/RigDescription.cpp:18-78, RigDescription.h:58-60
// Known answers: Camera::unitTest (Camera.cpp:291-410) is replayed by tests/test_cpu_oracle.py through
// oracle_capi.cpp; the reference's own Camera.cpp / RigDescription.cpp run inside oracle/_ref/TestRenderStereoPanorama,
// whose outputs equal this oracle's (tests/test_cpu_refprogram.py). Eigen itself is restated, here and in ref_shim.
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

namespace orc {

struct V2 { double x = 0, y = 0; };
struct V3 {
  double x = 0, y = 0, z = 0;
  V3() {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  V3 operator+(const V3& o) const { return V3(x + o.x, y + o.y, z + o.z); }
  V3 operator-(const V3& o) const { return V3(x - o.x, y - o.y, z - o.z); }
  V3 operator*(double s) const { return V3(x * s, y * s, z * s); }
  V3 operator-() const { return V3(-x, -y, -z); }
  double dot(const V3& o) const { return x * o.x + y * o.y + z * o.z; }
  V3 cross(const V3& o) const { return V3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
  double squaredNorm() const { return dot(*this); }
  double norm() const { return std::sqrt(squaredNorm()); }
};

struct Camera {
  enum Type { FTHETA = 0, RECTILINEAR = 1 };
  static constexpr double kNearInfinity = 1e6;  // Camera.cpp:14
  int type = FTHETA;
  V3 position;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major; rows = right, up, backward
  V2 resolution, principal, distortion, focal;
  double fovThreshold = -1;
  std::string id, group;

  V3 rowv(int r) const { return V3(R[r * 3], R[r * 3 + 1], R[r * 3 + 2]); }
  V3 right() const { return rowv(0); }
  V3 up() const { return rowv(1); }
  V3 backward() const { return rowv(2); }
  V3 forward() const { return -backward(); }

  // Camera.cpp:16-29: rows = (right, up, -forward), then re-unitarise through
  // Eigen::AngleAxis (matrix -> quaternion -> angle/axis -> matrix).
  void setRotation(const V3& fwd, const V3& upv, const V3& rightv) {
    double m[9] = {rightv.x, rightv.y, rightv.z, upv.x, upv.y, upv.z, -fwd.x, -fwd.y, -fwd.z};
    double q[4];  // x y z w
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
      t = std::sqrt(t + 1.0);
      q[3] = 0.5 * t;
      t = 0.5 / t;
      q[0] = (m[7] - m[5]) * t;
      q[1] = (m[2] - m[6]) * t;
      q[2] = (m[3] - m[1]) * t;
    } else {
      int i = 0;
      if (m[4] > m[0]) i = 1;
      if (m[8] > m[i * 4]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
      q[i] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
      q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
      q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    double angle, ax[3];
    if (n != 0) {
      angle = 2 * std::atan2(n, std::fabs(q[3]));
      if (q[3] < 0) n = -n;
      ax[0] = q[0] / n; ax[1] = q[1] / n; ax[2] = q[2] / n;
    } else {
      angle = 0; ax[0] = 1; ax[1] = 0; ax[2] = 0;
    }
    const double s = std::sin(angle), c = std::cos(angle);
    const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const double ca[3] = {(1 - c) * ax[0], (1 - c) * ax[1], (1 - c) * ax[2]};
    double tmp;
    tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
  }

  // Camera.cpp:144-167
  void setFov(double fov) { const double c = std::cos(fov); fovThreshold = c * std::fabs(c); }
  double getFov() const {
    return fovThreshold < 0 ? std::acos(-std::sqrt(-fovThreshold)) : std::acos(std::sqrt(fovThreshold));
  }
  void setDefaultFov() { fovThreshold = (type == FTHETA) ? -1 : 0; }
  bool isDefaultFov() const { return type == FTHETA ? fovThreshold == -1 : fovThreshold == 0; }

  // Camera.h:221-227
  double distortFactor(double r2) const { return 1 + r2 * (distortion.x + r2 * distortion.y); }
  double distort(double r) const { return distortFactor(r * r) * r; }
  // Camera.h:229-248
  double undistort(double d) const {
    if (distortion.x == 0 && distortion.y == 0) return d;
    double r0 = d;
    const double smidgen = 1.0 / kNearInfinity;
    for (int step = 0; step < 10; ++step) {
      const double d0 = distort(r0);
      if (std::fabs(d0 - d) < smidgen) break;
      const double r1 = r0 + smidgen;
      const double d1 = distort(r1);
      const double derivative = (d1 - d0) / smidgen;
      r0 -= (d0 - d) / derivative;
    }
    return r0;
  }

  // Camera.h:250-261
  V2 cameraToSensor(const V3& cam) const {
    V2 s;
    if (type == FTHETA) {
      const double norm = std::sqrt(cam.x * cam.x + cam.y * cam.y);
      const double r = std::atan2(norm, -cam.z);
      const double f = distort(r) / norm;
      s.x = f * cam.x; s.y = f * cam.y;
    } else {
      const double px = cam.x / -cam.z, py = cam.y / -cam.z;
      const double f = distortFactor(px * px + py * py);
      s.x = f * px; s.y = f * py;
    }
    return s;
  }
  // Camera.h:133-140
  V2 pixel(const V3& rig) const {
    const V3 d = rig - position;
    const V3 cam(R[0] * d.x + R[1] * d.y + R[2] * d.z, R[3] * d.x + R[4] * d.y + R[5] * d.z,
                 R[6] * d.x + R[7] * d.y + R[8] * d.z);
    const V2 s = cameraToSensor(cam);
    V2 p;
    p.x = focal.x * s.x + principal.x;
    p.y = focal.y * s.y + principal.y;
    return p;
  }
  // Camera.h:264-284
  V3 sensorToCamera(const V2& sensor) const {
    const double sq = sensor.x * sensor.x + sensor.y * sensor.y;
    if (sq == 0) return V3(0, 0, -1);
    const double norm = std::sqrt(sq);
    const double r = undistort(norm);
    const double angle = (type == FTHETA) ? r : std::atan(r);
    const double f = std::sin(angle) / norm;
    return V3(f * sensor.x, f * sensor.y, -std::cos(angle));
  }
  // Camera.h:143-150: direction of the ray through a pixel (origin = position)
  V3 rigDirection(const V2& pix) const {
    V2 sensor;
    sensor.x = (pix.x - principal.x) / focal.x;
    sensor.y = (pix.y - principal.y) / focal.y;
    const V3 u = sensorToCamera(sensor);
    // rotation.transpose() * unit
    return V3(R[0] * u.x + R[3] * u.y + R[6] * u.z, R[1] * u.x + R[4] * u.y + R[7] * u.z,
              R[2] * u.x + R[5] * u.y + R[8] * u.z);
  }
  V3 rigNearInfinity(const V2& pix) const { return position + rigDirection(pix) * kNearInfinity; }

  // Camera.h:157-180
  bool isBehind(const V3& rig) const { return backward().dot(rig - position) >= 0; }
  bool isOutsideFov(const V3& rig) const {
    if (fovThreshold == -1) return false;
    if (fovThreshold == 0) return isBehind(rig);
    const V3 v = rig - position;
    const double dot = -backward().dot(v);
    return dot * std::fabs(dot) <= fovThreshold * v.squaredNorm();
  }
  bool sees(const V3& rig) const {
    if (isOutsideFov(rig)) return false;
    const V2 p = pixel(rig);
    return 0 <= p.x && p.x < resolution.x && 0 <= p.y && p.y < resolution.y;
  }
};

// TestRenderStereoPanorama.cpp:75-88
static inline float approximateFov(const Camera& cam, bool vertical) {
  V2 a = cam.principal, b = cam.principal;
  if (vertical) { a.y = 0; b.y = cam.resolution.y; }
  else { a.x = 0; b.x = cam.resolution.x; }
  const V3 f = cam.forward();
  // ParametrizedLine normalises nothing: direction() is rotation^T * unit (already unit length)
  return (float)std::acos(std::max(cam.rigDirection(a).dot(f), cam.rigDirection(b).dot(f)));
}
// TestRenderStereoPanorama.cpp:91-97
static inline float approximateFov(const std::vector<Camera>& rig, bool vertical) {
  float result = 0;
  for (const Camera& c : rig) result = std::max(result, approximateFov(c, vertical));
  return result;
}

// Camera.h:201-212
static inline float approximateUsablePixelsRadius(const Camera& camera) {
  const double fov = camera.getFov();
  const double kStep = 2 * M_PI / 10.0;
  double result = std::sqrt(camera.resolution.x * camera.resolution.x + camera.resolution.y * camera.resolution.y);
  for (double a = 0; a < 2 * M_PI; a += kStep) {
    const V3 ortho = camera.right() * std::cos(a) + camera.up() * std::sin(a);
    const V3 direction = camera.forward() * std::cos(fov) + ortho * std::sin(fov);
    const V2 pixel = camera.pixel(camera.position + direction);
    const double dx = pixel.x - camera.resolution.x / 2.0, dy = pixel.y - camera.resolution.y / 2.0;
    result = std::min(result, std::sqrt(dx * dx + dy * dy));
  }
  return (float)result;
}

struct RigDescription {
  std::vector<Camera> rig, rigSideOnly;
  // RigDescription.cpp:18-31 (construction from an already-parsed camera list)
  void finalize() {
    rigSideOnly.clear();
    for (const Camera& c : rig)
      if (c.group.find("side") != std::string::npos) rigSideOnly.push_back(c);
  }
  // RigDescription.h:58-60: distance of the optical axis to the rig centre
  static double distCamAxisToRigCenter(const Camera& c) {
    const V3 d = c.rigDirection(c.principal);  // unit
    const V3 diff = V3(0, 0, 0) - c.position;
    const V3 perp = diff - d * d.dot(diff);  // Eigen ParametrizedLine::distance
    return perp.norm();
  }
  // RigDescription.cpp:33-47
  const Camera& findCameraByDirection(const V3& dir, double maxDist = 1.0) const {
    const Camera* best = nullptr;
    for (const Camera& c : rig)
      if (best == nullptr || best->forward().dot(dir) < c.forward().dot(dir))
        if (distCamAxisToRigCenter(c) <= maxDist) best = &c;
    return *best;
  }
  // RigDescription.cpp:46-54: the camera whose optical axis passes farthest from the rig centre (secondary bottom)
  const Camera& findLargestDistCamAxisToRigCenter() const {
    const Camera* best = &rig.back();
    for (const Camera& c : rig)
      if (distCamAxisToRigCenter(c) > distCamAxisToRigCenter(*best)) best = &c;
    return *best;
  }
  float getRingRadius() const { return (float)rigSideOnly[0].position.norm(); }
};

}  // namespace orc
