// rig.cpp — see rig.hpp. Host-only, double precision.
#include "rig.hpp"

#include "json_mini.hpp"

#include <cctype>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include "core.hpp"

namespace s360 {

// ------------------------------------------------------------------------------------------
// Camera::setRotation (Camera.cpp:16-29): rows (right, up, -forward), then re-unitarised the way
// Eigen::AngleAxis(matrix).toRotationMatrix() does it (matrix -> quaternion -> angle/axis -> matrix).
void camera_set_rotation(s360_camera* c, const double fwd[3], const double up[3], const double right[3]) {
  const double m[9] = {right[0], right[1], right[2], up[0], up[1], up[2], -fwd[0], -fwd[1], -fwd[2]};
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
  const double s = std::sin(angle), co = std::cos(angle);
  const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
  const double ca[3] = {(1 - co) * ax[0], (1 - co) * ax[1], (1 - co) * ax[2]};
  double* R = c->rotation;
  double tmp;
  tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
  tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
  tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
  R[0] = ca[0] * ax[0] + co; R[4] = ca[1] * ax[1] + co; R[8] = ca[2] * ax[2] + co;
}
void camera_set_fov(s360_camera* c, double fov) {  // Camera.cpp:144-148
  const double cf = std::cos(fov);
  c->fov_threshold = cf * std::fabs(cf);
}
void camera_set_default_fov(s360_camera* c) { c->fov_threshold = (c->type == S360_CAM_FTHETA) ? -1 : 0; }
double camera_get_fov(const s360_camera* c) {  // Camera.cpp:150-154
  return c->fov_threshold < 0 ? std::acos(-std::sqrt(-c->fov_threshold)) : std::acos(std::sqrt(c->fov_threshold));
}
static inline double distort_factor(const s360_camera* c, double r2) {
  return 1 + r2 * (c->distortion[0] + r2 * c->distortion[1]);
}
static inline double distort(const s360_camera* c, double r) { return distort_factor(c, r * r) * r; }
static double undistort(const s360_camera* c, double d) {  // Camera.h:229-248
  if (c->distortion[0] == 0 && c->distortion[1] == 0) return d;
  double r0 = d;
  const double smidgen = 1.0 / 1e6;
  for (int step = 0; step < 10; ++step) {
    const double d0 = distort(c, r0);
    if (std::fabs(d0 - d) < smidgen) break;
    const double r1 = r0 + smidgen;
    const double d1 = distort(c, r1);
    const double derivative = (d1 - d0) / smidgen;
    r0 -= (d0 - d) / derivative;
  }
  return r0;
}
void camera_pixel(const s360_camera* c, const double rig[3], double out[2]) {  // Camera.h:133-140, 250-261
  const double* R = c->rotation;
  const double d[3] = {rig[0] - c->position[0], rig[1] - c->position[1], rig[2] - c->position[2]};
  const double cx = R[0] * d[0] + R[1] * d[1] + R[2] * d[2];
  const double cy = R[3] * d[0] + R[4] * d[1] + R[5] * d[2];
  const double cz = R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
  double sx, sy;
  if (c->type == S360_CAM_FTHETA) {
    const double norm = std::sqrt(cx * cx + cy * cy);
    const double r = std::atan2(norm, -cz);
    const double f = distort(c, r) / norm;
    sx = f * cx; sy = f * cy;
  } else {
    const double px = cx / -cz, py = cy / -cz;
    const double f = distort_factor(c, px * px + py * py);
    sx = f * px; sy = f * py;
  }
  out[0] = c->focal[0] * sx + c->principal[0];
  out[1] = c->focal[1] * sy + c->principal[1];
}
void camera_rig_direction(const s360_camera* c, const double pix[2], double out[3]) {  // Camera.h:143-150, 264-284
  const double sx = (pix[0] - c->principal[0]) / c->focal[0];
  const double sy = (pix[1] - c->principal[1]) / c->focal[1];
  const double sq = sx * sx + sy * sy;
  double u[3];
  if (sq == 0) {
    u[0] = 0; u[1] = 0; u[2] = -1;
  } else {
    const double norm = std::sqrt(sq);
    const double r = undistort(c, norm);
    const double angle = (c->type == S360_CAM_FTHETA) ? r : std::atan(r);
    const double f = std::sin(angle) / norm;
    u[0] = f * sx; u[1] = f * sy; u[2] = -std::cos(angle);
  }
  const double* R = c->rotation;
  out[0] = R[0] * u[0] + R[3] * u[1] + R[6] * u[2];
  out[1] = R[1] * u[0] + R[4] * u[1] + R[7] * u[2];
  out[2] = R[2] * u[0] + R[5] * u[1] + R[8] * u[2];
}
float approximate_fov(const s360_camera* c, bool vertical) {  // TRSP:75-88
  double a[2] = {c->principal[0], c->principal[1]}, b[2] = {c->principal[0], c->principal[1]};
  if (vertical) { a[1] = 0; b[1] = c->resolution[1]; }
  else { a[0] = 0; b[0] = c->resolution[0]; }
  double f[3], da[3], db[3];
  camera_forward(c, f);
  camera_rig_direction(c, a, da);
  camera_rig_direction(c, b, db);
  const double dota = da[0] * f[0] + da[1] * f[1] + da[2] * f[2];
  const double dotb = db[0] * f[0] + db[1] * f[1] + db[2] * f[2];
  return (float)std::acos(std::max(dota, dotb));
}
float approximate_fov(const std::vector<s360_camera>& rig, bool vertical) {  // TRSP:91-97
  float r = 0;
  for (const s360_camera& c : rig) r = std::max(r, approximate_fov(&c, vertical));
  return r;
}

// ------------------------------------------------------------------------------------------
void Rig::finalize() {
  side.clear();
  for (const s360_camera& c : all)
    if (c.is_side) side.push_back(c);
}
static double axis_dist_to_centre(const s360_camera& c) {  // RigDescription.h:58-60
  double d[3];
  camera_rig_direction(&c, c.principal, d);
  const double diff[3] = {-c.position[0], -c.position[1], -c.position[2]};
  const double t = d[0] * diff[0] + d[1] * diff[1] + d[2] * diff[2];
  const double p[3] = {diff[0] - d[0] * t, diff[1] - d[1] * t, diff[2] - d[2] * t};
  return std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
}
int Rig::find_by_direction(const double dir[3], double maxd) const {  // RigDescription.cpp:33-47
  int best = -1;
  auto fdot = [&](const s360_camera& c) {
    double f[3];
    camera_forward(&c, f);
    return f[0] * dir[0] + f[1] * dir[1] + f[2] * dir[2];
  };
  for (size_t i = 0; i < all.size(); ++i)
    if (best < 0 || fdot(all[best]) < fdot(all[i]))
      if (axis_dist_to_centre(all[i]) <= maxd) best = (int)i;
  return best;
}
int Rig::find_largest_axis_dist() const {  // RigDescription.cpp:46-54
  int best = (int)all.size() - 1;
  for (size_t i = 0; i < all.size(); ++i)
    if (axis_dist_to_centre(all[i]) > axis_dist_to_centre(all[best])) best = (int)i;
  return best;
}
float approximate_usable_pixels_radius(const s360_camera* c) {  // Camera.h:201-212
  const double fov = camera_get_fov(c);
  const double kStep = 2 * M_PI / 10.0;
  double result = std::sqrt(c->resolution[0] * c->resolution[0] + c->resolution[1] * c->resolution[1]);
  double fwd[3];
  camera_forward(c, fwd);
  const double* right = c->rotation;
  const double* up = c->rotation + 3;
  for (double a = 0; a < 2 * M_PI; a += kStep) {
    double ortho[3], p[3], pix[2];
    for (int i = 0; i < 3; ++i) ortho[i] = right[i] * std::cos(a) + up[i] * std::sin(a);
    for (int i = 0; i < 3; ++i) p[i] = c->position[i] + (fwd[i] * std::cos(fov) + ortho[i] * std::sin(fov));
    camera_pixel(c, p, pix);
    const double dx = pix[0] - c->resolution[0] / 2.0, dy = pix[1] - c->resolution[1] / 2.0;
    result = std::min(result, std::sqrt(dx * dx + dy * dy));
  }
  return (float)result;
}
float Rig::ring_radius() const {
  const double* p = side[0].position;
  return (float)std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
}

namespace {
void vec(const JV& o, const char* key, int n, double* out) {
  const JV* a = o.get(key);
  if (!a || a->t != JV::ARR || (int)a->arr.size() != n) throw Error(S360_ERR_IO, std::string("rig json: bad vector ") + key);
  for (int i = 0; i < n; ++i) out[i] = a->arr[i].num;
}
}  // namespace

std::vector<s360_camera> parse_rig_json(const std::string& text) {
  JP p{text.data(), text.data() + text.size()};
  JV root = p.value();
  const JV* cams = root.get("cameras");
  if (!cams || cams->t != JV::ARR) throw Error(S360_ERR_IO, "rig json: no \"cameras\" array");
  std::vector<s360_camera> out;
  for (const JV& j : cams->arr) {  // Camera(const dynamic& json), Camera.cpp:44-83
    s360_camera c;
    std::memset(&c, 0, sizeof(c));
    const JV* ver = j.get("version");
    if (!ver || ver->num < 1.0) throw Error(S360_ERR_IO, "rig json: camera version < 1");
    const JV* id = j.get("id");
    if (id) std::strncpy(c.id, id->str.c_str(), sizeof(c.id) - 1);
    const JV* ty = j.get("type");
    if (!ty) throw Error(S360_ERR_IO, "rig json: camera without type");
    if (ty->str == "FTHETA") c.type = S360_CAM_FTHETA;
    else if (ty->str == "RECTILINEAR") c.type = S360_CAM_RECTILINEAR;
    else throw Error(S360_ERR_IO, "rig json: unknown camera type " + ty->str);
    double fwd[3], up[3], right[3];
    vec(j, "origin", 3, c.position);
    vec(j, "forward", 3, fwd);
    vec(j, "up", 3, up);
    vec(j, "right", 3, right);
    camera_set_rotation(&c, fwd, up, right);
    vec(j, "resolution", 2, c.resolution);
    if (j.get("principal")) vec(j, "principal", 2, c.principal);
    else { c.principal[0] = c.resolution[0] / 2; c.principal[1] = c.resolution[1] / 2; }
    if (j.get("distortion")) vec(j, "distortion", 2, c.distortion);
    if (j.get("fov")) camera_set_fov(&c, j.get("fov")->num);
    else camera_set_default_fov(&c);
    vec(j, "focal", 2, c.focal);
    const JV* grp = j.get("group");
    c.is_side = (grp && grp->str.find("side") != std::string::npos) ? 1 : 0;
    out.push_back(c);
  }
  return out;
}

// ------------------------------------------------------------------------------------------
s360_geometry derive_geometry(const Rig& rig, const s360_params& P) {
  s360_geometry g;
  std::memset(&g, 0, sizeof(g));
  const int numCams = (int)rig.side.size();
  g.h_radians = 2 * approximate_fov(rig.side, false);  // TRSP:155-156
  g.v_radians = 2 * approximate_fov(rig.side, true);
  g.cam_image_height = int(P.eqr_height * g.v_radians / M_PI);  // TRSP:159-162
  g.cam_image_width = int(P.eqr_width * g.h_radians / (2 * M_PI));
  const double fovHorizontal = 2 * approximate_fov(rig.side, false) * (180 / M_PI);  // TRSP:781-782
  const float camFovHorizontalDegrees = (float)fovHorizontal;
  g.fov_horizontal_radians = (float)(camFovHorizontalDegrees * M_PI / 180.0f);  // toRadians(float), TRSP:309
  const float overlapAngleDegrees = (float)((camFovHorizontalDegrees * float(numCams) - 360.0) / float(numCams));
  g.overlap_image_width = int(float(g.cam_image_width) * (overlapAngleDegrees / camFovHorizontalDegrees));
  g.num_novel_views = g.cam_image_width - g.overlap_image_width;
  const float cameraRingRadius = rig.ring_radius();
  const float v = atanf((float)(P.zero_parallax_dist / (P.interpupilary_dist / 2.0f)));  // TRSP:340-348
  const float psi = asinf((float)(sinf(v) * (P.interpupilary_dist / 2.0f) / cameraRingRadius));
  g.verge_at_infinity_slab_displacement = psi * (float(g.cam_image_width) / g.fov_horizontal_radians);
  const float theta = (float)(-M_PI / 2.0f + v + psi);
  g.zero_parallax_novel_view_shift_pixels = (float)(float(P.eqr_width) * (theta / (2.0f * M_PI)));
  const double up[3] = {0, 0, 1}, down[3] = {0, 0, -1};
  const int ti = rig.find_by_direction(up), bi = rig.find_by_direction(down);
  g.top_rows = ti >= 0 ? int(P.eqr_height * camera_get_fov(&rig.all[ti]) / M_PI) : 0;  // TRSP:656-659
  g.bottom_rows = bi >= 0 ? int(P.eqr_height * camera_get_fov(&rig.all[bi]) / M_PI) : 0;
  const bool resize = P.final_eqr_width != 0 && P.final_eqr_height != 0 && P.final_eqr_width != P.eqr_width &&
                      P.final_eqr_height != P.eqr_height / 2;  // TRSP:938-941
  g.out_width = resize ? P.final_eqr_width : P.eqr_width;
  g.out_height = resize ? 2 * (P.final_eqr_height / 2) : 2 * P.eqr_height;
  return g;
}
void side_camera_angles(const s360_geometry& g, int camIdx, int numCams, float* l, float* r, float* t, float* b) {
  const float direction = (float)(-float(camIdx) / float(numCams) * 2.0f * M_PI);  // TRSP:163-173
  *l = direction + g.h_radians / 2;
  *r = direction - g.h_radians / 2;
  *t = g.v_radians / 2;
  *b = -g.v_radians / 2;
}
PoleRamp pole_ramp(const Rig& rig) {  // TRSP:454-481
  const double down[3] = {0, 0, -1};
  const int bi = rig.find_by_direction(down);
  float poleCameraRadius = (float)camera_get_fov(&rig.all[bi]);
  float sideCameraRadius = approximate_fov(rig.side, true);
  float poleCameraCropRadius =
      (float)(0.5f * (M_PI / 2 - sideCameraRadius) + 0.5f * (std::min(float(M_PI / 2), poleCameraRadius)));
  poleCameraCropRadius = (float)(poleCameraCropRadius * (180 / M_PI));
  poleCameraRadius = (float)(poleCameraRadius * (180 / M_PI));
  sideCameraRadius = (float)(sideCameraRadius * (180 / M_PI));
  const float kRampFrac = 1.0f;
  const float phiFromPole = poleCameraCropRadius;
  const float phiFromSide = 90.0f - sideCameraRadius;
  PoleRamp r;
  r.poleCameraRadius = poleCameraRadius;
  r.phiMid = (phiFromPole + phiFromSide) / 2.0f;
  const float phiDiff = fabsf(phiFromPole - phiFromSide);
  r.phiRampStart = r.phiMid - kRampFrac * phiDiff / 2.0f;
  r.phiRampEnd = r.phiMid + kRampFrac * phiDiff / 2.0f;
  return r;
}

}  // namespace s360
