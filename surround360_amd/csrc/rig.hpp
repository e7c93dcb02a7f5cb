// rig.hpp — host-side rig geometry of libs360: the Camera model, rig JSON loader and the
// derived panorama geometry. Stays on the CPU by design (north_star: "the C++ host keeps the
// Camera/RigDescription loader"); double precision like the reference's Eigen code.
//   Camera:          SR/render/Camera.h:133-284, Camera.cpp:16-83, 144-167
//   RigDescription:  SR/render/RigDescription.cpp:18-78, RigDescription.h:58-60
//   geometry:        SR/test/TestRenderStereoPanorama.cpp:75-97, 153-173, 309-348, 454-481
#pragma once
#include <string>
#include <vector>

#include "../../include/s360.h"

namespace s360 {

// ---- camera model on s360_camera (POD) ----------------------------------------------------
void camera_set_rotation(s360_camera* c, const double fwd[3], const double up[3], const double right[3]);
void camera_set_fov(s360_camera* c, double fov);
void camera_set_default_fov(s360_camera* c);
double camera_get_fov(const s360_camera* c);
void camera_pixel(const s360_camera* c, const double rig[3], double out[2]);
// direction (rig space, unit) of the ray through a pixel: Camera::rig(pixel).direction()
void camera_rig_direction(const s360_camera* c, const double pix[2], double out[3]);
inline void camera_forward(const s360_camera* c, double f[3]) {
  f[0] = -c->rotation[6]; f[1] = -c->rotation[7]; f[2] = -c->rotation[8];
}
float approximate_fov(const s360_camera* c, bool vertical);
float approximate_fov(const std::vector<s360_camera>& rig, bool vertical);

// ---- rig -------------------------------------------------------------------------------------
struct Rig {
  std::vector<s360_camera> all, side;
  void finalize();  // split "side" group
  int find_by_direction(const double dir[3], double max_axis_dist = 1.0) const;  // index into all, -1 if none
  float ring_radius() const;
  int find_largest_axis_dist() const;  // RigDescription::findLargestDistCamAxisToRigCenter (secondary bottom camera)
};
// Camera::approximateUsablePixelsRadius (SR/render/Camera.h:201-212)
float approximate_usable_pixels_radius(const s360_camera* c);
// Parses the rig JSON text (RIG_JSON.md). Throws Error on malformed input.
std::vector<s360_camera> parse_rig_json(const std::string& text);

// ---- derived geometry -------------------------------------------------------------------------
struct PoleRamp { float poleCameraRadius, phiRampStart, phiMid, phiRampEnd; };
s360_geometry derive_geometry(const Rig& rig, const s360_params& p);
void side_camera_angles(const s360_geometry& g, int camIdx, int numCams, float* l, float* r, float* t, float* b);
PoleRamp pole_ramp(const Rig& rig);

}  // namespace s360
