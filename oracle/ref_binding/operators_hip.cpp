// oracle/ref_binding/operators_hip.cpp — TEST INFRASTRUCTURE. INTEGRATION.md section 1's table, executed: the reference's
// free functions bicubicRemapToSpherical (SR/render/ImageWarper.cpp:143-174), flattenLayersDeghostPreferBase,
// offsetHorizontalWrap, featherAlphaChannel (SR/util/CvUtil.cpp:93-115, 140-157, 224-260), saveFlowToFile and
// readFlowFromFile (CvUtil.cpp:159-199) with the reference's own signatures, each one a call into include/s360.h. `make -C oracle ref_binding` compiles the reference's ImageWarper.cpp /
// CvUtil.cpp with those four functions renamed (-D…=…_reference, per translation unit) and links this file in their
// place, so that the reference's own TestRenderStereoPanorama projects, blends, shifts and feathers through the library
// (_ops_hip / _ops_hip_emu binaries) and must still write the unmodified program's files.
#include <cstring>
#include <string>

#include "CvUtil.h"
#include "ImageWarper.h"
#include "PixFlowHip.h"  // globalS360Ctx()
#include "VrCamException.h"
#include "s360.h"

namespace {
using surround360::VrCamException;
void ck(int rc) {
  if (rc != S360_OK) throw VrCamException(s360_last_error(nullptr));
}
// "s360_camera is filled from the reference's Camera" (INTEGRATION.md): position, the rows of rotation, principal, focal,
// distortion, resolution, fovThreshold, type, group.find("side")
s360_camera to_pod(const surround360::Camera& c) {
  s360_camera p;
  std::memset(&p, 0, sizeof p);
  p.type = c.type == surround360::Camera::Type::FTHETA ? S360_CAM_FTHETA : S360_CAM_RECTILINEAR;
  p.is_side = c.group.find("side") != std::string::npos;
  for (int i = 0; i < 3; ++i) p.position[i] = c.position(i);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) p.rotation[i * 3 + j] = c.rotation(i, j);
  for (int i = 0; i < 2; ++i) {
    p.resolution[i] = c.resolution(i);
    p.principal[i] = c.principal(i);
    p.distortion[i] = c.distortion(i);
    p.focal[i] = c.focal(i);
  }
  p.fov_threshold = c.fovThreshold;
  std::strncpy(p.id, c.id.c_str(), sizeof(p.id) - 1);
  return p;
}
}  // namespace

namespace surround360 {
namespace warper {
void bicubicRemapToSpherical(Mat& dst, const Mat& src, const Camera& camera, const float leftAngle, const float rightAngle,
                             const float topAngle, const float bottomAngle) {
  const s360_camera cam = to_pod(camera);
  // cv::remap creates its destination with the SOURCE's type (ImageWarper.cpp:173): a pre-sized 3-channel dst becomes
  // 4-channel when the source is BGRA (the pole-removal result into bottomSpherical, TRSP:625-634)
  if (src.channels() == 4 && dst.channels() != 4) dst.create(dst.size(), CV_8UC4);
  ck(s360_bicubic_remap_to_spherical(optical_flow::globalS360Ctx(), dst.data, dst.cols, dst.rows, dst.channels(), src.data,
                                     src.cols, src.rows, src.channels(), &cam, leftAngle, rightAngle, topAngle, bottomAngle));
}
}  // namespace warper

namespace util {
Mat offsetHorizontalWrap(const Mat& srcImage, const float offset) {
  Mat out(srcImage.size(), srcImage.type());
  ck(s360_offset_horizontal_wrap(optical_flow::globalS360Ctx(), srcImage.data, srcImage.cols, srcImage.rows, srcImage.channels(),
                                 offset, out.data));
  return out;
}
Mat featherAlphaChannel(const Mat& src, int erodeSize) {
  Mat out(src.size(), CV_8UC4);
  ck(s360_feather_alpha_channel(optical_flow::globalS360Ctx(), src.data, src.cols, src.rows, erodeSize, out.data));
  return out;
}
Mat flattenLayersDeghostPreferBase(const Mat& bottomLayer, const Mat& topLayer) {
  Mat out(bottomLayer.size(), CV_8UC4);
  ck(s360_flatten_layers_deghost_prefer_base(optical_flow::globalS360Ctx(), bottomLayer.data, topLayer.data, bottomLayer.cols,
                                             bottomLayer.rows, out.data));
  return out;
}
// saveFlowToFile / readFlowFromFile (SR/util/CvUtil.cpp:159-199): the reference's .bin container through the library's
// byte-compatible writer and reader — the program's flow files and its --prev_frame_data_dir state go through them
void saveFlowToFile(const Mat& flow, const string& filename) {
  ck(s360_save_flow_to_file(filename.c_str(), (const float*)flow.data, flow.cols, flow.rows));
}
Mat readFlowFromFile(const string& filename) {
  int w = 0, h = 0;
  ck(s360_read_flow_from_file(filename.c_str(), nullptr, &w, &h, 0));  // size query
  Mat flow(Size(w, h), CV_32FC2);
  ck(s360_read_flow_from_file(filename.c_str(), (float*)flow.data, &w, &h, (size_t)w * h * 2));
  return flow;
}
}  // namespace util
}  // namespace surround360
