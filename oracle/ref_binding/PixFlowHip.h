// oracle/ref_binding/PixFlowHip.h — TEST INFRASTRUCTURE. INTEGRATION.md section 1 as a file that compiles: the subclass of the
// reference's OpticalFlowInterface (SR/optical_flow/OpticalFlowInterface.h:34-41) a maintainer adds to run PixFlow on
// the GPU through include/s360.h. Compiled into the reference's OWN TestRenderStereoPanorama program by
// `make -C oracle ref_binding` (the program's sources where they lie under /root/reference, this header and the two extra
// factory names of OpticalFlowFactory.h next to it injected from outside), so that
//   TestRenderStereoPanorama --side_flow_alg pixflow_low_hip --polar_flow_alg pixflow_low_hip
// runs the reference's frame with every flow computed by libs360 — and must write the files the unmodified program
// writes (tests/test_cpu_refprogram.py).
#pragma once
#include <cstring>
#include <mutex>
#include <string>

#include "OpticalFlowInterface.h"
#include "VrCamException.h"
#include "s360.h"

namespace surround360 {
namespace optical_flow {

// One context for the process: flows do not depend on the rig, so any rig with a side camera creates it. Its entry
// points lock it, the reference's 14 + 4 flow threads may share it (include/s360.h, "Thread safety").
inline s360_ctx* globalS360Ctx() {
  static s360_ctx* ctx = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    s360_camera cam;
    const double origin[3] = {0, 0, 0}, forward[3] = {1, 0, 0}, up[3] = {0, 0, 1}, right[3] = {0, -1, 0}, res[2] = {64, 64},
                 focal[2] = {64, 64};
    if (s360_camera_init(&cam, S360_CAM_RECTILINEAR, origin, forward, up, right, res, nullptr, nullptr, focal, nullptr, "side camera", "cam0") != S360_OK)
      throw VrCamException(std::string("s360_camera_init: ") + s360_last_error(nullptr));
    s360_params P = {};
    P.interpupilary_dist = 6.4; P.zero_parallax_dist = 10000; P.eqr_width = 256; P.eqr_height = 128;
    std::strncpy(P.side_flow_alg, "pixflow_low", sizeof(P.side_flow_alg) - 1);
    std::strncpy(P.polar_flow_alg, "pixflow_low", sizeof(P.polar_flow_alg) - 1);
    if (s360_create(&ctx, /*device*/ 0, &cam, 1, &P) != S360_OK)
      throw VrCamException(std::string("s360_create: ") + s360_last_error(nullptr));
  });
  return ctx;
}

struct PixFlowHip : public OpticalFlowInterface {
  s360_ctx* ctx; std::string alg;                       // alg: "pixflow_low" | "pixflow_search_20"
  PixFlowHip(s360_ctx* c, const std::string& a) : ctx(c), alg(a) {}
  void computeOpticalFlow(const Mat& I0BGRA, const Mat& I1BGRA, const Mat& prevFlow, const Mat& prevI0BGRA,
                          const Mat& prevI1BGRA, Mat& flow, DirectionHint hint) override {
    CHECK(I0BGRA.isContinuous() && I1BGRA.isContinuous() && I0BGRA.type() == CV_8UC4);
    flow.create(I0BGRA.size(), CV_32FC2);               // callee allocates, like PixFlow.h:176
    const bool prev = prevFlow.dims > 0;                 // empty Mat == no temporal regularisation (PixFlow.h:147)
    const int rc = s360_compute_optical_flow(ctx, alg.c_str(), I0BGRA.data, I1BGRA.data, I0BGRA.cols, I0BGRA.rows,
        prev ? (const float*)prevFlow.data : nullptr, prev ? prevI0BGRA.data : nullptr,
        prev ? prevI1BGRA.data : nullptr, (int)hint /* same enum order */, (float*)flow.data);
    if (rc != S360_OK) throw VrCamException(s360_last_error(ctx));   // reference error behaviour (VrCamException.h)
  }
};

}  // namespace optical_flow
}  // namespace surround360
