// TEST INFRASTRUCTURE (oracle/_ref): the reference's own PixFlow — optical_flow/PixFlow.h through
// OpticalFlowFactory.h's makeOpticalFlowByName — compiled from /root/reference where it lies, behind a C entry point.
// OpenCV is replaced by oracle/ref_shim: containers and element-wise arithmetic in opencv2/core.hpp, and the imgproc
// algorithms PixFlow calls (resize, GaussianBlur, Sobel, medianBlur, cvtColor, split) routed to the oracle's own
// restatements (cvlite.h). What this library pins is therefore the oracle's restatement of PixFlow's OWN logic
// (pyramids, search, sweeps, error function, diffusion, temporal regularisation, op order) against the reference's
// source; the OpenCV primitives are common to both sides and stay unpinned (cvlite.h header).
#include <cstdint>
#include <cstring>
#include <string>

#include "OpticalFlowFactory.h"

using namespace surround360::optical_flow;

extern "C" int ref_pixflow(const char* alg, const uint8_t* i0, const uint8_t* i1, int w, int h, const float* prev_flow,
                           const uint8_t* prev_i0, const uint8_t* prev_i1, int hint, float* flow_out, char* err, int cap) {
  try {
    cv::Mat I0(h, w, CV_8UC4, const_cast<uint8_t*>(i0)), I1(h, w, CV_8UC4, const_cast<uint8_t*>(i1));
    cv::Mat pf, p0, p1, flow;
    if (prev_flow) {
      pf = cv::Mat(h, w, CV_32FC2, const_cast<float*>(prev_flow));
      p0 = cv::Mat(h, w, CV_8UC4, const_cast<uint8_t*>(prev_i0));
      p1 = cv::Mat(h, w, CV_8UC4, const_cast<uint8_t*>(prev_i1));
    }
    OpticalFlowInterface* f = makeOpticalFlowByName(alg);
    f->computeOpticalFlow(I0, I1, pf, p0, p1, flow, static_cast<OpticalFlowInterface::DirectionHint>(hint));
    delete f;
    if (flow.rows != h || flow.cols != w || flow.type() != CV_32FC2) throw std::runtime_error("unexpected flow size / type");
    std::memcpy(flow_out, flow.data, (size_t)w * h * 2 * sizeof(float));
    return 0;
  } catch (const std::exception& e) {
    if (err && cap > 0) { std::strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
    return -1;
  }
}
