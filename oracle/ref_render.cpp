// TEST INFRASTRUCTURE (oracle/_ref): the reference's own novel-view and layer utilities — optical_flow/NovelView.cpp
// (renderLazyNovelView, combineLazyViews, combineLazyNovelViews) and util/CvUtil.cpp (flattenLayersDeghostPreferBase,
// featherAlphaChannel, offsetHorizontalWrap) — compiled from
// /root/reference where they lie over oracle/ref_shim (see ref_pixflow.cpp for what that pins and what it cannot).
// The one piece restated here rather than compiled is the LazyNovelViewBuffer fill of
// renderStereoPanoramaChunksThread (test/TestRenderStereoPanorama.cpp:271-285; that file needs gflags, folly and Eigen).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "CvUtil.h"
#include "NovelView.h"

using namespace surround360;
using namespace surround360::util;
using namespace surround360::optical_flow;

namespace {
int fail(const std::exception& e, char* err, int cap) {
  if (err && cap > 0) { std::strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
  return -1;
}
}  // namespace

extern "C" {
// imgL / imgR: camH x overlapW BGRA; flows camH x overlapW x 2. outL / outR: camH x chunkW BGRA (chunkW = eqr_width / numCams).
int ref_combine_lazy_novel_views(const uint8_t* imgL, const uint8_t* imgR, const float* flowLtoR, const float* flowRtoL,
                                 int overlapW, int camH, int chunkW, int numNovelViews, int camImageWidth,
                                 float vergeAtInfinitySlabDisplacement, uint8_t* outL, uint8_t* outR, char* err, int cap) {
  try {
    NovelViewGeneratorAsymmetricFlow gen("pixflow_low");
    gen.imageL = cv::Mat(camH, overlapW, CV_8UC4, const_cast<uint8_t*>(imgL));
    gen.imageR = cv::Mat(camH, overlapW, CV_8UC4, const_cast<uint8_t*>(imgR));
    gen.flowLtoR = cv::Mat(camH, overlapW, CV_32FC2, const_cast<float*>(flowLtoR));
    gen.flowRtoL = cv::Mat(camH, overlapW, CV_32FC2, const_cast<float*>(flowRtoL));
    LazyNovelViewBuffer buf(chunkW, camH);
    int currChunkX = 0;
    for (int nvIdx = 0; nvIdx < numNovelViews; ++nvIdx) {  // TestRenderStereoPanorama.cpp:271-285
      const float shift = float(nvIdx) / float(numNovelViews);
      const float slabShift = float(camImageWidth) * 0.5f - float(numNovelViews - nvIdx);
      for (int v = 0; v < camH; ++v) {
        buf.warpL[currChunkX][v] = cv::Point3f(slabShift + vergeAtInfinitySlabDisplacement, v, shift);
        buf.warpR[currChunkX][v] = cv::Point3f(slabShift - vergeAtInfinitySlabDisplacement, v, shift);
      }
      ++currChunkX;
    }
    std::pair<cv::Mat, cv::Mat> lr = gen.combineLazyNovelViews(buf);
    if (lr.first.rows != camH || lr.first.cols != chunkW || lr.first.type() != CV_8UC4) throw std::runtime_error("unexpected chunk size / type");
    std::memcpy(outL, lr.first.data, (size_t)camH * chunkW * 4);
    std::memcpy(outR, lr.second.data, (size_t)camH * chunkW * 4);
    return 0;
  } catch (const std::exception& e) { return fail(e, err, cap); }
}
int ref_flatten_layers(const uint8_t* base, const uint8_t* top, int w, int h, uint8_t* out, char* err, int cap) {
  try {
    cv::Mat r = flattenLayersDeghostPreferBase(cv::Mat(h, w, CV_8UC4, const_cast<uint8_t*>(base)), cv::Mat(h, w, CV_8UC4, const_cast<uint8_t*>(top)));
    std::memcpy(out, r.data, (size_t)w * h * 4);
    return 0;
  } catch (const std::exception& e) { return fail(e, err, cap); }
}
int ref_feather_alpha_channel(const uint8_t* src, int w, int h, int erode_size, uint8_t* out, char* err, int cap) {
  try {
    cv::Mat r = featherAlphaChannel(cv::Mat(h, w, CV_8UC4, const_cast<uint8_t*>(src)), erode_size);
    std::memcpy(out, r.data, (size_t)w * h * 4);
    return 0;
  } catch (const std::exception& e) { return fail(e, err, cap); }
}
int ref_offset_horizontal_wrap(const uint8_t* src, int w, int h, int channels, float offset, uint8_t* out, char* err, int cap) {
  try {
    cv::Mat r = offsetHorizontalWrap(cv::Mat(h, w, CV_MAKETYPE(cv::CV_8U, channels), const_cast<uint8_t*>(src)), offset);
    std::memcpy(out, r.data, (size_t)w * h * channels);
    return 0;
  } catch (const std::exception& e) { return fail(e, err, cap); }
}
}
