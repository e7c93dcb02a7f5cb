// DEVELOPER / TEST TOOL — not part of the product: the library's whole PixFlow path (FlowEngine: flow.hip,
// flow_kernels.hip, median.hip, sweep_lock.hip, sweep_quad.hip) compiled for the CPU over tools/hip_wave_shim and run
// kernel by kernel with the GPU's execution model (see the shim's header). tests/test_cpu_flow_emulation.py compares its
// flows with the oracle's computeOpticalFlow bit for bit, so that the HIP sources' indexing, batching, pyramid schedule
// and hand-offs are checked where no GPU is attached. Build: make -C tools libflow_emu.so.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../surround360_amd/csrc/flow.hpp"

using namespace s360;

// A batch like the library's: n_images images of h x w BGRA, n_flows flows, flow b matching image i0[b] against i1[b]
// (all with one direction hint, as FlowEngine::compute takes it); optional previous images / flows for the temporal
// regularisation; out: n_flows x h x w x 2. sweep_mode 2 = latency kernel, 3 = throughput kernel.
extern "C" int emu_flow_batch(const uint8_t* images, int n_images, int w, int h, const char* alg, int hint, int n_flows,
                              const int* i0, const int* i1, const uint8_t* prev_images, const float* prev_flows,
                              int sweep_mode, float* out_flows, char* err, int cap) {
  try {
    const PixFlowConsts pc = pixflow_consts_by_name(alg);
    const size_t n = (size_t)w * h;
    DevBuf img, pimg, out, pflow;
    img.ensure((size_t)n_images * n * 4);
    std::memcpy(img.p, images, (size_t)n_images * n * 4);
    out.ensure((size_t)n_flows * n * sizeof(float2));
    FlowBatch fb;
    fb.add_images(img.as<uchar4>(), n_images, n);
    if (prev_flows) {
      pimg.ensure((size_t)n_images * n * 4);
      std::memcpy(pimg.p, prev_images, (size_t)n_images * n * 4);
      pflow.ensure((size_t)n_flows * n * sizeof(float2));
      std::memcpy(pflow.p, prev_flows, (size_t)n_flows * n * sizeof(float2));
      fb.add_prev_images(pimg.as<uchar4>(), n_images, n);
    }
    for (int b = 0; b < n_flows; ++b)
      fb.add_flow(i0[b], i1[b], out.as<float2>() + n * b, prev_flows ? pflow.as<float2>() + n * b : nullptr);
    Profiler prof;
    FlowEngine eng(&prof);
    eng.set_sweep_mode(sweep_mode);
    eng.compute(nullptr, pc, fb, w, h, hint);
    if (eng.take_error(nullptr)) throw Error(-3, "a sweep band timed out waiting for its neighbour");
    std::memcpy(out_flows, out.p, (size_t)n_flows * n * sizeof(float2));
    return 0;
  } catch (const std::exception& e) {
    if (err && cap > 0) { std::strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
    return -1;
  }
}

// The pyramid's resize on its own (B one-channel planes sw x sh -> dw x dh): sizes with many tiles per plane and enough
// workgroups for the XCD-aware tile order to permute them — the frames the emulated tests can afford never get there.
extern "C" int emu_resize_linear_planes(const float* src, int sw, int sh, int B, int dw, int dh, float* dst) {
  try {
    launch_resize_linear_f32(nullptr, src, sw, sh, (size_t)sw * sh, dst, dw, dh, (size_t)dw * dh, 1, B, 1.f, 0);
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}
