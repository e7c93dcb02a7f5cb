// DEVELOPER TOOL — not part of the product: the ISP's HIP sources compiled for the CPU over tools/hip_cpu_shim and run
// thread by thread (see the shim's header). Build: make -C tools libisp_emu.so. Used by tests/test_cpu_isp.py to check
// the kernels' indexing and arithmetic against the oracle where no GPU is attached.
#include <hip/hip_runtime.h>

thread_local uint3_ threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
EmuBarrier g_emu_barrier;

#include "../surround360_amd/csrc/isp.hpp"
#include <cstring>

extern "C" int emu_isp_run_packed(const s360_isp_config* cfg, const uint8_t* frame, int bits, int w, int h, void* out, char* err, int cap) {
  try {
    s360_isp o;
    s360::isp_init(&o, 0, *cfg);
    s360::isp_process_packed(&o, frame, bits, w, h, out);
    s360::isp_release(&o);
    return 0;
  } catch (const std::exception& e) {
    if (err && cap > 0) { std::strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
    return -1;
  }
}
extern "C" int emu_isp_run(const s360_isp_config* cfg, const uint16_t* raw, int w, int h, void* out, char* err, int cap) {
  try {
    s360_isp o;
    s360::isp_init(&o, 0, *cfg);
    s360::isp_process(&o, raw, w, h, out);
    s360::isp_release(&o);
    return 0;
  } catch (const std::exception& e) {
    if (err && cap > 0) { std::strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
    return -1;
  }
}

// (isp.cpp asks api.hip whether the context an ISP object is bound to still lives; this tool has no contexts)
namespace s360 { bool context_alive(unsigned long long) { return true; } }
