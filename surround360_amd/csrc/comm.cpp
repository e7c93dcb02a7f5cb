// comm.cpp — the one exchange of the sharded frame: the strips rendered on the other ranks' GPUs
// arrive in the root's strip buffers through grouped RCCL point-to-point calls over xGMI.
//
// Reference shape: renderStereoPanoramaChunksThread per pair + join + stackHorizontal (TRSP:320-384); here every rank
// owns a contiguous block of pairs and the root receives each block straight into [eye][pair][camH][stripW] at the
// pair's offset, so the "stackHorizontal" is the destination address. Counts are unequal (14 pairs over 8 ranks:
// 2,2,2,2,2,2,1,1), which is why this is send/recv inside one ncclGroup rather than ncclGather; all of the root's
// inbound links are in flight at once. librccl is resolved at run time (dlopen) so that libs360.so loads on a box
// without RCCL and shares the copy a host process (e.g. torch) has already loaded.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/s360.h"
#include "ctx.hpp"
#include "render.hpp"

namespace s360 {

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
Rccl& rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (R.h) break;
    }
    if (!R.h) { R.err = std::string("librccl not found: ") + dlerror(); return; }
    auto sym = [&](const char* n) { void* p = dlsym(R.h, n); if (!p) R.err = std::string("librccl lacks ") + n; return p; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.Send = (decltype(R.Send))sym("ncclSend");
    R.Recv = (decltype(R.Recv))sym("ncclRecv");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  });
  if (!R.err.empty()) throw Error(S360_ERR_STATE, R.err);
  return R;
}
void nccl_ck(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw Error(S360_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(r));
}
}  // namespace

void comm_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == S360_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  nccl_ck(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id128, &id, sizeof id);
}
void comm_init_rank(s360_ctx* c, const void* id128, int rank, int nranks) {
  if (c->comm) throw Error(S360_ERR_STATE, "context already has a communicator");
  if (nranks < 1 || rank < 0 || rank >= nranks) throw Error(S360_ERR_INVALID_ARG, "bad rank / nranks");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  nccl_ck(rccl().CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_size = nranks;
}
void comm_init_all(s360_ctx* const* ctxs, int n) {
  if (n < 1) throw Error(S360_ERR_INVALID_ARG, "no contexts");
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i] || ctxs[i]->comm) throw Error(S360_ERR_STATE, "null context or communicator already set");
    devs[i] = ctxs[i]->device;
    for (int j = 0; j < i; ++j)
      if (devs[j] == devs[i]) throw Error(S360_ERR_INVALID_ARG, "one context per device: RCCL ranks cannot share a GPU");
  }
  std::vector<ncclComm_t> comms(n, nullptr);
  nccl_ck(rccl().CommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll");
  for (int i = 0; i < n; ++i) {
    ctxs[i]->comm = comms[i];
    ctxs[i]->comm_rank = i;
    ctxs[i]->comm_size = n;
  }
}
void comm_destroy(s360_ctx* c) {
  if (!c->comm) return;
  (void)rccl().CommDestroy((ncclComm_t)c->comm);
  c->comm = nullptr;
  c->comm_size = 1;
  c->comm_rank = 0;
}

// bounds[r] .. bounds[r+1]: the pairs rank r rendered (contiguous, possibly empty). Enqueued on the context stream:
// ordered after this rank's s360_frame_render_pairs and before its s360_frame_finish.
void frame_gather_strips(s360_ctx* c, const int* bounds, int root) {
  if (!c->comm) throw Error(S360_ERR_STATE, "no communicator: call s360_comm_init_rank / s360_comm_init_all first");
  Rccl& R = rccl();
  FrameState& F = frame_state(c);
  const int P = F.P, nr = c->comm_size, me = c->comm_rank;
  if (root < 0 || root >= nr) throw Error(S360_ERR_INVALID_ARG, "bad root");
  if (bounds[0] != 0 || bounds[nr] != P) throw Error(S360_ERR_INVALID_ARG, "bounds must cover the pairs [0, n_side)");
  for (int r = 0; r < nr; ++r)
    if (bounds[r + 1] < bounds[r]) throw Error(S360_ERR_INVALID_ARG, "bounds must be non-decreasing");
  const size_t per = (size_t)c->g.cam_image_height * (c->P.eqr_width / P) * sizeof(uchar4);
  F.strips.ensure(2 * P * per);
  uint8_t* base = F.strips.as<uint8_t>();
  if (nr > 1) {
    nccl_ck(R.GroupStart(), "ncclGroupStart");
    ncclResult_t rc = ncclSuccess;
    for (int eye = 0; eye < 2 && rc == ncclSuccess; ++eye) {
      uint8_t* e = base + (size_t)eye * P * per;
      if (me == root) {
        for (int r = 0; r < nr && rc == ncclSuccess; ++r) {
          const int n = bounds[r + 1] - bounds[r];
          if (r == root || n == 0) continue;
          rc = R.Recv(e + bounds[r] * per, n * per, ncclUint8, r, (ncclComm_t)c->comm, c->st);
        }
      } else {
        const int n = bounds[me + 1] - bounds[me];
        if (n > 0) rc = R.Send(e + bounds[me] * per, n * per, ncclUint8, root, (ncclComm_t)c->comm, c->st);
      }
    }
    const ncclResult_t rc2 = R.GroupEnd();
    nccl_ck(rc, "ncclSend/ncclRecv");
    nccl_ck(rc2, "ncclGroupEnd");
  }
  // frame pipelining: the finish stream must also wait for the gathered strips
  if (c->pipeline && c->evSideDone) S360_HIP(hipEventRecord(c->evSideDone, c->st));
}

// One grouped send+recv of a rank to itself through the same code path (pair `src` of eye 0 into the slot of pair
// `dst`): exercises RCCL point-to-point on the context's stream and buffers where only one GPU is available.
void comm_loopback(s360_ctx* c, int src, int dst) {
  if (!c->comm) throw Error(S360_ERR_STATE, "no communicator");
  Rccl& R = rccl();
  FrameState& F = frame_state(c);
  const int P = F.P;
  if (src < 0 || src >= P || dst < 0 || dst >= P || src == dst) throw Error(S360_ERR_INVALID_ARG, "bad pair index");
  const size_t per = (size_t)c->g.cam_image_height * (c->P.eqr_width / P) * sizeof(uchar4);
  F.strips.ensure(2 * P * per);
  uint8_t* base = F.strips.as<uint8_t>();
  nccl_ck(R.GroupStart(), "ncclGroupStart");
  const ncclResult_t a = R.Send(base + src * per, per, ncclUint8, c->comm_rank, (ncclComm_t)c->comm, c->st);
  const ncclResult_t b = R.Recv(base + dst * per, per, ncclUint8, c->comm_rank, (ncclComm_t)c->comm, c->st);
  const ncclResult_t e = R.GroupEnd();
  nccl_ck(a, "ncclSend");
  nccl_ck(b, "ncclRecv");
  nccl_ck(e, "ncclGroupEnd");
}

}  // namespace s360
