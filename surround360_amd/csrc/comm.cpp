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

#include <cstdio>
#include <cstdlib>
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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  std::string err, path;
};
Rccl& rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    // S360_RCCL_LIB names the library outright (a node whose default is not the one to use); otherwise by soname — in a process
    // that already holds an RCCL (torch ships its own copy) that one is shared, which is what one process wants —, then ROCm's
    const char* forced = std::getenv("S360_RCCL_LIB");
    if (forced && forced[0]) {
      R.h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
      if (!R.h) { const char* de = dlerror(); R.err = std::string("S360_RCCL_LIB=") + forced + ": " + (de ? de : "dlopen failed"); return; }
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (R.h) break;
      }
    }
    if (!R.h) { const char* de = dlerror(); R.err = std::string("librccl not found: ") + (de ? de : "dlopen failed"); return; }
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
    R.CommCount = (decltype(R.CommCount))sym("ncclCommCount");
    R.CommUserRank = (decltype(R.CommUserRank))sym("ncclCommUserRank");
    Dl_info di;
    if (R.GetUniqueId && dladdr((void*)R.GetUniqueId, &di) && di.dli_fname && di.dli_fname[0]) R.path = di.dli_fname;
    else R.path = "?";  // loaded, but the loader cannot name the file (NULL is kept for "no librccl": s360.h)
    const char* v = std::getenv("S360_RCCL_VERBOSE");
    if (v && v[0] == '1') std::fprintf(stderr, "libs360: RCCL entry points from %s\n", R.path.c_str());
  });
  if (!R.err.empty()) throw Error(S360_ERR_STATE, R.err);
  return R;
}
void nccl_ck(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw Error(S360_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(r));
}
}  // namespace

const char* comm_library_path() { return rccl().path.c_str(); }  // (the singleton's string: written once, under call_once)
// What the COMMUNICATOR says (ncclCommCount / ncclCommUserRank), not what this library remembers having asked for.
int comm_size(s360_ctx* c) {
  if (!c->comm) return 0;
  int n = 0;
  nccl_ck(rccl().CommCount((ncclComm_t)c->comm, &n), "ncclCommCount");
  return n;
}
int comm_rank(s360_ctx* c) {
  if (!c->comm) return -1;
  int r = -1;
  nccl_ck(rccl().CommUserRank((ncclComm_t)c->comm, &r), "ncclCommUserRank");
  return r;
}
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
  c->comm_stats[0] = c->comm_stats[1] = s360_ctx::CommStats();
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
    ctxs[i]->comm_stats[0] = ctxs[i]->comm_stats[1] = s360_ctx::CommStats();
  }
}
void comm_destroy(s360_ctx* c) {
  if (!c->comm) return;
  (void)rccl().CommDestroy((ncclComm_t)c->comm);
  c->comm = nullptr;
  c->comm_size = 1;
  c->comm_rank = 0;
}

// bounds[r] .. bounds[r+1]: the pairs rank r rendered (contiguous, possibly empty). need_mask[r]: the eyes (bit 0 left,
// bit 1 right) whose COMPLETE set of strips rank r needs next — the root of the composite needs both, the owner of a pole
// unit the eye(s) of its unit(s) (poleToSideFlowThread reads the whole side panorama of its eye, TRSP:388-398), the other
// ranks nothing. One ncclGroup: every rank sends its block of eye e to every other rank that needs eye e and receives
// the blocks it needs straight into [eye][pair][camH][stripW] at the pair's offset. Enqueued on the context stream:
// ordered after this rank's s360_frame_render_pairs and before its pole units / composite.
void frame_exchange_strips(s360_ctx* c, const int* bounds, const int* need_mask) {
  if (!c->comm) throw Error(S360_ERR_STATE, "no communicator: call s360_comm_init_rank / s360_comm_init_all first");
  Rccl& R = rccl();
  FrameState& F = frame_state(c);
  const int P = F.P, nr = c->comm_size, me = c->comm_rank;
  if (bounds[0] != 0 || bounds[nr] != P) throw Error(S360_ERR_INVALID_ARG, "bounds must cover the pairs [0, n_side)");
  for (int r = 0; r < nr; ++r) {
    if (bounds[r + 1] < bounds[r]) throw Error(S360_ERR_INVALID_ARG, "bounds must be non-decreasing");
    if (need_mask[r] & ~3) throw Error(S360_ERR_INVALID_ARG, "need_mask: bit 0 = left eye, bit 1 = right eye");
  }
  const size_t per = (size_t)c->g.cam_image_height * (c->P.eqr_width / P) * sizeof(uchar4);
  F.strips.ensure(2 * P * per);
  uint8_t* base = F.strips.as<uint8_t>();
  if (nr > 1) {
    ProfScope ps(c->prof, "exchange_strips");
    nccl_ck(R.GroupStart(), "ncclGroupStart");
    ncclResult_t rc = ncclSuccess;
    const int mine = bounds[me + 1] - bounds[me];
    c->comm_stats[0].calls++;
    for (int eye = 0; eye < 2 && rc == ncclSuccess; ++eye) {
      uint8_t* e = base + (size_t)eye * P * per;
      for (int r = 0; r < nr && rc == ncclSuccess; ++r) {
        if (r == me) continue;
        const int n = bounds[r + 1] - bounds[r];
        if (mine > 0 && ((need_mask[r] >> eye) & 1)) {  // r assembles this eye: it gets my block
          rc = R.Send(e + bounds[me] * per, mine * per, ncclUint8, r, (ncclComm_t)c->comm, c->st);
          c->comm_stats[0].sent += mine * per;
        }
        if (rc == ncclSuccess && n > 0 && ((need_mask[me] >> eye) & 1)) {  // I assemble this eye: r's block comes in
          rc = R.Recv(e + bounds[r] * per, n * per, ncclUint8, r, (ncclComm_t)c->comm, c->st);
          c->comm_stats[0].received += n * per;
        }
      }
    }
    const ncclResult_t rc2 = R.GroupEnd();
    nccl_ck(rc, "ncclSend/ncclRecv");
    nccl_ck(rc2, "ncclGroupEnd");
  }
  // frame pipelining: the finish stream must also wait for the gathered strips
  if (c->pipeline && c->evSideDone) S360_HIP(hipEventRecord(c->evSideDone, c->st));
}
// the gather of SURVEY 8e's first variant: only the root assembles (it runs all pole units and the composite)
void frame_gather_strips(s360_ctx* c, const int* bounds, int root) {
  if (!c->comm) throw Error(S360_ERR_STATE, "no communicator: call s360_comm_init_rank / s360_comm_init_all first");
  if (root < 0 || root >= c->comm_size) throw Error(S360_ERR_INVALID_ARG, "bad root");
  std::vector<int> need(c->comm_size, 0);
  need[root] = 3;
  frame_exchange_strips(c, bounds, need.data());
}

// The second exchange of a frame whose pole units run on several GPUs (the reference's four
// poleToSideFlowThread threads, TRSP:811-860): owner[u] is the rank that ran unit u (0 top_left, 1 top_right,
// 2 bottom_left, 3 bottom_right; -1 = unit not enabled). Each owner other than the root sends the unit's warped layer —
// the pole rows; below them the layer is transparent padding (TRSP:538-546) — and the root receives it where its own
// units' layers live, ready for s360_frame_composite. One ncclGroup on the context stream.
void frame_gather_pole_layers(s360_ctx* c, const int* owner, int root) {
  if (!c->comm) throw Error(S360_ERR_STATE, "no communicator: call s360_comm_init_rank / s360_comm_init_all first");
  Rccl& R = rccl();
  FrameState& F = frame_state(c);
  const int nr = c->comm_size, me = c->comm_rank;
  if (root < 0 || root >= nr) throw Error(S360_ERR_INVALID_ARG, "bad root");
  const int W = c->P.eqr_width, H = c->P.eqr_height;
  const size_t en = (size_t)W * H * sizeof(uchar4);
  int todo = 0;
  for (int u = 0; u < 4; ++u) {
    if (owner[u] >= nr) throw Error(S360_ERR_INVALID_ARG, "owner: not a rank");
    if (owner[u] < 0 || owner[u] == root) continue;
    const size_t bytes = (size_t)W * (u < 2 ? c->g.top_rows : c->g.bottom_rows) * sizeof(uchar4);
    if (me == owner[u] && !F.sc->poleWarped[u].p) throw Error(S360_ERR_STATE, "pole unit " + std::to_string(u) + " has not been run on its owner");
    if (me == root) {
      F.sc->poleWarped[u].ensure(en);
      S360_HIP(hipMemsetAsync(F.sc->poleWarped[u].as<uint8_t>() + bytes, 0, en - bytes, c->st));
    }
    if (me == owner[u] || me == root) ++todo;
  }
  if (!todo) return;
  ProfScope ps(c->prof, "exchange_pole_layers");
  nccl_ck(R.GroupStart(), "ncclGroupStart");
  ncclResult_t rc = ncclSuccess;
  c->comm_stats[1].calls++;
  for (int u = 0; u < 4 && rc == ncclSuccess; ++u) {
    if (owner[u] < 0 || owner[u] == root) continue;
    const size_t bytes = (size_t)W * (u < 2 ? c->g.top_rows : c->g.bottom_rows) * sizeof(uchar4);
    if (me == owner[u]) {
      rc = R.Send(F.sc->poleWarped[u].p, bytes, ncclUint8, root, (ncclComm_t)c->comm, c->st);
      c->comm_stats[1].sent += bytes;
    } else if (me == root) {
      rc = R.Recv(F.sc->poleWarped[u].p, bytes, ncclUint8, owner[u], (ncclComm_t)c->comm, c->st);
      c->comm_stats[1].received += bytes;
      F.poleFrame[u] = F.frames_done;  // this frame's layer (frame_composite refuses an earlier frame's)
      F.sc->poleOwner[u] = &F;
    }
  }
  const ncclResult_t rc2 = R.GroupEnd();
  nccl_ck(rc, "ncclSend/ncclRecv");
  nccl_ck(rc2, "ncclGroupEnd");
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
