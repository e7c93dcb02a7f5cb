// DEVELOPER / TEST TOOL — in-process stand-in for the nine RCCL entry points comm.cpp uses, for the CPU emulation of the
// library. Ranks are host threads of one process (ncclCommInitAll; or ncclCommInitRank with one thread per rank). The
// emulated "streams" run everything at call time, so at ncclGroupEnd the data of every send is final: a send is a copy
// into a mailbox keyed by (communicator group, source, destination), a receive blocks until the matching send has been
// posted, checks that the sizes agree (a mismatch is what would hang or corrupt on real RCCL) and copies it out. Sends
// of a group are posted before its receives are waited for, so grouped exchanges between threads cannot deadlock.
// With EMU_RCCL_DIR set, communicators made by ncclCommInitRank use files in that directory as the mailbox instead
// (message k from src to dst = <dir>/<id>_<src>_<dst>_<k>, written under a temporary name and renamed), so that the ranks
// can be separate PROCESSES — one per rank under torch.distributed.run, as on a real node.
// EMU_RCCL_STRICT=1 (both transports): nothing is buffered. A send completes only against a receive that its peer has
// posted in the group it is executing AT THE SAME TIME (a rendezvous, like RCCL's point-to-point kernels, which run on both
// sides at once): ranks whose groups are ordered differently — A: {send to B} then {recv from B}, B the same towards A —
// pass the buffered mailbox and hang on a real node; here they time out after EMU_RCCL_STRICT_SECONDS (20) with a message.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#undef dlopen
#undef dlsym
#undef dlerror

#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Active {  // strict mode: an operation of a group that is executing right now
  int src, dst;
  bool send;
  void* p;
  size_t n;
  bool done = false, mismatch = false;
};
struct Group {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<Active>> active;  // strict mode, in posting order
  std::map<std::pair<int, int>, std::deque<std::vector<unsigned char>>> box;  // (src, dst) -> messages in order
  int joined = 0;
};
}  // namespace
struct ncclComm {
  std::shared_ptr<Group> g;
  int rank = 0;
  std::string dir, id;              // file transport (EMU_RCCL_DIR): directory and hex id of the communicator
  std::vector<unsigned> sent, got;  // per peer: messages sent to / received from it so far
};
namespace {
std::mutex g_mu;
std::map<std::string, std::shared_ptr<Group>> g_byId;
std::atomic<unsigned> g_next{1};
struct Op { bool send; void* p; size_t n; int peer; ncclComm* c; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local std::string t_err;

std::string msg_path(const ncclComm* c, int src, int dst, unsigned k) {
  return c->dir + "/" + c->id + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(k);
}
bool strict_mode() {
  static const bool on = [] { const char* e = std::getenv("EMU_RCCL_STRICT"); return e && e[0] == '1'; }();
  return on;
}
int strict_seconds() {
  const char* se = std::getenv("EMU_RCCL_STRICT_SECONDS");
  return se ? std::atoi(se) : 20;
}
ncclResult_t run_files(const std::vector<Op>& ops) {
  if (strict_mode()) {
    // rendezvous through files: every receive of the group first announces itself (<message>.rdy); a send writes its message
    // only once the receive it pairs with has been announced — i.e. while the peer is inside a group that holds it
    std::map<std::pair<const ncclComm*, int>, unsigned> nrecv, nsend;  // operations of THIS group per (communicator, peer) so far
    for (const Op& o : ops)
      if (!o.send) {
        const std::string rdy = msg_path(o.c, o.peer, o.c->rank, o.c->got[o.peer] + nrecv[{o.c, o.peer}]++) + ".rdy";
        FILE* f = std::fopen(rdy.c_str(), "wb");
        if (f) std::fclose(f);
      }
    for (const Op& o : ops)
      if (o.send) {
        const std::string rdy = msg_path(o.c, o.c->rank, o.peer, o.c->sent[o.peer] + nsend[{o.c, o.peer}]++) + ".rdy";
        bool seen = false;
        for (int tries = 0; tries < strict_seconds() * 1000 && !(seen = access(rdy.c_str(), F_OK) == 0); ++tries)
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (!seen) {
          t_err = "emulated RCCL (strict): rank " + std::to_string(o.c->rank) + " sends to rank " + std::to_string(o.peer) +
                  ", which posts no matching receive in a group executing at the same time — on a real node this exchange hangs";
          return ncclInternalError;
        }
        std::remove(rdy.c_str());
      }
  }
  for (const Op& o : ops)
    if (o.send) {
      const std::string path = msg_path(o.c, o.c->rank, o.peer, o.c->sent[o.peer]++), tmp = path + ".part";
      FILE* f = std::fopen(tmp.c_str(), "wb");
      if (!f || std::fwrite(o.p, 1, o.n, f) != o.n) { t_err = "emulated ncclSend: cannot write " + tmp; if (f) std::fclose(f); return ncclInternalError; }
      std::fclose(f);
      if (std::rename(tmp.c_str(), path.c_str()) != 0) { t_err = "emulated ncclSend: rename failed"; return ncclInternalError; }
    }
  for (const Op& o : ops)
    if (!o.send) {
      const std::string path = msg_path(o.c, o.peer, o.c->rank, o.c->got[o.peer]++);
      FILE* f = nullptr;
      for (int tries = 0; tries < 120000 && !(f = std::fopen(path.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(1));
      if (!f) { t_err = "emulated ncclRecv: no matching send within 120 s (" + path + ")"; return ncclInternalError; }
      std::fseek(f, 0, SEEK_END);
      const long sz = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      if ((size_t)sz != o.n || std::fread(o.p, 1, o.n, f) != o.n) { std::fclose(f); t_err = "emulated ncclRecv: size mismatch between send and receive"; return ncclInvalidArgument; }
      std::fclose(f);
      std::remove(path.c_str());
    }
  return ncclSuccess;
}
// strict mode: post every operation of the group, pair my receives with the peers' posted sends (in order per peer) and wait
// until all of mine — sends included — have been paired; nothing outlives the call
ncclResult_t run_strict(const std::vector<Op>& ops) {
  if (ops.empty()) return ncclSuccess;
  Group& g = *ops[0].c->g;
  const int me = ops[0].c->rank;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(strict_seconds());
  std::vector<std::shared_ptr<Active>> mine;
  std::unique_lock<std::mutex> lk(g.mu);
  for (const Op& o : ops) {
    auto a = std::make_shared<Active>();
    a->src = o.send ? me : o.peer;
    a->dst = o.send ? o.peer : me;
    a->send = o.send;
    a->p = o.p;
    a->n = o.n;
    g.active.push_back(a);
    mine.push_back(a);
  }
  g.cv.notify_all();
  ncclResult_t res = ncclSuccess;
  for (;;) {
    for (auto& r : mine) {
      if (r->send || r->done) continue;
      for (auto& sdr : g.active)  // the earliest send of that peer to me that nobody has taken
        if (sdr->send && !sdr->done && sdr->src == r->src && sdr->dst == me) {
          if (sdr->n != r->n) r->mismatch = sdr->mismatch = true;
          else std::memcpy(r->p, sdr->p, r->n);
          r->done = sdr->done = true;
          break;
        }
    }
    g.cv.notify_all();
    bool all = true;
    for (auto& a : mine) all = all && a->done;
    if (all) break;
    if (g.cv.wait_until(lk, deadline) == std::cv_status::timeout) {
      t_err = "emulated RCCL (strict): rank " + std::to_string(me) + " waited for a peer that never posted the matching operation in a "
              "group executing at the same time — on a real node this exchange hangs (mismatched group order between ranks?)";
      res = ncclInternalError;
      break;
    }
  }
  for (auto& a : mine) {
    if (a->mismatch && res == ncclSuccess) { t_err = "emulated ncclRecv: size mismatch between send and receive"; res = ncclInvalidArgument; }
    for (auto it = g.active.begin(); it != g.active.end(); ++it)
      if (it->get() == a.get()) { g.active.erase(it); break; }
  }
  return res;
}
ncclResult_t run(const std::vector<Op>& ops) {
  if (!ops.empty() && !ops[0].c->dir.empty()) return run_files(ops);
  if (strict_mode()) return run_strict(ops);
  for (const Op& o : ops)
    if (o.send) {
      Group& g = *o.c->g;
      std::lock_guard<std::mutex> lk(g.mu);
      const unsigned char* b = static_cast<const unsigned char*>(o.p);
      g.box[{o.c->rank, o.peer}].emplace_back(b, b + o.n);
      g.cv.notify_all();
    }
  for (const Op& o : ops)
    if (!o.send) {
      Group& g = *o.c->g;
      std::unique_lock<std::mutex> lk(g.mu);
      auto& q = g.box[{o.peer, o.c->rank}];
      if (!g.cv.wait_for(lk, std::chrono::seconds(120), [&] { return !q.empty(); })) {
        t_err = "emulated ncclRecv: no matching send within 120 s";
        return ncclInternalError;
      }
      std::vector<unsigned char> m = std::move(q.front());
      q.pop_front();
      if (m.size() != o.n) {
        t_err = "emulated ncclRecv: size mismatch between send and receive";
        return ncclInvalidArgument;
      }
      std::memcpy(o.p, m.data(), o.n);
    }
  return ncclSuccess;
}
ncclResult_t post(bool send, void* p, size_t n, int peer, ncclComm_t c) {
  if (!c || peer < 0 || peer >= c->g->n) { t_err = "emulated RCCL: bad communicator or peer"; return ncclInvalidArgument; }
  Op o{send, p, n, peer, c};
  if (t_depth > 0) { t_ops.push_back(o); return ncclSuccess; }
  return run({o});
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  const unsigned k[3] = {g_next.fetch_add(1), (unsigned)getpid(),
                         (unsigned)std::chrono::steady_clock::now().time_since_epoch().count()};
  std::memcpy(id->internal, k, sizeof k);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
  if (const char* dir = std::getenv("EMU_RCCL_DIR")) {  // ranks are processes: files are the mailbox
    if (rank < 0 || rank >= n) return ncclInvalidArgument;
    auto g = std::make_shared<Group>();
    g->n = n;
    ncclComm* c = new ncclComm{g, rank};
    c->dir = dir;
    char hex[32];
    unsigned k[3];
    std::memcpy(k, id.internal, sizeof k);
    std::snprintf(hex, sizeof hex, "%08x%08x%08x", k[0], k[1], k[2]);
    c->id = hex;
    c->sent.assign(n, 0);
    c->got.assign(n, 0);
    *out = c;
    return ncclSuccess;
  }
  std::shared_ptr<Group> g;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto& slot = g_byId[std::string(id.internal, sizeof id.internal)];
    if (!slot) { slot = std::make_shared<Group>(); slot->n = n; }
    g = slot;
  }
  if (g->n != n || rank < 0 || rank >= n) return ncclInvalidArgument;
  {  // like the real call: returns once every rank has joined
    std::unique_lock<std::mutex> lk(g->mu);
    ++g->joined;
    g->cv.notify_all();
    if (!g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->joined >= n; })) return ncclInternalError;
  }
  *out = new ncclComm{g, rank};
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
  auto g = std::make_shared<Group>();
  g->n = n;
  g->joined = n;
  for (int i = 0; i < n; ++i) comms[i] = new ncclComm{g, i};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidArgument;
  if (--t_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run(ops);
}
ncclResult_t ncclSend(const void* p, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t) { return post(true, const_cast<void*>(p), n, peer, c); }
ncclResult_t ncclRecv(void* p, size_t n, ncclDataType_t, int peer, ncclComm_t c, hipStream_t) { return post(false, p, n, peer, c); }
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { if (!c || !n) return ncclInvalidArgument; *n = c->g->n; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { if (!c || !r) return ncclInvalidArgument; *r = c->rank; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : (t_err.empty() ? "emulated RCCL error" : t_err.c_str()); }
}

void* emu_rccl_sym(const char* name) {
  const std::string s(name);
#define S(n) if (s == #n) return (void*)&n
  S(ncclGetUniqueId); S(ncclCommInitRank); S(ncclCommInitAll); S(ncclCommDestroy); S(ncclGroupStart); S(ncclGroupEnd);
  S(ncclSend); S(ncclRecv); S(ncclGetErrorString); S(ncclCommCount); S(ncclCommUserRank);
#undef S
  return nullptr;
}
