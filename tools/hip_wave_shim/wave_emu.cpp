// DEVELOPER / TEST TOOL — the scheduler behind tools/hip_wave_shim/hip/hip_runtime.h (see its header): one OS thread
// per workgroup, the workgroup's GPU threads as coroutines on it (hand-written x86-64 context switch), rendezvous for
// cross-lane operations and barriers, deadlock / divergence detection.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <random>

#if !defined(__x86_64__)
#error "the coroutine switch below is x86-64 only"
#endif

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");

namespace emu {
namespace {
enum State { READY, AT_OP, AT_BARRIER, NAPPING, DONE };
constexpr size_t kStack = 256 << 10;
struct Lane {
  void* sp = nullptr;
  void* stack = nullptr;
  State state = READY;
  unsigned seq = 0;  // rendezvous taken so far
  Op op = OP_SYNC;
  unsigned long long val[2] = {0, 0};
  unsigned valseq[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
  uint3_ tid{0, 0, 0};
};
struct Group {
  std::vector<Lane> lanes;
  void* sched_sp = nullptr;
  int cur = -1;
  uint3_ bid{0, 0, 0};
  dim3 bdim, gdim;
  const std::function<void()>* body = nullptr;
  bool vote_fail = false;
};
thread_local Group* t_g = nullptr;
Lane& me() { return t_g->lanes[t_g->cur]; }
void yield_to_scheduler() { emu_switch(&me().sp, t_g->sched_sp); }
void lane_main() {
  (*t_g->body)();
  me().state = DONE;
  yield_to_scheduler();
  std::abort();  // a finished lane is never resumed
}
[[noreturn]] void die(const Group& g, const char* what) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  std::fprintf(stderr, "emu: %s in workgroup (%u,%u,%u)\n", what, g.bid.x, g.bid.y, g.bid.z);
  for (size_t i = 0; i < g.lanes.size(); i += 64) {
    std::fprintf(stderr, "  wave %zu:", i / 64);
    for (size_t l = i; l < std::min(g.lanes.size(), i + 64); ++l) std::fprintf(stderr, " %d/%u/%d", (int)g.lanes[l].state, g.lanes[l].seq, (int)g.lanes[l].op);
    std::fprintf(stderr, "\n");
  }
  std::abort();
}
void run_group(Group& g) {
  t_g = &g;
  const int n = (int)g.lanes.size(), nwaves = (n + 63) / 64;
  const char* ord = std::getenv("EMU_LANE_ORDER");
  const int order = !ord ? 0 : !std::strcmp(ord, "rev") ? 1 : !std::strcmp(ord, "shuffle") ? 2 : 0;
  std::mt19937 rng(12345u + g.bid.x);
  // lane stacks are kept per OS thread and reused by the workgroups (and launches) that thread runs
  thread_local std::vector<void*> t_stacks;
  while (t_stacks.size() < g.lanes.size()) {
    void* st = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (st == MAP_FAILED) die(g, "mmap of a lane stack failed");
    t_stacks.push_back(st);
  }
  for (size_t li = 0; li < g.lanes.size(); ++li) {
    Lane& L = g.lanes[li];
    L.stack = t_stacks[li];
    uintptr_t top = ((uintptr_t)L.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8);  // after the `ret` into lane_main: rsp == 8 (mod 16), as after a call
    *--sp = (void*)&lane_main;
    for (int k = 0; k < 6; ++k) *--sp = nullptr;
    L.sp = sp;
  }
  std::vector<int> idx(64);
  int idle_rounds = 0, nap_rounds = 0;
  for (;;) {
    bool progressed = false, alive = false, useful = false;  // useful: a lane did something other than come back from a nap
    for (int wv = 0; wv < nwaves; ++wv) {
      const int lo = wv * 64, hi = std::min(n, lo + 64), m = hi - lo;
      for (int k = 0; k < m; ++k) idx[k] = order == 1 ? hi - 1 - k : lo + k;
      if (order == 2) std::shuffle(idx.begin(), idx.begin() + m, rng);
      for (int k = 0; k < m; ++k) {
        Lane& L = g.lanes[idx[k]];
        if (L.state != READY) continue;
        g.cur = idx[k];
        emu_switch(&g.sched_sp, L.sp);
        progressed = true;
        if (L.state != NAPPING) useful = true;
      }
      // every live lane of the wave is now blocked; a cross-lane operation completes when all of them are at it
      int at_op = 0, at_bar = 0, nap = 0, live = 0;
      unsigned seq = 0;
      Op op = OP_SYNC;
      for (int l = lo; l < hi; ++l) {
        const Lane& L = g.lanes[l];
        if (L.state == DONE) continue;
        ++live;
        if (L.state == AT_OP) {
          if (at_op && (L.seq != seq || L.op != op)) die(g, "a wave is split between two different cross-lane operations");
          seq = L.seq; op = L.op; ++at_op;
        } else if (L.state == AT_BARRIER) ++at_bar;
        else if (L.state == NAPPING) ++nap;
      }
      if (live) alive = true;
      if (at_op && at_op != live) die(g, "cross-lane operation under divergent control flow (some lanes are elsewhere)");
      if (at_bar && at_bar != live) die(g, "a wave reached a barrier with part of its lanes");
      if (at_op)
        for (int l = lo; l < hi; ++l) if (g.lanes[l].state == AT_OP) { g.lanes[l].state = READY; progressed = true; }
      if (nap)
        for (int l = lo; l < hi; ++l) if (g.lanes[l].state == NAPPING) g.lanes[l].state = READY;
    }
    if (!alive) break;
    int at_bar = 0, live = 0, ready = 0;
    for (const Lane& L : g.lanes) {
      if (L.state == DONE) continue;
      ++live;
      at_bar += L.state == AT_BARRIER;
      ready += L.state == READY;
    }
    if (at_bar == live) {
      for (Lane& L : g.lanes) if (L.state == AT_BARRIER) L.state = READY;
      progressed = true;
    }
    if (progressed && !useful) {
      // every lane that ran is polling something another workgroup has to write: give that workgroup the core. (The
      // kernels bound their spins by counting them; on a loaded machine a pure yield loop can use up such a bound
      // before the other OS thread has been scheduled at all.)
      std::this_thread::sleep_for(std::chrono::microseconds(++nap_rounds > 64 ? 50 : 0));
    } else nap_rounds = 0;
    if (!progressed) {
      if (!ready) die(g, "deadlock: every lane waits and nothing can release them");
      // only napping lanes: another workgroup has to move first
      std::this_thread::yield();
      if (++idle_rounds > 200000000) die(g, "a workgroup has been spinning for too long");
    } else idle_rounds = 0;
  }
  t_g = nullptr;
}
}  // namespace

const uint3_& tid() { return me().tid; }
const uint3_& bid() { return t_g->bid; }
const dim3& bdim() { return t_g->bdim; }
const dim3& gdim() { return t_g->gdim; }
int lane_id() { return t_g->cur & 63; }
Rendezvous arrive(Op op, unsigned long long v) {
  Lane& L = me();
  Rendezvous r;
  r.seq = L.seq;
  r.slot = (int)(L.seq & 1u);
  L.val[r.slot] = v;
  L.valseq[r.slot] = L.seq;
  L.op = op;
  L.state = AT_OP;
  yield_to_scheduler();
  ++L.seq;
  return r;
}
bool peer(const Rendezvous& r, int lane, unsigned long long* v) {
  const int base = t_g->cur & ~63, l = base + lane;
  if (l >= (int)t_g->lanes.size()) return false;
  const Lane& P = t_g->lanes[l];
  if (P.valseq[r.slot] != r.seq) return false;
  *v = P.val[r.slot];
  return true;
}
void barrier() {
  me().state = AT_BARRIER;
  yield_to_scheduler();
}
void nap() {
  me().state = NAPPING;
  yield_to_scheduler();
}
int barrier_and(int pred) {  // everybody votes | everybody reads | everybody resets, before anybody can vote again
  Group& g = *t_g;
  if (!pred) g.vote_fail = true;
  barrier();
  const int r = g.vote_fail ? 0 : 1;
  barrier();
  g.vote_fail = false;  // (every lane; new votes come only after the third barrier)
  barrier();
  return r;
}
// Workgroups are taken in launch order by a pool of OS threads (EMU_THREADS, default 2 x the hardware threads): as on
// the GPU, a workgroup that has started runs to completion, and one that waits for another (the sweeps' band hand-off)
// waits for one that was started before it.
namespace {
struct Pool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  const std::function<void()>* body = nullptr;
  dim3 grid, block;
  unsigned long long next = 0, total = 0, finished = 0, epoch = 0;
  bool stop = false;
  void worker() {
    unsigned long long seen = 0;
    for (;;) {
      unsigned long long i;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || (epoch != seen && next < total) || (epoch != seen && next >= total); });
        if (stop) return;
        if (next >= total) { seen = epoch; continue; }
        i = next++;
      }
      Group g;
      g.bid = uint3_{(unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((unsigned long long)grid.x * grid.y))};
      g.bdim = block;
      g.gdim = grid;
      g.body = body;
      const unsigned nt = block.x * block.y * block.z;
      g.lanes.resize(nt);
      for (unsigned t = 0; t < nt; ++t) g.lanes[t].tid = uint3_{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
      run_group(g);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (++finished == total) done_cv.notify_all();
      }
    }
  }
  Pool() {
    const char* e = std::getenv("EMU_THREADS");
    unsigned n = e ? (unsigned)std::atoi(e) : 2 * std::max(1u, std::thread::hardware_concurrency());
    if (n < 1) n = 1;
    for (unsigned k = 0; k < n; ++k) th.emplace_back([this] { worker(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
  void run(const std::function<void()>& b, dim3 g, dim3 blk) {
    std::unique_lock<std::mutex> lk(mu);
    body = &b; grid = g; block = blk;
    next = 0; finished = 0; total = (unsigned long long)g.x * g.y * g.z;
    ++epoch;
    cv.notify_all();
    done_cv.wait(lk, [&] { return finished == total; });
  }
};
}  // namespace
void launch(std::function<void()> body, dim3 grid, dim3 block) {
  static std::mutex one;  // launches of different host threads run one after the other
  std::lock_guard<std::mutex> lk(one);
  static Pool pool;
  if ((unsigned long long)grid.x * grid.y * grid.z == 0) return;
  pool.run(body, grid, block);
}
}  // namespace emu
