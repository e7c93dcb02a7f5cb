// TEST TOOL (tests/test_cpu_parallel.py): the RCCL stand-in's strict rendezvous mode against a deliberately wrong and a right
// exchange between two ranks (threads). "bad": each rank sends in one group and receives in the next — the buffered mailbox lets
// it through, real RCCL hangs, EMU_RCCL_STRICT=1 must report it; "good": send and receive in one group.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#undef dlopen
#undef dlsym
#undef dlerror

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
typedef ncclResult_t (*InitAll)(ncclComm_t*, int, const int*);
typedef ncclResult_t (*Grp)();
typedef ncclResult_t (*SendF)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*RecvF)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
int main(int argc, char** argv) {
  const bool bad = argc > 1 && !std::strcmp(argv[1], "bad");
  const InitAll initAll = (InitAll)emu_rccl_sym("ncclCommInitAll");
  const Grp gs = (Grp)emu_rccl_sym("ncclGroupStart"), ge = (Grp)emu_rccl_sym("ncclGroupEnd");
  const SendF snd = (SendF)emu_rccl_sym("ncclSend");
  const RecvF rcv = (RecvF)emu_rccl_sym("ncclRecv");
  ncclComm_t comms[2];
  initAll(comms, 2, nullptr);
  int rc[2] = {0, 0};
  auto rank = [&](int r) {
    std::vector<int> out(1000, r + 1), in(1000, 0);
    if (bad) {
      gs(); snd(out.data(), 4000, ncclChar, 1 - r, comms[r], nullptr); if (ge() != ncclSuccess) { rc[r] = 1; return; }
      gs(); rcv(in.data(), 4000, ncclChar, 1 - r, comms[r], nullptr); if (ge() != ncclSuccess) { rc[r] = 1; return; }
    } else {
      gs();
      snd(out.data(), 4000, ncclChar, 1 - r, comms[r], nullptr);
      rcv(in.data(), 4000, ncclChar, 1 - r, comms[r], nullptr);
      if (ge() != ncclSuccess) { rc[r] = 1; return; }
    }
    if (in[0] != 2 - r || in[999] != 2 - r) rc[r] = 2;
  };
  std::thread a(rank, 0), b(rank, 1);
  a.join();
  b.join();
  std::printf("%s: rc %d %d\n", bad ? "bad" : "good", rc[0], rc[1]);
  return rc[0] || rc[1];
}
