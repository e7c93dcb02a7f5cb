// DEVELOPER / TEST TOOL: the RCCL names comm.cpp needs, for the CPU emulation of the library (tools/libs360_emu.so).
// comm.cpp resolves librccl with dlopen / dlsym at run time; under this shim those two calls resolve to the in-process
// stand-in of rccl_emu.cpp instead — ranks are host threads of ONE process (the way host/TestRenderStereoPanorama
// --num_gpus runs them), a send is a buffered copy, a receive waits for the matching send. The product never sees this.
#pragma once
#include <cstddef>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclInvalidArgument = 4, ncclInternalError = 3 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
void* emu_rccl_sym(const char* name);  // rccl_emu.cpp
#define dlopen(name, flags) ((void*)1)
#define dlsym(handle, name) emu_rccl_sym(name)
#define dlerror() "emulated librccl"
