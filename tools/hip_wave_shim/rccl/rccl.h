// DEVELOPER / TEST TOOL: the RCCL names comm.cpp needs to compile under tools/hip_wave_shim (it resolves librccl with
// dlopen at run time; nothing here is callable).
#pragma once
#include <cstddef>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
