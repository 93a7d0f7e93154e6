// rccl/rccl.h for the host wavefront emulator build: the types and constants csrc/gather.cpp takes from <rccl/rccl.h> (it binds the
// functions at run time with dlopen("librccl.so")), so that gather.cpp itself — not a stand-in — is compiled into libmaelsim_emu.so and
// its N > 1 branch can run on a machine without GPUs against tools/hipemu/rccl_stub.cpp.  Test infrastructure; values as in RCCL's header.
#ifndef MSIM_EMU_RCCL_H
#define MSIM_EMU_RCCL_H
#include <hip/hip_runtime.h>
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
#endif
