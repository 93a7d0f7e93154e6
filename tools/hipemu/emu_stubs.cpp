// emu_stubs.cpp — the RCCL gather (csrc/gather.cpp) has no meaning under the wavefront emulator: its entry points refuse.
#include "../../maelstrom_amd/csrc/engine_internal.h"
void msim_gather_free(msim_ctx *) {}
extern "C" int msim_comm_unique_id(unsigned char *) { return MSIM_E_UNSUPPORTED; }
extern "C" int msim_comm_init(msim_ctx *, const unsigned char *, int, int) { return MSIM_E_UNSUPPORTED; }
extern "C" int msim_gather(msim_ctx *, int, msim_gathered *) { return MSIM_E_UNSUPPORTED; }
