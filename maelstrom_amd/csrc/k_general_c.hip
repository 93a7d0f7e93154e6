// k_general_c.hip - instantiates sim_kernel<> (general layout) and sim_kernel_colo<> (colocated clients) for: acknowledged gossip with retries, rpc-to-all broadcast.
// One of three units of the family (as one file the family's instantiations took over twenty minutes to compile).
#include "sim_kernels.h"
#include "k_general_launch.inc"

hipError_t msim_launch_general_c(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  hipError_t e = msim_upload_tables();
  if (e != hipSuccess) return e;
  switch (kp.cfg.node_program) {
    case MSIM_NODE_BCAST_ACK_RETRY: return launch<MSIM_NODE_BCAST_ACK_RETRY>(kp, n, lds, st);
    case MSIM_NODE_BCAST_RPC_ALL: return launch<MSIM_NODE_BCAST_RPC_ALL>(kp, n, lds, st);
    default: return MSIM_LAYOUT_DOES_NOT_FIT;
  }
}
