// k_wide_ack.hip - instantiates sim_kernel_wide<> (33..127 nodes, one cluster per wavefront, two node / client pairs per lane) for: acknowledged gossip, rpc-to-all broadcast.
#include "sim_kernels.h"
#include "k_wide_launch.inc"

hipError_t msim_launch_wide_ack(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  hipError_t e = msim_upload_tables();
  if (e != hipSuccess) return e;
  switch (kp.cfg.node_program) {
    case MSIM_NODE_BCAST_ACK_RETRY: return launch_wide<2>(nullptr, kp, n, lds, st);
    case MSIM_NODE_BCAST_RPC_ALL: return launch_wide<3>(nullptr, kp, n, lds, st);
    default: return MSIM_LAYOUT_DOES_NOT_FIT;
  }
}
