// k_wide_bcast.hip - instantiates sim_kernel_wide<> (33..127 nodes, one cluster per wavefront, two node / client pairs per lane) for: fire-and-forget broadcast.
#include "sim_kernels.h"
#include "k_wide_launch.inc"

hipError_t msim_launch_wide_bcast(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  hipError_t e = msim_upload_tables();
  if (e != hipSuccess) return e;
  switch (kp.cfg.node_program) {
    case MSIM_NODE_BCAST_FF: return launch_wide<1>(nullptr, kp, n, lds, st);
    case MSIM_NODE_BCAST_FF_ECHOBACK: return launch_wide<1>(nullptr, kp, n, lds, st);
    default: return MSIM_LAYOUT_DOES_NOT_FIT;
  }
}
