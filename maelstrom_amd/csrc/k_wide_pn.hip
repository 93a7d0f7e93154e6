// k_wide_pn.hip - instantiates sim_kernel_wide<> (33..127 nodes, one cluster per wavefront, two node / client pairs per lane) for: the PN / G counters.
#include "sim_kernels.h"
#include "k_wide_launch.inc"

hipError_t msim_launch_wide_pn(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  hipError_t e = msim_upload_tables();
  if (e != hipSuccess) return e;
  switch (kp.cfg.node_program) {
    case MSIM_NODE_PN_COUNTER: return launch_wide<4>(nullptr, kp, n, lds, st);
    default: return MSIM_LAYOUT_DOES_NOT_FIT;
  }
}
