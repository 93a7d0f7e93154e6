// k_wide_gset.hip - instantiates sim_kernel_wide<> (33..127 nodes, one cluster per wavefront, two node / client pairs per lane) for: g-set (BASELINE configs[2]).
#include "sim_kernels.h"
#include "k_wide_launch.inc"

hipError_t msim_launch_wide_gset(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  hipError_t e = msim_upload_tables();
  if (e != hipSuccess) return e;
  switch (kp.cfg.node_program) {
    case MSIM_NODE_G_SET: return launch_wide<0>(nullptr, kp, n, lds, st);
    default: return MSIM_LAYOUT_DOES_NOT_FIT;
  }
}
