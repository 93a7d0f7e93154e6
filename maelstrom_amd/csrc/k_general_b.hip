// k_general_b.hip - instantiates sim_kernel<> (general layout) and sim_kernel_colo<> (colocated clients) for: fire-and-forget broadcast, with and without skip-sender.
// One of three units of the family (as one file the family's instantiations took over twenty minutes to compile).
#include "sim_kernels.h"
#include "k_general_launch.inc"

hipError_t msim_launch_general_b(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  hipError_t e = msim_upload_tables();
  if (e != hipSuccess) return e;
  switch (kp.cfg.node_program) {
    case MSIM_NODE_BCAST_FF: return launch<MSIM_NODE_BCAST_FF>(kp, n, lds, st);
    case MSIM_NODE_BCAST_FF_ECHOBACK: return launch<MSIM_NODE_BCAST_FF_ECHOBACK>(kp, n, lds, st);
    default: return MSIM_LAYOUT_DOES_NOT_FIT;
  }
}
