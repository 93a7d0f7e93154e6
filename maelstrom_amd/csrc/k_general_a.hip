// k_general_a.hip - instantiates sim_kernel<> (general layout) and sim_kernel_colo<> (colocated clients) for: echo, flake ids, g-set and the PN / G counters.
// One of three units of the family (as one file the family's instantiations took over twenty minutes to compile).
#include "sim_kernels.h"
#include "k_general_launch.inc"

hipError_t msim_launch_general_a(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  hipError_t e = msim_upload_tables();
  if (e != hipSuccess) return e;
  switch (kp.cfg.node_program) {
    case MSIM_NODE_ECHO: return launch<MSIM_NODE_ECHO>(kp, n, lds, st);
    case MSIM_NODE_FLAKE_IDS: return launch<MSIM_NODE_FLAKE_IDS>(kp, n, lds, st);
    case MSIM_NODE_G_SET: return launch<MSIM_NODE_G_SET>(kp, n, lds, st);
    case MSIM_NODE_PN_COUNTER: return launch<MSIM_NODE_PN_COUNTER>(kp, n, lds, st);
    default: return MSIM_LAYOUT_DOES_NOT_FIT;
  }
}
