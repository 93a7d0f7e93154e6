// k_mk.hip - instantiates mk_kernel<NEM, NET_RANDOM, KEYS> (txn-list-append, multi-key node, one cluster per wavefront; mk8.hip is the dense layout).
#include "sim_kernels.h"

hipError_t msim_launch_mk1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  if (mk_keys_for(kp.cfg) == 4u) MSIM_LAUNCH_NR(mk_kernel, , 4);
  MSIM_LAUNCH_NR(mk_kernel, , 8);
}

// ... and mkg_kernel<NEM, NET_RANDOM, KEYS>: the same node with several workers per node (a lane per endpoint; sim_kernel_mkg.inc)
hipError_t msim_launch_mkg(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  if (mk_keys_for(kp.cfg) == 4u) MSIM_LAUNCH_NR(mkg_kernel, , 4);
  MSIM_LAUNCH_NR(mkg_kernel, , 8);
}
