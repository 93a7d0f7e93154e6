// k_hat.hip - instantiates hat_kernel<NEM, NET_RANDOM> (txn-rw-register, one cluster per wavefront; hat8.hip is the dense layout).
#include "sim_kernels.h"

hipError_t msim_launch_hat1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(hat_kernel);
}

// ... and hatg_kernel<NEM, NET_RANDOM>: the same node with several workers per node (a lane per endpoint; sim_kernel_hatg.inc)
hipError_t msim_launch_hatg(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(hatg_kernel);
}
