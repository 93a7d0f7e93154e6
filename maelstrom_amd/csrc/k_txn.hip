// k_txn.hip - instantiates txn_kernel<NEM, NET_RANDOM> (txn-list-append, single-root node, one cluster per wavefront; txn8.hip is the dense layout).
#include "sim_kernels.h"

hipError_t msim_launch_txn1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(txn_kernel);
}

// ... and txng_kernel<NEM, NET_RANDOM>: the same node with several workers per node (a lane per endpoint; sim_kernel_txng.inc)
hipError_t msim_launch_txng(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(txng_kernel);
}
