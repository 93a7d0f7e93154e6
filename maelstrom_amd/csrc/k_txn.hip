// k_txn.hip - instantiates txn_kernel<NEM, NET_RANDOM> (txn-list-append, single-root node, one cluster per wavefront; txn8.hip is the dense layout).
#include "sim_kernels.h"

hipError_t msim_launch_txn1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(txn_kernel);
}
