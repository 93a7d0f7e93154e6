// k_raft.hip - instantiates raft_kernel<NEM, NET_RANDOM> (lin-kv over the Raft node, one cluster per wavefront; raft4.hip is the dense layout).
#include "sim_kernels.h"

hipError_t msim_launch_raft1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(raft_kernel);
}
