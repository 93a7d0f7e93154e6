// k_svc.hip - instantiates svc_kernel<NEM, NET_RANDOM, TSO> (the lin-kv proxy over a key-value service; unique-ids over lin-tso).
#include "sim_kernels.h"

hipError_t msim_launch_svc1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  if (kp.cfg.node_program == MSIM_NODE_TSO_IDS) MSIM_LAUNCH_NR(svc_kernel, , true);
  MSIM_LAUNCH_NR(svc_kernel, , false);
}
