// k_kafka.hip - instantiates kafka_kernel<NEM, NET_RANDOM>.
#include "sim_kernels.h"

hipError_t msim_launch_kafka1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(kafka_kernel);
}

// ... and kafkag_kernel<NEM, NET_RANDOM>: the same workload with several workers per node (a lane per endpoint; sim_kernel_kafkag.inc)
hipError_t msim_launch_kafkag(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(kafkag_kernel);
}
