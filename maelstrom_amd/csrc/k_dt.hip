// k_dt.hip - instantiates dt_kernel<NEM, NET_RANDOM> (txn-list-append over the Datomic-style transactor node, demo/ruby/datomic_list_append.rb; one cluster per wavefront).
#include "sim_kernels.h"

hipError_t msim_launch_dt1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(dt_kernel);
}

// ... and dtg_kernel<NEM, NET_RANDOM>: the same node with several workers per node (a lane per endpoint; sim_kernel_dtg.inc)
hipError_t msim_launch_dtg(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  MSIM_LAUNCH_NR(dtg_kernel);
}
