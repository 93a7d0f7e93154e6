// engine_internal.h — private to libmaelsim (engine.hip, checker.hip).
#ifndef MSIM_ENGINE_INTERNAL_H
#define MSIM_ENGINE_INTERNAL_H

#include <hip/hip_runtime.h>
#include <string>

#include "../../include/maelsim.h"
#include "engine_limits.h"

typedef uint32_t u32;
typedef uint64_t u64;

struct msim_ctx {
  msim_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  // device buffers of the last run
  uint32_t n_inst = 0, cap_inst = 0;
  uint64_t first_instance = 0;
  msim_op *d_rows = nullptr;
  uint32_t *d_payload = nullptr;
  msim_net_stats *d_stats = nullptr;
  msim_inst_meta *d_meta = nullptr;
  uint32_t *d_scratch = nullptr;
  uint64_t scratch_words_per_inst = 0;
  msim_check_result *d_check = nullptr;
  msim_event *d_journal = nullptr;
  // host (pinned) mirrors
  msim_op *h_rows = nullptr;
  uint32_t *h_payload = nullptr;
  msim_net_stats *h_stats = nullptr;
  msim_inst_meta *h_meta = nullptr;
  msim_check_result *h_check = nullptr;
  msim_event *h_journal = nullptr;
  uint64_t *h_row_off = nullptr, *h_pay_off = nullptr, *h_ev_off = nullptr;  // compacted offsets per instance
  // fetch path: device-side compaction buffers + grow-only capacities (bytes) of the pinned mirrors
  void *d_check_scratch = nullptr; size_t cap_check_scratch = 0;  // checker.hip: read records per instance
  void *d_compact = nullptr; uint64_t *d_off = nullptr;
  size_t cap_compact = 0, cap_off = 0, cap_h_rows = 0, cap_h_payload = 0, cap_h_journal = 0, cap_h_meta = 0;
  bool fetched = false, checked = false, check_fetched = false, ran = false;
  float sim_ms = 0.f, check_ms = 0.f;
  std::string err;
};

// Host threads one engine context may use for its host-side checkers: the hardware threads divided by the ranks sharing
// the node (LOCAL_WORLD_SIZE, set by torch.distributed.run), or MSIM_HOST_THREADS if set.
#include <cstdio>
#include <cstdlib>
#include <thread>
static inline unsigned msim_host_threads() {
  if (const char *e = std::getenv("MSIM_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) return (unsigned)v; }
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  // a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>"): more runnable threads than granted CPUs only get throttled
  if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long q = 0, per = 0;
    if (std::fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) { const unsigned lim = (unsigned)((q + per - 1) / per); if (lim >= 1 && lim < nt) nt = lim; }
    std::fclose(f);
  }
  if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) { const int v = std::atoi(e); if (v > 1) nt = nt / (unsigned)v ? nt / (unsigned)v : 1; }
  return nt;
}

// checker.hip
int msim_check_launch(msim_ctx *ctx);
// lin_check.cpp
int msim_check_lin_kv_host(msim_ctx *ctx);
// txn_check.cpp
int msim_check_txn_host(msim_ctx *ctx);
// pn_check.cpp
int msim_check_pn_host(msim_ctx *ctx);
int msim_check_unique_host(msim_ctx *ctx);

#define MSIM_HIP_TRY(ctx, call)                                                        \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
      return MSIM_E_HIP;                                                               \
    }                                                                                  \
  } while (0)

#endif
