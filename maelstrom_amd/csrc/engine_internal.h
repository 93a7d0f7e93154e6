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
  hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;   // gather.cpp's own timing events
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
  void *d_compact2 = nullptr; uint64_t *d_off2 = nullptr; size_t cap_compact2 = 0, cap_off2 = 0;   // the payload's own pair (both compactions are queued before the first copy)
  bool fetch_pending = false;      // msim_fetch_begin has queued the copies; msim_fetch waits for them
  size_t cap_compact = 0, cap_off = 0, cap_h_rows = 0, cap_h_payload = 0, cap_h_journal = 0, cap_h_meta = 0;
  // multi-GPU gather (gather.cpp): RCCL communicator of this rank, the rank's compacted slabs, the root's receive buffers
  void *comm = nullptr; int comm_rank = 0, comm_world = 1;
  void *d_grows = nullptr, *d_gpay = nullptr; uint64_t *d_goff = nullptr; size_t cap_grows = 0, cap_gpay = 0, cap_goff = 0;
  uint64_t g_row_units = 0, g_pay_words = 0;   // compacted sizes of the last msim_compact_on_device
  void *d_all[4] = {nullptr, nullptr, nullptr, nullptr}; size_t cap_all[4] = {0, 0, 0, 0};   // rows, payload, meta, stats of every rank (root)
  uint64_t *d_sizes = nullptr;   // world x 4 u64 for the size all-gather
  bool fetched = false, checked = false, check_fetched = false, ran = false;
  float sim_ms = 0.f, check_ms = 0.f;
  bool sim_ms_stale = false;        // the last launch was msim_run_async: sim_ms is read from ev0 / ev1 on demand
  uint32_t dev_flags = 0;           // msim_set_dev_flags: ORed with the MSIM_DEV_FLAGS of the environment (developer switches)
  uint32_t txn_big = 0;             // txn_check_dev.hip: histories of the last check whose tables did not fit LDS (HBM-table kernel)
  uint32_t lin_host_rechecks = 0;   // lin_check_dev.hip: histories of the last check the host search had to finish
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

// The developer switches in force for a context: those set through msim_set_dev_flags ORed with MSIM_DEV_FLAGS of the environment
// (read once per process).  0x100 round limit x20, 0x200 one cluster per wavefront, 0x400 fail instead of falling back to it,
// 0x800 checkers on the host cores, 0x1000 time the checkers' passes on stderr.
static inline uint32_t msim_dev_flags(const msim_ctx *ctx) {
  static const uint32_t env = []() { const char *e = std::getenv("MSIM_DEV_FLAGS"); return e ? (uint32_t)std::strtoul(e, nullptr, 0) : 0u; }();   // (decimal, 0x.. or 0..)
  return env | (ctx ? ctx->dev_flags : 0u);
}

// engine.hip: compacts the used prefix of every instance's row / payload slab into ctx->d_grows / ctx->d_gpay on the device
// (instance order); *row_units = 16-byte rows, *pay_words = u32 words.  Synchronises ctx->stream.
int msim_compact_on_device(msim_ctx *ctx, uint64_t *row_units, uint64_t *pay_words);
// gather.cpp
void msim_gather_free(msim_ctx *ctx);
// checker.hip
int msim_check_launch(msim_ctx *ctx);
// lin_check.cpp (host search) / lin_check_dev.hip (one wavefront per history)
int msim_check_lin_kv_host(msim_ctx *ctx);
int msim_check_lin_kv_device(msim_ctx *ctx);
// txn_check.cpp (host analysis) / txn_check_dev.hip (device: proves a list-append history clean, else hands it to the host)
int msim_check_txn_host(msim_ctx *ctx);
int msim_check_txn_device(msim_ctx *ctx);
// rw_check_dev.hip (device: proves a rw-register history free of what the consistency model proscribes, else hands it to the host)
int msim_check_rw_device(msim_ctx *ctx);
// kafka_check.cpp (host)
int msim_check_kafka_host(msim_ctx *ctx);
// kafka_check_dev.hip (device pass + host checker for what it cannot prove clean)
int msim_check_kafka_device(msim_ctx *ctx);
// unique_check_dev.hip, pn_check_dev.hip
int msim_check_unique_device(msim_ctx *ctx);
int msim_check_pn_device(msim_ctx *ctx);
// txn_check.cpp
// pn_check.cpp
int msim_check_pn_host(msim_ctx *ctx);
int msim_check_unique_host(msim_ctx *ctx);

// guard.cpp: the library's device allocator — hipMalloc / hipFree unless MSIM_GUARD asks for fenced slabs (developer's electric fence for HBM)
hipError_t msim_dev_malloc_impl(void **out, size_t bytes);
hipError_t msim_dev_free(void *ptr);
template <typename T> static inline hipError_t msim_dev_malloc(T **out, size_t bytes) { return msim_dev_malloc_impl(reinterpret_cast<void **>(out), bytes); }

#define MSIM_HIP_TRY(ctx, call)                                                        \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
      return MSIM_E_HIP;                                                               \
    }                                                                                  \
  } while (0)

#endif
