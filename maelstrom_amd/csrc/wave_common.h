// wave_common.h — device-side definitions shared by the simulation kernels of libmaelsim (engine.hip, duo.hip):
// message / phase enums, the kernel parameter block, the counter-based RNG, the 64-lane DPP
// primitives and the topology builders (broadcast.clj:40-185).  gfx950 only.
#ifndef MSIM_WAVE_COMMON_H
#define MSIM_WAVE_COMMON_H
#include <hip/hip_runtime.h>

#include "engine_internal.h"

#define INF 0xFFFFFFFFu
#define STAGE_ROWS 128u
#define CLIENT_INBOX_CAP 2u
#define ROUND_LIMIT 50000000u

// message types (doc/protocol.md, doc/workloads.md)
enum { M_INIT = 1, M_INIT_OK, M_TOPOLOGY, M_TOPOLOGY_OK, M_ECHO, M_ECHO_OK, M_BROADCAST, M_BROADCAST_OK,
       M_READ, M_READ_OK, M_ADD, M_ADD_OK, M_REPLICATE };
enum { M_GENERATE = 25, M_GENERATE_OK = 26 };  // unique-ids (after the raft and txn types, include/maelsim.h MSIM_M_*)
// RNG streams
enum { S_GEN = 1, S_GEN2 = 2, S_LATENCY = 4, S_LOSS = 5,
       S_NEM_STAGGER = 7, S_NEM_SPEC = 8, S_NEM_SHUFFLE = 9, S_NEM_PICK = 10 };
enum { PH_INIT, PH_INIT_WAIT, PH_TOPO, PH_TOPO_WAIT, PH_MAIN_START, PH_MAIN, PH_DRAIN, PH_NEM_FINAL,
       PH_SLEEP, PH_FINAL, PH_FINAL_WAIT, PH_DONE };
enum { K_NONE = 0, K_INIT, K_TOPO, K_OP };

struct KParams {
  msim_config cfg;
  u64 first_instance;
  msim_op *rows;
  u32 *payload;
  msim_net_stats *stats;
  msim_inst_meta *meta;
  u32 *scratch;
  u64 scratch_words;  // per instance
  uint4 *journal;     // n * journal_capacity events, or null
  u32 N, C, CS, W;
  u32 cap_node, spill_cap;
  u64 spill_off;  // word offset of the spill area inside the per-instance scratch
  u32 off_inbox, off_seen, off_misc;  // LDS byte offsets
  u32 gen_period2_us, nem_period2_us;
  u32 raft_log_cap;   // raft: entries per node log
  u32 dev_flags;      // developer switches (env MSIM_DEV_FLAGS): 1 = cascade rounds inline, 2 = no lone-operation path
  u32 mk_tcap, mk_ccap;   // multi-key transactional node: thunks a node may create; slots of a node's thunk cache (power of two)
};

// ---- RNG (counter-based) ---------------------------------------------------------
__device__ __host__ __forceinline__ u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ u64 draw64(u64 key, u32 stream, u64 ctr) {
  const u64 x = ((u64)stream << 48) | ctr;
  return mix64(key + x * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ u32 draw32(u64 key, u32 stream, u64 ctr) { return (u32)(draw64(key, stream, ctr) >> 32); }
__device__ __forceinline__ u32 scale32(u32 r, u32 n) { return __umulhi(r, n); }

// ---- wave primitives (64 lanes; DPP on gfx950) ---------------------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ u32 dpp_mov(u32 old, u32 src) {
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, BOUND);
}
__device__ __forceinline__ u32 rdlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }

// min over the 64 lanes, result uniform
__device__ __forceinline__ u32 wave_min(u32 v) {
  v = min(v, dpp_mov<0xB1, 0xF, 0xF, false>(v, v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_mov<0x4E, 0xF, 0xF, false>(v, v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_mov<0x141, 0xF, 0xF, false>(v, v));  // row_half_mirror
  v = min(v, dpp_mov<0x140, 0xF, 0xF, false>(v, v));  // row_mirror
  return min(min(rdlane(v, 0), rdlane(v, 16)), min(rdlane(v, 32), rdlane(v, 48)));
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  v += dpp_mov<0x111, 0xF, 0xF, true>(0, v);   // row_shr:1
  v += dpp_mov<0x112, 0xF, 0xF, true>(0, v);   // row_shr:2
  v += dpp_mov<0x114, 0xF, 0xF, true>(0, v);   // row_shr:4
  v += dpp_mov<0x118, 0xF, 0xF, true>(0, v);   // row_shr:8
  v += dpp_mov<0x142, 0xA, 0xF, false>(0, v);  // row_bcast:15 -> rows 1,3
  v += dpp_mov<0x143, 0xC, 0xF, false>(0, v);  // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) { return rdlane(wave_incl_scan(v), 63); }
// inclusive prefix sum over lanes 0..31 (node lanes); lanes >= 32 hold garbage
__device__ __forceinline__ u32 scan32(u32 v) {
  v += dpp_mov<0x111, 0xF, 0xF, true>(0, v);
  v += dpp_mov<0x112, 0xF, 0xF, true>(0, v);
  v += dpp_mov<0x114, 0xF, 0xF, true>(0, v);
  v += dpp_mov<0x118, 0xF, 0xF, true>(0, v);
  v += dpp_mov<0x142, 0xA, 0xF, false>(0, v);  // row_bcast:15 -> row 1 (and 3)
  return v;
}

// tells the compiler a value is dead here (freeze of undef): a register that is only meaningful inside a round must not
// be carried around the round loops as a PHI
__device__ __forceinline__ void forget(u32 &v) { v = __builtin_nondeterministic_value(v); }
__device__ __forceinline__ void forget(uint4 &v) { forget(v.x); forget(v.y); forget(v.z); forget(v.w); }
// hides how a per-lane value was computed from the optimizer: what is derived from it is recomputed where it is used instead of being
// hoisted out of the round loop and held in registers for its whole run (loop-invariant code motion does not weigh register pressure)
#ifdef MSIM_HIPEMU
#define MSIM_OPAQUE(x_) asm volatile("" : "+r"(x_))
#else
#define MSIM_OPAQUE(x_) asm volatile("" : "+v"(x_))
#endif
// value of `v` in lane `l` (per-lane l; every lane must execute this: ds_bpermute_b32)
__device__ __forceinline__ u32 lane_get(u32 v, u32 l) { return (u32)__builtin_amdgcn_ds_bpermute((int)(l << 2), (int)v); }

// ---- topologies (broadcast.clj:40-185): adjacency mask of node a, N <= 32 -------------------------
__device__ __forceinline__ u32 topo_adj(u32 topology, u32 n, u32 a) {
  u32 m = 0;
  switch (topology) {
    case MSIM_TOPO_GRID: {
      u32 side = 1; while (side * side < n) side++;
      const u32 i = a / side, j = a % side;
      if (j + 1 < side && a + 1 < n) m |= 1u << (a + 1);
      if (j > 0) m |= 1u << (a - 1);
      if (a + side < n) m |= 1u << (a + side);
      if (i > 0) m |= 1u << (a - side);
    } break;
    case MSIM_TOPO_LINE:
      if (a + 1 < n) m |= 1u << (a + 1);
      if (a > 0) m |= 1u << (a - 1);
      break;
    case MSIM_TOPO_TOTAL: m = ((n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1)) & ~(1u << a)); break;
    default: {
      const u32 b = topology == MSIM_TOPO_TREE2 ? 2 : topology == MSIM_TOPO_TREE3 ? 3 : 4;
      if (a > 0) m |= 1u << ((a - 1) / b);
      for (u32 c = 1; c <= b; c++) if (b * a + c < n) m |= 1u << (b * a + c);
    }
  }
  return m;
}

// Orders this wavefront's LDS traffic across its lanes (a lane reads what another lane of the SAME wavefront wrote): LDS operations
// of one wavefront execute in order, so all it takes is to keep the compiler from moving them — no s_barrier, and above all no
// s_waitcnt vmcnt(0), which `__syncthreads()` implies and which would wait for every global load still in flight (prefetches).
// Only for kernels whose workgroup is one wavefront.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What a multi-cluster launcher returns when the cluster state does not fit its layout (the caller then runs the one-cluster
// kernels): NOT a HIP error code, so that a genuine hipErrorInvalidValue of a launch is reported instead of silently falling back.
static const hipError_t MSIM_LAYOUT_DOES_NOT_FIT = static_cast<hipError_t>(0x7F01);
// Uploads a __constant__ table once per device and process: hipMemcpyToSymbol synchronises with the null stream, which a launch on
// a caller's stream (msim_run_async) must not do for every batch.
#include <atomic>
#define MSIM_UPLOAD_ONCE(sym, src, bytes)                                                                        \
  do {                                                                                                           \
    static std::atomic<unsigned long long> done_{0};                                                             \
    int d_ = 0;                                                                                                  \
    hipError_t e_ = hipGetDevice(&d_);                                                                           \
    if (e_ != hipSuccess) return e_;                                                                             \
    if (!((done_.load(std::memory_order_acquire) >> (d_ & 63)) & 1ull)) {                                        \
      e_ = hipMemcpyToSymbol(HIP_SYMBOL(sym), src, bytes);                                                       \
      if (e_ != hipSuccess) return e_;                                                                           \
      done_.fetch_or(1ull << (d_ & 63), std::memory_order_release);                                              \
    }                                                                                                            \
  } while (0)

// k_*.hip: the one-cluster-per-wavefront kernels (sim_kernels.h), one launcher per family / unit; MSIM_LAYOUT_DOES_NOT_FIT for a node
// program the unit does not hold
hipError_t msim_launch_general_a(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);   // echo, flake ids, g-set, the counters
hipError_t msim_launch_general_b(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);   // fire-and-forget broadcast
hipError_t msim_launch_general_c(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);   // acknowledged gossip, rpc-to-all
hipError_t msim_launch_wide_gset(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_wide_bcast(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_wide_ack(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_wide_pn(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_raft1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_svc1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_txn1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_mk1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_dt1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_dtg(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_txng(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_mkg(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_kafka1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_hat1(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_hatg(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
hipError_t msim_launch_kafkag(const KParams &kp, uint32_t n, size_t lds, hipStream_t st);
// duo.hip: two clusters per wavefront (fire-and-forget broadcast, constant latency, colocated clients)
bool msim_duo_eligible(const msim_config &c);
hipError_t msim_launch_duo(const KParams &kp, uint32_t n, hipStream_t st);
// raft4.hip: four Raft clusters per wavefront (lin-kv over the Raft node program, clusters of <= 16 endpoints)
bool msim_raft4_eligible(const msim_config &c);
uint64_t msim_raft4_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_raft4(const KParams &kp, uint32_t n, hipStream_t st);
// svc4.hip: four lin-kv-proxy clusters per wavefront (lin_kv_proxy.rb over lin-kv / lww-kv, clusters of <= 16 endpoints incl. the service)
bool msim_svc4_eligible(const msim_config &c);
uint64_t msim_svc4_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_svc4(const KParams &kp, uint32_t n, hipStream_t st);
// txng4.hip: four clusters of the single-root txn-list-append node with several workers per node per wavefront (nodes + workers + lin-kv <= 16)
bool msim_txng4_eligible(const msim_config &c);
uint64_t msim_txng4_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_txng4(const KParams &kp, uint32_t n, hipStream_t st);
// dtg4.hip: four clusters of the Datomic-style txn-list-append node with several workers per node per wavefront (nodes + workers + lin-kv + lww-kv <= 16)
bool msim_dtg4_eligible(const msim_config &c);
uint64_t msim_dtg4_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_dtg4(const KParams &kp, uint32_t n, hipStream_t st);
// txn8.hip: eight txn-list-append clusters per wavefront (single-root node over lin-kv, clusters of <= 8 lanes)
bool msim_txn8_eligible(const msim_config &c);
uint64_t msim_txn8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_txn8(const KParams &kp, uint32_t n, hipStream_t st);
// hat8.hip: sixteen / eight txn-rw-register clusters per wavefront (the highly-available-transactions node, clusters of <= 4 / 8 lanes)
bool msim_hat8_eligible(const msim_config &c);
uint64_t msim_hat8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_hat8(const KParams &kp, uint32_t n, hipStream_t st);
// kafka8.hip: eight kafka clusters per wavefront (the kafka node over lin-kv, clusters of <= 8 lanes)
bool msim_kafka8_eligible(const msim_config &c);
uint64_t msim_kafka8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_kafka8(const KParams &kp, uint32_t n, hipStream_t st);
// uid8.hip: eight echo / unique-ids clusters per wavefront (programs whose nodes talk to their clients only, clusters of <= 8 lanes)
bool msim_uid8_eligible(const msim_config &c);
uint64_t msim_uid8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_uid8(const KParams &kp, uint32_t n, hipStream_t st);
// crdt8.hip: eight g-set / pn-counter / g-counter clusters per wavefront (clusters of <= 8 lanes, states of <= 64 words)
bool msim_crdt8_eligible(const msim_config &c);
uint64_t msim_crdt8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_crdt8(const KParams &kp, uint32_t n, hipStream_t st);
// bcast8.hip: eight broadcast clusters per wavefront (the four broadcast programs, clusters of <= 8 lanes, sets of <= 64 words)
bool msim_bcast8_eligible(const msim_config &c);
uint64_t msim_bcast8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_bcast8(const KParams &kp, uint32_t n, hipStream_t st);
// mk8.hip: eight clusters of the multi-key transactional node per wavefront (n <= 6 nodes + lin-kv + lww-kv in an 8-lane group)
bool msim_mk8_eligible(const msim_config &c);
uint64_t msim_mk8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_mk8(const KParams &kp, uint32_t n, hipStream_t st);
// dt8.hip: the Datomic-style txn-list-append node, eight clusters per wavefront
bool msim_dt8_eligible(const msim_config &c);
uint64_t msim_dt8_extra_scratch_words(const msim_config &c);
hipError_t msim_launch_dt8(const KParams &kp, uint32_t n, hipStream_t st);

#endif
