// sim_kernels.h - the one-cluster-per-wavefront simulation kernels (templates) and what they share, for the translation units that
// instantiate them (k_general_*.hip, k_wide_*.hip, k_raft.hip, k_svc.hip, k_txn.hip, k_mk.hip, k_dt.hip, k_kafka.hip, k_hat.hip) and for engine.hip,
// which needs their LDS / scratch layout constants.  A template nobody instantiates costs a parse: every unit includes all of them (the
// families share message enums and constants in include order) and compiles only its own.
#ifndef MSIM_SIM_KERNELS_H
#define MSIM_SIM_KERNELS_H
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "engine_internal.h"
#include "wave_common.h"
#include "log2_table.h"

// the Q24 log2 table of the exponential latency sampler: one copy per translation unit, uploaded by msim_upload_tables() before the
// unit's first launch on a device
static __constant__ u32 d_log2_q24[257];
static inline hipError_t msim_upload_tables() {
  MSIM_UPLOAD_ONCE(d_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));
  return hipSuccess;
}

// -ln(u), u = (r+1)/2^32, Q16, integer only
__device__ __forceinline__ u32 neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = d_log2_q24[idx], l1 = d_log2_q24[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}

// min (deadline, id) over the n envelopes of an HBM spill area.  The scan is latency-bound — with one dependent load per
// step every queued envelope costs an L2/HBM round trip — so 8 independent loads are in flight per step.
__device__ __forceinline__ void spill_min(const uint4 *q, u32 n, u64 &bk, u32 &best, bool &hit) {
  for (u32 i0 = 0; i0 < n; i0 += 8) {
    uint2 k[8];
#pragma unroll
    for (u32 t = 0; t < 8; t++) k[t] = *reinterpret_cast<const uint2 *>(&q[min(i0 + t, n - 1)]);
#pragma unroll
    for (u32 t = 0; t < 8; t++) {
      const u64 kk = ((u64)k[t].x << 32) | k[t].y;
      if (i0 + t < n && kk < bk) { bk = kk; best = i0 + t; hit = true; }
    }
  }
}

// merges a W-word replicate snapshot (HBM scratch) into a node's state (LDS): `or` for sets, element-wise max for counters.
// Both merges are idempotent, so the tail of the last batch re-reads word W-1 instead of branching, and B independent loads
// are in flight per step (one dependent load per word made every replicate delivery cost W HBM round trips).
template <bool IS_MAX, int B = 16>
__device__ __forceinline__ void merge_snapshot(u32 *mine, const u32 *snap, u32 W) {
  for (u32 w0 = 0; w0 < W; w0 += B) {
    u32 v[B];
#pragma unroll
    for (u32 t = 0; t < B; t++) v[t] = snap[min(w0 + t, W - 1)];
#pragma unroll
    for (u32 t = 0; t < B; t++) { const u32 i = min(w0 + t, W - 1); mine[i] = IS_MAX ? max(mine[i], v[t]) : (mine[i] | v[t]); }
  }
}


// wide g-set / counters: replicate ticks a run can see (the snapshot slots and the per-tick unions in HBM scratch are indexed by tick)
__host__ __device__ static inline uint32_t wide_crdt_ticks(const msim_config &c) {
  return (uint32_t)(((uint64_t)c.time_limit_ms + c.quiesce_ms + 2ull * c.client_timeout_ms) / 5000 + 3);
}
// ... and the LDS a wide CRDT cluster keeps per node for the replicates it has received and not merged yet (sim_kernel_wide.inc: WPEND words)
#define WPEND 8u   /* {mask x 4, tick, arrived, hold time, hold count}: the last two belong to the quiet windows */
// ... and the table of its quiet windows (sim_kernel_wide.inc: per millisecond of the next 1024, how many deliveries some node has made ahead of the rounds at that time)
#define WIDE_QW_WORDS 1024u
// ... and in HBM scratch, per replicate tick of a wide g-set cluster: the frequent elements (one word per word of the sets), the holders of rare ones (four) and a flag
#define WIDE_RARE_MAX 16u
__host__ __device__ static inline uint64_t wide_sparse_words(uint64_t ticks, uint64_t W) { return ((ticks * W + 3) & ~3ull) + ticks * W * 4 + ((ticks + 3) & ~3ull); }
// ... and for every wide cluster the client state that lives in LDS (sim_kernel_wide.inc CL(): WIDE_CLW words per pair + a dummy entry)
#define WIDE_CLW 11u
static inline size_t wide_client_bytes(const msim_config &c) { return c.n_nodes > 32 ? (((size_t)c.n_nodes + 1) * WIDE_CLW * 4 + 15) & ~(size_t)15 : 0; }
static inline size_t wide_pending_bytes(const msim_config &c) {
  return c.n_nodes > 32 && (c.node_program == MSIM_NODE_G_SET || c.node_program == MSIM_NODE_PN_COUNTER) ? (size_t)c.n_nodes * WPEND * 4 + WIDE_QW_WORDS * 4 : 0;
}

#include "sim_kernel_general.inc"
#include "sim_kernel_colo.inc"
#include "sim_kernel_raft.inc"
#include "sim_kernel_wide.inc"
#include "sim_kernel_txn.inc"
#include "sim_kernel_mk.inc"
#include "sim_kernel_dt.inc"
#include "sim_kernel_dtg.inc"
#include "sim_kernel_txng.inc"
#include "sim_kernel_mkg.inc"
#include "sim_kernel_hat.inc"
#include "sim_kernel_hatg.inc"
#include "sim_kernel_kafka.inc"
#include "sim_kernel_kafkag.inc"
#include "sim_kernel_svc.inc"

// Which wide clusters keep their nodes' sets in LDS (SETL) instead of HBM scratch.  Round 3 took the layout for g-set when sets + client
// inboxes fit 40 KiB (a set union was a ds_or instead of an L2 atomic: cfg3 417 -> 348 ms per 16384 clusters).  Since replicate deliveries
// are noted and merged when the node's state is next looked at (round 4, sim_kernel_wide.inc: a union per tick instead of N - 1 snapshots)
// a set is rarely touched, and the 17.6 KiB of sets per cluster cost more in resident wavefronts than they save: cfg3 274 ms with the sets in
// LDS, 207 ms in HBM scratch (5 % loss: 410 / 312 ms, 50 %: 283 / 215 ms; profiles/r04f_wide_sets.txt).  HBM scratch is the default now,
// MSIM_DEV_FLAGS bit 14 asks for the LDS layout (both are parity-tested).
static inline bool wide_sets_in_lds(const msim_config &c, uint32_t dev_flags) {
  if (c.n_nodes <= 32 || !(dev_flags & 0x4000u)) return false;
  if (c.node_program != MSIM_NODE_G_SET) return false;
  const size_t bytes = ((size_t)c.n_nodes * c.inbox_capacity + (size_t)c.n_nodes * CLIENT_INBOX_CAP) * 16 + (size_t)c.n_nodes * (c.max_values / 32) * 4 + wide_pending_bytes(c) + wide_client_bytes(c) + (c.nemesis_mask ? 512 : 0) + 16;
  return bytes <= 40 * 1024;
}

// the <NEM, NET_RANDOM, ...> instantiation of a kernel template for this configuration, launched with `lds` bytes of dynamic LDS
#define MSIM_LAUNCH_NR(kernel_, ...)                                                                                        \
  do {                                                                                                                      \
    const bool rnd_ = kp.cfg.latency_dist != MSIM_LAT_CONSTANT || kp.cfg.p_loss_q32 != 0;                                   \
    void (*fn_)(const KParams) = kp.cfg.nemesis_mask ? (rnd_ ? kernel_<true, true __VA_ARGS__> : kernel_<true, false __VA_ARGS__>)         \
                                                     : (rnd_ ? kernel_<false, true __VA_ARGS__> : kernel_<false, false __VA_ARGS__>);      \
    hipError_t e_ = msim_upload_tables();                                                                                   \
    if (e_ == hipSuccess && lds > 64 * 1024) e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(fn_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e_ != hipSuccess) return e_;                                                                                        \
    hipLaunchKernelGGL(fn_, dim3(n), dim3(64), lds, st, kp);                                                                \
    return hipGetLastError();                                                                                               \
  } while (0)

#endif
