// engine.hip — the ensemble simulation kernel and the host runtime behind include/maelsim.h.
//
// HOT PATH: for every test instance, the message queue with simulated latency / loss /
// partitions (net.clj:189-247), node execution (process.clj:136-166 collapsed to "a node consumes one
// message at a time"), the sync RPC clients (client.clj:66-172), the generator/nemesis schedule
// (core.clj:67-80) and the node programs' state-transition functions (rows a13-a15).
//
// MI355X mapping:
//   * one wavefront (one 64-thread workgroup) simulates one cluster; lane e = endpoint e
//     (lanes [0,N) = nodes n0..n{N-1}, lanes [N,N+CS) = client worker slots);
//   * node sets (bitmaps) and the per-endpoint in-flight message queues live in LDS; every endpoint's
//     queue is written only by its own lane ("receiver-side pull": each round the wave walks the
//     senders with v_readlane and every addressed lane appends to its own queue) — no LDS atomics;
//   * the per-round "next event time" is a DPP min-reduction, message ids / history row indices come
//     from a DPP prefix sum + ballots, so ids and row order are canonical (endpoint order) and the
//     result is bit-identical to the sequential CPU oracle;
//   * history rows are staged in LDS and appended to HBM 64 rows (1 KiB) at a time, one 16-B store per
//     lane; read results (bitmaps) are copied LDS->HBM by the whole wave, 256 B per instruction.
//   No MFMA: this is integer/indexing work.  No CUDA/hipify/Triton layers.
//
// This file holds the host runtime and msim_run's choice of a kernel.  The one-cluster-per-wavefront kernels (sim_kernel*.inc, collected by
// sim_kernels.h) are instantiated by the k_*.hip units; denser layouts live in
// their own translation units and are taken where a configuration and the batch fit them (DESIGN.md §4.1b has the table): duo.hip (two
// broadcast clusters per wavefront, the headline), raft4.hip (four), txn8.hip / mk8.hip / hat8.hip / uid8.hip / crdt8.hip / bcast8.hip
// (eight: the transactional programs, echo / unique-ids, the CRDTs, the broadcast programs at tutorial sizes).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>


#include "sim_kernels.h"   // the layout constants of the kernels; this unit instantiates none of them (the k_*.hip units do)

// Reference (shuffle) versions, used only by the self-test to validate the DPP encodings on hardware.
__device__ u32 wave_min_ref(u32 v) { for (int o = 32; o; o >>= 1) v = min(v, (u32)__shfl_xor((int)v, o)); return v; }
__device__ u32 wave_incl_scan_ref(u32 v) {
  const u32 lane = threadIdx.x & 63;
  for (int o = 1; o < 64; o <<= 1) { u32 t = (u32)__shfl_up((int)v, o); if (lane >= (u32)o) v += t; }
  return v;
}
__global__ void wave_selftest_kernel(const u32 *in, u32 *out) {
  const u32 lane = threadIdx.x;
  const u32 v = in[blockIdx.x * 64 + lane];
  u32 bad = 0;
  bad |= wave_min(v) != wave_min_ref(v);
  bad |= wave_incl_scan(v & 0xFFFF) != wave_incl_scan_ref(v & 0xFFFF);
  bad |= wave_sum(v & 0xFF) != rdlane(wave_incl_scan_ref(v & 0xFF), 63);
  bad |= (lane < 32) && scan32(v & 0xFFFF) != wave_incl_scan_ref(v & 0xFFFF);
  bad |= lane_get(v, (lane * 7 + 3) & 63) != (u32)__shfl((int)v, (int)((lane * 7 + 3) & 63));
  out[blockIdx.x * 64 + lane] = bad;
}


// =====================================================================================================
// Host runtime
// =====================================================================================================
static void set_err(char *err, size_t n, const char *msg) { if (err && n) std::snprintf(err, n, "%s", msg); }

extern "C" int msim_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static void free_buffers(msim_ctx *c) {
  if (c->d_rows) (void)msim_dev_free(c->d_rows);
  if (c->d_payload) (void)msim_dev_free(c->d_payload);
  if (c->d_stats) (void)msim_dev_free(c->d_stats);
  if (c->d_meta) (void)msim_dev_free(c->d_meta);
  if (c->d_scratch) (void)msim_dev_free(c->d_scratch);
  if (c->d_check) (void)msim_dev_free(c->d_check);
  if (c->d_journal) (void)msim_dev_free(c->d_journal);
  if (c->h_rows) (void)hipHostFree(c->h_rows);
  if (c->h_payload) (void)hipHostFree(c->h_payload);
  if (c->h_stats) (void)hipHostFree(c->h_stats);
  if (c->h_meta) (void)hipHostFree(c->h_meta);
  if (c->h_check) (void)hipHostFree(c->h_check);
  if (c->h_journal) (void)hipHostFree(c->h_journal);
  if (c->d_check_scratch) (void)msim_dev_free(c->d_check_scratch);
  c->d_check_scratch = nullptr; c->cap_check_scratch = 0;
  if (c->d_compact) (void)msim_dev_free(c->d_compact);
  if (c->d_off) (void)msim_dev_free(c->d_off);
  if (c->d_compact2) (void)msim_dev_free(c->d_compact2);
  if (c->d_off2) (void)msim_dev_free(c->d_off2);
  c->d_compact2 = nullptr; c->d_off2 = nullptr; c->cap_compact2 = c->cap_off2 = 0;
  if (c->d_grows) (void)msim_dev_free(c->d_grows);
  if (c->d_gpay) (void)msim_dev_free(c->d_gpay);
  if (c->d_goff) (void)msim_dev_free(c->d_goff);
  c->d_grows = c->d_gpay = nullptr; c->d_goff = nullptr; c->cap_grows = c->cap_gpay = c->cap_goff = 0;
  c->d_compact = nullptr; c->d_off = nullptr;
  c->cap_compact = c->cap_off = c->cap_h_rows = c->cap_h_payload = c->cap_h_journal = c->cap_h_meta = 0;
  delete[] c->h_row_off; delete[] c->h_pay_off; delete[] c->h_ev_off;
  c->d_journal = nullptr; c->h_journal = nullptr; c->h_ev_off = nullptr;
  c->d_rows = nullptr; c->d_payload = nullptr; c->d_stats = nullptr; c->d_meta = nullptr; c->d_scratch = nullptr; c->d_check = nullptr;
  c->h_rows = nullptr; c->h_payload = nullptr; c->h_stats = nullptr; c->h_meta = nullptr; c->h_check = nullptr;
  c->h_row_off = nullptr; c->h_pay_off = nullptr;
  c->cap_inst = 0;
}

extern "C" int msim_create(const msim_config *cfg, int device, msim_ctx **out, char *err, size_t errlen) {
  if (!cfg || !out) { set_err(err, errlen, "null argument"); return MSIM_E_INVALID; }
  *out = nullptr;
  msim_config c = *cfg;
  int rc = msim_config_finalize(&c, err, errlen);
  if (rc != MSIM_OK) return rc;
  const bool dt_many = (c.node_program == MSIM_NODE_TXN_DATOMIC || c.node_program == MSIM_NODE_TXN_MULTI_KEY) && c.concurrency > c.n_nodes;   // several workers per node: dtg_kernel<> / mkg_kernel<> (endpoint per lane)
  if ((c.node_program == MSIM_NODE_TXN_MULTI_KEY || c.node_program == MSIM_NODE_TXN_DATOMIC) && !dt_many && (c.concurrency != c.n_nodes || c.n_nodes > 30)) {
    set_err(err, errlen, "multi_key_txn / datomic: one worker per node and at most 30 nodes (two service lanes), or k x node-count workers with nodes + workers + 2 <= 64");
    return MSIM_E_UNSUPPORTED;
  }
  if (dt_many && c.n_nodes + c.concurrency + 2 > 64) { set_err(err, errlen, "multi_key_txn / datomic with several workers per node: nodes + workers + 2 services <= 64"); return MSIM_E_UNSUPPORTED; }
  const bool txn_many = (c.node_program == MSIM_NODE_TXN_SINGLE_KEY || c.node_program == MSIM_NODE_KAFKA) && c.concurrency > c.n_nodes;   // txng_kernel<> / kafkag_kernel<>: + the lin-kv lane
  const uint32_t slots = (c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes) + (dt_many ? 2 : txn_many ? 1 : 0);
  // wide clusters (33..127 nodes): two node/client pairs per lane, one worker per node: the g-set CRDT and fire-and-forget broadcast
  const bool wide_prog = c.node_program == MSIM_NODE_G_SET || c.node_program == MSIM_NODE_BCAST_FF || c.node_program == MSIM_NODE_BCAST_FF_ECHOBACK ||
                         c.node_program == MSIM_NODE_BCAST_ACK_RETRY || c.node_program == MSIM_NODE_BCAST_RPC_ALL || c.node_program == MSIM_NODE_PN_COUNTER;
  const bool wide = c.n_nodes > 32 && c.n_nodes <= 127 && wide_prog && c.concurrency == c.n_nodes;
  const uint32_t svc_lanes = c.node_program == MSIM_NODE_LIN_KV_PROXY || c.node_program == MSIM_NODE_TSO_IDS ? 1 : 0;  // the service has a lane of its own after the client slots
  if (!wide && (c.n_nodes > 32 || c.n_nodes + slots + svc_lanes > 64)) {
    set_err(err, errlen, "this build maps one cluster to one wavefront: n_nodes <= 32 and n_nodes + max(concurrency, n_nodes) <= 64 "
                         "(g-set, the counters and the broadcast programs with concurrency == n_nodes: up to 127 nodes)");
    return MSIM_E_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err(err, errlen, "no HIP device visible: libmaelsim has no CPU path"); return MSIM_E_NO_DEVICE; }
  if (device < 0 || device >= ndev) { set_err(err, errlen, "device index out of range"); return MSIM_E_INVALID; }
  msim_ctx *ctx = new (std::nothrow) msim_ctx();
  if (!ctx) return MSIM_E_NOMEM;
  ctx->cfg = c; ctx->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev2);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev3);
  if (e != hipSuccess) { set_err(err, errlen, hipGetErrorString(e)); delete ctx; return MSIM_E_HIP; }
  *out = ctx;
  return MSIM_OK;
}

// multi-key transactional node: thunk ids a node may hand out (every attempt of a transaction writes its keys again: x4 for the
// retries) and the slots of its thunk cache (what it wrote + what it read: twice that, load factor 1/2)
static uint32_t mk_tcap(const msim_config &c) {
  const double ops = (double)c.rate_mhz * (double)c.time_limit_ms / 1e6 * 1.125 + 64.0;
  uint32_t t = 64; while (t < 4.0 * ops * c.max_txn_length / c.n_nodes + 64.0) t <<= 1;
  return t;
}
static uint32_t mk_ccap(const msim_config &c) { return 4 * mk_tcap(c); }

// per-instance scratch = [protocol scratch][spill area: n_nodes x spill_capacity envelopes]
static uint32_t raft_log_cap(const msim_config &c) {  // every client op is appended at most once, by the leader that takes it
  const double expected = (double)c.rate_mhz * (double)c.time_limit_ms / 1e6;
  return (uint32_t)(expected + expected / 8.0) + 64 + 8;
}
static uint64_t proto_scratch_words(const msim_config &c) {
  uint64_t w = 4;
  if (c.node_program == MSIM_NODE_RAFT) w = (uint64_t)c.n_nodes * raft_log_cap(c) * 2 + (uint64_t)c.n_nodes * R_ARENA_WORDS;
  if (c.node_program == MSIM_NODE_BCAST_ACK_RETRY) w = (uint64_t)c.n_nodes * c.max_values * 3;
  if (c.node_program == MSIM_NODE_TXN_SINGLE_KEY) w = (uint64_t)c.max_values * (c.max_writes_per_key + 1);  // elements + counts per key
  if (c.node_program == MSIM_NODE_TXN_MULTI_KEY)   // elements, counts, map position, entry version, thunk counts, thunk versions + ids, the nodes' caches, the replica bytes
    w = (uint64_t)c.max_values * (c.max_writes_per_key + 4 + 2 * (c.max_writes_per_key + 1)) + (uint64_t)c.n_nodes * mk_ccap(c) + (uint64_t)c.n_nodes * mk_tcap(c) / 4 + 4 +
        (uint64_t)c.n_nodes * ((c.concurrency > c.n_nodes ? MKG_SLOTS : MK_SLOTS) - MK_SL) * mk_slot_words(8);   // + the transaction slots that are not in LDS (several workers per node: mkg_kernel<> has MKG_SLOTS)
  if (c.node_program == MSIM_NODE_TXN_DATOMIC) w = dt_scratch_words(c);   // elements, counts, first versions, key hashes, the tree nodes, the nodes' caches, a round's write lists
  if (c.node_program == MSIM_NODE_KAFKA) w = (uint64_t)KF_KEYS * (2 * (c.max_writes_per_key + 1) + 1);   // the logs + the committed-offset lists of the keys
  if (c.node_program == MSIM_NODE_TXN_RW_HAT) {  // registers per node + txn table + pending masks (bytes) + replicate lists
    const uint64_t G = c.max_rows / 2;
    w = (uint64_t)c.n_nodes * c.max_values + 2 * G + ((uint64_t)c.n_nodes * G + 3) / 4 + c.replication_words;
  }
  if (c.node_program == MSIM_NODE_G_SET || c.node_program == MSIM_NODE_PN_COUNTER) {
    const uint64_t total_ms = (uint64_t)c.time_limit_ms + c.quiesce_ms + 2ull * c.client_timeout_ms;
    const uint64_t ticks = total_ms / 5000 + 3;
    w = ticks * c.n_nodes * (c.max_values / 32);
    // wide clusters: + the union of every tick's snapshots (wide_union_off: what a node that received all of a tick merges in one go) + the nodes' sets
    if (c.n_nodes > 32) w = ((w + ticks * (c.max_values / 32) + 3) & ~3ull) + (c.node_program == MSIM_NODE_G_SET ? wide_sparse_words(ticks, c.max_values / 32) : 0) + (((uint64_t)c.n_nodes * (c.max_values / 32) + 3) & ~3ull);   // (+ the ticks' frequent elements and rare holders: sim_kernel_wide.inc SPARSE)
  }
  const bool bcast_ff = c.node_program == MSIM_NODE_BCAST_FF || c.node_program == MSIM_NODE_BCAST_FF_ECHOBACK;
  const bool bcast_rpc = c.node_program == MSIM_NODE_BCAST_ACK_RETRY || c.node_program == MSIM_NODE_BCAST_RPC_ALL;
  if (bcast_ff && c.n_nodes > 32) w = 4 + (((uint64_t)c.n_nodes * (c.max_values / 32) + 3) & ~3ull);  // wide clusters: the nodes' sets
  // wide clusters, acknowledged gossip: 128-bit unacked masks per (node, value), the retry FIFO, the nodes' sets
  if (bcast_rpc && c.n_nodes > 32) w = (uint64_t)c.n_nodes * c.max_values * 6 + (((uint64_t)c.n_nodes * (c.max_values / 32) + 3) & ~3ull);
  return (w + 3) & ~3ull;  // keep the spill area 16-byte aligned
}
static uint64_t scratch_words(const msim_config &c) {
  const uint64_t queues = c.n_nodes + (c.node_program == MSIM_NODE_KAFKA || c.node_program == MSIM_NODE_TXN_SINGLE_KEY || c.node_program == MSIM_NODE_LIN_KV_PROXY || c.node_program == MSIM_NODE_TSO_IDS ? 1 : c.node_program == MSIM_NODE_TXN_MULTI_KEY || c.node_program == MSIM_NODE_TXN_DATOMIC ? 2 : 0);  // + the services
  uint64_t w = proto_scratch_words(c) + queues * c.spill_capacity * 4;
  if (c.node_program == MSIM_NODE_LIN_KV_PROXY || c.node_program == MSIM_NODE_TSO_IDS) w += (uint64_t)(c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes) * (R_CLIENT_CAP - PX_CL) * 4;   // svc_kernel<>: the clients' inboxes beyond PX_CL envelopes
  if (msim_raft4_eligible(c)) w += msim_raft4_extra_scratch_words(c);   // raft4.hip keeps fewer envelopes in LDS
  if (msim_svc4_eligible(c)) w += msim_svc4_extra_scratch_words(c);     // svc4.hip likewise
  if (msim_txng4_eligible(c)) w += msim_txng4_extra_scratch_words(c);   // txng4.hip likewise
  if (msim_dtg4_eligible(c)) w += msim_dtg4_extra_scratch_words(c);     // dtg4.hip likewise
  if (msim_txn8_eligible(c)) w += msim_txn8_extra_scratch_words(c);     // txn8.hip likewise
  if (msim_mk8_eligible(c)) w += msim_mk8_extra_scratch_words(c);       // mk8.hip likewise
  if (msim_dt8_eligible(c)) w += msim_dt8_extra_scratch_words(c);       // dt8.hip likewise (+ the nodes' save stacks)
  if (msim_hat8_eligible(c)) w += msim_hat8_extra_scratch_words(c);     // hat8.hip likewise
  if (msim_kafka8_eligible(c)) w += msim_kafka8_extra_scratch_words(c); // kafka8.hip likewise
  if (msim_uid8_eligible(c)) w += msim_uid8_extra_scratch_words(c);     // uid8.hip likewise
  if (msim_crdt8_eligible(c)) w += msim_crdt8_extra_scratch_words(c);   // crdt8.hip likewise
  if (msim_bcast8_eligible(c)) w += msim_bcast8_extra_scratch_words(c); // bcast8.hip likewise
  return w;
}

static int ensure_buffers(msim_ctx *ctx, uint32_t n) {
  if (n <= ctx->cap_inst) return MSIM_OK;
  free_buffers(ctx);
  const msim_config &c = ctx->cfg;
  ctx->scratch_words_per_inst = scratch_words(c);
  MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_rows, (size_t)n * c.max_rows * sizeof(msim_op)));
  MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_payload, (size_t)n * c.max_payload_words * 4));
  MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_stats, (size_t)n * sizeof(msim_net_stats)));
  MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_meta, (size_t)n * sizeof(msim_inst_meta)));
  MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_scratch, (size_t)n * ctx->scratch_words_per_inst * 4));
  MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_check, (size_t)n * sizeof(msim_check_result)));
  if (c.journal_capacity) MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_journal, (size_t)n * c.journal_capacity * sizeof(msim_event)));
  ctx->cap_inst = n;
  return MSIM_OK;
}

static int run_impl(msim_ctx *ctx, uint64_t first, uint32_t n, hipStream_t st, bool blocking) {
  if (!ctx) return MSIM_E_INVALID;
  if (n == 0) { ctx->err = "n_instances must be > 0"; return MSIM_E_INVALID; }
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // msim_fetch_begin has queued PCIe copies on the context's stream: they are over before this launch (which may sit on the caller's own
  // stream, msim_run_async) drops the claim to them — the pinned mirrors they write may be regrown by the next fetch
  if (ctx->fetch_pending) { MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); ctx->fetch_pending = false; }
  int rc = ensure_buffers(ctx, n);
  if (rc != MSIM_OK) return rc;
  const msim_config &c = ctx->cfg;
  KParams kp;
  std::memset(&kp, 0, sizeof kp);
  kp.cfg = c; kp.first_instance = first;
  kp.rows = ctx->d_rows; kp.payload = ctx->d_payload; kp.stats = ctx->d_stats; kp.meta = ctx->d_meta;
  kp.scratch = ctx->d_scratch; kp.scratch_words = ctx->scratch_words_per_inst;
  kp.journal = reinterpret_cast<uint4 *>(ctx->d_journal);
  kp.N = c.n_nodes; kp.C = c.concurrency; kp.CS = c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes;
  kp.W = c.max_values / 32;
  kp.cap_node = c.inbox_capacity; kp.spill_cap = c.spill_capacity; kp.spill_off = proto_scratch_words(c);
  const uint64_t period_us = 1000000000ull / (c.rate_mhz ? c.rate_mhz : 1);
  if (2 * period_us > 0xFFFFFFFFull || 2000ull * c.nemesis_interval_ms > 0xFFFFFFFFull) { ctx->err = "rate too low / nemesis interval too long for u32 microseconds"; return MSIM_E_INVALID; }
  kp.gen_period2_us = (u32)(2 * period_us);
  kp.nem_period2_us = (u32)(2000ull * c.nemesis_interval_ms);
  const bool is_raft = c.node_program == MSIM_NODE_RAFT;
  kp.raft_log_cap = is_raft ? raft_log_cap(c) : 0;
  kp.dev_flags = msim_dev_flags(ctx);
  const bool wide = c.n_nodes > 32;
  const bool wide_setl = wide_sets_in_lds(c, kp.dev_flags);   // (no row staging in that layout)
  const bool wide_crdt = wide && (c.node_program == MSIM_NODE_G_SET || c.node_program == MSIM_NODE_PN_COUNTER);   // (rows unstaged: sim_kernel_wide.inc ROWS_DIRECT)
  size_t off = (wide_setl || wide_crdt) ? 0 : (wide ? WIDE_STAGE_ROWS : STAGE_ROWS) * 16;
  const bool is_txn = c.node_program == MSIM_NODE_TXN_SINGLE_KEY, is_px = c.node_program == MSIM_NODE_LIN_KV_PROXY || c.node_program == MSIM_NODE_TSO_IDS;
  const bool is_hat = c.node_program == MSIM_NODE_TXN_RW_HAT, is_mk = c.node_program == MSIM_NODE_TXN_MULTI_KEY, is_kf = c.node_program == MSIM_NODE_KAFKA, is_dt = c.node_program == MSIM_NODE_TXN_DATOMIC;
  kp.mk_tcap = is_mk ? mk_tcap(c) : is_dt ? dt_tcap(c) : 0; kp.mk_ccap = is_mk ? mk_ccap(c) : 0;
  kp.off_inbox = (u32)off;
  const bool dt_many = (is_dt || is_mk) && c.concurrency > c.n_nodes;   // dtg_kernel<> / mkg_kernel<>: a lane per endpoint, a client inbox per worker slot
  const bool txn_many = (is_txn || is_kf) && c.concurrency > c.n_nodes; // txng_kernel<> / kafkag_kernel<>: likewise
  off += (is_mk || is_dt) ? ((size_t)(kp.N + 2) * kp.cap_node + (size_t)(dt_many ? kp.CS : kp.N) * T_CLIENT_CAP) * 16 : is_hat ? ((size_t)kp.N * kp.cap_node + (size_t)kp.CS * T_CLIENT_CAP) * 16   /* (hatg_kernel<>: an inbox per worker slot; CS == N otherwise) */ : is_px ? ((size_t)(kp.N + 1) * kp.cap_node + (size_t)kp.CS * PX_CL) * 16 : (is_txn || is_kf) ? ((size_t)(kp.N + 1) * kp.cap_node + (size_t)(txn_many ? kp.CS : kp.N) * T_CLIENT_CAP) * 16
                : ((size_t)kp.N * kp.cap_node + (size_t)kp.CS * (is_raft ? R_CLIENT_CAP : CLIENT_INBOX_CAP)) * 16 + wide_client_bytes(c);   // (wide: + the pairs' client state)
  kp.off_seen = (u32)off;
  off += is_mk ? ((size_t)kp.N * MK_SL * mk_slot_words(mk_keys_for(c)) + (size_t)kp.N * mk_keys_for(c) * 3 + 36) * 4   // transactions in flight (the first MK_SL per node), a round's messages per node, the generator's key pool
       : is_dt ? ((size_t)kp.N * (dt_many ? DG_WORDS : DC_WORDS) + 36) * 4   // the nodes' transactions (lock holder, waiting queue, save stack), the generator's key pool
       : is_hat ? 36 * 4   // the generator's key pool
       : is_px ? (size_t)kp.N * PX_SLOTS * 8 + ((c.node_program == MSIM_NODE_LIN_KV_PROXY && c.proxy_service == MSIM_SVC_SEQ_KV) ? 34 : 2) * 256 + 64 * 4   // callbacks per node + service states (seq-kv: + its ring of 32) + seq-kv indices
       : is_txn ? (size_t)kp.N * (txn_many ? TG_SLOTS : TXN_SLOTS) * 16 + 36 * 4   // transactions in flight per node + the generator's key pool
       : is_kf ? ((size_t)kp.N * (txn_many ? KFG_SLOTS : KF_SLOTS) * KSW + (size_t)kp.N * KF_KEYS + (size_t)(txn_many ? kp.CS : kp.N) * KF_KEYS + 36 + 2 * KF_KEYS) * 4   // request handlers, offset caches, client offsets (per worker slot), key pool, lin-kv lengths
       : is_raft ? (size_t)kp.N * 256 + (size_t)kp.N * kp.N * 3 * 4   // KV state + next/match index + append_entries refs
       : wide ? (wide_setl ? (size_t)kp.N * kp.W * 4 : 0) + wide_pending_bytes(c)   // the sets of a wide cluster live in HBM scratch unless they fit LDS (wide_sets_in_lds); + the CRDTs' delivered-but-unmerged replicates
                 : (size_t)kp.N * kp.W * 4;
  off = (off + 15) & ~(size_t)15;
  kp.off_misc = (u32)off; if (c.nemesis_mask) off += (wide ? 128 : 64) * 4;  // shuffle scratch, only the partition nemesis needs it
  const size_t lds = off;
  if (lds > 160 * 1024) { ctx->err = "cluster state exceeds the 160 KiB LDS of a CU (lower inbox_capacity / max_values)"; return MSIM_E_INVALID; }

  // developer check for reads of memory no kernel of this launch wrote: MSIM_POISON=<byte> fills every device buffer of the context
  // with that byte before each launch (HBM comes back from hipMalloc with whatever its last owner left there); results must not depend on it
  static const int poison = []() { const char *p = std::getenv("MSIM_POISON"); return p ? (int)(std::strtoul(p, nullptr, 0) & 0xFFu) : -1; }();
  if (poison >= 0) {
    MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_rows, poison, (size_t)ctx->cap_inst * c.max_rows * sizeof(msim_op), st));
    MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_payload, poison, (size_t)ctx->cap_inst * c.max_payload_words * 4, st));
    MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, poison, (size_t)ctx->cap_inst * sizeof(msim_net_stats), st));
    MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_meta, poison, (size_t)ctx->cap_inst * sizeof(msim_inst_meta), st));
    MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch, poison, (size_t)ctx->cap_inst * ctx->scratch_words_per_inst * 4, st));
    MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_check, poison, (size_t)ctx->cap_inst * sizeof(msim_check_result), st));
    if (ctx->d_journal) MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_journal, poison, (size_t)ctx->cap_inst * c.journal_capacity * sizeof(msim_event), st));
    if (ctx->d_check_scratch) MSIM_HIP_TRY(ctx, hipMemsetAsync(ctx->d_check_scratch, poison, ctx->cap_check_scratch, st));
  }
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));   // (the launch's duration is read from these events on its own stream: msim_last_kernel_ms)
  hipError_t e;
  // the headline layout: two clusters per wavefront (duo.hip); MSIM_DEV_FLAGS bit 9 keeps the one-cluster kernels
  e = MSIM_LAYOUT_DOES_NOT_FIT;
  if (msim_duo_eligible(c) && !(kp.dev_flags & 0x200u) && !((kp.dev_flags & 0x8000u) && msim_bcast8_eligible(c))) {   // (bit 15: small clusters eight per wavefront instead)
    e = msim_launch_duo(kp, n, st);
    if (e == MSIM_LAYOUT_DOES_NOT_FIT && (kp.dev_flags & 0x400u)) { ctx->err = "MSIM_DEV_FLAGS bit 10: the two-clusters-per-wavefront layout was required but this cluster state does not fit it"; return MSIM_E_UNSUPPORTED; }
  }
  // the broadcast programs at the tutorial's cluster sizes: eight clusters per wavefront (bcast8.hip) where the headline layout does not apply
  if (e == MSIM_LAYOUT_DOES_NOT_FIT && msim_bcast8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_bcast8(kp, n, st);
  // Raft: four clusters per wavefront (raft4.hip) when a cluster fits a 16-lane group
  if (msim_raft4_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_raft4(kp, n, st);
  // the lin-kv proxy over lin-kv / lww-kv: four clusters per wavefront (svc4.hip) for large batches when a cluster and its service fit a 16-lane group
  if (msim_svc4_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_svc4(kp, n, st);
  // txn-list-append, single-root node with several workers per node: four clusters per wavefront (txng4.hip) for large batches when nodes + workers + lin-kv fit a 16-lane group
  if (msim_txng4_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_txng4(kp, n, st);
  // the Datomic-style node with several workers per node: four clusters per wavefront (dtg4.hip) for large batches when nodes + workers + the two services fit a 16-lane group
  if (msim_dtg4_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_dtg4(kp, n, st);
  // txn-list-append: eight clusters per wavefront (txn8.hip) when a cluster fits an 8-lane group
  if (msim_txn8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_txn8(kp, n, st);
  // the canonical txn-list-append node: eight clusters per wavefront (mk8.hip) when a cluster fits an 8-lane group
  if (msim_mk8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_mk8(kp, n, st);
  // the Datomic-style txn-list-append node: eight clusters per wavefront (dt8.hip) when a cluster fits an 8-lane group
  if (msim_dt8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_dt8(kp, n, st);
  // txn-rw-register over the highly-available-transactions node: eight clusters per wavefront (hat8.hip)
  if (msim_hat8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_hat8(kp, n, st);
  // kafka: eight clusters per wavefront (kafka8.hip) for large batches
  if (msim_kafka8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_kafka8(kp, n, st);
  // echo / unique-ids (flake ids): eight clusters per wavefront (uid8.hip) for large batches of small clusters
  if (msim_uid8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_uid8(kp, n, st);
  // g-set / pn-counter / g-counter: eight clusters per wavefront (crdt8.hip) for large batches of small clusters
  if (msim_crdt8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_crdt8(kp, n, st);
  if (e == MSIM_LAYOUT_DOES_NOT_FIT && (kp.dev_flags & 0x400u) && is_raft) { ctx->err = "MSIM_DEV_FLAGS bit 10: the four-clusters-per-wavefront Raft layout was required but does not apply"; return MSIM_E_UNSUPPORTED; }
  if (e == MSIM_LAYOUT_DOES_NOT_FIT) switch (c.node_program) {   // not eligible, or the cluster state does not fit the dense layout: one cluster per wavefront (k_*.hip)
    case MSIM_NODE_ECHO: case MSIM_NODE_FLAKE_IDS: e = msim_launch_general_a(kp, n, lds, st); break;
    case MSIM_NODE_G_SET: e = wide ? msim_launch_wide_gset(kp, n, lds, st) : msim_launch_general_a(kp, n, lds, st); break;
    case MSIM_NODE_PN_COUNTER: e = wide ? msim_launch_wide_pn(kp, n, lds, st) : msim_launch_general_a(kp, n, lds, st); break;
    case MSIM_NODE_BCAST_FF: case MSIM_NODE_BCAST_FF_ECHOBACK: e = wide ? msim_launch_wide_bcast(kp, n, lds, st) : msim_launch_general_b(kp, n, lds, st); break;
    case MSIM_NODE_BCAST_ACK_RETRY: case MSIM_NODE_BCAST_RPC_ALL: e = wide ? msim_launch_wide_ack(kp, n, lds, st) : msim_launch_general_c(kp, n, lds, st); break;
    case MSIM_NODE_RAFT: e = msim_launch_raft1(kp, n, lds, st); break;
    case MSIM_NODE_LIN_KV_PROXY: case MSIM_NODE_TSO_IDS: e = msim_launch_svc1(kp, n, lds, st); break;   // (lin-tso ids: the proxy's layout with the timestamp oracle on the service lane)
    case MSIM_NODE_TXN_SINGLE_KEY: e = txn_many ? msim_launch_txng(kp, n, lds, st) : msim_launch_txn1(kp, n, lds, st); break;
    case MSIM_NODE_TXN_MULTI_KEY: e = dt_many ? msim_launch_mkg(kp, n, lds, st) : msim_launch_mk1(kp, n, lds, st); break;
    case MSIM_NODE_TXN_DATOMIC: e = dt_many ? msim_launch_dtg(kp, n, lds, st) : msim_launch_dt1(kp, n, lds, st); break;
    case MSIM_NODE_KAFKA: e = txn_many ? msim_launch_kafkag(kp, n, lds, st) : msim_launch_kafka1(kp, n, lds, st); break;
    case MSIM_NODE_TXN_RW_HAT: e = c.concurrency > c.n_nodes ? msim_launch_hatg(kp, n, lds, st) : msim_launch_hat1(kp, n, lds, st); break;
    default: ctx->err = "node program not built into this engine"; return MSIM_E_UNSUPPORTED;
  }
  if (e != hipSuccess) { ctx->err = std::string("kernel launch: ") + hipGetErrorString(e); return MSIM_E_HIP; }
  ctx->n_inst = n; ctx->first_instance = first;
  ctx->fetched = false; ctx->fetch_pending = false; ctx->checked = false; ctx->check_fetched = false; ctx->ran = true;
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
  ctx->sim_ms_stale = true;   // msim_run_async: the elapsed time is taken when somebody asks for it (the launch is over by then, or is waited for)
  if (blocking) {
    MSIM_HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    MSIM_HIP_TRY(ctx, hipEventElapsedTime(&ctx->sim_ms, ctx->ev0, ctx->ev1));
    ctx->sim_ms_stale = false;
  }
  return MSIM_OK;
}

extern "C" int msim_run(msim_ctx *ctx, uint64_t first_instance, uint32_t n_instances) {
  if (!ctx) return MSIM_E_INVALID;
  return run_impl(ctx, first_instance, n_instances, ctx->stream, true);
}

extern "C" int msim_run_async(msim_ctx *ctx, uint64_t first_instance, uint32_t n_instances, void *hip_stream) {
  if (!ctx) return MSIM_E_INVALID;
  return run_impl(ctx, first_instance, n_instances, hip_stream ? (hipStream_t)hip_stream : ctx->stream, false);
}

extern "C" int msim_check(msim_ctx *ctx) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "msim_check before msim_run"; return MSIM_E_RANGE; }
  if (ctx->cfg.workload == MSIM_WL_LIN_KV) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_lin_kv_host(ctx) : msim_check_lin_kv_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_TXN_LIST_APPEND) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_txn_host(ctx) : msim_check_txn_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_TXN_RW_REGISTER) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_txn_host(ctx) : msim_check_rw_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_KAFKA) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_kafka_host(ctx) : msim_check_kafka_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_PN_COUNTER || ctx->cfg.workload == MSIM_WL_G_COUNTER) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_pn_host(ctx) : msim_check_pn_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_UNIQUE_IDS) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_unique_host(ctx) : msim_check_unique_device(ctx);
  }
  return msim_check_launch(ctx);
}

// Gathers the used prefix of every instance's slab into one contiguous device buffer (16-byte units):
// block (i, j) copies units [j*256*UNROLL ...) of instance i.  off[i] = first unit of instance i in dst.
__global__ void __launch_bounds__(256) compact_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const uint64_t *__restrict__ off,
                                                      uint64_t stride_units) {
  const u32 i = blockIdx.x;
  const uint64_t o = off[i], cnt = off[i + 1] - o;
  const uint4 *s = src + (size_t)i * stride_units;
  for (uint64_t k = (uint64_t)blockIdx.y * 256 + threadIdx.x; k < cnt; k += (uint64_t)gridDim.y * 256) dst[o + k] = s[k];
}
// same for 4-byte units whose per-instance slabs are not 16-byte aligned relative to their compacted position
__global__ void __launch_bounds__(256) compact_words_kernel(const u32 *__restrict__ src, u32 *__restrict__ dst, const uint64_t *__restrict__ off,
                                                            uint64_t stride_words) {
  const u32 i = blockIdx.x;
  const uint64_t o = off[i], cnt = off[i + 1] - o;
  const u32 *s = src + (size_t)i * stride_words;
  for (uint64_t k = (uint64_t)blockIdx.y * 256 + threadIdx.x; k < cnt; k += (uint64_t)gridDim.y * 256) dst[o + k] = s[k];
}

template <typename T>
static int grow_pinned(msim_ctx *ctx, T **buf, size_t *cap, size_t bytes) {
  if (*buf && *cap >= bytes) return MSIM_OK;
  if (*buf) { (void)hipHostFree(*buf); *buf = nullptr; *cap = 0; }
  const size_t want = bytes + bytes / 8 + 4096;  // pinning is slow (pages are faulted in and locked): grow with slack, reuse
  MSIM_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(buf), want));
  *cap = want;
  return MSIM_OK;
}

// one slab kind: compact on the device, then ONE device-to-host copy
// One slab kind to the host: the used prefix of every instance's slab is compacted on the device (phase 0) and crosses PCIe as ONE
// copy (phase 1).  Asynchronous on the context's stream; rows and payload have their own compaction buffer and offset table (pair
// 0 / 1), so that both compactions are queued before the first copy; the caller synchronises.
static int fetch_compacted(msim_ctx *ctx, const void *d_src, uint64_t stride_units, bool units16, const uint64_t *h_off, void *h_dst, int pair, int phase) {
  const uint32_t n = ctx->n_inst;
  const uint64_t total = h_off[n];
  if (!total) return MSIM_OK;
  void **d_compact = pair ? &ctx->d_compact2 : &ctx->d_compact; size_t *cap_compact = pair ? &ctx->cap_compact2 : &ctx->cap_compact;
  uint64_t **d_off = pair ? &ctx->d_off2 : &ctx->d_off; size_t *cap_off = pair ? &ctx->cap_off2 : &ctx->cap_off;
  const size_t unit = units16 ? 16 : 4, bytes = (size_t)total * unit;
  if (phase == 1) { MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_dst, *d_compact, bytes, hipMemcpyDeviceToHost, ctx->stream)); return MSIM_OK; }
  if (*cap_compact < bytes) {
    if (*d_compact) (void)msim_dev_free(*d_compact);
    *d_compact = nullptr; *cap_compact = 0;
    MSIM_HIP_TRY(ctx, msim_dev_malloc(d_compact, bytes + bytes / 8));
    *cap_compact = bytes + bytes / 8;
  }
  if (*cap_off < (size_t)(n + 1) * 8) {
    if (*d_off) (void)msim_dev_free(*d_off);
    *d_off = nullptr; *cap_off = 0;
    MSIM_HIP_TRY(ctx, msim_dev_malloc(reinterpret_cast<void **>(d_off), (size_t)(n + 1) * 8));
    *cap_off = (size_t)(n + 1) * 8;
  }
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(*d_off, h_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  const uint64_t avg = total / n + 1;
  const unsigned gy = (unsigned)((avg + 1023) / 1024 < 1 ? 1 : ((avg + 1023) / 1024 > 64 ? 64 : (avg + 1023) / 1024));
  if (units16) hipLaunchKernelGGL(compact_kernel, dim3(n, gy), dim3(256), 0, ctx->stream, static_cast<const uint4 *>(d_src), static_cast<uint4 *>(*d_compact), *d_off, stride_units);
  else hipLaunchKernelGGL(compact_words_kernel, dim3(n, gy), dim3(256), 0, ctx->stream, static_cast<const u32 *>(d_src), static_cast<u32 *>(*d_compact), *d_off, stride_units);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  return MSIM_OK;
}

// Device-side compaction for the multi-GPU gather (gather.cpp): the same kernels msim_fetch uses, into buffers that stay on
// the device.  Only the instance meta (32 B each) crosses PCIe, to size the buffers and build the offsets.
template <typename T>
static int grow_device(msim_ctx *ctx, T **buf, size_t *cap, size_t bytes) {
  if (*buf && *cap >= bytes) return MSIM_OK;
  if (*buf) { (void)msim_dev_free(*buf); *buf = nullptr; *cap = 0; }
  const size_t want = bytes + bytes / 8 + 256;
  MSIM_HIP_TRY(ctx, msim_dev_malloc(reinterpret_cast<void **>(buf), want));
  *cap = want;
  return MSIM_OK;
}
int msim_compact_on_device(msim_ctx *ctx, uint64_t *row_units, uint64_t *pay_words) {
  if (!ctx->ran) { ctx->err = "gather before msim_run"; return MSIM_E_RANGE; }
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const msim_config &c = ctx->cfg;
  const uint32_t n = ctx->n_inst;
  std::vector<msim_inst_meta> meta(n);
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  MSIM_HIP_TRY(ctx, hipMemcpy(meta.data(), ctx->d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
  std::vector<uint64_t> off(2 * (size_t)(n + 1));
  uint64_t ro = 0, po = 0;
  for (uint32_t i = 0; i < n; i++) { off[i] = ro; off[n + 1 + i] = po; ro += meta[i].n_rows; po += meta[i].n_payload_words; }
  off[n] = ro; off[2 * n + 1] = po;
  int rc;
  if ((rc = grow_device(ctx, &ctx->d_grows, &ctx->cap_grows, (size_t)ro * 16 + 16)) != MSIM_OK) return rc;
  if ((rc = grow_device(ctx, &ctx->d_gpay, &ctx->cap_gpay, (size_t)po * 4 + 16)) != MSIM_OK) return rc;
  if ((rc = grow_device(ctx, &ctx->d_goff, &ctx->cap_goff, off.size() * 8)) != MSIM_OK) return rc;
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_goff, off.data(), off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  const auto gy = [n](uint64_t total) { const uint64_t a = (total / (n ? n : 1) + 1 + 1023) / 1024; return (unsigned)(a < 1 ? 1 : a > 64 ? 64 : a); };
  if (ro) hipLaunchKernelGGL(compact_kernel, dim3(n, gy(ro)), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4 *>(ctx->d_rows),
                             static_cast<uint4 *>(ctx->d_grows), ctx->d_goff, (uint64_t)c.max_rows);
  if (po) hipLaunchKernelGGL(compact_words_kernel, dim3(n, gy(po)), dim3(256), 0, ctx->stream, ctx->d_payload, static_cast<u32 *>(ctx->d_gpay),
                             ctx->d_goff + (n + 1), (uint64_t)c.max_payload_words);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host-side offsets vector goes out of scope
  ctx->g_row_units = ro; ctx->g_pay_words = po;
  if (row_units) *row_units = ro;
  if (pay_words) *pay_words = po;
  return MSIM_OK;
}

// The histories' way to the host in two halves: msim_fetch_begin copies meta / stats (small), compacts rows and payload on the
// device and QUEUES the two big copies; msim_fetch waits for them (and fetches the journal).  A caller that has other work for the
// GPU — the next batch on another context — calls _begin right after the checker and _fetch when it needs the data: the
// compaction kernels are in the queue before the next batch's simulation, the copies run beside it.  msim_fetch alone does both.
extern "C" int msim_fetch_begin(msim_ctx *ctx) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "msim_fetch before msim_run"; return MSIM_E_RANGE; }
  if (ctx->fetched || ctx->fetch_pending) return MSIM_OK;
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const msim_config &c = ctx->cfg;
  const uint32_t n = ctx->n_inst;
  int rc;
  if (ctx->cap_h_meta < n) {  // meta + stats mirrors are sized by instance count
    if (ctx->h_meta) { (void)hipHostFree(ctx->h_meta); ctx->h_meta = nullptr; }
    if (ctx->h_stats) { (void)hipHostFree(ctx->h_stats); ctx->h_stats = nullptr; }
    ctx->cap_h_meta = 0;
    MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_meta, (size_t)n * sizeof(msim_inst_meta)));
    MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_stats, (size_t)n * sizeof(msim_net_stats)));
    ctx->cap_h_meta = n;
  }
  delete[] ctx->h_row_off; delete[] ctx->h_pay_off; delete[] ctx->h_ev_off;
  ctx->h_row_off = new uint64_t[n + 1]; ctx->h_pay_off = new uint64_t[n + 1]; ctx->h_ev_off = new uint64_t[n + 1];
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->h_meta, ctx->d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->h_stats, ctx->d_stats, (size_t)n * sizeof(msim_net_stats), hipMemcpyDeviceToHost));
  uint64_t ro = 0, po = 0, eo = 0;
  for (uint32_t i = 0; i < n; i++) {
    ctx->h_row_off[i] = ro; ctx->h_pay_off[i] = po; ctx->h_ev_off[i] = eo;
    ro += ctx->h_meta[i].n_rows; po += ctx->h_meta[i].n_payload_words;
    eo += ctx->h_meta[i].n_events < c.journal_capacity ? ctx->h_meta[i].n_events : c.journal_capacity;
  }
  ctx->h_row_off[n] = ro; ctx->h_pay_off[n] = po; ctx->h_ev_off[n] = eo;
  if ((rc = grow_pinned(ctx, &ctx->h_rows, &ctx->cap_h_rows, (size_t)(ro + 1) * sizeof(msim_op))) != MSIM_OK) return rc;
  if ((rc = grow_pinned(ctx, &ctx->h_payload, &ctx->cap_h_payload, (size_t)(po + 1) * 4)) != MSIM_OK) return rc;
  if (c.journal_capacity && (rc = grow_pinned(ctx, &ctx->h_journal, &ctx->cap_h_journal, (size_t)(eo + 1) * sizeof(msim_event))) != MSIM_OK) return rc;
  // only the used prefix of every instance's slab crosses PCIe, as one copy per slab kind: both compactions, then both copies
  for (int phase = 0; phase < 2; phase++) {
    if ((rc = fetch_compacted(ctx, ctx->d_rows, c.max_rows, true, ctx->h_row_off, ctx->h_rows, 0, phase)) != MSIM_OK) return rc;
    if ((rc = fetch_compacted(ctx, ctx->d_payload, c.max_payload_words, false, ctx->h_pay_off, ctx->h_payload, 1, phase)) != MSIM_OK) return rc;
    // the compaction kernels are off the CUs before this returns: a simulation launched while they still hold wave slots is
    // placed unevenly over the SIMDs and runs 25 % longer (measured); the copies behind them are DMA and disturb nothing
    if (phase == 0) MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->fetch_pending = true;
  return MSIM_OK;
}

extern "C" int msim_fetch(msim_ctx *ctx) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "msim_fetch before msim_run"; return MSIM_E_RANGE; }
  if (ctx->fetched) return MSIM_OK;
  int rc = msim_fetch_begin(ctx);
  if (rc != MSIM_OK) return rc;
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const msim_config &c = ctx->cfg;
  if (c.journal_capacity) {
    for (int phase = 0; phase < 2; phase++)
      if ((rc = fetch_compacted(ctx, ctx->d_journal, c.journal_capacity, true, ctx->h_ev_off, ctx->h_journal, 0, phase)) != MSIM_OK) return rc;
    MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->fetch_pending = false;
  ctx->fetched = true;
  return MSIM_OK;
}

extern "C" int msim_history(msim_ctx *ctx, uint32_t inst, const msim_op **ops, uint32_t *n_ops, const uint32_t **payload, uint32_t *n_words) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_history: not fetched or instance out of range"; return MSIM_E_RANGE; }
  if (ops) *ops = ctx->h_rows + ctx->h_row_off[inst];
  if (n_ops) *n_ops = ctx->h_meta[inst].n_rows;
  if (payload) *payload = ctx->h_payload + ctx->h_pay_off[inst];
  if (n_words) *n_words = ctx->h_meta[inst].n_payload_words;
  return MSIM_OK;
}

extern "C" int msim_net_stats_get(msim_ctx *ctx, uint32_t inst, msim_net_stats *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_net_stats_get: not fetched or instance out of range"; return MSIM_E_RANGE; }
  *out = ctx->h_stats[inst];
  return MSIM_OK;
}

extern "C" int msim_journal(msim_ctx *ctx, uint32_t inst, const msim_event **events, uint32_t *n_events) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_journal: not fetched or instance out of range"; return MSIM_E_RANGE; }
  if (!ctx->cfg.journal_capacity) { ctx->err = "msim_journal: journal_capacity is 0 (journal off)"; return MSIM_E_INVALID; }
  if (events) *events = ctx->h_journal + ctx->h_ev_off[inst];
  if (n_events) *n_events = (uint32_t)(ctx->h_ev_off[inst + 1] - ctx->h_ev_off[inst]);
  return MSIM_OK;
}

extern "C" int msim_meta(msim_ctx *ctx, uint32_t inst, msim_inst_meta *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_meta: not fetched or instance out of range"; return MSIM_E_RANGE; }
  *out = ctx->h_meta[inst];
  return MSIM_OK;
}

extern "C" int msim_set_dev_flags(msim_ctx *ctx, uint32_t flags) {
  if (!ctx) return MSIM_E_INVALID;
  ctx->dev_flags = flags;
  return MSIM_OK;
}

extern "C" uint32_t msim_check_host_rechecks(const msim_ctx *ctx) { return ctx && ctx->checked && (ctx->cfg.workload == MSIM_WL_LIN_KV || ctx->cfg.workload == MSIM_WL_TXN_LIST_APPEND || ctx->cfg.workload == MSIM_WL_TXN_RW_REGISTER || ctx->cfg.workload == MSIM_WL_PN_COUNTER || ctx->cfg.workload == MSIM_WL_G_COUNTER || ctx->cfg.workload == MSIM_WL_KAFKA) ? ctx->lin_host_rechecks : 0u; }

extern "C" int msim_check_results(msim_ctx *ctx, const msim_check_result **results, uint32_t *n) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->checked) { ctx->err = "msim_check_results before msim_check"; return MSIM_E_RANGE; }
  if (!ctx->check_fetched) {
    MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
    MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)ctx->n_inst * sizeof(msim_check_result)));
    MSIM_HIP_TRY(ctx, hipMemcpy(ctx->h_check, ctx->d_check, (size_t)ctx->n_inst * sizeof(msim_check_result), hipMemcpyDeviceToHost));
    ctx->check_fetched = true;
  }
  if (results) *results = ctx->h_check;
  if (n) *n = ctx->n_inst;
  return MSIM_OK;
}

extern "C" int msim_device_buffers_get(msim_ctx *ctx, msim_device_buffers *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "no run yet"; return MSIM_E_RANGE; }
  const msim_config &c = ctx->cfg;
  out->rows = ctx->d_rows; out->payload = ctx->d_payload; out->stats = ctx->d_stats; out->meta = ctx->d_meta;
  out->check = ctx->d_check; out->check_bytes = (uint64_t)ctx->n_inst * sizeof(msim_check_result);
  out->journal = ctx->d_journal; out->journal_bytes = (uint64_t)ctx->n_inst * c.journal_capacity * sizeof(msim_event);
  out->rows_bytes = (uint64_t)ctx->n_inst * c.max_rows * sizeof(msim_op);
  out->payload_bytes = (uint64_t)ctx->n_inst * c.max_payload_words * 4;
  out->stats_bytes = (uint64_t)ctx->n_inst * sizeof(msim_net_stats);
  out->meta_bytes = (uint64_t)ctx->n_inst * sizeof(msim_inst_meta);
  out->n_instances = ctx->n_inst; out->max_rows = c.max_rows; out->max_payload_words = c.max_payload_words; out->journal_capacity = c.journal_capacity;
  return MSIM_OK;
}

extern "C" int msim_last_kernel_ms(msim_ctx *ctx, float *sim_ms, float *check_ms) {
  if (!ctx) return MSIM_E_INVALID;
  if (ctx->sim_ms_stale) {   // the last launch was asynchronous: its events are read now
    MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    MSIM_HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    MSIM_HIP_TRY(ctx, hipEventElapsedTime(&ctx->sim_ms, ctx->ev0, ctx->ev1));
    ctx->sim_ms_stale = false;
  }
  if (sim_ms) *sim_ms = ctx->sim_ms;
  if (check_ms) *check_ms = ctx->check_ms;
  return MSIM_OK;
}

extern "C" int msim_get_config(const msim_ctx *ctx, msim_config *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  *out = ctx->cfg;
  return MSIM_OK;
}

extern "C" const char *msim_last_error(const msim_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" void msim_destroy(msim_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  free_buffers(ctx);
  msim_gather_free(ctx);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
  if (ctx->ev3) (void)hipEventDestroy(ctx->ev3);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// Validates the DPP encodings of wave_min / wave_incl_scan against shuffle-based references on the
// device.  Returns 0 when they agree, >0 = number of mismatching lanes, <0 = MSIM_E_*.
extern "C" int msim_selftest_wave(int device) {
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_NO_DEVICE;
  const int blocks = 64, n = blocks * 64;
  u32 *h = new u32[n], *d_in = nullptr, *d_out = nullptr;
  u64 x = 12345;
  for (int i = 0; i < n; i++) { x = mix64(x + i); h[i] = (u32)x; if (i % 7 == 0) h[i] = 0xFFFFFFFFu; }
  int bad = 0;
  if (msim_dev_malloc(&d_in, n * 4) != hipSuccess || msim_dev_malloc(&d_out, n * 4) != hipSuccess) { delete[] h; return MSIM_E_HIP; }
  (void)hipMemcpy(d_in, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(wave_selftest_kernel, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
  if (hipMemcpy(h, d_out, n * 4, hipMemcpyDeviceToHost) != hipSuccess) bad = MSIM_E_HIP;
  else for (int i = 0; i < n; i++) bad += h[i] != 0;
  (void)msim_dev_free(d_in); (void)msim_dev_free(d_out);
  delete[] h;
  return bad;
}
