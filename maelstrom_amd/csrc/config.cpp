// config.cpp — host-side option handling of libmaelsim (no device code).
//
// Mirrors the reference's CLI option map and its defaults: src/maelstrom/core.clj:136-229 (opt-spec),
// core.clj:231-265 (parse-node-count / opt-fn), plus [upstream] jepsen.cli defaults (--time-limit 60,
// --concurrency 1n).  Capacities (history rows, payload words, set width, inbox depth) have no
// counterpart in the reference (JVM heap); they are derived here from rate x time-limit so that an
// instance overflows only ~10 sigma away from its expected size, and every overflow is reported
// (msim_inst_meta.flags), never silently truncated.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/maelsim.h"
#include "engine_limits.h"

static void set_err(char *err, size_t n, const char *msg) {
  if (err && n) { std::snprintf(err, n, "%s", msg); }
}

extern "C" uint32_t msim_abi_version(void) { return MSIM_ABI_VERSION; }

extern "C" int msim_config_defaults(msim_config *cfg, uint32_t workload, uint32_t n_nodes) {
  if (!cfg) return MSIM_E_INVALID;
  std::memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = sizeof *cfg;
  cfg->abi_version = MSIM_ABI_VERSION;
  cfg->workload = workload;
  switch (workload) {
    case MSIM_WL_ECHO: cfg->node_program = MSIM_NODE_ECHO; break;
    case MSIM_WL_BROADCAST: cfg->node_program = MSIM_NODE_BCAST_FF; break;
    case MSIM_WL_G_SET: cfg->node_program = MSIM_NODE_G_SET; break;
    case MSIM_WL_TXN_LIST_APPEND: cfg->node_program = MSIM_NODE_TXN_SINGLE_KEY; break;
    case MSIM_WL_PN_COUNTER: case MSIM_WL_G_COUNTER: cfg->node_program = MSIM_NODE_PN_COUNTER; break;
    case MSIM_WL_UNIQUE_IDS: cfg->node_program = MSIM_NODE_FLAKE_IDS; break;
    case MSIM_WL_TXN_RW_REGISTER: cfg->node_program = MSIM_NODE_TXN_RW_HAT; cfg->consistency_model = MSIM_CM_READ_COMMITTED; break;  // core.clj:115-121
    case MSIM_WL_KAFKA: cfg->node_program = MSIM_NODE_KAFKA; break;
    default: cfg->node_program = MSIM_NODE_RAFT; break;
  }
  cfg->n_nodes = n_nodes;
  cfg->concurrency = n_nodes;           // "1n"
  if (workload == MSIM_WL_LIN_KV) cfg->concurrency = 2 * n_nodes;  // linearizable-register needs 2n threads per key (core.clj:111)
  cfg->rate_mhz = 5000;                 // core.clj:219-222
  cfg->time_limit_ms = 60000;           // jepsen.cli
  cfg->latency_mean_ms = 0;             // core.clj:171-174
  cfg->latency_dist = MSIM_LAT_CONSTANT;
  cfg->p_loss_q32 = 0;                  // net.clj:100
  cfg->topology = MSIM_TOPO_GRID;       // core.clj:224-227
  cfg->nemesis_mask = 0;
  cfg->nemesis_interval_ms = 10000;     // core.clj:214-217
  cfg->client_timeout_ms = 5000;        // client.clj:18-20
  cfg->quiesce_ms = 10000;              // core.clj:78
  cfg->seed = 0;
  cfg->key_count = 0;                   // core.clj:167-169: no default here; [upstream] elle picks 10 for the exponential key choice
  cfg->max_txn_length = 4;              // core.clj:191-194
  cfg->max_writes_per_key = 16;         // core.clj:196-199
  if (workload == MSIM_WL_KAFKA) { cfg->key_count = 4; cfg->max_writes_per_key = 1024; }   // [upstream] jepsen.tests.kafka: a few long-lived keys
  return MSIM_OK;
}

static uint32_t max_degree(const msim_config *c) {
  uint32_t n = c->n_nodes;
  if (c->node_program == MSIM_NODE_BCAST_RPC_ALL || c->node_program == MSIM_NODE_G_SET || c->node_program == MSIM_NODE_PN_COUNTER) return n ? n - 1 : 0;
  switch (c->topology) {
    case MSIM_TOPO_GRID: return n > 4 ? 4 : (n ? n - 1 : 0);
    case MSIM_TOPO_LINE: return n > 2 ? 2 : (n ? n - 1 : 0);
    case MSIM_TOPO_TOTAL: return n ? n - 1 : 0;
    case MSIM_TOPO_TREE2: return 3;
    case MSIM_TOPO_TREE3: return 4;
    default: return 5;
  }
}

extern "C" int msim_config_finalize(msim_config *c, char *err, size_t errlen) {
  if (!c) return MSIM_E_INVALID;
  if (c->struct_size != sizeof(msim_config) || c->abi_version != MSIM_ABI_VERSION) {
    set_err(err, errlen, "msim_config: struct_size/abi_version mismatch"); return MSIM_E_INVALID; }
  if (c->n_nodes == 0 || c->n_nodes > MSIM_MAX_NODES) { set_err(err, errlen, "n_nodes must be in 1..128"); return MSIM_E_INVALID; }
  if (c->concurrency == 0) c->concurrency = c->n_nodes;
  uint32_t slots = c->concurrency > c->n_nodes ? c->concurrency : c->n_nodes;
  if (c->n_nodes + slots > 255) { set_err(err, errlen, "n_nodes + max(concurrency, n_nodes) must be <= 255"); return MSIM_E_INVALID; }
  if (c->workload > MSIM_WL_KAFKA) { set_err(err, errlen, "unknown workload"); return MSIM_E_INVALID; }
  if (c->latency_dist > MSIM_LAT_EXPONENTIAL) { set_err(err, errlen, "latency_dist must be constant, uniform, or exponential"); return MSIM_E_INVALID; }
  if (c->latency_dist == MSIM_LAT_EXPONENTIAL && c->latency_mean_ms == 0) {
    // net.clj:77 (exponential-distribution (/ mean)) throws "Divide by zero" for --latency 0
    set_err(err, errlen, "exponential latency needs a non-zero mean (net.clj:77 divides by zero)"); return MSIM_E_INVALID; }
  if (c->topology > MSIM_TOPO_TREE4) { set_err(err, errlen, "unknown topology"); return MSIM_E_INVALID; }
  if (c->nemesis_mask & ~MSIM_NEMESIS_PARTITION) { set_err(err, errlen, "unknown nemesis fault (only partition, core.clj:49-51)"); return MSIM_E_INVALID; }
  if (c->nemesis_interval_ms == 0) { set_err(err, errlen, "nemesis interval must be positive"); return MSIM_E_INVALID; }
  if (c->latency_mean_ms > 60000) { set_err(err, errlen, "latency mean above 60 s is not supported"); return MSIM_E_INVALID; }
  // virtual time is u32 microseconds and the kernels order events by 2 x time (+ 1 for client timeouts): 2^31 us ~ 35 min
  if ((uint64_t)c->time_limit_ms + c->quiesce_ms + 2ull * c->client_timeout_ms + 16ull * c->latency_mean_ms > 2000000ull) {
    set_err(err, errlen, "time-limit + quiesce + timeouts must fit in 2000 s of virtual time (2 x time in u32 microseconds)"); return MSIM_E_INVALID; }
  // workload <-> node program compatibility
  bool ok = false;
  switch (c->workload) {
    case MSIM_WL_ECHO: ok = c->node_program == MSIM_NODE_ECHO; break;
    case MSIM_WL_BROADCAST: ok = c->node_program >= MSIM_NODE_BCAST_FF && c->node_program <= MSIM_NODE_BCAST_RPC_ALL; break;
    case MSIM_WL_G_SET: ok = c->node_program == MSIM_NODE_G_SET; break;
    case MSIM_WL_LIN_KV: ok = c->node_program == MSIM_NODE_RAFT || c->node_program == MSIM_NODE_LIN_KV_PROXY; break;
    case MSIM_WL_TXN_LIST_APPEND: ok = c->node_program == MSIM_NODE_TXN_SINGLE_KEY || c->node_program == MSIM_NODE_TXN_MULTI_KEY || c->node_program == MSIM_NODE_TXN_DATOMIC; break;
    case MSIM_WL_PN_COUNTER: case MSIM_WL_G_COUNTER: ok = c->node_program == MSIM_NODE_PN_COUNTER; break;
    case MSIM_WL_UNIQUE_IDS: ok = c->node_program == MSIM_NODE_FLAKE_IDS || c->node_program == MSIM_NODE_TSO_IDS; break;
    case MSIM_WL_TXN_RW_REGISTER: ok = c->node_program == MSIM_NODE_TXN_RW_HAT; break;
    case MSIM_WL_KAFKA: ok = c->node_program == MSIM_NODE_KAFKA; break;
    default: break;
  }
  if (!ok) { set_err(err, errlen, "node_program does not implement this workload"); return MSIM_E_INVALID; }

  if (c->proxy_service > MSIM_SVC_LWW_KV) { set_err(err, errlen, "proxy_service must be lin-kv, seq-kv or lww-kv"); return MSIM_E_INVALID; }
  if (c->consistency_model > MSIM_CM_READ_UNCOMMITTED) { set_err(err, errlen, "unknown consistency model"); return MSIM_E_INVALID; }
  const bool hat = c->workload == MSIM_WL_TXN_RW_REGISTER;
  const bool txn = c->workload == MSIM_WL_TXN_LIST_APPEND || hat;
  if (txn) {
    if (c->key_count == 0) c->key_count = 10;
    if (c->max_txn_length == 0) c->max_txn_length = 4;
    if (c->max_writes_per_key == 0) c->max_writes_per_key = 16;
    if (c->key_count > 16 || c->max_txn_length > 8 || c->max_writes_per_key > 63) {
      set_err(err, errlen, "transactional workloads: key-count <= 16, max-txn-length <= 8, max-writes-per-key <= 63"); return MSIM_E_INVALID; }
    // Several workers per node (`--concurrency 10n`, doc/05-datomic/01-single-node.md:257,322: what puts transactions behind each other at a
    // node's lock) for the nodes whose kernels have the general lane layout: the Datomic-style node (dtg_kernel<>), the single-root node (txng_kernel<>), the multi-key node (mkg_kernel<>) and txn-rw-register's node (hatg_kernel<>).  Worker t talks to node
    // t mod n ([upstream] interpreter), so the count must be a multiple of the node count here.
    const bool many_workers_ok = (c->node_program == MSIM_NODE_TXN_DATOMIC || c->node_program == MSIM_NODE_TXN_SINGLE_KEY || c->node_program == MSIM_NODE_TXN_MULTI_KEY || hat) &&
                                 c->concurrency > c->n_nodes && c->concurrency % c->n_nodes == 0 && c->n_nodes + c->concurrency + (hat ? 0u : 2u) <= 64;   // (a lane per endpoint: nodes, workers, up to two services)
    if ((c->concurrency != c->n_nodes && !many_workers_ok) || c->n_nodes > 31) {
      set_err(err, errlen, "transactional workloads: one worker per node (concurrency == node-count <= 31) in this build; the three txn-list-append nodes and the txn-rw-register node also take k x node-count workers (nodes + workers + 2 <= 64)"); return MSIM_E_UNSUPPORTED; }
    // txn_rw_register_hat.clj:85-90: with no other node the pending set of a txn is empty and replicate-step! sends to nil
    if (hat && (c->n_nodes < 2 || c->n_nodes > 8)) { set_err(err, errlen, "txn-rw-register: 2..8 nodes in this build"); return MSIM_E_UNSUPPORTED; }
  }
  const bool kafka = c->workload == MSIM_WL_KAFKA;
  if (kafka) {
    if (c->key_count == 0) c->key_count = 4;
    if (c->max_writes_per_key == 0) c->max_writes_per_key = 1024;
    // the {key offset} maps of the protocol keep insertion order up to 8 entries (Clojure array-maps); message values and offsets
    // travel in 11 bits of a history row
    if (c->key_count > 8 || c->max_writes_per_key > 2046) { set_err(err, errlen, "kafka: key-count <= 8, max-writes-per-key <= 2046"); return MSIM_E_INVALID; }
    const bool kf_many = c->concurrency > c->n_nodes && c->concurrency % c->n_nodes == 0 && c->n_nodes + c->concurrency + 1 <= 64;   // kafkag_kernel<>: a lane per endpoint
    if ((c->concurrency != c->n_nodes && !kf_many) || c->n_nodes > 30) { set_err(err, errlen, "kafka: one worker per node (concurrency == node-count <= 30), or k x node-count workers with nodes + workers + 1 <= 64"); return MSIM_E_UNSUPPORTED; }
  }
  double expected = (double)c->rate_mhz * (double)c->time_limit_ms / 1e6;
  uint32_t ops_max = (uint32_t)(expected + expected / 8.0) + 64;
  uint32_t adds = (uint32_t)(ops_max / 2 + 4.0 * std::sqrt((double)ops_max)) + 32;
  uint32_t nem_ops = 0;
  if (c->nemesis_mask) nem_ops = 4 * (c->time_limit_ms / c->nemesis_interval_ms + 1) + 16;
  if (c->workload == MSIM_WL_LIN_KV && c->concurrency % (2 * c->n_nodes)) {
    set_err(err, errlen, "lin-kv: concurrency must be a multiple of 2 x node-count ([upstream] independent/concurrent-generator)"); return MSIM_E_INVALID; }
  const bool pn = c->workload == MSIM_WL_PN_COUNTER || c->workload == MSIM_WL_G_COUNTER;
  if (pn && c->n_nodes > 127) { set_err(err, errlen, "pn-counter: at most 127 nodes in this build"); return MSIM_E_UNSUPPORTED; }
  // pn-counter: a node's state is 2 x n_nodes counters (one G-counter for increments, one for decrements): max_values / 32 words
  if (pn) c->max_values = 64 * c->n_nodes;
  if (kafka) c->max_values = 32;   // (keys: at most 8)
  const bool no_sets = kafka || c->workload == MSIM_WL_UNIQUE_IDS || c->workload == MSIM_WL_ECHO || c->workload == MSIM_WL_LIN_KV || txn || pn;
  // txn-list-append: max_values = distinct keys ever used (a key is retired after max_writes_per_key appends)
  if (txn && c->max_values == 0) c->max_values = c->key_count + (ops_max * c->max_txn_length) / c->max_writes_per_key + 32;
  if (txn && c->max_values > 32767) { set_err(err, errlen, "txn-list-append: more than 32767 keys"); return MSIM_E_INVALID; }
  if (c->max_values == 0) c->max_values = no_sets ? 32 : ((adds + 31) / 32) * 32;
  if (c->max_values % 32) c->max_values = ((c->max_values + 31) / 32) * 32;
  // kafka: the final generator adds an :assign and up to (messages per key / 32 + 2) polls per worker
  const uint32_t kf_final_ops = kafka ? c->concurrency * (3 + (ops_max / 2 + 31) / 32 + c->max_writes_per_key / 32) : 0;
  if (c->max_rows == 0) c->max_rows = 2 * (ops_max + c->concurrency + kf_final_ops) + 2 * nem_ops + 16;
  if (c->max_payload_words == 0) {
    uint32_t w = c->max_values / 32;
    uint64_t words = no_sets ? 16 : (uint64_t)(adds + c->concurrency) * w;
    // a transaction: <= L header words at :invoke, <= L x (1 + ceil((writes-per-key + L) / 4)) at completion
    if (txn) words = (uint64_t)(ops_max + c->concurrency) * c->max_txn_length * (2 + (c->max_writes_per_key + c->max_txn_length + 3) / 4) + 64;  // worst case: all reads of full lists
    if (hat) words = (uint64_t)(ops_max + c->concurrency) * c->max_txn_length * 2 + 64;  // one word per micro-op at :invoke and at completion
    // kafka: a poll asks with <= 8 words and brings <= 8 x (1 + 16) back, the list_committed_offsets reply 8; every message is polled by
    // every worker in the final phase (two per word + headers)
    if (kafka) words = (uint64_t)(ops_max + c->concurrency) * 80 + (uint64_t)c->concurrency * (ops_max / 2 + 8ull * c->max_writes_per_key / 32 * 2 + 64) + (uint64_t)kf_final_ops * 24 + 64;
    words += (uint64_t)nem_ops * c->n_nodes * MSIM_MASK_WORDS + 16;
    if (words > 0xFFFFFFu) { set_err(err, errlen, "payload area above 2^24 words per instance"); return MSIM_E_INVALID; }
    c->max_payload_words = (uint32_t)words;
  }
  if (c->max_payload_words > 0xFFFFFFu) { set_err(err, errlen, "max_payload_words must be < 2^24"); return MSIM_E_INVALID; }
  if (c->max_values > 255u * 32u) { set_err(err, errlen, "max_values above 8160 (read length is 8 bits of words)"); return MSIM_E_INVALID; }
  {
    // Queue depth per node = inflow x how long recv! can sleep on one envelope (head-of-line blocking,
    // net.clj:236-238).  The LDS part stays small; the rest lives in an HBM spill area.
    uint32_t deg = max_degree(c);
    double per_s = (double)c->rate_mhz / 2000.0 * deg;             // server msgs per second into one node
    if (c->node_program == MSIM_NODE_BCAST_ACK_RETRY || c->node_program == MSIM_NODE_BCAST_RPC_ALL) per_s *= 2;  // + acks
    double lat_s = c->latency_mean_ms / 1000.0;
    if (c->latency_dist == MSIM_LAT_UNIFORM) lat_s *= 2.0;         // max of uniform [0, 2 mean)
    if (c->latency_dist == MSIM_LAT_EXPONENTIAL) lat_s *= 16.0;    // P(latency > 16 mean) ~ 1e-7 per message
    uint32_t depth = 32 + 2 * deg + (uint32_t)(per_s * lat_s * 6.0);  // x6: fan-in bursts (every neighbour forwards at once)
    // retrying gossip under partitions: at heal time every neighbour re-sends everything it could not deliver
    if (c->node_program == MSIM_NODE_BCAST_ACK_RETRY && c->nemesis_mask) depth += (deg < 4 ? deg : 4) * adds;
    // one replicate_full per peer per 5 s tick (g_set.rb:33-38): two ticks' worth, + one per 5 s a message can be under way (1 s exponential: 25 of 4096
    // instances of cfg3's shape overflowed the two-tick queues, profiles/r05_cfg3_latency_sweep.jsonl)
    if (c->node_program == MSIM_NODE_G_SET || c->node_program == MSIM_NODE_PN_COUNTER) depth = 16 + (2 + (uint32_t)(lat_s / 5.0)) * deg;
    if (c->node_program == MSIM_NODE_RAFT) depth = 24 + 1024;   // heartbeats / re-sent append_entries pile up behind a sleeping recv!
    if (txn || kafka) depth = 16 + 4 * c->n_nodes + (c->concurrency > c->n_nodes ? 2 * c->concurrency : 0);   // the service sees <= 2 requests per transaction in flight (+ every worker's at once)
    if (c->node_program == MSIM_NODE_TXN_MULTI_KEY) depth = 16 + 16 * c->n_nodes + (c->concurrency > c->n_nodes ? 8 * c->concurrency : 0);   // lww-kv: up to max-txn-length thunk reads / writes per transaction (several workers per node: per worker)
    if (c->node_program == MSIM_NODE_TXN_DATOMIC) depth = 16 + (c->n_nodes < 4 ? 304 : 96 * c->n_nodes) + (c->concurrency > c->n_nodes ? c->concurrency : 0);   // (+ every worker's request at its node at once); lww-kv: every node may have the new tree nodes of a transaction in flight (a path per append; one transaction writes at most DT_MAXW = 256)
    if (hat) depth = 16 + 4 * c->n_nodes + (uint32_t)(20.0 * c->n_nodes * lat_s) + (c->concurrency > c->n_nodes ? c->concurrency : 0);  // a replicate + n-1 acks per peer per 100 ms tick
    if (c->node_program == MSIM_NODE_LIN_KV_PROXY || c->node_program == MSIM_NODE_TSO_IDS) depth = 16 + 2 * c->concurrency;   // the service sees every worker's request at once
    // wide clusters: 100+ queues would take a fifth of the LDS budget of a cluster; their queues live in the HBM spill area
    // only (kept sorted, the head cached in registers: sim_kernel_wide.inc), which buys a sixth wavefront per CU
    const uint32_t lds_part = c->n_nodes > 32 ? 0 : 24;
    if (c->inbox_capacity == 0) c->inbox_capacity = depth < lds_part ? depth : lds_part;
    if (c->spill_capacity == 0) c->spill_capacity = depth > c->inbox_capacity ? depth - c->inbox_capacity : 0;
    if (c->spill_capacity > 65536) { set_err(err, errlen, "spill_capacity above 65536 envelopes per node"); return MSIM_E_INVALID; }
  }
  if (hat && c->replication_words == 0) {
    // Every 100 ms a node with unacknowledged txns sends ALL of them again (txn_rw_register_hat.clj:92-118): one word per txn
    // and message.  Healthy network: a txn is listed once per tick until its round trip completes, and with more than two
    // nodes every receiver relays what is still pending elsewhere (:143-150) — measured ~n^2 / 2.5 words per txn.  Behind a
    // partition the pending set grows for the whole partition (d <= 2 x interval): 10 ticks/s x (rate/n x t) txns, integrated
    // = 5 x rate/n x d^2 per partition.
    const double n = c->n_nodes, tl = c->time_limit_ms / 1000.0;
    const double ops = (double)c->rate_mhz / 1000.0 * tl + 64.0, r_n = (double)c->rate_mhz / 1000.0 / n;
    const double lat_s = c->latency_mean_ms / 1000.0 * (c->latency_dist == MSIM_LAT_EXPONENTIAL ? 8.0 : 2.0);
    double w = ops * (n * n / 2.0) * (1.0 + 2.0 * lat_s / 0.1) * (c->p_loss_q32 ? 2.0 : 1.0) * 1.5 + 1024.0;
    // (with relays every node of a component holds everybody's txns for the nodes outside: the full rate, not rate/n).
    // A stop followed by a start less than a tick later never lets the set drain, so the bound is ONE partition as long as
    // the test: 5 x rate x time-limit^2 words per node (seen in 4 of 16384 instances of the reference's demo shape).
    if (c->nemesis_mask) w += 1.05 * n * 5.0 * (c->n_nodes > 2 ? r_n * n : r_n) * tl * tl;
    if (w > 16.0 * 1024 * 1024) { set_err(err, errlen, "txn-rw-register: replicate lists above 2^24 words per instance (lower rate / time-limit / latency / nemesis interval)"); return MSIM_E_INVALID; }
    c->replication_words = ((uint32_t)w + 3u) & ~3u;
  }
  return MSIM_OK;
}
