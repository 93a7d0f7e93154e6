/* maelsim.h — C ABI of libmaelsim, the MI355X-native ensemble cluster-simulation engine.
 *
 * This is the drop-in boundary for Maelstrom's hot path.  One call replaces, for a
 * whole ensemble of independent seeded test instances, what the reference does per test with
 *   - maelstrom.net       /root/reference/src/maelstrom/net.clj:79-247   (queues, latency, loss, partitions)
 *   - maelstrom.process   /root/reference/src/maelstrom/process.clj:136-215 (node execution: recv! -> node -> send!)
 *   - maelstrom.nemesis   /root/reference/src/maelstrom/nemesis.clj:10-16 (partition schedule)
 *   - maelstrom.client    /root/reference/src/maelstrom/client.clj:41-172 (sync RPC, timeouts, op completion)
 *   - maelstrom.db        /root/reference/src/maelstrom/db.clj:46-69      (init handshake)
 *   - the demo node programs' state-transition functions (echo, broadcast, g-set)
 * and returns what `jepsen.core/run!` would have handed to `(checker/check (:checker test) test history opts)`
 * (core.clj:91-100): the op history, plus the numbers `maelstrom.net.checker` computes from the journal
 * (net/checker.clj:28-41).
 *
 * Plain C, plain pointers and sizes.  No C++ exception crosses this boundary: every entry point returns
 * an MSIM_E_* code (0 = OK) and msim_last_error() gives the text.  The engine requires a HIP device and
 * fails loudly (MSIM_E_NO_DEVICE) without one; there is no CPU fallback behind this ABI.
 */
#ifndef MAELSIM_H
#define MAELSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSIM_ABI_VERSION 1u

/* ---- error codes -------------------------------------------------------------------------------- */
enum {
  MSIM_OK = 0,
  MSIM_E_INVALID = -1,      /* bad config / argument (text in msim_last_error)                        */
  MSIM_E_NO_DEVICE = -2,    /* no HIP device visible; the engine has no CPU path                       */
  MSIM_E_HIP = -3,          /* a HIP runtime call failed                                               */
  MSIM_E_NOMEM = -4,
  MSIM_E_RANGE = -5,        /* instance index out of range / nothing run yet                           */
  MSIM_E_UNSUPPORTED = -6,  /* workload/protocol combination not built into this engine                */
  MSIM_E_OVERFLOW = -7      /* >=1 instance overflowed a capacity (see msim_inst_meta.flags)           */
};

/* ---- configuration (mirrors the CLI option map, core.clj:136-229, + ensemble/determinism fields) -- */
enum { MSIM_WL_ECHO = 0, MSIM_WL_BROADCAST = 1, MSIM_WL_G_SET = 2, MSIM_WL_LIN_KV = 3, MSIM_WL_TXN_LIST_APPEND = 4,
       MSIM_WL_PN_COUNTER = 5 /* workload/pn_counter.clj */, MSIM_WL_G_COUNTER = 6 /* workload/g_counter.clj: pn-counter without negative adds */,
       MSIM_WL_UNIQUE_IDS = 7 /* workload/unique_ids.clj */,
       MSIM_WL_TXN_RW_REGISTER = 8 /* workload/txn_rw_register.clj: transactions of reads / writes over registers */,
       MSIM_WL_KAFKA = 9 /* workload/kafka.clj: append-only logs per key: send / poll / assign / crash, committed offsets */ };

/* Built-in node programs (the `--bin` of the reference). */
enum {
  MSIM_NODE_ECHO = 0,           /* demo/ruby/echo.rb:20-41, demo/python/echo.py:9-10                         */
  MSIM_NODE_BCAST_FF = 1,       /* doc/03-broadcast/01-broadcast.md:525-547 + 02-performance.md:61-67:
                                   fire-and-forget gossip, dedup, skip-sender                                 */
  MSIM_NODE_BCAST_FF_ECHOBACK = 2, /* same without skip-sender (02-performance.md:22-28, KAT-3)               */
  MSIM_NODE_BCAST_ACK_RETRY = 3,/* doc/03-broadcast/02-performance.md:406-441, demo/js/gossip.js:24-38:
                                   ack everyone, resend un-acked every 1 s                                    */
  MSIM_NODE_BCAST_RPC_ALL = 4,  /* demo/ruby/broadcast.rb:29-47: RPC to every other node, no retry           */
  MSIM_NODE_G_SET = 5,          /* demo/ruby/g_set.rb:8-39: replicate_full to all others every 5 s          */
  MSIM_NODE_RAFT = 6,           /* demo/ruby/raft.rb:1-497 == demo/python/raft.py:1-593 (lin-kv)             */
  MSIM_NODE_TXN_SINGLE_KEY = 7, /* demo/clojure/single_key_txn.clj:116-180: whole database under one lin-kv key:
                                   read root -> apply txn -> cas root (create_if_not_exists), conflict => error 30.
                                   Brings the `lin-kv` service endpoint with it (service.clj:31-61,141-155,290-296) */
  MSIM_NODE_LIN_KV_PROXY = 10,  /* demo/ruby/lin_kv_proxy.rb:8-43: every read / write / cas is proxied to the key-value service
                                   named by msim_config.proxy_service (service.clj:31-114,141-243,290-296)                   */
  MSIM_NODE_FLAKE_IDS = 9,      /* demo/clojure/flake_ids.clj:10-33: id = [seconds, counter within that second, node id] */
  MSIM_NODE_PN_COUNTER = 8,     /* demo/ruby/pn_counter.rb:8-121 == demo/js/crdt_pn_counter.js: increments and decrements in two
                                   per-node G-counters, merged by element-wise max, replicated to all others every 5 s      */
  MSIM_NODE_TXN_RW_HAT = 11,    /* demo/clojure/txn_rw_register_hat.clj:1-190: highly available transactions — every node applies a
                                   txn locally at a Lamport timestamp (last write wins per key), then replicates it to the
                                   others every 100 ms until they acknowledge (the demo of core.clj:115-121)                */
  MSIM_NODE_TXN_MULTI_KEY = 12, /* demo/js/multi_key_txn.js:1-246 == demo/clojure/multi_key_txn.clj (same architecture as the workload's demo at
                                   core.clj:113-114, demo/ruby/datomic_list_append.rb, but a different program — that one keeps a persistent
                                   hash tree of lazily loaded nodes: MSIM_NODE_TXN_DATOMIC): thunks in lww-kv, the root
                                   map in lin-kv, retry when the root cas is lost (oracle/mk_nodes.inc, pinned by the real program on
                                   the process bridge; csrc/sim_kernel_mk.inc).  One worker per node, at most 30 nodes               */
  MSIM_NODE_KAFKA = 14,         /* demo/clojure/kafka.clj:1-172: logs in 32-message chunks under lin-kv keys (read + cas per send), committed
                                   offsets under one lin-kv key; brings the `lin-kv` service endpoint with it.  One worker per node, at most
                                   8 keys per test (oracle/kafka_nodes.inc, csrc/sim_kernel_kafka.inc; parity unpinned: babashka only)      */
  MSIM_NODE_TXN_DATOMIC = 15,   /* demo/ruby/datomic_list_append.rb:47-424, the node core.clj:113-114 runs for this workload: the database is a
                                   persistent hash tree (128 hash values, branch factor 8, CRC32 of the key) of immutable nodes in lww-kv under
                                   fresh pointers, the root pointer in lin-kv; a transaction takes the node's lock, reads the root pointer, loads the
                                   tree nodes on its keys' paths lazily (a cache of what the node has loaded), copies the paths it appends to, writes
                                   the new nodes children first and cas-es the root; a lost cas answers error 30 (oracle/dt_nodes.inc — parity
                                   unpinned: there is no Ruby here; csrc/sim_kernel_dt.inc, csrc/dt8.hip).  One worker per node, at most 30 nodes; every
                                   blocking step of a transaction gives up after 5 s (promise.rb:5): error 0 => :info with MSIM_ERR_TIMEOUT   */
  MSIM_NODE_TSO_IDS = 13        /* unique-ids over the `lin-tso` timestamp oracle (service.clj:116-132,290-296; doc/services.md): every
                                   `generate` becomes a {type "ts"} RPC to lin-tso, the timestamp is the id.  The reference ships the
                                   service but no demo that uses it; this node (tools/harness_tso_node.py is its process form) is what
                                   exercises it                                                                                  */
};

enum { MSIM_LAT_CONSTANT = 0, MSIM_LAT_UNIFORM = 1, MSIM_LAT_EXPONENTIAL = 2 };  /* net.clj:65-77 */
enum { MSIM_TOPO_GRID = 0, MSIM_TOPO_LINE = 1, MSIM_TOPO_TOTAL = 2,
       MSIM_TOPO_TREE2 = 3, MSIM_TOPO_TREE3 = 4, MSIM_TOPO_TREE4 = 5 };           /* broadcast.clj:171-179 */
enum { MSIM_SVC_LIN_KV = 0, MSIM_SVC_SEQ_KV = 1, MSIM_SVC_LWW_KV = 2, MSIM_SVC_LIN_TSO = 3 };   /* service.clj:290-296 */
/* --consistency-models (core.clj:160-165, default strict-serializable; the txn-rw-register demo asks for read-committed,
 * core.clj:118): which of the anomalies the transactional checkers find make a history invalid (see msim_check_txn_rows). */
enum { MSIM_CM_STRICT_SERIALIZABLE = 0, MSIM_CM_SERIALIZABLE = 1, MSIM_CM_SNAPSHOT_ISOLATION = 2, MSIM_CM_READ_COMMITTED = 3,
       MSIM_CM_READ_UNCOMMITTED = 4 };
enum { MSIM_NEMESIS_PARTITION = 1u };                                              /* core.clj:49-51 */

typedef struct msim_config {
  uint32_t struct_size;          /* = sizeof(msim_config); versioning guard                                */
  uint32_t abi_version;          /* = MSIM_ABI_VERSION                                                     */
  uint32_t workload;             /* MSIM_WL_*        (-w, core.clj:141-144)                                */
  uint32_t node_program;         /* MSIM_NODE_*      (stands in for --bin)                                 */
  uint32_t n_nodes;              /* --node-count     (core.clj:201-204)                                    */
  uint32_t concurrency;          /* --concurrency, default 1n = n_nodes [upstream jepsen.cli]              */
  uint32_t rate_mhz;             /* --rate in milli-ops/s (5/s -> 5000; core.clj:219-222); 0 = no client ops */
  uint32_t time_limit_ms;        /* --time-limit [upstream], default 60 s                                  */
  uint32_t latency_mean_ms;      /* --latency        (core.clj:171-174)                                    */
  uint32_t latency_dist;         /* MSIM_LAT_*       (core.clj:176-180)                                    */
  uint32_t p_loss_q32;           /* P(loss) * 2^32, clamped to 2^32-1; net.clj:100,122,214. 0 = reference default */
  uint32_t topology;             /* MSIM_TOPO_*      (core.clj:224-227)                                    */
  uint32_t nemesis_mask;         /* MSIM_NEMESIS_*   (core.clj:206-212)                                    */
  uint32_t nemesis_interval_ms;  /* --nemesis-interval, default 10 s (core.clj:214-217)                    */
  uint32_t client_timeout_ms;    /* client.clj:18-20 default 5000                                          */
  uint32_t quiesce_ms;           /* final-phase sleep, core.clj:78 (gen/sleep 10) -> 10000                 */
  uint64_t seed;                 /* base seed; instance i uses the stream keyed (seed, i) — the reference
                                    has no seed (SURVEY.md §0 finding 1)                                   */
  /* capacities (0 = let msim_config_defaults derive them from rate/time-limit) */
  uint32_t max_values;           /* bits per node set (distinct add/broadcast values)                      */
  uint32_t max_rows;             /* history rows per instance                                              */
  uint32_t max_payload_words;    /* u32 payload words per instance (read results, grudges)                 */
  uint32_t inbox_capacity;       /* envelopes queued per node endpoint in LDS                              */
  uint32_t spill_capacity;       /* further envelopes per node endpoint in an HBM spill area behind the LDS queue */
  uint32_t journal_capacity;     /* net-journal events per instance (journal.clj:53); 0 = journal off (default)   */
  /* txn-list-append generator ([upstream] jepsen.tests.cycle.append / elle.list-append gen; core.clj:167-199) */
  uint32_t key_count;            /* --key-count: keys worked on at once; 0 = 10 [upstream default, exponential key choice] */
  uint32_t max_txn_length;       /* --max-txn-length, default 4 (core.clj:191-194); min length is 1 [upstream]    */
  uint32_t max_writes_per_key;   /* --max-writes-per-key, default 16 (core.clj:196-199)                           */
  uint32_t proxy_service;        /* MSIM_SVC_*: which service lin_kv_proxy.rb talks to (its line 35 invites swapping it)  */
  uint32_t consistency_model;    /* MSIM_CM_*: --consistency-models for the transactional workloads (core.clj:160-165)            */
  uint32_t replication_words;    /* txn-rw-register: u32 words per instance for the txn lists of replicate messages; 0 = derive   */
} msim_config;

/* ---- outputs ------------------------------------------------------------------------------------- */

/* One history row = one Jepsen op map {:index :time :type :process :f :value [:error] [:final?]}
 * (SURVEY.md §8b "History surface").  16 bytes; :index is the row's position.
 *   time_len : bits 0..47 = :time in ns since test start; bits 48..63 = payload length in u32 words
 *              (0 = `value` is an immediate).
 *   packed   : bits 0-1 type (MSIM_T_*), 2-6 f (MSIM_F_*), 7-10 error (MSIM_ERR_*), 11 final?,
 *              12-31 process (MSIM_PROCESS_NEMESIS = :nemesis).
 *   value    : immediate (broadcast/add element, echo payload k of "Please echo k", partition spec) or
 *              offset in u32 words into the instance's payload area.
 * A read's :value is a bitmap over elements 0..32*len-1 (bit e set <=> e in the returned collection).
 * lin-kv ops (lin_kv.clj:53-67) pack their independent tuple into `value`: bits 0-7 key k, 8-15 v, 16-23 v'
 * (0xFF = nil): read [k v], write [k v], cas [k [v v']].
 * pn-counter ops (pn_counter.clj:22-58,134-137): `value` of an :add is the delta, of an :ok :read the counter, both as
 * two's-complement int32; a read's :invoke (and :fail / :info) has value nil.
 * unique-ids ops (unique_ids.clj:38-55): `value` of an :ok :generate is the flake id [time count node] packed as
 * time << 20 | count << 5 | node index (time in s, flake_ids.clj:19).
 * txn ops (txn_list_append.clj:27-39,54-60): `value` = payload offset, len = words of the transaction
 * [[f k v] ...].  One header word per micro-op: bit 0 f (0 = :r, 1 = :append), bits 1-15 key, bits 16-23 the appended
 * element (:append) or the length n of the list read (:r; 0xFF = nil, i.e. an :invoke or a key that does not exist);
 * a read of n elements is followed by ceil(n/4) words holding them one per byte, first element in the low byte.
 * kafka ops (workload/kafka.clj:203-232; keys 0..7, message values and offsets below 2047):
 *   :send   `value` = key | msg << 6 | offset << 17 — [[:send k msg]] while the offset field is 0x7FF (:invoke, :fail, :info),
 *           [[:send k [offset msg]]] for an :ok;
 *   :poll   :invoke (and :fail / :info): the {key offset} map the client asked with, `value` = payload offset, len = its entries
 *           (key | offset << 8 each; MSIM_NO_VALUE / 0 = the client has no assignment) — an engine abstraction, the op's :value is
 *           [[:poll]]; :ok: `value` / len = the poll_ok block: per requested key a header key | n << 8 | first offset << 16 followed
 *           by its n messages two per word (low half first): [[:poll {k [[offset msg] ...]}]];
 *   :assign `value` = payload offset, len = keys, one word per key (bit 31: :seek-to-beginning? true);  :crash carries no value. */
typedef struct msim_op {
  uint64_t time_len;
  uint32_t packed;
  uint32_t value;
} msim_op;

enum { MSIM_T_INVOKE = 0, MSIM_T_OK = 1, MSIM_T_FAIL = 2, MSIM_T_INFO = 3 };
enum { MSIM_F_ECHO = 0, MSIM_F_BROADCAST = 1, MSIM_F_READ = 2, MSIM_F_ADD = 3,
       MSIM_F_START_PARTITION = 4, MSIM_F_STOP_PARTITION = 5,
       MSIM_F_WRITE = 6, MSIM_F_CAS = 7, MSIM_F_TXN = 8, MSIM_F_GENERATE = 9,
       MSIM_F_SEND = 10, MSIM_F_POLL = 11, MSIM_F_ASSIGN = 12, MSIM_F_CRASH = 13 /* workload/kafka.clj:203-232 */ };
enum { MSIM_ERR_NONE = 0, MSIM_ERR_NET_TIMEOUT = 1 /* client.clj:158-162 */, MSIM_ERR_RPC = 2,
       /* RPC errors of resources/errors.edn, as :error [name text] (client.clj:163-172) */
       MSIM_ERR_TEMPORARILY_UNAVAILABLE = 3 /* code 11 */, MSIM_ERR_KEY_DOES_NOT_EXIST = 4 /* code 20 */,
       MSIM_ERR_PRECONDITION_FAILED = 5 /* code 22 */, MSIM_ERR_TXN_CONFLICT = 6 /* code 30 */,
       MSIM_ERR_TIMEOUT = 7 /* code 0: the NODE reports a timeout; not :definite? => :info */, MSIM_ERR_ABORT = 8 /* code 14 */ };
enum { MSIM_SPEC_ONE = 0, MSIM_SPEC_MAJORITY = 1, MSIM_SPEC_MAJORITIES_RING = 2, MSIM_SPEC_MINORITY_THIRD = 3 };
#define MSIM_PROCESS_NEMESIS 0xFFFFFu
#define MSIM_NO_VALUE 0xFFFFFFFFu   /* :value nil */

#define MSIM_OP_TIME_NS(op)  ((op).time_len & 0xFFFFFFFFFFFFull)
#define MSIM_OP_LEN(op)      ((uint32_t)((op).time_len >> 48))
#define MSIM_OP_TYPE(op)     ((op).packed & 3u)
#define MSIM_OP_F(op)        (((op).packed >> 2) & 31u)
#define MSIM_OP_ERR(op)      (((op).packed >> 7) & 15u)
#define MSIM_OP_FINAL(op)    (((op).packed >> 11) & 1u)
#define MSIM_OP_PROCESS(op)  ((op).packed >> 12)

/* What maelstrom.net.checker reports (net/checker.clj:28-41): journal :send / :recv event counts,
 * for all messages, messages involving a client (util.clj:12-16), and server<->server messages.
 * msg-count (distinct message ids, journal.clj:258-268) always equals send-count because every
 * message is journalled at send (net.clj:208) before the loss decision (net.clj:214). */
typedef struct msim_net_stats {
  uint64_t all_send, all_recv;
  uint64_t clients_send, clients_recv;
  uint64_t servers_send, servers_recv;
} msim_net_stats;

/* One net-journal event = `(Event. id time type message)` of net/journal.clj:53,220-239: a :send is logged
 * for every `send!` (before the loss decision, net.clj:208), a :recv for every delivery (net.clj:244).
 * 16 bytes; the event's :id is its position.  Feeds maelstrom.net.checker / net.viz.
 *   time_us : :time in microseconds since test start
 *   msg     : bits 8-31 message :id (net.clj:197), bit 7 = 1 for :recv / 0 for :send, bits 0-6 body :type (MSIM_M_*)
 *   a       : body payload (element / echo k / read payload ref (offset | words<<24) / replicate tick)
 *   route   : bits 0-7 src endpoint, 8-15 dest endpoint (nodes 0..n-1, then client slots), 16-31 low 16 bits of
 *             the body's msg_id (requests) or in_reply_to (replies); 0 = none */
typedef struct msim_event {
  uint32_t time_us;
  uint32_t msg;
  uint32_t a;
  uint32_t route;
} msim_event;
enum { MSIM_M_INIT = 1, MSIM_M_INIT_OK, MSIM_M_TOPOLOGY, MSIM_M_TOPOLOGY_OK, MSIM_M_ECHO, MSIM_M_ECHO_OK, MSIM_M_BROADCAST,
       MSIM_M_BROADCAST_OK, MSIM_M_READ, MSIM_M_READ_OK, MSIM_M_ADD, MSIM_M_ADD_OK, MSIM_M_REPLICATE,
       MSIM_M_WRITE, MSIM_M_WRITE_OK, MSIM_M_CAS, MSIM_M_CAS_OK, MSIM_M_ERROR,
       MSIM_M_REQUEST_VOTE, MSIM_M_REQUEST_VOTE_RES, MSIM_M_APPEND_ENTRIES, MSIM_M_APPEND_ENTRIES_RES,
       MSIM_M_TXN, MSIM_M_TXN_OK, MSIM_M_GENERATE, MSIM_M_GENERATE_OK, MSIM_M_REPLICATE_ACK,
       MSIM_M_TS, MSIM_M_TS_OK /* lin-tso, service.clj:121-123 */,
       MSIM_M_SEND, MSIM_M_SEND_OK, MSIM_M_POLL, MSIM_M_POLL_OK, MSIM_M_LIST_COMMITTED_OFFSETS, MSIM_M_LIST_COMMITTED_OFFSETS_OK,
       MSIM_M_COMMIT_OFFSETS, MSIM_M_COMMIT_OFFSETS_OK /* workload/kafka.clj:89-139 */ };

/* Per-instance bookkeeping (not part of the algorithmic output bytes). */
typedef struct msim_inst_meta {
  uint32_t n_rows;          /* history rows written                                                   */
  uint32_t n_payload_words; /* payload words written                                                  */
  uint32_t flags;           /* MSIM_FLAG_*                                                            */
  uint32_t n_rounds;        /* scheduler rounds executed (diagnostic)                                 */
  uint32_t n_events;        /* journal events produced (may exceed journal_capacity: then flagged)    */
  uint32_t reserved[3];
} msim_inst_meta;
enum { MSIM_FLAG_ROWS_OVERFLOW = 1u, MSIM_FLAG_PAYLOAD_OVERFLOW = 2u, MSIM_FLAG_INBOX_OVERFLOW = 4u,
       MSIM_FLAG_VALUES_OVERFLOW = 8u, MSIM_FLAG_ROUND_LIMIT = 16u, MSIM_FLAG_JOURNAL_OVERFLOW = 32u,
       MSIM_FLAG_ARENA_OVERRUN = 64u /* raft: a message descriptor was recycled while still in flight */ };

/* Result of the workload checker for one instance.  For broadcast / g-set this is jepsen's
 * `checker/set-full` result map (shape: doc/03-broadcast/01-broadcast.md:564-577, KAT-7); for echo the
 * pair comparison of workload/echo.clj:44-63. */
typedef struct msim_check_result {
  uint32_t valid;              /* 1 = :valid? true, 0 = false, 2 = :unknown                          */
  uint32_t attempt_count;
  uint32_t stable_count;
  uint32_t lost_count;
  uint32_t never_read_count;
  uint32_t stale_count;
  uint32_t duplicated_count;
  uint32_t error_count;        /* echo: mismatching pairs                                            */
  uint32_t stable_latency_ms[5]; /* quantiles 0, 0.5, 0.95, 0.99, 1 of stable latencies             */
  uint32_t op_count;           /* non-nemesis :invoke rows (net/checker.clj:55-58)                   */
  uint32_t ok_count, fail_count, info_count;  /* checker/stats                                        */
} msim_check_result;

typedef struct msim_ctx msim_ctx;

/* ---- entry points -------------------------------------------------------------------------------- */

/* Version of this ABI (compare with MSIM_ABI_VERSION). */
uint32_t msim_abi_version(void);

/* Number of HIP devices visible; 0 when none (never an error). */
int msim_device_count(void);

/* Fill *cfg with the reference's CLI defaults (core.clj:136-229 + [upstream] jepsen.cli: time-limit 60,
 * concurrency 1n) for the given workload/node count, and derive capacities. */
int msim_config_defaults(msim_config *cfg, uint32_t workload, uint32_t n_nodes);

/* Derive zero capacities in *cfg from rate/time-limit/topology; validates the rest.
 * Returns MSIM_E_INVALID (text in err) for configs the reference rejects, e.g. exponential latency
 * with mean 0 (net.clj:77 divides by zero). */
int msim_config_finalize(msim_config *cfg, char *err, size_t errlen);

/* Replaces core.clj:53-102 (test-map assembly) + jepsen.core/run! set-up: builds an engine for one
 * test configuration on HIP device `device`. */
int msim_create(const msim_config *cfg, int device, msim_ctx **out, char *err, size_t errlen);

/* Replaces N_instances sequential `lein run test ...` executions (core.clj:267-284): simulates global
 * instances [first_instance, first_instance + n_instances) to completion on the device.  Blocking.
 * Outputs stay resident in HBM (see msim_device_buffers) until the next run/destroy. */
int msim_run(msim_ctx *ctx, uint64_t first_instance, uint32_t n_instances);

/* As msim_run, but only enqueues the kernel on `hip_stream` (a hipStream_t; NULL = default stream)
 * and returns; the caller synchronises the stream. */
int msim_run_async(msim_ctx *ctx, uint64_t first_instance, uint32_t n_instances, void *hip_stream);

/* Runs the workload checker for every instance of the last run, reading the HBM-resident histories.  On the device:
 * set-full (broadcast, g-set), echo, lin-kv's per-key linearizability (one wavefront per history; a history that
 * exceeds what a wavefront's registers hold is finished by the host search) and the clean case of list-append (elle: no
 * anomaly + acyclic dependency graph; a history that is not provably clean is analysed by the host) — see
 * msim_check_host_rechecks.  On the host cores, after a fetch: rw-register (elle), pn-counter, unique-ids.  Blocking.  Results via msim_check_results. */
int msim_check(msim_ctx *ctx);

/* Developer switches of a context (A/B comparisons, tracing; never needed for normal use) — the same bits the environment variable
 * MSIM_DEV_FLAGS carries, which is ORed in: 0x100 round limit x 20, 0x200 run the one-cluster-per-wavefront kernels, 0x400 fail
 * instead of falling back to them, 0x800 keep the workload checkers on the host cores, 0x1000 time the checkers' passes on stderr. */
int msim_set_dev_flags(msim_ctx *ctx, uint32_t flags);

/* How many histories of the last msim_check the device handed to the host (lin-kv: the search needed more than 512
 * configurations; txn-list-append: not provably clean, i.e. the host analysed and classified it; else 0). */
uint32_t msim_check_host_rechecks(const msim_ctx *ctx);

/* txn-list-append: checks `n_histories` histories given on the host — rows / payload words of history i at row_offsets[i] /
 * payload_offsets[i] (n + 1 offsets each) — as msim_check does for the histories of a run: the device proves the clean ones clean
 * (no anomaly, acyclic dependency graph incl. realtime edges), the host analysis of msim_check_txn_rows finishes the others
 * (strict-serializable).  out[i] is what msim_check_txn_rows gives for history i. */
int msim_check_txn_batch(int device, const msim_op *rows, const uint64_t *row_offsets, const uint32_t *payload, const uint64_t *payload_offsets,
                         uint32_t n_histories, msim_check_result *out);

/* txn-rw-register: the same for the rw-register analysis under `consistency_model` (MSIM_CM_*): the device proves a history free of
 * everything the model proscribes (csrc/rw_check_dev.hip), the host analysis of msim_check_rw_rows finishes the others; *n_host (may be
 * null) = how many went to the host.  For a history the device proves valid, out[i] carries :valid?, the counts, the non-cycle anomalies
 * seen (none proscribed) in error_count and the edges built in lost_count; the allowed cycle classes are not searched for. */
int msim_check_rw_batch(int device, const msim_op *rows, const uint64_t *row_offsets, const uint32_t *payload, const uint64_t *payload_offsets,
                        uint32_t n_histories, uint32_t consistency_model, msim_check_result *out, uint32_t *n_host);

/* pn-counter / g-counter: the same for the counter checker (workload/pn_counter.clj:84-123); out[i] is what msim_check_pn_rows gives. */
int msim_check_pn_batch(int device, const msim_op *rows, const uint32_t *n_rows, uint32_t max_rows, uint32_t n_histories, msim_check_result *out);

/* broadcast / g-set (set-full) and echo: checks `n_histories` histories given on the host with the device checker of msim_check
 * (csrc/checker.hip: [upstream] jepsen.checker/set-full as workload/broadcast.clj:216-228 and workload/g_set.clj:62 use it; the pair
 * comparison of workload/echo.clj:44-63).  History i lies in the slabs rows + i * max_rows (n_rows[i] rows used) and
 * payload + i * max_payload_words; `workload` is MSIM_WL_BROADCAST / MSIM_WL_G_SET / MSIM_WL_ECHO, `concurrency` the number of worker
 * threads (process mod concurrency pairs a completion with its invocation), `max_values` bounds the elements (<= 2048). */
int msim_check_set_full_batch(int device, uint32_t workload, uint32_t concurrency, const msim_op *rows, const uint32_t *n_rows, uint32_t max_rows,
                              const uint32_t *payload, uint32_t max_payload_words, uint32_t max_values, uint32_t n_histories, msim_check_result *out);

/* kafka: the checker behind msim_check for MSIM_WL_KAFKA on ONE history given on the host (csrc/kafka_check.cpp): the anomalies
 * workload/kafka.clj:21-70 describes ([upstream] jepsen.tests.kafka's checker is not vendored: parity unpinned).  out->error_count =
 * MSIM_KAFKA_* bits, valid = 1 iff none (2 = :unknown: nothing acknowledged, nothing polled), attempt_count / stable_count = sends invoked
 * / acknowledged, lost_count = lost writes, never_read_count = unobserved writes (reported, not an error), duplicated_count. */
enum { MSIM_KAFKA_LOST_WRITE = 1u, MSIM_KAFKA_NONMONOTONIC_POLL = 2u, MSIM_KAFKA_NONMONOTONIC_SEND = 4u, MSIM_KAFKA_POLL_SKIP = 8u,
       MSIM_KAFKA_INT_NONMONOTONIC_POLL = 16u, MSIM_KAFKA_INT_POLL_SKIP = 32u, MSIM_KAFKA_INCONSISTENT_OFFSETS = 64u,
       MSIM_KAFKA_DUPLICATE = 128u, MSIM_KAFKA_ABORTED_READ = 256u, MSIM_KAFKA_MALFORMED = 512u /* a row that does not decode */ };
int msim_check_kafka_rows(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, msim_check_result *out);
/* kafka: `n_histories` histories given on the host (rows / payload words of history i at row_offsets[i] / payload_offsets[i]) as msim_check
 * does for the histories of a run: the device proves a history free of every anomaly above (csrc/kafka_check_dev.hip: the tables of the
 * checker in LDS, one workgroup per history), msim_check_kafka_rows' checker classifies the others; `concurrency` = the test's worker
 * threads (process mod concurrency is the worker); *n_host (may be null) = how many went to the host.  out[i] = what
 * msim_check_kafka_rows gives for history i. */
int msim_check_kafka_batch(int device, const msim_op *rows, const uint64_t *row_offsets, const uint32_t *payload, const uint64_t *payload_offsets,
                           uint32_t n_histories, uint32_t concurrency, msim_check_result *out, uint32_t *n_host);

/* unique-ids: checks `n_histories` histories given on the host — history i in the slab rows + i * max_rows, n_rows[i] rows used —
 * with the device checker of msim_check ([upstream] jepsen.checker/unique-ids); out[i] is what msim_check_unique_rows gives. */
int msim_check_unique_batch(int device, const msim_op *rows, const uint32_t *n_rows, uint32_t max_rows, uint32_t n_histories, msim_check_result *out);

/* lin-kv: checks `n_histories` histories given on the host — history i = rows[row_offsets[i] .. row_offsets[i + 1]) — with the
 * device search of msim_check on HIP device `device`; out[i] is what msim_check_lin_kv_rows gives for history i. */
int msim_check_lin_kv_batch(int device, const msim_op *rows, const uint64_t *row_offsets, uint32_t n_histories, msim_check_result *out);

/* Host-only utility behind msim_check for lin-kv: per-key linearizability (CAS-register model) of ONE history given
 * as rows — what `independent/checker` + Knossos do for workload/lin_kv.clj:84.  out->valid: 1 linearizable, 0 not,
 * 2 unknown; attempt_count = keys, error_count = non-linearizable keys.  Needs no device. */
int msim_check_lin_kv_rows(const msim_op *rows, uint32_t n_rows, msim_check_result *out);

/* Host-only utility behind msim_check for txn-list-append: the list-append analysis of [upstream] elle
 * (jepsen.tests.cycle.append, txn_list_append.clj:142) for the default --consistency-models strict-serializable
 * (core.clj:160-165): per-key version orders from the longest reads, ww/wr/rw + realtime dependency graph, cycle search,
 * plus the non-cycle anomalies.  out->error_count = bitmask of MSIM_ANOMALY_* found; valid = 1 iff none, 2 (unknown) if
 * no transaction completed :ok; attempt_count = transactions, stable_count = :ok transactions, lost_count = edges of
 * the dependency graph, stale_count = transactions inside some cycle.  Needs no device. */
enum { MSIM_ANOMALY_G0 = 1u, MSIM_ANOMALY_G1A = 2u, MSIM_ANOMALY_G1B = 4u, MSIM_ANOMALY_G1C = 8u, MSIM_ANOMALY_G_SINGLE = 16u,
       MSIM_ANOMALY_G2 = 32u, MSIM_ANOMALY_INTERNAL = 64u, MSIM_ANOMALY_DUPLICATE_ELEMENTS = 128u,
       MSIM_ANOMALY_INCOMPATIBLE_ORDER = 256u, MSIM_ANOMALY_REALTIME = 512u /* cycle needs a realtime edge (G*-realtime) */,
       MSIM_ANOMALY_DIRTY_UPDATE = 1024u, MSIM_ANOMALY_CYCLIC_VERSIONS = 2048u /* rw-register: a key's inferred version order has a cycle */ };
int msim_check_txn_rows(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, msim_check_result *out);

/* The anomalies (MSIM_ANOMALY_* bits) a consistency model forbids, after Adya's PL levels as elle.consistency-model
 * arranges them [upstream, not vendored]: read-uncommitted G0; read-committed + G1a G1b G1c; snapshot-isolation + G-single
 * and internal; serializable + G2; strict-serializable + the -realtime cycles.  Duplicate elements, incompatible orders,
 * dirty updates and cyclic versions make a history invalid under every model. */
uint32_t msim_proscribed_anomalies(uint32_t consistency_model);
/* The subset of `anomalies` (an error_count of msim_check_txn_rows / msim_check_rw_rows) that makes a history invalid under
 * `consistency_model` — what `--consistency-models` (core.clj:160-165) decides in the reference: a cycle that only closes through a
 * realtime edge (MSIM_ANOMALY_REALTIME: the G*-realtime anomalies) counts against strict-serializable alone
 * (doc/05-datomic/04-optimization.md:311-345: G-single-realtime, "Everything looks good" under serializable). */
uint32_t msim_violated_anomalies(uint32_t anomalies, uint32_t consistency_model);

/* Host-only utility behind msim_check for txn-rw-register: the rw-register analysis of [upstream] elle
 * (jepsen.tests.cycle.wr with :wfr-keys? true, txn_rw_register.clj:150-168): writes are unique per key; version orders come
 * from the initial nil and from writes that follow a read of the same key inside one transaction; ww/wr/rw (+ realtime)
 * dependency graph, cycle search, G1a / G1b / internal.  Micro-op words: bit 0 = write, bits 1-15 key, bits 16-23 value
 * (0xFF = nil).  Same result fields as msim_check_txn_rows; valid is judged against `consistency_model`. */
int msim_check_rw_rows(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t consistency_model,
                       msim_check_result *out);

/* Host-only utility behind msim_check for pn-counter: the checker of workload/pn_counter.clj:84-123 — every :ok read
 * marked :final? must lie in the acceptable set = sum of the :ok adds plus any subset of the :info adds, kept as merged
 * closed integer ranges.  out->valid 1/0; attempt_count = final reads, error_count = final reads outside the set,
 * stable_count = ranges in the set.  `ranges` (may be NULL) receives up to `cap` [lower, upper] pairs in ascending order,
 * *n_ranges the number of ranges (the reference's :acceptable).  Needs no device. */
/* Host-only utility behind msim_check for unique-ids: [upstream] jepsen.checker/unique-ids (unique_ids.clj:67) — every
 * :ok :generate value must be distinct.  out->attempt_count = :attempted-count (invocations), ok_count = :acknowledged-count,
 * duplicated_count = number of distinct values acknowledged more than once, stable_latency_ms[0..1] = :range [min max] of the
 * packed ids; valid = 1 iff duplicated_count == 0.  Needs no device. */
int msim_check_unique_rows(const msim_op *rows, uint32_t n_rows, msim_check_result *out);

int msim_check_pn_rows(const msim_op *rows, uint32_t n_rows, msim_check_result *out, int64_t *ranges, uint32_t cap, uint32_t *n_ranges);

/* Host-only utility: one history as the text of Jepsen's history.edn — one op map per line,
 * {:type :f :value :time :process :index [:error] [:final?]} with the :value shapes of the workload in cfg —
 * for a JVM side that wants to hand the history to an unchanged checker without decoding rows itself.  Writes at most `cap`
 * bytes including the terminating NUL; *needed (may be NULL) receives the size required; cap == 0 only queries it.
 * Returns MSIM_E_RANGE if the buffer is too small or a row points outside the payload.  Needs no device. */
int msim_history_edn_rows(const msim_config *cfg, const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words,
                          char *out, size_t cap, size_t *needed);

/* maelstrom.checker/availability-checker (checker.clj:6-39; part of every test's composed checker, core.clj:98): did the history
 * meet `--availability` (core.clj:149)?  :ok-fraction = :ok ops / :invoke ops (1 for an empty history); :valid? = true for nil,
 * fraction == 1 for :total, availability <= fraction for a number.  msim_check_availability judges every history of the last run
 * on the device (out[n_out >= n_instances]); msim_check_availability_rows one history on the host. */
enum { MSIM_AVAIL_NIL = 0, MSIM_AVAIL_TOTAL = 1, MSIM_AVAIL_FRACTION = 2 };
typedef struct msim_availability {
  uint32_t valid;          /* 1 / 0 */
  float ok_fraction;       /* (float (/ ok-count invoke-count)) */
  uint32_t ok_count, invoke_count;
} msim_availability;
int msim_check_availability(msim_ctx *ctx, uint32_t mode, double availability, msim_availability *out, uint32_t n_out);
int msim_check_availability_rows(const msim_op *rows, uint32_t n_rows, uint32_t mode, double availability, msim_availability *out);

/* Host-only utility: one instance's net journal in the reference's on-disk format — the bytes of a
 * `store/<test>/net-journal/<stripe>.fressian` file as maelstrom.net.journal writes it (net/journal.clj:55-141,220-239: one
 * Fressian "ev" struct per log-send! / log-recv!, "msg" structs with cached src / dest, bodies through write-body!), for the
 * unchanged maelstrom.net.checker (net/checker.clj:28-70) and net/viz.clj.  `payload` = the instance's payload area (read_ok
 * bodies refer to it).  Same calling convention as msim_history_edn_rows: cap == 0 only queries *needed.  Needs no device. */
int msim_journal_fressian_rows(const msim_config *cfg, const msim_event *events, uint32_t n_events, const uint32_t *payload, uint32_t n_words,
                               unsigned char *out, size_t cap, size_t *needed);

/* Copies the last run's outputs to host memory (pinned, owned by ctx, valid until next run/destroy). */
int msim_fetch(msim_ctx *ctx);

/* The first half of msim_fetch for callers that keep the GPU busy meanwhile: copies meta and stats, compacts rows and payload on the
 * device and queues the two PCIe copies, then returns; a later msim_fetch waits for them.  Called right after msim_check and before
 * the next batch is started on another context, the compaction kernels sit in the queue ahead of that batch's simulation and the
 * copies run beside it (bench.py's value_incl_fetch). */
int msim_fetch_begin(msim_ctx *ctx);

/* Per-instance views into the fetched outputs (require msim_fetch). `inst` is 0-based within the run. */
int msim_history(msim_ctx *ctx, uint32_t inst, const msim_op **ops, uint32_t *n_ops,
                 const uint32_t **payload, uint32_t *n_words);
int msim_net_stats_get(msim_ctx *ctx, uint32_t inst, msim_net_stats *out);
/* The instance's net journal (requires journal_capacity > 0 and msim_fetch). */
int msim_journal(msim_ctx *ctx, uint32_t inst, const msim_event **events, uint32_t *n_events);
int msim_meta(msim_ctx *ctx, uint32_t inst, msim_inst_meta *out);
/* Checker results of the last msim_check (copied to host on demand). */
int msim_check_results(msim_ctx *ctx, const msim_check_result **results, uint32_t *n);

/* Device-resident output buffers of the last run, for zero-copy consumers (RCCL gather, GPU checkers).
 * rows: n_instances * max_rows msim_op; payload: n_instances * max_payload_words u32;
 * stats: n_instances msim_net_stats; meta: n_instances msim_inst_meta; check: n_instances msim_check_result;
 * journal: n_instances * journal_capacity msim_event. */
typedef struct msim_device_buffers {
  void *rows; void *payload; void *stats; void *meta;
  void *check;                 /* n_instances msim_check_result (valid after msim_check) */
  void *journal;               /* n_instances * journal_capacity msim_event (NULL when the journal is off) */
  uint64_t rows_bytes, payload_bytes, stats_bytes, meta_bytes, check_bytes, journal_bytes;
  uint32_t n_instances, max_rows, max_payload_words, journal_capacity;
} msim_device_buffers;
int msim_device_buffers_get(msim_ctx *ctx, msim_device_buffers *out);

/* ---- multi-GPU ensemble: variable-length history gather over RCCL / xGMI ------------------------------------
 * Instances are independent, so an ensemble shards across the GPUs of a node with no collective on the data path: one engine
 * context per device simulates its block of instances.  The one exchange step is the gather of the emitted histories to a root
 * rank at the end of a batch — what `jepsen.store` collecting every test's history in one JVM amounts to.  It runs behind this
 * ABI so that a JVM (or any) host gets it without torch:
 *     rank 0: msim_comm_unique_id(id);   id reaches every rank through the host's own channel (files, sockets, JVM RPC)
 *     every rank: msim_comm_init(ctx, id, rank, world);  ...  msim_run(ctx, first_of_rank, n);  msim_gather(ctx, root, &g);
 * msim_gather = device-side compaction of the used prefix of every slab (the kernels behind msim_fetch) -> ncclAllGather of the
 * four byte counts per rank -> ONE grouped ncclSend per slab kind from every peer / ncclRecv on the root: what crosses xGMI is
 * the sum of the history bytes, once.  RCCL is loaded at the first call (librccl.so, dlopen): hosts that never gather need not
 * have it.  With world == 1 (or without msim_comm_init) the gather is the compaction alone. */
#define MSIM_COMM_ID_BYTES 128
typedef struct msim_gathered {
  /* on the root: device buffers owned by ctx (valid until the next msim_gather / msim_destroy), every rank's part in rank order,
   * instances in run order inside a rank; NULL on the other ranks */
  void *rows;      /* msim_op x total rows            */
  void *payload;   /* u32 x total payload words       */
  void *meta;      /* msim_inst_meta x n_instances    */
  void *stats;     /* msim_net_stats x n_instances    */
  uint64_t rows_bytes, payload_bytes, meta_bytes, stats_bytes;
  uint64_t bytes_received;   /* bytes that crossed the links into the root (0 on the other ranks and at world == 1) */
  uint32_t n_instances;      /* over all ranks */
  uint32_t world, rank;
  float ms;                  /* compaction + exchange on this rank, HIP events on the engine's stream */
} msim_gathered;
int msim_comm_unique_id(unsigned char id[MSIM_COMM_ID_BYTES]);
int msim_comm_init(msim_ctx *ctx, const unsigned char id[MSIM_COMM_ID_BYTES], int rank, int world);
int msim_gather(msim_ctx *ctx, int root, msim_gathered *out);

/* Kernel time of the last msim_run in milliseconds, measured with HIP events on the engine's stream. */
int msim_last_kernel_ms(msim_ctx *ctx, float *sim_ms, float *check_ms);

/* The finalized configuration the ctx runs with. */
int msim_get_config(const msim_ctx *ctx, msim_config *out);

/* Diagnostic: validates the engine's DPP wave primitives against shuffle-based references on `device`.
 * 0 = ok, >0 = mismatching lanes, <0 = MSIM_E_*. */
int msim_selftest_wave(int device);

const char *msim_last_error(const msim_ctx *ctx);
void msim_destroy(msim_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MAELSIM_H */
