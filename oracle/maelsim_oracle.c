/* maelsim_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded, deliberately simple restatement of Maelstrom's hot path for ONE test
 * instance at a time, in deterministic virtual time.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product (libmaelsim.so, maelstrom_amd/) never does.
 *
 * PARITY PINNING: the reference (Clojure/JVM + Ruby/babashka node processes) cannot
 * run in this environment and is non-deterministic (no seed: net.clj:187,205,214), so there is no
 * byte-level golden history to pin against.  What this oracle IS pinned against
 * (tests/test_oracle_kat.py, tests/test_golden_transitions.py):
 *   - the reference docs' exact message-count known-answers KAT-1..KAT-5,
 *   - golden node-transition vectors recorded from the reference's own runnable node programs
 *     (demo/python/echo.py, demo/js/gossip.js, demo/js/crdt_gset.js; tests/golden/make_golden.py),
 *   - checker verdicts (set-full :valid? true) on every emitted history,
 *   - the Raft node: golden transitions from a real demo/python/raft.py process, and whole runs replayed through the
 *     reference's own RaftNode objects (tests/test_raft_reference_replay.py; digests in tests/golden/raft_replay_digests.json),
 *   - the transactional node + lin-kv service: golden conversation with real demo/js/single_key_txn.js processes,
 *   - pn-counter: the reference's checker vectors (pn_counter_test.clj:10-36) and golden transitions from crdt_pn_counter.js,
 *   - g-set / pn-counter / g-counter / echo / txn-list-append / the acknowledged retrying broadcast: whole runs reproduced line by
 *     line by real node.js processes of demo/js/crdt_gset.js, crdt_pn_counter.js, echo.js, single_key_txn.js and gossip.js (the
 *     latter with its RPC timers in virtual time; tests/test_js_reference_replay.py, digests in tests/golden/),
 *   - cross-checks by independent transliterations that replay this oracle's own journal: net.clj send!/recv!
 *     (tests/test_net_semantics.py), client.clj (tests/test_client_semantics.py), service.clj + lin_kv_proxy.rb +
 *     single_key_txn.clj with real values (tests/services_ref.py), txn_rw_register_hat.clj (tests/hat_ref.py).
 * Everything that comes from un-vendored upstream Jepsen (generator interpreter, partition-package)
 * is restated from its published behaviour and is "parity unpinned" (DESIGN.md §3).
 *
 * What is restated, with the reference lines each part follows:
 *   network   : net.clj:189-221 send!  (id := ++next-message-id; latency 0 if a client is involved,
 *               util.clj:7-16 / net.clj:178-187; journal :send BEFORE the loss decision :208/:214)
 *               net.clj:223-247 recv!  (poll the min-deadline envelope even if not yet due :228;
 *               partition check at poll time :234; sleep (long dt) ms :236-238; journal :recv :244)
 *               net.clj:39-40 queue order = deadline (ties: this engine defines (deadline, id))
 *               net.clj:65-77 latency distributions; net.clj:109-113 drop!/heal!
 *   process   : process.clj:154-166 one message at a time per node (stdin thread loop)
 *   client    : client.clj:66-117 one outstanding RPC, stale replies skipped, timeout;
 *               client.clj:153-172 with-errors -> :fail / :info
 *   db        : db.clj:46-69 init handshake; broadcast.clj:195-197 topology RPC in setup!
 *   test map  : core.clj:67-80 generator phases (stagger, time-limit, nemesis, sleep 10, final reads)
 *   workloads : echo.clj:72-75, broadcast.clj:40-185 (topologies) :237-240 (generator),
 *               g_set.clj:59-61
 *   nodes     : echo (demo/ruby/echo.rb:20-41), broadcast variants (doc/03-broadcast/01-broadcast.md:
 *               525-547, 02-performance.md:61-67, :406-441, demo/ruby/broadcast.rb:29-47),
 *               g-set (demo/ruby/g_set.rb:8-39, node.rb:129-138 periodic task)
 *   stats     : net/checker.clj:28-41, journal.clj:241-337
 *
 * The deterministic schedule ("rounds", canonical id order, counter-based RNG) is specified in
 * DESIGN.md §2; this file and the HIP engine implement that text independently.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/maelsim.h"
#include "log2_table.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

#define INF 0xFFFFFFFFu
#define MAXN 128
#define MW 4 /* mask words: MAXN/32 */

/* message types (doc/protocol.md, doc/workloads.md) */
enum { M_INIT = 1, M_INIT_OK, M_TOPOLOGY, M_TOPOLOGY_OK, M_ECHO, M_ECHO_OK, M_BROADCAST, M_BROADCAST_OK,
       M_READ, M_READ_OK, M_ADD, M_ADD_OK, M_REPLICATE,
       M_WRITE, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR,                                   /* lin-kv RPCs, doc/workloads.md */
       M_REQUEST_VOTE, M_REQUEST_VOTE_RES, M_APPEND_ENTRIES, M_APPEND_ENTRIES_RES,      /* raft.py:290-297,412-420 */
       M_TXN, M_TXN_OK, M_GENERATE, M_GENERATE_OK, M_REPLICATE_ACK, M_TS, M_TS_OK };                                                                /* txn_list_append.clj:73-80 */

/* RNG streams */
enum { S_GEN = 1, S_GEN2 = 2, S_GEN3 = 3, S_LATENCY = 4, S_LOSS = 5, S_NODE = 11, S_SVC = 12,
       S_NEM_STAGGER = 7, S_NEM_SPEC = 8, S_NEM_SHUFFLE = 9, S_NEM_PICK = 10 };

enum { PH_INIT, PH_INIT_WAIT, PH_TOPO, PH_TOPO_WAIT, PH_MAIN_START, PH_MAIN, PH_DRAIN, PH_NEM_FINAL,
       PH_SLEEP, PH_FINAL, PH_FINAL_WAIT, PH_DONE, PH_KF_FINAL /* kafka: final polls */ };

enum { K_NONE = 0, K_INIT, K_TOPO, K_OP };

struct rext;
typedef struct { u32 deadline, id, a, b; u8 src, type; struct rext *r; } qent;
typedef struct { qent *v; u32 n, cap; } inbox_t;
typedef struct { u32 value, next_retry; } task_t;
typedef struct { task_t *v; u32 n, cap, head; } tasks_t;
typedef struct { u8 src_ep, dest_ep, type; u32 a, b; struct rext *r; } outmsg;

typedef struct {
  msim_config cfg;
  u32 N, C, CS, E, W; /* nodes, workers, client slots, endpoints, words per node set */
  u32 S;              /* services (endpoints after the client slots): 1 = lin-kv for the txn workload */
  struct txn_s *txn;  /* txn-list-append state (txn_nodes.inc) */
  struct svc_s *svc;  /* proxy node + key-value services (svc_nodes.inc) */
  struct hat_s *hat;  /* txn-rw-register highly-available-transactions node (hat_nodes.inc) */
  struct mk_s *mk;    /* txn-list-append over thunks in lww-kv and a root map in lin-kv (mk_nodes.inc) */
  struct dt_s *dt;    /* txn-list-append over a persistent hash tree in lww-kv and a root pointer in lin-kv (dt_nodes.inc) */
  struct kafka_s *kafka; /* kafka workload: node, lin-kv contents, clients' offsets, generator (kafka_nodes.inc) */
  u32 *rtrace; u32 n_rtrace, cap_rtrace; /* test hook: what every Raft node did in every round (oracle_raft_schedule) */
  u64 key;
  u32 adj[MAXN][MW];
  /* net (net.clj:79-103) */
  u32 next_msg_id;
  msim_net_stats st;
  u32 part[MAXN][MW]; /* part[dest] = set of src whose packets dest drops (net.clj:109-110) */
  inbox_t *inbox;
  qent *committed; u8 *has_committed; u32 *deliver_at;
  /* nodes */
  u32 *seen;            /* N * W words */
  u32 *node_msg_id;     /* per-node RPC msg_id counter (node.rb:91-98) */
  u32 *nbr_known;       /* node got its topology */
  tasks_t *tasks;       /* ack/retry: FIFO of gossip threads per node */
  u32 *unacked;         /* ack/retry: [node][value][MW] un-acked neighbour sets */
  u32 *timer_next;      /* g-set replicate timer */
  u32 *tick;            /* g-set: replicate ticks so far, per node */
  u32 *flake;           /* flake ids: [node][2] = {last time (s), counter within it}, flake_ids.clj:10-14 */
  struct rnode_s *raft; /* raft: per-node state (raft_nodes.inc) */
  u32 cur_key, key_procs; u32 *key_reg; /* lin-kv generator: current key; per thread the process id it registered on that key (+1) */
  u32 **snap; u32 n_snap, cap_snap; /* replicate_full payload snapshots */
  /* clients */
  struct cl { u8 busy, kind, mark; u32 want, timeout_at, next_msg_id, f, value, process, m_f, m_value, m_final; } *cl;
  /* scheduler */
  u32 phase, T, cutoff, gen_next, gen_k, next_value, nem_next, nem_j, sleep_until, loss_on;
  u32 rounds;
  /* outbox for the current round */
  outmsg *out; u32 n_out, cap_out;
  /* outputs */
  msim_op *rows; u32 *payload; msim_inst_meta meta;
  msim_event *journal;  /* net journal (journal.clj:53,220-239), optional */
  struct { u8 on; u32 type, f, err, final, process, value, len; } *pend; /* completion rows of the current round, by slot */
  int defer_rows;
} sim_t;

/* ---- RNG: counter-based, keyed (seed, instance, stream, counter) --------------------------------- */
static u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static u64 inst_key(u64 seed, u64 instance) { return mix64(seed + 0x9E3779B97F4A7C15ull * (instance + 1)); }
static u64 draw64(const sim_t *s, u32 stream, u64 ctr) {
  u64 x = ((u64)stream << 48) | ctr;
  return mix64(s->key + x * 0x9E3779B97F4A7C15ull);
}
static u32 draw32(const sim_t *s, u32 stream, u64 ctr) { return (u32)(draw64(s, stream, ctr) >> 32); }
static u32 scale32(u32 r, u32 n) { return (u32)(((u64)r * n) >> 32); } /* uniform int in [0, n) */

/* -ln(u) in Q16 for u = (r+1)/2^32, integer-only (table + linear interpolation of log2). */
static u32 neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0; /* u = 1 */
  u32 v = r + 1;
  u32 e = 31 - (u32)__builtin_clz(v);
  u32 m = v << (31 - e);            /* bit 31 set */
  u32 idx = (m >> 23) & 0xFF;
  u32 f = (m >> 7) & 0xFFFF;
  u32 l0 = oracle_log2_q24[idx], l1 = oracle_log2_q24[idx + 1];
  u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16); /* log2(v) in Q24 */
  u32 d = (32u << 24) - lg;                                   /* -log2(u) in Q24 */
  return (u32)(((u64)d * 2977044472ull) >> 40);               /* * ln2 (Q32) -> Q16 */
}

/* net.clj:178-187 latency-for, in ms */
static u32 latency_ms(const sim_t *s, u32 msg_id, int involves_client) {
  if (involves_client) return 0;
  u32 mean = s->cfg.latency_mean_ms;
  switch (s->cfg.latency_dist) {
    case MSIM_LAT_CONSTANT: return mean;
    case MSIM_LAT_UNIFORM: return scale32(draw32(s, S_LATENCY, msg_id), 2 * mean); /* [0, 2*mean) */
    default: return (u32)(((u64)mean * neg_ln_q16(draw32(s, S_LATENCY, msg_id))) >> 16);
  }
}

/* ---- small helpers ------------------------------------------------------------------------------- */
static int bit(const u32 *m, u32 i) { return (m[i >> 5] >> (i & 31)) & 1; }
static void setbit(u32 *m, u32 i) { m[i >> 5] |= 1u << (i & 31); }
static void clrbit(u32 *m, u32 i) { m[i >> 5] &= ~(1u << (i & 31)); }
static int is_client(const sim_t *s, u32 ep) { return ep >= s->N && ep < s->N + s->CS; }

/* lin_kv.clj:74-76, txn_list_append.clj:124-126: Reusable clients are not re-opened after a crash */
static int txn_workload(const sim_t *s) { return s->cfg.workload == MSIM_WL_TXN_LIST_APPEND || s->cfg.workload == MSIM_WL_TXN_RW_REGISTER; }
/* lin_kv.clj:74-76, txn_list_append.clj:124-126, txn_rw_register.clj:137-139 */
static int reusable_clients(const sim_t *s) { return s->cfg.workload == MSIM_WL_LIN_KV || txn_workload(s) || s->cfg.workload == MSIM_WL_UNIQUE_IDS; }

static void inbox_push(sim_t *s, u32 ep, qent q) {
  inbox_t *b = &s->inbox[ep];
  if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 8; b->v = (qent *)realloc(b->v, b->cap * sizeof(qent)); }
  b->v[b->n++] = q;
  u32 lim = is_client(s, ep) ? (reusable_clients(s) && (s->cfg.workload != MSIM_WL_UNIQUE_IDS || s->cfg.node_program == MSIM_NODE_TSO_IDS) ? 32u : s->cfg.workload == MSIM_WL_KAFKA ? 8u : 2u) /* Reusable clients collect late replies */
                             : s->cfg.inbox_capacity + s->cfg.spill_capacity; /* engine capacities: overflow is flagged, never silent */
  if (b->n > lim) s->meta.flags |= MSIM_FLAG_INBOX_OVERFLOW;
}

/* `sender` emits a message whose :src is `src` (a node may forward a client's message unchanged, raft.py:543-546) */
static void out_send_x(sim_t *s, u32 sender, u32 src, u32 dest, u32 type, u32 a, u32 b, struct rext *r) {
  (void)sender;
  if (s->n_out == s->cap_out) { s->cap_out = s->cap_out ? s->cap_out * 2 : 64; s->out = (outmsg *)realloc(s->out, s->cap_out * sizeof(outmsg)); }
  outmsg m = {(u8)src, (u8)dest, (u8)type, a, b, r};
  s->out[s->n_out++] = m;
}
static void out_send(sim_t *s, u32 src, u32 dest, u32 type, u32 a, u32 b) { out_send_x(s, src, src, dest, type, a, b, NULL); }

static void add_row(sim_t *s, u32 type, u32 f, u32 err, u32 final, u32 process, u32 value, u32 len) {
  if (s->meta.n_rows >= s->cfg.max_rows) { s->meta.flags |= MSIM_FLAG_ROWS_OVERFLOW; return; }
  msim_op *r = &s->rows[s->meta.n_rows++];
  r->time_len = ((u64)s->T * 1000ull) | ((u64)len << 48);
  r->packed = type | (f << 2) | (err << 7) | (final << 11) | (process << 12);
  r->value = value;
}

/* journal/log-send! / log-recv! (journal.clj:225-239): event id = position */
static void jlog(sim_t *s, u32 recv, u32 id, u32 type, u32 a, u32 b, u32 src, u32 dest) {
  if (!s->cfg.journal_capacity) return;
  if (s->meta.n_events < s->cfg.journal_capacity) {
    msim_event *e = &s->journal[s->meta.n_events];
    e->time_us = s->T; e->msg = (id << 8) | (recv << 7) | type; e->a = a; e->route = src | (dest << 8) | ((b & 0xFFFFu) << 16);
  } else s->meta.flags |= MSIM_FLAG_JOURNAL_OVERFLOW;
  s->meta.n_events++;
}

/* returns payload offset or INF on overflow */
static u32 payload_alloc(sim_t *s, u32 words) {
  if (s->meta.n_payload_words + words > s->cfg.max_payload_words) { s->meta.flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; return INF; }
  u32 off = s->meta.n_payload_words;
  s->meta.n_payload_words += words;
  return off;
}

/* ---- topologies (broadcast.clj:40-185) ----------------------------------------------------------- */
static void link2(sim_t *s, u32 a, u32 b) { setbit(s->adj[a], b); setbit(s->adj[b], a); }
static void build_topology(sim_t *s) {
  u32 n = s->N;
  memset(s->adj, 0, sizeof s->adj);
  switch (s->cfg.topology) {
    case MSIM_TOPO_GRID: { /* broadcast.clj:40-65: side = ceil(sqrt n), idx = i*side + j */
      u32 side = 1; while (side * side < n) side++;
      for (u32 i = 0; i < side; i++) for (u32 j = 0; j < side; j++) {
        u32 a = i * side + j; if (a >= n) continue;
        if (j + 1 < side && a + 1 < n) link2(s, a, a + 1);
        if (a + side < n) link2(s, a, a + side);
      }
    } break;
    case MSIM_TOPO_LINE: for (u32 i = 0; i + 1 < n; i++) link2(s, i, i + 1); break; /* :67-80 */
    case MSIM_TOPO_TOTAL: for (u32 i = 0; i < n; i++) for (u32 j = i + 1; j < n; j++) link2(s, i, j); break; /* :82-89 */
    default: { /* trees :91-167: BFS tiers 1,b,b^2.. => parent(i) = (i-1)/b */
      u32 b = s->cfg.topology == MSIM_TOPO_TREE2 ? 2 : s->cfg.topology == MSIM_TOPO_TREE3 ? 3 : 4;
      for (u32 i = 1; i < n; i++) link2(s, i, (i - 1) / b);
    }
  }
}

/* ---- nemesis: [upstream] jepsen.nemesis.combined partition-package, restated ------- */
static void complete_grudge(sim_t *s, const u8 *comp) { /* comp[node] = component id */
  for (u32 d = 0; d < s->N; d++) for (u32 x = 0; x < s->N; x++) if (comp[d] != comp[x]) setbit(s->part[d], x);
}
static void shuffle_nodes(sim_t *s, u32 j, u32 *perm) {
  for (u32 i = 0; i < s->N; i++) perm[i] = i;
  for (u32 i = s->N - 1; i >= 1; i--) {
    u32 k = scale32(draw32(s, S_NEM_SHUFFLE, ((u64)j << 16) | i), i + 1);
    u32 t = perm[i]; perm[i] = perm[k]; perm[k] = t;
  }
}
static void start_partition(sim_t *s, u32 j, u32 spec) {
  u32 n = s->N, perm[MAXN]; u8 comp[MAXN];
  memset(comp, 0, sizeof comp);
  switch (spec) {
    case MSIM_SPEC_ONE: comp[scale32(draw32(s, S_NEM_PICK, j), n)] = 1; complete_grudge(s, comp); break;
    case MSIM_SPEC_MAJORITY: /* bisect (shuffle nodes): split-at floor(n/2) */
      shuffle_nodes(s, j, perm);
      for (u32 i = 0; i < n / 2; i++) comp[perm[i]] = 1;
      complete_grudge(s, comp); break;
    case MSIM_SPEC_MINORITY_THIRD:
      shuffle_nodes(s, j, perm);
      for (u32 i = 0; i < (n - 1) / 3; i++) comp[perm[i]] = 1;
      complete_grudge(s, comp); break;
    default: { /* majorities-ring: node perm[(i+m/2)%n] sees only the window perm[i..i+m) */
      shuffle_nodes(s, j, perm);
      u32 m = n / 2 + 1;
      for (u32 i = 0; i < n; i++) {
        u32 c = perm[(i + m / 2) % n];
        u32 vis[MW] = {0, 0, 0, 0};
        for (u32 k = 0; k < m; k++) setbit(vis, perm[(i + k) % n]);
        for (u32 x = 0; x < n; x++) if (!bit(vis, x)) setbit(s->part[c], x);
      }
    }
  }
}

/* ---- node programs -------------------------------------------------------------------------------- */
static u32 *seen_of(sim_t *s, u32 node) { return s->seen + (size_t)node * s->W; }

/* read -> read_ok with the whole set (broadcast.rb:23-27, g_set.rb:13-15): snapshot into the payload */
static void node_read(sim_t *s, u32 node, const qent *q) {
  u32 words = (s->next_value + 31) / 32;
  u32 off = payload_alloc(s, words);
  if (off != INF) memcpy(s->payload + off, seen_of(s, node), words * 4);
  out_send(s, node, q->src, M_READ_OK, (off == INF ? 0 : off) | (words << 24), q->b);
}

static void gossip_targets(sim_t *s, u32 node, u32 src, u32 *tg) {
  u32 prog = s->cfg.node_program;
  for (u32 w = 0; w < MW; w++) tg[w] = 0;
  if (prog == MSIM_NODE_BCAST_RPC_ALL) { for (u32 i = 0; i < s->N; i++) if (i != node) setbit(tg, i); } /* broadcast.rb:37 other_node_ids */
  else for (u32 w = 0; w < MW; w++) tg[w] = s->adj[node][w];
  if (prog != MSIM_NODE_BCAST_FF_ECHOBACK && src < s->N) clrbit(tg, src); /* skip-sender, 02-performance.md:61-67 */
}

static u32 *unacked_of(sim_t *s, u32 node, u32 v) { return s->unacked + ((size_t)node * s->cfg.max_values + v) * MW; }

/* ack/retry tasks: one FIFO of (value, wake time) per node; unacked sets indexed by (node, value) */
static void task_push(sim_t *s, u32 node, u32 value, u32 wake) {
  tasks_t *t = &s->tasks[node];
  if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 16; t->v = (task_t *)realloc(t->v, t->cap * sizeof(task_t)); }
  t->v[t->n].value = value; t->v[t->n].next_retry = wake; t->n++;
}

static void node_broadcast(sim_t *s, u32 node, const qent *q) {
  u32 prog = s->cfg.node_program, v = q->a, has_id = q->b != 0;
  if (prog == MSIM_NODE_BCAST_ACK_RETRY && has_id) out_send(s, node, q->src, M_BROADCAST_OK, v, q->b); /* ack first, 02-performance.md:408-409 */
  u32 *sn = seen_of(s, node);
  if (!bit(sn, v)) {
    setbit(sn, v);
    u32 tg[MW]; gossip_targets(s, node, q->src, tg);
    int rpc = prog == MSIM_NODE_BCAST_ACK_RETRY || prog == MSIM_NODE_BCAST_RPC_ALL;
    int any = 0;
    for (u32 i = 0; i < s->N; i++) if (bit(tg, i)) { any = 1; out_send(s, node, i, M_BROADCAST, v, rpc ? ++s->node_msg_id[node] : 0); }
    if (prog == MSIM_NODE_BCAST_ACK_RETRY && any) { /* `until unacked.empty? ... sleep 1`, :421-438 */
      memcpy(unacked_of(s, node, v), tg, sizeof tg);
      task_push(s, node, v, s->T + 1000000u);
    }
  }
  if (prog != MSIM_NODE_BCAST_ACK_RETRY && has_id) out_send(s, node, q->src, M_BROADCAST_OK, v, q->b);
}

static void node_broadcast_ok(sim_t *s, u32 node, const qent *q) { /* callback: unacked.delete dest, :428-432 */
  if (s->cfg.node_program != MSIM_NODE_BCAST_ACK_RETRY) return;
  clrbit(unacked_of(s, node, q->a), q->src);
}

/* next timer of a node: g-set replicate tick, or the wake time of the oldest gossip task */
static u32 node_timer_time(const sim_t *s, u32 node) {
  u32 m = s->timer_next[node];
  const tasks_t *t = &s->tasks[node];
  if (t->head < t->n && t->v[t->head].next_retry < m) m = t->v[t->head].next_retry;
  return m;
}

/* stores a copy of W state words as snapshot (tick, node): what a replicate message refers to */
static void snap_put(sim_t *s, u32 tick, u32 node, const u32 *words) {
  u32 idx = tick * s->N + node;
  while (idx >= s->cap_snap) {
    u32 nc = s->cap_snap ? s->cap_snap * 2 : 64;
    s->snap = (u32 **)realloc(s->snap, nc * sizeof(u32 *));
    for (u32 i = s->cap_snap; i < nc; i++) s->snap[i] = NULL;
    s->cap_snap = nc;
  }
  u32 *cp = (u32 *)malloc(s->W * 4); memcpy(cp, words, s->W * 4);
  free(s->snap[idx]);
  s->snap[idx] = cp;
  if (idx + 1 > s->n_snap) s->n_snap = idx + 1;
}

static void hat_tick(sim_t *s, u32 node);
static void dt_timeout(sim_t *s, u32 node);
static void node_timer(sim_t *s, u32 node) {
  if (s->hat) { hat_tick(s, node); return; }
  if (s->dt) { dt_timeout(s, node); return; }
  if (s->timer_next[node] <= s->T) { /* g_set.rb:33-38: every 5 s, replicate_full to all other nodes */
    s->timer_next[node] = s->T + 5000000u;
    u32 tick = s->tick[node]++;           /* the message carries (sender, tick): a reference to the sender's set then */
    snap_put(s, tick, node, seen_of(s, node));
    for (u32 i = 0; i < s->N; i++) if (i != node) out_send(s, node, i, M_REPLICATE, tick, 0);
    return;
  }
  /* the gossip thread of the oldest task wakes: resend to whoever has not acked, sleep 1 s again;
   * if everyone acked the thread exits (02-performance.md:421-438) */
  tasks_t *t = &s->tasks[node];
  task_t k = t->v[t->head++];
  u32 *un = unacked_of(s, node, k.value), any = 0;
  for (u32 i = 0; i < s->N; i++) if (bit(un, i)) { any = 1; out_send(s, node, i, M_BROADCAST, k.value, ++s->node_msg_id[node]); }
  if (any) task_push(s, node, k.value, s->T + 1000000u);
}

#include "raft_nodes.inc"
#include "txn_nodes.inc"
#include "mk_nodes.inc"
#include "dt_nodes.inc"
#include "svc_nodes.inc"
#include "hat_nodes.inc"
#include "kafka_nodes.inc"

static void node_handle(sim_t *s, u32 node, const qent *q) {
  if (s->cfg.node_program == MSIM_NODE_RAFT) { raft_handle(s, node, q); return; }
  if (s->cfg.node_program == MSIM_NODE_TXN_SINGLE_KEY) { txn_node_handle(s, node, q); return; }
  if (s->cfg.node_program == MSIM_NODE_TXN_MULTI_KEY) { mk_node_handle(s, node, q); return; }
  if (s->cfg.node_program == MSIM_NODE_TXN_DATOMIC) { dt_node_handle(s, node, q); return; }
  if (s->cfg.node_program == MSIM_NODE_LIN_KV_PROXY) { px_node_handle(s, node, q); return; }
  if (s->cfg.node_program == MSIM_NODE_TSO_IDS) { tso_node_handle(s, node, q); return; }
  if (s->cfg.node_program == MSIM_NODE_TXN_RW_HAT) { hat_node_handle(s, node, q); return; }
  if (s->cfg.node_program == MSIM_NODE_KAFKA) { kf_node_handle(s, node, q); return; }
  switch (q->type) {
    case M_INIT: /* node.rb init handler -> init_ok; g-set starts its periodic task (node.rb:129-138) */
      if (s->cfg.node_program == MSIM_NODE_G_SET || s->cfg.node_program == MSIM_NODE_PN_COUNTER) s->timer_next[node] = s->T;
      out_send(s, node, q->src, M_INIT_OK, 0, q->b); break;
    case M_TOPOLOGY: s->nbr_known[node] = 1; out_send(s, node, q->src, M_TOPOLOGY_OK, 0, q->b); break;
    case M_ECHO: out_send(s, node, q->src, M_ECHO_OK, q->a, q->b); break; /* echo.rb:32-38 */
    case M_GENERATE: { /* flake_ids.clj:16-31: time = max(now in s, last time); count = same second ? count + 1 : 0 */
      u32 *f = s->flake + 2 * node, t = s->T / 1000000u;
      if (t < f[0]) t = f[0];
      f[1] = t == f[0] ? f[1] + 1 : 0; f[0] = t;
      out_send(s, node, q->src, M_GENERATE_OK, (t << 20) | ((f[1] & 0x7FFFu) << 5) | node, q->b); } break;
    case M_BROADCAST: node_broadcast(s, node, q); break;
    case M_BROADCAST_OK: node_broadcast_ok(s, node, q); break;
    case M_READ:
      if (s->cfg.node_program == MSIM_NODE_PN_COUNTER) { /* pn_counter.rb:69-71: increments minus decrements */
        u32 *st = seen_of(s, node), v = 0;
        for (u32 i = 0; i < s->N; i++) v += st[i] - st[s->N + i];
        out_send(s, node, q->src, M_READ_OK, v, q->b);
      } else node_read(s, node, q);
      break;
    case M_ADD:
      if (s->cfg.node_program == MSIM_NODE_PN_COUNTER) { /* pn_counter.rb:75-81: the node's own slot of the inc / dec G-counter */
        int d = (int)q->a;
        if (d >= 0) seen_of(s, node)[node] += (u32)d; else seen_of(s, node)[s->N + node] += (u32)(-d);
      } else setbit(seen_of(s, node), q->a); /* g_set.rb:17-21 */
      out_send(s, node, q->src, M_ADD_OK, q->a, q->b); break;
    case M_REPLICATE: { u32 *sn = seen_of(s, node), *v = s->snap[q->a * s->N + q->src];
      if (s->cfg.node_program == MSIM_NODE_PN_COUNTER) { for (u32 w = 0; w < s->W; w++) if (v[w] > sn[w]) sn[w] = v[w]; } /* pn_counter.rb:41-45 element-wise max */
      else for (u32 w = 0; w < s->W; w++) sn[w] |= v[w]; } break; /* g_set.rb:29-31 */
    default: break;
  }
}

/* ---- clients (client.clj) ------------------------------------------------------------------------- */
static int idempotent(const sim_t *s, u32 f) { /* with-errors sets: broadcast.clj:200 #{:read}; echo.clj:33 #{} */
  return (s->cfg.workload == MSIM_WL_BROADCAST || s->cfg.workload == MSIM_WL_LIN_KV) && f == MSIM_F_READ; /* lin_kv.clj:52 */
}

static void client_complete(sim_t *s, u32 slot, u32 type, u32 err, u32 value, u32 len) {
  struct cl *c = &s->cl[slot];
  c->busy = 0;
  if (c->kind != K_OP) { if (type != MSIM_T_OK) s->meta.flags |= MSIM_FLAG_ROUND_LIMIT; return; }
  if (s->defer_rows) { /* R4: completion rows of a round are written in slot order after the recv! loops */
    s->pend[slot].on = 1; s->pend[slot].type = type; s->pend[slot].f = c->f; s->pend[slot].err = err; s->pend[slot].final = c->m_final;
    s->pend[slot].process = c->process; s->pend[slot].value = value; s->pend[slot].len = len;
  } else add_row(s, type, c->f, err, c->m_final, c->process, value, len);
  if (type == MSIM_T_INFO) { /* crashed process: new process id, fresh client [upstream interpreter] */
    c->process += s->C;
    if (!reusable_clients(s)) { /* Reusable clients (lin_kv.clj:74-76) are not re-opened */
      c->next_msg_id = 0;
      s->inbox[s->N + slot].n = 0;
      if (s->kafka) kf_client_reopen(s, slot);
    }
  }
}

static void client_deliver(sim_t *s, u32 slot, const qent *q) {
  struct cl *c = &s->cl[slot];
  if (!c->busy || q->b != c->want) return; /* stale reply, client.clj:105-107 */
  if (s->kafka && c->kind == K_OP) { kf_client_deliver(s, slot, q); return; }
  switch (q->type) {
    case M_READ_OK:
      if (s->cfg.workload == MSIM_WL_LIN_KV) client_complete(s, slot, MSIM_T_OK, 0, (c->value & 0xFFu) | ((q->a & 0xFFu) << 8) | 0xFF0000u, 0); /* [k v], lin_kv.clj:56-61 */
      else if (s->cfg.workload == MSIM_WL_PN_COUNTER || s->cfg.workload == MSIM_WL_G_COUNTER) client_complete(s, slot, MSIM_T_OK, 0, q->a, 0); /* (long (:value ..)), pn_counter.clj:52-55 */
      else client_complete(s, slot, MSIM_T_OK, 0, q->a & 0xFFFFFFu, q->a >> 24);
      break;
    case M_ECHO_OK: case M_GENERATE_OK: client_complete(s, slot, MSIM_T_OK, 0, q->a, 0); break;
    case M_TXN_OK: client_complete(s, slot, MSIM_T_OK, 0, q->a & 0xFFFFFFu, q->a >> 24); break; /* txn_list_append.clj:109-117 */
    case M_ERROR: { /* client.clj:125-138 throw-errors!; every code the raft node emits is :definite? => :fail (errors.edn) */
      u32 err = q->a == 11 ? MSIM_ERR_TEMPORARILY_UNAVAILABLE : q->a == 20 ? MSIM_ERR_KEY_DOES_NOT_EXIST : q->a == 30 ? MSIM_ERR_TXN_CONFLICT : q->a == 14 ? MSIM_ERR_ABORT : MSIM_ERR_PRECONDITION_FAILED;
      if (txn_workload(s) && q->a == 0) client_complete(s, slot, MSIM_T_INFO, MSIM_ERR_TIMEOUT, c->value & 0xFFFFFFu, c->value >> 24); /* code 0 :timeout is not :definite? (errors.edn:2-4) */
      else if (txn_workload(s)) client_complete(s, slot, MSIM_T_FAIL, err, c->value & 0xFFFFFFu, c->value >> 24); /* :value stays the requested txn */
      else client_complete(s, slot, MSIM_T_FAIL, err, c->value, 0); } break;
    default: /* init_ok / topology_ok; for an operation of a transactional client (only after a failed init handshake — a flagged run — whose late init_ok meets the
              * first transaction's msg_id): completed :ok with the requested transaction, as the kernels do */
      if (c->kind == K_OP && c->f == MSIM_F_TXN) client_complete(s, slot, MSIM_T_OK, 0, c->value & 0xFFFFFFu, c->value >> 24);
      else client_complete(s, slot, MSIM_T_OK, 0, c->value, 0);
      break;
  }
}

static void client_timeout(sim_t *s, u32 slot) { /* client.clj:96-103 + :158-162 */
  struct cl *c = &s->cl[slot];
  if (s->kafka && c->kind == K_OP) { kf_client_timeout(s, slot); return; }
  u32 type = idempotent(s, c->f) ? MSIM_T_FAIL : MSIM_T_INFO;
  u32 v = c->f == MSIM_F_READ && s->cfg.workload != MSIM_WL_LIN_KV ? MSIM_NO_VALUE : c->value;
  if (c->f == MSIM_F_TXN) client_complete(s, slot, type, MSIM_ERR_NET_TIMEOUT, v & 0xFFFFFFu, v >> 24);
  else client_complete(s, slot, type, MSIM_ERR_NET_TIMEOUT, v, 0);
}

static void client_invoke(sim_t *s, u32 slot) {
  struct cl *c = &s->cl[slot];
  c->mark = 0; c->busy = 1;
  u32 ep = s->N + slot, dest, type, a = 0;
  if (c->kind == K_INIT) { dest = slot; type = M_INIT; c->next_msg_id = 0; }
  else if (c->kind == K_TOPO) { dest = slot; type = M_TOPOLOGY; c->next_msg_id = 0; }
  else if (s->kafka) { /* one or two RPCs per operation, or none (kafka_nodes.inc) */
    if (s->kafka->cl[slot].stage != 2) { c->f = c->m_f; c->value = c->m_value; }
    dest = c->process % s->N;
    if (!kf_client_invoke(s, slot, &type, &a)) return;
  }
  else {
    c->f = c->m_f; c->value = c->m_value;
    dest = c->process % s->N; /* worker -> node: nodes[process mod n] [upstream] */
    if (c->f == MSIM_F_TXN) add_row(s, MSIM_T_INVOKE, c->f, 0, 0, c->process, c->value & 0xFFFFFFu, c->value >> 24);
    else add_row(s, MSIM_T_INVOKE, c->f, 0, c->m_final, c->process, c->value, 0);
    switch (c->f) {
      case MSIM_F_ECHO: type = M_ECHO; a = c->value; break;
      case MSIM_F_BROADCAST: type = M_BROADCAST; a = c->value; break;
      case MSIM_F_ADD: type = M_ADD; a = c->value; break;
      case MSIM_F_WRITE: type = M_WRITE; a = c->value; break;
      case MSIM_F_CAS: type = M_CAS; a = c->value; break;
      case MSIM_F_TXN: type = M_TXN; a = c->value; break;
      case MSIM_F_GENERATE: type = M_GENERATE; a = 0; break;
      default: type = M_READ; a = s->cfg.workload == MSIM_WL_LIN_KV ? c->value : 0; break;
    }
  }
  c->want = ++c->next_msg_id; /* client.clj:61-64 */
  u32 to_ms = s->cfg.client_timeout_ms;
  if (s->cfg.workload == MSIM_WL_LIN_KV) { to_ms = 10 * s->cfg.latency_mean_ms; if (to_ms < 1000) to_ms = 1000; } /* lin_kv.clj:54 */
  c->timeout_at = s->T + (c->kind == K_OP ? to_ms : 10000u) * 1000u; /* db.clj:54 */
  out_send(s, ep, dest, type, a, c->want);
}

/* ---- scheduler: [upstream] generator interpreter for core.clj:67-80 ------------------------------- */
static u32 stagger_us(const sim_t *s, u32 stream, u32 k, u64 period_us) { /* uniform [0, 2*period) */
  return (u32)(((u64)draw32(s, stream, k) * (2 * period_us)) >> 32);
}
static int any_busy(const sim_t *s, u32 n) { for (u32 i = 0; i < n; i++) if (s->cl[i].busy) return 1; return 0; }
static int counter_workload(const sim_t *s) { return s->cfg.workload == MSIM_WL_PN_COUNTER || s->cfg.workload == MSIM_WL_G_COUNTER; }
static int has_final(const sim_t *s) { return s->cfg.workload == MSIM_WL_BROADCAST || s->cfg.workload == MSIM_WL_G_SET || counter_workload(s) || s->cfg.workload == MSIM_WL_KAFKA; }
static int nem_on(const sim_t *s) { return (s->cfg.nemesis_mask & MSIM_NEMESIS_PARTITION) != 0; }
static int gen_live(const sim_t *s) { return s->cfg.rate_mhz > 0 && s->gen_next < s->cutoff; }
static int nem_live(const sim_t *s) { return nem_on(s) && s->nem_next < s->cutoff; }

/* time-free phase transitions; evaluated at the top of every round */
static void sched_resolve(sim_t *s) {
  for (;;) {
    switch (s->phase) {
      case PH_INIT_WAIT: if (any_busy(s, s->CS)) return; s->phase = s->cfg.workload == MSIM_WL_BROADCAST ? PH_TOPO : PH_MAIN_START; continue;
      case PH_TOPO_WAIT: if (any_busy(s, s->CS)) return; s->phase = PH_MAIN_START; continue;
      case PH_MAIN_START:
        s->cutoff = s->T + s->cfg.time_limit_ms * 1000u; s->gen_next = s->T; s->nem_next = s->T;
        for (u32 i = 0; i < s->CS; i++) s->cl[i].next_msg_id = 0; /* workers open fresh clients (client.clj:41-53) */
        s->loss_on = 1; /* p-loss is a run-time fault (net.clj:121-122), never active during db setup */
        s->phase = PH_MAIN; continue;
      case PH_MAIN:
        if (gen_live(s) || nem_live(s)) return;
        if (s->cfg.rate_mhz == 0 && s->T < s->cutoff) return; /* (gen/sleep time-limit), core.clj:69 */
        s->phase = PH_DRAIN; continue;
      case PH_DRAIN:
        if (any_busy(s, s->C)) return;
        s->phase = nem_on(s) && has_final(s) ? PH_NEM_FINAL : has_final(s) ? PH_SLEEP : PH_DONE;
        if (s->phase == PH_SLEEP) s->sleep_until = s->T + s->cfg.quiesce_ms * 1000u;
        continue;
      case PH_FINAL_WAIT: if (any_busy(s, s->C)) return; s->phase = PH_DONE; continue;
      case PH_KF_FINAL: /* every worker has polled until nothing came (or the 10 s of workload/kafka.clj:305-306 are over) */
        for (u32 i = 0; i < s->C; i++) if (s->cl[i].busy || (!s->kafka->cl[i].done && s->T < s->kafka->final_deadline)) return;
        s->phase = PH_DONE; continue;
      default: return;
    }
  }
}

/* time at which the scheduler wants to act next (>= T), INF if it only waits for completions */
static u32 sched_due(const sim_t *s) {
  u32 T = s->T, d = INF;
  switch (s->phase) {
    case PH_INIT: case PH_TOPO: case PH_NEM_FINAL: case PH_FINAL: return T;
    case PH_SLEEP: return s->sleep_until;
    case PH_KF_FINAL:
      if (T < s->kafka->final_deadline) for (u32 i = 0; i < s->C; i++) if (!s->cl[i].busy && !s->kafka->cl[i].done) return T;
      return INF;
    case PH_MAIN:
      if (nem_live(s)) d = s->nem_next > T ? s->nem_next : T;
      if (gen_live(s)) { int fr = 0; for (u32 i = 0; i < s->C; i++) fr |= !s->cl[i].busy;
        if (fr) { u32 g = s->gen_next > T ? s->gen_next : T; if (g < d) d = g; } }
      if (s->cfg.rate_mhz == 0 && !nem_live(s) && s->cutoff < d) d = s->cutoff;
      return d;
    default: return INF;
  }
}

static void nemesis_rows(sim_t *s, u32 f, u32 v1, u32 v2, u32 len2) {
  add_row(s, MSIM_T_INFO, f, 0, 0, MSIM_PROCESS_NEMESIS, v1, 0);
  add_row(s, MSIM_T_INFO, f, 0, 0, MSIM_PROCESS_NEMESIS, v2, len2);
}
static void heal(sim_t *s) { memset(s->part, 0, sizeof s->part); } /* net.clj:112-113 */

static void sched_act(sim_t *s) {
  u32 T = s->T;
  u64 period = 1000000000ull / (s->cfg.rate_mhz ? s->cfg.rate_mhz : 1);
  switch (s->phase) {
    case PH_INIT: for (u32 i = 0; i < s->N; i++) { s->cl[i].mark = 1; s->cl[i].kind = K_INIT; } s->phase = PH_INIT_WAIT; break;
    case PH_TOPO: for (u32 i = 0; i < s->N; i++) { s->cl[i].mark = 1; s->cl[i].kind = K_TOPO; } s->phase = PH_TOPO_WAIT; break;
    case PH_MAIN:
      if (nem_live(s) && s->nem_next <= T) { /* flip-flop start/stop, staggered by the interval */
        u32 j = s->nem_j++;
        if ((j & 1) == 0) {
          u32 spec = scale32(draw32(s, S_NEM_SPEC, j), 4);
          start_partition(s, j, spec);
          u32 words = s->N * MW, off = payload_alloc(s, words);
          if (off != INF) memcpy(s->payload + off, s->part, words * 4);
          nemesis_rows(s, MSIM_F_START_PARTITION, spec, off == INF ? 0 : off, words);
        } else { heal(s); nemesis_rows(s, MSIM_F_STOP_PARTITION, MSIM_NO_VALUE, MSIM_NO_VALUE, 0); }
        s->nem_next = T + stagger_us(s, S_NEM_STAGGER, j, (u64)s->cfg.nemesis_interval_ms * 1000u);
      }
      if (gen_live(s) && s->gen_next <= T) {
        u32 nfree = 0; for (u32 i = 0; i < s->C; i++) nfree += !s->cl[i].busy;
        if (nfree) {
          /* one 64-bit draw per generated op k: high word -> stagger delay, low word -> process pick
           * (top bits), gen/mix choice (bit 0), echo payload (bits 4..10) */
          u32 k = s->gen_k++;
          u64 h = draw64(s, S_GEN, k);
          u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          u32 pick = scale32(r_lo, nfree), slot = 0;
          for (u32 i = 0; i < s->C; i++) if (!s->cl[i].busy) { if (pick == 0) { slot = i; break; } pick--; }
          struct cl *c = &s->cl[slot];
          c->mark = 1; c->kind = K_OP; c->m_final = 0;
          if (s->cfg.workload == MSIM_WL_LIN_KV) {
            /* [upstream] jepsen.tests.linearizable-register: one key at a time per group of 2n threads; the
             * first n threads read, the others mix [w cas cas]; values 0..4; (gen/process-limit 20) retires a key
             * once 20 distinct processes have used it */
            if (s->key_reg[slot] != 1 + c->process) {
              if (s->key_procs == 20) {
                /* keys travel in 8 bits of the op's value: a 257th key would alias the first (and the checker would merge their histories) */
                if (s->cur_key >= 255) { s->meta.flags |= MSIM_FLAG_VALUES_OVERFLOW; c->mark = 0; s->phase = PH_DONE; return; }
                s->cur_key++; s->key_procs = 0; memset(s->key_reg, 0, s->CS * 4);
              }
              s->key_reg[slot] = 1 + c->process; s->key_procs++;
            }
            u64 h2 = draw64(s, S_GEN2, k);
            u32 v1 = scale32((u32)(h2 >> 32), 5), v2 = (((u32)(h2 >> 20) & 0xFFFu) * 5u) >> 12, key = s->cur_key & 0xFF;
            if (slot < s->N) { c->m_f = MSIM_F_READ; c->m_value = key | 0xFFFF00u; }
            else if (scale32((u32)h2, 3) == 0) { c->m_f = MSIM_F_WRITE; c->m_value = key | (v1 << 8) | 0xFF0000u; }
            else { c->m_f = MSIM_F_CAS; c->m_value = key | (v1 << 8) | (v2 << 16); }
          }
          else if (s->kafka) {
            if (!kf_generate(s, k, c)) { c->mark = 0; s->phase = PH_DONE; return; }
          }
          else if (txn_workload(s)) {
            u32 ref = txn_generate(s, k);
            if (ref == INF) { c->mark = 0; s->phase = PH_DONE; return; }
            c->m_f = MSIM_F_TXN; c->m_value = ref;
          }
          else if (s->cfg.workload == MSIM_WL_UNIQUE_IDS) { c->m_f = MSIM_F_GENERATE; c->m_value = MSIM_NO_VALUE; } /* (gen/repeat {:f :generate}) */
          else if (s->cfg.workload == MSIM_WL_ECHO) { c->m_f = MSIM_F_ECHO; c->m_value = (r_lo >> 4) & 127; } /* echo.clj:72-75 */
          else if (s->cfg.workload == MSIM_WL_G_COUNTER) { /* g_counter.clj:37-41: (gen/filter ...) skips negative adds, takes the next op of the mix at once */
            u32 rr = r_lo, a = 0;
            int d = (int)((((rr >> 4) & 0xFFFFu) * 10u) >> 16) - 5;
            while (!(rr & 1) && d < 0 && a < 15) { a++; rr = (u32)draw64(s, S_GEN2, (u64)k * 16 + a); d = (int)((((rr >> 4) & 0xFFFFu) * 10u) >> 16) - 5; }
            if ((rr & 1) || d < 0) { c->m_f = MSIM_F_READ; c->m_value = MSIM_NO_VALUE; } else { c->m_f = MSIM_F_ADD; c->m_value = (u32)d; }
          }
          else if (r_lo & 1) { c->m_f = MSIM_F_READ; c->m_value = MSIM_NO_VALUE; }  /* gen/mix */
          else if (s->cfg.workload == MSIM_WL_PN_COUNTER) { /* {:f :add, :value (- (rand-int 10) 5)}, pn_counter.clj:134-135 */
            c->m_f = MSIM_F_ADD; c->m_value = (u32)((int)((((r_lo >> 4) & 0xFFFFu) * 10u) >> 16) - 5);
          }
          else {
            c->m_f = s->cfg.workload == MSIM_WL_BROADCAST ? MSIM_F_BROADCAST : MSIM_F_ADD;
            if (s->next_value >= s->cfg.max_values) { s->meta.flags |= MSIM_FLAG_VALUES_OVERFLOW; c->mark = 0; s->phase = PH_DONE; return; }
            c->m_value = s->next_value++;
          }
          s->gen_next = T + (u32)(((u64)r_hi * (2 * period)) >> 32); /* gen/stagger (/ rate), core.clj:68 */
        }
      }
      break;
    case PH_NEM_FINAL: heal(s); nemesis_rows(s, MSIM_F_STOP_PARTITION, MSIM_NO_VALUE, MSIM_NO_VALUE, 0);
      s->phase = PH_SLEEP; s->sleep_until = T + s->cfg.quiesce_ms * 1000u; break;
    case PH_SLEEP: if (T >= s->sleep_until) s->phase = PH_FINAL; else break; /* fallthrough */
    case PH_FINAL: /* (gen/clients (gen/each-thread {:f :read [:final? true]})), broadcast.clj:240, g_set.clj:61 */
      if (s->kafka) { /* [upstream] the final generator of jepsen.tests.kafka, clipped to 10 s (workload/kafka.clj:305-306) */
        for (u32 i = 0; i < s->C; i++) { struct cl *c = &s->cl[i]; c->mark = 1; c->kind = K_OP; c->m_final = 0; s->kafka->cl[i].fin = 1;
          if (!kf_final_assign(s, c)) { c->mark = 0; s->phase = PH_DONE; return; } }
        s->kafka->final_deadline = T + 10000000u; s->phase = PH_KF_FINAL; break;
      }
      for (u32 i = 0; i < s->C; i++) { struct cl *c = &s->cl[i]; c->mark = 1; c->kind = K_OP; c->m_f = MSIM_F_READ; c->m_value = MSIM_NO_VALUE;
        c->m_final = s->cfg.workload == MSIM_WL_BROADCAST || s->cfg.workload == MSIM_WL_PN_COUNTER || s->cfg.workload == MSIM_WL_G_COUNTER; } /* pn_counter.clj:137 */
      s->phase = PH_FINAL_WAIT; break;
    case PH_KF_FINAL:
      for (u32 i = 0; i < s->C; i++) { struct cl *c = &s->cl[i];
        if (!c->busy && !s->kafka->cl[i].done && T < s->kafka->final_deadline) { c->mark = 1; c->kind = K_OP; c->m_final = 0; c->m_f = MSIM_F_POLL; c->m_value = MSIM_NO_VALUE; } }
      break;
    default: break;
  }
}

/* ---- one instance --------------------------------------------------------------------------------- */

static void rext_free(struct rext *r) { if (r) { free(r->ents); free(r); } }

/* commit the round's staged sends in canonical order (net.clj:189-221) */
static void commit_sends(sim_t *s) {
  u32 T = s->T;
  for (u32 i = 0; i < s->n_out; i++) {
    outmsg *m = &s->out[i];
    u32 id = s->next_msg_id++;
    int cl = is_client(s, m->src_ep) || is_client(s, m->dest_ep);
    s->st.all_send++; if (cl) s->st.clients_send++; else s->st.servers_send++; /* journal :send before loss */
    jlog(s, 0, id, m->type, m->a, m->b, m->src_ep, m->dest_ep);
    u32 lat = latency_ms(s, id, cl);
    if (s->loss_on && s->cfg.p_loss_q32 && draw32(s, S_LOSS, id) < s->cfg.p_loss_q32) { rext_free(m->r); continue; } /* net.clj:214 */
    qent q = {T + lat * 1000u, id, m->a, m->b, m->src_ep, m->type, m->r};
    inbox_push(s, m->dest_ep, q);
  }
  s->n_out = 0;
}

/* an idle receiver polls: take the min-(deadline,id) envelope even if not due (net.clj:228-229), drop it
 * if the partition says so (:234), else sleep floor(dt) ms (:236-238). */
static void poll_endpoint(sim_t *s, u32 e) {
  u32 N = s->N, T = s->T;
  if (is_client(s, e) && !s->cl[e - N].busy) return; /* clients only poll inside recv! (client.clj:94-95) */
  inbox_t *b = &s->inbox[e];
  while (!s->has_committed[e] && b->n) {
    u32 k = 0;
    for (u32 i = 1; i < b->n; i++)
      if (b->v[i].deadline < b->v[k].deadline || (b->v[i].deadline == b->v[k].deadline && b->v[i].id < b->v[k].id)) k = i;
    qent q = b->v[k]; b->v[k] = b->v[--b->n];
    if (e < N && q.src < N && bit(s->part[e], q.src)) { rext_free(q.r); continue; } /* partitioned: dropped, no :recv */
    s->committed[e] = q; s->has_committed[e] = 1;
    s->deliver_at[e] = q.deadline <= T ? T : T + ((q.deadline - T) / 1000u) * 1000u;
  }
}

static void rtrace(sim_t *s, u32 node, u32 what) { /* {time, node, message id or 0xFFFFFFFF = an action of the main loop} */
  if (!s->rtrace) return;
  if (s->n_rtrace < s->cap_rtrace) { u32 *e = s->rtrace + 3 * (size_t)s->n_rtrace; e[0] = s->T; e[1] = node; e[2] = what; }
  s->n_rtrace++;
}

static void run_instance(sim_t *s) {
  u32 N = s->N, E = s->E;
  for (;;) {
    sched_resolve(s);
    if (s->phase == PH_DONE) break;
    if (++s->rounds > 50000000u) { s->meta.flags |= MSIM_FLAG_ROUND_LIMIT; break; }
    if (s->dt && (s->meta.flags & MSIM_FLAG_ARENA_OVERRUN)) break; /* an engine capacity was exceeded: what follows would not be the program's behaviour */

    /* R0: next event time.  Deliveries, node timers and the scheduler are "normal" events; client
     * timeouts only fire in a round where nothing else is due. */
    u32 tn = sched_due(s), tt = INF;
    for (u32 e = 0; e < E; e++) if (s->has_committed[e] && s->deliver_at[e] < tn) tn = s->deliver_at[e];
    for (u32 n = 0; n < N; n++) { u32 t = s->raft ? raft_next_time(s, n) : node_timer_time(s, n); if (t < tn) tn = t; }
    for (u32 c = 0; c < s->CS; c++) if (s->cl[c].busy && s->cl[c].timeout_at < tt) tt = s->cl[c].timeout_at;
    for (u32 c = 0; c < s->CS; c++) if (s->cl[c].mark) tn = s->T; /* a client between the two RPCs of one operation (kafka poll -> commit_offsets) sends the second at once */
    if (tn == INF && tt == INF) { s->meta.flags |= MSIM_FLAG_ROUND_LIMIT; break; } /* stuck */
    int timeout_round = tt < tn;
    u32 T = timeout_round ? tt : tn;
    if (T < s->T) T = s->T;
    s->T = T;
    s->n_out = 0;

    if (timeout_round) {
      for (u32 c = 0; c < s->CS; c++) if (s->cl[c].busy && s->cl[c].timeout_at <= T) client_timeout(s, c);
      continue;
    }
    /* R1: scheduler (generator interpreter + nemesis) */
    if (sched_due(s) <= T) sched_act(s);
    if (s->phase == PH_DONE) break;
    /* R2: marked clients invoke (slot order); their requests are committed and idle receivers poll */
    if (s->kafka) s->defer_rows = 1; /* an operation that needs no RPC (:crash, :assign with :seek-to-beginning?) completes here: its row follows the round's invocations */
    for (u32 c = 0; c < s->CS; c++) if (s->cl[c].mark) client_invoke(s, c);
    s->defer_rows = 0;
    commit_sends(s);
    for (u32 e = 0; e < E; e++) poll_endpoint(s, e);
    /* R3: one input per node (node order): a due timer, else the due committed message */
    for (u32 n = 0; n < N; n++) {
      int msg_due = s->has_committed[n] && s->deliver_at[n] <= T;
      if (s->raft ? !msg_due : node_timer_time(s, n) <= T) {  /* raft.py's loop takes a message first (raft.py:577-585) */
        if (!s->raft) node_timer(s, n);
        else if (raft_next_time(s, n) <= T) { rtrace(s, n, 0xFFFFFFFFu); raft_act(s, n); }
      } else if (msg_due) {
        qent q = s->committed[n]; s->has_committed[n] = 0;
        s->st.all_recv++; if (is_client(s, q.src)) s->st.clients_recv++; else s->st.servers_recv++; /* journal :recv */
        jlog(s, 1, q.id, q.type, q.a, q.b, q.src, n);
        if (s->raft) rtrace(s, n, q.id);
        node_handle(s, n, &q);
        rext_free(q.r);
      }
    }
    for (u32 e = N + s->CS; e < E; e++) /* services run after the nodes (endpoint order): one request per round (service.clj:252-258) */
      if (s->has_committed[e] && s->deliver_at[e] <= T) {
        qent q = s->committed[e]; s->has_committed[e] = 0;
        s->st.all_recv++; s->st.servers_recv++;
        jlog(s, 1, q.id, q.type, q.a, q.b, q.src, e);
        if (s->kafka) kf_svc_handle(s, &q); else if (s->mk) mk_svc_handle(s, e, &q); else if (s->dt) dt_svc_handle(s, e, &q); else if (s->cfg.node_program == MSIM_NODE_TSO_IDS) tso_svc_handle(s, &q); else if (s->svc) px_svc_handle(s, &q); else svc_handle(s, &q);
      }
    commit_sends(s);
    for (u32 e = 0; e < E; e++) poll_endpoint(s, e);
    /* R4: clients run their recv! loops: consume due envelopes until the awaited reply arrives; stale replies
     * are skipped (client.clj:94-107).  Envelope k of every client is handled before envelope k+1 of any
     * (journal order); the completion rows of the round are then written in slot order. */
    s->defer_rows = 1;
    for (int any = 1; any;) {
      any = 0;
      for (u32 c = 0; c < s->CS; c++) {
        u32 e = N + c;
        if (s->has_committed[e] && s->deliver_at[e] <= T) {
          qent q = s->committed[e]; s->has_committed[e] = 0;
          s->st.all_recv++; s->st.clients_recv++;
          jlog(s, 1, q.id, q.type, q.a, q.b, q.src, e);
          client_deliver(s, c, &q);
          poll_endpoint(s, e);
          any = 1;
        }
      }
    }
    s->defer_rows = 0;
    for (u32 c = 0; c < s->CS; c++) if (s->pend[c].on) {
      s->pend[c].on = 0;
      add_row(s, s->pend[c].type, s->pend[c].f, s->pend[c].err, s->pend[c].final, s->pend[c].process, s->pend[c].value, s->pend[c].len);
    }
  }
  s->meta.n_rounds = s->rounds;
}

/* ---- public entry points (loaded by tests via ctypes) --------------------------------------------- */

static sim_t *sim_new(const msim_config *cfg, uint64_t instance, msim_op *rows, uint32_t *payload) {
  if (cfg->n_nodes == 0 || cfg->n_nodes > MAXN || cfg->concurrency == 0) return NULL;
  sim_t *s = (sim_t *)calloc(1, sizeof(sim_t));
  s->cfg = *cfg;
  s->N = cfg->n_nodes; s->C = cfg->concurrency; s->CS = s->C > s->N ? s->C : s->N;
  s->S = cfg->node_program == MSIM_NODE_KAFKA || cfg->node_program == MSIM_NODE_TXN_SINGLE_KEY || cfg->node_program == MSIM_NODE_LIN_KV_PROXY || cfg->node_program == MSIM_NODE_TSO_IDS ? 1 : cfg->node_program == MSIM_NODE_TXN_MULTI_KEY || cfg->node_program == MSIM_NODE_TXN_DATOMIC ? 2 : 0; /* lin-kv (+ lww-kv) */
  s->E = s->N + s->CS + s->S;
  if (s->E > 255) { free(s); return NULL; }
  s->W = (cfg->max_values + 31) / 32;
  s->key = inst_key(cfg->seed, instance);
  s->next_msg_id = 0; /* net.clj:103,197: counter starts at -1 and is pre-incremented: first id 0 */
  build_topology(s);
  s->inbox = (inbox_t *)calloc(s->E, sizeof(inbox_t));
  s->committed = (qent *)calloc(s->E, sizeof(qent));
  s->has_committed = (u8 *)calloc(s->E, 1);
  s->deliver_at = (u32 *)calloc(s->E, 4);
  s->seen = (u32 *)calloc((size_t)s->N * s->W + 1, 4);
  s->node_msg_id = (u32 *)calloc(s->N, 4);
  s->nbr_known = (u32 *)calloc(s->N, 4);
  s->tasks = (tasks_t *)calloc(s->N, sizeof(tasks_t));
  if (cfg->node_program == MSIM_NODE_BCAST_ACK_RETRY) s->unacked = (u32 *)calloc((size_t)s->N * cfg->max_values * MW, 4);
  s->timer_next = (u32 *)malloc(s->N * 4);
  s->tick = (u32 *)calloc(s->N, 4);
  s->flake = (u32 *)calloc(2 * s->N, 4);
  for (u32 i = 0; i < s->N; i++) s->timer_next[i] = INF;
  s->cl = (struct cl *)calloc(s->CS, sizeof(struct cl));
  s->pend = calloc(s->CS, sizeof(*s->pend));
  s->key_reg = (u32 *)calloc(s->CS, 4);
  if (cfg->node_program == MSIM_NODE_RAFT) {
    s->raft = (rnode *)calloc(s->N, sizeof(rnode));
    for (u32 i = 0; i < s->N; i++) { rnode *r = &s->raft[i]; r->voted_for = -1; r->leader = -1; r->last_applied = 1; memset(r->kv, 0xFF, sizeof r->kv);
      rentry e0; memset(&e0, 0, sizeof e0); r_append(r, &e0, 1); }
  }
  if (cfg->node_program == MSIM_NODE_LIN_KV_PROXY || cfg->node_program == MSIM_NODE_TSO_IDS) {
    svc_t *v = (svc_t *)calloc(1, sizeof(svc_t));
    v->cb = (pslot *)calloc((size_t)s->N * PX_SLOTS, sizeof(pslot));
    v->client_idx = (u32 *)calloc(s->E, 4);
    memset(v->kv, 0xFF, 256); memset(v->ring, 0xFF, sizeof v->ring); memset(v->rep, 0xFF, sizeof v->rep);
    s->svc = v;
  } else if (cfg->node_program == MSIM_NODE_KAFKA) { s->kafka = kf_new(s);
  } else if (s->S || cfg->node_program == MSIM_NODE_TXN_RW_HAT) { /* (the multi-key node too: generator state + the elements' versions) */
    txn_t *t = (txn_t *)calloc(1, sizeof(txn_t)); /* the rw-register workload only uses the generator state */
    t->slots = (tslot *)calloc((size_t)s->N * TXN_SLOTS, sizeof(tslot));
    t->slots_cap = s->cfg.concurrency > s->N ? TXN_SLOTS : 8u;
    t->root = V_NIL;
    t->kv = (u32 *)calloc((size_t)cfg->max_values * cfg->max_writes_per_key, 4);
    t->kv_n = (u8 *)calloc(cfg->max_values, 1);
    for (u32 i = 0; i < cfg->key_count && i < 16; i++) { t->active[i] = i; t->next_val[i] = 1; }
    t->next_key = cfg->key_count;
    s->txn = t;
  }
  if (cfg->node_program == MSIM_NODE_TXN_RW_HAT) s->hat = hat_new(s);
  if (cfg->node_program == MSIM_NODE_TXN_MULTI_KEY) s->mk = mk_new(s);
  if (cfg->node_program == MSIM_NODE_TXN_DATOMIC) s->dt = dt_new(s);
  for (u32 i = 0; i < s->CS; i++) s->cl[i].process = i;
  s->rows = rows; s->payload = payload;
  s->phase = PH_INIT;
  return s;
}

static void sim_free(sim_t *s) {
  for (u32 e = 0; e < s->E; e++) {
    for (u32 i = 0; i < s->inbox[e].n; i++) rext_free(s->inbox[e].v[i].r);
    if (s->has_committed[e]) rext_free(s->committed[e].r);
    free(s->inbox[e].v);
  }
  for (u32 i = 0; i < s->n_out; i++) rext_free(s->out[i].r);
  for (u32 n = 0; n < s->N; n++) free(s->tasks[n].v);
  for (u32 i = 0; i < s->n_snap; i++) free(s->snap[i]);
  if (s->raft) { for (u32 i = 0; i < s->N; i++) free(s->raft[i].log); free(s->raft); }
  if (s->svc) { free(s->svc->cb); free(s->svc->client_idx); free(s->svc); }
  if (s->hat) hat_free(s->hat);
  mk_free(s->mk, s->N);
  dt_free(s->dt, s->N);
  kf_free(s->kafka);
  if (s->txn) { free(s->txn->slots); free(s->txn->kv); free(s->txn->kv_n); free(s->txn); }
  free(s->snap); free(s->inbox); free(s->committed); free(s->has_committed); free(s->deliver_at); free(s->seen);
  free(s->unacked); free(s->node_msg_id); free(s->nbr_known); free(s->tasks); free(s->key_reg); free(s->timer_next); free(s->tick); free(s->flake);
  free(s->cl); free(s->pend); free(s->out);
  free(s);
}

/* Simulates global instance `instance` of `cfg` (already finalized: capacities non-zero).
 * rows: max_rows entries; payload: max_payload_words words.  Returns 0, or -1 on a bad config. */
int oracle_run_instance(const msim_config *cfg, uint64_t instance, msim_op *rows, uint32_t *payload,
                        msim_net_stats *stats, msim_inst_meta *meta, msim_event *journal) {
  sim_t *s = sim_new(cfg, instance, rows, payload);
  if (!s) return -1;
  if (cfg->journal_capacity && !journal) { sim_free(s); return -1; }
  s->journal = journal;
  run_instance(s);
  *stats = s->st; *meta = s->meta;
  sim_free(s);
  return 0;
}

/* Test hook: the state-transition function of ONE node, outside the network.  Feeds `n_in` inputs
 * {src endpoint, message type (oracle_msg_type), a, b} to node `node` in order and records every message
 * the node emits as {input index, dest endpoint, type, a, b}.  An input of type 0 means "let `a`
 * microseconds pass and run every timer that becomes due" (g-set replicate tick, gossip retries).
 * Used to replay golden vectors recorded from the reference's own node programs
 * (tests/golden/make_golden.py).  Returns the number of emitted messages, or -1. */
int oracle_node_trace(const msim_config *cfg, uint32_t node, const uint32_t *in, uint32_t n_in,
                      uint32_t *out, uint32_t out_cap, uint32_t *payload, uint32_t *final_set) {
  msim_op *rows = (msim_op *)calloc(cfg->max_rows, sizeof(msim_op));
  sim_t *s = sim_new(cfg, 0, rows, payload);
  if (!s || node >= s->N) { free(rows); return -1; }
  u32 n = 0;
  u32 *stg = (u32 *)calloc(s->W + 1, 4); /* staging for a peer's replicate payload: inputs of type 0xFE set word a := b */
  for (u32 i = 0; i < n_in; i++) {
    const u32 *m = in + 4 * i;
    s->n_out = 0;
    if (m[1] == 0) {
      s->T += m[2];
      while (node_timer_time(s, node) <= s->T) node_timer(s, node);
    } else if (m[1] == 0xFE) {
      if (m[2] < s->W) stg[m[2]] = m[3];
    } else if (m[1] == M_REPLICATE && m[0] < s->N && m[0] != node) { /* a peer's replicate: its payload is the staged words */
      u32 tick = s->tick[m[0]]++;
      snap_put(s, tick, m[0], stg);
      memset(stg, 0, s->W * 4);
      qent q = {s->T, i, tick, 0, (u8)m[0], (u8)M_REPLICATE, NULL};
      node_handle(s, node, &q);
    } else {
      qent q = {s->T, i, m[2], m[3], (u8)m[0], (u8)m[1]};
      if (q.type == M_BROADCAST || q.type == M_ADD) { if (q.a + 1 > s->next_value) s->next_value = q.a + 1; }
      node_handle(s, node, &q);
    }
    for (u32 k = 0; k < s->n_out && n < out_cap; k++, n++) {
      u32 *o = out + 5 * n;
      o[0] = i; o[1] = s->out[k].dest_ep; o[2] = s->out[k].type; o[3] = s->out[k].a; o[4] = s->out[k].b;
      if (s->out[k].type == M_REPLICATE && final_set) memcpy(final_set, s->snap[s->out[k].a * s->N + node], s->W * 4); /* last replicated value */
    }
  }
  if (final_set && cfg->node_program != MSIM_NODE_G_SET && cfg->node_program != MSIM_NODE_PN_COUNTER) memcpy(final_set, seen_of(s, node), s->W * 4);
  free(stg);
  sim_free(s); free(rows);
  return (int)n;
}

uint32_t oracle_msg_type(const char *name) {
  static const char *names[] = {"", "init", "init_ok", "topology", "topology_ok", "echo", "echo_ok", "broadcast", "broadcast_ok",
                                "read", "read_ok", "add", "add_ok", "replicate", "write", "write_ok", "cas", "cas_ok", "error",
                                "request_vote", "request_vote_res", "append_entries", "append_entries_res", "txn", "txn_ok", "generate", "generate_ok", "replicate_ack"};
  for (u32 i = 1; i < sizeof(names) / sizeof(names[0]); i++) if (!strcmp(names[i], name)) return i;
  return 0;
}

/* Test hook for the Raft node program: like oracle_node_trace, with the extra message payload.
 * in : n_in x 12 words {src, type, a, b, x0..x4, n_ents, first_ent, 0}; type 0 = let `a` microseconds pass and
 *      run the main loop until it idles.  ents_in: entries {term, msg_id, type, key, v1, v2, client} (7 words each).
 * out: records of 12 words {input index, src, dest, type, a, b, x0..x4, n_ents}; entries of emitted
 *      append_entries are appended to ents_out (7 words each, in order).  Returns #records or -1. */
int oracle_raft_trace(const msim_config *cfg, uint32_t node, const uint32_t *in, uint32_t n_in, const uint32_t *ents_in,
                      uint32_t *out, uint32_t out_cap, uint32_t *ents_out, uint32_t ents_cap, uint32_t *state_out) {
  msim_op *rows = (msim_op *)calloc(cfg->max_rows, sizeof(msim_op));
  u32 *payload = (u32 *)calloc(cfg->max_payload_words + 1, 4);
  sim_t *s = sim_new(cfg, 0, rows, payload);
  if (!s || !s->raft || node >= s->N) { free(rows); free(payload); return -1; }
  u32 n = 0, ne = 0;
  for (u32 i = 0; i < n_in; i++) {
    const u32 *m = in + 12 * i;
    s->n_out = 0;
    u32 steps = 1;
    if (m[1] == 0) { s->T += m[2]; steps = 64; }
    else {
      qent q = {s->T, i, m[2], m[3], (u8)m[0], (u8)m[1], NULL};
      rext *x = rext_new(m[4], m[5], m[6], m[7], m[8]);
      if (m[9]) { x->n_ents = m[9]; x->ents = (rentry *)calloc(m[9], sizeof(rentry));
        for (u32 k = 0; k < m[9]; k++) { const u32 *e = ents_in + 7 * (m[10] + k); rentry *t = &x->ents[k];
          t->term = e[0]; t->msg_id = e[1]; t->type = (u8)e[2]; t->key = (u8)e[3]; t->v1 = (u8)e[4]; t->v2 = (u8)e[5]; t->client = (u8)e[6]; } }
      q.r = x;
      raft_handle(s, node, &q);
      rext_free(x);
      steps = 64;  /* then let the main loop run (commit / apply) */
    }
    while (steps-- && raft_next_time(s, node) <= s->T) raft_act(s, node);
    for (u32 k = 0; k < s->n_out && n < out_cap; k++, n++) {
      u32 *o = out + 12 * n; outmsg *om = &s->out[k];
      o[0] = i; o[1] = om->src_ep; o[2] = om->dest_ep; o[3] = om->type; o[4] = om->a; o[5] = om->b;
      for (u32 j = 0; j < 5; j++) o[6 + j] = om->r ? om->r->x[j] : 0;
      o[11] = om->r ? om->r->n_ents : 0;
      for (u32 j = 0; om->r && j < om->r->n_ents && ne < ents_cap; j++, ne++) { u32 *e = ents_out + 7 * ne; rentry *t = &om->r->ents[j];
        e[0] = t->term; e[1] = t->msg_id; e[2] = t->type; e[3] = t->key; e[4] = t->v1; e[5] = t->v2; e[6] = t->client; }
      rext_free(om->r); om->r = NULL;
    }
    s->n_out = 0;
  }
  rnode *r = &s->raft[node];
  state_out[0] = r->role; state_out[1] = r->term; state_out[2] = r->commit_index; state_out[3] = r->last_applied; state_out[4] = r->log_n;
  state_out[5] = (u32)r->voted_for; state_out[6] = (u32)r->leader;
  sim_free(s); free(rows); free(payload);
  return (int)n;
}

/* Test hook for the txn-list-append programs: replays a conversation between the transactional nodes and the lin-kv
 * service outside the network.  Each input is {target endpoint, src endpoint, message type, a, b}: the target (a node, or
 * the service at endpoint N + CS) handles it at once; every message it emits is recorded as
 * {input index, src endpoint, dest endpoint, type, a, b}.  `payload` must already hold the micro-ops that txn requests
 * refer to (a = offset | n << 24) and have room behind `payload_used` words for the completed transactions.  Used to
 * replay golden vectors recorded from the reference's own demo/js/single_key_txn.js (tests/golden/make_golden_txn.py). */
int oracle_txn_trace(const msim_config *cfg, const uint32_t *in, uint32_t n_in, uint32_t *out, uint32_t out_cap,
                     uint32_t *payload, uint32_t payload_used) {
  msim_op *rows = (msim_op *)calloc(cfg->max_rows, sizeof(msim_op));
  sim_t *s = sim_new(cfg, 0, rows, payload);
  if (!s || !s->txn) { free(rows); return -1; }
  s->meta.n_payload_words = payload_used;
  u32 n = 0;
  for (u32 i = 0; i < n_in; i++) {
    const u32 *m = in + 5 * i;
    s->n_out = 0;
    qent q = {s->T, i, m[3], m[4], (u8)m[1], (u8)m[2], NULL};
    if (m[0] < s->N) node_handle(s, m[0], &q);
    else if (m[0] == svc_ep(s)) svc_handle(s, &q);
    else { sim_free(s); free(rows); return -1; }
    for (u32 k = 0; k < s->n_out && n < out_cap; k++, n++) {
      u32 *o = out + 6 * n;
      o[0] = i; o[1] = s->out[k].src_ep; o[2] = s->out[k].dest_ep; o[3] = s->out[k].type; o[4] = s->out[k].a; o[5] = s->out[k].b;
    }
  }
  sim_free(s); free(rows);
  return (int)n;
}

/* Test hook for the key-value services behind the proxy node (svc_nodes.inc): feeds `n_in` requests {client id, message type,
 * a} straight to the service of cfg->proxy_service — what `(s/handle! kv {:src client :body ...})` does in the reference's own
 * test/maelstrom/service_test.clj — and records {reply type, a} per request ({0, 0}: no reply).  Client ids are service-side
 * "clients" (the :src of the request, service.clj:166): any value below 256.  `instance` seeds the service's rand-int stream. */
int oracle_svc_trace(const msim_config *cfg, uint64_t instance, const uint32_t *in, uint32_t n_in, uint32_t *out) {
  msim_op *rows = (msim_op *)calloc(cfg->max_rows, sizeof(msim_op));
  uint32_t *payload = (uint32_t *)calloc(cfg->max_payload_words, 4);
  sim_t *s = sim_new(cfg, instance, rows, payload);
  if (!s || !s->svc) { if (s) sim_free(s); free(rows); free(payload); return -1; }
  free(s->svc->client_idx);
  s->svc->client_idx = (u32 *)calloc(256, 4);
  for (u32 i = 0; i < n_in; i++) {
    const u32 *m = in + 3 * i;
    if (m[0] >= 256) { sim_free(s); free(rows); free(payload); return -1; }
    s->n_out = 0;
    qent q = {s->T, i, m[2], i + 1, (u8)m[0], (u8)m[1], NULL};
    px_svc_handle(s, &q);
    out[2 * i] = s->n_out ? s->out[0].type : 0; out[2 * i + 1] = s->n_out ? s->out[0].a : 0;
  }
  sim_free(s); free(rows); free(payload);
  return (int)n_in;
}

/* Runs instances [first, first+n) into instance-major output slabs (same layout as the engine). */
int oracle_run(const msim_config *cfg, uint64_t first, uint32_t n, msim_op *rows, uint32_t *payload,
               msim_net_stats *stats, msim_inst_meta *meta, msim_event *journal) {
  for (uint32_t i = 0; i < n; i++) {
    int rc = oracle_run_instance(cfg, first + i, rows + (size_t)i * cfg->max_rows,
                                 payload + (size_t)i * cfg->max_payload_words, stats + i, meta + i,
                                 journal ? journal + (size_t)i * cfg->journal_capacity : NULL);
    if (rc) return rc;
  }
  return 0;
}

/* Exposed so tests can pin the datomic node's key hash against zlib. */
uint32_t oracle_dt_hash(uint32_t key) { return dt_hash(key); }
/* Exposed so tests can pin the integer samplers directly. */
uint32_t oracle_neg_ln_q16(uint32_t r) { return neg_ln_q16(r); }
uint32_t oracle_draw32(uint64_t seed, uint64_t instance, uint32_t stream, uint64_t ctr) {
  sim_t s; s.key = inst_key(seed, instance); return draw32(&s, stream, ctr);
}
/* adjacency masks (MW words per node) for a topology: pins broadcast.clj:40-185 */
int oracle_topology(uint32_t topology, uint32_t n, uint32_t *adj_out) {
  if (n == 0 || n > MAXN) return -1;
  sim_t *s = (sim_t *)calloc(1, sizeof(sim_t));
  s->N = n; s->cfg.topology = topology; build_topology(s);
  memcpy(adj_out, s->adj, (size_t)n * MW * 4);
  free(s);
  return 0;
}

/* Test hook for the txn-rw-register node: runs instance `instance` like oracle_run_instance and also returns the nodes'
 * final registers (kv: n_nodes x max_values words, lamport << 11 | node << 8 | value), Lamport clocks and the number of
 * txns each node still holds as unreplicated; `journal` as in oracle_run_instance. */
int oracle_hat_state(const msim_config *cfg, uint64_t instance, msim_op *rows, uint32_t *payload, msim_net_stats *stats,
                     msim_inst_meta *meta, uint32_t *kv, uint32_t *lamport, uint32_t *npend, msim_event *journal) {
  sim_t *s = sim_new(cfg, instance, rows, payload);
  if (!s || !s->hat) return -1;
  if (cfg->journal_capacity && !journal) { sim_free(s); return -1; }
  s->journal = journal;
  run_instance(s);
  *stats = s->st; *meta = s->meta;
  memcpy(kv, s->hat->kv, (size_t)s->N * cfg->max_values * 4);
  memcpy(lamport, s->hat->lamport, s->N * 4);
  memcpy(npend, s->hat->npend, s->N * 4);
  sim_free(s);
  return 0;
}

/* Test hook for the Raft node: runs instance `instance` like oracle_run_instance and also returns, in order, what every node's
 * main loop did in every round: {time us, node, id of the message it handled, or 0xFFFFFFFF for one of its timer / commit /
 * apply actions} (raft.py:577-585).  tests/test_raft_reference_replay.py feeds that schedule to the reference's own raft.py.
 * Returns the number of trace entries (may exceed trace_cap: then only the first trace_cap are stored), or -1. */
int oracle_raft_schedule(const msim_config *cfg, uint64_t instance, msim_op *rows, uint32_t *payload, msim_net_stats *stats,
                      msim_inst_meta *meta, msim_event *journal, uint32_t *trace, uint32_t trace_cap) {
  sim_t *s = sim_new(cfg, instance, rows, payload);
  if (!s || !s->raft) return -1;
  if (cfg->journal_capacity && !journal) { sim_free(s); return -1; }
  s->journal = journal; s->rtrace = trace; s->cap_rtrace = trace_cap;
  run_instance(s);
  *stats = s->st; *meta = s->meta;
  int n = (int)s->n_rtrace;
  sim_free(s);
  return n;
}
