// kafka8.hip — EIGHT kafka clusters per wavefront (workload/kafka.clj over demo/clojure/kafka.clj + the lin-kv service).
//
// Same program and the same rounds as kafka_kernel<> (sim_kernel_kafka.inc; specification: oracle/kafka_nodes.inc): node =
// demo/clojure/kafka.clj:1-172 (logs in 32-message chunks under lin-kv keys: send = read the chunk of the cached offset, cas it one
// message longer; poll = read the chunk of every requested offset, one after the other; committed offsets under the lin-kv key "offsets"),
// service = service.clj:31-61,141-155,245-263, client = workload/kafka.clj:155-245 (send / poll + commit_offsets / assign / crash; not
// Reusable), generator and final phase = [upstream] jepsen.tests.kafka as workload/kafka.clj:288-311 configures it (restated; parity
// unpinned: babashka node, upstream generator and checker — DESIGN.md §3).  What changes is the mapping: kafka_kernel<> ran one cluster per
// wavefront — 5 nodes + lin-kv = 6 live lanes of 64 — and paid its whole instruction stream for them.  Here a cluster is a group of 8
// lanes (lane l < N = node l + its client, lane N = lin-kv) and a wavefront carries eight clusters (txn8.hip's scheme: what is uniform
// per cluster lives in VGPRs, a "ballot" is the group's slice, another lane's value comes by ds_bpermute within the group, the time
// reduction is three DPP steps, a barrier between the nodes' and the service's part of a round is a wavefront-scope fence).
//
// Scope (engine.hip picks this kernel when all of it holds, else kafka_kernel<> runs): n_nodes <= 7, one worker per node, net journal off.
//
// LDS of a wavefront (word-major / slot-major, lane e at [.. * 64 + e]: no bank conflicts): node / service queues (RQ envelopes, the
// rest spills to HBM), client inboxes (CQ + HBM spill: 8 in all, the oracle's limit), the nodes' request handlers (the first SL of 8 x 8
// words; the others — in use only while lin-kv replies are lost or late — in HBM), the nodes' offset caches and the clients' offsets (8
// words each), per cluster the generator's key pool, lin-kv's log lengths / offset-list lengths and the nemesis shuffle: 13.9 KiB.
// History rows go straight to HBM.  The logs and the committed-offset lists live in HBM scratch as in kafka_kernel<>.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "group8.h"
#include "layout_thresholds.h"

namespace {

constexpr u32 GS = 8u;            // lanes per cluster
constexpr u32 RQ = 3u;            // LDS envelopes per node / service queue
constexpr u32 CQ = 1u;            // LDS envelopes per client inbox
constexpr u32 K8_CLIENT_CAP = 8u; // envelopes a client inbox holds in all (oracle: inbox_push, MSIM_WL_KAFKA)
constexpr u32 KF_SLOTS = 8u;      // request handlers in flight per node (the oracle's limit) ...
constexpr u32 SL = 2u;            // ... of which in LDS
constexpr u32 KF_KEYS = 8u, KF_CHUNK = 32u, KF_ABSENT = 0xFFFFu, KF_OFFSETS_KEY = 0x80000000u, KF_SEND_NONE = 0x7FFu << 17;
constexpr u32 KSW = 8u;           // words per request handler: client msg_id, request block, rpc msg_id, flags, from | msg << 16, offset, chunk counts x 8
enum { M_WRITE = 14, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR,
       M_SEND = 30, M_SEND_OK, M_POLL, M_POLL_OK, M_LIST_OFFSETS, M_LIST_OFFSETS_OK, M_COMMIT_OFFSETS, M_COMMIT_OFFSETS_OK };
enum { KK_SEND = 1, KK_POLL = 2, KK_LIST = 3, KK_COMMIT = 4 };
enum { PH_KF_FINAL = PH_DONE + 1 };
enum { S_GEN3 = 3 };
// flags word of a handler: used | kind << 1 | stage << 4 | key << 6 | j << 9 | nk << 13
#define KS_USED(w_) ((w_) & 1u)
#define KS_KIND(w_) (((w_) >> 1) & 7u)
#define KS_STAGE(w_) (((w_) >> 4) & 3u)
#define KS_KEY(w_) (((w_) >> 6) & 7u)
#define KS_J(w_) (((w_) >> 9) & 15u)
#define KS_NK(w_) (((w_) >> 13) & 0xFFu)
#define KS_MAKE(kind_, stage_, key_, j_, nk_) (1u | ((kind_) << 1) | ((stage_) << 4) | ((key_) << 6) | ((j_) << 9) | ((nk_) << 13))

struct K8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_slots, off_cache, off_coff, off_gen, off_kl, off_misc;   // LDS byte offsets (queues at 0)
  u64 xslots_off;                                        // word offset of the nodes' handlers SL .. KF_SLOTS-1 inside the per-instance scratch
  u32 node_spill, client_spill;                          // HBM spill entries per node-or-service queue / client inbox
  u64 client_spill_off;                                  // word offset of the clients' spill area inside the per-instance scratch
  u32 round_limit;
};

template <bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) kafka8_kernel(const K8Params up) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = up.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 3, gbase = lane & 56u;
  const u32 N = p.N;
  const bool is_node = l < N, is_svc = l == N;
  const u32 SVC = 2 * N;   // the service's endpoint index
  const u32 inst_raw = blockIdx.x * 8u + grp;
  const bool real = inst_raw < up.n_inst;
  const u32 inst = real ? inst_raw : up.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << l) - 1u;
  const u32 all_nodes = (1u << N) - 1u;
  const u32 max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz, mw = p.cfg.max_writes_per_key, cap = mw + 1u;
  const u32 round_limit = up.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  u32 *const g_log = g_scr;                                           // [KF_KEYS][cap] the messages
  u32 *const g_upd = g_scr + (size_t)KF_KEYS * cap;                    // [KF_KEYS][cap + 1] version << 16 | committed offset
  const u32 qlane = l <= N ? l : 0u;
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)qlane * up.node_spill;
  uint4 *const my_cspill = reinterpret_cast<uint4 *>(g_scr + up.client_spill_off) + (size_t)(is_node ? l : 0u) * up.client_spill;
  const u32 my_spill_cap = l <= N ? up.node_spill : 0u;
  u32 *const xslots = g_scr + up.xslots_off;                           // [node][KF_SLOTS - SL][KSW]

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node / service queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + up.off_cq) + lane;                      // client inbox
  u32 *const lslots = reinterpret_cast<u32 *>(smem + up.off_slots);                             // handler i (< SL) of lane e, word w at [(i * KSW + w) * 64 + e]
  u32 *const my_cache = reinterpret_cast<u32 *>(smem + up.off_cache) + lane;                    // offset cache (kafka.clj:28-42): key k at my_cache[k * 64]
  u32 *const my_coff = reinterpret_cast<u32 *>(smem + up.off_coff) + lane;                      // the client's offsets, in the order of its keys: j at my_coff[j * 64]
  u32 *const gen = reinterpret_cast<u32 *>(smem + up.off_gen) + grp * 36;                       // active[16], next_val[16], next_key
  u32 *const klen = reinterpret_cast<u32 *>(smem + up.off_kl) + grp * 16;                       // [KF_KEYS] lin-kv: messages of every key's log
  u32 *const nupd = klen + KF_KEYS;                                                            // [KF_KEYS] entries of the key's committed-offset list
  u32 *const misc = reinterpret_cast<u32 *>(smem + up.off_misc) + grp * GS;
  // word w of handler i of node nd of THIS cluster (LDS for the first SL handlers, HBM beyond)
  auto hs_get = [&](u32 nd, u32 i, u32 w) -> u32 { return i < SL ? lslots[(i * KSW + w) * 64u + gbase + nd] : xslots[(nd * (KF_SLOTS - SL) + (i - SL)) * KSW + w]; };
  auto hs_set = [&](u32 nd, u32 i, u32 w, u32 v) { if (i < SL) lslots[(i * KSW + w) * 64u + gbase + nd] = v; else xslots[(nd * (KF_SLOTS - SL) + (i - SL)) * KSW + w] = v; };

  for (u32 i = lane; i < SL * KSW * 64u; i += 64) lslots[i] = 0;
  for (u32 k = 0; k < KF_KEYS; k++) { my_cache[k * 64u] = 0; my_coff[k * 64u] = 0; }
  if (real && is_node) for (u32 i = 0; i < (KF_SLOTS - SL) * KSW; i++) xslots[l * (KF_SLOTS - SL) * KSW + i] = 0;
  for (u32 i = l; i < 16; i += GS) { gen[i] = i; gen[16 + i] = 1; }
  if (l == 0) gen[32] = p.cfg.key_count;
  klen[l] = 0; nupd[l] = 0;   // (GS == KF_KEYS)
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node / service state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, node_msgid = 0, part = 0;
  u32 off_exists = 0, off_ver = 0;   // lin-kv lane: the key "offsets"
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_f = 0, c_value = 0, process = l, m_f = 0, m_value = 0, cin_n = 0, csp_n = 0;
  u32 kc_n = 0, kc_keys = 0, kc_stage = 0, kc_fin = 0, kc_done = 0, kc_ref = 0;   // offsets map: kc_n keys (3 bits each in kc_keys), values in my_coff
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, nem_next = 0, nem_j = 0, sleep_until = 0, final_deadline = 0;
  u32 loss_on = 0, next_id = 0, n_rows = 0, n_payload = 0, flags = 0, rounds = 0;
  bool alive = real;

  auto q_push = [&](const uint4 m) {
    if (in_n < RQ) { my_q[in_n * 64u] = m; in_n++; return; }
    if (sp_n < my_spill_cap) { my_spill[sp_n++] = m; return; }
    my_flags |= MSIM_FLAG_INBOX_OVERFLOW;
  };
  // an envelope for THIS lane's node/service arrives (net.clj:189-221)
  auto arrive = [&](u32 id, u32 type, u32 a, u32 b, u32 src) {
    u32 lat = 0;
    if (src < N || src == SVC) {  // neither end is a client
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * g8_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
    }
    if (NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss) return;
    uint4 m = make_uint4(T + lat * 1000u, (id << 8) | type, a, b | (src << 24));
    if (!have_pm) { pm = m; have_pm = true; return; }
    if (m.x < pm.x || (m.x == pm.x && m.y < pm.y)) { const uint4 t = m; m = pm; pm = t; }
    q_push(m);
  };
  auto try_commit = [&](const uint4 e) {
    const u32 src = e.w >> 24;
    if (NEM && src < N && ((part >> src) & 1)) return;  // partitioned (node <-> node only; never happens in this program)
    cm = e;
    deliver_at = e.x <= T ? T : T + ((e.x - T) / 1000u) * 1000u;  // (Thread/sleep (long dt)) net.clj:236-238
  };
  auto poll = [&]() {
    if (have_pm) {
      have_pm = false;
      if (alive && deliver_at == INF && (in_n | sp_n) == 0) try_commit(pm);
      else q_push(pm);
    }
    while (alive && l <= N && deliver_at == INF && (in_n | sp_n) != 0) {
      u32 best = 0; bool in_spill = false;
      uint2 bk = make_uint2(INF, INF);
      for (u32 i = 0; i < in_n; i++) {
        const uint2 kk = *reinterpret_cast<const uint2 *>(&my_q[i * 64u]);
        if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; }
      }
      for (u32 i0 = 0; i0 < sp_n; i0 += 8) {   // (the service takes every RPC of the cluster: eight keys per round trip)
        uint2 kq[8];
#pragma unroll
        for (u32 t = 0; t < 8; t++) kq[t] = *reinterpret_cast<const uint2 *>(&my_spill[min(i0 + t, sp_n - 1)]);
#pragma unroll
        for (u32 t = 0; t < 8; t++) if (i0 + t < sp_n && (kq[t].x < bk.x || (kq[t].x == bk.x && kq[t].y < bk.y))) { bk = kq[t]; best = i0 + t; in_spill = true; }
      }
      uint4 e;
      if (in_spill) { e = my_spill[best]; sp_n--; if (best != sp_n) my_spill[best] = my_spill[sp_n]; }
      else { e = my_q[best * 64u]; in_n--; if (best != in_n) my_q[best * 64u] = my_q[in_n * 64u]; }
      try_commit(e);
    }
  };
  // lin-kv: elements of chunk `ch` of key `k` (0: the lin-kv key does not exist)
  auto chunk_count = [&](u32 k, u32 ch) -> u32 {
    const u32 lo = ch * KF_CHUNK, len = klen[k];
    return len <= lo ? 0u : min(len - lo, KF_CHUNK);
  };
  // committed offset of key `k` in the "offsets" map at version `ver`: 1 + offset, 0 = no entry (the newest entry answers a commit with one
  // load, a reader of an older version bisects)
  auto committed_at = [&](u32 k, u32 ver) -> u32 {
    const u32 n = nupd[k];
    if (!n) return 0u;
    const u32 *const u = g_upd + (size_t)k * (cap + 1);
    const u32 last = u[n - 1u];
    if ((last >> 16) <= ver) return 1u + (last & 0xFFFFu);
    u32 a = 0, b = n - 1u;   // u[b] is newer than ver; a = entries known to be at or below it
    while (a < b) { const u32 mid = (a + b) >> 1; if ((u[mid] >> 16) <= ver) a = mid + 1u; else b = mid; }
    return a ? 1u + (u[a - 1u] & 0xFFFFu) : 0u;
  };

  const u64 pc = rate ? (u64)N * 65536ull * 1000ull / (30ull * (u64)rate) : ~0ull;   // a :crash every 30 s per worker on average ([upstream] jepsen.tests.kafka), as a 16-bit threshold
#ifdef K8_PROF   // developer build (tools/variant_lib.sh k8prof kafka8.hip -DK8_PROF; tools/kafka8_prof_report.py): cycles of a wavefront by section of the round
  u64 kp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kp_t = __builtin_readcyclecounter(); u32 kp_rounds = 0;
#define K8_MARK(i_) { const u64 kp_n = __builtin_readcyclecounter(); kp[i_] += kp_n - kp_t; kp_t = kp_n; }
#else
#define K8_MARK(i_)
#endif
  for (;;) {
    if (!__ballot(alive)) break;
#ifdef K8_PROF
    kp_rounds++;
#endif
    const u32 busy_mask = GB(busy);
    const u32 pend_mask = GB(mark);   // clients between the two RPCs of a poll

    // ---- time-free phase transitions ----
    if (__ballot(alive && !(phase == PH_MAIN && ((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff))))) {
      for (;;) {
        bool ch = false;
        if (alive) {
          if (phase == PH_INIT_WAIT && !busy_mask) { phase = PH_MAIN_START; ch = true; }
          if (phase == PH_MAIN_START) { cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; nem_next = T; next_msg_id = 0; loss_on = 1; phase = PH_MAIN; ch = true; }
          if (phase == PH_MAIN && !((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff)) && !(rate == 0 && T < cutoff)) { phase = PH_DRAIN; ch = true; }
          if (phase == PH_DRAIN && !busy_mask) {
            phase = NEM ? PH_NEM_FINAL : PH_SLEEP;
            if (phase == PH_SLEEP) sleep_until = T + p.cfg.quiesce_ms * 1000u;
            ch = true;
          }
          // every worker has polled until nothing came (or the 10 s of workload/kafka.clj:305-306 are over)
          if (phase == PH_KF_FINAL && GB(is_node && (busy || (!kc_done && T < final_deadline))) == 0) { phase = PH_DONE; ch = true; }
        }
        if (!__ballot(ch)) break;
      }
      if (phase == PH_DONE) alive = false;
      if (!__ballot(alive)) break;
    }
    if (alive && ++rounds > round_limit) { flags |= MSIM_FLAG_ROUND_LIMIT; alive = false; }

    // ---- R0: time ----
    const bool gen_live = rate > 0 && gen_next < cutoff;
    const bool nem_live = NEM && nem_next < cutoff;
    const u32 free_mask = all_nodes & ~busy_mask;
    u32 due = INF;
    if (phase == PH_INIT || phase == PH_NEM_FINAL || phase == PH_FINAL) due = T;
    else if (phase == PH_SLEEP) due = sleep_until;
    else if (phase == PH_KF_FINAL) { if (T < final_deadline && GB(is_node && !busy && !kc_done) != 0) due = T; }
    else if (phase == PH_MAIN) {
      if (nem_live) due = max(nem_next, T);
      if (gen_live && free_mask) due = min(due, max(gen_next, T));
      if (rate == 0 && !nem_live) due = min(due, cutoff);
    }
    bool timeout_round = false;
    {
      const bool none_due = GB(deliver_at <= T) == 0;
      const bool jump = alive && due > T && !pend_mask && none_due;
      if (__ballot(jump)) {
        u32 k = deliver_at == INF ? INF : deliver_at * 2;
        if (busy) k = min(k, timeout_at * 2 + 1);
        u32 km = g8_min<8>(k);
        if (due != INF) km = min(km, due * 2);
        if (jump) {
          if (km == INF) { flags |= MSIM_FLAG_ROUND_LIMIT; alive = false; }
          else { timeout_round = (km & 1) != 0; T = max(T, km >> 1); }
        }
      }
    }

    bool inv_row = false; u32 inv_packed = 0, inv_value = 0, inv_len = 0;
    bool cmp_row = false; u32 cmp_packed = 0, cmp_value = 0, cmp_len = 0;
    u32 nem_rows = 0, nem_f = 0, nem_v1 = 0, nem_v2 = 0, nem_len2 = 0;

    auto complete = [&](u32 type, u32 err, u32 value, u32 len) {
      busy = false;
      if (kind != K_OP) { if (type != MSIM_T_OK) my_flags |= MSIM_FLAG_ROUND_LIMIT; return; }
      cmp_row = true; cmp_packed = type | (c_f << 2) | (err << 7) | (process << 12);
      cmp_value = value; cmp_len = len;
      if (type == MSIM_T_INFO) {   // crashed process: a new process id and a fresh client (not Reusable, workload/kafka.clj:239-243)
        process += N; next_msg_id = 0; cin_n = 0; csp_n = 0;
        kc_n = 0; kc_stage = 0;
      }
    };
    // what the op's :fail / :info carries: the value it was invoked with
    auto abort_op = [&](u32 type, u32 err) {
      kc_stage = 0;
      if (c_f == MSIM_F_SEND) complete(type, err, c_value | KF_SEND_NONE, 0);
      else if (c_value == MSIM_NO_VALUE) complete(type, err, MSIM_NO_VALUE, 0);
      else complete(type, err, c_value & 0xFFFFFFu, c_value >> 24);
    };
    // the client's recv! consumes one envelope (client.clj:94-107)
    auto client_deliver = [&](u32 qtype, u32 qa, u32 qb) {
      s_recv_cl++;
      if (!(busy && qb == want)) return;
      if (kind != K_OP) { complete(MSIM_T_OK, 0, c_value, 0); return; }   // init_ok
      switch (qtype) {
        case M_SEND_OK: kc_stage = 0; complete(MSIM_T_OK, 0, c_value | (qa << 17), 0); break;
        case M_POLL_OK: {
          u32 pp = qa & 0xFFFFFFu; const u32 end = pp + (qa >> 24); bool any = false;
          while (pp < end) {   // advance the local offsets to max polled + 1, :181-186
            const u32 h = g_pay[pp], k = h & 7u, n = (h >> 8) & 0xFFu, o = h >> 16;
            pp += 1u + (n + 1u) / 2u;
            if (!n) continue;
            any = true;
            for (u32 j = 0; j < kc_n; j++) if (((kc_keys >> (3u * j)) & 7u) == k && o + n > my_coff[j * 64u]) my_coff[j * 64u] = o + n;
          }
          if (kc_fin) kc_done = any ? 0u : 1u;
          if (any) { kc_stage = 2; kc_ref = qa; mark = true; }   // (when (seq offsets) (commit_offsets! ...)): the next RPC of this op
          else { kc_stage = 0; complete(MSIM_T_OK, 0, (qa >> 24) ? (qa & 0xFFFFFFu) : MSIM_NO_VALUE, qa >> 24); }
        } break;
        case M_COMMIT_OFFSETS_OK: kc_stage = 0; complete(MSIM_T_OK, 0, kc_ref & 0xFFFFFFu, kc_ref >> 24); break;
        case M_LIST_OFFSETS_OK: {   // offsets := {k (or (offsets k) (committed k) 0)} over the op's keys, in their order, :211-219
          const u32 ref = qa & 0xFFFFFFu, n = qa >> 24;
          u32 nkeys = 0, no[KF_KEYS];
#pragma unroll
          for (u32 j = 0; j < KF_KEYS; j++) {
            no[j] = 0;
            if (j < n) {
              const u32 w = g_pay[ref + j], k = w & 7u; u32 o = (w >> 31) ? ((w >> 8) & 0x7FFFFFu) : 0u;
              for (u32 e = 0; e < kc_n; e++) if (((kc_keys >> (3u * e)) & 7u) == k) o = my_coff[e * 64u];
              nkeys |= k << (3u * j); no[j] = o;
            }
          }
#pragma unroll
          for (u32 j = 0; j < KF_KEYS; j++) if (j < n) my_coff[j * 64u] = no[j];
          kc_n = n; kc_keys = nkeys;
          kc_stage = 0; complete(MSIM_T_OK, 0, c_value & 0xFFFFFFu, c_value >> 24);
        } break;
        case M_ERROR:
          abort_op(MSIM_T_FAIL, qa == 11 ? MSIM_ERR_TEMPORARILY_UNAVAILABLE : qa == 20 ? MSIM_ERR_KEY_DOES_NOT_EXIST : qa == 30 ? MSIM_ERR_TXN_CONFLICT : MSIM_ERR_PRECONDITION_FAILED);
          break;
        default: break;
      }
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) {
        if (kind != K_OP) complete(MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_value, 0);
        else abort_op(c_f == MSIM_F_ASSIGN ? MSIM_T_FAIL : MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT);   // (c/with-errors op #{:assign} ..), :205
      }
    }
    K8_MARK(0)   // [0] = phase checks, R0 (time), timeouts
    bool normal = alive && !timeout_round;   // this cluster runs R1-R4 in this wave-round
    if (__ballot(normal)) {
      // ---- R1: scheduler ----
      const bool act = normal && due <= T;
      if (__ballot(act && phase == PH_INIT)) {
        if (act && phase == PH_INIT) { if (is_node) { mark = true; kind = K_INIT; } phase = PH_INIT_WAIT; }
      }
      #include "group8_nemesis.inc"
      {
        const bool gen_on = act && phase == PH_MAIN && gen_live && gen_next <= T && free_mask != 0;
        if (__ballot(gen_on)) {
          const u32 nfree = __popc(free_mask);
          const u32 kk = gen_k;
          const u64 h = draw64(key, S_GEN, kk);
          const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          const u32 pick = scale32(r_lo, nfree);
          const bool sel = gen_on && is_node && !busy && (u32)__popc(free_mask & lt) == pick;
          // one operation ([upstream] jepsen.tests.kafka, oracle/kafka_nodes.inc kf_generate): every lane of the cluster computes it, lane 0 owns the key pool
          const u64 h2 = draw64(key, S_GEN2, kk);
          u32 f = 0, val = MSIM_NO_VALUE, bad = 0;
          u32 npay = 0;   // payload words the operation takes (an :assign's keys)
          const bool is_crash = ((h2 >> 48) & 0xFFFFu) < pc;
          const bool is_assign = !is_crash && ((h2 >> 44) & 0xFu) < 2;
          const bool is_poll = !is_crash && !is_assign && !((h2 >> 43) & 1);
          const bool is_send = !is_crash && !is_assign && !is_poll;
          if (is_crash) f = MSIM_F_CRASH;
          if (__ballot(gen_on && is_assign)) {   // an :assign of a non-empty subset of the keys seen so far, in key order
            if (gen_on && is_assign) {
              const u32 nk = gen[32];
              u32 m = (u32)(h2 >> 8) & ((1u << nk) - 1u);
              if (!m) m = 1u << (((u32)h2 >> 4) % nk);
              const u32 n = (u32)__popc(m);
              if (n_payload + n > max_pay) bad = MSIM_FLAG_PAYLOAD_OVERFLOW;
              else {
                if (l < nk && ((m >> l) & 1)) g_pay[n_payload + __popc(m & lt)] = l;
                f = MSIM_F_ASSIGN; val = n_payload | (n << 24); npay = n;
              }
            }
          }
          if (is_poll) f = MSIM_F_POLL;
          if (__ballot(gen_on && is_send)) {
            u32 k = 0, v = 0, ki = 0, nk = 0;
            if (gen_on && is_send) {
              const u32 kc = p.cfg.key_count;
              const u64 h3 = draw64(key, S_GEN3, (u64)kk * 8);
              const u32 x = scale32((u32)(h3 >> 32), (1u << kc) - 1) + 1;
              ki = 31 - (u32)__clz((int)x);
              k = gen[ki]; v = gen[16 + ki]; nk = gen[32];
            }
            wave_lds_fence();
            if (gen_on && is_send) {
              if (v + 1 > mw && nk >= KF_KEYS) bad = MSIM_FLAG_VALUES_OVERFLOW;
              else if (l == 0) {
                gen[16 + ki] = v + 1;
                if (v + 1 > mw) { gen[ki] = nk; gen[32] = nk + 1; gen[16 + ki] = 1; }   // key used up: a fresh one takes its place in the pool
              }
              f = MSIM_F_SEND; val = k | (v << 6);
            }
            wave_lds_fence();
          }
          if (gen_on && bad) { flags |= bad; phase = PH_DONE; alive = false; normal = false; }
          else if (gen_on) {
            if (sel) { mark = true; kind = K_OP; m_f = f; m_value = val; }
            n_payload += npay;
            gen_k++;
            gen_next = T + __umulhi(r_hi, p.gen_period2_us);
          }
        }
      }
      if (__ballot(act && (phase == PH_NEM_FINAL || phase == PH_SLEEP || phase == PH_FINAL || phase == PH_KF_FINAL))) {
        if (NEM && act && phase == PH_NEM_FINAL) {
          part = 0; nem_rows = 2; nem_f = MSIM_F_STOP_PARTITION; nem_v1 = MSIM_NO_VALUE; nem_v2 = MSIM_NO_VALUE; nem_len2 = 0;
          phase = PH_SLEEP; sleep_until = T + p.cfg.quiesce_ms * 1000u;
        } else if (act && (phase == PH_SLEEP || phase == PH_FINAL)) {
          if (phase == PH_SLEEP && T >= sleep_until) phase = PH_FINAL;
          if (phase == PH_FINAL) {   // [upstream] the final generator: every worker assigns all keys from the beginning (then polls, PH_KF_FINAL)
            const u32 n = gen[32];
            if (n_payload + N * n > max_pay) { flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; phase = PH_DONE; alive = false; normal = false; }
            else {
              if (is_node) {
                for (u32 i = 0; i < n; i++) g_pay[n_payload + l * n + i] = i | 0x80000000u;
                mark = true; kind = K_OP; kc_fin = 1; m_f = MSIM_F_ASSIGN; m_value = (n_payload + l * n) | (n << 24);
              }
              n_payload += N * n;
              final_deadline = T + 10000000u; phase = PH_KF_FINAL;
            }
          }
        } else if (act && phase == PH_KF_FINAL) {
          if (is_node && !busy && !kc_done && T < final_deadline) { mark = true; kind = K_OP; m_f = MSIM_F_POLL; m_value = MSIM_NO_VALUE; }
        }
      }

      K8_MARK(1)   // [1] = R1 scheduler / generator
      // ---- R2: marked clients invoke (or send the second RPC of a poll); the request goes to this lane's own node ----
      if (__ballot(mark && normal)) {
        const bool inv = mark && normal;
        // poll requests carry the client's {key offset} map: payload words in slot order
        const bool is_pollrq = inv && kind == K_OP && kc_stage != 2 && m_f == MSIM_F_POLL;
        const u32 pw = is_pollrq ? kc_n : 0u;
        u32 pexcl = 0, ptotal = 0;
        for (u32 s = 0; s < N; s++) { const u32 v = GGET(pw, s); pexcl += s < l ? v : 0u; ptotal += v; }
        bool pay_ok = true;
        if (ptotal && n_payload + ptotal > max_pay) { flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; pay_ok = false; }
        bool send_rpc = false; u32 rq_type = 0, rq_a = 0;
        if (inv) {
          mark = false; busy = true; send_rpc = true;
          if (kind == K_INIT) { rq_type = M_INIT; next_msg_id = 0; }
          else if (kc_stage == 2) { rq_type = M_COMMIT_OFFSETS; rq_a = kc_ref; }   // :223-230
          else {
            c_f = m_f; c_value = m_value;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (c_f << 2) | (process << 12);
            if (c_f == MSIM_F_SEND) { inv_value = c_value | KF_SEND_NONE; inv_len = 0; rq_type = M_SEND; rq_a = c_value; kc_stage = 1; }
            else if (c_f == MSIM_F_POLL) {
              const u32 off = pay_ok ? n_payload + pexcl : 0u;
              if (pay_ok) for (u32 j = 0; j < kc_n; j++) g_pay[off + j] = ((kc_keys >> (3u * j)) & 7u) | (my_coff[j * 64u] << 8);
              c_value = kc_n ? (off | (kc_n << 24)) : MSIM_NO_VALUE;
              inv_value = kc_n ? off : MSIM_NO_VALUE; inv_len = kc_n;
              rq_type = M_POLL; rq_a = kc_n ? c_value : 0u; kc_stage = 1;
            } else if (c_f == MSIM_F_ASSIGN) {
              const u32 ref = c_value & 0xFFFFFFu, n = c_value >> 24;
              inv_value = ref; inv_len = n;
              if (g_pay[ref] >> 31) {   // (reset! offsets (zipmap value (repeat 0))), :207-209
                u32 nkeys = 0;
                for (u32 j = 0; j < n; j++) { nkeys |= (g_pay[ref + j] & 7u) << (3u * j); my_coff[j * 64u] = 0; }
                kc_n = n; kc_keys = nkeys;
                send_rpc = false; complete(MSIM_T_OK, 0, ref, n);
              } else { rq_type = M_LIST_OFFSETS; rq_a = c_value; kc_stage = 1; }
            } else {   // :crash, :222
              inv_value = MSIM_NO_VALUE; inv_len = 0;
              send_rpc = false; complete(MSIM_T_INFO, 0, MSIM_NO_VALUE, 0);
            }
          }
        }
        if (pay_ok) n_payload += ptotal;
        const u32 inv_mask = GB(send_rpc);
        if (send_rpc) {
          want = ++next_msg_id;
          timeout_at = T + (kind == K_OP ? p.cfg.client_timeout_ms : 10000u) * 1000u;
          s_send_cl++;
          arrive(next_id + __popc(inv_mask & lt), rq_type, rq_a, want, N + l);
        }
        next_id += __popc(inv_mask);
        poll();
      }

      K8_MARK(2)   // [2] = R2 invocations
      // ---- R3: one input per node, then one for the service (endpoint order) ----
      bool to_svc = false, rep = false, svc_rep = false;   // node -> service, node -> own client, service -> node
      u32 o_type = 0, o_a = 0, o_b = 0, o_dest = 0, need_words = 0, done_slot = 0;
      const bool svc_due = normal && is_svc && deliver_at <= T;
      if (normal && is_node && deliver_at <= T) {
        const uint4 q = cm; deliver_at = INF;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N && qsrc < SVC) s_recv_cl++; else s_recv_sv++;
        {
          // read the chunk of `offset` of key `k_` for handler `i` (whose flags word becomes fl with that key)
          auto read_chunk = [&](u32 i, u32 fl, u32 k_, u32 offset) {
            const u32 rid = ++node_msgid;
            hs_set(l, i, 2, rid); hs_set(l, i, 3, (fl & ~(7u << 6)) | (k_ << 6)); hs_set(l, i, 5, offset);
            to_svc = true; o_type = M_READ; o_a = k_ | ((offset / KF_CHUNK) << 8); o_b = rid;
          };
          switch (qtype) {
            case M_INIT: rep = true; o_type = M_INIT_OK; o_b = qb; break;
            case M_SEND: case M_POLL: case M_LIST_OFFSETS: case M_COMMIT_OFFSETS: {
              u32 i = 0; while (i < KF_SLOTS && KS_USED(hs_get(l, i, 3))) i++;
              if (i == KF_SLOTS) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; break; }
              hs_set(l, i, 0, qb); hs_set(l, i, 1, 0); hs_set(l, i, 4, 0); hs_set(l, i, 6, 0); hs_set(l, i, 7, 0);
              if (qtype == M_SEND) {
                hs_set(l, i, 4, (qa >> 6) << 16);
                read_chunk(i, KS_MAKE(KK_SEND, 1u, 0u, 0u, 0u), qa & 7u, my_cache[(qa & 7u) * 64u]);
              } else if (qtype == M_POLL) {
                const u32 nk = qa >> 24;
                if (nk == 0) { rep = true; o_type = M_POLL_OK; o_a = 0; o_b = qb; break; }   // no offsets: {:msgs {}}
                hs_set(l, i, 1, qa & 0xFFFFFFu);
                const u32 w = g_pay[qa & 0xFFFFFFu];
                read_chunk(i, KS_MAKE(KK_POLL, 1u, 0u, 0u, nk), w & 7u, w >> 8);
              } else {
                const u32 rid = ++node_msgid;
                hs_set(l, i, 1, qa & 0xFFFFFFu); hs_set(l, i, 2, rid); hs_set(l, i, 3, KS_MAKE(qtype == M_LIST_OFFSETS ? KK_LIST : KK_COMMIT, 1u, 0u, 0u, qa >> 24));
                to_svc = true; o_type = M_READ; o_a = KF_OFFSETS_KEY; o_b = rid;   // get-offsets, :141-147
              }
            } break;
            case M_READ_OK: case M_CAS_OK: case M_ERROR: {
              u32 i = 0;
              while (i < KF_SLOTS) { if (KS_USED(hs_get(l, i, 3)) && hs_get(l, i, 2) == qb) break; i++; }
              if (i == KF_SLOTS) break;  // handle-reply!: no such rpc
              const u32 fl = hs_get(l, i, 3), k_ = KS_KEY(fl), off = hs_get(l, i, 5), base = off - off % KF_CHUNK;
              switch (KS_KIND(fl)) {
                case KK_SEND:
                  if (KS_STAGE(fl) == 1) {
                    const u32 cnt = qtype == M_READ_OK ? qa : 0u;   // (exceptionally [_] [])
                    my_cache[k_ * 64u] = max(my_cache[k_ * 64u], base + cnt);
                    if (cnt >= KF_CHUNK) { my_cache[k_ * 64u] = max(my_cache[k_ * 64u], base + KF_CHUNK); read_chunk(i, fl, k_, my_cache[k_ * 64u]); break; }   // chunk full: recur
                    const u32 rid = ++node_msgid;
                    const u32 s4 = (hs_get(l, i, 4) & 0xFFFF0000u) | cnt;
                    hs_set(l, i, 2, rid); hs_set(l, i, 3, (fl & ~(3u << 4)) | (2u << 4)); hs_set(l, i, 4, s4);
                    to_svc = true; o_type = M_CAS; o_a = k_ | ((off / KF_CHUNK) << 3) | (cnt << 9) | ((s4 >> 16) << 14); o_b = rid;
                  } else {
                    rep = true; o_b = hs_get(l, i, 0);
                    if (qtype == M_CAS_OK) { const u32 o = base + (hs_get(l, i, 4) & 0xFFFFu); my_cache[k_ * 64u] = max(my_cache[k_ * 64u], o + 1u); o_type = M_SEND_OK; o_a = o; }
                    else { o_type = M_ERROR; o_a = qa == 22 ? 30u : qa; }   // "cas conflict", :108-110
                    hs_set(l, i, 3, 0);
                  }
                  break;
                case KK_POLL: {
                  const u32 cnt = qtype == M_READ_OK ? qa : 0u, j = KS_J(fl), nk = KS_NK(fl);
                  my_cache[k_ * 64u] = max(my_cache[k_ * 64u], base + cnt);
                  hs_set(l, i, 6 + (j >> 2), hs_get(l, i, 6 + (j >> 2)) | (cnt << (8u * (j & 3u))));
                  const u32 s1 = hs_get(l, i, 1);
                  if (j + 1 < nk) { const u32 w = g_pay[s1 + j + 1]; read_chunk(i, (fl & ~(15u << 9)) | ((j + 1u) << 9), w & 7u, w >> 8); break; }
                  // poll_ok: sized here, written below (payload words are handed out in node order)
                  rep = true; o_type = M_POLL_OK; o_b = hs_get(l, i, 0); done_slot = i;
                  for (u32 e = 0; e < nk; e++) {
                    const u32 w = g_pay[s1 + e], i0 = (w >> 8) % KF_CHUNK, c = (hs_get(l, i, 6 + (e >> 2)) >> (8u * (e & 3u))) & 0xFFu;
                    const u32 n = c > i0 ? c - i0 : 0u;
                    need_words += 1u + (n + 1u) / 2u;
                  }
                } break;
                case KK_LIST:
                  rep = true; o_type = M_LIST_OFFSETS_OK; o_b = hs_get(l, i, 0); done_slot = i;
                  hs_set(l, i, 4, qtype == M_READ_OK ? qa : KF_ABSENT);   // (exceptionally [res] {})
                  need_words = KS_NK(fl);
                  break;
                default:   // KK_COMMIT
                  if (KS_STAGE(fl) == 1) {
                    const u32 from = qtype == M_READ_OK ? qa : KF_ABSENT, rid = ++node_msgid;
                    hs_set(l, i, 2, rid); hs_set(l, i, 3, (fl & ~(3u << 4)) | (2u << 4)); hs_set(l, i, 4, from);
                    to_svc = true; o_type = M_CAS; o_a = KF_OFFSETS_KEY | from | (i << 16); o_b = rid;
                  } else {
                    rep = true; o_b = hs_get(l, i, 0);
                    if (qtype == M_CAS_OK) { o_type = M_COMMIT_OFFSETS_OK; o_a = 0; } else { o_type = M_ERROR; o_a = qa == 22 ? 30u : qa; }
                    hs_set(l, i, 3, 0);
                  }
                  break;
              }
            } break;
            default: break;
          }
        }
      }
      // poll_ok / list_committed_offsets_ok blocks: payload words allocated in node order, each node writes its own
      if (__ballot(need_words != 0)) {
        u32 excl = 0, total = 0;
        for (u32 s = 0; s < N; s++) { const u32 v = GGET(need_words, s); excl += s < l ? v : 0u; total += v; }
        if (total) {
          const bool fits = n_payload + total <= max_pay;
          if (!fits) flags |= MSIM_FLAG_PAYLOAD_OVERFLOW;
          if (need_words) {
            if (!fits) { rep = false; hs_set(l, done_slot, 3, 0); }   // (the oracle drops the reply with the handler)
            else {
              const u32 fl = hs_get(l, done_slot, 3), nk = KS_NK(fl), s1 = hs_get(l, done_slot, 1);
              u32 pp = n_payload + excl;
              o_a = pp | (need_words << 24);
              if (KS_KIND(fl) == KK_POLL) {
                for (u32 e = 0; e < nk; e++) {
                  const u32 w = g_pay[s1 + e], k_ = w & 7u, o = w >> 8, i0 = o % KF_CHUNK, c = (hs_get(l, done_slot, 6 + (e >> 2)) >> (8u * (e & 3u))) & 0xFFu;
                  const u32 n = c > i0 ? c - i0 : 0u;
                  g_pay[pp++] = k_ | (n << 8) | (o << 16);
                  for (u32 x = 0; x < n; x += 2) g_pay[pp++] = g_log[(size_t)k_ * cap + o + x] | (x + 1 < n ? g_log[(size_t)k_ * cap + o + x + 1] << 16 : 0u);
                }
              } else {
                const u32 ver = hs_get(l, done_slot, 4);
                for (u32 e = 0; e < nk; e++) {
                  const u32 k_ = g_pay[s1 + e] & 7u, c = ver == KF_ABSENT ? 0u : committed_at(k_, ver);
                  g_pay[pp++] = k_ | (c ? (((c - 1u) << 8) | 0x80000000u) : 0u);   // select-keys: only the keys the map has
                }
              }
              hs_set(l, done_slot, 3, 0);
            }
          }
          if (fits) n_payload += total;
        }
      }
      K8_MARK(3)   // [3] = R3 nodes (+ reply payloads)
      wave_lds_fence();   // the service reads the handlers' tables (LDS) after the nodes have written them
      // the lin-kv service (service.clj:31-61 over the chunk keys and "offsets"): one request per round, after the nodes
      if (__ballot(svc_due)) {
        if (svc_due) {
          const uint4 q = cm; deliver_at = INF;
          const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
          s_recv_sv++;
          svc_rep = true; o_dest = qsrc; o_b = qb;
          if (qtype == M_READ) {
            if (qa & KF_OFFSETS_KEY) { if (off_exists) { o_type = M_READ_OK; o_a = off_ver; } else { o_type = M_ERROR; o_a = 20; } }
            else { const u32 c = chunk_count(qa & 7u, qa >> 8); if (c) { o_type = M_READ_OK; o_a = c; } else { o_type = M_ERROR; o_a = 20; } }
          } else if (qtype == M_CAS) {
            if (qa & KF_OFFSETS_KEY) {
              const u32 from = qa & 0xFFFFu, i = (qa >> 16) & 0xFu;
              if (off_exists && from != off_ver) { o_type = M_ERROR; o_a = 22; }   // (from {} never equals a stored map: they are not empty)
              else {
                // the value did not change since it was read (or the key is created): to = (merge-with max from (:offsets body))
                u32 pp = hs_get(qsrc, i, 1); const u32 end = pp + KS_NK(hs_get(qsrc, i, 3)), newver = off_ver + 1u; bool changed = false;
                while (pp < end) {
                  const u32 h = g_pay[pp], k_ = h & 7u, n = (h >> 8) & 0xFFu, o = h >> 16;
                  pp += 1u + (n + 1u) / 2u;
                  if (!n) continue;   // txn-offsets: only keys something was polled from
                  const u32 hi = o + n - 1u, cur = committed_at(k_, off_ver);
                  if (cur == 0 || hi > cur - 1u) { const u32 e = nupd[k_]; g_upd[(size_t)k_ * (cap + 1) + e] = (newver << 16) | hi; nupd[k_] = e + 1u; changed = true; }
                }
                off_exists = 1;
                if (changed) off_ver = newver;
                o_type = M_CAS_OK; o_a = 0;
              }
            } else {
              const u32 k_ = qa & 7u, ch = (qa >> 3) & 63u, from = (qa >> 9) & 31u, msg = qa >> 14;
              const u32 cur = chunk_count(k_, ch);
              if (cur != 0 && cur != from) { o_type = M_ERROR; o_a = 22; }
              else {   // the chunk is what was read, or does not exist (create_if_not_exists, :52-55): it becomes that + [msg]
                const u32 o = ch * KF_CHUNK + (cur ? from : 0u);
                if (o >= cap) { my_flags |= MSIM_FLAG_VALUES_OVERFLOW; o_type = M_ERROR; o_a = 22; }
                else { g_log[(size_t)k_ * cap + o] = msg; klen[k_] = o + 1u; o_type = M_CAS_OK; o_a = 0; }
              }
            }
          } else svc_rep = false;
        }
      }
      wave_lds_fence();

      K8_MARK(4)   // [4] = R3 service
      // COMMIT: ids in lane order (nodes, then the service)
      bool c_arr = false; u32 ca_y = 0, ca_a = 0, ca_b = 0;
      {
        const u32 cnt = (to_svc || rep || svc_rep) ? 1u : 0u;
        const u32 smask = GB(cnt != 0);
        if (__ballot(cnt != 0)) {
          const u32 my_off = __popc(smask & lt);
          if (rep) s_send_cl++; else if (cnt) s_send_sv++;
          // node -> service: the service lane takes them in node order
          u32 ts = GB(to_svc);
          while (__ballot(ts != 0)) {
            const bool on = ts != 0;
            const u32 s = on ? (u32)__builtin_ctz(ts) : 0u; ts &= ts - 1u;
            const u32 ty = GGET(o_type, s), a = GGET(o_a, s), b = GGET(o_b, s), off = GGET(my_off, s);
            if (on && is_svc) arrive(next_id + off, ty, a, b, s);
          }
          // service -> node
          {
            const u32 sv = GB(svc_rep);
            const u32 ty = GGET(o_type, N), a = GGET(o_a, N), b = GGET(o_b, N), d = GGET(o_dest, N), off = GGET(my_off, N);
            if (sv && l == d) arrive(next_id + off, ty, a, b, SVC);
          }
          // node -> its own client: no latency; lost like any other message (net.clj:214)
          if (rep) {
            const u32 id = next_id + my_off;
            if (!(NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss)) { c_arr = true; ca_y = (id << 8) | o_type; ca_a = o_a; ca_b = o_b; }
          }
          next_id += __popc(smask);
        }
        poll();
      }

      K8_MARK(5)   // [5] = COMMIT + polls
      #include "group8_clients.inc"
    K8_MARK(6)   // [6] = R4 clients
    // ---- history rows: nemesis rows, invocations (lane order), completions (lane order) ----
    {
      const u32 imask = GB(inv_row), cmask = GB(cmp_row);
      const u32 ni = __popc(imask);
      const u32 nr = nem_rows + ni + __popc(cmask);
      if (__ballot(alive && nr != 0)) {
        const bool ovf = alive && nr != 0 && n_rows + nr > max_rows;
        if (ovf) { flags |= MSIM_FLAG_ROWS_OVERFLOW; alive = false; }
        const bool wr = alive && nr != 0;
        const u64 tns = (u64)T * 1000ull;
        const u32 tlo = (u32)tns, thi = (u32)(tns >> 32);
        uint4 *const out = reinterpret_cast<uint4 *>(g_rows) + n_rows;   // (no staging: a few 16-byte rows per round; the L2 merges them into lines)
        if (NEM && wr && nem_rows && l == 0) {
          const u32 pk = MSIM_T_INFO | (nem_f << 2) | (MSIM_PROCESS_NEMESIS << 12);
          out[0] = make_uint4(tlo, thi, pk, nem_v1);
          out[1] = make_uint4(tlo, thi | (nem_len2 << 16), pk, nem_v2);
        }
        if (wr && inv_row) out[nem_rows + __popc(imask & lt)] = make_uint4(tlo, thi | (inv_len << 16), inv_packed, inv_value);
        if (wr && cmp_row) out[nem_rows + ni + __popc(cmask & lt)] = make_uint4(tlo, thi | (cmp_len << 16), cmp_packed, cmp_value);
        n_rows = wr ? n_rows + nr : n_rows;
      }
    }
    K8_MARK(7)   // [7] = rows
  }

  // ---- epilogue ----
  u32 t_send_cl = 0, t_send_sv = 0, t_recv_cl = 0, t_recv_sv = 0;
  for (u32 s = 0; s < GS; s++) { t_send_cl += GGET(s_send_cl, s); t_send_sv += GGET(s_send_sv, s); t_recv_cl += GGET(s_recv_cl, s); t_recv_sv += GGET(s_recv_sv, s); }
  for (u32 b = 1; b <= MSIM_FLAG_ARENA_OVERRUN; b <<= 1) if (GB((my_flags & b) != 0)) flags |= b;
  if (real && l == 0) {
    msim_net_stats st;
    st.all_send = (u64)t_send_cl + t_send_sv; st.all_recv = (u64)t_recv_cl + t_recv_sv;
    st.clients_send = t_send_cl; st.clients_recv = t_recv_cl;
    st.servers_send = t_send_sv; st.servers_recv = t_recv_sv;
    p.stats[inst] = st;
    msim_inst_meta m; m.n_rows = n_rows; m.n_payload_words = n_payload; m.flags = flags; m.n_rounds = rounds;
    m.n_events = 0; m.reserved[0] = 0; m.reserved[1] = 0; m.reserved[2] = 0;
#ifdef K8_PROF
    if (grp == 0) { st.all_send = kp[0]; st.all_recv = kp[1]; st.clients_send = kp[2]; st.clients_recv = kp[3]; st.servers_send = kp[4]; st.servers_recv = kp[5]; p.stats[inst] = st;
                    m.reserved[0] = (u32)(kp[6] >> 6); m.reserved[1] = (u32)(kp[7] >> 6); m.n_events = kp_rounds; }
#endif
    p.meta[inst] = m;
  }
}

}  // namespace

// Whether eight clusters per wavefront simulate this configuration (see the header of this file).
bool msim_kafka8_eligible(const msim_config &c) {
  return c.node_program == MSIM_NODE_KAFKA && c.journal_capacity == 0 && c.n_nodes >= 1 && c.n_nodes <= GS - 1u && c.concurrency == c.n_nodes;
}

// Extra per-instance scratch words behind the queues' spill area: the clients' spill, the handlers beyond the LDS ones, and what of the
// LDS queues of kafka_kernel<> does not fit this kernel's RQ slots.
uint64_t msim_kafka8_extra_scratch_words(const msim_config &c) {
  return ((uint64_t)(c.n_nodes + 1) * c.inbox_capacity + (uint64_t)c.n_nodes * K8_CLIENT_CAP) * 4 + (uint64_t)c.n_nodes * (KF_SLOTS - SL) * KSW + 4;
}

hipError_t msim_launch_kafka8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < MSIM_KAFKA8_MIN_CLUSTERS && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;
  K8Params up;
  up.k = kp; up.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  up.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;
  up.client_spill = K8_CLIENT_CAP - CQ;
  up.client_spill_off = kp.spill_off + (uint64_t)(kp.N + 1) * up.node_spill * 4;
  up.xslots_off = (up.client_spill_off + (uint64_t)kp.N * up.client_spill * 4 + 3) & ~3ull;
  size_t off = (size_t)RQ * 64 * 16;
  up.off_cq = (u32)off; off += (size_t)CQ * 64 * 16;
  up.off_slots = (u32)off; off += (size_t)SL * KSW * 64 * 4;
  up.off_cache = (u32)off; off += (size_t)KF_KEYS * 64 * 4;
  up.off_coff = (u32)off; off += (size_t)KF_KEYS * 64 * 4;
  up.off_gen = (u32)off; off += (size_t)8 * 36 * 4;
  up.off_kl = (u32)off; off += (size_t)8 * 16 * 4;
  off = (off + 15) & ~(size_t)15;
  up.off_misc = (u32)off; off += 64 * 4;
  up.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  if (kp.dev_flags & 0x1000u) std::fprintf(stderr, "[kafka8] %u clusters, eight per wavefront, %zu B of LDS per wavefront\n", n, lds);   // developer trace bit
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(g8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 7) / 8), block(64);
  if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((kafka8_kernel<true, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((kafka8_kernel<true, false>), grid, block, lds, st, up); }
  else { if (rnd) hipLaunchKernelGGL((kafka8_kernel<false, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((kafka8_kernel<false, false>), grid, block, lds, st, up); }
  return hipGetLastError();
}
