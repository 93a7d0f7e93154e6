// txn8.hip — EIGHT txn-list-append clusters per wavefront (BASELINE configs[4]: 5 nodes, one worker
// per node, the single-root transactional node over the lin-kv service).
//
// Same program and the same rounds as txn_kernel<> (sim_kernel_txn.inc): node = demo/clojure/single_key_txn.clj:116-180 (read
// "root" from lin-kv -> apply the txn -> cas root with create_if_not_exists -> txn_ok, or error 30 when the cas lost), service =
// service.clj:31-61,141-155,245-263, client = workload/txn_list_append.clj:94-126, generator = [upstream] elle list-append;
// round for round what DESIGN.md §2 and the CPU oracle (oracle/txn_nodes.inc) specify.  What changes is the mapping: a cluster is
// n nodes (each with its client) + the service = 6 endpoints for the BASELINE shape — one lane each of an 8-lane group — and a
// wavefront carries eight clusters.  txn_kernel<> ran one cluster per wavefront (6 live lanes of 64) and was bound by instruction
// issue; here one instruction stream serves eight clusters: what is uniform per CLUSTER lives in VGPRs (equal within a group), a
// "ballot" is the group's 8 bits of the wave ballot, another lane's value comes by `ds_bpermute` within the group, the time
// reduction is three DPP steps.
//
// Scope (engine.hip picks this kernel when all of it holds, else txn_kernel<> runs): n_nodes <= 7, net journal off,
// max-txn-length <= 4 (the default), max-writes-per-key 4, 8, 12 or 16 (the default).
//
// LDS of a wavefront (slot-major: slot s of lane e at [s * 64 + e]): node / service queues (RQ envelopes, the rest spills to HBM:
// inbox_capacity + spill_capacity in all, the oracle's limit), client inboxes (CQ envelopes + HBM spill: 32 in all), the nodes'
// transactions in flight (the first 4 of 8 x 16 B; the others in HBM), per cluster the generator's key pool and the nemesis shuffle:
// 9.5 KiB, so that 16 wavefronts share a CU — this kernel waits for memory (scattered per-cluster pages), and wavefronts to switch
// to are what hides that.  History rows go straight to HBM (a few 16-byte rows per round and cluster; the L2 merges them).  The append log (elements per key) lives in HBM scratch as in txn_kernel<>; a key's row is read
// with independent loads (versions only grow along a row: "visible at version v" is a count, not a search).
#include <hip/hip_runtime.h>

#include "wave_common.h"
#include "log2_table.h"

namespace {

__constant__ u32 t8_log2_q24[257];

constexpr u32 GS = 8u;            // lanes per cluster
constexpr u32 RQ = 3u;            // LDS envelopes per node / service queue
constexpr u32 CQ = 1u;            // LDS envelopes per client inbox
constexpr u32 T8_SLOTS = 8u;      // transactions in flight per node (the oracle's limit) ...
constexpr u32 SL = 4u;            // ... of which in LDS; the others (in use only while clients time out) in HBM
constexpr u32 T8_CLIENT_CAP = 32u;
constexpr int MM = 4;             // micro-ops per transaction (--max-txn-length <= 4, the default)
constexpr u32 V_NIL = 0xFFFFu;
enum { M_WRITE = 14, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR, M_TXN = 23, M_TXN_OK = 24 };
enum { S_GEN3 = 3 };

struct T8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_slots, off_gen, off_misc;   // LDS byte offsets (queues at 0)
  u64 xslots_off;                                        // word offset of the nodes' slots SL .. T8_SLOTS-1 inside the per-instance scratch
  u32 node_spill, client_spill;                          // HBM spill entries per node-or-service queue / client inbox
  u64 client_spill_off;                                  // word offset of the clients' spill area inside the per-instance scratch
  u32 round_limit;
};

__device__ __forceinline__ u32 t8_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = t8_log2_q24[idx], l1 = t8_log2_q24[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}
// min over the 8 lanes of the caller's group, in every lane of it
__device__ __forceinline__ u32 oct_min(u32 v) {
  v = min(v, dpp_mov<0xB1, 0xF, 0xF, false>(v, v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_mov<0x4E, 0xF, 0xF, false>(v, v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_mov<0x141, 0xF, 0xF, false>(v, v));  // row_half_mirror
  return v;
}

template <bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) txn8_kernel(const T8Params tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = tp.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 3, gbase = lane & 56u;
  const u32 N = p.N;
  const bool is_node = l < N, is_svc = l == N;
  const u32 SVC = 2 * N;   // the service's endpoint index
  const u32 inst_raw = blockIdx.x * 8u + grp;
  const bool real = inst_raw < tp.n_inst;
  const u32 inst = real ? inst_raw : tp.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << l) - 1u;
  const u32 all_nodes = (1u << N) - 1u;
  const u32 max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz, mw = p.cfg.max_writes_per_key;
  const u32 round_limit = tp.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  u32 *const g_kv = g_scr;                                           // [max_values][mw]: element | version << 8
  u32 *const g_kvn = g_scr + (size_t)p.cfg.max_values * mw;          // [max_values]
  const u32 qlane = l <= N ? l : 0u;
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)qlane * tp.node_spill;
  uint4 *const my_cspill = reinterpret_cast<uint4 *>(g_scr + tp.client_spill_off) + (size_t)(is_node ? l : 0u) * tp.client_spill;
  const u32 my_spill_cap = l <= N ? tp.node_spill : 0u;

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node / service queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + tp.off_cq) + lane;                      // client inbox
  uint4 *const slots_g = reinterpret_cast<uint4 *>(smem + tp.off_slots) + grp * GS * SL;       // [lane of the group][SL] {client_msg, txn_ref, rpc_id, from | stage << 16 | used << 24}
  uint4 *const xslots = reinterpret_cast<uint4 *>(g_scr + tp.xslots_off);                       // [node][T8_SLOTS - SL]
  // slot i of node nd (LDS for the first SL, HBM beyond: only reached while a node has more than SL transactions in flight)
#define SLOT_PTR(nd_, i_) ((i_) < SL ? slots_g + (nd_) * SL + (i_) : xslots + (nd_) * (T8_SLOTS - SL) + ((i_) - SL))
  u32 *const gen = reinterpret_cast<u32 *>(smem + tp.off_gen) + grp * 36;                       // active[16], next_val[16], next_key
  u32 *const misc = reinterpret_cast<u32 *>(smem + tp.off_misc) + grp * GS;

  for (u32 i = lane; i < 8 * GS * SL; i += 64) reinterpret_cast<uint4 *>(smem + tp.off_slots)[i] = make_uint4(0, 0, 0, 0);
  if (real && is_node) for (u32 i = 0; i < T8_SLOTS - SL; i++) xslots[l * (T8_SLOTS - SL) + i] = make_uint4(0, 0, 0, 0);
  for (u32 i = l; i < 16; i += GS) { gen[i] = i; gen[16 + i] = 1; }
  if (l == 0) gen[32] = p.cfg.key_count;
  if (real) for (u32 i = l; i < p.cfg.max_values; i += GS) g_kvn[i] = 0;
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node / service state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, node_msgid = 0, part = 0, root = V_NIL;
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_value = 0, process = l, m_value = 0, cin_n = 0, csp_n = 0;
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, nem_next = 0, nem_j = 0;
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
      else lat = (u32)(((u64)lat_mean * t8_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
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
      for (u32 i = 0; i < sp_n; i++) {
        const uint2 kk = *reinterpret_cast<const uint2 *>(&my_spill[i]);
        if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; in_spill = true; }
      }
      uint4 e;
      if (in_spill) { e = my_spill[best]; sp_n--; if (best != sp_n) my_spill[best] = my_spill[sp_n]; }
      else { e = my_q[best * 64u]; in_n--; if (best != in_n) my_q[best * 64u] = my_q[in_n * 64u]; }
      try_commit(e);
    }
  };
  // What a node needs from HBM to complete a transaction (its micro-ops, the element counts and rows of the keys it reads) and
  // what the service needs to apply a cas (the micro-ops, the counts of the keys appended to): two dependent batches of
  // independent loads (three until round 6: the rows waited for the counts), shared by every lane of the wavefront that needs any this round, kept in registers for the completion.
  // (Issuing them rounds ahead — the plan is known when the envelope is committed to — was tried and bought nothing: with two
  // wavefronts per SIMD the round trips of ~10^4 clusters' scattered pages stay longer than a round of other work.)
  bool pf_valid = false, pf_done = false, pf_cas = false; u32 pf_off0 = 0, pf_n = 0, pf_from = V_NIL;
  u32 pf_wv[MM], pf_cn[MM], pf_vis[MM]; uint4 pf_el[MM];
#pragma unroll
  for (int j = 0; j < MM; j++) { pf_wv[j] = 0; pf_cn[j] = 0; pf_vis[j] = 0; pf_el[j] = make_uint4(0, 0, 0, 0); }
  u32 pf_stage = 0;   // batches done for the plan: 1 micro-ops, 2 element counts, 3 rows (complete)
#define T8_STAGE_W(go_) do {                                                                                                        \
    const bool tf_g = (go_);                                                                                                        \
    /* (loads under the lanes' own predicate straight into the plan's registers: a select against the old value would wait for them) */ \
    if (tf_g) { _Pragma("unroll") for (int j = 0; j < MM; j++) { pf_wv[j] = pf_cas ? 0u : 1u; if ((u32)j < pf_n) pf_wv[j] = g_pay[pf_off0 + (u32)j]; } } \
    pf_stage = tf_g ? 1u : pf_stage;                                                                                                \
  } while (0)
  /* element counts — of the keys a completing node reads (none if it started from nil), of the keys the service appends to — and, in the
     same batch, the rows of the keys a completing node reads -> number of visible elements and their values (16 bytes).  A row is
     requested whole before its count is known (what lies behind the count is masked by it): one round trip instead of two. */
#define T8_STAGE_CR(go_) do {                                                                                                       \
    const bool tf_g = (go_);                                                                                                        \
    uint4 tf_r[MM][4];                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < MM; j++) {                                                                                \
      const bool tf_want = tf_g && (u32)j < pf_n && (pf_cas ? (pf_wv[j] & 1u) != 0 : (!(pf_wv[j] & 1u) && pf_from != V_NIL));       \
      const u32 tf_k = (pf_wv[j] >> 1) & 0x7FFFu;                                                                                   \
      if (tf_g) pf_cn[j] = 0u;                                                                                                      \
      if (tf_want) pf_cn[j] = g_kvn[tf_k];                                                                                          \
      _Pragma("unroll") for (int qq = 0; qq < 4; qq++) tf_r[j][qq] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); \
      if (tf_want && pf_done) {                                                                                                     \
        const uint4 *tf_row = reinterpret_cast<const uint4 *>(g_kv + (size_t)tf_k * mw);                                            \
        _Pragma("unroll") for (int qq = 0; qq < 4; qq++) if (4u * (u32)qq < mw) tf_r[j][qq] = tf_row[qq];                           \
      }                                                                                                                             \
    }                                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < MM; j++) {                                                                                \
      if (tf_g) { pf_vis[j] = 0; pf_el[j] = make_uint4(0, 0, 0, 0); }                                                               \
      if (tf_g && pf_done && (u32)j < pf_n && !(pf_wv[j] & 1u) && pf_cn[j] != 0) {                                                  \
        u32 tf_v = 0, tf_e[4];                                                                                                      \
        _Pragma("unroll") for (int qq = 0; qq < 4; qq++) {                                                                          \
          const u32 tf_b = 4u * (u32)qq;                                                                                            \
          tf_v += (tf_b + 0 < pf_cn[j] && (tf_r[j][qq].x >> 8) <= pf_from) ? 1u : 0u; tf_v += (tf_b + 1 < pf_cn[j] && (tf_r[j][qq].y >> 8) <= pf_from) ? 1u : 0u; \
          tf_v += (tf_b + 2 < pf_cn[j] && (tf_r[j][qq].z >> 8) <= pf_from) ? 1u : 0u; tf_v += (tf_b + 3 < pf_cn[j] && (tf_r[j][qq].w >> 8) <= pf_from) ? 1u : 0u; \
          tf_e[qq] = (tf_r[j][qq].x & 0xFFu) | ((tf_r[j][qq].y & 0xFFu) << 8) | ((tf_r[j][qq].z & 0xFFu) << 16) | ((tf_r[j][qq].w & 0xFFu) << 24); \
        }                                                                                                                           \
        pf_vis[j] = tf_v; pf_el[j] = make_uint4(tf_e[0], tf_e[1], tf_e[2], tf_e[3]);                                                \
      }                                                                                                                             \
    }                                                                                                                               \
    pf_stage = tf_g ? 3u : pf_stage;                                                                                                \
  } while (0)
#ifdef T8_PROF
  u64 pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u32 wave_rounds = 0;
  u64 tprev = __builtin_readcyclecounter();
#define T8_MARK(i) { const u64 now_ = __builtin_readcyclecounter(); pacc[i] += now_ - tprev; tprev = now_; }
#else
#define T8_MARK(i)
#endif
  for (;;) {
    if (!__ballot(alive)) break;
#ifdef T8_PROF
    wave_rounds++;
#endif
    const u32 busy_mask = GB(busy);

    // ---- time-free phase transitions ----
    if (__ballot(alive && !(phase == PH_MAIN && ((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff))))) {
      for (;;) {
        bool ch = false;
        if (alive) {
          if (phase == PH_INIT_WAIT && !busy_mask) { phase = PH_MAIN_START; ch = true; }
          if (phase == PH_MAIN_START) { cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; nem_next = T; next_msg_id = 0; loss_on = 1; phase = PH_MAIN; ch = true; }
          if (phase == PH_MAIN && !((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff)) && !(rate == 0 && T < cutoff)) { phase = PH_DRAIN; ch = true; }
          if (phase == PH_DRAIN && !busy_mask) { phase = PH_DONE; ch = true; }   // no final phase (txn_list_append.clj:142)
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
    if (phase == PH_INIT) due = T;
    else if (phase == PH_MAIN) {
      if (nem_live) due = max(nem_next, T);
      if (gen_live && free_mask) due = min(due, max(gen_next, T));
      if (rate == 0 && !nem_live) due = min(due, cutoff);
    }
    bool timeout_round = false;
    {
      const bool none_due = GB(deliver_at <= T) == 0;
      const bool jump = alive && due > T && none_due;
      if (__ballot(jump)) {
        u32 k = deliver_at == INF ? INF : deliver_at * 2;
        if (busy) k = min(k, timeout_at * 2 + 1);
        u32 km = oct_min(k);
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

    auto complete = [&](u32 type, u32 err, u32 ref) {
      busy = false;
      if (kind != K_OP) { if (type != MSIM_T_OK) my_flags |= MSIM_FLAG_ROUND_LIMIT; return; }
      cmp_row = true; cmp_packed = type | (MSIM_F_TXN << 2) | (err << 7) | (process << 12);
      cmp_value = ref & 0xFFFFFFu; cmp_len = ref >> 24;
      if (type == MSIM_T_INFO) process += N;  // crashed process; the Reusable client itself lives on
    };
    // the client's recv! consumes one envelope (client.clj:94-107)
    auto client_deliver = [&](u32 qtype, u32 qa, u32 qb) {
      s_recv_cl++;
      if (busy && qb == want) {
        if (qtype == M_TXN_OK) complete(MSIM_T_OK, 0, qa);
        else if (qtype == M_ERROR)
          complete(MSIM_T_FAIL, qa == 11 ? MSIM_ERR_TEMPORARILY_UNAVAILABLE : qa == 20 ? MSIM_ERR_KEY_DOES_NOT_EXIST : qa == 30 ? MSIM_ERR_TXN_CONFLICT : MSIM_ERR_PRECONDITION_FAILED, c_value);
        else complete(MSIM_T_OK, 0, c_value);  // init_ok
      }
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) complete(MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_value);
    }
    bool normal = alive && !timeout_round;   // this cluster runs R1-R4 in this wave-round
    T8_MARK(0)
    if (__ballot(normal)) {
      // ---- R1: scheduler ----
      const bool act = normal && due <= T;
      if (__ballot(act && phase == PH_INIT)) {
        if (act && phase == PH_INIT) { if (is_node) { mark = true; kind = K_INIT; } phase = PH_INIT_WAIT; }
      }
      if (NEM) {
        const bool nem_act = act && phase == PH_MAIN && nem_live && nem_next <= T;
        if (__ballot(nem_act)) {   // flip-flop start/stop (nemesis.clj:10-16 + [upstream] partition package)
          const u32 j = nem_j;
          const u32 spec = scale32(draw32(key, S_NEM_SPEC, j), 4);
          const bool start = nem_act && (j & 1) == 0;
          if (nem_act) { nem_j++; nem_rows = 2; }
          if (__ballot(start)) {
            misc[l] = l;
            wave_lds_fence();
            if (start && l == 0 && spec != MSIM_SPEC_ONE) {
              for (u32 i = N - 1; i >= 1; i--) {
                const u32 kk = scale32(draw32(key, S_NEM_SHUFFLE, ((u64)j << 16) | i), i + 1);
                const u32 t = misc[i]; misc[i] = misc[kk]; misc[kk] = t;
              }
            }
            wave_lds_fence();
            u32 my_part = 0;
            if (start && is_node) {
              if (spec == MSIM_SPEC_ONE) {
                const u32 loner = scale32(draw32(key, S_NEM_PICK, j), N);
                my_part = l == loner ? (all_nodes & ~(1u << loner)) : (1u << loner);
              } else if (spec == MSIM_SPEC_MAJORITY || spec == MSIM_SPEC_MINORITY_THIRD) {
                const u32 cnt = spec == MSIM_SPEC_MAJORITY ? N / 2 : (N - 1) / 3;
                u32 comp = 0;
                for (u32 i = 0; i < cnt; i++) comp |= 1u << misc[i];
                my_part = ((comp >> l) & 1) ? (all_nodes & ~comp) : comp;
              } else {
                const u32 m = N / 2 + 1;
                u32 pos = 0;
                for (u32 i = 0; i < N; i++) if (misc[i] == l) pos = i;
                const u32 i0 = (pos + N - (m / 2) % N) % N;
                u32 vis = 0;
                for (u32 kk = 0; kk < m; kk++) vis |= 1u << misc[(i0 + kk) % N];
                my_part = all_nodes & ~vis;
              }
            }
            if (start) {
              part |= my_part;
              const u32 words = N * MSIM_MASK_WORDS;
              u32 off = 0;
              if (n_payload + words > max_pay) flags |= MSIM_FLAG_PAYLOAD_OVERFLOW;
              else {
                off = n_payload; n_payload += words;
                if (is_node) { g_pay[off + l * 4] = part; g_pay[off + l * 4 + 1] = 0; g_pay[off + l * 4 + 2] = 0; g_pay[off + l * 4 + 3] = 0; }
              }
              nem_f = MSIM_F_START_PARTITION; nem_v1 = spec; nem_v2 = off; nem_len2 = words;
            }
          }
          if (nem_act && (j & 1) != 0) {
            part = 0;
            nem_f = MSIM_F_STOP_PARTITION; nem_v1 = MSIM_NO_VALUE; nem_v2 = MSIM_NO_VALUE; nem_len2 = 0;
          }
          if (nem_act) nem_next = T + __umulhi(draw32(key, S_NEM_STAGGER, j), p.nem_period2_us);
        }
      }
      {
        const bool gen_on = act && phase == PH_MAIN && gen_live && gen_next <= T && free_mask != 0;
        if (__ballot(gen_on)) {
          const u32 nfree = __popc(free_mask);
          const u32 kk = gen_k;
          const u64 h = draw64(key, S_GEN, kk);
          const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          const u32 pick = scale32(r_lo, nfree);
          const bool sel = gen_on && is_node && !busy && (u32)__popc(free_mask & lt) == pick;
          // the transaction ([upstream] elle list-append gen): lane 0 of the cluster writes the micro-ops and owns the key pool
          const u32 n_mops = 1 + scale32((u32)(draw64(key, S_GEN2, kk) >> 32), p.cfg.max_txn_length);
          u32 bad = 0;
          if (gen_on && n_payload + n_mops > max_pay) bad = MSIM_FLAG_PAYLOAD_OVERFLOW;
          else if (gen_on && l == 0) {
            const u32 kc = p.cfg.key_count;
            for (u32 j = 0; j < n_mops; j++) {
              const u64 h3 = draw64(key, S_GEN3, (u64)kk * 8 + j);
              const u32 x = scale32((u32)(h3 >> 32), (1u << kc) - 1) + 1;
              const u32 ki = 31 - (u32)__clz((int)x);
              const u32 k = gen[ki];
              if (h3 & 1) {
                const u32 v = gen[16 + ki];
                gen[16 + ki] = v + 1;
                g_pay[n_payload + j] = 1u | (k << 1) | (v << 16);
                if (v + 1 > mw) {
                  const u32 nk = gen[32];
                  if (nk >= p.cfg.max_values) { bad = MSIM_FLAG_VALUES_OVERFLOW; break; }
                  gen[ki] = nk; gen[32] = nk + 1; gen[16 + ki] = 1;
                }
              } else g_pay[n_payload + j] = (k << 1) | (0xFFu << 16);
            }
          }
          bad = GGET(bad, 0);
          if (gen_on && bad) { flags |= bad; phase = PH_DONE; alive = false; normal = false; }
          else if (gen_on) {
            if (sel) { mark = true; kind = K_OP; m_value = n_payload | (n_mops << 24); }
            n_payload += n_mops;
            gen_k++;
            gen_next = T + __umulhi(r_hi, p.gen_period2_us);
          }
        }
      }

      T8_MARK(1)
      // ---- R2: marked clients invoke; the request goes to this lane's own node ----
      if (__ballot(mark && normal)) {
        const bool inv = mark && normal;
        const u32 inv_mask = GB(inv);
        if (inv) {
          mark = false; busy = true;
          u32 rq_type, rq_a = 0;
          if (kind == K_INIT) { rq_type = M_INIT; next_msg_id = 0; }
          else {
            c_value = m_value;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (MSIM_F_TXN << 2) | (process << 12); inv_value = c_value & 0xFFFFFFu; inv_len = c_value >> 24;
            rq_type = M_TXN; rq_a = c_value;
          }
          want = ++next_msg_id;
          timeout_at = T + (kind == K_OP ? p.cfg.client_timeout_ms : 10000u) * 1000u;
          s_send_cl++;
          arrive(next_id + __popc(inv_mask & lt), rq_type, rq_a, want, N + l);
        }
        next_id += __popc(inv_mask);
        poll();
      }

      T8_MARK(2)
      // ---- R3: one input per node, then one for the service (endpoint order) ----
      // Decisions first (registers and LDS only), then the HBM reads of every lane that needs any — a node completing a transaction,
      // the service applying a cas — in three batches of independent loads: the micro-ops, the keys' element counts, the rows.
      bool to_svc = false, rep = false, svc_rep = false;   // node -> service, node -> own client, service -> node
      u32 o_type = 0, o_a = 0, o_b = 0, o_dest = 0, need_words = 0, done_slot = 0;
      bool do_done = false, do_cas = false; u32 m_off0 = 0, m_n = 0, m_from = V_NIL, cas_base = 0;
      const bool take = normal && l <= N && deliver_at <= T;
      if (take) {
        const uint4 q = cm; deliver_at = INF;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N && qsrc < SVC) s_recv_cl++; else s_recv_sv++;
        if (is_node) {
          switch (qtype) {
            case M_INIT: rep = true; o_type = M_INIT_OK; o_b = qb; break;
            case M_TXN: {
              u32 i = 0; while (i < T8_SLOTS && (SLOT_PTR(l, i)->w >> 24)) i++;
              if (i == T8_SLOTS) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; break; }
              const u32 rid = ++node_msgid;
              *SLOT_PTR(l, i) = make_uint4(qb, qa, rid, (1u << 16) | (1u << 24));
              to_svc = true; o_type = M_READ; o_a = 0; o_b = rid;
            } break;
            case M_READ_OK: case M_CAS_OK: case M_ERROR: {
              u32 i = 0;
              while (i < T8_SLOTS) { const uint4 s = *SLOT_PTR(l, i); if ((s.w >> 24) && s.z == qb) break; i++; }
              if (i == T8_SLOTS) break;  // handle-reply!: no such rpc
              uint4 s = *SLOT_PTR(l, i);
              if (((s.w >> 16) & 0xFF) == 1) {
                u32 from;
                if (qtype == M_READ_OK) from = qa;
                else if (qtype == M_ERROR && qa == 20) from = V_NIL;
                else { rep = true; o_type = M_ERROR; o_a = qa; o_b = s.x; *SLOT_PTR(l, i) = make_uint4(0, 0, 0, 0); break; }
                const u32 rid = ++node_msgid;
                s.z = rid; s.w = from | (2u << 16) | (1u << 24);
                *SLOT_PTR(l, i) = s;
                to_svc = true; o_type = M_CAS; o_a = from | (i << 16); o_b = rid;
              } else {
                rep = true; o_b = s.x;
                if (qtype == M_CAS_OK) {  // the completed transaction goes into the payload area (sized and written below)
                  o_type = M_TXN_OK; done_slot = i; do_done = true;
                  m_off0 = s.y & 0xFFFFFFu; m_n = s.y >> 24; m_from = s.w & 0xFFFFu;
                } else { o_type = M_ERROR; o_a = qa == 22 ? 30u : qa; *SLOT_PTR(l, i) = make_uint4(0, 0, 0, 0); }
              }
            } break;
            default: break;
          }
        } else {  // the lin-kv service (service.clj:31-61 over the key "root")
          svc_rep = true; o_dest = qsrc; o_b = qb;
          if (qtype == M_READ) {
            if (root == V_NIL) { o_type = M_ERROR; o_a = 20; } else { o_type = M_READ_OK; o_a = root; }
          } else {  // cas with create_if_not_exists
            const u32 from = qa & 0xFFFFu, i = qa >> 16;
            if (root != V_NIL && root != from) { o_type = M_ERROR; o_a = 22; }
            else {
              cas_base = root == V_NIL ? 0u : root;
              const u32 ref = SLOT_PTR(qsrc, i)->y;
              m_off0 = ref & 0xFFFFFFu; m_n = ref >> 24; do_cas = true;
              o_type = M_CAS_OK; o_a = 0;
            }
          }
        }
      }
      T8_MARK(3)
      if (__ballot(do_done || do_cas)) {
        const bool have = pf_valid && pf_off0 == m_off0 && pf_n == m_n && ((do_done && pf_done && pf_from == m_from) || (do_cas && pf_cas));
        const bool mem = do_done || do_cas;
        if (mem && !have) { pf_valid = true; pf_done = do_done; pf_cas = do_cas; pf_off0 = m_off0; pf_n = m_n; pf_from = m_from; pf_stage = 0; }
        if (__ballot(mem && pf_stage < 1u)) T8_STAGE_W(mem && pf_stage < 1u);
#ifdef T8_FINE
        T8_MARK(5)
#endif
        if (__ballot(mem && pf_stage < 3u)) T8_STAGE_CR(mem && pf_stage < 3u);
#ifdef T8_FINE
        T8_MARK(6)
#endif
        if (do_done) {   // size of the completed form
#pragma unroll
          for (int j = 0; j < MM; j++) if ((u32)j < m_n) {
            need_words++;
            if (!(pf_wv[j] & 1u)) {
              u32 len = pf_vis[j];
#pragma unroll
              for (int e = 0; e < j; e++) if ((pf_wv[e] & 1u) && ((pf_wv[e] >> 1) & 0x7FFFu) == ((pf_wv[j] >> 1) & 0x7FFFu)) len++;
              need_words += (len + 3) / 4;
            }
          }
        }
        if (do_cas) {   // the service appends: element | version << 8 at the end of each key's row
          u32 na = 0;
#pragma unroll
          for (int j = 0; j < MM; j++) na += ((u32)j < m_n) ? (pf_wv[j] & 1u) : 0u;
#pragma unroll
          for (int j = 0; j < MM; j++) if ((u32)j < m_n && (pf_wv[j] & 1u)) {
            const u32 w = pf_wv[j], k = (w >> 1) & 0x7FFFu;
            u32 c = pf_cn[j];   // + this transaction's earlier appends to the same key
#pragma unroll
            for (int e = 0; e < j; e++) if ((pf_wv[e] & 1u) && ((pf_wv[e] >> 1) & 0x7FFFu) == k) c++;
            g_kv[(size_t)k * mw + c] = ((w >> 16) & 0xFFu) | ((cas_base + na) << 8); g_kvn[k] = c + 1;
          }
          root = cas_base + na;
        }
      }

#ifdef T8_FINE
      T8_MARK(7)
#endif
      if (take) pf_valid = false;   // (the registers stay as they are for the completion below)

      // completed transactions: payload words allocated in node order, each node writes its own (from the registers filled above)
      if (__ballot(need_words != 0)) {
        u32 excl = 0, total = 0;
        for (u32 s = 0; s < N; s++) { const u32 v = GGET(need_words, s); excl += s < l ? v : 0u; total += v; }
        if (total) {
          if (n_payload + total > max_pay) { flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; if (need_words) { o_a = 0; *SLOT_PTR(l, done_slot) = make_uint4(0, 0, 0, 0); } }
          else {
            if (need_words) {
              u32 pp = n_payload + excl;
              o_a = pp | (need_words << 24);
#pragma unroll
              for (int j = 0; j < MM; j++) if ((u32)j < m_n) {
                const u32 w = pf_wv[j], k = (w >> 1) & 0x7FFFu;
                if (w & 1) { g_pay[pp++] = w; continue; }
                const u32 vis = pf_vis[j];
                u32 e = 0, acc = 0;
                const u32 hdr = pp++;
                const u32 ew[4] = {pf_el[j].x, pf_el[j].y, pf_el[j].z, pf_el[j].w};
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {   // the visible prefix of the row, 4 elements per word as the payload packs them
                  if (4u * (u32)qq + 4u <= vis) { g_pay[pp++] = ew[qq]; e += 4; }
                  else if (4u * (u32)qq < vis) { const u32 r = vis - 4u * (u32)qq; acc = ew[qq] & ((1u << (8u * r)) - 1u); e += r; }
                }
#pragma unroll
                for (int i = 0; i < j; i++) { const u32 wi = pf_wv[i];
                  if ((wi & 1) && ((wi >> 1) & 0x7FFFu) == k) { acc |= ((wi >> 16) & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } } }
                if (e & 3) g_pay[pp++] = acc;
                g_pay[hdr] = (k << 1) | ((e ? e : 0xFFu) << 16);  // a key without elements reads nil
              }
              *SLOT_PTR(l, done_slot) = make_uint4(0, 0, 0, 0);
            }
            n_payload += total;
          }
        }
      }


      T8_MARK(4)
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

      T8_MARK(5)
      // ---- R4: the clients' recv! loops (client.clj:94-107) ----
      if (__ballot(c_arr || (busy && (cin_n | csp_n) != 0))) {
        for (;;) {
          const bool stale = normal && busy && (cin_n | csp_n) != 0;
          const bool fresh = normal && !stale && busy && c_arr;
          if (!__ballot(stale || fresh)) break;
          if (stale) {
            u32 best = 0; bool in_spill = false;
            uint2 bk = make_uint2(INF, INF);
            for (u32 i = 0; i < cin_n; i++) {
              const uint2 kk = *reinterpret_cast<const uint2 *>(&my_cq[i * 64u]);
              if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; }
            }
            for (u32 i = 0; i < csp_n; i++) {
              const uint2 kk = *reinterpret_cast<const uint2 *>(&my_cspill[i]);
              if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; in_spill = true; }
            }
            uint4 e;
            if (in_spill) { e = my_cspill[best]; csp_n--; if (best != csp_n) my_cspill[best] = my_cspill[csp_n]; }
            else { e = my_cq[best * 64u]; cin_n--; if (best != cin_n) my_cq[best * 64u] = my_cq[cin_n * 64u]; }
            client_deliver(e.y & 0xFFu, e.z, e.w & 0xFFFFFFu);
          } else if (fresh) {
            c_arr = false;
            client_deliver(ca_y & 0xFFu, ca_a, ca_b);
          }
        }
        if (c_arr && normal) {  // nobody is in recv!: the envelope waits for the next RPC (and is skipped there as stale)
          const uint4 e = make_uint4(T, ca_y, ca_a, ca_b | (l << 24));
          if (cin_n < CQ) { my_cq[cin_n * 64u] = e; cin_n++; }
          else if (csp_n < tp.client_spill) my_cspill[csp_n++] = e;
          else my_flags |= MSIM_FLAG_INBOX_OVERFLOW;
        }
      }
    }

    T8_MARK(6)
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
        const u32 new_n = wr ? n_rows + nr : n_rows;
        n_rows = new_n;
      }
    }
    T8_MARK(7)
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
#ifdef T8_PROF   // developer build (tools/variant_lib.sh t8prof txn8.hip -DT8_PROF): cycle counters of the round's sections in the meta of the wavefront's first three clusters
    if (grp == 0) { m.n_events = (u32)(pacc[0] >> 6); m.reserved[0] = (u32)(pacc[1] >> 6); m.reserved[1] = (u32)(pacc[2] >> 6); m.reserved[2] = (u32)(pacc[3] >> 6); }
    if (grp == 1) { m.n_events = (u32)(pacc[4] >> 6); m.reserved[0] = (u32)(pacc[5] >> 6); m.reserved[1] = (u32)(pacc[6] >> 6); m.reserved[2] = (u32)(pacc[7] >> 6); }
    if (grp == 2) { m.n_events = wave_rounds; }
#endif
    p.meta[inst] = m;
  }
}

}  // namespace

// Whether eight clusters per wavefront simulate this configuration (see the header of this file).
bool msim_txn8_eligible(const msim_config &c) {
  return c.node_program == MSIM_NODE_TXN_SINGLE_KEY && c.journal_capacity == 0 && c.n_nodes >= 1 && c.n_nodes <= GS - 1u && c.concurrency == c.n_nodes &&
         c.max_txn_length <= (uint32_t)MM && c.max_writes_per_key % 4 == 0 && c.max_writes_per_key <= 16;   // micro-ops and rows in registers
}

// Extra per-instance scratch words behind the queues' spill area: the clients' spill, and what of the LDS queues of txn_kernel<>
// does not fit this kernel's RQ slots.
uint64_t msim_txn8_extra_scratch_words(const msim_config &c) {
  return ((uint64_t)(c.n_nodes + 1) * c.inbox_capacity + (uint64_t)c.n_nodes * T8_CLIENT_CAP + (uint64_t)c.n_nodes * (T8_SLOTS - SL)) * 4;
}

hipError_t msim_launch_txn8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  T8Params tp;
  tp.k = kp; tp.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  tp.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;
  tp.client_spill = T8_CLIENT_CAP - CQ;
  tp.client_spill_off = kp.spill_off + (uint64_t)(kp.N + 1) * tp.node_spill * 4;
  size_t off = (size_t)RQ * 64 * 16;
  tp.off_cq = (u32)off; off += (size_t)CQ * 64 * 16;
  tp.off_slots = (u32)off; off += (size_t)8 * GS * SL * 16;
  tp.off_gen = (u32)off; off += (size_t)8 * 36 * 4;
  off = (off + 15) & ~(size_t)15;
  tp.off_misc = (u32)off; off += 64 * 4;
  tp.xslots_off = tp.client_spill_off + (uint64_t)kp.N * tp.client_spill * 4;
  tp.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(t8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 7) / 8), block(64);
  if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((txn8_kernel<true, true>), grid, block, lds, st, tp); else hipLaunchKernelGGL((txn8_kernel<true, false>), grid, block, lds, st, tp); }
  else { if (rnd) hipLaunchKernelGGL((txn8_kernel<false, true>), grid, block, lds, st, tp); else hipLaunchKernelGGL((txn8_kernel<false, false>), grid, block, lds, st, tp); }
  return hipGetLastError();
}
