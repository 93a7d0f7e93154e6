// mk8.hip — EIGHT clusters of the multi-key txn-list-append node per wavefront (BASELINE configs[4] over
// demo/js/multi_key_txn.js — the flat key -> thunk form of the architecture of demo/ruby/datomic_list_append.rb, which core.clj:113-114 runs and
// which is a different program (a persistent hash tree: dt8.hip / sim_kernel_dt.inc).
//
// Same program and the same rounds as mk_kernel<> (sim_kernel_mk.inc; specification: oracle/mk_nodes.inc): node =
// demo/js/multi_key_txn.js:1-246 (immutable thunks in lww-kv, one root map key -> thunk id in lin-kv; getState / applyTxn / writeThunks /
// casRoot, retry from a fresh root when the cas is lost), services = lin-kv service.clj:31-61,141-155 and lww-kv service.clj:214-243 over
// :65-114, client = workload/txn_list_append.clj:94-126, generator = [upstream] elle list-append.  What changes is the mapping, as in
// txn8.hip: a cluster is n nodes (each with its client) + lin-kv + lww-kv = n + 2 <= 8 endpoints, one lane each of an 8-lane group, and a
// wavefront carries eight clusters.  mk_kernel<> runs one cluster per wavefront — 7 live lanes of 64 — and is bound by instruction issue
// (780 instructions per simulated message; giving it 16 wavefronts per CU instead of 7 bought 15 %): here one instruction stream serves
// eight clusters.  What is uniform per CLUSTER lives in VGPRs (equal within a group), a "ballot" is the group's 8 bits of the wave ballot,
// another lane's value comes by ds_bpermute within the group, the time reduction is three DPP steps.
//
// Scope (engine.hip picks this kernel when all of it holds, else mk_kernel<> runs): n_nodes <= 6, one worker per node, net journal off,
// max-txn-length <= 4 (the default: a transaction touches at most 4 keys).
//
// LDS of a wavefront: node / service queues and client inboxes slot-major (slot s of lane e at [s * 64 + e]; RQ / CQ envelopes, the rest
// spills to HBM), per cluster the nodes' first MK8_SL transaction slots (the others in HBM scratch, in use only while clients time out), a
// round's outgoing messages per node, the generator's key pool and the nemesis shuffle.  History rows go straight to HBM.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "wave_common.h"
#include "log2_table.h"

namespace {

__constant__ u32 m8_log2_q24[257];

constexpr u32 GS = 8u;            // lanes per cluster
#ifndef M8_RQ
#define M8_RQ 8u
#endif
#ifndef M8_SL_N
#define M8_SL_N 1u
#endif
constexpr u32 RQ = M8_RQ;            // LDS envelopes per node / service queue (the services take every RPC of the cluster: a scan of spilled envelopes is a round trip per batch)
constexpr u32 CQ = 1u;            // LDS envelopes per client inbox
constexpr u32 M8_SLOTS = 8u;      // transactions in flight per node (the oracle's limit) ...
constexpr u32 M8_SL = M8_SL_N;         // ... of which in LDS (a second one only while a client has timed out; the HBM slots lie where mk_kernel<> keeps its own)
constexpr u32 M8_CLIENT_CAP = 32u;
constexpr u32 KEYS = 4u;          // distinct keys per transaction (--max-txn-length <= 4)
constexpr u32 MKW = 10u + 9u * KEYS;
constexpr u32 V_NIL = 0xFFFFu, MK_NONE = 0xFFFFFFFFu;
enum { M_WRITE = 14, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR, M_TXN = 23, M_TXN_OK = 24 };
enum { S_GEN3 = 3 };
enum { SK_HDR = 0 /* used | stage << 8: 1 thunk reads, 2 thunk writes, 3 cas, 4 root read */, SK_NK = 1, SK_NSTATE = 2, SK_NNEW = 3, SK_RDOUT = 4, SK_WROUT = 5,
       SK_CMSG = 6, SK_REF = 7, SK_RV = 8, SK_RPC = 9, SK_KEY = 10, SK_WR = SK_KEY + KEYS, SK_FA = SK_WR + KEYS, SK_SORD = SK_FA + KEYS, SK_NORD = SK_SORD + KEYS,
       SK_RDRPC = SK_NORD + KEYS, SK_RDTID = SK_RDRPC + KEYS, SK_WRRPC = SK_RDTID + KEYS, SK_WRTID = SK_WRRPC + KEYS };
enum { D_LIN = 0, D_LWW = 1 };

struct M8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_slots, off_mout, off_gen, off_misc;   // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;                          // HBM spill entries per node-or-service queue / client inbox
  u64 client_spill_off;                                  // word offset of the clients' spill area inside the per-instance scratch
  u32 round_limit;
};

__device__ __forceinline__ u32 m8_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = m8_log2_q24[idx], l1 = m8_log2_q24[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}
// min over the 8 lanes of the caller's group, in every lane of it
__device__ __forceinline__ u32 m8_oct_min(u32 v) {
  v = min(v, dpp_mov<0xB1, 0xF, 0xF, false>(v, v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_mov<0x4E, 0xF, 0xF, false>(v, v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_mov<0x141, 0xF, 0xF, false>(v, v));  // row_half_mirror
  return v;
}

template <bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) mk8_kernel(const M8Params tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = tp.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 3, gbase = lane & 56u;
  const u32 N = p.N;
  const bool is_node = l < N, is_lin = l == N;
  const u32 LIN = 2 * N;   // endpoint index of lin-kv (lane N of the group); lww-kv is LIN + 1 (lane N + 1)
  const u32 inst_raw = blockIdx.x * 8u + grp;
  const bool real = inst_raw < tp.n_inst;
  const u32 inst = real ? inst_raw : tp.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << l) - 1u;
  const u32 all_nodes = (1u << N) - 1u;
  const u32 max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz, mw = p.cfg.max_writes_per_key, mw1 = mw + 1u, mv = p.cfg.max_values;
  const u32 TC = p.mk_tcap, CC = p.mk_ccap;   // thunks a node may create; slots of its thunk cache (a power of two)
  const u32 round_limit = tp.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  // the per-instance scratch of mk_kernel<> (sim_kernel_mk.inc), same layout
  u32 *const g_kv = g_scr;                                           // [max_values][mw]: element | version << 8
  u32 *const g_kvn = g_kv + (size_t)mv * mw;                         // [max_values]
  u32 *const g_pos = g_kvn + mv;                                     // [max_values] position of the key in the root map
  u32 *const g_first = g_pos + mv;                                   // [max_values] version at which it entered (MK_NONE: never)
  u32 *const g_updn = g_first + mv;                                  // [max_values] thunks committed for the key
  u32 *const g_upd_v = g_updn + mv;                                  // [max_values][mw + 1] their versions
  u32 *const g_upd_t = g_upd_v + (size_t)mv * mw1;                   // [max_values][mw + 1] their ids
  u32 *const g_cache = g_upd_t + (size_t)mv * mw1;                   // [N][CC] the nodes' thunk caches
  unsigned char *const g_rep = reinterpret_cast<unsigned char *>(g_cache + (size_t)N * CC);   // [N][TC] replica holding thunk <node>.<i>
  u32 *const xslots = g_cache + (size_t)N * CC + (((size_t)N * TC + 15u) / 16u) * 4u;         // [N][M8_SLOTS - M8_SL][82]: room for slots of 8 keys; used with MKW
  const u32 qlane = l <= N + 1u ? l : 0u;
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)qlane * tp.node_spill;
  uint4 *const my_cspill = reinterpret_cast<uint4 *>(g_scr + tp.client_spill_off) + (size_t)(is_node ? l : 0u) * tp.client_spill;
  const u32 my_spill_cap = l <= N + 1u ? tp.node_spill : 0u;

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node / service queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + tp.off_cq) + lane;                      // client inbox
  u32 *const slots_g = reinterpret_cast<u32 *>(smem + tp.off_slots) + grp * (N * M8_SL * MKW);    // [node of the group][M8_SL][MKW]
  u32 *const mout_g = reinterpret_cast<u32 *>(smem + tp.off_mout) + grp * (N * KEYS * 3u);        // [node of the group][KEYS][3]: what a node sends to a service this round {type, a, b}
  u32 *const gen = reinterpret_cast<u32 *>(smem + tp.off_gen) + grp * 36;                       // active[16], next_val[16], next_key
  u32 *const misc = reinterpret_cast<u32 *>(smem + tp.off_misc) + grp * GS;
  // slot si of node nd: LDS for the first M8_SL, HBM scratch beyond (a generic pointer: flat loads / stores reach both)
  // (Every handler runs on one or the other through a lambda inlined at both call sites, so that the LDS slots — the ones in use nearly
  //  always — are read and written with ds_ instructions: through a generic pointer they cost flat accesses, which wait for every global
  //  load in flight as well.)
  auto lds_slot = [&](u32 nd, u32 si) -> u32 * { return slots_g + (nd * M8_SL + si) * MKW; };
  auto hbm_slot = [&](u32 nd, u32 si) -> u32 * { return xslots + ((size_t)nd * (M8_SLOTS - M8_SL) + (si - M8_SL)) * MKW; };
#define M8_ON_SLOT(nd_, si_, f_) ((si_) < M8_SL ? f_(lds_slot((nd_), (si_)), (si_)) : f_(hbm_slot((nd_), (si_)), (si_)))
  const u32 my_node = is_node ? l : 0u;
  u32 *const my_cache = g_cache + (size_t)my_node * CC;

  for (u32 i = lane; i < 8 * N * M8_SL * MKW; i += 64) reinterpret_cast<u32 *>(smem + tp.off_slots)[i] = 0;
  if (real && is_node) for (u32 i = 0; i < M8_SLOTS - M8_SL; i++) xslots[((size_t)l * (M8_SLOTS - M8_SL) + i) * MKW + SK_HDR] = 0;
  for (u32 i = l; i < 16; i += GS) { gen[i] = i; gen[16 + i] = 1; }
  if (l == 0) gen[32] = p.cfg.key_count;
  if (real) {
    for (u32 i = l; i < mv; i += GS) { g_kvn[i] = 0; g_updn[i] = 0; g_first[i] = MK_NONE; g_pos[i] = MK_NONE; }
    for (u32 r = 0; r < N; r++) for (u32 i = l; i < N * (TC >> 5); i += GS) g_cache[(size_t)r * CC + i] = 0;   // (the bitmaps at the head of every node's area)
    for (u32 i = l; i < N * TC / 4u; i += GS) reinterpret_cast<u32 *>(g_rep)[i] = 0xFFFFFFFFu;
  }
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node / service state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, node_msgid = 0, part = 0;
  u32 root_v = 0, next_tid = 0;                        // node: the cached root's version, thunk ids handed out
  u32 root_exists = 0, cur_v = 0, n_order = 0;         // lin-kv lane: the root
  u32 svc_ctr = 0;                                     // lww-kv lane: rand-int draws so far
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_value = 0, process = l, m_value = 0, cin_n = 0, csp_n = 0;
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, nem_next = 0, nem_j = 0;
  u32 loss_on = 0, next_id = 0, n_rows = 0, n_payload = 0, flags = 0, rounds = 0;
  bool alive = real;

  auto q_push = [&](const uint4 m) __attribute__((always_inline)) {
    if (in_n < RQ) { my_q[in_n * 64u] = m; in_n++; return; }
    if (sp_n < my_spill_cap) { my_spill[sp_n++] = m; return; }
    my_flags |= MSIM_FLAG_INBOX_OVERFLOW;
  };
  // an envelope for THIS lane's node/service arrives (net.clj:189-221)
  auto arrive = [&](u32 id, u32 type, u32 a, u32 b, u32 src) __attribute__((always_inline)) {
    u32 lat = 0;
    if (src < N || src >= LIN) {  // neither end is a client
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * m8_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
    }
    if (NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss) return;
    uint4 m = make_uint4(T + lat * 1000u, (id << 8) | type, a, b | (src << 24));
    if (!have_pm) { pm = m; have_pm = true; return; }
    if (m.x < pm.x || (m.x == pm.x && m.y < pm.y)) { const uint4 t = m; m = pm; pm = t; }
    q_push(m);
  };
  auto try_commit = [&](const uint4 e) __attribute__((always_inline)) {
    const u32 src = e.w >> 24;
    if (NEM && src < N && ((part >> src) & 1)) return;  // partitioned (node <-> node only; never happens in this program)
    cm = e;
    deliver_at = e.x <= T ? T : T + ((e.x - T) / 1000u) * 1000u;  // (Thread/sleep (long dt)) net.clj:236-238
  };
  auto poll = [&]() __attribute__((always_inline)) {
    if (have_pm) {
      have_pm = false;
      if (alive && deliver_at == INF && (in_n | sp_n) == 0) try_commit(pm);
      else q_push(pm);
    }
    while (alive && l <= N + 1u && deliver_at == INF && (in_n | sp_n) != 0) {
      u32 best = 0; bool in_spill = false;
      uint2 bk = make_uint2(INF, INF);
      for (u32 i = 0; i < in_n; i++) {
        const uint2 kk = *reinterpret_cast<const uint2 *>(&my_q[i * 64u]);
        if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; }
      }
      for (u32 i0 = 0; i0 < sp_n; i0 += 8) {   // eight spilled keys per round trip
        uint2 k8[8];
#pragma unroll
        for (u32 t = 0; t < 8; t++) k8[t] = *reinterpret_cast<const uint2 *>(&my_spill[min(i0 + t, sp_n - 1u)]);
#pragma unroll
        for (u32 t = 0; t < 8; t++) {
          const uint2 kk = k8[t];
          if (i0 + t < sp_n && (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y))) { bk = kk; best = i0 + t; in_spill = true; }
        }
      }
      uint4 e;
      if (in_spill) { e = my_spill[best]; sp_n--; if (best != sp_n) my_spill[best] = my_spill[sp_n]; }
      else { e = my_q[best * 64u]; in_n--; if (best != in_n) my_q[best * 64u] = my_q[in_n * 64u]; }
      try_commit(e);
    }
  };
  // elements of `k` visible at version `from`: versions only grow along a key's row, so the answer is a count — the row is read with
  // independent loads (one round trip) instead of one dependent load per element
  auto visible = [&](u32 k, u32 from) __attribute__((always_inline)) -> u32 {
    if (from == V_NIL) return 0u;
    const u32 cnt = g_kvn[k];
    u32 n = 0;
    if (mw <= 16u) {
      u32 row[16];
#pragma unroll
      for (u32 i = 0; i < 16u; i++) row[i] = i < cnt ? g_kv[k * mw + i] : 0xFFFFFFFFu;
#pragma unroll
      for (u32 i = 0; i < 16u; i++) n += (i < cnt && (row[i] >> 8) <= from) ? 1u : 0u;
      return n;
    }
    while (n < cnt && (g_kv[k * mw + n] >> 8) <= from) n++;
    return n;
  };

#ifdef M8_PROF   // developer build (tools/variant_lib.sh m8prof mk8.hip -DM8_PROF): cycle counters of the round's sections -> the meta of the wavefront's first three clusters
  u64 pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u32 wave_rounds = 0;
  u64 tprev = __builtin_readcyclecounter();
#define M8_MARK(i) { const u64 now_ = __builtin_readcyclecounter(); pacc[i] += now_ - tprev; tprev = now_; }
#else
#define M8_MARK(i)
#endif
  for (;;) {
    if (!__ballot(alive)) break;
#ifdef M8_PROF
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
        u32 km = m8_oct_min(k);
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

    auto complete = [&](u32 type, u32 err, u32 ref) __attribute__((always_inline)) {
      busy = false;
      if (kind != K_OP) { if (type != MSIM_T_OK) my_flags |= MSIM_FLAG_ROUND_LIMIT; return; }
      cmp_row = true; cmp_packed = type | (MSIM_F_TXN << 2) | (err << 7) | (process << 12);
      cmp_value = ref & 0xFFFFFFu; cmp_len = ref >> 24;
      if (type == MSIM_T_INFO) process += N;  // crashed process; the Reusable client itself lives on
    };
    // the client's recv! consumes one envelope (client.clj:94-107)
    auto client_deliver = [&](u32 qtype, u32 qa, u32 qb) __attribute__((always_inline)) {
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
    M8_MARK(0)
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

      M8_MARK(1)
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

      M8_MARK(2)
      // ---- R3: one input per node, then one for each service (endpoint order: lin-kv, lww-kv) ----
      bool rep = false, svc_rep = false;   // node -> own client, service -> node
      u32 n_out = 0, o_dest = 0;           // node -> service: n_out messages in mout[l][..], all to the same service
      u32 o_type = 0, o_a = 0, o_b = 0, o_to = 0, need_words = 0, done_slot = 0;
      // The three heavy steps of a transaction run ONCE per round each, after the handlers have said which slot needs which (hact: 1 start an attempt,
      // 2 write the thunks, 3 cas the root): lanes reaching them from different states would otherwise execute separate inlined copies one after the other.
      u32 hact = 0, hact_si = 0;
      u32 *const my_out = mout_g + l * (KEYS * 3u);
      auto out_msg = [&](u32 dest, u32 type, u32 a, u32 b) __attribute__((always_inline)) { o_dest = dest; my_out[n_out * 3u] = type; my_out[n_out * 3u + 1u] = a; my_out[n_out * 3u + 2u] = b; n_out++; };
      // the node's thunk cache (multi_key_txn.js:17,80-106): one bit per thunk id <node>.<i>, [owner][i / 32] in the first N x TC / 32 words of the
      // node's CC-word area (round 5: a probe of the open-addressing table it replaces was a dependent load into 256 KiB per node)
      auto cached = [&](u32 tid) __attribute__((always_inline)) -> bool { const u32 i = tid & 0xFFFFFu; return (my_cache[(tid >> 20) * (TC >> 5) + (i >> 5)] >> (i & 31u)) & 1u; };
      auto cache_add = [&](u32 tid) __attribute__((always_inline)) { const u32 i = tid & 0xFFFFFu; my_cache[(tid >> 20) * (TC >> 5) + (i >> 5)] |= 1u << (i & 31u); };
      // the thunk the root of version v names for `k` (MK_NONE: the map does not have the key)
      auto thunk_of = [&](u32 k, u32 v) __attribute__((always_inline)) -> u32 {
        const u32 first = g_first[k], cnt = g_updn[k];   // (never entered: MK_NONE > any version)
        if (first > v) return MK_NONE;
        if (mw1 <= 17u) {   // the versions of the key's thunks grow along the row: count those <= v with independent loads, then one more for the id
          u32 row[17];
#pragma unroll
          for (u32 i = 0; i < 17u; i++) row[i] = i < cnt ? g_upd_v[k * mw1 + i] : 0xFFFFFFFFu;
          u32 n = 0;
#pragma unroll
          for (u32 i = 0; i < 17u; i++) n += (i < cnt && row[i] <= v) ? 1u : 0u;
          return n ? g_upd_t[k * mw1 + n - 1u] : MK_NONE;
        }
        u32 t = MK_NONE;
        for (u32 i = 0; i < cnt && g_upd_v[k * mw1 + i] <= v; i++) t = g_upd_t[k * mw1 + i];
        return t;
      };
      auto send_cas = [&](u32 *sl, u32 si) __attribute__((always_inline)) {   // casRoot, :120-137
        const u32 rid = ++node_msgid;
        sl[SK_HDR] = 1u | (3u << 8); sl[SK_RPC] = rid;
        out_msg(D_LIN, M_CAS, sl[SK_RV] | (si << 16), rid);
      };
      // writeThunks (:160-177): state2's keys in insertion order — the thunks read, then the keys the transaction creates
      auto begin_writes = [&](u32 *sl, u32 si) __attribute__((always_inline)) {
        const u32 nk = sl[SK_NK], ns = sl[SK_NSTATE];
        u32 ord[KEYS], n = 0, in_state = 0;
        for (u32 i = 0; i < ns; i++) { const u32 j = sl[SK_SORD + i]; ord[n++] = j; in_state |= 1u << j; }
        for (u32 i = 0; i <= KEYS; i++)
          for (u32 j = 0; j < nk; j++) if (!((in_state >> j) & 1u) && sl[SK_WR + j] && sl[SK_FA + j] == i) ord[n++] = j;
        sl[SK_HDR] = 1u | (2u << 8); sl[SK_NNEW] = 0;
        u32 wr_out = 0;
        for (u32 i = 0; i < n; i++) {
          const u32 j = ord[i];
          if (!sl[SK_WR + j]) continue;
          if (next_tid >= TC) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; continue; }   // engine capacity
          const u32 tid = (l << 20) | next_tid++;
          cache_add(tid);
          const u32 rid = ++node_msgid;
          sl[SK_WRTID + j] = tid; sl[SK_WRRPC + j] = rid; wr_out++;
          out_msg(D_LWW, M_WRITE, tid, rid);
        }
        sl[SK_WROUT] = wr_out;
        hact = wr_out == 0 ? 3u : 0u; hact_si = si;
      };
      auto thunk_ready = [&](u32 *sl, u32 j) __attribute__((always_inline)) { const u32 ns = sl[SK_NSTATE]; sl[SK_SORD + ns] = j; sl[SK_NSTATE] = ns + 1u; sl[SK_RDRPC + j] = 0; };
      // transact (:213-236) from the node's cached root; getState (:141-156) walks the root's keys in map order
      // The loads of all of the transaction's keys go out together, stage by stage (map entry + thunk count + position; the last four thunk
      // versions of each — the cached root is recent, the thunk it names is nearly always among them —; the thunk ids; the first probe of
      // the cache): five round trips where a key at a time paid five each.
      auto start_attempt = [&](u32 *sl, u32 si) __attribute__((always_inline)) {
        const u32 nk = sl[SK_NK], rv = root_v;
        sl[SK_RV] = rv; sl[SK_HDR] = 1u | (1u << 8); sl[SK_NSTATE] = 0;
        u32 kk[KEYS], first[KEYS], cnt[KEYS], posn[KEYS], tids[KEYS], nle[KEYS], rd_out = 0;
#pragma unroll
        for (u32 j = 0; j < KEYS; j++) kk[j] = j < nk ? sl[SK_KEY + j] : 0u;
#pragma unroll
        for (u32 j = 0; j < KEYS; j++) {
          first[j] = MK_NONE; cnt[j] = 0; posn[j] = MK_NONE;
          if (j < nk) { sl[SK_RDRPC + j] = 0; first[j] = g_first[kk[j]]; cnt[j] = g_updn[kk[j]]; posn[j] = g_pos[kk[j]]; }
        }
        u32 l4[KEYS][4];
#pragma unroll
        for (u32 j = 0; j < KEYS; j++)
#pragma unroll
          for (u32 t = 0; t < 4u; t++) {
            const u32 idx = (cnt[j] >= 4u ? cnt[j] - 4u : 0u) + t;
            l4[j][t] = 0xFFFFFFFFu;
            if (j < nk && first[j] <= rv && idx < cnt[j]) l4[j][t] = g_upd_v[kk[j] * mw1 + idx];
          }
#pragma unroll
        for (u32 j = 0; j < KEYS; j++) {
          nle[j] = 0;
          if (j < nk && first[j] <= rv) {   // (never entered: MK_NONE > any version)
            const u32 base = cnt[j] >= 4u ? cnt[j] - 4u : 0u;
            if (base != 0u && l4[j][0] > rv) { u32 n = 0; while (n < base && g_upd_v[kk[j] * mw1 + n] <= rv) n++; nle[j] = n; }   // an old root: walk the row
            else {
              u32 n = base;
#pragma unroll
              for (u32 t = 0; t < 4u; t++) n += (base + t < cnt[j] && l4[j][t] <= rv) ? 1u : 0u;
              nle[j] = n;
            }
          }
        }
#pragma unroll
        for (u32 j = 0; j < KEYS; j++) { tids[j] = MK_NONE; if (nle[j]) tids[j] = g_upd_t[kk[j] * mw1 + nle[j] - 1u]; if (tids[j] == MK_NONE) posn[j] = MK_NONE; }
        u32 pr[KEYS];   // the thunk cache's word of every key, all keys at once
#pragma unroll
        for (u32 j = 0; j < KEYS; j++) { pr[j] = 0; if (tids[j] != MK_NONE) pr[j] = my_cache[(tids[j] >> 20) * (TC >> 5) + ((tids[j] & 0xFFFFFu) >> 5)]; }
        bool have[KEYS];
#pragma unroll
        for (u32 j = 0; j < KEYS; j++) have[j] = tids[j] != MK_NONE && ((pr[j] >> (tids[j] & 31u)) & 1u);
        for (u32 done = 0;;) {   // ascending position in the root map
          u32 best = MK_NONE, bj = 0;
#pragma unroll
          for (u32 j = 0; j < KEYS; j++) if (j < nk && !((done >> j) & 1u) && posn[j] < best) { best = posn[j]; bj = j; }
          if (best == MK_NONE) break;
          done |= 1u << bj;
          bool hv = false; u32 tb = 0;
#pragma unroll
          for (u32 j = 0; j < KEYS; j++) if (j == bj) { hv = have[j]; tb = tids[j]; }
          if (hv) thunk_ready(sl, bj);
          else { const u32 rid = ++node_msgid; sl[SK_RDTID + bj] = tb; sl[SK_RDRPC + bj] = rid; rd_out++; out_msg(D_LWW, M_READ, tb, rid); }
        }
        sl[SK_RDOUT] = rd_out;
        hact = rd_out == 0 ? 2u : 0u; hact_si = si;
      };
      const bool take = normal && l <= N + 1u && deliver_at <= T;
      if (take) {
        const uint4 q = cm; deliver_at = INF;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N && qsrc < LIN) s_recv_cl++; else s_recv_sv++;
        if (is_node) {
          switch (qtype) {
            case M_INIT: rep = true; o_type = M_INIT_OK; o_b = qb; break;
            case M_TXN: {
              u32 si = 0;
              while (si < M8_SL && (lds_slot(my_node, si)[SK_HDR] & 0xFFu)) si++;
              if (si == M8_SL) while (si < M8_SLOTS && (hbm_slot(my_node, si)[SK_HDR] & 0xFFu)) si++;
              if (si == M8_SLOTS) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; break; }   // engine capacity; the reference has no bound
              auto on_txn = [&](u32 *sl, u32 si_) __attribute__((always_inline)) -> bool {
                for (u32 i = 0; i < MKW; i++) sl[i] = 0;
                sl[SK_HDR] = 1u; sl[SK_CMSG] = qb; sl[SK_REF] = qa;
                const u32 off0 = qa & 0xFFFFFFu, n = qa >> 24;
                u32 wv[KEYS];   // the micro-ops (at most KEYS: --max-txn-length), one round trip
#pragma unroll
                for (u32 i = 0; i < KEYS; i++) wv[i] = i < n ? g_pay[off0 + i] : 0u;
                u32 nk = 0;
#pragma unroll
                for (u32 i = 0; i < KEYS; i++) if (i < n) {   // readSet / writeSet (:180-197)
                  const u32 w = wv[i], k = (w >> 1) & 0x7FFFu;
                  u32 j = 0; while (j < nk && sl[SK_KEY + j] != k) j++;
                  if (j == nk) { sl[SK_KEY + j] = k; nk++; }
                  if ((w & 1u) && !sl[SK_WR + j]) { sl[SK_WR + j] = 1u; sl[SK_FA + j] = i; }
                }
                sl[SK_NK] = nk;
                hact = 1u; hact_si = si_;
                return true;
              };
              (void)M8_ON_SLOT(my_node, si, on_txn);
            } break;
            case M_READ_OK: case M_WRITE_OK: case M_CAS_OK: case M_ERROR: {
              auto on_reply = [&](u32 *sl, u32 si_) __attribute__((always_inline)) -> bool {
                bool found = false;
                const u32 hdr = sl[SK_HDR];
                if (!(hdr & 0xFFu)) return false;
                const u32 stage_ = (hdr >> 8) & 0xFFu, nk = sl[SK_NK];
                if (stage_ == 1u) {
                  for (u32 j = 0; j < nk; j++) if (qb && sl[SK_RDRPC + j] == qb) {
                    found = true;
                    const u32 tid = sl[SK_RDTID + j];
                    if (qtype == M_READ_OK) { cache_add(tid); thunk_ready(sl, j); sl[SK_RDOUT]--; }
                    else if (qa == 20u) {   // not on the replica that answered: getThunk again (:92-96), from the cache if it is there by now
                      if (cached(tid)) { thunk_ready(sl, j); sl[SK_RDOUT]--; }
                      else { const u32 rid = ++node_msgid; sl[SK_RDRPC + j] = rid; out_msg(D_LWW, M_READ, tid, rid); }
                    }
                    if (sl[SK_RDOUT] == 0) { hact = 2u; hact_si = si_; }
                    break;
                  }
                } else if (stage_ == 2u) {
                  for (u32 j = 0; j < nk; j++) if (qb && sl[SK_WR + j] && sl[SK_WRRPC + j] == qb) {
                    found = true;
                    sl[SK_WRRPC + j] = 0;
                    // "the root of version rv has no thunk for the key" == the key entered the map later (or never): a key's first thunk is
                    // committed by the cas that enters it, so g_first[k] is also the version of its first thunk — one load, not the row
                    if (g_first[sl[SK_KEY + j]] > sl[SK_RV]) { const u32 nn = sl[SK_NNEW]; sl[SK_NORD + nn] = j; sl[SK_NNEW] = nn + 1u; }
                    if (--sl[SK_WROUT] == 0) { hact = 3u; hact_si = si_; }
                    break;
                  }
                } else if (sl[SK_RPC] == qb) {
                  found = true;
                  if (stage_ == 3u) {
                    if (qtype == M_CAS_OK) {   // :226-229: the cached root becomes the new map, the client gets the completed transaction
                      u32 writes = 0; for (u32 j = 0; j < nk; j++) writes |= sl[SK_WR + j];
                      const u32 rv = sl[SK_RV];
                      root_v = rv + (writes ? 1u : 0u);
                      rep = true; o_type = M_TXN_OK; o_b = sl[SK_CMSG]; done_slot = si_;
                      const u32 ref = sl[SK_REF], off0 = ref & 0xFFFFFFu, n = ref >> 24;
                      u32 wv[KEYS];
#pragma unroll
                      for (u32 j = 0; j < KEYS; j++) wv[j] = j < n ? g_pay[off0 + j] : 0u;
                      // how many elements each read sees at version rv: the counts of all read keys in one round trip, the last four
                      // versions of each row in a second (rv is recent: what it does not see is at the row's end)
                      u32 vcnt[KEYS], vl4[KEYS][4], vis[KEYS];
#pragma unroll
                      for (u32 j = 0; j < KEYS; j++) { vcnt[j] = 0; if (j < n && !(wv[j] & 1u) && rv != V_NIL) vcnt[j] = g_kvn[(wv[j] >> 1) & 0x7FFFu]; }
#pragma unroll
                      for (u32 j = 0; j < KEYS; j++)
#pragma unroll
                        for (u32 t = 0; t < 4u; t++) {
                          const u32 idx = (vcnt[j] >= 4u ? vcnt[j] - 4u : 0u) + t;
                          vl4[j][t] = 0xFFFFFFFFu;
                          if (idx < vcnt[j]) vl4[j][t] = g_kv[((wv[j] >> 1) & 0x7FFFu) * mw + idx];
                        }
#pragma unroll
                      for (u32 j = 0; j < KEYS; j++) {
                        const u32 base = vcnt[j] >= 4u ? vcnt[j] - 4u : 0u;
                        if (base != 0u && (vl4[j][0] >> 8) > rv) vis[j] = visible((wv[j] >> 1) & 0x7FFFu, rv);
                        else {
                          u32 c = base;
#pragma unroll
                          for (u32 t = 0; t < 4u; t++) c += (base + t < vcnt[j] && (vl4[j][t] >> 8) <= rv) ? 1u : 0u;
                          vis[j] = c;
                        }
                      }
#pragma unroll
                      for (u32 j = 0; j < KEYS; j++) if (j < n) {
                        const u32 w = wv[j], k = (w >> 1) & 0x7FFFu;
                        need_words++;
                        if (!(w & 1u)) {
                          u32 len = vis[j];
#pragma unroll
                          for (u32 e = 0; e < KEYS; e++) if (e < j) { const u32 we = wv[e]; if ((we & 1u) && ((we >> 1) & 0x7FFFu) == k) len++; }
                          need_words += (len + 3u) / 4u;
                        }
                      }
                    } else { const u32 rid = ++node_msgid; sl[SK_HDR] = 1u | (4u << 8); sl[SK_RPC] = rid; out_msg(D_LIN, M_READ, 0, rid); }   // :230-234
                  } else {   // getRoot (:112-116)
                    root_v = qtype == M_READ_OK ? qa : 0u;
                    hact = 1u; hact_si = si_;
                  }
                }
                return found;
              };
              bool found = false;
              for (u32 si = 0; si < M8_SL && !found; si++) found = on_reply(lds_slot(my_node, si), si);
              for (u32 si = M8_SL; si < M8_SLOTS && !found; si++) found = on_reply(hbm_slot(my_node, si), si);
            } break;   // no handler under that id: ignored (node.js:152-156)
            default: break;
          }
        } else if (is_lin) {   // lin-kv over the key "root" (service.clj:31-61)
          svc_rep = true; o_to = qsrc; o_b = qb;
          if (qtype == M_READ) {
            if (!root_exists) { o_type = M_ERROR; o_a = 20; } else { o_type = M_READ_OK; o_a = cur_v; }
          } else {   // cas with create_if_not_exists
            const u32 from = qa & 0xFFFFu, si = qa >> 16;
            if (root_exists && cur_v != from) { o_type = M_ERROR; o_a = 22; }
            else {
              auto on_cas = [&](u32 *sl, u32) __attribute__((always_inline)) -> bool {
              const u32 nk = sl[SK_NK], ref = sl[SK_REF], off0 = ref & 0xFFFFFFu, n = ref >> 24;
              u32 writes = 0; for (u32 j = 0; j < nk; j++) writes |= sl[SK_WR + j];
              root_exists = 1u;
              if (writes) {
                const u32 v = ++cur_v;
                const u32 nn = sl[SK_NNEW];
                for (u32 i = 0; i < nn; i++) { const u32 k = sl[SK_KEY + sl[SK_NORD + i]]; g_pos[k] = n_order++; g_first[k] = v; }
                // the thunk counts of the written keys and the element counts of the appended ones: every read-modify-write of the cas in
                // one round trip behind the micro-ops' (a key is read once: a transaction's later appends to it continue from the first's count)
                u32 wv[KEYS];
#pragma unroll
                for (u32 i = 0; i < KEYS; i++) wv[i] = i < n ? g_pay[off0 + i] : 0u;
                u32 kj[KEYS], cu[KEYS], ca[KEYS];
#pragma unroll
                for (u32 j = 0; j < KEYS; j++) { kj[j] = 0; cu[j] = 0; if (j < nk && sl[SK_WR + j]) { kj[j] = sl[SK_KEY + j]; cu[j] = g_updn[kj[j]]; } }
#pragma unroll
                for (u32 i = 0; i < KEYS; i++) { ca[i] = 0; if (i < n && (wv[i] & 1u)) ca[i] = g_kvn[(wv[i] >> 1) & 0x7FFFu]; }
#pragma unroll
                for (u32 j = 0; j < KEYS; j++) if (j < nk && sl[SK_WR + j]) { const u32 k = kj[j], c = cu[j]; g_upd_v[k * mw1 + c] = v; g_upd_t[k * mw1 + c] = sl[SK_WRTID + j]; g_updn[k] = c + 1u; }
#pragma unroll
                for (u32 i = 0; i < KEYS; i++) if (i < n && (wv[i] & 1u)) {
                  const u32 w = wv[i], k = (w >> 1) & 0x7FFFu;
                  u32 c = ca[i];
#pragma unroll
                  for (u32 e = 0; e < KEYS; e++) if (e < i && (wv[e] & 1u) && ((wv[e] >> 1) & 0x7FFFu) == k) c++;
                  g_kv[k * mw + c] = ((w >> 16) & 0xFFu) | (v << 8); g_kvn[k] = c + 1u;
                }
              }
                return true;
              };
              (void)M8_ON_SLOT(qsrc, si, on_cas);
              o_type = M_CAS_OK; o_a = 0;
            }
          }
        } else {   // lww-kv (service.clj:214-243 as written): merge-source, merge-dest, then the replica that serves the request
          svc_rep = true; o_to = qsrc; o_b = qb;
          svc_ctr += 2u;   // (merge-source and merge-dest are drawn and dropped)
          const u32 r = scale32(draw32(key, 12u /* S_SVC */, svc_ctr++), 2), tid = qa, tn = tid >> 20, ti = tid & 0xFFFFFu;
          unsigned char *const rp = g_rep + (size_t)tn * TC + ti;
          if (qtype == M_WRITE) { *rp = (unsigned char)r; o_type = M_WRITE_OK; o_a = tid; }
          else if (*rp == r) { o_type = M_READ_OK; o_a = tid; }
          else { o_type = M_ERROR; o_a = 20; }
        }
      }

      if (hact == 1u) { if (hact_si < M8_SL) start_attempt(lds_slot(my_node, hact_si), hact_si); else start_attempt(hbm_slot(my_node, hact_si), hact_si); }   // (sets hact = 2 when nothing has to be read)
      if (hact == 2u) { if (hact_si < M8_SL) begin_writes(lds_slot(my_node, hact_si), hact_si); else begin_writes(hbm_slot(my_node, hact_si), hact_si); }     // (sets hact = 3 when nothing has to be written)
      if (hact == 3u) { if (hact_si < M8_SL) send_cas(lds_slot(my_node, hact_si), hact_si); else send_cas(hbm_slot(my_node, hact_si), hact_si); }

      M8_MARK(3)
      // completed transactions: payload words allocated in node order, each node writes its own
      if (__ballot(need_words != 0)) {
        u32 excl = 0, total = 0;
        for (u32 s = 0; s < N; s++) { const u32 v = GGET(need_words, s); excl += s < l ? v : 0u; total += v; }
        if (total) {
          if (n_payload + total > max_pay) { flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; if (need_words) { o_a = 0; if (done_slot < M8_SL) lds_slot(my_node, done_slot)[SK_HDR] = 0; else hbm_slot(my_node, done_slot)[SK_HDR] = 0; } }
          else {
            if (need_words) {
              u32 ref, from;
              if (done_slot < M8_SL) { u32 *const sl = lds_slot(my_node, done_slot); ref = sl[SK_REF]; from = sl[SK_RV]; sl[SK_HDR] = 0; }
              else { u32 *const sl = hbm_slot(my_node, done_slot); ref = sl[SK_REF]; from = sl[SK_RV]; sl[SK_HDR] = 0; }
              const u32 off0 = ref & 0xFFFFFFu, n = ref >> 24;
              u32 pp = n_payload + excl;
              o_a = pp | (need_words << 24);
              u32 wv[KEYS];
#pragma unroll
              for (u32 j = 0; j < KEYS; j++) wv[j] = j < n ? g_pay[off0 + j] : 0u;
#pragma unroll
              for (u32 j = 0; j < KEYS; j++) if (j < n) {
                const u32 w = wv[j], k = (w >> 1) & 0x7FFFu;
                if (w & 1u) { g_pay[pp++] = w; continue; }
                u32 e = 0, acc = 0;
                const u32 hdr = pp++;
                if (from != V_NIL) {
                  const u32 cnt = g_kvn[k];
                  if (mw <= 16u) {   // the key's row once, with independent loads: the visible prefix is counted and packed from registers
                    u32 row[16];
#pragma unroll
                    for (u32 i = 0; i < 16u; i++) row[i] = i < cnt ? g_kv[k * mw + i] : 0xFFFFFFFFu;
#pragma unroll
                    for (u32 i = 0; i < 16u; i++) if (i < cnt && (row[i] >> 8) <= from) { acc |= (row[i] & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } }
                  } else {
                    const u32 vis = visible(k, from);
                    for (u32 i = 0; i < vis; i++) { acc |= (g_kv[k * mw + i] & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } }
                  }
                }
#pragma unroll
                for (u32 i = 0; i < KEYS; i++) if (i < j) { const u32 wi = wv[i];
                  if ((wi & 1u) && ((wi >> 1) & 0x7FFFu) == k) { acc |= ((wi >> 16) & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } } }
                if (e & 3) g_pay[pp++] = acc;
                g_pay[hdr] = (k << 1) | ((e ? e : 0xFFu) << 16);  // a key without elements reads nil
              }
            }
            n_payload += total;
          }
        }
      }

      M8_MARK(4)
      // COMMIT: ids in lane order (nodes, lin-kv, lww-kv); a node's messages in the order it emitted them
      bool c_arr = false; u32 ca_y = 0, ca_a = 0, ca_b = 0;
      {
        const u32 cnt = is_node ? (rep ? 1u : n_out) : (svc_rep ? 1u : 0u);
        if (__ballot(cnt != 0)) {
          wave_lds_fence();   // (the services read the nodes' mout rows)
          u32 my_off = 0, total = 0;
          for (u32 s = 0; s < N + 2u; s++) { const u32 v = GGET(cnt, s); my_off += s < l ? v : 0u; total += v; }
          if (rep) s_send_cl++; else s_send_sv += cnt;
          // node -> service: the service lane takes each node's run in node order
          u32 ts = GB(is_node && !rep && n_out != 0);
          while (__ballot(ts != 0)) {
            const bool on = ts != 0;
            const u32 s = on ? (u32)__builtin_ctz(ts) : 0u; ts &= ts - 1u;
            const u32 dst = GGET(o_dest, s), kn = GGET(n_out, s), off = GGET(my_off, s);
            if (on && l == N + dst) {
              const u32 *const mo = mout_g + s * (KEYS * 3u);
              for (u32 k = 0; k < kn; k++) arrive(next_id + off + k, mo[k * 3u], mo[k * 3u + 1u], mo[k * 3u + 2u], s);
            }
          }
          // service -> node (lin-kv, then lww-kv)
          {
            const u32 sv = GB(svc_rep);
#pragma unroll
            for (u32 q2 = 0; q2 < 2u; q2++) {
              const u32 s = N + q2;
              const u32 ty = GGET(o_type, s), a = GGET(o_a, s), b = GGET(o_b, s), d = GGET(o_to, s), off = GGET(my_off, s);
              if (((sv >> s) & 1u) && l == d) arrive(next_id + off, ty, a, b, N + s);
            }
          }
          // node -> its own client: no latency; lost like any other message (net.clj:214)
          if (rep) {
            const u32 id = next_id + my_off;
            if (!(NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss)) { c_arr = true; ca_y = (id << 8) | o_type; ca_a = o_a; ca_b = o_b; }
          }
          next_id += total;
        }
        poll();
      }

      M8_MARK(5)
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

    M8_MARK(6)
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
    M8_MARK(7)
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
#ifdef M8_PROF
    if (grp == 0) { m.n_events = (u32)(pacc[0] >> 6); m.reserved[0] = (u32)(pacc[1] >> 6); m.reserved[1] = (u32)(pacc[2] >> 6); m.reserved[2] = (u32)(pacc[3] >> 6); }
    if (grp == 1) { m.n_events = (u32)(pacc[4] >> 6); m.reserved[0] = (u32)(pacc[5] >> 6); m.reserved[1] = (u32)(pacc[6] >> 6); m.reserved[2] = (u32)(pacc[7] >> 6); }
    if (grp == 2) { m.n_events = wave_rounds; }
#endif
    p.meta[inst] = m;
  }
}

}  // namespace

// Whether eight clusters per wavefront simulate this configuration (see the header of this file).
bool msim_mk8_eligible(const msim_config &c) {
  return c.node_program == MSIM_NODE_TXN_MULTI_KEY && c.journal_capacity == 0 && c.n_nodes >= 1 && c.n_nodes <= GS - 2u && c.concurrency == c.n_nodes &&
         c.max_txn_length <= KEYS;
}

// Extra per-instance scratch words behind the queues' spill area: the clients' spill, and what of the LDS queues of mk_kernel<> does not
// fit this kernel's RQ slots.
uint64_t msim_mk8_extra_scratch_words(const msim_config &c) {
  return ((uint64_t)(c.n_nodes + 2) * c.inbox_capacity + (uint64_t)c.n_nodes * M8_CLIENT_CAP) * 4;
}

hipError_t msim_launch_mk8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  M8Params tp;
  tp.k = kp; tp.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  tp.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;
  tp.client_spill = M8_CLIENT_CAP - CQ;
  tp.client_spill_off = kp.spill_off + (uint64_t)(kp.N + 2) * tp.node_spill * 4;
  size_t off = (size_t)RQ * 64 * 16;
  tp.off_cq = (u32)off; off += (size_t)CQ * 64 * 16;
  tp.off_slots = (u32)off; off += (size_t)8 * kp.N * M8_SL * MKW * 4;
  tp.off_mout = (u32)off; off += (size_t)8 * kp.N * KEYS * 3 * 4;
  tp.off_gen = (u32)off; off += (size_t)8 * 36 * 4;
  off = (off + 15) & ~(size_t)15;
  tp.off_misc = (u32)off; off += 64 * 4;
  tp.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(m8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 7) / 8), block(64);
  if (kp.dev_flags & 0x1000u) std::fprintf(stderr, "[mk8] %u clusters, eight per wavefront, %zu B of LDS per wavefront\n", n, lds);   // developer trace bit
  if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((mk8_kernel<true, true>), grid, block, lds, st, tp); else hipLaunchKernelGGL((mk8_kernel<true, false>), grid, block, lds, st, tp); }
  else { if (rnd) hipLaunchKernelGGL((mk8_kernel<false, true>), grid, block, lds, st, tp); else hipLaunchKernelGGL((mk8_kernel<false, false>), grid, block, lds, st, tp); }
  return hipGetLastError();
}
