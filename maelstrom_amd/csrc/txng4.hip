// txng4.hip — FOUR clusters of the single-root txn-list-append node with SEVERAL WORKERS PER NODE per wavefront (`--concurrency k n`: the way the
// reference's own runs of this workload are invoked, doc/05-datomic/01-single-node.md:257,322 — 10n workers), in 16-lane groups.
//
// Same program and the same rounds as txng_kernel<> (sim_kernel_txng.inc; specification: oracle/txn_nodes.inc): node =
// demo/clojure/single_key_txn.clj:116-180 == demo/js/single_key_txn.js (every request in its own future: read the root, apply, cas with
// create_if_not_exists; a lost race answers error 30), service = lin-kv over the key "root" (service.clj:31-61), client =
// workload/txn_list_append.clj:94-126, generator = [upstream] elle list-append.  What changes is the mapping, as in txng4.hip (whose time /
// scheduler / COMMIT / client machinery this file shares line for line): a cluster is n nodes + its worker slots + lin-kv <= 16 endpoints, one
// lane each of a 16-lane group — 1 node with 10 workers is 12, 5 nodes with 10 workers are 16 — and a wavefront carries four clusters.
//
// Scope (engine.hip picks this kernel when all of it holds, else txng_kernel<> runs): concurrency a multiple of n above n,
// n + concurrency + 1 <= 16, net journal off, at least MSIM_TXNG4_MIN_CLUSTERS clusters in the launch.
//
// LDS of a wavefront: envelope queues slot-major (RQ envelopes per endpoint, the rest spills to HBM: servers inbox_capacity + spill_capacity in
// all, clients 32, the oracle's limits), per node the first few (its workers + 4) of its 64 transaction slots {client msg | client << 24, txn ref, rpc id, from |
// stage << 16 | used << 24} (the others in HBM scratch: in use only while replies are lost), per cluster the generator's key pool and the nemesis shuffle.  The append log lives in HBM scratch; history rows go straight to HBM.
//
// Envelope (16 B): x = deadline, y = (id << 8) | type, z = a, w = b | (src << 24); src = the sender's lane in its group (lin-kv: n + slots).
#include <hip/hip_runtime.h>

#include "wave_common.h"
#include "log2_table.h"
#include "layout_thresholds.h"

namespace {

__constant__ u32 g4_log2_q24[257];

constexpr u32 GS = 16u;           // lanes per cluster
#ifndef G4_RQ
#define G4_RQ 2u
#endif
#ifndef G4_WAVES
#define G4_WAVES 4
#endif
constexpr u32 RQ = G4_RQ;         // LDS envelopes per endpoint
constexpr u32 G4_CLIENT_CAP = 32u;   // Reusable lin-kv clients (lin_kv.clj:74-76) collect late replies between RPCs (the oracle's limit)
constexpr u32 G4_SLOTS = 64u;     // transactions in flight per node (TG_SLOTS of sim_kernel_txng.inc, the oracle's limit) ...
// ... of which the first G4Params.ls live in LDS (a node's workers + 4, at least 8); the others lie in HBM scratch (in use only while replies are lost: a node has no
// RPC timeout, a lost reply leaves its slot taken)
constexpr u32 V_NIL = 0xFFFFu;
enum { M_WRITE = 14, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR, M_TXN = 23, M_TXN_OK = 24 };
enum { S_GEN3 = 3 };

struct G4Params {
  KParams k;
  u32 n_inst;
  u32 off_slots, off_gen, off_misc;   // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;               // HBM spill entries per server endpoint / client behind the RQ LDS slots
  u64 client_spill_off;                       // word offset of the clients' spill area inside the per-instance scratch
  u64 xslots_off;                             // word offset of the nodes' transaction slots beyond the LDS ones
  u32 ls;                                     // transaction slots per node in LDS
  u32 round_limit;
};

__device__ __forceinline__ u32 g4_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = g4_log2_q24[idx], l1 = g4_log2_q24[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}
// min over the 16 lanes of the caller's DPP row (= its group), in every lane of the row
__device__ __forceinline__ u32 row_min(u32 v) {
  v = min(v, dpp_mov<0xB1, 0xF, 0xF, false>(v, v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_mov<0x4E, 0xF, 0xF, false>(v, v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_mov<0x141, 0xF, 0xF, false>(v, v));  // row_half_mirror
  v = min(v, dpp_mov<0x140, 0xF, 0xF, false>(v, v));  // row_mirror
  return v;
}
// inclusive prefix sum over the 16 lanes of the row
__device__ __forceinline__ u32 row_scan(u32 v) {
  v += dpp_mov<0x111, 0xF, 0xF, true>(0, v);   // row_shr:1
  v += dpp_mov<0x112, 0xF, 0xF, true>(0, v);   // row_shr:2
  v += dpp_mov<0x114, 0xF, 0xF, true>(0, v);   // row_shr:4
  v += dpp_mov<0x118, 0xF, 0xF, true>(0, v);   // row_shr:8
  return v;
}

template <bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(G4_WAVES))) txng4_kernel(const G4Params rp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = rp.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 4, gbase = lane & 48u;
  const u32 N = p.N, C = p.C, CS = p.CS;
  const u32 SVC = N + CS;
  const bool is_node = l < N;
  const bool is_client = l >= N && l < N + CS;
  const bool is_svc = l == SVC;   // lin-kv
  const bool is_server = is_node || is_svc;   // endpoints that poll all the time and see latency
  const u32 slot = l - N;
  const bool is_worker = is_client && slot < C;
  const u32 inst_raw = blockIdx.x * 4u + grp;
  const bool real = inst_raw < rp.n_inst;
  const u32 inst = real ? inst_raw : rp.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << l) - 1u;
  const u32 worker_mask = ((1u << C) - 1u) << N;
  const u32 all_nodes = (1u << N) - 1u;
  const u32 max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz, mw = p.cfg.max_writes_per_key, mv = p.cfg.max_values;
  const u32 rpc_timeout_ms = p.cfg.client_timeout_ms;
  const u32 round_limit = rp.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  u32 *const g_kv = g_scr;                          // [max_values][mw]: element | version << 8 (the one append log)
  u32 *const g_kvn = g_kv + (size_t)mv * mw;        // [max_values] elements so far
  const u32 my_spill_cap = is_server ? rp.node_spill : (is_client ? rp.client_spill : 0u);
  uint4 *const my_spill = is_server ? reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)(is_node ? l : N) * rp.node_spill
                                    : reinterpret_cast<uint4 *>(g_scr + rp.client_spill_off) + (size_t)(is_client ? slot : 0) * rp.client_spill;

  // LDS
  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                         // slot s at my_q[s * 64]
  const u32 G4_LS = rp.ls;
  uint4 *const slots_g = reinterpret_cast<uint4 *>(smem + rp.off_slots) + grp * N * G4_LS;              // [node of the group][G4_LS]
  uint4 *const xslots = reinterpret_cast<uint4 *>(g_scr + rp.xslots_off);                               // [node][G4_SLOTS - G4_LS]
  const u32 my_node = is_node ? l : 0u;
  // slot i of node nd: LDS for the first G4_LS, HBM scratch beyond (explicit branches: ds_ accesses for the slots in use nearly always)
  auto slot_ld = [&](u32 nd, u32 i) -> uint4 { return i < G4_LS ? slots_g[nd * G4_LS + i] : xslots[nd * (G4_SLOTS - G4_LS) + (i - G4_LS)]; };
  auto slot_st = [&](u32 nd, u32 i, const uint4 v) { if (i < G4_LS) slots_g[nd * G4_LS + i] = v; else xslots[nd * (G4_SLOTS - G4_LS) + (i - G4_LS)] = v; };
  u32 *const gpool = reinterpret_cast<u32 *>(smem + rp.off_gen) + grp * 36;                               // active[16], next_val[16], next_key
  u32 *const misc = reinterpret_cast<u32 *>(smem + rp.off_misc) + grp * GS;

  for (u32 i = lane; i < 4 * N * G4_LS; i += 64) reinterpret_cast<uint4 *>(smem + rp.off_slots)[i] = make_uint4(0, 0, 0, 0);
  if (real) for (u32 i = l; i < N * (G4_SLOTS - G4_LS); i += GS) xslots[i] = make_uint4(0, 0, 0, 0);
  gpool[l] = l; gpool[16 + l] = 1;
  if (l == 0) gpool[32] = p.cfg.key_count;
  if (real) for (u32 i = l; i < mv; i += GS) g_kvn[i] = 0;
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- endpoint state ----
  bool has_c = false; u32 deliver_at = 0; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 node_msgid = 0;
  u32 root = V_NIL;                                    // lin-kv lane: the version of "root" (V_NIL: the key does not exist)
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_value = 0, process = slot, m_value = 0;
  const u32 dest_node = is_client ? slot % N : 0u;     // worker t -> node t mod N; a crashed process's successor (process + C) keeps it, C being a multiple of N
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
  auto arrive = [&](u32 id, u32 type, u32 a, u32 b, u32 src) {
    u32 lat = 0;
    if ((src < N || src == SVC) && is_server) {  // neither end is a client (util.clj:7-16)
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * g4_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
    }
    if (NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss) return;
    uint4 m = make_uint4(T + lat * 1000u, (id << 8) | type, a, b | (src << 24));
    if (!have_pm) { pm = m; have_pm = true; return; }
    if (m.x < pm.x || (m.x == pm.x && m.y < pm.y)) { const uint4 t = m; m = pm; pm = t; }
    q_push(m);
  };
  auto try_commit = [&](const uint4 e) {
    const u32 src = e.w >> 24;
    if (NEM && is_node && src < N && ((part >> src) & 1)) return;
    cm = e; has_c = true;
    deliver_at = e.x <= T ? T : T + ((e.x - T) / 1000u) * 1000u;
  };
  auto poll = [&]() {
    const bool elig = alive && (is_server || busy);
    if (have_pm) {
      have_pm = false;
      if (elig && !has_c && (in_n | sp_n) == 0) try_commit(pm);
      else q_push(pm);
    }
    while (elig && !has_c && (in_n | sp_n) != 0) {
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
  // elements of `k` visible at version `from`
  auto visible = [&](u32 k, u32 from) -> u32 {
    if (from == V_NIL) return 0u;
    const u32 cnt = g_kvn[k];
    u32 n = 0;
    while (n < cnt && (g_kv[k * mw + n] >> 8) <= from) n++;
    return n;
  };

  for (;;) {
    if (!__ballot(alive)) break;

    const u32 busy_mask = GB(busy);

    // ---- time-free phase transitions: lin-kv has no final generator (core.clj:74-80 applies only with one) ----
    if (__ballot(alive && !(phase == PH_MAIN && ((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff))))) {
      for (;;) {
        bool ch = false;
        if (alive) {
          if (phase == PH_INIT_WAIT && !busy_mask) { phase = PH_MAIN_START; ch = true; }
          if (phase == PH_MAIN_START) { cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; nem_next = T; next_msg_id = 0; loss_on = 1; phase = PH_MAIN; ch = true; }
          if (phase == PH_MAIN && !((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff)) && !(rate == 0 && T < cutoff)) { phase = PH_DRAIN; ch = true; }
          if (phase == PH_DRAIN && !(busy_mask & worker_mask)) { phase = PH_DONE; ch = true; }
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
    const u32 free_mask = worker_mask & ~busy_mask;
    u32 due = INF;
    if (phase == PH_INIT) due = T;
    else if (phase == PH_MAIN) {
      if (nem_live) due = max(nem_next, T);
      if (gen_live && free_mask) due = min(due, max(gen_next, T));
      if (rate == 0 && !nem_live) due = min(due, cutoff);
    }
    u32 my_t = has_c ? deliver_at : INF;
    bool timeout_round = false;
    {
      const bool none_due = GB(my_t <= T) == 0;
      const bool jump = alive && due > T && none_due;
      if (__ballot(jump)) {
        u32 k = my_t == INF ? INF : my_t * 2;
        if (busy) k = min(k, timeout_at * 2 + 1);
        u32 km = row_min(k);
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
      if (type == MSIM_T_INFO) process += C;  // crashed process; the Reusable client itself lives on
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) complete(MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_value);
    }
    bool normal = alive && !timeout_round;   // this cluster runs R1-R4 in this wave-round
    if (__ballot(normal)) {
      // ---- R1: scheduler ----
      const bool act = normal && due <= T;
      if (__ballot(act && phase == PH_INIT)) {
        if (act && phase == PH_INIT) { if (is_client && slot < N) { mark = true; kind = K_INIT; } phase = PH_INIT_WAIT; }
      }
      if (NEM) {
        const bool nem_act = act && phase == PH_MAIN && nem_live && nem_next <= T;
        if (__ballot(nem_act)) {
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
        const bool gen = act && phase == PH_MAIN && gen_live && gen_next <= T && free_mask != 0;
        if (__ballot(gen)) {
          const u32 nfree = __popc(free_mask);
          const u32 kk = gen_k;
          const u64 h = draw64(key, S_GEN, kk);
          const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          const u32 pick = scale32(r_lo, nfree);
          const bool sel = gen && is_worker && !busy && (u32)__popc(free_mask & lt) == pick;
          // the transaction ([upstream] elle list-append gen): lane 0 of the group writes the micro-ops and owns the key pool
          const u32 n_mops = 1 + scale32((u32)(draw64(key, S_GEN2, kk) >> 32), p.cfg.max_txn_length);
          u32 bad = 0;
          if (gen && n_payload + n_mops > max_pay) bad = MSIM_FLAG_PAYLOAD_OVERFLOW;
          else if (gen && l == 0) {
            const u32 kc = p.cfg.key_count;
            for (u32 j = 0; j < n_mops; j++) {
              const u64 h3 = draw64(key, S_GEN3, (u64)kk * 8 + j);
              const u32 x = scale32((u32)(h3 >> 32), (1u << kc) - 1) + 1;
              const u32 ki = 31 - (u32)__clz((int)x);
              const u32 k = gpool[ki];
              if (h3 & 1) {
                const u32 v = gpool[16 + ki];
                gpool[16 + ki] = v + 1;
                g_pay[n_payload + j] = 1u | (k << 1) | (v << 16);
                if (v + 1 > mw) {
                  const u32 nk = gpool[32];
                  if (nk >= p.cfg.max_values) { bad = MSIM_FLAG_VALUES_OVERFLOW; break; }
                  gpool[ki] = nk; gpool[32] = nk + 1; gpool[16 + ki] = 1;
                }
              } else g_pay[n_payload + j] = (k << 1) | (0xFFu << 16);
            }
          }
          bad = GGET(bad, 0);
          if (gen && bad) { flags |= bad; phase = PH_DONE; alive = false; normal = false; }
          if (sel && !bad) { mark = true; kind = K_OP; m_value = n_payload | (n_mops << 24); }
          if (gen && !bad) { gen_k++; n_payload += n_mops; gen_next = T + __umulhi(r_hi, p.gen_period2_us); }
        }
      }

      // ---- R2: marked clients invoke ----
      if (__ballot(mark && normal)) {
        const bool inv = mark && normal;
        u32 rq_dest = 0, rq_type = 0, rq_a = 0;
        if (inv) {
          mark = false; busy = true;
          if (kind == K_INIT) { rq_dest = slot; rq_type = M_INIT; next_msg_id = 0; }
          else {
            c_value = m_value;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (MSIM_F_TXN << 2) | (process << 12); inv_value = c_value & 0xFFFFFFu; inv_len = c_value >> 24;
            rq_dest = dest_node; rq_type = M_TXN; rq_a = c_value;
          }
          want = ++next_msg_id;
          timeout_at = T + (kind == K_OP ? rpc_timeout_ms : 10000u) * 1000u;
          s_send_cl++;
        }
        const u32 rq_pack = rq_dest | (rq_type << 8);
        u32 im = GB(inv);
        const u32 n_inv = __popc(im);
        u32 idx = 0;
        while (__ballot(im != 0)) {
          const bool on = im != 0;
          const u32 s = on ? (u32)__builtin_ctz(im) : 0u; im &= im - 1u;
          const u32 pk = GGET(rq_pack, s), a = GGET(rq_a, s), b = GGET(want, s);
          if (on && l == (pk & 0xFF)) arrive(next_id + idx, pk >> 8, a, b, s);
          idx++;
        }
        next_id += n_inv;
        poll();
      }

      // ---- R3: one input per node, then one for the service (endpoint order) ----
      bool rep = false; u32 rep_dest = 0, rep_type = 0, rep_a = 0, rep_b = 0;   // what this server endpoint sends (at most one message a round)
      u32 need_words = 0, done_slot = 0;
      if (is_server && normal && has_c && deliver_at <= T) {
        const uint4 q = cm; has_c = false;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N && qsrc < SVC) s_recv_cl++; else s_recv_sv++;
        if (is_node) {
          switch (qtype) {
            case M_INIT: rep = true; rep_dest = qsrc; rep_type = M_INIT_OK; rep_b = qb; break;
            case M_TXN: {   // every request in its own future (:98-100): read the root (:150-157)
              u32 i = 0; while (i < G4_SLOTS && (slot_ld(my_node, i).w >> 24)) i++;
              if (i == G4_SLOTS) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; break; }
              const u32 rid = ++node_msgid;
              slot_st(my_node, i, make_uint4(qb | (qsrc << 24), qa, rid, (1u << 16) | (1u << 24)));
              rep = true; rep_dest = SVC; rep_type = M_READ; rep_a = 0; rep_b = rid;
            } break;
            case M_READ_OK: case M_CAS_OK: case M_ERROR: {
              u32 i = 0;
              while (i < G4_SLOTS) { const uint4 sl = slot_ld(my_node, i); if ((sl.w >> 24) && sl.z == qb) break; i++; }
              if (i == G4_SLOTS) break;  // handle-reply!: no such rpc
              uint4 sl = slot_ld(my_node, i);
              if (((sl.w >> 16) & 0xFF) == 1) {
                u32 from;
                if (qtype == M_READ_OK) from = qa;
                else if (qtype == M_ERROR && qa == 20) from = V_NIL;
                else { rep = true; rep_dest = sl.x >> 24; rep_type = M_ERROR; rep_a = qa; rep_b = sl.x & 0xFFFFFFu; slot_st(my_node, i, make_uint4(0, 0, 0, 0)); break; }
                const u32 rid = ++node_msgid;
                sl.z = rid; sl.w = from | (2u << 16) | (1u << 24);
                slot_st(my_node, i, sl);
                rep = true; rep_dest = SVC; rep_type = M_CAS; rep_a = from | (i << 16); rep_b = rid;   // cas-service!, :159-169
              } else {
                rep = true; rep_dest = sl.x >> 24; rep_b = sl.x & 0xFFFFFFu;
                if (qtype == M_CAS_OK) {  // the completed transaction goes into the payload area (sized here, written below)
                  rep_type = M_TXN_OK; done_slot = i;
                  const u32 off0 = sl.y & 0xFFFFFFu, n = sl.y >> 24, from = sl.w & 0xFFFFu;
                  for (u32 j = 0; j < n; j++) {
                    const u32 w = g_pay[off0 + j], k = (w >> 1) & 0x7FFFu;
                    need_words++;
                    if (!(w & 1)) {
                      u32 len = visible(k, from);
                      for (u32 e = 0; e < j; e++) { const u32 we = g_pay[off0 + e]; if ((we & 1) && ((we >> 1) & 0x7FFFu) == k) len++; }
                      need_words += (len + 3) / 4;
                    }
                  }
                } else { rep_type = M_ERROR; rep_a = qa == 22 ? 30u : qa; slot_st(my_node, i, make_uint4(0, 0, 0, 0)); }   // "root altered", :178-180
              }
            } break;
            default: break;
          }
        } else {  // the lin-kv service (service.clj:31-61 over the key "root")
          rep = true; rep_dest = qsrc; rep_b = qb;
          if (qtype == M_READ) {
            if (root == V_NIL) { rep_type = M_ERROR; rep_a = 20; } else { rep_type = M_READ_OK; rep_a = root; }
          } else {  // cas with create_if_not_exists
            const u32 from = qa & 0xFFFFu, i = qa >> 16;
            if (root != V_NIL && root != from) { rep_type = M_ERROR; rep_a = 22; }
            else {
              const u32 base = root == V_NIL ? 0u : root;
              const u32 ref = slot_ld(qsrc, i).y, off0 = ref & 0xFFFFFFu, n = ref >> 24;
              u32 na = 0;
              for (u32 j = 0; j < n; j++) na += g_pay[off0 + j] & 1;
              for (u32 j = 0; j < n; j++) {
                const u32 w = g_pay[off0 + j];
                if (w & 1) { const u32 k = (w >> 1) & 0x7FFFu; const u32 c = g_kvn[k]; g_kv[k * mw + c] = ((w >> 16) & 0xFFu) | ((base + na) << 8); g_kvn[k] = c + 1; }
              }
              root = base + na;
              rep_type = M_CAS_OK; rep_a = 0;
            }
          }
        }
      }

      // completed transactions: payload words allocated in node order, each node writes its own
      if (__ballot(need_words != 0)) {
        const u32 incl = row_scan(need_words);
        const u32 total = GGET(incl, GS - 1u);
        if (total) {
          if (n_payload + total > max_pay) { flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; if (need_words) { rep_a = 0; slot_st(my_node, done_slot, make_uint4(0, 0, 0, 0)); } }
          else {
            if (need_words) {
              const uint4 sl = slot_ld(my_node, done_slot);
              const u32 off0 = sl.y & 0xFFFFFFu, n = sl.y >> 24, from = sl.w & 0xFFFFu;
              u32 pp = n_payload + incl - need_words;
              rep_a = pp | (need_words << 24);
              for (u32 j = 0; j < n; j++) {
                const u32 w = g_pay[off0 + j], k = (w >> 1) & 0x7FFFu;
                if (w & 1) { g_pay[pp++] = w; continue; }
                const u32 vis = visible(k, from);
                u32 e = 0, acc = 0;
                const u32 hdr = pp++;
                for (u32 i = 0; i < vis; i++) { acc |= (g_kv[k * mw + i] & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } }
                for (u32 i = 0; i < j; i++) { const u32 wi = g_pay[off0 + i];
                  if ((wi & 1) && ((wi >> 1) & 0x7FFFu) == k) { acc |= ((wi >> 16) & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } } }
                if (e & 3) g_pay[pp++] = acc;
                g_pay[hdr] = (k << 1) | ((e ? e : 0xFFu) << 16);  // a key without elements reads nil
              }
              slot_st(my_node, done_slot, make_uint4(0, 0, 0, 0));
            }
            n_payload += total;
          }
        }
      }

      // COMMIT: one message per server endpoint at most; ids in lane order (nodes, then the service)
      {
        const u32 reps0 = GB(rep);
        if (__ballot(reps0 != 0)) {
          const u32 my_off = __popc(reps0 & lt);
          if (rep) { if (rep_dest >= N && rep_dest < SVC) s_send_cl++; else s_send_sv++; }
          const u32 rep_pack = rep_dest | (rep_type << 8);
          u32 reps = reps0;
          while (__ballot(reps != 0)) {
            const bool on = reps != 0;
            const u32 s = on ? (u32)__builtin_ctz(reps) : 0u; reps &= reps - 1u;
            const u32 pk = GGET(rep_pack, s), o = GGET(my_off, s);
            const u32 r_a = GGET(rep_a, s), r_b = GGET(rep_b, s);
            if (on && l == (pk & 0xFF)) arrive(next_id + o, pk >> 8, r_a, r_b, s);
          }
          next_id += __popc(reps0);
        }
        if (normal) poll();
      }


      // ---- R4: clients' recv! loops ----
      for (;;) {
        const bool dl = normal && is_client && has_c && deliver_at <= T;
        if (!__ballot(dl)) break;
        if (dl) {
          const uint4 q = cm; has_c = false;
          s_recv_cl++;
          const u32 qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
          if (busy && qb == want) {  // else stale (client.clj:105-107)
            if (qtype == M_TXN_OK) complete(MSIM_T_OK, 0, qa);
            else if (qtype == M_ERROR) {
              if (qa == 0u) complete(MSIM_T_INFO, MSIM_ERR_TIMEOUT, c_value);   // code 0 :timeout is not :definite? (errors.edn:2-4)
              else complete(MSIM_T_FAIL, qa == 11 ? MSIM_ERR_TEMPORARILY_UNAVAILABLE : qa == 20 ? MSIM_ERR_KEY_DOES_NOT_EXIST : qa == 30 ? MSIM_ERR_TXN_CONFLICT : qa == 14 ? MSIM_ERR_ABORT : MSIM_ERR_PRECONDITION_FAILED, c_value);
            } else complete(MSIM_T_OK, 0, c_value);  // init_ok
          }
          poll();
        }
      }
    }
    // ---- history rows: nemesis rows, invocations (slot order), completions (slot order) ----
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
        // rows straight to HBM: the rows of a round are adjacent 16-byte stores (a staging ring of 64 rows per cluster was 4 KiB of LDS and a wavefront per SIMD less)
        msim_op *const gr = g_rows;
        if (NEM && wr && nem_rows && l == 0) {
          const u32 pk = MSIM_T_INFO | (nem_f << 2) | (MSIM_PROCESS_NEMESIS << 12);
          reinterpret_cast<uint4 *>(gr)[n_rows] = make_uint4(tlo, thi, pk, nem_v1);
          reinterpret_cast<uint4 *>(gr)[n_rows + 1] = make_uint4(tlo, thi | (nem_len2 << 16), pk, nem_v2);
        }
        if (wr && inv_row) reinterpret_cast<uint4 *>(gr)[n_rows + nem_rows + __popc(imask & lt)] = make_uint4(tlo, thi | (inv_len << 16), inv_packed, inv_value);
        if (wr && cmp_row) reinterpret_cast<uint4 *>(gr)[n_rows + nem_rows + ni + __popc(cmask & lt)] = make_uint4(tlo, thi | (cmp_len << 16), cmp_packed, cmp_value);
        const u32 new_n = wr ? n_rows + nr : n_rows;
        n_rows = new_n;
      }
    }
  }

  const u32 t_send_cl = GGET(row_scan(s_send_cl), GS - 1u), t_send_sv = GGET(row_scan(s_send_sv), GS - 1u);
  const u32 t_recv_cl = GGET(row_scan(s_recv_cl), GS - 1u), t_recv_sv = GGET(row_scan(s_recv_sv), GS - 1u);
  for (u32 b = 1; b <= MSIM_FLAG_ARENA_OVERRUN; b <<= 1) if (GB((my_flags & b) != 0)) flags |= b;
  if (real && l == 0) {
    msim_net_stats st;
    st.all_send = (u64)t_send_cl + t_send_sv; st.all_recv = (u64)t_recv_cl + t_recv_sv;
    st.clients_send = t_send_cl; st.clients_recv = t_recv_cl;
    st.servers_send = t_send_sv; st.servers_recv = t_recv_sv;
    p.stats[inst] = st;
    msim_inst_meta m; m.n_rows = n_rows; m.n_payload_words = n_payload; m.flags = flags; m.n_rounds = rounds;
    m.n_events = 0; m.reserved[0] = 0; m.reserved[1] = 0; m.reserved[2] = 0;
    p.meta[inst] = m;
  }
}

}  // namespace

// Whether four clusters per wavefront simulate this configuration (see the header of this file).
bool msim_txng4_eligible(const msim_config &c) {
  if (c.node_program != MSIM_NODE_TXN_SINGLE_KEY || c.journal_capacity != 0 || c.concurrency <= c.n_nodes) return false;
  return c.n_nodes >= 1 && c.n_nodes + c.concurrency + 1 <= GS;
}

// Extra per-instance scratch words the layout needs behind txng_kernel<>'s spill area: the clients' whole inboxes and the part of the servers'
// LDS inboxes of txng_kernel<> that does not fit this kernel's RQ slots.
uint64_t msim_txng4_extra_scratch_words(const msim_config &c) {
  return ((uint64_t)(c.n_nodes + 1) * c.inbox_capacity + (uint64_t)c.concurrency * G4_CLIENT_CAP + (uint64_t)c.n_nodes * G4_SLOTS) * 4;
}

hipError_t msim_launch_txng4(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < MSIM_TXNG4_MIN_CLUSTERS && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;
  G4Params rp;
  rp.k = kp; rp.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  rp.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;             // <= spill_capacity + inbox_capacity entries per server endpoint
  rp.client_spill = G4_CLIENT_CAP > RQ ? G4_CLIENT_CAP - RQ : 0;
  rp.client_spill_off = kp.spill_off + (uint64_t)(kp.N + 1) * rp.node_spill * 4;
  rp.xslots_off = rp.client_spill_off + (uint64_t)kp.CS * rp.client_spill * 4;
  size_t off = (size_t)RQ * 64 * 16;
  rp.ls = kp.CS / kp.N + 4u; if (rp.ls < 8u) rp.ls = 8u; if (rp.ls > G4_SLOTS) rp.ls = G4_SLOTS;
  rp.off_slots = (u32)off; off += (size_t)4 * kp.N * rp.ls * 16;
  rp.off_gen = (u32)off; off += (size_t)4 * 36 * 4;
  rp.off_misc = (u32)off; off += 64 * 4;
  rp.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  if (lds > 64 * 1024) return MSIM_LAYOUT_DOES_NOT_FIT;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(g4_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 3) / 4), block(64);
  if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((txng4_kernel<true, true>), grid, block, lds, st, rp); else hipLaunchKernelGGL((txng4_kernel<true, false>), grid, block, lds, st, rp); }
  else { if (rnd) hipLaunchKernelGGL((txng4_kernel<false, true>), grid, block, lds, st, rp); else hipLaunchKernelGGL((txng4_kernel<false, false>), grid, block, lds, st, rp); }
  return hipGetLastError();
}
