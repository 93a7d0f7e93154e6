// dtg4.hip — FOUR clusters of the Datomic-style txn-list-append node with SEVERAL WORKERS PER NODE per wavefront: demo/ruby/datomic_list_append.rb
// as the reference's own runs invoke it (doc/05-datomic/01-single-node.md:257,322: one node, --concurrency 10n), in 16-lane groups.
//
// Same program and the same rounds as dtg_kernel<> (sim_kernel_dtg.inc; specification: oracle/dt_nodes.inc): the node's lock and its arrival-order
// waiting queue, the persistent hash tree in lww-kv, the root pointer in lin-kv, Promise#await's 5 s — its handlers are repeated here statement for
// statement (the node's index is its lane in the group).  What changes is the mapping, as in dtg4.hip / svc4.hip (whose time / scheduler / client
// machinery this file shares line for line): a cluster is n nodes + its worker slots + lin-kv + lww-kv <= 16 endpoints, one lane each of a 16-lane
// group — 1 node with 10 workers is 13 — and a wavefront carries four clusters.
//
// Scope (engine.hip picks this kernel when all of it holds, else dtg_kernel<> runs): concurrency a multiple of n above n, n + concurrency + 2 <= 16,
// net journal off, at least MSIM_DTG4_MIN_CLUSTERS clusters in the launch (half of that with two nodes or more).
//
// LDS of a wavefront: envelope queues slot-major (RQ envelopes per endpoint, the rest spills to HBM), per node the lock holder's cursor, the waiting
// ring of 64 transactions and the save stack (DG_WORDS words), per cluster the generator's key pool and the nemesis shuffle.  Tree records, write
// lists, the cas tables and the append log live in HBM scratch (dt_kernel<>'s layout); history rows go straight to HBM.
//
// Envelope (16 B): x = deadline, y = (id << 8) | type, z = a, w = b | (src << 24); src = the sender's lane in its group (lin-kv: n + slots, lww-kv: + 1).
#include <hip/hip_runtime.h>

#include "sim_kernels.h"
#include "layout_thresholds.h"

namespace {

__constant__ u32 d4_log2_q24[257];

constexpr u32 GS = 16u;           // lanes per cluster
#ifndef D4_RQ
#define D4_RQ 2u
#endif
#ifndef D4_WAVES
#define D4_WAVES 4
#endif
constexpr u32 RQ = D4_RQ;         // LDS envelopes per endpoint
constexpr u32 D4_CLIENT_CAP = 32u;   // Reusable lin-kv clients (lin_kv.clj:74-76) collect late replies between RPCs (the oracle's limit)
struct D4Params {
  KParams k;
  u32 n_inst;
  u32 off_curs, off_gen, off_misc;   // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;               // HBM spill entries per server endpoint / client behind the RQ LDS slots
  u64 client_spill_off;                       // word offset of the clients' spill area inside the per-instance scratch
  u32 round_limit;
};

__device__ __forceinline__ u32 d4_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = d4_log2_q24[idx], l1 = d4_log2_q24[idx + 1];
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
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(D4_WAVES))) dtg4_kernel(const D4Params rp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = rp.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 4, gbase = lane & 48u;
  const u32 N = p.N, C = p.C, CS = p.CS;
  const u32 LIN = N + CS;                                                     // lane of lin-kv in the group; lww-kv is LIN + 1
  const bool is_node = l < N;
  const bool is_client = l >= N && l < N + CS;
  const bool is_lin = l == LIN, is_lww = l == LIN + 1u;
  const bool is_server = is_node || is_lin || is_lww;   // endpoints that poll all the time and see latency
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
  const u32 TC = p.mk_tcap;   // tree nodes a node may create
  const u32 rpc_timeout_ms = p.cfg.client_timeout_ms;
  const u32 round_limit = rp.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  u32 *const g_kv = g_scr;                                           // [max_values][mw]: element | version << 8
  u32 *const g_kvn = g_kv + (size_t)mv * mw;                         // [max_values]
  u32 *const g_first = g_kvn + mv;                                   // [max_values] version at which the key entered the tree (DT_NONE: never)
  unsigned char *const g_hash = reinterpret_cast<unsigned char *>(g_first + mv);   // [max_values] Tree.hash of the key
  u32 *const g_rec = g_scr + (((size_t)mv * mw + 2u * mv + (mv + 3u) / 4u + 3u) & ~(size_t)3);   // [N][TC][DT_RW] tree nodes by pointer, on a 16-byte boundary
  u32 *const g_wl = g_rec + (size_t)N * TC * DT_RW;                  // [N][DT_MAXW] the pointers a node writes this round
  u32 *const g_cas = g_wl + (size_t)N * DT_MAXW;                     // [N][DT_CASQ] x {msg_id, from, transaction}: what a node's cas requests carry beside `to`
  const u32 my_spill_cap = is_server ? rp.node_spill : (is_client ? rp.client_spill : 0u);
  uint4 *const my_spill = is_server ? reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)(is_node ? l : N + (l - LIN)) * rp.node_spill
                                    : reinterpret_cast<uint4 *>(g_scr + rp.client_spill_off) + (size_t)(is_client ? slot : 0) * rp.client_spill;

  // LDS
  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                         // slot s at my_q[s * 64]
  u32 *const curs_g = reinterpret_cast<u32 *>(smem + rp.off_curs) + grp * N * DG_WORDS;                 // [node of the group][DG_WORDS]
  u32 *const cu = curs_g + (is_node ? l : 0) * DG_WORDS;
  u32 *const my_wl = g_wl + (size_t)(is_node ? l : 0) * DT_MAXW;
  u32 *const gpool = reinterpret_cast<u32 *>(smem + rp.off_gen) + grp * 36;                             // active[16], next_val[16], next_key
  u32 *const misc = reinterpret_cast<u32 *>(smem + rp.off_misc) + grp * GS;

  for (u32 i = lane; i < 4 * N * DG_WORDS; i += 64) reinterpret_cast<u32 *>(smem + rp.off_curs)[i] = 0;
  gpool[l] = l; gpool[16 + l] = 1;
  if (l == 0) gpool[32] = p.cfg.key_count;
  if (real) {
    for (u32 i = l; i < mv; i += GS) { g_kvn[i] = 0; g_first[i] = DT_NONE; g_hash[i] = (unsigned char)dt_hash(i); }
    for (u32 i = l; i < N * DT_CASQ * 3u; i += GS) g_cas[i] = 0;
  }
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- endpoint state ----
  bool has_c = false; u32 deliver_at = 0; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 node_msgid = 0;
  u32 next_p = 0;                                      // node: @ptr (:332, :352-355)
  u32 wait_until = INF;                                // node: when the lock holder's Promise#await gives up (promise.rb:5,17-30), INF: not waiting
  u32 casn = 0;                                        // node: cas requests so far
  u32 root = 0, root_exists = 0, cur_v = 0;            // lin-kv lane: the root pointer; versions so far
  u32 svc_ctr = 0;                                     // lww-kv lane: rand-int draws so far
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
    if ((src < N || src >= LIN) && is_server) {  // neither end is a client (util.clj:7-16)
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * d4_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
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
  auto visible = [&](u32 k, u32 from) -> u32 {
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
    if (alive && GB((my_flags & MSIM_FLAG_ARENA_OVERRUN) != 0)) alive = false;   // an engine capacity was exceeded: what follows would not be the program's behaviour

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
    my_t = min(my_t, wait_until);   // (a node's timer is a normal event)
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

      // ---- R3: one input per node, then one for each service (endpoint order: lin-kv, lww-kv) ----
      bool rep = false, svc_rep = false;   // node -> a client, service -> node
      u32 r_to = 0, r_type = 0, r_a = 0, r_b = 0;    // the answer to the client
      u32 n_out = 0, o_dest = 0;           // node -> service: n_out messages, all to the same service; one in registers (o1_*) or DT_MAXW writes in my_wl[]
      u32 o1_type = 0, o1_a = 0, o1_b = 0, o_wlo = 0;
      u32 o_type = 0, o_a = 0, o_b = 0, o_to = 0, need_words = 0, done_ref = 0, done_rv = 0;   // service -> node; the completed transaction's payload
      auto rec_of = [&](u32 ptr) -> u32 * { return g_rec + ((size_t)(ptr >> 20) * TC + (ptr & 0xFFFFFu)) * DT_RW; };
      auto is_new = [&](u32 ptr) -> bool { return (ptr >> 20) == l && (ptr & 0xFFFFFu) >= cu[DC_PSTART]; };
      auto has_key = [&](u32 k) -> bool {   // the key is in the lineage of the working tree
        if (g_first[k] <= cu[DC_RV]) return true;   // (DT_NONE is above every version)
        const u32 no = cu[DC_NOWN];
        for (u32 i = 0; i < no; i++) if (cu[DC_OWN + i] == k) return true;
        return false;
      };
      auto br_index = [&](u32 w0, u32 h) -> u32 {   // branch_index (:231-247) with the split's bounds (:170-181)
        const u32 lo = (w0 >> 8) & 0xFFu, hi = (w0 >> 16) & 0xFFu, bs = (hi - lo) / 8u;
        for (u32 i = 0; i < 7u; i++) if (h < lo + (i + 1u) * bs) return i;
        return 7u;
      };
      auto send1 = [&](u32 dest, u32 type, u32 a, u32 b) { o_dest = dest; n_out = 1; o1_type = type; o1_a = a; o1_b = b; };
      auto reply = [&](u32 type, u32 a, u32 cmsg) { rep = true; r_type = type; r_a = a; r_to = cmsg >> 24; r_b = cmsg & 0xFFFFFFu; };   // (cmsg = the client's msg_id | its endpoint << 24)
      auto start_txn = [&](u32 cmsg, u32 ref) {   // the lock is ours: current_tree (:358-365)
        cu[DC_STAGE] = DS_ROOT; cu[DC_CMSG] = cmsg; cu[DC_REF] = ref; cu[DC_J] = 0; cu[DC_NOWN] = 0;
        const u32 rid = ++node_msgid; cu[DC_RPC] = rid;
        send1(D_LIN, M_READ, 0, rid);
        wait_until = T + DT_AWAIT_US;
      };
      auto unlock = [&]() {   // the next waiting transaction takes the lock (:348, :371), in arrival order
        cu[DC_STAGE] = DS_IDLE;
        wait_until = INF;
        const u32 wq = cu[DG_WQN], cnt = wq & 0xFFu, head = wq >> 8;
        if (cnt) {
          const u32 cmsg = cu[DG_WQ + 2u * head], ref = cu[DG_WQ + 2u * head + 1u];
          cu[DG_WQN] = (cnt - 1u) | (((head + 1u) & (DG_WAITQ - 1u)) << 8);
          start_txn(cmsg, ref);
        }
      };
      auto load = [&](u32 ptr) {   // Tree.load with a cache miss (:83-101)
        const u32 rid = ++node_msgid;
        cu[DC_STAGE] = DS_LOAD; cu[DC_TARGET] = ptr; cu[DC_RPC] = rid;
        send1(D_LWW, M_READ, ptr, rid);
        wait_until = T + DT_AWAIT_US;
      };
      // walks to the key's leaf; the first tree node on the way that has to be fetched, DT_NONE if the path is in memory
      auto descend = [&](u32 k) -> u32 {
        const u32 h = g_hash[k];
        u32 pt = cu[DC_T];
        for (u32 d = 0; d < DT_MAXDEPTH; d++) {
          const u32 *const r = rec_of(pt);
          const u32 w0 = r[0], w3 = __hip_atomic_load(r + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (word 3 is the one word of a record that changes after its creation, by L2 atomics: read past the L1)
          u32 ch[8];
#pragma unroll
          for (u32 c = 0; c < 8u; c++) ch[c] = r[4u + c];
          if (!is_new(pt) && !((w3 >> (2u + l)) & 1u)) return pt;   // neither created by this transaction nor loaded by this node
          if ((w0 & 1u) == 0u) return DT_NONE;
          const u32 ci = br_index(w0, h);
          pt = ch[0];
#pragma unroll
          for (u32 c = 1; c < 8u; c++) pt = c == ci ? ch[c] : pt;
        }
        my_flags |= MSIM_FLAG_ARENA_OVERRUN;
        return DT_NONE;
      };
      // assoc (:158-197, :256-268) along a path that is in memory (sim_kernel_dt.inc)
      auto assoc = [&](u32 k) {
        const u32 h = g_hash[k];
        u32 n = 0, pt = cu[DC_T];
        for (; n + 1u < DT_MAXDEPTH; n++) {
          const u32 *const r = rec_of(pt);
          const u32 w0 = r[0];
          u32 ch[8];
#pragma unroll
          for (u32 c = 0; c < 8u; c++) ch[c] = r[4u + c];
          if ((w0 & 1u) == 0u) break;
          const u32 ci = br_index(w0, h);
          pt = ch[0];
#pragma unroll
          for (u32 c = 1; c < 8u; c++) pt = c == ci ? ch[c] : pt;
        }
        const u32 *const lf = rec_of(pt);
        const u32 lw0 = lf[0], lcount = lf[1];
        if (lw0 & 1u) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; return; }
        const bool has = has_key(k);
        const u32 L = (has || lcount < 8u) ? 1u : 9u, base = next_p, ver = cu[DC_RV] + 1u;
        if (base + L + n >= TC) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; return; }   // engine capacity
        const u32 lo = (lw0 >> 8) & 0xFFu, hi = (lw0 >> 16) & 0xFFu;
        auto put = [&](u32 idx, u32 w0, u32 cnt) -> u32 * { u32 *const r = g_rec + ((size_t)l * TC + idx) * DT_RW; r[0] = w0; r[1] = cnt; r[2] = ver; __hip_atomic_store(r + 3, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return r; };   // (word 3: lww-kv replica in bits 0-1, 3 = not written; bit 2 + i: node i has loaded it)
        if (L == 1u) put(base + 1u, lw0, lcount + (has ? 0u : 1u));
        else {   // eight leaves under a new branch: the lineage's keys of this range (and the new one) by sub-range
          const u32 bs = (hi - lo) / 8u, nk = gpool[32];
          u64 c_lo = 0, c_hi = 0;   // 4 x 16-bit counters each
          for (u32 q = 0; q < nk; q++) {
            if (q != k && !has_key(q)) continue;
            const u32 hq = g_hash[q];
            if (hq < lo || hq >= hi) continue;
            const u32 ci = bs ? min((hq - lo) / bs, 7u) : 7u;
            if (ci < 4u) c_lo += 1ull << (16u * ci); else c_hi += 1ull << (16u * (ci - 4u));
          }
          u32 *const br = put(base + 9u, 1u | (lo << 8) | (hi << 16), 0u);
          for (u32 i = 0; i < 8u; i++) {
            const u32 b_lo = lo + i * bs, b_hi = i == 7u ? hi : b_lo + bs;
            const u32 cnt = (u32)((i < 4u ? c_lo >> (16u * i) : c_hi >> (16u * (i - 4u))) & 0xFFFFu);
            put(base + 1u + i, (b_lo << 8) | (b_hi << 16), cnt);
            br[4u + i] = (l << 20) | (base + 1u + i);
          }
        }
        pt = cu[DC_T];
        for (u32 i = 0; i < n; i++) {   // a copy of every branch above, pointing at the new child
          const u32 *const r = rec_of(pt);
          const u32 w0 = r[0], ci = br_index(w0, h);
          u32 ch[8];
#pragma unroll
          for (u32 c = 0; c < 8u; c++) ch[c] = r[4u + c];
          u32 *const nr = put(base + L + (n - i), w0, 0u);
          const u32 child_new = (l << 20) | (i + 1u == n ? base + L : base + L + (n - i - 1u));
#pragma unroll
          for (u32 c = 0; c < 8u; c++) nr[4u + c] = c == ci ? child_new : ch[c];
          pt = ch[0];
#pragma unroll
          for (u32 c = 1; c < 8u; c++) pt = c == ci ? ch[c] : pt;
        }
        next_p = base + L + n;
        cu[DC_T] = (l << 20) | next_p;
        if (!has) { const u32 no = cu[DC_NOWN]; if (no < 8u) { cu[DC_OWN + no] = k; cu[DC_NOWN] = no + 1u; } }
      };
      // save! (:212-224, :291-320): the new tree nodes the final tree reaches, children before their parent (sim_kernel_dt.inc)
      auto save = [&]() {
        u32 *const stk = cu + DG_STK;
        u32 sp = 1, wn = 0;
        const u32 wlo = node_msgid + 1u;
        stk[0] = cu[DC_T]; stk[1] = 0x100u;   // (0x100: not looked at yet)
        while (sp) {
          const u32 pt = stk[2u * (sp - 1u)];
          u32 mask = stk[2u * (sp - 1u) + 1u];
          const u32 *const r = rec_of(pt);
          const u32 w0 = r[0];
          u32 ch[8];
#pragma unroll
          for (u32 c = 0; c < 8u; c++) ch[c] = r[4u + c];
          if (mask & 0x100u) {
            mask = 0;
            if (w0 & 1u) {
#pragma unroll
              for (u32 c = 0; c < 8u; c++) mask |= is_new(ch[c]) ? 1u << c : 0u;
            }
          }
          if (mask) {
            const u32 ci = (u32)__builtin_ctz(mask);
            u32 nxt = ch[0];
#pragma unroll
            for (u32 c = 1; c < 8u; c++) nxt = c == ci ? ch[c] : nxt;
            stk[2u * (sp - 1u) + 1u] = mask & (mask - 1u);
            if (sp > DT_MAXDEPTH) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; stk[2u * (sp - 1u) + 1u] = 0; continue; }
            stk[2u * sp] = nxt; stk[2u * sp + 1u] = 0x100u; sp++;
            continue;
          }
          if (wn >= DT_MAXW) my_flags |= MSIM_FLAG_ARENA_OVERRUN; else my_wl[wn++] = pt;
          sp--;
        }
        node_msgid += wn;
        cu[DC_STAGE] = DS_SAVE; cu[DC_WLO] = wlo; cu[DC_WN] = wn; cu[DC_WOUT] = wn;
        o_dest = D_LWW; n_out = wn; o_wlo = wlo;
        wait_until = T + DT_AWAIT_US;   // `tree2.save!.await` (:366)
      };
      auto reply_txn_ok = [&]() {   // the completed transaction: its reads see the version read + its own appends
        reply(M_TXN_OK, 0, cu[DC_CMSG]);
        done_ref = cu[DC_REF]; done_rv = cu[DC_RV];
        const u32 off0 = done_ref & 0xFFFFFFu, n = done_ref >> 24;
        for (u32 j = 0; j < n; j++) {
          const u32 w = g_pay[off0 + j], k = (w >> 1) & 0x7FFFu;
          need_words++;
          if (!(w & 1u)) {
            u32 len = visible(k, done_rv);
            for (u32 e = 0; e < j; e++) { const u32 we = g_pay[off0 + e]; if ((we & 1u) && ((we >> 1) & 0x7FFFu) == k) len++; }
            need_words += (len + 3u) / 4u;
          }
        }
      };
      // apply_txn (:391-415) from micro-op j on; stops at the first tree node that has to be fetched
      auto apply = [&]() {
        const u32 ref = cu[DC_REF], off0 = ref & 0xFFFFFFu, n = ref >> 24;
        u32 j = cu[DC_J];
        while (j < n) {
          const u32 w = g_pay[off0 + j], k = (w >> 1) & 0x7FFFu;
          const u32 miss = descend(k);   // t[k] — for an append too (:405)
          if (miss != DT_NONE) { cu[DC_J] = j; load(miss); return; }
          if (w & 1u) assoc(k);
          j++;
        }
        cu[DC_J] = j;
        if (cu[DC_T] == cu[DC_P1]) { reply_txn_ok(); unlock(); return; }   // nothing appended: no write, no cas
        save();
      };

      const bool await_over = is_node && normal && wait_until <= T;   // a node's due timer comes before its due message
      const bool take = is_server && normal && !await_over && has_c && deliver_at <= T;
      if (await_over) {   // Promise#await gave up (promise.rb:24-29): RPCError.timeout => error 0 to the client (node.rb:172), the lock is free
        reply(M_ERROR, 0, cu[DC_CMSG]);
        unlock();
      } else if (take) {
        const uint4 q = cm; has_c = false;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N && qsrc < LIN) s_recv_cl++; else s_recv_sv++;
        if (is_node) {
          const u32 st = cu[DC_STAGE];
          switch (qtype) {
            case M_INIT:
              if (l != 0u) { reply(M_INIT_OK, 0, qb | (qsrc << 24)); break; }
              {   // the first node writes the initial state (:337-345): Tree.empty, then the root pointer
                u32 *const r = g_rec;
                r[0] = (128u << 16); r[1] = 0; r[2] = 0; __hip_atomic_store(r + 3, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 rid = ++node_msgid;
                cu[DC_STAGE] = DS_INIT_LEAF; cu[DC_CMSG] = qb | (qsrc << 24); cu[DC_RPC] = rid;
                send1(D_LWW, M_WRITE, 0, rid);
              } break;
            case M_TXN:
              if (st == DS_IDLE) start_txn(qb | (qsrc << 24), qa);
              else { const u32 wq = cu[DG_WQN], cnt = wq & 0xFFu;
                if (cnt == DG_WAITQ) my_flags |= MSIM_FLAG_ARENA_OVERRUN;
                else { const u32 sl = ((wq >> 8) + cnt) & (DG_WAITQ - 1u); cu[DG_WQ + 2u * sl] = qb | (qsrc << 24); cu[DG_WQ + 2u * sl + 1u] = qa; cu[DG_WQN] = wq + 1u; } }
              break;
            case M_READ_OK: case M_WRITE_OK: case M_CAS_OK: case M_ERROR:
              switch (st) {
                case DS_INIT_LEAF:
                  if (qb != cu[DC_RPC]) break;
                  { const u32 rid = ++node_msgid; cu[DC_STAGE] = DS_INIT_ROOT; cu[DC_RPC] = rid; send1(D_LIN, M_WRITE, 0, rid); }
                  break;
                case DS_INIT_ROOT:
                  if (qb != cu[DC_RPC]) break;
                  cu[DC_STAGE] = DS_IDLE; reply(M_INIT_OK, 0, cu[DC_CMSG]);
                  break;
                case DS_ROOT:
                  if (qb != cu[DC_RPC]) break;
                  if (qtype != M_READ_OK) { reply(M_ERROR, 14, cu[DC_CMSG]); unlock(); break; }   // "Unsure how to handle" (:364)
                  cu[DC_P1] = qa; cu[DC_T] = qa; cu[DC_PSTART] = next_p + 1u;
                  { const u32 *const rr = rec_of(qa); const u32 rv2 = rr[2], rw3 = __hip_atomic_load(rr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cu[DC_RV] = rv2; if ((rw3 >> (2u + l)) & 1u) apply(); else load(qa); }
                  break;
                case DS_LOAD:
                  if (qb != cu[DC_RPC]) break;
                  if (qtype == M_READ_OK) { atomicOr(rec_of(cu[DC_TARGET]) + 3, 1u << (2u + l)); apply(); }   // @@cache[ptr] = tree (:95)
                  else load(cu[DC_TARGET]);   // "Retrying read of tree node" (:97)
                  break;
                case DS_SAVE:
                  if (qb < cu[DC_WLO] || qb >= cu[DC_WLO] + cu[DC_WN]) break;
                  { const u32 left = cu[DC_WOUT] - 1u; cu[DC_WOUT] = left;
                    if (left == 0u) { const u32 rid = ++node_msgid; cu[DC_STAGE] = DS_CAS; cu[DC_RPC] = rid;   // advance_root! (:376-388): cas root from the pointer read to the new one
                      { u32 *const ce = g_cas + ((size_t)l * DT_CASQ + (casn++ % DT_CASQ)) * 3u; ce[0] = rid; ce[1] = cu[DC_P1]; ce[2] = cu[DC_REF]; }
                      send1(D_LIN, M_CAS, cu[DC_T], rid); wait_until = T + DT_AWAIT_US; } }
                  break;
                case DS_CAS:
                  if (qb != cu[DC_RPC]) break;
                  if (qtype == M_CAS_OK) reply_txn_ok();
                  else reply(M_ERROR, 30, cu[DC_CMSG]);   // txn_conflict (:385)
                  unlock();
                  break;
                default: break;   // "Ignoring reply ... with no callback" (node.rb:160-162)
              }
              break;
            default: break;
          }
        } else if (is_lin) {   // lin-kv over the key "root" (service.clj:31-61)
          svc_rep = true; o_to = qsrc; o_b = qb;
          if (qtype == M_READ) {
            if (!root_exists) { o_type = M_ERROR; o_a = 20; } else { o_type = M_READ_OK; o_a = root; }
          } else if (qtype == M_WRITE) { root = qa; root_exists = 1u; o_type = M_WRITE_OK; o_a = 0; }
          else {   // cas, no create_if_not_exists: self-contained (:376-388) — `from` and the transaction under the msg_id in the sender's table
            u32 c_from = 0, c_ref = 0; bool c_hit = false;
            { const u32 *const ce = g_cas + (size_t)qsrc * DT_CASQ * 3u;
#pragma unroll
              for (u32 i = 0; i < DT_CASQ; i++) { const u32 e0 = ce[3u * i], e1 = ce[3u * i + 1u], e2 = ce[3u * i + 2u]; if (e0 == qb) { c_hit = true; c_from = e1; c_ref = e2; } } }
            if (!c_hit) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; o_type = M_ERROR; o_a = 22; }   // engine capacity: DT_CASQ outstanding cas requests per node
            else if (!root_exists) { o_type = M_ERROR; o_a = 20; }
            else if (root != c_from) { o_type = M_ERROR; o_a = 22; }
            else {
              const u32 ref = c_ref, off0 = ref & 0xFFFFFFu, n = ref >> 24, v = ++cur_v;
              root = qa;
              for (u32 i = 0; i < n; i++) { const u32 w = g_pay[off0 + i];
                if (w & 1u) { const u32 k = (w >> 1) & 0x7FFFu, c = g_kvn[k]; if (g_first[k] == DT_NONE) g_first[k] = v;
                  g_kv[k * mw + c] = ((w >> 16) & 0xFFu) | (v << 8); g_kvn[k] = c + 1u; } }
              o_type = M_CAS_OK; o_a = 0;
            }
          }
        } else {   // lww-kv (service.clj:214-243 as written): merge-source, merge-dest, then the replica that serves the request
          svc_rep = true; o_to = qsrc; o_b = qb;
          svc_ctr += 2u;
          const u32 r = scale32(draw32(key, 12u /* S_LIN */, svc_ctr++), 2);
          u32 *const rp = rec_of(qa) + 3;   // (the replica bits; the nodes set their "loaded" bits in the same word: atomics)
          if (qtype == M_WRITE) { atomicAnd(rp, ~3u); atomicOr(rp, r); o_type = M_WRITE_OK; o_a = qa; }
          else if ((__hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 3u) == r) { o_type = M_READ_OK; o_a = qa; }
          else { o_type = M_ERROR; o_a = 20; }
        }
      }

      // completed transactions: payload words allocated in node order, each node writes its own
      if (__ballot(need_words != 0)) {
        const u32 incl = row_scan(need_words);
        const u32 total = GGET(incl, GS - 1u);
        if (total) {
          if (n_payload + total > max_pay) { flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; if (need_words) r_a = 0; }
          else {
            if (need_words) {
              const u32 off0 = done_ref & 0xFFFFFFu, n = done_ref >> 24;
              u32 pp = n_payload + incl - need_words;
              r_a = pp | (need_words << 24);
              for (u32 j = 0; j < n; j++) {
                const u32 w = g_pay[off0 + j], k = (w >> 1) & 0x7FFFu;
                if (w & 1u) { g_pay[pp++] = w; continue; }
                const u32 vis = visible(k, done_rv);
                u32 e = 0, acc = 0;
                const u32 hdr = pp++;
                for (u32 i = 0; i < vis; i++) { acc |= (g_kv[k * mw + i] & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } }
                for (u32 i = 0; i < j; i++) { const u32 wi = g_pay[off0 + i];
                  if ((wi & 1u) && ((wi >> 1) & 0x7FFFu) == k) { acc |= ((wi >> 16) & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } } }
                if (e & 3) g_pay[pp++] = acc;
                g_pay[hdr] = (k << 1) | ((e ? e : 0xFFu) << 16);  // a key without elements reads nil
              }
            }
            n_payload += total;
          }
        }
      }

      // COMMIT: ids in lane order (nodes, lin-kv, lww-kv); a node's messages in the order it emitted them: the answer to a client,
      // then what the next step sends to a service
      {
        const u32 rcnt = rep ? 1u : 0u;
        const u32 cnt = is_node ? rcnt + n_out : (svc_rep ? 1u : 0u);
        if (__ballot(cnt != 0)) {
          const u32 incl = row_scan(cnt);
          const u32 total = GGET(incl, GS - 1u);
          const u32 my_off = incl - cnt;
          if (is_node) { s_send_cl += rcnt; s_send_sv += n_out; } else s_send_sv += cnt;
          __syncthreads();   // (the write lists of this round are in HBM scratch; a workgroup is one wavefront)
          u32 ts = GB(is_node && cnt != 0);
          while (__ballot(ts != 0)) {   // every sending node in turn: its answer to a client (taken in by that client's lane), then its messages to a service (taken in by the service's lane)
            const bool on = ts != 0;
            const u32 s = on ? (u32)__builtin_ctz(ts) : 0u; ts &= ts - 1u;
            const u32 rc = GGET(rcnt, s), off = GGET(my_off, s);
            const u32 to = GGET(r_to, s), ty = GGET(r_type, s), a = GGET(r_a, s), b = GGET(r_b, s);
            if (on && rc && l == to) arrive(next_id + off, ty, a, b, s);
            const u32 kn = GGET(n_out, s);
            const u32 dst = GGET(o_dest, s), t1 = GGET(o1_type, s), a1 = GGET(o1_a, s), b1 = GGET(o1_b, s), wlo = GGET(o_wlo, s);
            if (on && kn && l == LIN + dst) {
              if (wlo == 0u) arrive(next_id + off + rc, t1, a1, b1, s);
              else { const u32 *const wl = g_wl + (size_t)s * DT_MAXW;
                for (u32 k = 0; k < kn; k++) arrive(next_id + off + rc + k, M_WRITE, wl[k], wlo + k, s); }
            }
          }
          // service -> node
          u32 sv = GB(svc_rep);
          while (__ballot(sv != 0)) {
            const bool on = sv != 0;
            const u32 s = on ? (u32)__builtin_ctz(sv) : 0u; sv &= sv - 1u;
            const u32 ty = GGET(o_type, s), a = GGET(o_a, s), b = GGET(o_b, s), d = GGET(o_to, s), off = GGET(my_off, s);
            if (on && l == d) arrive(next_id + off, ty, a, b, s);
          }
          next_id += total;
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
bool msim_dtg4_eligible(const msim_config &c) {
  if (c.node_program != MSIM_NODE_TXN_DATOMIC || c.journal_capacity != 0 || c.concurrency <= c.n_nodes) return false;
  return c.n_nodes >= 1 && c.n_nodes + c.concurrency + 2 <= GS;
}

// Extra per-instance scratch words the layout needs behind dtg_kernel<>'s spill area: the clients' whole inboxes and the part of the servers'
// LDS inboxes of dtg_kernel<> that does not fit this kernel's RQ slots.
uint64_t msim_dtg4_extra_scratch_words(const msim_config &c) {
  return ((uint64_t)(c.n_nodes + 2) * c.inbox_capacity + (uint64_t)c.concurrency * D4_CLIENT_CAP) * 4;
}

hipError_t msim_launch_dtg4(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < (kp.N == 1 ? MSIM_DTG4_MIN_CLUSTERS : MSIM_DTG4_MIN_CLUSTERS / 2u) && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;   // (two nodes and more: even from 8192 on, profiles/r06f_dtg4_threshold_sweep.jsonl)
  D4Params rp;
  rp.k = kp; rp.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  rp.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;             // <= spill_capacity + inbox_capacity entries per server endpoint
  rp.client_spill = D4_CLIENT_CAP > RQ ? D4_CLIENT_CAP - RQ : 0;
  rp.client_spill_off = kp.spill_off + (uint64_t)(kp.N + 2) * rp.node_spill * 4;
  size_t off = (size_t)RQ * 64 * 16;
  rp.off_curs = (u32)off; off += (size_t)4 * kp.N * DG_WORDS * 4;
  rp.off_gen = (u32)off; off += (size_t)4 * 36 * 4;
  rp.off_misc = (u32)off; off += 64 * 4;
  rp.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  if (lds > 64 * 1024) return MSIM_LAYOUT_DOES_NOT_FIT;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(d4_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 3) / 4), block(64);
  if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((dtg4_kernel<true, true>), grid, block, lds, st, rp); else hipLaunchKernelGGL((dtg4_kernel<true, false>), grid, block, lds, st, rp); }
  else { if (rnd) hipLaunchKernelGGL((dtg4_kernel<false, true>), grid, block, lds, st, rp); else hipLaunchKernelGGL((dtg4_kernel<false, false>), grid, block, lds, st, rp); }
  return hipGetLastError();
}
