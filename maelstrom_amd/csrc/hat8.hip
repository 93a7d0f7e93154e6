// hat8.hip — EIGHT (or sixteen) txn-rw-register clusters per wavefront (the reference's own demo for the workload,
// core.clj:115-121 — `--node-count 2`, partitions, rate 100 — over demo/clojure/txn_rw_register_hat.clj).
//
// Same program and the same rounds as hat_kernel<> (sim_kernel_hat.inc): node = txn :120-130 (apply locally at a fresh [lamport, node]
// timestamp, last write wins per register, remember as unreplicated to every other node, txn_ok), the 100 ms replication thread :92-118,
// replicate :132-150, replicate_ack :152-172; client = workload/txn_rw_register.clj:108-139; generator = [upstream] elle rw-register gen —
// round for round what DESIGN.md §2.4 and the CPU oracle (oracle/hat_nodes.inc) specify.  What changes is the mapping: hat_kernel<> ran
// one cluster per wavefront — 2 live lanes of 64 for the demo shape — and paid a wavefront's instruction stream per cluster.  Here a
// cluster is a GROUP of GS lanes (8, or 4 for up to 4 nodes as an experiment; lane l of the group = node l + its client) and a wavefront carries
// 64 / GS clusters: what is uniform per CLUSTER lives in VGPRs (equal within a group), a "ballot" is the group's slice of the wave
// ballot, another lane's value comes by `ds_bpermute` within the group, the time reduction is two or three DPP steps (txn8.hip's scheme).
//
// Scope (engine.hip picks this kernel when all of it holds, else hat_kernel<> runs): net journal off, max-txn-length <= 4 (the default),
// at least 3200 clusters per node of a cluster in the launch (msim_launch_hat8 says why).
//
// LDS of a wavefront (slot-major: slot s of lane e at [s * 64 + e]): node queues (RQ envelopes, the rest spills to HBM: inbox_capacity +
// spill_capacity in all, the oracle's limit), client inboxes (CQ envelopes + HBM spill: 32 in all), per cluster the generator's key
// pool and the nemesis shuffle.  Registers, the txn table, the pending masks and the replicate lists live in HBM scratch exactly as in
// hat_kernel<>; history rows go straight to HBM.  What walks a LIST of txn slots — a ticking node's list of every txn it has not seen
// acknowledged (again every 100 ms: hundreds of them behind a partition), the receiver's replicate, the sender's replicate_ack — is done by
// the whole wavefront for one node at a time, 64 entries per step: as one lane's serial loop (four dependent HBM round trips per entry)
// it kept the other fifteen clusters of the wavefront waiting (first build: 127 ms per 16384 clusters against hat_kernel<>'s 84).
#include <hip/hip_runtime.h>

#include <cstdio>

#include "group8.h"
#include "layout_thresholds.h"

namespace {

constexpr u32 RQ = 4u;            // LDS envelopes per node queue
constexpr u32 CQ = 1u;            // LDS envelopes per client inbox
constexpr u32 H8_CLIENT_CAP = 32u;
constexpr int MM = 4;             // micro-ops per transaction (--max-txn-length <= 4, the default)
constexpr u32 SHORT = 8u;         // txn slots / list entries a node's own lane handles; longer ones go to the wavefront passes
enum { M_TXN = 23, M_TXN_OK = 24, M_REPLICATE_ACK = 27 };
enum { S_GEN3 = 3 };
constexpr u32 H8_TICK_US = 100000u;

struct H8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_gen, off_misc;                         // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;                          // HBM spill entries per node queue / client inbox
  u64 client_spill_off;                                  // word offset of the clients' spill area inside the per-instance scratch
  u32 round_limit;
};


template <int GS, bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) hat8_kernel(const H8Params up) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr u32 GM = (1u << GS) - 1u, NG = 64u / GS;
  const KParams &p = up.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane / GS, gbase = lane & ~(u32)(GS - 1);
  const u32 N = p.N;
  const bool is_node = l < N;
  const u32 inst_raw = blockIdx.x * NG + grp;
  const bool real = inst_raw < up.n_inst;
  const u32 inst = real ? inst_raw : up.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << l) - 1u;
  const u32 all_nodes = (1u << N) - 1u;
  const u32 max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz, mw = p.cfg.max_writes_per_key;
  const u32 K = p.cfg.max_values, G = max_rows / 2, area_cap = p.cfg.replication_words;
  const u32 round_limit = up.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  u32 *const g_kv = g_scr + (size_t)(is_node ? l : 0u) * K;                       // this node's registers
  u32 *const g_tab = g_scr + (size_t)N * K;                                        // [G][2]: ts, micro-ops ref
  unsigned char *const pend_all = reinterpret_cast<unsigned char *>(g_tab + 2 * (size_t)G);  // [N][G]
  unsigned char *const g_pend = pend_all + (size_t)(is_node ? l : 0u) * G;
  u32 *const g_area = g_tab + 2 * (size_t)G + ((size_t)N * G + 3) / 4;              // replicate lists
  const u32 qlane = is_node ? l : 0u;
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)qlane * up.node_spill;
  uint4 *const my_cspill = reinterpret_cast<uint4 *>(g_scr + up.client_spill_off) + (size_t)qlane * up.client_spill;
  const u32 my_spill_cap = is_node ? up.node_spill : 0u;

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + up.off_cq) + lane;                      // client inbox
  u32 *const gen = reinterpret_cast<u32 *>(smem + up.off_gen) + grp * 36;                       // active[16], next_val[16], next_key
  u32 *const misc = reinterpret_cast<u32 *>(smem + up.off_misc) + grp * GS;

  for (u32 i = l; i < 16; i += GS) { gen[i] = i; gen[16 + i] = 1; }
  if (l == 0) gen[32] = p.cfg.key_count;
  if (real) {
    for (u32 i = l; i < N * K; i += GS) g_scr[i] = 0;
    for (u32 i = l; i < ((size_t)N * G + 3) / 4; i += GS) g_tab[2 * (size_t)G + i] = 0;
  }
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & GM; };              // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 lamport = 0, lo = 0, npend = 0, timer_next = INF;
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_value = 0, process = l, m_value = 0, cin_n = 0, csp_n = 0;
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, nem_next = 0, nem_j = 0;
  u32 loss_on = 0, next_id = 0, n_rows = 0, n_payload = 0, flags = 0, rounds = 0;
  u32 n_txn = 0, n_area = 0;
  bool alive = real;

  #include "group8_net.inc"
  for (;;) {
    if (!__ballot(alive)) break;
    const u32 busy_mask = GB(busy);

    // ---- time-free phase transitions ----
    if (__ballot(alive && !(phase == PH_MAIN && ((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff))))) {
      for (;;) {
        bool ch = false;
        if (alive) {
          if (phase == PH_INIT_WAIT && !busy_mask) { phase = PH_MAIN_START; ch = true; }
          if (phase == PH_MAIN_START) { cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; nem_next = T; next_msg_id = 0; loss_on = 1; phase = PH_MAIN; ch = true; }
          if (phase == PH_MAIN && !((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff)) && !(rate == 0 && T < cutoff)) { phase = PH_DRAIN; ch = true; }
          if (phase == PH_DRAIN && !busy_mask) { phase = PH_DONE; ch = true; }   // no final phase (txn_rw_register.clj:162-166)
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
      const bool none_due = GB(deliver_at <= T || timer_next <= T) == 0;
      const bool jump = alive && due > T && none_due;
      if (__ballot(jump)) {
        u32 k = min(deliver_at, timer_next); k = k == INF ? INF : k * 2;
        if (busy) k = min(k, timeout_at * 2 + 1);
        u32 km = g8_min<GS>(k);
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
        else complete(MSIM_T_OK, 0, c_value);  // init_ok
      }
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) complete(MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_value);
    }
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
          // the transaction ([upstream] elle rw-register gen): lane 0 of the cluster writes the micro-ops and owns the key pool
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

      // ---- R3: one input per node: a due replication tick, else the due message ----
      bool rep = false; u32 dmask = 0;  // reply to the own client / node -> node sends (same type, a, b to every dest in dmask)
      u32 o_type = 0, o_a = 0, o_b = 0;
      const bool tick = normal && is_node && timer_next <= T;
      const bool msg = normal && is_node && !tick && deliver_at <= T;
      uint4 q = make_uint4(0, 0, 0, 0);
      if (msg) {
        q = cm; deliver_at = INF;
        if ((q.w >> 24) >= N) s_recv_cl++; else s_recv_sv++;
      }
      const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = msg ? (q.y & 0xFFu) : 0u, qa = q.z;

      const bool in_txn = qtype == M_TXN;
      const u32 need_pay = in_txn ? (qa >> 24) : 0u;
      // replicate-step!, :92-105: a ticking node sends the lowest node its oldest unreplicated txn still has to reach ALL the txns that
      // node has not acknowledged.  Finding the oldest one (the pending bytes before it were cleared by acknowledgements — hundreds at
      // once after a partition heals) and listing the others is done by the WHOLE wavefront, one ticking node at a time, 64 txn slots per
      // step; the lists go to n_area onward — the area's counter moves only if all of a cluster's lists of the round fit: engine
      // capacity, nothing is sent otherwise.
      u32 w_run = n_area, area_total = 0;
      // The usual tick (healthy network: a handful of txns are in flight) stays with the node's own lane: the SHORT pending bytes from
      // `lo` on come with one batch of independent loads, the lists of a cluster's nodes are laid out in node order.  Only a node with
      // more than that to look at goes to the wavefront pass below — one node at a time, each a chain of dependent round trips that the
      // other clusters of the wavefront would wait for.
      const bool tick_short = tick && n_txn - lo <= SHORT;
      if (__ballot(tick_short)) {
        u32 mk[SHORT];
#pragma unroll
        for (u32 t = 0; t < SHORT; t++) mk[t] = (tick_short && lo + t < n_txn) ? (u32)g_pend[lo + t] : 0u;
        u32 first = SHORT, d = 0;
#pragma unroll
        for (u32 t = SHORT; t-- > 0;) if (mk[t]) { first = t; d = (u32)__builtin_ctz(mk[t]); }
        u32 mine = 0;
#pragma unroll
        for (u32 t = 0; t < SHORT; t++) mine += (mk[t] >> d) & 1u;
        if (!tick_short || first == SHORT) mine = 0;
        u32 before = 0, total = 0;
        for (u32 sx = 0; sx < N; sx++) { const u32 v = GGET(mine, sx); before += sx < l ? v : 0u; total += v; }
        if (tick_short) {
          if (first == SHORT) { lo = n_txn; timer_next = INF; }
          else {
            u32 pos = w_run + before;
#pragma unroll
            for (u32 t = 0; t < SHORT; t++) if ((mk[t] >> d) & 1u) { if (pos < area_cap) g_area[pos] = (lo + t) | (mk[t] << 24); pos++; }
            o_type = M_REPLICATE; o_a = w_run + before; o_b = mine; dmask = 1u << d;
            lo += first; timer_next = T + H8_TICK_US;
          }
        }
        w_run += total; area_total += total;
      }
      for (u64 tw = __ballot(tick && !tick_short); tw; tw &= tw - 1) {
        const u32 j = (u32)__builtin_ctzll(tw);
        const u32 j_inst = rdlane(inst, j), lo_j = rdlane(lo, j), nt_j = rdlane(n_txn, j), w0 = rdlane(w_run, j), T_j = rdlane(T, j);
        u32 *const js = p.scratch + (size_t)j_inst * p.scratch_words;
        const unsigned char *const px = reinterpret_cast<const unsigned char *>(js + (size_t)N * K + 2 * (size_t)G) + (size_t)(j & (GS - 1u)) * G;
        u32 *const ja = js + (size_t)N * K + 2 * (size_t)G + ((size_t)N * G + 3) / 4;
        u32 first = nt_j, d_j = 0;
        for (u32 g0 = lo_j; g0 < nt_j; g0 += 64) {
          const u32 g = g0 + lane, m = g < nt_j ? (u32)px[g] : 0u;
          const u64 bal = __ballot(m != 0);
          if (bal) { const u32 src = (u32)__builtin_ctzll(bal); first = g0 + src; d_j = (u32)__builtin_ctz(rdlane(m, src)); break; }
        }
        if (lane == j) { lo = first; timer_next = first < nt_j ? T_j + H8_TICK_US : INF; }
        if (first >= nt_j) continue;
        u32 c = 0;
        for (u32 g0 = first; g0 < nt_j; g0 += 64) {
          const u32 g = g0 + lane, m = g < nt_j ? (u32)px[g] : 0u;
          const bool bit = ((m >> d_j) & 1u) != 0;
          const u64 bal = __ballot(bit);
          const u32 pos = w0 + c + (u32)__popcll(bal & ((1ull << lane) - 1ull));
          if (bit && pos < area_cap) ja[pos] = g | (m << 24);
          c += (u32)__popcll(bal);
        }
        if (lane == j) { o_type = M_REPLICATE; o_a = w0; o_b = c; dmask = 1u << d_j; }
        if (gbase == (j & ~(u32)(GS - 1))) { w_run += c; area_total += c; }   // (the lanes of j's group)
      }
      bool area_ok = true;
      if (area_total && n_area + area_total > area_cap) { flags |= MSIM_FLAG_ARENA_OVERRUN; area_ok = false; if (tick) { dmask = 0; o_type = 0; } }

      const u32 txn_mask = GB(in_txn);
      u32 pay_excl = 0, pay_total = 0;
      bool txn_ok = in_txn;
      if (__ballot(txn_mask != 0)) {
        for (u32 s = 0; s < N; s++) { const u32 v = GGET(need_pay, s); pay_excl += s < l ? v : 0u; pay_total += v; }
        if (txn_mask) {
          if (n_txn + __popc(txn_mask) > G || GB(in_txn && lamport >= (1u << 21) - 1)) { flags |= MSIM_FLAG_ARENA_OVERRUN; txn_ok = false; }
          else if (n_payload + pay_total > max_pay) { flags |= MSIM_FLAG_PAYLOAD_OVERFLOW; txn_ok = false; }
        }
      }

      if (__ballot(msg)) {
        if (tick) {
          // (its replicate, if any, was set up above)
        } else if (txn_ok) {  // :120-130
          const u32 g = n_txn + __popc(txn_mask & lt), off0 = qa & 0xFFFFFFu, n = qa >> 24;
          const u32 off = n_payload + pay_excl, ts = (lamport++ << 3) | l;
          u32 wv[MM], cur[MM];
#pragma unroll
          for (int j = 0; j < MM; j++) { wv[j] = 0; if ((u32)j < n) wv[j] = g_pay[off0 + (u32)j]; }
#pragma unroll
          for (int j = 0; j < MM; j++) { cur[j] = 0; if ((u32)j < n) cur[j] = g_kv[(wv[j] >> 1) & 0x7FFFu]; }
#pragma unroll
          for (int j = 0; j < MM; j++) if ((u32)j < n) {  // micro-ops in order: a read sees the transaction's own earlier writes
            const u32 w = wv[j], k = (w >> 1) & 0x7FFFu;
            u32 c = cur[j];
#pragma unroll
            for (int e = 0; e < j; e++) if ((wv[e] & 1u) && ((wv[e] >> 1) & 0x7FFFu) == k) c = cur[e];   // (cur[e] = what that write left)
            if (w & 1) {
              if (!(c && (c >> 8) > ts)) { c = (ts << 8) | ((w >> 16) & 0xFFu); g_kv[k] = c; }
              cur[j] = c;
              g_pay[off + (u32)j] = w;
            } else g_pay[off + (u32)j] = (k << 1) | ((c ? c & 0xFFu : 0xFFu) << 16);
          }
          g_tab[2 * g] = ts; g_tab[2 * g + 1] = qa;
          g_pend[g] = (unsigned char)(all_nodes & ~(1u << l));  // later-replicate!, :85-90
          if (npend++ == 0) { lo = g; if (timer_next == INF) timer_next = (T / H8_TICK_US + 1u) * H8_TICK_US; }
          rep = true; o_type = M_TXN_OK; o_a = off | (n << 24); o_b = qb;
        } else if (qtype == M_INIT) { rep = true; o_type = M_INIT_OK; o_b = qb; }
      }
      // replicate :132-150 and replicate_ack :152-172 walk a list of txn slots — hundreds after a partition heals: the WHOLE wavefront
      // takes one receiving node at a time, 64 list entries per step.  A register takes the write with the highest timestamp (last write
      // wins; the entries of a list are different txns, so their order does not matter: an atomic max on timestamp | value — two
      // writes of ONE txn to a register carry increasing values, the later one is the larger word); the entries name different slots,
      // so the pending bytes do not collide.
      {
        const bool is_rep = msg && qtype == M_REPLICATE, is_ack = msg && qtype == M_REPLICATE_ACK;
        // (a short list — nearly all of them on a healthy network — stays with the node's own lane, all clusters at once)
        const bool list_short = qb <= SHORT;
        if (__ballot(is_rep && list_short)) {
          if (is_rep && list_short) {
            for (u32 i = 0; i < qb; i++) {
              const u32 w = g_area[qa + i], g = w & 0xFFFFFFu, ts = g_tab[2 * g], ref = g_tab[2 * g + 1];
              lamport = max(lamport, (ts >> 3) + 1);
              const u32 off0 = ref & 0xFFFFFFu, n = ref >> 24;
              u32 wv[MM];
#pragma unroll
              for (int t = 0; t < MM; t++) { wv[t] = 0; if ((u32)t < n) wv[t] = g_pay[off0 + (u32)t]; }
#pragma unroll
              for (int t = 0; t < MM; t++) {  // apply-txn+ at the txn's own timestamp: last write wins
                if (!((u32)t < n) || !(wv[t] & 1)) continue;
                const u32 k = (wv[t] >> 1) & 0x7FFFu, c = g_kv[k];
                if (!(c && (c >> 8) > ts)) g_kv[k] = (ts << 8) | ((wv[t] >> 16) & 0xFFu);
              }
              const u32 rest = (w >> 24) & ~(1u << l);
              if (rest) {  // still pending elsewhere: this node relays it (its own entry for that timestamp is replaced)
                if (!g_pend[g]) { if (npend++ == 0) { lo = g; if (timer_next == INF) timer_next = (T / H8_TICK_US + 1u) * H8_TICK_US; } else lo = min(lo, g); }
                g_pend[g] = (unsigned char)rest;
              }
            }
            o_type = M_REPLICATE_ACK; o_a = qa; o_b = qb; dmask = all_nodes & ~(1u << l);
          }
        }
        if (__ballot(is_ack && list_short)) {
          if (is_ack && list_short) {
            for (u32 i = 0; i < qb; i++) {
              const u32 g = g_area[qa + i] & 0xFFFFFFu;
              u32 m = g_pend[g];
              if (!m) continue;  // txn already fully replicated
              m &= ~(1u << qsrc);
              g_pend[g] = (unsigned char)m;
              if (!m) npend--;
            }
            if (npend == 0) timer_next = INF;
          }
        }
        for (u64 lw = __ballot((is_rep || is_ack) && !list_short); lw; lw &= lw - 1) {
          const u32 j = (u32)__builtin_ctzll(lw), jl = j & (GS - 1u);
          const u32 j_inst = rdlane(inst, j), j_qa = rdlane(qa, j), j_qb = rdlane(qb, j), j_src = rdlane(qsrc, j);
          const bool j_ack = ((__ballot(is_ack) >> j) & 1ull) != 0;
          u32 *const js = p.scratch + (size_t)j_inst * p.scratch_words;
          u32 *const jkv = js + (size_t)jl * K;
          const u32 *const jtab = js + (size_t)N * K;
          unsigned char *const jp = reinterpret_cast<unsigned char *>(js + (size_t)N * K + 2 * (size_t)G) + (size_t)jl * G;
          const u32 *const ja = js + (size_t)N * K + 2 * (size_t)G + ((size_t)N * G + 3) / 4;
          const u32 *const jpay = p.payload + (size_t)j_inst * max_pay;
          if (!j_ack) {
            u32 lam = 0, newly = 0, lo_min = INF;
            for (u32 i0 = 0; i0 < j_qb; i0 += 64) {
              const u32 i = i0 + lane;
              if (i < j_qb) {
                const u32 w = ja[j_qa + i], g = w & 0xFFFFFFu, ts = jtab[2 * g], ref = jtab[2 * g + 1];
                lam = max(lam, (ts >> 3) + 1);
                const u32 off0 = ref & 0xFFFFFFu, n = ref >> 24;
                u32 wv[MM];
#pragma unroll
                for (int t = 0; t < MM; t++) { wv[t] = 0; if ((u32)t < n) wv[t] = jpay[off0 + (u32)t]; }
#pragma unroll
                for (int t = 0; t < MM; t++) if ((u32)t < n && (wv[t] & 1)) atomicMax(&jkv[(wv[t] >> 1) & 0x7FFFu], (ts << 8) | ((wv[t] >> 16) & 0xFFu));
                const u32 rest = (w >> 24) & ~(1u << jl);
                if (rest) {  // still pending elsewhere: this node relays it (its own entry for that timestamp is replaced)
                  if (!jp[g]) { newly++; lo_min = min(lo_min, g); }
                  jp[g] = (unsigned char)rest;
                }
              }
            }
            const u32 lam_w = ~wave_min(~lam), lo_w = wave_min(lo_min), newly_w = wave_sum(newly);
            if (lane == j) {
              lamport = max(lamport, lam_w);
              if (newly_w) {
                if (npend == 0) { lo = lo_w; if (timer_next == INF) timer_next = (T / H8_TICK_US + 1u) * H8_TICK_US; } else lo = min(lo, lo_w);
                npend += newly_w;
              }
              o_type = M_REPLICATE_ACK; o_a = qa; o_b = qb; dmask = all_nodes & ~(1u << l);
            }
          } else {
            u32 cleared = 0;
            for (u32 i0 = 0; i0 < j_qb; i0 += 64) {
              const u32 i = i0 + lane;
              if (i < j_qb) {
                const u32 g = ja[j_qa + i] & 0xFFFFFFu;
                u32 m = jp[g];
                if (m) {  // (else: txn already fully replicated)
                  m &= ~(1u << j_src);
                  jp[g] = (unsigned char)m;
                  if (!m) cleared++;
                }
              }
            }
            const u32 cleared_w = wave_sum(cleared);
            if (lane == j) { npend -= cleared_w; if (npend == 0) timer_next = INF; }
          }
        }
      }
      if (txn_mask && GB(txn_ok)) { n_txn += __popc(txn_mask); n_payload += pay_total; }
      if (area_ok) n_area += area_total;

      // COMMIT: ids in node order, then destination order
      bool c_arr = false; u32 ca_y = 0, ca_a = 0, ca_b = 0;
      {
        const u32 cnt = rep ? 1u : (u32)__popc(dmask);
        if (__ballot(cnt != 0)) {
          u32 my_off = 0, total = 0;
          for (u32 s = 0; s < N; s++) { const u32 v = GGET(cnt, s); my_off += s < l ? v : 0u; total += v; }
          if (rep) s_send_cl++; else s_send_sv += cnt;
          u32 ns = GB(dmask != 0);
          while (__ballot(ns != 0)) {  // node -> node: every receiver takes its envelope from each sender, in sender order
            const bool on = ns != 0;
            const u32 s = on ? (u32)__builtin_ctz(ns) : 0u; ns &= ns - 1u;
            const u32 dm = GGET(dmask, s), ty = GGET(o_type, s), a = GGET(o_a, s), b = GGET(o_b, s), off = GGET(my_off, s);
            if (on && is_node && ((dm >> l) & 1u)) arrive(next_id + off + __popc(dm & lt), ty, a, b, s);
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

      #include "group8_clients.inc"
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
  }

  // ---- epilogue ----
  u32 t_send_cl = 0, t_send_sv = 0, t_recv_cl = 0, t_recv_sv = 0;
  for (u32 s = 0; s < (u32)GS; s++) { t_send_cl += GGET(s_send_cl, s); t_send_sv += GGET(s_send_sv, s); t_recv_cl += GGET(s_recv_cl, s); t_recv_sv += GGET(s_recv_sv, s); }
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

template <int GS>
hipError_t h8_launch(const H8Params &up, uint32_t n, size_t lds, bool nem, bool rnd, hipStream_t st) {
  const dim3 grid((n + 64 / GS - 1) / (64 / GS)), block(64);
  if (nem) { if (rnd) hipLaunchKernelGGL((hat8_kernel<GS, true, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((hat8_kernel<GS, true, false>), grid, block, lds, st, up); }
  else { if (rnd) hipLaunchKernelGGL((hat8_kernel<GS, false, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((hat8_kernel<GS, false, false>), grid, block, lds, st, up); }
  return hipGetLastError();
}

}  // namespace

// Whether several clusters per wavefront simulate this configuration (see the header of this file).
bool msim_hat8_eligible(const msim_config &c) {
  return c.node_program == MSIM_NODE_TXN_RW_HAT && c.journal_capacity == 0 && c.n_nodes >= 2 && c.n_nodes <= 8 && c.concurrency == c.n_nodes &&
         c.max_txn_length <= (uint32_t)MM;   // micro-ops in registers
}

// Extra per-instance scratch words behind the queues' spill area: the clients' spill, and what of the LDS queues of hat_kernel<> does
// not fit this kernel's RQ slots.
uint64_t msim_hat8_extra_scratch_words(const msim_config &c) {
  return ((uint64_t)c.n_nodes * c.inbox_capacity + (uint64_t)c.n_nodes * H8_CLIENT_CAP) * 4;
}

hipError_t msim_launch_hat8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  // A small batch is better off with one cluster per wavefront: eight per wavefront are an eighth of the wavefronts, a wavefront's run is
  // a chain of dependent round trips (the longer the more nodes a cluster has), and below ~1024 wavefronts — one per SIMD — nothing hides
  // it.  Measured crossovers (profiles/r03ae_hat8.txt; both kernels with the wavefront-wide list passes): 2 nodes 8192 clusters (37.8
  // against 44.9 ms; 16384: 44.5 / 85.7), 3 nodes between 8192 and 16384 (63 / 53, 75 / 103), 5 nodes near 16384 (112 / 128; 8192:
  // 97 / 68).  MSIM_DEV_FLAGS bit 10 asks for this layout whatever the batch.
  if (n < MSIM_HAT8_MIN_CLUSTERS_PER_NODE * c.n_nodes && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;   // (6400 / 9600 / 16000 clusters of 2 / 3 / 5 nodes)
  H8Params up;
  up.k = kp; up.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  up.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;
  up.client_spill = H8_CLIENT_CAP - CQ;
  up.client_spill_off = kp.spill_off + (uint64_t)kp.N * up.node_spill * 4;
  size_t off = (size_t)RQ * 64 * 16;
  up.off_cq = (u32)off; off += (size_t)CQ * 64 * 16;
  up.off_gen = (u32)off; off += (size_t)16 * 36 * 4;
  off = (off + 15) & ~(size_t)15;
  up.off_misc = (u32)off; off += 64 * 4;
  up.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  if (kp.dev_flags & 0x1000u) std::fprintf(stderr, "[hat8] %u clusters, several per wavefront, %zu B of LDS per wavefront\n", n, lds);   // developer trace bit
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(g8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  // 8-lane groups for every cluster size: 4-lane groups halve the wavefronts again, and a wavefront's run is a chain of dependent round
  // trips that only other wavefronts hide (16384 clusters of 2 nodes: 58 ms in 4-lane groups, 45 ms in 8-lane groups, 84 ms one per
  // wavefront; at 65536 clusters 139 / 143 / 317 ms).  MSIM_DEV_FLAGS bit 15 selects the 4-lane groups.
  const bool gs4 = c.n_nodes <= 4 && (kp.dev_flags & 0x8000u);
  return gs4 ? h8_launch<4>(up, n, lds, c.nemesis_mask != 0, rnd, st) : h8_launch<8>(up, n, lds, c.nemesis_mask != 0, rnd, st);
}
