// crdt8.hip — EIGHT g-set / pn-counter / g-counter clusters per wavefront (the reference's demos
// `g_set.rb` and `pn_counter.rb`, core.clj:107-108, at the default five nodes).
//
// Same programs and the same rounds as sim_kernel_colo<MSIM_NODE_G_SET / MSIM_NODE_PN_COUNTER, ...> (sim_kernel_colo.inc): node = the CRDT
// with a 5 s replicate timer (doc/04-crdts/01-g-set.md: add -> into the local set, read -> the whole set, every 5 s the whole state to
// every other node, replicate -> union; demo/ruby/pn_counter.rb: a pair of G-counters merged by element-wise maximum); client =
// workload/g_set.clj:33-60 / pn_counter.clj:48-83; generators = g_set.clj:62-66, pn_counter.clj:134-135, g_counter.clj:37-41; final reads
// after the heal and the quiesce period (:final? true for the counters, pn_counter.clj:137) — round for round what DESIGN.md §2 and the
// CPU oracle specify.  What changes is the mapping (txn8.hip's scheme, as in hat8.hip / uid8.hip): a cluster is a group of 8 lanes (lane l =
// node l + its client), what is uniform per cluster lives in VGPRs, ballots are the group's slice, `ds_bpermute` within the group.
//
// Scope (engine.hip picks this kernel when all of it holds, else the colocated kernel runs): at most 8 nodes, one worker per node, net
// journal off, a node's state of at most 64 words (2048 set elements), at least 4096 clusters in the launch (MSIM_DEV_FLAGS bit 10 asks
// for the layout whatever the batch).
//
// LDS of a wavefront (slot-major: slot s of lane e at [s * 64 + e]): node queues (RQ envelopes, the rest spills to HBM), client inboxes (2
// envelopes), the nodes' states (W words each), the nemesis shuffle.  Replicate snapshots live in HBM scratch where the colocated kernel
// keeps them; history rows and read_ok payloads go straight to HBM.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "group8.h"
#include "layout_thresholds.h"

namespace {

constexpr u32 GS = 8u;            // lanes per cluster
constexpr u32 RQ = 4u;            // LDS envelopes per node queue
constexpr u32 CQ = CLIENT_INBOX_CAP;   // envelopes per client inbox (all of them in LDS)
constexpr u32 WMAX = 64u;         // words of a node's state (2048 set elements)

struct C8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_seen, off_misc;                        // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;                          // HBM spill entries per node queue / client inbox (clients: none)
  u64 client_spill_off;
  u32 round_limit;
};


template <bool PN, bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) crdt8_kernel(const C8Params up) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr u32 GM = (1u << GS) - 1u, NG = 64u / GS;
  const KParams &p = up.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane / GS, gbase = lane & ~(GS - 1u);
  const u32 N = p.N, W = p.W;
  const bool is_node = l < N;
  const u32 inst_raw = blockIdx.x * NG + grp;
  const bool real = inst_raw < up.n_inst;
  const u32 inst = real ? inst_raw : up.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << l) - 1u;
  const u32 all_nodes = (1u << N) - 1u;
  const u32 max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words, max_values = p.cfg.max_values;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz;
  const bool g_counter = p.cfg.workload == MSIM_WL_G_COUNTER;
  const u32 round_limit = up.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;   // replicate snapshots: [tick][node][W]
  const u32 qlane = is_node ? l : 0u;
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)qlane * up.node_spill;
  uint4 *const my_cspill = reinterpret_cast<uint4 *>(g_scr + up.client_spill_off);   // (never used: client_spill = 0)
  const u32 my_spill_cap = is_node ? up.node_spill : 0u;

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + up.off_cq) + lane;                      // client inbox
  u32 *const my_seen = reinterpret_cast<u32 *>(smem + up.off_seen) + lane * W;                  // this node's state (word-major per lane)
  u32 *const misc = reinterpret_cast<u32 *>(smem + up.off_misc) + grp * GS;

  for (u32 w = 0; w < W; w++) my_seen[w] = 0;
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & GM; };              // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 timer_next = INF, tick = 0;
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 timeout_at = 0, next_msg_id = 0, c_f = 0, c_value = 0, c_final = 0, process = l, m_f = 0, m_value = 0, m_final = 0, cin_n = 0, csp_n = 0;
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, nem_next = 0, nem_j = 0, next_value = 0, sleep_until = 0;
  u32 loss_on = 0, next_id = 0, n_rows = 0, n_payload = 0, flags = 0, rounds = 0;
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
          if (phase == PH_DRAIN && !busy_mask) {   // heal (nemesis), quiesce, final reads (g_set.clj / pn_counter.clj:137)
            phase = NEM ? PH_NEM_FINAL : PH_SLEEP;
            if (phase == PH_SLEEP) sleep_until = T + p.cfg.quiesce_ms * 1000u;
            ch = true;
          }
          if (phase == PH_FINAL_WAIT && !busy_mask) { phase = PH_DONE; ch = true; }
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
        u32 km = g8_min<8>(k);
        if (due != INF) km = min(km, due * 2);
        if (jump) {
          if (km == INF) { flags |= MSIM_FLAG_ROUND_LIMIT; alive = false; }
          else { timeout_round = (km & 1) != 0; T = max(T, km >> 1); }
        }
      }
    }

    bool inv_row = false; u32 inv_packed = 0, inv_value = 0;
    bool cmp_row = false; u32 cmp_packed = 0, cmp_value = 0, cmp_len = 0;
    u32 nem_rows = 0, nem_f = 0, nem_v1 = 0, nem_v2 = 0, nem_len2 = 0;

    auto complete = [&](u32 type, u32 err, u32 value, u32 len) {
      busy = false;
      if (kind != K_OP) { if (type != MSIM_T_OK) my_flags |= MSIM_FLAG_ROUND_LIMIT; return; }
      cmp_row = true; cmp_packed = type | (c_f << 2) | (err << 7) | (c_final << 11) | (process << 12);
      cmp_value = value; cmp_len = len;
      if (type == MSIM_T_INFO) { process += N; next_msg_id = 0; cin_n = 0; }  // crashed process, fresh client
    };
    // the client's recv! consumes one envelope (client.clj:94-107)
    auto client_deliver = [&](u32 qtype, u32 qa, u32 qb) {
      s_recv_cl++;
      if (busy && qb == next_msg_id) {
        if (qtype == M_READ_OK) { if (PN) complete(MSIM_T_OK, 0, qa, 0); else complete(MSIM_T_OK, 0, qa & 0xFFFFFFu, qa >> 24); }
        else complete(MSIM_T_OK, 0, c_value, 0);
      }
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) complete(MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_f == MSIM_F_READ ? MSIM_NO_VALUE : c_value, 0);
    }
    bool normal = alive && !timeout_round;   // this cluster runs R1-R4 in this wave-round
    if (__ballot(normal)) {
      // ---- R1: scheduler ----
      const bool act = normal && due <= T;
      if (__ballot(act && phase != PH_MAIN)) {
        if (act && phase == PH_INIT) { if (is_node) { mark = true; kind = K_INIT; } phase = PH_INIT_WAIT; }
        else if (NEM && act && phase == PH_NEM_FINAL) {
          part = 0; nem_rows = 2; nem_f = MSIM_F_STOP_PARTITION; nem_v1 = MSIM_NO_VALUE; nem_v2 = MSIM_NO_VALUE; nem_len2 = 0;
          phase = PH_SLEEP; sleep_until = T + p.cfg.quiesce_ms * 1000u;
        } else if (act && (phase == PH_SLEEP || phase == PH_FINAL)) {   // (PH_SLEEP acts when its sleep is over: due = sleep_until)
          if (is_node) { mark = true; kind = K_OP; m_f = MSIM_F_READ; m_value = MSIM_NO_VALUE; m_final = PN ? 1u : 0u; }
          phase = PH_FINAL_WAIT;
        }
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
          u32 f, val = MSIM_NO_VALUE;
          bool ok = true;
          if (PN && g_counter) {
            // g_counter.clj:37-41: (gen/filter ...) skips negative adds and takes the mix's next op at once
            u32 rr = r_lo, a = 0;
            int d = (int)((((rr >> 4) & 0xFFFFu) * 10u) >> 16) - 5;
            while (!(rr & 1) && d < 0 && a < 15) { a++; rr = (u32)draw64(key, S_GEN2, (u64)kk * 16 + a); d = (int)((((rr >> 4) & 0xFFFFu) * 10u) >> 16) - 5; }
            if ((rr & 1) || d < 0) f = MSIM_F_READ; else { f = MSIM_F_ADD; val = (u32)d; }
          }
          else if (r_lo & 1) f = MSIM_F_READ;
          else {
            f = MSIM_F_ADD;
            if (PN) val = (u32)((int)((((r_lo >> 4) & 0xFFFFu) * 10u) >> 16) - 5);  // (- (rand-int 10) 5), pn_counter.clj:134-135
            else if (gen_on && next_value >= max_values) ok = false;
            else val = next_value;
          }
          if (gen_on && !ok) { flags |= MSIM_FLAG_VALUES_OVERFLOW; phase = PH_DONE; alive = false; normal = false; }
          else if (gen_on) {
            if (!PN && f == MSIM_F_ADD) next_value++;
            if (sel) { mark = true; kind = K_OP; m_f = f; m_value = val; m_final = 0; }
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
            c_f = m_f; c_value = m_value; c_final = m_final;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (c_f << 2) | (c_final << 11) | (process << 12); inv_value = c_value;
            rq_type = c_f == MSIM_F_ADD ? (u32)M_ADD : (u32)M_READ; rq_a = c_f == MSIM_F_READ ? 0u : c_value;
          }
          const u32 want = ++next_msg_id;
          timeout_at = T + (kind == K_OP ? p.cfg.client_timeout_ms : 10000u) * 1000u;
          s_send_cl++;
          arrive(next_id + __popc(inv_mask & lt), rq_type, rq_a, want, N + l);
        }
        next_id += __popc(inv_mask);
        poll();
      }

      // ---- R3: one input per node: a due replicate timer, else the due message ----
      bool rep = false, rd = false; u32 dmask = 0;  // reply to the own client / replicate to every other node
      u32 o_type = 0, o_a = 0, o_b = 0;
      const bool tick_now = normal && is_node && timer_next <= T;
      const bool msg = normal && is_node && !tick_now && deliver_at <= T;
      if (tick_now) {
        timer_next = T + 5000000u;
        u32 *const snap = g_scr + ((size_t)tick * N + l) * W;
        for (u32 w = 0; w < W; w++) snap[w] = my_seen[w];
        dmask = all_nodes & ~(1u << l); o_type = M_REPLICATE; o_a = tick; tick++;
      } else if (msg) {
        const uint4 q = cm; deliver_at = INF;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N) s_recv_cl++; else s_recv_sv++;
        switch (qtype) {
          case M_INIT: timer_next = T; rep = true; o_type = M_INIT_OK; o_b = qb; break;
          case M_READ:
            rep = true; o_type = M_READ_OK; o_b = qb;
            if (PN) { u32 v = 0; for (u32 i = 0; i < N; i++) v += my_seen[i] - my_seen[N + i]; o_a = v; }  // increments - decrements
            else rd = true;
            break;
          case M_ADD:
            if (PN) { const int d = (int)qa; if (d >= 0) my_seen[l] += (u32)d; else my_seen[N + l] += (u32)(-d); }  // own slot of inc / dec
            else my_seen[qa >> 5] |= 1u << (qa & 31);
            rep = true; o_type = M_ADD_OK; o_a = qa; o_b = qb; break;
          case M_REPLICATE: {
            const u32 *const snap = g_scr + ((size_t)qa * N + qsrc) * W;
            for (u32 w0 = 0; w0 < W; w0 += 8) {   // eight words per round trip
              u32 v[8];
#pragma unroll
              for (u32 t = 0; t < 8; t++) v[t] = snap[min(w0 + t, W - 1u)];
#pragma unroll
              for (u32 t = 0; t < 8; t++) { const u32 i = min(w0 + t, W - 1u); my_seen[i] = PN ? max(my_seen[i], v[t]) : (my_seen[i] | v[t]); }
            }
          } break;
          default: break;
        }
      }
      // read_ok of a set: the words in use go to the payload area, the readers of a cluster in node order
      if (!PN) {
        const u32 rdm = GB(rd);
        if (__ballot(rd)) {
          const u32 words = (next_value + 31) >> 5;
          const u32 mine = __popc(rdm & lt);
          const bool fits = n_payload + (mine + 1u) * words <= max_pay;   // (allocated reader by reader: the first ones may still fit)
          if (rd) {
            u32 off = 0;
            if (!fits) my_flags |= MSIM_FLAG_PAYLOAD_OVERFLOW;
            else { off = n_payload + mine * words; for (u32 w = 0; w < words; w++) g_pay[off + w] = my_seen[w]; }
            o_a = off | (words << 24);
          }
          n_payload += __popc(GB(rd && fits)) * words;
        }
      }

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
        if (wr && inv_row) out[nem_rows + __popc(imask & lt)] = make_uint4(tlo, thi, inv_packed, inv_value);
        if (wr && cmp_row) out[nem_rows + ni + __popc(cmask & lt)] = make_uint4(tlo, thi | (cmp_len << 16), cmp_packed, cmp_value);
        n_rows = wr ? n_rows + nr : n_rows;
      }
    }
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
    p.meta[inst] = m;
  }
}



template <bool PN>
hipError_t c8_launch(const C8Params &up, uint32_t n, size_t lds, bool nem, bool rnd, hipStream_t st) {
  const dim3 grid((n + 7) / 8), block(64);
  if (nem) { if (rnd) hipLaunchKernelGGL((crdt8_kernel<PN, true, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((crdt8_kernel<PN, true, false>), grid, block, lds, st, up); }
  else { if (rnd) hipLaunchKernelGGL((crdt8_kernel<PN, false, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((crdt8_kernel<PN, false, false>), grid, block, lds, st, up); }
  return hipGetLastError();
}

}  // namespace

// Whether eight clusters per wavefront simulate this configuration (see the header of this file).
bool msim_crdt8_eligible(const msim_config &c) {
  return (c.node_program == MSIM_NODE_G_SET || c.node_program == MSIM_NODE_PN_COUNTER) && c.journal_capacity == 0 && c.n_nodes >= 1 && c.n_nodes <= GS &&
         c.concurrency == c.n_nodes && c.max_values / 32u <= WMAX;
}

// Extra per-instance scratch words behind the queues' spill area: what of the LDS queues of the colocated kernel does not fit this
// kernel's RQ slots.
uint64_t msim_crdt8_extra_scratch_words(const msim_config &c) { return (uint64_t)c.n_nodes * c.inbox_capacity * 4; }

hipError_t msim_launch_crdt8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < MSIM_CRDT8_MIN_CLUSTERS && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;   // (see the header)
  C8Params up;
  up.k = kp; up.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  up.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;
  up.client_spill = 0;
  up.client_spill_off = kp.spill_off;
  size_t off = (size_t)RQ * 64 * 16;
  up.off_cq = (u32)off; off += (size_t)CQ * 64 * 16;
  up.off_seen = (u32)off; off += (size_t)64 * kp.W * 4;
  off = (off + 15) & ~(size_t)15;
  up.off_misc = (u32)off; off += 64 * 4;
  up.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  if (kp.dev_flags & 0x1000u) std::fprintf(stderr, "[crdt8] %u clusters, eight per wavefront, %zu B of LDS per wavefront\n", n, lds);   // developer trace bit
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(g8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  return c.node_program == MSIM_NODE_PN_COUNTER ? c8_launch<true>(up, n, lds, c.nemesis_mask != 0, rnd, st) : c8_launch<false>(up, n, lds, c.nemesis_mask != 0, rnd, st);
}
