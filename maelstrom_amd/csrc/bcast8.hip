// bcast8.hip — EIGHT broadcast clusters per wavefront (the four broadcast programs of doc/03-broadcast at the
// cluster sizes the tutorial and the reference's demo run them at — five nodes, core.clj:105).
//
// Same programs and the same rounds as sim_kernel_colo<MSIM_NODE_BCAST_*, ...> (sim_kernel_colo.inc): fire-and-forget gossip with and without
// skip-sender (01-broadcast.md:525-547, 02-performance.md:61-67), acknowledged gossip with a one-second retry per message
// (02-performance.md:22-28 / 03-broadcast "ack + retry"), rpc to every other node; client = workload/broadcast.clj:190-231 (topology, then
// broadcast / read ops, reads idempotent), generator = broadcast.clj:233-240 with the final reads after the heal and the quiesce period —
// round for round what DESIGN.md §2 and the CPU oracle specify.  The headline layout (duo.hip) carries two clusters of up to 32 nodes per
// wavefront and takes only the fire-and-forget programs on a healthy network; everything else ran one cluster per wavefront.  Here a
// cluster is a group of 8 lanes (lane l = node l + its client) and a wavefront carries eight clusters (txn8.hip's scheme, as in
// crdt8.hip): with loss, partitions, the acknowledged programs, at the tutorial's cluster sizes.
//
// Scope (engine.hip picks this kernel when all of it holds, else duo / the colocated kernel run): at most 8 nodes, one worker per node, net
// journal off, a node's set of at most 64 words (2048 values), at least 12288 clusters in the launch (eight per wavefront are an eighth of
// the wavefronts and a wavefront's run is a chain of dependent steps: measured at 5 nodes, rate 100, 20 s, partitions — fire-and-forget
// 4096 clusters 23.7 ms against 14.8 one per wavefront, 16384: 26.8 / 47.6; acknowledged gossip 138 / 88 and 154 / 219,
// profiles/r03ar_bcast8.txt; MSIM_DEV_FLAGS bit 10 asks for the layout whatever the batch); the fire-and-forget programs on a healthy
// network stay with duo.hip (16384 clusters: 20.4 ms at latency 0 against 20.9 here, 39.4 / 27.8 at 10 ms) unless bit 15 is set.
//
// LDS of a wavefront (slot-major: slot s of lane e at [s * 64 + e]): node queues (RQ envelopes, the rest spills to HBM), client inboxes (2
// envelopes), the nodes' sets (W words each), the nemesis shuffle.  The unacknowledged-destination masks and the retry FIFO of the
// acknowledged program live in HBM scratch where the colocated kernel keeps them; rows and read_ok payloads go straight to HBM.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "group8.h"
#include "layout_thresholds.h"

namespace {

constexpr u32 GS = 8u;            // lanes per cluster
constexpr u32 RQ = 4u;            // LDS envelopes per node queue
constexpr u32 CQ = CLIENT_INBOX_CAP;   // envelopes per client inbox (all of them in LDS)
constexpr u32 WMAX = 64u;         // words of a node's set (2048 values)

struct B8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_seen, off_misc;                        // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;                          // HBM spill entries per node queue / client inbox (clients: none)
  u64 client_spill_off;
  u32 round_limit;
};


template <int PROG, bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) bcast8_kernel(const B8Params up) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr u32 GM = (1u << GS) - 1u, NG = 64u / GS;
  constexpr bool IS_RPC = PROG == MSIM_NODE_BCAST_ACK_RETRY || PROG == MSIM_NODE_BCAST_RPC_ALL, IS_ACK = PROG == MSIM_NODE_BCAST_ACK_RETRY;
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
  const u32 round_limit = up.round_limit;
  const u32 adj = is_node ? topo_adj(p.cfg.topology, N, l) : 0u;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  const u32 qlane = is_node ? l : 0u;
  u32 *const g_unacked = g_scr + (size_t)qlane * max_values;                                  // (acknowledged gossip) value -> destinations that have not acknowledged it
  u32 *const g_fifo = g_scr + (size_t)N * max_values + (size_t)qlane * max_values * 2;          // (acknowledged gossip) {value, retry time} in send order
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)qlane * up.node_spill;
  uint4 *const my_cspill = reinterpret_cast<uint4 *>(g_scr + up.client_spill_off);   // (never used: client_spill = 0)
  const u32 my_spill_cap = is_node ? up.node_spill : 0u;

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + up.off_cq) + lane;                      // client inbox
  u32 *const my_seen = reinterpret_cast<u32 *>(smem + up.off_seen) + lane * W;                  // this node's set
  u32 *const misc = reinterpret_cast<u32 *>(smem + up.off_misc) + grp * GS;

  for (u32 w = 0; w < W; w++) my_seen[w] = 0;
  if (IS_ACK && real && is_node) for (u32 v = 0; v < max_values; v++) g_unacked[v] = 0;
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & GM; };              // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 node_msgid = 0, fifo_head = 0, fifo_tail = 0, retry_time = INF;
  u32 fifo_hi = 0, fifo_ti = 0;   // (fifo_head / fifo_tail modulo max_values, kept by compare-and-wrap: a division by a run-time value is forty instructions in the round loop)
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
          if (phase == PH_INIT_WAIT && !busy_mask) { phase = PH_TOPO; ch = true; }
          if (phase == PH_TOPO_WAIT && !busy_mask) { phase = PH_MAIN_START; ch = true; }
          if (phase == PH_MAIN_START) { cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; nem_next = T; next_msg_id = 0; loss_on = 1; phase = PH_MAIN; ch = true; }
          if (phase == PH_MAIN && !((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff)) && !(rate == 0 && T < cutoff)) { phase = PH_DRAIN; ch = true; }
          if (phase == PH_DRAIN && !busy_mask) {   // heal (nemesis), quiesce, final reads (broadcast.clj:233-240)
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
    if (phase == PH_INIT || phase == PH_TOPO || phase == PH_NEM_FINAL || phase == PH_FINAL) due = T;
    else if (phase == PH_SLEEP) due = sleep_until;
    else if (phase == PH_MAIN) {
      if (nem_live) due = max(nem_next, T);
      if (gen_live && free_mask) due = min(due, max(gen_next, T));
      if (rate == 0 && !nem_live) due = min(due, cutoff);
    }
    bool timeout_round = false;
    {
      const bool none_due = GB(deliver_at <= T || retry_time <= T) == 0;
      const bool jump = alive && due > T && none_due;
      if (__ballot(jump)) {
        u32 k = min(deliver_at, retry_time); k = k == INF ? INF : k * 2;
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
        if (qtype == M_READ_OK) complete(MSIM_T_OK, 0, qa & 0xFFFFFFu, qa >> 24);
        else complete(MSIM_T_OK, 0, c_value, 0);
      }
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) complete(c_f == MSIM_F_READ ? MSIM_T_FAIL : MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_f == MSIM_F_READ ? MSIM_NO_VALUE : c_value, 0);   // (a read is idempotent: :fail)
    }
    bool normal = alive && !timeout_round;   // this cluster runs R1-R4 in this wave-round
    if (__ballot(normal)) {
      // ---- R1: scheduler ----
      const bool act = normal && due <= T;
      if (__ballot(act && phase != PH_MAIN)) {
        if (act && phase == PH_INIT) { if (is_node) { mark = true; kind = K_INIT; } phase = PH_INIT_WAIT; }
        else if (act && phase == PH_TOPO) { if (is_node) { mark = true; kind = K_TOPO; } phase = PH_TOPO_WAIT; }
        else if (NEM && act && phase == PH_NEM_FINAL) {
          part = 0; nem_rows = 2; nem_f = MSIM_F_STOP_PARTITION; nem_v1 = MSIM_NO_VALUE; nem_v2 = MSIM_NO_VALUE; nem_len2 = 0;
          phase = PH_SLEEP; sleep_until = T + p.cfg.quiesce_ms * 1000u;
        } else if (act && (phase == PH_SLEEP || phase == PH_FINAL)) {   // (PH_SLEEP acts when its sleep is over: due = sleep_until)
          if (is_node) { mark = true; kind = K_OP; m_f = MSIM_F_READ; m_value = MSIM_NO_VALUE; m_final = 1u; }   // (:final? true, broadcast.clj:240)
          phase = PH_FINAL_WAIT;
        }
      }
      #include "group8_nemesis.inc"
      {
        const bool gen_on = act && phase == PH_MAIN && gen_live && gen_next <= T && free_mask != 0;
        if (__ballot(gen_on)) {
          const u32 nfree = __popc(free_mask);
          const u64 h = draw64(key, S_GEN, gen_k);
          const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          const u32 pick = scale32(r_lo, nfree);
          const bool sel = gen_on && is_node && !busy && (u32)__popc(free_mask & lt) == pick;
          const bool is_rd = (r_lo & 1u) != 0;
          if (gen_on && !is_rd && next_value >= max_values) { flags |= MSIM_FLAG_VALUES_OVERFLOW; phase = PH_DONE; alive = false; normal = false; }
          else if (gen_on) {
            if (sel) { mark = true; kind = K_OP; m_f = is_rd ? (u32)MSIM_F_READ : (u32)MSIM_F_BROADCAST; m_value = is_rd ? MSIM_NO_VALUE : next_value; m_final = 0; }
            if (!is_rd) next_value++;
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
          else if (kind == K_TOPO) { rq_type = M_TOPOLOGY; next_msg_id = 0; }
          else {
            c_f = m_f; c_value = m_value; c_final = m_final;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (c_f << 2) | (c_final << 11) | (process << 12); inv_value = c_value;
            rq_type = c_f == MSIM_F_BROADCAST ? (u32)M_BROADCAST : (u32)M_READ; rq_a = c_f == MSIM_F_READ ? 0u : c_value;
          }
          const u32 want = ++next_msg_id;
          timeout_at = T + (kind == K_OP ? p.cfg.client_timeout_ms : 10000u) * 1000u;
          s_send_cl++;
          arrive(next_id + __popc(inv_mask & lt), rq_type, rq_a, want, N + l);
        }
        next_id += __popc(inv_mask);
        poll();
      }

      // ---- R3: one input per node: a due retry (acknowledged gossip), else the due message ----
      bool rep = false, rd = false; u32 rep_dest = 0, rep_type = 0, rep_a = 0, rep_b = 0;   // a reply: to the own client (dest >= N) or to a node
      u32 dmask = 0, fan_a = 0, fan_b0 = 0;                                                  // a fan-out of M_BROADCAST: destinations, value, first msg_id (rpc)
      const bool retry_now = IS_ACK && normal && is_node && retry_time <= T;
      const bool msg = normal && is_node && !retry_now && deliver_at <= T;
      if (retry_now) {
        const u32 v = g_fifo[fifo_hi * 2];
        fifo_head++; fifo_hi = fifo_hi + 1 == max_values ? 0u : fifo_hi + 1;
        const u32 un = g_unacked[v];
        if (un) {
          dmask = un; fan_a = v; fan_b0 = node_msgid + 1; node_msgid += __popc(un);
          const u32 ts = fifo_ti * 2;
          g_fifo[ts] = v; g_fifo[ts + 1] = T + 1000000u; fifo_tail++; fifo_ti = fifo_ti + 1 == max_values ? 0u : fifo_ti + 1;
        }
        retry_time = fifo_head < fifo_tail ? g_fifo[fifo_hi * 2 + 1] : INF;
      } else if (msg) {
        const uint4 q = cm; deliver_at = INF;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N) s_recv_cl++; else s_recv_sv++;
        switch (qtype) {
          case M_INIT: rep = true; rep_dest = qsrc; rep_type = M_INIT_OK; rep_b = qb; break;
          case M_TOPOLOGY: rep = true; rep_dest = qsrc; rep_type = M_TOPOLOGY_OK; rep_b = qb; break;
          case M_READ: rep = true; rep_dest = qsrc; rep_type = M_READ_OK; rep_b = qb; rd = true; break;
          case M_BROADCAST: {
            const u32 v = qa, bitm = 1u << (v & 31);
            const u32 wv = my_seen[v >> 5];
            if (!(wv & bitm)) {
              my_seen[v >> 5] = wv | bitm;
              u32 tg = PROG == MSIM_NODE_BCAST_RPC_ALL ? (all_nodes & ~(1u << l)) : adj;
              if (PROG != MSIM_NODE_BCAST_FF_ECHOBACK && qsrc < N) tg &= ~(1u << qsrc);
              dmask = tg; fan_a = v;
              if (IS_RPC) { fan_b0 = node_msgid + 1; node_msgid += __popc(tg); }
              if (IS_ACK && tg) {
                g_unacked[v] = tg;
                const u32 ts = fifo_ti * 2;
                g_fifo[ts] = v; g_fifo[ts + 1] = T + 1000000u;
                if (fifo_head == fifo_tail) retry_time = T + 1000000u;
                fifo_tail++; fifo_ti = fifo_ti + 1 == max_values ? 0u : fifo_ti + 1;
              }
            }
            if (qb != 0) { rep = true; rep_dest = qsrc; rep_type = M_BROADCAST_OK; rep_a = v; rep_b = qb; }
          } break;
          case M_BROADCAST_OK: if (IS_ACK) g_unacked[qa] &= ~(1u << qsrc); break;
          default: break;
        }
      }
      // read_ok: the words in use go to the payload area, the readers of a cluster in node order
      {
        const u32 rdm = GB(rd);
        if (__ballot(rd)) {
          const u32 words = (next_value + 31) >> 5;
          const u32 mine = __popc(rdm & lt);
          const bool fits = n_payload + (mine + 1u) * words <= max_pay;   // (allocated reader by reader: the first ones may still fit)
          if (rd) {
            u32 off = 0;
            if (!fits) my_flags |= MSIM_FLAG_PAYLOAD_OVERFLOW;
            else { off = n_payload + mine * words; for (u32 w = 0; w < words; w++) g_pay[off + w] = my_seen[w]; }
            rep_a = off | (words << 24);
          }
          n_payload += __popc(GB(rd && fits)) * words;
        }
      }

      // COMMIT: ids in node order; a node's reply before its fan-out in the acknowledged program, after it otherwise; a fan-out in
      // destination order
      bool c_arr = false; u32 ca_y = 0, ca_a = 0, ca_b = 0;
      {
        const u32 fan_cnt = (u32)__popc(dmask);
        const u32 cnt = fan_cnt + (rep ? 1u : 0u);
        if (__ballot(cnt != 0)) {
          u32 my_off = 0, total = 0;
          for (u32 s = 0; s < N; s++) { const u32 v = GGET(cnt, s); my_off += s < l ? v : 0u; total += v; }
          const bool to_cl = rep && rep_dest >= N;
          s_send_cl += to_cl ? 1u : 0u; s_send_sv += cnt - (to_cl ? 1u : 0u);
          const u32 fan_off = my_off + ((IS_ACK && rep) ? 1u : 0u), rep_off = my_off + (IS_ACK ? 0u : fan_cnt);
          u32 ns = GB(dmask != 0);
          while (__ballot(ns != 0)) {  // node -> node: every receiver takes its envelope from each sender, in sender order
            const bool on = ns != 0;
            const u32 s = on ? (u32)__builtin_ctz(ns) : 0u; ns &= ns - 1u;
            const u32 dm = GGET(dmask, s), a = GGET(fan_a, s), off = GGET(fan_off, s);
            u32 b0 = 0;
            if (IS_RPC) b0 = GGET(fan_b0, s);
            if (on && is_node && ((dm >> l) & 1u)) { const u32 rank = __popc(dm & lt); arrive(next_id + off + rank, M_BROADCAST, a, IS_RPC ? b0 + rank : 0u, s); }
          }
          if (IS_RPC) {  // node -> node replies (acknowledgements)
            u32 rs = GB(rep && rep_dest < N);
            while (__ballot(rs != 0)) {
              const bool on = rs != 0;
              const u32 s = on ? (u32)__builtin_ctz(rs) : 0u; rs &= rs - 1u;
              const u32 d = GGET(rep_dest, s), ty = GGET(rep_type, s), a = GGET(rep_a, s), b = GGET(rep_b, s), off = GGET(rep_off, s);
              if (on && l == d) arrive(next_id + off, ty, a, b, s);
            }
          }
          // node -> its own client: no latency; lost like any other message (net.clj:214)
          if (to_cl) {
            const u32 id = next_id + rep_off;
            if (!(NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss)) { c_arr = true; ca_y = (id << 8) | rep_type; ca_a = rep_a; ca_b = rep_b; }
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




template <int PROG>
hipError_t b8_launch(const B8Params &up, uint32_t n, size_t lds, bool nem, bool rnd, hipStream_t st) {
  const dim3 grid((n + 7) / 8), block(64);
  if (nem) { if (rnd) hipLaunchKernelGGL((bcast8_kernel<PROG, true, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((bcast8_kernel<PROG, true, false>), grid, block, lds, st, up); }
  else { if (rnd) hipLaunchKernelGGL((bcast8_kernel<PROG, false, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((bcast8_kernel<PROG, false, false>), grid, block, lds, st, up); }
  return hipGetLastError();
}

}  // namespace

// Whether eight clusters per wavefront simulate this configuration (see the header of this file).
bool msim_bcast8_eligible(const msim_config &c) {
  return c.node_program >= MSIM_NODE_BCAST_FF && c.node_program <= MSIM_NODE_BCAST_RPC_ALL && c.journal_capacity == 0 && c.n_nodes >= 1 && c.n_nodes <= GS &&
         c.concurrency == c.n_nodes && c.max_values / 32u <= WMAX;
}

// Extra per-instance scratch words behind the queues' spill area: what of the LDS queues of the colocated kernel does not fit this
// kernel's RQ slots.
uint64_t msim_bcast8_extra_scratch_words(const msim_config &c) { return (uint64_t)c.n_nodes * c.inbox_capacity * 4; }

hipError_t msim_launch_bcast8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < MSIM_BCAST8_MIN_CLUSTERS && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;   // (see the header)
  B8Params up;
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
  if (kp.dev_flags & 0x1000u) std::fprintf(stderr, "[bcast8] %u clusters, eight per wavefront, %zu B of LDS per wavefront\n", n, lds);   // developer trace bit
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(g8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const bool nem = c.nemesis_mask != 0;
  switch (c.node_program) {
    case MSIM_NODE_BCAST_FF: return b8_launch<MSIM_NODE_BCAST_FF>(up, n, lds, nem, rnd, st);
    case MSIM_NODE_BCAST_FF_ECHOBACK: return b8_launch<MSIM_NODE_BCAST_FF_ECHOBACK>(up, n, lds, nem, rnd, st);
    case MSIM_NODE_BCAST_ACK_RETRY: return b8_launch<MSIM_NODE_BCAST_ACK_RETRY>(up, n, lds, nem, rnd, st);
    default: return b8_launch<MSIM_NODE_BCAST_RPC_ALL>(up, n, lds, nem, rnd, st);
  }
}
