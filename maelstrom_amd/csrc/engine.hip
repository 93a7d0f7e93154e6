// engine.hip — the ensemble simulation kernel and the host runtime behind include/maelsim.h.
//
// HOT PATH: for every test instance, the message queue with simulated latency / loss /
// partitions (net.clj:189-247), node execution (process.clj:136-166 collapsed to "a node consumes one
// message at a time"), the sync RPC clients (client.clj:66-172), the generator/nemesis schedule
// (core.clj:67-80) and the node programs' state-transition functions (rows a13-a15).
//
// MI355X mapping:
//   * one wavefront (one 64-thread workgroup) simulates one cluster; lane e = endpoint e
//     (lanes [0,N) = nodes n0..n{N-1}, lanes [N,N+CS) = client worker slots);
//   * node sets (bitmaps) and the per-endpoint in-flight message queues live in LDS; every endpoint's
//     queue is written only by its own lane ("receiver-side pull": each round the wave walks the
//     senders with v_readlane and every addressed lane appends to its own queue) — no LDS atomics;
//   * the per-round "next event time" is a DPP min-reduction, message ids / history row indices come
//     from a DPP prefix sum + ballots, so ids and row order are canonical (endpoint order) and the
//     result is bit-identical to the sequential CPU oracle;
//   * history rows are staged in LDS and appended to HBM 64 rows (1 KiB) at a time, one 16-B store per
//     lane; read results (bitmaps) are copied LDS->HBM by the whole wave, 256 B per instruction.
//   No MFMA: this is integer/indexing work.  No CUDA/hipify/Triton layers.
//
// This file holds the one-cluster-per-wavefront kernels (and the wide layout) and msim_run's choice of a kernel.  Denser layouts live in
// their own translation units and are taken where a configuration and the batch fit them (DESIGN.md §4.1b has the table): duo.hip (two
// broadcast clusters per wavefront, the headline), raft4.hip (four), txn8.hip / mk8.hip / hat8.hip / uid8.hip / crdt8.hip / bcast8.hip
// (eight: the transactional programs, echo / unique-ids, the CRDTs, the broadcast programs at tutorial sizes).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "engine_internal.h"
#include "wave_common.h"
#include "log2_table.h"


__constant__ u32 d_log2_q24[257];


// -ln(u), u = (r+1)/2^32, Q16, integer only
__device__ __forceinline__ u32 neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = d_log2_q24[idx], l1 = d_log2_q24[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}

// min (deadline, id) over the n envelopes of an HBM spill area.  The scan is latency-bound — with one dependent load per
// step every queued envelope costs an L2/HBM round trip — so 8 independent loads are in flight per step.
__device__ __forceinline__ void spill_min(const uint4 *q, u32 n, u64 &bk, u32 &best, bool &hit) {
  for (u32 i0 = 0; i0 < n; i0 += 8) {
    uint2 k[8];
#pragma unroll
    for (u32 t = 0; t < 8; t++) k[t] = *reinterpret_cast<const uint2 *>(&q[min(i0 + t, n - 1)]);
#pragma unroll
    for (u32 t = 0; t < 8; t++) {
      const u64 kk = ((u64)k[t].x << 32) | k[t].y;
      if (i0 + t < n && kk < bk) { bk = kk; best = i0 + t; hit = true; }
    }
  }
}

// merges a W-word replicate snapshot (HBM scratch) into a node's state (LDS): `or` for sets, element-wise max for counters.
// Both merges are idempotent, so the tail of the last batch re-reads word W-1 instead of branching, and B independent loads
// are in flight per step (one dependent load per word made every replicate delivery cost W HBM round trips).
template <bool IS_MAX, int B = 16>
__device__ __forceinline__ void merge_snapshot(u32 *mine, const u32 *snap, u32 W) {
  for (u32 w0 = 0; w0 < W; w0 += B) {
    u32 v[B];
#pragma unroll
    for (u32 t = 0; t < B; t++) v[t] = snap[min(w0 + t, W - 1)];
#pragma unroll
    for (u32 t = 0; t < B; t++) { const u32 i = min(w0 + t, W - 1); mine[i] = IS_MAX ? max(mine[i], v[t]) : (mine[i] | v[t]); }
  }
}


// Reference (shuffle) versions, used only by the self-test to validate the DPP encodings on hardware.
__device__ u32 wave_min_ref(u32 v) { for (int o = 32; o; o >>= 1) v = min(v, (u32)__shfl_xor((int)v, o)); return v; }
__device__ u32 wave_incl_scan_ref(u32 v) {
  const u32 lane = threadIdx.x & 63;
  for (int o = 1; o < 64; o <<= 1) { u32 t = (u32)__shfl_up((int)v, o); if (lane >= (u32)o) v += t; }
  return v;
}
__global__ void wave_selftest_kernel(const u32 *in, u32 *out) {
  const u32 lane = threadIdx.x;
  const u32 v = in[blockIdx.x * 64 + lane];
  u32 bad = 0;
  bad |= wave_min(v) != wave_min_ref(v);
  bad |= wave_incl_scan(v & 0xFFFF) != wave_incl_scan_ref(v & 0xFFFF);
  bad |= wave_sum(v & 0xFF) != rdlane(wave_incl_scan_ref(v & 0xFF), 63);
  bad |= (lane < 32) && scan32(v & 0xFFFF) != wave_incl_scan_ref(v & 0xFFFF);
  bad |= lane_get(v, (lane * 7 + 3) & 63) != (u32)__shfl((int)v, (int)((lane * 7 + 3) & 63));
  out[blockIdx.x * 64 + lane] = bad;
}


// =====================================================================================================
// The simulation kernel: one wavefront = one cluster.  PROG = node program (MSIM_NODE_*), NEM = the
// partition nemesis is compiled in, NET_RANDOM = latency is drawn per message and/or messages can be lost
// (false: constant latency, no loss — no per-message RNG in the hot path).  Follows DESIGN.md §2 step by step; the CPU oracle implements the
// same text independently.
//
// One round:
//   R0 pick the time: stay at T while anything is due, else jump to the next event (DPP min)
//   R1 scheduler (generator interpreter, nemesis) — wave-uniform, scalar
//   R2 marked clients invoke -> COMMIT -> idle receivers poll
//   R3 one input per node (timer or due message) -> COMMIT -> idle receivers poll
//   R4 clients run their recv! loop -> completion rows
// so an RPC to an idle node completes inside one round, and a gossip hop costs one round.
// =====================================================================================================
template <int PROG, bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64, 4) sim_kernel(const KParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *const stage = reinterpret_cast<uint4 *>(smem);
  uint4 *const inbox = reinterpret_cast<uint4 *>(smem + p.off_inbox);
  u32 *const seen = reinterpret_cast<u32 *>(smem + p.off_seen);
  u32 *const misc = reinterpret_cast<u32 *>(smem + p.off_misc);

  constexpr bool IS_BCAST = PROG >= MSIM_NODE_BCAST_FF && PROG <= MSIM_NODE_BCAST_RPC_ALL;
  constexpr bool IS_RPC = PROG == MSIM_NODE_BCAST_ACK_RETRY || PROG == MSIM_NODE_BCAST_RPC_ALL;
  constexpr bool IS_ACK = PROG == MSIM_NODE_BCAST_ACK_RETRY;
  constexpr bool IS_PN = PROG == MSIM_NODE_PN_COUNTER;    // pn_counter.rb: same replication skeleton as g-set, counters instead of a set
  constexpr bool IS_GSET = PROG == MSIM_NODE_G_SET || IS_PN;  // "CRDT with a 5 s replicate timer"
  constexpr bool IS_ECHO = PROG == MSIM_NODE_ECHO;
  constexpr bool IS_FLAKE = PROG == MSIM_NODE_FLAKE_IDS;  // flake_ids.clj: unique-ids workload, Reusable clients (unique_ids.clj:59-61)
  constexpr bool HAS_FINAL = IS_BCAST || IS_GSET;
  constexpr bool FINAL_FLAG = IS_BCAST || IS_PN;  // :final? true on the last reads (broadcast.clj:240, pn_counter.clj:137)
  constexpr bool HAS_TIMERS = IS_ACK || IS_GSET;
  constexpr bool REP_FIRST = IS_ACK;  // the ack variant replies before it gossips
  constexpr u32 FAN_TYPE = IS_GSET ? M_REPLICATE : M_BROADCAST;
  // fan-outs of these programs only ever go to topology neighbours
  constexpr bool TOPO_BOUND = PROG == MSIM_NODE_BCAST_FF || PROG == MSIM_NODE_BCAST_FF_ECHOBACK || IS_ACK;
  // programs whose server<->server traffic is plain gossip (no msg_id, no reply): eligible for the cascade loop
  constexpr bool FAST_OK = PROG == MSIM_NODE_BCAST_FF || PROG == MSIM_NODE_BCAST_FF_ECHOBACK;

  const u32 lane = threadIdx.x;
  const u32 inst = blockIdx.x;
  const u32 N = p.N, C = p.C, CS = p.CS, W = p.W;
  const bool is_node = lane < N;
  const bool is_client = lane >= N && lane < N + CS;
  const u32 slot = lane - N;
  const bool is_worker = is_client && slot < C;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u64 lt_mask = (1ull << lane) - 1;
  const u32 lt32 = lane < 32 ? ((1u << lane) - 1) : 0xFFFFFFFFu;
  const u64 worker_mask = ((C >= 64 ? ~0ull : ((1ull << C) - 1)) << N);
  const u32 all_nodes = N >= 32 ? 0xFFFFFFFFu : ((1u << N) - 1);
  const u32 max_values = p.cfg.max_values, max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  // ack/retry scratch: unacked[N][V], fifo[N][V][2]; g-set scratch: snapshots[tick][N][W]
  u32 *const g_unacked = g_scr + (size_t)lane * max_values;
  u32 *const g_fifo = g_scr + (size_t)N * max_values + (size_t)lane * max_values * 2;

  const u32 jcap = p.cfg.journal_capacity;  // net journal (journal.clj:53,220-239); 0 = off
  uint4 *const g_ev = p.journal + (size_t)inst * jcap;
  const u32 my_cap = is_node ? p.cap_node : CLIENT_INBOX_CAP;
  const u32 my_spill_cap = is_node ? p.spill_cap : 0u;
  uint4 *const my_inbox = inbox + (is_node ? lane * p.cap_node : (is_client ? N * p.cap_node + slot * CLIENT_INBOX_CAP : 0));
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)(is_node ? lane : 0) * p.spill_cap;  // HBM spill behind the LDS queue
  u32 *const my_seen = seen + (is_node ? lane : 0) * W;

  for (u32 i = lane; i < N * W; i += 64) seen[i] = 0;
  if (IS_ACK) { if (is_node) for (u32 v = 0; v < max_values; v++) g_unacked[v] = 0; }
  __syncthreads();

  const u32 adj = is_node ? topo_adj(p.cfg.topology, N, lane) : 0;
  // which lanes can ever address a fan-out to this node
  const u32 cand_all = is_node ? (TOPO_BOUND ? adj : (all_nodes & ~(1u << lane))) : 0u;
  const u32 c_mod_n = C % N;

  // ---- per-lane endpoint state ----
  bool has_c = false; u32 deliver_at = 0; uint4 cm = make_uint4(0, 0, 0, 0);  // the envelope recv! is sleeping on
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);                    // smallest arrival of this commit
  u32 in_n = 0, sp_n = 0;                                                     // queued envelopes in LDS / in the HBM spill
  u32 node_msgid = 0, timer_next = INF, tick = 0, part = 0;
  u32 flake_time = 0, flake_count = 0;  // flake_ids.clj:10-14
  u32 fifo_head = 0, fifo_tail = 0, retry_time = INF;
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_f = 0, c_value = 0, process = slot, c_final = 0;
  u32 dest_node = is_client ? slot % N : 0;  // nodes[process mod n] [upstream], kept incrementally
  u32 m_f = 0, m_value = 0, m_final = 0;
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- wave-uniform state ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, next_value = 0, nem_next = 0, nem_j = 0;
  u32 sleep_until = 0, loss_on = 0, next_id = 0, n_rows = 0, n_payload = 0, flags = 0, rounds = 0;
  u32 n_ev = 0, ev_base = 0, id_base = 0;  // journal cursor; :send events of a COMMIT sit at ev_base + (id - id_base)

  // one journal event (event :id = idx); y = (message id << 8) | body type
  auto jwrite = [&](u32 idx, u32 recv, u32 y, u32 a, u32 b, u32 src, u32 dest) {
    if (idx < jcap) g_ev[idx] = make_uint4(T, (y & ~0x80u) | (recv << 7), a, src | (dest << 8) | ((b & 0xFFFFu) << 16));
    else my_flags |= MSIM_FLAG_JOURNAL_OVERFLOW;
  };
  // queue an envelope in this lane's LDS inbox
  auto lds_push = [&](const uint4 m) {
    if (in_n < my_cap) { my_inbox[in_n++] = m; return; }
    if (sp_n < my_spill_cap) { my_spill[sp_n++] = m; return; }
    my_flags |= MSIM_FLAG_INBOX_OVERFLOW;
  };
  // a message addressed to this lane arrives (net.clj:189-221: latency, loss, enqueue)
  auto arrive = [&](u32 id, u32 type, u32 a, u32 b, u32 src) {
    u32 lat = 0;
    if (src < N && is_node) {  // latency only between servers (net.clj:178-187)
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
    }
    if (jcap) jwrite(ev_base + (id - id_base), 0, (id << 8) | type, a, b, src, lane);  // journal :send precedes the loss decision (net.clj:208)
    if (NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss) return;  // net.clj:214
    uint4 m = make_uint4(T + lat * 1000u, (id << 8) | type, a, b | (src << 24));
    if (!have_pm) { pm = m; have_pm = true; return; }
    if (m.x < pm.x || (m.x == pm.x && m.y < pm.y)) { const uint4 t = m; m = pm; pm = t; }
    lds_push(m);
  };
  // commit an envelope to this receiver: partition check at poll time (net.clj:234), sleep floor(dt) ms (:236-238)
  auto try_commit = [&](const uint4 e) {
    const u32 src = e.w >> 24;
    if (NEM && is_node && src < N && ((part >> src) & 1)) return;  // dropped, no :recv
    cm = e; has_c = true;
    deliver_at = e.x <= T ? T : T + ((e.x - T) / 1000u) * 1000u;
  };
  // idle receivers poll (net.clj:223-247): min (deadline, id) over queued + just-arrived envelopes
  auto poll = [&]() {
    const bool elig = is_node || busy;  // clients only poll inside recv! (client.clj:94-95)
    if (have_pm) {
      have_pm = false;
      if (elig && !has_c && (in_n | sp_n) == 0) try_commit(pm);  // common case: nothing queued, no LDS traffic
      else lds_push(pm);
    }
    while (elig && !has_c && (in_n | sp_n) != 0) {
      u32 best = 0; bool in_spill = false;
      u64 bk = ~0ull;
      for (u32 i = 0; i < in_n; i++) {
        const uint2 kk = *reinterpret_cast<const uint2 *>(&my_inbox[i]);
        const u64 k2 = ((u64)kk.x << 32) | kk.y;
        if (k2 < bk) { bk = k2; best = i; }
      }
      spill_min(my_spill, sp_n, bk, best, in_spill);  // deep queues only (long head-of-line sleeps)
      uint4 e;
      if (in_spill) { e = my_spill[best]; sp_n--; if (best != sp_n) my_spill[best] = my_spill[sp_n]; }
      else { e = my_inbox[best]; in_n--; if (best != in_n) my_inbox[best] = my_inbox[in_n]; }
      try_commit(e);
    }
  };
  // COMMIT of the nodes' fan-outs, receiver side: every node pulls from the lanes that may address it.
  // ids = next_id + id_off(sender) + rank of the receiver inside the sender's fan mask (net.clj:197).
  auto commit_fan = [&](u32 fan_mask, u32 fan_a, u32 fan_b0, u32 id_off) {
    const u32 send = (u32)__ballot(fan_mask != 0);
    u32 cand = cand_all & send;
    while (__ballot(cand != 0)) {
      const bool has = cand != 0;
      const u32 s = has ? (u32)__builtin_ctz(cand) : 0u;
      cand &= cand - 1;
      const u32 fs = lane_get(fan_mask, s);
      const u32 as = lane_get(fan_a, s);
      const u32 os = lane_get(id_off, s);
      u32 bs = 0;
      if (IS_RPC) bs = lane_get(fan_b0, s);
      if (has && ((fs >> lane) & 1)) {
        const u32 rank = __popc(fs & lt32);
        arrive(next_id + os + rank, FAN_TYPE, as, IS_RPC ? bs + rank : 0u, s);
      }
    }
  };

  for (;;) {
    const u64 busy_mask = __ballot(busy);  // only client lanes are ever busy

    // ---- time-free phase transitions (oracle: sched_resolve) ----
    if (!(phase == PH_MAIN && ((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff)))) {
      for (bool again = true; again;) {
        again = false;
        switch (phase) {
          case PH_INIT_WAIT: if (!busy_mask) { phase = IS_BCAST ? PH_TOPO : PH_MAIN_START; again = true; } break;
          case PH_TOPO_WAIT: if (!busy_mask) { phase = PH_MAIN_START; again = true; } break;
          case PH_MAIN_START:
            cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; nem_next = T;
            next_msg_id = 0; loss_on = 1; phase = PH_MAIN; again = true; break;
          case PH_MAIN: {
            const bool gl = rate > 0 && gen_next < cutoff, nl = NEM && nem_next < cutoff;
            if (gl || nl) break;
            if (rate == 0 && T < cutoff) break;
            phase = PH_DRAIN; again = true;
          } break;
          case PH_DRAIN:
            if (busy_mask & worker_mask) break;
            phase = (NEM && HAS_FINAL) ? PH_NEM_FINAL : HAS_FINAL ? PH_SLEEP : PH_DONE;
            if (phase == PH_SLEEP) sleep_until = T + p.cfg.quiesce_ms * 1000u;
            again = true; break;
          case PH_FINAL_WAIT: if (!(busy_mask & worker_mask)) { phase = PH_DONE; again = true; } break;
          default: break;
        }
      }
      if (phase == PH_DONE) break;
    }
    if (++rounds > ROUND_LIMIT) { flags |= MSIM_FLAG_ROUND_LIMIT; break; }

    // ---- R0: time ----
    const bool gen_live = rate > 0 && gen_next < cutoff;
    const bool nem_live = NEM && nem_next < cutoff;
    const u64 free_mask = worker_mask & ~busy_mask;
    u32 due = INF;
    switch (phase) {
      case PH_INIT: case PH_TOPO: case PH_NEM_FINAL: case PH_FINAL: due = T; break;
      case PH_SLEEP: due = sleep_until; break;
      case PH_MAIN:
        if (nem_live) due = max(nem_next, T);
        if (gen_live && free_mask) due = min(due, max(gen_next, T));
        if (rate == 0 && !nem_live) due = min(due, cutoff);
        break;
      default: break;
    }
    u32 my_t = has_c ? deliver_at : INF;  // this lane's next "normal" event
    if (HAS_TIMERS && is_node) my_t = min(my_t, min(timer_next, retry_time));
    bool timeout_round = false;
    if (due > T && !__ballot(my_t <= T)) {  // nothing due now: jump to the next event
      u32 k = my_t == INF ? INF : my_t * 2;
      if (busy) k = min(k, timeout_at * 2 + 1);
      u32 km = wave_min(k);
      if (due != INF) km = min(km, due * 2);
      if (km == INF) { flags |= MSIM_FLAG_ROUND_LIMIT; break; }  // stuck
      timeout_round = (km & 1) != 0;
      T = max(T, km >> 1);
    }

    bool inv_row = false; u32 inv_packed = 0, inv_value = 0;               // invoke row of this round
    bool cmp_row = false; u32 cmp_packed = 0, cmp_value = 0, cmp_len = 0;  // completion row of this round
    u32 nem_rows = 0, nem_f = 0, nem_v1 = 0, nem_v2 = 0, nem_len2 = 0;

    // completion of a client op (oracle: client_complete)
    auto complete = [&](u32 type, u32 err, u32 value, u32 len) {
      busy = false;
      if (kind != K_OP) { if (type != MSIM_T_OK) my_flags |= MSIM_FLAG_ROUND_LIMIT; return; }
      cmp_row = true; cmp_packed = type | (c_f << 2) | (err << 7) | (c_final << 11) | (process << 12);
      cmp_value = value; cmp_len = len;
      if (type == MSIM_T_INFO) {  // crashed process: new process id, fresh client [upstream interpreter]
        process += C; dest_node += c_mod_n; if (dest_node >= N) dest_node -= N;
        if (!IS_FLAKE) { next_msg_id = 0; in_n = 0; }  // Reusable clients (unique_ids.clj:59-61) are not re-opened
      }
    };

    if (timeout_round) {
      if (busy && timeout_at <= T) {  // client.clj:96-103 + :158-162
        const bool idem = IS_BCAST && c_f == MSIM_F_READ;
        complete(idem ? MSIM_T_FAIL : MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_f == MSIM_F_READ ? MSIM_NO_VALUE : c_value, 0);
      }
    } else {
      // ---- R1: scheduler (generator interpreter, nemesis) — wave-uniform ----
      if (due <= T) {
        switch (phase) {
          case PH_INIT: if (is_client && slot < N) { mark = true; kind = K_INIT; } phase = PH_INIT_WAIT; break;
          case PH_TOPO: if (is_client && slot < N) { mark = true; kind = K_TOPO; } phase = PH_TOPO_WAIT; break;
          case PH_MAIN: {
            if (NEM && nem_live && nem_next <= T) {
              const u32 j = nem_j++;
              nem_rows = 2;
              if ((j & 1) == 0) {  // :start-partition (jepsen.nemesis.combined partition-package, restated)
                const u32 spec = scale32(draw32(key, S_NEM_SPEC, j), 4);
                // shuffle (Fisher-Yates) in LDS by lane 0; every node lane then derives its own grudge row
                if (lane < N) misc[lane] = lane;
                __syncthreads();
                if (lane == 0 && spec != MSIM_SPEC_ONE) {
                  for (u32 i = N - 1; i >= 1; i--) {
                    const u32 kk = scale32(draw32(key, S_NEM_SHUFFLE, ((u64)j << 16) | i), i + 1);
                    const u32 t = misc[i]; misc[i] = misc[kk]; misc[kk] = t;
                  }
                }
                __syncthreads();
                u32 my_part = 0;
                if (is_node) {
                  if (spec == MSIM_SPEC_ONE) {
                    const u32 loner = scale32(draw32(key, S_NEM_PICK, j), N);
                    my_part = lane == loner ? (all_nodes & ~(1u << loner)) : (1u << loner);
                  } else if (spec == MSIM_SPEC_MAJORITY || spec == MSIM_SPEC_MINORITY_THIRD) {
                    const u32 cnt = spec == MSIM_SPEC_MAJORITY ? N / 2 : (N - 1) / 3;
                    u32 comp = 0;
                    for (u32 i = 0; i < cnt; i++) comp |= 1u << misc[i];
                    my_part = ((comp >> lane) & 1) ? (all_nodes & ~comp) : comp;
                  } else {  // majorities-ring
                    const u32 m = N / 2 + 1;
                    u32 pos = 0;
                    for (u32 i = 0; i < N; i++) if (misc[i] == lane) pos = i;
                    const u32 i0 = (pos + N - (m / 2) % N) % N;
                    u32 vis = 0;
                    for (u32 kk = 0; kk < m; kk++) vis |= 1u << misc[(i0 + kk) % N];
                    my_part = all_nodes & ~vis;
                  }
                }
                part |= my_part;
                const u32 words = N * MSIM_MASK_WORDS;
                u32 off = 0;
                if (n_payload + words > max_pay) flags |= MSIM_FLAG_PAYLOAD_OVERFLOW;
                else {
                  off = n_payload; n_payload += words;
                  if (is_node) { g_pay[off + lane * 4] = part; g_pay[off + lane * 4 + 1] = 0; g_pay[off + lane * 4 + 2] = 0; g_pay[off + lane * 4 + 3] = 0; }
                }
                nem_f = MSIM_F_START_PARTITION; nem_v1 = spec; nem_v2 = off; nem_len2 = words;
              } else {  // :stop-partition -> heal! (net.clj:112-113)
                part = 0;
                nem_f = MSIM_F_STOP_PARTITION; nem_v1 = MSIM_NO_VALUE; nem_v2 = MSIM_NO_VALUE; nem_len2 = 0;
              }
              nem_next = T + __umulhi(draw32(key, S_NEM_STAGGER, j), p.nem_period2_us);
            }
            if (gen_live && gen_next <= T && free_mask) {
              // one 64-bit draw per generated op: high word -> stagger, low word -> pick / mix / echo payload
              const u32 nfree = __popcll(free_mask);
              const u32 kk = gen_k++;
              const u64 h = draw64(key, S_GEN, kk);
              const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
              const u32 pick = scale32(r_lo, nfree);
              const bool sel = is_worker && !busy && (u32)__popcll(free_mask & lt_mask) == pick;
              u32 f, val = MSIM_NO_VALUE;
              bool ok = true;
              if (IS_ECHO) { f = MSIM_F_ECHO; val = (r_lo >> 4) & 127; }
              else if (IS_FLAKE) f = MSIM_F_GENERATE;  // (gen/repeat {:f :generate}), unique_ids.clj:72
              else if (IS_PN && p.cfg.workload == MSIM_WL_G_COUNTER) {
                // g_counter.clj:37-41: (gen/filter ...) skips negative adds and takes the mix's next op at once
                u32 rr = r_lo, a = 0;
                int d = (int)((((rr >> 4) & 0xFFFFu) * 10u) >> 16) - 5;
                while (!(rr & 1) && d < 0 && a < 15) { a++; rr = (u32)draw64(key, S_GEN2, (u64)kk * 16 + a); d = (int)((((rr >> 4) & 0xFFFFu) * 10u) >> 16) - 5; }
                if ((rr & 1) || d < 0) f = MSIM_F_READ; else { f = MSIM_F_ADD; val = (u32)d; }
              }
              else if (r_lo & 1) f = MSIM_F_READ;
              else {
                f = IS_BCAST ? MSIM_F_BROADCAST : MSIM_F_ADD;
                if (IS_PN) val = (u32)((int)((((r_lo >> 4) & 0xFFFFu) * 10u) >> 16) - 5);  // (- (rand-int 10) 5), pn_counter.clj:134-135
                else if (next_value >= max_values) { flags |= MSIM_FLAG_VALUES_OVERFLOW; ok = false; }
                else val = next_value++;
              }
              if (!ok) { phase = PH_DONE; break; }
              if (sel) { mark = true; kind = K_OP; m_f = f; m_value = val; m_final = 0; }
              gen_next = T + __umulhi(r_hi, p.gen_period2_us);
            }
          } break;
          case PH_NEM_FINAL:
            part = 0; nem_rows = 2; nem_f = MSIM_F_STOP_PARTITION; nem_v1 = MSIM_NO_VALUE; nem_v2 = MSIM_NO_VALUE;
            phase = PH_SLEEP; sleep_until = T + p.cfg.quiesce_ms * 1000u; break;
          case PH_SLEEP:
            if (T < sleep_until) break;
            phase = PH_FINAL;
            [[fallthrough]];
          case PH_FINAL:
            if (is_worker) { mark = true; kind = K_OP; m_f = MSIM_F_READ; m_value = MSIM_NO_VALUE; m_final = FINAL_FLAG ? 1 : 0; }
            phase = PH_FINAL_WAIT; break;
          default: break;
        }
        if (phase == PH_DONE) break;
      }

      // ---- R2: marked clients invoke; COMMIT (ids in slot order); idle receivers poll ----
      u64 inv_mask = __ballot(mark);
      if (inv_mask) {
        u32 rq_dest = 0, rq_type = 0, rq_a = 0;
        if (mark) {  // oracle: client_invoke
          mark = false; busy = true;
          if (kind == K_INIT) { rq_dest = slot; rq_type = M_INIT; next_msg_id = 0; }
          else if (kind == K_TOPO) { rq_dest = slot; rq_type = M_TOPOLOGY; next_msg_id = 0; }
          else {
            c_f = m_f; c_value = m_value; c_final = m_final;
            rq_dest = dest_node;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (c_f << 2) | (c_final << 11) | (process << 12); inv_value = c_value;
            rq_type = c_f == MSIM_F_ECHO ? M_ECHO : c_f == MSIM_F_BROADCAST ? M_BROADCAST : c_f == MSIM_F_ADD ? M_ADD : c_f == MSIM_F_GENERATE ? M_GENERATE : M_READ;
            rq_a = (c_f == MSIM_F_READ || c_f == MSIM_F_GENERATE) ? 0u : c_value;
          }
          want = ++next_msg_id;
          timeout_at = T + (kind == K_OP ? p.cfg.client_timeout_ms : 10000u) * 1000u;
          s_send_cl++;
        }
        const u32 rq_pack = rq_dest | (rq_type << 8);
        ev_base = n_ev; id_base = next_id; n_ev += (u32)__popcll(inv_mask);
        while (inv_mask) {
          const u32 s = (u32)__builtin_ctzll(inv_mask); inv_mask &= inv_mask - 1;
          const u32 pk = rdlane(rq_pack, s);
          const u32 a = rdlane(rq_a, s), b = rdlane(want, s);
          if (lane == (pk & 0xFF)) arrive(next_id, pk >> 8, a, b, s);
          next_id++;
        }
        poll();
      }

      // ---- R3: one input per node: a due timer, else the due committed envelope ----
      u32 fan_mask = 0, fan_a = 0, fan_b0 = 0;
      bool rep = false; u32 rep_dest = 0, rep_type = 0, rep_a = 0, rep_b = 0;
      bool rd = false;
      u64 jd_mask = 0;  // nodes delivering an envelope this round (their :recv events come first, node order)
      if (jcap) jd_mask = __ballot(is_node && !(IS_GSET && timer_next <= T) && !(IS_ACK && retry_time <= T) && has_c && deliver_at <= T);
      if (is_node) {
        if (IS_GSET && timer_next <= T) {  // g_set.rb:33-38
          timer_next = T + 5000000u;
          u32 *snap = g_scr + ((size_t)tick * N + lane) * W;
          for (u32 w = 0; w < W; w++) snap[w] = my_seen[w];
          fan_mask = all_nodes & ~(1u << lane); fan_a = tick; tick++;
        } else if (IS_ACK && retry_time <= T) {  // gossip thread wakes (02-performance.md:421-438)
          const u32 slot_i = (fifo_head % max_values) * 2;
          const u32 v = g_fifo[slot_i];
          fifo_head++;
          const u32 un = g_unacked[v];
          if (un) {
            fan_mask = un; fan_a = v; fan_b0 = node_msgid + 1; node_msgid += __popc(un);
            const u32 ts = (fifo_tail % max_values) * 2;
            g_fifo[ts] = v; g_fifo[ts + 1] = T + 1000000u; fifo_tail++;
          }
          retry_time = fifo_head < fifo_tail ? g_fifo[(fifo_head % max_values) * 2 + 1] : INF;
        } else if (has_c && deliver_at <= T) {
          const uint4 q = cm; has_c = false;
          const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
          if (qsrc >= N) s_recv_cl++; else s_recv_sv++;  // journal :recv (net.clj:244)
          if (jcap) jwrite(n_ev + (u32)__popcll(jd_mask & lt_mask), 1, q.y, qa, qb, qsrc, lane);
          switch (qtype) {
            case M_INIT:
              if (IS_GSET) timer_next = T;
              rep = true; rep_dest = qsrc; rep_type = M_INIT_OK; rep_b = qb; break;
            case M_TOPOLOGY: rep = true; rep_dest = qsrc; rep_type = M_TOPOLOGY_OK; rep_b = qb; break;
            case M_ECHO: rep = true; rep_dest = qsrc; rep_type = M_ECHO_OK; rep_a = qa; rep_b = qb; break;
            case M_GENERATE: {  // flake_ids.clj:16-31: [max(now in s, last time), counter within that second, node]
              u32 t = T / 1000000u;
              if (t < flake_time) t = flake_time;
              flake_count = t == flake_time ? flake_count + 1 : 0u; flake_time = t;
              rep = true; rep_dest = qsrc; rep_type = M_GENERATE_OK; rep_a = (t << 20) | ((flake_count & 0x7FFFu) << 5) | lane; rep_b = qb;
            } break;
            case M_READ:
              rep = true; rep_dest = qsrc; rep_type = M_READ_OK; rep_b = qb;
              if (IS_PN) { u32 v = 0; for (u32 i = 0; i < N; i++) v += my_seen[i] - my_seen[N + i]; rep_a = v; }  // increments - decrements
              else rd = true;
              break;
            case M_ADD:
              if (IS_PN) { const int d = (int)qa; if (d >= 0) my_seen[lane] += (u32)d; else my_seen[N + lane] += (u32)(-d); }  // own slot of inc / dec
              else my_seen[qa >> 5] |= 1u << (qa & 31);
              rep = true; rep_dest = qsrc; rep_type = M_ADD_OK; rep_a = qa; rep_b = qb; break;
            case M_REPLICATE: {
              const u32 *snap = g_scr + ((size_t)qa * N + qsrc) * W;
              if (IS_PN) merge_snapshot<true>(my_seen, snap, W);  // element-wise max
              else merge_snapshot<false>(my_seen, snap, W);
            } break;
            case M_BROADCAST: {
              const u32 v = qa, bitm = 1u << (v & 31);
              const u32 wv = my_seen[v >> 5];
              if (!(wv & bitm)) {
                my_seen[v >> 5] = wv | bitm;
                u32 tg = PROG == MSIM_NODE_BCAST_RPC_ALL ? (all_nodes & ~(1u << lane)) : adj;
                if (PROG != MSIM_NODE_BCAST_FF_ECHOBACK && qsrc < N) tg &= ~(1u << qsrc);
                fan_mask = tg; fan_a = v;
                if (IS_RPC) { fan_b0 = node_msgid + 1; node_msgid += __popc(tg); }
                if (IS_ACK && tg) {
                  g_unacked[v] = tg;
                  const u32 ts = (fifo_tail % max_values) * 2;
                  g_fifo[ts] = v; g_fifo[ts + 1] = T + 1000000u;
                  if (fifo_head == fifo_tail) retry_time = T + 1000000u;
                  fifo_tail++;
                }
              }
              if (qb != 0) { rep = true; rep_dest = qsrc; rep_type = M_BROADCAST_OK; rep_a = v; rep_b = qb; }
            } break;
            case M_BROADCAST_OK: if (IS_ACK) g_unacked[qa] &= ~(1u << qsrc); break;
            default: break;
          }
        }
      }

      n_ev += (u32)__popcll(jd_mask);
      // read results: the whole wave copies the node's set LDS -> HBM payload (256 B per instruction)
      {
        u64 rdmask = __ballot(rd);
        if (rdmask) {
          __syncthreads();
          const u32 words = (next_value + 31) >> 5;
          while (rdmask) {
            const u32 r = (u32)__builtin_ctzll(rdmask); rdmask &= rdmask - 1;
            u32 off = 0;
            if (n_payload + words > max_pay) flags |= MSIM_FLAG_PAYLOAD_OVERFLOW;
            else {
              off = n_payload; n_payload += words;
              for (u32 w = lane; w < words; w += 64) g_pay[off + w] = seen[r * W + w];
            }
            if (lane == r) rep_a = off | (words << 24);
          }
        }
      }

      // COMMIT node sends (net.clj:189-221): ids in node order, then emission order
      {
        const u32 fan_cnt = __popc(fan_mask);
        const u32 cnt = fan_cnt + (rep ? 1u : 0u);
        if (__ballot(cnt != 0)) {
          const u32 incl = scan32(cnt);  // senders are node lanes (< 32)
          ev_base = n_ev; id_base = next_id; n_ev += rdlane(incl, 31);
          if (is_node) { s_send_sv += fan_cnt; if (rep) { if (rep_dest >= N) s_send_cl++; else s_send_sv++; } }
          if (__ballot(fan_mask != 0)) commit_fan(fan_mask, fan_a, fan_b0, incl - cnt + ((REP_FIRST && rep) ? 1u : 0u));
          u64 reps = __ballot(rep);
          if (reps) {
            const u32 rep_pack = rep_dest | (rep_type << 8);
            const u32 rep_off = incl - cnt + (REP_FIRST ? 0u : fan_cnt);
            while (reps) {
              const u32 s = (u32)__builtin_ctzll(reps); reps &= reps - 1;
              const u32 pk = rdlane(rep_pack, s), o = rdlane(rep_off, s);
              const u32 r_a = rdlane(rep_a, s), r_b = rdlane(rep_b, s);
              if (lane == (pk & 0xFF)) arrive(next_id + o, pk >> 8, r_a, r_b, s);
            }
          }
          next_id += rdlane(incl, 31);
        }
        poll();  // every round: a node that just went idle may still have queued envelopes
      }

      // ---- R4: clients run their recv! loops (client.clj:94-107); envelope k of every client before envelope k+1 ----
      for (;;) {
        const bool dl = is_client && has_c && deliver_at <= T;
        const u64 dm = __ballot(dl);
        if (!dm) break;
        if (dl) {
          const uint4 q = cm; has_c = false;
          s_recv_cl++;
          const u32 qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
          if (jcap) jwrite(n_ev + (u32)__popcll(dm & lt_mask), 1, q.y, qa, qb, q.w >> 24, lane);
          if (busy && qb == want) {  // else: stale reply, keep polling (client.clj:105-107)
            if (qtype == M_READ_OK) { if (IS_PN) complete(MSIM_T_OK, 0, qa, 0); else complete(MSIM_T_OK, 0, qa & 0xFFFFFFu, qa >> 24); }
            else if (qtype == M_ECHO_OK || qtype == M_GENERATE_OK) complete(MSIM_T_OK, 0, qa, 0);
            else complete(MSIM_T_OK, 0, c_value, 0);
          }
          poll();
        }
        n_ev += (u32)__popcll(dm);
      }
    }

    // ---- history rows: canonical order = nemesis rows, invokes (slot order), completions (slot order) ----
    {
      const u64 imask = __ballot(inv_row), cmask = __ballot(cmp_row);
      const u32 ni = (u32)__popcll(imask);
      const u32 nr = nem_rows + ni + (u32)__popcll(cmask);
      if (nr) {
        if (n_rows + nr > max_rows) { flags |= MSIM_FLAG_ROWS_OVERFLOW; break; }
        const u32 tlo = (u32)((u64)T * 1000ull), thi = (u32)(((u64)T * 1000ull) >> 32);
        if (NEM && nem_rows && lane == 0) {
          const u32 pk = MSIM_T_INFO | (nem_f << 2) | (MSIM_PROCESS_NEMESIS << 12);
          stage[n_rows % STAGE_ROWS] = make_uint4(tlo, thi, pk, nem_v1);
          stage[(n_rows + 1) % STAGE_ROWS] = make_uint4(tlo, thi | (nem_len2 << 16), pk, nem_v2);
        }
        if (inv_row) stage[(n_rows + nem_rows + (u32)__popcll(imask & lt_mask)) % STAGE_ROWS] = make_uint4(tlo, thi, inv_packed, inv_value);
        if (cmp_row) stage[(n_rows + nem_rows + ni + (u32)__popcll(cmask & lt_mask)) % STAGE_ROWS] = make_uint4(tlo, thi | (cmp_len << 16), cmp_packed, cmp_value);
        const u32 new_n = n_rows + nr;
        if ((new_n >> 6) != (n_rows >> 6)) {  // a 64-row block completed: coalesced 1 KiB append to HBM
          __syncthreads();
          for (u32 blk = n_rows >> 6; blk < (new_n >> 6); blk++) {
            const u32 gi = blk * 64 + lane;
            if (gi < max_rows) reinterpret_cast<uint4 *>(g_rows)[gi] = stage[gi % STAGE_ROWS];
          }
          __syncthreads();
        }
        n_rows = new_n;
      }
    }

    // ---- cascade loop: while the scheduler is quiet at T and only plain gossip is due, every round is
    //      R3 + COMMIT + poll (no scheduler, no clients, no rows).  Same rounds the loop above would run. ----
    if (FAST_OK && !timeout_round) {
      const u64 bm = __ballot(busy);
      bool quiet = false;
      u32 d2 = INF;  // when the scheduler next wants to act (the clients' busy set cannot change inside this loop)
      if (phase == PH_MAIN) {
        const bool gl = rate > 0 && gen_next < cutoff, nl = NEM && nem_next < cutoff;
        if (gl || nl) {
          if (nl) d2 = max(nem_next, T);
          if (gl && (worker_mask & ~bm)) d2 = min(d2, max(gen_next, T));
          quiet = d2 > T;
        }
      } else if (phase == PH_SLEEP) { d2 = sleep_until; quiet = d2 > T; }
      while (quiet) {
        bool due_now = has_c && deliver_at <= T;  // only node lanes hold an envelope across rounds
        if (!__ballot(due_now)) {
          // R0 inside the loop: jump to the next delivery if it precedes the scheduler and every client timeout
          u32 k = has_c ? deliver_at * 2 : INF;
          if (busy) k = min(k, timeout_at * 2 + 1);
          const u32 km = wave_min(k);
          if (km == INF || (km & 1) || (km >> 1) >= d2) break;
          T = km >> 1;
          due_now = has_c && deliver_at <= T;
        }
        const bool plain = (cm.y & 0xFFu) == M_BROADCAST && (cm.w & 0xFFFFFFu) == 0;
        if (__ballot(due_now && !plain)) break;
        if (++rounds > ROUND_LIMIT) { flags |= MSIM_FLAG_ROUND_LIMIT; phase = PH_DONE; break; }
        const u32 v = cm.z;
        u32 fan = 0;
        if (jcap) {
          const u64 dmj = __ballot(due_now);
          if (due_now) jwrite(n_ev + (u32)__popcll(dmj & lt_mask), 1, cm.y, v, 0, cm.w >> 24, lane);
          n_ev += (u32)__popcll(dmj);
        }
        if (due_now) {
          has_c = false; s_recv_sv++;
          const u32 wv = my_seen[v >> 5], bitm = 1u << (v & 31);
          if (!(wv & bitm)) {
            my_seen[v >> 5] = wv | bitm;
            fan = PROG == MSIM_NODE_BCAST_FF_ECHOBACK ? adj : (adj & ~(1u << (cm.w >> 24)));
          }
        }
        if (__ballot(fan != 0)) {
          const u32 cnt = __popc(fan);
          const u32 incl = scan32(cnt);
          s_send_sv += cnt;
          ev_base = n_ev; id_base = next_id; n_ev += rdlane(incl, 31);
          commit_fan(fan, v, 0, incl - cnt);
          next_id += rdlane(incl, 31);
        }
        poll();
      }
      if (phase == PH_DONE) break;
    }
  }

  // ---- epilogue: flush the partial row block, reduce counters, write stats + meta ----
  __syncthreads();
  {
    const u32 blk = n_rows >> 6;
    const u32 gi = blk * 64 + lane;
    if (gi < n_rows) reinterpret_cast<uint4 *>(g_rows)[gi] = stage[gi % STAGE_ROWS];
  }
  const u32 t_send_cl = wave_sum(s_send_cl), t_send_sv = wave_sum(s_send_sv);
  const u32 t_recv_cl = wave_sum(s_recv_cl), t_recv_sv = wave_sum(s_recv_sv);
  for (u32 b = 1; b <= MSIM_FLAG_JOURNAL_OVERFLOW; b <<= 1) if (__ballot((my_flags & b) != 0)) flags |= b;
  if (lane == 0) {
    msim_net_stats st;
    st.all_send = (u64)t_send_cl + t_send_sv; st.all_recv = (u64)t_recv_cl + t_recv_sv;
    st.clients_send = t_send_cl; st.clients_recv = t_recv_cl;
    st.servers_send = t_send_sv; st.servers_recv = t_recv_sv;
    p.stats[inst] = st;
    msim_inst_meta m; m.n_rows = n_rows; m.n_payload_words = n_payload; m.flags = flags; m.n_rounds = rounds;
    m.n_events = jcap ? n_ev : 0; m.reserved[0] = 0; m.reserved[1] = 0; m.reserved[2] = 0;
    p.meta[inst] = m;
  }
}

#include "sim_kernel_colo.inc"
#include "sim_kernel_raft.inc"
#include "sim_kernel_wide.inc"
#include "sim_kernel_txn.inc"
#include "sim_kernel_mk.inc"
#include "sim_kernel_hat.inc"
#include "sim_kernel_kafka.inc"
#include "sim_kernel_svc.inc"

// =====================================================================================================
// Host runtime
// =====================================================================================================
static void set_err(char *err, size_t n, const char *msg) { if (err && n) std::snprintf(err, n, "%s", msg); }

extern "C" int msim_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static void free_buffers(msim_ctx *c) {
  if (c->d_rows) (void)hipFree(c->d_rows);
  if (c->d_payload) (void)hipFree(c->d_payload);
  if (c->d_stats) (void)hipFree(c->d_stats);
  if (c->d_meta) (void)hipFree(c->d_meta);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_check) (void)hipFree(c->d_check);
  if (c->d_journal) (void)hipFree(c->d_journal);
  if (c->h_rows) (void)hipHostFree(c->h_rows);
  if (c->h_payload) (void)hipHostFree(c->h_payload);
  if (c->h_stats) (void)hipHostFree(c->h_stats);
  if (c->h_meta) (void)hipHostFree(c->h_meta);
  if (c->h_check) (void)hipHostFree(c->h_check);
  if (c->h_journal) (void)hipHostFree(c->h_journal);
  if (c->d_check_scratch) (void)hipFree(c->d_check_scratch);
  c->d_check_scratch = nullptr; c->cap_check_scratch = 0;
  if (c->d_compact) (void)hipFree(c->d_compact);
  if (c->d_off) (void)hipFree(c->d_off);
  if (c->d_compact2) (void)hipFree(c->d_compact2);
  if (c->d_off2) (void)hipFree(c->d_off2);
  c->d_compact2 = nullptr; c->d_off2 = nullptr; c->cap_compact2 = c->cap_off2 = 0;
  if (c->d_grows) (void)hipFree(c->d_grows);
  if (c->d_gpay) (void)hipFree(c->d_gpay);
  if (c->d_goff) (void)hipFree(c->d_goff);
  c->d_grows = c->d_gpay = nullptr; c->d_goff = nullptr; c->cap_grows = c->cap_gpay = c->cap_goff = 0;
  c->d_compact = nullptr; c->d_off = nullptr;
  c->cap_compact = c->cap_off = c->cap_h_rows = c->cap_h_payload = c->cap_h_journal = c->cap_h_meta = 0;
  delete[] c->h_row_off; delete[] c->h_pay_off; delete[] c->h_ev_off;
  c->d_journal = nullptr; c->h_journal = nullptr; c->h_ev_off = nullptr;
  c->d_rows = nullptr; c->d_payload = nullptr; c->d_stats = nullptr; c->d_meta = nullptr; c->d_scratch = nullptr; c->d_check = nullptr;
  c->h_rows = nullptr; c->h_payload = nullptr; c->h_stats = nullptr; c->h_meta = nullptr; c->h_check = nullptr;
  c->h_row_off = nullptr; c->h_pay_off = nullptr;
  c->cap_inst = 0;
}

extern "C" int msim_create(const msim_config *cfg, int device, msim_ctx **out, char *err, size_t errlen) {
  if (!cfg || !out) { set_err(err, errlen, "null argument"); return MSIM_E_INVALID; }
  *out = nullptr;
  msim_config c = *cfg;
  int rc = msim_config_finalize(&c, err, errlen);
  if (rc != MSIM_OK) return rc;
  if (c.node_program == MSIM_NODE_TXN_MULTI_KEY && (c.concurrency != c.n_nodes || c.n_nodes > 30)) {
    set_err(err, errlen, "multi_key_txn: one worker per node and at most 30 nodes (two service lanes) in this build");
    return MSIM_E_UNSUPPORTED;
  }
  const uint32_t slots = c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes;
  // wide clusters (33..127 nodes): two node/client pairs per lane, one worker per node: the g-set CRDT and fire-and-forget broadcast
  const bool wide_prog = c.node_program == MSIM_NODE_G_SET || c.node_program == MSIM_NODE_BCAST_FF || c.node_program == MSIM_NODE_BCAST_FF_ECHOBACK ||
                         c.node_program == MSIM_NODE_BCAST_ACK_RETRY || c.node_program == MSIM_NODE_BCAST_RPC_ALL || c.node_program == MSIM_NODE_PN_COUNTER;
  const bool wide = c.n_nodes > 32 && c.n_nodes <= 127 && wide_prog && c.concurrency == c.n_nodes;
  const uint32_t svc_lanes = c.node_program == MSIM_NODE_LIN_KV_PROXY || c.node_program == MSIM_NODE_TSO_IDS ? 1 : 0;  // the service has a lane of its own after the client slots
  if (!wide && (c.n_nodes > 32 || c.n_nodes + slots + svc_lanes > 64)) {
    set_err(err, errlen, "this build maps one cluster to one wavefront: n_nodes <= 32 and n_nodes + max(concurrency, n_nodes) <= 64 "
                         "(g-set, the counters and the broadcast programs with concurrency == n_nodes: up to 127 nodes)");
    return MSIM_E_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err(err, errlen, "no HIP device visible: libmaelsim has no CPU path"); return MSIM_E_NO_DEVICE; }
  if (device < 0 || device >= ndev) { set_err(err, errlen, "device index out of range"); return MSIM_E_INVALID; }
  msim_ctx *ctx = new (std::nothrow) msim_ctx();
  if (!ctx) return MSIM_E_NOMEM;
  ctx->cfg = c; ctx->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev2);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev3);
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(d_log2_q24), msim_log2_q24, sizeof(msim_log2_q24));
  if (e != hipSuccess) { set_err(err, errlen, hipGetErrorString(e)); delete ctx; return MSIM_E_HIP; }
  *out = ctx;
  return MSIM_OK;
}

// sim_kernel_wide<NET_RANDOM, BCAST, NEM, SETL> for this configuration
template <bool NR, int BC, bool NM, bool SL>
static hipError_t launch_wide_one(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sim_kernel_wide<NR, BC, NM, SL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((sim_kernel_wide<NR, BC, NM, SL>), dim3(n), dim3(64), lds, st, kp);
  return hipGetLastError();
}
template <int BC, bool SL>
static hipError_t launch_wide2(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  const msim_config &c = kp.cfg;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0, nem = c.nemesis_mask != 0;
  if (nem) return rnd ? launch_wide_one<true, BC, true, SL>(kp, n, lds, st) : launch_wide_one<false, BC, true, SL>(kp, n, lds, st);
  return rnd ? launch_wide_one<true, BC, false, SL>(kp, n, lds, st) : launch_wide_one<false, BC, false, SL>(kp, n, lds, st);
}
// Which wide clusters keep their nodes' sets in LDS (SETL): g-set, when sets + client inboxes + the LDS part of the queues leave a CU
// at least four clusters (40 KiB each); MSIM_DEV_FLAGS bit 14 keeps the sets in HBM scratch.  Measured (profiles/r03k_wide_sets.txt):
// cfg3 417 -> 348 ms per 16384 clusters.  Fire-and-forget broadcast stays in HBM scratch: its set traffic is one word per delivery,
// and at 20.8 KiB of LDS per cluster a CU holds 7 clusters where a batch of 2048 needs 8 — 121 -> 198 ms per 2048 clusters at n = 100.
static bool wide_sets_in_lds(const msim_config &c, uint32_t dev_flags) {
  if (c.n_nodes <= 32 || (dev_flags & 0x4000u)) return false;
  if (c.node_program != MSIM_NODE_G_SET) return false;
  const size_t bytes = ((size_t)c.n_nodes * c.inbox_capacity + (size_t)c.n_nodes * CLIENT_INBOX_CAP) * 16 + (size_t)c.n_nodes * (c.max_values / 32) * 4 + (c.nemesis_mask ? 512 : 0) + 16;
  return bytes <= 40 * 1024;
}
template <int BC>
static hipError_t launch_wide(msim_ctx *, const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  if constexpr (BC == 0) { if (wide_sets_in_lds(kp.cfg, kp.dev_flags)) return launch_wide2<BC, true>(kp, n, lds, st); }
  return launch_wide2<BC, false>(kp, n, lds, st);
}

// multi-key transactional node: thunk ids a node may hand out (every attempt of a transaction writes its keys again: x4 for the
// retries) and the slots of its thunk cache (what it wrote + what it read: twice that, load factor 1/2)
static uint32_t mk_tcap(const msim_config &c) {
  const double ops = (double)c.rate_mhz * (double)c.time_limit_ms / 1e6 * 1.125 + 64.0;
  uint32_t t = 64; while (t < 4.0 * ops * c.max_txn_length / c.n_nodes + 64.0) t <<= 1;
  return t;
}
static uint32_t mk_ccap(const msim_config &c) { return 4 * mk_tcap(c); }

// per-instance scratch = [protocol scratch][spill area: n_nodes x spill_capacity envelopes]
static uint32_t raft_log_cap(const msim_config &c) {  // every client op is appended at most once, by the leader that takes it
  const double expected = (double)c.rate_mhz * (double)c.time_limit_ms / 1e6;
  return (uint32_t)(expected + expected / 8.0) + 64 + 8;
}
static uint64_t proto_scratch_words(const msim_config &c) {
  uint64_t w = 4;
  if (c.node_program == MSIM_NODE_RAFT) w = (uint64_t)c.n_nodes * raft_log_cap(c) * 2 + (uint64_t)c.n_nodes * R_ARENA_WORDS;
  if (c.node_program == MSIM_NODE_BCAST_ACK_RETRY) w = (uint64_t)c.n_nodes * c.max_values * 3;
  if (c.node_program == MSIM_NODE_TXN_SINGLE_KEY) w = (uint64_t)c.max_values * (c.max_writes_per_key + 1);  // elements + counts per key
  if (c.node_program == MSIM_NODE_TXN_MULTI_KEY)   // elements, counts, map position, entry version, thunk counts, thunk versions + ids, the nodes' caches, the replica bytes
    w = (uint64_t)c.max_values * (c.max_writes_per_key + 4 + 2 * (c.max_writes_per_key + 1)) + (uint64_t)c.n_nodes * mk_ccap(c) + (uint64_t)c.n_nodes * mk_tcap(c) / 4 + 4 +
        (uint64_t)c.n_nodes * (MK_SLOTS - MK_SL) * mk_slot_words(8);   // + the transaction slots that are not in LDS
  if (c.node_program == MSIM_NODE_KAFKA) w = (uint64_t)KF_KEYS * (2 * (c.max_writes_per_key + 1) + 1);   // the logs + the committed-offset lists of the keys
  if (c.node_program == MSIM_NODE_TXN_RW_HAT) {  // registers per node + txn table + pending masks (bytes) + replicate lists
    const uint64_t G = c.max_rows / 2;
    w = (uint64_t)c.n_nodes * c.max_values + 2 * G + ((uint64_t)c.n_nodes * G + 3) / 4 + c.replication_words;
  }
  if (c.node_program == MSIM_NODE_G_SET || c.node_program == MSIM_NODE_PN_COUNTER) {
    const uint64_t total_ms = (uint64_t)c.time_limit_ms + c.quiesce_ms + 2ull * c.client_timeout_ms;
    const uint64_t ticks = total_ms / 5000 + 3;
    w = ticks * c.n_nodes * (c.max_values / 32);
    if (c.n_nodes > 32) w = ((w + 3) & ~3ull) + (((uint64_t)c.n_nodes * (c.max_values / 32) + 3) & ~3ull);  // wide clusters: + the nodes' sets
  }
  const bool bcast_ff = c.node_program == MSIM_NODE_BCAST_FF || c.node_program == MSIM_NODE_BCAST_FF_ECHOBACK;
  const bool bcast_rpc = c.node_program == MSIM_NODE_BCAST_ACK_RETRY || c.node_program == MSIM_NODE_BCAST_RPC_ALL;
  if (bcast_ff && c.n_nodes > 32) w = 4 + (((uint64_t)c.n_nodes * (c.max_values / 32) + 3) & ~3ull);  // wide clusters: the nodes' sets
  // wide clusters, acknowledged gossip: 128-bit unacked masks per (node, value), the retry FIFO, the nodes' sets
  if (bcast_rpc && c.n_nodes > 32) w = (uint64_t)c.n_nodes * c.max_values * 6 + (((uint64_t)c.n_nodes * (c.max_values / 32) + 3) & ~3ull);
  return (w + 3) & ~3ull;  // keep the spill area 16-byte aligned
}
static uint64_t scratch_words(const msim_config &c) {
  const uint64_t queues = c.n_nodes + (c.node_program == MSIM_NODE_KAFKA || c.node_program == MSIM_NODE_TXN_SINGLE_KEY || c.node_program == MSIM_NODE_LIN_KV_PROXY || c.node_program == MSIM_NODE_TSO_IDS ? 1 : c.node_program == MSIM_NODE_TXN_MULTI_KEY ? 2 : 0);  // + the services
  uint64_t w = proto_scratch_words(c) + queues * c.spill_capacity * 4;
  if (msim_raft4_eligible(c)) w += msim_raft4_extra_scratch_words(c);   // raft4.hip keeps fewer envelopes in LDS
  if (msim_txn8_eligible(c)) w += msim_txn8_extra_scratch_words(c);     // txn8.hip likewise
  if (msim_mk8_eligible(c)) w += msim_mk8_extra_scratch_words(c);       // mk8.hip likewise
  if (msim_hat8_eligible(c)) w += msim_hat8_extra_scratch_words(c);     // hat8.hip likewise
  if (msim_uid8_eligible(c)) w += msim_uid8_extra_scratch_words(c);     // uid8.hip likewise
  if (msim_crdt8_eligible(c)) w += msim_crdt8_extra_scratch_words(c);   // crdt8.hip likewise
  if (msim_bcast8_eligible(c)) w += msim_bcast8_extra_scratch_words(c); // bcast8.hip likewise
  return w;
}

static int ensure_buffers(msim_ctx *ctx, uint32_t n) {
  if (n <= ctx->cap_inst) return MSIM_OK;
  free_buffers(ctx);
  const msim_config &c = ctx->cfg;
  ctx->scratch_words_per_inst = scratch_words(c);
  MSIM_HIP_TRY(ctx, hipMalloc(&ctx->d_rows, (size_t)n * c.max_rows * sizeof(msim_op)));
  MSIM_HIP_TRY(ctx, hipMalloc(&ctx->d_payload, (size_t)n * c.max_payload_words * 4));
  MSIM_HIP_TRY(ctx, hipMalloc(&ctx->d_stats, (size_t)n * sizeof(msim_net_stats)));
  MSIM_HIP_TRY(ctx, hipMalloc(&ctx->d_meta, (size_t)n * sizeof(msim_inst_meta)));
  MSIM_HIP_TRY(ctx, hipMalloc(&ctx->d_scratch, (size_t)n * ctx->scratch_words_per_inst * 4));
  MSIM_HIP_TRY(ctx, hipMalloc(&ctx->d_check, (size_t)n * sizeof(msim_check_result)));
  if (c.journal_capacity) MSIM_HIP_TRY(ctx, hipMalloc(&ctx->d_journal, (size_t)n * c.journal_capacity * sizeof(msim_event)));
  ctx->cap_inst = n;
  return MSIM_OK;
}

template <int PROG, bool NEM, bool NET_RANDOM>
static hipError_t launch3(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  // colocated layout when every worker is pinned to "its" node (concurrency == n_nodes, the default 1n)
  const bool colo = kp.C == kp.N;
  // FIFO queues (constant latency only) when deep queues are expected: the scan-based poll is cheaper for shallow ones
  constexpr bool CAN_FIFO = !NET_RANDOM && PROG != MSIM_NODE_BCAST_ACK_RETRY && PROG != MSIM_NODE_BCAST_RPC_ALL;
  static const char *force = std::getenv("MSIM_QUEUE");  // developer knob: "fifo" / "scan" override the choice below
  const bool fifo = CAN_FIFO && colo && (force && force[0] == 'f' ? true : force && force[0] == 's' ? false : kp.spill_cap >= 64);
  const void *fn = !colo ? reinterpret_cast<const void *>(&sim_kernel<PROG, NEM, NET_RANDOM>)
                 : fifo ? reinterpret_cast<const void *>(&sim_kernel_colo<PROG, NEM, NET_RANDOM, CAN_FIFO>)
                        : reinterpret_cast<const void *>(&sim_kernel_colo<PROG, NEM, NET_RANDOM, false>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (!colo) hipLaunchKernelGGL((sim_kernel<PROG, NEM, NET_RANDOM>), dim3(n), dim3(64), lds, st, kp);
  else if (fifo) hipLaunchKernelGGL((sim_kernel_colo<PROG, NEM, NET_RANDOM, CAN_FIFO>), dim3(n), dim3(64), lds, st, kp);
  else hipLaunchKernelGGL((sim_kernel_colo<PROG, NEM, NET_RANDOM, false>), dim3(n), dim3(64), lds, st, kp);
  return hipGetLastError();
}
template <int PROG>
static hipError_t launch(const KParams &kp, uint32_t n, size_t lds, hipStream_t st) {
  const bool rnd = kp.cfg.latency_dist != MSIM_LAT_CONSTANT || kp.cfg.p_loss_q32 != 0;
  if (kp.cfg.nemesis_mask) return rnd ? launch3<PROG, true, true>(kp, n, lds, st) : launch3<PROG, true, false>(kp, n, lds, st);
  return rnd ? launch3<PROG, false, true>(kp, n, lds, st) : launch3<PROG, false, false>(kp, n, lds, st);
}

static int run_impl(msim_ctx *ctx, uint64_t first, uint32_t n, hipStream_t st, bool blocking) {
  if (!ctx) return MSIM_E_INVALID;
  if (n == 0) { ctx->err = "n_instances must be > 0"; return MSIM_E_INVALID; }
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_buffers(ctx, n);
  if (rc != MSIM_OK) return rc;
  const msim_config &c = ctx->cfg;
  KParams kp;
  std::memset(&kp, 0, sizeof kp);
  kp.cfg = c; kp.first_instance = first;
  kp.rows = ctx->d_rows; kp.payload = ctx->d_payload; kp.stats = ctx->d_stats; kp.meta = ctx->d_meta;
  kp.scratch = ctx->d_scratch; kp.scratch_words = ctx->scratch_words_per_inst;
  kp.journal = reinterpret_cast<uint4 *>(ctx->d_journal);
  kp.N = c.n_nodes; kp.C = c.concurrency; kp.CS = c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes;
  kp.W = c.max_values / 32;
  kp.cap_node = c.inbox_capacity; kp.spill_cap = c.spill_capacity; kp.spill_off = proto_scratch_words(c);
  const uint64_t period_us = 1000000000ull / (c.rate_mhz ? c.rate_mhz : 1);
  if (2 * period_us > 0xFFFFFFFFull || 2000ull * c.nemesis_interval_ms > 0xFFFFFFFFull) { ctx->err = "rate too low / nemesis interval too long for u32 microseconds"; return MSIM_E_INVALID; }
  kp.gen_period2_us = (u32)(2 * period_us);
  kp.nem_period2_us = (u32)(2000ull * c.nemesis_interval_ms);
  const bool is_raft = c.node_program == MSIM_NODE_RAFT;
  kp.raft_log_cap = is_raft ? raft_log_cap(c) : 0;
  kp.dev_flags = msim_dev_flags(ctx);
  const bool wide = c.n_nodes > 32;
  const bool wide_setl = wide_sets_in_lds(c, kp.dev_flags);   // (no row staging in that layout)
  size_t off = wide_setl ? 0 : (wide ? WIDE_STAGE_ROWS : STAGE_ROWS) * 16;
  const bool is_txn = c.node_program == MSIM_NODE_TXN_SINGLE_KEY, is_px = c.node_program == MSIM_NODE_LIN_KV_PROXY || c.node_program == MSIM_NODE_TSO_IDS;
  const bool is_hat = c.node_program == MSIM_NODE_TXN_RW_HAT, is_mk = c.node_program == MSIM_NODE_TXN_MULTI_KEY, is_kf = c.node_program == MSIM_NODE_KAFKA;
  kp.mk_tcap = is_mk ? mk_tcap(c) : 0; kp.mk_ccap = is_mk ? mk_ccap(c) : 0;
  kp.off_inbox = (u32)off;
  off += is_mk ? ((size_t)(kp.N + 2) * kp.cap_node + (size_t)kp.N * T_CLIENT_CAP) * 16 : is_hat ? ((size_t)kp.N * kp.cap_node + (size_t)kp.N * T_CLIENT_CAP) * 16 : is_px ? ((size_t)(kp.N + 1) * kp.cap_node + (size_t)kp.CS * R_CLIENT_CAP) * 16 : (is_txn || is_kf) ? ((size_t)(kp.N + 1) * kp.cap_node + (size_t)kp.N * T_CLIENT_CAP) * 16
                : ((size_t)kp.N * kp.cap_node + (size_t)kp.CS * (is_raft ? R_CLIENT_CAP : CLIENT_INBOX_CAP)) * 16;
  kp.off_seen = (u32)off;
  off += is_mk ? ((size_t)kp.N * MK_SL * mk_slot_words(mk_keys_for(c)) + (size_t)kp.N * mk_keys_for(c) * 3 + 36) * 4   // transactions in flight (the first MK_SL per node), a round's messages per node, the generator's key pool
       : is_hat ? 36 * 4   // the generator's key pool
       : is_px ? (size_t)kp.N * PX_SLOTS * 8 + 34 * 256 + 64 * 4   // callbacks per node + service states + seq-kv indices
       : is_txn ? (size_t)kp.N * TXN_SLOTS * 16 + 36 * 4   // transactions in flight per node + the generator's key pool
       : is_kf ? ((size_t)kp.N * KF_SLOTS * KSW + 2 * (size_t)kp.N * KF_KEYS + 36 + 2 * KF_KEYS) * 4   // request handlers, offset caches, client offsets, key pool, lin-kv lengths
       : is_raft ? (size_t)kp.N * 256 + (size_t)kp.N * kp.N * 3 * 4   // KV state + next/match index + append_entries refs
       : wide ? (wide_setl ? (size_t)kp.N * kp.W * 4 : 0)   // the sets of a wide cluster live in HBM scratch unless they fit LDS (wide_sets_in_lds)
                 : (size_t)kp.N * kp.W * 4;
  off = (off + 15) & ~(size_t)15;
  kp.off_misc = (u32)off; if (c.nemesis_mask) off += (wide ? 128 : 64) * 4;  // shuffle scratch, only the partition nemesis needs it
  const size_t lds = off;
  if (lds > 160 * 1024) { ctx->err = "cluster state exceeds the 160 KiB LDS of a CU (lower inbox_capacity / max_values)"; return MSIM_E_INVALID; }

  if (blocking) MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));
  hipError_t e;
#ifdef MSIM_ISA_PROBE  // developer hook (tools/isa_probe.sh): instantiate only one kernel so its ISA compiles in seconds
#ifdef MSIM_ISA_PROBE_WIDE   // BASELINE cfg3's kernel
  hipLaunchKernelGGL((sim_kernel_wide<true, 0, false, true>), dim3(n), dim3(64), lds, st, kp);
#else
  hipLaunchKernelGGL((sim_kernel_colo<MSIM_NODE_BCAST_FF, false, false, false>), dim3(n), dim3(64), lds, st, kp);
#endif
  e = hipGetLastError();
#else
  // the headline layout: two clusters per wavefront (duo.hip); MSIM_DEV_FLAGS bit 9 keeps the one-cluster kernels
  e = MSIM_LAYOUT_DOES_NOT_FIT;
  if (msim_duo_eligible(c) && !(kp.dev_flags & 0x200u) && !((kp.dev_flags & 0x8000u) && msim_bcast8_eligible(c))) {   // (bit 15: small clusters eight per wavefront instead)
    e = msim_launch_duo(kp, n, st);
    if (e == MSIM_LAYOUT_DOES_NOT_FIT && (kp.dev_flags & 0x400u)) { ctx->err = "MSIM_DEV_FLAGS bit 10: the two-clusters-per-wavefront layout was required but this cluster state does not fit it"; return MSIM_E_UNSUPPORTED; }
  }
  // the broadcast programs at the tutorial's cluster sizes: eight clusters per wavefront (bcast8.hip) where the headline layout does not apply
  if (e == MSIM_LAYOUT_DOES_NOT_FIT && msim_bcast8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_bcast8(kp, n, st);
  // Raft: four clusters per wavefront (raft4.hip) when a cluster fits a 16-lane group
  if (msim_raft4_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_raft4(kp, n, st);
  // txn-list-append: eight clusters per wavefront (txn8.hip) when a cluster fits an 8-lane group
  if (msim_txn8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_txn8(kp, n, st);
  // the canonical txn-list-append node: eight clusters per wavefront (mk8.hip) when a cluster fits an 8-lane group
  if (msim_mk8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_mk8(kp, n, st);
  // txn-rw-register over the highly-available-transactions node: eight clusters per wavefront (hat8.hip)
  if (msim_hat8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_hat8(kp, n, st);
  // echo / unique-ids (flake ids): eight clusters per wavefront (uid8.hip) for large batches of small clusters
  if (msim_uid8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_uid8(kp, n, st);
  // g-set / pn-counter / g-counter: eight clusters per wavefront (crdt8.hip) for large batches of small clusters
  if (msim_crdt8_eligible(c) && !(kp.dev_flags & 0x200u)) e = msim_launch_crdt8(kp, n, st);
  if (e == MSIM_LAYOUT_DOES_NOT_FIT && (kp.dev_flags & 0x400u) && is_raft) { ctx->err = "MSIM_DEV_FLAGS bit 10: the four-clusters-per-wavefront Raft layout was required but does not apply"; return MSIM_E_UNSUPPORTED; }
  if (e == MSIM_LAYOUT_DOES_NOT_FIT) switch (c.node_program) {   // not eligible, or the cluster state does not fit the duo layout
    case MSIM_NODE_ECHO: e = launch<MSIM_NODE_ECHO>(kp, n, lds, st); break;
    case MSIM_NODE_BCAST_FF:
    case MSIM_NODE_BCAST_FF_ECHOBACK:
      if (wide) {
        e = launch_wide<1>(ctx, kp, n, lds, st);
      } else if (c.node_program == MSIM_NODE_BCAST_FF) e = launch<MSIM_NODE_BCAST_FF>(kp, n, lds, st);
      else e = launch<MSIM_NODE_BCAST_FF_ECHOBACK>(kp, n, lds, st);
      break;
    case MSIM_NODE_BCAST_ACK_RETRY: e = wide ? launch_wide<2>(ctx, kp, n, lds, st) : launch<MSIM_NODE_BCAST_ACK_RETRY>(kp, n, lds, st); break;
    case MSIM_NODE_BCAST_RPC_ALL: e = wide ? launch_wide<3>(ctx, kp, n, lds, st) : launch<MSIM_NODE_BCAST_RPC_ALL>(kp, n, lds, st); break;
    case MSIM_NODE_G_SET:
      if (wide) {
        e = launch_wide<0>(ctx, kp, n, lds, st);
      } else e = launch<MSIM_NODE_G_SET>(kp, n, lds, st);
      break;
    case MSIM_NODE_PN_COUNTER: e = wide ? launch_wide<4>(ctx, kp, n, lds, st) : launch<MSIM_NODE_PN_COUNTER>(kp, n, lds, st); break;
    case MSIM_NODE_FLAKE_IDS: e = launch<MSIM_NODE_FLAKE_IDS>(kp, n, lds, st); break;
    case MSIM_NODE_RAFT: {
      const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
      if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((raft_kernel<true, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((raft_kernel<true, false>), dim3(n), dim3(64), lds, st, kp); }
      else { if (rnd) hipLaunchKernelGGL((raft_kernel<false, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((raft_kernel<false, false>), dim3(n), dim3(64), lds, st, kp); }
      e = hipGetLastError();
    } break;
    case MSIM_NODE_LIN_KV_PROXY: {
      const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
      if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((svc_kernel<true, true, false>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((svc_kernel<true, false, false>), dim3(n), dim3(64), lds, st, kp); }
      else { if (rnd) hipLaunchKernelGGL((svc_kernel<false, true, false>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((svc_kernel<false, false, false>), dim3(n), dim3(64), lds, st, kp); }
      e = hipGetLastError();
    } break;
    case MSIM_NODE_TSO_IDS: {   // unique-ids over the lin-tso service: the proxy's layout with the timestamp oracle on the service lane
      const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
      if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((svc_kernel<true, true, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((svc_kernel<true, false, true>), dim3(n), dim3(64), lds, st, kp); }
      else { if (rnd) hipLaunchKernelGGL((svc_kernel<false, true, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((svc_kernel<false, false, true>), dim3(n), dim3(64), lds, st, kp); }
      e = hipGetLastError();
    } break;
    case MSIM_NODE_TXN_SINGLE_KEY: {
      const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
      if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((txn_kernel<true, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((txn_kernel<true, false>), dim3(n), dim3(64), lds, st, kp); }
      else { if (rnd) hipLaunchKernelGGL((txn_kernel<false, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((txn_kernel<false, false>), dim3(n), dim3(64), lds, st, kp); }
      e = hipGetLastError();
    } break;
    case MSIM_NODE_TXN_MULTI_KEY: {
      const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
      void (*fn)(const KParams);
      if (mk_keys_for(c) == 4u) fn = c.nemesis_mask ? (rnd ? mk_kernel<true, true, 4> : mk_kernel<true, false, 4>) : (rnd ? mk_kernel<false, true, 4> : mk_kernel<false, false, 4>);
      else fn = c.nemesis_mask ? (rnd ? mk_kernel<true, true, 8> : mk_kernel<true, false, 8>) : (rnd ? mk_kernel<false, true, 8> : mk_kernel<false, false, 8>);
      e = lds > 64 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess;   // (above ~17 nodes)
      if (e == hipSuccess) { hipLaunchKernelGGL(fn, dim3(n), dim3(64), lds, st, kp); e = hipGetLastError(); }
    } break;
    case MSIM_NODE_KAFKA: {
      const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
      if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((kafka_kernel<true, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((kafka_kernel<true, false>), dim3(n), dim3(64), lds, st, kp); }
      else { if (rnd) hipLaunchKernelGGL((kafka_kernel<false, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((kafka_kernel<false, false>), dim3(n), dim3(64), lds, st, kp); }
      e = hipGetLastError();
    } break;
    case MSIM_NODE_TXN_RW_HAT: {
      const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
      if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((hat_kernel<true, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((hat_kernel<true, false>), dim3(n), dim3(64), lds, st, kp); }
      else { if (rnd) hipLaunchKernelGGL((hat_kernel<false, true>), dim3(n), dim3(64), lds, st, kp); else hipLaunchKernelGGL((hat_kernel<false, false>), dim3(n), dim3(64), lds, st, kp); }
      e = hipGetLastError();
    } break;
    default: ctx->err = "node program not built into this engine"; return MSIM_E_UNSUPPORTED;
  }
#endif
  if (e != hipSuccess) { ctx->err = std::string("kernel launch: ") + hipGetErrorString(e); return MSIM_E_HIP; }
  ctx->n_inst = n; ctx->first_instance = first;
  ctx->fetched = false; ctx->fetch_pending = false; ctx->checked = false; ctx->check_fetched = false; ctx->ran = true;
  if (blocking) {
    MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
    MSIM_HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    MSIM_HIP_TRY(ctx, hipEventElapsedTime(&ctx->sim_ms, ctx->ev0, ctx->ev1));
  }
  return MSIM_OK;
}

extern "C" int msim_run(msim_ctx *ctx, uint64_t first_instance, uint32_t n_instances) {
  if (!ctx) return MSIM_E_INVALID;
  return run_impl(ctx, first_instance, n_instances, ctx->stream, true);
}

extern "C" int msim_run_async(msim_ctx *ctx, uint64_t first_instance, uint32_t n_instances, void *hip_stream) {
  if (!ctx) return MSIM_E_INVALID;
  return run_impl(ctx, first_instance, n_instances, hip_stream ? (hipStream_t)hip_stream : ctx->stream, false);
}

extern "C" int msim_check(msim_ctx *ctx) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "msim_check before msim_run"; return MSIM_E_RANGE; }
  if (ctx->cfg.workload == MSIM_WL_LIN_KV) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_lin_kv_host(ctx) : msim_check_lin_kv_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_TXN_LIST_APPEND) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_txn_host(ctx) : msim_check_txn_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_TXN_RW_REGISTER) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_txn_host(ctx) : msim_check_rw_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_KAFKA) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_kafka_host(ctx) : msim_check_kafka_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_PN_COUNTER || ctx->cfg.workload == MSIM_WL_G_COUNTER) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_pn_host(ctx) : msim_check_pn_device(ctx);
  }
  if (ctx->cfg.workload == MSIM_WL_UNIQUE_IDS) {
    return (msim_dev_flags(ctx) & 0x800u) ?   // bit 11: keep the check on the host cores
           msim_check_unique_host(ctx) : msim_check_unique_device(ctx);
  }
  return msim_check_launch(ctx);
}

// Gathers the used prefix of every instance's slab into one contiguous device buffer (16-byte units):
// block (i, j) copies units [j*256*UNROLL ...) of instance i.  off[i] = first unit of instance i in dst.
__global__ void __launch_bounds__(256) compact_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const uint64_t *__restrict__ off,
                                                      uint64_t stride_units) {
  const u32 i = blockIdx.x;
  const uint64_t o = off[i], cnt = off[i + 1] - o;
  const uint4 *s = src + (size_t)i * stride_units;
  for (uint64_t k = (uint64_t)blockIdx.y * 256 + threadIdx.x; k < cnt; k += (uint64_t)gridDim.y * 256) dst[o + k] = s[k];
}
// same for 4-byte units whose per-instance slabs are not 16-byte aligned relative to their compacted position
__global__ void __launch_bounds__(256) compact_words_kernel(const u32 *__restrict__ src, u32 *__restrict__ dst, const uint64_t *__restrict__ off,
                                                            uint64_t stride_words) {
  const u32 i = blockIdx.x;
  const uint64_t o = off[i], cnt = off[i + 1] - o;
  const u32 *s = src + (size_t)i * stride_words;
  for (uint64_t k = (uint64_t)blockIdx.y * 256 + threadIdx.x; k < cnt; k += (uint64_t)gridDim.y * 256) dst[o + k] = s[k];
}

template <typename T>
static int grow_pinned(msim_ctx *ctx, T **buf, size_t *cap, size_t bytes) {
  if (*buf && *cap >= bytes) return MSIM_OK;
  if (*buf) { (void)hipHostFree(*buf); *buf = nullptr; *cap = 0; }
  const size_t want = bytes + bytes / 8 + 4096;  // pinning is slow (pages are faulted in and locked): grow with slack, reuse
  MSIM_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(buf), want));
  *cap = want;
  return MSIM_OK;
}

// one slab kind: compact on the device, then ONE device-to-host copy
// One slab kind to the host: the used prefix of every instance's slab is compacted on the device (phase 0) and crosses PCIe as ONE
// copy (phase 1).  Asynchronous on the context's stream; rows and payload have their own compaction buffer and offset table (pair
// 0 / 1), so that both compactions are queued before the first copy; the caller synchronises.
static int fetch_compacted(msim_ctx *ctx, const void *d_src, uint64_t stride_units, bool units16, const uint64_t *h_off, void *h_dst, int pair, int phase) {
  const uint32_t n = ctx->n_inst;
  const uint64_t total = h_off[n];
  if (!total) return MSIM_OK;
  void **d_compact = pair ? &ctx->d_compact2 : &ctx->d_compact; size_t *cap_compact = pair ? &ctx->cap_compact2 : &ctx->cap_compact;
  uint64_t **d_off = pair ? &ctx->d_off2 : &ctx->d_off; size_t *cap_off = pair ? &ctx->cap_off2 : &ctx->cap_off;
  const size_t unit = units16 ? 16 : 4, bytes = (size_t)total * unit;
  if (phase == 1) { MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_dst, *d_compact, bytes, hipMemcpyDeviceToHost, ctx->stream)); return MSIM_OK; }
  if (*cap_compact < bytes) {
    if (*d_compact) (void)hipFree(*d_compact);
    *d_compact = nullptr; *cap_compact = 0;
    MSIM_HIP_TRY(ctx, hipMalloc(d_compact, bytes + bytes / 8));
    *cap_compact = bytes + bytes / 8;
  }
  if (*cap_off < (size_t)(n + 1) * 8) {
    if (*d_off) (void)hipFree(*d_off);
    *d_off = nullptr; *cap_off = 0;
    MSIM_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(d_off), (size_t)(n + 1) * 8));
    *cap_off = (size_t)(n + 1) * 8;
  }
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(*d_off, h_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  const uint64_t avg = total / n + 1;
  const unsigned gy = (unsigned)((avg + 1023) / 1024 < 1 ? 1 : ((avg + 1023) / 1024 > 64 ? 64 : (avg + 1023) / 1024));
  if (units16) hipLaunchKernelGGL(compact_kernel, dim3(n, gy), dim3(256), 0, ctx->stream, static_cast<const uint4 *>(d_src), static_cast<uint4 *>(*d_compact), *d_off, stride_units);
  else hipLaunchKernelGGL(compact_words_kernel, dim3(n, gy), dim3(256), 0, ctx->stream, static_cast<const u32 *>(d_src), static_cast<u32 *>(*d_compact), *d_off, stride_units);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  return MSIM_OK;
}

// Device-side compaction for the multi-GPU gather (gather.cpp): the same kernels msim_fetch uses, into buffers that stay on
// the device.  Only the instance meta (32 B each) crosses PCIe, to size the buffers and build the offsets.
template <typename T>
static int grow_device(msim_ctx *ctx, T **buf, size_t *cap, size_t bytes) {
  if (*buf && *cap >= bytes) return MSIM_OK;
  if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
  const size_t want = bytes + bytes / 8 + 256;
  MSIM_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(buf), want));
  *cap = want;
  return MSIM_OK;
}
int msim_compact_on_device(msim_ctx *ctx, uint64_t *row_units, uint64_t *pay_words) {
  if (!ctx->ran) { ctx->err = "gather before msim_run"; return MSIM_E_RANGE; }
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const msim_config &c = ctx->cfg;
  const uint32_t n = ctx->n_inst;
  std::vector<msim_inst_meta> meta(n);
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  MSIM_HIP_TRY(ctx, hipMemcpy(meta.data(), ctx->d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
  std::vector<uint64_t> off(2 * (size_t)(n + 1));
  uint64_t ro = 0, po = 0;
  for (uint32_t i = 0; i < n; i++) { off[i] = ro; off[n + 1 + i] = po; ro += meta[i].n_rows; po += meta[i].n_payload_words; }
  off[n] = ro; off[2 * n + 1] = po;
  int rc;
  if ((rc = grow_device(ctx, &ctx->d_grows, &ctx->cap_grows, (size_t)ro * 16 + 16)) != MSIM_OK) return rc;
  if ((rc = grow_device(ctx, &ctx->d_gpay, &ctx->cap_gpay, (size_t)po * 4 + 16)) != MSIM_OK) return rc;
  if ((rc = grow_device(ctx, &ctx->d_goff, &ctx->cap_goff, off.size() * 8)) != MSIM_OK) return rc;
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_goff, off.data(), off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  const auto gy = [n](uint64_t total) { const uint64_t a = (total / (n ? n : 1) + 1 + 1023) / 1024; return (unsigned)(a < 1 ? 1 : a > 64 ? 64 : a); };
  if (ro) hipLaunchKernelGGL(compact_kernel, dim3(n, gy(ro)), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4 *>(ctx->d_rows),
                             static_cast<uint4 *>(ctx->d_grows), ctx->d_goff, (uint64_t)c.max_rows);
  if (po) hipLaunchKernelGGL(compact_words_kernel, dim3(n, gy(po)), dim3(256), 0, ctx->stream, ctx->d_payload, static_cast<u32 *>(ctx->d_gpay),
                             ctx->d_goff + (n + 1), (uint64_t)c.max_payload_words);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host-side offsets vector goes out of scope
  ctx->g_row_units = ro; ctx->g_pay_words = po;
  if (row_units) *row_units = ro;
  if (pay_words) *pay_words = po;
  return MSIM_OK;
}

// The histories' way to the host in two halves: msim_fetch_begin copies meta / stats (small), compacts rows and payload on the
// device and QUEUES the two big copies; msim_fetch waits for them (and fetches the journal).  A caller that has other work for the
// GPU — the next batch on another context — calls _begin right after the checker and _fetch when it needs the data: the
// compaction kernels are in the queue before the next batch's simulation, the copies run beside it.  msim_fetch alone does both.
extern "C" int msim_fetch_begin(msim_ctx *ctx) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "msim_fetch before msim_run"; return MSIM_E_RANGE; }
  if (ctx->fetched || ctx->fetch_pending) return MSIM_OK;
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const msim_config &c = ctx->cfg;
  const uint32_t n = ctx->n_inst;
  int rc;
  if (ctx->cap_h_meta < n) {  // meta + stats mirrors are sized by instance count
    if (ctx->h_meta) { (void)hipHostFree(ctx->h_meta); ctx->h_meta = nullptr; }
    if (ctx->h_stats) { (void)hipHostFree(ctx->h_stats); ctx->h_stats = nullptr; }
    ctx->cap_h_meta = 0;
    MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_meta, (size_t)n * sizeof(msim_inst_meta)));
    MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_stats, (size_t)n * sizeof(msim_net_stats)));
    ctx->cap_h_meta = n;
  }
  delete[] ctx->h_row_off; delete[] ctx->h_pay_off; delete[] ctx->h_ev_off;
  ctx->h_row_off = new uint64_t[n + 1]; ctx->h_pay_off = new uint64_t[n + 1]; ctx->h_ev_off = new uint64_t[n + 1];
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->h_meta, ctx->d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->h_stats, ctx->d_stats, (size_t)n * sizeof(msim_net_stats), hipMemcpyDeviceToHost));
  uint64_t ro = 0, po = 0, eo = 0;
  for (uint32_t i = 0; i < n; i++) {
    ctx->h_row_off[i] = ro; ctx->h_pay_off[i] = po; ctx->h_ev_off[i] = eo;
    ro += ctx->h_meta[i].n_rows; po += ctx->h_meta[i].n_payload_words;
    eo += ctx->h_meta[i].n_events < c.journal_capacity ? ctx->h_meta[i].n_events : c.journal_capacity;
  }
  ctx->h_row_off[n] = ro; ctx->h_pay_off[n] = po; ctx->h_ev_off[n] = eo;
  if ((rc = grow_pinned(ctx, &ctx->h_rows, &ctx->cap_h_rows, (size_t)(ro + 1) * sizeof(msim_op))) != MSIM_OK) return rc;
  if ((rc = grow_pinned(ctx, &ctx->h_payload, &ctx->cap_h_payload, (size_t)(po + 1) * 4)) != MSIM_OK) return rc;
  if (c.journal_capacity && (rc = grow_pinned(ctx, &ctx->h_journal, &ctx->cap_h_journal, (size_t)(eo + 1) * sizeof(msim_event))) != MSIM_OK) return rc;
  // only the used prefix of every instance's slab crosses PCIe, as one copy per slab kind: both compactions, then both copies
  for (int phase = 0; phase < 2; phase++) {
    if ((rc = fetch_compacted(ctx, ctx->d_rows, c.max_rows, true, ctx->h_row_off, ctx->h_rows, 0, phase)) != MSIM_OK) return rc;
    if ((rc = fetch_compacted(ctx, ctx->d_payload, c.max_payload_words, false, ctx->h_pay_off, ctx->h_payload, 1, phase)) != MSIM_OK) return rc;
    // the compaction kernels are off the CUs before this returns: a simulation launched while they still hold wave slots is
    // placed unevenly over the SIMDs and runs 25 % longer (measured); the copies behind them are DMA and disturb nothing
    if (phase == 0) MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->fetch_pending = true;
  return MSIM_OK;
}

extern "C" int msim_fetch(msim_ctx *ctx) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "msim_fetch before msim_run"; return MSIM_E_RANGE; }
  if (ctx->fetched) return MSIM_OK;
  int rc = msim_fetch_begin(ctx);
  if (rc != MSIM_OK) return rc;
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const msim_config &c = ctx->cfg;
  if (c.journal_capacity) {
    for (int phase = 0; phase < 2; phase++)
      if ((rc = fetch_compacted(ctx, ctx->d_journal, c.journal_capacity, true, ctx->h_ev_off, ctx->h_journal, 0, phase)) != MSIM_OK) return rc;
    MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->fetch_pending = false;
  ctx->fetched = true;
  return MSIM_OK;
}

extern "C" int msim_history(msim_ctx *ctx, uint32_t inst, const msim_op **ops, uint32_t *n_ops, const uint32_t **payload, uint32_t *n_words) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_history: not fetched or instance out of range"; return MSIM_E_RANGE; }
  if (ops) *ops = ctx->h_rows + ctx->h_row_off[inst];
  if (n_ops) *n_ops = ctx->h_meta[inst].n_rows;
  if (payload) *payload = ctx->h_payload + ctx->h_pay_off[inst];
  if (n_words) *n_words = ctx->h_meta[inst].n_payload_words;
  return MSIM_OK;
}

extern "C" int msim_net_stats_get(msim_ctx *ctx, uint32_t inst, msim_net_stats *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_net_stats_get: not fetched or instance out of range"; return MSIM_E_RANGE; }
  *out = ctx->h_stats[inst];
  return MSIM_OK;
}

extern "C" int msim_journal(msim_ctx *ctx, uint32_t inst, const msim_event **events, uint32_t *n_events) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_journal: not fetched or instance out of range"; return MSIM_E_RANGE; }
  if (!ctx->cfg.journal_capacity) { ctx->err = "msim_journal: journal_capacity is 0 (journal off)"; return MSIM_E_INVALID; }
  if (events) *events = ctx->h_journal + ctx->h_ev_off[inst];
  if (n_events) *n_events = (uint32_t)(ctx->h_ev_off[inst + 1] - ctx->h_ev_off[inst]);
  return MSIM_OK;
}

extern "C" int msim_meta(msim_ctx *ctx, uint32_t inst, msim_inst_meta *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  if (!ctx->fetched || inst >= ctx->n_inst) { ctx->err = "msim_meta: not fetched or instance out of range"; return MSIM_E_RANGE; }
  *out = ctx->h_meta[inst];
  return MSIM_OK;
}

extern "C" int msim_set_dev_flags(msim_ctx *ctx, uint32_t flags) {
  if (!ctx) return MSIM_E_INVALID;
  ctx->dev_flags = flags;
  return MSIM_OK;
}

extern "C" uint32_t msim_check_host_rechecks(const msim_ctx *ctx) { return ctx && ctx->checked && (ctx->cfg.workload == MSIM_WL_LIN_KV || ctx->cfg.workload == MSIM_WL_TXN_LIST_APPEND || ctx->cfg.workload == MSIM_WL_TXN_RW_REGISTER || ctx->cfg.workload == MSIM_WL_PN_COUNTER || ctx->cfg.workload == MSIM_WL_G_COUNTER) ? ctx->lin_host_rechecks : 0u; }

extern "C" int msim_check_results(msim_ctx *ctx, const msim_check_result **results, uint32_t *n) {
  if (!ctx) return MSIM_E_INVALID;
  if (!ctx->checked) { ctx->err = "msim_check_results before msim_check"; return MSIM_E_RANGE; }
  if (!ctx->check_fetched) {
    MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
    MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)ctx->n_inst * sizeof(msim_check_result)));
    MSIM_HIP_TRY(ctx, hipMemcpy(ctx->h_check, ctx->d_check, (size_t)ctx->n_inst * sizeof(msim_check_result), hipMemcpyDeviceToHost));
    ctx->check_fetched = true;
  }
  if (results) *results = ctx->h_check;
  if (n) *n = ctx->n_inst;
  return MSIM_OK;
}

extern "C" int msim_device_buffers_get(msim_ctx *ctx, msim_device_buffers *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "no run yet"; return MSIM_E_RANGE; }
  const msim_config &c = ctx->cfg;
  out->rows = ctx->d_rows; out->payload = ctx->d_payload; out->stats = ctx->d_stats; out->meta = ctx->d_meta;
  out->check = ctx->d_check; out->check_bytes = (uint64_t)ctx->n_inst * sizeof(msim_check_result);
  out->journal = ctx->d_journal; out->journal_bytes = (uint64_t)ctx->n_inst * c.journal_capacity * sizeof(msim_event);
  out->rows_bytes = (uint64_t)ctx->n_inst * c.max_rows * sizeof(msim_op);
  out->payload_bytes = (uint64_t)ctx->n_inst * c.max_payload_words * 4;
  out->stats_bytes = (uint64_t)ctx->n_inst * sizeof(msim_net_stats);
  out->meta_bytes = (uint64_t)ctx->n_inst * sizeof(msim_inst_meta);
  out->n_instances = ctx->n_inst; out->max_rows = c.max_rows; out->max_payload_words = c.max_payload_words; out->journal_capacity = c.journal_capacity;
  return MSIM_OK;
}

extern "C" int msim_last_kernel_ms(msim_ctx *ctx, float *sim_ms, float *check_ms) {
  if (!ctx) return MSIM_E_INVALID;
  if (sim_ms) *sim_ms = ctx->sim_ms;
  if (check_ms) *check_ms = ctx->check_ms;
  return MSIM_OK;
}

extern "C" int msim_get_config(const msim_ctx *ctx, msim_config *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  *out = ctx->cfg;
  return MSIM_OK;
}

extern "C" const char *msim_last_error(const msim_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" void msim_destroy(msim_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  free_buffers(ctx);
  msim_gather_free(ctx);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
  if (ctx->ev3) (void)hipEventDestroy(ctx->ev3);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// Validates the DPP encodings of wave_min / wave_incl_scan against shuffle-based references on the
// device.  Returns 0 when they agree, >0 = number of mismatching lanes, <0 = MSIM_E_*.
extern "C" int msim_selftest_wave(int device) {
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_NO_DEVICE;
  const int blocks = 64, n = blocks * 64;
  u32 *h = new u32[n], *d_in = nullptr, *d_out = nullptr;
  u64 x = 12345;
  for (int i = 0; i < n; i++) { x = mix64(x + i); h[i] = (u32)x; if (i % 7 == 0) h[i] = 0xFFFFFFFFu; }
  int bad = 0;
  if (hipMalloc(&d_in, n * 4) != hipSuccess || hipMalloc(&d_out, n * 4) != hipSuccess) { delete[] h; return MSIM_E_HIP; }
  (void)hipMemcpy(d_in, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(wave_selftest_kernel, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
  if (hipMemcpy(h, d_out, n * 4, hipMemcpyDeviceToHost) != hipSuccess) bad = MSIM_E_HIP;
  else for (int i = 0; i < n; i++) bad += h[i] != 0;
  (void)hipFree(d_in); (void)hipFree(d_out);
  delete[] h;
  return bad;
}
