// raft4.hip — FOUR Raft clusters per wavefront: the lin-kv workload over the Raft node program (SURVEY.md §8a row a16,
// BASELINE configs[3]: 5 nodes, concurrency 10) in 16-lane groups.
//
// Same program and the same round machinery as raft_kernel<> (sim_kernel_raft.inc, restated from demo/ruby/raft.rb:1-497 ==
// demo/python/raft.py:1-593; the network of net.clj:189-247, the clients of client.clj:41-172 / lin_kv.clj:40-85), round for
// round what DESIGN.md §2 and the CPU oracle specify.  What changes is the mapping: a 5-node cluster with its 10 client slots is
// 15 endpoints — one per lane of a 16-lane group — and a wavefront carries four of them.  raft_kernel<> paid ~850 vector and
// ~790 scalar instructions per round for 15 live lanes of 64 and was bound by instruction issue with 32 wavefronts per CU;
// here one instruction stream serves four clusters: everything that is uniform per CLUSTER (time, phase, generator, cursors)
// lives in VGPRs, a "ballot" is the cluster's 16-bit slice of the wave ballot, a lane of another group is never addressed
// (`ds_bpermute` inside the group replaces `v_readlane`), reductions and prefix sums are DPP row operations (a row = a group).
//
// Scope (engine.hip picks this kernel when all of it holds, else raft_kernel<> runs): n_nodes + max(concurrency, n_nodes) <= 16,
// net journal off.  Latency models, loss and the partition nemesis are all here.
//
// LDS of a wavefront: envelope queues (slot-major: slot s of lane e at [s * 64 + e], free of bank conflicts), per node the KV
// state (4 bits per key), next_index / match_index and the term runs of its log, per group a row staging ring (64 rows, 32-row
// coalesced appends) and the nemesis shuffle.  A queue holds RQ envelopes in LDS; what does not fit spills to HBM (nodes:
// inbox_capacity + spill_capacity in all, clients: 32, the oracle's limits).  Logs and append_entries bodies live in HBM scratch.
//
// A round of this program is one event per node, so what a round costs is the chain of dependent memory round trips in it
// (~2500 cycles each from HBM) and the instructions of the union of the paths its 64 lanes take.  Hence:
//   * the last four entries of a node's log stay in registers (what a leader replicates and what a node applies is nearly
//     always there); terms of older entries come from a run-length table in LDS (terms never decrease along a log) — the log in
//     HBM is written through and read only for backlogs;
//   * an append_entries body (header + entries) is written once per distinct next_index, not once per peer; a backlog of more
//     than four entries is copied by the 16 lanes of the cluster together; the receiver prefetches header and the first four
//     entries right after the round's time is known, so that the trip overlaps the scheduler and the clients' sends;
//   * request_vote travels in the envelope alone; append_entries_res echoes what the leader's closure captured (as the oracle
//     does: raft_nodes.inc "echo of the leader-side closure"), folded to (term matches, prev + count);
//   * median(match_index), "entries pending" and "can commit" are recomputed only in rounds in which the node did something.
//
// Envelope (16 B): x = deadline, y = (id << 8) | type, z = a, w = b | (src << 24).
//   request_vote        a = term | last_log_term << 16                         b = last_log_index
//   request_vote_res    a = term | candidacy term << 16 | granted << 31
//   append_entries      a = body ref (offset / 4 words | seq << 16)
//   append_entries_res  a = term | success << 16 | (request term == term) << 17   b = prev_log_index + number of entries
//   read/write/cas      a = key | v << 8 | v' << 16 (0xFF = nil)               b = msg_id
//   *_ok / error        a = value / code                                       b = in_reply_to
// Body in the sender's ring: uint4 {term | prev_log_term << 16, prev_log_index | count << 16, leader_commit | seq << 24, 0},
// then the entries (8 B each: term | type << 16 | key << 24, msg_id | client << 16 | v << 24 | v' << 28).
#include <hip/hip_runtime.h>

#include "wave_common.h"
#include "log2_table.h"

namespace {

__constant__ u32 r4_log2_q24[257];

constexpr u32 GS = 16u;           // lanes per cluster
constexpr u32 RQ = 8u;            // LDS envelopes per endpoint
constexpr u32 RK = 8u;            // term runs of a node's log kept in LDS (beyond that: terms are read from the log in HBM)
constexpr u32 R4_STAGE = 64u;     // staged history rows per cluster
constexpr u32 R4_CLIENT_CAP = 32u;   // Reusable lin-kv clients (lin_kv.clj:74-76) collect late replies between RPCs (the oracle's limit)
constexpr u32 R4_ARENA_WORDS = 16384u;
enum { R_NASCENT = 0, R_FOLLOWER, R_CANDIDATE, R_LEADER };
enum { M_WRITE = 14, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR, M_REQUEST_VOTE, M_REQUEST_VOTE_RES, M_APPEND_ENTRIES, M_APPEND_ENTRIES_RES };
enum { S_NODE = 11 };

struct R4Params {
  KParams k;
  u32 n_inst;
  u32 off_kv, off_nim, off_runs, off_stage, off_misc;   // LDS byte offsets (queues at 0)
  u32 nim_stride;                             // words of next/match/refs per cluster
  u32 arenas_off;                             // word offset of the nodes' body rings inside the per-instance scratch (16-byte aligned)
  u32 node_spill, client_spill;               // HBM spill entries per node / client behind the RQ LDS slots
  u64 client_spill_off;                       // word offset of the clients' spill area inside the per-instance scratch
  u32 round_limit;
};

__device__ __forceinline__ u32 r4_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = r4_log2_q24[idx], l1 = r4_log2_q24[idx + 1];
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

template <bool NEM, bool NET_RANDOM, int NN>   // NN: static bound of the loops over peers (n_nodes <= NN)
__global__ void __launch_bounds__(64) raft4_kernel(const R4Params rp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = rp.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 4, gbase = lane & 48u;
  const u32 N = p.N, C = p.C, CS = p.CS;
  const bool is_node = l < N;
  const bool is_client = l >= N && l < N + CS;
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
  const u32 rate = p.cfg.rate_mhz;
  const u32 log_cap = p.raft_log_cap;
  u32 rpc_timeout_ms = 10 * lat_mean; if (rpc_timeout_ms < 1000) rpc_timeout_ms = 1000;   // lin_kv.clj:54
  const u32 round_limit = rp.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  u32 *const my_log = g_scr + (size_t)(is_node ? l : 0) * log_cap * 2;                 // entry i at (i-1)*2
  u32 *const arenas = g_scr + rp.arenas_off;
  u32 *const my_arena = arenas + (size_t)(is_node ? l : 0) * R4_ARENA_WORDS;
  const u32 my_spill_cap = is_node ? rp.node_spill : (is_client ? rp.client_spill : 0u);
  uint4 *const my_spill = is_node ? reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)l * rp.node_spill
                                  : reinterpret_cast<uint4 *>(g_scr + rp.client_spill_off) + (size_t)(is_client ? slot : 0) * rp.client_spill;

  // LDS
  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                         // slot s at my_q[s * 64]
  unsigned char *const my_kv = smem + rp.off_kv + (grp * N + (is_node ? l : 0)) * 128;               // KVStore state, 4 bits per key, 0xF = absent
  u32 *const nim_g = reinterpret_cast<u32 *>(smem + rp.off_nim) + grp * rp.nim_stride;               // [node][peer][2] next/match, then aeref [node][peer]
  u32 *const my_runs = reinterpret_cast<u32 *>(smem + rp.off_runs) + lane;                           // run k at [k * 64]: first index | term << 16
  u32 *const my_nim = nim_g + (is_node ? l : 0) * N * 2;
  u32 *const aeref = nim_g + N * N * 2;
  uint4 *const stage = reinterpret_cast<uint4 *>(smem + rp.off_stage) + grp * R4_STAGE;
  u32 *const misc = reinterpret_cast<u32 *>(smem + rp.off_misc) + grp * GS;

  for (u32 i = lane; i < 4 * N * 32; i += 64) reinterpret_cast<u32 *>(smem + rp.off_kv)[i] = 0xFFFFFFFFu;
  for (u32 i = lane; i < 4 * rp.nim_stride; i += 64) reinterpret_cast<u32 *>(smem + rp.off_nim)[i] = 0;
  if (is_node && real) { my_log[0] = 0; my_log[1] = 0; }  // the default entry {term 0, op nil} (raft.py:109-112)
  my_runs[0] = 1u;                                        // ... is the first run: index 1, term 0
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- endpoint state ----
  bool has_c = false; u32 deliver_at = 0; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  // ---- raft node state (raft.py:197-226) ----
  u32 role = R_NASCENT, term = 0, commit_index = 0, last_applied = 1, rng_ctr = 0, log_n = 1, last_term = 0;
  int voted_for = -1, leader = -1;
  u32 election_deadline = 0, step_down_deadline = 0, last_replication = 0, votes = 0;
  u32 arena_head = 0, arena_seq = 0, act_time = INF;
  u32 t0x = 0, t0y = 0, t1x = 0, t1y = 0, t2x = 0, t2y = 0, t3x = 0, t3y = 0, tail_valid = 1;   // the last entries of the log (tl0 = entry log_n)
  u32 run_n = 1, lr_start = 1; bool runs_ovf = false;                                    // term runs; lr_start: first index of the last run
  u32 med = 0; bool can_commit = false, pending = false;                                 // leader: cached median(match_index) etc.
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_f = 0, c_value = 0, process = slot;
  u32 dest_node = is_client ? slot % N : 0; const u32 c_mod_n = C % N;
  u32 m_f = 0, m_value = 0, key_reg = INF;   // key_reg: the process id this thread registered on the current key
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, nem_next = 0, nem_j = 0, cur_key = 0, key_procs = 0;
  u32 loss_on = 0, next_id = 0, n_rows = 0, n_payload = 0, flags = 0, rounds = 0;
  bool alive = real;

  auto q_push = [&](const uint4 m) {
    if (in_n < RQ) { my_q[in_n * 64u] = m; in_n++; return; }
    if (sp_n < my_spill_cap) { my_spill[sp_n++] = m; return; }
    my_flags |= MSIM_FLAG_INBOX_OVERFLOW;
  };
  auto arrive = [&](u32 id, u32 type, u32 a, u32 b, u32 src) {
    u32 lat = 0;
    if (src < N && is_node) {
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * r4_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
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
    const bool elig = alive && (is_node || busy);
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

  // ---- raft helpers (node lanes) ----
  auto reset_election_deadline = [&]() {  // raft.py:251-253: now + 2 s * (random + 1)
    election_deadline = T + 2000000u + scale32(draw32(key, S_NODE, ((u64)l << 32) | rng_ctr++), 2000000u);
  };
  auto become_follower = [&]() { role = R_FOLLOWER; leader = -1; reset_election_deadline(); };           // :303-310
  auto maybe_step_down = [&](u32 remote) { if (term < remote) { term = remote; voted_for = -1; become_follower(); } };  // :259-272
  auto become_leader = [&]() {                                                                            // :323-336
    role = R_LEADER; leader = -1; last_replication = 0;
    for (u32 i = 0; i < N; i++) { my_nim[i * 2] = log_n + 1; my_nim[i * 2 + 1] = 0; }
    step_down_deadline = T + 2000000u;
  };
  auto arena_alloc = [&](u32 words) -> u32 {  // ring of message descriptors; returns offset in words (multiple of 4)
    words = (words + 3) & ~3u;
    if (words > R4_ARENA_WORDS) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; return 0; }
    if (arena_head + words > R4_ARENA_WORDS) arena_head = 0;
    const u32 off = arena_head; arena_head += words; arena_seq = (arena_seq + 1) & 0xFFu;
    return off;
  };
  // entry log_n - back (back < tail_valid) from the registers
#define TAIL_ENTRY(back) make_uint2((back) == 0u ? t0x : (back) == 1u ? t1x : (back) == 2u ? t2x : t3x, (back) == 0u ? t0y : (back) == 1u ? t1y : (back) == 2u ? t2y : t3y)
  // term of entry i (1 <= i <= log_n): terms never decrease along a log, so a few (first index, term) runs describe them all
  auto log_term = [&](u32 i) -> u32 {
    if (runs_ovf) return my_log[(i - 1) * 2] & 0xFFFFu;
    if (i >= lr_start) return last_term;
    u32 t = 0;
    for (u32 k = 0; k < run_n; k++) { const u32 r = my_runs[k * 64u]; if ((r & 0xFFFFu) <= i) t = r >> 16; }
    return t;
  };
  auto append_entry = [&](const uint2 e) {
    *reinterpret_cast<uint2 *>(my_log + log_n * 2) = e;
    const u32 t = e.x & 0xFFFFu;
    if (t != last_term) {
      if (!runs_ovf) {
        if (run_n < RK) { my_runs[run_n * 64u] = (log_n + 1) | (t << 16); run_n++; lr_start = log_n + 1; }
        else { runs_ovf = true; lr_start = INF; }
      }
      last_term = t;
    }
    log_n++;
    t3x = t2x; t3y = t2y; t2x = t1x; t2y = t1y; t1x = t0x; t1y = t0y; t0x = e.x; t0y = e.y; tail_valid = min(tail_valid + 1u, 4u);
  };
  auto truncate_to = [&](u32 pi, u32 pterm) {   // pi < log_n and log_term(pi) == pterm
    if (!runs_ovf && pi < lr_start) {
      u32 nn = 1, st = 1;
      for (u32 k = 0; k < run_n; k++) { const u32 r = my_runs[k * 64u]; if ((r & 0xFFFFu) <= pi) { nn = k + 1; st = r & 0xFFFFu; } }
      run_n = nn; lr_start = st;
    }
    last_term = pterm; log_n = pi; tail_valid = 0;
  };
  auto kv_get = [&](u32 k) -> u32 { return ((u32)my_kv[k >> 1] >> ((k & 1u) * 4u)) & 0xFu; };
  auto kv_put = [&](u32 k, u32 v) { const u32 sh = (k & 1u) * 4u; my_kv[k >> 1] = (unsigned char)((my_kv[k >> 1] & ~(0xFu << sh)) | ((v & 0xFu) << sh)); };
  // leader: median(match_index incl. our own log size), biased low: the (n - majority)-th smallest (raft.py:30-34,237-241);
  // whether a peer lacks entries (replicate_log's 50 ms pace, else the 1 s heartbeat); whether advance_commit_index applies
  auto leader_cache = [&]() {
    u32 mtv[NN]; bool pend = false;
#pragma unroll
    for (int i = 0; i < NN; i++) {
      uint2 v = make_uint2(INF, INF);
      if ((u32)i < N) v = *reinterpret_cast<const uint2 *>(&my_nim[i * 2]);
      if ((u32)i == l) { v.x = INF; v.y = log_n; }
      pend |= v.x <= log_n;
      mtv[i] = v.y;
    }
    u32 m = 0; const u32 kth = N - (N / 2 + 1);
#pragma unroll
    for (int i = 0; i < NN; i++) {
      u32 rk = 0;
#pragma unroll
      for (int j = 0; j < NN; j++) rk += (mtv[j] < mtv[i] || (mtv[j] == mtv[i] && j < i)) ? 1u : 0u;
      if (rk == kth) m = mtv[i];
    }
    med = m; pending = pend;
    can_commit = false;
    if (commit_index < m) can_commit = log_term(m) == term;
  };

#ifdef R4_PROF
  u64 pacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; u32 wave_rounds = 0;
  u64 tprev = __builtin_readcyclecounter();
#define R4_MARK(i) { const u64 now_ = __builtin_readcyclecounter(); pacc[i] += now_ - tprev; tprev = now_; }
#else
#define R4_MARK(i)
#endif
  for (;;) {
    if (!__ballot(alive)) break;
#ifdef R4_PROF
    wave_rounds++;
#endif
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
    if (is_node) my_t = min(my_t, act_time);
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

    bool inv_row = false; u32 inv_packed = 0, inv_value = 0;
    bool cmp_row = false; u32 cmp_packed = 0, cmp_value = 0;
    u32 nem_rows = 0, nem_f = 0, nem_v1 = 0, nem_v2 = 0, nem_len2 = 0;

    auto complete = [&](u32 type, u32 err, u32 value) {
      busy = false;
      if (kind != K_OP) { if (type != MSIM_T_OK) my_flags |= MSIM_FLAG_ROUND_LIMIT; return; }
      cmp_row = true; cmp_packed = type | (c_f << 2) | (err << 7) | (process << 12); cmp_value = value;
      if (type == MSIM_T_INFO) { process += C; dest_node += c_mod_n; if (dest_node >= N) dest_node -= N; }  // Reusable: client stays open
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) complete(c_f == MSIM_F_READ ? MSIM_T_FAIL : MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_value);  // lin_kv.clj:52
    }
    bool normal = alive && !timeout_round;   // this cluster runs R1-R4 in this wave-round
    // an append_entries a node has committed to and is due is what it handles in R3 (R2 commits client requests only, and only to
    // nodes without one): fetch header and first entries of its body now, use them in R3
    uint4 dh = make_uint4(0, 0, 0, 0), de0 = dh, de1 = dh;
    if (is_node && normal && has_c && deliver_at <= T && (cm.y & 0x7Fu) == M_APPEND_ENTRIES) {
      const uint4 *dp = reinterpret_cast<const uint4 *>(arenas + (size_t)(cm.w >> 24) * R4_ARENA_WORDS + (cm.z & 0xFFFu) * 4);
      dh = dp[0]; de0 = dp[1]; de1 = dp[2];
    }
    R4_MARK(0)
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
          // [upstream] jepsen.tests.linearizable-register: one key per group of 2n threads; first n threads read, the
          // rest mix [w cas cas]; values 0..4; (gen/process-limit 20) retires a key after 20 distinct processes
          const u32 nfree = __popc(free_mask);
          const u32 kk = gen_k;
          const u64 h = draw64(key, S_GEN, kk);
          const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          const u32 pick = scale32(r_lo, nfree);
          const bool sel = gen && is_worker && !busy && (u32)__popc(free_mask & lt) == pick;
          const u32 selm = GB(sel);
          const u32 sl = selm ? (u32)__builtin_ctz(selm) : 0u;                       // the chosen lane of my group
          const u32 s_proc = GGET(process, sl), s_reg = GGET(key_reg, sl);
          bool fresh_key = false, key_ovf = false;
          if (gen && s_reg != s_proc) {  // this process has not used the current key yet
            if (key_procs == 20) {
              if (cur_key >= 255) key_ovf = true;   // keys travel in 8 bits (oracle: same flag, same stop)
              else { cur_key++; key_procs = 0; fresh_key = true; }
            }
            if (!key_ovf) key_procs++;
          }
          if (key_ovf) { flags |= MSIM_FLAG_VALUES_OVERFLOW; phase = PH_DONE; alive = false; normal = false; }
          if (fresh_key) key_reg = INF;
          const u64 h2 = draw64(key, S_GEN2, kk);
          const u32 v1 = scale32((u32)(h2 >> 32), 5), v2 = (((u32)(h2 >> 20) & 0xFFFu) * 5u) >> 12, kx = cur_key & 0xFFu;
          if (sel && !key_ovf) {
            key_reg = process;
            mark = true; kind = K_OP;
            if (slot < N) { m_f = MSIM_F_READ; m_value = kx | 0xFFFF00u; }
            else if (scale32((u32)h2, 3) == 0) { m_f = MSIM_F_WRITE; m_value = kx | (v1 << 8) | 0xFF0000u; }
            else { m_f = MSIM_F_CAS; m_value = kx | (v1 << 8) | (v2 << 16); }
          }
          if (gen && !key_ovf) { gen_k++; gen_next = T + __umulhi(r_hi, p.gen_period2_us); }
        }
      }

      R4_MARK(1)
      // ---- R2: marked clients invoke ----
      if (__ballot(mark && normal)) {
        const bool inv = mark && normal;
        u32 rq_dest = 0, rq_type = 0, rq_a = 0;
        if (inv) {
          mark = false; busy = true;
          if (kind == K_INIT) { rq_dest = slot; rq_type = M_INIT; next_msg_id = 0; }
          else {
            c_f = m_f; c_value = m_value;
            rq_dest = dest_node;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (c_f << 2) | (process << 12); inv_value = c_value;
            rq_type = c_f == MSIM_F_WRITE ? M_WRITE : c_f == MSIM_F_CAS ? M_CAS : M_READ;
            rq_a = c_value;
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

      R4_MARK(2)
      // ---- R3: one main-loop iteration per node (raft.py:577-585): a message first, else the first action that applies ----
      u32 fan_mask = 0, fan_type = 0, fan_a = 0, fan_b = 0;        // fan_type M_APPEND_ENTRIES: per-peer body refs in aeref[]
      u32 bulk_mask = 0;                                            // peers whose (new) body holds a backlog the cluster copies together
      bool rep = false; u32 rep_dest = 0, rep_type = 0, rep_a = 0, rep_b = 0, rep_src = l;
      const bool take = is_node && normal && has_c && deliver_at <= T;
      const u32 qsrc = cm.w >> 24, qb = cm.w & 0xFFFFFFu, qtype = cm.y & 0x7Fu, qa = cm.z;
      const bool act_now = is_node && normal && !take && act_time <= T;
      if (take) {
        has_c = false;
        if (qsrc >= N) s_recv_cl++; else s_recv_sv++;
        switch (qtype) {
          case M_INIT:  // raft_init, raft.py:432-447
            become_follower(); last_applied = 1;
            rep = true; rep_dest = qsrc; rep_type = M_INIT_OK; rep_b = qb; break;
          case M_REQUEST_VOTE: {  // raft.py:449-481
            const u32 r_term = qa & 0xFFFFu, llt = qa >> 16, lli = qb;
            maybe_step_down(r_term);
            u32 grant = 0;
            if (r_term < term) {}
            else if (voted_for >= 0) {}
            else if (llt < last_term) {}
            else if (llt == last_term && lli < log_n) {}
            else { grant = 1; voted_for = (int)qsrc; reset_election_deadline(); }
            rep = true; rep_dest = qsrc; rep_type = M_REQUEST_VOTE_RES;
            rep_a = term | ((r_term & 0x7FFFu) << 16) | (grant << 31);
          } break;
          case M_REQUEST_VOTE_RES: {  // closure of request_votes, raft.py:280-297
            const u32 r_term = qa & 0xFFFFu, cand = (qa >> 16) & 0x7FFFu, granted = qa >> 31;
            step_down_deadline = T + 2000000u;
            maybe_step_down(r_term);
            if (role == R_CANDIDATE && term == cand && r_term == term && granted) {
              votes |= 1u << qsrc;
              if ((u32)__popc(votes) >= N / 2 + 1) become_leader();
            }
          } break;
          case M_APPEND_ENTRIES: {  // raft.py:483-531; header and first entries were prefetched (dh, de0, de1)
            if ((dh.z >> 24) != ((qa >> 16) & 0xFFu)) my_flags |= MSIM_FLAG_ARENA_OVERRUN;
            const u32 r_term = dh.x & 0xFFFFu, prev_term = dh.x >> 16, prev_idx = dh.y & 0xFFFFu, cnt = dh.y >> 16, lcommit = dh.z & 0xFFFFFFu;
            maybe_step_down(r_term);
            u32 ok = 0;
            if (r_term >= term) {
              leader = (int)qsrc; reset_election_deadline();
              if (prev_idx >= 1 && prev_idx <= log_n && log_term(prev_idx) == prev_term) {
                if (prev_idx + cnt > log_cap) my_flags |= MSIM_FLAG_ROWS_OVERFLOW;
                else {
                  if (prev_idx < log_n) truncate_to(prev_idx, prev_term);  // truncate(prev_log_index), then append
                  const uint2 *src = reinterpret_cast<const uint2 *>(arenas + (size_t)qsrc * R4_ARENA_WORDS + (qa & 0xFFFu) * 4 + 4);
                  for (u32 k = 0; k < cnt; k += 4) {
                    uint2 e0 = make_uint2(de0.x, de0.y), e1 = make_uint2(de0.z, de0.w), e2 = make_uint2(de1.x, de1.y), e3 = make_uint2(de1.z, de1.w);
                    if (k != 0) {
                      e0 = src[k];
                      if (k + 1 < cnt) e1 = src[k + 1];
                      if (k + 2 < cnt) e2 = src[k + 2];
                      if (k + 3 < cnt) e3 = src[k + 3];
                    }
                    append_entry(e0);
                    if (k + 1 < cnt) append_entry(e1);
                    if (k + 2 < cnt) append_entry(e2);
                    if (k + 3 < cnt) append_entry(e3);
                  }
                  if (commit_index < lcommit) commit_index = min(lcommit, log_n);
                  ok = 1;
                }
              }
            }
            rep = true; rep_dest = qsrc; rep_type = M_APPEND_ENTRIES_RES;
            rep_a = term | (ok << 16) | ((r_term == term ? 1u : 0u) << 17); rep_b = (prev_idx + cnt) & 0xFFFFFFu;
          } break;
          case M_APPEND_ENTRIES_RES: {  // handler closure of replicate_log, raft.py:395-410
            const u32 r_term = qa & 0xFFFFu, ok = (qa >> 16) & 1u, same = (qa >> 17) & 1u, pe = qb;
            maybe_step_down(r_term);
            if (role == R_LEADER && same && r_term == term) {   // <=> the term captured at send == our term now
              step_down_deadline = T + 2000000u;
              if (ok) { my_nim[qsrc * 2] = max(my_nim[qsrc * 2], pe + 1); my_nim[qsrc * 2 + 1] = max(my_nim[qsrc * 2 + 1], pe); }
              else if (my_nim[qsrc * 2] > 1) my_nim[qsrc * 2] -= 1;
            }
          } break;
          case M_READ: case M_WRITE: case M_CAS: {  // kv_req, raft.py:534-553
            if (role == R_LEADER) {
              if (log_n >= log_cap) my_flags |= MSIM_FLAG_ROWS_OVERFLOW;
              else {
                const u32 v1 = (qa >> 8) & 0xFFu, v2 = (qa >> 16) & 0xFFu;
                append_entry(make_uint2(term | (qtype << 16) | ((qa & 0xFFu) << 24), (qb & 0xFFFFu) | (qsrc << 16) | ((v1 & 0xFu) << 24) | ((v2 & 0xFu) << 28)));
              }
            } else if (leader >= 0) { rep = true; rep_dest = (u32)leader; rep_type = qtype; rep_a = qa; rep_b = qb; rep_src = qsrc; }  // proxy, client's src
            else { rep = true; rep_dest = qsrc; rep_type = M_ERROR; rep_a = 11; rep_b = qb; }
          } break;
          default: break;
        }
      }
      R4_MARK(3)
      if (act_now) {
        bool done = false;
        if (role == R_LEADER && step_down_deadline < T) { become_follower(); done = true; }              // :371-377
        if (!done && role == R_LEADER && T - last_replication > 50000u) {                                 // replicate_log :379-424
          const u32 elapsed = T - last_replication;
          u32 last_ni = INF, last_ref = 0;
#pragma unroll 1
          for (u32 i = 0; i < N; i++) if (i != l) {
            const u32 ni = my_nim[i * 2];
            const u32 cnt = ni <= log_n ? log_n - ni + 1 : 0;
            if (cnt > 0 || elapsed > 1000000u) {
              if (ni != last_ni) {   // one body per distinct next_index
                const u32 off = arena_alloc(4 + cnt * 2);
                u32 *d = my_arena + off;
                *reinterpret_cast<uint4 *>(d) = make_uint4(term | (log_term(ni - 1) << 16), (ni - 1) | (cnt << 16), commit_index | (arena_seq << 24), 0);
                if (cnt <= tail_valid) {   // entries ni .. log_n are the last cnt <= 4
                  if (cnt > 0) *reinterpret_cast<uint2 *>(d + 4) = TAIL_ENTRY(cnt - 1u);
                  if (cnt > 1) *reinterpret_cast<uint2 *>(d + 6) = TAIL_ENTRY(cnt - 2u);
                  if (cnt > 2) *reinterpret_cast<uint2 *>(d + 8) = TAIL_ENTRY(cnt - 3u);
                  if (cnt > 3) *reinterpret_cast<uint2 *>(d + 10) = TAIL_ENTRY(0u);
                } else bulk_mask |= 1u << i;
                last_ni = ni; last_ref = (off >> 2) | (arena_seq << 16);
              }
              aeref[l * N + i] = last_ref;
              fan_mask |= 1u << i;
            }
          }
          if (fan_mask) { fan_type = M_APPEND_ENTRIES; last_replication = T; done = true; }
        }
        if (!done && election_deadline < T) {                                                             // election :360-369
          if (role == R_FOLLOWER || role == R_CANDIDATE) {  // become_candidate :312-321 + request_votes :274-299
            role = R_CANDIDATE; term += 1; voted_for = (int)l; leader = -1;
            reset_election_deadline(); step_down_deadline = T + 2000000u;
            votes = 1u << l;
            fan_mask = all_nodes & ~(1u << l); fan_type = M_REQUEST_VOTE; fan_a = term | (last_term << 16); fan_b = log_n;
          } else reset_election_deadline();
          done = true;
        }
        if (!done && role == R_LEADER && can_commit) { commit_index = med; done = true; }                   // advance_commit_index :379-387
        if (!done && last_applied < commit_index) {                                                       // advance_state_machine :343-354
          last_applied += 1;
          const u32 back = log_n - last_applied;
          uint2 en;
          if (back < tail_valid) en = TAIL_ENTRY(back);
          else en = *reinterpret_cast<const uint2 *>(my_log + (last_applied - 1) * 2);
          const u32 et = (en.x >> 16) & 0xFFu, ek = en.x >> 24, v1 = (en.y >> 24) & 0xFu, v2 = en.y >> 28;
          const u32 cur = kv_get(ek);
          u32 rt, ra = 0;
          if (et == M_READ) { if (cur != 0xFu) { rt = M_READ_OK; ra = cur; } else { rt = M_ERROR; ra = 20; } }
          else if (et == M_WRITE) { kv_put(ek, v1); rt = M_WRITE_OK; }
          else { if (cur == 0xFu) { rt = M_ERROR; ra = 20; } else if (cur != v1) { rt = M_ERROR; ra = 22; } else { kv_put(ek, v2); rt = M_CAS_OK; } }
          if (role == R_LEADER) { rep = true; rep_dest = (en.y >> 16) & 0xFFu; rep_type = rt; rep_a = ra; rep_b = en.y & 0xFFFFu; }
        }
      }
      R4_MARK(4)
      if (take || act_now) {
        // when does this node's loop have something to do again, besides messages? (oracle: raft_next_time)
        if (role == R_LEADER) leader_cache(); else { can_commit = false; pending = false; }
        if (role == R_NASCENT) act_time = INF;
        else {
          u32 t = election_deadline + 1;
          if (role == R_LEADER) {
            t = min(t, step_down_deadline + 1);
            t = min(t, last_replication + (pending ? 50001u : 1000001u));
            if (can_commit) t = 0;
          }
          if (last_applied < commit_index) t = 0;
          act_time = t;
        }
      }

      R4_MARK(5)
      // COMMIT node sends: ids in node order; a node emits either a fan-out or one single message per round
      {
        const u32 fan_cnt = __popc(fan_mask);
        const u32 cnt = fan_cnt + (rep ? 1u : 0u);
        if (__ballot(cnt != 0)) {
          wave_lds_fence();  // aeref[] written above is read by other lanes
          // backlogs: the cluster's lanes copy entries next_index .. log_n of the leader's log into the body
          u32 bl = GB(bulk_mask != 0);
          while (__ballot(bl != 0)) {
            const bool on = bl != 0;
            const u32 s = on ? (u32)__builtin_ctz(bl) : 0u; bl &= bl - 1u;
            u32 bm = GGET(bulk_mask, s); const u32 ln = GGET(log_n, s);
            if (!on) bm = 0;
            while (__ballot(bm != 0)) {
              const bool on2 = bm != 0;
              const u32 i = on2 ? (u32)__builtin_ctz(bm) : 0u; bm &= bm - 1u;
              const u32 ni = nim_g[(s * N + i) * 2], ref = aeref[s * N + i];
              const u32 n_ent = on2 ? ln - ni + 1 : 0u;
              const uint2 *src = reinterpret_cast<const uint2 *>(g_scr + (size_t)s * log_cap * 2) + (ni - 1);
              uint2 *dst = reinterpret_cast<uint2 *>(arenas + (size_t)s * R4_ARENA_WORDS + (ref & 0xFFFu) * 4 + 4);
              for (u32 k = l; __ballot(k < n_ent); k += 64u) {
                uint2 c0 = make_uint2(0, 0), c1 = c0, c2 = c0, c3 = c0;
                if (k < n_ent) c0 = src[k];
                if (k + 16u < n_ent) c1 = src[k + 16u];
                if (k + 32u < n_ent) c2 = src[k + 32u];
                if (k + 48u < n_ent) c3 = src[k + 48u];
                if (k < n_ent) dst[k] = c0;
                if (k + 16u < n_ent) dst[k + 16u] = c1;
                if (k + 32u < n_ent) dst[k + 32u] = c2;
                if (k + 48u < n_ent) dst[k + 48u] = c3;
              }
            }
          }
          const u32 incl = row_scan(cnt);
          const u32 excl = incl - cnt;
          const u32 total = GGET(incl, GS - 1u);
          if (is_node) {
            s_send_sv += fan_cnt;
            if (rep) { if (rep_dest >= N || rep_src >= N) s_send_cl++; else s_send_sv++; }
          }
          u32 fans = GB(fan_mask != 0);
          while (__ballot(fans != 0)) {
            const bool on = fans != 0;
            const u32 s = on ? (u32)__builtin_ctz(fans) : 0u; fans &= fans - 1u;
            const u32 f_mask = GGET(fan_mask, s), f_type = GGET(fan_type, s), f_a = GGET(fan_a, s), f_b = GGET(fan_b, s);
            const u32 base = next_id + GGET(excl, s);
            if (on && l < N && ((f_mask >> l) & 1)) {
              const u32 rank = __popc(f_mask & lt);
              const u32 a = f_type == M_APPEND_ENTRIES ? aeref[s * N + l] : f_a;
              arrive(base + rank, f_type, a, f_b, s);
            }
          }
          u32 reps = GB(rep);
          const u32 rep_pack = rep_dest | (rep_type << 8) | (rep_src << 16);
          while (__ballot(reps != 0)) {
            const bool on = reps != 0;
            const u32 s = on ? (u32)__builtin_ctz(reps) : 0u; reps &= reps - 1u;
            const u32 pk = GGET(rep_pack, s), o = GGET(excl, s);
            const u32 r_a = GGET(rep_a, s), r_b = GGET(rep_b, s);
            if (on && l == (pk & 0xFF)) arrive(next_id + o, (pk >> 8) & 0xFF, r_a, r_b, pk >> 16);
          }
          next_id += total;
        }
        poll();
      }


      R4_MARK(6)
      // ---- R4: clients' recv! loops ----
      for (;;) {
        const bool dl = normal && is_client && has_c && deliver_at <= T;
        if (!__ballot(dl)) break;
        if (dl) {
          const uint4 q = cm; has_c = false;
          s_recv_cl++;
          const u32 qb = q.w & 0xFFFFFFu, qtype = q.y & 0x7Fu, qa = q.z;
          if (busy && qb == want) {  // else stale (client.clj:105-107)
            if (qtype == M_READ_OK) complete(MSIM_T_OK, 0, (c_value & 0xFFu) | ((qa & 0xFFu) << 8) | 0xFF0000u);   // [k v], lin_kv.clj:56-61
            else if (qtype == M_ERROR) complete(MSIM_T_FAIL, qa == 11 ? MSIM_ERR_TEMPORARILY_UNAVAILABLE : qa == 20 ? MSIM_ERR_KEY_DOES_NOT_EXIST : MSIM_ERR_PRECONDITION_FAILED, c_value);
            else complete(MSIM_T_OK, 0, c_value);
          }
          poll();
        }
      }
    }

    R4_MARK(7)
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
        if (NEM && wr && nem_rows && l == 0) {
          const u32 pk = MSIM_T_INFO | (nem_f << 2) | (MSIM_PROCESS_NEMESIS << 12);
          stage[n_rows % R4_STAGE] = make_uint4(tlo, thi, pk, nem_v1);
          stage[(n_rows + 1) % R4_STAGE] = make_uint4(tlo, thi | (nem_len2 << 16), pk, nem_v2);
        }
        if (wr && inv_row) stage[(n_rows + nem_rows + __popc(imask & lt)) % R4_STAGE] = make_uint4(tlo, thi, inv_packed, inv_value);
        if (wr && cmp_row) stage[(n_rows + nem_rows + ni + __popc(cmask & lt)) % R4_STAGE] = make_uint4(tlo, thi, cmp_packed, cmp_value);
        const u32 new_n = wr ? n_rows + nr : n_rows;
        const bool flush = (new_n >> 5) != (n_rows >> 5);   // a 32-row block completed (at most one per round: nr <= 22)
        if (__ballot(flush)) {
          wave_lds_fence();
          if (flush) {
            const u32 g0 = (n_rows >> 5) * 32u + l;
            if (g0 < max_rows) reinterpret_cast<uint4 *>(g_rows)[g0] = stage[g0 % R4_STAGE];
            if (g0 + 16u < max_rows) reinterpret_cast<uint4 *>(g_rows)[g0 + 16u] = stage[(g0 + 16u) % R4_STAGE];
          }
          wave_lds_fence();
        }
        n_rows = new_n;
      }
    }
    R4_MARK(8)
  }

  __syncthreads();
  {
    const u32 g0 = (n_rows >> 5) * 32u + l;
    if (real && g0 < n_rows) reinterpret_cast<uint4 *>(g_rows)[g0] = stage[g0 % R4_STAGE];
    if (real && g0 + 16u < n_rows) reinterpret_cast<uint4 *>(g_rows)[g0 + 16u] = stage[(g0 + 16u) % R4_STAGE];
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
#ifdef R4_PROF   // developer build (tools/variant_lib.sh r4prof raft4.hip -DR4_PROF): cycle counters of the round's sections, 4 per cluster of the wavefront
    {
      u32 v[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 9; i++) v[i] = (u32)(pacc[i] >> 6);
      v[9] = wave_rounds;
      u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (u32 g = 0; g < 4; g++) if (grp == g) { a0 = v[4 * g]; a1 = v[4 * g + 1]; a2 = v[4 * g + 2]; a3 = v[4 * g + 3]; }
      m.n_events = a0; m.reserved[0] = a1; m.reserved[1] = a2; m.reserved[2] = a3;
    }
#endif
    p.meta[inst] = m;
  }
}

}  // namespace

// Whether four clusters per wavefront simulate this configuration (see the header of this file).
bool msim_raft4_eligible(const msim_config &c) {
  if (c.node_program != MSIM_NODE_RAFT || c.journal_capacity != 0) return false;
  const uint32_t cs = c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes;
  return c.n_nodes >= 1 && c.n_nodes + cs <= GS;
}

// Extra per-instance scratch words the layout needs behind the node spill area: the clients' spill and the part of the nodes'
// LDS inbox of raft_kernel<> that does not fit this kernel's RQ slots.
uint64_t msim_raft4_extra_scratch_words(const msim_config &c) {
  const uint32_t cs = c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes;
  return ((uint64_t)c.n_nodes * c.inbox_capacity + (uint64_t)cs * R4_CLIENT_CAP) * 4;
}

template <int NN>
static void raft4_launch(const R4Params &rp, bool nem, bool rnd, dim3 grid, size_t lds, hipStream_t st) {
  const dim3 block(64);
  if (nem) { if (rnd) hipLaunchKernelGGL((raft4_kernel<true, true, NN>), grid, block, lds, st, rp); else hipLaunchKernelGGL((raft4_kernel<true, false, NN>), grid, block, lds, st, rp); }
  else { if (rnd) hipLaunchKernelGGL((raft4_kernel<false, true, NN>), grid, block, lds, st, rp); else hipLaunchKernelGGL((raft4_kernel<false, false, NN>), grid, block, lds, st, rp); }
}

hipError_t msim_launch_raft4(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (kp.raft_log_cap > 0xFFFFu) return MSIM_LAYOUT_DOES_NOT_FIT;   // log indices travel in 16 bits
  R4Params rp;
  rp.k = kp; rp.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  rp.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;             // <= spill_capacity + inbox_capacity entries per node
  rp.client_spill = R4_CLIENT_CAP > RQ ? R4_CLIENT_CAP - RQ : 0;
  rp.client_spill_off = kp.spill_off + (uint64_t)kp.N * rp.node_spill * 4;
  rp.arenas_off = (kp.N * kp.raft_log_cap * 2 + 3u) & ~3u;     // the slack is there: the protocol scratch is rounded up to 16 bytes
  rp.nim_stride = (kp.N * kp.N * 3 + 1u) & ~1u;
  size_t off = (size_t)RQ * 64 * 16;
  rp.off_kv = (u32)off; off += (size_t)4 * kp.N * 128;
  rp.off_nim = (u32)off; off += (size_t)4 * rp.nim_stride * 4;
  rp.off_runs = (u32)off; off += (size_t)RK * 64 * 4;
  off = (off + 15) & ~(size_t)15;
  rp.off_stage = (u32)off; off += (size_t)4 * R4_STAGE * 16;
  rp.off_misc = (u32)off; off += 64 * 4;
  rp.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  if (lds > 64 * 1024) return MSIM_LAYOUT_DOES_NOT_FIT;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(r4_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 3) / 4);
  if (kp.N <= 5) raft4_launch<5>(rp, c.nemesis_mask != 0, rnd, grid, lds, st);
  else raft4_launch<8>(rp, c.nemesis_mask != 0, rnd, grid, lds, st);
  return hipGetLastError();
}
