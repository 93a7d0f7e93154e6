// uid8.hip — EIGHT echo / unique-ids clusters per wavefront (the reference's demos `echo.rb` and
// `flake_ids.clj`, core.clj:104-106,122-126 — unique-ids with partitions at rate 1000).
//
// Same programs and the same rounds as sim_kernel_colo<MSIM_NODE_ECHO / MSIM_NODE_FLAKE_IDS, ...> (sim_kernel_colo.inc): node = echo
// (doc/02-echo: echo -> echo_ok with the same value) or flake_ids.clj:16-31 (generate -> [max(now in s, last time), counter within that
// second, node]); client = workload/echo.clj:40-60 / unique_ids.clj:45-65 (Reusable for unique-ids); generator = echo.clj:62-66 (a value
// below 128) / unique_ids.clj:72 (gen/repeat {:f :generate}) — round for round what DESIGN.md §2 and the CPU oracle specify.  In these two
// programs a node talks to its clients only: a cluster of 3 nodes used 3 lanes of a wavefront's 64 in the colocated kernel and paid its
// whole instruction stream.  Here a cluster is a group of 8 lanes (lane l = node l + its client) and a wavefront carries eight clusters
// (txn8.hip's scheme: what is uniform per cluster lives in VGPRs, ballots are the group's slice, `ds_bpermute` within the group).
//
// Scope (engine.hip picks this kernel when all of it holds, else the colocated kernel runs): at most 8 nodes, one worker per node, net
// journal off, at least 4096 clusters in the launch (eight per wavefront are an eighth of the wavefronts, and a wavefront's run is a
// chain of dependent steps that only other wavefronts hide — measured, unique-ids at 3 nodes, rate 1000, partitions: 4096 clusters 18.9
// ms against 38.8 one per wavefront, 16384: 22 / 137, 65536: 66 / 517, profiles/r03am_uid8.txt; MSIM_DEV_FLAGS bit 10 asks for the
// layout whatever the batch).
//
// LDS of a wavefront (slot-major: slot s of lane e at [s * 64 + e]): node queues (RQ envelopes, the rest spills to HBM), client inboxes
// (2 envelopes: the colocated kernel's CLIENT_INBOX_CAP), the nemesis shuffle.  History rows go straight to HBM.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "group8.h"
#include "layout_thresholds.h"

namespace {

constexpr u32 GS = 8u;            // lanes per cluster
constexpr u32 RQ = 4u;            // LDS envelopes per node queue
constexpr u32 CQ = CLIENT_INBOX_CAP;   // envelopes per client inbox (all of them in LDS)

struct U8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_misc;                                  // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;                          // HBM spill entries per node queue / client inbox (clients: none)
  u64 client_spill_off;
  u32 round_limit;
};


template <bool FLAKE, bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) uid8_kernel(const U8Params up) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr u32 GM = (1u << GS) - 1u, NG = 64u / GS;
  const KParams &p = up.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane / GS, gbase = lane & ~(GS - 1u);
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
  const u32 rate = p.cfg.rate_mhz;
  const u32 round_limit = up.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  const u32 qlane = is_node ? l : 0u;
  uint4 *const my_spill = reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)qlane * up.node_spill;
  uint4 *const my_cspill = reinterpret_cast<uint4 *>(g_scr + up.client_spill_off);   // (never used: client_spill = 0)
  const u32 my_spill_cap = is_node ? up.node_spill : 0u;

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + up.off_cq) + lane;                      // client inbox
  u32 *const misc = reinterpret_cast<u32 *>(smem + up.off_misc) + grp * GS;

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & GM; };              // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 flake_time = 0, flake_count = 0;  // flake_ids.clj:10-14
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 timeout_at = 0, next_msg_id = 0, c_f = 0, c_value = 0, process = l, m_f = 0, m_value = 0, cin_n = 0, csp_n = 0;
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, nem_next = 0, nem_j = 0;
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
          if (phase == PH_DRAIN && !busy_mask) { phase = PH_DONE; ch = true; }   // no final phase (echo.clj, unique_ids.clj:72-78)
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
        u32 km = g8_min<8>(k);
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
      cmp_row = true; cmp_packed = type | (c_f << 2) | (err << 7) | (process << 12);
      cmp_value = value;
      if (type == MSIM_T_INFO) { process += N; if (!FLAKE) { next_msg_id = 0; cin_n = 0; } }  // crashed process; fresh client unless Reusable
    };
    // the client's recv! consumes one envelope (client.clj:94-107)
    auto client_deliver = [&](u32 qtype, u32 qa, u32 qb) {
      s_recv_cl++;
      if (busy && qb == next_msg_id) {
        if (qtype == M_ECHO_OK || qtype == M_GENERATE_OK) complete(MSIM_T_OK, 0, qa);
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
          const u64 h = draw64(key, S_GEN, gen_k);
          const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          const u32 pick = scale32(r_lo, nfree);
          const bool sel = gen_on && is_node && !busy && (u32)__popc(free_mask & lt) == pick;
          if (sel) { mark = true; kind = K_OP; m_f = FLAKE ? (u32)MSIM_F_GENERATE : (u32)MSIM_F_ECHO; m_value = FLAKE ? MSIM_NO_VALUE : ((r_lo >> 4) & 127u); }
          if (gen_on) { gen_k++; gen_next = T + __umulhi(r_hi, p.gen_period2_us); }
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
            c_f = m_f; c_value = m_value;
            inv_row = true; inv_packed = MSIM_T_INVOKE | (c_f << 2) | (process << 12); inv_value = c_value;
            rq_type = FLAKE ? (u32)M_GENERATE : (u32)M_ECHO; rq_a = FLAKE ? 0u : c_value;
          }
          const u32 want = ++next_msg_id;
          timeout_at = T + (kind == K_OP ? p.cfg.client_timeout_ms : 10000u) * 1000u;
          s_send_cl++;
          arrive(next_id + __popc(inv_mask & lt), rq_type, rq_a, want, N + l);
        }
        next_id += __popc(inv_mask);
        poll();
      }

      // ---- R3: one input per node; its reply goes to the client that asked (this lane's own) ----
      bool rep = false; u32 o_type = 0, o_a = 0, o_b = 0;
      const bool msg = normal && is_node && deliver_at <= T;
      if (msg) {
        const uint4 q = cm; deliver_at = INF;
        const u32 qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if ((q.w >> 24) >= N) s_recv_cl++; else s_recv_sv++;
        rep = true; o_b = qb;
        if (qtype == M_INIT) o_type = M_INIT_OK;
        else if (qtype == M_ECHO) { o_type = M_ECHO_OK; o_a = qa; }
        else if (qtype == M_GENERATE) {  // flake_ids.clj:16-31: [max(now in s, last time), counter within that second, node]
          u32 t = T / 1000000u;
          if (t < flake_time) t = flake_time;
          flake_count = t == flake_time ? flake_count + 1 : 0u; flake_time = t;
          o_type = M_GENERATE_OK; o_a = (t << 20) | ((flake_count & 0x7FFFu) << 5) | l;
        } else rep = false;
      }

      // COMMIT: ids in node order; node -> its own client: no latency; lost like any other message (net.clj:214)
      bool c_arr = false; u32 ca_y = 0, ca_a = 0, ca_b = 0;
      {
        const u32 rmask = GB(rep);
        if (__ballot(rep)) {
          if (rep) {
            s_send_cl++;
            const u32 id = next_id + __popc(rmask & lt);
            if (!(NET_RANDOM && loss_on && p_loss && draw32(key, S_LOSS, id) < p_loss)) { c_arr = true; ca_y = (id << 8) | o_type; ca_a = o_a; ca_b = o_b; }
          }
          next_id += __popc(rmask);
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
        if (wr && cmp_row) out[nem_rows + ni + __popc(cmask & lt)] = make_uint4(tlo, thi, cmp_packed, cmp_value);
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


template <bool FLAKE>
hipError_t u8_launch(const U8Params &up, uint32_t n, size_t lds, bool nem, bool rnd, hipStream_t st) {
  const dim3 grid((n + 7) / 8), block(64);
  if (nem) { if (rnd) hipLaunchKernelGGL((uid8_kernel<FLAKE, true, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((uid8_kernel<FLAKE, true, false>), grid, block, lds, st, up); }
  else { if (rnd) hipLaunchKernelGGL((uid8_kernel<FLAKE, false, true>), grid, block, lds, st, up); else hipLaunchKernelGGL((uid8_kernel<FLAKE, false, false>), grid, block, lds, st, up); }
  return hipGetLastError();
}

}  // namespace

// Whether eight clusters per wavefront simulate this configuration (see the header of this file).
bool msim_uid8_eligible(const msim_config &c) {
  return (c.node_program == MSIM_NODE_ECHO || c.node_program == MSIM_NODE_FLAKE_IDS) && c.journal_capacity == 0 && c.n_nodes >= 1 && c.n_nodes <= GS &&
         c.concurrency == c.n_nodes;
}

// Extra per-instance scratch words behind the queues' spill area: what of the LDS queues of the colocated kernel does not fit this
// kernel's RQ slots.
uint64_t msim_uid8_extra_scratch_words(const msim_config &c) { return (uint64_t)c.n_nodes * c.inbox_capacity * 4; }

hipError_t msim_launch_uid8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < MSIM_UID8_MIN_CLUSTERS && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;   // (see the header)
  U8Params up;
  up.k = kp; up.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  up.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;
  up.client_spill = 0;
  up.client_spill_off = kp.spill_off;
  size_t off = (size_t)RQ * 64 * 16;
  up.off_cq = (u32)off; off += (size_t)CQ * 64 * 16;
  up.off_misc = (u32)off; off += 64 * 4;
  up.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  if (kp.dev_flags & 0x1000u) std::fprintf(stderr, "[uid8] %u clusters, eight per wavefront, %zu B of LDS per wavefront\n", n, lds);   // developer trace bit
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(g8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  return c.node_program == MSIM_NODE_FLAKE_IDS ? u8_launch<true>(up, n, lds, c.nemesis_mask != 0, rnd, st) : u8_launch<false>(up, n, lds, c.nemesis_mask != 0, rnd, st);
}
