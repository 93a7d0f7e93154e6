// svc4.hip — FOUR lin-kv-proxy clusters per wavefront: the lin-kv workload over demo/ruby/lin_kv_proxy.rb (the reference's own demo
// invocation, core.clj:112: 5 nodes, concurrency 10) with the key-value service on a lane of its own, in 16-lane groups.
//
// Same program and the same rounds as svc_kernel<> (sim_kernel_svc.inc; specification: oracle/svc_nodes.inc): node = lin_kv_proxy.rb:1-49
// (every request forwarded to the service under a fresh rpc id, the reply handed back to the client whose callback that id holds), services
// = lin-kv (service.clj:31-61,141-155) and lww-kv (service.clj:214-243 over :65-114, two replicas, the merge computed and discarded), clients
// = client.clj:41-172 / lin_kv.clj:40-85, generator = [upstream] jepsen.tests.linearizable-register.  What changes is the mapping, as in
// raft4.hip (whose time / scheduler / client / history-row machinery this file shares line for line): 5 nodes + 10 client slots + the
// service are 16 endpoints — one per lane of a 16-lane group — and a wavefront carries four clusters.  svc_kernel<> runs one cluster per
// wavefront (16 live lanes of 64) and is bound by instruction issue; here one instruction stream serves four clusters.
//
// Scope (engine.hip picks this kernel when all of it holds, else svc_kernel<> runs): the proxy over lin-kv or lww-kv (seq-kv's ring of 32
// states is 8 KiB per cluster: svc_kernel<>) or unique-ids over the lin-tso node (TSO: the service lane is the timestamp oracle), n_nodes + max(concurrency, n_nodes) + 1 <= 16, net journal off, at least
// MSIM_SVC4_MIN_CLUSTERS clusters in the launch.
//
// LDS of a wavefront: envelope queues slot-major (slot s of lane e at [s * 64 + e]; RQ envelopes, the rest spills to HBM: servers
// inbox_capacity + spill_capacity in all, clients 32, the oracle's limits), per node the 32 newest callbacks {rpc id | client, client msg id},
// per cluster the service's states (256 B each: lin-kv, or lww-kv's two replicas) and the nemesis shuffle — 8.3 KiB for the demo shape, and a
// register budget of four wavefronts per SIMD: 16384 clusters are one pass of the chip (measured: RQ 4 + a row staging ring, two wavefronts
// per SIMD: 41 ms per 16384; RQ 2, rows straight to HBM, four: 25 ms; profiles/r06f_svc4_variants.jsonl).  History rows go straight to HBM.
//
// Envelope (16 B): x = deadline, y = (id << 8) | type, z = a, w = b | (src << 24); src = the sender's lane in its group (the service: n + slots).
#include <hip/hip_runtime.h>

#include "wave_common.h"
#include "log2_table.h"
#include "layout_thresholds.h"

namespace {

__constant__ u32 s4_log2_q24[257];

constexpr u32 GS = 16u;           // lanes per cluster
#ifndef S4_RQ
#define S4_RQ 2u
#endif
#ifndef S4_WAVES
#define S4_WAVES 4
#endif
constexpr u32 RQ = S4_RQ;         // LDS envelopes per endpoint
constexpr u32 S4_CLIENT_CAP = 32u;   // Reusable lin-kv clients (lin_kv.clj:74-76) collect late replies between RPCs (the oracle's limit)
constexpr u32 S4_SLOTS = 32u;     // callbacks per node (PX_SLOTS of sim_kernel_svc.inc / oracle/svc_nodes.inc)
enum { M_WRITE = 14, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR };
enum { S_SVC = 12 };
enum { M_TS = 28, M_TS_OK = 29 };   // include/maelsim.h MSIM_M_TS*

struct S4Params {
  KParams k;
  u32 n_inst;
  u32 off_cbs, off_kvs, off_misc;   // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;               // HBM spill entries per server endpoint / client behind the RQ LDS slots
  u64 client_spill_off;                       // word offset of the clients' spill area inside the per-instance scratch
  u32 round_limit;
};

__device__ __forceinline__ u32 s4_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = s4_log2_q24[idx], l1 = s4_log2_q24[idx + 1];
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

template <bool NEM, bool NET_RANDOM, bool TSO>   // TSO: unique-ids over the lin-tso timestamp oracle (the node asks the service for a ts per generate)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S4_WAVES))) svc4_kernel(const S4Params rp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = rp.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 4, gbase = lane & 48u;
  const u32 N = p.N, C = p.C, CS = p.CS;
  const u32 SVC = N + CS;
  const bool is_node = l < N;
  const bool is_client = l >= N && l < N + CS;
  const bool is_svc = l == SVC;
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
  const u32 rate = p.cfg.rate_mhz;
  const bool lww = p.cfg.proxy_service == MSIM_SVC_LWW_KV;
  u32 rpc_timeout_ms = 10 * lat_mean; if (rpc_timeout_ms < 1000) rpc_timeout_ms = 1000;   // lin_kv.clj:54
  if (TSO) rpc_timeout_ms = p.cfg.client_timeout_ms;                                       // client.clj:18-20
  const u32 round_limit = rp.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  u32 *const g_scr = p.scratch + (size_t)inst * p.scratch_words;
  const u32 my_spill_cap = is_server ? rp.node_spill : (is_client ? rp.client_spill : 0u);
  uint4 *const my_spill = is_server ? reinterpret_cast<uint4 *>(g_scr + p.spill_off) + (size_t)(is_node ? l : N) * rp.node_spill
                                    : reinterpret_cast<uint4 *>(g_scr + rp.client_spill_off) + (size_t)(is_client ? slot : 0) * rp.client_spill;

  // LDS
  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                         // slot s at my_q[s * 64]
  uint2 *const my_cbs = reinterpret_cast<uint2 *>(smem + rp.off_cbs) + (grp * N + (is_node ? l : 0)) * S4_SLOTS;   // {rpc_id | client << 24, client_msg | used << 31}
  unsigned char *const kvs_g = smem + rp.off_kvs + grp * (lww ? 512u : 256u);                                        // the service's states: lin-kv, or lww-kv's replicas 0 and 1
  u32 *const misc = reinterpret_cast<u32 *>(smem + rp.off_misc) + grp * GS;

  for (u32 i = lane; i < 4 * N * S4_SLOTS; i += 64) reinterpret_cast<uint2 *>(smem + rp.off_cbs)[i] = make_uint2(0, 0);
  for (u32 i = lane; i < (lww ? 4u * 128u : 4u * 64u); i += 64) reinterpret_cast<u32 *>(smem + rp.off_kvs)[i] = 0xFFFFFFFFu;
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- endpoint state ----
  bool has_c = false; u32 deliver_at = 0; uint4 cm = make_uint4(0, 0, 0, 0);
  bool have_pm = false; uint4 pm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 node_msgid = 0, svc_ctr = 0;   // node: rpc ids; service lane: rand-int draws so far (lin-tso: the next timestamp)
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
    if ((src < N || src == SVC) && is_server) {  // neither end is a client (util.clj:7-16)
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * s4_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
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
  // PersistentKV/handle on the 256-byte state m (service.clj:31-61)
  auto kv_handle = [&](unsigned char *m, u32 type, u32 a, u32 &rt, u32 &ra) {
    const u32 k = a & 0xFF, v1 = (a >> 8) & 0xFF, v2 = (a >> 16) & 0xFF;
    const u32 cur = m[k];
    ra = 0;
    if (type == M_READ) { if (cur == 0xFF) { rt = M_ERROR; ra = 20; } else { rt = M_READ_OK; ra = cur; } return; }
    if (type == M_WRITE) { m[k] = (unsigned char)v1; rt = M_WRITE_OK; return; }
    if (cur == 0xFF) { rt = M_ERROR; ra = 20; return; }
    if (cur != v1) { rt = M_ERROR; ra = 22; return; }
    m[k] = (unsigned char)v2; rt = M_CAS_OK;
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
          if (TSO) {   // (gen/repeat {:f :generate}), unique_ids.clj:71
            if (sel) { mark = true; kind = K_OP; m_f = MSIM_F_GENERATE; m_value = MSIM_NO_VALUE; }
            if (gen) { gen_k++; gen_next = T + __umulhi(r_hi, p.gen_period2_us); }
          } else {
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
      }

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
            rq_type = TSO ? (u32)M_GENERATE : c_f == MSIM_F_WRITE ? (u32)M_WRITE : c_f == MSIM_F_CAS ? (u32)M_CAS : (u32)M_READ;
            rq_a = TSO ? 0u : c_value;
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

      // ---- R3: one input per node (lin_kv_proxy.rb), then one for the service (endpoint order) ----
      bool rep = false; u32 rep_dest = 0, rep_type = 0, rep_a = 0, rep_b = 0;
      if (is_server && normal && has_c && deliver_at <= T) {
        const uint4 q = cm; has_c = false;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0x7Fu, qa = q.z;
        if (qsrc >= N && qsrc < SVC) s_recv_cl++; else s_recv_sv++;
        if (is_node) {
          if (qtype == M_INIT) { rep = true; rep_dest = qsrc; rep_type = M_INIT_OK; rep_b = qb; }
          else if (TSO ? qtype == M_GENERATE : (qtype == M_READ || qtype == M_WRITE || qtype == M_CAS)) {  // proxy!, lin_kv_proxy.rb:27-38 / generate -> ts
            const u32 rid = ++node_msgid;
            // engine capacity: the 32 newest callbacks per node (an evicted one is flagged only if its reply still arrives)
            my_cbs[rid % S4_SLOTS] = make_uint2((rid & 0xFFFFFFu) | (qsrc << 24), (qb & 0xFFFFFFu) | 0x80000000u);
            rep = true; rep_dest = SVC; rep_type = TSO ? (u32)M_TS : qtype; rep_a = TSO ? 0u : qa; rep_b = rid;
          } else if (TSO ? qtype == M_TS_OK : (qtype == M_READ_OK || qtype == M_WRITE_OK || qtype == M_CAS_OK || qtype == M_ERROR)) {
            const uint2 c = my_cbs[qb % S4_SLOTS];
            if ((c.y >> 31) && (c.x & 0xFFFFFFu) == qb) {
              my_cbs[qb % S4_SLOTS] = make_uint2(0, 0);
              rep = true; rep_dest = c.x >> 24; rep_type = TSO ? (u32)M_GENERATE_OK : qtype; rep_a = qa; rep_b = c.y & 0x7FFFFFFFu;
            } else if (qb + S4_SLOTS <= node_msgid) my_flags |= MSIM_FLAG_ARENA_OVERRUN;
          }
        } else {  // the service
          u32 rt = 0, ra = 0, ri = 0;
          if (TSO) { if (qtype == M_TS) { rep = true; rep_dest = qsrc; rep_type = M_TS_OK; rep_a = svc_ctr++; rep_b = qb; } }   // PersistentTSO, service.clj:116-123
          else {
          if (lww) {  // Eventual over LWWKV, 2 replicas; the merge is computed and discarded (service.clj:222-235)
            svc_ctr += 2;  // merge-source, merge-dest
            ri = scale32(draw32(key, S_SVC, svc_ctr++), 2);
          }
          kv_handle(kvs_g + ri * 256u, qtype, qa, rt, ra);
          rep = true; rep_dest = qsrc; rep_type = rt; rep_a = ra; rep_b = qb;
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
          const u32 qb = q.w & 0xFFFFFFu, qtype = q.y & 0x7Fu, qa = q.z;
          if (busy && qb == want) {  // else stale (client.clj:105-107)
            if (TSO && qtype == M_GENERATE_OK) complete(MSIM_T_OK, 0, qa);   // unique_ids.clj:53-57
            else if (qtype == M_READ_OK) complete(MSIM_T_OK, 0, (c_value & 0xFFu) | ((qa & 0xFFu) << 8) | 0xFF0000u);   // [k v], lin_kv.clj:56-61
            else if (qtype == M_ERROR) complete(MSIM_T_FAIL, qa == 11 ? MSIM_ERR_TEMPORARILY_UNAVAILABLE : qa == 20 ? MSIM_ERR_KEY_DOES_NOT_EXIST : MSIM_ERR_PRECONDITION_FAILED, c_value);
            else complete(MSIM_T_OK, 0, c_value);
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
        if (wr && inv_row) reinterpret_cast<uint4 *>(gr)[n_rows + nem_rows + __popc(imask & lt)] = make_uint4(tlo, thi, inv_packed, inv_value);
        if (wr && cmp_row) reinterpret_cast<uint4 *>(gr)[n_rows + nem_rows + ni + __popc(cmask & lt)] = make_uint4(tlo, thi, cmp_packed, cmp_value);
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
bool msim_svc4_eligible(const msim_config &c) {
  if (c.journal_capacity != 0) return false;
  if (c.node_program != MSIM_NODE_TSO_IDS && (c.node_program != MSIM_NODE_LIN_KV_PROXY || c.proxy_service == MSIM_SVC_SEQ_KV)) return false;
  const uint32_t cs = c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes;
  return c.n_nodes >= 1 && c.n_nodes + cs + 1 <= GS;
}

// Extra per-instance scratch words the layout needs behind svc_kernel<>'s spill areas: the clients' whole inboxes and the part of the
// servers' LDS inboxes of svc_kernel<> that does not fit this kernel's RQ slots.
uint64_t msim_svc4_extra_scratch_words(const msim_config &c) {
  const uint32_t cs = c.concurrency > c.n_nodes ? c.concurrency : c.n_nodes;
  return ((uint64_t)(c.n_nodes + 1) * c.inbox_capacity + (uint64_t)cs * S4_CLIENT_CAP) * 4;
}

template <bool TSO>
static void svc4_launch(const S4Params &rp, bool nem, bool rnd, dim3 grid, size_t lds, hipStream_t st) {
  const dim3 block(64);
  if (nem) { if (rnd) hipLaunchKernelGGL((svc4_kernel<true, true, TSO>), grid, block, lds, st, rp); else hipLaunchKernelGGL((svc4_kernel<true, false, TSO>), grid, block, lds, st, rp); }
  else { if (rnd) hipLaunchKernelGGL((svc4_kernel<false, true, TSO>), grid, block, lds, st, rp); else hipLaunchKernelGGL((svc4_kernel<false, false, TSO>), grid, block, lds, st, rp); }
}

hipError_t msim_launch_svc4(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < MSIM_SVC4_MIN_CLUSTERS && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;
  S4Params rp;
  rp.k = kp; rp.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  rp.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;             // <= spill_capacity + inbox_capacity entries per server endpoint
  rp.client_spill = S4_CLIENT_CAP > RQ ? S4_CLIENT_CAP - RQ : 0;
  rp.client_spill_off = kp.spill_off + (uint64_t)(kp.N + 1) * rp.node_spill * 4;
  size_t off = (size_t)RQ * 64 * 16;
  rp.off_cbs = (u32)off; off += (size_t)4 * kp.N * S4_SLOTS * 8;
  rp.off_kvs = (u32)off; off += (size_t)4 * (c.proxy_service == MSIM_SVC_LWW_KV ? 512 : 256);
  rp.off_misc = (u32)off; off += 64 * 4;
  rp.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(s4_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 3) / 4);
  if (c.node_program == MSIM_NODE_TSO_IDS) svc4_launch<true>(rp, c.nemesis_mask != 0, rnd, grid, lds, st);
  else svc4_launch<false>(rp, c.nemesis_mask != 0, rnd, grid, lds, st);
  return hipGetLastError();
}
