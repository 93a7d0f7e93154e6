// duo.hip — TWO clusters per wavefront: the headline layout of the simulation kernel.
//
// Same hot path as sim_kernel_colo<> (engine.hip): net.clj:189-247 (send!/recv!, per-destination queues ordered by
// (deadline, id), head-of-line blocking), process.clj:136-166 (one input per node per round), client.clj:41-172 (sync RPC
// clients), core.clj:67-80 (generator phases) and the fire-and-forget broadcast node (doc/03-broadcast/01-broadcast.md:
// 525-547, 02-performance.md:61-67 and :22-28) — round for round what DESIGN.md §2 and the CPU oracle specify.
//
// Why a second layout.  A 25-node cluster fills 25 of a wavefront's 64 lanes, and everything that is uniform per cluster
// (time, phase, generator, cursors) was scalar work paid once per cluster and round: the colocated kernel ran at the
// CU's scalar issue limit.  Here lanes 0-31 simulate one cluster and lanes 32-63 another, and what is
// uniform per CLUSTER lives in VGPRs (every lane of a half holds the same value): one instruction stream, vector
// instructions, serves both clusters; a "ballot" is the cluster's 32-bit half of the wave ballot.  Nothing crosses
// between the halves: the two clusters share the program counter and nothing else.
//
// Scope (the host picks this kernel when all of it holds, else the kernels of engine.hip run):
//   * node program broadcast fire-and-forget (with or without skip-sender), colocated clients (concurrency == n_nodes <= 32);
//   * constant, uniform or exponential latency (bounded below the RPC timeouts), no loss, no nemesis, net journal off.
// With constant latency message ids are unobservable (no per-message RNG draw, no journal, arrival order = id order) and a
// node's queue is a FIFO.  With random latency (RND) every message's latency is drawn from its id (net.clj:178-187), so ids are
// assigned in the canonical order (a prefix sum over the senders of a round) and a node's queue is a bag ordered by
// (deadline, arrival sequence) — the arrival sequence at one node orders its envelopes exactly as their ids do.  No client can
// time out (an RPC completes within one maximal latency of virtual time), a client's reply is handled by the lane that
// sent the request, and a round is:
//   R0  per-cluster time: stay at T while something is due, else jump to the next delivery / scheduler event;
//   R1  generator (one op) / phase actions                          — GENERAL rounds only
//   R2  marked clients invoke: the request reaches its own node     — GENERAL rounds only
//   R3  every node with a due envelope handles it (dedup, fan-out);
//       COMMIT: receivers PULL the senders' fan-outs with ds_bpermute (one per topology neighbour), append to their
//       own LDS ring; idle receivers poll (pop the ring head / the pending client request);
//   R4  completions -> history rows (staged in LDS, 1 KiB coalesced appends)   — GENERAL rounds only
// A wave-round is GENERAL if either cluster needs it; pure gossip rounds of both clusters take the short body.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "wave_common.h"
#include "log2_table.h"

namespace {

__constant__ u32 duo_log2_q24[257];

enum { DK_PLAIN = 0, DK_BCAST = 1, DK_READ = 2, DK_READ_FINAL = 3, DK_INIT = 4, DK_TOPO = 5 };  // kind of an envelope (bits 24-26)
constexpr u32 DUO_STAGE_ROWS = 128u;
#ifndef DUO_BAG_N
#define DUO_BAG_N 16
#endif
constexpr u32 DUO_BAG = DUO_BAG_N;   // random latencies: envelopes of a node's queue that live in LDS (a power of two, 4 .. 16; the rest spills to HBM behind a cached minimum)
// History rows go from their lanes to HBM unstaged (two 16-byte rows per operation; the L2 merges them into lines: WRITE_SIZE stays at the
// algorithmic bytes): 9.00 -> 8.90 ms per 4096 clusters against staging 64 rows in LDS (-DDUO_STAGED_ROWS keeps that variant for A/B runs).
#ifdef DUO_STAGED_ROWS
constexpr bool DUO_DIRECT = false;
#else
constexpr bool DUO_DIRECT = true;
#endif

struct DuoParams {
  KParams k;
  u32 n_inst;        // clusters in this launch (the last wavefront may hold one)
  u32 R;             // LDS ring entries per node (power of two >= inbox_capacity)
  u32 S;             // HBM spill entries per node behind the ring (R + S = inbox_capacity + spill_capacity)
  u32 half_bytes;    // LDS bytes per cluster
  u32 off_ring, off_seen;  // byte offsets inside a cluster's LDS region
  u32 deg;           // maximum degree of the topology
  u32 echoback;      // node program without skip-sender
  u32 round_limit;
  u32 off_seq;       // RND: byte offset of the arrival-sequence array (u16 per ring entry) inside a cluster's LDS region
  u32 off_log2;      // RND: byte offset of the Q24 log2 table (one per wavefront, behind both clusters)
};

// -ln(u), u = (r+1)/2^32, Q16, integer only: the sampler of engine.hip / the oracle over a copy of the table in LDS
__device__ __forceinline__ u32 duo_neg_ln_q16(u32 r, const u32 *tab) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = tab[idx], l1 = tab[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}

// the cluster's half of a wave ballot
__device__ __forceinline__ u32 hb(bool pred, bool hi) {
  const u64 b = __ballot(pred);
  return hi ? (u32)(b >> 32) : (u32)b;
}
// true in every lane of a half iff pred holds in some lane of that half; scalar work only (the result is a lane mask in SGPRs)
__device__ __forceinline__ bool half_any(bool pred) {
  const u64 b = __ballot(pred);
  const u32 lo = (u32)b ? 0xFFFFFFFFu : 0u, up = (u32)(b >> 32) ? 0xFFFFFFFFu : 0u;
  return __builtin_amdgcn_inverse_ballot_w64(((u64)up << 32) | lo);
}
// min over the 32 lanes of the caller's half (result uniform per half)
__device__ __forceinline__ u32 half_min(u32 v, bool hi) {
  v = min(v, dpp_mov<0xB1, 0xF, 0xF, false>(v, v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_mov<0x4E, 0xF, 0xF, false>(v, v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_mov<0x141, 0xF, 0xF, false>(v, v));  // row_half_mirror
  v = min(v, dpp_mov<0x140, 0xF, 0xF, false>(v, v));  // row_mirror
  const u32 lo = min(rdlane(v, 0), rdlane(v, 16)), up = min(rdlane(v, 32), rdlane(v, 48));
  return hi ? up : lo;
}
__device__ __forceinline__ u32 bperm(u32 byte_addr, u32 v) { return (u32)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v); }

template <bool LAT0, bool DEG4, bool RND>
__global__ void __launch_bounds__(64) sim_kernel_duo(const DuoParams dp) {
  static_assert(!(RND && LAT0), "random latency needs deadlines");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = dp.k;
  const u32 lane = threadIdx.x, i = lane & 31u;
  const bool hi = lane >= 32u;
  const u32 N = p.N, W = p.W, R = dp.R, Rm = dp.R - 1u, S = dp.S;
  const bool is_node = i < N;
  const u32 inst_raw = blockIdx.x * 2u + (hi ? 1u : 0u);
  const bool real = inst_raw < dp.n_inst;
  const u32 inst = real ? inst_raw : dp.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << i) - 1u;
  const u32 all_nodes = N >= 32 ? 0xFFFFFFFFu : ((1u << N) - 1u);
  const u32 max_values = p.cfg.max_values, max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 rate = p.cfg.rate_mhz;
  const u32 lat_us = LAT0 ? 0u : p.cfg.latency_mean_ms * 1000u;
  // a sender's "src to skip" field never equals 64: without skip-sender every neighbour takes the value
  const u32 me16 = dp.echoback ? (64u << 16) : (i << 16);
  const u32 round_limit = dp.round_limit;

  msim_op *const g_rows = p.rows + (size_t)inst * max_rows;
  u32 *const g_pay = p.payload + (size_t)inst * max_pay;
  // HBM spill behind the LDS ring: {deadline, envelope} pairs in the node's slice of the spill area
  u64 *const my_spill = reinterpret_cast<u64 *>(reinterpret_cast<uint4 *>(p.scratch + (size_t)inst * p.scratch_words + p.spill_off) +
                                                    (size_t)(is_node ? i : 0) * p.spill_cap);

  // LDS of one cluster: [row staging][32 rings][N node sets][32 dummy words for the lanes that hold no node]
  unsigned char *const hmem = smem + (hi ? dp.half_bytes : 0u);
  uint4 *const stage = reinterpret_cast<uint4 *>(hmem);
  u32 *const seen = reinterpret_cast<u32 *>(hmem + dp.off_seen);
  const u32 Wp = W | 1u;   // odd stride between the nodes' sets: word w of every node sits on a different LDS bank
  u32 *const my_seen = is_node ? seen + i * Wp : seen + N * Wp + i;
  // slot-major rings: slot s of lane i at [s * 32 + i] (a wave-wide access to one slot position is contiguous)
  u32 *const ring32 = reinterpret_cast<u32 *>(hmem + dp.off_ring) + i;       // LAT0: the envelope word
  u64 *const ring64 = reinterpret_cast<u64 *>(hmem + dp.off_ring) + i;       // else deadline | envelope << 32
  // RND: the node's queue is a bag of R entries (ring64[0 .. in_n)) with their arrival sequence numbers beside them; lanes that
  // hold no node share one dummy bag; the spill holds {key, sequence} in 16 bytes
  // (slot-major: slot j of node i at [j * (N + 1) + i] — a wave-wide access to one slot is contiguous, free of bank conflicts;
  //  lane-major bags of 128 bytes put every lane on the same banks)
  const u32 BS = N + 1u;   // lanes per slot: the nodes and one dummy shared by the lanes that hold no node
  u32 *const bag_dl = reinterpret_cast<u32 *>(hmem + dp.off_ring) + (is_node ? i : N);             // deadlines: what recv! scans
  u32 *const bag_e = bag_dl + DUO_BAG * BS;                                                           // envelope words
  unsigned short *const bag_seq = reinterpret_cast<unsigned short *>(hmem + dp.off_seq) + (is_node ? i : N);
#define BAGDL(j_) bag_dl[(j_) * BS]
#define BAGE(j_) bag_e[(j_) * BS]
#define BAGSEQ(j_) bag_seq[(j_) * BS]
  u32 *const my_spill12 = p.scratch + (size_t)inst * p.scratch_words + p.spill_off + (size_t)(is_node ? i : 0) * p.spill_cap * 4;   // {deadline, envelope, sequence} x S
  u32 *const log2_tab = reinterpret_cast<u32 *>(smem + dp.off_log2);

  for (u32 k = i; k < N * Wp + 32u; k += 32) seen[k] = 0;
  if (RND) for (u32 k = lane; k < 257u; k += 64) log2_tab[k] = duo_log2_q24[k];
  if (RND) for (u32 k = i; k < DUO_BAG * BS; k += 32) reinterpret_cast<u32 *>(hmem + dp.off_ring)[k] = INF;   // every bag slot is free
  __syncthreads();

  const u32 adj = is_node ? topo_adj(p.cfg.topology, N, i) : 0u;
  const u32 hbase4 = (lane & 32u) << 2;  // byte address of the half's lane 0 for ds_bpermute
  // DEG4 (every node has <= 4 neighbours, lane 31 holds no node): the neighbours in ascending order as bpermute addresses
  // (an unused slot points at lane 31, which never publishes) and the constant part of an envelope received from each
  u32 nbl[4] = {0, 0, 0, 0}, kc[4] = {0, 0, 0, 0}, nb_adj[4] = {0, 0, 0, 0};   // nb_adj (RND): the neighbour's own adjacency mask
  if (DEG4) {
    u32 rem = adj;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 s = rem ? (u32)__builtin_ctz(rem) : 31u;
      rem &= rem - 1u;
      nbl[k] = hbase4 + (s << 2); kc[k] = s << 16;
      if (RND) nb_adj[k] = s < N ? topo_adj(p.cfg.topology, N, s) : 0u;
    }
  }

  // ---- per-lane state: node i and its client (u32 throughout: flags are 0 / 1) ----
  u32 deliver_at = INF;      // INF = recv! holds no envelope (then the queue is empty too: idle receivers poll at once)
  u32 cm = 0;                // the envelope recv! is sleeping on: value | src << 16 | kind << 24 (src 63 = the node's own client)
  u32 in_n = 0, head = 0;    // LDS ring
  u32 bag_used = 0;          // RND: the slots of the LDS bag that hold an envelope (in_n = their number)
  u32 sp_n = 0, s_head = 0;  // HBM spill ring (rare)
  u32 have_creq = 0, creq = 0, creq_t = 0;  // latency > 0: the client's request waits beside the FIFO of server envelopes
  u32 busy = 0;              // the client has an RPC outstanding
  u32 n_cl = 0, n_arr = 0, n_rsv = 0, my_flags = 0;  // client RPCs completed, server envelopes arrived / delivered
  // ---- per-cluster state (uniform within a half) ----
  u32 T = 0, phase = PH_INIT, cutoff = 0, gen_next = 0, gen_k = 0, next_value = 0, sleep_until = 0;
  u32 n_rows = 0, n_payload = 0, flags = 0, rounds = 0;
  u32 next_id = 0;           // RND: message ids (net.clj:103,197), every send! of the cluster in canonical order
  u32 my_seq = 0;            // RND (per lane): arrivals queued at this node so far
  u32 spm_dl = INF, spm_seq = 0, spm_e = 0, spm_i = 0;   // RND: the minimum of the spilled part of the bag, cached in registers
  u32 pbase = 0;             // RND (per lane): id of the first message this node sends in the current round
  const u32 lat_mean = p.cfg.latency_mean_ms;
  const bool lat_uniform = p.cfg.latency_dist == MSIM_LAT_UNIFORM;
  u32 alive = real ? 1u : 0u;
  // when the scheduler next acts (INF: it only waits), and whether every round has to be a GENERAL one until it says otherwise;
  // both change in GENERAL rounds only
  u32 sched_at = real ? 0u : INF, force_general = alive;
  // The generator's draws (one 64-bit draw per generated op, stream S_GEN, counter gen_k) are computed 32 at a time: lane i of a cluster
  // holds draw dc_base + i, the scheduler fetches the one it needs with two ds_bpermute (mix64's three 64-bit multiplications cost a
  // round of the scheduler a quarter of its cycles when every lane computed the same draw)
  u32 dc_base = 0; u64 dc = draw64(key, S_GEN, (u64)i);
  // RND: the same for the messages' latency draws (stream S_LATENCY, counter = message id, net.clj:178-187): a gossip round sends 1.3
  // messages on average, and every lane of the wavefront computed a draw (mix64, the logarithm) for them — a third of the round's vector
  // instructions.  Lane i of a cluster holds the latency (ms) of message id lc_base + i; DUO_RND_IDS keeps the block under the round's ids.
  u32 lc_base = 0xFFFFFFC0u, lc = 0; bool lc_all = false;   // (no block yet: the first round with a send draws one)

  // Two LDS reads are kept one round ahead of their use, so that a round's dependent chain holds one LDS round trip
  // (the ds_bpermute exchange) instead of three:
  //   sw = the word of the node's set that cm's value falls in (re-read after every change of cm or of the set);
  //   nx = the head entry of the ring (valid while in_n != 0; an append to an empty ring sets it from registers).
  u32 sw = 0;
  u32 nx = 0, nx_dl = 0;
  // The helpers below are macros on purpose: as lambdas capturing the state by reference they left the closures (and with
  // them every captured variable) in scratch memory once the optimizer turned a select of two captured values into a select
  // of their addresses.
  // recv! took an envelope: (Thread/sleep (long dt)), net.clj:236-238
#define DUO_COMMIT_TIME(dl_) (LAT0 ? T : ((dl_) <= T ? T : T + (((dl_) - T) / 1000u) * 1000u))
#define DUO_RING_STORE(slot_, e_, dl_) do { if (LAT0) ring32[(slot_) * 32u] = (e_); else ring64[(slot_) * 32u] = (u64)(dl_) | ((u64)(e_) << 32); } while (0)
#define DUO_SEEN_WORD() (reinterpret_cast<u32 *>(reinterpret_cast<unsigned char *>(my_seen) + ((cm >> 3) & 0x1FFCu)))   /* word (value >> 5) */
  // slow, checked append: ring, then spill; used when a ring may fill up this round
#define DUO_PUSH_CHECKED(got_, e_, dl_) do {                                                                              \
    const bool pc_got = (got_); const u32 pc_e = (e_), pc_dl = (dl_);                                                     \
    if (RND) {   /* a bag: append with the node's arrival sequence number (orders equal deadlines like the ids do) */      \
      if (pc_got) {                                                                                                       \
        if (in_n < R) {   /* any free slot will do: a free slot's deadline is INF, `bag_used` says which ones are taken */      \
          const u32 pc_s = (u32)__builtin_ctz(~bag_used);                                                                 \
          BAGDL(pc_s) = pc_dl; BAGE(pc_s) = pc_e; BAGSEQ(pc_s) = (unsigned short)my_seq; bag_used |= 1u << pc_s; in_n++;  \
        }                                                                                                                 \
        else if (sp_n < S) {                                                                                              \
          my_spill12[3 * sp_n] = pc_dl; my_spill12[3 * sp_n + 1] = pc_e; my_spill12[3 * sp_n + 2] = my_seq & 0xFFFFu;     \
          if (sp_n == 0 || pc_dl < spm_dl) { spm_dl = pc_dl; spm_seq = my_seq & 0xFFFFu; spm_e = pc_e; spm_i = sp_n; }   /* (equal deadline: the older one stays) */ \
          sp_n++;                                                                                                         \
        }                                                                                                                 \
        else my_flags |= MSIM_FLAG_INBOX_OVERFLOW;                                                                        \
        my_seq++;                                                                                                         \
      }                                                                                                                   \
    } else {                                                                                                              \
      const bool pc_fit = (in_n < R) & (sp_n == 0);                                                                       \
      if (pc_got & pc_fit) {                                                                                              \
        DUO_RING_STORE((head + in_n) & Rm, pc_e, pc_dl);                                                                  \
        if (in_n == 0) { nx = pc_e; nx_dl = pc_dl; }                                                                      \
        in_n++;                                                                                                           \
      }                                                                                                                   \
      if (pc_got & !pc_fit) {                                                                                             \
        if (sp_n >= S) my_flags |= MSIM_FLAG_INBOX_OVERFLOW;                                                              \
        else { u32 pc_idx = s_head + sp_n; if (pc_idx >= S) pc_idx -= S; my_spill[pc_idx] = (u64)pc_dl | ((u64)pc_e << 32); sp_n++; } \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)
  // idle receivers poll (net.clj:223-247): the minimum (deadline, id) = the FIFO head, or the client's request if its deadline
  // (its send time) is earlier; equal deadlines go to the envelope sent first, which is the queued server envelope.
  // Ends with the prefetch of the next head and of the set word of the (new) envelope.
#define DUO_POLL() do {                                                                                                   \
    const bool pl_idle = deliver_at == INF;                                                                               \
    if (RND) {                                                                                                            \
      /* recv! takes the minimum (deadline, id) of the bag — even if it is not due (net.clj:228-229) — and sleeps on it.  The 16      \
         deadlines of the LDS bag are read in one go (one wait) and reduced with 32-bit minima; only when the minimal deadline       \
         occurs twice (rare) do the arrival sequence numbers (16 bits, relative to the newest) decide. */                            \
      const bool pl_go = pl_idle & ((in_n | sp_n) != 0);                                                                  \
      if (__ballot(pl_go)) {                                                                                              \
        const u32 pl_sb = 32768u - my_seq;                                                                                \
        u32 pl_d[DUO_BAG];                                                                                                   \
        _Pragma("unroll") for (int pl_j = 0; pl_j < (int)DUO_BAG; pl_j++) pl_d[pl_j] = BAGDL(pl_j);   /* (free slots hold INF) */          \
        u32 pl_m[DUO_BAG / 2];                                                                                            \
        _Pragma("unroll") for (int pl_j = 0; pl_j < (int)(DUO_BAG / 2); pl_j++) pl_m[pl_j] = min(pl_d[pl_j], pl_d[pl_j + DUO_BAG / 2]);  \
        _Pragma("unroll") for (int pl_w = (int)(DUO_BAG / 4); pl_w >= 1; pl_w >>= 1)                                      \
          _Pragma("unroll") for (int pl_j = 0; pl_j < pl_w; pl_j++) pl_m[pl_j] = min(pl_m[pl_j], pl_m[pl_j + pl_w]);      \
        const u32 pl_dmin = pl_m[0];                                                                                      \
        u32 pl_bi = 0, pl_cnt = 0;                                                                                        \
        _Pragma("unroll") for (int pl_j = 0; pl_j < (int)DUO_BAG; pl_j++) { const bool pl_eq = pl_d[pl_j] == pl_dmin; pl_bi = pl_eq ? (u32)pl_j : pl_bi; pl_cnt += pl_eq ? 1u : 0u; } \
        u32 pl_bq = 0;   /* the winner's sequence number, relative; only read when it matters */                           \
        const bool pl_tie = pl_go & (in_n != 0) & ((pl_cnt > 1u) | (sp_n != 0 && spm_dl == pl_dmin));                      \
        if (__ballot(pl_tie)) {                                                                                           \
          if (pl_tie) {                                                                                                   \
            pl_bq = 0xFFFFFFFFu;                                                                                          \
            for (u32 pl_j = 0; pl_j < DUO_BAG; pl_j++) {                                                                      \
              if (BAGDL(pl_j) == pl_dmin) { const u32 pl_q = ((u32)BAGSEQ(pl_j) + pl_sb) & 0xFFFFu; if (pl_q < pl_bq) { pl_bq = pl_q; pl_bi = pl_j; } } \
            }                                                                                                             \
          }                                                                                                               \
        }                                                                                                                 \
        u32 pl_be = 0; bool pl_sp = false;                                                                                \
        /* the spilled part of the bag (HBM) takes part through its cached minimum: no memory access unless it wins */      \
        pl_sp = pl_go & (sp_n != 0) & ((in_n == 0) | (spm_dl < pl_dmin) | ((spm_dl == pl_dmin) & (((spm_seq + pl_sb) & 0xFFFFu) < pl_bq))); \
        if (pl_go) {                                                                                                      \
          if (!pl_sp) {   /* take it out of the LDS bag: its slot is free again */                                         \
            pl_be = BAGE(pl_bi);                                                                                          \
            BAGDL(pl_bi) = INF; bag_used &= ~(1u << pl_bi); in_n--;                                                       \
            cm = pl_be; deliver_at = DUO_COMMIT_TIME(pl_dmin);                                                            \
          } else { cm = spm_e; deliver_at = DUO_COMMIT_TIME(spm_dl); }                                                    \
        }                                                                                                                 \
        if (__ballot(pl_sp)) {   /* rare: a spilled envelope was taken — close the gap and find the new minimum of the spill */ \
          if (pl_sp) {                                                                                                    \
            sp_n--;                                                                                                       \
            if (spm_i != sp_n) { my_spill12[3 * spm_i] = my_spill12[3 * sp_n]; my_spill12[3 * spm_i + 1] = my_spill12[3 * sp_n + 1]; my_spill12[3 * spm_i + 2] = my_spill12[3 * sp_n + 2]; } \
            spm_dl = INF; u64 pl_mk = ~0ull;                                                                              \
            for (u32 pl_j = 0; pl_j < sp_n; pl_j++) {                                                                     \
              const u32 pl_d = my_spill12[3 * pl_j], pl_e2 = my_spill12[3 * pl_j + 1], pl_q = my_spill12[3 * pl_j + 2];   \
              const u64 pl_k = ((u64)pl_d << 20) | (u64)(((pl_q + pl_sb) & 0xFFFFu) << 4);                                \
              if (pl_k < pl_mk) { pl_mk = pl_k; spm_dl = pl_d; spm_seq = pl_q; spm_e = pl_e2; spm_i = pl_j; }             \
            }                                                                                                             \
          }                                                                                                               \
        }                                                                                                                 \
      }                                                                                                                   \
    } else if (LAT0) {                                                                                                    \
      const bool pl_can = pl_idle & (in_n != 0);                                                                          \
      cm = pl_can ? nx : cm; deliver_at = pl_can ? T : deliver_at;                                                        \
      head = (head + (pl_can ? 1u : 0u)) & Rm; in_n -= pl_can ? 1u : 0u;                                                  \
    } else {                                                                                                              \
      const u32 pl_hx = in_n != 0 ? nx_dl : INF;                                                                          \
      const bool pl_c = pl_idle & (have_creq != 0) & (creq_t < pl_hx);                                                    \
      const bool pl_r = pl_idle & !pl_c & (in_n != 0);                                                                    \
      const u32 pl_ex = pl_c ? creq_t : nx_dl;                                                                            \
      const u32 pl_e = pl_c ? creq : nx;                                                                                  \
      const u32 pl_t = DUO_COMMIT_TIME(pl_ex);                                                                            \
      cm = (pl_c | pl_r) ? pl_e : cm;                                                                                     \
      deliver_at = (pl_c | pl_r) ? pl_t : deliver_at;                                                                     \
      have_creq = pl_c ? 0u : have_creq;                                                                                  \
      head = (head + (pl_r ? 1u : 0u)) & Rm; in_n -= pl_r ? 1u : 0u;                                                      \
    }                                                                                                                     \
    if (!RND) {                                                                                                           \
      if (__ballot(sp_n != 0)) {   /* refill the ring from the spill: "ring empty" always means "queue empty" (rare) */    \
        while (sp_n != 0 && in_n < R) {                                                                                   \
          const u64 pl_s = my_spill[s_head];                                                                              \
          s_head++; if (s_head >= S) s_head = 0; sp_n--;                                                                  \
          DUO_RING_STORE((head + in_n) & Rm, (u32)(pl_s >> 32), (u32)pl_s);                                               \
          in_n++;                                                                                                         \
        }                                                                                                                 \
      }                                                                                                                   \
      if (LAT0) nx = ring32[head * 32u]; else { const u64 pl_h = ring64[head * 32u]; nx_dl = (u32)pl_h; nx = (u32)(pl_h >> 32); } \
    }                                                                                                                     \
    DUO_SW_PREFETCH();                                                                                                    \
  } while (0)
  // R3 for a broadcast envelope (gossip or the client's own): dedup against the node's set; pub_ = what the node publishes
  // to its neighbours: bit 31 | value | src to skip << 16, or 0
#ifdef DUO_NO_SWPF   /* A/B build: the set word is read where it is used */
#define DUO_SW_PREFETCH() do { } while (0)
#define DUO_SW_NOW() do { sw = *DUO_SEEN_WORD(); } while (0)
#else
#define DUO_SW_PREFETCH() do { sw = *DUO_SEEN_WORD(); } while (0)
#define DUO_SW_NOW() do { } while (0)
#endif
#define DUO_R3_SEEN(handle_, pub_) do {                                                                                   \
    DUO_SW_NOW();                                                                                                         \
    const u32 r3_bit = 1u << (cm & 31u);                                                                                  \
    const bool r3_new = (handle_) & ((sw & r3_bit) == 0);                                                                 \
    if (r3_new) *DUO_SEEN_WORD() = sw | r3_bit;                                                                           \
    pub_ = r3_new ? (0x80000000u | (cm & 0x3FFFFFu)) : 0u;                                                                \
  } while (0)
  // COMMIT of the fan-outs, receiver side: every node pulls what its neighbours publish, in ascending sender order (= id order,
  // net.clj:197), and appends it to its own queue.  got <=> the neighbour sends and does not skip this node:
  // z = (x & 0x803F0000) ^ (0x80000000 | me << 16) is > 0 exactly then (negative: not sending; 0: sending, skipping me).
  // RND: the message's id = the sender's first id of the round + the rank of this node in the sender's fan-out (ascending
  // destination, net.clj:197); its latency is drawn from the id (net.clj:178-187: uniform int in [0, 2 mean) or floor(mean * -ln u))
#define DUO_LAT_MS(r_) (lat_uniform ? scale32((r_), 2u * lat_mean) : (u32)(((u64)lat_mean * duo_neg_ln_q16((r_), log2_tab)) >> 16))
#define DUO_RND_DEADLINE(x_, base_, fanadj_, dl_) do {                                                                    \
    const u32 rd_src = ((x_) >> 16) & 63u;                                                                                \
    const u32 rd_fan = dp.echoback ? (fanadj_) : ((fanadj_) & ~(rd_src < 32u ? (1u << rd_src) : 0u));                      \
    const u32 rd_id = (base_) + __popc(rd_fan & lt);                                                                      \
    u32 rd_ms;                                                                                                            \
    if (lc_all) rd_ms = bperm(hbase4 + (((rd_id - lc_base) & 31u) << 2), lc);   /* the usual round: the draw is in the cluster's block */ \
    else rd_ms = DUO_LAT_MS(draw32(key, S_LATENCY, rd_id));                                                               \
    dl_ = T + rd_ms * 1000u;                                                                                              \
  } while (0)
#define DUO_ARRIVALS(pub_) do {                                                                                           \
    const u32 ar_dl = T + lat_us;                                                                                         \
    const u32 ar_zk = 0x80000000u | me16;                                                                                 \
    if (DEG4) {                                                                                                           \
      const u32 ar_x[4] = {bperm(nbl[0], pub_), bperm(nbl[1], pub_), bperm(nbl[2], pub_), bperm(nbl[3], pub_)};           \
      if (RND) {                                                                                                          \
        const u32 ar_b[4] = {bperm(nbl[0], pbase), bperm(nbl[1], pbase), bperm(nbl[2], pbase), bperm(nbl[3], pbase)};     \
        bool ar_g[4]; u32 ar_cnt = 0;                                                                                     \
        _Pragma("unroll") for (int ar_k = 0; ar_k < 4; ar_k++) { ar_g[ar_k] = (int)((ar_x[ar_k] & 0x803F0000u) ^ ar_zk) > 0; ar_cnt += ar_g[ar_k] ? 1u : 0u; } \
        n_arr += ar_cnt;                                                                                                  \
        if (!__ballot(ar_cnt > 1u)) {                                                                                     \
          /* the usual round: no node receives two envelopes — one latency draw serves every lane */                       \
          u32 ar_xx = 0, ar_bb = 0, ar_aa = 0, ar_cc = 0;                                                                 \
          _Pragma("unroll") for (int ar_k = 0; ar_k < 4; ar_k++) {                                                        \
            ar_xx = ar_g[ar_k] ? ar_x[ar_k] : ar_xx; ar_bb = ar_g[ar_k] ? ar_b[ar_k] : ar_bb;                             \
            ar_aa = ar_g[ar_k] ? nb_adj[ar_k] : ar_aa; ar_cc = ar_g[ar_k] ? kc[ar_k] : ar_cc;                             \
          }                                                                                                               \
          u32 ar_d; DUO_RND_DEADLINE(ar_xx, ar_bb, ar_aa, ar_d);                                                          \
          DUO_PUSH_CHECKED(ar_cnt != 0u, (ar_xx & 0xFFFFu) | ar_cc, ar_d);                                                \
        } else {                                                                                                          \
          _Pragma("unroll") for (int ar_k = 0; ar_k < 4; ar_k++) {                                                        \
            if (__ballot(ar_g[ar_k])) {                                                                                   \
              u32 ar_d; DUO_RND_DEADLINE(ar_x[ar_k], ar_b[ar_k], nb_adj[ar_k], ar_d);                                     \
              DUO_PUSH_CHECKED(ar_g[ar_k], (ar_x[ar_k] & 0xFFFFu) | kc[ar_k], ar_d);                                      \
            }                                                                                                             \
          }                                                                                                               \
        }                                                                                                                 \
      } else if (__builtin_expect(!__ballot((in_n + 4u > R) | (sp_n != 0)), 1)) {                                         \
        /* every ring has room for a full round of arrivals: plain stores at the tail, the count decides what stays */    \
        const u32 ar_in0 = in_n;                                                                                          \
        _Pragma("unroll") for (int ar_k = 0; ar_k < 4; ar_k++) {                                                          \
          const bool ar_got = (int)((ar_x[ar_k] & 0x803F0000u) ^ ar_zk) > 0;                                              \
          const u32 ar_e = (ar_x[ar_k] & 0xFFFFu) | kc[ar_k];                                                             \
          DUO_RING_STORE((head + in_n) & Rm, ar_e, ar_dl);                                                                \
          const bool ar_first = ar_got & (in_n == 0);                                                                     \
          nx = ar_first ? ar_e : nx; if (!LAT0) nx_dl = ar_first ? ar_dl : nx_dl;                                         \
          in_n += ar_got ? 1u : 0u;                                                                                       \
        }                                                                                                                 \
        n_arr += in_n - ar_in0;                                                                                           \
      } else {                                                                                                            \
        _Pragma("unroll") for (int ar_k = 0; ar_k < 4; ar_k++) {                                                          \
          const bool ar_got = (int)((ar_x[ar_k] & 0x803F0000u) ^ ar_zk) > 0;                                              \
          n_arr += ar_got ? 1u : 0u;                                                                                      \
          DUO_PUSH_CHECKED(ar_got, (ar_x[ar_k] & 0xFFFFu) | kc[ar_k], ar_dl);                                             \
        }                                                                                                                 \
      }                                                                                                                   \
    } else {                                                                                                              \
      u32 ar_rem = adj;                                                                                                   \
      for (u32 ar_k = 0; ar_k < dp.deg; ar_k++) {                                                                         \
        const bool ar_has = ar_rem != 0;                                                                                  \
        const u32 ar_s = ar_has ? (u32)__builtin_ctz(ar_rem) : i;                                                         \
        ar_rem &= ar_rem - 1u;                                                                                            \
        const u32 ar_xx = bperm(hbase4 + (ar_s << 2), pub_);                                                              \
        const bool ar_got = ar_has & ((int)((ar_xx & 0x803F0000u) ^ ar_zk) > 0);                                          \
        u32 ar_d = ar_dl;                                                                                                 \
        if (RND) {                                                                                                        \
          const u32 ar_bb = bperm(hbase4 + (ar_s << 2), pbase);                                                           \
          const u32 ar_sadj = bperm(hbase4 + (ar_s << 2), adj);                                                           \
          if (__ballot(ar_got)) DUO_RND_DEADLINE(ar_xx, ar_bb, ar_sadj, ar_d);                                            \
        }                                                                                                                 \
        n_arr += ar_got ? 1u : 0u;                                                                                        \
        DUO_PUSH_CHECKED(ar_got, (ar_xx & 0xFFFFu) | (ar_s << 16), ar_d);                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)
  // RND: ids of this round's sends in canonical order (node order; a node's fan-out in ascending destination, then its reply):
  // pbase = the node's first id; the cluster's id counter moves past all of them
#define DUO_RND_IDS(pub_, rep_) do {                                                                                      \
    const u32 id_src = (cm >> 16) & 63u;                                                                             \
    const u32 id_fan = (pub_) == 0 ? 0u : (dp.echoback ? adj : (adj & ~(id_src < 32u ? (1u << id_src) : 0u)));            \
    const u32 id_cnt = __popc(id_fan) + ((rep_) ? 1u : 0u);                                                               \
    const u32 id_incl = scan32(id_cnt);                                                                                   \
    pbase = next_id + id_incl - id_cnt;                                                                                   \
    const u32 id_lo = rdlane(id_incl, 31), id_up = rdlane(id_incl, 63);                                                   \
    const u32 id_first = next_id;                                                                                         \
    next_id += hi ? id_up : id_lo;                                                                                        \
    /* the latencies of the round's ids come from the cluster's block of 32 (lane i holds the one of id lc_base + i): a block that \
       does not reach the round's last id is drawn again from the round's first id on; a round of more than 32 sends draws its own */ \
    const bool id_rf = next_id - lc_base > 32u;                                                                           \
    if (__ballot(id_rf)) { const u32 id_nl = DUO_LAT_MS(draw32(key, S_LATENCY, id_first + i)); lc = id_rf ? id_nl : lc; lc_base = id_rf ? id_first : lc_base; } \
    lc_all = !__ballot(next_id - lc_base > 32u);                                                                          \
  } while (0)

#ifdef DUO_PROF   // developer build (tools/variant_lib.sh prof duo.hip -DDUO_PROF): wave-round counts and cycles of the two round bodies -> meta
  u64 pf_t0 = __builtin_readcyclecounter(), pf_gen = 0; u32 pf_ngen = 0, pf_nwave = 0;
#endif
#if defined(DUO_PROF2) || defined(DUO_PROF3)  // developer builds: cycles of the sections of the gossip round (PROF2) or of the GENERAL round (PROF3)
  u64 p2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p2_t = __builtin_readcyclecounter();   // -> meta of the wavefront's two instances (replaces DUO_PROF's numbers)
#define PX_MARK(i_) { const u64 p2_n = __builtin_readcyclecounter(); p2[i_] += p2_n - p2_t; p2_t = p2_n; }
#endif
#ifdef DUO_PROF2
#define P2_MARK(i_) PX_MARK(i_)
#else
#define P2_MARK(i_)
#endif
#ifdef DUO_PROF3
#define P3_MARK(i_) PX_MARK(i_)
#else
#define P3_MARK(i_)
#endif
  for (;;) {
    // ---- gossip rounds of both clusters, until one of them needs a GENERAL round ----
    for (;;) {
#ifdef DUO_PROF
      pf_nwave++;
#endif
      // R0: the cluster's time: stay at T while something is due, else jump to the next delivery / scheduler event.
      // Only looked at when one of the two clusters has nothing due (a scalar test on the halves of one ballot).
      P2_MARK(4)
      bool due_n = deliver_at <= T;
      bool stuck_any = false;
      {
        const u64 db = __ballot(due_n);
        if (__builtin_expect((u32)db == 0 || (u32)(db >> 32) == 0, 0)) {
          const bool none_due = hi ? (u32)(db >> 32) == 0 : (u32)db == 0;
          const bool idle_h = (alive != 0) & (sched_at > T) & none_due;
          if (__ballot(idle_h)) {
            const u32 km = min(half_min(deliver_at, hi), sched_at);
            const bool stuck = idle_h & (km == INF);   // nothing will ever happen (oracle: same flag, the round counts)
            flags |= stuck ? (u32)MSIM_FLAG_ROUND_LIMIT : 0u;
            alive = stuck ? 0u : alive; sched_at = stuck ? INF : sched_at; force_general = stuck ? 0u : force_general;
            rounds += stuck ? 1u : 0u;
            T = (idle_h & !stuck) ? km : T;
            stuck_any = __ballot(stuck) != 0;
            due_n = deliver_at <= T;
          }
          // the round limit is looked at here and in GENERAL rounds (a stretch of pure gossip always ends in one of the two)
          force_general = (alive != 0 && rounds >= round_limit) ? 1u : force_general;
        }
      }
      rounds += alive;
      const bool special = due_n & ((cm >> 24) != DK_PLAIN);
      const bool gen = (alive != 0) & ((force_general != 0) | (sched_at <= T) | special);
      if (__ballot(gen) != 0 || stuck_any) break;   // (a GENERAL round is a superset of a gossip round: harmless for the other cluster)
      {   // ---- a round in which both clusters only gossip ----
        P2_MARK(0)
        u32 pub; DUO_R3_SEEN(due_n, pub);
        deliver_at = due_n ? INF : deliver_at;
        n_rsv += due_n ? 1u : 0u;
        P2_MARK(1)
        if (__ballot(pub != 0)) {
          if (RND) DUO_RND_IDS(pub, false);
          P2_MARK(2)
          DUO_ARRIVALS(pub);
        }
        P2_MARK(3)
        DUO_POLL();
      }
    }
    if (!__ballot(alive != 0)) break;
#ifdef DUO_PROF
    const u64 pf_a = __builtin_readcyclecounter();
#endif
    {   // ---- a round in which a cluster's scheduler acts or a node handles its client's request ----
    P3_MARK(0)   // [0] = the gossip rounds
    u32 inv_row = 0, inv_packed = 0, inv_value = 0;
    u32 cmp_row = 0, cmp_packed = 0, cmp_value = 0, cmp_len = 0;
    // ---- R1: scheduler (core.clj:67-80): phase actions, one generated op ----
    u32 mark = 0, m_kind = 0, m_val = 0;
    const bool act = alive != 0 && sched_at <= T;
    if (__ballot(act && phase != PH_MAIN)) {   // rare: db setup, topology, final reads
      if (act && phase == PH_INIT) { mark = is_node; m_kind = DK_INIT; phase = PH_INIT_WAIT; }
      else if (act && phase == PH_TOPO) { mark = is_node; m_kind = DK_TOPO; phase = PH_TOPO_WAIT; }
      else if (act && (phase == PH_SLEEP || phase == PH_FINAL)) { mark = is_node; m_kind = DK_READ_FINAL; phase = PH_FINAL_WAIT; }  // broadcast.clj:240
    }
    if (__ballot(act && phase == PH_MAIN)) {
      const u32 free_mask = all_nodes & ~hb(busy != 0, hi);
      const bool gen = act && phase == PH_MAIN && rate > 0 && gen_next < cutoff && gen_next <= T && free_mask != 0;
      // one 64-bit draw per generated op: high word -> stagger, low word -> worker pick / gen/mix
      {
        const bool dc_refill = gen_k - dc_base >= 32u;   // (uniform within a cluster)
        if (__ballot(dc_refill)) { const u64 dc_new = draw64(key, S_GEN, (u64)gen_k + i); dc = dc_refill ? dc_new : dc; dc_base = dc_refill ? gen_k : dc_base; }
      }
      const u32 dc_at = hbase4 + ((gen_k - dc_base) << 2);
      const u32 r_hi = bperm(dc_at, (u32)(dc >> 32)), r_lo = bperm(dc_at, (u32)dc);
      const u32 pick = scale32(r_lo, __popc(free_mask));
      const bool sel = gen && is_node && busy == 0 && (u32)__popc(free_mask & lt) == pick;
      const bool is_rd = (r_lo & 1u) != 0;
      const bool ovf = gen && !is_rd && next_value >= max_values;
      if (ovf) { flags |= MSIM_FLAG_VALUES_OVERFLOW; alive = 0; }
      mark = sel && !ovf ? 1u : mark;
      m_kind = gen ? (is_rd ? DK_READ : DK_BCAST) : m_kind;
      m_val = gen && !is_rd ? next_value : m_val;
      next_value += gen && !is_rd && !ovf ? 1u : 0u;
      gen_k += gen ? 1u : 0u;
      gen_next = gen ? T + __umulhi(r_hi, p.gen_period2_us) : gen_next;
    }
    P3_MARK(1)   // [1] = R1 scheduler
    // ---- R2: marked clients invoke; the request reaches this lane's own node (no latency: a client is involved) ----
    if (__ballot(mark != 0 && alive != 0)) {
      const bool inv = mark != 0 && alive != 0;
      busy = inv ? 1u : busy;
      const bool is_op = inv && m_kind <= DK_READ_FINAL;
      inv_row = is_op ? 1u : 0u;
      inv_packed = MSIM_T_INVOKE | ((m_kind == DK_BCAST ? MSIM_F_BROADCAST : MSIM_F_READ) << 2) | ((m_kind == DK_READ_FINAL ? 1u : 0u) << 11) | (i << 12);
      inv_value = m_kind == DK_BCAST ? m_val : MSIM_NO_VALUE;
      const u32 e = (m_kind == DK_BCAST ? m_val : 0u) | (63u << 16) | (m_kind << 24);
      if (RND) next_id += __popc(hb(inv, hi));   // the requests' ids, slot order (their latency is 0: no draw)
      if (LAT0 || RND) {
        const bool direct = inv & (deliver_at == INF);   // an idle node's recv! takes the request at once (its queue is empty)
        cm = direct ? e : cm; deliver_at = direct ? T : deliver_at;
        DUO_PUSH_CHECKED(inv & !direct, e, T);
      } else {
        if (inv && have_creq != 0) my_flags |= MSIM_FLAG_INBOX_OVERFLOW;   // cannot happen without client timeouts
        have_creq = inv ? 1u : have_creq; creq = inv ? e : creq; creq_t = inv ? T : creq_t;
      }
      // LAT0 / RND: nobody has to poll here — an idle node took its request directly (only the set word of its new envelope is
      // missing), every other node holds an envelope already; constant latency > 0: the request waits beside the FIFO, recv! chooses
      if (LAT0 || RND) DUO_SW_PREFETCH(); else DUO_POLL();
    }

    P3_MARK(2)   // [2] = R2 invoke + poll
    // ---- R3: one input per node: the due envelope ----
    const bool due_n = alive != 0 && deliver_at <= T;
    const u32 kind = cm >> 24;
    const u32 v = cm & 0xFFFFu;
    u32 pub; DUO_R3_SEEN(due_n & (kind <= DK_BCAST), pub);
    deliver_at = due_n ? INF : deliver_at;
    n_rsv += (due_n && kind == DK_PLAIN) ? 1u : 0u;
    const bool req = due_n && kind != DK_PLAIN;   // a request of this lane's client: handled, answered and completed in this round
    n_cl += req ? 1u : 0u;
    busy = req ? 0u : busy;
    const bool rd = req && (kind == DK_READ || kind == DK_READ_FINAL);
    cmp_row = (req && kind == DK_BCAST) ? 1u : 0u;
    cmp_packed = MSIM_T_OK | (MSIM_F_BROADCAST << 2) | (i << 12); cmp_value = v;
    // read -> read_ok with the whole set: the cluster's lanes copy the node's set LDS -> HBM payload
    if (__ballot(rd)) {
      wave_lds_fence();
      const u32 rdm = hb(rd, hi);
      const u32 words = (next_value + 31u) >> 5;
      const u32 my_rank = __popc(rdm & lt);
      const bool ok = n_payload + (my_rank + 1u) * words <= max_pay;   // payload_alloc of the oracle, reader by reader
      const u32 my_off = ok ? n_payload + my_rank * words : 0u;
      if (rd) {
        if (!ok) my_flags |= MSIM_FLAG_PAYLOAD_OVERFLOW;
        cmp_row = 1; cmp_packed = MSIM_T_OK | (MSIM_F_READ << 2) | ((kind == DK_READ_FINAL ? 1u : 0u) << 11) | (i << 12);
        cmp_value = my_off; cmp_len = words;
      }
      const u32 okm = hb(rd && ok, hi);
      u32 m = okm;
      if (!__ballot((okm & (okm - 1u)) != 0)) {   // the usual round: at most one reader per cluster — lane w copies word w of its set
        const u32 r1 = okm ? (u32)__builtin_ctz(okm) : 0u;
        for (u32 w = i; __ballot(okm != 0 && w < words); w += 32)
          if (okm != 0 && w < words) g_pay[n_payload + w] = seen[r1 * Wp + w];
        m = 0;
      }
      while (__ballot(m != 0)) {
        const bool on = m != 0;
        const u32 r = on ? (u32)__builtin_ctz(m) : 0u;
        m &= m - 1u;
        const u32 r_off = bperm(hbase4 + (r << 2), my_off);
        for (u32 w = i; __ballot(on && w < words); w += 32)
          if (on && w < words) g_pay[r_off + w] = seen[r * Wp + w];
      }
      n_payload += __popc(okm) * words;
    }

    P3_MARK(3)   // [3] = R3 (dedup, read copies)
    if (RND) { if (__ballot((pub != 0) | req)) DUO_RND_IDS(pub, req); }
    if (__ballot(pub != 0)) DUO_ARRIVALS(pub);
    P3_MARK(4)   // [4] = ids + arrivals
    DUO_POLL();
    P3_MARK(5)   // [5] = poll

    // ---- R4 + history rows: invocations (slot order), then completions (slot order) ----
    if (__ballot((inv_row | cmp_row) != 0)) {
      const u32 imask = hb(inv_row != 0, hi), cmask = hb(cmp_row != 0, hi);
      const u32 ni = __popc(imask), nr = ni + __popc(cmask);
      const bool ovf = alive != 0 && nr != 0 && n_rows + nr > max_rows;
      if (ovf) { flags |= MSIM_FLAG_ROWS_OVERFLOW; alive = 0; }
      const u64 tns = (u64)T * 1000ull;
      const u32 tlo = (u32)tns, thi = (u32)(tns >> 32);
      if (RND || DUO_DIRECT) {   // the bags of the random-latency layout take the LDS a staging area would need: rows go straight to HBM
        if (inv_row != 0 && !ovf) reinterpret_cast<uint4 *>(g_rows)[n_rows + __popc(imask & lt)] = make_uint4(tlo, thi, inv_packed, inv_value);
        if (cmp_row != 0 && !ovf) reinterpret_cast<uint4 *>(g_rows)[n_rows + ni + __popc(cmask & lt)] = make_uint4(tlo, thi | (cmp_len << 16), cmp_packed, cmp_value);
      } else {
        if (inv_row != 0 && !ovf) stage[(n_rows + __popc(imask & lt)) % DUO_STAGE_ROWS] = make_uint4(tlo, thi, inv_packed, inv_value);
        if (cmp_row != 0 && !ovf) stage[(n_rows + ni + __popc(cmask & lt)) % DUO_STAGE_ROWS] = make_uint4(tlo, thi | (cmp_len << 16), cmp_packed, cmp_value);
      }
      const u32 new_n = ovf ? n_rows : n_rows + nr;
      const bool flush = !RND && !DUO_DIRECT && (new_n >> 6) != (n_rows >> 6);   // a 64-row block completed (at most one per round: nr <= 64)
      if (__ballot(flush)) {
        wave_lds_fence();
        if (flush) {
          const u32 g0 = (n_rows >> 6) * 64u + i;
          if (g0 < max_rows) reinterpret_cast<uint4 *>(g_rows)[g0] = stage[g0 % DUO_STAGE_ROWS];
          if (g0 + 32u < max_rows) reinterpret_cast<uint4 *>(g_rows)[g0 + 32u] = stage[(g0 + 32u) % DUO_STAGE_ROWS];
        }
        wave_lds_fence();
      }
      n_rows = new_n;
    }

    P3_MARK(6)   // [6] = rows
    // ---- the scheduler's view for the rounds to come: time-free phase transitions (oracle: sched_resolve), when it
    //      acts next (sched_due), and whether plain gossip rounds may run meanwhile ----
    const u32 hbusy = hb(busy != 0, hi);
    if (__ballot(alive != 0 && (phase != PH_MAIN || !(rate > 0 && gen_next < cutoff) || rounds > round_limit))) {
      for (;;) {
        bool ch = false;
        if (alive != 0) {
          if (phase == PH_INIT_WAIT && hbusy == 0) { phase = PH_TOPO; ch = true; }
          if (phase == PH_TOPO_WAIT && hbusy == 0) { phase = PH_MAIN_START; ch = true; }
          if (phase == PH_MAIN_START) { cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; phase = PH_MAIN; ch = true; }
          if (phase == PH_MAIN && !(rate > 0 && gen_next < cutoff) && !(rate == 0 && T < cutoff)) { phase = PH_DRAIN; ch = true; }
          if (phase == PH_DRAIN && hbusy == 0) { phase = PH_SLEEP; sleep_until = T + p.cfg.quiesce_ms * 1000u; ch = true; }
          if (phase == PH_FINAL_WAIT && hbusy == 0) { phase = PH_DONE; ch = true; }
        }
        if (!__ballot(ch)) break;
      }
      if (phase == PH_DONE) alive = 0;
      if (alive != 0 && rounds > round_limit) { flags |= MSIM_FLAG_ROUND_LIMIT; alive = 0; }
      const bool gen_live = rate > 0 && gen_next < cutoff;
      u32 sa = INF;
      if (phase == PH_MAIN) {
        if (gen_live && (all_nodes & ~hbusy) != 0) sa = gen_next;
        if (rate == 0) sa = min(sa, cutoff);
      } else if (phase == PH_INIT || phase == PH_TOPO || phase == PH_FINAL) sa = T;
      else if (phase == PH_SLEEP) sa = sleep_until;
      sched_at = alive != 0 ? sa : INF;
      force_general = (alive != 0 && !((phase == PH_MAIN && gen_live) || phase == PH_SLEEP)) ? 1u : 0u;
    } else {
      // every live cluster of the wavefront is in the main phase with its generator running (nearly every GENERAL round): the
      // scheduler acts again when the generator's next op is due and a worker is free, plain gossip rounds may run meanwhile
      sched_at = (alive != 0 && (all_nodes & ~hbusy) != 0) ? gen_next : INF;
      force_general = 0;
    }
    if (__ballot(alive == 0)) {
      if (alive == 0) { deliver_at = INF; in_n = 0; sp_n = 0; have_creq = 0; bag_used = 0; }   // a finished cluster takes no further part
    }
    P3_MARK(7)   // [7] = the scheduler's view
    }
#ifdef DUO_PROF
    pf_gen += __builtin_readcyclecounter() - pf_a; pf_ngen++;
#endif
    if (!__ballot(alive != 0)) break;
  }
#ifdef DUO_PROF
  const u64 pf_tot = __builtin_readcyclecounter() - pf_t0;
#endif

  // ---- epilogue: the partial row block, net stats, meta ----
  __syncthreads();
  {
    const u32 g0 = (n_rows >> 6) * 64u + i;
    if (!RND && !DUO_DIRECT && real && g0 < n_rows) reinterpret_cast<uint4 *>(g_rows)[g0] = stage[g0 % DUO_STAGE_ROWS];
    if (!RND && !DUO_DIRECT && real && g0 + 32u < n_rows) reinterpret_cast<uint4 *>(g_rows)[g0 + 32u] = stage[(g0 + 32u) % DUO_STAGE_ROWS];
  }
  const u32 sc_cl = wave_incl_scan(n_cl), sc_arr = wave_incl_scan(n_arr), sc_rsv = wave_incl_scan(n_rsv);
  const u32 lo_cl = rdlane(sc_cl, 31), lo_arr = rdlane(sc_arr, 31), lo_rsv = rdlane(sc_rsv, 31);
  const u32 t_cl = hi ? rdlane(sc_cl, 63) - lo_cl : lo_cl;
  const u32 t_arr = hi ? rdlane(sc_arr, 63) - lo_arr : lo_arr;
  const u32 t_rsv = hi ? rdlane(sc_rsv, 63) - lo_rsv : lo_rsv;
  for (u32 b = 1; b <= MSIM_FLAG_JOURNAL_OVERFLOW; b <<= 1) if (hb((my_flags & b) != 0, hi)) flags |= b;
  if (real && i == 0) {
    // every client RPC is a request and a reply, each sent and received once (no loss, no timeouts in this layout)
    msim_net_stats st;
    st.clients_send = 2ull * t_cl; st.clients_recv = 2ull * t_cl;
    st.servers_send = t_arr; st.servers_recv = t_rsv;
    st.all_send = st.clients_send + st.servers_send; st.all_recv = st.clients_recv + st.servers_recv;
    p.stats[inst] = st;
    msim_inst_meta m; m.n_rows = n_rows; m.n_payload_words = n_payload; m.flags = flags; m.n_rounds = rounds;
    m.n_events = 0; m.reserved[0] = 0; m.reserved[1] = 0; m.reserved[2] = 0;
#ifdef DUO_PROF
    m.n_events = pf_ngen; m.reserved[0] = pf_nwave; m.reserved[1] = (u32)(pf_gen >> 6); m.reserved[2] = (u32)(pf_tot >> 6);
#endif
#if defined(DUO_PROF2) || defined(DUO_PROF3)
    if (!hi) { m.n_events = (u32)(p2[0] >> 6); m.reserved[0] = (u32)(p2[1] >> 6); m.reserved[1] = (u32)(p2[2] >> 6); m.reserved[2] = (u32)(p2[3] >> 6); }
    else { m.n_events = (u32)(p2[4] >> 6); m.reserved[0] = (u32)(p2[5] >> 6); m.reserved[1] = (u32)(p2[6] >> 6); m.reserved[2] = (u32)(p2[7] >> 6); }
#endif
    p.meta[inst] = m;
  }
}

}  // namespace

// Whether the duo layout simulates this configuration (see the header of this file).
bool msim_duo_eligible(const msim_config &c) {
  if (c.node_program != MSIM_NODE_BCAST_FF && c.node_program != MSIM_NODE_BCAST_FF_ECHOBACK) return false;
  if (c.n_nodes > 32 || c.concurrency != c.n_nodes) return false;
  if (c.p_loss_q32 != 0 || c.nemesis_mask != 0 || c.journal_capacity != 0) return false;
  // an RPC completes within one (maximal) latency of virtual time: no client timeout can fire (client.clj:96-103, db.clj:54).
  // constant: the mean; uniform: below 2 x mean; exponential: mean x -ln(2^-32) < 22.2 x mean
  const uint64_t worst = c.latency_dist == MSIM_LAT_CONSTANT ? c.latency_mean_ms : c.latency_dist == MSIM_LAT_UNIFORM ? 2ull * c.latency_mean_ms : 23ull * c.latency_mean_ms;
  if (worst >= c.client_timeout_ms || worst >= 10000u) return false;
  if (c.latency_dist != MSIM_LAT_CONSTANT && (uint64_t)c.inbox_capacity + c.spill_capacity >= 16384u) return false;   // 16-bit arrival sequence numbers
  if (c.max_values > 65536u) return false;   // a value travels in 16 bits of the envelope word
  return true;
}

static uint32_t duo_degree(const msim_config &c) {
  uint32_t d = 0;
  for (uint32_t a = 0; a < c.n_nodes; a++) {
    uint32_t m = 0, n = c.n_nodes;
    switch (c.topology) {   // same shapes as topo_adj (broadcast.clj:40-185)
      case MSIM_TOPO_GRID: { uint32_t side = 1; while (side * side < n) side++; const uint32_t i = a / side, j = a % side;
        m = (j + 1 < side && a + 1 < n) + (j > 0) + (a + side < n) + (i > 0); } break;
      case MSIM_TOPO_LINE: m = (a + 1 < n) + (a > 0); break;
      case MSIM_TOPO_TOTAL: m = n - 1; break;
      default: { const uint32_t b = c.topology == MSIM_TOPO_TREE2 ? 2 : c.topology == MSIM_TOPO_TREE3 ? 3 : 4;
        m = a > 0; for (uint32_t k = 1; k <= b; k++) m += b * a + k < n; }
    }
    if (m > d) d = m;
  }
  return d;
}

template <bool LAT0, bool DEG4, bool RND>
static hipError_t duo_launch(const DuoParams &dp, dim3 grid, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sim_kernel_duo<LAT0, DEG4, RND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((sim_kernel_duo<LAT0, DEG4, RND>), grid, dim3(64), lds, st, dp);
  return hipGetLastError();
}

// Launches the duo kernel for n clusters on `st`; returns MSIM_LAYOUT_DOES_NOT_FIT if the cluster state does not fit (the caller
// then runs the one-cluster-per-wavefront kernels).
hipError_t msim_launch_duo(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  DuoParams dp;
  dp.k = kp; dp.n_inst = n;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT;
  const bool lat0 = !rnd && c.latency_mean_ms == 0;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  // LDS of one cluster = [row staging (not RND)] [queues] [node sets + a dummy word per lane].  The queues take what keeps EIGHT
  // wavefronts on a CU (20 KiB each: BASELINE's 4096 clusters are 2048 wavefronts on 256 CUs — a ninth would run alone in a second
  // pass), at least 8 entries per node; what does not fit goes to the HBM spill area.
  const size_t seen_bytes = (((size_t)kp.N * (kp.W | 1u) + 32) * 4 + 15) & ~(size_t)15;   // odd stride between the nodes' sets
  const size_t stage_bytes = (rnd || DUO_DIRECT) ? 0 : DUO_STAGE_ROWS * 16;   // (rows are staged only in the -DDUO_STAGED_ROWS build)
  const size_t fixed = seen_bytes + stage_bytes;
  const size_t per_entry = rnd ? 0 : (size_t)32 * (lat0 ? 4 : 8);   // (RND: bags of a fixed 16 entries)
  const size_t budget = (20 * 1024 - (rnd ? 257 * 4 + 16 : 0)) / 2;   // (RND with DUO_BAG = 8: 13 KiB per wavefront, twelve per CU)
  uint32_t R = rnd ? DUO_BAG : 8;
  while (!rnd && R < 64 && R < cap_tot && fixed + ((per_entry * (R * 2) + 15) & ~(size_t)15) + 32 <= budget && (rnd || R < c.inbox_capacity)) R <<= 1;
  if (!rnd && R > cap_tot) { R = 2; while (R * 2 <= cap_tot) R <<= 1; }
  if (rnd && R > cap_tot) R = cap_tot ? cap_tot : 1;
  dp.S = cap_tot > R ? cap_tot - R : 0;
  // the spill area is spill_capacity x 16 bytes per node: 8-byte ring entries (constant latency) or 12-byte bag entries (RND)
  if ((size_t)dp.S * (rnd ? 12 : 8) > (size_t)c.spill_capacity * 16) return MSIM_LAYOUT_DOES_NOT_FIT;
  if (rnd) MSIM_UPLOAD_ONCE(duo_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  size_t off = stage_bytes;
  dp.off_ring = (u32)off; off += rnd ? (size_t)(kp.N + 1) * DUO_BAG * 8 : (size_t)32 * R * (lat0 ? 4 : 8);
  dp.off_seq = (u32)off; if (rnd) off += (((size_t)(kp.N + 1) * DUO_BAG * 2) + 15) & ~(size_t)15;
  dp.R = R;
  off = (off + 15) & ~(size_t)15;
  dp.off_seen = (u32)off; off += seen_bytes;
  dp.half_bytes = (u32)off;
  dp.off_log2 = (u32)(2 * off);
  dp.deg = duo_degree(c);
  dp.echoback = c.node_program == MSIM_NODE_BCAST_FF_ECHOBACK;
  dp.round_limit = (kp.dev_flags & 0x100u) ? 2000000u : ROUND_LIMIT;
  const size_t lds = 2 * off + (rnd ? 257 * 4 + 12 : 0);
  if (lds > 160 * 1024) return MSIM_LAYOUT_DOES_NOT_FIT;
  const bool deg4 = dp.deg <= 4 && kp.N <= 31;   // (lane 31 must hold no node: unused neighbour slots point at it)
  const dim3 grid((n + 1) / 2);
  if (rnd) return deg4 ? duo_launch<false, true, true>(dp, grid, lds, st) : duo_launch<false, false, true>(dp, grid, lds, st);
  if (lat0) return deg4 ? duo_launch<true, true, false>(dp, grid, lds, st) : duo_launch<true, false, false>(dp, grid, lds, st);
  return deg4 ? duo_launch<false, true, false>(dp, grid, lds, st) : duo_launch<false, false, false>(dp, grid, lds, st);
}
