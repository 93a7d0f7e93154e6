// dt8.hip — EIGHT clusters of the Datomic-style txn-list-append node per wavefront (BASELINE configs[4] over
// demo/ruby/datomic_list_append.rb, the node core.clj:113-114 runs).
//
// Same program and the same rounds as dt_kernel<> (sim_kernel_dt.inc; specification: oracle/dt_nodes.inc — parity with the Ruby program
// unpinned): a persistent hash tree of immutable nodes in lww-kv, the root pointer in lin-kv, a lock per node, lazily loaded paths, path
// copies, save + cas.  What changes is the mapping, as in mk8.hip / txn8.hip: a cluster is n nodes (each with its client) + lin-kv + lww-kv
// = n + 2 <= 8 endpoints, one lane each of an 8-lane group, and a wavefront carries eight clusters.  dt_kernel<> runs one cluster per
// wavefront — 7 live lanes of 64 — and waits for dependent HBM loads (a walk down the tree per micro-op); here one instruction stream
// serves eight clusters and eight walks are in flight.  What is uniform per CLUSTER lives in VGPRs (equal within a group), a "ballot" is
// the group's 8 bits of the wave ballot, another lane's value comes by ds_bpermute within the group, the time reduction is three DPP steps.
//
// Scope (engine.hip picks this kernel when all of it holds, else dt_kernel<> runs): n_nodes <= 6, one worker per node, net journal off,
// launches of at least MSIM_DT8_MIN_CLUSTERS clusters (csrc/layout_thresholds.h; MSIM_DEV_FLAGS bit 10 takes it whatever the launch).
// Measured (profiles/r05_dt8_*): cfg5 x 32768 clusters 760 ms against 1042 ms one per wavefront; 224 registers, two wavefronts per SIMD —
// tighter register budgets (168 / 128) and fewer LDS queue slots were measured and lose (spills in the handlers).
//
// LDS of a wavefront: node / service queues and client inboxes slot-major (slot s of lane e at [s * 64 + e]; RQ / CQ envelopes, the rest
// spills to HBM), per cluster the nodes' transactions (lock holder + waiting queue: D8_CW words each; the save stack lives in HBM scratch),
// the generator's key pool and the nemesis shuffle.  History rows go straight to HBM.  The per-instance scratch is dt_kernel<>'s.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "wave_common.h"
#include "log2_table.h"
#include "layout_thresholds.h"

namespace {

__constant__ u32 m8_log2_q24[257];

constexpr u32 GS = 8u;            // lanes per cluster
#ifndef D8_RQ
#define D8_RQ 8u
#endif
constexpr u32 RQ = D8_RQ;         // LDS envelopes per node / service queue
#ifndef D8_CQ
#define D8_CQ 1u
#endif
constexpr u32 CQ = D8_CQ;         // LDS envelopes per client inbox (0: the inbox lives in its HBM spill alone)
constexpr u32 M8_CLIENT_CAP = 32u;
constexpr u32 V_NIL = 0xFFFFu;
enum { M_WRITE = 14, M_WRITE_OK, M_CAS, M_CAS_OK, M_ERROR, M_TXN = 23, M_TXN_OK = 24 };
enum { S_GEN3 = 3 };
enum { D_LIN = 0, D_LWW = 1 };
// what dt_kernel<> defines (sim_kernel_dt.inc): capacities, stages, the words of a node's transaction — here without the save stack
constexpr u32 DT_WAITQ = 8u, DT_MAXDEPTH = 40u, DT_MAXW = 256u, DT_RW = 12u, DT_AWAIT_US = 5000000u, DT_CASQ = 4u;
enum { DS_IDLE = 0, DS_ROOT, DS_LOAD, DS_SAVE, DS_CAS, DS_INIT_LEAF, DS_INIT_ROOT };
// a node's transaction (the lock holder) in LDS.  DC_J packs the next micro-op, the appends applied so far and the new tree nodes they have
// replaced again (j | appends << 8 | replaced << 16); DC_WQN the waiting transactions (count | ring head << 8; the ring itself is in HBM
// scratch); DC_RC is the record of the working tree's ROOT — kind / range word + eight children — which every micro-op's walk starts from.
enum { DC_STAGE = 0, DC_CMSG, DC_REF, DC_RPC, DC_P1, DC_RV, DC_T, DC_TARGET, DC_PSTART, DC_WLO, DC_WN, DC_WOUT, DC_J, DC_NOWN, DC_OWN /* 8 */, DC_WQN = DC_OWN + 8,
       DC_RC /* 9 */, D8_CW = DC_RC + 9 };
// a node's auxiliary words in HBM scratch (behind the clients' spill): the waiting ring (DT_WAITQ x {client msg, txn ref}), the walked path below
// depth 2, and one position key per tree node the transaction has created (what save! sorts by)
constexpr u32 D8_AUX_WQ = 0u, D8_AUX_PATH = 16u, D8_AUX_KEYS = 64u, D8_MAXNEW = 512u, D8_AUX = D8_AUX_KEYS + D8_MAXNEW + 4u;
constexpr u32 D8_DEAD = 0xFFFFFFFFu;
constexpr u32 D8_NOFIRST = 0xFFFFFFu;   // a key that is in no committed tree yet (above every version)

struct M8Params {
  KParams k;
  u32 n_inst;
  u32 off_cq, off_cur, off_gen, off_misc, off_cs;       // LDS byte offsets (queues at 0)
  u32 node_spill, client_spill;                          // HBM spill entries per node-or-service queue / client inbox
  u64 client_spill_off, stack_off;                       // word offsets inside the per-instance scratch: the clients' spill area, the nodes' auxiliary words (D8_AUX each)
  u32 round_limit;
};

// Tree.hash (:60-64): Zlib.crc32(k.to_s) % RING_SIZE
__device__ __forceinline__ u32 d8_hash(u32 k) {
  u32 dig[5], n = 0;
  do { dig[n++] = k % 10u; k /= 10u; } while (k);
  u32 c = 0xFFFFFFFFu;
  for (u32 i = n; i-- > 0;) {
    c ^= 48u + dig[i];
#pragma unroll
    for (u32 b = 0; b < 8; b++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
  }
  return (~c) & 127u;
}

__device__ __forceinline__ u32 m8_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = m8_log2_q24[idx], l1 = m8_log2_q24[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}
// min over the 8 lanes of the caller's group, in every lane of it
__device__ __forceinline__ u32 m8_oct_min(u32 v) {
  v = min(v, dpp_mov<0xB1, 0xF, 0xF, false>(v, v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_mov<0x4E, 0xF, 0xF, false>(v, v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_mov<0x141, 0xF, 0xF, false>(v, v));  // row_half_mirror
  return v;
}

// ---- apply_txn as a function of its own ---------------------------------------------------------------------------------------------------
// The walk down the tree, the path copies and the write list are the kernel's heaviest code and need registers of their own (three records' worth
// of children in flight); inlined into the round loop their pressure met the loop's own state and the kernel sat at 224-256 registers — one or
// two wavefronts per SIMD for a kernel that waits for dependent HBM loads.  As a separate function the body keeps only what it touches in
// registers (the recipe of sim_kernel_colo.inc's quiet_run<>): the lane's state it reads and writes crosses the call by value, the run's constants
// come as scalars-in-vector-registers and are made scalar again, LDS and the instance's scratch are addressed from the lane id and the instance.
struct D8ApplyIO {
  u32 np, mid, my_flags, wait_until;                       // per lane: the node (@ptr, msg_id of its last RPC, ..)
  u32 n_out, o_dest, o1_type, o1_a, o1_b, o_wlo;           // what the node sends after this step
  u32 done;                                                // bit 0: answer txn_ok, bit 1: unlock
};
struct D8ApplyK {   // constants of the launch
  u32 *scratch; u32 *payload; u64 scratch_words; u32 max_pay, N, TC, mv, mw, off_aux, off_cur, off_gen;
};
#ifdef D8_APPLY_NOINLINE   // (developer A/B: as a function of its own the call costs more than it saves — 1039 ms against 675 per 32768 clusters, profiles/r06_dt8_variants.jsonl)
#define D8_APPLY_ATTR __attribute__((noinline))
#else
#define D8_APPLY_ATTR __forceinline__
#endif
__device__ D8_APPLY_ATTR D8ApplyIO d8_apply(D8ApplyIO io, const D8ApplyK kc, u32 inst, u32 lane, u32 T_in, bool active) {   // (called by the whole wavefront: the constants are made scalar with every lane in)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#define UNI(x) ((u32)__builtin_amdgcn_readfirstlane((int)(x)))
  const u32 N = UNI(kc.N), TC = UNI(kc.TC), mv = UNI(kc.mv), mw = UNI(kc.mw), max_pay = UNI(kc.max_pay);
  u32 *const scratch0 = reinterpret_cast<u32 *>(((u64)UNI((u32)((u64)kc.scratch >> 32)) << 32) | UNI((u32)(u64)kc.scratch));
  u32 *const payload0 = reinterpret_cast<u32 *>(((u64)UNI((u32)((u64)kc.payload >> 32)) << 32) | UNI((u32)(u64)kc.payload));
  const u64 scratch_words = ((u64)UNI((u32)(kc.scratch_words >> 32)) << 32) | UNI((u32)kc.scratch_words);
  const u32 OFF_AUX = UNI(kc.off_aux), off_cur = UNI(kc.off_cur), off_gen = UNI(kc.off_gen);
#undef UNI
  const u32 l = lane & 7u, grp = lane >> 3;
  const u32 OFF_KVN = mv * mw, OFF_FIRST = OFF_KVN + mv, OFF_HASHW = OFF_FIRST + mv, OFF_REC = (OFF_HASHW + (mv + 3u) / 4u + 3u) & ~3u;   // (records are read and written 16 bytes at a time)
  const u32 OFF_WL = OFF_REC + N * TC * DT_RW;
  (void)OFF_KVN;
  u32 *const g_scr = scratch0 + (size_t)inst * scratch_words;
  const u32 *const g_pay = payload0 + (size_t)inst * max_pay;
  const u32 *const g_first = g_scr + OFF_FIRST;   // per key: the version at which it entered the tree << 8 | Tree.hash (one word, one request)
  (void)OFF_HASHW;
  u32 *const g_rec = g_scr + OFF_REC;
  u32 *const aux = g_scr + OFF_AUX + l * D8_AUX;
  u32 *const my_wl = g_scr + OFF_WL + l * DT_MAXW;
  u32 *const cu = reinterpret_cast<u32 *>(smem + off_cur) + (grp * N + l) * D8_CW;
  const u32 *const gen = reinterpret_cast<const u32 *>(smem + off_gen) + grp * 36;
  const u32 T = T_in;
  u32 next_p = io.np, node_msgid = io.mid, my_flags = io.my_flags, wait_until = io.wait_until;
  u32 n_out = io.n_out, o_dest = io.o_dest, o1_type = io.o1_type, o1_a = io.o1_a, o1_b = io.o1_b, o_wlo = io.o_wlo;
  bool do_reply_ok = false, do_unlock = false;
  auto rec_of = [&](u32 ptr) -> u32 * { return g_rec + ((size_t)(ptr >> 20) * TC + (ptr & 0xFFFFFu)) * DT_RW; };
  auto send1 = [&](u32 dest, u32 type, u32 a, u32 b) { o_dest = dest; n_out = 1; o1_type = type; o1_a = a; o1_b = b; };
  auto br_index = [&](u32 w0, u32 h) -> u32 {   // branch_index (:231-247) with the split's bounds (:170-181)
    const u32 lo = (w0 >> 8) & 0xFFu, hi = (w0 >> 16) & 0xFFu, bs = (hi - lo) / 8u;
    u32 r = 7u;
#pragma unroll
    for (u32 i = 7u; i-- > 0u;) r = h < lo + (i + 1u) * bs ? i : r;
    return r;
  };
  auto load = [&](u32 ptr) {   // Tree.load with a cache miss (:83-101)
    const u32 rid = ++node_msgid;
    cu[DC_STAGE] = DS_LOAD; cu[DC_TARGET] = ptr; cu[DC_RPC] = rid;
    send1(D_LWW, M_READ, ptr, rid);
    wait_until = T + DT_AWAIT_US;
  };
  // Where a tree node sits decides when save! writes it: children before their parent, siblings by child index (:291-320) — ascending
  // in this key.  Positions are (c0, c1, then the chain): a 128-wide root splits into 16-wide ranges, those into 2-wide ones, and a
  // 2-wide range can only hand everything to its LAST child (branch_index, :231-247), so below depth 2 a path is all sevens: the spine
  // node at depth 2 + m gets 1023 - m (deeper first), its seven empty siblings 8 m + i (before every spine node).  An absent digit
  // (the node IS the c0 / c1 subtree's root) is 8 / 1023: after everything below it.
  auto poskey = [&](u32 d, u32 c0, u32 c1) -> u32 { return d == 0u ? ((8u << 14) | (8u << 10) | 1023u) : d == 1u ? ((c0 << 14) | (8u << 10) | 1023u) : ((c0 << 14) | (c1 << 10) | (1023u - (d - 2u))); };
  auto childkey = [&](u32 n, u32 i, u32 c0, u32 c1) -> u32 {   // child i of the node at depth n of that path
    if (n == 0u) return (i << 14) | (8u << 10) | 1023u;
    if (n == 1u) return (c0 << 14) | (i << 10) | 1023u;
    return (c0 << 14) | (c1 << 10) | (i == 7u ? 1023u - (n - 1u) : (n - 1u) * 8u + i);
  };
  // save! (:212-224, :291-320): the new tree nodes the final tree reaches, children before their parent.  Nothing is walked: every node the
  // transaction created left its position key in aux[], a node whose position a later copy took is marked dead there, and a live node's
  // place in the write list is the number of live keys below its own.  One append alone: creation order IS that order (leaf level first, then upwards).
  auto save = [&](u32 napp, u32 ndead) {
    const u32 pstart = cu[DC_PSTART], M = next_p + 1u - pstart, wlo = node_msgid + 1u;
    u32 wn = M - ndead;
    if (wn > DT_MAXW) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; wn = DT_MAXW; }
    if (napp == 1u) {
      for (u32 i = 0; i < wn; i++) my_wl[i] = (l << 20) | (pstart + i);
    } else {
      u32 *const nk = aux + D8_AUX_KEYS;
      if (M <= 16u) {   // the usual case (two to four appends of three or four tree nodes each): every key in registers after ONE round trip, ranks by comparison
        const uint4 *const q = reinterpret_cast<const uint4 *>(nk);
        const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        u32 kk[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
        for (u32 i = 0; i < 16u; i++) kk[i] = i < M ? kk[i] : D8_DEAD;   // (what lies behind the transaction's keys is stale)
#pragma unroll
        for (u32 i = 0; i < 16u; i++) {
          u32 rank = 0;
#pragma unroll
          for (u32 j = 0; j < 16u; j++) rank += kk[j] < kk[i] ? 1u : 0u;
          if (kk[i] != D8_DEAD && rank < DT_MAXW) my_wl[rank] = (l << 20) | (pstart + i);
        }
      } else {   // (a load per key and four keys per load: a wait each — only for transactions that create more than sixteen tree nodes)
        nk[M] = D8_DEAD; nk[M + 1u] = D8_DEAD; nk[M + 2u] = D8_DEAD;   // (the rank loop reads four keys at a time)
        for (u32 i = 0; i < M; i++) {
          const u32 ki = nk[i];
          if (ki == D8_DEAD) continue;
          u32 rank = 0;
          for (u32 j4 = 0; j4 < M; j4 += 4u) {
            const uint4 q = *reinterpret_cast<const uint4 *>(nk + j4);
            rank += (q.x < ki ? 1u : 0u) + (q.y < ki ? 1u : 0u) + (q.z < ki ? 1u : 0u) + (q.w < ki ? 1u : 0u);
          }
          if (rank < DT_MAXW) my_wl[rank] = (l << 20) | (pstart + i);
        }
      }
    }
    node_msgid += wn;
    cu[DC_STAGE] = DS_SAVE; cu[DC_WLO] = wlo; cu[DC_WN] = wn; cu[DC_WOUT] = wn;
    o_dest = D_LWW; n_out = wn; o_wlo = wlo;
    wait_until = T + DT_AWAIT_US;   // `tree2.save!.await` (:366)
  };
  // apply_txn (:391-415) from micro-op j on; stops at the first tree node that has to be fetched.  ONE walk per micro-op: t[k] (:231-247, for an
  // append too: `t[k].clone`, :405) goes from the root record in LDS down through one record per level — kind / range, key count, "loaded by"
  // flags and the eight children in one round trip — and leaves behind what assoc (:158-197, :256-268) needs: the leaf's record, the path's
  // pointers (depth 1 and 2 in registers, a chain's below that in aux[]) and its child indices; the copies of the branches above are then read
  // at known addresses (independent loads) and written with the new pointers, which are known before anything is written: new pointers go
  // leaf first, then upwards (new_ptr, :352-355) — with n branches above a leaf level of L new nodes (1, or 8 leaves + their branch) the leaf
  // level takes base+1 .. base+L, the branch at depth i base+L+(n-i).
  auto apply = [&]() {
    const u32 ref = cu[DC_REF], off0 = ref & 0xFFFFFFu, nm = ref >> 24;
    const u32 rv = cu[DC_RV], pstart = cu[DC_PSTART];
    const u32 jw = cu[DC_J];
    u32 j = jw & 0xFFu, napp = (jw >> 8) & 0xFFu, ndead = jw >> 16;
    u32 troot = cu[DC_T];
    u32 *const nk = aux + D8_AUX_KEYS;
    auto is_new = [&](u32 ptr) -> bool { return (ptr >> 20) == l && (ptr & 0xFFFFFu) >= pstart; };
    u32 wq[4] = {0, 0, 0, 0}, hfq[4] = {0, 0, 0, 0}, qbase = 0xFFFFFFF0u;   // four micro-ops and their keys' words per pair of round trips
    while (j < nm) {
      if (j - qbase >= 4u) {
        qbase = j;
#pragma unroll
        for (u32 t = 0; t < 4u; t++) wq[t] = g_pay[off0 + min(j + t, nm - 1u)];
#pragma unroll
        for (u32 t = 0; t < 4u; t++) hfq[t] = g_first[(wq[t] >> 1) & 0x7FFFu];
      }
      const u32 tq = j - qbase;
      const u32 w = tq == 0u ? wq[0] : tq == 1u ? wq[1] : tq == 2u ? wq[2] : wq[3];
      const u32 hf = tq == 0u ? hfq[0] : tq == 1u ? hfq[1] : tq == 2u ? hfq[2] : hfq[3];
      const u32 k = (w >> 1) & 0x7FFFu, h = hf & 0xFFu, first = hf >> 8;
      u32 d = 0, pt = troot, c0 = 0, c1 = 0, pa = 0, pb = 0;
      u32 w0 = cu[DC_RC], w1 = 0;
      bool miss = false, deep = false;
      if ((w0 & 1u) == 0u) w1 = rec_of(pt)[1];   // the root is a leaf (the first few transactions of a run): its key count
      else {
        c0 = br_index(w0, h);
        pt = cu[DC_RC + 1u + c0]; d = 1u;
        for (;;) {
          const u32 *const r = rec_of(pt);
          const u32 w3 = __hip_atomic_load(r + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (word 3 is the one word of a record that changes after its creation, by L2 atomics: read past the L1)
          // a record is three 16-byte accesses (kind / range, key count, version, -; children 0-3; children 4-7): what a CU's L1 serves is
          // requests, not bytes — at full occupancy the kernel sits on that, so a record is never read word by word
          const uint4 rq0 = reinterpret_cast<const uint4 *>(r)[0], rq1 = reinterpret_cast<const uint4 *>(r)[1], rq2 = reinterpret_cast<const uint4 *>(r)[2];
          w0 = rq0.x; w1 = rq0.y;
          const u32 ch[8] = {rq1.x, rq1.y, rq1.z, rq1.w, rq2.x, rq2.y, rq2.z, rq2.w};
          if (!is_new(pt) && !((w3 >> (2u + l)) & 1u)) { miss = true; break; }   // neither created by this transaction nor loaded by this node
          if (d >= DT_MAXDEPTH) { deep = true; break; }   // engine capacity
          if ((w0 & 1u) == 0u) break;   // the key's leaf
          const u32 ci = br_index(w0, h);
          if (d == 1u) { c1 = ci; pa = pt; } else if (d == 2u) pb = pt; else aux[D8_AUX_PATH + d] = pt;
          pt = ch[0];
#pragma unroll
          for (u32 c = 1; c < 8u; c++) pt = c == ci ? ch[c] : pt;
          d++;
        }
      }
      if (miss) { cu[DC_J] = j | (napp << 8) | (ndead << 16); cu[DC_T] = troot; load(pt); return; }
      if (deep) my_flags |= MSIM_FLAG_ARENA_OVERRUN;
      else if (w & 1u) {   // assoc
        const u32 n = d, lw0 = w0, lcount = w1;
        bool has = first <= rv;   // (D8_NOFIRST is above every version)
        { const u32 no = cu[DC_NOWN]; for (u32 i = 0; i < no; i++) has = has || cu[DC_OWN + i] == k; }
        const u32 L = (has || lcount < 8u) ? 1u : 9u, base = next_p, ver = rv + 1u;
        if (base + L + n >= TC || base + L + n - pstart >= D8_MAXNEW) my_flags |= MSIM_FLAG_ARENA_OVERRUN;   // engine capacity
        else {
          auto put = [&](u32 idx, u32 pw0, u32 cnt, u32 key) -> u32 * {   // (word 3: lww-kv replica in bits 0-1, 3 = not written; bit 2 + i: node i has loaded it)
            u32 *const r = g_rec + ((size_t)l * TC + idx) * DT_RW; *reinterpret_cast<uint4 *>(r) = make_uint4(pw0, cnt, ver, 3u);
            nk[idx - pstart] = key; return r; };
          auto put_children = [&](u32 *nr, uint4 q1, uint4 q2, u32 ci, u32 child_new) {   // eight children, the one at ci replaced: two 16-byte stores
            q1.x = ci == 0u ? child_new : q1.x; q1.y = ci == 1u ? child_new : q1.y; q1.z = ci == 2u ? child_new : q1.z; q1.w = ci == 3u ? child_new : q1.w;
            q2.x = ci == 4u ? child_new : q2.x; q2.y = ci == 5u ? child_new : q2.y; q2.z = ci == 6u ? child_new : q2.z; q2.w = ci == 7u ? child_new : q2.w;
            reinterpret_cast<uint4 *>(nr)[1] = q1; reinterpret_cast<uint4 *>(nr)[2] = q2; };
          auto dies = [&](u32 ptr) { if (is_new(ptr)) { nk[(ptr & 0xFFFFFu) - pstart] = D8_DEAD; ndead++; } };   // its position is taken by a node created now
          // the copies' sources at depth 1 and 2, at known addresses: in flight together
          u32 wa = 0, wb = 0;
          uint4 qa1 = make_uint4(0, 0, 0, 0), qa2 = qa1, qb1 = qa1, qb2 = qa1;
          if (n >= 2u) { const uint4 *const r = reinterpret_cast<const uint4 *>(rec_of(pa)); wa = r[0].x; qa1 = r[1]; qa2 = r[2]; }
          if (n >= 3u) { const uint4 *const r = reinterpret_cast<const uint4 *>(rec_of(pb)); wb = r[0].x; qb1 = r[1]; qb2 = r[2]; }
          dies(pt);
          const u32 lo = (lw0 >> 8) & 0xFFu, hi = (lw0 >> 16) & 0xFFu;
          if (L == 1u) put(base + 1u, lw0, lcount + (has ? 0u : 1u), poskey(n, c0, c1));
          else {   // eight leaves under a new branch: the lineage's keys of this range (and the new one) by sub-range
            const u32 bs = (hi - lo) / 8u, nkeys = gen[32];
            u64 c_lo = 0, c_hi = 0;   // 4 x 16-bit counters each
            for (u32 q0 = 0; q0 < nkeys; q0 += 4u) {   // four keys per round trip
              u32 f4[4], h4[4];
#pragma unroll
              for (u32 t = 0; t < 4u; t++) { const u32 q = min(q0 + t, nkeys - 1u); const u32 hf = g_first[q]; f4[t] = hf >> 8; h4[t] = hf & 0xFFu; }
#pragma unroll
              for (u32 t = 0; t < 4u; t++) {
                const u32 q = q0 + t;
                if (q >= nkeys) continue;
                bool in = q == k || f4[t] <= rv;
                { const u32 no = cu[DC_NOWN]; for (u32 i = 0; i < no; i++) in = in || cu[DC_OWN + i] == q; }
                const u32 hq = h4[t];
                if (!in || hq < lo || hq >= hi) continue;
                const u32 ci = bs ? min((hq - lo) / bs, 7u) : 7u;
                if (ci < 4u) c_lo += 1ull << (16u * ci); else c_hi += 1ull << (16u * (ci - 4u));
              }
            }
            u32 *const br = put(base + 9u, 1u | (lo << 8) | (hi << 16), 0u, poskey(n, c0, c1));
            for (u32 i = 0; i < 8u; i++) {
              const u32 b_lo = lo + i * bs, b_hi = i == 7u ? hi : b_lo + bs;
              const u32 cnt = (u32)((i < 4u ? c_lo >> (16u * i) : c_hi >> (16u * (i - 4u))) & 0xFFFFu);
              put(base + 1u + i, (b_lo << 8) | (b_hi << 16), cnt, childkey(n, i, c0, c1));
            }
            { const u32 k0 = (l << 20) | (base + 1u);
              reinterpret_cast<uint4 *>(br)[1] = make_uint4(k0, k0 + 1u, k0 + 2u, k0 + 3u); reinterpret_cast<uint4 *>(br)[2] = make_uint4(k0 + 4u, k0 + 5u, k0 + 6u, k0 + 7u); }
            if (n == 0u) { cu[DC_RC] = 1u | (lo << 8) | (hi << 16); for (u32 i = 0; i < 8u; i++) cu[DC_RC + 1u + i] = (l << 20) | (base + 1u + i); }   // the new root is this branch
          }
          // a copy of every branch above, pointing at the new child (:256-268)
          for (u32 i = n; i-- > 0u;) {
            const u32 child_new = (l << 20) | (i + 1u == n ? base + L : base + L + (n - i - 1u));
            const u32 idx = base + L + (n - i);
            if (i == 0u) {   // the root: its record is in LDS, and stays there as the new root's
              dies(troot);
              const u32 rw0 = cu[DC_RC];
              u32 *const nr = put(idx, rw0, 0u, poskey(0u, c0, c1));
              cu[DC_RC + 1u + c0] = child_new;
              reinterpret_cast<uint4 *>(nr)[1] = make_uint4(cu[DC_RC + 1u], cu[DC_RC + 2u], cu[DC_RC + 3u], cu[DC_RC + 4u]);
              reinterpret_cast<uint4 *>(nr)[2] = make_uint4(cu[DC_RC + 5u], cu[DC_RC + 6u], cu[DC_RC + 7u], cu[DC_RC + 8u]);
            } else if (i == 1u) {
              dies(pa);
              u32 *const nr = put(idx, wa, 0u, poskey(1u, c0, c1));
              put_children(nr, qa1, qa2, c1, child_new);
            } else if (i == 2u) {
              dies(pb);
              const u32 ci = br_index(wb, h);
              u32 *const nr = put(idx, wb, 0u, poskey(2u, c0, c1));
              put_children(nr, qb1, qb2, ci, child_new);
            } else {   // a chain below depth 2: one link at a time
              const u32 pp = aux[D8_AUX_PATH + i];
              dies(pp);
              const uint4 *const r = reinterpret_cast<const uint4 *>(rec_of(pp));
              const uint4 x0 = r[0], x1 = r[1], x2 = r[2];
              const u32 xw0 = x0.x, ci = br_index(xw0, h);
              u32 *const nr = put(idx, xw0, 0u, poskey(i, c0, c1));
              put_children(nr, x1, x2, ci, child_new);
            }
          }
          if (n == 0u && L == 1u) { /* the root stays a leaf: its record in LDS (kind / range) is unchanged */ }
          next_p = base + L + n;
          troot = (l << 20) | next_p;
          napp++;
          if (!has) { const u32 no = cu[DC_NOWN]; if (no < 8u) { cu[DC_OWN + no] = k; cu[DC_NOWN] = no + 1u; } }
        }
      }
      j++;
    }
    cu[DC_J] = j | (napp << 8) | (ndead << 16); cu[DC_T] = troot;
    if (troot == cu[DC_P1]) { do_reply_ok = true; do_unlock = true; return; }   // nothing appended: no write, no cas
    save(napp, ndead);
  };
  if (active) apply();
  io.np = next_p; io.mid = node_msgid; io.my_flags = my_flags; io.wait_until = wait_until;
  io.n_out = n_out; io.o_dest = o_dest; io.o1_type = o1_type; io.o1_a = o1_a; io.o1_b = o1_b; io.o_wlo = o_wlo;
  io.done = (do_reply_ok ? 1u : 0u) | (do_unlock ? 2u : 0u);
  return io;
}

#ifndef D8_WAVES_PER_EU   // a register budget for that many wavefronts per SIMD.  Two: left alone the allocator takes 256 + a few accumulator registers for the
#define D8_WAVES_PER_EU 2 // nemesis variants and ONE wavefront per SIMD is what remains (1180 ms per 32768 clusters instead of 616); three / four spill in the round loop and lose
#endif
#define D8_OCC __attribute__((amdgpu_waves_per_eu(D8_WAVES_PER_EU)))
template <bool NEM, bool NET_RANDOM>
__global__ void __launch_bounds__(64) D8_OCC dt8_kernel(const M8Params tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KParams &p = tp.k;
  const u32 lane = threadIdx.x, l = lane & (GS - 1u), grp = lane >> 3, gbase = lane & 56u;
  const u32 N = p.N;
  const bool is_node = l < N, is_lin = l == N;
  const u32 LIN = 2 * N;   // endpoint index of lin-kv (lane N of the group); lww-kv is LIN + 1 (lane N + 1)
  const u32 inst_raw = blockIdx.x * 8u + grp;
  const bool real = inst_raw < tp.n_inst;
  const u32 inst = real ? inst_raw : tp.n_inst - 1u;
  const u64 key = mix64(p.cfg.seed + 0x9E3779B97F4A7C15ull * (p.first_instance + inst + 1));
  const u32 lt = (1u << l) - 1u;
  const u32 all_nodes = (1u << N) - 1u;
  const u32 max_rows = p.cfg.max_rows, max_pay = p.cfg.max_payload_words;
  const u32 p_loss = p.cfg.p_loss_q32, lat_mean = p.cfg.latency_mean_ms, lat_dist = p.cfg.latency_dist;
  const u32 rate = p.cfg.rate_mhz, mw = p.cfg.max_writes_per_key, mv = p.cfg.max_values;
  const u32 TC = p.mk_tcap;   // tree nodes a node may create
  const u32 round_limit = tp.round_limit;

  // A register diet for four wavefronts per SIMD (128 registers): nothing that can be recomputed is carried through the round loop.  Every
  // address in an instance's slabs is the slab's base (a kernel argument: scalar registers) + the instance's offset, formed where it is used from
  // `inst` behind an optimization barrier (loop-invariant code motion would otherwise hold a dozen 64-bit pointers per lane for the whole run);
  // the offsets of the regions inside an instance's scratch (dt_kernel<>'s layout, sim_kernel_dt.inc) are the same for every cluster: scalars.
  const u32 OFF_KVN = mv * mw, OFF_FIRST = OFF_KVN + mv, OFF_HASHW = OFF_FIRST + mv, OFF_REC = (OFF_HASHW + (mv + 3u) / 4u + 3u) & ~3u;   // (records are read and written 16 bytes at a time)
  const u32 OFF_WL = OFF_REC + N * TC * DT_RW, OFF_CAS = OFF_WL + N * DT_MAXW;
  const u32 OFF_SPILL = (u32)p.spill_off, OFF_CSPILL = (u32)tp.client_spill_off, OFF_AUX = (u32)tp.stack_off;
  const u32 qlane = l <= N + 1u ? l : 0u;
  const u32 my_spill_cap = l <= N + 1u ? tp.node_spill : 0u;
  const u32 my_node = is_node ? l : 0u;
#define D8_INST ([&]() -> size_t { u32 i_ = inst; MSIM_OPAQUE(i_); return (size_t)i_; }())
#define g_rows (p.rows + D8_INST * max_rows)
#define g_pay (p.payload + D8_INST * max_pay)
#define g_scr (p.scratch + D8_INST * p.scratch_words)
#define g_kv g_scr                                                  /* [max_values][mw]: element | version << 8 */
#define g_kvn (g_scr + OFF_KVN)                                     /* [max_values] */
#define g_first (g_scr + OFF_FIRST)                                 /* [max_values] version at which the key entered the tree (D8_NOFIRST: never) << 8 | Tree.hash of the key: what a walk needs of a key in ONE word (dt_kernel<> keeps the hashes in a byte array behind this one) */
#define g_rec (g_scr + OFF_REC)                                     /* [N][TC][DT_RW] tree nodes by pointer */
#define g_wl (g_scr + OFF_WL)                                       /* [N][DT_MAXW] the pointers a node writes this round */
#define g_cas (g_scr + OFF_CAS)                                     /* [N][DT_CASQ] x {msg_id, from, transaction}: what a node's cas requests carry beside `to` */
#define my_spill (reinterpret_cast<uint4 *>(g_scr + OFF_SPILL) + qlane * tp.node_spill)
#define my_cspill (reinterpret_cast<uint4 *>(g_scr + OFF_CSPILL) + my_node * tp.client_spill)
#define aux (g_scr + OFF_AUX + my_node * D8_AUX)
#define my_wl (g_scr + OFF_WL + my_node * DT_MAXW)

  uint4 *const my_q = reinterpret_cast<uint4 *>(smem) + lane;                                   // node / service queue: slot s at my_q[s * 64]
  uint4 *const my_cq = reinterpret_cast<uint4 *>(smem + tp.off_cq) + lane;                      // client inbox
  u32 *const curs_g = reinterpret_cast<u32 *>(smem + tp.off_cur) + grp * (N * D8_CW);           // [node of the group][D8_CW]
  u32 *const gen = reinterpret_cast<u32 *>(smem + tp.off_gen) + grp * 36;                       // active[16], next_val[16], next_key
  u32 *const misc = reinterpret_cast<u32 *>(smem + tp.off_misc) + grp * GS;
  u32 *const cu = curs_g + my_node * D8_CW;

  for (u32 i = lane; i < 8 * N * D8_CW; i += 64) reinterpret_cast<u32 *>(smem + tp.off_cur)[i] = 0;
  for (u32 i = l; i < 16; i += GS) { gen[i] = i; gen[16 + i] = 1; }
  if (l == 0) gen[32] = p.cfg.key_count;
  if (real) {
    for (u32 i = l; i < mv; i += GS) { g_kvn[i] = 0; g_first[i] = (D8_NOFIRST << 8) | d8_hash(i); }
    for (u32 i = l; i < N * DT_CASQ * 3u; i += GS) g_cas[i] = 0;
  }
  __syncthreads();

  auto GB = [&](bool pred) -> u32 { return (u32)(__ballot(pred) >> gbase) & 0xFFu; };            // the cluster's slice of a ballot
  auto GGET = [&](u32 v, u32 s) -> u32 { return (u32)__builtin_amdgcn_ds_bpermute((int)((gbase + s) << 2), (int)v); };   // v of lane s of my group

  // ---- node / service state ----
  u32 deliver_at = INF; uint4 cm = make_uint4(0, 0, 0, 0);
  u32 in_n = 0, sp_n = 0, part = 0;
  u32 wait_until = INF;                                // node: when the lock holder's Promise#await gives up (promise.rb:5,17-30), INF: not waiting
  // a lane is a node or a service, never both: what only one of them keeps shares a register with what only the other keeps
  u32 ra = 0, rb = 0, rc = 0;
#define next_p ra       /* node: @ptr (:332, :352-355) */
#define root ra         /* lin-kv lane: the root pointer */
#define casn rb         /* node: cas requests so far */
#define cur_v rb        /* lin-kv lane: versions so far */
#define node_msgid rc   /* node: msg_id of its last RPC */
#define svc_ctr rc      /* lww-kv lane: rand-int draws so far */
  u32 root_exists = 0;                                 // lin-kv lane
  // ---- client state ----
  bool busy = false, mark = false; u32 kind = K_NONE;
  u32 want = 0, timeout_at = 0, next_msg_id = 0, c_value = 0, process = l, m_value = 0, cin_n = 0, csp_n = 0;
  u32 s_send_cl = 0, s_send_sv = 0, s_recv_cl = 0, s_recv_sv = 0, my_flags = 0;
  // ---- per-cluster state (uniform within a group) ----
  u32 T = 0, phase = PH_INIT, next_id = 0, n_rows = 0, n_payload = 0;
  // ... and what a round only looks at briefly lives in LDS, one copy per cluster (written by the cluster's lane 0, or or-ed in): rounds so far,
  // the end of the main phase, the generator's and the nemesis' next times and counters, "losses are on", the instance's flags
  enum { CS_ROUNDS = 0, CS_CUTOFF, CS_GEN_NEXT, CS_NEM_NEXT, CS_GEN_K, CS_NEM_J, CS_LOSS_ON, CS_FLAGS, CS_WORDS };
  u32 *const cs = reinterpret_cast<u32 *>(smem + tp.off_cs) + grp * CS_WORDS;
  if (l < CS_WORDS) cs[l] = 0;
  wave_lds_fence();
#define CFLAG(x_) atomicOr(&cs[CS_FLAGS], (x_))
  bool alive = real;

  auto q_push = [&](const uint4 m) __attribute__((always_inline)) {
    if (in_n < RQ) { my_q[in_n * 64u] = m; in_n++; return; }
    if (sp_n < my_spill_cap) { my_spill[sp_n] = m; sp_n++; return; }
    my_flags |= MSIM_FLAG_INBOX_OVERFLOW;
  };
  // an envelope for THIS lane's node/service arrives (net.clj:189-221)
  auto arrive = [&](u32 id, u32 type, u32 a, u32 b, u32 src) __attribute__((always_inline)) {
    u32 lat = 0;
    if (src < N || src >= LIN) {  // neither end is a client
      if (!NET_RANDOM || lat_dist == MSIM_LAT_CONSTANT) lat = lat_mean;
      else if (lat_dist == MSIM_LAT_UNIFORM) lat = scale32(draw32(key, S_LATENCY, id), 2 * lat_mean);
      else lat = (u32)(((u64)lat_mean * m8_neg_ln_q16(draw32(key, S_LATENCY, id))) >> 16);
    }
    if (NET_RANDOM && p_loss && cs[CS_LOSS_ON] && draw32(key, S_LOSS, id) < p_loss) return;
    q_push(make_uint4(T + lat * 1000u, (id << 8) | type, a, b | (src << 24)));
  };
  auto try_commit = [&](const uint4 e) __attribute__((always_inline)) {
    const u32 src = e.w >> 24;
    if (NEM && src < N && ((part >> src) & 1)) return;  // partitioned (node <-> node only; never happens in this program)
    cm = e;
    deliver_at = e.x <= T ? T : T + ((e.x - T) / 1000u) * 1000u;  // (Thread/sleep (long dt)) net.clj:236-238
  };
  auto poll = [&]() __attribute__((always_inline)) {
    while (alive && l <= N + 1u && deliver_at == INF && (in_n | sp_n) != 0) {
      u32 best = 0; bool in_spill = false;
      uint2 bk = make_uint2(INF, INF);
      for (u32 i = 0; i < in_n; i++) {
        const uint2 kk = *reinterpret_cast<const uint2 *>(&my_q[i * 64u]);
        if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; }
      }
      uint4 *const spl = my_spill;
      for (u32 i0 = 0; i0 < sp_n; i0 += 8) {   // eight spilled keys per round trip
        uint2 k8[8];
#pragma unroll
        for (u32 t = 0; t < 8; t++) k8[t] = *reinterpret_cast<const uint2 *>(&spl[min(i0 + t, sp_n - 1u)]);
#pragma unroll
        for (u32 t = 0; t < 8; t++) {
          const uint2 kk = k8[t];
          if (i0 + t < sp_n && (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y))) { bk = kk; best = i0 + t; in_spill = true; }
        }
      }
      uint4 e;
      if (in_spill) { e = spl[best]; sp_n--; if (best != sp_n) spl[best] = spl[sp_n]; }
      else { e = my_q[best * 64u]; in_n--; if (best != in_n) my_q[best * 64u] = my_q[in_n * 64u]; }
      try_commit(e);
    }
  };
#ifdef D8_PROF   // developer build (tools/variant_lib.sh d8prof dt8.hip -DD8_PROF; tools/dt8_prof_report.py): cycle counters of the round's sections -> the meta of the wavefront's first three clusters
  u64 pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u32 wave_rounds = 0;
  u64 tprev = __builtin_readcyclecounter();
#define M8_MARK(i) { const u64 now_ = __builtin_readcyclecounter(); pacc[i] += now_ - tprev; tprev = now_; }
#ifdef D8_PROF2   // the finer split of tools/dt8_prof_report.py --fine: 0 top .. R2, 1 handlers, 2 apply_txn, 3 answer + unlock, 4 payload, 5 arrivals, 6 poll, 7 R4 + rows
#define M8_MARK2(i) { const u64 now_ = __builtin_readcyclecounter(); pacc[i] += now_ - tprev; tprev = now_; }
#undef M8_MARK
#define M8_MARK(i)
#else
#define M8_MARK2(i)
#endif
#else
#define M8_MARK(i)
#define M8_MARK2(i)
#endif
  for (;;) {
    if (!__ballot(alive)) break;
#ifdef D8_PROF
    wave_rounds++;
#endif
    const u32 busy_mask = GB(busy);

    // ---- time-free phase transitions ----
    // (every lane reads a shared word before a wave-wide operation, lane 0 of the cluster writes it after one: lanes of a wavefront run in step,
    // but the rule also holds for the host emulator's lanes, which only meet at such operations)
    u32 cutoff = cs[CS_CUTOFF], gen_next = cs[CS_GEN_NEXT], nem_next = NEM ? cs[CS_NEM_NEXT] : 0u;
    const u32 rounds_before = cs[CS_ROUNDS];
    if (__ballot(alive && !(phase == PH_MAIN && ((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff))))) {
      for (;;) {
        bool ch = false;
        if (alive) {
          if (phase == PH_INIT_WAIT && !busy_mask) { phase = PH_MAIN_START; ch = true; }
          if (phase == PH_MAIN_START) { cutoff = T + p.cfg.time_limit_ms * 1000u; gen_next = T; nem_next = T; next_msg_id = 0; phase = PH_MAIN; ch = true;
            if (l == 0) { cs[CS_CUTOFF] = cutoff; cs[CS_GEN_NEXT] = T; cs[CS_NEM_NEXT] = T; cs[CS_LOSS_ON] = 1u; } }
          if (phase == PH_MAIN && !((rate > 0 && gen_next < cutoff) || (NEM && nem_next < cutoff)) && !(rate == 0 && T < cutoff)) { phase = PH_DRAIN; ch = true; }
          if (phase == PH_DRAIN && !busy_mask) { phase = PH_DONE; ch = true; }   // no final phase (txn_list_append.clj:142)
        }
        if (!__ballot(ch)) break;
      }
      if (phase == PH_DONE) alive = false;
      if (!__ballot(alive)) break;
    }
    if (alive) { const u32 r = rounds_before + 1u; if (l == 0) cs[CS_ROUNDS] = r; if (r > round_limit) { CFLAG(MSIM_FLAG_ROUND_LIMIT); alive = false; } }
    if (GB((my_flags & MSIM_FLAG_ARENA_OVERRUN) != 0)) alive = false;   // an engine capacity was exceeded: what follows would not be the program's behaviour

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
      const bool none_due = GB(deliver_at <= T || wait_until <= T) == 0;
      const bool jump = alive && due > T && none_due;
      if (__ballot(jump)) {
        u32 k = deliver_at == INF ? INF : deliver_at * 2;
        if (wait_until != INF) k = min(k, wait_until * 2);   // (a node's timer is a normal event)
        if (busy) k = min(k, timeout_at * 2 + 1);
        u32 km = m8_oct_min(k);
        if (due != INF) km = min(km, due * 2);
        if (jump) {
          if (km == INF) { CFLAG(MSIM_FLAG_ROUND_LIMIT); alive = false; }
          else { timeout_round = (km & 1) != 0; T = max(T, km >> 1); }
        }
      }
    }

    bool inv_row = false; u32 inv_packed = 0, inv_value = 0, inv_len = 0;
    bool cmp_row = false; u32 cmp_packed = 0, cmp_value = 0, cmp_len = 0;
    u32 nem_rows = 0, nem_f = 0, nem_v1 = 0, nem_v2 = 0, nem_len2 = 0;

    auto complete = [&](u32 type, u32 err, u32 ref) __attribute__((always_inline)) {
      busy = false;
      if (kind != K_OP) { if (type != MSIM_T_OK) my_flags |= MSIM_FLAG_ROUND_LIMIT; return; }
      cmp_row = true; cmp_packed = type | (MSIM_F_TXN << 2) | (err << 7) | (process << 12);
      cmp_value = ref & 0xFFFFFFu; cmp_len = ref >> 24;
      if (type == MSIM_T_INFO) process += N;  // crashed process; the Reusable client itself lives on
    };
    // the client's recv! consumes one envelope (client.clj:94-107)
    auto client_deliver = [&](u32 qtype, u32 qa, u32 qb) __attribute__((always_inline)) {
      s_recv_cl++;
      if (busy && qb == want) {
        if (qtype == M_TXN_OK) complete(MSIM_T_OK, 0, qa);
        else if (qtype == M_ERROR)
          if (qa == 0u) complete(MSIM_T_INFO, MSIM_ERR_TIMEOUT, c_value);   // code 0 :timeout is not :definite? (errors.edn:2-4)
          else complete(MSIM_T_FAIL, qa == 11 ? MSIM_ERR_TEMPORARILY_UNAVAILABLE : qa == 20 ? MSIM_ERR_KEY_DOES_NOT_EXIST : qa == 30 ? MSIM_ERR_TXN_CONFLICT : qa == 14 ? MSIM_ERR_ABORT : MSIM_ERR_PRECONDITION_FAILED, c_value);
        else complete(MSIM_T_OK, 0, c_value);  // init_ok
      }
    };

    if (alive && timeout_round) {
      if (busy && timeout_at <= T) complete(MSIM_T_INFO, MSIM_ERR_NET_TIMEOUT, c_value);
    }
    bool normal = alive && !timeout_round;   // this cluster runs R1-R4 in this wave-round
    M8_MARK(0)
    if (__ballot(normal)) {
      // ---- R1: scheduler ----
      const bool act = normal && due <= T;
      if (__ballot(act && phase == PH_INIT)) {
        if (act && phase == PH_INIT) { if (is_node) { mark = true; kind = K_INIT; } phase = PH_INIT_WAIT; }
      }
      if (NEM) {
        const bool nem_act = act && phase == PH_MAIN && nem_live && nem_next <= T;
        if (__ballot(nem_act)) {   // flip-flop start/stop (nemesis.clj:10-16 + [upstream] partition package)
          const u32 j = cs[CS_NEM_J];
          const u32 spec = scale32(draw32(key, S_NEM_SPEC, j), 4);
          const bool start = nem_act && (j & 1) == 0;
          if (nem_act) nem_rows = 2;
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
              if (n_payload + words > max_pay) CFLAG(MSIM_FLAG_PAYLOAD_OVERFLOW);
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
          if (nem_act && l == 0) { cs[CS_NEM_J] = j + 1u; cs[CS_NEM_NEXT] = T + __umulhi(draw32(key, S_NEM_STAGGER, j), p.nem_period2_us); }
        }
      }
      {
        const bool gen_on = act && phase == PH_MAIN && gen_live && gen_next <= T && free_mask != 0;
        if (__ballot(gen_on)) {
          const u32 nfree = __popc(free_mask);
          const u32 kk = cs[CS_GEN_K];
          const u64 h = draw64(key, S_GEN, kk);
          const u32 r_hi = (u32)(h >> 32), r_lo = (u32)h;
          const u32 pick = scale32(r_lo, nfree);
          const bool sel = gen_on && is_node && !busy && (u32)__popc(free_mask & lt) == pick;
          // the transaction ([upstream] elle list-append gen): lane 0 of the cluster writes the micro-ops and owns the key pool
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
          if (gen_on && bad) { CFLAG(bad); phase = PH_DONE; alive = false; normal = false; }
          else if (gen_on) {
            if (sel) { mark = true; kind = K_OP; m_value = n_payload | (n_mops << 24); }
            n_payload += n_mops;
            if (l == 0) { cs[CS_GEN_K] = kk + 1u; cs[CS_GEN_NEXT] = T + __umulhi(r_hi, p.gen_period2_us); }
          }
        }
      }

      M8_MARK(1)
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

      M8_MARK(2)
      M8_MARK2(0)
      // ---- R3: one input per node, then one for each service (endpoint order: lin-kv, lww-kv) ----
      bool rep = false, svc_rep = false;   // node -> own client, service -> node
      u32 r_type = 0, r_a = 0, r_b = 0;    // the answer to the client
      // The heavy steps run ONCE per round, after the handlers have said who needs them: lanes that reach them from different states (a root
      // read answered, a load answered, a cas answered, a timeout) would otherwise execute separate inlined copies one after the other.
      bool do_apply = false, do_reply_ok = false, do_unlock = false;
      u32 n_out = 0, o_dest = 0;           // node -> service: n_out messages, all to the same service; one in registers (o1_*) or the writes of my_wl[]
      u32 o1_type = 0, o1_a = 0, o1_b = 0, o_wlo = 0;
#define o_type o1_type   /* service -> node: a service lane's answer lies where a node lane keeps its request */
#define o_a o1_a
#define o_b o1_b
#define o_to o_dest
      u32 need_words = 0, done_ref = 0, done_rv = 0;   // the completed transaction's payload
      auto rec_of = [&](u32 ptr) -> u32 * { return g_rec + ((size_t)(ptr >> 20) * TC + (ptr & 0xFFFFFu)) * DT_RW; };
      auto send1 = [&](u32 dest, u32 type, u32 a, u32 b) { o_dest = dest; n_out = 1; o1_type = type; o1_a = a; o1_b = b; };
      auto start_txn = [&](u32 cmsg, u32 ref) {   // the lock is ours: current_tree (:358-365)
        cu[DC_STAGE] = DS_ROOT; cu[DC_CMSG] = cmsg; cu[DC_REF] = ref; cu[DC_J] = 0; cu[DC_NOWN] = 0;
        const u32 rid = ++node_msgid; cu[DC_RPC] = rid;
        send1(D_LIN, M_READ, 0, rid);
        wait_until = T + DT_AWAIT_US;
      };
      auto unlock = [&]() {   // the next waiting transaction takes the lock (:348, :371), in arrival order
        cu[DC_STAGE] = DS_IDLE;
        wait_until = INF;
        const u32 wq = cu[DC_WQN], cnt = wq & 0xFFu, head = wq >> 8;
        if (cnt) {
          const u32 cmsg = aux[D8_AUX_WQ + 2u * head], ref = aux[D8_AUX_WQ + 2u * head + 1u];
          cu[DC_WQN] = (cnt - 1u) | (((head + 1u) & (DT_WAITQ - 1u)) << 8);
          start_txn(cmsg, ref);
        }
      };
      auto load = [&](u32 ptr) {   // Tree.load with a cache miss (:83-101)
        const u32 rid = ++node_msgid;
        cu[DC_STAGE] = DS_LOAD; cu[DC_TARGET] = ptr; cu[DC_RPC] = rid;
        send1(D_LWW, M_READ, ptr, rid);
        wait_until = T + DT_AWAIT_US;
      };
      // The completed transaction: its reads see the version read + its own appends.  Sized here, written behind the cross-lane allocation below.
      // Waits, not bytes, are what this costs (a wavefront runs ~28 memory waits a round, one after the other): the micro-ops come in ONE batch, the
      // keys' element counts in one, and a key's row is four 16-byte words (max-writes-per-key 16, the default; other widths take the loops).
      u32 vis_pack_lo = 0, vis_pack_hi = 0;   // visible elements of the keys the micro-ops read, 8 bits each (kept for the writer below)
      auto reply_txn_ok = [&]() {
        rep = true; r_type = M_TXN_OK; r_b = cu[DC_CMSG];
        done_ref = cu[DC_REF]; done_rv = cu[DC_RV];
        const u32 off0 = done_ref & 0xFFFFFFu, n = done_ref >> 24;
        u32 wv[8], cn[8];
#pragma unroll
        for (u32 j = 0; j < 8u; j++) wv[j] = g_pay[off0 + min(j, n - 1u)];
#pragma unroll
        for (u32 j = 0; j < 8u; j++) { cn[j] = 0; if (j < n && !(wv[j] & 1u) && done_rv != V_NIL) cn[j] = g_kvn[(wv[j] >> 1) & 0x7FFFu]; }
#pragma unroll
        for (u32 j = 0; j < 8u; j++) {
          if (j >= n) continue;
          const u32 w = wv[j], k = (w >> 1) & 0x7FFFu;
          need_words++;
          if (!(w & 1u)) {
            u32 len = 0;
            if (cn[j]) {
              const u32 *const kvr = g_kv + k * mw;
              if (mw == 16u) {
                const uint4 *const q = reinterpret_cast<const uint4 *>(kvr);
                const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const u32 row[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
                for (u32 i = 0; i < 16u; i++) len += (i < cn[j] && (row[i] >> 8) <= done_rv) ? 1u : 0u;
              } else { while (len < cn[j] && (kvr[len] >> 8) <= done_rv) len++; }
            }
            if (j < 4u) vis_pack_lo |= len << (8u * j); else vis_pack_hi |= len << (8u * (j - 4u));
#pragma unroll
            for (u32 e = 0; e < 8u; e++) len += (e < j && (wv[e] & 1u) && ((wv[e] >> 1) & 0x7FFFu) == k) ? 1u : 0u;
            need_words += (len + 3u) / 4u;
          }
        }
      };

      const bool await_over = normal && is_node && wait_until <= T;   // a node's due timer comes before its due message
      const bool take = normal && !await_over && l <= N + 1u && deliver_at <= T;
      if (await_over) {   // Promise#await gave up (promise.rb:24-29): RPCError.timeout => error 0 to the client (node.rb:172), the lock is free
        rep = true; r_type = M_ERROR; r_a = 0; r_b = cu[DC_CMSG];
        do_unlock = true;
      }
      if (take) {
        const uint4 q = cm; deliver_at = INF;
        const u32 qsrc = q.w >> 24, qb = q.w & 0xFFFFFFu, qtype = q.y & 0xFFu, qa = q.z;
        if (qsrc >= N && qsrc < LIN) s_recv_cl++; else s_recv_sv++;
        // What a handler reads from HBM first is known from the envelope alone, and the three kinds of lanes would otherwise wait for it one after
        // the other (divergent branches run in sequence): a node whose root read is answered wants the root's record, lin-kv its sender's table of
        // cas requests, lww-kv the word of the record that holds the replica.  One batch for all of them: three 16-byte words + the flags word past the L1.
        const u32 st = is_node ? cu[DC_STAGE] : 0u;
        const bool pf_root = is_node && st == DS_ROOT && qtype == M_READ_OK && qb == cu[DC_RPC];
        const bool pf_cas = is_lin && qtype == M_CAS, pf_lww = l == N + 1u;
        uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = pf0, pf2 = pf0; u32 pf3 = 0;
        if (pf_root || pf_cas || pf_lww) {
          const u32 *const pp = pf_cas ? g_cas + (size_t)qsrc * DT_CASQ * 3u : rec_of(qa);
          if (!pf_cas) pf3 = __hip_atomic_load(pp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (word 3 changes after a record's creation, by L2 atomics)
          if (!pf_lww) { pf0 = reinterpret_cast<const uint4 *>(pp)[0]; pf1 = reinterpret_cast<const uint4 *>(pp)[1]; pf2 = reinterpret_cast<const uint4 *>(pp)[2]; }
        }
        if (is_node) {
          switch (qtype) {
            case M_INIT:
              if (l != 0u) { rep = true; r_type = M_INIT_OK; r_b = qb; break; }
              {   // the first node writes the initial state (:337-345): Tree.empty, then the root pointer
                u32 *const r = g_rec;
                *reinterpret_cast<uint4 *>(r) = make_uint4(128u << 16, 0u, 0u, 3u);
                const u32 rid = ++node_msgid;
                cu[DC_STAGE] = DS_INIT_LEAF; cu[DC_CMSG] = qb; cu[DC_RPC] = rid;
                send1(D_LWW, M_WRITE, 0, rid);
              } break;
            case M_TXN:
              if (st == DS_IDLE) start_txn(qb, qa);
              else { const u32 wq = cu[DC_WQN], cnt = wq & 0xFFu;
                if (cnt == DT_WAITQ) my_flags |= MSIM_FLAG_ARENA_OVERRUN;
                else { const u32 slot = ((wq >> 8) + cnt) & (DT_WAITQ - 1u); aux[D8_AUX_WQ + 2u * slot] = qb; aux[D8_AUX_WQ + 2u * slot + 1u] = qa; cu[DC_WQN] = wq + 1u; } }
              break;
            case M_READ_OK: case M_WRITE_OK: case M_CAS_OK: case M_ERROR:
              switch (st) {
                case DS_INIT_LEAF:
                  if (qb != cu[DC_RPC]) break;
                  { const u32 rid = ++node_msgid; cu[DC_STAGE] = DS_INIT_ROOT; cu[DC_RPC] = rid; send1(D_LIN, M_WRITE, 0, rid); }
                  break;
                case DS_INIT_ROOT:
                  if (qb != cu[DC_RPC]) break;
                  cu[DC_STAGE] = DS_IDLE; rep = true; r_type = M_INIT_OK; r_b = cu[DC_CMSG];
                  break;
                case DS_ROOT:
                  if (qb != cu[DC_RPC]) break;
                  if (qtype != M_READ_OK) { rep = true; r_type = M_ERROR; r_a = 14; r_b = cu[DC_CMSG]; do_unlock = true; break; }   // "Unsure how to handle" (:364)
                  cu[DC_P1] = qa; cu[DC_T] = qa; cu[DC_PSTART] = next_p + 1u;
                  {   // the root's record (prefetched above) stays in LDS for the walks of this transaction (its content never changes)
                    const u32 rw3 = pf3;
                    const uint4 rq1 = pf1, rq2 = pf2;
                    const u32 rw0 = pf0.x, rv2 = pf0.z;
                    const u32 rch[8] = {rq1.x, rq1.y, rq1.z, rq1.w, rq2.x, rq2.y, rq2.z, rq2.w};
                    cu[DC_RV] = rv2; cu[DC_RC] = rw0;
#pragma unroll
                    for (u32 c = 0; c < 8u; c++) cu[DC_RC + 1u + c] = rch[c];
                    if ((rw3 >> (2u + l)) & 1u) do_apply = true; else load(qa); }
                  break;
                case DS_LOAD:
                  if (qb != cu[DC_RPC]) break;
                  if (qtype == M_READ_OK) { atomicOr(rec_of(cu[DC_TARGET]) + 3, 1u << (2u + l)); do_apply = true; }   // @@cache[ptr] = tree (:95)
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
                  if (qtype == M_CAS_OK) do_reply_ok = true;
                  else { rep = true; r_type = M_ERROR; r_a = 30; r_b = cu[DC_CMSG]; }   // txn_conflict (:385)
                  do_unlock = true;
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
          else {   // cas, no create_if_not_exists.  The request is self-contained (:376-388): its `from` and its transaction come from the sender's
            // table of cas requests under the msg_id, not from what the sender holds NOW (it may have given up on this cas and moved on)
            u32 c_from = 0, c_ref = 0; bool c_hit = false;
            { const u32 ce[12] = {pf0.x, pf0.y, pf0.z, pf0.w, pf1.x, pf1.y, pf1.z, pf1.w, pf2.x, pf2.y, pf2.z, pf2.w};   // (the sender's table, prefetched above)
#pragma unroll
              for (u32 i = 0; i < DT_CASQ; i++) { const u32 e0 = ce[3u * i], e1 = ce[3u * i + 1u], e2 = ce[3u * i + 2u]; if (e0 == qb) { c_hit = true; c_from = e1; c_ref = e2; } } }
            if (!c_hit) { my_flags |= MSIM_FLAG_ARENA_OVERRUN; o_type = M_ERROR; o_a = 22; }   // engine capacity: DT_CASQ outstanding cas requests per node
            else if (!root_exists) { o_type = M_ERROR; o_a = 20; }
            else if (root != c_from) { o_type = M_ERROR; o_a = 22; }
            else {
              const u32 ref = c_ref, off0 = ref & 0xFFFFFFu, n = ref >> 24, v = ++cur_v;
              root = qa;
              for (u32 i0 = 0; i0 < n; i0 += 4u) {   // the transaction's appends enter the log: four micro-ops per pair of round trips
                u32 w4[4], c4[4], hf4[4];
#pragma unroll
                for (u32 t = 0; t < 4u; t++) w4[t] = g_pay[off0 + min(i0 + t, n - 1u)];
#pragma unroll
                for (u32 t = 0; t < 4u; t++) { const u32 k = (w4[t] >> 1) & 0x7FFFu; c4[t] = g_kvn[k]; hf4[t] = g_first[k]; }
#pragma unroll
                for (u32 t = 0; t < 4u; t++) {
                  const u32 w = w4[t], k = (w >> 1) & 0x7FFFu;
                  if (i0 + t >= n || !(w & 1u)) continue;
                  u32 c = c4[t];
#pragma unroll
                  for (u32 u = 0; u < t; u++) c += ((w4[u] & 1u) && ((w4[u] >> 1) & 0x7FFFu) == k) ? 1u : 0u;   // (an earlier append of this batch to the same key)
                  if ((hf4[t] >> 8) == D8_NOFIRST) g_first[k] = (v << 8) | (hf4[t] & 0xFFu);
                  g_kv[k * mw + c] = ((w >> 16) & 0xFFu) | (v << 8); g_kvn[k] = c + 1u;
                }
              }
              o_type = M_CAS_OK; o_a = 0;
            }
          }
        } else {   // lww-kv (service.clj:214-243 as written): merge-source, merge-dest, then the replica that serves the request
          svc_rep = true; o_to = qsrc; o_b = qb;
          svc_ctr += 2u;
          const u32 r = scale32(draw32(key, 12u /* S_SVC */, svc_ctr++), 2);
          u32 *const rp = rec_of(qa) + 3;   // (the replica bits; the nodes set their "loaded" bits in the same word: atomics)
          if (qtype == M_WRITE) { atomicAnd(rp, ~3u); atomicOr(rp, r); o_type = M_WRITE_OK; o_a = qa; }
          else if ((pf3 & 3u) == r) { o_type = M_READ_OK; o_a = qa; }   // (the word was prefetched above)
          else { o_type = M_ERROR; o_a = 20; }
        }
      }

      M8_MARK2(1)
      if (__ballot(do_apply)) {   // apply_txn from where it stopped: may ask for a load, start the save, or finish a read-only transaction
        D8ApplyIO io;
        io.np = next_p; io.mid = node_msgid; io.my_flags = my_flags; io.wait_until = wait_until;
        io.n_out = n_out; io.o_dest = o_dest; io.o1_type = o1_type; io.o1_a = o1_a; io.o1_b = o1_b; io.o_wlo = o_wlo; io.done = 0;
        D8ApplyK kc;
        kc.scratch = p.scratch; kc.payload = p.payload; kc.scratch_words = p.scratch_words; kc.max_pay = max_pay; kc.N = N; kc.TC = TC; kc.mv = mv; kc.mw = mw;
        kc.off_aux = OFF_AUX; kc.off_cur = tp.off_cur; kc.off_gen = tp.off_gen;
        io = d8_apply(io, kc, inst, lane, T, do_apply);
        next_p = io.np; node_msgid = io.mid; my_flags = io.my_flags; wait_until = io.wait_until;
        n_out = io.n_out; o_dest = io.o_dest; o1_type = io.o1_type; o1_a = io.o1_a; o1_b = io.o1_b; o_wlo = io.o_wlo;
        do_reply_ok = do_reply_ok || (io.done & 1u) != 0; do_unlock = do_unlock || (io.done & 2u) != 0;
      }
      M8_MARK2(2)
      if (do_reply_ok) reply_txn_ok();
      if (do_unlock) unlock();          // (after the answer: the next lock holder's root read follows it)
      M8_MARK(3)
      M8_MARK2(3)
      // completed transactions: payload words allocated in node order, each node writes its own
      if (__ballot(need_words != 0)) {
        u32 excl = 0, total = 0;
        for (u32 s = 0; s < N; s++) { const u32 v = GGET(need_words, s); excl += s < l ? v : 0u; total += v; }
        if (total) {
          if (n_payload + total > max_pay) { CFLAG(MSIM_FLAG_PAYLOAD_OVERFLOW); if (need_words) r_a = 0; }
          else {
            if (need_words) {
              const u32 off0 = done_ref & 0xFFFFFFu, n = done_ref >> 24;
              u32 pp = n_payload + excl;
              r_a = pp | (need_words << 24);
              u32 wv[8];
#pragma unroll
              for (u32 j = 0; j < 8u; j++) wv[j] = g_pay[off0 + min(j, n - 1u)];
#pragma unroll
              for (u32 j = 0; j < 8u; j++) {
                if (j >= n) continue;
                const u32 w = wv[j], k = (w >> 1) & 0x7FFFu;
                if (w & 1u) { g_pay[pp++] = w; continue; }
                const u32 vis = ((j < 4u ? vis_pack_lo >> (8u * j) : vis_pack_hi >> (8u * (j - 4u))) & 0xFFu);
                u32 e = 0, acc = 0;
                const u32 hdr = pp++;
                if (vis) {
                  const u32 *const kvr = g_kv + k * mw;
                  if (mw == 16u) {   // the visible prefix of the row from four 16-byte words
                    const uint4 *const q = reinterpret_cast<const uint4 *>(kvr);
                    const uint4 q0 = q[0], q1 = vis > 4u ? q[1] : make_uint4(0, 0, 0, 0), q2 = vis > 8u ? q[2] : make_uint4(0, 0, 0, 0), q3 = vis > 12u ? q[3] : make_uint4(0, 0, 0, 0);
                    const u32 row[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
                    for (u32 i = 0; i < 16u; i++) if (i < vis) { acc |= (row[i] & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } }
                  } else for (u32 i = 0; i < vis; i++) { acc |= (kvr[i] & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } }
                }
#pragma unroll
                for (u32 i = 0; i < 8u; i++) { const u32 wi = wv[i];
                  if (i < j && (wi & 1u) && ((wi >> 1) & 0x7FFFu) == k) { acc |= ((wi >> 16) & 0xFFu) << (8 * (e & 3)); if ((++e & 3) == 0) { g_pay[pp++] = acc; acc = 0; } } }
                if (e & 3) g_pay[pp++] = acc;
                g_pay[hdr] = (k << 1) | ((e ? e : 0xFFu) << 16);  // a key without elements reads nil
              }
            }
            n_payload += total;
          }
        }
      }

      M8_MARK(4)
      M8_MARK2(4)
      // COMMIT: ids in lane order (nodes, lin-kv, lww-kv); a node's messages in the order it emitted them: the answer to its client, then
      // what the next step sends to a service
      bool c_arr = false; u32 ca_y = 0, ca_a = 0, ca_b = 0;
      {
        const u32 rcnt = rep ? 1u : 0u;
        const u32 cnt = is_node ? rcnt + n_out : (svc_rep ? 1u : 0u);
        if (__ballot(cnt != 0)) {
          u32 my_off = 0, total = 0;
          for (u32 s = 0; s < N + 2u; s++) { const u32 v = GGET(cnt, s); my_off += s < l ? v : 0u; total += v; }
          if (is_node) { s_send_cl += rcnt; s_send_sv += n_out; } else s_send_sv += cnt;
          // node -> service: the service lane takes each node's run in node order
          wave_lds_fence();   // (the write lists of this round were stored by other lanes of this wavefront: keep the compiler from moving the loads up)
          u32 ts = GB(is_node && n_out != 0);
          while (__ballot(ts != 0)) {
            const bool on = ts != 0;
            const u32 s = on ? (u32)__builtin_ctz(ts) : 0u; ts &= ts - 1u;
            const u32 dst = GGET(o_dest, s), kn = GGET(n_out, s), off = GGET(my_off, s) + GGET(rcnt, s);
            const u32 t1 = GGET(o1_type, s), a1 = GGET(o1_a, s), b1 = GGET(o1_b, s), wlo = GGET(o_wlo, s);
            if (on && l == N + dst) {
              if (wlo == 0u) arrive(next_id + off, t1, a1, b1, s);
              else { const u32 *const wl = g_wl + (size_t)s * DT_MAXW;   // the node's write list: eight pointers per round trip (a load per envelope made every write of a save a wait of its own)
                for (u32 k0 = 0; k0 < kn; k0 += 8u) {
                  u32 w8[8];
#pragma unroll
                  for (u32 t = 0; t < 8u; t++) w8[t] = wl[min(k0 + t, kn - 1u)];
#pragma unroll
                  for (u32 t = 0; t < 8u; t++) if (k0 + t < kn) arrive(next_id + off + k0 + t, M_WRITE, w8[t], wlo + k0 + t, s);
                } }
            }
          }
          // service -> node (lin-kv, then lww-kv)
          {
            const u32 sv = GB(svc_rep);
#pragma unroll
            for (u32 q2 = 0; q2 < 2u; q2++) {
              const u32 s = N + q2;
              const u32 ty = GGET(o_type, s), a = GGET(o_a, s), b = GGET(o_b, s), d = GGET(o_to, s), off = GGET(my_off, s);
              if (((sv >> s) & 1u) && l == d) arrive(next_id + off, ty, a, b, N + s);
            }
          }
          // node -> its own client: no latency; lost like any other message (net.clj:214)
          if (rep) {
            const u32 id = next_id + my_off;
            if (!(NET_RANDOM && p_loss && cs[CS_LOSS_ON] && draw32(key, S_LOSS, id) < p_loss)) { c_arr = true; ca_y = (id << 8) | r_type; ca_a = r_a; ca_b = r_b; }
          }
          next_id += total;
        }
        M8_MARK2(5)
        poll();
      }

      M8_MARK(5)
      M8_MARK2(6)
      // ---- R4: the clients' recv! loops (client.clj:94-107) ----
      if (__ballot(c_arr || (busy && (cin_n | csp_n) != 0))) {
        for (;;) {
          const bool stale = normal && busy && (cin_n | csp_n) != 0;
          const bool fresh = normal && !stale && busy && c_arr;
          if (!__ballot(stale || fresh)) break;
          if (stale) {
            u32 best = 0; bool in_spill = false;
            uint2 bk = make_uint2(INF, INF);
            for (u32 i = 0; i < cin_n; i++) {
              const uint2 kk = *reinterpret_cast<const uint2 *>(&my_cq[i * 64u]);
              if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; }
            }
            for (u32 i = 0; i < csp_n; i++) {
              const uint2 kk = *reinterpret_cast<const uint2 *>(&my_cspill[i]);
              if (kk.x < bk.x || (kk.x == bk.x && kk.y < bk.y)) { bk = kk; best = i; in_spill = true; }
            }
            uint4 e;
            if (in_spill) { e = my_cspill[best]; csp_n--; if (best != csp_n) my_cspill[best] = my_cspill[csp_n]; }
            else { e = my_cq[best * 64u]; cin_n--; if (best != cin_n) my_cq[best * 64u] = my_cq[cin_n * 64u]; }
            client_deliver(e.y & 0xFFu, e.z, e.w & 0xFFFFFFu);
          } else if (fresh) {
            c_arr = false;
            client_deliver(ca_y & 0xFFu, ca_a, ca_b);
          }
        }
        if (c_arr && normal) {  // nobody is in recv!: the envelope waits for the next RPC (and is skipped there as stale)
          const uint4 e = make_uint4(T, ca_y, ca_a, ca_b | (l << 24));
          if (cin_n < CQ) { my_cq[cin_n * 64u] = e; cin_n++; }
          else if (csp_n < tp.client_spill) my_cspill[csp_n++] = e;
          else my_flags |= MSIM_FLAG_INBOX_OVERFLOW;
        }
      }
    }

    M8_MARK(6)
    // ---- history rows: nemesis rows, invocations (lane order), completions (lane order) ----
    {
      const u32 imask = GB(inv_row), cmask = GB(cmp_row);
      const u32 ni = __popc(imask);
      const u32 nr = nem_rows + ni + __popc(cmask);
      if (__ballot(alive && nr != 0)) {
        const bool ovf = alive && nr != 0 && n_rows + nr > max_rows;
        if (ovf) { CFLAG(MSIM_FLAG_ROWS_OVERFLOW); alive = false; }
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
        const u32 new_n = wr ? n_rows + nr : n_rows;
        n_rows = new_n;
      }
    }
    M8_MARK(7)
    M8_MARK2(7)
  }

  // ---- epilogue ----
  u32 t_send_cl = 0, t_send_sv = 0, t_recv_cl = 0, t_recv_sv = 0;
  for (u32 s = 0; s < GS; s++) { t_send_cl += GGET(s_send_cl, s); t_send_sv += GGET(s_send_sv, s); t_recv_cl += GGET(s_recv_cl, s); t_recv_sv += GGET(s_recv_sv, s); }
  wave_lds_fence();
  u32 flags = cs[CS_FLAGS];
  const u32 rounds = cs[CS_ROUNDS];
  for (u32 b = 1; b <= MSIM_FLAG_ARENA_OVERRUN; b <<= 1) if (GB((my_flags & b) != 0)) flags |= b;
  if (real && l == 0) {
    msim_net_stats st;
    st.all_send = (u64)t_send_cl + t_send_sv; st.all_recv = (u64)t_recv_cl + t_recv_sv;
    st.clients_send = t_send_cl; st.clients_recv = t_recv_cl;
    st.servers_send = t_send_sv; st.servers_recv = t_recv_sv;
    p.stats[inst] = st;
    msim_inst_meta m; m.n_rows = n_rows; m.n_payload_words = n_payload; m.flags = flags; m.n_rounds = rounds;
    m.n_events = 0; m.reserved[0] = 0; m.reserved[1] = 0; m.reserved[2] = 0;
#ifdef D8_PROF
    if (grp == 0) { m.n_events = (u32)(pacc[0] >> 6); m.reserved[0] = (u32)(pacc[1] >> 6); m.reserved[1] = (u32)(pacc[2] >> 6); m.reserved[2] = (u32)(pacc[3] >> 6); }
    if (grp == 1) { m.n_events = (u32)(pacc[4] >> 6); m.reserved[0] = (u32)(pacc[5] >> 6); m.reserved[1] = (u32)(pacc[6] >> 6); m.reserved[2] = (u32)(pacc[7] >> 6); }
    if (grp == 2) { m.n_events = wave_rounds; }
#endif
    p.meta[inst] = m;
  }
}

#undef o_type
#undef o_a
#undef o_b
#undef o_to
#undef next_p
#undef root
#undef casn
#undef cur_v
#undef node_msgid
#undef svc_ctr
#undef CFLAG
#undef D8_INST
#undef g_rows
#undef g_pay
#undef g_scr
#undef g_kv
#undef g_kvn
#undef g_first
#undef g_rec
#undef g_wl
#undef g_cas
#undef my_spill
#undef my_cspill
#undef aux
#undef my_wl

}  // namespace

// Whether eight clusters per wavefront simulate this configuration (see the header of this file).
bool msim_dt8_eligible(const msim_config &c) {
  return c.node_program == MSIM_NODE_TXN_DATOMIC && c.journal_capacity == 0 && c.n_nodes >= 1 && c.n_nodes <= GS - 2u && c.concurrency == c.n_nodes;
}

// Extra per-instance scratch words behind the queues' spill area: what of the LDS queues of dt_kernel<> does not fit this kernel's RQ slots,
// the clients' spill, the nodes' auxiliary words (waiting ring, path, position keys).
uint64_t msim_dt8_extra_scratch_words(const msim_config &c) {
  return ((uint64_t)(c.n_nodes + 2) * c.inbox_capacity + (uint64_t)c.n_nodes * M8_CLIENT_CAP) * 4 + (((uint64_t)c.n_nodes * D8_AUX + 3) & ~3ull);   // (an instance's scratch stays a multiple of 16 bytes: the queues are read as uint4)
}

hipError_t msim_launch_dt8(const KParams &kp, uint32_t n, hipStream_t st) {
  const msim_config &c = kp.cfg;
  if (n < MSIM_DT8_MIN_CLUSTERS && !(kp.dev_flags & 0x400u)) return MSIM_LAYOUT_DOES_NOT_FIT;   // the kernel is latency-bound: below the measured crossover one cluster per wavefront is faster
  M8Params tp;
  tp.k = kp; tp.n_inst = n;
  const uint32_t cap_tot = c.inbox_capacity + c.spill_capacity;
  tp.node_spill = cap_tot > RQ ? cap_tot - RQ : 0;
  tp.client_spill = M8_CLIENT_CAP - CQ;
  tp.client_spill_off = kp.spill_off + (uint64_t)(kp.N + 2) * tp.node_spill * 4;
  tp.stack_off = tp.client_spill_off + (uint64_t)kp.N * tp.client_spill * 4;
  size_t off = (size_t)RQ * 64 * 16;
  tp.off_cq = (u32)off; off += (size_t)CQ * 64 * 16;
  tp.off_cur = (u32)off; off += (size_t)8 * kp.N * D8_CW * 4;
  tp.off_gen = (u32)off; off += (size_t)8 * 36 * 4;
  off = (off + 15) & ~(size_t)15;
  tp.off_misc = (u32)off; off += 64 * 4;
  tp.off_cs = (u32)off; off += 8 * 8 * 4;   // the clusters' shared words (CS_WORDS each)
  tp.round_limit = (kp.dev_flags & 0x100u) ? 4000000u : ROUND_LIMIT;
  const size_t lds = off;
  const bool rnd = c.latency_dist != MSIM_LAT_CONSTANT || c.p_loss_q32 != 0;
  if (rnd) MSIM_UPLOAD_ONCE(m8_log2_q24, msim_log2_q24, sizeof(msim_log2_q24));   // (1 KiB, once per device)
  const dim3 grid((n + 7) / 8), block(64);
  if (kp.dev_flags & 0x1000u) std::fprintf(stderr, "[dt8] %u clusters, eight per wavefront, %zu B of LDS per wavefront\n", n, lds);   // developer trace bit
  if (c.nemesis_mask) { if (rnd) hipLaunchKernelGGL((dt8_kernel<true, true>), grid, block, lds, st, tp); else hipLaunchKernelGGL((dt8_kernel<true, false>), grid, block, lds, st, tp); }
  else { if (rnd) hipLaunchKernelGGL((dt8_kernel<false, true>), grid, block, lds, st, tp); else hipLaunchKernelGGL((dt8_kernel<false, false>), grid, block, lds, st, tp); }
  return hipGetLastError();
}
