// lin_check_dev.hip — per-key linearizability of lin-kv histories on the device (msim_check for MSIM_WL_LIN_KV).
//
// The reference's lin-kv checker is `independent/checker` over Knossos' `checker/linearizable` with a CAS-register model
// (workload/lin_kv.clj:84, [upstream] jepsen.tests.linearizable-register).  lin_check.cpp restates it on the host as just-in-time
// linearization with dominance pruning (and the symmetry of identical never-returning calls); this file is the same search on the device:
//
//   * a CONFIGURATION (register value, set of pending calls already linearized) lives in the registers of one lane — at most 64
//     at a time; a PENDING CALL (process, f, values, will-it-return) lives in the registers of the lane that carries its bit;
//   * when a call returns, the frontier is expanded a call at a time for all its configurations at once; "is this successor
//     dominated / does it dominate" is one compare per lane and a ballot; a new configuration goes to the first free lane;
//   * pass 0 streams the rows once (64 rows = 1 KiB per load), pairs every invocation with its completion (the call's outcome
//     decides how the search treats it from the start: a :fail never happened, an :ok read must see its value, everything else
//     stays pending forever) and notes the row range of each key; then the keys are walked one by one;
//   * a closure that needs more than 64 configurations (many calls open at once: a partition that leaves a dozen writes and cas
//     indeterminate) moves to a POOL — a hash table over (value, linearized returning calls) whose chains are what dominance
//     compares, filled by every thread with compare-and-swap pushes — and the survivors come back to the registers as soon as 64
//     lanes hold them again.  Pass 1 (one wavefront per history, 14 per CU) has pools of 2048 configurations in an HBM workspace;
//     what outgrows those runs again in pass 2 (a workgroup of 1024 per history, the pool in LDS: 14 784 configurations), and a
//     closure that outgrows that too moves on to an HBM slot of 262 144.
//
// The search is exact, so its verdict is the host's.  What exceeds all of that (or 64 pending calls in the pairing table, 23 on one
// key while a pool is in use, more rows than the LDS table covers) goes to the host search of lin_check.cpp — never silently
// approximated.  BASELINE configs[3] with partitions: 8192 histories, 3147 use a pool, 73 reach pass 2, 2 the large HBM slots, none
// the host: 66 ms where the registers-then-host version of round 3 took 127 (profiles/r04v_lin_check.txt).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "engine_internal.h"

void msim_lin_check_instance_host(const msim_op *rows, uint32_t n_rows, uint32_t flags, msim_check_result *res);   // lin_check.cpp

namespace {

constexpr u32 NEEDS_HOST = 3u;          // msim_check_result.valid while a history awaits the host search
constexpr u32 EXPLORE_LIMIT = 50000u;   // configurations expanded for one returning call before the host takes over

struct LParams {
  const msim_op *rows;
  const msim_inst_meta *meta;   // per history: n_rows, flags (null: use off[])
  const uint64_t *off;          // per history: first row (null: history i at i * stride)
  msim_check_result *out;
  u32 stride;                   // rows per history slab
  u32 table_rows;               // rows the LDS outcome table covers
  const u32 *list;              // history indices to check (null: all)
};

__device__ __forceinline__ u32 rl(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u32 l_bperm(u32 v, u32 l) { return (u32)__builtin_amdgcn_ds_bpermute((int)(l << 2), (int)v); }
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ u32 l_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, BOUND); }
// maximum over the lanes BELOW this one (0 for lane 0), by DPP
__device__ __forceinline__ u32 l_excl_max(u32 v) {
  v = l_dpp<0x138, 0xF, true>(v);            // wave_shr:1
  v = max(v, l_dpp<0x111, 0xF, true>(v));    // row_shr:1 .. 8, then the rows before
  v = max(v, l_dpp<0x112, 0xF, true>(v));
  v = max(v, l_dpp<0x114, 0xF, true>(v));
  v = max(v, l_dpp<0x118, 0xF, true>(v));
  v = max(v, l_dpp<0x142, 0xA, false>(v));
  v = max(v, l_dpp<0x143, 0xC, false>(v));
  return v;
}
__device__ __forceinline__ u32 l_wave_sum(u32 v) {
  for (int o = 32; o; o >>= 1) v += (u32)__shfl_xor((int)v, o);
  return v;
}

// The outcome table of a history (LDS): per row 16 bits of value (bytes 1 and 2 of the completion's value word) and 2 bits of state
// (0 = no completion, else the completion's type), 2.125 bytes per row — pass 1 keeps 14 histories per CU resident with it.
struct Outcome { unsigned short *val; u32 *state; };
__host__ __device__ __forceinline__ size_t outcome_bytes(u32 rows) { return (((size_t)rows * 2 + 3) & ~(size_t)3) + (size_t)((rows + 15) / 16) * 4; }
__device__ __forceinline__ Outcome outcome_at(u32 *base, u32 rows) { Outcome o; o.state = base; o.val = reinterpret_cast<unsigned short *>(base + (rows + 15) / 16); return o; }
// bit 31 = has a completion, bits 0-1 its type, bits 8-23 its value bytes 1 and 2
__device__ __forceinline__ u32 outcome_get(const Outcome &o, u32 row) {
  const u32 st = (o.state[row >> 4] >> ((row & 15u) * 2u)) & 3u;
  return st ? (0x80000000u | st | ((u32)o.val[row] << 8)) : 0u;
}

// ---- pass 0: counts, key ranges, invocation -> completion (one wavefront) ------------------------------------------------------------
// Returns false when more than 64 calls are open at once (the pairing table is the wavefront).
__device__ bool pair_rows(const uint4 *r, u32 n, u32 *key_lo, u32 *key_hi, const Outcome outcome, u32 lane, u32 &c_inv, u32 &c_ok, u32 &c_fail, u32 &c_info) {
  bool o_used = false; u32 o_proc = 0, o_key = 0, o_row = 0;   // lane = one open call
  c_inv = 0; c_ok = 0; c_fail = 0; c_info = 0;
  bool fits = true;
  for (u32 i = lane; i < (n + 15) / 16; i += 64) outcome.state[i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  for (u32 base = 0; base < n && fits; base += 64) {
    const u32 idx = base + lane;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (idx < n) row = r[idx];
    const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12;
    const bool live = idx < n && proc != MSIM_PROCESS_NEMESIS;
    if (live) { c_inv += type == MSIM_T_INVOKE; c_ok += type == MSIM_T_OK; c_fail += type == MSIM_T_FAIL; c_info += type == MSIM_T_INFO; }
    const bool reg = live && (f == MSIM_F_READ || f == MSIM_F_WRITE || f == MSIM_F_CAS);
    if (reg) { atomicMin(&key_lo[row.w & 0xFFu], idx); atomicMax(&key_hi[row.w & 0xFFu], idx); }
    u64 m = __ballot(reg);
    // While no call is open, the rows of a chunk that come as whole pairs — an invocation directly followed (among the register rows) by
    // the completion of the same process on the same key: nearly all of them, the clients of a key seldom overlap — are paired by every
    // lane at once (round 6; before, every row walked the loop below: a readlane, two ballots, a table lane).  From the first row that
    // does not fit on, the loop below takes over.
    if (m && !__ballot(o_used)) {
      const u64 below = m & ((1ull << lane) - 1ull);
      const u32 rank = (u32)__popcll(below);
      const u32 pl = below ? 63u - (u32)__builtin_clzll(below) : 0u;
      const u32 pz = l_bperm(row.z, pl), pw = l_bperm(row.w, pl);
      const bool fit = (rank & 1u) ? (type != MSIM_T_INVOKE && (pz & 3u) == MSIM_T_INVOKE && (pz >> 12) == proc && (pw & 0xFFu) == (row.w & 0xFFu)) : type == MSIM_T_INVOKE;
      const u64 viol = __ballot(reg && !fit);
      const u32 nclean = (u32)__popcll(viol ? m & ((1ull << (u32)__builtin_ctzll(viol)) - 1ull) : m) & ~1u;   // whole pairs before the first row that does not fit
      if (reg && (rank & 1u) && rank < nclean) {
        const u32 irow = base + pl;
        outcome.val[irow] = (unsigned short)(row.w >> 8);
        atomicOr(&outcome.state[irow >> 4], type << ((irow & 15u) * 2u));
      }
      m &= ~__ballot(reg && rank < nclean);
    }
    while (m) {
      const u32 j = (u32)__builtin_ctzll(m); m &= m - 1;
      const u32 z = rl(row.z, j), w = rl(row.w, j);
      const u32 jt = z & 3u, jp = z >> 12, jk = w & 0xFFu;
      const u64 hit = __ballot(o_used && o_proc == jp && o_key == jk);
      if (jt == MSIM_T_INVOKE) {   // (a second invocation of an open process replaces the first, which then never completes)
        u32 s;
        if (hit) s = (u32)__builtin_ctzll(hit);
        else {
          const u64 used = __ballot(o_used);
          if (used == ~0ull) { fits = false; break; }
          s = (u32)__builtin_ctzll(~used);
        }
        if (lane == s) { o_used = true; o_proc = jp; o_key = jk; o_row = base + j; }
      } else if (hit) {
        const u32 s = (u32)__builtin_ctzll(hit);
        const u32 irow = rl(o_row, s);
        if (lane == s) { o_used = false; outcome.val[irow] = (unsigned short)(w >> 8); outcome.state[irow >> 4] |= jt << ((irow & 15u) * 2u); }   // (a completion's type is 1..3)
      }
    }
  }
  c_inv = l_wave_sum(c_inv); c_ok = l_wave_sum(c_ok); c_fail = l_wave_sum(c_fail); c_info = l_wave_sum(c_info);
  return fits;
}

enum { KEY_OK = 0, KEY_BAD = 1, KEY_UNKNOWN = 2, KEY_TOO_WIDE = 3 };   // what the search of one key returns

// ---- the configurations of one key in registers (one wavefront) ------------------------------------------------------------------
// lane i holds configuration i (c_lin, c_val) while bit i of `alive` is set; pending calls: lane s holds slot s (s_proc, s_op =
// f | v1 << 8 | v2 << 16 | skip << 31, s_ok, s_tw = for a never-returning call the never-returning calls in lower slots that are the
// same operation — the symmetry of lin_check.cpp: of identical ones only the lowest unused is tried).
struct RegSet { u64 c_lin; u32 c_val; u64 alive; };

// Every surviving configuration must have linearized the call with bit `bit` when it returns: close the set under linearizing
// pending calls, then keep what contains the bit.  The frontier is expanded a call at a time, all its configurations at once: one
// lane-parallel step decides which configurations may linearize call q next, and only those LEGAL successors are admitted one
// after the other (two ballots each: is it dominated, what does it dominate).  Returns KEY_OK (S = the survivors, the bit
// stripped), KEY_BAD (none) or KEY_TOO_WIDE (more than 64 at once: S = what was reached so far, ov_* = the one that did not fit).
__device__ int close_regs(RegSet &S, u64 pending, u64 info_bits, u32 s_op, u64 s_tw, u64 bit, u32 lane, u64 &ov_lin, u32 &ov_val) {
  u64 expanded = 0;
  u32 admitted = 0;
  for (;;) {
    const u64 front = S.alive & ~expanded;
    if (!front) break;
    expanded |= front;
    bool mine = ((front >> lane) & 1ull) && !(S.c_lin & bit);
    if (!__ballot(mine)) continue;
    u64 calls = pending;
    while (calls) {
      const u32 q = (u32)__builtin_ctzll(calls); calls &= calls - 1;
      const u32 op = rl(s_op, q);
      if (op >> 31) continue;                                             // an unfinished read constrains nothing
      u64 tw = 0;
      if ((info_bits >> q) & 1ull) tw = ((u64)rl((u32)(s_tw >> 32), q) << 32) | rl((u32)s_tw, q);
      const u32 of = op & 0xFFu, v1 = (op >> 8) & 0xFFu, v2 = (op >> 16) & 0xFFu;
      bool legal = mine && ((S.alive >> lane) & 1ull) && !((S.c_lin >> q) & 1ull) && !(tw & ~S.c_lin);
      u32 nv = S.c_val;
      if (of == MSIM_F_READ) legal = legal && S.c_val == v1;              // only :ok reads are stepped; they must see the current value
      else if (of == MSIM_F_WRITE) nv = v1;
      else { legal = legal && S.c_val == v1; nv = v2; }                   // cas [v v']
      u64 m = __ballot(legal);
      while (m) {
        const u32 j = (u32)__builtin_ctzll(m); m &= m - 1;
        const u64 lin2 = (((u64)rl((u32)(S.c_lin >> 32), j) << 32) | rl((u32)S.c_lin, j)) | (1ull << q);
        const u32 nv2 = rl(nv, j);
        // Dominance: for equal (value, linearized returning calls), a configuration that has linearized FEWER never-returning
        // calls can still do everything the other can.  Keep only the minimal ones.
        const u64 ib2 = lin2 & info_bits, key2 = lin2 & ~info_bits;
        const bool same = ((S.alive >> lane) & 1ull) && S.c_val == nv2 && (S.c_lin & ~info_bits) == key2;
        const u64 my_ib = S.c_lin & info_bits;
        if (__ballot(same && (my_ib & ib2) == my_ib)) continue;           // an existing subset dominates the new one (or is it)
        const u64 kill = __ballot(same && (my_ib & ib2) == ib2);          // the new one dominates these: their successors are dominated by its
        S.alive &= ~kill; m &= ~kill;
        if (++admitted > EXPLORE_LIMIT) { ov_lin = lin2; ov_val = nv2; return KEY_TOO_WIDE; }
        if (S.alive == ~0ull) { ov_lin = lin2; ov_val = nv2; return KEY_TOO_WIDE; }
        const u32 fl = (u32)__builtin_ctzll(~S.alive);
        if (lane == fl) { S.c_lin = lin2; S.c_val = nv2; mine = false; }  // (a new configuration: the next round expands it)
        S.alive |= 1ull << fl; expanded &= ~(1ull << fl); m &= ~(1ull << fl);
      }
    }
  }
  S.alive = __ballot(((S.alive >> lane) & 1ull) && (S.c_lin & bit));       // the others could not linearize the call in time
  S.c_lin &= ~bit;
  return S.alive ? KEY_OK : KEY_BAD;
}

// ---- the configurations in an LDS table (one workgroup) ----------------------------------------------------------------------------
// For the moments a key's search outgrows the registers (a partition that leaves a dozen writes and cas indeterminate: thousands of
// configurations while one call returns).  A configuration is one word — value << 24 | DEAD | linearized calls (23 slots) — in a pool
// that is also the work list (entries are expanded in the order they were admitted, a workgroup's worth at a time); the pool is
// chained into a hash table over (value, linearized RETURNING calls), so that what dominance has to compare — the sets of linearized
// never-returning calls of one such group — is one chain.  Every thread admits its own candidates: walk the chain (dominated: drop;
// dominating: mark the old entry DEAD), then push with a compare-and-swap on the chain's head; a push that loses the race walks what
// arrived meanwhile and tries again.  Two candidates admitted in the same instant may both survive although one dominates the other
// — that costs work, never the verdict: the search stays exact, dominance is only what keeps it small.
// The pool's arrays are in LDS (pass 2, until a closure outgrows them) or in a slot of an HBM workspace (pass 1, which keeps 14
// histories per CU resident and needs a pool for one closure in a hundred; pass 2 after a closure has outgrown the LDS one): the
// pointers are generic, what the threads hand each other through them is read with agent-scope loads (past the vector L1 when it
// is HBM) and published behind an agent-scope release.  ctr / ops / tws are always LDS.
struct Pool { u32 *cfg, *next, *heads, *outb; u32 *ctr, *ops, *tws; u32 cap, n_heads, out_cap; };
constexpr u32 P_NIL = 0xFFFFFFFFu;   // end of a chain
constexpr u32 P_DEAD = 1u << 23, P_LIN = 0x7FFFFFu;
// P.ctr: 0 entries in the pool, 1 survivors, 2 overflow, 3 command of the first wavefront to the others, 4 (trace) why a key was too wide,
//        5 pending calls, 6 never-returning calls, 7 the returning call's bit, 8-13 (trace) counters, 15 the workspace slot claimed
enum { CMD_CLOSE = 1, CMD_DONE = 2 };
__device__ __forceinline__ u32 pool_ld(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pool_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__host__ __device__ __forceinline__ size_t pool_words(u32 cap, u32 n_heads, u32 out_cap) { return (size_t)cap * 2 + n_heads + out_cap; }
__device__ __forceinline__ void pool_place(Pool &P, u32 *base) { P.cfg = base; P.next = base + P.cap; P.heads = P.next + P.cap; P.outb = P.heads + P.n_heads; }
__device__ __forceinline__ void pool_admit(const Pool &P, u32 c2, u32 info_bits) {
  const u32 val = c2 >> 24, lin = c2 & P_LIN, key2 = lin & ~info_bits, ib = lin & info_bits;
  u32 h = (val * 0x9E3779B1u) ^ (key2 * 0x85EBCA6Bu); h ^= h >> 15;
  u32 *const head = &P.heads[h & (P.n_heads - 1u)];
  u32 mine = P_NIL, stop = P_NIL;
  for (;;) {
    const u32 first = pool_ld(head);
    for (u32 e = first; e != stop; e = pool_ld(&P.next[e])) {
      const u32 w = pool_ld(&P.cfg[e]);
      if ((w & P_DEAD) || (w >> 24) != val || (w & P_LIN & ~info_bits) != key2) continue;
      const u32 eib = w & info_bits;
      if ((eib & ib) == eib) { if (mine != P_NIL) atomicOr(&P.cfg[mine], P_DEAD); return; }   // an existing subset dominates the new one (or is it)
      if ((eib & ib) == ib) atomicOr(&P.cfg[e], P_DEAD);                                       // the new one dominates this one
    }
    if (mine == P_NIL) {
      mine = atomicAdd(&P.ctr[0], 1u);
      if (mine >= P.cap) { P.ctr[2] = 1u; return; }
      P.cfg[mine] = c2;
    }
    P.next[mine] = first;
    pool_release();   // the entry before the head that publishes it
    if (atomicCAS(head, first, mine) == first) return;
    stop = first;   // (what lies below was compared already)
  }
}

// The closure for one returning call, by every thread of the workgroup: the pool holds the configurations to start from, P.ctr[5..7]
// the calls, P.ops / P.tws their operations.  Returns KEY_OK with the survivors (bit stripped) in P.outb[0 .. n_out) — and, when they
// are more than 64, also back in the pool for the next call —, KEY_BAD or KEY_TOO_WIDE (the pool / P.outb overflowed).
__device__ int pool_close(const Pool &P, u32 &n_out) {
  const u32 tid = threadIdx.x, T = blockDim.x;
  const u32 pending = P.ctr[5], info_bits = P.ctr[6], bit = P.ctr[7];
  u32 wptr = 0;
  for (;;) {
    __syncthreads();
    const u32 pn = P.ctr[0], ovf = P.ctr[2];
    __syncthreads();
    if (ovf) return KEY_TOO_WIDE;
    if (wptr >= pn) break;
    // a small level spreads the calls of a configuration over G threads (the latency of a level is the longest chain of admits one
    // thread makes); a large one gives every thread its own configuration
    const u32 span = pn - wptr;
    u32 G = 1; while (G < 16u && span * (G * 2u) <= T) G *= 2u;
    const u32 i = wptr + tid / G, sub = tid & (G - 1u);
    if (i < pn) {
      const u32 c = pool_ld(&P.cfg[i]);
      if (!(c & P_DEAD) && !(c & bit)) {
        const u32 lin_i = c & P_LIN, val_i = c >> 24;
        u32 cand = pending & ~lin_i, nth = 0;
        while (cand) {
          const u32 q = (u32)__builtin_ctz(cand); cand &= cand - 1;
          if ((nth++ & (G - 1u)) != sub) continue;
          const u32 op = P.ops[q];
          if (op >> 31) continue;
          if (((info_bits >> q) & 1u) && (P.tws[q] & ~lin_i)) continue;
          const u32 of = op & 0xFFu, v1 = (op >> 8) & 0xFFu, v2 = (op >> 16) & 0xFFu;
          u32 nv = val_i;
          if (of == MSIM_F_READ) { if (val_i != v1) continue; }
          else if (of == MSIM_F_WRITE) nv = v1;
          else { if (val_i != v1) continue; nv = v2; }
          pool_admit(P, lin_i | (1u << q) | (nv << 24), info_bits);
        }
      }
    }
    wptr = min(pn, wptr + T / G);
  }
  // the survivors are the configurations that linearized the call.  Without its bit their groups stay apart (all of them had it),
  // so they are still minimal and distinct.
  const u32 pn = P.ctr[0];
  for (u32 i = tid; i < pn; i += T) {
    const u32 c = pool_ld(&P.cfg[i]);
    if (!(c & P_DEAD) && (c & bit)) { const u32 o = atomicAdd(&P.ctr[1], 1u); if (o < P.out_cap) P.outb[o] = c & ~bit; else P.ctr[2] = 1u; }
  }
  pool_release();
  __syncthreads();
  n_out = P.ctr[1];
  const u32 ovf = P.ctr[2];
  __syncthreads();
  if (ovf) return KEY_TOO_WIDE;
  if (!n_out) return KEY_BAD;
  if (n_out > 64u) {   // they stay in the pool: rebuilt under their new keys
    for (u32 i = tid; i < P.n_heads; i += T) P.heads[i] = P_NIL;
    if (tid == 0) { P.ctr[0] = 0; P.ctr[1] = 0; }
    pool_release();
    __syncthreads();
    for (u32 i = tid; i < n_out; i += T) pool_admit(P, pool_ld(&P.outb[i]), info_bits);
    __syncthreads();
  }
  return KEY_OK;
}

// pool_close with a way out when the pool overflows: what it holds — configurations reached so far, a sound start for the same closure —
// moves to a slot of the workgroup-sized HBM workspace `g` (claimed here, kept for the rest of the history) and the closure runs again.
struct Grow { u32 *ws; u32 *claim; u32 n_slots, cap, n_heads, out_cap; };   // ws == nullptr: no such workspace (pass 1)
__device__ int pool_close_grow(Pool &P, u32 &n_out, const Grow &g) {
  int st = pool_close(P, n_out);
  if (st != KEY_TOO_WIDE || g.ws == nullptr || P.cap >= g.cap) return st;
  const u32 tid = threadIdx.x, T = blockDim.x;
  if (tid == 0) { const u32 slot = atomicAdd(g.claim, 1u); P.ctr[15] = slot < g.n_slots ? slot : P_NIL; }
  __syncthreads();
  const u32 slot = P.ctr[15], old_n = min(P.ctr[0], P.cap), info_bits = P.ctr[6];
  __syncthreads();
  if (slot == P_NIL) return KEY_TOO_WIDE;
  Pool B = P;
  B.cap = g.cap; B.n_heads = g.n_heads; B.out_cap = g.out_cap;
  pool_place(B, g.ws + (size_t)slot * pool_words(g.cap, g.n_heads, g.out_cap));
  for (u32 i = tid; i < B.n_heads; i += T) B.heads[i] = P_NIL;
  if (tid == 0) { P.ctr[0] = 0; P.ctr[1] = 0; P.ctr[2] = 0; }
  pool_release();
  __syncthreads();
  for (u32 i = tid; i < old_n; i += T) { const u32 c = pool_ld(&P.cfg[i]); if (!(c & P_DEAD)) pool_admit(B, c, info_bits); }
  P = B;
  return pool_close(P, n_out);   // (begins with a barrier)
}

// ---- one key -----------------------------------------------------------------------------------------------------------------------
// Walks the rows of key k (one wavefront: every cross-lane step below is the wavefront's) with the configurations in its registers.
// A closure that outgrows them moves to the pool P — `acquire` provides it at the first need, false = there is none: KEY_TOO_WIDE —
// where the whole workgroup finishes it (pass 1: the wavefront alone; pass 2: it is the FIRST of a workgroup whose other wavefronts
// wait in pool_workers()), and the survivors come back to the registers as soon as 64 lanes hold them again.
template <class Acquire>
__device__ int search_key(const uint4 *r, u32 n, u32 k, u32 lo, u32 hi, const Outcome outcome, u32 lane, Pool &P, Acquire acquire, const Grow &grow) {
  RegSet S; S.c_lin = 0; S.c_val = 0xFFu; S.alive = 1ull;
  const bool prof = P.ctr[8] != 0;   // (developer trace: ctr[9..13] = register closures, their time, pool closures, their time, entries they admitted)
  bool in_pool = false;                               // the configurations are in the pool (more than 64 survived the last call)
  u64 pending = 0, info_bits = 0;
  u32 s_proc = 0, s_op = 0; bool s_ok = false; u64 s_tw = 0;
  for (u32 base = lo & ~63u; base <= hi; base += 64) {
    const u32 idx = base + lane;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (idx < n) row = r[idx];
    const u32 f0 = (row.z >> 2) & 31u;
    const bool reg = idx < n && (row.z >> 12) != MSIM_PROCESS_NEMESIS && (f0 == MSIM_F_READ || f0 == MSIM_F_WRITE || f0 == MSIM_F_CAS) && (row.w & 0xFFu) == k;
    u64 m = __ballot(reg);
    bool fast_ok = true;
    while (m) {
      // ONE configuration, no call open (none that never returns either): the rows that come as whole pairs — an invocation directly followed
      // by its own :ok / :fail — have exactly one linearization, their order.  Every lane checks its own pair against the value the nearest
      // earlier write / cas of the run leaves (a prefix maximum over the lanes) and the run's last one is the register afterwards (round 6;
      // before, every row of such a run went through the loop below: 4000 cycles a row, and such runs are most of a healthy history).  A
      // pair that does not check leaves everything to the search below, which then finds the key not linearizable.
      if (fast_ok && pending == 0 && !in_pool && S.alive && !(S.alive & (S.alive - 1ull))) {
        fast_ok = false;
        const u32 a = (u32)__builtin_ctzll(S.alive), cur = rl(S.c_val, a);
        const u64 below = m & ((1ull << lane) - 1ull);
        const u32 rank = (u32)__popcll(below);
        const u32 pl = below ? 63u - (u32)__builtin_clzll(below) : 0u;
        const u32 pz = l_bperm(row.z, pl);
        const u32 type = row.z & 3u, proc = row.z >> 12;
        const bool mine_row = ((m >> lane) & 1ull) != 0;
        const bool fit = (rank & 1u) ? ((type == MSIM_T_OK || type == MSIM_T_FAIL) && (pz & 3u) == MSIM_T_INVOKE && (pz >> 12) == proc && ((pz >> 2) & 31u) == f0) : type == MSIM_T_INVOKE;
        const u64 viol = __ballot(mine_row && !fit);
        const u32 nclean = (u32)__popcll(viol ? m & ((1ull << (u32)__builtin_ctzll(viol)) - 1ull) : m) & ~1u;
        if (nclean) {
          const bool okc = mine_row && (rank & 1u) && rank < nclean && type == MSIM_T_OK;    // the completion of an :ok pair (a :fail never happened)
          const u32 v1 = (row.w >> 8) & 0xFFu, v2 = (row.w >> 16) & 0xFFu;                    // (an :ok carries the values: search below, `vv`)
          const bool setter = okc && (f0 == MSIM_F_WRITE || f0 == MSIM_F_CAS);
          const u32 sval = f0 == MSIM_F_WRITE ? v1 : v2;
          const u32 ps = l_excl_max(setter ? lane + 1u : 0u);
          const u32 got = l_bperm(sval, ps ? ps - 1u : 0u);
          const u32 before = ps ? got : cur;
          const bool bad = okc && f0 != MSIM_F_WRITE && before != v1;                         // a read sees the value, a cas finds it
          if (!__ballot(bad)) {
            const u64 sm = __ballot(setter);
            if (sm) { const u32 nv = rl(sval, 63u - (u32)__builtin_clzll(sm)); if (lane == a) S.c_val = nv; }
            m &= ~__ballot(mine_row && rank < nclean);
          }
        }
        continue;
      }
      fast_ok = true;
      const u32 j = (u32)__builtin_ctzll(m); m &= m - 1;
      const u32 z = rl(row.z, j), w = rl(row.w, j);
      const u32 jt = z & 3u, jf = (z >> 2) & 31u, jp = z >> 12;
      if (jt == MSIM_T_INVOKE) {
        const u32 oc = outcome_get(outcome, base + j);
        const bool done = (oc >> 31) != 0;
        const u32 ct = oc & 3u;
        if (done && ct == MSIM_T_FAIL) continue;                       // never happened
        const bool ok = done && ct == MSIM_T_OK;
        const u32 vv = ok ? (oc & 0xFFFF00u) : (w & 0xFFFF00u);        // an :ok read carries the value it saw
        const u32 skip = (!ok && jf == MSIM_F_READ) ? 1u : 0u;         // an unfinished read constrains nothing
        if (pending == ~0ull) return KEY_UNKNOWN;
        const u32 s = (u32)__builtin_ctzll(~pending);
        if (in_pool && s >= 23u) { if (lane == 0) P.ctr[4] = 1u; return KEY_TOO_WIDE; }   // (a pool word has 23 slots; the host search has 64)
        const u32 opw = jf | vv | (skip << 31);
        if (!ok && !skip) {
          const u64 same = __ballot(((info_bits >> lane) & 1ull) && s_op == opw);
          if (lane == s) s_tw = same & ((1ull << s) - 1ull);
          else if (((same >> lane) & 1ull) && lane > s) s_tw |= 1ull << s;
        } else if (lane == s) s_tw = 0;
        pending |= 1ull << s;
        if (!ok) info_bits |= 1ull << s;
        if (lane == s) { s_proc = jp; s_op = opw; s_ok = ok; }
        continue;
      }
      if (jt != MSIM_T_OK) continue;
      const u64 sm = __ballot(((pending >> lane) & 1ull) && s_ok && s_proc == jp);
      if (!sm) continue;
      const u32 s = (u32)__builtin_ctzll(sm);
      const u64 bit = 1ull << s;
      int st = KEY_TOO_WIDE;
      u64 ov_lin = 0; u32 ov_val = 0;
      const u64 tq0 = prof ? wall_clock64() : 0;
      if (!in_pool) st = close_regs(S, pending, info_bits, s_op, s_tw, bit, lane, ov_lin, ov_val);
      if (prof && lane == 0) { P.ctr[9]++; P.ctr[10] += (u32)(wall_clock64() - tq0); }
      if (st == KEY_TOO_WIDE) {
        if (!in_pool && !acquire()) return KEY_TOO_WIDE;   // (pass 1: no workspace slot left)
        if (pending >> 23) { if (lane == 0) P.ctr[4] = 1u; return KEY_TOO_WIDE; }
        if (lane < 32u) { P.ops[lane] = s_op; P.tws[lane] = (u32)s_tw; }
        if (lane == 0) { P.ctr[5] = (u32)pending; P.ctr[6] = (u32)info_bits; P.ctr[7] = (u32)bit; }
        if (!in_pool) {   // what the registers hold (and the one that did not fit) goes to the pool; the closure starts over there
          for (u32 i = lane; i < P.n_heads; i += 64) P.heads[i] = P_NIL;
          if (lane == 0) { P.ctr[0] = 0; P.ctr[1] = 0; P.ctr[2] = 0; }
          pool_release();
          __builtin_amdgcn_wave_barrier();
          if ((S.alive >> lane) & 1ull) pool_admit(P, (u32)S.c_lin | (S.c_val << 24), (u32)info_bits);
          if (lane == 0) pool_admit(P, (u32)ov_lin | (ov_val << 24), (u32)info_bits);
        }
        if (lane == 0) P.ctr[3] = CMD_CLOSE;
        __syncthreads();   // (the command barrier: the other wavefronts join)
        u32 n_out = 0;
        const u64 tq1 = prof ? wall_clock64() : 0;
        st = pool_close_grow(P, n_out, grow);
        if (prof && lane == 0) { P.ctr[11]++; P.ctr[12] += (u32)(wall_clock64() - tq1); P.ctr[13] += P.ctr[0]; }
        if (st == KEY_TOO_WIDE) return KEY_TOO_WIDE;
        if (st == KEY_OK) {
          in_pool = n_out > 64u;
          if (!in_pool) {
            const u32 c = lane < n_out ? pool_ld(&P.outb[lane]) : 0u;
            S.c_lin = c & P_LIN; S.c_val = c >> 24; S.alive = n_out == 64u ? ~0ull : (1ull << n_out) - 1ull;
          }
        }
      }
      pending &= ~bit;
      if (st == KEY_BAD) return KEY_BAD;
    }
  }
  return KEY_OK;
}

// the other wavefronts of the workgroup: wait for the first one's closures
__device__ void pool_workers(Pool &P, const Grow &grow) {
  for (;;) {
    __syncthreads();
    if (P.ctr[3] == CMD_DONE) return;
    u32 n_out;
    (void)pool_close_grow(P, n_out, grow);
  }
}

// A history that awaits a wider search says where to go on in its result record: lost_count = the key to resume at (RESUME_NONE: the
// device cannot do this history at all), attempt_count / error_count / stale_count = keys checked / not linearizable / unknown so far.
constexpr u32 RESUME_NONE = 0xFFFFFFFFu;

__device__ void clear_result(msim_check_result &res) {
  res.valid = 0; res.attempt_count = 0; res.stable_count = 0; res.lost_count = 0; res.never_read_count = 0; res.stale_count = 0;
  res.duplicated_count = 0; res.error_count = 0;
  for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = 0;
  res.op_count = 0; res.ok_count = 0; res.fail_count = 0; res.info_count = 0;
}

// pass 1: one wavefront per history; the configurations in its registers, the closures that outgrow them in a slot of the HBM workspace
struct HParams { u32 *pool; u32 *claim; u32 n_slots, cap, n_heads, out_cap, trace; };   // slot = cfg[cap] next[cap] heads[n_heads] outb[out_cap] words
__global__ void __launch_bounds__(64) lin_check_kernel(const LParams p, const HParams hp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *const key_lo = reinterpret_cast<u32 *>(smem);          // [256] first row of the key
  u32 *const key_hi = key_lo + 256;                           // [256] last row of the key
  const Outcome outcome = outcome_at(key_hi + 256, p.table_rows);   // [table_rows]
  const u32 lane = threadIdx.x, inst = p.list ? p.list[blockIdx.x] : blockIdx.x;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.off ? p.off[inst] : (u64)inst * p.stride);
  const u32 n = p.meta ? p.meta[inst].n_rows : (u32)(p.off[inst + 1] - p.off[inst]);
  const u32 flags = p.meta ? p.meta[inst].flags : 0u;

  msim_check_result res;
  clear_result(res);
  if (n > p.table_rows) { if (lane == 0) { res.valid = NEEDS_HOST; res.lost_count = RESUME_NONE; p.out[inst] = res; } return; }
  Pool P;
  P.cap = hp.cap; P.n_heads = hp.n_heads; P.out_cap = hp.out_cap; P.cfg = nullptr; P.next = nullptr; P.heads = nullptr; P.outb = nullptr;
  P.ctr = key_hi + 256 + outcome_bytes(p.table_rows) / 4; P.ops = P.ctr + 16; P.tws = P.ops + 32;
  for (u32 i = lane; i < 256; i += 64) { key_lo[i] = 0xFFFFFFFFu; key_hi[i] = 0; }
  if (lane < 16) P.ctr[lane] = lane == 8 ? hp.trace : 0u;
  const u64 t_start = hp.trace ? wall_clock64() : 0;
  __syncthreads();
  const bool paired = pair_rows(r, n, key_lo, key_hi, outcome, lane, res.op_count, res.ok_count, res.fail_count, res.info_count);
  __syncthreads();
  auto acquire = [&]() -> bool {   // the first closure of this history that needs the pool claims a slot of the workspace (kept to the end)
    if (P.cfg) return true;
    u32 slot = 0;
    if (lane == 0) slot = atomicAdd(hp.claim, 1u);
    slot = (u32)__builtin_amdgcn_readfirstlane((int)slot);
    if (slot >= hp.n_slots) return false;
    pool_place(P, hp.pool + (size_t)slot * pool_words(hp.cap, hp.n_heads, hp.out_cap));
    return true;
  };
  Grow no_grow; no_grow.ws = nullptr; no_grow.claim = nullptr; no_grow.n_slots = 0; no_grow.cap = 0; no_grow.n_heads = 0; no_grow.out_cap = 0;
  u32 n_keys = 0, n_bad = 0, n_unknown = 0, resume = paired ? 256u : RESUME_NONE;
  for (u32 k = 0; k < 256 && paired; k++) {
    const u32 lo = key_lo[k], hi = key_hi[k];
    if (lo == 0xFFFFFFFFu) continue;
    const int st = search_key(r, n, k, lo, hi, outcome, lane, P, acquire, no_grow);
    if (st == KEY_TOO_WIDE) { resume = k; break; }
    n_keys++; n_bad += st == KEY_BAD; n_unknown += st == KEY_UNKNOWN;
  }
  if (lane == 0) {
    res.attempt_count = n_keys;    // keys checked (independent/checker)
    res.error_count = n_bad;       // keys whose history is not linearizable
    if (resume != 256u) { res.valid = NEEDS_HOST; res.lost_count = resume; res.stale_count = n_unknown; }
    else res.valid = flags ? 0u : n_bad ? 0u : n_unknown ? 2u : 1u;
    if (hp.trace) { res.never_read_count = (u32)((wall_clock64() - t_start) / 100u); res.stable_latency_ms[3] = P.ctr[11]; res.stable_latency_ms[4] = P.ctr[12] / 100u; res.stable_count = P.ctr[13]; }   // (developer trace only)
    p.out[inst] = res;
  }
}

// pass 2: one workgroup per history still open, from the key it stopped at: the first wavefront walks the keys with the configurations
// in its registers, the whole workgroup closes the calls that outgrow them in the pool — in LDS (`cap` configurations = what 160 KiB
// hold); the handful of closures that outgrow that too move to a slot of the HBM workspace `grow`
struct WParams { u32 cap, n_heads, out_cap, trace; Grow grow; };
__global__ void __launch_bounds__(1024) lin_check_wg_kernel(const LParams p, const WParams wp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *const key_lo = reinterpret_cast<u32 *>(smem);
  u32 *const key_hi = key_lo + 256;
  const Outcome outcome = outcome_at(key_hi + 256, p.table_rows);
  Pool P;
  P.cap = wp.cap; P.n_heads = wp.n_heads; P.out_cap = wp.out_cap;
  P.ctr = key_hi + 256 + outcome_bytes(p.table_rows) / 4; P.ops = P.ctr + 16; P.tws = P.ops + 32;
  pool_place(P, P.tws + 32);
  const u32 tid = threadIdx.x, lane = tid & 63u, inst = p.list[blockIdx.x];
  msim_check_result res = p.out[inst];
  if (res.valid != NEEDS_HOST || res.lost_count == RESUME_NONE) return;   // decided by an earlier pass / not the device's
  const u32 k0 = res.lost_count;
  const u64 t_start = wp.trace ? wall_clock64() : 0;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.off ? p.off[inst] : (u64)inst * p.stride);
  const u32 n = p.meta ? p.meta[inst].n_rows : (u32)(p.off[inst + 1] - p.off[inst]);
  const u32 flags = p.meta ? p.meta[inst].flags : 0u;
  for (u32 i = tid; i < 256; i += blockDim.x) { key_lo[i] = 0xFFFFFFFFu; key_hi[i] = 0; }
  if (tid < 16) P.ctr[tid] = 0;
  __syncthreads();
  if (tid == 0) P.ctr[8] = wp.trace;
  __syncthreads();
  if (tid >= 64) { pool_workers(P, wp.grow); return; }
  const u64 t_p0 = wp.trace ? wall_clock64() : 0;
  { u32 a, b, c, d; (void)pair_rows(r, n, key_lo, key_hi, outcome, lane, a, b, c, d); }   // (pass 1 paired them: it fits)
  const u32 t_pair = wp.trace ? (u32)(wall_clock64() - t_p0) : 0;
  u32 n_keys = res.attempt_count, n_bad = res.error_count, n_unknown = res.stale_count, resume = 256u;
  for (u32 k = k0; k < 256; k++) {
    const u32 lo = key_lo[k], hi = key_hi[k];
    if (lo == 0xFFFFFFFFu) continue;
    const int st = search_key(r, n, k, lo, hi, outcome, lane, P, []() { return true; }, wp.grow);
    if (st == KEY_TOO_WIDE) { resume = k; break; }
    n_keys++; n_bad += st == KEY_BAD; n_unknown += st == KEY_UNKNOWN;
  }
  if (lane == 0) P.ctr[3] = CMD_DONE;
  __syncthreads();   // (the command barrier: the other wavefronts leave)
  if (tid == 0) {
    res.attempt_count = n_keys; res.error_count = n_bad;
    if (resume != 256u) { res.valid = NEEDS_HOST; res.lost_count = resume; res.stale_count = n_unknown; res.duplicated_count = P.ctr[4]; }   // (developer trace: 1 = more than 23 calls pending, 0 = the pool)
    else { res.valid = flags ? 0u : n_bad ? 0u : n_unknown ? 2u : 1u; res.lost_count = 0; res.stale_count = 0; }
    if (wp.trace) { res.never_read_count = (u32)((wall_clock64() - t_start) / 100u);   // developer trace only: microseconds this workgroup took (100 MHz counter)
      res.stable_latency_ms[0] = t_pair / 100u; res.stable_latency_ms[1] = P.ctr[9]; res.stable_latency_ms[2] = P.ctr[10] / 100u; res.stable_latency_ms[3] = P.ctr[11]; res.stable_latency_ms[4] = P.ctr[12] / 100u; res.stable_count = P.ctr[13]; }
    p.out[inst] = res;
  }
}

// launches the search over `n` histories: pass 1 (a wavefront each: 64 configurations in registers, the closures beyond that in a
// slot of an HBM workspace), then for what it left open pass 2 (a workgroup of 1024 each, the pool in LDS: the largest that fits) and
// the host search (all host threads) for what exceeds even that
int lin_check_dev_run(msim_ctx *ctx, const LParams &lp0, u32 n, u32 max_rows_any, msim_check_result *h_out, hipStream_t st, u32 *n_host) {
  LParams lp = lp0;
  const bool trace = (msim_dev_flags(ctx) & 0x1000u) != 0;   // developer: time the passes
  const bool tiny = (msim_dev_flags(ctx) & 0x2000u) != 0;    // developer / tests: pools small enough that every level is reached, the host search included
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const u32 table_cap = 24u * 1024u;   // rows (51 KiB of LDS)
  lp.table_rows = max_rows_any < table_cap ? max_rows_any : table_cap;
  lp.list = nullptr;
  const size_t lds = 2048 + outcome_bytes(lp.table_rows);
  const size_t fixed = lds + (16 + 32 + 32) * 4;   // + the pools' counters and the calls' operations
  HParams hp;
  hp.cap = tiny ? 66 : 2048; hp.n_heads = tiny ? 32 : 512; hp.out_cap = tiny ? 66 : 1024;
  hp.n_slots = n < 4096u ? n : 4096u;   // (a history claims one when a closure first outgrows the registers; without one it waits for pass 2)
  hp.pool = nullptr; hp.claim = nullptr; hp.trace = trace ? 1u : 0u;
  const size_t ws_bytes = (size_t)hp.n_slots * pool_words(hp.cap, hp.n_heads, hp.out_cap) * 4 + 256;
  MSIM_HIP_TRY(ctx, msim_dev_malloc(&hp.pool, ws_bytes));
  hp.claim = hp.pool + (ws_bytes - 256) / 4;
  struct Ws { u32 *p; ~Ws() { if (p) (void)msim_dev_free(p); } } ws_guard{hp.pool};
  MSIM_HIP_TRY(ctx, hipMemsetAsync(hp.claim, 0, 4, st));
  hipLaunchKernelGGL(lin_check_kernel, dim3(n), dim3(64), fixed, st, lp, hp);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_out, lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<u32> todo;
  for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HOST) todo.push_back(i);
  if (trace) { u32 claimed = 0; (void)hipMemcpy(&claimed, hp.claim, 4, hipMemcpyDeviceToHost);
    { std::vector<u32> us, pc, ad, pt; for (u32 i = 0; i < n; i++) { us.push_back(h_out[i].never_read_count); pc.push_back(h_out[i].stable_latency_ms[3]); pt.push_back(h_out[i].stable_latency_ms[4]); ad.push_back(h_out[i].stable_count); }
      auto q = [&](std::vector<u32> &v, const char *nm) { std::sort(v.begin(), v.end()); std::fprintf(stderr, "[lin-check]   pass 1 %s per history: median %u, 90%% %u, 95%% %u, 97%% %u, 99%% %u, max %u\n", nm, v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 95 / 100], v[v.size() * 97 / 100], v[v.size() * 99 / 100], v.back()); };
      q(us, "microseconds"); q(pc, "pool closures"); q(pt, "microseconds in pool closures"); q(ad, "pool entries admitted"); }
    std::fprintf(stderr, "[lin-check] pass 1 (registers + HBM pools of %u configurations, %zu MB of workspace): %.2f ms, %u histories claimed a pool, %zu of %u still open\n", hp.cap, ws_bytes >> 20, ms(), claimed, todo.size(), n); }
  if (!todo.empty()) {
    std::vector<msim_inst_meta> hm;
    std::vector<uint64_t> ho;
    if (lp.meta) { hm.resize(n); MSIM_HIP_TRY(ctx, hipMemcpy(hm.data(), lp.meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost)); }
    else { ho.resize(n + 1); MSIM_HIP_TRY(ctx, hipMemcpy(ho.data(), lp.off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost)); }
    std::vector<msim_check_result> h2(n);           // the later passes' results (h_out keeps pass 1's until merged)
    std::vector<msim_check_result> hh(todo.size()); // the host's results, by position in `order`
    std::vector<u32> order(todo);
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return h_out[x].info_count > h_out[y].info_count; });
    std::vector<char> host_done(todo.size(), 0);
    std::atomic<size_t> next{0};
    std::atomic<int> copy_err{0};
    std::vector<char> wanted;                       // after the device passes: which histories the host still has to do
    auto host_one = [&](size_t k) {
      const u32 i = order[k];
      const u32 nr = lp.meta ? hm[i].n_rows : (u32)(ho[i + 1] - ho[i]);
      const uint64_t first = lp.meta ? (uint64_t)i * lp.stride : ho[i];
      std::vector<msim_op> rows(nr ? nr : 1);
      if (nr && hipMemcpy(rows.data(), lp.rows + first, (size_t)nr * sizeof(msim_op), hipMemcpyDeviceToHost) != hipSuccess) { copy_err = 1; return; }
      msim_lin_check_instance_host(rows.data(), nr, lp.meta ? hm[i].flags : 0u, &hh[k]);
      host_done[k] = 1;
    };
    u32 *d_list = nullptr, *ws2 = nullptr;
    hipError_t e = msim_dev_malloc(&d_list, todo.size() * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_list, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      lp.list = d_list;
      // pass 2: what 160 KiB of LDS hold beside the pairing table
      WParams wp;
      wp.trace = trace ? 1u : 0u;
      if (tiny) { wp.cap = 72; wp.n_heads = 64; wp.out_cap = 72; }
      else {
        wp.n_heads = 4096; wp.out_cap = 4096;
        const size_t room = 160u * 1024u - fixed - (size_t)(wp.n_heads + wp.out_cap) * 4 - 256;
        wp.cap = (u32)(room / 8) & ~63u;
      }
      const size_t l2 = fixed + pool_words(wp.cap, wp.n_heads, wp.out_cap) * 4;
      // the workspace of the closures that outgrow the LDS pool: a quarter of a million configurations each, for up to 64 histories
      Grow &g = wp.grow;
      g.cap = tiny ? 80u : 262144u; g.n_heads = tiny ? 64u : 65536u; g.out_cap = tiny ? 80u : 65536u;
      g.n_slots = (u32)std::min<size_t>(todo.size(), tiny ? 2 : 64);
      g.ws = nullptr; g.claim = nullptr;
      e = msim_dev_malloc(&ws2, (size_t)g.n_slots * pool_words(g.cap, g.n_heads, g.out_cap) * 4 + 256);
      if (e == hipSuccess) { g.ws = ws2; g.claim = ws2 + (size_t)g.n_slots * pool_words(g.cap, g.n_heads, g.out_cap); e = hipMemsetAsync(g.claim, 0, 4, st); }
      if (e == hipSuccess && l2 > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lin_check_wg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(lin_check_wg_kernel, dim3((u32)todo.size()), dim3(1024), l2, st, lp, wp);
        e = hipGetLastError();
      }
      if (trace && e == hipSuccess) {
        (void)hipMemcpyAsync(h2.data(), lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        size_t left = 0; std::vector<u32> us;
        double a[6] = {0, 0, 0, 0, 0, 0};
        for (u32 i : todo) { left += h2[i].valid == NEEDS_HOST; us.push_back(h2[i].never_read_count); for (int q = 0; q < 5; q++) a[q] += h2[i].stable_latency_ms[q]; a[5] += h2[i].stable_count; }
        std::sort(us.begin(), us.end());
        u32 grown = 0; (void)hipMemcpy(&grown, g.claim, 4, hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[lin-check] pass 2 (pool of %u configurations, %zu B of LDS; %u histories moved on to an HBM pool of %u) done at %.2f ms, %zu histories still open\n", wp.cap, l2, grown, g.cap, ms(), left);
        std::fprintf(stderr, "[lin-check]   sums: pairing %.0f us; %.0f register closures %.0f us; %.0f pool closures %.0f us, %.0f entries admitted\n", a[0], a[1], a[2], a[3], a[4], a[5]);
        std::fprintf(stderr, "[lin-check]   microseconds per workgroup: median %u, 90%% %u, max %u\n", us[us.size() / 2], us[us.size() * 9 / 10], us.back());
      }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h2.data(), lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (ws2) (void)msim_dev_free(ws2);
    wanted.assign(order.size(), 0);
    if (e == hipSuccess) for (size_t k = 0; k < order.size(); k++) wanted[k] = h2[order[k]].valid == NEEDS_HOST;
    else std::fill(wanted.begin(), wanted.end(), 1);   // (the host can still do everything)
    const double t_dev = ms();
    if (d_list) (void)msim_dev_free(d_list);
    u32 n_host_needed = 0;
    for (size_t k = 0; k < order.size(); k++) n_host_needed += wanted[k] != 0;
    if (n_host_needed) {   // what even the HBM pools could not hold (or found no slot): the host search, all host threads
      unsigned nt = msim_host_threads();
      if (nt > n_host_needed) nt = n_host_needed;
      std::vector<std::thread> th;
      const int dev_id = ctx->device;
      for (unsigned w = 0; w < nt; w++)
        th.emplace_back([&]() {
          (void)hipSetDevice(dev_id);
          for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= order.size()) break;
            if (wanted[k]) host_one(k);
          }
        });
      for (auto &x : th) x.join();
    }
    if (copy_err) { ctx->err = "lin-kv check: copying a history to the host failed"; return MSIM_E_HIP; }
    for (size_t k = 0; k < order.size(); k++) {
      const u32 i = order[k];
      if (wanted[k]) { if (!host_done[k]) host_one(k); h_out[i] = hh[k]; MSIM_HIP_TRY(ctx, hipMemcpy(lp.out + i, &h_out[i], sizeof(msim_check_result), hipMemcpyHostToDevice)); }
      else h_out[i] = h2[i];
    }
    if (copy_err) { ctx->err = "lin-kv check: copying a history to the host failed"; return MSIM_E_HIP; }
    if (trace) { u32 slots = 0; for (size_t k = 0; k < order.size(); k++) if (wanted[k] && h2[order[k]].duplicated_count == 1u) slots++;
      std::fprintf(stderr, "[lin-check] device passes done at %.2f ms, %u histories needed the host search (%u of them: more than 23 calls pending on a key)\n", t_dev, n_host_needed, slots); }
    todo.resize(n_host_needed);
  }
  if (trace) std::fprintf(stderr, "[lin-check] done at %.2f ms\n", ms());
  if (n_host) *n_host = (u32)todo.size();
  return MSIM_OK;
}

}  // namespace

// msim_check for lin-kv: the histories of the last run, where they lie in HBM.
int msim_check_lin_kv_device(msim_ctx *ctx) {
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const u32 n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  LParams lp;
  lp.rows = ctx->d_rows; lp.meta = ctx->d_meta; lp.off = nullptr; lp.out = ctx->d_check; lp.stride = ctx->cfg.max_rows; lp.table_rows = 0; lp.list = nullptr;
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev2, ctx->stream));
  const auto t0 = std::chrono::steady_clock::now();
  u32 redone = 0;
  int rc = lin_check_dev_run(ctx, lp, n, ctx->cfg.max_rows, ctx->h_check, ctx->stream, &redone);
  if (rc != MSIM_OK) return rc;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->lin_host_rechecks = redone;
  ctx->checked = true; ctx->check_fetched = true;
  return MSIM_OK;
}

// Checks `n_histories` lin-kv histories given on the host (history i = rows[row_offsets[i] .. row_offsets[i+1])) on device
// `device`; out[i] as msim_check_lin_kv_rows would fill it.
extern "C" int msim_check_lin_kv_batch(int device, const msim_op *rows, const uint64_t *row_offsets, uint32_t n_histories, msim_check_result *out) {
  if (!rows || !row_offsets || !out || n_histories == 0) return MSIM_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_HIP;
  msim_ctx tmp_ctx;   // error reporting in the HIP_TRY macro; the device the host worker threads select
  tmp_ctx.device = device;
  msim_ctx *ctx = &tmp_ctx;
  const uint64_t total = row_offsets[n_histories];
  u32 max_n = 1;
  for (u32 i = 0; i < n_histories; i++) { const uint64_t c = row_offsets[i + 1] - row_offsets[i]; if (c > 0xFFFFFFFFull) return MSIM_E_RANGE; if (c > max_n) max_n = (u32)c; }
  msim_op *d_rows = nullptr; uint64_t *d_off = nullptr; msim_check_result *d_out = nullptr;
  int rc = MSIM_E_HIP;
  do {
    if (msim_dev_malloc(&d_rows, (size_t)(total ? total : 1) * sizeof(msim_op)) != hipSuccess) break;
    if (msim_dev_malloc(&d_off, (size_t)(n_histories + 1) * 8) != hipSuccess) break;
    if (msim_dev_malloc(&d_out, (size_t)n_histories * sizeof(msim_check_result)) != hipSuccess) break;
    if (total && hipMemcpy(d_rows, rows, (size_t)total * sizeof(msim_op), hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_off, row_offsets, (size_t)(n_histories + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
    LParams lp;
    lp.rows = d_rows; lp.meta = nullptr; lp.off = d_off; lp.out = d_out; lp.stride = 0; lp.table_rows = 0; lp.list = nullptr;
    rc = lin_check_dev_run(ctx, lp, n_histories, max_n, out, nullptr, nullptr);
  } while (false);
  if (d_rows) (void)msim_dev_free(d_rows);
  if (d_off) (void)msim_dev_free(d_off);
  if (d_out) (void)msim_dev_free(d_out);
  return rc;
}
