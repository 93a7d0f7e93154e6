// lin_check_dev.hip — per-key linearizability of lin-kv histories on the device (msim_check for MSIM_WL_LIN_KV).
//
// The reference's lin-kv checker is `independent/checker` over Knossos' `checker/linearizable` with a CAS-register model
// (workload/lin_kv.clj:84, [upstream] jepsen.tests.linearizable-register).  lin_check.cpp restates it on the host as just-in-time
// linearization with dominance pruning; this file is the same search laid out for a wavefront, one wavefront per history:
//
//   * a CONFIGURATION (register value, set of pending calls already linearized) lives in the registers of one lane — at most 64
//     at a time; a PENDING CALL (process, f, values, will-it-return) lives in the registers of the lane that carries its bit;
//   * "is this configuration dominated / does it dominate" is one compare per lane and a ballot; a new configuration goes to the
//     first free lane; the worklist is a 64-bit mask; what a step needs from another lane comes with v_readlane;
//   * pass 0 streams the rows once (64 rows = 1 KiB per load), pairs every invocation with its completion (the call's outcome
//     decides how the search treats it from the start: a :fail never happened, an :ok read must see its value, everything else
//     stays pending forever) and notes the row range of each key; pass 1 walks the ranges key by key.
//
// The search is exact, so its verdict is the host's.  A history that needs more than 64 configurations at a time (many calls
// open at once: a partition that keeps every client waiting) is marked and runs again with eight configurations per lane; what
// exceeds that too (or 64 pending calls in the pairing table, more rows than the LDS table covers, a runaway search) goes to the
// host search of lin_check.cpp (lin_check_dev_run below) — never silently approximated.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
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
__device__ __forceinline__ u32 l_wave_sum(u32 v) {
  for (int o = 32; o; o >>= 1) v += (u32)__shfl_xor((int)v, o);
  return v;
}

// ---- pass 0: counts, key ranges, invocation -> completion (one wavefront) ------------------------------------------------------------
// outcome word of an invocation row: bit 31 = has a completion, bits 0-1 its type, bits 8-23 its value bytes 1 and 2.
// Returns false when more than 64 calls are open at once (the pairing table is the wavefront).
__device__ bool pair_rows(const uint4 *r, u32 n, u32 *key_lo, u32 *key_hi, u32 *outcome, u32 lane, u32 &c_inv, u32 &c_ok, u32 &c_fail, u32 &c_info) {
  bool o_used = false; u32 o_proc = 0, o_key = 0, o_row = 0;   // lane = one open call
  c_inv = 0; c_ok = 0; c_fail = 0; c_info = 0;
  bool fits = true;
  for (u32 base = 0; base < n && fits; base += 64) {
    const u32 idx = base + lane;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (idx < n) row = r[idx];
    const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12;
    const bool live = idx < n && proc != MSIM_PROCESS_NEMESIS;
    if (live) { c_inv += type == MSIM_T_INVOKE; c_ok += type == MSIM_T_OK; c_fail += type == MSIM_T_FAIL; c_info += type == MSIM_T_INFO; }
    const bool reg = live && (f == MSIM_F_READ || f == MSIM_F_WRITE || f == MSIM_F_CAS);
    if (reg) { atomicMin(&key_lo[row.w & 0xFFu], idx); atomicMax(&key_hi[row.w & 0xFFu], idx); }
    if (reg && type == MSIM_T_INVOKE) outcome[idx] = 0;
    u64 m = __ballot(reg);
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
        if (lane == s) { o_used = false; outcome[irow] = 0x80000000u | jt | (w & 0xFFFF00u); }
      }
    }
  }
  c_inv = l_wave_sum(c_inv); c_ok = l_wave_sum(c_ok); c_fail = l_wave_sum(c_fail); c_info = l_wave_sum(c_info);
  return fits;
}

enum { KEY_OK = 0, KEY_BAD = 1, KEY_UNKNOWN = 2, KEY_TOO_WIDE = 3 };   // what the search of one key returns

// ---- one key, the configurations in registers (one wavefront) ---------------------------------------------------------------------
// configurations: lane i holds (c_lin[b], c_val[b]) while bit i of alive[b] is set; pending calls: lane s holds slot s
template <int CPL>   // configurations per lane: 64 * CPL at a time
__device__ int search_key_regs(const uint4 *r, u32 n, u32 k, u32 lo, u32 hi, const u32 *outcome, u32 lane) {
  u64 c_lin[CPL]; u32 c_val[CPL]; u64 alive[CPL], expanded[CPL];
#pragma unroll
  for (int b = 0; b < CPL; b++) { c_lin[b] = 0; c_val[b] = 0xFFu; alive[b] = 0; expanded[b] = 0; }
  alive[0] = 1ull;
  u64 pending = 0, info_bits = 0;
  u32 s_proc = 0, s_op = 0; bool s_ok = false;       // s_op = f | v1 << 8 | v2 << 16 | skip << 31
  u64 s_tw = 0;                                      // never-returning call: the never-returning calls in lower slots that are the same operation
  for (u32 base = lo & ~63u; base <= hi; base += 64) {
    const u32 idx = base + lane;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (idx < n) row = r[idx];
    const u32 f0 = (row.z >> 2) & 31u;
    const bool reg = idx < n && (row.z >> 12) != MSIM_PROCESS_NEMESIS && (f0 == MSIM_F_READ || f0 == MSIM_F_WRITE || f0 == MSIM_F_CAS) && (row.w & 0xFFu) == k;
    u64 m = __ballot(reg);
    while (m) {
      const u32 j = (u32)__builtin_ctzll(m); m &= m - 1;
      const u32 z = rl(row.z, j), w = rl(row.w, j);
      const u32 jt = z & 3u, jf = (z >> 2) & 31u, jp = z >> 12;
      if (jt == MSIM_T_INVOKE) {
        const u32 oc = outcome[base + j];
        const bool done = (oc >> 31) != 0;
        const u32 ct = oc & 3u;
        if (done && ct == MSIM_T_FAIL) continue;                       // never happened
        const bool ok = done && ct == MSIM_T_OK;
        const u32 vv = ok ? (oc & 0xFFFF00u) : (w & 0xFFFF00u);        // an :ok read carries the value it saw
        const u32 skip = (!ok && jf == MSIM_F_READ) ? 1u : 0u;         // an unfinished read constrains nothing
        if (pending == ~0ull) return KEY_UNKNOWN;
        const u32 s = (u32)__builtin_ctzll(~pending);
        const u32 opw = jf | vv | (skip << 31);
        if (!ok && !skip) {   // the symmetry of lin_check.cpp: identical never-returning calls, lowest slot first
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
      // every surviving configuration must linearize call s now: close the set under linearizing pending calls first
#pragma unroll
      for (int b = 0; b < CPL; b++) expanded[b] = 0;
      u32 explored = 0;
      for (;;) {
        u64 lin_i = 0; u32 val_i = 0; bool found = false;
#pragma unroll
        for (int b = 0; b < CPL; b++) {
          const u64 todo = alive[b] & ~expanded[b];
          if (!found && todo) {
            const u32 i = (u32)__builtin_ctzll(todo);
            expanded[b] |= 1ull << i;
            lin_i = ((u64)rl((u32)(c_lin[b] >> 32), i) << 32) | rl((u32)c_lin[b], i);
            val_i = rl(c_val[b], i);
            found = true;
          }
        }
        if (!found) break;
        if (lin_i & bit) continue;
        if (++explored > EXPLORE_LIMIT) return KEY_TOO_WIDE;
        u64 cand = pending & ~lin_i;
        while (cand) {
          const u32 q = (u32)__builtin_ctzll(cand); cand &= cand - 1;
          const u32 op = rl(s_op, q);
          if (op >> 31) continue;
          if ((info_bits >> q) & 1ull) { const u64 tw = ((u64)rl((u32)(s_tw >> 32), q) << 32) | rl((u32)s_tw, q); if (tw & ~lin_i) continue; }
          const u32 of = op & 0xFFu, v1 = (op >> 8) & 0xFFu, v2 = (op >> 16) & 0xFFu;
          u32 nv = val_i;
          if (of == MSIM_F_READ) { if (val_i != v1) continue; }          // only :ok reads are stepped; they must see the current value
          else if (of == MSIM_F_WRITE) nv = v1;
          else { if (val_i != v1) continue; nv = v2; }                   // cas [v v']
          const u64 lin2 = lin_i | (1ull << q);
          // Dominance: for equal (value, linearized returning calls), a configuration that has linearized FEWER never-returning
          // calls can still do everything the other can.  Keep only the minimal ones.
          const u64 ib2 = lin2 & info_bits, key2 = lin2 & ~info_bits;
          u64 killm[CPL]; bool dominated = false;
#pragma unroll
          for (int b = 0; b < CPL; b++) {
            killm[b] = 0;
            if (alive[b]) {   // (uniform: banks fill in order, the empty ones cost nothing)
              const bool same = ((alive[b] >> lane) & 1ull) && c_val[b] == nv && (c_lin[b] & ~info_bits) == key2;
              const u64 my_ib = c_lin[b] & info_bits;
              dominated |= __ballot(same && (my_ib & ib2) == my_ib) != 0;   // an existing subset dominates the new one
              killm[b] = __ballot(same && (my_ib & ib2) == ib2);            // the new one dominates these
            }
          }
          if (dominated) continue;
          bool placed = false;
#pragma unroll
          for (int b = 0; b < CPL; b++) {
            alive[b] &= ~killm[b];
            if (!placed && alive[b] != ~0ull) {
              const u32 fl = (u32)__builtin_ctzll(~alive[b]);
              if (lane == fl) { c_lin[b] = lin2; c_val[b] = nv; }
              alive[b] |= 1ull << fl; expanded[b] &= ~(1ull << fl);
              placed = true;
            }
          }
          if (!placed) return KEY_TOO_WIDE;
        }
      }
      u64 any = 0;
#pragma unroll
      for (int b = 0; b < CPL; b++) if (alive[b]) {
        alive[b] = __ballot(((alive[b] >> lane) & 1ull) && (c_lin[b] & bit));   // the others could not linearize the call in time
        c_lin[b] &= ~bit;
        any |= alive[b];
      }
      pending &= ~bit;
      if (!any) return KEY_BAD;
    }
  }
  return KEY_OK;
}

// ---- one key, the configurations in an LDS table (one workgroup) --------------------------------------------------------------------
// For the keys whose search outgrows the registers (a partition that leaves a dozen writes and cas indeterminate: thousands of
// configurations while one call returns).  A configuration is one word — value << 24 | DEAD | linearized calls (23 slots) — in a pool
// that is also the work list (entries are expanded in the order they were admitted, a workgroup's worth at a time); the pool is
// chained into a hash table over (value, linearized RETURNING calls), so that what dominance has to compare — the sets of linearized
// never-returning calls of one such group — is one chain.  Every thread admits its own candidates: walk the chain (dominated: drop;
// dominating: mark the old entry DEAD), then push with a compare-and-swap on the chain's head; a push that loses the race walks what
// arrived meanwhile and tries again.  Two candidates admitted in the same instant may both survive although one dominates the other
// — that costs work, never the verdict: the search stays exact, dominance is only what keeps it small.
struct LdsPool { u32 *cfg; unsigned short *next; u32 *heads; u32 *outb; u32 *ctr; u32 *ops; u32 *tws; u32 cap, n_heads, out_cap; };
constexpr u32 P_NIL = 0xFFFFu, P_DEAD = 1u << 23, P_LIN = 0x7FFFFFu, P_SLOTS = 23u;

__device__ int search_key_lds(const uint4 *r, u32 n, u32 k, u32 lo, u32 hi, const u32 *outcome, const LdsPool P) {
  const u32 tid = threadIdx.x, lane = tid & 63u, T = blockDim.x;
  u32 pending = 0, info_bits = 0;                    // uniform over the workgroup: every wavefront walks the same rows
  u32 s_proc = 0, s_op = 0, s_tw = 0; bool s_ok = false;   // lane s of EVERY wavefront holds slot s; P.ops / P.tws mirror them for the divergent reads
  auto admit = [&](u32 c2) {
    const u32 val = c2 >> 24, lin = c2 & P_LIN, key2 = lin & ~info_bits, ib = lin & info_bits;
    u32 h = (val * 0x9E3779B1u) ^ (key2 * 0x85EBCA6Bu); h ^= h >> 15;
    u32 *const head = &P.heads[h & (P.n_heads - 1u)];
    u32 mine = P_NIL, stop = P_NIL;
    for (;;) {
      const u32 first = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      for (u32 e = first; e != stop; e = P.next[e]) {
        const u32 w = __hip_atomic_load(&P.cfg[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
      P.next[mine] = (unsigned short)first;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the entry before the head that publishes it
      if (atomicCAS(head, first, mine) == first) return;
      stop = first;   // (what lies below was compared already)
    }
  };
  auto clear = [&]() { for (u32 i = tid; i < P.n_heads; i += T) P.heads[i] = P_NIL; if (tid == 0) { P.ctr[0] = 0; P.ctr[1] = 0; P.ctr[2] = 0; } };
  __syncthreads();
  clear();
  __syncthreads();
  if (tid == 0) admit(0xFFu << 24);
  for (u32 base = lo & ~63u; base <= hi; base += 64) {
    const u32 idx = base + lane;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (idx < n) row = r[idx];
    const u32 f0 = (row.z >> 2) & 31u;
    const bool reg = idx < n && (row.z >> 12) != MSIM_PROCESS_NEMESIS && (f0 == MSIM_F_READ || f0 == MSIM_F_WRITE || f0 == MSIM_F_CAS) && (row.w & 0xFFu) == k;
    u64 m = __ballot(reg);
    while (m) {
      const u32 j = (u32)__builtin_ctzll(m); m &= m - 1;
      const u32 z = rl(row.z, j), w = rl(row.w, j);
      const u32 jt = z & 3u, jf = (z >> 2) & 31u, jp = z >> 12;
      if (jt == MSIM_T_INVOKE) {
        const u32 oc = outcome[base + j];
        const bool done = (oc >> 31) != 0;
        const u32 ct = oc & 3u;
        if (done && ct == MSIM_T_FAIL) continue;
        const bool ok = done && ct == MSIM_T_OK;
        const u32 vv = ok ? (oc & 0xFFFF00u) : (w & 0xFFFF00u);
        const u32 skip = (!ok && jf == MSIM_F_READ) ? 1u : 0u;
        if ((pending & P_LIN) == P_LIN) return KEY_TOO_WIDE;   // (the host search has 64 slots)
        const u32 s = (u32)__builtin_ctz(~pending);
        const u32 opw = jf | vv | (skip << 31);
        if (!ok && !skip) {
          const u32 same = (u32)__ballot(lane < 32u && ((info_bits >> lane) & 1u) && s_op == opw);
          if (lane == s) s_tw = same & ((1u << s) - 1u);
          else if (lane < 32u && ((same >> lane) & 1u) && lane > s) s_tw |= 1u << s;
        } else if (lane == s) s_tw = 0;
        pending |= 1u << s;
        if (!ok) info_bits |= 1u << s;
        if (lane == s) { s_proc = jp; s_op = opw; s_ok = ok; }
        if (lane < 32u) { P.ops[lane] = s_op; P.tws[lane] = s_tw; }   // (every wavefront writes the same words; read after the next barrier)
        continue;
      }
      if (jt != MSIM_T_OK) continue;
      const u32 sm = (u32)__ballot(lane < 32u && ((pending >> lane) & 1u) && s_ok && s_proc == jp);
      if (!sm) continue;
      const u32 s = (u32)__builtin_ctz(sm);
      const u32 bit = 1u << s;
      // close the set under linearizing pending calls: the pool is the work list
      u32 wptr = 0;
      for (;;) {
        __syncthreads();
        const u32 pn = P.ctr[0], ovf = P.ctr[2];
        __syncthreads();
        if (ovf) return KEY_TOO_WIDE;
        if (wptr >= pn) break;
        const u32 i = wptr + tid;
        if (i < pn) {
          const u32 c = P.cfg[i];
          if (!(c & P_DEAD) && !(c & bit)) {
            const u32 lin_i = c & P_LIN, val_i = c >> 24;
            u32 cand = pending & ~lin_i;
            while (cand) {
              const u32 q = (u32)__builtin_ctz(cand); cand &= cand - 1;
              const u32 op = P.ops[q];
              if (op >> 31) continue;
              if (((info_bits >> q) & 1u) && (P.tws[q] & ~lin_i)) continue;
              const u32 of = op & 0xFFu, v1 = (op >> 8) & 0xFFu, v2 = (op >> 16) & 0xFFu;
              u32 nv = val_i;
              if (of == MSIM_F_READ) { if (val_i != v1) continue; }
              else if (of == MSIM_F_WRITE) nv = v1;
              else { if (val_i != v1) continue; nv = v2; }
              admit(lin_i | (1u << q) | (nv << 24));
            }
          }
        }
        wptr = min(pn, wptr + T);
      }
      // the survivors are the configurations that linearized the call; without its bit they are compared anew
      {
        const u32 pn = P.ctr[0];
        for (u32 i = tid; i < pn; i += T) {
          const u32 c = P.cfg[i];
          if (!(c & P_DEAD) && (c & bit)) { const u32 o = atomicAdd(&P.ctr[1], 1u); if (o < P.out_cap) P.outb[o] = c & ~bit; else P.ctr[2] = 1u; }
        }
        __syncthreads();
        const u32 no = P.ctr[1], ovf = P.ctr[2];
        __syncthreads();
        if (ovf) return KEY_TOO_WIDE;
        pending &= ~bit;
        if (!no) return KEY_BAD;
        clear();
        __syncthreads();
        for (u32 i = tid; i < no; i += T) admit(P.outb[i]);
      }
    }
  }
  return KEY_OK;
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

// pass 1: one wavefront per history, 64 configurations
__global__ void __launch_bounds__(64) lin_check_kernel(const LParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *const key_lo = reinterpret_cast<u32 *>(smem);          // [256] first row of the key
  u32 *const key_hi = key_lo + 256;                           // [256] last row of the key
  u32 *const outcome = key_hi + 256;                          // [table_rows]
  const u32 lane = threadIdx.x, inst = p.list ? p.list[blockIdx.x] : blockIdx.x;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.off ? p.off[inst] : (u64)inst * p.stride);
  const u32 n = p.meta ? p.meta[inst].n_rows : (u32)(p.off[inst + 1] - p.off[inst]);
  const u32 flags = p.meta ? p.meta[inst].flags : 0u;

  msim_check_result res;
  clear_result(res);
  if (n > p.table_rows) { if (lane == 0) { res.valid = NEEDS_HOST; res.lost_count = RESUME_NONE; p.out[inst] = res; } return; }
  for (u32 i = lane; i < 256; i += 64) { key_lo[i] = 0xFFFFFFFFu; key_hi[i] = 0; }
  __syncthreads();
  const bool paired = pair_rows(r, n, key_lo, key_hi, outcome, lane, res.op_count, res.ok_count, res.fail_count, res.info_count);
  __syncthreads();
  u32 n_keys = 0, n_bad = 0, n_unknown = 0, resume = paired ? 256u : RESUME_NONE;
  for (u32 k = 0; k < 256 && paired; k++) {
    const u32 lo = key_lo[k], hi = key_hi[k];
    if (lo == 0xFFFFFFFFu) continue;
    const int st = search_key_regs<1>(r, n, k, lo, hi, outcome, lane);
    if (st == KEY_TOO_WIDE) { resume = k; break; }
    n_keys++; n_bad += st == KEY_BAD; n_unknown += st == KEY_UNKNOWN;
  }
  if (lane == 0) {
    res.attempt_count = n_keys;    // keys checked (independent/checker)
    res.error_count = n_bad;       // keys whose history is not linearizable
    if (resume != 256u) { res.valid = NEEDS_HOST; res.lost_count = resume; res.stale_count = n_unknown; }
    else res.valid = flags ? 0u : n_bad ? 0u : n_unknown ? 2u : 1u;
    p.out[inst] = res;
  }
}

// passes 2 and 3: one workgroup per history that pass 1 left open, from the key it stopped at; the registers of the first wavefront
// first, the LDS table (pool of `cap` configurations) for the keys that outgrow them
struct WParams { u32 cap, n_heads, out_cap; };
__global__ void __launch_bounds__(256) lin_check_wg_kernel(const LParams p, const WParams wp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *const key_lo = reinterpret_cast<u32 *>(smem);
  u32 *const key_hi = key_lo + 256;
  u32 *const outcome = key_hi + 256;
  LdsPool P;
  P.cap = wp.cap; P.n_heads = wp.n_heads; P.out_cap = wp.out_cap;
  P.cfg = outcome + p.table_rows; P.heads = P.cfg + wp.cap; P.outb = P.heads + wp.n_heads; P.ctr = P.outb + wp.out_cap; P.ops = P.ctr + 8; P.tws = P.ops + 32;
  P.next = reinterpret_cast<unsigned short *>(P.tws + 32);
  const u32 tid = threadIdx.x, lane = tid & 63u, inst = p.list[blockIdx.x];
  msim_check_result res = p.out[inst];
  if (res.valid != NEEDS_HOST || res.lost_count == RESUME_NONE) return;   // decided by an earlier pass / not the device's
  const u32 k0 = res.lost_count;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.off ? p.off[inst] : (u64)inst * p.stride);
  const u32 n = p.meta ? p.meta[inst].n_rows : (u32)(p.off[inst + 1] - p.off[inst]);
  const u32 flags = p.meta ? p.meta[inst].flags : 0u;
  for (u32 i = tid; i < 256; i += blockDim.x) { key_lo[i] = 0xFFFFFFFFu; key_hi[i] = 0; }
  __syncthreads();
  if (tid < 64) { u32 a, b, c, d; (void)pair_rows(r, n, key_lo, key_hi, outcome, lane, a, b, c, d); }   // (pass 1 paired them: it fits)
  __syncthreads();
  u32 n_keys = res.attempt_count, n_bad = res.error_count, n_unknown = res.stale_count, resume = 256u;
  for (u32 k = k0; k < 256; k++) {
    const u32 lo = key_lo[k], hi = key_hi[k];
    if (lo == 0xFFFFFFFFu) continue;
    int st = KEY_TOO_WIDE;
    if (k != k0) {   // (the key pass 1 stopped at is known not to fit the registers)
      if (tid < 64) { st = search_key_regs<1>(r, n, k, lo, hi, outcome, lane); if (tid == 0) P.ctr[3] = (u32)st; }
      __syncthreads();
      st = (int)P.ctr[3];
      __syncthreads();
    }
    if (st == KEY_TOO_WIDE) st = search_key_lds(r, n, k, lo, hi, outcome, P);
    if (st == KEY_TOO_WIDE) { resume = k; break; }
    n_keys++; n_bad += st == KEY_BAD; n_unknown += st == KEY_UNKNOWN;
  }
  if (tid == 0) {
    res.attempt_count = n_keys; res.error_count = n_bad;
    if (resume != 256u) { res.valid = NEEDS_HOST; res.lost_count = resume; res.stale_count = n_unknown; }
    else { res.valid = flags ? 0u : n_bad ? 0u : n_unknown ? 2u : 1u; res.lost_count = 0; res.stale_count = 0; }
    p.out[inst] = res;
  }
}

// launches the search over `n` histories: pass 1 (a wavefront each, 64 configurations in registers), then for what it left open
// passes 2 and 3 (a workgroup each, the LDS table: a small pool at several workgroups per CU, then the largest that fits), and the
// host search (all host threads, overlapped with passes 2 and 3) for what exceeds even that
int lin_check_dev_run(msim_ctx *ctx, const LParams &lp0, u32 n, u32 max_rows_any, msim_check_result *h_out, hipStream_t st, u32 *n_host) {
  LParams lp = lp0;
  const bool trace = (msim_dev_flags(ctx) & 0x1000u) != 0;   // developer: time the passes
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const u32 table_cap = (60u * 1024u - 2048u) / 4u;
  lp.table_rows = max_rows_any < table_cap ? max_rows_any : table_cap;
  lp.list = nullptr;
  const size_t lds = 2048 + (size_t)lp.table_rows * 4;
  hipLaunchKernelGGL(lin_check_kernel, dim3(n), dim3(64), lds, st, lp);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_out, lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<u32> todo;
  for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HOST) todo.push_back(i);
  if (trace) std::fprintf(stderr, "[lin-check] pass 1 (64 configurations): %.2f ms, %zu of %u histories marked\n", ms(), todo.size(), n);
  if (!todo.empty()) {
    // While the device works through the marked histories, the host cores already search them — those with the most indeterminate
    // calls first: they are the likeliest to exceed the device's pools too — so that what the device leaves over is mostly done
    // by the time it is known.  Whichever side finishes a history first, the result is the same (both searches are exact).
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
    std::atomic<bool> device_done{false};
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
    // (the speculative host search only pays where the device may leave something over: it is started for the histories the device
    // cannot take at all, and otherwise after the passes)
    u32 *d_list = nullptr;
    hipError_t e = hipMalloc(&d_list, todo.size() * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_list, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      lp.list = d_list;
      // pass 2: a pool of 2048 configurations; pass 3: what 160 KiB of LDS hold beside the pairing table
      const size_t fixed = lds + (8 + 32 + 32) * 4;
      for (int pass = 2; pass <= 3 && e == hipSuccess; pass++) {
        WParams wp;
        const bool tiny = (msim_dev_flags(ctx) & 0x2000u) != 0;   // developer / tests: pools small enough that every level is reached, the host search included
        if (pass == 2) { wp.cap = tiny ? 128 : 2048; wp.n_heads = tiny ? 64 : 1024; wp.out_cap = tiny ? 128 : 1024; }
        else if (tiny) { wp.cap = 512; wp.n_heads = 256; wp.out_cap = 256; }
        else {
          wp.n_heads = 4096; wp.out_cap = 4096;
          const size_t room = 160u * 1024u - fixed - (size_t)(wp.n_heads + wp.out_cap) * 4 - 256;
          wp.cap = (u32)std::min<size_t>(room / 6, 0xFFF0u) & ~63u;
        }
        const size_t l2 = fixed + (size_t)(wp.cap + wp.n_heads + wp.out_cap) * 4 + (size_t)wp.cap * 2;
        if (l2 > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lin_check_wg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(lin_check_wg_kernel, dim3((u32)todo.size()), dim3(256), l2, st, lp, wp);
        e = hipGetLastError();
        if (trace && e == hipSuccess) {
          (void)hipMemcpyAsync(h2.data(), lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st);
          (void)hipStreamSynchronize(st);
          size_t left = 0; for (u32 i : todo) left += h2[i].valid == NEEDS_HOST;
          std::fprintf(stderr, "[lin-check] pass %d (pool of %u configurations, %zu B of LDS) done at %.2f ms, %zu histories still open\n", pass, wp.cap, l2, ms(), left);
        }
      }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h2.data(), lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    wanted.assign(order.size(), 0);
    if (e == hipSuccess) for (size_t k = 0; k < order.size(); k++) wanted[k] = h2[order[k]].valid == NEEDS_HOST;
    else std::fill(wanted.begin(), wanted.end(), 1);   // (the host can still do everything)
    device_done = true;
    const double t_dev = ms();
    if (d_list) (void)hipFree(d_list);
    u32 n_host_needed = 0;
    for (size_t k = 0; k < order.size(); k++) n_host_needed += wanted[k] != 0;
    if (n_host_needed) {
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
    if (trace) std::fprintf(stderr, "[lin-check] device passes done at %.2f ms, %u histories needed the host search\n", t_dev, n_host_needed);
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
    if (hipMalloc(&d_rows, (size_t)(total ? total : 1) * sizeof(msim_op)) != hipSuccess) break;
    if (hipMalloc(&d_off, (size_t)(n_histories + 1) * 8) != hipSuccess) break;
    if (hipMalloc(&d_out, (size_t)n_histories * sizeof(msim_check_result)) != hipSuccess) break;
    if (total && hipMemcpy(d_rows, rows, (size_t)total * sizeof(msim_op), hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_off, row_offsets, (size_t)(n_histories + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
    LParams lp;
    lp.rows = d_rows; lp.meta = nullptr; lp.off = d_off; lp.out = d_out; lp.stride = 0; lp.table_rows = 0; lp.list = nullptr;
    rc = lin_check_dev_run(ctx, lp, n_histories, max_n, out, nullptr, nullptr);
  } while (false);
  if (d_rows) (void)hipFree(d_rows);
  if (d_off) (void)hipFree(d_off);
  if (d_out) (void)hipFree(d_out);
  return rc;
}
