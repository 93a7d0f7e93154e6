// rw_check_dev.hip — the rw-register analysis of txn-rw-register histories on the device (msim_check for MSIM_WL_TXN_RW_REGISTER;
// SURVEY.md §8f rank 1).  What the reference wires in at workload/txn_rw_register.clj:138-168 is [upstream] elle.rw-register through
// jepsen.tests.cycle.wr (:wfr-keys? true), judged by --consistency-models (core.clj:115-121: read-committed for the demo node);
// txn_check.cpp (check_rw) restates it on the host.
//
// The histories of the reference's own demo node (demo/clojure/txn_rw_register_hat.clj, "highly available transactions") are full of
// cycles the consistency model ALLOWS (G-single, G2, their realtime variants), so "prove the history clean or hand it to the host" —
// what txn_check_dev.hip does for list-append — would hand nearly everything over.  This pass instead proves, one wavefront per history,
// that NOTHING THE CONSISTENCY MODEL PROSCRIBES is present:
//   * the non-cycle anomalies exactly as the host finds them (duplicate writes, internal, G1a, G1b, cyclic version orders);
//   * the cycle anomalies by Kahn's algorithm over the edge kinds whose cycles the model proscribes: read-uncommitted ww (G0),
//     read-committed ww + wr (G0, G1c); the stronger models every kind (ww, wr, rw, and the realtime order for strict-serializable) —
//     there an acyclic graph is the only thing this pass can prove, anything else goes to the host.
// A history it cannot prove valid — a proscribed anomaly, a cycle in the subgraph, a shape beyond the capacities below — is handed to
// check_rw, whose verdict and anomaly set are then the result.  For a history it does prove valid the result carries :valid? and the
// counts of the host's analysis, the non-cycle anomalies it saw (none of them proscribed) and the edges it built; the ALLOWED cycle classes
// are not searched for (msim_check_rw_rows gives the full classification of one history).
//
// Steps (lane = transaction unless said otherwise), tables in an HBM workspace as in txn_check_kernel:
//   A  rows -> transactions, completions paired by process (the walk of txn_check_kernel);
//   B  key / value ranges; C  writer table (key, value) -> transaction by compare-and-swap, versions seen per key;
//   D  per :ok transaction: internal consistency, G1a / G1b, wr edges, "writes follow reads" version edges (64-bit sets by atomics);
//   E  per key (lane = key): nil precedes every version; a cyclic version order is found by peeling the versions without predecessor;
//      ww edges along the version order; rw edges from the external reads (stronger models only);
//   F  Kahn's algorithm over the edges built.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "engine_internal.h"

void msim_rw_check_instance_host(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, uint32_t cm,
                                 msim_check_result *res);   // txn_check.cpp

namespace {

constexpr u32 NEEDS_HOST = 3u;
constexpr u32 NONE = 0xFFFFFFFFu;
constexpr u32 KMAX = 4096u;      // keys per history
constexpr u32 WMAX = 65536u;     // writer table entries (keys x values)

struct RParams {
  const msim_op *rows; const u32 *payload; const msim_inst_meta *meta;
  const uint64_t *row_off, *pay_off;   // (null: history i at i * max_rows / i * max_pay)
  msim_check_result *out;
  u32 *ws;                       // workspace, ws_words per history of this launch
  uint64_t ws_words;
  u32 max_rows, max_pay, nmax, emax, first, cm, proscribed;
  u32 kmax, wmax;                // keys / writer-table entries the workspace of a history holds (<= KMAX / WMAX: what the configuration can name)
};

__device__ __forceinline__ u32 r_rl(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u32 r_sum(u32 v) { for (int o = 32; o; o >>= 1) v += (u32)__shfl_xor((int)v, o); return v; }
__device__ __forceinline__ u32 r_max(u32 v) { for (int o = 32; o; o >>= 1) v = max(v, (u32)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ u32 r_or(u32 v) { for (int o = 32; o; o >>= 1) v |= (u32)__shfl_xor((int)v, o); return v; }
__device__ __forceinline__ u32 r_excl_scan(u32 v, u32 lane) {
  u32 x = v;
  for (int o = 1; o < 64; o <<= 1) { const u32 y = (u32)__shfl_up((int)x, o); if (lane >= (u32)o) x += y; }
  return x - v;
}
// one micro-op = one payload word: f (1 = write), key, value (0xFF: a read of nil)
#define M_F(w_) ((w_) & 1u)
#define M_KEY(w_) (((w_) >> 1) & 0x7FFFu)
#define M_VAL(w_) (((w_) >> 16) & 0xFFu)

__global__ void __launch_bounds__(64) rw_check_kernel(const RParams p) {
  const u32 lane = threadIdx.x, hist = p.first + blockIdx.x;
  const u64 lt = (1ull << lane) - 1ull;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.row_off ? p.row_off[hist] : (u64)hist * p.max_rows);
  const u32 *const pay = p.payload + (p.pay_off ? p.pay_off[hist] : (u64)hist * p.max_pay);
  const u32 n_rows = p.meta ? p.meta[hist].n_rows : (u32)(p.row_off[hist + 1] - p.row_off[hist]);
  const u32 n_words = p.meta ? p.meta[hist].n_payload_words : (u32)(p.pay_off[hist + 1] - p.pay_off[hist]);
  const u32 flags = p.meta ? p.meta[hist].flags : 0u;
  const u32 NM = p.nmax;
  const bool strong = p.cm <= MSIM_CM_SNAPSHOT_ISOLATION;            // every edge kind counts
  const bool want_rt = p.cm == MSIM_CM_STRICT_SERIALIZABLE;
  const bool want_wr = p.cm != MSIM_CM_READ_UNCOMMITTED;
  u32 *const ws = p.ws + (u64)blockIdx.x * p.ws_words;
  u32 *const t_inv = ws, *const t_cmp = t_inv + NM, *const t_off = t_cmp + NM, *const t_lt = t_off + NM;   // t_lt: words | type << 16
  u32 *const t_first = t_lt + NM;            // transactions invoked before this one's completion row
  u32 *const sm = t_first + NM, *const smf = sm + NM + 1;
  u32 *const indeg = smf + NM + 1, *const off = indeg + NM;   // off [NM + 1]
  u32 *const cur = off + NM + 1, *const queue = cur + NM;
  u32 *const vseen = queue + NM;             // [kmax][2] versions of the key that exist (bit v; bit 0 = nil)
  u32 *const writer = vseen + 2 * p.kmax;    // [wmax]
  u32 *const vsucc = writer + p.wmax;        // [wmax][2] successors of (key, version) in the key's version order
  u32 *const adj = vsucc + 2 * p.wmax;       // [emax]

  msim_check_result res;
  res.valid = NEEDS_HOST; res.attempt_count = 0; res.stable_count = 0; res.lost_count = 0; res.never_read_count = 0; res.stale_count = 0;
  res.duplicated_count = 0; res.error_count = 0;
  for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = 0;
  res.op_count = 0; res.ok_count = 0; res.fail_count = 0; res.info_count = 0;
#define TO_HOST() do { if (lane == 0) p.out[hist] = res; return; } while (0)
  if (n_words >= (1u << 24)) TO_HOST();

  // ---- A: transactions (txn_check_kernel's pairing) ---------------------------------------------------------------------------------
  u32 n = 0, c_ok = 0, c_fail = 0, c_info = 0;
  {
    bool o_used = false; u32 o_proc = 0, o_txn = 0, o_len = 0; bool bad = false;   // lane = one open call (o_len: words of its request)
    for (u32 base = 0; base < n_rows; base += 64) {
      const u32 idx = base + lane;
      uint4 row = make_uint4(0, 0, 0, 0);
      if (idx < n_rows) row = r[idx];
      const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12, len = row.y >> 16, woff = row.w;
      const bool is = idx < n_rows && proc != MSIM_PROCESS_NEMESIS && f == MSIM_F_TXN;
      if (__ballot(is && (u64)woff + len > n_words)) { bad = true; break; }
      const bool inv = is && type == MSIM_T_INVOKE;
      const u64 im = __ballot(inv);
      const u32 my_t = n + (u32)__popcll(im & lt);
      if (n + (u32)__popcll(im) > NM) { bad = true; break; }
      if (inv) { t_inv[my_t] = idx; t_cmp[my_t] = NONE; t_off[my_t] = woff; t_lt[my_t] = len | (MSIM_T_INFO << 16); t_first[my_t] = 0; }   // never completed = indeterminate
      u64 m = __ballot(is);
      while (m) {
        const u32 j = (u32)__builtin_ctzll(m); m &= m - 1;
        const u32 z = r_rl(row.z, j);
        const u32 jt = z & 3u, jp = z >> 12;
        const u64 hit = __ballot(o_used && o_proc == jp);
        if (jt == MSIM_T_INVOKE) {
          u32 s;
          if (hit) s = (u32)__builtin_ctzll(hit);
          else { const u64 used = __ballot(o_used); if (used == ~0ull) { bad = true; break; } s = (u32)__builtin_ctzll(~used); }
          const u32 tj = n + (u32)__popcll(im & ((1ull << j) - 1ull));
          const u32 jl = r_rl(row.y, j) >> 16;
          if (lane == s) { o_used = true; o_proc = jp; o_txn = tj; o_len = jl; }
        } else if (hit) {
          const u32 s = (u32)__builtin_ctzll(hit);
          const u32 id = r_rl(o_txn, s), ilen = r_rl(o_len, s);
          if (lane == s) o_used = false;
          if (lane == j) {
            t_cmp[id] = idx; t_first[id] = n + (u32)__popcll(im & ((1ull << j) - 1ull));
            if (jt == MSIM_T_OK) { t_off[id] = woff; t_lt[id] = len | (MSIM_T_OK << 16); }   // the completed form replaces the requested one
            else t_lt[id] = ilen | (jt << 16);
          }
          c_ok += jt == MSIM_T_OK; c_fail += jt == MSIM_T_FAIL; c_info += jt == MSIM_T_INFO;
        }
      }
      if (bad) break;
      n += (u32)__popcll(im);
    }
    if (bad) TO_HOST();
  }
  __syncthreads();
  res.op_count = n; res.attempt_count = n; res.ok_count = c_ok; res.stable_count = c_ok; res.fail_count = c_fail; res.info_count = c_info;

  // ---- B: ranges, and the tables cleared ----------------------------------------------------------------------------------------------
  u32 max_key = 0, max_val = 0;
  for (u32 t = lane; t < n; t += 64) {
    const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
    for (u32 i = 0; i < wn; i++) { const u32 x = w[i]; max_key = max(max_key, M_KEY(x)); if (M_VAL(x) != 0xFFu) max_val = max(max_val, M_VAL(x)); }
  }
  max_key = r_max(max_key); max_val = r_max(max_val);
  const u32 stride = max_val + 1u;
  if (max_val >= 64u || max_key >= p.kmax || (u64)(max_key + 1u) * stride > p.wmax) TO_HOST();   // (check_rw answers :unknown for values >= 64)
  for (u32 k = lane; k <= max_key; k += 64) { vseen[2 * k] = 0; vseen[2 * k + 1] = 0; }
  for (u32 k = lane; k < (max_key + 1u) * stride; k += 64) { writer[k] = NONE; vsucc[2 * k] = 0; vsucc[2 * k + 1] = 0; }
  for (u32 t = lane; t <= n; t += 64) { off[t] = 0; if (t < n) { indeg[t] = 0; cur[t] = 0; } }
  __syncthreads();
#define TYPE(t_) (t_lt[t_] >> 16)
#define SETBIT(arr_, idx_, v_) atomicOr(&(arr_)[2 * (idx_) + ((v_) >> 5)], 1u << ((v_) & 31u))

  // ---- C: writers (every transaction, whatever became of it); versions written by transactions that did not fail -----------------------
  u32 anomalies = 0;
  for (u32 t = lane; t < n; t += 64) {
    const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu; const bool failed = TYPE(t) == MSIM_T_FAIL;
    for (u32 i = 0; i < wn; i++) {
      const u32 x = w[i];
      if (!M_F(x)) continue;
      if (M_VAL(x) == 0xFFu) { anomalies |= MSIM_ANOMALY_INTERNAL; continue; }    // (a write of nil: the generator has none; let the host say what it is)
      if (atomicCAS(&writer[M_KEY(x) * stride + M_VAL(x)], NONE, t) != NONE) anomalies |= MSIM_ANOMALY_DUPLICATE_ELEMENTS;   // the generator never repeats (k, v)
      if (!failed) SETBIT(vseen, M_KEY(x), M_VAL(x));
    }
  }
  __syncthreads();
  if (__ballot(anomalies != 0)) TO_HOST();   // (duplicate writes are proscribed by every model)

  // the last value transaction t_ writes to key k_ (NONE: it does not write it)
  auto final_of = [&](u32 t_, u32 k_) -> u32 {
    const u32 *w = pay + t_off[t_]; const u32 wn = t_lt[t_] & 0xFFFFu; u32 v = NONE;
    for (u32 i = 0; i < wn; i++) { const u32 x = w[i]; if (M_F(x) && M_KEY(x) == k_) v = M_VAL(x); }
    return v;
  };

  // ---- D/E: edges: pass 0 counts degrees (and finds the non-cycle anomalies), pass 1 fills the CSR ----------------------------------------
  u32 n_edges = 0;
  if (want_rt) {   // realtime order in closed form (txn_check_kernel): sm[j] = earliest :ok completion among transactions j .. n-1
    u64 carry = ~0ull;
    for (int b = (int)((n + 63u) / 64u) - 1; b >= 0; b--) {
      const u32 t = (u32)b * 64u + lane;
      u64 v = (t < n && TYPE(t) == MSIM_T_OK) ? (((u64)t_cmp[t] << 32) | t_first[t]) : ~0ull;
      for (int o = 1; o < 64; o <<= 1) {
        const u32 ylo = (u32)__shfl_down((int)(u32)v, o), yhi = (u32)__shfl_down((int)(u32)(v >> 32), o);
        const u64 y = ((u64)yhi << 32) | ylo;
        if (lane + (u32)o < 64u) v = min(v, y);
      }
      v = min(v, carry);
      if (t < n) { sm[t] = (u32)(v >> 32); smf[t] = (u32)v; }
      carry = ((u64)r_rl((u32)(v >> 32), 0) << 32) | r_rl((u32)v, 0);
    }
    if (lane == 0) { sm[n] = NONE; smf[n] = n; }
    __syncthreads();
  }
  for (int pass = 0; pass < 2; pass++) {
    u32 my_edges = 0;
#define ADD(a_, b_) do { const u32 ea = (a_), eb = (b_); if (ea != eb) { if (pass == 0) { atomicAdd(&off[ea], 1u); atomicAdd(&indeg[eb], 1u); my_edges++; } \
                                                                         else adj[off[ea] + atomicAdd(&cur[ea], 1u)] = eb; } } while (0)
    // per :ok transaction: the reads
    for (u32 t = lane; t < n; t += 64) {
      if (TYPE(t) != MSIM_T_OK) continue;
      const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
      for (u32 k = 0; k < wn; k++) {
        const u32 x = w[k];
        if (M_F(x)) continue;
        const u32 key = M_KEY(x), val = M_VAL(x); const bool nil = val == 0xFFu;
        // internal consistency: the latest earlier micro-op on this key decides what the read must return
        int prev = -1;
        for (u32 e = 0; e < k; e++) if (M_KEY(w[e]) == key) prev = (int)e;
        if (prev >= 0) {
          const u32 px = w[prev]; const bool pnil = M_VAL(px) == 0xFFu;
          const bool same = M_F(px) ? (!nil && val == M_VAL(px)) : (nil == pnil && (nil || val == M_VAL(px)));
          if (!same) anomalies |= MSIM_ANOMALY_INTERNAL;
          continue;
        }
        // an external read
        if (!nil) {
          if (pass == 0) SETBIT(vseen, key, val);
          const u32 wr = writer[key * stride + val];
          if (wr == NONE || TYPE(wr) == MSIM_T_FAIL) { anomalies |= MSIM_ANOMALY_G1A; }   // garbage / aborted read
          else {
            if (wr != t) {
              if (final_of(wr, key) != val) anomalies |= MSIM_ANOMALY_G1B;
              if (want_wr) ADD(wr, t);
            }
            const u32 fw = final_of(t, key);   // writes follow reads
            if (pass == 0 && fw != NONE && fw != val) SETBIT(vsucc, key * stride + val, fw);
          }
        }
      }
      if (want_rt) {
        const u32 first = t_first[t];                          // the transactions invoked after t completed start here ...
        const u32 last = sm[first] == NONE ? n : smf[first];   // ... and end where the first of them to complete :ok did
        for (u32 v = first; v < last; v++) if (TYPE(v) != MSIM_T_FAIL) ADD(t, v);
      }
    }
    __syncthreads();
    if (pass == 0) {
      // nil precedes every version; cyclic version orders (lane = key): peel the versions that have no predecessor among those left
      for (u32 key = lane; key <= max_key; key += 64) {
        const u64 seen = (((u64)vseen[2 * key + 1] << 32) | vseen[2 * key]) & ~1ull;
        u64 alive = seen;   // (nil precedes everything and follows nothing: it never sits on a cycle)
        while (alive) {
          u64 has_in = 0;
          for (u64 b = alive; b; b &= b - 1) { const u32 v = (u32)__builtin_ctzll(b); if (v < stride) has_in |= ((u64)vsucc[2 * (key * stride + v) + 1] << 32) | vsucc[2 * (key * stride + v)]; }
          const u64 roots = alive & ~has_in;
          if (!roots) { anomalies |= MSIM_ANOMALY_CYCLIC_VERSIONS; break; }
          alive &= ~roots;
        }
      }
    }
    if (__ballot((anomalies & p.proscribed) != 0)) TO_HOST();
    if (__ballot((anomalies & MSIM_ANOMALY_CYCLIC_VERSIONS) != 0)) TO_HOST();
    // ww along the version order of every key (lane = key); nil has no writer
    for (u32 key = lane; key <= max_key; key += 64) {
      for (u32 v1 = 1; v1 < stride; v1++) {
        u64 su = ((u64)vsucc[2 * (key * stride + v1) + 1] << 32) | vsucc[2 * (key * stride + v1)];
        if (!su) continue;
        const u32 a = writer[key * stride + v1];
        if (a == NONE || TYPE(a) == MSIM_T_FAIL) continue;
        for (; su; su &= su - 1) {
          const u32 v2 = (u32)__builtin_ctzll(su);
          const u32 b = v2 < stride ? writer[key * stride + v2] : NONE;
          if (b != NONE && TYPE(b) != MSIM_T_FAIL) ADD(a, b);
        }
      }
    }
    // rw: a transaction's first micro-op on a key, if it is a read of v1, precedes the writers of v1's successors
    if (strong) {
      for (u32 t = lane; t < n; t += 64) {
        if (TYPE(t) != MSIM_T_OK) continue;
        const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
        for (u32 k = 0; k < wn; k++) {
          const u32 x = w[k];
          if (M_F(x)) continue;
          const u32 key = M_KEY(x);
          bool first = true;
          for (u32 e = 0; e < k; e++) if (M_KEY(w[e]) == key) first = false;
          if (!first) continue;
          const u32 v1 = M_VAL(x) == 0xFFu ? 0u : M_VAL(x);
          u64 su = v1 == 0 ? ((((u64)vseen[2 * key + 1] << 32) | vseen[2 * key]) & ~1ull)
                           : (((u64)vsucc[2 * (key * stride + v1) + 1] << 32) | vsucc[2 * (key * stride + v1)]);
          for (; su; su &= su - 1) {
            const u32 v2 = (u32)__builtin_ctzll(su);
            const u32 b = v2 < stride ? writer[key * stride + v2] : NONE;
            if (b != NONE && TYPE(b) != MSIM_T_FAIL) ADD(t, b);
          }
        }
      }
    }
#undef ADD
    __syncthreads();
    if (pass == 0) {
      n_edges = r_sum(my_edges);
      if (n_edges > p.emax) TO_HOST();
      u32 carry = 0;   // out-degrees -> CSR offsets (exclusive prefix sums, 64 at a time)
      for (u32 base = 0; base <= n; base += 64) {
        const u32 t = base + lane;
        const u32 d = t < n ? off[t] : 0u;
        const u32 ex = r_excl_scan(d, lane);
        if (t <= n) off[t] = carry + ex;
        carry += r_sum(d);
      }
      __syncthreads();
    }
  }
  anomalies = r_or(anomalies);

  // ---- F: acyclic?  Kahn's algorithm, 64 ready transactions per step ---------------------------------------------------------------------
  u32 tail = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 t = base + lane;
    const bool z = t < n && indeg[t] == 0;
    const u64 zm = __ballot(z);
    if (z) queue[tail + (u32)__popcll(zm & lt)] = t;
    tail += (u32)__popcll(zm);
  }
  __syncthreads();
  u32 head = 0;
  while (head < tail) {
    const u32 snap = tail;
    const u32 cnt = min(64u, snap - head);
    const bool on = lane < cnt;
    const u32 v = on ? queue[head + lane] : 0u;
    const u32 a0 = on ? off[v] : 0u, a1 = on ? off[v + 1] : 0u;
    for (u32 k = 0; __ballot(a0 + k < a1); k++) {
      bool push = false; u32 wv = 0;
      if (a0 + k < a1) { wv = adj[a0 + k]; push = atomicSub(&indeg[wv], 1u) == 1u; }
      const u64 pm = __ballot(push);
      if (push) queue[tail + (u32)__popcll(pm & lt)] = wv;
      tail += (u32)__popcll(pm);
    }
    head += cnt;
    __syncthreads();
  }
  if (tail != n) TO_HOST();   // a cycle over edge kinds the model proscribes cycles of (or, for the stronger models, any cycle): the host classifies it

  if (lane == 0) {
    res.lost_count = n_edges;      // edges of the subgraph built
    res.error_count = anomalies;   // non-cycle anomalies seen (none of them proscribed)
    res.valid = flags ? 0u : (c_ok == 0 ? 2u : 1u);
    p.out[hist] = res;
  }
#undef TO_HOST
#undef TYPE
#undef SETBIT
}

uint64_t rw_ws_words(u32 nmax, u32 emax, u32 kmax, u32 wmax) { return (uint64_t)nmax * 11 + 4 + 2 * (uint64_t)kmax + 3 * (uint64_t)wmax + emax; }

int rw_dev_run(msim_ctx *ctx, RParams rp, u32 n, const std::vector<msim_inst_meta> *hmeta, msim_check_result *h_out, hipStream_t st, u32 *n_host,
               void **ws_buf, size_t *ws_cap) {
  const bool trace = (msim_dev_flags(ctx) & 0x1000u) != 0;   // developer: time the passes
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  if (rp.kmax == 0 || rp.kmax > KMAX) rp.kmax = KMAX;
  if (rp.wmax == 0 || rp.wmax > WMAX) rp.wmax = WMAX;
  rp.ws_words = rw_ws_words(rp.nmax, rp.emax, rp.kmax, rp.wmax);
  const uint64_t budget = 6ull << 30;   // as many histories per launch as a few GB of workspace hold
  const u32 chunk = (u32)std::min<uint64_t>(n, std::max<uint64_t>(1, budget / (rp.ws_words * 4)));
  const size_t need = (size_t)chunk * rp.ws_words * 4;
  if (*ws_cap < need) {
    if (*ws_buf) (void)msim_dev_free(*ws_buf);
    *ws_buf = nullptr; *ws_cap = 0;
    MSIM_HIP_TRY(ctx, msim_dev_malloc(ws_buf, need));
    *ws_cap = need;
  }
  rp.ws = static_cast<u32 *>(*ws_buf);
  for (u32 first = 0; first < n; first += chunk) {
    rp.first = first;
    hipLaunchKernelGGL(rw_check_kernel, dim3(std::min(chunk, n - first)), dim3(64), 0, st, rp);
    MSIM_HIP_TRY(ctx, hipGetLastError());
  }
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_out, rp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<u32> todo;
  for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HOST) todo.push_back(i);
  if (trace) std::fprintf(stderr, "[rw-check] device pass: %.2f ms, %zu of %u histories for the host\n", ms(), todo.size(), n);
  if (!todo.empty()) {
    std::vector<uint64_t> ro, po;
    if (rp.row_off) { ro.resize(n + 1); po.resize(n + 1);
      MSIM_HIP_TRY(ctx, hipMemcpy(ro.data(), rp.row_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
      MSIM_HIP_TRY(ctx, hipMemcpy(po.data(), rp.pay_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost)); }
    std::vector<std::vector<msim_op>> rows(todo.size());
    std::vector<std::vector<u32>> pays(todo.size());
    for (size_t k = 0; k < todo.size(); k++) {
      const u32 i = todo[k];
      const u32 nr = hmeta ? (*hmeta)[i].n_rows : (u32)(ro[i + 1] - ro[i]), nw = hmeta ? (*hmeta)[i].n_payload_words : (u32)(po[i + 1] - po[i]);
      rows[k].resize(nr ? nr : 1); pays[k].resize(nw ? nw : 1);
      if (nr) MSIM_HIP_TRY(ctx, hipMemcpy(rows[k].data(), rp.rows + (hmeta ? (uint64_t)i * rp.max_rows : ro[i]), (size_t)nr * sizeof(msim_op), hipMemcpyDeviceToHost));
      if (nw) MSIM_HIP_TRY(ctx, hipMemcpy(pays[k].data(), rp.payload + (hmeta ? (uint64_t)i * rp.max_pay : po[i]), (size_t)nw * 4, hipMemcpyDeviceToHost));
    }
    unsigned nt = msim_host_threads();
    if (nt > todo.size()) nt = (unsigned)todo.size();
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; w++)
      th.emplace_back([&, w]() {
        for (size_t k = w; k < todo.size(); k += nt) {
          const u32 i = todo[k];
          msim_rw_check_instance_host(rows[k].data(), hmeta ? (*hmeta)[i].n_rows : (u32)(ro[i + 1] - ro[i]), pays[k].data(),
                                      hmeta ? (*hmeta)[i].n_payload_words : (u32)(po[i + 1] - po[i]), hmeta ? (*hmeta)[i].flags : 0u, rp.cm, &h_out[i]);
        }
      });
    for (auto &x : th) x.join();
    for (u32 i : todo) MSIM_HIP_TRY(ctx, hipMemcpy(rp.out + i, &h_out[i], sizeof(msim_check_result), hipMemcpyHostToDevice));
    if (trace) std::fprintf(stderr, "[rw-check] host analysis of those: done at %.2f ms\n", ms());
  }
  if (n_host) *n_host = (u32)todo.size();
  return MSIM_OK;
}

}  // namespace

// msim_check for txn-rw-register: the histories of the last run, where they lie in HBM.
int msim_check_rw_device(msim_ctx *ctx) {
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const u32 n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<msim_inst_meta> hm(n);
  MSIM_HIP_TRY(ctx, hipMemcpy(hm.data(), ctx->d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
  RParams rp;
  std::memset(&rp, 0, sizeof rp);
  rp.rows = ctx->d_rows; rp.payload = ctx->d_payload; rp.meta = ctx->d_meta; rp.out = ctx->d_check;
  rp.max_rows = ctx->cfg.max_rows; rp.max_pay = ctx->cfg.max_payload_words;
  rp.nmax = ctx->cfg.max_rows / 2 + 1; rp.emax = rp.nmax * 16;
  rp.cm = ctx->cfg.consistency_model; rp.proscribed = msim_proscribed_anomalies(rp.cm);
  // a run names keys below max_values and values up to max-writes-per-key (<= 63): tables of that size instead of the 4096 x 16 a caller's
  // histories may need — 1 MB of workspace per history was four launches of 4096 for the demo shape, each waiting for its slowest history
  rp.kmax = ctx->cfg.max_values ? ctx->cfg.max_values : 1u;
  rp.wmax = (u32)std::min<uint64_t>(WMAX, (uint64_t)rp.kmax * (ctx->cfg.max_writes_per_key + 2u));
  u32 redone = 0;
  int rc = rw_dev_run(ctx, rp, n, &hm, ctx->h_check, ctx->stream, &redone, &ctx->d_check_scratch, &ctx->cap_check_scratch);
  if (rc != MSIM_OK) return rc;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->lin_host_rechecks = redone;
  ctx->checked = true; ctx->check_fetched = true;
  return MSIM_OK;
}

// Checks `n_histories` rw-register histories given on the host (rows / payload words of history i at row_offsets[i] /
// payload_offsets[i]) as msim_check does for the histories of a run.
extern "C" int msim_check_rw_batch(int device, const msim_op *rows, const uint64_t *row_offsets, const uint32_t *payload, const uint64_t *payload_offsets,
                                   uint32_t n_histories, uint32_t consistency_model, msim_check_result *out, uint32_t *n_host) {
  if (!rows || !row_offsets || !payload_offsets || !out || n_histories == 0 || consistency_model > MSIM_CM_READ_UNCOMMITTED) return MSIM_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_HIP;
  msim_ctx tmp_ctx; msim_ctx *ctx = &tmp_ctx;   // only for error text
  tmp_ctx.device = device;
  const uint64_t tr = row_offsets[n_histories], tw = payload_offsets[n_histories];
  u32 max_r = 1;
  for (u32 i = 0; i < n_histories; i++) { const uint64_t c = row_offsets[i + 1] - row_offsets[i]; if (c > 0x7FFFFFFFull) return MSIM_E_RANGE; if (c > max_r) max_r = (u32)c; }
  msim_op *d_rows = nullptr; u32 *d_pay = nullptr; uint64_t *d_ro = nullptr, *d_po = nullptr; msim_check_result *d_out = nullptr; void *ws = nullptr; size_t ws_cap = 0;
  int rc = MSIM_E_HIP;
  do {
    if (msim_dev_malloc(&d_rows, (size_t)(tr ? tr : 1) * sizeof(msim_op)) != hipSuccess) break;
    if (msim_dev_malloc(&d_pay, (size_t)(tw ? tw : 1) * 4) != hipSuccess) break;
    if (msim_dev_malloc(&d_ro, (size_t)(n_histories + 1) * 8) != hipSuccess || msim_dev_malloc(&d_po, (size_t)(n_histories + 1) * 8) != hipSuccess) break;
    if (msim_dev_malloc(&d_out, (size_t)n_histories * sizeof(msim_check_result)) != hipSuccess) break;
    if (tr && hipMemcpy(d_rows, rows, (size_t)tr * sizeof(msim_op), hipMemcpyHostToDevice) != hipSuccess) break;
    if (tw && hipMemcpy(d_pay, payload, (size_t)tw * 4, hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_ro, row_offsets, (size_t)(n_histories + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_po, payload_offsets, (size_t)(n_histories + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
    RParams rp;
    std::memset(&rp, 0, sizeof rp);
    rp.rows = d_rows; rp.payload = d_pay; rp.row_off = d_ro; rp.pay_off = d_po; rp.out = d_out;
    rp.nmax = max_r / 2 + 65; rp.emax = rp.nmax * 16; rp.cm = consistency_model; rp.proscribed = msim_proscribed_anomalies(consistency_model);
    rc = rw_dev_run(ctx, rp, n_histories, nullptr, out, nullptr, n_host, &ws, &ws_cap);
  } while (false);
  for (void *q : {(void *)d_rows, (void *)d_pay, (void *)d_ro, (void *)d_po, (void *)d_out, ws}) if (q) (void)msim_dev_free(q);
  return rc;
}
