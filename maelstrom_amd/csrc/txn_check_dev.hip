// txn_check_dev.hip — the list-append analysis of txn-list-append histories on the device (msim_check for MSIM_WL_TXN_LIST_APPEND;
// SURVEY.md §8f).  What the reference wires in at workload/txn_list_append.clj:142 is [upstream] elle's list-append checker;
// txn_check.cpp restates it on the host (version orders from the longest reads, ww / wr / rw + transitively reduced realtime
// edges, cycle search and classification, the non-cycle anomalies).
//
// This file is the COMMON CASE of that analysis, one wavefront per history: it PROVES a history clean — no non-cycle anomaly, and
// the dependency graph (every edge kind, realtime included) acyclic — and then the result is what the host computes for a clean
// history: the counts, the number of edges, :valid? true.  Anything else (an anomaly of any kind, a cycle, a shape beyond the
// capacities below) is handed to txn_check.cpp, which analyses, classifies and judges it by the consistency model; the device
// never decides an unclean history.  So the verdicts are the host's by construction, and the work the host cores used to do for
// every history (0.6 ms of pointer chasing each, after 5 GB over PCIe for cfg5) is left for the rare bad ones.
//
// Data-parallel restatement of the host's steps (lane = transaction unless said otherwise):
//   A  rows -> transactions: invocations numbered by prefix sums; completions paired by process through a table of open calls
//      in lane registers (the one serial walk);
//   B  key / element ranges; C  writer table (key, element) -> transaction by compare-and-swap (a second writer = duplicate);
//   D  per :ok read: duplicates (256-bit set in registers), internal consistency against the transaction's own earlier micro-ops,
//      G1a / G1b through the writer table, the key's longest read by atomic max;
//   E  edges, generated twice (count, then fill a CSR): ww along each key's longest read (lane = key), wr / rw per read, and the
//      realtime order in closed form — an :ok transaction u stays on the host's "frontier" from its completion until the first
//      completion of a transaction invoked after it, so its successors are the transactions invoked in between: a contiguous
//      index range read off two prefix counts taken while pairing and a suffix minimum (the host walks the history and edits a
//      frontier list);
//   F  acyclicity by Kahn's algorithm: 64 ready transactions per step, in-degrees by atomics, the ready queue by ballots.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "engine_internal.h"

void msim_txn_check_instance_host(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, uint32_t cm,
                                  msim_check_result *res);   // txn_check.cpp

namespace {

constexpr u32 NEEDS_HOST = 3u;
constexpr u32 NEEDS_HBM = 4u;    // txn_check_lds_kernel: the history's tables do not fit the LDS of a wavefront; txn_check_kernel (tables in HBM) takes it
constexpr u32 NONE = 0xFFFFFFFFu;
constexpr u32 KMAX = 4096u;      // keys per history
constexpr u32 WMAX = 65536u;     // writer table entries (keys x elements)

struct TParams {
  const msim_op *rows; const u32 *payload; const msim_inst_meta *meta;
  const uint64_t *row_off, *pay_off;   // (null: history i at i * max_rows / i * max_pay)
  msim_check_result *out;
  u32 *ws;                       // workspace, ws_words per history of this launch
  uint64_t ws_words;
  u32 max_rows, max_pay, nmax, emax, first;   // first: index of the launch's first history
  const u32 *list;               // (not null: the launch's histories are list[first + blockIdx.x])
  u32 lds_bytes;                 // txn_check_lds_kernel: dynamic LDS of a wavefront
};

__device__ __forceinline__ u32 t_rl(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u32 t_sum(u32 v) { for (int o = 32; o; o >>= 1) v += (u32)__shfl_xor((int)v, o); return v; }
__device__ __forceinline__ u32 t_max(u32 v) { for (int o = 32; o; o >>= 1) v = max(v, (u32)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ u32 t_excl_scan(u32 v, u32 lane) {   // exclusive prefix sum over the wavefront
  u32 x = v;
  for (int o = 1; o < 64; o <<= 1) { const u32 y = (u32)__shfl_up((int)x, o); if (lane >= (u32)o) x += y; }
  return x - v;
}
__device__ __forceinline__ u32 elem(const u32 *lw, u32 e) { return (lw[e >> 2] >> (8u * (e & 3u))) & 0xFFu; }

// one micro-op of a transaction's payload words w[0 .. n): header, then the list of a read
struct Mop { u32 f, key, val, len; const u32 *list; bool bad; };   // read of nil: len 0
__device__ __forceinline__ Mop next_mop(const u32 *w, u32 n, u32 &i) {
  Mop m; const u32 h = w[i++];
  m.f = h & 1u; m.key = (h >> 1) & 0x7FFFu; m.val = 0; m.len = 0; m.list = w + i; m.bad = false;
  const u32 x = (h >> 16) & 0xFFu;
  if (m.f) m.val = x;
  else if (x != 0xFFu) { m.len = x; const u32 words = (x + 3u) >> 2; if (i + words > n) m.bad = true; i += words; }
  return m;
}

__global__ void __launch_bounds__(64) txn_check_kernel(const TParams p) {
  const u32 lane = threadIdx.x, hist = p.list ? p.list[p.first + blockIdx.x] : p.first + blockIdx.x;
  const u64 lt = (1ull << lane) - 1ull;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.row_off ? p.row_off[hist] : (u64)hist * p.max_rows);
  const u32 *const pay = p.payload + (p.pay_off ? p.pay_off[hist] : (u64)hist * p.max_pay);
  const u32 n_rows = p.meta ? p.meta[hist].n_rows : (u32)(p.row_off[hist + 1] - p.row_off[hist]);
  const u32 n_words = p.meta ? p.meta[hist].n_payload_words : (u32)(p.pay_off[hist + 1] - p.pay_off[hist]);
  const u32 flags = p.meta ? p.meta[hist].flags : 0u;
  const u32 NM = p.nmax;
  u32 *const ws = p.ws + (u64)blockIdx.x * p.ws_words;
  u32 *const t_inv = ws, *const t_cmp = t_inv + NM, *const t_off = t_cmp + NM, *const t_lt = t_off + NM;   // t_lt: words | type << 16
  u32 *const t_first = t_lt + NM;            // transactions invoked before this one's completion row (= index of the first one after it)
  u32 *const sm = t_first + NM;              // [NM + 1] suffix minimum of the :ok completions ...
  u32 *const smf = sm + NM + 1;              // [NM + 1] ... and t_first of the transaction that attains it
  u32 *const indeg = smf + NM + 1, *const off = indeg + NM;   // off [NM + 1]: out-degrees, then CSR offsets
  u32 *const cur = off + NM + 1, *const queue = cur + NM;
  u32 *const longest = queue + NM;           // [KMAX] (len + 1) << 24 | first payload word of the list
  u32 *const writer = longest + KMAX;        // [WMAX]
  u32 *const adj = writer + WMAX;            // [emax]

  msim_check_result res;
  res.valid = NEEDS_HOST; res.attempt_count = 0; res.stable_count = 0; res.lost_count = 0; res.never_read_count = 0; res.stale_count = 0;
  res.duplicated_count = 0; res.error_count = 0;
  for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = 0;
  res.op_count = 0; res.ok_count = 0; res.fail_count = 0; res.info_count = 0;
#define TO_HOST() do { if (lane == 0) p.out[hist] = res; return; } while (0)
#ifdef TC_PROF   // developer build: cycles per phase in the result's unused fields
  u64 tc_prev = __builtin_readcyclecounter(); u32 tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TC_MARK(i_) { const u64 now_ = __builtin_readcyclecounter(); tc[i_] += (u32)((now_ - tc_prev) >> 6); tc_prev = now_; }
#else
#define TC_MARK(i_)
#endif
  if (n_words >= (1u << 24)) TO_HOST();

  // ---- A: transactions ---------------------------------------------------------------------------------------------------------
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
      if (inv) { t_inv[my_t] = idx; t_cmp[my_t] = NONE; t_off[my_t] = woff; t_lt[my_t] = len | (MSIM_T_INFO << 16); }   // never completed = indeterminate
      u64 m = __ballot(is);
      while (m) {
        const u32 j = (u32)__builtin_ctzll(m); m &= m - 1;
        const u32 z = t_rl(row.z, j);
        const u32 jt = z & 3u, jp = z >> 12;
        const u64 hit = __ballot(o_used && o_proc == jp);
        if (jt == MSIM_T_INVOKE) {
          u32 s;
          if (hit) s = (u32)__builtin_ctzll(hit);
          else { const u64 used = __ballot(o_used); if (used == ~0ull) { bad = true; break; } s = (u32)__builtin_ctzll(~used); }
          const u32 tj = n + (u32)__popcll(im & ((1ull << j) - 1ull));
          const u32 jl = t_rl(row.y, j) >> 16;
          if (lane == s) { o_used = true; o_proc = jp; o_txn = tj; o_len = jl; }
        } else if (hit) {
          const u32 s = (u32)__builtin_ctzll(hit);
          const u32 id = t_rl(o_txn, s), ilen = t_rl(o_len, s);
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

  TC_MARK(0)
  // ---- B: ranges, and the tables cleared ------------------------------------------------------------------------------------------
  u32 max_key = 0, max_val = 0; bool bad = false;
  for (u32 t = lane; t < n; t += 64) {
    const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
    for (u32 i = 0; i < wn;) { const Mop m = next_mop(w, wn, i); bad |= m.bad; max_key = max(max_key, m.key); if (m.f) max_val = max(max_val, m.val); }
  }
  max_key = t_max(max_key); max_val = t_max(max_val);
  const u32 stride = max_val + 1u;
  if (__ballot(bad) || max_key >= KMAX || (u64)(max_key + 1u) * stride > WMAX) TO_HOST();
  for (u32 k = lane; k <= max_key; k += 64) longest[k] = 0;
  for (u32 k = lane; k < (max_key + 1u) * stride; k += 64) writer[k] = NONE;
  for (u32 t = lane; t <= n; t += 64) { off[t] = 0; if (t < n) { indeg[t] = 0; cur[t] = 0; } }
  __syncthreads();

  TC_MARK(1)
  // ---- C: writers (every transaction, whatever became of it) -------------------------------------------------------------------------
  for (u32 t = lane; t < n; t += 64) {
    const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
    for (u32 i = 0; i < wn;) { const Mop m = next_mop(w, wn, i); if (m.f && atomicCAS(&writer[m.key * stride + m.val], NONE, t) != NONE) bad = true; }   // the generator never repeats (k, v)
  }
  __syncthreads();
  if (__ballot(bad)) TO_HOST();
#define WRITER(k_, el_) ((el_) < stride ? writer[(k_) * stride + (el_)] : NONE)

  TC_MARK(2)
  // ---- D: the reads of :ok transactions ------------------------------------------------------------------------------------------------
  for (u32 t = lane; t < n; t += 64) {
    if ((t_lt[t] >> 16) != MSIM_T_OK) continue;
    const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
    u32 k = 0;
    for (u32 i = 0; i < wn; k++) {
      const Mop m = next_mop(w, wn, i);
      if (m.f) continue;
      // duplicates
      { u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (u32 e = 0; e < m.len; e++) {
          const u32 x = elem(m.list, e); const u64 b = 1ull << (x & 63u); const u32 q = x >> 6;
          const u64 s = q == 0 ? s0 : q == 1 ? s1 : q == 2 ? s2 : s3;
          if (s & b) bad = true;
          s0 |= q == 0 ? b : 0; s1 |= q == 1 ? b : 0; s2 |= q == 2 ? b : 0; s3 |= q == 3 ? b : 0;
        } }
      // internal consistency: what the transaction's own earlier micro-ops imply for this read
      { int prev = -1; Mop pm = m; u32 e_i = 0, e_k = 0;
        for (e_i = 0, e_k = 0; e_k < k; e_k++) { const Mop q = next_mop(w, wn, e_i); if (!q.f && q.key == m.key) { prev = (int)e_k; pm = q; } }
        const u32 e0 = prev < 0 ? 0u : (u32)prev + 1u;
        u32 n_app = 0;
        for (e_i = 0, e_k = 0; e_k < k; e_k++) { const Mop q = next_mop(w, wn, e_i); if (e_k >= e0 && q.f && q.key == m.key) n_app++; }
        bool ok = true; u32 at = 0;
        if (prev >= 0) { ok = m.len == pm.len + n_app; if (ok) for (u32 e = 0; e < pm.len; e++) ok &= elem(m.list, e) == elem(pm.list, e); at = pm.len; }
        else { ok = m.len >= n_app; at = m.len - n_app; }
        if (ok) { u32 a = 0; for (e_i = 0, e_k = 0; e_k < k; e_k++) { const Mop q = next_mop(w, wn, e_i); if (e_k >= e0 && q.f && q.key == m.key) { if (elem(m.list, at + a) != q.val) ok = false; a++; } } }
        if (!ok) bad = true; }
      // the externally visible part: without the transaction's own appends at the tail
      u32 ext = m.len;
      while (ext > 0 && WRITER(m.key, elem(m.list, ext - 1)) == t) ext--;
      for (u32 e = 0; e < ext; e++) { const u32 wr = WRITER(m.key, elem(m.list, e)); if (wr == NONE || (t_lt[wr] >> 16) == MSIM_T_FAIL) bad = true; }   // G1a
      if (ext > 0) {   // G1b: the last visible element must be its writer's last append to the key
        const u32 last = elem(m.list, ext - 1), wr = WRITER(m.key, last);
        if (wr != NONE && wr != t) {
          const u32 *w2 = pay + t_off[wr]; const u32 wn2 = t_lt[wr] & 0xFFFFu; u32 fin = NONE;
          for (u32 i2 = 0; i2 < wn2;) { const Mop q = next_mop(w2, wn2, i2); if (q.f && q.key == m.key) fin = q.val; }
          if (fin != last) bad = true;
        }
      }
      atomicMax(&longest[m.key], ((m.len + 1u) << 24) | (u32)(m.list - pay));
    }
  }
  __syncthreads();
  if (__ballot(bad)) TO_HOST();

  TC_MARK(3)
  // realtime order in closed form: sm[j] = earliest :ok completion among transactions j .. n-1 (smf[j]: how many transactions were
  // invoked before that completion)
  {
    u64 carry = ~0ull;
    for (int b = (int)((n + 63u) / 64u) - 1; b >= 0; b--) {
      const u32 t = (u32)b * 64u + lane;
      u64 v = (t < n && (t_lt[t] >> 16) == MSIM_T_OK) ? (((u64)t_cmp[t] << 32) | t_first[t]) : ~0ull;
      for (int o = 1; o < 64; o <<= 1) {
        const u32 ylo = (u32)__shfl_down((int)(u32)v, o), yhi = (u32)__shfl_down((int)(u32)(v >> 32), o);
        const u64 y = ((u64)yhi << 32) | ylo;
        if (lane + (u32)o < 64u) v = min(v, y);
      }
      v = min(v, carry);
      if (t < n) { sm[t] = (u32)(v >> 32); smf[t] = (u32)v; }
      carry = ((u64)t_rl((u32)(v >> 32), 0) << 32) | t_rl((u32)v, 0);
    }
    if (lane == 0) { sm[n] = NONE; smf[n] = n; }
  }
  __syncthreads();

  TC_MARK(4)
  // ---- E: edges: pass 0 counts degrees, pass 1 fills the CSR ------------------------------------------------------------------------------
  u32 n_edges = 0;
  for (int pass = 0; pass < 2; pass++) {
    u32 my_edges = 0;
#define ADD(a_, b_) do { const u32 ea = (a_), eb = (b_); if (ea != eb) { if (pass == 0) { atomicAdd(&off[ea], 1u); atomicAdd(&indeg[eb], 1u); my_edges++; } \
                                                                         else adj[off[ea] + atomicAdd(&cur[ea], 1u)] = eb; } } while (0)
    // ww along each key's version order (lane = key)
    for (u32 key = lane; key <= max_key; key += 64) {
      const u32 L = longest[key];
      if (L == 0) continue;
      const u32 len = (L >> 24) - 1u; const u32 *ord = pay + (L & 0xFFFFFFu);
      for (u32 i = 0; i + 1 < len; i++) {
        const u32 a = WRITER(key, elem(ord, i)), b = WRITER(key, elem(ord, i + 1));
        if (a == NONE || b == NONE) continue;
        const bool fa = (t_lt[a] >> 16) == MSIM_T_FAIL, fb = (t_lt[b] >> 16) == MSIM_T_FAIL;
        if (fa && !fb) bad = true;   // dirty update
        if (!fa && !fb) ADD(a, b);
      }
    }
    // wr / rw per read of an :ok transaction; realtime successors of an :ok transaction
    for (u32 t = lane; t < n; t += 64) {
      if ((t_lt[t] >> 16) != MSIM_T_OK) continue;
      const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
      for (u32 i = 0; i < wn;) {
        const Mop m = next_mop(w, wn, i);
        if (m.f) continue;
        const u32 L = longest[m.key];
        const u32 llen = (L >> 24) - 1u; const u32 *ord = pay + (L & 0xFFFFFFu);
        bool pre = m.len <= llen;
        if (pre) for (u32 e = 0; e < m.len; e++) pre &= elem(m.list, e) == elem(ord, e);
        if (!pre) { bad = true; continue; }   // incompatible order
        u32 ext = m.len;
        while (ext > 0 && WRITER(m.key, elem(m.list, ext - 1)) == t) ext--;
        if (ext > 0) { const u32 wr = WRITER(m.key, elem(m.list, ext - 1)); if (wr != NONE && (t_lt[wr] >> 16) != MSIM_T_FAIL) ADD(wr, t); }
        if (m.len < llen) { const u32 wr = WRITER(m.key, elem(ord, m.len)); if (wr != NONE && (t_lt[wr] >> 16) != MSIM_T_FAIL) ADD(t, wr); }   // anti-dependency
      }
      const u32 first = t_first[t];                      // the transactions invoked after t completed start here ...
      const u32 last = sm[first] == NONE ? n : smf[first];   // ... and end where the first of them to complete :ok did
      for (u32 v = first; v < last; v++) if ((t_lt[v] >> 16) != MSIM_T_FAIL) ADD(t, v);
    }
#undef ADD
    __syncthreads();
    if (__ballot(bad)) TO_HOST();
    if (pass == 0) {
      n_edges = t_sum(my_edges);
      if (n_edges > p.emax) TO_HOST();
      // out-degrees -> CSR offsets (exclusive prefix sums, 64 at a time)
      u32 carry = 0;
      for (u32 base = 0; base <= n; base += 64) {
        const u32 t = base + lane;
        const u32 d = t < n ? off[t] : 0u;
        const u32 ex = t_excl_scan(d, lane);
        if (t <= n) off[t] = carry + ex;
        carry += t_sum(d);
      }
      __syncthreads();
    }
  }

  TC_MARK(5)
  // ---- F: acyclic?  Kahn's algorithm, 64 ready transactions per step -------------------------------------------------------------------------
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
  TC_MARK(6)
  if (tail != n) TO_HOST();   // a cycle: the host finds and classifies it

  if (lane == 0) {
    res.lost_count = n_edges;   // edges of the dependency graph
    res.valid = flags ? 0u : (c_ok == 0 ? 2u : 1u);
#ifdef TC_PROF
    for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = tc[i];
    res.never_read_count = tc[5]; res.duplicated_count = tc[6];
#endif
    p.out[hist] = res;
  }
#undef TO_HOST
#undef WRITER
}


// ---- the same analysis with its tables in LDS (round 3) ---------------------------------------------------------------------------------
// txn_check_kernel above keeps the writer table, in-degrees, CSR and ready queue of a history in an HBM workspace (0.6 MB per history):
// every edge costs three L2 atomics whose lines are evicted before they are touched again — 20.6 GB of HBM traffic per 8192 histories for
// 1.35 GB of history bytes, 0.80 of the wave cycles waiting (profiles/r02_cfg5_counters.json).  Here a workgroup takes one history
// (see the kernel's header for how its wavefronts share the work), and
//   * what is hit at random lives in LDS as 16-bit entries: the writer table (key, element) -> transaction | type << 13 | "the writer's
//     last append to the key" << 15 (so G1a / G1b / the edge passes never read another transaction's words), the longest read per key,
//     in-degrees and CSR offsets (two per word, updated by one 32-bit LDS atomic on the right half), the adjacency of the DEPENDENCY edges,
//     and during Kahn's algorithm — in the space the writer table no longer needs — the ready queue and every transaction's realtime range;
//   * realtime edges are never materialised: an :ok transaction's successors are the index range [first, last) of the closed form above;
//     Kahn's step walks the CSR entries and then that range;
//   * completions are paired per PROCESS, not per row: the rows of one process alternate, so a completion's invocation is the previous
//     row of its process — in the same 64-row block (a ballot and a shuffle) or the process's open call (lane registers);
//   * the HBM workspace holds only streamed per-transaction words (6 x n).
// A history whose tables do not fit (more than 8191 transactions, 65535 edges, or the LDS budget) is answered NEEDS_HBM and runs on
// txn_check_kernel; anything not provably clean is answered NEEDS_HOST, as before.
__device__ __forceinline__ u32 h16_add(u32 *w, u32 i, u32 d) { const u32 sh = (i & 1u) * 16u; return (atomicAdd(&w[i >> 1], d << sh) >> sh) & 0xFFFFu; }   // returns the old half
__device__ __forceinline__ u32 h16_sub(u32 *w, u32 i) { const u32 sh = (i & 1u) * 16u; return (atomicSub(&w[i >> 1], 1u << sh) >> sh) & 0xFFFFu; }
__device__ __forceinline__ u32 h16_get(const u32 *w, u32 i) { return (w[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu; }

// A workgroup of 1 .. 8 wavefronts takes one history (blockDim.x = 64 x wavefronts, TParams.lds_bytes of LDS per workgroup): the passes
// whose unit of work is a transaction or a key (B, C, D, E, the realtime ranges) stride over all threads, so that many more of the
// dependent payload loads they consist of are in flight per history; the passes that are serial by nature (pairing the rows, the suffix
// minimum, the CSR prefix sums, Kahn's queue) stay on wavefront 0 while the others wait at the next barrier.  smem[0 .. 64) is the
// workgroup's header: [0] verdict so far (0 go on, 1 tables do not fit -> txn_check_kernel, 2 not provably clean -> host), [1] n,
// [2..4] ok / fail / info counts, [5] max key, [6] max element, [7] edges, [8] dependency edges.
__device__ __forceinline__ void t_wave_fence() {   // orders the LDS / workspace traffic of ONE wavefront across its lanes (no s_barrier)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Two workgroups of sixteen wavefronts share a CU only at EIGHT wavefronts per SIMD: <= 64 vector and <= 96 scalar registers (the compiler's own choice,
// 106 scalar registers, allowed seven — the second workgroup waited for the first to END: 252 histories in flight, not 512; tools/txn_check_prof_report.py).
#ifndef TC_WAVES_PER_EU
#define TC_WAVES_PER_EU 8
#endif
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(TC_WAVES_PER_EU, TC_WAVES_PER_EU))) txn_check_lds_kernel(const TParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, NT = blockDim.x, lane = tid & 63u, wave = tid >> 6;
  const u32 hist = p.list ? p.list[p.first + blockIdx.x] : p.first + blockIdx.x;
  const u64 lt = (1ull << lane) - 1ull;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.row_off ? p.row_off[hist] : (u64)hist * p.max_rows);
  const u32 *const pay = p.payload + (p.pay_off ? p.pay_off[hist] : (u64)hist * p.max_pay);
  const u32 n_rows = p.meta ? p.meta[hist].n_rows : (u32)(p.row_off[hist + 1] - p.row_off[hist]);
  const u32 n_words = p.meta ? p.meta[hist].n_payload_words : (u32)(p.pay_off[hist + 1] - p.pay_off[hist]);
  const u32 flags = p.meta ? p.meta[hist].flags : 0u;
  const u32 NM = p.nmax;
  u32 *const ws = p.ws + (u64)blockIdx.x * p.ws_words;
  u32 *const t_cmp = ws, *const t_off = t_cmp + NM, *const t_lt = t_off + NM;   // t_lt: words | type << 16
  u32 *const t_first = t_lt + NM;            // transactions invoked before this one's completion row (= index of the first one after it)
  u32 *const sm = t_first + NM;              // [NM + 1] suffix minimum of the :ok completions ...
  u32 *const smf = sm + NM + 1;              // [NM + 1] ... and t_first of the transaction that attains it
  u32 *const hdr = reinterpret_cast<u32 *>(smem);
  unsigned char *const tab = smem + 64;

  msim_check_result res;
  res.valid = NEEDS_HOST; res.attempt_count = 0; res.stable_count = 0; res.lost_count = 0; res.never_read_count = 0; res.stale_count = 0;
  res.duplicated_count = 0; res.error_count = 0;
  for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = 0;
  res.op_count = 0; res.ok_count = 0; res.fail_count = 0; res.info_count = 0;
  // every thread of the workgroup sees the same verdict behind a barrier and leaves with it
#define VERDICT(v_) atomicMax(&hdr[0], (u32)(v_))
#define LEAVE_IF_DECIDED() do { __syncthreads(); const u32 vd_ = hdr[0]; if (vd_) { if (tid == 0) { res.valid = vd_ == 1u ? NEEDS_HBM : NEEDS_HOST; p.out[hist] = res; } return; } } while (0)
#ifdef TC_PROF   // developer build: cycles per phase (as wavefront 0 sees them: the barriers close a phase for the whole workgroup)
  u64 tl_prev = __builtin_readcyclecounter(); u32 tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const u32 tl_real0 = (u32)__builtin_readsteadycounter();   // 100 MHz, the same for every CU: when this workgroup began (error_count) and ended (stable_count)
#define TL_MARK(i_) { const u64 now_ = __builtin_readcyclecounter(); tl[i_] += (u32)((now_ - tl_prev) >> 6); tl_prev = now_; }
#else
#define TL_MARK(i_)
#endif
  if (tid < 16) hdr[tid] = 0;
  __syncthreads();
  if (n_words >= (1u << 24)) { if (tid == 0) p.out[hist] = res; return; }

  // ---- A: transactions; completions paired process by process — every wavefront takes blocks of 64 rows ------------------------------------
  // The rows of one process alternate, so a completion's invocation is the previous row of its process.  Wavefront 0 used to walk the
  // history block by block with the processes' open calls in its lanes (half of the kernel once Kahn's queue had gone, fifteen wavefronts
  // waiting).  Now, with the tables' LDS still free:
  //   A1  per block (all wavefronts): the number of invocations, every row's previous row of the same process INSIDE the block (a lane
  //       index, one byte per row), and for every process present its LAST row in the block {process, invocation? | its rank among the
  //       block's invocations | the words of its request} — at most TC_PROCS entries per block;
  //   A2  blocks' invocation counts -> transaction numbers (prefix sums, wavefront 0);
  //   A3  invocations write their transaction's defaults;  A4  completions look their invocation up — in the block (the byte), or in the
  //       nearest earlier block that holds the process (its entry) — and write what became of the transaction.
  // A history with more rows than the LDS holds bytes for, or a block with more than TC_PROCS processes, goes to txn_check_kernel.
  constexpr u32 TC_PROCS = 16;
  const u32 NB = (n_rows + 63u) / 64u, NW = NT / 64u;
  u32 *const blk_base = reinterpret_cast<u32 *>(tab);            // [NB + 1] invocations in the block, then before it
  u32 *const blk_np = blk_base + NB + 1;                         // [NB] entries of the block's list
  u32 *const blk_list = blk_np + NB;                             // [NB][TC_PROCS] {process, info}: info = invocation << 31 | rank << 16 | words
  unsigned char *const prevl = reinterpret_cast<unsigned char *>(blk_list + (size_t)NB * TC_PROCS * 2u);   // [n_rows] 64: none, 255: not a transaction's row
  if (64u + ((size_t)NB * (2u + TC_PROCS * 2u) + 1u) * 4u + n_rows + 64u > p.lds_bytes) { if (tid == 0) { res.valid = NEEDS_HBM; p.out[hist] = res; } return; }
  {
    u32 bad = 0;   // 1 host, 2 txn_check_kernel
    for (u32 b = wave; b < NB; b += NW) {   // A1
      const u32 idx = b * 64u + lane;
      uint4 row = make_uint4(0, 0, 0, 0);
      if (idx < n_rows) row = r[idx];
      const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12, len = row.y >> 16, woff = row.w;
      const bool is = idx < n_rows && proc != MSIM_PROCESS_NEMESIS && f == MSIM_F_TXN;
      if (__ballot(is && (u64)woff + len > n_words)) bad = max(bad, 1u);
      const bool inv = is && type == MSIM_T_INVOKE;
      const u64 im = __ballot(inv);
      const u32 rank = (u32)__popcll(im & lt);
      u32 pl = is ? 64u : 255u, np = 0;
      u64 rem = __ballot(is);
      while (rem) {
        const u32 j = (u32)__builtin_ctzll(rem);
        const u32 pj = t_rl(proc, j);
        const u64 same = __ballot(is && proc == pj);
        rem &= ~same;
        const u64 below = same & lt;
        if (is && proc == pj && below) pl = 63u - (u32)__builtin_clzll(below);
        const u32 last = 63u - (u32)__builtin_clzll(same);
        const u32 info = (t_rl(inv ? 1u : 0u, last) << 31) | (t_rl(rank, last) << 16) | t_rl(len, last);
        if (np < TC_PROCS) { if (lane == 0) { blk_list[((size_t)b * TC_PROCS + np) * 2u] = pj; blk_list[((size_t)b * TC_PROCS + np) * 2u + 1u] = info; } }
        else bad = max(bad, 2u);
        np++;
      }
      if (idx < n_rows) prevl[idx] = (unsigned char)pl;
      if (lane == 0) { blk_base[b] = (u32)__popcll(im); blk_np[b] = min(np, TC_PROCS); }
    }
    if (bad && lane == 0) VERDICT(bad == 2u ? 1u : 2u);
  }
  LEAVE_IF_DECIDED();
  if (wave == 0) {   // A2
    u32 carry = 0;
    for (u32 base = 0; base <= NB; base += 64) {
      const u32 b = base + lane;
      const u32 d = b < NB ? blk_base[b] : 0u;
      const u32 ex = t_excl_scan(d, lane), tot = t_sum(d);
      if (b <= NB) blk_base[b] = carry + ex;
      carry += tot;
    }
    if (lane == 0) {
      hdr[1] = carry;
      if (carry > NM) VERDICT(2);
      else if (carry > 8190u) VERDICT(1);   // 13-bit ids in the writer table, 0x1FFF | INFO | fin would read as "nobody"
    }
  }
  LEAVE_IF_DECIDED();
  for (u32 b = wave; b < NB; b += NW) {   // A3
    const u32 idx = b * 64u + lane;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (idx < n_rows) row = r[idx];
    const bool inv = idx < n_rows && prevl[idx] != 255u && (row.z & 3u) == MSIM_T_INVOKE;
    const u32 my_t = blk_base[b] + (u32)__popcll(__ballot(inv) & lt);
    if (inv) { t_cmp[my_t] = NONE; t_off[my_t] = row.w; t_lt[my_t] = (row.y >> 16) | (MSIM_T_INFO << 16); t_first[my_t] = 0; }   // never completed = indeterminate
  }
  __threadfence_block();
  __syncthreads();
  {
    u32 c_ok = 0, c_fail = 0, c_info = 0;
    for (u32 b = wave; b < NB; b += NW) {   // A4
      const u32 idx = b * 64u + lane;
      uint4 row = make_uint4(0, 0, 0, 0);
      if (idx < n_rows) row = r[idx];
      const u32 type = row.z & 3u, proc = row.z >> 12, len = row.y >> 16, woff = row.w;
      const u32 pl = idx < n_rows ? (u32)prevl[idx] : 255u;
      const bool is = pl != 255u, inv = is && type == MSIM_T_INVOKE;
      const u64 im = __ballot(inv);
      const u32 rank = (u32)__popcll(im & lt), base = blk_base[b];
      const u32 p_rank = (u32)__shfl((int)rank, (int)(pl & 63u)), p_inv = (u32)__shfl((int)(inv ? 1u : 0u), (int)(pl & 63u)), p_len = (u32)__shfl((int)len, (int)(pl & 63u));
      u32 m_id = NONE, m_len = 0;   // a completion lane: the transaction it completes, the words of its request
      if (is && !inv) {
        if (pl < 64u) { if (p_inv) { m_id = base + p_rank; m_len = p_len; } }   // (after a completion nothing is open: a stray one)
        else {
          for (u32 bb = b; bb-- > 0 && m_id == NONE;) {
            const u32 ne = blk_np[bb];
            for (u32 e = 0; e < ne; e++)
              if (blk_list[((size_t)bb * TC_PROCS + e) * 2u] == proc) {
                const u32 info = blk_list[((size_t)bb * TC_PROCS + e) * 2u + 1u];
                if (info >> 31) { m_id = blk_base[bb] + ((info >> 16) & 0x7Fu); m_len = info & 0xFFFFu; }
                bb = 0; break;   // the process's last row before this block: an invocation (paired) or a completion (this one is stray)
              }
          }
        }
      }
      const bool matched = m_id != NONE;
      if (matched) {
        t_cmp[m_id] = idx; t_first[m_id] = base + rank;
        if (type == MSIM_T_OK) { t_off[m_id] = woff; t_lt[m_id] = len | (MSIM_T_OK << 16); }   // the completed form replaces the requested one
        else t_lt[m_id] = m_len | (type << 16);
      }
      c_ok += (u32)__popcll(__ballot(matched && type == MSIM_T_OK));
      c_fail += (u32)__popcll(__ballot(matched && type == MSIM_T_FAIL));
      c_info += (u32)__popcll(__ballot(matched && type == MSIM_T_INFO));
    }
    if (lane == 0) { atomicAdd(&hdr[2], c_ok); atomicAdd(&hdr[3], c_fail); atomicAdd(&hdr[4], c_info); }
  }
  __threadfence_block();
  LEAVE_IF_DECIDED();
  TL_MARK(0)
  const u32 n = hdr[1], c_ok = hdr[2], c_fail = hdr[3], c_info = hdr[4];
  res.op_count = n; res.attempt_count = n; res.ok_count = c_ok; res.stable_count = c_ok; res.fail_count = c_fail; res.info_count = c_info;

  // ---- B: ranges; the LDS tables laid out and cleared ----------------------------------------------------------------------------------------
  bool bad = false;
  {
    u32 max_key = 0, max_val = 0;
    for (u32 t = tid; t < n; t += NT) {
      const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
      for (u32 i = 0; i < wn;) { const Mop m = next_mop(w, wn, i); bad |= m.bad; max_key = max(max_key, m.key); if (m.f) max_val = max(max_val, m.val); }
    }
    max_key = t_max(max_key); max_val = t_max(max_val);
    if (lane == 0) { atomicMax(&hdr[5], max_key); atomicMax(&hdr[6], max_val); }
    if (bad) VERDICT(2);
  }
  LEAVE_IF_DECIDED();
  const u32 max_key = hdr[5], max_val = hdr[6];
  const u32 stride = max_val + 1u, K = max_key + 1u;
  // LDS (bytes, behind the header): in-degrees [n] and CSR offsets [n + 1] as halves of words | R1 = writer [K x stride] u16 + longest [K] u32,
  // later the ready queue [n] u16 + realtime ranges [n] 2 x u16 + positions [n] u32 | adjacency of the dependency edges, u16, whatever is left
  const u32 hw = (n + 2u) >> 1;                                   // words for n + 1 halves
  const u32 r1_a = ((K * stride + 1u) >> 1) + K, r1_b = ((n + 1u) >> 1) + 2u * n;
  const u32 r1_words = max(r1_a, r1_b);
  if (max_key >= KMAX || (u64)K * stride > WMAX || (u64)(2u * hw + r1_words) * 4u + 128u > p.lds_bytes) {   // (the same for every thread)
    if (tid == 0) { res.valid = NEEDS_HBM; p.out[hist] = res; }
    return;
  }
  u32 *const l_indeg = reinterpret_cast<u32 *>(tab), *const l_off = l_indeg + hw, *const l_r1 = l_off + hw;
  unsigned short *const l_writer = reinterpret_cast<unsigned short *>(l_r1);
  u32 *const l_longest = l_r1 + ((K * stride + 1u) >> 1);
  unsigned short *const l_adj = reinterpret_cast<unsigned short *>(l_r1 + r1_words);
  const u32 adj_cap = (p.lds_bytes - 64u - (2u * hw + r1_words) * 4u) / 2u;
  for (u32 i = tid; i < 2u * hw; i += NT) l_indeg[i] = 0;        // (indeg and off are adjacent)
  for (u32 i = tid; i < ((K * stride + 1u) >> 1); i += NT) l_r1[i] = 0xFFFFFFFFu;
  for (u32 k = tid; k < K; k += NT) l_longest[k] = 0;
  __syncthreads();
  TL_MARK(1)
#define WENT(k_, el_) ((el_) < stride ? (u32)l_writer[(k_) * stride + (el_)] : 0xFFFFu)   // 0xFFFF: nobody wrote it
#define W_TXN(e_) ((e_) & 0x1FFFu)
#define W_TYPE(e_) (((e_) >> 13) & 3u)
#define W_FIN(e_) ((e_) >> 15)

  // ---- C: writers (every transaction, whatever became of it); a second writer of a (key, element) finds the slot taken over ------------------
  for (int pass = 0; pass < 2; pass++) {
    for (u32 t = tid; t < n; t += NT) {
      const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu, ty = (t_lt[t] >> 16) & 3u;
      for (u32 i = 0; i < wn;) {
        const Mop m = next_mop(w, wn, i);
        if (!m.f) continue;
        u32 fin = 1u;   // no later append of this transaction to the key
        for (u32 i2 = i; i2 < wn;) { const Mop q = next_mop(w, wn, i2); if (q.f && q.key == m.key) fin = 0u; }
        const u32 ent = t | (ty << 13) | (fin << 15);
        if (pass == 0) l_writer[m.key * stride + m.val] = (unsigned short)ent;
        else if (l_writer[m.key * stride + m.val] != ent) bad = true;   // the generator never repeats (k, v)
      }
    }
    __syncthreads();
  }
  if (bad) VERDICT(2);
  LEAVE_IF_DECIDED();
  TL_MARK(2)

  // ---- D: the reads of :ok transactions ----------------------------------------------------------------------------------------------------------
  for (u32 t = tid; t < n; t += NT) {
    if ((t_lt[t] >> 16) != MSIM_T_OK) continue;
    const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
    u32 k = 0;
    for (u32 i = 0; i < wn; k++) {
      const Mop m = next_mop(w, wn, i);
      if (m.f) continue;
      { u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;   // duplicates
        for (u32 e = 0; e < m.len; e++) {
          const u32 x = elem(m.list, e); const u64 b = 1ull << (x & 63u); const u32 q = x >> 6;
          const u64 s = q == 0 ? s0 : q == 1 ? s1 : q == 2 ? s2 : s3;
          if (s & b) bad = true;
          s0 |= q == 0 ? b : 0; s1 |= q == 1 ? b : 0; s2 |= q == 2 ? b : 0; s3 |= q == 3 ? b : 0;
        } }
      { int prev = -1; Mop pm = m; u32 e_i = 0, e_k = 0;   // internal consistency: what the transaction's own earlier micro-ops imply for this read
        for (e_i = 0, e_k = 0; e_k < k; e_k++) { const Mop q = next_mop(w, wn, e_i); if (!q.f && q.key == m.key) { prev = (int)e_k; pm = q; } }
        const u32 e0 = prev < 0 ? 0u : (u32)prev + 1u;
        u32 n_app = 0;
        for (e_i = 0, e_k = 0; e_k < k; e_k++) { const Mop q = next_mop(w, wn, e_i); if (e_k >= e0 && q.f && q.key == m.key) n_app++; }
        bool ok = true; u32 at = 0;
        if (prev >= 0) { ok = m.len == pm.len + n_app; if (ok) for (u32 e = 0; e < pm.len; e++) ok &= elem(m.list, e) == elem(pm.list, e); at = pm.len; }
        else { ok = m.len >= n_app; at = m.len - n_app; }
        if (ok) { u32 a = 0; for (e_i = 0, e_k = 0; e_k < k; e_k++) { const Mop q = next_mop(w, wn, e_i); if (e_k >= e0 && q.f && q.key == m.key) { if (elem(m.list, at + a) != q.val) ok = false; a++; } } }
        if (!ok) bad = true; }
      u32 ext = m.len;   // the externally visible part: without the transaction's own appends at the tail
      while (ext > 0) { const u32 we = WENT(m.key, elem(m.list, ext - 1)); if (we != 0xFFFFu && W_TXN(we) == t) ext--; else break; }
      for (u32 e = 0; e < ext; e++) { const u32 we = WENT(m.key, elem(m.list, e)); if (we == 0xFFFFu || W_TYPE(we) == MSIM_T_FAIL) bad = true; }   // G1a
      if (ext > 0) { const u32 we = WENT(m.key, elem(m.list, ext - 1)); if (we != 0xFFFFu && W_TXN(we) != t && !W_FIN(we)) bad = true; }              // G1b
      atomicMax(&l_longest[m.key], ((m.len + 1u) << 24) | (u32)(m.list - pay));
    }
  }
  if (bad) VERDICT(2);

  TL_MARK(3)
  // realtime order in closed form (see txn_check_kernel): wavefront 0, beside the other wavefronts' share of pass D
  if (wave == 0) {
    u64 carry = ~0ull;
    for (int b = (int)((n + 63u) / 64u) - 1; b >= 0; b--) {
      const u32 t = (u32)b * 64u + lane;
      u64 v = (t < n && (t_lt[t] >> 16) == MSIM_T_OK) ? (((u64)t_cmp[t] << 32) | t_first[t]) : ~0ull;
      for (int o = 1; o < 64; o <<= 1) {
        const u32 ylo = (u32)__shfl_down((int)(u32)v, o), yhi = (u32)__shfl_down((int)(u32)(v >> 32), o);
        const u64 y = ((u64)yhi << 32) | ylo;
        if (lane + (u32)o < 64u) v = min(v, y);
      }
      v = min(v, carry);
      if (t < n) { sm[t] = (u32)(v >> 32); smf[t] = (u32)v; }
      carry = ((u64)t_rl((u32)(v >> 32), 0) << 32) | t_rl((u32)v, 0);
    }
    if (lane == 0) { sm[n] = NONE; smf[n] = n; }
  }
  LEAVE_IF_DECIDED();
  TL_MARK(4)

  // ---- E: edges: pass 0 counts degrees, pass 1 fills the CSR of the dependency edges (realtime successors stay a range) -----------------------------
  u32 n_edges = 0;
  for (int pass = 0; pass < 2; pass++) {
    u32 my_edges = 0, my_dep = 0;
#define ADD(a_, b_) do { const u32 ea = (a_), eb = (b_); if (ea != eb) { if (pass == 0) { h16_add(l_off, ea, 1u); h16_add(l_indeg, eb, 1u); my_edges++; my_dep++; } \
                                                                         else l_adj[h16_add(l_off, ea, 1u)] = (unsigned short)eb; } } while (0)
    for (u32 key = tid; key < K; key += NT) {   // ww along each key's version order (thread = key)
      const u32 L = l_longest[key];
      if (L == 0) continue;
      const u32 len = (L >> 24) - 1u; const u32 *ord = pay + (L & 0xFFFFFFu);
      for (u32 i = 0; i + 1 < len; i++) {
        const u32 a = WENT(key, elem(ord, i)), b = WENT(key, elem(ord, i + 1));
        if (a == 0xFFFFu || b == 0xFFFFu) continue;
        const bool fa = W_TYPE(a) == MSIM_T_FAIL, fb = W_TYPE(b) == MSIM_T_FAIL;
        if (fa && !fb) bad = true;   // dirty update
        if (!fa && !fb) ADD(W_TXN(a), W_TXN(b));
      }
    }
    for (u32 t = tid; t < n; t += NT) {   // wr / rw per read of an :ok transaction; its realtime successors
      if ((t_lt[t] >> 16) != MSIM_T_OK) continue;
      const u32 *w = pay + t_off[t]; const u32 wn = t_lt[t] & 0xFFFFu;
      for (u32 i = 0; i < wn;) {
        const Mop m = next_mop(w, wn, i);
        if (m.f) continue;
        const u32 L = l_longest[m.key];
        const u32 llen = (L >> 24) - 1u; const u32 *ord = pay + (L & 0xFFFFFFu);
        bool pre = m.len <= llen;
        if (pre) for (u32 e = 0; e < m.len; e++) pre &= elem(m.list, e) == elem(ord, e);
        if (!pre) { bad = true; continue; }   // incompatible order
        u32 ext = m.len;
        while (ext > 0) { const u32 we = WENT(m.key, elem(m.list, ext - 1)); if (we != 0xFFFFu && W_TXN(we) == t) ext--; else break; }
        if (ext > 0) { const u32 we = WENT(m.key, elem(m.list, ext - 1)); if (we != 0xFFFFu && W_TYPE(we) != MSIM_T_FAIL) ADD(W_TXN(we), t); }
        if (m.len < llen) { const u32 we = WENT(m.key, elem(ord, m.len)); if (we != 0xFFFFu && W_TYPE(we) != MSIM_T_FAIL) ADD(t, W_TXN(we)); }   // anti-dependency
      }
      if (pass == 0) {
        const u32 first = t_first[t];                          // the transactions invoked after t completed start here ...
        const u32 last = sm[first] == NONE ? n : smf[first];   // ... and end where the first of them to complete :ok did
        for (u32 v = first; v < last; v++) {
          h16_add(l_indeg, v, 1u);                             // (a :fail transaction has no edges of its own: counting this one in keeps Kahn's bookkeeping uniform)
          if ((t_lt[v] >> 16) != MSIM_T_FAIL) my_edges++;      // ... but it is not an edge of the graph the host counts
        }
      }
    }
#undef ADD
    if (bad) VERDICT(2);
    if (pass == 0) {
      my_edges = t_sum(my_edges); my_dep = t_sum(my_dep);
      if (lane == 0) { atomicAdd(&hdr[7], my_edges); atomicAdd(&hdr[8], my_dep); }
    }
    LEAVE_IF_DECIDED();
    if (pass == 0) {
      n_edges = hdr[7];
      const u32 n_dep = hdr[8];
      if (n_edges > p.emax) { if (tid == 0) p.out[hist] = res; return; }   // (NEEDS_HOST; the same for every thread)
      if (n_dep > adj_cap || n_dep > 65535u || n_edges > 65535u) { if (tid == 0) { res.valid = NEEDS_HBM; p.out[hist] = res; } return; }
      // out-degrees -> CSR offsets (exclusive prefix sums, 64 at a time, wavefront 0); pass 1 advances off[a] to the END of a's entries
      if (wave == 0) {
        u32 carry = 0;
        for (u32 base = 0; base <= n; base += 64) {
          const u32 t = base + lane;
          const u32 d = t < n ? h16_get(l_off, t) : 0u;
          const u32 ex = t_excl_scan(d, lane);
          const u32 tot = t_sum(d);
          t_wave_fence();
          // two lanes share a word: the even one writes both halves
          const u32 mine = carry + ex, next = (u32)__shfl_down((int)mine, 1);
          if (t <= n && !(t & 1u)) l_off[t >> 1] = mine | ((lane < 63u && t + 1u <= n ? next : 0u) << 16);
          carry += tot;
          t_wave_fence();
        }
      }
      __syncthreads();
    }
  }
#undef WENT
#undef W_TXN
#undef W_TYPE
#undef W_FIN

  TL_MARK(5)
  // ---- F: acyclic?  Realtime ranges, positions and (if needed) Kahn's queue where the writer table was ----------------------------------------------
  unsigned short *const l_queue = reinterpret_cast<unsigned short *>(l_r1);
  u32 *const l_rt = l_r1 + ((n + 1u) >> 1);   // first | last << 16
  u32 *const l_pos = l_rt + n;
  for (u32 t = tid; t < n; t += NT) {
    u32 first = 0, last = 0, pos = 0;
    if ((t_lt[t] >> 16) == MSIM_T_OK) { first = t_first[t]; last = sm[first] == NONE ? n : smf[first]; pos = t_cmp[t] + 1u; }
    l_rt[t] = first | (last << 16);
    l_pos[t] = pos;
  }
  if (tid == 0) { hdr[9] = 0; hdr[10] = 0; }
  __syncthreads();
  // F1: a POTENTIAL instead of a queue.  The graph is acyclic iff positions exist that grow along every edge.  Start from where a clean
  // history of a serializable store nearly is — an :ok transaction at its completion row (every realtime edge already grows: u's completion
  // precedes v's invocation, hence v's completion), an indeterminate one at 0 — and raise the head of every edge that does not grow
  // (pos[w] = max(pos[w], pos[t] + 1), all threads, all edges, until a sweep changes nothing).  What is out of order at the start are
  // dependencies between transactions that were open together and completed the other way round; a raise travels on through the few
  // transactions completed inside the lifetime of a slower one, so a handful of sweeps settle it, every one of them the whole workgroup's
  // work (Kahn's steps are one wavefront's, a ready set about as wide as the clients' concurrency: 55 % of the kernel before).  A cycle
  // never settles: after TC_SWEEPS sweeps Kahn's algorithm below decides, exactly as before.
  bool proven = false; u32 sweeps = 0;
#ifndef TC_SWEEPS
#define TC_SWEEPS 48u
#endif
#ifndef TC_NO_POTENTIAL
  for (u32 sweep = 0; sweep < TC_SWEEPS; sweep++) {
    bool raised = false;
    for (u32 t = tid; t < n; t += NT) {
      const u32 a1 = h16_get(l_off, t), a0 = t ? h16_get(l_off, t - 1u) : 0u;   // (after pass 1 off[t] is the end of t's entries)
      const u32 rt = l_rt[t], r0 = rt & 0xFFFFu, r1 = rt >> 16;
      const u32 mine = l_pos[t] + 1u;
      for (u32 k = a0; k < a1; k++) { const u32 w = l_adj[k]; if (l_pos[w] < mine) { atomicMax(&l_pos[w], mine); raised = true; } }
      for (u32 w = r0; w < r1; w++) if (l_pos[w] < mine) { atomicMax(&l_pos[w], mine); raised = true; }
    }
    u32 *const flag = &hdr[9 + (sweep & 1u)];   // (two words in turn: a thread still reading this sweep's never meets the next sweep's write)
    if (raised) *flag = sweep + 1u;
    __syncthreads();
    sweeps = sweep + 1u;
    if (*flag != sweep + 1u) { proven = true; break; }
  }
#endif
  if (wave != 0) return;
  if (proven) {
    if (lane == 0) {
      res.lost_count = n_edges;   // edges of the dependency graph
      res.valid = flags ? 0u : (c_ok == 0 ? 2u : 1u);
#ifdef TC_PROF
      TL_MARK(6)
      for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = tl[i];
      res.never_read_count = tl[5]; res.duplicated_count = tl[6]; res.stale_count = sweeps; res.error_count = tl_real0; res.stable_count = (u32)__builtin_readsteadycounter();
#endif
      p.out[hist] = res;
    }
    return;
  }
  (void)sweeps;
  // F2: Kahn's algorithm, 64 ready transactions per step — the queue is one wavefront's work
  u32 tail = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 t = base + lane;
    const bool z = t < n && h16_get(l_indeg, t) == 0;
    const u64 zm = __ballot(z);
    if (z) l_queue[tail + (u32)__popcll(zm & lt)] = (unsigned short)t;
    tail += (u32)__popcll(zm);
  }
  t_wave_fence();
  // A step takes up to 64 ready transactions; the graph is deep and narrow (the realtime order: a ready set is about as wide as the
  // clients' concurrency), so the wavefront's lanes are split among the ready ones — G lanes each, a power of two — and a transaction's
  // successors are taken G at a time: a step is one or two turns of the loop below instead of one per successor (78 % of a history's
  // cycles before: profiles/r03w_txn_check_phases.txt).
  u32 head = 0;
  while (head < tail) {
    const u32 cnt = min(64u, tail - head);
    const u32 gsh = cnt <= 1u ? 6u : 6u - (32u - (u32)__clz(cnt - 1u));   // log2 of the lanes per ready transaction: 64 >> ceil(log2(cnt))
    const u32 ti = lane >> gsh, sub = lane & ((1u << gsh) - 1u), G = 1u << gsh;
    const bool on = ti < cnt;
    const u32 v = on ? (u32)l_queue[head + ti] : 0u;
    const u32 a1 = on ? h16_get(l_off, v) : 0u, a0 = on ? (v ? h16_get(l_off, v - 1u) : 0u) : 0u;   // (after pass 1 off[v] is the end of v's entries)
    const u32 rt = on ? l_rt[v] : 0u, r0 = rt & 0xFFFFu, r1 = rt >> 16;
    const u32 deg = (a1 - a0) + (r1 - r0);
    for (u32 k = sub; __ballot(k < deg); k += G) {
      bool push = false; u32 wv = 0;
      if (k < deg) { wv = k < a1 - a0 ? (u32)l_adj[a0 + k] : r0 + (k - (a1 - a0)); push = h16_sub(l_indeg, wv) == 1u; }
      const u64 pm = __ballot(push);
      if (push) l_queue[tail + (u32)__popcll(pm & lt)] = (unsigned short)wv;
      tail += (u32)__popcll(pm);
    }
    head += cnt;
    t_wave_fence();
  }
  if (lane == 0) {
    if (tail == n) {   // (else a cycle: the host finds and classifies it — res.valid is still NEEDS_HOST)
      res.lost_count = n_edges;   // edges of the dependency graph
      res.valid = flags ? 0u : (c_ok == 0 ? 2u : 1u);
    }
#ifdef TC_PROF
    TL_MARK(6)
    for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = tl[i];
    res.never_read_count = tl[5]; res.duplicated_count = tl[6]; res.stale_count = 1000u + sweeps;   // (Kahn decided)
    res.error_count = tl_real0; res.stable_count = (u32)__builtin_readsteadycounter();
#endif
    p.out[hist] = res;
  }
#undef VERDICT
#undef LEAVE_IF_DECIDED
}

// words of workspace per history: the tables of txn_check_kernel / the per-transaction words of txn_check_lds_kernel
uint64_t ws_words_for(u32 nmax, u32 emax) { return (uint64_t)nmax * 11 + 4 + KMAX + WMAX + emax; }
uint64_t ws_words_lds(u32 nmax) { return (uint64_t)nmax * 6 + 8; }

// LDS of a workgroup of txn_check_lds_kernel: what a history of `nmax` transactions over `keys` keys with elements below `stride`
// needs with 4.5 dependency edges per transaction, at most 78 KiB (two workgroups per CU)
u32 lds_bytes_for(u32 nmax, u32 keys, u32 stride) {
  const uint64_t hw = (nmax + 2u) / 2, r1 = std::max<uint64_t>(((uint64_t)keys * stride + 1) / 2 + keys, (nmax + 1u) / 2 + 2ull * nmax);
  const uint64_t need = (2 * hw + r1) * 4 + (uint64_t)nmax * 9 + 256;
  return (u32)std::min<uint64_t>(78 * 1024, std::max<uint64_t>(8 * 1024, (need + 255) & ~255ull));
}

int grow_ws(msim_ctx *ctx, void **ws_buf, size_t *ws_cap, size_t need) {
  if (*ws_cap >= need) return MSIM_OK;
  if (*ws_buf) (void)msim_dev_free(*ws_buf);
  *ws_buf = nullptr; *ws_cap = 0;
  MSIM_HIP_TRY(ctx, msim_dev_malloc(ws_buf, need));
  *ws_cap = need;
  return MSIM_OK;
}

int txn_dev_run(msim_ctx *ctx, TParams tp, u32 n, u32 cm, const std::vector<msim_inst_meta> *hmeta, msim_check_result *h_out, hipStream_t st, u32 *n_host,
                void **ws_buf, size_t *ws_cap) {
  const bool trace = (msim_dev_flags(ctx) & 0x1000u) != 0;   // developer: time the passes
  // Which kernel takes the first pass: the one with its tables in LDS, a workgroup of sixteen wavefronts per history.  Measured on cfg5,
  // 32768 histories (profiles/r03b_cfg5_txn_check_*, r03k_cfg5_txn_check.txt, r03x_txn_check.txt): tables in an HBM workspace, one
  // wavefront per history: 82 GB of HBM traffic, 130 ms (64 histories in flight per CU); tables in LDS (78 KiB: two histories per CU),
  // 22 GB: 264 ms with one wavefront per history; 149 ms with the streaming passes spread over eight; 75 ms once Kahn's steps — 78 % of
  // what was left: a ready set as wide as the clients' concurrency, one turn of a loop per successor — split the wavefront's lanes among
  // the ready transactions.  MSIM_DEV_FLAGS bit 13 (0x2000) keeps every history on the HBM-table kernel (which stays the second pass
  // for histories whose tables do not fit); MSIM_TXN_WG = threads per history (64 .. 1024, the default: 66 ms; 512: 75 ms).
  const bool hbm_only = (msim_dev_flags(ctx) & 0x2000u) != 0;
  static const u32 wg_threads = []() { const char *e = std::getenv("MSIM_TXN_WG"); u32 v = e ? (u32)std::atoi(e) : 1024u; v = (v / 64u) * 64u; return v < 64u ? 64u : v > 1024u ? 1024u : v; }();
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const uint64_t budget = 6ull << 30;   // as many histories per launch as a few GB of workspace hold (every one of them has its own slice)
  int rc;
  std::vector<u32> big;   // histories for txn_check_kernel
  if (!hbm_only) {
    // pass 1: tables in LDS
    tp.ws_words = ws_words_lds(tp.nmax); tp.list = nullptr;
    const u32 chunk = (u32)std::min<uint64_t>(n, std::max<uint64_t>(1, budget / (tp.ws_words * 4)));
    if ((rc = grow_ws(ctx, ws_buf, ws_cap, (size_t)chunk * tp.ws_words * 4)) != MSIM_OK) return rc;
    tp.ws = static_cast<u32 *>(*ws_buf);
    if (tp.lds_bytes > 64 * 1024) MSIM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(txn_check_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tp.lds_bytes));
    for (u32 first = 0; first < n; first += chunk) {
      tp.first = first;
      hipLaunchKernelGGL(txn_check_lds_kernel, dim3(std::min(chunk, n - first)), dim3(wg_threads), tp.lds_bytes, st, tp);
      MSIM_HIP_TRY(ctx, hipGetLastError());
    }
    MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_out, tp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st));
    MSIM_HIP_TRY(ctx, hipStreamSynchronize(st));
    for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HBM) big.push_back(i);
    if (trace) std::fprintf(stderr, "[txn-check] LDS pass (%u B per workgroup): %.2f ms, %zu of %u histories do not fit\n", tp.lds_bytes, ms(), big.size(), n);
  } else { big.resize(n); for (u32 i = 0; i < n; i++) big[i] = i; }
  if (!big.empty()) {
    // pass 2: the histories whose tables do not fit LDS, tables in an HBM workspace
    u32 *d_list = nullptr;
    MSIM_HIP_TRY(ctx, msim_dev_malloc(&d_list, big.size() * 4));
    MSIM_HIP_TRY(ctx, hipMemcpy(d_list, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    tp.ws_words = ws_words_for(tp.nmax, tp.emax); tp.list = d_list;
    const u32 nb = (u32)big.size();
    const u32 chunk = (u32)std::min<uint64_t>(nb, std::max<uint64_t>(1, budget / (tp.ws_words * 4)));
    rc = grow_ws(ctx, ws_buf, ws_cap, (size_t)chunk * tp.ws_words * 4);
    if (rc == MSIM_OK) {
      tp.ws = static_cast<u32 *>(*ws_buf);
      for (u32 first = 0; first < nb; first += chunk) {
        tp.first = first;
        hipLaunchKernelGGL(txn_check_kernel, dim3(std::min(chunk, nb - first)), dim3(64), 0, st, tp);
        if (hipGetLastError() != hipSuccess) { rc = MSIM_E_HIP; ctx->err = "txn_check_kernel launch"; break; }
      }
    }
    if (rc == MSIM_OK && (hipMemcpyAsync(h_out, tp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) { rc = MSIM_E_HIP; ctx->err = "txn check: copy of the results"; }
    (void)msim_dev_free(d_list);
    if (rc != MSIM_OK) return rc;
    if (trace) std::fprintf(stderr, "[txn-check] HBM-table pass over %u histories: done at %.2f ms\n", nb, ms());
  }
  if (ctx) ctx->txn_big = (u32)big.size();
  std::vector<u32> todo;
  for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HOST || h_out[i].valid == NEEDS_HBM) todo.push_back(i);
  if (trace) std::fprintf(stderr, "[txn-check] device passes: %.2f ms, %zu of %u histories for the host\n", ms(), todo.size(), n);
  if (!todo.empty()) {
    std::vector<uint64_t> ro, po;
    if (tp.row_off) { ro.resize(n + 1); po.resize(n + 1);
      MSIM_HIP_TRY(ctx, hipMemcpy(ro.data(), tp.row_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
      MSIM_HIP_TRY(ctx, hipMemcpy(po.data(), tp.pay_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost)); }
    std::vector<std::vector<msim_op>> rows(todo.size());
    std::vector<std::vector<u32>> pays(todo.size());
    for (size_t k = 0; k < todo.size(); k++) {
      const u32 i = todo[k];
      const u32 nr = hmeta ? (*hmeta)[i].n_rows : (u32)(ro[i + 1] - ro[i]), nw = hmeta ? (*hmeta)[i].n_payload_words : (u32)(po[i + 1] - po[i]);
      rows[k].resize(nr ? nr : 1); pays[k].resize(nw ? nw : 1);
      if (nr) MSIM_HIP_TRY(ctx, hipMemcpy(rows[k].data(), tp.rows + (hmeta ? (uint64_t)i * tp.max_rows : ro[i]), (size_t)nr * sizeof(msim_op), hipMemcpyDeviceToHost));
      if (nw) MSIM_HIP_TRY(ctx, hipMemcpy(pays[k].data(), tp.payload + (hmeta ? (uint64_t)i * tp.max_pay : po[i]), (size_t)nw * 4, hipMemcpyDeviceToHost));
    }
    unsigned nt = msim_host_threads();
    if (nt > todo.size()) nt = (unsigned)todo.size();
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; w++)
      th.emplace_back([&, w]() {
        for (size_t k = w; k < todo.size(); k += nt) {
          const u32 i = todo[k];
          msim_txn_check_instance_host(rows[k].data(), hmeta ? (*hmeta)[i].n_rows : (u32)(ro[i + 1] - ro[i]), pays[k].data(),
                                       hmeta ? (*hmeta)[i].n_payload_words : (u32)(po[i + 1] - po[i]), hmeta ? (*hmeta)[i].flags : 0u, cm, &h_out[i]);
        }
      });
    for (auto &x : th) x.join();
    for (u32 i : todo) MSIM_HIP_TRY(ctx, hipMemcpy(tp.out + i, &h_out[i], sizeof(msim_check_result), hipMemcpyHostToDevice));
    if (trace) std::fprintf(stderr, "[txn-check] host analysis of those: done at %.2f ms\n", ms());
  }
  if (n_host) *n_host = (u32)todo.size();
  return MSIM_OK;
}

}  // namespace

// msim_check for txn-list-append: the histories of the last run, where they lie in HBM.
int msim_check_txn_device(msim_ctx *ctx) {
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const u32 n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<msim_inst_meta> hm(n);
  MSIM_HIP_TRY(ctx, hipMemcpy(hm.data(), ctx->d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
  TParams tp;
  tp.rows = ctx->d_rows; tp.payload = ctx->d_payload; tp.meta = ctx->d_meta; tp.row_off = nullptr; tp.pay_off = nullptr; tp.out = ctx->d_check;
  tp.max_rows = ctx->cfg.max_rows; tp.max_pay = ctx->cfg.max_payload_words;
  tp.nmax = ctx->cfg.max_rows / 2 + 1; tp.emax = tp.nmax * 16; tp.first = 0; tp.ws = nullptr; tp.ws_words = 0; tp.list = nullptr;
  tp.lds_bytes = lds_bytes_for(tp.nmax, ctx->cfg.max_values, ctx->cfg.max_writes_per_key + 1);
  u32 redone = 0;
  int rc = txn_dev_run(ctx, tp, n, ctx->cfg.consistency_model, &hm, ctx->h_check, ctx->stream, &redone, &ctx->d_check_scratch, &ctx->cap_check_scratch);
  if (rc != MSIM_OK) return rc;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->lin_host_rechecks = redone;
  ctx->checked = true; ctx->check_fetched = true;
  return MSIM_OK;
}

// Checks `n_histories` list-append histories given on the host (rows / payload words of history i at row_offsets[i] /
// payload_offsets[i]) with the device pass of msim_check on HIP device `device`; out[i] as msim_check_txn_rows would fill it.
extern "C" int msim_check_txn_batch(int device, const msim_op *rows, const uint64_t *row_offsets, const uint32_t *payload, const uint64_t *payload_offsets,
                                    uint32_t n_histories, msim_check_result *out) {
  if (!rows || !row_offsets || !payload_offsets || !out || n_histories == 0) return MSIM_E_INVALID;
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
    TParams tp;
    tp.rows = d_rows; tp.payload = d_pay; tp.meta = nullptr; tp.row_off = d_ro; tp.pay_off = d_po; tp.out = d_out;
    tp.max_rows = 0; tp.max_pay = 0; tp.nmax = max_r / 2 + 65; tp.emax = tp.nmax * 16; tp.first = 0; tp.ws = nullptr; tp.ws_words = 0; tp.list = nullptr;
    tp.lds_bytes = lds_bytes_for(tp.nmax, std::min<u32>(KMAX, tp.nmax * 2 + 16), 17);
    rc = txn_dev_run(ctx, tp, n_histories, MSIM_CM_STRICT_SERIALIZABLE, nullptr, out, nullptr, nullptr, &ws, &ws_cap);
  } while (false);
  for (void *q : {(void *)d_rows, (void *)d_pay, (void *)d_ro, (void *)d_po, (void *)d_out, ws}) if (q) (void)msim_dev_free(q);
  return rc;
}
