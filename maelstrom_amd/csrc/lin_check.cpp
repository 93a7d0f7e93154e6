// lin_check.cpp — per-key linearizability of lin-kv histories (host side of msim_check for MSIM_WL_LIN_KV).
//
// In the reference the lin-kv workload's checker is [upstream] jepsen.tests.linearizable-register:
// `independent/checker` over Knossos' `checker/linearizable` with a CAS-register model
// (workload/lin_kv.clj:84).  Restated here as just-in-time linearization (Lowe; Knossos' "linear" analysis):
// walk one key's history; keep the set of configurations (register value, set of pending ops already
// linearized); when an op returns, every surviving configuration must have linearized it.  :fail ops never
// happened; :info ops stay pending forever (they may take effect at any later time, or never).
// Histories are small per key (process-limit 20 retires a key), so this runs on the host cores, one
// thread per slice of instances.  msim_check runs the device version (lin_check_dev.hip, one wavefront per history) and
// comes here only for histories that exceed it; MSIM_DEV_FLAGS bit 11 keeps everything on the host.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "engine_internal.h"

namespace {

struct Op { uint32_t f, v1, v2; bool ok; bool skip; };  // effective op of a call

struct Cfg { uint64_t lin; uint32_t val; };

// The configurations seen while one operation returns, grouped by (register value, linearized RETURNING ops); per group the
// minimal sets of linearized never-returning ops.  Open addressing with a generation stamp and one node pool: clearing it is O(1)
// and nothing is allocated per event (the std::unordered_map<Cfg, std::vector<..>> it replaces spent most of the search in
// malloc / free: 15 -> 9 ms per 12 000-row history with partitions).
struct Seen {
  struct Slot { uint64_t base; uint32_t val, gen; int head; };
  struct Node { uint64_t ib; int next; bool dead; };
  std::vector<Slot> slots = std::vector<Slot>(1024, Slot{0, 0, 0, -1});
  std::vector<Node> pool;
  std::vector<uint32_t> used;   // slots of this generation, in insertion order
  uint32_t gen = 1;
  void clear() { if (++gen == 0) { for (Slot &s : slots) s.gen = 0; gen = 1; } pool.clear(); used.clear(); }
  static size_t hash(uint64_t base, uint32_t val) { uint64_t h = (base ^ ((uint64_t)val << 56)) * 0x9E3779B97F4A7C15ull; return (size_t)(h ^ (h >> 29)); }
  void grow() {
    std::vector<Slot> old; old.swap(slots);
    slots.assign(old.size() * 2, Slot{0, 0, 0, -1});
    for (uint32_t &u : used) {
      const Slot &o = old[u];
      size_t h = hash(o.base, o.val) & (slots.size() - 1);
      while (slots[h].gen == gen) h = (h + 1) & (slots.size() - 1);
      slots[h] = o; u = (uint32_t)h;
    }
  }
  Slot *find(uint64_t base, uint32_t val, bool add) {
    if (add && (used.size() + 1) * 2 > slots.size()) grow();
    size_t h = hash(base, val) & (slots.size() - 1);
    for (;; h = (h + 1) & (slots.size() - 1)) {
      Slot &s = slots[h];
      if (s.gen != gen) { if (!add) return nullptr; s = Slot{base, val, gen, -1}; used.push_back((uint32_t)h); return &s; }
      if (s.base == base && s.val == val) return &s;
    }
  }
};

// apply op to register value `val` (0..4, 0xFF = nil); returns legal?
inline bool step(uint32_t val, const Op &op, uint32_t *out) {
  *out = val;
  if (op.f == MSIM_F_READ) return val == op.v1;             // only :ok reads are stepped; they must see the current value
  if (op.f == MSIM_F_WRITE) { *out = op.v1; return true; }
  if (val != op.v1) return false;                           // cas [v v']
  *out = op.v2; return true;
}

// one key: rows (indices into the instance history) in order.  Returns 1 linearizable, 0 not, 2 unknown (too wide).
int check_key(const msim_op *rows, const std::vector<uint32_t> &idx) {
  // pair invocations with completions by process
  std::unordered_map<uint32_t, uint32_t> open;         // process -> position in idx of its invoke
  std::vector<int> comp(idx.size(), -1);               // invoke position -> completion position
  for (uint32_t k = 0; k < idx.size(); k++) {
    const msim_op &r = rows[idx[k]];
    const uint32_t proc = MSIM_OP_PROCESS(r);
    if (MSIM_OP_TYPE(r) == MSIM_T_INVOKE) open[proc] = k;
    else { auto it = open.find(proc); if (it != open.end()) { comp[it->second] = (int)k; open.erase(it); } }
  }
  struct Ev { bool call; uint32_t id; };
  std::vector<Ev> events;
  std::vector<Op> eff(idx.size());
  std::vector<int> inv_of(idx.size(), -1);
  for (uint32_t k = 0; k < idx.size(); k++) {
    const msim_op &r = rows[idx[k]];
    if (MSIM_OP_TYPE(r) == MSIM_T_INVOKE) {
      const int c = comp[k];
      if (c >= 0 && MSIM_OP_TYPE(rows[idx[c]]) == MSIM_T_FAIL) continue;      // never happened
      const bool ok = c >= 0 && MSIM_OP_TYPE(rows[idx[c]]) == MSIM_T_OK;
      const msim_op &src = ok ? rows[idx[c]] : r;                              // an :ok read carries the value it saw
      Op o; o.f = MSIM_OP_F(r); o.v1 = (src.value >> 8) & 0xFF; o.v2 = (src.value >> 16) & 0xFF; o.ok = ok;
      o.skip = !ok && o.f == MSIM_F_READ;                                      // an unfinished read constrains nothing
      eff[k] = o;
      events.push_back({true, k});
      if (c >= 0) inv_of[c] = (int)k;
    } else if (MSIM_OP_TYPE(r) == MSIM_T_OK && inv_of[k] >= 0) events.push_back({false, (uint32_t)inv_of[k]});
  }
  // pending ops get slots 0..63 in a bitmask
  std::unordered_map<uint32_t, uint32_t> slot_of;
  std::vector<uint32_t> op_in_slot(64, 0);
  uint64_t pending = 0, info_bits = 0;   // info_bits: pending ops that will never return (:info, or no completion)
  uint64_t twins_below[64];              // per never-returning op: the never-returning ops in LOWER slots that are the same operation
  // Dominance: for equal (register value, linearized returning ops), a configuration that has linearized FEWER
  // never-returning ops can still do everything the other can (it may apply them later, or never).  Only the
  // minimal ones are kept — without this, k indeterminate writes cost 2^k configurations.
  static thread_local Seen seen;
  auto admit = [&](const Cfg &c) -> bool {   // true if c is not dominated; removes what c dominates
    Seen::Slot *g = seen.find(c.lin & ~info_bits, c.val, true);
    const uint64_t ib = c.lin & info_bits;
    for (int i = g->head; i >= 0; i = seen.pool[(size_t)i].next) { const Seen::Node &nd = seen.pool[(size_t)i]; if (!nd.dead && (nd.ib & ib) == nd.ib) return false; }   // an existing subset dominates c
    for (int i = g->head; i >= 0; i = seen.pool[(size_t)i].next) { Seen::Node &nd = seen.pool[(size_t)i]; if (!nd.dead && (nd.ib & ib) == ib) nd.dead = true; }
    seen.pool.push_back(Seen::Node{ib, g->head, false});
    g->head = (int)seen.pool.size() - 1;
    return true;
  };
  auto is_minimal = [&](const Cfg &c) -> bool {
    const Seen::Slot *g = seen.find(c.lin & ~info_bits, c.val, false);
    const uint64_t ib = c.lin & info_bits;
    if (g) for (int i = g->head; i >= 0; i = seen.pool[(size_t)i].next) { const Seen::Node &nd = seen.pool[(size_t)i]; if (!nd.dead && nd.ib == ib) return true; }
    return false;
  };
  std::vector<Cfg> configs{{0, 0xFF}}, stack, out;
  for (const Ev &ev : events) {
    if (ev.call) {
      if (pending == ~0ull) return 2;
      const uint32_t s = (uint32_t)__builtin_ctzll(~pending);
      pending |= 1ull << s; slot_of[ev.id] = s; op_in_slot[s] = ev.id;
      if (!eff[ev.id].ok) {
        const Op &o = eff[ev.id];
        twins_below[s] = 0;
        for (uint64_t m = info_bits; m && !o.skip; m &= m - 1) {
          const uint32_t j = (uint32_t)__builtin_ctzll(m); const Op &t = eff[op_in_slot[j]];
          if (t.f == o.f && t.v1 == o.v1 && t.v2 == o.v2 && !t.skip) { if (j < s) twins_below[s] |= 1ull << j; else twins_below[j] |= 1ull << s; }
        }
        info_bits |= 1ull << s;
      }
      continue;
    }
    const uint32_t s = slot_of[ev.id];
    const uint64_t bit = 1ull << s;
    seen.clear(); out.clear(); stack.clear();
    for (const Cfg &c : configs) if (admit(c)) stack.push_back(c);
    size_t explored = 0;
    while (!stack.empty()) {
      const Cfg c = stack.back(); stack.pop_back();
      if (!is_minimal(c)) continue;   // superseded by a smaller configuration found later
      if (++explored > 2000000) return 2;
      if (c.lin & bit) { out.push_back({c.lin & ~bit, c.val}); continue; }
      uint64_t cand = pending & ~c.lin;
      while (cand) {
        const uint32_t j = (uint32_t)__builtin_ctzll(cand); cand &= cand - 1;
        const Op &o = eff[op_in_slot[j]];
        if (o.skip) continue;
        // Symmetry: never-returning calls that are the same operation are interchangeable from now on (none of them constrains
        // anything later), so of those not yet linearized only the one in the lowest slot is tried — exact, and it keeps k
        // identical timed-out writes from costing 2^k configurations that dominance cannot compare.
        if (((info_bits >> j) & 1) && (twins_below[j] & ~c.lin)) continue;
        uint32_t nv;
        if (!step(c.val, o, &nv)) continue;
        const Cfg c2{c.lin | (1ull << j), nv};
        if (admit(c2)) stack.push_back(c2);
      }
    }
    pending &= ~bit; slot_of.erase(ev.id);
    // re-minimise the survivors (bit s is gone from their keys)
    seen.clear(); configs.clear();
    for (const Cfg &c : out) (void)admit(c);
    for (uint32_t u : seen.used) { const Seen::Slot &g = seen.slots[u];
      for (int i = g.head; i >= 0; i = seen.pool[(size_t)i].next) if (!seen.pool[(size_t)i].dead) configs.push_back({g.base | seen.pool[(size_t)i].ib, g.val}); }
    if (configs.empty()) return 0;
  }
  return 1;
}

void check_instance(const msim_op *rows, uint32_t n_rows, uint32_t flags, msim_check_result *res) {
  std::memset(res, 0, sizeof *res);
  std::unordered_map<uint32_t, std::vector<uint32_t>> by_key;
  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    if (MSIM_OP_PROCESS(r) == MSIM_PROCESS_NEMESIS) continue;
    const uint32_t t = MSIM_OP_TYPE(r);
    if (t == MSIM_T_INVOKE) res->op_count++; else if (t == MSIM_T_OK) res->ok_count++; else if (t == MSIM_T_FAIL) res->fail_count++; else res->info_count++;
    const uint32_t f = MSIM_OP_F(r);
    if (f == MSIM_F_READ || f == MSIM_F_WRITE || f == MSIM_F_CAS) by_key[r.value & 0xFF].push_back(i);
  }
  uint32_t bad = 0, unknown = 0;
  for (auto &kv : by_key) { const int v = check_key(rows, kv.second); bad += v == 0; unknown += v == 2; }
  res->attempt_count = (uint32_t)by_key.size();    // keys checked (independent/checker)
  res->error_count = bad;                          // keys whose history is not linearizable
  res->valid = flags ? 0u : bad ? 0u : unknown ? 2u : 1u;
}

}  // namespace

// the host search for one history (lin_check_dev.hip hands over what exceeds the device's registers)
void msim_lin_check_instance_host(const msim_op *rows, uint32_t n_rows, uint32_t flags, msim_check_result *res) { check_instance(rows, n_rows, flags, res); }

// Host-only entry point: checks one lin-kv history given as rows (no device involved).
extern "C" int msim_check_lin_kv_rows(const msim_op *rows, uint32_t n_rows, msim_check_result *out) {
  if (!rows || !out) return MSIM_E_INVALID;
  check_instance(rows, n_rows, 0, out);
  return MSIM_OK;
}

// Runs the lin-kv checker over the fetched histories of the last run; results to ctx->h_check and ctx->d_check.
int msim_check_lin_kv_host(msim_ctx *ctx) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc = msim_fetch(ctx);
  if (rc != MSIM_OK) return rc;
  const uint32_t n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  unsigned nt = msim_host_threads();
  if (nt > n) nt = n;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([ctx, n, nt, t]() {
      for (uint32_t i = t; i < n; i += nt)
        check_instance(ctx->h_rows + ctx->h_row_off[i], ctx->h_meta[i].n_rows, ctx->h_meta[i].flags, &ctx->h_check[i]);
    });
  for (auto &x : th) x.join();
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->d_check, ctx->h_check, (size_t)n * sizeof(msim_check_result), hipMemcpyHostToDevice));
  ctx->checked = true; ctx->check_fetched = true; ctx->lin_host_rechecks = n;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return MSIM_OK;
}
