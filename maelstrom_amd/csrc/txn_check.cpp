// txn_check.cpp — host-side transaction checkers: list-append (txn-list-append) and rw-register (txn-rw-register).
//
// What the reference wires in at workload/txn_list_append.clj:142 is [upstream] jepsen.tests.cycle.append, i.e. elle's
// list-append analysis, asked for --consistency-models strict-serializable by default (core.clj:160-165).  elle is not
// vendored in /root/reference; this restates its published algorithm (Kingsbury & Alvaro, "Elle: Inferring Isolation
// Anomalies from Experimental Observations", VLDB 2020, §4-§5):
//   * every element is appended at most once per key, so a read [x1 .. xn] of key k reveals the version order of k up to
//     xn; the longest read of a key gives its order, every other read must be a prefix of it (else incompatible-order);
//   * ww: writer(x_i) -> writer(x_i+1);  wr: writer(last element read) -> reader;  rw: reader of a list ending at x_i
//     (or of nil) -> writer(x_i+1);  realtime: T1 completed before T2 was invoked (transitively reduced);
//   * a cycle of ww edges is G0, of ww+wr G1c, with exactly one rw G-single, with more G2; a cycle that needs a realtime
//     edge is the -realtime variant (forbidden under strict serializability only);
//   * non-cycle anomalies: G1a (read of an element written by a failed transaction), G1b (read of an intermediate
//     state of another transaction's appends to a key), internal (a read that contradicts the transaction's own earlier
//     reads/appends), duplicate elements, dirty update (a committed append on top of an aborted one).
// :info transactions (timeouts) may or may not have happened: their appends count as writes when observed; they never
// complete, so they have no outgoing realtime edges.  :fail transactions are not part of the graph.
//
// Host code: the graph is pointer-chasing over a few thousand transactions per history — threads over histories here,
// the simulation itself stays on the GPU.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "engine_internal.h"

namespace {

// All working storage of one checker thread: flat arenas that are cleared, not freed, between histories (256 threads
// allocating and unmapping per-history vectors serialise on the kernel's address-space lock).
struct Mop { uint8_t f; uint8_t val; uint8_t nil; uint8_t len; uint16_t key; uint32_t off; };  // list = bytes[off, off+len)
struct Txn { uint32_t process; int inv, cmp; uint8_t type; uint32_t mop0, n_mops, rt0, n_rt; };
enum { E_WW = 1, E_WR = 2, E_RW = 4, E_RT = 8 };
struct Edge { uint32_t from, to; uint8_t kind; };

struct Scratch {
  std::vector<Txn> txns; std::vector<Mop> mops; std::vector<uint8_t> bytes; std::vector<uint32_t> rt, frontier, nf;
  std::vector<int> open;        // process -> txn
  std::vector<int> writer;      // (key << 8 | element) -> txn
  std::vector<int> longest;     // key -> mop index of the longest read
  std::vector<Edge> edges; std::vector<uint32_t> adj_off; std::vector<Edge> adj;
  std::vector<uint64_t> vsucc, vseen;  // rw-register: version graph per key
  std::vector<int> idx, low, comp; std::vector<char> on, seen; std::vector<uint32_t> st, work, pos, q;
};

void parse_txn(Scratch &S, const uint32_t *w, uint32_t n, bool rw) {
  for (uint32_t i = 0; i < n;) {
    const uint32_t h = w[i++];
    Mop m; m.f = h & 1; m.key = (h >> 1) & 0x7FFF; m.val = 0; m.nil = 0; m.len = 0; m.off = (uint32_t)S.bytes.size();
    const uint32_t x = (h >> 16) & 0xFF;
    if (m.f) m.val = (uint8_t)x;
    else if (x == 0xFF) m.nil = 1;
    else if (rw) m.val = (uint8_t)x;  // a register read: the value itself
    else {
      for (uint32_t e = 0; e < x && i + e / 4 < n; e++) S.bytes.push_back((uint8_t)(w[i + e / 4] >> (8 * (e % 4))));
      m.len = (uint8_t)(S.bytes.size() - m.off);
      i += (x + 3) / 4;
    }
    S.mops.push_back(m);
  }
}

// strongly connected components (iterative Tarjan) over the CSR edges whose kind is in `mask`; comp[v] = component id,
// returns the number of vertices that sit in a component of size > 1
uint32_t scc(Scratch &S, uint32_t n, uint8_t mask) {
  S.idx.assign(n, -1); S.low.assign(n, 0); S.on.assign(n, 0); S.comp.assign(n, -1);
  S.st.clear(); S.work.clear(); S.pos.clear();
  int counter = 0, ncomp = 0; uint32_t in_cycles = 0;
  for (uint32_t r = 0; r < n; r++) {
    if (S.idx[r] != -1) continue;
    S.work.push_back(r); S.pos.push_back(S.adj_off[r]);
    while (!S.work.empty()) {
      const uint32_t v = S.work.back();
      if (S.pos.back() == S.adj_off[v] && S.idx[v] == -1) { S.idx[v] = S.low[v] = counter++; S.st.push_back(v); S.on[v] = 1; }
      bool descended = false;
      while (S.pos.back() < S.adj_off[v + 1]) {
        const Edge e = S.adj[S.pos.back()++];
        if (!(e.kind & mask)) continue;
        if (S.idx[e.to] == -1) { S.work.push_back(e.to); S.pos.push_back(S.adj_off[e.to]); descended = true; break; }
        if (S.on[e.to]) S.low[v] = std::min(S.low[v], S.idx[e.to]);
      }
      if (descended) continue;
      if (S.low[v] == S.idx[v]) {
        uint32_t size = 0, w;
        do { w = S.st.back(); S.st.pop_back(); S.on[w] = 0; S.comp[w] = ncomp; size++; } while (w != v);
        if (size > 1) in_cycles += size;
        ncomp++;
      }
      S.work.pop_back(); S.pos.pop_back();
      if (!S.work.empty()) S.low[S.work.back()] = std::min(S.low[S.work.back()], S.low[v]);
    }
  }
  return in_cycles;
}

// is `dst` reachable from `src` over edges of `mask`, staying inside the component of `src`?
bool reach(Scratch &S, uint32_t n, uint8_t mask, uint32_t src, uint32_t dst) {
  if (src == dst) return true;
  S.seen.assign(n, 0); S.q.clear(); S.q.push_back(src); S.seen[src] = 1;
  while (!S.q.empty()) {
    const uint32_t v = S.q.back(); S.q.pop_back();
    for (uint32_t k = S.adj_off[v]; k < S.adj_off[v + 1]; k++) {
      const Edge &e = S.adj[k];
      if (!(e.kind & mask) || S.comp[e.to] != S.comp[src] || S.seen[e.to]) continue;
      if (e.to == dst) return true;
      S.seen[e.to] = 1; S.q.push_back(e.to);
    }
  }
  return false;
}

// classify the cycles of the dependency graph (+ realtime edges when rt): returns anomaly bits
uint32_t classify(Scratch &S, uint32_t n, bool rt, uint32_t *in_cycles) {
  const uint8_t x = rt ? E_RT : 0;
  uint32_t bits = 0;
  if (scc(S, n, E_WW | x)) bits |= MSIM_ANOMALY_G0;
  if (scc(S, n, E_WW | E_WR | x) && !bits) bits |= MSIM_ANOMALY_G1C;
  const uint32_t cyc = scc(S, n, E_WW | E_WR | E_RW | x);
  if (in_cycles) *in_cycles = cyc;
  if (cyc && !bits) {
    // a cycle with exactly one anti-dependency: some rw edge u->v inside a component with v ~> u over ww/wr(/rt)
    bool single = false;
    for (const Edge &e : S.adj)
      if ((e.kind & E_RW) && S.comp[e.from] == S.comp[e.to] && e.from != e.to && reach(S, n, E_WW | E_WR | x, e.to, e.from)) { single = true; break; }
    bits |= single ? MSIM_ANOMALY_G_SINGLE : MSIM_ANOMALY_G2;
  }
  return bits;
}

// pairs invocations with completions: S.txns / S.mops (the completed form of :ok transactions) and the transitively
// reduced realtime order; returns anomaly bits (a row whose payload lies outside the area)
uint32_t collect(Scratch &S, const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, bool rw, msim_check_result *out) {
  std::memset(out, 0, sizeof *out);
  S.txns.clear(); S.mops.clear(); S.bytes.clear(); S.rt.clear(); S.frontier.clear(); S.open.clear(); S.edges.clear();
  uint32_t anomalies = 0;

  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    const uint32_t proc = MSIM_OP_PROCESS(r), type = MSIM_OP_TYPE(r);
    if (proc == MSIM_PROCESS_NEMESIS || MSIM_OP_F(r) != MSIM_F_TXN) continue;
    const uint32_t off = r.value, len = MSIM_OP_LEN(r);
    if ((uint64_t)off + len > n_words) { anomalies |= MSIM_ANOMALY_INTERNAL; continue; }
    if (proc >= S.open.size()) S.open.resize(proc + 64, -1);
    if (type == MSIM_T_INVOKE) {
      out->op_count++;
      Txn t; t.process = proc; t.inv = (int)i; t.cmp = -1; t.type = MSIM_T_INFO;  // never completed = indeterminate
      t.mop0 = (uint32_t)S.mops.size();
      parse_txn(S, payload + off, len, rw);
      t.n_mops = (uint32_t)S.mops.size() - t.mop0;
      t.rt0 = (uint32_t)S.rt.size(); t.n_rt = (uint32_t)S.frontier.size();   // realtime predecessors = the frontier now
      S.rt.insert(S.rt.end(), S.frontier.begin(), S.frontier.end());
      S.open[proc] = (int)S.txns.size();
      S.txns.push_back(t);
    } else {
      const int id = S.open[proc];
      if (id < 0) continue;
      S.open[proc] = -1;
      Txn &t = S.txns[(size_t)id];
      t.cmp = (int)i; t.type = (uint8_t)type;
      if (type == MSIM_T_OK) {
        out->ok_count++;
        t.mop0 = (uint32_t)S.mops.size();
        parse_txn(S, payload + off, len, rw);   // the completed form replaces the requested one (the old mops stay unused)
        t.n_mops = (uint32_t)S.mops.size() - t.mop0;
        // frontier := (frontier - predecessors of t) + t  (transitive reduction of the realtime order)
        S.nf.clear();
        const uint32_t *pb = S.rt.data() + t.rt0, *pe = pb + t.n_rt;
        for (uint32_t f : S.frontier) if (std::find(pb, pe, f) == pe) S.nf.push_back(f);
        S.nf.push_back((uint32_t)id);
        S.frontier.swap(S.nf);
      } else if (type == MSIM_T_FAIL) out->fail_count++;
      else out->info_count++;
    }
  }
  out->attempt_count = (uint32_t)S.txns.size(); out->stable_count = out->ok_count;
  return anomalies;
}

uint32_t judge(uint32_t anomalies, uint32_t cm) {
  const uint32_t cycles = MSIM_ANOMALY_G0 | MSIM_ANOMALY_G1C | MSIM_ANOMALY_G_SINGLE | MSIM_ANOMALY_G2;
  if ((anomalies & MSIM_ANOMALY_REALTIME) && cm != MSIM_CM_STRICT_SERIALIZABLE) anomalies &= ~(cycles | MSIM_ANOMALY_REALTIME);  // only -realtime cycles were found
  return anomalies & msim_proscribed_anomalies(cm);
}

// realtime edges, adjacency, cycle classification, verdict
void finish(Scratch &S, uint32_t anomalies, uint32_t flags, uint32_t cm, msim_check_result *out) {
  const uint32_t n = (uint32_t)S.txns.size();
  auto add = [&](uint32_t a, uint32_t b2, uint8_t kind) { if (a != b2) S.edges.push_back(Edge{a, b2, kind}); };
  for (uint32_t t = 0; t < n; t++) if (S.txns[t].type != MSIM_T_FAIL) for (uint32_t k = 0; k < S.txns[t].n_rt; k++) add(S.rt[S.txns[t].rt0 + k], t, E_RT);

  // CSR adjacency
  S.adj_off.assign(n + 1, 0);
  for (const Edge &e : S.edges) S.adj_off[e.from + 1]++;
  for (uint32_t v = 0; v < n; v++) S.adj_off[v + 1] += S.adj_off[v];
  S.adj.resize(S.edges.size());
  S.pos.assign(S.adj_off.begin(), S.adj_off.end() - 1);
  for (const Edge &e : S.edges) S.adj[S.pos[e.from]++] = e;

  uint32_t cyc = scc(S, n, E_WW | E_WR | E_RW | E_RT);  // acyclic over every edge kind (the common case): nothing to classify
  if (cyc) {
    uint32_t dep = classify(S, n, false, &cyc);
    if (!dep) { const uint32_t with_rt = classify(S, n, true, &cyc); if (with_rt) dep = with_rt | MSIM_ANOMALY_REALTIME; }
    anomalies |= dep;
  }

  out->lost_count = (uint32_t)S.edges.size(); out->stale_count = cyc; out->error_count = anomalies;
  out->valid = flags ? 0u : judge(anomalies, cm) ? 0u : (out->ok_count == 0 ? 2u : 1u);
}

// ---- list-append -------------------------------------------------------------------------------------------------------
void check_history(Scratch &S, const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, uint32_t cm, msim_check_result *out) {
  uint32_t anomalies = collect(S, rows, n_rows, payload, n_words, false, out), max_key = 0;
  const uint32_t n = (uint32_t)S.txns.size();
  uint32_t max_val = 0;
  for (const Txn &t : S.txns) for (uint32_t k = 0; k < t.n_mops; k++) { const Mop &m = S.mops[t.mop0 + k]; max_key = std::max<uint32_t>(max_key, m.key); if (m.f) max_val = std::max<uint32_t>(max_val, m.val); }
  for (uint8_t b : S.bytes) max_val = std::max<uint32_t>(max_val, b);   // elements that were read but never appended (garbage reads)

  // writers: (key, element) -> transaction; the table is as wide as the largest element (16 by default), not 256
  const uint32_t stride = max_val + 1;
  auto kv = [stride](uint32_t k, uint32_t v) { return k * stride + v; };
  S.writer.assign((size_t)(max_key + 1) * stride, -1);
  for (uint32_t t = 0; t < n; t++)
    for (uint32_t k = 0; k < S.txns[t].n_mops; k++) {
      const Mop &m = S.mops[S.txns[t].mop0 + k];
      if (!m.f) continue;
      if (S.writer[kv(m.key, m.val)] >= 0) anomalies |= MSIM_ANOMALY_DUPLICATE_ELEMENTS;  // the generator never repeats (k, v)
      S.writer[kv(m.key, m.val)] = (int)t;
    }
  // last element transaction t appended to key (for G1b)
  auto final_of = [&](uint32_t t, uint32_t key) -> int {
    const Txn &x = S.txns[t]; int v = -1;
    for (uint32_t k = 0; k < x.n_mops; k++) { const Mop &m = S.mops[x.mop0 + k]; if (m.f && m.key == key) v = m.val; }
    return v;
  };
  auto is_own = [&](uint32_t t, uint32_t key, uint8_t el) { const int w = S.writer[kv(key, el)]; return w >= 0 && (uint32_t)w == t; };

  // per-transaction checks + longest read per key
  S.longest.assign(max_key + 1, -1);
  for (uint32_t t = 0; t < n; t++) {
    const Txn &x = S.txns[t];
    if (x.type != MSIM_T_OK) continue;
    for (uint32_t k = 0; k < x.n_mops; k++) {
      const Mop &m = S.mops[x.mop0 + k];
      if (m.f) continue;
      const uint8_t *l = S.bytes.data() + m.off; const uint32_t len = m.len;
      // duplicates (lists are <= 63 long)
      for (uint32_t a = 0; a < len; a++) for (uint32_t b2 = a + 1; b2 < len; b2++) if (l[a] == l[b2]) anomalies |= MSIM_ANOMALY_DUPLICATE_ELEMENTS;
      // internal consistency: what the transaction's own earlier micro-ops imply for this read
      {
        int prev = -1;  // the latest earlier read of this key
        for (uint32_t e = 0; e < k; e++) { const Mop &p = S.mops[x.mop0 + e]; if (!p.f && p.key == m.key) prev = (int)e; }
        const uint32_t e0 = prev < 0 ? 0 : (uint32_t)prev + 1;
        uint32_t n_app = 0;   // the transaction's own appends to this key since that read (any number: compared in place)
        for (uint32_t e = e0; e < k; e++) { const Mop &p = S.mops[x.mop0 + e]; if (p.f && p.key == m.key) n_app++; }
        bool ok = true;
        uint32_t at = 0;      // where the own appends must sit in this read
        if (prev >= 0) {  // must equal the earlier read followed by the appends since
          const Mop &p = S.mops[x.mop0 + (uint32_t)prev];
          ok = len == (uint32_t)p.len + n_app && std::equal(l, l + p.len, S.bytes.data() + p.off);
          at = p.len;
        } else { ok = len >= n_app; at = len - n_app; }  // must end with the own appends
        if (ok) for (uint32_t e = e0, a = 0; e < k; e++) { const Mop &p = S.mops[x.mop0 + e]; if (p.f && p.key == m.key) { if (l[at + a] != p.val) ok = false; a++; } }
        if (!ok) anomalies |= MSIM_ANOMALY_INTERNAL;
      }
      // the externally visible part of the read: without the transaction's own appends at the tail
      uint32_t ext = len;
      while (ext > 0 && is_own(t, m.key, l[ext - 1])) ext--;
      for (uint32_t e = 0; e < ext; e++) {
        const int w = S.writer[kv(m.key, l[e])];
        if (w < 0) { anomalies |= MSIM_ANOMALY_G1A; continue; }                         // an element nobody appended (garbage read)
        if (S.txns[(size_t)w].type == MSIM_T_FAIL) anomalies |= MSIM_ANOMALY_G1A;       // aborted read
      }
      if (ext > 0) {
        const int w = S.writer[kv(m.key, l[ext - 1])];
        if (w >= 0 && (uint32_t)w != t && final_of((uint32_t)w, m.key) != (int)l[ext - 1]) anomalies |= MSIM_ANOMALY_G1B;
      }
      const int lg = S.longest[m.key];
      if (lg < 0 || S.mops[(size_t)lg].len < len) S.longest[m.key] = (int)(x.mop0 + k);
    }
  }

  // version orders, prefix property, dependency edges
  auto add = [&](uint32_t a, uint32_t b2, uint8_t kind) { if (a != b2) S.edges.push_back(Edge{a, b2, kind}); };
  for (uint32_t key = 0; key <= max_key; key++) {
    if (S.longest[key] < 0) continue;
    const Mop &lm = S.mops[(size_t)S.longest[key]];
    const uint8_t *ord = S.bytes.data() + lm.off;
    for (uint32_t i = 0; i + 1 < lm.len; i++) {
      const int a = S.writer[kv(key, ord[i])], b2 = S.writer[kv(key, ord[i + 1])];
      if (a < 0 || b2 < 0) continue;
      const bool fa = S.txns[(size_t)a].type == MSIM_T_FAIL, fb = S.txns[(size_t)b2].type == MSIM_T_FAIL;
      if (fa && !fb) anomalies |= MSIM_ANOMALY_DIRTY_UPDATE;
      if (!fa && !fb) add((uint32_t)a, (uint32_t)b2, E_WW);
    }
  }
  for (uint32_t t = 0; t < n; t++) {
    const Txn &x = S.txns[t];
    if (x.type != MSIM_T_OK) continue;
    for (uint32_t k = 0; k < x.n_mops; k++) {
      const Mop &m = S.mops[x.mop0 + k];
      if (m.f || S.longest[m.key] < 0) continue;
      const Mop &lm = S.mops[(size_t)S.longest[m.key]];
      const uint8_t *ord = S.bytes.data() + lm.off, *l = S.bytes.data() + m.off;
      if (m.len > lm.len || !std::equal(l, l + m.len, ord)) { anomalies |= MSIM_ANOMALY_INCOMPATIBLE_ORDER; continue; }
      uint32_t ext = m.len;
      while (ext > 0 && is_own(t, m.key, l[ext - 1])) ext--;
      if (ext > 0) { const int w = S.writer[kv(m.key, l[ext - 1])]; if (w >= 0 && S.txns[(size_t)w].type != MSIM_T_FAIL) add((uint32_t)w, t, E_WR); }
      // anti-dependency: the next version after the one read
      if (m.len < lm.len) { const int w = S.writer[kv(m.key, ord[m.len])]; if (w >= 0 && S.txns[(size_t)w].type != MSIM_T_FAIL) add(t, (uint32_t)w, E_RW); }
    }
  }
  finish(S, anomalies, flags, cm, out);
}

// ---- rw-register -------------------------------------------------------------------------------------------------------
// [upstream] elle.rw-register as jepsen.tests.cycle.wr wires it for workload/txn_rw_register.clj:162-166 (:wfr-keys? true):
//   * writes are unique per key, so a read of v names its writer: wr writer(v) -> reader;
//   * version order of a key, from two sources only: the initial nil precedes every other version, and a transaction
//     that reads v1 of a key and then writes v2 puts v1 before v2 (writes follow reads); a key whose version graph has a
//     cycle is reported (cyclic-versions) and contributes no dependencies;
//   * for every v1 -> v2 of that graph: ww writer(v1) -> writer(v2), rw each reader of v1 -> writer(v2);
//   * only a transaction's external reads (before its own first write of the key) and final writes count; internal
//     reads must agree with the transaction's own earlier micro-ops (internal); G1a / G1b as for list-append.
// Returns false (out->valid = 2, unknown) when a register value is >= 64: the version graphs below hold 64 versions per key
// (the engine's generator stops at max-writes-per-key <= 63; externally produced histories may not).
bool check_rw(Scratch &S, const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, uint32_t cm, msim_check_result *out) {
  uint32_t anomalies = collect(S, rows, n_rows, payload, n_words, true, out), max_key = 0;
  const uint32_t n = (uint32_t)S.txns.size();
  uint32_t max_val = 0;
  for (const Txn &t : S.txns) for (uint32_t k = 0; k < t.n_mops; k++) { const Mop &m = S.mops[t.mop0 + k]; max_key = std::max<uint32_t>(max_key, m.key); max_val = std::max<uint32_t>(max_val, m.val); }
  if (max_val >= 64) { out->valid = 2; return false; }
  constexpr uint32_t stride = 64u;  // the version graphs below walk versions 0..63 of every key
  auto kv = [](uint32_t k, uint32_t v) { return k * stride + v; };
  S.writer.assign((size_t)(max_key + 1) * stride, -1);
  for (uint32_t t = 0; t < n; t++)
    for (uint32_t k = 0; k < S.txns[t].n_mops; k++) {
      const Mop &m = S.mops[S.txns[t].mop0 + k];
      if (!m.f) continue;
      if (S.writer[kv(m.key, m.val)] >= 0) anomalies |= MSIM_ANOMALY_DUPLICATE_ELEMENTS;  // the generator never repeats (k, v)
      S.writer[kv(m.key, m.val)] = (int)t;
    }
  auto final_of = [&](uint32_t t, uint32_t key) -> int {
    const Txn &x = S.txns[t]; int v = -1;
    for (uint32_t k = 0; k < x.n_mops; k++) { const Mop &m = S.mops[x.mop0 + k]; if (m.f && m.key == key) v = m.val; }
    return v;
  };
  // external read of `key` by :ok transaction t: -1 none, 0 nil, else the value (values are >= 1)
  auto ext_read = [&](uint32_t t, uint32_t key) -> int {
    const Txn &x = S.txns[t];
    for (uint32_t k = 0; k < x.n_mops; k++) { const Mop &m = S.mops[x.mop0 + k]; if (m.key == key) return m.f ? -1 : m.nil ? 0 : (int)m.val; }
    return -1;
  };
  auto add = [&](uint32_t a, uint32_t b2, uint8_t kind) { if (a != b2) S.edges.push_back(Edge{a, b2, kind}); };

  // succ[key][v1] = bit set of versions v2 with v1 -> v2 (version 0 = nil, values 1..63)
  S.vsucc.assign((size_t)(max_key + 1) * 64, 0);
  S.vseen.assign(max_key + 1, 0);
  for (uint32_t t = 0; t < n; t++) {
    const Txn &x = S.txns[t];
    for (uint32_t k = 0; k < x.n_mops; k++) { const Mop &m = S.mops[x.mop0 + k]; if (m.f && x.type != MSIM_T_FAIL) S.vseen[m.key] |= 1ull << (m.val & 63); }
    if (x.type != MSIM_T_OK) continue;
    for (uint32_t k = 0; k < x.n_mops; k++) {
      const Mop &m = S.mops[x.mop0 + k];
      if (m.f) continue;
      // internal consistency: the latest earlier micro-op on this key decides what the read must return
      int prev = -1;
      for (uint32_t e = 0; e < k; e++) if (S.mops[x.mop0 + e].key == m.key) prev = (int)e;
      if (prev >= 0) {
        const Mop &p = S.mops[x.mop0 + (uint32_t)prev];
        const bool same = p.f ? (!m.nil && m.val == p.val) : (m.nil == p.nil && (m.nil || m.val == p.val));
        if (!same) anomalies |= MSIM_ANOMALY_INTERNAL;
        continue;
      }
      if (m.nil) continue;
      S.vseen[m.key] |= 1ull << (m.val & 63);
      const int w = S.writer[kv(m.key, m.val)];
      if (w < 0 || S.txns[(size_t)w].type == MSIM_T_FAIL) { anomalies |= MSIM_ANOMALY_G1A; continue; }  // garbage / aborted read
      if ((uint32_t)w != t) {
        if (final_of((uint32_t)w, m.key) != (int)m.val) anomalies |= MSIM_ANOMALY_G1B;
        add((uint32_t)w, t, E_WR);
      }
      // writes follow reads
      const int fw = final_of(t, m.key);
      if (fw >= 0 && fw != (int)m.val) S.vsucc[(size_t)m.key * 64 + m.val] |= 1ull << (fw & 63);
    }
  }
  for (uint32_t key = 0; key <= max_key; key++) {
    uint64_t *succ = S.vsucc.data() + (size_t)key * 64;
    succ[0] |= S.vseen[key] & ~1ull;  // nil precedes every version
    // cyclic version order?  reach[v] = transitive closure over <= 64 versions
    uint64_t reach[64];
    for (uint32_t v = 0; v < 64; v++) reach[v] = succ[v];
    for (bool grown = true; grown;) {
      grown = false;
      for (uint32_t v = 0; v < 64; v++) {
        uint64_t r = reach[v], add_ = 0;
        for (uint64_t b = r; b; b &= b - 1) add_ |= reach[__builtin_ctzll(b)];
        if (add_ & ~r) { reach[v] = r | add_; grown = true; }
      }
    }
    bool cyclic = false;
    for (uint32_t v = 0; v < 64; v++) if ((reach[v] >> v) & 1) cyclic = true;
    if (cyclic) { anomalies |= MSIM_ANOMALY_CYCLIC_VERSIONS; for (uint32_t v = 0; v < 64; v++) succ[v] = 0; continue; }
    for (uint32_t v1 = 1; v1 < 64; v1++) {
      const int a = S.writer[kv(key, v1)];
      if (a < 0 || S.txns[(size_t)a].type == MSIM_T_FAIL) continue;
      for (uint64_t b = succ[v1]; b; b &= b - 1) {
        const int b2 = S.writer[kv(key, (uint32_t)__builtin_ctzll(b))];
        if (b2 >= 0 && S.txns[(size_t)b2].type != MSIM_T_FAIL) add((uint32_t)a, (uint32_t)b2, E_WW);
      }
    }
  }
  for (uint32_t t = 0; t < n; t++) {
    const Txn &x = S.txns[t];
    if (x.type != MSIM_T_OK) continue;
    for (uint32_t k = 0; k < x.n_mops; k++) {
      const Mop &m = S.mops[x.mop0 + k];
      if (m.f || ext_read(t, m.key) < 0) continue;
      bool first = true;
      for (uint32_t e = 0; e < k; e++) if (S.mops[x.mop0 + e].key == m.key) first = false;
      if (!first) continue;
      const uint32_t v1 = m.nil ? 0u : m.val;
      for (uint64_t b = S.vsucc[(size_t)m.key * 64 + v1]; b; b &= b - 1) {
        const int w = S.writer[kv(m.key, (uint32_t)__builtin_ctzll(b))];
        if (w >= 0 && S.txns[(size_t)w].type != MSIM_T_FAIL) add(t, (uint32_t)w, E_RW);
      }
    }
  }
  finish(S, anomalies, flags, cm, out);
  return true;
}

}  // namespace

// the host analysis of one list-append history (txn_check_dev.hip hands over what it cannot prove clean)
void msim_txn_check_instance_host(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, uint32_t cm, msim_check_result *res) {
  thread_local Scratch S;
  check_history(S, rows, n_rows, payload, n_words, flags, cm, res);
}

// the host analysis of one rw-register history (rw_check_dev.hip hands over what it cannot prove valid)
void msim_rw_check_instance_host(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, uint32_t cm, msim_check_result *res) {
  thread_local Scratch S;
  (void)check_rw(S, rows, n_rows, payload, n_words, flags, cm, res);
}

extern "C" int msim_check_txn_rows(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, msim_check_result *out) {
  if (!rows || !out || (!payload && n_words)) return MSIM_E_INVALID;
  Scratch S;
  check_history(S, rows, n_rows, payload, n_words, 0, MSIM_CM_STRICT_SERIALIZABLE, out);
  return MSIM_OK;
}

extern "C" uint32_t msim_proscribed_anomalies(uint32_t cm) {
  uint32_t p = MSIM_ANOMALY_DUPLICATE_ELEMENTS | MSIM_ANOMALY_INCOMPATIBLE_ORDER | MSIM_ANOMALY_DIRTY_UPDATE | MSIM_ANOMALY_CYCLIC_VERSIONS | MSIM_ANOMALY_G0;
  if (cm <= MSIM_CM_READ_COMMITTED) p |= MSIM_ANOMALY_G1A | MSIM_ANOMALY_G1B | MSIM_ANOMALY_G1C;
  if (cm <= MSIM_CM_SNAPSHOT_ISOLATION) p |= MSIM_ANOMALY_G_SINGLE | MSIM_ANOMALY_INTERNAL;
  if (cm <= MSIM_CM_SERIALIZABLE) p |= MSIM_ANOMALY_G2;
  if (cm == MSIM_CM_STRICT_SERIALIZABLE) p |= MSIM_ANOMALY_REALTIME;
  return p;
}

extern "C" uint32_t msim_violated_anomalies(uint32_t anomalies, uint32_t cm) { return cm > MSIM_CM_READ_UNCOMMITTED ? anomalies : judge(anomalies, cm); }

extern "C" int msim_check_rw_rows(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t consistency_model,
                                  msim_check_result *out) {
  if (!rows || !out || (!payload && n_words) || consistency_model > MSIM_CM_READ_UNCOMMITTED) return MSIM_E_INVALID;
  Scratch S;
  return check_rw(S, rows, n_rows, payload, n_words, 0, consistency_model, out) ? MSIM_OK : MSIM_E_INVALID;
}

// Checker threads keep their working storage across calls: a fresh Scratch per call would fault in (and give back) a few MB
// per thread and history batch, and 256 threads doing that serialise in the kernel.
static std::vector<Scratch *> &scratch_pool(unsigned nt) {
  static std::vector<Scratch *> pool;  // one engine context checks at a time per process in practice; guarded by the caller's ctx use
  while (pool.size() < nt) pool.push_back(new Scratch());
  return pool;
}
static std::mutex g_check_mutex;

int msim_check_txn_host(msim_ctx *ctx) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc = msim_fetch(ctx);
  if (rc != MSIM_OK) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  const uint32_t n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  unsigned nt = msim_host_threads();
  if (nt > n) nt = n;
  {
    std::lock_guard<std::mutex> lock(g_check_mutex);
    std::vector<Scratch *> &pool = scratch_pool(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
      th.emplace_back([ctx, n, nt, t, &pool]() {
        Scratch &S = *pool[t];
        const bool rw = ctx->cfg.workload == MSIM_WL_TXN_RW_REGISTER;
        for (uint32_t i = t; i < n; i += nt)
          if (rw) (void)check_rw(S, ctx->h_rows + ctx->h_row_off[i], ctx->h_meta[i].n_rows, ctx->h_payload + ctx->h_pay_off[i],
                                 ctx->h_meta[i].n_payload_words, ctx->h_meta[i].flags, ctx->cfg.consistency_model, &ctx->h_check[i]);
          else check_history(S, ctx->h_rows + ctx->h_row_off[i], ctx->h_meta[i].n_rows, ctx->h_payload + ctx->h_pay_off[i],
                             ctx->h_meta[i].n_payload_words, ctx->h_meta[i].flags, ctx->cfg.consistency_model, &ctx->h_check[i]);
      });
    for (auto &x : th) x.join();
  }
  const auto t2 = std::chrono::steady_clock::now();
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->d_check, ctx->h_check, (size_t)n * sizeof(msim_check_result), hipMemcpyHostToDevice));
  ctx->checked = true; ctx->check_fetched = true;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  static const char *timing = std::getenv("MSIM_TIMING");  // developer knob: where the host check spends its time
  if (timing) std::fprintf(stderr, "[msim] txn check: fetch %.1f ms, %u threads x graph work %.1f ms, total %.1f ms\n",
                           std::chrono::duration<float, std::milli>(t1 - t0).count(), nt, std::chrono::duration<float, std::milli>(t2 - t1).count(), ctx->check_ms);
  return MSIM_OK;
}
