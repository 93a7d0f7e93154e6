// txn_check.cpp — host-side list-append transaction checker for the txn-list-append workload.
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
#include <cstring>
#include <thread>
#include <unordered_map>
#include <vector>

#include "engine_internal.h"

namespace {

struct Mop { uint8_t f; uint16_t key; uint8_t val; bool nil; std::vector<uint8_t> list; };
struct Txn {
  uint32_t process; int inv, cmp; uint8_t type;  // MSIM_T_OK / FAIL / INFO
  std::vector<Mop> mops;
};
enum { E_WW = 1, E_WR = 2, E_RW = 4, E_RT = 8 };
struct Edge { uint32_t to; uint8_t kind; };

void parse_txn(const uint32_t *w, uint32_t n, std::vector<Mop> &out) {
  for (uint32_t i = 0; i < n;) {
    const uint32_t h = w[i++];
    Mop m; m.f = h & 1; m.key = (h >> 1) & 0x7FFF; m.val = 0; m.nil = false;
    const uint32_t x = (h >> 16) & 0xFF;
    if (m.f) m.val = (uint8_t)x;
    else if (x == 0xFF) m.nil = true;
    else { for (uint32_t e = 0; e < x && i + e / 4 < n; e++) m.list.push_back((uint8_t)(w[i + e / 4] >> (8 * (e % 4)))); i += (x + 3) / 4; }
    out.push_back(std::move(m));
  }
}

// strongly connected components (iterative Tarjan) over the edges whose kind is in `mask`; comp[v] = component id,
// returns the number of vertices that sit in a component of size > 1
uint32_t scc(const std::vector<std::vector<Edge>> &g, uint8_t mask, std::vector<int> &comp) {
  const uint32_t n = (uint32_t)g.size();
  std::vector<int> idx(n, -1), low(n, 0); std::vector<char> on(n, 0);
  std::vector<uint32_t> st, work, pos;
  comp.assign(n, -1);
  int counter = 0, ncomp = 0; uint32_t in_cycles = 0;
  for (uint32_t r = 0; r < n; r++) {
    if (idx[r] != -1) continue;
    work.push_back(r); pos.push_back(0);
    while (!work.empty()) {
      const uint32_t v = work.back();
      if (pos.back() == 0) { idx[v] = low[v] = counter++; st.push_back(v); on[v] = 1; }
      bool descended = false;
      while (pos.back() < g[v].size()) {
        const Edge e = g[v][pos.back()++];
        if (!(e.kind & mask)) continue;
        if (idx[e.to] == -1) { work.push_back(e.to); pos.push_back(0); descended = true; break; }
        if (on[e.to]) low[v] = std::min(low[v], idx[e.to]);
      }
      if (descended) continue;
      if (low[v] == idx[v]) {
        uint32_t size = 0, w;
        do { w = st.back(); st.pop_back(); on[w] = 0; comp[w] = ncomp; size++; } while (w != v);
        if (size > 1) in_cycles += size;
        ncomp++;
      }
      work.pop_back(); pos.pop_back();
      if (!work.empty()) low[work.back()] = std::min(low[work.back()], low[v]);
    }
  }
  return in_cycles;
}

// is `dst` reachable from `src` over edges of `mask`, staying inside component `c` of `comp`?
bool reach(const std::vector<std::vector<Edge>> &g, uint8_t mask, const std::vector<int> &comp, uint32_t src, uint32_t dst) {
  if (src == dst) return true;
  std::vector<char> seen(g.size(), 0); std::vector<uint32_t> q{src}; seen[src] = 1;
  while (!q.empty()) {
    const uint32_t v = q.back(); q.pop_back();
    for (const Edge &e : g[v]) {
      if (!(e.kind & mask) || comp[e.to] != comp[src] || seen[e.to]) continue;
      if (e.to == dst) return true;
      seen[e.to] = 1; q.push_back(e.to);
    }
  }
  return false;
}

// classify the cycles of the graph restricted to `dep_mask` (+ E_RT when rt): returns anomaly bits
uint32_t classify(const std::vector<std::vector<Edge>> &g, bool rt, uint32_t *in_cycles) {
  const uint8_t x = rt ? E_RT : 0;
  std::vector<int> comp;
  uint32_t bits = 0;
  if (scc(g, E_WW | x, comp)) bits |= MSIM_ANOMALY_G0;
  if (scc(g, E_WW | E_WR | x, comp) && !bits) bits |= MSIM_ANOMALY_G1C;
  const uint32_t cyc = scc(g, E_WW | E_WR | E_RW | x, comp);
  if (in_cycles) *in_cycles = cyc;
  if (cyc && !bits) {
    // a cycle with exactly one anti-dependency: some rw edge u->v inside a component with v ~> u over ww/wr(/rt)
    bool single = false;
    for (uint32_t u = 0; u < g.size() && !single; u++)
      for (const Edge &e : g[u])
        if ((e.kind & E_RW) && comp[u] == comp[e.to] && u != e.to && reach(g, E_WW | E_WR | x, comp, e.to, u)) { single = true; break; }
    bits |= single ? MSIM_ANOMALY_G_SINGLE : MSIM_ANOMALY_G2;
  }
  return bits;
}

void check_history(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, msim_check_result *out) {
  std::memset(out, 0, sizeof *out);
  std::vector<Txn> txns;
  std::unordered_map<uint32_t, uint32_t> open;  // process -> txn index
  std::vector<std::vector<uint32_t>> rt_in;     // realtime predecessors (the frontier at invocation)
  std::vector<uint32_t> frontier;
  uint32_t anomalies = 0;

  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    const uint32_t proc = MSIM_OP_PROCESS(r), type = MSIM_OP_TYPE(r);
    if (proc == MSIM_PROCESS_NEMESIS || MSIM_OP_F(r) != MSIM_F_TXN) continue;
    const uint32_t off = r.value, len = MSIM_OP_LEN(r);
    if ((uint64_t)off + len > n_words) { anomalies |= MSIM_ANOMALY_INTERNAL; continue; }
    if (type == MSIM_T_INVOKE) {
      out->op_count++;
      Txn t; t.process = proc; t.inv = (int)i; t.cmp = -1; t.type = MSIM_T_INFO;  // never completed = indeterminate
      parse_txn(payload + off, len, t.mops);
      open[proc] = (uint32_t)txns.size();
      txns.push_back(std::move(t));
      rt_in.push_back(frontier);
    } else {
      auto it = open.find(proc);
      if (it == open.end()) continue;
      const uint32_t id = it->second; open.erase(it);
      Txn &t = txns[id];
      t.cmp = (int)i; t.type = (uint8_t)type;
      if (type == MSIM_T_OK) {
        out->ok_count++;
        t.mops.clear(); parse_txn(payload + off, len, t.mops);
        // frontier := (frontier - predecessors of t) + t  (transitive reduction of the realtime order)
        std::vector<uint32_t> nf;
        for (uint32_t f : frontier) if (std::find(rt_in[id].begin(), rt_in[id].end(), f) == rt_in[id].end()) nf.push_back(f);
        nf.push_back(id);
        frontier.swap(nf);
      } else if (type == MSIM_T_FAIL) out->fail_count++;
      else out->info_count++;
    }
  }
  const uint32_t n = (uint32_t)txns.size();
  out->attempt_count = n; out->stable_count = out->ok_count;

  // writers: (key, element) -> transaction; last append of each transaction per key (for G1b)
  auto kv = [](uint32_t k, uint32_t v) { return (k << 8) | v; };
  std::unordered_map<uint32_t, uint32_t> writer, final_of;  // final_of[(txn<<15 | key)] = last element that txn appended to key
  for (uint32_t t = 0; t < n; t++)
    for (const Mop &m : txns[t].mops) if (m.f) {
      if (writer.count(kv(m.key, m.val))) anomalies |= MSIM_ANOMALY_DUPLICATE_ELEMENTS;  // the generator never repeats (k, v)
      writer[kv(m.key, m.val)] = t;
      final_of[(t << 15) | m.key] = m.val;
    }

  // per-transaction checks + longest read per key
  std::unordered_map<uint32_t, const std::vector<uint8_t> *> longest;
  for (uint32_t t = 0; t < n; t++) {
    const Txn &x = txns[t];
    if (x.type != MSIM_T_OK) continue;
    std::unordered_map<uint32_t, std::vector<uint8_t>> known;   // what this txn must see for a key from its own earlier mops
    std::unordered_map<uint32_t, std::vector<uint8_t>> own;     // own appends so far (key never read yet)
    for (const Mop &m : x.mops) {
      if (m.f) {
        auto k = known.find(m.key);
        if (k != known.end()) k->second.push_back(m.val); else own[m.key].push_back(m.val);
        continue;
      }
      const std::vector<uint8_t> &l = m.list;
      // duplicates
      { std::vector<uint8_t> s = l; std::sort(s.begin(), s.end()); if (std::adjacent_find(s.begin(), s.end()) != s.end()) anomalies |= MSIM_ANOMALY_DUPLICATE_ELEMENTS; }
      // internal consistency
      auto k = known.find(m.key);
      if (k != known.end()) { if (k->second != l) anomalies |= MSIM_ANOMALY_INTERNAL; }
      else {
        const std::vector<uint8_t> &o = own[m.key];
        if (o.size() > l.size() || !std::equal(o.begin(), o.end(), l.end() - (long)o.size())) anomalies |= MSIM_ANOMALY_INTERNAL;
      }
      known[m.key] = l;
      // the externally visible part of the read: without the transaction's own appends at the tail
      size_t ext = l.size();
      while (ext > 0) { auto w = writer.find(kv(m.key, l[ext - 1])); if (w != writer.end() && w->second == t) ext--; else break; }
      for (size_t e = 0; e < ext; e++) {
        auto w = writer.find(kv(m.key, l[e]));
        if (w == writer.end()) { anomalies |= MSIM_ANOMALY_G1A; continue; }         // an element nobody appended (garbage read)
        if (txns[w->second].type == MSIM_T_FAIL) anomalies |= MSIM_ANOMALY_G1A;     // aborted read
      }
      if (ext > 0) {
        auto w = writer.find(kv(m.key, l[ext - 1]));
        if (w != writer.end() && w->second != t && final_of[(w->second << 15) | m.key] != l[ext - 1]) anomalies |= MSIM_ANOMALY_G1B;
      }
      auto lg = longest.find(m.key);
      if (lg == longest.end() || lg->second->size() < l.size()) longest[m.key] = &l;
    }
  }

  // version orders, prefix property, dependency edges
  std::vector<std::vector<Edge>> g(n);
  uint32_t n_edges = 0;
  auto add = [&](uint32_t a, uint32_t b, uint8_t kind) { if (a != b) { g[a].push_back(Edge{b, kind}); n_edges++; } };
  std::unordered_map<uint32_t, uint32_t> pos;  // (key, element) -> index in the key's version order
  for (auto &kvp : longest) {
    const uint32_t key = kvp.first; const std::vector<uint8_t> &ord = *kvp.second;
    for (size_t i = 0; i < ord.size(); i++) pos[kv(key, ord[i])] = (uint32_t)i;
    for (size_t i = 0; i + 1 < ord.size(); i++) {
      auto a = writer.find(kv(key, ord[i])), b = writer.find(kv(key, ord[i + 1]));
      if (a == writer.end() || b == writer.end()) continue;
      if (txns[a->second].type == MSIM_T_FAIL && txns[b->second].type != MSIM_T_FAIL) anomalies |= MSIM_ANOMALY_DIRTY_UPDATE;
      if (txns[a->second].type != MSIM_T_FAIL && txns[b->second].type != MSIM_T_FAIL) add(a->second, b->second, E_WW);
    }
  }
  for (uint32_t t = 0; t < n; t++) {
    const Txn &x = txns[t];
    if (x.type != MSIM_T_OK) continue;
    for (const Mop &m : x.mops) {
      if (m.f) continue;
      auto lg = longest.find(m.key);
      if (lg == longest.end()) continue;
      const std::vector<uint8_t> &ord = *lg->second, &l = m.list;
      if (l.size() > ord.size() || !std::equal(l.begin(), l.end(), ord.begin())) { anomalies |= MSIM_ANOMALY_INCOMPATIBLE_ORDER; continue; }
      size_t ext = l.size();
      while (ext > 0) { auto w = writer.find(kv(m.key, l[ext - 1])); if (w != writer.end() && w->second == t) ext--; else break; }
      if (ext > 0) { auto w = writer.find(kv(m.key, l[ext - 1])); if (w != writer.end() && txns[w->second].type != MSIM_T_FAIL) add(w->second, t, E_WR); }
      // anti-dependency: the next version after the one read (skipping this transaction's own appends)
      size_t nx = l.size();
      if (nx < ord.size()) { auto w = writer.find(kv(m.key, ord[nx])); if (w != writer.end() && txns[w->second].type != MSIM_T_FAIL) add(t, w->second, E_RW); }
    }
  }
  for (uint32_t t = 0; t < n; t++) if (txns[t].type != MSIM_T_FAIL) for (uint32_t f : rt_in[t]) add(f, t, E_RT);

  uint32_t cyc = 0;
  uint32_t dep = classify(g, false, &cyc);
  if (!dep) { const uint32_t with_rt = classify(g, true, &cyc); if (with_rt) dep = with_rt | MSIM_ANOMALY_REALTIME; }
  anomalies |= dep;

  out->lost_count = n_edges; out->stale_count = cyc; out->error_count = anomalies;
  out->valid = flags ? 0u : anomalies ? 0u : (out->ok_count == 0 ? 2u : 1u);
}

}  // namespace

extern "C" int msim_check_txn_rows(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, msim_check_result *out) {
  if (!rows || !out || (!payload && n_words)) return MSIM_E_INVALID;
  check_history(rows, n_rows, payload, n_words, 0, out);
  return MSIM_OK;
}

int msim_check_txn_host(msim_ctx *ctx) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc = msim_fetch(ctx);
  if (rc != MSIM_OK) return rc;
  const uint32_t n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > n) nt = n;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([ctx, n, nt, t]() {
      for (uint32_t i = t; i < n; i += nt)
        check_history(ctx->h_rows + ctx->h_row_off[i], ctx->h_meta[i].n_rows, ctx->h_payload + ctx->h_pay_off[i],
                      ctx->h_meta[i].n_payload_words, ctx->h_meta[i].flags, &ctx->h_check[i]);
    });
  for (auto &x : th) x.join();
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->d_check, ctx->h_check, (size_t)n * sizeof(msim_check_result), hipMemcpyHostToDevice));
  ctx->checked = true; ctx->check_fetched = true;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return MSIM_OK;
}
