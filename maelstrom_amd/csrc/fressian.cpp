// fressian.cpp — host-side writer of the net journal in the reference's on-disk format: what maelstrom.net.journal appends to
// `store/<test>/net-journal/<stripe>.fressian` (net/journal.clj:55-141,220-239), so that the UNCHANGED maelstrom.net.checker
// (net/checker.clj:28-70) and net/viz.clj (:281-325) can read an engine run.  No device code.
//
// One Fressian object per journal event, written with journal.clj's own handlers:
//   Event    -> struct "ev" 4:  id (int), time (int, ns), type (:send / :recv, cached), message
//   Message  -> struct "msg" 4: id (int), src (string, cached), dest (string, cached), body
//   body     -> write-body! (journal.clj:55-69): tag "map", a closed list of key / value pairs; every key is cached, a value is
//               cached iff its key is :type.  Keys are keywords (process.clj:45 parses JSON bodies with keyword keys).
// Encoding = [upstream] org.fressian 0.6.x as clojure.data.fressian 1.0.0 drives it (not vendored, no JVM here: the byte layout
// below restates org.fressian.impl.Codes / FressianWriter from their published source and is PARITY UNPINNED against a JVM reader;
// tests/fressian_reader.py reads it back independently of this file's tables):
//   ints        -1..63 one byte; then 2..7-byte packed forms 0x50+(i>>8) / 0x68+(i>>16) / 0x72+(i>>24) / 0x76+(i>>32) / 0x7A+(i>>40) /
//               0x7E+(i>>48) followed by the low bytes big-endian; else 0xF8 + 8 bytes
//   strings     0xDA+len (len < 8) or 0xE3 + int len, then the bytes (ASCII here)
//   keywords    0xCA (tag "key"), namespace (nil = 0xF7), name (string); both components written with caching on
//   lists       0xE4+len (len < 8) or 0xEC + int len; closed list 0xED ... 0xFD; tag "map" = 0xC0
//   structs     first use 0xEF + tag string + int component count, later 0xA0+index (< 16) or 0xF0 + int index
//   cache       first use 0xCD + object, later 0x80+index (< 32) or 0xCC + int index; an object takes its index BEFORE its
//               components are written; nil, booleans, one-byte ints and "" are never cached
// Bodies are rebuilt from the 16-byte journal events (include/maelsim.h msim_event) and the instance's payload area:
// echo / broadcast / g-set / counter / unique-ids traffic, lin-kv reads / writes / cas and the client <-> node `txn` / `txn_ok`
// messages with the fields of doc/protocol.md and doc/workloads.md; messages whose contents the engine never materialises (Raft's
// request_vote / append_entries, replicate snapshots, the transactional nodes' traffic with their storage services, where a value is
// a version number) are written as {:type ..., :elided true, :a <the envelope's payload word>} plus ids — counts and Lamport
// diagrams stay right, the bodies say that they are abbreviated.
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/maelsim.h"
#include "engine_limits.h"

namespace {

struct FW {
  std::string out;
  std::unordered_map<std::string, uint32_t> cache;    // priority cache: key = kind byte + text
  std::unordered_map<std::string, uint32_t> structs;  // struct cache

  void raw(uint8_t b) { out.push_back((char)b); }
  void raw_be(uint64_t v, int bytes) { for (int i = bytes - 1; i >= 0; i--) raw((uint8_t)(v >> (8 * i))); }
  static bool one_byte(int64_t i) { return i >= -1 && i <= 63; }
  void integer(int64_t i) {
    if (one_byte(i)) { raw((uint8_t)i); return; }
    // packed form with k payload bytes holds a (8k + bits)-bit signed value; bits of the lead byte: 2:4+1, 3:3+1, 4..7:1+1
    if (i >= -(1ll << 12) && i < (1ll << 12)) { raw((uint8_t)(0x50 + (i >> 8))); raw_be((uint64_t)i, 1); return; }
    if (i >= -(1ll << 19) && i < (1ll << 19)) { raw((uint8_t)(0x68 + (i >> 16))); raw_be((uint64_t)i, 2); return; }
    if (i >= -(1ll << 25) && i < (1ll << 25)) { raw((uint8_t)(0x72 + (i >> 24))); raw_be((uint64_t)i, 3); return; }
    if (i >= -(1ll << 33) && i < (1ll << 33)) { raw((uint8_t)(0x76 + (i >> 32))); raw_be((uint64_t)i, 4); return; }
    if (i >= -(1ll << 41) && i < (1ll << 41)) { raw((uint8_t)(0x7A + (i >> 40))); raw_be((uint64_t)i, 5); return; }
    if (i >= -(1ll << 49) && i < (1ll << 49)) { raw((uint8_t)(0x7E + (i >> 48))); raw_be((uint64_t)i, 6); return; }
    raw(0xF8); raw_be((uint64_t)i, 8);
  }
  void str_plain(const std::string &s) {
    if (s.size() < 8) raw((uint8_t)(0xDA + s.size())); else { raw(0xE3); integer((int64_t)s.size()); }
    out += s;
  }
  // returns true if the object was emitted as a cache reference
  bool cached(const std::string &key) {
    auto it = cache.find(key);
    if (it == cache.end()) { const uint32_t idx = (uint32_t)cache.size(); cache.emplace(key, idx); raw(0xCD); return false; }
    if (it->second < 32) raw((uint8_t)(0x80 + it->second)); else { raw(0xCC); integer(it->second); }
    return true;
  }
  void str(const std::string &s, bool cache_it) {
    if (cache_it && !s.empty() && cached("s" + s)) return;
    str_plain(s);
  }
  void keyword(const std::string &name, bool cache_it) {
    if (cache_it && cached("k" + name)) return;
    raw(0xCA); raw(0xF7); str(name, true);   // tag "key": namespace nil, name (components are cached too)
  }
  void tag(const std::string &t, int components) {
    auto it = structs.find(t);
    if (it == structs.end()) { const uint32_t idx = (uint32_t)structs.size(); structs.emplace(t, idx); raw(0xEF); str_plain(t); integer(components); }
    else if (it->second < 16) raw((uint8_t)(0xA0 + it->second));
    else { raw(0xF0); integer(it->second); }
  }
  void list_header(size_t n) { if (n < 8) raw((uint8_t)(0xE4 + n)); else { raw(0xEC); integer((int64_t)n); } }
};

// endpoints behind the client slots are services (service.clj:290-296): the proxy's is the one it was pointed at, the single-root
// transactional node uses lin-kv, the multi-key one lin-kv then lww-kv
std::string endpoint(uint32_t e, uint32_t n_nodes, uint32_t slots, const msim_config *cfg) {
  char b[16];
  if (e < n_nodes) std::snprintf(b, sizeof b, "n%u", e);
  else if (e < n_nodes + slots) std::snprintf(b, sizeof b, "c%u", e - n_nodes);
  else {
    const char *name = "lin-kv";
    if (cfg->node_program == MSIM_NODE_LIN_KV_PROXY) name = cfg->proxy_service == MSIM_SVC_SEQ_KV ? "seq-kv" : cfg->proxy_service == MSIM_SVC_LWW_KV ? "lww-kv" : "lin-kv";
    else if ((cfg->node_program == MSIM_NODE_TXN_MULTI_KEY || cfg->node_program == MSIM_NODE_TXN_DATOMIC) && e == n_nodes + slots + 1) name = "lww-kv";
    else if (cfg->node_program == MSIM_NODE_TSO_IDS) name = "lin-tso";
    std::snprintf(b, sizeof b, "%s", name);
  }
  return b;
}

const char *const MSG_TYPES[] = {"", "init", "init_ok", "topology", "topology_ok", "echo", "echo_ok", "broadcast", "broadcast_ok", "read", "read_ok",
                                 "add", "add_ok", "replicate", "write", "write_ok", "cas", "cas_ok", "error", "request_vote", "request_vote_res",
                                 "append_entries", "append_entries_res", "txn", "txn_ok", "generate", "generate_ok", "replicate_ack", "ts", "ts_ok",
                                 "send", "send_ok", "poll", "poll_ok", "list_committed_offsets", "list_committed_offsets_ok", "commit_offsets", "commit_offsets_ok"};

bool is_reply(uint32_t t) {
  switch (t) {
    case MSIM_M_INIT_OK: case MSIM_M_TOPOLOGY_OK: case MSIM_M_ECHO_OK: case MSIM_M_BROADCAST_OK: case MSIM_M_READ_OK: case MSIM_M_ADD_OK:
    case MSIM_M_WRITE_OK: case MSIM_M_CAS_OK: case MSIM_M_ERROR: case MSIM_M_REQUEST_VOTE_RES: case MSIM_M_APPEND_ENTRIES_RES: case MSIM_M_TXN_OK:
    case MSIM_M_GENERATE_OK: case MSIM_M_TS_OK: case MSIM_M_SEND_OK: case MSIM_M_POLL_OK: case MSIM_M_LIST_COMMITTED_OFFSETS_OK: case MSIM_M_COMMIT_OFFSETS_OK: return true;
    default: return false;
  }
}

// topology builders of workload/broadcast.clj:40-185 (same shapes as topo_adj in csrc/wave_common.h)
std::vector<uint32_t> neighbours(uint32_t topology, uint32_t n, uint32_t a) {
  std::vector<uint32_t> v;
  switch (topology) {
    case MSIM_TOPO_GRID: {
      uint32_t side = 1; while (side * side < n) side++;
      const uint32_t i = a / side, j = a % side;
      if (i > 0) v.push_back(a - side);
      if (j > 0) v.push_back(a - 1);
      if (j + 1 < side && a + 1 < n) v.push_back(a + 1);
      if (a + side < n) v.push_back(a + side);
    } break;
    case MSIM_TOPO_LINE: if (a > 0) v.push_back(a - 1); if (a + 1 < n) v.push_back(a + 1); break;
    case MSIM_TOPO_TOTAL: for (uint32_t b = 0; b < n; b++) if (b != a) v.push_back(b); break;
    default: {
      const uint32_t b = topology == MSIM_TOPO_TREE2 ? 2 : topology == MSIM_TOPO_TREE3 ? 3 : 4;
      if (a > 0) v.push_back((a - 1) / b);
      for (uint32_t c = 1; c <= b; c++) if (b * a + c < n) v.push_back(b * a + c);
    }
  }
  return v;
}

void int_list_from_bitmap(FW &w, const uint32_t *words, uint32_t n) {
  size_t cnt = 0;
  for (uint32_t i = 0; i < n; i++) cnt += (size_t)__builtin_popcount(words[i]);
  w.list_header(cnt);
  for (uint32_t i = 0; i < n; i++) for (uint32_t x = words[i]; x; x &= x - 1) w.integer((int64_t)i * 32 + __builtin_ctz(x));
}

}  // namespace

extern "C" int msim_journal_fressian_rows(const msim_config *cfg, const msim_event *events, uint32_t n_events, const uint32_t *payload, uint32_t n_words,
                                          unsigned char *out, size_t cap, size_t *needed) {
  if (!cfg || (!events && n_events) || (!payload && n_words) || (!out && cap)) return MSIM_E_INVALID;
  const uint32_t N = cfg->n_nodes, slots = cfg->concurrency > N ? cfg->concurrency : N, wl = cfg->workload;
  FW w;
  w.out.reserve((size_t)n_events * 24 + 256);
  for (uint32_t i = 0; i < n_events; i++) {
    const msim_event &e = events[i];
    const uint32_t type = e.msg & 0x7Fu, id = e.msg >> 8, src = e.route & 0xFFu, dest = (e.route >> 8) & 0xFFu, mid = e.route >> 16;
    const bool recv = (e.msg & 0x80u) != 0;
    if (type == 0 || type >= sizeof MSG_TYPES / sizeof MSG_TYPES[0]) return MSIM_E_RANGE;
    w.tag("ev", 4);
    w.integer(i);                                    // :id = position: next-id starts at -1 and is pre-incremented (journal.clj:195,225-239)
    w.integer((int64_t)e.time_us * 1000);            // :time in ns
    w.keyword(recv ? "recv" : "send", true);
    w.tag("msg", 4);
    w.integer(id);
    w.str(endpoint(src, N, slots, cfg), true);
    w.str(endpoint(dest, N, slots, cfg), true);
    // ---- body ----
    w.raw(0xC0);   // tag "map"
    w.raw(0xED);   // begin closed list
    auto kv_type = [&](const char *t) { w.keyword("type", true); w.str(t, true); };
    auto kv_int = [&](const char *k, int64_t v) { w.keyword(k, true); w.integer(v); };
    kv_type(MSG_TYPES[type]);
    switch (type) {
      case MSIM_M_INIT:
        w.keyword("node_id", true); w.str(endpoint(dest, N, slots, cfg), false);
        w.keyword("node_ids", true); w.list_header(N); for (uint32_t k = 0; k < N; k++) w.str(endpoint(k, N, slots, cfg), false);
        break;
      case MSIM_M_TOPOLOGY:
        w.keyword("topology", true);
        w.raw(0xC0); w.list_header(2 * (size_t)N);   // an ordinary map: tag "map" + a list of keys and values
        for (uint32_t a = 0; a < N; a++) {
          w.keyword(endpoint(a, N, slots, cfg), false);
          const std::vector<uint32_t> nb = neighbours(cfg->topology, N, a);
          w.list_header(nb.size());
          for (uint32_t b : nb) w.str(endpoint(b, N, slots, cfg), false);
        }
        break;
      case MSIM_M_ECHO: case MSIM_M_ECHO_OK: { char b[32]; std::snprintf(b, sizeof b, "Please echo %u", e.a); w.keyword("echo", true); w.str(b, false); } break;
      case MSIM_M_BROADCAST: kv_int("message", e.a); break;
      case MSIM_M_ADD: kv_int(wl == MSIM_WL_G_SET ? "element" : "delta", wl == MSIM_WL_G_SET ? (int64_t)e.a : (int64_t)(int32_t)e.a); break;
      case MSIM_M_READ_OK:
        if (wl == MSIM_WL_BROADCAST || wl == MSIM_WL_G_SET) {
          const uint32_t off = e.a & 0xFFFFFFu, words = e.a >> 24;
          if ((uint64_t)off + words > n_words) return MSIM_E_RANGE;
          w.keyword(wl == MSIM_WL_BROADCAST ? "messages" : "value", true);
          int_list_from_bitmap(w, payload + off, words);
        } else if (wl == MSIM_WL_TXN_LIST_APPEND || wl == MSIM_WL_KAFKA) { w.keyword("elided", true); w.raw(0xF5); kv_int("a", e.a); }   // a root version / a thunk id; kafka: a chunk's element count / the offsets map's version (engine-internal, never a real body)
        else kv_int("value", wl == MSIM_WL_PN_COUNTER || wl == MSIM_WL_G_COUNTER ? (int64_t)(int32_t)e.a : (int64_t)e.a);
        break;
      case MSIM_M_ERROR: kv_int("code", e.a); break;
      case MSIM_M_TS: break;
      case MSIM_M_TS_OK: kv_int("ts", e.a); break;   // service.clj:123
      case MSIM_M_GENERATE_OK:
        if (cfg->node_program == MSIM_NODE_TSO_IDS) { kv_int("id", e.a); break; }
        w.keyword("id", true); w.list_header(3); w.integer(e.a >> 20); w.integer((e.a >> 5) & 0x7FFF); w.str(endpoint(e.a & 31, N, slots, cfg), false);
        break;
      case MSIM_M_INIT_OK: case MSIM_M_TOPOLOGY_OK: case MSIM_M_BROADCAST_OK: case MSIM_M_ADD_OK: case MSIM_M_READ: case MSIM_M_GENERATE:
        if (type == MSIM_M_READ && wl == MSIM_WL_LIN_KV) kv_int("key", e.a & 0xFF);
        else if (type == MSIM_M_READ && (wl == MSIM_WL_TXN_LIST_APPEND || wl == MSIM_WL_KAFKA)) { w.keyword("elided", true); w.raw(0xF5); kv_int("a", e.a); }   // the root / a thunk / a kafka chunk or the offsets map, by id
        break;
      case MSIM_M_TXN: case MSIM_M_TXN_OK: {   // [[f k v] ...] (doc/workloads.md txn-list-append / txn-rw-register); micro-ops in the payload area
        const uint32_t off = e.a & 0xFFFFFFu, nw = e.a >> 24;
        if ((uint64_t)off + nw > n_words) return MSIM_E_RANGE;
        const bool rw = wl == MSIM_WL_TXN_RW_REGISTER;
        size_t n_mops = 0;
        for (uint32_t k = 0; k < nw;) { const uint32_t h = payload[off + k++], x = (h >> 16) & 0xFFu; n_mops++; if (!rw && !(h & 1u) && x != 0xFFu) k += (x + 3) / 4; }
        w.keyword("txn", true);
        w.list_header(n_mops);
        for (uint32_t k = 0; k < nw;) {
          const uint32_t h = payload[off + k++], key = (h >> 1) & 0x7FFFu, x = (h >> 16) & 0xFFu;
          w.list_header(3);
          if (h & 1u) { w.str(rw ? "w" : "append", true); w.integer(key); w.integer(x); continue; }
          w.str("r", true); w.integer(key);
          if (x == 0xFFu) { w.raw(0xF7); continue; }   // nil
          if (rw) { w.integer(x); continue; }
          w.list_header(x);
          for (uint32_t el = 0; el < x; el++) w.integer((payload[off + k + el / 4] >> (8 * (el % 4))) & 0xFFu);
          k += (x + 3) / 4;
        }
      } break;
      case MSIM_M_WRITE: case MSIM_M_CAS: case MSIM_M_WRITE_OK: case MSIM_M_CAS_OK:
        if (wl == MSIM_WL_LIN_KV && type == MSIM_M_WRITE) { kv_int("key", e.a & 0xFF); kv_int("value", (e.a >> 8) & 0xFF); }
        else if (wl == MSIM_WL_LIN_KV && type == MSIM_M_CAS) { kv_int("key", e.a & 0xFF); kv_int("from", (e.a >> 8) & 0xFF); kv_int("to", (e.a >> 16) & 0xFF); }
        else if (wl != MSIM_WL_LIN_KV) { w.keyword("elided", true); w.raw(0xF5); kv_int("a", e.a); }   // storage traffic of the transactional nodes: versions, not values
        break;
      default: w.keyword("elided", true); w.raw(0xF5); kv_int("a", e.a); break;   // contents live in engine scratch (raft entries, replicate snapshots)
    }
    if (mid) kv_int(is_reply(type) ? "in_reply_to" : "msg_id", mid);
    w.raw(0xFD);   // end of the closed list
  }
  if (needed) *needed = w.out.size();
  if (cap == 0) return MSIM_OK;
  if (cap < w.out.size()) return MSIM_E_RANGE;
  std::memcpy(out, w.out.data(), w.out.size());
  return MSIM_OK;
}
