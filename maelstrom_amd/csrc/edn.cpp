// edn.cpp — host-side history.edn writer: binary history rows -> the text Jepsen stores and its checkers read
// (one op map per line, {:type :f :value :time :process :index [:error] [:final?]}; SURVEY.md §8b "History surface",
// sample at doc/05-datomic/02-shared-state.md:384-386).  :value shapes per workload: echo.clj:36-37, broadcast.clj:206-209,
// g_set.clj:43-45, lin_kv.clj:53-67 (independent tuples), txn_list_append.clj:27-39, txn_rw_register.clj:82-84,
// pn_counter.clj:22-58, unique_ids.clj / flake_ids.clj:30-31; nemesis ops as [upstream] jepsen.nemesis.combined writes them.
// No device code; the same text as maelstrom_amd.engine.history_edn(decode_history(...)) (tests/test_edn_writer.py).
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/maelsim.h"
#include "engine_limits.h"

namespace {

void num(std::string &o, long long v) { char b[32]; std::snprintf(b, sizeof b, "%lld", v); o += b; }
void nil_or(std::string &o, uint32_t v) { if (v == 0xFF) o += "nil"; else num(o, v); }

void bitmap(std::string &o, const uint32_t *w, uint32_t n) {
  o += '[';
  bool first = true;
  for (uint32_t i = 0; i < n; i++)
    for (uint32_t x = w[i]; x; x &= x - 1) { if (!first) o += ' '; first = false; num(o, i * 32 + (uint32_t)__builtin_ctz(x)); }
  o += ']';
}

void txn(std::string &o, const uint32_t *w, uint32_t n, bool rw) {
  o += '[';
  bool first = true;
  for (uint32_t i = 0; i < n;) {
    const uint32_t h = w[i++], key = (h >> 1) & 0x7FFF, x = (h >> 16) & 0xFF;
    if (!first) o += ' ';
    first = false;
    o += '[';
    o += (h & 1) ? (rw ? ":w " : ":append ") : ":r ";
    num(o, key); o += ' ';
    if (h & 1) num(o, x);
    else if (x == 0xFF) o += "nil";
    else if (rw) num(o, x);
    else {
      o += '[';
      for (uint32_t e = 0; e < x && i + e / 4 < n; e++) { if (e) o += ' '; num(o, (w[i + e / 4] >> (8 * (e % 4))) & 0xFF); }
      o += ']';
      i += (x + 3) / 4;
    }
    o += ']';
  }
  o += ']';
}

const char *const TYPES[] = {":invoke", ":ok", ":fail", ":info"};
const char *const FS[] = {":echo", ":broadcast", ":read", ":add", ":start-partition", ":stop-partition", ":write", ":cas", ":txn", ":generate", ":send", ":poll", ":assign", ":crash"};
const char *const SPECS[] = {":one", ":majority", ":majorities-ring", ":minority-third"};

}  // namespace

extern "C" int msim_history_edn_rows(const msim_config *cfg, const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words,
                                     char *out, size_t cap, size_t *needed) {
  if (!cfg || (!rows && n_rows) || (!payload && n_words) || (!out && cap)) return MSIM_E_INVALID;
  const uint32_t wl = cfg->workload, n = cfg->n_nodes;
  std::string o;
  o.reserve((size_t)n_rows * 96);
  for (uint32_t idx = 0; idx < n_rows; idx++) {
    const msim_op &r = rows[idx];
    const uint32_t typ = MSIM_OP_TYPE(r), f = MSIM_OP_F(r), err = MSIM_OP_ERR(r), fin = MSIM_OP_FINAL(r), proc = MSIM_OP_PROCESS(r), ln = MSIM_OP_LEN(r);
    const uint32_t v = r.value;
    if ((uint64_t)v + ln > n_words && ln) return MSIM_E_RANGE;
    o += "{:type "; o += TYPES[typ];
    o += ", :f "; if (f < sizeof FS / sizeof FS[0]) o += FS[f]; else num(o, f);
    o += ", :value ";
    const bool counter = wl == MSIM_WL_PN_COUNTER || wl == MSIM_WL_G_COUNTER;
    if (wl == MSIM_WL_LIN_KV && (f == MSIM_F_READ || f == MSIM_F_WRITE || f == MSIM_F_CAS)) {   // [k v] / [k [v v']]
      o += '['; num(o, v & 0xFF); o += ' ';
      if (f == MSIM_F_CAS) { o += '['; nil_or(o, (v >> 8) & 0xFF); o += ' '; nil_or(o, (v >> 16) & 0xFF); o += ']'; }
      else nil_or(o, (v >> 8) & 0xFF);
      o += ']';
    } else if (f == MSIM_F_GENERATE && cfg->node_program == MSIM_NODE_TSO_IDS) {   // a lin-tso timestamp
      if (typ == MSIM_T_OK) num(o, v); else o += "nil";
    } else if (f == MSIM_F_GENERATE) {
      if (typ == MSIM_T_OK) { o += '['; num(o, v >> 20); o += ' '; num(o, (v >> 5) & 0x7FFF); o += " \"n"; num(o, v & 31); o += "\"]"; }
      else o += "nil";
    } else if (f == MSIM_F_SEND) {   // [[:send "k" msg]] / [[:send "k" [offset msg]]] (workload/kafka.clj:188-190; keys are strings, :247-283)
      o += "[[:send \""; num(o, v & 63u); o += "\" ";
      if ((v >> 17) == 0x7FFu) num(o, (v >> 6) & 0x7FFu); else { o += '['; num(o, v >> 17); o += ' '; num(o, (v >> 6) & 0x7FFu); o += ']'; }
      o += "]]";
    } else if (f == MSIM_F_POLL) {   // [[:poll]] / [[:poll {"k" [[offset msg] ...]}]] (:171-186); a key's runs are joined
      if (typ != MSIM_T_OK) o += "[[:poll]]";
      else {
        o += "[[:poll {";
        bool firstk = true;
        for (uint32_t k = 0; k < 8; k++) {
          bool has = false;
          for (uint32_t i = 0; i < ln;) { const uint32_t h = payload[v + i]; if ((h & 7u) == k) has = true; i += 1 + (((h >> 8) & 0xFFu) + 1) / 2; }
          if (!has) continue;
          if (!firstk) o += ", ";
          firstk = false;
          o += '"'; num(o, k); o += "\" [";
          bool firstp = true;
          for (uint32_t i = 0; i < ln;) {
            const uint32_t h = payload[v + i], cnt = (h >> 8) & 0xFFu, o0 = h >> 16;
            if ((h & 7u) == k) for (uint32_t e = 0; e < cnt && i + 1 + e / 2 < ln; e++) {
              if (!firstp) o += ' ';
              firstp = false;
              o += '['; num(o, o0 + e); o += ' '; num(o, (payload[v + i + 1 + e / 2] >> (16 * (e & 1))) & 0xFFFFu); o += ']';
            }
            i += 1 + (cnt + 1) / 2;
          }
          o += ']';
        }
        o += "}]]";
      }
    } else if (f == MSIM_F_ASSIGN) {   // ["k" ...] (:207-220)
      o += '[';
      for (uint32_t i = 0; i < ln; i++) { if (i) o += ' '; o += '"'; num(o, payload[v + i] & 7u); o += '"'; }
      o += ']';
    } else if (f == MSIM_F_CRASH) { o += "nil";
    } else if (f == MSIM_F_TXN) txn(o, payload + v, ln, wl == MSIM_WL_TXN_RW_REGISTER);
    else if (counter && (f == MSIM_F_ADD || f == MSIM_F_READ)) {
      if (f == MSIM_F_ADD || typ == MSIM_T_OK) num(o, (int32_t)v); else o += "nil";
    } else if (f == MSIM_F_READ) {
      if (typ == MSIM_T_OK) bitmap(o, payload + v, ln); else o += "nil";
    } else if (f == MSIM_F_ECHO) {
      if (typ == MSIM_T_OK) { o += "{:type \"echo_ok\", :echo \"Please echo "; num(o, v); o += "\"}"; }
      else { o += "\"Please echo "; num(o, v); o += '"'; }
    } else if (f == MSIM_F_START_PARTITION) {
      if (ln) {   // the grudge: [:isolated {"n1" ["n0" ..] ..}], only nodes that drop someone
        o += "[:isolated {";
        bool first = true;
        for (uint32_t d = 0; d < n && (d + 1) * MSIM_MASK_WORDS <= ln; d++) {
          const uint32_t *g = payload + v + d * MSIM_MASK_WORDS;
          bool any = false;
          for (uint32_t k = 0; k < MSIM_MASK_WORDS; k++) any |= g[k] != 0;
          if (!any) continue;
          if (!first) o += ", ";
          first = false;
          o += "\"n"; num(o, d); o += "\" [";
          bool f2 = true;
          for (uint32_t k = 0; k < MSIM_MASK_WORDS; k++)
            for (uint32_t x = g[k]; x; x &= x - 1) { if (!f2) o += ' '; f2 = false; o += "\"n"; num(o, k * 32 + (uint32_t)__builtin_ctz(x)); o += '"'; }
          o += ']';
        }
        o += "}]";
      } else o += v < 4 ? SPECS[v] : "nil";
    } else if (f == MSIM_F_STOP_PARTITION) {
      o += (idx == 0 || rows[idx - 1].packed != r.packed) ? "nil" : ":network-healed";
    } else if (v == MSIM_NO_VALUE) o += "nil";
    else num(o, v);
    o += ", :time "; num(o, (long long)(r.time_len & 0xFFFFFFFFFFFFull));
    o += ", :process "; if (proc == MSIM_PROCESS_NEMESIS) o += ":nemesis"; else num(o, proc);
    o += ", :index "; num(o, idx);
    switch (err) {   // with-errors: :net-timeout, or [name text] of an RPC error (client.clj:155-172)
      case MSIM_ERR_NET_TIMEOUT: o += ", :error :net-timeout"; break;
      case MSIM_ERR_TEMPORARILY_UNAVAILABLE: o += ", :error [:temporarily-unavailable \"not a leader\"]"; break;
      case MSIM_ERR_KEY_DOES_NOT_EXIST: o += ", :error [:key-does-not-exist \"not found\"]"; break;
      case MSIM_ERR_PRECONDITION_FAILED: o += ", :error [:precondition-failed \"cas mismatch\"]"; break;
      case MSIM_ERR_TXN_CONFLICT: o += ", :error [:txn-conflict \"root altered\"]"; break;
      case MSIM_ERR_TIMEOUT: o += ", :error [:timeout \"promise timed out\"]"; break;
      case MSIM_ERR_ABORT: o += ", :error [:abort \"aborted\"]"; break;
      default: break;
    }
    if (fin) o += ", :final? true";
    if (f == MSIM_F_ASSIGN && ln && (payload[v] >> 31)) o += ", :seek-to-beginning? true";
    o += "}\n";
  }
  if (needed) *needed = o.size() + 1;
  if (cap < o.size() + 1) return cap ? MSIM_E_RANGE : MSIM_OK;   // cap == 0: size query
  std::memcpy(out, o.c_str(), o.size() + 1);
  return MSIM_OK;
}
