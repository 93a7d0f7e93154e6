// kafka_check.cpp — the checker of the kafka workload (msim_check for MSIM_WL_KAFKA), on the host cores.
//
// What the reference wires in is [upstream] jepsen.tests.kafka's checker (workload/kafka.clj:288-311 takes the whole workload from
// it); it is not vendored.  What IS in the reference is the description of the anomalies it looks for, workload/kafka.clj:21-70, and
// this file restates exactly those (PARITY UNPINNED beyond the one example the reference prints, :42-60, which tests/ reproduces):
//   * the log of a key is what all observations agree on: every :ok send gives (offset, message), every :ok poll a run of them.
//     An offset seen with two different messages is `inconsistent-offsets`, a message seen at two offsets of a key `duplicate`;
//   * lost write (:23-25): an acknowledged send whose offset no poll ever returned although some poll returned a HIGHER offset of
//     that key; unobserved (:25-28): acknowledged, never polled, and nothing above it polled either — reported, not an error
//     ("there is no recency requirement");
//   * nonmonotonic (:30-37): a client's offsets of a key must strictly increase — inside one poll (`int-nonmonotonic-poll`),
//     between two polls of one process (`nonmonotonic-poll`), between two acknowledged sends of one process (`nonmonotonic-send`);
//   * skip (:39-62): a client's polls of a key jump over an offset that is known to exist — inside one poll (`int-poll-skip`) or
//     between two polls of one process (`poll-skip`, the example at :42-60);
//   * "external nonmonotonic and skip errors are not tracked across assign operations" (:64-70) — nor across a crash, which gives the
//     worker a new process and a new client, nor across a poll that did not complete :ok (its client has already moved on, see below);
//   * aborted read: a poll returns a message whose send definitely failed (the cas lost: error 30).
// out: valid = 1 iff none of the errors above (2 = :unknown when nothing was acknowledged and nothing polled), attempt_count = sends
// invoked, stable_count = sends acknowledged, lost_count = lost writes, never_read_count = unobserved, duplicated_count = duplicates,
// error_count = MSIM_KAFKA_* bits, op / ok / fail / info counts as everywhere.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "engine_internal.h"

namespace {

constexpr uint32_t KEYS = 8, OFFS = 2048, NOMSG = 0xFFFFu;

struct Tables {
  uint16_t msg[KEYS][OFFS];            // the key's log as observed (NOMSG: offset never seen)
  uint8_t polled[KEYS][OFFS], acked[KEYS][OFFS];
  uint16_t where[KEYS][OFFS];          // message value -> 1 + offset it was seen at
  uint8_t failed[KEYS][OFFS];          // message value: its send definitely failed
  uint8_t dup_at[KEYS][OFFS];          // offset already counted as a duplicate's second home
};

struct Proc { uint32_t process; bool open; int32_t last_poll[KEYS], last_send[KEYS]; };

void observe(Tables &t, uint32_t k, uint32_t o, uint32_t m, uint32_t &bits, uint32_t &dups) {
  if (k >= KEYS || o >= OFFS || m >= OFFS) { bits |= MSIM_KAFKA_MALFORMED; return; }
  if (t.msg[k][o] != NOMSG && t.msg[k][o] != m) bits |= MSIM_KAFKA_INCONSISTENT_OFFSETS;
  t.msg[k][o] = (uint16_t)m;
  if (t.where[k][m] && t.where[k][m] != o + 1) { bits |= MSIM_KAFKA_DUPLICATE; if (!t.dup_at[k][o]) { t.dup_at[k][o] = 1; dups++; } }
  else t.where[k][m] = (uint16_t)(o + 1);
}

// is some offset strictly between a and b known to exist?
bool known_between(const Tables &t, uint32_t k, int32_t a, int32_t b) {
  for (int32_t o = a + 1; o < b; o++) if (o >= 0 && (uint32_t)o < OFFS && t.msg[k][o] != NOMSG) return true;
  return false;
}

void check_kafka(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, msim_check_result *out, Tables &t) {
  std::memset(out, 0, sizeof *out);
  std::memset(t.msg, 0xFF, sizeof t.msg);
  std::memset(t.polled, 0, sizeof t.polled); std::memset(t.acked, 0, sizeof t.acked);
  std::memset(t.where, 0, sizeof t.where); std::memset(t.failed, 0, sizeof t.failed); std::memset(t.dup_at, 0, sizeof t.dup_at);
  uint32_t bits = 0, dups = 0;

  // pass 1: what exists (every observation), what was acknowledged, what was polled, which sends failed
  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    const uint32_t type = MSIM_OP_TYPE(r), f = MSIM_OP_F(r);
    if (MSIM_OP_PROCESS(r) == MSIM_PROCESS_NEMESIS) continue;
    if (type == MSIM_T_INVOKE) out->op_count++;
    else if (type == MSIM_T_OK) out->ok_count++; else if (type == MSIM_T_FAIL) out->fail_count++; else out->info_count++;
    if (f == MSIM_F_SEND) {
      const uint32_t k = r.value & 63u, m = (r.value >> 6) & 0x7FFu, o = r.value >> 17;
      if (type == MSIM_T_INVOKE) out->attempt_count++;
      else if (type == MSIM_T_OK) {
        if (o == 0x7FFu || k >= KEYS) { bits |= MSIM_KAFKA_MALFORMED; continue; }
        out->stable_count++;
        observe(t, k, o, m, bits, dups);
        if (k < KEYS && o < OFFS) t.acked[k][o] = 1;
      } else if (type == MSIM_T_FAIL && k < KEYS && m < OFFS) t.failed[k][m] = 1;
    } else if (f == MSIM_F_POLL && type == MSIM_T_OK) {
      const uint32_t len = MSIM_OP_LEN(r);
      if (len == 0) continue;
      if ((uint64_t)r.value + len > n_words) { bits |= MSIM_KAFKA_MALFORMED; continue; }
      const uint32_t *w = payload + r.value;
      for (uint32_t p = 0; p < len;) {
        const uint32_t h = w[p++], k = h & 7u, n = (h >> 8) & 0xFFu, o0 = h >> 16;
        if (p + (n + 1) / 2 > len) { bits |= MSIM_KAFKA_MALFORMED; break; }
        for (uint32_t e = 0; e < n; e++) {
          const uint32_t m = (w[p + e / 2] >> (16 * (e & 1))) & 0xFFFFu;
          observe(t, k, o0 + e, m, bits, dups);
          if (o0 + e < OFFS) t.polled[k][o0 + e] = 1;
        }
        p += (n + 1) / 2;
      }
    }
  }

  // lost / unobserved writes; aborted reads
  for (uint32_t k = 0; k < KEYS; k++) {
    int32_t top = -1;
    for (uint32_t o = 0; o < OFFS; o++) if (t.polled[k][o]) top = (int32_t)o;
    for (uint32_t o = 0; o < OFFS; o++) {
      if (t.acked[k][o] && !t.polled[k][o]) { if ((int32_t)o < top) { out->lost_count++; bits |= MSIM_KAFKA_LOST_WRITE; } else out->never_read_count++; }
      if (t.polled[k][o] && t.msg[k][o] != NOMSG && t.failed[k][t.msg[k][o]]) bits |= MSIM_KAFKA_ABORTED_READ;
    }
  }

  // pass 2: every process's own sequence of offsets per key
  std::vector<Proc> procs;
  auto proc_of = [&](uint32_t process) -> Proc & {
    for (Proc &p : procs) if (p.process == process) return p;
    Proc p; p.process = process; p.open = true;
    for (uint32_t k = 0; k < KEYS; k++) { p.last_poll[k] = -1; p.last_send[k] = -1; }
    procs.push_back(p);
    return procs.back();
  };
  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    const uint32_t type = MSIM_OP_TYPE(r), f = MSIM_OP_F(r), process = MSIM_OP_PROCESS(r);
    if (process == MSIM_PROCESS_NEMESIS) continue;
    // the positions change: nothing is tracked across an assign — nor across a poll that failed or crashed: the client has moved its
    // offsets past what that poll brought (workload/kafka.clj:181-186 runs before the commit that can fail, :223-230), the history never shows it
    if ((f == MSIM_F_ASSIGN && type == MSIM_T_INVOKE) || (f == MSIM_F_POLL && (type == MSIM_T_FAIL || type == MSIM_T_INFO))) {
      Proc &p = proc_of(process);
      for (uint32_t k = 0; k < KEYS; k++) p.last_poll[k] = -1;
      continue;
    }
    if (type != MSIM_T_OK) continue;
    if (f == MSIM_F_SEND) {
      const uint32_t k = r.value & 63u, o = r.value >> 17;
      if (k >= KEYS || o >= OFFS) continue;
      Proc &p = proc_of(process);
      if (p.last_send[k] >= 0 && (int32_t)o <= p.last_send[k]) bits |= MSIM_KAFKA_NONMONOTONIC_SEND;
      p.last_send[k] = (int32_t)o;
    } else if (f == MSIM_F_POLL) {
      const uint32_t len = MSIM_OP_LEN(r);
      if (len == 0 || (uint64_t)r.value + len > n_words) continue;
      Proc &p = proc_of(process);
      const uint32_t *w = payload + r.value;
      uint32_t in_this_poll = 0;   // keys that already had a run in this poll: what follows is judged as an internal error
      for (uint32_t q = 0; q < len;) {
        const uint32_t h = w[q++], k = h & 7u, n = (h >> 8) & 0xFFu, o0 = h >> 16;
        if (q + (n + 1) / 2 > len) break;
        q += (n + 1) / 2;
        if (!n) continue;
        // a block holds one run o0, o0 + 1, ...: a key's pairs that do not continue the run start another block of the same key
        const int32_t first = (int32_t)o0, last = (int32_t)(o0 + n - 1);
        const bool internal = (in_this_poll >> k) & 1u;
        if (p.last_poll[k] >= 0) {
          if (first <= p.last_poll[k]) bits |= internal ? MSIM_KAFKA_INT_NONMONOTONIC_POLL : MSIM_KAFKA_NONMONOTONIC_POLL;
          else if (first > p.last_poll[k] + 1 && known_between(t, k, p.last_poll[k], first)) bits |= internal ? MSIM_KAFKA_INT_POLL_SKIP : MSIM_KAFKA_POLL_SKIP;
        }
        p.last_poll[k] = last; in_this_poll |= 1u << k;
      }
    }
  }

  out->duplicated_count = dups;
  out->error_count = bits;
  const uint32_t errors = bits & ~0u;
  if (flags || errors) out->valid = 0;
  else out->valid = (out->stable_count == 0 && out->ok_count == 0) ? 2u : 1u;
}

}  // namespace

// one history on the calling thread (kafka_check_dev.hip: what the device pass could not prove clean)
void msim_kafka_check_instance_host(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, msim_check_result *out) {
  std::vector<Tables> t(1);
  check_kafka(rows, n_rows, payload, n_words, flags, out, t[0]);
}

extern "C" int msim_check_kafka_rows(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, msim_check_result *out) {
  if ((!rows && n_rows) || !out || (!payload && n_words)) return MSIM_E_INVALID;
  std::vector<Tables> t(1);
  check_kafka(rows, n_rows, payload, n_words, 0, out, t[0]);
  return MSIM_OK;
}

int msim_check_kafka_host(msim_ctx *ctx) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc = msim_fetch(ctx);
  if (rc != MSIM_OK) return rc;
  const uint32_t n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  unsigned nt = msim_host_threads();
  if (nt > n) nt = n;
  std::vector<std::thread> th;
  for (unsigned w = 0; w < nt; w++)
    th.emplace_back([ctx, n, nt, w]() {
      std::vector<Tables> t(1);
      for (uint32_t i = w; i < n; i += nt)
        check_kafka(ctx->h_rows + ctx->h_row_off[i], ctx->h_meta[i].n_rows, ctx->h_payload + ctx->h_pay_off[i], ctx->h_meta[i].n_payload_words,
                    ctx->h_meta[i].flags, &ctx->h_check[i], t[0]);
    });
  for (auto &x : th) x.join();
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->d_check, ctx->h_check, (size_t)n * sizeof(msim_check_result), hipMemcpyHostToDevice));
  ctx->checked = true; ctx->check_fetched = true;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return MSIM_OK;
}
