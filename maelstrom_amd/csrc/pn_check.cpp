// pn_check.cpp — the small host-side checkers: pn-counter / g-counter (workload/pn_counter.clj:84-123) and unique-ids
// ([upstream] jepsen.checker/unique-ids, unique_ids.clj:67).
//
// "Every final read is the sum of all known-completed adds plus any number of possibly-completed adds": the acceptable
// set starts as {sum of :ok adds}; every :info add (a timed-out add may or may not have happened) unions in the set
// shifted by its delta.  The reference keeps the set in a Guava TreeRangeSet of open ranges (lower-1, upper+1) so that
// adjacent integers merge; the same set is kept here as sorted closed integer ranges, merged when they touch.  Reads
// marked :final? that completed :ok must lie in it.  Pinned by the reference's own vectors (test/maelstrom/workload/
// pn_counter_test.clj:10-36) in tests/test_pn_counter.py.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>

#include "engine_internal.h"

namespace {

typedef std::vector<std::pair<int64_t, int64_t>> Ranges;

void merge_in(Ranges &r) {
  std::sort(r.begin(), r.end());
  size_t w = 0;
  for (size_t i = 0; i < r.size(); i++) {
    if (w && r[i].first <= r[w - 1].second + 1) r[w - 1].second = std::max(r[w - 1].second, r[i].second);
    else r[w++] = r[i];
  }
  r.resize(w);
}

void check_history(const msim_op *rows, uint32_t n_rows, uint32_t flags, msim_check_result *out, Ranges &acc) {
  std::memset(out, 0, sizeof *out);
  int64_t definite = 0;
  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    if (MSIM_OP_PROCESS(r) == MSIM_PROCESS_NEMESIS) continue;
    const uint32_t t = MSIM_OP_TYPE(r);
    if (t == MSIM_T_INVOKE) out->op_count++; else if (t == MSIM_T_OK) out->ok_count++; else if (t == MSIM_T_FAIL) out->fail_count++; else out->info_count++;
    if (MSIM_OP_F(r) == MSIM_F_ADD && t == MSIM_T_OK) definite += (int32_t)r.value;
  }
  acc.clear();
  acc.emplace_back(definite, definite);
  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    if (MSIM_OP_PROCESS(r) == MSIM_PROCESS_NEMESIS || MSIM_OP_F(r) != MSIM_F_ADD || MSIM_OP_TYPE(r) != MSIM_T_INFO) continue;
    const int64_t d = (int32_t)r.value;
    const size_t n = acc.size();
    for (size_t k = 0; k < n; k++) acc.emplace_back(acc[k].first + d, acc[k].second + d);
    merge_in(acc);
  }
  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    if (MSIM_OP_PROCESS(r) == MSIM_PROCESS_NEMESIS || !MSIM_OP_FINAL(r) || MSIM_OP_TYPE(r) != MSIM_T_OK) continue;
    out->attempt_count++;
    const int64_t v = (int32_t)r.value;
    bool ok = false;
    for (const auto &g : acc) if (g.first <= v && v <= g.second) { ok = true; break; }
    if (!ok) out->error_count++;
  }
  out->stable_count = (uint32_t)acc.size();
  out->valid = flags ? 0u : (out->error_count == 0 ? 1u : 0u);
}

// [upstream] jepsen.checker/unique-ids: :attempted-count = :invoke :generate ops, :acknowledged-count = :ok ones,
// :duplicated = values acknowledged more than once, :range = [min max]; valid iff nothing is duplicated.
void check_unique(const msim_op *rows, uint32_t n_rows, uint32_t flags, msim_check_result *out, std::vector<uint32_t> &ids) {
  std::memset(out, 0, sizeof *out);
  ids.clear();
  for (uint32_t i = 0; i < n_rows; i++) {
    const msim_op &r = rows[i];
    if (MSIM_OP_PROCESS(r) == MSIM_PROCESS_NEMESIS) continue;
    const uint32_t t = MSIM_OP_TYPE(r);
    if (t == MSIM_T_INVOKE) out->op_count++; else if (t == MSIM_T_OK) out->ok_count++; else if (t == MSIM_T_FAIL) out->fail_count++; else out->info_count++;
    if (MSIM_OP_F(r) != MSIM_F_GENERATE) continue;
    if (t == MSIM_T_INVOKE) out->attempt_count++;
    if (t == MSIM_T_OK) ids.push_back(r.value);
  }
  std::sort(ids.begin(), ids.end());
  uint32_t dups = 0;
  for (size_t i = 1; i < ids.size(); i++) if (ids[i] == ids[i - 1] && (i < 2 || ids[i] != ids[i - 2])) dups++;
  out->duplicated_count = dups;
  if (!ids.empty()) { out->stable_latency_ms[0] = ids.front(); out->stable_latency_ms[1] = ids.back(); }
  out->valid = flags ? 0u : (dups == 0 ? 1u : 0u);
}

}  // namespace

// the host checker for one pn-counter / g-counter history (pn_check_dev.hip hands over what its bitmap does not cover)
void msim_pn_check_instance_host(const msim_op *rows, uint32_t n_rows, uint32_t flags, msim_check_result *res) { Ranges acc; check_history(rows, n_rows, flags, res, acc); }

extern "C" int msim_check_unique_rows(const msim_op *rows, uint32_t n_rows, msim_check_result *out) {
  if ((!rows && n_rows) || !out) return MSIM_E_INVALID;
  std::vector<uint32_t> ids;
  check_unique(rows, n_rows, 0, out, ids);
  return MSIM_OK;
}

int msim_check_unique_host(msim_ctx *ctx) {
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
      std::vector<uint32_t> ids;
      for (uint32_t i = t; i < n; i += nt)
        check_unique(ctx->h_rows + ctx->h_row_off[i], ctx->h_meta[i].n_rows, ctx->h_meta[i].flags, &ctx->h_check[i], ids);
    });
  for (auto &x : th) x.join();
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->d_check, ctx->h_check, (size_t)n * sizeof(msim_check_result), hipMemcpyHostToDevice));
  ctx->checked = true; ctx->check_fetched = true;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return MSIM_OK;
}

extern "C" int msim_check_pn_rows(const msim_op *rows, uint32_t n_rows, msim_check_result *out, int64_t *ranges, uint32_t cap, uint32_t *n_ranges) {
  if ((!rows && n_rows) || !out) return MSIM_E_INVALID;
  Ranges acc;
  check_history(rows, n_rows, 0, out, acc);
  if (n_ranges) *n_ranges = (uint32_t)acc.size();
  if (ranges) for (uint32_t i = 0; i < cap && i < acc.size(); i++) { ranges[2 * i] = acc[i].first; ranges[2 * i + 1] = acc[i].second; }
  return MSIM_OK;
}

int msim_check_pn_host(msim_ctx *ctx) {
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
      Ranges acc;
      for (uint32_t i = t; i < n; i += nt)
        check_history(ctx->h_rows + ctx->h_row_off[i], ctx->h_meta[i].n_rows, ctx->h_meta[i].flags, &ctx->h_check[i], acc);
    });
  for (auto &x : th) x.join();
  MSIM_HIP_TRY(ctx, hipMemcpy(ctx->d_check, ctx->h_check, (size_t)n * sizeof(msim_check_result), hipMemcpyHostToDevice));
  ctx->checked = true; ctx->check_fetched = true;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return MSIM_OK;
}
