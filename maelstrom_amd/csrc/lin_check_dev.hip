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

// outcome word of an invocation row: bit 31 = has a completion, bits 0-1 its type, bits 8-23 its value bytes 1 and 2
template <int CPL>   // configurations per lane: 64 * CPL at a time
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
  res.valid = 0; res.attempt_count = 0; res.stable_count = 0; res.lost_count = 0; res.never_read_count = 0; res.stale_count = 0;
  res.duplicated_count = 0; res.error_count = 0;
  for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = 0;
  res.op_count = 0; res.ok_count = 0; res.fail_count = 0; res.info_count = 0;
  if (n > p.table_rows) { if (lane == 0) { res.valid = NEEDS_HOST; p.out[inst] = res; } return; }

  for (u32 i = lane; i < 256; i += 64) { key_lo[i] = 0xFFFFFFFFu; key_hi[i] = 0; }
  __syncthreads();

  bool needs_host = false;   // wave-uniform

  // ---- pass 0: counts, key ranges, invocation -> completion --------------------------------------------------------------
  u32 c_inv = 0, c_ok = 0, c_fail = 0, c_info = 0;
  {
    bool o_used = false; u32 o_proc = 0, o_key = 0, o_row = 0;   // lane = one open call
    for (u32 base = 0; base < n; base += 64) {
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
            if (used == ~0ull) { needs_host = true; break; }
            s = (u32)__builtin_ctzll(~used);
          }
          if (lane == s) { o_used = true; o_proc = jp; o_key = jk; o_row = base + j; }
        } else if (hit) {
          const u32 s = (u32)__builtin_ctzll(hit);
          const u32 irow = rl(o_row, s);
          if (lane == s) { o_used = false; outcome[irow] = 0x80000000u | jt | (w & 0xFFFF00u); }
        }
      }
      if (needs_host) break;
    }
  }
  __syncthreads();
  c_inv = l_wave_sum(c_inv); c_ok = l_wave_sum(c_ok); c_fail = l_wave_sum(c_fail); c_info = l_wave_sum(c_info);
  res.op_count = c_inv; res.ok_count = c_ok; res.fail_count = c_fail; res.info_count = c_info;

  // ---- pass 1: key by key ---------------------------------------------------------------------------------------------------
  u32 n_keys = 0, n_bad = 0, n_unknown = 0;
  for (u32 k = 0; k < 256 && !needs_host; k++) {
    const u32 lo = key_lo[k], hi = key_hi[k];
    if (lo == 0xFFFFFFFFu) continue;
    n_keys++;
    // configurations: lane i holds (c_lin[b], c_val[b]) while bit i of alive[b] is set; pending calls: lane s holds slot s
    u64 c_lin[CPL]; u32 c_val[CPL]; u64 alive[CPL], expanded[CPL];
#pragma unroll
    for (int b = 0; b < CPL; b++) { c_lin[b] = 0; c_val[b] = 0xFFu; alive[b] = 0; expanded[b] = 0; }
    alive[0] = 1ull;
    u64 pending = 0, info_bits = 0;
    u32 s_proc = 0, s_op = 0; bool s_ok = false;       // s_op = f | v1 << 8 | v2 << 16 | skip << 31
    bool bad = false, unknown = false;
    for (u32 base = lo & ~63u; base <= hi && !bad && !unknown && !needs_host; base += 64) {
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
          if (pending == ~0ull) { unknown = true; break; }
          const u32 s = (u32)__builtin_ctzll(~pending);
          pending |= 1ull << s;
          if (!ok) info_bits |= 1ull << s;
          if (lane == s) { s_proc = jp; s_op = jf | vv | (skip << 31); s_ok = ok; }
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
          if (++explored > EXPLORE_LIMIT) { needs_host = true; break; }
          u64 cand = pending & ~lin_i;
          while (cand) {
            const u32 q = (u32)__builtin_ctzll(cand); cand &= cand - 1;
            const u32 op = rl(s_op, q);
            if (op >> 31) continue;
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
            if (!placed) { needs_host = true; break; }
          }
          if (needs_host) break;
        }
        if (needs_host) break;
        u64 any = 0;
#pragma unroll
        for (int b = 0; b < CPL; b++) if (alive[b]) {
          alive[b] = __ballot(((alive[b] >> lane) & 1ull) && (c_lin[b] & bit));   // the others could not linearize the call in time
          c_lin[b] &= ~bit;
          any |= alive[b];
        }
        pending &= ~bit;
        if (!any) { bad = true; break; }
      }
    }
    n_bad += bad; n_unknown += unknown;
  }

  if (lane == 0) {
    res.attempt_count = n_keys;    // keys checked (independent/checker)
    res.error_count = n_bad;       // keys whose history is not linearizable
    res.valid = needs_host ? NEEDS_HOST : flags ? 0u : n_bad ? 0u : n_unknown ? 2u : 1u;
    p.out[inst] = res;
  }
}

// launches the kernel over `n` histories; the ones that exceed 64 configurations run again with 512, and the host search
// (all host threads) finishes what is left
int lin_check_dev_run(msim_ctx *ctx, const LParams &lp0, u32 n, u32 max_rows_any, msim_check_result *h_out, hipStream_t st, u32 *n_host) {
  LParams lp = lp0;
  const bool trace = (msim_dev_flags(ctx) & 0x1000u) != 0;   // developer: time the passes
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const u32 table_cap = (60u * 1024u - 2048u) / 4u;
  lp.table_rows = max_rows_any < table_cap ? max_rows_any : table_cap;
  lp.list = nullptr;
  const size_t lds = 2048 + (size_t)lp.table_rows * 4;
  hipLaunchKernelGGL((lin_check_kernel<1>), dim3(n), dim3(64), lds, st, lp);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_out, lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<u32> todo;
  for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HOST) todo.push_back(i);
  if (trace) std::fprintf(stderr, "[lin-check] pass 1 (64 configurations): %.2f ms, %zu of %u histories marked\n", ms(), todo.size(), n);
  if (!todo.empty()) {
    // Second pass on the device: eight configurations per lane.  While it runs, the host cores already search marked histories
    // — those with the most indeterminate calls first: they are the likeliest to exceed 512 configurations too — so that what the
    // second pass leaves over is mostly done by the time it is known.  Whichever side finishes a history first, the result is the
    // same (both searches are exact).
    std::vector<msim_inst_meta> hm;
    std::vector<uint64_t> ho;
    if (lp.meta) { hm.resize(n); MSIM_HIP_TRY(ctx, hipMemcpy(hm.data(), lp.meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost)); }
    else { ho.resize(n + 1); MSIM_HIP_TRY(ctx, hipMemcpy(ho.data(), lp.off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost)); }
    std::vector<msim_check_result> h2(n);           // the second pass's results (h_out keeps pass 1's until merged)
    std::vector<msim_check_result> hh(todo.size()); // the host's results, by position in `order`
    std::vector<u32> order(todo);
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return h_out[x].info_count > h_out[y].info_count; });
    std::vector<char> host_done(todo.size(), 0);
    std::atomic<size_t> next{0};
    std::atomic<bool> device_done{false};
    std::atomic<int> copy_err{0};
    std::vector<char> wanted;                       // after the second pass: which histories the host still has to do
    auto host_one = [&](size_t k) {
      const u32 i = order[k];
      const u32 nr = lp.meta ? hm[i].n_rows : (u32)(ho[i + 1] - ho[i]);
      const uint64_t first = lp.meta ? (uint64_t)i * lp.stride : ho[i];
      std::vector<msim_op> rows(nr ? nr : 1);
      if (nr && hipMemcpy(rows.data(), lp.rows + first, (size_t)nr * sizeof(msim_op), hipMemcpyDeviceToHost) != hipSuccess) { copy_err = 1; return; }
      msim_lin_check_instance_host(rows.data(), nr, lp.meta ? hm[i].flags : 0u, &hh[k]);
      host_done[k] = 1;
    };
    unsigned nt = msim_host_threads();
    if (nt > todo.size()) nt = (unsigned)todo.size();
    std::vector<std::thread> th;
    const int dev_id = ctx->device;
    for (unsigned w = 0; w < nt; w++)
      th.emplace_back([&]() {
        (void)hipSetDevice(dev_id);
        for (;;) {   // phase 1: speculative, hardest first, until the device is done; phase 2: what the device left over
          const size_t k = next.fetch_add(1);
          if (k >= order.size()) break;
          if (device_done.load() && !wanted[k]) continue;
          host_one(k);
        }
      });
    u32 *d_list = nullptr;
    hipError_t e = hipMalloc(&d_list, todo.size() * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_list, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      lp.list = d_list;
      hipLaunchKernelGGL((lin_check_kernel<8>), dim3((u32)todo.size()), dim3(64), lds, st, lp);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h2.data(), lp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    wanted.assign(order.size(), 0);
    if (e == hipSuccess) for (size_t k = 0; k < order.size(); k++) wanted[k] = h2[order[k]].valid == NEEDS_HOST;
    else std::fill(wanted.begin(), wanted.end(), 1);   // (the host can still do everything)
    device_done = true;
    const double t_dev = ms();
    for (auto &x : th) x.join();
    if (d_list) (void)hipFree(d_list);
    if (copy_err) { ctx->err = "lin-kv check: copying a history to the host failed"; return MSIM_E_HIP; }
    // a history the host threads skipped in phase 1 order but the device could not finish either (taken by no one): do it now
    u32 n_host_needed = 0;
    for (size_t k = 0; k < order.size(); k++) {
      const u32 i = order[k];
      if (wanted[k]) { n_host_needed++; if (!host_done[k]) host_one(k); h_out[i] = hh[k]; MSIM_HIP_TRY(ctx, hipMemcpy(lp.out + i, &h_out[i], sizeof(msim_check_result), hipMemcpyHostToDevice)); }
      else h_out[i] = h2[i];
    }
    if (copy_err) { ctx->err = "lin-kv check: copying a history to the host failed"; return MSIM_E_HIP; }
    if (trace) std::fprintf(stderr, "[lin-check] pass 2 (512 configurations) done at %.2f ms, %u histories needed the host search (overlapped)\n", t_dev, n_host_needed);
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
