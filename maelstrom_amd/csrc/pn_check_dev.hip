// pn_check_dev.hip — the pn-counter / g-counter checker (workload/pn_counter.clj:84-123) on the device, one wavefront per history.
//
// "Every final read is the sum of all known-completed adds plus any number of possibly-completed adds": the acceptable values are
// {sum of :ok adds} + every subset sum of the :info adds' deltas.  pn_check.cpp keeps that set as sorted integer ranges (the
// reference: a Guava TreeRangeSet) on the host after a fetch.  Here it is a BITMAP over [sum + negative deltas, sum + positive
// deltas] — bit b <=> value base + b — one 64-bit word per lane, 4096 values wide: an :info add of delta d is B |= B shifted by d
// (two `ds_bpermute`s per half word), the number of ranges is the number of 0 -> 1 transitions, a final read is one bit test.
// A history whose window is wider than 4096 values or that has more than 1024 indeterminate adds goes to the host checker
// (pn_check.cpp) — same result either way.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "engine_internal.h"

void msim_pn_check_instance_host(const msim_op *rows, uint32_t n_rows, uint32_t flags, msim_check_result *res);   // pn_check.cpp

namespace {

constexpr u32 NEEDS_HOST = 3u;
constexpr u32 MAX_INFO = 1024u;

typedef long long s64;

__device__ __forceinline__ u32 p_sum(u32 v) { for (int o = 32; o; o >>= 1) v += (u32)__shfl_xor((int)v, o); return v; }
__device__ __forceinline__ s64 p_sum64(s64 v) {
  for (int o = 32; o; o >>= 1) { const u32 lo = (u32)__shfl_xor((int)(u32)v, o), hi = (u32)__shfl_xor((int)(u32)((u64)v >> 32), o); v += (s64)(((u64)hi << 32) | lo); }
  return v;
}
__device__ __forceinline__ u64 p_get64(u64 v, int src_lane) {   // v of lane src_lane (0 outside the wavefront)
  const bool in = src_lane >= 0 && src_lane < 64;
  const int a = (in ? src_lane : 0) << 2;
  const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(a, (int)(u32)v), hi = (u32)__builtin_amdgcn_ds_bpermute(a, (int)(u32)(v >> 32));
  return in ? (((u64)hi << 32) | lo) : 0ull;
}

__global__ void __launch_bounds__(64) pn_check_kernel(const msim_op *rows_all, const msim_inst_meta *meta, msim_check_result *out, u32 max_rows) {
  __shared__ int deltas[MAX_INFO];
  __shared__ u64 bits[64];
  __shared__ u32 n_info_s;
  const u32 lane = threadIdx.x, hist = blockIdx.x;
  const uint4 *const r = reinterpret_cast<const uint4 *>(rows_all) + (u64)hist * max_rows;
  const u32 n = meta[hist].n_rows, flags = meta[hist].flags;
  if (lane == 0) n_info_s = 0;
  __syncthreads();

  u32 c_inv = 0, c_ok = 0, c_fail = 0, c_info = 0;
  s64 definite = 0, neg = 0, pos = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 idx = base + lane;
    if (idx >= n) continue;
    const uint4 row = r[idx];
    const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12;
    if (proc == MSIM_PROCESS_NEMESIS) continue;
    c_inv += type == MSIM_T_INVOKE; c_ok += type == MSIM_T_OK; c_fail += type == MSIM_T_FAIL; c_info += type == MSIM_T_INFO;
    if (f != MSIM_F_ADD) continue;
    const int d = (int)row.w;
    if (type == MSIM_T_OK) definite += d;
    if (type == MSIM_T_INFO) {
      if (d < 0) neg += d; else pos += d;
      const u32 k = atomicAdd(&n_info_s, 1u);
      if (k < MAX_INFO) deltas[k] = d;
    }
  }
  __syncthreads();
  definite = p_sum64(definite); neg = p_sum64(neg); pos = p_sum64(pos);
  c_inv = p_sum(c_inv); c_ok = p_sum(c_ok); c_fail = p_sum(c_fail); c_info = p_sum(c_info);
  const u32 n_info = n_info_s;
  msim_check_result o;
  o.valid = NEEDS_HOST; o.attempt_count = 0; o.stable_count = 0; o.lost_count = 0; o.never_read_count = 0; o.stale_count = 0; o.duplicated_count = 0; o.error_count = 0;
  for (int i = 0; i < 5; i++) o.stable_latency_ms[i] = 0;
  o.op_count = c_inv; o.ok_count = c_ok; o.fail_count = c_fail; o.info_count = c_info;
  if (n_info > MAX_INFO || pos - neg >= 4096) { if (lane == 0) out[hist] = o; return; }

  // the acceptable set: bit b of the 4096-bit map (word `lane`) <=> value lo_v + b
  const s64 lo_v = definite + neg;
  u64 B = 0;
  { const u32 b0 = (u32)(-neg); if ((b0 >> 6) == lane) B = 1ull << (b0 & 63u); }
  for (u32 k = 0; k < n_info; k++) {   // (any order: the set of subset sums does not depend on it)
    const int d = deltas[k];
    if (d == 0) continue;
    const u32 s = (u32)(d < 0 ? -d : d), q = s >> 6, rr = s & 63u;
    u64 sh;
    if (d > 0) {   // towards higher values: word w takes from words w - q and w - q - 1
      const u64 a = p_get64(B, (int)lane - (int)q), b = p_get64(B, (int)lane - (int)q - 1);
      sh = rr ? ((a << rr) | (b >> (64u - rr))) : a;
    } else {
      const u64 a = p_get64(B, (int)lane + (int)q), b = p_get64(B, (int)lane + (int)q + 1);
      sh = rr ? ((a >> rr) | (b << (64u - rr))) : a;
    }
    B |= sh;
  }
  bits[lane] = B;
  __syncthreads();
  // ranges = maximal runs of ones
  { const u64 prev_top = p_get64(B, (int)lane - 1) >> 63; o.stable_count = p_sum((u32)__popcll(B & ~((B << 1) | prev_top))); }
  // final reads that completed :ok must lie in the set
  u32 attempts = 0, errors = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 idx = base + lane;
    if (idx >= n) continue;
    const uint4 row = r[idx];
    if ((row.z >> 12) == MSIM_PROCESS_NEMESIS || !((row.z >> 11) & 1u) || (row.z & 3u) != MSIM_T_OK) continue;
    attempts++;
    const s64 rel = (s64)(int)row.w - lo_v;
    const bool ok = rel >= 0 && rel < 4096 && ((bits[(u32)rel >> 6] >> ((u32)rel & 63u)) & 1ull);
    errors += ok ? 0u : 1u;
  }
  attempts = p_sum(attempts); errors = p_sum(errors);
  if (lane == 0) {
    o.attempt_count = attempts; o.error_count = errors;
    o.valid = flags ? 0u : (errors == 0 ? 1u : 0u);
    out[hist] = o;
  }
}

}  // namespace

// runs the kernel over n histories in slabs of max_rows rows and lets the host checker finish what the bitmap does not cover
static int pn_dev_run(msim_ctx *ctx, const msim_op *d_rows, const msim_inst_meta *d_meta, msim_check_result *d_out, u32 max_rows, u32 n,
                      msim_check_result *h_out, hipStream_t st, u32 *n_host) {
  hipLaunchKernelGGL(pn_check_kernel, dim3(n), dim3(64), 0, st, d_rows, d_meta, d_out, max_rows);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_out, d_out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<u32> todo;
  for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HOST) todo.push_back(i);
  if (!todo.empty()) {   // wider than the bitmap: the host checker
    std::vector<msim_inst_meta> hm(n);
    MSIM_HIP_TRY(ctx, hipMemcpy(hm.data(), d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
    std::vector<msim_op> rows;
    for (u32 i : todo) {
      rows.resize(hm[i].n_rows ? hm[i].n_rows : 1);
      if (hm[i].n_rows) MSIM_HIP_TRY(ctx, hipMemcpy(rows.data(), d_rows + (size_t)i * max_rows, (size_t)hm[i].n_rows * sizeof(msim_op), hipMemcpyDeviceToHost));
      msim_pn_check_instance_host(rows.data(), hm[i].n_rows, hm[i].flags, &h_out[i]);
      MSIM_HIP_TRY(ctx, hipMemcpy(d_out + i, &h_out[i], sizeof(msim_check_result), hipMemcpyHostToDevice));
    }
  }
  if (n_host) *n_host = (u32)todo.size();
  return MSIM_OK;
}

// msim_check for pn-counter / g-counter: the histories of the last run, where they lie in HBM.
int msim_check_pn_device(msim_ctx *ctx) {
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const u32 n = ctx->n_inst;
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  const auto t0 = std::chrono::steady_clock::now();
  u32 redone = 0;
  int rc = pn_dev_run(ctx, ctx->d_rows, ctx->d_meta, ctx->d_check, ctx->cfg.max_rows, n, ctx->h_check, ctx->stream, &redone);
  if (rc != MSIM_OK) return rc;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->lin_host_rechecks = redone;
  ctx->checked = true; ctx->check_fetched = true;
  return MSIM_OK;
}

// Checks `n_histories` pn-counter / g-counter histories given on the host, each in a slab of `max_rows` rows (history i at
// rows + i * max_rows, n_rows[i] of them used), with the device checker of msim_check; out[i] as msim_check_pn_rows would fill it.
extern "C" int msim_check_pn_batch(int device, const msim_op *rows, const uint32_t *n_rows, uint32_t max_rows, uint32_t n_histories, msim_check_result *out) {
  if (!rows || !n_rows || !out || n_histories == 0 || max_rows == 0) return MSIM_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_HIP;
  msim_ctx tmp_ctx; msim_ctx *ctx = &tmp_ctx;   // only for error text
  tmp_ctx.device = device;
  std::vector<msim_inst_meta> hm(n_histories);
  for (u32 i = 0; i < n_histories; i++) { std::memset(&hm[i], 0, sizeof hm[i]); if (n_rows[i] > max_rows) return MSIM_E_RANGE; hm[i].n_rows = n_rows[i]; }
  msim_op *d_rows = nullptr; msim_inst_meta *d_meta = nullptr; msim_check_result *d_out = nullptr;
  int rc = MSIM_E_HIP;
  do {
    if (msim_dev_malloc(&d_rows, (size_t)n_histories * max_rows * sizeof(msim_op)) != hipSuccess) break;
    if (msim_dev_malloc(&d_meta, (size_t)n_histories * sizeof(msim_inst_meta)) != hipSuccess) break;
    if (msim_dev_malloc(&d_out, (size_t)n_histories * sizeof(msim_check_result)) != hipSuccess) break;
    if (hipMemcpy(d_rows, rows, (size_t)n_histories * max_rows * sizeof(msim_op), hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_meta, hm.data(), (size_t)n_histories * sizeof(msim_inst_meta), hipMemcpyHostToDevice) != hipSuccess) break;
    rc = pn_dev_run(ctx, d_rows, d_meta, d_out, max_rows, n_histories, out, nullptr, nullptr);
  } while (false);
  for (void *q : {(void *)d_rows, (void *)d_meta, (void *)d_out}) if (q) (void)msim_dev_free(q);
  return rc;
}
