// unique_check_dev.hip — [upstream] jepsen.checker/unique-ids (workload/unique_ids.clj:67) on the device, one wavefront per history:
// :attempted-count = :invoke :generate ops, :acknowledged-count = :ok ones, :duplicated = values acknowledged more than once,
// :range = [min max]; valid iff nothing is duplicated.  pn_check.cpp sorts the acknowledged ids on the host after a fetch; here the
// rows are streamed once (1 KiB per load), every :ok id goes into an open-addressing table in HBM workspace (compare-and-swap on
// the key word, an atomic count beside it), and the duplicated values are the table slots counted twice or more.  Complete: no
// host pass behind it.  Histories whose ids fit a table in LDS (up to 19660 acknowledged ids: every bench shape) take
// unique_check_lds_kernel instead — a workgroup of 1024 threads (round 6; 256 before) per history, 32768 id slots and a "seen again" bitmap in 132 KiB of LDS:
// the HBM tables cost 256 KiB of initialisation and scattered compare-and-swaps per history (25 ms per 16384 histories of the demo shape).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "engine_internal.h"

namespace {

constexpr u32 EMPTY = 0xFFFFFFFFu;

struct UParams {
  const msim_op *rows; const msim_inst_meta *meta; msim_check_result *out;
  uint2 *ws;             // table_slots {id, count} per history of the launch
  u32 max_rows, table_slots /* power of two >= 2 x the ids of a history */, first;
};

__device__ __forceinline__ u32 u_sum(u32 v) { for (int o = 32; o; o >>= 1) v += (u32)__shfl_xor((int)v, o); return v; }
__device__ __forceinline__ u32 u_min(u32 v) { for (int o = 32; o; o >>= 1) v = min(v, (u32)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ u32 u_max(u32 v) { for (int o = 32; o; o >>= 1) v = max(v, (u32)__shfl_xor((int)v, o)); return v; }

__global__ void __launch_bounds__(64) unique_check_kernel(const UParams p) {
  const u32 lane = threadIdx.x, hist = p.first + blockIdx.x;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (u64)hist * p.max_rows;
  const u32 n = p.meta[hist].n_rows, flags = p.meta[hist].flags;
  uint2 *const tab = p.ws + (u64)blockIdx.x * p.table_slots;
  const u32 mask = p.table_slots - 1u;
  for (u32 i = lane; i < p.table_slots; i += 64) tab[i] = make_uint2(EMPTY, 0u);
  __syncthreads();

  u32 c_inv = 0, c_ok = 0, c_fail = 0, c_info = 0, c_att = 0, lo = EMPTY, hi = 0, n_empty_id = 0, n_ids = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u32 idx = base + lane;
    if (idx >= n) continue;
    const uint4 row = r[idx];
    const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12;
    if (proc == MSIM_PROCESS_NEMESIS) continue;
    c_inv += type == MSIM_T_INVOKE; c_ok += type == MSIM_T_OK; c_fail += type == MSIM_T_FAIL; c_info += type == MSIM_T_INFO;
    if (f != MSIM_F_GENERATE) continue;
    c_att += type == MSIM_T_INVOKE;
    if (type != MSIM_T_OK) continue;
    const u32 id = row.w;
    lo = min(lo, id); hi = max(hi, id); n_ids++;
    if (id == EMPTY) { n_empty_id++; continue; }   // (the one value the table cannot hold is counted apart)
    u32 h = (id * 0x9E3779B1u) >> 7;
    for (;;) {
      h &= mask;
      const u32 old = atomicCAS(&tab[h].x, EMPTY, id);
      if (old == EMPTY || old == id) { atomicAdd(&tab[h].y, 1u); break; }
      h++;
    }
  }
  __syncthreads();
  u32 dups = 0;
  for (u32 i = lane; i < p.table_slots; i += 64) dups += tab[i].y >= 2u ? 1u : 0u;
  dups = u_sum(dups); n_empty_id = u_sum(n_empty_id); n_ids = u_sum(n_ids);
  if (n_empty_id >= 2) dups++;
  c_inv = u_sum(c_inv); c_ok = u_sum(c_ok); c_fail = u_sum(c_fail); c_info = u_sum(c_info); c_att = u_sum(c_att);
  lo = u_min(lo); hi = u_max(hi);
  if (lane == 0) {
    msim_check_result o;
    o.valid = flags ? 0u : (dups == 0 ? 1u : 0u);
    o.attempt_count = c_att; o.stable_count = 0; o.lost_count = 0; o.never_read_count = 0; o.stale_count = 0; o.duplicated_count = dups; o.error_count = 0;
    for (int i = 0; i < 5; i++) o.stable_latency_ms[i] = 0;
    if (n_ids) { o.stable_latency_ms[0] = lo; o.stable_latency_ms[1] = hi; }   // :range
    o.op_count = c_inv; o.ok_count = c_ok; o.fail_count = c_fail; o.info_count = c_info;
    p.out[hist] = o;
  }
}

#ifndef UNIQ_WG   // threads per history: the table's 132 KiB leave a CU one workgroup, so the workgroup is what hides the rows' way from HBM (round 6: 256 -> 1024, four wavefronts per SIMD)
#define UNIQ_WG 1024
#endif
constexpr u32 LDS_SLOTS = 32768u;   // ids a workgroup's table holds (power of two); + LDS_SLOTS / 32 words of "seen again" bits

// the same check with the table in LDS: one workgroup per history
__global__ void __launch_bounds__(UNIQ_WG) unique_check_lds_kernel(const UParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *const tab = reinterpret_cast<u32 *>(smem);            // [LDS_SLOTS] id (EMPTY: free)
  u32 *const again = tab + LDS_SLOTS;                        // [LDS_SLOTS / 32] the slot's id was acknowledged more than once
  u32 *const hdr = again + LDS_SLOTS / 32;                   // [16] counters
  const u32 tid = threadIdx.x, hist = p.first + blockIdx.x;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (u64)hist * p.max_rows;
  const u32 n = p.meta[hist].n_rows, flags = p.meta[hist].flags;
  for (u32 i = tid; i < LDS_SLOTS; i += UNIQ_WG) tab[i] = EMPTY;
  for (u32 i = tid; i < LDS_SLOTS / 32 + 16; i += UNIQ_WG) again[i] = 0;
  __syncthreads();
  if (tid == 0) hdr[8] = EMPTY;   // (the minimum; the loop above zeroed the counters)
  __syncthreads();

  u32 c_inv = 0, c_ok = 0, c_fail = 0, c_info = 0, c_att = 0, lo = EMPTY, hi = 0, n_empty_id = 0, n_ids = 0;
  for (u32 idx = tid; idx < n; idx += UNIQ_WG) {
    const uint4 row = r[idx];
    const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12;
    if (proc == MSIM_PROCESS_NEMESIS) continue;
    c_inv += type == MSIM_T_INVOKE; c_ok += type == MSIM_T_OK; c_fail += type == MSIM_T_FAIL; c_info += type == MSIM_T_INFO;
    if (f != MSIM_F_GENERATE) continue;
    c_att += type == MSIM_T_INVOKE;
    if (type != MSIM_T_OK) continue;
    const u32 id = row.w;
    lo = min(lo, id); hi = max(hi, id); n_ids++;
    if (id == EMPTY) { n_empty_id++; continue; }   // (the one value the table cannot hold is counted apart)
    u32 h = (id * 0x9E3779B1u) >> 7;
    // (the host only takes this kernel when every row of the history could be an acknowledged id and the table still stays below a
    //  load factor of 0.6; the probe count is bounded all the same: a full table must end in a verdict, not in a hung GPU)
    u32 probes = 0;
    for (; probes < LDS_SLOTS; probes++) {
      h &= LDS_SLOTS - 1u;
      const u32 old = atomicCAS(&tab[h], EMPTY, id);
      if (old == EMPTY) break;
      if (old == id) { atomicOr(&again[h >> 5], 1u << (h & 31u)); break; }
      h++;
    }
    if (probes == LDS_SLOTS) atomicOr(&hdr[10], 1u);   // table full
  }
  atomicAdd(&hdr[0], c_inv); atomicAdd(&hdr[1], c_ok); atomicAdd(&hdr[2], c_fail); atomicAdd(&hdr[3], c_info); atomicAdd(&hdr[4], c_att);
  atomicAdd(&hdr[5], n_empty_id); atomicAdd(&hdr[6], n_ids); atomicMin(&hdr[8], lo); atomicMax(&hdr[9], hi);
  __syncthreads();
  u32 dups = 0;
  for (u32 i = tid; i < LDS_SLOTS / 32; i += UNIQ_WG) dups += (u32)__popc(again[i]);
  atomicAdd(&hdr[7], dups);
  __syncthreads();
  if (tid == 0) {
    u32 d = hdr[7];
    if (hdr[5] >= 2) d++;
    msim_check_result o;
    o.valid = (flags || hdr[10]) ? 0u : (d == 0 ? 1u : 0u);
    o.attempt_count = hdr[4]; o.stable_count = 0; o.lost_count = 0; o.never_read_count = 0; o.stale_count = 0; o.duplicated_count = d; o.error_count = hdr[10];
    for (int i = 0; i < 5; i++) o.stable_latency_ms[i] = 0;
    if (hdr[6]) { o.stable_latency_ms[0] = hdr[8]; o.stable_latency_ms[1] = hdr[9]; }   // :range
    o.op_count = hdr[0]; o.ok_count = hdr[1]; o.fail_count = hdr[2]; o.info_count = hdr[3];
    p.out[hist] = o;
  }
}

}  // namespace

// `paired`: the histories are the engine's own — every :ok row follows its :invoke row, so at most max_rows / 2 ids are acknowledged;
// rows handed in by a caller (msim_check_unique_batch) may be ALL acknowledgements (a history filtered to its completions).
static int unique_dev_run(msim_ctx *ctx, UParams up, u32 n, void **ws, size_t *ws_cap, hipStream_t st, bool paired) {
  // a load factor of 0.6 keeps the LDS table's probe sequences short (and the table can never fill up)
  const uint64_t max_ids = paired ? up.max_rows / 2 : up.max_rows;
  if (max_ids * 10 <= (uint64_t)LDS_SLOTS * 6 && !(msim_dev_flags(ctx) & 0x2000u)) {   // (MSIM_DEV_FLAGS bit 13: the HBM tables)
    const size_t lds = ((size_t)LDS_SLOTS + LDS_SLOTS / 32 + 16) * 4;
    MSIM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&unique_check_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    up.first = 0;
    hipLaunchKernelGGL(unique_check_lds_kernel, dim3(n), dim3(UNIQ_WG), lds, st, up);
    MSIM_HIP_TRY(ctx, hipGetLastError());
    return MSIM_OK;
  }
  u32 slots = 64; while (slots < (paired ? up.max_rows : 2 * up.max_rows)) slots <<= 1;   // >= 2 x the ids a history can acknowledge
  up.table_slots = slots;
  const uint64_t budget = 4ull << 30;
  const u32 chunk = (u32)std::min<uint64_t>(n, std::max<uint64_t>(1, budget / ((uint64_t)slots * 8)));
  const size_t need = (size_t)chunk * slots * 8;
  if (*ws_cap < need) {
    if (*ws) (void)msim_dev_free(*ws);
    *ws = nullptr; *ws_cap = 0;
    MSIM_HIP_TRY(ctx, msim_dev_malloc(ws, need));
    *ws_cap = need;
  }
  up.ws = static_cast<uint2 *>(*ws);
  for (u32 first = 0; first < n; first += chunk) {
    up.first = first;
    hipLaunchKernelGGL(unique_check_kernel, dim3(std::min(chunk, n - first)), dim3(64), 0, st, up);
    MSIM_HIP_TRY(ctx, hipGetLastError());
  }
  return MSIM_OK;
}

// msim_check for unique-ids: the histories of the last run, where they lie in HBM.
int msim_check_unique_device(msim_ctx *ctx) {
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  UParams up;
  up.rows = ctx->d_rows; up.meta = ctx->d_meta; up.out = ctx->d_check; up.max_rows = ctx->cfg.max_rows; up.ws = nullptr; up.table_slots = 0; up.first = 0;
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev2, ctx->stream));
  int rc = unique_dev_run(ctx, up, ctx->n_inst, &ctx->d_check_scratch, &ctx->cap_check_scratch, ctx->stream, true);
  if (rc != MSIM_OK) return rc;
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev3, ctx->stream));
  MSIM_HIP_TRY(ctx, hipEventSynchronize(ctx->ev3));
  MSIM_HIP_TRY(ctx, hipEventElapsedTime(&ctx->check_ms, ctx->ev2, ctx->ev3));
  ctx->checked = true; ctx->check_fetched = false;
  return MSIM_OK;
}

// Checks `n_histories` unique-ids histories given on the host, each in a slab of `max_rows` rows (history i at rows + i * max_rows,
// n_rows[i] of them used), with the device checker of msim_check; out[i] as msim_check_unique_rows would fill it.
extern "C" int msim_check_unique_batch(int device, const msim_op *rows, const uint32_t *n_rows, uint32_t max_rows, uint32_t n_histories, msim_check_result *out) {
  if (!rows || !n_rows || !out || n_histories == 0 || max_rows == 0) return MSIM_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_HIP;
  msim_ctx tmp_ctx; msim_ctx *ctx = &tmp_ctx;   // only for error text
  tmp_ctx.device = device;
  std::vector<msim_inst_meta> hm(n_histories);
  for (u32 i = 0; i < n_histories; i++) { std::memset(&hm[i], 0, sizeof hm[i]); if (n_rows[i] > max_rows) return MSIM_E_RANGE; hm[i].n_rows = n_rows[i]; }
  msim_op *d_rows = nullptr; msim_inst_meta *d_meta = nullptr; msim_check_result *d_out = nullptr; void *ws = nullptr; size_t ws_cap = 0;
  int rc = MSIM_E_HIP;
  do {
    if (msim_dev_malloc(&d_rows, (size_t)n_histories * max_rows * sizeof(msim_op)) != hipSuccess) break;
    if (msim_dev_malloc(&d_meta, (size_t)n_histories * sizeof(msim_inst_meta)) != hipSuccess) break;
    if (msim_dev_malloc(&d_out, (size_t)n_histories * sizeof(msim_check_result)) != hipSuccess) break;
    if (hipMemcpy(d_rows, rows, (size_t)n_histories * max_rows * sizeof(msim_op), hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_meta, hm.data(), (size_t)n_histories * sizeof(msim_inst_meta), hipMemcpyHostToDevice) != hipSuccess) break;
    UParams up;
    up.rows = d_rows; up.meta = d_meta; up.out = d_out; up.max_rows = max_rows; up.ws = nullptr; up.table_slots = 0; up.first = 0;
    rc = unique_dev_run(ctx, up, n_histories, &ws, &ws_cap, nullptr, false);
    if (rc != MSIM_OK) break;
    rc = hipMemcpy(out, d_out, (size_t)n_histories * sizeof(msim_check_result), hipMemcpyDeviceToHost) == hipSuccess ? MSIM_OK : MSIM_E_HIP;
  } while (false);
  for (void *q : {(void *)d_rows, (void *)d_meta, (void *)d_out, ws}) if (q) (void)msim_dev_free(q);
  return rc;
}
