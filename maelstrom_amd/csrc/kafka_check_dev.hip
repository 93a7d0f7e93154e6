// kafka_check_dev.hip — the kafka checker on the device (msim_check for MSIM_WL_KAFKA), one workgroup per history.
//
// kafka_check.cpp restates the anomalies workload/kafka.clj:21-70 describes and runs on the host cores after the histories crossed
// PCIe (139 ms per 16384 histories of the bench shape, all of them clean).  This file does for it what txn_check_dev.hip does for the
// list-append analysis: the device PROVES a history free of every anomaly and then reports what the host reports for such a history
// (the op counts, sends attempted / acknowledged, the unobserved writes — "there is no recency requirement", :25-28 —, :valid?);
// anything else — an observation that disagrees with another one, a lost write, an aborted read, a process whose offsets do not
// strictly increase or jump over a known offset, a row that does not decode, an offset or message beyond the tables — is answered
// NEEDS_HOST and the host checker classifies it (kafka_dev_run below): never approximated.
//
// Layout.  The tables of kafka_check.cpp (the key's log as observed, where each message was seen, polled / acknowledged / failed
// flags) live in LDS, `T` entries per key (the configuration bounds offsets and messages by max-writes-per-key; a first pass over
// the launch's histories finds the highest one they name, kafka_bound_kernel: the bench shape needs 288 of the 1056 its configuration
// allows, 27 KB of LDS per history instead of 77):
//   pass 1  the workgroup's threads stride over the rows: every :ok send and every pair of every :ok poll is one observation — a
//           compare-and-swap on the log entry and one on the message's home, an atomic OR for the flags, an atomic max for the
//           highest polled offset of the key;
//   pass 1b threads stride over (key, offset): acknowledged and never polled -> lost if something above it was polled, else
//           unobserved; polled with a message whose send failed -> aborted read;
//   pass 2  a process's own sequence of offsets (:30-62).  A worker thread's processes follow one another in the history (a crash gives
//           the worker process + concurrency), so worker w's rows are one sequential stream: wavefront (w mod 4) walks the rows 64 at
//           a time, takes its workers' relevant rows in order (a ballot), and lane k holds key k's last polled / sent offset of the
//           worker in LDS; a poll's payload is staged into LDS with one coalesced load.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "engine_internal.h"

void msim_kafka_check_instance_host(const msim_op *rows, uint32_t n_rows, const uint32_t *payload, uint32_t n_words, uint32_t flags, msim_check_result *out);   // kafka_check.cpp

namespace {

constexpr u32 NEEDS_HOST = 3u;
constexpr u32 EMPTY = 0xFFFFFFFFu;
constexpr u32 KEYS = 8u;          // kafka_check.cpp's bound (a poll block names its key in 3 bits)
constexpr u32 OFFS = 2048u;       // kafka_check.cpp's bound on offsets and messages
constexpr u32 NT = 512u, NWV = NT / 64u;   // (round 6: 256 -> 512 threads per history: 14.1 -> 10.0 ms per 16384 at the bench shape; 1024: 21.8)
constexpr u32 STAGE = 128u;       // payload words of a poll staged per wavefront (longer polls read the rest from HBM)
constexpr u32 CMAX = 64u;         // worker threads
constexpr u32 WST = 17u;          // per worker: last polled offset x 8, last sent offset x 8, current process

struct KCParams {
  const msim_op *rows; const u32 *payload; const msim_inst_meta *meta;
  const uint64_t *row_off, *pay_off;   // (null: history i at i * max_rows / i * max_pay)
  msim_check_result *out;
  u32 max_rows, max_pay;
  u32 T;      // table entries per key (a multiple of 32, <= OFFS)
  u32 C;      // worker threads: process p belongs to worker p mod C
};

__device__ __forceinline__ u32 k_rl(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ void k_wave_fence() {   // orders the LDS traffic of ONE wavefront across its lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

size_t kafka_lds_bytes(u32 T) { return ((size_t)KEYS * T * 2 + 3 * (size_t)KEYS * T / 32 + KEYS + 16 + CMAX * WST + NWV * STAGE) * 4; }

__global__ void __launch_bounds__(NT) kafka_check_kernel(const KCParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 hist = blockIdx.x, T = p.T, KT = KEYS * T, C = p.C;
  u32 *const msg = reinterpret_cast<u32 *>(smem);          // [KEYS][T] the key's log as observed (EMPTY: offset never seen)
  u32 *const where = msg + KT;                            // [KEYS][T] message -> 1 + the offset it was seen at
  u32 *const polled = where + KT;                         // [KEYS * T / 32] bits
  u32 *const acked = polled + KT / 32;
  u32 *const failed = acked + KT / 32;                    // by message: its send definitely failed
  u32 *const top1 = failed + KT / 32;                     // [KEYS] 1 + the highest polled offset
  u32 *const hdr = top1 + KEYS;                           // [16]: [0] not provably clean, [1..] counters
  u32 *const wst = hdr + 16;                              // [CMAX][WST]
  u32 *const stg = wst + CMAX * WST + wave * STAGE;       // this wavefront's staging area

  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.row_off ? p.row_off[hist] : (u64)hist * p.max_rows);
  const u32 *const pay = p.payload + (p.pay_off ? p.pay_off[hist] : (u64)hist * p.max_pay);
  const u32 n = p.meta ? p.meta[hist].n_rows : (u32)(p.row_off[hist + 1] - p.row_off[hist]);
  const u32 n_words = p.meta ? p.meta[hist].n_payload_words : (u32)(p.pay_off[hist + 1] - p.pay_off[hist]);
  const u32 flags = p.meta ? p.meta[hist].flags : 0u;

  for (u32 i = tid; i < KT; i += NT) { msg[i] = EMPTY; where[i] = 0; }
  for (u32 i = tid; i < 3 * KT / 32 + KEYS + 16; i += NT) polled[i] = 0;
  for (u32 i = tid; i < CMAX * WST; i += NT) wst[i] = EMPTY;
  __syncthreads();

#define TO_HOST() do { hdr[0] = 1u; } while (0)
#define SETBIT(a_, i_) atomicOr(&(a_)[(i_) >> 5], 1u << ((i_) & 31u))
#define GETBIT(a_, i_) (((a_)[(i_) >> 5] >> ((i_) & 31u)) & 1u)
  // every observation of (key, offset, message) must agree with every other one (inconsistent offsets, duplicates: the host's)
  auto observe = [&](u32 k, u32 o, u32 m) __attribute__((always_inline)) {
    if (k >= KEYS || o >= T || m >= T) { TO_HOST(); return; }
    const u32 was = atomicCAS(&msg[k * T + o], EMPTY, m);
    if (was != EMPTY && was != m) TO_HOST();
    const u32 home = atomicCAS(&where[k * T + m], 0u, o + 1u);
    if (home != 0u && home != o + 1u) TO_HOST();
  };

  // ---- pass 1: what exists, what was acknowledged, what was polled, which sends failed ----
  u32 c_inv = 0, c_ok = 0, c_fail = 0, c_info = 0, c_att = 0, c_stable = 0;
  for (u32 idx = tid; idx < n; idx += NT) {
    const uint4 row = r[idx];
    const u32 type = row.z & 3u, f = (row.z >> 2) & 31u, proc = row.z >> 12;
    if (proc == MSIM_PROCESS_NEMESIS) continue;
    c_inv += type == MSIM_T_INVOKE; c_ok += type == MSIM_T_OK; c_fail += type == MSIM_T_FAIL; c_info += type == MSIM_T_INFO;
    if (f == MSIM_F_SEND) {
      const u32 k = row.w & 63u, m = (row.w >> 6) & 0x7FFu, o = row.w >> 17;
      if (type == MSIM_T_INVOKE) c_att++;
      else if (type == MSIM_T_OK) {
        if (o == 0x7FFu || k >= KEYS) { TO_HOST(); continue; }
        c_stable++;
        observe(k, o, m);
        if (o < T) SETBIT(acked, k * T + o);
      } else if (type == MSIM_T_FAIL) {
        if (k < KEYS && m < T) SETBIT(failed, k * T + m); else if (k < KEYS && m < OFFS) TO_HOST();
      }
    } else if (f == MSIM_F_POLL && type == MSIM_T_OK) {
      const u32 len = row.y >> 16;
      if (len == 0) continue;
      if ((u64)row.w + len > n_words) { TO_HOST(); continue; }
      const u32 *const w = pay + row.w;
      for (u32 q = 0; q < len;) {
        const u32 h = w[q++], k = h & 7u, cnt = (h >> 8) & 0xFFu, o0 = h >> 16;
        if (q + (cnt + 1) / 2 > len) { TO_HOST(); break; }
        for (u32 e = 0; e < cnt; e++) {
          const u32 m = (w[q + e / 2] >> (16 * (e & 1))) & 0xFFFFu;
          observe(k, o0 + e, m);
          if (o0 + e < T) { SETBIT(polled, k * T + o0 + e); atomicMax(&top1[k], o0 + e + 1u); }
        }
        q += (cnt + 1) / 2;
      }
    }
  }
  atomicAdd(&hdr[1], c_inv); atomicAdd(&hdr[2], c_ok); atomicAdd(&hdr[3], c_fail); atomicAdd(&hdr[4], c_info);
  atomicAdd(&hdr[5], c_att); atomicAdd(&hdr[6], c_stable);
  __syncthreads();

  msim_check_result res;
  res.valid = NEEDS_HOST; res.attempt_count = 0; res.stable_count = 0; res.lost_count = 0; res.never_read_count = 0; res.stale_count = 0;
  res.duplicated_count = 0; res.error_count = 0;
  for (int i = 0; i < 5; i++) res.stable_latency_ms[i] = 0;
  res.op_count = 0; res.ok_count = 0; res.fail_count = 0; res.info_count = 0;
#define LEAVE_IF_DECIDED() do { if (hdr[0]) { if (tid == 0) p.out[hist] = res; return; } } while (0)
  LEAVE_IF_DECIDED();

  // ---- pass 1b: lost / unobserved writes, aborted reads ----
  {
    u32 unobs = 0;
    for (u32 i = tid; i < KT; i += NT) {
      const u32 k = i / T, o = i - k * T;
      const bool pl = GETBIT(polled, i) != 0;
      if (GETBIT(acked, i) && !pl) { if (o + 1u < top1[k]) TO_HOST(); else unobs++; }
      if (pl) { const u32 m = msg[i]; if (m != EMPTY && GETBIT(failed, k * T + m)) TO_HOST(); }
    }
    atomicAdd(&hdr[7], unobs);
  }

  // ---- pass 2: every process's own sequence of offsets per key ----
  for (u32 base = 0; base < n; base += 64) {
    const u32 idx = base + lane;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (idx < n) row = r[idx];
    const u32 proc_l = row.z >> 12, type_l = row.z & 3u, f_l = (row.z >> 2) & 31u;
    const bool rel = idx < n && proc_l != MSIM_PROCESS_NEMESIS && (proc_l % C) % NWV == wave &&
                     ((f_l == MSIM_F_ASSIGN && type_l == MSIM_T_INVOKE) || (f_l == MSIM_F_POLL && (type_l == MSIM_T_FAIL || type_l == MSIM_T_INFO)) ||
                      (type_l == MSIM_T_OK && (f_l == MSIM_F_SEND || f_l == MSIM_F_POLL)));
    u64 todo = __ballot(rel);
    while (todo) {
      const u32 j = (u32)__builtin_ctzll(todo);
      todo &= todo - 1;
      const u32 rz = k_rl(row.z, j), rw = k_rl(row.w, j), ry = k_rl(row.y, j);
      const u32 type = rz & 3u, f = (rz >> 2) & 31u, proc = rz >> 12;
      u32 *const s = wst + (proc % C) * WST;
      const u32 cur = s[16];
      if (cur != proc) {   // the worker's next process starts with nothing remembered; a process that RETURNS is not a worker's stream
        if (cur != EMPTY && proc < cur) TO_HOST();
        k_wave_fence();
        if (lane < 16) s[lane] = EMPTY; else if (lane == 16) s[16] = proc;
        k_wave_fence();
      }
      // the positions change: nothing is tracked across an assign, nor across a poll that failed or crashed (kafka_check.cpp)
      if (type != MSIM_T_OK) { if (lane < 8) s[lane] = EMPTY; continue; }
      if (f == MSIM_F_SEND) {
        const u32 k = rw & 63u, o = rw >> 17;
        if (k >= KEYS || o >= OFFS) continue;
        if (lane == k) { const u32 last = s[8 + k]; if (last != EMPTY && o <= last) TO_HOST(); s[8 + k] = o; }
        continue;
      }
      const u32 len = ry >> 16;
      if (len == 0) continue;
      for (u32 q = lane; q < len && q < STAGE; q += 64) stg[q] = pay[rw + q];
      k_wave_fence();
      for (u32 q = 0; q < len;) {
        const u32 h = q < STAGE ? stg[q] : pay[rw + q];
        q++;
        const u32 k = h & 7u, cnt = (h >> 8) & 0xFFu, o0 = h >> 16;
        if (q + (cnt + 1) / 2 > len) break;
        q += (cnt + 1) / 2;
        if (!cnt) continue;
        // a block holds one run o0, o0 + 1, ...: it must start above the worker's last offset of the key and skip nothing known
        if (lane == k) {
          const u32 lp = s[k];
          if (lp != EMPTY) {
            if (o0 <= lp) TO_HOST();
            else for (u32 o = lp + 1; o < o0; o++) if (o < T && msg[k * T + o] != EMPTY) { TO_HOST(); break; }
          }
          s[k] = o0 + cnt - 1u;
        }
      }
      k_wave_fence();   // (the staging area is rewritten by the next poll)
    }
  }
  __syncthreads();
  LEAVE_IF_DECIDED();

  if (tid == 0) {
    res.op_count = hdr[1]; res.ok_count = hdr[2]; res.fail_count = hdr[3]; res.info_count = hdr[4];
    res.attempt_count = hdr[5]; res.stable_count = hdr[6]; res.never_read_count = hdr[7];
    res.valid = flags ? 0u : ((hdr[6] == 0 && hdr[2] == 0) ? 2u : 1u);
    p.out[hist] = res;
  }
#undef LEAVE_IF_DECIDED
#undef TO_HOST
#undef SETBIT
#undef GETBIT
}

// The highest offset / message value any :ok send or poll of the launch names: sizes the tables (most tests use a fraction of what
// max-writes-per-key allows, and the LDS a history's tables take decides how many histories a CU works on at once).
__global__ void __launch_bounds__(NT) kafka_bound_kernel(const KCParams p, u32 *bound) {
  const u32 tid = threadIdx.x, hist = blockIdx.x;
  const uint4 *const r = reinterpret_cast<const uint4 *>(p.rows) + (p.row_off ? p.row_off[hist] : (u64)hist * p.max_rows);
  const u32 *const pay = p.payload + (p.pay_off ? p.pay_off[hist] : (u64)hist * p.max_pay);
  const u32 n = p.meta ? p.meta[hist].n_rows : (u32)(p.row_off[hist + 1] - p.row_off[hist]);
  const u32 n_words = p.meta ? p.meta[hist].n_payload_words : (u32)(p.pay_off[hist + 1] - p.pay_off[hist]);
  u32 hi = 0;
  for (u32 idx = tid; idx < n; idx += NT) {
    const uint4 row = r[idx];
    const u32 type = row.z & 3u, f = (row.z >> 2) & 31u;
    if ((row.z >> 12) == MSIM_PROCESS_NEMESIS || type == MSIM_T_INVOKE) continue;
    if (f == MSIM_F_SEND) { hi = max(hi, (row.w >> 6) & 0x7FFu); if (type == MSIM_T_OK && (row.w >> 17) != 0x7FFu) hi = max(hi, row.w >> 17); }
    else if (f == MSIM_F_POLL && type == MSIM_T_OK) {
      const u32 len = row.y >> 16;
      if (len == 0 || (u64)row.w + len > n_words) continue;
      const u32 *const w = pay + row.w;
      for (u32 q = 0; q < len;) {
        const u32 h = w[q++], cnt = (h >> 8) & 0xFFu, o0 = h >> 16;
        if (q + (cnt + 1) / 2 > len) break;
        if (cnt) hi = max(hi, o0 + cnt - 1u);
        for (u32 e = 0; e < cnt; e++) hi = max(hi, (w[q + e / 2] >> (16 * (e & 1))) & 0xFFFFu);
        q += (cnt + 1) / 2;
      }
    }
  }
  for (int o = 32; o; o >>= 1) hi = max(hi, (u32)__shfl_xor((int)hi, o));
  if ((tid & 63u) == 0 && hi) atomicMax(bound, hi);
}

int kafka_dev_run(msim_ctx *ctx, KCParams kp, u32 n, const std::vector<msim_inst_meta> *hmeta, msim_check_result *h_out, hipStream_t st, u32 *n_host) {
  const bool trace = (msim_dev_flags(ctx) & 0x1000u) != 0;   // developer: time the passes
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  {   // tables no larger than the launch's histories need (kp.T: what the configuration / the checker's own bound allows)
    u32 *d_bound = nullptr, h_bound = 0;
    MSIM_HIP_TRY(ctx, msim_dev_malloc(&d_bound, 4));
    hipError_t e = hipMemsetAsync(d_bound, 0, 4, st);
    if (e == hipSuccess) { hipLaunchKernelGGL(kafka_bound_kernel, dim3(n), dim3(NT), 0, st, kp, d_bound); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bound, d_bound, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)msim_dev_free(d_bound);
    MSIM_HIP_TRY(ctx, e);
    const u32 t = ((h_bound + 1u + 31u) & ~31u);
    if (t < kp.T) kp.T = t < 32u ? 32u : t;
  }
  const size_t lds = kafka_lds_bytes(kp.T);
  if (lds > 64 * 1024) MSIM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&kafka_check_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kafka_check_kernel, dim3(n), dim3(NT), lds, st, kp);
  MSIM_HIP_TRY(ctx, hipGetLastError());
  MSIM_HIP_TRY(ctx, hipMemcpyAsync(h_out, kp.out, (size_t)n * sizeof(msim_check_result), hipMemcpyDeviceToHost, st));
  MSIM_HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<u32> todo;
  for (u32 i = 0; i < n; i++) if (h_out[i].valid == NEEDS_HOST) todo.push_back(i);
  if (trace) std::fprintf(stderr, "[kafka-check] device pass (%zu B of LDS per history): %.2f ms, %zu of %u histories for the host\n", lds, ms(), todo.size(), n);
  if (!todo.empty()) {
    std::vector<uint64_t> ro, po;
    if (kp.row_off) { ro.resize(n + 1); po.resize(n + 1);
      MSIM_HIP_TRY(ctx, hipMemcpy(ro.data(), kp.row_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
      MSIM_HIP_TRY(ctx, hipMemcpy(po.data(), kp.pay_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost)); }
    std::vector<std::vector<msim_op>> rows(todo.size());
    std::vector<std::vector<u32>> pays(todo.size());
    for (size_t k = 0; k < todo.size(); k++) {
      const u32 i = todo[k];
      const u32 nr = hmeta ? (*hmeta)[i].n_rows : (u32)(ro[i + 1] - ro[i]), nw = hmeta ? (*hmeta)[i].n_payload_words : (u32)(po[i + 1] - po[i]);
      rows[k].resize(nr ? nr : 1); pays[k].resize(nw ? nw : 1);
      if (nr) MSIM_HIP_TRY(ctx, hipMemcpy(rows[k].data(), kp.rows + (hmeta ? (uint64_t)i * kp.max_rows : ro[i]), (size_t)nr * sizeof(msim_op), hipMemcpyDeviceToHost));
      if (nw) MSIM_HIP_TRY(ctx, hipMemcpy(pays[k].data(), kp.payload + (hmeta ? (uint64_t)i * kp.max_pay : po[i]), (size_t)nw * 4, hipMemcpyDeviceToHost));
    }
    unsigned nt = msim_host_threads();
    if (nt > todo.size()) nt = (unsigned)todo.size();
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; w++)
      th.emplace_back([&, w]() {
        for (size_t k = w; k < todo.size(); k += nt) {
          const u32 i = todo[k];
          msim_kafka_check_instance_host(rows[k].data(), hmeta ? (*hmeta)[i].n_rows : (u32)(ro[i + 1] - ro[i]), pays[k].data(),
                                         hmeta ? (*hmeta)[i].n_payload_words : (u32)(po[i + 1] - po[i]), hmeta ? (*hmeta)[i].flags : 0u, &h_out[i]);
        }
      });
    for (auto &x : th) x.join();
    for (u32 i : todo) MSIM_HIP_TRY(ctx, hipMemcpy(kp.out + i, &h_out[i], sizeof(msim_check_result), hipMemcpyHostToDevice));
    if (trace) std::fprintf(stderr, "[kafka-check] host checker on those: done at %.2f ms\n", ms());
  }
  if (n_host) *n_host = (u32)todo.size();
  return MSIM_OK;
}

// table entries per key that cover offsets / messages below `bound`
u32 kafka_table_entries(u32 bound) { const u32 t = (bound + 31u) & ~31u; return t < 32u ? 32u : (t > OFFS ? OFFS : t); }

}  // namespace

// msim_check for kafka: the histories of the last run, where they lie in HBM.
int msim_check_kafka_device(msim_ctx *ctx) {
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const u32 n = ctx->n_inst;
  if (ctx->cfg.concurrency == 0 || ctx->cfg.concurrency > CMAX) return msim_check_kafka_host(ctx);
  if (ctx->h_check) { (void)hipHostFree(ctx->h_check); ctx->h_check = nullptr; }
  MSIM_HIP_TRY(ctx, hipHostMalloc(&ctx->h_check, (size_t)n * sizeof(msim_check_result)));
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<msim_inst_meta> hm(n);
  MSIM_HIP_TRY(ctx, hipMemcpy(hm.data(), ctx->d_meta, (size_t)n * sizeof(msim_inst_meta), hipMemcpyDeviceToHost));
  KCParams kp;
  std::memset(&kp, 0, sizeof kp);
  kp.rows = ctx->d_rows; kp.payload = ctx->d_payload; kp.meta = ctx->d_meta; kp.out = ctx->d_check;
  kp.max_rows = ctx->cfg.max_rows; kp.max_pay = ctx->cfg.max_payload_words;
  // a key takes max-writes-per-key sends before it is retired: its offsets and messages stay below that (+ slack: what exceeds the
  // tables goes to the host checker, whose own bound is 2048)
  kp.T = kafka_table_entries(ctx->cfg.max_writes_per_key + 8u);
  kp.C = ctx->cfg.concurrency;
  u32 redone = 0;
  int rc = kafka_dev_run(ctx, kp, n, &hm, ctx->h_check, ctx->stream, &redone);
  if (rc != MSIM_OK) return rc;
  ctx->check_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->lin_host_rechecks = redone;
  ctx->checked = true; ctx->check_fetched = true;
  return MSIM_OK;
}

// Checks `n_histories` kafka histories given on the host (rows / payload words of history i at row_offsets[i] / payload_offsets[i]) as
// msim_check does for the histories of a run; `concurrency` = the worker threads of the test (process mod concurrency).
extern "C" int msim_check_kafka_batch(int device, const msim_op *rows, const uint64_t *row_offsets, const uint32_t *payload, const uint64_t *payload_offsets,
                                      uint32_t n_histories, uint32_t concurrency, msim_check_result *out, uint32_t *n_host) {
  if (!rows || !row_offsets || !payload_offsets || !out || n_histories == 0 || concurrency == 0 || concurrency > CMAX) return MSIM_E_INVALID;
  const uint64_t tr = row_offsets[n_histories], tw = payload_offsets[n_histories];
  if (tw != 0 && !payload) return MSIM_E_INVALID;   // (as msim_check_kafka_rows: payload words announced, none given)
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_HIP;
  msim_ctx tmp_ctx; msim_ctx *ctx = &tmp_ctx;   // only for error text
  tmp_ctx.device = device;
  for (u32 i = 0; i < n_histories; i++) if (row_offsets[i + 1] - row_offsets[i] > 0x7FFFFFFFull || payload_offsets[i + 1] - payload_offsets[i] > 0x7FFFFFFFull) return MSIM_E_RANGE;
  msim_op *d_rows = nullptr; u32 *d_pay = nullptr; uint64_t *d_ro = nullptr, *d_po = nullptr; msim_check_result *d_out = nullptr;
  int rc = MSIM_E_HIP;
  do {
    if (msim_dev_malloc(&d_rows, (size_t)(tr ? tr : 1) * sizeof(msim_op)) != hipSuccess) break;
    if (msim_dev_malloc(&d_pay, (size_t)(tw ? tw : 1) * 4) != hipSuccess) break;
    if (msim_dev_malloc(&d_ro, (size_t)(n_histories + 1) * 8) != hipSuccess || msim_dev_malloc(&d_po, (size_t)(n_histories + 1) * 8) != hipSuccess) break;
    if (msim_dev_malloc(&d_out, (size_t)n_histories * sizeof(msim_check_result)) != hipSuccess) break;
    if (tr && hipMemcpy(d_rows, rows, (size_t)tr * sizeof(msim_op), hipMemcpyHostToDevice) != hipSuccess) break;
    if (tw && payload && hipMemcpy(d_pay, payload, (size_t)tw * 4, hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_ro, row_offsets, (size_t)(n_histories + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_po, payload_offsets, (size_t)(n_histories + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
    KCParams kp;
    std::memset(&kp, 0, sizeof kp);
    kp.rows = d_rows; kp.payload = d_pay; kp.row_off = d_ro; kp.pay_off = d_po; kp.out = d_out;
    kp.T = OFFS; kp.C = concurrency;
    rc = kafka_dev_run(ctx, kp, n_histories, nullptr, out, nullptr, n_host);
  } while (false);
  for (void *q : {(void *)d_rows, (void *)d_pay, (void *)d_ro, (void *)d_po, (void *)d_out}) if (q) (void)msim_dev_free(q);
  return rc;
}
