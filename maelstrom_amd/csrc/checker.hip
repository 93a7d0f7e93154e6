// checker.hip — workload checkers at ensemble scale, on the device.
//
// The reference hands each history to `(checker/check (:checker test) test history opts)` (core.clj:91-100).
// For broadcast / g-set that is jepsen's `checker/set-full` (workload/broadcast.clj:216-228 maps
// :broadcast -> :add; workload/g_set.clj:62); for echo the pair comparison of workload/echo.clj:44-63;
// checker/stats counts ok/fail/info.  [upstream] jepsen.checker/set-full is restated here from its
// published algorithm (result-map shape: doc/03-broadcast/01-broadcast.md:564-577):
//   per element: known        = first :ok add completion or first :ok read containing it
//                last-present = latest-invoked :ok read containing it (by invoke :index)
//                last-absent  = latest-invoked :ok read (completed after the add's invoke) lacking it
//   stable  <=> last-present and index(last-absent, -1) < index(last-present)
//   lost    <=> known and last-absent and index(last-present, -1) < index(last-absent) and index(known) < index(last-absent)
//   stable-latency = long(nanos->ms(max(0, (time(last-absent)+1 | 0) - time(known))));  stale <=> stable-latency > 0
//   :valid? = false if any lost, :unknown if nothing stable, else true
//   :stable-latencies = points {0 .5 .95 .99 1} -> sorted[min(n-1, floor(n*q))]
//
// One wavefront checks one history, bit-parallel over the read bitmaps:
//   pass 1  rows are read back from HBM 64 at a time (1 KiB, coalesced, the next chunk in flight); every lane classifies its own
//           row.  A completion is paired with its invocation — the previous row of the same worker thread — without a serial walk
//           (round 2 walked the read rows one by one with v_readlane: 2/3 of the kernel's instructions were scalar): the read rows of
//           a chunk are taken in rounds, in each round the EARLIEST pending row of every worker thread (ds_min over the thread's
//           slot) acts on the thread's table entry in LDS — an invocation leaves {row, rank}, a completion takes it — so a chunk costs
//           as many rounds as its busiest thread has read rows in it (2-3), not one step per row.  A read :ok RECORDS itself at the
//           rank of its invocation {payload ref, invoke index, elements existing at completion, :ok index} (12 B, HBM scratch).  Four chunks
//           of rows are in flight.
//   pass 2  ONE sweep over the recorded reads in invocation order, 64 reads at a time, lane r = read r of the chunk (round 6; before: lane w =
//           word w of one read's bitmap at a time).  Every lane loads its own read's bitmap 16 bytes at a time; per word the highest / lowest
//           lane with a bit decides all its bits at once, words every read of the chunk holds completely are only noted — see the comment
//           at the sweep.
// Per-element state: three u16 row indices in LDS (8.4 KB for 1408 elements keeps 4096 histories resident).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "engine_internal.h"

#define NONE 0xFFFFu      /* per-element row index: none */
#define NONE32 0xFFFFFFFFu /* per-thread table entry: no invocation pending */
#define NOLAT 0xFFFFFFFFu  /* per-element latency: not stable */

struct CParams {
  const msim_op *rows;
  const u32 *payload;
  const msim_inst_meta *meta;
  msim_check_result *out;
  u32 *recs;    // per instance: max_reads records of 3 words, at the rank of the read's invocation
  u32 max_rows, max_pay, max_values, C, workload, max_reads;
  u32 off_valid, off_tbl;   // LDS offsets (bytes): the valid bitmap, the per-thread tables
  float rcp_C;
};

#ifdef MSIM_HIPEMU
struct __attribute__((packed, aligned(4))) ck_u32x4 { u32 x, y, z, w; };
#else
typedef u32 ck_u32x4 __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes of a u32 array from any word on: global memory takes one dwordx4 at a 4-byte boundary
#endif
__device__ __forceinline__ u32 c_rdlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u32 c_wave_sum(u32 v) {
  for (int o = 32; o; o >>= 1) v += (u32)__shfl_xor((int)v, o);
  return v;
}

// maximum over the 64 lanes (uniform) / inclusive prefix maximum, by DPP
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ u32 c_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, BOUND); }
__device__ __forceinline__ u32 c_wave_incl_max(u32 v) {
  v = max(v, c_dpp<0x111, 0xF, true>(v));   // row_shr:1 (lanes without a source read 0)
  v = max(v, c_dpp<0x112, 0xF, true>(v));   // row_shr:2
  v = max(v, c_dpp<0x114, 0xF, true>(v));   // row_shr:4
  v = max(v, c_dpp<0x118, 0xF, true>(v));   // row_shr:8
  v = max(v, c_dpp<0x142, 0xA, false>(v));  // row_bcast:15 -> rows 1, 3
  v = max(v, c_dpp<0x143, 0xC, false>(v));  // row_bcast:31 -> rows 2, 3
  return v;
}
// OR over the lanes BELOW this one
__device__ __forceinline__ u32 c_wave_excl_or(u32 v) {
  v = c_dpp<0x138, 0xF, true>(v);           // wave_shr:1 (lane 0 reads 0)
  v |= c_dpp<0x111, 0xF, true>(v);
  v |= c_dpp<0x112, 0xF, true>(v);
  v |= c_dpp<0x114, 0xF, true>(v);
  v |= c_dpp<0x118, 0xF, true>(v);
  v |= c_dpp<0x142, 0xA, false>(v);
  v |= c_dpp<0x143, 0xC, false>(v);
  return v;
}
// orders this wavefront's LDS traffic across its lanes (LDS operations of one wavefront execute in order: the compiler must not move them)
__device__ __forceinline__ void c_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32 c_wave_max(u32 v) { return c_rdlane(c_wave_incl_max(v), 63); }

__global__ void __launch_bounds__(64) check_kernel(const CParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char csmem[];
  typedef unsigned short u16;
  u16 *const known = reinterpret_cast<u16 *>(csmem);
  u16 *const lp_idx = known + p.max_values;
  u16 *const la_idx = lp_idx + p.max_values;
  u32 *const lat = reinterpret_cast<u32 *>(csmem);  // reused after the walk: stable latency per element (needs 4 B each)
  u32 *const valid = reinterpret_cast<u32 *>(csmem + p.off_valid);  // bitmap over invocation ranks: read completed :ok
  u32 *const slot = reinterpret_cast<u32 *>(csmem + p.off_tbl);     // [C] the worker thread's pending invocation: row | rank << 16 (echo: row)
  u32 *const first = slot + p.C;                                     // [C] ds_min target: (round, lane) of the thread's earliest pending row
  u32 *const slotv = first + p.C;                                    // [C] echo: the :value of the thread's last invocation

  const u32 lane = threadIdx.x, inst = blockIdx.x;
  const msim_inst_meta meta = p.meta[inst];
  const uint4 *const rows = reinterpret_cast<const uint4 *>(p.rows + (size_t)inst * p.max_rows);
  const u32 *const pay = p.payload + (size_t)inst * p.max_pay;
  u32 *const rec = p.recs + (size_t)inst * p.max_reads * 3;
  const u32 n_rows = meta.n_rows, C = p.C;
  const bool setfull = p.workload != MSIM_WL_ECHO;
  const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;   // lanes below this one

  for (u32 i = lane; i < p.max_values; i += 64) { known[i] = NONE; lp_idx[i] = NONE; la_idx[i] = NONE; }
  for (u32 i = lane; i < (p.max_reads + 31) / 32 + 2; i += 64) valid[i] = 0;
  for (u32 i = lane; i < C; i += 64) { slot[i] = NONE32; first[i] = 0xFFFFFFFFu; slotv[i] = 0; }
  __syncthreads();

  u32 v_cur = 0, n_ri = 0, errors = 0;
#ifdef CK_PROF   // developer build (tools/check_prof_report.py): cycles of pass 1 / pass 2 / the rest in stable_latency_ms[0..2], pairing rounds in [3], reads in [4]
  const u64 ck_t0 = __builtin_readcyclecounter(); u64 ck_t1 = 0, ck_t2 = 0; u32 ck_rounds = 0;
#endif
  u32 cnt_lo = 0, cnt_hi = 0;    // per lane: rows of type :invoke | :ok << 16, :fail | :info << 16 (summed over the wavefront at the end)
  u32 round_key = 0x3FFFFFFu;    // decreases with every pairing round: a later round's ds_min beats whatever an earlier one left

  // ---- pass 1 ----
  // Four chunks of rows (4 KiB) are in flight: with one, a chunk's 150 instructions waited out an HBM round trip each time (round 6: pass 1 took
  // 1.7e5 cycles per history, 2700 per chunk).  The buffers keep their registers — a rotation by moves would wait for the load just issued —
  // and the loads are unconditional (past the end: the last row again; every use of a row is behind `lane < cnt`).
  auto ldrow = [&](u32 b) -> uint4 { return rows[min(b + lane, n_rows ? n_rows - 1 : 0u)]; };
  auto rowchunk = [&](const u32 base, const uint4 r) {
    const u32 cnt = min(64u, n_rows - base);
    const u32 my_type = r.z & 3, my_f = (r.z >> 2) & 31, my_proc = r.z >> 12, idx = base + lane;
    const bool live = lane < cnt && my_proc != MSIM_PROCESS_NEMESIS;  // (r/filter (comp number? :process))
    const bool add_like = my_f == MSIM_F_ADD || my_f == MSIM_F_BROADCAST;
    if (live) { const u32 one = 1u << ((my_type & 1) * 16); if (my_type & 2) cnt_hi += one; else cnt_lo += one; }
    // add :ok -> known (first of add-ok / first containing read, by :index): order-free as a minimum; one add per element
    if (live && add_like && my_type == MSIM_T_OK && r.w < p.max_values && known[r.w] > idx) known[r.w] = (u16)idx;
    const u64 add_inv = __ballot(live && add_like && my_type == MSIM_T_INVOKE);  // elements come into existence
    const u32 v_here = v_cur + (u32)__popcll(add_inv & lt);                      // elements existing at this row
    v_cur += (u32)__popcll(add_inv);  // values are handed out 0,1,2,... in invoke order
    const bool is_r = live && my_f == (setfull ? MSIM_F_READ : MSIM_F_ECHO);
    const u64 ri = __ballot(is_r && my_type == MSIM_T_INVOKE);
    const u32 rank = n_ri + (u32)__popcll(ri & lt);                                // invocation order of the reads
    n_ri += (u32)__popcll(ri);
    // what a read :ok leaves behind: its record at the rank of its invocation `sv` = row | rank << 16
    auto record = [&](u32 sv) {
      const u32 rk = sv >> 16;
      if (rk < p.max_reads) {
        u32 *q = rec + (size_t)rk * 3;
        q[0] = r.w | ((r.y >> 16) << 24); q[1] = (sv & 0xFFFFu) | (v_here << 16); q[2] = idx;
        atomicOr(&valid[rk >> 5], 1u << (rk & 31));
      }
    };
    // A completion that directly follows its invocation (all but a few: the rows of one operation are only separated when another
    // worker thread's row falls between them) is settled from the lane below; the pair never touches the thread's table entry.
    // (Same PROCESS, which is the same worker thread for the two rows of one operation; the invocation's rank is this lane's count of the
    //  invocations below it, less one.  Rows reach the lane above by DPP: the packed word, and for echo the value.)
    const u32 below_z = c_dpp<0x138, 0xF, true>(r.z);   // wave_shr:1
    const u32 below_v = setfull ? 0u : c_dpp<0x138, 0xF, true>(r.w);
    const bool adj = is_r && my_type != MSIM_T_INVOKE && lane > 0 && (below_z & 3) == MSIM_T_INVOKE && ((below_z ^ r.z) & 0xFFFFF07Cu) == 0;
    const u64 adj_m = __ballot(adj);
    if (adj) {
      if (setfull) { if (my_type == MSIM_T_OK) record((idx - 1) | ((rank - 1) << 16)); }
      // echo.clj:44-63: every :invoke whose completion is not an :ok carrying the same :echo is an error — a :fail, an
      // :info (its :value is the request string, (:echo "...") = nil) and an invocation that never completes included
      else if (my_type != MSIM_T_OK || below_v != r.w) errors++;
    }
    bool pend = is_r && !adj && !((adj_m >> 1 >> lane) & 1);
    u64 pm = __ballot(pend);
    // worker thread = process mod C ([upstream] jepsen's interpreter): one float multiply and a correction instead of a division
    u32 tt = 0;
    if (pm && pend) {
      const u32 qd = (u32)((float)my_proc * p.rcp_C);
      int rem = (int)(my_proc - qd * C);
      if (rem < 0) rem += (int)C;
      if (rem >= (int)C) rem -= (int)C;
      tt = (u32)rem;
    }
    while (pm) {   // rounds: the earliest pending read row of every worker thread acts on the thread's table entry
      const u32 key = (round_key << 6) | lane;
      if (pend) atomicMin(&first[tt], key);
      c_lds_fence();   // (one wavefront: LDS is in order; __syncthreads would also wait for the rows in flight)
      if (pend && first[tt] == key) {
        pend = false;
        const u32 s = slot[tt];
        if (my_type == MSIM_T_INVOKE) {
          slot[tt] = setfull ? (idx | (rank << 16)) : idx;
          if (!setfull) slotv[tt] = r.w;
        } else {
          slot[tt] = NONE32;
          if (setfull) { if (my_type == MSIM_T_OK && s != NONE32) record(s); }
          else if (my_type != MSIM_T_OK || slotv[tt] != r.w) errors++;
        }
      }
      round_key--;
      c_lds_fence();   // (one wavefront: LDS is in order; __syncthreads would also wait for the rows in flight)
      pm = __ballot(pend);
#ifdef CK_PROF
      ck_rounds++;
#endif
    }
  };
  if (n_rows) {
    uint4 q0 = ldrow(0), q1 = ldrow(64), q2 = ldrow(128), q3 = ldrow(192);
    for (u32 base = 0; base < n_rows; base += 256) {
      rowchunk(base, q0); q0 = ldrow(base + 256);
      if (base + 64 >= n_rows) break;
      rowchunk(base + 64, q1); q1 = ldrow(base + 320);
      if (base + 128 >= n_rows) break;
      rowchunk(base + 128, q2); q2 = ldrow(base + 384);
      if (base + 192 >= n_rows) break;
      rowchunk(base + 192, q3); q3 = ldrow(base + 448);
    }
  }
  if (!setfull) { for (u32 i = lane; i < C; i += 64) errors += slot[i] != NONE32 ? 1u : 0u; }   // invocations left without a completion
  errors = c_wave_sum(errors);
  cnt_lo = c_wave_sum(cnt_lo & 0xFFFFu) | (c_wave_sum(cnt_lo >> 16) << 16);   // (at most 65534 rows: every count fits 16 bits)
  cnt_hi = c_wave_sum(cnt_hi & 0xFFFFu) | (c_wave_sum(cnt_hi >> 16) << 16);
  const u32 op_count = cnt_lo & 0xFFFFu, n_ok = cnt_lo >> 16, n_fail = cnt_hi & 0xFFFFu, n_info = cnt_hi >> 16;
  __threadfence_block();
  __syncthreads();
  if (n_ri > p.max_reads) n_ri = p.max_reads;
#ifdef CK_PROF
  ck_t1 = __builtin_readcyclecounter();
#endif

  // ---- pass 2: ONE sweep over the :ok reads in invocation order, 64 reads at a time, LANE r = READ r of the chunk ----------------------------
  // Per element three reductions over the reads (§ header): last-present = the LAST read that has it, last-absent = the LAST read that lacks
  // it while it exists, known = the smallest :ok index among the reads that have it.  Rounds 3-6 swept with lane w = word w of ONE read's
  // bitmap at a time: a read cost ~140 issued instructions whatever it changed (its rank, three readlanes, the load, the masks, three bit
  // loops), half the lanes idle (17 words on average).  Now a lane holds its OWN read's bitmap (16-byte loads at the read's own address, three
  // in flight) and the wavefront walks the words 0 .. max words of the chunk; for word w
  //   last-present: the highest lane with a bit decides ALL its bits at once (one 32-lane store, lane b = element 32w + b); what is left
  //     (bits some lower lane has and that one has not: only at the frontier of the elements still spreading) takes the next-highest lane with
  //     one of them, and so on.  Later chunks overwrite: the last chunk's last read wins.  last-absent likewise over ~word & existing
  //     (nothing to do for words whose elements every read of the chunk holds);
  //   known: the FIRST lane that has a not-yet-seen element gives its :ok index (lowest lane first, all its new bits at once) — right unless a
  //     read invoked later completed earlier; such a read has an :ok index below the maximum of the reads before it, so every lane of which
  //     that is true (`nm`: rare) takes the minimum over ALL its bits — which is always allowed, `known` being a minimum over containing reads.
  // Element 32w + b is always written by lane b: no cross-lane ordering in LDS.
  {
    const u32 sub = lane & 31u;
    uint4 tiny = make_uint4(0, 0, 0, 0);   // a payload slab of less than four words, whole
    if (p.max_pay < 4) { if (p.max_pay > 0) tiny.x = pay[0]; if (p.max_pay > 1) tiny.y = pay[1]; if (p.max_pay > 2) tiny.z = pay[2]; }
    u32 unkv = 0xFFFFFFFFu;          // lane j: the elements of word j no read has contained yet
    u64 unk_any = ~0ull;             // wave-uniform: the lanes of unkv that are not zero
    u32 pend = NONE;                 // lane j: word j was SETTLED in the last chunk that had it (below) — last-present of its 32 elements, not yet written
    u32 ok_carry = 0;                // the largest :ok index of the chunks before
    auto chunk_mask = [&](u32 cb) -> u64 {
      if (cb >= n_ri) return 0ull;
      const u32 cn = min(64u, n_ri - cb);
      return ((u64)valid[cb / 32 + 1] << 32 | valid[cb / 32]) & (cn >= 64 ? ~0ull : ((1ull << cn) - 1));
    };
    // the records of the next chunk are requested while this one is swept
    u32 nx = 0, ny = 0, nz = 0;
    u64 nmask = chunk_mask(0);
    if ((nmask >> lane) & 1) { const u32 *q = rec + (size_t)lane * 3; nx = q[0]; ny = q[1]; nz = q[2]; }
    for (u32 cb = 0; cb < n_ri; cb += 64) {
      const u64 vmask = nmask;
      const u32 rx = nx, ry = ny, rz = nz;   // {payload ref | words << 24, invoke index | elements existing at completion << 16, :ok index}; 0 for a rank without an :ok
      nmask = chunk_mask(cb + 64);
      nx = ny = nz = 0;
      if ((nmask >> lane) & 1) { const u32 *q = rec + (size_t)(cb + 64 + lane) * 3; nx = q[0]; ny = q[1]; nz = q[2]; }
      if (!vmask) continue;
      const u32 ref = rx & 0xFFFFFFu, nw = rx >> 24, inv = ry & 0xFFFFu, v_here = ry >> 16, ok = rz;
      // four words of this lane's bitmap from word 4g on: ONE unconditional 16-byte load (a lane with nothing there reads the slab's first
      // words and ignores them: no branch around the load, so the loads in flight stay countable).  A bitmap that ends in the slab's last
      // three words — one instance in thousands — reads the slab's last four words and shifts.  (A slab of less than four words holds no bitmap
      // worth the name: it is read once, before the sweep.)
      auto ld = [&](u32 g) -> uint4 {
        const u32 at = ref + 4 * g;
        const bool in = 4 * g < nw;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p.max_pay >= 4) {
          const u32 last4 = p.max_pay - 4;
          const ck_u32x4 q = *reinterpret_cast<const ck_u32x4 *>(pay + (in ? min(at, last4) : 0u));   // (16 bytes at a 4-byte boundary)
          v = in ? make_uint4(q.x, q.y, q.z, q.w) : make_uint4(0, 0, 0, 0);   // (a lane without a word in this group — a rank without a record among them — holds zeros)
          if (__ballot(in && at > last4)) {
            const u32 sh = in && at > last4 ? at - last4 : 0u;
            if (sh == 1) v = make_uint4(v.y, v.z, v.w, 0); else if (sh == 2) v = make_uint4(v.z, v.w, 0, 0); else if (sh >= 3) v = make_uint4(sh == 3 ? v.w : 0u, 0, 0, 0);
          }
        } else if (in) {   // (no load here: a load on this path would make the loads in flight uncountable for the path that matters)
          auto tw = [&](u32 i) { return i == 0 ? tiny.x : (i == 1 ? tiny.y : (i == 2 ? tiny.z : 0u)); };
          v = make_uint4(tw(at), tw(at + 1), tw(at + 2), 0);
        }
        return v;
      };
      uint4 b0 = ld(0), b1 = ld(1), b2 = ld(2), b3 = ld(3);
      const u32 maxw = min(64u, c_wave_max(max(nw, (v_here + 31) >> 5)));   // (a bitmap may be shorter than the elements that exist: they are absent)
      // lanes whose :ok index is below the largest of the reads before them: inclusive prefix maximum, then the lane below's
      const u32 pm = c_wave_incl_max(ok);
      u32 before = (u32)__shfl_up((int)pm, 1);
      before = max(lane ? before : 0u, ok_carry);
      const u64 nm_mask = __ballot(((vmask >> lane) & 1) && ok < before);
      ok_carry = max(ok_carry, c_rdlane(pm, 63));
      const u32 inv_top = c_rdlane(inv, 63u - (u32)__builtin_clzll(vmask));
      const u32 minw = ~c_wave_max(((vmask >> lane) & 1) ? ~nw : 0u);   // every read of the chunk has at least this many words
      // one word of the 64 bitmaps.  What the loops decide for element 32w + b collects in lane b's registers; LDS is touched once per word and kind
      auto word = [&](const u32 w, const u32 raw) {
        u32 W = raw;
        if (w >= minw) W = w < nw ? raw : 0u;   // (wave-uniform test: below minw every read of the chunk has the word, and a lane without a read holds zeros)
        u32 unk_w = (unk_any >> w) & 1 ? c_rdlane(unkv, w) : 0u;
        // A SETTLED word — every read of the chunk holds all 32 elements, all of them seen before, no overtaker among the reads: most words,
        // the elements behind the frontier — changes one thing: last-present = the chunk's last read, for all 32.  That is noted in `pend`
        // and written when a chunk finds the word unsettled, or at the end.
        if (__ballot(W == 0xFFFFFFFFu) == vmask && unk_w == 0 && nm_mask == 0) { if (lane == w) pend = inv_top; return; }
        const u32 e = 32 * w + sub;
        const bool e_ok = lane < 32 && e < p.max_values;
        const u32 pw = c_rdlane(pend, w);
        if (pw != NONE) { if (e_ok) lp_idx[e] = (u16)pw; if (lane == w) pend = NONE; }
        const int d = (int)v_here - (int)(32 * w);
        const u32 ex = d >= 32 ? 0xFFFFFFFFu : (d <= 0 ? 0u : ((1u << d) - 1));  // elements that exist at this read
        const u32 A = ~W & ex;
        u64 m = __ballot(W != 0);
        const u64 holders = m;
        if (m) {   // last-present
          u32 rem = 0xFFFFFFFFu, got = 0;
          do {
            const u32 L = 63u - (u32)__builtin_clzll(m);
            const u32 bits = c_rdlane(W, L) & rem, iv = c_rdlane(inv, L);
            got = ((bits >> sub) & 1) ? iv : got;
            rem &= ~bits;
            m = __ballot((W & rem) != 0);
          } while (m);
          if (e_ok && !((rem >> sub) & 1)) lp_idx[e] = (u16)got;
        }
        m = __ballot(A != 0);
        if (m) {   // last-absent
          u32 rem = 0xFFFFFFFFu, got = 0;
          do {
            const u32 L = 63u - (u32)__builtin_clzll(m);
            const u32 bits = c_rdlane(A, L) & rem, iv = c_rdlane(inv, L);
            got = ((bits >> sub) & 1) ? iv : got;
            rem &= ~bits;
            m = __ballot((A & rem) != 0);
          } while (m);
          if (e_ok && !((rem >> sub) & 1)) la_idx[e] = (u16)got;
        }
        if (unk_w) {   // known: the first read that holds a not-yet-seen element gives its :ok index.  Every lane takes the bits NO LOWER lane
          const u32 nw_bits = W & unk_w;   // holds (a prefix OR across the lanes) and writes them itself: as many turns as ONE read has new elements
          if (__ballot(nw_bits != 0)) {
            const u32 below = c_wave_excl_or(nw_bits);
            u32 mine = nw_bits & ~below;
            while (mine) {
              const u32 x = 32 * w + (u32)__builtin_ctz(mine); mine &= mine - 1;
              if (x < p.max_values && known[x] > ok) known[x] = (u16)ok;
            }
            c_lds_fence();   // (the elements of word w are lane b's again below)
            unk_w &= ~(c_rdlane(below, 63) | c_rdlane(nw_bits, 63));   // everything some lane holds
            if (lane == w) unkv = unk_w;
            unk_any = __ballot(unkv != 0);
          }
        }
        u32 kmin = NONE;
        bool any = false;   // wave-uniform
        m = nm_mask & holders;
        if (m) {   // ... and the reads that overtook an earlier one
          any = true;
          do {
            const u32 L = (u32)__builtin_ctzll(m); m &= m - 1;
            const u32 bits = c_rdlane(W, L), okl = c_rdlane(ok, L);
            kmin = ((bits >> sub) & 1) ? min(kmin, okl) : kmin;
          } while (m);
        }
        if (any && e_ok && known[e] > kmin) known[e] = (u16)kmin;
      };
      auto group = [&](const u32 g, const uint4 &c) {
        // four settled words at once (most groups: everything well behind the frontier)
        if (4 * g + 3 < minw && nm_mask == 0 && ((unk_any >> (4 * g)) & 0xFu) == 0 && __ballot((c.x & c.y & c.z & c.w) == 0xFFFFFFFFu) == vmask) {
          if ((lane >> 2) == g) pend = inv_top;
          return;
        }
        word(4 * g, c.x);
        if (4 * g + 1 < maxw) word(4 * g + 1, c.y);
        if (4 * g + 2 < maxw) word(4 * g + 2, c.z);
        if (4 * g + 3 < maxw) word(4 * g + 3, c.w);
      };
      // four 16-byte loads per lane in flight; the buffers keep their registers (a rotation by moves would wait for the load just issued)
      const u32 ng = (maxw + 3) >> 2;
      for (u32 g = 0; g < ng; g += 4) {
        group(g, b0); if (g + 1 >= ng) break; b0 = ld(g + 4);
        group(g + 1, b1); if (g + 2 >= ng) break; b1 = ld(g + 5);
        group(g + 2, b2); if (g + 3 >= ng) break; b2 = ld(g + 6);
        group(g + 3, b3); b3 = ld(g + 7);
      }
    }
    for (u64 m = __ballot(pend != NONE); m; m &= m - 1) {   // the words that were settled to the end
      const u32 w = (u32)__builtin_ctzll(m), pw = c_rdlane(pend, w), e = 32 * w + sub;
      if (lane < 32 && e < p.max_values) lp_idx[e] = (u16)pw;
    }
  }
  __syncthreads();

#ifdef CK_PROF
  ck_t2 = __builtin_readcyclecounter();
#endif
  // ---- per-element outcomes (each lane owns elements e = lane, lane+64, ...; at most 32 per lane) ----
  u32 c_stable = 0, c_lost = 0, c_never = 0, c_stale = 0;
  u32 my_lat[32];  // statically indexed (unrolled): stays in VGPRs
#pragma unroll
  for (u32 k = 0; k < 32; k++) {
    const u32 e = lane + 64 * k;
    u32 l = NOLAT;
    if (e < v_cur) {
      const u32 kn = known[e], lp = lp_idx[e], la = la_idx[e];
      const bool stable = lp != NONE && (la == NONE || la < lp);
      const bool lost = kn != NONE && la != NONE && (lp == NONE || lp < la) && kn < la;
      if (stable) {
        l = 0;
        if (la != NONE) {
          const uint4 ra = rows[la], rk = rows[kn];
          const u64 ta = (((u64)(ra.y & 0xFFFF) << 32) | ra.x) + 1, tk = ((u64)(rk.y & 0xFFFF) << 32) | rk.x;
          if (ta > tk) l = (u32)((ta - tk) / 1000000ull);
        }
        c_stable++; if (l > 0) c_stale++;
      } else if (lost) c_lost++; else c_never++;
    }
    my_lat[k] = l;
  }
  __syncthreads();  // everyone has read the u16 arrays: the same LDS now holds the latencies
#pragma unroll
  for (u32 k = 0; k < 32; k++) { const u32 e = lane + 64 * k; if (e < v_cur) lat[e] = my_lat[k]; }
  __syncthreads();
  const u32 n_stable = c_wave_sum(c_stable), n_lost = c_wave_sum(c_lost), n_never = c_wave_sum(c_never), n_stale = c_wave_sum(c_stale);

  // ---- quantiles of the stable latencies: idx-th smallest by bisection on the value ----
  u32 q[5] = {0, 0, 0, 0, 0};
  if (n_stable) {
    u32 lmax = 0;  // the search range is [0, largest stable latency]
    for (u32 e = lane; e < v_cur; e += 64) { const u32 l = lat[e]; if (l != NOLAT) lmax = max(lmax, l); }
    for (int o = 32; o; o >>= 1) lmax = max(lmax, (u32)__shfl_xor((int)lmax, o));
    const double pts[5] = {0.0, 0.5, 0.95, 0.99, 1.0};
    for (int qi = 0; qi < 5; qi++) {
      const u32 want = min(n_stable - 1, (u32)floor((double)n_stable * pts[qi]));
      u32 lo = 0, hi = lmax;  // smallest v with count(lat <= v) > want
      while (lo < hi) {
        const u32 mid = lo + (hi - lo) / 2;
        u32 c = 0;
        for (u32 e = lane; e < v_cur; e += 64) { const u32 l = lat[e]; c += (l != NOLAT && l <= mid) ? 1u : 0u; }
        if (c_wave_sum(c) > want) hi = mid; else lo = mid + 1;
      }
      q[qi] = lo;
    }
  }

  if (lane == 0) {
    msim_check_result o;
    o.attempt_count = v_cur; o.stable_count = n_stable; o.lost_count = n_lost; o.never_read_count = n_never;
    o.stale_count = n_stale; o.duplicated_count = 0; o.error_count = errors;
    for (int i = 0; i < 5; i++) o.stable_latency_ms[i] = q[i];
#ifdef CK_PROF
    o.stable_latency_ms[0] = (u32)(ck_t1 - ck_t0); o.stable_latency_ms[1] = (u32)(ck_t2 - ck_t1); o.stable_latency_ms[2] = (u32)(__builtin_readcyclecounter() - ck_t2); o.stable_latency_ms[3] = ck_rounds; o.stable_latency_ms[4] = n_ri;
#endif
    o.op_count = op_count; o.ok_count = n_ok; o.fail_count = n_fail; o.info_count = n_info;
    if (meta.flags) o.valid = 0;
    else if (!setfull) o.valid = errors == 0 ? 1 : 0;
    else o.valid = n_lost ? 0u : (n_stable == 0 ? 2u : 1u);
    p.out[inst] = o;
  }
}

// LDS of one wavefront: three u16 indices per element (reused as one u32 latency per element), the valid bitmap over the read
// ranks (read one word past the last chunk), the per-thread tables slot / first / slotv
static size_t check_lds_layout(CParams &cp) {
  size_t lds = (size_t)cp.max_values * 3 * 2 < (size_t)cp.max_values * 4 ? (size_t)cp.max_values * 4 : (size_t)cp.max_values * 3 * 2;
  lds = (lds + 3) & ~(size_t)3;
  cp.off_valid = (u32)lds;
  lds += ((size_t)(cp.max_reads + 31) / 32 + 2) * 4;
  cp.off_tbl = (u32)lds;
  lds += (size_t)cp.C * 3 * 4;
  return lds;
}

static const char *check_limits(u32 max_rows, u32 max_values, u32 concurrency) {
  if (max_rows >= 0xFFFF || max_values > 2048 || concurrency > 128 || concurrency == 0) return "device checker: max_rows must be < 65535, max_values <= 2048, concurrency 1..128";
  return nullptr;
}

// one wavefront per history over n slabs; the caller owns the buffers and the stream
static hipError_t check_dispatch(CParams &cp, u32 n, hipStream_t st) {
  cp.max_reads = cp.max_rows / 2 + 1;  // every :ok read has its own :invoke row
  cp.rcp_C = 1.0f / (float)cp.C;
  const size_t lds = check_lds_layout(cp);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&check_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(check_kernel, dim3(n), dim3(64), lds, st, cp);
  return hipGetLastError();
}

int msim_check_launch(msim_ctx *ctx) {
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const msim_config &c = ctx->cfg;
  CParams cp;
  std::memset(&cp, 0, sizeof cp);
  cp.rows = ctx->d_rows; cp.payload = ctx->d_payload; cp.meta = ctx->d_meta; cp.out = ctx->d_check;
  cp.max_rows = c.max_rows; cp.max_pay = c.max_payload_words; cp.max_values = c.max_values; cp.C = c.concurrency; cp.workload = c.workload;
  if (const char *why = check_limits(c.max_rows, c.max_values, c.concurrency)) { ctx->err = why; return MSIM_E_UNSUPPORTED; }
  const size_t rec_bytes = (size_t)ctx->n_inst * (c.max_rows / 2 + 1) * 3 * sizeof(u32);
  if (ctx->cap_check_scratch < rec_bytes) {
    if (ctx->d_check_scratch) (void)msim_dev_free(ctx->d_check_scratch);
    ctx->d_check_scratch = nullptr; ctx->cap_check_scratch = 0;
    MSIM_HIP_TRY(ctx, msim_dev_malloc(&ctx->d_check_scratch, rec_bytes));
    ctx->cap_check_scratch = rec_bytes;
  }
  cp.recs = static_cast<u32 *>(ctx->d_check_scratch);
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev2, ctx->stream));
  MSIM_HIP_TRY(ctx, check_dispatch(cp, ctx->n_inst, ctx->stream));
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev3, ctx->stream));
  MSIM_HIP_TRY(ctx, hipEventSynchronize(ctx->ev3));
  MSIM_HIP_TRY(ctx, hipEventElapsedTime(&ctx->check_ms, ctx->ev2, ctx->ev3));
  ctx->checked = true; ctx->check_fetched = false;
  return MSIM_OK;
}

// Checks `n_histories` broadcast / g-set (set-full) or echo histories given on the host with the device checker of msim_check:
// history i lies in the slabs rows + i * max_rows (n_rows[i] rows used) and payload + i * max_payload_words.
extern "C" int msim_check_set_full_batch(int device, uint32_t workload, uint32_t concurrency, const msim_op *rows, const uint32_t *n_rows, uint32_t max_rows,
                                         const uint32_t *payload, uint32_t max_payload_words, uint32_t max_values, uint32_t n_histories, msim_check_result *out) {
  if (!rows || !n_rows || !out || n_histories == 0 || max_rows == 0 || (!payload && max_payload_words)) return MSIM_E_INVALID;
  if (workload != MSIM_WL_ECHO && workload != MSIM_WL_BROADCAST && workload != MSIM_WL_G_SET) return MSIM_E_INVALID;
  if (check_limits(max_rows, max_values, concurrency)) return MSIM_E_UNSUPPORTED;
  if (hipSetDevice(device) != hipSuccess) return MSIM_E_HIP;
  std::vector<msim_inst_meta> hm(n_histories);
  for (u32 i = 0; i < n_histories; i++) { std::memset(&hm[i], 0, sizeof hm[i]); if (n_rows[i] > max_rows) return MSIM_E_RANGE; hm[i].n_rows = n_rows[i]; }
  CParams cp;
  std::memset(&cp, 0, sizeof cp);
  cp.max_rows = max_rows; cp.max_pay = max_payload_words; cp.max_values = max_values; cp.C = concurrency; cp.workload = workload;
  msim_op *d_rows = nullptr; u32 *d_pay = nullptr; msim_inst_meta *d_meta = nullptr; msim_check_result *d_out = nullptr; u32 *d_rec = nullptr;
  const size_t pay_bytes = (size_t)n_histories * (max_payload_words ? max_payload_words : 1) * sizeof(u32);
  int rc = MSIM_E_HIP;
  do {
    if (msim_dev_malloc(&d_rows, (size_t)n_histories * max_rows * sizeof(msim_op)) != hipSuccess) break;
    if (msim_dev_malloc(&d_pay, pay_bytes) != hipSuccess) break;
    if (msim_dev_malloc(&d_meta, (size_t)n_histories * sizeof(msim_inst_meta)) != hipSuccess) break;
    if (msim_dev_malloc(&d_out, (size_t)n_histories * sizeof(msim_check_result)) != hipSuccess) break;
    if (msim_dev_malloc(&d_rec, (size_t)n_histories * (max_rows / 2 + 1) * 3 * sizeof(u32)) != hipSuccess) break;
    if (hipMemcpy(d_rows, rows, (size_t)n_histories * max_rows * sizeof(msim_op), hipMemcpyHostToDevice) != hipSuccess) break;
    if (max_payload_words && hipMemcpy(d_pay, payload, pay_bytes, hipMemcpyHostToDevice) != hipSuccess) break;
    if (hipMemcpy(d_meta, hm.data(), (size_t)n_histories * sizeof(msim_inst_meta), hipMemcpyHostToDevice) != hipSuccess) break;
    cp.rows = d_rows; cp.payload = d_pay; cp.meta = d_meta; cp.out = d_out; cp.recs = d_rec;
    if (check_dispatch(cp, n_histories, nullptr) != hipSuccess) break;
    if (hipDeviceSynchronize() != hipSuccess) break;
    if (hipMemcpy(out, d_out, (size_t)n_histories * sizeof(msim_check_result), hipMemcpyDeviceToHost) != hipSuccess) break;
    rc = MSIM_OK;
  } while (false);
  for (void *q : {(void *)d_rows, (void *)d_pay, (void *)d_meta, (void *)d_out, (void *)d_rec}) if (q) (void)msim_dev_free(q);
  return rc;
}

// ---- maelstrom.checker/availability-checker (checker.clj:6-39) ---------------------------------------------------------
// :ok-fraction = (float (/ ok-count invoke-count)) over (h/oks history) and (h/invokes history) — every :ok and every :invoke row
// (nemesis ops are :info) — 1 for an empty history; :valid? by --availability (core.clj:149): nil -> true, :total -> the
// fraction == 1, a number a -> a <= fraction.  One wavefront per history counts the two row types from HBM; the verdicts are
// formed on the host (n x 2 counters cross PCIe).
__global__ void __launch_bounds__(64) availability_kernel(const msim_op *rows, const msim_inst_meta *meta, u32 max_rows, uint2 *out) {
  const u32 lane = threadIdx.x, inst = blockIdx.x;
  const u32 n = meta[inst].n_rows;
  const uint4 *r = reinterpret_cast<const uint4 *>(rows + (size_t)inst * max_rows);
  u32 ok = 0, inv = 0;
  for (u32 i = lane; i < n; i += 64) { const u32 t = r[i].z & 3u; ok += t == MSIM_T_OK; inv += t == MSIM_T_INVOKE; }
  ok = c_wave_sum(ok); inv = c_wave_sum(inv);
  if (lane == 0) out[inst] = make_uint2(ok, inv);
}

static void availability_verdict(uint32_t ok, uint32_t inv, uint32_t mode, double a, msim_availability *o) {
  o->ok_count = ok; o->invoke_count = inv;
  o->ok_fraction = inv == 0 ? 1.0f : (float)((double)ok / (double)inv);   // (float (/ ok-count invoke-count)): a ratio rounded once
  o->valid = mode == MSIM_AVAIL_NIL ? 1u : mode == MSIM_AVAIL_TOTAL ? (o->ok_fraction == 1.0f ? 1u : 0u) : (a <= (double)o->ok_fraction ? 1u : 0u);
}

extern "C" int msim_check_availability_rows(const msim_op *rows, uint32_t n_rows, uint32_t mode, double availability, msim_availability *out) {
  if ((!rows && n_rows) || !out || mode > MSIM_AVAIL_FRACTION || (mode == MSIM_AVAIL_FRACTION && !(availability >= 0.0 && availability <= 1.0))) return MSIM_E_INVALID;
  uint32_t ok = 0, inv = 0;
  for (uint32_t i = 0; i < n_rows; i++) { const uint32_t t = MSIM_OP_TYPE(rows[i]); ok += t == MSIM_T_OK; inv += t == MSIM_T_INVOKE; }
  availability_verdict(ok, inv, mode, availability, out);
  return MSIM_OK;
}

extern "C" int msim_check_availability(msim_ctx *ctx, uint32_t mode, double availability, msim_availability *out, uint32_t n_out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  if (!ctx->ran) { ctx->err = "msim_check_availability before msim_run"; return MSIM_E_RANGE; }
  if (mode > MSIM_AVAIL_FRACTION || (mode == MSIM_AVAIL_FRACTION && !(availability >= 0.0 && availability <= 1.0))) {
    ctx->err = "--availability is nil, total, or a number from 0 to 1 (core.clj:149, checker.clj:36-39)"; return MSIM_E_INVALID; }
  if (n_out < ctx->n_inst) { ctx->err = "msim_check_availability: output array shorter than the run"; return MSIM_E_RANGE; }
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)ctx->n_inst * sizeof(uint2);
  uint2 *d = nullptr;
  MSIM_HIP_TRY(ctx, msim_dev_malloc(reinterpret_cast<void **>(&d), bytes));
  hipLaunchKernelGGL(availability_kernel, dim3(ctx->n_inst), dim3(64), 0, ctx->stream, ctx->d_rows, ctx->d_meta, ctx->cfg.max_rows, d);
  std::vector<uint2> h(ctx->n_inst);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)msim_dev_free(d);
  if (e != hipSuccess) { ctx->err = std::string("msim_check_availability: ") + hipGetErrorString(e); return MSIM_E_HIP; }
  for (uint32_t i = 0; i < ctx->n_inst; i++) availability_verdict(h[i].x, h[i].y, mode, availability, &out[i]);
  return MSIM_OK;
}
