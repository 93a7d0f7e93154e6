// group8.h — device helpers shared by the eight-clusters-per-wavefront kernels (hat8.hip, uid8.hip, crdt8.hip, bcast8.hip): the latency
// sampler's logarithm over a per-translation-unit copy of the Q24 table, the minimum over the lanes of a group.
#ifndef MSIM_GROUP8_H
#define MSIM_GROUP8_H
#include "wave_common.h"
#include "log2_table.h"

namespace {

__constant__ u32 g8_log2_q24[257];   // (uploaded once per device by the launcher of the translation unit: MSIM_UPLOAD_ONCE)

// -ln(u), u = (r+1)/2^32, Q16, integer only: the sampler of engine.hip / the oracle
__device__ __forceinline__ u32 g8_neg_ln_q16(u32 r) {
  if (r == 0xFFFFFFFFu) return 0;
  const u32 v = r + 1;
  const u32 e = 31 - __clz(v);
  const u32 m = v << (31 - e);
  const u32 idx = (m >> 23) & 0xFF;
  const u32 f = (m >> 7) & 0xFFFF;
  const u32 l0 = g8_log2_q24[idx], l1 = g8_log2_q24[idx + 1];
  const u32 lg = (e << 24) + l0 + (u32)(((u64)(l1 - l0) * f) >> 16);
  const u32 d = (32u << 24) - lg;
  return (u32)(((u64)d * 2977044472ull) >> 40);
}
// min over the GS (4 or 8) lanes of the caller's group, in every lane of it
template <int GS>
__device__ __forceinline__ u32 g8_min(u32 v) {
  v = min(v, dpp_mov<0xB1, 0xF, 0xF, false>(v, v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_mov<0x4E, 0xF, 0xF, false>(v, v));   // quad_perm [2,3,0,1]
  if (GS == 8) v = min(v, dpp_mov<0x141, 0xF, 0xF, false>(v, v));  // row_half_mirror
  return v;
}

}  // namespace
#endif
