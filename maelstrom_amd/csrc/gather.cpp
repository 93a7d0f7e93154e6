// gather.cpp — the ensemble's one exchange step behind the C-ABI: variable-length history gather to a root rank over RCCL
// (include/maelsim.h "multi-GPU ensemble").  Host code only: the device work is msim_compact_on_device
// (engine.hip) and RCCL's own kernels.
//
// RCCL is bound at run time (dlopen "librccl.so"): the library must load on single-GPU hosts and in CPU-only test
// containers that never gather.  Only types and constants come from <rccl/rccl.h>.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "engine_internal.h"

static_assert(MSIM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "msim_comm_unique_id hands out an ncclUniqueId");

namespace {

struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { r.err = std::string("RCCL not found: ") + dlerror(); return; }
#define BIND(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym)); if (!r.field) { r.err = std::string("RCCL symbol missing: ") + sym; return; }
    BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommDestroy, "ncclCommDestroy")
    BIND(AllGather, "ncclAllGather") BIND(Send, "ncclSend") BIND(Recv, "ncclRecv") BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd") BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    r.ok = true;
  });
  return r;
}

#define MSIM_NCCL_TRY(ctx, call)                                                              \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) { (ctx)->err = std::string(#call) + ": " + rccl().GetErrorString(r_); return MSIM_E_HIP; } \
  } while (0)

template <typename T>
int grow(msim_ctx *ctx, T **buf, size_t *cap, size_t bytes) {
  if (*buf && *cap >= bytes) return MSIM_OK;
  if (*buf) { (void)msim_dev_free(*buf); *buf = nullptr; *cap = 0; }
  const size_t want = bytes + bytes / 8 + 256;
  MSIM_HIP_TRY(ctx, msim_dev_malloc(reinterpret_cast<void **>(buf), want));
  *cap = want;
  return MSIM_OK;
}

}  // namespace

// Where every rank's part of every slab kind lands in the root's buffers: offs[kind][rank] in bytes, totals[kind].
// sizes = world x 4 byte counts {rows, payload, meta, stats} as all-gathered.  Shared with the host-side mirror
// (maelstrom_amd/ensemble.py computes the same layout for the torch transport).
void msim_gather_layout(const uint64_t *sizes, int world, std::vector<uint64_t> offs[4], uint64_t totals[4]) {
  for (int k = 0; k < 4; k++) {
    offs[k].assign((size_t)world, 0);
    uint64_t o = 0;
    for (int r = 0; r < world; r++) { offs[k][(size_t)r] = o; o += sizes[(size_t)r * 4 + k]; }
    totals[k] = o;
  }
}

void msim_gather_free(msim_ctx *ctx) {
  for (int k = 0; k < 4; k++) { if (ctx->d_all[k]) (void)msim_dev_free(ctx->d_all[k]); ctx->d_all[k] = nullptr; ctx->cap_all[k] = 0; }
  if (ctx->d_sizes) { (void)msim_dev_free(ctx->d_sizes); ctx->d_sizes = nullptr; }
  if (ctx->ev_g0) { (void)hipEventDestroy(ctx->ev_g0); (void)hipEventDestroy(ctx->ev_g1); ctx->ev_g0 = ctx->ev_g1 = nullptr; }
  if (ctx->comm && rccl().ok) (void)rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm));
  ctx->comm = nullptr;
}

extern "C" int msim_comm_unique_id(unsigned char id[MSIM_COMM_ID_BYTES]) {
  if (!id) return MSIM_E_INVALID;
  Rccl &r = rccl();
  if (!r.ok) return MSIM_E_UNSUPPORTED;
  ncclUniqueId u;
  if (r.GetUniqueId(&u) != ncclSuccess) return MSIM_E_HIP;
  std::memcpy(id, u.internal, MSIM_COMM_ID_BYTES);
  return MSIM_OK;
}

extern "C" int msim_comm_init(msim_ctx *ctx, const unsigned char id[MSIM_COMM_ID_BYTES], int rank, int world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) { if (ctx) ctx->err = "msim_comm_init: bad rank / world"; return MSIM_E_INVALID; }
  Rccl &r = rccl();
  if (!r.ok) { ctx->err = r.err; return MSIM_E_UNSUPPORTED; }
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->comm) { (void)r.CommDestroy(static_cast<ncclComm_t>(ctx->comm)); ctx->comm = nullptr; }
  if (ctx->d_sizes) { (void)msim_dev_free(ctx->d_sizes); ctx->d_sizes = nullptr; }   // sized for the previous world
  ncclUniqueId u;
  std::memcpy(u.internal, id, MSIM_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  MSIM_NCCL_TRY(ctx, r.CommInitRank(&comm, world, u, rank));
  ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
  return MSIM_OK;
}

extern "C" int msim_gather(msim_ctx *ctx, int root, msim_gathered *out) {
  if (!ctx || !out) return MSIM_E_INVALID;
  std::memset(out, 0, sizeof *out);
  const int world = ctx->comm ? ctx->comm_world : 1, rank = ctx->comm ? ctx->comm_rank : 0;
  if (root < 0 || root >= world) { ctx->err = "msim_gather: root out of range"; return MSIM_E_INVALID; }
  MSIM_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->ev_g0) { MSIM_HIP_TRY(ctx, hipEventCreate(&ctx->ev_g0)); MSIM_HIP_TRY(ctx, hipEventCreate(&ctx->ev_g1)); }   // (the checkers time themselves with ev2 / ev3)
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev_g0, ctx->stream));
  uint64_t row_units = 0, pay_words = 0;
  int rc = msim_compact_on_device(ctx, &row_units, &pay_words);
  if (rc != MSIM_OK) return rc;
  const uint32_t n = ctx->n_inst;
  const uint64_t mine[4] = {row_units * 16, pay_words * 4, (uint64_t)n * sizeof(msim_inst_meta), (uint64_t)n * sizeof(msim_net_stats)};
  const void *src[4] = {ctx->d_grows, ctx->d_gpay, ctx->d_meta, ctx->d_stats};
  std::vector<uint64_t> sizes((size_t)world * 4);
  if (world == 1) std::memcpy(sizes.data(), mine, sizeof mine);
  else {
    // (1) every rank learns every rank's four byte counts
    Rccl &r = rccl();
    if (!ctx->d_sizes) MSIM_HIP_TRY(ctx, msim_dev_malloc(reinterpret_cast<void **>(&ctx->d_sizes), ((size_t)world + 1) * 4 * 8));
    uint64_t *d_mine = ctx->d_sizes + (size_t)world * 4;
    MSIM_HIP_TRY(ctx, hipMemcpyAsync(d_mine, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    MSIM_NCCL_TRY(ctx, r.AllGather(d_mine, ctx->d_sizes, 4, ncclUint64, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    MSIM_HIP_TRY(ctx, hipMemcpyAsync(sizes.data(), ctx->d_sizes, sizes.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    MSIM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<uint64_t> offs[4];
  uint64_t totals[4];
  msim_gather_layout(sizes.data(), world, offs, totals);
  uint64_t received = 0;
  if (rank == root)
    for (int k = 0; k < 4; k++) if ((rc = grow(ctx, &ctx->d_all[k], &ctx->cap_all[k], (size_t)totals[k] + 16)) != MSIM_OK) return rc;
  if (world > 1) {
    // (2) one grouped exchange: every peer sends its four slabs, the root receives each at its place (bytes: ncclUint8)
    Rccl &r = rccl();
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    MSIM_NCCL_TRY(ctx, r.GroupStart());
    ncclResult_t bad = ncclSuccess;   // a failing call must not leave the group open on this thread
    for (int k = 0; k < 4 && bad == ncclSuccess; k++) {
      if (rank == root) {
        for (int p = 0; p < world && bad == ncclSuccess; p++) {
          const uint64_t sz = sizes[(size_t)p * 4 + k];
          if (p == root || sz == 0) continue;
          bad = r.Recv(static_cast<unsigned char *>(ctx->d_all[k]) + offs[k][(size_t)p], sz, ncclUint8, p, comm, ctx->stream);
          received += sz;
        }
      } else if (mine[k]) bad = r.Send(src[k], mine[k], ncclUint8, root, comm, ctx->stream);
    }
    const ncclResult_t ended = r.GroupEnd();
    MSIM_NCCL_TRY(ctx, bad);
    MSIM_NCCL_TRY(ctx, ended);
  }
  if (rank == root)   // the root's own part: device-to-device
    for (int k = 0; k < 4; k++)
      if (mine[k]) MSIM_HIP_TRY(ctx, hipMemcpyAsync(static_cast<unsigned char *>(ctx->d_all[k]) + offs[k][(size_t)root], src[k], mine[k], hipMemcpyDeviceToDevice, ctx->stream));
  MSIM_HIP_TRY(ctx, hipEventRecord(ctx->ev_g1, ctx->stream));
  MSIM_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_g1));
  MSIM_HIP_TRY(ctx, hipEventElapsedTime(&out->ms, ctx->ev_g0, ctx->ev_g1));
  if (rank == root) {
    out->rows = ctx->d_all[0]; out->payload = ctx->d_all[1]; out->meta = ctx->d_all[2]; out->stats = ctx->d_all[3];
    out->rows_bytes = totals[0]; out->payload_bytes = totals[1]; out->meta_bytes = totals[2]; out->stats_bytes = totals[3];
    out->n_instances = (uint32_t)(totals[2] / sizeof(msim_inst_meta));
  }
  out->bytes_received = received; out->world = (uint32_t)world; out->rank = (uint32_t)rank;
  return MSIM_OK;
}
