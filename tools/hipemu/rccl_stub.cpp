// rccl_stub.cpp -> tools/hipemu/_build/librccl.so: the nine RCCL entry points csrc/gather.cpp binds, between PROCESSES of one machine
// over a Unix-domain socket hub, for "device" memory that is host memory (the wavefront emulator's).  Test infrastructure: it exists so
// that the N > 1 branch of msim_gather (size all-gather, grouped send / recv to the root) executes on a machine without GPUs; it is never
// linked into, loaded by or shipped with the product (which dlopens the real librccl.so).
//   ncclGetUniqueId      a random abstract socket name
//   ncclCommInitRank     rank 0 of the communicator listens there and runs the hub thread; every rank (0 included) connects
//   ncclAllGather        every rank sends its part to the hub, the hub answers each with the concatenation in rank order
//   ncclSend / ncclRecv  framed messages through the hub; inside a group they are queued and run at ncclGroupEnd, sends first
// Streams are ignored: every call has completed when it returns (the emulator's copies are synchronous too).
#include <poll.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

namespace {

enum { OP_HELLO = 1, OP_ALLGATHER = 2, OP_P2P = 3 };
struct Hdr { uint32_t op, src, dst, pad; uint64_t bytes; };

bool wr(int fd, const void *p, size_t n) { const char *c = static_cast<const char *>(p); while (n) { ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) return false; c += k; n -= (size_t)k; } return true; }
bool rd(int fd, void *p, size_t n) { char *c = static_cast<char *>(p); while (n) { ssize_t k = ::recv(fd, c, n, 0); if (k <= 0) return false; c += k; n -= (size_t)k; } return true; }

sockaddr_un addr_of(const ncclUniqueId &id, socklen_t *len) {
  sockaddr_un a; std::memset(&a, 0, sizeof a); a.sun_family = AF_UNIX;
  a.sun_path[0] = 0;   // abstract namespace: nothing to unlink
  std::memcpy(a.sun_path + 1, id.internal, 40);
  *len = (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + 40);
  return a;
}

// the hub: relays point-to-point frames, answers all-gathers once every rank's part is in
void hub_main(int lfd, int world) {
  std::vector<int> fd((size_t)world, -1);
  for (int got = 0; got < world;) {
    const int c = ::accept(lfd, nullptr, nullptr);
    if (c < 0) return;
    Hdr h; if (!rd(c, &h, sizeof h) || h.op != OP_HELLO || h.src >= (uint32_t)world) { ::close(c); continue; }
    fd[h.src] = c; got++;
  }
  ::close(lfd);
  std::vector<std::vector<char>> parts((size_t)world);
  std::vector<bool> have((size_t)world, false);
  std::vector<pollfd> pf((size_t)world);
  int open_n = world;
  while (open_n > 0) {
    for (int r = 0; r < world; r++) { pf[(size_t)r].fd = fd[(size_t)r]; pf[(size_t)r].events = POLLIN; pf[(size_t)r].revents = 0; }
    if (::poll(pf.data(), (nfds_t)world, -1) < 0) return;
    for (int r = 0; r < world; r++) {
      if (fd[(size_t)r] < 0 || !(pf[(size_t)r].revents & (POLLIN | POLLHUP))) continue;
      Hdr h;
      if (!rd(fd[(size_t)r], &h, sizeof h)) { ::close(fd[(size_t)r]); fd[(size_t)r] = -1; open_n--; continue; }
      std::vector<char> body((size_t)h.bytes);
      if (h.bytes && !rd(fd[(size_t)r], body.data(), body.size())) { ::close(fd[(size_t)r]); fd[(size_t)r] = -1; open_n--; continue; }
      if (h.op == OP_P2P) {
        if (h.dst < (uint32_t)world && fd[h.dst] >= 0) { wr(fd[h.dst], &h, sizeof h); if (h.bytes) wr(fd[h.dst], body.data(), body.size()); }
      } else if (h.op == OP_ALLGATHER) {
        parts[(size_t)r] = std::move(body); have[(size_t)r] = true;
        bool all = true; for (int q = 0; q < world; q++) all = all && have[(size_t)q];
        if (all) {
          std::vector<char> cat; for (int q = 0; q < world; q++) cat.insert(cat.end(), parts[(size_t)q].begin(), parts[(size_t)q].end());
          Hdr o{OP_ALLGATHER, 0, 0, 0, cat.size()};
          for (int q = 0; q < world; q++) if (fd[(size_t)q] >= 0) { wr(fd[(size_t)q], &o, sizeof o); wr(fd[(size_t)q], cat.data(), cat.size()); }
          for (int q = 0; q < world; q++) { have[(size_t)q] = false; parts[(size_t)q].clear(); }
        }
      }
    }
  }
}

size_t dt_size(ncclDataType_t t) { return t == ncclInt8 || t == ncclUint8 ? 1 : t == ncclInt32 || t == ncclUint32 ? 4 : 8; }

struct Pending { bool send; void *buf; size_t bytes; int peer; struct ncclComm *comm; };
thread_local int g_group = 0;
thread_local std::vector<Pending> g_pending;

}  // namespace

struct ncclComm {
  int rank = 0, world = 1, fd = -1;
  std::thread hub;
  std::multimap<uint32_t, std::vector<char>> early;   // frames that arrived before their ncclRecv was posted, by source rank
};

namespace {

ncclResult_t do_send(ncclComm *c, const void *buf, size_t bytes, int peer) {
  Hdr h{OP_P2P, (uint32_t)c->rank, (uint32_t)peer, 0, bytes};
  return wr(c->fd, &h, sizeof h) && (!bytes || wr(c->fd, buf, bytes)) ? ncclSuccess : ncclSystemError;
}
ncclResult_t do_recv(ncclComm *c, void *buf, size_t bytes, int peer) {
  auto it = c->early.find((uint32_t)peer);
  if (it != c->early.end()) { if (it->second.size() != bytes) return ncclInvalidArgument; if (bytes) std::memcpy(buf, it->second.data(), bytes); c->early.erase(it); return ncclSuccess; }
  for (;;) {
    Hdr h; if (!rd(c->fd, &h, sizeof h)) return ncclSystemError;
    if (h.op == OP_P2P && h.src == (uint32_t)peer) { if (h.bytes != bytes) return ncclInvalidArgument; return !bytes || rd(c->fd, buf, bytes) ? ncclSuccess : ncclSystemError; }
    std::vector<char> body((size_t)h.bytes);
    if (h.bytes && !rd(c->fd, body.data(), body.size())) return ncclSystemError;
    c->early.emplace(h.src, std::move(body));
  }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof *id);
  std::random_device rd_;
  std::snprintf(id->internal, 41, "msim-rccl-stub-%08x%08x%08x", (unsigned)rd_(), (unsigned)rd_(), (unsigned)::getpid());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  socklen_t len; const sockaddr_un a = addr_of(id, &len);
  ncclComm *c = new ncclComm; c->rank = rank; c->world = world;
  if (rank == 0) {
    const int lfd = ::socket(AF_UNIX, SOCK_STREAM, 0);
    if (lfd < 0 || ::bind(lfd, reinterpret_cast<const sockaddr *>(&a), len) != 0 || ::listen(lfd, world) != 0) { delete c; return ncclSystemError; }
    c->hub = std::thread(hub_main, lfd, world);
  }
  c->fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
  int tries = 0;
  while (::connect(c->fd, reinterpret_cast<const sockaddr *>(&a), len) != 0) { if (++tries > 600) { delete c; return ncclSystemError; } ::usleep(50000); }
  Hdr h{OP_HELLO, (uint32_t)rank, 0, 0, 0};
  if (!wr(c->fd, &h, sizeof h)) { delete c; return ncclSystemError; }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->fd >= 0) ::close(c->fd);
  if (c->hub.joinable()) c->hub.join();
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t) {
  const size_t bytes = count * dt_size(dt);
  Hdr h{OP_ALLGATHER, (uint32_t)c->rank, 0, 0, bytes};
  if (!wr(c->fd, &h, sizeof h) || (bytes && !wr(c->fd, send, bytes))) return ncclSystemError;
  for (;;) {
    Hdr o; if (!rd(c->fd, &o, sizeof o)) return ncclSystemError;
    if (o.op == OP_ALLGATHER) { if (o.bytes != bytes * (size_t)c->world) return ncclInternalError; return !o.bytes || rd(c->fd, recv, (size_t)o.bytes) ? ncclSuccess : ncclSystemError; }
    std::vector<char> body((size_t)o.bytes);
    if (o.bytes && !rd(c->fd, body.data(), body.size())) return ncclSystemError;
    c->early.emplace(o.src, std::move(body));
  }
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t) {
  if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
  if (g_group) { g_pending.push_back({true, const_cast<void *>(buf), count * dt_size(dt), peer, c}); return ncclSuccess; }
  return do_send(c, buf, count * dt_size(dt), peer);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t) {
  if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
  if (g_group) { g_pending.push_back({false, buf, count * dt_size(dt), peer, c}); return ncclSuccess; }
  return do_recv(c, buf, count * dt_size(dt), peer);
}
ncclResult_t ncclGroupStart() { g_group++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_group == 0) return ncclInvalidUsage;
  if (--g_group) return ncclSuccess;
  ncclResult_t res = ncclSuccess;
  for (const Pending &p : g_pending) if (p.send && res == ncclSuccess) res = do_send(p.comm, p.buf, p.bytes, p.peer);
  for (const Pending &p : g_pending) if (!p.send && res == ncclSuccess) res = do_recv(p.comm, p.buf, p.bytes, p.peer);
  g_pending.clear();
  return res;
}
const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) { case ncclSuccess: return "no error"; case ncclSystemError: return "stub: socket error"; case ncclInvalidArgument: return "stub: invalid argument (size mismatch between a send and its recv?)";
    case ncclInvalidUsage: return "stub: invalid usage"; default: return "stub: internal error"; }
}

}  // extern "C"
