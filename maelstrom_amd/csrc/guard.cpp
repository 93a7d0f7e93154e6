// guard.cpp — every device allocation of libmaelsim goes through msim_dev_malloc / msim_dev_free (engine_internal.h).  Without
// MSIM_GUARD in the environment they ARE hipMalloc / hipFree.  With it they are a developer's electric fence for HBM — built to
// hunt the GPU memory-access fault that ended the driver's round-4 bench run (BENCH_r04.json) and never came back:
//
//   MSIM_GUARD=1  "fence after":  the slab is mapped with HIP's virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap)
//                 so that its LAST byte is the last mapped byte of its reservation: one byte read or written past the end of a slab is a
//                 memory-access fault at a known address, whatever hipMalloc would have put there.  The unused head of the mapping is
//                 filled with a pattern and verified (msim_guard_check, msim_dev_free): writes before the slab are counted.
//   MSIM_GUARD=2  "fence before": the slab starts on the first mapped byte (under-runs fault, over-runs are counted in the tail pattern).
//   MSIM_GUARD=3  red zones only: hipMalloc(bytes + 2 x 4 KiB), both zones pattern-filled and verified (no VMM needed; writes only).
//   MSIM_GUARD_LOG=1 prints every slab's address range to stderr, so that a fault address names its slab (or the gap behind it).
//
// Modes 1 / 2 fall back to mode 3 when the device has no virtual-memory management.  Slabs are 16-byte aligned in every mode.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "engine_internal.h"

namespace {

constexpr unsigned char PATTERN = 0xA5;
constexpr size_t REDZONE = 4096;

struct Slab {
  int mode;                 // 1 / 2: VMM, 3: red zones
  size_t bytes;             // what the caller asked for
  char *va; size_t va_size; // VMM: the reservation (guard gap + mapping + guard gap)
  char *map; size_t map_size;   // VMM: the mapped part; mode 3: the hipMalloc block
#if !MSIM_HIPEMU
  hipMemGenericAllocationHandle_t handle;
#endif
};

std::mutex g_mu;
std::map<void *, Slab> g_slabs;
unsigned long long g_bad_bytes = 0, g_allocs = 0, g_fill_failures = 0;

int guard_mode() {
  static const int m = []() { const char *e = std::getenv("MSIM_GUARD"); return e ? std::atoi(e) : 0; }();
  return m;
}
bool guard_log() {
  static const bool l = []() { const char *e = std::getenv("MSIM_GUARD_LOG"); return e && std::atoi(e) != 0; }();
  return l;
}

// bytes of [p, p + n) that no longer hold the pattern
unsigned long long count_damage(const char *p, size_t n, const char *what, const void *slab) {
  if (!n) return 0;
  std::vector<unsigned char> h(n);
  if (hipMemcpy(h.data(), p, n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  unsigned long long bad = 0; size_t first = n;
  for (size_t i = 0; i < n; i++) if (h[i] != PATTERN) { bad++; if (first == n) first = i; }
  if (bad) std::fprintf(stderr, "[msim guard] %llu byte(s) of the %s zone of slab %p overwritten (first at %p)\n", bad, what, slab, (const void *)(p + first));
  return bad;
}

unsigned long long check_slab(void *ptr, const Slab &s) {
  char *const p = static_cast<char *>(ptr);
  unsigned long long bad = 0;
  bad += count_damage(s.map, (size_t)(p - s.map), "leading", ptr);
  bad += count_damage(p + s.bytes, (size_t)(s.map + s.map_size - (p + s.bytes)), "trailing", ptr);
  return bad;
}

#if !MSIM_HIPEMU   // (the host wavefront emulator has no virtual-memory API: red zones there)
hipError_t alloc_vmm(void **out, size_t bytes, int mode) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop;
  std::memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorNotSupported;
  const size_t need = (bytes + 15) & ~(size_t)15;
  const size_t map_size = ((need ? need : 16) + gran - 1) / gran * gran;
  Slab s;
  std::memset(&s, 0, sizeof s);
  s.mode = mode; s.bytes = bytes; s.map_size = map_size; s.va_size = map_size + 2 * gran;
  void *va = nullptr;
  e = hipMemAddressReserve(&va, s.va_size, gran, nullptr, 0);
  if (e != hipSuccess) return e;
  s.va = static_cast<char *>(va); s.map = s.va + gran;
  e = hipMemCreate(&s.handle, map_size, &prop, 0);
  if (e != hipSuccess) { (void)hipMemAddressFree(va, s.va_size); return e; }
  e = hipMemMap(s.map, map_size, 0, s.handle, 0);
  if (e != hipSuccess) { (void)hipMemRelease(s.handle); (void)hipMemAddressFree(va, s.va_size); return e; }
  hipMemAccessDesc acc;
  std::memset(&acc, 0, sizeof acc);
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(s.map, map_size, &acc, 1);
  if (e != hipSuccess) { (void)hipMemUnmap(s.map, map_size); (void)hipMemRelease(s.handle); (void)hipMemAddressFree(va, s.va_size); return e; }
  char *const p = mode == 1 ? s.map + (map_size - need) : s.map;   // fence after: the slab ends where the mapping ends
  // the pattern goes into the zones only (a slab of tens of GB is not filled), and is read back at once: a fill that did not land is the
  // guard's own malfunction, not a kernel's overrun
  const size_t lead = (size_t)(p - s.map), trail = (size_t)(s.map + map_size - (p + bytes));
  hipError_t f1 = lead ? hipMemset(s.map, PATTERN, lead) : hipSuccess, f2 = trail ? hipMemset(p + bytes, PATTERN, trail) : hipSuccess;
  (void)hipDeviceSynchronize();
  if (f1 != hipSuccess || f2 != hipSuccess || check_slab(p, s) != 0) {
    std::fprintf(stderr, "[msim guard] the pattern fill of slab %p did not land (%s / %s)\n", (void *)p, hipGetErrorString(f1), hipGetErrorString(f2));
    std::lock_guard<std::mutex> lk(g_mu); g_fill_failures++;
  }
  *out = p;
  std::lock_guard<std::mutex> lk(g_mu);
  g_slabs[p] = s; g_allocs++;
  if (guard_log()) std::fprintf(stderr, "[msim guard] slab %p .. %p (%zu bytes) mapped %p .. %p, unmapped gaps of %zu bytes on both sides\n",
                                (void *)p, (void *)(p + bytes), bytes, (void *)s.map, (void *)(s.map + map_size), gran);
  return hipSuccess;
}

#endif

hipError_t alloc_redzone(void **out, size_t bytes) {
  const size_t need = (bytes + 15) & ~(size_t)15;
  Slab s;
  std::memset(&s, 0, sizeof s);
  s.mode = 3; s.bytes = bytes; s.map_size = need + 2 * REDZONE;
  void *blk = nullptr;
  hipError_t e = hipMalloc(&blk, s.map_size);
  if (e != hipSuccess) return e;
  s.map = static_cast<char *>(blk);
  (void)hipMemset(blk, PATTERN, s.map_size);
  (void)hipDeviceSynchronize();
  char *const p = s.map + REDZONE;
  *out = p;
  std::lock_guard<std::mutex> lk(g_mu);
  g_slabs[p] = s; g_allocs++;
  if (guard_log()) std::fprintf(stderr, "[msim guard] slab %p .. %p (%zu bytes) with red zones of %zu bytes\n", (void *)p, (void *)(p + bytes), bytes, REDZONE);
  return hipSuccess;
}

}  // namespace

hipError_t msim_dev_malloc_impl(void **out, size_t bytes) {
  const int mode = guard_mode();
  if (mode <= 0) return hipMalloc(out, bytes);
#if !MSIM_HIPEMU
  if (mode == 1 || mode == 2) {
    static bool vmm_ok = true;
    if (vmm_ok) {
      const hipError_t e = alloc_vmm(out, bytes, mode);
      if (e == hipSuccess) return e;
      if (e == hipErrorOutOfMemory) return e;
      (void)hipGetLastError();
      vmm_ok = false;
      std::fprintf(stderr, "[msim guard] no virtual-memory management on this device (%s): red zones only\n", hipGetErrorString(e));
    }
  }
#endif
  return alloc_redzone(out, bytes);
}

hipError_t msim_dev_free(void *ptr) {
  if (!ptr) return hipSuccess;
  if (guard_mode() <= 0) return hipFree(ptr);
  Slab s;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_slabs.find(ptr);
    if (it == g_slabs.end()) return hipFree(ptr);   // (allocated before the guard was switched on: cannot happen, the mode is read once)
    s = it->second;
    g_slabs.erase(it);
  }
  (void)hipDeviceSynchronize();
  const unsigned long long bad = check_slab(ptr, s);
  if (bad) { std::lock_guard<std::mutex> lk(g_mu); g_bad_bytes += bad; }
  if (s.mode == 3) return hipFree(s.map);
#if !MSIM_HIPEMU
  (void)hipMemUnmap(s.map, s.map_size);
  (void)hipMemRelease(s.handle);
  // The reservation is NOT handed back: an address range that is reserved again and mapped to other memory was seen to keep serving stale
  // translations on this stack (zones "damaged" wholesale, faults that an isolated run of the same test does not have: gpurun_out/r6b) — and
  // an address that is never reused turns every use-after-free into a fault too.  A test suite consumes a few TB of the 128 TB address space.
  return hipSuccess;
#else
  return hipSuccess;
#endif
}

// Developer entry point (not part of the drop-in boundary): verifies the pattern zones of every live slab and returns the number of damaged
// bytes (freed slabs since the library was loaded + the live ones now) — 0 without MSIM_GUARD.  *n_allocs: slabs handed out under the guard so far.
extern "C" unsigned long long msim_guard_check(unsigned long long *n_allocs) {
  unsigned long long live_bad = 0;
  if (guard_mode() > 0) {
    (void)hipDeviceSynchronize();
    std::vector<std::pair<void *, Slab>> live;
    { std::lock_guard<std::mutex> lk(g_mu); live.assign(g_slabs.begin(), g_slabs.end()); }
    for (auto &kv : live) live_bad += check_slab(kv.first, kv.second);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  if (n_allocs) *n_allocs = g_allocs;
  if (g_fill_failures) std::fprintf(stderr, "[msim guard] %llu slab(s) whose pattern fill did not land\n", g_fill_failures);
  return g_bad_bytes + live_bad;   // slabs already freed + the live ones as they are now
}

// Developer entry point: proves that the guard sees an overrun.  Allocates a 100-byte slab through the guard, writes ONE byte behind it (inside
// the 16-byte rounding, which every mode keeps mapped and pattern-filled) and — unless the slab starts on the first mapped byte — one before it,
// frees it; returns the damaged bytes the guard counted for it: 2 (MSIM_GUARD=1, 3) or 1 (MSIM_GUARD=2), -1 if the allocation failed.
extern "C" int msim_guard_selftest(void) {
  if (guard_mode() <= 0) return 0;
  char *p = nullptr;
  if (msim_dev_malloc(&p, 100) != hipSuccess) return -1;
  unsigned long long before;
  { std::lock_guard<std::mutex> lk(g_mu); before = g_bad_bytes; }
  const unsigned char x = 0x5A;
  (void)hipMemcpy(p + 100 + 5, &x, 1, hipMemcpyHostToDevice);
  if (guard_mode() != 2) (void)hipMemcpy(p - 1, &x, 1, hipMemcpyHostToDevice);
  (void)msim_dev_free(p);
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)(g_bad_bytes - before);
}
