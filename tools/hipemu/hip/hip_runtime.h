// hip/hip_runtime.h of tools/hipemu — a HOST stand-in for the HIP runtime and the gfx950 wavefront intrinsics the kernels of
// maelstrom_amd/csrc use.  DEVELOPER / TEST TOOL ONLY: it lets the kernel SOURCES be compiled with a host compiler
// (tools/hipemu/build_emu.py -> tools/hipemu/_build/libmaelsim_emu.so) and stepped through on a machine without a GPU, so that
// the logic of a new kernel can be compared with the oracle before GPU minutes are spent on it.  It is not a backend of the
// product: libmaelsim.so is only ever built by hipcc for gfx950 (maelstrom_amd/build.py), nothing under maelstrom_amd/ knows this
// directory exists, and the engine fails loudly without a GPU.
//
// Execution model: one OS thread runs one workgroup (= one wavefront of 64 lanes) at a time; every lane is a fiber with its own
// stack; a lane runs until it reaches a cross-lane operation (ballot, readlane, DPP, ds_bpermute, shuffles, barriers, wavefront
// fences), where it parks; when every live lane is parked the operation is evaluated for all of them and they continue.  Between two
// such points lanes run one after the other in lane order, which is ONE of the interleavings the hardware's lockstep execution
// allows for code that orders its cross-lane memory traffic with barriers / fences (as the kernels here do).
// Limits: cross-lane operations must be reached by all live lanes of the wavefront together (wave-uniform control flow around
// them); a kernel that calls one under a divergent branch is reported (HIPEMU_DIVERGENT=1: lowest call site first, a guess).
// Workgroups are one-dimensional, up to 1024 threads (wavefront operations act within each group of 64 lanes).  LDS is 160 KiB per workgroup (`smem`, `csmem`).
#ifndef HIPEMU_HIP_RUNTIME_H
#define HIPEMU_HIP_RUNTIME_H

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define HIPEMU 1
using std::floor; using std::ceil; using std::sqrt; using std::fabs;

// ---- language keywords ---------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local
#define HIP_SYMBOL(x) (x)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

extern thread_local __attribute__((aligned(16))) unsigned char smem[];
extern thread_local __attribute__((aligned(16))) unsigned char csmem[];

// ---- vector types ----------------------------------------------------------------------------------------------------------------
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

// ---- runtime API -----------------------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef struct hipemu_stream *hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point t; };
typedef hipemu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDefault = 0, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorInvalidValue ? "invalid argument" : e == hipErrorOutOfMemory ? "out of memory" : "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)aligned_alloc(256, (n + 255) & ~(size_t)255); return *p || !n ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { *p = (T *)aligned_alloc(256, (n + 255) & ~(size_t)255); return *p || !n ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T &sym, const void *s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) { memcpy((char *)&sym + off, s, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event(); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "hipemu"); p->multiProcessorCount = 256; return hipSuccess; }

// ---- the wavefront emulator ------------------------------------------------------------------------------------------------------
namespace hipemu {
enum Op { OP_BALLOT, OP_READLANE, OP_READFIRST, OP_DPP, OP_BPERMUTE, OP_SHFL, OP_SHFL_XOR, OP_SHFL_UP, OP_SHFL_DOWN, OP_BARRIER };
struct Wave;
Wave *cur();                       // the wavefront this thread is executing
unsigned cur_lane();               // the lane whose fiber is running
unsigned cur_block();
unsigned cur_grid();
unsigned cur_block_y();
unsigned cur_grid_y();
unsigned cur_block_dim();
// parks the calling lane at a cross-lane operation; returns its result.  a = the lane's value, b = the lane's second operand
// (lane index / xor mask / delta / byte address), c = constant operands (DPP: ctrl | row_mask << 16 | bank_mask << 20 | bound << 24, old in d)
unsigned long long collective(Op op, const void *site, unsigned long long a, unsigned b, unsigned c, unsigned d);
void launch(const std::function<void()> &body, dim3 grid, dim3 block);
}  // namespace hipemu

struct hipemu_idx { unsigned x, y, z; };
#define threadIdx (hipemu_idx{hipemu::cur_lane(), 0u, 0u})
#define blockIdx (hipemu_idx{hipemu::cur_block(), hipemu::cur_block_y(), 0u})
#define blockDim (hipemu_idx{hipemu::cur_block_dim(), 1u, 1u})
#define gridDim (hipemu_idx{hipemu::cur_grid(), hipemu::cur_grid_y(), 1u})

#define hipLaunchKernelGGL(F, G, B, LDS, ST, ...) hipemu::launch([=]() { F(__VA_ARGS__); }, dim3(G), dim3(B))

// The "site" of a cross-lane operation = source file + line of the call (default arguments are evaluated at the call site; the
// code address is no identity: the optimiser duplicates calls along different paths to the same statement).
#define HIPEMU_AT const char *file_ = __builtin_FILE(), int line_ = __builtin_LINE()
#define HIPEMU_COLL static inline
static inline unsigned long long hipemu_coll(hipemu::Op op, unsigned long long a, unsigned b, unsigned c, unsigned d, const char *file_, int line_) {
  return hipemu::collective(op, (const void *)((uintptr_t)file_ + ((uintptr_t)line_ << 40)), a, b, c, d);
}
HIPEMU_COLL unsigned long long __ballot(int pred, HIPEMU_AT) { return hipemu_coll(hipemu::OP_BALLOT, pred ? 1u : 0u, 0, 0, 0, file_, line_); }
HIPEMU_COLL void __syncthreads(HIPEMU_AT) { (void)hipemu_coll(hipemu::OP_BARRIER, 0, 0, 0, 0, file_, line_); }
HIPEMU_COLL void __threadfence_block(HIPEMU_AT) { (void)hipemu_coll(hipemu::OP_BARRIER, 0, 0, 0, 0, file_, line_); }
HIPEMU_COLL void __threadfence(HIPEMU_AT) { (void)hipemu_coll(hipemu::OP_BARRIER, 0, 0, 0, 0, file_, line_); }
HIPEMU_COLL int __shfl(int v, int l, int = 64, HIPEMU_AT) { return (int)hipemu_coll(hipemu::OP_SHFL, (unsigned)v, (unsigned)l, 0, 0, file_, line_); }
HIPEMU_COLL int __shfl_xor(int v, int m, int = 64, HIPEMU_AT) { return (int)hipemu_coll(hipemu::OP_SHFL_XOR, (unsigned)v, (unsigned)m, 0, 0, file_, line_); }
HIPEMU_COLL int __shfl_up(int v, unsigned d, int = 64, HIPEMU_AT) { return (int)hipemu_coll(hipemu::OP_SHFL_UP, (unsigned)v, d, 0, 0, file_, line_); }
HIPEMU_COLL int __shfl_down(int v, unsigned d, int = 64, HIPEMU_AT) { return (int)hipemu_coll(hipemu::OP_SHFL_DOWN, (unsigned)v, d, 0, 0, file_, line_); }
HIPEMU_COLL int hipemu_readlane(int v, int l, HIPEMU_AT) { return (int)hipemu_coll(hipemu::OP_READLANE, (unsigned)v, (unsigned)l, 0, 0, file_, line_); }
HIPEMU_COLL int hipemu_readfirstlane(int v, HIPEMU_AT) { return (int)hipemu_coll(hipemu::OP_READFIRST, (unsigned)v, 0, 0, 0, file_, line_); }
HIPEMU_COLL int hipemu_bpermute(int addr, int v, HIPEMU_AT) { return (int)hipemu_coll(hipemu::OP_BPERMUTE, (unsigned)v, (unsigned)addr, 0, 0, file_, line_); }
HIPEMU_COLL int hipemu_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound, HIPEMU_AT) {
  return (int)hipemu_coll(hipemu::OP_DPP, (unsigned)src, 0, (unsigned)ctrl | ((unsigned)row_mask << 16) | ((unsigned)bank_mask << 20) | ((bound ? 1u : 0u) << 24), (unsigned)old, file_, line_);
}
HIPEMU_COLL void hipemu_wave_barrier(HIPEMU_AT) { (void)hipemu_coll(hipemu::OP_BARRIER, 0, 0, 0, 0, file_, line_); }
#define __builtin_amdgcn_readlane(v, l) hipemu_readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane((v))
#define __builtin_amdgcn_ds_bpermute(a, v) hipemu_bpermute((a), (v))
#define __builtin_amdgcn_update_dpp(o, s, c, r, b, bc) hipemu_dpp((o), (s), (c), (r), (b), (bc))
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned v) { const unsigned l = hipemu::cur_lane() & 63u; return v + (unsigned)__builtin_popcount(l >= 32u ? m : (m & ((1u << l) - 1u))); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned v) { const unsigned l = hipemu::cur_lane() & 63u; return v + (l > 32u ? (unsigned)__builtin_popcount(m & ((1u << (l - 32u)) - 1u)) : 0u); }
// global_load_lds_dword: lane l's dword lands at the (wave-uniform) LDS address + l x size; here the copy happens at once
static inline void __builtin_amdgcn_global_load_lds(const void *g, void *l, unsigned size, int off, unsigned) {
  std::memcpy((char *)l + off + (size_t)(hipemu::cur_lane() & 63u) * size, (const char *)g + off, size);
}
static inline bool __builtin_amdgcn_inverse_ballot_w64(unsigned long long m) { return (m >> (hipemu::cur_lane() & 63u)) & 1ull; }
#if !defined(__clang__)
#define __builtin_nondeterministic_value(v) (v)
#define __builtin_readcyclecounter() 0ull
#endif

// ---- scalar device functions ---------------------------------------------------------------------------------------------------
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned long long wall_clock64() { return 0; }
static inline long long clock64() { return 0; }
#define HIPEMU_MINMAX(T) static inline T min(T a, T b) { return a < b ? a : b; } static inline T max(T a, T b) { return a > b ? a : b; }
HIPEMU_MINMAX(unsigned) HIPEMU_MINMAX(int) HIPEMU_MINMAX(unsigned long long) HIPEMU_MINMAX(long long) HIPEMU_MINMAX(unsigned long) HIPEMU_MINMAX(long) HIPEMU_MINMAX(float) HIPEMU_MINMAX(double)
static inline unsigned long long min(unsigned long long a, unsigned b) { return a < b ? a : b; }
static inline unsigned long long min(unsigned a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned b) { return a > b ? a : b; }
static inline unsigned long long max(unsigned a, unsigned long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned b) { return a < b ? a : b; }
static inline unsigned long min(unsigned a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned b) { return a > b ? a : b; }
static inline unsigned long max(unsigned a, unsigned long b) { return a > b ? a : b; }

// atomics: workgroups of one launch may run on several host threads, so these are real atomics
template <class T> struct hipemu_id { typedef T type; };
#define HIPEMU_V(T) typename hipemu_id<T>::type
template <class T> static inline T atomicAdd(T *p, HIPEMU_V(T) v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicSub(T *p, HIPEMU_V(T) v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T *p, HIPEMU_V(T) v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T *p, HIPEMU_V(T) v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicXor(T *p, HIPEMU_V(T) v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T *p, HIPEMU_V(T) v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMax(T *p, HIPEMU_V(T) v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T> static inline T atomicMin(T *p, HIPEMU_V(T) v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T> static inline T atomicCAS(T *p, HIPEMU_V(T) cmp, HIPEMU_V(T) v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_max(p, v, order, scope) atomicMax((p), (v))
#define __hip_atomic_fetch_min(p, v, order, scope) atomicMin((p), (v))

#endif
