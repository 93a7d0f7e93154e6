// hipemu.cpp — the wavefront emulator behind tools/hipemu/hip/hip_runtime.h (developer / test tool, see that header).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <mutex>
#include <sys/mman.h>

thread_local __attribute__((aligned(16))) unsigned char smem[160 * 1024];
thread_local __attribute__((aligned(16))) unsigned char csmem[160 * 1024];
// kernels defined inside an anonymous namespace name `(anonymous namespace)::smem` with their block-scope extern declaration
thread_local __attribute__((aligned(16))) unsigned char hipemu_smem_anon[160 * 1024] asm("_ZN12_GLOBAL__N_14smemE");
thread_local __attribute__((aligned(16))) unsigned char hipemu_csmem_anon[160 * 1024] asm("_ZN12_GLOBAL__N_15csmemE");

// ---- context switch (x86-64 SysV): callee-saved registers on the old stack, swap stack pointers ---------------------------------
extern "C" void hipemu_switch(void **from_sp, void *to_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

static const size_t STACK_BYTES = 512 * 1024;

struct Lane {
  void *sp = nullptr;
  unsigned char *stack = nullptr;
  bool done = true, parked = false;
  Op op = OP_BARRIER; const void *site = nullptr;
  unsigned long long a = 0, result = 0; unsigned b = 0, c = 0, d = 0;
};

struct Wave {   // one workgroup: up to 1024 threads = 16 wavefronts of 64 lanes
  Lane lane[1024];
  void *sched_sp = nullptr;
  unsigned cur = 0, block = 0, grid = 0, grid_x = 1, n_lanes = 64;
  const std::function<void()> *body = nullptr;
  unsigned long long n_coll = 0;
};

static thread_local Wave *tl_wave = nullptr;
Wave *cur() { return tl_wave; }
unsigned cur_lane() { return tl_wave->cur; }   // threadIdx.x
unsigned cur_block() { return tl_wave->block % tl_wave->grid_x; }
unsigned cur_block_y() { return tl_wave->block / tl_wave->grid_x; }
unsigned cur_grid() { return tl_wave->grid_x; }
unsigned cur_grid_y() { return tl_wave->grid / tl_wave->grid_x; }
unsigned cur_block_dim() { return tl_wave->n_lanes; }

static void fiber_main() {
  Wave *w = tl_wave;
  Lane &l = w->lane[w->cur];
  (*w->body)();
  w = tl_wave;
  l.done = true;
  hipemu_switch(&l.sp, w->sched_sp);
  abort();  // a finished fiber is never resumed
}

static void fiber_init(Lane &l) {
  if (!l.stack) {
    l.stack = (unsigned char *)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (l.stack == (unsigned char *)MAP_FAILED) { perror("hipemu: mmap"); abort(); }
  }
  uintptr_t top = ((uintptr_t)l.stack + STACK_BYTES) & ~(uintptr_t)15;
  void **sp = (void **)(top - 64);      // 16-byte aligned; after the six pops and the ret the stack pointer is = 8 (mod 16), as after a call
  for (int i = 0; i < 6; i++) sp[i] = nullptr;
  sp[6] = (void *)&fiber_main;
  sp[7] = nullptr;
  l.sp = sp; l.done = false; l.parked = false;
}

unsigned long long collective(Op op, const void *site, unsigned long long a, unsigned b, unsigned c, unsigned d) {
  Wave *w = tl_wave;
  Lane &l = w->lane[w->cur];
  l.op = op; l.site = site; l.a = a; l.b = b; l.c = c; l.d = d; l.parked = true;
  hipemu_switch(&l.sp, w->sched_sp);
  return l.result;
}

// source lane of a DPP control word for lane i (64 = none)
static unsigned dpp_src(unsigned ctrl, unsigned i) {
  const unsigned row = i & ~15u, r = i & 15u;
  if (ctrl <= 0xFF) return (i & ~3u) | ((ctrl >> (2 * (i & 3u))) & 3u);                    // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const unsigned n = ctrl & 15u; return r + n < 16 ? row + r + n : 64; }   // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const unsigned n = ctrl & 15u; return r >= n ? row + r - n : 64; }       // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const unsigned n = ctrl & 15u; return row + ((r + 16 - n) & 15u); }      // row_ror
  if (ctrl == 0x130) return i + 1 < 64 ? i + 1 : 64;       // wave_shl:1
  if (ctrl == 0x134) return (i + 1) & 63u;                 // wave_rol:1
  if (ctrl == 0x138) return i >= 1 ? i - 1 : 64;           // wave_shr:1
  if (ctrl == 0x13C) return (i + 63) & 63u;                // wave_ror:1
  if (ctrl == 0x140) return row + (15 - r);                // row_mirror
  if (ctrl == 0x141) return (i & ~7u) | (7 - (i & 7u));    // row_half_mirror
  if (ctrl == 0x142) return i >= 16 ? row - 1 : 64;        // row_bcast:15 (lane 15 of the previous row)
  if (ctrl == 0x143) return i >= 32 ? 31 : 64;             // row_bcast:31
  fprintf(stderr, "hipemu: DPP control 0x%x not implemented\n", ctrl); abort();
}

static void evaluate(Lane *lanes, unsigned long long live) {
  struct { Lane *lane; } wv{lanes}, *w = &wv;
  unsigned first = 64;
  for (unsigned i = 0; i < 64; i++) if ((live >> i) & 1) { first = i; break; }
  const Lane &f = w->lane[first];
  const Op op = f.op;
  unsigned long long ballot = 0;
  if (op == OP_BALLOT) for (unsigned i = 0; i < 64; i++) if (((live >> i) & 1) && w->lane[i].a) ballot |= 1ull << i;
  auto val = [&](unsigned j) -> unsigned long long { return j < 64 && ((live >> j) & 1) ? w->lane[j].a : 0ull; };   // an inactive source reads as 0
  for (unsigned i = 0; i < 64; i++) {
    if (!((live >> i) & 1)) continue;
    Lane &l = w->lane[i];
    switch (op) {
      case OP_BALLOT: l.result = ballot; break;
      case OP_BARRIER: l.result = 0; break;
      case OP_READLANE: l.result = val(l.b & 63u); break;
      case OP_READFIRST: l.result = f.a; break;
      case OP_BPERMUTE: l.result = val((l.b >> 2) & 63u); break;
      case OP_SHFL: l.result = val(l.b & 63u); break;
      case OP_SHFL_XOR: l.result = val((i ^ l.b) & 63u); break;
      case OP_SHFL_UP: l.result = i >= l.b ? val(i - l.b) : l.a; break;
      case OP_SHFL_DOWN: l.result = i + l.b < 64 ? val(i + l.b) : l.a; break;
      case OP_DPP: {
        const unsigned ctrl = l.c & 0xFFFFu, row_mask = (l.c >> 16) & 15u, bank_mask = (l.c >> 20) & 15u, bound = (l.c >> 24) & 1u;
        const unsigned src = dpp_src(ctrl, i);
        if (!((row_mask >> (i >> 4)) & 1u) || !((bank_mask >> ((i & 15u) >> 2)) & 1u)) l.result = l.d;     // write disabled: old
        else if (src >= 64 || !((live >> src) & 1)) l.result = bound ? 0u : l.d;                               // no source: 0 (bound_ctrl) or old
        else l.result = w->lane[src].a;
      } break;
    }
  }
}

static void run_wave(Wave *w) {
  static const bool allow_divergent = getenv("HIPEMU_DIVERGENT") != nullptr;
  const unsigned nl = w->n_lanes;
  for (unsigned i = 0; i < nl; i++) fiber_init(w->lane[i]);
  for (;;) {
    // run every lane that can run until it parks or finishes
    for (unsigned i = 0; i < nl; i++) {
      Lane &l = w->lane[i];
      if (l.done || l.parked) continue;
      w->cur = i;
      hipemu_switch(&w->sched_sp, l.sp);
    }
    unsigned first = nl;
    for (unsigned i = 0; i < nl; i++) if (!w->lane[i].done && w->lane[i].parked) { first = i; break; }
    if (first == nl) return;   // all lanes finished
    // all live lanes are parked: they must be at the same operation
    const void *site = w->lane[first].site; bool uniform = true;
    for (unsigned i = 0; i < nl; i++) if (!w->lane[i].done && (w->lane[i].site != site || w->lane[i].op != w->lane[first].op)) { uniform = false; if (w->lane[i].site < site) site = w->lane[i].site; }
    if (!uniform && !allow_divergent) {
      fprintf(stderr, "hipemu: block %u: lanes are parked at different cross-lane operations (divergent control flow around a wavefront operation):\n", w->block);
      for (unsigned i = 0; i < nl; i++) if (!w->lane[i].done) {
        const uintptr_t sv = (uintptr_t)w->lane[i].site;
        fprintf(stderr, "  lane %2u op %d at %s:%u\n", i, (int)w->lane[i].op, (const char *)(sv & ((1ull << 40) - 1)), (unsigned)(sv >> 40));
      }
      abort();
    }
    for (unsigned g = 0; g < nl; g += 64) {   // wavefront by wavefront (a barrier has no data)
      unsigned long long group = 0;
      for (unsigned i = 0; i < 64 && g + i < nl; i++) if (!w->lane[g + i].done && w->lane[g + i].site == site) group |= 1ull << i;
      if (!group) continue;
      evaluate(w->lane + g, group);
      for (unsigned i = 0; i < 64; i++) if ((group >> i) & 1) w->lane[g + i].parked = false;
    }
    w->n_coll++;
  }
}

void launch(const std::function<void()> &body, dim3 grid, dim3 block) {
  const unsigned n_threads = block.x * block.y * block.z;
  if (n_threads > 1024 || block.y * block.z != 1) { fprintf(stderr, "hipemu: one-dimensional workgroups of at most 1024 threads only (got %u x %u x %u)\n", block.x, block.y, block.z); abort(); }
  const unsigned n_blocks = grid.x * grid.y;
  if (grid.z != 1) { fprintf(stderr, "hipemu: two-dimensional grids at most\n"); abort(); }
  unsigned nt = 1;
  if (const char *e = getenv("HIPEMU_THREADS")) nt = (unsigned)atoi(e); else nt = std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > n_blocks) nt = n_blocks;
  std::atomic<unsigned> next{0};
  auto worker = [&]() {
    Wave *w = new Wave();
    tl_wave = w;
    w->body = &body; w->grid = n_blocks; w->grid_x = grid.x; w->n_lanes = n_threads;
    for (;;) {
      const unsigned b = next.fetch_add(1);
      if (b >= n_blocks) break;
      w->block = b;
      run_wave(w);
    }
    for (unsigned i = 0; i < 1024; i++) if (w->lane[i].stack) munmap(w->lane[i].stack, STACK_BYTES);
    tl_wave = nullptr;
    delete w;
  };
  if (nt == 1) { worker(); return; }
  std::vector<std::thread> ts;
  for (unsigned t = 0; t < nt; t++) ts.emplace_back(worker);
  for (auto &t : ts) t.join();
}

}  // namespace hipemu
