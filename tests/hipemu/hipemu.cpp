// TEST INFRASTRUCTURE ONLY: fiber scheduler of the SIMT emulator (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <vector>

namespace hipemu {
Idx g_tid, g_bid, g_bdim, g_gdim;

enum State { RUN = 0, COLL = 1, BARRIER = 2, DONE = 3, GBARRIER = 4 };
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  State st = RUN;
  Idx tid;
  // pending collective
  int kind = 0, arg = 0;
  uint32_t value = 0;
  const void* site = nullptr;
  unsigned long long result = 0;
};
static const size_t STACK_BYTES = 4096 * 1024;
static std::vector<Fiber> g_f;
static Fiber* g_cur = nullptr;
static void* g_sched_sp = nullptr;
static const std::function<void()>* g_body = nullptr;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
#if defined(__x86_64__)
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
#else
#error "tests/hipemu needs x86-64 (hand-written context switch)"
#endif

static void yield_to_scheduler() { hipemu_switch(&g_cur->sp, g_sched_sp); }

static void fiber_main() {
  (*g_body)();
  g_cur->st = DONE;
  yield_to_scheduler();
  abort();   // a finished fiber is never resumed
}

void barrier() {
  g_cur->st = BARRIER;
  yield_to_scheduler();
}

static long g_group_barriers = 0;
extern "C" long hipemu_group_barrier_count() { return g_group_barriers; }   // threads that have passed a subset barrier (tests: was the path taken?)
void group_barrier(int count) {
  ++g_group_barriers;
  g_cur->arg = count;
  g_cur->st = GBARRIER;
  yield_to_scheduler();
}

unsigned long long collective(int kind, uint32_t value, int arg, const void* site) {
  Fiber* f = g_cur;
  f->kind = kind; f->value = value; f->arg = arg; f->site = site; f->st = COLL;
  yield_to_scheduler();
  return f->result;
}

static void resume(Fiber& f) {
  g_cur = &f;
  g_tid = f.tid;
  hipemu_switch(&g_sched_sp, f.sp);
}

// resolve the parked collectives of one wave (lanes [lo, hi)): groups = same call site and kind
static void resolve_wave(int lo, int hi) {
  for (int i = lo; i < hi; ++i) {
    if (g_f[i].st != COLL) continue;
    const void* site = g_f[i].site;
    const int kind = g_f[i].kind;
    int members[64], nm = 0;
    unsigned long long mask = 0;
    for (int j = i; j < hi; ++j)
      if (g_f[j].st == COLL && g_f[j].site == site && g_f[j].kind == kind) { members[nm++] = j; mask |= 1ull << (j - lo); }
    unsigned long long bal = 0;
    if (kind == K_BALLOT)
      for (int m = 0; m < nm; ++m) if (g_f[members[m]].value) bal |= 1ull << (members[m] - lo);
    for (int m = 0; m < nm; ++m) {
      Fiber& f = g_f[members[m]];
      const int lane = members[m] - lo;
      int src = lane;
      if (kind == K_BALLOT) { f.result = bal; continue; }
      if (kind == K_SHFL) src = f.arg & 63;
      else if (kind == K_SHFL_XOR) src = lane ^ f.arg;
      else if (kind == K_SHFL_DOWN) src = lane + f.arg;
      else if (kind == K_SHFL_UP) src = lane - f.arg;
      else if (kind == K_DPP) {
        if (f.arg >= 0 && f.arg <= 0xff) src = (lane & ~3) | ((f.arg >> (2 * (lane & 3))) & 3);   // quad_perm
        else if (f.arg == 0x141) src = (lane & ~7) | (7 - (lane & 7));                          // row_half_mirror
        else if (f.arg == 0x140) src = (lane & ~15) | (15 - (lane & 15));                       // row_mirror
        else if (f.arg > 0x110 && f.arg <= 0x11f) {                                             // row_shr:n, bound_ctrl (old = 0): no source lane inside the row -> 0
          const int sh = f.arg - 0x110;
          if ((lane & 15) < sh) { f.result = 0; continue; }
          src = lane - sh;
        }
        else { fprintf(stderr, "hipemu: dpp control 0x%x not emulated\n", f.arg); abort(); }
      }
      f.result = (src >= 0 && src < 64 && ((mask >> src) & 1ull)) ? g_f[lo + src].value : f.value;
    }
    for (int m = 0; m < nm; ++m) g_f[members[m]].st = RUN;
  }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  if (g_cur) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
  if (shmem > 160 * 1024) { fprintf(stderr, "hipemu: %zu bytes of dynamic LDS requested\n", shmem); abort(); }
  const int nt = (int)(block.x * block.y * block.z);
  g_f.assign(nt, Fiber());
  for (auto& f : g_f) {
    f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f.stack == (char*)MAP_FAILED) { perror("hipemu mmap"); abort(); }
  }
  g_body = &body;
  g_bdim = {block.x, block.y, block.z};
  g_gdim = {grid.x, grid.y, grid.z};
  const int nwave = (nt + 63) / 64;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_bid = {bx, by, bz};
        for (int t = 0; t < nt; ++t) {
          Fiber& f = g_f[t];
          f.st = RUN;
          f.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
          uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
          void** sp = (void**)top;
          *--sp = nullptr;                 // return address slot of fiber_main (never used)
          *--sp = (void*)&fiber_main;      // popped by the first `ret`
          for (int r = 0; r < 6; ++r) *--sp = nullptr;
          f.sp = sp;
        }
        for (;;) {
          bool all_done = true, any_barrier = false;
          for (int w = 0; w < nwave; ++w) {
            const int lo = w * 64, hi = lo + 64 < nt ? lo + 64 : nt;
            for (;;) {
              bool ran = false;
              for (int t = lo; t < hi; ++t)
                while (g_f[t].st == RUN) { resume(g_f[t]); ran = true; }
              bool coll = false;
              for (int t = lo; t < hi; ++t) coll |= g_f[t].st == COLL;
              if (coll) { resolve_wave(lo, hi); continue; }
              if (!ran) break;
            }
            for (int t = lo; t < hi; ++t) {
              if (g_f[t].st != DONE) all_done = false;
              if (g_f[t].st == BARRIER) any_barrier = true;
            }
          }
          if (all_done) break;
          {   // a subset barrier opens as soon as all its members have arrived (before any workgroup barrier is considered)
            int parked = 0, want = 0;
            for (auto& f : g_f) if (f.st == GBARRIER) { ++parked; want = f.arg; }
            if (parked > 0 && parked >= want) {
              for (auto& f : g_f) if (f.st == GBARRIER) f.st = RUN;
              continue;
            }
            if (parked > 0) { fprintf(stderr, "hipemu: %d of %d threads reached a subset barrier and nothing else can run\n", parked, want); abort(); }
          }
          if (!any_barrier) { fprintf(stderr, "hipemu: block (%u,%u,%u) cannot make progress\n", bx, by, bz); abort(); }
          for (auto& f : g_f) if (f.st == BARRIER) f.st = RUN;
        }
      }
  for (auto& f : g_f) munmap(f.stack, STACK_BYTES);
  g_f.clear();
  g_cur = nullptr;
  g_body = nullptr;
}
}  // namespace hipemu

// the kernels' `extern __shared__ char smem[]`
alignas(16) thread_local char smem[160 * 1024];
