// cn_emul.cpp -- runtime of the TEST-ONLY SIMT emulator (see cn_emul.h).
// One workgroup at a time; each GPU thread is a ucontext fiber scheduled round-robin.
#include "cn_emul.h"
#include <ucontext.h>
#include <vector>
#include <sys/mman.h>

namespace cn_emul {

dim3 g_tid, g_bid, g_bdim, g_gdim;

namespace {
enum State { RUNNABLE = 0, AT_BARRIER = 1, DONE = 2 };
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  State state = DONE;
  dim3 tid;
};
struct Wave {
  const void* ptrs[64];
  int arrived = 0, departed = 0, phase = 0;  // phase 0: collecting, 1: published
  unsigned long gen = 0;                     // bumped when a collective has fully drained
};
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
ucontext_t g_sched;
int g_cur = -1;
const std::function<void()>* g_body = nullptr;

void fiber_entry() {
  (*g_body)();
  g_fibers[g_cur].state = DONE;
  swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

void yield_to_sched() {
  int me = g_cur;
  swapcontext(&g_fibers[me].ctx, &g_sched);
}

void run_block(unsigned nthreads) {
  if (g_fibers.size() < nthreads) {
    size_t old = g_fibers.size();
    g_fibers.resize(nthreads);
    for (size_t i = old; i < nthreads; ++i) {
      g_fibers[i].stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE,
                                      MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
      if (g_fibers[i].stack == MAP_FAILED) { perror("mmap"); abort(); }
    }
  }
  g_waves.assign((nthreads + 63) / 64, Wave());
  for (unsigned t = 0; t < nthreads; ++t) {
    Fiber& f = g_fibers[t];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &g_sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    f.state = RUNNABLE;
    f.tid = dim3(t % g_bdim.x, (t / g_bdim.x) % g_bdim.y, t / (g_bdim.x * g_bdim.y));
  }
  unsigned done = 0;
  unsigned long idle_passes = 0;
  while (done < nthreads) {
    unsigned at_barrier = 0;
    done = 0;
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = g_fibers[t];
      if (f.state == DONE) { ++done; continue; }
      if (f.state == AT_BARRIER) { ++at_barrier; continue; }
      g_cur = (int)t;
      g_tid = f.tid;
      swapcontext(&g_sched, &f.ctx);
      if (f.state == DONE) ++done;
      else if (f.state == AT_BARRIER) ++at_barrier;
    }
    if (done < nthreads && at_barrier + done == nthreads) {
      for (unsigned t = 0; t < nthreads; ++t)
        if (g_fibers[t].state == AT_BARRIER) g_fibers[t].state = RUNNABLE;
      idle_passes = 0;
    } else if (++idle_passes > 100000000ul) {
      fprintf(stderr, "cn_emul: deadlock suspected (divergent barrier / partial-wave collective)\n");
      abort();
    }
  }
}
}  // namespace

void sync_threads() {
  g_fibers[g_cur].state = AT_BARRIER;
  yield_to_sched();
}

const void* const* wave_gather(const void* mine) {
  int me = g_cur;
  Wave& w = g_waves[me / 64];
  while (w.phase != 0) yield_to_sched();
  w.ptrs[me % 64] = mine;
  unsigned nthreads = g_bdim.x * g_bdim.y * g_bdim.z;
  int wave_size = (int)std::min<unsigned>(64u, nthreads - (me / 64) * 64u);
  if (++w.arrived == wave_size) w.phase = 1;
  while (w.phase != 1) yield_to_sched();
  return w.ptrs;
}

void wave_release() {
  int me = g_cur;
  Wave& w = g_waves[me / 64];
  unsigned nthreads = g_bdim.x * g_bdim.y * g_bdim.z;
  int wave_size = (int)std::min<unsigned>(64u, nthreads - (me / 64) * 64u);
  // Payloads live on the lanes' fiber stacks: nobody may leave before everybody has read.
  unsigned long my_gen = w.gen;
  if (++w.departed == wave_size) { w.arrived = 0; w.departed = 0; w.phase = 0; ++w.gen; }
  while (w.gen == my_gen) yield_to_sched();
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  g_gdim = grid;
  g_bdim = block;
  g_body = &body;
  unsigned nthreads = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_bid = dim3(bx, by, bz);
        run_block(nthreads);
      }
  g_body = nullptr;
}

}  // namespace cn_emul
