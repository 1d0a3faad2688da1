// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/hipemu/hip/hip_runtime.h (see the header there).
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

#include "hip/hip_runtime.h"

namespace hipemu {

Idx g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;

namespace {

enum State { kRunnable, kWaveWait, kBlockWait, kDone };

struct Fiber {
  ucontext_t ctx;
  State state;
  Idx tid;
  int linear;
  uint64_t slot;        // value offered to the wave exchange / predicate of __syncthreads_or
  uint64_t* out;        // where the released lane reads the snapshot
  uint64_t* out_active;
};

constexpr size_t kStackBytes = 256 * 1024;
std::vector<Fiber> g_fibers;
char* g_stacks = nullptr;
size_t g_stack_count = 0;
ucontext_t g_sched;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
std::vector<unsigned char> g_dyn;
int g_block_or = 0;
hipError_t g_error = hipSuccess;

void trampoline() {
  (*g_body)();
  g_cur->state = kDone;
  swapcontext(&g_cur->ctx, &g_sched);
}

void park(State s) {
  g_cur->state = s;
  Fiber* me = g_cur;
  swapcontext(&me->ctx, &g_sched);
  g_threadIdx = me->tid;   // (the scheduler already restored it; kept for clarity)
}

void ensure_stacks(size_t n) {
  if (n <= g_stack_count) return;
  if (g_stacks) munmap(g_stacks, g_stack_count * kStackBytes);
  g_stacks = (char*)mmap(nullptr, n * kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (g_stacks == MAP_FAILED) { perror("hipemu: mmap"); abort(); }
  g_stack_count = n;
}

// returns false on deadlock
bool run_block(int nthreads) {
  const int nwaves = (nthreads + 63) / 64;
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = g_fibers[t];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * kStackBytes;
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, trampoline, 0);
    f.state = kRunnable;
    f.linear = t;
    f.tid.x = (unsigned)t % g_blockDim.x;
    f.tid.y = ((unsigned)t / g_blockDim.x) % g_blockDim.y;
    f.tid.z = (unsigned)t / (g_blockDim.x * g_blockDim.y);
  }
  int live = nthreads;
  while (live > 0) {
    bool progress = false;
    for (int t = 0; t < nthreads; ++t) {
      Fiber& f = g_fibers[t];
      if (f.state != kRunnable) continue;
      g_cur = &f;
      g_threadIdx = f.tid;
      swapcontext(&g_sched, &f.ctx);
      progress = true;
      if (f.state == kDone) --live;
    }
    if (live == 0) break;
    // every live fiber is parked now.  Wave exchanges first: the lanes of a wave that arrived form the active mask.
    bool released = false;
    for (int w = 0; w < nwaves; ++w) {
      uint64_t snap[64] = {0}, act = 0;
      const int lo = w * 64, hi = std::min(nthreads, lo + 64);
      for (int t = lo; t < hi; ++t)
        if (g_fibers[t].state == kWaveWait) { snap[t - lo] = g_fibers[t].slot; act |= 1ull << (t - lo); }
      if (!act) continue;
      for (int t = lo; t < hi; ++t) {
        Fiber& f = g_fibers[t];
        if (f.state != kWaveWait) continue;
        memcpy(f.out, snap, sizeof(snap));
        *f.out_active = act;
        f.state = kRunnable;
      }
      released = true;
    }
    if (released) continue;
    // no wave exchange pending: everything live must be at the workgroup barrier
    bool all_block = true;
    int any = 0;
    for (int t = 0; t < nthreads; ++t) {
      const Fiber& f = g_fibers[t];
      if (f.state == kDone) continue;
      if (f.state != kBlockWait) all_block = false;
      else any |= (int)(f.slot != 0);
    }
    if (all_block) {
      g_block_or = any;
      for (int t = 0; t < nthreads; ++t)
        if (g_fibers[t].state == kBlockWait) g_fibers[t].state = kRunnable;
      continue;
    }
    if (!progress) return false;
  }
  return true;
}

}  // namespace

void* dyn_lds() { return g_dyn.data(); }

void block_barrier() {
  g_cur->slot = 0;
  park(kBlockWait);
}

int block_barrier_or(int pred) {
  g_cur->slot = pred ? 1 : 0;
  park(kBlockWait);
  return g_block_or;
}

void wave_exchange(uint64_t mine, uint64_t out[64], uint64_t* active) {
  g_cur->slot = mine;
  g_cur->out = out;
  g_cur->out_active = active;
  park(kWaveWait);
}

hipError_t take_error() {
  const hipError_t e = g_error;
  g_error = hipSuccess;
  return e;
}

void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads <= 0 || nthreads > 1024) { g_error = hipErrorLaunchFailure; return; }
  ensure_stacks((size_t)nthreads);
  g_fibers.resize((size_t)nthreads);
  g_dyn.assign(dyn_lds_bytes + 64, 0xCD);   // LDS starts as garbage on the hardware: poison it
  g_blockDim = block;
  g_gridDim = grid;
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = Idx{bx, by, bz};
        if (!run_block(nthreads)) {
          fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u)\n", bx, by, bz);
          g_error = hipErrorLaunchFailure;
          return;
        }
      }
}

}  // namespace hipemu
