// hipemu runtime: cooperative fibers per workgroup, workgroups spread over OS threads.
// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <atomic>
#include <thread>
#include <vector>

extern "C" void hipemu_switch(void** from_sp, void* to_sp);
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
)");

namespace hipemu {

static constexpr size_t kStack = 256 * 1024;
static constexpr size_t kDynSmem = 160 * 1024;

struct Fiber {
  Ctx ctx;
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
};

struct WaveState {
  uint32_t slots[2][64][4];
  int arrived = 0;
  int gen = 0;
  int released = 0;
  int rgen = 0;
};

struct Worker {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  void* main_sp = nullptr;
  int nthreads = 0;
  int bar_arrived = 0;
  int bar_gen = 0;
  int live = 0;
  long progress = 0;
  const std::function<void()>* body = nullptr;
  alignas(16) char dyn[kDynSmem];
};

thread_local Ctx* cur = nullptr;
static thread_local Worker* W = nullptr;
static thread_local Fiber* curf = nullptr;

char* dyn_smem() { return W->dyn; }

static void yield_to_main() {
  Fiber* f = curf;
  hipemu_switch(&f->sp, W->main_sp);
}

static void fiber_entry() {
  Fiber* f = curf;
  (*W->body)();
  f->done = true;
  W->live--;
  yield_to_main();
  abort();
}

void block_barrier() {
  Worker* w = W;
  const int my = w->bar_gen;
  if (++w->bar_arrived == w->live) {
    w->bar_arrived = 0;
    w->bar_gen++;
    w->progress++;
    return;
  }
  while (w->bar_gen == my) yield_to_main();
}

void wave_exchange(const uint32_t* mine, int n, uint32_t (*all)[4]) {
  Worker* w = W;
  WaveState& ws = w->waves[cur->wave];
  const int my = ws.gen;
  uint32_t(*slot)[4] = ws.slots[my & 1];
  for (int i = 0; i < n; ++i) slot[cur->lane][i] = mine[i];
  if (++ws.arrived == 64) {
    ws.arrived = 0;
    ws.gen++;
    w->progress++;
  } else {
    while (ws.gen == my) yield_to_main();
  }
  memcpy(all, slot, sizeof(uint32_t) * 64 * 4);
}
void wave_exchange_done() {}

static void run_block(Worker* w, dim3 bid, dim3 grid, dim3 block) {
  const int n = block.x * block.y * block.z;
  w->nthreads = n;
  w->live = n;
  w->bar_arrived = 0;
  for (auto& ws : w->waves) { ws.arrived = 0; }
  for (int t = 0; t < n; ++t) {
    Fiber& f = w->fibers[t];
    f.done = false;
    f.ctx.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    f.ctx.bid = bid;
    f.ctx.bdim = block;
    f.ctx.gdim = grid;
    f.ctx.lane = t & 63;
    f.ctx.wave = t >> 6;
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
    void** base = reinterpret_cast<void**>(top - 64);
    for (int i = 0; i < 6; ++i) base[i] = nullptr;
    base[6] = reinterpret_cast<void*>(&fiber_entry);
    base[7] = nullptr;
    f.sp = base;
  }
  long spins = 0;
  while (w->live > 0) {
    const int before = w->live;
    const long gen_before = w->progress;
    for (int t = 0; t < n; ++t) {
      Fiber& f = w->fibers[t];
      if (f.done) continue;
      curf = &f;
      cur = &f.ctx;
      hipemu_switch(&w->main_sp, f.sp);
    }
    if (w->live == before && w->progress == gen_before) {
      if (++spins > 1000) {
        fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): a barrier or wave collective was not reached "
                        "by all threads\n", bid.x, bid.y, bid.z);
        abort();
      }
    } else {
      spins = 0;
    }
  }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const int n = block.x * block.y * block.z;
  if (n % 64 != 0 || n > 1024 || shmem > kDynSmem) {
    fprintf(stderr, "hipemu: unsupported launch (block %d threads, %zu B dynamic LDS)\n", n, shmem);
    abort();
  }
  const long total = long(grid.x) * grid.y * grid.z;
  unsigned hw = std::thread::hardware_concurrency();
  const char* env = getenv("HIPEMU_THREADS");
  if (env) hw = atoi(env);
  const int nthreads = int(std::max<long>(1, std::min<long>(hw ? hw : 1, total)));
  std::atomic<long> next{0};
  auto worker_fn = [&]() {
    Worker* w = new Worker;
    w->fibers.resize(n);
    w->waves.resize(n / 64);
    for (auto& f : w->fibers) {
      f.stack = static_cast<char*>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
      if (f.stack == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    }
    w->body = &body;
    W = w;
    for (;;) {
      const long b = next.fetch_add(1);
      if (b >= total) break;
      dim3 bid(unsigned(b % grid.x), unsigned((b / grid.x) % grid.y), unsigned(b / (long(grid.x) * grid.y)));
      run_block(w, bid, grid, block);
    }
    for (auto& f : w->fibers) munmap(f.stack, kStack);
    W = nullptr;
    delete w;
  };
  if (nthreads == 1) {
    worker_fn();
  } else {
    std::vector<std::thread> ts;
    for (int i = 0; i < nthreads; ++i) ts.emplace_back(worker_fn);
    for (auto& t : ts) t.join();
  }
}

}  // namespace hipemu
