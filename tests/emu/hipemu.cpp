// TEST INFRASTRUCTURE ONLY -- fiber-based SIMT interpreter runtime (see
// include/hip/hip_runtime.h in this directory).  x86-64 SysV only.
#include <hip/hip_runtime.h>
#include <vector>
#include <sys/mman.h>

namespace hipemu {

Fiber* g_cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
long g_progress = 0;

static void* g_sched_sp = nullptr;
static std::vector<Fiber> g_fibers;
static std::vector<WaveXchg> g_waves;
static const std::function<void()>* g_body = nullptr;
static int g_live = 0;
static int g_bar_count = 0;
static long g_bar_gen = 0;
static constexpr size_t kStack = 256 * 1024;

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
.size hipemu_switch,.-hipemu_switch
)");

void yield() {
  Fiber* f = g_cur;
  hipemu_switch(&f->sp, g_sched_sp);
}

static void fiber_entry() {
  (*g_body)();
  g_cur->done = true;
  g_live--;
  g_progress++;
  for (;;) yield();
}

void barrier() {
  long gen = g_bar_gen;
  g_bar_count++;
  g_progress++;
  while (g_bar_gen == gen) {
    if (g_bar_count >= g_live) {  // last arriver releases
      g_bar_count = 0;
      g_bar_gen++;
      g_progress++;
      break;
    }
    yield();
  }
}

WaveXchg* cur_wave() { return &g_waves[g_cur->wave]; }

// per-lane collective sequence number (all lanes of a wave execute the same
// sequence of collectives); parity selects the buffer.
static std::vector<long> g_lane_seq;

unsigned char (*wave_exchange(const void* src, int bytes))[64] {
  Fiber* f = g_cur;
  WaveXchg* w = &g_waves[f->wave];
  long seq = g_lane_seq[f->linear]++;
  int p = (int)(seq & 1);
  memcpy(w->buf[p][f->lane], src, bytes);
  w->arrived[p]++;
  g_progress++;
  // wait until every lane of the wave has deposited for THIS sequence number.
  // arrived[p] counts deposits for seq parity p; it is reset by the first lane
  // that enters the collective two steps later (safe: that lane can only get
  // there after all lanes completed step seq+1, hence finished reading buf[p]).
  while (w->arrived[p] < w->nlanes) yield();
  // mark consumption: when a lane starts collective seq+1 it resets the
  // counter of parity (seq+1)&1 lazily below.
  int q = p ^ 1;
  // first lane to pass resets the *other* parity counter for the next use
  if (w->arrived[q] >= w->nlanes) w->arrived[q] = 0;
  return w->buf[p];
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  int T = (int)(block.x * block.y * block.z);
  if (T <= 0 || T > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", T); abort(); }
  if ((int)g_fibers.size() < T) {
    size_t old = g_fibers.size();
    g_fibers.resize(T);
    for (size_t i = old; i < (size_t)T; ++i) {
      g_fibers[i].stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE,
                                      MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (g_fibers[i].stack == (char*)MAP_FAILED) { perror("mmap"); abort(); }
    }
  }
  int nw = (T + kWave - 1) / kWave;
  g_waves.assign(nw, WaveXchg());
  g_lane_seq.assign(T, 0);
  g_blockDim = block;
  g_gridDim = grid;
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        for (int w = 0; w < nw; ++w) {
          g_waves[w].arrived[0] = g_waves[w].arrived[1] = 0;
          g_waves[w].nlanes = std::min(kWave, T - w * kWave);
        }
        std::fill(g_lane_seq.begin(), g_lane_seq.end(), 0);
        g_live = T;
        g_bar_count = 0;
        for (int t = 0; t < T; ++t) {
          Fiber& f = g_fibers[t];
          f.linear = t;
          f.wave = t / kWave;
          f.lane = t % kWave;
          f.done = false;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
          void** sp = (void**)top;
          *(--sp) = nullptr;               // fake return address of fiber_entry
          *(--sp) = (void*)&fiber_entry;   // popped by `ret` in hipemu_switch
          for (int r = 0; r < 6; ++r) *(--sp) = nullptr;  // rbp rbx r12-r15
          f.sp = (void*)sp;
        }
        long stall_rounds = 0;
        while (g_live > 0) {
          long before = g_progress;
          for (int t = 0; t < T; ++t) {
            Fiber& f = g_fibers[t];
            if (f.done) continue;
            g_cur = &f;
            hipemu_switch(&g_sched_sp, f.sp);
          }
          if (g_progress == before) {
            if (++stall_rounds > 4) {
              fprintf(stderr, "hipemu: deadlock (divergent barrier/collective?) block=(%u,%u,%u)\n", bx, by, bz);
              abort();
            }
          } else {
            stall_rounds = 0;
          }
        }
      }
  g_cur = nullptr;
  g_body = nullptr;
}

}  // namespace hipemu
