// tests/emu/emu.cpp -- fiber scheduler behind tests/emu/msmc_rt.hpp.  TEST INFRASTRUCTURE ONLY.
//
// One workgroup runs at a time; every work-item is a ucontext fiber.  A fiber runs until it
// reaches a workgroup barrier or a wave-collective, then yields to the round-robin scheduler,
// which releases a barrier once every live work-item (of the workgroup / of that wave) is waiting
// on it.  Divergent barriers and collectives with a partially exited wave abort with a message.
#include <msmc_rt.hpp>

#include <ucontext.h>

#include <vector>

namespace emu {
dim3 tid, bid, bdim, gdim;
char* dyn_lds = nullptr;

namespace {
enum { READY = 0, AT_BLOCK = 1, AT_WAVE = 2, DONE = 3 };
struct Fiber {
    ucontext_t ctx;
    char* stack;
    int state;
    dim3 id;
};
const size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
int cur = -1;
void (*g_body)(void*) = nullptr;
void* g_closure = nullptr;
std::vector<uint32_t> slots;        // [waves][64][8]
std::vector<char> lds_buf;

void yield_as(int st) {
    fibers[cur].state = st;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}
void entry() {
    g_body(g_closure);
    fibers[cur].state = DONE;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}
}  // namespace

int lane() { return cur & 63; }
int wave() { return cur >> 6; }
uint32_t* slot(int lane_index) { return &slots[((size_t)wave() * 64 + lane_index) * 8]; }
void block_barrier() { yield_as(AT_BLOCK); }
void wave_barrier() { yield_as(AT_WAVE); }

static void run_block(dim3 block) {
    int n = block.x * block.y * block.z;
    if (n % 64 != 0) { fprintf(stderr, "emu: block size %d is not a multiple of 64\n", n); abort(); }
    if ((int)fibers.size() < n) {
        size_t old = fibers.size();
        fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) fibers[i].stack = (char*)malloc(kStack);
    }
    slots.assign((size_t)(n / 64) * 64 * 8, 0);
    for (int i = 0; i < n; ++i) {
        Fiber& f = fibers[i];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &sched_ctx;
        makecontext(&f.ctx, (void (*)())entry, 0);
        f.state = READY;
        f.id = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    }
    for (;;) {
        bool progressed = false;
        int done = 0;
        for (int i = 0; i < n; ++i) {
            if (fibers[i].state == READY) {
                cur = i;
                tid = fibers[i].id;
                swapcontext(&sched_ctx, &fibers[i].ctx);
                progressed = true;
            }
            if (fibers[i].state == DONE) ++done;
        }
        if (done == n) break;
        // wave-level releases
        for (int w = 0; w < n / 64; ++w) {
            int waiting = 0, live = 0;
            for (int l = 0; l < 64; ++l) {
                int s = fibers[w * 64 + l].state;
                if (s != DONE) ++live;
                if (s == AT_WAVE) ++waiting;
            }
            if (waiting && waiting == live) {
                if (live != 64) { fprintf(stderr, "emu: wave collective with exited lanes\n"); abort(); }
                for (int l = 0; l < 64; ++l) fibers[w * 64 + l].state = READY;
                progressed = true;
            }
        }
        // workgroup barrier release
        int at_block = 0, live = 0;
        for (int i = 0; i < n; ++i) {
            if (fibers[i].state != DONE) ++live;
            if (fibers[i].state == AT_BLOCK) ++at_block;
        }
        if (at_block && at_block == live) {
            if (live != n) { fprintf(stderr, "emu: __syncthreads with exited work-items\n"); abort(); }
            for (int i = 0; i < n; ++i) fibers[i].state = READY;
            progressed = true;
        }
        if (!progressed) { fprintf(stderr, "emu: deadlock (divergent barrier / collective)\n"); abort(); }
    }
}

void run_grid(dim3 grid, dim3 block, size_t lds_bytes, void (*body)(void*), void* closure) {
    if (lds_bytes > 160 * 1024) { fprintf(stderr, "emu: %zu bytes of LDS requested (>160 KiB)\n", lds_bytes); abort(); }
    g_body = body;
    g_closure = closure;
    gdim = grid;
    bdim = block;
    lds_buf.assign(lds_bytes + 64, 0x7f);          // poison: kernels must not read LDS they did not write
    dyn_lds = (char*)(((uintptr_t)lds_buf.data() + 15) & ~(uintptr_t)15);
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                bid = dim3(x, y, z);
                memset(lds_buf.data(), 0x7f, lds_buf.size());
                run_block(block);
            }
}
}  // namespace emu
