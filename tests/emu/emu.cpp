// tests/emu/emu.cpp -- fiber scheduler behind tests/emu/msmc_rt.hpp.  TEST INFRASTRUCTURE ONLY.
//
// One workgroup runs at a time; every work-item is a fiber with its own stack (hand-rolled x86-64
// context switch: callee-saved registers + stack pointer, no signal-mask syscalls).  A fiber runs
// until it reaches a workgroup barrier or a wave-collective, then yields to the round-robin
// scheduler, which releases a barrier once every live work-item (of the workgroup / of that wave)
// is waiting on it.  Divergent barriers and collectives with a partially exited wave abort.
#include <msmc_rt.hpp>

#include <vector>

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

namespace emu {
dim3 tid, bid, bdim, gdim;
char* dyn_lds = nullptr;

namespace {
enum { READY = 0, AT_BLOCK = 1, AT_WAVE = 2, DONE = 3 };
struct Fiber {
    void* sp;
    char* stack;
    int state;
    int ncoll;          // collectives completed by this lane (selects the exchange-slot parity)
    dim3 id;
};
const size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
void* sched_sp = nullptr;
int cur = -1;
void (*g_body)(void*) = nullptr;
void* g_closure = nullptr;
std::vector<uint32_t> slots;        // [waves][2 parities][64][32]  (128-byte slots)
std::vector<int> computed;          // per wave: collective count whose group computation is done
std::vector<char> lds_buf;

void yield_as(int st) {
    fibers[cur].state = st;
    emu_switch(&fibers[cur].sp, sched_sp);
}
void entry() {
    g_body(g_closure);
    fibers[cur].state = DONE;
    emu_switch(&fibers[cur].sp, sched_sp);
    abort();
}
}  // namespace

int lane() { return cur & 63; }
int wave() { return cur >> 6; }
// Exchange slots are double-buffered per wave: a lane can be at most one collective ahead of the
// slowest lane of its wave, so one barrier per collective suffices.
uint32_t* slot(int lane_index) {
    return &slots[(((size_t)wave() * 2 + (fibers[cur].ncoll & 1)) * 64 + lane_index) * 32];
}
void collective_done() { ++fibers[cur].ncoll; }
bool first_after_barrier() {
    const int w = wave(), id = fibers[cur].ncoll + 1;
    if (computed[w] == id) return false;
    computed[w] = id;
    return true;
}
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
    slots.assign((size_t)(n / 64) * 2 * 64 * 32, 0);
    computed.assign(n / 64, 0);
    for (int i = 0; i < n; ++i) {
        Fiber& f = fibers[i];
        // initial frame: 6 callee-saved registers (popped by emu_switch) then the return address
        uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);           // so that rsp % 16 == 8 on entry, as after a call
        *--sp = (void*)entry;
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.sp = sp;
        f.state = READY;
        f.ncoll = 0;
        f.id = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    }
    for (;;) {
        bool progressed = false;
        int done = 0;
        for (int i = 0; i < n; ++i) {
            if (fibers[i].state == READY) {
                cur = i;
                tid = fibers[i].id;
                emu_switch(&sched_sp, fibers[i].sp);
                progressed = true;
            }
            if (fibers[i].state == DONE) ++done;
        }
        if (done == n) break;
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
