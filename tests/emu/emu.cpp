// tests/emu/emu.cpp — TEST-ONLY SIMT emulator: runs a HIP-style kernel on the CPU, one cooperative fiber per
// GPU thread, blocks one after another.  __syncthreads() and the wave64 collectives (__ballot, __shfl_xor,
// wave barrier) are rendezvous points among the live fibers of the block / wave.  See hip/hip_runtime.h.
#include "hip/hip_runtime.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <mutex>
#include <vector>

namespace emu {

enum State { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

// Context switches.  glibc's swapcontext saves and restores the signal mask - two system calls per switch, and a kernel with
// one collective per few instructions switches millions of times (the serial CPU suite spent 6 of its 16 minutes in the kernel).
// On x86-64 without a sanitizer the fibers therefore switch with twenty lines of assembly (callee-saved registers + stack pointer;
// nothing here touches the signal mask, MXCSR or the x87 control word); sanitizer builds keep ucontext, which ASan knows about.
// ThreadSanitizer has to be told: every emulator fiber is a TSan fiber (__tsan_create_fiber), announced before each swapcontext,
// with flags 0 = "the switch synchronises" - the fibers of a launch are one logical thread of the host thread that holds
// g_launch_mtx.  In the ucontext build a fiber lives for the whole process and loops over the kernels it is given, so that the
// call stack TSan shadows stays balanced (a fiber re-made per block would leak a few shadow frames per block and overflow).
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__)
#define EMU_ASM_SWITCH 1
struct EmuCtx { void* rsp = nullptr; };
extern "C" void emu_switch(EmuCtx* from, EmuCtx* to);
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
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");
#else
#define EMU_ASM_SWITCH 0
#endif
#if defined(__SANITIZE_THREAD__)
#define EMU_TSAN 1
extern "C" void* __tsan_get_current_fiber(void);
extern "C" void* __tsan_create_fiber(unsigned flags);
extern "C" void __tsan_switch_to_fiber(void* fiber, unsigned flags);
#else
#define EMU_TSAN 0
#endif

struct Fiber {
#if EMU_ASM_SWITCH
    EmuCtx ctx;
#else
    ucontext_t ctx;
    bool made = false;          // the context exists and is parked at the end of fiber_entry's loop
#endif
#if EMU_TSAN
    void* tsan = nullptr;
#endif
    char* stack = nullptr;
    int state = DONE;
    emu_uint3 tid{0, 0, 0};
    int linear = 0;
    unsigned gen = 0;
};

Fiber* g_cur = nullptr;
emu_uint3 g_blockIdx{0, 0, 0};
dim3 g_blockDim, g_gridDim;

static const size_t kStack = 256 * 1024;
static std::vector<Fiber> g_fibers;
#if EMU_ASM_SWITCH
static EmuCtx g_sched;
#else
static ucontext_t g_sched;
#endif
#if EMU_TSAN
static void* g_sched_tsan = nullptr;        // the launching host thread, as TSan sees it
#endif
// fiber -> scheduler and scheduler -> fiber
static inline void to_sched(Fiber* f) {
#if EMU_ASM_SWITCH
    emu_switch(&f->ctx, &g_sched);
#else
#if EMU_TSAN
    __tsan_switch_to_fiber(g_sched_tsan, 0);
#endif
    swapcontext(&f->ctx, &g_sched);
#endif
}
static inline void to_fiber(Fiber& f) {
#if EMU_ASM_SWITCH
    emu_switch(&g_sched, &f.ctx);
#else
#if EMU_TSAN
    __tsan_switch_to_fiber(f.tsan, 0);
#endif
    swapcontext(&g_sched, &f.ctx);
#endif
}
static const std::function<void()>* g_body = nullptr;
// per wave exchange buffers, double buffered by the per-fiber generation counter
static std::vector<uint64_t> g_xbuf;   // [wave][2][64]

emu_uint3 cur_tid() { return g_cur->tid; }
int cur_lane() { return g_cur->linear & 63; }

static void fiber_entry() {
#if !EMU_ASM_SWITCH
    for (;;) {              // ucontext build: the fiber is parked in to_sched() below and resumed with the next block's body
        (*g_body)();
        Fiber* f = g_cur;
        f->state = DONE;    // (see the note on finished lanes below)
        to_sched(f);
    }
#endif
    (*g_body)();
    Fiber* f = g_cur;
    // A lane that leaves the kernel must read as "not there" in every later collective of its wave - but NOT yet in the one its
    // siblings are still reading: a lane that returns right after a readlane used to zero its exchange entry before the lanes
    // scheduled behind it had fetched it (k_xtc_wave: the walk of a wave that does not own the last tile ends on a readlane; found
    // by scripts/fuzz_xtc.py seed 11).  The entries of finished lanes are cleared where a rendezvous is released (run_block).
    f->state = DONE;
    to_sched(f);
    abort();            // a finished fiber is never resumed
}

static void yield(State s) {
    Fiber* f = g_cur;
    f->state = s;
    to_sched(f);
}

void sync_block() { yield(WAIT_BLOCK); }
void sync_wave() { yield(WAIT_WAVE); }

static uint64_t* xslot(Fiber* f, unsigned gen) { return &g_xbuf[(size_t)((f->linear >> 6) * 2 + (gen & 1)) * 64]; }

unsigned long long ballot(int pred) {
    Fiber* f = g_cur;
    const unsigned gen = f->gen++;
    uint64_t* s = xslot(f, gen);
    s[f->linear & 63] = pred ? 1 : 0;
    yield(WAIT_WAVE);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (s[l]) m |= 1ull << l;
    return m;
}

uint32_t shfl_xor_bits(uint32_t v, int mask) {
    Fiber* f = g_cur;
    const unsigned gen = f->gen++;
    uint64_t* s = xslot(f, gen);
    s[f->linear & 63] = v;
    yield(WAIT_WAVE);
    return (uint32_t)s[(f->linear & 63) ^ (mask & 63)];
}

uint32_t shfl_idx_bits(uint32_t v, int src) {
    Fiber* f = g_cur;
    const unsigned gen = f->gen++;
    uint64_t* s = xslot(f, gen);
    s[f->linear & 63] = v;
    yield(WAIT_WAVE);
    return (uint32_t)s[src & 63];
}

uint32_t readfirstlane_bits(uint32_t v) {
    Fiber* f = g_cur;
    const unsigned gen = f->gen++;
    uint64_t* s = xslot(f, gen);
    s[f->linear & 63] = (uint64_t)v | (1ull << 63);      // bit 63 marks a live, participating lane
    yield(WAIT_WAVE);
    uint32_t r = v;
    for (int l = 0; l < 64; ++l) if (s[l] >> 63) { r = (uint32_t)s[l]; break; }
    // the slot is reused two collectives later; ballots/shuffles overwrite their own lane before reading
    return r;
}

static void run_block(int nthreads) {
    const int nwaves = (nthreads + 63) / 64;
    g_xbuf.assign((size_t)nwaves * 2 * 64, 0);
    if ((int)g_fibers.size() < nthreads) {
        g_fibers.reserve(1024);         // the launch limit: a parked ucontext must never move (it points into itself)
        const size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < g_fibers.size(); ++i) {
            g_fibers[i].stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (g_fibers[i].stack == MAP_FAILED) { fprintf(stderr, "emu: mmap failed\n"); abort(); }
        }
    }
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = g_fibers[t];
        f.state = RUNNABLE;
        f.linear = t;
        f.gen = 0;
        f.tid.x = t % g_blockDim.x;
        f.tid.y = (t / g_blockDim.x) % g_blockDim.y;
        f.tid.z = t / (g_blockDim.x * g_blockDim.y);
#if EMU_ASM_SWITCH
        // first switch into the fiber: six callee-saved registers are popped, then `ret` enters fiber_entry with the stack pointer
        // where a call would have left it (8 modulo 16, a return address nobody uses on top)
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                    // fake return address of fiber_entry
        *--sp = (void*)&fiber_entry;
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.ctx.rsp = sp;
#else
        if (!f.made) {
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = &g_sched;
            makecontext(&f.ctx, fiber_entry, 0);
            f.made = true;
#if EMU_TSAN
            f.tsan = __tsan_create_fiber(0);
#endif
        }
#endif
    }
    for (;;) {
        bool ran = false;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = g_fibers[t];
            if (f.state != RUNNABLE) continue;
            g_cur = &f;
            to_fiber(f);
            ran = true;
        }
        // release rendezvous points
        bool released = false, all_done = true, block_ready = true;
        for (int t = 0; t < nthreads; ++t) {
            const int s = g_fibers[t].state;
            if (s != DONE) all_done = false;
            if (s != DONE && s != WAIT_BLOCK) block_ready = false;
        }
        if (all_done) break;
        if (block_ready) {
            for (int t = 0; t < nthreads; ++t) if (g_fibers[t].state == WAIT_BLOCK) { g_fibers[t].state = RUNNABLE; released = true; }
        }
        for (int w = 0; w < nwaves; ++w) {
            bool ready = true, any = false;
            const int end = (w + 1) * 64 < nthreads ? (w + 1) * 64 : nthreads;
            for (int t = w * 64; t < end; ++t) {
                const int s = g_fibers[t].state;
                if (s == WAIT_WAVE) any = true;
                else if (s != DONE) ready = false;
            }
            if (ready && any) {
                // every live lane has written its entry of this collective and read the previous one: finished lanes drop out of both
                for (int t = w * 64; t < end; ++t)
                    if (g_fibers[t].state == DONE) { g_xbuf[(size_t)(w * 2 + 0) * 64 + (t & 63)] = 0; g_xbuf[(size_t)(w * 2 + 1) * 64 + (t & 63)] = 0; }
                for (int t = w * 64; t < end; ++t) if (g_fibers[t].state == WAIT_WAVE) { g_fibers[t].state = RUNNABLE; released = true; }
            }
        }
        if (!ran && !released) {
            fprintf(stderr, "emu: deadlock (divergent barrier / collective) in block (%u,%u,%u)\n", g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
            abort();
        }
    }
}

static std::vector<uint64_t> g_dynshm;
void* dyn_shared() { return g_dynshm.data(); }

// One kernel at a time, process-wide: the fibers, the exchange buffers and the kernels' static __shared__ arrays are globals.  Host
// threads that drive different evals concurrently (full + filtered evaluation) therefore take turns launch by launch.
static std::mutex g_launch_mtx;

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mtx);
    // the hardware's launch limits (a grid.y above 65535 went unnoticed here and would fail on the GPU)
    if (grid.y > 65535u || grid.z > 65535u || grid.x > 2147483647u || block.x * block.y * block.z > 1024u || shmem > 160u * 1024u ||
        grid.x == 0 || grid.y == 0 || grid.z == 0) {
        fprintf(stderr, "emu: launch outside the hardware limits: grid (%u, %u, %u), block (%u, %u, %u), %zu bytes of LDS\n", grid.x, grid.y,
                grid.z, block.x, block.y, block.z, shmem);
        abort();
    }
    g_dynshm.assign(shmem / 8 + 2, 0);
#if EMU_TSAN
    g_sched_tsan = __tsan_get_current_fiber();
#endif
    g_body = &body;
    g_gridDim = grid;
    g_blockDim = block;
    const int nthreads = (int)(block.x * block.y * block.z);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = emu_uint3{bx, by, bz};
                run_block(nthreads);
            }
    g_body = nullptr;
}

}  // namespace emu
