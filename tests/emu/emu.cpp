// tests/emu/emu.cpp — TEST INFRASTRUCTURE: the scheduler behind tests/emu/hip/hip_runtime.h.
//
// A kernel launch runs its workgroups one after the other; the threads of a workgroup are cooperative fibers on the calling OS thread
// (hand-written x86-64 context switch: a rendezvous of 64 lanes costs ~2 us).  A lane runs until its next cross-lane operation and
// waits there for the other live lanes of its wavefront (or workgroup, for __syncthreads); lanes that have left the kernel no longer
// take part.  Lanes of a wavefront that meet at DIFFERENT kinds of operation, or a sweep in which nobody can run, abort with a
// description: that is divergent control flow around a cross-lane operation, which the kernels here never do on purpose.
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <time.h>

#include <mutex>
#include <vector>

namespace emu {

Idx g_thread, g_block, g_grid, g_bdim;

namespace {

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

constexpr size_t kStack = 512 * 1024;
struct Wave { uint64_t buf[64]; unsigned arrived = 0, live = 0; uint64_t gen = 0; int kind = 0; void* bt[24]; int nbt = 0; unsigned first = 0; };
struct Fiber { void* sp = nullptr; bool done = false; const uint64_t* wait = nullptr; uint64_t wait_gen = 0; int kind = 0; };

std::vector<Fiber> fibers;
std::vector<Wave> waves;
Wave blockw;                                    // __syncthreads
char* stacks = nullptr; size_t stacks_n = 0;
void* sched_sp = nullptr;
unsigned cur = 0, nthreads = 0;
const std::function<void()>* body = nullptr;
unsigned char* lds = nullptr; size_t lds_bytes = 0;
bool reverse_order = false;

void yield() { emu_switch(&fibers[cur].sp, sched_sp); }

void fiber_exit() {
    Fiber& f = fibers[cur]; f.done = true;
    waves[cur >> 6].buf[cur & 63] = 0;       // (a lane that has left contributes nothing to later ballots)
    Wave& w = waves[cur >> 6];
    w.live--; if (w.live && w.arrived == w.live) { w.arrived = 0; w.gen++; }
    blockw.live--; if (blockw.live && blockw.arrived == blockw.live) { blockw.arrived = 0; blockw.gen++; }
    yield();
    abort();                                    // never resumed
}
void trampoline() { (*body)(); fiber_exit(); }

void rendezvous(Wave& w, int kind) {
    Fiber& f = fibers[cur];
    if (w.arrived == 0) { w.kind = kind; w.first = cur; w.nbt = backtrace(w.bt, 24); }
    else if (w.kind != kind) {
        void* bt[24]; const int n = backtrace(bt, 24);
        fprintf(stderr, "emu: thread %u arrived first at:\n", w.first); backtrace_symbols_fd(w.bt, w.nbt, 2);
        fprintf(stderr, "emu: thread %u is at:\n", cur); backtrace_symbols_fd(bt, n, 2); fprintf(stderr, "emu: lanes of one wavefront meet at different cross-lane operations (%d vs %d), block %u thread %u\n", w.kind, kind, g_block.x, cur); abort(); }
    const uint64_t gen = w.gen;
    if (++w.arrived == w.live) { w.arrived = 0; w.gen++; return; }
    f.wait = &w.gen; f.wait_gen = gen; f.kind = kind;
    yield();
    f.wait = nullptr;
}

void run_block() {
    const unsigned nw = (nthreads + 63) / 64;
    fibers.assign(nthreads, Fiber()); waves.assign(nw, Wave());
    for (unsigned i = 0; i < nthreads; i++) {
        waves[i >> 6].live++;
        char* top = stacks + (size_t)(i + 1) * kStack;
        void** sp = (void**)top;
        *--sp = nullptr;                         // fake return address of the trampoline
        *--sp = (void*)&trampoline;
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        fibers[i].sp = sp;
    }
    blockw = Wave(); blockw.live = nthreads;
    unsigned live = nthreads;
    while (live) {
        bool ran = false;
        for (unsigned k = 0; k < nthreads; k++) {
            const unsigned i = reverse_order ? nthreads - 1 - k : k;
            Fiber& f = fibers[i];
            if (f.done) continue;
            if (f.wait && *f.wait == f.wait_gen) continue;
            cur = i; g_thread.x = i; g_thread.y = g_thread.z = 0;
            emu_switch(&sched_sp, f.sp);
            ran = true;
            if (f.done) live--;
        }
        if (!ran) {
            fprintf(stderr, "emu: deadlock in block %u: no lane can run; waiting lanes:", g_block.x);
            for (unsigned i = 0; i < nthreads; i++) if (!fibers[i].done) fprintf(stderr, " %u(kind %d)", i, fibers[i].kind);
            fprintf(stderr, "\n"); abort();
        }
    }
}

}  // namespace

unsigned char* dyn_lds() { return lds; }

static void on_segv(int sig, siginfo_t* si, void*) {              // a wild access inside a kernel: say where (block, lane, address) before dying
    fprintf(stderr, "emu: signal %d at address %p in block %u thread %u (%u threads per block, dynamic LDS %p + %zu); call stack:\n", sig, si->si_addr, g_block.x, cur, nthreads, (void*)lds, lds_bytes);
    void* bt[32]; const int n = backtrace(bt, 32); backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}
void launch(const std::function<void()>& b, dim3 grid, dim3 block, size_t dyn_bytes) {
    static bool init = false;
    if (!init) {
        init = true; const char* o = getenv("BVG_EMU_ORDER"); reverse_order = o && !strcmp(o, "rev");
        static char altstack[1 << 16];
        stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0; sigaltstack(&ss, nullptr);
        struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO | SA_ONSTACK; sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr);
    }
    static std::mutex one_launch;                // host threads (the NodeIterator's helpers, bvg_scan_multi) launch one after the other: the lanes' state is global
    std::lock_guard<std::mutex> lk(one_launch);
    if (body) { fprintf(stderr, "emu: nested launch\n"); abort(); }
    nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || grid.x == 0) return;
    if (nthreads > stacks_n) {
        if (stacks) munmap(stacks, stacks_n * kStack);
        stacks_n = nthreads; stacks = (char*)mmap(nullptr, stacks_n * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == MAP_FAILED) { perror("emu: mmap"); abort(); }
    }
    // the dynamic LDS of this launch: exactly as many bytes as the launch asked for, so that AddressSanitizer sees an access beyond them
    lds_bytes = dyn_bytes ? dyn_bytes : 16; lds = (unsigned char*)aligned_alloc(16, (lds_bytes + 15) & ~(size_t)15);
    body = &b;
    g_grid = {grid.x, grid.y, grid.z}; g_bdim = {block.x, block.y, block.z};
    for (unsigned z = 0; z < grid.z; z++) for (unsigned y = 0; y < grid.y; y++) for (unsigned x = 0; x < grid.x; x++) {
        g_block = {x, y, z};
        memset(lds, 0xA5, lds_bytes);            // LDS is not zeroed between workgroups
        run_block();
    }
    body = nullptr;
    free(lds); lds = nullptr;
}

uint64_t wave_gather(uint64_t v, const uint64_t** all) {
    Wave& w = waves[cur >> 6];
    w.buf[cur & 63] = v;
    rendezvous(w, 1);
    *all = w.buf;
    return v;
}
void wave_release() { rendezvous(waves[cur >> 6], 2); }
void wave_barrier() { rendezvous(waves[cur >> 6], 3); }
void block_barrier() { rendezvous(blockw, 4); }
uint64_t clock() { static uint64_t c = 0; return c += 16; }

}  // namespace emu

// tests: the next `n` device allocations fail (hipErrorOutOfMemory), as on a card whose memory another handle holds
static int g_fail_mallocs = 0;
extern "C" void emu_fail_next_mallocs(int n) { g_fail_mallocs = n; }
hipError_t emu_malloc(void** p, size_t n) {
    *p = nullptr;
    if (g_fail_mallocs > 0) { g_fail_mallocs--; return hipErrorOutOfMemory; }
    if (posix_memalign(p, 256, std::max<size_t>(n, 1)) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
    memset(*p, 0xCD, n);                        // device memory is not zeroed
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); e->t = (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((double)(b->t - a->t) * 1e-6); return hipSuccess; }
