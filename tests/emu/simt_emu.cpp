// simt_emu.cpp — runtime of the test-only SIMT emulator (see simt_emu.h).
#include "simt_emu.h"

#include <mutex>

namespace emu {

thread_local Block* g_blk = nullptr;

// x86-64 SysV cooperative context switch: callee-saved registers + stack pointer.
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
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
.size emu_ctx_switch, .-emu_ctx_switch
)");

static void fiber_main() {
    Block* b = g_blk;
    (*b->body)();
    // re-read: the body ran on this OS thread the whole time
    b = g_blk;
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    int linear = (int)(f.tid.x + f.tid.y * b->bdim.x + f.tid.z * b->bdim.x * b->bdim.y);
    barrier_drop(b->bar);
    barrier_drop(b->waves[linear >> 6].bar);
    void* dummy;
    emu_ctx_switch(&dummy, b->sched_sp);
    std::fprintf(stderr, "simt_emu: resumed a finished fiber\n");
    std::abort();
}

static constexpr size_t kStack = 192 * 1024;

static void run_block(Block& blk, std::vector<unsigned char>& stacks) {
    const unsigned nthr = blk.bdim.x * blk.bdim.y * blk.bdim.z;
    blk.fibers.assign(nthr, Fiber());
    blk.waves.assign((nthr + 63) / 64, Wave());
    blk.bar = Barrier();
    blk.bar.expected = (int)nthr;
    for (unsigned w = 0; w < blk.waves.size(); ++w) {
        unsigned lanes = nthr - w * 64 < 64 ? nthr - w * 64 : 64;
        blk.waves[w].bar.expected = (int)lanes;
    }
    for (unsigned i = 0; i < nthr; ++i) {
        Fiber& f = blk.fibers[i];
        f.tid = dim3(i % blk.bdim.x, (i / blk.bdim.x) % blk.bdim.y, i / (blk.bdim.x * blk.bdim.y));
        unsigned char* top = stacks.data() + (size_t)(i + 1) * kStack;
        top = (unsigned char*)((uintptr_t)top & ~(uintptr_t)15);
        void** sp = (void**)top;
        *--sp = nullptr;                       // fake return address of fiber_main (keeps rsp%16==8 at entry)
        *--sp = (void*)&fiber_main;            // popped by `ret`
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.sp = (void*)sp;
    }
    g_blk = &blk;
    unsigned long long spins = 0;
    for (;;) {
        bool any = false;
        for (unsigned i = 0; i < nthr; ++i) {
            if (blk.fibers[i].done) continue;
            any = true;
            blk.cur = (int)i;
            emu_ctx_switch(&blk.sched_sp, blk.fibers[i].sp);
        }
        if (!any) break;
        if (++spins > 50000000ull) {
            std::fprintf(stderr, "simt_emu: block (%u,%u,%u) made no progress (barrier deadlock?)\n",
                         blk.bid.x, blk.bid.y, blk.bid.z);
            std::abort();
        }
    }
    g_blk = nullptr;
}

void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body) {
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    const unsigned nthr = block.x * block.y * block.z;
    unsigned nworkers = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("SY_EMU_THREADS")) nworkers = (unsigned)std::atoi(e);
    if (nworkers < 1) nworkers = 1;
    if (nworkers > nblocks) nworkers = (unsigned)nblocks;
    std::atomic<unsigned long long> next{0};
    auto worker = [&]() {
        std::vector<unsigned char> stacks((size_t)(nthr + 1) * kStack);
        std::vector<unsigned char> smem(dyn_smem + 64);
        Block blk;
        blk.bdim = block;
        blk.gdim = grid;
        blk.body = &body;
        blk.dyn_smem = (unsigned char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
        for (;;) {
            unsigned long long b = next.fetch_add(1);
            if (b >= nblocks) break;
            blk.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y)));
            run_block(blk, stacks);
        }
    };
    if (nworkers == 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nworkers; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
}

}  // namespace emu
