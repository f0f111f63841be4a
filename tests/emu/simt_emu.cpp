// simt_emu.cpp — runtime of the test-only SIMT emulator (see simt_emu.h).
#include "simt_emu.h"

#include <atomic>
#include <condition_variable>
#include <memory>
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

// Fiber stacks and dynamic LDS of one worker thread: allocated once per thread and reused by every launch, UNINITIALISED (a
// zero-filled vector of (threads + 1) x 192 KiB per worker per launch — 50-200 MB — was ~95 % of the emulator's run time, all
// of it page faults in the kernel).
struct Scratch {
    std::unique_ptr<unsigned char[]> stacks, smem;
    size_t stacks_n = 0, smem_n = 0;
    unsigned char* stack_base(size_t n) {
        if (n > stacks_n) { stacks.reset(new unsigned char[n]); stacks_n = n; }
        return stacks.get();
    }
    unsigned char* smem_base(size_t n) {
        if (n > smem_n) { smem.reset(new unsigned char[n]); smem_n = n; }
        return smem.get();
    }
};
static thread_local Scratch t_scratch;

static void run_block(Block& blk, unsigned char* stacks) {
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
        unsigned char* top = stacks + (size_t)(i + 1) * kStack;
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

// Persistent worker pool: the threads (and their thread-local scratch) live for the process; a launch hands them one job.
namespace {
struct Pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable wake, done;
    const std::function<void()>* job = nullptr;
    unsigned long long gen = 0;
    unsigned active = 0, pending = 0;
    bool stop = false;

    void loop(unsigned idx) {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            wake.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            if (idx >= active) continue;
            const std::function<void()>* j = job;
            lk.unlock();
            (*j)();
            lk.lock();
            if (--pending == 0) done.notify_all();
        }
    }
    void run(unsigned n, const std::function<void()>& fn) {
        std::unique_lock<std::mutex> lk(m);
        while (threads.size() < n) {
            const unsigned idx = (unsigned)threads.size();
            threads.emplace_back([this, idx] { loop(idx); });
        }
        job = &fn;
        active = n;
        pending = n;
        ++gen;
        wake.notify_all();
        done.wait(lk, [&] { return pending == 0; });
        job = nullptr;
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        wake.notify_all();
        for (auto& t : threads) t.join();
    }
};
Pool& pool() {
    static Pool p;
    return p;
}
std::mutex g_launch_mutex;      // one emulated launch at a time per process (launches from several host threads serialise)
}  // namespace

void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body, bool coresident) {
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    const unsigned nthr = block.x * block.y * block.z;
    unsigned nworkers = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("SY_EMU_THREADS")) nworkers = (unsigned)std::atoi(e);
    if (nworkers < 1) nworkers = 1;
    if (nworkers > nblocks || coresident) nworkers = (unsigned)nblocks;
    std::atomic<unsigned long long> next{0};
    const std::function<void()> worker = [&]() {
        unsigned char* const stacks = t_scratch.stack_base((size_t)(nthr + 1) * kStack);
        unsigned char* const smem = t_scratch.smem_base(dyn_smem + 64);
        Block blk;
        blk.bdim = block;
        blk.gdim = grid;
        blk.body = &body;
        blk.dyn_smem = (unsigned char*)(((uintptr_t)smem + 63) & ~(uintptr_t)63);
        for (;;) {
            unsigned long long b = next.fetch_add(1);
            if (b >= nblocks) break;
            blk.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y)));
            run_block(blk, stacks);
        }
    };
    if (nworkers == 1) { worker(); return; }
    // (a co-resident launch: each worker takes exactly one block only if all of them start — they do, the pool grows to nworkers;
    //  a worker that finishes early finds the block counter exhausted)
    std::lock_guard<std::mutex> lk(g_launch_mutex);
    pool().run(nworkers, worker);
}

}  // namespace emu
