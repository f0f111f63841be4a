// simt_emu.h — TEST INFRASTRUCTURE ONLY.  A tiny lock-step SIMT emulator so that the SAME kernel
// sources under streamyolo_amd/csrc/ can be compiled for the host (clang++, -DSY_EMU) and their
// indexing / predication / epilogue logic checked against the CPU oracle in the GPU-less build
// container.  It is never loaded by the product package, never measured, and does not model
// timing, caches or memory ordering.  What it models:
//   * a grid of blocks; each block's threads are cooperative fibers (hand-rolled x86-64 context
//     switch) scheduled round-robin on one OS thread, so __syncthreads() and wave collectives are
//     exact rendez-vous points;  blocks are spread over a few OS worker threads;
//   * `__shared__` = `static thread_local` (one live block per OS thread);
//   * wave64 collectives (shuffles, ballot) and the gfx950 MFMA shapes the kernels use, with the
//     lane->element maps documented in /opt/skills/guides/cdna_hip_programming.md §3:
//       32x32xK A/B: lane l supplies row/col (l&31), k-group (l>>5)
//       32x32   C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16)
#pragma once
#if !defined(__x86_64__)
#error "simt_emu.h supports x86-64 hosts only"
#endif
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

namespace emu {

struct Barrier { int expected = 0, count = 0; unsigned gen = 0; };

struct Wave {
    Barrier bar;
    alignas(16) unsigned char xchg[64][64];      // per-lane exchange slots for collectives
};

struct Fiber {
    void* sp = nullptr;
    unsigned char* stack = nullptr;
    bool done = false;
    dim3 tid;
};

struct Block {
    dim3 bid, bdim, gdim;
    Barrier bar;
    std::vector<Wave> waves;
    std::vector<Fiber> fibers;
    int cur = 0;
    void* sched_sp = nullptr;
    const std::function<void()>* body = nullptr;
    unsigned char* dyn_smem = nullptr;
};

extern thread_local Block* g_blk;

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);

inline Fiber& self() { return g_blk->fibers[g_blk->cur]; }
inline void yield() { Block* b = g_blk; emu_ctx_switch(&b->fibers[b->cur].sp, b->sched_sp); }

inline void barrier_wait(Barrier& b) {
    unsigned gen = b.gen;
    if (++b.count == b.expected) { b.count = 0; ++b.gen; return; }
    while (b.gen == gen) yield();
}
// A finished thread no longer takes part in barriers (mirrors hardware: exited waves don't block s_barrier).
inline void barrier_drop(Barrier& b) {
    --b.expected;
    if (b.expected > 0 && b.count == b.expected) { b.count = 0; ++b.gen; }
}

// coresident: every workgroup of the launch runs at the same time (one OS thread each) — for kernels whose workgroups wait for
// one another (a bounded grid that is fully resident on the device)
void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body, bool coresident = false);

inline int lane_id() { const dim3& t = self().tid; Block* b = g_blk; return (int)((t.x + t.y * b->bdim.x + t.z * b->bdim.x * b->bdim.y) & 63); }
inline int wave_id() { const dim3& t = self().tid; Block* b = g_blk; return (int)((t.x + t.y * b->bdim.x + t.z * b->bdim.x * b->bdim.y) >> 6); }

// publish `n` bytes from this lane, rendez-vous, let `f(slots)` read every lane's bytes, rendez-vous.
template <typename F>
inline void wave_exchange(const void* mine, size_t n, F&& f) {
    Wave& w = g_blk->waves[wave_id()];
    std::memcpy(w.xchg[lane_id()], mine, n);
    barrier_wait(w.bar);
    f(w.xchg);
    barrier_wait(w.bar);
}

}  // namespace emu

#define threadIdx (emu::self().tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)

static inline void __syncthreads() { emu::barrier_wait(emu::g_blk->bar); }

// ---- wave64 collectives ---------------------------------------------------------------------
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    T out;
    int lane = emu::lane_id();
    int base = lane & ~(width - 1);
    emu::wave_exchange(&v, sizeof(T), [&](unsigned char (*s)[64]) { std::memcpy(&out, s[base + (src & (width - 1))], sizeof(T)); });
    return out;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    return __shfl(v, (emu::lane_id() & (width - 1)) ^ mask, width);
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = emu::lane_id() & (width - 1);
    return __shfl(v, (l + (int)d < width) ? l + (int)d : l, width);
}
template <typename T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = emu::lane_id() & (width - 1);
    return __shfl(v, (l - (int)d >= 0) ? l - (int)d : l, width);
}
static inline unsigned long long __ballot(int pred) {
    unsigned long long out = 0;
    unsigned char p = pred ? 1 : 0;
    int nl = emu::g_blk->waves[emu::wave_id()].bar.expected;   // live lanes
    (void)nl;
    emu::wave_exchange(&p, 1, [&](unsigned char (*s)[64]) {
        for (int i = 0; i < 64; ++i) if (s[i][0]) out |= (1ull << i);
    });
    return out;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

// ---- atomics (blocks may run on different OS threads) -----------------------------------------
static inline float atomicAdd(float* p, float v) {
    auto* a = reinterpret_cast<std::atomic<float>*>(p);
    float old = a->load(std::memory_order_relaxed);
    while (!a->compare_exchange_weak(old, old + v, std::memory_order_relaxed)) {}
    return old;
}
static inline int atomicAdd(int* p, int v) { return reinterpret_cast<std::atomic<int>*>(p)->fetch_add(v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return reinterpret_cast<std::atomic<unsigned>*>(p)->fetch_add(v); }
static inline int atomicMax(int* p, int v) {
    auto* a = reinterpret_cast<std::atomic<int>*>(p);
    int old = a->load();
    while (old < v && !a->compare_exchange_weak(old, v)) {}
    return old;
}
static inline unsigned atomicOr(unsigned* p, unsigned v) { return reinterpret_cast<std::atomic<unsigned>*>(p)->fetch_or(v); }


// ---- MFMA (gfx950 shapes used by the kernels) ---------------------------------------------------
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));

static inline float emu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
static inline float emu_f16_to_f32(unsigned short h) { _Float16 x; std::memcpy(&x, &h, 2); return (float)x; }

// a, b: 8 x 16-bit elements per lane (k = 8*(lane>>5) + j);  kind 0 = bf16, 1 = f16
static inline emu_f32x16 emu_mfma_32x32x16(uint4 a, uint4 b, emu_f32x16 c, int kind) {
    struct Slot { uint4 a, b; } mine{a, b};
    emu_f32x16 d = c;
    int lane = emu::lane_id();
    emu::wave_exchange(&mine, sizeof(mine), [&](unsigned char (*s)[64]) {
        int col = lane & 31, hi = lane >> 5;
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            double acc = 0.0;
            for (int h = 0; h < 2; ++h) {
                Slot sa, sb;
                std::memcpy(&sa, s[row + 32 * h], sizeof(Slot));
                std::memcpy(&sb, s[col + 32 * h], sizeof(Slot));
                unsigned short ea[8], eb[8];
                std::memcpy(ea, &sa.a, 16);
                std::memcpy(eb, &sb.b, 16);
                for (int j = 0; j < 8; ++j) {
                    float fa = kind ? emu_f16_to_f32(ea[j]) : emu_bf16_to_f32(ea[j]);
                    float fb = kind ? emu_f16_to_f32(eb[j]) : emu_bf16_to_f32(eb[j]);
                    acc += (double)fa * (double)fb;
                }
            }
            d[r] = (float)((double)c[r] + acc);
        }
    });
    return d;
}
// f32-input 32x32x2: a = A[row=lane&31][k=lane>>5], b = B[k=lane>>5][col=lane&31]; k-ordered fmaf chain.
static inline emu_f32x16 emu_mfma_32x32x2_f32(float a, float b, emu_f32x16 c) {
    struct Slot { float a, b; } mine{a, b};
    emu_f32x16 d = c;
    int lane = emu::lane_id();
    emu::wave_exchange(&mine, sizeof(mine), [&](unsigned char (*s)[64]) {
        int col = lane & 31, hi = lane >> 5;
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float acc = c[r];
            for (int k = 0; k < 2; ++k) {
                Slot sa, sb;
                std::memcpy(&sa, s[row + 32 * k], sizeof(Slot));
                std::memcpy(&sb, s[col + 32 * k], sizeof(Slot));
                acc = std::fmaf(sa.a, sb.b, acc);
            }
            d[r] = acc;
        }
    });
    return d;
}
