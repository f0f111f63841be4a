// sy_device.h — device-side vocabulary shared by every kernel in this directory.
//
// Product build: hipcc --offload-arch=gfx950 (CDNA4 only: wave64, MFMA 32x32x16 bf16/f16,
// MFMA 32x32x2 f32).  There is deliberately no other GPU back end.
// Test build (-DSY_EMU, host clang++): tests/emu/simt_emu.h supplies a lock-step emulation of the
// few HIP constructs used here so the same sources can be checked on CPU against the oracle.
#pragma once

#ifdef SY_EMU
#include "simt_emu.h"
#define SY_DYN_SMEM(name) unsigned char* name = emu::g_blk->dyn_smem
#include "sy_tape.h"
#include <tuple>
#define SY_LAUNCH(kernel, grid, block, smem, stream, ...)                                                            \
    do {                                                                                                             \
        const dim3 sy_grid_ = (grid), sy_block_ = (block);                                                           \
        const size_t sy_smem_ = (smem);                                                                              \
        auto sy_args_ = std::make_tuple(__VA_ARGS__);          /* argument VALUES, evaluated now */                  \
        auto sy_launch_fn_ = [=](void*) {                                                                            \
            emu::launch(sy_grid_, sy_block_, sy_smem_,                                                               \
                        [=]() { std::apply([](auto... sy_a_) { kernel(sy_a_...); }, sy_args_); });                   \
        };                                                                                                           \
        if (sy_tape_recording()) sy_tape_push(std::function<void(void*)>(sy_launch_fn_));                            \
        sy_launch_fn_(nullptr);                                                                                      \
    } while (0)
#define SY_LAUNCH_OK() 0
#else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#define SY_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#include "sy_tape.h"
#include <tuple>
// Every launch is a by-value closure (kernel, grid, block, dynamic LDS, argument VALUES — all evaluated at the launch
// site, nothing is re-evaluated later): executed now and, while a recording is open on this thread, appended to the
// native launch tape (sy_tape.h) for sy_tape_replay.
#define SY_LAUNCH(kernel, grid, block, smem, stream, ...)                                                            \
    do {                                                                                                             \
        const dim3 sy_grid_ = (grid), sy_block_ = (block);                                                           \
        const size_t sy_smem_ = (smem);                                                                              \
        auto sy_args_ = std::make_tuple(__VA_ARGS__);                                                                \
        auto sy_launch_fn_ = [=](void* sy_stream_) {                                                                 \
            hipEvent_t const sy_stop_ = (hipEvent_t)sy_tape_stop_event_take();   /* a dependency follows this launch */ \
            std::apply([&](auto... sy_a_) {                                                                          \
                if (sy_stop_ != nullptr)                                                                             \
                    hipExtLaunchKernelGGL(kernel, sy_grid_, sy_block_, (std::uint32_t)sy_smem_, (hipStream_t)sy_stream_,  \
                                          nullptr, sy_stop_, 0u, sy_a_...);                                          \
                else                                                                                                 \
                    hipLaunchKernelGGL(kernel, sy_grid_, sy_block_, sy_smem_, (hipStream_t)sy_stream_, sy_a_...);    \
            }, sy_args_);                                                                                            \
        };                                                                                                           \
        if (sy_tape_recording()) sy_tape_push(std::function<void(void*)>(sy_launch_fn_));                            \
        sy_launch_fn_((void*)(stream));                                                                              \
    } while (0)
#define SY_LAUNCH_OK() ((int)hipGetLastError())
#endif

#include <stdint.h>
#include <atomic>

// "done once" flag of a per-kernel host-side setting (hipFuncSetAttribute: the opt-in for > 64 KiB of dynamic LDS) — PER DEVICE: one
// process may drive several GPUs, and the attribute set on one is not set on the others (a process-wide `static bool`
// left the second device's launches of the 102 KB tiles failing with SY_ERR_LAUNCH).  Threads may race: the call is idempotent.
inline int& sy_dev_once_current() { static thread_local int d = -1; return d; }      // device seen by this thread's last need()
struct sy_dev_once {
    std::atomic<unsigned long long> mask{0ull};
    bool need() {
#ifdef SY_EMU
        return false;
#else
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) { sy_dev_once_current() = -1; return true; }   // unknown device: set it every time
        sy_dev_once_current() = d;
        return ((mask.load(std::memory_order_relaxed) >> d) & 1ull) == 0ull;
#endif
    }
    void mark() { const int d = sy_dev_once_current(); if (d >= 0) mask.fetch_or(1ull << d, std::memory_order_relaxed); }
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- element types --------------------------------------------------------------------------
// Storage dtypes of activations / packed weights.  Codes are part of the C ABI (include/streamyolo_hip.h).
enum : int { SY_BF16 = 0, SY_F16 = 1, SY_F32 = 2 };

struct BF16 {   // bfloat16 storage; arithmetic is always fp32
    typedef unsigned short elem;
    static constexpr int kCode = SY_BF16;
    static constexpr int kEPC = 8;     // elements per 16-byte chunk
    static __device__ __forceinline__ float to_f32(elem h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
    static __device__ __forceinline__ elem from_f32(float f) {        // round-to-nearest-even
#ifdef SY_EMU
        unsigned u = __builtin_bit_cast(unsigned, f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (elem)((u >> 16) | 0x40);   // quiet NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        return (elem)(u >> 16);
#else
        // the hardware conversion (v_cvt_pk_bf16_f32, same rounding): the integer formulation above costs ~10 VALU
        // instructions and two exec-mask branches PER ELEMENT, and the BatchNorm / pooling / resize kernels convert every
        // element they store
        return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
    }
    // two fp32 -> packed pair, round-to-nearest-even: v_cvt_pk_bf16_f32 on gfx950 (one VALU op for two elements)
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
#ifdef SY_EMU
        return (unsigned)from_f32(a) | ((unsigned)from_f32(b) << 16);
#else
        typedef __bf16 sy_bf2 __attribute__((ext_vector_type(2)));
        typedef float sy_f2 __attribute__((ext_vector_type(2)));
        const sy_f2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sy_bf2));
#endif
    }
};
struct F16 {
    typedef unsigned short elem;
    static constexpr int kCode = SY_F16;
    static constexpr int kEPC = 8;
    static __device__ __forceinline__ float to_f32(elem h) { return (float)__builtin_bit_cast(_Float16, h); }
    static __device__ __forceinline__ elem from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)from_f32(a) | ((unsigned)from_f32(b) << 16); }
};
struct F32 {
    typedef float elem;
    static constexpr int kCode = SY_F32;
    static constexpr int kEPC = 4;
    static __device__ __forceinline__ float to_f32(elem h) { return h; }
    static __device__ __forceinline__ elem from_f32(float f) { return f; }
    static __device__ __forceinline__ unsigned pack2(float, float) { return 0u; }   // 16-bit staging only
};

// ---- MFMA wrappers ------------------------------------------------------------------------------
// One 64-byte K-slab per operand row is consumed as two 32-byte groups g; within a group lane-half
// h = lane>>5 owns bytes [16h, 16h+16).  For 16-bit types that is one 32x32x16 MFMA per group; for
// fp32 it is four 32x32x2 MFMAs (element j of each half).  A and B use the same (h, j) -> k map,
// so the contraction is exact whatever the hardware's internal k order is.
#ifdef SY_EMU
static inline f32x16 sy_mfma_group(BF16, uint4 a, uint4 b, f32x16 c) { return emu_mfma_32x32x16(a, b, c, 0); }
static inline f32x16 sy_mfma_group(F16, uint4 a, uint4 b, f32x16 c) { return emu_mfma_32x32x16(a, b, c, 1); }
static inline f32x16 sy_mfma_group(F32, uint4 a, uint4 b, f32x16 c) {
    float fa[4], fb[4];
    __builtin_memcpy(fa, &a, 16);
    __builtin_memcpy(fb, &b, 16);
    for (int j = 0; j < 4; ++j) c = emu_mfma_32x32x2_f32(fa[j], fb[j], c);
    return c;
}
#else
typedef __bf16 sy_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sy_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 sy_mfma_group(BF16, uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sy_bf16x8, a), __builtin_bit_cast(sy_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 sy_mfma_group(F16, uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sy_f16x8, a), __builtin_bit_cast(sy_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 sy_mfma_group(F32, uint4 a, uint4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), c, 0, 0, 0);
    return c;
}
#endif

// ---- LDS-DMA (global -> LDS without a VGPR round trip) and counted waits ----------------------------
// sy_glds16: every lane copies 16 bytes from its own global address to (wave-uniform LDS base + lane*16).
// The copy is asynchronous on hardware: it retires on the VM counter, so the consumer does
// sy_wait_vmcnt<N>() (N = loads allowed to stay in flight) and then a raw barrier before reading.
#ifdef SY_EMU
static inline void sy_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_memcpy(lds_wave_base + emu::lane_id() * 16, gsrc, 16);
}
template <int N> static inline void sy_wait_vmcnt() {}
static inline void sy_barrier() { __syncthreads(); }
static inline void sy_barrier_lds() { __syncthreads(); }

#else
// Issued through inline asm on purpose: when hipcc sees an LDS-DMA it cannot prove disjoint from a later
// ds_read it drains the whole VM queue (s_waitcnt vmcnt(0)) in front of that read, which serialises a
// multi-stage ring.  Hidden in asm, the loads are ours to count (guide §5.7): M0 carries the wave-uniform
// LDS destination, is saved/restored inside the statement, and completion is awaited with sy_wait_vmcnt.
__device__ __forceinline__ void sy_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
template <int N> __device__ __forceinline__ void sy_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void sy_barrier() { __builtin_amdgcn_s_barrier(); }
// Barrier behind ordinary LDS stores (ds_write): the raw s_barrier above does not drain the LGKM counter, and gfx950's back-off
// barrier means hipcc adds no wait of its own in front of it — a wave could pass the barrier while its stores are still in
// flight and another wave read stale LDS.  sy_barrier() is for DMA-filled LDS (paired with sy_wait_vmcnt) only.
__device__ __forceinline__ void sy_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
#endif

// ---- LDS transpose read (ds_read_b64_tr_b16) ---------------------------------------------------------
// Per 16-lane group: lane i supplies the 8-byte-aligned address of 4 consecutive 16-bit elements (row i>>2,
// column quad i&3 of a [4 rows][16 cols] block whose row pitch the addresses imply) and receives COLUMN i:
// element j of the result = block[row j][col i].  A K-major MFMA fragment straight out of a row-major
// [pixel][channel] LDS image — the transpose the weight-gradient GEMM needs (its contraction index is the pixel).
#ifdef SY_EMU
static inline uint2 sy_lds_read_tr16(const unsigned char* p) {
    unsigned long long mine;
    __builtin_memcpy(&mine, p, 8);
    uint2 out = make_uint2(0u, 0u);
    const int lane = emu::lane_id(), G = lane >> 4, i = lane & 15;
    emu::wave_exchange(&mine, 8, [&](unsigned char (*s)[64]) {
        unsigned short e[4];
        for (int j = 0; j < 4; ++j) __builtin_memcpy(&e[j], s[G * 16 + j * 4 + (i >> 2)] + (i & 3) * 2, 2);
        __builtin_memcpy(&out, e, 8);
    });
    return out;
}
#else
__device__ __forceinline__ uint2 sy_lds_read_tr16(const unsigned char* p) {
    typedef short sy_v4s __attribute__((ext_vector_type(4)));
    const sy_v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sy_v4s*)p);
    return __builtin_bit_cast(uint2, v);
}
#endif

// ---- bounds-checked 16-byte buffer loads -------------------------------------------------------------
// A buffer descriptor (base, extent) + a 32-bit byte offset per lane: the address math of a gather is one
// integer add, and an offset >= extent (we use 0xFFFFFFFF) returns zeros — convolution padding for free.
#ifdef SY_EMU
struct sy_buffer { const unsigned char* base; unsigned extent; };
static inline sy_buffer sy_make_buffer(const void* p, unsigned extent) { return sy_buffer{(const unsigned char*)p, extent}; }
static inline uint4 sy_buffer_load16(const sy_buffer& b, unsigned voff) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned long long)voff + 16ull <= (unsigned long long)b.extent) __builtin_memcpy(&v, b.base + voff, 16);
    return v;
}
// voffset per lane + wave-uniform soffset (the hardware adds the scalar for free; range check on the sum)
static inline uint4 sy_buffer_load16_s(const sy_buffer& b, unsigned voff, unsigned soff) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned long long)voff + soff + 16ull <= (unsigned long long)b.extent) __builtin_memcpy(&v, b.base + voff + soff, 16);
    return v;
}
static inline void sy_glds16_buf(const sy_buffer& b, unsigned voff, unsigned char* lds_wave_base) {
    const uint4 v = sy_buffer_load16(b, voff);
    __builtin_memcpy(lds_wave_base + emu::lane_id() * 16, &v, 16);
}
#else
typedef __amdgpu_buffer_rsrc_t sy_buffer;
typedef unsigned int sy_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sy_buffer sy_make_buffer(const void* p, unsigned extent) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, extent, 0x00020000);
}
__device__ __forceinline__ uint4 sy_buffer_load16(const sy_buffer& b, unsigned voff) {
    sy_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 sy_buffer_load16_s(const sy_buffer& b, unsigned voff, unsigned soff) {
    sy_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// LDS-DMA through a buffer descriptor: out-of-range lanes deposit zeros.  Inline asm for the same reason as
// sy_glds16 (the loads must stay invisible to hipcc's conservative vmcnt(0) before ds_read).
__device__ __forceinline__ void sy_glds16_buf(const sy_buffer& b, unsigned voff, unsigned char* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(b), "s"(dst)
                 : "memory");
}
#endif

// LDS-DMA with the destination given as (wave-uniform LDS base, byte offset): the base is resolved to an LDS address
// once per kernel (sy_lds_base), so the per-load destination math is scalar adds — no generic->LDS pointer cast
// (and its null check) in the loop.
#ifdef SY_EMU
typedef unsigned char* sy_lds_base_t;
static inline sy_lds_base_t sy_lds_base(unsigned char* p) { return p; }
static inline void sy_glds16_buf_at(const sy_buffer& b, unsigned voff, sy_lds_base_t base, unsigned off) {
    sy_glds16_buf(b, voff, base + off);
}
#else
typedef unsigned sy_lds_base_t;
__device__ __forceinline__ sy_lds_base_t sy_lds_base(unsigned char* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)p);
}
__device__ __forceinline__ void sy_glds16_buf_at(const sy_buffer& b, unsigned voff, sy_lds_base_t base, unsigned off) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + off);
    // M0 as a register-constrained input: the compiler loads it (and knows it is live), no save / restore around the DMA
    asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(b), "{m0}"(dst) : "memory");
}
#endif

// Wave-level ordering point for wave-PRIVATE LDS hand-offs (one lane writes, another lane of the same wave reads): the
// hardware executes a wave's LDS instructions in order, so nothing is needed beyond keeping the compiler from reordering;
// the host emulator runs lanes as fibers and needs a real rendezvous of the wave.
#ifdef SY_EMU
static inline void sy_wave_fence() { (void)__shfl(0, 0); }
#else
__device__ __forceinline__ void sy_wave_fence() { __builtin_amdgcn_wave_barrier(); }
#endif

// Sum of x over the 32 lanes that share lane >> 5 (the pixel columns of one MFMA accumulator half), valid in the lanes with
// (lane & 16) != 0.  Five DPP adds on the VALU (quad swaps, half-row and row mirrors, then lane 15 of rows 0 / 2 broadcast into
// rows 1 / 3) instead of five __shfl_xor = ds_bpermute round trips through the LDS pipeline: the conv epilogue's BatchNorm
// statistics reduce 32 values per wave and tile, and the bpermute version was 40 % of the 1x1 kernels' time
// (profiles/r02/m_stats_ablation.txt).
#ifdef SY_EMU
static inline float sy_sum32_upper(float x) {
    for (int off = 1; off < 32; off <<= 1) x += __shfl_xor(x, off);
    return x;
}
#else
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sy_add_dpp(float x) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, true);
    return x + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float sy_sum32_upper(float x) {
    x = sy_add_dpp<0xB1, 0xF>(x);        // quad_perm [1, 0, 3, 2]
    x = sy_add_dpp<0x4E, 0xF>(x);        // quad_perm [2, 3, 0, 1]
    x = sy_add_dpp<0x141, 0xF>(x);       // row_half_mirror
    x = sy_add_dpp<0x140, 0xF>(x);       // row_mirror: every lane of a 16-lane row holds the row's sum
    x = sy_add_dpp<0x142, 0xA>(x);       // row_bcast15 into rows 1 and 3: + the sum of the row below
    return x;
}
#endif

// Sum of EACH of 16 registers over the 32 lanes that share lane >> 5, as a value-halving butterfly: at stage b (lane bit b) a lane
// keeps one register of every pair and adds its partner lane's copy of it, so 16 registers become 8, 4, 2, 1 — afterwards lane l holds
// the total of register (l & 15) (both 16-lane rows of the half hold the same 16 totals).  24 + 12 + 10 + 5 VALU instructions and
// one cross-row exchange instead of 16 x 5 DPP adds (sy_sum32_upper per register): the BatchNorm statistics of the conv epilogue
// reduce 32 registers per wave and tile, 12-20 % of the forward convolutions' time before (profiles/r04, statistics ablation).
#ifdef SY_EMU
static inline float sy_reduce16_over32(float (&v)[16]) {
    const int lane = emu::lane_id();
    float w[16];
    for (int i = 0; i < 16; ++i) w[i] = v[i];
    int n = 16;
    for (int b = 0; b < 4; ++b) {
        const bool hi = (lane >> b) & 1;
        for (int m = 0; m < n / 2; ++m) {
            const float A = w[2 * m], B = w[2 * m + 1];
            const float send = hi ? A : B, keep = hi ? B : A;
            w[m] = keep + __shfl_xor(send, 1 << b);
        }
        n /= 2;
    }
    return w[0] + __shfl_xor(w[0], 16);
}
#else
template <int CTRL, int BANK>
__device__ __forceinline__ float sy_mov_dpp(float old, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, 0xF, BANK, false));
}
__device__ __forceinline__ float sy_reduce16_over32(float (&v)[16]) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) {                               // lane bit 0: partner = quad_perm [1, 0, 3, 2]
        const float A = v[2 * m], B = v[2 * m + 1];
        w8[m] = (b0 ? B : A) + sy_mov_dpp<0xB1, 0xF>(0.0f, b0 ? A : B);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {                               // lane bit 1: quad_perm [2, 3, 0, 1]
        const float A = w8[2 * m], B = w8[2 * m + 1];
        w4[m] = (b1 ? B : A) + sy_mov_dpp<0x4E, 0xF>(0.0f, b1 ? A : B);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {                               // lane bit 2: lanes 0-3 / 8-11 read lane + 4 (row_shl:4), the others lane - 4
        const float A = w4[2 * m], B = w4[2 * m + 1];
        const float send = b2 ? A : B;
        w2[m] = (b2 ? B : A) + sy_mov_dpp<0x114, 0xA>(sy_mov_dpp<0x104, 0x5>(0.0f, send), send);
    }
    const float A = w2[0], B = w2[1];                           // lane bit 3: lanes 0-7 read lane + 8 (row_shl:8), lanes 8-15 lane - 8
    const float send = b3 ? A : B;
    const float r = (b3 ? B : A) + sy_mov_dpp<0x118, 0xC>(sy_mov_dpp<0x108, 0x3>(0.0f, send), send);
    return r + __shfl_xor(r, 16);                               // the other 16-lane row of this half (one ds_bpermute)
}
#endif

// Scheduling fence: the compiler keeps the instruction order of a hand-pipelined loop body on both sides of it (nothing
// is moved across); no instruction is emitted.
#ifdef SY_EMU
static inline void sy_sched_fence() {}
#else
__device__ __forceinline__ void sy_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

// wave-uniform value hint (lets hipcc keep per-wave constants in SGPRs and branch on them with SALU)
#ifdef SY_EMU
static inline int sy_uniform(int v) { return v; }
#else
__device__ __forceinline__ int sy_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// 64-bit value of lane `l` (wave-uniform l): two v_readlane_b32 instead of the LDS-routed ds_bpermute of __shfl
#ifdef SY_EMU
static inline unsigned long long sy_readlane64(unsigned long long v, int l) { return __shfl(v, l); }
#else
__device__ __forceinline__ unsigned long long sy_readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
#endif

// ---- write-through 16-byte global store ---------------------------------------------------------------------------------------------
// A kernel boundary on an in-order stream costs 1.2-1.3 us — unless the finished kernel leaves dirty lines in the eight (mutually
// non-coherent) L2s: then the end-of-kernel release writes them back first, 2.6-3.2 us from ~17 MB written (tools/probes/
// chain_gap_probe.hip, profiles/r06 stage N).  Every launch of the training step writes 10-150 MB.  `sc1` makes the store write
// through to the memory side while the kernel is still running (same kernel time in the probe), so nothing is left for the boundary.
// -DSY_WT_STORES=0 builds the plain-store library for A/B.
#ifndef SY_WT_STORES
#define SY_WT_STORES 1
#endif
// per-site switches (A/B builds): convolution epilogue, BatchNorm forward apply, BatchNorm backward apply, weight-gradient slabs,
// the fold's read-modify-write of dW, the arena clears
#ifndef SY_WT_CONV
#define SY_WT_CONV 1
#endif
#ifndef SY_WT_BNF
#define SY_WT_BNF 1
#endif
#ifndef SY_WT_BNB
#define SY_WT_BNB 1
#endif
#ifndef SY_WT_SLAB
#define SY_WT_SLAB 0
#endif
#ifndef SY_WT_FOLD
#define SY_WT_FOLD 0
#endif
#ifndef SY_WT_ZERO
#define SY_WT_ZERO 0
#endif
#if defined(SY_EMU) || !SY_WT_STORES
static inline __host__ __device__ void sy_store16_wt(void* dst, const uint4& v) { *reinterpret_cast<uint4*>(dst) = v; }
static inline __host__ __device__ void sy_store8_wt(void* dst, const uint2& v) { *reinterpret_cast<uint2*>(dst) = v; }
static inline __host__ __device__ void sy_store4_wt(void* dst, unsigned v) { *reinterpret_cast<unsigned*>(dst) = v; }
#else
__device__ __forceinline__ void sy_store8_wt(void* dst, const uint2& v) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v2u d;
    __builtin_memcpy(&d, &v, 8);
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(dst), "v"(d) : "memory");
}
__device__ __forceinline__ void sy_store4_wt(void* dst, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void sy_store16_wt(void* dst, const uint4& v) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u d;
    __builtin_memcpy(&d, &v, 16);
    // s_nop: a store of more than 8 bytes reads its data registers over several cycles, a VALU write to them in the next cycle is a
    // hazard (gfx90a / gfx940 "VMEM store data" hazard, one wait state) — the compiler inserts the wait state for its own stores,
    // not behind inline assembly (found the hard way: every fourth 64-byte piece of a BatchNorm output held integers)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(d) : "memory");
}
#endif

#ifdef SY_EMU
static inline unsigned long long sy_uniform64(unsigned long long v) { return v; }
#else
__device__ __forceinline__ unsigned long long sy_uniform64(unsigned long long v) {      // wave-uniform 64-bit value -> SGPR pair
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
#endif

// ---- XCD-aware workgroup order ------------------------------------------------------------------------------
// MI355X dispatches consecutive workgroup ids round-robin over its 8 XCDs, each with a private L2.  Tiles that
// share operand rows (the channel tiles of one pixel tile, neighbouring pixel tiles with their halo rows, the
// K-splits' row / column tiles of one pixel range) have CONSECUTIVE logical ids in our grids, so left alone
// they would land on 8 different L2s and every XCD would fetch the same rows.  The remap hands XCD k one
// contiguous range of logical ids instead: hardware id w (XCD w % 8, its (w / 8)-th workgroup) -> logical id
// start(k) + w / 8.  A bijection for any grid size (the first nwg % 8 XCDs take one extra workgroup).
struct sy_block_id { int x, y, z; };
__device__ __forceinline__ sy_block_id sy_xcd_block_id() {
    constexpr unsigned kXcd = 8;
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const unsigned nwg = gx * gy * gz;
    const unsigned w = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned k = w % kXcd, q = nwg / kXcd, r = nwg % kXcd;
    const unsigned L = k * q + (k < r ? k : r) + w / kXcd;
    sy_block_id b;
    b.x = (int)(L % gx);
    b.y = (int)((L / gx) % gy);
    b.z = (int)(L / (gx * gy));
    return b;
}

// ---- in-kernel timeline probe (development builds only: `make probe`, tools/kernel_timeline.py) --------------------------------
// sy_probe(slot): thread 0 of the workgroup stamps the device's constant 100 MHz clock (s_memrealtime) into slot `slot` of its
// workgroup's row; one buffer per translation unit, read back through that unit's SY_PROBE_READER entry point.  Compiled out of
// the product library.
#if defined(SY_PROBE) && !defined(SY_EMU)
constexpr int kProbeWG = 8192, kProbeSlots = 8;
static __device__ unsigned long long sy_probe_buf[kProbeWG * kProbeSlots];
__device__ __forceinline__ void sy_probe(int slot) {
    if (threadIdx.x == 0) {
        const unsigned w = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (w < (unsigned)kProbeWG) sy_probe_buf[w * kProbeSlots + slot] = __builtin_amdgcn_s_memrealtime();
    }
}
// Launch timeline (tools/step_timeline.py): every workgroup's thread 0 folds its entry / exit stamps into ONE record per launch,
// found by hashing the launch's kernel-argument segment address (unique among the ~1600 launches of a step: the runtime hands the
// segments out of a ring far larger than that) — min entry, max exit, workgroup count, a family tag.  Unlike rocprofv3's kernel
// trace (which serialises the streams on this stack) the records show the launches of a taped multi-stream step as they really
// overlap.
struct sy_tl_entry { unsigned long long key, t0, t1; unsigned tag, count; };
constexpr int kTlSlots = 65536;
static __device__ sy_tl_entry sy_tl[kTlSlots];
__device__ __forceinline__ int sy_tl_begin(unsigned tag) {
    // Workgroup 0 (dispatched first) opens the launch's record and stamps its start; exits are folded in by a SAMPLE of the
    // workgroups (every 16th and the last four in dispatch order): thousands of same-address atomics per launch would serialise in
    // L2 and stretch the step severalfold (the first version of this probe did: 101 ms instead of 23).
    // (the block arithmetic stays in uniform control flow: the kernels reuse these values as scalar operands)
    const unsigned w = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const bool first = w == 0;
    const bool sample = first || (w & 15u) == 15u || w + 4u >= nwg;
    if (!sample || threadIdx.x != 0) return -1;
    const unsigned long long key = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned h = (unsigned)((key >> 6) * 2654435761ull) & (unsigned)(kTlSlots - 1);      // direct-mapped: a collision loses the record
    if (first) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        const unsigned long long old = atomicCAS(&sy_tl[h].key, 0ull, key);
        if (old != 0ull && old != key) return -1;
        sy_tl[h].t0 = t;
        sy_tl[h].tag = tag;
        sy_tl[h].count = nwg;
        return (int)h;
    }
    return __hip_atomic_load(&sy_tl[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key ? (int)h : -1;
}
__device__ __forceinline__ void sy_tl_end(int slot) {
    if (slot >= 0) atomicMax(&sy_tl[slot].t1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
#define SY_TL_BEGIN(tag) const int sy_tl_slot_ = sy_tl_begin(tag)
#define SY_TL_END() sy_tl_end(sy_tl_slot_)
#define SY_PROBE_READER(name)                                                                                          \
    extern "C" __attribute__((visibility("default"))) int name(unsigned long long* dst, int clear) {                 \
        if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(sy_probe_buf), sizeof(sy_probe_buf)) != hipSuccess) return 2;          \
        if (clear) { static unsigned long long z[kProbeWG * kProbeSlots]; (void)hipMemcpyToSymbol(HIP_SYMBOL(sy_probe_buf), z, sizeof(z)); } \
        return 0;                                                                                                    \
    }                                                                                                                \
    extern "C" __attribute__((visibility("default"))) int name##_tl(void* dst, int clear) {                          \
        if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(sy_tl), sizeof(sy_tl)) != hipSuccess) return 2;                       \
        if (clear) {                                                                                                 \
            static sy_tl_entry z[kTlSlots];                                                                          \
            for (int i = 0; i < kTlSlots; ++i) { z[i].key = 0; z[i].t0 = ~0ull; z[i].t1 = 0; z[i].tag = 0; z[i].count = 0; } \
            (void)hipMemcpyToSymbol(HIP_SYMBOL(sy_tl), z, sizeof(z));                                                \
        }                                                                                                            \
        return 0;                                                                                                    \
    }
#else
#define sy_probe(slot) ((void)0)
#define SY_PROBE_READER(name)
#define SY_TL_BEGIN(tag) ((void)0)
#define SY_TL_END() ((void)0)
#endif

// ---- late kernel arguments -----------------------------------------------------------------------------------
// hipcc loads every field of a by-value argument struct into SGPRs at kernel entry and keeps the ones the epilogue
// needs alive across the main loop, where they crowd out (spill) the loop's own uniforms.  SY_LATE_ARGS re-reads the
// struct from the kernarg segment through a laundered pointer: fields touched only after the loop are then loaded
// after the loop (scalar loads, K$-resident).  The struct must be the kernel's first (only) parameter.
#ifdef SY_EMU
#define SY_LATE_ARGS(Type, p) const Type& p##_late = (p)
#define SY_LAUNDER_INT(x) ((void)0)
#else
#define SY_LATE_ARGS(Type, p)                                                                                       \
    const __attribute__((address_space(4))) Type* p##_late_ptr =                                                     \
        (const __attribute__((address_space(4))) Type*)__builtin_amdgcn_kernarg_segment_ptr();                      \
    asm volatile("" : "+s"(p##_late_ptr));                                                                          \
    const __attribute__((address_space(4))) Type& p##_late = *p##_late_ptr
#define SY_LAUNDER_INT(x) asm volatile("" : "+s"(x))
#endif

// ---- small math ---------------------------------------------------------------------------------
#ifdef SY_EMU
static inline float sy_exp(float x) { return expf(x); }
#else
__device__ __forceinline__ float sy_exp(float x) { return __expf(x); }      // v_exp_f32 based, ~1e-6 relative
#endif
// 1 / x to 1 ulp (v_rcp_f32) — a correctly rounded fp32 division is a 12-instruction sequence on this ISA, and every SiLU of
// the BatchNorm passes and conv epilogues has one
#ifdef SY_EMU
static inline float sy_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float sy_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
__device__ __forceinline__ float sy_sigmoid(float z) { return sy_rcp(1.0f + sy_exp(-z)); }
__device__ __forceinline__ float sy_silu(float z) { return z * sy_sigmoid(z); }
// d silu(z)/dz = s * (1 + z * (1 - s))
__device__ __forceinline__ float sy_silu_grad(float z) { float s = sy_sigmoid(z); return s * (1.0f + z * (1.0f - s)); }

// compile-time for: f(std::integral_constant<int, I>) for I in [B, E)
template <int I> struct sy_int { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void sy_static_for(F&& f) {
    if constexpr (B < E) {
        f(sy_int<B>());
        sy_static_for<B + 1, E>(f);
    }
}

template <typename T> __device__ __forceinline__ T sy_min(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ __forceinline__ T sy_max(T a, T b) { return a > b ? a : b; }

// status codes returned through the C ABI
enum : int { SY_OK = 0, SY_ERR_ARG = 1, SY_ERR_LAUNCH = 2, SY_ERR_UNSUPPORTED = 3 };
