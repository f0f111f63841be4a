// What does a DEPENDENT launch cost on an in-order HIP stream of the MI355X (development aid, round 6)?
//   A  N kernels back to back on one stream (barrier bit: kernel i + 1 starts after kernel i has ended): gap = first workgroup entry of
//      i + 1 minus last workgroup exit of i, by the 100 MHz device clock;
//   B  the same kernels issued with hipExtAnyOrderLaunch (no barrier bit) and the dependency carried in the kernels: every workgroup of
//      launch i bumps done[i] when it leaves (agent-scope release), every workgroup of launch i + 1 waits at its entry until done[i] has
//      reached the grid size (acquire; bounded by a 2 ms timeout, so the probe cannot hang): gap = first workgroup PAST its wait minus
//      last exit of i — the consumer's workgroups are resident (and would have their prologue behind them) when the producer ends.
// hipcc --offload-arch=gfx950 -O3 tools/probes/chain_gap_probe.hip -o /tmp/chain_gap_probe && /tmp/chain_gap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

//   C  as A, but every workgroup also WRITES (and, in the next launch, reads) `wr_bytes` of a buffer: what the end-of-kernel write-back of
//      dirty L2 lines (eight non-coherent L2s) and the next kernel's cold reads add to the boundary.
__global__ __launch_bounds__(256) void link_kernel(unsigned long long* stamps, unsigned* done, int i, int grid, int spin_ticks, int wait_on_prev,
                                                   unsigned* timeouts, uint4* data, int wr_bytes, int nt) {
    const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
    if (wait_on_prev && i > 0) {
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(&done[i - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)grid) {
                if (__builtin_amdgcn_s_memrealtime() - t_in > 200000ull) { atomicAdd(timeouts, 1u); break; }   // 2 ms
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    const unsigned long long t_go = __builtin_amdgcn_s_memrealtime();
    if (wr_bytes > 0) {
        typedef unsigned v4 __attribute__((ext_vector_type(4)));
        v4* mine = reinterpret_cast<v4*>(data) + (size_t)blockIdx.x * (wr_bytes / 16);
        v4 acc = {0u, 0u, 0u, 0u};
        for (int k = threadIdx.x; k < wr_bytes / 16; k += 256) {
            const v4 v = mine[k];                                        // what the previous launch wrote
            acc += v;
        }
        for (int k = threadIdx.x; k < wr_bytes / 16; k += 256) {
            v4 o = acc; o.x += (unsigned)k;
            if (nt == 1) __builtin_nontemporal_store(o, &mine[k]);
            else if (nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(&mine[k]), "v"(o) : "memory");   // write-through, system scope
            else if (nt == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(&mine[k]), "v"(o) : "memory");
            else if (nt == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(&mine[k]), "v"(o) : "memory");
            else mine[k] = o;
        }
    }
    while (__builtin_amdgcn_s_memrealtime() - t_go < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* s = stamps + ((long long)i * grid + blockIdx.x) * 3;
        s[0] = t_in; s[1] = t_go; s[2] = __builtin_amdgcn_s_memrealtime();
        if (wait_on_prev) __hip_atomic_fetch_add(&done[i], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // (an agent-scope release per
                                                                                      // workgroup: ~18 ns each, serialised — see README)
    }
}

//   D  is a write-through store a legal replacement?  Writer launch i stores the value i into every word of a buffer (store variant
//      `mode`, optionally s_waitcnt vmcnt(0) before the wave ends), reader launch i (next on the stream, in order) has every workgroup
//      check ANOTHER workgroup's region (another XCD: blockIdx + 3) and counts the words that are not i.  The reader's L2 holds the
//      lines of launch i - 1.
typedef unsigned v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr_kernel(v4* data, int words16_per_wg, unsigned val, int mode, int drain) {
    v4* mine = data + (size_t)blockIdx.x * words16_per_wg;
    const v4 o = {val, val, val, val};
    for (int k = threadIdx.x; k < words16_per_wg; k += 256) {
        if (mode == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(&mine[k]), "v"(o) : "memory");
        else if (mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(&mine[k]), "v"(o) : "memory");
        else mine[k] = o;
    }
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__global__ __launch_bounds__(256) void rd_kernel(const v4* data, int words16_per_wg, unsigned val, unsigned* bad) {
    const v4* theirs = data + (size_t)((blockIdx.x + 3) % gridDim.x) * words16_per_wg;
    unsigned n = 0;
    for (int k = threadIdx.x; k < words16_per_wg; k += 256) {
        const v4 v = theirs[k];
        n += (v.x != val) + (v.y != val) + (v.z != val) + (v.w != val);
    }
    if (n) atomicAdd(bad, n);
}
static void coherence(const char* name, int grid, int bytes_per_wg, int mode, int drain, uint4* d_data, unsigned* d_bad) {
    hipStream_t st;
    hipStreamCreate(&st);
    hipMemsetAsync(d_bad, 0, sizeof(unsigned), st);
    const int N = 300;
    for (int i = 1; i <= N; ++i) {
        hipLaunchKernelGGL(wr_kernel, dim3(grid), dim3(256), 0, st, (v4*)d_data, bytes_per_wg / 16, (unsigned)i, mode, drain);
        hipLaunchKernelGGL(rd_kernel, dim3(grid), dim3(256), 0, st, (const v4*)d_data, bytes_per_wg / 16, (unsigned)i, d_bad);
    }
    hipStreamSynchronize(st);
    unsigned bad = 0;
    hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    printf("D  %-52s grid %5d x %6d B: %u stale words in %d write -> read rounds\n", name, grid, bytes_per_wg, bad, N);
    hipStreamDestroy(st);
}

static void run(const char* name, int N, int grid, int spin_ticks, bool any_order, unsigned long long* d_st, unsigned* d_done, unsigned* d_to,
                uint4* d_data = nullptr, int wr_bytes = 0, int nt = 0) {
    hipStream_t st;
    hipStreamCreate(&st);
    std::vector<unsigned long long> h((size_t)N * grid * 3);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemsetAsync(d_done, 0, sizeof(unsigned) * N, st);
        hipMemsetAsync(d_to, 0, sizeof(unsigned), st);
        for (int i = 0; i < N; ++i) {
            if (any_order)
                hipExtLaunchKernelGGL(link_kernel, dim3(grid), dim3(256), 0, st, nullptr, nullptr, (i > 0 ? hipExtAnyOrderLaunch : 0), d_st, d_done, i, grid,
                                      spin_ticks, 1, d_to, (uint4*)nullptr, 0, 0);
            else
                hipLaunchKernelGGL(link_kernel, dim3(grid), dim3(256), 0, st, d_st, d_done, i, grid, spin_ticks, 0, d_to, d_data, wr_bytes, nt);
        }
        hipStreamSynchronize(st);
    }
    unsigned to = 0;
    hipMemcpy(&to, d_to, sizeof(to), hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> gap, early;
    for (int i = 1; i < N; ++i) {
        unsigned long long last_exit = 0, first_go = ~0ull, first_in = ~0ull;
        for (int b = 0; b < grid; ++b) {
            last_exit = std::max(last_exit, h[((size_t)(i - 1) * grid + b) * 3 + 2]);
            first_go = std::min(first_go, h[((size_t)i * grid + b) * 3 + 1]);
            first_in = std::min(first_in, h[((size_t)i * grid + b) * 3 + 0]);
        }
        gap.push_back(((double)first_go - (double)last_exit) / 100.0);
        early.push_back(((double)last_exit - (double)first_in) / 100.0);
    }
    std::sort(gap.begin(), gap.end());
    std::sort(early.begin(), early.end());
    const double total = ((double)h[((size_t)(N - 1) * grid) * 3 + 2] - (double)h[0]) / 100.0;
    if (wr_bytes > 0) printf("   [%5.1f MB written + read per launch%s] ", (double)grid * wr_bytes / 1e6, nt == 1 ? ", nontemporal stores" : nt == 2 ? ", stores sc0 sc1" : nt == 3 ? ", stores sc1" : nt == 4 ? ", stores sc0 sc1 nt" : "");
    printf("%-44s grid %5d  body %5.1f us: gap p10 %6.2f  p50 %6.2f  p90 %6.2f us | entry before producer's end p50 %6.2f us | %d launches in %8.1f us = %6.2f us each | timeouts %u\n",
           name, grid, spin_ticks / 100.0, gap[gap.size() / 10], gap[gap.size() / 2], gap[gap.size() * 9 / 10], early[early.size() / 2], N, total, total / N, to);
    hipStreamDestroy(st);
}

int main() {
    const int N = 200, GMAX = 2048;
    unsigned long long* d_st;
    unsigned *d_done, *d_to;
    hipMalloc(&d_st, sizeof(unsigned long long) * (size_t)N * GMAX * 3);
    hipMalloc(&d_done, sizeof(unsigned) * N);
    hipMalloc(&d_to, sizeof(unsigned));
    for (int grid : {64, 256, 512, 1024, 2048})
        for (int spin : {500, 2000}) {
            run("A  in-order stream (barrier bit)", N, grid, spin, false, d_st, d_done, d_to);
            run("B  any-order launch + in-kernel wait", N, grid, spin, true, d_st, d_done, d_to);
        }
    uint4* d_data;
    hipMalloc(&d_data, (size_t)64 << 20);
    hipMemset(d_data, 0, (size_t)64 << 20);
    for (int grid : {256, 1024})
        for (int wr : {4096, 16384, 65536})
            for (int nt : {0, 1, 2, 3, 4}) {
                if ((size_t)grid * wr > ((size_t)64 << 20)) continue;
                run("C  in-order stream, writes + reads", N, grid, 500, false, d_st, d_done, d_to, d_data, wr, nt);
            }
    for (int grid : {8, 64, 1024})
        for (int bpw : {256, 4096, 65536}) {
            if ((size_t)grid * bpw > ((size_t)64 << 20)) continue;
            coherence("plain stores", grid, bpw, 0, 0, d_data, d_to);
            coherence("sc1 stores", grid, bpw, 1, 0, d_data, d_to);
            coherence("sc1 stores + s_waitcnt vmcnt(0) at the wave's end", grid, bpw, 1, 1, d_data, d_to);
            coherence("sc0 sc1 stores", grid, bpw, 2, 0, d_data, d_to);
            coherence("sc0 sc1 stores + s_waitcnt vmcnt(0)", grid, bpw, 2, 1, d_data, d_to);
        }
    return 0;
}
