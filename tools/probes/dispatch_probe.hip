// Workgroup dispatch ramp of the MI355X (development aid, round 6): how long after the first workgroup of a launch does the last one
// START, as a function of workgroup size, static LDS and register footprint?  Every workgroup's thread 0 stamps the 100 MHz device clock
// at entry and exit.  hipcc --offload-arch=gfx950 -O3 tools/probes/dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int T, int LDS_BYTES, int REGS>
__global__ __launch_bounds__(T) void probe_kernel(unsigned long long* stamps, const float* in, float* out, int work) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __shared__ float lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    float acc[REGS];
#pragma unroll
    for (int r = 0; r < REGS; ++r) acc[r] = in[(threadIdx.x + r * T) & 1023];
    for (int w = 0; w < work; ++w)
#pragma unroll
        for (int r = 0; r < REGS; ++r) acc[r] = acc[r] * 1.0001f + 0.5f;
    if (LDS_BYTES > 0) { lds[threadIdx.x % (LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1)] = acc[0]; __syncthreads(); acc[0] += lds[0]; }
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < REGS; ++r) s += acc[r];
    if (s == 123456.0f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int T, int LDS_BYTES, int REGS>
void run(const char* name, int total_threads, int work, unsigned long long* d_st, const float* d_in, float* d_out) {
    const int G = total_threads / T;
    std::vector<unsigned long long> h(2 * G);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((probe_kernel<T, LDS_BYTES, REGS>), dim3(G), dim3(T), 0, 0, d_st, d_in, d_out, work);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost);
    std::vector<double> st(G), en(G);
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < G; ++i) t0 = std::min(t0, h[2 * i]);
    for (int i = 0; i < G; ++i) { st[i] = (h[2 * i] - t0) / 100.0; en[i] = (h[2 * i + 1] - t0) / 100.0; }
    std::sort(st.begin(), st.end());
    std::sort(en.begin(), en.end());
    printf("%-34s G %5d x T %4d  lds %5d  regs %3d : start p50 %5.2f p90 %5.2f max %5.2f us | end max %5.2f us | life p50 %5.2f\n", name, G, T, LDS_BYTES, REGS,
           st[G / 2], st[G * 9 / 10], st[G - 1], en[G - 1], en[G / 2] - st[G / 2]);
}

int main() {
    unsigned long long* d_st; float *d_in, *d_out;
    hipMalloc(&d_st, sizeof(unsigned long long) * 2 * 16384);
    hipMalloc(&d_in, 4096); hipMalloc(&d_out, 4096);
    hipMemset(d_in, 0, 4096);
    const int N = 768 * 256;                               // the thread count of a BatchNorm-backward reduce launch
    run<64, 0, 8>("64 threads", N, 50, d_st, d_in, d_out);
    run<256, 0, 8>("256 threads", N, 50, d_st, d_in, d_out);
    run<512, 0, 8>("512 threads", N, 50, d_st, d_in, d_out);
    run<1024, 0, 8>("1024 threads", N, 50, d_st, d_in, d_out);
    run<256, 16384, 8>("256 threads, 16 KB LDS", N, 50, d_st, d_in, d_out);
    run<256, 0, 64>("256 threads, 64+ regs", N, 50, d_st, d_in, d_out);
    run<256, 16384, 64>("256 thr, 16 KB LDS, 64+ regs", N, 50, d_st, d_in, d_out);
    run<1024, 16384, 64>("1024 thr, 16 KB LDS, 64+ regs", N, 50, d_st, d_in, d_out);
    run<256, 0, 8>("256 threads, short", N, 1, d_st, d_in, d_out);
    run<256, 0, 8>("256 threads, 2x grid", 2 * N, 50, d_st, d_in, d_out);
    run<256, 0, 8>("256 threads, 256 WGs", 256 * 256, 50, d_st, d_in, d_out);
    return 0;
}
