// tr16_probe.hip — prints the lane/element mapping of ds_read_b64_tr_b16 (gfx950) for a given address pattern.
// Build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o tr16_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(v4s* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;       // element value = its own index
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, G = l >> 4;
    int elem;                                                             // element index of this lane's 8-byte piece
    if (mode == 0) elem = l * 4;                                          // contiguous: lane l -> elements [4l, 4l+4)
    else elem = G * 1024 + (i >> 2) * 16 + (i & 3) * 4;                   // [4 rows][16 cols] block per group, row pitch 16 el
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + elem));
    out[l] = v;
}
int main() {
    v4s* d; hipMalloc(&d, 64 * sizeof(v4s));
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode);
        v4s h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
    }
    return 0;
}
