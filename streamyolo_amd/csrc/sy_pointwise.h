// sy_pointwise.h — shared scaffolding of the HBM-bound kernels: 16-byte channel chunks, grid sizing,
// dtype dispatch.
#pragma once
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(long long work) {
    long long b = (work + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;        // 256 CUs x 8 workgroups, grid-stride beyond
    return (int)b;
}

template <typename T> struct Chunk {
    typedef typename T::elem elem;
    static constexpr int N = T::kEPC;
    elem e[N];
    __device__ __forceinline__ static Chunk load(const void* p) {
        Chunk c;
        uint4 v = *reinterpret_cast<const uint4*>(p);
        __builtin_memcpy(c.e, &v, 16);
        return c;
    }
    __device__ __forceinline__ void store(void* p) const {
        uint4 v;
        __builtin_memcpy(&v, e, 16);
        *reinterpret_cast<uint4*>(p) = v;
    }
    __device__ __forceinline__ void store_wt(void* p) const {      // GLOBAL destinations only: write-through (sy_store16_wt)
        uint4 v;
        __builtin_memcpy(&v, e, 16);
        sy_store16_wt(p, v);
    }
};

inline int epc_of(int dtype) { return dtype == SY_DT_F32 ? 4 : 8; }

}  // namespace

#define SY_DISPATCH_DTYPE(dtype, CALL)                    \
    switch (dtype) {                                      \
        case SY_DT_BF16: { typedef BF16 T; CALL; break; } \
        case SY_DT_F16: { typedef F16 T; CALL; break; }   \
        case SY_DT_F32: { typedef F32 T; CALL; break; }   \
        default: return SY_ERR_ARG;                       \
    }                                                     \
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH
