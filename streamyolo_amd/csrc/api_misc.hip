// api_misc.hip — version / ABI probes of libstreamyolo_hip.so.
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

extern "C" const char* sy_version(void) {
#ifdef SY_EMU
    return "streamyolo-hip 0.1 (SIMT-EMULATOR TEST BUILD - not a product binary)";
#else
    return "streamyolo-hip 0.1 (gfx950)";
#endif
}
extern "C" int sy_abi_version(void) { return SY_ABI_VERSION; }

// ---- sy_pack_weights ------------------------------------------------------------------------------------
namespace {

template <typename T> __device__ __forceinline__ void put(void* base, long long i, float v) {
    reinterpret_cast<typename T::elem*>(base)[i] = T::from_f32(v);
}
// fragment order of a [rows][taps][kc] matrix (see pack_conv_weight_frag / conv_igemm STG 5)
template <typename T> __device__ __forceinline__ long long frag_index(int row, int tap, int c, int taps, int kc) {
    constexpr int EPC = T::kEPC, BK = 4 * EPC;
    const int cslab = c / BK, within = c % BK;
    const int g = within / (2 * EPC), half = (within / EPC) & 1, e = within % EPC;
    const int ct = row >> 5, r = row & 31, nslab = kc / BK;
    return ((((((long long)ct * nslab + cslab) * taps + tap) * 2 + g) * 2 + half) * 32 + r) * EPC + e;
}

// grid (x, entry, pass).  pass 0: threads over (co, ci) ci-fastest -> packed / frag (coalesced writes);
// pass 1: threads over (ci, co) co-fastest -> packed_t / frag_t.  A thread owns all taps of its (co, ci).
template <typename T> __device__ void pack_entry(const sy_pack_entry& e, int pass) {
    const long long n = (long long)e.co_n * e.ci_n;
    const int taps = e.taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int co, ci;
        if (pass == 0) { co = (int)(i / e.ci_n); ci = (int)(i - (long long)co * e.ci_n); }
        else { ci = (int)(i / e.co_n); co = (int)(i - (long long)ci * e.co_n); }
        const float* src = e.w + ((long long)co * e.ci_n + ci) * taps;
        const int row = e.r0 + co;
        for (int t = 0; t < taps; ++t) {
            const float v = src[t];
            if (pass == 0) {
                if (e.packed != nullptr) put<T>(e.packed, ((long long)row * taps + t) * e.CI + ci, v);
                if (e.frag != nullptr) put<T>(e.frag, frag_index<T>(row, t, ci, taps, e.CI), v);
            } else {
                if (e.packed_t != nullptr) put<T>(e.packed_t, ((long long)ci * taps + t) * e.R_t + row, v);
                if (e.frag_t != nullptr) put<T>(e.frag_t, frag_index<T>(ci, t, row, taps, e.R_t), v);
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const sy_pack_entry* entries) {
    const sy_pack_entry e = entries[blockIdx.y];
    const int pass = blockIdx.z;
    if (pass == 0 ? (e.packed == nullptr && e.frag == nullptr) : (e.packed_t == nullptr && e.frag_t == nullptr)) return;
    if ((long long)blockIdx.x * blockDim.x >= (long long)e.co_n * e.ci_n) return;
    switch (e.dtype) {
        case SY_DT_BF16: pack_entry<BF16>(e, pass); break;
        case SY_DT_F16: pack_entry<F16>(e, pass); break;
        default: pack_entry<F32>(e, pass); break;
    }
}

}  // namespace

extern "C" int sy_pack_weights(const sy_pack_entry* entries, int n_entries, void* stream) {
    if (entries == nullptr || n_entries <= 0) return SY_ERR_ARG;
    SY_LAUNCH(pack_weights_kernel, dim3(64, n_entries, 2), dim3(256), 0, stream, entries);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
