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

// One workgroup = one 32 (co) x 32 (ci) tile of one entry (found by bisection over the entries' tile0 prefix).
// Phase A: threads (co, ci) ci-fastest read their taps straight from the OIHW parameter (a wave reads 32 x taps
// consecutive floats) and write packed / frag, whose fast index is ci; the fp32 tile is parked in LDS.  Phase B:
// threads (ci, co) co-fastest write packed_t / frag_t, whose fast index is co.  Every global access is coalesced.
constexpr int kPackTile = 32, kPackTaps = 9;

template <typename T> __device__ void pack_tile(const sy_pack_entry& e, int tile, float (*lds)[kPackTile][kPackTile + 1]) {
    const int tiles_ci = (e.ci_n + kPackTile - 1) / kPackTile;
    const int co0 = (tile / tiles_ci) * kPackTile, ci0 = (tile % tiles_ci) * kPackTile;
    const int taps = e.taps;
    const int a_l = threadIdx.x >> 5, b_l = threadIdx.x & 31;
    for (int pass = 0; pass < kPackTile / 8; ++pass) {
        const int col = a_l + 8 * pass, co = co0 + col, ci = ci0 + b_l;
        const bool ok = co < e.co_n && ci < e.ci_n;
        const float* src = e.w + ((long long)co * e.ci_n + ci) * taps;
        const int row = e.r0 + co;
        for (int t = 0; t < taps; ++t) {
            const float v = ok ? src[t] : 0.0f;
            lds[t][col][b_l] = v;
            if (!ok) continue;
            if (e.packed != nullptr) put<T>(e.packed, ((long long)row * taps + t) * e.CI + ci, v);
            if (e.frag != nullptr) put<T>(e.frag, frag_index<T>(row, t, ci, taps, e.CI), v);
        }
    }
    __syncthreads();
    if (e.packed_t == nullptr && e.frag_t == nullptr) return;
    for (int pass = 0; pass < kPackTile / 8; ++pass) {
        const int cil = a_l + 8 * pass, ci = ci0 + cil, co = co0 + b_l;
        if (co >= e.co_n || ci >= e.ci_n) continue;
        const int row = e.r0 + co;
        for (int t = 0; t < taps; ++t) {
            const float v = lds[t][b_l][cil];
            if (e.packed_t != nullptr) put<T>(e.packed_t, ((long long)ci * taps + t) * e.R_t + row, v);
            if (e.frag_t != nullptr) put<T>(e.frag_t, frag_index<T>(ci, t, row, taps, e.R_t), v);
        }
    }
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const sy_pack_entry* entries, int n_entries) {
    __shared__ float lds[kPackTaps][kPackTile][kPackTile + 1];
    int lo = 0, hi = n_entries - 1;                           // last entry with tile0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (entries[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const sy_pack_entry e = entries[lo];
    const int tile = (int)blockIdx.x - e.tile0;
    const int tiles = ((e.co_n + kPackTile - 1) / kPackTile) * ((e.ci_n + kPackTile - 1) / kPackTile);
    if (tile >= tiles || e.taps > kPackTaps) return;
    switch (e.dtype) {
        case SY_DT_BF16: pack_tile<BF16>(e, tile, lds); break;
        case SY_DT_F16: pack_tile<F16>(e, tile, lds); break;
        default: pack_tile<F32>(e, tile, lds); break;
    }
}

}  // namespace

extern "C" int sy_pack_weights(const sy_pack_entry* entries, int n_entries, int total_tiles, void* stream) {
    if (entries == nullptr || n_entries <= 0 || total_tiles <= 0) return SY_ERR_ARG;
    SY_LAUNCH(pack_weights_kernel, dim3(total_tiles), dim3(256), 0, stream, entries, n_entries);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
