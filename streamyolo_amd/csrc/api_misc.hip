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
    SY_TL_BEGIN(15);
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
    SY_TL_END();
}

}  // namespace

extern "C" int sy_pack_weights(const sy_pack_entry* entries, int n_entries, int total_tiles, void* stream) {
    if (entries == nullptr || n_entries <= 0 || total_tiles <= 0) return SY_ERR_ARG;
    SY_LAUNCH(pack_weights_kernel, dim3(total_tiles), dim3(256), 0, stream, entries, n_entries);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

// ---- small fp32 helpers of the backward plan (keep the step free of host-side tensor ops) --------------------------------
namespace {

// dst[r][0..cols) += src[r][0..cols) for r < rows (row pitches ldd / lds); zero_src: src rows are zeroed afterwards, so a
// scratch that the NEXT step's accumulating weight-gradient launch reuses is clean again without a separate memset
__global__ __launch_bounds__(256) void rows_add_kernel(float* dst, long long ldd, float* src, long long lds, int rows, int cols,
                                                      int zero_src) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cols, c = i - r * cols;
        dst[r * ldd + c] += src[r * lds + c];
        if (zero_src) src[r * lds + c] = 0.0f;
    }
}

// column sums of one level's rows of d_raw [B][rows][nch] (batch stride bs): block (g, cg) sums rows g, g + G, ... for the 16
// columns of group cg -> partial[g][cg * 16 + c]  (fixed order: the result does not depend on scheduling)
__global__ __launch_bounds__(256) void pred_colsum_kernel(const float* d_raw, int B, long long bs, int rows, int nch, float* partial,
                                                         int nch_pad) {
    SY_TL_BEGIN(15);
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, rl = threadIdx.x >> 4, col = blockIdx.y * 16 + c;
    const long long total = (long long)B * rows;
    float acc = 0.0f;
    if (col < nch)
        for (long long i = (long long)blockIdx.x * 16 + rl; i < total; i += (long long)gridDim.x * 16) {
            const long long b = i / rows, p = i - b * rows;
            acc += d_raw[b * bs + p * nch + col];
        }
    red[rl][c] = acc;
    __syncthreads();
    if (rl == 0) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][c];
        partial[(long long)blockIdx.x * nch_pad + col] = s;
    }
    SY_TL_END();
}

// one workgroup: bias gradients += column sums (partials folded in index order), weight gradients += the wgrad scratch
// (scratch [2][srows][ld]: [0][0..3] reg, [0][4] obj, [1][0..nc) cls), scratch zeroed behind the read
__global__ __launch_bounds__(256) void pred_fold_kernel(const float* partial, int G, int nch_pad, int nc, float* scratch, int srows, int ld, int cin,
                                                       float* g_reg, float* g_obj, float* g_cls, float* gb_reg, float* gb_obj,
                                                       float* gb_cls) {
    SY_TL_BEGIN(15);
    const int t = threadIdx.x;
    for (int col = t; col < 5 + nc; col += 256) {
        float s = 0.0f;
        for (int g = 0; g < G; ++g) s += partial[(long long)g * nch_pad + col];
        if (col < 4) gb_reg[col] += s;
        else if (col == 4) gb_obj[0] += s;
        else gb_cls[col - 5] += s;
    }
    float* s0 = scratch;                           // [srows][ld]: rows 0-3 reg, 4 obj
    float* s1 = scratch + srows * (long long)ld;   // [srows >= nc][ld]: cls
    for (int i = t; i < 5 * cin; i += 256) {
        const int r = i / cin, k = i - r * cin;
        const float v = s0[r * ld + k];
        s0[r * ld + k] = 0.0f;
        if (r < 4) g_reg[r * cin + k] += v; else g_obj[k] += v;
    }
    for (int i = t; i < nc * cin; i += 256) {
        const int r = i / cin, k = i - r * cin;
        g_cls[i] += s1[r * ld + k];
        s1[r * ld + k] = 0.0f;
    }
    SY_TL_END();
}

}  // namespace

extern "C" int sy_rows_add_f32(float* dst, int64_t ldd, float* src, int64_t lds, int rows, int cols, int zero_src, void* stream) {
    if (dst == nullptr || src == nullptr || rows <= 0 || cols <= 0 || ldd < cols || lds < cols) return SY_ERR_ARG;
    long long blocks = ((long long)rows * cols + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    SY_LAUNCH(rows_add_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, (long long)ldd, src, (long long)lds, rows, cols, zero_src);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

extern "C" int64_t sy_pred_grad_fold_workspace_floats(int num_classes) {
    return 64LL * (((5 + num_classes) + 15) / 16 * 16);
}

extern "C" int sy_pred_grad_fold(const float* d_raw, int B, int64_t batch_stride, int rows, int num_classes, float* scratch,
                                 int scratch_rows, int ld, int cin, float* g_reg, float* g_obj, float* g_cls, float* gb_reg, float* gb_obj, float* gb_cls,
                                 float* workspace, void* stream) {
    if (d_raw == nullptr || scratch == nullptr || workspace == nullptr || g_reg == nullptr || g_obj == nullptr || g_cls == nullptr ||
        gb_reg == nullptr || gb_obj == nullptr || gb_cls == nullptr || B <= 0 || rows <= 0 || num_classes <= 0 || cin <= 0 || ld < cin || scratch_rows < 5 || scratch_rows < num_classes)
        return SY_ERR_ARG;
    const int nch = 5 + num_classes, groups = (nch + 15) / 16, nch_pad = groups * 16, G = 64;
    SY_LAUNCH(pred_colsum_kernel, dim3(G, groups), dim3(256), 0, stream, d_raw, B, (long long)batch_stride, rows, nch, workspace, nch_pad);
    SY_LAUNCH(pred_fold_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, G, nch_pad, num_classes, scratch, scratch_rows, ld, cin, g_reg,
              g_obj, g_cls, gb_reg, gb_obj, gb_cls);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
SY_PROBE_READER(sy_probe_read_api_misc)
