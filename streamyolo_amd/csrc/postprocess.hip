// postprocess.hip — box decode to corners, confidence filter, class-aware greedy NMS, on device,
// for a whole batch, with no host synchronisation.
//
// Replaces yolox.utils.postprocess (exps/evaluators/onex_stream_evaluator.py:148-150 of the
// reference) and the inline `inference()` of sAP/streamyolo/streamyolo_det.py:62-83, i.e. a per-image
// Python loop + torchvision.ops.batched_nms + D2H copy.  Index selection must be BIT-EXACT with the
// reference semantics, so every float operation below is written in the order
// the reference evaluates it and this library is built with -ffp-contract=off:
//     corners  x1 = cx - w/2 ...;  class_conf = max_c cls (first max wins);  score = obj*class_conf
//     keep     score >= conf_thre
//     order    score descending, ties by ascending anchor index (stable argsort)
//     offset   box + class_id * (max_coord_over_kept_boxes + 1)        (torchvision batched_nms)
//     NMS      area = (x2-x1)*(y2-y1); inter = max(0,·)*max(0,·); iou = inter/(a_i+a_j-inter);
//              a later box is suppressed iff iou > nms_thre with an earlier KEPT box
//
// Three kernels per batch:
//   1. rank   — one 1024-thread workgroup per image: score, filter, max-coord reduction, LDS bitonic
//               sort of 64-bit keys (~score_bits << 32 | anchor), emit sorted offset boxes.
//   2. mask   — 64x64 tiles of the upper-triangular suppression bit matrix, one wave per tile,
//               grid-strided (the candidate count only exists on the device).
//   3. sweep  — one 256-thread workgroup per image walks the sorted list 64 boxes at a time: wave 0 resolves the
//               chunk with readlane steps on the diagonal word, then every thread ORs the chunk's kept rows into
//               its word column of the running `removed` bitmap (64 unconditional, coalesced loads in flight);
//               the detections are written afterwards by all threads from the per-chunk survivor words.
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

namespace {

constexpr int kRankThreads = 1024;
constexpr int kRec = 8;                       // floats per sorted record: x1o,y1o,x2o,y2o,area,anchor,pad,pad

struct PostLayout {                           // per-image slices of the workspace
    long long acap;                           // anchors rounded up to 64
    long long words;                          // acap / 64
    long long rec_off, mask_off, count_off, image_bytes;
};

inline PostLayout make_layout(int A) {
    PostLayout L;
    L.acap = ((long long)A + 63) / 64 * 64;
    L.words = L.acap / 64;
    L.rec_off = 0;
    L.mask_off = L.rec_off + L.acap * kRec * 4;
    L.count_off = L.mask_off + L.acap * L.words * 8;
    L.image_bytes = (L.count_off + 64 + 255) / 256 * 256;
    return L;
}

inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__global__ __launch_bounds__(kRankThreads) void nms_rank_kernel(const float* pred, int A, int nc, float conf_thre,
                                                                unsigned char* ws, PostLayout L, int sort_n) {
    SY_DYN_SMEM(smem);
    // all LDS lives in the dynamic region so its base stays 16-byte aligned (guide §6 G17)
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);        // [sort_n]
    float* s_max = reinterpret_cast<float*>(smem + (size_t)sort_n * 8);           // [kRankThreads/64]
    int& s_count = *reinterpret_cast<int*>(smem + (size_t)sort_n * 8 + (kRankThreads / 64) * 4);
    const int img = blockIdx.x;
    const int tid = threadIdx.x;
    const float* P = pred + (long long)img * A * (5 + nc);
    unsigned char* wsi = ws + (long long)img * L.image_bytes;
    float* rec = reinterpret_cast<float*>(wsi + L.rec_off);
    int* count_out = reinterpret_cast<int*>(wsi + L.count_off);

    if (tid == 0) s_count = 0;
    for (int i = tid; i < sort_n; i += kRankThreads) keys[i] = ~0ull;
    __syncthreads();

    float lmax = -INFINITY;
    for (int a = tid; a < A; a += kRankThreads) {
        const float* r = P + (long long)a * (5 + nc);
        float best = r[5];
        for (int c = 1; c < nc; ++c) { const float v = r[5 + c]; if (v > best) best = v; }
        const float score = r[4] * best;
        if (score >= conf_thre) {
            const int slot = atomicAdd(&s_count, 1);
            const unsigned bits = __builtin_bit_cast(unsigned, score);
            keys[slot] = ((unsigned long long)(0xffffffffu - bits) << 32) | (unsigned)a;
            const float hw = r[2] / 2, hh = r[3] / 2;
            const float x1 = r[0] - hw, y1 = r[1] - hh, x2 = r[0] + hw, y2 = r[1] + hh;
            lmax = fmaxf(fmaxf(fmaxf(lmax, x1), fmaxf(y1, x2)), y2);
        }
    }
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) s_max[tid >> 6] = lmax;
    __syncthreads();
    const int count = s_count;
    float max_coord = s_max[0];
    for (int w = 1; w < kRankThreads / 64; ++w) max_coord = fmaxf(max_coord, s_max[w]);

    // bitonic sort (ascending) of the first `n2` keys; empty slots hold ~0 and sink to the end
    int n2 = 1;
    while (n2 < count) n2 <<= 1;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += kRankThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }

    const float off_unit = max_coord + 1.0f;
    for (int i = tid; i < count; i += kRankThreads) {
        const int a = (int)(keys[i] & 0xffffffffu);
        const float* r = P + (long long)a * (5 + nc);
        float best = r[5];
        int bc = 0;
        for (int c = 1; c < nc; ++c) { const float v = r[5 + c]; if (v > best) { best = v; bc = c; } }
        const float hw = r[2] / 2, hh = r[3] / 2;
        const float off = (float)bc * off_unit;
        const float x1 = (r[0] - hw) + off, y1 = (r[1] - hh) + off, x2 = (r[0] + hw) + off, y2 = (r[1] + hh) + off;
        float* o = rec + (long long)i * kRec;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
        o[4] = (x2 - x1) * (y2 - y1);
        o[5] = __builtin_bit_cast(float, a);
        o[6] = 0.0f; o[7] = 0.0f;
    }
    if (tid == 0) *count_out = count;
}

// One wave per 64x64 tile (row block rb <= col block cb).  Lane l owns row i = rb*64 + l.
__global__ __launch_bounds__(64) void nms_mask_kernel(unsigned char* ws, PostLayout L, float thr, int tiles_per_image_cap) {
    const int img = blockIdx.y;
    unsigned char* wsi = ws + (long long)img * L.image_bytes;
    const float* rec = reinterpret_cast<const float*>(wsi + L.rec_off);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(wsi + L.mask_off);
    const int count = *reinterpret_cast<const int*>(wsi + L.count_off);
    const int nblk = (count + 63) / 64;
    const long long ntiles = (long long)nblk * (nblk + 1) / 2;
    __shared__ float cbox[64 * 5];
    const int lane = threadIdx.x;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // unrank t -> (rb, cb) with rb <= cb, row-major over the upper triangle
        int rb = 0;
        long long rem = t;
        while (rem >= nblk - rb) { rem -= nblk - rb; ++rb; }
        const int cb = rb + (int)rem;
        const int j0 = cb * 64;
        __syncthreads();
        if (j0 + lane < count) {
            const float* r = rec + (long long)(j0 + lane) * kRec;
            cbox[lane * 5 + 0] = r[0]; cbox[lane * 5 + 1] = r[1]; cbox[lane * 5 + 2] = r[2];
            cbox[lane * 5 + 3] = r[3]; cbox[lane * 5 + 4] = r[4];
        }
        __syncthreads();
        const int i = rb * 64 + lane;
        if (i < count) {
            const float* r = rec + (long long)i * kRec;
            const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3], ai = r[4];
            unsigned long long bits = 0ull;
            const int jn = (count - j0) < 64 ? (count - j0) : 64;
            for (int jj = 0; jj < jn; ++jj) {
                if (j0 + jj <= i) continue;
                const float lx = fmaxf(x1, cbox[jj * 5 + 0]), ly = fmaxf(y1, cbox[jj * 5 + 1]);
                const float rx = fminf(x2, cbox[jj * 5 + 2]), ry = fminf(y2, cbox[jj * 5 + 3]);
                const float w = fmaxf(rx - lx, 0.0f), h = fmaxf(ry - ly, 0.0f);
                const float inter = w * h;
                const float iou = inter / (ai + cbox[jj * 5 + 4] - inter);
                if (iou > thr) bits |= (1ull << jj);
            }
            mask[(long long)i * L.words + cb] = bits;
        }
    }
    (void)tiles_per_image_cap;
}

// One 256-thread workgroup per image.  The walk over the sorted list is serial in 64-box chunks (a chunk's survivors
// depend on every earlier chunk), so the latency of ONE chunk step is what matters (185 steps when every anchor is a
// candidate): wave 0 resolves the chunk from its diagonal word (prefetched one chunk ahead) with 64 readlane steps;
// then ALL threads fold the chunk's 64 mask rows into the running `removed` bitmap, one 64-bit word column per thread,
// the 64 row loads issued unconditionally in batches of 16 (independent, coalesced along the row) and OR-ed in under
// the survivor bits — instead of one wave chasing a data-dependent load per surviving row.
constexpr int kSweepThreads = 256;
__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(const float* pred, int A, int nc, unsigned char* ws,
                                                                  PostLayout L, int max_det, float* out_det,
                                                                  int* out_index, int* out_count) {
    SY_DYN_SMEM(smem);
    unsigned long long* removed = reinterpret_cast<unsigned long long*>(smem);     // [L.words] suppressed-so-far bitmap
    unsigned long long* kept = removed + L.words;                                   // [L.words] survivors of every chunk
    int* base = reinterpret_cast<int*>(kept + L.words);                             // [L.words] survivors before the chunk
    const int img = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool wave0 = tid < 64;
    unsigned char* wsi = ws + (long long)img * L.image_bytes;
    const float* rec = reinterpret_cast<const float*>(wsi + L.rec_off);
    const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(wsi + L.mask_off);
    const int count = *reinterpret_cast<const int*>(wsi + L.count_off);
    const int nblk = (count + 63) / 64;
    const float* P = pred + (long long)img * A * (5 + nc);
    float* det = out_det + (long long)img * max_det * 7;
    int* oidx = out_index + (long long)img * max_det;

    for (int w = tid; w < nblk; w += kSweepThreads) removed[w] = 0ull;
    unsigned long long diag_next = (wave0 && lane < count) ? mask[(long long)lane * L.words] : 0ull;
    __syncthreads();
    // ---- the serial walk: nothing but the survivor bits of each chunk and the bitmap update is on this chain ----
    for (int k = 0; k < nblk; ++k) {
        if (wave0) {
            const unsigned long long diag = diag_next;
            const int in = (k + 1) * 64 + lane;                        // next chunk's diagonal word, in flight during the fold
            diag_next = (k + 1 < nblk && in < count) ? mask[(long long)in * L.words + k + 1] : 0ull;
            unsigned long long rem = sy_uniform64(removed[k]);         // rem / alive / row live in SGPRs: a scalar loop
            unsigned long long alive = 0ull;
            const int nl = sy_uniform((count - k * 64) < 64 ? (count - k * 64) : 64);
            for (int l = 0; l < nl; ++l) {
                const unsigned long long row = sy_readlane64(diag, l);
                if (!((rem >> l) & 1ull)) { alive |= (1ull << l); rem |= row; }
            }
            if (lane == 0) kept[k] = alive;
        }
        __syncthreads();
        // fold the kept rows of this chunk into the running bitmap of later chunks: thread = word column (L.words <=
        // kSweepThreads); the 64 row loads are issued unconditionally, all before the first use (rows past `count` are
        // never selected).  Prefetching the next chunk's rows during the resolve was measured and bought nothing: the
        // ~3.3 us per chunk are the two barriers and ~400 VALU instructions, not load latency.
        const unsigned long long alive = kept[k];
        const int w = k + 1 + tid;
        if (alive != 0ull && w < nblk) {
            const unsigned long long* col = mask + (long long)(k * 64) * L.words + w;
            unsigned long long v[64];
#pragma unroll
            for (int j = 0; j < 64; ++j) v[j] = col[(long long)j * L.words];
            unsigned long long acc = removed[w];
#pragma unroll
            for (int j = 0; j < 64; ++j)
                if ((alive >> j) & 1ull) acc |= v[j];
            removed[w] = acc;
        }
        __syncthreads();
    }
    // ---- detections, in order, by all threads: position = survivors of earlier chunks + earlier bits of the own chunk ----
    if (tid == 0) {
        int n = 0;
        for (int k = 0; k < nblk; ++k) { base[k] = n; n += __popcll(kept[k]); }
        out_count[img] = n < max_det ? n : max_det;
    }
    __syncthreads();
    for (int i = tid; i < count; i += kSweepThreads) {
        const int k = i >> 6, b = i & 63;
        const unsigned long long alive = kept[k];
        if (!((alive >> b) & 1ull)) continue;
        const int pos = base[k] + __popcll(alive & ((1ull << b) - 1ull));
        if (pos >= max_det) continue;
        const int a = __builtin_bit_cast(int, rec[(long long)i * kRec + 5]);
        const float* r = P + (long long)a * (5 + nc);
        float best = r[5];
        int bc = 0;
        for (int c = 1; c < nc; ++c) { const float v = r[5 + c]; if (v > best) { best = v; bc = c; } }
        const float hw = r[2] / 2, hh = r[3] / 2;
        float* d = det + (long long)pos * 7;
        d[0] = r[0] - hw; d[1] = r[1] - hh; d[2] = r[0] + hw; d[3] = r[1] + hh;
        d[4] = r[4]; d[5] = best; d[6] = (float)bc;
        oidx[pos] = a;
    }
}

// ---- sy_head_decode: the head's box decode / objectness sigmoid as a stand-alone pass over [B, A, 5+nc] -------------------------
struct DecodeLevels {
    int n;
    int a0[8], w[8];          // first anchor of the level, its grid width
    float stride[8];
};

__global__ __launch_bounds__(256) void head_decode_kernel(float* out, int B, int A, int nch, DecodeLevels L, int flags) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * A) return;
    const int a = (int)(i % A);
    float* r = out + i * nch;
    if (flags & 1) {
        int l = 0;
        while (l + 1 < L.n && a >= L.a0[l + 1]) ++l;
        const int k = a - L.a0[l];
        const float gx = (float)(k % L.w[l]), gy = (float)(k / L.w[l]), st = L.stride[l];
        // (xy + grid) * stride, exp(wh) * stride: the reference's two statements, each rounded once (no contraction: -ffp-contract=off)
        r[0] = (r[0] + gx) * st;
        r[1] = (r[1] + gy) * st;
        r[2] = expf(r[2]) * st;
        r[3] = expf(r[3]) * st;
    }
    if (flags & 2) r[4] = 1.0f / (1.0f + expf(-r[4]));
    if (flags & 4) {            // (cx, cy, w, h) -> (x1, y1, x2, y2) in place: yolox.utils.postprocess's first four statements, each rounded once
        const float cx = r[0], cy = r[1], hw = r[2] / 2.0f, hh = r[3] / 2.0f;
        r[0] = cx - hw; r[1] = cy - hh; r[2] = cx + hw; r[3] = cy + hh;
    }
}

}  // namespace

extern "C" int sy_head_decode(float* out, int B, int A, int nch, const int32_t* level_h, const int32_t* level_w,
                              const float* level_stride, int nlevels, int flags, void* stream) {
    if (out == nullptr || B <= 0 || A <= 0 || nch < 5 || (flags & ~7) != 0) return SY_ERR_ARG;
    DecodeLevels L;
    L.n = 0;
    if (flags & 1) {
        if (level_h == nullptr || level_w == nullptr || level_stride == nullptr || nlevels < 1 || nlevels > 8) return SY_ERR_ARG;
        int a0 = 0;
        for (int l = 0; l < nlevels; ++l) {
            if (level_h[l] <= 0 || level_w[l] <= 0) return SY_ERR_ARG;
            L.a0[l] = a0; L.w[l] = level_w[l]; L.stride[l] = level_stride[l];
            a0 += level_h[l] * level_w[l];
        }
        if (a0 != A) return SY_ERR_ARG;
        L.n = nlevels;
    }
    const long long n = (long long)B * A;
    SY_LAUNCH(head_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, B, A, nch, L, flags);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

extern "C" int64_t sy_postprocess_workspace_bytes(int B, int A) {
    if (B <= 0 || A <= 0) return 0;
    return make_layout(A).image_bytes * (int64_t)B;
}

extern "C" int sy_postprocess(const float* pred, int B, int A, int num_classes, float conf_thre, float nms_thre,
                              int max_det, float* out_det, int32_t* out_index, int32_t* out_count, void* workspace,
                              void* stream) {
    if (pred == nullptr || out_det == nullptr || out_index == nullptr || out_count == nullptr || workspace == nullptr)
        return SY_ERR_ARG;
    if (B <= 0 || A <= 0 || num_classes <= 0 || max_det <= 0) return SY_ERR_ARG;
    const int sort_n = next_pow2(A);
    if (sort_n > 16384) return SY_ERR_UNSUPPORTED;          // 128 KiB of LDS keys (160 KiB per CU on gfx950); words <= 256 = kSweepThreads
    PostLayout L = make_layout(A);
    const size_t rank_smem = (size_t)sort_n * 8 + 128;
#ifndef SY_EMU
    static sy_dev_once attr_done;
    if (attr_done.need()) {
        if (hipFuncSetAttribute((const void*)nms_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8 + 128) != hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    SY_LAUNCH(nms_rank_kernel, dim3(B), dim3(kRankThreads), rank_smem, stream, pred, A, num_classes, conf_thre,
              (unsigned char*)workspace, L, sort_n);
    if (SY_LAUNCH_OK() != 0) return SY_ERR_LAUNCH;
    SY_LAUNCH(nms_mask_kernel, dim3(512, B), dim3(64), 0, stream, (unsigned char*)workspace, L, nms_thre, 0);
    if (SY_LAUNCH_OK() != 0) return SY_ERR_LAUNCH;
    SY_LAUNCH(nms_sweep_kernel, dim3(B), dim3(kSweepThreads), (size_t)L.words * 20, stream, pred, A, num_classes,
              (unsigned char*)workspace, L, max_det, out_det, out_index, out_count);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
