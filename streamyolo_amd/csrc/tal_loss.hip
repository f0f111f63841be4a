// tal_loss.hip — SimOTA label assignment + Trend-Aware loss, forward AND gradient, for a whole
// batch, with no host synchronisation.
//
// Replaces TALHead.get_losses / get_assignments / get_in_boxes_info / dynamic_k_matching of the
// reference (exps/model/tal_head.py:262-470, :479-592, :594-677, :679-712): a per-image Python loop
// of ~50 tiny kernels with `.item()` / `int()` host syncs and a `torch.cuda.empty_cache()` per image
// (:306-307, :376, :690, :702), followed by autograd's backward over the same small tensors.
//
//   kernel 1a tal_prep     grid over anchors x images
//       candidates  = anchors whose centre lies in any GT box or in any GT's 2.5-stride centre square (:644-672); their decoded
//                     box, objectness and class-cost base; the GT bookkeeping incl. the trend weights (:394-406)
//   kernel 1b tal_match    one workgroup per (GT, image)
//       cost[g][a]  = BCE(sqrt(sigmoid(cls)*sigmoid(obj)), onehot_g) + 3*(-log(IoU+1e-8)) + 1e5*[not in both]
//                     (:534-553; BCE log terms clamped at -100 like F.binary_cross_entropy)
//       dynamic k   = clamp(int(sum of the 10 largest IoUs of g), 1)        (:685-687)
//       matching    = the k_g cheapest candidates of the GT                  (:688-692)
//   kernel 1c tal_resolve  one workgroup per image
//       conflicts   = an anchor claimed by several GTs keeps the arg-min cost over ALL GTs  (:696-700)
//       plus per-image partial sums of the IoU / L1 losses needed by the TAL weight normalisation (:429-438)
//   (Round 4: the three were ONE 1024-thread workgroup per image — 0.25 ms with eight workgroups on a 256-CU chip and nothing else
//    runnable between the forward and the backward pass; now ~0.04 ms.)
//   kernel 2  tal_grad     grid-stride over B*A anchors
//       loss = 5 * sum(w_iou (1 - IoU^2))/N + sum BCE(obj)/N + sum BCE(cls | fg)/N + sum(w_l1 |l1|)/N  (:441-461)
//       and its closed-form gradient w.r.t. the RAW head output (decode chain rule included: xy*stride,
//       wh = exp(v)*stride), written for all 5+nc channels of every anchor.
// Ties (equal costs / IoUs) resolve to the lower candidate index; the reference leaves them to
// torch.topk, so parity is defined on tie-free inputs.
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

namespace {

constexpr int kAssignThreads = 1024;
constexpr int kMaxLevels = 8;
constexpr int kMaxGT = 128;
constexpr int kGradBlocksMax = 2048;      // workgroups of tal_grad_kernel = rows of its partial-sum buffer (workspace tail)

struct TalGeom {
    int nlevels;
    int h[kMaxLevels], w[kMaxLevels], a0[kMaxLevels + 1];
    float stride[kMaxLevels];
};

struct TalLayout {       // per-image workspace slices (byte offsets)
    long long cand_idx, cbox, cobj, csum, cost, iou, mcnt, mgt, agt, aiou, part, image_bytes;
    int acap;
};

inline TalLayout tal_layout(int A, int max_gt) {
    TalLayout L;
    long long o = 0;
    L.acap = (A + 63) / 64 * 64;
    auto take = [&](long long bytes) { long long r = o; o += (bytes + 255) / 256 * 256; return r; };
    L.cand_idx = take(4LL * L.acap);    // per ANCHOR: 1 = candidate (centre in a GT box or centre square)
    L.cbox = take(16LL * L.acap);
    L.cobj = take(4LL * L.acap);
    L.csum = take(4LL * L.acap);
    L.cost = 0;                         // (the pairwise cost / IoU matrices of the single-workgroup version are gone: the per-GT
    L.iou = 0;                          //  workgroups select in one pass, conflicts recompute their few rows)
    L.mcnt = take(4LL * L.acap);
    L.mgt = take(4LL * L.acap);
    L.agt = take(4LL * L.acap);         // per ANCHOR: matched GT index or -1
    L.aiou = take(4LL * L.acap);        // per ANCHOR: IoU with the matched GT
    L.part = take((16 + kMaxGT) * 4);   // [0..4] partial sums, [5] #GT, [16+g] trend weight of GT g
    L.image_bytes = o;
    return L;
}

__device__ __forceinline__ void anchor_geom(const TalGeom& g, int a, float& gx, float& gy, float& s) {
    int l = 0;
    while (l + 1 < g.nlevels && a >= g.a0[l + 1]) ++l;
    const int r = a - g.a0[l];
    const int y = r / g.w[l];
    gx = (float)(r - y * g.w[l]);
    gy = (float)y;
    s = g.stride[l];
}

__device__ __forceinline__ float clamp_log(float p) { float l = logf(p); return l < -100.0f ? -100.0f : l; }

// IoU of two cxcywh boxes, yolox bboxes_iou(xyxy=False): no epsilon
__device__ __forceinline__ float iou_cxcywh(float ax, float ay, float aw, float ah, float bx, float by, float bw, float bh) {
    const float lx = fmaxf(ax - aw / 2, bx - bw / 2), ly = fmaxf(ay - ah / 2, by - bh / 2);
    const float rx = fminf(ax + aw / 2, bx + bw / 2), ry = fminf(ay + ah / 2, by + bh / 2);
    const float en = (lx < rx && ly < ry) ? 1.0f : 0.0f;
    const float inter = (rx - lx) * (ry - ly) * en;
    return inter / (aw * ah + bw * bh - inter);
}

// wave-wide arg-min of (value, index): smaller value wins, then smaller index
__device__ __forceinline__ void wave_argmin(float& v, int& i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// ---- shared pieces of the three assignment kernels -------------------------------------------------------------------------------
struct TalWs {            // typed views of one image's workspace slice
    int* cand; float* cbox; float* cobj; float* csum; int* mcnt; int* mgt; int* agt; float* aiou; float* part;
    __device__ __forceinline__ TalWs(unsigned char* wsi, const TalLayout& L)
        : cand((int*)(wsi + L.cand_idx)), cbox((float*)(wsi + L.cbox)), cobj((float*)(wsi + L.cobj)), csum((float*)(wsi + L.csum)),
          mcnt((int*)(wsi + L.mcnt)), mgt((int*)(wsi + L.mgt)), agt((int*)(wsi + L.agt)), aiou((float*)(wsi + L.aiou)),
          part((float*)(wsi + L.part)) {}
};

// nlabel = #rows with sum > 0; the FIRST nlabel rows are used (:285, :317-319).  Called by every thread of the workgroup.
__device__ __forceinline__ int tal_count_rows(const float* rows, int max_labels, int* s_cnt) {
    if (threadIdx.x == 0) *s_cnt = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < max_labels; t += blockDim.x) {
        const float* r = rows + t * 5;
        if (r[0] + r[1] + r[2] + r[3] + r[4] > 0.0f) atomicAdd(s_cnt, 1);
    }
    __syncthreads();
    const int n = *s_cnt;
    __syncthreads();
    return n;
}

// centre of anchor (gx, gy, stride st) inside GT box / inside its 2.5-stride centre square (:644-672)
__device__ __forceinline__ void tal_in_box(float gx, float gy, float st, float bx, float by, float bw, float bh, bool& inb, bool& inc) {
    const float xc = gx * st + 0.5f * st, yc = gy * st + 0.5f * st, rad = 2.5f * st;
    inb = fminf(fminf(xc - (bx - 0.5f * bw), yc - (by - 0.5f * bh)), fminf((bx + 0.5f * bw) - xc, (by + 0.5f * bh) - yc)) > 0.0f;
    inc = fminf(fminf(xc - (bx - rad), yc - (by - rad)), fminf((bx + rad) - xc, (by + rad) - yc)) > 0.0f;
}

// IoU and SimOTA cost of candidate anchor `a` against one GT (:534-553).  A diverged step (exp(raw) -> inf) makes IoU / cost NaN,
// and NaN fails every comparison of the selections: NaN IoU counts as 0 and NaN cost as +huge, so every selection still finds a
// candidate (the reference would carry the NaN into the loss; here the loss terms computed from the raw tensor stay NaN, only the
// indexing is safe).
struct TalCand { float b0, b1, b2, b3, obj, csum, rc; };      // what tal_pair reads of a candidate: decoded box, objectness, class-cost
                                                              // base, the GT class's logit
__device__ __forceinline__ TalCand tal_fetch(int a, const float* r, const TalWs& W, int gcls) {
    TalCand c;
    c.b0 = W.cbox[a * 4]; c.b1 = W.cbox[a * 4 + 1]; c.b2 = W.cbox[a * 4 + 2]; c.b3 = W.cbox[a * 4 + 3];
    c.obj = W.cobj[a]; c.csum = W.csum[a]; c.rc = r[5 + gcls];
    return c;
}
__device__ __forceinline__ void tal_pair(const TalGeom& geom, int a, const TalCand& k, float bx, float by, float bw, float bh,
                                         float& iou_s, float& cst) {
    float gx, gy, st;
    anchor_geom(geom, a, gx, gy, st);
    bool inb, inc;
    tal_in_box(gx, gy, st, bx, by, bw, bh, inb, inc);
    const float iou = iou_cxcywh(bx, by, bw, bh, k.b0, k.b1, k.b2, k.b3);
    const float p = sqrtf((1.0f / (1.0f + expf(-k.rc))) * k.obj);
    const float cls_cost = k.csum - (-clamp_log(1.0f - p)) + (-clamp_log(p));
    iou_s = (iou == iou) ? iou : 0.0f;
    const float c = cls_cost + 3.0f * (-logf(iou_s + 1e-8f)) + ((inb && inc) ? 0.0f : 100000.0f);
    cst = (c == c) ? c : 3.0e38f;
}
__device__ __forceinline__ void tal_pair(const TalGeom& geom, int a, const float* r, const TalWs& W, float bx, float by, float bw,
                                         float bh, int gcls, float& iou_s, float& cst) {
    tal_pair(geom, a, tal_fetch(a, r, W, gcls), bx, by, bw, bh, iou_s, cst);
}

// ---- kernel 1a: per anchor — candidate test, decoded box / objectness / class-cost base of the candidates, cleared match state;
//      block (0, img) also does the GT bookkeeping (trend weights, :394-406, :429).  Grid (ceil(A / 256), B).
__global__ __launch_bounds__(256) void tal_prep_kernel(const float* raw, int A, int nc, const float* labels, const float* support,
                                                       int max_labels, TalGeom geom, float gamma, float ignore_thr,
                                                       float ignore_value, unsigned char* ws, TalLayout L) {
    SY_TL_BEGIN(14);
    __shared__ float s_gt[kMaxGT][4];
    __shared__ int s_cnt;
    const int img = blockIdx.y, tid = threadIdx.x;
    const int nch = 5 + nc;
    const float* R = raw + (long long)img * A * nch;
    const float* LB = labels + (long long)img * max_labels * 5;
    const float* SP = support + (long long)img * max_labels * 5;
    const TalWs W(ws + (long long)img * L.image_bytes, L);
    int G = tal_count_rows(LB, max_labels, &s_cnt);
    if (G > kMaxGT) G = kMaxGT;
    for (int g = tid; g < G; g += 256) {
        const float* r = LB + g * 5;
        s_gt[g][0] = r[1]; s_gt[g][1] = r[2]; s_gt[g][2] = r[3]; s_gt[g][3] = r[4];
    }
    if (blockIdx.x == 0) {
        const int S = tal_count_rows(SP, max_labels, &s_cnt);
        if (tid < 16) W.part[tid] = (tid == 5) ? (float)G : 0.0f;
        for (int g = tid; g < G; g += 256) {
            const float* r = LB + g * 5;
            float tr = 1.0f;
            if (S > 0) {
                tr = -INFINITY;
                for (int s = 0; s < S; ++s) {
                    const float* q = SP + s * 5;
                    tr = fmaxf(tr, iou_cxcywh(r[1], r[2], r[3], r[4], q[1], q[2], q[3], q[4]));
                }
                if (tr < ignore_thr) tr = ignore_value;
            }
            W.part[16 + g] = 1.0f / (powf(tr, gamma) + 1e-8f);
        }
    }
    __syncthreads();
    const int a = blockIdx.x * 256 + tid;
    if (a >= A) return;
    bool is_cand = false;
    float gx, gy, st;
    anchor_geom(geom, a, gx, gy, st);
    for (int g = 0; g < G && !is_cand; ++g) {
        bool inb, inc;
        tal_in_box(gx, gy, st, s_gt[g][0], s_gt[g][1], s_gt[g][2], s_gt[g][3], inb, inc);
        is_cand = inb || inc;
    }
    W.cand[a] = is_cand ? 1 : 0;
    W.mcnt[a] = 0;
    W.mgt[a] = -1;
    W.agt[a] = -1;
    W.aiou[a] = 0.0f;
    if (is_cand) {
        const float* r = R + (long long)a * nch;
        W.cbox[a * 4 + 0] = (r[0] + gx) * st;
        W.cbox[a * 4 + 1] = (r[1] + gy) * st;
        W.cbox[a * 4 + 2] = expf(r[2]) * st;
        W.cbox[a * 4 + 3] = expf(r[3]) * st;
        const float so = 1.0f / (1.0f + expf(-r[4]));
        W.cobj[a] = so;
        float acc = 0.0f;
        for (int k = 0; k < nc; ++k) {
            const float p = sqrtf((1.0f / (1.0f + expf(-r[5 + k]))) * so);
            acc += -clamp_log(1.0f - p);
        }
        W.csum[a] = acc;
    }
    SY_TL_END();
}

// ---- kernel 1b: dynamic k + matching, one 256-thread workgroup per (GT row, image); rows >= nlabel exit at once.
// Both selections need at most ten entries of the GT's (IoU | cost) row over the candidates (k = int(sum of the ten largest IoUs) <=
// 10), in (value, smaller anchor index first) order — the candidates are visited in ascending anchor order in the reference, so the
// anchor index IS its tie-break order.  One pass: every thread keeps the ten best of its own anchors for both keys (sorted, in
// registers), each wave merges its lanes' heads entry by entry (wave_argmin) into its own top ten, wave 0 merges the four lists.
__global__ __launch_bounds__(256) void tal_match_kernel(const float* raw, int A, int nc, const float* labels, int max_labels,
                                                        TalGeom geom, unsigned char* ws, TalLayout L) {
    SY_TL_BEGIN(14);
    __shared__ int s_cnt;
    __shared__ int s_wcnt[4];
    __shared__ float s_v[2][4][10];
    __shared__ int s_i[2][4][10];
    const int img = blockIdx.y, g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = 5 + nc;
    const float* R = raw + (long long)img * A * nch;
    const float* LB = labels + (long long)img * max_labels * 5;
    const TalWs W(ws + (long long)img * L.image_bytes, L);
    int G = tal_count_rows(LB, max_labels, &s_cnt);
    if (G > kMaxGT) G = kMaxGT;
    if (g >= G) return;
    const float* lb = LB + g * 5;
    const int gcls = (int)lb[0];
    const float bx = lb[1], by = lb[2], bw = lb[3], bh = lb[4];

    float tv[2][10];
    int ti[2][10];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 10; ++j) { tv[q][j] = INFINITY; ti[q][j] = 0x7fffffff; }
    auto insert = [&](auto q_, float v, int i) {
        constexpr int Q = decltype(q_)::value;
        if (!(v < tv[Q][9] || (v == tv[Q][9] && i < ti[Q][9]))) return;
#pragma unroll
        for (int j = 0; j < 10; ++j) {                    // pass the entry through the sorted list
            const bool lt = v < tv[Q][j] || (v == tv[Q][j] && i < ti[Q][j]);
            const float ov = tv[Q][j];
            const int oi = ti[Q][j];
            tv[Q][j] = lt ? v : ov; ti[Q][j] = lt ? i : oi;
            v = lt ? ov : v; i = lt ? oi : i;
        }
    };
    // Two passes per 64 of the thread's anchors: the candidate flags first (independent loads, all in flight together) as a bit mask,
    // then only the candidates (~10-20 % of the anchors) through the dependent raw-row / decoded-box loads and the two sorted lists.
    // (One pass with `if (!cand) continue` paid a load latency per ANCHOR: 115 us of the step's loss phase, round 6.)  The lists are
    // ordered by (value, anchor index), so the visiting order does not matter.
    int mine = 0;
    for (int base = tid; base < A; base += 256 * 64) {
        unsigned long long m = 0;
#pragma unroll 16
        for (int j = 0; j < 64; ++j) {
            const int a = base + j * 256;
            if (a < A && W.cand[a]) m |= 1ull << j;
        }
        if (m == 0) continue;
        int a_nx = base + __builtin_ctzll(m) * 256;              // the next candidate's loads fly while this one is ranked
        m &= m - 1;
        TalCand k_nx = tal_fetch(a_nx, R + (long long)a_nx * nch, W, gcls);
        for (bool more = true; more;) {
            const int a = a_nx;
            const TalCand k = k_nx;
            more = m != 0;
            if (more) {
                a_nx = base + __builtin_ctzll(m) * 256;
                m &= m - 1;
                k_nx = tal_fetch(a_nx, R + (long long)a_nx * nch, W, gcls);
            }
            ++mine;
            float iou_s, cst;
            tal_pair(geom, a, k, bx, by, bw, bh, iou_s, cst);
            insert(sy_int<0>(), -iou_s, a);               // ten largest IoUs
            insert(sy_int<1>(), cst, a);                  // ten smallest costs
        }
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) s_wcnt[wave] = mine;
    // the wave's own top ten of both keys -> LDS
    sy_static_for<0, 2>([&](auto q_) {
        constexpr int Q = decltype(q_)::value;
        for (int it = 0; it < 10; ++it) {
            float bv = tv[Q][0];
            int bi = ti[Q][0];
            const float mv = bv;
            const int mi = bi;
            wave_argmin(bv, bi);
            if (mv == bv && mi == bi && bi != 0x7fffffff) {   // this lane's head was taken: drop it
#pragma unroll
                for (int j = 0; j < 9; ++j) { tv[Q][j] = tv[Q][j + 1]; ti[Q][j] = ti[Q][j + 1]; }
                tv[Q][9] = INFINITY; ti[Q][9] = 0x7fffffff;
            }
            if (lane == 0) { s_v[Q][wave][it] = bv; s_i[Q][wave][it] = bi; }
        }
    });
    __syncthreads();
    if (wave != 0) return;
    const int C = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (C == 0) return;
    const int nk = C < 10 ? C : 10;
    float v = lane < 40 ? s_v[0][lane / 10][lane % 10] : INFINITY;
    int i = lane < 40 ? s_i[0][lane / 10][lane % 10] : 0x7fffffff;
    float sum = 0.0f;
    for (int it = 0; it < nk; ++it) {
        float bv = v;
        int bi = i;
        wave_argmin(bv, bi);
        if (bi == 0x7fffffff) break;
        if (v == bv && i == bi) { v = INFINITY; i = 0x7fffffff; }
        sum += -bv;
    }
    int kg = (sum == sum && sum < 1.0e9f) ? (int)sum : 1;    // dynamic k = clamp(int(sum of the 10 largest IoUs), 1)  (:685-687)
    if (kg < 1) kg = 1;
    if (kg > C) kg = C;
    v = lane < 40 ? s_v[1][lane / 10][lane % 10] : INFINITY;
    i = lane < 40 ? s_i[1][lane / 10][lane % 10] : 0x7fffffff;
    for (int it = 0; it < kg && it < 10; ++it) {            // the k_g cheapest candidates of this GT (:688-692)
        float bv = v;
        int bi = i;
        wave_argmin(bv, bi);
        if (bi == 0x7fffffff) break;
        if (v == bv && i == bi) { v = INFINITY; i = 0x7fffffff; }
        if (lane == 0) { atomicAdd(&W.mcnt[bi], 1); W.mgt[bi] = g; }
    }
    SY_TL_END();
}

// ---- kernel 1c: conflicts (an anchor claimed by several GTs keeps the arg-min cost over ALL GTs, :696-700), foreground list and the
//      per-image partial sums of the IoU / L1 losses needed by the TAL weight normalisation (:429-438).  One workgroup per image.
__global__ __launch_bounds__(kAssignThreads) void tal_resolve_kernel(const float* raw, int A, int nc, const float* labels,
                                                                     int max_labels, TalGeom geom, int use_l1, unsigned char* ws,
                                                                     TalLayout L) {
    SY_TL_BEGIN(14);
    __shared__ float s_gt[kMaxGT][4];
    __shared__ int s_gcls[kMaxGT];
    __shared__ float s_w[kMaxGT];
    __shared__ int s_cnt;
    __shared__ float s_red[kAssignThreads / 64][8];
    const int img = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kAssignThreads / 64;
    const int nch = 5 + nc;
    const float* R = raw + (long long)img * A * nch;
    const float* LB = labels + (long long)img * max_labels * 5;
    const TalWs W(ws + (long long)img * L.image_bytes, L);
    int G = tal_count_rows(LB, max_labels, &s_cnt);
    if (G > kMaxGT) G = kMaxGT;
    if (G == 0) return;
    if (tid < G) {
        const float* r = LB + tid * 5;
        s_gcls[tid] = (int)r[0];
        s_gt[tid][0] = r[1]; s_gt[tid][1] = r[2]; s_gt[tid][2] = r[3]; s_gt[tid][3] = r[4];
        s_w[tid] = W.part[16 + tid];
    }
    __syncthreads();
    float p_iou = 0, p_wiou = 0, p_l1 = 0, p_wl1 = 0, p_nfg = 0;
    // claimed anchors as a bit mask first (independent loads), then only those, in ascending anchor order (the partial sums keep
    // their summation order) — as in tal_match_kernel
    for (int base = tid; base < A; base += kAssignThreads * 64) {
      unsigned long long m = 0;
#pragma unroll 16
      for (int j = 0; j < 64; ++j) {
          const int a_ = base + j * kAssignThreads;
          if (a_ < A && W.mcnt[a_] != 0) m |= 1ull << j;
      }
      while (m) {
        const int a = base + __builtin_ctzll(m) * kAssignThreads;
        m &= m - 1;
        const int cnt = W.mcnt[a];
        int g = W.mgt[a];
        const float* r = R + (long long)a * nch;
        if (cnt > 1) {
            float bv = INFINITY;
            for (int gg = 0; gg < G; ++gg) {
                float iou_s, v;
                tal_pair(geom, a, r, W, s_gt[gg][0], s_gt[gg][1], s_gt[gg][2], s_gt[gg][3], s_gcls[gg], iou_s, v);
                if (v < bv) { bv = v; g = gg; }
            }
        }
        float miou, unused;
        tal_pair(geom, a, r, W, s_gt[g][0], s_gt[g][1], s_gt[g][2], s_gt[g][3], s_gcls[g], miou, unused);
        W.agt[a] = g;
        W.aiou[a] = miou;
        // IoU loss of this foreground anchor (IOUloss: +1e-16 in the denominator) and its L1 loss
        const float px = W.cbox[a * 4], py = W.cbox[a * 4 + 1], pw = W.cbox[a * 4 + 2], ph = W.cbox[a * 4 + 3];
        const float tx = s_gt[g][0], ty = s_gt[g][1], tw = s_gt[g][2], th = s_gt[g][3];
        const float lx = fmaxf(px - pw / 2, tx - tw / 2), ly = fmaxf(py - ph / 2, ty - th / 2);
        const float rx = fminf(px + pw / 2, tx + tw / 2), ry = fminf(py + ph / 2, ty + th / 2);
        const float en = (lx < rx && ly < ry) ? 1.0f : 0.0f;
        const float inter = (rx - lx) * (ry - ly) * en;
        const float iou = inter / (pw * ph + tw * th - inter + 1e-16f);
        const float il = 1.0f - iou * iou;
        const float w = s_w[g];
        p_iou += il;
        p_wiou += w * il;
        p_nfg += 1.0f;
        if (use_l1) {
            float gx, gy, st;
            anchor_geom(geom, a, gx, gy, st);
            const float l1 = fabsf(r[0] - (tx / st - gx)) + fabsf(r[1] - (ty / st - gy)) +
                             fabsf(r[2] - logf(tw / st + 1e-8f)) + fabsf(r[3] - logf(th / st + 1e-8f));
            p_l1 += l1;
            p_wl1 += w * l1;
        }
      }
    }
    float vals[5] = {p_iou, p_wiou, p_l1, p_wl1, p_nfg};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        float v = vals[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) s_red[wave][j] = v;
    }
    __syncthreads();
    if (tid < 5) {
        float v = 0.0f;
        for (int w = 0; w < NW; ++w) v += s_red[w][tid];
        W.part[tid] = v;
    }
    SY_TL_END();
}

__global__ __launch_bounds__(256) void tal_grad_kernel(const float* raw, int B, int A, int nc, const float* labels,
                                                       int max_labels, TalGeom geom, float gamma, int use_l1,
                                                       const unsigned char* ws, TalLayout L,
                                                       float* d_raw, float* losses, int* fg_mask, void* d_pad, int pad_dtype,
                                                       float* block_part) {
    SY_TL_BEGIN(14);
    __shared__ float s_tot[8];
    __shared__ float s_acc[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // batch totals of the per-image partial sums (every workgroup recomputes them: B is small)
    if (tid < 8) {
        float v = 0.0f;
        for (int b = 0; b < B; ++b) v += ((const float*)(ws + (long long)b * L.image_bytes + L.part))[tid];
        s_tot[tid] = v;
    }
    __syncthreads();
    const float sum_iou = s_tot[0], sum_wiou = s_tot[1], sum_l1 = s_tot[2], sum_wl1 = s_tot[3];
    const float num_fg = s_tot[4], num_gt = s_tot[5];
    const float nf = num_fg > 1.0f ? num_fg : 1.0f;
    const float inv_nf = 1.0f / nf;
    const int nch = 5 + nc;
    float l_iou = 0, l_obj = 0, l_cls = 0, l_l1 = 0;
    const long long total = (long long)B * A;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / A), a = (int)(i - (long long)b * A);
        const unsigned char* wsi = ws + (long long)b * L.image_bytes;
        const int g = ((const int*)(wsi + L.agt))[a];
        const float* r = raw + i * nch;
        float* const d_row = d_raw + i * nch;
        // d(total)/d(raw) of channel k: the fp32 row, and (optional) the same value in the MFMA operand layout of the backward pass's
        // first kernels, [reg 4 | obj | 0 0 0 | cls nc] x 16 in the compute dtype (the columns in between stay zero)
        auto put = [&](int k, float v) {
            d_row[k] = v;
            if (d_pad != nullptr) {
                const long long o = i * 16 + (k < 5 ? k : 3 + k);
                if (pad_dtype == SY_DT_BF16) ((BF16::elem*)d_pad)[o] = BF16::from_f32(v);
                else if (pad_dtype == SY_DT_F16) ((F16::elem*)d_pad)[o] = F16::from_f32(v);
                else ((float*)d_pad)[o] = v;
            }
        };
        const float tobj = g >= 0 ? 1.0f : 0.0f;
        if (fg_mask != nullptr) fg_mask[i] = g >= 0 ? 1 : 0;
        {   // objectness BCE-with-logits on every anchor (:445-447)
            const float z = r[4];
            l_obj += fmaxf(z, 0.0f) - z * tobj + log1pf(expf(-fabsf(z)));
            put(4, (1.0f / (1.0f + expf(-z)) - tobj) * inv_nf);
        }
        if (g < 0) {
            put(0, 0.0f); put(1, 0.0f); put(2, 0.0f); put(3, 0.0f);
            for (int k = 0; k < nc; ++k) put(5 + k, 0.0f);
            continue;
        }
        const float miou = ((const float*)(wsi + L.aiou))[a];
        const float* lb = labels + ((long long)b * max_labels + g) * 5;
        const int gcls = (int)lb[0];
        const float tx = lb[1], ty = lb[2], tw = lb[3], th = lb[4];
        for (int k = 0; k < nc; ++k) {   // class BCE on foreground anchors, target = one-hot * matched IoU (:379-381, :448-452)
            const float z = r[5 + k];
            const float t = (k == gcls) ? miou : 0.0f;
            l_cls += fmaxf(z, 0.0f) - z * t + log1pf(expf(-fabsf(z)));
            put(5 + k, (1.0f / (1.0f + expf(-z)) - t) * inv_nf);
        }
        float gx, gy, st;
        anchor_geom(geom, a, gx, gy, st);
        const float wg = ((const float*)(wsi + L.part))[16 + g];       // trend weight of GT g (written by kernel 1)
        // IoU loss and gradient (:431-433, :442-444)
        const float px = (r[0] + gx) * st, py = (r[1] + gy) * st, pw = expf(r[2]) * st, ph = expf(r[3]) * st;
        const float plx = px - pw / 2, ply = py - ph / 2, prx = px + pw / 2, pry = py + ph / 2;
        const float tlx = tx - tw / 2, tly = ty - th / 2, trx = tx + tw / 2, try_ = ty + th / 2;
        const float lx = fmaxf(plx, tlx), ly = fmaxf(ply, tly), rx = fminf(prx, trx), ry = fminf(pry, try_);
        const bool en = (lx < rx) && (ly < ry);
        const float wi = rx - lx, hi = ry - ly;
        const float inter = en ? wi * hi : 0.0f;
        const float ap = pw * ph, ag = tw * th;
        const float D = ap + ag - inter + 1e-16f;
        const float iou = inter / D;
        const float w_iou = (sum_wiou != 0.0f) ? wg * sum_iou / sum_wiou : 0.0f;
        l_iou += w_iou * (1.0f - iou * iou);
        float dIx = 0, dIy = 0, dIw = 0, dIh = 0;
        if (en) {
            const float dl = plx > tlx ? 1.0f : 0.0f, dr = prx < trx ? 1.0f : 0.0f;
            const float dt = ply > tly ? 1.0f : 0.0f, db = pry < try_ ? 1.0f : 0.0f;
            dIx = hi * (dr - dl);
            dIw = hi * 0.5f * (dr + dl);
            dIy = wi * (db - dt);
            dIh = wi * 0.5f * (db + dt);
        }
        // d iou = (dI * D - I * (dAp - dI)) / D^2 ; loss = 5 * w_iou * (1 - iou^2) / nf
        const float c0 = -2.0f * iou * 5.0f * w_iou * inv_nf / (D * D);
        const float gpx = c0 * (dIx * D - inter * (0.0f - dIx));
        const float gpy = c0 * (dIy * D - inter * (0.0f - dIy));
        const float gpw = c0 * (dIw * D - inter * (ph - dIw));
        const float gph = c0 * (dIh * D - inter * (pw - dIh));
        float d0 = gpx * st, d1 = gpy * st, d2 = gpw * pw, d3 = gph * ph;
        if (use_l1) {   // L1 on the raw regression outputs (:384-391, :435-438, :453-456)
            const float w_l1 = (sum_wl1 != 0.0f) ? wg * sum_l1 / sum_wl1 : 0.0f;
            const float t0 = tx / st - gx, t1 = ty / st - gy, t2 = logf(tw / st + 1e-8f), t3 = logf(th / st + 1e-8f);
            const float e0 = r[0] - t0, e1 = r[1] - t1, e2 = r[2] - t2, e3 = r[3] - t3;
            l_l1 += w_l1 * (fabsf(e0) + fabsf(e1) + fabsf(e2) + fabsf(e3));
            const float c1 = w_l1 * inv_nf;
            d0 += c1 * ((e0 > 0.0f) - (e0 < 0.0f));
            d1 += c1 * ((e1 > 0.0f) - (e1 < 0.0f));
            d2 += c1 * ((e2 > 0.0f) - (e2 < 0.0f));
            d3 += c1 * ((e3 > 0.0f) - (e3 < 0.0f));
        }
        put(0, d0); put(1, d1); put(2, d2); put(3, d3);
    }
    float vals[4] = {l_iou, l_obj, l_cls, l_l1};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = vals[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) s_acc[wave][j] = v;
    }
    __syncthreads();
    // This workgroup's four partial sums (iou, obj, cls, l1) go to ITS row of `block_part`; tal_finish_kernel adds the rows in a
    // fixed order.  (Until round 5 every workgroup added into losses[] with fp32 atomics: the loss scalars of two identical steps
    // differed in their last bits with the arrival order of up to 2048 workgroups.)
    if (tid < 4) block_part[blockIdx.x * 4 + tid] = (s_acc[0][tid] + s_acc[1][tid] + s_acc[2][tid] + s_acc[3][tid]) * inv_nf;
    if (blockIdx.x == 0 && tid == 0) {
        losses[5] = num_fg / (num_gt > 1.0f ? num_gt : 1.0f);
        losses[6] = num_fg;
        losses[7] = num_gt;
    }
    SY_TL_END();
}

// One workgroup of four waves: wave j totals component j of the per-workgroup rows — lane i the rows i, i + 64, ... in index
// order, then a butterfly over the lanes — so the result does not depend on which workgroup of tal_grad_kernel finished first.
__global__ __launch_bounds__(256) void tal_finish_kernel(const float* block_part, int blocks, float* losses) {
    __shared__ float s_tot[4];
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    float v = 0.0f;
    for (int b = lane; b < blocks; b += 64) v += block_part[b * 4 + j];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) s_tot[j] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        // losses: [0] total, [1] 5*iou, [2] l1, [3] conf, [4] cls, [5] num_fg/num_gt, [6] num_fg, [7] num_gt
        const float iou5 = 5.0f * s_tot[0], conf = s_tot[1], cls = s_tot[2], l1 = s_tot[3];
        losses[1] = iou5; losses[3] = conf; losses[4] = cls; losses[2] = l1;
        losses[0] = ((iou5 + conf) + cls) + l1;               // the reference's order (tal_head.py:461)
    }
}

}  // namespace

extern "C" int64_t sy_tal_loss_workspace_bytes(int B, int A, int max_gt) {
    if (B <= 0 || A <= 0 || max_gt <= 0) return 0;
    if (max_gt > kMaxGT) max_gt = kMaxGT;
    return tal_layout(A, max_gt).image_bytes * (int64_t)B + kGradBlocksMax * 4 * (int64_t)sizeof(float);   // + tal_grad_kernel's partial rows
}

extern "C" int sy_tal_loss(const float* raw, int B, int A, int num_classes, const float* labels, const float* support,
                           int max_labels, const int32_t* level_h, const int32_t* level_w, const float* level_stride,
                           int nlevels, float gamma, float ignore_thr, float ignore_value, int use_l1, float* d_raw,
                           float* losses, int32_t* fg_mask, void* workspace, void* d_pad, int pad_dtype, void* stream) {
    if (raw == nullptr || labels == nullptr || support == nullptr || d_raw == nullptr || losses == nullptr ||
        workspace == nullptr || level_h == nullptr || level_w == nullptr || level_stride == nullptr)
        return SY_ERR_ARG;
    if (B <= 0 || A <= 0 || num_classes <= 0 || nlevels <= 0 || nlevels > kMaxLevels) return SY_ERR_ARG;
    if (max_labels <= 0 || max_labels > kMaxGT) return SY_ERR_UNSUPPORTED;
    if (d_pad != nullptr && (num_classes > 8 || pad_dtype < SY_DT_BF16 || pad_dtype > SY_DT_F32)) return SY_ERR_UNSUPPORTED;
    TalGeom g;
    g.nlevels = nlevels;
    int a0 = 0;
    for (int l = 0; l < nlevels; ++l) {
        g.h[l] = level_h[l]; g.w[l] = level_w[l]; g.stride[l] = level_stride[l];
        g.a0[l] = a0;
        a0 += level_h[l] * level_w[l];
    }
    g.a0[nlevels] = a0;
    if (a0 != A) return SY_ERR_ARG;
    TalLayout L = tal_layout(A, max_labels);
    SY_LAUNCH(tal_prep_kernel, dim3((A + 255) / 256, B), dim3(256), 0, stream, raw, A, num_classes, labels, support, max_labels, g,
              gamma, ignore_thr, ignore_value, (unsigned char*)workspace, L);
    SY_LAUNCH(tal_match_kernel, dim3(max_labels, B), dim3(256), 0, stream, raw, A, num_classes, labels, max_labels, g,
              (unsigned char*)workspace, L);
    SY_LAUNCH(tal_resolve_kernel, dim3(B), dim3(kAssignThreads), 0, stream, raw, A, num_classes, labels, max_labels, g, use_l1,
              (unsigned char*)workspace, L);
    if (SY_LAUNCH_OK() != 0) return SY_ERR_LAUNCH;
    long long work = (long long)B * A;
    int blocks = (int)((work + 255) / 256);
    if (blocks > kGradBlocksMax) blocks = kGradBlocksMax;
    float* const block_part = reinterpret_cast<float*>((unsigned char*)workspace + L.image_bytes * (long long)B);
    SY_LAUNCH(tal_grad_kernel, dim3(blocks), dim3(256), 0, stream, raw, B, A, num_classes, labels, max_labels, g, gamma,
              use_l1, (const unsigned char*)workspace, L, d_raw, losses, fg_mask, d_pad, pad_dtype, block_part);
    SY_LAUNCH(tal_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)block_part, blocks, losses);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
// Read-out of the last sy_tal_loss call's assignment (tests / diagnostics): per anchor the matched ground-truth index (-1 =
// background) and its IoU, as the reference's get_assignments returns them (tal_head.py:559-600: matched_gt_inds,
// pred_ious_this_matching) — the foreground MASK alone cannot show an anchor re-matched to another box.
namespace {
__global__ void tal_assignment_kernel(const unsigned char* ws, TalLayout L, int A, int32_t* matched_gt, float* matched_iou) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (a >= A) return;
    const unsigned char* wsi = ws + (long long)b * L.image_bytes;
    const int g = reinterpret_cast<const int*>(wsi + L.agt)[a];
    if (matched_gt != nullptr) matched_gt[(long long)b * A + a] = g;
    if (matched_iou != nullptr) matched_iou[(long long)b * A + a] = g >= 0 ? reinterpret_cast<const float*>(wsi + L.aiou)[a] : 0.0f;
}
}  // namespace

extern "C" int sy_tal_loss_assignment(const void* workspace, int B, int A, int max_labels, int32_t* matched_gt,
                                      float* matched_iou, void* stream) {
    if (workspace == nullptr || B <= 0 || A <= 0 || max_labels <= 0 || max_labels > kMaxGT || (matched_gt == nullptr && matched_iou == nullptr))
        return SY_ERR_ARG;
    SY_LAUNCH(tal_assignment_kernel, dim3((A + 255) / 256, B), dim3(256), 0, stream, (const unsigned char*)workspace,
              tal_layout(A, max_labels), A, matched_gt, matched_iou);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
SY_PROBE_READER(sy_probe_read_tal_loss)
