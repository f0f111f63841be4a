// optim.hip — the whole optimizer + EMA step of a training iteration in ONE launch.
//
// Replaces: `self.scaler.step(self.optimizer)` + `self.ema_model.update(self.model)`
// (exps/train_utils/double_trainer.py:115-119) — in the reference 2 x 768 tiny launches per iteration:
//   * torch.optim.SGD(momentum, nesterov=True) over the three YOLOX parameter groups (BatchNorm weights, conv /
//     linear weights with weight decay, biases), on fp32 master parameters;
//   * yolox ModelEMA.update: every floating state_dict entry v <- d*v + (1-d)*x, d = decay*(1 - exp(-updates/2000))
//     (parameters AND BatchNorm running statistics; d is computed by the host).
// HBM-bound streaming work: per parameter element read p, g, buf, ema and write p, buf, ema (28 B); per buffer
// element 12 B.  One workgroup = one 1024-element chunk of one tensor, found by bisection over the table's chunk
// prefix.  Arithmetic order follows torch.optim.SGD exactly (d = g*scale; d += wd*p; buf = first ? d : m*buf + d;
// d += m*buf; p -= lr*d) with -ffp-contract=off, then the EMA reads the UPDATED parameter, as the reference's
// call order does.
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

namespace {

constexpr int kChunk = 1024;

__global__ __launch_bounds__(256) void sgd_ema_kernel(const sy_optim_entry* entries, int n_entries, float lr, float momentum,
                                                      float grad_scale, float ema_d, int first_step) {
    int lo = 0, hi = n_entries - 1;                           // last entry with chunk0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (entries[mid].chunk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const sy_optim_entry e = entries[lo];
    const long long base = ((long long)blockIdx.x - e.chunk0) * kChunk;
    const float one_minus_d = 1.0f - ema_d;
    for (int k = threadIdx.x; k < kChunk; k += 256) {
        const long long i = base + k;
        if (i >= e.n) break;
        float x = e.p[i];
        if (e.g != nullptr) {
            float d = e.g[i] * grad_scale;
            if (e.weight_decay != 0.0f) d = d + e.weight_decay * x;
            float b = d;
            if (e.buf != nullptr) {
                if (!first_step) b = e.buf[i] * momentum + d;
                e.buf[i] = b;
                d = d + momentum * b;                         // nesterov
            }
            x = x - (lr * e.lr_mult) * d;
            e.p[i] = x;
        }
        if (e.ema != nullptr) {
            float v = e.ema[i] * ema_d;
            v = v + one_minus_d * x;
            e.ema[i] = v;
        }
    }
}

}  // namespace

extern "C" int sy_sgd_ema_step(const sy_optim_entry* entries, int n_entries, int total_chunks, float lr, float momentum,
                               float grad_scale, float ema_decay, int first_step, void* stream) {
    if (entries == nullptr || n_entries <= 0 || total_chunks <= 0) return SY_ERR_ARG;
    SY_LAUNCH(sgd_ema_kernel, dim3(total_chunks), dim3(256), 0, stream, entries, n_entries, lr, momentum, grad_scale, ema_decay,
              first_step);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
