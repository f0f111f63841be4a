// tape.hip — native launch tapes (sy_tape_* of include/streamyolo_hip.h): record a plan's step once, replay it from C.
//
// The reference has nothing like this: its trainer re-dispatches ~2000 eager ops per iteration from Python
// (exps/train_utils/double_trainer.py:95-131).  Here a step is a static launch list; the tape holds that
// list as closures captured by SY_LAUNCH (sy_device.h) plus the plan's control marks:
//
//   SIDE / FORK   main records an event, the side stream waits for it; SIDE also moves the launch cursor to the side stream
//   SIDE_NW       cursor to the side stream without a new dependency
//   MAIN(slot)    cursor back to the main stream; slot >= 0: an event on the side stream marks raw-gradient ring slot `slot` free
//   ACQUIRE(slot) the main stream waits for that event before it overwrites the slot
//   JOIN          the main stream waits for everything issued on the side stream
//   BREAK(id)     return to the caller (a torch snippet of the plan runs there), resume with the returned position
//   BUCKET(k)     gradient bucket k is final: returns to the caller only when `stop_buckets` (data-parallel runs)
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

#include <vector>

namespace {

struct Entry {
    int kind, arg;
    std::function<void(void*)> fn;
};

constexpr int kMaxSlots = 32;

struct Tape {
    std::vector<Entry> entries;
    int launches = 0;
    // replay state (persists across the BREAK / BUCKET returns of one pass)
    bool on_side = false;
    int ei = 0;
    int ring_done[kMaxSlots];
#ifndef SY_EMU
    std::vector<hipEvent_t> pool;
    hipEvent_t next_event() {
        if (ei == (int)pool.size()) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
            pool.push_back(ev);
        }
        return pool[ei++];
    }
    ~Tape() {
        for (hipEvent_t ev : pool) (void)hipEventDestroy(ev);
    }
#endif
};

thread_local Tape* g_rec = nullptr;

}  // namespace

bool sy_tape_recording() { return g_rec != nullptr; }

void sy_tape_push(std::function<void(void*)>&& fn) {
    g_rec->entries.push_back(Entry{SY_TAPE_LAUNCH, 0, std::move(fn)});
    g_rec->launches++;
}

extern "C" void* sy_tape_begin(void) {
    if (g_rec != nullptr) return nullptr;                      // recordings do not nest
    g_rec = new Tape();
    return g_rec;
}

extern "C" int sy_tape_mark(int kind, int arg) {
    if (g_rec == nullptr || kind <= SY_TAPE_LAUNCH || kind > SY_TAPE_BUCKET) return SY_ERR_ARG;
    if ((kind == SY_TAPE_MAIN || kind == SY_TAPE_ACQUIRE) && arg >= kMaxSlots) return SY_ERR_ARG;
    g_rec->entries.push_back(Entry{kind, arg, nullptr});
    return SY_OK;
}

extern "C" void* sy_tape_end(void) {
    Tape* t = g_rec;
    g_rec = nullptr;
    return t;
}

extern "C" int sy_tape_size(const void* tape, int* n_entries, int* n_launches) {
    if (tape == nullptr) return SY_ERR_ARG;
    const Tape* t = (const Tape*)tape;
    if (n_entries != nullptr) *n_entries = (int)t->entries.size();
    if (n_launches != nullptr) *n_launches = t->launches;
    return SY_OK;
}

extern "C" void sy_tape_free(void* tape) {
    if (tape != nullptr && tape != g_rec) delete (Tape*)tape;
}

extern "C" int sy_tape_replay(void* tape, void* main_stream, void* side_stream, int* pos, int stop_buckets, int* stop_kind,
                              int* stop_arg, int* stop_on_side) {
    if (tape == nullptr || pos == nullptr || stop_kind == nullptr || stop_arg == nullptr || g_rec != nullptr) return SY_ERR_ARG;
    Tape* t = (Tape*)tape;
    const int n = (int)t->entries.size();
    int i = *pos;
    if (i < 0 || i > n) return SY_ERR_ARG;
    if (i == 0) {
        t->on_side = false;
        t->ei = 0;
        for (int s = 0; s < kMaxSlots; ++s) t->ring_done[s] = -1;
    }
    const bool two = side_stream != nullptr && side_stream != main_stream;
#ifndef SY_EMU
    hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
#endif
    void* cur = (t->on_side && two) ? side_stream : main_stream;
    for (; i < n; ++i) {
        Entry& e = t->entries[i];
        switch (e.kind) {
            case SY_TAPE_LAUNCH:
                e.fn(cur);
                break;
            case SY_TAPE_BREAK:
            case SY_TAPE_BUCKET:
                if (e.kind == SY_TAPE_BUCKET && !stop_buckets) break;
                *pos = i + 1;
                *stop_kind = e.kind;
                *stop_arg = e.arg;
                if (stop_on_side != nullptr) *stop_on_side = (t->on_side && two) ? 1 : 0;
                return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
            case SY_TAPE_SIDE:
            case SY_TAPE_FORK:
                if (!two) break;
#ifndef SY_EMU
                {
                    hipEvent_t ev = t->next_event();
                    if (ev == nullptr || hipEventRecord(ev, ms) != hipSuccess || hipStreamWaitEvent(ss, ev, 0) != hipSuccess)
                        return SY_ERR_LAUNCH;
                }
#endif
                if (e.kind == SY_TAPE_SIDE) { cur = side_stream; t->on_side = true; }
                break;
            case SY_TAPE_SIDE_NW:
                if (!two) break;
                cur = side_stream; t->on_side = true;
                break;
            case SY_TAPE_MAIN:
                if (!two) break;
#ifndef SY_EMU
                if (e.arg >= 0) {
                    hipEvent_t ev = t->next_event();
                    if (ev == nullptr || hipEventRecord(ev, ss) != hipSuccess) return SY_ERR_LAUNCH;
                    t->ring_done[e.arg] = t->ei - 1;
                }
#endif
                cur = main_stream; t->on_side = false;
                break;
            case SY_TAPE_ACQUIRE:
                if (!two) break;
#ifndef SY_EMU
                if (t->ring_done[e.arg] >= 0) {
                    if (hipStreamWaitEvent(ms, t->pool[t->ring_done[e.arg]], 0) != hipSuccess) return SY_ERR_LAUNCH;
                    t->ring_done[e.arg] = -1;
                }
#endif
                break;
            case SY_TAPE_JOIN:
                if (!two) break;
#ifndef SY_EMU
                {
                    hipEvent_t ev = t->next_event();
                    if (ev == nullptr || hipEventRecord(ev, ss) != hipSuccess || hipStreamWaitEvent(ms, ev, 0) != hipSuccess)
                        return SY_ERR_LAUNCH;
                }
#endif
                break;
            default:
                return SY_ERR_ARG;
        }
    }
    *pos = n;
    *stop_kind = SY_TAPE_END;
    *stop_arg = 0;
    if (stop_on_side != nullptr) *stop_on_side = 0;
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
