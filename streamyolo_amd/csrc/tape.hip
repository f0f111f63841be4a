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
// and, for plans that run more than two chains (sy_tape_replay_n, streams[0] = main, [1] = side, [2..] = further chains):
//   CUR(k)        cursor to stream k, no new dependency
//   DEP(a, b)     stream a records an event, stream b waits for it (arg = a * 16 + b)
//   SLOT_DONE(s)  event on the CURRENT stream = "this stream's readers of ring slot s are done".  A slot collects one such event
//                 PER STREAM (the weight gradient on its stream, the data gradient(s) of the same raw gradient on the chain(s)
//                 that produced it): the first SLOT_DONE after an acquisition starts a new set.
//   ACQUIRE_CUR(s) the CURRENT stream waits for EVERY event of the slot's set that was recorded on another stream (the set stays:
//                 several chains may acquire the same slot — each frame's half of it)
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

#include <stdlib.h>
#include <vector>

namespace {

struct Entry {
    int kind, arg;
    std::function<void(void*)> fn;
};

constexpr int kMaxSlots = 32, kMaxStreams = 8;

struct Tape {
    std::vector<Entry> entries;
    int launches = 0;
    // attach[i] >= 0 for a LAUNCH entry i: the index of the DEP entry that follows it on its stream with nothing in between — the
    // launch carries that dependency's event as its stop event (sy_tape.h); folded[d] = the event index the launch was given in this
    // pass (-1: the DEP records for itself)
    std::vector<int> attach, folded;
    void plan_attachments();
    // replay state (persists across the BREAK / BUCKET returns of one pass)
    int cur_k = 0;                       // index of the cursor stream (0 = main, 1 = side, ...)
    int ei = 0;                          // events used so far in this pass
    int waits = 0;                       // stream-waits issued so far in this pass
    struct Slot {
        int ev[kMaxStreams];             // latest "done" event of the slot per stream index (-1: none)
        bool acquired;                   // somebody acquired the slot since the last SLOT_DONE: the next one starts a new set
    } slots[kMaxSlots];
    void slot_clear(int s) {
        for (int k = 0; k < kMaxStreams; ++k) slots[s].ev[k] = -1;
        slots[s].acquired = false;
    }
    void slot_done(int s, int k, int ev) {
        if (slots[s].acquired) slot_clear(s);
        slots[s].ev[k] = ev;
    }
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
thread_local void* g_stop_event = nullptr;

// Walk the recorded marks once: a DEP whose producing stream has issued a launch and NOTHING else since (no wait, no record, no
// return to the host on that stream) can ride on that launch.  Anything else on the stream in between — a wait would make the
// original record cover what the stream waited for — keeps the separate record.
void Tape::plan_attachments() {
    const int n = (int)entries.size();
    attach.assign(n, -1);
    folded.assign(n, -1);
    int last[kMaxStreams];
    for (int k = 0; k < kMaxStreams; ++k) last[k] = -1;
    int cur = 0;
    auto touch = [&](int k) { if (k >= 0 && k < kMaxStreams) last[k] = -1; };
    for (int i = 0; i < n; ++i) {
        const Entry& e = entries[i];
        switch (e.kind) {
            case SY_TAPE_LAUNCH: last[cur] = i; break;
            case SY_TAPE_SIDE: touch(0); touch(1); cur = 1; break;
            case SY_TAPE_FORK: touch(0); touch(1); break;
            case SY_TAPE_SIDE_NW: cur = 1; break;
            case SY_TAPE_MAIN: if (e.arg >= 0) touch(1); cur = 0; break;
            case SY_TAPE_ACQUIRE: touch(0); break;
            case SY_TAPE_JOIN: touch(0); touch(1); break;
            case SY_TAPE_BREAK:
            case SY_TAPE_BUCKET: for (int k = 0; k < kMaxStreams; ++k) last[k] = -1; break;
            case SY_TAPE_CUR: cur = e.arg; break;
            case SY_TAPE_DEP: {
                const int f = e.arg >> 4, to = e.arg & 15;
                if (f >= 0 && f < kMaxStreams && last[f] >= 0 && attach[last[f]] < 0) attach[last[f]] = i;
                touch(f); touch(to);
                break;
            }
            case SY_TAPE_SLOT_DONE: touch(cur); break;
            case SY_TAPE_ACQUIRE_CUR: touch(cur); break;
            default: break;
        }
    }
}

}  // namespace

void* sy_tape_stop_event_take() {
    void* const e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}

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
    if (g_rec == nullptr || kind <= SY_TAPE_LAUNCH || kind > SY_TAPE_ACQUIRE_CUR) return SY_ERR_ARG;
    if ((kind == SY_TAPE_MAIN || kind == SY_TAPE_ACQUIRE || kind == SY_TAPE_SLOT_DONE || kind == SY_TAPE_ACQUIRE_CUR) && arg >= kMaxSlots)
        return SY_ERR_ARG;
    if ((kind == SY_TAPE_SLOT_DONE || kind == SY_TAPE_ACQUIRE_CUR || kind == SY_TAPE_ACQUIRE || kind == SY_TAPE_CUR || kind == SY_TAPE_DEP) && arg < 0)
        return SY_ERR_ARG;
    if (kind == SY_TAPE_CUR && arg >= kMaxStreams) return SY_ERR_ARG;
    g_rec->entries.push_back(Entry{kind, arg, nullptr});
    return SY_OK;
}

extern "C" void* sy_tape_end(void) {
    Tape* t = g_rec;
    g_rec = nullptr;
    if (t != nullptr) t->plan_attachments();
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

extern "C" int sy_tape_counters(const void* tape, int* n_events, int* n_waits) {
    if (tape == nullptr) return SY_ERR_ARG;
    const Tape* t = (const Tape*)tape;
    if (n_events != nullptr) *n_events = t->ei;
    if (n_waits != nullptr) *n_waits = t->waits;
    return SY_OK;
}

extern "C" int sy_tape_replay_n(void* tape, void* const* streams, int n_streams, int* pos, int stop_buckets, int* stop_kind,
                                int* stop_arg, int* stop_stream) {
    if (tape == nullptr || pos == nullptr || stop_kind == nullptr || stop_arg == nullptr || g_rec != nullptr || streams == nullptr ||
        n_streams < 1 || n_streams > kMaxStreams)
        return SY_ERR_ARG;
    Tape* t = (Tape*)tape;
    const int n = (int)t->entries.size();
    int i = *pos;
    if (i < 0 || i > n) return SY_ERR_ARG;
    if ((int)t->attach.size() != n) t->plan_attachments();
    if (i == 0) {
        t->cur_k = 0;
        t->ei = 0;
        t->waits = 0;
        for (int& f : t->folded) f = -1;
        for (int s = 0; s < kMaxSlots; ++s) t->slot_clear(s);
    }
    void* const main_stream = streams[0];
    void* const side_stream = n_streams > 1 ? streams[1] : nullptr;
    const bool two = side_stream != nullptr && side_stream != main_stream;
    // a chain index the caller gave no stream for runs on the main stream (one- / two-stream replay of a three-chain tape)
    auto stream_of = [&](int k) -> void* { return (two && k >= 0 && k < n_streams && streams[k] != nullptr) ? streams[k] : main_stream; };
    // event bookkeeping (the same in the emulator build, which issues nothing: tests/test_tape.py counts events and waits there)
    bool failed = false;
    auto record = [&](void* on) -> int {                     // event on stream `on` -> its index in the pool
#ifndef SY_EMU
        hipEvent_t ev = t->next_event();
        if (ev == nullptr || hipEventRecord(ev, (hipStream_t)on) != hipSuccess) failed = true;
        return t->ei - 1;
#else
        (void)on;
        return t->ei++;
#endif
    };
    auto wait = [&](void* who, int ev) {                     // stream `who` waits for event `ev`
        t->waits++;
#ifndef SY_EMU
        if (hipStreamWaitEvent((hipStream_t)who, t->pool[ev], 0) != hipSuccess) failed = true;
#else
        (void)who; (void)ev;
#endif
    };
    static const bool use_stop_events = [] { const char* v = getenv("SY_TAPE_STOP_EVENTS"); return v == nullptr || atoi(v) != 0; }();   // "0": A/B timing
    bool stop_events_now = use_stop_events;
#ifndef SY_EMU
    {   // inside a hipGraph capture the events of the tape become graph edges: keep the plain record / wait form there
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)main_stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) stop_events_now = false;
    }
#endif
    (void)stop_events_now;
    void* cur = stream_of(t->cur_k);
    for (; i < n && !failed; ++i) {
        Entry& e = t->entries[i];
        switch (e.kind) {
            case SY_TAPE_LAUNCH:
#ifndef SY_EMU
                if (two && stop_events_now && t->attach[i] >= 0) {          // the dependency behind this launch rides on it (no marker packet)
                    const Entry& d = t->entries[t->attach[i]];
                    void* const from = stream_of(d.arg >> 4);
                    void* const to = stream_of(d.arg & 15);
                    if (from == cur && from != to) {
                        hipEvent_t ev = t->next_event();
                        if (ev == nullptr) { failed = true; break; }
                        g_stop_event = ev;
                        e.fn(cur);
                        if (g_stop_event != nullptr) {   // the closure did not take it: record the ordinary way
                            g_stop_event = nullptr;
                            if (hipEventRecord(ev, (hipStream_t)cur) != hipSuccess) failed = true;
                        }
                        t->folded[t->attach[i]] = t->ei - 1;
                        break;
                    }
                }
#endif
                e.fn(cur);
                break;
            case SY_TAPE_BREAK:
            case SY_TAPE_BUCKET:
                if (e.kind == SY_TAPE_BUCKET && !stop_buckets) break;
                *pos = i + 1;
                *stop_kind = e.kind;
                *stop_arg = e.arg;
                // index of the stream the cursor REALLY issues on: a chain the caller gave no stream for runs on main (0), and
                // the snippet that runs at this break must be issued there too
                if (stop_stream != nullptr) *stop_stream = (two && cur != main_stream) ? t->cur_k : 0;
                return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
            case SY_TAPE_SIDE:
            case SY_TAPE_FORK:
                if (!two) break;
                wait(side_stream, record(main_stream));
                if (e.kind == SY_TAPE_SIDE) { cur = side_stream; t->cur_k = 1; }
                break;
            case SY_TAPE_SIDE_NW:
                if (!two) break;
                cur = side_stream; t->cur_k = 1;
                break;
            case SY_TAPE_MAIN:
                if (!two) break;
                if (e.arg >= 0) t->slot_done(e.arg, 1, record(side_stream));
                cur = main_stream; t->cur_k = 0;
                break;
            case SY_TAPE_ACQUIRE:
                if (!two) break;
                for (int k = 1; k < kMaxStreams; ++k)
                    if (t->slots[e.arg].ev[k] >= 0 && stream_of(k) != main_stream) wait(main_stream, t->slots[e.arg].ev[k]);
                t->slot_clear(e.arg);
                break;
            case SY_TAPE_JOIN:
                if (!two) break;
                wait(main_stream, record(side_stream));
                break;
            case SY_TAPE_CUR:
                if (!two) break;
                t->cur_k = e.arg;
                cur = stream_of(e.arg);
                break;
            case SY_TAPE_DEP: {
                if (!two) break;
                void* const from = stream_of(e.arg >> 4);
                void* const to = stream_of(e.arg & 15);
                if (from == to) break;
                if (t->folded[i] >= 0) {                 // the producing launch carried the event
                    wait(to, t->folded[i]);
                    t->folded[i] = -1;
                    break;
                }
                wait(to, record(from));
                break;
            }
            case SY_TAPE_SLOT_DONE:
                if (!two) break;
                t->slot_done(e.arg, t->cur_k, record(cur));
                break;
            case SY_TAPE_ACQUIRE_CUR:
                if (!two) break;
                for (int k = 0; k < kMaxStreams; ++k)
                    if (t->slots[e.arg].ev[k] >= 0 && stream_of(k) != cur) wait(cur, t->slots[e.arg].ev[k]);
                t->slots[e.arg].acquired = true;
                break;
            default:
                return SY_ERR_ARG;
        }
    }
    if (failed) return SY_ERR_LAUNCH;
    *pos = n;
    *stop_kind = SY_TAPE_END;
    *stop_arg = 0;
    if (stop_stream != nullptr) *stop_stream = 0;
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

extern "C" int sy_tape_replay(void* tape, void* main_stream, void* side_stream, int* pos, int stop_buckets, int* stop_kind,
                              int* stop_arg, int* stop_on_side) {
    void* const streams[2] = {main_stream, side_stream};
    return sy_tape_replay_n(tape, streams, side_stream != nullptr ? 2 : 1, pos, stop_buckets, stop_kind, stop_arg, stop_on_side);
}
