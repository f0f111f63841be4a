// sy_tape.h — native launch tapes: the recording hook every SY_LAUNCH goes through (host side only).
//
// A plan's step is a FIXED list of kernel launches: same kernels, same grids, same by-value arguments every iteration
// (the tensors live in plan-owned buffers).  While a recording is open on the calling thread (sy_tape_begin), every
// SY_LAUNCH of every entry point is executed as usual AND its closure — kernel, grid, block, dynamic LDS, the argument
// values — is appended to the tape; sy_tape_replay then re-issues the whole list from C on explicit hipStream_t handles,
// switching streams and recording / waiting hipEvents where the plan put its marks.  A training step becomes a handful
// of ctypes calls instead of ~1200 (the Python-side tape cost ~12 us of host time per launch: streamyolo_amd/_lib.py).
#pragma once
#include <functional>

// non-zero while a recording is open on this thread
bool sy_tape_recording();
// append one launch closure (called with the stream to launch on) to the open recording
void sy_tape_push(std::function<void(void*)>&& fn);
// Round 6: an event record on a frame chain is a marker packet its queue stops at (~3 us each, ~250 per l step: a replay with one
// EXTRA record per dependency ran 0.36 ms longer, profiles/r06 stage A).  When the tape replay knows that a cross-stream dependency
// follows a launch directly, it hands that launch the event instead (this thread's pending "stop event"): SY_LAUNCH then issues the
// kernel through hipExtLaunchKernelGGL, whose stop event is the dispatch packet's own completion signal — no marker packet.
// Returns the pending event (a hipEvent_t) and clears it; nullptr: a plain launch.
void* sy_tape_stop_event_take();
