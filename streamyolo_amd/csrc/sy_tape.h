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
