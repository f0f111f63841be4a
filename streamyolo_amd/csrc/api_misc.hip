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
