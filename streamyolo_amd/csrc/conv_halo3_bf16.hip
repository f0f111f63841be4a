// conv_halo3_bf16.hip — BF16 instantiations of the third-generation 3x3 stride-1 kernel (conv3x3_halo3.h, tile codes 98, 100, 101, 109);
// reached through sy_conv::launch_typed (conv_igemm_impl.h).  Its own translation unit: builds beside conv_extra_bf16.hip.
#include "conv3x3_halo3.h"

namespace sy_conv {
template int launch_halo3_typed<BF16>(const ConvArgs&, void*);
}  // namespace sy_conv
SY_PROBE_READER(sy_probe_read_conv_halo3_bf16)
