// conv_extra_bf16.hip — BF16 instantiations of the 3x3 halo kernels (conv3x3_halo.h) and the 1x1 kernel (
// conv1x1_tile.h); reached through sy_conv::launch_typed (conv_igemm_impl.h).
#include "conv3x3_halo.h"
#include "conv3x3_s2dgrad.h"
#include "conv1x1_tile.h"
#include "bottleneck_fused.h"

namespace sy_conv {
template int launch_halo_typed<BF16>(const ConvArgs&, void*);
template int launch_s2dgrad<BF16>(const ConvArgs&, void*);
template int launch_s2dgrad4<BF16>(const ConvArgs&, void*);
template int launch_1x1_tile<BF16>(const ConvArgs&, void*);
template int launch_bottleneck_fused<BF16>(const ConvArgs&, void*);
}  // namespace sy_conv
SY_PROBE_READER(sy_probe_read_conv_extra)
