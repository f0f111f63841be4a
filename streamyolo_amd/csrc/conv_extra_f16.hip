// conv_extra_f16.hip — F16 instantiations of the 3x3 halo kernels (conv3x3_halo.h) and the 1x1 kernel (
// conv1x1_tile.h); reached through sy_conv::launch_typed (conv_igemm_impl.h).
#include "conv3x3_halo.h"
#include "conv3x3_s2dgrad.h"
#include "conv1x1_tile.h"
#include "bottleneck_fused.h"

namespace sy_conv {
template int launch_halo_typed<F16>(const ConvArgs&, void*);
template int launch_s2dgrad<F16>(const ConvArgs&, void*);
template int launch_s2dgrad4<F16>(const ConvArgs&, void*);
template int launch_1x1_tile<F16>(const ConvArgs&, void*);
template int launch_bottleneck_fused<F16>(const ConvArgs&, void*);
}  // namespace sy_conv
