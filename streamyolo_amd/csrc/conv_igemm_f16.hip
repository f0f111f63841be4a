// conv_igemm_f16.hip — F16 instantiations of the implicit-GEMM convolution (conv_igemm_impl.h).  The halo / 1x1 kernels this
// dispatcher forwards to (tile codes 112..123) are instantiated in conv_extra_f16.hip: two translation units build in parallel.
#include "conv_igemm_impl.h"

int sy_conv_launch_f16(const sy_conv::ConvArgs& a, void* stream) { return sy_conv::launch_typed<F16>(a, stream); }
