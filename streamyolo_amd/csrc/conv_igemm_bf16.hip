// conv_igemm_bf16.hip — BF16 instantiations of the implicit-GEMM convolution (conv_igemm_impl.h).  The halo / 1x1 kernels this
// dispatcher forwards to (tile codes 112..123) are instantiated in conv_extra_bf16.hip: two translation units build in parallel.
#include "conv_igemm_impl.h"

int sy_conv_launch_bf16(const sy_conv::ConvArgs& a, void* stream) { return sy_conv::launch_typed<BF16>(a, stream); }
SY_PROBE_READER(sy_probe_read_conv_igemm)
