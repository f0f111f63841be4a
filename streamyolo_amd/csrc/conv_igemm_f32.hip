// conv_igemm_f32.hip — F32 instantiations of the implicit-GEMM convolution (conv_igemm_impl.h).  The halo / 1x1 kernels this
// dispatcher forwards to (tile codes 112..123) are instantiated in conv_extra_f32.hip: two translation units build in parallel.
#include "conv_igemm_impl.h"

int sy_conv_launch_f32(const sy_conv::ConvArgs& a, void* stream) { return sy_conv::launch_typed<F32>(a, stream); }
