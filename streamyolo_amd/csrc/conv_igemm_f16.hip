// conv_igemm_f16.hip — F16 instantiations of the implicit-GEMM convolution (conv_igemm_impl.h).
#include "conv3x3_halo.h"
#include "conv1x1_stream.h"
#include "conv1x1_tile.h"

int sy_conv_launch_f16(const sy_conv::ConvArgs& a, void* stream) { return sy_conv::launch_typed<F16>(a, stream); }
