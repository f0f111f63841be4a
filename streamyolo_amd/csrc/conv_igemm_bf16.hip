// conv_igemm_bf16.hip — BF16 instantiations of the implicit-GEMM convolution (conv_igemm_impl.h).
#include "conv3x3_halo.h"
#include "conv1x1_stream.h"
#include "conv1x1_tile.h"

int sy_conv_launch_bf16(const sy_conv::ConvArgs& a, void* stream) { return sy_conv::launch_typed<BF16>(a, stream); }
