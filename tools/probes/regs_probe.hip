// Register / spill census of selected kernel instantiations (development aid): tools/regs_census.sh
#include "../../streamyolo_amd/csrc/conv3x3_halo.h"
#include "../../streamyolo_amd/csrc/conv1x1_tile.h"
namespace sy_conv {
template __global__ void conv3x3_halo_kernel<BF16, 4, 1, 1, 2>(ConvArgs);
template __global__ void conv3x3_halo2_kernel<BF16, 4, 1, 1, 2>(ConvArgs);
template __global__ void conv3x3_halo2_kernel<BF16, 4, 1, 1, 4>(ConvArgs);
template __global__ void conv3x3_halo2_kernel<BF16, 2, 2, 2, 2>(ConvArgs);
template __global__ void conv1x1_tile_kernel<BF16, 4, 1, 1, 2, 8>(ConvArgs);

template __global__ void conv1x1_tile_kernel<BF16, 4, 1, 1, 2, 16>(ConvArgs);
template __global__ void conv1x1_tile_kernel<BF16, 4, 1, 1, 4, 8>(ConvArgs);
template __global__ void conv1x1_tile_kernel<BF16, 2, 2, 1, 2, 4>(ConvArgs);
}
