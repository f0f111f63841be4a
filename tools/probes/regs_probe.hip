// Register / spill census of selected kernel instantiations (development aid):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I streamyolo_amd/csrc -S --cuda-device-only tools/probes/regs_probe.hip -o /tmp/regs.s
#include "../../streamyolo_amd/csrc/conv3x3_halo.h"
namespace sy_conv {
template __global__ void conv3x3_halo_kernel<BF16, 2, 2, 2, 2, 0>(ConvArgs);
template __global__ void conv3x3_halo_kernel<BF16, 4, 1, 2, 4, 0>(ConvArgs);
template __global__ void conv3x3_halo_kernel<BF16, 2, 2, 2, 4, 0>(ConvArgs);
template __global__ void conv3x3_halo_kernel<BF16, 4, 1, 1, 2, 0>(ConvArgs);
template __global__ void conv3x3_halo_kernel<BF16, 1, 4, 2, 2, 0>(ConvArgs);
template __global__ void conv_igemm_kernel<BF16, 4, 1, 1, 2, 6, 1, 0>(ConvArgs);
template __global__ void conv_igemm_kernel<BF16, 4, 1, 1, 2, 5, 1, 0>(ConvArgs);
template __global__ void conv_igemm_kernel<BF16, 2, 2, 2, 2, 5, 1, 0>(ConvArgs);
template __global__ void conv_igemm_kernel<BF16, 2, 2, 1, 1, 6, 1, 0>(ConvArgs);
template __global__ void conv3x3_halo_kernel<BF16, 4, 1, 1, 2, 1>(ConvArgs);
template __global__ void conv3x3_halo_kernel<BF16, 2, 2, 2, 2, 1>(ConvArgs);
template __global__ void conv_igemm_kernel<BF16, 4, 1, 1, 2, 6, 1, 1>(ConvArgs);
template __global__ void conv_igemm_kernel<BF16, 2, 2, 2, 2, 5, 1, 1>(ConvArgs);
template __global__ void conv_igemm_kernel<BF16, 2, 2, 1, 1, 6, 1, 1>(ConvArgs);
}
