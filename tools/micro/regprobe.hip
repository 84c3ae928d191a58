// register-budget probe: instantiates a handful of stencil kernels alone (seconds instead of the minute hopping.hip takes), e.g.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -I../../tmlqcd_amd/csrc -c regprobe.hip -o /dev/null 2> ru.txt
//   python3 ../check_resources.py ru.txt
#include "hopping_common.h"
namespace hop64 {
TMHIP_SCALAR_COMPLEX_OPS(v2d, double)
TMHIP_SPINOR_IO_PLANES
#define HOP_SITES 1
#define HOP_KERNELS_ONLY
#include "hopping_impl.inc"
#ifndef PROBE_EPI
#define PROBE_EPI 0
#endif
template __global__ void hop_kernel<PROBE_EPI, 0, true, 256, 3, -1, 64>(const HopArgs);
template __global__ void hop_kernel<PROBE_EPI, 1, true, 256, 3, -1, 64>(const HopArgs);
template __global__ void hop_kernel<PROBE_EPI, 3, true, 256, 3, -1, 64>(const HopArgs);
template __global__ void hop_kernel<PROBE_EPI, 1, true, 256, 3, -1, 0>(const HopArgs);
template __global__ void hop_kernel<PROBE_EPI, 3, true, 256, 3, -1, 0>(const HopArgs);
template __global__ void hop_kernel<PROBE_EPI, 1, true, 64, 3, -1, 0>(const HopArgs);
template __global__ void hop_kernel<PROBE_EPI, 3, true, 64, 3, -1, 0>(const HopArgs);
}  // namespace hop64
