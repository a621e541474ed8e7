// A-direct convolution kernels of the split-precision family on v_mfma_f32_32x32x16_f16 (see conv_ad_split32.inc; own translation unit for build time).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "dmath.h"
#include "conv_internal.h"

namespace eagle {

#include "conv_kernels.inc"
#include "conv_ad_split32.inc"

ConvKernel conv_ad_split32_kernel_w64(bool wide, int n_res)      // the wave's two pixel blocks side by side: BN 192, tile 2 x 64 (variant 23) / BN 96, tile 4 x 64 (variant 24)
{
    static const ConvKernel fn[2][3] = {
        {conv_split_ad32_kernel<1, 4, 0, 3, 2>, conv_split_ad32_kernel<1, 4, 1, 3, 2>, conv_split_ad32_kernel<1, 4, 2, 3, 2>},
        {conv_split_ad32_kernel<2, 2, 0, 3, 2>, conv_split_ad32_kernel<2, 2, 1, 3, 2>, conv_split_ad32_kernel<2, 2, 2, 3, 2>}};
    return fn[wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

ConvKernel conv_ad_split32_kernel(bool wide, int n_res)      // wide: two Cout groups of 96 x two pixel groups (BN 192, tile 4 x 32; variant 21); else one x four (BN 96, tile 8 x 32; variant 22)
{
    static const ConvKernel fn[2][3] = {
        {conv_split_ad32_kernel<1, 4, 0>, conv_split_ad32_kernel<1, 4, 1>, conv_split_ad32_kernel<1, 4, 2>},
        {conv_split_ad32_kernel<2, 2, 0>, conv_split_ad32_kernel<2, 2, 1>, conv_split_ad32_kernel<2, 2, 2>}};
    // developer measurement (EAGLE_CONV_M32_RING=6): the weight ring five steps ahead instead of two, residual-free launches only (72 ring registers)
    static const bool ring6 = getenv("EAGLE_CONV_M32_RING") && atoi(getenv("EAGLE_CONV_M32_RING")) == 6;
    if (ring6 && n_res <= 0) return wide ? (ConvKernel)conv_split_ad32_kernel<2, 2, 0, 6> : (ConvKernel)conv_split_ad32_kernel<1, 4, 0, 6>;
    return fn[wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

}  // namespace eagle
