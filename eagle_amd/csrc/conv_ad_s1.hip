// A-direct convolution kernels, stride 1 instances (see conv_ad_kernel.inc; split from conv.hip for build time).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "conv_internal.h"

namespace eagle {

#include "conv_ad_kernel.inc"

ConvKernel conv_ad_kernel_s1(bool wide, int n_res)
{
    static const ConvKernel fn[2][3] = {
        {conv_f16_ad_kernel<2, 2, 0, false>, conv_f16_ad_kernel<2, 2, 1, false>, conv_f16_ad_kernel<2, 2, 2, false>},
        {conv_f16_ad_kernel<4, 1, 0, false>, conv_f16_ad_kernel<4, 1, 1, false>, conv_f16_ad_kernel<4, 1, 2, false>}};
    return fn[wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

}  // namespace eagle
