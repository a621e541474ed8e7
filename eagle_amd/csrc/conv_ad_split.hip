// A-direct convolution kernels of the split-precision family (see conv_ad_split.inc; own translation unit for build time).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "dmath.h"
#include "conv_internal.h"

namespace eagle {

#include "conv_kernels.inc"
#include "conv_ad_split.inc"

ConvKernel conv_ad_split_kernel48(int n_res)           // Cout = 48: one Cout group, two pixel groups, K split over wave pairs (variant 12)
{
    static const ConvKernel fn[3] = {conv_split_ad_kernel<1, 2, 0, 2>, conv_split_ad_kernel<1, 2, 1, 2>, conv_split_ad_kernel<1, 2, 2, 2>};
    return fn[n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

ConvKernel conv_ad_split_kernel48sb(int n_res)         // Cout = 48: four pixel groups (16 x 32 tile), one halo buffer (variant 13)
{
    static const ConvKernel fn[3] = {conv_split_ad_kernel<1, 4, 0, 1, true>, conv_split_ad_kernel<1, 4, 1, 1, true>, conv_split_ad_kernel<1, 4, 2, 1, true>};
    return fn[n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

ConvKernel conv_ad_split_kernel48ring(int n_res)       // Cout = 48: four pixel groups (16 x 32 tile), two-deep halo ring, one persistent workgroup per CU (variant 19)
{
    static const ConvKernel fn[3] = {conv_split_ad_kernel<1, 4, 0, 1, false>, conv_split_ad_kernel<1, 4, 1, 1, false>, conv_split_ad_kernel<1, 4, 2, 1, false>};
    return fn[n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

ConvKernel conv_ad_split_kernel_s2(bool wide, int n_res)      // stride 2 over the space-to-depth image (variants 10 / 11)
{
    static const ConvKernel fn[2][3] = {
        {conv_split_ad_kernel<2, 2, 0, 1, false, true>, conv_split_ad_kernel<2, 2, 1, 1, false, true>, conv_split_ad_kernel<2, 2, 2, 1, false, true>},
        {conv_split_ad_kernel<4, 1, 0, 1, false, true>, conv_split_ad_kernel<4, 1, 1, 1, false, true>, conv_split_ad_kernel<4, 1, 2, 1, false, true>}};
    return fn[wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

ConvKernel conv_ad_split_kernel_s2t(bool wide, int n_res)      // TRUE stride 2: column-plane halo, stride-1 weight image, one halo buffer (variants 14 / 15)
{
    static const ConvKernel fn[2][3] = {
        {conv_split_ad_kernel<2, 1, 0, 2, true, false, true>, conv_split_ad_kernel<2, 1, 1, 2, true, false, true>, conv_split_ad_kernel<2, 1, 2, 2, true, false, true>},
        {conv_split_ad_kernel<4, 1, 0, 1, true, false, true>, conv_split_ad_kernel<4, 1, 1, 1, true, false, true>, conv_split_ad_kernel<4, 1, 2, 1, true, false, true>}};
    return fn[wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

ConvKernel conv_ad_split_kernel(bool wide, int n_res)
{
    static const ConvKernel fn[2][3] = {
        {conv_split_ad_kernel<2, 2, 0>, conv_split_ad_kernel<2, 2, 1>, conv_split_ad_kernel<2, 2, 2>},
        {conv_split_ad_kernel<4, 1, 0>, conv_split_ad_kernel<4, 1, 1>, conv_split_ad_kernel<4, 1, 2>}};
    return fn[wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

}  // namespace eagle
