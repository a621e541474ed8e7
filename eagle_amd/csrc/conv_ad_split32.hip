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

// deep (round 6): the weight ring five steps ahead instead of two (RING = 6: 72 ring registers).  For launches with at most one workgroup per CU — the small-batch
// regime, where no co-resident workgroup hides the L2 round trip of the A fragments and an item's time is that latency chain: one-frame call 9.2 -> 8.4 ms with the
// residual-free launches alone (profiles/r06v_latency_ring6_ab.log).  Same arithmetic in the same order: records stay byte-identical across batch sizes.  At B = 50 the
// two workgroups of a CU cover each other and the deeper ring measured nothing (round 5), so full launches keep RING = 3.  EAGLE_CONV_M32_RING=3 / 6 forces one.
static bool ring_deep(bool deep)
{
    static const int forced = getenv("EAGLE_CONV_M32_RING") ? atoi(getenv("EAGLE_CONV_M32_RING")) : 0;
    return forced == 6 ? true : forced == 3 ? false : deep;
}

ConvKernel conv_ad_split32_kernel_w64(bool wide, int n_res, bool deep)      // the wave's two pixel blocks side by side: BN 192, tile 2 x 64 (variant 23) / BN 96, tile 4 x 64 (variant 24)
{
    static const ConvKernel fn[2][2][3] = {
        {{conv_split_ad32_kernel<1, 4, 0, 3, 2>, conv_split_ad32_kernel<1, 4, 1, 3, 2>, conv_split_ad32_kernel<1, 4, 2, 3, 2>},
         {conv_split_ad32_kernel<2, 2, 0, 3, 2>, conv_split_ad32_kernel<2, 2, 1, 3, 2>, conv_split_ad32_kernel<2, 2, 2, 3, 2>}},
        {{conv_split_ad32_kernel<1, 4, 0, 6, 2>, conv_split_ad32_kernel<1, 4, 1, 6, 2>, conv_split_ad32_kernel<1, 4, 2, 6, 2>},
         {conv_split_ad32_kernel<2, 2, 0, 6, 2>, conv_split_ad32_kernel<2, 2, 1, 6, 2>, conv_split_ad32_kernel<2, 2, 2, 6, 2>}}};
    return fn[ring_deep(deep) ? 1 : 0][wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

ConvKernel conv_ad_split32_kernel(bool wide, int n_res, bool deep)      // wide: two Cout groups of 96 x two pixel groups (BN 192, tile 4 x 32; variant 21); else one x four (BN 96, tile 8 x 32; variant 22)
{
    static const ConvKernel fn[2][2][3] = {
        {{conv_split_ad32_kernel<1, 4, 0>, conv_split_ad32_kernel<1, 4, 1>, conv_split_ad32_kernel<1, 4, 2>},
         {conv_split_ad32_kernel<2, 2, 0>, conv_split_ad32_kernel<2, 2, 1>, conv_split_ad32_kernel<2, 2, 2>}},
        {{conv_split_ad32_kernel<1, 4, 0, 6>, conv_split_ad32_kernel<1, 4, 1, 6>, conv_split_ad32_kernel<1, 4, 2, 6>},
         {conv_split_ad32_kernel<2, 2, 0, 6>, conv_split_ad32_kernel<2, 2, 1, 6>, conv_split_ad32_kernel<2, 2, 2, 6>}}};
    return fn[ring_deep(deep) ? 1 : 0][wide ? 1 : 0][n_res < 0 ? 0 : n_res > 2 ? 2 : n_res];
}

}  // namespace eagle
