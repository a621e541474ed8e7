// Convolution kernel instances of the split-precision family (EAGLE_PREC_F32S), part 0 of 3: conv_f16_kernel<..., SPLIT = true> (conv_kernels.inc).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "dmath.h"
#include "conv_internal.h"

namespace eagle {

#include "conv_kernels.inc"

#define IS(KS, S, KC, NT) {EAGLE_PREC_F32S, KS, S, KC, NT, 0, conv_f16_kernel<KS, S, KC, NT, false, 4, true>}
#define ISH(KS, S, KC, NT) {EAGLE_PREC_F32S, KS, S, KC, NT, 3, conv_f16_kernel<KS, S, KC, NT, false, 2, true>}
#define ALLS(KS, S, KC) IS(KS, S, KC, 1), IS(KS, S, KC, 2), IS(KS, S, KC, 3), IS(KS, S, KC, 4), IS(KS, S, KC, 6)
#define ALLSH(KS, S, KC) ISH(KS, S, KC, 1), ISH(KS, S, KC, 2), ISH(KS, S, KC, 3), ISH(KS, S, KC, 4), ISH(KS, S, KC, 6)

static const Inst g_split0[] = {
    // 3x3 stride 1
    ALLS(3, 1, 16), ALLS(3, 1, 32), ALLSH(3, 1, 16), ALLSH(3, 1, 32),
    // 8 x 48 tiles (variant 18, wx = 3: six 16-pixel sub-tiles per wave) for 240-pixel-wide maps: 240 = 5 x 48, 135 = 17 x 8 - 1 -> 0.7 % of the tile area is
    // outside the image instead of 7.4 % with 8 x 32 tiles, and the weight slice is staged once per 384 instead of per 256 pixels
    // (Cout = 64 with NT = 4 / 2 measured: the strips of six sub-tiles leave one workgroup per CU, 16 x 16 tiles stay ahead: 64->64 373-388 us)
    {EAGLE_PREC_F32S, 3, 1, 16, 3, 18, conv_f16_kernel<3, 1, 16, 3, false, 6, true>},
    // (measured and not kept: chunk-pipelined staging, variant 2 of the fp16 family — 48->48 276 vs 273 us, 96->96 227 vs 224 us)
};
const Inst* conv_inst_split0(int* n) { *n = (int)(sizeof(g_split0) / sizeof(g_split0[0])); return g_split0; }

}  // namespace eagle
