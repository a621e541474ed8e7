// Convolution kernel instances, part 3 of 4 (the templates are conv_kernels.inc; split for build time, see conv.hip).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "dmath.h"
#include "conv_internal.h"

namespace eagle {

#include "conv_kernels.inc"

#define I16(KS, S, KC, NT) {EAGLE_PREC_F16, KS, S, KC, NT, 0, conv_f16_kernel<KS, S, KC, NT, false, 4>}
#define I16P(KS, S, KC, NT) {EAGLE_PREC_F16, KS, S, KC, NT, 2, conv_f16_kernel<KS, S, KC, NT, true, 4>}
#define I16H(KS, S, KC, NT) {EAGLE_PREC_F16, KS, S, KC, NT, 3, conv_f16_kernel<KS, S, KC, NT, false, 2>}
#define ALLNT16H(KS, S, KC) I16H(KS, S, KC, 1), I16H(KS, S, KC, 2), I16H(KS, S, KC, 3), I16H(KS, S, KC, 4), I16H(KS, S, KC, 6)
#define I16Q(KS, S, KC, NT) {EAGLE_PREC_F16, KS, S, KC, NT, 4, conv_f16_kernel<KS, S, KC, NT, false, 1>}
#define ALLNT16Q(KS, S, KC) I16Q(KS, S, KC, 1), I16Q(KS, S, KC, 2), I16Q(KS, S, KC, 3), I16Q(KS, S, KC, 4), I16Q(KS, S, KC, 6)
#define ALLNT16P(KS, S, KC) I16P(KS, S, KC, 1), I16P(KS, S, KC, 2), I16P(KS, S, KC, 3), I16P(KS, S, KC, 4), I16P(KS, S, KC, 6)
#define I32(KS, S, KC, NT) {EAGLE_PREC_F32, KS, S, KC, NT, 0, conv_f32_kernel<KS, S, KC, NT>}
#define ALLNT16(KS, S, KC) I16(KS, S, KC, 1), I16(KS, S, KC, 2), I16(KS, S, KC, 3), I16(KS, S, KC, 4), I16(KS, S, KC, 6)
#define ALLNT32(KS, S, KC) I32(KS, S, KC, 1), I32(KS, S, KC, 2), I32(KS, S, KC, 3), I32(KS, S, KC, 4), I32(KS, S, KC, 6)
#define I32H(KS, S, KC, NT) {EAGLE_PREC_F32, KS, S, KC, NT, 3, conv_f32_kernel<KS, S, KC, NT, 2>}
#define ALLNT32H(KS, S, KC) I32H(KS, S, KC, 1), I32H(KS, S, KC, 2), I32H(KS, S, KC, 3), I32H(KS, S, KC, 4), I32H(KS, S, KC, 6)
#define I32Q(KS, S, KC, NT) {EAGLE_PREC_F32, KS, S, KC, NT, 4, conv_f32_kernel<KS, S, KC, NT, 1>}
#define ALLNT32Q(KS, S, KC) I32Q(KS, S, KC, 1), I32Q(KS, S, KC, 2), I32Q(KS, S, KC, 3), I32Q(KS, S, KC, 4), I32Q(KS, S, KC, 6)

static const Inst g_part3[] = {
    // quarter-size tiles (variant 4): 1 pixel sub-tile per wave
    ALLNT16Q(3, 1, 16), ALLNT16Q(3, 1, 32), ALLNT16Q(3, 1, 48), ALLNT16Q(3, 1, 64), ALLNT16Q(3, 2, 16), ALLNT16Q(3, 2, 32), ALLNT16Q(1, 1, 32), ALLNT16Q(1, 1, 64),
    // weight-stationary persistent 3x3 stride-1 kernels (variant 6: 4 pixel sub-tiles per wave, 7: 2); kc = Cin
    {EAGLE_PREC_F16, 3, 1, 48, 3, 6, conv_f16_ws_kernel<48, 3, 4, 2>}, {EAGLE_PREC_F16, 3, 1, 48, 3, 7, conv_f16_ws_kernel<48, 3, 2, 2>},
    {EAGLE_PREC_F16, 3, 1, 64, 4, 6, conv_f16_ws_kernel<64, 4, 4, 1>}, {EAGLE_PREC_F16, 3, 1, 64, 4, 7, conv_f16_ws_kernel<64, 4, 2, 1>},
    {EAGLE_PREC_F16, 3, 1, 96, 3, 6, conv_f16_ws_kernel<96, 3, 4, 1>}, {EAGLE_PREC_F16, 3, 1, 96, 3, 7, conv_f16_ws_kernel<96, 3, 2, 1>},
    {EAGLE_PREC_F16, 3, 1, 96, 2, 7, conv_f16_ws_kernel<96, 2, 2, 1>},
    // exact fp32 family: full tiles (variant 0), half tiles (variant 3: two 16-pixel sub-tiles per wave), quarter tiles (variant 4)
    ALLNT32(3, 1, 16), ALLNT32(3, 2, 16), ALLNT32(3, 2, 4), ALLNT32(1, 1, 16),
    ALLNT32H(3, 1, 16), ALLNT32H(3, 2, 16), ALLNT32H(3, 2, 4), ALLNT32H(1, 1, 16),
    ALLNT32Q(3, 1, 16), ALLNT32Q(3, 2, 16), ALLNT32Q(1, 1, 16),
};
const Inst* conv_inst_part3(int* n) { *n = (int)(sizeof(g_part3) / sizeof(g_part3[0])); return g_part3; }

}  // namespace eagle
