// Convolution kernel instances, part 0 of 4 (the templates are conv_kernels.inc; split for build time, see conv.hip).
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

static const Inst g_part0[] = {
    // 3x3 stride 1: register-staged (variant 0) and chunk-pipelined staging (variant 2)
    ALLNT16(3, 1, 16), ALLNT16(3, 1, 32), ALLNT16(3, 1, 48), ALLNT16(3, 1, 64), ALLNT16P(3, 1, 16), ALLNT16P(3, 1, 32),
};
const Inst* conv_inst_part0(int* n) { *n = (int)(sizeof(g_part0) / sizeof(g_part0[0])); return g_part0; }

}  // namespace eagle
