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
    // (measured and not kept: chunk-pipelined staging, variant 2 of the fp16 family — 48->48 276 vs 273 us, 96->96 227 vs 224 us)
};
const Inst* conv_inst_split0(int* n) { *n = (int)(sizeof(g_split0) / sizeof(g_split0[0])); return g_split0; }

}  // namespace eagle
