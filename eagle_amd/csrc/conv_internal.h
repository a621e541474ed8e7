// Shared between conv.hip and the A-direct translation units (conv_ad_s1.hip / conv_ad_s2.hip include conv_ad_kernel.inc): splitting the
// instantiations over three files keeps a clean parallel build at about a third of the single-file compile time.
#pragma once
#include "common.h"

namespace eagle {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

struct ConvArgs {
    const void* x; int xcs, xoff; int N, H, W;
    const void* w; const float* bias;
    void* y; int ycs, yoff; int Ho, Wo;
    const void* r1; int r1cs, r1off;
    const void* r2; int r2cs, r2off;
    int pre_act, post_act, out_f32;
    int wx, tiles_x, tiles_y, nchunks;
    int gy;                 // number of Cout blocks
    int stack;              // EAGLE_PREC_F32, stride 1: the batch tiled as one image of N (H + 1) rows (conv_f32_kernel); tiles_y then counts the whole batch
    int xcd;                // 1: XCD-aware work order (1-D grid; each XCD owns a contiguous range of (tile, Cout-block) items)
    ArgmaxPart* am;         // fused heat-map maxima (head convolution): partials [frame][tile][am_cs] instead of the output tensor
    int am_cs;
    float descale;          // EAGLE_PREC_F32S: 2^-(weight scale + 4), undoes the power-of-two operand scaling of the accumulator
    const void* zeros;      // >= 16 zero bytes in global memory (source of out-of-image pixels for unconditional loads / LDS-DMA)
    void* trash;            // >= 4 KiB of scratch global memory (target of out-of-image results for unconditional stores)
    unsigned* sat;          // EAGLE_PREC_F32S: per-frame saturation counters [N] (a lane that stores a value beyond the split format's range of
                            // +-4094 adds one to its frame's word), or nullptr: not counted (operator-level calls)
};

struct BneckArgs {          // bneck.hip; channel strides / offsets in fp16 elements (two per logical channel)
    const void* x; int xcs, xoff; int N, H, W, nch1;      // nch1 = Cin / 16
    const void *w1, *w2, *w3; const float *b1, *b2, *b3; float ds1, ds2, ds3;
    const void* r; int rcs, roff;
    void* y; int ycs, yoff;
    int tiles_x, tiles_y;
    unsigned* sat;
    unsigned long long* dbg;   // developer timing builds (-DEAGLE_BNECK_TIMING): per workgroup 8 accumulated s_memrealtime spans (100 MHz ticks), else unused
};

typedef void (*ConvKernel)(ConvArgs);
struct Inst { int prec, ks, s, kc, nt, variant; ConvKernel fn; };
const Inst* conv_inst_part(int part, int* n);       // instance tables of conv_inst_0..3.hip (declared per part below)
const Inst* conv_inst_part0(int* n); const Inst* conv_inst_part1(int* n); const Inst* conv_inst_part2(int* n); const Inst* conv_inst_part3(int* n);
const Inst* conv_inst_split0(int* n); const Inst* conv_inst_split1(int* n); const Inst* conv_inst_split2(int* n);      // EAGLE_PREC_F32S instances (conv_inst_s0..2.hip)
// the A-direct kernel instance for (cout_groups x pixel_groups, number of residual operands): wide = 4 x 1 (BN 192), otherwise 2 x 2 (BN 96)
ConvKernel conv_ad_kernel_s1(bool wide, int n_res);
ConvKernel conv_ad_kernel_s2(bool wide, int n_res);
ConvKernel conv_ad_split_kernel(bool wide, int n_res);      // EAGLE_PREC_F32S form (conv_ad_split.inc), stride 1
ConvKernel conv_ad_split32_kernel_w64(bool wide, int n_res, bool deep); // ... with the wave's two pixel blocks side by side: BN = 192, tile 2 x 64 (variant 23) / BN = 96, tile 4 x 64 (variant 24)
ConvKernel conv_ad_split32_kernel(bool wide, int n_res, bool deep);    // the 32x32x16 form (conv_ad_split32.inc): BN = 192, tile 4 x 32 (variant 21) / BN = 96, tile 8 x 32 (variant 22)
ConvKernel conv_ad_split_kernel48(int n_res);               // the same for Cout = 48 (K split over wave pairs; variant 12)
ConvKernel conv_ad_split_kernel_s2(bool wide, int n_res);    // stride 2 (variants 10 / 11 of the split family)
ConvKernel conv_ad_split_kernel_s2t(bool wide, int n_res);   // TRUE stride 2 on a column-plane halo, single halo buffer: BN = 192 (variant 14) / BN = 96 with the K split (variant 15)
ConvKernel conv_ad_split_kernel48sb(int n_res);             // Cout = 48, 16 x 32 tile, single halo buffer (variant 13)
ConvKernel conv_ad_split_kernel48ring(int n_res);           // Cout = 48, 16 x 32 tile, two-deep halo ring, one persistent workgroup per CU (variant 19)

}  // namespace eagle
