// Implicit-GEMM convolution for gfx950 (CDNA4), NHWC, BatchNorm pre-folded, fused epilogue.
//
// Replaces nn.Conv2d + BatchNorm2d + ReLU (+ residual) of the reference's HRNet (eagle/models/keypoint_hrnet.py:65-137,
// 215-278, 353-391, 553-558) and ultralytics' Conv/Bottleneck.
//
// GEMM view:  D[Cout x pixels] = W[Cout x K] * X[K x pixels],  K = ks*ks*Cin.
//   * weights are the MFMA "A" operand, activations the "B" operand, so each lane ends up holding 4 CONSECUTIVE
//     output channels of one pixel -> 8-byte (fp16) / 16-byte (fp32) NHWC stores and residual loads.
//   * a workgroup (4 waves) owns a (16/wx) x (16*wx) output-pixel tile and BN = 16*NT output channels; each wave
//     owns 4 sub-tiles of 16 consecutive pixels.  The input halo tile and the weight slice of one Cin-chunk (KC
//     channels) are staged in LDS; pixel stride in LDS is padded so ds_read_b128 over 16 pixels is conflict-free.
//   * fp16 family: v_mfma_f32_16x16x32_f16, K flattened over (tap, 8-channel group); fp32 accumulate.
//   * fp32 family: v_mfma_f32_16x16x4_f32 in the canonical K order (16-channel chunk, tap, channel): gfx950
//     accumulates these as a k-ordered fmaf chain, so outputs are bit-identical to oracle/eo_prims.c.
// Epilogue: v = acc + bias; v = pre(v); v = r1 + v; v = v + r2; v = post(v); store (fp16 RNE / fp32).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "dmath.h"
#include "conv_internal.h"

namespace eagle {

// ------------------------------------------------------------------------------------------------------------
// host side: geometry, instance table, weight tiling
// ------------------------------------------------------------------------------------------------------------
static int f16_ps(int kc) { const int g = kc / 8; return kc * 2 + ((g % 2 == 0) ? 16 : 0); }
static int f16_ni(int ks, int kc) { return (ks * ks * (kc / 8) + 3) / 4; }

static int conv_pw(const ConvConfig& c) { return (c.variant == 3 || c.variant == 7) ? 2 : c.variant == 4 ? 1 : c.variant == 18 ? 6 : 4; }
static bool conv_ws(const ConvConfig& c) { return c.variant == 6 || c.variant == 7; }
static bool conv_ad_m32(const ConvConfig& c) { return c.variant >= 21 && c.variant <= 24; }      // split family on v_mfma_f32_32x32x16_f16 (conv_ad_split32.inc): 21 = BN 192, tile 4 x 32; 22 = BN 96, tile 8 x 32; 23 = BN 192, tile 2 x 64; 24 = BN 96, tile 4 x 64
static int conv_ad_tw(const ConvConfig& c) { return (c.variant == 23 || c.variant == 24) ? 64 : 32; }      // output columns of an A-direct tile
static bool conv_ad(const ConvConfig& c) { return (c.variant >= 8 && c.variant <= 15) || c.variant == 19 || conv_ad_m32(c); }
static bool conv_ad_s2t(const ConvConfig& c) { return c.variant == 14 || c.variant == 15; }      // TRUE stride 2 on a column-plane halo (14: BN 192; 15: BN 96, K split over wave pairs)
static bool conv_ad_s2d(const ConvConfig& c) { return c.variant == 10 || c.variant == 11; }      // stride 2 over the space-to-depth image (variants 14 / 15: true stride 2, the stride-1 weight image)
static int conv_ad_rows(const ConvConfig& c) { return c.variant == 23 ? 2 : (c.variant == 8 || c.variant == 10 || c.variant == 21 || c.variant == 24 || conv_ad_s2t(c)) ? 4 : (c.variant == 13 || c.variant == 19) ? 16 : 8; }      // output rows of an A-direct tile      // A-direct: 8 = 4 Cout groups x 1 pixel group (BN 192), 9 = 2 x 2 (BN 96); 10 / 11 = the same for stride 2
static bool conv_ad_wide(const ConvConfig& c) { return c.variant == 8 || c.variant == 10 || c.variant == 14 || c.variant == 21 || c.variant == 23; }

static size_t lds_bytes(int precision, const ConvConfig& c)
{
    const int th = 4 * conv_pw(c) / c.wx, tw = 16 * c.wx;
    const int hw = (tw - 1) * c.stride + c.ks, hh = (th - 1) * c.stride + c.ks;
    const int bn = c.nt * 16;
    if (precision == EAGLE_PREC_F16 && conv_ad(c)) {
        const int pgn = conv_ad_wide(c) ? 1 : 2, slabs = (((c.stride == 2 ? (4 * pgn + 1) * 33 : (4 * pgn + 2) * 34) * 96 + 1023) / 1024 + 3) / 4 * 4;
        return (size_t)2 * slabs * 1024 + 4 * 32 * 112;
    }
    if (precision == EAGLE_PREC_F16) {
        const size_t operands = (size_t)f16_ni(c.ks, c.kc) * 4 * bn * 16 + (size_t)hh * hw * f16_ps(c.kc);
        const size_t strips = (size_t)4 * conv_pw(c) * 16 * (bn * 2 + 16);        // output transpose, one strip per wave
        return conv_ws(c) ? operands + strips + 16 : std::max(operands + 16, strips);
    }
    if (precision == EAGLE_PREC_F32S && conv_ad_m32(c)) {  // 80-byte halo records, two-deep ring, strips of 32 pixels x 32 channels x 4 bytes
        const int slabs = (((conv_ad_rows(c) + 2) * (conv_ad_tw(c) + 2) * 80 + 1023) / 1024 + 3) / 4 * 4;
        return (size_t)2 * slabs * 1024 + 4 * 32 * 128;       // strips: 128-byte records, units swizzled by the pixel
    }
    if (precision == EAGLE_PREC_F32S && conv_ad(c)) {      // same halo ring as the fp16 kernel (16 logical channels = the 96-byte record), strips of 16 pixels x 48 channels x 4 bytes
        const int hpix = conv_ad_s2t(c) ? (2 * conv_ad_rows(c) + 1) * 66 : (conv_ad_rows(c) + 2) * 34;      // variants 14 / 15: 9 rows x (33 even + 33 odd columns)
        const int slabs = ((hpix * 96 + 1023) / 1024 + 3) / 4 * 4;
        return (size_t)((c.variant == 13 || conv_ad_s2t(c)) ? 1 : 2) * slabs * 1024 + 4 * 16 * 208;       // variants 13 - 15: one halo buffer
    }
    if (precision == EAGLE_PREC_F32S) {                    // hi and lo fragment blocks per K-step; 2 * kc fp16 values per staged pixel; 4-byte outputs
        const size_t operands = (size_t)f16_ni(c.ks, c.kc) * 2 * 4 * bn * 16 + (size_t)hh * hw * f16_ps(2 * c.kc);
        const size_t strips = (size_t)4 * conv_pw(c) * 16 * (bn * 4 + 16);
        return std::max(operands + 16, strips);
    }
    return (size_t)c.ks * c.ks * (c.kc / 4) * 4 * bn * 4 + (size_t)hh * hw * (c.kc + 1) * 4;
}

size_t conv_lds_bytes(int precision, const ConvConfig& c) { return lds_bytes(precision, c); }
int conv_tiles_per_frame(const ConvConfig& c, int ho, int wo)
{
    if (conv_ad(c)) return ((wo + conv_ad_tw(c) - 1) / conv_ad_tw(c)) * ((ho + conv_ad_rows(c) - 1) / conv_ad_rows(c));
    const int th = 4 * conv_pw(c) / c.wx, tw = 16 * c.wx;
    return ((wo + tw - 1) / tw) * ((ho + th - 1) / th);
}

// The kernel instances live in four translation units (conv_inst_0..3.hip) so that a clean build compiles them in parallel; the A-direct
// kernels in conv_ad_s1.hip / conv_ad_s2.hip.
static const Inst g_ad_inst[] = {
    // A-direct 3x3 kernels (variant 8: BN = 192, tile 4 x 32; 9: BN = 96, tile 8 x 32; 10 / 11: the same for stride 2); kc = 32
    {EAGLE_PREC_F16, 3, 1, 32, 12, 8, nullptr}, {EAGLE_PREC_F16, 3, 1, 32, 6, 9, nullptr}, {EAGLE_PREC_F16, 3, 2, 32, 12, 10, nullptr}, {EAGLE_PREC_F16, 3, 2, 32, 6, 11, nullptr},
    // split family: chunks of 16 logical channels (conv_ad_split.inc)
    {EAGLE_PREC_F32S, 3, 1, 16, 12, 8, nullptr}, {EAGLE_PREC_F32S, 3, 1, 16, 6, 9, nullptr},
    // split family, Cout = 48 per workgroup: tile 8 x 32, the K dimension split over wave pairs (variant 12)
    {EAGLE_PREC_F32S, 3, 1, 16, 3, 12, nullptr},
    // the same Cout with four pixel groups (16 x 32 tile) and a single halo buffer (variant 13)
    {EAGLE_PREC_F32S, 3, 1, 16, 3, 13, nullptr},
    // ... and with the two-deep halo ring: 130 KB of LDS, ONE persistent workgroup per CU whose ring runs on across its items (variant 19, round 4)
    {EAGLE_PREC_F32S, 3, 1, 16, 3, 19, nullptr},
    // split family, stride 2 over the space-to-depth image (variants 10 / 11)
    {EAGLE_PREC_F32S, 3, 2, 16, 12, 10, nullptr}, {EAGLE_PREC_F32S, 3, 2, 16, 6, 11, nullptr},
    // split family, TRUE stride 2 on an even / odd column-plane halo, the stride-1 weight image (variant 14: BN = 192, tile 4 x 32, one halo buffer)
    {EAGLE_PREC_F32S, 3, 2, 16, 12, 14, nullptr},
    // the same with BN = 96: two Cout groups, the K dimension split over wave pairs (variant 15)
    {EAGLE_PREC_F32S, 3, 2, 16, 6, 15, nullptr},
    // split family, stride 1, on v_mfma_f32_32x32x16_f16 (round 5, conv_ad_split32.inc): BN = 192 (variant 21) / BN = 96 (variant 22)
    {EAGLE_PREC_F32S, 3, 1, 16, 12, 21, nullptr}, {EAGLE_PREC_F32S, 3, 1, 16, 6, 22, nullptr},
    // ... with the wave's two pixel blocks side by side (tiles 2 x 64 / 4 x 64: 34- and 68-row maps without a row remainder): variants 23 / 24
    {EAGLE_PREC_F32S, 3, 1, 16, 12, 23, nullptr}, {EAGLE_PREC_F32S, 3, 1, 16, 6, 24, nullptr}};

const Inst* conv_inst_part(int part, int* n)
{
    return part == 0 ? conv_inst_part0(n) : part == 1 ? conv_inst_part1(n) : part == 2 ? conv_inst_part2(n) : conv_inst_part3(n);
}

static const Inst* find_inst(int precision, const ConvConfig& c)
{
    auto match = [&](const Inst& i) { return i.prec == precision && i.ks == c.ks && i.s == c.stride && i.kc == c.kc && i.nt == c.nt && i.variant == c.variant; };
    for (const Inst& i : g_ad_inst)
        if (match(i)) return &i;
    for (int part = 0; part < 7; ++part) {
        int n = 0;
        const Inst* t = part < 4 ? conv_inst_part(part, &n) : part == 4 ? conv_inst_split0(&n) : part == 5 ? conv_inst_split1(&n) : conv_inst_split2(&n);
        for (int k = 0; k < n; ++k)
            if (match(t[k])) return &t[k];
    }
    return nullptr;
}
bool conv_supported(int precision, const ConvConfig& c) { return find_inst(precision, c) != nullptr; }

// Default form of the split family's 3x3 stride-1 layers with Cout = 96 k (EAGLE_CONV_M32 overrides; 0 = the 16x16x32 A-direct forms of rounds 3 / 4).
// 6 since round 5: the 32x32x16 kernels, 4 x 32 tiles for BN = 192 and 4 x 64 tiles for BN = 96 on maps wider than 32 columns — same box, three alternating
// pairs through the whole pipeline: 750.0 / 751.0 / 753.4 -> 761.3 / 761.5 / 761.8 frames/s with the tile per map (mode 4; +1.4 %), 96->96 @68x120 208.5 -> 193.5 us.
#ifndef EAGLE_CONV_M32_DEFAULT
#define EAGLE_CONV_M32_DEFAULT 6
#endif
struct Tuned { int ks, s, cin, cout, wo, kc, nt, wx, variant; };
static const Tuned g_tuned[] = {
#include "conv_tuned.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0}};
static const Tuned g_tuned_f32[] = {         // EAGLE_PREC_F32: tile shape only (nt, wx); the summation order, hence every bit, is the same for all rows
#include "conv_tuned_f32.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0}};
static const Tuned g_tuned_split[] = {       // EAGLE_PREC_F32S (tools/autotune_split.py); rows of one shape are ordered best first
#include "conv_tuned_split.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0}};

ConvConfig conv_choose(int precision, int ks, int stride, int cin_pad, int cout_pad, int wo, bool plain_epilogue, bool second_residual, bool any_residual)
{
    // plain_epilogue: no activation before the residual adds, none / ReLU after them, fp16 output.  The weight-stationary kernels also need
    // at most one residual; the A-direct kernels take two.
    const bool plain_one = plain_epilogue && !second_residual;
    ConvConfig c;
    c.ks = ks; c.stride = stride; c.cin = cin_pad; c.cout_pad = cout_pad;
    c.wx = (wo > 16) ? 2 : 1;
    static const int nts[] = {6, 4, 3, 2, 1};
    if (precision == EAGLE_PREC_F32) {
        if (const char* f = getenv("EAGLE_F32_FORCE")) {    // "nt,wx,variant": parity tests of the tilings (every tiling gives the same bits)
            ConvConfig q = c; q.kc = (cin_pad < 16) ? 4 : 16;
            if (sscanf(f, "%d,%d,%d", &q.nt, &q.wx, &q.variant) == 3 && (q.wx == 1 || q.wx == 2) && q.nt >= 1 && cout_pad % (16 * q.nt) == 0 && find_inst(precision, q) && lds_bytes(precision, q) <= 160 * 1024) return q;
        }
        static const bool tuned32 = !(getenv("EAGLE_CONV_TUNED") && atoi(getenv("EAGLE_CONV_TUNED")) == 0);
        if (tuned32)
            for (const Tuned& t : g_tuned_f32)
                if (t.ks == ks && t.s == stride && t.cin == cin_pad && t.cout == cout_pad && t.wo == wo) {
                    ConvConfig q = c; q.kc = t.kc; q.nt = t.nt; q.wx = t.wx; q.variant = t.variant;      // variant: 0 full, 3 half, 4 quarter tiles
                    if (find_inst(precision, q)) return q;
                }
        c.nt = 1;
        for (int nt : nts)
            if (cout_pad % (16 * nt) == 0) { c.nt = nt; break; }
        c.kc = (cin_pad < 16) ? 4 : 16;
        if (stride == 2 && ks == 3) {                       // the halo of a stride-2 tile is four times the tile: half tiles keep two workgroups on a CU
            ConvConfig q = c; q.variant = 3;
            if (find_inst(precision, q)) return q;
        }
        return c;
    }
    if (precision == EAGLE_PREC_F32S) {
        // split family: generic kernel, full (variant 0) or half (variant 3) tiles; the (NT, KC) pair with the most MFMA work per staged byte
        // whose LDS footprint lets two workgroups share a CU.  EAGLE_CONV_FORCE applies as below.
        if (const char* f = getenv("EAGLE_CONV_FORCE")) {
            ConvConfig q = c;
            if (sscanf(f, "%d,%d,%d", &q.kc, &q.nt, &q.variant) == 3 && cin_pad % q.kc == 0 && cout_pad % (16 * q.nt) == 0 && find_inst(precision, q)) {
                if (q.variant == 18) q.wx = 3;              // the 8 x 48 tile
                if (lds_bytes(precision, q) <= 160 * 1024) return q;
            }
        }
        static const bool sad_on = !(getenv("EAGLE_CONV_AD") && atoi(getenv("EAGLE_CONV_AD")) == 0);
        static const bool tuned_on = !(getenv("EAGLE_CONV_TUNED") && atoi(getenv("EAGLE_CONV_TUNED")) == 0);
        // (measured on MI355X: 48->48@135x240 305 / 551 us without / with residual against 273 / 322 us of the generic kernel — an 8 x 32 x 48 item is
        //  21 K-steps per wave, too little work against the exchange, the epilogue and three barriers; kept for the tuner, off by default)
        const bool kq_on = getenv("EAGLE_CONV_KQ") && atoi(getenv("EAGLE_CONV_KQ")) != 0;      // (read per call: the parity test switches it on)
        // EAGLE_CONV_48NR=1 (measured, round 4): the Cout = 48 layers WITHOUT a residual operand (conv1 of every BasicBlock of HRNet's widest branch) on the
        // 16 x 32 single-buffer A-direct form (variant 13), which is 6 % ahead of the 8 x 48 generic tile on that case in isolation (251 vs 268 us)
        const bool nr48 = getenv("EAGLE_CONV_48NR") && atoi(getenv("EAGLE_CONV_48NR")) != 0;
        if (sad_on && nr48 && !any_residual && plain_epilogue && ks == 3 && stride == 1 && cin_pad == 48 && cout_pad == 48 && wo > 64) {
            ConvConfig q = c; q.kc = 16; q.nt = 3; q.variant = 13;
            return q;
        }
        if (sad_on && kq_on && plain_epilogue && ks == 3 && stride == 1 && cin_pad % 48 == 0 && cout_pad % 48 == 0 && cout_pad % 96 != 0) {      // Cout = 48 (144, ...): K split over wave pairs
            ConvConfig q = c; q.kc = 16; q.nt = 3; q.variant = atoi(getenv("EAGLE_CONV_KQ")) == 13 ? 13 : atoi(getenv("EAGLE_CONV_KQ")) == 19 ? 19 : 12;
            return q;
        }
        // 3x3 stride 2 with Cin = 48 k, Cout = 96 k (HRNet's transition / fuse down-sampling chains): the TRUE stride-2 A-direct forms.  Same box, all
        // instances per layer (tools/convbench/split_tune, B = 50, best other form -> this one): 96->192 145 -> 108 us, 48->192 80 -> 62, 192->384
        // 139 -> 102, 96->384 76 -> 55, 48->384 45 -> 33 (variant 14); 48->96 203 -> 175, 96->96 88 -> 70 (variant 15).  EAGLE_CONV_S2T=0: off.
        static const bool s2t_on = !(getenv("EAGLE_CONV_S2T") && atoi(getenv("EAGLE_CONV_S2T")) == 0);
        if (sad_on && s2t_on && plain_epilogue && ks == 3 && stride == 2 && cin_pad % 48 == 0 && cout_pad % 96 == 0) {
            ConvConfig q = c; q.kc = 16;
            if (cout_pad % 192 == 0) { q.nt = 12; q.variant = 14; } else { q.nt = 6; q.variant = 15; }
            return q;
        }
        // round 5: the 32x32x16 form of the A-direct kernel (variants 21 - 24) for the 3x3 stride-1 layers with Cout = 96 k; any Cin = 16 k.  EAGLE_CONV_M32: 0 off,
        // 1 the 4 x 32 / 8 x 32 tiles (variants 21 / 22), 2 only BN = 192 (variant 21), 3 only BN = 96 (variant 22), 4: the tile per map — 2 x 64 / 4 x 64 (variants 23 / 24)
        // where the map is wider than 32 columns and its rows then divide without a larger remainder (34 x 60, 68 x 120), otherwise 21 / 22
        const int m32 = getenv("EAGLE_CONV_M32") ? atoi(getenv("EAGLE_CONV_M32")) : EAGLE_CONV_M32_DEFAULT;      // (read per call: the parity tests switch it)
        if (sad_on && m32 && plain_epilogue && ks == 3 && stride == 1 && cin_pad % 16 == 0 && cout_pad % 96 == 0) {
            ConvConfig q = c; q.kc = 16;
            if (cout_pad % 192 == 0) { q.nt = 12; q.variant = 21; } else { q.nt = 6; q.variant = 22; }
            if (m32 == 4 && wo > 32) q.variant += 2;
            if (m32 == 6) q.variant = q.variant == 21 ? 21 : (wo > 32 ? 24 : 22);      // 6: 4 x 32 tiles for BN = 192 (2 x 64 measured 1.5 % slower on 34 x 60), 4 x 64 for BN = 96 on wide maps
            if (m32 >= 4 || (q.variant == 21 && m32 != 3) || (q.variant == 22 && m32 != 2)) return q;
        }
        if (tuned_on)
            for (const Tuned& t : g_tuned_split)
                if (t.ks == ks && t.s == stride && t.cin == cin_pad && t.cout == cout_pad && t.wo == wo) {
                    ConvConfig q = c; q.kc = t.kc; q.nt = t.nt; q.wx = t.wx; q.variant = t.variant;
                    if (conv_ad(q) && !(sad_on && plain_epilogue && (conv_ad_s2d(q) ? cin_pad % 16 == 0 : cin_pad % 48 == 0))) continue;       // the A-direct kernels: ReLU / none after at most two residual adds, three chunks per loop body
                    if (find_inst(precision, q) && lds_bytes(precision, q) <= 160 * 1024) return q;
                }
        if (sad_on && plain_epilogue && ks == 3 && stride == 1 && cin_pad % 48 == 0 && cout_pad % 96 == 0) {      // A-direct, split form (three 16-channel chunks per loop body)
            ConvConfig q = c; q.kc = 16;
            if (cout_pad % 192 == 0) { q.nt = 12; q.variant = 8; } else { q.nt = 6; q.variant = 9; }
            return q;
        }
        static const int skcs[] = {32, 16, 8};
        long best = -1;
        c.nt = 1; c.kc = 8; c.variant = 0;
        for (int variant : {0, 3})
            for (int nt : nts) {
                if (cout_pad % (16 * nt)) continue;
                for (int kc : skcs) {
                    if (cin_pad % kc) continue;
                    ConvConfig t = c; t.nt = nt; t.kc = kc; t.variant = variant;
                    if (!find_inst(precision, t)) continue;
                    if (lds_bytes(precision, t) > 80 * 1024) continue;
                    if (nt * conv_pw(t) > 12) continue;                    // registers: 4 accumulator VGPRs per (NT, PW) pair next to twice the fragments of the fp16 kernel
                    const long score = (long)nt * conv_pw(t) * 1000 + kc * 10 + (variant == 0 ? 1 : 0);
                    if (score > best) { best = score; c.nt = nt; c.kc = kc; c.variant = variant; }
                }
            }
        return c;
    }
    // developer override (parity tests of a specific kernel variant): EAGLE_CONV_FORCE="kc,nt,variant"
    if (const char* f = getenv("EAGLE_CONV_FORCE")) {
        ConvConfig q = c;
        if (sscanf(f, "%d,%d,%d", &q.kc, &q.nt, &q.variant) == 3 && cin_pad % q.kc == 0 && cout_pad % (16 * q.nt) == 0 &&
            find_inst(precision, q) && lds_bytes(precision, q) <= 160 * 1024 && (conv_ws(q) ? plain_one : conv_ad(q) ? plain_epilogue : true))
            return q;
    }
    // 3x3 stride-1 layers whose Cout is a multiple of 96 (HRNet's 96 / 192 / 384-channel branches): the A-direct kernel (EAGLE_CONV_AD=0: off)
    static const bool ad_on = !(getenv("EAGLE_CONV_AD") && atoi(getenv("EAGLE_CONV_AD")) == 0);
    if (ad_on && plain_epilogue && ks == 3 && stride == 1 && cin_pad % 32 == 0 && cout_pad % 96 == 0) {
        ConvConfig q = c; q.kc = 32;
        if (cout_pad % 192 == 0) { q.nt = 12; q.variant = 8; } else { q.nt = 6; q.variant = 9; }
        return q;
    }
    // 3x3 stride-2 layers with the same Cout: the A-direct kernel over the space-to-depth image (EAGLE_CONV_AD2=0: off)
    static const bool ad2_on = !(getenv("EAGLE_CONV_AD2") && atoi(getenv("EAGLE_CONV_AD2")) == 0);
    if (ad_on && ad2_on && plain_epilogue && ks == 3 && stride == 2 && cin_pad % 8 == 0 && cin_pad >= 32 && cout_pad % 96 == 0) {
        ConvConfig q = c; q.kc = 32;
        if (cout_pad % 192 == 0) { q.nt = 12; q.variant = 10; } else { q.nt = 6; q.variant = 11; }
        return q;
    }
    // fp16: per-layer table measured on MI355X (tools/autotune_conv.py); shapes not in the table use the heuristic below
    for (const Tuned& t : g_tuned)
        if (t.ks == ks && t.s == stride && t.cin == cin_pad && t.cout == cout_pad && t.wo == wo) {
            ConvConfig q = c; q.kc = t.kc; q.nt = t.nt; q.wx = t.wx; q.variant = t.variant;
            if (conv_ws(q) && !plain_one) continue;   // the weight-stationary kernel has no SiLU / second-residual / fp32-output epilogue
            if (find_inst(precision, q) && lds_bytes(precision, q) <= 160 * 1024) return q;
        }
    // heuristic: the (NT, KC) pair with the most work per staged item whose LDS footprint still lets two workgroups share a CU
    static const int kcs[] = {64, 48, 32, 16, 8};
    long best = -1;
    c.nt = 1; c.kc = 8;
    for (int nt : nts) {
        if (cout_pad % (16 * nt)) continue;
        for (int kc : kcs) {
            if (cin_pad % kc) continue;
            ConvConfig t = c; t.nt = nt; t.kc = kc;
            if (!find_inst(precision, t)) continue;
            if (lds_bytes(precision, t) > 80 * 1024) continue;
            const long score = (long)kc * nt * 1000 + kc;
            if (score > best) { best = score; c.nt = nt; c.kc = kc; }
        }
    }
    return c;
}

size_t conv_weight_elems(int precision, const ConvConfig& c)
{
    if (precision == EAGLE_PREC_F16 && conv_ad(c) && c.stride == 2) return (size_t)(c.cout_pad / (c.nt * 16)) * (4 * c.cin / 32) * 16 * (c.nt * 16) * 8;
    const int bn = c.nt * 16, nblk = c.cout_pad / bn, nch = c.cin / c.kc;
    if (precision == EAGLE_PREC_F16) return (size_t)nblk * nch * f16_ni(c.ks, c.kc) * 4 * bn * 8;
    if (precision == EAGLE_PREC_F32S && conv_ad_s2d(c)) return (size_t)nblk * (4 * c.cin / 16) * 6 * 4 * bn * 8;      // 6 K-steps per 16-channel chunk of the space-to-depth image
    if (precision == EAGLE_PREC_F32S && conv_ad_m32(c)) return (size_t)nblk * (c.cin / 16) * 18 * (bn / 32) * 512;      // 18 steps (tap, hi | lo) per 16-channel chunk, one 1-KiB fragment per 32 output channels
    if (precision == EAGLE_PREC_F32S && conv_ad(c)) return (size_t)nblk * (c.cin / 16) * 14 * 4 * bn * 8;   // 14 K-steps per 16-channel chunk
    if (precision == EAGLE_PREC_F32S) return (size_t)nblk * nch * f16_ni(c.ks, c.kc) * 2 * 4 * bn * 8;      // fp16 elements: a hi and a lo block per K-step
    return (size_t)nblk * nch * c.ks * c.ks * (c.kc / 4) * 4 * bn;
}

void conv_tile_weights(int precision, const ConvConfig& c, const float* w, int cin_real, int cout_real, void* dst, float* descale)
{
    const int bn = c.nt * 16, nblk = c.cout_pad / bn, nch = c.cin / c.kc, taps = c.ks * c.ks;
    auto W = [&](int tap, int ci, int co) -> float {
        return (ci < cin_real && co < cout_real) ? w[((size_t)tap * cin_real + ci) * cout_real + co] : 0.0f;
    };
    if (descale) *descale = 1.0f;
    if (precision == EAGLE_PREC_F32S) {
        // [Cout block][Cin chunk][K-step][hi | lo][q][BN][8]; K-step i, lane group q: (tap, channel group) pair 4 i + q in (tap-major) order
        float amax = 0.f;
        for (size_t k = 0; k < (size_t)taps * cin_real * cout_real; ++k) amax = std::max(amax, std::fabs(w[k]));
        int e = 0;
        if (amax > 0.f) (void)std::frexp(amax, &e);                        // amax in [2^(e-1), 2^e)
        const int sw = 15 - e;                                             // scaled maximum in [2^14, 2^15)
        const float scale = std::ldexp(1.0f, sw);
        if (descale) *descale = std::ldexp(1.0f, -(sw + 4));
        _Float16* d = (_Float16*)dst;
        if (conv_ad_s2d(c)) {
            // stride 2: [Cout block][s2d chunk][K-step 0..5][q][BN][8].  s2d chunk -> (phase (ry, rx), 16 real channels); K-steps 0..3 = the taps' (tyy, txx)
            // of the 2x2 kernel over the space-to-depth image, lane groups (hi g0, hi g1, hi g0, hi g1); K-steps 4, 5 = the tap' pairs (0|1), (2|3), lane
            // groups (lo g0, lo g1 | lo g0, lo g1).  tap' row 0 is the s2d row above: only its odd phase contributes (ky = 0); row 1: ky = 1 (even phase), 2 (odd).
            const int per_phase = c.cin / 16;
            for (int b = 0; b < nblk; ++b)
                for (int ch = 0; ch < 4 * per_phase; ++ch)
                    for (int k = 0; k < 6; ++k)
                        for (int qq = 0; qq < 4; ++qq)
                            for (int nn = 0; nn < bn; ++nn)
                                for (int j = 0; j < 8; ++j) {
                                    const int ph = ch / per_phase, c0 = (ch - ph * per_phase) * 16, ry = ph >> 1, rx = ph & 1;
                                    const int tp = k < 4 ? k : 2 * (k - 4) + (qq >> 1), tyy = tp >> 1, txx = tp & 1;
                                    const int ky = tyy == 0 ? (ry == 1 ? 0 : -1) : (ry == 0 ? 1 : 2), kx = txx == 0 ? (rx == 1 ? 0 : -1) : (rx == 0 ? 1 : 2);
                                    float v = 0.f;
                                    if (ky >= 0 && kx >= 0) v = W(ky * 3 + kx, c0 + (qq & 1) * 8 + j, b * bn + nn) * scale;
                                    const _Float16 hi = (_Float16)v;
                                    *d++ = k < 4 ? hi : (_Float16)(v - (float)hi);
                                }
            return;
        }
        if (conv_ad_m32(c)) {
            // 32x32x16 form: [Cout block][16-channel chunk][tap 0..8][hi | lo][BN / 32 blocks][lane 0..63][8]: lane l of a block's A fragment holds output
            // channel (l & 31) of the block and the chunk's 8-channel group (l >> 5)
            for (int b = 0; b < nblk; ++b)
                for (int ch = 0; ch < c.cin / 16; ++ch)
                    for (int tap = 0; tap < 9; ++tap)
                        for (int part = 0; part < 2; ++part)
                            for (int mb = 0; mb < bn / 32; ++mb)
                                for (int l = 0; l < 64; ++l)
                                    for (int j = 0; j < 8; ++j) {
                                        const float v = W(tap, ch * 16 + (l >> 5) * 8 + j, b * bn + mb * 32 + (l & 31)) * scale;
                                        const _Float16 hi = (_Float16)v;
                                        *d++ = part == 0 ? hi : (_Float16)(v - (float)hi);
                                    }
            return;
        }
        if (conv_ad(c)) {
            // A-direct form: [Cout block][16-channel chunk][K-step 0..13][q][BN][8].  K-steps 0..8 = taps, lane groups (hi g0, hi g1, hi g0, hi g1):
            // against the record slots (hi g0, hi g1, lo g0, lo g1) that is hi*hi + hi*lo.  K-steps 9..13 = tap pairs (0|1, 2|3, 4|5, 6|7, 8|-),
            // lane groups (lo g0, lo g1 at the first tap | lo g0, lo g1 at the second): lo*hi.
            for (int b = 0; b < nblk; ++b)
                for (int ch = 0; ch < c.cin / 16; ++ch)
                    for (int k = 0; k < 14; ++k)
                        for (int qq = 0; qq < 4; ++qq)
                            for (int nn = 0; nn < bn; ++nn)
                                for (int j = 0; j < 8; ++j) {
                                    const int tap = k < 9 ? k : 2 * (k - 9) + (qq >> 1);
                                    float v = 0.f;
                                    if (tap < 9) v = W(tap, ch * 16 + (qq & 1) * 8 + j, b * bn + nn) * scale;
                                    const _Float16 hi = (_Float16)v;
                                    *d++ = k < 9 ? hi : (_Float16)(v - (float)hi);
                                }
            return;
        }
        const int G = c.kc / 8, NGR = taps * G, NI = f16_ni(c.ks, c.kc);
        for (int b = 0; b < nblk; ++b)
            for (int ch = 0; ch < nch; ++ch)
                for (int i = 0; i < NI; ++i)
                    for (int part = 0; part < 2; ++part)
                        for (int qq = 0; qq < 4; ++qq)
                            for (int nn = 0; nn < bn; ++nn)
                                for (int j = 0; j < 8; ++j) {
                                    const int g = 4 * i + qq;
                                    float v = 0.f;
                                    if (g < NGR) { const int tap = g / G, cg = g % G; v = W(tap, ch * c.kc + cg * 8 + j, b * bn + nn) * scale; }
                                    const _Float16 hi = (_Float16)v;
                                    *d++ = part == 0 ? hi : (_Float16)(v - (float)hi);
                                }
        return;
    }
    if (precision == EAGLE_PREC_F16 && conv_ad(c) && c.stride == 2) {
        // space-to-depth form: [Cout block][chunk of 32 s2d channels][tap' (2x2) * 4 + channel group][BN][8]; s2d channel = phase * Cin + channel,
        // phase = 2 * ry + rx; tap' row 0 is the s2d row above (only its odd phase contributes: ky = 0), tap' row 1 the same row (ky = 1, 2)
        _Float16* d = (_Float16*)dst;
        const int nch2 = 4 * c.cin / 32;
        for (int b = 0; b < nblk; ++b)
            for (int ch = 0; ch < nch2; ++ch)
                for (int g = 0; g < 16; ++g)
                    for (int nn = 0; nn < bn; ++nn)
                        for (int j = 0; j < 8; ++j) {
                            const int tp = g / 4, cg = g % 4, tyy = tp >> 1, txx = tp & 1;
                            const int sc = ch * 32 + cg * 8 + j, ph = sc / c.cin, ci = sc - ph * c.cin, ry = ph >> 1, rx = ph & 1;
                            const int ky = tyy == 0 ? (ry == 1 ? 0 : -1) : (ry == 0 ? 1 : 2), kx = txx == 0 ? (rx == 1 ? 0 : -1) : (rx == 0 ? 1 : 2);
                            *d++ = (_Float16)((ky < 0 || kx < 0) ? 0.f : W(ky * 3 + kx, ci, b * bn + nn));
                        }
        return;
    }
    if (precision == EAGLE_PREC_F16) {
        const int G = c.kc / 8, NGR = taps * G, NI = f16_ni(c.ks, c.kc);
        _Float16* d = (_Float16*)dst;
        for (int b = 0; b < nblk; ++b)
            for (int ch = 0; ch < nch; ++ch)
                for (int g = 0; g < NI * 4; ++g)
                    for (int nn = 0; nn < bn; ++nn)
                        for (int j = 0; j < 8; ++j) {
                            float v = 0.f;
                            if (g < NGR) {
                                const int tap = g / G, cg = g % G;
                                v = W(tap, ch * c.kc + cg * 8 + j, b * bn + nn);
                            }
                            *d++ = (_Float16)v;
                        }
    } else {
        const int CSTEPS = c.kc / 4;
        float* d = (float*)dst;
        for (int b = 0; b < nblk; ++b)
            for (int ch = 0; ch < nch; ++ch)
                for (int tap = 0; tap < taps; ++tap)
                    for (int cs = 0; cs < CSTEPS; ++cs)
                        for (int qq = 0; qq < 4; ++qq)
                            for (int nn = 0; nn < bn; ++nn) *d++ = W(tap, ch * c.kc + cs * 4 + qq, b * bn + nn);
    }
}

static std::mutex g_page_mutex;
static const void* conv_zero_page()
{
    static void* z[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_page_mutex);
    if (!z[dev]) { HIP_CHECK(hipMalloc(&z[dev], 256)); HIP_CHECK(hipMemset(z[dev], 0, 256)); }
    return z[dev];
}
static void* conv_trash_page()
{
    static void* z[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_page_mutex);
    if (!z[dev]) HIP_CHECK(hipMalloc(&z[dev], 8192));
    return z[dev];
}

void conv_launch(int precision, const ConvLaunch& L, hipStream_t s)
{
    const ConvConfig& c = L.cfg;
    const Inst* inst = find_inst(precision, c);
    if (!inst) fail(EAGLE_E_NOKERNEL, "no conv kernel instance: prec=%d ks=%d s=%d kc=%d nt=%d", precision, c.ks, c.stride, c.kc, c.nt);
    ConvArgs a;
    a.x = L.x.p; a.xcs = L.x.cs; a.xoff = L.x.off; a.N = L.x.n; a.H = L.x.h; a.W = L.x.w;
    a.w = L.w; a.bias = L.bias;
    a.y = L.y.p; a.ycs = L.y.cs; a.yoff = L.y.off; a.Ho = L.y.h; a.Wo = L.y.w;
    a.r1 = L.r1.p; a.r1cs = L.r1.cs; a.r1off = L.r1.off;
    a.r2 = L.r2.p; a.r2cs = L.r2.cs; a.r2off = L.r2.off;
    a.descale = L.descale;
    if (precision == EAGLE_PREC_F32S) {                     // the kernels address split tensors in fp16 elements: two per logical channel
        if (L.x.f32 != 2 || (L.r1.p && L.r1.f32 != 2) || (L.r2.p && L.r2.f32 != 2) || L.y.f32 != (L.out_f32 ? 1 : 2))
            fail(EAGLE_E_INVALID, "split conv: operand tensor formats do not match the family");
        a.xcs *= 2; a.xoff *= 2; a.r1cs *= 2; a.r1off *= 2; a.r2cs *= 2; a.r2off *= 2;
        if (!L.out_f32) { a.ycs *= 2; a.yoff *= 2; }
    }
    if (!a.r1 && a.r2) { a.r1 = a.r2; a.r1cs = a.r2cs; a.r1off = a.r2off; a.r2 = nullptr; }       // a single residual is always operand 1 (IEEE addition is commutative: same bits)
    a.pre_act = L.pre_act; a.post_act = L.post_act; a.out_f32 = L.out_f32 || precision == EAGLE_PREC_F32;
    a.wx = c.wx;
    const int th = 4 * conv_pw(c) / c.wx, tw = 16 * c.wx;
    a.tiles_x = (a.Wo + tw - 1) / tw; a.tiles_y = (a.Ho + th - 1) / th;
    a.nchunks = c.cin / c.kc;
    a.zeros = conv_zero_page(); a.trash = conv_trash_page(); a.xcd = 0; a.gy = 1; a.stack = 0;
    if (precision == EAGLE_PREC_F32) {
        // row stacking (conv_f32_kernel): stride-1 layers of a batch are tiled as one image of N (Ho + 1) rows
        const char* se = getenv("EAGLE_F32_STACK");          // (read per launch: the parity test switches it)
        const bool stack_on = !(se && atoi(se) == 0);
        if (stack_on && c.stride == 1 && a.N > 1 && a.Ho == a.H) { a.stack = 1; a.tiles_y = (a.N * (a.Ho + 1) + th - 1) / th; }
        // raw buffer descriptors with 32-bit byte offsets for the activation loads
        if ((size_t)a.N * a.H * a.W * a.xcs * 4 >= ((size_t)1 << 31)) fail(EAGLE_E_INVALID, "fp32 conv: the input tensor of %d frames reaches 2 GiB; use a smaller device batch", a.N);
    }
    a.am = L.am_slot ? *L.am_slot : nullptr; a.am_cs = c.cout_pad;
    a.sat = (L.sat_slot && precision == EAGLE_PREC_F32S) ? *L.sat_slot : nullptr;
    if (conv_ad(c)) {                                       // A-direct: persistent over XCD-contiguous item ranges, two workgroups per CU
        const bool split = precision == EAGLE_PREC_F32S;
        if (a.out_f32 || a.pre_act != 0 || a.post_act > 1 || L.am_slot || c.kc != (split ? 16 : 32) || c.ks != 3 || c.stride != ((conv_ad_s2d(c) || conv_ad_s2t(c)) ? 2 : 1) || (split && !conv_ad_s2d(c) && !conv_ad_m32(c) && c.cin % 48) || (split && (conv_ad_s2d(c) || conv_ad_m32(c)) && c.cin % 16) || (conv_ad_m32(c) && !split) || (conv_ad_s2t(c) && !split))
            fail(EAGLE_E_NOKERNEL, "A-direct conv needs 3x3, kc = 32 (16 in the split family, stride 1 only), 2-byte / split output, pre_act none, post_act in {none, ReLU}");
        if (conv_ad_s2d(c)) a.nchunks = split ? 4 * c.cin / 16 : 4 * c.cin / 32;      // chunks of the space-to-depth image
        const int thh = conv_ad_rows(c);
        a.tiles_x = (a.Wo + conv_ad_tw(c) - 1) / conv_ad_tw(c); a.tiles_y = (a.Ho + thh - 1) / thh;
        a.gy = c.cout_pad / (c.nt * 16);
        const size_t lim = (size_t)1 << 31;
        if ((size_t)a.N * a.H * a.W * a.xcs * 2 >= lim || (size_t)a.N * a.Ho * a.Wo * std::max(std::max(a.ycs, a.r1 ? a.r1cs : 0), a.r2 ? a.r2cs : 0) * 2 >= lim)
            fail(EAGLE_E_INVALID, "fp16 conv: a tensor of %d frames reaches 2 GiB; use a smaller device batch", a.N);
        const int nres = (a.r1 ? 1 : 0) + (a.r2 ? 1 : 0);
        const int items = a.tiles_x * a.tiles_y * a.N * a.gy;
        const bool deep = items <= 256;                     // at most one workgroup per CU: the deep weight ring (conv_ad_split32.hip)
        const ConvKernel fn = conv_ad_m32(c) ? (conv_ad_tw(c) == 64 ? conv_ad_split32_kernel_w64(conv_ad_wide(c), nres, deep) : conv_ad_split32_kernel(conv_ad_wide(c), nres, deep)) : (split && conv_ad_s2t(c)) ? conv_ad_split_kernel_s2t(c.variant == 14, nres) : (split && c.stride == 2) ? conv_ad_split_kernel_s2(conv_ad_wide(c), nres) : (split && c.variant == 13) ? conv_ad_split_kernel48sb(nres) : (split && c.variant == 19) ? conv_ad_split_kernel48ring(nres) : (split && c.variant == 12) ? conv_ad_split_kernel48(nres) : split ? conv_ad_split_kernel(conv_ad_wide(c), nres) : c.stride == 2 ? conv_ad_kernel_s2(conv_ad_wide(c), nres) : conv_ad_kernel_s1(conv_ad_wide(c), nres);
        ensure_max_dynamic_lds((const void*)fn, 160 * 1024);
        // workgroups per launch: one per item (the hardware hands a queued workgroup to whichever CU frees a slot: dynamic balance) rather than 512
        // resident ones walking static item ranges — same box, alternating: 774.4 / 774.5 -> 779.0 / 780.7 frames/s, 96->96 209.9 -> 204.4 us,
        // 192->192 180.3 -> 177.2 (768 workgroups: 707 frames/s — 1.5 rounds of uneven ranges).  EAGLE_CONV_AD_SLOTS=512 restores the persistent form.
        static const int slots = getenv("EAGLE_CONV_AD_SLOTS") ? atoi(getenv("EAGLE_CONV_AD_SLOTS")) : (1 << 30);
        // variant 19 (one workgroup per CU by its LDS footprint): persistent, so that the halo ring's prefetch runs on from item to item
        hipLaunchKernelGGL(fn, dim3(std::min(items, c.variant == 19 ? 256 : slots)), dim3(256), lds_bytes(precision, c), s, a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (a.am && (!prec_is_f16_kernels(precision) || conv_ws(c)))
        fail(EAGLE_E_NOKERNEL, "fused heat-map maxima need the generic fp16 kernel");
    if (prec_is_f16_kernels(precision)) {                   // the fp16 kernels address tensors through raw buffer descriptors with 32-bit byte offsets
        const size_t lim = (size_t)1 << 31;
        const size_t cs_out = std::max(std::max(a.out_f32 ? 2 * a.ycs : a.ycs, a.r1 ? a.r1cs : 0), a.r2 ? a.r2cs : 0);
        if ((size_t)a.N * a.H * a.W * a.xcs * 2 >= lim || (size_t)a.N * a.Ho * a.Wo * cs_out * 2 >= lim)
            fail(EAGLE_E_INVALID, "fp16 conv: a tensor of %d frames reaches 2 GiB; use a smaller device batch", a.N);
    }
    const size_t lds = lds_bytes(precision, c);
    ensure_max_dynamic_lds((const void*)inst->fn, 160 * 1024);
    const int gy = c.cout_pad / (c.nt * 16);
    int gx = a.stack ? a.tiles_x * a.tiles_y : a.tiles_x * a.tiles_y * a.N;
    if (conv_ws(c)) {                                       // persistent, weight-stationary: 8*gy | grid, as many workgroups as stay resident
        if (c.kc != c.cin || a.out_f32 || a.r2 || a.pre_act != 0 || a.post_act > 1 || (size_t)a.N * a.H * a.W * a.xcs * 2 >= (1ull << 31) || (size_t)a.N * a.Ho * a.Wo * std::max(a.ycs, a.r1 ? a.r1cs : 0) * 2 >= (1ull << 31))
            fail(EAGLE_E_NOKERNEL, "weight-stationary conv needs kc == cin, fp16 output, at most one residual, pre_act none, post_act in {none, ReLU} and tensors below 2 GiB (kc=%d cin=%d)", c.kc, c.cin);
        static const int ws_cap = getenv("EAGLE_CONV_WS_PER_CU") ? atoi(getenv("EAGLE_CONV_WS_PER_CU")) : 2;     // developer knob, as above
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(ws_cap, (160 * 1024) / lds));
        const int unit = 8 * gy;
        gx = std::max(unit, std::min((gx * gy + unit - 1) / unit * unit, 256 * per_cu / unit * unit));
    }
    a.gy = gy;
    static const int xcd_env = getenv("EAGLE_CONV_XCD") ? atoi(getenv("EAGLE_CONV_XCD")) : 1;
    // 1x1 layers with several Cout blocks re-read their input tile once per block: in tile-major / block-minor order on one XCD the
    // re-reads hit that XCD's L2 instead of HBM
    a.xcd = (prec_is_f16_kernels(precision) && !conv_ws(c) && (c.ks == 3 || (c.ks == 1 && gy > 1)) && xcd_env) ? 1 : 0;
    dim3 grid(gx, gy);
    if (a.xcd) grid = dim3(gx * gy, 1);
    if (conv_ws(c)) grid = dim3(gx, 1);
    hipLaunchKernelGGL(inst->fn, grid, dim3(256), lds, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
