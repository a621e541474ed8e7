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
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "dmath.h"

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
    int xcd;                // 1: XCD-aware work order (1-D grid; each XCD owns a contiguous range of (tile, Cout-block) items)
    ArgmaxPart* am;         // fused heat-map maxima (head convolution): partials [frame][tile][am_cs] instead of the output tensor
    int am_cs;
    const void* zeros;      // >= 16 zero bytes in global memory (source of out-of-image pixels for unconditional loads / LDS-DMA)
    void* trash;            // >= 4 KiB of scratch global memory (target of out-of-image results for unconditional stores)
};

// ------------------------------------------------------------------------------------------------------------
// fp16 family
// ------------------------------------------------------------------------------------------------------------
template <int KC> struct F16Geom {
    static constexpr int G = KC / 8;                                   // 16-byte groups per pixel per chunk
    static constexpr int PS = KC * 2 + ((G % 2 == 0) ? 16 : 0);        // LDS pixel stride (bytes), odd in 16-B units
};

// Epilogue of the fp16 kernels.  The MFMA result layout gives a lane 4 consecutive channels of one pixel, i.e. 8-byte stores
// in 32-byte runs: measured on MI355X those partial-line writes, not the MFMAs or the loads, bound the narrow layers (a
// 48->48 3x3 at 135x240 spent 95 of 173 us in them).  So every wave transposes its PW x 16 pixels x BN channels through a
// private LDS strip (the tile's operand space is free once the last MFMA has read it) and writes 16 bytes per lane,
// contiguous across the wave wherever the tensor is (BN*2-byte runs per pixel, whole pixel rows when ycs == BN).
using i32x4 = __attribute__((ext_vector_type(4))) int;
// raw buffer descriptor (gfx9 layout): base, stride 0, num_records bytes, DATA_FORMAT = 32-bit: out-of-range loads return 0, stores are dropped
__device__ __forceinline__ i32x4 make_rsrc(const void* p, int bytes)
{
    const unsigned long long b = (unsigned long long)p;
    i32x4 r = {(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xFFFF), bytes, 0x00020000};
    return r;
}
// act in {0: none, 1: ReLU} without a branch per value (d_act's SiLU arm keeps the compiler from if-converting it)
__device__ __forceinline__ float relu_if(float v, bool relu) { return (relu && !(v > 0.0f)) ? 0.0f : v; }

using rsrc_t = __amdgpu_buffer_rsrc_t;
// raw buffer view of a tensor (the host guarantees < 2 GiB per tensor for the fp16 kernels): 32-bit byte offsets, loads beyond
// num_records return 0 and stores there are dropped, so out-of-image lanes need no branch and no 64-bit pointer arithmetic
__device__ __forceinline__ rsrc_t tensor_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, p ? 0x7FFFFFFF : 0, 0x00020000); }
constexpr unsigned OOB_OFF = 0x80000000u;

// Staged epilogue: every stage is ONE uniform test around straight-line code over all NT*PW*4 values of the lane (a test per
// value, as d_act() would give, cost more instructions than the MFMAs of a narrow layer).
// r1pre: the first residual, already loaded by the caller under the last chunk's MFMAs (nullptr: load it here)
template <int NT, int PW>
__device__ __forceinline__ void f16_epilogue(const ConvArgs& a, f32x4 (&acc)[NT][PW], char* strip,
                                             int n, int oy0, int ox0, int nb, int wave, int q, int lx, int lane,
                                             const u32x2 (*r1pre)[PW] = nullptr, char* smem_base = nullptr)
{
    constexpr int BN = NT * 16, GO = BN / 8, RS = BN * 2 + 16;
    const int WX = a.wx;
    unsigned pix[PW];                                      // linear output pixel of sub-tile p, or OOB
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int s = wave * PW + p, row = s / WX, xb = s - row * WX;
        const int oy = oy0 + row, ox = ox0 + xb * 16 + lx;
        pix[p] = (oy < a.Ho && ox < a.Wo) ? (unsigned)((n * a.Ho + oy) * a.Wo + ox) : OOB_OFF;
    }
    const int co0 = nb * BN + q * 4;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const float4 bv = *(const float4*)(a.bias + co0 + tt * 16);
#pragma unroll
        for (int p = 0; p < PW; ++p) { acc[tt][p][0] += bv.x; acc[tt][p][1] += bv.y; acc[tt][p][2] += bv.z; acc[tt][p][3] += bv.w; }
    }
#define EP_ALL(EXPR_) _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) _Pragma("unroll") for (int p = 0; p < PW; ++p) _Pragma("unroll") for (int r = 0; r < 4; ++r) { const float v = acc[tt][p][r]; acc[tt][p][r] = (EXPR_); }
#define EP_ACT(ACT_) if ((ACT_) == 1) { EP_ALL(v > 0.0f ? v : 0.0f) } else if ((ACT_) == 2) { EP_ALL(v * d_sigmoidf(v)) }
#define EP_RES(PTR_, CS_, OFF_, ORDER_)                                                                                \
    if (PTR_) {                                                                                                        \
        const rsrc_t rs_ = tensor_rsrc(PTR_);                                                                          \
        u32x2 rr_[NT][PW];                                                                                             \
        _Pragma("unroll") for (int p = 0; p < PW; ++p) {                                                               \
            const unsigned vo_ = pix[p] == OOB_OFF ? OOB_OFF : (pix[p] * (unsigned)(CS_) + (unsigned)((OFF_) + co0)) * 2u; \
            _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) rr_[tt][p] = __builtin_amdgcn_raw_buffer_load_b64(rs_, vo_ + tt * 32, 0, 0); \
        }                                                                                                              \
        _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) _Pragma("unroll") for (int p = 0; p < PW; ++p) {             \
            const half4 rh_ = __builtin_bit_cast(half4, rr_[tt][p]);                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[tt][p][r] = ORDER_ ? (float)rh_[r] + acc[tt][p][r] : acc[tt][p][r] + (float)rh_[r]; \
        }                                                                                                              \
    }
    EP_ACT(a.pre_act)
    if (r1pre) {
        if (a.r1) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    const half4 rh_ = __builtin_bit_cast(half4, r1pre[tt][p]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[tt][p][r] = (float)rh_[r] + acc[tt][p][r];
                }
        }
    } else {
        EP_RES(a.r1, a.r1cs, a.r1off, true)
    }
    EP_RES(a.r2, a.r2cs, a.r2off, false)
    EP_ACT(a.post_act)
#undef EP_ALL
#undef EP_ACT
#undef EP_RES
    if (a.am) {
        // K5 fused (KeypointModel.get_keypoints, kh.py:581-593): per channel the first maximum of sigmoid(logit) over this tile.
        // The sigmoid is applied before the compare, as np.argmax sees it; ties go to the smaller row-major index.
        float bs[NT * 4]; int bi[NT * 4];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float best = -1.0f; int idx = 0x7fffffff;
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    if (pix[p] == OOB_OFF) continue;
                    const float sg = d_sigmoidf(acc[tt][p][r]);
                    const int li = (int)pix[p] - n * a.Ho * a.Wo;
                    if (sg > best || (sg == best && li < idx)) { best = sg; idx = li; }
                }
                bs[tt * 4 + r] = best; bi[tt * 4 + r] = idx;
            }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1)                   // the 16 lanes that share q hold the same channels for 16 different pixels
#pragma unroll
            for (int k = 0; k < NT * 4; ++k) {
                const float ob = __shfl_xor(bs[k], m, 64); const int oi = __shfl_xor(bi[k], m, 64);
                if (ob > bs[k] || (ob == bs[k] && oi < bi[k])) { bs[k] = ob; bi[k] = oi; }
            }
        ArgmaxPart* red = (ArgmaxPart*)smem_base;            // [4 waves][BN]: the operand space is free after the last MFMA
        if (lx == 0) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ArgmaxPart v; v.score = bs[tt * 4 + r]; v.idx = bi[tt * 4 + r]; red[wave * BN + tt * 16 + q * 4 + r] = v; }
        }
        __syncthreads();
        const int tid = wave * 64 + lane;
        if (tid < BN) {
            ArgmaxPart b = red[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) { const ArgmaxPart o = red[w * BN + tid]; if (o.score > b.score || (o.score == b.score && o.idx < b.idx)) b = o; }
            const int TH = 4 * PW / WX, TW = 16 * WX;
            const int tile = (oy0 / TH) * a.tiles_x + ox0 / TW;
            a.am[((size_t)n * a.tiles_x * a.tiles_y + tile) * a.am_cs + nb * BN + tid] = b;
        }
        return;
    }
    if (a.out_f32) {
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            if (pix[p] == OOB_OFF) continue;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                *(float4*)((float*)a.y + (size_t)pix[p] * a.ycs + a.yoff + co0 + tt * 16) = make_float4(acc[tt][p][0], acc[tt][p][1], acc[tt][p][2], acc[tt][p][3]);
        }
        return;
    }
    // fp16: transpose through this wave's LDS strip (LDS operations of one wave execute in order), then 16 bytes per lane,
    // BN*2-byte runs per pixel (whole pixel rows when ycs == BN)
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            half4 o = {(_Float16)acc[tt][p][0], (_Float16)acc[tt][p][1], (_Float16)acc[tt][p][2], (_Float16)acc[tt][p][3]};
            *(half4*)(strip + (p * 16 + lx) * RS + (tt * 16 + q * 4) * 2) = o;
        }
    const rsrc_t ys = tensor_rsrc(a.y);
#pragma unroll
    for (int e0 = 0; e0 < PW * 16 * GO; e0 += 64) {
        const int e = e0 + lane;
        const int px = e / GO, grp = e - px * GO, p = px >> 4, lxp = px & 15;
        const int s = wave * PW + p, row = s / WX, xb = s - row * WX;
        const int oy = oy0 + row, ox = ox0 + xb * 16 + lxp;
        const bool ok = e < PW * 16 * GO && oy < a.Ho && ox < a.Wo;
        const u32x4 v = *(const u32x4*)(strip + (ok ? px * RS + grp * 16 : 0));
        const unsigned vo = ok ? ((unsigned)((n * a.Ho + oy) * a.Wo + ox) * (unsigned)a.ycs + (unsigned)(a.yoff + nb * BN + grp * 8)) * 2u : OOB_OFF;
        __builtin_amdgcn_raw_buffer_store_b128(v, ys, vo, 0, 0);
    }
}

// One workgroup per (output tile, Cout block).  Per Cin-chunk every thread first ISSUES all of its 16-byte global loads
// (weight slice + halo tile) back to back into registers and only then writes them to LDS, so a chunk costs one memory
// round trip instead of one per staging iteration; tile shapes are chosen so that two workgroups share a CU (LDS <= 80 KiB,
// <= 256 VGPRs) and one workgroup's MFMAs hide the other's staging.
// Scheduling recipe of the MFMA phase (one basic block: NI steps of NT + PW ds_read_b128 and NT * PW MFMAs).  Left alone, hipcc
// issues a step's reads right in front of the MFMAs that consume them and waits (measured: the LDS latency of every step is
// exposed, ~20 % of the kernel).  The recipe spreads the reads of step i+1, one at a time, over the MFMAs of step i.
template <int R_, int M_, int r> __device__ __forceinline__ void sg_step()
{
    if constexpr (r < R_) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        // one DS read
        __builtin_amdgcn_sched_group_barrier(0x008, (M_ * (r + 1)) / R_ - (M_ * r) / R_, 0);      // its share of the MFMAs
        sg_step<R_, M_, r + 1>();
    }
}
template <int NI_, int R_, int M_, int i> __device__ __forceinline__ void sg_all()
{
    if constexpr (i + 1 < NI_) { sg_step<R_, M_, 0>(); sg_all<NI_, R_, M_, i + 1>(); }
}
template <int NI_, int NT_, int PW_> __device__ __forceinline__ void mfma_phase_schedule()
{
    __builtin_amdgcn_sched_group_barrier(0x100, NT_ + PW_, 0);          // step 0's fragments
    sg_all<NI_, NT_ + PW_, NT_ * PW_, 0>();
    __builtin_amdgcn_sched_group_barrier(0x008, NT_ * PW_, 0);          // the last step's MFMAs
}

template <int KS, int S, int KC, int NT, bool PIPE, int PW>
__global__ __launch_bounds__(256, 2) void conv_f16_kernel(ConvArgs a)
{
    constexpr int G = F16Geom<KC>::G;
    constexpr int PS = F16Geom<KC>::PS;
    constexpr int TAPS = KS * KS;
    constexpr int NGR = TAPS * G;
    constexpr int NI = (NGR + 3) / 4;
    constexpr int BN = NT * 16;
    constexpr int WBYTES = NI * 4 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w = smem;
    char* lds_a = smem + WBYTES + 16;                      // 16 spare bytes below the halo image: target of the items beyond it

    const int WX = a.wx, TH = 4 * PW / WX, TW = 16 * WX;
    const int halo_w = (TW - 1) * S + KS, halo_h = (TH - 1) * S + KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, lx = lane & 15;
    int t = blockIdx.x, nb = blockIdx.y;
    if (a.xcd) {
        // The dispatcher places workgroup b on XCD b % 8 (observed; used for L2 locality only).  Give every XCD a contiguous
        // range of items ordered (tile-major, Cout-block-minor): a tile's Cout-blocks re-read its halo from the same L2, and
        // neighbouring tiles share their halo rows there.
        const int gy = a.gy, total = a.tiles_x * a.tiles_y * a.N * gy;
        const int b = blockIdx.x, xcd = b & 7, k = b >> 3, qn = total >> 3, rn = total & 7;
        const int item = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + k;
        t = item / gy; nb = item - t * gy;
    }
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - KS / 2, ix0 = ox0 * S - KS / 2;

    f32x4 acc[NT][PW];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int p = 0; p < PW; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};

    int abase[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int s = wave * PW + p, row = s / WX, xb = s - row * WX;
        abase[p] = ((row * S) * halo_w + (xb * 16 + lx) * S) * PS;
    }
    int koff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = 4 * i + q;
        if (g < NGR) {
            const int tap = g / G, cg = g - tap * G, ky = tap / KS, kx = tap - ky * KS;
            koff[i] = (ky * halo_w + kx) * PS + cg * 16;
        } else {
            koff[i] = 0;
        }
    }
    const int wlane = (q * BN + lx) * 16;
    const int ngroups = halo_h * halo_w * G;

    // Staging of one Cin-chunk: every thread issues all of its 16-byte global loads back to back (LOAD), and writes them to
    // LDS later (STORE).  PIPE = true: the loads of chunk ch+1 are issued before the MFMAs of chunk ch and land under them.
    // The geometry of a thread's items does not depend on the chunk: their global byte offsets (out-of-image -> beyond the buffer,
    // which loads as zeros) and LDS offsets are computed once; a chunk adds only a scalar offset.
    constexpr int MAXPIX = (KS == 1) ? 64 * PW : ((S == 1) ? (PW == 4 ? 340 : (PW == 2 ? 204 : 136)) : (PW == 4 ? 1105 : (PW == 2 ? 585 : 325)));   // largest halo over wx in {1,2}
    constexpr int NPA = (MAXPIX * G + 255) / 256;
    constexpr int NPW = (WBYTES / 16 + 255) / 256;
    constexpr int RND = PIPE ? NPA : 8;                  // activation groups in flight per thread and round
    static_assert(!PIPE || NPA <= 8, "pipelined staging is meant for small chunks");
    const rsrc_t xrs = tensor_rsrc(a.x), wrs = tensor_rsrc(a.w);
    unsigned aoff[NPA];
    int ldso[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int idx = tid + 256 * j;
        const int pix = idx / G, g = idx - pix * G;
        const int hy = pix / halo_w, hx = pix - hy * halo_w;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = idx < ngroups && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        aoff[j] = ok ? (unsigned)(((n * a.H + iy) * a.W + ix) * a.xcs + g * 8) * 2u : OOB_OFF;
        ldso[j] = idx < ngroups ? pix * PS + g * 16 : -16;                 // -16: the spare slot below the halo image
    }
    u32x4 pw[NPW], pa[RND];
#define STAGE_LOAD(CH_, R0_)                                                                                            \
    {                                                                                                                   \
        if ((R0_) == 0) {                                                                                               \
            const unsigned wso_ = (unsigned)(nb * a.nchunks + (CH_)) * (unsigned)WBYTES;                                \
            _Pragma("unroll") for (int i = 0; i < NPW; ++i) { const int o = (tid + 256 * i) * 16; pw[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, o < WBYTES ? o : 0, wso_, 0); } \
        }                                                                                                               \
        const unsigned xso_ = (unsigned)(a.xoff + (CH_) * KC) * 2u;                                                     \
        _Pragma("unroll") for (int j = 0; j < RND; ++j)                                                                 \
            if ((R0_) + j < NPA) pa[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff[(R0_) + j < NPA ? (R0_) + j : 0], xso_, 0); \
    }
#define STAGE_STORE(R0_)                                                                                                \
    {                                                                                                                   \
        if ((R0_) == 0) {                                                                                               \
            _Pragma("unroll") for (int i = 0; i < NPW; ++i) { const int o = (tid + 256 * i) * 16; if (o < WBYTES) *(u32x4*)(lds_w + o) = pw[i]; } \
        }                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < RND; ++j)                                                                 \
            if ((R0_) + j < NPA) *(u32x4*)(lds_a + ldso[(R0_) + j < NPA ? (R0_) + j : 0]) = pa[j];                      \
    }
    // residual prefetch: only where the registers are there (measured: +5 % on NT*PW = 12, -8 % on NT*PW = 16, which is at 252 VGPRs with it)
    constexpr bool RPRE = KS == 3 && NT * PW <= 12;
    u32x2 r1pre[NT][PW];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int p = 0; p < PW; ++p) r1pre[tt][p] = u32x2{0u, 0u};
    if (PIPE) STAGE_LOAD(0, 0)
    for (int ch = 0; ch < a.nchunks; ++ch) {
        if (PIPE) {
            if (ch > 0) __syncthreads();                   // every wave is done reading the previous chunk from LDS
            STAGE_STORE(0)
            __syncthreads();
            if (ch + 1 < a.nchunks) STAGE_LOAD(ch + 1, 0)
        } else {
#pragma unroll
            for (int r0 = 0; r0 < NPA; r0 += RND) {
                STAGE_LOAD(ch, r0)
                STAGE_STORE(r0)
            }
            __syncthreads();
        }
        if (RPRE && ch == a.nchunks - 1 && a.r1) {          // the residual lands under the last chunk's MFMAs instead of stalling the epilogue
            const rsrc_t rs_ = tensor_rsrc(a.r1);
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                const int s = wave * PW + p, row = s / WX, xb_ = s - row * WX;
                const int oy = oy0 + row, ox = ox0 + xb_ * 16 + lx;
                const unsigned vo_ = (oy < a.Ho && ox < a.Wo) ? ((unsigned)((n * a.Ho + oy) * a.Wo + ox) * (unsigned)a.r1cs + (unsigned)(a.r1off + nb * BN + q * 4)) * 2u : OOB_OFF;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) r1pre[tt][p] = __builtin_amdgcn_raw_buffer_load_b64(rs_, vo_ + tt * 32, 0, 0);
            }
        }
        // a wave in its MFMA phase outranks the co-resident workgroup's staging / epilogue instructions (measured per class, 5 alternating
        // runs: 384->384 95.8 -> 90.1 us, 192->192 88.0 -> 86.7 us with NT = 4; 96->96 with NT = 3 loses 2 %, hence the condition)
#ifndef EAGLE_NO_SETPRIO
        if constexpr (NT >= 4) __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            half8 wa[NT], xb[PW];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) wa[tt] = *(const half8*)(lds_w + wlane + i * (4 * BN * 16) + tt * 256);
#pragma unroll
            for (int p = 0; p < PW; ++p) xb[p] = *(const half8*)(lds_a + abase[p] + koff[i]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int p = 0; p < PW; ++p)
                    acc[tt][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[tt], xb[p], acc[tt][p], 0, 0, 0);
        }
        mfma_phase_schedule<NI, NT, PW>();
#ifndef EAGLE_NO_SETPRIO
        if constexpr (NT >= 4) __builtin_amdgcn_s_setprio(0);
#endif
        if (!PIPE) __syncthreads();
    }
#undef STAGE_LOAD
#undef STAGE_STORE

    // epilogue (every wave is past the barrier that follows the last chunk's MFMAs, so the operand space is free)
    if (PIPE) __syncthreads();
    constexpr int RS_ = BN * 2 + 16;
    f16_epilogue<NT, PW>(a, acc, smem + wave * (PW * 16 * RS_), n, oy0, ox0, nb, wave, q, lx, lane, RPRE ? r1pre : nullptr, smem);
}

// ------------------------------------------------------------------------------------------------------------
// fp16 family, variants 6/7: weight-stationary persistent 3x3 stride-1 kernel for the narrow, wide-map layers (HRNet's 48- and
// 96-channel branches), which are bound by memory latency rather than by MFMA issue in the one-tile-per-workgroup kernel.
//   * the WHOLE K = 9*CIN weight slice of the workgroup's Cout block is copied to LDS once; the workgroup then walks a
//     contiguous range of output tiles, so per tile only the activation halo is staged (the generic kernel re-stages the
//     weights, the larger operand for these layers, for every tile and every Cin-chunk);
//   * software pipeline across tiles: the global loads of tile t+1's halo and of tile t's residual are issued before the
//     MFMAs of tile t and land under them; one memory round trip per tile, off the critical path;
//   * halo item geometry (LDS offset, global offset, halo row/column) is tile-invariant and computed once per thread.
// Weight tiling = conv_tile_weights with kc = CIN (a single chunk).
// ------------------------------------------------------------------------------------------------------------
template <int CIN, int NT, int PW, int WGS>
__global__ __launch_bounds__(256, WGS) void conv_f16_ws_kernel(ConvArgs a)
{
    constexpr int KS = 3;
    constexpr int G = CIN / 8;
    constexpr int PS = F16Geom<CIN>::PS;
    constexpr int NGR = 9 * G;
    constexpr int NI = (NGR + 3) / 4;
    constexpr int BN = NT * 16, GO = BN / 8, RS = BN * 2 + 16;
    constexpr int WBYTES = NI * 4 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w = smem;
    char* lds_a = smem + WBYTES + 16;                      // 16 spare bytes below the halo image: target of the items beyond it

    const int WX = a.wx, TH = 4 * PW / WX, TW = 16 * WX;
    const int halo_w = TW + 2, halo_h = TH + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, lx = lane & 15;
    char* strip = lds_a + halo_h * halo_w * PS + wave * (PW * 16 * RS);     // this wave's output transpose strip

    // workgroup -> (Cout block, slot); slots of one XCD are neighbours in tile order, and the gy workgroups that share a
    // slot's tiles sit on the same XCD (workgroup b runs on XCD b % 8), so the second one finds the halo in that L2
    const int gy = a.gy, nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, kx8 = blockIdx.x >> 3, per_xcd = nwg >> 3;
    const int nb = kx8 % gy, slot = xcd * (per_xcd / gy) + kx8 / gy, nslots = nwg / gy;
    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    const int t_begin = (int)((long)ntiles * slot / nslots), t_end = (int)((long)ntiles * (slot + 1) / nslots);
    if (t_begin >= t_end) return;

    {   // weights: once per workgroup, straight into LDS (lane-linear 1 KiB slabs); they are older than every halo load, so the
        // first counted wait on the halo registers covers them
        static_assert(WBYTES % 1024 == 0, "weight slice is a whole number of 1 KiB slabs");
        const char* wsrc = (const char*)a.w + (size_t)nb * WBYTES + lane * 16;
        for (int ws = wave; ws < WBYTES / 1024; ws += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + ws * 1024),
                                             (__attribute__((address_space(3))) void*)(lds_w + ws * 1024), 16, 0, 0);
    }

    int abase[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int s = wave * PW + p, row = s / WX, xb = s - row * WX;
        abase[p] = (row * halo_w + xb * 16 + lx) * PS;
    }
    int koff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = 4 * i + q;
        if (g < NGR) {
            const int tap = g / G, cg = g - tap * G, ky = tap / KS, kx = tap - ky * KS;
            koff[i] = (ky * halo_w + kx) * PS + cg * 16;
        } else {
            koff[i] = 0;
        }
    }
    const int wlane = (q * BN + lx) * 16;
    const int ngroups = halo_h * halo_w * G;

    // Global memory goes through raw buffer descriptors: the address of an item is (tile base + per-thread constant) in ONE 32-bit
    // VGPR add, out-of-image items get an offset beyond num_records (loads return 0, stores are dropped), and no 64-bit pointer
    // arithmetic, zero page or scratch page is needed.  (The host selects this kernel only for tensors below 2 GiB.)
    const i32x4 xrs = make_rsrc(a.x, 0x7FFFFFFF), yrs = make_rsrc(a.y, 0x7FFFFFFF), rrs = make_rsrc(a.r1, a.r1 ? 0x7FFFFFFF : 0);
    constexpr int OOB = (int)0x80000000;

    // tile-invariant geometry of this thread's halo items, residual items and output items (byte offsets from the tile origin)
    constexpr int MAXPIX = (PW == 4) ? 340 : ((PW == 2) ? 204 : 136);        // largest halo over wx in {1,2}
    constexpr int NPA = (MAXPIX * G + 255) / 256;
    int loff[NPA], goff[NPA], hyx[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int idx = tid + 256 * j;
        const int pix = idx / G, g = idx - pix * G;
        const int hy = pix / halo_w, hx = pix - hy * halo_w;
        loff[j] = (idx < ngroups) ? pix * PS + g * 16 : -16;
        goff[j] = ((hy * a.W + hx) * a.xcs + g * 8) * 2;
        hyx[j] = (idx < ngroups) ? ((hy << 16) | hx) : (0x4000 << 16);       // beyond the halo: a row no image has
    }
    int r_off[PW], r_yx[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int s = wave * PW + p, row = s / WX, col = (s - row * WX) * 16 + lx;
        r_off[p] = ((row * a.Wo + col) * a.r1cs + a.r1off + nb * BN + q * 4) * 2;
        r_yx[p] = (row << 16) | col;
    }
    constexpr int NSO = (PW * 16 * GO + 63) / 64;                            // 16-byte output items per lane
    int so_l[NSO], so_yx[NSO], so_g[NSO];
#pragma unroll
    for (int k = 0; k < NSO; ++k) {
        const int e = k * 64 + lane;
        const int pix = e / GO, grp = e - pix * GO, p = pix >> 4, lxp = pix & 15;
        const int s = wave * PW + p, row = s / WX, col = (s - row * WX) * 16 + lxp;
        so_l[k] = (e < PW * 16 * GO) ? pix * RS + grp * 16 : 0;
        so_yx[k] = (e < PW * 16 * GO) ? ((row << 16) | col) : (0x4000 << 16);
        so_g[k] = ((row * a.Wo + col) * a.ycs + a.yoff + nb * BN + grp * 8) * 2;
    }
    float4 bvr[NT];                                        // bias: once per workgroup
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) bvr[tt] = *(const float4*)(a.bias + nb * BN + tt * 16 + q * 4);

    // Every load and store of the loop is UNCONDITIONAL and the prefetch loads are issued from inline asm with HAND-COUNTED
    // s_waitcnt's: the compiler's own bookkeeping falls back to vmcnt(0) at the loop header, which would make every tile wait for
    // the stores and the residual issued after its halo.  Issue order per iteration:
    //     halo(t+1) [NPA loads] ... stores(t) [NSO] ... residual(t+1) [NRV loads]
    //   * halo(t+1) is consumed at the top of iteration t+1: younger operations = NSO + NRV          -> vmcnt(NSO + NRV)
    //   * residual(t) is consumed in the epilogue of iteration t: younger operations = halo(t+1)   -> vmcnt(NPA)
    // (memory operations of one wave retire in order; the named registers are tied through the waits so that no consumer can be
    // scheduled above them.)
    const bool post_relu = a.post_act == 1;                // the host only selects this kernel for pre_act = none, post_act in {none, ReLU}
    constexpr int NRV = NT * PW;
    static_assert(NPA <= 16 && NRV <= 16 && NSO <= 16, "prefetch registers");
    u32x4 pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7, pa8, pa9, pa10, pa11, pa12, pa13, pa14, pa15;
    u32x2 rv0, rv1, rv2, rv3, rv4, rv5, rv6, rv7, rv8, rv9, rv10, rv11, rv12, rv13, rv14, rv15;      // entry e = p * NT + tt
#define WS_LIST16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define WS_LD1(j)                                                                                                       \
    if constexpr (j < NPA) {                                                                                            \
        const int iy = iy0_ + (hyx[j < NPA ? j : 0] >> 16), ix = ix0_ + (hyx[j < NPA ? j : 0] & 0xFFFF);                \
        const bool ok_ = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;                                   \
        const int vo_ = ok_ ? xb_ + goff[j < NPA ? j : 0] : OOB;                                                        \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pa##j) : "v"(vo_), "s"(xrs));                     \
    }
#define WS_LOAD_HALO(N_, OY0_, OX0_)                                                                                     \
    {                                                                                                                   \
        const int iy0_ = (OY0_) - 1, ix0_ = (OX0_) - 1;                                                                 \
        const int xb_ = ((((N_) * a.H + iy0_) * a.W + ix0_) * a.xcs + a.xoff) * 2;                                      \
        WS_LIST16(WS_LD1)                                                                                               \
    }
#define WS_RS1(e)                                                                                                       \
    if constexpr (e < NRV) {                                                                                            \
        constexpr int p = (e < NRV ? e : 0) / NT, tt = (e < NRV ? e : 0) % NT;                                          \
        asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:%3" : "=v"(rv##e) : "v"(rvo_[p]), "s"(rrs), "n"(tt * 32)); \
    }
#define WS_LOAD_RES(N_, OY0_, OX0_)                                                                                      \
    {                                                                                                                   \
        const int rb_ = (((N_) * a.Ho + (OY0_)) * a.Wo + (OX0_)) * a.r1cs * 2;                                          \
        int rvo_[PW];                                                                                                   \
        _Pragma("unroll") for (int p = 0; p < PW; ++p)                                                                  \
            rvo_[p] = ((OY0_) + (r_yx[p] >> 16) < a.Ho && (OX0_) + (r_yx[p] & 0xFFFF) < a.Wo) ? rb_ + r_off[p] : OOB;   \
        WS_LIST16(WS_RS1)                                                                                               \
    }
#define WS_PIN_PA(j) if constexpr (j < NPA) asm volatile("" : "+v"(pa##j));
#define WS_PIN_RV(e) if constexpr (e < NRV) asm volatile("" : "+v"(rv##e));
#define WS_ST1(j) if constexpr (j < NPA) *(u32x4*)(lds_a + loff[j < NPA ? j : 0]) = pa##j;
#define WS_EP1(e)                                                                                                       \
    if constexpr (e < NRV) {                                                                                            \
        constexpr int p = (e < NRV ? e : 0) / NT, tt = (e < NRV ? e : 0) % NT;                                          \
        const float4 bv = bvr[tt];                                                                                      \
        float v[4] = {acc[tt][p][0] + bv.x, acc[tt][p][1] + bv.y, acc[tt][p][2] + bv.z, acc[tt][p][3] + bv.w};          \
        if constexpr (hasr_) {                                                                                          \
            const half4 rh = __builtin_bit_cast(half4, rv##e);                                                          \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = (float)rh[r] + v[r];                                   \
        }                                                                                                               \
        if constexpr (relu_) {                                                                                          \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaxf(v[r], 0.0f);                           \
        }                                                                                                               \
        half4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};                                     \
        *(half4*)(strip + (p * 16 + lx) * RS + (tt * 16 + q * 4) * 2) = o;                                              \
    }
    // the activation / residual combination is uniform: one specialised copy of the per-value code each, no test per value
#define WS_EPILOGUE(RELU_, HASR_) { constexpr bool relu_ = RELU_, hasr_ = HASR_; WS_LIST16(WS_EP1) }
    const bool has_r1 = a.r1 != nullptr;

    // the weight DMA must have landed before the loop (an LDS-DMA pending at the loop header would put vmcnt(0) in front of every LDS access)
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    int cx, cy, cn;                                        // tile cursor
    {
        const int tx0 = t_begin % a.tiles_x, r0 = t_begin / a.tiles_x;
        cn = r0 / a.tiles_y; cy = r0 - cn * a.tiles_y; cx = tx0;
        WS_LOAD_HALO(cn, cy * TH, cx * TW)
        WS_LOAD_RES(cn, cy * TH, cx * TW)
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRV));        // first tile: no stores between its halo and the loop yet
    for (int t = t_begin; t < t_end; ++t) {
        __builtin_amdgcn_s_barrier();                      // every wave has finished reading the previous tile's halo (its MFMAs consumed the reads)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSO + NRV));
        WS_LIST16(WS_PIN_PA)
        WS_LIST16(WS_ST1)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");

        const int n = cn, oy0 = cy * TH, ox0 = cx * TW;    // tile t; (cx, cy, cn) then steps to tile t+1 without a division
        if (t + 1 < t_end) {                               // (the last iteration re-reads its own tile rather than branch around the loads)
            if (++cx == a.tiles_x) { cx = 0; if (++cy == a.tiles_y) { cy = 0; ++cn; } }
        }
        const int n1 = cn, oy1 = cy * TH, ox1 = cx * TW;
        WS_LOAD_HALO(n1, oy1, ox1)                         // in flight during the MFMAs below

        f32x4 acc[NT][PW];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int p = 0; p < PW; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            half8 wa[NT], xb[PW];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) wa[tt] = *(const half8*)(lds_w + wlane + i * (4 * BN * 16) + tt * 256);
#pragma unroll
            for (int p = 0; p < PW; ++p) xb[p] = *(const half8*)(lds_a + abase[p] + koff[i]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int p = 0; p < PW; ++p)
                    acc[tt][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[tt], xb[p], acc[tt][p], 0, 0, 0);
        }
        mfma_phase_schedule<NI, NT, PW>();

        // epilogue: bias, activations and residual in the MFMA layout, fp16 into this wave's strip
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPA));
        WS_LIST16(WS_PIN_RV)
        if (post_relu) {
            if (has_r1) WS_EPILOGUE(true, true) else WS_EPILOGUE(true, false)
        } else {
            if (has_r1) WS_EPILOGUE(false, true) else WS_EPILOGUE(false, false)
        }
        // LDS operations of one wave execute in order: the strip is complete for every lane of this wave here.
        // 16 bytes per lane, BN*2-byte runs per pixel (whole pixel rows when ycs == BN).
        const int yb = ((n * a.Ho + oy0) * a.Wo + ox0) * a.ycs * 2;
#pragma unroll
        for (int k = 0; k < NSO; ++k) {
            const u32x4 v = *(const u32x4*)(strip + so_l[k]);
            const int vo = (oy0 + (so_yx[k] >> 16) < a.Ho && ox0 + (so_yx[k] & 0xFFFF) < a.Wo) ? yb + so_g[k] : OOB;
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(v), "v"(vo), "s"(yrs) : "memory");
        }
        WS_LOAD_RES(n1, oy1, ox1)                          // residual of the next tile: a whole iteration ahead of its use
    }
    asm volatile("s_waitcnt vmcnt(0)");                    // the clamped prefetches of the last iteration still target live registers
#undef WS_LOAD_HALO
#undef WS_LOAD_RES
#undef WS_LD1
#undef WS_RS1
#undef WS_ST1
#undef WS_EP1
#undef WS_EPILOGUE
#undef WS_PIN_PA
#undef WS_PIN_RV
#undef WS_LIST16
}



// ------------------------------------------------------------------------------------------------------------
// fp16 family, variants 8 / 9: "A-direct" 3x3 stride-1 kernel for Cout a multiple of 96 (HRNet's 96 / 192 / 384-channel branches).
//   * four waves = PG pixel groups x CQ Cout groups; a wave owns 4 rows x 32 pixels (8 sub-tiles) x 48 output channels: 24 MFMAs per
//     K-step of 32 from 3 weight and 8 activation fragments;
//   * WEIGHT fragments never touch LDS: the weight image is fragment-major (16 bytes per lane), so the three A fragments of a K-step are
//     three coalesced buffer_load_dwordx4 straight into registers, requested two K-steps ahead (ring of three, static slots);
//   * ACTIVATIONS: the halo tile of a 32-channel chunk by LDS-DMA into a two-deep ring that runs continuously over the workgroup's
//     items (XCD-contiguous ranges), one s_barrier per chunk (9 K-steps = 216 MFMAs per wave).  Pixel stride 96 B (4 channel groups + 2
//     pad slots) makes the ds_read_b128 of a B fragment conflict-free with ONE address register and immediates;
//   * every wave requests the same number of LDS-DMA slabs, so all waits are static vmcnt counts; out-of-image pixels and stores use the
//     descriptor's range check (offset 2^31), no divergent branches;
//   * epilogue wave-private (own LDS strip, no barrier): residual fetched as coalesced 16-byte pieces for all rows at once, through the
//     strip into the MFMA layout, results back through the strip as 16-byte stores.  Two workgroups per CU: one's epilogue runs
//     under the other's MFMAs.
// Measured (tools/convbench/ad_main.hip, B = 50): 96->96@68x120 83 us (tuned generic kernel 94), 192->192@34x60 70 us (85),
// 384->384@17x30 67 us (100).
// ------------------------------------------------------------------------------------------------------------
template <int CQ, int PG, int RES, bool S2>          // RES: number of residual operands (0, 1, 2)
__global__ __launch_bounds__(256, 2) void conv_f16_ad_kernel(ConvArgs a)
{
    static_assert(CQ * PG == 4, "four waves per workgroup");
    using rsrc_t = __amdgpu_buffer_rsrc_t;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NT = 3, PW = 8, NW = 4, BN = CQ * 48, TH = 4 * PG;
    constexpr int KSTEP = 4 * BN * 16;                   // bytes of one K-step (32 input channels of one tap) of the weight image
    // S2 (stride 2): the input is read as its space-to-depth image (pixel (Y, X) holds the four phases (2Y + ry, 2X + rx) side by side, 4 Cin
    // channels), on which the 3x3 stride-2 kernel is a 2x2 stride-1 kernel with 7 of its 16 (tap, phase) blocks zero; the re-arrangement
    // happens in the LDS-DMA addressing, the zeros are in the weight image (conv_tile_weights)
    constexpr int HW_ = S2 ? 33 : 34, HPIX = (S2 ? TH + 1 : TH + 2) * HW_, PS = 96, KPC = S2 ? 4 : 9, RING = S2 ? 4 : 3;
    constexpr int HSLABS = ((HPIX * PS + 1023) / 1024 + NW - 1) / NW * NW, HB = HSLABS * 1024, HK = HSLABS / NW;
    constexpr int RS = 48 * 2 + 16, STRIP = 32 * RS;     // one output row of the wave (32 pixels x 48 channels) per pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Hb = smem;                                // 2 x HB
    char* const strip = smem + 2 * HB + (threadIdx.x >> 6) * STRIP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cq = wave % CQ, pg = wave / CQ, q = lane >> 4, lx = lane & 15;
    const int gy = a.gy, nitems = a.tiles_x * a.tiles_y * a.N * gy, nwg = gridDim.x;
    int item0, item_end;
    {
        const int b = blockIdx.x, xcd = b & 7, k = b >> 3;
        const int wgs_here = (nwg + 7 - xcd) >> 3;
        const int qn = nitems >> 3, rn = nitems & 7;
        const int x0 = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        const int xc = qn + (xcd < rn ? 1 : 0);
        item0 = x0 + (int)((long)xc * k / wgs_here);
        item_end = x0 + (int)((long)xc * (k + 1) / wgs_here);
    }
    if (item0 >= item_end) return;
    const int nloc = item_end - item0, nch = a.nchunks, GC = nloc * nch;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.r1, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t rrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.r2, 0, 0x7FFFFFFF, 0x00020000);
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, 0x7FFFFFFF, 0x00020000);
    int hpk[HK];                                          // per slab: halo row | column << 8 | channel slot << 16, or -1 (padding)
#pragma unroll
    for (int k = 0; k < HK; ++k) {
        const int e = (wave + NW * k) * 64 + lane;
        const int pix = e / 6, slot = e - pix * 6;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        hpk[k] = (pix < HPIX && slot < 4) ? (hy | (hx << 8) | (slot << 16)) : -1;
    }
    const int bbase = ((pg * 4) * HW_ + lx) * PS + q * 16;      // B fragments: one address register, everything else is an immediate
    const unsigned wlane = (unsigned)((q * BN + cq * 48 + lx) * 16);

    auto decode = [&](int item, int& n, int& ty, int& tx, int& nb) {
        int t = item / gy; nb = item - t * gy;
        tx = t % a.tiles_x; t /= a.tiles_x;
        ty = t % a.tiles_y; n = t / a.tiles_y;
    };
    // ---- halo ring: global chunk index = local item * nch + chunk; slot = index & 1 -------------------------------------------
    int h_g = 0, h_ch = 0, h_item = item0;
    int h_iy0, h_ix0, h_gb;                                // geometry of the item the next request belongs to (scalars)
    const int cin_s2 = nch * 8;                            // S2: real input channels (the space-to-depth image has 4 * Cin = 32 * nch)
    auto halo_origin = [&](int item) {
        int n, ty, tx, nb; decode(item, n, ty, tx, nb);
        h_iy0 = ty * TH - 1; h_ix0 = tx * 32 - 1;
        h_gb = S2 ? n * a.H : (((n * a.H + h_iy0) * a.W + h_ix0) * a.xcs + a.xoff) * 2;
    };
    halo_origin(h_item);
    auto issue_h = [&]() {
        char* dst = Hb + (h_g & 1) * HB;                  // (past the last chunk: the same requests again, harmlessly, so that the count stays static)
        const unsigned so = S2 ? 0u : (unsigned)(h_ch * 32) * 2u;
        const int cb = h_ch * 32, ph0 = S2 ? cb / cin_s2 : 0, r0 = cb - ph0 * cin_s2;      // S2: first phase / channel of this 32-channel chunk (a chunk spans at most two phases)
#pragma unroll
        for (int k = 0; k < HK; ++k) {
            const int hy = hpk[k] & 0xFF, hx = (hpk[k] >> 8) & 0xFF, slot = (hpk[k] >> 16) & 7;
            unsigned off;
            if (S2) {
                const int t = r0 + slot * 8, over = t >= cin_s2 ? 1 : 0, ph = ph0 + over, co = t - over * cin_s2;
                const int iy = 2 * (h_iy0 + hy) + (ph >> 1), ix = 2 * (h_ix0 + hx) + (ph & 1);
                off = (hpk[k] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                          ? (unsigned)((((h_gb + iy) * a.W + ix) * a.xcs + a.xoff + co) * 2) : OOB;
            } else {
                const int iy = h_iy0 + hy, ix = h_ix0 + hx;
                off = (hpk[k] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                          ? (unsigned)(h_gb + ((hy * a.W + hx) * a.xcs + slot * 8) * 2) : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(dst + (wave + NW * k) * 1024), 16, off, so, 0, 0);
        }
        if (h_g + 1 >= GC) return;
        ++h_g;
        if (++h_ch == nch) { h_ch = 0; ++h_item; halo_origin(h_item); }
    };
    issue_h();
    int gc = 0;
    for (int item = item0; item < item_end; ++item) {
        int n, ty, tx, nb; decode(item, n, ty, tx, nb);
        const int oy0 = ty * TH, ox0 = tx * 32;
        f32x4 acc[NT][PW];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int p = 0; p < PW; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A ring: fragments of K-steps kk, kk + 1, kk + 2 (slot = kk % RING; KPC K-steps per chunk keep the slots static)
        u32x4 A[RING][NT];
        unsigned wsrc = (unsigned)(nb * nch * KPC * 4) * (unsigned)(BN * 16);     // K-step 0 of this item
        const unsigned wend = wsrc + (unsigned)(nch * KPC) * KSTEP;
        auto load_a = [&](int slot) {                      // requests the next K-step of the item (the last two requests of an item repeat its last K-step)
            const unsigned so = wsrc < wend ? wsrc : wend - KSTEP;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) A[slot][tt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + tt * 256, so, 0);
            wsrc += KSTEP;
        };
        load_a(0); load_a(1);
        for (int ch = 0; ch < nch; ++ch, ++gc) {
            // chunk gc has landed (requested one chunk ago); every wave is done with the other slot
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // everything older than the two K-steps of A in flight
            __builtin_amdgcn_s_barrier();
            issue_h();                                              // chunk gc + 1 into the slot chunk gc - 1 used
            const char* hb = Hb + (gc & 1) * HB + bbase;
            half8 B[2][PW];
            auto read_b = [&](int slot, int kk) {
                const int ky = S2 ? kk >> 1 : kk / 3, kx = S2 ? kk & 1 : kk - ky * 3;
#pragma unroll
                for (int p = 0; p < PW; ++p)
                    B[slot][p] = *(const half8*)(hb + (((p >> 1) + ky) * HW_ + (p & 1) * 16 + kx) * PS);
            };
            read_b(0, 0);
#pragma unroll
            for (int kk = 0; kk < KPC; ++kk) {
                load_a((kk + 2) % RING);
                if (kk + 1 < KPC) read_b((kk + 1) & 1, kk + 1);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int p = 0; p < PW; ++p)
                        acc[tt][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16((half8)A[kk % RING][tt], B[kk & 1][p], acc[tt][p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int co0 = nb * BN + cq * 48;
        __builtin_amdgcn_sched_barrier(0);
        float4 bias[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) bias[tt] = *(const float4*)(a.bias + co0 + tt * 16 + q * 4);
        // piece e = i * 64 + lane of a row: pixel e / 6, 16-byte group e % 6.  Offsets are recomputed where they are used (registers are
        // scarce here); out-of-image pieces get the out-of-range offset (loads return zeros, stores are dropped)
        int pstrip[3], ppx[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = i * 64 + lane, px = e / 6, gq = e - px * 6;
            pstrip[i] = px * RS + gq * 16;
            ppx[i] = px | (gq << 8);
        }
        auto piece_off = [&](int r2, int i, int cs, int off) -> unsigned {
            const int oy = oy0 + pg * 4 + r2, ox2 = ox0 + (ppx[i] & 0xFF), gq = ppx[i] >> 8;
            return (oy < a.Ho && ox2 < a.Wo) ? (unsigned)((((n * a.Ho + oy) * a.Wo + ox2) * cs + off + co0 + gq * 8) * 2) : OOB;
        };
        u32x4 rres[4][3], rres2[4][3];
        if (RES >= 1) {
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
                for (int i = 0; i < 3; ++i) rres[r2][i] = __builtin_amdgcn_raw_buffer_load_b128(rrs, piece_off(r2, i, a.r1cs, a.r1off), 0, 0);
        }
        if (RES >= 2) {
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
                for (int i = 0; i < 3; ++i) rres2[r2][i] = __builtin_amdgcn_raw_buffer_load_b128(rrs2, piece_off(r2, i, a.r2cs, a.r2off), 0, 0);
        }
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
            if (RES >= 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i) *(u32x4*)(strip + pstrip[i]) = rres[r2][i];
            }
            float v[2][NT][4];
#pragma unroll
            for (int xb2 = 0; xb2 < 2; ++xb2) {
                const int p = r2 * 2 + xb2;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const char* sp = strip + (xb2 * 16 + lx) * RS + (tt * 16 + q * 4) * 2;
                    v[xb2][tt][0] = acc[tt][p][0] + bias[tt].x; v[xb2][tt][1] = acc[tt][p][1] + bias[tt].y;
                    v[xb2][tt][2] = acc[tt][p][2] + bias[tt].z; v[xb2][tt][3] = acc[tt][p][3] + bias[tt].w;
                    if (RES >= 1) {
                        const half4 rv = *(const half4*)sp;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[xb2][tt][r] = (float)rv[r] + v[xb2][tt][r];
                    }
                }
            }
            if (RES >= 2) {                                 // second residual through the same strip (LDS operations of a wave execute in order)
#pragma unroll
                for (int i = 0; i < 3; ++i) *(u32x4*)(strip + pstrip[i]) = rres2[r2][i];
#pragma unroll
                for (int xb2 = 0; xb2 < 2; ++xb2)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const half4 rv = *(const half4*)(strip + (xb2 * 16 + lx) * RS + (tt * 16 + q * 4) * 2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[xb2][tt][r] = v[xb2][tt][r] + (float)rv[r];
                    }
            }
#pragma unroll
            for (int xb2 = 0; xb2 < 2; ++xb2)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    if (a.post_act == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[xb2][tt][r] = v[xb2][tt][r] > 0.f ? v[xb2][tt][r] : 0.f;
                    }
                    half4 o = {(_Float16)v[xb2][tt][0], (_Float16)v[xb2][tt][1], (_Float16)v[xb2][tt][2], (_Float16)v[xb2][tt][3]};
                    *(half4*)(strip + (xb2 * 16 + lx) * RS + (tt * 16 + q * 4) * 2) = o;
                }
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)(strip + pstrip[i]), yrs, piece_off(r2, i, a.ycs, a.yoff), 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 exact family.  KC = 16 (or 4 for the 3-channel stems); canonical K order.
// ------------------------------------------------------------------------------------------------------------
template <int KS, int S, int KC, int NT>
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvArgs a)
{
    constexpr int PSF = KC + 1;                 // LDS pixel stride in floats (odd)
    constexpr int TAPS = KS * KS;
    constexpr int CSTEPS = KC / 4;
    constexpr int NI = TAPS * CSTEPS;
    constexpr int BN = NT * 16;
    constexpr int WFLOATS = NI * 4 * BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds_w = (float*)smem;
    float* lds_a = lds_w + WFLOATS;

    const int WX = a.wx, TH = 16 / WX, TW = 16 * WX;
    const int halo_w = (TW - 1) * S + KS, halo_h = (TH - 1) * S + KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, lx = lane & 15;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int nb = blockIdx.y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - KS / 2, ix0 = ox0 * S - KS / 2;

    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    int abase[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int s = wave * 4 + p, row = s / WX, xb = s - row * WX;
        abase[p] = ((row * S) * halo_w + (xb * 16 + lx) * S) * PSF + q;
    }
    const int wlane = q * BN + lx;
    const float* xg = (const float*)a.x;
    const int nelem = halo_h * halo_w * CSTEPS;   // float4 groups

    for (int ch = 0; ch < a.nchunks; ++ch) {
        {
            const float* wsrc = (const float*)a.w + (size_t)(nb * a.nchunks + ch) * WFLOATS;
            for (int o = tid * 4; o < WFLOATS; o += 256 * 4) *(float4*)(lds_w + o) = *(const float4*)(wsrc + o);
        }
        const int c0 = a.xoff + ch * KC;
        for (int idx = tid; idx < nelem; idx += 256) {
            const int pix = idx / CSTEPS, g = idx - pix * CSTEPS;
            const int hy = pix / halo_w, hx = pix - hy * halo_w;
            const int iy = iy0 + hy, ix = ix0 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v = *(const float4*)(xg + ((size_t)(n * a.H + iy) * a.W + ix) * a.xcs + c0 + g * 4);
            float* d = lds_a + pix * PSF + g * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const int toff = (ky * halo_w + kx) * PSF;
#pragma unroll
            for (int cs = 0; cs < CSTEPS; ++cs) {
                const int i = tap * CSTEPS + cs;
                float wa[NT], xb[4];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) wa[tt] = lds_w[wlane + i * (4 * BN) + tt * 16];
#pragma unroll
                for (int p = 0; p < 4; ++p) xb[p] = lds_a[abase[p] + toff + cs * 4];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[tt][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tt], xb[p], acc[tt][p], 0, 0, 0);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int s = wave * 4 + p, row = s / WX, xb = s - row * WX;
        const int oy = oy0 + row, ox = ox0 + xb * 16 + lx;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const size_t pidx = (size_t)(n * a.Ho + oy) * a.Wo + ox;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int co = nb * BN + tt * 16 + q * 4;
            const float4 bv = *(const float4*)(a.bias + co);
            float v[4] = {acc[tt][p][0] + bv.x, acc[tt][p][1] + bv.y, acc[tt][p][2] + bv.z, acc[tt][p][3] + bv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = d_act(v[r], a.pre_act);
            if (a.r1) {
                const float4 rv = *(const float4*)((const float*)a.r1 + pidx * a.r1cs + a.r1off + co);
                v[0] = rv.x + v[0]; v[1] = rv.y + v[1]; v[2] = rv.z + v[2]; v[3] = rv.w + v[3];
            }
            if (a.r2) {
                const float4 rv = *(const float4*)((const float*)a.r2 + pidx * a.r2cs + a.r2off + co);
                v[0] = v[0] + rv.x; v[1] = v[1] + rv.y; v[2] = v[2] + rv.z; v[3] = v[3] + rv.w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = d_act(v[r], a.post_act);
            *(float4*)((float*)a.y + pidx * a.ycs + a.yoff + co) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side: geometry, instance table, weight tiling
// ------------------------------------------------------------------------------------------------------------
static int f16_ps(int kc) { const int g = kc / 8; return kc * 2 + ((g % 2 == 0) ? 16 : 0); }
static int f16_ni(int ks, int kc) { return (ks * ks * (kc / 8) + 3) / 4; }

static int conv_pw(const ConvConfig& c) { return (c.variant == 3 || c.variant == 7) ? 2 : (c.variant == 4 ? 1 : 4); }
static bool conv_ws(const ConvConfig& c) { return c.variant == 6 || c.variant == 7; }
static bool conv_ad(const ConvConfig& c) { return c.variant >= 8 && c.variant <= 11; }      // A-direct: 8 = 4 Cout groups x 1 pixel group (BN 192), 9 = 2 x 2 (BN 96); 10 / 11 = the same for stride 2
static bool conv_ad_wide(const ConvConfig& c) { return c.variant == 8 || c.variant == 10; }

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
    return (size_t)c.ks * c.ks * (c.kc / 4) * 4 * bn * 4 + (size_t)hh * hw * (c.kc + 1) * 4;
}

size_t conv_lds_bytes(int precision, const ConvConfig& c) { return lds_bytes(precision, c); }
int conv_tiles_per_frame(const ConvConfig& c, int ho, int wo)
{
    if (conv_ad(c)) return ((wo + 31) / 32) * ((ho + (conv_ad_wide(c) ? 4 : 8) - 1) / (conv_ad_wide(c) ? 4 : 8));
    const int th = 4 * conv_pw(c) / c.wx, tw = 16 * c.wx;
    return ((wo + tw - 1) / tw) * ((ho + th - 1) / th);
}

typedef void (*ConvKernel)(ConvArgs);
struct Inst { int prec, ks, s, kc, nt, variant; ConvKernel fn; };

// (The LDS-DMA pipeline variants 1 / 5 lost to plain occupancy on every measured layer; they live in tools/convbench/conv_exp.hip.)
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
static const Inst g_inst[] = {
    // 3x3 stride 1
    ALLNT16(3, 1, 16), ALLNT16(3, 1, 32), ALLNT16(3, 1, 48), ALLNT16(3, 1, 64),
    // 3x3 stride 2 (halo is 4x larger: small chunks)
    ALLNT16(3, 2, 8), ALLNT16(3, 2, 16), ALLNT16(3, 2, 32), ALLNT16(3, 2, 48),
    // 1x1
    ALLNT16(1, 1, 16), ALLNT16(1, 1, 32), ALLNT16(1, 1, 48), ALLNT16(1, 1, 64),
    // chunk-pipelined staging (variant 2): the next chunk's loads are issued before this chunk's MFMAs; small chunks only
    ALLNT16P(3, 1, 16), ALLNT16P(3, 1, 32), ALLNT16P(1, 1, 32), ALLNT16P(1, 1, 64),
    // half-size tiles (variant 3): 2 pixel sub-tiles per wave -> fewer registers / less LDS -> more resident workgroups
    ALLNT16H(3, 1, 16), ALLNT16H(3, 1, 32), ALLNT16H(3, 1, 48), ALLNT16H(3, 1, 64), ALLNT16H(3, 2, 8), ALLNT16H(3, 2, 16), ALLNT16H(3, 2, 32), ALLNT16H(1, 1, 16), ALLNT16H(1, 1, 32), ALLNT16H(1, 1, 48), ALLNT16H(1, 1, 64),
    // quarter-size tiles (variant 4): 1 pixel sub-tile per wave
    ALLNT16Q(3, 1, 16), ALLNT16Q(3, 1, 32), ALLNT16Q(3, 1, 48), ALLNT16Q(3, 1, 64), ALLNT16Q(3, 2, 16), ALLNT16Q(3, 2, 32), ALLNT16Q(1, 1, 32), ALLNT16Q(1, 1, 64),
    // weight-stationary persistent 3x3 stride-1 kernels (variant 6: 4 pixel sub-tiles per wave, 7: 2); kc = Cin
    {EAGLE_PREC_F16, 3, 1, 48, 3, 6, conv_f16_ws_kernel<48, 3, 4, 2>}, {EAGLE_PREC_F16, 3, 1, 48, 3, 7, conv_f16_ws_kernel<48, 3, 2, 2>},
    {EAGLE_PREC_F16, 3, 1, 64, 4, 6, conv_f16_ws_kernel<64, 4, 4, 1>}, {EAGLE_PREC_F16, 3, 1, 64, 4, 7, conv_f16_ws_kernel<64, 4, 2, 1>},
    {EAGLE_PREC_F16, 3, 1, 96, 3, 6, conv_f16_ws_kernel<96, 3, 4, 1>}, {EAGLE_PREC_F16, 3, 1, 96, 3, 7, conv_f16_ws_kernel<96, 3, 2, 1>},
    {EAGLE_PREC_F16, 3, 1, 96, 2, 7, conv_f16_ws_kernel<96, 2, 2, 1>},
    // A-direct 3x3 stride-1 kernels (variant 8: BN = 192, tile 4 x 32; variant 9: BN = 96, tile 8 x 32); kc = 32
    {EAGLE_PREC_F16, 3, 1, 32, 12, 8, conv_f16_ad_kernel<4, 1, 1, false>}, {EAGLE_PREC_F16, 3, 1, 32, 6, 9, conv_f16_ad_kernel<2, 2, 1, false>},
    {EAGLE_PREC_F16, 3, 2, 32, 12, 10, conv_f16_ad_kernel<4, 1, 1, true>}, {EAGLE_PREC_F16, 3, 2, 32, 6, 11, conv_f16_ad_kernel<2, 2, 1, true>},
    // exact fp32 family
    ALLNT32(3, 1, 16), ALLNT32(3, 2, 16), ALLNT32(3, 2, 4), ALLNT32(1, 1, 16),
};

static const Inst* find_inst(int precision, const ConvConfig& c)
{
    for (const Inst& i : g_inst)
        if (i.prec == precision && i.ks == c.ks && i.s == c.stride && i.kc == c.kc && i.nt == c.nt && i.variant == c.variant) return &i;
    return nullptr;
}
bool conv_supported(int precision, const ConvConfig& c) { return find_inst(precision, c) != nullptr; }

struct Tuned { int ks, s, cin, cout, wo, kc, nt, wx, variant; };
static const Tuned g_tuned[] = {
#include "conv_tuned.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0}};

ConvConfig conv_choose(int precision, int ks, int stride, int cin_pad, int cout_pad, int wo, bool plain_epilogue, bool second_residual)
{
    // plain_epilogue: no activation before the residual adds, none / ReLU after them, fp16 output.  The weight-stationary kernels also need
    // at most one residual; the A-direct kernels take two.
    const bool plain_one = plain_epilogue && !second_residual;
    ConvConfig c;
    c.ks = ks; c.stride = stride; c.cin = cin_pad; c.cout_pad = cout_pad;
    c.wx = (wo > 16) ? 2 : 1;
    static const int nts[] = {6, 4, 3, 2, 1};
    if (precision == EAGLE_PREC_F32) {
        c.nt = 1;
        for (int nt : nts)
            if (cout_pad % (16 * nt) == 0) { c.nt = nt; break; }
        c.kc = (cin_pad < 16) ? 4 : 16;
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
    return (size_t)nblk * nch * c.ks * c.ks * (c.kc / 4) * 4 * bn;
}

void conv_tile_weights(int precision, const ConvConfig& c, const float* w, int cin_real, int cout_real, void* dst)
{
    const int bn = c.nt * 16, nblk = c.cout_pad / bn, nch = c.cin / c.kc, taps = c.ks * c.ks;
    auto W = [&](int tap, int ci, int co) -> float {
        return (ci < cin_real && co < cout_real) ? w[((size_t)tap * cin_real + ci) * cout_real + co] : 0.0f;
    };
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
    if (!a.r1 && a.r2) { a.r1 = a.r2; a.r1cs = a.r2cs; a.r1off = a.r2off; a.r2 = nullptr; }       // a single residual is always operand 1 (IEEE addition is commutative: same bits)
    a.pre_act = L.pre_act; a.post_act = L.post_act; a.out_f32 = L.out_f32 || precision == EAGLE_PREC_F32;
    a.wx = c.wx;
    const int th = 4 * conv_pw(c) / c.wx, tw = 16 * c.wx;
    a.tiles_x = (a.Wo + tw - 1) / tw; a.tiles_y = (a.Ho + th - 1) / th;
    a.nchunks = c.cin / c.kc;
    a.zeros = conv_zero_page(); a.trash = conv_trash_page(); a.xcd = 0; a.gy = 1;
    a.am = L.am_slot ? *L.am_slot : nullptr; a.am_cs = c.cout_pad;
    if (conv_ad(c)) {                                       // A-direct: persistent over XCD-contiguous item ranges, two workgroups per CU
        if (a.out_f32 || a.pre_act != 0 || a.post_act > 1 || L.am_slot || c.kc != 32 || c.ks != 3 || c.stride != (c.variant >= 10 ? 2 : 1))
            fail(EAGLE_E_NOKERNEL, "A-direct conv needs 3x3, kc = 32, fp16 output, pre_act none, post_act in {none, ReLU}");
        if (c.stride == 2) a.nchunks = 4 * c.cin / 32;      // chunks of the space-to-depth image
        const int thh = conv_ad_wide(c) ? 4 : 8;
        a.tiles_x = (a.Wo + 31) / 32; a.tiles_y = (a.Ho + thh - 1) / thh;
        a.gy = c.cout_pad / (c.nt * 16);
        const size_t lim = (size_t)1 << 31;
        if ((size_t)a.N * a.H * a.W * a.xcs * 2 >= lim || (size_t)a.N * a.Ho * a.Wo * std::max(std::max(a.ycs, a.r1 ? a.r1cs : 0), a.r2 ? a.r2cs : 0) * 2 >= lim)
            fail(EAGLE_E_INVALID, "fp16 conv: a tensor of %d frames reaches 2 GiB; use a smaller device batch", a.N);
        const int nres = (a.r1 ? 1 : 0) + (a.r2 ? 1 : 0);
        static const ConvKernel ad_fn[4][3] = {
            {conv_f16_ad_kernel<4, 1, 0, false>, conv_f16_ad_kernel<4, 1, 1, false>, conv_f16_ad_kernel<4, 1, 2, false>},
            {conv_f16_ad_kernel<2, 2, 0, false>, conv_f16_ad_kernel<2, 2, 1, false>, conv_f16_ad_kernel<2, 2, 2, false>},
            {conv_f16_ad_kernel<4, 1, 0, true>, conv_f16_ad_kernel<4, 1, 1, true>, conv_f16_ad_kernel<4, 1, 2, true>},
            {conv_f16_ad_kernel<2, 2, 0, true>, conv_f16_ad_kernel<2, 2, 1, true>, conv_f16_ad_kernel<2, 2, 2, true>}};
        const ConvKernel fn = ad_fn[c.variant - 8][nres];
        ensure_max_dynamic_lds((const void*)fn, 160 * 1024);
        const int items = a.tiles_x * a.tiles_y * a.N * a.gy;
        static const int slots = getenv("EAGLE_CONV_AD_SLOTS") ? atoi(getenv("EAGLE_CONV_AD_SLOTS")) : 512;      // developer knob: resident workgroups (co-residency experiments)
        hipLaunchKernelGGL(fn, dim3(std::min(items, slots)), dim3(256), lds_bytes(precision, c), s, a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (a.am && (precision != EAGLE_PREC_F16 || conv_ws(c)))
        fail(EAGLE_E_NOKERNEL, "fused heat-map maxima need the generic fp16 kernel");
    if (precision == EAGLE_PREC_F16) {                      // the fp16 kernels address tensors through raw buffer descriptors with 32-bit byte offsets
        const size_t lim = (size_t)1 << 31;
        const size_t cs_out = std::max(std::max(a.ycs, a.r1 ? a.r1cs : 0), a.r2 ? a.r2cs : 0);
        if ((size_t)a.N * a.H * a.W * a.xcs * 2 >= lim || (size_t)a.N * a.Ho * a.Wo * cs_out * 2 >= lim)
            fail(EAGLE_E_INVALID, "fp16 conv: a tensor of %d frames reaches 2 GiB; use a smaller device batch", a.N);
    }
    const size_t lds = lds_bytes(precision, c);
    ensure_max_dynamic_lds((const void*)inst->fn, 160 * 1024);
    const int gy = c.cout_pad / (c.nt * 16);
    int gx = a.tiles_x * a.tiles_y * a.N;
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
    a.xcd = (precision == EAGLE_PREC_F16 && !conv_ws(c) && (c.ks == 3 || (c.ks == 1 && gy > 1)) && xcd_env) ? 1 : 0;
    dim3 grid(gx, gy);
    if (a.xcd) grid = dim3(gx * gy, 1);
    if (conv_ws(c)) grid = dim3(gx, 1);
    hipLaunchKernelGGL(inst->fn, grid, dim3(256), lds, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
