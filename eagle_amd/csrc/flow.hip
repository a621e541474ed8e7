// Optical-flow key-point propagation: the image side of the reference's calculate_optical_flow
// (eagle/models/coordinate_model.py:419-478) and of the gray conversion at cm.py:280.
//   K11  gray_kernel / pyrdown_kernel : BGR u8 -> gray u8 (cv2 BGR2GRAY, 15-bit fixed point) and the two cv2.pyrDown levels
//        (5x5 binomial, (sum+128)>>8, BORDER_REFLECT_101) of EVERY frame of the clip, once, in parallel.  HBM-bound:
//        2,764,800 B read + 921,600 + 230,400 + 57,600 B written per 1280x720 frame.
//   K12  lk_kernel : cv2.calcOpticalFlowPyrLK(winSize 15x15, maxLevel 2, 10 iterations / eps 0.03), one workgroup per
//        key-point, one thread per window pixel.  The reference computes Scharr derivative images of the whole frame
//        at every level; here only the 16x16 neighbourhood of each of the <= 57 tracked points is differentiated, in LDS.
// Integer arithmetic is OpenCV's (W_BITS 14 bilinear weights, CV_DESCALE); the window sums are exact 64-bit integer
// sums converted to float once, exactly as oracle/eo_flow.c does (see its header), so results are bit-identical to it.
#include <string>

#include "common.h"

namespace eagle {

// ---- K11 ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, size_t npix4, size_t npix)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix4; i += stride) {
        const uint32_t* p = (const uint32_t*)(bgr + i * 12);
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        const uint8_t b[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24), (uint8_t)w1, (uint8_t)(w1 >> 8),
                               (uint8_t)(w1 >> 16), (uint8_t)(w1 >> 24), (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o |= (uint32_t)((b[3 * k] * 3735 + b[3 * k + 1] * 19235 + b[3 * k + 2] * 9798 + (1 << 14)) >> 15) << (8 * k);
        *(uint32_t*)(gray + i * 4) = o;
    }
    // tail (frame sizes whose pixel count is not a multiple of 4)
    for (size_t i = npix4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride)
        gray[i] = (uint8_t)((bgr[3 * i] * 3735 + bgr[3 * i + 1] * 19235 + bgr[3 * i + 2] * 9798 + (1 << 14)) >> 15);
}

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void pyrdown_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int h, int w)
{
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    const size_t total = (size_t)n * dh * dw, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % dw), y = (int)((i / dw) % dh), f = (int)(i / ((size_t)dw * dh));
        const uint8_t* s = src + (size_t)f * h * w;
        int xs[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) xs[k] = reflect101(2 * x - 2 + k, w);
        int rows[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint8_t* r = s + (size_t)reflect101(2 * y - 2 + k, h) * w;
            rows[k] = r[xs[0]] + r[xs[4]] + 4 * (r[xs[1]] + r[xs[3]]) + 6 * r[xs[2]];
        }
        dst[i] = (uint8_t)((rows[0] + rows[4] + 4 * (rows[1] + rows[3]) + 6 * rows[2] + 128) >> 8);
    }
}

void gray_pyramid_launch(const uint8_t* d_bgr, int n, int h, int w, uint8_t* g0, uint8_t* g1, uint8_t* g2, hipStream_t s)
{
    const size_t npix = (size_t)n * h * w;
    hipLaunchKernelGGL(gray_kernel, dim3(256 * 16), dim3(256), 0, s, d_bgr, g0, npix / 4, npix);
    const int h1 = (h + 1) / 2, w1 = (w + 1) / 2;
    hipLaunchKernelGGL(pyrdown_kernel, dim3(256 * 8), dim3(256), 0, s, g0, g1, n, h, w);
    hipLaunchKernelGGL(pyrdown_kernel, dim3(256 * 4), dim3(256), 0, s, g1, g2, n, h1, w1);
    HIP_CHECK(hipGetLastError());
}

// ---- K12 ----------------------------------------------------------------------------------------------------------
#define LK_WIN 15
#define LK_WBITS 14
#define LK_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

struct LkImg { const uint8_t* p; int h, w; };
__device__ __forceinline__ int lk_px(const LkImg& im, int y, int x) { return im.p[(size_t)reflect101(y, im.h) * im.w + reflect101(x, im.w)]; }

// exact workgroup sum of three values per thread (T / 64 waves); every thread receives the totals.  A wave's partial sums fit 32 bits
// (|diff * derivative| <= 8160 * 4080 per pixel, 64 pixels per wave: 2.13e9 < 2^31; the one-wave form adds four pixels per lane in
// 64 bits first and splits the value), so the cross-lane steps move 32-bit values and only the per-wave partials are added in 64 bits.
__device__ __forceinline__ int wave_sum32(int v)
{
    // wave reduction on the DPP path (row shifts inside 16 lanes, then row broadcasts): the total ends up in lane 63
#define LK_DPP_STEP(CTRL_, ROWM_, BANKM_) v += __builtin_amdgcn_update_dpp(0, v, CTRL_, ROWM_, BANKM_, false);
    LK_DPP_STEP(0x111, 0xf, 0xf)                         // row_shr:1
    LK_DPP_STEP(0x112, 0xf, 0xf)                         // row_shr:2
    LK_DPP_STEP(0x114, 0xf, 0xe)                         // row_shr:4
    LK_DPP_STEP(0x118, 0xf, 0xc)                         // row_shr:8
    LK_DPP_STEP(0x142, 0xa, 0xf)                         // row_bcast:15
    LK_DPP_STEP(0x143, 0xc, 0xf)                         // row_bcast:31
#undef LK_DPP_STEP
    return v;
}
template <int T>
__device__ __forceinline__ void block_sum3(long long& a, long long& b, long long& c, long long (*red)[3], int tid)
{
    if constexpr (T == 64) {
        // one wave: no LDS, no barrier.  A lane's four-pixel sum can exceed 32 bits only by a few bits: reduce low and high halves
        // separately (low half as 16-bit pieces so that 64 of them cannot overflow)
        long long v[3] = {a, b, c};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long u = (unsigned long long)v[k];
            const int p0 = wave_sum32((int)(u & 0xFFFF)), p1 = wave_sum32((int)((u >> 16) & 0xFFFF)), p2 = wave_sum32((int)(long long)(v[k] >> 32));
            const long long s0 = __builtin_amdgcn_readlane(p0, 63), s1 = __builtin_amdgcn_readlane(p1, 63), s2 = __builtin_amdgcn_readlane(p2, 63);
            v[k] = s0 + (s1 << 16) + (s2 << 32);
        }
        a = v[0]; b = v[1]; c = v[2];
    } else {
        const int a32 = wave_sum32((int)a), b32 = wave_sum32((int)b), c32 = wave_sum32((int)c);
        __syncthreads();                                     // the previous round's readers are done with `red`
        if ((tid & 63) == 63) { red[tid >> 6][0] = a32; red[tid >> 6][1] = b32; red[tid >> 6][2] = c32; }
        __syncthreads();
        a = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        c = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    }
}

__device__ __forceinline__ bool lk_far(float x, float y) { return !(fabsf(x) < 1e8f && fabsf(y) < 1e8f); }

// Developer diagnostics of K12 (eagle_debug, tools/probe_lk_concurrency.py): DBG bit 0 = per-level / per-iteration trace of the
// quantities every thread agrees on, bit 1 = guard words around the LDS arrays (poisoned at entry, checked at exit).
struct LkDebug { long long* trace; int* counters; };
#define LK_TRACE_SLOTS 12                         // per level: slot 0 = level header, 1..10 = iterations, 11 = level trailer
#define LK_TRACE_WORDS 8

struct LkArgs {
    ClipView cv; int src_frame, dst_frame;
    ChainState* st;
    const MemList* mem; int kint;          // chain mode (mem != nullptr): the step decides itself whether frame dst_frame needs a flow
    int max_count; double eps2;
    LkDebug dbg;
};

#define LK_R 8                                   // J is cached this many pixels around the start window; beyond it the loop reads global memory
#define LK_PJ (LK_WIN + 1 + 2 * LK_R)
#define LK_GUARD 0x5AFEC0DE
struct LkLds {
    int g0[16];
    int patch[18 * 18];
    int g1[16];
    unsigned char jpatch[LK_PJ * LK_PJ];
    int g2[16];
    int sdx[16 * 16];
    int g3[16];
    int sdy[16 * 16];
    int g4[16];
    long long red[4][3];
    int g5[16];
    unsigned chk[8];
    long long trc[3 * LK_TRACE_SLOTS * LK_TRACE_WORDS];   // DBG bit 0: written by thread 0, dumped to global memory at the end of the kernel
};
// T = threads per key-point: 256 (one window pixel per thread, 4 waves) or 64 (one wave, 4 window pixels per lane: no cross-wave exchange)
template <int T, int DBG>
__device__ __forceinline__ void lk_body(const LkArgs& a)
{
    constexpr int PP = 256 / T;
    __shared__ LkLds L;
    int (&patch)[18 * 18] = L.patch;
    unsigned char (&jpatch)[LK_PJ * LK_PJ] = L.jpatch;
    int (&sdx)[16 * 16] = L.sdx;
    int (&sdy)[16 * 16] = L.sdy;
    long long (*red)[3] = L.red;
    ChainState* st = a.st;
    const int tid = threadIdx.x, pt = blockIdx.x;
    auto px = [](const LkImg& im, int y, int x) -> int {               // DBG bit 3: gray loads bypass the CU's L1 (nt)
        if constexpr ((DBG & 8) != 0) return (int)__builtin_nontemporal_load(im.p + (size_t)reflect101(y, im.h) * im.w + reflect101(x, im.w));
        else return lk_px(im, y, x);
    };
    if (st->stalled >= 0 || st->error) return;
    if (a.mem) {                                        // cm.py:282-322: flow on unscheduled frames, and on scheduled ones that detected < 4 key-points
        const int i = a.dst_frame;
        const bool scheduled = i == 0 || i % a.kint == 0;
        const bool need = !scheduled || (i > 0 && a.mem[i].n >= 0 && a.mem[i].n < 4);
        if (pt == 0 && tid == 0) st->lk_valid = need ? 1 : 0;
        if (!need) return;
    } else if (pt == 0 && tid == 0) st->lk_valid = 1;
    const int n = st->n_prev;
    if (pt == 0 && tid == 0) st->lk_n = n;
    if (pt >= n) return;
    if constexpr ((DBG & 2) != 0) {
        if (tid < 16) { L.g0[tid] = LK_GUARD; L.g1[tid] = LK_GUARD; L.g2[tid] = LK_GUARD; L.g3[tid] = LK_GUARD; L.g4[tid] = LK_GUARD; L.g5[tid] = LK_GUARD; }
        for (int e = tid; e < 18 * 18; e += T) patch[e] = 0x7FFFFFFF;
        for (int e = tid; e < LK_PJ * LK_PJ; e += T) jpatch[e] = 0xFF;
        for (int e = tid; e < 256; e += T) { sdx[e] = 0x7FFFFFFF; sdy[e] = 0x7FFFFFFF; }
        __syncthreads();
    }
    long long* tr = nullptr;
    if constexpr ((DBG & 1) != 0) { tr = L.trc; for (int e = tid; e < 3 * LK_TRACE_SLOTS * LK_TRACE_WORDS; e += T) L.trc[e] = 0; }
    // checksum of the four LDS arrays (position-weighted, wrapping): [0] patch, [1] jpatch, [2] sdx, [3] sdy
#define LK_CHECKSUMS(BASE_)                                                                       \
    {                                                                                             \
        if (tid < 4) L.chk[(BASE_) + tid] = 0;                                                    \
        __syncthreads();                                                                          \
        unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;                                                  \
        for (int e = tid; e < 18 * 18; e += T) c0 += (unsigned)patch[e] * (2u * e + 1u);          \
        for (int e = tid; e < LK_PJ * LK_PJ; e += T) c1 += (unsigned)jpatch[e] * (2u * e + 1u);   \
        for (int e = tid; e < 256; e += T) { c2 += (unsigned)sdx[e] * (2u * e + 1u); c3 += (unsigned)sdy[e] * (2u * e + 1u); } \
        atomicAdd(&L.chk[(BASE_) + 0], c0); atomicAdd(&L.chk[(BASE_) + 1], c1); atomicAdd(&L.chk[(BASE_) + 2], c2); atomicAdd(&L.chk[(BASE_) + 3], c3); \
        __syncthreads();                                                                          \
    }

    const float px0 = (float)st->prev[pt].x, py0 = (float)st->prev[pt].y;      // np.array(list(values), dtype=np.float32)
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (LK_WIN - 1) * 0.5f;
    float nxt_x = 0.f, nxt_y = 0.f;
    int status = 1;
    const int levels = a.cv.levels;
    for (int level = levels; level >= 0; --level) {
        long long* trl = nullptr;
        if constexpr ((DBG & 1) != 0) trl = tr + (size_t)level * LK_TRACE_SLOTS * LK_TRACE_WORDS;
        LkImg I, J;
        I.h = J.h = a.cv.lh[level]; I.w = J.w = a.cv.lw[level];
        I.p = a.cv.g[level] + (size_t)a.src_frame * I.h * I.w;
        J.p = a.cv.g[level] + (size_t)a.dst_frame * J.h * J.w;
        float ppx = px0 * (float)(1. / (1 << level)), ppy = py0 * (float)(1. / (1 << level));
        float nx, ny;
        if (level == levels) { nx = ppx; ny = ppy; }
        else { nx = nxt_x * 2.f; ny = nxt_y * 2.f; }
        nxt_x = nx; nxt_y = ny;
        ppx -= half; ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (lk_far(ppx, ppy) || ipx < -LK_WIN || ipx >= I.w || ipy < -LK_WIN || ipy >= I.h) {
            if (level == 0) status = 0;
            continue;
        }
        float fa = ppx - ipx, fb = ppy - ipy;
        int iw00 = (int)rintf((1.f - fa) * (1.f - fb) * (1 << LK_WBITS));
        int iw01 = (int)rintf(fa * (1.f - fb) * (1 << LK_WBITS));
        int iw10 = (int)rintf((1.f - fa) * fb * (1 << LK_WBITS));
        int iw11 = (1 << LK_WBITS) - iw00 - iw01 - iw10;
        __syncthreads();                                 // previous level's readers are done with the LDS patch
        // 18x18 gray neighbourhood (rows ipy-1.., cols ipx-1..), reflected at the image edges
        constexpr int NPV = (18 * 18 + T - 1) / T, NJV = (LK_PJ * LK_PJ + T - 1) / T;
        int pv[NPV]; unsigned char jv[NJV];                  // DBG bit 2: what this thread loaded, kept for the end-of-level check
#pragma unroll
        for (int k = 0; k < NPV; ++k) {
            const int e = tid + k * T;
            if (e < 18 * 18) {
                const int r = e / 18, c = e - r * 18;
                const int v = px(I, ipy - 1 + r, ipx - 1 + c);
                patch[e] = v;
                if constexpr ((DBG & 4) != 0) pv[k] = v;
            }
        }
        // ... and, in the same memory round trip, J's neighbourhood of the start window (see the iteration loop)
        const float sx_ = nx - half, sy_ = ny - half;
        const int jx0 = (lk_far(sx_, sy_) ? 0 : (int)floorf(sx_)) - LK_R, jy0 = (lk_far(sx_, sy_) ? 0 : (int)floorf(sy_)) - LK_R;
#pragma unroll
        for (int k = 0; k < NJV; ++k) {
            const int e = tid + k * T;
            if (e < LK_PJ * LK_PJ) {
                const int r = e / LK_PJ, c = e - r * LK_PJ;
                const unsigned char v = (unsigned char)px(J, jy0 + r, jx0 + c);
                jpatch[e] = v;
                if constexpr ((DBG & 4) != 0) jv[k] = v;
            }
        }
        __syncthreads();
        // Scharr at the 16x16 positions (ipy + r, ipx + c); zero outside the image (derivative buffer is BORDER_CONSTANT)
#pragma unroll
        for (int k4 = 0; k4 < PP; ++k4) {
            const int e = tid + k4 * T;
            const int r = e >> 4, c = e & 15;
            const int yy = ipy + r, xx = ipx + c;
            int dx = 0, dy = 0;
            if (xx >= 0 && xx < I.w && yy >= 0 && yy < I.h) {
                int t0[3], t1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int pa = patch[r * 18 + c + k], pb = patch[(r + 1) * 18 + c + k], pc = patch[(r + 2) * 18 + c + k];
                    t0[k] = (pa + pc) * 3 + pb * 10;
                    t1[k] = pc - pa;
                }
                dx = t0[2] - t0[0];
                dy = (t1[2] + t1[0]) * 3 + t1[1] * 10;
            }
            sdx[e] = dx; sdy[e] = dy;
        }
        __syncthreads();
        int ival[PP], ixval[PP], iyval[PP];
        const int lw00 = iw00, lw01 = iw01, lw10 = iw10, lw11 = iw11;     // DBG bit 2: the level's I-window weights, for the end-of-level check
        long long sA11 = 0, sA12 = 0, sA22 = 0;
#pragma unroll
        for (int k4 = 0; k4 < PP; ++k4) {
            const int e = tid + k4 * T;
            const int wy = e / LK_WIN, wx = e - wy * LK_WIN;
            ival[k4] = ixval[k4] = iyval[k4] = 0;
            if (e < LK_WIN * LK_WIN) {
                const int p00 = patch[(wy + 1) * 18 + wx + 1], p01 = patch[(wy + 1) * 18 + wx + 2];
                const int p10 = patch[(wy + 2) * 18 + wx + 1], p11 = patch[(wy + 2) * 18 + wx + 2];
                int iv = LK_DESCALE(p00 * iw00 + p01 * iw01 + p10 * iw10 + p11 * iw11, LK_WBITS - 5);
                const int d00 = wy * 16 + wx, d01 = d00 + 1, d10 = d00 + 16, d11 = d00 + 17;
                int ix = LK_DESCALE(sdx[d00] * iw00 + sdx[d01] * iw01 + sdx[d10] * iw10 + sdx[d11] * iw11, LK_WBITS);
                int iy = LK_DESCALE(sdy[d00] * iw00 + sdy[d01] * iw01 + sdy[d10] * iw10 + sdy[d11] * iw11, LK_WBITS);
                iv = (short)iv; ix = (short)ix; iy = (short)iy;
                ival[k4] = iv; ixval[k4] = ix; iyval[k4] = iy;
                sA11 += (long long)ix * ix; sA12 += (long long)ix * iy; sA22 += (long long)iy * iy;
            }
        }
        block_sum3<T>(sA11, sA12, sA22, red, tid);
        if constexpr ((DBG & 1) != 0) {
            if (tid == 0) {
                trl[0] = sA11; trl[1] = sA12; trl[2] = sA22; trl[3] = 0; trl[4] = 0;
                trl[5] = ((long long)iw00 << 32) | (unsigned)iw01; trl[6] = ((long long)ipx << 32) | (unsigned)ipy; trl[7] = 0x1111000000000000LL | (unsigned)level;
            }
        }
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LK_WIN * LK_WIN);
        if ((double)minEig < 1e-4 || D < 1.1920929e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        // the iterations re-read the 16x16 neighbourhood of a window that moves by a fraction of a pixel to a few pixels: J's
        // neighbourhood of the start window is in LDS (one global round trip per level instead of one per iteration)
        float pdx = 0.f, pdy = 0.f;
        int xr = 0, xj = -1;                               // DBG bit 0: why and when the loop ended
        for (int j = 0; j < a.max_count; ++j) {
            xj = j;
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (lk_far(nx, ny) || inx < -LK_WIN || inx >= J.w || iny < -LK_WIN || iny >= J.h) {
                if (level == 0) status = 0;
                xr = 1;
                break;
            }
            fa = nx - inx; fb = ny - iny;
            iw00 = (int)rintf((1.f - fa) * (1.f - fb) * (1 << LK_WBITS));
            iw01 = (int)rintf(fa * (1.f - fb) * (1 << LK_WBITS));
            iw10 = (int)rintf((1.f - fa) * fb * (1 << LK_WBITS));
            iw11 = (1 << LK_WBITS) - iw00 - iw01 - iw10;
            long long sb1 = 0, sb2 = 0, dummy = 0;
            const int ry = iny - jy0, rx = inx - jx0;                     // uniform: the whole window is inside the cached patch or not
            const bool cached = ry >= 0 && rx >= 0 && ry + LK_WIN < LK_PJ && rx + LK_WIN < LK_PJ;
#pragma unroll
            for (int k4 = 0; k4 < PP; ++k4) {
                const int e = tid + k4 * T;
                const int wy = e / LK_WIN, wx = e - wy * LK_WIN;
                if (e < LK_WIN * LK_WIN) {
                    const int yy = iny + wy, xx = inx + wx;
                    int j00, j01, j10, j11;
                    if (cached) {
                        const unsigned char* q = jpatch + (ry + wy) * LK_PJ + rx + wx;
                        j00 = q[0]; j01 = q[1]; j10 = q[LK_PJ]; j11 = q[LK_PJ + 1];
                    } else {
                        j00 = px(J, yy, xx); j01 = px(J, yy, xx + 1); j10 = px(J, yy + 1, xx); j11 = px(J, yy + 1, xx + 1);
                    }
                    const int diff = LK_DESCALE(j00 * iw00 + j01 * iw01 + j10 * iw10 + j11 * iw11, LK_WBITS - 5) - ival[k4];
                    sb1 += (long long)diff * ixval[k4]; sb2 += (long long)diff * iyval[k4];
                    if constexpr ((DBG & 1) != 0) dummy += j00 + 3 * j01 + 5 * j10 + 7 * j11 + ((long long)(ival[k4] & 0xFFFF) << 20) + ((long long)(ixval[k4] & 0xFFFF) << 36);   // what the lanes read / hold
                }
            }
            long long vb1 = sb1, vb2 = sb2;
            block_sum3<T>(sb1, sb2, dummy, red, tid);
            if constexpr ((DBG & 4) != 0) {                   // the same two sums once more through LDS atomics
                __syncthreads();
                if (tid == 0) { *(unsigned long long*)&L.chk[0] = 0; *(unsigned long long*)&L.chk[2] = 0; }
                __syncthreads();
                atomicAdd((unsigned long long*)&L.chk[0], (unsigned long long)vb1); atomicAdd((unsigned long long*)&L.chk[2], (unsigned long long)vb2);
                __syncthreads();
                if (tid == 0 && ((long long)*(unsigned long long*)&L.chk[0] != sb1 || (long long)*(unsigned long long*)&L.chk[2] != sb2)) atomicAdd(&a.dbg.counters[15], 1);
            }
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float ddx = (float)((A12 * b2 - A22 * b1) * D), ddy = (float)((A12 * b1 - A11 * b2) * D);
            if constexpr ((DBG & 1) != 0) {
                if (tid == 0) {
                    long long* t = trl + (1 + j) * LK_TRACE_WORDS;
                    t[0] = sb1; t[1] = sb2; t[2] = ((long long)__float_as_int(nx) << 32) | (unsigned)__float_as_int(ny);
                    t[3] = ((long long)__float_as_int(ddx) << 32) | (unsigned)__float_as_int(ddy); t[4] = ((long long)inx << 32) | (unsigned)iny;
                    t[5] = (long long)cached | ((long long)(rx & 0xFFF) << 8) | ((long long)(ry & 0xFFF) << 20) | (dummy << 32); t[6] = ((long long)iw00 << 32) | (unsigned)iw01; t[7] = 0x2222000000000000LL | ((long long)iw10 << 16) | (unsigned)j;
                }
            }
            nx += ddx; ny += ddy;
            nxt_x = nx + half; nxt_y = ny + half;
            if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= a.eps2) { xr = 2; break; }
            if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) {
                nxt_x -= ddx * 0.5f; nxt_y -= ddy * 0.5f;
                xr = 3;
                break;
            }
            pdx = ddx; pdy = ddy;
        }
        if constexpr ((DBG & 1) != 0) {
            if (tid == 0) {
                long long* t = trl + 11 * LK_TRACE_WORDS;
                t[2] = ((long long)__float_as_int(nxt_x) << 32) | (unsigned)__float_as_int(nxt_y); t[7] = 0x3333000000000000LL | (unsigned)level;
                t[3] = __builtin_amdgcn_s_getreg(20 | (31 << 11));      // HW_REG_XCC_ID
                t[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID
                t[5] = ((long long)a.src_frame << 32) | (unsigned)a.dst_frame; t[6] = ((long long)xr << 32) | (unsigned)xj; t[0] = a.max_count; t[1] = __double_as_longlong(a.eps2);
            }
        }
        if constexpr ((DBG & 4) != 0) {                  // end of level: registers vs LDS vs a second load, Scharr recomputed from LDS
            __syncthreads();
            int c_load = 0, c_lds = 0, c_jload = 0, c_jlds = 0, c_sch = 0;
#pragma unroll
            for (int k = 0; k < NPV; ++k) {
                const int e = tid + k * T;
                if (e < 18 * 18) {
                    const int r = e / 18, c = e - r * 18;
                    c_load += px(I, ipy - 1 + r, ipx - 1 + c) != pv[k];
                    c_lds += patch[e] != pv[k];
                }
            }
#pragma unroll
            for (int k = 0; k < NJV; ++k) {
                const int e = tid + k * T;
                if (e < LK_PJ * LK_PJ) {
                    const int r = e / LK_PJ, c = e - r * LK_PJ;
                    c_jload += (unsigned char)px(J, jy0 + r, jx0 + c) != jv[k];
                    c_jlds += jpatch[e] != jv[k];
                }
            }
            for (int e = tid; e < 256; e += T) {
                const int r = e >> 4, c = e & 15;
                const int yy = ipy + r, xx = ipx + c;
                int dx = 0, dy = 0;
                if (xx >= 0 && xx < I.w && yy >= 0 && yy < I.h) {
                    int t0[3], t1[3];
                    for (int k = 0; k < 3; ++k) {
                        const int pa = patch[r * 18 + c + k], pb = patch[(r + 1) * 18 + c + k], pc = patch[(r + 2) * 18 + c + k];
                        t0[k] = (pa + pc) * 3 + pb * 10; t1[k] = pc - pa;
                    }
                    dx = t0[2] - t0[0]; dy = (t1[2] + t1[0]) * 3 + t1[1] * 10;
                }
                c_sch += (sdx[e] != dx) + (sdy[e] != dy);
            }
            int c_reg = 0;                                   // registers that carried the I window through the iterations vs a re-derivation from LDS
#pragma unroll
            for (int k4 = 0; k4 < PP; ++k4) {
                const int e = tid + k4 * T;
                const int wy = e / LK_WIN, wx = e - wy * LK_WIN;
                if (e < LK_WIN * LK_WIN) {
                    const int p00 = patch[(wy + 1) * 18 + wx + 1], p01 = patch[(wy + 1) * 18 + wx + 2];
                    const int p10 = patch[(wy + 2) * 18 + wx + 1], p11 = patch[(wy + 2) * 18 + wx + 2];
                    const int iv = (short)LK_DESCALE(p00 * lw00 + p01 * lw01 + p10 * lw10 + p11 * lw11, LK_WBITS - 5);
                    const int d00 = wy * 16 + wx, d01 = d00 + 1, d10 = d00 + 16, d11 = d00 + 17;
                    const int ix = (short)LK_DESCALE(sdx[d00] * lw00 + sdx[d01] * lw01 + sdx[d10] * lw10 + sdx[d11] * lw11, LK_WBITS);
                    const int iy = (short)LK_DESCALE(sdy[d00] * lw00 + sdy[d01] * lw01 + sdy[d10] * lw10 + sdy[d11] * lw11, LK_WBITS);
                    c_reg += (iv != ival[k4]) + (ix != ixval[k4]) + (iy != iyval[k4]);
                }
            }
            if (c_reg) atomicAdd(&a.dbg.counters[14], c_reg);
            if (c_load) atomicAdd(&a.dbg.counters[8], c_load);
            if (c_lds) atomicAdd(&a.dbg.counters[9], c_lds);
            if (c_jload) atomicAdd(&a.dbg.counters[10], c_jload);
            if (c_jlds) atomicAdd(&a.dbg.counters[11], c_jlds);
            if (c_sch) atomicAdd(&a.dbg.counters[12], c_sch);
            if (tid == 0) atomicAdd(&a.dbg.counters[13], 1);
        }
        if (status && level == 0) {                      // the final window must start inside J's frame
            const float fx = nxt_x - half, fy = nxt_y - half;
            const int rx = (int)rintf(fx), ry = (int)rintf(fy);
            if (lk_far(fx, fy) || rx < -LK_WIN || rx >= J.w || ry < -LK_WIN || ry >= J.h) status = 0;
        }
    }
#undef LK_CHECKSUMS
    if constexpr ((DBG & 2) != 0) {
        __syncthreads();
        if (tid < 16) {
            const int* gs[6] = {L.g0, L.g1, L.g2, L.g3, L.g4, L.g5};
            for (int k = 0; k < 6; ++k)
                if (gs[k][tid] != (int)LK_GUARD) { atomicAdd(&a.dbg.counters[0], 1); atomicAdd(&a.dbg.counters[1 + k], 1); }
        }
    }
    if constexpr ((DBG & 1) != 0) {
        __syncthreads();
        long long* g = a.dbg.trace + (size_t)pt * 3 * LK_TRACE_SLOTS * LK_TRACE_WORDS;
        for (int e = tid; e < 3 * LK_TRACE_SLOTS * LK_TRACE_WORDS; e += T) g[e] = L.trc[e];
    }
    if (tid == 0) {
        st->lk_prev[2 * pt] = px0; st->lk_prev[2 * pt + 1] = py0;
        st->lk_next[2 * pt] = nxt_x; st->lk_next[2 * pt + 1] = nxt_y;
        st->lk_status[pt] = (unsigned char)status;
    }
}

template <int T, int DBG>
__global__ __launch_bounds__(T) void lk_kernel(LkArgs a) { lk_body<T, DBG>(a); }
// ---- developer diagnostics (eagle_debug): process-wide, not part of the data path ----------------------------------------------
static struct { int threads = 256, dbg = 0, excl_lds = 0; long long* trace = nullptr; int* counters = nullptr; } g_lk;
static const size_t LK_TRACE_BYTES = (size_t)EAGLE_N_LANDMARKS * 3 * LK_TRACE_SLOTS * LK_TRACE_WORDS * sizeof(long long);

int lk_debug(const char* key, long long value, void* out, long long out_bytes)
{
    const std::string k = key ? key : "";
    if (k == "lk_threads") { if (value != 64 && value != 256) return -1; g_lk.threads = (int)value; return 0; }
    if (k == "lk_excl_lds") { g_lk.excl_lds = (int)value; return 0; }      // dynamic LDS bytes requested on top: keeps other workgroups off the CU
    if (k == "lk_dbg") {
        g_lk.dbg = (int)value & 15;
        if (g_lk.dbg && !g_lk.trace) {
            HIP_CHECK(hipMalloc((void**)&g_lk.trace, LK_TRACE_BYTES));
            HIP_CHECK(hipMalloc((void**)&g_lk.counters, 64));
            HIP_CHECK(hipMemset(g_lk.counters, 0, 64));
        }
        return 0;
    }
    if (k == "lk_trace") {                                                  // copy out (device-synchronous) and clear
        if (!g_lk.trace || !out || out_bytes < (long long)LK_TRACE_BYTES) return -1;
        HIP_CHECK(hipMemcpy(out, g_lk.trace, LK_TRACE_BYTES, hipMemcpyDeviceToHost));
        return 0;
    }
    if (k == "lk_counters_reset") { if (g_lk.counters) HIP_CHECK(hipMemset(g_lk.counters, 0, 64)); return 0; }
    if (k == "lk_counters") {
        if (!g_lk.counters || !out || out_bytes < 64) return -1;
        HIP_CHECK(hipMemcpy(out, g_lk.counters, 64, hipMemcpyDeviceToHost));
        return 0;
    }
    return -2;
}

void lk_launch(const ClipView& cv, int src_frame, int dst_frame, ChainState* st, const MemList* mem, int kint, hipStream_t s)
{
    LkArgs a; a.cv = cv; a.src_frame = src_frame; a.dst_frame = dst_frame; a.st = st; a.mem = mem; a.kint = kint;
    a.max_count = 10; a.eps2 = 0.03 * 0.03;             // cm.py:65 criteria (EPS | COUNT, 10, 0.03)
    a.dbg.trace = g_lk.trace; a.dbg.counters = g_lk.counters;
    void (*fn)(LkArgs) = lk_kernel<256, 0>;
    if (g_lk.threads == 64) fn = g_lk.dbg == 0 ? lk_kernel<64, 0> : (g_lk.dbg == 1 ? lk_kernel<64, 1> : (g_lk.dbg == 2 ? lk_kernel<64, 2> : lk_kernel<64, 4>));
    else if (g_lk.dbg) fn = g_lk.dbg == 1 ? lk_kernel<256, 1> : (g_lk.dbg == 2 ? lk_kernel<256, 2> : (g_lk.dbg == 4 ? lk_kernel<256, 4> : lk_kernel<256, 8>));
    if (g_lk.excl_lds > 0) ensure_max_dynamic_lds((const void*)fn, g_lk.excl_lds);
    hipLaunchKernelGGL(fn, dim3(EAGLE_N_LANDMARKS), dim3(g_lk.threads), (size_t)g_lk.excl_lds, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
