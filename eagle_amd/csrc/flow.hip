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

// exact workgroup sum of three values per thread (4 waves); every thread receives the totals.  A wave's partial sums fit 32 bits
// (|diff * derivative| <= 8160 * 4080 per pixel, 64 pixels per wave: 2.13e9 < 2^31), so the cross-lane steps move 32-bit values and
// only the four per-wave partials are added in 64 bits.
__device__ __forceinline__ void block_sum3(long long& a, long long& b, long long& c, long long (*red)[3], int tid)
{
    int a32 = (int)a, b32 = (int)b, c32 = (int)c;
    // wave reduction on the DPP path (row shifts inside 16 lanes, then row broadcasts): the total ends up in lane 63
#define LK_DPP_STEP(CTRL_, ROWM_, BANKM_)                                                \
    a32 += __builtin_amdgcn_update_dpp(0, a32, CTRL_, ROWM_, BANKM_, false);              \
    b32 += __builtin_amdgcn_update_dpp(0, b32, CTRL_, ROWM_, BANKM_, false);              \
    c32 += __builtin_amdgcn_update_dpp(0, c32, CTRL_, ROWM_, BANKM_, false);
    LK_DPP_STEP(0x111, 0xf, 0xf)                         // row_shr:1
    LK_DPP_STEP(0x112, 0xf, 0xf)                         // row_shr:2
    LK_DPP_STEP(0x114, 0xf, 0xe)                         // row_shr:4
    LK_DPP_STEP(0x118, 0xf, 0xc)                         // row_shr:8
    LK_DPP_STEP(0x142, 0xa, 0xf)                         // row_bcast:15
    LK_DPP_STEP(0x143, 0xc, 0xf)                         // row_bcast:31
#undef LK_DPP_STEP
    __syncthreads();                                     // the previous round's readers are done with `red`
    if ((tid & 63) == 63) { red[tid >> 6][0] = a32; red[tid >> 6][1] = b32; red[tid >> 6][2] = c32; }
    __syncthreads();
    a = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    c = red[0][2] + red[1][2] + red[2][2] + red[3][2];
}

__device__ __forceinline__ bool lk_far(float x, float y) { return !(fabsf(x) < 1e8f && fabsf(y) < 1e8f); }

struct LkArgs {
    ClipView cv; int src_frame, dst_frame;
    ChainState* st;
    const MemList* mem; int kint;          // chain mode (mem != nullptr): the step decides itself whether frame dst_frame needs a flow
    int max_count; double eps2;
};

#define LK_R 8                                   // J is cached this many pixels around the start window; beyond it the loop reads global memory
#define LK_PJ (LK_WIN + 1 + 2 * LK_R)
__global__ __launch_bounds__(256) void lk_kernel(LkArgs a)
{
    __shared__ int patch[18 * 18];
    __shared__ unsigned char jpatch[LK_PJ * LK_PJ];
    __shared__ int sdx[16 * 16], sdy[16 * 16];
    __shared__ long long red[4][3];
    ChainState* st = a.st;
    const int tid = threadIdx.x, pt = blockIdx.x;
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

    const float px0 = (float)st->prev[pt].x, py0 = (float)st->prev[pt].y;      // np.array(list(values), dtype=np.float32)
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (LK_WIN - 1) * 0.5f;
    const int wy = tid / LK_WIN, wx = tid - wy * LK_WIN;
    const bool inwin = tid < LK_WIN * LK_WIN;
    float nxt_x = 0.f, nxt_y = 0.f;
    int status = 1;
    const int levels = a.cv.levels;
    for (int level = levels; level >= 0; --level) {
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
        for (int e = tid; e < 18 * 18; e += 256) {
            const int r = e / 18, c = e - r * 18;
            patch[e] = lk_px(I, ipy - 1 + r, ipx - 1 + c);
        }
        // ... and, in the same memory round trip, J's neighbourhood of the start window (see the iteration loop)
        const float sx_ = nx - half, sy_ = ny - half;
        const int jx0 = (lk_far(sx_, sy_) ? 0 : (int)floorf(sx_)) - LK_R, jy0 = (lk_far(sx_, sy_) ? 0 : (int)floorf(sy_)) - LK_R;
        for (int e = tid; e < LK_PJ * LK_PJ; e += 256) {
            const int r = e / LK_PJ, c = e - r * LK_PJ;
            jpatch[e] = (unsigned char)lk_px(J, jy0 + r, jx0 + c);
        }
        __syncthreads();
        {   // Scharr at the 16x16 positions (ipy + r, ipx + c); zero outside the image (derivative buffer is BORDER_CONSTANT)
            const int r = tid >> 4, c = tid & 15;
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
            sdx[tid] = dx; sdy[tid] = dy;
        }
        __syncthreads();
        int ival = 0, ixval = 0, iyval = 0;
        long long sA11 = 0, sA12 = 0, sA22 = 0;
        if (inwin) {
            const int p00 = patch[(wy + 1) * 18 + wx + 1], p01 = patch[(wy + 1) * 18 + wx + 2];
            const int p10 = patch[(wy + 2) * 18 + wx + 1], p11 = patch[(wy + 2) * 18 + wx + 2];
            ival = LK_DESCALE(p00 * iw00 + p01 * iw01 + p10 * iw10 + p11 * iw11, LK_WBITS - 5);
            const int d00 = wy * 16 + wx, d01 = d00 + 1, d10 = d00 + 16, d11 = d00 + 17;
            ixval = LK_DESCALE(sdx[d00] * iw00 + sdx[d01] * iw01 + sdx[d10] * iw10 + sdx[d11] * iw11, LK_WBITS);
            iyval = LK_DESCALE(sdy[d00] * iw00 + sdy[d01] * iw01 + sdy[d10] * iw10 + sdy[d11] * iw11, LK_WBITS);
            ival = (short)ival; ixval = (short)ixval; iyval = (short)iyval;
            sA11 = (long long)ixval * ixval; sA12 = (long long)ixval * iyval; sA22 = (long long)iyval * iyval;
        }
        block_sum3(sA11, sA12, sA22, red, tid);
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
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (lk_far(nx, ny) || inx < -LK_WIN || inx >= J.w || iny < -LK_WIN || iny >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            fa = nx - inx; fb = ny - iny;
            iw00 = (int)rintf((1.f - fa) * (1.f - fb) * (1 << LK_WBITS));
            iw01 = (int)rintf(fa * (1.f - fb) * (1 << LK_WBITS));
            iw10 = (int)rintf((1.f - fa) * fb * (1 << LK_WBITS));
            iw11 = (1 << LK_WBITS) - iw00 - iw01 - iw10;
            long long sb1 = 0, sb2 = 0, dummy = 0;
            if (inwin) {
                const int yy = iny + wy, xx = inx + wx;
                int j00, j01, j10, j11;
                const int ry = iny - jy0, rx = inx - jx0;                 // uniform: the whole window is inside the cached patch or not
                if (ry >= 0 && rx >= 0 && ry + LK_WIN < LK_PJ && rx + LK_WIN < LK_PJ) {
                    const unsigned char* q = jpatch + (ry + wy) * LK_PJ + rx + wx;
                    j00 = q[0]; j01 = q[1]; j10 = q[LK_PJ]; j11 = q[LK_PJ + 1];
                } else {
                    j00 = lk_px(J, yy, xx); j01 = lk_px(J, yy, xx + 1); j10 = lk_px(J, yy + 1, xx); j11 = lk_px(J, yy + 1, xx + 1);
                }
                const int diff = LK_DESCALE(j00 * iw00 + j01 * iw01 + j10 * iw10 + j11 * iw11, LK_WBITS - 5) - ival;
                sb1 = (long long)diff * ixval; sb2 = (long long)diff * iyval;
            }
            block_sum3(sb1, sb2, dummy, red, tid);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float ddx = (float)((A12 * b2 - A22 * b1) * D), ddy = (float)((A12 * b1 - A11 * b2) * D);
            nx += ddx; ny += ddy;
            nxt_x = nx + half; nxt_y = ny + half;
            if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= a.eps2) break;
            if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) {
                nxt_x -= ddx * 0.5f; nxt_y -= ddy * 0.5f;
                break;
            }
            pdx = ddx; pdy = ddy;
        }
        if (status && level == 0) {                      // the final window must start inside J's frame
            const float fx = nxt_x - half, fy = nxt_y - half;
            const int rx = (int)rintf(fx), ry = (int)rintf(fy);
            if (lk_far(fx, fy) || rx < -LK_WIN || rx >= J.w || ry < -LK_WIN || ry >= J.h) status = 0;
        }
    }
    if (tid == 0) {
        st->lk_prev[2 * pt] = px0; st->lk_prev[2 * pt + 1] = py0;
        st->lk_next[2 * pt] = nxt_x; st->lk_next[2 * pt + 1] = nxt_y;
        st->lk_status[pt] = (unsigned char)status;
    }
}

void lk_launch(const ClipView& cv, int src_frame, int dst_frame, ChainState* st, const MemList* mem, int kint, hipStream_t s)
{
    LkArgs a; a.cv = cv; a.src_frame = src_frame; a.dst_frame = dst_frame; a.st = st; a.mem = mem; a.kint = kint;
    a.max_count = 10; a.eps2 = 0.03 * 0.03;             // cm.py:65 criteria (EPS | COUNT, 10, 0.03)
    hipLaunchKernelGGL(lk_kernel, dim3(EAGLE_N_LANDMARKS), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
