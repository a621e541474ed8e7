/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under eagle_amd/ may include, link or call this file.
 *
 * CPU restatement of the OpenCV routines behind the reference's optical-flow key-point propagation
 * (eagle/models/coordinate_model.py:280 cvtColor BGR2GRAY, :434 calcOpticalFlowPyrLK with lk_params of :65
 * [winSize 15x15, maxLevel 2, criteria EPS|COUNT 10 0.03], :459/:469 cvtColor BGR2HSV, :538-545 brightness).
 *
 * PARITY UNPINNED: opencv-python 4.11.0.86 (uv.lock:992-993) is absent from /root/reference and from this image.
 * Restated from the published algorithm (modules/video/src/lkpyramid.cpp scalar path, modules/imgproc color_hsv /
 * color_yuv 8-bit paths, pyrDown 8-bit fixed point).  One deliberate deviation, applied identically in the HIP kernel:
 * the window sums A11/A12/A22 and b1/b2 are accumulated EXACTLY in 64-bit integers and converted to float once
 * (OpenCV accumulates the integer products in float, in an order that differs between its scalar and SIMD builds),
 * so the result does not depend on the summation order and CPU and GPU agree bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* cv::cvtColor(COLOR_BGR2GRAY), 8-bit: 15-bit fixed point, R2Y 9798, G2Y 19235, B2Y 3735 (sum 32768) */
void eo_bgr2gray(const uint8_t* bgr, int h, int w, uint8_t* gray)
{
    for (long i = 0; i < (long)h * w; ++i)
        gray[i] = (uint8_t)((bgr[3 * i] * 3735 + bgr[3 * i + 1] * 19235 + bgr[3 * i + 2] * 9798 + (1 << 14)) >> 15);
}

static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

/* cv::pyrDown, 8-bit: [1 4 6 4 1] x [1 4 6 4 1], (sum + 128) >> 8, BORDER_REFLECT_101, dst = (h+1)/2 x (w+1)/2 */
void eo_pyrdown(const uint8_t* src, int h, int w, uint8_t* dst)
{
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            int rows[5];
            for (int k = 0; k < 5; ++k) {
                const uint8_t* r = src + (long)reflect101(2 * y - 2 + k, h) * w;
                rows[k] = r[reflect101(2 * x - 2, w)] + r[reflect101(2 * x + 2, w)] +
                          4 * (r[reflect101(2 * x - 1, w)] + r[reflect101(2 * x + 1, w)]) + 6 * r[reflect101(2 * x, w)];
            }
            dst[(long)y * dw + x] = (uint8_t)((rows[0] + rows[4] + 4 * (rows[1] + rows[3]) + 6 * rows[2] + 128) >> 8);
        }
}

/* cv::cvtColor(COLOR_BGR2HSV), 8-bit, hue range 180: the table-driven fixed-point path (hsv_shift = 12) */
void eo_bgr2hsv_px(int b, int g, int r, int* ho, int* so, int* vo)
{
    int v = b, vmin = b;
    if (g > v) v = g; if (r > v) v = r;
    if (g < vmin) vmin = g; if (r < vmin) vmin = r;
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    /* sdiv_table[v] = saturate_cast<int>((255 << 12) / (1. * v)), hdiv_table180[d] = saturate_cast<int>((180 << 12) / (6. * d)); [0] = 0 */
    const int sdiv = v ? (int)lrint((255 << 12) / (1. * v)) : 0;
    const int hdiv = diff ? (int)lrint((180 << 12) / (6. * diff)) : 0;
    const int s = (diff * sdiv + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    *ho = hh & 255; *so = s & 255; *vo = v;           /* saturate_cast<uchar> of values already in range */
}

void eo_bgr2hsv(const uint8_t* bgr, long npix, uint8_t* hsv)
{
    for (long i = 0; i < npix; ++i) {
        int h, s, v;
        eo_bgr2hsv_px(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], &h, &s, &v);
        hsv[3 * i] = (uint8_t)h; hsv[3 * i + 1] = (uint8_t)s; hsv[3 * i + 2] = (uint8_t)v;
    }
}

/* ---- calcOpticalFlowPyrLK ---------------------------------------------------------------------------------------- */
typedef struct { const uint8_t* p; int h, w; } Img;
/* the pyramid levels carry a winSize-wide BORDER_REFLECT_101 frame in OpenCV; reads beyond the image are reflected */
static inline int px(const Img* im, int y, int x) { return im->p[(long)reflect101(y, im->h) * im->w + reflect101(x, im->w)]; }
/* calcScharrDeriv on the image, BORDER_REFLECT_101 at its edges; outside the image the derivative buffer is BORDER_CONSTANT 0 */
static inline void scharr(const Img* im, int y, int x, int* dx, int* dy)
{
    if (x < 0 || x >= im->w || y < 0 || y >= im->h) { *dx = 0; *dy = 0; return; }
    int t0[3], t1[3];
    for (int k = 0; k < 3; ++k) {
        const int xx = x - 1 + k;
        const int a = px(im, y - 1, xx), b = px(im, y, xx), c = px(im, y + 1, xx);
        t0[k] = (a + c) * 3 + b * 10;
        t1[k] = c - a;
    }
    *dx = (short)(t0[2] - t0[0]);
    *dy = (short)((t1[2] + t1[0]) * 3 + t1[1] * 10);
}
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
#define WIN 15
/* coordinates beyond any image (float -> int conversion of such values is unspecified in C and saturating on the GPU) count as outside */
#define FAR(x, y) (!(fabsf(x) < 1e8f && fabsf(y) < 1e8f))
#define W_BITS 14

/* prev_pts/next_pts: n x 2 float32; status: n bytes.  flags = 0, minEigThreshold = 1e-4, err requested (the Python binding always does). */
void eo_calc_optical_flow_pyr_lk(const uint8_t* prev_gray, const uint8_t* next_gray, int h, int w, const float* prev_pts, int n,
                                 int max_level, int max_count, double epsilon, float* next_pts, uint8_t* status)
{
    Img I[8], J[8];
    uint8_t* owned[16]; int n_owned = 0;
    I[0].p = prev_gray; I[0].h = h; I[0].w = w;
    J[0].p = next_gray; J[0].h = h; J[0].w = w;
    int levels = 0;
    for (int l = 1; l <= max_level && l < 8; ++l) {
        const int ph = I[l - 1].h, pw = I[l - 1].w, dh = (ph + 1) / 2, dw = (pw + 1) / 2;
        if (dw <= WIN || dh <= WIN) break;            /* buildOpticalFlowPyramid stops when a level would not exceed the window */
        uint8_t* a = (uint8_t*)malloc((size_t)dh * dw); uint8_t* b = (uint8_t*)malloc((size_t)dh * dw);
        eo_pyrdown(I[l - 1].p, ph, pw, a); eo_pyrdown(J[l - 1].p, ph, pw, b);
        I[l].p = a; I[l].h = dh; I[l].w = dw; J[l].p = b; J[l].h = dh; J[l].w = dw;
        owned[n_owned++] = a; owned[n_owned++] = b;
        levels = l;
    }
    if (max_count < 0) max_count = 0; if (max_count > 100) max_count = 100;
    if (epsilon < 0) epsilon = 0; if (epsilon > 10) epsilon = 10;
    epsilon *= epsilon;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (WIN - 1) * 0.5f;
    for (int i = 0; i < n; ++i) { status[i] = 1; next_pts[2 * i] = 0.f; next_pts[2 * i + 1] = 0.f; }

    for (int level = levels; level >= 0; --level) {
        const Img* Il = &I[level]; const Img* Jl = &J[level];
        for (int pt = 0; pt < n; ++pt) {
            float ppx = prev_pts[2 * pt] * (float)(1. / (1 << level)), ppy = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == levels) { nx = ppx; ny = ppy; }
            else { nx = next_pts[2 * pt] * 2.f; ny = next_pts[2 * pt + 1] * 2.f; }
            next_pts[2 * pt] = nx; next_pts[2 * pt + 1] = ny;
            ppx -= half; ppy -= half;
            const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
            if (FAR(ppx, ppy) || ipx < -WIN || ipx >= Il->w || ipy < -WIN || ipy >= Il->h) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            float a = ppx - ipx, b = ppy - ipy;
            int iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
            int iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            short Iw[WIN * WIN], dIx[WIN * WIN], dIy[WIN * WIN];
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < WIN; ++y)
                for (int x = 0; x < WIN; ++x) {
                    const int yy = ipy + y, xx = ipx + x;
                    int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
                    scharr(Il, yy, xx, &dx00, &dy00); scharr(Il, yy, xx + 1, &dx01, &dy01);
                    scharr(Il, yy + 1, xx, &dx10, &dy10); scharr(Il, yy + 1, xx + 1, &dx11, &dy11);
                    const int ival = DESCALE(px(Il, yy, xx) * iw00 + px(Il, yy, xx + 1) * iw01 + px(Il, yy + 1, xx) * iw10 + px(Il, yy + 1, xx + 1) * iw11, W_BITS - 5);
                    const int ixval = DESCALE(dx00 * iw00 + dx01 * iw01 + dx10 * iw10 + dx11 * iw11, W_BITS);
                    const int iyval = DESCALE(dy00 * iw00 + dy01 * iw01 + dy10 * iw10 + dy11 * iw11, W_BITS);
                    Iw[y * WIN + x] = (short)ival; dIx[y * WIN + x] = (short)ixval; dIy[y * WIN + x] = (short)iyval;
                    sA11 += (int64_t)ixval * ixval; sA12 += (int64_t)ixval * iyval; sA22 += (int64_t)iyval * iyval;
                }
            const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * WIN * WIN);
            if ((double)minEig < 1e-4 || D < 1.1920929e-07f) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; ++j) {
                const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                if (FAR(nx, ny) || inx < -WIN || inx >= Jl->w || iny < -WIN || iny >= Jl->h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
                iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                for (int y = 0; y < WIN; ++y)
                    for (int x = 0; x < WIN; ++x) {
                        const int yy = iny + y, xx = inx + x;
                        const int diff = DESCALE(px(Jl, yy, xx) * iw00 + px(Jl, yy, xx + 1) * iw01 + px(Jl, yy + 1, xx) * iw10 + px(Jl, yy + 1, xx + 1) * iw11, W_BITS - 5) - Iw[y * WIN + x];
                        sb1 += (int64_t)diff * dIx[y * WIN + x]; sb2 += (int64_t)diff * dIy[y * WIN + x];
                    }
                const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                const float ddx = (float)((A12 * b2 - A22 * b1) * D), ddy = (float)((A12 * b1 - A11 * b2) * D);
                nx += ddx; ny += ddy;
                next_pts[2 * pt] = nx + half; next_pts[2 * pt + 1] = ny + half;
                if ((double)ddx * ddx + (double)ddy * ddy <= epsilon) break;
                if (j > 0 && fabs(ddx + pdx) < 0.01 && fabs(ddy + pdy) < 0.01) {
                    next_pts[2 * pt] -= ddx * 0.5f; next_pts[2 * pt + 1] -= ddy * 0.5f;
                    break;
                }
                pdx = ddx; pdy = ddy;
            }
            if (status[pt] && level == 0) {               /* the err branch: the final window must start inside J's frame */
                const float fx = next_pts[2 * pt] - half, fy = next_pts[2 * pt + 1] - half;
                const int rx = (int)lrintf(fx), ry = (int)lrintf(fy);
                if (FAR(fx, fy) || rx < -WIN || rx >= Jl->w || ry < -WIN || ry >= Jl->h) status[pt] = 0;
            }
        }
    }
    for (int k = 0; k < n_owned; ++k) free(owned[k]);
}
