/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under eagle_amd/ may include, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * CPU restatement (plain C, fp32/fp64) of the arithmetic on the reference's per-frame path.  The reference
 * gets this arithmetic from torch / cv2 / ultralytics; every function cites the call site it
 * restates.  Parity status:
 *   - conv / BN / ReLU / bilinear / sigmoid / argmax: pinned against the reference's own HRNet module
 *     (eagle/models/keypoint_hrnet.py imported in the build container, tests/golden/make_golden.py).
 *   - resize / letterbox / fitLine / findHomography / perspectiveTransform / YOLOv8 decode / NMS:
 *     PARITY UNPINNED — cv2, ultralytics and torchvision are absent from /root/reference and from this
 *     image; restated from their published algorithms and checked against independent
 *     float64 numpy solvers and brute-force known answers only.
 *
 * Numeric contract shared with the HIP kernels (so that integer outputs can be compared bit-for-bit):
 *   - a convolution output is ONE fp32 fmaf chain:  acc = 0; for c16 in Cin/16-chunks: for (ky,kx): for c in
 *     chunk: acc = fmaf(x, w, acc);  then v = acc + bias.  (Zero-padded taps contribute fmaf(0,w,acc)=acc.)
 *     This is the order gfx950's v_mfma_f32_16x16x4_f32 accumulates in (k-ordered fmaf chain, exact fp32).
 *   - exp() is eo_expf below (pure fp32 fmaf polynomial), never libm, on both sides.
 *   - built with -ffp-contract=off: every fused multiply-add is an explicit fmaf().
 *   - "f16" mode emulates fp16 STORAGE (tensors/weights rounded to binary16 RNE, fp32 accumulate); the
 *     fp16 MFMA's internal summation order is not reproducible, so f16 results are compared by tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>

#define EO_ACT_NONE 0
#define EO_ACT_RELU 1
#define EO_ACT_SILU 2

/* ---------------------------------------------------------------------------------------------------- */
/* deterministic fp32 exp / sigmoid / silu                                                               */
/* ---------------------------------------------------------------------------------------------------- */
static inline float eo_expf(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);          /* ln2 hi (exact in 11 bits) */
    r = fmaf(n, -1.42860682030941723e-6f, r);           /* ln2 lo */
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return p * s.f;
}
static inline float eo_sigmoidf(float x) { return 1.0f / (1.0f + eo_expf(-x)); }
static inline float eo_act(float v, int act)
{
    if (act == EO_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == EO_ACT_SILU) return v * eo_sigmoidf(v);
    return v;
}
static inline float eo_q16(float v) { return _cvtsh_ss(_cvtss_sh(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }

void eo_exp_array(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = eo_expf(x[i]); }
void eo_sigmoid_array(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = eo_sigmoidf(x[i]); }
void eo_round_f16_array(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = eo_q16(x[i]); }

/* ---------------------------------------------------------------------------------------------------- */
/* convolution, NHWC, folded BN.  restates nn.Conv2d+BatchNorm2d(eval)+act (+residuals):                   */
/*   keypoint_hrnet.py:65-137 (BasicBlock/Bottleneck), :215-278 (fuse), :353-391 (transition); ultralytics */
/*   Conv / Bottleneck.                                                                  */
/* x [N,H,W,Cin]  w [KS,KS,Cin,Cout]  bias [Cout]  y [N,Ho,Wo,Cout]; r1/r2 optional [N,Ho,Wo,Cout].        */
/* v = acc + bias; v = pre(v); v = r1 + v; v = v + r2; v = post(v); optional fp16 rounding of the store.   */
/* ---------------------------------------------------------------------------------------------------- */
void eo_conv2d_nhwc(const float* x, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                    int KS, int stride, int pad, int Ho, int Wo, float* y, int pre_act, const float* r1,
                    const float* r2, int post_act, int f16_out)
{
    const int CH = 16;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy) {
            float* acc = (float*)aligned_alloc(64, sizeof(float) * (size_t)((Cout + 15) & ~15) * 4);
            for (int ox0 = 0; ox0 < Wo; ox0 += 4) {
                const int nx = (Wo - ox0) < 4 ? (Wo - ox0) : 4;
                for (int i = 0; i < nx * Cout; ++i) acc[i] = 0.0f;
                for (int c0 = 0; c0 < Cin; c0 += CH) {
                    const int c1 = (c0 + CH < Cin) ? c0 + CH : Cin;
                    for (int ky = 0; ky < KS; ++ky) {
                        const int iy = oy * stride + ky - pad;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < KS; ++kx) {
                            const float* wt = w + ((size_t)(ky * KS + kx) * Cin) * Cout;
                            for (int px = 0; px < nx; ++px) {
                                const int ix = (ox0 + px) * stride + kx - pad;
                                if (ix < 0 || ix >= W) continue;
                                const float* xp = x + (((size_t)n * H + iy) * W + ix) * Cin;
                                float* a = acc + (size_t)px * Cout;
                                for (int c = c0; c < c1; ++c) {
                                    const float xv = xp[c];
                                    const float* wr = wt + (size_t)c * Cout;
                                    #pragma omp simd
                                    for (int co = 0; co < Cout; ++co) a[co] = fmaf(xv, wr[co], a[co]);
                                }
                            }
                        }
                    }
                }
                for (int px = 0; px < nx; ++px) {
                    const size_t o = (((size_t)n * Ho + oy) * Wo + ox0 + px) * Cout;
                    const float* a = acc + (size_t)px * Cout;
                    for (int co = 0; co < Cout; ++co) {
                        float v = a[co] + bias[co];
                        v = eo_act(v, pre_act);
                        if (r1) v = r1[o + co] + v;
                        if (r2) v = v + r2[o + co];
                        v = eo_act(v, post_act);
                        y[o + co] = f16_out ? eo_q16(v) : v;
                    }
                }
            }
            free(acc);
        }
}

/* ---------------------------------------------------------------------------------------------------- */
/* bilinear upsample, align_corners=True  (F.interpolate, keypoint_hrnet.py:299-304).  x [N,h,w,C] -> y   */
/* ---------------------------------------------------------------------------------------------------- */
void eo_upsample_bilinear_ac(const float* x, int N, int h, int w, int C, int H, int W, float* y)
{
    const float sh = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sw = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < H; ++oy) {
            const float fy = sh * (float)oy;
            const int y0 = (int)fy;
            const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
            for (int ox = 0; ox < W; ++ox) {
                const float fx = sw * (float)ox;
                const int x0 = (int)fx;
                const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const float* p00 = x + (((size_t)n * h + y0) * w + x0) * C;
                const float* p01 = x + (((size_t)n * h + y0) * w + x1) * C;
                const float* p10 = x + (((size_t)n * h + y1) * w + x0) * C;
                const float* p11 = x + (((size_t)n * h + y1) * w + x1) * C;
                float* o = y + (((size_t)n * H + oy) * W + ox) * C;
                for (int c = 0; c < C; ++c) {
                    const float top = fmaf(lx1, p01[c], lx0 * p00[c]);
                    const float bot = fmaf(lx1, p11[c], lx0 * p10[c]);
                    o[c] = fmaf(ly1, bot, ly0 * top);
                }
            }
        }
}

/* ---------------------------------------------------------------------------------------------------- */
/* per-channel first-occurrence argmax of sigmoid(logits): KeypointModel.get_keypoints                    */
/* (keypoint_hrnet.py:581-593: np.argmax over the sigmoid heat-map, row-major, first maximum).            */
/* logits [HW, Cs] (NHWC, channel stride Cs), C real channels.                                            */
/* ---------------------------------------------------------------------------------------------------- */
void eo_heatmap_argmax(const float* logits, int HW, int Cs, int C, int32_t* idx, float* score)
{
    for (int c = 0; c < C; ++c) {
        float best = -1.0f; int bi = 0;
        for (int p = 0; p < HW; ++p) {
            const float s = eo_sigmoidf(logits[(size_t)p * Cs + c]);
            if (s > best) { best = s; bi = p; }
        }
        idx[c] = bi; score[c] = best;
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* u8 bilinear resize (cv2.resize INTER_LINEAR as used by A.Resize(540,960), coordinate_model.py:62-64,   */
/* and by ultralytics LetterBox).  PARITY UNPINNED (cv2 absent).                    */
/*  - exact 2x decimation takes cv2's INTER_AREA fast path: (a+b+c+d+2)>>2                                 */
/*  - otherwise half-pixel centres, 11-bit coefficients, cv2's 8u vertical pass rounding.                  */
/* src [sh,sw,3] (row stride in bytes), dst [dh,dw,3] dense.                                               */
/* ---------------------------------------------------------------------------------------------------- */
static void eo_lin_coef_scaled(int dsize, int ssize, double scale, int* ofs, short* co)
{
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
        ofs[d] = s;
        co[2 * d] = (short)lrintf((1.f - f) * 2048.f);
        co[2 * d + 1] = (short)lrintf(f * 2048.f);
    }
}
static void eo_resize_linear_general(const uint8_t* src, int sh, int sw, int64_t sstride, uint8_t* dst, int dh, int dw, double scale_x, double scale_y);
/* cv2.resize(src, (0, 0), fx, fy, INTER_LINEAR): dsize = round(ssize * f) is the caller's, the coordinate map uses 1 / fx and 1 / fy (OpenCV keeps
   inv_scale = fx when dsize is empty and only derives it from dsize / ssize when dsize is given) */
void eo_resize_linear_u8c3_fxfy(const uint8_t* src, int sh, int sw, int64_t sstride, uint8_t* dst, int dh, int dw, double fx, double fy)
{
    eo_resize_linear_general(src, sh, sw, sstride, dst, dh, dw, 1.0 / fx, 1.0 / fy);
}
void eo_resize_linear_u8c3(const uint8_t* src, int sh, int sw, int64_t sstride, uint8_t* dst, int dh, int dw)
{
    if (sh == dh && sw == dw) {
        for (int y = 0; y < dh; ++y) memcpy(dst + (size_t)y * dw * 3, src + (size_t)y * sstride, (size_t)dw * 3);
        return;
    }
    if (sh == 2 * dh && sw == 2 * dw) {
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < 3; ++c) {
                    const uint8_t* p = src + (size_t)(2 * y) * sstride + (size_t)(2 * x) * 3 + c;
                    dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)((p[0] + p[3] + p[sstride] + p[sstride + 3] + 2) >> 2);
                }
        return;
    }
    eo_resize_linear_general(src, sh, sw, sstride, dst, dh, dw, (double)sw / dw, (double)sh / dh);
}
static void eo_resize_linear_general(const uint8_t* src, int sh, int sw, int64_t sstride, uint8_t* dst, int dh, int dw, double scale_x, double scale_y)
{
    int* xo = (int*)malloc(sizeof(int) * dw); short* xc = (short*)malloc(sizeof(short) * 2 * dw);
    int* yo = (int*)malloc(sizeof(int) * dh); short* yc = (short*)malloc(sizeof(short) * 2 * dh);
    eo_lin_coef_scaled(dw, sw, scale_x, xo, xc);
    eo_lin_coef_scaled(dh, sh, scale_y, yo, yc);
    for (int y = 0; y < dh; ++y) {
        const int y0 = yo[y], y1 = (y0 + 1 < sh) ? y0 + 1 : y0;
        const int b0 = yc[2 * y], b1 = yc[2 * y + 1];
        for (int x = 0; x < dw; ++x) {
            const int x0 = xo[x], x1 = (x0 + 1 < sw) ? x0 + 1 : x0;
            const int a0 = xc[2 * x], a1 = xc[2 * x + 1];
            for (int c = 0; c < 3; ++c) {
                const int t0 = src[(size_t)y0 * sstride + x0 * 3 + c] * a0 + src[(size_t)y0 * sstride + x1 * 3 + c] * a1;
                const int t1 = src[(size_t)y1 * sstride + x0 * 3 + c] * a0 + src[(size_t)y1 * sstride + x1 * 3 + c] * a1;
                dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xo); free(xc); free(yo); free(yc);
}

/* ---------------------------------------------------------------------------------------------------- */
/* YOLOv8 Detect decode for one level (ultralytics Detect inference path):               */
/* box [A,64] DFL logits, cls [A,>=nc] logits (NHWC), anchors = cell centres, stride s.                    */
/* out rows [A, 4+nc] = (cx,cy,w,h)*stride, sigmoid(cls).                                                  */
/* ---------------------------------------------------------------------------------------------------- */
void eo_yolo_decode_level(const float* box, int box_cs, const float* cls, int cls_cs, int nc, int gh, int gw,
                          float stride, float* out)
{
    for (int a = 0; a < gh * gw; ++a) {
        const float ax = (float)(a % gw) + 0.5f, ay = (float)(a / gw) + 0.5f;
        float d[4];
        for (int s = 0; s < 4; ++s) {
            const float* l = box + (size_t)a * box_cs + s * 16;
            float m = l[0];
            for (int i = 1; i < 16; ++i) m = l[i] > m ? l[i] : m;
            float e[16], den = 0.f;
            for (int i = 0; i < 16; ++i) { e[i] = eo_expf(l[i] - m); den = den + e[i]; }
            float num = 0.f;
            for (int i = 0; i < 16; ++i) num = fmaf((float)i, e[i] / den, num);
            d[s] = num;
        }
        const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
        float* o = out + (size_t)a * (4 + nc);
        o[0] = ((x1 + x2) / 2.0f) * stride;
        o[1] = ((y1 + y2) / 2.0f) * stride;
        o[2] = (x2 - x1) * stride;
        o[3] = (y2 - y1) * stride;
        for (int c = 0; c < nc; ++c) o[4 + c] = eo_sigmoidf(cls[(size_t)a * cls_cs + c]);
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* cv2.fitLine(pts, DIST_L2, 0, 0.01, 0.01) closed form (coordinate_model.py:106).        */
/* PARITY UNPINNED.  pts float32 [n,2] -> (vx,vy,x0,y0) float32.                                          */
/* ---------------------------------------------------------------------------------------------------- */
void eo_fit_line_l2(const float* pts, int n, float* line)
{
    double x = 0, y = 0, x2 = 0, y2 = 0, xy = 0;
    for (int i = 0; i < n; ++i) {
        const double px = pts[2 * i], py = pts[2 * i + 1];
        x += px; y += py; x2 += px * px; y2 += py * py; xy += px * py;
    }
    const double w = (double)n;
    x /= w; y /= w; x2 /= w; y2 /= w; xy /= w;
    const double dx2 = x2 - x * x, dy2 = y2 - y * y, dxy = xy - x * y;
    const float t = (float)atan2(2 * dxy, dx2 - dy2) / 2;
    line[0] = (float)cos(t); line[1] = (float)sin(t);
    line[2] = (float)x; line[3] = (float)y;
}

/* ---------------------------------------------------------------------------------------------------- */
/* cv2.findHomography(src, dst, RANSAC, 5.0) (coordinate_model.py:355). PARITY UNPINNED. */
/* ---------------------------------------------------------------------------------------------------- */
/* symmetric 9x9 eigen-solve in float64: classic cyclic Jacobi on the UPPER triangle (diagonal kept in d[], rotations
 * in the tau form, small off-diagonals zeroed after 4 sweeps); returns the eigenvector of the smallest eigenvalue.
 * cv::eigen uses a max-pivot Jacobi; both converge to the same eigenvector to ~1e-15.  The HIP kernel
 * (eagle_amd/csrc/geom.hip::jacobi9_smallest) performs exactly these operations in exactly this order. */
#define EO_ROT(x, y) do { const double g_ = (x), h_ = (y); (x) = g_ - s * (h_ + g_ * tau); (y) = h_ + s * (g_ - h_ * tau); } while (0)
static void eo_jacobi9_smallest(double a[9][9], double v[9])
{
    double V[9][9], d[9], b[9], z[9];
    for (int i = 0; i < 9; ++i) { for (int j = 0; j < 9; ++j) V[i][j] = (i == j) ? 1.0 : 0.0; d[i] = b[i] = a[i][i]; z[i] = 0.0; }
    for (int sweep = 1; sweep <= 50; ++sweep) {
        double sm = 0.0;
        for (int p = 0; p < 8; ++p) for (int q = p + 1; q < 9; ++q) sm += fabs(a[p][q]);
        if (sm == 0.0) break;
        const double tresh = sweep < 4 ? 0.2 * sm / 81.0 : 0.0;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double g = 100.0 * fabs(a[p][q]);
                if (sweep > 4 && fabs(d[p]) + g == fabs(d[p]) && fabs(d[q]) + g == fabs(d[q])) { a[p][q] = 0.0; continue; }
                if (!(fabs(a[p][q]) > tresh)) continue;
                double h = d[q] - d[p], t;
                if (fabs(h) + g == fabs(h)) t = a[p][q] / h;
                else {
                    const double theta = 0.5 * h / a[p][q];
                    t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                    if (theta < 0.0) t = -t;
                }
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);
                h = t * a[p][q];
                z[p] -= h; z[q] += h; d[p] -= h; d[q] += h; a[p][q] = 0.0;
                for (int j = 0; j < p; ++j) EO_ROT(a[j][p], a[j][q]);
                for (int j = p + 1; j < q; ++j) EO_ROT(a[p][j], a[j][q]);
                for (int j = q + 1; j < 9; ++j) EO_ROT(a[p][j], a[q][j]);
                for (int j = 0; j < 9; ++j) EO_ROT(V[j][p], V[j][q]);
            }
        for (int i = 0; i < 9; ++i) { b[i] += z[i]; d[i] = b[i]; z[i] = 0.0; }
    }
    int m = 0;
    for (int i = 1; i < 9; ++i) if (d[i] < d[m]) m = i;
    for (int k = 0; k < 9; ++k) v[k] = V[k][m];
}

/* normalised DLT ("runKernel"): src/dst double [n][2]; H[9]; returns 1 on success */
int eo_dlt_homography(const double* src, const double* dst, const int* sel, int n, double* H)
{
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const int k = sel ? sel[i] : i;
        cM[0] += src[2 * k]; cM[1] += src[2 * k + 1]; cm[0] += dst[2 * k]; cm[1] += dst[2 * k + 1];
    }
    cM[0] /= n; cM[1] /= n; cm[0] /= n; cm[1] /= n;
    for (int i = 0; i < n; ++i) {
        const int k = sel ? sel[i] : i;
        sM[0] += fabs(src[2 * k] - cM[0]); sM[1] += fabs(src[2 * k + 1] - cM[1]);
        sm[0] += fabs(dst[2 * k] - cm[0]); sm[1] += fabs(dst[2 * k + 1] - cm[1]);
    }
    if (fabs(sM[0]) < 2.220446049250313e-16 || fabs(sM[1]) < 2.220446049250313e-16 ||
        fabs(sm[0]) < 2.220446049250313e-16 || fabs(sm[1]) < 2.220446049250313e-16) return 0;
    sM[0] = n / sM[0]; sM[1] = n / sM[1]; sm[0] = n / sm[0]; sm[1] = n / sm[1];
    double LtL[9][9];
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < n; ++i) {
        const int k = sel ? sel[i] : i;
        const double X = (src[2 * k] - cM[0]) * sM[0], Y = (src[2 * k + 1] - cM[1]) * sM[1];
        const double x = (dst[2 * k] - cm[0]) * sm[0], y = (dst[2 * k + 1] - cm[1]) * sm[1];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int a = 0; a < 9; ++a) for (int b = a; b < 9; ++b) LtL[a][b] += Lx[a] * Lx[b] + Ly[a] * Ly[b];
    }
    for (int a = 0; a < 9; ++a) for (int b = 0; b < a; ++b) LtL[a][b] = LtL[b][a];
    double h[9];
    eo_jacobi9_smallest(LtL, h);
    /* H = inv(T_dst) * H0 * T_src,  T = [[s0,0,-c0*s0],[0,s1,-c1*s1],[0,0,1]] */
    const double iT[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double T[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double t[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += iT[3 * r + k] * h[3 * k + c]; t[3 * r + c] = s; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += t[3 * r + k] * T[3 * k + c]; H[3 * r + c] = s; }
    if (fabs(H[8]) < 2.220446049250313e-16) return 0;
    const double inv = 1.0 / H[8];
    for (int k = 0; k < 9; ++k) H[k] *= inv;
    H[8] = 1.0;
    return 1;
}

/* Minimal-sample (4 correspondences) model of the RANSAC hypothesis stage.  cv2's runKernel takes the null vector of the
 * 9x9 LtL also for 4 points; the 8 equations determine H up to scale, so the same H is obtained by fixing h33 = 1 in the
 * NORMALISED frame and solving the 8x8 system by Gaussian elimination with partial pivoting (fails only when the normalised
 * h33 vanishes).  This is what eagle_amd/csrc/geom.hip::h4_homography computes, operation for operation.  The final fit on
 * all inliers still goes through eo_dlt_homography (LtL eigen-solve) + LM like cv2. */
int eo_h4_homography(const double* src, const double* dst, const int* idx, double* H)
{
    const int n = 4;
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; ++i) { const int k = idx[i]; cM[0] += src[2 * k]; cM[1] += src[2 * k + 1]; cm[0] += dst[2 * k]; cm[1] += dst[2 * k + 1]; }
    cM[0] /= n; cM[1] /= n; cm[0] /= n; cm[1] /= n;
    for (int i = 0; i < n; ++i) {
        const int k = idx[i];
        sM[0] += fabs(src[2 * k] - cM[0]); sM[1] += fabs(src[2 * k + 1] - cM[1]);
        sm[0] += fabs(dst[2 * k] - cm[0]); sm[1] += fabs(dst[2 * k + 1] - cm[1]);
    }
    if (fabs(sM[0]) < 2.220446049250313e-16 || fabs(sM[1]) < 2.220446049250313e-16 ||
        fabs(sm[0]) < 2.220446049250313e-16 || fabs(sm[1]) < 2.220446049250313e-16) return 0;
    sM[0] = n / sM[0]; sM[1] = n / sM[1]; sm[0] = n / sm[0]; sm[1] = n / sm[1];
    double M[8][9];
    for (int i = 0; i < n; ++i) {
        const int k = idx[i];
        const double X = (src[2 * k] - cM[0]) * sM[0], Y = (src[2 * k + 1] - cM[1]) * sM[1];
        const double x = (dst[2 * k] - cm[0]) * sm[0], y = (dst[2 * k + 1] - cm[1]) * sm[1];
        double* a = M[2 * i]; double* b = M[2 * i + 1];
        a[0] = X; a[1] = Y; a[2] = 1; a[3] = 0; a[4] = 0; a[5] = 0; a[6] = -x * X; a[7] = -x * Y; a[8] = x;
        b[0] = 0; b[1] = 0; b[2] = 0; b[3] = X; b[4] = Y; b[5] = 1; b[6] = -y * X; b[7] = -y * Y; b[8] = y;
    }
    for (int c = 0; c < 8; ++c) {
        int p = c; double best = fabs(M[c][c]);
        for (int r = c + 1; r < 8; ++r) if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); p = r; }
        if (best < 1e-13) return 0;
        if (p != c) for (int j = c; j < 9; ++j) { const double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
        for (int r = c + 1; r < 8; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int j = c + 1; j < 9; ++j) M[r][j] = M[r][j] - f * M[c][j];
        }
    }
    double h[9];
    for (int i = 7; i >= 0; --i) {
        double s = M[i][8];
        for (int j = i + 1; j < 8; ++j) s = s - M[i][j] * h[j];
        h[i] = s / M[i][i];
    }
    h[8] = 1.0;
    const double iT[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double T[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double t[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += iT[3 * r + k] * h[3 * k + c]; t[3 * r + c] = s; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += t[3 * r + k] * T[3 * k + c]; H[3 * r + c] = s; }
    if (fabs(H[8]) < 2.220446049250313e-16) return 0;
    const double inv = 1.0 / H[8];
    for (int k = 0; k < 9; ++k) H[k] *= inv;
    H[8] = 1.0;
    return 1;
}

/* squared reprojection error in dst units, float like cv2's computeError */
static void eo_reproj_err(const double* src, const double* dst, int n, const double* H, float* err)
{
    for (int i = 0; i < n; ++i) {
        const double X = src[2 * i], Y = src[2 * i + 1];
        const double ww = 1.0 / (H[6] * X + H[7] * Y + 1.0);
        const double dx = (H[0] * X + H[1] * Y + H[2]) * ww - dst[2 * i];
        const double dy = (H[3] * X + H[4] * Y + H[5]) * ww - dst[2 * i + 1];
        err[i] = (float)(dx * dx + dy * dy);
    }
}

/* cv2 "checkSubset" for 4 points: no 3 collinear in either set + consistent orientation */
static int eo_collinear_last(const double* p, const int* idx, int count)
{
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = p[2 * idx[j]] - p[2 * idx[i]], dy1 = p[2 * idx[j] + 1] - p[2 * idx[i] + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = p[2 * idx[k]] - p[2 * idx[i]], dy2 = p[2 * idx[k] + 1] - p[2 * idx[i] + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}
static int eo_check_subset4(const double* src, const double* dst, const int* idx)
{
    for (int c = 3; c <= 4; ++c)
        if (eo_collinear_last(src, idx, c) || eo_collinear_last(dst, idx, c)) return 0;
    static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; ++i) {
        const int a = idx[tt[i][0]], b = idx[tt[i][1]], c = idx[tt[i][2]];
        const double A = src[2 * a] * (src[2 * b + 1] - src[2 * c + 1]) - src[2 * a + 1] * (src[2 * b] - src[2 * c]) +
                         (src[2 * b] * src[2 * c + 1] - src[2 * c] * src[2 * b + 1]);
        const double B = dst[2 * a] * (dst[2 * b + 1] - dst[2 * c + 1]) - dst[2 * a + 1] * (dst[2 * b] - dst[2 * c]) +
                         (dst[2 * b] * dst[2 * c + 1] - dst[2 * c] * dst[2 * b + 1]);
        negative += (A * B < 0);
    }
    return negative == 0 || negative == 4;
}

static uint32_t eo_rng_next(uint64_t* st)
{
    *st = (uint64_t)(uint32_t)(*st) * 4164903690ULL + (uint32_t)(*st >> 32);
    return (uint32_t)(*st);
}

void eo_rng_stream(uint32_t* out, int n)
{
    uint64_t st = 0xffffffffffffffffULL;
    for (int i = 0; i < n; ++i) out[i] = eo_rng_next(&st);
}

static int eo_ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    if (p < 0) p = 0; if (p > 1) p = 1;
    if (ep < 0) ep = 0; if (ep > 1) ep = 1;
    const double num0 = 1.0 - p;
    const double num = num0 > 2.2250738585072014e-308 ? num0 : 2.2250738585072014e-308;
    const double denom = 1.0 - pow(1.0 - ep, model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    const double ln = log(num), ld = log(denom);
    return (ld >= 0 || -ln >= max_iters * (-ld)) ? max_iters : (int)lrint(ln / ld);
}

/* Levenberg-Marquardt polish of the 8 free entries (h33 = 1), at most 10 iterations, residual =
 * reprojection error in dst units (cv2 HomographyRefineCallback + LMSolver). */
static void eo_lm_residual(const double* src, const double* dst, int n, const double* h, double* r, double* J)
{
    for (int i = 0; i < n; ++i) {
        const double Mx = src[2 * i], My = src[2 * i + 1];
        double ww = h[6] * Mx + h[7] * My + 1.0;
        ww = fabs(ww) > 2.220446049250313e-16 ? 1.0 / ww : 0.0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww, yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        r[2 * i] = xi - dst[2 * i]; r[2 * i + 1] = yi - dst[2 * i + 1];
        if (J) {
            double* a = J + (size_t)(2 * i) * 8; double* b = a + 8;
            a[0] = Mx * ww; a[1] = My * ww; a[2] = ww; a[3] = a[4] = a[5] = 0.0;
            a[6] = -Mx * ww * xi; a[7] = -My * ww * xi;
            b[0] = b[1] = b[2] = 0.0; b[3] = Mx * ww; b[4] = My * ww; b[5] = ww;
            b[6] = -Mx * ww * yi; b[7] = -My * ww * yi;
        }
    }
}
static int eo_solve8(double A[8][8], double b[8], double x[8])
{   /* Gaussian elimination with partial pivoting */
    double M[8][9];
    for (int i = 0; i < 8; ++i) { for (int j = 0; j < 8; ++j) M[i][j] = A[i][j]; M[i][8] = b[i]; }
    for (int c = 0; c < 8; ++c) {
        int p = c;
        for (int r = c + 1; r < 8; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (fabs(M[p][c]) < 1e-300) return 0;
        if (p != c) for (int j = 0; j < 9; ++j) { const double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
        for (int r = c + 1; r < 8; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int j = c; j < 9; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 7; i >= 0; --i) {
        double s = M[i][8];
        for (int j = i + 1; j < 8; ++j) s -= M[i][j] * x[j];
        x[i] = s / M[i][i];
    }
    return 1;
}
static const double EO_P10[33] = {1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5,
                                   1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10,
                                   1e11, 1e12, 1e13, 1e14, 1e15, 1e16};
void eo_lm_refine(const double* src, const double* dst, int n, double* H, int max_iters)
{
    double* r = (double*)malloc(sizeof(double) * 2 * n);
    double* rn = (double*)malloc(sizeof(double) * 2 * n);
    double* J = (double*)malloc(sizeof(double) * 2 * n * 8);
    double h[8]; for (int k = 0; k < 8; ++k) h[k] = H[k];
    eo_lm_residual(src, dst, n, h, r, J);
    double S = 0; for (int i = 0; i < 2 * n; ++i) S += r[i] * r[i];
    int lambdaLg10 = -3;
    for (int it = 0; it < max_iters; ++it) {
        double A[8][8], g[8];
        for (int a = 0; a < 8; ++a) {
            g[a] = 0; for (int i = 0; i < 2 * n; ++i) g[a] += J[(size_t)i * 8 + a] * r[i];
            for (int b = 0; b < 8; ++b) { double s = 0; for (int i = 0; i < 2 * n; ++i) s += J[(size_t)i * 8 + a] * J[(size_t)i * 8 + b]; A[a][b] = s; }
        }
        int improved = 0;
        for (int tries = 0; tries < 16 && !improved; ++tries) {
            double Ap[8][8], d[8], hn[8], gm[8];
            const double lam = EO_P10[lambdaLg10 + 16];
            for (int a = 0; a < 8; ++a) { for (int b = 0; b < 8; ++b) Ap[a][b] = A[a][b]; Ap[a][a] += lam * A[a][a]; gm[a] = -g[a]; }
            if (!eo_solve8(Ap, gm, d)) { lambdaLg10 = lambdaLg10 + 1 > 16 ? 16 : lambdaLg10 + 1; continue; }
            for (int k = 0; k < 8; ++k) hn[k] = h[k] + d[k];
            eo_lm_residual(src, dst, n, hn, rn, 0);
            double Sn = 0; for (int i = 0; i < 2 * n; ++i) Sn += rn[i] * rn[i];
            if (Sn < S) {
                for (int k = 0; k < 8; ++k) h[k] = hn[k];
                S = Sn; improved = 1;
                lambdaLg10 = lambdaLg10 - 1 < -16 ? -16 : lambdaLg10 - 1;
            } else {
                lambdaLg10 = lambdaLg10 + 1 > 16 ? 16 : lambdaLg10 + 1;
            }
        }
        if (!improved) break;
        eo_lm_residual(src, dst, n, h, r, J);
    }
    for (int k = 0; k < 8; ++k) H[k] = h[k];
    H[8] = 1.0;
    free(r); free(rn); free(J);
}

/* src (image) / dst (world) float32 [n,2]; returns 1 and fills H[9] (double), mask[n] on success. */
int eo_find_homography_ransac(const float* srcf, const float* dstf, int n, double thresh, int max_iters,
                              double confidence, int refine_iters, double* H, uint8_t* mask)
{
    if (n < 4) return 0;
    double* src = (double*)malloc(sizeof(double) * 2 * n);
    double* dst = (double*)malloc(sizeof(double) * 2 * n);
    float* err = (float*)malloc(sizeof(float) * n);
    uint8_t* m = (uint8_t*)malloc(n);
    for (int i = 0; i < 2 * n; ++i) { src[i] = srcf[i]; dst[i] = dstf[i]; }
    int ok = 0;
    double best[9];
    if (n == 4) {
        ok = eo_dlt_homography(src, dst, 0, 4, best);
        for (int i = 0; i < n; ++i) mask[i] = 1;
    } else {
        uint64_t rng = 0xffffffffffffffffULL;
        int niters = max_iters, max_good = 0;
        const float t2 = (float)(thresh * thresh);
        for (int iter = 0; iter < niters; ++iter) {
            int idx[4], found = 0;
            for (int attempt = 0; attempt < 1000 && !found; ++attempt) {
                for (int i = 0; i < 4; ++i) {
                    int v, dup;
                    do {
                        v = (int)(eo_rng_next(&rng) % (uint32_t)n);
                        dup = 0;
                        for (int j = 0; j < i; ++j) dup |= (idx[j] == v);
                    } while (dup);
                    idx[i] = v;
                }
                found = eo_check_subset4(src, dst, idx);
            }
            if (!found) { if (iter == 0) { ok = 0; goto done; } break; }
            double Hc[9];
            if (!eo_h4_homography(src, dst, idx, Hc)) continue;
            eo_reproj_err(src, dst, n, Hc, err);
            int good = 0;
            for (int i = 0; i < n; ++i) { m[i] = err[i] <= t2; good += m[i]; }
            if (good > (max_good > 3 ? max_good : 3)) {
                memcpy(mask, m, n); memcpy(best, Hc, sizeof(best));
                max_good = good;
                niters = eo_ransac_update_iters(confidence, (double)(n - good) / n, 4, niters);
            }
        }
        ok = max_good > 0;
        if (ok) {
            /* compress to inliers, DLT on all of them, then LM polish */
            int* sel = (int*)malloc(sizeof(int) * n); int ni = 0;
            for (int i = 0; i < n; ++i) if (mask[i]) sel[ni++] = i;
            double* s2 = (double*)malloc(sizeof(double) * 2 * ni); double* d2 = (double*)malloc(sizeof(double) * 2 * ni);
            for (int i = 0; i < ni; ++i) { s2[2 * i] = src[2 * sel[i]]; s2[2 * i + 1] = src[2 * sel[i] + 1]; d2[2 * i] = dst[2 * sel[i]]; d2[2 * i + 1] = dst[2 * sel[i] + 1]; }
            double Hr[9];
            if (eo_dlt_homography(s2, d2, 0, ni, Hr)) {
                memcpy(best, Hr, sizeof(best));
                if (refine_iters > 0) eo_lm_refine(s2, d2, ni, best, refine_iters);
            }
            free(sel); free(s2); free(d2);
        }
    }
done:
    if (ok) memcpy(H, best, sizeof(best));
    free(src); free(dst); free(err); free(m);
    return ok;
}

/* ---------------------------------------------------------------------------------------------------- */
/* Second CPU mode: the solver OpenCV 4.11 itself uses (parity unpinned: cv2 is absent; restated from the published
 * sources), to BOUND what the production mode's two deliberate deviations can change (tests/test_oracle_host.py):
 *   (i)  runKernel for the minimal 4-point hypotheses too: 9x9 LtL + eigenvector of the smallest eigenvalue
 *        (production: normalised 8x8 Gaussian elimination, eo_h4_homography);
 *   (ii) cv::eigen's Jacobi (JacobiImpl_): every rotation annihilates the LARGEST off-diagonal element, found through
 *        per-row / per-column maximum indices, at most 30 n^2 rotations, eigenvalues sorted descending
 *        (production: cyclic sweeps, eo_jacobi9_smallest).
 * Also cv2.LMEDS (LMeDSPointSetRegistrator), the last fall-back of cm.py:354-357. */
/* ---------------------------------------------------------------------------------------------------- */
static void eo_jacobi_cv(double* A, int n, double* W, double* V)
{
    const double eps = 2.220446049250313e-16;
    int indR[16], indC[16];
    int i, j, k, m;
    double mv = 0;
    for (i = 0; i < n; ++i) { for (j = 0; j < n; ++j) V[i * n + j] = 0; V[i * n + i] = 1; }
    for (k = 0; k < n; ++k) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) { for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; ++i) { const double val = fabs(A[n * k + i]); if (mv < val) mv = val, m = i; } indR[k] = m; }
        if (k > 0) { for (m = 0, mv = fabs(A[k]), i = 1; i < k; ++i) { const double val = fabs(A[n * i + k]); if (mv < val) mv = val, m = i; } indC[k] = m; }
    }
    if (n > 1) for (int iters = 0; iters < n * n * 30; ++iters) {
        for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; ++i) { const double val = fabs(A[n * i + indR[i]]); if (mv < val) mv = val, k = i; }
        int l = indR[k];
        for (i = 1; i < n; ++i) { const double val = fabs(A[n * indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
        const double p = A[n * k + l];
        if (fabs(p) <= eps) break;
        const double y = (W[l] - W[k]) * 0.5;
        double t = fabs(y) + hypot(p, y);
        double s = hypot(p, t);
        const double c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t; W[l] += t;
        double a0, b0;
#define EO_CVROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; ++i) EO_CVROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; ++i) EO_CVROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; ++i) EO_CVROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; ++i) EO_CVROT(V[n * k + i], V[n * l + i]);
#undef EO_CVROT
        for (j = 0; j < 2; ++j) {
            const int idx = j == 0 ? k : l;
            if (idx < n - 1) { for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; ++i) { const double val = fabs(A[n * idx + i]); if (mv < val) mv = val, m = i; } indR[idx] = m; }
            if (idx > 0) { for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; ++i) { const double val = fabs(A[n * i + idx]); if (mv < val) mv = val, m = i; } indC[idx] = m; }
        }
    }
    for (k = 0; k < n - 1; ++k) {                          /* descending eigenvalues; eigenvectors are the ROWS of V */
        m = k;
        for (i = k + 1; i < n; ++i) if (W[m] < W[i]) m = i;
        if (k != m) { double tw = W[m]; W[m] = W[k]; W[k] = tw; for (i = 0; i < n; ++i) { double tv = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tv; } }
    }
}

/* runKernel exactly as HomographyEstimatorCallback does it, for any n >= 4 (sel: optional index list) */
int eo_dlt_homography_cv(const double* src, const double* dst, const int* sel, int n, double* H)
{
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; ++i) { const int k = sel ? sel[i] : i; cm[0] += dst[2 * k]; cm[1] += dst[2 * k + 1]; cM[0] += src[2 * k]; cM[1] += src[2 * k + 1]; }
    cm[0] /= n; cm[1] /= n; cM[0] /= n; cM[1] /= n;
    for (int i = 0; i < n; ++i) {
        const int k = sel ? sel[i] : i;
        sm[0] += fabs(dst[2 * k] - cm[0]); sm[1] += fabs(dst[2 * k + 1] - cm[1]);
        sM[0] += fabs(src[2 * k] - cM[0]); sM[1] += fabs(src[2 * k + 1] - cM[1]);
    }
    if (fabs(sm[0]) < 2.220446049250313e-16 || fabs(sm[1]) < 2.220446049250313e-16 || fabs(sM[0]) < 2.220446049250313e-16 || fabs(sM[1]) < 2.220446049250313e-16) return 0;
    sm[0] = n / sm[0]; sm[1] = n / sm[1]; sM[0] = n / sM[0]; sM[1] = n / sM[1];
    double LtL[81], W[9], V[81];
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < n; ++i) {
        const int k = sel ? sel[i] : i;
        const double x = (dst[2 * k] - cm[0]) * sm[0], y = (dst[2 * k + 1] - cm[1]) * sm[1];
        const double X = (src[2 * k] - cM[0]) * sM[0], Y = (src[2 * k + 1] - cM[1]) * sM[1];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int a = 0; a < 9; ++a) for (int b = a; b < 9; ++b) LtL[a * 9 + b] += Lx[a] * Lx[b] + Ly[a] * Ly[b];
    }
    for (int a = 0; a < 9; ++a) for (int b = 0; b < a; ++b) LtL[a * 9 + b] = LtL[b * 9 + a];
    eo_jacobi_cv(LtL, 9, W, V);
    const double* h = V + 8 * 9;                           /* row of the smallest eigenvalue */
    const double iT[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double T[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double t[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += iT[3 * r + k] * h[3 * k + c]; t[3 * r + c] = s; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += t[3 * r + k] * T[3 * k + c]; H[3 * r + c] = s; }
    if (fabs(H[8]) < 2.220446049250313e-16) return 0;
    const double inv = 1.0 / H[8];
    for (int k = 0; k < 9; ++k) H[k] *= inv;
    H[8] = 1.0;
    return 1;
}

static int eo_get_subset4(const double* src, const double* dst, int n, uint64_t* rng, int* idx)
{
    for (int attempt = 0; attempt < 1000; ++attempt) {
        for (int i = 0; i < 4; ++i) {
            int v, dup;
            do { v = (int)(eo_rng_next(rng) % (uint32_t)n); dup = 0; for (int j = 0; j < i; ++j) dup |= (idx[j] == v); } while (dup);
            idx[i] = v;
        }
        if (eo_check_subset4(src, dst, idx)) return 1;
    }
    return 0;
}
static int eo_cmp_float(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return (x > y) - (x < y); }

/* ---------------------------------------------------------------------------------------------------- */
/* cv2.RHO (coordinate_model.py:354-357 tries it between cv2.RANSAC and cv2.LMEDS, with the threshold argument None = findHomography's      */
/* default 3.0).  OpenCV 4.11 modules/calib3d/src/rho.cpp is NOT in /root/reference nor in this image: restated from the publication it        */
/* implements — Bazargani, Bilaniuk, Laganiere, "A fast and robust homography scheme for real-time planar target detection" (2015):           */
/* PROSAC sampling (Chum & Matas 2005) over the correspondences in the order given, SPRT verification (Matas & Chum 2005) with the            */
/* adaptive (epsilon, delta) design, termination by the PROSAC maximality bound / the RANSAC confidence bound, a final refinement over the   */
/* consensus set.  PARITY UNPINNED.  What is kept from rho.cpp as far as the publication and its defaults define it: sample size 4, SPRT      */
/* constants t_M = 25, m_S = 1, epsilon0 = 0.1, delta0 = 0.01, re-design when delta moves by more than 10 %, the sample test "no coincident    */
/* points, no three collinear, same side of the lines (0,1) and (2,3) in both images" (signed distances compared after truncation to int),   */
/* the model test "no NaN",                                                                                                                  */
/* acceptance with >= minInl = 4 consensus members, findHomography's arguments (maxIters 2000, confidence 0.995, NR and final refinement     */
/* enabled, beta 0.35).  Stated deviations: the 4-point model comes from this file's normalised 8 x 8 solver (rho.cpp: float Gauss-Jordan on  */
/* the un-normalised system), the random stream is cv::RNG's multiply-with-carry (rho.cpp: its own xorshift128+), the refinement is           */
/* eo_lm_refine (rho.cpp: its own Levenberg-Marquardt with Cholesky steps), no NR-step inside the loop.  None of these changes WHICH inputs  */
/* yield a model — the question the GPU path depends on (tests/test_oracle_host.py::test_rho_fallback_never_rescues_what_ransac_rejects).     */
/* ---------------------------------------------------------------------------------------------------- */
static double eo_sprt_design(double delta, double eps, double tM, double mS)
{
    const double C = (1 - delta) * log((1 - delta) / (1 - eps)) + delta * log(delta / eps);
    const double K = tM * C / mS + 1;
    double An1 = K, An = K;
    for (int i = 0; i < 10; ++i) { An = K + log(An1); if (An - An1 < 1.5e-8) break; An1 = An; }
    return An;
}
static int eo_rho_sample_degenerate(const float* src, const float* dst, const int* s)
{
    float p[16];
    for (int k = 0; k < 4; ++k) { p[2 * k] = src[2 * s[k]]; p[2 * k + 1] = src[2 * s[k] + 1]; p[8 + 2 * k] = dst[2 * s[k]]; p[8 + 2 * k + 1] = dst[2 * s[k] + 1]; }
    for (int a = 0; a < 4; ++a)                               /* two source points that share both coordinates */
        for (int b = a + 1; b < 4; ++b)
            if (p[2 * a] == p[2 * b] && p[2 * a + 1] == p[2 * b + 1]) return 1;
    /* three collinear points in either image: the 8 x 8 system is singular, no homography is determined (the publication's degeneracy test; what
       rho.cpp's float Gauss-Jordan returns for such a sample is a rounding artefact that this restatement does not imitate) */
    {
        const int id[4] = {0, 1, 2, 3};
        double ps[8], pd[8];
        for (int k = 0; k < 8; ++k) { ps[k] = p[k]; pd[k] = p[8 + k]; }
        for (int c = 3; c <= 4; ++c)
            if (eo_collinear_last(ps, id, c) || eo_collinear_last(pd, id, c)) return 1;
    }
    static const int lines[4][3] = {{0, 1, 2}, {0, 1, 3}, {2, 3, 0}, {2, 3, 1}};      /* (line through a, b) . c */
    for (int t = 0; t < 4; ++t) {
        const int a = lines[t][0], b = lines[t][1], c = lines[t][2];
        float d[2];
        for (int side = 0; side < 2; ++side) {
            const float* q = p + 8 * side;
            const float c0 = q[2 * a + 1] - q[2 * b + 1], c1 = q[2 * b] - q[2 * a], c2 = q[2 * a] * q[2 * b + 1] - q[2 * a + 1] * q[2 * b];
            d[side] = c0 * q[2 * c] + c1 * q[2 * c + 1] + c2;
        }
        if ((((int)d[0]) ^ ((int)d[1])) < 0) return 1;        /* opposite sides (the comparison is made on the truncated integers, as rho.cpp does) */
    }
    return 0;
}
int eo_find_homography_rho(const float* srcf, const float* dstf, int n, double maxD, int maxI, double cfd, int minInl, double* H, uint8_t* mask)
{
    if (n < 4) return 0;
    double* src = (double*)malloc(sizeof(double) * 2 * n);
    double* dst = (double*)malloc(sizeof(double) * 2 * n);
    uint8_t* cur = (uint8_t*)malloc(n);
    for (int i = 0; i < 2 * n; ++i) { src[i] = srcf[i]; dst[i] = dstf[i]; }
    uint64_t rng = 0xffffffffffffffffULL;
    const double tM = 25.0, mS = 1.0;
    double eps = 0.1, delta = 0.01;
    double A = eo_sprt_design(delta, eps, tM, mS), lamAcc = delta / eps, lamRej = (1 - delta) / (1 - eps);
    /* PROSAC growth function: T_4 = rConvg * prod (4 - i) / (N - i); phase n draws 3 points from the first n - 1 and takes point n */
    int phNum = 4, phEndI = 1;
    double phEndFp = (double)maxI;
    for (int i = 0; i < 4; ++i) phEndFp *= (double)(4 - i) / (double)(n - i);
    int best_inl = 0, limit = maxI;
    double best[9];
    const float maxD2 = (float)(maxD * maxD);
    for (int it = 0; it < limit; ++it) {
        if (it > phEndI && phNum < n) {                       /* next PROSAC phase */
            ++phNum;
            const double next = phEndFp * phNum / (phNum - 4);
            phEndI += (int)ceil(next - phEndFp);
            phEndFp = next;
        }
        int s[4];
        const int pool = (it > phEndI || phNum >= n) ? n : phNum - 1, m = pool == n ? 4 : 3;
        for (int i = 0; i < m; ++i) {
            int v, dup;
            do { v = (int)(eo_rng_next(&rng) % (uint32_t)pool); dup = 0; for (int j = 0; j < i; ++j) dup |= (s[j] == v); } while (dup);
            s[i] = v;
        }
        if (m == 3) s[3] = phNum - 1;
        if (eo_rho_sample_degenerate(srcf, dstf, s)) continue;
        double Hc[9];
        if (!eo_h4_homography(src, dst, s, Hc)) continue;
        double sum = 0; for (int k = 0; k < 8; ++k) sum += Hc[k];
        if (sum != sum) continue;                             /* model test: NaN */
        /* SPRT: accumulate the likelihood ratio point by point, reject as soon as it passes A */
        double lam = 1.0; int inl = 0, tested = 0, good = 1;
        for (int j = 0; j < n; ++j) {
            const double X = src[2 * j], Y = src[2 * j + 1];
            const double ww = 1.0 / (Hc[6] * X + Hc[7] * Y + 1.0);
            const double dx = (Hc[0] * X + Hc[1] * Y + Hc[2]) * ww - dst[2 * j], dy = (Hc[3] * X + Hc[4] * Y + Hc[5]) * ww - dst[2 * j + 1];
            const int in = (float)(dx * dx + dy * dy) <= maxD2;
            cur[j] = (uint8_t)in; inl += in; ++tested;
            lam *= in ? lamAcc : lamRej;
            if (lam > A) { good = 0; break; }
        }
        if (good) {
            if (inl > best_inl) {
                best_inl = inl; memcpy(best, Hc, sizeof(best)); memcpy(mask, cur, n);
                /* termination bounds: RANSAC confidence on the inlier ratio; a larger consensus set also raises epsilon */
                limit = eo_ransac_update_iters(cfd, (double)(n - inl) / n, 4, limit);
                const double e2 = (double)inl / n;
                if (e2 > eps && e2 < 1.0) { eps = e2; if (delta >= eps) delta = eps * 0.5; A = eo_sprt_design(delta, eps, tM, mS); lamAcc = delta / eps; lamRej = (1 - delta) / (1 - eps); }
            }
        } else {
            const double d2 = (double)inl / tested;           /* a rejected model estimates delta */
            if (d2 > 0 && d2 < eps && fabs(d2 - delta) / delta > 0.1) { delta = d2; A = eo_sprt_design(delta, eps, tM, mS); lamAcc = delta / eps; lamRej = (1 - delta) / (1 - eps); }
        }
    }
    int ok = best_inl >= minInl;
    if (ok) {
        if (n > 4) {                                          /* final refinement over the consensus set */
            double* s2 = (double*)malloc(sizeof(double) * 2 * best_inl); double* d2 = (double*)malloc(sizeof(double) * 2 * best_inl);
            int k = 0;
            for (int i = 0; i < n; ++i) if (mask[i]) { s2[2 * k] = src[2 * i]; s2[2 * k + 1] = src[2 * i + 1]; d2[2 * k] = dst[2 * i]; d2[2 * k + 1] = dst[2 * i + 1]; ++k; }
            eo_lm_refine(s2, d2, k, best, 10);
            free(s2); free(d2);
        }
        memcpy(H, best, sizeof(best));
    }
    free(src); free(dst); free(cur);
    return ok;
}

/* mode 0: production deviations (8x8 minimal solver, cyclic Jacobi); mode 1: cv2's own solver throughout.
 * method 8 = cv2.RANSAC (thresh used), 4 = cv2.LMEDS (thresh ignored; confidence 0.995, maxIters 2000 as findHomography passes them). */
int eo_find_homography_ex(const float* srcf, const float* dstf, int n, int method, double thresh, int max_iters,
                          double confidence, int refine_iters, int mode, double* H, uint8_t* mask)
{
    if (n < 4) return 0;
    if (method == 16) return eo_find_homography_rho(srcf, dstf, n, thresh, max_iters, confidence, 4, H, mask);      /* cv2.RHO */
    if (method == 8 && mode == 0) return eo_find_homography_ransac(srcf, dstf, n, thresh, max_iters, confidence, refine_iters, H, mask);
    double* src = (double*)malloc(sizeof(double) * 2 * n);
    double* dst = (double*)malloc(sizeof(double) * 2 * n);
    float* err = (float*)malloc(sizeof(float) * n);
    float* srt = (float*)malloc(sizeof(float) * n);
    uint8_t* m = (uint8_t*)malloc(n);
    for (int i = 0; i < 2 * n; ++i) { src[i] = srcf[i]; dst[i] = dstf[i]; }
    int ok = 0;
    double best[9];
    if (n == 4) {
        ok = mode ? eo_dlt_homography_cv(src, dst, 0, 4, best) : eo_dlt_homography(src, dst, 0, 4, best);
        for (int i = 0; i < n; ++i) mask[i] = 1;
    } else if (method == 8) {
        uint64_t rng = 0xffffffffffffffffULL;
        int niters = max_iters, max_good = 0;
        const float t2 = (float)(thresh * thresh);
        for (int iter = 0; iter < niters; ++iter) {
            int idx[4];
            if (!eo_get_subset4(src, dst, n, &rng, idx)) { if (iter == 0) { ok = 0; goto done; } break; }
            double Hc[9];
            if (!eo_dlt_homography_cv(src, dst, idx, 4, Hc)) continue;
            eo_reproj_err(src, dst, n, Hc, err);
            int good = 0;
            for (int i = 0; i < n; ++i) { m[i] = err[i] <= t2; good += m[i]; }
            if (good > (max_good > 3 ? max_good : 3)) {
                memcpy(mask, m, n); memcpy(best, Hc, sizeof(best));
                max_good = good;
                niters = eo_ransac_update_iters(confidence, (double)(n - good) / n, 4, niters);
            }
        }
        ok = max_good > 0;
    } else {                                               /* LMEDS */
        uint64_t rng = 0xffffffffffffffffULL;
        int niters = eo_ransac_update_iters(confidence, 0.45, 4, max_iters);
        double min_median = 1.7976931348623157e308;
        for (int iter = 0; iter < niters; ++iter) {
            int idx[4];
            if (!eo_get_subset4(src, dst, n, &rng, idx)) { if (iter == 0) { ok = 0; goto done; } break; }
            double Hc[9];
            const int got = mode ? eo_dlt_homography_cv(src, dst, idx, 4, Hc) : eo_h4_homography(src, dst, idx, Hc);
            if (!got) continue;
            eo_reproj_err(src, dst, n, Hc, err);
            memcpy(srt, err, sizeof(float) * n);
            qsort(srt, n, sizeof(float), eo_cmp_float);
            const double median = n % 2 != 0 ? srt[n / 2] : (srt[n / 2 - 1] + srt[n / 2]) * 0.5;
            if (median < min_median) { min_median = median; memcpy(best, Hc, sizeof(best)); ok = 1; }
        }
        if (ok) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 4)) * sqrt(min_median);
            sigma = sigma > 0.001 ? sigma : 0.001;
            eo_reproj_err(src, dst, n, best, err);
            int good = 0;
            for (int i = 0; i < n; ++i) { mask[i] = err[i] <= (float)(sigma * sigma); good += mask[i]; }
            ok = good >= 4;
        }
    }
    if (ok && n > 4) {
        int* sel = (int*)malloc(sizeof(int) * n); int ni = 0;
        for (int i = 0; i < n; ++i) if (mask[i]) sel[ni++] = i;
        double* s2 = (double*)malloc(sizeof(double) * 2 * ni); double* d2 = (double*)malloc(sizeof(double) * 2 * ni);
        for (int i = 0; i < ni; ++i) { s2[2 * i] = src[2 * sel[i]]; s2[2 * i + 1] = src[2 * sel[i] + 1]; d2[2 * i] = dst[2 * sel[i]]; d2[2 * i + 1] = dst[2 * sel[i] + 1]; }
        double Hr[9];
        if (mode ? eo_dlt_homography_cv(s2, d2, 0, ni, Hr) : eo_dlt_homography(s2, d2, 0, ni, Hr)) {
            memcpy(best, Hr, sizeof(best));
            if (refine_iters > 0) eo_lm_refine(s2, d2, ni, best, refine_iters);
        }
        free(sel); free(s2); free(d2);
    }
done:
    if (ok) memcpy(H, best, sizeof(best));
    free(src); free(dst); free(err); free(srt); free(m);
    return ok;
}

/* cv2.perspectiveTransform (coordinate_model.py:383,400-403): double compute, float store */
void eo_perspective_transform(const float* pts, int n, const double* H, float* out)
{
    for (int i = 0; i < n; ++i) {
        const double x = pts[2 * i], y = pts[2 * i + 1];
        double w = H[6] * x + H[7] * y + H[8];
        if (fabs(w) > 2.220446049250313e-16) {
            w = 1.0 / w;
            out[2 * i] = (float)((H[0] * x + H[1] * y + H[2]) * w);
            out[2 * i + 1] = (float)((H[3] * x + H[4] * y + H[5]) * w);
        } else {
            out[2 * i] = out[2 * i + 1] = 0.f;
        }
    }
}
