// Bandwidth-bound kernels of the per-frame path: fused frame preprocess (K1), HRNet fuse = bilinear
// align_corners upsample + N-way sum + ReLU (K4), SPPF max-pool / nearest x2 / concat-by-slice (K8), and the
// per-channel first-maximum of sigmoid(logits) (K5).  All NHWC, 16-byte vector accesses per lane.
#include <type_traits>
#include "common.h"
#include "dmath.h"

namespace eagle {

using half8 = __attribute__((ext_vector_type(8))) _Float16;

// ------------------------------------------------------------------------------------------------------------
// K1: one read of the BGR u8 frame -> (a) 540x960 ImageNet-normalised RGB tensor for HRNet
//     (cv2.cvtColor + A.Resize + A.Normalize, eagle/models/coordinate_model.py:62-64,489-491) and
//     (b) letter-boxed RGB/255 tensor for the detector (ultralytics LetterBox).
// u8 resize restates cv2.resize INTER_LINEAR: 2x decimation = area fast path, else 11-bit fixed point.
// ------------------------------------------------------------------------------------------------------------
struct ResizeAxis { int s0, s1; int a0, a1; };

__device__ __forceinline__ ResizeAxis lin_coef(int d, int dsize, int ssize)
{
    const double scale = (double)ssize / (double)dsize;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
    ResizeAxis r;
    r.s0 = s; r.s1 = (s + 1 < ssize) ? s + 1 : s;
    r.a0 = (int)(short)lrintf((1.f - f) * 2048.f);
    r.a1 = (int)(short)lrintf(f * 2048.f);
    return r;
}

// resized RGB u8 pixel (dy,dx) of an (sh,sw)->(dh,dw) resize; src is BGR
__device__ __forceinline__ void resize_px(const uint8_t* src, int sh, int sw, int dh, int dw, int dy, int dx, int rgb[3])
{
    const size_t rs = (size_t)sw * 3;
    if (sh == dh && sw == dw) {
        const uint8_t* p = src + dy * rs + dx * 3;
        rgb[0] = p[2]; rgb[1] = p[1]; rgb[2] = p[0];
    } else if (sh == 2 * dh && sw == 2 * dw) {
        const uint8_t* p = src + (size_t)(2 * dy) * rs + (size_t)(2 * dx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[2 - c] = (p[c] + p[3 + c] + p[rs + c] + p[rs + 3 + c] + 2) >> 2;
    } else {
        const ResizeAxis ax = lin_coef(dx, dw, sw), ay = lin_coef(dy, dh, sh);
        const uint8_t* r0 = src + ay.s0 * rs; const uint8_t* r1 = src + ay.s1 * rs;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int t0 = r0[ax.s0 * 3 + c] * ax.a0 + r0[ax.s1 * 3 + c] * ax.a1;
            const int t1 = r1[ax.s0 * 3 + c] * ax.a0 + r1[ax.s1 * 3 + c] * ax.a1;
            rgb[2 - c] = (((ay.a0 * (t0 >> 4)) >> 16) + ((ay.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
        }
    }
}

// EAGLE_PREC_F32S tensors: a value v is stored as hi = rn(16 v), lo = rn(16 v - hi) (binary16), 8 channels = [hi x 8][lo x 8] (32 bytes);
// hi + lo has at most 23 significant bits, so (hi + lo) / 16 is exact in fp32 and re-splitting a loaded value returns the same pair value.
struct SplitT { char b[4]; };
__device__ __forceinline__ void split8_store(void* dst, const float* v)
{
    half8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float s = __builtin_amdgcn_fmed3f(v[i] * 16.0f, -65504.0f, 65504.0f); hi[i] = (_Float16)s; lo[i] = (_Float16)(s - (float)hi[i]); }      // saturating, as split_store4
    *(half8*)dst = hi; *((half8*)dst + 1) = lo;
}

template <typename T>
__device__ __forceinline__ void store_px(const TView& v, size_t pix, float r, float g, float b)
{
    if constexpr (sizeof(T) == 4 && !__is_same(T, float)) {
        const float o[8] = {r, g, b, 0.f, 0.f, 0.f, 0.f, 0.f};
        split8_store((char*)v.p + (pix * v.cs + v.off) * 4, o);
    } else if constexpr (sizeof(T) == 2) {
        half8 o = {(_Float16)r, (_Float16)g, (_Float16)b, 0, 0, 0, 0, 0};
        *(half8*)((_Float16*)v.p + pix * v.cs + v.off) = o;
    } else {
        *(float4*)((float*)v.p + pix * v.cs + v.off) = make_float4(r, g, b, 0.f);
    }
}

template <typename T, typename TD = T>      // T: element type of the key-point tensor, TD: of the detector tensor (a mixed handle runs its two networks in two families)
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* bgr, int n, int h, int w, TView kp, TView det, LetterBox lb, int which)
{
    const size_t fsz = (size_t)h * w * 3;
    const int kp_px = kp.h * kp.w, det_px = det.h * det.w;
    const int lo = (which & 1) ? 0 : kp_px, hi = (which & 2) ? kp_px + det_px : kp_px;      // which: 1 = key-point tensor, 2 = detector tensor
    const int per = hi - lo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * per; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / per);
        int r = lo + (int)(i - (size_t)f * per);
        const uint8_t* src = bgr + f * fsz;
        int rgb[3];
        if (r < kp_px) {
            const int dy = r / kp.w, dx = r - dy * kp.w;
            resize_px(src, h, w, kp.h, kp.w, dy, dx, rgb);
            const float m0 = 0x1.eeb334p+6f, m1 = 0x1.d11eb8p+6f, m2 = 0x1.9e1eb8p+6f;      // f32(mean)*255
            const float s0 = 0x1.18926cp-6f, s1 = 0x1.1ed5bp-6f, s2 = 0x1.1d8f56p-6f;       // 1/(f32(std)*255)
            store_px<T>(kp, (size_t)f * kp_px + r, ((float)rgb[0] - m0) * s0, ((float)rgb[1] - m1) * s1, ((float)rgb[2] - m2) * s2);
        } else {
            r -= kp_px;
            const int dy = r / det.w, dx = r - dy * det.w;
            const int yy = dy - lb.top, xx = dx - lb.left;
            if (yy >= 0 && yy < lb.new_h && xx >= 0 && xx < lb.new_w) resize_px(src, h, w, lb.new_h, lb.new_w, yy, xx, rgb);
            else rgb[0] = rgb[1] = rgb[2] = 114;
            store_px<TD>(det, (size_t)f * det_px + r, (float)rgb[0] / 255.0f, (float)rgb[1] / 255.0f, (float)rgb[2] / 255.0f);
        }
    }
}

LetterBox letterbox_geometry(int h, int w, int imgsz, int square)
{
    // ultralytics LetterBox(new_shape=imgsz, auto=True, stride=32, center=True, scaleup=True); square: auto=False (the static imgsz x imgsz input of an exported
    // ONNX detector, cm.py:54-55): the padding is NOT reduced modulo the stride
    const double r = std::min((double)imgsz / h, (double)imgsz / w);
    LetterBox lb;
    lb.new_w = (int)nearbyint(w * r); lb.new_h = (int)nearbyint(h * r);
    double dw = square ? (imgsz - lb.new_w) : (imgsz - lb.new_w) % 32, dh = square ? (imgsz - lb.new_h) : (imgsz - lb.new_h) % 32;
    dw /= 2; dh /= 2;
    lb.top = (int)nearbyint(dh - 0.1); lb.left = (int)nearbyint(dw - 0.1);
    const int bottom = (int)nearbyint(dh + 0.1), right = (int)nearbyint(dw + 0.1);
    lb.out_h = lb.new_h + lb.top + bottom; lb.out_w = lb.new_w + lb.left + right;
    return lb;
}

template <typename T>
static void preprocess_launch_det(int det_precision, dim3 grid, hipStream_t s, const uint8_t* d_bgr, int n, int h, int w, const TView& kp, const TView& det, const LetterBox& lb, int which)
{
    if (det_precision == EAGLE_PREC_F16) hipLaunchKernelGGL((preprocess_kernel<T, _Float16>), grid, dim3(256), 0, s, d_bgr, n, h, w, kp, det, lb, which);
    else if (det_precision == EAGLE_PREC_F32S) hipLaunchKernelGGL((preprocess_kernel<T, SplitT>), grid, dim3(256), 0, s, d_bgr, n, h, w, kp, det, lb, which);
    else hipLaunchKernelGGL((preprocess_kernel<T, float>), grid, dim3(256), 0, s, d_bgr, n, h, w, kp, det, lb, which);
}

void preprocess_launch(int precision, const uint8_t* d_bgr, int n, int h, int w, const TView& kp, const TView& det,
                       const LetterBox& lb, hipStream_t s, int which, int det_precision)
{
    const size_t total = (size_t)n * (((which & 1) ? kp.h * kp.w : 0) + ((which & 2) ? det.h * det.w : 0));
    if (total == 0) return;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    if (det_precision >= 0 && det_precision != precision) {      // mixed handle: ONE launch writes both tensors, each in its network's format (round 5; two launches before)
        if (precision == EAGLE_PREC_F16) preprocess_launch_det<_Float16>(det_precision, dim3(blocks), s, d_bgr, n, h, w, kp, det, lb, which);
        else if (precision == EAGLE_PREC_F32S) preprocess_launch_det<SplitT>(det_precision, dim3(blocks), s, d_bgr, n, h, w, kp, det, lb, which);
        else preprocess_launch_det<float>(det_precision, dim3(blocks), s, d_bgr, n, h, w, kp, det, lb, which);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (precision == EAGLE_PREC_F16) hipLaunchKernelGGL(preprocess_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, d_bgr, n, h, w, kp, det, lb, which);
    else if (precision == EAGLE_PREC_F32S) hipLaunchKernelGGL(preprocess_kernel<SplitT>, dim3(blocks), dim3(256), 0, s, d_bgr, n, h, w, kp, det, lb, which);
    else hipLaunchKernelGGL(preprocess_kernel<float>, dim3(blocks), dim3(256), 0, s, d_bgr, n, h, w, kp, det, lb, which);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// vector helpers: VEC channels per thread (8 fp16 / 4 fp32)
// ------------------------------------------------------------------------------------------------------------
#ifndef EAGLE_FUSE_NT
#define EAGLE_FUSE_NT 0      // 1: fuse_sum's split-family output stores non-temporal (developer measurement, round 5)
#endif
template <typename T> struct Vec;
template <> struct Vec<_Float16> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const void* p, size_t e) { half8 h = *(const half8*)((const _Float16*)p + e); for (int i = 0; i < 8; ++i) v[i] = (float)h[i]; }
    __device__ __forceinline__ void store(void* p, size_t e) const { half8 h; for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i]; *(half8*)((_Float16*)p + e) = h; }
};
template <> struct Vec<SplitT> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const void* p, size_t e)
    {
        const half8 hi = *(const half8*)((const char*)p + e * 4), lo = *(const half8*)((const char*)p + e * 4 + 16);
        for (int i = 0; i < 8; ++i) v[i] = (float)hi[i] + (float)lo[i];      // = 16 x the value: every user of this type (bilinear sum, max, copy,
    }                                                                          // ReLU) is linear or monotone, and scaling by 2^4 commutes with fp32 rounding
    __device__ __forceinline__ void store(void* p, size_t e) const
    {
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float s = __builtin_amdgcn_fmed3f(v[i], -65504.0f, 65504.0f); hi[i] = (_Float16)s; lo[i] = (_Float16)(s - (float)hi[i]); }
        *(half8*)((char*)p + e * 4) = hi; *((half8*)((char*)p + e * 4) + 1) = lo;
    }
    __device__ __forceinline__ void store_nt(void* p, size_t e) const      // non-temporal: an output nothing re-reads inside the launch (fuse_sum; EAGLE_FUSE_NT)
    {
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float s = __builtin_amdgcn_fmed3f(v[i], -65504.0f, 65504.0f); hi[i] = (_Float16)s; lo[i] = (_Float16)(s - (float)hi[i]); }
        __builtin_nontemporal_store(hi, (half8*)((char*)p + e * 4)); __builtin_nontemporal_store(lo, (half8*)((char*)p + e * 4) + 1);
    }
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const void* p, size_t e) { float4 f = *(const float4*)((const float*)p + e); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
    __device__ __forceinline__ void store(void* p, size_t e) const { *(float4*)((float*)p + e) = make_float4(v[0], v[1], v[2], v[3]); }
};

// ------------------------------------------------------------------------------------------------------------
// K4: y = relu(((base + up(z0)) + up(z1)) + up(z2)), bilinear align_corners=True
//     (F.interpolate + sum + ReLU, eagle/models/keypoint_hrnet.py:290-309)
// ------------------------------------------------------------------------------------------------------------
struct FuseArgs { TView base, y; TView z[3]; float sh[3], sw[3]; int n_up; int relu; unsigned* sat; };     // sh / sw: (z.h - 1) / (H - 1), (z.w - 1) / (W - 1) (fp32 division, done once on the host)

// One workgroup = 256 (column, channel-group) items of FUSE_ROWS consecutive output rows of one frame.  The kernel is not bound by HBM but by what it
// pulls through the texture path: 1 + 4 * NUP operand loads per 32-byte output (13 with three low-resolution operands), almost all of them L2 hits on
// rows that 2 - 8 neighbouring output rows share.  Walking down the rows inside the thread keeps the two low-resolution rows of every operand in
// registers and loads a row only when the bilinear footprint moves on: 2 loads per operand per NEW low-resolution row instead of 4 per output row
// (x8 up-sampling: 0.5 per output row instead of 4).  Same arithmetic per output, in the same order: results are bit-identical to the one-row form.
// The row decomposition is scalar, one integer division per thread remains, and the scale factors arrive as arguments.
constexpr int FUSE_ROWS = 8;       // (16: 116.3 / 112.9 -> 119.8 / 115.6 us, same box)
template <typename T, int NUP>
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseArgs a)
{
    constexpr int VN = Vec<T>::N;
    const int H = a.y.h, W = a.y.w, groups = a.y.c / VN;
    const unsigned col = blockIdx.y * 256u + threadIdx.x;
    if (col >= (unsigned)(W * groups)) return;
    // XCD-aware order: workgroup b runs on XCD b % 8, so XCD x takes the contiguous band of row blocks [x * band, (x + 1) * band): the low-resolution
    // rows that neighbouring blocks share are then fetched into ONE L2 instead of several (PMC: 1.9x read amplification without it)
    const int hb = (H + FUSE_ROWS - 1) / FUSE_ROWS;
    const int band = gridDim.x >> 3, rb = (int)(blockIdx.x & 7) * band + (int)(blockIdx.x >> 3);
    if (rb >= a.y.n * hb) return;
    const int n = rb / hb, oy0 = (rb - n * hb) * FUSE_ROWS;
    const int ox = (int)(col / (unsigned)groups), g = (int)(col - (unsigned)ox * (unsigned)groups);
    constexpr int NU = NUP > 0 ? NUP : 1;
    Vec<T> p00[NU], p01[NU], p10[NU], p11[NU];
    float lx1[NU];
    int cy0[NU], cy1[NU];
    unsigned zc0[NU], zc1[NU];                            // element offsets of the two columns within a low-resolution row (channel group included)
#pragma unroll
    for (int j = 0; j < NUP; ++j) {
        const TView& z = a.z[j];
        const float fx = a.sw[j] * (float)ox;
        const int x0 = (int)fx, x1 = x0 + (x0 < z.w - 1 ? 1 : 0);
        lx1[j] = fx - (float)x0;
        const unsigned co = (unsigned)(z.off + g * VN);
        zc0[j] = (unsigned)x0 * (unsigned)z.cs + co; zc1[j] = (unsigned)x1 * (unsigned)z.cs + co;
        cy0[j] = -1; cy1[j] = -1;
    }
    for (int r = 0; r < FUSE_ROWS; ++r) {
        const int oy = oy0 + r;
        if (oy >= H) break;
        const unsigned pix = ((unsigned)n * H + (unsigned)oy) * (unsigned)W + (unsigned)ox;
        Vec<T> acc; acc.load(a.base.p, (size_t)pix * a.base.cs + a.base.off + g * VN);
        float ly1[NU];
#pragma unroll
        for (int j = 0; j < NUP; ++j) {
            const TView& z = a.z[j];
            const float fy = a.sh[j] * (float)oy;
            const int y0 = __builtin_amdgcn_readfirstlane((int)fy);                 // (uniform: oy and the scale are)
            const int y1 = y0 + (y0 < z.h - 1 ? 1 : 0);
            ly1[j] = fy - (float)y0;
            const unsigned b = (unsigned)n * z.h;
            if (y0 != cy0[j]) {                              // the footprint moved down: the old lower row becomes the upper one where it can
                if (y0 == cy1[j]) { p00[j] = p10[j]; p01[j] = p11[j]; }
                else {
                    const unsigned ro = (b + (unsigned)y0) * (unsigned)z.w * (unsigned)z.cs;
                    p00[j].load(z.p, (size_t)(ro + zc0[j])); p01[j].load(z.p, (size_t)(ro + zc1[j]));
                }
                cy0[j] = y0;
            }
            if (y1 != cy1[j]) {
                if (y1 == y0) { p10[j] = p00[j]; p11[j] = p01[j]; }          // bottom edge: y1 clamps onto y0
                else {
                    const unsigned ro = (b + (unsigned)y1) * (unsigned)z.w * (unsigned)z.cs;
                    p10[j].load(z.p, (size_t)(ro + zc0[j])); p11[j].load(z.p, (size_t)(ro + zc1[j]));
                }
                cy1[j] = y1;
            }
        }
#pragma unroll
        for (int j = 0; j < NUP; ++j) {
            const float ly0 = 1.0f - ly1[j], lx0 = 1.0f - lx1[j];
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                const float top = fmaf(lx1[j], p01[j].v[k], lx0 * p00[j].v[k]);
                const float bot = fmaf(lx1[j], p11[j].v[k], lx0 * p10[j].v[k]);
                acc.v[k] = acc.v[k] + fmaf(ly1[j], bot, ly0 * top);
            }
        }
        if (a.relu)
#pragma unroll
            for (int k = 0; k < VN; ++k) acc.v[k] = acc.v[k] > 0.f ? acc.v[k] : 0.f;
        if constexpr (EAGLE_FUSE_NT && std::is_same<T, SplitT>::value) acc.store_nt(a.y.p, (size_t)pix * a.y.cs + a.y.off + g * VN);
        else acc.store(a.y.p, (size_t)pix * a.y.cs + a.y.off + g * VN);
        if constexpr (__is_same(T, SplitT)) {              // (Vec<SplitT> holds 16 x the value: the format's range is |16 v| <= 65504)
            float m = 0.0f;
#pragma unroll
            for (int k = 0; k < VN; ++k) m = __builtin_fmaxf(m, __builtin_fabsf(acc.v[k]));
            if (a.sat != nullptr && m > 65504.0f) atomicAdd(a.sat + n, 1u);
        }
    }
}

static int ew_blocks(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 256 * 16); }

void fuse_sum_launch(const TView& base, const FuseUp* ups, int n_up, int relu, const TView& y, hipStream_t s, unsigned* sat)
{
    FuseArgs a; a.base = base; a.y = y; a.n_up = n_up; a.relu = relu; a.sat = sat;
    for (int i = 0; i < n_up; ++i) {
        a.z[i] = ups[i].z;
        a.sh[i] = (y.h > 1) ? (float)(ups[i].z.h - 1) / (float)(y.h - 1) : 0.f;
        a.sw[i] = (y.w > 1) ? (float)(ups[i].z.w - 1) / (float)(y.w - 1) : 0.f;
    }
    const int vn = y.f32 == 1 ? 4 : 8;
    const size_t total = (size_t)y.n * y.h * y.w * (y.c / vn);
    size_t biggest = (size_t)y.n * y.h * y.w * std::max(y.cs, base.cs);
    for (int i = 0; i < n_up; ++i) biggest = std::max(biggest, (size_t)ups[i].z.n * ups[i].z.h * ups[i].z.w * ups[i].z.cs);
    if (biggest >= ((size_t)1 << 31)) fail(EAGLE_E_INVALID, "fuse: a tensor of %d frames reaches 2^31 elements; use a smaller device batch", y.n);
    (void)total;
    const dim3 grid((unsigned)((y.n * ((y.h + FUSE_ROWS - 1) / FUSE_ROWS) + 7) / 8 * 8), (unsigned)((y.w * (y.c / vn) + 255) / 256));
#define FUSE_LAUNCH(T_) \
    switch (n_up) { \
    case 0: hipLaunchKernelGGL((fuse_sum_kernel<T_, 0>), grid, dim3(256), 0, s, a); break; \
    case 1: hipLaunchKernelGGL((fuse_sum_kernel<T_, 1>), grid, dim3(256), 0, s, a); break; \
    case 2: hipLaunchKernelGGL((fuse_sum_kernel<T_, 2>), grid, dim3(256), 0, s, a); break; \
    default: hipLaunchKernelGGL((fuse_sum_kernel<T_, 3>), grid, dim3(256), 0, s, a); break; \
    }
    if (n_up > 3) fail(EAGLE_E_INVALID, "fuse: at most three low-resolution operands");
    if (y.f32 != base.f32) fail(EAGLE_E_INVALID, "fuse: operand formats differ");
    for (int i = 0; i < n_up; ++i) if (ups[i].z.f32 != y.f32) fail(EAGLE_E_INVALID, "fuse: operand formats differ");
    if (y.f32 == 1) { FUSE_LAUNCH(float) } else if (y.f32 == 2) { FUSE_LAUNCH(SplitT) } else { FUSE_LAUNCH(_Float16) }
#undef FUSE_LAUNCH
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// K8: SPPF MaxPool2d(5,1,2) and nearest x2 upsample, reading/writing channel slices of concat buffers
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool5_kernel(TView x, TView y)
{
    constexpr int VN = Vec<T>::N;
    const int H = x.h, W = x.w, groups = x.c / VN;
    const size_t total = (size_t)x.n * H * W * groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const size_t pix = i / groups;
        const int ox = (int)(pix % W), oy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
        Vec<T> m;
        for (int k = 0; k < VN; ++k) m.v[k] = -INFINITY;
        for (int dy = -2; dy <= 2; ++dy) {
            const int iy = oy + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = -2; dx <= 2; ++dx) {
                const int ix = ox + dx;
                if (ix < 0 || ix >= W) continue;
                Vec<T> v; v.load(x.p, ((size_t)(n * H + iy) * W + ix) * x.cs + x.off + g * VN);
                for (int k = 0; k < VN; ++k) m.v[k] = v.v[k] > m.v[k] ? v.v[k] : m.v[k];
            }
        }
        m.store(y.p, pix * y.cs + y.off + g * VN);
    }
}
void maxpool5_launch(const TView& x, const TView& y, hipStream_t s)
{
    const int vn = x.f32 == 1 ? 4 : 8;
    const size_t total = (size_t)x.n * x.h * x.w * (x.c / vn);
    if (x.f32 == 1) hipLaunchKernelGGL(maxpool5_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, x, y);
    else if (x.f32 == 2) hipLaunchKernelGGL(maxpool5_kernel<SplitT>, dim3(ew_blocks(total)), dim3(256), 0, s, x, y);
    else hipLaunchKernelGGL(maxpool5_kernel<_Float16>, dim3(ew_blocks(total)), dim3(256), 0, s, x, y);
    HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(TView x, TView y)
{
    constexpr int VN = Vec<T>::N;
    const int H = y.h, W = y.w, groups = x.c / VN;
    const size_t total = (size_t)y.n * H * W * groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const size_t pix = i / groups;
        const int ox = (int)(pix % W), oy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
        Vec<T> v; v.load(x.p, ((size_t)(n * x.h + oy / 2) * x.w + ox / 2) * x.cs + x.off + g * VN);
        v.store(y.p, pix * y.cs + y.off + g * VN);
    }
}
void upsample2_launch(const TView& x, const TView& y, hipStream_t s)
{
    const int vn = x.f32 == 1 ? 4 : 8;
    const size_t total = (size_t)y.n * y.h * y.w * (x.c / vn);
    if (x.f32 == 1) hipLaunchKernelGGL(upsample2_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, x, y);
    else if (x.f32 == 2) hipLaunchKernelGGL(upsample2_kernel<SplitT>, dim3(ew_blocks(total)), dim3(256), 0, s, x, y);
    else hipLaunchKernelGGL(upsample2_kernel<_Float16>, dim3(ew_blocks(total)), dim3(256), 0, s, x, y);
    HIP_CHECK(hipGetLastError());
}

// split fp32 -> fp32 (the seam of the mixed-precision detector, EAGLE_DET_PREC_MIXED: trunk in the split family, the last C2f of every level and Detect in the
// exact family): value = (hi + lo) / 16, exact — hi + lo has at most 22 significant bits.  x, y: same n, h, w, c; any channel slices.
__global__ __launch_bounds__(256) void split_to_f32_kernel(TView x, TView y)
{
    const int groups = x.c / 8;
    const size_t total = (size_t)x.n * x.h * x.w * groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const size_t pix = i / groups;
        Vec<SplitT> v; v.load(x.p, pix * x.cs + x.off + g * 8);
        float* o = (float*)y.p + pix * y.cs + y.off + g * 8;
        *(float4*)o = make_float4(v.v[0] * 0.0625f, v.v[1] * 0.0625f, v.v[2] * 0.0625f, v.v[3] * 0.0625f);
        *(float4*)(o + 4) = make_float4(v.v[4] * 0.0625f, v.v[5] * 0.0625f, v.v[6] * 0.0625f, v.v[7] * 0.0625f);
    }
}
void split_to_f32_launch(const TView& x, const TView& y, hipStream_t s)
{
    const size_t total = (size_t)x.n * x.h * x.w * (x.c / 8);
    hipLaunchKernelGGL(split_to_f32_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
// K5: per-channel first maximum of sigmoid(logits) over a pixel range (KeypointModel.get_keypoints,
//     eagle/models/keypoint_hrnet.py:581-593: np.argmax of the sigmoid map = first maximum, row-major).
//     Stage 1 of 2: grid (chunks, n); the per-frame post kernel reduces the chunk partials.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heat_argmax_kernel(TView lg, ArgmaxPart* parts, int chunks)
{
    const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int HW = lg.h * lg.w;
    const int per = (HW + chunks - 1) / chunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const float* base = (const float*)lg.p + (size_t)n * HW * lg.cs + lg.off + c;
    float best = -1.0f; int bi = 0x7fffffff;
    for (int p = p0 + pl; p < p1; p += 4) {
        const float sg = d_sigmoidf(base[(size_t)p * lg.cs]);
        if (sg > best) { best = sg; bi = p; }
    }
    __shared__ float sb[256];
    __shared__ int si[256];
    sb[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < 4; ++k) {
            const float ob = sb[k * 64 + c]; const int oi = si[k * 64 + c];
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        ArgmaxPart r; r.score = best; r.idx = bi;
        parts[((size_t)n * chunks + chunk) * 64 + c] = r;
    }
}
void heat_argmax_launch(const TView& logits, ArgmaxPart* parts, int chunks, hipStream_t s)
{
    hipLaunchKernelGGL(heat_argmax_kernel, dim3(chunks, logits.n), dim3(256), 0, s, logits, parts, chunks);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
