// K16 — appearance embeddings for the tracker: OSNet-x0.25 on player crops of a clip resident in HBM.
// The reference builds boxmot's BotSort with ReID on (eagle/models/coordinate_model.py:66-72, reid_weights "osnet_x0_25_msmt17.pt") and hands it
// the BGR frame at cm.py:577; boxmot crops every high-confidence detection, resizes it to 128 x 256, normalises it and runs torchreid's
// osnet_x0_25 (architecture table: eagle_amd/osnet.py).  Neither package nor checkpoint exists here: restated from the publication, PARITY
// UNPINNED; oracle/reid.py is the torch-CPU statement the tests compare with.
//
// This file holds the kernels the convolution family does not have: crop + resize + normalise, the 7 x 7 stem, 3 x 3 / stride-2 max-pool,
// depthwise 3 x 3 (+ BatchNorm + ReLU), the shared channel gate with the four-stream sum, 2 x 2 average pool, and the head (global average
// -> Linear(128, 512) -> BatchNorm1d -> ReLU).  All 1 x 1 convolutions (three quarters of the multiply-adds) run on the exact fp32 MFMA
// convolution kernels (conv_f32_kernel); the schedule is built in runtime.hip::build_reid.  fp32 NHWC throughout, channel counts padded to 16
// (padding channels are written as zeros by every kernel here).  The network is 0.08 GMAC per crop — a few per cent of a frame's HRNet —
// and only runs when the caller asks for track identities with appearance, so these kernels are written for clarity, one thread per output
// element, not for a roofline.
#include "common.h"
#include "dmath.h"
#include "resize.h"

namespace eagle {

// crop i = frame[y1:y2, x1:x2] (EagleCrop) -> cv2.resize(..., (128, 256), INTER_LINEAR) -> BGR2RGB -> / 255 -> (v - mean) / std, 4 floats per pixel
__global__ __launch_bounds__(256) void reid_crop_kernel(const uint8_t* bgr, int n_frames, int fh, int fw, const EagleCrop* crops, int n, TView out)
{
    const int OH = out.h, OW = out.w;
    const size_t total = (size_t)n * OH * OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / ((size_t)OH * OW));
        const int r = (int)(i - (size_t)c * OH * OW), dy = r / OW, dx = r - dy * OW;
        const EagleCrop k = crops[c];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const int sh = k.y2 - k.y1, sw = k.x2 - k.x1;
        if (k.frame >= 0 && k.frame < n_frames && sh > 0 && sw > 0 && k.x1 >= 0 && k.y1 >= 0 && k.x2 <= fw && k.y2 <= fh) {
            int rgb[3];
            resize_px_strided(bgr + ((size_t)k.frame * fh + k.y1) * fw * 3 + (size_t)k.x1 * 3, (size_t)fw * 3, sh, sw, OH, OW, dy, dx, rgb);
            o.x = ((float)rgb[0] / 255.0f - 0.485f) / 0.229f;
            o.y = ((float)rgb[1] / 255.0f - 0.456f) / 0.224f;
            o.z = ((float)rgb[2] / 255.0f - 0.406f) / 0.225f;
        }
        *(float4*)((float*)out.p + i * out.cs + out.off) = o;
    }
}
void reid_crop_launch(const uint8_t* d_bgr, int n_frames, int fh, int fw, const EagleCrop* d_crops, int n, const TView& out, hipStream_t s)
{
    if (n <= 0) return;
    const size_t total = (size_t)n * out.h * out.w;
    hipLaunchKernelGGL(reid_crop_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, d_bgr, n_frames, fh, fw, d_crops, n, out);
    HIP_CHECK(hipGetLastError());
}

// 7 x 7, stride 2, pad 3, 3 -> 16 channels, BatchNorm folded into (w, b), ReLU.  w: [7][7][3][16]; one thread per output pixel
__global__ __launch_bounds__(256) void reid_conv7_kernel(TView x, const float* __restrict__ w, const float* __restrict__ b, TView y, int n)
{
    __shared__ float sw[7 * 7 * 3 * 16];
    for (int i = threadIdx.x; i < 7 * 7 * 3 * 16; i += 256) sw[i] = w[i];
    __syncthreads();
    const size_t total = (size_t)n * y.h * y.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / ((size_t)y.h * y.w));
        const int r = (int)(i - (size_t)c * y.h * y.w), oy = r / y.w, ox = r - oy * y.w;
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = oy * 2 - 3 + ky;
            if (iy < 0 || iy >= x.h) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = ox * 2 - 3 + kx;
                if (ix < 0 || ix >= x.w) continue;
                const float4 v = *(const float4*)((const float*)x.p + ((size_t)(c * x.h + iy) * x.w + ix) * x.cs + x.off);
                const float* wk = sw + (ky * 7 + kx) * 48;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fmaf(v.z, wk[32 + k], fmaf(v.y, wk[16 + k], fmaf(v.x, wk[k], acc[k])));
            }
        }
        float* o = (float*)y.p + i * y.cs + y.off;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float t = acc[k] + b[k]; o[k] = t > 0.f ? t : 0.f; }
    }
}
void reid_conv7_launch(const TView& x, const float* w, const float* b, const TView& y, int n, hipStream_t s)
{
    if (n <= 0) return;
    const size_t total = (size_t)n * y.h * y.w;
    hipLaunchKernelGGL(reid_conv7_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, x, w, b, y, n);
    HIP_CHECK(hipGetLastError());
}

// generic element-wise frame for the small ops: one thread per (pixel, 4-channel group)
template <class F>
__global__ __launch_bounds__(256) void reid_px4_kernel(int n, int H, int W, int C4, F f)
{
    const size_t total = (size_t)n * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % C4);
        const size_t p = i / C4;
        const int ox = (int)(p % W), oy = (int)((p / W) % H), c = (int)(p / ((size_t)W * H));
        f(c, oy, ox, g);
    }
}
template <class F> static void px4_launch(int n, int H, int W, int C4, hipStream_t s, F f)
{
    if (n <= 0) return;
    const size_t total = (size_t)n * H * W * C4;
    hipLaunchKernelGGL(reid_px4_kernel<F>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, n, H, W, C4, f);
    HIP_CHECK(hipGetLastError());
}
__device__ __forceinline__ float4 ld4(const TView& v, int c, int y, int x, int g) { return *(const float4*)((const float*)v.p + ((size_t)(c * v.h + y) * v.w + x) * v.cs + v.off + g * 4); }
__device__ __forceinline__ void st4(const TView& v, int c, int y, int x, int g, float4 o) { *(float4*)((float*)v.p + ((size_t)(c * v.h + y) * v.w + x) * v.cs + v.off + g * 4) = o; }

void reid_maxpool3s2_launch(const TView& x, const TView& y, int n, hipStream_t s)       // MaxPool2d(3, stride 2, padding 1)
{
    px4_launch(n, y.h, y.w, y.c / 4, s, [=] __device__(int c, int oy, int ox, int g) {
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= x.h) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= x.w) continue;
                const float4 v = ld4(x, c, iy, ix, g);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        st4(y, c, oy, ox, g, m);
    });
}

void reid_avgpool2_launch(const TView& x, const TView& y, int n, hipStream_t s)          // AvgPool2d(2, stride 2)
{
    px4_launch(n, y.h, y.w, y.c / 4, s, [=] __device__(int c, int oy, int ox, int g) {
        const float4 a = ld4(x, c, 2 * oy, 2 * ox, g), b = ld4(x, c, 2 * oy, 2 * ox + 1, g), d = ld4(x, c, 2 * oy + 1, 2 * ox, g), e = ld4(x, c, 2 * oy + 1, 2 * ox + 1, g);
        st4(y, c, oy, ox, g, make_float4(((a.x + b.x) + (d.x + e.x)) * 0.25f, ((a.y + b.y) + (d.y + e.y)) * 0.25f, ((a.z + b.z) + (d.z + e.z)) * 0.25f, ((a.w + b.w) + (d.w + e.w)) * 0.25f));
    });
}

// depthwise 3 x 3 (pad 1) + folded BatchNorm + ReLU.  w: [9][Cpad] (already multiplied by the BN scale), b: [Cpad]; padding channels have zero weights and bias
void reid_dw3_launch(const TView& x, const float* w, const float* b, const TView& y, int n, hipStream_t s)
{
    const int C = y.c;
    px4_launch(n, y.h, y.w, C / 4, s, [=] __device__(int c, int oy, int ox, int g) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
            if (iy < 0 || iy >= x.h) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox - 1 + kx;
                if (ix < 0 || ix >= x.w) continue;
                const float4 v = ld4(x, c, iy, ix, g);
                const float4 k = *(const float4*)(w + (ky * 3 + kx) * C + g * 4);
                acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y); acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
            }
        }
        const float4 bb = *(const float4*)(b + g * 4);
        st4(y, c, oy, ox, g, make_float4(fmaxf(acc.x + bb.x, 0.f), fmaxf(acc.y + bb.y, 0.f), fmaxf(acc.z + bb.z, 0.f), fmaxf(acc.w + bb.w, 0.f)));
    });
}

// ChannelGate, shared by the four streams of an OSBlock: per (crop, stream) the channel means -> fc1 (+ bias) -> ReLU -> fc2 (+ bias) -> sigmoid.
// g[(crop * 4 + stream) * Cpad + c]; one workgroup per (crop, stream)
struct GateArgs { TView s[4]; const float *w1, *b1, *w2, *b2; int c_real, r; float* g; };
__global__ __launch_bounds__(256) void reid_gate_kernel(GateArgs a)
{
    __shared__ float part[256], mean[128], hid[8];
    const int crop = blockIdx.x >> 2, k = blockIdx.x & 3;
    const TView& x = a.s[k];
    const int HW = x.h * x.w, C = x.c;
    // thread t sums channel (t % C) over pixels t / C, t / C + 256 / C, ...   (C in {16, 32}: 256 % C == 0)
    const int ch = threadIdx.x % C, lane = threadIdx.x / C, step = 256 / C;
    float sum = 0.f;
    for (int p = lane; p < HW; p += step) sum += ((const float*)x.p)[((size_t)crop * HW + p) * x.cs + x.off + ch];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int l = 0; l < step; ++l) t += part[l * C + threadIdx.x];
        mean[threadIdx.x] = t / (float)HW;
    }
    __syncthreads();
    if (threadIdx.x < a.r) {
        float t = a.b1[threadIdx.x];
        for (int c = 0; c < a.c_real; ++c) t = fmaf(a.w1[threadIdx.x * a.c_real + c], mean[c], t);
        hid[threadIdx.x] = t > 0.f ? t : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < C) {
        float o = 0.f;
        if (threadIdx.x < a.c_real) {
            float t = a.b2[threadIdx.x];
            for (int j = 0; j < a.r; ++j) t = fmaf(a.w2[threadIdx.x * a.r + j], hid[j], t);
            o = d_sigmoidf(t);
        }
        a.g[((size_t)crop * 4 + k) * C + threadIdx.x] = o;
    }
}
void reid_gate_launch(const TView* streams, const float* w1, const float* b1, const float* w2, const float* b2, int c_real, int r, float* g, const TView& y, int n, hipStream_t s)
{
    if (n <= 0) return;
    GateArgs a; for (int k = 0; k < 4; ++k) a.s[k] = streams[k];
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.c_real = c_real; a.r = r; a.g = g;
    if (y.c > 128 || 256 % y.c || r > 8) fail(EAGLE_E_INVALID, "reid gate: unsupported channel count %d / reduction %d", y.c, r);
    hipLaunchKernelGGL(reid_gate_kernel, dim3(n * 4), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
    const TView s0 = streams[0], s1 = streams[1], s2 = streams[2], s3 = streams[3];
    const int C = y.c;
    px4_launch(n, y.h, y.w, C / 4, s, [=] __device__(int c, int oy, int ox, int gq) {                  // x2 = sum_k gate(stream_k)
        const float* gg = g + (size_t)c * 4 * C + gq * 4;
        const float4 v0 = ld4(s0, c, oy, ox, gq), v1 = ld4(s1, c, oy, ox, gq), v2 = ld4(s2, c, oy, ox, gq), v3 = ld4(s3, c, oy, ox, gq);
        const float4 g0 = *(const float4*)gg, g1 = *(const float4*)(gg + C), g2 = *(const float4*)(gg + 2 * C), g3 = *(const float4*)(gg + 3 * C);
        st4(y, c, oy, ox, gq, make_float4(((v0.x * g0.x + v1.x * g1.x) + v2.x * g2.x) + v3.x * g3.x, ((v0.y * g0.y + v1.y * g1.y) + v2.y * g2.y) + v3.y * g3.y,
                                          ((v0.z * g0.z + v1.z * g1.z) + v2.z * g2.z) + v3.z * g3.z, ((v0.w * g0.w + v1.w * g1.w) + v2.w * g2.w) + v3.w * g3.w));
    });
}

// head: global average over the map -> Linear(C, 512) (+ bias) -> BatchNorm1d (folded into w, b on the host) -> ReLU.  One workgroup per crop
__global__ __launch_bounds__(256) void reid_head_kernel(TView x, const float* __restrict__ w, const float* __restrict__ b, float* feats, int dim)
{
    __shared__ float part[256], mean[128];
    const int crop = blockIdx.x, HW = x.h * x.w, C = x.c;
    const int ch = threadIdx.x % C, lane = threadIdx.x / C, step = 256 / C;
    float sum = 0.f;
    for (int p = lane; p < HW; p += step) sum += ((const float*)x.p)[((size_t)crop * HW + p) * x.cs + x.off + ch];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int l = 0; l < step; ++l) t += part[l * C + threadIdx.x];
        mean[threadIdx.x] = t / (float)HW;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < dim; o += 256) {
        float t = b[o];
        for (int c = 0; c < C; ++c) t = fmaf(w[(size_t)o * C + c], mean[c], t);
        feats[(size_t)crop * dim + o] = t > 0.f ? t : 0.f;
    }
}
void reid_head_launch(const TView& x, const float* w, const float* b, float* feats, int dim, int n, hipStream_t s)
{
    if (n <= 0) return;
    if (x.c > 128 || 256 % x.c) fail(EAGLE_E_INVALID, "reid head: unsupported channel count %d", x.c);
    hipLaunchKernelGGL(reid_head_kernel, dim3(n), dim3(256), 0, s, x, w, b, feats, dim);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
