// K17: ECC camera-motion estimation — the `cmc_method="ecc"` default of the BotSort the reference constructs
// (eagle/models/coordinate_model.py:66-72; the frame reaches it at cm.py:577).  boxmot's wrapper: gray frame -> cv2.resize(fx = fy = 0.15)
// -> cv2.findTransformECC(prev, cur, eye(2, 3), MOTION_EUCLIDEAN, (EPS | COUNT, 100, 1e-5), None, 1); restated from the published algorithm
// (Evangelidis & Psarakis 2008; OpenCV video/src/ecc.cpp) in oracle/ecc.py, which this file follows operation by operation:
//   ecc_small_kernel : the clip session's gray level 0 -> the 0.15-scale image of every frame (cv2's 11-bit fixed-point bilinear taps), parallel
//                      over the clip.  HBM: 921,600 B read (sparsely: 4 of every ~44 pixels) + 20,736 B written per 1280x720 frame.
//   ecc_kernel       : one 512-thread workgroup per (template, image) frame pair; all <= 100 Gauss-Newton iterations run inside the launch.
//                      Per iteration three passes over the template's pixels (masked moments -> Hessian / projections -> error projection),
//                      each ending in a workgroup reduction of double accumulators; the warped image, its two warped gradients and the mask
//                      are RE-COMPUTED in every pass from the 20 KB u8 image (L1/L2-resident) instead of being stored: warpAffine's 10-bit
//                      fixed-point coordinates, 1/32-pixel bilinear weights, constant border; gradients 0.5 * (I[x+1] - I[x-1]) with
//                      reflect-101 borders taken on the fly.  All pixel arithmetic is float32 with separate multiplies and adds
//                      (-ffp-contract=off), all sums are float64 over exact float32 products, like Mat::dot.
// The pairs of a clip are independent except after a failed alignment (boxmot keeps the OLD template then): the host side of the library
// (runtime.hip::eagle_clip_motion_ecc) launches all adjacent pairs at once and re-runs the rare pair behind a failure.
// Latency-bound by design (a 108 x 192 image per workgroup, ~10 iterations): 1000 pairs fill the 256 CUs four times over.
#include "common.h"
#include "resize.h"

namespace eagle {

__global__ __launch_bounds__(256) void ecc_small_kernel(const uint8_t* __restrict__ gray, uint8_t* __restrict__ small, int n, int h, int w, int dh, int dw, double inv_scale)
{
    const size_t total = (size_t)n * dh * dw, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % dw), y = (int)((i / dw) % dh), f = (int)(i / ((size_t)dw * dh));
        const uint8_t* s = gray + (size_t)f * h * w;
        // cv2.resize(dsize = (0, 0), fx, fy) maps with 1 / fx, not with ssize / dsize: the two differ whenever 0.15 * size is not an integer
        const ResizeTap ax = resize_tap_scaled(x, inv_scale, w), ay = resize_tap_scaled(y, inv_scale, h);
        const uint8_t* r0 = s + (size_t)ay.s0 * w; const uint8_t* r1 = s + (size_t)ay.s1 * w;
        const int t0 = r0[ax.s0] * ax.a0 + r0[ax.s1] * ax.a1;
        const int t1 = r1[ax.s0] * ax.a0 + r1[ax.s1] * ax.a1;
        small[i] = (uint8_t)((((ay.a0 * (t0 >> 4)) >> 16) + ((ay.a1 * (t1 >> 4)) >> 16) + 2) >> 2);
    }
}

constexpr int ECC_THREADS = 512, ECC_WAVES = ECC_THREADS / 64, ECC_NACC = 13;

struct EccShared {
    double red[ECC_WAVES][ECC_NACC];
    double tot[ECC_NACC];
    float M[6];
    float i_mean, t_mean, lambda;
    float Hinv[9];
    int go;
};

template <int N>
__device__ __forceinline__ void ecc_reduce(double (&acc)[N], EccShared& S, int tid)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[k] = v;
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) S.red[tid >> 6][k] = acc[k];
    }
    __syncthreads();
    if (tid < N) {
        double v = 0.0;
        for (int wv = 0; wv < ECC_WAVES; ++wv) v += S.red[wv][tid];
        S.tot[tid] = v;
    }
    __syncthreads();
}

struct EccPix { float iw, gx, gy; bool in; };

// u8 image tap with warpAffine's constant border
__device__ __forceinline__ float ecc_tap(const uint8_t* __restrict__ img, int h, int w, int y, int x)
{
    return ((unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) ? (float)img[y * w + x] : 0.f;
}
// filter2D(I, [-0.5, 0, 0.5]) at (y, x) with BORDER_REFLECT_101; 0 outside the image (the warp's constant border)
__device__ __forceinline__ float ecc_gx(const uint8_t* __restrict__ img, int h, int w, int y, int x)
{
    if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return 0.f;
    const int xm = x == 0 ? 1 : x - 1, xp = x == w - 1 ? w - 2 : x + 1;
    return 0.5f * ((float)img[y * w + xp] - (float)img[y * w + xm]);
}
__device__ __forceinline__ float ecc_gy(const uint8_t* __restrict__ img, int h, int w, int y, int x)
{
    if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return 0.f;
    const int ym = y == 0 ? 1 : y - 1, yp = y == h - 1 ? h - 2 : y + 1;
    return 0.5f * ((float)img[yp * w + x] - (float)img[ym * w + x]);
}

template <bool GRAD>
__device__ __forceinline__ EccPix ecc_sample(const uint8_t* __restrict__ img, int h, int w, const double (&M)[6], int y, int x)
{
    constexpr int AB = 1024;
    const long long ad = llrint(M[0] * (double)x * AB), bd = llrint(M[3] * (double)x * AB);
    const long long X0 = llrint((M[1] * (double)y + M[2]) * AB), Y0 = llrint((M[4] * (double)y + M[5]) * AB);
    EccPix p;
    {   // INTER_NEAREST mask: round_delta = AB / 2
        const long long sx = (X0 + 512 + ad) >> 10, sy = (Y0 + 512 + bd) >> 10;
        p.in = sx >= 0 && sx < w && sy >= 0 && sy < h;
    }
    const long long X = (X0 + 16 + ad) >> 5, Y = (Y0 + 16 + bd) >> 5;       // 1/32-pixel coordinates
    const long long lx = X >> 5, ly = Y >> 5;
    const int sx = (int)(lx < -4 ? -4 : (lx > w + 4 ? w + 4 : lx)), sy = (int)(ly < -4 ? -4 : (ly > h + 4 ? h + 4 : ly));
    const float fx = (float)(int)(X & 31) * (1.0f / 32), fy = (float)(int)(Y & 31) * (1.0f / 32);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    const float w00 = wy0 * wx0, w01 = wy0 * fx, w10 = fy * wx0, w11 = fy * fx;
    p.iw = ((ecc_tap(img, h, w, sy, sx) * w00 + ecc_tap(img, h, w, sy, sx + 1) * w01) + ecc_tap(img, h, w, sy + 1, sx) * w10) + ecc_tap(img, h, w, sy + 1, sx + 1) * w11;
    if (GRAD) {
        p.gx = ((ecc_gx(img, h, w, sy, sx) * w00 + ecc_gx(img, h, w, sy, sx + 1) * w01) + ecc_gx(img, h, w, sy + 1, sx) * w10) + ecc_gx(img, h, w, sy + 1, sx + 1) * w11;
        p.gy = ((ecc_gy(img, h, w, sy, sx) * w00 + ecc_gy(img, h, w, sy, sx + 1) * w01) + ecc_gy(img, h, w, sy + 1, sx) * w10) + ecc_gy(img, h, w, sy + 1, sx + 1) * w11;
    } else { p.gx = p.gy = 0.f; }
    return p;
}

__device__ inline void ecc_inv3(const float (&Hf)[9], float (&out)[9])
{
    double S[9];
    for (int k = 0; k < 9; ++k) S[k] = (double)Hf[k];
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.0) { for (int k = 0; k < 9; ++k) out[k] = 0.f; return; }
    d = 1.0 / d;
    out[0] = (float)((S[4] * S[8] - S[5] * S[7]) * d);
    out[1] = (float)((S[2] * S[7] - S[1] * S[8]) * d);
    out[2] = (float)((S[1] * S[5] - S[2] * S[4]) * d);
    out[3] = (float)((S[5] * S[6] - S[3] * S[8]) * d);
    out[4] = (float)((S[0] * S[8] - S[2] * S[6]) * d);
    out[5] = (float)((S[2] * S[3] - S[0] * S[5]) * d);
    out[6] = (float)((S[3] * S[7] - S[4] * S[6]) * d);
    out[7] = (float)((S[1] * S[6] - S[0] * S[7]) * d);
    out[8] = (float)((S[0] * S[4] - S[1] * S[3]) * d);
}

__global__ __launch_bounds__(ECC_THREADS) void ecc_kernel(const uint8_t* __restrict__ small, const uint8_t* __restrict__ carry, const int2* __restrict__ pairs,
                                                          EccResult* __restrict__ out, int h, int w, int max_iter, double eps)
{
    __shared__ EccShared S;
    const int tid = threadIdx.x, npx = h * w;
    const int2 pr = pairs[blockIdx.x];
    const uint8_t* T = pr.x < 0 ? carry : small + (size_t)pr.x * npx;
    const uint8_t* I = small + (size_t)pr.y * npx;
    if (tid == 0) { S.M[0] = 1.f; S.M[1] = 0.f; S.M[2] = 0.f; S.M[3] = 0.f; S.M[4] = 1.f; S.M[5] = 0.f; S.go = 1; }
    __syncthreads();
    double rho = -1.0, last_rho = -eps;       // thread 0's copies are the ones that count
    int it = 0, ok = 1;
    while (true) {
        if (tid == 0) S.go = (it < max_iter && fabs(rho - last_rho) >= eps) ? 1 : 0;
        __syncthreads();
        if (!S.go) break;
        ++it;
        double M[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) M[k] = (double)S.M[k];
        const float h0 = S.M[0], h1 = S.M[3];
        // ---- pass A: masked moments of the warped image and the template
        {
            double acc[5] = {0, 0, 0, 0, 0};
            for (int p = tid; p < npx; p += ECC_THREADS) {
                const int y = p / w, x = p - y * w;
                const EccPix s = ecc_sample<false>(I, h, w, M, y, x);
                if (s.in) {
                    const double a = (double)s.iw, t = (double)T[p];
                    acc[0] += 1.0; acc[1] += a; acc[2] += a * a; acc[3] += t; acc[4] += t * t;
                }
            }
            ecc_reduce<5>(acc, S, tid);
        }
        const double n = S.tot[0];
        double i_norm = 0.0, t_norm = 0.0;
        if (tid == 0) {
            if (n == 0.0) { ok = 0; S.go = 0; }
            else {
                const double im = S.tot[1] / n, tm = S.tot[3] / n;
                const double iv = fmax(S.tot[2] / n - im * im, 0.0), tv = fmax(S.tot[4] / n - tm * tm, 0.0);
                S.i_mean = (float)im; S.t_mean = (float)tm;
                i_norm = sqrt(n * iv); t_norm = sqrt(n * tv);
                S.go = 1;
            }
        }
        __syncthreads();
        if (!S.go) break;
        const float i_mean = S.i_mean, t_mean = S.t_mean;
        // ---- pass B: Hessian, correlation, projections of both zero-mean images onto the Jacobian
        {
            double acc[ECC_NACC];
#pragma unroll
            for (int k = 0; k < ECC_NACC; ++k) acc[k] = 0.0;
            for (int p = tid; p < npx; p += ECC_THREADS) {
                const int y = p / w, x = p - y * w;
                const EccPix s = ecc_sample<true>(I, h, w, M, y, x);
                const float izm = s.in ? s.iw - i_mean : s.iw;
                const float tzm = s.in ? (float)T[p] - t_mean : 0.f;
                const float xf = (float)x, yf = (float)y;
                const float hat_x = -(xf * h1) - (yf * h0), hat_y = (xf * h0) - (yf * h1);
                const double j0 = (double)(s.gx * hat_x + s.gy * hat_y), j1 = (double)s.gx, j2 = (double)s.gy;
                const double di = (double)izm, dt = (double)tzm;
                acc[0] += j0 * j0; acc[1] += j0 * j1; acc[2] += j0 * j2; acc[3] += j1 * j1; acc[4] += j1 * j2; acc[5] += j2 * j2;
                acc[6] += dt * di;
                acc[7] += j0 * di; acc[8] += j1 * di; acc[9] += j2 * di;
                acc[10] += j0 * dt; acc[11] += j1 * dt; acc[12] += j2 * dt;
            }
            ecc_reduce<ECC_NACC>(acc, S, tid);
        }
        if (tid == 0) {
            const float Hs[9] = {(float)S.tot[0], (float)S.tot[1], (float)S.tot[2], (float)S.tot[1], (float)S.tot[3], (float)S.tot[4],
                                 (float)S.tot[2], (float)S.tot[4], (float)S.tot[5]};
            float Hinv[9];
            ecc_inv3(Hs, Hinv);
            const double corr = S.tot[6];
            last_rho = rho;
            rho = corr / (i_norm * t_norm);
            S.go = 1;
            if (rho != rho) { ok = 0; S.go = 0; }
            else {
                const float ip[3] = {(float)S.tot[7], (float)S.tot[8], (float)S.tot[9]}, tp[3] = {(float)S.tot[10], (float)S.tot[11], (float)S.tot[12]};
                float iph[3];
                for (int r = 0; r < 3; ++r) iph[r] = (float)(((double)Hinv[3 * r] * ip[0] + (double)Hinv[3 * r + 1] * ip[1]) + (double)Hinv[3 * r + 2] * ip[2]);
                const double lam_n = i_norm * i_norm - (((double)ip[0] * iph[0] + (double)ip[1] * iph[1]) + (double)ip[2] * iph[2]);
                const double lam_d = corr - (((double)tp[0] * iph[0] + (double)tp[1] * iph[1]) + (double)tp[2] * iph[2]);
                if (lam_d <= 0.0) { ok = 0; S.go = 0; }
                else {
                    S.lambda = (float)(lam_n / lam_d);
                    for (int k = 0; k < 9; ++k) S.Hinv[k] = Hinv[k];
                }
            }
        }
        __syncthreads();
        if (!S.go) break;
        const float lambda = S.lambda;
        // ---- pass C: projection of the error image
        {
            double acc[3] = {0, 0, 0};
            for (int p = tid; p < npx; p += ECC_THREADS) {
                const int y = p / w, x = p - y * w;
                const EccPix s = ecc_sample<true>(I, h, w, M, y, x);
                const float izm = s.in ? s.iw - i_mean : s.iw;
                const float tzm = s.in ? (float)T[p] - t_mean : 0.f;
                const float xf = (float)x, yf = (float)y;
                const float hat_x = -(xf * h1) - (yf * h0), hat_y = (xf * h0) - (yf * h1);
                const double j0 = (double)(s.gx * hat_x + s.gy * hat_y), j1 = (double)s.gx, j2 = (double)s.gy;
                const double e = (double)(lambda * tzm - izm);
                acc[0] += j0 * e; acc[1] += j1 * e; acc[2] += j2 * e;
            }
            ecc_reduce<3>(acc, S, tid);
        }
        if (tid == 0) {
            const float ep[3] = {(float)S.tot[0], (float)S.tot[1], (float)S.tot[2]};
            float dp[3];
            for (int r = 0; r < 3; ++r) dp[r] = (float)(((double)S.Hinv[3 * r] * ep[0] + (double)S.Hinv[3 * r + 1] * ep[1]) + (double)S.Hinv[3 * r + 2] * ep[2]);
            const double theta = (double)dp[0] + asin((double)S.M[3]);
            S.M[2] = S.M[2] + dp[1]; S.M[5] = S.M[5] + dp[2];
            S.M[0] = S.M[4] = (float)cos(theta);
            S.M[3] = (float)sin(theta);
            S.M[1] = -S.M[3];
        }
        __syncthreads();
    }
    if (tid == 0) {
        EccResult& r = out[blockIdx.x];
        r.ok = ok; r.iters = it; r.rho = rho;
        for (int k = 0; k < 6; ++k) r.M[k] = S.M[k];
    }
}

void ecc_small_launch(const uint8_t* gray, uint8_t* small, int n, int h, int w, int dh, int dw, double inv_scale, hipStream_t s)
{
    if (n <= 0) return;
    const size_t total = (size_t)n * dh * dw;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(ecc_small_kernel, dim3(blocks), dim3(256), 0, s, gray, small, n, h, w, dh, dw, inv_scale);
    HIP_CHECK(hipGetLastError());
}

void ecc_launch(const uint8_t* small, const uint8_t* carry, const int2* pairs, int n_pairs, EccResult* out, int h, int w, int max_iter, double eps, hipStream_t s)
{
    if (n_pairs <= 0) return;
    hipLaunchKernelGGL(ecc_kernel, dim3(n_pairs), dim3(ECC_THREADS), 0, s, small, carry, pairs, out, h, w, max_iter, eps);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
