// K15 — team colours (second half): Processor.detect_color of the reference's post-processor
// (eagle/processor.py:466-503; "This is pretty slow", :405) for every player crop of a clip that is resident in HBM.
// One workgroup per crop: 2-means segmentation of the crop's RGB pixels (the reference calls scikit-learn's KMeans(n_clusters=2,
// random_state=0)), the cluster that owns the majority of the four crop corners is the background, and the other cluster's pixels are
// counted per colour range of proc.py:10-23 in cv2's 8-bit HSV (hue 0..180, table-driven fixed point like hue180 in geom.hip).
// Deviation, stated: instead of ONE k-means++ start drawn from numpy's RandomState(0), Lloyd's iterations run to their fixed point (exact
// integer sums) from TWO deterministic starts (farthest-point pair; principal-axis split) and the partition with the smaller
// within-cluster sum of squares is kept.  Measured against the reference's own outputs (tests/golden/team_golden.json, sklearn's real
// KMeans): 105 of 108 crops identical in every count; the other three have two fixed points and sklearn's random start picked the worse.
#include "common.h"

namespace eagle {

__device__ __forceinline__ void bgr2hsv_px(int b, int g, int r, int* ho, int* so, int* vo)
{
    int v = b, vmin = b;
    if (g > v) v = g; if (r > v) v = r;
    if (g < vmin) vmin = g; if (r < vmin) vmin = r;
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int sdiv = v ? (int)rint((double)(255 << 12) / (double)v) : 0;
    const int hdiv = diff ? (int)rint((double)(180 << 12) / (6. * (double)diff)) : 0;
    const int s = (diff * sdiv + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    *ho = hh & 255; *so = s & 255; *vo = v;
}

// colour ranges of proc.py:10-23 in output order (red2 is merged into red): {h_lo, s_lo, v_lo, h_hi, s_hi, v_hi}
__constant__ unsigned char TEAM_RANGES[12][6] = {
    {0, 100, 100, 10, 255, 255}, {11, 100, 100, 25, 255, 255}, {26, 100, 100, 35, 255, 255}, {36, 100, 100, 85, 255, 255},
    {86, 100, 100, 95, 255, 255}, {96, 100, 100, 125, 255, 255}, {126, 100, 100, 145, 255, 255}, {146, 100, 100, 159, 255, 255},
    {0, 0, 200, 180, 30, 255}, {0, 0, 50, 180, 30, 200}, {0, 0, 0, 180, 255, 50}, {160, 100, 100, 179, 255, 255} /* red2 -> slot 0 */};

struct TeamArgs { const uint8_t* bgr; int n_frames, fh, fw; const EagleCrop* crops; int* counts; };

__global__ __launch_bounds__(256) void team_color_kernel(TeamArgs a)
{
    __shared__ unsigned long long s_sum[8];          // per cluster: r, g, b, count
    __shared__ unsigned long long s_best; __shared__ unsigned s_idx;
    __shared__ double s_c[2][3];
    __shared__ int s_cnt[12];
    __shared__ int s_flag;
    const EagleCrop c = a.crops[blockIdx.x];
    const int tid = threadIdx.x;
    int* out = a.counts + (size_t)blockIdx.x * 12;
    const int w = c.x2 - c.x1, h = c.y2 - c.y1;
    if (tid < 12) out[tid] = 0;
    if (w <= 0 || h <= 0 || c.frame < 0 || c.frame >= a.n_frames || c.x1 < 0 || c.y1 < 0 || c.x2 > a.fw || c.y2 > a.fh) return;
    const int n = w * h;
    const uint8_t* base = a.bgr + ((size_t)c.frame * a.fh + c.y1) * a.fw * 3 + (size_t)c.x1 * 3;
    auto px = [&](int i, int* r, int* g, int* b) {
        const int yy = i / w, xx = i - yy * w;
        const uint8_t* p = base + ((size_t)yy * a.fw + xx) * 3;
        *b = p[0]; *g = p[1]; *r = p[2];
    };
    // (A) mean, as integer sums
    if (tid < 8) s_sum[tid] = 0;
    __syncthreads();
    {
        unsigned long long sr = 0, sg = 0, sb = 0;
        for (int i = tid; i < n; i += 256) { int r, g, b; px(i, &r, &g, &b); sr += r; sg += g; sb += b; }
        atomicAdd(&s_sum[0], sr); atomicAdd(&s_sum[1], sg); atomicAdd(&s_sum[2], sb);
    }
    __syncthreads();
    const long long SR = (long long)s_sum[0], SG = (long long)s_sum[1], SB = (long long)s_sum[2];
    // (B) first centre: the pixel farthest from the mean (n^2 * distance^2 in exact integers; ties: lowest index); (C) second: farthest from it
    int cr[2] = {0, 0}, cg[2] = {0, 0}, cb[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        if (tid == 0) { s_best = 0; s_idx = 0xFFFFFFFFu; }
        __syncthreads();
        unsigned long long best = 0;
        for (int i = tid; i < n; i += 256) {
            int r, g, b; px(i, &r, &g, &b);
            long long dr, dg, db;
            if (pass == 0) { dr = (long long)n * r - SR; dg = (long long)n * g - SG; db = (long long)n * b - SB; }
            else { dr = r - cr[0]; dg = g - cg[0]; db = b - cb[0]; }
            const unsigned long long d = (unsigned long long)(dr * dr + dg * dg + db * db);
            if (d > best) best = d;
        }
        atomicMax(&s_best, best);
        __syncthreads();
        const unsigned long long gb = s_best;
        for (int i = tid; i < n; i += 256) {
            int r, g, b; px(i, &r, &g, &b);
            long long dr, dg, db;
            if (pass == 0) { dr = (long long)n * r - SR; dg = (long long)n * g - SG; db = (long long)n * b - SB; }
            else { dr = r - cr[0]; dg = g - cg[0]; db = b - cb[0]; }
            if ((unsigned long long)(dr * dr + dg * dg + db * db) == gb) { atomicMin(&s_idx, (unsigned)i); break; }
        }
        __syncthreads();
        { int r, g, b; px((int)s_idx, &r, &g, &b); cr[pass] = r; cg[pass] = g; cb[pass] = b; }
        __syncthreads();
    }
    // (D) Lloyd's iterations to the fixed point (centres are ratios of exact integer sums, so "unchanged" is an exact test), from two
    // deterministic starts: the farthest-point pair above, and the two halves of the crop split along its principal colour axis (a thin
    // bright line through the crop attracts the farthest-point start; k-means++ weights by mass, the principal-axis start does too).
    // The converged pair with the smaller within-cluster sum of squares is kept.
    __shared__ double s_try[2][2][3]; __shared__ double s_sse[2];
    __shared__ unsigned long long s_cov[6];
    if (tid < 6) s_cov[tid] = 0;
    __syncthreads();
    {
        unsigned long long q[6] = {0, 0, 0, 0, 0, 0};
        for (int i = tid; i < n; i += 256) { int r, g, b; px(i, &r, &g, &b); q[0] += r * r; q[1] += g * g; q[2] += b * b; q[3] += r * g; q[4] += r * b; q[5] += g * b; }
        for (int k = 0; k < 6; ++k) atomicAdd(&s_cov[k], q[k]);
    }
    __syncthreads();
    if (tid == 0) {
        const double N = n, mr = SR / N, mg = SG / N, mb = SB / N;
        const double C[3][3] = {{s_cov[0] / N - mr * mr, s_cov[3] / N - mr * mg, s_cov[4] / N - mr * mb},
                                {s_cov[3] / N - mr * mg, s_cov[1] / N - mg * mg, s_cov[5] / N - mg * mb},
                                {s_cov[4] / N - mr * mb, s_cov[5] / N - mg * mb, s_cov[2] / N - mb * mb}};
        double v[3] = {1.0, 1.0, 1.0};
        for (int it = 0; it < 32; ++it) {
            double u[3];
            for (int i = 0; i < 3; ++i) u[i] = C[i][0] * v[0] + C[i][1] * v[1] + C[i][2] * v[2];
            const double nn = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            if (!(nn > 0)) break;
            for (int i = 0; i < 3; ++i) v[i] = u[i] / nn;
        }
        s_try[1][0][0] = v[0]; s_try[1][0][1] = v[1]; s_try[1][0][2] = v[2];      // (axis, handed to the split pass below)
        s_try[1][1][0] = mr; s_try[1][1][1] = mg; s_try[1][1][2] = mb;
    }
    if (tid < 8) s_sum[tid] = 0;
    __syncthreads();
    {
        const double ax = s_try[1][0][0], ay = s_try[1][0][1], az = s_try[1][0][2], mr = s_try[1][1][0], mg = s_try[1][1][1], mb = s_try[1][1][2];
        unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < n; i += 256) {
            int r, g, b; px(i, &r, &g, &b);
            const int l = ((r - mr) * ax + (g - mg) * ay + (b - mb) * az) > 0.0 ? 4 : 0;
            acc[l] += r; acc[l + 1] += g; acc[l + 2] += b; acc[l + 3] += 1;
        }
        for (int k = 0; k < 8; ++k) if (acc[k]) atomicAdd(&s_sum[k], acc[k]);
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < 2; ++k) { s_try[0][k][0] = cr[k]; s_try[0][k][1] = cg[k]; s_try[0][k][2] = cb[k]; }
        for (int k = 0; k < 2; ++k) {
            const unsigned long long cnt = s_sum[4 * k + 3];
            for (int j = 0; j < 3; ++j) s_try[1][k][j] = cnt ? (double)s_sum[4 * k + j] / (double)cnt : (double)(k ? cr[1] : cr[0]);
        }
    }
    __syncthreads();
    for (int start = 0; start < 2; ++start) {
        if (tid == 0) for (int k = 0; k < 2; ++k) for (int j = 0; j < 3; ++j) s_c[k][j] = s_try[start][k][j];
        __syncthreads();
        for (int it = 0; it < 100; ++it) {
            if (tid < 8) s_sum[tid] = 0;
            if (tid == 0) s_flag = 0;
            __syncthreads();
            const double c0r = s_c[0][0], c0g = s_c[0][1], c0b = s_c[0][2], c1r = s_c[1][0], c1g = s_c[1][1], c1b = s_c[1][2];
            unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = tid; i < n; i += 256) {
                int r, g, b; px(i, &r, &g, &b);
                const double d0 = (r - c0r) * (r - c0r) + (g - c0g) * (g - c0g) + (b - c0b) * (b - c0b);
                const double d1 = (r - c1r) * (r - c1r) + (g - c1g) * (g - c1g) + (b - c1b) * (b - c1b);
                const int l = d1 < d0 ? 4 : 0;
                acc[l] += r; acc[l + 1] += g; acc[l + 2] += b; acc[l + 3] += 1;
            }
            for (int k = 0; k < 8; ++k) if (acc[k]) atomicAdd(&s_sum[k], acc[k]);
            __syncthreads();
            if (tid == 0) {
                int changed = 0;
                for (int k = 0; k < 2; ++k) {
                    const unsigned long long cnt = s_sum[4 * k + 3];
                    if (!cnt) continue;
                    for (int j = 0; j < 3; ++j) {
                        const double v = (double)s_sum[4 * k + j] / (double)cnt;
                        if (v != s_c[k][j]) { s_c[k][j] = v; changed = 1; }
                    }
                }
                s_flag = changed;
            }
            __syncthreads();
            if (!s_flag) break;
        }
        // within-cluster sum of squares of this fixed point: sum |p|^2 - sum_k count_k |c_k|^2 (exact sums, double arithmetic in one thread)
        if (tid == 0) {
            double sse = (double)s_cov[0] + (double)s_cov[1] + (double)s_cov[2];
            for (int k = 0; k < 2; ++k) sse -= (double)s_sum[4 * k + 3] * (s_c[k][0] * s_c[k][0] + s_c[k][1] * s_c[k][1] + s_c[k][2] * s_c[k][2]);
            s_sse[start] = sse;
            for (int k = 0; k < 2; ++k) for (int j = 0; j < 3; ++j) s_try[start][k][j] = s_c[k][j];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int best = s_sse[1] < s_sse[0] ? 1 : 0;
        for (int k = 0; k < 2; ++k) for (int j = 0; j < 3; ++j) s_c[k][j] = s_try[best][k][j];
    }
    __syncthreads();
    // (E) corner vote -> background cluster; (F) colour-range counts of the other cluster's pixels
    const double c0r = s_c[0][0], c0g = s_c[0][1], c0b = s_c[0][2], c1r = s_c[1][0], c1g = s_c[1][1], c1b = s_c[1][2];
    auto label = [&](int i) {
        int r, g, b; px(i, &r, &g, &b);
        const double d0 = (r - c0r) * (r - c0r) + (g - c0g) * (g - c0g) + (b - c0b) * (b - c0b);
        const double d1 = (r - c1r) * (r - c1r) + (g - c1g) * (g - c1g) + (b - c1b) * (b - c1b);
        return d1 < d0 ? 1 : 0;
    };
    const int corner[4] = {label(0), label(w - 1), label((h - 1) * w), label(n - 1)};
    const int ones = corner[0] + corner[1] + corner[2] + corner[3];
    const int background = ones > 2 ? 1 : (ones < 2 ? 0 : corner[0]);      // 2-2 tie: the cluster of the top-left corner (the reference's choice depends on sklearn's label numbering there)
    if (tid < 12) s_cnt[tid] = 0;
    __syncthreads();
    int cnt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        if (label(i) == background) continue;
        int r, g, b; px(i, &r, &g, &b);
        int hh, ss, vv;
        bgr2hsv_px(b, g, r, &hh, &ss, &vv);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const unsigned char* q = TEAM_RANGES[k];
            cnt[k == 11 ? 0 : k] += (hh >= q[0] && hh <= q[3] && ss >= q[1] && ss <= q[4] && vv >= q[2] && vv <= q[5]) ? 1 : 0;
        }
        cnt[11] += 1;                                                      // slot 11: pixels of the player cluster
    }
    for (int k = 0; k < 12; ++k) if (cnt[k]) atomicAdd(&s_cnt[k], cnt[k]);
    __syncthreads();
    if (tid < 12) out[tid] = s_cnt[tid];
}

void team_colors_launch(const uint8_t* d_bgr, int n_frames, int fh, int fw, const EagleCrop* d_crops, int n_crops, int* d_counts, hipStream_t s)
{
    TeamArgs a{d_bgr, n_frames, fh, fw, d_crops, d_counts};
    hipLaunchKernelGGL(team_color_kernel, dim3(n_crops), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
