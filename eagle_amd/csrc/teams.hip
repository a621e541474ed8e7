// K15 — team colours (second half): Processor.detect_color of the reference's post-processor
// (eagle/processor.py:466-503; "This is pretty slow", :405) for every player crop of a clip that is resident in HBM.
// One workgroup per crop: 2-means segmentation of the crop's RGB pixels (the reference calls scikit-learn's KMeans(n_clusters=2,
// random_state=0)), the cluster that owns the majority of the four crop corners is the background, and the other cluster's pixels are
// counted per colour range of proc.py:10-23 in cv2's 8-bit HSV (hue 0..180, table-driven fixed point like hue180 in geom.hip).
// The 2-means is scikit-learn's run restated step by step (k-means++ seeding with RandomState(0)'s first three doubles, Lloyd's iterations with
// the tol rule, final assignment; oracle/colors.py::kmeans2_labels is the same restatement and is pinned to sklearn's own labels on
// several hundred crops): colour counts identical to the reference's own outputs on all crops of tests/golden/team_golden.json.
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
    // (A) k-means++ seeding exactly as scikit-learn 1.7 runs it for KMeans(n_clusters=2, random_state=0): RandomState(0) is created per call,
    //     so its first three doubles are constants.  centre 0 = X[choice(n)] = X[floor(u0 * n)]; two candidates for centre 1 at the
    //     quantiles u1, u2 of the cumulative squared distance to centre 0 (searchsorted, side = left); the candidate with the smaller
    //     potential wins.  Distances between pixels are exact integers, so the cumulative sums and potentials are exact.
    constexpr double U0 = 0.5488135039273248, U1 = 0.7151893663724195, U2 = 0.6027633760716439;
    __shared__ unsigned long long s_part[256];
    __shared__ int s_cand[2];
    __shared__ unsigned long long s_pot[2];
    int i0 = (int)floor(U0 * (double)n); if (i0 > n - 1) i0 = n - 1;
    int c0r_, c0g_, c0b_; px(i0, &c0r_, &c0g_, &c0b_);
    const int chunk = (n + 255) / 256, lo = tid * chunk, hi = min(n, lo + chunk);
    auto dist0 = [&](int i) -> unsigned long long {
        int r, g, b; px(i, &r, &g, &b);
        const long long dr = r - c0r_, dg = g - c0g_, db = b - c0b_;
        return (unsigned long long)(dr * dr + dg * dg + db * db);
    };
    {
        unsigned long long ps = 0;
        for (int i = lo; i < hi; ++i) ps += dist0(i);
        s_part[tid] = ps;
    }
    if (tid < 8) s_sum[tid] = 0;
    __syncthreads();
    if (tid < 2) {                                        // thread t finds candidate t: first index whose cumulative distance reaches u * potential
        unsigned long long pot = 0;
        for (int k = 0; k < 256; ++k) pot += s_part[k];
        const double val = (tid == 0 ? U1 : U2) * (double)pot;
        unsigned long long cum = 0; int k = 0;
        while (k < 255 && (double)(cum + s_part[k]) < val) { cum += s_part[k]; ++k; }
        int i = k * chunk; const int e = min(n, i + chunk);
        for (; i < e; ++i) { cum += dist0(i); if ((double)cum >= val) break; }
        s_cand[tid] = min(i, n - 1);
        s_pot[tid] = 0;
    }
    __syncthreads();
    {
        int ar, ag, ab, br, bg, bb; px(s_cand[0], &ar, &ag, &ab); px(s_cand[1], &br, &bg, &bb);
        unsigned long long pa = 0, pb = 0;
        for (int i = tid; i < n; i += 256) {
            int r, g, b; px(i, &r, &g, &b);
            const unsigned long long d0 = dist0(i);
            const long long x1 = r - ar, y1 = g - ag, z1 = b - ab, x2 = r - br, y2 = g - bg, z2 = b - bb;
            const unsigned long long da = (unsigned long long)(x1 * x1 + y1 * y1 + z1 * z1), db2 = (unsigned long long)(x2 * x2 + y2 * y2 + z2 * z2);
            pa += da < d0 ? da : d0; pb += db2 < d0 ? db2 : d0;
        }
        atomicAdd(&s_pot[0], pa); atomicAdd(&s_pot[1], pb);
        // per-channel sums for the tolerance (mean of the channel variances * 1e-4)
        unsigned long long q[6] = {0, 0, 0, 0, 0, 0};
        for (int i = tid; i < n; i += 256) { int r, g, b; px(i, &r, &g, &b); q[0] += r; q[1] += g; q[2] += b; q[3] += r * r; q[4] += g * g; q[5] += b * b; }
        for (int k = 0; k < 6; ++k) atomicAdd(&s_sum[k], q[k]);
    }
    __syncthreads();
    __shared__ double s_prev[2][3];
    __shared__ double s_tol;
    __shared__ unsigned s_changed;
    if (tid == 0) {
        const int i1 = s_cand[s_pot[1] < s_pot[0] ? 1 : 0];
        int r, g, b; px(i1, &r, &g, &b);
        s_c[0][0] = c0r_; s_c[0][1] = c0g_; s_c[0][2] = c0b_;
        s_c[1][0] = r; s_c[1][1] = g; s_c[1][2] = b;
        double var = 0.0;
        for (int j = 0; j < 3; ++j) { const double m = (double)s_sum[j] / (double)n; var += (double)s_sum[3 + j] / (double)n - m * m; }
        s_tol = var / 3.0 * 1e-4;
        for (int k = 0; k < 2; ++k) for (int j = 0; j < 3; ++j) s_prev[k][j] = -1.0;     // no previous assignment: the first iteration cannot be "unchanged"
    }
    __syncthreads();
    // (B) Lloyd's iterations as sklearn's _kmeans_single_lloyd runs them (float64; label = argmin_k |c_k|^2 - 2 x.c_k, ties to cluster 0): stop when
    //     the assignment repeats, or when the summed squared centre shift is <= tol; at most 300 iterations.  The final labels are the
    //     assignment to the final centres either way.
    auto assign = [&](int r, int g, int b, const double (*C)[3]) -> int {
        const double e0 = (C[0][0] * C[0][0] + C[0][1] * C[0][1] + C[0][2] * C[0][2]) - 2.0 * (r * C[0][0] + g * C[0][1] + b * C[0][2]);
        const double e1 = (C[1][0] * C[1][0] + C[1][1] * C[1][1] + C[1][2] * C[1][2]) - 2.0 * (r * C[1][0] + g * C[1][1] + b * C[1][2]);
        return e1 < e0 ? 1 : 0;
    };
    for (int it = 0; it < 300; ++it) {
        if (tid < 8) s_sum[tid] = 0;
        if (tid == 0) s_changed = 0;
        __syncthreads();
        const bool have_prev = it > 0;
        unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned changed = 0;
        for (int i = tid; i < n; i += 256) {
            int r, g, b; px(i, &r, &g, &b);
            const int l = assign(r, g, b, s_c);
            if (!have_prev || assign(r, g, b, s_prev) != l) changed = 1;
            acc[4 * l] += r; acc[4 * l + 1] += g; acc[4 * l + 2] += b; acc[4 * l + 3] += 1;
        }
        for (int k = 0; k < 8; ++k) if (acc[k]) atomicAdd(&s_sum[k], acc[k]);
        if (changed) atomicOr(&s_changed, 1u);
        __syncthreads();
        if (tid == 0) {
            double shift = 0.0;
            for (int k = 0; k < 2; ++k) {
                const unsigned long long cnt = s_sum[4 * k + 3];
                for (int j = 0; j < 3; ++j) {
                    const double v = cnt ? (double)s_sum[4 * k + j] / (double)cnt : s_c[k][j];
                    shift += (v - s_c[k][j]) * (v - s_c[k][j]);
                    s_prev[k][j] = s_c[k][j];
                    s_c[k][j] = v;
                }
            }
            s_flag = (!s_changed || shift <= s_tol) ? 0 : 1;
        }
        __syncthreads();
        if (!s_flag) break;
    }
    // (E) corner vote -> background cluster; (F) colour-range counts of the other cluster's pixels
    auto label = [&](int i) {
        int r, g, b; px(i, &r, &g, &b);
        return assign(r, g, b, s_c);
    };
    const int corner[4] = {label(0), label(w - 1), label((h - 1) * w), label(n - 1)};
    const int ones = corner[0] + corner[1] + corner[2] + corner[3];
    const int background = ones > 2 ? 1 : 0;               // 2-2 tie: max(set(corners), key=corners.count) visits 0 first (proc.py:477-478; the label numbering is sklearn's: cluster 0 grew from the first seed)
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
