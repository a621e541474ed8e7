// Per-frame geometry on the GPU (K9/K10): one workgroup per frame turns the heat-map maxima into the reference's
// keypoint dict, synthesises keypoints by line intersection, solves the image->pitch homography
// (normalised DLT inside RANSAC + Levenberg-Marquardt polish, i.e. cv2.findHomography(src, dst, cv2.RANSAC, 5.0)),
// projects every detection's foot point and the four image corners, and completes the fixed-size record.
//
// Replaces eagle/models/keypoint_hrnet.py:583-594 and eagle/models/coordinate_model.py:500-518 (decode, threshold,
// dedup), :76-186 (synthesis, cv2.fitLine), :333-367 (findHomography), :369-392 (perspectiveTransform, astype(int),
// bounds test), :396-414 + :32-44 (boundaries).  All float64 arithmetic below is written operation-for-operation
// like oracle/eo_prims.c and built with -ffp-contract=off, so H and every integer derived from it can be compared
// bit-for-bit.  RANSAC's sequential semantics are preserved: subsets are drawn serially from the MWC generator,
// the 4-point models of a batch of iterations are solved in parallel (they do not depend on the running best),
// and the batch is then replayed in iteration order with the adaptive iteration bound.
#include <mutex>

#include "common.h"
#include "pitch_table.h"

namespace eagle {

#define POST_T 256
#define RANSAC_WIN (POST_T * 6)
#define RNG_N 65536        // precomputed draws of cv::RNG(-1): the stream does not depend on the data
#define DEPS 2.220446049250313e-16

// Ordering point between the lanes of ONE wave working on shared LDS data: the LDS operations of a wave execute in issue order, so all that is needed is that the
// compiler neither moves LDS accesses across the point nor keeps values in registers over it.
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define JROT(x, y) do { const double g_ = (x), h_ = (y); (x) = g_ - s * (h_ + g_ * tau); (y) = h_ + s * (g_ - h_ * tau); } while (0)
__device__ __forceinline__ float reproj_err1(const double* src, const double* dst, int i, const double* H)
{
    const double X = src[2 * i], Y = src[2 * i + 1];
    const double ww = 1.0 / (H[6] * X + H[7] * Y + 1.0);
    const double dx = (H[0] * X + H[1] * Y + H[2]) * ww - dst[2 * i];
    const double dy = (H[3] * X + H[4] * Y + H[5]) * ww - dst[2 * i + 1];
    return (float)(dx * dx + dy * dy);
}

// ---- minimal-sample model: normalised 8x8 system (h33 = 1), Gaussian elimination with partial pivoting.
// Operation-for-operation oracle/eo_prims.c::eo_h4_homography; all indices static -> the 8x9 system stays in registers
// (row swaps are selects).
__device__ __forceinline__ int h4_homography(const double* src, const double* dst, const int (&idx)[4], double (&H)[9])
{
    const int n = 4;
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    double px[4], py[4], qx[4], qy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int k = idx[i]; px[i] = src[2 * k]; py[i] = src[2 * k + 1]; qx[i] = dst[2 * k]; qy[i] = dst[2 * k + 1]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { cM[0] += px[i]; cM[1] += py[i]; cm[0] += qx[i]; cm[1] += qy[i]; }
    cM[0] /= n; cM[1] /= n; cm[0] /= n; cm[1] /= n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sM[0] += fabs(px[i] - cM[0]); sM[1] += fabs(py[i] - cM[1]);
        sm[0] += fabs(qx[i] - cm[0]); sm[1] += fabs(qy[i] - cm[1]);
    }
    if (fabs(sM[0]) < DEPS || fabs(sM[1]) < DEPS || fabs(sm[0]) < DEPS || fabs(sm[1]) < DEPS) return 0;
    sM[0] = n / sM[0]; sM[1] = n / sM[1]; sm[0] = n / sm[0]; sm[1] = n / sm[1];
    double M[8][9];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double X = (px[i] - cM[0]) * sM[0], Y = (py[i] - cM[1]) * sM[1];
        const double x = (qx[i] - cm[0]) * sm[0], y = (qy[i] - cm[1]) * sm[1];
        M[2 * i][0] = X; M[2 * i][1] = Y; M[2 * i][2] = 1; M[2 * i][3] = 0; M[2 * i][4] = 0; M[2 * i][5] = 0;
        M[2 * i][6] = -x * X; M[2 * i][7] = -x * Y; M[2 * i][8] = x;
        M[2 * i + 1][0] = 0; M[2 * i + 1][1] = 0; M[2 * i + 1][2] = 0; M[2 * i + 1][3] = X; M[2 * i + 1][4] = Y; M[2 * i + 1][5] = 1;
        M[2 * i + 1][6] = -y * X; M[2 * i + 1][7] = -y * Y; M[2 * i + 1][8] = y;
    }
    bool fail_ = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int p = c; double best = fabs(M[c][c]);
#pragma unroll
        for (int r = c + 1; r < 8; ++r) { const double v = fabs(M[r][c]); if (v > best) { best = v; p = r; } }
        fail_ = fail_ || (best < 1e-13);
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const bool sw = (p == r);
#pragma unroll
            for (int j = c; j < 9; ++j) { const double t = M[c][j], u = M[r][j]; M[c][j] = sw ? u : t; M[r][j] = sw ? t : u; }
        }
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const double f = M[r][c] / M[c][c];
#pragma unroll
            for (int j = c + 1; j < 9; ++j) M[r][j] = M[r][j] - f * M[c][j];
        }
    }
    if (fail_) return 0;
    double h[9];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        double s = M[i][8];
#pragma unroll
        for (int j = i + 1; j < 8; ++j) s = s - M[i][j] * h[j];
        h[i] = s / M[i][i];
    }
    h[8] = 1.0;
    const double iT[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double T[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double t[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += iT[3 * r + k] * h[3 * k + c]; t[3 * r + c] = s; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += t[3 * r + k] * T[3 * k + c]; H[3 * r + c] = s; }
    if (fabs(H[8]) < DEPS) return 0;
    const double inv = 1.0 / H[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] *= inv;
    H[8] = 1.0;
    return 1;
}

__device__ int collinear_last(const double* p, const int* idx, int count)
{
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = p[2 * idx[j]] - p[2 * idx[i]], dy1 = p[2 * idx[j] + 1] - p[2 * idx[i] + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = p[2 * idx[k]] - p[2 * idx[i]], dy2 = p[2 * idx[k] + 1] - p[2 * idx[i] + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
        }
    }
    return 0;
}
__device__ int check_subset4(const double* src, const double* dst, const int* idx)
{
    for (int c = 3; c <= 4; ++c)
        if (collinear_last(src, idx, c) || collinear_last(dst, idx, c)) return 0;
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
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
__device__ __forceinline__ unsigned rng_next(unsigned long long* st)
{
    *st = (unsigned long long)(unsigned)(*st) * 4164903690ULL + (unsigned)(*st >> 32);
    return (unsigned)(*st);
}
__device__ int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    if (p < 0) p = 0; if (p > 1) p = 1;
    if (ep < 0) ep = 0; if (ep > 1) ep = 1;
    const double num0 = 1.0 - p;
    const double num = num0 > 2.2250738585072014e-308 ? num0 : 2.2250738585072014e-308;
    const double denom = 1.0 - pow(1.0 - ep, (double)model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    const double ln = log(num), ld = log(denom);
    return (ld >= 0 || -ln >= max_iters * (-ld)) ? max_iters : (int)lrint(ln / ld);
}

// ---- Levenberg-Marquardt polish: workgroup-cooperative, same summation order as oracle/eo_prims.c::eo_lm_refine ------
__device__ __forceinline__ void lm_point(const double* src, const double* dst, int i, const double* h, double* r, double* J)
{
    const double Mx = src[2 * i], My = src[2 * i + 1];
    double ww = h[6] * Mx + h[7] * My + 1.0;
    ww = fabs(ww) > DEPS ? 1.0 / ww : 0.0;
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
// 8 x 8 solve of the LM step by one WAVE on the augmented system M in LDS: lane (r, j) = (lane >> 3, lane & 7) owns entry M[r][j], the lanes of column 7 also the
// right-hand side; the pivot choice and the back substitution are computed redundantly by every lane (uniform control flow).  Per value the operations of
// oracle/eo_prims.c::eo_solve8 in the same order (one thread walked the system until round 6: dynamic row indices on a private array = the 385 scratch instructions
// that were left in find_homography_block, and ~15 us per solve).  Result in x (LDS); every lane returns the same flag.
__device__ int solve8_wave(double (*M)[9], double* x, int lane)
{
    const int r = lane >> 3, j = lane & 7;
    for (int c = 0; c < 8; ++c) {
        int p = c; double bestv = fabs(M[c][c]);
        for (int r2 = c + 1; r2 < 8; ++r2) { const double v = fabs(M[r2][c]); if (v > bestv) { bestv = v; p = r2; } }
        if (bestv < 1e-300) return 0;
        WAVE_SYNC();                                       // every lane has read column c
        if (p != c) {
            if (lane < 9) { const double t = M[c][lane]; M[c][lane] = M[p][lane]; M[p][lane] = t; }
            WAVE_SYNC();
        }
        const double f = M[r][c] / M[c][c];
        const double mj = M[c][j], mr = M[r][j], m8c = M[c][8], m8r = M[r][8];
        WAVE_SYNC();                                       // every lane has read its operands
        if (r > c && j >= c) M[r][j] = mr - f * mj;
        if (r > c && j == 7) M[r][8] = m8r - f * m8c;
        WAVE_SYNC();
    }
    double xr[8];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        double s = M[i][8];
#pragma unroll
        for (int j2 = i + 1; j2 < 8; ++j2) s -= M[i][j2] * xr[j2];
        xr[i] = s / M[i][i];
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = xr[i];
    }
    WAVE_SYNC();
    return 1;
}
__device__ const double D_P10[33] = {1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5,
                                     1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10,
                                     1e11, 1e12, 1e13, 1e14, 1e15, 1e16};

// ---- workgroup-level findHomography(RANSAC) -------------------------------------------------------------------
struct HomoShared {
    double src[2 * EAGLE_MAX_KP], dst[2 * EAGLE_MAX_KP];
    double s2[2 * EAGLE_MAX_KP], d2[2 * EAGLE_MAX_KP];
    union {
        double cand[POST_T][9];                  // phase (c) -> (d): the 4-point models of the round's attempts
        unsigned short jump[3][RANSAC_WIN];      // phase (b): next^4, next^16, next^64 of the attempt chain (dead before (c) writes cand)
    };
    double best[9];
    double jA[9][9], jV[9][9], jd[9], jb[9], jz[9], jH[9];      // the all-inlier fit: 9 x 9 LtL (upper triangle), eigenvectors, Jacobi vectors, result
    int jok;
    double lm_r[2 * EAGLE_MAX_KP], lm_rn[2 * EAGLE_MAX_KP], lm_J[16 * EAGLE_MAX_KP];
    double lm_A[8][8], lm_g[8], lm_h[8], lm_hn[8], lm_M[8][9], lm_d[8];
    double lm_S;
    int lm_lambda, lm_flag;            // flag: 0 = solve failed (retry), 1 = candidate ready, 2 = improved, 3 = stop
    alignas(16) unsigned char code[POST_T];   // per attempt: 255 degenerate subset (rejected), 254 model failed, else inlier count
    unsigned char mask[EAGLE_MAX_KP];
    int niters, max_good, ok, stop, iter, fail_run, ni;
    int pos, newpos;                   // cursor into the precomputed MWC stream
    int wlive[POST_T / 64];            // attempts of the round per wave
    alignas(4) unsigned char draw[RANSAC_WIN + 256];   // the round's slice of the random stream, already reduced modulo the point count
    unsigned tuple[RANSAC_WIN];        // 4 packed point indices of the sampling attempt starting at window position w
    unsigned char len[RANSAC_WIN];     // draws it consumes (0 = stream exhausted)
    int start[POST_T];
};

// ---- normalised DLT ("runKernel") of n points + 9 x 9 symmetric eigen-solve (classic cyclic Jacobi on the upper triangle, eigenvector of the
// smallest eigenvalue), by the whole workgroup.  Same operations in the same order PER VALUE as oracle/eo_prims.c::eo_dlt_homography /
// eo_jacobi9_smallest, so H stays bit-identical: every LtL entry is its own sequential sum over the points (one lane per entry); a Jacobi rotation's
// parameters are computed redundantly by every thread from the same LDS values (uniform control flow), and its element updates — each touches only
// its own pair of entries — are spread over lanes.  Until round 4 one lane ran all of it with the matrices in registers: 306 live doubles, i.e. the
// 941 scratch instructions of this file; now the matrices live in LDS and the kernel has no scratch.
// All threads call; barriers inside; result in S.jH, returns S.jok (uniform).
__device__ int dlt_homography_block(HomoShared& S, const double* src, const double* dst, int n)
{
    const int tid = threadIdx.x;
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};      // redundantly in every thread (n <= 87)
    for (int i = 0; i < n; ++i) { cM[0] += src[2 * i]; cM[1] += src[2 * i + 1]; cm[0] += dst[2 * i]; cm[1] += dst[2 * i + 1]; }
    cM[0] /= n; cM[1] /= n; cm[0] /= n; cm[1] /= n;
    for (int i = 0; i < n; ++i) {
        sM[0] += fabs(src[2 * i] - cM[0]); sM[1] += fabs(src[2 * i + 1] - cM[1]);
        sm[0] += fabs(dst[2 * i] - cm[0]); sm[1] += fabs(dst[2 * i + 1] - cm[1]);
    }
    if (fabs(sM[0]) < DEPS || fabs(sM[1]) < DEPS || fabs(sm[0]) < DEPS || fabs(sm[1]) < DEPS) {              // uniform (every thread computed the same sums)
        __syncthreads();                                   // no thread is still reading src / dst / the caller's shared state when the caller goes on to write it
        return 0;
    }
    sM[0] = n / sM[0]; sM[1] = n / sM[1]; sm[0] = n / sm[0]; sm[1] = n / sm[1];
    __syncthreads();                                       // (the previous user of S.jA / S.jV is done)
    if (tid < 81) {
        const int a = tid / 9, b = tid - a * 9;
        double acc = 0.0;
        if (b >= a) {                                      // entry (a, b) of the upper triangle: its own sum over the points, in point order
            for (int i = 0; i < n; ++i) {
                const double X = (src[2 * i] - cM[0]) * sM[0], Y = (src[2 * i + 1] - cM[1]) * sM[1];
                const double x = (dst[2 * i] - cm[0]) * sm[0], y = (dst[2 * i + 1] - cm[1]) * sm[1];
                const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
                const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
                double lxa = Lx[0], lxb = Lx[0], lya = Ly[0], lyb = Ly[0];
#pragma unroll
                for (int k = 1; k < 9; ++k) { lxa = (a == k) ? Lx[k] : lxa; lxb = (b == k) ? Lx[k] : lxb; lya = (a == k) ? Ly[k] : lya; lyb = (b == k) ? Ly[k] : lyb; }
                acc += lxa * lxb + lya * lyb;
            }
        }
        S.jA[a][b] = acc;
        S.jV[a][b] = (a == b) ? 1.0 : 0.0;
    }
    __syncthreads();
    // The sweeps run in wave 0 alone (round 6): a rotation's work is 19 lanes wide and its parameters are a dependent chain of fp64 divisions and square roots, so the
    // other three waves only added two workgroup barriers per rotation (~250 rotations per fit: 200 us of the kernel); between the lanes of one wave WAVE_SYNC orders.
    if (tid < 64) {
    if (tid < 9) { S.jd[tid] = S.jb[tid] = S.jA[tid][tid]; S.jz[tid] = 0.0; }
    WAVE_SYNC();
    for (int sweep = 1; sweep <= 50; ++sweep) {
        double smm = 0.0;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) smm += fabs(S.jA[p][q]);
        if (smm == 0.0) break;
        const double tresh = sweep < 4 ? 0.2 * smm / 81.0 : 0.0;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = S.jA[p][q], dp = S.jd[p], dq = S.jd[q];
                const double g = 100.0 * fabs(apq);
                if (sweep > 4 && fabs(dp) + g == fabs(dp) && fabs(dq) + g == fabs(dq)) {
                    WAVE_SYNC();                       // every thread has read a[p][q]
                    if (tid == 0) S.jA[p][q] = 0.0;
                    WAVE_SYNC();
                } else if (fabs(apq) > tresh) {
                    double h = dq - dp, t;
                    if (fabs(h) + g == fabs(h)) {
                        t = apq / h;
                    } else {
                        const double theta = 0.5 * h / apq;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);
                    h = t * apq;
                    WAVE_SYNC();                       // every thread has read a[p][q], d[p], d[q]
                    if (tid == 0) {
                        S.jz[p] -= h; S.jz[q] += h; S.jd[p] -= h; S.jd[q] += h; S.jA[p][q] = 0.0;
                    } else if (tid >= 1 && tid <= 9) {     // the rotation on the other entries of rows / columns p, q of the upper triangle
                        const int j = tid - 1;
                        if (j < p) JROT(S.jA[j][p], S.jA[j][q]);
                        else if (j > p && j < q) JROT(S.jA[p][j], S.jA[j][q]);
                        else if (j > q) JROT(S.jA[p][j], S.jA[q][j]);
                    } else if (tid >= 16 && tid < 25) {    // and on the eigenvector columns
                        const int j = tid - 16;
                        JROT(S.jV[j][p], S.jV[j][q]);
                    }
                    WAVE_SYNC();
                }
            }
        WAVE_SYNC();                                   // the sweep's last pairs may have passed without a barrier: every thread has read S.jd[p], S.jd[q] (they steer the branches above)
        if (tid < 9) { S.jb[tid] += S.jz[tid]; S.jd[tid] = S.jb[tid]; S.jz[tid] = 0.0; }
        WAVE_SYNC();
    }
    }
    __syncthreads();
    int m = 0;
    double dm = S.jd[0];
    for (int i = 1; i < 9; ++i) if (S.jd[i] < dm) { dm = S.jd[i]; m = i; }
    double hv[9];
    for (int k = 0; k < 9; ++k) hv[k] = S.jV[k][m];
    const double iT[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double T[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double t3[9], Hh[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double sacc = 0; for (int k = 0; k < 3; ++k) sacc += iT[3 * r + k] * hv[3 * k + c]; t3[3 * r + c] = sacc; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double sacc = 0; for (int k = 0; k < 3; ++k) sacc += t3[3 * r + k] * T[3 * k + c]; Hh[3 * r + c] = sacc; }
    int ok = 1;
    if (fabs(Hh[8]) < DEPS) ok = 0;
    if (ok) {
        const double inv = 1.0 / Hh[8];
        for (int k = 0; k < 9; ++k) Hh[k] *= inv;
        Hh[8] = 1.0;
    }
    __syncthreads();                                       // every thread has read S.jV / S.jd
    if (tid == 0) { for (int k = 0; k < 9; ++k) S.jH[k] = Hh[k]; S.jok = ok; }
    __syncthreads();
    return ok;
}

// refine S.best over the ni inlier points in S.s2/S.d2 (all threads call; wave 0 works — n <= 87 points, an 8 x 8 system: nothing here is wider than a wave, and
// the ~10 workgroup barriers per iteration cost more than the arithmetic — and a barrier closes)
__device__ void lm_refine_block(HomoShared& S, int n, int max_iters)
{
    const int tid = threadIdx.x;
    if (tid < 64) {
        if (tid < 8) S.lm_h[tid] = S.best[tid];
        WAVE_SYNC();
        for (int i = tid; i < n; i += 64) lm_point(S.s2, S.d2, i, S.lm_h, S.lm_r, S.lm_J);
        WAVE_SYNC();
        if (tid == 0) {
            double Ssum = 0;
            for (int i = 0; i < 2 * n; ++i) Ssum += S.lm_r[i] * S.lm_r[i];
            S.lm_S = Ssum; S.lm_lambda = -3;
        }
        WAVE_SYNC();
        for (int it = 0; it < max_iters; ++it) {
            {
                const int a = tid >> 3, b = tid & 7;
                double acc = 0;
                for (int i = 0; i < 2 * n; ++i) acc += S.lm_J[(size_t)i * 8 + a] * S.lm_J[(size_t)i * 8 + b];
                S.lm_A[a][b] = acc;
                if (tid < 8) {
                    double ag = 0;
                    for (int i = 0; i < 2 * n; ++i) ag += S.lm_J[(size_t)i * 8 + tid] * S.lm_r[i];
                    S.lm_g[tid] = ag;
                }
            }
            WAVE_SYNC();
            int improved = 0;
            for (int tries = 0; tries < 16 && !improved; ++tries) {
                {
                    const int a = tid >> 3, b = tid & 7;
                    const double lam = D_P10[S.lm_lambda + 16];
                    double v = S.lm_A[a][b];
                    if (a == b) v += lam * S.lm_A[a][a];
                    S.lm_M[a][b] = v;
                    if (tid < 8) S.lm_M[tid][8] = -S.lm_g[tid];
                }
                WAVE_SYNC();
                const int solved = solve8_wave(S.lm_M, S.lm_d, tid);                 // (uniform)
                if (!solved) {
                    if (tid == 0) S.lm_lambda = S.lm_lambda + 1 > 16 ? 16 : S.lm_lambda + 1;
                    WAVE_SYNC();
                    continue;
                }
                if (tid < 8) S.lm_hn[tid] = S.lm_h[tid] + S.lm_d[tid];
                WAVE_SYNC();
                for (int i = tid; i < n; i += 64) lm_point(S.s2, S.d2, i, S.lm_hn, S.lm_rn, nullptr);
                WAVE_SYNC();
                if (tid == 0) {
                    double Sn = 0;
                    for (int i = 0; i < 2 * n; ++i) Sn += S.lm_rn[i] * S.lm_rn[i];
                    if (Sn < S.lm_S) {
                        for (int k = 0; k < 8; ++k) S.lm_h[k] = S.lm_hn[k];
                        S.lm_S = Sn; S.lm_flag = 2;
                        S.lm_lambda = S.lm_lambda - 1 < -16 ? -16 : S.lm_lambda - 1;
                    } else {
                        S.lm_flag = 1;
                        S.lm_lambda = S.lm_lambda + 1 > 16 ? 16 : S.lm_lambda + 1;
                    }
                }
                WAVE_SYNC();
                improved = S.lm_flag == 2;
                WAVE_SYNC();
            }
            if (!improved) break;
            for (int i = tid; i < n; i += 64) lm_point(S.s2, S.d2, i, S.lm_h, S.lm_r, S.lm_J);
            WAVE_SYNC();
        }
        if (tid < 8) S.best[tid] = S.lm_h[tid];
        if (tid == 8) S.best[8] = 1.0;
    }
    __syncthreads();
}

// img/world: float [n][2].  On return (after a barrier) S.ok, S.best, S.mask are valid for every thread.
__device__ void find_homography_block(HomoShared& S, const unsigned* __restrict__ rng_raw, const float* img, const float* world,
                                      int n, double thresh, int max_iters, int lm_iters)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * n; i += POST_T) { S.src[i] = (double)img[i]; S.dst[i] = (double)world[i]; }
    if (tid == 0) { S.ok = 0; S.niters = max_iters; S.max_good = 0; S.stop = 0; S.iter = 0; S.fail_run = 0; S.pos = 0; }
    __syncthreads();
    if (n < 4) return;
    if (n == 4) {
        const int ok4 = dlt_homography_block(S, S.src, S.dst, 4);
        if (tid == 0) {
            S.ok = ok4;
            for (int k = 0; k < 9; ++k) S.best[k] = S.jH[k];
            for (int i = 0; i < n; ++i) S.mask[i] = 1;
        }
        __syncthreads();
        return;
    }
    const float t2 = (float)(thresh * thresh);
#ifdef EAGLE_DEBUG_RANSAC
    long long tA = 0, tB = 0, tC = 0, tD = 0, t0_ = wall_clock64(); int rounds_ = 0;
#define RT(acc_) { const long long now_ = wall_clock64(); acc_ += now_ - t0_; t0_ = now_; }
#else
#define RT(acc_)
#endif
    constexpr int NDRAW = (RANSAC_WIN + 256) / POST_T;
    unsigned pre[NDRAW];                               // the next round's slice of the stream, requested as soon as the chain walk knows where it starts: its
    bool have_pre = false;                             // ~2 us of memory latency pass under the models and the replay instead of opening the round
    for (;;) {
        if (S.stop || S.iter >= S.niters) break;       // uniform: shared values, read after a barrier
        // (a) every window position: the sampling attempt that would start at that draw (getSubset's inner loops:
        //     draw until 4 distinct indices), its packed indices and the number of draws it consumes
#pragma unroll
        for (int ei = 0; ei < NDRAW; ++ei) {      // one coalesced pass over the stream, one modulo per draw
            const int e = tid + ei * POST_T, j = S.pos + e;
            const unsigned raw = have_pre ? pre[ei] : (j < RNG_N ? rng_raw[j] : 0u);
            S.draw[e] = j < RNG_N ? (unsigned char)(raw % (unsigned)n) : (unsigned char)0;
        }
        __syncthreads();
#pragma unroll
        for (int wi = 0; wi < RANSAC_WIN / POST_T; ++wi) {
            const int w = tid + wi * POST_T;
            // Eight draws of the window from LDS as three aligned words, parsed in registers without a branch (each draw against the up to three indices accepted before it);
            // the draw-by-draw loop (one dependent LDS byte read per draw: 0.5 - 0.65 ms of the 2.3-ms kernel until round 5) only continues the rare attempts
            // that repeat indices five times in eight draws, and serves the end of the stream.
            int j = S.pos + w, cnt = 0;
            const int j0 = j;
            unsigned tuple = 0;
            if (j0 + 8 <= RNG_N) {
                const unsigned* dw = (const unsigned*)(S.draw + (w & ~3));
                const unsigned long long lo = (unsigned long long)dw[0] | ((unsigned long long)dw[1] << 32), hi = dw[2];
                const int sh = 8 * (w & 3);
                const unsigned long long v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                unsigned i0 = 256, i1 = 256, i2 = 256;       // the indices accepted so far (256: none yet)
                int len8 = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const unsigned d = (unsigned)(v >> (8 * b)) & 255u;
                    const bool fresh = d != i0 && d != i1 && d != i2 && cnt < 4;
                    tuple |= fresh ? d << (8 * cnt) : 0u;
                    i2 = (fresh && cnt == 2) ? d : i2; i1 = (fresh && cnt == 1) ? d : i1; i0 = (fresh && cnt == 0) ? d : i0;
                    cnt += fresh;
                    len8 = (cnt == 4 && len8 == 0) ? b + 1 : len8;
                }
                j = cnt == 4 ? j0 + len8 : j0 + 8;
            }
            if (cnt < 4) {
                int idx[4] = {(int)(tuple & 255), (int)((tuple >> 8) & 255), (int)((tuple >> 16) & 255), (int)(tuple >> 24)};
                while (cnt < 4 && j < RNG_N && j - j0 < 250) {
                    const int v = (int)S.draw[j++ - S.pos];
                    bool dup = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dup = dup || (q < cnt && idx[q] == v);
                    if (!dup) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (q == cnt) idx[q] = v;
                        ++cnt;
                    }
                }
                tuple = (unsigned)idx[0] | ((unsigned)idx[1] << 8) | ((unsigned)idx[2] << 16) | ((unsigned)idx[3] << 24);
            }
            S.len[w] = cnt == 4 ? (unsigned char)(j - j0) : 0;
            S.tuple[w] = tuple;
        }
        __syncthreads();
        RT(tA)
        // (b) the attempt chain: attempt k starts where attempt k-1 stopped drawing, s_0 = 0, s_k+1 = next(s_k) with next(w) = w + len[w].  A position without
        //     an attempt (len 0: stream exhausted) or beyond the window is absorbing.  Until round 5 one wave walked the chain with a guess-and-check per 64
        //     attempts (0.56 - 0.63 ms of the kernel at ~5.3 draws per attempt); now pointer doubling: next^4, next^16 and next^64 for every window position,
        //     three passes of four dependent reads, and thread k reaches s_k in at most twelve hops along the base-4 digits of k.
        {
            auto nx0 = [&](int w) -> int { if (w >= RANSAC_WIN) return w; const int l = (int)S.len[w]; return w + l; };      // (len 0: w itself)
            constexpr int NPOS = RANSAC_WIN / POST_T;          // window positions per thread: their hop chains are independent and interleave (the loops are unrolled)
            int xs[NPOS];
#pragma unroll
            for (int i = 0; i < NPOS; ++i) xs[i] = tid + i * POST_T;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NPOS; ++i) xs[i] = nx0(xs[i]);
#pragma unroll
            for (int i = 0; i < NPOS; ++i) S.jump[0][tid + i * POST_T] = (unsigned short)xs[i];
            __syncthreads();
#pragma unroll
            for (int lv = 1; lv < 3; ++lv) {
                const unsigned short* P = S.jump[lv - 1];
#pragma unroll
                for (int i = 0; i < NPOS; ++i) xs[i] = tid + i * POST_T;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < NPOS; ++i) xs[i] = xs[i] < RANSAC_WIN ? (int)P[xs[i]] : xs[i];
#pragma unroll
                for (int i = 0; i < NPOS; ++i) S.jump[lv][tid + i * POST_T] = (unsigned short)xs[i];
                __syncthreads();
            }
            int x = 0;
            for (int r = 0; r < (tid >> 6); ++r) x = x < RANSAC_WIN ? (int)S.jump[2][x] : x;
            for (int r = 0; r < ((tid >> 4) & 3); ++r) x = x < RANSAC_WIN ? (int)S.jump[1][x] : x;
            for (int r = 0; r < ((tid >> 2) & 3); ++r) x = x < RANSAC_WIN ? (int)S.jump[0][x] : x;
            for (int r = 0; r < (tid & 3); ++r) x = nx0(x);
            const int lenx = x < RANSAC_WIN ? (int)S.len[x] : 0;
            const bool live = lenx > 0;                  // attempt tid exists; the chain is monotone: live for tid < natt, and every dead thread sits on the same position
            S.start[tid] = x;
            const int nl = __popcll(__ballot(live));
            if ((tid & 63) == 0) S.wlive[tid >> 6] = nl;
            if (!live) S.newpos = S.pos + x;             // (the same value from every dead thread)
            else if (tid == POST_T - 1) S.newpos = S.pos + x + lenx;
        }
        __syncthreads();
        RT(tB)
        const int natt = S.wlive[0] + S.wlive[1] + S.wlive[2] + S.wlive[3];      // (uniform)
        {
            const int np = S.newpos;
#pragma unroll
            for (int ei = 0; ei < NDRAW; ++ei) { const int j = np + tid + ei * POST_T; pre[ei] = j < RNG_N ? rng_raw[j] : 0u; }
            have_pre = true;
        }
        if (tid < natt) {                           // (c) degeneracy test, 4-point model and its support, in parallel
            const unsigned tp = S.tuple[S.start[tid]];
            const int idx[4] = {(int)(tp & 255), (int)((tp >> 8) & 255), (int)((tp >> 16) & 255), (int)(tp >> 24)};
            unsigned char code = 255;
            if (check_subset4(S.src, S.dst, idx)) {
                code = 254;
                double Hc[9];
                if (h4_homography(S.src, S.dst, idx, Hc)) {
                    int good = 0;
                    for (int i = 0; i < n; ++i) good += (reproj_err1(S.src, S.dst, i, Hc) <= t2);
#pragma unroll
                    for (int k = 0; k < 9; ++k) S.cand[tid][k] = Hc[k];
                    code = (unsigned char)good;
                }
            }
            S.code[tid] = code;
        } else {
            S.code[tid] = 255;
        }
        __syncthreads();
        RT(tC)
        if (tid < 64) {                               // (d) replay the attempts in order = the sequential RANSAC loop
            // One wave, 64 attempts at a time.  Between two EVENTS the loop only counts: iter += the attempts that are not degenerate subsets, fail_run = the run of
            // degenerate subsets at the end.  The events, each found with a ballot: (A) the first attempt whose inlier count beats the running best (new best model,
            // new iteration bound); (B) the attempt with which iter reaches the bound (nothing behind it is looked at); (C) the 1000th consecutive degenerate subset
            // (only a run that starts at the cursor can get there).  The segment up to and including the first event is folded in one step, then the masks are
            // rebuilt behind it.  Until round 5 one lane walked the codes (16 at a time where nothing could happen): 0.22 - 0.29 ms of the kernel.
            int iter = S.iter, fail_run = S.fail_run, niters = S.niters, max_good = S.max_good, stop = 0;
            if (natt == 0) stop = 1;                  // precomputed stream exhausted
            for (int base = 0; base < natt && !stop && iter < niters; base += 64) {
                const int k = base + tid;
                const int code = k < natt ? (int)S.code[k] : 255;
                const int last = natt - 1 - base < 63 ? natt - 1 - base : 63;       // last lane with an attempt
                int cur = 0;
                while (cur <= last && !stop && iter < niters) {
                    const bool here = tid >= cur && tid <= last;
                    const int thr = max_good > 3 ? max_good : 3;
                    const unsigned long long vmask = __ballot(here && code != 255);
                    const unsigned long long imask = __ballot(here && code < 254 && code > thr);
                    const int fv = vmask ? __ffsll((long long)vmask) - 1 : last + 1;                     // first attempt that counts as an iteration
                    const int need = 1000 - fail_run;
                    const int fc = fv - cur >= need ? cur + need - 1 : 64;                               // (C)
                    const int fa = imask ? __ffsll((long long)imask) - 1 : 64;                           // (A)
                    const int R = niters - iter;                                                         // (B): the R-th counting attempt
                    const int pc = __popcll(vmask & ((2ull << tid) - 1ull));
                    const unsigned long long bmask = __ballot(here && code != 255 && pc == R);
                    const int fb = bmask ? __ffsll((long long)bmask) - 1 : 64;
                    int E = fc < fa ? fc : fa; E = fb < E ? fb : E;
                    const bool event = E < 64;
                    if (!event) E = last;
                    const unsigned long long seg = vmask & ((2ull << E) - 1ull);                         // counting attempts of [cur, E]
                    const int nv = __popcll(seg);
                    iter += nv;
                    fail_run = nv ? E - (63 - __clzll((long long)seg)) : fail_run + (E - cur + 1);
                    if (event && E == fc) {                                                              // (in front of every counting attempt: nv == 0)
                        if (iter == 0) max_good = -1;
                        stop = 1;
                    } else if (event && E == fa) {
                        const int kb = base + E, cb = __shfl(code, E);
                        if (tid < 9) S.best[tid] = S.cand[kb][tid];
                        max_good = cb;
                        niters = ransac_update_iters(0.995, (double)(n - cb) / n, 4, niters);
                    }
                    cur = E + 1;
                }
            }
            if (tid == 0) {
                S.niters = niters; S.max_good = max_good; if (stop) S.stop = 1;
                S.iter = iter; S.fail_run = fail_run; S.pos = S.newpos;
            }
        }
        __syncthreads();
        RT(tD)
#ifdef EAGLE_DEBUG_RANSAC
        ++rounds_;
#endif
    }
    __syncthreads();
#ifdef EAGLE_DEBUG_RANSAC
    if (tid == 0) printf("ransac: rounds %d iter %d  parse %.1f us  hop %.1f us  models %.1f us  replay %.1f us (100 MHz ticks)\n", rounds_, S.iter, tA * 0.01, tB * 0.01, tC * 0.01, tD * 0.01);
    long long tE = 0, tF = 0; t0_ = wall_clock64();
#endif
    if (S.max_good <= 0) return;                      // S.ok == 0
    for (int i = tid; i < n; i += POST_T) S.mask[i] = reproj_err1(S.src, S.dst, i, S.best) <= t2;
    __syncthreads();
    if (tid == 0) {
        int ni = 0;
        for (int i = 0; i < n; ++i)
            if (S.mask[i]) { S.s2[2 * ni] = S.src[2 * i]; S.s2[2 * ni + 1] = S.src[2 * i + 1]; S.d2[2 * ni] = S.dst[2 * i]; S.d2[2 * ni + 1] = S.dst[2 * i + 1]; ++ni; }
        S.ni = ni;
        S.ok = 1;
    }
    __syncthreads();
    {
        const int ni = S.ni;                               // (uniform)
        const int okf = dlt_homography_block(S, S.s2, S.d2, ni);
        if (tid == 0) {
            if (okf) { for (int k = 0; k < 9; ++k) S.best[k] = S.jH[k]; }
            else S.ni = 0;
        }
    }
    __syncthreads();
    RT(tE)
    if (S.ni > 0 && lm_iters > 0) lm_refine_block(S, S.ni, lm_iters);
    __syncthreads();
    RT(tF)
#ifdef EAGLE_DEBUG_RANSAC
    if (tid == 0) printf("        inlier fit (mask + DLT + Jacobi) %.1f us  LM polish %.1f us  (%d inliers)\n", tE * 0.01, tF * 0.01, S.ni);
#endif
}

// ---- cv2.fitLine(DIST_L2) closed form and the 2x2 intersection -------------------------------------------------
__device__ bool fit_line(const float* pts, int n, double line[4])
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
    line[0] = (double)(float)cos((double)t); line[1] = (double)(float)sin((double)t);
    line[2] = (double)(float)x; line[3] = (double)(float)y;
    return !(fabs(line[0]) + fabs(line[1]) < 1e-6);
}
__device__ bool intersect_lines(const double* l1, const double* l2, double* px, double* py)
{
    const double vx1 = l1[0], vy1 = l1[1], x01 = l1[2], y01 = l1[3];
    const double vx2 = l2[0], vy2 = l2[1], x02 = l2[2], y02 = l2[3];
    const double det = vx1 * (-vy2) - vy1 * (-vx2);
    if (fabs(det) < 1e-8) return false;
    double a00 = vx1, a01 = -vx2, a10 = vy1, a11 = -vy2, b0 = x02 - x01, b1 = y02 - y01;
    if (fabs(a10) > fabs(a00)) {
        double t;
        t = a00; a00 = a10; a10 = t; t = a01; a01 = a11; a11 = t; t = b0; b0 = b1; b1 = t;
    }
    if (a00 == 0.0) return false;
    const double l = a10 * (1.0 / a00);
    const double u11 = a11 - l * a01;
    if (u11 == 0.0) return false;
    const double t1 = (b1 - l * b0) / u11;
    const double t = (b0 - a01 * t1) / a00;
    *px = x01 + t * vx1; *py = y01 + t * vy1;
    return true;
}

// ---- the per-frame kernel ---------------------------------------------------------------------------------------
struct SynthShared {
    double lines[2][PT_NY][4];                           // [0]: lines of constant world y, [1]: of constant world x
    int lok[2][PT_NY];
    int cx[PT_NY * PT_NX], cy[PT_NY * PT_NX];
    unsigned char cok[PT_NY * PT_NX];
    signed char slot_of[64];
};
struct PostShared {
    HomoShared hs;
    SynthShared syn;
    int hm_idx[64]; float hm_score[64];
    EagleKeypoint kp[EAGLE_MAX_KP];
    int nkp;
    float img[2 * EAGLE_MAX_KP], world[2 * EAGLE_MAX_KP];
    int used[EAGLE_MAX_KP];
    int npts;
    double H[9]; int H_ok;
};

struct PostArgs { const ArgmaxPart* parts; PostParams pp; EagleFrameResult* out; const unsigned* rng_raw; };

__device__ __forceinline__ void persp(const double* H, float fx, float fy, float* ox, float* oy)
{
    const double x = fx, y = fy;
    double w = H[6] * x + H[7] * y + H[8];
    if (fabs(w) > DEPS) {
        w = 1.0 / w;
        *ox = (float)((H[0] * x + H[1] * y + H[2]) * w);
        *oy = (float)((H[3] * x + H[4] * y + H[5]) * w);
    } else {
        *ox = 0.f; *oy = 0.f;
    }
}

__device__ bool find_x_at_y(double x1, double y1, double x2, double y2, double yt, double* out)
{
    if (x2 - x1 == 0.0) return false;
    const double m = (y2 - y1) / (x2 - x1);
    const double c = y1 - m * x1;
    if (m == 0.0) return false;
    *out = (yt - c) / m;
    return true;
}

// (2) threshold / pixel mapping / same-pixel dedup (kh.py:590-593, cm.py:500-518 == cm.py:231-251): heat-map maxima -> kp[0..n)
__device__ int decode_dedup(const int* hm_idx, const float* hm_score, const PostParams& pp, EagleKeypoint* kp)
{
    int lx[57], ly[57], q[57], nq = 0;
    for (int i = 0; i < 57; ++i) {
        const double s = (double)hm_score[i];
        if (!(s > 0.01)) continue;
        if (s < pp.keypoint_conf) continue;
        const int py = hm_idx[i] / pp.hm_w, px = hm_idx[i] - py * pp.hm_w;
        const double xn = (double)px / (double)(pp.hm_w - 1 > 1 ? pp.hm_w - 1 : 1);
        const double yn = (double)py / (double)(pp.hm_h - 1 > 1 ? pp.hm_h - 1 : 1);
        lx[i] = (int)(xn * (double)pp.frame_w); ly[i] = (int)(yn * (double)pp.frame_h);
        q[nq++] = i;
    }
    int nkp = 0;
    for (int a_ = 0; a_ < nq; ++a_) {
        const int i = q[a_];
        int count = 0; float mx = -1.f;
        for (int b_ = 0; b_ < nq; ++b_) {
            const int j = q[b_];
            if (lx[j] == lx[i] && ly[j] == ly[i]) { ++count; mx = hm_score[j] > mx ? hm_score[j] : mx; }
        }
        if (count > 1 && hm_score[i] != mx) continue;
        int slot = -1;
        for (int k = 0; k < nkp; ++k) if (kp[k].x == lx[i] && kp[k].y == ly[i]) slot = k;
        if (slot < 0) slot = nkp++;
        EagleKeypoint e; e.label = i; e.x = lx[i]; e.y = ly[i]; e.score = hm_score[i];
        e.synthesized = 0; e.on_plane = 0; e.inlier = 0; e.pad = 0;
        kp[slot] = e;
    }
    return nkp;
}

// (3) synthesis by line intersection (cm.py:140-186), workgroup-cooperative: the 38 line fits and the 361 candidate
// intersections are independent and run one per thread; thread 0 then walks the candidates in the reference's order (y-lines
// outer, x-lines inner, at most 30 additions) so the result is the serial one.  All threads call; appends to kp[0..*nkp).
__device__ void synthesize_block(SynthShared& Y, EagleKeypoint* kp, int* nkp_io, int tid)
{
    static_assert(PT_NY == PT_NX, "one table shape for both line families");
    __syncthreads();                                     // *nkp_io and kp[] were written by thread 0
    const int nkp0 = *nkp_io;
    if (nkp0 < 2) return;                                // cm.py:326 (uniform: nkp lives in shared memory)
    if (tid < 64) Y.slot_of[tid] = -1;
    __syncthreads();
    if (tid < nkp0) Y.slot_of[kp[tid].label] = (signed char)tid;          // labels are unique within a dict
    __syncthreads();
    if (tid < 2 * PT_NY) {
        const int pass = tid / PT_NY, g = tid - pass * PT_NY;
        float pts[2 * PT_MAXG]; int np = 0;
        for (int m = 0; m < PT_MAXG; ++m) {
            const int lab = pass == 0 ? PT_YGROUP[g][m] : PT_XGROUP[g][m];
            if (lab < 0) break;
            if (PT_NOT_ON_PLANE_IDX[lab]) continue;
            const int sl = Y.slot_of[lab];
            if (sl < 0) continue;
            pts[2 * np] = (float)kp[sl].x; pts[2 * np + 1] = (float)kp[sl].y; ++np;
        }
        double line[4] = {0, 0, 0, 0};
        Y.lok[pass][g] = (np >= 2 && fit_line(pts, np, line)) ? 1 : 0;
        for (int k = 0; k < 4; ++k) Y.lines[pass][g][k] = line[k];
    }
    __syncthreads();
    for (int c = tid; c < PT_NY * PT_NX; c += POST_T) {
        const int gy = c / PT_NX, gx = c - gy * PT_NX;
        unsigned char ok = 0;
        if (Y.lok[0][gy] && Y.lok[1][gx]) {
            const int lab = PT_CROSS[gy][gx];
            double px, py;
            if (lab >= 0 && Y.slot_of[lab] < 0 && intersect_lines(Y.lines[0][gy], Y.lines[1][gx], &px, &py)) {
                ok = 1; Y.cx[c] = (int)rint(px); Y.cy[c] = (int)rint(py);
            }
        }
        Y.cok[c] = ok;
    }
    __syncthreads();
    if (tid < 64) {                                      // one wave walks the candidates in order, 64 at a time through a ballot
        int nkp = nkp0, added = 0;                       // (every lane keeps the same counters; lane 0 writes)
        for (int base = 0; base < PT_NY * PT_NX && added < 30; base += 64) {
            const int c = base + tid;
            unsigned long long m = __ballot(c < PT_NY * PT_NX && Y.cok[c] != 0);
            while (m && added < 30) {
                const int cc = base + __ffsll((long long)m) - 1;
                m &= m - 1;
                const int lab = PT_CROSS[cc / PT_NX][cc % PT_NX];
                if (Y.slot_of[lab] >= 0) continue;       // written by lane 0 below; LDS operations of one wave execute in order
                if (tid == 0) {
                    EagleKeypoint e; e.label = lab; e.x = Y.cx[cc]; e.y = Y.cy[cc]; e.score = 0.f;
                    e.synthesized = 1; e.on_plane = 0; e.inlier = 0; e.pad = 0;
                    Y.slot_of[lab] = (signed char)nkp;
                    kp[nkp] = e;
                }
                ++nkp; ++added;
            }
        }
        if (tid == 0) *nkp_io = nkp;
    }
    __syncthreads();
}

// (4) on-plane selection (cm.py:338-349): float32 image and world points
__device__ int select_plane_points(EagleKeypoint* kp, int nkp, float* img, float* world, int* used)
{
    int np = 0;
    for (int k = 0; k < nkp; ++k) {
        const int lab = kp[k].label;
        if (!PT_ON_PLANE[lab]) continue;
        kp[k].on_plane = 1;
        img[2 * np] = (float)kp[k].x; img[2 * np + 1] = (float)kp[k].y;
        world[2 * np] = (float)PT_WORLD[lab][0]; world[2 * np + 1] = (float)PT_WORLD[lab][1];
        used[np] = k; ++np;
    }
    return np;
}

// (7) boundaries (cm.py:396-414)
__device__ void write_bounds(EagleFrameResult* R, const double* H, bool Hok, int frame_h, int frame_w)
{
    bool bok = false;
    double bx[4] = {0, 0, 0, 0};
    if (Hok) {
        float cx[4], cy[4];
        persp(H, 0.f, 0.f, &cx[0], &cy[0]);
        persp(H, (float)frame_w, 0.f, &cx[1], &cy[1]);
        persp(H, 0.f, (float)frame_h, &cx[2], &cy[2]);
        persp(H, (float)frame_w, (float)frame_h, &cx[3], &cy[3]);
        const double tlx = (int)cx[0], tly = (int)cy[0], trx = (int)cx[1], try_ = (int)cy[1];
        const double blx = (int)cx[2], bly = (int)cy[2], brx = (int)cx[3], bry = (int)cy[3];
        double ntl, ntr, nbl, nbr;
        bok = find_x_at_y(tlx, tly, blx, bly, 68.0, &ntl) && find_x_at_y(trx, try_, brx, bry, 68.0, &ntr) &&
              find_x_at_y(blx, bly, ntl, 68.0, 0.0, &nbl) && find_x_at_y(brx, bry, ntr, 68.0, 0.0, &nbr);
        if (bok) { bx[0] = nbl; bx[1] = ntl; bx[2] = ntr; bx[3] = nbr; }
    }
    R->bounds_valid = bok;
    for (int k = 0; k < 4; ++k) R->bounds[k] = bx[k];
}

// (6) projection of foot points (cm.py:369-392); all threads
__device__ void project_detections(EagleFrameResult* R, const double* H, bool Hok, int tid)
{
    const int nd = R->n_det;
    for (int k = tid; k < nd; k += POST_T) {
        EagleDet* d = &R->det[k];
        float ox = 0.f, oy = 0.f; int tx = 0, ty = 0; unsigned char inb = 0;
        if (Hok) {
            persp(H, (float)d->foot_x, (float)d->foot_y, &ox, &oy);
            tx = (int)ox; ty = (int)oy;
            inb = !(tx < 0 || tx > 105 || ty < 0 || ty > 68);
        }
        d->pitch_xf = ox; d->pitch_yf = oy; d->pitch_x = tx; d->pitch_y = ty; d->in_bounds = inb;
    }
}

__global__ __launch_bounds__(POST_T) void post_kernel(PostArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PostShared& S = *(PostShared*)smem;
    const int f = blockIdx.x, tid = threadIdx.x;
    const PostParams& pp = a.pp;
    EagleFrameResult* R = a.out + f;

    // (1) reduce the per-chunk heat-map maxima: first maximum wins (kh.py:588)
    if (tid < 64) {
        float best = -1.f; int bi = 0;
        for (int c = 0; c < pp.hm_chunks; ++c) {
            const ArgmaxPart p = a.parts[((size_t)f * pp.hm_chunks + c) * 64 + tid];
            if (p.score > best || (p.score == best && p.idx < bi)) { best = p.score; bi = p.idx; }     // partials come from pixel ranges or 2-D tiles
        }
        S.hm_idx[tid] = bi; S.hm_score[tid] = best;
        if (tid < EAGLE_N_LANDMARKS) { R->hm_idx[tid] = bi; R->hm_score[tid] = best; }
    }
    __syncthreads();

    if (tid == 0) S.nkp = decode_dedup(S.hm_idx, S.hm_score, pp, S.kp);
    synthesize_block(S.syn, S.kp, &S.nkp, tid);                       // cm.py:326 (barriers inside)
    if (tid == 0) S.npts = select_plane_points(S.kp, S.nkp, S.img, S.world, S.used);
    __syncthreads();

    // (5) homography
    find_homography_block(S.hs, a.rng_raw, S.img, S.world, S.npts, pp.ransac_thresh, pp.ransac_max_iters, pp.lm_iters);
    __syncthreads();
    const bool Hok = S.npts >= 4 && S.hs.ok;
    if (tid == 0) {
        for (int k = 0; k < 9; ++k) { S.H[k] = Hok ? S.hs.best[k] : 0.0; R->H[k] = S.H[k]; }
        R->H_valid = Hok; R->pad[0] = Hok; R->pad[1] = 0;
        if (Hok) for (int i = 0; i < S.npts; ++i) S.kp[S.used[i]].inlier = S.hs.mask[i];
        R->n_kp = S.nkp;
        write_bounds(R, S.H, Hok, pp.frame_h, pp.frame_w);
    }
    __syncthreads();
    for (int k = tid; k < S.nkp; k += POST_T) R->kp[k] = S.kp[k];
    project_detections(R, S.H, Hok, tid);
}

// ---- the loop body in any cadence ---------------------------------------------------------------------
// cv2.cvtColor(grid, COLOR_BGR2HSV)[..., 0] of one 8-bit pixel (hue range 180, table-driven fixed point, hsv_shift 12)
__device__ __forceinline__ int hue180(int b, int g, int r)
{
    int v = b, vmin = b;
    if (g > v) v = g; if (r > v) v = r;
    if (g < vmin) vmin = g; if (r < vmin) vmin = r;
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int hdiv = diff ? (int)rint((double)(180 << 12) / (6. * (double)diff)) : 0;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    return hh & 255;
}

// np.mean(hsv[y-1:y+2, x-1:x+2, 0]) around a float32 point truncated to int and clipped into the frame (cm.py:452-470)
__device__ double hue_mean_at(const uint8_t* frame, int h, int w, float fx, float fy)
{
    int x = (int)fx, y = (int)fy;
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x); y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    const int x0 = x - 1 > 0 ? x - 1 : 0, x1 = x + 2 < w ? x + 2 : w, y0 = y - 1 > 0 ? y - 1 : 0, y1 = y + 2 < h ? y + 2 : h;
    int sum = 0;
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) {
            const uint8_t* p = frame + ((size_t)yy * w + xx) * 3;
            sum += hue180(p[0], p[1], p[2]);
        }
    return (double)sum / (double)((y1 - y0) * (x1 - x0));
}

// float32 add.reduce in numpy's order (pairwise sum, n <= 128: eight running partial sums, then the tail)
__device__ float np_sum_f32(const float* a, int n)
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

struct FlowShared {
    float nxt[2 * EAGLE_N_LANDMARKS], prv[2 * EAGLE_N_LANDMARKS]; int lab[EAGLE_N_LANDMARKS]; unsigned char status[EAGLE_N_LANDMARKS];
    double hue[2 * EAGLE_N_LANDMARKS];     // [2*p] around the new point, [2*p+1] around the previous point
    float move[EAGLE_N_LANDMARKS], sq[EAGLE_N_LANDMARKS];
    int surv[EAGLE_N_LANDMARKS];
    EagleKeypoint flow[EAGLE_N_LANDMARKS]; int nflow;
};

// calculate_optical_flow after the cv2 call (cm.py:438-478): status filter, z-score filter, hue filter; -> F.flow in dict order.
// Row j of the surviving points is labelled with the j-th key of the UNFILTERED dict (cm.py:446), as the reference does.
__device__ void flow_filter_block(FlowShared& F, const ChainState* st, const uint8_t* frame, int h, int w, int tid)
{
    const int n = st->lk_n;
    if (tid < 2 * n) { F.nxt[tid] = st->lk_next[tid]; F.prv[tid] = st->lk_prev[tid]; }
    if (tid < n) { F.status[tid] = st->lk_status[tid]; F.lab[tid] = st->prev[tid].label; }
    for (int t = tid; t < 2 * n; t += POST_T) {
        const int p = t >> 1;
        const float* q = (t & 1) ? st->lk_prev : st->lk_next;
        F.hue[t] = hue_mean_at(frame, h, w, q[2 * p], q[2 * p + 1]);
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int p = 0; p < n; ++p)
            if (F.status[p]) {
                const float dx = F.nxt[2 * p] - F.prv[2 * p], dy = F.nxt[2 * p + 1] - F.prv[2 * p + 1];
                const float sx = dx * dx, sy = dy * dy;
                F.move[c] = sqrtf(sx + sy);                        // np.linalg.norm(axis=1) on float32
                F.surv[c++] = p;
            }
        int nf = 0;
        if (c > 0) {
            const float mean = np_sum_f32(F.move, c) / (float)c;   // np.mean / np.std on float32 (numpy 2 scalar rules: stays float32)
            for (int j = 0; j < c; ++j) { const float d = F.move[j] - mean; F.sq[j] = d * d; }
            const float sd = sqrtf(np_sum_f32(F.sq, c) / (float)c) + 1e-6f;
            for (int j = 0; j < c; ++j) {
                const int p = F.surv[j];
                const float z = (F.move[j] - mean) / sd;
                if (z > 2.f) continue;
                if (fabs(F.hue[2 * p] - F.hue[2 * p + 1]) > 25.0) continue;
                EagleKeypoint e; e.label = F.lab[j];               // keys[j] of the unfiltered dict
                e.x = (int)F.nxt[2 * p]; e.y = (int)F.nxt[2 * p + 1]; e.score = 0.f;
                e.synthesized = 0; e.on_plane = 0; e.inlier = 0; e.pad = 1;
                F.flow[nf++] = e;
            }
        }
        F.nflow = nf;
    }
    __syncthreads();
}

// {**dst, **src}: values of src win, new keys are appended in src order (one pass over each dict through a label -> slot map)
__device__ int dict_merge(EagleKeypoint* dst, int nd, const EagleKeypoint* src, int ns)
{
    signed char slot_of[EAGLE_N_LANDMARKS];
    for (int k = 0; k < EAGLE_N_LANDMARKS; ++k) slot_of[k] = -1;
    for (int k = 0; k < nd; ++k) slot_of[dst[k].label] = (signed char)k;
    for (int s = 0; s < ns; ++s) {
        int slot = slot_of[src[s].label];
        if (slot < 0) { slot = nd++; slot_of[src[s].label] = (signed char)slot; }
        dst[slot] = src[s];
    }
    return nd;
}

// calibrate_keypoints (cm.py:520-555): move a dark key-point to the brightest pixel (V = max(B,G,R)) of the 6x6 grid around it.
// Returns false where the reference raises IndexError (grid_hsv[3, 3] of a grid clipped to fewer than 4 rows or columns).
__device__ bool calibrate_keypoints(EagleKeypoint* kp, int nkp, const uint8_t* frame, int h, int w)
{
    for (int k = 0; k < nkp; ++k) {
        const int x = kp[k].x, y = kp[k].y;
        if (!(0 <= x && x < w && 0 <= y && y < h)) continue;
        const uint8_t* p = frame + ((size_t)y * w + x) * 3;
        int v = p[0] > p[1] ? p[0] : p[1]; v = v > p[2] ? v : p[2];
        if (v >= 150) continue;
        const int x0 = x - 3 > 0 ? x - 3 : 0, x1 = x + 3 < w ? x + 3 : w, y0 = y - 3 > 0 ? y - 3 : 0, y1 = y + 3 < h ? y + 3 : h;
        if (y1 - y0 < 4 || x1 - x0 < 4) return false;
        int best = -1, bx = 0, by = 0;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
                const uint8_t* q = frame + ((size_t)yy * w + xx) * 3;
                int vv = q[0] > q[1] ? q[0] : q[1]; vv = vv > q[2] ? vv : q[2];
                if (vv > best) { best = vv; bx = xx - x0; by = yy - y0; }
            }
        int ax = x + bx - 3, ay = y + by - 3;
        ax = ax < 0 ? 0 : (ax > w - 1 ? w - 1 : ax); ay = ay < 0 ? 0 : (ay > h - 1 ? h - 1 : ay);
        kp[k].x = ax; kp[k].y = ay; kp[k].pad |= 2;          // value now is a numpy integer in the reference (np.clip)
    }
    return true;
}

struct ChainShared { PostShared P; FlowShared F; EagleKeypoint tmp[EAGLE_N_LANDMARKS]; int attempt, own, stop; };
struct ChainArgs { ClipView cv; ChainState* st; const MemList* mem; EagleFrameResult* recs; PostParams pp; const unsigned* rng_raw; int frame, kint, hint, calib; };

__global__ __launch_bounds__(POST_T) void chain_kernel(ChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ChainShared& C = *(ChainShared*)smem;
    PostShared& S = C.P;
    ChainState* st = a.st;
    const int tid = threadIdx.x, i = a.frame;
    if (st->stalled >= 0 || st->error) return;
    const PostParams& pp = a.pp;
    EagleFrameResult* R = a.recs + i;
    const uint8_t* frame = a.cv.bgr + (size_t)i * a.cv.h * a.cv.w * 3;
    const bool have_flow = st->lk_valid != 0;
#ifdef EAGLE_DEBUG_CHAIN
    long long tq[8]; int nq = 0; tq[nq++] = wall_clock64();
#define CT() tq[nq++] = wall_clock64();
#else
#define CT()
#endif
    // mem[i] -> LDS (one entry per thread)
    const int Mn = a.mem[i].n;
    if (tid < Mn) {
        const EagleFlowKp m = a.mem[i].kp[tid];
        EagleKeypoint e; e.label = m.label; e.x = m.x; e.y = m.y; e.score = m.score;
        e.synthesized = 0; e.on_plane = 0; e.inlier = 0; e.pad = 0;
        C.tmp[tid] = e;
    }
    if (have_flow) flow_filter_block(C.F, st, frame, a.cv.h, a.cv.w, tid);
    else __syncthreads();
    CT()
    if (tid == 0) {
        // key-points of this frame (cm.py:282-324)
        const int nm = Mn > 0 ? Mn : 0;
        const int nf = have_flow ? C.F.nflow : 0;
        const bool scheduled = i == 0 || i % a.kint == 0;
        int nkp = 0, stop = 0;
        if (scheduled) {
            if (Mn < 0) stop = 1;                                    // the caller detects scheduled frames up front
            else {
                for (int k = 0; k < nm; ++k) S.kp[k] = C.tmp[k];
                nkp = nm;
                if (Mn < 4 && i > 0) nkp = dict_merge(S.kp, nkp, C.F.flow, nf);            // {**keypoints, **optical_flow_keypoints}
            }
        } else if (nf < 4) {
            if (Mn < 0) stop = 1;                                    // on-demand detection (cm.py:317)
            else {
                for (int k = 0; k < nm; ++k) S.kp[k] = C.tmp[k];
                nkp = dict_merge(S.kp, nm, C.F.flow, nf);
            }
        } else {
            for (int k = 0; k < nf; ++k) S.kp[k] = C.F.flow[k];
            nkp = nf;
        }
        if (!stop && Mn >= 0) nkp = dict_merge(S.kp, nkp, C.tmp, nm);                      // {**keypoints, **mem.get(i, {})}
        if (stop) st->stalled = i;
        S.nkp = stop ? 0 : nkp;
        C.stop = stop;
    }
    __syncthreads();
    if (C.stop) return;
    CT()
    synthesize_block(S.syn, S.kp, &S.nkp, tid);                      // cm.py:325-326
    CT()
    if (tid == 0) {
        int stop = 0;
        if (a.calib && !calibrate_keypoints(S.kp, S.nkp, frame, a.cv.h, a.cv.w)) { st->error = 1 + i; stop = 2; }
        if (!stop) {
            C.attempt = (i % a.hint == 0) || st->compute_h;
            S.npts = C.attempt ? select_plane_points(S.kp, S.nkp, S.img, S.world, S.used) : 0;
        }
        C.stop = stop;
    }
    __syncthreads();
    if (C.stop) return;
    const int nkp = S.nkp;
    if (tid == 0) st->n_prev = nkp;                                  // prev_keypoints = keypoints (cm.py:329)
    if (tid < nkp) { EagleFlowKp e; e.label = S.kp[tid].label; e.x = S.kp[tid].x; e.y = S.kp[tid].y; e.score = S.kp[tid].score; st->prev[tid] = e; }
    CT()
    if (C.attempt) {
        find_homography_block(S.hs, a.rng_raw, S.img, S.world, S.npts, pp.ransac_thresh, pp.ransac_max_iters, pp.lm_iters);
        __syncthreads();
    }
    CT()
    if (tid == 0) {
        int own = 0;
        if (C.attempt) {
            if (S.npts >= 4 && S.hs.ok) {                            // cm.py:358-365
                own = 1;
                int np_ = 0;
                for (int k = 0; k < S.npts; ++k) {
                    S.kp[S.used[k]].inlier = S.hs.mask[k];
                    if (S.hs.mask[k]) { const EagleKeypoint& q = S.kp[S.used[k]]; EagleFlowKp e; e.label = q.label; e.x = q.x; e.y = q.y; e.score = q.score; st->prev[np_++] = e; }
                }
                st->n_prev = np_;
                for (int k = 0; k < 9; ++k) st->H[k] = S.hs.best[k];
                st->has_H = 1; st->compute_h = 0;
            } else st->compute_h = 1;
        }
        const bool Hok = st->has_H != 0;
        for (int k = 0; k < 9; ++k) { S.H[k] = Hok ? st->H[k] : 0.0; R->H[k] = S.H[k]; }
        S.H_ok = Hok;
        R->H_valid = Hok; R->pad[0] = (uint8_t)own; R->pad[1] = 0;
        R->n_kp = S.nkp;
        write_bounds(R, S.H, Hok, pp.frame_h, pp.frame_w);
    }
    if (tid < EAGLE_N_LANDMARKS) { R->hm_idx[tid] = 0; R->hm_score[tid] = 0.f; }
    __syncthreads();
    for (int k = tid; k < S.nkp; k += POST_T) R->kp[k] = S.kp[k];
    project_detections(R, S.H, S.H_ok != 0, tid);
#ifdef EAGLE_DEBUG_CHAIN
    CT()
    if (tid == 0 && (i == 3 || i == 25)) printf("chain frame %d: flow-filter %.1f  merge %.1f  synth %.1f  calib/select %.1f  H %.1f  record %.1f us\n", i,
        (tq[1]-tq[0])*0.01, (tq[2]-tq[1])*0.01, (tq[3]-tq[2])*0.01, (tq[4]-tq[3])*0.01, (tq[5]-tq[4])*0.01, (tq[6]-tq[5])*0.01);
#endif
}

// operator form of calculate_optical_flow's filter stage
struct FilterArgs { ClipView cv; ChainState* st; int hue_frame; };
__global__ __launch_bounds__(POST_T) void flow_filter_kernel(FilterArgs a)
{
    __shared__ FlowShared F;
    const uint8_t* frame = a.cv.bgr + (size_t)a.hue_frame * a.cv.h * a.cv.w * 3;
    flow_filter_block(F, a.st, frame, a.cv.h, a.cv.w, threadIdx.x);
    if (threadIdx.x == 0) {
        a.st->flow_n = F.nflow;
        for (int k = 0; k < F.nflow; ++k) { EagleFlowKp e; e.label = F.flow[k].label; e.x = F.flow[k].x; e.y = F.flow[k].y; e.score = 0.f; a.st->flow[k] = e; }
    }
}

// heat-map maxima -> mem[] entries (the up-front batch of cm.py:217-276 and on-demand detections)
struct DecodeArgs { const ArgmaxPart* parts; PostParams pp; MemList* mem; int first, stride; };
__global__ __launch_bounds__(64) void decode_mem_kernel(DecodeArgs a)
{
    __shared__ int hm_idx[64]; __shared__ float hm_score[64];
    __shared__ EagleKeypoint kp[EAGLE_N_LANDMARKS];
    const int f = blockIdx.x, tid = threadIdx.x;
    float best = -1.f; int bi = 0;
    for (int c = 0; c < a.pp.hm_chunks; ++c) {
        const ArgmaxPart p = a.parts[((size_t)f * a.pp.hm_chunks + c) * 64 + tid];
        if (p.score > best || (p.score == best && p.idx < bi)) { best = p.score; bi = p.idx; }
    }
    hm_idx[tid] = bi; hm_score[tid] = best;
    __syncthreads();
    if (tid == 0) {
        const int n = decode_dedup(hm_idx, hm_score, a.pp, kp);
        MemList& M = a.mem[a.first + f * a.stride];
        M.n = n;
        for (int k = 0; k < n; ++k) { EagleFlowKp e; e.label = kp[k].label; e.x = kp[k].x; e.y = kp[k].y; e.score = kp[k].score; M.kp[k] = e; }
    }
}

// cv::RNG(-1) multiply-with-carry stream, generated once per process and device (never freed)
static const unsigned* ransac_rng_table()
{
    static const unsigned* tab[64] = {};
    static std::mutex m;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(m);
    if (!tab[dev]) {
        std::vector<unsigned> h(RNG_N);
        unsigned long long st = 0xffffffffffffffffULL;
        for (int i = 0; i < RNG_N; ++i) { st = (unsigned long long)(unsigned)st * 4164903690ULL + (unsigned)(st >> 32); h[i] = (unsigned)st; }
        unsigned* d = nullptr;
        HIP_CHECK(hipMalloc((void**)&d, sizeof(unsigned) * RNG_N));
        HIP_CHECK(hipMemcpy(d, h.data(), sizeof(unsigned) * RNG_N, hipMemcpyHostToDevice));
        tab[dev] = d;
    }
    return tab[dev];
}

void post_launch(const ArgmaxPart* parts, int n, const PostParams& pp, EagleFrameResult* d_out, hipStream_t s)
{
    PostArgs a; a.parts = parts; a.pp = pp; a.out = d_out; a.rng_raw = ransac_rng_table();
    ensure_max_dynamic_lds((const void*)post_kernel, (int)sizeof(PostShared));
    hipLaunchKernelGGL(post_kernel, dim3(n), dim3(POST_T), sizeof(PostShared), s, a);
    HIP_CHECK(hipGetLastError());
}

// Re-projection with a carried homography (reference cadence homography_interval > 1, cm.py:333-415: between scheduled
// frames the last successful H is reused for the foot points and the boundaries).  flag 1: use Hs[i]; flag 2: no H yet.
struct ReprojArgs { EagleFrameResult* recs; const double* Hs; const unsigned char* flags; int frame_h, frame_w; };
__global__ __launch_bounds__(POST_T) void reproject_kernel(ReprojArgs a)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const int flag = a.flags[f];
    if (flag == 0) return;
    EagleFrameResult* R = a.recs + f;
    __shared__ double H[9];
    const bool Hok = flag == 1;
    if (tid < 9) { H[tid] = Hok ? a.Hs[(size_t)f * 9 + tid] : 0.0; R->H[tid] = H[tid]; }
    __syncthreads();
    if (tid == 0) {
        R->H_valid = Hok;
        bool bok = false;
        double bx[4] = {0, 0, 0, 0};
        if (Hok) {
            float cx[4], cy[4];
            persp(H, 0.f, 0.f, &cx[0], &cy[0]);
            persp(H, (float)a.frame_w, 0.f, &cx[1], &cy[1]);
            persp(H, 0.f, (float)a.frame_h, &cx[2], &cy[2]);
            persp(H, (float)a.frame_w, (float)a.frame_h, &cx[3], &cy[3]);
            const double tlx = (int)cx[0], tly = (int)cy[0], trx = (int)cx[1], try_ = (int)cy[1];
            const double blx = (int)cx[2], bly = (int)cy[2], brx = (int)cx[3], bry = (int)cy[3];
            double ntl, ntr, nbl, nbr;
            bok = find_x_at_y(tlx, tly, blx, bly, 68.0, &ntl) && find_x_at_y(trx, try_, brx, bry, 68.0, &ntr) &&
                  find_x_at_y(blx, bly, ntl, 68.0, 0.0, &nbl) && find_x_at_y(brx, bry, ntr, 68.0, 0.0, &nbr);
            if (bok) { bx[0] = nbl; bx[1] = ntl; bx[2] = ntr; bx[3] = nbr; }
        }
        R->bounds_valid = bok;
        for (int k = 0; k < 4; ++k) R->bounds[k] = bx[k];
    }
    const int nd = R->n_det;
    for (int k = tid; k < nd; k += POST_T) {
        EagleDet* d = &R->det[k];
        float ox = 0.f, oy = 0.f; int tx = 0, ty = 0; unsigned char inb = 0;
        if (Hok) {
            persp(H, (float)d->foot_x, (float)d->foot_y, &ox, &oy);
            tx = (int)ox; ty = (int)oy;
            inb = !(tx < 0 || tx > 105 || ty < 0 || ty > 68);
        }
        d->pitch_xf = ox; d->pitch_yf = oy; d->pitch_x = tx; d->pitch_y = ty; d->in_bounds = inb;
    }
}
void reproject_launch(EagleFrameResult* d_recs, const double* d_Hs, const unsigned char* d_flags, int n, int frame_h, int frame_w, hipStream_t s)
{
    ReprojArgs a{d_recs, d_Hs, d_flags, frame_h, frame_w};
    hipLaunchKernelGGL(reproject_kernel, dim3(n), dim3(POST_T), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

// operator-level entry for the parity tests: findHomography only
struct HomoArgs { const float* img; const float* world; int n; double thresh; int max_iters, lm_iters; double* H; uint8_t* mask; int* ok; const unsigned* rng_raw; };
__global__ __launch_bounds__(POST_T) void homography_kernel(HomoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    HomoShared& S = *(HomoShared*)smem;
    find_homography_block(S, a.rng_raw, a.img, a.world, a.n, a.thresh, a.max_iters, a.lm_iters);
    __syncthreads();
    const bool ok = a.n >= 4 && S.ok;
    if (threadIdx.x == 0) { *a.ok = ok; for (int k = 0; k < 9; ++k) a.H[k] = ok ? S.best[k] : 0.0; }
    for (int i = threadIdx.x; i < a.n; i += POST_T) a.mask[i] = ok ? S.mask[i] : 0;
}
void homography_only_launch(const float* d_img, const float* d_world, int npts, double thresh, int max_iters, int lm_iters,
                            double* d_H, uint8_t* d_mask, int* d_ok, hipStream_t s)
{
    HomoArgs a{d_img, d_world, npts, thresh, max_iters, lm_iters, d_H, d_mask, d_ok, ransac_rng_table()};
    ensure_max_dynamic_lds((const void*)homography_kernel, (int)sizeof(HomoShared));
    hipLaunchKernelGGL(homography_kernel, dim3(1), dim3(POST_T), sizeof(HomoShared), s, a);
    HIP_CHECK(hipGetLastError());
}

void decode_mem_launch(const ArgmaxPart* parts, int n, const PostParams& pp, MemList* mem, int first, int stride, hipStream_t s)
{
    DecodeArgs a{parts, pp, mem, first, stride};
    hipLaunchKernelGGL(decode_mem_kernel, dim3(n), dim3(64), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

void chain_launch(const ClipView& cv, ChainState* st, const MemList* mem, EagleFrameResult* recs, const PostParams& pp, int frame,
                  int kint, int hint, int calib, hipStream_t s)
{
    ChainArgs a{cv, st, mem, recs, pp, ransac_rng_table(), frame, kint, hint, calib};
    ensure_max_dynamic_lds((const void*)chain_kernel, (int)sizeof(ChainShared));
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(POST_T), sizeof(ChainShared), s, a);
    HIP_CHECK(hipGetLastError());
}

void flow_filter_launch(const ClipView& cv, ChainState* st, int hue_frame, hipStream_t s)
{
    FilterArgs a{cv, st, hue_frame};
    hipLaunchKernelGGL(flow_filter_kernel, dim3(1), dim3(POST_T), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace eagle
