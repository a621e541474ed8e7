// K14 — track identities: what the reference gets from boxmot's BotSort behind `self.tracker.update(dets, frame)`
// (eagle/models/coordinate_model.py:66-72, 574-596).  BoT-SORT's motion / IoU association (Aharon et al. 2022, on ByteTrack's two-stage
// scheme): constant-velocity Kalman filter on (cx, cy, w, h), high / low confidence detection sets, three linear assignments with cost
// limits, track life cycle, BoT-SORT's IoU / appearance fusion with the per-track feature EMA (embeddings from K16, reid.hip) and the camera-motion
// warp applied to the track states before association (warps from K17, ecc.hip = boxmot's default estimator, or from the sparse-LK grid);
// the same restatement is oracle/tracker.py, which the tests compare this code with.
//
// Why this stage runs on the HOST side of the library (native C++, not a kernel): it is a strictly sequential recurrence over the
// frames of a clip on <= 300 boxes — per frame a handful of 8x8 Kalman updates and one ~30x30 assignment, tens of microseconds on
// one core, latency-bound.  A single workgroup would need milliseconds per frame for the same dependent chain (the reference runs it
// on the CPU as well).  It consumes the detection records the GPU produced and hands the smoothed boxes back to the GPU for the
// projection (eagle_reproject's kernel).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"

namespace eagle {

namespace {
constexpr double STD_POS = 1.0 / 20, STD_VEL = 1.0 / 160;
enum { T_NEW = 0, T_TRACKED = 1, T_LOST = 2, T_REMOVED = 3 };

struct Track {
    double z[4];                 // the detection that created / last updated it: cx, cy, w, h
    double mean[8], cov[8][8];
    bool has_state = false;
    float conf = 0; int cls = 0, det_ind = -1;
    int state = T_NEW, id = -1, frame_id = 0, start_frame = 0;
    bool is_activated = false;
    // appearance (BoT-SORT with_reid): the detection's normalised embedding and the track's exponential moving average of them (alpha 0.9, re-normalised)
    std::vector<double> curr_feat, smooth_feat;
    void update_features(const std::vector<double>& f)
    {
        if (f.empty()) return;
        curr_feat = f;
        if (smooth_feat.empty()) smooth_feat = f;
        else for (size_t k = 0; k < f.size(); ++k) smooth_feat[k] = 0.9 * smooth_feat[k] + 0.1 * f[k];
        double n2 = 0;
        for (double v : smooth_feat) n2 += v * v;
        const double nn = std::sqrt(n2);
        if (nn > 0) for (double& v : smooth_feat) v /= nn;
    }
    void xyxy(double* b) const
    {
        const double* c = has_state ? mean : z;
        b[0] = c[0] - c[2] / 2; b[1] = c[1] - c[3] / 2; b[2] = c[0] + c[2] / 2; b[3] = c[1] + c[3] / 2;
    }
};
using TP = std::shared_ptr<Track>;

void kf_initiate(Track& t)
{
    for (int i = 0; i < 4; ++i) { t.mean[i] = t.z[i]; t.mean[4 + i] = 0; }
    const double w = t.z[2], h = t.z[3];
    const double sd[8] = {2 * STD_POS * w, 2 * STD_POS * h, 2 * STD_POS * w, 2 * STD_POS * h, 10 * STD_VEL * w, 10 * STD_VEL * h, 10 * STD_VEL * w, 10 * STD_VEL * h};
    memset(t.cov, 0, sizeof(t.cov));
    for (int i = 0; i < 8; ++i) t.cov[i][i] = sd[i] * sd[i];
    t.has_state = true;
}
void kf_predict(Track& t)
{
    const double w = t.mean[2], h = t.mean[3];
    const double sd[8] = {STD_POS * w, STD_POS * h, STD_POS * w, STD_POS * h, STD_VEL * w, STD_VEL * h, STD_VEL * w, STD_VEL * h};
    for (int i = 0; i < 4; ++i) t.mean[i] += t.mean[4 + i];
    // F P F^T with F = [[I, I], [0, I]]
    double A[8][8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) A[i][j] = t.cov[i][j] + (i < 4 ? t.cov[i + 4][j] : 0.0);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) t.cov[i][j] = A[i][j] + (j < 4 ? A[i][j + 4] : 0.0);
    for (int i = 0; i < 8; ++i) t.cov[i][i] += sd[i] * sd[i];
}
void kf_update(Track& t, const double* z)
{
    const double w = t.mean[2], h = t.mean[3];
    const double sd[4] = {STD_POS * w, STD_POS * h, STD_POS * w, STD_POS * h};
    double S[4][4], y[4];
    for (int i = 0; i < 4; ++i) {
        y[i] = z[i] - t.mean[i];
        for (int j = 0; j < 4; ++j) S[i][j] = t.cov[i][j] + (i == j ? sd[i] * sd[i] : 0.0);
    }
    // K = P H^T S^-1 (8x4): solve S X = (P H^T)^T by Gaussian elimination with partial pivoting
    double M[4][12];
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) M[i][j] = S[i][j]; for (int k = 0; k < 8; ++k) M[i][4 + k] = t.cov[k][i]; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (p != c) for (int j = 0; j < 12; ++j) std::swap(M[c][j], M[p][j]);
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = M[r][c] / M[c][c];
            for (int j = c; j < 12; ++j) M[r][j] -= f * M[c][j];
        }
    }
    double K[8][4];
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 4; ++i) K[k][i] = M[i][4 + k] / M[i][i];
    for (int k = 0; k < 8; ++k) { double s = 0; for (int i = 0; i < 4; ++i) s += K[k][i] * y[i]; t.mean[k] += s; }
    double KS[8][4];
    for (int k = 0; k < 8; ++k) for (int j = 0; j < 4; ++j) { double s = 0; for (int i = 0; i < 4; ++i) s += K[k][i] * S[i][j]; KS[k][j] = s; }
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 8; ++b) { double s = 0; for (int j = 0; j < 4; ++j) s += KS[a][j] * K[b][j]; t.cov[a][b] -= s; }
}

double iou_cost(const Track& a, const Track& b)
{
    double p[4], q[4];
    a.xyxy(p); b.xyxy(q);
    const double iw = std::min(p[2], q[2]) - std::max(p[0], q[0]), ih = std::min(p[3], q[3]) - std::max(p[1], q[1]);
    if (!(iw > 0 && ih > 0)) return 1.0;
    const double inter = iw * ih;
    return 1.0 - inter / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter);
}

// Minimum-cost assignment of an n x n matrix (Hungarian algorithm with potentials, O(n^3)); col_of[row].
void hungarian(const std::vector<double>& a, int n, std::vector<int>& col_of)
{
    const double INF = 1e300;
    std::vector<double> u(n + 1, 0), v(n + 1, 0), minv(n + 1);
    std::vector<int> p(n + 1, 0), way(n + 1, 0);
    std::vector<char> used(n + 1);
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        std::fill(minv.begin(), minv.end(), INF);
        std::fill(used.begin(), used.end(), 0);
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            double delta = INF; int j1 = 0;
            for (int j = 1; j <= n; ++j) {
                if (used[j]) continue;
                const double cur = a[(size_t)(i0 - 1) * n + (j - 1)] - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
            for (int j = 0; j <= n; ++j) {
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (p[j0] != 0);
        do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0);
    }
    col_of.assign(n, -1);
    for (int j = 1; j <= n; ++j) if (p[j]) col_of[p[j] - 1] = j - 1;
}

// lap.lapjv(cost, extend_cost=True, cost_limit=thresh): leaving a row and a column unmatched costs thresh (thresh / 2 each)
// fuse: boxmot's fuse_score — cost = 1 - IoU * detection confidence (BoT-SORT applies it to the unconfirmed-track association;
// boxmot 15.0.2 BotSort._handle_unconfirmed_tracks / the original bot_sort.py "if not self.args.mot20: ious_dists = matching.fuse_score(...)")
// scipy cdist(..., "cosine") as boxmot's embedding_distance uses it: 1 - u.v / (|u| |v|), clipped at 0
double cosine_distance(const std::vector<double>& u, const std::vector<double>& v)
{
    double uv = 0, uu = 0, vv = 0;
    for (size_t k = 0; k < u.size(); ++k) { uv += u[k] * v[k]; uu += u[k] * u[k]; vv += v[k] * v[k]; }
    const double d = 1.0 - uv / (std::sqrt(uu) * std::sqrt(vv));
    return d > 0.0 ? d : 0.0;
}

// reid: BoT-SORT's appearance fusion — emb = cosine distance(track smooth feature, detection feature) / 2; emb > appearance_thresh 0.25 -> 1;
// IoU distance (before the score fusion) > proximity_thresh 0.5 -> emb = 1; cost = min(IoU cost, emb).  Pairs without both features keep the IoU cost.
void assign(const std::vector<TP>& rows, const std::vector<TP>& cols, double thresh, std::vector<std::pair<int, int>>& m, std::vector<int>& ur, std::vector<int>& uc, bool fuse = false, bool reid = false)
{
    m.clear(); ur.clear(); uc.clear();
    const int n = (int)rows.size(), k = (int)cols.size();
    if (n == 0 || k == 0) { for (int i = 0; i < n; ++i) ur.push_back(i); for (int j = 0; j < k; ++j) uc.push_back(j); return; }
    const int N = n + k;
    std::vector<double> ext((size_t)N * N, thresh / 2.0);
    for (int i = n; i < N; ++i) for (int j = k; j < N; ++j) ext[(size_t)i * N + j] = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) {
            const double c = iou_cost(*rows[i], *cols[j]);
            double cost = fuse ? 1.0 - (1.0 - c) * (double)cols[j]->conf : c;
            if (reid && !rows[i]->smooth_feat.empty() && !cols[j]->curr_feat.empty()) {
                double e = cosine_distance(rows[i]->smooth_feat, cols[j]->curr_feat) / 2.0;
                if (e > 0.25) e = 1.0;
                if (c > 0.5) e = 1.0;
                cost = std::min(cost, e);
            }
            ext[(size_t)i * N + j] = cost;
        }
    std::vector<int> col_of;
    hungarian(ext, N, col_of);
    std::vector<char> cm(k, 0);
    for (int i = 0; i < n; ++i) {
        if (col_of[i] >= 0 && col_of[i] < k) { m.push_back({i, col_of[i]}); cm[col_of[i]] = 1; }
        else ur.push_back(i);
    }
    for (int j = 0; j < k; ++j) if (!cm[j]) uc.push_back(j);
}

bool contains(const std::vector<TP>& v, const TP& t) { return std::find(v.begin(), v.end(), t) != v.end(); }
}  // namespace

// BoT-SORT's STrack.multi_gmc for one track: R8 = kron(I4, R) on the 8-state (it rotates (w, h) and the velocity pairs too), t added to (cx, cy)
static void apply_warp(Track& t, const double* w)
{
    const double r00 = w[0], r01 = w[1], tx = w[2], r10 = w[3], r11 = w[4], ty = w[5];
    double m[8];
    for (int k = 0; k < 4; ++k) { m[2 * k] = r00 * t.mean[2 * k] + r01 * t.mean[2 * k + 1]; m[2 * k + 1] = r10 * t.mean[2 * k] + r11 * t.mean[2 * k + 1]; }
    m[0] += tx; m[1] += ty;
    for (int i = 0; i < 8; ++i) t.mean[i] = m[i];
    double a[8][8], c[8][8];                             // a = R8 * cov, c = a * R8^T
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 8; ++j) {
            a[2 * k][j] = r00 * t.cov[2 * k][j] + r01 * t.cov[2 * k + 1][j];
            a[2 * k + 1][j] = r10 * t.cov[2 * k][j] + r11 * t.cov[2 * k + 1][j];
        }
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 4; ++k) {
            c[i][2 * k] = a[i][2 * k] * r00 + a[i][2 * k + 1] * r01;
            c[i][2 * k + 1] = a[i][2 * k] * r10 + a[i][2 * k + 1] * r11;
        }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) t.cov[i][j] = c[i][j];
}

// Camera motion from matched points (oracle/tracker.py::similarity_ransac, operation by operation): 200 two-point similarity hypotheses drawn
// by a fixed LCG, most residuals < 3 px wins (first on ties), least squares on the consensus set.  W: [[a, -b, tx], [b, a, ty]] row-major.
void similarity_ransac(const double* p0, const double* p1, int n, double* W)
{
    W[0] = 1; W[1] = 0; W[2] = 0; W[3] = 0; W[4] = 1; W[5] = 0;
    if (n < 2) return;
    const double thr2 = 3.0 * 3.0;
    unsigned long long state = 12345;
    int best_cnt = -1; double ba = 0, bb = 0, btx = 0, bty = 0; bool have = false;
    for (int it = 0; it < 200; ++it) {
        state = (state * 1103515245ull + 12345ull) & 0x7FFFFFFFull; const int i = (int)((state >> 8) % (unsigned long long)n);
        state = (state * 1103515245ull + 12345ull) & 0x7FFFFFFFull; const int j = (int)((state >> 8) % (unsigned long long)n);
        if (i == j) continue;
        const double dx0 = p0[2 * j] - p0[2 * i], dy0 = p0[2 * j + 1] - p0[2 * i + 1];
        const double dx1 = p1[2 * j] - p1[2 * i], dy1 = p1[2 * j + 1] - p1[2 * i + 1];
        const double den = dx0 * dx0 + dy0 * dy0;
        if (den < 1e-9) continue;
        const double a = (dx0 * dx1 + dy0 * dy1) / den, b = (dx0 * dy1 - dy0 * dx1) / den;
        const double tx = p1[2 * i] - (a * p0[2 * i] - b * p0[2 * i + 1]), ty = p1[2 * i + 1] - (b * p0[2 * i] + a * p0[2 * i + 1]);
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            const double ex = a * p0[2 * k] - b * p0[2 * k + 1] + tx - p1[2 * k], ey = b * p0[2 * k] + a * p0[2 * k + 1] + ty - p1[2 * k + 1];
            cnt += (ex * ex + ey * ey) < thr2;
        }
        if (cnt > best_cnt) { best_cnt = cnt; ba = a; bb = b; btx = tx; bty = ty; have = true; }
    }
    if (!have) return;
    std::vector<int> idx;
    for (int k = 0; k < n; ++k) {
        const double ex = ba * p0[2 * k] - bb * p0[2 * k + 1] + btx - p1[2 * k], ey = bb * p0[2 * k] + ba * p0[2 * k + 1] + bty - p1[2 * k + 1];
        if (ex * ex + ey * ey < thr2) idx.push_back(k);
    }
    if (idx.size() < 2) return;
    const double m = (double)idx.size();
    double c0x = 0, c0y = 0, c1x = 0, c1y = 0;
    for (int k : idx) { c0x += p0[2 * k]; c0y += p0[2 * k + 1]; c1x += p1[2 * k]; c1y += p1[2 * k + 1]; }
    c0x /= m; c0y /= m; c1x /= m; c1y /= m;
    double sxx = 0, sdot = 0, scross = 0;
    for (int k : idx) {
        const double qx = p0[2 * k] - c0x, qy = p0[2 * k + 1] - c0y, rx = p1[2 * k] - c1x, ry = p1[2 * k + 1] - c1y;
        sxx += qx * qx + qy * qy; sdot += qx * rx + qy * ry; scross += qx * ry - qy * rx;
    }
    if (sxx < 1e-9) return;
    const double a = sdot / sxx, b = scross / sxx;
    W[0] = a; W[1] = -b; W[2] = c1x - (a * c0x - b * c0y); W[3] = b; W[4] = a; W[5] = c1y - (b * c0x + a * c0y);
}

struct Tracker {
    double hi = 0.5, lo = 0.1, newt = 0.6, match = 0.8;
    int max_time_lost = 30;
    int frame_id = 0, next_id = 1;
    std::vector<TP> tracked, lost, removed;

    void update_track(Track& t, const Track& d)
    {
        kf_update(t, d.z);
        t.update_features(d.curr_feat);
        t.state = T_TRACKED; t.is_activated = true; t.frame_id = frame_id;
        t.conf = d.conf; t.cls = d.cls; t.det_ind = d.det_ind;
    }
    // dets: x1,y1,x2,y2,conf,cls rows -> the activated tracked tracks
    // warp: optional 2 x 3 camera motion of the previous frame -> this one, applied after the prediction like BoT-SORT's multi_gmc
    void update(const EagleDet* dets, int n, std::vector<TP>& out, const double* warp = nullptr, const float* feats = nullptr, const int* feat_det = nullptr, int n_feat = 0)
    {
        ++frame_id;
        const bool reid = feats != nullptr;
        std::vector<TP> first, second;
        for (int i = 0; i < n; ++i) {
            const EagleDet& d = dets[i];
            const double c = d.conf;
            if (!(c > hi) && !(c > lo && c < hi)) continue;
            TP t = std::make_shared<Track>();
            const double x1 = d.x1, y1 = d.y1, x2 = d.x2, y2 = d.y2;
            t->z[0] = (x1 + x2) / 2; t->z[1] = (y1 + y2) / 2; t->z[2] = x2 - x1; t->z[3] = y2 - y1;
            t->conf = d.conf; t->cls = d.cls; t->det_ind = i;
            if (reid && c > hi)                               // boxmot extracts features for the high-confidence set only
                for (int k = 0; k < n_feat; ++k)
                    if (feat_det[k] == i) {
                        std::vector<double> f(EAGLE_REID_DIM);
                        double n2 = 0;
                        for (int q = 0; q < EAGLE_REID_DIM; ++q) { f[q] = feats[(size_t)k * EAGLE_REID_DIM + q]; n2 += f[q] * f[q]; }
                        const double nn = std::sqrt(n2);
                        if (nn > 0) for (double& v : f) v /= nn;      // STrack.update_features: feat /= np.linalg.norm(feat)
                        t->curr_feat = f;
                        break;
                    }
            (c > hi ? first : second).push_back(t);
        }
        std::vector<TP> unconfirmed, act, pool;
        for (auto& t : tracked) (t->is_activated ? act : unconfirmed).push_back(t);
        pool = act;
        for (auto& t : lost) if (!contains(pool, t)) pool.push_back(t);
        for (auto& t : pool) {
            if (t->state != T_TRACKED) { t->mean[6] = 0; t->mean[7] = 0; }
            kf_predict(*t);
        }
        if (warp) { for (auto& t : pool) apply_warp(*t, warp); for (auto& t : unconfirmed) apply_warp(*t, warp); }
        std::vector<TP> activated, refind, lost_now, removed_now;
        std::vector<std::pair<int, int>> m; std::vector<int> ur, uc;
        assign(pool, first, match, m, ur, uc, false, reid);
        for (auto& ij : m) {
            TP& t = pool[ij.first];
            (t->state == T_TRACKED ? activated : refind).push_back(t);
            update_track(*t, *first[ij.second]);
        }
        std::vector<TP> r_tracked;
        for (int i : ur) if (pool[i]->state == T_TRACKED) r_tracked.push_back(pool[i]);
        std::vector<TP> rest;
        for (int j : uc) rest.push_back(first[j]);
        std::vector<int> ur2, uc2;
        assign(r_tracked, second, 0.5, m, ur2, uc2);
        for (auto& ij : m) {
            TP& t = r_tracked[ij.first];
            (t->state == T_TRACKED ? activated : refind).push_back(t);
            update_track(*t, *second[ij.second]);
        }
        for (int i : ur2) { r_tracked[i]->state = T_LOST; lost_now.push_back(r_tracked[i]); }
        std::vector<int> ur3, uc3;
        assign(unconfirmed, rest, 0.7, m, ur3, uc3, true, reid);
        for (auto& ij : m) { update_track(*unconfirmed[ij.first], *rest[ij.second]); activated.push_back(unconfirmed[ij.first]); }
        for (int i : ur3) { unconfirmed[i]->state = T_REMOVED; removed_now.push_back(unconfirmed[i]); }
        for (int j : uc3) {
            TP& t = rest[j];
            if (!((double)t->conf >= newt)) continue;
            kf_initiate(*t);
            t->update_features(t->curr_feat);
            t->id = next_id++;
            t->state = T_TRACKED; t->is_activated = frame_id == 1;
            t->frame_id = t->start_frame = frame_id;
            activated.push_back(t);
        }
        for (auto& t : lost)
            if (frame_id - t->frame_id > max_time_lost) { t->state = T_REMOVED; removed_now.push_back(t); }
        std::vector<TP> nt;
        for (auto& t : tracked) if (t->state == T_TRACKED) nt.push_back(t);
        for (auto& t : activated) if (!contains(nt, t)) nt.push_back(t);
        for (auto& t : refind) if (!contains(nt, t)) nt.push_back(t);
        std::vector<TP> nl;
        for (auto& t : lost) if (!contains(nt, t)) nl.push_back(t);
        for (auto& t : lost_now) nl.push_back(t);
        std::vector<TP> nl2;
        for (auto& t : nl) if (!contains(removed, t)) nl2.push_back(t);       // (published order: this frame's removals leave on the next frame)
        for (auto& t : removed_now) removed.push_back(t);
        if (removed.size() > 4096) removed.erase(removed.begin(), removed.begin() + 2048);   // (old entries can never be in `lost` again)
        // duplicates between tracked and lost: the older track stays
        std::vector<char> da(nt.size(), 0), db(nl2.size(), 0);
        for (size_t i = 0; i < nt.size(); ++i)
            for (size_t j = 0; j < nl2.size(); ++j)
                if (iou_cost(*nt[i], *nl2[j]) < 0.15) {
                    if (nt[i]->frame_id - nt[i]->start_frame > nl2[j]->frame_id - nl2[j]->start_frame) db[j] = 1; else da[i] = 1;
                }
        tracked.clear(); lost.clear();
        for (size_t i = 0; i < nt.size(); ++i) if (!da[i]) tracked.push_back(nt[i]);
        for (size_t j = 0; j < nl2.size(); ++j) if (!db[j]) lost.push_back(nl2[j]);
        out.clear();
        for (auto& t : tracked) if (t->is_activated) out.push_back(t);
    }
};

Tracker* tracker_create(const EagleTrackParams* p)
{
    Tracker* t = new Tracker;
    if (p) {
        t->hi = p->track_high_thresh; t->lo = p->track_low_thresh; t->newt = p->new_track_thresh; t->match = p->match_thresh;
        t->max_time_lost = (int)(p->frame_rate / 30.0 * p->track_buffer);
    }
    return t;
}
void tracker_destroy(Tracker* t) { delete t; }

// cm.py:577-616 on one record: track rows -> Player / Goalkeeper entries keyed by track id (smoothed boxes); when the tracker reports
// no player at all the reference falls back to the raw detections keyed by detection index — which is what the record already holds.
bool tracker_apply(Tracker* T, EagleFrameResult* R, int frame_h, int frame_w, double detector_conf, const double* warp, const float* feats, const int* feat_det, int n_feat)
{
    std::vector<TP> out;
    const int n = std::max(0, std::min(R->n_det, EAGLE_MAX_DET));
    T->update(R->det, n, out, warp, feats, feat_det, n_feat);
    int persons = 0;
    for (auto& t : out) persons += (t->cls == 0 || t->cls == 1) && !((double)t->conf < detector_conf);
    if (persons == 0) return false;
    for (int i = 0; i < n; ++i)
        if (R->det[i].cls == 0 || R->det[i].cls == 1) { R->det[i].reported = 0; R->det[i].id = -1; }
    for (auto& t : out) {
        if (!(t->cls == 0 || t->cls == 1) || (double)t->conf < detector_conf) continue;
        double b[4];
        t->xyxy(b);
        EagleDet& d = R->det[t->det_ind];
        auto clipi = [](double v, int hi) { return (int)std::min(std::max(v, 0.0), (double)hi); };
        d.bx1 = clipi(b[0], frame_w - 1); d.by1 = clipi(b[1], frame_h - 1); d.bx2 = clipi(b[2], frame_w - 1); d.by2 = clipi(b[3], frame_h - 1);
        d.foot_x = (int)((d.bx1 + d.bx2) / 2.0); d.foot_y = d.by2;
        d.id = t->id; d.reported = 1;
        d.conf = t->conf;
    }
    return true;
}

}  // namespace eagle
