// Host C++ core of the multi-object tracker behind YOLO.track (/root/reference/model.py:38-46: model.track(frame, persist=True,
// classes=[0]) -> boxes.id / boxes.xywhn).  In ultralytics==8.3.225 that call runs the default botsort.yaml tracker once per frame:
// BoT-SORT = ByteTrack's two-stage association (trackers/byte_tracker.py:BYTETracker.update) on IoU cost fused with the
// detection score (trackers/utils/matching.py), a constant-velocity Kalman filter over (cx, cy, w, h)
// (trackers/utils/kalman_filter.py:KalmanFilterXYWH), the camera-motion warp of every predicted state (STrack.multi_gmc) and
// lap.lapjv(extend_cost=True, cost_limit=thresh) as the assignment solver (matching.py:linear_assignment, use_lap=True;
// lap==0.5.12, requirements.txt:41).  Everything here is sequential per video and tiny (tens of boxes), so it stays on the host,
// as in the reference -- but as compiled code behind the C ABI instead of 0.3 ms of numpy per frame, which made the tracker, not
// the detector, the bottleneck of the reference's frame loop.
//
// The checker is oracle/tracker_oracle.py (numpy, the round-2/3 implementation with a pure-Python statement of the same
// assignment algorithm); tests compare ids exactly and boxes to float32 rounding.  PARITY UNPINNED against a real Ultralytics /
// lap run (neither is installable here); the published algorithms are restated: ByteTrack (Zhang et al., ECCV 2022), BoT-SORT
// (Aharon et al., 2022), Jonker & Volgenant, "A shortest augmenting path algorithm for dense and sparse linear assignment
// problems", Computing 38 (1987) in the three-phase dense form lap implements (column reduction + reduction transfer, two
// rounds of augmenting row reduction, shortest-path augmentation), including its tie rules (first minimum wins in a row scan,
// last column wins in column reduction).
#include "../../include/mi355_yolo.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#pragma clang fp contract(off)      // the fp32 IoU cost is compared with thresholds: no fused multiply-adds the numpy statement does not have

namespace {

// botsort.yaml defaults (ultralytics/cfg/trackers/botsort.yaml)
constexpr double kTrackHigh = 0.25, kTrackLow = 0.1, kNewTrack = 0.25, kMatch = 0.8;
constexpr int kTrackBuffer = 30;
constexpr double kPosStd = 1.0 / 20.0, kVelStd = 1.0 / 160.0;
enum { TRACKED = 1, LOST = 2, RETIRED = 3 };

// ------------------------------------------------------------------------------------------------ linear assignment
// Dense Jonker-Volgenant on an n x n matrix (row-major).  x[i] = column of row i, y[j] = row of column j.
constexpr double kLarge = 1000000.0;

struct Lap {
    int n; const double* c;
    std::vector<int> x, y, free_rows, cols, pred;
    std::vector<double> v, d;
    double at(int i, int j) const { return c[(size_t)i * n + j]; }

    // phase 1: every column takes its cheapest row (scanning rows upwards, a later row wins only when strictly cheaper); columns
    // are then handed out from the LAST to the first, so that of several columns claiming one row the lowest-numbered keeps it;
    // rows that own exactly one column lower that column's price by the slack to their second-best column (reduction transfer)
    int column_reduction() {
        for (int i = 0; i < n; ++i) { x[i] = -1; v[i] = kLarge; y[i] = 0; }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (at(i, j) < v[j]) { v[j] = at(i, j); y[j] = i; }
        std::vector<char> unique(n, 1);
        for (int j = n - 1; j >= 0; --j) {
            const int i = y[j];
            if (x[i] < 0) x[i] = j;
            else { unique[i] = 0; y[j] = -1; }
        }
        int n_free = 0;
        for (int i = 0; i < n; ++i) {
            if (x[i] < 0) free_rows[n_free++] = i;
            else if (unique[i]) {
                const int j = x[i];
                double mn = kLarge;
                for (int j2 = 0; j2 < n; ++j2) {
                    if (j2 == j) continue;
                    const double r = at(i, j2) - v[j2];
                    if (r < mn) mn = r;
                }
                v[j] -= mn;
            }
        }
        return n_free;
    }

    // phase 2: each free row bids for its best column at the second-best reduced cost; a displaced row goes back to the front of
    // the queue when the price really dropped, to the next round otherwise
    int augmenting_row_reduction(int n_free) {
        int current = 0, next_free = 0;
        long rounds = 0;
        while (current < n_free) {
            ++rounds;
            const int fi = free_rows[current++];
            int j1 = 0, j2 = -1;
            double v1 = at(fi, 0) - v[0], v2 = kLarge;
            for (int j = 1; j < n; ++j) {
                const double r = at(fi, j) - v[j];
                if (r < v2) {
                    if (r >= v1) { v2 = r; j2 = j; }
                    else { v2 = v1; v1 = r; j2 = j1; j1 = j; }
                }
            }
            int i0 = y[j1];
            const double v1_new = v[j1] - (v2 - v1);
            const bool lowers = v1_new < v[j1];
            if (rounds < (long)current * n) {
                if (lowers) v[j1] = v1_new;
                else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = y[j2]; }
                if (i0 >= 0) {
                    if (lowers) free_rows[--current] = i0;
                    else free_rows[next_free++] = i0;
                }
            } else if (i0 >= 0) {
                free_rows[next_free++] = i0;
            }
            x[fi] = j1; y[j1] = fi;
        }
        return next_free;
    }

    // phase 3: Dijkstra over reduced costs from one free row to the nearest unassigned column.  cols[0, lo) = columns whose distance
    // is final ("ready"), [lo, hi) = columns at the current minimum still to be scanned, [hi, n) = the rest
    int shortest_path(int start) {
        int lo = 0, hi = 0, n_ready = 0, final_j = -1;
        for (int j = 0; j < n; ++j) { cols[j] = j; pred[j] = start; d[j] = at(start, j) - v[j]; }
        while (final_j < 0) {
            if (lo == hi) {                                  // nothing left to scan: collect the columns at the next minimum
                n_ready = lo;
                hi = lo + 1;
                double mind = d[cols[lo]];
                for (int k = hi; k < n; ++k) {
                    const int j = cols[k];
                    if (d[j] <= mind) {
                        if (d[j] < mind) { hi = lo; mind = d[j]; }
                        cols[k] = cols[hi]; cols[hi++] = j;
                    }
                }
                for (int k = lo; k < hi; ++k)
                    if (y[cols[k]] < 0) final_j = cols[k];
            }
            if (final_j < 0) {
                // scan the columns at the current minimum; when an unassigned column is reached at that same distance the search ends
                // at once and [lo, hi) stays as it was on entry (the price update below reads the minimum from cols[lo])
                int slo = lo, shi = hi;
                while (slo != shi && final_j < 0) {
                    int j = cols[slo++];
                    const int i = y[j];
                    const double mind = d[j];
                    const double h = at(i, j) - v[j] - mind;
                    for (int k = shi; k < n; ++k) {
                        j = cols[k];
                        const double r = at(i, j) - v[j] - h;
                        if (r < d[j]) {
                            d[j] = r; pred[j] = i;
                            if (r == mind) {
                                if (y[j] < 0) { final_j = j; break; }
                                cols[k] = cols[shi]; cols[shi++] = j;
                            }
                        }
                    }
                }
                if (final_j < 0) { lo = slo; hi = shi; }
            }
        }
        const double mind = d[cols[lo]];
        for (int k = 0; k < n_ready; ++k) { const int j = cols[k]; v[j] += d[j] - mind; }
        return final_j;
    }

    void solve(int n_, const double* c_) {
        n = n_; c = c_;
        x.assign(n, -1); y.assign(n, -1); free_rows.assign(n, 0); cols.assign(n, 0); pred.assign(n, 0); v.assign(n, 0.0); d.assign(n, 0.0);
        int n_free = column_reduction();
        for (int round = 0; n_free > 0 && round < 2; ++round) n_free = augmenting_row_reduction(n_free);
        for (int f = 0; f < n_free; ++f) {
            const int fi = free_rows[f];
            int j = shortest_path(fi), i = -1;
            while (i != fi) { i = pred[j]; y[j] = i; std::swap(j, x[i]); }
        }
    }
};

// lap.lapjv(cost, extend_cost=True, cost_limit=limit): the (rows + cols)^2 problem whose off-diagonal blocks cost limit / 2 per
// cell and whose lower-right block is free, so that leaving a row AND a column unmatched costs exactly `limit`; assignments into
// the padding come back as -1.
void lapjv_extended(const double* cost, int nr, int nc, double limit, std::vector<int>* x_out, std::vector<int>* y_out) {
    x_out->assign(nr, -1); y_out->assign(nc, -1);
    if (nr == 0 || nc == 0) return;
    const int n = nr + nc;
    std::vector<double> ext((size_t)n * n, limit / 2.0);
    for (int i = nr; i < n; ++i)
        for (int j = nc; j < n; ++j) ext[(size_t)i * n + j] = 0.0;
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) ext[(size_t)i * n + j] = cost[(size_t)i * nc + j];
    Lap lap;
    lap.solve(n, ext.data());
    for (int i = 0; i < nr; ++i) (*x_out)[i] = lap.x[i] < nc ? lap.x[i] : -1;
    for (int j = 0; j < nc; ++j) (*y_out)[j] = lap.y[j] < nr ? lap.y[j] : -1;
}

// ------------------------------------------------------------------------------------------------ Kalman filter (cx, cy, w, h | velocities)
struct Kalman {
    static void noise(double w, double h, double pos, double vel, double* q8) {
        const double s[8] = {pos * w, pos * h, pos * w, pos * h, vel * w, vel * h, vel * w, vel * h};
        for (int i = 0; i < 8; ++i) q8[i] = s[i] * s[i];
    }
    static void initiate(const double* z, double* mean, double* cov) {
        for (int i = 0; i < 4; ++i) { mean[i] = z[i]; mean[4 + i] = 0.0; }
        double q[8];
        noise(z[2], z[3], 2 * kPosStd, 10 * kVelStd, q);
        std::memset(cov, 0, 64 * sizeof(double));
        for (int i = 0; i < 8; ++i) cov[i * 9] = q[i];
    }
    // x <- x + v;  P <- F P F' + Q with F = [[I, I], [0, I]]: A <- A + B + B' + C, B <- B + C, C <- C (4x4 blocks)
    static void predict(double* mean, double* cov) {
        double q[8];
        noise(mean[2], mean[3], kPosStd, kVelStd, q);
        double out[64];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const double a = cov[i * 8 + j], b = cov[i * 8 + 4 + j], bt = cov[j * 8 + 4 + i], c = cov[(4 + i) * 8 + 4 + j];
                out[i * 8 + j] = ((a + b) + bt) + c;
                out[i * 8 + 4 + j] = b + c;
                out[(4 + i) * 8 + 4 + j] = c;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) out[(4 + i) * 8 + j] = out[j * 8 + 4 + i];
        for (int i = 0; i < 8; ++i) out[i * 9] += q[i];
        std::memcpy(cov, out, sizeof(out));
        for (int i = 0; i < 4; ++i) mean[i] += mean[4 + i];
    }
    // measurement = the first four components, R = diag((w/20)^2, (h/20)^2, ...): S = A + R (SPD), K = [A; B'] S^-1 by Cholesky
    // (scipy.linalg.cho_factor / cho_solve in kalman_filter.py), x += K (z - x[:4]), P -= K S K'
    static void update(double* mean, double* cov, const double* z) {
        double r[8];
        noise(mean[2], mean[3], kPosStd, kVelStd, r);
        double S[16], L[16] = {0};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j] + (i == j ? r[i] : 0.0);
        for (int j = 0; j < 4; ++j) {
            double s = S[j * 4 + j];
            for (int k = 0; k < j; ++k) s -= L[j * 4 + k] * L[j * 4 + k];
            L[j * 4 + j] = std::sqrt(s);
            for (int i = j + 1; i < 4; ++i) {
                double t = S[i * 4 + j];
                for (int k = 0; k < j; ++k) t -= L[i * 4 + k] * L[j * 4 + k];
                L[i * 4 + j] = t / L[j * 4 + j];
            }
        }
        double K[32];                                    // [8][4]: row m of K solves S k = P[0:4, m]  (P symmetric: column m of the first four rows)
        for (int m = 0; m < 8; ++m) {
            double yv[4], kv[4];
            for (int i = 0; i < 4; ++i) {
                double t = cov[i * 8 + m];
                for (int k = 0; k < i; ++k) t -= L[i * 4 + k] * yv[k];
                yv[i] = t / L[i * 4 + i];
            }
            for (int i = 3; i >= 0; --i) {
                double t = yv[i];
                for (int k = i + 1; k < 4; ++k) t -= L[k * 4 + i] * kv[k];
                kv[i] = t / L[i * 4 + i];
            }
            for (int i = 0; i < 4; ++i) K[m * 4 + i] = kv[i];
        }
        double inn[4];
        for (int i = 0; i < 4; ++i) inn[i] = z[i] - mean[i];
        for (int m = 0; m < 8; ++m) {
            double t = 0.0;
            for (int i = 0; i < 4; ++i) t += K[m * 4 + i] * inn[i];
            mean[m] += t;
        }
        double KS[32];
        for (int m = 0; m < 8; ++m)
            for (int j = 0; j < 4; ++j) {
                double t = 0.0;
                for (int i = 0; i < 4; ++i) t += K[m * 4 + i] * S[i * 4 + j];
                KS[m * 4 + j] = t;
            }
        for (int a = 0; a < 8; ++a)
            for (int b = 0; b < 8; ++b) {
                double t = 0.0;
                for (int j = 0; j < 4; ++j) t += KS[a * 4 + j] * K[b * 4 + j];
                cov[a * 8 + b] -= t;
            }
    }
    // STrack.multi_gmc for one track: the 2x2 rotation / scale block acts on every (x, y)-like pair of the state, the translation on
    // the centre only; P <- R8 P R8' with R8 = kron(I4, H[:2, :2])
    static void warp(double* mean, double* cov, const double* H) {
        const double r00 = H[0], r01 = H[1], r10 = H[3], r11 = H[4];
        for (int p = 0; p < 4; ++p) {
            const double a = mean[2 * p], b = mean[2 * p + 1];
            mean[2 * p] = r00 * a + r01 * b;
            mean[2 * p + 1] = r10 * a + r11 * b;
        }
        mean[0] += H[2]; mean[1] += H[5];
        double t[64];
        for (int p = 0; p < 4; ++p)                       // t = R8 P
            for (int j = 0; j < 8; ++j) {
                const double a = cov[(2 * p) * 8 + j], b = cov[(2 * p + 1) * 8 + j];
                t[(2 * p) * 8 + j] = r00 * a + r01 * b;
                t[(2 * p + 1) * 8 + j] = r10 * a + r11 * b;
            }
        for (int i = 0; i < 8; ++i)                       // P = t R8'
            for (int p = 0; p < 4; ++p) {
                const double a = t[i * 8 + 2 * p], b = t[i * 8 + 2 * p + 1];
                cov[i * 8 + 2 * p] = a * r00 + b * r01;
                cov[i * 8 + 2 * p + 1] = a * r10 + b * r11;
            }
    }
};

// ------------------------------------------------------------------------------------------------ tracker
struct Det { double xywh[4]; double score, cls, idx; };
struct Track {
    int id; double mean[8]; double cov[64]; double score, cls, idx; int state; bool confirmed; int born, seen;
};
typedef std::shared_ptr<Track> TP;

inline void xyxy_of(const double* m, float* o) {         // float64 corners, then the float32 cast of the IoU routine
    o[0] = (float)(m[0] - m[2] / 2); o[1] = (float)(m[1] - m[3] / 2); o[2] = (float)(m[0] + m[2] / 2); o[3] = (float)(m[1] + m[3] / 2);
}

// 1 - IoU in float32 with the operation order of utils/metrics.py:bbox_ioa(iou=True) (eps 1e-7)
inline float iou_cost(const float* a, const float* b) {
    const float iw = std::max(std::min(a[2], b[2]) - std::max(a[0], b[0]), 0.0f);
    const float ih = std::max(std::min(a[3], b[3]) - std::max(a[1], b[1]), 0.0f);
    const float inter = iw * ih;
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]), area_b = (b[2] - b[0]) * (b[3] - b[1]);
    return 1.0f - inter / (((area_a + area_b) - inter) + 1e-7f);
}

}  // namespace

struct mi355_tracker {
    int frame_id = 0, max_time_lost = kTrackBuffer, ids_issued = 0;
    std::vector<TP> live, lost;
    std::vector<int> retired_ids;            // ids retired on EARLIER frames (sorted)
    std::vector<double> cost;                // scratch
    std::vector<float> last_rows;            // the rows of the last update (a caller whose buffer was too small fetches them again)

    bool retired(int id) const { return std::binary_search(retired_ids.begin(), retired_ids.end(), id); }

    // cost matrix [tracks][dets] (double): IoU cost in fp32, fused with the detection score when asked (matching.py:fuse_score)
    void costs(const std::vector<TP>& tr, const std::vector<Det>& dets, bool fuse) {
        cost.assign(tr.size() * dets.size(), 0.0);
        std::vector<float> db(dets.size() * 4);
        for (size_t j = 0; j < dets.size(); ++j) {
            const double* z = dets[j].xywh;
            const double m[4] = {z[0], z[1], z[2], z[3]};
            xyxy_of(m, &db[j * 4]);
        }
        for (size_t i = 0; i < tr.size(); ++i) {
            float tb[4];
            xyxy_of(tr[i]->mean, tb);
            for (size_t j = 0; j < dets.size(); ++j) {
                const float c = iou_cost(tb, &db[j * 4]);
                cost[i * dets.size() + j] = fuse ? 1.0 - (double)(1.0f - c) * dets[j].score : (double)c;
            }
        }
    }
    void assign(size_t nr, size_t nc, double limit, std::vector<std::pair<int, int>>* pairs, std::vector<int>* free_r, std::vector<int>* free_c) {
        pairs->clear(); free_r->clear(); free_c->clear();
        std::vector<int> x, y;
        lapjv_extended(cost.data(), (int)nr, (int)nc, limit, &x, &y);
        for (size_t i = 0; i < nr; ++i) { if (x[i] >= 0) pairs->push_back({(int)i, x[i]}); else free_r->push_back((int)i); }
        for (size_t j = 0; j < nc; ++j) if (y[j] < 0) free_c->push_back((int)j);
    }

    int update(const float* det, int n, const double* H, float* out, int cap) {
        ++frame_id;
        const int frame = frame_id;
        std::vector<Det> strong, weak;
        for (int i = 0; i < n; ++i) {
            const float* r = det + (size_t)i * 6;
            // the thresholds are compared in float32, as numpy compares a float32 score array with a Python float
            const double s = (double)r[4];
            const bool hi = r[4] >= (float)kTrackHigh, lo = r[4] > (float)kTrackLow && r[4] < (float)kTrackHigh;
            if (!hi && !lo) continue;
            const double x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
            Det d;
            d.xywh[0] = (double)(float)((x1 + x2) / 2); d.xywh[1] = (double)(float)((y1 + y2) / 2);
            d.xywh[2] = (double)(float)(x2 - x1); d.xywh[3] = (double)(float)(y2 - y1);
            d.score = s; d.cls = (double)r[5]; d.idx = (double)i;
            (hi ? strong : weak).push_back(d);
        }
        std::vector<TP> confirmed, tentative, pool;
        for (const TP& t : live) (t->confirmed ? confirmed : tentative).push_back(t);
        pool = confirmed;
        for (const TP& t : lost) {
            bool dup = false;
            for (const TP& c : confirmed) dup |= c->id == t->id;
            if (!dup) pool.push_back(t);
        }
        for (const TP& t : pool) {                          // a track that is not currently tracked stops changing size
            if (t->state != TRACKED) t->mean[6] = t->mean[7] = 0.0;
            Kalman::predict(t->mean, t->cov);
        }
        if (H) {
            const bool identity = H[0] == 1.0 && H[1] == 0.0 && H[2] == 0.0 && H[3] == 0.0 && H[4] == 1.0 && H[5] == 0.0;
            if (!identity) {
                for (const TP& t : pool) Kalman::warp(t->mean, t->cov, H);
                for (const TP& t : tentative) Kalman::warp(t->mean, t->cov, H);
            }
        }
        std::vector<TP> touched, revived, newly_lost, retired_now;
        auto absorb = [&](const TP& t, const Det& d) {
            Kalman::update(t->mean, t->cov, d.xywh);
            t->score = d.score; t->cls = d.cls; t->idx = d.idx;
            t->state = TRACKED; t->confirmed = true; t->seen = frame;
        };
        auto take = [&](const TP& t, const Det& d) {
            const bool was_tracked = t->state == TRACKED;
            absorb(t, d);
            (was_tracked ? touched : revived).push_back(t);
        };
        std::vector<std::pair<int, int>> pairs;
        std::vector<int> free_t, free_d, free_rest, free_w, free_tent, free_left;
        // 1. strong detections against the pool (IoU cost fused with the detection score)
        costs(pool, strong, true);
        assign(pool.size(), strong.size(), kMatch, &pairs, &free_t, &free_d);
        for (auto& p : pairs) take(pool[p.first], strong[p.second]);
        // 2. weak detections against the still-unmatched TRACKED tracks (plain IoU cost, limit 0.5)
        std::vector<TP> rest;
        for (int i : free_t) if (pool[i]->state == TRACKED) rest.push_back(pool[i]);
        costs(rest, weak, false);
        assign(rest.size(), weak.size(), 0.5, &pairs, &free_rest, &free_w);
        for (auto& p : pairs) take(rest[p.first], weak[p.second]);
        for (int i : free_rest)
            if (rest[i]->state != LOST) { rest[i]->state = LOST; newly_lost.push_back(rest[i]); }
        // 3. leftover strong detections against tracks awaiting confirmation (limit 0.7); unmatched ones are dropped
        std::vector<Det> leftover;
        for (int j : free_d) leftover.push_back(strong[j]);
        costs(tentative, leftover, true);
        assign(tentative.size(), leftover.size(), 0.7, &pairs, &free_tent, &free_left);
        for (auto& p : pairs) { absorb(tentative[p.first], leftover[p.second]); touched.push_back(tentative[p.first]); }
        for (int i : free_tent) { tentative[i]->state = RETIRED; retired_now.push_back(tentative[i]); }
        // 4. births
        for (int j : free_left) {
            const Det& d = leftover[j];
            if (d.score < kNewTrack) continue;
            TP t = std::make_shared<Track>();
            t->id = ++ids_issued;
            Kalman::initiate(d.xywh, t->mean, t->cov);
            t->score = d.score; t->cls = d.cls; t->idx = d.idx; t->state = TRACKED; t->confirmed = frame == 1; t->born = t->seen = frame;
            touched.push_back(t);
        }
        // 5. lost tracks past the buffer
        for (const TP& t : lost)
            if (frame - t->seen > max_time_lost) { t->state = RETIRED; retired_now.push_back(t); }
        // ---- bookkeeping in ByteTrack's order: the lists are rebuilt BEFORE this frame's retirements are recorded, so a track
        // retired now leaves the candidate pool one frame later
        std::vector<TP> nl, nlost;
        auto has = [](const std::vector<TP>& v, int id) { for (const TP& t : v) if (t->id == id) return true; return false; };
        for (const TP& t : live) if (t->state == TRACKED) nl.push_back(t);
        for (const std::vector<TP>* group : {&touched, &revived})
            for (const TP& t : *group) if (!has(nl, t->id)) nl.push_back(t);
        for (const TP& t : lost) if (!has(nl, t->id)) nlost.push_back(t);
        for (const TP& t : newly_lost) nlost.push_back(t);
        nlost.erase(std::remove_if(nlost.begin(), nlost.end(), [&](const TP& t) { return retired(t->id); }), nlost.end());
        {   // a tracked and a lost track on (nearly) the same box (IoU > 0.85): the one with the longer history survives
            std::vector<char> kill_live(nl.size(), 0), kill_lost(nlost.size(), 0);
            for (size_t p = 0; p < nl.size(); ++p) {
                float a[4]; xyxy_of(nl[p]->mean, a);
                for (size_t q = 0; q < nlost.size(); ++q) {
                    float b[4]; xyxy_of(nlost[q]->mean, b);
                    if (iou_cost(a, b) < 0.15f) {
                        if (nl[p]->seen - nl[p]->born > nlost[q]->seen - nlost[q]->born) kill_lost[q] = 1; else kill_live[p] = 1;
                    }
                }
            }
            std::vector<TP> a, b;
            for (size_t p = 0; p < nl.size(); ++p) if (!kill_live[p]) a.push_back(nl[p]);
            for (size_t q = 0; q < nlost.size(); ++q) if (!kill_lost[q]) b.push_back(nlost[q]);
            nl.swap(a); nlost.swap(b);
        }
        for (const TP& t : retired_now) retired_ids.insert(std::upper_bound(retired_ids.begin(), retired_ids.end(), t->id), t->id);
        live.swap(nl); lost.swap(nlost);
        last_rows.clear();
        for (const TP& t : live) {
            if (!t->confirmed) continue;
            const double* s = t->mean;
            const float o[8] = {(float)(s[0] - s[2] / 2), (float)(s[1] - s[3] / 2), (float)(s[0] + s[2] / 2), (float)(s[1] + s[3] / 2),
                                (float)t->id, (float)t->score, (float)t->cls, (float)t->idx};
            last_rows.insert(last_rows.end(), o, o + 8);
        }
        const int m = (int)(last_rows.size() / 8);
        if (cap > 0 && m > 0) std::memcpy(out, last_rows.data(), (size_t)std::min(m, cap) * 8 * sizeof(float));
        return m;
    }
};

extern "C" {

int mi355_lapjv(const double* cost, int n_rows, int n_cols, double cost_limit, int* x_out, int* y_out) {
    if (n_rows < 0 || n_cols < 0 || ((n_rows > 0 && n_cols > 0) && !cost) || (n_rows > 0 && !x_out) || (n_cols > 0 && !y_out) || !(cost_limit < 1e300))
        return -1;
    std::vector<int> x, y;
    lapjv_extended(cost, n_rows, n_cols, cost_limit, &x, &y);
    for (int i = 0; i < n_rows; ++i) x_out[i] = x[i];
    for (int j = 0; j < n_cols; ++j) y_out[j] = y[j];
    return 0;
}

int mi355_kalman_initiate(const double* z, double* mean, double* cov) { if (!z || !mean || !cov) return -1; Kalman::initiate(z, mean, cov); return 0; }
int mi355_kalman_predict(double* mean, double* cov) { if (!mean || !cov) return -1; Kalman::predict(mean, cov); return 0; }
int mi355_kalman_update(double* mean, double* cov, const double* z) { if (!z || !mean || !cov) return -1; Kalman::update(mean, cov, z); return 0; }
int mi355_kalman_warp(double* mean, double* cov, const double* H) { if (!H || !mean || !cov) return -1; Kalman::warp(mean, cov, H); return 0; }

int mi355_tracker_create(int frame_rate, mi355_tracker** out) {
    if (!out || frame_rate <= 0) return -1;
    mi355_tracker* t = new mi355_tracker();
    t->max_time_lost = (int)(frame_rate / 30.0 * kTrackBuffer);
    *out = t;
    return 0;
}
void mi355_tracker_destroy(mi355_tracker* t) { delete t; }

int mi355_tracker_update(mi355_tracker* t, const float* det, int n, const double* warp, float* out_rows, int cap) {
    if (!t || n < 0 || (n > 0 && !det) || cap < 0 || (cap > 0 && !out_rows)) return -1;
    return t->update(det, n, warp, out_rows, cap);
}

int mi355_tracker_last_rows(const mi355_tracker* t, float* out_rows, int cap) {
    if (!t || cap < 0 || (cap > 0 && !out_rows)) return -1;
    const int m = (int)(t->last_rows.size() / 8);
    if (cap > 0 && m > 0) std::memcpy(out_rows, t->last_rows.data(), (size_t)std::min(m, cap) * 8 * sizeof(float));
    return m;
}

int mi355_tracker_state(const mi355_tracker* t, int* frame_id, int* ids_issued, int* n_tracked, int* n_lost) {
    if (!t) return -1;
    if (frame_id) *frame_id = t->frame_id;
    if (ids_issued) *ids_issued = t->ids_issued;
    if (n_tracked) *n_tracked = (int)t->live.size();
    if (n_lost) *n_lost = (int)t->lost.size();
    return 0;
}

int mi355_tracker_tracks(const mi355_tracker* t, int which, double* out, int cap) {
    if (!t || (which != 0 && which != 1) || cap < 0 || (cap > 0 && !out)) return -1;
    const std::vector<TP>& v = which == 0 ? t->live : t->lost;
    int m = 0;
    for (const TP& p : v) {
        if (m < cap) {
            double* o = out + (size_t)m * 16;
            o[0] = p->id; o[1] = p->state; o[2] = p->confirmed ? 1 : 0; o[3] = p->born; o[4] = p->seen; o[5] = p->score; o[6] = p->cls; o[7] = p->idx;
            std::memcpy(o + 8, p->mean, 8 * sizeof(double));
        }
        ++m;
    }
    return m;
}

}  // extern "C"
