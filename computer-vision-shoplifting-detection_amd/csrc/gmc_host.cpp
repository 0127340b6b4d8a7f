// Host side of BoT-SORT's global motion compensation: pyramidal Lucas-Kanade tracking of sparse corners between two gray
// frames (what cv2.calcOpticalFlowPyrLK does for ultralytics/trackers/utils/gmc.py:GMC.apply_sparseoptflow, reached from
// /root/reference/model.py:38 through model.track).  Plain C++ on the host -- the tracker is sequential per video and stays on
// the host in the reference too -- with the points dealt to a few threads.  The algorithm and its parameters are stated once,
// in numpy, in cvsd_amd/gmc.py:calc_optical_flow_pyr_lk; this file is the same arithmetic (float64) as loops, 50-100x faster.
#include "../../include/mi355_yolo.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#pragma clang fp contract(off)      // float64 expressions are compared bit for bit with their numpy statements

namespace {

inline int reflect101(int i, int n) {          // BORDER_REFLECT_101
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

struct Plane {                                   // float64 image with `pad` reflected pixels on every side
    int h = 0, w = 0, pad = 0, stride = 0;
    std::vector<double> d;
    double at(int y, int x) const { return d[(size_t)(y + pad) * stride + (x + pad)]; }
};

template <class Src>
Plane make_plane(int h, int w, int pad, Src src) {
    Plane p; p.h = h; p.w = w; p.pad = pad; p.stride = w + 2 * pad;
    p.d.resize((size_t)(h + 2 * pad) * p.stride);
    for (int y = -pad; y < h + pad; ++y)
        for (int x = -pad; x < w + pad; ++x)
            p.d[(size_t)(y + pad) * p.stride + (x + pad)] = src(reflect101(y, h), reflect101(x, w));
    return p;
}

// cv2.pyrDown on uint8: separable [1 4 6 4 1] / 16, reflect-101 borders, every second pixel, round to nearest
std::vector<uint8_t> pyr_down(const std::vector<uint8_t>& img, int h, int w, int* oh, int* ow) {
    const int nh = (h + 1) / 2, nw = (w + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    std::vector<int> rows((size_t)h * nw);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < nw; ++x) {
            int s = 0;
            for (int i = 0; i < 5; ++i) s += k[i] * img[(size_t)y * w + reflect101(2 * x + i - 2, w)];
            rows[(size_t)y * nw + x] = s;
        }
    std::vector<uint8_t> out((size_t)nh * nw);
    for (int y = 0; y < nh; ++y)
        for (int x = 0; x < nw; ++x) {
            int s = 0;
            for (int i = 0; i < 5; ++i) s += k[i] * rows[(size_t)reflect101(2 * y + i - 2, h) * nw + x];
            out[(size_t)y * nw + x] = (uint8_t)((s + 128) >> 8);
        }
    *oh = nh; *ow = nw;
    return out;
}

struct Level { int h, w; Plane I, J, Ix, Iy; };

// win x win bilinear patch whose top-left corner is (px - half, py - half)
inline void patch(const Plane& p, double px, double py, int win, double* out) {
    const int half = win / 2;
    const double x = px - half, y = py - half;
    const int ix = (int)std::floor(x), iy = (int)std::floor(y);
    const double ax = x - ix, ay = y - iy;
    const double w00 = (1 - ay) * (1 - ax), w01 = (1 - ay) * ax, w10 = ay * (1 - ax), w11 = ay * ax;
    for (int i = 0; i < win; ++i) {
        const double* r0 = &p.d[(size_t)(iy + i + p.pad) * p.stride + (ix + p.pad)];
        const double* r1 = r0 + p.stride;
        for (int j = 0; j < win; ++j) out[i * win + j] = w00 * r0[j] + w01 * r0[j + 1] + w10 * r1[j] + w11 * r1[j + 1];
    }
}

}  // namespace

extern "C" int mi355_gmc_pyr_lk(const uint8_t* prev, const uint8_t* cur, int height, int width, const float* pts, int n, int win,
                                int max_level, int max_iters, double eps, double min_eig, float* next_pts, uint8_t* status) {
    if (!prev || !cur || height <= 0 || width <= 0 || n < 0 || (n > 0 && (!pts || !next_pts || !status)) || win < 3 || !(win & 1) || win > 63)
        return -1;
    if (n == 0) return 0;
    // pyramids
    std::vector<std::vector<uint8_t>> pp{std::vector<uint8_t>(prev, prev + (size_t)height * width)}, pc{std::vector<uint8_t>(cur, cur + (size_t)height * width)};
    std::vector<int> hs{height}, ws{width};
    for (int l = 0; l < max_level; ++l) {
        const int nh = (hs.back() + 1) / 2, nw = (ws.back() + 1) / 2;
        if (nh <= win || nw <= win) break;                     // buildOpticalFlowPyramid stops at levels not larger than the window
        int oh, ow;
        pp.push_back(pyr_down(pp.back(), hs.back(), ws.back(), &oh, &ow));
        pc.push_back(pyr_down(pc.back(), hs.back(), ws.back(), &oh, &ow));
        hs.push_back(oh); ws.push_back(ow);
    }
    const int top = (int)pp.size() - 1, pad = win + 2, half = win / 2;
    std::vector<Level> lv(pp.size());
    for (size_t l = 0; l < pp.size(); ++l) {
        const int h = hs[l], w = ws[l];
        const std::vector<uint8_t>& a = pp[l];
        const std::vector<uint8_t>& b = pc[l];
        auto A = [&](int y, int x) { return (double)a[(size_t)reflect101(y, h) * w + reflect101(x, w)]; };
        lv[l].h = h; lv[l].w = w;
        lv[l].I = make_plane(h, w, pad, [&](int y, int x) { return (double)a[(size_t)y * w + x]; });
        lv[l].J = make_plane(h, w, pad, [&](int y, int x) { return (double)b[(size_t)y * w + x]; });
        // Scharr gradients of the previous frame (3 / 10 / 3 weights), taken on the reflected image
        lv[l].Ix = make_plane(h, w, pad, [&](int y, int x) {
            return 3 * (A(y - 1, x + 1) - A(y - 1, x - 1)) + 10 * (A(y, x + 1) - A(y, x - 1)) + 3 * (A(y + 1, x + 1) - A(y + 1, x - 1)); });
        lv[l].Iy = make_plane(h, w, pad, [&](int y, int x) {
            return 3 * (A(y + 1, x - 1) - A(y - 1, x - 1)) + 10 * (A(y + 1, x) - A(y - 1, x)) + 3 * (A(y + 1, x + 1) - A(y - 1, x + 1)); });
    }
    const int W2 = win * win;
    const double s = 1.0 / (double)(1 << 20);                  // OpenCV's scaling of the gradient products
    auto work = [&](int i0, int i1) {
        std::vector<double> I(W2), Ix(W2), Iy(W2), Jp(W2);
        for (int i = i0; i < i1; ++i) {
            bool ok = true;
            double nx = 0, ny = 0;
            for (int l = top; l >= 0; --l) {
                const Level& L = lv[l];
                const double px = (double)pts[2 * i] / (double)(1 << l), py = (double)pts[2 * i + 1] / (double)(1 << l);
                if (l == top) { nx = px; ny = py; } else { nx *= 2.0; ny *= 2.0; }
                const double tlx = std::floor(px - half), tly = std::floor(py - half);
                const bool inside = tlx >= -win && tlx < L.w && tly >= -win && tly < L.h;
                if (!inside) { if (l == 0) ok = false; continue; }
                const double cx = std::min(std::max(px, (double)-half), (double)(L.w - 1 + half));
                const double cy = std::min(std::max(py, (double)-half), (double)(L.h - 1 + half));
                patch(L.I, cx, cy, win, I.data()); patch(L.Ix, cx, cy, win, Ix.data()); patch(L.Iy, cx, cy, win, Iy.data());
                double a11 = 0, a12 = 0, a22 = 0;
                for (int k = 0; k < W2; ++k) { a11 += Ix[k] * Ix[k]; a12 += Ix[k] * Iy[k]; a22 += Iy[k] * Iy[k]; }
                a11 *= s; a12 *= s; a22 *= s;
                const double det = a11 * a22 - a12 * a12;
                const double mineig = (a22 + a11 - std::sqrt((a11 - a22) * (a11 - a22) + 4 * a12 * a12)) / (2.0 * W2);
                if (!(mineig >= min_eig) || !(det >= (double)std::numeric_limits<float>::epsilon())) { if (l == 0) ok = false; continue; }
                double pdx = 0, pdy = 0;
                for (int it = 0; it < max_iters; ++it) {
                    const double qx = std::floor(nx - half), qy = std::floor(ny - half);
                    if (!(qx >= -win && qx < L.w && qy >= -win && qy < L.h)) { if (l == 0) ok = false; break; }
                    const double ccx = std::min(std::max(nx, (double)-half), (double)(L.w - 1 + half));
                    const double ccy = std::min(std::max(ny, (double)-half), (double)(L.h - 1 + half));
                    patch(L.J, ccx, ccy, win, Jp.data());
                    double b1 = 0, b2 = 0;
                    for (int k = 0; k < W2; ++k) { const double d = (Jp[k] - I[k]) * 32.0; b1 += d * Ix[k]; b2 += d * Iy[k]; }
                    b1 *= s; b2 *= s;
                    const double dx = (a12 * b2 - a22 * b1) / det, dy = (a12 * b1 - a11 * b2) / det;
                    nx += dx; ny += dy;
                    if (dx * dx + dy * dy <= eps * eps) break;
                    if (it > 0 && std::fabs(dx + pdx) < 0.01 && std::fabs(dy + pdy) < 0.01) { nx -= dx * 0.5; ny -= dy * 0.5; break; }
                    pdx = dx; pdy = dy;
                }
            }
            if (ok && (nx < 0 || ny < 0 || nx >= width || ny >= height)) ok = false;
            next_pts[2 * i] = (float)nx; next_pts[2 * i + 1] = (float)ny;
            status[i] = ok ? 1 : 0;
        }
    };
    const int nthreads = (int)std::max(1u, std::min(std::min(8u, std::thread::hardware_concurrency()), (unsigned)((n + 31) / 32)));
    if (nthreads <= 1) { work(0, n); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, (int)((long long)n * t / nthreads), (int)((long long)n * (t + 1) / nthreads));
    for (auto& t : th) t.join();
    return 0;
}

// ---- frame preparation on the host (the device form is csrc/gmc_kernels.hip: gray_resize_kernel, min_eig_kernel, corner_mask_kernel) ----
// cv2.cvtColor(BGR2GRAY) + cv2.resize(INTER_LINEAR) in their fixed-point arithmetic, cornerMinEigenVal (3x3 Sobel scaled by
// 1 / (4 * block * 255), 3x3 box sums of the products in row-major order, smaller eigenvalue, float64 -> float32) and the mask of the
// corners goodFeaturesToTrack keeps before it orders them (THRESH_TOZERO at quality * max, 3x3 non-maximum suppression, border
// excluded).  Same expressions in the same order as the kernels, so the plane and the corner list are the same on both paths.
// xtab / ytab: per output column / row (source index, tap 0, tap 1), 11-bit taps; unused when oh x ow is the frame size.
extern "C" int mi355_gmc_prepare_host(const uint8_t* bgr, int height, int width, int oh, int ow, const int* xtab, const int* ytab, double quality,
                                      uint8_t* gray_out, float* eig_out, uint8_t* ok_out) {
    if (!bgr || height <= 0 || width <= 0 || oh <= 0 || ow <= 0 || !gray_out || !eig_out || !ok_out) return -1;
    const bool resize = !(oh == height && ow == width);
    if (resize && (!xtab || !ytab)) return -1;
    auto luma = [&](int yy, int xx) {
        const uint8_t* p = bgr + ((size_t)yy * width + xx) * 3;
        return (int)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
    };
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            if (!resize) { gray_out[(size_t)y * ow + x] = (uint8_t)luma(y, x); continue; }
            const int xi = xtab[3 * x], xa0 = xtab[3 * x + 1], xa1 = xtab[3 * x + 2];
            const int yi = ytab[3 * y], yb0 = ytab[3 * y + 1], yb1 = ytab[3 * y + 2];
            const int xj = std::min(xi + 1, width - 1), yj = std::min(yi + 1, height - 1);
            const int h0 = luma(yi, xi) * xa0 + luma(yi, xj) * xa1;
            const int h1 = luma(yj, xi) * xa0 + luma(yj, xj) * xa1;
            int v = (((yb0 * (h0 >> 4)) >> 16) + ((yb1 * (h1 >> 4)) >> 16) + 2) >> 2;
            gray_out[(size_t)y * ow + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    const int h = oh, w = ow;
    const uint8_t* g = gray_out;
    // Sobel products once per pixel, then the 3x3 sums over the reflect-padded product planes
    std::vector<double> pxx((size_t)h * w), pxy((size_t)h * w), pyy((size_t)h * w);
    const double sc = 1.0 / (4.0 * 3.0 * 255.0);
    auto G = [&](int yy, int xx) { return (double)g[(size_t)reflect101(yy, h) * w + reflect101(xx, w)]; };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const double dx = ((G(y - 1, x + 1) - G(y - 1, x - 1)) + 2 * (G(y, x + 1) - G(y, x - 1)) + (G(y + 1, x + 1) - G(y + 1, x - 1))) * sc;
            const double dy = ((G(y + 1, x - 1) - G(y - 1, x - 1)) + 2 * (G(y + 1, x) - G(y - 1, x)) + (G(y + 1, x + 1) - G(y - 1, x + 1))) * sc;
            const size_t k = (size_t)y * w + x;
            pxx[k] = dx * dx; pxy[k] = dx * dy; pyy[k] = dy * dy;
        }
    float mx = 0.f;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double sxx = 0.0, sxy = 0.0, syy = 0.0;
            for (int i = -1; i <= 1; ++i)
                for (int j = -1; j <= 1; ++j) {
                    const size_t k = (size_t)reflect101(y + i, h) * w + reflect101(x + j, w);
                    sxx += pxx[k]; sxy += pxy[k]; syy += pyy[k];
                }
            const double a = sxx * 0.5, b = sxy, c = syy * 0.5;
            const float e = (float)((a + c) - std::sqrt((a - c) * (a - c) + b * b));
            eig_out[(size_t)y * w + x] = e;
            if (e > mx) mx = e;
        }
    const float thr = (float)((double)mx * quality);
    auto T = [&](int yy, int xx) {
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) return -std::numeric_limits<float>::infinity();
        const float v = eig_out[(size_t)yy * w + xx];
        return v > thr ? v : 0.f;
    };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float v = T(y, x);
            float d = -std::numeric_limits<float>::infinity();
            for (int i = -1; i <= 1; ++i)
                for (int j = -1; j <= 1; ++j) d = std::max(d, T(y + i, x + j));
            ok_out[(size_t)y * w + x] = (mx > 0.f && v != 0.f && v == d && y > 0 && y < h - 1 && x > 0 && x < w - 1) ? 1 : 0;
        }
    return 0;
}

// ---- the two small host stages that follow the GPU step (same algorithms as cvsd_amd/gmc.py states them in numpy) ----------------

// goodFeaturesToTrack's last step: the kept corners (ok != 0) strongest first, raster order among equals (numpy: nonzero + stable
// argsort of -eig), at most max_corners; xy_out [max_corners][2] = (x, y) float32.  Returns the number written, or -1.
extern "C" int mi355_gmc_order_corners(const float* eig, const uint8_t* ok, int height, int width, int max_corners, float* xy_out) {
    if (!eig || !ok || height <= 0 || width <= 0 || max_corners < 0 || (max_corners > 0 && !xy_out)) return -1;
    std::vector<int> idx;
    const int n = height * width;
    for (int i = 0; i < n; ++i) if (ok[i]) idx.push_back(i);
    // strongest first, raster order among equals = a strict total order on (strength, index): only the first max_corners places are
    // sorted (a textured frame keeps several thousand corners; the tracker uses a thousand)
    auto before = [&](int a, int b) { return eig[a] > eig[b] || (eig[a] == eig[b] && a < b); };
    const int m = std::min((int)idx.size(), max_corners);
    if (m < (int)idx.size()) std::partial_sort(idx.begin(), idx.begin() + m, idx.end(), before);
    else std::sort(idx.begin(), idx.end(), before);
    for (int k = 0; k < m; ++k) { xy_out[2 * k] = (float)(idx[k] % width); xy_out[2 * k + 1] = (float)(idx[k] / width); }
    return m;
}

namespace {
// least-squares similarity [[a, -b, tx], [b, a, ty]] mapping p -> q over the selected pairs (exact for two pairs)
void similarity_fit(const double* src, const double* dst, const std::vector<int>& sel, double* H) {
    double pmx = 0, pmy = 0, qmx = 0, qmy = 0;
    for (int i : sel) { pmx += src[2 * i]; pmy += src[2 * i + 1]; qmx += dst[2 * i]; qmy += dst[2 * i + 1]; }
    const double inv = 1.0 / (double)sel.size();
    pmx *= inv; pmy *= inv; qmx *= inv; qmy *= inv;
    double den = 0, sa = 0, sb = 0;
    for (int i : sel) {
        const double px = src[2 * i] - pmx, py = src[2 * i + 1] - pmy, qx = dst[2 * i] - qmx, qy = dst[2 * i + 1] - qmy;
        den += px * px + py * py; sa += px * qx + py * qy; sb += px * qy - py * qx;
    }
    if (den <= 0) { H[0] = 1; H[1] = 0; H[2] = qmx - pmx; H[3] = 0; H[4] = 1; H[5] = qmy - pmy; return; }
    const double a = sa / den, b = sb / den;
    H[0] = a; H[1] = -b; H[2] = qmx - (a * pmx - b * pmy);
    H[3] = b; H[4] = a;  H[5] = qmy - (b * pmx + a * pmy);
}
inline bool same_point(const double* a, const double* b) {       // np.allclose on a 2-vector (rtol 1e-5, atol 1e-8)
    return std::fabs(a[0] - b[0]) <= 1e-8 + 1e-5 * std::fabs(b[0]) && std::fabs(a[1] - b[1]) <= 1e-8 + 1e-5 * std::fabs(b[1]);
}
}  // namespace

// cv2.estimateAffinePartial2D(src, dst, RANSAC): 2-point similarity hypotheses, reprojection threshold, adaptive stopping at the asked
// confidence, least-squares refit on the consensus set (gmc.estimate_affine_partial_2d states the same in numpy; the draws come from a
// generator of this routine's own -- splitmix64 seeded with `seed` -- so on data with outliers the two may stop on different, equally
// valid consensus sets).  src / dst: float64 [n][2]; H_out: 6 doubles (row-major 2 x 3); inliers_out: n bytes or NULL.
// Returns 1 when a transform was found, 0 when not (H_out untouched), -1 on bad arguments.
extern "C" int mi355_gmc_affine_partial(const double* src, const double* dst, int n, double threshold, double confidence, int max_iters,
                                        unsigned long long seed, double* H_out, uint8_t* inliers_out) {
    if (n < 0 || (n > 0 && (!src || !dst)) || !H_out || max_iters < 0) return -1;
    if (inliers_out) std::memset(inliers_out, 0, (size_t)n);
    if (n < 2) return 0;
    unsigned long long state = seed + 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        unsigned long long z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    std::vector<uint8_t> best((size_t)n, 0), mask((size_t)n, 0);
    int best_count = 0;
    const double thr2 = threshold * threshold;
    int iters = max_iters;
    std::vector<int> pair(2);
    for (int it = 0; it < iters;) {
        ++it;
        const int i = (int)(next() % (unsigned long long)n);
        int j = (int)(next() % (unsigned long long)(n - 1));
        if (j >= i) ++j;
        if (same_point(src + 2 * i, src + 2 * j) || same_point(dst + 2 * i, dst + 2 * j)) continue;
        pair[0] = i; pair[1] = j;
        double H[6];
        similarity_fit(src, dst, pair, H);
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            const double ex = H[0] * src[2 * k] + H[1] * src[2 * k + 1] + H[2] - dst[2 * k];
            const double ey = H[3] * src[2 * k] + H[4] * src[2 * k + 1] + H[5] - dst[2 * k + 1];
            mask[k] = (ex * ex + ey * ey) <= thr2;
            cnt += mask[k];
        }
        if (cnt > std::max(best_count, 1)) {
            best.swap(mask); best_count = cnt;
            const double w = (double)cnt / (double)n;
            const double denom = std::log(std::max(1.0 - w * w, 1e-12));
            if (denom < 0) iters = std::min(iters, (int)std::ceil(std::log(1.0 - confidence) / denom)); else iters = it;
        }
    }
    if (best_count < 2) return 0;
    std::vector<int> sel;
    for (int k = 0; k < n; ++k) if (best[k]) sel.push_back(k);
    similarity_fit(src, dst, sel, H_out);
    if (inliers_out) std::memcpy(inliers_out, best.data(), (size_t)n);
    return 1;
}
