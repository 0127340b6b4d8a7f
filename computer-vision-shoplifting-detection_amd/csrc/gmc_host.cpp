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
