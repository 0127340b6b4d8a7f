// BoT-SORT global motion compensation on the GPU: pyramidal Lucas-Kanade tracking of sparse corners between two gray frames
// (what cv2.calcOpticalFlowPyrLK does for ultralytics/trackers/utils/gmc.py:GMC.apply_sparseoptflow, reached from
// /root/reference/model.py:38 through model.track).  The algorithm and its parameters are stated once, in numpy, in
// cvsd_amd/gmc.py:calc_optical_flow_pyr_lk_numpy; csrc/gmc_host.cpp is the same arithmetic as host loops.  On the host the
// 1000-corner budget of goodFeaturesToTrack costs 27 us per point -- 10-27 ms per frame, thirty times the whole detector pass
// (0.42 ms at batch 1), so the reference's frame loop (model.track -> CSV) ran at the tracker's pace.  Here every point is one
// wavefront: its 21 x 21 window is 441 pixels = 7 per lane, the window's I / Ix / Iy samples stay in registers for all
// iterations of a level, the 2 x 2 normal equations are wave reductions, all in float64 (the same formulae as the host code;
// window sums are associated differently -- lane-wise then butterfly -- so results agree to rounding, not bit for bit).
// The pyramids (5-tap Gaussian pyrDown, integer arithmetic: exact) are built by a kernel per level.
#include "../../include/mi355_yolo.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int kMaxLevels = 8;          // pyramid levels held (level 0 = the frame)
constexpr int kMaxPer = 7;             // window pixels per lane: win <= 21 (441 = 6.9 * 64; OpenCV's and Ultralytics' default window)

__device__ __forceinline__ int reflect101(int i, int n) {          // BORDER_REFLECT_101, any distance (as the host code)
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// cv2.pyrDown on uint8: separable [1 4 6 4 1] / 16 twice, reflect-101 borders, every second pixel, (s + 128) >> 8
// blockIdx.y = frame of a batch (mi355_gmc_track_batch): planes of consecutive frames are `fstride` bytes apart (0 for one frame)
__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t* src, int h, int w, uint8_t* dst, int nh, int nw, size_t fstride = 0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nh * nw) return;
    src += blockIdx.y * fstride; dst += blockIdx.y * fstride;
    const int y = idx / nw, x = idx - y * nw;
    const int k[5] = {1, 4, 6, 4, 1};
    int s = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint8_t* row = src + (size_t)reflect101(2 * y + i - 2, h) * w;
        int r = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) r += k[j] * row[reflect101(2 * x + j - 2, w)];
        s += k[i] * r;
    }
    dst[idx] = (uint8_t)((s + 128) >> 8);
}

struct LkArgs {
    const uint8_t* prev[kMaxLevels]; const uint8_t* cur[kMaxLevels];
    int h[kMaxLevels], w[kMaxLevels];
    int top, n, win, max_iters, width, height;
    double eps2, min_eig;
    const float* pts; float* next; uint8_t* status;
    // batch of frame pairs (blockIdx.y = pair; mi355_gmc_track_batch): pair p tracks n_arr[p] points from plane p into plane p + 1, the
    // planes of consecutive frames `pair_stride` bytes apart, its points / results at p * max_pts.  One pair: all zero / null.
    size_t pair_stride; int max_pts; const int* n_arr;
};

// Sum over the 64 lanes, the same bits in every lane.  Four row_shr steps inside each row of 16 lanes (data-parallel-primitive moves:
// a few cycles each, against ~100 for the LDS-crossbar permute a generic shuffle becomes), two row broadcasts, one read of lane 63.
// The iteration of the Lucas-Kanade loop is latency-bound and does two of these back to back.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    const double o = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);    // 0.0 where the source lane is outside the row / masked
    return v + o;
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0x111, 0xf>(v);          // row_shr:1
    v = dpp_add<0x112, 0xf>(v);          // row_shr:2
    v = dpp_add<0x114, 0xf>(v);          // row_shr:4
    v = dpp_add<0x118, 0xf>(v);          // row_shr:8  -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xa>(v);          // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);          // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// value of plane `kind` (0 = intensity, 1 = Scharr d/dx, 2 = Scharr d/dy of the same image) at integer position (y, x) of the
// reflect-padded plane: the host code pads the plane of gradients, so the position is reflected first and the stencil
// reflects its own neighbours
template <int KIND>
__device__ __forceinline__ double plane_at(const uint8_t* img, int h, int w, int y, int x) {
    y = reflect101(y, h); x = reflect101(x, w);
    if (KIND == 0) return (double)img[(size_t)y * w + x];
    const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h), xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    auto A = [&](int yy, int xx) { return (int)img[(size_t)yy * w + xx]; };
    if (KIND == 1) return (double)(3 * (A(ym, xp) - A(ym, xm)) + 10 * (A(y, xp) - A(y, xm)) + 3 * (A(yp, xp) - A(yp, xm)));
    return (double)(3 * (A(yp, xm) - A(ym, xm)) + 10 * (A(yp, x) - A(ym, x)) + 3 * (A(yp, xp) - A(ym, xp)));
}

// this lane's samples of the win x win bilinear patch whose top-left corner is (px - half, py - half): window pixel k = lane + 64 j
template <int KIND>
__device__ __forceinline__ void patch(const uint8_t* img, int h, int w, double px, double py, int win, int lane, double (&out)[kMaxPer], int per) {
    const int half = win / 2;
    const double x = px - half, y = py - half;
    const double fx = floor(x), fy = floor(y);
    const int ix = (int)fx, iy = (int)fy;
    const double ax = x - fx, ay = y - fy;
    const double w00 = (1 - ay) * (1 - ax), w01 = (1 - ay) * ax, w10 = ay * (1 - ax), w11 = ay * ax;
    const int W2 = win * win;
#pragma unroll
    for (int j = 0; j < kMaxPer; ++j) {
        if (j >= per) break;
        const int k = lane + 64 * j;
        double v = 0.0;
        if (k < W2) {
            const int r = k / win, c = k - r * win;
            const int yy = iy + r, xx = ix + c;
            v = w00 * plane_at<KIND>(img, h, w, yy, xx) + w01 * plane_at<KIND>(img, h, w, yy, xx + 1) +
                w10 * plane_at<KIND>(img, h, w, yy + 1, xx) + w11 * plane_at<KIND>(img, h, w, yy + 1, xx + 1);
        }
        out[j] = v;
    }
}

// The iteration reads the current frame through a per-wavefront LDS copy of the neighbourhood it is walking in: (win + 1 + 2 M)^2
// reflect-padded intensities (bytes), refilled only when the window's corner leaves the margin M.  The values are those
// plane_at<0> returns, so the arithmetic -- and every bit of the result -- is that of reading global memory each time; what goes
// away is four byte loads with reflected addresses per window pixel per iteration (three quarters of the kernel's instructions).
constexpr int kMargin = 3;
constexpr int kRegMax = 21 + 1 + 2 * kMargin;          // region side for the largest window

__device__ __forceinline__ void fill_region(const uint8_t* img, int h, int w, int ry, int rx, int R, int lane, uint8_t* reg) {
    for (int k = lane; k < R * R; k += 64) {
        const int r = k / R, c = k - r * R;
        reg[k] = img[(size_t)reflect101(ry + r, h) * w + reflect101(rx + c, w)];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The previous frame's window samples of one level (I, Scharr Ix, Iy at the win x win bilinear positions around (px, py)), this lane's
// share.  patch<> evaluates every sample's four corners from global memory -- for the gradients six reflected byte loads per corner,
// 364 loads per lane and level, four fifths of the kernel's time.  Here each INTEGER position of the (win + 1)^2 grid is evaluated
// once: the image bytes the grid and its 3x3 stencils touch (a box of at most (win + 3)^2 pixels after reflection) are copied to LDS,
// the three planes (byte, int16, int16 -- the stencil sums are integers, so nothing is rounded) are built from that copy, and the
// bilinear samples read the planes.  Same integers, same float64 expression per sample: the same bits as patch<>.
constexpr int kGridMax = 21 + 1;                                   // win + 1
struct LkScratch {
    uint8_t box[(21 + 3) * (21 + 3)];                              // image bytes under the grid and its stencils
    uint8_t pI[kGridMax * kGridMax];
    short pIx[kGridMax * kGridMax], pIy[kGridMax * kGridMax];
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void setup_patches(const uint8_t* img, int h, int w, double px, double py, int win, int lane, uint8_t* box, uint8_t* pI,
                                              short* pIx, short* pIy, double (&I)[kMaxPer], double (&Ix)[kMaxPer], double (&Iy)[kMaxPer], int per) {
    const int half = win / 2, G = win + 1;
    const double x = px - half, y = py - half;
    const double fx = floor(x), fy = floor(y);
    const int ix = (int)fx, iy = (int)fy;
    const double ax = x - fx, ay = y - fy;
    const double w00 = (1 - ay) * (1 - ax), w01 = (1 - ay) * ax, w10 = ay * (1 - ax), w11 = ay * ax;
    // image rows / columns the grid's reflected positions fall on (wave-uniform), one pixel more on each side for the stencils
    int ylo = h, yhi = -1, xlo = w, xhi = -1;
    for (int r = 0; r < G; ++r) {
        const int yy = reflect101(iy + r, h), xx = reflect101(ix + r, w);
        ylo = min(ylo, yy); yhi = max(yhi, yy); xlo = min(xlo, xx); xhi = max(xhi, xx);
    }
    ylo = max(0, ylo - 1); yhi = min(h - 1, yhi + 1); xlo = max(0, xlo - 1); xhi = min(w - 1, xhi + 1);
    const int Hb = yhi - ylo + 1, Wb = xhi - xlo + 1;              // <= win + 3 each
    wave_lds_fence();                                              // the previous level's readers of the box and the planes are done
    for (int k = lane; k < Hb * Wb; k += 64) {
        const int r = k / Wb, c = k - r * Wb;
        box[k] = img[(size_t)(ylo + r) * w + (xlo + c)];
    }
    wave_lds_fence();
    auto A = [&](int yy, int xx) { return (int)box[(yy - ylo) * Wb + (xx - xlo)]; };
    for (int k = lane; k < G * G; k += 64) {
        const int r = k / G, c = k - r * G;
        const int yy = reflect101(iy + r, h), xx = reflect101(ix + c, w);
        const int ym = reflect101(yy - 1, h), yp = reflect101(yy + 1, h), xm = reflect101(xx - 1, w), xp = reflect101(xx + 1, w);
        pI[k] = (uint8_t)A(yy, xx);
        pIx[k] = (short)(3 * (A(ym, xp) - A(ym, xm)) + 10 * (A(yy, xp) - A(yy, xm)) + 3 * (A(yp, xp) - A(yp, xm)));
        pIy[k] = (short)(3 * (A(yp, xm) - A(ym, xm)) + 10 * (A(yp, xx) - A(ym, xx)) + 3 * (A(yp, xp) - A(ym, xp)));
    }
    wave_lds_fence();
    const int W2 = win * win;
#pragma unroll
    for (int j = 0; j < kMaxPer; ++j) {
        if (j >= per) break;
        const int k = lane + 64 * j;
        double vi = 0.0, vx = 0.0, vy = 0.0;
        if (k < W2) {
            const int r = k / win, c = k - r * win;
            const int q = r * G + c;
            vi = w00 * (double)pI[q] + w01 * (double)pI[q + 1] + w10 * (double)pI[q + G] + w11 * (double)pI[q + G + 1];
            vx = w00 * (double)pIx[q] + w01 * (double)pIx[q + 1] + w10 * (double)pIx[q + G] + w11 * (double)pIx[q + G + 1];
            vy = w00 * (double)pIy[q] + w01 * (double)pIy[q + 1] + w10 * (double)pIy[q + G] + w11 * (double)pIy[q + G + 1];
        }
        I[j] = vi; Ix[j] = vx; Iy[j] = vy;
    }
}

// one wavefront per point; block = kLkWaves wavefronts (4: eight or sixteen points per block, meant to leave more CUs to the detector
// pass this kernel runs beside, measured equal / 5-10 % slower on the track loop -- tools/lk_waves_ab.sh)
#ifndef MI355_LK_WAVES
#define MI355_LK_WAVES 4
#endif
constexpr int kLkWaves = MI355_LK_WAVES;
__global__ __launch_bounds__(64 * kLkWaves) void lk_kernel(LkArgs a) {
    __shared__ uint8_t region[kLkWaves][(kRegMax * kRegMax + 15) & ~15];
    __shared__ LkScratch scratch[kLkWaves];
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kLkWaves + (threadIdx.x >> 6);
    const int pair = blockIdx.y;
    if (i >= (a.n_arr ? a.n_arr[pair] : a.n)) return;
    {
        const size_t po = (size_t)pair * a.pair_stride;
#pragma unroll
        for (int l = 0; l < kMaxLevels; ++l) { a.prev[l] += po; a.cur[l] += po; }
        const size_t qo = (size_t)pair * (size_t)a.max_pts;
        a.pts += 2 * qo; a.next += 2 * qo; a.status += qo;
    }
    uint8_t* reg = region[threadIdx.x >> 6];
    LkScratch& sc = scratch[threadIdx.x >> 6];
    const int win = a.win, half = win / 2, W2 = win * win;
    const int per = (W2 + 63) / 64;
    const int R = win + 1 + 2 * kMargin;
    const double s = 1.0 / (double)(1 << 20);                    // OpenCV's scaling of the gradient products
    const double p0x = (double)a.pts[2 * i], p0y = (double)a.pts[2 * i + 1];
    bool ok = true;
    double nx = 0, ny = 0;
    double I[kMaxPer], Ix[kMaxPer], Iy[kMaxPer];
    int koff[kMaxPer];                                           // this lane's window pixels as offsets inside the region
#pragma unroll
    for (int j = 0; j < kMaxPer; ++j) {
        const int k = lane + 64 * j;
        const int r = k / win, c = k - r * win;
        koff[j] = (j < per && k < W2) ? r * R + c : -1;
    }
    for (int l = a.top; l >= 0; --l) {
        const int h = a.h[l], w = a.w[l];
        const double px = p0x / (double)(1 << l), py = p0y / (double)(1 << l);
        if (l == a.top) { nx = px; ny = py; } else { nx *= 2.0; ny *= 2.0; }
        const double tlx = floor(px - half), tly = floor(py - half);
        const bool inside = tlx >= -win && tlx < w && tly >= -win && tly < h;
        if (!inside) { if (l == 0) ok = false; continue; }
        const double cx = fmin(fmax(px, (double)-half), (double)(w - 1 + half));
        const double cy = fmin(fmax(py, (double)-half), (double)(h - 1 + half));
#if MI355_LK_GLOBAL_SETUP
        patch<0>(a.prev[l], h, w, cx, cy, win, lane, I, per);
        patch<1>(a.prev[l], h, w, cx, cy, win, lane, Ix, per);
        patch<2>(a.prev[l], h, w, cx, cy, win, lane, Iy, per);
#else
        setup_patches(a.prev[l], h, w, cx, cy, win, lane, sc.box, sc.pI, sc.pIx, sc.pIy, I, Ix, Iy, per);
#endif
        double a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            if (j >= per) break;
            a11 += Ix[j] * Ix[j]; a12 += Ix[j] * Iy[j]; a22 += Iy[j] * Iy[j];
        }
        a11 = wave_sum(a11) * s; a12 = wave_sum(a12) * s; a22 = wave_sum(a22) * s;
        const double det = a11 * a22 - a12 * a12;
        const double mineig = (a22 + a11 - sqrt((a11 - a22) * (a11 - a22) + 4 * a12 * a12)) / (2.0 * W2);
        if (!(mineig >= a.min_eig) || !(det >= (double)1.1920928955078125e-07)) { if (l == 0) ok = false; continue; }
        double pdx = 0, pdy = 0;
        int rx = 0, ry = 0; bool have_region = false;
        for (int it = 0; it < a.max_iters; ++it) {
            const double qx = floor(nx - half), qy = floor(ny - half);
            if (!(qx >= -win && qx < w && qy >= -win && qy < h)) { if (l == 0) ok = false; break; }
            const double ccx = fmin(fmax(nx, (double)-half), (double)(w - 1 + half));
            const double ccy = fmin(fmax(ny, (double)-half), (double)(h - 1 + half));
            // the window's top-left sample (as patch<0> derives it) and its bilinear weights
            const double x = ccx - half, y = ccy - half;
            const double fx = floor(x), fy = floor(y);
            const int ix = (int)fx, iy = (int)fy;
            const double ax = x - fx, ay = y - fy;
            const double w00 = (1 - ay) * (1 - ax), w01 = (1 - ay) * ax, w10 = ay * (1 - ax), w11 = ay * ax;
            if (!have_region || ix < rx || iy < ry || ix > rx + 2 * kMargin || iy > ry + 2 * kMargin) {
                __builtin_amdgcn_wave_barrier();                   // every lane is done reading the old region
                rx = ix - kMargin; ry = iy - kMargin;
                fill_region(a.cur[l], h, w, ry, rx, R, lane, reg);
                have_region = true;
            }
            const uint8_t* r0 = reg + (iy - ry) * R + (ix - rx);
            double b1 = 0, b2 = 0;
#pragma unroll
            for (int j = 0; j < kMaxPer; ++j) {
                if (j >= per) break;
                double Jv = 0.0;                                 // lanes past the window hold zeros in I / Ix / Iy too
                if (koff[j] >= 0) {
                    const uint8_t* q = r0 + koff[j];
                    Jv = w00 * (double)q[0] + w01 * (double)q[1] + w10 * (double)q[R] + w11 * (double)q[R + 1];
                }
                const double d = (Jv - I[j]) * 32.0;
                b1 += d * Ix[j]; b2 += d * Iy[j];
            }
            b1 = wave_sum(b1) * s; b2 = wave_sum(b2) * s;
            const double dx = (a12 * b2 - a22 * b1) / det, dy = (a12 * b1 - a11 * b2) / det;
            nx += dx; ny += dy;
            if (dx * dx + dy * dy <= a.eps2) break;
            if (it > 0 && fabs(dx + pdx) < 0.01 && fabs(dy + pdy) < 0.01) { nx -= dx * 0.5; ny -= dy * 0.5; break; }
            pdx = dx; pdy = dy;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (ok && (nx < 0 || ny < 0 || nx >= a.width || ny >= a.height)) ok = false;
    if (lane == 0) {
        a.next[2 * i] = (float)nx; a.next[2 * i + 1] = (float)ny;
        a.status[i] = ok ? 1 : 0;
    }
}

// ---- frame preparation: cvtColor(BGR2GRAY) + resize(INTER_LINEAR) + the corner map of goodFeaturesToTrack -------------------
// gmc.py states these in numpy (bgr_to_gray, resize_linear, good_features_to_track); the kernels evaluate the same expressions
// in the same order (integers for luma and resize; float64 for the structure tensor, its sums taken in numpy's order), so the
// corner list is the numpy one.

// gray = (1868 B + 9617 G + 4899 R + 8192) >> 14, then INTER_LINEAR with 11-bit coefficients (tables from the host: source index,
// two taps per output column / row), both passes in cv2's fixed point: ((c0 * (h0 >> 4)) >> 16) + ((c1 * (h1 >> 4)) >> 16) + 2) >> 2
__global__ __launch_bounds__(256) void gray_resize_kernel(const uint8_t* bgr, int H, int W, const int* xtab, const int* ytab, uint8_t* out, int oh, int ow,
                                                          int resize, size_t in_fstride = 0, size_t out_fstride = 0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= oh * ow) return;
    bgr += blockIdx.y * in_fstride; out += blockIdx.y * out_fstride;
    const int y = idx / ow, x = idx - y * ow;
    auto gray = [&](int yy, int xx) {
        const uint8_t* p = bgr + ((size_t)yy * W + xx) * 3;
        return (int)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
    };
    if (!resize) { out[idx] = (uint8_t)gray(y, x); return; }
    const int xi = xtab[3 * x], xa0 = xtab[3 * x + 1], xa1 = xtab[3 * x + 2];
    const int yi = ytab[3 * y], yb0 = ytab[3 * y + 1], yb1 = ytab[3 * y + 2];
    const int xj = min(xi + 1, W - 1), yj = min(yi + 1, H - 1);
    const int h0 = gray(yi, xi) * xa0 + gray(yi, xj) * xa1;
    const int h1 = gray(yj, xi) * xa0 + gray(yj, xj) * xa1;
    int v = (((yb0 * (h0 >> 4)) >> 16) + ((yb1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    out[idx] = (uint8_t)v;
}

// cornerMinEigenVal (3x3 Sobel scaled by 1 / (4 * block * 255), block x block box sums of the products, smaller eigenvalue) as
// float32, and its maximum over the plane (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void min_eig_kernel(const uint8_t* g, int h, int w, float* eig, unsigned* max_bits, size_t g_fstride = 0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    g += blockIdx.y * g_fstride; eig += (size_t)blockIdx.y * h * w; max_bits += blockIdx.y;
    float e = 0.f;
    if (idx < h * w) {
        const int y = idx / w, x = idx - y * w;
        const double sc = 1.0 / (4.0 * 3.0 * 255.0);
        auto G = [&](int yy, int xx) { return (double)g[(size_t)reflect101(yy, h) * w + reflect101(xx, w)]; };
        double sxx = 0.0, sxy = 0.0, syy = 0.0;                 // Python's sum(): 0 + p00 + p01 + ... in (i, j) row-major order
#pragma unroll
        for (int i = -1; i <= 1; ++i)
#pragma unroll
            for (int j = -1; j <= 1; ++j) {
                const int yy = reflect101(y + i, h), xx = reflect101(x + j, w);     // the product arrays are reflect-padded
                const double dx = ((G(yy - 1, xx + 1) - G(yy - 1, xx - 1)) + 2 * (G(yy, xx + 1) - G(yy, xx - 1)) + (G(yy + 1, xx + 1) - G(yy + 1, xx - 1))) * sc;
                const double dy = ((G(yy + 1, xx - 1) - G(yy - 1, xx - 1)) + 2 * (G(yy + 1, xx) - G(yy - 1, xx)) + (G(yy + 1, xx + 1) - G(yy - 1, xx + 1))) * sc;
                sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
            }
        const double a = sxx * 0.5, b = sxy, c = syy * 0.5;
        e = (float)((a + c) - sqrt((a - c) * (a - c) + b * b));
        eig[idx] = e;
    }
    // block maximum, then one atomic per block
    float m = e > 0.f ? e : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(max_bits, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// THRESH_TOZERO at quality * max, 3x3 non-maximum suppression (a corner equals the maximum of its neighbourhood), image border excluded
__global__ __launch_bounds__(256) void corner_mask_kernel(const float* eig, int h, int w, const unsigned* max_bits, double quality, uint8_t* ok) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * w) return;
    eig += (size_t)blockIdx.y * h * w; ok += (size_t)blockIdx.y * h * w; max_bits += blockIdx.y;
    const int y = idx / w, x = idx - y * w;
    const float mx = __uint_as_float(*max_bits);
    const float thr = (float)((double)mx * quality);
    auto T = [&](int yy, int xx) {
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) return -INFINITY;
        const float v = eig[(size_t)yy * w + xx];
        return v > thr ? v : 0.f;
    };
    const float v = T(y, x);
    float d = -INFINITY;
#pragma unroll
    for (int i = -1; i <= 1; ++i)
#pragma unroll
        for (int j = -1; j <= 1; ++j) d = fmaxf(d, T(y + i, x + j));
    const bool keep = mx > 0.f && v != 0.f && v == d && y > 0 && y < h - 1 && x > 0 && x < w - 1;
    ok[idx] = keep ? 1 : 0;
}

// per-device scratch, grow-only; calls are serialised (the tracker is sequential per video)
struct GmcCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    uint8_t* d_planes = nullptr; size_t planes_cap = 0;        // [prev pyramid | cur pyramid]
    float* d_pts = nullptr; float* d_next = nullptr; uint8_t* d_status = nullptr; int pts_cap = 0;
    uint8_t* h_pin = nullptr; size_t pin_cap = 0;              // pinned staging: frames + points in, points + status out
    uint8_t* d_front = nullptr; size_t front_cap = 0;          // frame preparation: [bgr | gray | eig | ok | tables | max]
    uint8_t* h_front = nullptr; size_t hfront_cap = 0;
};
std::mutex g_mu;
GmcCtx g_ctx[16];

#define GCHK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return -2; } } while (0)

}  // namespace

// Same contract as mi355_gmc_pyr_lk (gmc_host.cpp) with the work done on GPU `device`: 0 = ok, -1 = bad argument, -2 = HIP error.
extern "C" int mi355_gmc_pyr_lk_device(int device, const uint8_t* prev, const uint8_t* cur, int height, int width, const float* pts, int n,
                                       int win, int max_level, int max_iters, double eps, double min_eig, float* next_pts, uint8_t* status) {
    if (!prev || !cur || height <= 0 || width <= 0 || n < 0 || (n > 0 && (!pts || !next_pts || !status)) || win < 3 || !(win & 1) || win > 21 ||
        device < 0 || device >= 16 || max_level < 0)
        return -1;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lock(g_mu);
    GCHK(hipSetDevice(device));
    GmcCtx& c = g_ctx[device];
    if (!c.stream) { GCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking)); c.device = device; }
    // level geometry (buildOpticalFlowPyramid stops at levels not larger than the window)
    int hs[kMaxLevels], ws[kMaxLevels], levels = 1;
    hs[0] = height; ws[0] = width;
    for (int l = 0; l < max_level && levels < kMaxLevels; ++l) {
        const int nh = (hs[levels - 1] + 1) / 2, nw = (ws[levels - 1] + 1) / 2;
        if (nh <= win || nw <= win) break;
        hs[levels] = nh; ws[levels] = nw; ++levels;
    }
    size_t off[kMaxLevels], pyr_bytes = 0;
    for (int l = 0; l < levels; ++l) { off[l] = pyr_bytes; pyr_bytes += ((size_t)hs[l] * ws[l] + 255) & ~(size_t)255; }
    if (c.planes_cap < 2 * pyr_bytes) {
        if (c.d_planes) (void)hipFree(c.d_planes);
        c.d_planes = nullptr; c.planes_cap = 0;
        GCHK(hipMalloc(&c.d_planes, 2 * pyr_bytes)); c.planes_cap = 2 * pyr_bytes;
    }
    if (c.pts_cap < n) {
        if (c.d_pts) (void)hipFree(c.d_pts); if (c.d_next) (void)hipFree(c.d_next); if (c.d_status) (void)hipFree(c.d_status);
        c.d_pts = c.d_next = nullptr; c.d_status = nullptr; c.pts_cap = 0;
        const int cap = std::max(1024, n);
        GCHK(hipMalloc(&c.d_pts, (size_t)cap * 8)); GCHK(hipMalloc(&c.d_next, (size_t)cap * 8)); GCHK(hipMalloc(&c.d_status, (size_t)cap));
        c.pts_cap = cap;
    }
    const size_t frame = (size_t)height * width;
    const size_t pin_need = 2 * frame + (size_t)n * 8 + (size_t)n * 8 + (size_t)n + 64;
    if (c.pin_cap < pin_need) {
        if (c.h_pin) (void)hipHostFree(c.h_pin);
        c.h_pin = nullptr; c.pin_cap = 0;
        GCHK(hipHostMalloc(&c.h_pin, pin_need)); c.pin_cap = pin_need;
    }
    uint8_t* h_prev = c.h_pin; uint8_t* h_cur = h_prev + frame;
    float* h_pts = (float*)(c.h_pin + ((2 * frame + 15) & ~(size_t)15));               // 16-byte aligned behind the frames
    float* h_next = h_pts + 2 * (size_t)n;
    uint8_t* h_status = (uint8_t*)(h_next + 2 * (size_t)n);
    std::memcpy(h_prev, prev, frame); std::memcpy(h_cur, cur, frame); std::memcpy(h_pts, pts, (size_t)n * 8);
    uint8_t* dp = c.d_planes; uint8_t* dc = c.d_planes + pyr_bytes;
    GCHK(hipMemcpyAsync(dp, h_prev, frame, hipMemcpyHostToDevice, c.stream));
    GCHK(hipMemcpyAsync(dc, h_cur, frame, hipMemcpyHostToDevice, c.stream));
    GCHK(hipMemcpyAsync(c.d_pts, h_pts, (size_t)n * 8, hipMemcpyHostToDevice, c.stream));
    LkArgs a{};
    for (int l = 0; l < levels; ++l) { a.prev[l] = dp + off[l]; a.cur[l] = dc + off[l]; a.h[l] = hs[l]; a.w[l] = ws[l]; }
    for (int l = 1; l < levels; ++l) {
        const int np = hs[l] * ws[l];
        hipLaunchKernelGGL(pyr_down_kernel, dim3((np + 255) / 256), dim3(256), 0, c.stream, dp + off[l - 1], hs[l - 1], ws[l - 1], dp + off[l], hs[l], ws[l]);
        hipLaunchKernelGGL(pyr_down_kernel, dim3((np + 255) / 256), dim3(256), 0, c.stream, dc + off[l - 1], hs[l - 1], ws[l - 1], dc + off[l], hs[l], ws[l]);
    }
    a.top = levels - 1; a.n = n; a.win = win; a.max_iters = max_iters; a.width = width; a.height = height;
    a.eps2 = eps * eps; a.min_eig = min_eig;
    a.pts = c.d_pts; a.next = c.d_next; a.status = c.d_status;
    hipLaunchKernelGGL(lk_kernel, dim3((n + kLkWaves - 1) / kLkWaves), dim3(64 * kLkWaves), 0, c.stream, a);
    GCHK(hipGetLastError());
    GCHK(hipMemcpyAsync(h_next, c.d_next, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
    GCHK(hipMemcpyAsync(h_status, c.d_status, (size_t)n, hipMemcpyDeviceToHost, c.stream));
    GCHK(hipStreamSynchronize(c.stream));
    std::memcpy(next_pts, h_next, (size_t)n * 8); std::memcpy(status, h_status, (size_t)n);
    return 0;
}

// Frame preparation of GMC.apply on GPU `device`: BGR frame [height][width][3] -> gray plane of (height / downscale) x (width /
// downscale) (cv2.cvtColor + cv2.resize INTER_LINEAR, bit for bit the fixed-point arithmetic gmc.py states), the float32
// min-eigenvalue map of cornerMinEigenVal and the 0/1 mask of the corners goodFeaturesToTrack keeps before it orders them
// (quality threshold, 3x3 non-maximum suppression, border excluded).  xtab / ytab: per output column / row (source index, tap 0,
// tap 1) as gmc._linear_coeffs gives them; ignored when downscale == 1.  Outputs are host buffers of oh * ow elements.
extern "C" int mi355_gmc_prepare_device(int device, const uint8_t* bgr, int height, int width, int oh, int ow, const int* xtab, const int* ytab,
                                        double quality, uint8_t* gray_out, float* eig_out, uint8_t* ok_out) {
    if (!bgr || height <= 0 || width <= 0 || oh <= 0 || ow <= 0 || !gray_out || !eig_out || !ok_out || device < 0 || device >= 16) return -1;
    const int resize = !(oh == height && ow == width);
    if (resize && (!xtab || !ytab)) return -1;
    std::lock_guard<std::mutex> lock(g_mu);
    GCHK(hipSetDevice(device));
    GmcCtx& c = g_ctx[device];
    if (!c.stream) { GCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking)); c.device = device; }
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nb = (size_t)height * width * 3, np = (size_t)oh * ow;
    const size_t o_gray = al(nb), o_eig = o_gray + al(np), o_ok = o_eig + al(np * 4), o_xt = o_ok + al(np), o_yt = o_xt + al((size_t)ow * 12),
                 o_max = o_yt + al((size_t)oh * 12), total = o_max + 256;
    if (c.front_cap < total) {
        if (c.d_front) (void)hipFree(c.d_front);
        c.d_front = nullptr; c.front_cap = 0;
        GCHK(hipMalloc(&c.d_front, total)); c.front_cap = total;
    }
    const size_t h_in = al(nb) + al((size_t)ow * 12) + al((size_t)oh * 12), h_total = h_in + al(np) + al(np * 4) + al(np);
    if (c.hfront_cap < h_total) {
        if (c.h_front) (void)hipHostFree(c.h_front);
        c.h_front = nullptr; c.hfront_cap = 0;
        GCHK(hipHostMalloc(&c.h_front, h_total)); c.hfront_cap = h_total;
    }
    uint8_t* hp = c.h_front;
    uint8_t* h_bgr = hp; uint8_t* h_xt = hp + al(nb); uint8_t* h_yt = h_xt + al((size_t)ow * 12);
    uint8_t* h_gray = hp + h_in; uint8_t* h_eig = h_gray + al(np); uint8_t* h_ok = h_eig + al(np * 4);
    std::memcpy(h_bgr, bgr, nb);
    GCHK(hipMemcpyAsync(c.d_front, h_bgr, nb, hipMemcpyHostToDevice, c.stream));
    if (resize) {
        std::memcpy(h_xt, xtab, (size_t)ow * 12); std::memcpy(h_yt, ytab, (size_t)oh * 12);
        GCHK(hipMemcpyAsync(c.d_front + o_xt, h_xt, (size_t)ow * 12, hipMemcpyHostToDevice, c.stream));
        GCHK(hipMemcpyAsync(c.d_front + o_yt, h_yt, (size_t)oh * 12, hipMemcpyHostToDevice, c.stream));
    }
    GCHK(hipMemsetAsync(c.d_front + o_max, 0, 4, c.stream));
    const unsigned blocks = (unsigned)((np + 255) / 256);
    hipLaunchKernelGGL(gray_resize_kernel, dim3(blocks), dim3(256), 0, c.stream, c.d_front, height, width, (const int*)(c.d_front + o_xt),
                       (const int*)(c.d_front + o_yt), c.d_front + o_gray, oh, ow, resize);
    hipLaunchKernelGGL(min_eig_kernel, dim3(blocks), dim3(256), 0, c.stream, c.d_front + o_gray, oh, ow, (float*)(c.d_front + o_eig),
                       (unsigned*)(c.d_front + o_max));
    hipLaunchKernelGGL(corner_mask_kernel, dim3(blocks), dim3(256), 0, c.stream, (const float*)(c.d_front + o_eig), oh, ow,
                       (const unsigned*)(c.d_front + o_max), quality, c.d_front + o_ok);
    GCHK(hipGetLastError());
    GCHK(hipMemcpyAsync(h_gray, c.d_front + o_gray, np, hipMemcpyDeviceToHost, c.stream));
    GCHK(hipMemcpyAsync(h_eig, c.d_front + o_eig, np * 4, hipMemcpyDeviceToHost, c.stream));
    GCHK(hipMemcpyAsync(h_ok, c.d_front + o_ok, np, hipMemcpyDeviceToHost, c.stream));
    GCHK(hipStreamSynchronize(c.stream));
    std::memcpy(gray_out, h_gray, np); std::memcpy(eig_out, h_eig, np * 4); std::memcpy(ok_out, h_ok, np);
    return 0;
}

// ---- one motion-compensation step as two calls: enqueue, then collect ------------------------------------------------------------
// model.track() enqueues the step for a frame BEFORE the detector runs on it and collects it when the tracker asks for the warp: the
// frame preparation and the optical flow (0.7 ms for a thousand corners) then run beside the detector pass on a stream of their own
// instead of after it.  The object keeps the previous frame's pyramid on the device (two slots, swapped per step).
struct mi355_gmc {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_front = nullptr; size_t front_cap = 0;          // [bgr | eig | ok | x table | y table | max]
    uint8_t* d_pyr[2] = {nullptr, nullptr}; size_t pyr_cap = 0;
    int slot = 0; bool have_prev = false; int ph = 0, pw = 0;  // the slot and plane size of the last prepared frame
    int tab_key[4] = {0, 0, 0, 0};                             // (height, width, oh, ow) the resize tables on the device belong to
    float* d_pts = nullptr; float* d_next = nullptr; uint8_t* d_status = nullptr; int pts_cap = 0;
    uint8_t* h_pin = nullptr; size_t pin_cap = 0;
    hipEvent_t ev_up = nullptr; int up_h = 0, up_w = 0;        // recorded behind the pending step's frame upload (mi355_gmc_pending_frame)
    // the pending step
    bool pending = false; int oh = 0, ow = 0, n_lk = 0;
    size_t o_hgray = 0, o_heig = 0, o_hok = 0, o_hnext = 0, o_hstatus = 0;
    // ---- GMC.apply_sparseoptflow's state machine (mi355_gmc_track_*): the previous frame's plane and ordered corners live here, so a
    // step is two calls from the tracker's language binding (enqueue, collect -> 2 x 3 matrix) and nothing per frame is done in it
    bool host = false;                                         // mi355_gmc_create(-1): every stage in host C++ (csrc/gmc_host.cpp)
    int downscale = 2;
    std::vector<int> xt, yt; int tkey[4] = {0, 0, 0, 0};       // INTER_LINEAR tables of (height, width, oh, ow)
    std::vector<uint8_t> prev_gray, cur_gray, ok; std::vector<float> eig;
    std::vector<float> prev_pts, lk_pts, next_pts; std::vector<uint8_t> status;
    int prev_h = 0, prev_w = 0; bool have_prev_pts = false;
    bool track_pending = false; int t_oh = 0, t_ow = 0, t_n = 0;
    std::vector<uint8_t> host_frame; int hf_h = 0, hf_w = 0;   // host object: the frame of the pending step
    // collect worker (device objects): the host half of a step -- wait for the stream, order the new corners, RANSAC -- runs on a thread of
    // its own from the moment track_begin has enqueued the step, i.e. beside the detector pass the caller runs next; track_finish joins it
    std::thread worker; std::mutex mu; std::condition_variable cv;
    bool job_ready = false, job_done = false, worker_stop = false; int job_rc = 0; double job_H[6] = {1, 0, 0, 0, 1, 0};
    // mi355_gmc_batch_frames: the frames of the batch a (concurrent) mi355_gmc_track_batch call has uploaded
    hipEvent_t ev_batch_up = nullptr; unsigned long long batch_up_seq = 0; int batch_n = 0, batch_h = 0, batch_w = 0; size_t batch_fstride = 0;
    bool batch_failed = false;
    bool job_active = false;                                   // written by the calling thread only: this step's collect belongs to the worker
    // mi355_gmc_track_batch: device buffers of one batch (grow-only) and their pinned mirror
    uint8_t* d_batch = nullptr; size_t batch_cap = 0;
    uint8_t* h_batch = nullptr; size_t hbatch_cap = 0;
};

extern "C" int mi355_gmc_create(int device, mi355_gmc** out) {
    if (!out || device < -1) return -1;
    *out = nullptr;
    if (device == -1) {                                         // host object: no HIP call is ever made through it
        mi355_gmc* g = new mi355_gmc();
        g->device = -1; g->host = true;
        *out = g;
        return 0;
    }
    GCHK(hipSetDevice(device));
    mi355_gmc* g = new mi355_gmc();
    g->device = device;
    // The step runs BESIDE the detector pass (model.track enqueues it first), on a stream of its own at the default priority.  Measured
    // (tools/track_prio_ab.sh, round 4): giving this stream the LOWEST and the detector's the HIGHEST priority does not speed the detector
    // up (637-644 us per frame either way) and delays the collect (119-139 -> 163-171 us): 805-838 -> 863-893 us per frame.
    // MI355_GMC_PRIO=1 asks for the lowest priority (A/B only).
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    static const bool low_prio = getenv("MI355_GMC_PRIO") && atoi(getenv("MI355_GMC_PRIO")) == 1;
    if (hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, low_prio ? least : 0) != hipSuccess) { (void)hipGetLastError(); delete g; return -2; }
    *out = g;
    return 0;
}

extern "C" void mi355_gmc_destroy(mi355_gmc* g) {
    if (g && g->worker.joinable()) {
        { std::lock_guard<std::mutex> lk(g->mu); g->worker_stop = true; }
        g->cv.notify_all();
        g->worker.join();
    }
    if (!g) return;
    if (g->host) { delete g; return; }
    (void)hipSetDevice(g->device);
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    if (g->ev_up) (void)hipEventDestroy(g->ev_up);
    if (g->ev_batch_up) (void)hipEventDestroy(g->ev_batch_up);
    if (g->d_front) (void)hipFree(g->d_front);
    for (int i = 0; i < 2; ++i) if (g->d_pyr[i]) (void)hipFree(g->d_pyr[i]);
    if (g->d_pts) (void)hipFree(g->d_pts); if (g->d_next) (void)hipFree(g->d_next); if (g->d_status) (void)hipFree(g->d_status);
    if (g->h_pin) (void)hipHostFree(g->h_pin);
    if (g->d_batch) (void)hipFree(g->d_batch);
    if (g->h_batch) (void)hipHostFree(g->h_batch);
    delete g;
}

// Enqueue one step: frame preparation of `bgr` (as mi355_gmc_prepare_device) and, when n_prev > 0, Lucas-Kanade tracking of the
// n_prev points `prev_pts` from the PREVIOUS step's plane into this one (as mi355_gmc_pyr_lk_device; the previous step must have
// prepared a plane of the same oh x ow).  Returns at once; nothing of `bgr` / `prev_pts` is read after the call returns.
extern "C" int mi355_gmc_step_begin(mi355_gmc* g, const uint8_t* bgr, int height, int width, int oh, int ow, const int* xtab, const int* ytab,
                                    double quality, const float* prev_pts, int n_prev, int win, int max_level, int max_iters, double eps,
                                    double min_eig) {
    if (g && g->host) return -1;
    if (!g || !bgr || height <= 0 || width <= 0 || oh <= 0 || ow <= 0 || n_prev < 0 || (n_prev > 0 && !prev_pts) || win < 3 || !(win & 1) || win > 21 ||
        max_level < 0)
        return -1;
    const int resize = !(oh == height && ow == width);
    if (resize && (!xtab || !ytab)) return -1;
    if (g->pending) return -1;                                  // collect the previous step first
    if (n_prev > 0 && !(g->have_prev && g->ph == oh && g->pw == ow)) return -1;
    GCHK(hipSetDevice(g->device));
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nb = (size_t)height * width * 3, np = (size_t)oh * ow;
    // pyramid geometry of the oh x ow plane
    int hs[kMaxLevels], ws[kMaxLevels], levels = 1;
    hs[0] = oh; ws[0] = ow;
    for (int l = 0; l < max_level && levels < kMaxLevels; ++l) {
        const int nh = (hs[levels - 1] + 1) / 2, nw = (ws[levels - 1] + 1) / 2;
        if (nh <= win || nw <= win) break;
        hs[levels] = nh; ws[levels] = nw; ++levels;
    }
    size_t off[kMaxLevels], pyr_bytes = 0;
    for (int l = 0; l < levels; ++l) { off[l] = pyr_bytes; pyr_bytes += al((size_t)hs[l] * ws[l]); }
    if (g->pyr_cap < pyr_bytes) {
        for (int i = 0; i < 2; ++i) { if (g->d_pyr[i]) (void)hipFree(g->d_pyr[i]); g->d_pyr[i] = nullptr; }
        g->pyr_cap = 0; g->have_prev = false;
        if (n_prev > 0) return -1;
        GCHK(hipMalloc(&g->d_pyr[0], pyr_bytes)); GCHK(hipMalloc(&g->d_pyr[1], pyr_bytes)); g->pyr_cap = pyr_bytes;
    }
    const size_t o_eig = al(nb), o_ok = o_eig + al(np * 4), o_xt = o_ok + al(np), o_yt = o_xt + al((size_t)ow * 12), o_max = o_yt + al((size_t)oh * 12),
                 total = o_max + 256;
    if (g->front_cap < total) {
        if (g->d_front) (void)hipFree(g->d_front);
        g->d_front = nullptr; g->front_cap = 0; g->tab_key[0] = 0;
        GCHK(hipMalloc(&g->d_front, total)); g->front_cap = total;
    }
    if (g->pts_cap < n_prev) {
        if (g->d_pts) (void)hipFree(g->d_pts); if (g->d_next) (void)hipFree(g->d_next); if (g->d_status) (void)hipFree(g->d_status);
        g->d_pts = g->d_next = nullptr; g->d_status = nullptr; g->pts_cap = 0;
        const int cap = std::max(1024, n_prev);
        GCHK(hipMalloc(&g->d_pts, (size_t)cap * 8)); GCHK(hipMalloc(&g->d_next, (size_t)cap * 8)); GCHK(hipMalloc(&g->d_status, (size_t)cap));
        g->pts_cap = cap;
    }
    // pinned staging: [bgr | x table | y table | prev points] in, [gray | eig | ok | next points | status] out
    const size_t i_xt = al(nb), i_yt = i_xt + al((size_t)ow * 12), i_pts = i_yt + al((size_t)oh * 12), i_end = i_pts + al((size_t)n_prev * 8);
    g->o_hgray = i_end; g->o_heig = g->o_hgray + al(np); g->o_hok = g->o_heig + al(np * 4); g->o_hnext = g->o_hok + al(np);
    g->o_hstatus = g->o_hnext + al((size_t)n_prev * 8);
    const size_t pin_need = g->o_hstatus + al((size_t)n_prev) + 256;
    if (g->pin_cap < pin_need) {
        if (g->h_pin) (void)hipHostFree(g->h_pin);
        g->h_pin = nullptr; g->pin_cap = 0;
        GCHK(hipHostMalloc(&g->h_pin, pin_need)); g->pin_cap = pin_need;
    }
    uint8_t* hp = g->h_pin;
    std::memcpy(hp, bgr, nb);
    GCHK(hipMemcpyAsync(g->d_front, hp, nb, hipMemcpyHostToDevice, g->stream));
    if (!g->ev_up) GCHK(hipEventCreateWithFlags(&g->ev_up, hipEventDisableTiming));
    GCHK(hipEventRecord(g->ev_up, g->stream)); g->up_h = height; g->up_w = width;
    if (resize && !(g->tab_key[0] == height && g->tab_key[1] == width && g->tab_key[2] == oh && g->tab_key[3] == ow)) {
        std::memcpy(hp + i_xt, xtab, (size_t)ow * 12); std::memcpy(hp + i_yt, ytab, (size_t)oh * 12);
        GCHK(hipMemcpyAsync(g->d_front + o_xt, hp + i_xt, (size_t)ow * 12, hipMemcpyHostToDevice, g->stream));
        GCHK(hipMemcpyAsync(g->d_front + o_yt, hp + i_yt, (size_t)oh * 12, hipMemcpyHostToDevice, g->stream));
        g->tab_key[0] = height; g->tab_key[1] = width; g->tab_key[2] = oh; g->tab_key[3] = ow;
    }
    GCHK(hipMemsetAsync(g->d_front + o_max, 0, 4, g->stream));
    const int nslot = g->slot ^ 1;
    uint8_t* dc = g->d_pyr[nslot];
    uint8_t* dp = g->d_pyr[g->slot];
    const unsigned blocks = (unsigned)((np + 255) / 256);
    hipLaunchKernelGGL(gray_resize_kernel, dim3(blocks), dim3(256), 0, g->stream, g->d_front, height, width, (const int*)(g->d_front + o_xt),
                       (const int*)(g->d_front + o_yt), dc + off[0], oh, ow, resize);
    hipLaunchKernelGGL(min_eig_kernel, dim3(blocks), dim3(256), 0, g->stream, dc + off[0], oh, ow, (float*)(g->d_front + o_eig),
                       (unsigned*)(g->d_front + o_max));
    hipLaunchKernelGGL(corner_mask_kernel, dim3(blocks), dim3(256), 0, g->stream, (const float*)(g->d_front + o_eig), oh, ow,
                       (const unsigned*)(g->d_front + o_max), quality, g->d_front + o_ok);
    for (int l = 1; l < levels; ++l) {
        const int npx = hs[l] * ws[l];
        hipLaunchKernelGGL(pyr_down_kernel, dim3((npx + 255) / 256), dim3(256), 0, g->stream, dc + off[l - 1], hs[l - 1], ws[l - 1], dc + off[l], hs[l], ws[l]);
    }
    if (n_prev > 0) {
        std::memcpy(hp + i_pts, prev_pts, (size_t)n_prev * 8);
        GCHK(hipMemcpyAsync(g->d_pts, hp + i_pts, (size_t)n_prev * 8, hipMemcpyHostToDevice, g->stream));
        LkArgs a{};
        for (int l = 0; l < levels; ++l) { a.prev[l] = dp + off[l]; a.cur[l] = dc + off[l]; a.h[l] = hs[l]; a.w[l] = ws[l]; }
        a.top = levels - 1; a.n = n_prev; a.win = win; a.max_iters = max_iters; a.width = ow; a.height = oh;
        a.eps2 = eps * eps; a.min_eig = min_eig;
        a.pts = g->d_pts; a.next = g->d_next; a.status = g->d_status;
        hipLaunchKernelGGL(lk_kernel, dim3((n_prev + kLkWaves - 1) / kLkWaves), dim3(64 * kLkWaves), 0, g->stream, a);
        GCHK(hipMemcpyAsync(hp + g->o_hnext, g->d_next, (size_t)n_prev * 8, hipMemcpyDeviceToHost, g->stream));
        GCHK(hipMemcpyAsync(hp + g->o_hstatus, g->d_status, (size_t)n_prev, hipMemcpyDeviceToHost, g->stream));
    }
    GCHK(hipGetLastError());
    GCHK(hipMemcpyAsync(hp + g->o_hgray, dc + off[0], np, hipMemcpyDeviceToHost, g->stream));
    GCHK(hipMemcpyAsync(hp + g->o_heig, g->d_front + o_eig, np * 4, hipMemcpyDeviceToHost, g->stream));
    GCHK(hipMemcpyAsync(hp + g->o_hok, g->d_front + o_ok, np, hipMemcpyDeviceToHost, g->stream));
    g->slot = nslot; g->have_prev = true; g->ph = oh; g->pw = ow;
    g->pending = true; g->oh = oh; g->ow = ow; g->n_lk = n_prev;
    return 0;
}

// The frame of the pending step as it sits on the device (dense BGR [height][width][3], uploaded once by mi355_gmc_step_begin): waits -- on
// the host, ~10 us -- until that upload has landed and hands out the pointer, so that the detector pass of the same frame
// (mi355_yolo_infer_device) reads this copy instead of uploading its own.  Besides saving the second upload this is what lets the two
// overlap at all: a second host -> device copy queues up behind the step's device -> host copies, which wait for its Lucas-Kanade launch
// (measured: the detector's first kernel started when the whole step was over, tools/track_timeline.py).  Valid until the next step_begin.
extern "C" int mi355_gmc_pending_frame(mi355_gmc* g, const uint8_t** dev_bgr, int* height, int* width) {
    if (!g || g->host || !(g->pending || g->track_pending) || !g->ev_up || !dev_bgr || !height || !width) return -1;
    GCHK(hipSetDevice(g->device));
    GCHK(hipEventSynchronize(g->ev_up));
    *dev_bgr = g->d_front; *height = g->up_h; *width = g->up_w;
    return 0;
}

// Collect the enqueued step: gray / eig / ok of oh * ow elements, next_pts [n_prev][2] and status [n_prev] (untouched when the step
// had n_prev == 0).
extern "C" int mi355_gmc_step_finish(mi355_gmc* g, uint8_t* gray_out, float* eig_out, uint8_t* ok_out, float* next_pts, uint8_t* status) {
    if (!g || g->host || !g->pending || !gray_out || !eig_out || !ok_out || (g->n_lk > 0 && (!next_pts || !status))) return -1;
    GCHK(hipSetDevice(g->device));
    g->pending = false;
    if (hipStreamSynchronize(g->stream) != hipSuccess) { (void)hipGetLastError(); g->have_prev = false; return -2; }
    const size_t np = (size_t)g->oh * g->ow;
    std::memcpy(gray_out, g->h_pin + g->o_hgray, np); std::memcpy(eig_out, g->h_pin + g->o_heig, np * 4); std::memcpy(ok_out, g->h_pin + g->o_hok, np);
    if (g->n_lk > 0) { std::memcpy(next_pts, g->h_pin + g->o_hnext, (size_t)g->n_lk * 8); std::memcpy(status, g->h_pin + g->o_hstatus, (size_t)g->n_lk); }
    return 0;
}

// ---- the whole step of GMC.apply_sparseoptflow on the object (ultralytics/trackers/utils/gmc.py, reached from /root/reference/model.py:38) ----
// track_begin = enqueue: frame preparation of `bgr` and Lucas-Kanade tracking of the previous frame's corners into it (GPU object: on the
// object's stream, returns at once; host object: the frame is copied and the work happens in track_finish).  track_finish = collect:
// orders the new frame's corners (kept for the next step), estimates the partial affine transform prev -> cur from the tracked pairs
// (RANSAC, seed 0) when more than 4 survive, scales its translation back to frame pixels.  H_out: 6 doubles, row-major 2 x 3; the
// identity on the first frame of a plane size, when the previous frame had no corners, or when too few points were tracked.
namespace {
constexpr int kMaxCorners = 1000, kLkWin = 21, kLkLevels = 3, kLkIters = 30;
constexpr double kQuality = 0.01, kLkEps = 0.01, kLkMinEig = 1e-4, kRansacThr = 3.0, kRansacConf = 0.99;
constexpr int kRansacIters = 2000;

// INTER_LINEAR sample table of a dn-long axis resampled from sn: (source index, tap 0, tap 1), 11-bit taps, float32 coordinates as cv2
void linear_table(int dn, int sn, std::vector<int>& tab) {
    tab.resize((size_t)dn * 3);
    const double scale = (double)sn / dn;
    for (int d = 0; d < dn; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)floorf(fx);
        fx -= (float)s0;
        if (s0 < 0) { s0 = 0; fx = 0.f; }
        if (s0 >= sn - 1) { s0 = sn - 1; fx = 0.f; }
        tab[d * 3] = s0;
        tab[d * 3 + 1] = (int)lrintf((1.f - fx) * 2048.f);
        tab[d * 3 + 2] = (int)lrintf(fx * 2048.f);
    }
}
}  // namespace

static void collect_worker(mi355_gmc* g);
extern "C" int mi355_gmc_track_begin(mi355_gmc* g, const uint8_t* bgr, int height, int width, int downscale) {
    if (!g || !bgr || height <= 0 || width <= 0 || downscale < 1 || g->track_pending) return -1;
    const int oh = downscale > 1 ? height / downscale : height, ow = downscale > 1 ? width / downscale : width;
    if (oh <= 0 || ow <= 0) return -1;
    g->downscale = downscale;
    if (downscale > 1 && !(g->tkey[0] == height && g->tkey[1] == width && g->tkey[2] == oh && g->tkey[3] == ow)) {
        linear_table(ow, width, g->xt); linear_table(oh, height, g->yt);
        g->tkey[0] = height; g->tkey[1] = width; g->tkey[2] = oh; g->tkey[3] = ow;
    }
    // the previous frame's corners are tracked only into a plane of the same size (GMC.apply resets otherwise)
    const bool lk = g->have_prev_pts && g->prev_h == oh && g->prev_w == ow && !g->prev_pts.empty();
    g->lk_pts.clear();
    if (lk) g->lk_pts = g->prev_pts;
    const int n = (int)(g->lk_pts.size() / 2);
    if (g->host) {
        g->host_frame.assign(bgr, bgr + (size_t)height * width * 3); g->hf_h = height; g->hf_w = width;
    } else {
        if (!lk) g->have_prev = false;                          // the device pyramid of another plane size is not a predecessor
        const int rc = mi355_gmc_step_begin(g, bgr, height, width, oh, ow, downscale > 1 ? g->xt.data() : nullptr, downscale > 1 ? g->yt.data() : nullptr,
                                            kQuality, n ? g->lk_pts.data() : nullptr, n, kLkWin, kLkLevels, kLkIters, kLkEps, kLkMinEig);
        if (rc) return rc;
    }
    g->track_pending = true; g->t_oh = oh; g->t_ow = ow; g->t_n = n;
    static const bool async_collect = !getenv("MI355_GMC_ASYNC") || atoi(getenv("MI355_GMC_ASYNC")) != 0;
    if (!g->host && async_collect) {
        if (!g->worker.joinable()) g->worker = std::thread(collect_worker, g);
        { std::lock_guard<std::mutex> lk(g->mu); g->job_ready = true; g->job_done = false; }
        g->job_active = true;
        g->cv.notify_all();
    }
    return 0;
}

static int track_collect(mi355_gmc* g, double* H_out);
static void collect_worker(mi355_gmc* g) {
    (void)hipSetDevice(g->device);
    std::unique_lock<std::mutex> lk(g->mu);
    for (;;) {
        g->cv.wait(lk, [&] { return g->job_ready || g->worker_stop; });
        if (g->worker_stop) return;
        g->job_ready = false;
        lk.unlock();
        const int rc = track_collect(g, g->job_H);
        lk.lock();
        g->job_rc = rc; g->job_done = true;
        g->cv.notify_all();
    }
}

extern "C" int mi355_gmc_track_finish(mi355_gmc* g, double* H_out) {
    if (!g || !H_out || !g->track_pending) return -1;
    if (g->job_active) {                                        // this step's collect waits, runs or ran on the worker
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return g->job_done; });
        g->job_done = false; g->job_active = false; g->track_pending = false;
        std::memcpy(H_out, g->job_H, sizeof(g->job_H));
        return g->job_rc;
    }
    g->track_pending = false;
    return track_collect(g, H_out);
}

static int track_collect(mi355_gmc* g, double* H_out) {
    const int oh = g->t_oh, ow = g->t_ow, n = g->t_n;
    const size_t np = (size_t)oh * ow;
    g->cur_gray.resize(np); g->eig.resize(np); g->ok.resize(np);
    g->next_pts.assign((size_t)n * 2, 0.f); g->status.assign((size_t)n, 0);
    if (g->host) {
        int rc = mi355_gmc_prepare_host(g->host_frame.data(), g->hf_h, g->hf_w, oh, ow, g->downscale > 1 ? g->xt.data() : nullptr,
                                        g->downscale > 1 ? g->yt.data() : nullptr, kQuality, g->cur_gray.data(), g->eig.data(), g->ok.data());
        if (rc) return rc;
        if (n > 0) {
            rc = mi355_gmc_pyr_lk(g->prev_gray.data(), g->cur_gray.data(), oh, ow, g->lk_pts.data(), n, kLkWin, kLkLevels, kLkIters, kLkEps, kLkMinEig,
                                  g->next_pts.data(), g->status.data());
            if (rc) return rc;
        }
    } else {
        const int rc = mi355_gmc_step_finish(g, g->cur_gray.data(), g->eig.data(), g->ok.data(), n ? g->next_pts.data() : nullptr, n ? g->status.data() : nullptr);
        if (rc) { g->have_prev_pts = false; return rc; }
    }
    double H[6] = {1, 0, 0, 0, 1, 0};
    if (n > 0) {
        std::vector<double> src, dst;
        for (int i = 0; i < n; ++i)
            if (g->status[i]) {
                src.push_back(g->lk_pts[2 * i]); src.push_back(g->lk_pts[2 * i + 1]);
                dst.push_back(g->next_pts[2 * i]); dst.push_back(g->next_pts[2 * i + 1]);
            }
        const int m = (int)(src.size() / 2);
        if (m > 4) {
            double E[6];
            if (mi355_gmc_affine_partial(src.data(), dst.data(), m, kRansacThr, kRansacConf, kRansacIters, 0ull, E, nullptr) == 1) {
                std::memcpy(H, E, sizeof(H));
                H[2] *= g->downscale; H[5] *= g->downscale;
            }
        }
    }
    // this frame becomes the previous one: its plane and its corners, strongest first
    g->prev_pts.resize((size_t)kMaxCorners * 2);
    const int nc = mi355_gmc_order_corners(g->eig.data(), g->ok.data(), oh, ow, kMaxCorners, g->prev_pts.data());
    g->prev_pts.resize((size_t)std::max(nc, 0) * 2);
    g->prev_gray.swap(g->cur_gray);
    g->prev_h = oh; g->prev_w = ow; g->have_prev_pts = true;
    std::memcpy(H_out, H, sizeof(H));
    return 0;
}

// n consecutive frames of one video in ONE call (a batched sweep holds the frames of a detector batch before the tracker needs their
// warps): the frame preparation of all n frames as one set of launches, the corner ordering of the n planes on host threads, the
// Lucas-Kanade tracking of all n frame pairs as ONE launch (n x <= 1000 wavefronts instead of n dependent launches of <= 1000), RANSAC
// per pair on host threads.  Same kernels, same arithmetic and the same state transitions as n track_begin / track_finish steps --
// H_out [n][6] equals theirs bit for bit -- at a fraction of the latency: a step's kernels are latency-bound (148 us for 1000 corners
// whatever the chip could do beside them).  Continues from / leaves behind the object's previous frame.  frames: n pointers to BGR
// frames of height x width.  Host objects run the n steps one after the other.
extern "C" int mi355_gmc_track_batch(mi355_gmc* g, const uint8_t* const* frames, int n, int height, int width, int downscale, double* H_out) {
    if (!g || !frames || n <= 0 || height <= 0 || width <= 0 || downscale < 1 || !H_out || g->track_pending) return -1;
    for (int f = 0; f < n; ++f) if (!frames[f]) return -1;
    if (g->host) {
        for (int f = 0; f < n; ++f) {
            int rc = mi355_gmc_track_begin(g, frames[f], height, width, downscale); if (rc) return rc;
            rc = mi355_gmc_track_finish(g, H_out + 6 * f); if (rc) return rc;
        }
        return 0;
    }
    const int oh = downscale > 1 ? height / downscale : height, ow = downscale > 1 ? width / downscale : width;
    if (oh <= 0 || ow <= 0) return -1;
    GCHK(hipSetDevice(g->device));
    g->downscale = downscale;
    const int resize = downscale > 1;
    if (resize && !(g->tkey[0] == height && g->tkey[1] == width && g->tkey[2] == oh && g->tkey[3] == ow)) {
        linear_table(ow, width, g->xt); linear_table(oh, height, g->yt);
        g->tkey[0] = height; g->tkey[1] = width; g->tkey[2] = oh; g->tkey[3] = ow;
    }
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nb = (size_t)height * width * 3, np = (size_t)oh * ow;
    int hs[kMaxLevels], ws[kMaxLevels], levels = 1;
    hs[0] = oh; ws[0] = ow;
    for (int l = 0; l < kLkLevels && levels < kMaxLevels; ++l) {
        const int nh = (hs[levels - 1] + 1) / 2, nw = (ws[levels - 1] + 1) / 2;
        if (nh <= kLkWin || nw <= kLkWin) break;
        hs[levels] = nh; ws[levels] = nw; ++levels;
    }
    size_t off[kMaxLevels], pyr_bytes = 0;
    for (int l = 0; l < levels; ++l) { off[l] = pyr_bytes; pyr_bytes += al((size_t)hs[l] * ws[l]); }
    // does the object's previous frame precede frames[0]?  (same plane size, corners known, its pyramid on the device)
    const bool cont = g->have_prev_pts && g->prev_h == oh && g->prev_w == ow && g->have_prev && g->ph == oh && g->pw == ow && g->pyr_cap >= pyr_bytes;
    // device: [frames n x nb | pyramids (n + 1) x pyr_bytes | eig n x np x 4 | ok n x np | x table | y table | max n x 4 | pts n x kMaxCorners x 8 |
    //          next (same) | status n x kMaxCorners | counts n x 4]
    const size_t fstride = (nb % 16 == 0) ? nb : al(nb);       // dense frames when 16-byte aligned: the detector may read them in place (mi355_gmc_batch_frames)
    const size_t o_pyr = al((size_t)n * fstride), o_eig = o_pyr + (size_t)(n + 1) * pyr_bytes, o_ok = o_eig + al((size_t)n * np * 4), o_xt = o_ok + al((size_t)n * np),
                 o_yt = o_xt + al((size_t)ow * 12), o_max = o_yt + al((size_t)oh * 12), o_pts = o_max + al((size_t)n * 4),
                 o_next = o_pts + al((size_t)n * kMaxCorners * 8), o_st = o_next + al((size_t)n * kMaxCorners * 8), o_cnt = o_st + al((size_t)n * kMaxCorners),
                 d_total = o_cnt + al((size_t)n * 4);
    if (g->batch_cap < d_total) {
        if (g->d_batch) (void)hipFree(g->d_batch);
        g->d_batch = nullptr; g->batch_cap = 0;
        GCHK(hipMalloc(&g->d_batch, d_total)); g->batch_cap = d_total;
    }
    // pinned: [frames | tables | eig | ok | pts | next | status | counts | last gray]
    const size_t p_xt = al((size_t)n * fstride), p_yt = p_xt + al((size_t)ow * 12), p_eig = p_yt + al((size_t)oh * 12), p_ok = p_eig + al((size_t)n * np * 4),
                 p_pts = p_ok + al((size_t)n * np), p_next = p_pts + al((size_t)n * kMaxCorners * 8), p_st = p_next + al((size_t)n * kMaxCorners * 8),
                 p_cnt = p_st + al((size_t)n * kMaxCorners), p_gray = p_cnt + al((size_t)n * 4), h_total = p_gray + al(np);
    if (g->hbatch_cap < h_total) {
        if (g->h_batch) (void)hipHostFree(g->h_batch);
        g->h_batch = nullptr; g->hbatch_cap = 0;
        GCHK(hipHostMalloc(&g->h_batch, h_total)); g->hbatch_cap = h_total;
    }
    uint8_t* D = g->d_batch; uint8_t* P = g->h_batch;
    {   // frames -> pinned staging on a few threads (64 frames of 320 x 240 are 14.7 MB: 2.5 ms on one core, a third of this call)
        const int nt = std::max(1, std::min(std::min(8, n), (int)std::thread::hardware_concurrency()));
        if (nt <= 1 || (size_t)n * nb < (1u << 20)) { for (int f = 0; f < n; ++f) std::memcpy(P + (size_t)f * fstride, frames[f], nb); }
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (int f = t; f < n; f += nt) std::memcpy(P + (size_t)f * fstride, frames[f], nb); });
            for (auto& t : th) t.join();
        }
    }
    GCHK(hipMemcpyAsync(D, P, (size_t)n * fstride, hipMemcpyHostToDevice, g->stream));
    if (!g->ev_batch_up) GCHK(hipEventCreateWithFlags(&g->ev_batch_up, hipEventDisableTiming));
    GCHK(hipEventRecord(g->ev_batch_up, g->stream));
    { std::lock_guard<std::mutex> lk(g->mu); g->batch_n = n; g->batch_h = height; g->batch_w = width; g->batch_fstride = fstride; ++g->batch_up_seq; }
    g->cv.notify_all();
    if (resize) {
        std::memcpy(P + p_xt, g->xt.data(), (size_t)ow * 12); std::memcpy(P + p_yt, g->yt.data(), (size_t)oh * 12);
        GCHK(hipMemcpyAsync(D + o_xt, P + p_xt, (size_t)ow * 12, hipMemcpyHostToDevice, g->stream));
        GCHK(hipMemcpyAsync(D + o_yt, P + p_yt, (size_t)oh * 12, hipMemcpyHostToDevice, g->stream));
    }
    GCHK(hipMemsetAsync(D + o_max, 0, (size_t)n * 4, g->stream));
    if (cont) GCHK(hipMemcpyAsync(D + o_pyr, g->d_pyr[g->slot], pyr_bytes, hipMemcpyDeviceToDevice, g->stream));   // slot 0 = the previous frame's pyramid
    uint8_t* pyr1 = D + o_pyr + pyr_bytes;                     // frame f's pyramid at pyr1 + f * pyr_bytes
    const unsigned blocks = (unsigned)((np + 255) / 256);
    hipLaunchKernelGGL(gray_resize_kernel, dim3(blocks, n), dim3(256), 0, g->stream, D, height, width, (const int*)(D + o_xt), (const int*)(D + o_yt),
                       pyr1 + off[0], oh, ow, resize, fstride, pyr_bytes);
    hipLaunchKernelGGL(min_eig_kernel, dim3(blocks, n), dim3(256), 0, g->stream, pyr1 + off[0], oh, ow, (float*)(D + o_eig), (unsigned*)(D + o_max), pyr_bytes);
    hipLaunchKernelGGL(corner_mask_kernel, dim3(blocks, n), dim3(256), 0, g->stream, (const float*)(D + o_eig), oh, ow, (const unsigned*)(D + o_max), kQuality,
                       D + o_ok);
    for (int l = 1; l < levels; ++l) {
        const int npx = hs[l] * ws[l];
        hipLaunchKernelGGL(pyr_down_kernel, dim3((npx + 255) / 256, n), dim3(256), 0, g->stream, pyr1 + off[l - 1], hs[l - 1], ws[l - 1], pyr1 + off[l], hs[l],
                           ws[l], pyr_bytes);
    }
    GCHK(hipGetLastError());
    GCHK(hipMemcpyAsync(P + p_eig, D + o_eig, (size_t)n * np * 4, hipMemcpyDeviceToHost, g->stream));
    GCHK(hipMemcpyAsync(P + p_ok, D + o_ok, (size_t)n * np, hipMemcpyDeviceToHost, g->stream));
    GCHK(hipMemcpyAsync(P + p_gray, pyr1 + (size_t)(n - 1) * pyr_bytes + off[0], np, hipMemcpyDeviceToHost, g->stream));
    if (hipStreamSynchronize(g->stream) != hipSuccess) { (void)hipGetLastError(); g->have_prev = false; g->have_prev_pts = false; return -2; }
    // corners of every frame, strongest first (host threads); pair f tracks the corners of frame f - 1 (f = 0: the object's previous frame)
    std::vector<std::vector<float>> corners(n);
    std::vector<int> ncorn(n, 0);
    const int nthreads = std::max(1, std::min(std::min(8, n), (int)std::thread::hardware_concurrency()));
    auto parallel = [&](auto&& body) {
        if (nthreads <= 1) { for (int f = 0; f < n; ++f) body(f); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t] { for (int f = t; f < n; f += nthreads) body(f); });
        for (auto& t : th) t.join();
    };
    parallel([&](int f) {
        corners[f].resize((size_t)kMaxCorners * 2);
        ncorn[f] = std::max(0, mi355_gmc_order_corners((const float*)(P + p_eig) + (size_t)f * np, P + p_ok + (size_t)f * np, oh, ow, kMaxCorners, corners[f].data()));
        corners[f].resize((size_t)ncorn[f] * 2);
    });
    int* cnt = (int*)(P + p_cnt);
    float* hp = (float*)(P + p_pts);
    int any = 0;
    for (int f = 0; f < n; ++f) {
        const std::vector<float>* src = f == 0 ? (cont && !g->prev_pts.empty() ? &g->prev_pts : nullptr) : &corners[f - 1];
        cnt[f] = src ? (int)(src->size() / 2) : 0;
        if (cnt[f]) std::memcpy(hp + (size_t)f * kMaxCorners * 2, src->data(), src->size() * sizeof(float));
        any |= cnt[f];
    }
    if (any) {
        GCHK(hipMemcpyAsync(D + o_pts, P + p_pts, (size_t)n * kMaxCorners * 8, hipMemcpyHostToDevice, g->stream));
        GCHK(hipMemcpyAsync(D + o_cnt, P + p_cnt, (size_t)n * 4, hipMemcpyHostToDevice, g->stream));
        LkArgs a{};
        for (int l = 0; l < levels; ++l) { a.prev[l] = D + o_pyr + off[l]; a.cur[l] = pyr1 + off[l]; a.h[l] = hs[l]; a.w[l] = ws[l]; }
        a.top = levels - 1; a.n = 0; a.win = kLkWin; a.max_iters = kLkIters; a.width = ow; a.height = oh;
        a.eps2 = kLkEps * kLkEps; a.min_eig = kLkMinEig;
        a.pts = (const float*)(D + o_pts); a.next = (float*)(D + o_next); a.status = D + o_st;
        a.pair_stride = pyr_bytes; a.max_pts = kMaxCorners; a.n_arr = (const int*)(D + o_cnt);
        hipLaunchKernelGGL(lk_kernel, dim3((kMaxCorners + kLkWaves - 1) / kLkWaves, n), dim3(64 * kLkWaves), 0, g->stream, a);
        GCHK(hipGetLastError());
        GCHK(hipMemcpyAsync(P + p_next, D + o_next, (size_t)n * kMaxCorners * 8, hipMemcpyDeviceToHost, g->stream));
        GCHK(hipMemcpyAsync(P + p_st, D + o_st, (size_t)n * kMaxCorners, hipMemcpyDeviceToHost, g->stream));
    }
    // the last frame's pyramid becomes the object's previous one (the per-frame entry points continue from it)
    if (g->pyr_cap < pyr_bytes) {
        for (int i = 0; i < 2; ++i) { if (g->d_pyr[i]) (void)hipFree(g->d_pyr[i]); g->d_pyr[i] = nullptr; }
        g->pyr_cap = 0;
        GCHK(hipMalloc(&g->d_pyr[0], pyr_bytes)); GCHK(hipMalloc(&g->d_pyr[1], pyr_bytes)); g->pyr_cap = pyr_bytes;
    }
    GCHK(hipMemcpyAsync(g->d_pyr[g->slot], pyr1 + (size_t)(n - 1) * pyr_bytes, pyr_bytes, hipMemcpyDeviceToDevice, g->stream));
    if (hipStreamSynchronize(g->stream) != hipSuccess) { (void)hipGetLastError(); g->have_prev = false; g->have_prev_pts = false; return -2; }
    g->have_prev = true; g->ph = oh; g->pw = ow;
    const float* hn = (const float*)(P + p_next);
    const uint8_t* hst = P + p_st;
    parallel([&](int f) {
        double* H = H_out + 6 * f;
        H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0;
        const int m0 = cnt[f];
        if (!m0) return;
        std::vector<double> src, dst;
        const float* p0 = hp + (size_t)f * kMaxCorners * 2; const float* p1 = hn + (size_t)f * kMaxCorners * 2; const uint8_t* st = hst + (size_t)f * kMaxCorners;
        for (int i = 0; i < m0; ++i)
            if (st[i]) { src.push_back(p0[2 * i]); src.push_back(p0[2 * i + 1]); dst.push_back(p1[2 * i]); dst.push_back(p1[2 * i + 1]); }
        const int m = (int)(src.size() / 2);
        if (m > 4) {
            double E[6];
            if (mi355_gmc_affine_partial(src.data(), dst.data(), m, kRansacThr, kRansacConf, kRansacIters, 0ull, E, nullptr) == 1) {
                std::memcpy(H, E, sizeof(E));
                H[2] *= downscale; H[5] *= downscale;
            }
        }
    });
    g->prev_pts = corners[n - 1];
    g->prev_gray.assign(P + p_gray, P + p_gray + np);
    g->prev_h = oh; g->prev_w = ow; g->have_prev_pts = true;
    return 0;
}

// The frames of the batch that a mi355_gmc_track_batch call -- running on ANOTHER thread, or already returned -- has uploaded: blocks until that
// call has issued its upload (at most timeout_ms) and the upload has landed, then hands out the device pointer (frame f at dev + f * stride).
// The detector pass of the same batch reads them in place (mi355_yolo_infer_device when stride == height * width * 3): one staging copy and one
// upload per batch instead of two of each.  Each upload is handed out once; valid until the next mi355_gmc_track_batch call on the object.
// after_seq: the value of mi355_gmc_batch_seq taken BEFORE that call was started (uploads are numbered; an older one is never handed out).
// Returns 0, 1 on timeout / nothing new, -1 on bad arguments, -2 on a HIP error.
extern "C" int mi355_gmc_batch_frames(mi355_gmc* g, unsigned long long after_seq, int timeout_ms, const uint8_t** dev, int* n, int* height, int* width,
                                      long long* stride) {
    if (!g || g->host || !dev || !n || !height || !width || !stride) return -1;
    {
        std::unique_lock<std::mutex> lk(g->mu);
        if (!g->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms), [&] { return g->batch_up_seq > after_seq; })) return 1;
        *n = g->batch_n; *height = g->batch_h; *width = g->batch_w; *stride = (long long)g->batch_fstride;
    }
    GCHK(hipSetDevice(g->device));
    GCHK(hipEventSynchronize(g->ev_batch_up));
    *dev = g->d_batch;
    return 0;
}

extern "C" unsigned long long mi355_gmc_batch_seq(mi355_gmc* g) {
    if (!g) return 0;
    std::lock_guard<std::mutex> lk(g->mu);
    return g->batch_up_seq;
}

// Forget the previous frame (GMC.reset_params); a pending step is collected and dropped.
extern "C" int mi355_gmc_track_reset(mi355_gmc* g) {
    if (!g) return -1;
    if (g->track_pending) { double H[6]; (void)mi355_gmc_track_finish(g, H); }
    g->have_prev_pts = false; g->prev_pts.clear(); g->prev_gray.clear(); g->prev_h = g->prev_w = 0;
    if (!g->host) g->have_prev = false;
    return 0;
}

// The previous frame as the object holds it (tests): plane size, number of corners; gray_out [oh * ow] and pts_out [pts_cap][2] when given.
extern "C" int mi355_gmc_track_state(const mi355_gmc* g, int* oh, int* ow, int* n_pts, uint8_t* gray_out, float* pts_out, int pts_cap) {
    if (!g) return -1;
    const int n = g->have_prev_pts ? (int)(g->prev_pts.size() / 2) : 0;
    if (oh) *oh = g->have_prev_pts ? g->prev_h : 0;
    if (ow) *ow = g->have_prev_pts ? g->prev_w : 0;
    if (n_pts) *n_pts = n;
    if (gray_out && g->have_prev_pts) std::memcpy(gray_out, g->prev_gray.data(), g->prev_gray.size());
    if (pts_out && pts_cap > 0 && n > 0) std::memcpy(pts_out, g->prev_pts.data(), (size_t)std::min(n, pts_cap) * 8);
    return 0;
}
