/*
 * det_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * A plain-C restatement of the floating-point kernels of the per-frame YOLO path that
 * /root/reference/model.py:38 reaches inside ultralytics==8.3.225 (conv + folded BN + SiLU, the u8 stem,
 * the Detect/Pose decode), with ONE property torch's CPU kernels do not have: the order of every fp32
 * operation is fixed and written down here.  torch/oneDNN pick a summation order per host and thread count
 * (measured: the same torch oracle differs from a float64 run by up to 6e-3 px on one host and 1.5e-2 px on
 * another, SURVEY 8(c) "parity unpinned"), which makes "identical NMS indices" untestable against it.  This
 * file is the algorithm with a canonical order, so an implementation can be compared with it BIT FOR BIT.
 *
 * Canonical arithmetic (also stated in DESIGN.md):
 *   conv      tot = +0; for cb in 16-channel blocks: { p = +0; for tap (kh-major): for s in 0..3: for g in 0..3:
 *                 ci = 16*cb + 4*g + s;  p = fmaf(w[co][ci][tap], x[pixel@tap][ci], p);   tot = tot + p; }
 *             (taps outside the image and channels >= Cin contribute nothing): a two-level ("blocked") summation -- every
 *             16-channel block is one fma chain from +0, the block partials are added in block order.  y = tot + bias;
 *             y = silu(y) if act; y = y + residual if given.  (Round 1 used ONE chain over all of K; the blocked order is
 *             1.4x closer to a float64 evaluation -- as close as torch/oneDNN's SIMD-blocked sums -- and lets an
 *             implementation split K over workers and still reproduce the bits.)
 *   stem      acc = +0; for kh: for kw: for byte channel (B, G, R): acc = fmaf(lut[byte], w[co][2-ch][kh][kw], acc)
 *             with lut[i] = (float)i / 255.0f; y = silu(acc + bias).
 *   exp       det_expf below (Cody-Waite reduction + degree-5 polynomial, fmaf only, no libm);
 *             silu(v) = v / (1 + det_expf_silu(-v)) (range-restricted exp, same bits on [-87.25, 87]; IEEE division); sigmoid(v) = 1 / (1 + det_expf(-v)).
 *   decode    as ultralytics head.py / tal.py, evaluated left to right without contraction (see det_decode).
 * Compile with -ffp-contract=off -mfma: every fused operation is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float int_as_float(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

float det_expf(float x) {
    x = fminf(fmaxf(x, -104.0f), 89.0f);
    const float t = fmaf(x, 1.44269504088896341f, 12582912.0f);   /* 1.5 * 2^23: round(x * log2 e), ties to even */
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693145751953125f, x);                    /* ln 2, high part */
    r = fmaf(n, -1.428606765330187045e-06f, r);                   /* ln 2, low part */
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = fmaf(p, r2, r);
    e = e + 1.0f;
    const int ni = (int)n;
    const int n1 = ni / 2, n2 = ni - n1;
    const float s1 = int_as_float((n1 + 127) << 23), s2 = int_as_float((n2 + 127) << 23);
    return (e * s1) * s2;
}

/* exp for SiLU only (csrc/detmath.h:det_expf_silu): argument clamped to [-87.25, 87] where 2^n is a normal float (one exact
 * scaling) and 1 + e < 2^126 (its reciprocal is a normal float: the GPU's division sequence needs no range scaling there and
 * returns the IEEE quotient written below); identical bits to det_expf inside that range. */
static inline float det_expf_silu(float x) {
    x = fminf(fmaxf(x, -87.25f), 87.0f);
    const float t = fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045e-06f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = fmaf(p, r2, r);
    e = e + 1.0f;
    return e * int_as_float(((int32_t)n + 127) << 23);
}

static inline float det_silu(float v) { return v / (1.0f + det_expf_silu(-v)); }
static inline float det_sigmoid(float v) { return 1.0f / (1.0f + det_expf(-v)); }

void det_expf_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = det_expf(x[i]); }
void det_silu_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = det_silu(x[i]); }

/* x [n][h][w][cin] NHWC, wt OIHW [cout][cin][k][k], bias [cout], res/y [n][ho][wo][cout]; ho = h/stride. */
void det_conv2d(const float* x, int n, int h, int w, int cin, const float* wt, const float* bias, int cout, int k,
                int stride, int pad, int act, const float* res, float* y) {
    const int ho = h / stride, wo = w / stride, taps = k * k;
    const int cib = (cin + 15) / 16;
    /* the canonical k sequence: (cb, tap, s, g) -> (ci, tap); weights re-laid as [seq][cout]; blk_end[cb] = end of block cb */
    const int nseq_max = cib * taps * 16;
    int* seq_ci = (int*)malloc(sizeof(int) * nseq_max);
    int* seq_tap = (int*)malloc(sizeof(int) * nseq_max);
    int* blk_end = (int*)malloc(sizeof(int) * cib);
    int nseq = 0;
    for (int cb = 0; cb < cib; blk_end[cb] = nseq, ++cb)
        for (int tap = 0; tap < taps; ++tap)
            for (int s = 0; s < 4; ++s)
                for (int g = 0; g < 4; ++g) {
                    const int ci = 16 * cb + 4 * g + s;
                    if (ci < cin) { seq_ci[nseq] = ci; seq_tap[nseq] = tap; ++nseq; }
                }
    float* wseq = (float*)malloc(sizeof(float) * (size_t)nseq * cout);
    for (int q = 0; q < nseq; ++q)
        for (int co = 0; co < cout; ++co)
            wseq[(size_t)q * cout + co] = wt[((size_t)co * cin + seq_ci[q]) * taps + seq_tap[q]];
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * cout);
        float* tot = (float*)malloc(sizeof(float) * cout);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < n; ++b)
            for (int oy = 0; oy < ho; ++oy)
                for (int ox = 0; ox < wo; ++ox) {
                    for (int co = 0; co < cout; ++co) tot[co] = 0.0f;
                    for (int cb = 0, q = 0; cb < cib; ++cb) {
                        for (int co = 0; co < cout; ++co) acc[co] = 0.0f;
                        for (; q < blk_end[cb]; ++q) {
                            const int tap = seq_tap[q];
                            const int iy = oy * stride - pad + tap / k, ix = ox * stride - pad + tap % k;
                            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                            const float xv = x[(((size_t)b * h + iy) * w + ix) * cin + seq_ci[q]];
                            const float* wr = wseq + (size_t)q * cout;
                            for (int co = 0; co < cout; ++co) acc[co] = fmaf(wr[co], xv, acc[co]);
                        }
                        for (int co = 0; co < cout; ++co) tot[co] = tot[co] + acc[co];
                    }
                    const size_t po = (((size_t)b * ho + oy) * wo + ox) * cout;
                    for (int co = 0; co < cout; ++co) {
                        float v = tot[co] + bias[co];
                        if (act) v = det_silu(v);
                        if (res) v = v + res[po + co];
                        y[po + co] = v;
                    }
                }
        free(acc); free(tot);
    }
    free(wseq); free(seq_ci); free(seq_tap); free(blk_end);
}

/* bgr [n][h][w][3] uint8 (letterboxed), wt OIHW [cout][3][k][k] over RGB model channels -> y [n][h/s][w/s][cout] */
void det_stem(const uint8_t* bgr, int n, int h, int w, const float* wt, const float* bias, int cout, int k, int stride,
              int pad, float* y) {
    const int ho = h / stride, wo = w / stride;
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (float)i / 255.0f;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int oy = 0; oy < ho; ++oy)
            for (int ox = 0; ox < wo; ++ox)
                for (int co = 0; co < cout; ++co) {
                    float acc = 0.0f;
                    for (int kh = 0; kh < k; ++kh) {
                        const int iy = oy * stride - pad + kh;
                        if (iy < 0 || iy >= h) continue;
                        for (int kw = 0; kw < k; ++kw) {
                            const int ix = ox * stride - pad + kw;
                            if (ix < 0 || ix >= w) continue;
                            const uint8_t* px = bgr + (((size_t)b * h + iy) * w + ix) * 3;
                            for (int cb = 0; cb < 3; ++cb)
                                acc = fmaf(lut[px[cb]], wt[(((size_t)co * 3 + (2 - cb)) * k + kh) * k + kw], acc);
                        }
                    }
                    y[(((size_t)b * ho + oy) * wo + ox) * cout + co] = det_silu(acc + bias[co]);
                }
}

/* One level of Detect/Pose decode.  box [n][h][w][64], cls [n][h][w][nc], kpt [n][h][w][nkpt*kdim] (or NULL), all NHWC.
 * out: pred [n][no][A_total] (Ultralytics layout), this level's anchors at [anchor0, anchor0 + h*w). */
void det_decode_level(const float* box, const float* cls, const float* kpt, int n, int h, int w, int nc, int nkpt,
                      int kdim, int stride, int anchor0, int a_total, float* pred) {
    const int nk = nkpt * kdim, no = 4 + nc + nk;
    const float st = (float)stride;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t pix = ((size_t)b * h + y) * w + x;
                const int a = anchor0 + y * w + x;
                float* o = pred + (size_t)b * no * a_total + a;        /* o[c * a_total] */
                const float ax = (float)x + 0.5f, ay = (float)y + 0.5f;
                float dist[4];
                for (int s = 0; s < 4; ++s) {
                    const float* v = box + pix * 64 + 16 * s;
                    float m = v[0];
                    for (int j = 1; j < 16; ++j) m = fmaxf(m, v[j]);
                    float e[16], sum = 0.0f;
                    for (int j = 0; j < 16; ++j) { e[j] = det_expf(v[j] - m); sum += e[j]; }
                    float d = 0.0f;
                    for (int j = 0; j < 16; ++j) d += (e[j] / sum) * (float)j;
                    dist[s] = d;
                }
                const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
                o[0 * (size_t)a_total] = ((x1 + x2) / 2.0f) * st;
                o[1 * (size_t)a_total] = ((y1 + y2) / 2.0f) * st;
                o[2 * (size_t)a_total] = (x2 - x1) * st;
                o[3 * (size_t)a_total] = (y2 - y1) * st;
                for (int c = 0; c < nc; ++c) o[(size_t)(4 + c) * a_total] = det_sigmoid(cls[pix * nc + c]);
                for (int q = 0; q < nkpt; ++q) {
                    const float* kp = kpt + pix * nk + q * kdim;
                    float* ko = o + (size_t)(4 + nc + q * kdim) * a_total;
                    ko[0] = (kp[0] * 2.0f + (ax - 0.5f)) * st;
                    ko[(size_t)a_total] = (kp[1] * 2.0f + (ay - 0.5f)) * st;
                    if (kdim == 3) ko[2 * (size_t)a_total] = det_sigmoid(kp[2]);
                }
            }
}
