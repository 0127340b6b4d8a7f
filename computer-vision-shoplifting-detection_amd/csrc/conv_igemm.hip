// Fused conv (1x1 / 3x3, stride 1 / 2) + bias + SiLU (+ residual) as an implicit GEMM on the CDNA4 fp32
// matrix pipe (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered fma chain, 64 FLOP/clk/SIMD).
//
// Replaces ultralytics/nn/modules/conv.py:Conv.forward_fuse (act(conv(x)) with the BN folded) and the
// residual add of block.py:Bottleneck.forward, reached from /root/reference/model.py:38.
//
// GEMM view:  D[cout][pixel] = sum_{tap, ci} W[cout][tap][ci] * X[pixel @ tap][ci]
//   MFMA A operand = weights  (row = cout,  k = input channel)   -> read from HBM/L2 in pre-packed
//                                                                   fragment order (1 KiB per wave load)
//   MFMA B operand = pixels   (col = pixel, k = input channel)   -> read from an LDS-staged NHWC halo tile
//   so the accumulator of a lane holds 4 CONSECUTIVE couts of ONE pixel: the epilogue is one 16-byte store.
// The k index inside a 16-channel block is permuted (MFMA step s covers channels {4g+s}) so that a lane's
// four k-steps are one aligned float4 in both operands (ds_read_b128 / global_load_dwordx4).
//
// Block = 256 threads = 4 waves, arranged WP (along pixels) x WC (along couts); a wave owns PT pixel tiles
// x CT cout tiles of 16x16.  The input tile (with halo) is staged through LDS in chunks of `ck` channels.
#include "common.h"
#include "detmath.h"
#include <algorithm>
#include <cstdlib>
#include <vector>
#include <type_traits>
#ifndef MI355_CONV_WAVES
#define MI355_CONV_WAVES 3   // min waves per SIMD the register allocator must leave room for (4 blocks of 256 threads / CU = 4)
#endif

#pragma clang fp contract(off)

namespace mi355 {

typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ float silu_f(float v) { return det_silu(v); }

// bias + SiLU (+ residual) and the 16-byte stores: a lane holds 4 consecutive couts of one pixel per tile.
// Two passes: all the ALU work first (16 independent SiLU chains per lane interleave freely), then the stores back to
// back.  t_mid (diagnostics) receives the time between the passes.
template <int STRIDE, int PT, int CT, int WP>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, f32x4 (&acc)[CT][PT], const f32x4 (&bias4)[CT], int lane, int wp,
                                              int ct0, int b, int oy0, int ox0, int npix, unsigned long long* t_mid = nullptr) {
    if (a.act) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                f32x4 v = acc[ct][pt] + bias4[ct];
                v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
                acc[ct][pt] = v;
            }
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = acc[ct][pt] + bias4[ct];
    }
    if (t_mid) *t_mid = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wp * PT + pt) * 16 + (lane & 15);
        const int pp = p < npix ? p : 0;
        const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
        const int lx = pp - ly * a.TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        const bool ok = (p < npix) && (oy < a.Hout) && (ox < a.Wout);
        const size_t po = ((size_t)b * a.Hout + oy) * a.Wout + ox;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = (ct0 + ct) * 16 + (lane >> 4) * 4;
            if (!ok || c >= a.Cout) continue;
            f32x4 v = acc[ct][pt];
            float* d = a.dst + po * a.dst_cs + c;
            if (c + 3 < a.Cout) {
                if (a.res) v += *(const f32x4*)(a.res + po * a.res_cs + c);
                *(f32x4*)d = v;
            } else {
                for (int i = 0; i < 4 && c + i < a.Cout; ++i) {
                    float r = v[i];
                    if (a.res) r += a.res[po * a.res_cs + c + i];
                    d[i] = r;
                }
            }
        }
    }
}

template <int KS, int STRIDE, int PT, int CT, int WP>
__global__ __launch_bounds__(256) void conv_igemm_f32(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WC = 4 / WP;
    constexpr int TAPS = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
    const int ct0 = (blockIdx.y * WC + wc) * CT;
    const int npix = a.TW * a.TH;
    // diagnostics only (a.debug == nullptr in every product launch): per-wave phase stamps
    unsigned long long t_start = 0, t_stage = 0, t_loop = 0, acc_stage = 0, acc_loop = 0, acc_ld = 0;
    unsigned long long r_start = 0;
    if (a.debug) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
    int xoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int p = (wp * PT + pt) * 16 + (lane & 15);
        p = p < npix ? p : 0;
        const int ly = (int)(((float)p + 0.5f) * a.inv_TW);
        const int lx = p - ly * a.TW;
        xoff[pt] = ((ly * STRIDE) * a.TWin + lx * STRIDE) * a.ldp + (lane >> 4) * 4;
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* srcb = a.src + (size_t)b * a.Hin * a.Win * a.src_cs;
    // cout tiles beyond the last one re-read the last tile's weights (their outputs are discarded by the epilogue):
    // no branch around the fragment loads, so hipcc keeps a counted s_waitcnt vmcnt(N) and the prefetch stays in flight
    const float* wbase[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = a.wpk + (size_t)ctile * TAPS * a.cib * 256 + lane * 4;
    }
    // the bias of this lane's 4 couts per cout tile is fetched now (the load's L2 latency hides under the staging) and
    // not in the epilogue, where it would sit on the block's critical path
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        bias4[ct] = *(const f32x4*)(a.bias + ctile * 16 + (lane >> 4) * 4);
    }
    const int ck4m = (a.ck >> 2) - 1;
    const int total_f4 = a.npix_in << a.ck4_shift;

    for (int c0 = 0; c0 < a.Cin; c0 += a.ck) {
        if (c0) __syncthreads();
        if (a.debug) t_stage = __builtin_amdgcn_s_memtime();
        // ---- stage the halo tile, channels [c0, c0+ck), zero-filled outside the image / beyond Cin ----
        // Loads are issued in batches of 8 per thread BEFORE any of them is consumed (out-of-image / beyond-Cin slots
        // read a zero page instead of branching), so one HBM/L2 latency is paid per batch, not per float4.
        for (int base = 0; base < total_f4; base += 8 * 256) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                const int pix = idx >> a.ck4_shift, q = idx & ck4m;
                const int iy = (int)(((float)pix + 0.5f) * a.inv_TWin);
                const int ix = pix - iy * a.TWin;
                const int gy = iy0 + iy, gx = ix0 + ix, c = c0 + 4 * q;
                const bool inb = idx < total_f4 && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win && c < a.cin4;
                const float* g = inb ? srcb + ((size_t)gy * a.Win + gx) * a.src_cs + c : a.zeros;
                v[u] = *(const f32x4*)g;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                if (idx < total_f4) {
                    const int pix = idx >> a.ck4_shift, q = idx & ck4m;
                    *(f32x4*)(lds + pix * a.ldp + 4 * q) = v[u];
                }
            }
        }
        if (a.debug) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); acc_ld += __builtin_amdgcn_s_memtime() - t_stage; }
        __syncthreads();
        if (a.debug) { t_loop = __builtin_amdgcn_s_memtime(); acc_stage += t_loop - t_stage; }
        const int rem = a.Cin - c0;
        const int nkk = ((rem < a.ck ? rem : a.ck) + 15) >> 4;
        const int cib0 = c0 >> 4;
        // canonical accumulation order of one output: 16-channel block (outer), tap, MFMA step s, k-group g.
        // (block, tap) is flattened into one runtime loop of pipeline steps; all offsets advance as wave-uniform
        // scalars.  Weight fragments come from L2 with 1-2k cycles of latency under load, so they are prefetched WD
        // steps ahead through a ring of WD register sets; pixel fragments (LDS, ~100 cycles) one step ahead through two
        // sets.  Both cursors CLAMP at the last step instead of guarding the loads: every load is unconditional, which
        // is what lets hipcc keep counted s_waitcnt vmcnt(N) / lgkmcnt(N) instead of draining the queues.
        constexpr int WD = (CT <= 2) ? 4 : 2;
        const int n_it = nkk * TAPS;
        const int wstep = a.cib * 256;
        int w_it = 0, w_kw = 0, w_kh = 0, w_kk = 0, w_off = cib0 * 256;     // weight cursor: (tap*cib + cib0 + kk)*256 floats
        int x_it = 0, x_kw = 0, x_kh = 0, x_kk = 0, x_off = 0;              // pixel cursor: (kh*TWin + kw)*ldp + kk*16 floats
        f32x4 wf[WD][CT], xf[2][PT];
        auto load_w = [&](f32x4* w) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f32x4*)(wbase[ct] + w_off);
            if (w_it + 1 < n_it) {
                ++w_it; ++w_kw; w_off += wstep;
                if (w_kw == KS) {
                    w_kw = 0; ++w_kh;
                    if (w_kh == KS) { w_kh = 0; ++w_kk; w_off = (cib0 + w_kk) * 256; }
                }
            }
        };
        auto load_x = [&](f32x4* x) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) x[pt] = *(const f32x4*)__builtin_assume_aligned(lds + xoff[pt] + x_off, 16);
            if (x_it + 1 < n_it) {
                ++x_it; ++x_kw; x_off += a.ldp;
                if (x_kw == KS) {
                    x_kw = 0; ++x_kh; x_off += (a.TWin - KS) * a.ldp;
                    if (x_kh == KS) { x_kh = 0; ++x_kk; x_off = x_kk * 16; }
                }
            }
        };
        auto mma = [&](const f32x4* w, const f32x4* x) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][s], x[pt][s], acc[ct][pt], 0, 0, 0);
        };
#pragma unroll
        for (int j = 0; j < WD - 1; ++j) load_w(wf[j]);        // steps 0 .. WD-2
        load_x(xf[0]);                                          // step 0
        for (int it = 0; it < n_it; it += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                load_w(wf[(j + WD - 1) % WD]);                  // step it+j+WD-1 (clamped)
                load_x(xf[(j + 1) & 1]);                        // step it+j+1    (clamped)
                __builtin_amdgcn_sched_barrier(0);
                if (it + j < n_it) mma(wf[j % WD], xf[j & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (a.debug) acc_loop += __builtin_amdgcn_s_memtime() - t_loop;
    }

    unsigned long long t_epi = 0;
    if (a.debug) t_epi = __builtin_amdgcn_s_memtime();
    unsigned long long t_mid = 0;
    conv_epilogue<STRIDE, PT, CT, WP>(a, acc, bias4, lane, wp, ct0, b, oy0, ox0, npix, a.debug ? &t_mid : nullptr);
    if (a.debug) {
        const unsigned long long t_alu = t_mid;                               // end of the ALU pass
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        acc_ld = (acc_ld << 24) | ((t_alu - t_epi) & 0xffffff);
        if (lane == 0) {
            const size_t w = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
            unsigned long long* o = a.debug + w * 6;
            o[0] = acc_ld; o[1] = acc_stage; o[2] = acc_loop; o[3] = t_end - t_epi; o[4] = t_end - t_start;
            o[5] = __builtin_amdgcn_s_memrealtime() - r_start;      // 100 MHz ticks over the same span
        }
    }
}

// ---------------------------------------------------------------------------------------------- v5
// v1 made persistent and software-pipelined (the 3x3 / strided form of what v4 does for pointwise convs): a block walks
// (output tile, channel chunk) items; before the K loop of item i starts, the halo tile of item i+1 is requested into
// NV registers per thread, and it is written to LDS after the loop -- the HBM/L2 latency of the staging hides under
// ~10-70k cycles of MFMAs instead of being exposed at the head of every block.  vmcnt retires in order, so the first
// WD-1 weight fragments of the next item are requested BEFORE the epilogue stores and the prefetch: waiting for them never
// waits for either.  An fp32 pipeline step is >= 512 cycles, so the prefetch has landed long before the first weight
// fragment requested after it is needed.  Accumulation order and epilogue are v1's: same bits.
template <int KS, int STRIDE, int PT, int CT, int WP, bool SINGLE>
__global__ __launch_bounds__(256) void conv_igemm_f32_v5(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WC = 4 / WP, TAPS = KS * KS, NV = 8;
    constexpr int WD = (CT <= 2) ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP;
    const int ct0 = (blockIdx.y * WC + wc) * CT;
    const int npix = a.TW * a.TH;
    const int nst = SINGLE ? 1 : (a.Cin + a.ck - 1) / a.ck;
    const int n_tiles = a.n_tiles_total;
    const int my_tiles = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_items = my_tiles * nst;
    if (n_items == 0) return;
    const int ck4m = (a.ck >> 2) - 1;
    const int total_f4 = a.npix_in << a.ck4_shift;

    auto tile_origin = [&](int item, int& b, int& oy0, int& ox0) {
        int t = (int)blockIdx.x + (SINGLE ? item : item / nst) * (int)gridDim.x;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        b = t / a.tiles_y;
        oy0 = ty * a.TH; ox0 = tx * a.TW;
    };
    auto prefetch = [&](int item, f32x4 (&v)[NV]) {      // item >= n_items: every lane reads the zero page
        const bool live = item < n_items;
        int b, oy0, ox0;
        tile_origin(live ? item : 0, b, oy0, ox0);
        const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
        const int c0 = SINGLE ? 0 : (item % nst) * a.ck;
        const float* srcb = a.src + (size_t)b * a.Hin * a.Win * a.src_cs;
        int tq = tid; asm volatile("" : "+v"(tq));       // opaque: the per-slot addresses are recomputed per item, not kept in ~50 VGPRs
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = u * 256 + tq;
            const int pix = idx >> a.ck4_shift, q = idx & ck4m;
            const int iy = (int)(((float)pix + 0.5f) * a.inv_TWin);
            const int ix = pix - iy * a.TWin;
            const int gy = iy0 + iy, gx = ix0 + ix, c = c0 + 4 * q;
            const bool inb = live && idx < total_f4 && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win && c < a.cin4;
            const float* gp = inb ? srcb + ((size_t)gy * a.Win + gx) * a.src_cs + c : a.zeros;
            v[u] = *(const f32x4*)gp;
        }
    };
    auto commit = [&](const f32x4 (&v)[NV]) {
        int tq = tid; asm volatile("" : "+v"(tq));
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = u * 256 + tq;
            if (idx < total_f4) *(f32x4*)(lds + (idx >> a.ck4_shift) * a.ldp + 4 * (idx & ck4m)) = v[u];
        }
    };
    int xoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int p = (wp * PT + pt) * 16 + (lane & 15);
        p = p < npix ? p : 0;
        const int ly = (int)(((float)p + 0.5f) * a.inv_TW);
        const int lx = p - ly * a.TW;
        xoff[pt] = ((ly * STRIDE) * a.TWin + lx * STRIDE) * a.ldp + (lane >> 4) * 4;
    }
    const float* wbase[CT];
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = a.wpk + (size_t)ctile * TAPS * a.cib * 256 + lane * 4;
        bias4[ct] = *(const f32x4*)(a.bias + ctile * 16 + (lane >> 4) * 4);
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K-loop state of the item whose weights are being fetched (set up by begin_item, consumed by the loop below)
    const int wstep = a.cib * 256;
    int n_it = 0, cib0 = 0;
    int w_it = 0, w_kw = 0, w_kh = 0, w_kk = 0, w_off = 0;
    f32x4 wf[WD][CT];
    auto load_w = [&](f32x4* w) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f32x4*)(wbase[ct] + w_off);
        if (w_it + 1 < n_it) {
            ++w_it; ++w_kw; w_off += wstep;
            if (w_kw == KS) {
                w_kw = 0; ++w_kh;
                if (w_kh == KS) { w_kh = 0; ++w_kk; w_off = (cib0 + w_kk) * 256; }
            }
        }
    };
    auto begin_item = [&](int item) {                       // cursor reset + the first WD-1 weight fragments
        const int st = SINGLE ? 0 : item % nst;
        const int c0 = st * a.ck;
        const int rem = a.Cin - c0;
        n_it = (((rem < a.ck ? rem : a.ck) + 15) >> 4) * TAPS;
        cib0 = c0 >> 4;
        w_it = 0; w_kw = 0; w_kh = 0; w_kk = 0; w_off = cib0 * 256;
#pragma unroll
        for (int j = 0; j < WD - 1; ++j) load_w(wf[j]);
    };

    f32x4 pv[NV];
    prefetch(0, pv);
    commit(pv);
    __syncthreads();
    begin_item(0);
    prefetch(1, pv);
    for (int item = 0; item < n_items; ++item) {
        int x_it = 0, x_kw = 0, x_kh = 0, x_kk = 0, x_off = 0;
        f32x4 xf[2][PT];
        auto load_x = [&](f32x4* x) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) x[pt] = *(const f32x4*)__builtin_assume_aligned(lds + xoff[pt] + x_off, 16);
            if (x_it + 1 < n_it) {
                ++x_it; ++x_kw; x_off += a.ldp;
                if (x_kw == KS) {
                    x_kw = 0; ++x_kh; x_off += (a.TWin - KS) * a.ldp;
                    if (x_kh == KS) { x_kh = 0; ++x_kk; x_off = x_kk * 16; }
                }
            }
        };
        auto mma = [&](const f32x4* w, const f32x4* x) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][s], x[pt][s], acc[ct][pt], 0, 0, 0);
        };
        load_x(xf[0]);
        const int n_cur = n_it;
        for (int it = 0; it < n_cur; it += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                load_w(wf[(j + WD - 1) % WD]);
                load_x(xf[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (it + j < n_cur) mma(wf[j % WD], xf[j & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                   // every wave is done reading this item's LDS image
        commit(pv);                                        // item + 1 (zeros after the last one)
        const bool tile_done = SINGLE || (item % nst) == nst - 1;
        begin_item(item + 1 < n_items ? item + 1 : item);  // next weights requested before the stores and the prefetch
        if (tile_done) {
            int b, oy0, ox0;
            tile_origin(item, b, oy0, ox0);
            conv_epilogue<STRIDE, PT, CT, WP>(a, acc, bias4, lane, wp, ct0, b, oy0, ox0, npix);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        prefetch(item + 2, pv);
        __syncthreads();                                   // item + 1's LDS image is complete
    }
}

// ---------------------------------------------------------------------------------------------- v2
// Same GEMM, same canonical accumulation order, different staging: a 5th wave is a pure LOADER.  It fills the next
// stage's halo tile with LDS-DMA (global_load_lds_dwordx4: 1 KiB per instruction, no VGPR round trip) into the second
// LDS buffer while the four compute waves issue MFMAs on the current one, so no compute wave ever waits on an
// activation load (vmcnt is in-order per wave: the compute waves only have weight-fragment loads in flight).
// The LDS image is dense [pixel][ck] with the 16-byte slot index XOR-swizzled by the pixel index (applied on the
// DMA *source* side and on the ds_read_b128 side) so that the 16 lanes of a ds_read_b128 group hit 16 distinct slots.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ int lds_swz(int pix, int sshift) {
    return sshift == 2 ? ((-(pix >> 2)) & 3) : (sshift == 3 ? ((pix >> 1) & 7) : (pix & 15));
}

template <int KS, int STRIDE, int PT, int CT, int WP>
__global__ __launch_bounds__(320) void conv_igemm_f32_v2(ConvKArgs a) {
    // PERSISTENT over output tiles: block x handles tiles x, x + gridDim.x, ...  The unit of pipelining is one
    // (tile, K-stage) item; the loader wave always works one item ahead of the compute waves, so the HBM reads of
    // tile t+1 (and, through other blocks, the epilogue writes of tile t-1) overlap the MFMAs of tile t.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WC = 4 / WP;
    constexpr int TAPS = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave == 4;
    const int wp = wave % WP, wc = (wave & 3) / WP;
    const int ct0 = (blockIdx.y * WC + wc) * CT;
    const int npix = a.TW * a.TH;
    const int sshift = a.ck4_shift, S = a.ck >> 2;
    const int nst = (a.Cin + a.ck - 1) / a.ck;
    const int pieces = ((a.npix_in << sshift) + 63) >> 6;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int n_tiles = a.n_tiles_total;
    const int my_tiles = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_items = my_tiles * nst;

    auto tile_origin = [&](int item, int& b, int& oy0, int& ox0) {
        int t = (int)blockIdx.x + (item / nst) * (int)gridDim.x;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        b = t / a.tiles_y;
        oy0 = ty * a.TH; ox0 = tx * a.TW;
    };
    (void)tiles_per_img;

    auto issue_item = [&](int item) {                        // loader wave only
        int b, oy0, ox0;
        tile_origin(item, b, oy0, ox0);
        const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
        const int c0 = (item % nst) * a.ck;
        float* buf = lds + (item & 1) * a.lds_buf_floats;
        const float* srcb = a.src + (size_t)b * a.Hin * a.Win * a.src_cs;
        for (int j = 0; j < pieces; ++j) {
            const int i = j * 64 + lane;
            const int pix = i >> sshift;
            const int q = (i & (S - 1)) ^ lds_swz(pix, sshift);
            const int iy = (int)(((float)pix + 0.5f) * a.inv_TWin);
            const int ix = pix - iy * a.TWin;
            const int gy = iy0 + iy, gx = ix0 + ix, c = c0 + 4 * q;
            const float* g = a.zeros;
            if (pix < a.npix_in && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win && c < a.cin4)
                g = srcb + ((size_t)gy * a.Win + gx) * a.src_cs + c;
            __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)(buf + j * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    int pbase[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int p = (wp * PT + pt) * 16 + (lane & 15);
        p = p < npix ? p : 0;
        const int ly = (int)(((float)p + 0.5f) * a.inv_TW);
        const int lx = p - ly * a.TW;
        pbase[pt] = (ly * STRIDE) * a.TWin + lx * STRIDE;                  // LDS pixel index of tap (0,0)
    }
    const int g4 = lane >> 4;
    const float* wbase[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = a.wpk + (size_t)ctile * TAPS * a.cib * 256 + lane * 4;
    }
    f32x4 acc[CT][PT];
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        bias4[ct] = *(const f32x4*)(a.bias + ctile * 16 + (lane >> 4) * 4);
    }

    if (loader && n_items > 0) issue_item(0);
    __syncthreads();
    for (int item = 0; item < n_items; ++item) {
        if (loader) {
            if (item + 1 < n_items) issue_item(item + 1);
        } else {
            const int st = item % nst;
            if (st == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const float* buf = lds + (item & 1) * a.lds_buf_floats;
            const int c0 = st * a.ck;
            const int rem = a.Cin - c0;
            const int nkk = ((rem < a.ck ? rem : a.ck) + 15) >> 4;
            const int cib0 = c0 >> 4;
            constexpr int WD = (CT <= 2) ? 4 : 2;
            const int n_it = nkk * TAPS;
            const int wstep = a.cib * 256;
            int w_it = 0, w_kw = 0, w_kh = 0, w_kk = 0, w_off = cib0 * 256;
            int x_it = 0, x_kw = 0, x_kh = 0, x_kk = 0, x_pix = 0;          // pixel cursor: kh*TWin + kw (LDS pixels)
            f32x4 wf[WD][CT], xf[2][PT];
            auto load_w = [&](f32x4* w) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f32x4*)(wbase[ct] + w_off);
                if (w_it + 1 < n_it) {
                    ++w_it; ++w_kw; w_off += wstep;
                    if (w_kw == KS) {
                        w_kw = 0; ++w_kh;
                        if (w_kh == KS) { w_kh = 0; ++w_kk; w_off = (cib0 + w_kk) * 256; }
                    }
                }
            };
            auto load_x = [&](f32x4* x) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const int pix = pbase[pt] + x_pix;
                    x[pt] = *(const f32x4*)__builtin_assume_aligned(
                        buf + (pix << (sshift + 2)) + ((((x_kk << 2) + g4) ^ lds_swz(pix, sshift)) << 2), 16);
                }
                if (x_it + 1 < n_it) {
                    ++x_it; ++x_kw; ++x_pix;
                    if (x_kw == KS) {
                        x_kw = 0; ++x_kh; x_pix += a.TWin - KS;
                        if (x_kh == KS) { x_kh = 0; ++x_kk; x_pix = 0; }
                    }
                }
            };
            auto mma = [&](const f32x4* w, const f32x4* x) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][s], x[pt][s], acc[ct][pt], 0, 0, 0);
            };
#pragma unroll
            for (int j = 0; j < WD - 1; ++j) load_w(wf[j]);
            load_x(xf[0]);
            for (int it = 0; it < n_it; it += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    load_w(wf[(j + WD - 1) % WD]);
                    load_x(xf[(j + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (it + j < n_it) mma(wf[j % WD], xf[j & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (st == nst - 1) {
                int b, oy0, ox0;
                tile_origin(item, b, oy0, ox0);
                conv_epilogue<STRIDE, PT, CT, WP>(a, acc, bias4, lane, wp, ct0, b, oy0, ox0, npix);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- v3 (1x1 only)
// Pointwise convs have no tap reuse, so staging pixels through LDS buys nothing: here every wave streams its pixel
// fragments straight from global memory into the MFMA B-operand layout (lane (p, g) reads the 16 bytes of channels
// 4g..4g+3 of pixel p: 64 contiguous bytes per pixel per 16-channel block), with a 4-deep register prefetch ring for
// pixels (HBM latency) and weights (L2 latency).  No LDS, no barriers: waves drift apart and overlap each other's
// epilogues.  Same canonical accumulation order as v1 (16-channel block outer, step s, k-group g).
template <int PT, int CT>
__global__ __launch_bounds__(256) void conv1x1_stream_f32(ConvKArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int total = a.Wout;                                        // flattened pixels (Hout == 1)
    const int tile0 = ((int)blockIdx.x * 4 + wave) * PT;             // first 16-pixel tile of this wave
    const int ct0 = (int)blockIdx.y * CT;
    const float* xbase[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int p = (tile0 + pt) * 16 + (lane & 15);
        p = p < total ? p : total - 1;                               // clamp: results of padded pixels are never stored
        xbase[pt] = a.src + (size_t)p * a.src_cs + 4 * g;
    }
    const float* wbase[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = a.wpk + (size_t)ctile * a.cib * 256 + lane * 4;
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        bias4[ct] = *(const f32x4*)(a.bias + ctile * 16 + g * 4);
    }
    constexpr int D = 4;                                             // prefetch depth (steps in flight: D-1)
    const int n_it = a.cib;
    // in the last 16-channel block the lanes whose 4 channels lie beyond round_up(Cin, 4) read zeros instead
    const bool tail_oob = (n_it - 1) * 16 + 4 * g >= a.cin4;
    f32x4 wf[D][CT], xf[D][PT];
    int l_it = 0;
    auto load = [&](f32x4* w, f32x4* x) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f32x4*)(wbase[ct] + l_it * 256);
        const bool oob = tail_oob && (l_it == n_it - 1);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const float* px = oob ? a.zeros : xbase[pt] + l_it * 16;
            x[pt] = *(const f32x4*)px;
        }
        if (l_it + 1 < n_it) ++l_it;                                 // clamp instead of guarding the loads
    };
    auto mma = [&](const f32x4* w, const f32x4* x) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][s], x[pt][s], acc[ct][pt], 0, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load(wf[j], xf[j]);
    for (int it = 0; it < n_it; it += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            load(wf[(j + D - 1) % D], xf[(j + D - 1) % D]);
            __builtin_amdgcn_sched_barrier(0);
            if (it + j < n_it) mma(wf[j], xf[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue (same math as conv_epilogue, flattened pixel index)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (tile0 + pt) * 16 + (lane & 15);
        const bool ok = p < total;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = (ct0 + ct) * 16 + g * 4;
            if (!ok || c >= a.Cout) continue;
            f32x4 v = acc[ct][pt] + bias4[ct];
            if (a.act) {
                v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            }
            float* d = a.dst + (size_t)p * a.dst_cs + c;
            if (c + 3 < a.Cout) {
                if (a.res) v += *(const f32x4*)(a.res + (size_t)p * a.res_cs + c);
                *(f32x4*)d = v;
            } else {
                for (int i = 0; i < 4 && c + i < a.Cout; ++i) {
                    float r = v[i];
                    if (a.res) r += a.res[(size_t)p * a.res_cs + c + i];
                    d[i] = r;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- v4 (1x1 only)
// Pointwise convs have a short K loop per staged chunk (Cin/16 steps), so in v1 the block spends as long waiting for its
// activation loads as it spends on MFMAs.  v4 is a PERSISTENT, software-pipelined form of v1 for k = 1: the unit of work
// is one (pixel tile, channel chunk) item; while the MFMAs of item i run from LDS, the global loads of item i+1 are
// already in flight into registers, and the stores of the previous tile drain behind them.  vmcnt retires in order, so
// the order of issue is what makes this work: the chunk's weight fragments (all of them: a 1x1 chunk has at most 4 k-blocks)
// are requested BEFORE the prefetch, hence waiting for them never waits for the prefetch.
// Same canonical accumulation order as v1 (16-channel block outer, step s, k-group g): same bits.
template <int PT, int CT, int WP, bool SINGLE, int NKK>   // SINGLE: Cin fits one chunk -> every item ends a tile; NKK: k-blocks per chunk
__global__ __launch_bounds__(256) void conv1x1_pipe_f32(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WC = 4 / WP, NV = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP, g = lane >> 4;
    const int ct0 = (blockIdx.y * WC + wc) * CT;
    const int P = a.TW, total = a.Wout;
    const int sh = a.ck4_shift, ck4m = (a.ck >> 2) - 1, tile_v = P << sh;
    const int nst = SINGLE ? 1 : (a.Cin + a.ck - 1) / a.ck;
    const int n_tiles = a.n_tiles_total;
    const int my_tiles = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_items = my_tiles * nst;
    if (n_items == 0) return;

    auto prefetch = [&](int item, f32x4 (&v)[NV]) {      // item >= n_items: every lane reads the zero page (loads stay unconditional)
        const bool live = item < n_items;
        const int ti = SINGLE ? item : item / nst;
        const int p0 = ((int)blockIdx.x + ti * (int)gridDim.x) * P;
        const int c0 = SINGLE ? 0 : (item - ti * nst) * a.ck;
        // fused upsample: (image, row, column) of the tile's first pixel, once per item (uniform)
        int ub = 0, uy = 0, ux = 0;
        if (a.up_c) { const int hw = a.up_W * a.up_H; ub = p0 / hw; const int r = p0 - ub * hw; uy = r / a.up_W; ux = r - uy * a.up_W; }
        int tq = tid; asm volatile("" : "+v"(tq));       // opaque: slot addresses recomputed per item instead of living in VGPRs
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = u * 256 + tq;
            const int pix = idx >> sh, q = idx & ck4m;
            const int p = p0 + pix, c = c0 + 4 * q;
            const bool inb = live && idx < tile_v && p < total && c < a.cin4;
            const float* src = a.src + (size_t)p * a.src_cs + c;
            if (c < a.up_c) {                            // channels of the upsampled operand: read pixel (y/2, x/2) of the half-size map
                const int xx = ux + pix;                 // < W + tile: the quotient is tiny, the float form is exact
                const int wr = (int)(((float)xx + 0.5f) * a.inv_TW);        // inv_TW = 1 / up_W for these launches
                const int x = xx - wr * a.up_W, yy = uy + wr;
                const int hr = (int)(((float)yy + 0.5f) * a.inv_TWin);      // inv_TWin = 1 / up_H; a tile may span several small images
                const int y = yy - hr * a.up_H, b = ub + hr;
                src = a.src2 + (((size_t)b * (a.up_H >> 1) + (y >> 1)) * (a.up_W >> 1) + (x >> 1)) * a.src2_cs + c;
            }
            if (!inb) src = a.zeros;
            v[u] = *(const f32x4*)src;
        }
    };
    auto commit = [&](const f32x4 (&v)[NV]) {
        int tq = tid; asm volatile("" : "+v"(tq));
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = u * 256 + tq;
            if (idx < tile_v) *(f32x4*)(lds + (idx >> sh) * a.ldp + 4 * (idx & ck4m)) = v[u];
        }
    };
    int xoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xoff[pt] = ((wp * PT + pt) * 16 + (lane & 15)) * a.ldp + 4 * g;
    const float* wbase[CT];
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = a.wpk + (size_t)ctile * a.cib * 256 + lane * 4;
        bias4[ct] = *(const f32x4*)(a.bias + ctile * 16 + g * 4);
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 pv[NV];
    prefetch(0, pv);
    commit(pv);
    __syncthreads();
    f32x4 w[NKK][CT];
    auto load_w = [&](int item) {
        const int st = SINGLE ? 0 : item % nst;
        const int cib0 = (st * a.ck) >> 4;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int kb = cib0 + kk < a.cib ? cib0 + kk : a.cib - 1;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) w[kk][ct] = *(const f32x4*)(wbase[ct] + kb * 256);
        }
    };
    load_w(0);
    prefetch(1, pv);
    for (int item = 0; item < n_items; ++item) {
        const int st = SINGLE ? 0 : item % nst;
        const int c0 = st * a.ck;
        const int rem = a.Cin - c0;
        const int nkk = ((rem < a.ck ? rem : a.ck) + 15) >> 4;
        // ---- K loop of this item: pixel fragments one k-block ahead (two register sets), weights already in registers
        f32x4 xf[2][PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xf[0][pt] = *(const f32x4*)__builtin_assume_aligned(lds + xoff[pt], 16);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int kn = (kk + 1 < nkk ? kk + 1 : nkk - 1) * 16;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) xf[(kk + 1) & 1][pt] = *(const f32x4*)__builtin_assume_aligned(lds + xoff[pt] + kn, 16);
            __builtin_amdgcn_sched_barrier(0);
            if (kk < nkk) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kk][ct][s], xf[kk & 1][pt][s], acc[ct][pt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // every wave is done reading this item's LDS image
        commit(pv);                                        // item + 1 (zeros after the last one)
        load_w(item + 1 < n_items ? item + 1 : item);      // requested before the stores and the next prefetch
        if (SINGLE || st == nst - 1) {
            // ---- epilogue of the tile that just finished (same math as conv_epilogue, flattened pixel index)
            const int ti = SINGLE ? item : item / nst;
            const int p0 = ((int)blockIdx.x + ti * (int)gridDim.x) * P;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int lp = (wp * PT + pt) * 16 + (lane & 15);
                const int p = p0 + lp;
                const bool ok = lp < P && p < total;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = (ct0 + ct) * 16 + g * 4;
                    f32x4 v = acc[ct][pt] + bias4[ct];
                    acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (!ok || c >= a.Cout) continue;
                    if (a.act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
                    float* d = a.dst + (size_t)p * a.dst_cs + c;
                    if (c + 3 < a.Cout) {
                        if (a.res) v += *(const f32x4*)(a.res + (size_t)p * a.res_cs + c);
                        *(f32x4*)d = v;
                    } else {
                        for (int i = 0; i < 4 && c + i < a.Cout; ++i) {
                            float r = v[i];
                            if (a.res) r += a.res[(size_t)p * a.res_cs + c + i];
                            d[i] = r;
                        }
                    }
                }
            }
        }
        prefetch(item + 2, pv);
        __syncthreads();                                   // item + 1's LDS image is complete
    }
}

// ---------------------------------------------------------------------------------------------- host side
size_t packed_weight_floats(int cout, int cin, int k) {
    return (size_t)((cout + 15) / 16) * k * k * ((cin + 15) / 16) * 256;
}

void pack_conv_weights(const float* w, int cout, int cin, int k, float* out) {
    const int nct = (cout + 15) / 16, cib = (cin + 15) / 16, taps = k * k;
    for (int ct = 0; ct < nct; ++ct)
        for (int tap = 0; tap < taps; ++tap)
            for (int cb = 0; cb < cib; ++cb) {
                float* o = out + ((size_t)(ct * taps + tap) * cib + cb) * 256;
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) {
                        const int co = ct * 16 + (lane & 15);
                        const int ci = cb * 16 + 4 * (lane >> 4) + s;
                        o[lane * 4 + s] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * taps + tap] : 0.f;
                    }
            }
}

namespace {

struct Plan { int CT, WP, TW, TH, ck; size_t lds; double cost; int version; int buf_floats; int PT; };   // PT 0 = default (4, or 3 with CT 5)

typedef void (*KernelFn)(ConvKArgs);

template <int KS, int STRIDE, int CT, int WP>
KernelFn inst() { return &conv_igemm_f32<KS, STRIDE, (CT == 5 ? 3 : 4), CT, WP>; }

template <int KS, int STRIDE, int CT, int WP>
KernelFn inst2() { return &conv_igemm_f32_v2<KS, STRIDE, (CT == 5 ? 3 : 4), CT, WP>; }

template <int KS, int STRIDE>
KernelFn pick_ct_wp2(int CT, int WP) {
#define MI355_CASE(ct, wp) if (CT == ct && WP == wp) return inst2<KS, STRIDE, ct, wp>();
    MI355_CASE(1, 4) MI355_CASE(2, 4) MI355_CASE(3, 4) MI355_CASE(4, 4) MI355_CASE(5, 4)
    MI355_CASE(1, 2) MI355_CASE(2, 2) MI355_CASE(3, 2) MI355_CASE(4, 2) MI355_CASE(5, 2)
    MI355_CASE(1, 1) MI355_CASE(2, 1) MI355_CASE(3, 1) MI355_CASE(4, 1) MI355_CASE(5, 1)
#undef MI355_CASE
    return nullptr;
}

template <int KS, int STRIDE>
KernelFn pick_ct_wp(int CT, int WP) {
#define MI355_CASE(ct, wp) if (CT == ct && WP == wp) return inst<KS, STRIDE, ct, wp>();
    MI355_CASE(1, 4) MI355_CASE(2, 4) MI355_CASE(3, 4) MI355_CASE(4, 4) MI355_CASE(5, 4)
    MI355_CASE(1, 2) MI355_CASE(2, 2) MI355_CASE(3, 2) MI355_CASE(4, 2) MI355_CASE(5, 2)
    MI355_CASE(1, 1) MI355_CASE(2, 1) MI355_CASE(3, 1) MI355_CASE(4, 1) MI355_CASE(5, 1)
#undef MI355_CASE
    return nullptr;
}

KernelFn pick_stream(int CT, int PT) {
    if (CT == 1 && PT == 2) return &conv1x1_stream_f32<2, 1>;
    if (CT == 1 && PT == 4) return &conv1x1_stream_f32<4, 1>;
    if (CT == 2 && PT == 2) return &conv1x1_stream_f32<2, 2>;
    if (CT == 2 && PT == 4) return &conv1x1_stream_f32<4, 2>;
    if (CT == 4 && PT == 2) return &conv1x1_stream_f32<2, 4>;
    if (CT == 4 && PT == 4) return &conv1x1_stream_f32<4, 4>;
    return nullptr;
}

template <int KS, int STRIDE, bool SINGLE>
KernelFn pick_v5_s(int CT, int WP) {
#define MI355_CASE5(ct, wp) if (CT == ct && WP == wp) return &conv_igemm_f32_v5<KS, STRIDE, 4, ct, wp, SINGLE>;
    MI355_CASE5(1, 4) MI355_CASE5(2, 4) MI355_CASE5(3, 4) MI355_CASE5(4, 4)
    MI355_CASE5(1, 2) MI355_CASE5(2, 2) MI355_CASE5(3, 2) MI355_CASE5(4, 2)
    MI355_CASE5(1, 1) MI355_CASE5(2, 1) MI355_CASE5(3, 1) MI355_CASE5(4, 1)
#undef MI355_CASE5
    return nullptr;
}

KernelFn pick_v5(int ks, int stride, int CT, int WP, bool single) {
    if (ks != 3) return nullptr;
    if (stride == 1) return single ? pick_v5_s<3, 1, true>(CT, WP) : pick_v5_s<3, 1, false>(CT, WP);
    if (stride == 2) return single ? pick_v5_s<3, 2, true>(CT, WP) : pick_v5_s<3, 2, false>(CT, WP);
    return nullptr;
}

template <bool SINGLE, int NKK>
KernelFn pick_pipe_s(int CT, int WP) {
#define MI355_CASE4(ct, wp) if (CT == ct && WP == wp) return &conv1x1_pipe_f32<4, ct, wp, SINGLE, NKK>;
    MI355_CASE4(1, 1) MI355_CASE4(2, 1) MI355_CASE4(4, 1)
    MI355_CASE4(1, 2) MI355_CASE4(2, 2) MI355_CASE4(4, 2)
    MI355_CASE4(1, 4) MI355_CASE4(2, 4) MI355_CASE4(4, 4)
#undef MI355_CASE4
    return nullptr;
}

KernelFn pick_pipe(int CT, int WP, bool single, int ck) {
    if (ck > 64) {                       // 8 k-blocks per chunk: weights of a chunk = 8*CT fragments in registers
        if (CT > 2) return nullptr;
        return single ? pick_pipe_s<true, 8>(CT, WP) : pick_pipe_s<false, 8>(CT, WP);
    }
    return single ? pick_pipe_s<true, 4>(CT, WP) : pick_pipe_s<false, 4>(CT, WP);
}

KernelFn pick_kernel(int ks, int stride, int CT, int WP, int version) {
    if (version == 2) {
        if (ks == 1 && stride == 1) return pick_ct_wp2<1, 1>(CT, WP);
        if (ks == 3 && stride == 1) return pick_ct_wp2<3, 1>(CT, WP);
        if (ks == 3 && stride == 2) return pick_ct_wp2<3, 2>(CT, WP);
        return nullptr;
    }
    if (ks == 1 && stride == 1) return pick_ct_wp<1, 1>(CT, WP);
    if (ks == 3 && stride == 1) return pick_ct_wp<3, 1>(CT, WP);
    if (ks == 3 && stride == 2) return pick_ct_wp<3, 2>(CT, WP);
    return nullptr;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

constexpr size_t LDS_SOFT = 40 * 1024, LDS_HARD = 64 * 1024;

// Candidate launch plans for one conv: for every wave arrangement (CT, WC) the best output tile, with every
// feasible staged-channel count.  Sorted by a static cost model; the engine may time the first few (autotune).
std::vector<Plan> enumerate_plans(int H, int W, int n_ctiles, int cin, int ks, int stride, bool allow_v2, bool half) {
    static const int use_v2 = env_int("MI355_CONV_V2", 0);   // the loader-wave kernel never won on this network: opt-in
    static const int max_ct = env_int("MI355_MAX_CT", 5);          // tuning knobs (experiments only)
    static const int min_wc = env_int("MI355_MIN_WC", 1);
    std::vector<Plan> out;
    const int cin16 = round_up(cin, 16);
    // fp16 only: wave tiles of 8 pixel tiles (128 pixels x CT*16 couts) halve the weight bytes fetched per MFMA -- the f16
    // MFMA retires a 1-KiB fragment pair in 16 cycles, so these kernels are bound by L1/L2 fragment traffic, not by the pipe
    for (int PTsel = 0; PTsel <= (half ? 8 : 0); PTsel += 8)
    for (int WC = 1; WC <= 4; WC *= 2)
        for (int CT = 1; CT <= 5; ++CT) {
            if (CT > max_ct || WC < min_wc) continue;
            if (PTsel == 8 && CT > 4) continue;
            const int WP = 4 / WC, PT = PTsel ? PTsel : (CT == 5 ? 3 : 4), P = WP * PT * 16;
            const int cover = CT * WC, nblk = (n_ctiles + cover - 1) / cover;
            if (cover >= 2 * n_ctiles && cover > CT) continue;          // more than half of the cout tiles would be padding
            const double waste_c = (double)nblk * cover / n_ctiles;
            // staged channels per chunk, in 4-byte units; the fp16 kernels (2 channels per unit) also get 128: their K loop
            // is so short that the two barriers + pipeline refill per chunk show, above all in the 1x1 layers
            for (int ck = half ? 128 : 64; ck >= 16; ck >>= 1) {
                if (ck > cin16 && ck != 16) continue;
                Plan best{}; best.cost = 1e30;
                for (int TW = 1; TW <= P && TW <= W; ++TW) {
                    int TH = P / TW; if (TH > H) TH = H;
                    if (TH < 1) continue;
                    const long tiles = (long)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
                    const int THin = (TH - 1) * stride + ks, TWin = (TW - 1) * stride + ks;
                    const size_t lds = (size_t)THin * TWin * (ck + 4) * 4;
                    if (lds > LDS_HARD) continue;
                    const double infl = waste_c * (double)tiles * P / ((double)W * H);
                    const double halo = (double)THin * TWin / ((double)TH * TW * stride * stride);
                    const int stages = (cin16 + ck - 1) / ck;
                    double cost = infl * (1.0 + 0.03 * halo * nblk) * (1.0 + 0.04 * (stages - 1)) * (1.0 + 0.04 * (CT - 1))
                                  + (lds > LDS_SOFT ? 0.15 : 0.0);
                    if (cost < best.cost) best = Plan{CT, WP, TW, TH, ck, lds, cost, 1, 0, PTsel};
                }
                if (best.cost < 1e30) {
                    out.push_back(best);
                    {   // v5: the same tile, persistent + prefetched (the halo chunk must fit 8 float4 registers per thread)
                        static const int use_v5 = env_int("MI355_CONV_V5", 0);   // measured: loses to v1 (3 vs 5 waves/SIMD); opt-in, tested
                        const int THin5 = (best.TH - 1) * stride + ks, TWin5 = (best.TW - 1) * stride + ks;
                        if (use_v5 && allow_v2 && !half && ks == 3 && CT <= 4 && PTsel == 0 && THin5 * TWin5 * ck / 4 <= 2048) {
                            Plan v5 = best;
                            v5.version = 5; v5.cost = best.cost * 0.9;
                            out.push_back(v5);
                        }
                    }
                    // v2 variant of the same tile: dense double-buffered LDS image filled by a DMA loader wave
                    const int stages = (cin16 + ck - 1) / ck;
                    const int THin = (best.TH - 1) * stride + ks, TWin = (best.TW - 1) * stride + ks;
                    const int buf_floats = round_up(THin * TWin * ck, 256);
                    const size_t lds2 = (size_t)buf_floats * 4 * 2;      // always double-buffered: the pipeline runs across tiles
                    if (allow_v2 && use_v2 && !half && lds2 <= LDS_HARD) {
                        Plan v2 = best;
                        v2.version = 2; v2.buf_floats = buf_floats; v2.lds = lds2;
                        v2.cost = best.cost * 0.98 + (lds2 > LDS_SOFT + 24 * 1024 ? 0.1 : 0.0);
                        out.push_back(v2);
                    }
                }
            }
        }
    if (ks == 1 && allow_v2) {        // streaming pointwise kernel (needs the zero page as well): CT x PT register tiles
        static const int use_v3 = env_int("MI355_CONV_V3", 1);
        const int cts[3] = {1, 2, 4}, pts[2] = {2, 4};
        for (int ci = 0; ci < 3 && use_v3; ++ci)
            for (int pi = 0; pi < 2; ++pi) {
                const int CT = cts[ci], PT = pts[pi];
                if (CT > n_ctiles && CT != 1) continue;
                const int nblk = (n_ctiles + CT - 1) / CT;
                Plan p3{CT, 4, PT * 64, 1, 16, 0, 0.0, 3, PT, 0};
                p3.cost = (double)nblk * CT / n_ctiles * (1.0 + 0.05 * nblk) * 0.9;
                out.push_back(p3);
            }
    }
    if (ks == 1 && allow_v2) {   // persistent software-pipelined pointwise kernel (v4); ck in 4-byte units (fp16: 2 channels each)
        static const int use_v4 = env_int("MI355_CONV_V4", 1);
        const int wps[3] = {1, 2, 4}, cts[4] = {1, 2, 4, 3}, cks[4] = {128, 64, 32, 16};
        for (int wi = 0; wi < 3 && use_v4; ++wi)
            for (int ci = 0; ci < (half ? 4 : 3); ++ci)
                for (int ki = 0; ki < 4; ++ki) {
                    const int WP = wps[wi], WC = 4 / WP, CT = cts[ci], ck = cks[ki], P = WP * 64;
                    const int cover = CT * WC, nblk = (n_ctiles + cover - 1) / cover;
                    if (cover >= 2 * n_ctiles && cover > CT) continue;
                    if (ck > cin16 && ck != 16) continue;
                    if (P * ck / 4 > 2048) continue;                       // 8 prefetch registers (float4) per thread
                    if (ck > 64 && CT > (half ? 3 : 2)) continue;          // a chunk's weights live in registers: 8 k-blocks x CT fragments
                    const size_t lds = (size_t)P * (ck + 4) * 4;
                    const int stages = (cin16 + ck - 1) / ck;
                    Plan p4{CT, WP, P, 1, ck, lds, 0.0, 4, 0, 0};
                    p4.cost = (double)nblk * cover / n_ctiles * (1.0 + 0.03 * (stages - 1)) * (1.0 + 0.02 * nblk) * 0.8;
                    out.push_back(p4);
                }
    }
    std::sort(out.begin(), out.end(), [](const Plan& a, const Plan& b) { return a.cost < b.cost; });
    return out;
}

}  // namespace

static const char* check_args(const ConvArgs& c) {
    if (!((c.k == 1 && c.stride == 1) || (c.k == 3 && (c.stride == 1 || c.stride == 2))))
        return "conv: only 1x1/s1, 3x3/s1 and 3x3/s2 are supported";
    if (c.dtype == 1) {     // fp16 storage: 16-byte source vectors = 8 halfs, 8-byte (or fp32 16-byte) destination vectors
        if ((c.src_cs & 7) || (c.dst_cs & 3) || (c.res && (c.res_cs & 3))) return "conv(f16): channel strides must be multiples of 8 (src) / 4 (dst)";
        if (((uintptr_t)c.src | (uintptr_t)c.wpk | (uintptr_t)c.bias) & 15) return "conv(f16): src / weight / bias pointers must be 16-byte aligned";
        const bool wide = !c.out_f32 && conv_f16_pairs(c.Cout);          // 16-byte fp16 stores of 8 consecutive couts
        if (((uintptr_t)c.dst | (uintptr_t)c.res) & ((c.out_f32 || wide) ? 15 : 7)) return "conv(f16): dst / residual pointers are misaligned";
        if (wide && ((c.dst_cs & 7) || (c.res && (c.res_cs & 7)))) return "conv(f16): dst / residual strides must be multiples of 8";
        if (!c.zeros) return "conv: zero page missing";
        return nullptr;
    }
    if ((c.src_cs & 3) || (c.dst_cs & 3) || (c.res && (c.res_cs & 3))) return "conv: channel strides must be multiples of 4";
    if (((uintptr_t)c.src | (uintptr_t)c.dst | (uintptr_t)c.res | (uintptr_t)c.wpk | (uintptr_t)c.bias) & 15)
        return "conv: pointers must be 16-byte aligned";
    if (!c.zeros) return "conv: zero page missing";
    return nullptr;
}

static const char* build_launch(const ConvArgs& c, const Plan& p, ConvLaunch* out) {
    ConvKArgs a{};
    a.src = c.src; a.dst = c.dst; a.res = c.res; a.wpk = c.wpk; a.bias = c.bias;
    a.src_cs = c.src_cs; a.dst_cs = c.dst_cs; a.res_cs = c.res_cs;
    a.Cin = c.Cin; a.Cout = c.Cout; a.pad = c.pad; a.act = c.act;
    const bool half = c.dtype == 1;
    a.cib = half ? (c.Cin + 31) / 32 : (c.Cin + 15) / 16; a.n_ctiles = (c.Cout + 15) / 16;
    a.cin4 = half ? round_up(c.Cin, 8) : round_up(c.Cin, 4);       // channels covered by whole 16-byte vectors
    a.out_f32 = c.out_f32;
    a.src2 = c.src2; a.src2_cs = c.src2_cs; a.up_c = c.src2 ? c.up_c : 0; a.up_W = c.Win; a.up_H = c.Hin;
    int B = c.B;
    if (c.k == 1) {   // pointwise: flatten batch and space into one row of pixels
        a.Hin = 1; a.Win = c.B * c.Hin * c.Win; a.Hout = 1; a.Wout = a.Win; B = 1;
    } else {
        a.Hin = c.Hin; a.Win = c.Win; a.Hout = c.Hout; a.Wout = c.Wout;
    }
    KernelFn fn = half ? (p.version == 4 ? (KernelFn)pick_conv_pipe_f16(p.CT, p.WP, (c.Cin + 1) / 2 <= p.ck, p.ck > 64)
                                         : (KernelFn)pick_conv_kernel_f16(c.k, c.stride, p.CT, p.WP, p.version, p.version == 3 ? p.buf_floats : p.PT))
                       : (p.version == 3 ? pick_stream(p.CT, p.buf_floats)
                          : p.version == 4 ? pick_pipe(p.CT, p.WP, c.Cin <= p.ck, p.ck)
                          : p.version == 5 ? pick_v5(c.k, c.stride, p.CT, p.WP, c.Cin <= p.ck) : pick_kernel(c.k, c.stride, p.CT, p.WP, p.version));
    if (!fn) return "conv: no kernel instance";
    a.zeros = c.zeros; a.lds_buf_floats = p.buf_floats;
    if (half && p.version == 4) a.lds_buf_floats = 0;
    if (half && p.version == 1) { static const int ex = env_int("MI355_F16_EXP", 0); a.lds_buf_floats = ex; }
    a.TW = p.TW; a.TH = p.TH;
    a.tiles_x = (a.Wout + p.TW - 1) / p.TW; a.tiles_y = (a.Hout + p.TH - 1) / p.TH;
    a.TWin = (p.TW - 1) * c.stride + c.k;
    const int THin = (p.TH - 1) * c.stride + c.k;
    a.npix_in = a.TWin * THin;
    a.inv_TW = 1.0f / (float)p.TW; a.inv_TWin = 1.0f / (float)a.TWin;
    if (p.version == 4 && a.up_c) { a.inv_TW = 1.0f / (float)c.Win; a.inv_TWin = 1.0f / (float)c.Hin; }   // v4 has no other use for them
    // plans count staged channels in 4-byte units; the fp16 kernels stage twice as many channels in the same bytes
    a.ck = half ? 2 * p.ck : p.ck; a.ldp = half ? a.ck + 8 : a.ck + 4;
    a.ck4_shift = (p.ck == 128 ? 5 : p.ck == 64 ? 4 : p.ck == 32 ? 3 : 2);
    const int WC = 4 / p.WP;
    out->fn = (const void*)fn;
    a.n_tiles_total = (int)((long)B * a.tiles_x * a.tiles_y);
    out->grid_x = (unsigned)a.n_tiles_total;
    if (p.version == 4 || p.version == 5) {
        // persistent: exactly as many blocks as stay resident (asked of the runtime: registers, LDS), each walks tiles
        // blockIdx.x, + gridDim.x, ...; blocks that had to queue behind others would leave CUs half empty at the end
        const int gy = std::max(1, (a.n_ctiles + p.CT * (4 / p.WP) - 1) / (p.CT * (4 / p.WP)));
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, 256, p.lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = std::max(1, std::min(3, (int)((size_t)(160 * 1024) / std::max<size_t>(p.lds, 1))));
        }
        out->grid_x = std::min(out->grid_x, (unsigned)std::max(1, 256 * per_cu / gy));
    }
    if (p.version == 2) {      // persistent: as many blocks as stay resident (LDS-limited), each loops over tiles
        const int per_cu = std::max(1, std::min(6, (int)((size_t)(160 * 1024) / std::max<size_t>(p.lds, 1))));
        const unsigned cap = (unsigned)std::max(1, 256 * per_cu / (int)std::max(1, (a.n_ctiles + p.CT * (4 / p.WP) - 1) / (p.CT * (4 / p.WP))));
        out->grid_x = std::min(out->grid_x, cap);
    }
    out->grid_y = (unsigned)((a.n_ctiles + p.CT * WC - 1) / (p.CT * WC));
    out->lds = p.lds;
    out->a = a;
    out->CT = p.CT; out->WP = p.WP; out->version = p.version;
    out->PT = p.version == 3 ? p.buf_floats : (p.PT ? p.PT : (p.CT == 5 ? 3 : 4)); out->threads = p.version == 2 ? 320 : 256;
    if (p.version == 3) {          // streaming 1x1: block = 4 waves x PT pixel tiles, grid.y over cout blocks of CT tiles
        const int PT = p.buf_floats;
        out->grid_x = (unsigned)((a.Wout + 4 * PT * 16 - 1) / (4 * PT * 16));
        out->grid_y = (unsigned)((a.n_ctiles + p.CT - 1) / p.CT);
        out->lds = 0;
        out->a.tiles_x = (int)out->grid_x;
    }
    out->flops = 2.0 * c.B * c.Hout * c.Wout * (double)c.Cout * c.Cin * c.k * c.k;
    return nullptr;
}

// all candidate launches for one conv, best static guess first
const char* plan_conv_candidates(const ConvArgs& c, std::vector<ConvLaunch>* out) {
    if (const char* e = check_args(c)) return e;
    const int H = c.k == 1 ? 1 : c.Hout, W = c.k == 1 ? c.B * c.Hout * c.Wout : c.Wout;
    const bool half = c.dtype == 1;
    const std::vector<Plan> plans = enumerate_plans(H, W, (c.Cout + 15) / 16, half ? (c.Cin + 1) / 2 : c.Cin, c.k, c.stride,
                                                    c.zeros != nullptr, half);
    if (plans.empty()) return "conv: no launch plan fits in LDS";
    for (const Plan& p : plans) {
        if (c.src2 && p.version != 4) continue;       // upsample-on-read exists in the v4 kernels only
        ConvLaunch l{};
        if (const char* e = build_launch(c, p, &l)) return e;
        out->push_back(l);
    }
    if (out->empty()) return "conv: no launch plan supports the fused upsample";
    return nullptr;
}

const char* plan_conv(const ConvArgs& c, ConvLaunch* out) {
    std::vector<ConvLaunch> v;
    if (const char* e = plan_conv_candidates(c, &v)) return e;
    *out = v[0];
    return nullptr;
}

const char* run_conv(const ConvLaunch& l, hipStream_t st) {
    hipLaunchKernelGGL((KernelFn)l.fn, dim3(l.grid_x, l.grid_y), dim3(l.threads), l.lds, st, l.a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace mi355
