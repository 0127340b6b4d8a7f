// half=True path, device code (instances: conv_igemm_f16.hip and conv_f16_fused.hip): the same fused conv (1x1 / 3x3,
// stride 1 / 2) + bias + SiLU (+ residual) as conv_f32.h with
// fp16 STORAGE (activations and weights) and fp32 ARITHMETIC: v_mfma_f32_16x16x32_f16 accumulates in fp32, bias / SiLU /
// residual are applied in fp32 and the result is rounded to fp16 once (round-to-nearest-even) on the store.  The final
// 1x1 convs of the head write fp32, so the decode / NMS kernels are the fp32 ones.
//
// Replaces ultralytics' `half=True` predictor mode (engine/predictor.py: model.half(), im.half()) reached from the
// reference through model.py:38; BASELINE config 5 (YOLOv8m 1280x1280 fp16).
//
// Byte-for-byte the operand layouts are those of the fp32 kernel with "16 floats" replaced by "32 halfs": a k-block is
// 32 channels = 64 bytes per pixel, lane (p, g) reads the 16 bytes of channels 8g..8g+7 (one ds_read_b128 /
// global_load_dwordx4) and ONE MFMA consumes what four fp32 MFMAs did.  The sum over a k-block does not depend on how
// the instruction assigns k to lanes, because both operands use the same assignment.
#pragma once
#include "common.h"
#include "detmath.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace mi355 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// SiLU for the fp16-storage path: x * rcp(1 + 2^(-x log2 e)) on the hardware transcendental units (v_exp_f32 / v_rcp_f32,
// ~1e-7 relative error, far below the fp16 rounding that follows).  The fp32 path keeps det_silu (libm-free, IEEE
// division, ~35 VALU instructions) because it has to match the C oracle bit for bit; here that epilogue would cost
// two thirds of the K loop's MFMA time.
__device__ __forceinline__ float fast_silu(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896341f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// first cout of the 4 consecutive couts lane group g holds for cout tile `ctile` (see conv_f16_pairs)
__device__ __forceinline__ int tile_cout0(int ctile, int g, bool pairs) {
    return pairs ? ((ctile >> 1) * 32 + 8 * g + 4 * (ctile & 1)) : (ctile * 16 + 4 * g);
}

// bias + SiLU (+ residual) in fp32, one rounding to fp16 (or a plain fp32 store for the head outputs).  One 16x16 tile (or
// one pair of tiles) at a time, accumulator registers -> store: the 128 accumulators of a 128x64 wave tile never sit in
// arch VGPRs together.
template <int PT, int CT>
__device__ __forceinline__ void store_tiles_f16(const ConvKArgs& a, f32x4 (&acc)[CT][PT], const f32x4 (&bias4)[CT], int lane,
                                                int ct0, const size_t (&po)[PT], const bool (&ok)[PT]) {
    const bool pairs = conv_f16_pairs(a.Cout);
    const int g = lane >> 4;
    auto act = [&](f32x4 v, const f32x4& bias) {
        v += bias;
        if (a.act) { v[0] = fast_silu(v[0]); v[1] = fast_silu(v[1]); v[2] = fast_silu(v[2]); v[3] = fast_silu(v[3]); }
        return v;
    };
    if (!a.out_f32 && pairs && (CT % 2 == 0)) {
        // ct0 is even (a multiple of CT): tiles (ct, ct+1) are a pair -> 8 consecutive couts per lane, one 16-byte store
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int ct = 0; ct < CT; ct += 2) {
                const int c = tile_cout0(ct0 + ct, g, true);
                if (!ok[pt] || c >= a.Cout) continue;
                f32x4 v0 = act(acc[ct][pt], bias4[ct]), v1 = act(acc[ct + 1][pt], bias4[ct + 1]);
                if (a.res) {
                    const f16x8 rv = *(const f16x8*)((const _Float16*)a.res + po[pt] * a.res_cs + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v0[i] += (float)rv[i]; v1[i] += (float)rv[4 + i]; }
                }
                f16x8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o[i] = (_Float16)v0[i]; o[4 + i] = (_Float16)v1[i]; }
                *(f16x8*)((_Float16*)a.dst + po[pt] * a.dst_cs + c) = o;
            }
        return;
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = tile_cout0(ct0 + ct, g, pairs);
            if (!ok[pt] || c >= a.Cout) continue;
            f32x4 v = act(acc[ct][pt], bias4[ct]);
            if (a.out_f32) {
                float* d = a.dst + po[pt] * a.dst_cs + c;
                if (c + 3 < a.Cout) {
                    if (a.res) v += *(const f32x4*)(a.res + po[pt] * a.res_cs + c);
                    *(f32x4*)d = v;
                } else {
                    for (int i = 0; i < 4 && c + i < a.Cout; ++i) d[i] = v[i] + (a.res ? a.res[po[pt] * a.res_cs + c + i] : 0.f);
                }
            } else {
                _Float16* d = (_Float16*)a.dst + po[pt] * a.dst_cs + c;
                const _Float16* r = (const _Float16*)a.res + po[pt] * a.res_cs + c;
                if (c + 3 < a.Cout) {
                    if (a.res) { const f16x4 rv = *(const f16x4*)r; v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3]; }
                    f16x4 o;
                    o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
                    *(f16x4*)d = o;
                } else {
                    for (int i = 0; i < 4 && c + i < a.Cout; ++i) d[i] = (_Float16)(v[i] + (a.res ? (float)r[i] : 0.f));
                }
            }
        }
}

// Block = 256 threads = 4 waves, WP along pixels x WC along couts; a wave owns PT pixel tiles x CT cout tiles of 16x16.
// The halo tile is staged through LDS in chunks of a.ck channels (a.ck halfs, pixel stride a.ldp = ck + 8 halfs).
// F2 (3x3 convs): a pointwise conv fused behind this one.  The block must cover ALL couts of the 3x3 conv (grid.y == 1); its
// output tile -- bias + SiLU applied and rounded to fp16, i.e. the values the unfused launch would have stored -- goes to LDS
// as [pixel][channel] (what a 1x1 kernel would have staged from HBM) and the same four waves run the 1x1 from there.
template <int KS, int STRIDE, int PT, int CT, int WP, bool F2 = false>
__global__ __launch_bounds__(256, (PT == 8 ? (CT <= 3 ? 3 : 2) : 1)) void conv_igemm_f16(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds_h[];
    constexpr int WC = 4 / WP;
    constexpr int TAPS = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave index in an SGPR and weight fragments through a buffer descriptor (voffset = the lane's 16 bytes, soffset = (cout tile,
    // tap, k-block) on the scalar unit): no vector instruction and no 64-bit pointer pair per load (as conv_igemm_f32)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, 0x7fffffff, 0x00020000);
    const int wp = wave % WP, wc = wave / WP;
    int t, cgrp0;
    xcd_work_item(t, cgrp0);
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
    const int ct0 = (cgrp0 * WC + wc) * CT;
    const int npix = a.TW * a.TH;
    int xoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wp * PT + pt) * 16 + (lane & 15);
        const int pp = p < npix ? p : 0;
        const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
        const int lx = pp - ly * a.TW;
        xoff[pt] = ((ly * STRIDE) * a.TWin + lx * STRIDE) * a.ldp + (lane >> 4) * 8;
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const _Float16* srcb = (const _Float16*)a.src + (size_t)b * a.Hin * a.Win * a.src_cs;
    const _Float16* zeros = (const _Float16*)a.zeros;
    int wbase[CT];                                                                   // halfs from a.wpk, wave-uniform
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);   // padded cout tiles re-read the last one
        wbase[ct] = ctile * TAPS * a.cib * 512;
    }
    const int ck8m = (a.ck >> 3) - 1;
    const int total_v = a.npix_in << a.ck4_shift;          // 16-byte slots of one staged chunk

    // What-if diagnostics: built only with -DMI355_F16_DIAG=1, then MI355_F16_EXP selects
    // 1 skip staging, 2 weights not re-fetched, 4 no stores, 8 no weight loads in the K loop, 16 no LDS fragment reads
#ifdef MI355_F16_DIAG
    const int exp_flags = a.lds_buf_floats;
#else
    constexpr int exp_flags = 0;
#endif
    for (int c0 = 0; c0 < a.Cin; c0 += a.ck) {
        if (c0) __syncthreads();
        // stage the halo tile, channels [c0, c0+ck): 8 loads per thread in flight, zero page outside the image / beyond Cin
        for (int base = 0; base < ((exp_flags & 1) ? 0 : total_v); base += 8 * 256) {
            f16x8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                const int pix = idx >> a.ck4_shift, q = idx & ck8m;
                const int iy = (int)(((float)pix + 0.5f) * a.inv_TWin);
                const int ix = pix - iy * a.TWin;
                const int gy = iy0 + iy, gx = ix0 + ix, c = c0 + 8 * q;
                const bool inb = idx < total_v && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win && c < a.cin4;
                const _Float16* g = inb ? srcb + ((size_t)gy * a.Win + gx) * a.src_cs + c : zeros;
                v[u] = *(const f16x8*)g;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                if (idx < total_v) {
                    const int pix = idx >> a.ck4_shift, q = idx & ck8m;
                    *(f16x8*)(lds_h + pix * a.ldp + 8 * q) = v[u];
                }
            }
        }
        __syncthreads();
        const int rem = a.Cin - c0;
        const int nkk = ((rem < a.ck ? rem : a.ck) + 31) >> 5;
        const int cib0 = c0 >> 5;
        // K loop.  One f16 MFMA retires a 1-KiB fragment pair in 16 cycles, so a pipeline step (CT*PT MFMAs) is 8x shorter
        // than in the fp32 kernel and the per-step bookkeeping decides the rate: the taps are unrolled at compile time
        // (their LDS / weight offsets are loop-invariant scalars), the only running scalars are the current and the next
        // k-block base, and nothing in the loop branches.  Fragments travel through rings of R = 3 register sets:
        // weights (L2) are fetched two steps ahead, pixels (LDS) one step ahead; loads past the last k-block of the
        // chunk re-read the last one (unconditional loads keep the s_waitcnt counters counted).
        constexpr int R = 3;
        const int wstep = a.cib * 512;
        const int klast = nkk - 1;
        f16x8 wf[R][CT], xf[R][PT];
        auto opaque = [](int v) { asm volatile("" : "+s"(v)); return v; };     // keeps a scalar sum out of LICM's hands
        auto load_w = [&](f16x8* w, int off) {
            if (exp_flags & 8) return;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                w[ct] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)lane16, (wbase[ct] + off) * 2, 0));
        };
        auto load_x = [&](f16x8* x, int off) {
            if (exp_flags & 16) return;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) x[pt] = *(const f16x8*)__builtin_assume_aligned(lds_h + xoff[pt] + off, 16);
        };
        auto mma = [&](const f16x8* w, const f16x8* x) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ct], x[pt], acc[ct][pt], 0, 0, 0);
        };
        if constexpr (TAPS > 1 && PT == 8) {
            // 128-pixel wave tiles (128 x CT*16 outputs): the pixel fragments STREAM through a ring of XR registers sets, XD pixel
            // tiles ahead of the MFMAs that consume them, instead of a whole step's PT fragments being resident -- 16 VGPRs for
            // the pixel operand instead of 96, which is what lets CT = 3 / 4 (96 / 128 accumulator registers) run at two waves
            // per SIMD.  Per MFMA this wave tile asks LDS for 1 KiB / CT and L1 for 1 KiB / 8: at CT = 4 half of what
            // 64 x 64 tiles need from either.
            static_assert(TAPS % R == 0, "ring slots must line up across k-blocks");
            constexpr int XR = 4, XD = 2;
            static_assert(PT % XR == 0 && XD < XR, "ring slots must line up across steps");
            int xt[TAPS], wt[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) { xt[t] = ((t / KS) * a.TWin + (t % KS)) * a.ldp; wt[t] = t * wstep; }
            f16x8 xr[XR];
            auto load_x1 = [&](int slot, int pt, int off) {
                xr[slot] = *(const f16x8*)__builtin_assume_aligned(lds_h + xoff[pt] + off, 16);
            };
            load_w(wf[0], cib0 * 512 + wt[0]);
            load_w(wf[1], cib0 * 512 + wt[1]);
#pragma unroll
            for (int pt = 0; pt < XD; ++pt) load_x1(pt, pt, xt[0]);
            for (int kb = 0; kb < nkk; ++kb) {
                const int kn = kb < klast ? kb + 1 : klast;
                const int wk = opaque((cib0 + kb) * 512), wkn = opaque((cib0 + kn) * 512);
                const int xk = opaque(kb * 32), xkn = opaque(kn * 32);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int tw = (t + 2) % TAPS, tn = (t + 1) % TAPS;
                    load_w(wf[(t + 2) % R], opaque(((t + 2) >= TAPS ? wkn : wk) + wt[tw]));
                    const int cur = opaque(xk + xt[t]), nxt = opaque(((t + 1) >= TAPS ? xkn : xk) + xt[tn]);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        const int pf = pt + XD;
                        load_x1(pf % XR, pf % PT, pf >= PT ? nxt : cur);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t % R][ct], xr[pt % XR], acc[ct][pt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else if constexpr (TAPS > 1) {
            static_assert(TAPS % R == 0, "ring slots must line up across k-blocks");
            int xt[TAPS], wt[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) { xt[t] = ((t / KS) * a.TWin + (t % KS)) * a.ldp; wt[t] = t * wstep; }
            const bool frozen = (exp_flags & 2) != 0;
            // prologue: steps 0 and 1 of k-block 0
            load_w(wf[0], cib0 * 512 + wt[0]);
            load_w(wf[1], cib0 * 512 + wt[1]);
            load_x(xf[0], xt[0]);
            for (int kb = 0; kb < nkk; ++kb) {
                const int kn = kb < klast ? kb + 1 : klast;
                const int wk = opaque((cib0 + (frozen ? 0 : kb)) * 512), wkn = opaque((cib0 + (frozen ? 0 : kn)) * 512);
                const int xk = opaque(kb * 32), xkn = opaque(kn * 32);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int tw = (t + 2) % TAPS, tx = (t + 1) % TAPS;
                    load_w(wf[(t + 2) % R], opaque(((t + 2) >= TAPS ? wkn : wk) + wt[tw]));
                    load_x(xf[(t + 1) % R], opaque(((t + 1) >= TAPS ? xkn : xk) + xt[tx]));
                    __builtin_amdgcn_sched_barrier(0);
                    mma(wf[t % R], xf[t % R]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (PT == 8) {
            // pointwise, 128-pixel wave tiles: the streamed-pixel form of the loop above; a step is a k-block, R per trip
            constexpr int XR = 4, XD = 2;
            f16x8 xr[XR];
            auto load_x1 = [&](int slot, int pt, int off) {
                xr[slot] = *(const f16x8*)__builtin_assume_aligned(lds_h + xoff[pt] + off, 16);
            };
            auto koff = [&](int kb) { return kb < klast ? kb : klast; };
            load_w(wf[0], (cib0 + koff(0)) * 512);
            load_w(wf[1], (cib0 + koff(1)) * 512);
#pragma unroll
            for (int pt = 0; pt < XD; ++pt) load_x1(pt, pt, 0);
            auto trip = [&](int kb0, auto guarded) {
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    load_w(wf[(t + 2) % R], opaque((cib0 + koff(kb0 + t + 2)) * 512));
                    const int cur = opaque(koff(kb0 + t) * 32), nxt = opaque(koff(kb0 + t + 1) * 32);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        const int pf = pt + XD;
                        load_x1(pf % XR, pf % PT, pf >= PT ? nxt : cur);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!decltype(guarded)::value || kb0 + t < nkk) {
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct)
                                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t % R][ct], xr[pt % XR], acc[ct][pt], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            int kb0 = 0;
            for (; kb0 + R <= nkk; kb0 += R) trip(kb0, std::false_type{});
            if (kb0 < nkk) trip(kb0, std::true_type{});
        } else {
            // pointwise: a step is a k-block; R k-blocks per trip so that the ring slots are compile-time
            auto koff = [&](int kb) { return kb < klast ? kb : klast; };
            load_w(wf[0], (cib0 + koff(0)) * 512);
            load_w(wf[1], (cib0 + koff(1)) * 512);
            load_x(xf[0], 0);
            int kb0 = 0;
            for (; kb0 + R <= nkk; kb0 += R) {               // whole trips: no guard, no branch inside
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const int kw = koff(kb0 + t + 2), kx = koff(kb0 + t + 1);
                    load_w(wf[(t + 2) % R], opaque((cib0 + ((exp_flags & 2) ? 0 : kw)) * 512));
                    load_x(xf[(t + 1) % R], opaque(kx * 32));
                    __builtin_amdgcn_sched_barrier(0);
                    mma(wf[t % R], xf[t % R]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kb0 < nkk) {                                  // ragged tail trip
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const int kw = koff(kb0 + t + 2), kx = koff(kb0 + t + 1);
                    load_w(wf[(t + 2) % R], opaque((cib0 + ((exp_flags & 2) ? 0 : kw)) * 512));
                    load_x(xf[(t + 1) % R], opaque(kx * 32));
                    __builtin_amdgcn_sched_barrier(0);
                    if (kb0 + t < nkk) mma(wf[t % R], xf[t % R]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // epilogue operands are derived here, not before the K loop: 3 registers per pixel tile and 4 per cout tile less to carry
    size_t po[PT]; bool ok[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wp * PT + pt) * 16 + (lane & 15);
        const int pp = p < npix ? p : 0;
        const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
        const int lx = pp - ly * a.TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        ok[pt] = (p < npix) && (oy < a.Hout) && (ox < a.Wout);
        po[pt] = ((size_t)b * a.Hout + oy) * a.Wout + ox;
    }
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        bias4[ct] = *(const f32x4*)(a.bias + tile_cout0(ctile, lane >> 4, conv_f16_pairs(a.Cout)));
    }
    if (exp_flags & 4) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) ok[pt] = ok[pt] && acc[0][pt][0] == 12345.678f;
    }
    if constexpr (!F2) {
        store_tiles_f16<PT, CT>(a, acc, bias4, lane, ct0, po, ok);
    } else {
        // ---- first-stage image -> LDS behind the halo tile, fp16, pixel stride ldp2 = 32 * cib2 + 8 halfs ----
        _Float16* y1 = lds_h + 2 * a.lds_buf_floats;
        const int P = WP * PT * 16, g = lane >> 4;
        const int c1pad = a.cib2 * 32, c1t = a.n_ctiles * 16;
        // channels [16 * n_ctiles, 32 * cib2) of every pixel are zero: their weights are, and 0 * garbage must not be NaN
        const int q = (c1pad - c1t) >> 2;
        for (int i = tid; i < P * q; i += 256) {
            const int p = i / q, c = c1t + 4 * (i - p * q);
            *(f16x4*)(y1 + p * a.ldp2 + c) = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        }
        const bool pairs1 = conv_f16_pairs(a.Cout);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int p = (wp * PT + pt) * 16 + (lane & 15);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (ct0 + ct >= a.n_ctiles) continue;
                f32x4 v = acc[ct][pt] + bias4[ct];
                if (a.act) { v[0] = fast_silu(v[0]); v[1] = fast_silu(v[1]); v[2] = fast_silu(v[2]); v[3] = fast_silu(v[3]); }
                f16x4 o;
                o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
                *(f16x4*)(y1 + p * a.ldp2 + tile_cout0(ct0 + ct, g, pairs1)) = o;
            }
        }
        __syncthreads();
        // ---- pointwise stage: wave (wp, wc) keeps its PT pixel tiles and takes the cout-tile pairs wc, wc + WC, ... ----
        ConvKArgs a2 = a;
        a2.dst = a.dst2; a2.dst_cs = a.dst2_cs; a2.Cout = a.Cout2; a2.act = a.act2; a2.res = nullptr; a2.out_f32 = a.out2_f32;
        int x2[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) x2[pt] = ((wp * PT + pt) * 16 + (lane & 15)) * a.ldp2 + 8 * g;
        const bool pairs2 = conv_f16_pairs(a.Cout2);
        for (int c2 = 2 * wc; c2 < a.n_ctiles2; c2 += 2 * WC) {
            f32x4 acc2[2][PT];
            const _Float16* w2[2];
            f32x4 b2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ctile = (c2 + j) < a.n_ctiles2 ? (c2 + j) : (a.n_ctiles2 - 1);
                w2[j] = (const _Float16*)a.w2 + (size_t)ctile * a.cib2 * 512 + lane * 8;
                b2[j] = *(const f32x4*)(a.bias2 + tile_cout0(ctile, g, pairs2));
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc2[j][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll 2
            for (int kb = 0; kb < a.cib2; ++kb) {
                const f16x8 w0 = *(const f16x8*)(w2[0] + kb * 512), w1 = *(const f16x8*)(w2[1] + kb * 512);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const f16x8 xv = *(const f16x8*)__builtin_assume_aligned(y1 + x2[pt] + kb * 32, 16);
                    acc2[0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xv, acc2[0][pt], 0, 0, 0);
                    acc2[1][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xv, acc2[1][pt], 0, 0, 0);
                }
            }
            store_tiles_f16<PT, 2>(a2, acc2, b2, lane, c2, po, ok);
        }
    }
}

// Streaming pointwise conv (no LDS): every wave reads its pixel fragments straight from global memory in the MFMA
// B-operand layout (64 contiguous bytes per pixel per 32-channel block), D-deep register ring for pixels and weights.
template <int PT, int CT>
__global__ __launch_bounds__(256) void conv1x1_stream_f16(ConvKArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int total = a.Wout;                                        // flattened pixels (Hout == 1)
    int pblk, cgrp0;
    xcd_work_item(pblk, cgrp0);
    const int tile0 = (pblk * 4 + wave) * PT;
    const int ct0 = cgrp0 * CT;
    const _Float16* xbase[PT];
    size_t po[PT]; bool ok[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (tile0 + pt) * 16 + (lane & 15);
        ok[pt] = p < total;
        po[pt] = (size_t)(ok[pt] ? p : total - 1);                   // clamp: results of padded pixels are never stored
        xbase[pt] = (const _Float16*)a.src + po[pt] * a.src_cs + 8 * g;
    }
    const _Float16* zeros = (const _Float16*)a.zeros;
    const _Float16* wbase[CT];
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = (const _Float16*)a.wpk + (size_t)ctile * a.cib * 512 + lane * 8;
        bias4[ct] = *(const f32x4*)(a.bias + tile_cout0(ctile, g, conv_f16_pairs(a.Cout)));
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int D = 4;
    const int n_it = a.cib;
    // lanes whose 8 channels lie beyond round_up(Cin, 8) read the zero page (only possible in the last 32-channel block)
    const bool tail_oob = (n_it - 1) * 32 + 8 * g >= a.cin4;
    f16x8 wf[D][CT], xf[D][PT];
    int l_it = 0;
    auto load = [&](f16x8* w, f16x8* x) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f16x8*)(wbase[ct] + l_it * 512);
        const bool oob = tail_oob && (l_it == n_it - 1);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const _Float16* px = oob ? zeros : xbase[pt] + l_it * 32;
            x[pt] = *(const f16x8*)px;
        }
        if (l_it + 1 < n_it) ++l_it;
    };
    auto mma = [&](const f16x8* w, const f16x8* x) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ct], x[pt], acc[ct][pt], 0, 0, 0);
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
    store_tiles_f16<PT, CT>(a, acc, bias4, lane, ct0, po, ok);
}

// Persistent, software-pipelined pointwise kernel: the fp16 form of conv1x1_pipe_f32 (conv_igemm.hip, "v4").  Unit of work =
// one (pixel tile, channel chunk) item; the loads of item i+1 are in flight into registers while the MFMAs of item i run
// from LDS.  All weight fragments of a chunk (NKK k-blocks x CT) are requested before the prefetch, so the in-order
// vmcnt never makes the K loop wait for the prefetch.
template <int PT, int CT, int WP, bool SINGLE, int NKK>
__global__ __launch_bounds__(256) void conv1x1_pipe_f16(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds_h[];
    constexpr int WC = 4 / WP, NV = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP, g = lane >> 4;
    const int ct0 = (blockIdx.y * WC + wc) * CT;
    const int P = a.TW, total = a.Wout;
    const int sh = a.ck4_shift, ck8m = (a.ck >> 3) - 1, tile_v = P << sh;
    const int nst = SINGLE ? 1 : (a.Cin + a.ck - 1) / a.ck;
    const int n_tiles = a.n_tiles_total;
    const int my_tiles = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_items = my_tiles * nst;
    if (n_items == 0) return;
    const _Float16* srcp = (const _Float16*)a.src;
    const _Float16* zeros = (const _Float16*)a.zeros;

    auto prefetch = [&](int item, f16x8 (&v)[NV]) {      // item >= n_items: every lane reads the zero page
        const bool live = item < n_items;
        const int ti = SINGLE ? item : item / nst;
        const int p0 = ((int)blockIdx.x + ti * (int)gridDim.x) * P;
        const int c0 = SINGLE ? 0 : (item - ti * nst) * a.ck;
        // fused upsample (see conv1x1_pipe_f32): (image, row, column) of the tile's first pixel, once per item
        int ub = 0, uy = 0, ux = 0;
        if (a.up_c) { const int hw = a.up_W * a.up_H; ub = p0 / hw; const int r = p0 - ub * hw; uy = r / a.up_W; ux = r - uy * a.up_W; }
        int tq = tid; asm volatile("" : "+v"(tq));       // opaque: slot addresses recomputed per item instead of living in VGPRs
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = u * 256 + tq;
            const int pix = idx >> sh, q = idx & ck8m;
            const int p = p0 + pix, c = c0 + 8 * q;
            const bool inb = live && idx < tile_v && p < total && c < a.cin4;
            const _Float16* src = srcp + (size_t)p * a.src_cs + c;
            if (c < a.up_c) {
                const int xx = ux + pix;
                const int wr = (int)(((float)xx + 0.5f) * a.inv_TW);        // inv_TW = 1 / up_W for these launches
                const int x = xx - wr * a.up_W, yy = uy + wr;
                const int hr = (int)(((float)yy + 0.5f) * a.inv_TWin);      // inv_TWin = 1 / up_H; a tile may span several small images
                const int y = yy - hr * a.up_H, b = ub + hr;
                src = (const _Float16*)a.src2 + (((size_t)b * (a.up_H >> 1) + (y >> 1)) * (a.up_W >> 1) + (x >> 1)) * a.src2_cs + c;
            }
            if (!inb) src = zeros;
            v[u] = *(const f16x8*)src;
        }
    };
    auto commit = [&](const f16x8 (&v)[NV]) {
        int tq = tid; asm volatile("" : "+v"(tq));
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = u * 256 + tq;
            if (idx < tile_v) *(f16x8*)(lds_h + (idx >> sh) * a.ldp + 8 * (idx & ck8m)) = v[u];
        }
    };
    int xoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xoff[pt] = ((wp * PT + pt) * 16 + (lane & 15)) * a.ldp + 8 * g;
    const _Float16* wbase[CT];
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = (const _Float16*)a.wpk + (size_t)ctile * a.cib * 512 + lane * 8;
        bias4[ct] = *(const f32x4*)(a.bias + tile_cout0(ctile, g, conv_f16_pairs(a.Cout)));
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f16x8 pv[NV];
    prefetch(0, pv);
    commit(pv);
    __syncthreads();
    f16x8 w[NKK][CT];
    auto load_w = [&](int item) {
        const int st = SINGLE ? 0 : item % nst;
        const int cib0 = (st * a.ck) >> 5;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int kb = cib0 + kk < a.cib ? cib0 + kk : a.cib - 1;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) w[kk][ct] = *(const f16x8*)(wbase[ct] + kb * 512);
        }
    };
    load_w(0);
    prefetch(1, pv);
    for (int item = 0; item < n_items; ++item) {
        const int st = SINGLE ? 0 : item % nst;
        const int c0 = st * a.ck;
        const int rem = a.Cin - c0;
        const int nkk = ((rem < a.ck ? rem : a.ck) + 31) >> 5;
        f16x8 xf[2][PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xf[0][pt] = *(const f16x8*)__builtin_assume_aligned(lds_h + xoff[pt], 16);
        // a taken branch costs this loop ~10 % of a step, so full chunks (all but possibly the last one of a tile) run a
        // copy of the loop without the per-k-block guard
        auto kloop = [&](auto guarded) {
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int kn = decltype(guarded)::value ? (kk + 1 < nkk ? kk + 1 : nkk - 1) * 32 : (kk + 1 < NKK ? kk + 1 : NKK - 1) * 32;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) xf[(kk + 1) & 1][pt] = *(const f16x8*)__builtin_assume_aligned(lds_h + xoff[pt] + kn, 16);
                __builtin_amdgcn_sched_barrier(0);
                if (!decltype(guarded)::value || kk < nkk) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[kk][ct], xf[kk & 1][pt], acc[ct][pt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (nkk == NKK) kloop(std::false_type{}); else kloop(std::true_type{});
        __syncthreads();                                   // every wave is done reading this item's LDS image
        commit(pv);                                        // item + 1 (zeros after the last one)
        load_w(item + 1 < n_items ? item + 1 : item);      // requested before the stores and the next prefetch
        if (SINGLE || st == nst - 1) {
            const int ti = SINGLE ? item : item / nst;
            const int p0 = ((int)blockIdx.x + ti * (int)gridDim.x) * P;
            size_t po[PT]; bool ok[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int p = p0 + (wp * PT + pt) * 16 + (lane & 15);
                ok[pt] = p < total;
                po[pt] = (size_t)(ok[pt] ? p : 0);
            }
            store_tiles_f16<PT, CT>(a, acc, bias4, lane, ct0, po, ok);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        prefetch(item + 2, pv);
        __syncthreads();                                   // item + 1's LDS image is complete
    }
}

}  // namespace mi355
