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
//
// This header holds the DEVICE code of the fp32 kernels; the instances the planner can pick are compiled in
// conv_f32_k3s1.hip / conv_f32_k3s2.hip / conv_f32_k1.hip / conv_f32_pipe.hip (four translation units: they build in
// parallel) and looked up through the pick_* functions of conv_f32_inst.h.  Host side (planner, weight packing):
// conv_plan.hip.
#pragma once
#include "common.h"
#include "detmath.h"
#include <type_traits>
#ifndef MI355_CONV_WAVES
#define MI355_CONV_WAVES 3   // min waves per SIMD the register allocator must leave room for (4 blocks of 256 threads / CU = 4)
#endif

#pragma clang fp contract(off)

namespace mi355 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) { return det_silu(v); }
// act == 2 (mi355_opts.fast_act, tolerance mode): x * rcp(1 + 2^(-x log2 e)) on the hardware transcendental units (v_exp_f32 / v_rcp_f32, 8 issue
// cycles each) -- 5 vector instructions per output instead of the canonical form's 24, ~1e-7 relative error, NOT reproducible on a CPU.
// The branch is block-uniform (act is a kernel argument).
__device__ __forceinline__ float silu_fast_f(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
    if (act == 2) { v[0] = silu_fast_f(v[0]); v[1] = silu_fast_f(v[1]); v[2] = silu_fast_f(v[2]); v[3] = silu_fast_f(v[3]); }
    else if (act) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
    return v;
}

// bias + SiLU (+ residual) and the 16-byte stores: a lane holds 4 consecutive couts of one pixel per tile.
// Two passes: all the ALU work first (16 independent SiLU chains per lane interleave freely), then the stores back to
// back.
// KA: the argument block's type -- ConvKArgs (a kernel's own by-value parameter) or a constant-address-space view of one (a
// member of a grouped launch reads its block from the kernarg segment on demand: conv_f32_group.hip)
template <int STRIDE, int PT, int CT, int WP, class KA>
__device__ __forceinline__ void conv_epilogue(const KA& a, f32x4 (&acc)[CT][PT], const f32x4 (&bias4)[CT], int lane, int wp,
                                              int ct0, int b, int oy0, int ox0, int npix, int act) {
    if (act == 2) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                f32x4 v = acc[ct][pt] + bias4[ct];
                v[0] = silu_fast_f(v[0]); v[1] = silu_fast_f(v[1]); v[2] = silu_fast_f(v[2]); v[3] = silu_fast_f(v[3]);
                acc[ct][pt] = v;
            }
    } else if (act) {
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
    // Stores (and residual loads) go through buffer descriptors over THIS image of the destination / residual slice: the lane
    // part of the address (pixel, k-group) is one 32-bit offset per pixel tile, the cout tile is the instruction's scalar
    // offset -- no 64-bit vector arithmetic per store; lanes outside the tile / image get an offset past num_records, which
    // drops the store (and reads zeros).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dst + (size_t)b * (size_t)a.img_dst), 0, (int)((unsigned)a.img_dst * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res + (size_t)b * (size_t)a.img_res : a.dst), 0,
                                                                         a.res ? (int)((unsigned)a.img_res * 4u) : 0, 0x00020000);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wp * PT + pt) * 16 + (lane & 15);
        const int pp = p < npix ? p : 0;
        const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
        const int lx = pp - __mul24(ly, a.TW);
        const int oy = oy0 + ly, ox = ox0 + lx;
        const bool ok = (p < npix) && (oy < a.Hout) && (ox < a.Wout);
        const int pix = __mul24(oy, a.Wout) + ox;
        const unsigned g16 = (unsigned)(lane >> 4) * 16u;
        const unsigned dvo = ok ? (unsigned)__mul24(pix, a.dst_cs) * 4u + g16 : 0x80000000u;
        const unsigned rvo = ok ? (unsigned)__mul24(pix, a.res_cs) * 4u + g16 : 0x80000000u;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c0t = (ct0 + ct) * 16;                             // wave-uniform
            if (c0t >= a.Cout) continue;
            f32x4 v = acc[ct][pt];
            if (c0t + 16 <= a.Cout) {                                    // whole cout tile: 16-byte accesses
                if (a.res) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)rvo, c0t * 4, 0));
                buffer_store_b128(__builtin_bit_cast(u32x4, v), drs, (int)dvo, c0t * 4);
            } else {                                                     // ragged last tile (Cout % 16 != 0): dword accesses
                const int c = c0t + (lane >> 4) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = c + i < a.Cout;                      // channels beyond Cout: offset past num_records
                    float r = v[i];
                    if (a.res) r += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (int)(in ? rvo + 4u * i : 0x80000000u), c0t * 4, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), drs, (int)(in ? dvo + 4u * i : 0x80000000u), c0t * 4, 0);
                }
            }
        }
    }
}

// One tile's worth of output for ONE cout tile: bias + activation (+ residual) and the 16-byte store (scalar tail when the
// channel count is not a multiple of 4).  Used by the fused second stage and the split-K finalizer.
__device__ __forceinline__ void store_tile(f32x4 v, const float* bias, int act, const float* res, int res_cs, float* dst, int dst_cs,
                                           int cout, int ctile, int lane, size_t po, bool ok) {
    const int c = ctile * 16 + (lane >> 4) * 4;
    if (!ok || c >= cout) return;
    v += *(const f32x4*)(bias + c);
    v = act4(v, act);
    float* d = dst + po * dst_cs + c;
    if (c + 3 < cout) {
        if (res) v += *(const f32x4*)(res + po * res_cs + c);
        *(f32x4*)d = v;
    } else {
        for (int i = 0; i < 4 && c + i < cout; ++i) {
            float r = v[i];
            if (res) r += res[po * res_cs + c + i];
            d[i] = r;
        }
    }
}

// F2 = a pointwise conv fused behind this one (Conv3x3 -> SiLU -> Conv1x1 whose only reader is that 1x1: the stride-2
// convs in front of every C2f, and the last two convs of every head branch): the block keeps its PT*WP*16 output pixels x ALL
// of the first conv's channels in LDS ([pixel][channel], exactly the image a 1x1 kernel would have staged from HBM), and
// the same four waves then run the 1x1 from there -- the intermediate tensor is never written to or read from HBM.
// Values, operation order and therefore bits are those of the two separate launches.
#ifndef MI355_V1_MINWAVES16
#define MI355_V1_MINWAVES16 1    // PT*CT >= 15: asking for two waves costs 16 spilled registers and gains nothing (A/B)
#endif
#ifndef MI355_V1_MINWAVES8
#define MI355_V1_MINWAVES8 3     // PT*CT == 8 instances: 168 registers, no spills, three waves per SIMD instead of two (+0.2-0.6 % A/B)
#endif
#ifndef MI355_V1_MINWAVES
#define MI355_V1_MINWAVES 4      // min waves per SIMD asked of the register allocator for the PT*CT == 4 instances (A/B with
                                 // tools/ab_build.sh: 1 -> 4 costs a 12-byte spill outside the loop, buys 2-3 % on the stride-2 layers, 0-1 % elsewhere)
#endif
template <int KS, int STRIDE, int PT, int CT, int WP, bool F2 = false, class KA = ConvKArgs>
__device__ __forceinline__ void conv_igemm_f32_body(const KA& a_in, float* lds, const BlockId& bid) {
    // What-if diagnostics, built only with -DMI355_F32_DIAG=1 (tools/ab_build.sh) and selected by MI355_F32_EXP (conv_plan.hip puts it in the high
    // bits of `act`): 1 = the halo tile is read from the zero page (the loads stay, their HBM traffic goes), 4 = every store gets an offset past the
    // descriptor's range (dropped)
    const KA& a = a_in;
#ifdef MI355_F32_DIAG
    const int diag = a.act >> 8, act1 = a.act & 255;
#else
    constexpr int diag = 0;
    const int act1 = a.act;
#endif
    constexpr int WC = 4 / WP;
    constexpr int TAPS = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction: with the wave index in an SGPR the cout-tile base, the weight pointers and their per-tap
    // offsets are scalar arithmetic, and the fragment loads take the (scalar base + lane offset) form -- no vector instruction
    // per load (the fp32 matrix instructions and the vector ALU share issue cycles: DESIGN.md 3.1)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, 0x7fffffff, 0x00020000);
    const int wp = wave % WP, wc = wave / WP;
    // tile decomposition on the scalar unit (FastDiv), per-lane products as 24-bit multiplies: the prologue's integer
    // divisions and 32/64-bit multiplies were ~50 slow vector instructions per block
    int t, cgrp0;
    xcd_work_item(t, cgrp0, FastDiv{a.fd_gy.ml, a.fd_gy.mh}, bid);
    const int tq = (int)fastdiv((unsigned)t, FastDiv{a.fd_tx.ml, a.fd_tx.mh}), tx = t - tq * a.tiles_x;
    const int b = (int)fastdiv((unsigned)tq, FastDiv{a.fd_ty.ml, a.fd_ty.mh}), ty = tq - b * a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
    const int npix = a.TW * a.TH;
    int xoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int p = (wp * PT + pt) * 16 + (lane & 15);
        p = p < npix ? p : 0;
        const int ly = (int)(((float)p + 0.5f) * a.inv_TW);
        const int lx = p - __mul24(ly, a.TW);
        xoff[pt] = __mul24(__mul24(ly * STRIDE, a.TWin) + lx * STRIDE, a.ldp) + (lane >> 4) * 4;
    }
    const float* srcb = a.src + (size_t)b * (size_t)a.img_src;
    const int ck4m = (a.ck >> 2) - 1;
    const int total_f4 = a.npix_in << a.ck4_shift;
    const int wstep = a.cib * 256;
    int xt[TAPS], wt[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) { xt[k] = ((k / KS) * a.TWin + (k % KS)) * a.ldp; wt[k] = k * wstep; }

    // A wave owns CT cout tiles at a time and walks a.cgroups such groups one after the other over the SAME staged input
    // (cgroups > 1 only when all of Cin is staged at once): narrow register tiles (acc + tot = 8 * CT * PT registers) at high
    // occupancy, yet the halo tile is fetched once per block for all CT * WC * cgroups cout tiles.
    for (int cg = 0; cg < a.cgroups; ++cg) {
        const int ct0 = ((cgrp0 * WC + wc) * a.cgroups + cg) * CT;
        // canonical accumulation: `acc` is the fma chain of ONE 16-channel block (all taps), started from +0;
        // `tot` is the running sum of the block partials in block order
        f32x4 acc[CT][PT], tot[CT][PT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) tot[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // cout tiles beyond the last one re-read the last tile's weights (their outputs are discarded by the epilogue):
        // no branch around the fragment loads, so hipcc keeps a counted s_waitcnt vmcnt(N) and the prefetch stays in flight
        // Weight fragments are fetched through a buffer descriptor over the packed weights: voffset = the lane's 16 bytes (one
        // VGPR for all loads), soffset = (cout tile, tap, block) in bytes (scalar arithmetic) -- no vector instruction per load.
        int wbase[CT];
        // the bias of this lane's 4 couts per cout tile is fetched now (the load's L2 latency hides under the staging) and
        // not in the epilogue, where it would sit on the block's critical path
        f32x4 bias4[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
            wbase[ct] = ctile * TAPS * a.cib * 256;                          // floats from a.wpk, wave-uniform
            bias4[ct] = *(const f32x4*)(a.bias + ctile * 16 + (lane >> 4) * 4);
        }

        for (int c0 = 0; c0 < a.Cin; c0 += a.ck) {
            if (cg == 0) {
                if (c0) __syncthreads();
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
                        const bool inb = idx < total_f4 && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win && c < a.cin4 && !(diag & 1);
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
                __syncthreads();
            }
            const int rem = a.Cin - c0;
            const int nkk = ((rem < a.ck ? rem : a.ck) + 15) >> 4;
            const int cib0 = c0 >> 4;
            // K loop over the chunk's 16-channel blocks.  Canonical order of one output: per block ONE fma
            // chain over (tap kh-major, MFMA step s, k-group g) started from +0 -- the block's first MFMA takes the constant 0
            // as its C operand --, then the block partial is added to the running total; blocks in ascending order.
            // The taps are unrolled at compile time (their LDS / weight offsets are loop-invariant), the only running scalars
            // are the current and the next block base, and nothing in the loop branches: the accumulators stay in place and
            // every load is unconditional, so hipcc keeps counted s_waitcnt vmcnt(N) / lgkmcnt(N).  Fragments travel through
            // register rings: weights (L2, 500+ cycles) RW - 1 steps ahead, pixels (LDS) one step ahead; loads past the
            // chunk's last block re-read it.
            constexpr int RX = TAPS > 1 ? 2 : 3;                       // 3x3: two pixel sets; the odd tap count is squared up by
                                                                       // one register copy per block (below)
            constexpr int RW = (TAPS > 1 && PT * CT <= 2) ? 9 : 3;     // small wave tiles: a step is only 128-256 MFMA cycles
            static_assert(TAPS == 1 || TAPS % RW == 0, "weight ring slots must line up across blocks");
            const int klast = nkk - 1;
            f32x4 wf[RW][CT], xf[RX][PT];
            auto opaque = [](int v) { asm volatile("" : "+s"(v)); return v; };     // keeps a scalar sum out of LICM's hands
            auto load_w = [&](f32x4* w, int off) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    w[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)lane16, (wbase[ct] + off) * 4, 0));
            };
            auto load_x = [&](f32x4* x, int off) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) x[pt] = *(const f32x4*)__builtin_assume_aligned(lds + xoff[pt] + off, 16);
            };
            auto mma = [&](const f32x4* w, const f32x4* x, bool first) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][s], x[pt][s],
                                                                               (first && s == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[ct][pt], 0, 0, 0);
            };
            auto bank = [&]() {                                      // end of a block: its partial joins the total
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) tot[ct][pt] += acc[ct][pt];
            };
            if constexpr (TAPS > 1) {
                // prologue: weight steps 0 .. RW-2 and pixel step 0 of the first block (a block has TAPS >= RW - 1 steps)
#pragma unroll
                for (int k = 0; k < RW - 1; ++k) load_w(wf[k], cib0 * 256 + wt[k]);
                load_x(xf[0], xt[0]);
                for (int kb = 0; kb < nkk; ++kb) {
                    const int kn = kb < klast ? kb + 1 : klast;
                    const int wk = opaque((cib0 + kb) * 256), wkn = opaque((cib0 + kn) * 256);
                    const int xk = opaque(kb * 16), xkn = opaque(kn * 16);
#pragma unroll
                    for (int k = 0; k < TAPS; ++k) {
                        constexpr int AW = RW - 1;
                        load_w(wf[(k + AW) % RW], opaque(((k + AW) >= TAPS ? wkn : wk) + wt[(k + AW) % TAPS]));
                        load_x(xf[(k + 1) % RX], opaque(((k + 1) >= TAPS ? xkn : xk) + xt[(k + 1) % TAPS]));
                        __builtin_amdgcn_sched_barrier(0);
                        mma(wf[k % RW], xf[k % RX], k == 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    bank();
                    if constexpr (TAPS % RX != 0) {                  // the next block's tap 0 landed in slot TAPS % RX: move it to slot 0
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt) xf[0][pt] = xf[TAPS % RX][pt];
                    }
                }
            } else {
                // pointwise: a step is a block; RX blocks per trip so that the ring slots are compile-time
                auto koff = [&](int kb) { return kb < klast ? kb : klast; };
                load_w(wf[0], (cib0 + koff(0)) * 256);
                load_w(wf[1], (cib0 + koff(1)) * 256);
                load_x(xf[0], 0);
                int kb0 = 0;
                for (; kb0 + RX <= nkk; kb0 += RX) {              // whole trips: no guard, no branch inside
#pragma unroll
                    for (int k = 0; k < RX; ++k) {
                        load_w(wf[(k + 2) % RW], opaque((cib0 + koff(kb0 + k + 2)) * 256));
                        load_x(xf[(k + 1) % RX], opaque(koff(kb0 + k + 1) * 16));
                        __builtin_amdgcn_sched_barrier(0);
                        mma(wf[k % RW], xf[k % RX], true);
                        bank();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (kb0 < nkk) {                                  // ragged tail trip
#pragma unroll
                    for (int k = 0; k < RX; ++k) {
                        load_w(wf[(k + 2) % RW], opaque((cib0 + koff(kb0 + k + 2)) * 256));
                        load_x(xf[(k + 1) % RX], opaque(koff(kb0 + k + 1) * 16));
                        __builtin_amdgcn_sched_barrier(0);
                        if (kb0 + k < nkk) { mma(wf[k % RW], xf[k % RX], true); bank(); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }

        if constexpr (!F2) {
            conv_epilogue<STRIDE, PT, CT, WP, KA>(a, tot, bias4, lane, wp, ct0, b, oy0, ox0, (diag & 4) ? 0 : npix, act1);
        } else {
            // first conv's output (bias + activation applied: the value the unfused launch would have stored) -> LDS image
            // [pixel][channel] with pixel stride ldp2, behind the halo tile; padded cout tiles (all-zero weights and bias)
            // give exact zeros, which is what the second conv's padded k-blocks expect
            // The first conv may carry the Bottleneck's residual (C2f / C3 with shortcut): y = silu(tot + bias) + residual, read
            // through a descriptor over this image of the residual slice (pixels outside the tile / image read zeros).
            float* y1 = lds + a.lds_buf_floats;
            if (a.lds_buf_floats == 0) __syncthreads();          // the image takes the halo tile's place: every wave is done reading the tile
            const __amdgpu_buffer_rsrc_t rrs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res + (size_t)b * (size_t)a.img_res : a.dst), 0,
                                                                                  a.res ? (int)((unsigned)a.img_res * 4u) : 0, 0x00020000);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                unsigned rvo = 0x80000000u;
                if (a.res) {
                    const int p = (wp * PT + pt) * 16 + (lane & 15);
                    const int pp = p < npix ? p : 0;
                    const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
                    const int lx = pp - __mul24(ly, a.TW);
                    const int oy = oy0 + ly, ox = ox0 + lx;
                    if ((p < npix) && (oy < a.Hout) && (ox < a.Wout)) rvo = (unsigned)__mul24(__mul24(oy, a.Wout) + ox, a.res_cs) * 4u + (unsigned)(lane >> 4) * 16u;
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (ct0 + ct >= a.n_ctiles) continue;
                    f32x4 v = tot[ct][pt] + bias4[ct];
                    v = act4(v, act1);
                    if (a.res) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs1, (int)rvo, (ct0 + ct) * 64, 0));
                    *(f32x4*)(y1 + ((wp * PT + pt) * 16 + (lane & 15)) * a.ldp2 + (ct0 + ct) * 16 + (lane >> 4) * 4) = v;
                }
            }
        }
    }
    if constexpr (F2) {
        __syncthreads();
        // ---- second stage: the 1x1 conv over the LDS image; wave (wp, wc) keeps its pixel tiles and takes cout tiles
        // wc, wc + WC, ... of the second conv.  Canonical order as everywhere: per 16-channel block a chain from +0.
        const float* y1 = lds + a.lds_buf_floats;
        // addressing as in conv_epilogue: one image of the destination slice behind a buffer descriptor, 32-bit lane offsets
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int img2 = a.Hout * a.Wout * a.dst2_cs;
        const __amdgpu_buffer_rsrc_t drs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dst2 + (size_t)b * (size_t)img2), 0, (int)((unsigned)img2 * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t wrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, 0x7fffffff, 0x00020000);
        // The pointwise conv may read MORE input channels than the first conv produced (C2f.cv2 over cat(ys): the last
        // Bottleneck's output is the block's LDS image, the earlier slices of the concat buffer are the `lead` channels in
        // front of it): its first a.lead_cib k-blocks come straight from global memory in the MFMA B-operand layout (lane (p, g):
        // 16 bytes of pixel p, channels 16 kb + 4 g -- a pointwise conv needs no halo), the remaining ones from the LDS image.
        // Blocks in ascending order as in the stand-alone launch: the same chain per block, the same sum of partials.
        const int img_lead = a.Hout * a.Wout * a.lead_cs;
        const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.lead_cib ? a.lead + (size_t)b * (size_t)img_lead : a.dst2), 0,
                                                                             a.lead_cib ? (int)((unsigned)img_lead * 4u) : 0, 0x00020000);
        int x2off[PT]; unsigned dvo[PT], lvo[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int p = (wp * PT + pt) * 16 + (lane & 15);
            x2off[pt] = __mul24(p, a.ldp2) + (lane >> 4) * 4;
            const int pp = p < npix ? p : 0;
            const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
            const int lx = pp - __mul24(ly, a.TW);
            const int oy = oy0 + ly, ox = ox0 + lx;
            const bool ok = (p < npix) && (oy < a.Hout) && (ox < a.Wout) && !(diag & 4);
            const int pix = __mul24(oy, a.Wout) + ox;
            dvo[pt] = ok ? (unsigned)__mul24(pix, a.dst2_cs) * 4u + (unsigned)(lane >> 4) * 16u : 0x80000000u;
            lvo[pt] = ok ? (unsigned)__mul24(pix, a.lead_cs) * 4u + (unsigned)(lane >> 4) * 16u : 0x80000000u;
        }
        const int nlead = a.lead_cib;
        for (int ct2 = wc; ct2 < a.n_ctiles2; ct2 += WC) {
            const int wb = ct2 * a.cib2 * 256;                           // floats from a.w2, wave-uniform
            f32x4 tot2[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) tot2[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 w_cur = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs2, (int)lane16, wb * 4, 0));
            auto block = [&](const f32x4 (&x2)[PT], int cb) {           // one 16-channel block: chain from +0, partial added to the sum
                const int cn = cb + 1 < a.cib2 ? cb + 1 : cb;
                const f32x4 w_nxt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs2, (int)lane16, (wb + cn * 256) * 4, 0));
                f32x4 p2[PT];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        p2[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_cur[s], x2[pt][s], s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : p2[pt], 0, 0, 0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) tot2[pt] += p2[pt];
                w_cur = w_nxt;
            };
            if (nlead > 0) {
                // lead blocks: two register sets, the next block's pixels in flight while this one's MFMAs run (loads past the
                // last lead block re-read it: unconditional, so the waits stay counted)
                auto load_lead = [&](f32x4 (&x)[PT], int cb) {
                    const int k = cb < nlead ? cb : nlead - 1;
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) x[pt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)lvo[pt], k * 64, 0));
                };
                f32x4 xa[PT], xb[PT];
                load_lead(xa, 0);
                int cb = 0;
                for (; cb + 2 <= nlead; cb += 2) {
                    load_lead(xb, cb + 1);
                    block(xa, cb);
                    load_lead(xa, cb + 2);
                    block(xb, cb + 1);
                }
                if (cb < nlead) block(xa, cb);
            }
            for (int cb = nlead; cb < a.cib2; ++cb) {
                f32x4 x2[PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) x2[pt] = *(const f32x4*)__builtin_assume_aligned(y1 + x2off[pt] + (cb - nlead) * 16, 16);
                block(x2, cb);
            }
            const int c0t = ct2 * 16, c = c0t + (lane >> 4) * 4;
            const f32x4 bias2 = *(const f32x4*)(a.bias2 + c);            // the bias array is padded to whole cout tiles
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                f32x4 v = tot2[pt] + bias2;
                v = act4(v, a.act2);
                if (c0t + 16 <= a.Cout2) {
                    buffer_store_b128(__builtin_bit_cast(u32x4, v), drs2, (int)dvo[pt], c0t * 4);
                } else {                                                 // ragged last cout tile: dword stores
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = v[i];                            // (bit_cast of the vector-element lvalue itself reads element 0)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), drs2,
                                                              (int)(c + i < a.Cout2 ? dvo[pt] + 4u * i : 0x80000000u), c0t * 4, 0);
                    }
                }
            }
        }
    }
}

template <int KS, int STRIDE, int PT, int CT, int WP, bool F2 = false>
__global__ __launch_bounds__(256, (PT * CT == 4 ? MI355_V1_MINWAVES : PT * CT == 8 ? MI355_V1_MINWAVES8 : PT * CT >= 15 ? MI355_V1_MINWAVES16 : 1)) void conv_igemm_f32(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv_igemm_f32_body<KS, STRIDE, PT, CT, WP, F2>(a, lds, MI355_BLOCK_ID());
}

// ---------------------------------------------------------------------------------------------- v6 (3x3, split K)
// Latency-bound launches (batch 1: a 20x20 map is 25 pixel tiles, a 3x3 conv over 256 channels a chain of 576 MFMAs per
// accumulator): the four waves of a block work on the SAME PT x CT tiles and split the 16-channel blocks between them
// (wave w takes blocks w, w + 4, ... of every staged chunk), so a wave's serial MFMA chain is a quarter as long.  This is
// bit-exact BECAUSE the canonical order is blocked: a block partial is one chain from +0 whoever computes it; the partials
// go to LDS (1 KiB per tile and block) and are then added in ascending block order, exactly as the one-wave kernels do in
// registers.  LDS = [halo tile of one chunk | cib x CT x PT partial tiles].
template <int KS, int STRIDE, int PT, int CT, class KA = ConvKArgs>
__device__ __forceinline__ void conv_splitk_f32_body(const KA& a, float* lds, const BlockId& bid) {
    constexpr int TAPS = KS * KS;
    static_assert(TAPS == 9, "split-K kernel is written for 3x3 convs");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform by construction: lets the K offsets live in SGPRs
    int t, cgrp0;
    xcd_work_item(t, cgrp0, bid);
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
    const int ct0 = cgrp0 * CT;
    const int npix = a.TW * a.TH;
    float* part = lds + a.lds_buf_floats;                       // [cib][CT][PT][64 lanes][4]
    int xoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int p = pt * 16 + (lane & 15);
        p = p < npix ? p : 0;
        const int ly = (int)(((float)p + 0.5f) * a.inv_TW);
        const int lx = p - ly * a.TW;
        xoff[pt] = ((ly * STRIDE) * a.TWin + lx * STRIDE) * a.ldp + (lane >> 4) * 4;
    }
    const float* srcb = a.src + (size_t)b * a.Hin * a.Win * a.src_cs;
    const float* wbase[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = a.wpk + (size_t)ctile * TAPS * a.cib * 256 + lane * 4;
    }
    const int ck4m = (a.ck >> 2) - 1;
    const int total_f4 = a.npix_in << a.ck4_shift;
    const int wstep = a.cib * 256;
    int xt[TAPS], wt[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) { xt[k] = ((k / KS) * a.TWin + (k % KS)) * a.ldp; wt[k] = k * wstep; }

    for (int c0 = 0; c0 < a.Cin; c0 += a.ck) {
        if (c0) __syncthreads();
        for (int base = 0; base < total_f4; base += 8 * 256) {  // stage the halo tile of this chunk (as in conv_igemm_f32)
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
        __syncthreads();
        const int rem = a.Cin - c0;
        const int nkk = ((rem < a.ck ? rem : a.ck) + 15) >> 4;
        const int cib0 = c0 >> 4;
        const int n_my = wave < nkk ? (nkk - wave + 3) >> 2 : 0;   // this wave's blocks: wave, wave + 4, ...
        if (n_my > 0) {
            constexpr int RW = TAPS, AW = RW - 1;                   // a whole block of weight fragments in flight
            f32x4 wf[RW][CT], xf[2][PT], acc[CT][PT];
            auto opaque = [](int v) { asm volatile("" : "+s"(v)); return v; };
            auto load_w = [&](f32x4* w, int off) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) w[ct] = *(const f32x4*)(wbase[ct] + off);
            };
            auto load_x = [&](f32x4* x, int off) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) x[pt] = *(const f32x4*)__builtin_assume_aligned(lds + xoff[pt] + off, 16);
            };
#pragma unroll
            for (int k = 0; k < AW; ++k) load_w(wf[k], (cib0 + wave) * 256 + wt[k]);
            load_x(xf[0], wave * 16 + xt[0]);
            for (int i = 0; i < n_my; ++i) {
                const int kb = wave + 4 * i, kn = i + 1 < n_my ? kb + 4 : kb;
                const int wk = opaque((cib0 + kb) * 256), wkn = opaque((cib0 + kn) * 256);
                const int xk = opaque(kb * 16), xkn = opaque(kn * 16);
#pragma unroll
                for (int k = 0; k < TAPS; ++k) {
                    load_w(wf[(k + AW) % RW], opaque(((k + AW) >= TAPS ? wkn : wk) + wt[(k + AW) % TAPS]));
                    load_x(xf[(k + 1) & 1], opaque(((k + 1) >= TAPS ? xkn : xk) + xt[(k + 1) % TAPS]));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                            for (int pt = 0; pt < PT; ++pt)
                                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[k % RW][ct][s], xf[k & 1][pt][s],
                                                                                   (k == 0 && s == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[ct][pt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float* pb = part + (size_t)(cib0 + kb) * (CT * PT * 256) + lane * 4;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) *(f32x4*)(pb + (ct * PT + pt) * 256) = acc[ct][pt];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) xf[0][pt] = xf[1][pt];   // TAPS is odd: next block's tap 0 landed in slot 1
            }
        }
    }
    __syncthreads();
    // ---- finalize: tile i = ct * PT + pt goes to wave i % 4; partials added in ascending block order, then the epilogue
#pragma unroll
    for (int i = 0; i < CT * PT; ++i) {
        if ((i & 3) != wave) continue;
        const int ct = i / PT, pt = i % PT;
        f32x4 tot = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* pb = part + i * 256 + lane * 4;
        for (int cb = 0; cb < a.cib; ++cb) tot += *(const f32x4*)(pb + (size_t)cb * (CT * PT * 256));
        const int ctile = ct0 + ct;
        const int c = ctile * 16 + (lane >> 4) * 4;
        const int p = pt * 16 + (lane & 15);
        const int pp = p < npix ? p : 0;
        const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
        const int lx = pp - ly * a.TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        if (!((p < npix) && (oy < a.Hout) && (ox < a.Wout)) || ctile >= a.n_ctiles || c >= a.Cout) continue;
        f32x4 v = tot + *(const f32x4*)(a.bias + ctile * 16 + (lane >> 4) * 4);
        v = act4(v, a.act);
        const size_t po = ((size_t)b * a.Hout + oy) * a.Wout + ox;
        float* d = a.dst + po * a.dst_cs + c;
        if (c + 3 < a.Cout) {
            if (a.res) v += *(const f32x4*)(a.res + po * a.res_cs + c);
            *(f32x4*)d = v;
        } else {
            for (int j = 0; j < 4 && c + j < a.Cout; ++j) {
                float r = v[j];
                if (a.res) r += a.res[po * a.res_cs + c + j];
                d[j] = r;
            }
        }
    }
}

template <int KS, int STRIDE, int PT, int CT>
__global__ __launch_bounds__(256) void conv_splitk_f32(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv_splitk_f32_body<KS, STRIDE, PT, CT>(a, lds, MI355_BLOCK_ID());
}

// ---------------------------------------------------------------------------------------------- v3 (1x1 only)
// Pointwise convs have no tap reuse, so staging pixels through LDS buys nothing: here every wave streams its pixel
// fragments straight from global memory into the MFMA B-operand layout (lane (p, g) reads the 16 bytes of channels
// 4g..4g+3 of pixel p: 64 contiguous bytes per pixel per 16-channel block), with a 4-deep register prefetch ring for
// pixels (HBM latency) and weights (L2 latency).  No LDS, no barriers: waves drift apart and overlap each other's
// epilogues.  Same canonical accumulation order as v1 (per 16-channel block a chain from +0; partials summed in block order).
// UP (round 4): the nearest-2x upsample fused into the read side, as conv1x1_pipe_f32 has it -- the first a.up_c input channels (whole
// 16-channel blocks) are read from the half-resolution tensor a.src2 at (y >> 1, x >> 1) instead of from a.src; same values in the same
// order, so the same bits as the upsample kernel followed by the plain launch.  fd_tx / fd_ty divide by up_W / up_W * up_H.
template <int PT, int CT, class KA = ConvKArgs, bool UP = false>
__device__ __forceinline__ void conv1x1_stream_f32_body(const KA& a, const BlockId& bid) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: tile bases and weight offsets stay scalar
    const int g = lane >> 4;
    const int total = a.Wout;                                        // flattened pixels (Hout == 1)
    int pblk, cgrp0;
    xcd_work_item(pblk, cgrp0, bid);
    const int ct0 = cgrp0 * CT;
    // Every access goes through a buffer descriptor (32-bit lane offset + scalar offset, no 64-bit vector arithmetic; the fp32
    // matrix instructions share issue cycles with the vector ALU): source / destination / residual descriptors start at THIS
    // block's first pixel and end at the tensor's last one, so pixels beyond the end read zeros and their stores are dropped.
    constexpr int BP = 4 * PT * 16;                                  // pixels per block
    const int p0 = pblk * BP, left = total - p0 < BP ? total - p0 : BP;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src + (size_t)p0 * a.src_cs), 0, left * a.src_cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dst + (size_t)p0 * a.dst_cs), 0, left * a.dst_cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res + (size_t)p0 * a.res_cs : a.dst), 0,
                                                                         a.res ? left * a.res_cs * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned xvo[PT];
    int pl[PT];                                                      // pixel index inside the block
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        pl[pt] = (wave * PT + pt) * 16 + (lane & 15);
        xvo[pt] = (unsigned)(__mul24(pl[pt], a.src_cs) + 4 * g) * 4u;
    }
    unsigned xvo2[UP ? PT : 1];                                      // UP: byte offset of the pixel's half-resolution source inside a.src2
    __amdgpu_buffer_rsrc_t x2rs = xrs;
    const int up_blocks = UP ? a.up_c >> 4 : 0;
    if constexpr (UP) {
        const int W2 = a.up_W >> 1, H2 = a.up_H >> 1;
        x2rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src2, 0, (int)((unsigned)(total >> 2) * (unsigned)a.src2_cs * 4u), 0x00020000);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const unsigned p = (unsigned)min(p0 + pl[pt], total - 1);
            const unsigned b = fastdiv(p, FastDiv{a.fd_ty.ml, a.fd_ty.mh}), r = p - b * (unsigned)(a.up_W * a.up_H);
            const unsigned y = fastdiv(r, FastDiv{a.fd_tx.ml, a.fd_tx.mh}), x = r - y * (unsigned)a.up_W;
            xvo2[pt] = (((b * (unsigned)H2 + (y >> 1)) * (unsigned)W2 + (x >> 1)) * (unsigned)a.src2_cs + 4u * (unsigned)g) * 4u;
        }
    }
    int wbase[CT];                                                   // floats from a.wpk
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = ctile * a.cib * 256;
    }
    f32x4 acc[CT][PT];                                               // running sum of the 16-channel block partials
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
        for (int ct = 0; ct < CT; ++ct)
            w[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)lane16, (wbase[ct] + l_it * 256) * 4, 0));
        const bool oob = tail_oob && (l_it == n_it - 1);
        if (UP && l_it < up_blocks) {                                // wave-uniform: this 16-channel block lives in the half-resolution tensor
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                x[pt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x2rs, (int)xvo2[UP ? pt : 0], l_it * 64, 0));
        } else {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                x[pt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(oob ? 0x80000000u : xvo[pt]), l_it * 64, 0));
        }
        if (l_it + 1 < n_it) ++l_it;                                 // clamp instead of guarding the loads
    };
    auto mma = [&](const f32x4* w, const f32x4* x) {              // one 16-channel block: chain from +0, partial added to the sum
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                f32x4 p = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][0], x[pt][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; ++s) p = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][s], x[pt][s], p, 0, 0, 0);
                acc[ct][pt] += p;
            }
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
        const unsigned dvo = (unsigned)__mul24(pl[pt], a.dst_cs) * 4u + (unsigned)g * 16u;
        const unsigned rvo = (unsigned)__mul24(pl[pt], a.res_cs) * 4u + (unsigned)g * 16u;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c0t = (ct0 + ct) * 16;                             // wave-uniform
            if (c0t >= a.Cout) continue;
            f32x4 v = acc[ct][pt] + bias4[ct];
            v = act4(v, a.act);
            if (c0t + 16 <= a.Cout) {
                if (a.res) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)rvo, c0t * 4, 0));
                buffer_store_b128(__builtin_bit_cast(u32x4, v), drs, (int)dvo, c0t * 4);
            } else {                                                     // ragged last cout tile: dword accesses
                const int c = c0t + g * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = c + i < a.Cout;
                    float r = v[i];
                    if (a.res) r += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (int)(in ? rvo + 4u * i : 0x80000000u), c0t * 4, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), drs, (int)(in ? dvo + 4u * i : 0x80000000u), c0t * 4, 0);
                }
            }
        }
    }
}

template <int PT, int CT>
__global__ __launch_bounds__(256, (PT * CT >= 16 ? 2 : PT * CT >= 8 ? 3 : 1)) void conv1x1_stream_f32(ConvKArgs a) {
    conv1x1_stream_f32_body<PT, CT>(a, MI355_BLOCK_ID());
}
template <int PT, int CT>
__global__ __launch_bounds__(256, (PT * CT >= 16 ? 2 : PT * CT >= 8 ? 3 : 1)) void conv1x1_stream_up_f32(ConvKArgs a) {
    conv1x1_stream_f32_body<PT, CT, ConvKArgs, true>(a, MI355_BLOCK_ID());
}

// ---------------------------------------------------------------------------------------------- v4 (1x1 only)
// Pointwise convs have a short K loop per staged chunk (Cin/16 steps), so in v1 the block spends as long waiting for its
// activation loads as it spends on MFMAs.  v4 is a PERSISTENT, software-pipelined form of v1 for k = 1: the unit of work
// is one (pixel tile, channel chunk) item; while the MFMAs of item i run from LDS, the global loads of item i+1 are
// already in flight into registers, and the stores of the previous tile drain behind them.  vmcnt retires in order, so
// the order of issue is what makes this work: the chunk's weight fragments (all of them: a 1x1 chunk has at most 4 k-blocks)
// are requested BEFORE the prefetch, hence waiting for them never waits for the prefetch.
// Same canonical accumulation order as v1 (per 16-channel block a chain over step s, k-group g; partials summed in block order): same bits.
#ifndef MI355_PIPE_MINWAVES
#define MI355_PIPE_MINWAVES 2      // 3 fits (168 registers) only with 28-41 spilled registers in the loop: no gain (A/B)
#endif
template <int PT, int CT, int WP, bool SINGLE, int NKK>   // SINGLE: Cin fits one chunk -> every item ends a tile; NKK: k-blocks per chunk
__global__ __launch_bounds__(256, (CT <= 2 ? (NKK <= 4 ? MI355_PIPE_MINWAVES : 2) : 1)) void conv1x1_pipe_f32(ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WC = 4 / WP, NV = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: cout-tile base and weight offsets stay scalar
    const unsigned lane16 = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, 0x7fffffff, 0x00020000);
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
    int wbase[CT];                                                   // floats from a.wpk, wave-uniform
    f32x4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
        wbase[ct] = ctile * a.cib * 256;
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
            for (int ct = 0; ct < CT; ++ct)
                w[kk][ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)lane16, (wbase[ct] + kb * 256) * 4, 0));
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
            if (kk < nkk) {                                    // one 16-channel block: chain from +0, partial added to the sum
                f32x4 p[CT][PT];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            p[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kk][ct][s], xf[kk & 1][pt][s],
                                                                             s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : p[ct][pt], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] += p[ct][pt];
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
                    v = act4(v, a.act);
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

}  // namespace mi355
