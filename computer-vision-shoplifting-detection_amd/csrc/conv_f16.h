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

// ---- round 3: the same kernel on a vector-instruction diet --------------------------------------------------------------
// One f16 MFMA is 16 cycles, so at K = 432 .. 1728 a wave's whole K loop is 3.5k .. 14k cycles -- and the staging loop and the
// epilogue used to cost as much again in address arithmetic (PMC round 2: 3.0 - 7.5 vector instructions per MFMA, matrix pipe
// busy 20 - 42 %).  MI355_F16_DIET=1 (default): the halo tile is staged row by row through a buffer descriptor over this
// image (per load: one multiply-add, two compares, one select; lanes outside the image read zeros because their offset lies
// past num_records -- no zero page, no 64-bit pointer per load), the tile decomposition runs on the scalar unit, and the
// epilogue stores / residual loads go through descriptors with one 32-bit offset per pixel tile, branch-free.
#ifndef MI355_F16_DIET
#define MI355_F16_DIET 1
#endif
constexpr unsigned kOOB = 0x80000000u;       // byte offset past any descriptor's num_records (planner: images stay below 2^31 bytes)

typedef float f32x2 __attribute__((ext_vector_type(2)));
// SiLU of two values: the multiplies and the add as packed fp32 instructions, two transcendentals each
__device__ __forceinline__ f32x2 fast_silu2(f32x2 x) {
    const f32x2 t = x * (f32x2){-1.44269504088896341f, -1.44269504088896341f};
    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = e + (f32x2){1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return x * r;
}
__device__ __forceinline__ f32x4 bias_act4(f32x4 v, const f32x4& bias, int act) {
    v += bias;
    if (act) {
        const f32x2 lo = fast_silu2((f32x2){v[0], v[1]}), hi = fast_silu2((f32x2){v[2], v[3]});
        v = (f32x4){lo[0], lo[1], hi[0], hi[1]};
    }
    return v;
}

// Destination of an epilogue: one image of the destination (and residual) slice behind buffer descriptors.
struct OutF16 {
    __amdgpu_buffer_rsrc_t drs, rrs;
    int dst_cs, res_cs, Cout, act, out_f32; bool has_res;
    const float* dst; const float* res;      // image bases (ragged cout tiles take the pointer path)
};
__device__ __forceinline__ OutF16 make_out_f16(const float* dst, int dst_cs, int img_dst, const float* res, int res_cs, int img_res, int b,
                                               int Cout, int act, int out_f32) {
    OutF16 o;
    const unsigned esz = out_f32 ? 4u : 2u;
    const char* db = (const char*)dst + (size_t)b * (size_t)img_dst * esz;
    const char* rb = res ? (const char*)res + (size_t)b * (size_t)img_res * esz : db;
    o.drs = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (int)((unsigned)img_dst * esz), 0x00020000);
    o.rrs = __builtin_amdgcn_make_buffer_rsrc((void*)rb, 0, res ? (int)((unsigned)img_res * esz) : 0, 0x00020000);
    o.dst_cs = dst_cs; o.res_cs = res_cs; o.Cout = Cout; o.act = act; o.out_f32 = out_f32; o.has_res = res != nullptr;
    o.dst = (const float*)db; o.res = (const float*)rb;
    return o;
}

// pix[pt] = pixel index inside the image (row-major), or -1 for lanes whose pixel lies outside the tile / image.
template <int PT, int CT>
__device__ __forceinline__ void store_tiles_f16_v2(const OutF16& o, f32x4 (&acc)[CT][PT], const f32x4 (&bias4)[CT], int lane, int ct0,
                                                   const int (&pix)[PT]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const bool pairs = conv_f16_pairs(o.Cout);
    const unsigned g = (unsigned)lane >> 4;
    const int n_full = o.Cout >> 4;                       // cout tiles that are complete
    // last tile of a Cout that is not a multiple of 16 (the 51-channel keypoint branch, one-class models): element stores through
    // the same descriptors, channels beyond Cout dropped by the offset marker.  c = first cout of this lane's four.
    auto ragged = [&](f32x4 accv, const f32x4& bias, int c, unsigned dvo, unsigned rvo, int so) {
        const f32x4 v = bias_act4(accv, bias, o.act);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = c + i < o.Cout;
            if (o.out_f32) {
                float r = v[i];
                if (o.has_res) r += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(o.rrs, (int)(in ? rvo + 4u * i : kOOB), so, 0));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), o.drs, (int)(in ? dvo + 4u * i : kOOB), so, 0);
            } else {
                float r = v[i];
                if (o.has_res) r += (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(o.rrs, (int)(in ? rvo + 2u * i : kOOB), so, 0));
                const _Float16 h = (_Float16)r;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), o.drs, (int)(in ? dvo + 2u * i : kOOB), so, 0);
            }
        }
    };
    if (o.out_f32) {
        // fp32 destination (the head's final convs): tile t holds couts 16 t + 4 g (bytes 64 t + 16 g) or, in the paired row
        // order (Cout % 32 == 0), 32 (t >> 1) + 8 g + 4 (t & 1)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const unsigned lane_b = pairs ? g * 32u : g * 16u;
            const unsigned dvo = pix[pt] >= 0 ? (unsigned)__mul24(pix[pt], o.dst_cs) * 4u + lane_b : kOOB;
            const unsigned rvo = pix[pt] >= 0 ? (unsigned)__mul24(pix[pt], o.res_cs) * 4u + lane_b : kOOB;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int t = ct0 + ct;                                   // wave-uniform
                if (t * 16 >= o.Cout) continue;
                const int so = pairs ? (t >> 1) * 128 + (t & 1) * 16 : t * 64;
                if (t >= n_full) { ragged(acc[ct][pt], bias4[ct], t * 16 + 4 * (int)g, dvo, rvo, so); continue; }
                f32x4 v = bias_act4(acc[ct][pt], bias4[ct], o.act);
                if (o.has_res) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(o.rrs, (int)rvo, so, 0));
                buffer_store_b128(__builtin_bit_cast(u32x4, v), o.drs, (int)dvo, so);
            }
            __builtin_amdgcn_sched_barrier(0);      // one pixel tile at a time: the accumulators leave their registers tile by tile
        }
        return;
    }
    // Residual reads of PG pixel tiles are ALL issued before the first of them is used (round 4): the epilogue used to load each tile's
    // residual right where it was added, one memory latency per pixel tile and, in the persistent kernels, per unit -- the 48-channel
    // bottlenecks of config 5 took 244 us with the residual against 160 us without.  PG = all pixel tiles for small register tiles, 2 for
    // the large ones (a 16-byte residual per tile pair and pixel tile is 4 registers).
    constexpr int PG = (CT * PT <= 12) ? PT : (PT % 2 == 0 ? 2 : 1);
    static_assert(PT % PG == 0, "pixel tiles come in whole groups");
    if (pairs && (CT % 2 == 0)) {
        // ct0 is even: tiles (ct, ct + 1) are a pair -> 8 consecutive couts per lane, one 16-byte fp16 store; pair j at byte 64 j
#pragma unroll
        for (int pt0 = 0; pt0 < PT; pt0 += PG) {
            f16x8 rvp[PG][CT / 2];
            if (o.has_res) {
#pragma unroll
                for (int p = 0; p < PG; ++p) {
                    const unsigned rvo = pix[pt0 + p] >= 0 ? (unsigned)__mul24(pix[pt0 + p], o.res_cs) * 2u + g * 16u : kOOB;
#pragma unroll
                    for (int ct = 0; ct < CT; ct += 2)
                        rvp[p][ct / 2] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(o.rrs, (int)((ct0 + ct) * 16 < o.Cout ? rvo : kOOB),
                                                                                                         ((ct0 + ct) >> 1) * 64, 0));
                }
            }
#pragma unroll
            for (int p = 0; p < PG; ++p) {
                const int pt = pt0 + p;
                const unsigned dvo = pix[pt] >= 0 ? (unsigned)__mul24(pix[pt], o.dst_cs) * 2u + g * 16u : kOOB;
#pragma unroll
                for (int ct = 0; ct < CT; ct += 2) {
                    const int t = ct0 + ct;
                    if (t * 16 >= o.Cout) continue;                           // pairs: Cout % 32 == 0, so a pair is whole or absent
                    const f32x4 v0 = bias_act4(acc[ct][pt], bias4[ct], o.act), v1 = bias_act4(acc[ct + 1][pt], bias4[ct + 1], o.act);
                    f32x4 w0 = v0, w1 = v1;
                    if (o.has_res) {
                        const f16x8 rv = rvp[p][ct / 2];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { w0[i] += (float)rv[i]; w1[i] += (float)rv[4 + i]; }
                    }
                    f16x8 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { h[i] = (_Float16)w0[i]; h[4 + i] = (_Float16)w1[i]; }
                    buffer_store_b128(__builtin_bit_cast(u32x4, h), o.drs, (int)dvo, (t >> 1) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    // single tiles, 8-byte fp16 stores: tile t starts at cout 16 t (lane group at + 4 g), or -- paired layout, odd CT -- at
    // 32 (t >> 1) + 4 (t & 1) (lane group at + 8 g)
#pragma unroll
    for (int pt0 = 0; pt0 < PT; pt0 += PG) {
        const unsigned lane_b = pairs ? g * 16u : g * 8u;
        f16x4 rvs[PG][CT];
        if (o.has_res) {
#pragma unroll
            for (int p = 0; p < PG; ++p) {
                const unsigned rvo = pix[pt0 + p] >= 0 ? (unsigned)__mul24(pix[pt0 + p], o.res_cs) * 2u + lane_b : kOOB;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int t = ct0 + ct;
                    const int so = pairs ? (t >> 1) * 64 + (t & 1) * 8 : t * 32;
                    rvs[p][ct] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(o.rrs, (int)(t < n_full ? rvo : kOOB), so, 0));   // ragged tiles read their own
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PG; ++p) {
            const int pt = pt0 + p;
            const unsigned dvo = pix[pt] >= 0 ? (unsigned)__mul24(pix[pt], o.dst_cs) * 2u + lane_b : kOOB;
            const unsigned rvo = pix[pt] >= 0 ? (unsigned)__mul24(pix[pt], o.res_cs) * 2u + lane_b : kOOB;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int t = ct0 + ct;
                if (t * 16 >= o.Cout) continue;
                const int so = pairs ? (t >> 1) * 64 + (t & 1) * 8 : t * 32;   // bytes, wave-uniform
                if (t >= n_full) { ragged(acc[ct][pt], bias4[ct], t * 16 + 4 * (int)g, dvo, rvo, so); continue; }
                f32x4 v = bias_act4(acc[ct][pt], bias4[ct], o.act);
                if (o.has_res) {
                    const f16x4 rv = rvs[p][ct];
                    v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
                }
                f16x4 h;
                h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), o.drs, (int)dvo, so, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Stage the halo tile (3x3) or the pixel tile (1x1) of channels [c0, c0 + ck) into LDS as [pixel][ldp halfs].
// 3x3: 256 threads cover `st_rpi` whole rows of 16-byte slots per pass (or one row in `st_nseg` segments when a row has more
// than 256 slots); a thread keeps its column for all rows, so everything but the row term of its address is computed once.
// 1x1: a pass covers 256 >> sh pixels; a thread keeps its 8-channel slot and walks pixels at a constant byte stride.
template <int KS>
__device__ __forceinline__ void stage_tile_f16(const ConvKArgs& a, _Float16* lds_h, const __amdgpu_buffer_rsrc_t srs, int tid, int iy0, int ix0, int c0) {
    const int sh = a.ck4_shift, smask = (1 << sh) - 1;
    if constexpr (KS == 1) {
        const int pps = 256 >> sh;                                   // pixels per pass
        const int pix0 = tid >> sh, q = tid & smask;
        const bool okc = c0 + 8 * q < a.cin4;
        const unsigned vo0 = (unsigned)(__mul24(ix0 + pix0, a.src_cs) + 8 * q) * 2u;
        const unsigned vstep = (unsigned)__mul24(pps, a.src_cs) * 2u;
        const int l0 = __mul24(pix0, a.ldp) + 8 * q, lstep = __mul24(pps, a.ldp);
        const int left = min(a.npix_in, a.Win - ix0);                // pixels of this tile that exist
        const int npass = (a.npix_in + pps - 1) >> (8 - sh);
        for (int p0 = 0; p0 < npass; p0 += 8) {
            f16x8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pix = pix0 + (p0 + u) * pps;
                const unsigned vo = (okc && pix < left) ? vo0 + (unsigned)(p0 + u) * vstep : kOOB;
                v[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(srs, (int)vo, c0 * 2, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pix = pix0 + (p0 + u) * pps;
                if (pix < a.npix_in) *(f16x8*)(lds_h + l0 + (p0 + u) * lstep) = v[u];
            }
        }
    } else {
        const int row_slots = a.TWin << sh;
        const int rpi = a.st_rpi;
        const unsigned rowbytes = (unsigned)__mul24(a.Win, a.src_cs) * 2u;
        const int lrow = __mul24(a.TWin, a.ldp);
        for (int seg = 0; seg < a.st_nseg; ++seg) {
            int r_in = 0, col = seg * 256 + tid;
            if (rpi > 1) { r_in = (int)(((float)tid + 0.5f) * a.inv_row_slots); col = tid - __mul24(r_in, row_slots); }
            const bool colok = col < row_slots && r_in < rpi;
            const int ix = col >> sh, q = col & smask;
            const int gx = ix0 + ix;
            const bool xok = colok && (unsigned)gx < (unsigned)a.Win && c0 + 8 * q < a.cin4;
            const unsigned vbase = (unsigned)(__mul24(gx, a.src_cs) + 8 * q) * 2u;
            const int lbase = __mul24(__mul24(r_in, a.TWin) + ix, a.ldp) + 8 * q;
            for (int row0 = 0; row0 < a.THin; row0 += 8 * rpi) {
                f16x8 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = row0 + u * rpi + r_in, gy = iy0 + row;
                    const bool ok = xok && row < a.THin && (unsigned)gy < (unsigned)a.Hin;
                    const unsigned vo = ok ? vbase + __umul24((unsigned)gy, rowbytes) : kOOB;
                    v[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(srs, (int)vo, c0 * 2, 0));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = row0 + u * rpi + r_in;
                    if (colok && row < a.THin) *(f16x8*)(lds_h + lbase + __mul24(row0 + u * rpi, lrow)) = v[u];
                }
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
#if MI355_F16_DIET
    // tile decomposition on the scalar unit (as conv_igemm_f32)
    xcd_work_item(t, cgrp0, FastDiv{a.fd_gy.ml, a.fd_gy.mh}, MI355_BLOCK_ID());
    const int tq = (int)fastdiv((unsigned)t, FastDiv{a.fd_tx.ml, a.fd_tx.mh}), tx = t - tq * a.tiles_x;
    const int b = (int)fastdiv((unsigned)tq, FastDiv{a.fd_ty.ml, a.fd_ty.mh}), ty = tq - b * a.tiles_y;
#else
    xcd_work_item(t, cgrp0);
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
#endif
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
        const int lx = pp - __mul24(ly, a.TW);
        xoff[pt] = __mul24(__mul24(ly * STRIDE, a.TWin) + lx * STRIDE, a.ldp) + (lane >> 4) * 8;
    }
#if MI355_F16_DIET
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)a.src + (size_t)b * (size_t)a.img_src), 0,
                                                                         (int)((unsigned)a.img_src * 2u), 0x00020000);
#endif
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
#if MI355_F16_DIET
        if (!(exp_flags & 1)) stage_tile_f16<KS>(a, lds_h, srs, tid, iy0, ix0, c0);
        for (int base = 0; base < 0; base += 8 * 256) {
#else
        // stage the halo tile, channels [c0, c0+ck): 8 loads per thread in flight, zero page outside the image / beyond Cin
        for (int base = 0; base < ((exp_flags & 1) ? 0 : total_v); base += 8 * 256) {
#endif
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
    int pixi[PT];                                   // pixel index inside image b, -1 = no store (MI355_F16_DIET)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wp * PT + pt) * 16 + (lane & 15);
        const int pp = p < npix ? p : 0;
        const int ly = (int)(((float)pp + 0.5f) * a.inv_TW);
        const int lx = pp - __mul24(ly, a.TW);
        const int oy = oy0 + ly, ox = ox0 + lx;
        ok[pt] = (p < npix) && (oy < a.Hout) && (ox < a.Wout);
        pixi[pt] = ok[pt] ? __mul24(oy, a.Wout) + ox : -1;
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
#if MI355_F16_DIET
        if (exp_flags & 4) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) pixi[pt] = ok[pt] ? pixi[pt] : -1;
        }
        const OutF16 o = make_out_f16(a.dst, a.dst_cs, a.img_dst, a.res, a.res_cs, a.img_res, b, a.Cout, a.act, a.out_f32);
        store_tiles_f16_v2<PT, CT>(o, acc, bias4, lane, ct0, pixi);
#else
        store_tiles_f16<PT, CT>(a, acc, bias4, lane, ct0, po, ok);
#endif
    } else {
        // ---- first-stage image -> LDS behind the halo tile, fp16, pixel stride ldp2 = 32 * cib2 + 8 halfs ----
        _Float16* y1 = lds_h + 2 * a.lds_buf_floats;
        if (a.lds_buf_floats == 0) __syncthreads();              // the image takes the halo tile's place: every wave is done reading the tile
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
                const f32x4 v = bias_act4(acc[ct][pt], bias4[ct], a.act);
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
#if MI355_F16_DIET
        const OutF16 o2 = make_out_f16(a.dst2, a.dst2_cs, a.Hout * a.Wout * a.dst2_cs, nullptr, 0, 0, b, a.Cout2, a.act2, a.out2_f32);
#endif
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
#if MI355_F16_DIET
            store_tiles_f16_v2<PT, 2>(o2, acc2, b2, lane, c2, pixi);
#else
            store_tiles_f16<PT, 2>(a2, acc2, b2, lane, c2, po, ok);
#endif
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
