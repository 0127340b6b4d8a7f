// Bandwidth-bound kernels around the conv stack: the u8 stem conv, nearest 2x upsample, the SPPF max-pools
// and the letterbox resize/pad.  All NHWC, 16-byte accesses per lane wherever the layout allows.
#include "common.h"
#include "detmath.h"
#include <cstdlib>
#include <vector>

#pragma clang fp contract(off)

namespace mi355 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef MI355_STEM_DIAG
#define MI355_STEM_DIAG 0      // 1: stem3s2_u8_h reads what-if bits from StemArgs::variant >> 8 (A/B builds only)
#endif
#ifndef MI355_STEM_EXP
#define MI355_STEM_EXP 0       // what-if builds only (tools/ab_build.sh): 1 no SiLU, 2 no input conversion, 4 no stores, 8 no MFMAs
#endif
__device__ __forceinline__ float silu_m(float v) { return (MI355_STEM_EXP & 1) ? v : det_silu(v); }
// half=True path only (the value is rounded to fp16 next; same formulation as conv_igemm_f16.hip: 1e-7 relative on the
// hardware transcendental units instead of the 32-instruction bit-exact form)
__device__ __forceinline__ float silu_fast(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---------------------------------------------------------------------------------------------- stem
// model.0: Conv(3 -> Cout, k, s) on the letterboxed uint8 BGR frame.  Fuses the rest of
// engine/predictor.py:preprocess (BGR->RGB, uint8 -> float32, /255 via an exact 256-entry table) into the
// conv read, so the 4.9 MB/frame fp32 input tensor never exists.  The layer is HBM-write bound (16-48 floats out per
// 27 bytes in) and, because of its bit-exact SiLU, issue-bound.
constexpr int STEM_TO = 16;        // output tile 16 x 16 pixels per block

// An implicit GEMM on the fp32 matrix pipe with K = KS*KS*3 (27 or 108, padded to a multiple of 4) whose
// B operand is gathered from the LDS input tile through a per-lane offset table (im2col on the fly, one ds_read_b32 per
// MFMA operand).  k runs (kh, kw, byte channel B,G,R) ascending and v_mfma_f32_16x16x4_f32 chains its 4 k-values in
// order, so the sum is the fma chain of the oracle's det_stem -- bit for bit -- at 4-5x the rate of a vector-ALU direct
// conv (measured in round 1: ~15 TFLOP/s, below both the HBM and the matrix roof of this layer).
// Block = 256 threads = 4 waves, output tile 16 x 16: wave w owns pixel rows 4w..4w+3 (4 MFMA pixel tiles) x CT cout tiles.
// Input bytes are fetched as aligned dwords (4 pixels' worth of bytes per load) instead of one byte per lane.
template <int KS, int CT>
__global__ __launch_bounds__(256) void stem_mfma_u8(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sl[];
    constexpr int K = KS * KS * 3, NS = (K + 3) / 4, PT = 4;
    const int TIN = (STEM_TO - 1) * a.stride + KS;
    const int trow = TIN * 3;                                   // floats (= source bytes) per tile row
    float* lut = sl;
    float* tin = sl + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    lut[tid] = a.lut[tid];
    const int tiles_x = (a.Wout + STEM_TO - 1) / STEM_TO, tiles_y = (a.Hout + STEM_TO - 1) / STEM_TO;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * STEM_TO, ox0 = tx * STEM_TO;
    const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;
    const uint8_t* img = a.img + (size_t)b * a.H * a.W * 3;
    // A operand: lane (row r = cout, k-group g) holds W[cout][k = 4s + g] for every step s (zero beyond K / Cout);
    // byte channel cb (0=B,1=G,2=R) feeds model channel 2-cb (im[..., ::-1])
    float wa[NS][CT];
    int koff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + g;
        const int tap = k / 3, cb = k - tap * 3;
        const int kh = tap / KS, kw = tap - kh * KS;
        koff[s] = k < K ? (kh * TIN + kw) * 3 + cb : 0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int co = ct * 16 + (lane & 15);
            wa[s][ct] = (k < K && co < a.Cout) ? a.w[((size_t)co * 3 + (2 - cb)) * (KS * KS) + tap] : 0.f;
        }
    }
    __syncthreads();                                            // lut visible
    // ---- input tile: u8 -> float through the table, 0 outside the image (fma(0, w, acc) == acc: same as skipping) ----
    const int nd = (trow + 6) >> 2;                             // aligned dwords that cover a tile row at any alignment
    const int n_items = TIN * nd;
    // batches of 4 dwords per thread, all loads issued before the first table lookup (one memory latency per batch).
    // Byte positions are 32-bit offsets from the frame base (`mis` = its misalignment), so the address math is int32.
    const int mis = (int)((uintptr_t)img & 3);
    const int wrow = a.W * 3;
    for (int base_item = 0; base_item < ((MI355_STEM_EXP & 2) ? 0 : n_items); base_item += 4 * 256) {
        unsigned bytes[4];
        int dd[4], tlo[4], rlo[4], riy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int item = base_item + u * 256 + tid;
            const int iy = item / nd, j = item - iy * nd;
            const int gy = iy0 + iy;
            const bool rowin = item < n_items && (unsigned)gy < (unsigned)a.H;
            const int row_lo = (rowin ? gy : 0) * wrow, row_hi = row_lo + wrow;
            const int tile_lo = row_lo + ix0 * 3;
            const int d = ((tile_lo + mis) & ~3) - mis + 4 * j;   // offset of this dword; img + d is 4-byte aligned
            unsigned v = 0;
            if (rowin) {
                if (d >= row_lo && d + 4 <= row_hi) {
                    v = *(const unsigned*)(img + d);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (d + q >= row_lo && d + q < row_hi) v |= (unsigned)img[d + q] << (8 * q);
                }
            }
            bytes[u] = v; dd[u] = d; tlo[u] = tile_lo; riy[u] = item < n_items ? iy : -1;
            rlo[u] = rowin ? row_lo : 0x3fffffff;              // rows outside the image: no byte is "in the image"
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (riy[u] < 0) continue;
            const int row_hi = rlo[u] + wrow;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ab = dd[u] + q, rel = ab - tlo[u];
                if (rel >= 0 && rel < trow) {
                    const bool inimg = ab >= rlo[u] && ab < row_hi;
                    tin[riy[u] * trow + rel] = inimg ? lut[(bytes[u] >> (8 * q)) & 255u] : 0.f;
                }
            }
        }
    }
    __syncthreads();
    int base[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int ly = wave * PT + pt, lx = lane & 15;
        base[pt] = ((ly * a.stride) * TIN + lx * a.stride) * 3;
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ((MI355_STEM_EXP & 8) ? 0 : NS); ++s) {
        float xb[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xb[pt] = tin[base[pt] + koff[s]];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][ct], xb[pt], acc[ct][pt], 0, 0, 0);
    }
    // ---- epilogue: lane holds couts 16ct + 4g .. +3 of pixel (row 4*wave + pt, column lane & 15) ----
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + wave * PT + pt, ox = ox0 + (lane & 15);
        if (oy >= a.Hout || ox >= a.Wout) continue;
        if ((MI355_STEM_EXP & 4) && acc[0][pt][0] != 12345.678f) continue;
        const size_t po = ((size_t)b * a.Hout + oy) * a.Wout + ox;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + 4 * g;
            if (c >= a.Cout) continue;
            const f32x4 v = acc[ct][pt];
            if (a.out_half) {
                _Float16* dh = (_Float16*)a.dst + po * a.dst_cs + c;
                if (c + 3 < a.Cout) {
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)silu_fast(v[j] + a.bias[c + j]);
                    *(f16x4*)dh = o;
                } else {
                    for (int j = 0; j < 4 && c + j < a.Cout; ++j) dh[j] = (_Float16)silu_fast(v[j] + a.bias[c + j]);
                }
            } else {
                float* d = a.dst + po * a.dst_cs + c;
                if (c + 3 < a.Cout) {
                    f32x4 o;
                    if (a.fast_act) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = silu_fast(v[j] + a.bias[c + j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = silu_m(v[j] + a.bias[c + j]);
                    }
                    *(f32x4*)d = o;
                } else {
                    for (int j = 0; j < 4 && c + j < a.Cout; ++j) d[j] = a.fast_act ? silu_fast(v[j] + a.bias[c + j]) : silu_m(v[j] + a.bias[c + j]);
                }
            }
        }
    }
}

// half=True form of the stem: Ultralytics' half predictor rounds the /255 input AND the stem's weights to fp16 like every other
// conv's, so the products are exact in fp32 and K = 27 (108) fits ONE (four) v_mfma_f32_16x16x32_f16 per output tile instead of
// 7 (27) fp32 MFMAs of 32 cycles each.  Same block / tile geometry and input staging as stem_mfma_u8; the LDS tile holds fp16
// values (table lookup, then one rounding), the B operand of lane (pixel p, k-group g) is the 8 im2col values k = 32 kb + 8 g ..
// + 7 gathered with eight ds_read_u16; fp32 accumulation, bias + SiLU in fp32, one rounding on the store.
template <int KS, int CT>
__global__ __launch_bounds__(256) void stem_mfma_u8_h(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sl[];
    constexpr int K = KS * KS * 3, KB = (K + 31) / 32, PT = 4;
    const int TIN = (STEM_TO - 1) * a.stride + KS;
    const int trow = TIN * 3;                                   // halfs (= source bytes) per tile row
    float* lut = sl;
    _Float16* tin = (_Float16*)(sl + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    lut[tid] = a.lut[tid];
    const int tiles_x = (a.Wout + STEM_TO - 1) / STEM_TO, tiles_y = (a.Hout + STEM_TO - 1) / STEM_TO;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * STEM_TO, ox0 = tx * STEM_TO;
    const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;
    const uint8_t* img = a.img + (size_t)b * a.H * a.W * 3;
    // A operand: lane (row r = cout, k-group g) holds W[cout][k = 32 kb + 8 g + j], j = 0..7, rounded to fp16 (zero beyond K / Cout);
    // byte channel cb (0=B,1=G,2=R) feeds model channel 2-cb (im[..., ::-1]).  koff = LDS offset of k relative to the pixel's base.
    f16x8 wa[KB][CT];
    int koff[KB][8];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * kb + 8 * g + j;
            const int tap = k / 3, cb = k - tap * 3;
            const int kh = tap / KS, kw = tap - kh * KS;
            koff[kb][j] = k < K ? (kh * TIN + kw) * 3 + cb : 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int co = ct * 16 + (lane & 15);
                wa[kb][ct][j] = (_Float16)((k < K && co < a.Cout) ? a.w[((size_t)co * 3 + (2 - cb)) * (KS * KS) + tap] : 0.f);
            }
        }
    __syncthreads();                                            // lut visible
    // ---- input tile: u8 -> (float)i / 255 -> fp16, 0 outside the image (staging as in stem_mfma_u8) ----
    const int nd = (trow + 6) >> 2;
    const int n_items = TIN * nd;
    const int mis = (int)((uintptr_t)img & 3);
    const int wrow = a.W * 3;
    for (int base_item = 0; base_item < n_items; base_item += 4 * 256) {
        unsigned bytes[4];
        int dd[4], tlo[4], rlo[4], riy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int item = base_item + u * 256 + tid;
            const int iy = item / nd, j = item - iy * nd;
            const int gy = iy0 + iy;
            const bool rowin = item < n_items && (unsigned)gy < (unsigned)a.H;
            const int row_lo = (rowin ? gy : 0) * wrow, row_hi = row_lo + wrow;
            const int tile_lo = row_lo + ix0 * 3;
            const int d = ((tile_lo + mis) & ~3) - mis + 4 * j;
            unsigned v = 0;
            if (rowin) {
                if (d >= row_lo && d + 4 <= row_hi) {
                    v = *(const unsigned*)(img + d);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (d + q >= row_lo && d + q < row_hi) v |= (unsigned)img[d + q] << (8 * q);
                }
            }
            bytes[u] = v; dd[u] = d; tlo[u] = tile_lo; riy[u] = item < n_items ? iy : -1;
            rlo[u] = rowin ? row_lo : 0x3fffffff;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (riy[u] < 0) continue;
            const int row_hi = rlo[u] + wrow;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ab = dd[u] + q, rel = ab - tlo[u];
                if (rel >= 0 && rel < trow) {
                    const bool inimg = ab >= rlo[u] && ab < row_hi;
                    tin[riy[u] * trow + rel] = (_Float16)(inimg ? lut[(bytes[u] >> (8 * q)) & 255u] : 0.f);
                }
            }
        }
    }
    __syncthreads();
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int ly = wave * PT + pt, lx = lane & 15;
        const _Float16* px = tin + ((ly * a.stride) * TIN + lx * a.stride) * 3;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f16x8 xb;
#pragma unroll
            for (int j = 0; j < 8; ++j) xb[j] = px[koff[kb][j]];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[kb][ct], xb, acc[ct][pt], 0, 0, 0);
        }
    }
    // ---- epilogue: lane holds couts 16ct + 4g .. +3 of pixel (row 4*wave + pt, column lane & 15); fp16 stores ----
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + wave * PT + pt, ox = ox0 + (lane & 15);
        if (oy >= a.Hout || ox >= a.Wout) continue;
        const size_t po = ((size_t)b * a.Hout + oy) * a.Wout + ox;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + 4 * g;
            if (c >= a.Cout) continue;
            const f32x4 v = acc[ct][pt];
            _Float16* dh = (_Float16*)a.dst + po * a.dst_cs + c;
            if (c + 3 < a.Cout) {
                f16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)silu_fast(v[j] + a.bias[c + j]);
                *(f16x4*)dh = o;
            } else {
                for (int j = 0; j < 4 && c + j < a.Cout; ++j) dh[j] = (_Float16)silu_fast(v[j] + a.bias[c + j]);
            }
        }
    }
}

// The half stem for the shape every Ultralytics v8 model has (k 3, stride 2, pad 1).  The general form above is bound by its
// memory INSTRUCTIONS, not by arithmetic or bytes (what-if runs, tools/stem_whatif.py on config 5's 16 frames of 1280 x 1280: 389 us
// whole, 180 us without the stores, 238 us without the input and weight loads, 77 us with neither); this one issues a third of them:
//  * the input tile is fetched as ALIGNED DWORDS: with W a multiple of 4 a tile row starts at byte 96 tx - 3 of its image row, so the
//    26 dwords from byte 96 tx - 4 cover it and each lies wholly inside or wholly outside the image -- 4 loads per thread instead of
//    14 byte loads, no edge cases.  Byte c of the tile row is stored as the fp16 value (half)(b * (1/255.f)) at half index c + 1 of the
//    LDS row ((half)(b * (1/255.f)) == (half)(b / 255.f) for all 256 bytes: tests/test_half_stem_table.py), one ds_write_b64 per load;
//  * the A operand comes from a table the host prepares once per model (stem3_weight_frags below): three 16-byte loads per lane
//    instead of 24 dword loads and their index arithmetic;
//  * k is PERMUTED inside the one MFMA (K = 27 <= 32; the sum of 27 exact products defines no order, the general form's does not
//    either): pixel lx's nine values of kernel row kh sit at half indices 6 lx + 1 .. 6 lx + 9 of tile row 2 ly + kh, so lane group
//    g < 3 takes the four aligned dwords 6 lx .. 6 lx + 7 of row kh = g (the neighbour's last byte meets a zero weight, then values
//    0..6) and group 3 takes the dword 6 lx + 8, 6 lx + 9 (values 7, 8) of each of the three rows -- no per-value gather, no packing;
//  * the couts are PERMUTED over the MFMA rows so that a lane's 4 CT results are 4 CT consecutive channels of its pixel
//    (row 4 g + j of cout tile ct is channel 4 CT g + 4 ct + j): a wave's stores of one pixel row are one contiguous run of
//    16 pixels x Cout halfs, written with 16-byte stores.
// LDS rows are 108 halfs (216 B): 8-byte aligned for the staging writes, and the four lane groups' dword reads fall in distinct banks.
constexpr int STEM3_ROW = 108;

// [ct][lane][8] fp16 bit patterns: the A fragments of stem3s2_u8_h.  Byte channel cb (0=B,1=G,2=R) feeds model channel 2-cb (im[..., ::-1]).
void stem3_weight_frags(const float* w_oihw, int cout, std::vector<uint16_t>& out) {
    const int CT = (cout + 15) / 16;
    std::vector<float> f((size_t)CT * 64 * 8, 0.f);
    for (int ct = 0; ct < CT; ++ct)
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, r16 = lane & 15;
            const int co = 4 * CT * (r16 >> 2) + 4 * ct + (r16 & 3);
            if (co >= cout) continue;
            for (int s = 0; s < 8; ++s) {
                int kh, r;
                if (g < 3) { kh = g; r = s - 1; } else { kh = s >> 1; r = 7 + (s & 1); }
                if (r < 0 || kh > 2) continue;
                const int kw = r / 3, cb = r % 3;
                f[((size_t)ct * 64 + lane) * 8 + s] = w_oihw[((size_t)co * 3 + (2 - cb)) * 9 + kh * 3 + kw];
            }
        }
    out.resize(f.size());
    floats_to_halfs(f.data(), out.data(), f.size());
}

template <int CT>
__global__ __launch_bounds__(256) void stem3s2_u8_h(StemArgs a) {
    constexpr int TIN = 2 * (STEM_TO - 1) + 3, ROW = STEM3_ROW, PT = 4, NDW = 26, NIT = TIN * NDW;
    __shared__ __attribute__((aligned(16))) _Float16 tin[TIN * ROW];
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int tiles_x = (a.Wout + STEM_TO - 1) / STEM_TO, tiles_y = (a.Hout + STEM_TO - 1) / STEM_TO;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * STEM_TO, ox0 = tx * STEM_TO;
    const int iy0 = oy0 * 2 - 1, gd0 = (ox0 * 2 - 1) * 3 - 1;  // first tile row; byte (in its image row) of the row's first dword: 96 tx - 4
    const uint8_t* img = a.img + (size_t)b * a.H * a.W * 3;
    const int wrow = a.W * 3;
#if MI355_STEM_DIAG
    const int diag = a.variant >> 8;                            // what-if runs (tools/stem_whatif.py): 1 no SiLU, 2 no input loads, 4 no stores, 16 no weights
#else
    constexpr int diag = 0;
#endif
    // ---- input tile: 4 dwords per thread, all loads before the first conversion; outside the image -> 0 ----
    unsigned dw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = tid + 256 * u;
        const int iy = it / NDW, m = it - iy * NDW;
        const int gy = iy0 + iy, gd = gd0 + 4 * m;
        const bool in = it < NIT && (unsigned)gy < (unsigned)a.H && gd >= 0 && gd + 4 <= wrow;
        const unsigned v = (diag & 2) ? (unsigned)it * 0x01010101u : *(const unsigned*)(img + (in ? gy * wrow + gd : 0));
        dw[u] = in ? v : 0u;
    }
    f16x8 wa[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        if (diag & 16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wa[ct][j] = (_Float16)(0.01f * (float)(lane + j));
        } else {
            wa[ct] = *(const f16x8*)((const uint16_t*)a.wfrag + ((size_t)ct * 64 + lane) * 8);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = tid + 256 * u;
        const int iy = it / NDW, m = it - iy * NDW;
        f16x2 lo, hi;
        lo[0] = (_Float16)((float)(dw[u] & 255u) * (1.0f / 255.0f));
        lo[1] = (_Float16)((float)((dw[u] >> 8) & 255u) * (1.0f / 255.0f));
        hi[0] = (_Float16)((float)((dw[u] >> 16) & 255u) * (1.0f / 255.0f));
        hi[1] = (_Float16)((float)(dw[u] >> 24) * (1.0f / 255.0f));
        u32x2 pk;
        pk[0] = __builtin_bit_cast(unsigned, lo); pk[1] = __builtin_bit_cast(unsigned, hi);
        if (it < NIT) *(u32x2*)(tin + iy * ROW + 4 * m) = pk;
    }
    __syncthreads();
    int off[4];                                                 // byte offsets of the lane's four dwords from its pixel's base
#pragma unroll
    for (int d = 0; d < 4; ++d) off[d] = g < 3 ? g * (ROW * 2) + 4 * d : (d < 3 ? d : 0) * (ROW * 2) + 16;
    const char* tb = (const char*)tin;
    // lane holds channels 4 CT g + 4 ct + j of pixel (row 4*wave + pt, column lane & 15)
    const int c0 = 4 * CT * g;
    float bs[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[ct][j] = a.bias[c0 + 4 * ct + j];          // the bias array is padded to 16 CT floats
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int base = ((wave * PT + pt) * 2 * ROW + (lane & 15) * 6) * 2;
        u32x4 xv;
#pragma unroll
        for (int d = 0; d < 4; ++d) xv[d] = *(const unsigned*)(tb + base + off[d]);
        const f16x8 xb = __builtin_bit_cast(f16x8, xv);
        f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ct], xb, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const int oy = oy0 + wave * PT + pt, ox = ox0 + (lane & 15);
        if (oy >= a.Hout || ox >= a.Wout || c0 >= a.Cout) continue;
        if ((diag & 4) && acc[0][0] != 12345.678f) continue;
        const size_t po = ((size_t)b * a.Hout + oy) * a.Wout + ox;
        _Float16* dh = (_Float16*)a.dst + po * a.dst_cs + c0;
        _Float16 o[4 * CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[ct][j] + bs[ct][j];
                o[4 * ct + j] = (_Float16)((diag & 1) ? v : silu_fast(v));
            }
        if (c0 + 4 * CT <= a.Cout) {
            typedef _Float16 f16x8a __attribute__((ext_vector_type(8), aligned(8)));
#pragma unroll
            for (int q = 0; q + 1 < CT; q += 2) {
                f16x8a v8;
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = o[4 * q + j];
                *(f16x8a*)(dh + 4 * q) = v8;
            }
            if (CT & 1) {
                f16x4 v4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = o[4 * (CT - 1) + j];
                *(f16x4*)(dh + 4 * (CT - 1)) = v4;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4 * CT; ++j)
                if (c0 + j < a.Cout) dh[j] = o[j];
        }
    }
}

// The fp32 stem for k 3 / stride 2 / pad 1 with the same staging as stem3s2_u8_h: aligned dwords (4 loads per thread, a dword is wholly
// inside or outside the image; outside it is loaded as 0 and lut[0] = 0.f is the padding value, so the conversion has no edge case), tile
// byte c at float index c + 1 of a 104-float LDS row (one ds_write_b128 per dword), A operand from a host-prepared table
// (stem3_weight_frags_f32: 7 coalesced loads per lane instead of 7 gathered ones and their index arithmetic), couts permuted over the MFMA
// rows so that a lane stores 4 CT consecutive channels.  The MFMA chain is the general kernel's: k = (kh, kw, byte channel) ascending in
// steps of four, the oracle's det_stem order -- same operands, same order, same bits.
void stem3_weight_frags_f32(const float* w_oihw, int cout, std::vector<float>& out) {
    const int CT = (cout + 15) / 16;
    out.assign((size_t)CT * 7 * 64, 0.f);
    for (int ct = 0; ct < CT; ++ct)
        for (int s = 0; s < 7; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int g = lane >> 4, r16 = lane & 15, k = 4 * s + g;
                const int co = 4 * CT * (r16 >> 2) + 4 * ct + (r16 & 3);
                if (k >= 27 || co >= cout) continue;
                const int tap = k / 3, cb = k % 3;
                out[((size_t)ct * 7 + s) * 64 + lane] = w_oihw[((size_t)co * 3 + (2 - cb)) * 9 + tap];
            }
}

template <int CT>
__global__ __launch_bounds__(256) void stem3s2_u8_f32(StemArgs a) {
    constexpr int TIN = 2 * (STEM_TO - 1) + 3, ROW = 104, PT = 4, NDW = 26, NIT = TIN * NDW, NS = 7;
    __shared__ __attribute__((aligned(16))) float tin[TIN * ROW];
    __shared__ float lut[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    lut[tid] = a.lut[tid];
    const int tiles_x = (a.Wout + STEM_TO - 1) / STEM_TO, tiles_y = (a.Hout + STEM_TO - 1) / STEM_TO;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * STEM_TO, ox0 = tx * STEM_TO;
    const int iy0 = oy0 * 2 - 1, gd0 = (ox0 * 2 - 1) * 3 - 1;  // first tile row; byte (in its image row) of the row's first dword: 96 tx - 4
    const uint8_t* img = a.img + (size_t)b * a.H * a.W * 3;
    const int wrow = a.W * 3;
    unsigned dw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = tid + 256 * u;
        const int iy = it / NDW, m = it - iy * NDW;
        const int gy = iy0 + iy, gd = gd0 + 4 * m;
        const bool in = it < NIT && (unsigned)gy < (unsigned)a.H && gd >= 0 && gd + 4 <= wrow;
        const unsigned v = *(const unsigned*)(img + (in ? gy * wrow + gd : 0));
        dw[u] = in ? v : 0u;
    }
    float wa[NS][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int s = 0; s < NS; ++s) wa[s][ct] = ((const float*)a.wfrag)[((size_t)ct * NS + s) * 64 + lane];
    int koff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + g;                                // k / 9 and k % 9 with s a constant and g < 4: no division
        const int kh = (k >= 9) + (k >= 18), r = k - 9 * kh;
        koff[s] = k < 27 ? kh * ROW + r : 0;
    }
    __syncthreads();                                            // lut visible
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = tid + 256 * u;
        const int iy = it / NDW, m = it - iy * NDW;
        f32x4 v;
        v[0] = lut[dw[u] & 255u]; v[1] = lut[(dw[u] >> 8) & 255u]; v[2] = lut[(dw[u] >> 16) & 255u]; v[3] = lut[dw[u] >> 24];
        if (it < NIT) *(f32x4*)(tin + iy * ROW + 4 * m) = v;
    }
    __syncthreads();
    int base[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) base[pt] = (wave * PT + pt) * 2 * ROW + (lane & 15) * 6 + 1;
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float xb[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xb[pt] = tin[base[pt] + koff[s]];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][ct], xb[pt], acc[ct][pt], 0, 0, 0);
    }
    // ---- epilogue: lane holds channels 4 CT g + 4 ct + j of pixel (row 4*wave + pt, column lane & 15) ----
    const int c0 = 4 * CT * g;
    f32x4 bs[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bs[ct] = *(const f32x4*)(a.bias + c0 + 4 * ct);          // the bias array is padded to 16 CT floats
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + wave * PT + pt, ox = ox0 + (lane & 15);
        if (oy >= a.Hout || ox >= a.Wout) continue;
        const size_t po = ((size_t)b * a.Hout + oy) * a.Wout + ox;
        float* d = a.dst + po * a.dst_cs + c0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = c0 + 4 * ct;
            if (c >= a.Cout) continue;
            const f32x4 v = acc[ct][pt] + bs[ct];
            f32x4 o;
            if (a.fast_act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = silu_fast(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = silu_m(v[j]);
            }
            if (c + 3 < a.Cout) *(f32x4*)(d + 4 * ct) = o;
            else for (int j = 0; j < 4 && c + j < a.Cout; ++j) d[4 * ct + j] = o[j];
        }
    }
}

template <int KS>
static bool launch_stem_mfma(const StemArgs& a, unsigned grid, hipStream_t st) {
    const int tin = (STEM_TO - 1) * a.stride + KS;
    const size_t lds = (256 + (size_t)tin * tin * 3) * sizeof(float);
    if (lds > 64 * 1024) return false;
    const int ct = (a.Cout + 15) / 16;
    static const bool h_stem = !getenv("MI355_STEM_F16") || atoi(getenv("MI355_STEM_F16")) != 0;
    const char* lean_env = getenv("MI355_STEM_LEAN");             // read per launch: the tests switch it
    const bool h_lean = !lean_env || atoi(lean_env) != 0;
    if (!a.out_half && KS == 3 && a.stride == 2 && a.pad == 1 && a.wfrag && !(a.W & 3) && !((uintptr_t)a.img & 3) && h_lean) {
        switch (ct) {
            case 1: hipLaunchKernelGGL((stem3s2_u8_f32<1>), dim3(grid), dim3(256), 0, st, a); return true;
            case 2: hipLaunchKernelGGL((stem3s2_u8_f32<2>), dim3(grid), dim3(256), 0, st, a); return true;
            case 3: hipLaunchKernelGGL((stem3s2_u8_f32<3>), dim3(grid), dim3(256), 0, st, a); return true;
            case 4: hipLaunchKernelGGL((stem3s2_u8_f32<4>), dim3(grid), dim3(256), 0, st, a); return true;
            case 5: hipLaunchKernelGGL((stem3s2_u8_f32<5>), dim3(grid), dim3(256), 0, st, a); return true;
            default: return false;
        }
    }
    if (a.out_half && KS == 3 && a.stride == 2 && a.pad == 1 && a.wfrag && !(a.W & 3) && !((uintptr_t)a.img & 3) && (a.variant & 255) != 1 &&
        (h_lean || (a.variant & 255) == 2)) {
        switch (ct) {
            case 1: hipLaunchKernelGGL((stem3s2_u8_h<1>), dim3(grid), dim3(256), 0, st, a); return true;
            case 2: hipLaunchKernelGGL((stem3s2_u8_h<2>), dim3(grid), dim3(256), 0, st, a); return true;
            case 3: hipLaunchKernelGGL((stem3s2_u8_h<3>), dim3(grid), dim3(256), 0, st, a); return true;
            case 4: hipLaunchKernelGGL((stem3s2_u8_h<4>), dim3(grid), dim3(256), 0, st, a); return true;
            case 5: hipLaunchKernelGGL((stem3s2_u8_h<5>), dim3(grid), dim3(256), 0, st, a); return true;
            default: return false;
        }
    }
    if (a.out_half && h_stem) {                      // half=True: fp16 operands (the fp32 kernel's LDS size covers the fp16 tile)
        switch (ct) {
            case 1: hipLaunchKernelGGL((stem_mfma_u8_h<KS, 1>), dim3(grid), dim3(256), lds, st, a); return true;
            case 2: hipLaunchKernelGGL((stem_mfma_u8_h<KS, 2>), dim3(grid), dim3(256), lds, st, a); return true;
            case 3: hipLaunchKernelGGL((stem_mfma_u8_h<KS, 3>), dim3(grid), dim3(256), lds, st, a); return true;
            case 4: hipLaunchKernelGGL((stem_mfma_u8_h<KS, 4>), dim3(grid), dim3(256), lds, st, a); return true;
            case 5: hipLaunchKernelGGL((stem_mfma_u8_h<KS, 5>), dim3(grid), dim3(256), lds, st, a); return true;
            default: return false;
        }
    }
    switch (ct) {
        case 1: hipLaunchKernelGGL((stem_mfma_u8<KS, 1>), dim3(grid), dim3(256), lds, st, a); return true;
        case 2: hipLaunchKernelGGL((stem_mfma_u8<KS, 2>), dim3(grid), dim3(256), lds, st, a); return true;
        case 3: hipLaunchKernelGGL((stem_mfma_u8<KS, 3>), dim3(grid), dim3(256), lds, st, a); return true;
        case 4: hipLaunchKernelGGL((stem_mfma_u8<KS, 4>), dim3(grid), dim3(256), lds, st, a); return true;
        case 5: hipLaunchKernelGGL((stem_mfma_u8<KS, 5>), dim3(grid), dim3(256), lds, st, a); return true;
        default: return false;
    }
}

const char* launch_stem(const StemArgs& a, hipStream_t st) {
    if (a.k != 3 && a.k != 6) return "stem: only 3x3 and 6x6 stems are supported";
    if (a.dst_cs & 3) return "stem: destination pixel stride must be a multiple of 4 elements";
    const unsigned grid = (unsigned)((long)a.B * ((a.Wout + STEM_TO - 1) / STEM_TO) * ((a.Hout + STEM_TO - 1) / STEM_TO));
    const bool ok = a.k == 3 ? launch_stem_mfma<3>(a, grid, st) : launch_stem_mfma<6>(a, grid, st);
    if (!ok) return "stem: more than 80 output channels or a tile that does not fit in LDS";
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------------------- upsample
// torch.nn.Upsample(scale_factor=2, mode="nearest") written straight into the concat buffer's channel slice.
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* src, int src_cs, float* dst, int dst_cs,
                                                         int B, int H, int W, int c4n, int C) {
    const long total = (long)B * (2 * H) * (2 * W) * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % c4n);
        long p = i / c4n;
        const int ox = (int)(p % (2 * W)); p /= (2 * W);
        const int oy = (int)(p % (2 * H));
        const int b = (int)(p / (2 * H));
        const float* s = src + (((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * src_cs + 4 * q;
        float* d = dst + (((size_t)b * 2 * H + oy) * (2 * W) + ox) * dst_cs + 4 * q;
        if (4 * q + 3 < C) *(f32x4*)d = *(const f32x4*)s;
        else for (int j = 0; 4 * q + j < C; ++j) d[j] = s[j];
    }
}

const char* launch_upsample2x(const float* src, int src_cs, float* dst, int dst_cs, int B, int H, int W, int C,
                              hipStream_t st) {
    const int c4n = (C + 3) / 4;
    const long total = (long)B * 4 * H * W * c4n;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid), dim3(256), 0, st, src, src_cs, dst, dst_cs, B, H, W, c4n, C);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------------------- SPPF pools
// Three chained MaxPool2d(5, stride 1, pad 2) (block.py:SPPF.forward): x1 = pool(x0), x2 = pool(x1), x3 = pool(x2).
// One block owns (image, 16 channels): the H/32 x W/32 map of those channels lives in LDS and the three pools run
// back to back on it (25 LDS reads per output float4 and pool, instead of a 13x13 global window).  max is exact.

__global__ __launch_bounds__(256) void sppf_pools_kernel(const float* src, int src_cs, float* dst, int dst_cs,
                                                         int B, int H, int W, int C, int POOL_C) {
    extern __shared__ __attribute__((aligned(16))) float pl[];              // two maps [H*W][POOL_C]
    const int cgroups = (C + POOL_C - 1) / POOL_C;
    const int b = blockIdx.x / cgroups, c0 = (blockIdx.x % cgroups) * POOL_C;
    const int npx = H * W, nq = POOL_C / 4;
    float* cur = pl;
    float* tmp = pl + (size_t)npx * POOL_C;
    const float NEG = -__builtin_huge_valf();
    for (int i = threadIdx.x; i < npx * nq; i += 256) {
        const int p = i / nq, q = i % nq;
        f32x4 v = (f32x4){NEG, NEG, NEG, NEG};
        if (c0 + 4 * q < C) v = *(const f32x4*)(src + ((size_t)b * npx + p) * src_cs + c0 + 4 * q);
        *(f32x4*)(cur + p * POOL_C + 4 * q) = v;
    }
    __syncthreads();
    // a 5x5 max is a 5-wide row max followed by a 5-tall column max (max is associative and exact): 10 LDS reads per output
    // and pool instead of 25
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = threadIdx.x; i < npx * nq; i += 256) {                 // rows: cur -> tmp
            const int p = i / nq, q = i % nq;
            const int y = p / W, x = p - y * W;
            f32x4 m = (f32x4){NEG, NEG, NEG, NEG};
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const f32x4 v = *(const f32x4*)(cur + (y * W + xx) * POOL_C + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
            }
            *(f32x4*)(tmp + p * POOL_C + 4 * q) = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < npx * nq; i += 256) {                 // columns: tmp -> cur (+ the global store)
            const int p = i / nq, q = i % nq;
            const int y = p / W, x = p - y * W;
            f32x4 m = (f32x4){NEG, NEG, NEG, NEG};
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                const f32x4 v = *(const f32x4*)(tmp + (yy * W + x) * POOL_C + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
            }
            *(f32x4*)(cur + p * POOL_C + 4 * q) = m;                        // cur's old value at p is only read through tmp now
            if (c0 + 4 * q < C) {
                float* d = dst + ((size_t)b * npx + p) * dst_cs + pass * C + c0 + 4 * q;
                if (c0 + 4 * q + 3 < C) *(f32x4*)d = m;
                else for (int j = 0; c0 + 4 * q + j < C; ++j) d[j] = m[j];
            }
        }
        __syncthreads();
    }
}

// fallback for maps too large for LDS: 5/9/13-window maxima straight from global memory (max composes exactly)
__global__ __launch_bounds__(256) void sppf_pools_global_kernel(const float* src, int src_cs, float* dst, int dst_cs,
                                                         int B, int H, int W, int c4n, int C) {
    const long total = (long)B * H * W * c4n;
    const float NEG = -__builtin_huge_valf();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % c4n);
        long p = i / c4n;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        f32x4 m1 = (f32x4){NEG, NEG, NEG, NEG}, m2 = m1, m3 = m1;
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)H) continue;
            const int ady = dy < 0 ? -dy : dy;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const int adx = dx < 0 ? -dx : dx;
                const int r = ady > adx ? ady : adx;
                const f32x4 v = *(const f32x4*)(src + (((size_t)b * H + yy) * W + xx) * src_cs + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m3[j] = fmaxf(m3[j], v[j]);
                    if (r <= 4) m2[j] = fmaxf(m2[j], v[j]);
                    if (r <= 2) m1[j] = fmaxf(m1[j], v[j]);
                }
            }
        }
        float* d = dst + (((size_t)b * H + y) * W + x) * dst_cs + 4 * q;
        for (int j = 0; j < 4 && 4 * q + j < C; ++j) { d[j] = m1[j]; d[C + j] = m2[j]; d[2 * C + j] = m3[j]; }
    }
}

const char* launch_sppf_pools(const float* src, int src_cs, float* dst, int dst_cs, int B, int H, int W, int C,
                              hipStream_t st) {
    if (C & 3) return "sppf: channel count must be a multiple of 4";
    int pc = 0;
    for (int c = 16; c >= 4; c >>= 1)
        if ((size_t)2 * H * W * c * sizeof(float) <= 64 * 1024) { pc = c; break; }     // 1280x1280: 40x40 map, 4 channels = 51 KB
    while (pc > 4 && (long)B * ((C + pc - 1) / pc) < 256) pc >>= 1;      // small batches: more, narrower blocks (one per CU at least)
    if (!pc) {
        const int c4n = C / 4;
        const long total = (long)B * H * W * c4n;
        const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(sppf_pools_global_kernel, dim3(grid), dim3(256), 0, st, src, src_cs, dst, dst_cs, B, H, W, c4n, C);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? nullptr : hipGetErrorString(e);
    }
    const size_t lds = (size_t)2 * H * W * pc * sizeof(float);
    const unsigned grid = (unsigned)(B * ((C + pc - 1) / pc));
    hipLaunchKernelGGL(sppf_pools_kernel, dim3(grid), dim3(256), lds, st, src, src_cs, dst, dst_cs, B, H, W, C, pc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// fp16 variants of the two pool kernels: 8 channels per 16-byte vector
__device__ __forceinline__ f16x8 max8(f16x8 a, f16x8 b) {
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = b[j] > a[j] ? b[j] : a[j];
    return r;
}

__global__ __launch_bounds__(256) void sppf_pools_f16_kernel(const _Float16* src, int src_cs, _Float16* dst, int dst_cs,
                                                             int B, int H, int W, int C, int POOL_C) {
    extern __shared__ __attribute__((aligned(16))) _Float16 plh[];          // two maps [H*W][POOL_C]
    const int cgroups = (C + POOL_C - 1) / POOL_C;
    const int b = blockIdx.x / cgroups, c0 = (blockIdx.x % cgroups) * POOL_C;
    const int npx = H * W, nq = POOL_C / 8;
    _Float16* cur = plh;
    _Float16* nxt = plh + (size_t)npx * POOL_C;
    const _Float16 NEG = (_Float16)(-__builtin_huge_valf());
    const f16x8 NEG8 = (f16x8){NEG, NEG, NEG, NEG, NEG, NEG, NEG, NEG};
    for (int i = threadIdx.x; i < npx * nq; i += 256) {
        const int p = i / nq, q = i % nq;
        f16x8 v = NEG8;
        if (c0 + 8 * q < C) v = *(const f16x8*)(src + ((size_t)b * npx + p) * src_cs + c0 + 8 * q);
        *(f16x8*)(cur + p * POOL_C + 8 * q) = v;
    }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {                                 // 5x5 max = row max, then column max (see sppf_pools_kernel)
        for (int i = threadIdx.x; i < npx * nq; i += 256) {
            const int p = i / nq, q = i % nq;
            const int y = p / W, x = p - y * W;
            f16x8 m = NEG8;
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                m = max8(m, *(const f16x8*)(cur + (y * W + xx) * POOL_C + 8 * q));
            }
            *(f16x8*)(nxt + p * POOL_C + 8 * q) = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < npx * nq; i += 256) {
            const int p = i / nq, q = i % nq;
            const int y = p / W, x = p - y * W;
            f16x8 m = NEG8;
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                m = max8(m, *(const f16x8*)(nxt + (yy * W + x) * POOL_C + 8 * q));
            }
            *(f16x8*)(cur + p * POOL_C + 8 * q) = m;
            if (c0 + 8 * q < C) *(f16x8*)(dst + ((size_t)b * npx + p) * dst_cs + pass * C + c0 + 8 * q) = m;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sppf_pools_f16_global_kernel(const _Float16* src, int src_cs, _Float16* dst, int dst_cs,
                                                                    int B, int H, int W, int c8n, int C) {
    const long total = (long)B * H * W * c8n;
    const _Float16 NEG = (_Float16)(-__builtin_huge_valf());
    const f16x8 NEG8 = (f16x8){NEG, NEG, NEG, NEG, NEG, NEG, NEG, NEG};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % c8n);
        long p = i / c8n;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        f16x8 m1 = NEG8, m2 = NEG8, m3 = NEG8;
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)H) continue;
            const int ady = dy < 0 ? -dy : dy;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const int adx = dx < 0 ? -dx : dx;
                const int r = ady > adx ? ady : adx;
                const f16x8 v = *(const f16x8*)(src + (((size_t)b * H + yy) * W + xx) * src_cs + 8 * q);
                m3 = max8(m3, v);
                if (r <= 4) m2 = max8(m2, v);
                if (r <= 2) m1 = max8(m1, v);
            }
        }
        _Float16* d = dst + (((size_t)b * H + y) * W + x) * dst_cs + 8 * q;
        *(f16x8*)d = m1; *(f16x8*)(d + C) = m2; *(f16x8*)(d + 2 * C) = m3;
    }
}

const char* launch_sppf_pools_f16(const void* src, int src_cs, void* dst, int dst_cs, int B, int H, int W, int C,
                                  hipStream_t st) {
    if (C & 7) return "sppf(f16): channel count must be a multiple of 8";
    int pc = 0;
    for (int c = 32; c >= 8; c >>= 1)
        if ((size_t)2 * H * W * c * 2 <= 64 * 1024) { pc = c; break; }                 // 1280x1280: 40x40 map, 8 channels = 51 KB
    if (!pc) {
        const int c8n = C / 8;
        const long total = (long)B * H * W * c8n;
        const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(sppf_pools_f16_global_kernel, dim3(grid), dim3(256), 0, st, (const _Float16*)src, src_cs, (_Float16*)dst,
                           dst_cs, B, H, W, c8n, C);
    } else {
        const size_t lds = (size_t)2 * H * W * pc * 2;
        const unsigned grid = (unsigned)(B * ((C + pc - 1) / pc));
        hipLaunchKernelGGL(sppf_pools_f16_kernel, dim3(grid), dim3(256), lds, st, (const _Float16*)src, src_cs, (_Float16*)dst, dst_cs,
                           B, H, W, C, pc);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------------------- letterbox
// data/augment.py:LetterBox: cv2.resize(INTER_LINEAR) in OpenCV's 11-bit fixed point, then a 114 border.
// Integer arithmetic only -> bit-exact against the oracle's restatement.
__global__ __launch_bounds__(256) void letterbox_kernel(LetterboxArgs a) {
    const long total = (long)a.B * a.Hd * a.Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.Wd);
        const int y = (int)((i / a.Wd) % a.Hd);
        const int b = (int)(i / ((long)a.Wd * a.Hd));
        uint8_t* d = a.dst + (size_t)i * 3;
        const int rx = x - a.left, ry = y - a.top;
        if ((unsigned)rx >= (unsigned)a.Wr || (unsigned)ry >= (unsigned)a.Hr) { d[0] = d[1] = d[2] = 114; continue; }
        const uint8_t* s = a.src + (size_t)b * a.frame_stride;
        if (!a.resize) {
            const uint8_t* p = s + (size_t)ry * a.row_stride + (size_t)rx * 3;
            d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
            continue;
        }
        const int x0 = a.xtab[rx * 3], ax0 = a.xtab[rx * 3 + 1], ax1 = a.xtab[rx * 3 + 2];
        const int y0 = a.ytab[ry * 3], by0 = a.ytab[ry * 3 + 1], by1 = a.ytab[ry * 3 + 2];
        const int x1 = x0 + 1 < a.W ? x0 + 1 : a.W - 1;
        const int y1 = y0 + 1 < a.H ? y0 + 1 : a.H - 1;
        const uint8_t* r0 = s + (size_t)y0 * a.row_stride;
        const uint8_t* r1 = s + (size_t)y1 * a.row_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;
            const int s1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
            int v = (((by0 * (s0 >> 4)) >> 16) + ((by1 * (s1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            d[c] = (uint8_t)v;
        }
    }
}

const char* launch_letterbox(const LetterboxArgs& a, hipStream_t st) {
    const long total = (long)a.B * a.Hd * a.Wd;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace mi355
