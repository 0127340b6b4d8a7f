// half=True path, second kernel family (round 4): 3x3 / stride-1 convs with the block's WEIGHTS in LDS ("lw"), a persistent block
// loop and register-staged prefetch across chunk AND tile boundaries.  Replaces nothing in the reference by itself: it is another
// launch plan (version 7) of the conv that conv_f16.h implements -- ultralytics' half=True predictor mode reached through
// /root/reference/model.py:38, BASELINE config 5 (YOLOv8m 1280x1280 fp16) -- and produces the same bits as every other plan
// (same k-block-major, tap-minor accumulation order on the same instruction).
//
// Why (profiles/r03_cfg5_*): in conv_igemm_f16 every wave streams its own weight fragments through the 64 B/clk vector
// L1 inside the K loop (1 KiB per MFMA per PT pixel tiles), the halo tile of the next chunk is not requested before the current
// chunk's MFMAs are done, and a block's prologue / epilogue are covered only by whatever other block happens to share the CU.
// Here
//   * a block = 4 waves = one 16 x 16 output tile x (CT * 16) couts; the wave owns 4 rows of the tile (PT = 4 pixel tiles) and
//     ALL of the block's cout tiles, so the four waves read the same weight fragments -- staged ONCE per block and k-block into
//     LDS (9 * CT KiB, fragment order: one ds_read_b128 per fragment, conflict-free) instead of four times through L1;
//   * work = (tile, cout group, k-block) items walked by a persistent block; the global loads of item i + 1 (halo tile slice
//     and weights, next tile's first k-block included) are in flight into registers while item i's MFMAs run from LDS;
//   * the halo tile is [pixel][32 channels] with a 64-byte pixel stride and the 16-byte slot index XOR-swizzled by bit 2 of the
//     pixel index: B-operand reads are conflict-free for ds_read_b128's lane groups without padding (20.7 KB instead of 31 KB,
//     which is what lets CT = 6 run two blocks per CU);
//   * two blocks per CU: one block's LDS write phase / epilogue under the other's MFMAs.
#include "conv_f16.h"

namespace mi355 {

namespace {

constexpr int kLwTile = 16, kLwHalo = 18, kLwPix = kLwHalo * kLwHalo;       // output tile edge, halo tile edge, halo pixels
constexpr int kLwXBytes = kLwPix * 64;                                      // 32 channels x 2 bytes per halo pixel

template <int CT>
__global__ __launch_bounds__(256, 2) void conv3x3_lw_f16(ConvKArgs a) {
    constexpr int PT = 4, NF = 9 * CT, NWU = (NF + 3) / 4, NXU = (kLwPix * 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLwXBytes + NWU * 4096];       // weight region: whole passes of 4 fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const unsigned g = (unsigned)lane >> 4;

    // What-if diagnostics, built only with -DMI355_F16_DIAG=1 (tools/ab_build.sh), selected by MI355_F16_EXP: 1 no global loads, 2 no LDS writes,
    // 4 no stores, 8 weight fragments not re-read from LDS, 16 pixel fragments not re-read, 32 no barriers, 64 no bias load in the epilogue
#ifdef MI355_F16_DIAG
    const int exp_flags = a.lds_buf_floats;
#else
    constexpr int exp_flags = 0;
#endif
    // A/B knobs (tools/ab_build.sh): MI355_LW_STAGGER = 64-cycle units the second co-resident block of a CU sleeps before its first item (the two
    // blocks start in lockstep: same program, same durations); MI355_LW_OPAQUE_X = 1 keeps the 18 pixel-fragment addresses out of registers
#ifndef MI355_LW_STAGGER
#define MI355_LW_STAGGER 0
#endif
#ifndef MI355_LW_OPAQUE_X
#define MI355_LW_OPAQUE_X (CT >= 6)
#endif
    if (MI355_LW_STAGGER > 0) {
        const unsigned tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4);       // HW_ID.TG_ID: which of the CU's resident blocks this is
        if (tg & 1u) __builtin_amdgcn_s_sleep(MI355_LW_STAGGER);
    }
    const int gy = a.cgroups, n_units = a.n_tiles_total * gy, G = (int)gridDim.x;
    const int my_units = ((int)blockIdx.x < n_units) ? (n_units - 1 - (int)blockIdx.x) / G + 1 : 0;
    const int cib = a.cib, n_items = my_units * cib;
    if (n_items == 0) return;

    // unit j of this block -> (image, tile row, tile column, cout group): XCD-aware order (common.h), tile-major, cout group innermost
    auto decode = [&](int j, int& b, int& ty, int& tx, int& cg) {
        const unsigned u = blockIdx.x + (unsigned)j * (unsigned)G;
        const unsigned n = (unsigned)n_units, qn = n >> 3, rn = n & 7, x = u & 7;
        const unsigned logical = (x < rn ? x * (qn + 1) : rn * (qn + 1) + (x - rn) * qn) + (u >> 3);
        const unsigned t = fastdiv(logical, FastDiv{a.fd_gy.ml, a.fd_gy.mh});
        cg = (int)(logical - t * (unsigned)gy);
        const unsigned tq = fastdiv(t, FastDiv{a.fd_tx.ml, a.fd_tx.mh});
        tx = (int)(t - tq * (unsigned)a.tiles_x);
        const unsigned bb = fastdiv(tq, FastDiv{a.fd_ty.ml, a.fd_ty.mh});
        ty = (int)(tq - bb * (unsigned)a.tiles_y);
        b = (int)bb;
    };

    // ---- load stream: item (jL, kL) = (unit, k-block) whose global loads are issued next --------------------------------------
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned q = (unsigned)tid & 3u;                                   // this thread's 16-byte slot (8 channels) of a halo pixel
    const bool qok_last = (cib - 1) * 32 + 8 * (int)q < a.cin4;              // last k-block: channels beyond round_up(Cin, 8) read zeros
    const int n_wfrag = a.n_ctiles * 9;
    int jL = 0, kL = 0, wfr0 = 0;
    unsigned voff[NXU];
    __amdgpu_buffer_rsrc_t srs = wrsrc;
    f16x8 px[NXU], pw[NWU];
    // Every load is unconditional (a conditional load makes the compiler keep two homes for its destination and copy between them
    // right behind the load, which waits for it): beyond the last item the stream re-reads the last one, weight fragments beyond
    // 9 * CT read zeros through an offset past num_records.
    // Split in two: the set-up of an item (scalar work, six offsets when a new unit begins) and its NXU + NWU loads, which compute()
    // issues one at a time between its first micro-steps (MI355_LW_INTERLEAVE, default on): in the shadow of the MFMAs instead of as a
    // phase of its own between the two barriers.
    bool drop = false;
    auto prefetch_setup = [&]() {
        if (kL == 0) {
            int b, ty, tx, cg;
            decode(jL, b, ty, tx, cg);
            srs = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)a.src + (size_t)b * (size_t)a.img_src), 0, (int)((unsigned)a.img_src * 2u),
                                                    0x00020000);
            const int iy0 = ty * kLwTile - 1, ix0 = tx * kLwTile - 1;
#pragma unroll
            for (int u = 0; u < NXU; ++u) {
                const int pix = u * 64 + (tid >> 2);
                const int hy = (pix * 3641) >> 16, hx = pix - hy * kLwHalo;                      // pix / 18 for pix < 448
                const int gyy = iy0 + hy, gxx = ix0 + hx;
                const bool ok = pix < kLwPix && (unsigned)gyy < (unsigned)a.Hin && (unsigned)gxx < (unsigned)a.Win;
                voff[u] = ok ? (unsigned)(__mul24(__mul24(gyy, a.Win) + gxx, a.src_cs) * 2) + q * 16u : kOOB;
            }
            wfr0 = cg * CT * 9;
        }
        drop = (kL == cib - 1) && !qok_last;
    };
    auto prefetch_load = [&](int l) {                      // l: compile-time index, halo slots first
        if (exp_flags & 1) return;
        if (l < NXU) {
            px[l] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(drop ? kOOB : voff[l]), kL * 64, 0));
        } else {
            const int u = l - NXU, f = 4 * u + wave;
            const int fi = min(wfr0 + f, n_wfrag - 1);              // cout tiles beyond the last one re-read it (their results are never stored)
            pw[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(f < NF ? lane16 : kOOB), (fi * cib + kL) * 1024, 0));
        }
    };
    auto prefetch_advance = [&]() {
        if (jL + 1 < my_units || kL + 1 < cib) { if (++kL == cib) { kL = 0; ++jL; } }
    };
    auto prefetch = [&]() {
        prefetch_setup();
#pragma unroll
        for (int l = 0; l < NXU + NWU; ++l) prefetch_load(l);
        prefetch_advance();
    };
    // registers -> LDS.  Halo slot (pix, q) lives at pix * 64 + ((q ^ 2 * bit2(pix)) * 16); pix = u * 64 + tid / 4, so the swizzle bit
    // is bit 4 of tid for every u.  Weight fragment f = 4 u + wave at f * 1024 + lane * 16 = u * 4096 + tid * 16.
    const unsigned cx = (unsigned)(tid >> 2) * 64u + ((q ^ ((((unsigned)tid >> 4) & 1u) << 1)) * 16u);
    auto commit = [&]() {
        if (exp_flags & 2) return;
#pragma unroll
        for (int u = 0; u < NXU; ++u)
            if (u * 64 + (tid >> 2) < kLwPix) *(f16x8*)(lds + cx + u * 4096) = px[u];
#pragma unroll
        for (int u = 0; u < NWU; ++u) *(f16x8*)(lds + kLwXBytes + u * 4096 + tid * 16) = pw[u];
    };

    // ---- compute stream ------------------------------------------------------------------------------------------------------
    // B operand of pixel tile pt at tap (tr, tc): halo pixel (4 wave + pt + tr, (lane & 15) + tc), channel group g.  With r = pt + tr:
    // byte = xbase + (r * 18 + tc) * 64 + 32 * swz(r, tc), xbase = (4 wave * 18 + (lane & 15)) * 64 + (g & 1) * 16 and
    // swz(r, tc) = bit2(halo pixel index) ^ (g >> 1): 18 bits per lane, computed once.
    const unsigned pl = (unsigned)(4 * wave * kLwHalo + (lane & 15));
    const unsigned xbase = pl * 64u + (g & 1u) * 16u;
    unsigned swm = 0;
#pragma unroll
    for (int i = 0; i < 18; ++i) swm |= ((((pl + (unsigned)((i / 3) * kLwHalo + (i % 3))) >> 2) ^ (g >> 1)) & 1u) << i;
    const unsigned wbase = (unsigned)kLwXBytes + lane16;
    auto xfrag = [&](int r, int tc) -> f16x8 {
        unsigned sw = swm;
        if (MI355_LW_OPAQUE_X) asm volatile("" : "+v"(sw));             // recomputed per read (2 vector instructions) instead of 18 live addresses
        const unsigned off = xbase + (((sw >> (r * 3 + tc)) & 1u) << 5);
        return *(const f16x8*)__builtin_assume_aligned(lds + off + (r * kLwHalo + tc) * 64, 16);
    };
    auto wfrag = [&](int m) -> f16x8 {                                  // micro-step m = tap * CT + ct
        return *(const f16x8*)__builtin_assume_aligned(lds + wbase + ((m % CT) * 9 + m / CT) * 1024, 16);
    };
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One item = one k-block: 9 taps x CT cout tiles x 4 pixel tiles.  Micro-step (tap, ct) = 4 MFMAs on one weight fragment; weight
    // fragments travel through a 3-slot ring two micro-steps ahead, pixel fragments are double-buffered per tap and fetched during
    // the previous tap's micro-steps.
#ifndef MI355_LW_INTERLEAVE
#define MI355_LW_INTERLEAVE 1
#endif
#ifndef MI355_LW_WRING
#define MI355_LW_WRING 3
#endif
#ifndef MI355_LW_PRIO
#define MI355_LW_PRIO 0
#endif
    auto compute = [&](auto with_loads) {
        constexpr bool WL = decltype(with_loads)::value;
        constexpr int NL = NXU + NWU, M = 9 * CT;
        constexpr int WR = MI355_LW_WRING, WA = WR - 1;          // weight-fragment ring: WR slots, WA micro-steps ahead
        f16x8 xf[2][PT], wr[WR];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xf[0][pt] = xfrag(pt, 0);
#pragma unroll
        for (int m = 0; m < WA; ++m) wr[m] = wfrag(m);
        if (MI355_LW_PRIO) __builtin_amdgcn_s_setprio(1);          // the MFMA phase outranks the co-resident block's staging instructions
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int m = t * CT + ct;
                if (m + WA < M && !(exp_flags & 8)) wr[(m + WA) % WR] = wfrag(m + WA);
                if (t + 1 < 9 && !(exp_flags & 16)) {
#pragma unroll
                    for (int pt = ct * PT / CT; pt < (ct + 1) * PT / CT; ++pt) xf[(t + 1) & 1][pt] = xfrag(pt + (t + 1) / 3, (t + 1) % 3);
                }
                if (WL && m < NL) prefetch_load(m);                 // next item's global loads: one per micro-step, all in the first half
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[m % WR], xf[t & 1][pt], acc[ct][pt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (MI355_LW_PRIO) __builtin_amdgcn_s_setprio(0);
        static_assert(NL <= M, "one load per micro-step");
    };
    auto epilogue = [&](int j) {
        int b, ty, tx, cg;
        decode(j, b, ty, tx, cg);
        const int ct0 = cg * CT;
        const OutF16 o = make_out_f16(a.dst, a.dst_cs, a.img_dst, a.res, a.res_cs, a.img_res, b, a.Cout, a.act, a.out_f32);
        int pixi[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int oy = ty * kLwTile + 4 * wave + pt, ox = tx * kLwTile + (lane & 15);
            pixi[pt] = (oy < a.Hout && ox < a.Wout) ? __mul24(oy, a.Wout) + ox : -1;
        }
        f32x4 bias4[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
            bias4[ct] = *(const f32x4*)(a.bias + tile_cout0(ctile, lane >> 4, conv_f16_pairs(a.Cout)));
            if (exp_flags & 64) bias4[ct] = (f32x4){0.01f, 0.02f, 0.03f, 0.04f};      // what-if: no bias load in the epilogue
        }
        if (exp_flags & 4) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) pixi[pt] = acc[0][pt][0] == 12345.678f ? pixi[pt] : -1;
        }
        store_tiles_f16_v2<PT, CT>(o, acc, bias4, lane, ct0, pixi);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // Loop order: the iteration OPENS with the step that needs the previous iteration's global loads anyway (registers -> LDS), so the
    // wait the compiler places at the loop header costs nothing; the loads of item i + 1 are then in flight during compute(i).  A
    // finished unit's epilogue runs after the LDS writes and BEFORE the next loads are issued (vmcnt retires in order: waiting for those
    // loads one iteration later must not mean waiting for younger stores), under the co-resident block's MFMAs.
    prefetch();                                            // item 0
    int jC = 0, kC = 0, jE = -1;
    for (int i = 0;; ++i) {
        if (i < n_items) commit();                         // item i
        if (jE >= 0) { epilogue(jE); jE = -1; }
        if (i == n_items) break;
        if (MI355_LW_INTERLEAVE) {
            prefetch_setup();                              // item i + 1: its loads are issued inside compute()
            if (!(exp_flags & 32)) __syncthreads();
            compute(std::true_type{});
            prefetch_advance();
        } else {
            prefetch();
            if (!(exp_flags & 32)) __syncthreads();
            compute(std::false_type{});
        }
        if (!(exp_flags & 32)) __syncthreads();            // every wave is done reading this item's LDS image
        if (kC == cib - 1) jE = jC;
        if (++kC == cib) { kC = 0; ++jC; }
    }
}

// ---- version 10 (round 4): pointwise convs in the same style ---------------------------------------------------------------------------
// What-if runs (profiles/r04_pointwise_whatif.txt) showed the half-mode 1x1 launches spending half their time in an input-staging phase nothing
// overlaps, and a fifth waiting for weight fragments every wave fetches for itself.  Here a block = 4 waves x 64 pixels (PT = 4) = 256
// flattened pixels x CT * 16 couts, persistent over (pixel tile, cout group) units in the XCD-aware order.
//   * The waves share the WEIGHTS, which travel through a double-buffered LDS region in chunks of KC = 4 k-blocks (registers -> LDS inside
//     the MFMA phase of the previous chunk: one barrier per chunk, no phase of its own).
//   * A wave stages ITS OWN 64 pixels two k-blocks (128 bytes per pixel = one cache line) at a time: a load instruction covers 8 pixels x one full
//     line (version 9 of this round streamed MFMA-fragment-shaped loads -- 16 half lines per instruction, what the texture addresser is slowest
//     at -- and lost to the older kernels on every compute-side shape: 183 vs 145 us on 1152 -> 384 at 80 x 80), the data goes registers -> a
//     WAVE-PRIVATE 8 KiB LDS image (XOR-swizzled 16-byte slots: slot' = slot ^ ((pixel >> 1) & 7), conflict-free for ds_read_b128's lane groups
//     and for the 8-lane write groups), and the B fragments are read from there.  Only the wave itself touches its image, so the pixel side
//     needs no block barrier (LDS executes a wave's accesses in order); its global loads run one X-chunk ahead, across chunk and unit boundaries.
// Same single accumulation chain over ascending k-blocks as every other half-mode plan: the same bits.
// UP: the nearest-2x upsample fused into the read side, as conv1x1_pipe_f16 / conv1x1_stream_up_f32 have it -- the X-chunks of the first a.up_c input
// channels (whole 64-channel chunks) are staged from the half-resolution tensor a.src2 at (y >> 1, x >> 1); fd_tx / fd_ty divide by up_W / up_W * up_H.
template <int CT, bool UP = false>
__global__ __launch_bounds__(256, 2) void conv1x1_lwx_f16(ConvKArgs a) {
    constexpr int PT = 4, KC = 4, NFW = CT * KC, NWU = (NFW + 3) / 4, WBUF = NWU * 4096;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WBUF + 4 * 8192];          // weights (two buffers) | one 8 KiB pixel image per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const unsigned g = (unsigned)lane >> 4;
    const int total = a.Wout;                                  // flattened pixels (batch x height x width)
    const int gy = a.cgroups, n_units = a.n_tiles_total * gy, G = (int)gridDim.x;
    const int my_units = ((int)blockIdx.x < n_units) ? (n_units - 1 - (int)blockIdx.x) / G + 1 : 0;
    const int cib = a.cib, n_chunks = (cib + KC - 1) / KC;
    if (my_units == 0) return;
    auto decode = [&](int j, int& tile, int& cg) {
        const unsigned u = blockIdx.x + (unsigned)j * (unsigned)G;
        const unsigned n = (unsigned)n_units, qn = n >> 3, rn = n & 7, x = u & 7;
        const unsigned logical = (x < rn ? x * (qn + 1) : rn * (qn + 1) + (x - rn) * qn) + (u >> 3);
        const unsigned t = fastdiv(logical, FastDiv{a.fd_gy.ml, a.fd_gy.mh});
        cg = (int)(logical - t * (unsigned)gy);
        tile = (int)t;
    };
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, 0x7fffffff, 0x00020000);
    // the whole source behind one descriptor (the planner keeps it below 2^31 bytes); pixels beyond the end get an offset past num_records
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)((unsigned)total * (unsigned)a.src_cs * 2u), 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    const bool tail_oob = (cib - 1) * 32 + 8 * (int)g >= a.cin4;    // last k-block: this lane's 8 channels lie beyond round_up(Cin, 8)
    const int n_wfrag = a.n_ctiles;                                  // cout tiles; fragment (ctile, kb) at (ctile * cib + kb) KiB

    // ---- pixel stream: X-chunk (jX, cX) = two k-blocks (128 bytes per pixel) of this wave's 64 pixels.  A unit counts 2 * n_chunks X-chunks
    // (those beyond cib are dummies).  Registers hold the X-chunk AFTER the one in the wave's LDS image.
    const int nxc = 2 * n_chunks;
    int jX = 0, cX = 0;
    unsigned xv = kOOB; int nvi = 0;                             // lane's byte offset of (first pixel + lane / 8, slot lane % 8); load instructions whose pixel exists
    const unsigned row8 = (unsigned)a.src_cs * 16u;              // bytes between the pixels of consecutive load instructions (8 pixels)
    unsigned xv2[UP ? 8 : 1];                                    // UP: byte offset of instruction i's pixel in the half-resolution tensor
    const __amdgpu_buffer_rsrc_t x2rs = UP ? __builtin_amdgcn_make_buffer_rsrc((void*)a.src2, 0, (int)((unsigned)(total >> 2) * (unsigned)a.src2_cs * 2u), 0x00020000) : xrs;
    auto x_unit = [&]() {
        int tile, cg;
        decode(jX, tile, cg);
        const int p0 = tile * 256 + wave * 64 + (lane >> 3);
        xv = (unsigned)__mul24(p0 < total ? p0 : 0, a.src_cs) * 2u + (unsigned)(lane & 7) * 16u;
        const int left = total - p0;                             // pixels p0, p0 + 8, ...: instruction i is valid while 8 i < left
        nvi = left <= 0 ? 0 : min(8, (left + 7) >> 3);
        if constexpr (UP) {
            const unsigned W2 = (unsigned)a.up_W >> 1, H2 = (unsigned)a.up_H >> 1, HW = (unsigned)(a.up_W * a.up_H);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned p = (unsigned)min(p0 + 8 * i, total - 1);
                const unsigned b = fastdiv(p, FastDiv{a.fd_ty.ml, a.fd_ty.mh}), r = p - b * HW;
                const unsigned y = fastdiv(r, FastDiv{a.fd_tx.ml, a.fd_tx.mh}), x = r - y * (unsigned)a.up_W;
                xv2[i] = ((b * H2 + (y >> 1)) * W2 + (x >> 1)) * (unsigned)a.src2_cs * 2u + (unsigned)(lane & 7) * 16u;
            }
        }
    };
    f16x8 xp[8];
    auto x_load = [&]() {
        const bool live = jX < my_units;
        if (live && cX == 0) x_unit();
        const int c0 = cX * 64 + 8 * (lane & 7);                 // this lane's first channel
        const bool chan_ok = live && c0 < a.cin4;
        const int so = min(cX, (cib - 1) >> 1) * 128;
        if (UP && cX * 64 < a.up_c) {                            // wave-uniform: this X-chunk lives in the half-resolution tensor
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xp[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(x2rs, (int)((chan_ok && i < nvi) ? xv2[UP ? i : 0] : kOOB), so, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xp[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)((chan_ok && i < nvi) ? xv + (unsigned)i * row8 : kOOB), so, 0));
        }
        if (live && ++cX == nxc) { cX = 0; ++jX; }
    };
    unsigned char* ximg = lds + 2 * WBUF + wave * 8192;
    const unsigned wv0 = (unsigned)(lane >> 3) * 128u + ((((unsigned)lane & 7u) ^ (((unsigned)lane >> 4) & 7u)) * 16u);
    const unsigned wv1 = (unsigned)(lane >> 3) * 128u + ((((unsigned)lane & 7u) ^ ((((unsigned)lane >> 4) + 4u) & 7u)) * 16u);
    auto x_commit = [&]() {                                     // registers -> this wave's image (in order behind the wave's own reads of the old one)
#pragma unroll
        for (int i = 0; i < 8; ++i) *(f16x8*)(ximg + i * 1024 + ((i & 1) ? wv1 : wv0)) = xp[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const unsigned pr = (unsigned)lane & 15u, sz = (pr >> 1) & 7u;
    const unsigned rv0 = pr * 128u + ((g ^ sz) * 16u), rv1 = pr * 128u + (((4u + g) ^ sz) * 16u);
    auto xfrag = [&](int pt, int k2) -> f16x8 { return *(const f16x8*)__builtin_assume_aligned(ximg + pt * 2048 + (k2 ? rv1 : rv0), 16); };
    // ---- weight stream: chunk (jW, cW) = KC k-blocks x CT cout tiles, registers one chunk ahead of LDS, LDS one chunk ahead of the MFMAs ------
    int jW = 0, cW = 0, wct0 = 0;
    f16x8 pw[NWU];
    auto w_load = [&](int u) {                                  // fragment f = 4 u + wave of the chunk: (ct, kk) = (f / KC, f % KC)
        const int f = 4 * u + wave;
        const int ct = f / KC, kk = f - ct * KC;
        const int ctile = min(wct0 + ct, n_wfrag - 1), kb = min(cW * KC + kk, cib - 1);
        pw[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)((f < NFW && jW < my_units) ? lane16 : kOOB), (ctile * cib + kb) * 1024, 0));
    };
    auto w_begin = [&]() {                                      // before the loads of chunk (jW, cW)
        if (cW == 0 && jW < my_units) { int tile, cg; decode(jW, tile, cg); wct0 = cg * CT; }
    };
    auto w_advance = [&]() { if (jW < my_units && ++cW == n_chunks) { cW = 0; ++jW; } };
    auto w_commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NWU; ++u) *(f16x8*)(lds + buf * WBUF + u * 4096 + tid * 16) = pw[u];
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto epilogue = [&](int j) {
        int tile, cg;
        decode(j, tile, cg);
        const int ct0 = cg * CT;
        const OutF16 o = make_out_f16(a.dst, a.dst_cs, total * a.dst_cs, a.res, a.res_cs, total * a.res_cs, 0, a.Cout, a.act, a.out_f32);
        int pixi[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int p = tile * 256 + (wave * PT + pt) * 16 + (lane & 15);
            pixi[pt] = p < total ? p : -1;
        }
        f32x4 bias4[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int ctile = (ct0 + ct) < a.n_ctiles ? (ct0 + ct) : (a.n_ctiles - 1);
            bias4[ct] = *(const f32x4*)(a.bias + tile_cout0(ctile, lane >> 4, conv_f16_pairs(a.Cout)));
        }
        store_tiles_f16_v2<PT, CT>(o, acc, bias4, lane, ct0, pixi);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // prologue: weights of chunk 0 into LDS buffer 0, chunk 1 into registers; the first chunk's pixel k-blocks into the ring
    w_begin();
#pragma unroll
    for (int u = 0; u < NWU; ++u) w_load(u);
    w_advance();
    w_commit(0);
    w_begin();
#pragma unroll
    for (int u = 0; u < NWU; ++u) w_load(u);
    w_advance();
    x_load();
    x_commit();
    x_load();
    __syncthreads();
    const unsigned wl = lane16;
    int buf = 0;
    for (int j = 0; j < my_units; ++j) {
        for (int c = 0; c < n_chunks; ++c) {
            const int nkk = min(KC, cib - c * KC);
            const unsigned wb = (unsigned)buf * (unsigned)WBUF + wl;
            // chunk (j, c) from LDS buffer `buf`.  Inside its MFMA phase: the next chunk's weights registers -> the other buffer (every wave
            // has passed the barrier below, hence is done with that buffer), then the chunk after that into the registers.
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                if (kk < nkk) {
                    f16x8 wf[CT], xf[PT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) wf[ct] = *(const f16x8*)__builtin_assume_aligned(lds + wb + (ct * KC + kk) * 1024, 16);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) xf[pt] = xfrag(pt, kk & 1);
                    if (kk == 0) {
                        w_commit(buf ^ 1);
                        w_begin();
#pragma unroll
                        for (int u = 0; u < NWU; ++u) w_load(u);
                        w_advance();
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct], xf[pt], acc[ct][pt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kk & 1) {                                   // an X-chunk is consumed: the next one registers -> image, the one after into the registers
                    x_commit();
                    x_load();
                }
            }
            __syncthreads();                                   // everyone is done with `buf`; the other buffer's chunk is complete
            buf ^= 1;
        }
        epilogue(j);
    }
}

typedef void (*KernelFn)(ConvKArgs);

}  // namespace

// version-10 launch plans (pointwise): CT 3 or 6; up: with the nearest-2x upsample fused into the read side
const void* pick_conv1x1_lwx_f16(int CT, bool up) {
    if (up) {
        if (CT == 3) return (const void*)(KernelFn)&conv1x1_lwx_f16<3, true>;
        if (CT == 6) return (const void*)(KernelFn)&conv1x1_lwx_f16<6, true>;
        return nullptr;
    }
    if (CT == 3) return (const void*)(KernelFn)&conv1x1_lwx_f16<3>;
    if (CT == 6) return (const void*)(KernelFn)&conv1x1_lwx_f16<6>;
    return nullptr;
}

// version-7 launch plans: CT cout tiles per block (48 / 64 / 96 couts)
const void* pick_conv_lw_f16(int CT) {
    if (CT == 3) return (const void*)(KernelFn)&conv3x3_lw_f16<3>;
    if (CT == 4) return (const void*)(KernelFn)&conv3x3_lw_f16<4>;
    if (CT == 6) return (const void*)(KernelFn)&conv3x3_lw_f16<6>;
    return nullptr;
}

}  // namespace mi355
