// Host side of the conv kernels: weight packing into MFMA fragment order, the launch planner (tile shape x wave
// arrangement x staged channels x kernel version per conv and shape; the engine times the best candidates) and run_conv.
// Device code: conv_f32.h (instances: conv_f32_*.hip) and conv_igemm_f16.hip.
#include "common.h"
#include "conv_f32_inst.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace mi355 {

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

struct Plan { int CT, WP, TW, TH, ck; size_t lds; double cost; int version; int buf_floats; int PT; int G = 1; bool f2 = false; };
// PT 0 = default (4, or 3 with CT 5); G = groups of CT cout tiles a wave walks over one staged input; f2 = fused pointwise stage

KernelFn pick_kernel(int ks, int stride, int CT, int WP, int PT) {
    if (ks == 1 && stride == 1) return pick_f32_k1(CT, WP, PT);
    if (ks == 3 && stride == 1) return pick_f32_k3s1(CT, WP, PT);
    if (ks == 3 && stride == 2) return pick_f32_k3s2(CT, WP, PT);
    return nullptr;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

constexpr size_t LDS_SOFT = 40 * 1024, LDS_HARD = 64 * 1024;

// Padding of a staged pixel's channel record in LDS, in 4-byte units.  ds_read_b128 is served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32): a B-operand read (lane (p, g): 16 bytes of pixel p at channel group g)
// is conflict-free when the pixel stride is 32 bytes mod 64 -- with the 16-byte pad of rounds 1-3 (stride 16 mod 64) every such read
// took two passes (SQ_LDS_BANK_CONFLICT = 45-62 % of the LDS cycles of every conv kernel, fp32 and fp16).  MI355_LDS_PAD=4 restores it.
int lds_pad() { static const int v = env_int("MI355_LDS_PAD", 8); return v; }

// Candidate launch plans for one conv: for every wave arrangement (CT, WC) the best output tile, with every
// feasible staged-channel count.  Sorted by a static cost model; the engine may time the first few (autotune).
// Launch-time model used to rank candidates, in matrix-pipe cycles: a launch takes at least the padded MFMA work spread over
// the chip's 1024 SIMDs (throughput) and at least one wave's own serial MFMA chain plus a fixed start-up (staging latency,
// epilogue).  Large batches are throughput-bound and rank by padding waste as before; a batch-1 map has so few pixel tiles
// that the serial chain of a 64-pixel wave tile dominates, and plans with 1-2 pixel tiles per wave (more, shorter waves)
// move to the front.
constexpr double SIMDS = 1024.0, MFMA_CYCLES = 32.0, WAVE_STARTUP_CYCLES = 6000.0;

struct Shape { int H, W, n_ctiles, cin16, ks, stride; };

double latency_factor(const Shape& sh, long blocks, int PT, int CT, int split = 1) {
    const int kblocks = (sh.cin16 / 16 + split - 1) / split;                                          // 16-channel blocks per wave
    const double per_wave = (double)PT * CT * sh.ks * sh.ks * kblocks * 4 * MFMA_CYCLES;              // one wave's MFMA chain
    const double thru = (double)blocks * 4.0 * per_wave / SIMDS;
    return std::max(1.0, (per_wave + WAVE_STARTUP_CYCLES) / std::max(thru, 1.0));
}

// Candidate launch plans for one conv: for every wave arrangement (CT, WC) the best output tile, with every
// feasible staged-channel count.  Sorted by a static cost model; the engine may time the first few (autotune).
std::vector<Plan> enumerate_plans(int H, int W, int images, int n_ctiles, int cin, int ks, int stride, bool have_zero_page, bool half,
                                  int f2_cin16 = 0, bool need_v4 = false) {
    static const int max_ct = env_int("MI355_MAX_CT", 5);          // tuning knobs (experiments only)
    static const int min_wc = env_int("MI355_MIN_WC", 1);
    static const int small_pt = env_int("MI355_SMALL_PT", 1);      // 0: never offer the 1- / 2-pixel-tile wave shapes
    std::vector<Plan> out;
    const int cin16 = round_up(cin, 16);
    const Shape sh{H, W, n_ctiles, cin16, ks, stride};
    // Is the default wave tile (4 pixel tiles) latency-bound on this shape?  Only then are the small wave tiles offered: at
    // large batches they would only crowd the autotuner's candidate list.
    // ... or leaves the chip thin: fewer than ~6 waves per SIMD in all, where shorter waves also hide each other's latencies better.
    const long blocks_default = ((long)W * H + 63) / 64 * images * ((n_ctiles + 3) / 4);
    const bool latency_bound = !half && small_pt && (latency_factor(sh, blocks_default, 4, 1) > 1.0 || blocks_default < 1536);
    // Pixel tiles per wave.  fp16: wave tiles of 8 pixel tiles (128 pixels x CT*16 couts) halve the weight bytes fetched per
    // MFMA -- the f16 MFMA retires a 1-KiB fragment pair in 16 cycles, so those kernels are bound by L1/L2 fragment traffic.
    // fp32: 0 = the default (4, or 3 with CT 5); 2 and 1 for latency-bound launches.
    std::vector<int> pt_sel = {0};
    static const int small_k_pt2 = env_int("MI355_SMALLK_PT2", 1);
    // fp16, thin launches (the default wave tile makes fewer than ~1.5 waves per SIMD: config 5 at its stated 2 frames per GPU): also
    // 2 and 1 pixel tiles per wave -- more, shorter waves (conv_f16_small.hip)
    static const int f16_small = env_int("MI355_F16_SMALL_PT", 1);
    const bool thin_f16 = half && f16_small && blocks_default < 1536 && !f2_cin16;
    if (half) { pt_sel.push_back(8); if (thin_f16) { pt_sel.push_back(2); pt_sel.push_back(1); } }
    else if (latency_bound) { pt_sel.push_back(2); pt_sel.push_back(1); }
    else if (small_k_pt2 && ks == 3 && cin16 * ks * ks <= 288) pt_sel.push_back(2);   // short K loops: staging + epilogue weigh as much
                                                                                     // as the MFMAs, so more (narrower) waves per SIMD pay
    for (int PTsel : pt_sel)
    for (int WC = 1; WC <= 4; WC *= 2)
        for (int CT = 1; CT <= 5; ++CT) {
            if (CT > max_ct || WC < min_wc) continue;
            if (PTsel == 8 && CT > 4) continue;
            if (PTsel != 0 && PTsel != 8 && CT > 2) continue;          // small wave tiles exist for CT 1 and 2
            const int WP = 4 / WC, PT = PTsel ? PTsel : (CT == 5 ? 3 : 4), P = WP * PT * 16;
            // cout groups: G = 1 (one group; more cout tiles = more blocks along grid.y, each staging the input again) or, when
            // all of Cin is staged at once, G = as many groups as cover every cout tile from ONE staged input (fp32 kernels)
            const int g_full = (n_ctiles + CT * WC - 1) / (CT * WC);
            for (int G : (half || g_full < 2 || g_full > 8) ? std::vector<int>{1} : std::vector<int>{1, g_full}) {
            const int cover = CT * WC * G, nblk = (n_ctiles + cover - 1) / cover;
            if (cover >= 2 * n_ctiles && cover > CT) continue;          // more than half of the cout tiles would be padding
            if (f2_cin16 && nblk != 1) continue;                        // a fused pointwise stage needs ALL first-conv channels in the block
            const double waste_c = (double)nblk * cover / n_ctiles;
            // staged channels per chunk, in 4-byte units; the fp16 kernels (2 channels per unit) also get 128: their K loop
            // is so short that the two barriers + pipeline refill per chunk show, above all in the 1x1 layers
            for (int ck = half ? 128 : 64; ck >= 16; ck >>= 1) {
                if (ck > cin16 && ck != 16) continue;
                if (G > 1 && ck < cin16) continue;                      // groups re-run the K loop over ONE staged chunk
                Plan best{}; best.cost = 1e30;
                for (int TW = 1; TW <= P && TW <= W; ++TW) {
                    int TH = P / TW; if (TH > H) TH = H;
                    if (TH < 1) continue;
                    const long tiles = (long)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
                    const int THin = (TH - 1) * stride + ks, TWin = (TW - 1) * stride + ks;
                    const int stage_floats = round_up(THin * TWin * (ck + lds_pad()), 4);
                    // fused pointwise stage: the first conv's output image [P pixels][f2_cin16 + 4] lives behind the halo tile
                    // (fp16: [P][round_up(C1, 32) + 8] halfs)
                    const size_t lds = (size_t)stage_floats * 4 + (!f2_cin16 ? 0 : half ? (size_t)P * (round_up(f2_cin16, 32) + 2 * lds_pad()) * 2
                                                                                        : (size_t)P * (f2_cin16 + lds_pad()) * 4);
                    if (lds > LDS_HARD) continue;
                    const double infl = waste_c * (double)tiles * P / ((double)W * H);
                    const double halo = (double)THin * TWin / ((double)TH * TW * stride * stride);
                    const int stages = (cin16 + ck - 1) / ck;
                    double cost = infl * (1.0 + 0.03 * halo * nblk) * (1.0 + 0.04 * (stages - 1)) * (1.0 + 0.04 * (CT - 1))
                                  + (lds > LDS_SOFT ? 0.15 : 0.0);
                    if (latency_bound) cost *= latency_factor(sh, tiles * images * nblk, PT, CT * G);
                    if (cost < best.cost) { best = Plan{CT, WP, TW, TH, ck, lds, cost, 1, f2_cin16 ? stage_floats : 0, PTsel}; best.G = G; best.f2 = f2_cin16 != 0; }
                }
                if (best.cost < 1e30) out.push_back(best);
                // fused pointwise stage, one cout group: the first conv's output image may take the halo tile's place in LDS (the tile is
                // dead once the K loop is over; one more barrier) -- LDS per block = max instead of sum, i.e. more blocks per CU
                static const int alias_on = env_int("MI355_F2_ALIAS", 1);
                if (alias_on && best.cost < 1e30 && best.f2 && G == 1) {
                    Plan al = best;
                    const size_t stage_b = (size_t)al.buf_floats * 4, img_b = al.lds - stage_b;
                    if (img_b > 0 && std::max(stage_b, img_b) + 4096 <= al.lds) {          // worth a plan only if it frees a few KiB
                        al.lds = std::max(stage_b, img_b); al.buf_floats = 0; al.cost = best.cost * 0.999;
                        out.push_back(al);
                    }
                }
            }
            }
        }
    if (f2_cin16) {                       // only the LDS-staged kernel has the fused form
        std::sort(out.begin(), out.end(), [](const Plan& a, const Plan& b) { return a.cost < b.cost; });
        return out;
    }
    if (ks == 3 && latency_bound && cin16 >= 64) {
        // v6: the four waves of a block split the 16-channel blocks of the SAME PT x CT tiles (bit-exact: the canonical order
        // sums block partials); LDS = halo tile of one chunk + cib x CT x PT partial tiles of 1 KiB
        static const int use_v6 = env_int("MI355_CONV_V6", 1);
        const int cib = cin16 / 16;
        for (int PT = 1; PT <= 2 && use_v6; ++PT)
            for (int CT = 1; CT <= 2; ++CT) {
                if (CT > n_ctiles) continue;
                const int P = PT * 16, nblk = (n_ctiles + CT - 1) / CT;
                const double waste_c = (double)nblk * CT / n_ctiles;
                for (int ck = 256; ck >= 64; ck >>= 1) {
                    if (ck > cin16 && ck != 64) continue;
                    Plan best{}; best.cost = 1e30;
                    for (int TW = 1; TW <= P && TW <= W; ++TW) {
                        int TH = P / TW; if (TH > H) TH = H;
                        if (TH < 1) continue;
                        const long tiles = (long)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
                        const int THin = (TH - 1) * stride + ks, TWin = (TW - 1) * stride + ks;
                        const int stage_floats = round_up(THin * TWin * (ck + lds_pad()), 4);
                        const size_t lds = (size_t)stage_floats * 4 + (size_t)cib * CT * PT * 1024;
                        if (lds > LDS_HARD) continue;
                        const double infl = waste_c * (double)tiles * P / ((double)W * H);
                        const int stages = (cin16 + ck - 1) / ck;
                        double cost = infl * (1.0 + 0.04 * (stages - 1)) * latency_factor(sh, tiles * images * nblk, PT, CT, 4);
                        if (cost < best.cost) best = Plan{CT, 1, TW, TH, ck, lds, cost, 6, stage_floats, PT};
                    }
                    if (best.cost < 1e30) out.push_back(best);
                }
            }
    }
    if (half && ks == 3 && stride == 1 && H >= 8 && W >= 8) {
        // v7 (conv_f16_lw.hip): 16 x 16 output tile x CT * 16 couts per block, weights through LDS, persistent blocks with prefetch
        // across k-blocks and tiles.  ck = one 32-channel k-block (16 four-byte units); LDS is static in the kernel.
        static const int use_v7 = env_int("MI355_CONV_V7", 1);
        const int cts[3] = {6, 4, 3};
        const long tiles = (long)((W + 15) / 16) * ((H + 15) / 16);
        for (int ci = 0; ci < 3 && use_v7; ++ci) {
            const int CT = cts[ci];
            if (CT > n_ctiles && !(CT == 3 && n_ctiles >= 2)) continue;
            const int nblk = (n_ctiles + CT - 1) / CT;
            if (nblk * CT >= 2 * n_ctiles && nblk * CT > CT) continue;
            Plan p7{CT, 4, 16, 16, 16, 0, 0.0, 7, 0, 4};
            p7.cost = (double)nblk * CT / n_ctiles * (double)tiles * 256.0 / ((double)W * H) * 0.7;
            out.push_back(p7);
        }
    }
    if (half && ks == 1 && stride == 1) {                        // also with the fused upsample (need_v4: the conv reads through an upsample)
        // v10 (conv_f16_lw.hip: conv1x1_lwx_f16): 256 flattened pixels x CT * 16 couts per block, the block's weights through a double-buffered LDS
        // region, each wave's pixels staged in full cache lines through a wave-private LDS image, persistent blocks.  Not with the fused upsample.
        static const int use_v10 = env_int("MI355_CONV_V10", 1);
        const int cts[2] = {6, 3};                              // 64- and 32-cout blocks never won a shape (tools/r04_call31.sh)
        for (int ci = 0; ci < 2 && use_v10; ++ci) {
            const int CT = cts[ci];
            if (CT > n_ctiles) continue;
            const int nblk = (n_ctiles + CT - 1) / CT;
            if (nblk * CT >= 2 * n_ctiles && nblk * CT > CT) continue;
            Plan p10{CT, 4, 256, 1, 16, 0, 0.0, 10, 0, 4};
            p10.cost = (double)nblk * CT / n_ctiles * (1.0 + 0.05 * (nblk - 1)) * 0.7;
            out.push_back(p10);
        }
    }
    if (ks == 1 && have_zero_page) {        // streaming pointwise kernel (needs the zero page as well): CT x PT register tiles
        static const int use_v3 = env_int("MI355_CONV_V3", 1);
        const int cts[3] = {1, 2, 4}, pts[3] = {2, 4, 1};
        for (int ci = 0; ci < 3 && use_v3; ++ci)
            for (int pi = 0; pi < ((latency_bound && !half) ? 3 : 2); ++pi) {
                const int CT = cts[ci], PT = pts[pi];
                if (CT > n_ctiles && CT != 1) continue;
                const int nblk = (n_ctiles + CT - 1) / CT;
                Plan p3{CT, 4, PT * 64, 1, 16, 0, 0.0, 3, PT, 0};
                p3.cost = (double)nblk * CT / n_ctiles * (1.0 + 0.05 * nblk) * 0.9;
                if (latency_bound) p3.cost *= latency_factor(sh, ((long)W + PT * 64 - 1) / (PT * 64) * nblk, PT, CT);
                out.push_back(p3);
            }
    }
    // Latency-bound launches: the pipelined kernel (two barriers per item, a prologue of its own) never won there (10+ us against
    // 6 us for the streaming kernel on the 20x20 .. 80x80 maps of a single frame) -- and its many shapes crowded the streaming
    // kernel's best variants out of the candidates the autotuner times (the list is cut at 32): offered only where the conv needs
    // it (the upsample fused into its read side exists in this kernel alone).
    if (ks == 1 && have_zero_page && (!latency_bound || need_v4 || half)) {   // persistent software-pipelined pointwise kernel (v4); ck in 4-byte units (fp16: 2 channels each)
        static const int use_v4 = env_int("MI355_CONV_V4", 1);
        const int wps[3] = {1, 2, 4}, cts[4] = {1, 2, 4, 3}, cks[4] = {128, 64, 32, 16};
        for (int PT : (latency_bound && !half) ? std::vector<int>{4, 2, 1} : std::vector<int>{4})
        for (int wi = 0; wi < 3 && use_v4; ++wi)
            for (int ci = 0; ci < (half ? 4 : 3); ++ci)
                for (int ki = 0; ki < 4; ++ki) {
                    const int WP = wps[wi], WC = 4 / WP, CT = cts[ci], ck = cks[ki], P = WP * PT * 16;
                    const int cover = CT * WC, nblk = (n_ctiles + cover - 1) / cover;
                    if (cover >= 2 * n_ctiles && cover > CT) continue;
                    if (ck > cin16 && ck != 16) continue;
                    if (P * ck / 4 > 2048) continue;                       // 8 prefetch registers (float4) per thread
                    if (ck > 64 && (CT > (half ? 3 : 2) || PT != 4)) continue;   // a chunk's weights live in registers: 8 k-blocks x CT fragments
                    const size_t lds = (size_t)P * (ck + lds_pad()) * 4;
                    const int stages = (cin16 + ck - 1) / ck;
                    Plan p4{CT, WP, P, 1, ck, lds, 0.0, 4, 0, PT == 4 ? 0 : PT};
                    p4.cost = (double)nblk * cover / n_ctiles * (1.0 + 0.03 * (stages - 1)) * (1.0 + 0.02 * nblk) * 0.8;
                    if (latency_bound) p4.cost *= latency_factor(sh, ((long)W + P - 1) / P * nblk, PT, CT);
                    out.push_back(p4);
                }
    }
    std::sort(out.begin(), out.end(), [](const Plan& a, const Plan& b) { return a.cost < b.cost; });
    if (latency_bound || thin_f16) {
        // the latency model is crude: let the timed top-N hold both families, best of each alternating
        std::vector<Plan> big, small, merged;
        for (const Plan& p : out) ((p.PT == 0 || p.PT >= 4) && p.version != 6 && !(p.version == 3 && p.buf_floats == 1) ? big : small).push_back(p);
        for (size_t i = 0; i < std::max(big.size(), small.size()); ++i) {
            if (i < small.size()) merged.push_back(small[i]);
            if (i < big.size()) merged.push_back(big[i]);
        }
        out.swap(merged);
    }
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
#ifdef MI355_F32_DIAG
    if (c.dtype != 1 && p.version == 1) { static const int ex32 = env_int("MI355_F32_EXP", 0); a.act |= ex32 << 8; }
#endif
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
    KernelFn fn = half ? (p.version == 7 ? (KernelFn)pick_conv_lw_f16(p.CT) : p.version == 10 ? (KernelFn)pick_conv1x1_lwx_f16(p.CT, c.src2 != nullptr) : p.version == 4 ? (KernelFn)pick_conv_pipe_f16(p.CT, p.WP, (c.Cin + 1) / 2 <= p.ck, p.ck > 64)
                          : p.f2 ? (KernelFn)pick_conv_fused_f16(c.stride, p.CT, p.WP, p.PT)
                                 : (p.version == 1 && (p.PT == 1 || p.PT == 2)) ? (KernelFn)pick_conv_small_f16(c.k, c.stride, p.CT, p.WP, p.PT)
                                 : (KernelFn)pick_conv_kernel_f16(c.k, c.stride, p.CT, p.WP, p.version, p.version == 3 ? p.buf_floats : p.PT))
                       : (p.version == 3 ? (c.src2 ? pick_f32_stream_up(p.CT, p.buf_floats) : pick_f32_stream(p.CT, p.buf_floats))
                          : p.version == 4 ? pick_f32_pipe(p.CT, p.WP, c.Cin <= p.ck, p.ck, p.PT)
                          : p.version == 6 ? pick_f32_splitk(c.stride, p.CT, p.PT)
                          : p.f2 ? (c.stride == 1 ? pick_f32_fused_s1(p.CT, p.WP, p.PT) : pick_f32_fused_s2(p.CT, p.WP, p.PT))
                          : pick_kernel(c.k, c.stride, p.CT, p.WP, p.PT));
    if (!fn) return "conv: no kernel instance";
    a.zeros = c.zeros; a.lds_buf_floats = p.buf_floats;
    a.cgroups = std::max(1, p.G);
    if (p.f2) {
        a.w2 = c.f2_wpk; a.bias2 = c.f2_bias; a.dst2 = c.f2_dst; a.dst2_cs = c.f2_dst_cs; a.Cout2 = c.f2_cout; a.act2 = c.f2_act;
        a.n_ctiles2 = (c.f2_cout + 15) / 16; a.cib2 = (c.Cout + 15) / 16; a.ldp2 = round_up(c.Cout, 16) + lds_pad();
        if (half) { a.cib2 = (c.Cout + 31) / 32; a.ldp2 = 32 * a.cib2 + 2 * lds_pad(); a.out2_f32 = c.f2_out_f32; }
        if (c.f2_lead_c) { a.lead = c.f2_lead; a.lead_cs = c.f2_lead_cs; a.lead_cib = c.f2_lead_c / 16; a.cib2 += a.lead_cib; }
    }
    if (half && p.version == 4) a.lds_buf_floats = 0;
    if (half && (p.version == 1 || p.version == 7) && !p.f2) { static const int ex = env_int("MI355_F16_EXP", 0); a.lds_buf_floats = ex; }
    a.TW = p.TW; a.TH = p.TH;
    a.tiles_x = (a.Wout + p.TW - 1) / p.TW; a.tiles_y = (a.Hout + p.TH - 1) / p.TH;
    a.TWin = (p.TW - 1) * c.stride + c.k;
    const int THin = (p.TH - 1) * c.stride + c.k;
    a.npix_in = a.TWin * THin;
    a.inv_TW = 1.0f / (float)p.TW; a.inv_TWin = 1.0f / (float)a.TWin;
    if (p.version == 4 && a.up_c) { a.inv_TW = 1.0f / (float)c.Win; a.inv_TWin = 1.0f / (float)c.Hin; }   // v4 has no other use for them
    // plans count staged channels in 4-byte units; the fp16 kernels stage twice as many channels in the same bytes
    a.ck = half ? 2 * p.ck : p.ck; a.ldp = half ? a.ck + 2 * lds_pad() : a.ck + lds_pad();
    a.ck4_shift = 0;
    while ((4 << a.ck4_shift) < p.ck) ++a.ck4_shift;               // log2 of the 16-byte slots per staged pixel
    const int WC = p.version == 6 ? 1 : 4 / p.WP;
    out->fn = (const void*)fn;
    a.n_tiles_total = (int)((long)B * a.tiles_x * a.tiles_y);
    out->grid_x = (unsigned)a.n_tiles_total;
    if (p.version == 4) {
        // persistent: exactly as many blocks as stay resident (asked of the runtime: registers, LDS), each walks tiles
        // blockIdx.x, + gridDim.x, ...; blocks that had to queue behind others would leave CUs half empty at the end
        const int gy = std::max(1, (a.n_ctiles + p.CT * (4 / p.WP) - 1) / (p.CT * (4 / p.WP)));
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, 256, p.lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = std::max(1, std::min(3, (int)((size_t)(160 * 1024) / std::max<size_t>(p.lds, 1))));
        }
        out->grid_x = std::min(out->grid_x, (unsigned)std::max(1, 256 * per_cu / gy));
        // workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest): with grid.x a multiple of 8 the
        // gy cout groups of one tile column run on the same XCD and stage the same pixels through one L2
        static const int gx8 = env_int("MI355_PIPE_GX8", 1);
        if (gx8 && gy > 1 && out->grid_x >= 16) out->grid_x &= ~7u;
    }
    out->grid_y = (unsigned)((a.n_ctiles + p.CT * WC * a.cgroups - 1) / (p.CT * WC * a.cgroups));
    unsigned v7_gy = 1;
    if (p.version == 7 || p.version == 10) {
        // persistent: as many blocks as stay resident (two per CU), each walks units blockIdx.x, + gridDim.x, ...; unit = (tile, cout group)
        if (p.version == 7 && c.pad != 1) return "conv: the LDS-weights kernel needs pad 1";
        if (p.version == 10 && c.src2) {      // upsample fused into the read side: whole 64-channel X-chunks from the half-resolution tensor
            if ((c.up_c & 63) || (long long)(a.Wout / 4) * c.src2_cs * 2 >= (1ll << 31)) return "conv: the LDS-weights pointwise kernel cannot fuse this upsample";
        }
        v7_gy = (unsigned)((a.n_ctiles + p.CT - 1) / p.CT);
        a.cgroups = (int)v7_gy;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, 256, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; }
        const unsigned long long units = (unsigned long long)a.n_tiles_total * v7_gy;
        if (units >= (1u << 24)) return "conv: more than 2^24 work units in one launch";
        out->grid_x = (unsigned)std::min<unsigned long long>(units, 256ull * (unsigned)per_cu);
        if (out->grid_x >= 16) out->grid_x &= ~7u;
        out->grid_y = 1;
    }
    if ((unsigned long long)out->grid_x * out->grid_y >= (1u << 24)) return "conv: more than 2^24 blocks in one launch";
    a.fd_tx = make_fastdiv((unsigned)std::max(1, a.tiles_x)); a.fd_ty = make_fastdiv((unsigned)std::max(1, a.tiles_y));
    a.fd_gy = make_fastdiv(std::max(1u, (p.version == 7 || p.version == 10) ? v7_gy : out->grid_y));
    if (!half && (p.version == 1 || p.f2)) {
        // conv_igemm_f32 addresses one image of each slice through a buffer descriptor with 32-bit byte offsets (a pointwise
        // launch sees the flattened batch as one image), and lanes / pad channels whose store must be DROPPED are given the
        // byte offset 0x80000000: that marker is out of range only while num_records <= 2^31 bytes, i.e. while every slice
        // image -- source, destination, residual and the fused stage's destination -- stays below 2^29 elements.  Beyond that
        // such a plan is not offered (the streaming / pipelined pointwise kernels use per-block descriptors and remain).
        const long long lim = 1ll << 29;
        const long long e_src = (long long)a.Hin * a.Win * a.src_cs, e_dst = (long long)a.Hout * a.Wout * std::max(a.dst_cs, a.res_cs);
        const long long e_dst2 = p.f2 ? (long long)a.Hout * a.Wout * std::max(a.dst2_cs, a.lead_cs) : 0;
        if (e_src >= lim || e_dst >= lim || e_dst2 >= lim) return "conv: image too large for the 32-bit offsets of conv_igemm_f32";
    }
    a.img_src = a.Hin * a.Win * a.src_cs; a.img_dst = a.Hout * a.Wout * a.dst_cs; a.img_res = a.Hout * a.Wout * a.res_cs;
    a.THin = THin;
    {
        const int row_slots = a.TWin << a.ck4_shift;
        a.st_rpi = row_slots >= 256 ? 1 : 256 / row_slots;
        a.st_nseg = row_slots >= 256 ? (row_slots + 255) / 256 : 1;
        a.inv_row_slots = 1.0f / (float)row_slots;
    }
    if (p.version == 10 && c.src2) { a.fd_tx = make_fastdiv((unsigned)c.Win); a.fd_ty = make_fastdiv((unsigned)(c.Win * c.Hin)); }
    if (half && (p.version == 1 || p.version == 7 || p.version == 10 || p.f2)) {
        // conv_igemm_f16 (round 3): source / destination / residual images behind buffer descriptors with 32-bit byte offsets and
        // the drop marker 0x80000000 -- every image must stay below 2^31 bytes, and a source row below 2^24 bytes (24-bit multiply)
        const long long lim = 1ll << 31;
        const long long esz_out = c.out_f32 ? 4 : 2, esz2 = c.f2_out_f32 ? 4 : 2;
        const long long b_src = (long long)a.Hin * a.Win * a.src_cs * 2, b_dst = (long long)a.Hout * a.Wout * std::max(a.dst_cs, a.res_cs) * esz_out;
        const long long b_dst2 = p.f2 ? (long long)a.Hout * a.Wout * a.dst2_cs * esz2 : 0;
        if (b_src >= lim || b_dst >= lim || b_dst2 >= lim || (long long)a.Win * a.src_cs * 2 >= (1ll << 24) && c.k != 1)
            return "conv: image too large for the 32-bit offsets of conv_igemm_f16";
    }
    out->lds = p.lds;
    out->a = a;
    out->CT = p.CT; out->WP = p.WP; out->version = p.version;
    out->PT = p.version == 3 ? p.buf_floats : (p.PT ? p.PT : (p.CT == 5 ? 3 : 4)); out->threads = 256;
    if (p.version == 3) {          // streaming 1x1: block = 4 waves x PT pixel tiles, grid.y over cout blocks of CT tiles
        const int PT = p.buf_floats;
        out->grid_x = (unsigned)((a.Wout + 4 * PT * 16 - 1) / (4 * PT * 16));
        out->grid_y = (unsigned)((a.n_ctiles + p.CT - 1) / p.CT);
        out->lds = 0;
        out->a.tiles_x = (int)out->grid_x;
        if (c.src2) {            // upsample fused into the read side: per-lane (image, row, column) of a flattened pixel by scalar-made magic numbers
            if ((c.up_c & 15) || (long long)(a.Wout / 4) * c.src2_cs * 4 >= (1ll << 31)) return "conv: streaming kernel cannot fuse this upsample";
            out->a.fd_tx = make_fastdiv((unsigned)c.Win); out->a.fd_ty = make_fastdiv((unsigned)(c.Win * c.Hin));
        }
    }
    out->flops = 2.0 * c.B * c.Hout * c.Wout * ((double)c.Cout * c.Cin * c.k * c.k + (p.f2 ? (double)c.f2_cout * (c.Cout + c.f2_lead_c) : 0.0));
    return nullptr;
}

// all candidate launches for one conv, best static guess first
const char* plan_conv_candidates(const ConvArgs& c, std::vector<ConvLaunch>* out) {
    if (const char* e = check_args(c)) return e;
    const int H = c.k == 1 ? 1 : c.Hout, W = c.k == 1 ? c.B * c.Hout * c.Wout : c.Wout;
    const bool half = c.dtype == 1;
    // H x W = the map one block grid walks: the image for 3x3 convs (`images` of them), batch x space flattened for 1x1
    if (c.f2_cout) {
        const bool wide2 = half && !c.f2_out_f32 && conv_f16_pairs(c.f2_cout);      // 16-byte fp16 stores of the second stage
        const int cs_mask = wide2 ? 7 : 3, ptr_mask = (!half || wide2 || c.f2_out_f32) ? 15 : 7;
        if (c.k != 3 || (half && c.res) || !c.f2_wpk || !c.f2_bias || !c.f2_dst || (c.f2_dst_cs & cs_mask) || ((uintptr_t)c.f2_dst & ptr_mask) ||
            (half && c.out_f32))
            return "conv: a fused pointwise stage needs a 3x3 conv (half=True: without residual) and aligned second-stage buffers";
        if (c.f2_lead_c && (half || (c.f2_lead_c & 15) || (c.Cout & 15) || !c.f2_lead || (c.f2_lead_cs & 3) || ((uintptr_t)c.f2_lead & 15)))
            return "conv: lead channels of a fused pointwise stage must be whole 16-channel blocks of an aligned fp32 slice";
    }
    const std::vector<Plan> plans = enumerate_plans(H, W, c.k == 1 ? 1 : c.B, (c.Cout + 15) / 16, half ? (c.Cin + 1) / 2 : c.Cin, c.k,
                                                    c.stride, c.zeros != nullptr, half, c.f2_cout ? round_up(c.Cout, 16) : 0, c.src2 != nullptr);
    if (plans.empty()) return "conv: no launch plan fits in LDS";
    const char* last_err = nullptr;
    for (const Plan& p : plans) {
        if (c.src2 && p.version != 4 && !(p.version == 3 && !half) && !(p.version == 10 && half)) continue;       // upsample-on-read: the pipelined kernels and the fp32 streaming kernel
        ConvLaunch l{};
        if (const char* e = build_launch(c, p, &l)) {   // e.g. the fused form exists for fewer wave shapes than the plain one
            last_err = e;
            continue;
        }
        out->push_back(l);
    }
    if (out->empty()) return last_err ? last_err : c.f2_cout ? "conv: no fused launch plan for this shape" : "conv: no launch plan supports the fused upsample";
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
