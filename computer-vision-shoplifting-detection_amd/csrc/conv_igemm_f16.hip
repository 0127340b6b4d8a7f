// half=True path: kernel instances (LDS-staged 1x1 / 3x3, streaming and pipelined pointwise) and the fp16 weight packer.
// Device code: conv_f16.h; the fused Conv3x3 -> Conv1x1 instances live in conv_f16_fused.hip.
#include "conv_f16.h"

namespace mi355 {

// ---------------------------------------------------------------------------------------------- host side
size_t packed_weight_halfs(int cout, int cin, int k) {
    return (size_t)((cout + 15) / 16) * k * k * ((cin + 31) / 32) * 512;
}

// OIHW fp32 -> fp16 (round-to-nearest-even) in MFMA fragment order [cout_tile][tap][cin_block32][lane(64)][8]
void pack_conv_weights_f16(const float* w, int cout, int cin, int k, uint16_t* out_bits) {
    _Float16* out = (_Float16*)out_bits;
    const int nct = (cout + 15) / 16, cib = (cin + 31) / 32, taps = k * k;
    const bool pairs = conv_f16_pairs(cout);
    for (int ct = 0; ct < nct; ++ct)
        for (int tap = 0; tap < taps; ++tap)
            for (int cb = 0; cb < cib; ++cb) {
                _Float16* o = out + ((size_t)(ct * taps + tap) * cib + cb) * 512;
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 8; ++s) {
                        const int r = lane & 15;         // MFMA row of this cout tile
                        const int co = pairs ? ((ct >> 1) * 32 + 8 * (r >> 2) + 4 * (ct & 1) + (r & 3)) : (ct * 16 + r);
                        const int ci = cb * 32 + 8 * (lane >> 4) + s;
                        o[lane * 8 + s] = (co < cout && ci < cin) ? (_Float16)w[((size_t)co * cin + ci) * taps + tap] : (_Float16)0.f;
                    }
            }
}

void floats_to_halfs(const float* in, uint16_t* out_bits, size_t n) {
    _Float16* o = (_Float16*)out_bits;
    for (size_t i = 0; i < n; ++i) o[i] = (_Float16)in[i];
}

void halfs_to_floats(const uint16_t* in_bits, float* out, size_t n) {
    const _Float16* p = (const _Float16*)in_bits;
    for (size_t i = 0; i < n; ++i) out[i] = (float)p[i];
}

namespace {

typedef void (*KernelFn)(ConvKArgs);

template <int KS, int STRIDE>
KernelFn pick_ct_wp_h(int CT, int WP, int PT) {
    if (PT == 8) {
#define MI355_CASE8(ct, wp) if (CT == ct && WP == wp) return &conv_igemm_f16<KS, STRIDE, 8, ct, wp>;
        MI355_CASE8(1, 4) MI355_CASE8(2, 4) MI355_CASE8(3, 4) MI355_CASE8(4, 4)
        MI355_CASE8(1, 2) MI355_CASE8(2, 2) MI355_CASE8(3, 2) MI355_CASE8(4, 2)
        MI355_CASE8(1, 1) MI355_CASE8(2, 1) MI355_CASE8(3, 1) MI355_CASE8(4, 1)
#undef MI355_CASE8
        return nullptr;
    }
#define MI355_CASE(ct, wp) if (CT == ct && WP == wp) return &conv_igemm_f16<KS, STRIDE, (ct == 5 ? 3 : 4), ct, wp>;
    MI355_CASE(1, 4) MI355_CASE(2, 4) MI355_CASE(3, 4) MI355_CASE(4, 4) MI355_CASE(5, 4)
    MI355_CASE(1, 2) MI355_CASE(2, 2) MI355_CASE(3, 2) MI355_CASE(4, 2) MI355_CASE(5, 2)
    MI355_CASE(1, 1) MI355_CASE(2, 1) MI355_CASE(3, 1) MI355_CASE(4, 1) MI355_CASE(5, 1)
#undef MI355_CASE
    return nullptr;
}

}  // namespace

namespace {
template <bool SINGLE, int NKK>
KernelFn pick_pipe_h(int CT, int WP) {
#define MI355_CASEP(ct, wp) if (CT == ct && WP == wp) return &conv1x1_pipe_f16<4, ct, wp, SINGLE, NKK>;
    MI355_CASEP(1, 1) MI355_CASEP(2, 1) MI355_CASEP(3, 1) MI355_CASEP(4, 1)
    MI355_CASEP(1, 2) MI355_CASEP(2, 2) MI355_CASEP(3, 2) MI355_CASEP(4, 2)
    MI355_CASEP(1, 4) MI355_CASEP(2, 4) MI355_CASEP(3, 4) MI355_CASEP(4, 4)
#undef MI355_CASEP
    return nullptr;
}
}  // namespace

// v4 (pipelined pointwise): single = Cin fits one chunk; nkk8 = 8 k-blocks (256 channels) per chunk instead of <= 4
const void* pick_conv_pipe_f16(int CT, int WP, bool single, bool nkk8) {
    if (nkk8) {
        if (CT > 3) return nullptr;
        return single ? (const void*)pick_pipe_h<true, 8>(CT, WP) : (const void*)pick_pipe_h<false, 8>(CT, WP);
    }
    return single ? (const void*)pick_pipe_h<true, 4>(CT, WP) : (const void*)pick_pipe_h<false, 4>(CT, WP);
}

const void* pick_conv_kernel_f16(int ks, int stride, int CT, int WP, int version, int stream_pt /* v3: PT; v1: 0 or 8 */) {
    if (version == 3) {
        if (CT == 1 && stream_pt == 2) return (const void*)&conv1x1_stream_f16<2, 1>;
        if (CT == 1 && stream_pt == 4) return (const void*)&conv1x1_stream_f16<4, 1>;
        if (CT == 2 && stream_pt == 2) return (const void*)&conv1x1_stream_f16<2, 2>;
        if (CT == 2 && stream_pt == 4) return (const void*)&conv1x1_stream_f16<4, 2>;
        if (CT == 4 && stream_pt == 2) return (const void*)&conv1x1_stream_f16<2, 4>;
        if (CT == 4 && stream_pt == 4) return (const void*)&conv1x1_stream_f16<4, 4>;
        return nullptr;
    }
    if (version != 1) return nullptr;
    if (ks == 1 && stride == 1) return (const void*)pick_ct_wp_h<1, 1>(CT, WP, stream_pt);
    if (ks == 3 && stride == 1) return (const void*)pick_ct_wp_h<3, 1>(CT, WP, stream_pt);
    if (ks == 3 && stride == 2) return (const void*)pick_ct_wp_h<3, 2>(CT, WP, stream_pt);
    return nullptr;
}

}  // namespace mi355
