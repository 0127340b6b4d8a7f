// half=True path: instances of conv_igemm_f16 with SMALL wave tiles (1 or 2 pixel tiles of 16 pixels per wave, 1 or 2 cout tiles) for
// launches that leave the chip thin -- BASELINE config 5 as stated is 2 frames of 1280x1280 per GPU: its 40x40 .. 80x80 maps make a few
// hundred 64-pixel wave tiles, one wave per SIMD or fewer, each walking its whole K loop alone.  Narrow tiles trade register blocking for
// two to four times as many waves (the fp32 path has had them since round 2: conv_f32_k3s1.hip).  Same kernel body as the wide tiles
// (conv_f16.h), so the same bits.
#include "conv_f16.h"

namespace mi355 {
namespace {
typedef void (*KernelFn)(ConvKArgs);

template <int KS, int STRIDE>
KernelFn pick_small_h(int CT, int WP, int PT) {
#define MI355_CASES(pt, ct, wp) if (PT == pt && CT == ct && WP == wp) return &conv_igemm_f16<KS, STRIDE, pt, ct, wp>;
    MI355_CASES(1, 1, 4) MI355_CASES(1, 2, 4) MI355_CASES(1, 1, 2) MI355_CASES(1, 2, 2) MI355_CASES(1, 1, 1) MI355_CASES(1, 2, 1)
    MI355_CASES(2, 1, 4) MI355_CASES(2, 2, 4) MI355_CASES(2, 1, 2) MI355_CASES(2, 2, 2) MI355_CASES(2, 1, 1) MI355_CASES(2, 2, 1)
#undef MI355_CASES
    return nullptr;
}
}  // namespace

const void* pick_conv_small_f16(int ks, int stride, int CT, int WP, int PT) {
    if (ks == 1 && stride == 1) return (const void*)pick_small_h<1, 1>(CT, WP, PT);
    if (ks == 3 && stride == 1) return (const void*)pick_small_h<3, 1>(CT, WP, PT);
    if (ks == 3 && stride == 2) return (const void*)pick_small_h<3, 2>(CT, WP, PT);
    return nullptr;
}

}  // namespace mi355
