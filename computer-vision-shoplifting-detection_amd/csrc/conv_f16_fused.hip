// half=True path: instances of conv_igemm_f16<3, S, PT, CT, WP, F2 = true> (Conv3x3 + SiLU -> Conv1x1 in one launch).
#include "conv_f16.h"

namespace mi355 {
namespace {
typedef void (*KernelFn)(ConvKArgs);

template <int STRIDE>
KernelFn pick_fused_h(int CT, int WP, int PT) {
#define MI355_CASEF(pt, ct, wp) if (PT == pt && CT == ct && WP == wp) return &conv_igemm_f16<3, STRIDE, pt, ct, wp, true>;
    MI355_CASEF(4, 1, 1) MI355_CASEF(4, 2, 1) MI355_CASEF(4, 3, 1) MI355_CASEF(4, 4, 1)
    MI355_CASEF(4, 2, 2) MI355_CASEF(4, 3, 2) MI355_CASEF(4, 4, 2)
    MI355_CASEF(4, 3, 4) MI355_CASEF(4, 4, 4)
    MI355_CASEF(8, 1, 1) MI355_CASEF(8, 2, 1) MI355_CASEF(8, 3, 1) MI355_CASEF(8, 4, 1)
    MI355_CASEF(8, 2, 2) MI355_CASEF(8, 3, 2) MI355_CASEF(8, 4, 2)
    MI355_CASEF(8, 3, 4) MI355_CASEF(8, 4, 4)
#undef MI355_CASEF
    return nullptr;
}
}  // namespace

const void* pick_conv_fused_f16(int stride, int CT, int WP, int PT) {
    const int pt = PT ? PT : 4;
    return stride == 1 ? (const void*)pick_fused_h<1>(CT, WP, pt) : stride == 2 ? (const void*)pick_fused_h<2>(CT, WP, pt) : nullptr;
}

}  // namespace mi355
