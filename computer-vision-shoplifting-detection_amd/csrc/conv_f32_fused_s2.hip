// Instances of conv_igemm_f32<3, 2, PT, CT, WP, F2 = true>: a 3x3 conv of stride 2 with a pointwise conv fused behind it
// (device code: conv_f32.h).
#include "conv_f32.h"
#include "conv_f32_inst.h"

namespace mi355 {

KernelFn pick_f32_fused_s2(int CT, int WP, int PT) {
    if (PT == 0) {
#define MI355_CASE(ct, wp) if (CT == ct && WP == wp) return &conv_igemm_f32<3, 2, (ct == 5 ? 3 : 4), ct, wp, true>;
        MI355_CASE(1, 4) MI355_CASE(2, 4) MI355_CASE(3, 4) MI355_CASE(4, 4) MI355_CASE(5, 4)
        MI355_CASE(1, 2) MI355_CASE(2, 2) MI355_CASE(3, 2) MI355_CASE(4, 2) MI355_CASE(5, 2)
        MI355_CASE(1, 1) MI355_CASE(2, 1) MI355_CASE(3, 1) MI355_CASE(4, 1) MI355_CASE(5, 1)
#undef MI355_CASE
        return nullptr;
    }
#define MI355_CASE(pt, ct, wp) if (PT == pt && CT == ct && WP == wp) return &conv_igemm_f32<3, 2, pt, ct, wp, true>;
    MI355_CASE(1, 1, 4) MI355_CASE(1, 2, 4) MI355_CASE(1, 1, 2) MI355_CASE(1, 2, 2) MI355_CASE(1, 1, 1) MI355_CASE(1, 2, 1)
    MI355_CASE(2, 1, 4) MI355_CASE(2, 2, 4) MI355_CASE(2, 1, 2) MI355_CASE(2, 2, 2) MI355_CASE(2, 1, 1) MI355_CASE(2, 2, 1)
#undef MI355_CASE
    return nullptr;
}

}  // namespace mi355
