// Instances of conv_igemm_f32 for 3x3 convs of stride 2 (device code: conv_f32.h).
#include "conv_f32.h"
#include "conv_f32_inst.h"

namespace mi355 {

KernelFn pick_f32_k3s2(int CT, int WP, int PT) {
    if (PT == 0) {
#define MI355_CASE(ct, wp) if (CT == ct && WP == wp) return &conv_igemm_f32<3, 2, (ct == 5 ? 3 : 4), ct, wp>;
        MI355_CASE(1, 4) MI355_CASE(2, 4) MI355_CASE(3, 4) MI355_CASE(4, 4) MI355_CASE(5, 4)
        MI355_CASE(1, 2) MI355_CASE(2, 2) MI355_CASE(3, 2) MI355_CASE(4, 2) MI355_CASE(5, 2)
        MI355_CASE(1, 1) MI355_CASE(2, 1) MI355_CASE(3, 1) MI355_CASE(4, 1) MI355_CASE(5, 1)
#undef MI355_CASE
        return nullptr;
    }
    // small wave tiles (1 or 2 pixel tiles per wave) for latency-bound launches: a batch-1 map has few pixels, so the
    // planner trades register blocking for more waves
#define MI355_CASE(pt, ct, wp) if (PT == pt && CT == ct && WP == wp) return &conv_igemm_f32<3, 2, pt, ct, wp>;
    MI355_CASE(1, 1, 4) MI355_CASE(1, 2, 4) MI355_CASE(1, 1, 2) MI355_CASE(1, 2, 2) MI355_CASE(1, 1, 1) MI355_CASE(1, 2, 1)
    MI355_CASE(2, 1, 4) MI355_CASE(2, 2, 4) MI355_CASE(2, 1, 2) MI355_CASE(2, 2, 2) MI355_CASE(2, 1, 1) MI355_CASE(2, 2, 1)
#undef MI355_CASE
    return nullptr;
}

}  // namespace mi355
