// Instances of conv1x1_pipe_f32, the persistent software-pipelined pointwise kernel (device code: conv_f32.h).
#include "conv_f32.h"
#include "conv_f32_inst.h"

namespace mi355 {

template <int PT, bool SINGLE, int NKK>
static KernelFn pick_pipe_s(int CT, int WP) {
#define MI355_CASE4(ct, wp) if (CT == ct && WP == wp) return &conv1x1_pipe_f32<PT, ct, wp, SINGLE, NKK>;
    MI355_CASE4(1, 1) MI355_CASE4(2, 1) MI355_CASE4(4, 1)
    MI355_CASE4(1, 2) MI355_CASE4(2, 2) MI355_CASE4(4, 2)
    MI355_CASE4(1, 4) MI355_CASE4(2, 4) MI355_CASE4(4, 4)
#undef MI355_CASE4
    return nullptr;
}

KernelFn pick_f32_pipe(int CT, int WP, bool single, int ck, int PT) {
    if (PT == 1 || PT == 2) {            // small pixel tiles (latency-bound launches): 4 k-blocks per chunk only
        if (ck > 64) return nullptr;
        if (PT == 1) return single ? pick_pipe_s<1, true, 4>(CT, WP) : pick_pipe_s<1, false, 4>(CT, WP);
        return single ? pick_pipe_s<2, true, 4>(CT, WP) : pick_pipe_s<2, false, 4>(CT, WP);
    }
    if (ck > 64) {                       // 8 k-blocks per chunk: weights of a chunk = 8*CT fragments in registers
        if (CT > 2) return nullptr;
        return single ? pick_pipe_s<4, true, 8>(CT, WP) : pick_pipe_s<4, false, 8>(CT, WP);
    }
    return single ? pick_pipe_s<4, true, 4>(CT, WP) : pick_pipe_s<4, false, 4>(CT, WP);
}

}  // namespace mi355
