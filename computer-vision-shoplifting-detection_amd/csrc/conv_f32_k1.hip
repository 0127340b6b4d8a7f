// Instances of the pointwise (1x1) fp32 kernels that are not software-pipelined: conv_igemm_f32<1, 1, ...> (LDS-staged,
// the fallback for every shape) and conv1x1_stream_f32 (register-streamed, no LDS).  Device code: conv_f32.h.
#include "conv_f32.h"
#include "conv_f32_inst.h"

namespace mi355 {

KernelFn pick_f32_k1(int CT, int WP, int PT) {
    if (PT == 0) {
#define MI355_CASE(ct, wp) if (CT == ct && WP == wp) return &conv_igemm_f32<1, 1, (ct == 5 ? 3 : 4), ct, wp>;
        MI355_CASE(1, 4) MI355_CASE(2, 4) MI355_CASE(3, 4) MI355_CASE(4, 4) MI355_CASE(5, 4)
        MI355_CASE(1, 2) MI355_CASE(2, 2) MI355_CASE(3, 2) MI355_CASE(4, 2) MI355_CASE(5, 2)
        MI355_CASE(1, 1) MI355_CASE(2, 1) MI355_CASE(3, 1) MI355_CASE(4, 1) MI355_CASE(5, 1)
#undef MI355_CASE
        return nullptr;
    }
    // small wave tiles (1 or 2 pixel tiles per wave) for latency-bound launches: a batch-1 map has few pixels, so the
    // planner trades register blocking for more waves
#define MI355_CASE(pt, ct, wp) if (PT == pt && CT == ct && WP == wp) return &conv_igemm_f32<1, 1, pt, ct, wp>;
    MI355_CASE(1, 1, 4) MI355_CASE(1, 2, 4) MI355_CASE(1, 1, 2) MI355_CASE(1, 2, 2) MI355_CASE(1, 1, 1) MI355_CASE(1, 2, 1)
    MI355_CASE(2, 1, 4) MI355_CASE(2, 2, 4) MI355_CASE(2, 1, 2) MI355_CASE(2, 2, 2) MI355_CASE(2, 1, 1) MI355_CASE(2, 2, 1)
#undef MI355_CASE
    return nullptr;
}

KernelFn pick_f32_stream(int CT, int PT) {
    if (CT == 1 && PT == 1) return &conv1x1_stream_f32<1, 1>;
    if (CT == 2 && PT == 1) return &conv1x1_stream_f32<1, 2>;
    if (CT == 4 && PT == 1) return &conv1x1_stream_f32<1, 4>;
    if (CT == 1 && PT == 2) return &conv1x1_stream_f32<2, 1>;
    if (CT == 1 && PT == 4) return &conv1x1_stream_f32<4, 1>;
    if (CT == 2 && PT == 2) return &conv1x1_stream_f32<2, 2>;
    if (CT == 2 && PT == 4) return &conv1x1_stream_f32<4, 2>;
    if (CT == 4 && PT == 2) return &conv1x1_stream_f32<2, 4>;
    if (CT == 4 && PT == 4) return &conv1x1_stream_f32<4, 4>;
    return nullptr;
}
// the same with the nearest-2x upsample fused into the read side (latency-bound launches: the only other kernel with the fused read is the pipelined one)
KernelFn pick_f32_stream_up(int CT, int PT) {
    if (CT == 1 && PT == 1) return &conv1x1_stream_up_f32<1, 1>;
    if (CT == 2 && PT == 1) return &conv1x1_stream_up_f32<1, 2>;
    if (CT == 4 && PT == 1) return &conv1x1_stream_up_f32<1, 4>;
    if (CT == 1 && PT == 2) return &conv1x1_stream_up_f32<2, 1>;
    if (CT == 1 && PT == 4) return &conv1x1_stream_up_f32<4, 1>;
    if (CT == 2 && PT == 2) return &conv1x1_stream_up_f32<2, 2>;
    if (CT == 2 && PT == 4) return &conv1x1_stream_up_f32<4, 2>;
    if (CT == 4 && PT == 2) return &conv1x1_stream_up_f32<2, 4>;
    if (CT == 4 && PT == 4) return &conv1x1_stream_up_f32<4, 4>;
    return nullptr;
}

}  // namespace mi355
