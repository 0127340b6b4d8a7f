// Instances of conv_splitk_f32, the latency-oriented 3x3 kernel whose four waves split K (device code: conv_f32.h).
#include "conv_f32.h"
#include "conv_f32_inst.h"

namespace mi355 {

KernelFn pick_f32_splitk(int stride, int CT, int PT) {
#define MI355_CASE(st, pt, ct) if (stride == st && PT == pt && CT == ct) return &conv_splitk_f32<3, st, pt, ct>;
    MI355_CASE(1, 1, 1) MI355_CASE(1, 1, 2) MI355_CASE(1, 2, 1) MI355_CASE(1, 2, 2)
    MI355_CASE(2, 1, 1) MI355_CASE(2, 1, 2) MI355_CASE(2, 2, 1) MI355_CASE(2, 2, 2)
#undef MI355_CASE
    return nullptr;
}

}  // namespace mi355
