// Lookup of the compiled fp32 conv kernel instances (device code: conv_f32.h).  One translation unit per kernel family so
// that the library builds in parallel; the planner (conv_plan.hip) only sees these functions.
#pragma once
#include "common.h"

namespace mi355 {

typedef void (*KernelFn)(ConvKArgs);

// conv_igemm_f32<KS, STRIDE, PT, CT, WP>: PT 0 = the default wave tile (4 pixel tiles, 3 with CT 5), CT in 1..5;
// PT 1 / 2 = small wave tiles for latency-bound launches, CT in {1, 2}; WP in {1, 2, 4}
KernelFn pick_f32_k3s1(int CT, int WP, int PT);       // conv_f32_k3s1.hip
KernelFn pick_f32_k3s2(int CT, int WP, int PT);       // conv_f32_k3s2.hip
KernelFn pick_f32_k1(int CT, int WP, int PT);         // conv_f32_k1.hip
// conv1x1_stream_f32<PT, CT>: CT in {1, 2, 4}, PT in {1, 2, 4}
KernelFn pick_f32_stream(int CT, int PT);     // conv_f32_k1.hip
KernelFn pick_f32_stream_up(int CT, int PT);  // conv_f32_k1.hip: upsample fused into the read side
// conv1x1_pipe_f32<PT, CT, WP, SINGLE, NKK>: PT 0 / 4 = 4 pixel tiles per wave (NKK = 8 when ck > 64, then CT <= 2), PT 1 / 2 =
// small pixel tiles for latency-bound launches (ck <= 64); CT in {1, 2, 4}, WP in {1, 2, 4}
KernelFn pick_f32_pipe(int CT, int WP, bool single, int ck, int PT);   // conv_f32_pipe.hip

// conv_igemm_f32<3, STRIDE, PT, CT, WP, F2 = true> (a pointwise conv fused behind the 3x3): same (PT, CT, WP) space as above
KernelFn pick_f32_fused_s1(int CT, int WP, int PT);     // conv_f32_fused_s1.hip
KernelFn pick_f32_fused_s2(int CT, int WP, int PT);     // conv_f32_fused_s2.hip
// conv_splitk_f32<3, STRIDE, PT, CT>: PT, CT in {1, 2}
KernelFn pick_f32_splitk(int stride, int CT, int PT);   // conv_f32_splitk.hip

}  // namespace mi355
