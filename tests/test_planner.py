"""Host-side unit tests of the conv launch planner (no GPU: mi355_plan_query only enumerates plans)."""
import pytest

from cvsd_amd import ops


def test_pointwise_conv_is_offered_every_kernel_family():
    v = ops.plan_versions(64, 80, 80, 128, 64, 1)
    assert {1, 3, 4} <= set(v)


def test_latency_bound_pointwise_conv_is_not_offered_the_pipelined_kernel():
    """A few frames per pass: the persistent pipelined kernel (v4) never won there and crowded the streaming kernel's variants
    out of the timed candidates, so it is only offered where the conv needs its fused upsample (conv_plan.hip)."""
    v = ops.plan_versions(1, 80, 80, 128, 64, 1)
    assert {1, 3} <= set(v) and 4 not in v


def test_3x3_conv_is_offered_staged_and_fused_plans():
    assert 1 in ops.plan_versions(8, 80, 80, 64, 64, 3)
    assert set(ops.plan_versions(8, 80, 80, 64, 64, 3, f2_cout=64)) == {101}


@pytest.mark.parametrize("pixels,cs", [
    (512 * 6400, 196),          # yolov8n-pose, chunk 512, head level 0: 642 M elements (2.57 GB) -- the advisor's example
    (512 * 25600, 64),          # yolov8s C2f pointwise conv at batch 512: 839 M elements
    ((1 << 29) // 64, 64),      # exactly 2^29 elements: the first size whose dropped-store marker 0x80000000 would be in range
    ((1 << 30) // 64 - 1, 64),  # just below 2^30 (the old guard's limit)
])
def test_no_descriptor_addressed_plan_for_slices_of_2_pow_29_elements_or_more(pixels, cs):
    """conv_igemm_f32 drops stores of out-of-tile lanes / pad channels by giving them byte offset 0x80000000 inside a buffer
    descriptor over one image of the slice; that only works while the image is at most 2^31 bytes.  A pointwise launch sees the
    flattened batch as one image, so for >= 2^29 elements no version-1 plan may be offered (v3 / v4 remain)."""
    for kw in (dict(dst_cs=cs), dict(src_cs=cs), dict(dst_cs=64, res_cs=cs)):
        v = ops.plan_versions(1, 1, pixels, 48, 51, 1, **kw)
        assert v and 1 not in v and 101 not in v, (kw, sorted(set(v)))
    # just below the limit the staged kernel is still on offer
    assert 1 in ops.plan_versions(1, 1, (1 << 29) // 64 - 64, 48, 51, 1, dst_cs=64)


def test_fused_pointwise_stage_checks_its_own_destination_stride():
    # a 3x3 image stays small, but the FUSED stage's destination image (dst2) may not: 2^29 elements -> no fused plan
    H = W = 2048
    cs2 = (1 << 29) // (H * W)              # 128
    assert ops.plan_versions(1, H, W, 16, 16, 3, f2_cout=32, f2_dst_cs=cs2 // 2)
    with pytest.raises(ValueError):
        ops.plan_versions(1, H, W, 16, 16, 3, f2_cout=32, f2_dst_cs=cs2)


def test_half_3x3_stride_1_convs_are_offered_the_lds_weights_kernels_and_nothing_else_is():
    """Round 4: launch-plan version 7 (csrc/conv_f16_lw.hip: block weights staged in LDS, persistent blocks, 16 x 16 output tile x 48 / 64 / 96
    couts) exists for half=True 3x3 / stride-1 convs without a fused pointwise stage; every other conv keeps its earlier candidate list."""
    v = ops.plan_versions(16, 80, 80, 192, 192, 3, half=True, src_cs=192, dst_cs=192)
    assert 7 in v and 1 in v
    assert v.count(7) == 3                                                                     # CT 6, 4 and 3 all tile 12 cout tiles
    assert 7 not in ops.plan_versions(16, 80, 80, 192, 192, 3)                                 # fp32
    assert 7 not in ops.plan_versions(16, 80, 80, 192, 192, 3, stride=2, half=True, src_cs=192, dst_cs=192)
    assert 7 not in ops.plan_versions(16, 80, 80, 192, 192, 1, half=True, src_cs=192, dst_cs=192)
    assert set(ops.plan_versions(16, 80, 80, 64, 64, 3, f2_cout=64, half=True, src_cs=64, dst_cs=64, f2_dst_cs=64)) == {101}
    assert 7 not in ops.plan_versions(16, 80, 80, 16, 16, 3, half=True, src_cs=16, dst_cs=16)  # one cout tile: the smallest block covers three
    assert 7 not in ops.plan_versions(16, 4, 4, 192, 192, 3, half=True, src_cs=192, dst_cs=192)    # maps smaller than half a tile


def test_half_pointwise_convs_are_offered_the_lds_weights_kernel():
    """Round 4: launch-plan version 10 (csrc/conv_f16_lw.hip: conv1x1_lwx_f16 -- shared weights through double-buffered LDS, each wave's pixels staged
    in full cache lines through a wave-private LDS image, persistent blocks of 256 pixels x 96 / 48 couts) for half=True pointwise convs; never for
    fp32, 3x3 convs or fewer than three cout tiles."""
    v = ops.plan_versions(16, 160, 160, 576, 192, 1, half=True, src_cs=576, dst_cs=192)
    assert v.count(10) == 2 and {1, 4} <= set(v)
    assert 10 not in ops.plan_versions(16, 160, 160, 576, 192, 1)
    assert 10 not in ops.plan_versions(16, 160, 160, 192, 192, 3, half=True, src_cs=192, dst_cs=192)
    assert 10 not in ops.plan_versions(16, 40, 40, 64, 32, 1, half=True, src_cs=64, dst_cs=32)
