"""The liveness-based activation planner (csrc/engine.hip:plan_memory) checked from the outside, without a GPU: the engine's
placement is queried through the C ABI and verified against an INDEPENDENT restatement of the program's dependency DAG.

Safety property: two buffers may overlap in memory only if every op that touches the earlier one (reads or writes any channel
of it) is a read-after-write ANCESTOR of every op that writes the later one.  Then any schedule that honours RAW dependencies
-- the 4-stream schedule along the DAG, the step/group schedule, plain program order -- also honours every write-after-read
and write-after-write hazard the sharing creates: race-free without extra edges, and without losing any parallelism."""
import itertools

import numpy as np
import pytest

from cvsd_amd import ops
from cvsd_amd.graph import OP_CONV, OP_STEM, OP_SPPF_POOL, OP_UPSAMPLE, engine_program, parse_model_name
from cvsd_amd.weights import build_from_state_dict


def _blob(name):
    from tools import synth
    _, sd = synth.synthetic_checkpoint(name, seed=0)
    return build_from_state_dict(name, sd)


def _dag(prog, half=False):
    """RAW ancestors per op + users / writers per buffer, restated from the op program alone (graph.py semantics)."""
    n = len(prog.ops)
    written = lambda o: 3 * o.src.c if o.type == OP_SPPF_POOL else o.dst.c
    overlap = lambda a0, ac, b0, bc: a0 < b0 + bc and b0 < a0 + ac
    deps = [set() for _ in range(n)]
    users, writers = {}, {}
    for i, o in enumerate(prog.ops):
        reads = []
        if o.type != OP_STEM:
            reads.append((o.src.buf, o.src.choff, o.src.c))
        if o.res is not None:
            reads.append((o.res.buf, o.res.choff, o.dst.c))
        for buf, off, c in reads:
            users.setdefault(buf, set()).add(i)
            for j in range(i):
                w = prog.ops[j]
                if w.dst.buf == buf and overlap(w.dst.choff, written(w), off, c):
                    deps[i].add(j)
        users.setdefault(o.dst.buf, set()).add(i)
        writers.setdefault(o.dst.buf, set()).add(i)
    # a pointwise conv may read an upsample's SOURCE directly (upsample fused into its read side): it then uses that buffer
    for i, o in enumerate(prog.ops):
        if o.type != OP_UPSAMPLE:
            continue
        for j in range(i + 1, n):
            c = prog.ops[j]
            if c.type != OP_STEM and c.src.buf == o.dst.buf and overlap(c.src.choff, c.src.c, o.dst.choff, o.src.c):
                users[o.src.buf].add(j)
                deps[j] |= {k for k in range(i) if prog.ops[k].dst.buf == o.src.buf and overlap(prog.ops[k].dst.choff, written(prog.ops[k]), o.src.choff, o.src.c)}
    # Conv3x3 -> Conv1x1 pairs the engine may run as ONE launch (the 1x1 inside the 3x3's launch): the pair's output is then
    # written while the 3x3 still reads its input, so the 3x3 must count as a writer / user of the 1x1's output buffer
    head = {lv.buf for lv in prog.levels}
    for i, o in enumerate(prog.ops):
        if o.type != OP_CONV or prog.convs[o.conv].k != 3 or o.dst.buf in head:
            continue
        readers = [j for j, r in enumerate(prog.ops) if j != i and r.type != OP_STEM and
                   ((r.src.buf == o.dst.buf and overlap(r.src.choff, r.src.c, o.dst.choff, o.dst.c)) or
                    (r.res is not None and r.res.buf == o.dst.buf and overlap(r.res.choff, r.dst.c, o.dst.choff, o.dst.c)))]
        if len(readers) == 1 and readers[0] > i:
            r = prog.ops[readers[0]]
            exact = (r.src.buf, r.src.choff, r.src.c) == (o.dst.buf, o.dst.choff, o.dst.c)
            # ... or the tail of a C2f block: C2f.cv2 reads cat(ys), whose LAST slice is this Bottleneck conv's output
            # (fp32 engine only: the half=True kernels fuse exact-slice pairs without residual, nothing else)
            if half and (not exact or o.res is not None):
                continue
            tail = (r.src.buf == o.dst.buf and r.src.choff < o.dst.choff and r.src.choff + r.src.c == o.dst.choff + o.dst.c
                    and (o.dst.choff - r.src.choff) % 16 == 0 and o.dst.c % 16 == 0 and prog.convs[o.conv].s == 1)
            if r.type == OP_CONV and prog.convs[r.conv].k == 1 and r.res is None and (exact or tail):
                users[r.dst.buf].add(i)
                writers[r.dst.buf].add(i)
                if tail:                                  # the fused launch reads the earlier slices of the concat buffer
                    deps[i] |= {k for k in range(i) if prog.ops[k].dst.buf == r.src.buf and overlap(prog.ops[k].dst.choff, written(prog.ops[k]), r.src.choff, o.dst.choff - r.src.choff)}
    anc = [set() for _ in range(n)]
    for i in range(n):
        for d in deps[i]:
            anc[i] |= {d} | anc[d]
    return anc, users, writers


@pytest.mark.parametrize("name,half", [("yolov8n", False), ("yolov8n-pose", False), ("yolov8s-pose", False), ("yolov8m", True), ("yolov5mu", False)])
def test_overlapping_buffers_are_ordered_by_raw_dependencies(name, half):
    prog = engine_program(*parse_model_name(name))
    off, size, arena, plain = ops.memory_plan(_blob(name), 4, 640, 640, half=half)
    assert len(off) == len(prog.buffers) and arena == int((off + size).max()) and plain == int(size.sum())
    anc, users, writers = _dag(prog, half)
    head = {lv.buf for lv in prog.levels}
    shared_pairs = 0
    for a, b in itertools.combinations(range(len(off)), 2):
        if not (off[a] < off[b] + size[b] and off[b] < off[a] + size[a]):
            continue
        shared_pairs += 1
        assert a not in head and b not in head                      # the decode kernel reads head outputs after the last op
        for x in (a, b):                                            # buffers with pad channels keep bytes of their own (zeros)
            assert prog.buffers[x][0] % (8 if half else 4) == 0
        first, second = (a, b) if min(writers[a]) < min(writers[b]) else (b, a)
        for w in writers[second]:
            missing = users[first] - anc[w]
            assert not missing, f"{name}: buffer {second} (written by op {w}) overlaps buffer {first} still used by ops {sorted(missing)}"
    assert shared_pairs > 0
    # every buffer is 16-byte aligned (the kernels' vector accesses) and inside the arena
    assert (off % 256 == 0).all() and (off >= 0).all()


def test_footprint_shrinks_at_least_threefold_for_the_headline_workload():
    off, size, arena, plain = ops.memory_plan(_blob("yolov8n"), 512, 640, 640)
    print(f"yolov8n batch 512: {plain / 2**30:.1f} GiB unshared -> {arena / 2**30:.1f} GiB arena ({plain / arena:.2f}x)")
    assert plain >= 3.0 * arena
    _, _, arena0, plain0 = ops.memory_plan(_blob("yolov8n"), 512, 640, 640, reuse=False)
    assert arena0 == plain0 == plain                               # reuse off: one region per buffer


def test_placement_scales_with_the_batch_and_the_frame_size():
    b = _blob("yolov8n")
    o1, s1, a1, _ = ops.memory_plan(b, 1, 320, 320, imgsz=320)
    o8, s8, a8, _ = ops.memory_plan(b, 8, 320, 320, imgsz=320)
    assert (s8 >= 8 * s1 - 8 * 256).all() and a8 <= 8 * a1 + 256 * len(o1)
