"""Per-layer roofline table from a rocprofv3 kernel trace of bench.py (test/bench infrastructure).

    python tools/layer_report.py gpurun_out/prof/.../*_kernel_trace.csv [model] [chunk]

Matches the dispatch sequence of one engine pass (stem, program ops..., decode, nms) against the op program and
prints, per conv launch, the median duration over all passes, achieved TFLOP/s and the layer's algorithmic bytes."""
import csv, sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvsd_amd.graph import engine_program, parse_model_name, OP_CONV, OP_STEM, OP_UPSAMPLE, OP_SPPF_POOL

path = sys.argv[1]
model = sys.argv[2] if len(sys.argv) > 2 else "yolov8n"
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 16
size = int(sys.argv[4]) if len(sys.argv) > 4 else 640
es = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0      # activation element size: 4 = fp32, 2 = the half=True engine
prog = engine_program(*parse_model_name(model))     # the program the engine runs (sibling convs merged)
sched_log = sys.argv[6] if len(sys.argv) > 6 else None      # bench log with the engine's "[sched] pos op stream launched" lines
rows = [r for r in csv.DictReader(open(path)) if "mi355" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))              # host enqueue order (kernels of different streams overlap in time)
kind = {OP_STEM: "stem", OP_CONV: "conv", OP_UPSAMPLE: "upsample2x", OP_SPPF_POOL: "sppf_pools"}
tail = ["decode_kernel"]          # ... followed by the NMS launches (nms_sort + nms_greedy, or the multi-launch sort of big maps), folded into one row
orders = [list(range(len(prog.ops)))]                       # program order (profiling passes) ...
launched = {i: True for i in range(len(prog.ops))}
if sched_log and os.path.exists(sched_log):
    sched = []
    for l in open(sched_log):
        if l.startswith("[sched] "):
            _, pos, op, stream, on = l.split()
            if int(pos) == 0: sched = []
            sched.append(int(op)); launched[int(op)] = on == "1"
    if sched: orders.append(sched)                          # ... and the multi-stream schedule (timed passes)
# The latency-bound regime runs a STEP schedule ("[step] k: singles i:name ... | group i:name ..." lines of the engine log): the ops
# of a step are independent, its grouped convs are ONE launch (conv_group_f32).  Reported by its own code path below.
steps = []
if sched_log and os.path.exists(sched_log):
    for l in open(sched_log):
        if l.startswith("[step] "):
            head, _, grp = l.partition(" | group")
            k = int(head.split()[1].rstrip(":"))
            if k == 0: steps = []
            singles = [int(t.split(":")[0]) for t in head.split("singles")[1].split()]
            group = [int(t.split(":")[0]) for t in grp.split()] if grp else []
            steps.append((singles, group))
if steps:
    entries = []                                  # (kernel-name fragment, [ops])
    for singles, group in steps:
        entries += [(kind[prog.ops[i].type], [i]) for i in singles]
        if group: entries.append(("conv_group", group))
    names = [e[0] for e in entries] + tail
    passes, i = [], 0
    while i < len(rows):
        if i + len(names) <= len(rows) and all(names[j] in rows[i + j]["Kernel_Name"] for j in range(len(names))):
            e = i + len(names)
            nms = []
            while e < len(rows) and "nms_" in rows[e]["Kernel_Name"]:
                nms.append(rows[e]); e += 1
            passes.append((rows[i:i + len(names)], sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in nms) / 1e3, len(nms)))
            i = e
        else:
            i += 1
    print(f"{len(passes)} passes of {len(names) + 1} launches matched ({model}, chunk {chunk}; step schedule, grouped launches = one row)")
    print(f"{'op(s)':64s} {'kernel':34s} {'grid':>12s} {'lds':>6s} {'vgpr':>5s} {'us':>8s} {'TFLOP/s':>8s}")
    tot = conv_us = 0.0; conv_launches = 0
    for j, (frag, ops_l) in enumerate(entries + [("decode_kernel", [])]):
        d = [(int(p[0][j]["End_Timestamp"]) - int(p[0][j]["Start_Timestamp"])) / 1e3 for p in passes]
        us = statistics.median(d); tot += us
        r = passes[0][0][j]
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", ""); kn = kn[:kn.find("(")] if "(" in kn else kn
        kn = kn.replace("void mi355::", "").replace("mi355::", "")
        fl = sum(2.0 * c.cout * c.cin * c.k * c.k * (size // c.stride_div) ** 2 * chunk for c in (prog.convs[prog.ops[o].conv] for o in ops_l if prog.ops[o].type in (OP_CONV, OP_STEM)))
        label = " | ".join(prog.convs[prog.ops[o].conv].name if prog.ops[o].type in (OP_CONV, OP_STEM) else kind[prog.ops[o].type] for o in ops_l) or frag
        if frag in ("conv", "conv_group"): conv_us += us; conv_launches += 1
        print(f"{label[:64]:64s} {kn[:34]:34s} {r['Grid_Size_X'] + 'x' + r['Grid_Size_Y']:>12s} {r['LDS_Block_Size']:>6s} {int(r['VGPR_Count'])+int(r['Accum_VGPR_Count']):>5d} {us:8.1f} " + (f"{fl / us / 1e6:8.2f}" if fl else ""))
    nms_us = statistics.median([p[1] for p in passes]); tot += nms_us
    print(f"{'nms (' + str(passes[0][2]) + ' launches)':64s} {'':34s} {'':>12s} {'':>6s} {'':>5s} {nms_us:8.1f}")
    conv_flops = sum(2.0 * c.cout * c.cin * c.k * c.k * (size // c.stride_div) ** 2 * chunk for c in prog.convs if c.cin != 3)
    print(f"sum of medians: {tot:.1f} us per pass of {chunk} frames -> {chunk / tot * 1e6:.0f} frames/s device-only; {len(names) + passes[0][2]} launches per pass")
    print(f"conv launches (single convs + grouped launches): {conv_launches} per pass, {conv_us:.1f} us -> {conv_flops / conv_us / 1e6:.2f} TFLOP/s")
    sys.exit(0)
# split the trace into passes: a pass is the kernel sequence of one of the orders (ops that are fused away are not launched)
cands = []
for o in orders:
    ops_l = [i for i in o if launched[i]]
    cands.append((ops_l, [kind[prog.ops[i].type] for i in ops_l] + tail))
passes, i = [], 0
while i < len(rows):
    hit = None
    for ops_l, names in cands:
        if i + len(names) <= len(rows) and all(names[j] in rows[i + j]["Kernel_Name"] for j in range(len(names))):
            hit = (ops_l, names); break
    if hit:
        # store the pass re-ordered into program order of the launched ops, so that column j means the same op in every pass
        ops_l, names = hit
        by_op = {op: rows[i + j] for j, op in enumerate(ops_l)}
        prog_l = [k for k in range(len(prog.ops)) if launched[k]]
        e = i + len(names)
        nms = []
        while e < len(rows) and "nms_" in rows[e]["Kernel_Name"]:
            nms.append(rows[e]); e += 1
        nms_row = dict(nms[0]) if nms else None
        if nms_row:                                          # one synthetic row: total NMS time of the pass
            nms_row["Kernel_Name"] = f"nms ({len(nms)} launches)"
            nms_row["Start_Timestamp"] = "0"
            nms_row["End_Timestamp"] = str(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in nms))
        passes.append([by_op[k] for k in prog_l] + rows[i + len(ops_l):i + len(names)] + ([nms_row] if nms_row else [])); i = e
    else:
        i += 1
prog_l = [k for k in range(len(prog.ops)) if launched[k]]
seq = [kind[prog.ops[k].type] for k in prog_l] + tail + ["nms"]
passes = [p for p in passes if len(p) == len(seq)]
print(f"{len(passes)} passes of {len(seq)} launches matched ({model}, chunk {chunk})")
tot = 0.0
print(f"{'op':40s} {'kernel<KS,S,PT,CT,WP>':24s} {'shape':30s} {'grid':>12s} {'lds':>6s} {'vgpr':>5s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}")
for j, name in enumerate(seq):
    d = [(int(p[j]["End_Timestamp"]) - int(p[j]["Start_Timestamp"])) / 1e3 for p in passes]
    us = statistics.median(d); tot += us
    r = passes[0][j]
    kn = r["Kernel_Name"]
    tmpl = kn[kn.find("<"):kn.find(">") + 1] if "<" in kn else ""
    if j < len(prog_l) and prog.ops[prog_l[j]].type in (OP_CONV, OP_STEM):
        op = prog.ops[prog_l[j]]; c = prog.convs[op.conv]
        hw = (size // c.stride_div) ** 2
        fl = 2.0 * c.cout * c.cin * c.k * c.k * hw * chunk
        by = es * chunk * (c.cin * hw * c.s * c.s + c.cout * hw)
        print(f"{c.name[:40]:40s} {tmpl:24s} {f'{c.cin}->{c.cout} k{c.k}s{c.s} @{size//c.stride_div}':30s} "
              f"{r['Grid_Size_X'] + 'x' + r['Grid_Size_Y']:>12s} {r['LDS_Block_Size']:>6s} {int(r['VGPR_Count'])+int(r['Accum_VGPR_Count']):>5d} {us:8.1f} {fl / us / 1e6:8.2f} {by / us / 1e3:7.0f}")
    else:
        print(f"{name:40s} {'':24s} {'':30s} {r['Grid_Size_X'] + 'x' + r['Grid_Size_Y']:>12s} {r['LDS_Block_Size']:>6s} {int(r['VGPR_Count'])+int(r['Accum_VGPR_Count']):>5d} {us:8.1f}")
print(f"sum of medians: {tot:.1f} us per pass of {chunk} frames -> {chunk / tot * 1e6:.0f} frames/s device-only")
# --stats style summary over the matched passes only (the raw rocprofv3 kernel_stats.csv also counts the engine's
# one-off autotune launches, which are not part of a step)
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for p in passes:
    for r in p:
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", ""); kn = kn[:kn.find("(")] if "(" in kn else kn
        agg[kn][0] += 1; agg[kn][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
is_conv = lambda k: "conv_igemm" in k or "conv1x1_" in k or "conv_splitk" in k or "conv3x3_lw" in k
conv_calls = sum(v[0] for k, v in agg.items() if is_conv(k)); conv_us = sum(v[1] for k, v in agg.items() if is_conv(k))
all_us = sum(v[1] for v in agg.values())
print(f"\nkernel summary over {len(passes)} real passes:")
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:60]:60s} {v[0]:6d} {v[1]:11.1f} {v[1] / v[0]:9.1f} {100 * v[1] / all_us:6.2f}")
conv_flops = sum(2.0 * c.cout * c.cin * c.k * c.k * (size // c.stride_div) ** 2 * chunk for c in prog.convs if c.cin != 3)
print(f"conv kernels (conv_igemm_* + conv3x3_lw_* + conv1x1_stream_* + conv1x1_pipe_*, all instances): {conv_calls} launches, avg {conv_us / conv_calls:.1f} us, {conv_us / len(passes):.1f} us per pass "
      f"-> {conv_flops / (conv_us / len(passes)) / 1e6:.2f} TFLOP/s")
