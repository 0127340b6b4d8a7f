"""Timeline of the reference's frame loop from a rocprofv3 kernel trace of tools/track_stages.py (test / bench infrastructure): per frame, the span of
the detector pass (stem .. NMS on the engine's queue), the motion-compensation kernels (gray_resize, min_eig, corner_mask, pyr_down, lk) that start
inside that frame's period, and how much of their time lies INSIDE the detector's span (overlap) -- i.e. whether the two share the GPU or queue.
usage: track_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, statistics as st, sys
kt = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
GMC = ("gray_resize", "min_eig", "corner_mask", "pyr_down", "lk_kernel")
is_gmc = lambda n: any(g in n for g in GMC)
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
stems = [i for i, k in enumerate(K) if "::stem" in k[2]]
out = []
for a, b in zip(stems[20:-3], stems[21:-2]):
    q = K[a][3]
    det = [k for k in K[a:b] if k[3] == q and not is_gmc(k[2])]
    d0, d1, period = det[0][0], max(k[1] for k in det), K[b][0] - det[0][0]
    g = [k for k in K if is_gmc(k[2]) and d0 - period // 2 <= k[0] < d0 + period // 2 + period // 4]
    g = [k for k in g if k[0] >= d0 - 400_000 and k[0] < K[b][0] - 5_000]          # this frame's step: enqueued right before the pass
    if not g:
        continue
    gsum = sum(k[1] - k[0] for k in g)
    inside = sum(max(0, min(k[1], d1) - max(k[0], d0)) for k in g)
    out.append((d1 - d0, sum(k[1] - k[0] for k in det), len(det), gsum, inside, min(k[0] for k in g) - d0, max(k[1] for k in g) - d0, period))
names = ["detector span", "detector kernel sum", "detector launches", "motion-comp kernel sum", "... of it inside the detector span", "motion-comp start - detector start",
         "motion-comp end - detector start", "frame period"]
print(f"{len(out)} frames; medians, microseconds")
for i, n in enumerate(names):
    print(f"  {n:38s} {st.median(o[i] for o in out) / (1 if 'launches' in n else 1e3):9.1f}")
