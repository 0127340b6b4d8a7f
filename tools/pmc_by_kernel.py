"""Per-kernel sums of the counters of ONE rocprofv3 --pmc pass (any command): pmc_by_kernel.py <dir> [regex of kernel names]
Prints, per kernel instance, dispatches, total microseconds and every counter's sum (plus bank-conflict share when both LDS counters are there)."""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"(conv\w+<[^>]*>|stem\w+<[^>]*>)")
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
data = collections.defaultdict(lambda: collections.defaultdict(float))
dur, seen = collections.defaultdict(float), collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    m = pat.search(r["Kernel_Name"])
    if not m:
        continue
    k = m.group(0)
    data[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[k]:
        seen[k].add(r["Dispatch_Id"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, us in sorted(dur.items(), key=lambda kv: -kv[1]):
    c = data[k]
    extra = ""
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        extra = f"  bank conflict {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f}% of LDS cycles"
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        extra += f"  wait_any {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.1f}%  wait_inst {100 * c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.1f}%"
    print(f"{k:46s} n={len(seen[k]):4d} {us:9.0f} us " + " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())) + extra)
