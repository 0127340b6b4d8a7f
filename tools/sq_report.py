"""Per-kernel SQ counters of one bench pass from the four rocprofv3 --pmc passes of tools/collect_sq.sh.
usage: sq_report.py [TAG=sq] [GHz=2.1]   (reads gpurun_out/TAG_{1..4}/*/*counter_collection.csv, last real pass of each run)"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "sq"
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.1
data = collections.defaultdict(lambda: collections.defaultdict(float))
dur, cnt = collections.defaultdict(float), collections.defaultdict(int)
for i in range(1, 5):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{i}", "*", "*counter_collection.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    start = max(int(r["Dispatch_Id"]) for r in rows if "::stem" in r["Kernel_Name"])      # last pass of the run
    seen = set()
    for r in rows:
        if int(r["Dispatch_Id"]) < start:
            continue
        m = re.search(r"(conv\w+<[^>]*>|stem\w+<[^>]*>)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        data[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if i == 3 and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[k] += 1
print(f"clock assumed {ghz} GHz; 1024 SIMDs.  VALU/MFMA = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA (SQ_INSTS_VALU counts the MFMAs too);")
print("implied cycles per vector instruction = 32 * (1 / mfma_busy - 1) / (VALU/MFMA): what one non-MFMA vector instruction costs if nothing overlaps")
print(f"{'kernel':42s} {'n':>2s} {'us':>6s} {'mfma busy':>9s} {'coexec':>7s} {'VALU/MFMA':>9s} {'cyc/VALU':>8s} {'LDS/MFMA':>8s} {'bank conflict':>13s} {'wait any':>8s}")
for k, us in sorted(dur.items(), key=lambda kv: -kv[1]):
    d = data[k]
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * ghz * 1e3 * 1024)
    vm = (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / max(d["SQ_INSTS_MFMA"], 1)
    cyc = 32 * (1 / busy - 1) / vm if busy > 0 and vm > 0 else 0
    print(f"{k:42s} {cnt[k]:2d} {us:6.0f} {100 * busy:8.1f}% {100 * d['SQ_VALU_MFMA_COEXEC_CYCLES'] / max(d['SQ_VALU_MFMA_BUSY_CYCLES'], 1):6.1f}% {vm:9.2f} {cyc:8.1f} "
          f"{d['SQ_INSTS_LDS'] / max(d['SQ_INSTS_MFMA'], 1):8.2f} {100 * d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1):12.1f}% {100 * d['SQ_WAIT_ANY'] / max(d['SQ_WAVE_CYCLES'], 1):7.1f}%")
