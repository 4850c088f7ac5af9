"""Sum rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and pat not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    print(k[:100], "dispatches", len(cnt[k]))
    for n, v in sorted(c.items()): print(f"   {n:32s} {v / len(cnt[k]):.4g} per dispatch")
