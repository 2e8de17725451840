"""Summarise rocprofv3 --pmc counter CSVs per kernel (mean per launch).  usage: pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void aecf::", "").replace("(anonymous namespace)::", "")[:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names) + ",launches")
for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0], [0])))):
    n = max(len(v) for v in acc[k].values())
    print(k + "," + ",".join("%.3g" % (sum(acc[k][c]) / max(len(acc[k][c]), 1)) if c in acc[k] else "" for c in names) + ",%d" % n)
