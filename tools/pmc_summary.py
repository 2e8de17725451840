"""Summarise rocprofv3 --pmc counter passes per kernel (mean per launch), from the csv output or the rocpd sqlite
output (this image's default).  usage: pmc_summary.py <dir> [<dir> ...]"""
import collections, csv, glob, re, sqlite3, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void aecf::", "").replace("aecf::", "")
    return re.sub(r"\(.*", "", name)[:60]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        for k, c, v in sqlite3.connect(f).execute("select kernel_name, counter_name, value from counters_collection"):
            acc[short(k)][c].append(float(v))
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names) + ",launches")
for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0], [0])))):
    n = max(len(v) for v in acc[k].values())
    print(k + "," + ",".join("%.3g" % (sum(acc[k][c]) / max(len(acc[k][c]), 1)) if c in acc[k] else "" for c in names) + ",%d" % n)
