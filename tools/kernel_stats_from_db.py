"""Per-kernel summary (the columns of rocprofv3 --stats' kernel_stats.csv) from the rocpd sqlite file that
`rocprofv3 --kernel-trace --stats -d <dir> -o <name>` leaves behind.   usage: kernel_stats_from_db.py <results.db> <out.csv>"""
import collections, csv, math, sqlite3, sys


def main():
    db, out = sys.argv[1:3]
    per = collections.defaultdict(list)
    for name, dur in sqlite3.connect(db).execute("select name, (end - start) from kernels"):
        per[name].append(dur)
    total = sum(sum(v) for v in per.values())
    rows = []
    for name, v in per.items():
        mean = sum(v) / len(v)
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / len(v))
        rows.append((name, len(v), sum(v), round(mean, 6), round(100.0 * sum(v) / total, 2), min(v), max(v), round(sd, 6)))
    rows.sort(key=lambda r: -r[2])
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerows(rows)
    for r in rows[:20]:
        print(f"{r[1]:5d} {r[3] / 1000:9.1f} us {r[4]:6.2f} %  {r[0][:100]}")


if __name__ == "__main__":
    main()
