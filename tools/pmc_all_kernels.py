"""Average of one rocprofv3 --pmc counter per kernel name (all kernels of the run, torch's included): python3 tools/pmc_all_kernels.py DIR LABEL"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/*.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, count(*), avg(p.value) from {pmc} p join {kd} d on p.event_id = d.event_id "
     f"join {ks} s on d.kernel_id = s.id group by s.kernel_name")
for name, n, val in c.execute(q):
    print(sys.argv[2], name[:90], n, round(val * 1024 / 1e6, 2), "MB")
