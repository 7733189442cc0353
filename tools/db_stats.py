"""Per-kernel totals of a rocprofv3 --kernel-trace result database (rocpd sqlite).  python tools/db_stats.py DB [top]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(c.execute("select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), avg(d.grid_size_x) from %s d join %s s "
                      "on d.kernel_id=s.id group by s.kernel_name order by 4 desc" % (kd, sym)))
tot = sum(r[3] for r in rows)
print("total kernel ms %.3f" % (tot / 1e6))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r[0])
    print("  %-96s n=%6d avg %8.2f us  %5.1f%% grid %d" % (name[:96], r[1], r[2] / 1e3, 100 * r[3] / tot, r[4]))
