"""Print per-kernel PMC counter sums from a rocprofv3 rocpd database (last N dispatches of a kernel filter)."""
import sqlite3
import sys

path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "conv_igemm")
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
g = lambda s: [t for t in tabs if s in t][0]
kd, ks, pe, ip = g("kernel_dispatch"), g("kernel_symbol"), g("rocpd_pmc_event"), g("rocpd_info_pmc")
rows = c.execute(f"select d.id, d.event_id, s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id "
                 f"where s.kernel_name like '%{filt}%' order by d.id").fetchall()
if not rows:
    print("no dispatch matches", filt)
    sys.exit(0)
last = rows[-1]
print("kernel:", last[2][:90], "duration %.2f us" % (last[3] / 1e3), "(n=%d)" % len(rows))
cols = [r[1] for r in c.execute(f"pragma table_info({pe})")]
q = f"select i.name, sum(e.value) from {pe} e join {ip} i on e.pmc_id=i.id where e.event_id={last[1]} group by i.name"
for name, v in c.execute(q):
    print("  %-36s %16.0f" % (name, v))
