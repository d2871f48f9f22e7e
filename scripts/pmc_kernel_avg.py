"""Per-kernel average of one PMC counter (e.g. the derived MfmaUtil) over the last `nlast` dispatches of a rocprofv3
database, time-weighted total at the end."""
import sqlite3
import sys

path, counter, nlast = sys.argv[1], sys.argv[2], int(sys.argv[3])
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
g = lambda s: [t for t in tabs if s in t][0]
kd, ks, pe, ip = g("kernel_dispatch"), g("kernel_symbol"), g("rocpd_pmc_event"), g("rocpd_info_pmc")
rows = c.execute(f"select d.id, d.event_id, s.kernel_name, d.end-d.start, d.grid_size_x/d.workgroup_size_x from {kd} d "
                 f"join {ks} s on d.kernel_id=s.id order by d.id").fetchall()[-nlast:]
agg = {}
tw = tt = 0.0
for _, ev, name, dur, blocks in rows:
    v = c.execute(f"select avg(e.value) from {pe} e join {ip} i on e.pmc_id=i.id where e.event_id={ev} and i.name='{counter}'").fetchone()[0]
    if v is None:
        continue
    key = (name.replace("_ZN12saber_mi355x", "").replace("NS_9ConvKArgsE", "")[:60], blocks)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += v; a[2] += dur
    tw += v * dur; tt += dur
print("%-62s %7s %4s %10s %9s" % ("kernel", "blocks", "n", counter, "avg_us"))
for (name, blocks), (n, sv, sd) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-62s %7d %4d %10.2f %9.2f" % (name, blocks, n, sv / n, sd / n / 1e3))
print("time-weighted %s over %d dispatches: %.2f" % (counter, len(rows), tw / max(tt, 1)))
