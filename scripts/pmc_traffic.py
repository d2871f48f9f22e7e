"""Sum a PMC counter (FETCH_SIZE / WRITE_SIZE, in KB) over the kernels of the LAST forward pass found in a
rocprofv3 database: the last `nops` dispatches whose kernel name matches the filter."""
import json
import sqlite3
import sys

path, counter, nlast = sys.argv[1], sys.argv[2], int(sys.argv[3])
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
g = lambda s: [t for t in tabs if s in t][0]
kd, ks, pe, ip = g("kernel_dispatch"), g("kernel_symbol"), g("rocpd_pmc_event"), g("rocpd_info_pmc")
rows = c.execute(f"select d.id, d.event_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.id").fetchall()
rows = rows[-nlast:]
tot_conv = tot_all = 0.0
n_conv = 0
for _, ev, name in rows:
    v = c.execute(f"select sum(e.value) from {pe} e join {ip} i on e.pmc_id=i.id where e.event_id={ev} and i.name='{counter}'").fetchone()[0] or 0.0
    tot_all += v
    if "conv_igemm" in name or "conv3x3_halo" in name or "conv_stem" in name:
        tot_conv += v
        n_conv += 1
print(json.dumps({"counter": counter, "dispatches": len(rows), "conv_dispatches": n_conv,
                  "sum_all_KB": tot_all, "sum_conv_KB": tot_conv}))
