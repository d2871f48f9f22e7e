"""Per-kernel instruction mix of ONE forward pass from a rocprofv3 PMC database: sums of SQ_INSTS_VALU / SQ_INSTS_MFMA /
SQ_INSTS_LDS / SQ_INSTS_VMEM_RD / SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (whichever were collected) per dispatch
of the last `nlast` dispatches, in launch order.

usage: python scripts/pmc_mix.py <results.db> <nlast>"""
import sqlite3
import sys

path, nlast = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
g = lambda s: [t for t in tabs if s in t][0]
kd, ks, pe, ip = g("kernel_dispatch"), g("kernel_symbol"), g("rocpd_pmc_event"), g("rocpd_info_pmc")
names = [r[0] for r in c.execute(f"select distinct i.name from {pe} e join {ip} i on e.pmc_id=i.id")]
rows = c.execute(f"select d.event_id, s.kernel_name, d.end-d.start, d.grid_size_x*d.grid_size_y/d.workgroup_size_x from {kd} d "
                 f"join {ks} s on d.kernel_id=s.id order by d.start").fetchall()[-nlast:]
print("%3s %8s %6s " % ("#", "dur_us", "blocks") + " ".join("%14s" % n[-14:] for n in names) + "  kernel")
for i, (ev, name, dur, blocks) in enumerate(rows):
    vals = []
    for n in names:
        v = c.execute(f"select sum(e.value) from {pe} e join {ip} i on e.pmc_id=i.id where e.event_id={ev} and i.name='{n}'").fetchone()[0]
        vals.append(v or 0.0)
    short = name.replace("_ZN12saber_mi355x", "").replace("NS_9ConvKArgsE", "")[:60]
    print("%3d %8.2f %6d " % (i, dur / 1e3, blocks) + " ".join("%14.0f" % v for v in vals) + "  " + short)
