"""Per-dispatch table of TWO counters of one rocprofv3 PMC pass over the LAST forward pass (e.g. the L2's memory-side read requests and
the subset of them destined for DRAM): kernel, duration, both sums, their ratio; totals over the conv / fc launches. JSON on the last line.

usage: python scripts/pmc_two_counters.py <results.db> <counter_a> <counter_b> <nops>"""
import json
import sqlite3
import sys

STREAMING = ("softmax_f32", "pool2d", "quantize", "transpose", "eltwise", "relu_f32", "gemm_pack", "null_kernel")
path, ca, cb, nlast = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
g = lambda s: [t for t in tabs if s in t][0]
kd, ks, pe, ip = g("kernel_dispatch"), g("kernel_symbol"), g("rocpd_pmc_event"), g("rocpd_info_pmc")
rows = c.execute(f"select d.id, d.event_id, s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()[-nlast:]
ta = tb = 0.0
print("%3s %9s %14s %14s %7s  %s" % ("#", "dur_us", ca, cb, "b/a", "kernel"))
for i, (_, ev, name, dur) in enumerate(rows):
    def val(cn):
        return c.execute(f"select sum(e.value) from {pe} e join {ip} i on e.pmc_id=i.id where e.event_id={ev} and i.name='{cn}'").fetchone()[0] or 0.0
    a, b = val(ca), val(cb)
    short = name.replace("_ZN12saber_mi355x", "").replace("NS_9ConvKArgsE", "")[:60]
    print("%3d %9.2f %14.0f %14.0f %7.3f  %s" % (i, dur / 1e3, a, b, b / a if a else 0.0, short))
    if not any(k in name for k in STREAMING):
        ta += a
        tb += b
print(json.dumps({"a": ca, "b": cb, "sum_conv_a": ta, "sum_conv_b": tb, "b_over_a": tb / ta if ta else None}))
