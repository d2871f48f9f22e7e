"""Per-dispatch PMC table of the LAST forward pass in a rocprofv3 database: kernel, workgroups, duration and the summed
value of one counter (FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE printed raw AND doubled per the gfx950 correction of
MI355X_MICROARCH.md section HBM). Totals split into conv/fc kernels and the rest. JSON summary on the last line.

usage: python scripts/pmc_per_kernel.py <results.db> <counter> <nops>"""
import json
import sqlite3
import sys

# conv / fc launches = everything that is not one of the streaming kernels (an inclusion list missed every kernel added after it
# was written: the cooperative chains, the stage launch, the conv + global-pooling kernel - round-4 finding)
STREAMING = ("softmax_f32", "pool2d", "quantize", "transpose", "eltwise", "relu_f32", "gemm_pack", "null_kernel")
path, counter, nlast = sys.argv[1], sys.argv[2], int(sys.argv[3])
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
g = lambda s: [t for t in tabs if s in t][0]
kd, ks, pe, ip = g("kernel_dispatch"), g("kernel_symbol"), g("rocpd_pmc_event"), g("rocpd_info_pmc")
rows = c.execute(f"select d.id, d.event_id, s.kernel_name, d.end-d.start, d.grid_size_x*d.grid_size_y/d.workgroup_size_x "
                 f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()[-nlast:]
mult = 2.0 if counter == "FETCH_SIZE" else 1.0
tot_conv = tot_all = 0.0
n_conv = 0
print("%3s %9s %7s %12s %12s  %s" % ("#", "dur_us", "blocks", counter + "_KB", "corrected_KB", "kernel"))
per = []
for i, (_, ev, name, dur, blocks) in enumerate(rows):
    v = c.execute(f"select sum(e.value) from {pe} e join {ip} i on e.pmc_id=i.id where e.event_id={ev} and i.name='{counter}'").fetchone()[0] or 0.0
    short = name.replace("_ZN12saber_mi355x", "").replace("NS_9ConvKArgsE", "")[:64]
    print("%3d %9.2f %7d %12.1f %12.1f  %s" % (i, dur / 1e3, blocks, v, v * mult, short))
    tot_all += v
    per.append(v * mult * 1024)
    if not any(k in name for k in STREAMING):
        tot_conv += v
        n_conv += 1
print(json.dumps({"counter": counter, "dispatches": len(rows), "conv_dispatches": n_conv, "sum_all_KB": tot_all,
                  "sum_conv_KB": tot_conv, "per_dispatch_bytes_corrected": per}))
