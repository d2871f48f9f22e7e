"""Per-dispatch view of ONE forward pass out of a rocprofv3 rocpd sqlite database (kernel trace): the last `nops`
dispatches in launch order with their device duration, the idle gap since the previous kernel ended, grid size, VGPRs
(arch + accumulation), LDS bytes and how many times the grid fills the chip at that occupancy ("fill": blocks / (256 CUs x
workgroups per CU); just above 1.0, 2.0 ... means a short tail wave) — the table that shows where a latency-bound op
list loses its time (kernel body vs. boundary).

usage: python scripts/trace_sequence.py <results.db> <nops> [forwards_to_average]
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as KR  # noqa: E402


def main(path, nops, nfwd=1):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    try:
        res = KR.load()            # total (arch + accumulation) registers per kernel, from the library's code objects
    except Exception:              # noqa: BLE001 - the table is still useful without the occupancy column
        res = {}
    rows = c.execute(f"select d.start, d.end, s.kernel_name, d.grid_size_x*d.grid_size_y/d.workgroup_size_x, "
                     f"d.workgroup_size_x, s.arch_vgpr_count, d.group_segment_size from {kd} d join {ks} s "
                     f"on d.kernel_id=s.id order by d.start").fetchall()
    rows = rows[-nops * nfwd:]
    dur = [0.0] * nops
    gap = [0.0] * nops
    gaps = [[] for _ in range(nops)]       # per dispatch: the gap in every forward (a mean hides WHERE it comes from: one host stall
    for f in range(nfwd):                  # of 60 us in one of 20 forwards reads as "3 us in front of this kernel")
        seg = rows[f * nops:(f + 1) * nops]
        for i, r in enumerate(seg):
            dur[i] += (r[1] - r[0]) / 1e3 / nfwd
            if i:
                gap[i] += (r[0] - seg[i - 1][1]) / 1e3 / nfwd
                gaps[i].append((r[0] - seg[i - 1][1]) / 1e3)
    seg = rows[-nops:]
    span = (seg[-1][1] - seg[0][0]) / 1e3
    print("one forward: %d dispatches, span %.1f us, sum of kernel durations %.1f us, sum of gaps %.1f us (avg over %d)"
          % (nops, span, sum(dur), sum(gap), nfwd))
    print("%3s %8s %7s %7s %6s %5s %7s %6s %5s  %s" % ("#", "dur_us", "gap_us", "blocks", "wgsz", "vgpr", "lds", "wg/CU", "fill", "kernel"))
    if nfwd > 1:
        big = [(i, sorted(g)) for i, g in enumerate(gaps) if g and sum(g) / len(g) > 0.3]
        for i, g in big:
            print("    gap in front of #%d over %d forwards: median %.2f us, max %.2f us, forwards above 1 us: %d"
                  % (i, len(g), g[len(g) // 2], g[-1], sum(1 for v in g if v > 1.0)))
    for i, r in enumerate(seg):
        name = r[2].replace("_ZN12saber_mi355x", "").replace("NS_9ConvKArgsE", "")[:70]
        kr = res.get(r[2].replace(".kd", ""), dict(vgpr=r[5], lds=r[6], wg=r[4]))
        per_cu = KR.workgroups_per_cu(kr, wgsz=r[4], lds=r[6])
        fill = r[3] / (256.0 * max(per_cu, 1))
        print("%3d %8.2f %7.2f %7d %6d %5d %7d %6d %5.2f  %s" % (i, dur[i], gap[i], r[3], r[4], kr["vgpr"], r[6], per_cu, fill, name))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)
