"""Summarise a rocprofv3 rocpd sqlite database: per (kernel, grid) count / avg / min / total duration."""
import sqlite3
import sys


def main(path, limit=40, tail=None):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    where = ""
    if tail:
        mx = c.execute(f"select max(id) from {kd}").fetchone()[0]
        where = f"where d.id > {mx - tail}"
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), sum(d.end-d.start), "
         f"d.grid_size_x*d.grid_size_y/ d.workgroup_size_x, s.arch_vgpr_count, d.group_segment_size from {kd} d join {ks} s on d.kernel_id=s.id {where} "
         f"group by s.kernel_name, d.grid_size_x, d.grid_size_y order by 5 desc limit {limit}")
    tot = c.execute(f"select sum(d.end-d.start) from {kd} d {where}").fetchone()[0]
    print("total kernel time %.1f us" % (tot / 1e3))
    print("%-58s %6s %9s %9s %10s %7s %5s %6s" % ("kernel", "n", "avg_us", "min_us", "total_us", "blocks", "vgpr", "lds"))
    for r in c.execute(q):
        name = r[0].replace("_ZN12saber_mi355x", "").replace("NS_9ConvKArgsE", "")[:58]
        print("%-58s %6d %9.2f %9.2f %10.1f %7d %5d %6d" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], r[6], r[7]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, int(sys.argv[3]) if len(sys.argv) > 3 else None)
