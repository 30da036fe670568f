#!/usr/bin/env python3
"""Per-kernel summary (calls, total, avg, min, max, %) from a rocprofv3 rocpd SQLite result
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db).  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/r_results.db > profiles/r01_kernel_stats.txt"""
import sqlite3
import subprocess
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = ("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s "
         "on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col))
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    names = [r[0] for r in rows]
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    except Exception:
        dem = names
    print("%-100s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r, n in zip(rows, dem):
        n = n.replace("macx::", "").replace("void ", "")
        n = n.split(" [clone")[0][:100]
        print("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n, r[1], r[2] / 1e3, r[2] / r[1] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[2] / total))
    print("%-100s %7d %12.1f" % ("TOTAL", sum(r[1] for r in rows), total / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
