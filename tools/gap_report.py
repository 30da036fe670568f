"""Gap report of a rocprofv3 kernel trace (rocpd db): GPU busy fraction per 0.25 s window, and the largest idle gaps between
consecutive kernels with their neighbours -- where does a slow leg of bench.py lose its time, kernels or gaps?"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
rows = list(c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)))
t0 = rows[0][1]
W = 0.25e9
win = {}
for n, s, e in rows:
    w = int((s - t0) // W)
    b = win.setdefault(w, [0, 0, 0.0])
    b[0] += 1
    b[1] += e - s
    if "chain_fwd" in n:
        b[2] = max(b[2], (e - s) / 1e3)
print("window  kernels  busy%   longest chain_fwd (us)")
for w in sorted(win):
    print("%6.2fs %7d  %5.1f   %7.1f" % (w * 0.25, win[w][0], 100.0 * win[w][1] / W, win[w][2]))
gaps = sorted(((rows[j + 1][1] - rows[j][2], j) for j in range(len(rows) - 1)), reverse=True)[:25]
print("largest gaps:")
for g, j in gaps:
    print("  at %7.3f s  gap %8.3f ms   %s -> %s" % ((rows[j][2] - t0) / 1e9, g / 1e6, rows[j][0][:36], rows[j + 1][0][:36]))
