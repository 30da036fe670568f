"""Post-process a rocprofv3 kernel trace of tools/slow_step_loop.py: per forward pass (delimited by init_state_kernel), the span from
the first kernel's start to the last kernel's end, the sum of kernel durations, the largest gap and the longest kernel."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
rows = list(c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)))
# split into steps at chain_fwd step 0: use control_attend_kernel (once per forward) as the delimiter
steps, cur = [], []
for n, s, e in rows:
    if "control_attend_kernel" in n and cur:
        steps.append(cur)
        cur = []
    cur.append((n, s, e))
if cur:
    steps.append(cur)
for i, st in enumerate(steps):
    span = (st[-1][2] - st[0][1]) / 1e6
    busy = sum(e - s for _, s, e in st) / 1e6
    gaps = [(st[j + 1][1] - st[j][2], st[j][0], st[j + 1][0]) for j in range(len(st) - 1)]
    g = max(gaps) if gaps else (0, "", "")
    lk = max(st, key=lambda r: r[2] - r[1])
    print("step %2d: %3d kernels span %8.2f ms busy %7.2f ms  largest gap %8.2f ms (%s -> %s)  longest kernel %7.2f ms %s" % (
        i, len(st), span, busy, g[0] / 1e6, g[1][:28], g[2][:28], (lk[2] - lk[1]) / 1e6, lk[0][:40]))
