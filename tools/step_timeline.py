#!/usr/bin/env python3
"""One training step of a rocprofv3 kernel trace (rocpd db) as a timeline, in launch order: offset from the step's first
kernel, duration, idle gap in front of each kernel -- where the serial [B,d] tail sits between the chain kernels, and what
the boundaries cost (eager launches vs a replayed graph).  A step is cut at `pack_weights_kernel` (the forward pass's first
launch); the LAST complete step of the trace is printed, plus the sums: busy time, gap time, span.
    python tools/step_timeline.py gpurun_out/x/r_results.db [--step -1] [--brief]"""
import sqlite3
import sys


def load(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    rows = list(c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)))
    import re
    short = {}
    for n in set(r[0] for r in rows):
        m = re.search(r"macx\d+([a-z0-9_]+_kernel)(I[A-Za-z0-9_]*?E)?E", n)      # mangled: _ZN4macx<len><name>[I<template args>E]E...
        short[n] = (m.group(1) + (("<" + m.group(2)[1:-1] + ">") if m.group(2) else "")) if m else n[:64]
    return [(short[n], s, e) for n, s, e in rows]


def main():
    path = sys.argv[1]
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -1
    brief = "--brief" in sys.argv
    rows = load(path)
    cuts = [i for i, r in enumerate(rows) if "pack_weights_kernel" in r[0]]
    if len(cuts) < 2:
        print("fewer than two steps in the trace")
        return
    steps = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    a, b = steps[which - 1 if which < 0 else which]      # (the last cut has no end: -1 names the last COMPLETE step)
    seg = rows[a:b]
    t0 = seg[0][1]
    busy = gap = 0.0
    prev_end = None
    per = {}
    for n, s, e in seg:
        g = 0.0 if prev_end is None else (s - prev_end) / 1e3
        d = (e - s) / 1e3
        busy += d
        gap += max(g, 0.0)
        p = per.setdefault(n, [0, 0.0, 0.0])
        p[0] += 1; p[1] += d; p[2] += max(g, 0.0)
        if not brief:
            print("%9.1f us  +%6.2f gap  %8.2f us  %s" % ((s - t0) / 1e3, g, d, n))
        prev_end = max(e, prev_end or e)
    span = (seg[-1][2] - t0) / 1e3
    print("step: %d launches, span %.1f us, busy %.1f us, gaps %.1f us (overlap %.1f)" % (len(seg), span, busy, gap, busy + gap - span))
    print("%-64s %5s %10s %10s" % ("kernel", "n", "busy_us", "gap_before_us"))
    for n, p in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-64s %5d %10.1f %10.1f" % (n, p[0], p[1], p[2]))


if __name__ == "__main__":
    main()
