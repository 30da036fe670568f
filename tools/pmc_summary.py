#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch).
    python tools/pmc_summary.py gpurun_out/pmcX/NAME_counter_collection.csv [more.csv ...]"""
import collections, csv, sys
def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
    names = []
    for path in paths:
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].replace('macx::', '').replace('void ', '')[:78]
            c = r['Counter_Name']
            if c not in names: names.append(c)
            agg[k][c] += float(r['Counter_Value'])
            key = (path, r['Dispatch_Id'])
            if key not in cnt[(k, c)]:
                cnt[(k, c)].add(key)
    print("%-80s " % "kernel (mean per dispatch)" + " ".join("%16s" % n[-16:] for n in names))
    order = sorted(agg, key=lambda k: -sum(agg[k].values()))
    for k in order[:18]:
        print("%-80s " % k + " ".join("%16.0f" % (agg[k][c] / max(1, len(cnt[(k, c)]))) for c in names))
if __name__ == "__main__":
    main(sys.argv[1:])
