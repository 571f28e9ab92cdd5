#!/usr/bin/env python3
"""Per-kernel averages of the SQ counter passes written by scripts/gpu_pmc_sift.sh:  pmc_sift_summary.py <tag> <name> [kernel substring ...]"""
import collections
import csv
import glob
import re
import sys

tag, name = sys.argv[1], sys.argv[2]
want = sys.argv[3:] or ["k_"]
for ps in ("sq1", "sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("gpurun_out/%s_%s_%s/*counter_collection.csv" % (tag, name, ps)):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_\w+", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(0)
            if not any(w in k for w in want):
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in sorted(agg):
        print(name, ps, k, " ".join("%s=%.4g" % (c.replace("SQ_", ""), v / max(cnt[(k, c)], 1)) for c, v in sorted(agg[k].items())))
