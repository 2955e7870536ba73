#!/usr/bin/env python
"""print per-launch averages of every PMC counter found under the given rocprofv3 output dirs,
for kernels whose name contains the given substring.  usage: pmc_summary.py <substr> <dir>..."""
import collections
import csv
import glob
import sys

sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(d.split("/")[-1], k, len(v), "%.5g" % (sum(v) / len(v)))
