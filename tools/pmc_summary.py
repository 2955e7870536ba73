#!/usr/bin/env python
"""print per-launch averages of every PMC counter found under the given rocprofv3 output dirs, per kernel INSTANTIATION
(full name up to the argument list) for kernels whose name contains the given substring.
usage: pmc_summary.py <substr> <dir>..."""
import collections
import csv
import glob
import re
import sys

sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("msc::", "")
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name in sorted(agg):
            print(d.split("/")[-1], name)
            for k, v in sorted(agg[name].items()):
                print("    %-28s %4d  %.5g" % (k, len(v), sum(v) / len(v)))
