#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries kept under profiles/.

usage: tools/summarize_prof.py <round-tag> <kernel-trace-dir> [<pmc-dir> ...]      (MSC_PROFILES_DIR overrides profiles/)
Writes profiles/<tag>_kernel_stats.csv (name truncated to 90 chars, our msc:: kernels first) and
profiles/<tag>_pmc.json with per-kernel per-launch averages of every counter found, plus the HBM
traffic per launch derived as MI355X_MICROARCH.md prescribes for gfx950:
  read bytes  = 2 * FETCH_SIZE * 1024   (FETCH_SIZE reads exactly half of a coalesced stream)
  write bytes = WRITE_SIZE * 1024
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    if cut > 0:
        name = name[:cut]
    return name[:90]


def main():
    tag, ktdir = sys.argv[1], sys.argv[2]
    out = os.environ.get("MSC_PROFILES_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(ktdir, "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        rows.sort(key=lambda r: (not r["Name"].startswith(("msc::", "void msc::")), -float(r["TotalDurationNs"])))
        with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                            r["Percentage"], r["MinNs"], r["MaxNs"], "%.1f" % float(r["StdDev"])])
    # the bench's timed region = the LAST `MSC_TIMED_STEPS` launches of the headline kernel (nothing launches it after
    # the timed loop): its average is the one bench.py's HIP events must agree with
    traces = glob.glob(os.path.join(ktdir, "**", "*_kernel_trace.csv"), recursive=True)
    timed_steps = int(os.environ.get("MSC_TIMED_STEPS", "500"))
    # bench.py says which launches of its headline kernel were the timed region (roofline.timed_region_launches); its line
    # is in the log beside the trace (tools/profile_round.sh: kt.log)
    region, headline = None, None
    log = os.path.join(out, "kt.log")
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{"):
                try:
                    roof = json.loads(ln).get("roofline", {})
                    region, headline = roof.get("timed_region_launches"), roof.get("kernel")
                except ValueError:
                    pass
    if traces:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            if "msc::" in r["Kernel_Name"]:
                dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        with open(os.path.join(out, tag + "_timed_region.txt"), "w") as fh:
            fh.write("kernel-trace durations (ns), all launches vs the last %d (bench.py's timed region for the headline kernel)\n" % timed_steps)
            for k, v in sorted(dur.items(), key=lambda kv: -sum(d for _, d in kv[1])):
                v.sort()
                d = [x for _, x in v]
                last = d[-timed_steps:]
                fh.write("%-60s launches %5d avg %10.1f | last %4d: avg %10.1f min %9d max %9d\n"
                         % (k[:60], len(d), sum(d) / len(d), len(last), sum(last) / len(last), min(last), max(last)))
                if region and headline and headline in k and len(d) >= region[1]:
                    r = d[region[0]:region[1]]
                    fh.write("    bench.py's timed region = launches [%d, %d) of this kernel: avg %10.1f min %9d max %9d\n"
                             % (region[0], region[1], sum(r) / len(r), min(r), max(r)))
    # further kernel traces of the same bench under other allocation regimes (tools/profile_round.sh: MSC_EXTRA_TRACES =
    # "label=dir label=dir"): the headline kernel's instantiation differs with the regime (plain / non-temporal stores), and
    # the line's `frac` must be reproducible from this file whichever regime the driver's box offers
    for item in os.environ.get("MSC_EXTRA_TRACES", "").split():
        label, d = item.split("=", 1)
        tr = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
        if not tr:
            continue
        region2, headline2 = None, None
        lg = os.path.join(out, "kt_%s.log" % label)
        if os.path.exists(lg):
            for ln in open(lg):
                if ln.startswith("{"):
                    try:
                        roof = json.loads(ln).get("roofline", {})
                        region2, headline2 = roof.get("timed_region_launches"), roof.get("kernel")
                    except ValueError:
                        pass
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            if "k_score_nich1" in r["Kernel_Name"]:
                dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        with open(os.path.join(out, tag + "_timed_region.txt"), "a") as fh:
            fh.write("\n== the same bench command, regime \"%s\" (headline kernel: %s)\n" % (label, headline2))
            for k, v in sorted(dur.items(), key=lambda kv: -sum(d_ for _, d_ in kv[1])):
                v.sort()
                d_ = [x for _, x in v]
                fh.write("%-60s launches %5d avg %10.1f\n" % (k[:60], len(d_), sum(d_) / len(d_)))
                if region2 and headline2 and headline2 in k and len(d_) >= region2[1]:
                    r = d_[region2[0]:region2[1]]
                    fh.write("    bench.py's timed region = launches [%d, %d) of this kernel: avg %10.1f min %9d max %9d\n"
                             % (region2[0], region2[1], sum(r) / len(r), min(r), max(r)))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[3:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "msc::" in r["Kernel_Name"]:
                    pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary = {}
    for k, counters in pmc.items():
        e = {c: {"launches": len(v), "avg": sum(v) / len(v)} for c, v in counters.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            rd, wr = 2.0 * e["FETCH_SIZE"]["avg"] * 1024.0, e["WRITE_SIZE"]["avg"] * 1024.0
            e["hbm_bytes_per_launch"] = {"read_corrected": rd, "write": wr, "total": rd + wr}
        summary[k] = e
    if summary:
        with open(os.path.join(out, tag + "_pmc.json"), "w") as fh:
            json.dump(summary, fh, indent=1, sort_keys=True)
            fh.write("\n")
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
