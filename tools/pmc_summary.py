#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile_round.sh: per kernel, the average of each
PMC counter over its dispatches (summed over the counter's dimensions) and the kernel-trace
duration statistics."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
short = lambda n: re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "")).replace("void ", "").strip()
res = defaultdict(dict)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(lambda: defaultdict(float))       # (kernel, counter) -> dispatch -> value
    for row in csv.DictReader(open(f)):
        per[(short(row["Kernel_Name"]), row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (k, c), d in per.items():
        res[k][c] = sum(d.values()) / len(d)
        res[k]["dispatches"] = len(d)
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        res[k]["trace_calls"] = int(row["Calls"])
        res[k]["trace_avg_us"] = float(row["AverageNs"]) / 1e3
        res[k]["trace_min_us"] = float(row["MinNs"]) / 1e3
        res[k]["trace_max_us"] = float(row["MaxNs"]) / 1e3
print(json.dumps({k: v for k, v in res.items() if "dqp" in k or "al_newton" in k or "al_chol" in k}, indent=1, sort_keys=True))
