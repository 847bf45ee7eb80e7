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
        vals = list(d.values())
        # the forward kernel is dispatched twice per solve in the batch-termination mode (pass 1: the
        # solve; pass 2: re-solve of the flagged problems, an empty launch in a large batch): average
        # the full dispatches only (counter value above half of the largest)
        # (only there: the Newton kernel of the short-horizon models writes its factor from one launch in eight)
        big = [v for v in vals if v > 0.5 * max(vals)] if (max(vals) > 0 and "forward_kernel" in k) else vals
        res[k][c] = sum(big) / len(big)
        res[k]["dispatches"] = len(vals)
        res[k]["dispatches_full"] = len(big)
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        res[k]["trace_calls"] = int(row["Calls"])
        res[k]["trace_avg_us"] = float(row["AverageNs"]) / 1e3
        res[k]["trace_min_us"] = float(row["MinNs"]) / 1e3
        res[k]["trace_max_us"] = float(row["MaxNs"]) / 1e3
# pass-1 / pass-2 split of the forward kernel from the raw trace
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        if "forward_kernel" in k and len(v) > 1 and min(v) < 0.2 * max(v):
            full = [x for x in v if x > 0.5 * max(v)]
            rest = [x for x in v if x <= 0.5 * max(v)]
            res[k]["trace_pass1_calls"], res[k]["trace_pass1_avg_us"] = len(full), sum(full) / len(full)
            res[k]["trace_pass2_calls"], res[k]["trace_pass2_avg_us"] = len(rest), sum(rest) / max(len(rest), 1)
# the bench command launches the forward kernel in two modes (batch rule: every problem 20 iterations, with
# history and snapshots; per-problem: the fast mode's own launches): split the trace by duration so that
# each mode's average can be compared with the HIP-event figure of the bench line
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        if "forward_kernel" in k and len(v) > 3 and max(v) > 1.1 * min(v):
            mid = 0.5 * (min(v) + max(v))
            slow, fast = [x for x in v if x >= mid], [x for x in v if x < mid]
            res[k]["trace_batch_mode_calls"], res[k]["trace_batch_mode_avg_us"] = len(slow), sum(slow) / len(slow)
            res[k]["trace_per_problem_calls"], res[k]["trace_per_problem_avg_us"] = len(fast), sum(fast) / max(len(fast), 1)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res["_library_fingerprint"] = bench.library_fingerprint()
# the library's own kernels only (torch / rocBLAS kernels of the policy network are not this package's)
skip = ("at::", "Cijk_", "rocblas", "hipcub", "rocprim", "__amd", "elementwise", "vectorized", "reduce_kernel", "void at")
print(json.dumps({k: v for k, v in res.items() if k == "_library_fingerprint" or not any(t in k for t in skip)}, indent=1, sort_keys=True))
