#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the stage-wise
# MPC QP bench (tools/bench_ric.py, config-4 shape); outputs under gpurun_out/prof_ric/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_ric
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/bench_ric.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc1" -o pmc -- $CMD > "$OUT/pmc1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc2" -o pmc -- $CMD > "$OUT/pmc2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(dict)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        per[(row["Kernel_Name"], row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (k, c), d in per.items():
        if "ric::" in k:
            res[k.split("(")[0]][c + "_KB_per_dispatch"] = sorted(d.values())
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "ric::" in row["Name"]:
            res[row["Name"].split("(")[0]].update(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3,
                                                 min_us=float(row["MinNs"]) / 1e3, max_us=float(row["MaxNs"]) / 1e3)
print(json.dumps(res, indent=1))
PY
