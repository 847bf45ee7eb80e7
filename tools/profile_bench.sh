#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of one bench.py configuration;
# tools/pmc_summary.py condenses the CSVs (the summaries to keep are copied to profiles/r3/ by hand:
# pmc_config<C>.json + config<C>_kernel_stats.csv).
#   usage: tools/profile_bench.sh <config 1..5> [steps] [warmup]     (outputs under gpurun_out/prof_cfg<C>/)
# The program goes directly after `--` (python3 bench.py ...): no env / shell hop under the profiler.
set -e
CFG=${1:-1}
STEPS=${2:-10}
WARM=${3:-2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_cfg$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --config $CFG --no-cpu-baseline --steps $STEPS --warmup $WARM"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i + 1))
    timeout -k 10 400 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1
    echo "cfg $CFG pmc pass $i done: $PMC"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_config$CFG.json"
cp "$OUT"/stats/*/*kernel_stats.csv "$OUT/config${CFG}_kernel_stats.csv" 2>/dev/null || cp "$OUT"/stats/*kernel_stats.csv "$OUT/config${CFG}_kernel_stats.csv" 2>/dev/null || true
grep -h '"metric"' "$OUT/stats.log" > "$OUT/bench_under_rocprof.json" || true
# the raw per-dispatch CSVs are large: keep the condensed files only
rm -rf "$OUT"/stats/*/*kernel_trace.csv "$OUT"/pmc*/ 2>/dev/null || true
ls -la "$OUT"
