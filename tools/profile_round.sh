#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the
# default bench command; tools/pmc_summary.py turns the CSVs into profiles/<round>/*.json.
#   usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1
    echo "pmc pass $i done: $PMC"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json" | head -60
