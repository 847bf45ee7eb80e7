#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + PMC passes of tools/bench_al.py (the AL Newton step)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_al
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/bench_al.py"
timeout -k 10 300 $CMD > "$OUT/bench_al.txt" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1
    echo "pmc pass $i done: $PMC"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
grep -v amdgpu.ids "$OUT/bench_al.txt"
