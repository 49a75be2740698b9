#!/bin/bash
# GPU box: scalar-side cache counters of one kernel on one config (instruction cache and scalar data cache of the SQC).
# usage: tools/icache_pass.sh <outdir> <config> <kernel id>
set -u
OUT=$(realpath -m ${1:-gpurun_out/icache}); CFG=${2:-city_4k}; K=${3:-8}; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_INST[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*" | sort -u > $OUT/avail.txt
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$TAG -- python3 $REPO/bench.py --pmc-child --config $CFG --kernel $K --prewarm-seconds 0 > $OUT/$TAG.out 2> $OUT/$TAG.err || echo "pass $TAG failed"
  F=$(find $OUT/$TAG -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "shadowMask" not in row["Kernel_Name"]: continue
    a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()): print(f"{k}: {v / n:.0f} per launch ({n} launches)")
PY
  rm -rf $OUT/$TAG
done
