#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of a few launches per option set.  usage: tools/trace_ab.sh <outdir> <config> "<opts A>" "<opts B>" ...
set -u
OUT=$(realpath -m $1); CFG=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for OPTS in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run$i -- python3 $REPO/tools/run_launches.py $CFG launches=300 $OPTS > $OUT/run$i.log 2>&1
  echo "== $CFG $OPTS (rc=$?)"
  python3 - $OUT/run$i <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        print(f"   {r['Name'][:90]}: {r['Calls']} calls, avg {float(r['AverageNs']) / 1e3:.1f} us, min {float(r['MinNs']) / 1e3:.1f}")
PY
done
