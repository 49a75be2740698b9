#!/bin/bash
# GPU box: the same counters for several option sets of one config.  usage: tools/pmc_ab.sh <outdir> <config> "<counters>" "<opts A>" "<opts B>" ...
set -u
OUT=$(realpath -m $1); CFG=$2; PMC=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for OPTS in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/run$i -- python3 $REPO/tools/run_launches.py $CFG $OPTS > $OUT/run$i.log 2>&1
  echo "== $OPTS (rc=$?)"
  python3 - "$OUT/run$i" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "shadowMask" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])       # (per instantiation: a planning launch is another kernel)
            acc[key] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(f"   {k[0]} {k[1]}: {acc[k] / len(n[k]):.0f} per launch ({len(n[k])} launches)")
PY
done
