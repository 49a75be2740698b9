set -u
OUT=$(realpath -m gpurun_out/r06/prologue); mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for K in 8 3; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/k$K -- python3 $REPO/bench.py --pmc-child --config city_4k --kernel $K --prewarm-seconds 0 > $OUT/k$K.log 2>&1
python3 - $OUT/k$K $K <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(dict)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "shadowMask" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
cal, real = ids[:6], ids[6:]
def avg(idl, c): return sum(rows[i][c] for i in idl) / len(idl)
for name, idl in (("1-triangle BVH (ray set-up, one node, store)", cal), ("city", real)):
    w = avg(idl, "SQ_WAVES")
    print(f"kernel {sys.argv[2]} {name}: {len(idl)} launches, per wave: VALU {avg(idl,'SQ_INSTS_VALU')/w:.1f} SALU {avg(idl,'SQ_INSTS_SALU')/w:.1f} SMEM {avg(idl,'SQ_INSTS_SMEM')/w:.1f}  ({w:.0f} waves)")
PY
done
