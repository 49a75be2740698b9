#!/bin/bash
# Runs on the GPU box (via gpurun).  rocprofv3 kernel trace + separate PMC passes of bench.py.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-pmc --no-probes $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 200 --warmup 20 $ARGS > $OUT/trace.json 2> $OUT/trace.err
echo "trace rc=$?"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_CVT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ_READ_8 SQC_TC_DATA_READ_REQ" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $REPO/bench.py --steps 5 --warmup 2 --prewarm-seconds 0 $ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc$i ($PMC) rc=$?"
done
# calibration of FETCH_SIZE / WRITE_SIZE: same dispatch, 1-triangle BVH => reads = the 132.7 MB position stream
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/calib_$PMC -- python3 $REPO/bench.py --steps 5 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-pmc --config calib_4k > $OUT/calib_$PMC.json 2> $OUT/calib_$PMC.err
  echo "calib $PMC rc=$?"
done
cd $REPO
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
