#!/bin/bash
# GPU box: everything the round's evidence directory is made of, in one call.  usage: tools/final_evidence.sh <outdir>
set -u
OUT=$(realpath -m ${1:-gpurun_out/final}); mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gputests.log; tail -3 $OUT/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# the driver's command, then the long form
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_city4k_20_5.json 2> $OUT/bench_city4k_20_5.err; echo "bench 20/5 rc=$?"; cut -c1-260 $OUT/bench_city4k_20_5.json
timeout -k 10 400 python bench.py --save-counters $OUT/counters_city_4k.json > $OUT/bench_city4k.json 2> $OUT/bench_city4k.err; echo "bench rc=$?"
for CFG in courtyard_4k atrium_1080p cornell_256 city_4k_soft16 courtyard_4k_soft16; do
  timeout -k 10 400 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err; echo "bench $CFG rc=$?"; cut -c1-200 $OUT/bench_$CFG.json
done
# multi-rank flow on the one device (2 and 4 ranks share GPU 0): striped frame, default kernel
for N in 2 4; do
  RTS_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2955$N bench.py --gpus $N --steps 20 --warmup 5 > $OUT/bench_${N}rank_shared_device.json 2> $OUT/bench_${N}rank_shared_device.err; echo "bench $N ranks rc=$?"
done
# rocprofv3 kernel trace + stats of the same command (probes off: only the timed kernel runs)
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --no-pmc --no-probes --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace_bench.err); echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/city4k_kernel_stats.csv \;
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_courtyard -- python3 $REPO/bench.py --config courtyard_4k --no-pmc --no-probes --no-cpu-baseline > $OUT/trace_bench_courtyard.json 2> $OUT/trace_bench_courtyard.err); echo "trace courtyard rc=$?"
find $OUT/trace_courtyard -name "*kernel_stats.csv" -exec cp {} $OUT/courtyard4k_kernel_stats.csv \;
rm -rf $OUT/trace $OUT/trace_courtyard
timeout -k 10 300 python tools/floor_analysis.py > $OUT/floor_analysis_city4k.log 2>&1
timeout -k 10 300 python tools/dispatch_floor.py > $OUT/dispatch_floor.log 2>&1
timeout -k 10 400 python tests/experiments/soak.py 180 5000 > $OUT/soak_randomised_parity.log 2>&1; tail -1 $OUT/soak_randomised_parity.log
timeout -k 10 300 python tools/host_path_timing.py > $OUT/host_path_timing.log 2>&1; tail -2 $OUT/host_path_timing.log
