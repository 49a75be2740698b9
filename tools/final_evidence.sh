#!/bin/bash
# GPU box: everything the round's evidence directory is made of, in one call.  usage: tools/final_evidence.sh <outdir>
set -u
OUT=$(realpath -m ${1:-gpurun_out/final}); mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gputests.log; tail -3 $OUT/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# the driver's command (headline + the secondary workloads in one line), then the long form with the counters saved
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_city4k_20_5.json 2> $OUT/bench_city4k_20_5.err; echo "bench 20/5 rc=$?"; cut -c1-260 $OUT/bench_city4k_20_5.json
timeout -k 10 500 python bench.py --no-secondary --save-counters $OUT/counters_city_4k.json > $OUT/bench_city4k.json 2> $OUT/bench_city4k.err; echo "bench rc=$?"
for CFG in courtyard_4k atrium_1080p cornell_256 city_4k_soft16 courtyard_4k_soft16 city_4k_soft16_pp courtyard_4k_soft16_pp city_4k_directional; do
  timeout -k 10 500 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err; echo "bench $CFG rc=$?"; cut -c1-200 $OUT/bench_$CFG.json
done
# the two packet families side by side on the three scenes (same protocol, kernel forced)
for CFG in city_4k courtyard_4k atrium_1080p; do for K in 3 8; do
  timeout -k 10 300 python bench.py --config $CFG --kernel $K --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$CFG', d['config']['kernel'], d['value'], 'Mrays/s', d['config']['ms_per_frame_gpu_median'], 'ms', d['roofline']['shader_clock_mhz'], 'MHz')" >> $OUT/packet_vs_wide_ab.log
done; done; cat $OUT/packet_vs_wide_ab.log
# multi-rank flow on the one device (2 and 4 ranks share GPU 0): striped frame
for N in 2 4; do
  RTS_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2955$N bench.py --gpus $N --steps 20 --warmup 5 > $OUT/bench_${N}rank_shared_device.json 2> $OUT/bench_${N}rank_shared_device.err; echo "bench $N ranks rc=$?"
done
# rocprofv3 kernel trace + stats of the same command (probes off: only the timed kernel runs)
KID=$(python -c "import json; print({'shadowMaskPacketKernel<1,wide>': 8}.get(json.load(open('$OUT/bench_city4k.json'))['config']['kernel'], 3))")
KOPT=$(python -c "import json; print(json.load(open('$OUT/bench_city4k.json'))['config'].get('launch_options', ''))")
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --kernel $KID --options "$KOPT" --no-secondary --no-pmc --no-probes --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace_bench.err); echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/city4k_kernel_stats.csv \;
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_courtyard -- python3 $REPO/bench.py --config courtyard_4k --kernel 8 --no-secondary --no-pmc --no-probes --no-cpu-baseline > $OUT/trace_bench_courtyard.json 2> $OUT/trace_bench_courtyard.err); echo "trace courtyard rc=$?"
find $OUT/trace_courtyard -name "*kernel_stats.csv" -exec cp {} $OUT/courtyard4k_kernel_stats.csv \;
rm -rf $OUT/trace $OUT/trace_courtyard
for a in "city_4k 8" "courtyard_4k 8" "atrium_1080p 3"; do set -- $a; CFG=$1; K=$2
  timeout -k 10 300 python tools/floor_analysis.py --config $CFG --kernel $K 2>&1 | grep -v "one-triangle" | grep -A6 "BVH\]" > $OUT/floor_analysis_${CFG}_kernel$K.log
  timeout -k 10 300 python tools/long_waves.py --config $CFG --kernel $K 2>&1 | tail -2 > $OUT/long_waves_${CFG}_kernel$K.log
done
for a in "city_4k 8" "courtyard_4k 8" "city_4k_soft16 3"; do set -- $a
  timeout -k 10 300 python tools/stripe_scaling.py --config $1 --kernel $2 2>&1 | grep "stripe(s)" > $OUT/stripe_scaling_$1_kernel$2.log; tail -1 $OUT/stripe_scaling_$1_kernel$2.log | cut -c1-60; done
# ... and what rank r of `bench.py --gpus N` executes since round 4: every stripe tuned on its own dispatch (kernel, share, split table)
for CFG in city_4k courtyard_4k city_4k_soft16; do
  timeout -k 10 400 python tools/stripe_scaling.py --config $CFG --tune-stripes 2>&1 | grep "stripe(s)" > $OUT/stripe_scaling_${CFG}_tuned_per_stripe.log; tail -1 $OUT/stripe_scaling_${CFG}_tuned_per_stripe.log | cut -c1-120; done
# split tables: the pieces' lives against the tiles' own waves; the tuner's choice on a moved camera / light; kernel stats of the table launch
timeout -k 10 200 python tools/piece_stats.py --config atrium_1080p --kernel 3 --life 33 --end 0.5 --piece 13 --front 0 2>&1 | grep -v "^   " > $OUT/piece_stats_atrium_1080p.log
timeout -k 10 900 python tools/tuning_robustness.py atrium_1080p city_4k courtyard_4k > $OUT/tuning_robustness.log 2>&1; grep -c reuse $OUT/tuning_robustness.log
# ... and by the granularity of the lives a whole-dispatch table is sorted by (per tile / per block of tiles / not at all), with and without XCD squares
timeout -k 10 400 python tools/table_granularity.py courtyard_4k city_4k 2>&1 | grep "^\[" > $OUT/table_granularity.log; KERNEL=3 timeout -k 10 200 python tools/table_granularity.py atrium_1080p 2>&1 | grep "^\[" >> $OUT/table_granularity.log; wc -l $OUT/table_granularity.log
ASPLITS=$(python -c "import json,sys; sys.path.insert(0,'$REPO'); import bench; t=json.load(open('$OUT/bench_atrium_1080p.json'))['config'].get('split_table'); print(bench.splits_arg(t['plan']) if t else '')")
AOPT=$(python -c "import json; print(json.load(open('$OUT/bench_atrium_1080p.json'))['config'].get('launch_options', ''))")
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_atrium -- python3 $REPO/bench.py --config atrium_1080p --kernel 3 --options "$AOPT" --splits "$ASPLITS" --no-secondary --no-pmc --no-probes --no-cpu-baseline > $OUT/trace_bench_atrium.json 2> $OUT/trace_bench_atrium.err); echo "trace atrium rc=$?"
find $OUT/trace_atrium -name "*kernel_stats.csv" -exec cp {} $OUT/atrium1080p_kernel_stats.csv \;
rm -rf $OUT/trace_atrium
timeout -k 10 100 python tools/phase_shares.py courtyard_4k 8 2>&1 | grep -A4 "^\[" > $OUT/phase_shares.log; timeout -k 10 100 python tools/phase_shares.py city_4k 8 2>&1 | grep -A4 "^\[" >> $OUT/phase_shares.log
timeout -k 10 300 python tools/host_path_timing.py > $OUT/host_path_timing.log 2>&1; tail -2 $OUT/host_path_timing.log
