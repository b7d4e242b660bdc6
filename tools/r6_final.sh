#!/bin/bash
# Round-6 validation run on ONE box, in two calls (each inside gpurun's 20-minute limit):
#     bash tools/r6_final.sh TAG a     GPU tests, smoke, cfg2 bench + per-kernel breakdown, rocprofv3 trace of cfg2 with the per-launch extraction of the
#                                      dominant kernel, the other four bench lines, the cfg2 task-count sweep
#     bash tools/r6_final.sh TAG b     HBM traffic counters (cfg2, cfg3), SQ issue counters of the hidden convolutions, cfg2 at 4 / 1 tasks per call,
#                                      the cfg4 task-count sweep, rocprofv3 kernel stats of the other workloads, kernel traces at 1 / 4 / 32 tasks per call
#                                      (launches per iteration), the stage stamps of the one-launch tail
set -u
TAG=${1:-r6_final}; PART=${2:-a}
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=gpurun_out/$TAG; mkdir -p $O
DOM='conv3x3_s1_b16_kernel<32, 2, 2, 0>'
if [ "$PART" = a ]; then
  timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/gpu_tests.log; tail -n 3 $O/gpu_tests.log
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --breakdown $O/event_breakdown_cfg2.csv > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench2 rc=$?"
  python tools/roofline_table.py $O/event_breakdown_cfg2.csv > $O/roofline_table_cfg2.md 2>/dev/null
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_cfg2 -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-overlap --pool 2 > $ROOT/$O/prof_cfg2.log 2>&1; echo "trace2 rc=$?" )
  f=$(find $O/prof_cfg2 -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_cfg2.csv
  python3 tools/dominant_kernel_trace.py $O/prof_cfg2 --kernel "$DOM" --cycle 2 --bench $O/bench_cfg2.json > $O/rocprofv3_dominant_kernel_cfg2.txt 2>&1
  rm -rf $O/prof_cfg2
  for C in cfg1 cfg3 cfg4; do
    timeout -k 10 300 python bench.py --workload $C --steps 20 --warmup 3 --breakdown $O/event_breakdown_$C.csv > $O/bench_$C.json 2> $O/bench_$C.err; echo "$C rc=$?"
  done
  timeout -k 10 300 python bench.py --workload cfg5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
  timeout -k 10 300 python tools/t_sweep.py --workload cfg2 --tasks 1,2,4,8,16,32 --out $O/t_sweep_cfg2.md > $O/t_sweep_cfg2.log 2>&1; echo "sweep rc=$?"
else
  WORKLOADS="cfg2 cfg3" bash tools/pmc_all.sh gpurun_out/$TAG/pmc r6 > $O/pmc.log 2>&1; echo "pmc rc=$?"
  bash tools/pmc_conv_issue.sh gpurun_out/$TAG/pmc_conv > $O/pmc_conv.log 2>&1; echo "pmc_conv rc=$?"
  for T in 4 1; do
    timeout -k 10 200 python bench.py --tasks $T --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-fp32-pipe --breakdown $O/event_breakdown_cfg2_T$T.csv > $O/bench_cfg2_T$T.json 2> $O/bench_cfg2_T$T.err
  done
  timeout -k 10 300 python tools/t_sweep.py --workload cfg4 --tasks 8,16,32,64,256 --out $O/t_sweep_cfg4.md > $O/t_sweep_cfg4.log 2>&1
  ( cd /tmp && export TMPDIR=/tmp && for W in cfg1 cfg3 cfg4 cfg5; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_$W -- python3 $ROOT/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --pool 2 > $ROOT/$O/prof_$W.log 2>&1
    f=$(find $ROOT/$O/prof_$W -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $ROOT/$O/rocprofv3_kernel_stats_$W.csv
    rm -rf $ROOT/$O/prof_$W
  done )
  TS="1 4 32" bash tools/r6_floor.sh $O/floor > $O/floor.log 2>&1; echo "floor rc=$?"; grep "per iteration (mean" $O/floor.log
  timeout -k 10 120 python tools/tail_stamps.py --tasks 1 > $O/tail_stamps_T1.txt 2>&1; timeout -k 10 120 python tools/tail_stamps.py --tasks 32 > $O/tail_stamps_T32.txt 2>&1
fi
for C in cfg1 cfg2 cfg3 cfg4 cfg5; do [ -s $O/bench_$C.json ] && python3 -c "
import json
d=json.loads(open('$O/bench_$C.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$C', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', r.get('kernel'), r['frac'], 'traffic', r['traffic'], '| fp32_pipe', (d.get('fp32_pipe') or {}).get('ms_per_step'), '| sec', (d.get('secondary') or {}).get('ms_per_iteration'), '| clock', d.get('clock'))"; done
true
