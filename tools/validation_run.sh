#!/bin/bash
# One-box validation run of a build: GPU tests, smoke, the five bench lines with per-kernel breakdowns, rocprofv3 kernel stats of cfg2 and
# cfg5, T-sweeps of cfg2 / cfg4.  Everything lands under gpurun_out/<tag>/.      bash tools/validation_run.sh r3_v22 [notests]
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/gpu_tests.log
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
fi
for W in cfg2 cfg1 cfg3 cfg4; do
  timeout -k 10 300 python bench.py --workload $W --steps 20 --warmup 3 --breakdown $O/event_breakdown_$W.csv > $O/bench_$W.json 2> $O/bench_$W.err
done
timeout -k 10 300 python bench.py --workload cfg5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python tools/roofline_table.py $O/event_breakdown_cfg2.csv > $O/roofline_table_cfg2.md 2>/dev/null
timeout -k 10 300 python tools/t_sweep.py --workload cfg2 --tasks 1,2,4,8,16,32 --out $O/t_sweep_cfg2.md > $O/t_sweep_cfg2.log 2>&1
timeout -k 10 300 python tools/t_sweep.py --workload cfg4 --tasks 8,16,32,64,256 --out $O/t_sweep_cfg4.md > $O/t_sweep_cfg4.log 2>&1
cd /tmp && export TMPDIR=/tmp
for W in cfg2 cfg5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 $ROOT/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled > $O/prof_$W.log 2>&1
  f=$(find $O/prof_$W -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_$W.csv
  rm -rf $O/prof_$W
done
tail -n 2 $O/gpu_tests.log $O/smoke.log 2>/dev/null
for W in cfg1 cfg2 cfg3 cfg4 cfg5; do python3 -c "
import json,sys
d=json.loads(open('$O/bench_$W.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$W', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', r['op'], r['frac'], 'traffic', r['traffic'])"; done
