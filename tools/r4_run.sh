#!/bin/bash
# Round-4 GPU-box runs, selected by words:  bash tools/r4_run.sh TAG tests bench2 trace2 benchall sweep pmc2
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
for W in "$@"; do
  case $W in
    tests)
      timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/gpu_tests.log
      tail -n 3 $O/gpu_tests.log
      timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log ;;
    bench2)
      timeout -k 10 400 python bench.py --steps 20 --warmup 5 --breakdown $O/event_breakdown_cfg2.csv > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench2 rc=$?"
      python tools/roofline_table.py $O/event_breakdown_cfg2.csv > $O/roofline_table_cfg2.md 2>/dev/null ;;
    benchall)
      for C in cfg1 cfg3 cfg4; do
        timeout -k 10 300 python bench.py --workload $C --steps 20 --warmup 3 --breakdown $O/event_breakdown_$C.csv > $O/bench_$C.json 2> $O/bench_$C.err; echo "$C rc=$?"
      done
      timeout -k 10 300 python bench.py --workload cfg5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?" ;;
    trace2)
      ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-overlap --pool 2 > $O/prof_cfg2.log 2>&1; echo "trace2 rc=$?" )
      f=$(find $O/prof_cfg2 -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_cfg2.csv
      python3 tools/dominant_kernel_trace.py $O/prof_cfg2 --kernel 'conv3x3_s1_mfma_kernel<32, 2, 2, 0, true, false>' --cycle 3 --bench $O/bench_cfg2.json > $O/rocprofv3_dominant_kernel_cfg2.txt 2>&1
      rm -rf $O/prof_cfg2 ;;
    sweep)
      timeout -k 10 300 python tools/t_sweep.py --workload cfg2 --tasks 1,2,4,8,16,32 --out $O/t_sweep_cfg2.md > $O/t_sweep_cfg2.log 2>&1; echo "sweep rc=$?" ;;
    pmc2)
      WORKLOADS=cfg2 bash tools/pmc_all.sh gpurun_out/$TAG/pmc r4 > $O/pmc2.log 2>&1; echo "pmc2 rc=$?" ;;
    pmc3)
      WORKLOADS=cfg3 bash tools/pmc_all.sh gpurun_out/$TAG/pmc r4 > $O/pmc3.log 2>&1; echo "pmc3 rc=$?" ;;
  esac
done
for C in cfg1 cfg2 cfg3 cfg4 cfg5; do [ -s $O/bench_$C.json ] && python3 -c "
import json
d=json.loads(open('$O/bench_$C.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$C', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', r['op'], r['frac'], 'traffic', r['traffic'], '| fp32_pipe', d.get('fp32_pipe'), '| sec', d.get('secondary',{}).get('ms_per_iteration'), '| acc_last', d.get('post_adapt',{}).get('query_acc_mean_last_step'), '| coll', (d.get('collective') or {}).get('allreduce_us'), '| clock', d.get('clock'))"; done
true
