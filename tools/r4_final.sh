#!/bin/bash
# Round-4 validation run (one box, one call): bash tools/r4_final.sh TAG
TAG=${1:-r4_final}
cd $GRAFT_REPO_ROOT
bash tools/r4_run.sh $TAG tests bench2 trace2 benchall sweep
O=gpurun_out/$TAG
timeout -k 10 300 python tools/t_sweep.py --workload cfg4 --tasks 8,16,32,64,256 --out $O/t_sweep_cfg4.md > $O/t_sweep_cfg4.log 2>&1
for T in 4 1; do
  timeout -k 10 200 python bench.py --tasks $T --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-fp32-pipe --breakdown $O/event_breakdown_cfg2_T$T.csv > $O/bench_cfg2_T$T.json 2> $O/bench_cfg2_T$T.err
done
# the opt-in split-bf16 form of block 1's lean forward kernels, same box
MI_B1_BF16X3=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-clock --no-other --no-sampled --breakdown $O/event_breakdown_cfg2_b1bf16.csv > $O/bench_cfg2_b1bf16.json 2> $O/bench_cfg2_b1bf16.err
WORKLOADS="cfg2 cfg3" bash tools/pmc_all.sh gpurun_out/$TAG/pmc r4 > $O/pmc.log 2>&1; echo "pmc rc=$?"
( cd /tmp && export TMPDIR=/tmp && for W in cfg1 cfg3 cfg4 cfg5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$W -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --pool 2 > $GRAFT_REPO_ROOT/$O/prof_$W.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$W -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/rocprofv3_kernel_stats_$W.csv
  rm -rf $GRAFT_REPO_ROOT/$O/prof_$W
done )
python3 -c "
import json
d=json.loads(open('$O/bench_cfg2_b1bf16.json').read().strip().splitlines()[-1]); print('cfg2 with MI_B1_BF16X3=1:', d['ms_per_step'], 'ms', d['value'])"
