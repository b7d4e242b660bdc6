#!/bin/bash
# Round-5 A/B of one environment switch on ONE box: bash tools/r5_ab_env.sh OUT VAR "v0 v1" [bench args] -- cfg2 (or --workload ...) alternating, two rounds
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=$ROOT/$1; VAR=$2; VALS=$3; shift 3
mkdir -p $O
for R in 1 2; do
  for V in $VALS; do
    env $VAR=$V timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-pipe --no-secondary --no-clock --no-other --no-sampled "$@" > $O/bench_${VAR}_${V}_r$R.json 2> $O/bench_${VAR}_${V}_r$R.err
    python - $O/bench_${VAR}_${V}_r$R.json $VAR $V $R <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
iso = {(x['op'], x['block']): x['avg_launch_ms'] for x in r.get('single_stream_step', [])}
pick = ' '.join(f"{k[0]}@{k[1]}={v}" for k, v in iso.items() if ('conv' in k[0] or 'dgrad' in k[0]) and k[1] == 2)
print(f"{sys.argv[2]}={sys.argv[3]} round {sys.argv[4]}: {d['ms_per_step']} ms/step; dominant frac {r.get('frac')}; isolated: {pick}")
PY
  done
done
