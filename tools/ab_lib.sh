#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: the cfg2 meta-iteration with MI_MAML_LIB = <base .so> and with the tree's own.
#   bash tools/ab_lib.sh <base .so (path inside the repo)> [out dir] [extra bench args]
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
BASE=$ROOT/${1:?base library}
O=$ROOT/${2:-gpurun_out/ab_lib}; mkdir -p $O
shift; shift || true
for R in 1 2 3; do
  for V in base new; do
    if [ $V = base ]; then export MI_MAML_LIB=$BASE; else unset MI_MAML_LIB; fi
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-pipe --no-secondary --no-clock --no-other --no-sampled "$@" --breakdown $O/breakdown_${V}_r$R.csv > $O/bench_${V}_r$R.json 2> $O/bench_${V}_r$R.err
    python - $O/bench_${V}_r$R.json $V $R <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(f"{sys.argv[2]} round {sys.argv[3]}: {d['ms_per_step']} ms/step, {d['value']} tasks/s; dominant {r.get('kernel')} {r.get('avg_ms')} ms frac {r.get('frac')}")
PY
  done
done
for V in base new; do python tools/roofline_table.py $O/breakdown_${V}_r3.csv > $O/roofline_table_$V.md 2>/dev/null; done
