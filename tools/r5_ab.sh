#!/bin/bash
# Round-5 A/B on ONE box: the cfg2 meta-iteration with the stride-1 convolutions on the 16x16x32 kernel (MI_CONV_B16=1) and on the
# 32x32x16 kernel of round 4 (MI_CONV_B16=0), alternating.   bash tools/r5_ab.sh [out dir] [extra bench args]
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=$ROOT/${1:-gpurun_out/r5_ab}; mkdir -p $O
shift || true
for R in 1 2; do
  for B in 0 1; do
    MI_CONV_B16=$B timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-pipe --no-secondary --no-clock --no-other --no-sampled "$@" --breakdown $O/breakdown_b16_${B}_r$R.csv > $O/bench_b16_${B}_r$R.json 2> $O/bench_b16_${B}_r$R.err
    python - $O/bench_b16_${B}_r$R.json $B $R <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(f"b16={sys.argv[2]} round {sys.argv[3]}: {d['ms_per_step']} ms/step, {d['value']} tasks/s; dominant {r.get('kernel')} {r.get('avg_ms')} ms frac {r.get('frac')}")
PY
  done
done
python tools/roofline_table.py $O/breakdown_b16_1_r2.csv > $O/roofline_table_b16_1.md 2>/dev/null
python tools/roofline_table.py $O/breakdown_b16_0_r2.csv > $O/roofline_table_b16_0.md 2>/dev/null
