#!/bin/bash
# Block 1's sparse weight gradient on the split-bf16 form (MI_SPARSE_WGRAD_BF16=1) against fp32-input MFMAs (=0): kernel tests, then alternating bench pairs on ONE box
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box}
cd $ROOT
O=$ROOT/${1:-gpurun_out/sparse_ab}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tangent_kernels.py -q -m gpu -p no:cacheprovider -k "block1_kernels" -x > $O/tests_block1.log 2>&1; echo "block1 kernel tests rc=$?"; tail -n 3 $O/tests_block1.log
run() {  # tag, value, workload args
  env MI_SPARSE_WGRAD_BF16=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-pipe --no-secondary --no-clock --no-other --no-sampled --no-dist "${@:3}" > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
iso = {(x['op'], x['block']): x['avg_launch_ms'] for x in d['roofline'].get('single_stream_step', [])}
print(sys.argv[2], d['ms_per_step'], 'ms/step;', ' '.join(f"{k[0]}@{k[1]}={v}" for k, v in iso.items() if k[1] == 1 and 'wgrad' in k[0]))
PY
}
for R in 1 2 3; do run cfg2_fp32_$R 0 && run cfg2_bf16_$R 1; done
for R in 1 2; do run cfg4_fp32_$R 0 --workload cfg4 && run cfg4_bf16_$R 1 --workload cfg4; done
run cfg3_fp32 0 --workload cfg3 && run cfg3_bf16 1 --workload cfg3
