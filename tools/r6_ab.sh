#!/bin/bash
# Round-6 A/B of the one-launch tail (csrc/tail.hip, MI_FUSE_LAST) on ONE box: bash tools/r6_ab.sh [out dir]
#   1. the new parity tests; 2. cfg2 at 32 tasks per call, alternating pairs; 3. the few-task sweep (1, 2, 4, 8 tasks per call) both ways;
#   4. cfg4 / cfg1 both ways; 5. launches per meta-iteration both ways (tools/launch_floor.py on a kernel trace at 1 task per call)
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=$ROOT/${1:-gpurun_out/r6_ab}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_full_size.py -m gpu -q -k "fused_last_block" > $O/tests_fused_last.txt 2>&1; tail -3 $O/tests_fused_last.txt
bash tools/r5_ab_env.sh ${O#$ROOT/} MI_FUSE_LAST "0 1" 2>&1 | tee $O/ab_cfg2.txt
for V in 0 1; do
  MI_FUSE_LAST=$V timeout -k 10 300 python tools/t_sweep.py --tasks 1,2,4,8,32 --steps 30 --out $O/t_sweep_fuse_last_$V.md > $O/t_sweep_$V.log 2>&1; echo "MI_FUSE_LAST=$V"; cat $O/t_sweep_$V.log
done
for W in cfg4 cfg1; do
  for V in 0 1 0 1; do
    MI_FUSE_LAST=$V timeout -k 10 200 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-pipe --no-secondary --no-clock --no-other --no-sampled > $O/bench_${W}_$V.json 2> $O/bench_${W}_$V.err
    python - $O/bench_${W}_$V.json $W $V <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} MI_FUSE_LAST={sys.argv[3]}: {d['ms_per_step']} ms/step, {d['value']} tasks/s")
PY
  done
done
