#!/bin/bash
# Round-5: kernel traces of the cfg2 meta-iteration at 1 / 4 / 32 tasks per call and their launch-floor summaries (tools/launch_floor.py).
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
O=$ROOT/${1:-gpurun_out/r5_floor}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for T in ${TS:-1 4 32}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_T$T -- python3 $ROOT/tools/t_sweep.py --workload cfg2 --tasks $T --steps 6 > $O/sweep_T$T.log 2>&1; echo "trace T=$T rc=$?"
  python3 $ROOT/tools/launch_floor.py $O/trace_T$T --tasks $T > $O/launch_floor_cfg2_T$T.txt 2>&1
  rm -rf $O/trace_T$T
  head -n 8 $O/launch_floor_cfg2_T$T.txt
done
