#!/bin/bash
# Round 6: kernel traces of the cfg2 meta-iteration at 1 / 4 / 32 tasks per call -> launches per iteration, kernels by total time (tools/launch_floor.py)
#   bash tools/r6_floor.sh [out dir] ; TS="1 4 32"
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box}
O=$ROOT/${1:-gpurun_out/r6_floor}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for T in ${TS:-1 4 32}; do
  timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $O/trace_T$T -- python3 $ROOT/tools/t_sweep.py --workload ${WL:-cfg2} --tasks $T --steps 6 > $O/sweep_T$T.log 2>&1
  python3 $ROOT/tools/launch_floor.py $O/trace_T$T --tasks $T > $O/launch_floor_${WL:-cfg2}_T$T.txt 2>&1
  head -42 $O/launch_floor_${WL:-cfg2}_T$T.txt
  rm -rf $O/trace_T$T
done
