#!/bin/bash
# The cfg5 step's phase timings (tools/trpo_step_timing.py) and the kernel trace of one step (tools/trpo_trace.py).
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
O=$ROOT/${1:-gpurun_out/trpo_trace}; mkdir -p $O
cd $ROOT
timeout -k 10 300 python3 tools/trpo_step_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/trpo_step_timing.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $ROOT/tools/trpo_step_timing.py --whole-only > $O/trace.log 2>&1; echo "trace rc=$?"
python3 $ROOT/tools/trpo_trace.py $O/trace > $O/trpo_step_trace_cfg5.txt 2>&1
rm -rf $O/trace
tail -n 40 $O/trpo_step_trace_cfg5.txt
