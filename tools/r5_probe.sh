#!/bin/bash
# Round-5: build and run tools/conv_sched_probe.hip on the GPU box (bash tools/r5_probe.sh [out dir] [extra hipcc flags]).
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=$ROOT/${1:-gpurun_out/r5_probe}; mkdir -p $O /tmp/r5probe
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Iexploring_meta_amd/csrc -Iinclude ${2:-} \
  tools/conv_sched_probe.hip -o /tmp/r5probe/conv_sched_probe 2> $O/build.log || { echo "probe build failed"; tail -n 20 $O/build.log; exit 1; }
timeout -k 10 400 /tmp/r5probe/conv_sched_probe 1 > $O/conv_sched_probe_relu.txt 2>&1; echo "probe(relu) rc=$?"
timeout -k 10 400 /tmp/r5probe/conv_sched_probe 0 > $O/conv_sched_probe_dense.txt 2>&1; echo "probe(dense) rc=$?"
cat $O/conv_sched_probe_relu.txt
