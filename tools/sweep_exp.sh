#!/bin/bash
# Diagnostic builds of the fused sweep (-DSW_EXP=N, csrc/policy_sweep.hip) and tools/sweep_stamps.py on each: what the cold start of a sweep waits for.
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
O=$ROOT/${1:-gpurun_out/sweep_exp}; mkdir -p $O /tmp/swx
cd $ROOT/exploring_meta_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
FLA="${FLAGS/-mllvm -amdgpu-mfma-vgpr-form=1/}"
OBJS=""
for f in conv_mfma wgrad_bf16 block1 gram bn_pool head misc engine policy gae test_entry; do
  if [ $f = wgrad_bf16 ]; then FL="$FLA"; else FL="$FLAGS"; fi
  /opt/rocm/bin/hipcc $FL -c $f.hip -o /tmp/swx/$f.o 2>> $O/build.log &
  OBJS="$OBJS /tmp/swx/$f.o"
done
for X in ${EXPS:-0 1 2 3}; do /opt/rocm/bin/hipcc $FLA -DSW_EXP=$X -c policy_sweep.hip -o /tmp/swx/policy_sweep_$X.o 2>> $O/build.log & done
wait
cd $ROOT
for X in ${EXPS:-0 1 2 3}; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/swx/policy_sweep_$X.o $OBJS -o /tmp/swx/libmi_maml_sw$X.so 2>> $O/build.log || { echo "build $X failed"; tail -n 20 $O/build.log; exit 1; }
  echo "== SW_EXP=$X"
  MI_MAML_LIB=/tmp/swx/libmi_maml_sw$X.so timeout -k 10 200 python3 tools/sweep_stamps.py 2>&1 | grep -v amdgpu.ids | head -n ${LINES_PER:-12}
done | tee $O/sweep_exp.txt
