#!/bin/bash
# Round-5: the diagnostic build of the library with phase stamps in the 16x16x32 convolution, and tools/conv_b16_stamps.py on it.
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
O=$ROOT/${1:-gpurun_out/r5_stamps}; mkdir -p $O /tmp/r5st
cd $ROOT/exploring_meta_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=""
for f in wgrad_bf16 block1 gram bn_pool head misc engine policy policy_sweep gae test_entry; do      # (*.o do not travel to the box: rebuilt here, in parallel)
  if [ $f = policy_sweep ] || [ $f = wgrad_bf16 ]; then FL="${FLAGS/-mllvm -amdgpu-mfma-vgpr-form=1/}"; else FL="$FLAGS"; fi
  /opt/rocm/bin/hipcc $FL -c $f.hip -o /tmp/r5st/$f.o 2>> $O/build.log &
  OBJS="$OBJS /tmp/r5st/$f.o"
done
for X in ${EXPS:-0}; do
  /opt/rocm/bin/hipcc $FLAGS -DMI_B16_STAMPS -DMI_B16_EXP=$X -c conv_mfma.hip -o /tmp/r5st/conv_mfma_$X.o 2>> $O/build.log &
done
wait
cd $ROOT
for X in ${EXPS:-0}; do            # MI_B16_EXP: 0 = the kernel as shipped; 1 no epilogue stores, 2 no epilogue statistics, 3 neither (timing only)
  ( /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/r5st/conv_mfma_$X.o $OBJS -o /tmp/r5st/libmi_maml_stamps_$X.so 2>> $O/build.log ) || { echo "stamps build $X failed"; tail -n 20 $O/build.log; exit 1; }
  for S in ${STAGGERS:-0}; do
    echo "== MI_B16_EXP=$X MI_CONV_STAGGER=$S"
    MI_CONV_STAGGER=$S MI_MAML_LIB=/tmp/r5st/libmi_maml_stamps_$X.so timeout -k 10 300 python tools/conv_b16_stamps.py 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/conv_b16_stamps.txt
