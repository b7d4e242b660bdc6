#!/bin/bash
# Issue / wait counters of the stride-1 hidden convolutions and weight gradients at cfg2 in the operand form in force (rocprofv3 --pmc, one
# pass per counter set; run on the GPU box: bash tools/pmc_conv_issue.sh [out dir]).  Per launch, whole chip, averaged over the launches of a grid.
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=$ROOT/${1:-gpurun_out/pmc_conv}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
N=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD"; do
  N=$((N + 1))
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/p_$N -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-fp32-pipe --no-secondary --pool 2 > $O/log_$N.txt 2>&1; echo "pass $N ($SET) rc=$?"
  f=$(find $O/p_$N -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && python3 - "$f" > $O/table_$N.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not any(s in k for s in ('conv3x3_s1_mfma_kernel<32', 'conv3x3_s1_b16_kernel<32', 'wgrad3x3_strip', 'wgrad3x3_rows_bf16')):
        continue
    acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), cs in sorted(acc.items()):
    print(k, 'grid', g, ' '.join(f'{c}={sum(v)/len(v):.4g} (n={len(v)})' for c, v in sorted(cs.items())))
PY
  rm -rf $O/p_$N
  cat $O/table_$N.txt
done
