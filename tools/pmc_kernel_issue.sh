#!/bin/bash
# (PMC_KERNELS=substr,substr: any kernels) Issue / wait counters of the stride-1 hidden convolutions and weight gradients at cfg2 in the operand form in force (rocprofv3 --pmc, one
# pass per counter set; run on the GPU box: bash tools/pmc_conv_issue.sh [out dir]).  Per launch, whole chip, averaged over the launches of a grid.
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=$ROOT/${1:-gpurun_out/pmc_kernels}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
N=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  N=$((N + 1))
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/p_$N -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-fp32-pipe --no-secondary --pool 2 > $O/log_$N.txt 2>&1; echo "pass $N ($SET) rc=$?"
  f=$(find $O/p_$N -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && python3 - "$f" > $O/table_$N.txt <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not any(s in k for s in tuple(os.environ.get('PMC_KERNELS', 'sparse_wgrad_rows_kernel').split(','))):
        continue
    acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), cs in sorted(acc.items()):
    print(k, 'grid', g, ' '.join(f'{c}={sum(v)/len(v):.4g} (n={len(v)})' for c, v in sorted(cs.items())))
PY
  rm -rf $O/p_$N
  cat $O/table_$N.txt
done
