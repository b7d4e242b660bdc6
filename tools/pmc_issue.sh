#!/bin/bash
# Issue counters of the block-1 kernels and the dominant convolutions (run on the GPU box: bash tools/pmc_issue.sh [out dir]).
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
cd $ROOT
O=$ROOT/${1:-gpurun_out/pmc_issue}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/avail.txt | sort -u | grep -E "INSTS_VALU|MFMA|BUSY_CYCLES|WAVE_CYCLES|ACTIVE_INST|INSTS_LDS|INSTS_VMEM|INSTS_SALU|WAIT_INST|INST_CYCLES" | tr '\n' ' ' > $O/sq_names.txt
cat $O/sq_names.txt
for SET in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  T=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/p_$T -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-fp32-pipe --no-secondary --pool 2 > $O/log_$T.txt 2>&1; echo "pass $T rc=$?"
  f=$(find $O/p_$T -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && python3 - "$f" > $O/table_$T.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not any(s in k for s in ('block1_fwd_kernel', 'sparse_wgrad_rows_kernel', 'conv3x3_s1_mfma_kernel<32, 2, 2, 0, true>', 'conv3x3_s1_mfma_kernel<32, 1, 1, 0, true>', 'conv3x3_s1_b16_kernel<32, 2, 2, 0>', 'conv3x3_s1_b16_kernel<32, 1, 1, 0>')):
        continue
    acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), cs in sorted(acc.items()):
    print(k, 'grid', g, ' '.join(f'{c}={sum(v)/len(v):.4g} (n={len(v)})' for c, v in sorted(cs.items())))
PY
  rm -rf $O/p_$T
  cat $O/table_$T.txt
done
