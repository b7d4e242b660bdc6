#!/bin/bash
# rocprofv3 kernel stats of block 1's sparse weight-gradient kernels inside the cfg2 step, split-bf16 form on (1) / off (0)
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box}
O=$ROOT/${1:-gpurun_out/sparse_stats}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in ${FORMS:-0 1}; do
  MI_SPARSE_WGRAD_BF16=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$V -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-fp32-pipe --no-secondary --pool 2 > $O/log_$V.txt 2>&1
  f=$(find $O/p_$V -name "*kernel_stats.csv" | head -n 1)
  echo "MI_SPARSE_WGRAD_BF16=$V"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'sparse_wgrad' in r['Name'] or 'block1_kernel<3, 3>' in r['Name'] or 'block1_fwd' in r['Name']: print(r['Name'][:70], r['Calls'], 'avg_us', float(r['AverageNs'])/1e3, 'min_us', float(r['MinNs'])/1e3)
"
  rm -rf $O/p_$V
done
