#!/bin/bash
# HBM traffic counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes: MI355X_MICROARCH.md) for every BASELINE workload's
# dominant kernel -> profiles/pmc_traffic.json (bench.py's roofline.traffic) and one table per workload.  Run on the GPU box:
#     bash tools/pmc_all.sh gpurun_out/pmc r3
set -e
OUT=${1:-gpurun_out/pmc}; TAG=${2:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
declare -A PICK
PICK[cfg1]='conv3x3_mfma_kernel<64, 1, 1, 0, 2>::conv_fwd_stats,1'
PICK[cfg3]='conv3x3_s1_(mfma_kernel<64, 1, 1, 0, true|b16_kernel<64, 1, 1, 0>)::conv_fwd_stats,1'
PICK[cfg4]='conv3x3_s1_(mfma_kernel<32, 2, 2, 0, true|b16_kernel<32, 2, 2, 0>)::tangent_conv_fwd,1;;sparse_wgrad_rows_kernel<3, false::wgrad,0;;sparse_wgrad_rows_kernel<3, true::tangent_wgrad,0;;block1_fwd_kernel<3, false[,>]::bn_relu_pool_fwd,0;;block1_fwd_kernel<3, true[,>]::bn_tangent_fwd,0'
PICK[cfg5]='policy_sweep_kernel<100, 0>::fisher_vector_product,0'
PICK[cfg2]='conv3x3_s1_(mfma_kernel<32, 2, 2, 0, true|b16_kernel<32, 2, 2, 0>)::tangent_conv_fwd,1;;sparse_wgrad_rows_kernel<3, false::wgrad,0;;sparse_wgrad_rows_kernel<3, true::tangent_wgrad,0;;block1_fwd_kernel<3, false[,>]::bn_relu_pool_fwd,0;;block1_fwd_kernel<3, true[,>]::bn_tangent_fwd,0'
for W in ${WORKLOADS:-cfg2 cfg1 cfg3 cfg4 cfg5}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/${W}_$C -- python3 $ROOT/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-clock --no-other --no-sampled --no-dist --no-fp32-pipe --no-secondary --pool 2 > $ROOT/$OUT/${W}_$C.log 2>&1
  done
  PARGS=(); IFS=$'\n' read -r -d '' -a PL < <(echo "${PICK[$W]}" | sed 's/;;/\n/g' && printf '\0'); for P in "${PL[@]}"; do PARGS+=(--pick "$P"); done   # ';;' separates picks
  python3 $ROOT/tools/pmc_traffic.py $ROOT/$OUT/${W}_FETCH_SIZE $ROOT/$OUT/${W}_WRITE_SIZE --workload $W --table $ROOT/$OUT/pmc_hbm_traffic_${W}_$TAG.txt --json $ROOT/profiles/pmc_traffic.json "${PARGS[@]}"
  rm -rf $ROOT/$OUT/${W}_FETCH_SIZE $ROOT/$OUT/${W}_WRITE_SIZE
done
cp $ROOT/profiles/pmc_traffic.json $ROOT/$OUT/pmc_traffic.json
