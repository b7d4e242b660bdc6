set -e
cd /tmp && export TMPDIR=/tmp
for T in ${TS:-4}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3k/trace_T$T -- python3 $GRAFT_REPO_ROOT/tools/t_sweep.py --workload ${WL:-cfg2} --tasks $T --steps 5 --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/r3k/sweep_T$T.log 2>&1
done
