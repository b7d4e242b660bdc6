cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_b1
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -q -x > gpurun_out/r4_b1/engine.log 2>&1; echo "engine rc=$?"
tail -n 3 gpurun_out/r4_b1/engine.log
timeout -k 10 300 python tools/t_sweep.py --workload cfg2 --tasks 1,4,32 --out gpurun_out/r4_b1/t_sweep.md > gpurun_out/r4_b1/t_sweep.log 2>&1; cat gpurun_out/r4_b1/t_sweep.md | tail -4
