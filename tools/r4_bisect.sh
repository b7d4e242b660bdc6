cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -q -x > $O/engine.log 2>&1; echo "engine rc=$?"
tail -n 3 $O/engine.log
timeout -k 10 300 python tools/t_sweep.py --workload cfg2 --tasks 1,4,32 --out $O/t_sweep.md > $O/t_sweep.log 2>&1; cat $O/t_sweep.md | tail -4
timeout -k 10 200 python bench.py --tasks 4 --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --breakdown $O/bd_cfg2_T4.csv > $O/cfg2_T4.json 2> $O/cfg2_T4.err
grep "^misc,2" $O/bd_cfg2_T4.csv
timeout -k 10 200 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --breakdown $O/bd_cfg4.csv > $O/cfg4.json 2> $O/cfg4.err
grep "^misc,2" $O/bd_cfg4.csv; python3 -c "import json;d=json.loads(open('$O/cfg4.json').read().splitlines()[-1]);print('cfg4', d['ms_per_step'])"
