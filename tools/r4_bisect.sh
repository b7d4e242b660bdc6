cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b5; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 3 $O/tests.log
timeout -k 10 300 python tools/t_sweep.py --workload cfg2 --tasks 1,2,4,8,16,32 --out $O/t_sweep.md > $O/t_sweep.log 2>&1; cat $O/t_sweep.md | tail -7
for T in 1 4; do
timeout -k 10 200 python bench.py --tasks $T --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --breakdown $O/bd_cfg2_T$T.csv > $O/cfg2_T$T.json 2> $O/cfg2_T$T.err
done
timeout -k 10 200 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --breakdown $O/bd_cfg4.csv > $O/cfg4.json 2> $O/cfg4.err
python3 -c "import json;d=json.loads(open('$O/cfg4.json').read().splitlines()[-1]);print('cfg4', d['ms_per_step'])"
