cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b10; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -q -x > $O/engine.log 2>&1; echo "engine rc=$?"; tail -n 2 $O/engine.log
for T in 32 4 1; do
timeout -k 10 200 python bench.py --tasks $T --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --no-fp32-pipe --breakdown $O/bd_cfg2_T$T.csv > $O/cfg2_T$T.json 2> $O/cfg2_T$T.err
echo "T=$T $(python3 -c "import json;d=json.loads(open('$O/cfg2_T$T.json').read().splitlines()[-1]);print(d['ms_per_step'])") $(grep '^misc,2' $O/bd_cfg2_T$T.csv)"
done
timeout -k 10 200 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --no-fp32-pipe --breakdown $O/bd_cfg4.csv > $O/cfg4.json 2> $O/cfg4.err
echo "cfg4 $(python3 -c "import json;d=json.loads(open('$O/cfg4.json').read().splitlines()[-1]);print(d['ms_per_step'])") $(grep '^misc,2' $O/bd_cfg4.csv)"
