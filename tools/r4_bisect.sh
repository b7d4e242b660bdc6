cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b2; mkdir -p $O
for D in 0 1 2 4 8 15; do
  MI_ADV_DBG=$D timeout -k 10 200 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --breakdown $O/bd_cfg4_$D.csv > $O/cfg4_$D.json 2> $O/cfg4_$D.err
  echo "dbg=$D $(grep '^misc,2' $O/bd_cfg4_$D.csv) | $(python3 -c "import json;d=json.loads(open('$O/cfg4_$D.json').read().splitlines()[-1]);print(d['ms_per_step'])")"
done
timeout -k 10 200 python bench.py --tasks 4 --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --breakdown $O/bd_cfg2_T4.csv > $O/cfg2_T4.json 2> $O/cfg2_T4.err
timeout -k 10 200 python bench.py --tasks 1 --steps 20 --warmup 3 --no-cpu-baseline --no-clock --no-dist --breakdown $O/bd_cfg2_T1.csv > $O/cfg2_T1.json 2> $O/cfg2_T1.err
head -30 $O/bd_cfg2_T4.csv
