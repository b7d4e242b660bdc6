cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b3; mkdir -p $O
for rep in 1 2; do
for W in 0 1; do
  if [ $W = 1 ]; then export MI_NO_WPL=1; else unset MI_NO_WPL; fi
  for T in 4 32; do
    timeout -k 10 200 python bench.py --tasks $T --steps 30 --warmup 5 --no-cpu-baseline --no-clock --no-dist > $O/cfg2_T${T}_nowpl$W.json 2> $O/err.txt
    echo "rep=$rep nowpl=$W T=$T $(python3 -c "import json;d=json.loads(open('$O/cfg2_T${T}_nowpl$W.json').read().splitlines()[-1]);print(d['ms_per_step'])")"
  done
  timeout -k 10 200 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline --no-clock --no-dist > $O/cfg4_nowpl$W.json 2> $O/err.txt
  echo "rep=$rep nowpl=$W cfg4 $(python3 -c "import json;d=json.loads(open('$O/cfg4_nowpl$W.json').read().splitlines()[-1]);print(d['ms_per_step'])")"
done
done
