cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b8; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests_all rc=$?"; tail -n 6 $O/tests_all.log
for B in 0 1; do
  if [ $B = 1 ]; then export MI_B1_FP32=1; else unset MI_B1_FP32; fi
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-clock --no-dist --no-fp32-pipe --breakdown $O/bd_b1fp32_$B.csv > $O/cfg2_b1fp32_$B.json 2> $O/err.txt
  echo "b1_fp32=$B $(python3 -c "import json;d=json.loads(open('$O/cfg2_b1fp32_$B.json').read().splitlines()[-1]);print(d['ms_per_step'])") $(grep -E '^bn_relu_pool_fwd,0|^bn_tangent_fwd,0' $O/bd_b1fp32_$B.csv | tr '\n' ' ')"
done
