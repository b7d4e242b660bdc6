cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b13; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -q -x -k "teacher_forced" --durations=5 > $O/tf.log 2>&1; echo "tf rc=$?"; tail -n 12 $O/tf.log
