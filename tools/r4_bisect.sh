cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_acc
timeout -k 10 800 python tools/accuracy_parity.py --tasks 256 --out gpurun_out/r4_acc/accuracy_parity_cfg2 > gpurun_out/r4_acc/log.txt 2>&1; echo "acc rc=$?"; tail -n 4 gpurun_out/r4_acc/log.txt; ls gpurun_out/r4_acc
MI_B1_BF16X3=1 timeout -k 10 800 python tools/accuracy_parity.py --tasks 256 --out gpurun_out/r4_acc/accuracy_parity_cfg2_block1_bf16 > gpurun_out/r4_acc/log_bf.txt 2>&1; echo "acc bf rc=$?"; tail -n 2 gpurun_out/r4_acc/log_bf.txt
