#!/bin/bash
# What block 1's split-form kernels would cost with the patch planes delivered ready-made (x split once per call by the producer instead of per tile):
# form 2 (default), form 1, and form 1 of a build with -DMI_B1_ABLATE_XSPLIT (wrong results; timing only), isolated launch averages of one box.
set -u
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box}
cd $ROOT
O=$ROOT/${1:-gpurun_out/b1_xsplit}; mkdir -p $O
run() {  # tag, form
  env MI_B1_BF16X3=$2 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pipe --no-secondary --no-clock --no-other --no-sampled --no-dist > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
iso = {(x['op'], x['block']): x['avg_launch_ms'] for x in d['roofline'].get('single_stream_step', [])}
print(sys.argv[2], d['ms_per_step'], 'ms/step;', ' '.join(f"{k[0]}@{k[1]}={v}" for k, v in iso.items() if k[1] == 1))
PY
}
run form2_a 2 && run form1_a 1
L=exploring_meta_amd/csrc/libmi_maml.so     # (the probe build is made on the build host: hipcc ... -DMI_B1_ABLATE_XSPLIT -c block1.hip, linked with the other objects)
cp $L $O/lib.keep && cp scratch_libmi_maml_noxsplit.so $L && run form1_noxsplit 1 && run form2_noxsplit 2
cp $O/lib.keep $L && rm $O/lib.keep
run form2_b 2 && run form1_b 1
# write-only and read-only HBM rates of this box (block 1's forward writes 406 MB and reads 77 MB per launch)
python - <<'PY'
import torch
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
def t(f, n=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ms = t(lambda: x.fill_(1.0)); print(f'fill 1 GiB: {ms:.3f} ms = {x.numel() * 4 / ms / 1e6:.0f} GB/s written')
y = x[: 100 * 1024 * 1024]
ms = t(lambda: y.fill_(2.0)); print(f'fill 400 MiB: {ms:.3f} ms = {y.numel() * 4 / ms / 1e6:.0f} GB/s written')
ms = t(lambda: x.sum()); print(f'sum 1 GiB: {ms:.3f} ms = {x.numel() * 4 / ms / 1e6:.0f} GB/s read')
PY
