cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_v6
WORKLOADS="cfg2 cfg3 cfg4" bash tools/pmc_all.sh gpurun_out/r4_v6/pmc r4 > gpurun_out/r4_v6/pmc.log 2>&1; echo "pmc rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4_v6/pmc/pmc_traffic.json')); print({k:v for k,v in d.items() if 'cfg2' in k or 'cfg3' in k or 'cfg4' in k})"
