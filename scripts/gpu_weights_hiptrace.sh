# HIP API time of the weight build (where the host side of K2 spends its time): bash scripts/gpu_weights_hiptrace.sh <tag> <name> <counts>
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r05}; N=${2:-weights_hiptrace}; C=${3:-65}
mkdir -p gpurun_out/$R
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/$R/$N -- python3 tools/weights_scaling.py hiptrace $C > gpurun_out/$R/$N.log 2>&1
python3 - $R $N <<'PY'
import csv, glob, sys
R, N = sys.argv[1], sys.argv[2]
out = open(f'gpurun_out/{R}/{N}.txt', 'w')
print(open(f'gpurun_out/{R}/{N}.log').read().strip().split('\n')[-1], file=out)
for pat in ('*hip_api_stats.csv', '*kernel_stats.csv'):
    fn = glob.glob(f'gpurun_out/{R}/{N}/**/{pat}', recursive=True)
    if not fn:
        continue
    for r in list(csv.DictReader(open(fn[0])))[:12]:
        print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} total {int(r['TotalDurationNs']) / 1e6:9.2f} ms avg {float(r['AverageNs']) / 1e6:9.3f} ms  {float(r['Percentage']):5.1f} %", file=out)
    print(file=out)
out.close()
print(open(f'gpurun_out/{R}/{N}.txt').read())
PY
rm -rf gpurun_out/$R/$N
